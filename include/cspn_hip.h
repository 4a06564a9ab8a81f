/*
 * cspn_hip.h — C ABI of libcspn_hip.so, the MI355X (gfx950) CSPN affinity-propagation engine.
 *
 * This is the drop-in boundary for the reference's hot path (paths relative to the reference repo):
 *   network/libs/post_process/CSPN_new.py:26-128   AffinityPropagate.forward (3x3, 24 steps)
 *   network/libs/post_process/CSPN_ours.py:24-54   AffinityPropagate.forward (K x K softmax / PAC)
 *   network/libs/base/pac.py:75-121                Conv2dFn.forward / backward
 *
 * Conventions (modelled on the reference's only native interface, In-Place ABN:
 * network/libs/inplace_abn/src/bn.h:7-19, bn.cu:237-300, lib_cffi.cpp:1-2,36-63):
 *   - every entry point returns 1 on success and 0 on failure (bn.cu:249 `return 1` after the
 *     cudaGetLastError check); on failure cspn_last_error() returns a thread-local message;
 *   - "all functions assume input and output tensors are already initialised and have the
 *     correct dimensions" (lib_cffi.cpp:1-2): the caller validates shapes / contiguity;
 *   - optional tensors are passed as NULL (lib_cffi.cpp:62-63 maps empty tensors to NULL);
 *   - the caller owns every buffer (including workspace); the library never allocates,
 *     frees or retains device pointers, enqueues on the given stream and does not synchronise;
 *   - no global mutable state in the library (beyond the thread-local error string and caches of immutable device
 *     properties); the caller selects the device (hipSetDevice) before the call, as `with torch.cuda.device(...)` does at
 *     encoding.py:168.  The streaming entry points are re-entrant: any number of host threads, devices and streams at
 *     once.  The WEIGHT-RESIDENT entry points (cspn3_forward_resident, cspn3_transposed_resident, cspnk_forward_resident*)
 *     are thread-safe but put a protocol on the caller, stated with each of them: one such launch at a time per DEVICE
 *     (their workgroups wait for each other), a zero-initialised workspace that only these entries ever touch, a `seq`
 *     that grows by at least 256 per call on a workspace, two pinned host words for the time-out / completion reports.
 *     cspn_monodepth_amd/functional.py implements that protocol (a lock per device, a journal that repairs a timed-out
 *     inference call on the streaming schedule); a host that cannot keep to it uses the streaming entries, which give
 *     the same bits.
 *
 * Scope of the 3x3 entries: the reference's CSPN_new.AffinityPropagate is only meaningful for prop_kernel = 3 (with 5 its
 * ones-kernel becomes 1x2x2 and the output silently shrinks to (H-1)x(W-1), CSPN_new.py:122); the host module raises
 * for prop_kernel != 3 and K x K windows (K = 3, 5, 7) are served by the softmax / PAC entries (cspn_pac_prepare +
 * cspn_propagate with K), which is what the reference's CSPN_ours module computes.
 *
 * Tensor layout: NCHW contiguous planes.  "taps" are the K*K-1 non-centre offsets (dy,dx) in
 * row-major order over [-K/2, K/2]^2; tap j of a weight volume multiplies depth[p + off_j].
 *
 * Row padding.  The 16-byte quad kernels need W % 4 == 0.  Images of any other width are handled by the caller
 * zero-padding every row to the next multiple of 4 and passing the true width as `W_valid` (0 or W = no padding):
 * columns [W_valid, W) are then treated exactly like pixels outside the image — zero gate, zero depth, forced to
 * 0 after every step — which is the reference's own boundary condition, so in-image results are unchanged.
 * (Plain zero padding WITHOUT W_valid is not equivalent: pad pixels with an all-zero neighbourhood become 0/0.)
 * Outputs: the padding columns of the last valid quad, [W_valid, round_up4(W_valid)), receive zeros (also from the guard's
 * re-computation of a timed-out resident launch); columns beyond that quad are not written at all.
 * With W % 4 != 0 and no padding the library falls back to generic one-pixel-per-thread kernels.
 *
 * Tap-volume layout (the `w8` / `wk` / `wT` buffers; produced and consumed only by this library):
 *   CSPN_F32: planar [B, NT, H, W], NT = K*K-1.
 *   CSPN_F16: tap PAIRS interleaved per 4-pixel quad, [B, NT/2, ceil(H*W/4), 2, 4] halfs (B*NT*ceil4(H*W)
 *             elements), so one 16-byte load yields taps (2i, 2i+1) of a quad: 8-byte loads stream at about
 *             half the per-byte rate on gfx950.
 */
#ifndef CSPN_HIP_H_
#define CSPN_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSPN_ABI_VERSION 10

/* What a tile of a weight-resident launch that gave up (co-residency time-out) leaves in its part of the result: a quiet NaN with
 * a payload no arithmetic produces.  A consumer sees NaN; the host tells a failed tile from a NaN the recurrence itself produced
 * (0/0 at an all-zero gate pixel, CSPN_new.py:127) by the bit pattern, and re-runs exactly the calls that hold one. */
#define CSPN_POISON_F32 0x7fc0deadu
#define CSPN_POISON_F16 0x7eadu

typedef void* cspn_stream_t; /* hipStream_t */

/* element types */
enum { CSPN_F32 = 0, CSPN_F16 = 1 };

/* blend modes of one propagation step */
enum {
    CSPN_BLEND_NONE = 0,   /* d' = u                                                            */
    CSPN_BLEND_SPARSE = 1, /* d' = (1-m) u + m d0, m = sign(sparse)    CSPN_new.py:90, CSPN_ours.py:53 */
    CSPN_BLEND_PREMASK = 2 /* u = stencil((1-m) d): transposed recurrence used by the backward pass */
};

/* Launch plan of the propagation loop.  All zero (or a NULL plan) = built-in heuristic. */
typedef struct cspn_plan {
    int steps_per_launch; /* S: propagation steps fused into one kernel launch (temporal blocking) */
    int tile_w;           /* interior tile width in pixels, multiple of 4                          */
    int tile_h;           /* interior tile height in pixels                                        */
    int quads_per_thread; /* NQ: vertically consecutive 4-pixel groups owned by one thread         */
    int threads;          /* workgroup size (256 / 512 / 1024)                                     */
    int force_scalar;     /* 1 = use the generic per-pixel kernel (any W, any alignment)           */
} cspn_plan;

int cspn_abi_version(void);
const char* cspn_last_error(void);

/* Effective launch plan for a problem (defaults merged with `plan_or_null`, W%4 fallback applied) and a
 * check that it fits the machine (threads, 160 KiB LDS).  Lets the host size its expectations (number of
 * launches = ceil(T / steps_per_launch)) without duplicating the heuristic. */
int cspn_plan_resolve(int K, int B, int H, int W, int T, int keep_history, const cspn_plan* plan_or_null,
                      cspn_plan* resolved);

/* ---- 3x3, gate-at-the-neighbour, sum-normalised variant (CSPN_new.py) ------------------------ */

/* w8[b][j][p] = |guidance[b][7-j][p+off_j]| / sum_k |guidance[b][k][p+o_k]|   (CSPN_new.py:29-70,:124-127)
 * guidance is read through explicit batch / channel strides (in elements), so the first 8
 * channels of the UNet's 12-channel head (unet_cspn_nyu.py:332) are used in place.
 * w8: [B,8,H,W] of w_dtype.  s_or_null: optional [B,H,W] f32 receiving the normaliser S. */
int cspn3_prepare(const void* guidance, int g_dtype, long g_batch_stride, long g_chan_stride,
                  int B, int H, int W, int W_valid, void* w8, int w_dtype, float* s_or_null, cspn_stream_t stream);

/* ---- K x K, centre-indexed softmax variant (CSPN_ours.py + pac.py) -------------------------- */

/* wk[b][j][p] = softmax_c(guided[b][:, p])[j]  for the K*K-1 taps (CSPN_ours.py:35-41). */
int cspn_pac_prepare(const void* guided, int g_dtype, int B, int H, int W, int K,
                     void* wk, int w_dtype, cspn_stream_t stream);

/* ---- the propagation loop (both variants) ---------------------------------------------------- */

/* Bytes of caller-provided workspace cspn_propagate needs (ping-pong planes; 0 when history is kept). */
size_t cspn_propagate_workspace_bytes(int B, int H, int W, int T, int d_dtype, int keep_history);

/* Runs T steps  d_{t+1}[p] = blend( sum_j w[j][p] * d_t[p+off_j] ).
 *   w       [B,K*K-1,H,W] w_dtype      d0 [B,H,W] d_dtype      sparse (NULL unless blend != NONE)
 *   out     [B,H,W] d_dtype; receives d_T.  Ignored (may be NULL) when history != NULL.
 *   history NULL, or [T,B,H,W] d_dtype receiving d_1..d_T (d_T = history[T-1]) — what backward needs.
 *   work    cspn_propagate_workspace_bytes() bytes, or NULL if that is 0.
 * Replaces the 24x {8 pads, cat, mul, 2 conv3d, div, crop, blend} loop at CSPN_new.py:80-90 and the
 * unfold/mul/einsum loop at CSPN_ours.py:47-53 / pac.py:89-92. */
int cspn_propagate(const void* w, int w_dtype, const void* d0, const void* sparse, void* out,
                   void* history, void* work, int d_dtype, int B, int H, int W, int W_valid, int K, int T,
                   int blend, const cspn_plan* plan, cspn_stream_t stream);

/* cspn_propagate + cspn_metrics_accumulate in one: the launch that produces d_T also accumulates the depth
 * metrics of its pixels against `target` [B,H,W] (d_dtype) into acc[nslots][10] (see cspn_metrics_accumulate),
 * so the refined batch is not read back by a separate reduction pass.  Inference only (no history), prepared
 * tap volume, K in {3,5}, w_dtype == d_dtype, W % 4 == 0 and 16-byte aligned tensors, a plan that has a scoring
 * kernel instance (one quad per thread; also two at 512 threads for K = 3); returns 0 otherwise and the caller
 * uses the two-call form. */
int cspn_propagate_scored(const void* w, int w_dtype, const void* d0, const void* sparse, void* out, void* work,
                          int d_dtype, int B, int H, int W, int W_valid, int K, int T, int blend,
                          const void* target, double* acc, int nslots, const cspn_plan* plan, cspn_stream_t stream);

/* 3x3 variant: cspn3_prepare and cspn_propagate in one — the launch itself derives the normalised weights from
 * the raw guidance (same arithmetic as cspn3_prepare, bit-identical results); no separate prepare pass.
 * Arguments as cspn3_prepare (guidance, strides) + cspn_propagate.
 * w8_out_or_null: NULL = every launch re-derives the weights (no weight volume at all); a [B,8,H,W] tap volume of
 * g_dtype = the FIRST launch derives the weights and also publishes them there, the following launches stream
 * them (one prepare pass and one read of the volume saved).
 * s_out_or_null: optional [B,H,W] f32 receiving the normaliser S (as cspn3_prepare's s_or_null; needs w8_out): with
 * `history` this is the training forward — weights, S and the T depth planes the backward needs, no prepare pass.
 * target/acc/nslots: optional scoring of d_T as in cspn_propagate_scored (needs w8_out, no history, more than
 * one launch, g_dtype == d_dtype).
 * Needs W % 4 == 0 and 16-byte aligned tensors (returns 0 otherwise: use the two-call form).
 * blend: CSPN_BLEND_NONE or CSPN_BLEND_SPARSE.  Replaces all of CSPN_new.py:29-92 for a forward pass. */
int cspn3_propagate_from_guidance(const void* guidance, int g_dtype, long g_batch_stride, long g_chan_stride,
                                  void* w8_out_or_null, float* s_out_or_null, const void* d0, const void* sparse, void* out,
                                  void* history, void* work,
                                  int d_dtype, int B, int H, int W, int W_valid, int T, int blend,
                                  const void* target_or_null, double* acc_or_null, int nslots,
                                  const cspn_plan* plan, cspn_stream_t stream);

/* ---- backward ---------------------------------------------------------------------------------- */

/* The backward recurrence  G_t = stencil^T((1-m) G_{t+1}),  G_T = dL/dout,  on the FORWARD tap volume: every launch
 * reads tap j of the transposed stencil as w[NT-1-j][p+off_j] (aligned quads + lane shifts), so no transposed
 * copy of the weights is made.  g_T [B,H,W] f32; history [T,B,H,W] f32 receives G_{T-1}..G_0; sparse_f32 (with
 * premask != 0) is the sparse depth as f32.  Vector path only (W % 4 == 0, 16-byte aligned; returns 0 otherwise:
 * use cspn_transpose_weights + cspn_propagate(…, CSPN_BLEND_PREMASK)). */
int cspn_propagate_transposed(const void* w, int w_dtype, const float* g_T, const float* sparse_f32, float* history,
                              int B, int H, int W, int W_valid, int K, int T, int premask, const cspn_plan* plan,
                              cspn_stream_t stream);

/* wT[b][j][q] = w[b][NT-1-j][q+off_j] (0 outside): weights of the transposed stencil, so that the
 * backward recurrence G_t = stencilT((1-m) G_{t+1}) runs through cspn_propagate(…, CSPN_BLEND_PREMASK). */
int cspn_transpose_weights(const void* w, void* wT, int w_dtype, int B, int H, int W, int K,
                           cspn_stream_t stream);

/* gw[b][j][p] = (1-m[p]) * sum_{t=0}^{T-1} G_{t+1}[p] * d_t[p+off_j]      (dL/dw, f32)
 * gd0[b][p]   = G_0[p] + m[p] * sum_{t=1}^{T} G_t[p]                       (dL/d coarse depth, f32)
 *   d0 [B,H,W], dhist [T,B,H,W] = d_1..d_T (forward history; plane T-1 is not read),
 *   g_T [B,H,W] f32 = G_T = dL/dout (read in place), ghist [T,B,H,W] f32 in backward order: ghist[s] = G_{T-1-s},
 *   the history written by cspn_propagate_transposed(w, g_T, ..., history = ghist) (NULL when T == 0). */
int cspn_grad_weights(const void* d0, const void* dhist, const float* g_T, const float* ghist, const void* sparse,
                      float* gw, float* gd0, int d_dtype, int B, int H, int W, int K, int T,
                      cspn_stream_t stream);

/* 3x3 variant: dL/dguidance from dL/dw (quotient rule of w = A/S, un-shift, sign(guidance)):
 *   gA_j = (gw_j - sum_k gw_k w_k) / S ;  dL/dg[b][7-j][q] = sign(g) * gA_j[q - off_j]
 * grad_guidance uses the guidance strides; channels >= 8 are zero-filled (never read by the forward). */
int cspn3_grad_guidance(const void* guidance, int g_dtype, long g_batch_stride, long g_chan_stride,
                        int C, const void* w8, int w_dtype, const float* s, const float* gw,
                        void* grad_guidance, int B, int H, int W, cspn_stream_t stream);

/* K x K variant: softmax backward  dL/dguided_c = sm_c (gw_c - sum_k gw_k sm_k). */
int cspn_pac_grad_guided(const void* wk, int w_dtype, const float* gw, void* grad_guided, int g_dtype,
                         int B, int H, int W, int K, cspn_stream_t stream);

/* Fused backward tails (vector path: W % 4 == 0, 16-byte aligned tensors; return 0 otherwise so the caller uses
 * the three-call form above).  One pass over the histories accumulates dL/dw in registers and applies the
 * guidance / softmax epilogue there, so dL/dw never goes to HBM:
 *   cspn3_backward_tail    = cspn_grad_weights + cspn3_grad_guidance   (all tensors of one dtype)
 *   cspn_pac_backward_tail = cspn_grad_weights + cspn_pac_grad_guided
 * cspn3_backward_tail does not read the tap volume (ABI 9): w_j[p] = |g_{7-j}[p + off_j]| / S[p] is rebuilt from the guidance
 * quads its epilogue loads anyway and from s, by the forward's recipe — |g| x (v_rcp_f32(S) + one Newton step): the fp32 taps are the
 * forward's bits for every normaliser in [2^-100, 2^100]; outside that range the forward takes true divisions (cspn_common.hpp:
 * div8_shared_reciprocal) and the rebuilt taps may differ from them in the last bit.  w8_or_null is accepted for source
 * compatibility and ignored. */
int cspn3_backward_tail(const void* d0, const void* dhist, const float* g_T, const float* ghist, const void* sparse,
                        const void* guidance, long g_batch_stride, long g_chan_stride, int C, const void* w8_or_null,
                        const float* s, void* grad_guidance, float* gd0, int dtype, int B, int H, int W, int T,
                        cspn_stream_t stream);
/* gd0_dtype: CSPN_F32, or CSPN_F16 — dL/dx0 rounded to half where it is produced (what a cast of the fp32 plane gives). */
int cspn_pac_backward_tail(const void* d0, const void* dhist, const float* g_T, const float* ghist, const void* sparse,
                           const void* wk, void* grad_guided, void* gd0, int gd0_dtype, int d_dtype, int w_dtype,
                           int B, int H, int W, int K, int T, cspn_stream_t stream);

/* ---- evaluation (libs/metrics.py:49-83, base_model.py:28-73) ------------ */

/* acc[nslots][10] (f64, zero-initialised by the caller): row (block % nslots) += masked sums over target>0 of
 * {inv^2, inv, diff^2, diff, diff/t, |log10 o - log10 t|, #(r<1.25), #(r<1.25^2), #(r<1.25^3), n};
 * the host adds the rows and turns them into irmse, imae, mse, rmse, mae, absrel, lg10, delta1..3 (+ n).
 * Several rows only spread the fp64 atomics (one address serialises at ~45 ns each); nslots = 1 is valid. */
int cspn_metrics_accumulate(const void* pred, const void* target, int dtype, size_t n,
                            double* acc, int nslots, cspn_stream_t stream);

/* ---- general pixel-adaptive convolution (network/libs/base/pac.py) ------- *
 * The single-step, multi-channel, strided / dilated form of the op the K x K recurrence iterates.  All tensors
 * are contiguous and of one dtype (CSPN_F32 / CSPN_F16, fp32 accumulation):
 *   input  [B, C, H, W]
 *   kernel [B, kernel_ch, kh, kw, Ho, Wo], kernel_ch = 1 (shared by all channels) or C
 *   out    [B, C, Ho, Wo],  Ho = (H + 2 ph - dh (kh-1) - 1) / sh + 1 (same for Wo)             pac.py:61-62
 * out[b,c,y,x] = sum_ij kernel[b,c|0,i,j,y,x] * in0[b,c, y sh - ph + i dh, x sw - pw + j dw], in0 = zero-extended
 * input (pac.py:89-92).  Plain layouts here: this op does NOT use the engine's interleaved fp16 tap volume. */
typedef struct cspn_conv_geometry {
    int kh, kw;               /* window                    (kernel_size, pac.py:125) */
    int sh, sw;               /* stride                    (pac.py:126) */
    int ph, pw;               /* zero padding              (pac.py:127) */
    int dh, dw;               /* dilation                  (pac.py:128) */
    int oph, opw;             /* output_padding            (nd2col transposed only, pac.py:56) */
    int transposed;           /* nd2col only: fractional stride (pac.py:51-58); must be 0 for conv2d */
} cspn_conv_geometry;

/* Output extent for an input extent under `geom` (pac.py:41-42); writes Ho, Wo; 0 if the result is empty/invalid. */
int cspn_pac_out_size(int H, int W, const cspn_conv_geometry* geom, int* Ho, int* Wo);

/* ---- weight-resident single-launch 3x3 forward (inference; CSPN_new.py:26-92) ------------------------------------------
 * One launch per chunk of whole images: every workgroup owns one tile for all T steps, derives its normalised weights
 * from the raw guidance once and keeps them in registers (no tap volume is written or re-read), runs the steps in phases
 * of `steps_per_phase` on the depth tile in LDS and swaps tile borders with its 8 neighbouring tiles between phases
 * through two scratch planes (device-scope stores/loads + one phase flag per tile).  Results are bit-identical to
 * cspn3_propagate_from_guidance.  fp32, W % 4 == 0 (W_valid for narrower images), 16-byte aligned tensors.
 *
 * Co-residency: the workgroups that refine one image wait for each other, so at most images_per_launch whole images — never more
 * workgroups than the device has CUs — are in flight at a time (cspn3_resident_plan chunks the batch into `launches` launches of
 * images_per_launch images).  The device must not be shared with ANOTHER resident launch at the
 * same time (a second process on the GPU, or a second stream of this process): callers serialise resident launches per
 * device (cspn_monodepth_amd/functional.py chains them with events).  The neighbour wait is bounded (seconds): on
 * time-out the launch stores 1 to status word 1 of the workspace (and to host_err[0], if given),
 * drains, and leaves `out` incomplete — the tiles that gave up are filled with NaN (in `out`, or in the last history
 * plane) — and the caller must look at the word before trusting the result and re-run the call on the streaming entries
 * (cspn3_propagate_from_guidance: the same bits) when it is set — or set plan->guard (below, ABI 10) and let the library
 * re-compute a call that gave up on the stream, before anything can read it.  cspn_monodepth_amd/functional.py does both for its
 * callers: every unscored call (inference and training forms) carries the guard; calls without it (the scored forward by
 * default) are journaled and repaired in place at the next launch or host-side consumer (ensure_resident_ok), unguarded
 * training-form calls raise ResidentLaunchTimeout before `.backward()` returns; after three time-outs the resident
 * schedule switches itself off for the process.
 *
 * host_err_or_null: TWO host-mapped 32-bit words (pinned host memory the device can write): [0] receives 1 on a time-out;
 * [1] receives `seq` when the last launch of a TRAINING-form call (history != NULL) or of cspn3_transposed_resident has
 * finished — the last workgroup to count itself out (status word 2 of the workspace) stores it — so a host can learn
 * "this call is complete and host_err[0] is final" by polling host memory, without a HIP call or an event on the stream (functional._ResidentCheckpoint: the end-of-backward check).
 *
 * work: cspn3_resident_workspace_bytes(B,H,W) bytes, ZERO-initialised once by the caller, then only ever passed to this
 * entry.  seq: any value in [1, 2^31-256] that grows by at least 256 from one call on the same workspace to the next
 * (a tile that finished phase p publishes seq + p + 1; at most 255 phases per call, i.e. T <= 1020 with 4-step phases).
 * Layout: the two exchange planes [2][B,H,W] f32 (never need initialising), rounded up to 16 bytes, then the control words
 * (status + tile flags).  Re-zeroing the control words makes any seq valid again: a captured HIP graph records that memset
 * in front of the launch and replays with a constant seq. */
typedef struct cspn_resident_plan {
    int steps_per_phase;   /* in: 0 = choose (12, 8, 6 or 4); out: the value used.  Even whenever T needs more  */
                           /* than one phase (an odd request then has no plan); any value <= T otherwise   */
    int tiles_x, tiles_y;  /* tiles per image                                                             */
    int tile_w, tile_h;
    int quads_per_thread, threads;   /* threads, cspn3_*: in 0 / 512 = 512-thread workgroups (all forms), 1024 = one quad per thread on 1024
                                      * threads, four wavefronts per SIMD (inference forms only; small shards, round 5); out: the value used */
    int images_per_launch, launches;   /* images in flight at a time; launches of that many images (the K = 5 reverse sweep: rounds of ONE launch) */
    int lds_bytes, n_cu;
    float region_over_tile; /* (tile + halo) area / tile area: the redundant-compute factor of the phases  */
    unsigned spin_limit;    /* in: polls before a neighbour wait gives up; 0 = default (~seconds)          */
    unsigned long long* debug_stamps; /* in: developer probe, device buffer [images_per_launch x tiles][16] of 100 MHz wall-clock stamps (round 0)
                                       * (start, weights derived, then per phase: staged, steps done, exchanged) or NULL */
    int step_form;          /* cspnk_forward_resident, in: CSPN_STEP_AUTO (0), CSPN_STEP_FMA or CSPN_STEP_DOT2 — see there */
    int guard;              /* in (ABI 10): != 0 enqueues a guard kernel behind the call's launch(es).  It reads the call's abort word and
                             * returns at once unless a tile gave up (co-residency time-out); then it re-computes the whole result of the
                             * call on the stream with the resident kernel's arithmetic.  Whatever consumes the result later on that
                             * stream — any GPU kernel, a copy to the host — sees the finished tensor, as with the reference's ATen
                             * module (CSPN_new.py:80-92); the error words are still set, for the host's statistics.
                             * cspn3_forward_resident: every form — inference, scored inference (the guard also adds the metric terms of
                             * the pixels the failed launch left unscored), training forward; cspn3_transposed_resident(_guidance);
                             * the K = 3 fp32 softmax model through cspnk_forward_resident(_history): bit-identical results.
                             * cspnk_forward_resident, unscored: the bits of the FMA step form; the dot-product form is re-computed
                             * with its own arithmetic (tap pairs, v_dot2_f32_f16 in its order, one rounding per step): its bits too.
                             * cspnk_forward_resident_history, K = 5 fp16 (the dot-product kernel): history and tap volume re-computed as
                             * cspn_pac_prepare + cspn_propagate (history) at one step per launch store them, bit for bit (the
                             * dot-product kernel's own bits are within fp16 rounding of those); cspnk_transposed_resident: bit-identical.
                             * T * (K / 2) <= 54; refused where no form exists (K x K scored inference).
                             * Costs the success path one small launch (~1-2 us on the stream). */
} cspn_resident_plan;
#define CSPN_STEP_AUTO 0   /* the dot-product form where it exists (K = 5, fp16 guidance, fp16 planes), else the FMA form  */
#define CSPN_STEP_FMA 1    /* one v_fma_mix_f32 per tap, fp32 state inside a phase: the bits of the multi-launch schedule   */
#define CSPN_STEP_DOT2 2   /* v_dot2_f32_f16 on fp16 state pairs, state rounded to half after EVERY step, one launch        */
/* n_cu <= 0: ask the current device.  Returns 0 (with a message) when no tiling fits. */
int cspn3_resident_plan(int B, int H, int W, int T, int blend, int n_cu, cspn_resident_plan* in_out);
size_t cspn3_resident_workspace_bytes(int B, int H, int W);
/* Training form: history != NULL ([T,B,H,W] f32, receives d_1..d_T; `out` may be NULL) together with w8_out ([B,8,H,W])
 * and s_out ([B,H,W]): the launch also publishes the normalised weights and the normaliser S once — exactly what
 * cspn3_propagate_from_guidance hands the backward (cspn_propagate_transposed, cspn3_backward_tail).  No scoring then.
 * w8_out may be NULL (ABI 9): only S is published — enough for cspn3_transposed_resident_guidance + cspn3_backward_tail, which
 * rebuild the taps from guidance and S; the 8-plane volume (53 MB at config 2) is then neither written nor read. */
int cspn3_forward_resident(const void* guidance, long g_batch_stride, long g_chan_stride, const void* d0,
                           const void* sparse_or_null, void* out, void* history_or_null, void* w8_out_or_null,
                           float* s_out_or_null, void* work, unsigned seq, unsigned* host_err_or_null,
                           int B, int H, int W, int W_valid, int T, int blend, const void* target_or_null,
                           double* acc_or_null, int nslots, const cspn_resident_plan* plan_or_null, cspn_stream_t stream);

/* The backward's reverse sweep G_t = stencil^T((1-m) G_{t+1}), t = T-1..0, as ONE weight-resident launch per chunk of
 * images: the transposed taps are gathered once from the forward tap volume w8 [B,8,H,W] (f32) and stay in registers.
 * history [T,B,H,W] receives G_{T-1} .. G_0 in that order (as cspn_propagate_transposed); premask != 0 applies (1-m),
 * m = sign(sparse_f32).  Workspace / seq / host_err / plan exactly as cspn3_forward_resident (the two may share one
 * workspace).  Bit-identical to cspn_propagate_transposed. */
int cspn3_transposed_resident(const void* w8, const float* g_T, const float* sparse_f32_or_null, float* history, void* work,
                              unsigned seq, unsigned* host_err_or_null, int B, int H, int W, int W_valid, int T, int premask,
                              const cspn_resident_plan* plan_or_null, cspn_stream_t stream);
/* The same reverse sweep without a tap volume (ABI 9): the transposed tap j at p is w_{7-j}[p + off_j] = |g_j[p]| / S[p + off_j],
 * so it is rebuilt from the raw guidance (f32, channels 0..7 through the strides, as cspn3_forward_resident reads it) at the
 * quad ITSELF and the forward's refined reciprocal of the published normaliser S [B,H,W] at the 3 x 3 neighbours.  For
 * normalisers in [2^-100, 2^100] the taps — hence history — are bit-identical to cspn3_transposed_resident on the volume the
 * forward would have published.  Everything else as cspn3_transposed_resident. */
int cspn3_transposed_resident_guidance(const void* guidance, long g_batch_stride, long g_chan_stride, const float* s,
                                       const float* g_T, const float* sparse_f32_or_null, float* history, void* work,
                                       unsigned seq, unsigned* host_err_or_null, int B, int H, int W, int W_valid, int T,
                                       int premask, const cspn_resident_plan* plan_or_null, cspn_stream_t stream);

/* The K x K softmax / pixel-adaptive variant (reference: network/libs/post_process/CSPN_ours.py:24-54 — softmax :35, zero
 * centre tap :37-39, prop_time x { pac.conv2d :49 -> base/pac.py:89-92, sparse blend :51-53 }) as weight-resident launches:
 * `guided` [B, K*K-1, H, W] (g_dtype fp16 or fp32) is read ONCE, the softmax weights stay in registers for all T steps —
 * fp16 guidance: taps packed two pixels per register; fp32 guidance (fp32 depth planes only; what the reference's unet_ours
 * feeds: 8-channel fp32, K = 3 — served by cspn3_resident's kernel in its softmax-weight form, `threads` ignored): fp32
 * taps, the 1e-5 parity of the multi-launch fp32 path — no tap volume is written
 * (cspn_pac_prepare + cspn_propagate move 4.8x the compulsory bytes at BASELINE config 3).  K = 3 or 5; W % 8 == 0 (whole octs).  x0 / sparse / out / target are [B,H,W] planes of `state_dtype` (CSPN_F16 or CSPN_F32); with fp16
 * planes the state is rounded to half at every phase boundary — exactly where cspn_propagate with steps_per_launch =
 * steps_per_phase rounds it between launches, so the two schedules agree bit for bit (weights: same softmax arithmetic as
 * cspn_pac_prepare).  Workspace, seq, host_err, plan, co-residency, time-out and completion words: exactly as
 * cspn3_forward_resident (cspnk_resident_workspace_bytes sizes the exchange planes for the state dtype; the plan's
 * quads_per_thread field holds the OCTS per thread; `threads` in: 0 = choose, 512 or 768 = pin the workgroup size).  A batch whose taps do not fit the register files of the chip is
 * chunked into several launches of whole images (config 3: two launches of 12).
 * plan->step_form: K = 5 with fp16 guidance and fp16 planes — BASELINE config 3 — has a second kernel (csrc/cspnk_d2.hip,
 * CSPN_STEP_DOT2, what CSPN_STEP_AUTO picks there): the state is kept as packed fp16 pairs and rounded to half after every
 * step — as the reference's half tensors are between steps — two taps per v_dot2_f32_f16 with fp32 accumulation, and the
 * chunks of a batch are refined back to back by ONE launch.  Its results equal the oracle's within the fp16 tolerance of the
 * configuration but are not the bits of the phase-rounded schedules; CSPN_STEP_FMA keeps those (and is all the other
 * configurations have). */
int cspnk_resident_plan(int K, int g_dtype, int B, int H, int W, int T, int blend, int n_cu, cspn_resident_plan* in_out);
size_t cspnk_resident_workspace_bytes(int B, int H, int W, int state_dtype);
int cspnk_forward_resident(const void* guided, int g_dtype, int K, const void* x0, const void* sparse_or_null, void* out,
                           int state_dtype, void* work, unsigned seq, unsigned* host_err_or_null, int B, int H, int W,
                           int T, int blend, const void* target_or_null, double* acc_or_null, int nslots,
                           const cspn_resident_plan* plan_or_null, cspn_stream_t stream);
/* Training form of the K x K resident forward.  K = 3, fp32 guidance (the configuration the reference trains, unet_ours.py:305,
 * :333): history [T,B,H,W] f32 receives x_1 .. x_T and wk_out [B,8,H,W] f32 the softmax taps once — what cspn_pac_prepare +
 * cspn_propagate(history) hand the backward, bit for bit.  The reverse sweep is cspn3_transposed_resident on wk_out, the
 * tail cspn_pac_backward_tail.  K = 5, fp16 guidance (BASELINE config 3's shape): x0 / sparse / history are fp16 planes, the
 * dot-product kernel (CSPN_STEP_DOT2: state rounded to half after every step) writes every step's state to its plane and
 * wk_out receives the fp16 tap volume (pair-interleaved layout, as cspn_pac_prepare writes it) — one launch for the whole
 * batch; the backward is cspnk_transposed_resident (or cspn_transpose_weights + cspn_propagate (history)) +
 * cspn_pac_backward_tail.  Workspace / seq / host_err (completion word included) / plan as cspn3_forward_resident. */
int cspnk_forward_resident_history(const void* guided, int g_dtype, int K, const void* x0, const void* sparse_or_null,
                                   void* history, void* wk_out, void* work, unsigned seq, unsigned* host_err_or_null,
                                   int B, int H, int W, int T, int blend, const cspn_resident_plan* plan_or_null,
                                   cspn_stream_t stream);

/* The K = 5 reverse sweep of the backward, G_t = stencil^T((1-m) G_{t+1}), t = T-1..0 (pac.py:96-121 applied T times), as
 * weight-resident launches: the transposed taps w_{23-j}[p + off_j] are gathered once from the forward's fp16 tap volume `wk`
 * (the layout cspn_pac_prepare / cspnk_forward_resident_history write) and stay packed in registers; fp32 state.  history
 * [T,B,H,W] f32 receives G_{T-1} .. G_0 in that order; premask != 0 applies (1-m), m = sign(sparse).  Bit-identical to
 * cspn_transpose_weights + cspn_propagate (history, CSPN_BLEND_PREMASK) on the same inputs.  in_dtype = dtype of g_T and sparse:
 * CSPN_F32, or CSPN_F16 — the training step on half planes hands its cotangent and sparse plane over as they are (no cast
 * launches); the state is fp32 either way, and g_T_f32_out [B,H,W], when given, receives G_T as fp32 for cspn_pac_backward_tail.
 * Workspace (cspnk_resident_workspace_bytes with CSPN_F32), seq, host_err (completion word included), plan: as
 * cspn3_transposed_resident.  ONE launch for up to 8 x images_per_launch images: rounds x images_per_launch x tiles workgroups, the
 * first round's resident at once, the later rounds' dispatched as those finish (debug_stamps: round 0's workgroups only). */
int cspnk_transposed_resident(const void* wk, int w_dtype, int K, const void* g_T, const void* sparse_or_null, int in_dtype,
                              float* g_T_f32_out_or_null, float* history, void* work, unsigned seq, unsigned* host_err_or_null,
                              int B, int H, int W, int T, int premask, const cspn_resident_plan* plan_or_null, cspn_stream_t stream);

/* A/B + test switch (process-wide): on != 0 makes every cspn_pac_* entry skip its LDS-tiled kernels and run the generic
 * one-quad-per-thread kernels; the previous setting is stored to *previous_or_null.  The initial value is read once from
 * CSPN_PAC_SCALAR=1 in the environment (the hot path never calls getenv).  Results are identical either way. */
int cspn_pac_force_generic(int on, int* previous_or_null);

/* Conv2dFn.forward (pac.py:75-94) == the native_impl branch (pac.py:130-140). */
int cspn_pac_conv2d(const void* input, const void* kernel, void* out, int dtype, int B, int C, int kernel_ch,
                    int H, int W, const cspn_conv_geometry* geom, cspn_stream_t stream);
/* Conv2dFn.backward (pac.py:98-121): grad_input [B,C,H,W] = fold(grad_out (x) kernel) (:104-113) — computed as a
 * gather per input pixel, no atomics, deterministic; grad_kernel [B,kernel_ch,kh,kw,Ho,Wo] =
 * grad_out * unfold(input), summed over channels when kernel_ch == 1 (:115-119). */
int cspn_pac_conv2d_grad_input(const void* grad_out, const void* kernel, void* grad_input, int dtype, int B, int C,
                               int kernel_ch, int H, int W, const cspn_conv_geometry* geom, cspn_stream_t stream);
int cspn_pac_conv2d_grad_kernel(const void* grad_out, const void* input, void* grad_kernel, int dtype, int B, int C,
                                int kernel_ch, int H, int W, const cspn_conv_geometry* geom, cspn_stream_t stream);
/* nd2col (pac.py:35-70), 2-D: cols [B, C, kh, kw, Ho, Wo]; geom->transposed selects the zero-insertion form. */
int cspn_pac_nd2col(const void* input, void* cols, int dtype, int B, int C, int H, int W,
                    const cspn_conv_geometry* geom, cspn_stream_t stream);

/* ---- zero-insertion un-pooling -------------------------------------- *
 * out[p, s*h, s*w] = in[p, h, w], 0 elsewhere, cropped to oH x oW (1 <= oH <= s*H, 1 <= oW <= s*W); p runs over the
 * B*C planes (fewer than 2^31 output quads per call).  network/unet_ours.py:138-150 (grouped conv_transpose2d with a one-hot weight + crop) and
 * network/unet_cspn_nyu.py:202-213 (nearest upsample x checkerboard mask).  The kernel writes the zeros too: `out`
 * needs no memset.  The backward is the strided gather grad_in[p,h,w] = grad_out[p, s*h, s*w] (0 past the crop). */
int cspn_unpool2d(const void* input, void* out, int dtype, long planes, int H, int W, int scale, int oH, int oW,
                  cspn_stream_t stream);
int cspn_unpool2d_backward(const void* grad_out, void* grad_input, int dtype, long planes, int H, int W, int scale,
                           int oH, int oW, cspn_stream_t stream);

/* ---- debugging aid: poisoned LDS (round 5) --------------------------------------------------- *
 * A kernel does not get a cleared LDS: it inherits what the previous workgroup on the CU left there, so a kernel that reads an
 * LDS word it never wrote returns results that depend on what ran before it.  (That was the root cause of the one bit mismatch
 * ever seen on the default path: cspn3_resident's blended CLEAN instances added a never-written private slot to the rows past
 * the region, and for a bottom-edge tile that sum is the zero padding below the image — DESIGN.md §4.1b.)
 * enabled != 0: from now on EVERY kernel this library launches is preceded, on the same stream, by a fill of all 160 KB of LDS
 * of every CU with `pattern` (0x7fc00000 = NaN makes any such read visible in the result).  Process-wide, off by default,
 * meant for test runs (tests/: CSPN_DEBUG_LDS_POISON=nan runs the whole GPU suite that way); costs one predictable host branch
 * per launch when off.  Results of a correct kernel are unchanged.  *previous_or_null receives the previous setting. */
int cspn_debug_set_lds_poison(int enabled, unsigned pattern, int* previous_or_null);

#ifdef __cplusplus
}
#endif
#endif /* CSPN_HIP_H_ */

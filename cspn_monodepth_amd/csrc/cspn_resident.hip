// cspn_resident.hip — the 3x3 inference forward as ONE launch with the weights RESIDENT in registers for all T steps.
//
// Reference path: network/libs/post_process/CSPN_new.py:26-92 (abs / shift / normalise once, then 24 x {8-way propagate,
// blend}).  The multi-launch schedule of cspn_propagate.hip re-streams the 8 weight planes once per launch of S steps
// (config 2: 53 MB written once, read twice = 160 of the 271 MB a forward moves).  Here a workgroup owns ONE tile for the
// whole forward:
//   * it derives the normalised weights of its tile + (S-1)-pixel halo from the raw guidance ONCE (same arithmetic as
//     cspn3_prepare_kernel / the WSRC=1 launch, div8_shared_reciprocal) and keeps them in VGPRs: no weight volume exists;
//   * the T steps run in phases of S steps on the depth tile in LDS (the step loop of cspn_prop_fused);
//   * between phases the workgroups exchange their tile borders through a global scratch plane with DEVICE-SCOPE (sc1)
//     stores / loads — coherent across the 8 XCDs' private L2s without any cache-wide write-back or invalidate — and a
//     per-tile phase flag the 8 neighbouring tiles poll (point-to-point, no grid barrier).
// HBM traffic of a forward drops to ~the compulsory 8 guidance planes + depth in / out (+ the border exchange).
//
// All workgroups of a launch must be co-resident (they wait for each other): a launch never has more workgroups than the
// device has CUs (every instance fits at least once per CU), images are chunked over several launches when a batch needs
// more tiles than CUs.  The neighbour wait is bounded: on time-out (another tenant holding CUs for seconds) the launch
// sets the abort / error words of its workspace, the remaining workgroups drain, and the host raises on the next call
// instead of hanging (include/cspn_hip.h: cspn3_forward_resident).
//
// Measured on MI355X (config 2: B=24, 228x304, T=24; profiles/r02_*): 2x5 tiles of 152x46 per image = 240 workgroups of
// 512 threads x 5 quads (160 weight registers per thread, 256 VGPRs, 2 wavefronts per SIMD), 8-step phases.  Timeline per
// workgroup (tools/resident_stamps.py): weights derived 11.0 us (the guidance stream, HBM-bound) + 3.0 us until the
// slowest wavefront has parked, 3 x 6.0 us of steps (VALU-bound: 228 instructions per wavefront-step, 160 of them FMAs),
// 2 x (2.6 us exchange + 1.35 us halo staging), fused metrics 3.6 us + launch ~2 us: 46 us per scored forward against 73 us for
// the three S=8 launches, with 2.3x less HBM traffic.
//
// What keeps the 256-VGPR instances spill-free (each item was a measured regression before it was written this way):
//   * every scratch reload inside the derive is an s_waitcnt vmcnt(0) in the middle of the in-order guidance stream, so
//     the derive carries nothing it does not need: loads / stores go through an SGPR base + 32-bit byte offset in address
//     space 1 (at32: no address pairs, no flat loads), staging addresses advance without divisions, and everything the
//     later stages address with is recomputed from late, opaque copies of the strip coordinates;
//   * the step loop is unrolled by two with the LDS buffers RES_PP apart (the other buffer is an immediate offset, the
//     carried quads alternate between two register sets instead of being copied), sits inside a wave-uniform branch,
//     and computes unconditionally with masked stores;
//   * v_pk_fma_f32 does not help: it issues at half the rate of v_fma_f32 here (tools/probes/pk_fma_probe.hip).
#include "cspn_common.hpp"

#include <atomic>

namespace {

struct ResArgs {
    const float* g;          // guidance [B,C>=8,H,W], channels 0..7 read in place
    long g_bs, g_cs;
    const float* d0;         // [B,H,W]
    const float* sparse;     // [B,H,W] or null
    float* out;              // [B,H,W]
    float* xbuf;             // exchange planes [2][B,H,W] (workspace)
    unsigned* flags;         // [B * tiles_per_img] phase flags (workspace, zero-initialised once)
    unsigned* status;        // [0] abort word of the running launch chain, [1] sticky error word (workspace)
    unsigned* host_err;      // optional TWO host-mapped words: [0] receives the error as well, [1] receives `seq` when the
                             // last launch of the call has finished (all workgroups passed their end)
    int last_chunk;          // this launch is the last one of the call
    unsigned seq;            // flag base of this call: a tile that finished phase p publishes seq + p + 1
    float* hist;             // MODE 2: [T][B,H,W] receives the state after every step (d_1 .. d_T), `out` is unused
    float* w_out;            // MODE 2: [B,8,H,W] receives the normalised weights (the backward streams them)
    float* s_out;            // MODE 2: [B,H,W] receives the normaliser S (the backward's quotient rule needs it)
    const float* s_in;       // MODE 4: [B,H,W] the normaliser S the training forward published (a.g is the raw guidance)
    const float* target;     // MODE 1: [B,H,W]
    double* macc;
    int nslots;
    int B, H, W, Wv, T, S;
    int tw, th, tiles_x, tiles_y;
    int wq, wr, hxw, hyw, dr, ls;
    int b0, nb;              // images [b0, b0 + nb) are refined by this launch
    unsigned spin_limit;
    unsigned long long* dbg;  // developer probe: [grid][16] wall-clock stamps (100 MHz) per workgroup, or null
};

__device__ __forceinline__ void st4_dev(float* base, unsigned elem, float a, float b, float c, float d) {
    // ONE 16-byte device-scope store (global_store_dwordx4 ... sc1): visible to every XCD once vmcnt drops.  Written as
    // asm because the compiler only offers <= 8-byte atomics, and two 8-byte stores per quad touch every 64-byte line
    // twice with half masks (PMC: 2x the written bytes).  The caller waits with s_waitcnt vmcnt(0) before it signals.
    // `base` is wave-uniform (the image's exchange plane): SGPR base + 32-bit VGPR byte offset, no 64-bit address math.
    const v4f v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(elem * 4u), "v"(v), "s"(base) : "memory");
}
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {      // a wave-uniform pointer, pinned to an SGPR pair
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// element `e` of a wave-uniform plane as SGPR base + 32-bit VGPR BYTE offset (planes stay below 2^30 elements: the host
// checks), so that no load / store of the kernel needs 64-bit address arithmetic or an address register pair
// The result is typed as a GLOBAL (address space 1) pointer: a pointer rebuilt from integers is otherwise generic, and a
// flat_load counts on vmcnt AND lgkmcnt and may return out of order — every wait after one becomes vmcnt(0).
#define GLB __attribute__((address_space(1)))
typedef GLB char* gptr;
__device__ __forceinline__ gptr at32(const float* base, unsigned e) {
    return (gptr) reinterpret_cast<unsigned long long>(base) + (e << 2);
}
__device__ __forceinline__ float4 ld4(gptr p) {
    const v4f v = *reinterpret_cast<const GLB v4f*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld1(gptr p) { return *reinterpret_cast<const GLB float*>(p); }
__device__ __forceinline__ void st4(gptr p, float4 v) {
    const v4f w = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<GLB v4f*>(p) = w;
}
// The history planes of the training forms (d_1..d_T of MODE 2, G_{T-1}..G_0 of MODE 3 / 4: 160 MB per launch at config 2) are
// written once and read by a LATER kernel: stored non-temporally they do not push the guidance / exchange lines out of the L2
// while the launch runs, and the tail that streams them afterwards runs faster as well — same-box A/B, two alternating
// repetitions: tail 95.2-97.1 -> 86.8-87.3 us, sparse training forward 69.7-70.3 -> 64.9-65.4, sparse reverse sweep 75.4-76.9 ->
// 72.4-73.7, the plain forward / sweep -0 / -1.4 us (-DCSPN_RES_HIST_NT=0 for A/B).
#ifndef CSPN_RES_HIST_NT
#define CSPN_RES_HIST_NT 1
#endif
// Round 6, measured and NOT the default (NEGATIVE_RESULTS #53-#55; profiles/r06_step_pipe_ab.txt): the step's LDS round trips.  A step is
//   barrier -> ds_read of the two halo rows -> wait -> DPP of all rows -> exec-masked ds_read of the strip-end lanes' columns -> wait -> FMAs
//   -> ds_write -> wait -> barrier:  two LDS read latencies in series behind every barrier.
// -DCSPN_RES_STEP_PIPE=1 takes the own rows' neighbour columns (DPP: registers of the neighbouring lanes, no LDS) BEFORE the barrier, from the rows
// just computed, in the shadow of the ds_write latency, and issues EVERY LDS read of the step right behind the barrier (the halo rows' strip-end
// columns into temporaries, merged with one select after the halo rows' DPP): one read latency instead of two, bit-identical — and SLOWER on every
// shape (config 2 +1.6 us, shards +1.0 / +1.2 us).  The two perf-only experiments below say why nothing of this kind can pay: without ANY
// strip-end read the forward is no faster (+-0.3 us), and without ANY step barrier it gains 2.3 us at config 2, 0.4 us at NYU B = 3, 1.4 us at
// KITTI B = 1 — the step is bound by VALU issue (one instruction per wavefront every ~5 cycles, 2.5 per SIMD with its two wavefronts:
// tools/probes/vgpr_bank_probe.hip), not by its barrier or its LDS latency, so "two steps per barrier" (+20..100 % FMAs for half of that) loses.
#ifndef CSPN_RES_STEP_PIPE
#define CSPN_RES_STEP_PIPE 0
#endif
#ifndef CSPN_RES_PUBLISH_ALL
#define CSPN_RES_PUBLISH_ALL 1      // 0: only the quads a neighbour will read are published — 6 MB less traffic per forward at config 2 but
                                    // NOT faster (same-box A/B, profiles/r05_publish_ab.txt: 48.0 vs 47.8 us, sparse 53.9 vs 53.1): kept for A/B runs
#endif
__device__ __forceinline__ void st4_hist(gptr p, float4 v) {
    const v4f w = {v.x, v.y, v.z, v.w};
    if (CSPN_RES_HIST_NT) __builtin_nontemporal_store(w, reinterpret_cast<GLB v4f*>(p));
    else *reinterpret_cast<GLB v4f*>(p) = w;
}
__device__ __forceinline__ float4 ld4_dev(gptr p) {
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const GLB unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const GLB unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                       __uint_as_float((unsigned)(hi >> 32)));
}
__device__ __forceinline__ float ld1_dev(gptr p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const GLB unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

constexpr int RES_THREADS = 512;
// The two depth buffers of the step loop sit a COMPILE-TIME distance apart (RES_PP floats; the second one starts there
// whatever the region size): with the step loop unrolled by two, "the other buffer" is then an immediate offset of the
// ds_read / ds_write instead of a second set of row addresses (7 VGPRs the 256-register instances do not have), and no
// pointer is swapped per step.  65408 bytes + the largest row-relative offset still fit the 16-bit offset field.
constexpr int RES_PP = 16352;

// CLEAN = 1: the host verified that every region of the launch lies inside the image (shifted regions, image at least one
// region large, no row padding): the step body carries no zero-padding selects.  A launch-level property, so that each
// instance holds ONE copy of the step body (+ the peeled final step) — two copies in one kernel cost the 256-VGPR
// instances their spill-free hot loop.
// MODE 0: inference; 1: inference + fused depth metrics; 2: training forward — every step's state goes to its history
// plane and the weights + S are published once (what cspn3_propagate_from_guidance hands the backward).
// PAC = 1: the weights are the softmax over the 8 guidance channels at the CENTRE pixel (tap j = channel j: the K = 3 case of
// CSPN_ours.py:35-41) instead of the neighbour-indexed, sum-normalised gates of CSPN_new — everything after the derive (steps,
// exchange, blend, metrics, history planes) is the same recurrence.  MODE 2 publishes the softmax taps as the [B,8,H,W] volume the
// backward streams (no S: the softmax backward does not need one); the reverse sweep is the plain MODE 3 on that volume.
template <int NQ, int NTHREADS, int BLEND, int MODE, int CLEAN, int PAC = 0>
__global__ __launch_bounds__(NTHREADS, NTHREADS / 256) void cspn3_resident(const ResArgs a) {
    static_assert(!PAC || MODE <= 2, "the softmax-weight form has no transposed sweep of its own (the published volume feeds MODE 3)");
    constexpr int R = 1, NT = 8, WIN = 6;
    // MODE 3: the backward's reverse sweep  G_t = stencil^T((1-m) G_{t+1})  as the same recurrence on the TRANSPOSED taps:
    // tap j = w_{7-j}[p + off_j], gathered from the forward tap volume [B,8,H,W] (a.g) exactly like channel 7-j of the
    // guidance is gathered for the forward — minus |.| and the normalisation.  BLEND then means PREMASK: the state that
    // travels (LDS, exchange planes) is (1-m) G, the history planes receive G itself.  d0 is G_T = dL/dout.
    // MODE 4: the same reverse sweep with the transposed taps REBUILT from the raw guidance and the published normaliser instead
    // of gathered from a tap volume:  w_{7-j}[p + off_j] = |g_j[p]| / S[p + off_j]  — the guidance channel j at the quad itself
    // (no shifted gather at all), times the forward's refined reciprocal of S at the 3 x 3 neighbours (NQ + 2 aligned row quads
    // of ONE plane per thread, neighbours by DPP, strip ends by scalars).  The training forward then publishes S only: the
    // 53 MB volume of config 2 is neither written nor read (the tail rebuilds its w_j the same way, cspn_backward.hip).
    constexpr bool SCORE = MODE == 1, TRANSG = MODE == 4, TRANS = MODE == 3 || TRANSG, HIST = MODE == 2 || TRANS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int wg_bad;

    const int tid = threadIdx.x;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    // A launch may hold several ROUNDS of nb images (-DCSPN_RES_ONE_LAUNCH=1; by default a launch is one round): rounds x (nb x tiles)
    // workgroups, round 0's — images b0 .. b0 + nb - 1, at most one workgroup per CU — resident at once, those of round r + 1 dispatched
    // in blockIdx order as round r's finish.  Flags and exchange planes are per image, so the rounds share nothing.  (cspnk_resident's
    // reverse sweep runs that way: cspnk_resident.hip.)
    const int wg_round = a.nb * tiles_per_img;
    const int round = (int)blockIdx.x / wg_round;
    const int tile = xcd_contiguous_id((int)blockIdx.x - round * wg_round, wg_round);
    const int bl = tile / tiles_per_img;                 // image within this round
    const int trem = tile - bl * tiles_per_img;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int b = a.b0 + round * a.nb + bl;
    const int H = a.H, W = a.W;
    const int y0 = ty * a.th, x0 = tx * a.tw;
    const size_t HW = (size_t)H * W;
    const size_t plane = (size_t)a.B * HW;
    if (tid == 0) wg_bad = 0;
    int n_stamp = 0;
    auto stamp = [&]() { if (a.dbg && tid == 0 && round == 0 && n_stamp < 16) a.dbg[(size_t)blockIdx.x * 16 + n_stamp++] = wall_clock64(); };   // (round 0's workgroups: [nb x tiles][16])
    stamp();
    // Completion word: every workgroup counts itself out (status word 2); the last one re-arms the counter for the next
    // launch — launches on a workspace never overlap — and, in the last launch of a call, stores `seq` to the second host
    // word.  A host that polls that word (no HIP call, no event on the stream) knows the call has finished and that
    // host_err[0] is final: a workgroup that gave up stored the error, fenced at system scope, and only then counted out.
    // Only the training-form launches (MODE 2 / 3) report: their reader is the end-of-backward check, inference results are
    // checked where the host synchronises anyway.  (Same-box A/B at config 2: the returning atomic at the end of every
    // workgroup costs the inference launch 0.6 us; counting out at the start of the last phase instead — a tile that waits
    // for nobody any more cannot time out — costs 1.3 us, the atomic then sits in the register-bound loop nest.)
    auto count_out = [&]() {
        if (HIST && tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.status + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == gridDim.x) {
                __hip_atomic_store(a.status + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.host_err && a.last_chunk) __hip_atomic_store(a.host_err + 1, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };

    if (b >= a.B) { count_out(); return; }               // the last round of a ragged batch has fewer images

    const float* __restrict__ din0 = uniform_ptr(a.d0 + (size_t)b * HW);
    const float* __restrict__ spg = BLEND ? uniform_ptr(a.sparse + (size_t)b * HW) : nullptr;

    // ---- ownership: strip (sx, sy) = NQ vertically consecutive quads of the weight region (tile + halo) -------------
    const int wq = a.wq, wr = a.wr;
    const int sy = tid / wq;
    const int sx = tid - sy * wq;
    const int r0 = sy * NQ;
    // Region origin: the tile's halo is laid out around it, but SHIFTED back into the image at image edges (a region that
    // would stick out on one side extends further on the other instead).  With W >= 4 wq and H >= wr every owned quad then
    // lies inside the image: no zero-padding selects in the step loop, no wasted halo, equal work for edge and inner tiles.
    // The shift never moves a region start before the previous tile's start, so that the halo always comes from the 8
    // ADJACENT tiles (the ones the exchange waits for) — a last tile cut short by the image edge keeps some out-of-image
    // rows / columns instead, and its wavefronts run the step body with the zero-padding selects.
    // (the "previous tile's start" bound belongs to the exchange: a single-phase launch stages everything from the coarse
    // depth, and its halo may be deeper than a tile — there the bound would cut the region short of y0 - hyw)
    const bool exch = a.T > a.S;
    const int rx0 = max(exch ? max(0, x0 - a.tw) : 0, min(x0 - a.hxw, W - 4 * wq));
    const int ry0 = max(exch ? max(0, y0 - a.th) : 0, min(y0 - a.hyw, H - wr));
    const int xq = rx0 + 4 * sx;
    const int yq0 = ry0 + r0;
    const bool x_in = (xq >= 0) && (xq < a.Wv);
    const int nval = a.Wv - xq;
    const int lane = tid & 63;
    const bool fix_left = (sx == 0) || (lane == 0);
    const bool fix_right = (sx == wq - 1) || (lane == 63);

    // ---- 0. the coarse-depth region of phase 0 is requested FIRST: loads return in order, so it lands in LDS while the
    //         (8x larger) guidance stream below is still in flight, and the staging costs no round trip of its own
    const int dr = a.dr, ls = a.ls;
    float* const cur = lds;                    // buffer 0: where every phase starts (phases have an even number of steps)
    float* const nxt = lds + RES_PP;
    const int yd0 = ry0 - R;                   // image y of depth-region row 0
    const int xd0 = rx0 - 4;                   // image x of LDS column 0
    float4 st0[NQ + 1];                        // dr * wq <= (NQ + 1) * NTHREADS quads
    unsigned st0_in = 0;
    // quad u * NTHREADS + tid of the region, row-major: (row, qx) advances by a uniform (NTHREADS / wq, NTHREADS % wq) per
    // u — no per-quad division, whose temporaries the register-bound derive cannot afford
    const int step_r = NTHREADS / wq, step_q = NTHREADS - step_r * wq;
    {
    int row = sy, qx = sx;
#pragma unroll
    for (int u = 0; u <= NQ; ++u) {
        const int y = yd0 + row, x = xd0 + 4 + 4 * qx;
        const bool valid = row < dr;
        const bool in = valid && (y >= 0 && y < H && x >= 0 && x < a.Wv);
        if (in) st0_in |= 1u << u;
        st0[u] = ld4(at32(din0, in ? (unsigned)(y * W + x) : 0u));
        if (TRANS && BLEND) {                          // PREMASK: the sweep starts from (1-m) G_T
            const float4 sp = ld4(at32(spg, in ? (unsigned)(y * W + x) : 0u));
            st0[u].x *= 1.f - sgnf(sp.x); st0[u].y *= 1.f - sgnf(sp.y); st0[u].z *= 1.f - sgnf(sp.z); st0[u].w *= 1.f - sgnf(sp.w);
        }
        row += step_r; qx += step_q;
        if (qx >= wq) { qx -= wq; ++row; }
    }
    }
    float ring0 = 0.f;
    bool ring0_in = false;
    if (tid < dr * 2 * R) {
        const int row = tid / (2 * R), c = tid - row * (2 * R);
        const int lc = (c < R) ? (4 - R + c) : (4 + 4 * wq + (c - R));
        const int y = yd0 + row, x = xd0 + lc;
        ring0_in = (y >= 0 && y < H && x >= 0 && x < a.Wv);
        ring0 = ld1(at32(din0, ring0_in ? (unsigned)(y * W + x) : 0u));
        if (TRANS && BLEND) ring0 *= 1.f - sgnf(ld1(at32(spg, ring0_in ? (unsigned)(y * W + x) : 0u)));
    }

    // ---- 1. weights of the owned quads, derived once from the raw guidance (CSPN_new.py:29-70, :124-127) -------------
    float wreg[NQ][NT][4];
    unsigned in_img = 0, interior = 0;
    // private slots: m * d0 of the owned quads, NQ consecutive 16-byte slots per THREAD — quad i is base + an immediate offset, so
    // the step loop carries no address arithmetic for them (indexed by region row and column they cost the blended training
    // instances 6-7 scratch reloads per step: +25 us per launch); stride NQ * 16 bytes is bank-conflict free for NQ = 1, 3, 5
    float* const md_lds = lds + RES_PP + (size_t)a.dr * a.ls;
    const float* __restrict__ gq = uniform_ptr(a.g + (size_t)b * a.g_bs);
    // tap j = (dy,dx) row-major without the centre reads channel 7-j at p+off_j: the aligned quad of row y+dy gives three of
    // the four shifted values, the fourth is the neighbouring lane's quad (DPP) or, at strip ends / wave edges, a scalar.
    // All loads are branch-free (safe address + select: a conditional load becomes its own basic block with its own wait)
    // and are requested for every owned quad before the arithmetic starts.
    // (32-bit element offsets from the image's guidance base: the loads take the SGPR-base + VGPR-offset form and cost no
    // 64-bit address arithmetic; the host refuses guidance images of >= 2^30 elements.  Loads return in order, so quad i's
    // scalars are requested right behind quad i's planes and the arithmetic below can start on quad 0 while quads 1.. stream.)
    float edge[NQ][6];                     // [0..2]: column xq-1 of the dx<0 taps (j = 0,3,5); [3..5]: column xq+4 of the dx>0 taps (j = 2,4,7)
    const unsigned ucs = (unsigned)a.g_cs;
    // MODE 4: rows yq0-1 .. yq0+NQ of S, requested in front of the guidance (in-order return: the reciprocals are formed while
    // the guidance streams); [m][0..3] the aligned quad, [m][4] / [m][5] columns xq-1 / xq+4 for the strip-end lanes
    float srow[TRANSG ? NQ + 2 : 1][6];
    if constexpr (TRANSG) {
        const float* __restrict__ sq = uniform_ptr(a.s_in + (size_t)b * HW);
#pragma unroll
        for (int m = 0; m < NQ + 2; ++m) {
            const int ys = yq0 - 1 + m;
            const bool v = x_in && ys >= 0 && ys < H;
            const unsigned o = v ? (unsigned)(ys * W + xq) : 0u;
            const float4 q4 = ld4(at32(sq, o));
            srow[m][0] = q4.x; srow[m][1] = q4.y; srow[m][2] = q4.z; srow[m][3] = q4.w;
            srow[m][4] = ld1(at32(sq, (v && fix_left && xq >= 1) ? o - 1u : 0u));
            srow[m][5] = ld1(at32(sq, (v && fix_right && xq + 4 < a.Wv) ? o + 4u : 0u));
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int r = r0 + i, y = yq0 + i;
        const bool ok = (r < wr) && x_in && (y >= 0) && (y < H);
        if (ok) in_img |= 1u << i;
        if (ok && y >= y0 && y < y0 + a.th && xq >= x0 && xq < x0 + a.tw) interior |= 1u << i;
        unsigned orow[3];
        bool rokv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int row = y + d - 1;
            rokv[d] = ok && row >= 0 && row < H;
            orow[d] = rokv[d] ? (unsigned)(row * W + xq) : 0u;   // outside the image: a safe address of the plane, zeroed below
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int lin = j < 4 ? j : j + 1;
            const int d = (PAC || TRANSG) ? 1 : lin / 3;              // PAC / MODE 4: channel j at the quad itself
            const float4 v = ld4(at32(gq, (unsigned)((PAC || TRANSG) ? j : 7 - j) * ucs + orow[d]));
            wreg[i][j][0] = rokv[d] ? v.x : 0.f; wreg[i][j][1] = rokv[d] ? v.y : 0.f;
            wreg[i][j][2] = rokv[d] ? v.z : 0.f; wreg[i][j][3] = rokv[d] ? v.w : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) edge[i][t] = 0.f;
        if (!PAC && !TRANSG && (fix_left || fix_right)) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const bool lft = t < 3;
                const int j = lft ? (t == 0 ? 0 : (t == 1 ? 3 : 5)) : (t == 3 ? 2 : (t == 4 ? 4 : 7));
                const int lin = j < 4 ? j : j + 1;
                const int d = lin / 3;
                const int xx = lft ? xq - 1 : xq + 4;
                const bool c = rokv[d] && (lft ? fix_left : fix_right) && xx >= 0 && xx < W;
                // raw value only: consuming it here would make this block wait for the quad's loads (in-order return)
                edge[i][t] = ld1(at32(gq, (unsigned)(7 - j) * ucs + (c ? orow[d] + (unsigned)(xx - xq) : 0u)));
            }
        }
    }
    // park the depth region (its loads were requested before the guidance: only those are waited for here)
    // (the LDS slots are worked out here, from an opaque copy of the thread id: the derive above is register-bound, and a
    // spilled value reloaded in the middle of it waits for every outstanding guidance load)
    int tidk = tid, prow = sy, pqx = sx;
    asm volatile("" : "+v"(tidk), "+v"(prow), "+v"(pqx));
#pragma unroll
    for (int u = 0; u <= NQ; ++u) {
        const bool in = (st0_in >> u) & 1u;
        if (prow < dr)
            *reinterpret_cast<float4*>(&cur[prow * ls + 4 + 4 * pqx]) = make_float4(in ? st0[u].x : 0.f, in ? st0[u].y : 0.f, in ? st0[u].z : 0.f, in ? st0[u].w : 0.f);
        prow += step_r; pqx += step_q;
        if (pqx >= wq) { pqx -= wq; ++prow; }
    }
    if (tidk < dr * 2 * R) {
        const int row = tidk / (2 * R), c = tidk - row * (2 * R);
        const int ring0_at = row * ls + ((c < R) ? (4 - R + c) : (4 + 4 * wq + (c - R)));
        cur[ring0_at] = ring0_in ? ring0 : 0.f; nxt[ring0_at] = 0.f;
    }
    // The ring ROWS of the second buffer are never computed either.  With the regions shifted into the image they ARE the
    // zero padding above / below the image for edge tiles, so they must read as exactly 0 in both buffers (LDS keeps
    // whatever the previous kernel left there).
    for (int c = tidk; c < 2 * ls; c += NTHREADS) nxt[(c < ls ? 0 : (dr - 1) * ls - ls) + c] = 0.f;
    // sparse blend: the masks of the owned quads are requested now, behind the guidance (loads return in order: they are
    // there when the last quad's weights are), into the registers the parked depth quads just left
    float4 mraw[BLEND ? NQ : 1];
    if (BLEND) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const bool ok = (in_img >> i) & 1u;
            mraw[i] = ld4(at32(spg, ok ? (unsigned)((yq0 + i) * W + xq) : 0u));
        }
    }
    if constexpr (TRANSG) {
        // 1 / S by the forward's recipe (div8_shared_reciprocal), 0 where the neighbour lies outside the (valid) image: in place,
        // columns xq-1 .. xq+4 of every row end up as srow[m][4], [0..3], [5]
#pragma unroll
        for (int m = 0; m < NQ + 2; ++m) {
            const int ys = yq0 - 1 + m;
            const bool v = x_in && ys >= 0 && ys < H;
            auto rcp_fwd = [](float S) -> float {
                const bool okr = (S <= 0x1p+100f) && (S >= 0x1p-100f || S == 0.f);
                float r = __builtin_amdgcn_rcpf(S);
                r = fmaf(fmaf(-S, r, 1.0f), r, r);
                return okr ? r : 1.0f / S;
            };
#pragma unroll
            for (int e = 0; e < 4; ++e) srow[m][e] = (v && e < nval) ? rcp_fwd(srow[m][e]) : 0.f;
            const float nl = dpp_from_prev_lane(srow[m][3]), nr = dpp_from_next_lane(srow[m][0]);
            srow[m][4] = fix_left ? ((v && xq >= 1) ? rcp_fwd(srow[m][4]) : 0.f) : nl;
            srow[m][5] = fix_right ? ((v && xq + 4 < a.Wv) ? rcp_fwd(srow[m][5]) : 0.f) : nr;
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const bool ok = (in_img >> i) & 1u;
        if constexpr (TRANSG) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int lin = j < 4 ? j : j + 1;
                const int m = i + lin / 3, dx = lin % 3 - 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = e + dx;
                    const float rs = c < 0 ? srow[m][4] : (c > 3 ? srow[m][5] : srow[m][c]);
                    wreg[i][j][e] = (ok && e < nval) ? fabsf(wreg[i][j][e]) * rs : 0.f;
                }
            }
            continue;
        }
        // does edge[i][t] hold a pixel inside the image?  (recomputed: a bit mask built while the loads are issued is one
        // more live register there)
        auto edge_in = [&](int j, bool lft) -> bool {
            const int lin = j < 4 ? j : j + 1;
            const int row = yq0 + i + lin / 3 - 1;
            return ok && row >= 0 && row < H && (lft ? xq - 1 >= 0 : xq + 4 < W);
        };
        const unsigned off = ok ? (unsigned)((yq0 + i) * W + xq) : 0u;
        if constexpr (PAC) {
            // softmax over the 8 channels per pixel: the arithmetic of cspn_pac_prepare_kernel<3, float, float> (maximum,
            // two-piece exponential, sum in channel order, one refined reciprocal), so the taps are those of the prepared volume
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < NT; ++j) mx = fmaxf(mx, wreg[i][j][e]);
                float den = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) { wreg[i][j][e] = softmax_exp<float>(wreg[i][j][e] - mx); den += wreg[i][j][e]; }
                const float inv = reciprocal_refined(den);
#pragma unroll
                for (int j = 0; j < NT; ++j) wreg[i][j][e] = (ok && e < nval) ? softmax_weight<float>(wreg[i][j][e], inv) : 0.f;
            }
            if (MODE == 2 && ((interior >> i) & 1u)) store_taps_quad<NT>(a.w_out + (size_t)b * NT * HW, off, HW, wreg[i]);
            continue;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int lin = j < 4 ? j : j + 1;
            const int dx = lin % 3 - 1;
            const float q0 = wreg[i][j][0], q1 = wreg[i][j][1], q2 = wreg[i][j][2], q3 = wreg[i][j][3];
            if (dx < 0) {
                const int t = j == 0 ? 0 : (j == 3 ? 1 : 2);
                const float nb = dpp_from_prev_lane(q3);
                const float lf = fix_left ? (edge_in(j, true) ? edge[i][t] : 0.f) : nb;
                wreg[i][j][0] = TRANS ? lf : fabsf(lf); wreg[i][j][1] = TRANS ? q0 : fabsf(q0); wreg[i][j][2] = TRANS ? q1 : fabsf(q1); wreg[i][j][3] = TRANS ? q2 : fabsf(q2);
            } else if (dx > 0) {
                const int t = j == 2 ? 3 : (j == 4 ? 4 : 5);
                const float nb = dpp_from_next_lane(q0);
                const float rt = fix_right ? (edge_in(j, false) ? edge[i][t] : 0.f) : nb;
                wreg[i][j][0] = TRANS ? q1 : fabsf(q1); wreg[i][j][1] = TRANS ? q2 : fabsf(q2); wreg[i][j][2] = TRANS ? q3 : fabsf(q3); wreg[i][j][3] = TRANS ? rt : fabsf(rt);
            } else {
                wreg[i][j][0] = TRANS ? q0 : fabsf(q0); wreg[i][j][1] = TRANS ? q1 : fabsf(q1); wreg[i][j][2] = TRANS ? q2 : fabsf(q2); wreg[i][j][3] = TRANS ? q3 : fabsf(q3);
            }
        }
        // S in the reference's channel order k = 0..7 (tap 7..0), one shared refined reciprocal; 0 for padding quads
        float Sq[4] = {0.f, 0.f, 0.f, 0.f};
        if (TRANS) {                                   // the taps are final: only quads outside the image are zeroed
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j) wreg[i][j][e] = (ok && e < nval) ? wreg[i][j][e] : 0.f;
        } else
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float S = wreg[i][7][e];
#pragma unroll
            for (int k = 1; k < 8; ++k) S += wreg[i][7 - k][e];
            Sq[e] = S;
            float av[8], qv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = wreg[i][j][e];
            div8_shared_reciprocal(av, S, qv);
#pragma unroll
            for (int j = 0; j < NT; ++j) wreg[i][j][e] = (ok && e < nval) ? qv[j] : 0.f;
        }
        if (MODE == 2 && ((interior >> i) & 1u)) {   // publish before the blend is folded in: the backward wants w and S themselves
            if (a.w_out) store_taps_quad<NT>(a.w_out + (size_t)b * NT * HW, off, HW, wreg[i]);     // null: S only (the backward rebuilds w)
            st4(at32(uniform_ptr(a.s_out + (size_t)b * HW), off), make_float4(Sq[0], Sq[1], Sq[2], Sq[3]));
        }
    }
    // the blend is folded in by a second pass: the masks were requested BEHIND the guidance, so touching the first one
    // waits for the whole stream — after the weight arithmetic, which consumes the stream as it arrives, that wait is free
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const bool ok = (in_img >> i) & 1u;
        // EVERY slot of the thread is written, also those of rows past the region (r0 + i >= wr: m = 0 there).  The step body
        // computes those rows too (zero taps) and a CLEAN instance trusts the result to be exactly 0 — it is the lower neighbour
        // of the region's last row, i.e. the zero padding below the image for a bottom-edge tile — so the blend term it adds must
        // not be whatever the previous kernel on this CU left in LDS (round 5: the root cause of the one-off bit mismatch at
        // KITTI B = 8 with sparse depth, the one production shape with wr % NQ != 0; tests: the poisoned-LDS runs).
        if (BLEND) {
            // (1-m) u + m d0  ==  sum_j ((1-m) w_j) d_j + m d0 with 1-m in {0,1,2}: exact, so bit-identical (CSPN_new.py:90)
            const float4 m = make_float4(ok ? sgnf(mraw[i].x) : 0.f, ok ? sgnf(mraw[i].y) : 0.f, ok ? sgnf(mraw[i].z) : 0.f, ok ? sgnf(mraw[i].w) : 0.f);
            const float omq[4] = {1.f - m.x, 1.f - m.y, 1.f - m.z, 1.f - m.w};
            if (TRANS) {                               // PREMASK: the private plane holds 1-m, applied to every step's result
                *reinterpret_cast<float4*>(md_lds + (tid * NQ + i) * 4) = make_float4(omq[0], omq[1], omq[2], omq[3]);
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) wreg[i][j][e] *= omq[e];
                // the private slot receives m now and becomes m * d0 once the depth region is staged (the owned quads of
                // d0 are read from LDS there anyway: no second global load, no exposed round trip)
                *reinterpret_cast<float4*>(md_lds + (tid * NQ + i) * 4) = m;
            }
        }
    }

    stamp();                                   // weights derived
    // Everything below addresses through LATE, opaque copies of the strip coordinates: row / pixel offsets of the step loop,
    // the staging and the epilogue are then computed here, after the derive — computed early they sit in registers the
    // derive needs, get spilled, and every scratch reload in the derive waits for ALL outstanding guidance loads
    // (s_waitcnt vmcnt(0): loads return in order), which serialises the load stream and the weight arithmetic.
    int r0L = r0, sxL = sx, yq0L = yq0, xqL = xq;
    asm volatile("" : "+v"(r0L), "+v"(sxL), "+v"(yq0L), "+v"(xqL));
    // ---- 2. phases of S steps; between phases the tile borders travel through the exchange planes ---------------------
    const bool active = (r0L < wr);
    const int cb = 4 + 4 * sxL;
    float* __restrict__ dout = HIST ? nullptr : uniform_ptr(a.out + (size_t)b * HW);
    float* hist_step = HIST ? uniform_ptr(a.hist + (size_t)b * HW) : nullptr;      // plane of the step being computed (uniform)
    const int n_phase = (a.T + a.S - 1) / a.S;
    const int tile_global = b * tiles_per_img + trem;
    float own[NQ][4];
    float nbl[NQ], nbr[NQ];                    // -DCSPN_RES_STEP_PIPE=1: columns xq-1 / xq+4 of the own rows as the neighbouring lanes hold them (DPP)

    for (int p = 0; p < n_phase; ++p) {
        const int steps = (a.T - p * a.S) < a.S ? (a.T - p * a.S) : a.S;
        const bool last_phase = (p == n_phase - 1);
        // -- stage the depth region (weight region + 1 ring).  Phase 0: everything from the coarse depth.  Later phases:
        //    the tile's own interior is already in LDS (`cur`), only the halo comes from the neighbours' published borders
        const float* __restrict__ xin = uniform_ptr(a.xbuf + (size_t)((p + 1) & 1) * plane + (size_t)b * HW);   // written in phase p-1
        if (p > 0) {
            // halo quads only: the full rows above and below the tile rows, the quads left and right of the tile columns.  All
            // device-scope loads of a batch are requested before the first one is consumed (branch-free: safe address + select).
            int tidp = tid;                                    // opaque copy: the per-thread halo addressing below is recomputed
            asm volatile("" : "+v"(tidp));                     // every phase instead of living (spilled) across the step loop
            const int tq_in = min(a.tw, rx0 + 4 * wq - x0) >> 2;   // tile columns / rows that lie inside the region (the last
            const int th_in = min(a.th, ry0 + wr - y0);            // tile of an image may be cut short by the image edge)
            const int nl = (x0 - rx0) >> 2;                    // quads left of the tile columns inside the region
            const int nside = wq - tq_in;                      // ... left + right
            const int nrow_t = (y0 - ry0) + R;                 // depth-region rows above the tile rows (ring row included)
            const int n_top = nrow_t * wq;
            const int n_bot = (dr - nrow_t - th_in) * wq;
            const int n_side = th_in * nside;
            const int n_halo = n_top + n_bot + n_side;
            // the 1-pixel ring columns left / right of the region: one scalar per row and side
            float ring_v = 0.f;
            int ring_at = -1;
            if (tidp < dr * 2 * R) {
                const int row = tidp / (2 * R), c = tidp - row * (2 * R);
                const int lc = (c < R) ? (4 - R + c) : (4 + 4 * wq + (c - R));
                const int y = yd0 + row, x = xd0 + lc;
                const bool in = (y >= 0 && y < H && x >= 0 && x < a.Wv);
                const float v = ld1_dev(at32(xin, in ? (unsigned)(y * W + x) : 0u));
                ring_v = in ? v : 0.f;
                ring_at = row * ls + lc;
            }
            for (int base = 0; base < n_halo; base += 2 * NTHREADS) {
                float4 hv[2];
                int at[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int h = base + u * NTHREADS + tidp;
                    int row, qx;
                    if (h < n_top) { row = h / wq; qx = h - row * wq; }
                    else if (h < n_top + n_bot) { const int h2 = h - n_top; row = h2 / wq; qx = h2 - row * wq; row += nrow_t + th_in; }
                    else {
                        const int h3 = h - n_top - n_bot;
                        const int ns = nside > 0 ? nside : 1;
                        row = h3 / ns;
                        const int c = h3 - row * ns;
                        row += nrow_t;
                        qx = c < nl ? c : c + tq_in;
                    }
                    const int y = yd0 + row, x = xd0 + 4 + 4 * qx;
                    const bool valid = h < n_halo;
                    const bool in = valid && y >= 0 && y < H && x >= 0 && x < a.Wv;
                    const float4 v = ld4_dev(at32(xin, in ? (unsigned)(y * W + x) : 0u));
                    hv[u] = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
                    at[u] = valid ? row * ls + 4 + 4 * qx : -1;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (at[u] >= 0) *reinterpret_cast<float4*>(&cur[at[u]]) = hv[u];
            }
            if (ring_at >= 0) { cur[ring_at] = ring_v; nxt[ring_at] = 0.f; }
            for (int c = tidp; c < 2 * ls; c += NTHREADS) nxt[(c < ls ? 0 : (dr - 1) * ls - ls) + c] = 0.f;   // see phase 0
        }
        __syncthreads();

#pragma unroll
        for (int i = 0; i < NQ; ++i) {             // rows past the region read the clamped last row; they are never stored
            int drow = r0L + i + R;
            drow = drow < dr ? drow : dr - 1;
            const v4f mid = *(lds_cv4f_ptr)(cur + drow * ls + cb);
            own[i][0] = mid.x; own[i][1] = mid.y; own[i][2] = mid.z; own[i][3] = mid.w;
        }
        if (CSPN_RES_STEP_PIPE) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) { nbl[i] = dpp_from_prev_lane(own[i][3]); nbr[i] = dpp_from_next_lane(own[i][0]); }
        }
        if (BLEND && !TRANS && p == 0) {           // private slots: m -> m * d0 (own = d0 here)
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (r0L + i < wr) {
                    float4* slot = reinterpret_cast<float4*>(md_lds + (tid * NQ + i) * 4);
                    const float4 m = *slot;
                    // plain products, as the reference's  m * d0  (0 * inf = nan spreads like there); quads outside the
                    // image hold m = 0 and a zero-padded d0
                    *slot = make_float4(m.x * own[i][0], m.y * own[i][1], m.z * own[i][2], m.w * own[i][3]);
                }
            }
        }
        // One propagation step on the LDS tile.  FINAL (the very last step of the forward) is peeled into its own copy so
        // that the target quads of the fused metrics are only live there, not across the hot loop.
        auto step = [&](auto final_c, auto on_c, auto par_c, int final_par) __attribute__((always_inline)) {
            constexpr bool FINAL = decltype(final_c)::value;
            constexpr int PAR = decltype(par_c)::value;       // buffer this step reads (the final step: `final_par`)
            if (decltype(on_c)::value) {
                float win[NQ + 2 * R][WIN];
                const float* const rd = FINAL ? lds + final_par * RES_PP : lds + PAR * RES_PP;
                float* const wrb = lds + (1 - PAR) * RES_PP;
                // the peeled final step runs once: its row / pixel offsets are recomputed from opaque copies instead of being
                // kept (spilled) across the hot loop
                int r0x = r0L, yqx = yq0L;
                if (FINAL) asm volatile("" : "+v"(r0x), "+v"(yqx));
                // ... and ONE pixel offset for its stores: formed per row from the strip coordinates, a spilled coordinate was reloaded from
                // scratch — with a wait for the reload — in front of every row's store (5 x ~0.7 us in the blended scored instance: round 6)
                unsigned o0x = 0u;
                if (FINAL && !HIST) { o0x = (unsigned)(yqx * W + xqL); asm volatile("" : "+v"(o0x)); }
                auto row_ptr = [&](int rr) -> const float* {
                    int drow = r0x + rr;
                    drow = drow < dr ? drow : dr - 1;
                    return rd + drow * ls + cb;
                };
                if (CSPN_RES_STEP_PIPE) {
                    // every LDS read of the step is issued right behind the barrier: the two halo rows' quads, then — exec-masked, strip-end /
                    // wave-edge lanes only — the own rows' neighbour columns (straight into the window, over the DPP values taken before the
                    // barrier) and the halo rows' (into temporaries: the DPP of a halo row has to wait for its quad, the read need not)
                    v4f hq[2 * R];
                    float tl[2 * R], tr[2 * R];
#pragma unroll
                    for (int h = 0; h < 2 * R; ++h) {
                        hq[h] = *(lds_cv4f_ptr)(row_ptr(h < R ? h : NQ + h));
                        tl[h] = 0.f; tr[h] = 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) win[i + R][R + c] = own[i][c];
                        win[i + R][0] = nbl[i]; win[i + R][5] = nbr[i];
                    }
                    if (fix_left) {
#pragma unroll
                        for (int i = 0; i < NQ; ++i) win[i + R][0] = row_ptr(i + R)[-1];
#pragma unroll
                        for (int h = 0; h < 2 * R; ++h) tl[h] = row_ptr(h < R ? h : NQ + h)[-1];
                    }
                    if (fix_right) {
#pragma unroll
                        for (int i = 0; i < NQ; ++i) win[i + R][5] = row_ptr(i + R)[4];
#pragma unroll
                        for (int h = 0; h < 2 * R; ++h) tr[h] = row_ptr(h < R ? h : NQ + h)[4];
                    }
#pragma unroll
                    for (int h = 0; h < 2 * R; ++h) {
                        const int rr = h < R ? h : NQ + h;
                        win[rr][1] = hq[h].x; win[rr][2] = hq[h].y; win[rr][3] = hq[h].z; win[rr][4] = hq[h].w;
                        const float dl = dpp_from_prev_lane(hq[h].w), dr2 = dpp_from_next_lane(hq[h].x);
                        win[rr][0] = fix_left ? tl[h] : dl;
                        win[rr][5] = fix_right ? tr[h] : dr2;
                    }
                } else {
#pragma unroll
                for (int rr = 0; rr < NQ + 2 * R; ++rr) {
                    float m4[4];
                    const bool own_row = rr >= R && rr < R + NQ;
                    if (own_row) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) m4[c] = own[rr - R][c];
                    } else {
                        const v4f mid = *(lds_cv4f_ptr)(row_ptr(rr));
                        m4[0] = mid.x; m4[1] = mid.y; m4[2] = mid.z; m4[3] = mid.w;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) win[rr][R + c] = m4[c];
                    win[rr][0] = dpp_from_prev_lane(m4[3]);
                    win[rr][5] = dpp_from_next_lane(m4[0]);
                }
                // strip-end / wave-edge lanes patch their halo column from LDS: exec-masked ds_read straight into the window
                // registers (a merged block with selects costs 28 v_mov per step and measured 7 % slower)
#ifdef CSPN_RES_EXPERIMENT_NOFIX      // perf experiment ONLY (wrong results): what do the masked 4-byte reads cost?
                if (false) {
#else
                if (fix_left) {
#endif
#pragma unroll
                    for (int rr = 0; rr < NQ + 2 * R; ++rr) win[rr][0] = row_ptr(rr)[-1];
                }
#ifdef CSPN_RES_EXPERIMENT_NOFIX
                if (false) {
#else
                if (fix_right) {
#endif
#pragma unroll
                    for (int rr = 0; rr < NQ + 2 * R; ++rr) win[rr][5] = row_ptr(rr)[4];
                }
                }
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    // rows past the region (r0L + i >= wr, only in the last row group) are computed too — their taps are
                    // zero — and only their LDS store is masked: a branch around the arithmetic costs exec-masked copies of
                    // the carried quads on every step
                    {
                        float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                            for (int dx = -1; dx <= 1; ++dx) {
                                if (dy == 0 && dx == 0) continue;
                                const int lin = (dy + 1) * 3 + (dx + 1);
                                const int j = lin < 4 ? lin : lin - 1;
#pragma unroll
                                for (int e = 0; e < 4; ++e) u[e] = fmaf(wreg[i][j][e], win[i + dy + 1][e + dx + 1], u[e]);
                            }
                        float keep[4];                 // the state carried to the next step
                        if (BLEND) {
                            const float4 m4 = *reinterpret_cast<const float4*>(md_lds + (tid * NQ + i) * 4);
                            if (TRANS) {
                                keep[0] = m4.x * u[0]; keep[1] = m4.y * u[1]; keep[2] = m4.z * u[2]; keep[3] = m4.w * u[3];
                            } else {
                                u[0] += m4.x; u[1] += m4.y; u[2] += m4.z; u[3] += m4.w;
                            }
                        }
                        if (!(TRANS && BLEND)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) keep[e] = u[e];
                        }
                        if (!CLEAN) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (!((in_img >> i) & 1u) || e >= nval) { u[e] = 0.f; keep[e] = 0.f; }   // zero padding stays exactly zero
                        }
                        if (!FINAL && r0L + i < wr)
                            *reinterpret_cast<float4*>(&wrb[(r0L + i + R) * ls + cb]) = make_float4(keep[0], keep[1], keep[2], keep[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) own[i][e] = keep[e];
                        if (HIST) {
                            if ((interior >> i) & 1u) st4_hist(at32(hist_step, (unsigned)((yq0L + i) * W + xqL)), make_float4(u[0], u[1], u[2], u[3]));
                        } else if (FINAL && ((interior >> i) & 1u)) {
                            st4(at32(dout, o0x + (unsigned)(i * W)), make_float4(u[0], u[1], u[2], u[3]));
                        }
                    }
                }
            }
            if (HIST) hist_step += plane;
            if (CSPN_RES_STEP_PIPE && !FINAL && decltype(on_c)::value) {
                // the next step's neighbour columns of the own rows, in the shadow of the ds_write latency (pinned in front of the barrier)
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    nbl[i] = dpp_from_prev_lane(own[i][3]); nbr[i] = dpp_from_next_lane(own[i][0]);
                    asm volatile("" : "+v"(nbl[i]), "+v"(nbr[i]));
                }
            }
#ifndef CSPN_RES_EXPERIMENT_NOBARRIER  // perf experiment ONLY (wrong results): what do the step barriers cost?
            if (!FINAL) __syncthreads();
#endif
        };
        stamp();                               // depth staged
        const int plain_steps = last_phase ? steps - 1 : steps;
        // Wavefronts without a single owned row (the tail of the last row group) only keep the barriers company.  The test is
        // wave-uniform (a scalar branch), and the step loop sits INSIDE it: a per-step `if (active)` makes every carried quad
        // a phi of "old" and "new" that is resolved with 20 v_mov per step.  Idle lanes of a working wavefront run the
        // arithmetic on clamped rows with zero taps; only their stores are masked.  Two steps per trip let the carried
        // quads alternate between two register sets.
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        if (__ballot(active) != 0ull) {
            int s = 0;
            for (; s + 1 < plain_steps; s += 2) {
                step(std::false_type{}, std::true_type{}, P0{}, 0);
                step(std::false_type{}, std::true_type{}, P1{}, 0);
            }
            if (s < plain_steps) step(std::false_type{}, std::true_type{}, P0{}, 0);
            if (last_phase) step(std::true_type{}, std::true_type{}, P0{}, plain_steps & 1);
        } else {
            for (int s = 0; s < plain_steps; ++s) step(std::false_type{}, std::false_type{}, P0{}, 0);
            if (last_phase) step(std::true_type{}, std::false_type{}, P0{}, 0);
        }
        stamp();                               // steps of the phase done
        if (!last_phase) {
            // -- publish the interior quads (device scope), then the phase flag; wait for the 8 neighbouring tiles
            float* __restrict__ xout = uniform_ptr(a.xbuf + (size_t)(p & 1) * plane + (size_t)b * HW);
            if (active) {
                // (opaque to the optimiser: the per-quad offsets are recomputed here, not hoisted out of the phase loop into
                // registers the step loop needs — they would be spilled and reloaded from scratch every phase)
                unsigned o0 = (unsigned)(yq0L * W + xqL);
                asm volatile("" : "+v"(o0));
                // -DCSPN_RES_PUBLISH_ALL=0 (round 5 experiment, not faster): only what a neighbour will READ is published — the columns / rows of this
                // tile that lie inside an adjacent tile's depth region (its weight region + the 1-pixel ring), worked out from the
                // neighbours' region origins exactly as they work them out themselves.  Corner neighbours read the intersection of
                // a column band and a row band, so the union of the four bands covers all eight.  At config 2 that is 57 % of the
                // interior quads: fewer device-scope stores to wait for before the flag, 6 MB less write traffic per forward.
#if !CSPN_RES_PUBLISH_ALL
                auto reg_x0 = [&](int t) { const int xx0 = t * a.tw; return max(max(0, xx0 - a.tw), min(xx0 - a.hxw, W - 4 * wq)); };
                auto reg_y0 = [&](int t) { const int yy0 = t * a.th; return max(max(0, yy0 - a.th), min(yy0 - a.hyw, H - wr)); };
                const int pubL = tx > 0 ? reg_x0(tx - 1) + 4 * wq + 1 : -(1 << 30);            // columns x < pubL are read by the tiles on the left
                const int pubR = tx + 1 < a.tiles_x ? reg_x0(tx + 1) - 1 : (1 << 30);          // columns x >= pubR by the tiles on the right
                const int pubT = ty > 0 ? reg_y0(ty - 1) + wr + 1 : -(1 << 30);                // rows y < pubT by the tiles above
                const int pubB = ty + 1 < a.tiles_y ? reg_y0(ty + 1) - 1 : (1 << 30);          // rows y >= pubB by the tiles below
                const bool col_read = (xqL < pubL) || (xqL + 3 >= pubR);
#endif
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
#if !CSPN_RES_PUBLISH_ALL
                    const bool read = col_read || (yq0L + i < pubT) || (yq0L + i >= pubB);
#else
                    const bool read = true;
#endif
                    if (((interior >> i) & 1u) && read) st4_dev(xout, o0 + (unsigned)(i * W), own[i][0], own[i][1], own[i][2], own[i][3]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's device-scope stores have landed
            __syncthreads();                                       // ... and so have everybody else's in the workgroup
            const unsigned want = a.seq + (unsigned)p + 1u;
            if (tid == 0) __hip_atomic_store(a.flags + tile_global, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 9 && tid != 4) {
                const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                if (ny >= 0 && ny < a.tiles_y && nx >= 0 && nx < a.tiles_x) {
                    // (SGPR base + a 32-bit VGPR offset: as a 64-bit per-thread pointer the flag address was spilled in the blended instances and
                    //  re-loaded from scratch — with a wait — in front of EVERY poll)
                    const gptr f = (gptr) reinterpret_cast<unsigned long long>(uniform_ptr(a.flags)) + ((unsigned)(b * tiles_per_img + ny * a.tiles_x + nx) << 2);
                    unsigned spins = 0;
                    while ((int)(__hip_atomic_load(reinterpret_cast<const GLB unsigned*>(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                        ++spins;
                        if ((spins & 255u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.seq) {
                            wg_bad = 1;
                            break;
                        }
                        if (spins >= a.spin_limit) {                // a neighbour never became resident / finished: give up
                            __hip_atomic_store(a.status, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            wg_bad = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
            }
            __syncthreads();
            stamp();                           // neighbours' borders published
            if (wg_bad) {
                if (tid == 0) {
                    __hip_atomic_store(a.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (a.host_err) __hip_atomic_store(a.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __atomic_thread_fence(__ATOMIC_SEQ_CST);       // the error is visible to the host before this tile counts out
                }
                // Poison what this tile will never produce: its part of the refined depth (or of the last history plane —
                // d_T in the training forward, G_0 in the reverse sweep) becomes NaN, so that a result consumed before the
                // host has looked at the error word is visibly wrong instead of stale allocator memory.
                {
                    const float qnan = __uint_as_float(CSPN_POISON_F32);      // NaN with the payload the host looks for
                    float* const pz = HIST ? uniform_ptr(a.hist + (size_t)(a.T - 1) * plane + (size_t)b * HW) : dout;
                    unsigned o0 = (unsigned)(yq0L * W + xqL);
                    asm volatile("" : "+v"(o0));
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        if ((interior >> i) & 1u) st4(at32(pz, o0 + (unsigned)(i * W)), make_float4(qnan, qnan, qnan, qnan));
                }
                count_out();
                return;
            }
        }
    }
    if (SCORE) {
        float mf[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) mf[k] = 0.f;
        if (active) {
            // the target quads are requested only now: the weight registers are dead, nothing spills, and the one exposed
            // round trip costs less than carrying NQ quads through the final step did
            float4 scored_t[NQ];
            const float* tgt_b = uniform_ptr(a.target + (size_t)b * HW);
            unsigned o0s = (unsigned)(yq0L * W + xqL);       // one offset for all rows (see the final step's stores)
            asm volatile("" : "+v"(o0s));
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const bool in = (interior >> i) & 1u;
                scored_t[i] = ld4(at32(tgt_b, in ? o0s + (unsigned)(i * W) : 0u));
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if ((interior >> i) & 1u) {
                    const float4 tg = scored_t[i];
                    const float t4[4] = {tg.x, tg.y, tg.z, tg.w};
                    const float o4[4] = {own[i][0], own[i][1], own[i][2], own[i][3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) metric_terms(o4[e], t4[e], mf);
                }
            }
        }
        float* part = lds + RES_PP + (size_t)a.dr * a.ls + (size_t)(BLEND ? 1 : 0) * NTHREADS * NQ * 4;
        const int wave = tid >> 6;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float v = wave_sum_to_lane63(mf[k]);
            if (lane == 63) part[wave * 10 + k] = v;
        }
        __syncthreads();
        if (tid < 10) {
            double v = 0.0;
            for (int w = 0; w < NTHREADS / 64; ++w) v += (double)part[w * 10 + tid];
            if (v != 0.0) atomicAdd(a.macc + (size_t)(blockIdx.x % a.nslots) * 10 + tid, v);
        }
    }
    count_out();
}

// ------------------------------------------------------------------------------------------------ host side
constexpr int RES_MAX_NQ = 5;

constexpr int RES_MAX_ROUNDS = 8;       // rounds of images_per_launch images in one launch (-DCSPN_RES_ONE_LAUNCH=1)
// Rounds in one launch: measured, NOT the default (same box, three alternating passes, profiles/r05_rounds_ab.txt): KITTI B = 8 (2 rounds of
// 4 images on 256 workgroups) 92.4-93.4 us per scored forward against 92.2-93.7 with one launch per round, NYU B = 48 91.7-92.1 against
// 90.4-90.9 (slower), only a ragged third round gains (KITTI B = 9: 121-123 against 126-127).  A round-1 workgroup starts when a CU is freed,
// but its neighbours start as scattered as round 0's workgroups finish: the neighbour waits pay what the drain and the launch gap cost.
// The K = 5 reverse sweep (cspnk_resident.hip) does gain (2 x 45.7 + gap -> 88.5 us) and runs its rounds in one launch.
#ifndef CSPN_RES_ONE_LAUNCH
#define CSPN_RES_ONE_LAUNCH 0
#endif
struct ResGeom {
    int S, tiles_x, tiles_y, tw, th, nq, wq, wr, hxw, hyw, dr, ls;
    int imgs_per_launch, launches;
    size_t lds_bytes;
    double cost;
    int threads = RES_THREADS;      // 512, or 1024 with one quad per thread (four wavefronts per SIMD: small shards, round 5)
};

int cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int n = cached[dev & 63].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    cached[dev & 63].store(prop.multiProcessorCount, std::memory_order_relaxed);
    return prop.multiProcessorCount;
}

// LDS of one workgroup: buffer 0 in the first RES_PP floats, buffer 1 behind it, then the private m * d0 quads (sparse
// blend) and the 10 x 16 partial sums of the fused metrics.
size_t res_lds_bytes(int dr, int ls, int nq, int blend, int threads = RES_THREADS) {
    return ((size_t)RES_PP + (size_t)dr * ls + (size_t)(blend ? 1 : 0) * threads * nq * 4 + 16 * 10) * sizeof(float);
}

// Row stride (floats) of a depth buffer in LDS: 4 wq + 8 at least ([4 pad incl. ring][wq quads][ring + pad]).  Round 5 (-DCSPN_RES_LS_SWIZZLE=0
// for A/B; same-box, profiles/r05_swizzle_ab.txt: NYU B = 3 20.8 -> 20.1 us per scored forward, the other shapes within noise): the smallest such stride for which the next strip of a thread column (NQ rows further down) continues the bank
// sequence of this one inside a wavefront — NQ ls = 4 wq (mod 64 dwords) — so that the 16 lanes of a ds_read_b128 / ds_write_b128
// pass that straddle two strips do not share banks (SQ counters of round 4: 16.5 % of the LDS cycles were bank conflicts).
#ifndef CSPN_RES_LS_SWIZZLE
#define CSPN_RES_LS_SWIZZLE 1
#endif
int res_row_stride(int wq, int nq, int dr) {
    const int lo = 4 * wq + 8;
    if (!CSPN_RES_LS_SWIZZLE) return lo;
    for (int cand = lo; cand < lo + 64; cand += 4)
        if ((((nq * cand - 4 * wq) % 64) + 64) % 64 == 0 && (size_t)dr * cand <= (size_t)RES_PP) return cand;
    return lo;
}

// Mirror of the kernel's region placement: does every region of every tile lie inside the (valid part of the) image?
bool regions_inside_image(const ResGeom& g, int H, int W, int Wv, int T) {
    if (Wv != W) return false;
    const bool exch = T > g.S;            // mirrors the kernel: see the region origin there
    for (int tx = 0; tx < g.tiles_x; ++tx) {
        const int x0 = tx * g.tw;
        int rx0 = x0 - g.hxw; if (rx0 > W - 4 * g.wq) rx0 = W - 4 * g.wq;
        int lo = exch ? x0 - g.tw : 0; if (lo < 0) lo = 0;
        if (rx0 < lo) rx0 = lo;
        if (rx0 + 4 * g.wq > W) return false;
    }
    for (int ty = 0; ty < g.tiles_y; ++ty) {
        const int y0 = ty * g.th;
        int ry0 = y0 - g.hyw; if (ry0 > H - g.wr) ry0 = H - g.wr;
        int lo = exch ? y0 - g.th : 0; if (lo < 0) lo = 0;
        if (ry0 < lo) ry0 = lo;
        if (ry0 + g.wr > H) return false;
    }
    return true;
}

// Tiling of one image for the resident kernel: every workgroup owns a tw x th tile + (S-1) halo; a launch holds
// imgs_per_launch whole images on at most `ncu` workgroups.  Cost model: launches x (quads per thread + latency floor).
bool resident_geometry(int B, int H, int W, int T, int blend, int ncu, int S_user, ResGeom* best, int threads = RES_THREADS) {
    if (W % 4 != 0 || ncu < 1 || T < 1) return false;
    if (threads != 1024) threads = RES_THREADS;
    const int max_nq = threads == 1024 ? 1 : RES_MAX_NQ;       // 1024 threads = 128 VGPRs: one quad's weights per thread
    bool found = false;
    // phase lengths 12 / 8 / 6 / 4: twelve (one exchange at T = 24) wins on small shards whose tiles stay at two quads per
    // thread (NYU B = 6: 22.2 vs 23.9 us), eight everywhere else (tools/probes/resident_s_sweep.py)
    for (int S = (S_user > 0 ? S_user : 12); S >= (S_user > 0 ? S_user : 4); S -= (S > 8 ? 4 : 2)) {
        const int Se = S > T ? T : S;
        const int hyw = Se - 1, hxw = round_up4(Se - 1);
        const int phases = ceil_div(T, Se);
        if (phases > 1 && (Se & 1)) continue;          // every phase must start in buffer 0 (see RES_PP)
        for (int tx = 1; tx <= 32; ++tx) {
            const int tw = round_up4(ceil_div(W, tx));
            if (tx > 1 && (tw < 16 || ceil_div(W, tw) != tx)) continue;
            if (phases > 1 && tx > 1 && tw < 2 * hxw) continue;                   // the (shifted) halo must come from adjacent tiles only
            const int wq = (tw + 2 * hxw) / 4;
            if (wq > threads) continue;
            for (int ty = 1; ty <= 64; ++ty) {
                const int th = ceil_div(H, ty);
                if (ty > 1 && (th < 4 || ceil_div(H, th) != ty)) continue;
                if (phases > 1 && ty > 1 && th < 2 * hyw) continue;
                const int tiles = tx * ty;
                if (tiles > ncu) continue;
                const int wr = th + 2 * hyw;
                const int rows_per_thread_col = threads / wq;                     // strips per quad column
                const int nq = ceil_div(wr, rows_per_thread_col);
                if (nq > max_nq) continue;
                const int dr = wr + 2, ls = res_row_stride(wq, nq, dr);
                if ((size_t)dr * ls > (size_t)RES_PP) continue;                      // one depth buffer per RES_PP slot
                const size_t ldsb = res_lds_bytes(dr, ls, nq, blend, threads);
                if (ldsb > 160 * 1024) continue;
                int ipl = ncu / tiles;
                if (ipl > B) ipl = B;
                const int launches = ceil_div(B, ipl);
                ResGeom cand{Se, tx, ty, tw, th, nq, wq, wr, hxw, hyw, dr, ls, ipl, launches, ldsb, 0.0};
                // microseconds per launch, fitted on MI355X (profiles/r02_resident_vs_multilaunch.jsonl): launch + epilogue,
                // derive (~2 us per quad of a thread), T steps (VALU-bound: 0.13 us per quad; x 1.13 with the zero-padding
                // selects of a launch whose regions stick out of the image), and per phase boundary the publish / wait /
                // halo staging (3 us + the border bytes)
                const double pen = regions_inside_image(cand, H, W, W, T) ? 1.0 : 1.13;
                const double cost = launches * (8.0 + 2.0 * nq + T * (0.13 * nq + 0.1) * pen + (phases - 1) * (3.0 + 0.6 * nq));
                if (!found || cost < best->cost) {
                    found = true;
                    *best = ResGeom{Se, tx, ty, tw, th, nq, wq, wr, hxw, hyw, dr, ls, ipl, launches, ldsb, cost};
                    best->threads = threads;
                }
            }
        }
    }
    return found;
}

template <int NQ, int BLEND, int MODE, int CLEAN, int PAC = 0, int NTH = RES_THREADS>
int launch_resident_inst(const ResArgs& a, int grid, size_t lds_bytes, hipStream_t st) {
    constexpr auto kern = cspn3_resident<NQ, NTH, BLEND, MODE, CLEAN, PAC>;
    static std::atomic<size_t> granted[64];
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (lds_bytes > 64 * 1024 && granted[dev & 63].load(std::memory_order_acquire) < lds_bytes) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        granted[dev & 63].store(lds_bytes, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), lds_bytes, st, a);
    HIP_OK(hipGetLastError());
    return 1;
}

// 1024 threads, one quad per thread: the inference forms only (plain / scored), CSPN_new weights
template <int CLEAN>
int launch_resident_1024(const ResArgs& a, int grid, size_t lds, int blend, int mode, hipStream_t st) {
    if (mode == 0) return blend ? launch_resident_inst<1, 1, 0, CLEAN, 0, 1024>(a, grid, lds, st) : launch_resident_inst<1, 0, 0, CLEAN, 0, 1024>(a, grid, lds, st);
    if (mode == 1) return blend ? launch_resident_inst<1, 1, 1, CLEAN, 0, 1024>(a, grid, lds, st) : launch_resident_inst<1, 0, 1, CLEAN, 0, 1024>(a, grid, lds, st);
    return fail("cspn3_forward_resident: 1024-thread workgroups serve the inference forms only");
}

template <int NQ, int CLEAN>
int launch_resident_c(const ResArgs& a, int grid, size_t lds, int blend, int mode, hipStream_t st) {
    if (mode >= 4 && mode <= 6) {          // softmax-weight (PAC) forms of MODE 0 / 1 / 2
        if (mode == 6) return blend ? launch_resident_inst<NQ, 1, 2, CLEAN, 1>(a, grid, lds, st) : launch_resident_inst<NQ, 0, 2, CLEAN, 1>(a, grid, lds, st);
        if (blend) return mode == 5 ? launch_resident_inst<NQ, 1, 1, CLEAN, 1>(a, grid, lds, st) : launch_resident_inst<NQ, 1, 0, CLEAN, 1>(a, grid, lds, st);
        return mode == 5 ? launch_resident_inst<NQ, 0, 1, CLEAN, 1>(a, grid, lds, st) : launch_resident_inst<NQ, 0, 0, CLEAN, 1>(a, grid, lds, st);
    }
    if (mode == 7) return blend ? launch_resident_inst<NQ, 1, 4, CLEAN>(a, grid, lds, st) : launch_resident_inst<NQ, 0, 4, CLEAN>(a, grid, lds, st);
    if (mode == 3) return blend ? launch_resident_inst<NQ, 1, 3, CLEAN>(a, grid, lds, st) : launch_resident_inst<NQ, 0, 3, CLEAN>(a, grid, lds, st);
    if (blend) {
        if (mode == 2) return launch_resident_inst<NQ, 1, 2, CLEAN>(a, grid, lds, st);
        return mode ? launch_resident_inst<NQ, 1, 1, CLEAN>(a, grid, lds, st) : launch_resident_inst<NQ, 1, 0, CLEAN>(a, grid, lds, st);
    }
    if (mode == 2) return launch_resident_inst<NQ, 0, 2, CLEAN>(a, grid, lds, st);
    return mode ? launch_resident_inst<NQ, 0, 1, CLEAN>(a, grid, lds, st) : launch_resident_inst<NQ, 0, 0, CLEAN>(a, grid, lds, st);
}
template <int NQ>
int launch_resident_nq(const ResArgs& a, int grid, size_t lds, int blend, int mode, bool clean, hipStream_t st) {
    return clean ? launch_resident_c<NQ, 1>(a, grid, lds, blend, mode, st) : launch_resident_c<NQ, 0>(a, grid, lds, blend, mode, st);
}


}  // namespace

namespace {
int resident_launch(const void* guidance, long bs, long cs, const void* d0, const void* sparse, void* out, void* history,
                    void* w8_out, float* s_out, void* work, unsigned seq, unsigned* host_err, int B, int H, int W, int W_valid,
                    int T, int blend, const void* target, double* acc, int nslots, const cspn_resident_plan* plan,
                    cspn_stream_t stream, bool transposed, bool pac = false, const float* s_in = nullptr);
}

extern "C" {

int cspn3_resident_plan(int B, int H, int W, int T, int blend, int n_cu, cspn_resident_plan* out) {
    if (!out || B < 1 || H < 1 || W < 1 || T < 0) return fail("cspn3_resident_plan: bad arguments");
    if (n_cu <= 0) n_cu = cu_count();
    if (n_cu <= 0) return fail("cspn3_resident_plan: no device (pass n_cu > 0 to plan without one)");
    ResGeom g;
    if (T < 1 || !resident_geometry(B, H, W, T, blend, n_cu, out->steps_per_phase, &g, out->threads))
        return fail("cspn3_resident_plan: no resident tiling for B=%d %dx%d T=%d on %d CUs (W %% 4 == 0 needed)", B, H, W, T, n_cu);
    if (ceil_div(T, g.S) > 255) return fail("cspn3_resident_plan: T=%d in %d-step phases is more than 255 phases", T, g.S);
    out->steps_per_phase = g.S; out->tiles_x = g.tiles_x; out->tiles_y = g.tiles_y; out->tile_w = g.tw; out->tile_h = g.th;
    out->quads_per_thread = g.nq; out->threads = g.threads; out->images_per_launch = g.imgs_per_launch;
    out->launches = g.launches; out->lds_bytes = (int)g.lds_bytes; out->n_cu = n_cu;
    out->region_over_tile = (float)((double)(4 * g.wq) * g.wr / ((double)g.tw * g.th));
    return 1;
}

size_t cspn3_resident_workspace_bytes(int B, int H, int W) {
    // 2 exchange planes, then 4 status words (abort, error, 2 reserved), then one flag per 16 pixels at most (tiles are >= 4 x 4)
    const size_t planes = (size_t)2 * B * H * W * sizeof(float);
    const size_t flags = ((size_t)B * (((size_t)H * W) / 16 + 1) + 4) * sizeof(unsigned);
    return ((planes + 15) & ~(size_t)15) + ((flags + 15) & ~(size_t)15);
}

int cspn3_forward_resident(const void* guidance, long bs, long cs, const void* d0, const void* sparse, void* out,
                           void* history, void* w8_out, float* s_out,
                           void* work, unsigned seq, unsigned* host_err, int B, int H, int W, int W_valid, int T, int blend,
                           const void* target, double* acc, int nslots, const cspn_resident_plan* plan,
                           cspn_stream_t stream) {
    return resident_launch(guidance, bs, cs, d0, sparse, out, history, w8_out, s_out, work, seq, host_err, B, H, W, W_valid, T,
                           blend, target, acc, nslots, plan, stream, /*transposed=*/false);
}

int cspn3_transposed_resident(const void* w8, const float* g_T, const float* sparse_f32, float* history, void* work,
                              unsigned seq, unsigned* host_err, int B, int H, int W, int W_valid, int T, int premask,
                              const cspn_resident_plan* plan, cspn_stream_t stream) {
    if (!w8 || !g_T || !history) return fail("cspn3_transposed_resident: bad arguments");
    if (premask && !sparse_f32) return fail("cspn3_transposed_resident: premask needs sparse");
    return resident_launch(w8, (long)8 * H * W, (long)H * W, g_T, premask ? sparse_f32 : nullptr, nullptr, history, nullptr, nullptr,
                           work, seq, host_err, B, H, W, W_valid, T, premask ? 1 : 0, nullptr, nullptr, 0, plan, stream,
                           /*transposed=*/true);
}

int cspn3_transposed_resident_guidance(const void* guidance, long bs, long cs, const float* S, const float* g_T,
                                       const float* sparse_f32, float* history, void* work, unsigned seq, unsigned* host_err,
                                       int B, int H, int W, int W_valid, int T, int premask, const cspn_resident_plan* plan,
                                       cspn_stream_t stream) {
    if (!guidance || !S || !g_T || !history) return fail("cspn3_transposed_resident_guidance: bad arguments");
    if (premask && !sparse_f32) return fail("cspn3_transposed_resident_guidance: premask needs sparse");
    if (!aligned16(S)) return fail("cspn3_transposed_resident_guidance: S must be 16-byte aligned");
    return resident_launch(guidance, bs, cs, g_T, premask ? sparse_f32 : nullptr, nullptr, history, nullptr, nullptr, work, seq,
                           host_err, B, H, W, W_valid, T, premask ? 1 : 0, nullptr, nullptr, 0, plan, stream,
                           /*transposed=*/true, /*pac=*/false, S);
}

}  // extern "C"

namespace cspn_detail {
// K = 3 softmax (CSPN_ours) forward with fp32 guidance [B,8,H,W] on the quad kernel: called by cspnk_forward_resident
// (cspnk_resident.hip), whose oct kernel holds only two or three octs of fp32 taps per thread.
int resident_pac3_f32(const void* guided, const void* x0, const void* sparse, void* out, void* history, void* wk_out, void* work,
                      unsigned seq, unsigned* host_err, int B, int H, int W, int T, int blend, const void* target, double* acc,
                      int nslots, const cspn_resident_plan* plan, cspn_stream_t stream) {
    return resident_launch(guided, (long)8 * H * W, (long)H * W, x0, sparse, out, history, wk_out, nullptr, work, seq, host_err, B, H, W, 0, T,
                           blend, target, acc, nslots, plan, stream, /*transposed=*/false, /*pac=*/true);
}
}  // namespace cspn_detail

namespace {

int resident_launch(const void* guidance, long bs, long cs, const void* d0, const void* sparse, void* out,
                    void* history, void* w8_out, float* s_out,
                    void* work, unsigned seq, unsigned* host_err, int B, int H, int W, int W_valid, int T, int blend,
                    const void* target, double* acc, int nslots, const cspn_resident_plan* plan,
                    cspn_stream_t stream, bool transposed, bool pac, const float* s_in) {
    if (!guidance || !d0 || !work || B <= 0 || H <= 0 || W <= 0 || T < 1 || (!out && !history))
        return fail("cspn3_forward_resident: bad arguments");
    if (!transposed && history && ((pac && !w8_out) || (!pac && !s_out) || target || acc || !aligned16(history) || (w8_out && !aligned16(w8_out)) || (s_out && !aligned16(s_out))))
        return fail("cspn3_forward_resident: the training form (history) needs s_out (w8_out is optional: null publishes S only), 16-byte aligned, and no scoring");
    if (!history && (w8_out || s_out)) return fail("cspn3_forward_resident: w8_out / s_out are outputs of the training form (history)");
    if (transposed && !aligned16(history)) return fail("cspn3_transposed_resident: history must be 16-byte aligned");
    if (blend != CSPN_BLEND_NONE && blend != CSPN_BLEND_SPARSE) return fail("cspn3_forward_resident: blend %d", blend);
    if (blend && !sparse) return fail("cspn3_forward_resident: blend needs sparse");
    if ((target || acc) && (!target || !acc || nslots < 1 || !aligned16(target)))
        return fail("cspn3_forward_resident: scoring needs target (16-byte aligned), acc and nslots >= 1");
    if ((W & 3) || (cs & 3) || (bs & 3)) return fail("cspn3_forward_resident: W and the guidance strides must be multiples of 4");
    if (cs < 0 || bs < 0 || cs >= (1L << 27) || (long)H * W >= (1L << 27))
        return fail("cspn3_forward_resident: images of >= 2^27 pixels / channel strides >= 2^27 elements are not supported (32-bit offsets)");
    if (!aligned16(guidance) || !aligned16(d0) || (out && !aligned16(out)) || !aligned16(work) || (sparse && !aligned16(sparse)))
        return fail("cspn3_forward_resident: tensors must be 16-byte aligned");
    if (W_valid < 0 || W_valid > W) return fail("cspn3_forward_resident: W_valid=%d outside (0, W=%d]", W_valid, W);
    if (seq == 0 || seq > 0x7fffff00u) return fail("cspn3_forward_resident: seq must be in [1, 2^31 - 256]");

    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ncu = cu_count();
    if (ncu <= 0) return fail("cspn3_forward_resident: no device");
    ResGeom g;
    cspn_resident_plan rp{};
    if (plan) rp = *plan;
    if (rp.tiles_x > 0 && rp.tiles_y > 0 && rp.tile_w > 0 && rp.tile_h > 0 && rp.steps_per_phase > 0 && rp.images_per_launch > 0) {
        // a plan that came out of cspn3_resident_plan for this problem: re-derive the dependent fields and re-check the
        // limits, skip the search (it is ~6000 candidate tilings: tens of microseconds per call on the host)
        const int Se = rp.steps_per_phase > T ? T : rp.steps_per_phase;
        g.S = Se; g.tiles_x = rp.tiles_x; g.tiles_y = rp.tiles_y; g.tw = rp.tile_w; g.th = rp.tile_h;
        g.hyw = Se - 1; g.hxw = round_up4(Se - 1);
        g.wq = (g.tw + 2 * g.hxw) / 4; g.wr = g.th + 2 * g.hyw;
        g.dr = g.wr + 2;
        g.threads = rp.threads == 1024 ? 1024 : RES_THREADS;
        g.nq = g.wq > 0 && g.wq <= g.threads ? ceil_div(g.wr, g.threads / g.wq) : RES_MAX_NQ + 1;
        if (g.threads == 1024 && g.nq > 1) g.nq = RES_MAX_NQ + 1;
        g.ls = res_row_stride(g.wq, g.nq, g.dr);
        g.lds_bytes = res_lds_bytes(g.dr, g.ls, g.nq, blend, g.threads);
        g.imgs_per_launch = rp.images_per_launch;
        const int phases = ceil_div(T, Se);
        if ((g.tw & 3) || g.nq > RES_MAX_NQ || g.lds_bytes > 160 * 1024 || (size_t)g.dr * g.ls > (size_t)RES_PP || (phases > 1 && (Se & 1)) || g.tiles_x * g.tw < W || g.tiles_y * g.th < H ||
            (long)g.imgs_per_launch * g.tiles_x * g.tiles_y > ncu ||
            (phases > 1 && ((g.tiles_x > 1 && g.tw < 2 * g.hxw) || (g.tiles_y > 1 && g.th < 2 * g.hyw))))
            return fail("cspn3_forward_resident: the plan does not fit this problem / device (use cspn3_resident_plan)");
    } else if (!resident_geometry(B, H, W, T, blend, ncu, rp.steps_per_phase, &g, rp.threads)) {
        return fail("cspn3_forward_resident: no resident tiling for B=%d %dx%d T=%d", B, H, W, T);
    }
    // a tile that finished phase p publishes seq + p + 1, and the next call on the workspace brings seq + 256: more than
    // 255 phases would let one call's flags satisfy the next call's waits
    if (ceil_div(T, g.S) > 255)
        return fail("cspn3_forward_resident: T=%d in %d-step phases is more than 255 phases (use the multi-launch schedule)", T, g.S);
    ResArgs a{};
    a.g = static_cast<const float*>(guidance); a.g_bs = bs; a.g_cs = cs;
    a.d0 = static_cast<const float*>(d0); a.sparse = static_cast<const float*>(sparse); a.out = static_cast<float*>(out);
    const size_t planes = (((size_t)2 * B * H * W * sizeof(float)) + 15) & ~(size_t)15;
    a.xbuf = static_cast<float*>(work);
    a.status = reinterpret_cast<unsigned*>(static_cast<char*>(work) + planes);   // fixed place, whatever the tiling
    a.flags = a.status + 4;
    if ((size_t)B * g.tiles_x * g.tiles_y > (size_t)B * (((size_t)H * W) / 16 + 1))
        return fail("cspn3_forward_resident: workspace too small for %d tiles", B * g.tiles_x * g.tiles_y);
    a.host_err = host_err;
    a.hist = static_cast<float*>(history); a.w_out = static_cast<float*>(w8_out); a.s_out = s_out;
    if (pac && transposed) return fail("cspn3_forward_resident: the softmax-weight form has no transposed sweep (use cspn3_transposed_resident on the volume)");
    const int mode = transposed ? (s_in ? 7 : 3) : (history ? 2 : (acc ? 1 : 0)) + (pac ? 4 : 0);
    a.s_in = s_in;
    a.seq = seq;
    a.target = static_cast<const float*>(target); a.macc = acc; a.nslots = nslots;
    a.B = B; a.H = H; a.W = W; a.Wv = (W_valid > 0 && W_valid < W) ? W_valid : W; a.T = T; a.S = g.S;
    a.tw = g.tw; a.th = g.th; a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
    a.wq = g.wq; a.wr = g.wr; a.hxw = g.hxw; a.hyw = g.hyw; a.dr = g.dr; a.ls = g.ls;
    a.spin_limit = rp.spin_limit ? rp.spin_limit : (4u << 20);   // x (sc1 load + s_sleep) ~ seconds
    a.dbg = rp.debug_stamps;
    const bool clean = regions_inside_image(g, H, W, a.Wv, T);
    // (refused rather than ignored: a host that asked for the guard relies on `out` being complete for GPU-side consumers)
    // mode: 0 / 1 / 2 inference / scored / training forward, 3 / 7 reverse sweep from a tap volume / from guidance + S, 4 / 5 / 6 the softmax forms
    if (rp.guard && ((mode != 0 && mode != 1 && mode != 2 && mode != 3 && mode != 7 && mode != 4 && mode != 6) || ((mode == 0 || mode == 1 || mode == 4) && !out) ||
                     !cspn_detail::resident_repair_fits(T)))
        return fail("cspn3_forward_resident: plan->guard serves inference (plain / scored), the training forward and the reverse sweeps, T <= 54 steps");
    // one launch per round of images_per_launch images (-DCSPN_RES_ONE_LAUNCH=1: per RES_MAX_ROUNDS rounds — measured, not faster: see there)
    const int ipl = g.imgs_per_launch < B ? g.imgs_per_launch : B;
    const int max_rounds = CSPN_RES_ONE_LAUNCH ? RES_MAX_ROUNDS : 1;
    for (int b0 = 0; b0 < B; b0 += ipl * max_rounds) {
        a.b0 = b0;
        a.nb = (B - b0) < ipl ? (B - b0) : ipl;
        const int rounds = ceil_div(B - b0, ipl) < max_rounds ? ceil_div(B - b0, ipl) : max_rounds;
        a.last_chunk = (b0 + ipl * max_rounds >= B) ? 1 : 0;
        const int grid = rounds * a.nb * g.tiles_x * g.tiles_y;
        int ok = 0;
        if (g.threads == 1024) {
            ok = clean ? launch_resident_1024<1>(a, grid, g.lds_bytes, blend, mode, st) : launch_resident_1024<0>(a, grid, g.lds_bytes, blend, mode, st);
            if (!ok) return 0;
            continue;
        }
        switch (g.nq) {
            case 1: ok = launch_resident_nq<1>(a, grid, g.lds_bytes, blend, mode, clean, st); break;
            case 2: ok = launch_resident_nq<2>(a, grid, g.lds_bytes, blend, mode, clean, st); break;
            case 3: ok = launch_resident_nq<3>(a, grid, g.lds_bytes, blend, mode, clean, st); break;
            case 4: ok = launch_resident_nq<4>(a, grid, g.lds_bytes, blend, mode, clean, st); break;
            case 5: ok = launch_resident_nq<5>(a, grid, g.lds_bytes, blend, mode, clean, st); break;
            default: return fail("cspn3_forward_resident: no instance for %d quads per thread", g.nq);
        }
        if (!ok) return 0;
    }
    if (rp.guard)
        return cspn_detail::resident_repair_launch(a.g, bs, cs, a.d0, a.sparse, a.out, a.hist, a.s_out, a.w_out, a.s_in,
                                                   mode == 7 ? 4 : mode == 4 ? 10 : mode == 6 ? 12 : mode, a.status, seq, B, H, W, a.Wv, T, blend ? 1 : 0, ncu, st,
                                                   a.target, a.macc, a.nslots);
    return 1;
}

}  // namespace

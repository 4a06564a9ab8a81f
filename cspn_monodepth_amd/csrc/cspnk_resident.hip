// cspnk_resident.hip — the K x K softmax (pixel-adaptive) propagation as weight-RESIDENT launches: fp16 taps in registers.
//
// Reference path: network/libs/post_process/CSPN_ours.py:24-54 (softmax over the K*K-1 guidance channels :35, zero centre
// tap :37-39, prop_time x { pac.conv2d :49, sparse blend :51-53 }) with network/libs/base/pac.py:89-92 as the step.
// BASELINE config 3: K = 5, 12 steps, fp16, B = 24 at 228x304.  The multi-launch schedule (cspn_pac_prepare + three S = 4
// launches of cspn_prop_fused + cspn_metrics) moves 429 MB per forward against 90 MB compulsory: the softmax writes an
// 80 MB tap volume that every launch streams again.  Here a workgroup owns ONE tile for the whole forward:
//   * it derives the softmax weights of its tile + halo from the raw fp16 guidance ONCE — same arithmetic as
//     cspn_pac_prepare_vec_kernel (softmax_exp<__half>, reciprocal_refined, round-to-nearest-even to fp16), so the taps are
//     bit-identical to the prepared volume's — and keeps them PACKED in VGPRs: a thread owns NO horizontally aligned
//     OCTS (8 pixels of one row: one 16-byte load per guidance channel), K*K-1 taps x 4 registers per oct, fed to
//     v_fma_mix_f32 through its op_sel bits (fma_h8).  No tap volume exists;
//   * the T steps run in phases of S steps on the fp32 depth tile in LDS, rows stored as two arrays of quads (even / odd
//     quads of the row) so that an oct is two conflict-free ds_read_b128 and the R-pixel halo columns are two ds_read_b64
//     of the neighbouring octs: no DPP, no divergent patch blocks (the step is VALU-bound on its 8 (K*K-1) FMAs per oct);
//   * between phases the tiles exchange borders through a global plane with device-scope (sc1) stores / loads and per-tile
//     phase flags, exactly as cspn3_resident (cspn_resident.hip) — same workspace layout, same bounded wait, same sticky
//     error word.
// State dtype: the depth planes (x0, sparse, out, target, exchange) are fp16 or fp32.  With fp16 state the multi-launch
// schedule rounds the state to half between launches; the resident launch rounds it at the phase boundaries, so with
// steps_per_phase = steps_per_launch the two schedules produce the same bits (tests/test_hip_kres.py).
//
// Register budget: 24 taps x 4 = 96 VGPRs per oct; NO = 2 at 512 threads (2 wavefronts per SIMD, 256 VGPRs).  The tap volume of
// config 3 (80 MB) does not fit the chip's register files with its halo (131 MB in all), so the batch goes through two
// launches of 12 images (20 tiles each); see DESIGN.md §4.1c for the arithmetic.
#include "cspnk_helpers.hpp"

#include <atomic>

namespace {

constexpr int KRES_THREADS = 512;        // default workgroup; 768 threads (3 wavefronts per SIMD, <= 168 VGPRs) serve one-oct strips

// LDS layout of one depth buffer: dr rows of `ls` floats; a row holds the EVEN quads of its octs, E[k+1] = pixels 0..3 of oct k
// (k = -1 .. wo: one ring oct on each side), then the ODD quads O[k+1] = pixels 4..7, each array (wo + 2) quads long.  Thread
// (sy, sx) reads E[sx], O[sx] with ds_read_b128 (consecutive lanes = consecutive 16-byte slots: conflict-free) and the R
// pixels left / right of its oct as the tail of O[sx-1] / the head of E[sx+1].
// GT = the guidance dtype: __half -> taps packed two pixels per register (v_fma_mix_f32); float -> fp32 taps, 8 (K*K-1) registers
// per oct, plain v_fma_f32 (launched for K = 5 only: K = 3 in fp32 — the reference model's own configuration — runs on the quad
// kernel of cspn_resident.hip in its softmax-weight form, see cspnk_resident_plan).
// TRANS = 1: the backward's reverse sweep  G_t = stencil^T((1-m) G_{t+1})  (pac.py:96-121 run T times) as the same recurrence on
// the TRANSPOSED taps: tap j = w_{NT-1-j}[p + off_j], gathered ONCE from the forward's fp16 tap volume (pair-interleaved layout,
// cspn_common.hpp Taps<__half>) instead of being derived by a softmax — what cspn_transpose_kernel + three streaming launches
// did with a second 80 MB volume.  x0 is G_T = dL/dout (fp32), BLEND means PREMASK (the state that travels — LDS, exchange
// planes — is (1-m) G, the history planes receive G itself: history[s] = G_{T-1-s}), fp32 planes, no scoring.
template <int K, int NO, int BLEND, int SCORE, int CLEAN, typename ST, int NTH = KRES_THREADS, typename GT = __half, int TRANS = 0>
__global__ __launch_bounds__(NTH, NTH / 256) void cspnk_resident(const KResArgs a) {
    constexpr int R = K / 2, NT = K * K - 1;
    constexpr bool PK = std::is_same<GT, __half>::value;
    static_assert(!TRANS || (PK && std::is_same<ST, float>::value && !SCORE && NO == 1), "the reverse sweep: packed taps, fp32 planes, one oct per thread");
    static_assert(PK || std::is_same<ST, float>::value, "fp32 guidance runs with fp32 depth planes");
    static_assert(R == 1 || R == 2, "K = 3 or 5");
    using IO = StateIO<ST>;
    using Oct = typename IO::Oct;
    using Pair = typename IO::Pair;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int wg_bad;

    const int tid = threadIdx.x;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    // The reverse sweep takes the whole batch in ONE launch of a.rounds * (nb * tiles) workgroups: the first nb * tiles of them (round 0:
    // images b0 .. b0 + nb - 1, one workgroup per CU) are resident at once, the workgroups of round r + 1 (images b0 + (r+1) nb ...) are
    // dispatched in blockIdx order as those of round r finish — no second launch and no drain between the halves, and nothing carried
    // from one image to the next in registers (a round LOOP inside the kernel, as cspnk_d2 has, costs this kernel 14-33 more spilled
    // VGPRs: 97 us instead of 2 x 45).  Flags and exchange planes are per image: the rounds share nothing.
    const int wg_round = TRANS ? a.nb * tiles_per_img : (int)gridDim.x;
    const int round = TRANS ? (int)blockIdx.x / wg_round : 0;
    const int tile = xcd_contiguous_id(TRANS ? (int)blockIdx.x - round * wg_round : (int)blockIdx.x, wg_round);
    const int bl = tile / tiles_per_img;
    const int trem = tile - bl * tiles_per_img;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int b = a.b0 + round * a.nb + bl;
    const int H = a.H, W = a.W;
    const int y0 = ty * a.th, x0 = tx * a.tw;
    const unsigned HW = (unsigned)(H * W);
    const size_t plane = (size_t)a.B * HW;
    if (tid == 0) wg_bad = 0;
    int n_stamp = 0;
    auto stamp = [&]() { if (a.dbg && tid == 0 && round == 0 && n_stamp < 16) a.dbg[(size_t)blockIdx.x * 16 + n_stamp++] = wall_clock64(); };   // (round 0's workgroups: [nb x tiles][16])
    stamp();
    // (no completion word for the inference launches: their results are checked where the host synchronises — functional.
    // ensure_resident_ok; the reverse sweep reports like cspn3_resident's training forms do: the last workgroup to count itself
    // out stores `seq` to the second host word)
    auto count_out = [&]() {
        if (TRANS && tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.status + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == gridDim.x) {
                __hip_atomic_store(a.status + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.host_err && a.last_chunk) __hip_atomic_store(a.host_err + 1, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };

    if (TRANS && b >= a.B) { count_out(); return; }      // the last round of a ragged batch has fewer images

    // TRANS = 2: the cotangent and the sparse plane arrive as fp16 (the training step on half planes hands them over as they
    // are: no cast kernels); the state is fp32 all the same, and a.out, when set, receives G_T as fp32 for the backward tail
    typedef typename std::conditional<TRANS == 2, __half, ST>::type INT;
    typedef StateIO<INT> IN;
    typedef typename IN::Oct InOct;
    typedef typename IN::Pair InPair;
    const INT* __restrict__ x0b = kuniform_ptr(static_cast<const INT*>(a.x0) + (size_t)b * HW);
    const INT* __restrict__ spb = BLEND ? kuniform_ptr(static_cast<const INT*>(a.sparse) + (size_t)b * HW) : nullptr;

    // ---- ownership: strip (sx, sy) = NO vertically consecutive octs of the weight region (tile + halo) ----------------
    const int wo = a.wo, wr = a.wr;
    const int sy = tid / wo;
    const int sx = tid - sy * wo;
    const int r0 = sy * NO;
    // region origin, shifted back into the image at image edges (cspn3_resident: never before the previous tile's start, so
    // that the halo always comes from the 8 adjacent tiles)
    // (the "previous tile's start" bound belongs to the exchange: a single-phase launch stages everything from the coarse
    // depth, and its halo may be deeper than a tile — there the bound would cut the region short of y0 - hyw)
    const bool exch = a.T > a.S;
    const int rx0 = max(exch ? max(0, x0 - a.tw) : 0, min(x0 - a.hxw, W - 8 * wo));
    const int ry0 = max(exch ? max(0, y0 - a.th) : 0, min(y0 - a.hyw, H - wr));
    const int xo = rx0 + 8 * sx;
    const int yo0 = ry0 + r0;
    const bool x_in = xo < W;                  // W % 8 == 0: an oct lies inside the image or outside as a whole

    // ---- 0. the depth region of phase 0 is requested FIRST (loads return in order: it is parked in LDS while the guidance
    //         stream is still in flight)
    const int dr = a.dr, ls = a.ls;
    const int pp = dr * ls;                    // floats per depth buffer
    float* const cur = lds;                    // buffer 0: where every phase starts (phases have an even number of steps)
    float* const nxt = lds + pp;
    const int yd0 = ry0 - R;
    const int eo = 4 * (wo + 2);               // offset of the odd-quad array inside a row
    InOct st0[NO + 1];                         // dr * wo <= (NO + 1) * NTH octs (the host checks)
    InOct sp0[(TRANS && BLEND) ? NO + 1 : 1];
    InPair spr[2];
    unsigned st0_in = 0;
    const int step_r = NTH / wo, step_q = NTH - step_r * wo;
    {
        int row = sy, oc = sx;
#pragma unroll
        for (int u = 0; u <= NO; ++u) {
            const int y = yd0 + row, x = rx0 + 8 * oc;
            const bool in = (row < dr) && y >= 0 && y < H && x < W;
            if (in) st0_in |= 1u << u;
            st0[u] = IN::ld_oct(x0b, in ? (unsigned)(y * W + x) : 0u);
            if (TRANS && BLEND) sp0[u] = IN::ld_oct(spb, in ? (unsigned)(y * W + x) : 0u);      // PREMASK: the sweep starts from (1-m) G_T
            row += step_r; oc += step_q;
            if (oc >= wo) { oc -= wo; ++row; }
        }
    }
    // the R-pixel ring left / right of the region rows: one pair (R = 2) or one pixel (R = 1, read as the pair it sits in)
    // per row and side; 2 * dr <= 2 * NTH items
    InPair rg[2];
    unsigned rg_in = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int t = tid + k * NTH;
        const int row = t >> 1, side = t & 1;
        const int y = yd0 + row, x = side ? rx0 + 8 * wo : rx0 - 2;
        const bool in = (t < 2 * dr) && y >= 0 && y < H && x >= 0 && x < W;
        if (in) rg_in |= 1u << k;
        rg[k] = IN::ld_pair(x0b, in ? (unsigned)(y * W + x) : 0u);
        if (TRANS && BLEND) spr[k] = IN::ld_pair(spb, in ? (unsigned)(y * W + x) : 0u);
    }

    // ---- 1. guidance of the owned octs: NT 16-byte loads per oct, all requested before the arithmetic ------------------
    uint4 wpk[PK ? NO : 1][PK ? NT : 1];       // packed fp16 taps ...
    float wf[PK ? 1 : NO][PK ? 1 : NT][8];     // ... or fp32 taps
    unsigned in_img = 0, interior = 0;
    const GT* __restrict__ gb = kuniform_ptr(static_cast<const GT*>(a.g) + (size_t)b * NT * HW);
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        const int y = yo0 + i;
        const bool ok = (r0 + i < wr) && x_in && y < H;
        if (ok) in_img |= 1u << i;
        if (ok && y >= y0 && y < y0 + a.th && xo >= x0 && xo < x0 + a.tw) interior |= 1u << i;
        const unsigned off = ok ? (unsigned)(y * W + xo) : 0u;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            if constexpr (TRANS) {
                // transposed tap j = c: channel NT-1-j of the forward volume at p + off_j.  The volume keeps tap pairs per quad
                // ([NT/2][HW/4][2][4] halfs): the 8 pixels x+dx .. x+dx+7 of a tap are pieces of two or three quads (8 bytes
                // each), put together by dword selects (dx even) or 16-bit funnel shifts (dx odd).  Branch-free: safe addresses
                // + selects, all requested before they are used.
                const int lin = c < NT / 2 ? c : c + 1;
                const int dy = lin / K - R, dx = lin % K - R;
                const int cs = NT - 1 - c;
                const int ys = y + dy;
                const bool rok = ok && ys >= 0 && ys < H;
                const unsigned pq = rok ? ((unsigned)(ys * W + xo) >> 2) : 0u;
                const unsigned cbase = (unsigned)(cs >> 1) * 2u * HW + ((unsigned)(cs & 1) << 2);      // halfs (HW % 8 == 0: hw4 = HW)
                const bool lok = rok && xo > 0, r2ok = rok && xo + 8 < W;
                uint2 q0 = ld8u(atb(gb, (cbase + (pq << 3)) * 2u));
                uint2 q1 = ld8u(atb(gb, (cbase + ((pq + 1u) << 3)) * 2u));
                uint2 qm = make_uint2(0u, 0u), q2 = make_uint2(0u, 0u);
                if (dx < 0) qm = ld8u(atb(gb, (cbase + ((lok ? pq - 1u : pq) << 3)) * 2u));
                if (dx > 0) q2 = ld8u(atb(gb, (cbase + ((r2ok ? pq + 2u : pq) << 3)) * 2u));
                if (!rok) { q0 = make_uint2(0u, 0u); q1 = make_uint2(0u, 0u); }
                if (!lok) qm = make_uint2(0u, 0u);
                if (!r2ok) q2 = make_uint2(0u, 0u);
                auto fs = [](unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); };      // (lo.hi, hi.lo)
                uint4 t;
                if (dx == -2) t = make_uint4(qm.y, q0.x, q0.y, q1.x);
                else if (dx == -1) t = make_uint4(fs(q0.x, qm.y), fs(q0.y, q0.x), fs(q1.x, q0.y), fs(q1.y, q1.x));
                else if (dx == 0) t = make_uint4(q0.x, q0.y, q1.x, q1.y);
                else if (dx == 1) t = make_uint4(fs(q0.y, q0.x), fs(q1.x, q0.y), fs(q1.y, q1.x), fs(q2.x, q1.y));
                else t = make_uint4(q0.y, q1.x, q1.y, q2.x);
                wpk[PK ? i : 0][PK ? c : 0] = t;
            } else if constexpr (PK) {
                wpk[i][c] = ld16(atb(gb, ((unsigned)c * HW + off) * 2u));
            } else {
                const uint4 lo = ld16(atb(gb, ((unsigned)c * HW + off) * 4u)), hi = ld16(atb(gb, ((unsigned)c * HW + off) * 4u + 16u));
                wf[i][c][0] = __uint_as_float(lo.x); wf[i][c][1] = __uint_as_float(lo.y); wf[i][c][2] = __uint_as_float(lo.z); wf[i][c][3] = __uint_as_float(lo.w);
                wf[i][c][4] = __uint_as_float(hi.x); wf[i][c][5] = __uint_as_float(hi.y); wf[i][c][6] = __uint_as_float(hi.z); wf[i][c][7] = __uint_as_float(hi.w);
            }
        }
    }

    // ---- park the depth region (its loads came first: only those are waited for here) ---------------------------------
    {
        int tidk = tid, prow = sy, poc = sx;
        asm volatile("" : "+v"(tidk), "+v"(prow), "+v"(poc));
#pragma unroll
        for (int u = 0; u <= NO; ++u) {
            float v[8];
            IN::to_f8(st0[u], v);
            const bool in = (st0_in >> u) & 1u;
            if constexpr (TRANS == 2) {
                // the tile's own octs of G_T, as fp32, for the tail (before the mask)
                const int y = yd0 + prow, x = rx0 + 8 * poc;
                if (a.out && in && prow < dr && y >= y0 && y < y0 + a.th && x >= x0 && x < x0 + a.tw)
                    IO::st_oct(kuniform_ptr(static_cast<ST*>(a.out) + (size_t)b * HW), (unsigned)(y * W + x), IO::from_f8(v));
            }
            if (TRANS && BLEND) {
                float m8[8];
                IN::to_f8(sp0[(TRANS && BLEND) ? u : 0], m8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= 1.f - sgnf(m8[e]);
            }
            if (prow < dr) {
                float* p = cur + prow * ls + 4 * (poc + 1);
                *reinterpret_cast<float4*>(p) = make_float4(in ? v[0] : 0.f, in ? v[1] : 0.f, in ? v[2] : 0.f, in ? v[3] : 0.f);
                *reinterpret_cast<float4*>(p + eo) = make_float4(in ? v[4] : 0.f, in ? v[5] : 0.f, in ? v[6] : 0.f, in ? v[7] : 0.f);
            }
            prow += step_r; poc += step_q;
            if (poc >= wo) { poc -= wo; ++prow; }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int t = tidk + k * NTH;
            if (t < 2 * dr) {
                const int row = t >> 1, side = t & 1;
                float p0, p1;
                IN::to_f2(rg[k], p0, p1);
                if (TRANS && BLEND) {
                    float m0, m1;
                    IN::to_f2(spr[k], m0, m1);
                    p0 *= 1.f - sgnf(m0); p1 *= 1.f - sgnf(m1);
                }
                const bool in = (rg_in >> k) & 1u;
                const int at = row * ls + (side ? 4 * (wo + 1) : eo + 2);     // E[wo].xy  /  O[-1].zw
                cur[at] = in ? p0 : 0.f; cur[at + 1] = in ? p1 : 0.f;
                nxt[at] = 0.f; nxt[at + 1] = 0.f;       // the ring of the second buffer is never computed: it must read as 0
            }
        }
        // ... and so must its ring ROWS (with shifted regions they are the zero padding above / below the image)
        for (int c = tidk; c < 2 * R * ls; c += NTH) {
            const int rr = c / ls, col = c - rr * ls;
            nxt[(rr < R ? rr : dr - 2 * R + rr) * ls + col] = 0.f;
        }
    }

    stamp();                                   // depth region parked (the first guidance loads have arrived)
    // ---- 2. softmax over the NT channels of every owned pixel, in place: raw halfs -> fp16 weights ----------------------
    // (CSPN_ours.py:35; the arithmetic of cspn_pac_prepare_vec_kernel: max, softmax_exp<__half>, sum in channel order,
    // one refined reciprocal, round to nearest even.)  One pixel at a time: 24 temporaries next to the 96 * NO tap registers.
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        if constexpr (TRANS) {
            // (the gathered taps are final)
        } else if constexpr (PK) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned w[NT];                               // the pixel pair q of every channel
#pragma unroll
                for (int c = 0; c < NT; ++c) w[c] = comp(wpk[i][c], q);
                softmax_pair<NT>(w);
#pragma unroll
                for (int c = 0; c < NT; ++c) set_comp(wpk[i][c], q, w[c]);
            }
        } else {
            // fp32: the arithmetic of cspn_pac_prepare_kernel<K, float, float> (two-piece exponential, refined reciprocal)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float mx = -INFINITY;
#pragma unroll
                for (int c = 0; c < NT; ++c) mx = fmaxf(mx, wf[i][c][e]);
                float den = 0.f;
#pragma unroll
                for (int c = 0; c < NT; ++c) { wf[i][c][e] = softmax_exp<float>(wf[i][c][e] - mx); den += wf[i][c][e]; }
                const float inv = reciprocal_refined(den);
#pragma unroll
                for (int c = 0; c < NT; ++c) wf[i][c][e] = softmax_weight<float>(wf[i][c][e], inv);
            }
        }
    }

    // ---- sparse blend: (1-m) u + m x0 with m = sign(sparse) (CSPN_ours.py:51-53).  1-m is 0, 1 or 2, so it is FOLDED into the
    // taps of the owned octs once (exact: the scaled taps and every product are the unscaled ones times a power of two or
    // zero, so sum_j ((1-m) w_j) x_j == (1-m) sum_j w_j x_j bit for bit — what cspn_prop_fused's FOLD does for fp32 taps);
    // the steps then only add md = m * x0, kept in private LDS slots.
    float* const md_lds = lds + 2 * pp;
    if (BLEND) {
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const bool ok = (in_img >> i) & 1u;
            const unsigned off = ok ? (unsigned)((yo0 + i) * W + xo) : 0u;
            float sp[8], dv[8];
            IN::to_f8(IN::ld_oct(spb, off), sp);
            IN::to_f8(IN::ld_oct(x0b, off), dv);
            float om[8], md[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float m = ok ? sgnf(sp[e]) : 0.f;
                om[e] = 1.f - m;
                md[e] = TRANS ? om[e] : m * (ok ? dv[e] : 0.f);      // PREMASK: the private slots hold 1-m, applied to every step's result
            }
            if constexpr (TRANS) {
                // (nothing is folded into the taps: the history receives the unmasked G)
            } else if constexpr (PK) {
                unsigned omp[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) omp[q] = pack_h2(om[2 * q], om[2 * q + 1]);
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) set_comp(wpk[PK ? i : 0][PK ? c : 0], q, pk_mul_f16(comp(wpk[PK ? i : 0][PK ? c : 0], q), omp[q]));
            } else {
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) wf[PK ? 0 : i][PK ? 0 : c][e] *= om[e];
            }
            if (r0 + i < wr) {
                float* pm = md_lds + ((r0 + i) * wo + sx) * 8;
                *reinterpret_cast<float4*>(pm) = make_float4(md[0], md[1], md[2], md[3]);
                *reinterpret_cast<float4*>(pm + 4) = make_float4(md[4], md[5], md[6], md[7]);
            }
        }
    }

    stamp();                                   // weights derived
    // ---- 3. phases of S steps; between phases the tile borders travel through the exchange planes ----------------------
    const bool active = r0 < wr;
    ST* __restrict__ outb = TRANS ? nullptr : kuniform_ptr(static_cast<ST*>(a.out) + (size_t)b * HW);      // (TRANS: a.out is the G_T copy)
    ST* hist_step = TRANS ? kuniform_ptr(static_cast<ST*>(a.hist) + (size_t)b * HW) : nullptr;      // plane of the step being computed
    const int n_phase = (a.T + a.S - 1) / a.S;
    const int tile_global = b * tiles_per_img + trem;
    int fin_buf = 0;                           // LDS buffer that receives the final step's (stored) values: the fused metrics read them back

    for (int p = 0; p < n_phase; ++p) {
        const int steps = (a.T - p * a.S) < a.S ? (a.T - p * a.S) : a.S;
        const bool last_phase = (p == n_phase - 1);
        ST* __restrict__ xout = kuniform_ptr(static_cast<ST*>(a.xbuf) + (size_t)(p & 1) * plane + (size_t)b * HW);
        if (p > 0) {
            // halo octs only (the tile's own interior is in LDS already): full rows above and below the tile rows, the octs
            // left and right of the tile columns; device-scope loads, two per thread and trip, requested before they are used
            const ST* __restrict__ xin = kuniform_ptr(static_cast<const ST*>(a.xbuf) + (size_t)((p + 1) & 1) * plane + (size_t)b * HW);
            const int tq_in = min(a.tw, rx0 + 8 * wo - x0) >> 3;
            const int th_in = min(a.th, ry0 + wr - y0);
            const int nl = (x0 - rx0) >> 3;
            const int nside = wo - tq_in;
            const int nrow_t = (y0 - ry0) + R;
            const int n_top = nrow_t * wo;
            const int n_bot = (dr - nrow_t - th_in) * wo;
            const int n_halo = n_top + n_bot + th_in * nside;
            Pair rgp[2];
            unsigned rgp_in = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = tid + k * NTH;
                const int row = t >> 1, side = t & 1;
                const int y = yd0 + row, x = side ? rx0 + 8 * wo : rx0 - 2;
                const bool in = (t < 2 * dr) && y >= 0 && y < H && x >= 0 && x < W;
                if (in) rgp_in |= 1u << k;
                rgp[k] = IO::ld_pair_dev(xin, in ? (unsigned)(y * W + x) : 0u);
            }
            for (int base = 0; base < n_halo; base += 2 * NTH) {
                Oct hv[2];
                int at[2];
                bool hin[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int h = base + u * NTH + tid;
                    int row, oc;
                    if (h < n_top) { row = h / wo; oc = h - row * wo; }
                    else if (h < n_top + n_bot) { const int h2 = h - n_top; row = h2 / wo; oc = h2 - row * wo; row += nrow_t + th_in; }
                    else {
                        const int h3 = h - n_top - n_bot;
                        const int ns = nside > 0 ? nside : 1;
                        row = h3 / ns;
                        const int c = h3 - row * ns;
                        row += nrow_t;
                        oc = c < nl ? c : c + tq_in;
                    }
                    const int y = yd0 + row, x = rx0 + 8 * oc;
                    const bool valid = h < n_halo;
                    hin[u] = valid && y >= 0 && y < H && x < W;
                    hv[u] = IO::ld_oct_dev(xin, hin[u] ? (unsigned)(y * W + x) : 0u);
                    at[u] = valid ? row * ls + 4 * (oc + 1) : -1;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (at[u] >= 0) {
                        float v[8];
                        IO::to_f8(hv[u], v);
                        const bool in = hin[u];
                        *reinterpret_cast<float4*>(cur + at[u]) = make_float4(in ? v[0] : 0.f, in ? v[1] : 0.f, in ? v[2] : 0.f, in ? v[3] : 0.f);
                        *reinterpret_cast<float4*>(cur + at[u] + eo) = make_float4(in ? v[4] : 0.f, in ? v[5] : 0.f, in ? v[6] : 0.f, in ? v[7] : 0.f);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = tid + k * NTH;
                if (t < 2 * dr) {
                    const int row = t >> 1, side = t & 1;
                    float p0, p1;
                    IO::to_f2(rgp[k], p0, p1);
                    const bool in = (rgp_in >> k) & 1u;
                    const int at = row * ls + (side ? 4 * (wo + 1) : eo + 2);
                    cur[at] = in ? p0 : 0.f; cur[at + 1] = in ? p1 : 0.f;
                }
            }
        }
        __syncthreads();

        // One propagation step on the LDS tile: kind 0 = plain, 1 = last step of a phase (the interior is published, the
        // state rounded to the plane dtype first), 2 = the final step of the forward (refined depth stored, kept for scoring).
        auto step = [&](const int kind, const float* rd, float* wrb) __attribute__((always_inline)) {
            float acc[NO][8];
#pragma unroll
            for (int i = 0; i < NO; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
#pragma unroll
            for (int rr = 0; rr < NO + 2 * R; ++rr) {
                int drow = r0 + rr;                      // depth-region row of window row rr; rows past the region (idle
                drow = drow < dr ? drow : dr - 1;        // octs of the last strip) read the clamped last row, never stored
                const float* rowp = rd + drow * ls + 4 * (sx + 1);
                float x[8 + 2 * R];
                const v4f e4 = *(lds_cv4f_ptr)(rowp);
                const v4f o4 = *(lds_cv4f_ptr)(rowp + eo);
                x[R + 0] = e4.x; x[R + 1] = e4.y; x[R + 2] = e4.z; x[R + 3] = e4.w;
                x[R + 4] = o4.x; x[R + 5] = o4.y; x[R + 6] = o4.z; x[R + 7] = o4.w;
                if constexpr (R == 2) {
                    const v2f l2 = *(lds_cv2f_ptr)(rowp + eo - 2);          // O[sx-1].zw
                    const v2f r2 = *(lds_cv2f_ptr)(rowp + 4);               // E[sx+1].xy
                    x[0] = l2.x; x[1] = l2.y; x[10] = r2.x; x[11] = r2.y;
                } else {
                    x[0] = *(lds_cf_ptr)(rowp + eo - 1);                    // O[sx-1].w
                    x[9] = *(lds_cf_ptr)(rowp + 4);                         // E[sx+1].x
                }
#pragma unroll
                for (int i = 0; i < NO; ++i) {
                    const int dy = rr - R - i;
                    if (dy < -R || dy > R) continue;
#pragma unroll
                    for (int dx = -R; dx <= R; ++dx) {
                        if (dy == 0 && dx == 0) continue;
                        const int lin = (dy + R) * K + (dx + R);
                        const int j = lin < (K * K) / 2 ? lin : lin - 1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if constexpr (PK) acc[i][e] = fma_h8(wpk[PK ? i : 0][PK ? j : 0], e, x[R + e + dx], acc[i][e]);
                            else acc[i][e] = fmaf(wf[PK ? 0 : i][PK ? 0 : j][e], x[R + e + dx], acc[i][e]);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = acc[i][e];
                if (!CLEAN) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = ((in_img >> i) & 1u) ? u[e] : 0.f;   // zero padding stays exactly zero
                }
                const bool inner = (interior >> i) & 1u;
                const unsigned off = (unsigned)((yo0 + i) * W + xo);
                if (TRANS && inner) IO::st_oct_hist(hist_step, off, IO::from_f8(u));            // G_t itself goes to its history plane
                if (BLEND) {
                    const int q = (min(r0 + i, wr - 1) * wo + sx) * 8;
                    const float4 ma = *reinterpret_cast<const float4*>(md_lds + q), mb = *reinterpret_cast<const float4*>(md_lds + q + 4);
                    const float md[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = TRANS ? md[e] * u[e] : u[e] + md[e];      // forward: (1-m) lives in the taps, + m x0; sweep: (1-m) G travels
                }
                if (kind != 0) {                         // the state leaves the launch in the plane dtype
                    const Oct o = IO::from_f8(u);
                    if (sizeof(ST) == 2) IO::to_f8(o, u);
                    if (kind == 1) { if (inner) IO::st_oct_dev(xout, off, o); }
                    else if (inner && !TRANS) IO::st_oct(outb, off, o);
                }
                // (the final step writes LDS too when the metrics are fused: they score the stored values, read back from the
                // thread's own slots after the loop — sixteen registers carried out of the step cost the hot loop its spill-free form)
                if ((kind != 2 || SCORE) && r0 + i < wr) {
                    float* p = wrb + (r0 + i + R) * ls + 4 * (sx + 1);
                    *reinterpret_cast<float4*>(p) = make_float4(u[0], u[1], u[2], u[3]);
                    *reinterpret_cast<float4*>(p + eo) = make_float4(u[4], u[5], u[6], u[7]);
                }
            }
        };
        stamp();                               // depth staged
        const bool any = __ballot(active) != 0ull;          // wavefronts without a single owned row only keep the barriers company
        for (int s = 0; s < steps; ++s) {
            const int kind = (s == steps - 1) ? (last_phase ? 2 : 1) : 0;
            if (any) step(kind, lds + (s & 1) * pp, lds + ((s + 1) & 1) * pp);
            if (TRANS) hist_step += plane;
            if (kind == 2) fin_buf = (s + 1) & 1;
            if (kind == 0) __syncthreads();
        }
        stamp();                               // steps of the phase done
        if (!last_phase) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's device-scope stores have landed
            __syncthreads();                                       // ... and so have everybody else's in the workgroup
            const unsigned want = a.seq + (unsigned)p + 1u;
            if (tid == 0) __hip_atomic_store(a.flags + tile_global, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 9 && tid != 4) {
                const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                if (ny >= 0 && ny < a.tiles_y && nx >= 0 && nx < a.tiles_x) {
                    const unsigned* f = a.flags + b * tiles_per_img + ny * a.tiles_x + nx;
                    unsigned spins = 0;
                    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
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
                    __atomic_thread_fence(__ATOMIC_SEQ_CST);
                }
                // poison what this tile will never produce (NaN in the plane dtype), as cspn3_resident does
                // (bit patterns, not conversions: the payload is what the host recognises a failed tile by)
                Oct o;
                if constexpr (sizeof(ST) == 2) {
                    constexpr unsigned pz2 = CSPN_POISON_F16 | (CSPN_POISON_F16 << 16);
                    o.a = make_uint4(pz2, pz2, pz2, pz2);
                } else {
                    o.a = make_uint4(CSPN_POISON_F32, CSPN_POISON_F32, CSPN_POISON_F32, CSPN_POISON_F32);
                    o.b = o.a;
                }
                ST* const pz = TRANS ? kuniform_ptr(static_cast<ST*>(a.hist) + (size_t)(a.T - 1) * plane + (size_t)b * HW) : outb;   // G_0 / the refined depth
#pragma unroll
                for (int i = 0; i < NO; ++i)
                    if ((interior >> i) & 1u) IO::st_oct(pz, (unsigned)((yo0 + i) * W + xo), o);
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
            const ST* tgt_b = kuniform_ptr(static_cast<const ST*>(a.target) + (size_t)b * HW);
            Oct tq[NO];
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const bool in = (interior >> i) & 1u;
                tq[i] = IO::ld_oct(tgt_b, in ? (unsigned)((yo0 + i) * W + xo) : 0u);
            }
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                if ((interior >> i) & 1u) {
                    float t8[8];
                    IO::to_f8(tq[i], t8);
                    const float* p = lds + fin_buf * pp + (r0 + i + R) * ls + 4 * (sx + 1);
                    const float4 fa = *reinterpret_cast<const float4*>(p), fb = *reinterpret_cast<const float4*>(p + eo);
                    const float f8[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) metric_terms(f8[e], t8[e], mf);
                }
            }
        }
        float* part = lds + 2 * pp + (BLEND ? 1 : 0) * wr * wo * 8;
        const int wave = tid >> 6, lane = tid & 63;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float v = wave_sum_to_lane63(mf[k]);
            if (lane == 63) part[wave * 10 + k] = v;
        }
        __syncthreads();
        if (tid < 10) {
            double v = 0.0;
            for (int w = 0; w < NTH / 64; ++w) v += (double)part[w * 10 + tid];
            if (v != 0.0) atomicAdd(a.macc + (size_t)(blockIdx.x % a.nslots) * 10 + tid, v);
        }
    }
    stamp();                                   // epilogue done
    count_out();
}

// ------------------------------------------------------------------------------------------------ host side
constexpr int KRES_MAX_NO_K5 = 2;
constexpr int KRES_T_MAX_ROUNDS = 8;    // rounds of the reverse sweep in one launch
#ifndef CSPN_KT_ONE_LAUNCH
#define CSPN_KT_ONE_LAUNCH 1
#endif
constexpr int KRES_MAX_NO_K3 = 4;

struct KGeom {
    int S, tiles_x, tiles_y, tw, th, no, wo, wr, hxw, hyw, dr, ls, threads;
    int imgs_per_launch, launches;
    size_t lds_bytes;
    double cost;
};

int kcu_count() {
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

inline int round_up8(int a) { return (a + 7) & ~7; }

// Row stride of a depth buffer: at least the (wo + 2) even and (wo + 2) odd quads, and such that NO * ls = 4 * wo (mod 64
// dwords) where possible — the 16-byte slot a lane reads is then 4 * tid + const (mod 64): strips stacked in a wavefront
// continue the bank pattern of the strip before them.
int kres_row_stride(int wo, int no) {
    const int lo = 8 * (wo + 2);
    int best = lo, best_score = 1 << 30;
    for (int cand = lo; cand < lo + 64; cand += 4) {
        const int score = (((no * cand - 4 * wo) % 64) + 64) % 64;
        if (score < best_score) { best_score = score; best = cand; }
    }
    return best;
}

size_t kres_lds_bytes(int dr, int ls, int wr, int wo, int blend) {
    return ((size_t)2 * dr * ls + (size_t)(blend ? 1 : 0) * wr * wo * 8 + 16 * 10) * sizeof(float);
}

bool kregions_inside_image(const KGeom& g, int H, int W, int T) {
    const bool exch = T > g.S;            // mirrors the kernel: see the region origin there
    for (int tx = 0; tx < g.tiles_x; ++tx) {
        const int x0 = tx * g.tw;
        int rx0 = x0 - g.hxw; if (rx0 > W - 8 * g.wo) rx0 = W - 8 * g.wo;
        int lo = exch ? x0 - g.tw : 0; if (lo < 0) lo = 0;
        if (rx0 < lo) rx0 = lo;
        if (rx0 + 8 * g.wo > W) return false;
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

// threads: 512 (two wavefronts per SIMD, 256 VGPRs) or 768 (three per SIMD, 168 VGPRs: one oct per thread at K = 5, two at
// K = 3).  A step costs a SIMD (octs per thread) x (its wavefronts that own any): a 735-oct region is 4 such units on 512
// threads (6 of 8 wavefronts busy, two octs each) and 3 on 768 (12 wavefronts, one oct each).
bool kgeom_fill(int K, int gdt, int H, int W, int T, int blend, int ncu, int B, int Se, int tx, int ty, int tw, int th, int threads, KGeom* g) {
    const int R = K / 2;
    if (threads != 512 && threads != 768) return false;
    int max_no = threads == 768 ? (K == 5 ? 1 : 2) : (K == 5 ? KRES_MAX_NO_K5 : KRES_MAX_NO_K3);
    if (gdt == CSPN_F32) max_no = (K == 5 && threads == 512) ? 1 : 0;       // fp32 taps: 192 registers per oct at K = 5 (K = 3: quad kernel)
    const int hyw = (Se - 1) * R, hxw = round_up8((Se - 1) * R);
    const int phases = ceil_div(T, Se);
    if (phases > 1 && (Se & 1)) return false;          // every phase must start in buffer 0
    if (phases > 255) return false;
    if ((tw & 7) || tw < 8 || th < 1 || tx * tw < W || ty * th < H) return false;
    if (phases > 1 && ((tx > 1 && tw < 2 * hxw) || (ty > 1 && th < 2 * hyw))) return false;   // halo from adjacent tiles only
    const int wo = (tw + 2 * hxw) / 8;
    if (wo > 128 || wo < 1) return false;
    const int wr = th + 2 * hyw;
    if (wo > threads) return false;
    const int no = ceil_div(wr, threads / wo);
    if (no > max_no) return false;
    const int dr = wr + 2 * R;
    if (2 * dr > 2 * threads || dr * wo > (no + 1) * threads) return false;
    const int ls = kres_row_stride(wo, no);
    const size_t ldsb = kres_lds_bytes(dr, ls, wr, wo, blend);
    if (ldsb > 160 * 1024) return false;
    const int tiles = tx * ty;
    if (tiles > ncu) return false;
    int ipl = ncu / tiles;
    if (ipl > B) ipl = B;
    const int launches = ceil_div(B, ipl);
    *g = KGeom{Se, tx, ty, tw, th, no, wo, wr, hxw, hyw, dr, ls, threads, ipl, launches, ldsb, 0.0};
    // microseconds per launch, fitted on MI355X at config 3 (tools/probes/kres_probe.py plans -> profiles/r03_kres_plans.txt):
    // launch + epilogue 3; the derive (softmax: 3.3 per unit) and every step (0.5 per unit) cost VALU time in proportion to
    // unit = (octs per thread) x (wavefronts per SIMD that own any) — v_fma_mix_f32 issues once per 4 cycles and wavefront;
    // each phase boundary a publish / wait / halo staging round trip of ~3.2.  Measured / modelled per launch: S=4 768 threads
    // 39.4 / 38.3, S=6 512 threads 46.5 / 44.4, S=2 768 threads 47.1 / 47.9, S=6 768 threads (3 launches) 35.3 / 35.0.
    const int strips = ceil_div(wr, no) * wo;
    const int waves_per_simd = ceil_div(ceil_div(strips, 64), 4);
    const double per_oct = (double)no * waves_per_simd;
    const double taps = (double)(K * K - 1) / 24.0 * (gdt == CSPN_F32 ? 0.8 : 1.0);
    const double pen = kregions_inside_image(*g, H, W, T) ? 1.0 : 1.1;
    g->cost = launches * (3.0 + 3.3 * taps * per_oct + T * (0.5 * taps * per_oct * pen + 0.08) + (phases - 1) * 3.2);
    return true;
}

bool kres_geometry(int K, int gdt, int B, int H, int W, int T, int blend, int ncu, int S_user, int threads_user, KGeom* best) {
    if (W % 8 != 0 || ncu < 1 || T < 1 || (K != 3 && K != 5)) return false;
    bool found = false;
    const int s_hi = S_user > 0 ? S_user : (K == 3 ? 8 : 6), s_lo = S_user > 0 ? S_user : 2;
    for (int S = s_hi; S >= s_lo; S -= (S_user > 0 ? 1 : 2)) {
        const int Se = S > T ? T : S;
        for (int tx = 1; tx <= 32; ++tx) {
            const int tw = round_up8(ceil_div(W, tx));
            if (tx > 1 && (tw < 16 || ceil_div(W, tw) != tx)) continue;
            for (int ty = 1; ty <= 64; ++ty) {
                const int th = ceil_div(H, ty);
                if (ty > 1 && (th < 4 || ceil_div(H, th) != ty)) continue;
                for (int threads = 512; threads <= 768; threads += 256) {
                    if (threads_user > 0 && threads != threads_user) continue;
                    KGeom cand;
                    if (!kgeom_fill(K, gdt, H, W, T, blend, ncu, B, Se, tx, ty, tw, th, threads, &cand)) continue;
                    if (!found || cand.cost < best->cost) { found = true; *best = cand; }
                }
            }
        }
    }
    return found;
}

template <int K, int NO, int BLEND, int SCORE, int CLEAN, typename ST, int NTH, typename GT, int TRANS = 0>
int klaunch_inst(const KResArgs& a, int grid, size_t lds_bytes, hipStream_t st) {
    constexpr auto kern = cspnk_resident<K, NO, BLEND, SCORE, CLEAN, ST, NTH, GT, TRANS>;
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
template <int K, int NO, typename ST, int NTH, typename GT = __half>
int klaunch_no(const KResArgs& a, int grid, size_t lds, int blend, int score, bool clean, hipStream_t st) {
#define KRES_CASE(BL, SC, CL) \
    if (blend == BL && score == SC && (int)clean == CL) return klaunch_inst<K, NO, BL, SC, CL, ST, NTH, GT>(a, grid, lds, st)
    KRES_CASE(0, 0, 0); KRES_CASE(0, 0, 1); KRES_CASE(0, 1, 0); KRES_CASE(0, 1, 1);
    KRES_CASE(1, 0, 0); KRES_CASE(1, 0, 1); KRES_CASE(1, 1, 0); KRES_CASE(1, 1, 1);
#undef KRES_CASE
    return fail("cspnk_forward_resident: internal dispatch");
}
template <int K, typename ST>
int klaunch_k(const KResArgs& a, int no, int threads, int grid, size_t lds, int blend, int score, bool clean, hipStream_t st) {
    if (threads == 768) {
        if (no == 1) return klaunch_no<K, 1, ST, 768>(a, grid, lds, blend, score, clean, st);
        if constexpr (K == 3) {
            if (no == 2) return klaunch_no<K, 2, ST, 768>(a, grid, lds, blend, score, clean, st);
        }
        return fail("cspnk_forward_resident: no 768-thread instance for K=%d with %d octs per thread", K, no);
    }
    switch (no) {
        case 1: return klaunch_no<K, 1, ST, 512>(a, grid, lds, blend, score, clean, st);
        case 2: return klaunch_no<K, 2, ST, 512>(a, grid, lds, blend, score, clean, st);
        default: break;
    }
    if constexpr (K == 3) {
        if (no == 3) return klaunch_no<K, 3, ST, 512>(a, grid, lds, blend, score, clean, st);
        if (no == 4) return klaunch_no<K, 4, ST, 512>(a, grid, lds, blend, score, clean, st);
    }
    return fail("cspnk_forward_resident: no instance for K=%d with %d octs per thread", K, no);
}

// fp32 guidance (fp32 depth planes) on the oct kernel: K = 5, one oct per thread (192 tap registers); K = 3 goes to the quad kernel
template <int K>
int klaunch_f32(const KResArgs& a, int no, int threads, int grid, size_t lds, int blend, int score, bool clean, hipStream_t st) {
    if (threads == 512 && no == 1) return klaunch_no<K, 1, float, 512, float>(a, grid, lds, blend, score, clean, st);
    return fail("cspnk_forward_resident: no fp32-guidance instance for K=%d with %d octs per thread on %d threads", K, no, threads);
}

}  // namespace

extern "C" {

int cspnk_resident_plan(int K, int g_dtype, int B, int H, int W, int T, int blend, int n_cu, cspn_resident_plan* out) {
    if (!out || B < 1 || H < 1 || W < 1 || T < 0) return fail("cspnk_resident_plan: bad arguments");
    if (g_dtype != CSPN_F16 && g_dtype != CSPN_F32) return fail("cspnk_resident_plan: guidance dtype %d", g_dtype);
    if (K != 3 && K != 5) return fail("cspnk_resident_plan: K=%d (3 or 5)", K);
    if (n_cu <= 0) n_cu = kcu_count();
    if (n_cu <= 0) return fail("cspnk_resident_plan: no device (pass n_cu > 0 to plan without one)");
    if (K == 3 && g_dtype == CSPN_F32) {
        // fp32 taps at K = 3 (the unet_ours configuration): the quad kernel of cspn_resident.hip in its softmax-weight form —
        // five quads of taps per thread, all 24 frames of a 228 x 304 batch in one launch; the oct kernel held at most three
        // octs of fp32 taps, with spills (65-78 us against 46).  `threads` is ignored (that kernel has 512).
        out->threads = 0;
        return cspn3_resident_plan(B, H, W, T, blend, n_cu, out);
    }
    KGeom g;
    if (T < 1 || !kres_geometry(K, g_dtype, B, H, W, T, blend, n_cu, out->steps_per_phase, out->threads, &g))
        return fail("cspnk_resident_plan: no resident tiling for K=%d B=%d %dx%d T=%d on %d CUs (W %% 8 == 0 needed)", K, B, H, W, T, n_cu);
    out->steps_per_phase = g.S; out->tiles_x = g.tiles_x; out->tiles_y = g.tiles_y; out->tile_w = g.tw; out->tile_h = g.th;
    out->quads_per_thread = g.no; out->threads = g.threads; out->images_per_launch = g.imgs_per_launch;
    out->launches = g.launches; out->lds_bytes = (int)g.lds_bytes; out->n_cu = n_cu;
    out->region_over_tile = (float)((double)(8 * g.wo) * g.wr / ((double)g.tw * g.th));
    return 1;
}

size_t cspnk_resident_workspace_bytes(int B, int H, int W, int state_dtype) {
    const size_t planes = (size_t)2 * B * H * W * esize(state_dtype);
    const size_t flags = ((size_t)B * (((size_t)H * W) / 8 + 1) + 4) * sizeof(unsigned);       // tiles are >= 8 x 1
    return ((planes + 15) & ~(size_t)15) + ((flags + 15) & ~(size_t)15);
}

int cspnk_forward_resident(const void* guided, int g_dtype, int K, const void* x0, const void* sparse, void* out, int state_dtype,
                           void* work, unsigned seq, unsigned* host_err, int B, int H, int W, int T, int blend,
                           const void* target, double* acc, int nslots, const cspn_resident_plan* plan, cspn_stream_t stream) {
    if (!guided || !x0 || !out || !work || B <= 0 || H <= 0 || W <= 0 || T < 1) return fail("cspnk_forward_resident: bad arguments");
    if (K != 3 && K != 5) return fail("cspnk_forward_resident: K=%d (3 or 5)", K);
    if (state_dtype != CSPN_F16 && state_dtype != CSPN_F32) return fail("cspnk_forward_resident: state dtype %d", state_dtype);
    if (g_dtype != CSPN_F16 && g_dtype != CSPN_F32) return fail("cspnk_forward_resident: guidance dtype %d", g_dtype);
    if (g_dtype == CSPN_F32 && state_dtype != CSPN_F32) return fail("cspnk_forward_resident: fp32 guidance runs with fp32 depth planes");
    if (blend != CSPN_BLEND_NONE && blend != CSPN_BLEND_SPARSE) return fail("cspnk_forward_resident: blend %d", blend);
    if (blend && !sparse) return fail("cspnk_forward_resident: blend needs sparse");
    if ((target || acc) && (!target || !acc || nslots < 1)) return fail("cspnk_forward_resident: scoring needs target, acc and nslots >= 1");
    if (W & 7) return fail("cspnk_forward_resident: W must be a multiple of 8 (whole octs of guidance)");
    if ((long)(K * K - 1) * H * W >= (1L << 30)) return fail("cspnk_forward_resident: guidance images of >= 2^30 elements are not supported (32-bit offsets)");
    if (!aligned16(guided) || !aligned16(x0) || !aligned16(out) || !aligned16(work) || (sparse && !aligned16(sparse)) || (target && !aligned16(target)))
        return fail("cspnk_forward_resident: tensors must be 16-byte aligned");
    if (seq == 0 || seq > 0x7fffff00u) return fail("cspnk_forward_resident: seq must be in [1, 2^31 - 256]");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ncu = kcu_count();
    if (ncu <= 0) return fail("cspnk_forward_resident: no device");
    KGeom g;
    cspn_resident_plan rp{};
    if (plan) rp = *plan;
    if (K == 3 && g_dtype == CSPN_F32) {       // see cspnk_resident_plan
        cspn_resident_plan qp = rp;
        qp.threads = 0;
        return cspn_detail::resident_pac3_f32(guided, x0, sparse, out, nullptr, nullptr, work, seq, host_err, B, H, W, T, blend, target, acc,
                                              nslots, &qp, stream);
    }
    if (rp.guard && (acc || !cspn_detail::kres_repair_fits(K, T)))
        return fail("cspnk_forward_resident: plan->guard serves unscored calls with T * (K / 2) <= 54");
    if (rp.tiles_x > 0 && rp.tiles_y > 0 && rp.tile_w > 0 && rp.tile_h > 0 && rp.steps_per_phase > 0 && rp.images_per_launch > 0) {
        const int Se = rp.steps_per_phase > T ? T : rp.steps_per_phase;
        if (!kgeom_fill(K, g_dtype, H, W, T, blend, ncu, B, Se, rp.tiles_x, rp.tiles_y, rp.tile_w, rp.tile_h, rp.threads > 0 ? rp.threads : KRES_THREADS, &g) ||
            (long)rp.images_per_launch * g.tiles_x * g.tiles_y > ncu)
            return fail("cspnk_forward_resident: the plan does not fit this problem / device (use cspnk_resident_plan)");
        g.imgs_per_launch = rp.images_per_launch;
    } else if (!kres_geometry(K, g_dtype, B, H, W, T, blend, ncu, rp.steps_per_phase, rp.threads, &g)) {
        return fail("cspnk_forward_resident: no resident tiling for K=%d B=%d %dx%d T=%d", K, B, H, W, T);
    }
    KResArgs a{};
    a.g = guided; a.x0 = x0; a.sparse = sparse; a.out = out;
    const size_t planes = (((size_t)2 * B * H * W * esize(state_dtype)) + 15) & ~(size_t)15;
    a.xbuf = work;
    a.status = reinterpret_cast<unsigned*>(static_cast<char*>(work) + planes);
    a.flags = a.status + 4;
    if ((size_t)g.tiles_x * g.tiles_y > ((size_t)H * W) / 8 + 1) return fail("cspnk_forward_resident: workspace too small for %d tiles per image", g.tiles_x * g.tiles_y);
    a.host_err = host_err; a.seq = seq;
    a.target = target; a.macc = acc; a.nslots = nslots;
    a.B = B; a.H = H; a.W = W; a.T = T; a.S = g.S;
    a.tw = g.tw; a.th = g.th; a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
    a.wo = g.wo; a.wr = g.wr; a.hxw = g.hxw; a.hyw = g.hyw; a.dr = g.dr; a.ls = g.ls;
    a.spin_limit = rp.spin_limit ? rp.spin_limit : (4u << 20);
    a.dbg = rp.debug_stamps;
    const bool clean = kregions_inside_image(g, H, W, T);
    const int score = acc ? 1 : 0;
    // The dot-product form (cspnk_d2.hip): K = 5, fp16 guidance, fp16 planes, one oct per thread.  One launch for the whole
    // batch: a workgroup refines its tile of image b, b + images_per_launch, ... back to back.
    const bool d2_able = K == 5 && g_dtype == CSPN_F16 && state_dtype == CSPN_F16 && g.no == 1;
    if (rp.step_form == CSPN_STEP_DOT2 && !d2_able)
        return fail("cspnk_forward_resident: the dot-product step form exists for K = 5 with fp16 guidance, fp16 planes and one oct per thread");
    if (d2_able && rp.step_form != CSPN_STEP_FMA) {
        a.ls = cspn_detail::kres_d2_row_stride(g.wo);
        a.b0 = 0;
        a.nb = g.imgs_per_launch < B ? g.imgs_per_launch : B;
        a.rounds = ceil_div(B, a.nb);
        a.last_chunk = 1;
        const int npf = cspn_detail::kres_d2_prefetch_channels(g.dr, a.ls, g.threads, a.rounds);
        const size_t ldsb = cspn_detail::kres_d2_lds_bytes(g.dr, a.ls, g.threads, npf);
        if (ldsb > 160 * 1024) return fail("cspnk_forward_resident: the dot-product form needs %zu bytes of LDS", ldsb);
        if (!cspn_detail::kres_d2_launch(&a, g.threads, a.nb * g.tiles_x * g.tiles_y, ldsb, blend, score, clean ? 1 : 0, npf, stream)) return 0;
        // (the dot-product form rounds the state to half after every step)
        if (rp.guard) return cspn_detail::kres_repair_launch(guided, g_dtype, K, x0, sparse, out, state_dtype, a.status, seq, B, H, W, T, 1, blend ? 1 : 0, ncu, stream);
        return 1;
    }
    for (int b0 = 0; b0 < B; b0 += g.imgs_per_launch) {
        a.b0 = b0;
        a.nb = (B - b0) < g.imgs_per_launch ? (B - b0) : g.imgs_per_launch;
        a.last_chunk = (b0 + g.imgs_per_launch >= B) ? 1 : 0;
        const int grid = a.nb * g.tiles_x * g.tiles_y;
        int ok = 0;
        if (g_dtype == CSPN_F32) {
            ok = klaunch_f32<5>(a, g.no, g.threads, grid, g.lds_bytes, blend, score, clean, st);
        } else if (K == 5) {
            ok = state_dtype == CSPN_F16 ? klaunch_k<5, __half>(a, g.no, g.threads, grid, g.lds_bytes, blend, score, clean, st)
                                         : klaunch_k<5, float>(a, g.no, g.threads, grid, g.lds_bytes, blend, score, clean, st);
        } else {
            ok = state_dtype == CSPN_F16 ? klaunch_k<3, __half>(a, g.no, g.threads, grid, g.lds_bytes, blend, score, clean, st)
                                         : klaunch_k<3, float>(a, g.no, g.threads, grid, g.lds_bytes, blend, score, clean, st);
        }
        if (!ok) return 0;
    }
    // (the FMA form rounds the state to the plane dtype at its phase boundaries)
    if (rp.guard) return cspn_detail::kres_repair_launch(guided, g_dtype, K, x0, sparse, out, state_dtype, a.status, seq, B, H, W, T, g.S, blend ? 1 : 0, ncu, stream);
    return 1;
}

int cspnk_transposed_resident(const void* wk, int w_dtype, int K, const void* g_T, const void* sparse_f32, int in_dtype, float* g_T_f32_out,
                              float* history, void* work, unsigned seq, unsigned* host_err, int B, int H, int W, int T, int premask,
                              const cspn_resident_plan* plan, cspn_stream_t stream) {
    if (plan && plan->guard && !cspn_detail::kres_repair_fits(5, T)) return fail("cspnk_transposed_resident: the guard re-computes at most 27 steps (T=%d)", T);
    if (!wk || !g_T || !history || !work || B <= 0 || H <= 0 || W <= 0 || T < 1) return fail("cspnk_transposed_resident: bad arguments");
    if (K != 5 || w_dtype != CSPN_F16) return fail("cspnk_transposed_resident: K = 5 with an fp16 tap volume (K=%d, dtype %d)", K, w_dtype);
    if (premask && !sparse_f32) return fail("cspnk_transposed_resident: premask needs sparse");
    if (in_dtype != CSPN_F32 && in_dtype != CSPN_F16) return fail("cspnk_transposed_resident: bad in_dtype %d", in_dtype);
    if (g_T_f32_out && (in_dtype != CSPN_F16 || !aligned16(g_T_f32_out)))
        return fail("cspnk_transposed_resident: g_T_f32_out goes with fp16 inputs (16-byte aligned)");
    if (W & 7) return fail("cspnk_transposed_resident: W must be a multiple of 8");
    if ((long)24 * H * W >= (1L << 30)) return fail("cspnk_transposed_resident: tap volumes of >= 2^30 elements per image are not supported");
    if (!aligned16(wk) || !aligned16(g_T) || !aligned16(history) || !aligned16(work) || (sparse_f32 && !aligned16(sparse_f32)))
        return fail("cspnk_transposed_resident: tensors must be 16-byte aligned");
    if (seq == 0 || seq > 0x7fffff00u) return fail("cspnk_transposed_resident: seq must be in [1, 2^31 - 256]");
    const int ncu = kcu_count();
    if (ncu <= 0) return fail("cspnk_transposed_resident: no device");
    const int blend = premask ? 1 : 0;
    cspn_resident_plan rp{};
    if (plan) rp = *plan;
    KGeom g;
    if (rp.tiles_x > 0 && rp.tiles_y > 0 && rp.tile_w > 0 && rp.tile_h > 0 && rp.steps_per_phase > 0 && rp.images_per_launch > 0) {
        const int Se = rp.steps_per_phase > T ? T : rp.steps_per_phase;
        if (!kgeom_fill(K, CSPN_F16, H, W, T, blend, ncu, B, Se, rp.tiles_x, rp.tiles_y, rp.tile_w, rp.tile_h, rp.threads > 0 ? rp.threads : KRES_THREADS, &g) ||
            (long)rp.images_per_launch * g.tiles_x * g.tiles_y > ncu)
            return fail("cspnk_transposed_resident: the plan does not fit this problem / device (use cspnk_resident_plan)");
        g.imgs_per_launch = rp.images_per_launch;
    } else if (!kres_geometry(K, CSPN_F16, B, H, W, T, blend, ncu, rp.steps_per_phase, rp.threads, &g)) {
        return fail("cspnk_transposed_resident: no resident tiling for K=%d B=%d %dx%d T=%d", K, B, H, W, T);
    }
    if (g.no != 1) return fail("cspnk_transposed_resident: needs a one-oct-per-thread tiling");
    KResArgs a{};
    a.g = wk; a.x0 = g_T; a.sparse = premask ? sparse_f32 : nullptr; a.out = g_T_f32_out; a.hist = history;
    const int tr = in_dtype == CSPN_F16 ? 2 : 1;
    const size_t planes = (((size_t)2 * B * H * W * esize(CSPN_F32)) + 15) & ~(size_t)15;
    a.xbuf = work;
    a.status = reinterpret_cast<unsigned*>(static_cast<char*>(work) + planes);
    a.flags = a.status + 4;
    a.host_err = host_err; a.seq = seq;
    a.B = B; a.H = H; a.W = W; a.T = T; a.S = g.S;
    a.tw = g.tw; a.th = g.th; a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
    a.wo = g.wo; a.wr = g.wr; a.hxw = g.hxw; a.hyw = g.hyw; a.dr = g.dr; a.ls = g.ls;
    a.spin_limit = rp.spin_limit ? rp.spin_limit : (4u << 20);
    a.dbg = rp.debug_stamps;
    const bool clean = kregions_inside_image(g, H, W, T);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // ONE launch for the whole batch (KRES_T_MAX_ROUNDS rounds of images_per_launch images at most per launch: the bounded neighbour wait
    // is sized for launches of a few hundred microseconds).  -DCSPN_KT_ONE_LAUNCH=0: one launch per round, the schedule up to round 4.
    const int ipl = g.imgs_per_launch < B ? g.imgs_per_launch : B;
    const int max_rounds = CSPN_KT_ONE_LAUNCH ? KRES_T_MAX_ROUNDS : 1;
    for (int b0 = 0; b0 < B; b0 += ipl * max_rounds) {
        a.b0 = b0;
        a.nb = (B - b0) < ipl ? (B - b0) : ipl;
        a.rounds = ceil_div(B - b0, ipl) < max_rounds ? ceil_div(B - b0, ipl) : max_rounds;
        a.last_chunk = (b0 + ipl * max_rounds >= B) ? 1 : 0;
        const int grid = a.rounds * a.nb * g.tiles_x * g.tiles_y;
        int ok = 0;
#define KT_CASE(BL, CL, NTHR) \
        if (!ok && blend == BL && (int)clean == CL && g.threads == NTHR) \
            ok = tr == 2 ? klaunch_inst<5, 1, BL, 0, CL, float, NTHR, __half, 2>(a, grid, g.lds_bytes, st) \
                         : klaunch_inst<5, 1, BL, 0, CL, float, NTHR, __half, 1>(a, grid, g.lds_bytes, st)
        KT_CASE(0, 0, 768); KT_CASE(0, 1, 768); KT_CASE(1, 0, 768); KT_CASE(1, 1, 768);
        KT_CASE(0, 0, 512); KT_CASE(0, 1, 512); KT_CASE(1, 0, 512); KT_CASE(1, 1, 512);
#undef KT_CASE
        if (!ok) return cspn_detail::last_error()[0] ? 0 : fail("cspnk_transposed_resident: no instance for %d threads", g.threads);
    }
    // the guard: a sweep that gave up is re-computed on the stream before the tail can read its planes (cspn_repair.hip)
    if (rp.guard) return cspn_detail::kres_sweep_repair_launch(wk, g_T, premask ? sparse_f32 : nullptr, in_dtype, g_T_f32_out, history, a.status, seq, B, H, W, T, blend, ncu, stream);
    return 1;
}

int cspnk_forward_resident_history(const void* guided, int g_dtype, int K, const void* x0, const void* sparse, void* history, void* wk_out,
                                   void* work, unsigned seq, unsigned* host_err, int B, int H, int W, int T, int blend,
                                   const cspn_resident_plan* plan, cspn_stream_t stream) {
    if (plan && plan->guard && !(K == 3 && g_dtype == CSPN_F32) && !(K == 5 && g_dtype == CSPN_F16 && cspn_detail::kres_repair_fits(5, T)))
        return fail("cspnk_forward_resident_history: plan->guard exists for K = 3 with fp32 guidance and for K = 5 with fp16 guidance up to 27 steps");
    if (!guided || !x0 || !history || !wk_out || !work || B <= 0 || H <= 0 || W <= 0 || T < 1)
        return fail("cspnk_forward_resident_history: bad arguments");
    if (K == 5 && g_dtype == CSPN_F16) {
        // BASELINE config 3's training forward: the dot-product kernel (cspnk_d2.hip) with fp16 planes — x0 / sparse / history are
        // fp16, every step's state goes to its history plane, wk_out receives the fp16 tap volume (pair-interleaved layout)
        if (blend != CSPN_BLEND_NONE && blend != CSPN_BLEND_SPARSE) return fail("cspnk_forward_resident_history: blend %d", blend);
        if (blend && !sparse) return fail("cspnk_forward_resident_history: blend needs sparse");
        if (W & 7) return fail("cspnk_forward_resident_history: W must be a multiple of 8");
        if ((long)24 * H * W >= (1L << 30)) return fail("cspnk_forward_resident_history: guidance images of >= 2^30 elements are not supported");
        if (!aligned16(guided) || !aligned16(x0) || !aligned16(history) || !aligned16(wk_out) || !aligned16(work) || (sparse && !aligned16(sparse)))
            return fail("cspnk_forward_resident_history: tensors must be 16-byte aligned");
        if (seq == 0 || seq > 0x7fffff00u) return fail("cspnk_forward_resident_history: seq must be in [1, 2^31 - 256]");
        const int ncu = kcu_count();
        if (ncu <= 0) return fail("cspnk_forward_resident_history: no device");
        cspn_resident_plan rp{};
        if (plan) rp = *plan;
        KGeom g;
        if (rp.tiles_x > 0 && rp.tiles_y > 0 && rp.tile_w > 0 && rp.tile_h > 0 && rp.steps_per_phase > 0 && rp.images_per_launch > 0) {
            const int Se = rp.steps_per_phase > T ? T : rp.steps_per_phase;
            if (!kgeom_fill(K, g_dtype, H, W, T, blend, ncu, B, Se, rp.tiles_x, rp.tiles_y, rp.tile_w, rp.tile_h, rp.threads > 0 ? rp.threads : KRES_THREADS, &g) ||
                (long)rp.images_per_launch * g.tiles_x * g.tiles_y > ncu)
                return fail("cspnk_forward_resident_history: the plan does not fit this problem / device (use cspnk_resident_plan)");
            g.imgs_per_launch = rp.images_per_launch;
        } else if (!kres_geometry(K, g_dtype, B, H, W, T, blend, ncu, rp.steps_per_phase, rp.threads, &g)) {
            return fail("cspnk_forward_resident_history: no resident tiling for K=%d B=%d %dx%d T=%d", K, B, H, W, T);
        }
        if (g.no != 1) return fail("cspnk_forward_resident_history: the K = 5 training form needs a one-oct-per-thread tiling");
        KResArgs a{};
        a.g = guided; a.x0 = x0; a.sparse = sparse; a.out = nullptr; a.hist = history; a.wk_out = wk_out;
        const size_t planes = (((size_t)2 * B * H * W * esize(CSPN_F16)) + 15) & ~(size_t)15;
        a.xbuf = work;
        a.status = reinterpret_cast<unsigned*>(static_cast<char*>(work) + planes);
        a.flags = a.status + 4;
        a.host_err = host_err; a.seq = seq;
        a.B = B; a.H = H; a.W = W; a.T = T; a.S = g.S;
        a.tw = g.tw; a.th = g.th; a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
        a.wo = g.wo; a.wr = g.wr; a.hxw = g.hxw; a.hyw = g.hyw; a.dr = g.dr;
        a.ls = cspn_detail::kres_d2_row_stride(g.wo);
        a.spin_limit = rp.spin_limit ? rp.spin_limit : (4u << 20);
        a.dbg = rp.debug_stamps;
        a.b0 = 0;
        a.nb = g.imgs_per_launch < B ? g.imgs_per_launch : B;
        a.rounds = ceil_div(B, a.nb);
        a.last_chunk = 1;
        const int npf = cspn_detail::kres_d2_prefetch_channels(g.dr, a.ls, g.threads, a.rounds);
        const size_t ldsb = cspn_detail::kres_d2_lds_bytes(g.dr, a.ls, g.threads, npf);
        if (ldsb > 160 * 1024) return fail("cspnk_forward_resident_history: %zu bytes of LDS", ldsb);
        if (!cspn_detail::kres_d2_launch(&a, g.threads, a.nb * g.tiles_x * g.tiles_y, ldsb, blend, 2, kregions_inside_image(g, H, W, T) ? 1 : 0, npf, stream)) return 0;
        // the guard: a forward that gave up gets its T planes and its tap volume re-computed on the stream (cspn_repair.hip)
        if (rp.guard) return cspn_detail::kres_history_repair_launch(guided, x0, blend ? sparse : nullptr, history, wk_out, a.status, seq, B, H, W, T, blend ? 1 : 0, ncu, stream);
        return 1;
    }
    if (K != 3 || g_dtype != CSPN_F32)
        return fail("cspnk_forward_resident_history: the training form exists for K = 3 with fp32 guidance and for K = 5 with fp16 guidance "
                    "and planes (K=%d, dtype %d: use cspn_pac_prepare + cspn_propagate with history)", K, g_dtype);
    cspn_resident_plan qp{};
    if (plan) qp = *plan;
    qp.threads = 0;
    return cspn_detail::resident_pac3_f32(guided, x0, sparse, nullptr, history, wk_out, work, seq, host_err, B, H, W, T, blend, nullptr, nullptr,
                                          0, &qp, stream);
}

}  // extern "C"

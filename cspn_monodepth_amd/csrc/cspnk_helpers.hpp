// cspnk_helpers.hpp — device helpers shared by the K x K weight-resident kernels (cspnk_resident.hip, cspnk_d2.hip):
// launch arguments, SGPR-base + 32-bit-offset global accesses, device-scope (sc1) exchange accesses, depth-plane I/O by
// octs, the in-register fp16 softmax.  Internal linkage (anonymous namespace), like cspn_common.hpp.
#pragma once
#include "cspn_common.hpp"

namespace {

struct KResArgs {
    const void* g;           // guided [B, K*K-1, H, W] f16 or f32 (GT)
    const void* x0;          // [B,H,W] ST: the coarse depth
    const void* sparse;      // [B,H,W] ST or null
    void* out;               // [B,H,W] ST
    void* xbuf;              // exchange planes [2][B,H,W] ST (workspace)
    unsigned* flags;         // [B * tiles_per_img] phase flags
    unsigned* status;        // [0] abort, [1] sticky error, [2] count-out counter
    unsigned* host_err;      // optional two host-mapped words (error, completion: include/cspn_hip.h)
    unsigned seq;
    const void* target;      // SCORE: [B,H,W] ST
    double* macc;
    int nslots;
    int B, H, W, T, S;
    int tw, th, tiles_x, tiles_y;
    int wo, wr, hxw, hyw, dr, ls;
    int b0, nb, last_chunk;
    void* hist;              // cspnk_d2 training form: [T][B,H,W] ST receives x_1 .. x_T (`out` unused)
    void* wk_out;            // cspnk_d2 training form: the softmax taps as the fp16 tap volume the backward streams (Taps<__half> layout)
    int rounds;              // cspnk_d2: a workgroup refines its tile of images b0 + r * nb + (tile / tiles_per_img), r = 0 .. rounds-1, back to back
    unsigned spin_limit;
    unsigned long long* dbg;  // developer probe: [grid][16] wall-clock stamps (100 MHz) per workgroup, or null
};

#define GLB __attribute__((address_space(1)))
typedef GLB char* gptr;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) v2f* lds_cv2f_ptr;
typedef const volatile __attribute__((address_space(3))) float* lds_cf_ptr;

template <typename T>
__device__ __forceinline__ T* kuniform_ptr(T* p) {      // a wave-uniform pointer, pinned to an SGPR pair
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// SGPR base + 32-bit BYTE offset, typed as a global (address space 1) pointer: no 64-bit address arithmetic, no flat loads
__device__ __forceinline__ gptr atb(const void* base, unsigned byte_off) {
    return (gptr) reinterpret_cast<unsigned long long>(base) + byte_off;
}
__device__ __forceinline__ uint4 ld16(gptr p) {
    const v4u v = *reinterpret_cast<const GLB v4u*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16(gptr p, uint4 v) {
    const v4u w = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<GLB v4u*>(p) = w;
}
#ifndef CSPN_KRES_HIST_NT
#define CSPN_KRES_HIST_NT 0     // A/B: 1 = non-temporal stores of the fp16 history planes / tap volume of the K x K training forms — measured
                                // SLOWER (cspnk_d2 with history 83-85 -> 92-93 us, the transposed launches 44.4 -> 47.9 us), unlike the fp32 planes
                                // of the 3 x 3 forms (cspn_resident.hip CSPN_RES_HIST_NT)
#endif
__device__ __forceinline__ void st16_hist(gptr p, uint4 v) {
    const v4u w = {v.x, v.y, v.z, v.w};
    if (CSPN_KRES_HIST_NT) __builtin_nontemporal_store(w, reinterpret_cast<GLB v4u*>(p));
    else *reinterpret_cast<GLB v4u*>(p) = w;
}
__device__ __forceinline__ unsigned ld4u(gptr p) { return *reinterpret_cast<const GLB unsigned*>(p); }
__device__ __forceinline__ uint2 ld8u(gptr p) {
    const v2u v = *reinterpret_cast<const GLB v2u*>(p);
    return make_uint2(v.x, v.y);
}
// device-scope (sc1) accesses: coherent across the XCDs' private L2s without cache-wide write-back / invalidate
__device__ __forceinline__ uint4 ld16_dev(gptr p) {
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const GLB unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const GLB unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
}
__device__ __forceinline__ uint2 ld8u_dev(gptr p) {
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const GLB unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)lo, (unsigned)(lo >> 32));
}
__device__ __forceinline__ unsigned ld4u_dev(gptr p) {
    return __hip_atomic_load(reinterpret_cast<const GLB unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st16_dev(const void* base, unsigned byte_off, uint4 v) {
    // ONE 16-byte device-scope store (the compiler only offers <= 8-byte atomics; two of them touch every line twice)
    const v4u w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(byte_off), "v"(w), "s"(base) : "memory");
}

__device__ __forceinline__ float h2f_lo(unsigned w) { return __half2float(__ushort_as_half((unsigned short)(w & 0xffffu))); }
__device__ __forceinline__ float h2f_hi(unsigned w) { return __half2float(__ushort_as_half((unsigned short)(w >> 16))); }
__device__ __forceinline__ unsigned f2h_bits(float v) { return (unsigned)__half_as_ushort(__float2half_rn(v)); }
__device__ __forceinline__ unsigned pack_h2(float lo, float hi) { return f2h_bits(lo) | (f2h_bits(hi) << 16); }

// The depth planes of a call are fp16 or fp32: an OCT (8 pixels of a row) is one or two 16-byte accesses, a PAIR (the two
// ring pixels left / right of a region row) one 4- or 8-byte access.
template <typename ST> struct StateIO;
template <> struct StateIO<__half> {
    struct Oct { uint4 a; };
    struct Pair { unsigned a; };
    static __device__ __forceinline__ Oct ld_oct(const void* b, unsigned e) { return Oct{ld16(atb(b, e * 2u))}; }
    static __device__ __forceinline__ Oct ld_oct_dev(const void* b, unsigned e) { return Oct{ld16_dev(atb(b, e * 2u))}; }
    static __device__ __forceinline__ Pair ld_pair(const void* b, unsigned e) { return Pair{ld4u(atb(b, e * 2u))}; }
    static __device__ __forceinline__ Pair ld_pair_dev(const void* b, unsigned e) { return Pair{ld4u_dev(atb(b, e * 2u))}; }
    static __device__ __forceinline__ void to_f8(const Oct& o, float (&v)[8]) {
        v[0] = h2f_lo(o.a.x); v[1] = h2f_hi(o.a.x); v[2] = h2f_lo(o.a.y); v[3] = h2f_hi(o.a.y);
        v[4] = h2f_lo(o.a.z); v[5] = h2f_hi(o.a.z); v[6] = h2f_lo(o.a.w); v[7] = h2f_hi(o.a.w);
    }
    static __device__ __forceinline__ void to_f2(const Pair& p, float& a, float& b) { a = h2f_lo(p.a); b = h2f_hi(p.a); }
    static __device__ __forceinline__ Oct from_f8(const float (&v)[8]) {
        return Oct{make_uint4(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7]))};
    }
    static __device__ __forceinline__ void st_oct(void* b, unsigned e, const Oct& o) { st16(atb(b, e * 2u), o.a); }
    static __device__ __forceinline__ void st_oct_hist(void* b, unsigned e, const Oct& o) { st16_hist(atb(b, e * 2u), o.a); }
    static __device__ __forceinline__ void st_oct_dev(void* b, unsigned e, const Oct& o) { st16_dev(b, e * 2u, o.a); }
};
template <> struct StateIO<float> {
    struct Oct { uint4 a, b; };
    struct Pair { uint2 a; };
    static __device__ __forceinline__ Oct ld_oct(const void* b, unsigned e) { return Oct{ld16(atb(b, e * 4u)), ld16(atb(b, e * 4u + 16u))}; }
    static __device__ __forceinline__ Oct ld_oct_dev(const void* b, unsigned e) { return Oct{ld16_dev(atb(b, e * 4u)), ld16_dev(atb(b, e * 4u + 16u))}; }
    static __device__ __forceinline__ Pair ld_pair(const void* b, unsigned e) { return Pair{ld8u(atb(b, e * 4u))}; }
    static __device__ __forceinline__ Pair ld_pair_dev(const void* b, unsigned e) { return Pair{ld8u_dev(atb(b, e * 4u))}; }
    static __device__ __forceinline__ void to_f8(const Oct& o, float (&v)[8]) {
        v[0] = __uint_as_float(o.a.x); v[1] = __uint_as_float(o.a.y); v[2] = __uint_as_float(o.a.z); v[3] = __uint_as_float(o.a.w);
        v[4] = __uint_as_float(o.b.x); v[5] = __uint_as_float(o.b.y); v[6] = __uint_as_float(o.b.z); v[7] = __uint_as_float(o.b.w);
    }
    static __device__ __forceinline__ void to_f2(const Pair& p, float& a, float& b) { a = __uint_as_float(p.a.x); b = __uint_as_float(p.a.y); }
    static __device__ __forceinline__ Oct from_f8(const float (&v)[8]) {
        return Oct{make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])),
                   make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]))};
    }
    static __device__ __forceinline__ void st_oct(void* b, unsigned e, const Oct& o) { st16(atb(b, e * 4u), o.a); st16(atb(b, e * 4u + 16u), o.b); }
    static __device__ __forceinline__ void st_oct_hist(void* b, unsigned e, const Oct& o) { st16_hist(atb(b, e * 4u), o.a); st16_hist(atb(b, e * 4u + 16u), o.b); }
    static __device__ __forceinline__ void st_oct_dev(void* b, unsigned e, const Oct& o) { st16_dev(b, e * 4u, o.a); st16_dev(b, e * 4u + 16u, o.b); }
};

__device__ __forceinline__ unsigned pk_mul_f16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_max_f16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// float(half `hi ? high : low` of w) + f, as ONE v_fma_mix_f32 (half * 1.0 + float): exact product, one rounding
__device__ __forceinline__ float half_plus_float(unsigned w, int hi, float f) {
    float out;
    if (hi) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(f));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(f));
    return out;
}
// Softmax over the NT channels of ONE pixel — the low (HF = 0) or the high half of the NT packed words — in place: raw logits
// -> fp16 weights.  nmx = -max over the channels.  HF is a template parameter: with the asm variants selected by a loop
// variable the compiler kept the two-trip loop rolled (selects and branches around every asm: 3x the time).
template <int NT, int HF>
__device__ __forceinline__ void softmax_half(unsigned (&w)[NT], float nmx) {
    float v[NT];
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        // v - max straight from the packed half (v_fma_mix_f32: half * 1 + float, one rounding = the subtraction's)
        v[c] = softmax_exp<__half>(half_plus_float(w[c], HF, nmx));
        den += v[c];
    }
    const float inv = reciprocal_refined(den);
#pragma unroll
    for (int c = 0; c < NT; ++c) w[c] = HF ? mul_into_half_hi(w[c], v[c], inv) : mul_into_half_lo(w[c], v[c], inv);
}
template <int NT>
__device__ __forceinline__ void softmax_pair(unsigned (&w)[NT]) {
    unsigned mx2 = 0xfc00fc00u;                           // (-inf, -inf): channel maximum of both pixels at once, on the raw
#pragma unroll
    for (int c = 0; c < NT; ++c) mx2 = pk_max_f16(mx2, w[c]);                           // halfs (exact: v_pk_max_f16)
    const float nlo = -h2f_lo(mx2), nhi = -h2f_hi(mx2);
    softmax_half<NT, 0>(w, nlo);
    softmax_half<NT, 1>(w, nhi);
}
__device__ __forceinline__ unsigned comp(const uint4& r, int k) { return k == 0 ? r.x : (k == 1 ? r.y : (k == 2 ? r.z : r.w)); }
__device__ __forceinline__ void set_comp(uint4& r, int k, unsigned v) {
    if (k == 0) r.x = v; else if (k == 1) r.y = v; else if (k == 2) r.z = v; else r.w = v;
}


// ---- the dot-product step form (cspnk_d2.hip) and its bit-exact re-computation (cspn_repair.hip) ------------------------------------
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot2(unsigned w, unsigned x, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), __builtin_bit_cast(h2, x), acc, false);
}
// acc + half(WH of w) * half(XH of x): one v_fma_mix_f32 with both factors taken from packed halfs.  Written as C++ (the
// backend folds the two conversions into the instruction), NOT as inline asm: a v_dot2c_f32_f16 result read by the very next
// VALU instruction needs wait states that the hazard recogniser only inserts in front of instructions it can see — the asm
// form read stale accumulators (tools/probes/d2_debug.py).  The step loop keeps the conversions from being hoisted out of the
// loop (32 registers) by passing the single-tap registers through an empty asm once per step.
template <int WH, int XH>
__device__ __forceinline__ float mix_hh(unsigned w, unsigned x, float acc) {
    const h2 wv = __builtin_bit_cast(h2, w), xv = __builtin_bit_cast(h2, x);
    return __builtin_fmaf((float)wv[WH], (float)xv[XH], acc);
}
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {      // round to nearest even, both halves in one instruction (gfx950)
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// half(HF of w) * scale + add, as ONE v_fma_mix_f32: the softmax exponent's argument (v - max) * log2(e) with a single rounding
template <int HF>
__device__ __forceinline__ float half_scaled(unsigned w, float scale, float add) {
    float out;
    if constexpr (HF != 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(scale), "v"(add));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(out) : "v"(w), "v"(scale), "v"(add));
    return out;
}

// Tap-pair slots of one pixel (12 registers): window row rr = dy + 2, channel c of tap (dy, dx) = lin < 12 ? lin : lin - 1 with
// lin = 5 rr + dx + 2 (CSPN_ours.py:35-39: the 24 channels are the taps in row-major order without the centre).
//   slot 2 rr, 2 rr + 1 (rr = 0, 1)        : (dx -2, -1), (dx 0, +1)          slot 10: single dx +2 of rows 0 (low half), 1 (high)
//   slot 4, 5 (centre row)                  : (dx -2, -1), (dx +1, +2)
//   slot 6 + 2 (rr - 3), 7 + 2 (rr - 3)     : (dx -2, -1), (dx 0, +1)          slot 11: single dx +2 of rows 3 (low half), 4 (high)
constexpr int D2_LO[12] = {0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 4, 18};
constexpr int D2_HI[12] = {1, 3, 6, 8, 11, 13, 15, 17, 20, 22, 9, 23};

constexpr int d2_slot(int c) {            // the tap-pair register that holds channel c ...
    for (int s = 0; s < 12; ++s)
        if (D2_LO[s] == c || D2_HI[s] == c) return s;
    return -1;
}
constexpr int d2_half(int c) { return D2_HI[d2_slot(c)] == c ? 1 : 0; }      // ... and the half of it

// softmax over the 24 channels of ONE pixel (half HF of the 24 packed words) -> its 12 tap-pair registers.  nmx = -max * log2(e).
template <int HF>
__device__ __forceinline__ void softmax_to_pairs(const unsigned (&w)[24], float nmx, unsigned (&out)[12]) {
    constexpr float L2E = 1.44269502162933349609375f;
    float v[24];
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < 24; ++c) {
        v[c] = __builtin_amdgcn_exp2f(half_scaled<HF>(w[c], L2E, nmx));
        den += v[c];                                          // channel order, as every softmax of the engine
    }
    const float inv = reciprocal_refined(den);
#pragma unroll
    for (int s = 0; s < 12; ++s) out[s] = cvt_pk_f16(v[D2_LO[s]] * inv, v[D2_HI[s]] * inv);
}


}  // namespace

// exchange_litmus.hip — a cross-XCD message-passing litmus on EXACTLY the instruction sequences of the resident kernels' border exchange
// (cspn_resident.hip "publish the interior quads ... wait for the 8 neighbouring tiles" / the halo staging of the next phase):
//
//   producer                                           consumer
//   global_store_dwordx4 ... sc1   (payload, st4_dev)  global_load_dword ... sc1  (relaxed agent-scope poll of the flag, one lane)
//   s_waitcnt vmcnt(0)                                  s_barrier
//   s_barrier                                           global_load_dwordx2 ... sc1 x 2  (payload, ld4_dev)
//   global_store_dword ... sc1 (relaxed flag, lane 0)
//
// Nothing in it is a fence: the protocol rests on (1) a device-scope store being globally performed once vmcnt has dropped, (2) the
// barrier ordering every wavefront's wait before lane 0's flag store, (3) device-scope loads issued after the flag was seen not being
// served from a stale cache line (sc1 loads miss the non-coherent L2 lines of another XCD by definition of the scope).  DESIGN.md §4.1b
// argues it; this probe tests it where the whole-kernel soak cannot point: workgroup pairs pinned to DIFFERENT XCDs (blockIdx % 8: checked
// with HW_REG_XCC_ID), both sides producer and consumer in every round as the tiles are, two rotating planes as the kernels' xbuf, payloads
// of 1 .. 512 quads, optionally a co-running stream kernel that keeps the memory system busy, and a NEGATIVE CONTROL (the s_waitcnt vmcnt(0)
// dropped) that must produce stale reads — a litmus that cannot fail proves nothing.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exchange_litmus tools/probes/exchange_litmus.hip
//   /tmp/exchange_litmus [rounds_per_launch=200000] [launches=8] [pressure=1] [quick=0]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define GLB __attribute__((address_space(1)))

// the product's st4_dev (cspn_resident.hip): SGPR base + 32-bit VGPR byte offset, ONE 16-byte device-scope store
__device__ __forceinline__ void st4_dev(float* base, unsigned elem, v4f v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(elem * 4u), "v"(v), "s"(base) : "memory");
}
// the product's ld4_dev: two relaxed agent-scope 8-byte atomic loads
__device__ __forceinline__ v4f ld4_dev(const float* p) {
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v4f v;
    v.x = __uint_as_float((unsigned)lo); v.y = __uint_as_float((unsigned)(lo >> 32));
    v.z = __uint_as_float((unsigned)hi); v.w = __uint_as_float((unsigned)(hi >> 32));
    return v;
}

struct Args {
    float* plane[2];             // [pairs * 2][MAXQ quads] each
    unsigned* flags;             // [pairs * 2], one per workgroup (64-byte apart)
    unsigned long long* bad;     // [0] stale payload words seen, [1] handshakes completed, [2] time-outs
    unsigned* xcc;               // [grid]
    int rounds, quads;           // handshakes per launch; payload quads per workgroup and round
    unsigned seq0;               // flag value base of this launch (flags are monotonic across launches)
};
constexpr int THREADS = 512, MAXQ = 512, FSTRIDE = 16;

__device__ __forceinline__ float payload_word(unsigned wg, unsigned round, unsigned idx) {
    return __uint_as_float(((wg * 2654435761u) ^ (round * 40503u + idx * 97u)) & 0x7fffffu | 0x3f000000u);      // a finite float, unique per (wg, round, idx)
}

// WAIT = 1: the product's sequence.  WAIT = 0: the negative control — no s_waitcnt vmcnt(0) between the payload stores and the barrier.
template <int WAIT>
__global__ __launch_bounds__(THREADS) void litmus(const Args a) {
    const unsigned wg = blockIdx.x, partner = wg ^ 1u, tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[wg] = id & 0xf;
    }
    __shared__ int give_up;
    if (tid == 0) give_up = 0;
    __builtin_amdgcn_s_barrier();
    unsigned long long stale = 0;
    int done = 0;
    for (int r = 1; r <= a.rounds; ++r) {
        const unsigned want = a.seq0 + (unsigned)r;
        float* const out = a.plane[r & 1] + (size_t)wg * MAXQ * 4;
        const float* const in = a.plane[r & 1] + (size_t)partner * MAXQ * 4;
        // -- publish
        if ((int)tid < a.quads) {
            v4f v;
            v.x = payload_word(wg, want, 4 * tid); v.y = payload_word(wg, want, 4 * tid + 1);
            v.z = payload_word(wg, want, 4 * tid + 2); v.w = payload_word(wg, want, 4 * tid + 3);
            st4_dev(out, 4 * tid, v);
        }
        if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // (bare: __syncthreads() would add the release fence the product does not rely on)
        if (tid == 0) __hip_atomic_store(a.flags + (size_t)wg * FSTRIDE, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // -- wait for the partner's flag (one lane, relaxed agent-scope loads, bounded)
        if (tid == 0) {
            const unsigned* f = a.flags + (size_t)partner * FSTRIDE;
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (++spins > (64u << 20)) { give_up = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (give_up) break;
        // -- read the partner's payload of THIS round
        if ((int)tid < a.quads) {
            const v4f v = ld4_dev(in + 4 * tid);
            stale += (v.x != payload_word(partner, want, 4 * tid)) + (v.y != payload_word(partner, want, 4 * tid + 1)) +
                     (v.z != payload_word(partner, want, 4 * tid + 2)) + (v.w != payload_word(partner, want, 4 * tid + 3));
        }
        ++done;
    }
    if (stale) atomicAdd(a.bad, stale);
    if (tid == 0) { atomicAdd(a.bad + 1, (unsigned long long)done); if (give_up) atomicAdd(a.bad + 2, 1ull); }
}

// memory pressure: a stream triad over a buffer larger than the Infinity Cache, looping until told to stop
__global__ __launch_bounds__(256) void pressure(float4* x, const float4* y, size_t n, const volatile unsigned* stop, int max_passes) {
    for (int pass = 0; pass < max_passes && !*stop; ++pass)
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            float4 v = y[i];
            v.x += 1.f; v.y += 2.f; v.z += 3.f; v.w += 4.f;
            x[i] = v;
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200000, launches = argc > 2 ? atoi(argv[2]) : 8;
    const int with_pressure = argc > 3 ? atoi(argv[3]) : 1, quick = argc > 4 ? atoi(argv[4]) : 0;
    const int pairs = 64, grid = 2 * pairs;      // 128 workgroups: each of a pair on another XCD (2k % 8 != (2k + 1) % 8), half the CUs left to the stream kernel
    Args a{};
    CK(hipMalloc(&a.plane[0], (size_t)grid * MAXQ * 16)); CK(hipMalloc(&a.plane[1], (size_t)grid * MAXQ * 16));
    CK(hipMalloc(&a.flags, (size_t)grid * FSTRIDE * 4)); CK(hipMalloc(&a.bad, 3 * 8)); CK(hipMalloc(&a.xcc, grid * 4));
    CK(hipMemset(a.plane[0], 0, (size_t)grid * MAXQ * 16)); CK(hipMemset(a.plane[1], 0, (size_t)grid * MAXQ * 16));
    CK(hipMemset(a.flags, 0, (size_t)grid * FSTRIDE * 4));
    hipStream_t s_l, s_p;
    CK(hipStreamCreateWithFlags(&s_l, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_p, hipStreamNonBlocking));
    float4 *px = nullptr, *py = nullptr;
    unsigned* stop = nullptr;
    const size_t pn = (size_t)24 << 20;          // 2 x 384 MB
    if (with_pressure) {
        CK(hipMalloc(&px, pn * 16)); CK(hipMalloc(&py, pn * 16)); CK(hipMemset(py, 0, pn * 16));
        CK(hipHostMalloc(&stop, 4, hipHostMallocMapped)); *stop = 0;
    }
    unsigned seq = 0;
    int rc = 0;
    const int qs_full[] = {1, 4, 16, 64, 512}, qs_quick[] = {1, 64};
    const int* qs = quick ? qs_quick : qs_full;
    const int nq = quick ? 2 : 5;
    for (int control = 0; control < 2; ++control) {            // 0: the product's sequence (must be clean); 1: negative control (must fail)
        unsigned long long tot_bad = 0, tot_done = 0, tot_to = 0;
        for (int qi = 0; qi < nq; ++qi) {
            a.quads = qs[qi];
            a.rounds = control ? (rounds < 20000 ? rounds : 20000) : rounds;
            const int nl = control ? 2 : launches;
            CK(hipMemsetAsync(a.bad, 0, 24, s_l)); CK(hipStreamSynchronize(s_l));
            if (with_pressure) { *stop = 0; hipLaunchKernelGGL(pressure, dim3(128), dim3(256), 0, s_p, px, py, pn, stop, 1 << 20); }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, s_l));
            for (int l = 0; l < nl; ++l) {
                a.seq0 = seq; seq += (unsigned)a.rounds;
                if (control) hipLaunchKernelGGL(litmus<0>, dim3(grid), dim3(THREADS), 0, s_l, a);
                else hipLaunchKernelGGL(litmus<1>, dim3(grid), dim3(THREADS), 0, s_l, a);
            }
            CK(hipEventRecord(e1, s_l)); CK(hipStreamSynchronize(s_l));
            if (with_pressure) { *stop = 1; CK(hipStreamSynchronize(s_p)); }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[3]; CK(hipMemcpy(h, a.bad, 24, hipMemcpyDeviceToHost));
            std::vector<unsigned> xcc(grid); CK(hipMemcpy(xcc.data(), a.xcc, grid * 4, hipMemcpyDeviceToHost));
            int cross = 0; for (int p = 0; p < pairs; ++p) cross += xcc[2 * p] != xcc[2 * p + 1];
            printf("%s  payload %3d quad(s): %llu handshakes (%d of %d pairs on two XCDs), %.2f us each, stale words %llu, time-outs %llu%s\n",
                   control ? "NEGATIVE CONTROL (no vmcnt(0))" : "product sequence              ", a.quads, h[1], cross, pairs,
                   h[1] ? ms * 1e3 * grid / (double)h[1] : 0.0, h[0], h[2], with_pressure ? "  [stream kernel co-running]" : "");
            tot_bad += h[0]; tot_done += h[1]; tot_to += h[2];
            if (cross != pairs) { printf("FAIL: not every pair sat on two XCDs\n"); rc = 1; }
        }
        if (!control) {
            printf("== product sequence: %llu handshakes, %llu stale words, %llu time-outs -> %s\n", tot_done, tot_bad, tot_to, tot_bad || tot_to ? "FAIL" : "clean");
            if (tot_bad || tot_to) rc = 1;
        } else {
            printf("== negative control: %llu handshakes, %llu stale words -> %s\n", tot_done, tot_bad,
                   tot_bad ? "fails as it must (the litmus is sensitive to the ordering it tests)" : "DID NOT FAIL (the litmus may be insensitive)");
            if (!tot_bad) rc = rc ? rc : 3;
        }
    }
    return rc;
}

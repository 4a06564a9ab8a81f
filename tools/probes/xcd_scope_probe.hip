// xcd_scope_probe.hip — developer probe (not product): can tiles that sit on the SAME XCD exchange borders through that XCD's L2
// (stores + loads with sc0: "workgroup scope", which has to be coherent at L2 under threadgroup-split mode) instead of through
// memory (sc1: agent scope, coherent across the 8 XCDs)?  Measures the cost of one publish -> flag -> poll -> read exchange per
// phase for both scopes, checks for stale data, and prints which XCC every workgroup ran on (is blockIdx % 8 the XCD?).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/xcd_scope_probe tools/probes/xcd_scope_probe.hip && /tmp/xcd_scope_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
template <int SCOPE> __device__ inline void st4_s(float* p, v4f v) {
    if (SCOPE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (SCOPE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int SCOPE> __device__ inline v4f ld4_s(const float* p) {
    v4f v;
    if (SCOPE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE> __device__ inline void st1_s(unsigned* p, unsigned v) {
    if (SCOPE == 1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (SCOPE == 0) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int SCOPE> __device__ inline unsigned ld1_s(const unsigned* p) {
    unsigned v;
    if (SCOPE == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Args {
    float* buf[2];
    unsigned* flags;
    unsigned* err;
    unsigned* xcc;          // [grid] XCC id of every workgroup
    unsigned long long* t;  // [grid] wall-clock ticks spent in the exchanges
    int tiles_x, tiles_y, per_img, phases, work;
    unsigned seq;
};
constexpr int TILE = 4096, THREADS = 512;          // floats published per tile and phase (16 KB: a config-2 tile border is ~13 KB)

__device__ inline int xcd_contiguous_id(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, x = bid & 7, j = bid >> 3;
    return x * q + (x < r ? x : r) + j;
}

// SCOPE 1: sc1 everywhere (what the product does).  SCOPE 0: sc0 stores / loads.  SCOPE 2: plain stores, sc0 loads.
// SCOPE 3: plain stores (write-through to the XCD's L2), the flag polled with a returning L2 atomic (add 0), then the CU's L1
// invalidated (buffer_inv sc0) and the halo read with plain loads — coherent through the L2 for tiles of one XCD.
__device__ inline unsigned poll_l2(unsigned* p) {
    unsigned v, z = 0u;       // a returning atomic is executed at the L2 (written as asm: the compiler turns fetch_add(p, 0) into a plain load)
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
    return v;
}
__device__ inline v4f ld4_plain(const float* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE>
__global__ __launch_bounds__(THREADS) void xchg(const Args a) {
    const int tile = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int img = tile / a.per_img, t = tile - img * a.per_img;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[tile] = id & 0xf;
    }
    float acc = (float)tid;
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    unsigned long long spent = 0;
    for (int p = 0; p < a.phases; ++p) {
        for (int i = 0; i < a.work; ++i) acc = fmaf(acc, 1.0000001f, 0.5f);
        const unsigned long long t0 = wall_clock64();
        float* mine = a.buf[p & 1] + (size_t)tile * TILE;
        for (int i = tid * 4; i < TILE; i += THREADS * 4) {
            const v4f v = {acc, acc + 1, acc + 2, (float)(p + tile + (int)(a.seq & 1023))};
            st4_s<SCOPE>(mine + i, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned want = a.seq + (unsigned)p + 1u;
        if (tid == 0) st1_s<SCOPE>(a.flags + tile, want);
        if (tid < 9 && tid != 4) {
            const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
            if (ny >= 0 && ny < a.tiles_y && nx >= 0 && nx < a.tiles_x) {
                const unsigned* f = a.flags + img * a.per_img + ny * a.tiles_x + nx;
                int spins = 0;
                while ((int)((SCOPE == 3 ? poll_l2(const_cast<unsigned*>(f)) : ld1_s<SCOPE == 1 ? 1 : 0>(f)) - want) < 0) {
                    if (++spins > (1 << 20)) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
        if (bad) { if (tid == 0) atomicAdd(a.err, 1u); return; }
        if (SCOPE == 3) asm volatile("buffer_inv sc0" ::: "memory");
        for (int k = 0; k < 9; ++k) {
            if (k == 4) continue;
            const int ny = ty + k / 3 - 1, nx = tx + k % 3 - 1;
            if (ny < 0 || ny >= a.tiles_y || nx < 0 || nx >= a.tiles_x) continue;
            const int nt = img * a.per_img + ny * a.tiles_x + nx;
            const float* nb = a.buf[p & 1] + (size_t)nt * TILE;
            for (int i = tid * 4; i < TILE / 8; i += THREADS * 4) {
                const v4f q = SCOPE == 3 ? ld4_plain(nb + i) : ld4_s<SCOPE == 1 ? 1 : 0>(nb + i);
                acc += q.x;
                if (q.w != (float)(p + nt + (int)(a.seq & 1023))) bad = 2;
            }
        }
        __syncthreads();
        if (bad == 2) { if (tid == 0) atomicAdd(a.err, 1000u); bad = 0; }
        __syncthreads();
        spent += wall_clock64() - t0;
    }
    if (tid == 0) { a.t[tile] = spent; a.buf[0][(size_t)tile * TILE] = acc; }
}

int main() {
    const int imgs = 24, tiles_x = 2, tiles_y = 5;
    Args a{};
    a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.per_img = tiles_x * tiles_y;
    const int ntiles = imgs * a.per_img;
    hipMalloc(&a.buf[0], (size_t)ntiles * TILE * 4); hipMalloc(&a.buf[1], (size_t)ntiles * TILE * 4);
    hipMalloc(&a.flags, ntiles * 4); hipMalloc(&a.err, 4); hipMalloc(&a.xcc, ntiles * 4); hipMalloc(&a.t, ntiles * 8);
    hipMemset(a.flags, 0, ntiles * 4); hipMemset(a.err, 0, 4);
    unsigned seq = 0;
    for (int scope : {1, 3, 1, 3}) {
        for (int phases : {0, 8}) {
            a.phases = phases; a.work = 4000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0);
                for (int i = 0; i < 20; ++i) {
                    a.seq = seq; seq += 64;
                    if (scope == 1) hipLaunchKernelGGL(xchg<1>, dim3(ntiles), dim3(THREADS), 0, 0, a);
                    else if (scope == 0) hipLaunchKernelGGL(xchg<0>, dim3(ntiles), dim3(THREADS), 0, 0, a);
                    else if (scope == 3) hipLaunchKernelGGL(xchg<3>, dim3(ntiles), dim3(THREADS), 0, 0, a);
                    else hipLaunchKernelGGL(xchg<2>, dim3(ntiles), dim3(THREADS), 0, 0, a);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms * 50.f < best) best = ms * 50.f;
            }
            unsigned err = 0; hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost); hipMemset(a.err, 0, 4);
            std::vector<unsigned long long> t(ntiles); hipMemcpy(t.data(), a.t, ntiles * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : t) mean += (double)v; mean = phases ? mean / ntiles / phases / 100.0 : 0.0;
            printf("scope %s phases %d: %.2f us per launch, exchange %.2f us per phase (mean over tiles), err %u\n",
                   scope == 1 ? "sc1      " : (scope == 0 ? "sc0      " : (scope == 3 ? "L2-local " : "plain+sc0")), phases, best, mean, err);
        }
    }
    std::vector<unsigned> x(ntiles); hipMemcpy(x.data(), a.xcc, ntiles * 4, hipMemcpyDeviceToHost);
    int mixed = 0;
    for (int i = 0; i < imgs; ++i) { bool same = true; for (int k = 1; k < a.per_img; ++k) same = same && x[i * a.per_img + k] == x[i * a.per_img]; if (!same) ++mixed; }
    printf("images whose %d tiles span more than one XCC: %d of %d;  XCC of the first tile of each image:", a.per_img, mixed, imgs);
    for (int i = 0; i < imgs; ++i) printf(" %u", x[i * a.per_img]);
    printf("\n");
    return 0;
}

// coop_probe.hip — developer probe (not product): what does a neighbour-flag halo exchange between co-resident
// workgroups cost on MI355X, and what does hipLaunchCooperativeKernel add to a launch?
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/coop_probe tools/probes/coop_probe.hip && gpurun_out/coop_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args {
    float* buf[2];          // exchange planes [ntiles][TILE]
    unsigned* flags;        // [ntiles]
    unsigned* err;          // [1]
    float* sink;            // [ntiles]
    int tiles_x, tiles_y, per_img, ntiles, phases;
    unsigned seq;
    int spin_work;          // fake compute per phase (fma iterations)
    int mode;               // 0: fences in every thread, 1: only the flag writer / pollers synchronise at agent scope
};

typedef float v4f __attribute__((ext_vector_type(4)));
// device-scope (sc1) 128-bit accesses: coherent across the XCDs' private L2s without cache-wide write-back / invalidate
__device__ inline void st4_dev(float* p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ inline v4f ld4_dev(const float* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

constexpr int TILE = 7168;          // floats per tile (28 KB)
constexpr int THREADS = 512;

__global__ __launch_bounds__(THREADS) void xchg(const Args a) {
    const int tile = blockIdx.x;
    const int img = tile / a.per_img, t = tile - img * a.per_img;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int tid = threadIdx.x;
    float acc = (float)tid;
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int p = 0; p < a.phases; ++p) {
        // fake compute
        for (int i = 0; i < a.spin_work; ++i) acc = fmaf(acc, 1.0000001f, 0.5f);
        // publish my tile
        float* mine = a.buf[p & 1] + (size_t)tile * TILE;
        if (a.mode >= 2) {
            for (int i = tid * 4; i < TILE; i += THREADS * 4) {
                v4f v = {acc, acc + 1, acc + 2, (float)(p + tile)};
                st4_dev(mine + i, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int i = tid * 4; i < TILE; i += THREADS * 4)
                *reinterpret_cast<float4*>(mine + i) = make_float4(acc, acc + 1, acc + 2, (float)(p + tile));
        }
        if (a.mode == 0) __threadfence();   // every thread: release (agent scope) before the flag
        __syncthreads();                     // workgroup-scope release/acquire: all stores of the workgroup have been issued and acknowledged
        if (tid == 0) __hip_atomic_store(a.flags + tile, a.seq + (unsigned)p + 1u, a.mode >= 2 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // wait for the 8 neighbours
        if (tid < 8) {
            const int k = tid < 4 ? tid : tid + 1;
            const int ny = ty + k / 3 - 1, nx = tx + k % 3 - 1;
            if (ny >= 0 && ny < a.tiles_y && nx >= 0 && nx < a.tiles_x) {
                const unsigned* f = a.flags + img * a.per_img + ny * a.tiles_x + nx;
                const unsigned want = a.seq + (unsigned)p + 1u;
                int spins = 0;
                if (a.mode >= 2) {
                    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                        if (++spins > (1 << 22)) { bad = 1; break; }
                        if (a.mode == 2) __builtin_amdgcn_s_sleep(1);
                    }
                } else {
                    while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                        if (++spins > (1 << 22)) { bad = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
        }
        __syncthreads();
        if (bad) { if (tid == 0) atomicAdd(a.err, 1u); return; }
        if (a.mode == 0) __threadfence();   // acquire side in every thread; mode 1 relies on the pollers' acquire loads + the barrier
        // read the neighbours' border data (here: 1/8 of each neighbour tile)
        for (int k = 0; k < 9; ++k) {
            if (k == 4) continue;
            const int ny = ty + k / 3 - 1, nx = tx + k % 3 - 1;
            if (ny < 0 || ny >= a.tiles_y || nx < 0 || nx >= a.tiles_x) continue;
            const float* nb = a.buf[p & 1] + (size_t)(img * a.per_img + ny * a.tiles_x + nx) * TILE;
            for (int i = tid * 4; i < TILE / 8; i += THREADS * 4) {
                float4 v;
                if (a.mode >= 2) { const v4f q = ld4_dev(nb + i); v = make_float4(q.x, q.y, q.z, q.w); }
                else v = *reinterpret_cast<const float4*>(nb + i);
                acc += v.x + v.w;
                if (v.w != (float)(p + img * a.per_img + ny * a.tiles_x + nx)) bad = 2;   // stale data check
            }
        }
        __syncthreads();
        if (bad == 2) { if (tid == 0) atomicAdd(a.err, 1000u); bad = 0; }
    }
    if (tid == 0) a.sink[tile] = acc;
}

int main(int argc, char** argv) {
    const int imgs = 24, tiles_x = 2, tiles_y = 5;
    Args a{};
    a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.per_img = tiles_x * tiles_y; a.ntiles = imgs * a.per_img;
    CK(hipMalloc(&a.buf[0], (size_t)a.ntiles * TILE * 4));
    CK(hipMalloc(&a.buf[1], (size_t)a.ntiles * TILE * 4));
    CK(hipMalloc(&a.flags, a.ntiles * 4));
    CK(hipMalloc(&a.err, 4));
    CK(hipMalloc(&a.sink, a.ntiles * 4));
    CK(hipMemset(a.flags, 0, a.ntiles * 4));
    CK(hipMemset(a.err, 0, 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, xchg, THREADS, 0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d, max blocks/CU %d, cooperativeLaunch %d, grid %d\n", prop.multiProcessorCount, nb, prop.cooperativeLaunch, a.ntiles);
    unsigned seq = 0;
    for (int mode = 1; mode <= 3; ++mode)
    for (int coop = 0; coop <= 0; ++coop) {
        a.mode = mode;
        for (int work : {0, 2000}) {
            for (int phases : {0, 1, 2, 8}) {
                a.phases = phases; a.spin_work = work;
                float best = 1e9f, tot = 0.f;
                const int reps = 30;
                for (int r = 0; r < 3; ++r) {
                    CK(hipEventRecord(e0, st));
                    for (int i = 0; i < reps; ++i) {
                        a.seq = seq; seq += 64;
                        if (coop) {
                            void* params[] = {(void*)&a};
                            CK(hipLaunchCooperativeKernel((const void*)xchg, dim3(a.ntiles), dim3(THREADS), params, 0, st));
                        } else {
                            hipLaunchKernelGGL(xchg, dim3(a.ntiles), dim3(THREADS), 0, st, a);
                        }
                    }
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    tot = ms * 1e3f / reps;
                    if (tot < best) best = tot;
                }
                unsigned err = 0;
                CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
                printf("mode=%d coop=%d work=%d phases=%2d: %.2f us per launch (err %u)\n", mode, coop, work, phases, best, err);
                if (err) { CK(hipMemset(a.err, 0, 4)); }
            }
        }
    }
    return 0;
}

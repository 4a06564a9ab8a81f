// pk_fma_probe.hip — developer probe (not product): issue rate of v_pk_fma_f32 against v_fma_f32 on MI355X, with
// 2 wavefronts per SIMD (the resident kernel's occupancy) and 8, as independent accumulator chains of a given number.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/pk_fma_probe tools/probes/pk_fma_probe.hip && gpurun_out/pk_fma_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int CHAINS, bool PK>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float w) {
    v2f acc[CHAINS];
    v2f ww = {w, w * 0.5f};
    v2f dd = {(float)threadIdx.x, 1.f};
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = v2f{(float)c, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (PK) acc[c] = __builtin_elementwise_fma(ww, dd, acc[c]);
                else { acc[c].x = fmaf(ww.x, dd.x, acc[c].x); acc[c].y = fmaf(ww.y, dd.y, acc[c].y); }
            }
        asm volatile("" : "+v"(ww), "+v"(dd));
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, bool PK>
void run(float* out, int threads) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<CHAINS, PK><<<grid, threads>>>(out, 10, 1.0001f);
    hipEventRecord(e0);
    probe<CHAINS, PK><<<grid, threads>>>(out, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves = threads / 256; instructions per wave = iters * 8 * CHAINS * (PK ? 1 : 2)
    const double waves = threads / 256.0, instr = (double)iters * 8 * CHAINS * (PK ? 1 : 2);
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%s chains=%d threads=%4d: %.3f ms, %.2f cycles per instruction per SIMD, %.1f fma lanes/clk/CU\n", PK ? "pk_fma" : "fma   ",
           CHAINS, threads, ms, cycles / (instr * waves), (double)iters * 8 * CHAINS * 2 * threads / cycles);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    for (int threads : {256, 512, 1024}) {
        run<1, false>(out, threads); run<1, true>(out, threads);
        run<2, false>(out, threads); run<2, true>(out, threads);
        run<4, false>(out, threads); run<4, true>(out, threads);
        run<8, false>(out, threads); run<8, true>(out, threads);
    }
    return 0;
}

// valu_rate_probe.hip — developer probe (not product): issue cost of the VALU instructions the K x K resident kernel could be
// built from, per wavefront and SIMD, at the kernel's occupancies (512 / 768 / 1024 threads, one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/valu_rate_probe tools/probes/valu_rate_probe.hip && gpurun_out/valu_rate_probe
// Eight independent chains, 64 instructions per loop trip.  Printed: cycles per instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>

#include <cstdio>

enum Op { FMA_F32, FMA_MIX_HW, FMA_MIX_HH, DOT2_F32_F16, DOT2C_F32_F16, EXP_F32, ALIGNBIT, PERM, PK_FMA_F16, PK_MUL_F16, MIXLO_F16,
          CVT_F16_F32, PACK_B32_F16, DOT2_THEN_MIX, MOV, PK_MAX_F16, LOG_F32, RCP_F32, N_OPS };
const char* names[] = {"v_fma_f32", "v_fma_mix_f32(h,f,f)", "v_fma_mix_f32(h,h,f)", "v_dot2_f32_f16", "v_dot2c_f32_f16", "v_exp_f32", "v_alignbit_b32",
                       "v_perm_b32", "v_pk_fma_f16", "v_pk_mul_f16", "v_fma_mixlo_f16", "v_cvt_f16_f32", "v_pack_b32_f16", "10dot2+4mix+3align", "v_mov_b32",
                       "v_pk_max_f16", "v_log_f32", "v_rcp_f32"};

template <int OP>
__device__ __forceinline__ void one(float& acc, unsigned& u, unsigned a, unsigned b, float x) {
    if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(__uint_as_float(a)));
    if (OP == FMA_MIX_HW) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(a), "v"(x));
    if (OP == FMA_MIX_HH) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(a), "v"(b));
    if (OP == DOT2_F32_F16) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if (OP == DOT2C_F32_F16) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    if (OP == EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(acc));
    if (OP == LOG_F32) asm volatile("v_log_f32 %0, %0" : "+v"(acc));
    if (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc));
    if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %0, 16" : "+v"(u) : "v"(a));
    if (OP == PERM) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(u) : "v"(a), "v"(b));
    if (OP == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(u) : "v"(a), "v"(b));
    if (OP == PK_MUL_F16) asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(u) : "v"(a));
    if (OP == PK_MAX_F16) asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(u) : "v"(a));
    if (OP == MIXLO_F16) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u) : "v"(acc), "v"(x));
    if (OP == CVT_F16_F32) asm volatile("v_cvt_f16_f32 %0, %1" : "+v"(u) : "v"(acc));
    if (OP == PACK_B32_F16) asm volatile("v_pack_b32_f16 %0, %1, %0" : "+v"(u) : "v"(a));
    if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(u) : "v"(a));
}

template <int OP>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, unsigned a0, unsigned b0, float x0) {
    float acc[8];
    unsigned u[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { acc[c] = (float)(c + threadIdx.x) * 1e-3f; u[c] = a0 + c; }
    unsigned a = a0, b = b0;
    float x = x0;
    for (int i = 0; i < iters; ++i) {
        if (OP == DOT2_THEN_MIX) {
            // the instruction mix of one pixel-step of a 5 x 5 stencil on packed fp16 state: 10 dot2 + 4 mixed FMAs per pixel, 25 alignbits per 8 pixels
#pragma unroll
            for (int px = 0; px < 8; ++px) {
#pragma unroll
                for (int k = 0; k < 10; ++k) one<DOT2_F32_F16>(acc[px], u[px], a, b, x);
#pragma unroll
                for (int k = 0; k < 4; ++k) one<FMA_MIX_HH>(acc[px], u[px], a, b, x);
#pragma unroll
                for (int k = 0; k < 3; ++k) one<ALIGNBIT>(acc[px], u[px], a, b, x);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) one<OP>(acc[c], u[c], a, b, x);
        }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(x));
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += acc[c] + __uint_as_float(u[c]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(float* out, int threads) {
    const int iters = 1000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<OP><<<grid, threads>>>(out, 10, 0x3c003c00u, 0x38003800u, 0.5f);
    hipEventRecord(e0);
    probe<OP><<<grid, threads>>>(out, iters, 0x3c003c00u, 0x38003800u, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = threads / 256.0, instr = (double)iters * (OP == DOT2_THEN_MIX ? 8 * 17 : 64);
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-24s threads=%4d: %.3f ms, %.2f cycles per instruction per SIMD\n", names[OP], threads, ms, cycles / (instr * waves));
}

template <int OP>
void run_all(float* out) {
    for (int threads : {256, 512, 768, 1024}) run<OP>(out, threads);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    run_all<FMA_F32>(out); run_all<FMA_MIX_HW>(out); run_all<FMA_MIX_HH>(out); run_all<DOT2_F32_F16>(out); run_all<DOT2C_F32_F16>(out);
    run_all<EXP_F32>(out); run_all<LOG_F32>(out); run_all<RCP_F32>(out); run_all<ALIGNBIT>(out); run_all<PERM>(out); run_all<PK_FMA_F16>(out); run_all<PK_MUL_F16>(out);
    run_all<PK_MAX_F16>(out); run_all<MIXLO_F16>(out); run_all<CVT_F16_F32>(out); run_all<PACK_B32_F16>(out); run_all<MOV>(out); run_all<DOT2_THEN_MIX>(out);
    return 0;
}

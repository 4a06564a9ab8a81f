// c_host_demo.cpp — the engine driven from plain C++ through the C ABI only (no PyTorch, no Python):
//   hipMalloc / hipMemcpy, cspn3_propagate_from_guidance (the default inference path) and, for comparison, the
//   two-call form cspn3_prepare + cspn_propagate; both results must be bit-identical.  Where a tiling exists, also the
//   weight-resident launch (cspn3_forward_resident: the schedule the Python host runs by default) with its whole host
//   protocol — plan, zero-initialised workspace, growing sequence number, pinned error word — again bit-identical.
//
//   g++ -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/c_host_demo.cpp \
//       -L cspn_monodepth_amd -lcspn_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/cspn_monodepth_amd -o c_host_demo
//   ./c_host_demo in.bin out.bin B C H W T [sparse]
// in.bin : float32 guidance [B,C,H,W], blur [B,1,H,W], (sparse [B,1,H,W])      out.bin : float32 refined [B,1,H,W]
// tests/test_c_host.py builds and runs this on the GPU box and checks out.bin against the oracle.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cspn_hip.h"

#define CHECK_HIP(x)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define CHECK_CSPN(x)                                                                      \
    do {                                                                                   \
        if (!(x)) { std::fprintf(stderr, "%s failed: %s\n", #x, cspn_last_error()); return 3; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: %s in.bin out.bin B C H W T [sparse]\n", argv[0]); return 1; }
    const int B = std::atoi(argv[3]), C = std::atoi(argv[4]), H = std::atoi(argv[5]), W = std::atoi(argv[6]), T = std::atoi(argv[7]);
    const bool sparse = argc > 8 && std::atoi(argv[8]) != 0;
    if (cspn_abi_version() != CSPN_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    const size_t plane = (size_t)B * H * W, ng = plane * C;
    std::vector<float> h(ng + plane * (sparse ? 2 : 1));
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(h.data(), sizeof(float), h.size(), f) != h.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    std::fclose(f);

    float *g, *d0, *sp = nullptr, *w8, *out_a, *out_b, *work;
    CHECK_HIP(hipMalloc((void**)&g, ng * 4));
    CHECK_HIP(hipMalloc((void**)&d0, plane * 4));
    CHECK_HIP(hipMalloc((void**)&w8, plane * 8 * 4));
    CHECK_HIP(hipMalloc((void**)&out_a, plane * 4));
    CHECK_HIP(hipMalloc((void**)&out_b, plane * 4));
    const size_t wbytes = cspn_propagate_workspace_bytes(B, H, W, T, CSPN_F32, 0);
    CHECK_HIP(hipMalloc((void**)&work, wbytes ? wbytes : 16));
    CHECK_HIP(hipMemcpy(g, h.data(), ng * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d0, h.data() + ng, plane * 4, hipMemcpyHostToDevice));
    if (sparse) {
        CHECK_HIP(hipMalloc((void**)&sp, plane * 4));
        CHECK_HIP(hipMemcpy(sp, h.data() + ng + plane, plane * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    const int blend = sparse ? CSPN_BLEND_SPARSE : CSPN_BLEND_NONE;

    cspn_plan plan;
    CHECK_CSPN(cspn_plan_resolve(3, B, H, W, T, 0, nullptr, &plan));
    std::printf("plan: S=%d tile=%dx%d NQ=%d threads=%d scalar=%d\n", plan.steps_per_launch, plan.tile_w, plan.tile_h,
                plan.quads_per_thread, plan.threads, plan.force_scalar);
    // (a) one call: weights derived in the first launch (needs W % 4 == 0); (b) prepare + propagate
    const bool fused = W % 4 == 0;
    if (fused)
        CHECK_CSPN(cspn3_propagate_from_guidance(g, CSPN_F32, (long)C * H * W, (long)H * W, w8, nullptr, d0, sp, out_a, nullptr,
                                                 work, CSPN_F32, B, H, W, 0, T, blend, nullptr, nullptr, 0, nullptr, st));
    CHECK_CSPN(cspn3_prepare(g, CSPN_F32, (long)C * H * W, (long)H * W, B, H, W, 0, w8, CSPN_F32, nullptr, st));
    CHECK_CSPN(cspn_propagate(w8, CSPN_F32, d0, sp, out_b, nullptr, work, CSPN_F32, B, H, W, 0, 3, T, blend, nullptr, st));
    CHECK_HIP(hipStreamSynchronize(st));

    std::vector<float> ra(plane), rb(plane);
    CHECK_HIP(hipMemcpy(rb.data(), out_b, plane * 4, hipMemcpyDeviceToHost));
    if (fused) {
        CHECK_HIP(hipMemcpy(ra.data(), out_a, plane * 4, hipMemcpyDeviceToHost));
        if (std::memcmp(ra.data(), rb.data(), plane * 4) != 0) { std::fprintf(stderr, "one-call and two-call results differ\n"); return 4; }
    }
    // (c) the weight-resident launch: one launch per chunk of whole images, weights in registers for all T steps
    bool resident = false;
    cspn_resident_plan rp;
    std::memset(&rp, 0, sizeof rp);
    if (fused && cspn3_resident_plan(B, H, W, T, sparse ? 1 : 0, 0, &rp)) {
        void* rwork;
        unsigned* host_err;                                       // two pinned words the device writes: [0] time-out, [1] completion
        float* out_c;
        const size_t rbytes = cspn3_resident_workspace_bytes(B, H, W);
        CHECK_HIP(hipMalloc(&rwork, rbytes));
        CHECK_HIP(hipMemset(rwork, 0, rbytes));                   // once; afterwards only the sequence number grows
        CHECK_HIP(hipHostMalloc((void**)&host_err, 2 * sizeof(unsigned), hipHostMallocMapped));
        host_err[0] = host_err[1] = 0;
        CHECK_HIP(hipMalloc((void**)&out_c, plane * 4));
        unsigned seq = 256;
        for (int call = 0; call < 2; ++call, seq += 256) {        // twice: the second call reuses the workspace of the first
            CHECK_HIP(hipMemsetAsync(out_c, 0xff, plane * 4, st));
            CHECK_CSPN(cspn3_forward_resident(g, (long)C * H * W, (long)H * W, d0, sp, out_c, nullptr, nullptr, nullptr, rwork, seq,
                                              host_err, B, H, W, 0, T, blend, nullptr, nullptr, 0, &rp, st));
            CHECK_HIP(hipStreamSynchronize(st));
            if (host_err[0] != 0) { std::fprintf(stderr, "resident launch timed out (device shared?)\n"); return 5; }
            std::vector<float> rc(plane);
            CHECK_HIP(hipMemcpy(rc.data(), out_c, plane * 4, hipMemcpyDeviceToHost));
            if (std::memcmp(rc.data(), rb.data(), plane * 4) != 0) { std::fprintf(stderr, "resident and multi-launch results differ (call %d)\n", call); return 4; }
        }
        resident = true;
        std::printf("resident plan: %d-step phases, %dx%d tiles of %dx%d, %d quads/thread, %d image(s) per launch, %d launch(es)\n",
                    rp.steps_per_phase, rp.tiles_x, rp.tiles_y, rp.tile_w, rp.tile_h, rp.quads_per_thread, rp.images_per_launch, rp.launches);
    }
    f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(rb.data(), sizeof(float), plane, f) != plane) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    std::fclose(f);
    std::printf("ok: %zu pixels, %d steps%s%s\n", plane, T, fused ? ", one-call == two-call bit for bit" : "",
                resident ? ", resident == multi-launch bit for bit" : "");
    return 0;
}

// cspn_unpool.hip — zero-insertion un-pooling of the UNet decoders.
//
//   out[p, s*h, s*w] = in[p, h, w], 0 elsewhere, cropped to oH x oW      network/unet_ours.py:138-150 (grouped
//   conv_transpose2d with a one-hot s x s weight + crop) and network/unet_cspn_nyu.py:202-213 (nearest upsample times a
//   checkerboard mask the reference builds with an O(H*W) Python double loop on every call).
// One streaming pass: each thread writes a quad of the output (all of it, zeros included, so no memset) and the
// backward is the strided gather (it does not reproduce 0 * NaN from non-finite gradients at dropped positions).  HBM-bound: (1 + s*s) elements per input pixel forward.
#include "cspn_common.hpp"

namespace {

// Generic: any scale / width.  Thread = output quad; the index runs over (plane, row, quad) flattened so that small
// planes (15 x 19 at the first decoder stage) still fill 256-thread groups.
template <typename T>
__global__ __launch_bounds__(256) void unpool_fwd(const T* __restrict__ in, T* __restrict__ out, int H, int W, int s,
                                                  int oH, int oW, int oWQ, unsigned total, int vec) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= total) return;
    const unsigned per_plane = (unsigned)oH * oWQ;
    const size_t p = q / per_plane;
    const unsigned r = q - (unsigned)p * per_plane;
    const int y = r / oWQ, x0 = (r - y * oWQ) * 4;
    // Inserted positions hold in * 0, not a literal 0: both reference formulations multiply (one-hot conv_transpose
    // weight / checkerboard mask), so a non-finite activation turns its whole s x s block into NaN there.
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const T* row = in + (p * H + y / s) * (size_t)W;
    const float rowgate = y % s == 0 ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int x = x0 + e;
        if (x < oW) v[e] = ld1(row + x / s) * (x % s == 0 ? rowgate : 0.f);      // x / s < W because oW <= s * W
    }
    T* dst = out + (p * oH + y) * (size_t)oW + x0;
    if (vec) {
        st4(dst, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x0 + e < oW) st1(dst + e, v[e]);
    }
}

// scale 2, oW % 4 == 0 (every decoder stage from 57 x 76 up): thread = two input pixels -> the output quad they
// expand to on rows 2h and 2h+1.  Two scalar loads, two 16-byte stores, no per-element division.
template <typename T>
__global__ __launch_bounds__(256) void unpool_fwd_s2(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                     int oH, int oW, int oWQ, unsigned total) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= total) return;
    const int rows = (oH + 1) >> 1;
    const unsigned per_plane = (unsigned)rows * oWQ;
    const size_t p = q / per_plane;
    const unsigned r = q - (unsigned)p * per_plane;
    const int h = r / oWQ, k = r - h * oWQ;
    const T* src = in + (p * H + h) * (size_t)W + 2 * k;
    const float a = ld1(src), b = ld1(src + 1);              // 2k+1 < W because 4k+3 < oW <= 2W
    const float az = a * 0.f, bz = b * 0.f;                   // x * 0: NaN / inf fill their 2 x 2 block, as the reference
    T* dst = out + (p * oH + 2 * h) * (size_t)oW + 4 * k;
    st4(dst, make_float4(a, az, b, bz));
    if (2 * h + 1 < oH) st4(dst + oW, make_float4(az, az, bz, bz));
}

template <typename T>
__global__ __launch_bounds__(256) void unpool_bwd(const T* __restrict__ gout, T* __restrict__ gin, int H, int W, int s,
                                                  int oH, int oW, unsigned total) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned hw = (unsigned)H * W;
    const size_t p = i / hw;
    const unsigned r = i - (unsigned)p * hw;
    const int h = r / W, w = r - h * W;
    const int y = h * s, x = w * s;
    st1(gin + i, (y < oH && x < oW) ? ld1(gout + (p * oH + y) * (size_t)oW + x) : 0.f);
}

int check(const char* who, int dtype, long planes, int H, int W, int s, int oH, int oW) {
    if (dtype != CSPN_F32 && dtype != CSPN_F16) return fail("%s: dtype must be CSPN_F32 or CSPN_F16", who);
    if (planes < 1 || H < 1 || W < 1) return fail("%s: bad shape (planes=%ld, H=%d, W=%d)", who, planes, H, W);
    if (s < 1) return fail("%s: scale must be >= 1", who);
    if (oH < 1 || oW < 1 || (long)oH > (long)s * H || (long)oW > (long)s * W)
        return fail("%s: output %dx%d must lie in [1, scale*H] x [1, scale*W] = %ldx%ld", who, oH, oW, (long)s * H, (long)s * W);
    if ((double)planes * oH * ((oW + 3) / 4) >= 2147483648.0 || (double)planes * H * W >= 2147483648.0)
        return fail("%s: more than 2^31 quads / pixels in one call; split the batch", who);
    return 1;
}

}  // namespace

extern "C" {

int cspn_unpool2d(const void* input, void* out, int dtype, long planes, int H, int W, int scale, int oH, int oW,
                  cspn_stream_t stream) {
    if (!check("cspn_unpool2d", dtype, planes, H, W, scale, oH, oW)) return 0;
    if (!input || !out) return fail("cspn_unpool2d: null pointer");
    const int oWQ = ceil_div(oW, 4);
    const bool aligned = (reinterpret_cast<uintptr_t>(out) & (dtype == CSPN_F16 ? 7 : 15)) == 0;
    const int vec = oW % 4 == 0 && aligned;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (scale == 2 && vec) {
        const unsigned total = (unsigned)(planes * ((oH + 1) / 2) * oWQ);
        const dim3 grid((total + 255u) / 256u), block(256);
        if (dtype == CSPN_F16)
            CSPN_PRE(st), unpool_fwd_s2<__half><<<grid, block, 0, st>>>(static_cast<const __half*>(input), static_cast<__half*>(out), H, W, oH, oW, oWQ, total);
        else
            CSPN_PRE(st), unpool_fwd_s2<float><<<grid, block, 0, st>>>(static_cast<const float*>(input), static_cast<float*>(out), H, W, oH, oW, oWQ, total);
    } else {
        const unsigned total = (unsigned)(planes * oH * oWQ);
        const dim3 grid((total + 255u) / 256u), block(256);
        if (dtype == CSPN_F16)
            CSPN_PRE(st), unpool_fwd<__half><<<grid, block, 0, st>>>(static_cast<const __half*>(input), static_cast<__half*>(out), H, W, scale, oH, oW, oWQ, total, vec);
        else
            CSPN_PRE(st), unpool_fwd<float><<<grid, block, 0, st>>>(static_cast<const float*>(input), static_cast<float*>(out), H, W, scale, oH, oW, oWQ, total, vec);
    }
    HIP_OK(hipGetLastError());
    return 1;
}

int cspn_unpool2d_backward(const void* grad_out, void* grad_input, int dtype, long planes, int H, int W, int scale,
                           int oH, int oW, cspn_stream_t stream) {
    if (!check("cspn_unpool2d_backward", dtype, planes, H, W, scale, oH, oW)) return 0;
    if (!grad_out || !grad_input) return fail("cspn_unpool2d_backward: null pointer");
    const unsigned total = (unsigned)(planes * H * W);
    const dim3 grid((total + 255u) / 256u), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == CSPN_F16)
        CSPN_PRE(st), unpool_bwd<__half><<<grid, block, 0, st>>>(static_cast<const __half*>(grad_out), static_cast<__half*>(grad_input), H, W, scale, oH, oW, total);
    else
        CSPN_PRE(st), unpool_bwd<float><<<grid, block, 0, st>>>(static_cast<const float*>(grad_out), static_cast<float*>(grad_input), H, W, scale, oH, oW, total);
    HIP_OK(hipGetLastError());
    return 1;
}

}  // extern "C"

// Device side of load_and_preprocess_images (vggt/load.py:38-183): the reference resizes every
// frame with PIL's Image.resize(..., BICUBIC) on the host.  Pillow's 8-bit resampler
// (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
// ImagingResampleHorizontal_8bpc / Vertical_8bpc) is two separable integer passes with 22-bit
// fixed-point coefficients and a uint8 intermediate; the coefficient tables are built on the host
// exactly as Pillow builds them (preprocess.py), these kernels apply them, so the result is
// bit-identical to PIL's.
#include "common.h"
#include "kernels.h"

namespace skimi {

static inline int grid_for(long n, int per_block, int cap) {
    long b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// One pass along an axis of length n_in -> n_out.  Element (o, n, i) of the input lives at
// (o * n_in + n) * inner + i: horizontal pass of an HWC image = (outer H, axis W, inner C),
// vertical pass = (outer 1, axis H, inner W*C).
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ in,
                                                          unsigned char* __restrict__ out, long outer, int n_in,
                                                          int n_out, long inner, const int* __restrict__ kk,
                                                          const int* __restrict__ bounds, int ksize) {
    const long total = outer * n_out * inner;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long i = idx % inner;
        const long r = idx / inner;
        const int xx = (int)(r % n_out);
        const long o = r / n_out;
        const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
        const int* k = kk + (long)xx * ksize;
        int ss = 1 << 21;   // 1 << (PRECISION_BITS - 1), PRECISION_BITS = 32 - 8 - 2
        const unsigned char* p = in + (o * n_in + xmin) * inner + i;
        for (int x = 0; x < xmax; ++x) ss += (int)p[(long)x * inner] * k[x];
        ss >>= 22;   // arithmetic shift, then clip8
        out[idx] = (unsigned char)min(max(ss, 0), 255);
    }
}

int resample_u8_launch(const unsigned char* in, unsigned char* out, long outer, int n_in, int n_out, long inner,
                       const int* kk, const int* bounds, int ksize, hipStream_t st) {
    SKIMI_CHECK_ARG(in && out && kk && bounds && outer > 0 && n_in > 0 && n_out > 0 && inner > 0 && ksize > 0,
                    "skimi_resample_u8: bad arguments");
    const long total = outer * n_out * inner;
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st, in, out, outer, n_in, n_out,
                       inner, kk, bounds, ksize);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// uint8 HWC (3 channels) -> fp32 CHW / 255 placed into an [3, OH, OW] frame: output pixel (y, x)
// takes input pixel (y + y_off, x + x_off) (centre crop: positive offsets; padding: negative
// offsets, pixels outside the input get `fill`) -- TF.ToTensor + crop / white pad of load.py
__global__ __launch_bounds__(256) void u8_hwc_to_f32_chw_kernel(const unsigned char* __restrict__ in, int H, int W,
                                                                float* __restrict__ out, int OH, int OW, int y_off,
                                                                int x_off, float fill) {
    const int total = 3 * OH * OW;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int x = idx % OW;
        const int y = (idx / OW) % OH;
        const int c = idx / (OW * OH);
        const int sy = y + y_off, sx = x + x_off;
        float v = fill;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = (float)in[((long)sy * W + sx) * 3 + c] / 255.0f;
        out[idx] = v;
    }
}

int u8_hwc_to_f32_chw_launch(const unsigned char* in, int H, int W, float* out, int OH, int OW, int y_off, int x_off,
                             float fill, hipStream_t st) {
    SKIMI_CHECK_ARG(in && out && H > 0 && W > 0 && OH > 0 && OW > 0, "skimi_u8_hwc_to_f32_chw: bad arguments");
    hipLaunchKernelGGL(u8_hwc_to_f32_chw_kernel, dim3(grid_for(3L * OH * OW, 256, 4096)), dim3(256), 0, st, in, H, W, out, OH,
                       OW, y_off, x_off, fill);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

// Small kernels of the VGGT track head (vggt/vggt/heads/track_modules/*): feature pyramid,
// bilinear feature sampling, fused correlation sampling, transformer-input assembly, coordinate
// update.  All fp32, channels-last; one wave per output vector where a channel reduction is
// needed, one thread per element otherwise.
#include <algorithm>

#include "common.h"
#include "track_kernels.h"

namespace skimi {

static inline int grid_for(long n, int per_block = 256, int cap = 8192) {
    return (int)std::max<long>(1, std::min<long>(cdiv(n, per_block), cap));
}

// F.avg_pool2d(k=2, s=2) on NHWC (track_modules/blocks.py:158-166)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int N,
                                                       int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % Wo);
        const int y = (int)((i / ((long)C * Wo)) % Ho);
        const long n = i / ((long)C * Wo * Ho);
        const float* p = in + ((n * H + 2 * y) * (long)W + 2 * x) * C + c;
        out[i] = (p[0] + p[C] + p[(long)W * C] + p[(long)W * C + C]) * 0.25f;
    }
}
int avgpool2_launch(const float* in, float* out, int N, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for((long)N * (H / 2) * (W / 2) * C)), dim3(256), 0, st, in, out, N, H, W, C);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// sample_features4d (track_modules/utils.py:198-223): bilinear, align_corners=True, border
// padding, at pixel coords (x, y).  fmap: image `img_stride`-strided [B][H,W,C]; coords [B,N,2];
// out [B,N,C].
__global__ __launch_bounds__(256) void sample_border_kernel(const float* __restrict__ fmap, long img_stride,
                                                            const float* __restrict__ coords, long coord_stride,
                                                            float* __restrict__ out, int B, int N, int H, int W, int C) {
    const long total = (long)B * N * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int n = (int)((i / C) % N);
        const long b = i / ((long)C * N);
        float x = coords[(b * N + n) * coord_stride], y = coords[(b * N + n) * coord_stride + 1];
        // grid_sample(padding_mode="border", align_corners=True): clip the coordinate to [0, size-1]
        x = fminf(fmaxf(x, 0.f), (float)(W - 1));
        y = fminf(fmaxf(y, 0.f), (float)(H - 1));
        const int x0 = (int)floorf(x), y0 = (int)floorf(y);
        const float lx = x - x0, ly = y - y0;
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float* f = fmap + b * img_stride + c;
        const float v00 = f[((long)y0 * W + x0) * C], v01 = f[((long)y0 * W + x1) * C];
        const float v10 = f[((long)y1 * W + x0) * C], v11 = f[((long)y1 * W + x1) * C];
        out[i] = v00 * (1 - lx) * (1 - ly) + v01 * lx * (1 - ly) + v10 * (1 - lx) * ly + v11 * lx * ly;
    }
}
int sample_border_launch(const float* fmap, long img_stride, const float* coords, long coord_stride, float* out, int B,
                         int N, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(sample_border_kernel, dim3(grid_for((long)B * N * C)), dim3(256), 0, st, fmap, img_stride, coords,
                       coord_stride, out, B, N, H, W, C);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// Fused CorrBlock.corr_sample for one pyramid level (track_modules/blocks.py:171-229):
//   corr[hw] = <target, fmap[hw]> / sqrt(C); sampled bilinearly (zeros padding, align_corners)
//   on the (2r+1)^2 grid around coords / 2^level.
// Bilinear sampling is linear, so sample(corr) = <target, sample(fmap)> / sqrt(C): the
// correlation volume is never materialised.  One wave per (track row, sample); lanes over C.
// Quirk kept: delta = stack(meshgrid(dy, dx, "ij")) is added to (x, y), i.e. the ROW index i
// of the (2r+1)^2 grid offsets x and the column index j offsets y.
//   targets [rows, C] with rows ordered (b, n, s); fmap [B*S, H, W, C]; coords [rows, 2];
//   out[row, out_off + i*(2r+1) + j], row stride ldo.
__global__ __launch_bounds__(256) void corr_sample_kernel(const float* __restrict__ tgt, const float* __restrict__ fmap,
                                                          const float* __restrict__ coords, float* __restrict__ out,
                                                          long rows, int N, int S, int H, int W, int C, int r,
                                                          float inv_scale, float inv_sqrt_c, long ldo, int out_off) {
    const int lane = threadIdx.x & 63;
    const int side = 2 * r + 1, ns = side * side;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= rows * ns) return;
    const long row = wv / ns;
    const int sidx = (int)(wv - row * ns);
    const int i = sidx / side, j = sidx - i * side;
    const int s = (int)(row % S);
    const long bn = row / S;
    const long b = bn / N;
    const float x = coords[row * 2] * inv_scale + (float)(i - r);
    const float y = coords[row * 2 + 1] * inv_scale + (float)(j - r);
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float lx = x - x0f, ly = y - y0f;
    const float* img = fmap + (b * S + s) * (long)H * W * C;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
        float v = 0.f;
        if (y0 >= 0 && y0 < H) {
            if (x0 >= 0 && x0 < W) v += img[((long)y0 * W + x0) * C + c] * (1 - lx) * (1 - ly);
            if (x0 + 1 >= 0 && x0 + 1 < W) v += img[((long)y0 * W + x0 + 1) * C + c] * lx * (1 - ly);
        }
        if (y0 + 1 >= 0 && y0 + 1 < H) {
            if (x0 >= 0 && x0 < W) v += img[((long)(y0 + 1) * W + x0) * C + c] * (1 - lx) * ly;
            if (x0 + 1 >= 0 && x0 + 1 < W) v += img[((long)(y0 + 1) * W + x0 + 1) * C + c] * lx * ly;
        }
        acc += v * tgt[row * C + c];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row * ldo + out_off + sidx] = acc * inv_sqrt_c;
}
int corr_sample_launch(const float* tgt, const float* fmap, const float* coords, float* out, long rows, int N, int S,
                       int H, int W, int C, int r, int level, long ldo, int out_off, hipStream_t st) {
    const int ns = (2 * r + 1) * (2 * r + 1);
    const long waves = rows * ns;
    hipLaunchKernelGGL(corr_sample_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, tgt, fmap, coords, out, rows,
                       N, S, H, W, C, r, 1.0f / (float)(1 << level), 1.0f / sqrtf((float)C), ldo, out_off);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// get_2d_sincos_pos_embed (track_modules/utils.py:18-87) sampled at the query points
// (sample_features4d, border): pe[b, n, :] for coords[b, n] = (x, y) in feature-map pixels.
// Table values are computed in float64 and rounded to float as the reference does, then
// interpolated in fp32.  First D/2 channels encode x, last D/2 encode y; each half is
// [sin(p*w_0..), cos(p*w_0..)] with w_k = 1/10000^(k/(D/4)).
__global__ __launch_bounds__(256) void pos_embed_sample_kernel(const float* __restrict__ coords, long coord_stride,
                                                               float* __restrict__ out, int BN, int H, int W, int D) {
    const long total = (long)BN * D;
    const int half = D / 2, quarter = D / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D);
        const long bn = i / D;
        float x = coords[bn * coord_stride], y = coords[bn * coord_stride + 1];
        x = fminf(fmaxf(x, 0.f), (float)(W - 1));
        y = fminf(fmaxf(y, 0.f), (float)(H - 1));
        const bool is_x = c < half;
        const int cc = is_x ? c : c - half;
        const bool is_sin = cc < quarter;
        const int k = is_sin ? cc : cc - quarter;
        const double omega = 1.0 / pow(10000.0, (double)k / (double)quarter);
        const float p = is_x ? x : y;
        const int lim = is_x ? W : H;
        const int p0 = (int)floorf(p);
        const int p1 = min(p0 + 1, lim - 1);
        const float l = p - p0;
        const double a0 = (double)p0 * omega, a1 = (double)p1 * omega;
        const float v0 = (float)(is_sin ? sin(a0) : cos(a0));
        const float v1 = (float)(is_sin ? sin(a1) : cos(a1));
        // the other axis' two neighbours carry the same value: their weights sum to 1
        out[i] = v0 * (1 - l) + v1 * l;
    }
}
int pos_embed_sample_launch(const float* coords, long coord_stride, float* out, int BN, int H, int W, int D,
                            hipStream_t st) {
    hipLaunchKernelGGL(pos_embed_sample_kernel, dim3(grid_for((long)BN * D)), dim3(256), 0, st, coords, coord_stride, out,
                       BN, H, W, D);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// transformer input of one refinement iteration (base_track_predictor.py:139-169):
//   x[row] = [emb2d(flow, L/2) | flow/ms | flow/ms | corr_feat | track_feat] + pos_emb[b,n] + qrt[s == 0 ? 0 : 1]
// rows ordered (b, n, s); x has leading dim ldx (zero padded beyond 3L+4).
__global__ __launch_bounds__(256) void track_input_kernel(const float* __restrict__ coords, const float* __restrict__ fcorr,
                                                          const float* __restrict__ tfeat, const float* __restrict__ pos,
                                                          const float* __restrict__ qrt, float* __restrict__ x, long rows,
                                                          int S, int L, long ldx, float max_scale) {
    const int D = 3 * L + 4, E = L / 2;
    const long total = rows * ldx;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % ldx);
        const long row = i / ldx;
        if (c >= D) { x[i] = 0.f; continue; }
        const int s = (int)(row % S);
        const long bn = row / S;
        const float fx = coords[row * 2] - coords[(bn * S) * 2];
        const float fy = coords[row * 2 + 1] - coords[(bn * S) * 2 + 1];
        float v;
        if (c < L) {
            // get_2d_embedding(flows, E, cat_coords=False): [pe_x (E) | pe_y (E)], sin at even / cos at odd
            const bool isx = c < E;
            const int cc = isx ? c : c - E;
            const float div = (float)(cc & ~1) * (1000.0f / (float)E);
            const float a = (isx ? fx : fy) * div;
            v = (cc & 1) ? cosf(a) : sinf(a);
        } else if (c < L + 4) {
            const int k = c - L;
            v = ((k & 1) ? fy : fx) / max_scale;
        } else if (c < 2 * L + 4) {
            v = fcorr[row * L + (c - L - 4)];
        } else {
            v = tfeat[row * L + (c - 2 * L - 4)];
        }
        x[i] = v + pos[bn * D + c] + qrt[(s == 0 ? 0 : 1) * D + c];
    }
}
int track_input_launch(const float* coords, const float* fcorr, const float* tfeat, const float* pos, const float* qrt,
                       float* x, long rows, int S, int L, long ldx, float max_scale, hipStream_t st) {
    hipLaunchKernelGGL(track_input_kernel, dim3(grid_for(rows * ldx)), dim3(256), 0, st, coords, fcorr, tfeat, pos, qrt, x,
                       rows, S, L, ldx, max_scale);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// coords += delta[:, :2]; coords[s == 0] = query (base_track_predictor.py:182-187)
// delta [rows, ldd] (rows (b,n,s)); optional export pred[b, s, n, :] = coords * stride
__global__ void track_coord_update_kernel(float* __restrict__ coords, const float* __restrict__ delta, long ldd,
                                          const float* __restrict__ query, float* __restrict__ pred, long rows, int N,
                                          int S, float stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 2) return;
    const long row = i >> 1;
    const int k = (int)(i & 1);
    const int s = (int)(row % S);
    const long bn = row / S;
    float v = coords[i] + delta[row * ldd + k];
    if (s == 0) v = query[bn * 2 + k];
    coords[i] = v;
    if (pred) {
        const long b = bn / N;
        const int n = (int)(bn - b * N);
        pred[((b * S + s) * N + n) * 2 + k] = v * stride;
    }
}
int track_coord_update_launch(float* coords, const float* delta, long ldd, const float* query, float* pred, long rows,
                              int N, int S, float stride, hipStream_t st) {
    hipLaunchKernelGGL(track_coord_update_kernel, dim3((unsigned)cdiv(rows * 2, 128)), dim3(128), 0, st, coords, delta, ldd,
                       query, pred, rows, N, S, stride);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// init: coords[b,n,s] = query[b,n] / stride (all s); query_scaled[b,n] = query / stride
__global__ void track_init_kernel(const float* __restrict__ q, float* __restrict__ coords, float* __restrict__ qs,
                                  long BN, int S, float inv_stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BN * S * 2) return;
    const int k = (int)(i & 1);
    const long bn = (i >> 1) / S;
    const float v = q[bn * 2 + k] * inv_stride;
    coords[i] = v;
    if (((i >> 1) % S) == 0) qs[bn * 2 + k] = v;
}
int track_init_launch(const float* q, float* coords, float* qs, long BN, int S, float stride, hipStream_t st) {
    hipLaunchKernelGGL(track_init_kernel, dim3((unsigned)cdiv(BN * S * 2, 128)), dim3(128), 0, st, q, coords, qs, BN, S,
                       1.0f / stride);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// dst[b, n, s, :] = src[b, n, :]  (track_feats init: query feature repeated over S)
__global__ void repeat_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long BN, int S, int C) {
    const long total = BN * S * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long bn = i / ((long)C * S);
        dst[i] = src[bn * C + c];
    }
}
int repeat_rows_launch(const float* src, float* dst, long BN, int S, int C, hipStream_t st) {
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(grid_for(BN * S * C)), dim3(256), 0, st, src, dst, BN, S, C);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// out[b, s, n] = in[(b, n, s)]  (vis / conf export)
__global__ void bns_to_bsn_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int N, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * N * S) return;
    const int s = (int)(i % S);
    const int n = (int)((i / S) % N);
    const long b = i / ((long)S * N);
    out[(b * S + s) * N + n] = in[i];
}
int bns_to_bsn_launch(const float* in, float* out, int B, int N, int S, hipStream_t st) {
    hipLaunchKernelGGL(bns_to_bsn_kernel, dim3((unsigned)cdiv((long)B * N * S, 128)), dim3(128), 0, st, in, out, B, N, S);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

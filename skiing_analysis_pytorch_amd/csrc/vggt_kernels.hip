// HBM-bound helper kernels of the VGGT forward: patch gather, token assembly, bilinear resize,
// positional embeddings, AdaLN modulation, head activations, weight repacking.
// All are channels-last, one thread per 4 contiguous channels (16-B fp32 / 8-B bf16 accesses)
// or one wave per row; grids are capped and grid-strided.
#include <algorithm>

#include "common.h"
#include "vggt_kernels.h"

namespace skimi {

static inline int grid_for(long n, int per_block = 256, int cap = 8192) {
    return (int)std::max<long>(1, std::min<long>(cdiv(n, per_block), cap));
}

// ---------------------------------------------------------------------------------------
// patch gather + ImageNet normalisation:  images [F,3,H,W] -> A [F*ph*pw, Kp]
//   A[(f,py,px)][c*p*p + dy*p + dx] = (img[f,c,py*p+dy,px*p+dx] - mean[c]) / std[c]
// (vggt/vggt/models/aggregator.py:201 + the im2col of patch_embed.py:62,72-74's Conv2d k=s=14;
//  column order = Conv2d weight.flatten(1) order, zero padded to Kp)
// ---------------------------------------------------------------------------------------
template <typename T, bool F16 = false>
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ img, T* __restrict__ out,
                                                           int F, int H, int W, int p, int Kp) {
    const int ph = H / p, pw = W / p, K = 3 * p * p;
    const long total = (long)F * ph * pw * Kp;
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const long r = i / Kp;
        float v = 0.f;
        if (k < K) {
            const int px = (int)(r % pw);
            const int py = (int)((r / pw) % ph);
            const long f = r / ((long)pw * ph);
            const int c = k / (p * p);
            const int rem = k - c * p * p;
            const int dy = rem / p, dx = rem - dy * p;
            const float x = img[((f * 3 + c) * H + (py * p + dy)) * (long)W + (px * p + dx)];
            v = (x - mean[c]) / stdv[c];
        }
        if (sizeof(T) == 4) ((float*)out)[i] = v;
        else ((unsigned short*)out)[i] = F16 ? f2h(v) : f2bf(v);
    }
}

int patch_gather_launch(const float* img, void* out, int out_dtype, int F, int H, int W, int p, int Kp,
                        hipStream_t st) {
    const long total = (long)F * (H / p) * (W / p) * Kp;
    if (out_dtype == SKIMI_F32)
        hipLaunchKernelGGL(patch_gather_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, img, (float*)out, F, H, W, p, Kp);
    else if (out_dtype == SKIMI_F16)
        hipLaunchKernelGGL((patch_gather_kernel<unsigned short, true>), dim3(grid_for(total)), dim3(256), 0, st, img,
                           (unsigned short*)out, F, H, W, p, Kp);
    else
        hipLaunchKernelGGL(patch_gather_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, st, img,
                           (unsigned short*)out, F, H, W, p, Kp);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// special tokens: x[f, 0:n, :] = table[(f % S == 0) ? 0 : 1][0:n, :]
//   aggregator: camera_token / register_token, frame 0 vs the rest (aggregator.py:308-331);
//   DINOv2: cls(+pos_embed[0]) and register tokens, same for every frame (table rows equal).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void special_tokens_kernel(float* __restrict__ x, const float* __restrict__ table,
                                                             int F, int S, int P, int n, int C) {
    const long total = (long)F * n * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int t = (int)((i / C) % n);
        const long f = i / ((long)C * n);
        const int sel = (f % S == 0) ? 0 : 1;
        x[(f * P + t) * C + c] = table[((long)sel * n + t) * C + c];
    }
}

int special_tokens_launch(float* x, const float* table, int F, int S, int P, int n, int C, hipStream_t st) {
    hipLaunchKernelGGL(special_tokens_kernel, dim3(grid_for((long)F * n * C)), dim3(256), 0, st, x, table, F, S, P, n, C);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// bilinear resize, align_corners=True, channels-last (F.interpolate as used by
// heads/dpt_head.py:459-484,442).  Index / weight arithmetic follows ATen's CPU kernel:
//   scale = (in-1)/(out-1) (float), src = scale*dst, i0 = min(int(src), in-1), i1 = min(i0+1, in-1),
//   l1 = clamp(src - i0, 0, 1), out = l0y*(l0x*p00 + l1x*p01) + l1y*(l0x*p10 + l1x*p11)
// ---------------------------------------------------------------------------------------
// T / TO: element type of the input / output map (float, or unsigned short = a 16-bit format: bf16, or fp16 when f16 is set)
// LN: the resized pixel's C = 128 channels (one 32-lane group, 4 channels per lane) are LayerNorm'ed before they are stored
// (the track head's fmap_norm over the feature map it has just resized, base_track_predictor.py:103-106: two-pass mean /
// variance as layernorm_kernel, reduced inside the 32-lane group) -- no fp32 round trip of the 1.1 GB map
template <typename T, typename TO = T, bool LN = false>
__global__ __launch_bounds__(256) void bilinear_ac_kernel(const T* __restrict__ in, TO* __restrict__ out, int N,
                                                          int h, int w, int H, int W, int C,
                                                          const float* __restrict__ tabx,
                                                          const float* __restrict__ taby, bool f16 = false,
                                                          const float* __restrict__ ln_g = nullptr,
                                                          const float* __restrict__ ln_b = nullptr, float ln_eps = 0.f) {
    // one output row (n, Y) per blockIdx.y; threads run over X * C/4: only 32-bit index math
    const int C4 = C / 4;
    const int rowlen = W * C4;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int Y = (int)(blockIdx.y % (unsigned)H);
    const long n = blockIdx.y / (unsigned)H;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rowlen; i += gridDim.x * 256) {
        const int X = i / C4;
        const int c4 = i - X * C4;
        const float fy = sy * Y, fx = sx * X;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly1 = fminf(fmaxf(fy - y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - x0, 0.f), 1.f);
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const T* base = in + n * (long)h * w * C + c4 * 4;
        float p00[4], p01[4], p10[4], p11[4];
        auto ld = [&](const T* p, float* d) {
            if (sizeof(T) == 4) {
                const float4 v = *reinterpret_cast<const float4*>(p);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            } else {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = x16tof((unsigned short)v[k], f16);
            }
        };
        ld(base + ((long)y0 * w + x0) * C, p00);
        ld(base + ((long)y0 * w + x1) * C, p01);
        ld(base + ((long)y1 * w + x0) * C, p10);
        ld(base + ((long)y1 * w + x1) * C, p11);
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            r[k] = ly0 * (lx0 * p00[k] + lx1 * p01[k]) + ly1 * (lx0 * p10[k] + lx1 * p11[k]);
        if (tabx != nullptr) {
            // fused UV positional embedding of the upsampled map (add_uv_pos_kernel below, same add)
            const int half = C / 2, c = c4 * 4;
            const float4 e = *reinterpret_cast<const float4*>(c < half ? tabx + (long)X * half + c
                                                                        : taby + (long)Y * half + (c - half));
            r[0] += e.x; r[1] += e.y; r[2] += e.z; r[3] += e.w;
        }
        if constexpr (LN) {   // C == 128: lanes 32 g .. 32 g + 31 hold one pixel (rowlen is a multiple of 32: whole groups run)
            float s = (r[0] + r[1]) + (r[2] + r[3]);
#pragma unroll
            for (int o_ = 16; o_ > 0; o_ >>= 1) s += __shfl_xor(s, o_, 32);
            const float mean = s * (1.0f / 128.0f);
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] -= mean;
                q += r[k] * r[k];
            }
#pragma unroll
            for (int o_ = 16; o_ > 0; o_ >>= 1) q += __shfl_xor(q, o_, 32);
            const float rstd = rsqrtf(q * (1.0f / 128.0f) + ln_eps);
            const float4 g4 = *reinterpret_cast<const float4*>(ln_g + c4 * 4), b4 = *reinterpret_cast<const float4*>(ln_b + c4 * 4);
            r[0] = r[0] * rstd * g4.x + b4.x;
            r[1] = r[1] * rstd * g4.y + b4.y;
            r[2] = r[2] * rstd * g4.z + b4.z;
            r[3] = r[3] * rstd * g4.w + b4.w;
        }
        TO* o = out + ((n * H + Y) * (long)W + X) * C + c4 * 4;
        if (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            bf16x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (short)f2x16(r[k], f16);
            *reinterpret_cast<bf16x4*>(o) = v;
        }
    }
}

// the same resize (+ optional UV embedding) written as bf16 hi / lo halves of the fp32 result
// (hi = bf16(v), lo = bf16(v - hi)), one [hi C | lo C] record per pixel: the input format of
// conv_direct.hip, same bytes and the same single write stream as the fp32 map
__global__ __launch_bounds__(256) void bilinear_ac_planes_kernel(const float* __restrict__ in,
                                                                 unsigned short* __restrict__ out, int N, int h, int w,
                                                                 int H, int W, int C, const float* __restrict__ tabx,
                                                                 const float* __restrict__ taby, int slice_records,
                                                                 uint4* __restrict__ zpage) {
    // slice_records: pixel = [C/32][hi 32 | lo 32] (operand records of the LDS-DMA bf16x3 GEMM) instead of
    // [hi C | lo C]; zpage: 256 bytes to clear (that kernel's zero page, behind the records)
    if (zpage != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) zpage[threadIdx.x] = uint4{0, 0, 0, 0};
    // 8 channels per thread: 16-B stores into each plane
    // one output row (n, Y) per blockIdx.y; threads run over X * C/8: only 32-bit index math
    const int C8 = C / 8;
    const int rowlen = W * C8;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int Y = (int)(blockIdx.y % (unsigned)H);
    const long n = blockIdx.y / (unsigned)H;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rowlen; i += gridDim.x * 256) {
        const int X = i / C8;
        const int c8 = i - X * C8;
        const float fy = sy * Y, fx = sx * X;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly1 = fminf(fmaxf(fy - y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - x0, 0.f), 1.f);
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* base = in + n * (long)h * w * C + c8 * 8;
        float p00[8], p01[8], p10[8], p11[8];
        auto ld8 = [&](const float* q, float* d) {
            const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        };
        ld8(base + ((long)y0 * w + x0) * C, p00);
        ld8(base + ((long)y0 * w + x1) * C, p01);
        ld8(base + ((long)y1 * w + x0) * C, p10);
        ld8(base + ((long)y1 * w + x1) * C, p11);
        float r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            r[k] = ly0 * (lx0 * p00[k] + lx1 * p01[k]) + ly1 * (lx0 * p10[k] + lx1 * p11[k]);
        if (tabx != nullptr) {
            const int half = C / 2, c = c8 * 8;
            float e[8];
            ld8(c < half ? tabx + (long)X * half + c : taby + (long)Y * half + (c - half), e);
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += e[k];
        }
        bf16x8 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned short hb = f2bf(r[k]);
            hi[k] = (short)hb;
            lo[k] = (short)f2bf(r[k] - bf2f(hb));
        }
        const long pix = ((n * H + Y) * (long)W + X) * 2 * C;
        // written once, read back much later by the conv (gigabytes in between): streaming stores
        if (slice_records) {
            const int c = c8 * 8;
            const long o = pix + (c >> 5) * 64 + (c & 31);
            __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8*>(out + o));
            __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(out + o + 32));
        } else {
            const long o = pix + c8 * 8;
            __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8*>(out + o));
            __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(out + o + C));
        }
    }
}

int bilinear_ac_planes_launch(const float* in, unsigned short* out, int N, int h, int w, int H, int W, int C, hipStream_t st,
                              const float* tabx, const float* taby, int slice_records, void* zpage) {
    SKIMI_CHECK_ARG(C % 16 == 0, "bilinear resize into planes needs C %% 16 == 0");
    SKIMI_CHECK_ARG(!slice_records || C % 32 == 0, "bilinear resize into records needs C %% 32 == 0");
    SKIMI_CHECK_ARG(tabx == nullptr || taby != nullptr, "fused uv pos embed needs both tables");
    SKIMI_CHECK_ARG((long)N * H < 65536, "bilinear resize into planes: N * H must be < 65536");
    const int rowlen = W * (C / 8);
    hipLaunchKernelGGL(bilinear_ac_planes_kernel, dim3((unsigned)cdiv(rowlen, 256), (unsigned)(N * H)), dim3(256), 0, st, in, out, N,
                       h, w, H, W, C, tabx, taby, slice_records, (uint4*)zpage);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

int bilinear_ac_launch(const void* in, void* out, int dtype, int N, int h, int w, int H, int W, int C,
                       hipStream_t st, const float* tabx, const float* taby, int out_dtype, const float* ln_g, const float* ln_b,
                       float ln_eps) {
    if (out_dtype < 0) out_dtype = dtype;
    if (ln_g != nullptr) {   // resize + LayerNorm over the 128 channels of every output pixel, fp32 out
        SKIMI_CHECK_ARG(ln_b != nullptr && C == 128 && out_dtype == SKIMI_F32 && (((uintptr_t)ln_g | (uintptr_t)ln_b) & 15) == 0,
                        "bilinear resize + LayerNorm: C == 128, fp32 output, 16-byte aligned gamma / beta");
        SKIMI_CHECK_ARG(tabx == nullptr || taby != nullptr, "fused uv pos embed needs both tables");
        SKIMI_CHECK_ARG((long)N * H < 65536, "bilinear resize: N * H must be < 65536");
        const dim3 grid((unsigned)cdiv((long)W * (C / 4), 256), (unsigned)(N * H));
        if (dtype == SKIMI_F32)
            hipLaunchKernelGGL((bilinear_ac_kernel<float, float, true>), grid, dim3(256), 0, st, (const float*)in, (float*)out, N, h, w,
                               H, W, C, tabx, taby, false, ln_g, ln_b, ln_eps);
        else
            hipLaunchKernelGGL((bilinear_ac_kernel<unsigned short, float, true>), grid, dim3(256), 0, st, (const unsigned short*)in,
                               (float*)out, N, h, w, H, W, C, tabx, taby, dtype == SKIMI_F16, ln_g, ln_b, ln_eps);
        SKIMI_LAUNCH_CHECK();
        return SKIMI_OK;
    }
    SKIMI_CHECK_ARG(out_dtype == dtype || (dtype != SKIMI_F32 && out_dtype == SKIMI_F32),
                    "bilinear resize: the output type is the input's, or fp32 from a 16-bit map");
    SKIMI_CHECK_ARG(C % 4 == 0, "bilinear resize needs C %% 4 == 0");
    SKIMI_CHECK_ARG(tabx == nullptr || (taby != nullptr && C % 8 == 0), "fused uv pos embed needs both tables, C %% 8 == 0");
    SKIMI_CHECK_ARG((long)N * H < 65536, "bilinear resize: N * H must be < 65536");
    const dim3 grid((unsigned)cdiv((long)W * (C / 4), 256), (unsigned)(N * H));
    if (dtype == SKIMI_F32)
        hipLaunchKernelGGL(bilinear_ac_kernel<float>, grid, dim3(256), 0, st, (const float*)in, (float*)out, N, h, w, H, W, C,
                           tabx, taby);
    else if (out_dtype == SKIMI_F32)
        hipLaunchKernelGGL((bilinear_ac_kernel<unsigned short, float>), grid, dim3(256), 0, st, (const unsigned short*)in,
                           (float*)out, N, h, w, H, W, C, tabx, taby, dtype == SKIMI_F16);
    else
        hipLaunchKernelGGL(bilinear_ac_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)in,
                           (unsigned short*)out, N, h, w, H, W, C, tabx, taby, dtype == SKIMI_F16);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// separable sin/cos UV positional embedding (heads/dpt_head.py:249-259, heads/utils.py):
//   x[n,y,xx,c] += c < C/2 ? tabx[xx][c] : taby[y][c - C/2]     (tables already x ratio 0.1)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_uv_pos_kernel(T* __restrict__ x, const float* __restrict__ tabx,
                                                         const float* __restrict__ taby, int N, int H, int W, int C, bool f16 = false) {
    const int C4 = C / 4, half = C / 2;
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const int X = (int)((i / C4) % W);
        const int Y = (int)((i / ((long)C4 * W)) % H);
        const float* t = c < half ? tabx + (long)X * half + c : taby + (long)Y * half + (c - half);
        const float4 e = *reinterpret_cast<const float4*>(t);
        T* p = x + i * 4;
        if (sizeof(T) == 4) {
            float4 v = *reinterpret_cast<float4*>(p);
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
            *reinterpret_cast<float4*>(p) = v;
        } else {
            bf16x4 v = *reinterpret_cast<bf16x4*>(p);
            v[0] = (short)f2x16(x16tof((unsigned short)v[0], f16) + e.x, f16);
            v[1] = (short)f2x16(x16tof((unsigned short)v[1], f16) + e.y, f16);
            v[2] = (short)f2x16(x16tof((unsigned short)v[2], f16) + e.z, f16);
            v[3] = (short)f2x16(x16tof((unsigned short)v[3], f16) + e.w, f16);
            *reinterpret_cast<bf16x4*>(p) = v;
        }
    }
}

int add_uv_pos_launch(void* x, int dtype, const float* tabx, const float* taby, int N, int H, int W, int C,
                      hipStream_t st) {
    SKIMI_CHECK_ARG(C % 8 == 0, "uv pos embed needs C %% 8 == 0");
    const long total = (long)N * H * W * (C / 4);
    if (dtype == SKIMI_F32)
        hipLaunchKernelGGL(add_uv_pos_kernel<float>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st, (float*)x, tabx, taby, N, H, W, C);
    else
        hipLaunchKernelGGL(add_uv_pos_kernel<unsigned short>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st,
                           (unsigned short*)x, tabx, taby, N, H, W, C, dtype == SKIMI_F16);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// camera head AdaLN: out = gate * (xn * (1 + scale) + shift) + x,  mod = [shift | scale | gate]
// (heads/camera_head.py:117-124,144-149)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adaln_kernel(const float* __restrict__ xn, const float* __restrict__ x,
                                                    const float* __restrict__ mod, float* __restrict__ out, long rows,
                                                    int D) {
    const long total = rows * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / D;
        const int c = (int)(i - r * D);
        const float shift = mod[r * 3 * D + c], scale = mod[r * 3 * D + D + c], gate = mod[r * 3 * D + 2 * D + c];
        out[i] = gate * (xn[i] * (1.f + scale) + shift) + x[i];
    }
}

int adaln_launch(const float* xn, const float* x, const float* mod, float* out, long rows, int D, hipStream_t st) {
    hipLaunchKernelGGL(adaln_kernel, dim3(grid_for(rows * D)), dim3(256), 0, st, xn, x, mod, out, rows, D);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// pred = first ? delta : pred + delta;  act = [T, quat, relu(fov)]  (camera_head.py:129-139, head_act.py:12-35)
// pred_pad [rows, 16] is the zero-padded copy fed back through embed_pose.
__global__ void pose_update_kernel(const float* __restrict__ delta, float* __restrict__ pred_pad,
                                   float* __restrict__ act_out, long rows, int first) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 9) return;
    const long r = i / 9;
    const int c = (int)(i - r * 9);
    const float p = first ? delta[i] : pred_pad[r * 16 + c] + delta[i];
    pred_pad[r * 16 + c] = p;
    act_out[i] = c >= 7 ? fmaxf(p, 0.f) : p;
}

int pose_update_launch(const float* delta, float* pred_pad, float* act_out, long rows, int first, hipStream_t st) {
    hipLaunchKernelGGL(pose_update_kernel, dim3((unsigned)cdiv(rows * 9, 128)), dim3(128), 0, st, delta, pred_pad, act_out, rows, first);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// DPT output stage: conv1x1(32 -> n_out) + activate_head (heads/dpt_head.py:106-110,241-242;
// head_act.py:61-125).  in [npix, 32] (already ReLU'd), W [n_out, 32], b [n_out].
//   mode 0 "exp":     pts = exp(xyz)                  (depth head, n_out = 2)
//   mode 1 "inv_log": pts = sign(y) * expm1(|y|)      (point head, n_out = 4)
//   conf = 1 + exp(last channel)
// ---------------------------------------------------------------------------------------
template <typename T, int NOUT>
__global__ __launch_bounds__(256) void dpt_out_kernel(const T* __restrict__ in, const float* __restrict__ Wt,
                                                      const float* __restrict__ b, float* __restrict__ pts,
                                                      float* __restrict__ conf, long npix, int mode, bool f16 = false) {
    __shared__ float ws[NOUT * 32 + NOUT];
    for (int i = threadIdx.x; i < NOUT * 32 + NOUT; i += 256) ws[i] = i < NOUT * 32 ? Wt[i] : b[i - NOUT * 32];
    __syncthreads();
    for (long px = (long)blockIdx.x * 256 + threadIdx.x; px < npix; px += (long)gridDim.x * 256) {
        float v[32];
        if (sizeof(T) == 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 t = *reinterpret_cast<const float4*>((const float*)in + px * 32 + 4 * k);
                v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>((const unsigned short*)in + px * 32 + 8 * k);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[8 * k + j] = x16tof((unsigned short)t[j], f16);
            }
        }
        float o[NOUT];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) s += v[k] * ws[n * 32 + k];
            o[n] = s + ws[NOUT * 32 + n];
        }
#pragma unroll
        for (int n = 0; n < NOUT - 1; ++n) {
            float y = o[n];
            if (mode == 0) y = expf(y);
            else y = copysignf(expm1f(fabsf(y)), y) * (y != 0.f ? 1.f : 0.f);
            pts[px * (NOUT - 1) + n] = y;
        }
        conf[px] = 1.f + expf(o[NOUT - 1]);
    }
}

int dpt_out_launch(const void* in, int dtype, const float* W, const float* b, int n_out, float* pts, float* conf,
                   long npix, int mode, hipStream_t st) {
    SKIMI_CHECK_ARG(n_out == 2 || n_out == 4, "dpt output stage supports 2 or 4 channels (got %d)", n_out);
    dim3 g(grid_for(npix, 256, 16384)), blk(256);
#define GO(T, N) hipLaunchKernelGGL((dpt_out_kernel<T, N>), g, blk, 0, st, (const T*)in, W, b, pts, conf, npix, mode, dtype == SKIMI_F16)
    if (dtype == SKIMI_F32) { if (n_out == 2) GO(float, 2); else GO(float, 4); }
    else { if (n_out == 2) GO(unsigned short, 2); else GO(unsigned short, 4); }
#undef GO
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// weight repacking (finalize time)
// ---------------------------------------------------------------------------------------
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = f2bf(in[i]);
}
int f32_to_bf16_launch(const float* in, void* out, long n, hipStream_t st) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, (unsigned short*)out, n);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}
__global__ void f32_to_f16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = f2h(in[i]);
}
int f32_to_f16_launch(const float* in, void* out, long n, hipStream_t st) {
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, (unsigned short*)out, n);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// Conv2d weight [Co, Ci, kh, kw] -> [Co, kh, kw, Ci]  (tap-major K of the implicit gather)
__global__ void permute_conv_kernel(const float* __restrict__ in, float* __restrict__ out, int Co, int Ci, int kh, int kw,
                                    int slice_major) {
    const long n = (long)Co * Ci * kh * kw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int ci, x, y;
        long co;
        if (slice_major) {
            // [co][ci / 32][y][x][ci % 32]: a K-tile of 32 is one (channel slice, tap); the 9 taps of a
            // slice are consecutive K-tiles, so their shifted input windows are re-read from L1 / L2
            // while still hot (tap-major order re-reads a window only after a whole pass over Cin)
            const int c32 = (int)(i % 32);
            x = (int)((i / 32) % kw);
            y = (int)((i / (32L * kw)) % kh);
            const int cs = (int)((i / (32L * kw * kh)) % (Ci / 32));
            co = i / ((long)Ci * kw * kh);
            ci = cs * 32 + c32;
        } else {
            ci = (int)(i % Ci);
            x = (int)((i / Ci) % kw);
            y = (int)((i / ((long)Ci * kw)) % kh);
            co = i / ((long)Ci * kw * kh);
        }
        out[i] = in[((co * Ci + ci) * kh + y) * kw + x];
    }
}
int permute_conv_launch(const float* in, float* out, int Co, int Ci, int kh, int kw, hipStream_t st, int slice_major) {
    SKIMI_CHECK_ARG(!slice_major || Ci % 32 == 0, "slice-major conv weights need Cin %% 32 == 0");
    hipLaunchKernelGGL(permute_conv_kernel, dim3(grid_for((long)Co * Ci * kh * kw)), dim3(256), 0, st, in, out, Co, Ci, kh, kw,
                       slice_major);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ConvTranspose2d weight [Ci, Co, s, s] -> [(a, b, co), Ci]  (pixel-shuffle GEMM, kernel == stride)
__global__ void permute_convT_kernel(const float* __restrict__ in, float* __restrict__ out, int Ci, int Co, int s) {
    const long n = (long)Ci * Co * s * s;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % Ci);
        const long r = i / Ci;              // (a*s + b)*Co + co
        const int co = (int)(r % Co);
        const int ab = (int)(r / Co);
        const int a = ab / s, b = ab - a * s;
        out[i] = in[(((long)ci * Co + co) * s + a) * s + b];
    }
}
int permute_convT_launch(const float* in, float* out, int Ci, int Co, int s, hipStream_t st) {
    hipLaunchKernelGGL(permute_convT_kernel, dim3(grid_for((long)Ci * Co * s * s)), dim3(256), 0, st, in, out, Ci, Co, s);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// [rows, K] -> [rows, Kp] zero padded (and tiling of a vector: out[i] = in[i % n])
__global__ void pad_cols_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int K, int Kp) {
    const long n = rows * Kp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        out[i] = k < K ? in[(i / Kp) * K + k] : 0.f;
    }
}
int pad_cols_launch(const float* in, float* out, long rows, int K, int Kp, hipStream_t st) {
    hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for(rows * Kp)), dim3(256), 0, st, in, out, rows, K, Kp);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}
__global__ void tile_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int reps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n * reps) out[i] = in[i % n];
}
int tile_vec_launch(const float* in, float* out, int n, int reps, hipStream_t st) {
    hipLaunchKernelGGL(tile_vec_kernel, dim3((unsigned)cdiv((long)n * reps, 256)), dim3(256), 0, st, in, out, n, reps);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

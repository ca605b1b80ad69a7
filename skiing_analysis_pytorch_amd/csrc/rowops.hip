// Row-wise, HBM-bound kernels: LayerNorm, per-head q/k LayerNorm + 2D RoPE, small gathers.
// One 64-lane wave owns one row; loads are coalesced 256-B (or 1-KiB float4) wave accesses,
// reductions are wave shuffles — no LDS, no block barrier.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace skimi {

// ---------------------------------------------------------------------------------------
// LayerNorm (torch.nn.LayerNorm: vggt/vggt/layers/block.py:49,66; heads/dpt_head.py:56).
// x2 != NULL: the row is the concatenation [x | x2] of two C/2-channel sources
// (aggregator.py:250-253's torch.cat of frame and global intermediates, never materialised).
// ---------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ x2, long ldx, long rows,
                                                        int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, void* out,
                                                        int out_dtype, long ldo, long grp_rows,
                                                        long grp_stride, long grp_off) {
    const int lane = threadIdx.x & 63;
    const long orow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (orow >= rows) return;
    // optional input row gather: output row r reads input row (r / grp_rows)*grp_stride + grp_off + r % grp_rows
    const long row = grp_rows > 0 ? (orow / grp_rows) * grp_stride + grp_off + orow % grp_rows : orow;
    const int half = C >> 1;
    float v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        float t = 0.f;
        if (c < C) {
            if (x2 == nullptr) t = x[row * ldx + c];
            else t = (c < half) ? x[row * ldx + c] : x2[row * ldx + (c - half)];
        }
        v[i] = t;
        s += t;
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        const float d = (c < C) ? (v[i] - mean) : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            float y = (v[i] - mean) * rstd;
            if (gamma) y *= gamma[c];
            if (beta) y += beta[c];
            if (out_dtype == SKIMI_F32) ((float*)out)[orow * ldo + c] = y;
            else ((unsigned short*)out)[orow * ldo + c] = f2x16(y, out_dtype == SKIMI_F16);
        }
    }
}

// 16-B-per-lane variant for the token streams (C a multiple of 256, one or two sources whose
// halves are multiples of 256 too): lane l owns columns 256*i + 4*l .. +3, so every wave access
// is a full 1-KiB row segment (4x fewer memory instructions than the scalar kernel).
template <int NV>   // float4 chunks per lane = C / 256
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                            long ldx, long rows, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, void* out,
                                                            int out_dtype, long ldo, long grp_rows, long grp_stride,
                                                            long grp_off, unsigned char* __restrict__ out_scales = nullptr) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63;
    const long orow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (orow >= rows) return;
    const long row = grp_rows > 0 ? (orow / grp_rows) * grp_stride + grp_off + orow % grp_rows : orow;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 256 * i + 4 * lane;
        const float* src = (x2 != nullptr && c >= C / 2) ? x2 + row * ldx + (c - C / 2) : x + row * ldx + c;
        v[i] = *reinterpret_cast<const float4*>(src);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c2 = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c2 * c2 + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / (float)C) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 256 * i + 4 * lane;
        float y[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
        if (gamma) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            y[0] *= g.x; y[1] *= g.y; y[2] *= g.z; y[3] *= g.w;
        }
        if (beta) {
            const float4 b = *reinterpret_cast<const float4*>(beta + c);
            y[0] += b.x; y[1] += b.y; y[2] += b.z; y[3] += b.w;
        }
        if (out_dtype == SKIMI_F32) {
            *reinterpret_cast<float4*>((float*)out + orow * ldo + c) = make_float4(y[0], y[1], y[2], y[3]);
        } else if (out_dtype == SKIMI_FP8MX) {
            // MXFP8 (gemm_fp8.hip): payload [rows][C] e4m3, scales [rows][C / 32]; 8 lanes hold one 32-column block
            const unsigned sb = mx_scale_byte(max8(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])))));
            *reinterpret_cast<int*>((unsigned char*)out + orow * (long)C + c) = mx_pack4(y[0], y[1], y[2], y[3], mx_inv_scale(sb));
            if ((lane & 7) == 0) out_scales[orow * (long)(C >> 5) + (c >> 5)] = (unsigned char)sb;
        } else if (out_dtype == SKIMI_BF16X3_REC) {
            // bf16x3 records [rows][C/32][hi 32 | lo 32] (the A operand of the LDS-DMA bf16x3 GEMM; ldo unused)
            bf16x4 h, l;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned short hb = f2bf(y[k]);
                h[k] = (short)hb;
                l[k] = (short)f2bf(y[k] - bf2f(hb));
            }
            unsigned short* q = (unsigned short*)out + orow * (2L * C) + (c >> 5) * 64 + (c & 31);
            *reinterpret_cast<bf16x4*>(q) = h;
            *reinterpret_cast<bf16x4*>(q + 32) = l;
        } else {   // bf16, or fp16 (the operand of SKIMI_PREC_F16's Linears)
            const bool h16 = out_dtype == SKIMI_F16;
            bf16x4 h;
            h[0] = (short)f2x16(y[0], h16); h[1] = (short)f2x16(y[1], h16); h[2] = (short)f2x16(y[2], h16); h[3] = (short)f2x16(y[3], h16);
            *reinterpret_cast<bf16x4*>((unsigned short*)out + orow * ldo + c) = h;
        }
    }
    // records: the 256 zero bytes behind them
    if (out_dtype == SKIMI_BF16X3_REC && blockIdx.x == 0 && threadIdx.x < 16)
        reinterpret_cast<uint4*>((unsigned short*)out + rows * (2L * C))[threadIdx.x] = uint4{0, 0, 0, 0};
}

int layernorm_launch(const float* x, const float* x2, int64_t ldx, int64_t rows, int C, const float* gamma,
                     const float* beta, float eps, void* out, int out_dtype, int64_t ldo, hipStream_t st,
                     int64_t grp_rows, int64_t grp_stride, int64_t grp_off) {
    SKIMI_CHECK_ARG(x && out && rows > 0 && C > 0, "skimi_layernorm: bad arguments");
    SKIMI_CHECK_ARG(C <= 64 * 32, "skimi_layernorm: C=%d exceeds 2048", C);
    SKIMI_CHECK_ARG(x2 == nullptr || (C % 2 == 0), "skimi_layernorm: concat needs even C");
    SKIMI_CHECK_ARG(out_dtype == SKIMI_F32 || out_dtype == SKIMI_BF16 || out_dtype == SKIMI_F16 || out_dtype == SKIMI_BF16X3_REC,
                    "skimi_layernorm: bad out_dtype %d", out_dtype);
    dim3 grid((unsigned)cdiv(rows, 4)), block(256);
    SKIMI_CHECK_ARG(out_dtype != SKIMI_BF16X3_REC || (C % 256 == 0 && ((uintptr_t)out & 127) == 0),
                    "skimi_layernorm: records output needs C %% 256 == 0 and a 128-byte aligned buffer");
    {
        auto al16 = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
        const bool vec = C % 256 == 0 && (x2 == nullptr || (C / 2) % 256 == 0) && ldx % 4 == 0 && ldo % 4 == 0 &&
                         al16(x) && al16(x2) && al16(gamma) && al16(beta) &&
                         ((uintptr_t)out & (out_dtype == SKIMI_F32 ? 15 : 7)) == 0;
        SKIMI_CHECK_ARG(out_dtype != SKIMI_BF16X3_REC || vec, "skimi_layernorm: records output needs 16-byte aligned, 4-element strided operands");
        if (vec) {
#define LNV_GO(V) hipLaunchKernelGGL(layernorm_vec_kernel<V>, grid, block, 0, st, x, x2, (long)ldx, (long)rows, gamma, beta, eps, out, out_dtype, (long)ldo, (long)grp_rows, (long)grp_stride, (long)grp_off)
            switch (C / 256) {
                case 1: LNV_GO(1); break;
                case 2: LNV_GO(2); break;
                case 3: LNV_GO(3); break;
                case 4: LNV_GO(4); break;
                case 6: LNV_GO(6); break;
                case 8: LNV_GO(8); break;
                default: goto scalar_path;
            }
#undef LNV_GO
            SKIMI_LAUNCH_CHECK();
            return SKIMI_OK;
        }
    }
scalar_path:
#define LN_GO(V) hipLaunchKernelGGL(layernorm_kernel<V>, grid, block, 0, st, x, x2, (long)ldx, (long)rows, C, gamma, beta, eps, out, out_dtype, (long)ldo, (long)grp_rows, (long)grp_stride, (long)grp_off)
    const int nv = (int)cdiv(C, 64);
    if (nv <= 2) LN_GO(2);
    else if (nv <= 8) LN_GO(8);
    else if (nv <= 16) LN_GO(16);
    else LN_GO(32);
#undef LN_GO
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

int layernorm_mx_launch(const float* x, int64_t ldx, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                        void* payload, void* scales, hipStream_t st) {
    SKIMI_CHECK_ARG(x && payload && scales && rows > 0, "layernorm_mx: bad arguments");
    SKIMI_CHECK_ARG(C % 256 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)payload & 3) == 0,
                    "layernorm_mx: C must be a multiple of 256 and the rows 16-byte aligned");
    dim3 grid((unsigned)cdiv(rows, 4)), block(256);
#define LNM_GO(V) hipLaunchKernelGGL(layernorm_vec_kernel<V>, grid, block, 0, st, x, (const float*)nullptr, (long)ldx, (long)rows, gamma, beta, eps, payload, (int)SKIMI_FP8MX, (long)C, 0L, 0L, 0L, (unsigned char*)scales)
    switch (C / 256) {
        case 1: LNM_GO(1); break;
        case 2: LNM_GO(2); break;
        case 3: LNM_GO(3); break;
        case 4: LNM_GO(4); break;
        case 6: LNM_GO(6); break;
        case 8: LNM_GO(8); break;
        default: set_error("layernorm_mx: C = %d not supported", C); return SKIMI_ERR_ARG;
    }
#undef LNM_GO
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// q/k LayerNorm(64, affine, eps) then 2D RoPE, in place on qkv [tokens, 3, heads, 64].
// Reference: vggt/vggt/layers/attention.py:54-58, vggt/vggt/layers/rope.py:119-188.
//   head_dim 64 = [vertical 32 | horizontal 32]; in each half t[0:32]:
//     out[d] = t[d]*cos[pos][d%16] + rot[d]*sin[pos][d%16],  rot = [-t[16:32], t[0:16]]
//   cos/sin tables are [npos, 16] fp32 built by the host exactly as rope.py:86-117.
// One wave per (token, head, q|k): lane = feature index.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(T* __restrict__ qkv, long tokens, int heads,
                                                          const float* __restrict__ qn_w,
                                                          const float* __restrict__ qn_b,
                                                          const float* __restrict__ kn_w,
                                                          const float* __restrict__ kn_b, float eps,
                                                          const int* __restrict__ pos,
                                                          const float* __restrict__ rcos,
                                                          const float* __restrict__ rsin, int npos) {
    const int lane = threadIdx.x & 63;
    const long vec = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // over tokens*2*heads
    const long total = tokens * 2 * heads;
    if (vec >= total) return;
    const long tok = vec / (2 * heads);
    const int rem = (int)(vec - tok * 2 * heads);
    const int which = rem / heads;   // 0 = q, 1 = k
    const int h = rem - which * heads;
    T* p = qkv + ((tok * 3 + which) * heads + h) * 64;
    float t;
    if (sizeof(T) == 4) t = ((const float*)p)[lane];
    else t = bf2f(((const unsigned short*)p)[lane]);
    const float* w = which ? kn_w : qn_w;
    const float* b = which ? kn_b : qn_b;
    if (w != nullptr) {
        const float mean = wave_sum(t) * (1.f / 64.f);
        const float d = t - mean;
        const float var = wave_sum(d * d) * (1.f / 64.f);
        t = d * rsqrtf(var + eps) * w[lane] + (b ? b[lane] : 0.f);
    }
    if (pos != nullptr) {
        const int sect = lane >> 5;           // 0: vertical (y), 1: horizontal (x)
        const int d = lane & 31;
        int pp = pos[tok * 2 + sect];
        pp = min(max(pp, 0), npos - 1);
        const float c = rcos[pp * 16 + (d & 15)];
        const float s = rsin[pp * 16 + (d & 15)];
        // partner lane within the 32-half: d<16 -> d+16 (negated), d>=16 -> d-16
        const float other = __shfl_xor(t, 16, 64);
        const float rot = (d < 16) ? -other : other;
        t = t * c + rot * s;
    }
    if (sizeof(T) == 4) ((float*)p)[lane] = t;
    else ((unsigned short*)p)[lane] = f2bf(t);
}

// Vectorised form: 8 lanes own one 64-wide head vector (8 contiguous features each, one 16-B
// bf16 / two 16-B fp32 accesses), so a wave moves 8 head vectors per instruction instead of one.
// LayerNorm reductions: in-lane over 8 values + 3 xor-shuffles inside the 8-lane group; the
// RoPE partner of feature d (d +- 16 inside its 32-half) lives 2 lanes away.
template <typename T>
__global__ __launch_bounds__(256) void qknorm_rope_vec_kernel(T* __restrict__ qkv, long tokens, int heads,
                                                              const float* __restrict__ qn_w,
                                                              const float* __restrict__ qn_b,
                                                              const float* __restrict__ kn_w,
                                                              const float* __restrict__ kn_b, float eps,
                                                              const int* __restrict__ pos,
                                                              const float* __restrict__ rcos,
                                                              const float* __restrict__ rsin, int npos) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long vec = gid >> 3;               // over tokens * 2 * heads
    const int sub = (int)(gid & 7);          // features 8*sub .. 8*sub+7
    const long total = tokens * 2 * heads;
    const bool live = vec < total;
    const long vv = live ? vec : total - 1;  // keep every lane in the shuffles
    const long tok = vv / (2 * heads);
    const int rem = (int)(vv - tok * 2 * heads);
    const int which = rem / heads;
    const int h = rem - which * heads;
    T* p = qkv + ((tok * 3 + which) * heads + h) * 64 + 8 * sub;
    float t[8];
    if (sizeof(T) == 4) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>((const float*)p + 4);
        t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
    } else {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = bf2f((unsigned short)a[j]);
    }
    const float* w = which ? kn_w : qn_w;
    const float* b = which ? kn_b : qn_b;
    if (w != nullptr) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += t[j];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { t[j] -= mean; q += t[j] * t[j]; }
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        const float rstd = rsqrtf(q * (1.f / 64.f) + eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = t[j] * rstd * w[8 * sub + j] + (b ? b[8 * sub + j] : 0.f);
    }
    if (pos != nullptr) {
        const int sect = sub >> 2;                    // features 0-31: y, 32-63: x
        const int d0 = (8 * sub) & 31;                // first feature inside the 32-half
        int pp = pos[tok * 2 + sect];
        pp = min(max(pp, 0), npos - 1);
        const float* ct = rcos + pp * 16 + (d0 & 15);
        const float* st = rsin + pp * 16 + (d0 & 15);
        const bool lower = d0 < 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float other = __shfl_xor(t[j], 2, 64);   // feature d +- 16
            const float rot = lower ? -other : other;
            t[j] = t[j] * ct[j] + rot * st[j];
        }
    }
    if (!live) return;
    if (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<float4*>((float*)p + 4) = make_float4(t[4], t[5], t[6], t[7]);
    } else {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(t[j]);
        *reinterpret_cast<bf16x8*>(p) = o;
    }
}

// Second-generation bf16 q/k-norm + RoPE (the aggregator's shape: head_dim 64, affine LayerNorm, 2-D
// RoPE).  The first vector kernel re-read 128 B of LayerNorm parameters and 64 B of RoPE table per
// 16-B vector through the L1 (9x the payload) and crossed lanes through ds_bpermute; here
//   * a thread keeps its 8 features' q and k LayerNorm parameters in registers for its whole
//     grid-stride loop (the feature slot of a thread never changes);
//   * the two RoPE tables ([npos][16] cos / sin, a few KiB) are staged in LDS once per workgroup;
//   * the 8-lane reductions and the RoPE partner exchange are DPP moves (quad_perm / half-row
//     mirror), no LDS traffic;
//   * q and k rows of one (token, head) are processed together: two independent 16-B loads in flight.
__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float dpp_mirror8(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
}
__device__ __forceinline__ float sum8(float s) {
    s += dpp_xor1(s);
    s += dpp_xor2(s);
    s += dpp_mirror8(s);
    return s;
}

__global__ __launch_bounds__(256) void qknorm_rope_bf16_kernel(unsigned short* __restrict__ qkv, long tokens, int heads,
                                                               const float* __restrict__ qn_w,
                                                               const float* __restrict__ qn_b,
                                                               const float* __restrict__ kn_w,
                                                               const float* __restrict__ kn_b, float eps,
                                                               const int* __restrict__ pos,
                                                               const float* __restrict__ rcos,
                                                               const float* __restrict__ rsin, int npos, float q_scale) {
    extern __shared__ __attribute__((aligned(16))) float tab[];   // [2][npos][16]
    for (int i = threadIdx.x; i < npos * 16; i += 256) {
        tab[i] = rcos[i];
        tab[npos * 16 + i] = rsin[i];
    }
    __syncthreads();
    const int sub = threadIdx.x & 7;                  // features 8*sub .. 8*sub+7
    float w[2][8], b[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        w[0][j] = qn_w[8 * sub + j]; b[0][j] = qn_b ? qn_b[8 * sub + j] : 0.f;
        w[1][j] = kn_w[8 * sub + j]; b[1][j] = kn_b ? kn_b[8 * sub + j] : 0.f;
    }
    const int sect = sub >> 2;                        // features 0-31 rotate by y, 32-63 by x
    const int d0 = (8 * sub) & 15;                    // table column of the first feature
    const bool lower = ((8 * sub) & 31) < 16;         // partner is d + 16 (else d - 16)
    const long total = tokens * heads;
    const long step = (long)gridDim.x * 32;
    // every lane runs the same trip count (DPP needs its 8-lane group complete): clamp, mask the store
    const long trips = (total + step - 1) / step;
    long idx = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    for (long it = 0; it < trips; ++it, idx += step) {
        const bool live = idx < total;
        const long ii = live ? idx : total - 1;
        const long tok = ii / heads;
        const int h = (int)(ii - tok * heads);
        unsigned short* pq = qkv + ((tok * 3) * heads + h) * 64 + 8 * sub;
        unsigned short* pk = pq + (long)heads * 64;
        const bf16x8 raw[2] = {*reinterpret_cast<const bf16x8*>(pq), *reinterpret_cast<const bf16x8*>(pk)};
        int pp = pos[tok * 2 + sect];
        pp = min(max(pp, 0), npos - 1);
        const float4 c0 = *reinterpret_cast<const float4*>(&tab[pp * 16 + d0]);
        const float4 c1 = *reinterpret_cast<const float4*>(&tab[pp * 16 + d0 + 4]);
        const float4 s0 = *reinterpret_cast<const float4*>(&tab[npos * 16 + pp * 16 + d0]);
        const float4 s1 = *reinterpret_cast<const float4*>(&tab[npos * 16 + pp * 16 + d0 + 4]);
        const float ct[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float st[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        bf16x8 outv[2];
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = bf2f((unsigned short)raw[z][j]);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
            const float mean = sum8(s) * (1.f / 64.f);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { t[j] -= mean; q += t[j] * t[j]; }
            const float rstd = rsqrtf(sum8(q) * (1.f / 64.f) + eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = t[j] * rstd * w[z][j] + b[z][j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float other = dpp_xor2(t[j]);   // feature d +- 16 lives two lanes away
                const float rot = lower ? -other : other;
                const float r = t[j] * ct[j] + rot * st[j];
                outv[z][j] = (short)f2bf(z == 0 ? r * q_scale : r);   // q: softmax scale folded in before the one rounding
            }
        }
        if (live) {
            *reinterpret_cast<bf16x8*>(pq) = outv[0];
            *reinterpret_cast<bf16x8*>(pk) = outv[1];
        }
    }
}

int qknorm_rope_launch(void* qkv, int dtype, int64_t tokens, int heads, const float* qn_w, const float* qn_b,
                       const float* kn_w, const float* kn_b, float eps, const int32_t* pos,
                       const float* rope_cos, const float* rope_sin, int rope_npos, hipStream_t st, float q_scale,
                       int* q_scaled) {
    SKIMI_CHECK_ARG(qkv && tokens > 0 && heads > 0, "skimi_qknorm_rope: bad arguments");
    if (q_scaled) *q_scaled = 0;
    SKIMI_CHECK_ARG(pos == nullptr || (rope_cos && rope_sin && rope_npos > 0), "skimi_qknorm_rope: pos without tables");
    if (((uintptr_t)qkv & 15) == 0 && dtype == SKIMI_BF16 && qn_w && kn_w && pos && rope_npos * 128 <= 48 * 1024) {
        const long total = tokens * heads;
        const unsigned blocks = (unsigned)std::min<long>(cdiv(total, 32), 256 * 8);
        hipLaunchKernelGGL(qknorm_rope_bf16_kernel, dim3(blocks), dim3(256), (size_t)rope_npos * 128, st,
                           (unsigned short*)qkv, (long)tokens, heads, qn_w, qn_b, kn_w, kn_b, eps, pos, rope_cos, rope_sin,
                           rope_npos, q_scaled ? q_scale : 1.f);
        if (q_scaled) *q_scaled = 1;
        SKIMI_LAUNCH_CHECK();
        return SKIMI_OK;
    }
    if (((uintptr_t)qkv & 15) == 0) {
        const long threads = tokens * 2 * heads * 8;
        dim3 g((unsigned)cdiv(threads, 256)), blk(256);
        if (dtype == SKIMI_F32)
            hipLaunchKernelGGL(qknorm_rope_vec_kernel<float>, g, blk, 0, st, (float*)qkv, (long)tokens, heads, qn_w, qn_b,
                               kn_w, kn_b, eps, pos, rope_cos, rope_sin, rope_npos);
        else
            hipLaunchKernelGGL(qknorm_rope_vec_kernel<unsigned short>, g, blk, 0, st, (unsigned short*)qkv, (long)tokens,
                               heads, qn_w, qn_b, kn_w, kn_b, eps, pos, rope_cos, rope_sin, rope_npos);
        SKIMI_LAUNCH_CHECK();
        return SKIMI_OK;
    }
    const long total = tokens * 2 * heads;
    dim3 grid((unsigned)cdiv(total, 4)), block(256);
    if (dtype == SKIMI_F32)
        hipLaunchKernelGGL(qknorm_rope_kernel<float>, grid, block, 0, st, (float*)qkv, (long)tokens, heads, qn_w,
                           qn_b, kn_w, kn_b, eps, pos, rope_cos, rope_sin, rope_npos);
    else
        hipLaunchKernelGGL(qknorm_rope_kernel<unsigned short>, grid, block, 0, st, (unsigned short*)qkv,
                           (long)tokens, heads, qn_w, qn_b, kn_w, kn_b, eps, pos, rope_cos, rope_sin, rope_npos);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------
// VideoPose3D expand_conv im2col (VideoPose3D/common/model.py:103,127): Conv1d(34 -> C, k)
// on channels-last input is a plain GEMM over windows of k*Cin contiguous floats; Cin = 34
// breaks 16-B alignment, so the windows are copied once into a zero-padded [rows, Kpad] matrix.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vp3d_im2col_kernel(const float* __restrict__ x, float* __restrict__ a0,
                                                          int B, int L, int Cin, int k, int Kpad) {
    const int Lout = L - k + 1;
    const long total = (long)B * Lout * Kpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Kpad);
        const long r = i / Kpad;
        const int b = (int)(r / Lout);
        const int t = (int)(r - (long)b * Lout);
        a0[i] = (c < k * Cin) ? x[((long)b * L + t) * Cin + c] : 0.f;
    }
}

int vp3d_im2col_launch(const float* x, float* a0, int B, int L, int Cin, int k, int Kpad, hipStream_t st) {
    const long total = (long)B * (L - k + 1) * Kpad;
    int blocks = (int)std::min<long>(cdiv(total, 256), 1024);
    hipLaunchKernelGGL(vp3d_im2col_kernel, dim3(blocks), dim3(256), 0, st, x, a0, B, L, Cin, k, Kpad);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

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
            else ((unsigned short*)out)[orow * ldo + c] = f2bf(y);
        }
    }
}

int layernorm_launch(const float* x, const float* x2, int64_t ldx, int64_t rows, int C, const float* gamma,
                     const float* beta, float eps, void* out, int out_dtype, int64_t ldo, hipStream_t st,
                     int64_t grp_rows, int64_t grp_stride, int64_t grp_off) {
    SKIMI_CHECK_ARG(x && out && rows > 0 && C > 0, "skimi_layernorm: bad arguments");
    SKIMI_CHECK_ARG(C <= 64 * 32, "skimi_layernorm: C=%d exceeds 2048", C);
    SKIMI_CHECK_ARG(x2 == nullptr || (C % 2 == 0), "skimi_layernorm: concat needs even C");
    dim3 grid((unsigned)cdiv(rows, 4)), block(256);
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

int qknorm_rope_launch(void* qkv, int dtype, int64_t tokens, int heads, const float* qn_w, const float* qn_b,
                       const float* kn_w, const float* kn_b, float eps, const int32_t* pos,
                       const float* rope_cos, const float* rope_sin, int rope_npos, hipStream_t st) {
    SKIMI_CHECK_ARG(qkv && tokens > 0 && heads > 0, "skimi_qknorm_rope: bad arguments");
    SKIMI_CHECK_ARG(pos == nullptr || (rope_cos && rope_sin && rope_npos > 0), "skimi_qknorm_rope: pos without tables");
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

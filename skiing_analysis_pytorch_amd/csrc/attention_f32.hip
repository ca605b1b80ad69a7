// Exact-fp32 flash attention on the f32-input MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fp32
// fmaf chain, bit-for-bit fp32).  This is the BF16X3-mode / head-side attention:
//   F.scaled_dot_product_attention in vggt/vggt/layers/attention.py:60-61 (fp32 CPU path),
//   the camera-head trunk (heads/camera_head.py:55-60, seq = S, head_dim 128) and the track
//   update-former's nn.MultiheadAttention (heads/track_modules/modules.py:150,184, head_dim 48).
//
// Orientation (both products "swapped" so the softmax row lives in one lane):
//   S^T[key, q]  = sum_d K[key, d] * Q[q, d]     A = K (key on lane&31), B = Q^T (q on lane&31)
//   O^T[d, q]   += sum_key V[key, d] * P^T[key, q]
// The S^T accumulator puts query q on the lane and 16 keys in the registers, so the row max /
// row sum are in-lane reductions plus one cross-half shuffle, and register r of the P tile is
// already the B operand of PV's k-step r (lane half h supplies key (r&3)+8(r>>2)+4h); the
// matching V row is fetched from LDS by the A operand.
#include "common.h"
#include "kernels.h"

namespace skimi {

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnArgs a) {
    constexpr int HDH = HD / 2;
    constexpr int LDK = HD + 4;   // padded LDS row (floats): conflict-free ds_read_b128 over 16 rows
    constexpr int DT = HD / 32;
    __shared__ __attribute__((aligned(16))) float Ks[32 * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[32 * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int hd = a.head_dim;
    const int q0 = blockIdx.x * 128 + wave * 32;

    const int go = a.batch_inner > 0 ? b / a.batch_inner : 0, gi = a.batch_inner > 0 ? b - go * a.batch_inner : b;
    const float* Q = (const float*)a.q + (long)go * a.q_batch2 + (long)gi * a.q_batch + (long)head * a.q_head;
    const float* K = (const float*)a.k + (long)go * a.k_batch2 + (long)gi * a.k_batch + (long)head * a.k_head;
    const float* V = (const float*)a.v + (long)go * a.v_batch2 + (long)gi * a.v_batch + (long)head * a.v_head;
    float* O = (float*)a.out + (long)go * a.o_batch2 + (long)gi * a.o_batch + (long)head * a.o_head;

    // Q fragment: lane (q, h) holds Q[q][HDH*h + t], pre-multiplied by the softmax scale
    float qreg[HDH];
    {
        const int q = min(q0 + l31, a.seq_q - 1);
        const float* qp = Q + (long)q * a.q_row + HDH * lh;
#pragma unroll
        for (int t = 0; t < HDH; t += 4) {
            const int d = HDH * lh + t;
            float4 v = make_float4(0, 0, 0, 0);
            if (d < hd) v = *reinterpret_cast<const float4*>(qp + t);
            qreg[t] = v.x * a.scale;
            qreg[t + 1] = v.y * a.scale;
            qreg[t + 2] = v.z * a.scale;
            qreg[t + 3] = v.w * a.scale;
        }
    }

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;

    const int nkt = (a.seq_k + 31) / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        // stage K and V tiles (32 keys x HD, zero padded)
        for (int c = tid; c < 32 * (HD / 4); c += 256) {
            const int r = c / (HD / 4);
            const int d = (c - r * (HD / 4)) * 4;
            const int key = kt * 32 + r;
            float4 kv = make_float4(0, 0, 0, 0), vv = kv;
            if (key < a.seq_k && d < hd) {
                kv = *reinterpret_cast<const float4*>(K + (long)key * a.k_row + d);
                vv = *reinterpret_cast<const float4*>(V + (long)key * a.v_row + d);
            }
            *reinterpret_cast<float4*>(&Ks[r * LDK + d]) = kv;
            *reinterpret_cast<float4*>(&Vs[r * LDK + d]) = vv;
        }
        __syncthreads();

        // S^T tile: 32 keys x 32 queries
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = &Ks[l31 * LDK + HDH * lh];
#pragma unroll
        for (int t = 0; t < HDH; t += 4) {
            const float4 kk = *reinterpret_cast<const float4*>(kp + t);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qreg[t], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qreg[t + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qreg[t + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qreg[t + 3], s, 0, 0, 0);
        }

        // online softmax; register r <-> key kt*32 + (r&3) + 8*(r>>2) + 4*lh
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= a.seq_k) s[r] = -INFINITY;
            mloc = fmaxf(mloc, s[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        const float alpha = expf(m - mnew);
        m = mnew;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - mnew);
            psum += s[r];
        }
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

        // O^T += V^T P^T : k-step r uses key (r&3)+8(r>>2)+4*lh from this lane half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const float vv = Vs[kr * LDK + i * 32 + l31];
                o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[r], o[i], 0, 0, 0);
            }
        }
    }

    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = 1.f / lsum;
    const int q = q0 + l31;
    if (q < a.seq_q) {
        float* op = O + (long)q * a.o_row;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = i * 32 + 8 * g + 4 * lh;
                if (d < hd) {
                    float4 v = make_float4(o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv,
                                           o[i][4 * g + 3] * inv);
                    *reinterpret_cast<float4*>(op + d) = v;
                }
            }
    }
}

int attention_f32_launch(const AttnArgs& a, hipStream_t st) {
    SKIMI_CHECK_ARG(a.q && a.k && a.v && a.out, "skimi_attention: null buffer");
    SKIMI_CHECK_ARG(a.batch > 0 && a.heads > 0 && a.seq_q > 0 && a.seq_k > 0, "skimi_attention: empty shape");
    SKIMI_CHECK_ARG(a.head_dim % 4 == 0 && a.head_dim > 0 && a.head_dim <= 128,
                    "skimi_attention: fp32 head_dim %d unsupported (multiple of 4, <= 128)", a.head_dim);
    SKIMI_CHECK_ARG(a.q_row % 4 == 0 && a.k_row % 4 == 0 && a.v_row % 4 == 0 && a.o_row % 4 == 0 &&
                    a.q_head % 4 == 0 && a.k_head % 4 == 0 && a.v_head % 4 == 0 && a.o_head % 4 == 0,
                    "skimi_attention: strides must keep 16-B alignment");
    dim3 grid((unsigned)cdiv(a.seq_q, 128), a.heads, a.batch), block(256);
    // the split of d over lane halves needs head_dim <= HD with the upper half starting at HD/2
    if (a.head_dim <= 64) hipLaunchKernelGGL(attn_f32_kernel<64>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(attn_f32_kernel<128>, grid, block, 0, st, a);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

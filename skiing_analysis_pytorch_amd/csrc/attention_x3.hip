// fp32-accurate flash attention for head_dim 64 on the bf16 matrix pipe (the attention of SKIMI_PREC_BF16X3, the
// parity mode): every operand is x = hi + lo (two bf16), every product three MFMAs (lo*hi + hi*lo + hi*hi), as
// the bf16x3 GEMMs do.  The exact-fp32 kernel (attention_f32.hip, v_mfma_f32_32x32x2_f32) runs at 65 TFLOP/s and
// made attention three quarters of a parity-mode step (228 of 290 ms per 8-view step); this one is bound by
// 50 MFMAs per 64-key tile and 32 queries.
//
// Structure = attention_bf16.hip's 32-query kernel: 4 waves x 32 queries per workgroup, swapped products
// (S^T = K Q^T, O^T += V^T P^T, a softmax row per lane pair), K / V tiles of 64 keys double-buffered in LDS by
// LDS-DMA -- here FOUR tiles per stage (K hi, K lo, V hi, V lo: 32 KiB, 64 KiB for the two stages, two
// workgroups per CU), streamed from the hi / lo bf16 planes that split_planes_kernel makes of the fp32 qkv
// buffer once per call (the split is then not repeated by every query block).  Q is read as fp32, multiplied
// by scale x log2(e) and split in registers, so the scores arrive in log2 units and the softmax is
// v_sub + v_exp2 per score in fp32; P is split like every other operand (p_hi = bf16(p), p_lo = bf16(p - p_hi));
// row sums are fp32 VALU adds of the unsplit p.  Output fp32.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace skimi {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

struct AttnX3Args {
    const float* q;                       // fp32, AttnArgs strides (elements)
    const unsigned short *khi, *klo, *vhi, *vlo;   // bf16 planes, the same element strides
    float* out;
    unsigned short* out_rec;              // or null: the result as bf16x3 records [token][C / 32][hi 32 | lo 32] instead of fp32
    long rec_row, rec_batch;              // ushorts per token row / per batch element
    long q_row, k_row, v_row, o_row, q_batch, k_batch, v_batch, o_batch, q_head, k_head, v_head, o_head;
    int batch, heads, seq_q, seq_k;
    float scale;
};

__device__ __forceinline__ bf16x8 read_tr(const char* vb, int t, int ks, int dt, int lane, int lh) {
    // V^T fragment of the row-major [key][d] tile by ds_read_b64_tr_b16 (attention_bf16.hip)
    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int dcol = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;
    bf16x8 vf;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = t * 32 + 16 * ks + 8 * half + 4 * lh + q4;
        const int chunk = (dcol >> 3) ^ (((row >> 1) & 1) << 2);
        const char* addr = vb + row * 128 + (chunk << 4) + ((dcol & 7) << 1);
        const s16x4 v4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)addr);
        vf[4 * half + 0] = v4[0];
        vf[4 * half + 1] = v4[1];
        vf[4 * half + 2] = v4[2];
        vf[4 * half + 3] = v4[3];
    }
    return vf;
}

__global__ __launch_bounds__(256, 2) void attn_x3_kernel(const AttnX3Args a, int nqb) {
    constexpr int KV = 64;
    constexpr int TILE = KV * 64 * 2;      // bytes of one bf16 tile
    __shared__ __attribute__((aligned(16))) char smem[8 * TILE];   // [buf][K hi | K lo | V hi | V lo]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    int id;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
        const int q = nblk >> 3, r = nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qb = id % nqb;
    const int bh = id / nqb;
    const int head = bh % a.heads, b = bh / a.heads;
    const int q0 = qb * 128 + wave * 32;

    const long koff = (long)b * a.k_batch + (long)head * a.k_head, voff = (long)b * a.v_batch + (long)head * a.v_head;
    const unsigned short *Kh = a.khi + koff, *Kl = a.klo + koff, *Vh = a.vhi + voff, *Vl = a.vlo + voff;

    // Q fragments (B operand): lane (q, h) holds Q[q][16 s + 8 h + j] * scale * log2(e), split hi + lo
    bf16x8 qh[4], ql[4];
    {
        const int q = min(q0 + l31, a.seq_q - 1);
        const float* qp = a.q + (long)b * a.q_batch + (long)head * a.q_head + (long)q * a.q_row + 8 * lh;
        const float c2 = a.scale * 1.44269504088896340736f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(qp + 16 * s);
            const float4 v1 = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = f[j] * c2;
                const unsigned short hb = f2bf(x);
                qh[s][j] = (short)hb;
                ql[s][j] = (short)f2bf(x - bf2f(hb));
            }
        }
    }

    const int nkt = (a.seq_k + KV - 1) / KV;
    const bool ragged = (a.seq_k & (KV - 1)) != 0;
    const int srow = lane >> 3, sch = lane & 7;
    auto issue = [&](int buf, int kt) {
        char* base = smem + buf * 4 * TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * (2 * wave + j) + srow;
            const int key = min(kt * KV + row, a.seq_k - 1);
            const int kc = sch ^ ((row >> 1) & 7);
            const int vc = sch ^ (((row >> 1) & 1) << 2);
            const int dst = (2 * wave + j) * 8 * 128;
            __builtin_amdgcn_global_load_lds((gbl_void*)(Kh + (long)key * a.k_row + kc * 8), (lds_void*)(base + dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(Kl + (long)key * a.k_row + kc * 8), (lds_void*)(base + TILE + dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(Vh + (long)key * a.v_row + vc * 8), (lds_void*)(base + 2 * TILE + dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(Vl + (long)key * a.v_row + vc * 8), (lds_void*)(base + 3 * TILE + dst), 16, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float lsum = 0.f;          // this lane's partial row sum (its 32 of every 64 keys)
    float m = -INFINITY;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): own LDS-DMA landed, then the barrier (protocol: attention_q64.hip)
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) issue(cur ^ 1, kt + 1);
        const char* kbh = smem + cur * 4 * TILE;
        const char* kbl = kbh + TILE;
        const char* vbh = kbh + 2 * TILE;
        const char* vbl = kbh + 3 * TILE;

        // ---- S^T = K Q^T, three terms, small ones first ----
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4);
                const bf16x8 kfh = *reinterpret_cast<const bf16x8*>(kbh + off);
                const bf16x8 kfl = *reinterpret_cast<const bf16x8*>(kbl + off);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh[ks], ks == 0 ? zero16 : s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, ql[ks], s[t], 0, 0, 0);
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh[ks], s[t], 0, 0, 0);
            }
        }
        if (ragged && kt == nkt - 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KV + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= a.seq_k) s[t][r] = -INFINITY;
                }
        }
        // ---- online softmax in fp32 (scores are in log2 units) ----
        float mloc = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[t][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        if (!__all(mnew == m)) {   // exact skip when no row's maximum moved
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
            lsum *= alpha;
            m = mnew;
        }
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[0][r] = __builtin_amdgcn_exp2f(s[0][r] - mnew);
            s[1][r] = __builtin_amdgcn_exp2f(s[1][r] - mnew);
            l0 += s[0][r];
            l1 += s[1][r];
        }
        lsum += l0 + l1;

        // ---- O^T += V^T P^T, P split like every other operand ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = s[t][8 * ks + j];
                    const unsigned short hb = f2bf(p);
                    ph[j] = (short)hb;
                    pl[j] = (short)f2bf(p - bf2f(hb));
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 vfh = read_tr(vbh, t, ks, dt, lane, lh);
                    const bf16x8 vfl = read_tr(vbl, t, ks, dt, lane, lh);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph, o[dt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) ahead of the barrier, written out
        __syncthreads();
    }

    const float tot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.f / tot;
    const int q = q0 + l31;
    if (a.out_rec != nullptr) {
        // records for the proj GEMM (no fp32 copy, no split pass).  A lane pair holds a row's 64 values as 4-element pieces
        // alternating between its two halves; one v_permlane32_swap per value pair gives every lane 8 consecutive elements
        // (lh = 0: the even 8-element groups, lh = 1: the odd ones), i.e. one 16-byte hi and one 16-byte lo store each.
        unsigned short* rp = a.out_rec + (long)b * a.rec_batch + (long)q * a.rec_row + (long)head * 128;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float w[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[dt][8 * pr + j] * inv),
                                                                    __float_as_uint(o[dt][8 * pr + 4 + j] * inv), false, false);
                    w[j] = __uint_as_float(r[0]);
                    w[4 + j] = __uint_as_float(r[1]);
                }
                bf16x8 h8, l8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned short hb = f2bf(w[j]);
                    h8[j] = (short)hb;
                    l8[j] = (short)f2bf(w[j] - bf2f(hb));
                }
                if (q < a.seq_q) {
                    unsigned short* sp = rp + dt * 64 + 8 * (2 * pr + lh);
                    *reinterpret_cast<bf16x8*>(sp) = h8;
                    *reinterpret_cast<bf16x8*>(sp + 32) = l8;
                }
            }
        return;
    }
    if (q < a.seq_q) {
        float* op = a.out + (long)b * a.o_batch + (long)head * a.o_head + (long)q * a.o_row;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(op + dt * 32 + 8 * g + 4 * lh) =
                    make_float4(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
    }
}

size_t attention_x3_scratch_bytes(long tokens, long row_elems) { return (size_t)tokens * row_elems * 4 + 512; }

// a: fp32 AttnArgs (q, k, v inside ONE packed buffer of `tokens` rows x `row_elems` floats starting at a.q).
// scratch: >= attention_x3_scratch_bytes: the hi / lo planes of that buffer.
// out_records: a.out receives bf16x3 records [tokens][heads * 64 / 32][hi 32 | lo 32] (+ 256 zero bytes behind them) instead of
// fp32 rows -- the operand form of the proj GEMM on the LDS-DMA bf16x3 kernel (o_row = heads * 64, plain [tokens, C] output)
int attention_x3_launch(const AttnArgs& a, long tokens, long row_elems, void* scratch, size_t scratch_bytes, hipStream_t st,
                        bool out_records) {
    SKIMI_CHECK_ARG(a.q && a.k && a.v && a.out && scratch, "attention_x3: null buffer");
    SKIMI_CHECK_ARG(a.head_dim == 64, "attention_x3: head_dim 64 only");
    SKIMI_CHECK_ARG(scratch_bytes >= attention_x3_scratch_bytes(tokens, row_elems), "attention_x3: scratch too small");
    SKIMI_CHECK_ARG(a.q_row % 8 == 0 && a.k_row % 8 == 0 && a.v_row % 8 == 0 && a.o_row % 4 == 0 && a.q_head % 8 == 0 &&
                    a.k_head % 8 == 0 && a.v_head % 8 == 0 && a.o_head % 4 == 0 && a.k_batch % 8 == 0 && a.v_batch % 8 == 0 &&
                    a.q_batch % 4 == 0 && a.o_batch % 4 == 0 && row_elems % 8 == 0,
                    "attention_x3: strides must keep 16-B alignment");
    const float* base = (const float*)a.q;
    const long koff = (const float*)a.k - base, voff = (const float*)a.v - base;
    SKIMI_CHECK_ARG(koff >= 0 && voff >= 0 && koff < row_elems && voff < row_elems, "attention_x3: q, k, v must share one packed buffer");
    // only k and v are read as planes (q is split in registers): when they are the tail of the packed rows ([q | k | v]),
    // the planes hold just those columns
    const long s0 = koff < voff ? koff : voff;
    const bool tail = a.k_row == row_elems && a.v_row == row_elems && a.k_batch % row_elems == 0 && a.v_batch % row_elems == 0 &&
                      s0 % 8 == 0;
    const long c0 = tail ? s0 : 0, pw = row_elems - c0;   // first column and width of the planes
    unsigned short* hi = (unsigned short*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    unsigned short* lo = hi + tokens * pw;
    int rc = split_planes_launch(base + c0, row_elems, tokens, (int)pw, hi, lo, st);
    if (rc) return rc;
    AttnX3Args x;
    x.q = base; x.khi = hi + (koff - c0); x.klo = lo + (koff - c0); x.vhi = hi + (voff - c0); x.vlo = lo + (voff - c0); x.out = (float*)a.out;
    x.q_row = a.q_row; x.k_row = tail ? pw : a.k_row; x.v_row = tail ? pw : a.v_row; x.o_row = a.o_row;
    x.q_batch = a.q_batch; x.o_batch = a.o_batch;
    x.k_batch = tail ? a.k_batch / row_elems * pw : a.k_batch;
    x.v_batch = tail ? a.v_batch / row_elems * pw : a.v_batch;
    x.q_head = a.q_head; x.k_head = a.k_head; x.v_head = a.v_head; x.o_head = a.o_head;
    x.batch = a.batch; x.heads = a.heads; x.seq_q = a.seq_q; x.seq_k = a.seq_k; x.scale = a.scale;
    x.out_rec = nullptr; x.rec_row = 0; x.rec_batch = 0;
    if (out_records) {
        SKIMI_CHECK_ARG(a.o_head == 64 && a.o_row == (long)a.heads * 64 && a.o_batch == (long)a.seq_q * a.o_row && ((uintptr_t)a.out & 127) == 0,
                        "attention_x3: records output needs a plain [tokens, heads * 64] result, 128-byte aligned");
        x.out_rec = (unsigned short*)a.out;
        x.rec_row = a.o_row * 2;               // 4 bytes per element: [hi 32 | lo 32] per 32-column slice
        x.rec_batch = (long)a.seq_q * x.rec_row;
        if (hipMemsetAsync((char*)a.out + (size_t)tokens * a.o_row * 4, 0, 256, st) != hipSuccess) {   // the zero page behind the records
            set_error("hipMemsetAsync(attention records zero page) failed");
            return SKIMI_ERR_HIP;
        }
    }
    const int nqb = (int)cdiv(a.seq_q, 128);
    const long nblk = (long)nqb * a.heads * a.batch;
    SKIMI_CHECK_ARG(nblk < (1l << 31), "attention_x3: grid too large");
    hipLaunchKernelGGL(attn_x3_kernel, dim3((unsigned)nblk), dim3(256), 0, st, x, nqb);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

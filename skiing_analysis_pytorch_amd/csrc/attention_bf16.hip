// bf16 flash attention for head_dim 64 on gfx950 MFMA (v_mfma_f32_32x32x16_bf16), fp32 softmax:
// frame attention (batch = S frames, seq 1374) and global attention (batch = B, seq = S*1374) of
// vggt/vggt/models/aggregator.py:260-305 via F.scaled_dot_product_attention
// (vggt/vggt/layers/attention.py:60-61), and the 24 DINOv2 blocks.
//
// This file: the launcher (attention_bf16_launch picks the kernel: the 64-query-per-wave kernel of
// attention_q64.hip by default) and the FIRST kernel, kept for A/B timing (SKIMI_ATTN_Q64=0):
// per workgroup 4 waves x 32 queries; K/V tiles of 64 keys double-buffered in LDS by LDS-DMA, one
// barrier per tile, row sums on the matrix pipe (ones-operand MFMAs).
// Products are "swapped" so each softmax row is lane-local:
//   S^T[key, q] = K Q^T   A = K from LDS (ds_read_b128, XOR-swizzled rows), B = Q from registers
//   O^T[d, q]  += V^T P^T A = V^T from LDS via ds_read_b64_tr_b16 (hardware transpose of a
//                          row-major [key][d] tile), B = P: the S^T accumulator converted to bf16
//                          in place (k order inside a step: key = 16s + 8(j>>2) + 4h + (j&3)).
// blockIdx -> (batch, head, q-block) is XCD-contiguous so one XCD's L2 holds a head's K/V.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace skimi {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ bf16x8 pack8(const f32x16& p, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(p[8 * s + j]);
    return r;
}

template <int dbg>   // dbg != 0: timing ablations only (SKIMI_ATTN_ABL), results are wrong
__global__ __launch_bounds__(256, 4) void attn_bf16_kernel(const AttnArgs a, int nqb) {
    constexpr int KV = 64;                 // keys per tile
    constexpr int TILE = KV * 64 * 2;      // bytes of one K (or V) tile
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // [buf][K|V]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-contiguous remap of the 1-D grid
    int id;
    {
        const int nblk = gridDim.x;
        const int bid = blockIdx.x;
        const int xcd = bid & 7;
        const int q = nblk >> 3, r = nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qb = id % nqb;
    const int bh = id / nqb;
    const int head = bh % a.heads, b = bh / a.heads;
    const int q0 = qb * 128 + wave * 32;

    const unsigned short* Q = (const unsigned short*)a.q + (long)b * a.q_batch + (long)head * a.q_head;
    const unsigned short* K = (const unsigned short*)a.k + (long)b * a.k_batch + (long)head * a.k_head;
    const unsigned short* V = (const unsigned short*)a.v + (long)b * a.v_batch + (long)head * a.v_head;
    unsigned short* O = (unsigned short*)a.out + (long)b * a.o_batch + (long)head * a.o_head;

    // Q fragments (B operand): lane (q, h) holds Q[q][16s + 8h + j]
    bf16x8 qf[4];
    {
        const int q = min(q0 + l31, a.seq_q - 1);
        const unsigned short* qp = Q + (long)q * a.q_row + 8 * lh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }

    // K/V staging by LDS-DMA (global_load_lds_dwordx4): one wave-instruction lands 8 rows x 128 B
    // linearly in LDS, so both bank swizzles are applied to the per-lane SOURCE chunk:
    //   K rows: chunk ^= (row>>1)&7          (conflict-free ds_read_b128 of 32 rows at one chunk)
    //   V rows: chunk ^= ((row>>1)&1)<<2     (conflict-free ds_read_b64_tr_b16 of 4-row blocks)
    // Rows past the end are clamped to the last valid row (finite data; their scores are masked
    // to -inf, so their P is exactly 0).  No staging registers, no ds_write pass.
    const int nkt = (a.seq_k + KV - 1) / KV;
    const bool ragged = (a.seq_k & (KV - 1)) != 0;
    const int srow = lane >> 3, sch = lane & 7;
    auto issue = [&](int buf, int kt) {
        char* kb = smem + buf * 2 * TILE;
        char* vb = kb + TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * (2 * wave + j) + srow;                 // tile row 0..63
            const int key = min(kt * KV + row, a.seq_k - 1);
            const int kc = sch ^ ((row >> 1) & 7);
            const int vc = sch ^ (((row >> 1) & 1) << 2);
            __builtin_amdgcn_global_load_lds((gbl_void*)(K + (long)key * a.k_row + kc * 8),
                                             (lds_void*)(kb + (2 * wave + j) * 8 * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(V + (long)key * a.v_row + vc * 8),
                                             (lds_void*)(vb + (2 * wave + j) * 8 * 128), 16, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    // row sums l[q] ride on the matrix pipe: lacc = ones^T . P^T accumulates sum_key P[key][q] into
    // every register of a 32x32 tile (the loop is VALU-bound: 4 extra MFMAs replace 32 v_add),
    // and it sums exactly the bf16-rounded P that the PV product uses
    f32x16 lacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    float m = -INFINITY;
    const float c2 = a.scale * 1.44269504088896340736f;   // softmax scale folded into exp2

    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): own LDS-DMA landed, then the barrier (protocol: attention_q64.hip)
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) issue(cur ^ 1, kt + 1);
        const char* kb = smem + cur * 2 * TILE;
        const char* vb = kb + TILE;

        // ---- S^T = K Q^T : two 32-key sub-tiles ----
        f32x16 s[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = (dbg & 1) ? qf[ks] :
                    *reinterpret_cast<const bf16x8*>(kb + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
                // the first k-step takes the constant 0 as C (inline operand: no 16-register clear)
                s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero16 : s[t], 0, 0, 0);
            }
        }
        // mask keys past the end (last tile only)
        if (ragged && kt == nkt - 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KV + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= a.seq_k) s[t][r] = -INFINITY;
                }
        }
        // ---- online softmax (row = query = this lane, both halves) ----
        float mloc = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[t][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        const float nmb = -(mnew * c2);
        // exact skip of the O / l rescale when no lane's running max moved (alpha == 1 for the
        // whole wave): after the first few tiles this is the common case
        if (!__all(mnew == m)) {
            const float alpha = __builtin_amdgcn_exp2f((m - mnew) * c2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha;
                o[1][r] *= alpha;
                lacc[r] *= alpha;
            }
            m = mnew;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // one v_fma per score (the build runs with -ffp-contract=off, so spell the fma out)
            if (!(dbg & 4)) {
                s[0][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][r], c2, nmb));
                s[1][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[1][r], c2, nmb));
            }
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pf = pack8(s[t], ks);
                if (!(dbg & 8)) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    // tr-read block: rows key0 + q (q = (lane&15)>>2), cols dcol0 + 4p (p = lane&3)
                    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
                    const int dcol = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;   // first of 4 d columns
                    bf16x8 vf = qf[0];
                    if (!(dbg & 2))
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
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        }

        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) ahead of the barrier, written out
        __syncthreads();
    }

    const float inv = 1.f / lacc[0];   // every register of lacc holds this lane's (query's) row sum
    const int q = q0 + l31;
    if (q < a.seq_q) {
        unsigned short* op = O + (long)q * a.o_row;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (short)f2x16(o[dt][4 * g + j] * inv, a.out_f16 != 0);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * lh) = v;
            }
    }
}

static int q64_enabled() {   // SKIMI_ATTN_Q64: 1 (default) the 64-query-per-wave kernel; 0 this file's 32-query kernel
    static const int q64 = getenv("SKIMI_ATTN_Q64") ? atoi(getenv("SKIMI_ATTN_Q64")) : 1;
    return q64;
}

// SKIMI_ATTN_MX=0: SKIMI_PREC_FP8 quantises the bf16 output rows in a separate pass instead (A/B timing, tests)
bool attention_mx_output_ok(int dtype, int heads, int head_dim) {
    static const bool dyn = getenv("SKIMI_ENV_DYNAMIC") && atoi(getenv("SKIMI_ENV_DYNAMIC"));
    static int use_mx = -1;
    if (use_mx < 0 || dyn) use_mx = getenv("SKIMI_ATTN_MX") ? atoi(getenv("SKIMI_ATTN_MX")) : 1;
    return use_mx && dtype == SKIMI_BF16 && head_dim == 64 && heads % 2 == 0 && q64_enabled() != 0;
}

int attention_bf16_launch(const AttnArgs& a, hipStream_t st) {
    SKIMI_CHECK_ARG(a.q && a.k && a.v && (a.out || a.out_mx), "skimi_attention: null buffer");
    SKIMI_CHECK_ARG(a.batch > 0 && a.heads > 0 && a.seq_q > 0 && a.seq_k > 0, "skimi_attention: empty shape");
    SKIMI_CHECK_ARG(a.head_dim == 64, "skimi_attention: the bf16 MFMA kernel is head_dim 64 only (got %d)", a.head_dim);
    SKIMI_CHECK_ARG(a.q_row % 8 == 0 && a.k_row % 8 == 0 && a.v_row % 8 == 0 && a.o_row % 4 == 0 &&
                    a.q_head % 8 == 0 && a.k_head % 8 == 0 && a.v_head % 8 == 0 && a.o_head % 4 == 0 &&
                    a.q_batch % 8 == 0 && a.k_batch % 8 == 0 && a.v_batch % 8 == 0 && a.o_batch % 4 == 0,
                    "skimi_attention: bf16 strides must keep 16-B alignment");
    const int nqb = (int)cdiv(a.seq_q, 128);
    const long nblk = (long)nqb * a.heads * a.batch;
    SKIMI_CHECK_ARG(nblk < (1l << 31), "skimi_attention: grid too large");
    const bool prof = prof_armed(PROF_ATTN_BF16, a.seq_k);
    if (prof) prof_before(st);
    // timing ablations (wrong results) exist only in a -DSKIMI_ABLATIONS build (SKIMI_ABLATIONS=1 python -m ...build)
#ifdef SKIMI_ABLATIONS
    static const int dbg = getenv("SKIMI_ATTN_ABL") ? atoi(getenv("SKIMI_ATTN_ABL")) : 0;
#else
    const int dbg = 0;
#endif
    const int q64 = q64_enabled();
    SKIMI_CHECK_ARG(!a.out_mx || q64 != 0, "skimi_attention: MXFP8 output rows are written by the 64-query kernel only");
    if (q64 != 0) {
        attention_q64_dispatch(a, st);
    } else if (a.q_prescaled) {   // this file's kernel multiplies by scale * log2(e): make that product 1
        AttnArgs a1 = a;
        a1.scale = 0.69314718055994530942f;
        hipLaunchKernelGGL(attn_bf16_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, st, a1, nqb);
    } else
    switch (dbg) {
#ifdef SKIMI_ABLATIONS
#define SKIMI_ABL_CASE(D) case D: hipLaunchKernelGGL(attn_bf16_kernel<D>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb); break;
        SKIMI_ABL_CASE(1) SKIMI_ABL_CASE(2) SKIMI_ABL_CASE(3) SKIMI_ABL_CASE(4) SKIMI_ABL_CASE(8) SKIMI_ABL_CASE(12)
        SKIMI_ABL_CASE(15)
#undef SKIMI_ABL_CASE
#endif
        default: hipLaunchKernelGGL(attn_bf16_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb);
    }
    if (prof) {
        const double bh = (double)a.batch * a.heads;
        prof_after(st, 4.0 * bh * a.seq_q * (double)a.seq_k * 64.0,
                   2.0 * bh * 64.0 * (2.0 * a.seq_q + 2.0 * a.seq_k));
    }
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

int attention_launch(const void* qkv, void* out, int dtype, int batch, int seq, int heads, int head_dim,
                     hipStream_t st, int q_prescaled, void* x3_scratch, size_t x3_scratch_bytes, int* out_records, int out_f16,
                     void* out_mx, void* out_mx_scales) {
    SKIMI_CHECK_ARG(qkv && (out || out_mx), "skimi_attention: null buffer");
    SKIMI_CHECK_ARG(!out_mx || (out_mx_scales && attention_mx_output_ok(dtype, heads, head_dim)),
                    "skimi_attention: MXFP8 output needs the 64-query bf16 kernel (head_dim 64, heads even) and a scale buffer");
    const bool want_rec = out_records != nullptr && *out_records != 0;
    if (out_records) *out_records = 0;
    AttnArgs a;
    const long C = (long)heads * head_dim;
    const size_t es = dtype == SKIMI_F32 ? 4 : 2;
    a.q = qkv;
    a.k = (const char*)qkv + C * es;
    a.v = (const char*)qkv + 2 * C * es;
    a.out = out;
    a.q_row = a.k_row = a.v_row = 3 * C;
    a.o_row = C;
    a.q_batch = a.k_batch = a.v_batch = (long)seq * 3 * C;
    a.o_batch = (long)seq * C;
    a.q_head = a.k_head = a.v_head = a.o_head = head_dim;
    a.batch = batch;
    a.heads = heads;
    a.seq_q = a.seq_k = seq;
    a.head_dim = head_dim;
    a.scale = 1.0f / sqrtf((float)head_dim);
    a.q_prescaled = dtype == SKIMI_F32 ? 0 : q_prescaled;
    a.out_f16 = dtype == SKIMI_F32 ? 0 : out_f16;
    if (out_mx) {   // [token][align128(C)] payload + [token][align128(C) / 32] scales: gemm_fp8_launch's A operand
        a.out_mx = out_mx;
        a.out_mx_scales = out_mx_scales;
        a.mx_row = (long)align_up((size_t)C, 128);
        a.mx_srow = a.mx_row / 32;
    }
    if (dtype == SKIMI_F32) {
        // fp32-accurate mode: the bf16x3 kernel when the caller lends scratch for the hi / lo planes (SKIMI_ATTN_X3=0: the
        // exact-fp32 MFMA kernel, A/B timing and a cross-check in the tests)
        static const bool dyn = getenv("SKIMI_ENV_DYNAMIC") && atoi(getenv("SKIMI_ENV_DYNAMIC"));
        static int use_x3 = -1;
        if (use_x3 < 0 || dyn) use_x3 = getenv("SKIMI_ATTN_X3") ? atoi(getenv("SKIMI_ATTN_X3")) : 1;
        const long tokens = (long)batch * seq;
        if (use_x3 && head_dim == 64 && x3_scratch && x3_scratch_bytes >= attention_x3_scratch_bytes(tokens, 3 * C)) {
            const bool rec = want_rec && ((uintptr_t)out & 127) == 0;
            if (rec) *out_records = 1;
            return attention_x3_launch(a, tokens, 3 * C, x3_scratch, x3_scratch_bytes, st, rec);
        }
        return attention_f32_launch(a, st);
    }
    return attention_bf16_launch(a, st);
}

}  // namespace skimi

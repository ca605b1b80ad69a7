// bf16 flash attention, head_dim 64, 64 queries per wave (gfx950, v_mfma_f32_32x32x16_bf16).
//
// Same products and layouts as attention_bf16.hip (swapped: S^T = K Q^T, O^T += V^T P^T, one
// softmax row per lane pair), restructured around what that kernel's profile showed: at 4 waves
// per SIMD and 126 VGPRs every K / V fragment was read from LDS right before the MFMA that used it
// (16 exposed LDS round trips per 32-query tile), and each wave re-read the whole K/V tile for
// only 32 queries.  Here a wave owns two 32-query blocks and 256 VGPRs:
//   * the 8 K fragments and 8 V^T fragments of a 64-key tile are read once into registers and
//     feed both query blocks (half the LDS traffic per FLOP, one exposed latency per tile);
//   * the two query blocks give the scheduler independent MFMA and VALU work inside one wave
//     (softmax of block 1 under the PV MFMAs of block 0); the second wave on the SIMD comes from
//     an independent workgroup (2 workgroups per CU, own barriers);
//   * softmax on SCALAR fp32 VALU ops (packed VOP3P ops do not co-issue with an MFMA in flight:
//     build.py compiles this file with -fno-slp-vectorize), the row maximum crosses the two 32-lane
//     halves with one v_permlane32_swap instead of an LDS bpermute, row sums are plain per-lane
//     partial sums merged once at the end (no ones-matrix MFMAs).
// Workgroup: 4 waves x 64 queries; K/V tiles of 64 keys double-buffered in LDS by LDS-DMA.
//
// Softmax VALU diet (the loop is VALU-bound: profiles/r01_attn_mfma_util.json, MFMA pipes busy 31-36 % of the
// launch).  Per score the loop used to issue v_fma (scale and subtract the row maximum), v_exp, v_add (row sum),
// half a v_max3 and half a v_cvt_pk.  The v_fma is gone:
//   * the softmax scale x log2(e) is folded into q before q's ONE rounding to bf16 (qknorm_rope_launch's q_scale;
//     q without qk-norm -- the DINOv2 blocks -- is rescaled here when its fragments are loaded);
//   * the row reference r (a stale row maximum) is subtracted BY THE MATRIX PIPE: S^T starts from one extra
//     MFMA k-step whose K-side fragment is the unit vector e0 and whose Q-side fragment carries -r, so the
//     accumulator is already s - r and v_exp2 takes it as it is.  r lives in bf16 (it has to pass through the
//     operand), which is fine: softmax is invariant to the shift as long as every tile of a row uses the same one;
//   * r is only moved when a tile runs away from it (lazy rescale, exact: O, the row sum and the tile's scores are
//     shifted by the same r' - r), so after the first tiles the rescale branch is dead;
//   * (round 3) and the test for that is on the tile's partial row SUM, which the loop forms anyway, not on a row
//     maximum: after the first tile no v_max3 is issued at all (see the loop).
// 4 extra MFMAs per 64-key tile (36 instead of 32) against 64 fewer VALU instructions per lane.
//
// Tried on top of this, no gain (2327-2344 us against 2307-2311 for the bench's global-attention launch):
// the two query blocks skewed, i.e. block 1's S^T MFMAs issued in one straight-line region with block 0's
// softmax (last tile peeled) so that they run under its VALU stream.  The loop is VALU- and power-bound
// (zero operands: 1144-1218 TFLOP/s, tools/mb_attn_power.py), not short of MFMA / VALU overlap.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace skimi {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int dbg>   // dbg != 0: timing ablations (SKIMI_ATTN_ABL), results are wrong
__global__ __launch_bounds__(256, 2) void attn_q64_kernel(const AttnArgs a, int nqb) {
    constexpr int KV = 64;                 // keys per tile
    constexpr int TILE = KV * 64 * 2;      // bytes of one K (or V) tile
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // [buf][K|V]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

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
    const int q0 = qb * 256 + wave * 64;

    const unsigned short* Q = (const unsigned short*)a.q + (long)b * a.q_batch + (long)head * a.q_head;
    const unsigned short* K = (const unsigned short*)a.k + (long)b * a.k_batch + (long)head * a.k_head;
    const unsigned short* V = (const unsigned short*)a.v + (long)b * a.v_batch + (long)head * a.v_head;
    unsigned short* O = (unsigned short*)a.out + (long)b * a.o_batch + (long)head * a.o_head;

    // Q fragments (B operand): lane (q, h) holds Q[q][16s + 8h + j]
    bf16x8 qf[2][4];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int q = min(q0 + qi * 32 + l31, a.seq_q - 1);
        const unsigned short* qp = Q + (long)q * a.q_row + 8 * lh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qi][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    if (!a.q_prescaled) {   // q straight from a projection (no qk-norm kernel in front): fold scale * log2(e) in here
        const float c2q = a.scale * 1.44269504088896340736f;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qi][s][j] = (short)f2bf(bf2f((unsigned short)qf[qi][s][j]) * c2q);
    }

    // K/V staging by LDS-DMA: one wave-instruction lands 8 rows x 128 B linearly, bank swizzles on
    // the per-lane SOURCE chunk (K: chunk ^= (row>>1)&7; V: chunk ^= ((row>>1)&1)<<2), rows past
    // the end clamped (their scores are masked to -inf)
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

    f32x16 o[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qi][dt][r] = 0.f;
    float lsum[2] = {0.f, 0.f};   // this lane's partial row sums (its 32 of every 64 keys)
    float mr[2] = {0.f, 0.f};     // row reference (log2 units, a bf16 value), set from the first tile
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the extra k-step: K side = e0 for every key (element k = 0 sits in the lh = 0 half), Q side = -r of the lane's query
    bf16x8 kone = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lh == 0) kone[0] = (short)0x3F80;
    bf16x8 qneg[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};

    // LDS-DMA double buffer, visibility protocol: a wave's global_load_lds writes land in LDS when ITS vmcnt
    // reaches 0, so every wave waits vmcnt(0) on its own DMA and only then enters the workgroup barrier; after
    // the barrier all four waves' shares of the tile are in LDS.  The wait is written out (not left to how
    // hipcc lowers __syncthreads(); gfx950 barriers do not wait for memory counters by themselves), and
    // tests/test_abi.py greps the emitted ISA for `s_waitcnt vmcnt(0)` ahead of every s_barrier of this kernel.
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const char* kb = smem + cur * 2 * TILE;
        const char* vb = kb + TILE;

        // ---- S^T = K Q^T for both query blocks: every K fragment feeds two MFMAs ----
        f32x16 s[2][2];   // [query block][32-key block]
        bf16x8 kf[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                kf[t][ks] = *reinterpret_cast<const bf16x8*>(kb + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);   // all 8 reads in flight before the first MFMA waits (counted lgkmcnt)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
                s[qi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qneg[qi], zero16, 0, 0, 0);   // s = -r
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int qi = 0; qi < 2; ++qi)
                    s[qi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t][ks], qf[qi][ks], s[qi][t], 0, 0, 0);
        }

        // ---- V^T fragments (ds_read_b64_tr_b16 of the row-major [key][d] tile), shared by both blocks ----
        bf16x8 vf[2][2][2];   // [32-key block][16-key step][d tile]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
                    const int dcol = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;   // first of 4 d columns
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int row = t * 32 + 16 * ks + 8 * half + 4 * lh + q4;
                        const int chunk = (dcol >> 3) ^ (((row >> 1) & 1) << 2);
                        const char* addr = vb + row * 128 + (chunk << 4) + ((dcol & 7) << 1);
                        const s16x4 v4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)addr);
                        vf[t][ks][dt][4 * half + 0] = v4[0];
                        vf[t][ks][dt][4 * half + 1] = v4[1];
                        vf[t][ks][dt][4 * half + 2] = v4[2];
                        vf[t][ks][dt][4 * half + 3] = v4[3];
                    }
                }

        // Next tile's DMA goes out only now: hipcc cannot tell the two LDS buffers apart and puts
        // s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 after an LDS-DMA is issued,
        // i.e. issuing at the top of the iteration exposed the whole DMA latency on every tile.
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nkt) issue(cur ^ 1, kt + 1);

        // mask keys past the end (last tile only)
        if (ragged && kt == nkt - 1) {
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * KV + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key >= a.seq_k) s[qi][t][r] = -INFINITY;
                    }
        }

#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            // ---- online softmax: row = query = this lane pair (l31, both halves); s holds score - r ----
            // Fast path (every tile but the first): no row maximum at all.  The scores are exponentiated against the
            // standing reference and only the tile's partial row sum is looked at: while it stays below 2^40 nothing can
            // overflow (P, O and the row sum have > 80 binades of headroom) and softmax is invariant to where the reference
            // sits, so the maximum -- 16 v_max3 + a cross-half swap + a vote per tile and query block, ~15 % of the VALU
            // issue that is level with the MFMA time here -- is not needed.  A row whose scores run away from its reference
            // trips the sum test (inf and NaN included); then the tile's scores are recomputed from the K tile still in LDS
            // and the reference is moved by the exact procedure of the first tile.
            bool redo = kt == 0;
            if (kt != 0) {
                // scalar VALU on purpose: packed-fp32 (VOP3P) ops do not co-issue with MFMAs (build.py)
                float l0 = 0.f, l1 = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float v0 = s[qi][t][r], v1 = s[qi][t][r + 1];
                        if (!(dbg & 2)) {
                            v0 = __builtin_amdgcn_exp2f(v0);
                            v1 = __builtin_amdgcn_exp2f(v1);
                        }
                        l0 += v0;
                        l1 += v1;
                        s[qi][t][r] = v0;
                        s[qi][t][r + 1] = v1;
                    }
                const float ts = l0 + l1;
                redo = __any(!(ts < 1.099511627776e12f));   // 2^40
                if (!redo) lsum[qi] += ts;
            }
            if (redo) {   // wave-uniform
                if (kt != 0) {
                    // the scores again (the fast path overwrote them with their exponentials)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = t * 32 + l31;
                        s[qi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qneg[qi], zero16, 0, 0, 0);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const bf16x8 kr = *reinterpret_cast<const bf16x8*>(kb + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
                            s[qi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr, qf[qi][ks], s[qi][t], 0, 0, 0);
                        }
                        if (ragged && kt == nkt - 1) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int key = kt * KV + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                if (key >= a.seq_k) s[qi][t][r] = -INFINITY;
                            }
                        }
                    }
                }
                float mloc = -INFINITY;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mloc = __builtin_fmaxf(mloc, s[qi][t][r]);
                mloc = xhalf_max(mloc);
                // move the reference: r := the tile's maximum on the first tile, r += max(tile maximum, 0) later.
                // Exact: O, the row sum and this tile's scores shift together.
                const float want = mr[qi] + (kt == 0 ? mloc : __builtin_fmaxf(mloc, 0.f));
                const float mnew = bf2f(f2bf(want));          // the reference has to be a bf16 value
                const float dlt = mnew - mr[qi];              // exact in fp32
                if (kt != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-dlt);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o[qi][0][r] *= alpha;
                        o[qi][1][r] *= alpha;
                    }
                    lsum[qi] *= alpha;
                }
                mr[qi] = mnew;
                qneg[qi][0] = lh == 0 ? (short)f2bf(-mnew) : (short)0;
                float l0 = 0.f, l1 = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float v0 = s[qi][t][r] - dlt, v1 = s[qi][t][r + 1] - dlt;
                        if (!(dbg & 2)) {
                            v0 = __builtin_amdgcn_exp2f(v0);
                            v1 = __builtin_amdgcn_exp2f(v1);
                        }
                        l0 += v0;
                        l1 += v1;
                        s[qi][t][r] = v0;
                        s[qi][t][r + 1] = v1;
                    }
                lsum[qi] += l0 + l1;
            }

            // ---- O^T += V^T P^T (P = the S^T accumulator as bf16, k order of attention_bf16.hip) ----
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (short)f2bf(s[qi][t][8 * ks + j]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[qi][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[t][ks][dt], pf, o[qi][dt], 0, 0, 0);
                }
        }

        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of tile kt+1 has landed in LDS
        if (!(dbg & 1)) __syncthreads();
    }

#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const float inv = 1.f / xhalf_sum(lsum[qi]);
        const int q = q0 + qi * 32 + l31;
        if (a.out_mx) {
            // MXFP8 rows (proj's A operand under SKIMI_PREC_FP8): a 32-channel block = one dt half of this head, 16 of its
            // values in this lane and 16 in lane ^ 32 -> one permlane swap for the block maximum, quantised from the fp32
            // quotient (one rounding, like LayerNorm's MX output)
            const long tok = (long)b * a.seq_q + q;
            unsigned char* pq = (unsigned char*)a.out_mx + tok * a.mx_row + head * 64;
            unsigned char* ps = (unsigned char*)a.out_mx_scales + tok * a.mx_srow + head * 2;
            unsigned sb2 = 0;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                float amax = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) amax = __builtin_fmaxf(amax, __builtin_fabsf(o[qi][dt][j] * inv));
                amax = xhalf_max(amax);
                const unsigned sb = mx_scale_byte(amax);
                const float si = mx_inv_scale(sb);
                sb2 |= sb << (8 * dt);
                if (q < a.seq_q) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<int*>(pq + dt * 32 + 8 * g + 4 * lh) =
                            mx_pack4(o[qi][dt][4 * g] * inv, o[qi][dt][4 * g + 1] * inv, o[qi][dt][4 * g + 2] * inv,
                                     o[qi][dt][4 * g + 3] * inv, si);
                }
            }
            if (q < a.seq_q && lh == 0) *reinterpret_cast<unsigned short*>(ps) = (unsigned short)sb2;
            continue;
        }
        if (q < a.seq_q) {
            unsigned short* op = O + (long)q * a.o_row;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (short)f2x16(o[qi][dt][4 * g + j] * inv, a.out_f16 != 0);
                    *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * lh) = v;
                }
        }
    }
}

void attention_q64_dispatch(const AttnArgs& a, hipStream_t st) {
    const int nqb = (int)cdiv(a.seq_q, 256);
    const long nblk = (long)nqb * a.heads * a.batch;
#ifdef SKIMI_ABLATIONS   // timing ablations (wrong results): only in a -DSKIMI_ABLATIONS build
    static const int dbg = getenv("SKIMI_ATTN_ABL") ? atoi(getenv("SKIMI_ATTN_ABL")) : 0;
#else
    const int dbg = 0;
#endif
    // SKIMI_ATTN_LDSPAD (bytes of unused dynamic LDS): occupancy experiments, e.g. 65536 -> one workgroup per CU
    static const size_t pad = getenv("SKIMI_ATTN_LDSPAD") ? (size_t)atol(getenv("SKIMI_ATTN_LDSPAD")) : 0;
    if (pad) {
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_q64_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
            once = true;
        }
        hipLaunchKernelGGL(attn_q64_kernel<0>, dim3((unsigned)nblk), dim3(256), pad, st, a, nqb);
        return;
    }
    switch (dbg) {
#ifdef SKIMI_ABLATIONS
        case 1: hipLaunchKernelGGL(attn_q64_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb); break;
        case 2: hipLaunchKernelGGL(attn_q64_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb); break;
        case 3: hipLaunchKernelGGL(attn_q64_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb); break;
#endif
        default: hipLaunchKernelGGL(attn_q64_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb);
    }
}

}  // namespace skimi

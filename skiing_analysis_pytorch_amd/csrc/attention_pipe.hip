// bf16 flash attention, head_dim 64: software-pipelined 32-key half-steps (gfx950).
//
// attention_q64.hip measured MFMA busy 42 % / vector issue 60 % with almost no co-execution: a
// wave's S = K Q^T MFMAs, its softmax VALU work and its P V MFMAs depend on each other in a row,
// and with two waves per SIMD nothing forces one to be on the matrix pipe while the other is on
// the VALU.  This kernel makes every phase of ONE wave carry an independent MFMA stream and VALU
// stream of about equal length, by skewing the two 32-query blocks (q0, q1) of the wave by half
// a step over 32-key half-steps j:
//
//   X(j):  VALU  softmax of S0(j)          MFMA  S1(j) = K(j) Q1^T ,  O1 += V(j-1)^T P1(j-1)
//   Y(j):  VALU  softmax of S1(j)          MFMA  S0(j+1) = K(j+1) Q0^T ,  O0 += V(j)^T P0(j)
//
// Products, fragment layouts and the P k-order are those of attention_bf16.hip (swapped
// products, one softmax row per lane pair).  K(j+1) and V(j) fragments are loaded inside X(j),
// each right after the last MFMA that reads the registers they replace, so their LDS latency
// runs under the rest of the phase and one register set serves Y(j) and X(j+1).
// K/V tiles of 64 keys (two half-steps) are staged by LDS-DMA into a 3-deep ring; tile t+2 is
// issued in iteration t behind the barrier that publishes tile t+1.  The V^T reads are
// ds_read_b64_tr_b16 in inline asm: with the builtin, hipcc puts s_waitcnt vmcnt(0) in front of
// the first transposed read after any LDS-DMA issue (it cannot tell the ring slots apart), which
// would drain the prefetch every tile.
//
// STATUS: correct (same tests as the other two kernels) but NOT the default: 2186-2210 us vs 2098 us
// (attention_q64.hip) on 4 x 10992 keys x 16 heads; selectable with SKIMI_ATTN_Q64=2.
// What the experiments say (tools/coissue.hip, coissue2.hip, coissue3.hip; MI355X):
//   * ablations here: softmax VALU stream alone 1285 us, MFMA + loads alone ~700-900 us, together
//     2186 us: the two streams add up instead of overlapping; rebalancing the regions (2 + 6 MFMAs)
//     and sched_group_barrier pipelines change nothing;
//   * the q64 kernel with ONE workgroup per CU (one wave per SIMD) is only 18 % slower than with two:
//     throughput is set by a single wave's in-order stream, the second wave barely fills gaps;
//   * every VALU op kind of the softmax (v_fma / v_fmamk / v_exp / v_add / v_max3 / v_cvt_pk /
//     v_perm / v_mov, RAW chains included) hides under v_mfma_f32_32x32x16_bf16 when it is
//     independent of the MFMAs (only packed-fp32 VOP3P ops never co-issue);
//   * a mini attention step without memory traffic reproduces the problem: MFMA stream 140 ns,
//     softmax stream 131 ns, strictly interleaved (1 MFMA + its 8 VALU, order pinned) 288 ns; with
//     the softmax input and P cut loose from the MFMAs 186 ns.  Feeding P (VALU-written registers)
//     to the PV MFMAs and S (MFMA-written registers) to the softmax, even one iteration apart,
//     brings back the sum.  Which interlock does that is the open question for the next round.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace skimi {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float xh_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xh_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// All LDS reads of the loop are inline asm, so that this file owns every lgkmcnt wait (a compiler
// wait for one of its own ds_reads would also wait for the younger asm reads: LDS returns in
// order) and so that hipcc's LDS-DMA alias guard (s_waitcnt vmcnt(0) before an LDS read it cannot
// tell apart from the in-flight ring slot) never fires.  Inline asm is a scheduling barrier: the
// statements below cut a half-step into regions, and the extra "+v" tie operands pin which
// region the surrounding (pure) VALU work may not sink out of.
//
// 4 x ds_read_b128 = the K fragments [k-step] of one 32-key half-step
#define SKIMI_KREAD(kn, ka, tie)                                                                     \
    asm volatile(                                                                                    \
        "ds_read_b128 %0, %5\n\t"                                                                    \
        "ds_read_b128 %1, %6\n\t"                                                                    \
        "ds_read_b128 %2, %7\n\t"                                                                    \
        "ds_read_b128 %3, %8"                                                                        \
        : "=v"(kn[0]), "=v"(kn[1]), "=v"(kn[2]), "=v"(kn[3]), "+v"(tie)                              \
        : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]), "v"(ka[3]))
// 8 transposed reads = the four V^T fragments [16-key step][d tile] of one 32-key half-step.
// a0 / a1: per-lane LDS byte address for d tile 0 / 1 at (key step 0, first 8 rows); the other
// three row groups are immediates (+1024 second 8 rows, +2048 second key step).
#define SKIMI_TR8(h, a0, a1, tie)                                                                    \
    asm volatile(                                                                                    \
        "ds_read_b64_tr_b16 %0, %9\n\t"                                                              \
        "ds_read_b64_tr_b16 %1, %9 offset:1024\n\t"                                                  \
        "ds_read_b64_tr_b16 %2, %10\n\t"                                                             \
        "ds_read_b64_tr_b16 %3, %10 offset:1024\n\t"                                                 \
        "ds_read_b64_tr_b16 %4, %9 offset:2048\n\t"                                                  \
        "ds_read_b64_tr_b16 %5, %9 offset:3072\n\t"                                                  \
        "ds_read_b64_tr_b16 %6, %10 offset:2048\n\t"                                                 \
        "ds_read_b64_tr_b16 %7, %10 offset:3072"                                                     \
        : "=v"(h[0]), "=v"(h[1]), "=v"(h[2]), "=v"(h[3]), "=v"(h[4]), "=v"(h[5]), "=v"(h[6]), "=v"(h[7]), \
          "+v"(tie)                                                                                  \
        : "v"(a0), "v"(a1))
// K fragments landed (the 8 younger transposed reads may still be in flight); the data of an asm
// read exists only behind its wait: the "+v" operands order every consumer after it
#define SKIMI_KWAIT(kn)                                                                              \
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(kn[0]), "+v"(kn[1]), "+v"(kn[2]), "+v"(kn[3]))
#define SKIMI_TR8_WAIT(h, tie)                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                              \
                 : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), \
                   "+v"(tie))

template <int dbg>   // dbg != 0: timing ablations (SKIMI_ATTN_ABL), results are wrong
__global__ __launch_bounds__(256, 2) void attn_pipe_kernel(const AttnArgs a, int nqb) {
    constexpr int KV = 64;                 // keys per staged tile (two half-steps)
    constexpr int TILE = KV * 64 * 2;      // bytes of one K (or V) tile
    constexpr int NBUF = 3;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * TILE];   // [slot][K|V]

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

    // LDS-DMA staging (as attention_bf16.hip): one wave-instruction = 8 rows x 128 B, bank
    // swizzles on the per-lane SOURCE chunk, rows past the end clamped (scores masked to -inf)
    const int nkt = (a.seq_k + KV - 1) / KV;
    const int srow = lane >> 3, sch = lane & 7;
    auto issue = [&](int kt) {
        char* kb = smem + (kt % NBUF) * 2 * TILE;
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

    // per-lane fragment addresses inside a tile (the half-step adds hh * 4096 bytes)
    //   K: row = 32 hh + l31, 16-B chunk (2 ks + lh) ^ ((row >> 1) & 7)
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4);
    //   V^T (transposed read): rows 32 hh + 16 ks + 8 half + 4 lh + q4, d columns dcol .. dcol+3
    unsigned voff[2];
    {
        const int q4 = (lane & 15) >> 2, p4 = lane & 3;
        const int row = 4 * lh + q4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int dcol = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * p4;
            const int chunk = (dcol >> 3) ^ (((row >> 1) & 1) << 2);
            voff[dt] = (unsigned)(size_t)smem + TILE + row * 128 + (chunk << 4) + ((dcol & 7) << 1);
        }
    }

    f32x16 o[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qi][dt][r] = 0.f;
    float lsum[2] = {0.f, 0.f};   // per-lane partial row sums (this lane's 16 of every 32 keys)
    float m[2] = {-INFINITY, -INFINITY};
    const float c2 = a.scale * 1.44269504088896340736f;   // softmax scale folded into exp2
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8 kf[4];            // K(j) fragments [ks]
    bf16x8 vf[2][2];         // V(j)^T fragments [16-key step][d tile]
    bf16x8 pf[2][2];         // P_q fragments [q block][16-key step]
    f32x16 s0, s1;           // S_q^T accumulators of the current half-step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[ks][dt] = zero8;   // X(0) multiplies them by P1(-1) = 0
    pf[1][0] = pf[1][1] = zero8;

    auto load_k = [&](int kt, int hh) {
        const char* kb = smem + (kt % NBUF) * 2 * TILE + hh * 4096;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kb + koff[ks]);
    };
    // mask the keys of half-step j that lie past seq_k (only the last one or two half-steps)
    auto mask = [&](f32x16& s, int j) {
        if (32 * (j + 1) > a.seq_k) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (key >= a.seq_k) s[r] = -INFINITY;
            }
        }
    };
    // online-softmax head: new running max; rescale O_q and l_q only when some lane's max moved
    auto sm_head = [&](const f32x16& s, int qi) -> float {
        if (dbg & 4) return 0.f;
        float mloc = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = __builtin_fmaxf(mloc, s[r]);
        mloc = xh_max(mloc);
        const float mnew = __builtin_fmaxf(m[qi], mloc);
        if (!__all(mnew == m[qi])) {
            const float alpha = __builtin_amdgcn_exp2f((m[qi] - mnew) * c2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[qi][0][r] *= alpha;
                o[qi][1][r] *= alpha;
            }
            lsum[qi] *= alpha;
            m[qi] = mnew;
        }
        return -(mnew * c2);
    };
    // softmax body: P = exp2(S c2 - m c2), row-sum partials, bf16 fragments in the PV k-order
    // (scalar VALU on purpose: packed-fp32 VOP3P ops do not co-issue with the MFMAs, see build.py)
    auto sm_body = [&](f32x16& s, int qi, float nmb) {
        if (dbg & 4) return;
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float v0 = __builtin_fmaf(s[r], c2, nmb), v1 = __builtin_fmaf(s[r + 1], c2, nmb);
            if (!(dbg & 2)) {
                v0 = __builtin_amdgcn_exp2f(v0);
                v1 = __builtin_amdgcn_exp2f(v1);
            }
            l0 += v0;
            l1 += v1;
            s[r] = v0;
            s[r + 1] = v1;
        }
        lsum[qi] += l0 + l1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[qi][ks][j] = (short)f2bf(s[8 * ks + j]);
    };
#define SKIMI_QK(S, QI)                                                                                          \
    if (!(dbg & 1)) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) S =                                                          \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[QI][ks], ks == 0 ? zero16 : S, 0, 0, 0)
// the same product in two halves (k-steps 0,1 / 2,3), so that a max region carries 2 MFMAs and an
// exp region 6: one MFMA per ~30 cycles of VALU issue everywhere (a wave issues in order; a run of
// MFMAs blocks its own VALU stream behind the busy matrix pipe and vice versa)
#define SKIMI_QK_A(S, QI)                                                                                        \
    if (!(dbg & 1)) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) S =                                                          \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[QI][ks], ks == 0 ? zero16 : S, 0, 0, 0)
#define SKIMI_QK_B(S, QI)                                                                                        \
    if (!(dbg & 1)) _Pragma("unroll") for (int ks = 2; ks < 4; ++ks) S =                                                          \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[QI][ks], S, 0, 0, 0)
// ask the scheduler for 6 x (1 MFMA, then NV VALU) in the current region
#define SKIMI_INTERLEAVE6(NV)                                                                                    \
    _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) {                                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                                      \
    }
#define SKIMI_PV(QI)                                                                                             \
    if (!(dbg & 1)) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) o[QI][dt] = \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks][dt], pf[QI][ks], o[QI][dt], 0, 0, 0)

    issue(0);
    if (nkt > 1) issue(1);
    __syncthreads();   // drains the LDS-DMA (vmcnt(0)) ahead of the barrier
    load_k(0, 0);
    SKIMI_QK(s0, 0);

    // One staged tile = half-steps j = 2 kt, 2 kt + 1; regions of a half-step (| = asm statement):
    //   QK1 + head0 | KREAD K(j+1) | PV1 + body0 | TR8 V(j), KWAIT | QK0' + head1 | TR8_WAIT | PV0 + body1
    // inside a region the scheduler alternates the MFMA stream with the softmax VALU stream.
    // `last` peels the key masking (and the missing successor tile) out of the steady state.
    s16x4 h[8];
    if (dbg & 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = s16x4{0, 0, 0, 0};
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto tile = [&](int kt, auto last) {
        constexpr bool LAST = decltype(last)::value;
        const unsigned slot = (unsigned)((kt % NBUF) * 2 * TILE);
        const unsigned nslot = (unsigned)(((kt + 1) % NBUF) * 2 * TILE);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int j = 2 * kt + hh;
            if (hh == 1 && !LAST) {
                // tile kt+1 is needed from here on (K(j+1)); everyone is past tile kt-1
                __syncthreads();
                if (kt + 2 < nkt) issue(kt + 2);
            }
            const bool more = hh == 0 || !LAST;   // compile-time after unrolling
            // ---------------- X(j) ----------------
            if (LAST) mask(s0, j);
            SKIMI_QK_A(s1, 1);                     // K(j) in kf
            float nmb0 = sm_head(s0, 0);
            bf16x8 kn[4];
            if (more) {
                unsigned ka[4];
                const unsigned kb = lds0 + (hh == 0 ? slot + 4096u : nslot);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ka[ks] = kb + (unsigned)koff[ks];
                if (!(dbg & 8)) SKIMI_KREAD(kn, ka, nmb0);         // K(j+1) into its own registers
                else { kn[0] = kf[0]; kn[1] = kf[1]; kn[2] = kf[2]; kn[3] = kf[3]; }
            }
            SKIMI_QK_B(s1, 1);
            SKIMI_PV(1);                           // O1 += V(j-1)^T P1(j-1)
            sm_body(s0, 0, nmb0);
            SKIMI_INTERLEAVE6(11);
            if (!(dbg & 8)) SKIMI_TR8(h, voff[0] + slot + hh * 4096, voff[1] + slot + hh * 4096, pf[0][1]);   // V(j)
            // ---------------- Y(j) ----------------
            if (LAST) mask(s1, j);
            if (more) {
                if (!(dbg & 8)) SKIMI_KWAIT(kn);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[ks] = kn[ks];
                SKIMI_QK_A(s0, 0);                 // S0(j+1) from K(j+1)
            }
            float nmb1 = sm_head(s1, 1);
            if (!(dbg & 8)) SKIMI_TR8_WAIT(h, nmb1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vf[ks][dt][e] = h[4 * ks + 2 * dt][e];
                        vf[ks][dt][4 + e] = h[4 * ks + 2 * dt + 1][e];
                    }
            if (more) SKIMI_QK_B(s0, 0);
            SKIMI_PV(0);                           // O0 += V(j)^T P0(j)
            sm_body(s1, 1, nmb1);
            SKIMI_INTERLEAVE6(11);
        }
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) tile(kt, std::false_type{});
    tile(nkt - 1, std::true_type{});
    SKIMI_PV(1);   // O1 += V(J-1)^T P1(J-1)
#undef SKIMI_QK
#undef SKIMI_QK_A
#undef SKIMI_QK_B
#undef SKIMI_INTERLEAVE6
#undef SKIMI_PV

#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const float inv = 1.f / xh_sum(lsum[qi]);
        const int q = q0 + qi * 32 + l31;
        if (q < a.seq_q) {
            unsigned short* op = O + (long)q * a.o_row;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (short)f2bf(o[qi][dt][4 * g + j] * inv);
                    *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * lh) = v;
                }
        }
    }
}

void attention_pipe_dispatch(const AttnArgs& a, hipStream_t st) {
    const int nqb = (int)cdiv(a.seq_q, 256);
    const long nblk = (long)nqb * a.heads * a.batch;
    static const int dbg = getenv("SKIMI_ATTN_ABL") ? atoi(getenv("SKIMI_ATTN_ABL")) : 0;
    switch (dbg) {
#define SKIMI_ABL_CASE(D) case D: hipLaunchKernelGGL(attn_pipe_kernel<D>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb); break;
        SKIMI_ABL_CASE(1) SKIMI_ABL_CASE(2) SKIMI_ABL_CASE(4) SKIMI_ABL_CASE(5) SKIMI_ABL_CASE(8) SKIMI_ABL_CASE(12)
        SKIMI_ABL_CASE(9) SKIMI_ABL_CASE(13)
#undef SKIMI_ABL_CASE
        default: hipLaunchKernelGGL(attn_pipe_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, st, a, nqb);
    }
}

}  // namespace skimi

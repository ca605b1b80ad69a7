// bf16x3 (fp32-accurate) MFMA contraction with LDS-DMA staging, for the fp32-path heads (wide DPT
// convs) whose operands live in HBM PRE-SPLIT into bf16:
//     x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi)      acc += Alo*Whi + Ahi*Wlo + Ahi*Whi
// The generic kernel (gemm.hip, NSPLIT = 3) splits fp32 operands in registers for every tile that
// touches them (a 3x3 conv re-splits each activation 9 x N/128 times) and stages through VGPRs.
// Here the split is done once (weights at finalize, activations by split_records_kernel) into
// 128-byte records [hi 32 | lo 32] per (row, 32-element K-slice), which stream global -> LDS with
// global_load_lds_dwordx4 a whole cache line per row; implicit-im2col gather = per-lane source
// pixel offset, padding taps read a zero page; epilogue shared with gemm.hip (bias / act /
// LayerScale / residuals / row remaps).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "gemm_epilogue.h"

namespace skimi {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// fp32 [rows, C] (row stride ld) -> hi, lo bf16 planes [rows, C] contiguous
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long ld, long rows, int C,
                                                           unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const int C4 = C / 4;
    const long total = rows * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C4;
        const int c = (int)(i - r * C4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        const float f[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned short hb = f2bf(f[k]);
            h[k] = (short)hb;
            l[k] = (short)f2bf(f[k] - bf2f(hb));
        }
        *reinterpret_cast<bf16x4*>(hi + r * C + c) = h;
        *reinterpret_cast<bf16x4*>(lo + r * C + c) = l;
    }
}

int split_planes_launch(const float* x, long ld, long rows, int C, void* hi, void* lo, hipStream_t st) {
    SKIMI_CHECK_ARG(C % 4 == 0 && ld % 4 == 0, "split_planes: C and ld must be multiples of 4");
    const long total = rows * (C / 4);
    const int blocks = (int)std::max<long>(1, std::min<long>(cdiv(total, 256), 65536));
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, st, x, ld, rows, C, (unsigned short*)hi,
                       (unsigned short*)lo);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// fp32 [rows, C] (row stride ld) -> records [rows][ceil(C / 32)][hi 32 | lo 32] bf16: the hi and lo
// halves of a 32-element K-slice are one 128-byte line, so every staging access of the kernel
// below is a whole cache line (as separate planes a K-tile of 32 is 64-byte pieces, which the
// L2 -> LDS path moves at half the rate: tools/dma_depth.hip, 55 vs 101 GB/s per CU).  A ragged
// last slice is zero-filled.  zpage: optional 256 bytes to clear (zero page of the conv gather).
__global__ __launch_bounds__(256) void split_records_kernel(const float* __restrict__ x, long ld, long rows, int C,
                                                            unsigned short* __restrict__ rec, uint4* __restrict__ zpage) {
    if (zpage != nullptr && blockIdx.x == 0 && threadIdx.x < 16) zpage[threadIdx.x] = uint4{0, 0, 0, 0};
    const int S = (C + 31) / 32;          // slices per row
    const long total = rows * S * 8;      // one thread per 4 elements
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long rs = i >> 3;           // (row, slice)
        const int c = (int)(i & 7) * 4;   // element within the slice
        const long r = rs / S;
        const int col = (int)(rs - r * S) * 32 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < C) v = *reinterpret_cast<const float4*>(x + r * ld + col);   // C % 4 == 0
        const float f[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned short hb = f2bf(f[k]);
            h[k] = (short)hb;
            l[k] = (short)f2bf(f[k] - bf2f(hb));
        }
        *reinterpret_cast<bf16x4*>(rec + rs * 64 + c) = h;
        *reinterpret_cast<bf16x4*>(rec + rs * 64 + 32 + c) = l;
    }
}

int split_records_launch(const float* x, long ld, long rows, int C, void* rec, hipStream_t st, void* zpage) {
    SKIMI_CHECK_ARG(C % 4 == 0 && ld % 4 == 0, "split_records: C and ld must be multiples of 4");
    const long total = rows * ((C + 31) / 32) * 8;
    const int blocks = (int)std::max<long>(1, std::min<long>(cdiv(total, 256), 65536));
    hipLaunchKernelGGL(split_records_kernel, dim3(blocks), dim3(256), 0, st, x, ld, rows, C, (unsigned short*)rec,
                       (uint4*)zpage);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

struct X3Rec {
    const char* a;      // A records: [rows][Cp / 32][hi 32 | lo 32] bf16
    const char* w;      // W records: [N][Kp / 32][hi 32 | lo 32]
    const char* zero;   // >= 128 zero bytes behind the A records (padding taps of the conv gather)
    long a_row_bytes;   // bytes of one A row (pixel) = 4 * Cp
    long w_row_bytes;   // bytes of one W row = 4 * Kp
    long a_bias;        // conv: lane offsets are biased so that a window starting in the padding stays >= 0
};

// bf16x3 MFMA contraction, single-stream loop (same structure as gemm256w4_kernel): 256x256 tile,
// 4 waves (one per SIMD), each a 128x128 quadrant = 256 accumulator registers, every wave
// software-pipelining its own stream with ONE workgroup barrier per K-tile.  Per k-step of 16:
// 16 fragment reads (A hi/lo x 4 row blocks, W hi/lo x 4 column blocks) feed 48 MFMAs, issued
// term-major (all lo*hi, all hi*lo, all hi*hi) so that no MFMA waits for the accumulator of the
// one before it.
//
// LDS: ring of ten 16-KiB slots; a K-tile (BK = 32) is four pieces of 128 rows x 128 B (a row =
// one record, hi | lo): q = 0,1 the A rows of wave row 0 / 1, q = 2,3 the W rows of wave column
// 0 / 1; piece q of K-tile kt sits in slot (4 kt + q) mod 10, so the A pieces of K-tile kt+2 go
// out a whole K-tile early:
//   k-step 0 of kt : read (kt, k-step 1); issue A pieces of kt+2   -> the slots K-tile kt-1's W left
//   k-step 1 of kt : lgkmcnt(0), vmcnt(8) = all of kt+1 landed, barrier (= K-tile kt released),
//                    read (kt+1, k-step 0); issue W pieces of kt+2 -> K-tile kt's A slots
// A wave-instruction of LDS-DMA writes 8 rows x 128 B linearly; the bank swizzle (16-B chunk ^=
// (row >> 1) & 7) is applied to the per-lane SOURCE address and again on the ds_read_b128; the lo
// half of a fragment is the hi half's address ^ 64.  Staging addresses are (wave-uniform 64-bit
// base) + (per-lane unsigned 32-bit offset): the K-tile / tap part goes into the base on the
// scalar unit, a padding-tap lane only swaps its offset for the zero page's.
#define SKIMI_X3_BAR()                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
#define SKIMI_X3_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))

template <int AMODE, int ABL = 0>   // a_mode of the launch (0 plain rows, 1 tap-major, 2 slice-major gather), compile-time:
                      // the K-tile body must stay one straight-line scheduling region.  ABL: timing
                      // ablations (bit 0 no staging, bit 1 no fragment reads, bit 2 no MFMAs)
__global__ __launch_bounds__(256, 1) void gemm_x3w4_kernel(const GemmArgs p, const X3Rec pl) {
    constexpr int BM = 256, BN = 256, BK = 32, RB = 128, PIECE = 128 * RB, NSLOT = 10;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    int id;
    {
        const int nblk = p.ntm * p.ntn;
        const int bid = blockIdx.x;
        const int xcd = bid & 7;
        const int q = nblk >> 3, r = nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = id / p.ntn, tn = id - tm * p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt = (p.K + BK - 1) / BK;
    // records output: the 256 zero bytes behind them (the consumer's padding taps)
    if (p.out_rec != nullptr && blockIdx.x == 0 && tid < 16)
        reinterpret_cast<uint4*>(p.out_rec + (long)p.M * p.rec_row)[tid] = uint4{0, 0, 0, 0};

    // ---- staging: a piece is 16 wave-instructions of 8 rows x 128 B; this wave issues 4*wave + j ----
    unsigned a_off[2][4], w_off[2][4];
    unsigned a_ok[2][4];   // conv: bit t = tap t of this lane's window is inside the image
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (4 * wave + j) * 8 + (lane >> 3);      // row within the piece
            const int c = (lane & 7) ^ ((row >> 1) & 7);           // source chunk of this lane's LDS chunk
            const int m = min(m0 + 128 * q + row, p.M - 1);
            a_ok[q][j] = ~0u;
            if (AMODE == 0) {
                a_off[q][j] = (unsigned)((long)m * pl.a_row_bytes + c * 16);
            } else {
                const int ohw = p.OH * p.OW;
                const int img = m / ohw;
                const int rem = m - img * ohw;
                const int oy = rem / p.OW;
                const int iy0 = oy * p.stride - p.pad, ix0 = (rem - oy * p.OW) * p.stride - p.pad;
                a_off[q][j] = (unsigned)((((long)img * p.cH + iy0) * p.cW + ix0) * pl.a_row_bytes + c * 16 + pl.a_bias);
                unsigned ok = 0;
                for (int ky = 0; ky < p.KH; ++ky)
                    for (int kx = 0; kx < p.KW; ++kx) {
                        const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
                        if (iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW) ok |= 1u << (ky * p.KW + kx);
                    }
                a_ok[q][j] = ok;
            }
            w_off[q][j] = (unsigned)((long)min(n0 + 128 * q + row, p.N - 1) * pl.w_row_bytes + c * 16);
        }
    // K-tiles are requested in order, so the (tap, channel slice) of the next one is kept as scalar
    // counters: no integer division in the loop
    int cur_tap = 0, cur_ky = 0, cur_kx = 0, cur_cs = 0;
    auto issue_a = [&](int kt, int slot0, int slot1) {
        const int tap = cur_tap, tap_dy = cur_ky * p.dil, tap_dx = cur_kx * p.dil;
        const int cs = AMODE == 0 ? kt : cur_cs;
        if (AMODE == 2) {          // slice-major K: the taps of a 32-channel slice are consecutive K-tiles
            ++cur_tap;
            if (++cur_kx == p.KW) { cur_kx = 0; ++cur_ky; }
            if (cur_tap == p.KH * p.KW) { cur_tap = 0; cur_ky = 0; ++cur_cs; }
        } else if (AMODE == 1) {   // tap-major K: the channel slices of a tap are consecutive K-tiles
            if (++cur_cs == p.cC / 32) {
                cur_cs = 0;
                ++cur_tap;
                if (++cur_kx == p.KW) { cur_kx = 0; ++cur_ky; }
            }
        }
        // wave-uniform: base of this K-tile's records, and the zero page seen from it
        if (ABL & 1) return;
        const long kbytes = ((long)tap_dy * p.cW + tap_dx) * pl.a_row_bytes + (long)cs * 128 - (AMODE != 0 ? pl.a_bias : 0);
        const char* base = pl.a + kbytes;
        const unsigned z = (unsigned)(pl.zero - base);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            char* dst = smem + (q == 0 ? slot0 : slot1) * PIECE + (4 * wave) * 8 * RB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = AMODE == 0 || ((a_ok[q][j] >> tap) & 1u) != 0;
                const unsigned o = ok ? a_off[q][j] : z + ((lane & 7) << 4);
                __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)o), (lds_void*)(dst + j * 8 * RB), 16, 0, 0);
            }
        }
    };
    auto issue_w = [&](int kt, int slot0, int slot1) {
        if (ABL & 1) return;
        const char* base = pl.w + (long)kt * 128;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            char* dst = smem + (q == 0 ? slot0 : slot1) * PIECE + (4 * wave) * 8 * RB;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)w_off[q][j]), (lds_void*)(dst + j * 8 * RB), 16, 0, 0);
        }
    };
    auto slot = [&](int x) { return x >= NSLOT ? x - NSLOT : x; };

    // ---- fragment reads: row block i of this wave's A piece / W piece, k-step s ----
    const int t = lh ^ ((l31 >> 1) & 7);
    const int lane_off = l31 * RB;
    bf16x8 ah0[4], al0[4], wh0[4], wl0[4], ah1[4], al1[4], wh1[4], wl1[4];
    if (ABL & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 c = {(short)lane, 1, 2, 3, 4, 5, 6, 7};
            ah0[i] = al0[i] = wh0[i] = wl0[i] = ah1[i] = al1[i] = wh1[i] = wl1[i] = c;
        }
    }
    auto read = [&](int sb, int s, bf16x8 (&ah)[4], bf16x8 (&al)[4], bf16x8 (&wh)[4], bf16x8 (&wl)[4]) {
        if (ABL & 2) return;
        const int x = ((2 * s) ^ t) << 4;   // hi chunk; the lo chunk is at x ^ 64
        const char* pa = smem + slot(sb + wr) * PIECE + lane_off;
        const char* pw = smem + slot(sb + 2 + wc) * PIECE + lane_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * RB + x);
            al[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * RB + (x ^ 64));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wh[i] = *reinterpret_cast<const bf16x8*>(pw + i * 32 * RB + x);
            wl[i] = *reinterpret_cast<const bf16x8*>(pw + i * 32 * RB + (x ^ 64));
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define SKIMI_X3_MFMA(AH, AL, WH, WL)                                                                             \
    do {                                                                                                           \
        if (ABL & 4) break;                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 4; ++i)                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL[i], WH[j], acc[i][j], 0, 0, 0);                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 4; ++i)                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[i], WL[j], acc[i][j], 0, 0, 0);                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 4; ++i)                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[i], WH[j], acc[i][j], 0, 0, 0);                 \
    } while (0)
#define SKIMI_X3_HEAD()                       \
    do {                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);   \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
    // 16 fragment reads behind the first 8 MFMAs, NDMA staging instructions behind the next 2 NDMA
#define SKIMI_X3_TAIL(NDMA)                                                        \
    do {                                                                           \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                     \
        }                                                                          \
        _Pragma("unroll") for (int g = 0; g < NDMA; ++g) {                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                     \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                     \
        }                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 40 - 2 * NDMA, 0);             \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)

    // prologue: K-tiles 0 and 1 whole
    issue_a(0, 0, 1);
    issue_w(0, 2, 3);
    if (nkt > 1) {
        issue_a(1, 4, 5);
        issue_w(1, 6, 7);
        SKIMI_X3_VMCNT(16);
    } else {
        SKIMI_X3_VMCNT(0);
    }
    SKIMI_X3_BAR();
    int sb = 0;   // slot of piece 0 of K-tile kt
    read(sb, 0, ah0, al0, wh0, wl0);
#define SKIMI_X3_KTILE(N1, N2)                                                      \
    do {                                                                            \
        const int nsb = slot(sb + 4);                                               \
        /* k-step 0 */                                                              \
        SKIMI_X3_HEAD();                                                            \
        read(sb, 1, ah1, al1, wh1, wl1);                                            \
        if (N2) issue_a(kt + 2, slot(sb + 8), slot(sb + 9));                        \
        SKIMI_X3_MFMA(ah0, al0, wh0, wl0);                                          \
        SKIMI_X3_TAIL((N2 ? 8 : 0));                                                \
        /* k-step 1 */                                                              \
        SKIMI_X3_HEAD();                                                            \
        if (N1) {                                                                   \
            if (N2) SKIMI_X3_VMCNT(8); else SKIMI_X3_VMCNT(0);                      \
            SKIMI_X3_BAR();                                                         \
            read(nsb, 0, ah0, al0, wh0, wl0);                                       \
            if (N2) issue_w(kt + 2, sb, slot(sb + 1));                              \
        }                                                                           \
        SKIMI_X3_MFMA(ah1, al1, wh1, wl1);                                          \
        if (N1) SKIMI_X3_TAIL((N2 ? 8 : 0)); else __builtin_amdgcn_sched_barrier(0); \
        sb = nsb;                                                                   \
    } while (0)
    int kt = 0;
    for (; kt + 2 < nkt; ++kt) SKIMI_X3_KTILE(true, true);
    if (kt + 1 < nkt) {
        SKIMI_X3_KTILE(true, false);
        ++kt;
    }
    SKIMI_X3_KTILE(false, false);
#undef SKIMI_X3_KTILE
#undef SKIMI_X3_TAIL
#undef SKIMI_X3_HEAD
#undef SKIMI_X3_MFMA
    __builtin_amdgcn_s_waitcnt(0xC07F);
    SKIMI_X3_BAR();   // nobody reads operand pieces any more: the epilogue slabs alias slots 0..3

    // ---- epilogue: per-wave 32-row x 128-column passes through a private 16-KiB LDS slab ----
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 128);
    const int n = n0 + wc * 128 + 4 * (lane & 31);
    // Interior tiles of the common case (fp32 rows, bias, ReLU before / after one fp32 residual): a
    // straight-line path, 8 rows of loads and stores in flight per lane (the checked loop below
    // pays a vmcnt(0) per row: stores share the counter with loads and hipcc cannot count across
    // its branches).  Measured on the 148x148 DPT convs: 400 -> ~200 us of a 1.8 ms launch.
    const bool relu_ok = (p.act == SKIMI_ACT_NONE || p.act == SKIMI_ACT_RELU) &&
                         (p.post_act == SKIMI_ACT_NONE || p.post_act == SKIMI_ACT_RELU);
    const bool fast = p.vec4 && p.store_mode == 0 && p.out_rpb == 0 && p.out_off == 0 && p.out2 == nullptr &&
                      p.out_dtype == SKIMI_F32 && p.gamma == nullptr && relu_ok && (p.resid != nullptr || p.resid2 == nullptr) &&
                      (p.resid == nullptr || p.resid_dtype == SKIMI_F32) &&
                      m0 + BM <= p.M && n0 + BN <= p.N;   // block-uniform
    if (fast) {
        const float lo1 = p.act == SKIMI_ACT_RELU ? 0.f : -__builtin_inff();
        const float lo2 = p.post_act == SKIMI_ACT_RELU ? 0.f : -__builtin_inff();
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bs = *reinterpret_cast<const float4*>(p.bias + n);
        const float* rs = reinterpret_cast<const float*>(p.resid);
        const float* rs2 = reinterpret_cast<const float*>(p.resid2);
        float* out = reinterpret_cast<float*>(p.out);
        // residual row of output row m (row_map's remap, branch-free: rows_per_batch 0 = one batch of all rows)
        const int rpb = p.resid_rpb > 0 ? p.resid_rpb : 0x7fffffff;
        auto res_row = [&](long m) {
            const int b = (int)m / rpb;
            return ((long)b * p.resid_bs + ((int)m - b * rpb) + p.resid_off) * p.ldr;
        };
#define SKIMI_X3_EPI_PASS(HAS_RES, HAS_OUT, HAS_REC)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int r = 0; r < 16; ++r)               \
            stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + j * 32 + l31] = acc[i][j][r];                           \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                     \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
        _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                   \
            const long mrow = m0 + wr * 128 + i * 32 + half * 16 + lh;                                             \
            float4 v[8], rr[8], r2[8];                                                                             \
            _Pragma("unroll") for (int it = 0; it < 8; ++it)                                                       \
                v[it] = *reinterpret_cast<const float4*>(&stg[(half * 16 + it * 2 + lh) * 128 + 4 * (lane & 31)]); \
            if (HAS_RES) _Pragma("unroll") for (int it = 0; it < 8; ++it)                                          \
                rr[it] = *reinterpret_cast<const float4*>(rs + res_row(mrow + it * 2) + n);                       \
            if (HAS_RES == 2) _Pragma("unroll") for (int it = 0; it < 8; ++it)                                     \
                r2[it] = *reinterpret_cast<const float4*>(rs2 + (mrow + it * 2) * p.ldr2 + n);                     \
            _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                     \
                float4 y = make_float4(fmaxf(v[it].x + bs.x, lo1), fmaxf(v[it].y + bs.y, lo1),                     \
                                       fmaxf(v[it].z + bs.z, lo1), fmaxf(v[it].w + bs.w, lo1));                    \
                if (HAS_RES) { y.x += rr[it].x; y.y += rr[it].y; y.z += rr[it].z; y.w += rr[it].w; }               \
                if (HAS_RES == 2) { y.x += r2[it].x; y.y += r2[it].y; y.z += r2[it].z; y.w += r2[it].w; }          \
                y = make_float4(fmaxf(y.x, lo2), fmaxf(y.y, lo2), fmaxf(y.z, lo2), fmaxf(y.w, lo2));               \
                if (HAS_OUT) *reinterpret_cast<float4*>(out + (mrow + it * 2) * p.ldo + n) = y;                    \
                if (HAS_REC) store_rec4(p, mrow + it * 2, n, y.x, y.y, y.z, y.w);                                  \
            }                                                                                                      \
        }                                                                                                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
    }
        // compile-time variants: a uniform branch inside the unrolled passes would cost a vmcnt(0) per store
        const int variant = (rs2 ? 8 : rs ? 4 : 0) | (out ? 2 : 0) | (p.out_rec ? 1 : 0);
        switch (variant) {
            case 11: SKIMI_X3_EPI_PASS(2, true, true) break;
            case 10: SKIMI_X3_EPI_PASS(2, true, false) break;
            case 9: SKIMI_X3_EPI_PASS(2, false, true) break;
            case 7: SKIMI_X3_EPI_PASS(1, true, true) break;
            case 6: SKIMI_X3_EPI_PASS(1, true, false) break;
            case 5: SKIMI_X3_EPI_PASS(1, false, true) break;
            case 3: SKIMI_X3_EPI_PASS(0, true, true) break;
            case 2: SKIMI_X3_EPI_PASS(0, true, false) break;
            case 1: SKIMI_X3_EPI_PASS(0, false, true) break;
            default: break;
        }
#undef SKIMI_X3_EPI_PASS
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            const int row_l = it * 2 + lh;
            const int m = m0 + wr * 128 + i * 32 + row_l;
            if (m >= p.M || n >= p.N) continue;
            const float4 v = *reinterpret_cast<const float4*>(&stg[row_l * 128 + 4 * (lane & 31)]);
            const RowMap rm = row_map(p, m);
            if (p.vec4) {
                store_four(p, rm, n, v);
            } else {
                store_one(p, rm, n, v.x);
                if (n + 1 < p.N) store_one(p, rm, n + 1, v.y);
                if (n + 2 < p.N) store_one(p, rm, n + 2, v.z);
                if (n + 3 < p.N) store_one(p, rm, n + 3, v.w);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
}

// The same loop for 128-column tiles (the 256 -> 128 channel conv of the DPT heads): wave (wr, wc)
// owns 128 rows x 64 columns (4 x 2 MFMA tiles, 24 MFMAs and 12 fragment reads per k-step); a
// K-tile is three pieces (A rows of wave row 0 / 1, the 128 W rows), the ring holds nine slots =
// three whole K-tiles: K-tile kt+3 is requested into K-tile kt's slots right after the barrier that
// releases them, so two K-tiles (96 KiB) are in flight while one is multiplied.
template <int AMODE>
__global__ __launch_bounds__(256, 1) void gemm_x3w4n_kernel(const GemmArgs p, const X3Rec pl) {
    constexpr int BM = 256, BN = 128, BK = 32, RB = 128, PIECE = 128 * RB, NSLOT = 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    int id;
    {
        const int nblk = p.ntm * p.ntn;
        const int bid = blockIdx.x;
        const int xcd = bid & 7;
        const int q = nblk >> 3, r = nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = id / p.ntn, tn = id - tm * p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt = (p.K + BK - 1) / BK;
    // records output: the 256 zero bytes behind them (the consumer's padding taps)
    if (p.out_rec != nullptr && blockIdx.x == 0 && tid < 16)
        reinterpret_cast<uint4*>(p.out_rec + (long)p.M * p.rec_row)[tid] = uint4{0, 0, 0, 0};

    // ---- staging (see gemm_x3w4_kernel): a piece is 16 wave-instructions of 8 rows x 128 B ----
    unsigned a_off[2][4], w_off[4];
    unsigned a_ok[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (4 * wave + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = min(m0 + 128 * q + row, p.M - 1);
            a_ok[q][j] = ~0u;
            if (AMODE == 0) {
                a_off[q][j] = (unsigned)((long)m * pl.a_row_bytes + c * 16);
            } else {
                const int ohw = p.OH * p.OW;
                const int img = m / ohw;
                const int rem = m - img * ohw;
                const int oy = rem / p.OW;
                const int iy0 = oy * p.stride - p.pad, ix0 = (rem - oy * p.OW) * p.stride - p.pad;
                a_off[q][j] = (unsigned)((((long)img * p.cH + iy0) * p.cW + ix0) * pl.a_row_bytes + c * 16 + pl.a_bias);
                unsigned ok = 0;
                for (int ky = 0; ky < p.KH; ++ky)
                    for (int kx = 0; kx < p.KW; ++kx) {
                        const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
                        if (iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW) ok |= 1u << (ky * p.KW + kx);
                    }
                a_ok[q][j] = ok;
            }
        }
        w_off[j] = (unsigned)((long)min(n0 + row, p.N - 1) * pl.w_row_bytes + c * 16);
    }
    int cur_tap = 0, cur_ky = 0, cur_kx = 0, cur_cs = 0;
    // all three pieces of K-tile kt -> slots s0 (A rows 0..127), s1 (A rows 128..255), s2 (W rows)
    auto issue = [&](int kt, int s0, int s1, int s2) {
        const int tap = cur_tap, tap_dy = cur_ky * p.dil, tap_dx = cur_kx * p.dil;
        const int cs = AMODE == 0 ? kt : cur_cs;
        if (AMODE == 2) {
            ++cur_tap;
            if (++cur_kx == p.KW) { cur_kx = 0; ++cur_ky; }
            if (cur_tap == p.KH * p.KW) { cur_tap = 0; cur_ky = 0; ++cur_cs; }
        } else if (AMODE == 1) {
            if (++cur_cs == p.cC / 32) {
                cur_cs = 0;
                ++cur_tap;
                if (++cur_kx == p.KW) { cur_kx = 0; ++cur_ky; }
            }
        }
        const long kbytes = ((long)tap_dy * p.cW + tap_dx) * pl.a_row_bytes + (long)cs * 128 - (AMODE != 0 ? pl.a_bias : 0);
        const char* base = pl.a + kbytes;
        const unsigned z = (unsigned)(pl.zero - base);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            char* dst = smem + (q == 0 ? s0 : s1) * PIECE + (4 * wave) * 8 * RB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = AMODE == 0 || ((a_ok[q][j] >> tap) & 1u) != 0;
                const unsigned o = ok ? a_off[q][j] : z + ((lane & 7) << 4);
                __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)o), (lds_void*)(dst + j * 8 * RB), 16, 0, 0);
            }
        }
        const char* wbase = pl.w + (long)kt * 128;
        char* dst = smem + s2 * PIECE + (4 * wave) * 8 * RB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wbase + (size_t)w_off[j]), (lds_void*)(dst + j * 8 * RB), 16, 0, 0);
    };
    auto slot = [&](int x) { return x >= NSLOT ? x - NSLOT : x; };

    // ---- fragment reads ----
    const int t = lh ^ ((l31 >> 1) & 7);
    const int fa_off = l31 * RB, fw_off = (wc * 64 + l31) * RB;
    bf16x8 ah0[4], al0[4], wh0[2], wl0[2], ah1[4], al1[4], wh1[2], wl1[2];
    auto read = [&](int sb, int s, bf16x8 (&ah)[4], bf16x8 (&al)[4], bf16x8 (&wh)[2], bf16x8 (&wl)[2]) {
        const int x = ((2 * s) ^ t) << 4;
        const char* pa = smem + slot(sb + wr) * PIECE + fa_off;
        const char* pw = smem + slot(sb + 2) * PIECE + fw_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * RB + x);
            al[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * RB + (x ^ 64));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wh[i] = *reinterpret_cast<const bf16x8*>(pw + i * 32 * RB + x);
            wl[i] = *reinterpret_cast<const bf16x8*>(pw + i * 32 * RB + (x ^ 64));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define SKIMI_X3N_MFMA(AH, AL, WH, WL)                                                                            \
    do {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int i = 0; i < 4; ++i)                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL[i], WH[j], acc[i][j], 0, 0, 0);                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int i = 0; i < 4; ++i)                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[i], WL[j], acc[i][j], 0, 0, 0);                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int i = 0; i < 4; ++i)                \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[i], WH[j], acc[i][j], 0, 0, 0);                 \
    } while (0)
#define SKIMI_X3N_HEAD()                      \
    do {                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);   \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
    // 12 fragment reads behind the first 6 MFMAs, NDMA staging instructions one per MFMA after them
#define SKIMI_X3N_TAIL(NDMA)                                                       \
    do {                                                                           \
        _Pragma("unroll") for (int g = 0; g < 6; ++g) {                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                     \
        }                                                                          \
        _Pragma("unroll") for (int g = 0; g < NDMA; ++g) {                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                     \
        }                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 18 - NDMA, 0);                 \
        __builtin_amdgcn_sched_barrier(0);                                         \
    } while (0)

    // prologue: the first three K-tiles
    issue(0, 0, 1, 2);
    if (nkt > 1) issue(1, 3, 4, 5);
    if (nkt > 2) issue(2, 6, 7, 8);
    if (nkt > 2) SKIMI_X3_VMCNT(24); else if (nkt > 1) SKIMI_X3_VMCNT(12); else SKIMI_X3_VMCNT(0);
    SKIMI_X3_BAR();
    int sb = 0;   // slot of piece 0 of K-tile kt
    read(sb, 0, ah0, al0, wh0, wl0);
    // N1 / N2 / N3: K-tiles kt+1 / kt+2 / kt+3 exist (literals; the last three K-tiles are peeled)
#define SKIMI_X3N_KTILE(N1, N2, N3)                                                 \
    do {                                                                            \
        const int nsb = slot(sb + 3);                                               \
        /* k-step 0 */                                                              \
        SKIMI_X3N_HEAD();                                                           \
        read(sb, 1, ah1, al1, wh1, wl1);                                            \
        SKIMI_X3N_MFMA(ah0, al0, wh0, wl0);                                         \
        SKIMI_X3N_TAIL(0);                                                          \
        /* k-step 1 */                                                              \
        SKIMI_X3N_HEAD();                                                           \
        if (N1) {                                                                   \
            if (N2) SKIMI_X3_VMCNT(12); else SKIMI_X3_VMCNT(0);                     \
            SKIMI_X3_BAR();                                                         \
            read(nsb, 0, ah0, al0, wh0, wl0);                                       \
            if (N3) issue(kt + 3, sb, slot(sb + 1), slot(sb + 2));                  \
        }                                                                           \
        SKIMI_X3N_MFMA(ah1, al1, wh1, wl1);                                         \
        if (N1) SKIMI_X3N_TAIL((N3 ? 12 : 0)); else __builtin_amdgcn_sched_barrier(0); \
        sb = nsb;                                                                   \
    } while (0)
    int kt = 0;
    for (; kt + 3 < nkt; ++kt) SKIMI_X3N_KTILE(true, true, true);
    if (kt + 2 < nkt) {
        SKIMI_X3N_KTILE(true, true, false);
        ++kt;
    }
    if (kt + 1 < nkt) {
        SKIMI_X3N_KTILE(true, false, false);
        ++kt;
    }
    SKIMI_X3N_KTILE(false, false, false);
#undef SKIMI_X3N_KTILE
#undef SKIMI_X3N_TAIL
#undef SKIMI_X3N_HEAD
#undef SKIMI_X3N_MFMA
    __builtin_amdgcn_s_waitcnt(0xC07F);
    SKIMI_X3_BAR();   // nobody reads operand pieces any more: the epilogue slabs alias slots 0, 1

    // ---- epilogue: per-wave 32-row x 64-column passes through a private 8-KiB LDS slab ----
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    const int n = n0 + wc * 64 + 4 * (lane & 15);
    const bool relu_ok = (p.act == SKIMI_ACT_NONE || p.act == SKIMI_ACT_RELU) &&
                         (p.post_act == SKIMI_ACT_NONE || p.post_act == SKIMI_ACT_RELU);
    const bool fast = p.vec4 && p.store_mode == 0 && p.out_rpb == 0 && p.out_off == 0 && p.out2 == nullptr &&
                      p.out_dtype == SKIMI_F32 && p.gamma == nullptr && relu_ok && (p.resid != nullptr || p.resid2 == nullptr) &&
                      (p.resid == nullptr || p.resid_dtype == SKIMI_F32) &&
                      m0 + BM <= p.M && n0 + BN <= p.N;   // block-uniform
    if (fast) {
        const float lo1 = p.act == SKIMI_ACT_RELU ? 0.f : -__builtin_inff();
        const float lo2 = p.post_act == SKIMI_ACT_RELU ? 0.f : -__builtin_inff();
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bs = *reinterpret_cast<const float4*>(p.bias + n);
        const float* rs = reinterpret_cast<const float*>(p.resid);
        const float* rs2 = reinterpret_cast<const float*>(p.resid2);
        float* out = reinterpret_cast<float*>(p.out);
        // residual row of output row m (row_map's remap, branch-free: rows_per_batch 0 = one batch of all rows)
        const int rpb = p.resid_rpb > 0 ? p.resid_rpb : 0x7fffffff;
        auto res_row = [&](long m) {
            const int b = (int)m / rpb;
            return ((long)b * p.resid_bs + ((int)m - b * rpb) + p.resid_off) * p.ldr;
        };
#define SKIMI_X3N_EPI_PASS(HAS_RES, HAS_OUT, HAS_REC)                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int r = 0; r < 16; ++r)               \
            stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + j * 32 + l31] = acc[i][j][r];                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                     \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
        const long mrow = m0 + wr * 128 + i * 32 + (lane >> 4);                                                    \
        float4 v[8], rr[8], r2[8];                                                                                 \
        _Pragma("unroll") for (int it = 0; it < 8; ++it)                                                           \
            v[it] = *reinterpret_cast<const float4*>(&stg[(it * 4 + (lane >> 4)) * 64 + 4 * (lane & 15)]);         \
        if (HAS_RES) _Pragma("unroll") for (int it = 0; it < 8; ++it)                                              \
            rr[it] = *reinterpret_cast<const float4*>(rs + res_row(mrow + it * 4) + n);                           \
        if (HAS_RES == 2) _Pragma("unroll") for (int it = 0; it < 8; ++it)                                         \
            r2[it] = *reinterpret_cast<const float4*>(rs2 + (mrow + it * 4) * p.ldr2 + n);                         \
        _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                         \
            float4 y = make_float4(fmaxf(v[it].x + bs.x, lo1), fmaxf(v[it].y + bs.y, lo1),                         \
                                   fmaxf(v[it].z + bs.z, lo1), fmaxf(v[it].w + bs.w, lo1));                        \
            if (HAS_RES) { y.x += rr[it].x; y.y += rr[it].y; y.z += rr[it].z; y.w += rr[it].w; }                   \
            if (HAS_RES == 2) { y.x += r2[it].x; y.y += r2[it].y; y.z += r2[it].z; y.w += r2[it].w; }              \
            y = make_float4(fmaxf(y.x, lo2), fmaxf(y.y, lo2), fmaxf(y.z, lo2), fmaxf(y.w, lo2));                   \
            if (HAS_OUT) *reinterpret_cast<float4*>(out + (mrow + it * 4) * p.ldo + n) = y;                        \
            if (HAS_REC) store_rec4(p, mrow + it * 4, n, y.x, y.y, y.z, y.w);                                      \
        }                                                                                                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                        \
    }
        const int variant = (rs2 ? 8 : rs ? 4 : 0) | (out ? 2 : 0) | (p.out_rec ? 1 : 0);
        switch (variant) {
            case 11: SKIMI_X3N_EPI_PASS(2, true, true) break;
            case 10: SKIMI_X3N_EPI_PASS(2, true, false) break;
            case 9: SKIMI_X3N_EPI_PASS(2, false, true) break;
            case 7: SKIMI_X3N_EPI_PASS(1, true, true) break;
            case 6: SKIMI_X3N_EPI_PASS(1, true, false) break;
            case 5: SKIMI_X3N_EPI_PASS(1, false, true) break;
            case 3: SKIMI_X3N_EPI_PASS(0, true, true) break;
            case 2: SKIMI_X3N_EPI_PASS(0, true, false) break;
            case 1: SKIMI_X3N_EPI_PASS(0, false, true) break;
            default: break;
        }
#undef SKIMI_X3N_EPI_PASS
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
            const int row_l = it * 4 + (lane >> 4);
            const int m = m0 + wr * 128 + i * 32 + row_l;
            if (m >= p.M || n >= p.N) continue;
            const float4 v = *reinterpret_cast<const float4*>(&stg[row_l * 64 + 4 * (lane & 15)]);
            const RowMap rm = row_map(p, m);
            if (p.vec4) {
                store_four(p, rm, n, v);
            } else {
                store_one(p, rm, n, v.x);
                if (n + 1 < p.N) store_one(p, rm, n + 1, v.y);
                if (n + 2 < p.N) store_one(p, rm, n + 2, v.z);
                if (n + 3 < p.N) store_one(p, rm, n + 3, v.w);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
}

static long x3_rows_in(const skimi_gemm_desc* d) { return d->a_mode == 0 ? (long)d->M : (long)d->cN * d->cH * d->cW; }
static long x3_cp(const skimi_gemm_desc* d) { return ((d->a_mode == 0 ? (long)d->K : (long)d->cC) + 31) / 32 * 32; }

size_t gemm_x3dma_scratch_bytes(const skimi_gemm_desc* d) {
    return (size_t)x3_rows_in(d) * x3_cp(d) * 4 + 256;   // A records + 256 zero bytes
}

// a desc qualifies when its weights also come as records (W_split), the caller lent scratch for
// the activation records, and the problem fills the chip with 256x256 tiles
bool gemm_x3dma_eligible(const skimi_gemm_desc* d) {
    if (d->prec != SKIMI_PREC_BF16X3 || d->W_split == nullptr) return false;
    if (d->a_dtype == SKIMI_F32) {
        if (d->x3_scratch == nullptr || d->x3_scratch_bytes < gemm_x3dma_scratch_bytes(d)) return false;
        if (d->lda % 4 != 0) return false;
    } else if (d->a_dtype == SKIMI_BF16X3_REC) {
        // the caller's zero page must lie behind the records, within reach of a 32-bit lane offset
        const long rec_bytes = (long)gemm_x3dma_scratch_bytes(d) - 256;
        const long dz = (const char*)d->x3_scratch - (const char*)d->A;
        if (d->x3_scratch == nullptr || d->x3_scratch_bytes < 256 || dz < rec_bytes || dz >= (1ll << 32) - (1 << 28)) return false;
        if (((uintptr_t)d->A & 127) != 0) return false;
    } else {
        return false;
    }
    if (d->store_mode != 0 && d->store_mode != 1) return false;
    if (d->K % 4 != 0) return false;
    if (d->a_mode != 0 && (d->cC % 32 != 0 || d->KH * d->KW > 32)) return false;
    // 32-bit lane offsets from a scalar base
    const long kp = ((long)d->K + 31) / 32 * 32;
    const long bias = d->a_mode == 0 ? 0 : ((long)d->pad * d->cW + d->pad) * x3_cp(d) * 4;
    if ((long)gemm_x3dma_scratch_bytes(d) + bias >= (1ll << 32) || (long)d->N * kp * 4 >= (1ll << 32)) return false;
    // enough 256-row tiles for most of the chip (one workgroup per CU).  Measured in the bench: the level
    // with 172 tiles (M = 43808) is already faster here than on the generic 128x128 kernel (17.38 vs
    // 17.18 frames/s at a threshold of 160 vs 200); below that the generic kernel's smaller tiles win.
    // (SKIMI_X3_MIN_TILES overrides the threshold: the tests run small shapes through these kernels)
    static const bool dyn = getenv("SKIMI_ENV_DYNAMIC") && atoi(getenv("SKIMI_ENV_DYNAMIC"));
    static long min_tiles = -1;
    if (min_tiles < 0 || dyn) min_tiles = getenv("SKIMI_X3_MIN_TILES") ? atol(getenv("SKIMI_X3_MIN_TILES")) : 160;
    const long tiles = cdiv(d->M, 256) * (d->N > 128 ? cdiv(d->N, 256) : 1);
    return d->M >= (min_tiles <= 1 ? 256 : 4096) && d->N >= 96 && tiles >= min_tiles;
}

template <int AMODE, int ABL = 0>
static int launch_x3w4(GemmArgs& a, const X3Rec& pl, hipStream_t st) {
    constexpr size_t lds = 10ull * 128 * 128;
    SKIMI_LDS_OPT_IN((gemm_x3w4_kernel<AMODE, ABL>), lds, "gemm_x3w4");
    a.ntm = (int)cdiv(a.M, 256);
    a.ntn = (int)cdiv(a.N, 256);
    a.splitk = 1;
    hipLaunchKernelGGL((gemm_x3w4_kernel<AMODE, ABL>), dim3(a.ntm * a.ntn), dim3(256), lds, st, a, pl);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

template <int AMODE>
static int launch_x3w4n(GemmArgs& a, const X3Rec& pl, hipStream_t st) {
    constexpr size_t lds = 9ull * 128 * 128;
    SKIMI_LDS_OPT_IN(gemm_x3w4n_kernel<AMODE>, lds, "gemm_x3w4n");
    a.ntm = (int)cdiv(a.M, 256);
    a.ntn = (int)cdiv(a.N, 128);
    a.splitk = 1;
    hipLaunchKernelGGL(gemm_x3w4n_kernel<AMODE>, dim3(a.ntm * a.ntn), dim3(256), lds, st, a, pl);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// d->x3_scratch: the A records (4 bytes per element of the A buffer, rows padded to 32) + 256 zero bytes
int gemm_x3dma_launch(GemmArgs& a, const skimi_gemm_desc* d, hipStream_t st) {
    const long rows_in = x3_rows_in(d), cp = x3_cp(d);
    const int Cw = d->a_mode == 0 ? d->K : d->cC;
    char* rec = (char*)d->x3_scratch;
    char* zpage = rec + rows_in * cp * 4;
    if (d->a_dtype == SKIMI_BF16X3_REC) {   // the producer already wrote records (and cleared its zero page)
        rec = (char*)d->A;
        zpage = (char*)d->x3_scratch;
    } else {
        int rc = split_records_launch((const float*)d->A, d->lda, rows_in, Cw, rec, st, zpage);
        if (rc) return rc;
    }
    X3Rec pl;
    pl.a = rec;
    pl.w = (const char*)d->W_split;
    pl.zero = zpage;
    pl.a_row_bytes = cp * 4;
    pl.w_row_bytes = ((long)d->K + 31) / 32 * 32 * 4;
    pl.a_bias = d->a_mode == 0 ? 0 : ((long)d->pad * d->cW + d->pad) * pl.a_row_bytes;
    a.dbg = 0;
    if (d->N <= 128) {
        if (a.a_mode == 0) return launch_x3w4n<0>(a, pl, st);
        if (a.a_mode == 1) return launch_x3w4n<1>(a, pl, st);
        return launch_x3w4n<2>(a, pl, st);
    }
#ifdef SKIMI_ABLATIONS   // timing ablations (wrong results): only in a -DSKIMI_ABLATIONS build
    if (a.a_mode == 0) {
        static const int abl = getenv("SKIMI_X3_ABL") ? atoi(getenv("SKIMI_X3_ABL")) : 0;
        switch (abl) {
            case 1: return launch_x3w4<0, 1>(a, pl, st);
            case 2: return launch_x3w4<0, 2>(a, pl, st);
            case 3: return launch_x3w4<0, 3>(a, pl, st);
            case 5: return launch_x3w4<0, 5>(a, pl, st);
            case 6: return launch_x3w4<0, 6>(a, pl, st);
            default: break;
        }
    }
    if (a.a_mode == 2) {
        static const int abl = getenv("SKIMI_X3_ABL") ? atoi(getenv("SKIMI_X3_ABL")) : 0;
        switch (abl) {
            case 1: return launch_x3w4<2, 1>(a, pl, st);
            case 2: return launch_x3w4<2, 2>(a, pl, st);
            case 3: return launch_x3w4<2, 3>(a, pl, st);
            case 5: return launch_x3w4<2, 5>(a, pl, st);
            case 6: return launch_x3w4<2, 6>(a, pl, st);
            default: break;
        }
    }
#endif
    if (a.a_mode == 0) return launch_x3w4<0>(a, pl, st);
    if (a.a_mode == 1) return launch_x3w4<1>(a, pl, st);
    return launch_x3w4<2>(a, pl, st);
}

}  // namespace skimi

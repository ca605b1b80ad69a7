// 3 x 3 / stride 1 / pad 1 convolution to 128 output channels on single-term 16-bit operands (bf16 or fp16), channels-last:
// the convolutions of the track head's DPT feature extractor (dpt_head.py:261-291 with features = 128, run inside
// autocast: vggt.py:85-91).
//
// As an implicit-gather GEMM (gemm.hip, a_mode 1) these layers are bound by the L2 -> CU fill rate, not by the matrix pipe:
// N = 128 leaves one column tile, so a 128 x 128 tile stages 32 KB per 512 MFMA cycles (64 B/clk against the ~45 B/clk a CU's
// port delivers), every input pixel crosses it nine times, and K = 9 x 128 is too short for a 256-row LDS-DMA tile to
// amortise its prologue (380-460 TFLOP/s on either).  Here a workgroup owns a 16 x 16 pixel tile x all 128
// output channels, like conv_direct_n32_kernel:
//   * the 18 x 18 halo window of one 32-channel slice is brought into LDS ONCE by LDS-DMA and read by all nine taps
//     (7 x less gather traffic); the weights of one (slice, tap) -- [128 cout][32 ch] = 8 KiB, rows of the packed
//     [N][tap][C] matrix as they lie -- stream through a 4-deep ring, three taps ahead;
//   * 4 waves, wave = (pixel half, cout half): 128 pixels x 64 couts = 4 x 2 accumulator tiles of 32 x 32; per tap and
//     16-channel k-step 4 A + 2 B fragment reads feed 8 MFMAs;
//   * one raw s_barrier per tap and ONE counted s_waitcnt: every wave issues the same number of DMA instructions per
//     step (2 weight pieces, + 1 window piece of the next slice during taps 0..5; out-of-range pieces are clamped to a
//     valid duplicate), so "the weights of this tap have landed" is a compile-time vmcnt;
//   * 78 KiB of LDS and <= 256 registers: two workgroups per CU, one's prologue / epilogue under the other's taps;
//   * the epilogue goes through the shared GemmArgs epilogue (bias, activation, residuals, 16-bit or fp32 rows).
// LDS rows are 64 B (32 channels) per pixel / per cout; the 16-B chunk of a row is XOR-swizzled by (column >> 2) & 3 of the
// window column / cout, and the window's row pitch is 20 pixels: conflict-free ds_read_b128 for every tap shift (checked
// against the instruction's four 16-lane groups, MI355X_MICROARCH.md §LDS).
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace skimi {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ uint4 g_cw_zero_page[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

constexpr int CW_WROW = 20;                              // window row pitch in pixels: 18 + 2 unused, a multiple of the 4 pixels of a
                                                         // 256-B bank row -- with a pitch of 18 the second row of a fragment sits two pixels
                                                         // off and every A read is a 2-way conflict (computed per ds_read_b128 lane group)
constexpr int CW_NPIECE = 23;                            // 18 x 20 px in DMA pieces of 16
constexpr int CW_WIN = CW_NPIECE * 1024;
constexpr int CW_TAP = 128 * 64;                         // one (slice, tap) of the weights: [cout 128][32 ch]
constexpr int CW_RING = 4;
constexpr int CW_LDS = 2 * CW_WIN + CW_RING * CW_TAP;    // 79872 B: two workgroups per CU
constexpr int CW_STG_ROW = 72;                           // epilogue staging: floats per pixel row (64 + pad: the two lane halves hit different banks)

#define CW_BAR()                              \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields at their maxima)
#define CW_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))

// DMA instructions a wave issues in the step of tap t: two weight pieces, and a window piece while t < 6
__host__ __device__ constexpr int cw_ops(int t) { return 2 + (t < 6 ? 1 : 0); }
// outstanding DMA instructions allowed when tap t starts: everything issued after the weights of this tap
// (= the window piece of step q-3, if it had one, and all of steps q-2, q-1)
__host__ __device__ constexpr int cw_allowed(int t) {
    return ((t + 6) % 9 < 6 ? 1 : 0) + cw_ops((t + 7) % 9) + cw_ops((t + 8) % 9);
}

static_assert(cw_allowed(0) == 4 && cw_allowed(1) == 5 && cw_allowed(2) == 6 && cw_allowed(3) == 7 && cw_allowed(4) == 7 &&
              cw_allowed(5) == 7 && cw_allowed(6) == 7 && cw_allowed(7) == 6 && cw_allowed(8) == 5, "vmcnt table of conv_win128_kernel");

// ABL != 0: timing ablations (SKIMI_CONV_WIN_ABL in a -DSKIMI_ABLATIONS build; results are wrong): 1 no weight DMA in the loop,
// 2 no window DMA in the loop, 4 no MFMA, 8 no fragment reads, 16 no barrier
template <bool F16, int ABL>
__global__ __launch_bounds__(256, 2) void conv_win128_kernel(const GemmArgs p, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int ph = wave >> 1, ch = wave & 1;   // pixel half (tile rows 8 ph ..), cout half

    // tile id contiguous per XCD (blocks b and b + 8 share an XCD): neighbouring tiles share their halo rows in one L2
    int bid;
    {
        const int nblk = gridDim.x, b = blockIdx.x, xcd = b & 7;
        const int qn = nblk >> 3, rn = nblk & 7;
        bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (b >> 3);
    }
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int f = bid / tiles_y;
    const int y0 = ty * 16, x0 = tx * 16;
    const int H = p.cH, Wd = p.cW, C = p.cC;
    const int nsl = C >> 5;
    const unsigned short* A = (const unsigned short*)p.A;
    const unsigned short* Wt = (const unsigned short*)p.W;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_cw_zero_page);

    // ---- DMA plan ----
    // window: pieces of 16 pixels (1 KiB), 23 per slice; this wave issues piece min(4 j + wave, 22) in tap j < 6
    int win_off[6];   // element offset of this lane's source chunk (slice 0), -1 = zero page (outside the image / the window)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = min(4 * j + wave, CW_NPIECE - 1);
        const int px = 16 * q + (lane >> 2);
        const int wy = px / CW_WROW, wx = px - wy * CW_WROW;
        const int y = y0 - 1 + wy, x = x0 - 1 + wx;
        const int sc = (lane & 3) ^ ((wx >> 2) & 3);
        win_off[j] = (wy < 18 && wx < 18 && y >= 0 && y < H && x >= 0 && x < Wd) ? (int)((((long)f * H + y) * Wd + x) * p.lda) + sc * 8 : -1;
    }
    // weights: pieces of 16 couts, 8 per tap; this wave issues pieces wave and wave + 4
    int w_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * (wave + 4 * j) + (lane >> 2);
        w_off[j] = row * (int)p.ldw + (((lane & 3) ^ ((row >> 2) & 3)) << 3);
    }
    char* const ring = smem + 2 * CW_WIN;
    auto issue_w = [&](int s, int tap, int slot) {   // (slice, tap) -> ring slot
        const unsigned short* src = Wt + tap * C + s * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + w_off[j]), (lds_void*)(ring + slot * CW_TAP + (wave + 4 * j) * 1024), 16, 0, 0);
    };
    auto issue_win = [&](int j, int s, int buf) {
        const unsigned short* src = win_off[j] >= 0 ? A + win_off[j] + s * 32 : zero;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + buf * CW_WIN + min(4 * j + wave, CW_NPIECE - 1) * 1024), 16, 0, 0);
    };

    // ---- fragment addresses ----
    // A: lane (l31, lh) of pixel tile i reads window pixel (8 ph + 2 i + (l31 >> 4) + dy, (l31 & 15) + dx), channels 16 ks + 8 lh ..
    int a_base[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int wx = (l31 & 15) + dx;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            a_base[dx][ks] = ((8 * ph + (l31 >> 4)) * CW_WROW + wx) * 64 + (((2 * ks + lh) ^ ((wx >> 2) & 3)) << 4);
    }
    // B: cout 64 ch + 32 j + l31
    int b_base[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_base[ks] = (ch * 64 + l31) * 64 + (((2 * ks + lh) ^ ((l31 >> 2) & 3)) << 4);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: the window of slice 0 (6 pieces per wave, duplicates included), then the weights of taps 0, 1, 2 ----
#pragma unroll
    for (int j = 0; j < 6; ++j) issue_win(j, 0, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);

    for (int s = 0; s < nsl; ++s) {
        const char* win = smem + (s & 1) * CW_WIN;
        const int sn = min(s + 1, nsl - 1);   // past the end: a harmless reload (keeps the DMA count per step fixed)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // this wave's pieces of tap t (and, at t = 0, of the window) have landed: vmcnt(cw_allowed(t)), an immediate
            switch (t) {
                case 0: CW_VMCNT(4); break;
                case 1: case 8: CW_VMCNT(5); break;
                case 2: case 7: CW_VMCNT(6); break;
                default: CW_VMCNT(7); break;
            }
            if constexpr (!(ABL & 16)) CW_BAR();                          // ... and everybody else's; all waves are done with step q - 1
            {
                const int t3 = (t + 3) % 9;
                const int s3 = min(s + (t + 3 >= 9 ? 1 : 0), nsl - 1);
                if constexpr (!(ABL & 1)) issue_w(s3, t3, (s * 9 + t + 3) & 3);
                if constexpr (!(ABL & 2)) { if (t < 6) issue_win(t, sn, (s + 1) & 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
            const char* wt = ring + ((s * 9 + t) & 3) * CW_TAP;
            const int dy = t / 3, dx = t - 3 * dy;
            // all twelve fragments of the tap are requested up front: the second k-step's reads land under the first's MFMAs
            bf16x8 a[2][4], b[2][2];
            if constexpr ((ABL & 8) != 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[ks][j] = (bf16x8)(short)(t + j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[ks][i] = (bf16x8)(short)(t + i);
                }
            } else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) b[ks][j] = *reinterpret_cast<const bf16x8*>(wt + b_base[ks] + j * 2048);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(win + a_base[dx][ks] + (2 * i + dy) * (CW_WROW * 64));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr ((ABL & 4) != 0) acc[i][j][0] += (float)a[ks][i][0] * (float)b[ks][j][0];
                        else acc[i][j] = mfma_32x32x16<F16>(a[ks][i], b[ks][j], acc[i][j]);
                    }
            // schedule: the six reads of k-step 0, then one read of k-step 1 behind each of the first six MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
        }
    }
    CW_VMCNT(0);   // the clamped tail reloads
    CW_BAR();      // every wave is past its last fragment read: the whole LDS is free

    // ---- epilogue: 32 pixels x 64 couts at a time through a wave-private LDS tile -> row-contiguous 16-B accesses ----
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * CW_STG_ROW);
    const int n = ch * 64 + 4 * (lane & 15);
#pragma unroll   // (a runtime i would index the accumulator array dynamically and send it to scratch)
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * CW_STG_ROW + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the tile is written (one wave: LDS executes its accesses in order)
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
            const int row_l = it * 4 + (lane >> 4);
            const int y = y0 + 8 * ph + 2 * i + (row_l >> 4), x = x0 + (row_l & 15);
            const float4 v = *reinterpret_cast<const float4*>(&stg[row_l * CW_STG_ROW + 4 * (lane & 15)]);
            if (y < H && x < Wd) {
                const int m = (f * H + y) * Wd + x;
                const RowMap rm = row_map(p, m);
                store_four(p, rm, n, v);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // the reads are done before the next pass overwrites the tile
    }
}

// SKIMI_CONV_WIN: 1 (default) where the shape pays, 0 never (A/B timing), 2 wherever the kernel applies (tests)
static int conv_win_mode() {
    static const bool dyn = getenv("SKIMI_ENV_DYNAMIC") && atoi(getenv("SKIMI_ENV_DYNAMIC"));
    static int mode = -1;
    if (mode < 0 || dyn) mode = getenv("SKIMI_CONV_WIN") ? atoi(getenv("SKIMI_CONV_WIN")) : 1;
    return mode;
}

bool conv_win_eligible(const skimi_gemm_desc* d) {
    const int mode = conv_win_mode();
    if (mode == 0) return false;
    const bool h16 = d->prec == SKIMI_PREC_F16;
    const int dt = h16 ? SKIMI_F16 : SKIMI_BF16;
    if (!(d->prec == SKIMI_PREC_BF16 || h16) || d->a_dtype != dt || d->w_dtype != dt) return false;
    if (d->a_mode != 1 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1) return false;
    if (d->OH != d->cH || d->OW != d->cW || d->N != 128 || d->cC % 32 != 0 || d->cC < 32) return false;
    if (d->store_mode != 0 || d->out_records != nullptr || d->out == nullptr || d->force_splitk > 0) return false;
    if (d->lda % 8 != 0 || d->ldw % 8 != 0 || (((uintptr_t)d->A | (uintptr_t)d->W) & 15) != 0) return false;
    if ((long)d->M * d->lda >= (1l << 31) || (long)d->N * d->ldw >= (1l << 31)) return false;
    // the shared epilogue's 16-byte accesses (what gemm_dispatch calls vec4)
    auto al = [](const void* q, int a) { return ((uintptr_t)q & (a - 1)) == 0; };
    const int oa = d->out_dtype == SKIMI_F32 ? 16 : 8, o2a = d->out_dtype == SKIMI_F32 ? 8 : 16;
    const int ra = d->resid_dtype == SKIMI_F32 ? 16 : 8;
    if (d->ldo % 4 != 0 || !al(d->out, oa) || (d->out2 && (d->ldo2 % 4 != 0 || !al(d->out2, o2a)))) return false;
    if ((d->bias && !al(d->bias, 16)) || (d->gamma && !al(d->gamma, 16))) return false;
    if ((d->resid && (d->ldr % 4 != 0 || !al(d->resid, ra))) || (d->resid2 && (d->ldr2 % 4 != 0 || !al(d->resid2, ra)))) return false;
    if (mode >= 2) return true;
    // worth it where whole tiles dominate and the launch fills the chip twice over
    const long tiles = (long)d->cN * cdiv(d->cH, 16) * cdiv(d->cW, 16);
    return tiles >= 512 && (double)d->cH * d->cW >= 0.7 * 256.0 * (double)(cdiv(d->cH, 16) * cdiv(d->cW, 16));
}

int conv_win_launch(GemmArgs& a, hipStream_t st) {
    const int tiles_x = (int)cdiv(a.cW, 16), tiles_y = (int)cdiv(a.cH, 16);
    const long nblk = (long)a.cN * tiles_x * tiles_y;
    SKIMI_CHECK_ARG(nblk < (1l << 31), "conv_win: grid too large");
    a.vec4 = 1;
    a.splitk = 1;
    a.partial = nullptr;
    // SKIMI_CONV_WIN_LDSPAD (bytes of unused dynamic LDS): occupancy experiments, e.g. 8192 -> one workgroup per CU
    static const int pad = getenv("SKIMI_CONV_WIN_LDSPAD") ? atoi(getenv("SKIMI_CONV_WIN_LDSPAD")) : 0;
    const int lds = CW_LDS + pad;
#define CW_GO(F16_, ABL_)                                                                                                      \
    do {                                                                                                                       \
        SKIMI_LDS_OPT_IN((conv_win128_kernel<F16_, ABL_>), lds, "conv_win");                                                   \
        hipLaunchKernelGGL((conv_win128_kernel<F16_, ABL_>), dim3((unsigned)nblk), dim3(256), lds, st, a, tiles_x, tiles_y);   \
    } while (0)
#ifdef SKIMI_ABLATIONS
    const int abl = getenv("SKIMI_CONV_WIN_ABL") ? atoi(getenv("SKIMI_CONV_WIN_ABL")) : 0;
    switch (abl) {
        case 1: CW_GO(true, 1); break;
        case 2: CW_GO(true, 2); break;
        case 3: CW_GO(true, 3); break;
        case 4: CW_GO(true, 4); break;
        case 8: CW_GO(true, 8); break;
        case 12: CW_GO(true, 12); break;
        case 16: CW_GO(true, 16); break;
        case 7: CW_GO(true, 7); break;
        case 15: CW_GO(true, 15); break;
        case 31: CW_GO(true, 31); break;
        default: CW_GO(true, 0);
    }
#else
    if (a.f16) CW_GO(true, 0);
    else CW_GO(false, 0);
#endif
#undef CW_GO
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

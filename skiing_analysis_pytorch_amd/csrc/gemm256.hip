// 256x256-tile bf16 MFMA contraction with direct global->LDS staging (global_load_lds_dwordx4),
// the fast path of the aggregator / DINOv2 linears (plain rows, bf16 A and W, K % 64 == 0).
//
// Why a second kernel: the generic 128x128 kernel moves 64 FLOP per byte staged from L2; at
// MFMA speed that asks the 8 XCD L2s for more than they deliver.  A 256x256 tile doubles the
// reuse (128 FLOP/B), and LDS-DMA staging removes the VGPR round trip and the ds_write pass.
//   * 8 waves as 2(M) x 4(N), each 128x64 = 4x2 v_mfma_f32_32x32x16_bf16 tiles (128 acc VGPRs);
//   * BK = 64: A and W tiles are [256][128 B] row images, two buffers = 128 KiB LDS, 1 block/CU;
//   * a wave-instruction of global_load_lds writes 1 KiB = 8 rows x 128 B linearly, so the
//     bank-conflict swizzle (16-B chunk ^= (row>>1)&7) is applied to the per-lane SOURCE
//     address and again on the ds_read_b128 (both sides or neither);
//   * two-phase loop: issue tile t+1's DMA, MFMA tile t, vmcnt(0) + barrier;
//   * epilogue through per-wave LDS slabs -> row-contiguous 16-B stores (shared with gemm.hip).
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

namespace skimi {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// EPI: 0 = generic epilogue (runtime flags); 1 = bias -> bf16 rows (qkv); 3 = bias, GELU -> bf16
//      rows (fc1); 2 = bias, LayerScale, fp32 residual -> fp32 rows, plain row map (proj, fc2)
template <int MT, int EPI>   // M tiles of 32 rows per wave: BM = 64 * MT (192 or 256)
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs p) {
    constexpr int BM = 64 * MT, BN = 256, BK = 64;
    constexpr int RB = 128;                    // LDS row bytes
    constexpr int A_TILE = BM * RB, W_TILE = BN * RB, BUF = A_TILE + W_TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
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
    const int nkt = p.K / BK;
    if ((p.dbg & 16) && blockIdx.x < 256) {
        // experiment: de-phase the CUs so epilogue store bursts overlap other CUs' main loops
        const long long t0 = wall_clock64();
        const long long wait = (long long)((blockIdx.x >> 3) & 3) * (nkt * 50 + 300) / 4;
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }

    // staging: wave-instruction j of this wave covers tile rows 8*(4*wave + j) .. +7;
    // lane -> (row = lane>>3, LDS chunk = lane&7), source chunk = LDS chunk ^ ((row>>1)&7)
    const unsigned short* A = (const unsigned short*)p.A;
    const unsigned short* W = (const unsigned short*)p.W;
    const unsigned short* a_src[MT];
    const unsigned short* w_src[4];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int row = 8 * (MT * wave + j) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        a_src[j] = A + (long)min(m0 + row, p.M - 1) * p.lda + c * 8;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (4 * wave + j) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        w_src[j] = W + (long)min(n0 + row, p.N - 1) * p.ldw + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* ab = smem + buf * BUF + (MT * wave) * 8 * RB;
        char* wb = smem + buf * BUF + A_TILE + (4 * wave) * 8 * RB;
#pragma unroll
        for (int j = 0; j < MT; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(a_src[j] + kt * BK), (lds_void*)(ab + j * 8 * RB), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(w_src[j] + kt * BK), (lds_void*)(wb + j * 8 * RB), 16, 0, 0);
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0, 0);
    __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt && !(p.dbg & 1)) stage(cur ^ 1, kt + 1);
        const char* ab = smem + cur * BUF;
        const char* wb = ab + A_TILE;
        // fragment reads are software-pipelined one k-step ahead of the MFMAs (two register sets,
        // pinned by sched_barrier): the ds_read latency of step s+1 hides under step s's 8 MFMAs
        bf16x8 af[2][MT], wf[2][2];
        auto load_frags = [&](int s, int set) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = wr * (32 * MT) + i * 32 + l31;
                af[set][i] = *reinterpret_cast<const bf16x8*>(ab + row * RB + (((2 * s + lh) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wc * 64 + j * 32 + l31;
                wf[set][j] = *reinterpret_cast<const bf16x8*>(wb + row * RB + (((2 * s + lh) ^ ((row >> 1) & 7)) << 4));
            }
        };
        if (!(p.dbg & 2)) {
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep set 0's reads ahead of set 1's: counted lgkmcnt
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) load_frags(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], wf[s & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        __syncthreads();
    }

    // ---- epilogue: MT passes of 32 rows through this wave's private 8-KiB LDS slab ----
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    const int n = n0 + wc * 64 + 4 * (lane & 15);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + j * 32 + l31] = acc[i][j][r];
        // same-wave LDS write -> read: DS ops of one wave execute in order; the compiler only
        // needs to keep them in program order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        if (EPI != 0) {
            // straight-line fast path: all 8 row reads in flight, column constants hoisted
            float4 v[8];
#pragma unroll
            for (int it = 0; it < 8; ++it)
                v[it] = *reinterpret_cast<const float4*>(&stg[(it * 4 + (lane >> 4)) * 64 + 4 * (lane & 15)]);
            if (n < p.N) {
                const float4 bs = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0, 0, 0, 0);
                float4 gm = make_float4(1, 1, 1, 1);
                if (EPI == 2) gm = *reinterpret_cast<const float4*>(p.gamma + n);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = m0 + wr * (32 * MT) + i * 32 + it * 4 + (lane >> 4);
                    if (m < p.M && !(p.dbg & 4)) {
                        float y0 = v[it].x + bs.x, y1 = v[it].y + bs.y, y2 = v[it].z + bs.z, y3 = v[it].w + bs.w;
                        if (EPI == 1 || EPI == 3) {
                            if (EPI == 3) { y0 = gelu_erf(y0); y1 = gelu_erf(y1); y2 = gelu_erf(y2); y3 = gelu_erf(y3); }
                            bf16x4 hb;
                            hb[0] = (short)f2bf(y0); hb[1] = (short)f2bf(y1); hb[2] = (short)f2bf(y2); hb[3] = (short)f2bf(y3);
                            bf16x4* dst = reinterpret_cast<bf16x4*>((unsigned short*)p.out + (long)m * p.ldo + n);
                            if (p.dbg & 8) __builtin_nontemporal_store(hb, dst); else *dst = hb;
                        } else {
                            const float4 r = *reinterpret_cast<const float4*>((const float*)p.resid + (long)m * p.ldr + n);
                            f32x4 ov = {r.x + gm.x * y0, r.y + gm.y * y1, r.z + gm.z * y2, r.w + gm.w * y3};
                            f32x4* dst = reinterpret_cast<f32x4*>((float*)p.out + (long)m * p.ldo + n);
                            if (p.dbg & 8) __builtin_nontemporal_store(ov, dst); else *dst = ov;
                        }
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < 8; ++it) {
                const int row_l = it * 4 + (lane >> 4);
                const int m = m0 + wr * (32 * MT) + i * 32 + row_l;
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
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
}

// which compile-time epilogue serves this launch (0 = generic)
static int epi_kind(const GemmArgs& a) {
    const bool plain = a.vec4 && a.store_mode == 0 && a.out_rpb == 0 && a.out_off == 0 && a.out2 == nullptr &&
                       a.resid2 == nullptr && a.post_act == SKIMI_ACT_NONE && a.N % 4 == 0;
    if (!plain) return 0;
    if (a.out_dtype == SKIMI_BF16 && a.gamma == nullptr && a.resid == nullptr) {
        if (a.act == SKIMI_ACT_NONE) return 1;
        if (a.act == SKIMI_ACT_GELU) return 3;
        return 0;
    }
    if (a.out_dtype == SKIMI_F32 && a.gamma != nullptr && a.resid != nullptr && a.resid_dtype == SKIMI_F32 &&
        a.resid_rpb == 0 && a.resid_off == 0 && a.act == SKIMI_ACT_NONE)
        return 2;
    return 0;
}

bool gemm256_eligible(const skimi_gemm_desc* d) {
    return d->prec == SKIMI_PREC_BF16 && d->a_dtype == SKIMI_BF16 && d->w_dtype == SKIMI_BF16 && d->a_mode == 0 &&
           d->store_mode == 0 && d->K % 64 == 0 && d->lda % 8 == 0 && d->ldw % 8 == 0 && d->M >= 2048 &&
           d->N >= 512 && (((uintptr_t)d->A | (uintptr_t)d->W) & 15) == 0;
}

template <int MT, int EPI>
static int launch256(GemmArgs& a, hipStream_t st) {
    constexpr size_t lds = 2ull * (64 * MT + 256) * 128;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<MT, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(gemm256) failed: %s", hipGetErrorString(e));
            return SKIMI_ERR_HIP;
        }
        attr_done = true;
    }
    a.ntm = (int)cdiv(a.M, 64 * MT);
    a.ntn = (int)cdiv(a.N, 256);
    a.splitk = 1;
    hipLaunchKernelGGL((gemm256_kernel<MT, EPI>), dim3(a.ntm * a.ntn), dim3(512), lds, st, a);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// pick the tile height (192 or 256 rows) that wastes the fewest CU-rounds for this shape:
// one workgroup per CU, so time ~ ceil(tiles / 256) * (rows per tile)
int gemm256_launch(GemmArgs& a, hipStream_t st) {
    static const int dbg = getenv("SKIMI_GEMM256_ABL") ? atoi(getenv("SKIMI_GEMM256_ABL")) : 0;
    a.dbg = dbg;
    auto cost = [&](int bm) {
        const long tiles = cdiv(a.M, bm) * cdiv(a.N, 256);
        return (double)cdiv(tiles, 256) * bm;
    };
    const int epi = epi_kind(a);
    if (cost(192) < cost(256)) {
        if (epi == 1) return launch256<3, 1>(a, st);
        if (epi == 3) return launch256<3, 3>(a, st);
        if (epi == 2) return launch256<3, 2>(a, st);
        return launch256<3, 0>(a, st);
    }
    if (epi == 1) return launch256<4, 1>(a, st);
    if (epi == 3) return launch256<4, 3>(a, st);
    if (epi == 2) return launch256<4, 2>(a, st);
    return launch256<4, 0>(a, st);
}

}  // namespace skimi

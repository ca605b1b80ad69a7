// bf16x3 (fp32-accurate) MFMA contraction with LDS-DMA staging, for the fp32-path heads (DPT
// convs, camera / track linears) whose operands live in HBM as PRE-SPLIT bf16 planes:
//     x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi)      acc += Ahi*Whi + Ahi*Wlo + Alo*Whi
// The generic kernel (gemm.hip, NSPLIT = 3) splits fp32 operands in registers for every tile that
// touches them (a 3x3 conv re-splits each activation 9 x N/128 times) and stages through VGPRs.
// Here the split is done once (weights at finalize, activations by split_planes_kernel) and the
// four planes stream global -> LDS with global_load_lds_dwordx4:
//   * tile 256 x (128 | 256), BK = 32, 8 waves as 2(M) x 4(N); per k-tile and wave 4 x NT x 2 x 3
//     MFMAs (48 at NT = 2) for 64 KiB of DMA: MFMA-bound, unlike the 1-MFMA bf16 case;
//   * LDS rows are 64 B (BK = 32); a wave-instruction writes 16 rows x 64 B linearly; the
//     16-B-chunk swizzle (chunk ^= (row>>2)&3) is applied to the per-lane SOURCE address and to
//     the ds_read_b128 (both or neither);
//   * implicit-im2col gather = per-lane source pixel address; padding taps read a zeroed page;
//   * epilogue shared with gemm.hip (bias / act / LayerScale / residuals / row remaps).
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

struct X3Planes {
    const unsigned short *a_hi, *a_lo;   // [rows, lda] bf16
    const unsigned short *w_hi, *w_lo;   // [N, ldw] bf16
    const unsigned short* zero;          // zero bytes (padding taps, K tail)
};

__device__ uint4 g_zero_page[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

template <int NT>   // N tiles of 32 columns per wave: BN = 128 * NT
__global__ __launch_bounds__(512, 2) void gemm_x3dma_kernel(const GemmArgs p, const X3Planes pl) {
    constexpr int MT = 4, BM = 256, BN = 128 * NT, BK = 32;
    constexpr int RB = 64;                        // LDS row bytes
    constexpr int A_PLANE = BM * RB, W_PLANE = BN * RB;
    constexpr int BUF = 2 * (A_PLANE + W_PLANE);  // [A_hi | A_lo | W_hi | W_lo]
    constexpr int A_INS = BM / 16 / 8;            // staging wave-instructions per plane per wave (16 rows each)
    constexpr int W_INS = (BN / 16 + 7) / 8;      // 2 (BN 256) or 1 (BN 128)
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
    const int nkt = (p.K + BK - 1) / BK;

    // ---- staging coordinates: lane -> (row = lane>>2 of the 16-row group, LDS chunk = lane&3) ----
    const int srow = lane >> 2, schunk = lane & 3;
    int a_row[A_INS];          // tile row
    long a_base[A_INS];        // element offset of the row (plain) / of the window's top-left pixel (conv), + chunk
    int a_iy0[A_INS], a_ix0[A_INS];
    int a_c[A_INS];            // source chunk (swizzled)
#pragma unroll
    for (int j = 0; j < A_INS; ++j) {
        const int row = 16 * (A_INS * wave + j) + srow;
        a_row[j] = row;
        a_c[j] = schunk ^ ((row >> 2) & 3);
        const int m = min(m0 + row, p.M - 1);
        if (p.a_mode == 0) {
            a_base[j] = (long)m * p.lda;
            a_iy0[j] = a_ix0[j] = 0;
        } else {
            const int ohw = p.OH * p.OW;
            const int img = m / ohw;
            const int rem = m - img * ohw;
            const int oy = rem / p.OW;
            a_iy0[j] = oy * p.stride - p.pad;
            a_ix0[j] = (rem - oy * p.OW) * p.stride - p.pad;
            // the tap and channel-slice part of the address is wave-uniform and added per K-tile
            a_base[j] = (((long)img * p.cH + a_iy0[j]) * p.cW + a_ix0[j]) * p.lda + a_c[j] * 8;
        }
    }
    long w_base[W_INS];
    int w_c[W_INS];
    bool w_on[W_INS];
#pragma unroll
    for (int j = 0; j < W_INS; ++j) {
        const int grp = W_INS * wave + j;          // 16-row group of the W tile
        const int row = 16 * grp + srow;
        w_on[j] = grp < BN / 16;
        w_c[j] = schunk ^ ((row >> 2) & 3);
        w_base[j] = (long)min(n0 + row, p.N - 1) * p.ldw;
    }

    auto stage = [&](int buf, int kt) {
        const int k0 = kt * BK;
        int tap_dy = 0, tap_dx = 0, cin0 = k0;
        if (p.a_mode != 0) {
            int tap = k0 / p.cC;
            cin0 = k0 - tap * p.cC;
            if (p.a_mode == 2) {   // slice-major K: k = ((c / 32) * taps + tap) * 32 + c % 32
                const int u = k0 >> 5, nt = p.KH * p.KW;
                const int cs = u / nt;
                tap = u - cs * nt;
                cin0 = cs * 32;
            }
            const int ky = tap / p.KW;
            tap_dy = ky * p.dil;
            tap_dx = (tap - ky * p.KW) * p.dil;
        }
        char* base = smem + buf * BUF;
        const long tap_off = ((long)tap_dy * p.cW + tap_dx) * p.lda + cin0;   // wave-uniform (scalar unit)
#pragma unroll
        for (int j = 0; j < A_INS; ++j) {
            const int kk = k0 + a_c[j] * 8;
            bool ok = kk < p.K;
            long off;
            if (p.a_mode == 0) {
                off = a_base[j] + kk;
            } else {
                const int iy = a_iy0[j] + tap_dy, ix = a_ix0[j] + tap_dx;
                ok = ok && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
                off = a_base[j] + tap_off;
            }
            const unsigned short* sh = ok ? pl.a_hi + off : pl.zero;
            const unsigned short* sl = ok ? pl.a_lo + off : pl.zero;
            char* dst = base + (A_INS * wave + j) * 16 * RB;
            __builtin_amdgcn_global_load_lds((gbl_void*)sh, (lds_void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)sl, (lds_void*)(dst + A_PLANE), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < W_INS; ++j) {
            if (!w_on[j]) continue;   // wave-uniform
            const int kk = k0 + w_c[j] * 8;
            const bool ok = kk < p.K;
            const unsigned short* sh = ok ? pl.w_hi + w_base[j] + kk : pl.zero;
            const unsigned short* sl = ok ? pl.w_lo + w_base[j] + kk : pl.zero;
            char* dst = base + 2 * A_PLANE + (W_INS * wave + j) * 16 * RB;
            __builtin_amdgcn_global_load_lds((gbl_void*)sh, (lds_void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)sl, (lds_void*)(dst + W_PLANE), 16, 0, 0);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
        const char* ah = smem + cur * BUF;
        const char* wh = ah + 2 * A_PLANE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a_hi[MT], a_lo[MT], w_hi[NT], w_lo[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = wr * 128 + i * 32 + l31;
                const int o = row * RB + (((2 * s + lh) ^ ((row >> 2) & 3)) << 4);
                a_hi[i] = *reinterpret_cast<const bf16x8*>(ah + o);
                a_lo[i] = *reinterpret_cast<const bf16x8*>(ah + A_PLANE + o);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = wc * (32 * NT) + j * 32 + l31;
                const int o = row * RB + (((2 * s + lh) ^ ((row >> 2) & 3)) << 4);
                w_hi[j] = *reinterpret_cast<const bf16x8*>(wh + o);
                w_lo[j] = *reinterpret_cast<const bf16x8*>(wh + W_PLANE + o);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i], w_hi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], w_lo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], w_hi[j], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }

    // ---- epilogue: per-wave 32-row passes through a private LDS slab ----
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 32 * NT);
    constexpr int WN = 32 * NT;            // wave's column span
    constexpr int LPR = WN / 4;            // lanes per row (8 or 16)
    constexpr int RPI = 64 / LPR;          // rows per wave access (8 or 4)
    const int n = n0 + wc * WN + 4 * (lane % LPR);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * WN + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row_l = it * RPI + lane / LPR;
            const int m = m0 + wr * 128 + i * 32 + row_l;
            if (m >= p.M || n >= p.N) continue;
            const float4 v = *reinterpret_cast<const float4*>(&stg[row_l * WN + 4 * (lane % LPR)]);
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

size_t gemm_x3dma_scratch_bytes(const skimi_gemm_desc* d);

// a desc qualifies when its weights also come as pre-split planes (W_split), the caller lent
// scratch for the activation planes, and the problem fills the chip with 256-row tiles
bool gemm_x3dma_eligible(const skimi_gemm_desc* d) {
    if (d->prec != SKIMI_PREC_BF16X3 || d->W_split == nullptr || d->a_dtype != SKIMI_F32) return false;
    if (d->x3_scratch == nullptr || d->x3_scratch_bytes < gemm_x3dma_scratch_bytes(d)) return false;
    if (d->store_mode != 0 && d->store_mode != 1) return false;
    if (d->K % 8 != 0 || d->lda % 8 != 0 || d->ldw % 8 != 0) return false;
    if (d->a_mode != 0 && d->cC % 32 != 0) return false;
    return d->M >= 4096 && d->N >= 96;
}

template <int NT>
static int launch_x3(GemmArgs& a, const X3Planes& pl, hipStream_t st) {
    constexpr size_t lds = 2ull * 2 * (256 + 128 * NT) * 64;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3dma_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(gemm_x3dma) failed: %s", hipGetErrorString(e));
            return SKIMI_ERR_HIP;
        }
        attr_done = true;
    }
    a.ntm = (int)cdiv(a.M, 256);
    a.ntn = (int)cdiv(a.N, 128 * NT);
    a.splitk = 1;
    hipLaunchKernelGGL(gemm_x3dma_kernel<NT>, dim3(a.ntm * a.ntn), dim3(512), lds, st, a, pl);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

size_t gemm_x3dma_scratch_bytes(const skimi_gemm_desc* d) {
    const long rows_in = d->a_mode == 0 ? (long)d->M : (long)d->cN * d->cH * d->cW;
    const int Cw = d->a_mode == 0 ? d->K : d->cC;
    return (size_t)rows_in * Cw * 4;
}

// d->x3_scratch: 2 * rows_in * Cw * 2 bytes for the activation planes (hi then lo)
int gemm_x3dma_launch(GemmArgs& a, const skimi_gemm_desc* d, hipStream_t st) {
    void* a_planes = d->x3_scratch;
    // rows of the A buffer that the launch can touch
    const long rows_in = d->a_mode == 0 ? (long)d->M : (long)d->cN * d->cH * d->cW;
    // the planes are stored densely with the row stride of the source's used width
    const int Cw = d->a_mode == 0 ? d->K : d->cC;
    unsigned short* hi = (unsigned short*)a_planes;
    unsigned short* lo = hi + rows_in * Cw;
    int rc = split_planes_launch((const float*)d->A, d->lda, rows_in, Cw, hi, lo, st);
    if (rc) return rc;
    X3Planes pl;
    pl.a_hi = hi;
    pl.a_lo = lo;
    pl.w_hi = (const unsigned short*)d->W_split;
    pl.w_lo = pl.w_hi + (long)d->N * d->ldw;
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)) != hipSuccess) {
        set_error("hipGetSymbolAddress(g_zero_page) failed");
        return SKIMI_ERR_HIP;
    }
    pl.zero = (const unsigned short*)zp;
    a.lda = Cw;
    a.dbg = 0;
    if (d->N > 128) return launch_x3<2>(a, pl, st);
    return launch_x3<1>(a, pl, st);
}

}  // namespace skimi

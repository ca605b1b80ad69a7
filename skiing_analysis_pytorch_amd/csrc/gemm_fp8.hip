// MXFP8 contraction for the aggregator's Linear layers (BASELINE config 5: "VGGT with fp8 weights, CDNA4 fp8
// MFMA"):  out[m][n] = sum_k A[m][k] W[n][k]  on  v_mfma_scale_f32_32x32x64_f8f6f4.
//
// Number format: OCP microscaling FP8 -- e4m3 elements (the OCP `e4m3fn` of gfx950, not MI300's fnuz) with one
// power-of-two scale (E8M0 byte) per 32 consecutive K elements of a row, for BOTH operands.  That is the format
// the CDNA4 matrix pipe consumes natively: the scaled MFMA takes the two E8M0 bytes of a lane's (row, 32-K block)
// next to the operands and applies them inside the instruction, so block scaling costs no VALU work, and the
// block granularity keeps the quantisation error local (a per-output-channel scale, SURVEY §7 step 9's first
// idea, would let one outlier of a 4096-long row set the step for all of it).
//   payload [rows][Kp] e4m3 bytes, Kp = K rounded up to 128 (the tail reads as zero)
//   scales  [rows][Kp / 32] E8M0 bytes; value = 2^(byte - 127); chosen as the smallest power of two with
//           amax / scale <= 448 (no element clips)
// quant_mx_kernel produces that pair from bf16 / fp32 rows (weights once at finalize, activations per call).
//
// gemm_fp8_kernel: 256 x 256 (or, for small shapes, 128 x 128) output tile, 4 waves as 2 x 2, K-tile 128 bytes,
// both operand tiles double-buffered in LDS by LDS-DMA (global_load_lds_dwordx4, bank swizzle on the source
// chunk as in the other kernels).  The product is computed swapped (W as the MFMA's A
// operand): a lane then owns 4 consecutive output columns of one token row, i.e. 16-byte fp32 / 8-byte bf16
// stores.  The E8M0 dword of a row's K-tile is read straight from global memory (L2) one tile ahead; the byte
// for k-step ks and lane half lh (= scale block 2 ks + lh of the tile) is picked by a per-lane shift (8 lh) plus
// the instruction's op_sel (0 / 2).
// Epilogue (compile-time variants): bias; bias + erf-GELU; bias, LayerScale, residual (fp32, in place).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace skimi {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// ---------------------------------------------------------------------------------------------------------
// quantisation: one thread per 32-element block
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void quant_mx_kernel(const T* __restrict__ x, long ldx, long rows, int K, int Kp,
                                                       unsigned char* __restrict__ q, unsigned char* __restrict__ sc) {
    const int nb = Kp >> 5;
    const long total = rows * nb;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / nb;
        const int k0 = (int)(i - r * nb) * 32;
        float v[32];
        const T* xr = x + r * ldx + k0;
        if (k0 + 32 <= K) {
            if (sizeof(T) == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8 t = *reinterpret_cast<const bf16x8*>(xr + 8 * j);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[8 * j + e] = bf2f((unsigned short)t[e]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 t = *reinterpret_cast<const float4*>(xr + 4 * j);
                    v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float f = 0.f;
                if (k0 + j < K) f = sizeof(T) == 2 ? bf2f(*reinterpret_cast<const unsigned short*>(xr + j)) : (float)*reinterpret_cast<const float*>(xr + j);
                v[j] = f;
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[j]));
        const unsigned sb = mx_scale_byte(amax);
        const float inv = mx_inv_scale(sb);   // 2^-(sb - 127), exact
        i32x8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = mx_pack4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3], inv);
        unsigned char* qp = q + r * Kp + k0;
        *reinterpret_cast<i32x4*>(qp) = i32x4{out[0], out[1], out[2], out[3]};
        *reinterpret_cast<i32x4*>(qp + 16) = i32x4{out[4], out[5], out[6], out[7]};
        sc[r * nb + (k0 >> 5)] = (unsigned char)sb;
    }
}

int quant_mx_launch(const void* x, int dtype, long ldx, long rows, int K, void* q, void* scales, hipStream_t st) {
    SKIMI_CHECK_ARG(x && q && scales && rows > 0 && K > 0, "quant_mx: bad arguments");
    SKIMI_CHECK_ARG(dtype == SKIMI_BF16 || dtype == SKIMI_F32, "quant_mx: input must be bf16 or fp32");
    SKIMI_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "quant_mx: rows must be 16-byte aligned");
    const int Kp = (int)align_up((size_t)K, 128);
    const long total = rows * (Kp / 32);
    const unsigned blocks = (unsigned)std::max<long>(1, std::min<long>(cdiv(total, 256), 1 << 16));
    if (dtype == SKIMI_BF16)
        hipLaunchKernelGGL(quant_mx_kernel<unsigned short>, dim3(blocks), dim3(256), 0, st, (const unsigned short*)x, ldx, rows, K, Kp,
                           (unsigned char*)q, (unsigned char*)scales);
    else
        hipLaunchKernelGGL(quant_mx_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, ldx, rows, K, Kp,
                           (unsigned char*)q, (unsigned char*)scales);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// contraction
// ---------------------------------------------------------------------------------------------------------
struct Fp8Args {
    const unsigned char* A;    // [M][Kp]
    const unsigned char* As;   // [M][Kp / 32]
    const unsigned char* W;    // [N][Kp]
    const unsigned char* Ws;   // [N][Kp / 32]
    int M, N, Kp;
    const float* bias;         // [N] or null
    const float* gamma;        // [N] or null (EPI 2)
    const float* resid;        // fp32 [M][ldr] (EPI 2)
    long ldr;
    void* out;
    int out_dtype;             // SKIMI_F32 / SKIMI_BF16
    long ldo;
    int ntm, ntn;
};

// EPI 0: bias; 1: bias + GELU; 2: bias, * gamma, + resid.  TW: 32 x 32 MFMA tiles per wave and side: 2 -> 128 x 128
// workgroup tile (64 KiB of LDS, 2 workgroups per CU), 4 -> 256 x 256 (128 KiB, one per CU, 256 accumulator registers
// per lane): per K-tile a wave then issues 32 scaled MFMAs (2048 cycles) against 64 KiB of LDS-DMA per CU, i.e. the
// L2 -> LDS stream (measured ~42 B/clk per CU, tools/dma_depth.hip) stays under the matrix pipe.
template <int EPI, int TW>
__global__ __launch_bounds__(256, TW == 2 ? 2 : 1) void gemm_fp8_kernel(const Fp8Args p) {
    constexpr int BK = 128, BT = 64 * TW, TILE = BT * BK;          // rows and bytes of one operand tile
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [buf][W | A], 4 * TILE bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wn = wave & 1, wm = wave >> 1;
    // tile order: XCD-contiguous ids, column-of-tiles-major inside (the 8 M-tiles next to each other share a W panel)
    int id;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
        const int q = nblk >> 3, r = nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GM = 8;
    const int per_group = GM * p.ntn;
    const int group = id / per_group, within = id - group * per_group;
    const int rows_g = min(GM, p.ntm - group * GM);
    const int tn = within / rows_g, tm = group * GM + within - tn * rows_g;
    const int m0 = tm * BT, n0 = tn * BT;
    const int nkt = p.Kp / BK, nsb = p.Kp >> 5;
    constexpr int WR = 32 * TW;   // rows of a wave's MFMA tiles (2 waves per side)
    constexpr int SR = 16 * TW;   // rows a wave stages of each operand tile (4 waves per tile)

    // staging: a wave-instruction lands 8 rows x 128 B; wave w stages rows SR w .. SR w + SR - 1 of both tiles.
    // (wave-uniform 64-bit base) + (per-lane 32-bit offset): the per-lane part is the clamped row and the swizzled chunk
    const int srow = lane >> 3, sch = lane & 7;
    unsigned woff[SR / 8], aoff[SR / 8];
#pragma unroll
    for (int j = 0; j < SR / 8; ++j) {
        const int row = SR * wave + 8 * j + srow;
        const int c = sch ^ ((row >> 1) & 7);
        woff[j] = (unsigned)((long)(min(n0 + row, p.N - 1) - n0) * p.Kp + c * 16);
        aoff[j] = (unsigned)((long)(min(m0 + row, p.M - 1) - m0) * p.Kp + c * 16);
    }
    const unsigned char* wbase = p.W + (long)n0 * p.Kp;
    const unsigned char* abase = p.A + (long)m0 * p.Kp;
    auto issue = [&](int buf, int kt) {
        char* wb = smem + buf * 2 * TILE;
        char* ab = wb + TILE;
        const unsigned char* wk = wbase + (long)kt * BK;
        const unsigned char* ak = abase + (long)kt * BK;
#pragma unroll
        for (int j = 0; j < SR / 8; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_void*)(wk + woff[j]), (lds_void*)(wb + (SR * wave + 8 * j) * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(ak + aoff[j]), (lds_void*)(ab + (SR * wave + 8 * j) * 128), 16, 0, 0);
        }
    };
    // E8M0 dwords of this lane's rows: [tile i of W, tile j of A], one dword per K-tile
    const unsigned* wsc[TW];
    const unsigned* asc[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        wsc[i] = reinterpret_cast<const unsigned*>(p.Ws + (long)min(n0 + wn * WR + i * 32 + l31, p.N - 1) * nsb);
        asc[i] = reinterpret_cast<const unsigned*>(p.As + (long)min(m0 + wm * WR + i * 32 + l31, p.M - 1) * nsb);
    }
    f32x16 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned swn[TW], san[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        swn[i] = wsc[i][0];
        san[i] = asc[i][0];
    }
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        unsigned sw[TW], sa[TW];
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            sw[i] = swn[i] >> (8 * lh);
            sa[i] = san[i] >> (8 * lh);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of tile kt (and the scales) has landed
        __syncthreads();                      // ... everybody's; and everybody is done reading buffer cur ^ 1
        if (kt + 1 < nkt) {
            issue(cur ^ 1, kt + 1);
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                swn[i] = wsc[i][kt + 1];
                san[i] = asc[i][kt + 1];
            }
        }
        const char* wb = smem + cur * 2 * TILE;
        const char* ab = wb + TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x8 wf[TW], af[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                const int wr = wn * WR + i * 32 + l31, ar = wm * WR + i * 32 + l31;
                // operand image of v_mfma_scale_f32_32x32x64_f8f6f4 (probed with one-hot rows, tools/dbg_fp8.py): the 64 K
                // of a step are two 32-element scale blocks; lane (row, lh) holds elements 16 lh .. 16 lh + 15 of block 0
                // in its first 16 bytes and of block 1 in its second 16 bytes; block 0's E8M0 scale is taken from the
                // lh = 0 lane of the row, block 1's from the lh = 1 lane
                const int c0 = 4 * ks + lh, c1 = c0 + 2;
                const i32x4 w0 = *reinterpret_cast<const i32x4*>(wb + wr * 128 + ((c0 ^ ((wr >> 1) & 7)) << 4));
                const i32x4 w1 = *reinterpret_cast<const i32x4*>(wb + wr * 128 + ((c1 ^ ((wr >> 1) & 7)) << 4));
                const i32x4 a0 = *reinterpret_cast<const i32x4*>(ab + ar * 128 + ((c0 ^ ((ar >> 1) & 7)) << 4));
                const i32x4 a1 = *reinterpret_cast<const i32x4*>(ab + ar * 128 + ((c1 ^ ((ar >> 1) & 7)) << 4));
                wf[i] = i32x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                af[i] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            }
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    if (ks == 0)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], af[j], acc[i][j], 0, 0, 0, (int)sw[i], 0, (int)sa[j]);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], af[j], acc[i][j], 0, 0, 2, (int)sw[i], 2, (int)sa[j]);
                }
        }
    }

    // D[n][m]: lane (m = l31, lh), register r -> n = (r & 3) + 8 (r >> 2) + 4 lh: 4 consecutive columns per group
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int m = m0 + wm * WR + j * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WR + i * 32 + 8 * g + 4 * lh;
                if (n >= p.N) continue;      // N % 4 == 0 (checked by the launcher)
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                if (EPI == 2) {
                    const float4 gm = *reinterpret_cast<const float4*>(p.gamma + n);
                    const float4 rs = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + n);
                    v[0] = v[0] * gm.x + rs.x; v[1] = v[1] * gm.y + rs.y; v[2] = v[2] * gm.z + rs.z; v[3] = v[3] * gm.w + rs.w;
                }
                if (p.out_dtype == SKIMI_F32) {
                    *reinterpret_cast<float4*>((float*)p.out + (long)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (short)f2bf(v[e]);
                    *reinterpret_cast<bf16x4*>((unsigned short*)p.out + (long)m * p.ldo + n) = o;
                }
            }
    }
}

// large launches go to the single-stream 256-row loop (gemm256w4_fp8_kernel); SKIMI_FP8_W4=0 keeps them here (A/B timing)
static bool fp8_w4_shape(int M, int N) {
    static const bool dyn = getenv("SKIMI_ENV_DYNAMIC") && atoi(getenv("SKIMI_ENV_DYNAMIC"));
    static int use_w4 = -1;
    if (use_w4 < 0 || dyn) use_w4 = getenv("SKIMI_FP8_W4") ? atoi(getenv("SKIMI_FP8_W4")) : 1;
    return use_w4 && M >= 2048 && N >= 512 && N % 4 == 0 && cdiv(M, 256) * cdiv(N, 256) >= 160;
}
// whether a bias + GELU launch of this shape can write its result as MXFP8 (out_dtype SKIMI_FP8MX)
bool gemm_fp8_mx_output_ok(int M, int N) { return N % 128 == 0 && fp8_w4_shape(M, N); }

int gemm_fp8_launch(const void* A, const void* As, const void* W, const void* Ws, int M, int N, int K, const float* bias, int act,
                    const float* gamma, const float* resid, long ldr, void* out, int out_dtype, long ldo, hipStream_t st,
                    void* out_scales) {
    SKIMI_CHECK_ARG(A && As && W && Ws && out && M > 0 && N > 0 && K > 0, "gemm_fp8: bad arguments");
    SKIMI_CHECK_ARG(N % 4 == 0 && ldo % 4 == 0 && (ldr % 4 == 0), "gemm_fp8: N and the row strides must be multiples of 4");
    SKIMI_CHECK_ARG(out_dtype == SKIMI_F32 || out_dtype == SKIMI_BF16 || out_dtype == SKIMI_FP8MX, "gemm_fp8: output must be fp32, bf16 or MXFP8");
    SKIMI_CHECK_ARG(out_dtype != SKIMI_FP8MX || (out_scales && act == SKIMI_ACT_GELU && !gamma && !resid && bias && N % 128 == 0 && ldo == N),
                    "gemm_fp8: MXFP8 output needs out_scales, the bias + GELU epilogue, N %% 128 == 0 and ldo == N");
    SKIMI_CHECK_ARG((((uintptr_t)As | (uintptr_t)Ws) & 3) == 0, "gemm_fp8: the scale arrays are read as dwords (4-byte aligned; K padded to 128 keeps every row so)");
    SKIMI_CHECK_ARG(act == SKIMI_ACT_NONE || act == SKIMI_ACT_GELU, "gemm_fp8: activation must be none or GELU");
    SKIMI_CHECK_ARG(!(gamma != nullptr) || resid != nullptr, "gemm_fp8: LayerScale needs the residual");
    SKIMI_CHECK_ARG(!(act == SKIMI_ACT_GELU && gamma), "gemm_fp8: GELU and LayerScale epilogues are separate");
    Fp8Args p;
    p.A = (const unsigned char*)A; p.As = (const unsigned char*)As; p.W = (const unsigned char*)W; p.Ws = (const unsigned char*)Ws;
    p.M = M; p.N = N; p.Kp = (int)align_up((size_t)K, 128);
    p.bias = bias; p.gamma = gamma; p.resid = resid; p.ldr = ldr; p.out = out; p.out_dtype = out_dtype; p.ldo = ldo;
    // Shapes that fill the chip with 256-row tiles and end in one of the three block epilogues (bias -> bf16; bias,
    // GELU -> bf16; bias, LayerScale, fp32 residual -> fp32) run on the single-stream loop of gemm256.hip
    // (gemm256w4_fp8_kernel).  SKIMI_FP8_W4=0 keeps them on this file's kernel (A/B timing).
    {
        const bool epi_ok = ((out_dtype == SKIMI_BF16 || out_dtype == SKIMI_FP8MX) && !gamma && !resid && bias) ||
                            (out_dtype == SKIMI_F32 && gamma && resid && bias);
        if (epi_ok && fp8_w4_shape(M, N) && (((uintptr_t)A | (uintptr_t)W | (uintptr_t)As | (uintptr_t)Ws) & 15) == 0) {
            GemmArgs g;
            memset(&g, 0, sizeof g);
            g.M = M; g.N = N; g.K = p.Kp;
            g.A = A; g.W = W; g.lda = p.Kp; g.ldw = p.Kp;
            g.a_scales = (const unsigned char*)As; g.w_scales = (const unsigned char*)Ws; g.lsa = p.Kp / 32; g.lsw = p.Kp / 32;
            g.bias = bias; g.gamma = gamma; g.resid = resid; g.resid_dtype = SKIMI_F32; g.ldr = ldr;
            g.act = act; g.out = out; g.out_dtype = out_dtype; g.ldo = ldo; g.vec4 = 1;
            g.out_scales = (unsigned char*)out_scales;
            const bool prof2 = prof_armed(PROF_GEMM, N);
            if (prof2) prof_before(st);
            const int rc = gemm256_fp8_launch(g, st);
            if (prof2) prof_after(st, 2.0 * M * (double)N * K, (double)M * K + (double)N * K + (double)M * N * (out_dtype == SKIMI_F32 ? 4 : 2));
            return rc;
        }
    }
    SKIMI_CHECK_ARG(out_dtype != SKIMI_FP8MX, "gemm_fp8: MXFP8 output is served by the single-stream 256-row loop only (M >= 2048, "
                    "N >= 512, >= 160 tiles of 256 x 256)");
    // 256 x 256 tiles where the K loop is long enough to pay for the larger epilogue and they fill most of the chip
    // (measured at M = 43968, tools/mb_fp8.py: K = 4096 -> 1207 vs 1071 TFLOP/s, K = 1024 -> 736-803 vs 828-865),
    // else 128 x 128 (two workgroups per CU).  SKIMI_FP8_TILE=128|256 forces a choice (A/B timing).
    static const int force_tile = getenv("SKIMI_FP8_TILE") ? atoi(getenv("SKIMI_FP8_TILE")) : 0;
    const bool big = force_tile ? force_tile == 256 : (K >= 2048 && cdiv(M, 256) * cdiv(N, 256) >= 200);
    const int bt = big ? 256 : 128;
    p.ntm = (int)cdiv(M, bt); p.ntn = (int)cdiv(N, bt);
    const long nblk = (long)p.ntm * p.ntn;
    SKIMI_CHECK_ARG(nblk < (1l << 31), "gemm_fp8: grid too large");
    SKIMI_CHECK_ARG((long)bt * p.Kp < (1l << 32), "gemm_fp8: K too large for 32-bit staging offsets");
    const int epi = gamma ? 2 : act == SKIMI_ACT_GELU ? 1 : 0;
    const bool prof = prof_armed(PROF_GEMM, N);
    if (prof) prof_before(st);
#define SKIMI_FP8_LAUNCH(E, T)                                                                                          \
    do {                                                                                                                \
        constexpr int lds = 4 * 64 * T * 128;                                                                           \
        if (lds > 64 * 1024) SKIMI_LDS_OPT_IN((gemm_fp8_kernel<E, T>), lds, "gemm_fp8");                                \
        hipLaunchKernelGGL((gemm_fp8_kernel<E, T>), dim3((unsigned)nblk), dim3(256), lds, st, p);                       \
    } while (0)
    if (big) {
        if (epi == 2) SKIMI_FP8_LAUNCH(2, 4); else if (epi == 1) SKIMI_FP8_LAUNCH(1, 4); else SKIMI_FP8_LAUNCH(0, 4);
    } else {
        if (epi == 2) SKIMI_FP8_LAUNCH(2, 2); else if (epi == 1) SKIMI_FP8_LAUNCH(1, 2); else SKIMI_FP8_LAUNCH(0, 2);
    }
#undef SKIMI_FP8_LAUNCH
    if (prof) prof_after(st, 2.0 * M * (double)N * K, (double)M * K + (double)N * K + (double)M * N * (out_dtype == SKIMI_F32 ? 4 : 2));
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

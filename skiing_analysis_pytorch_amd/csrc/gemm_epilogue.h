// GemmArgs + the shared fused epilogue (bias, activation, LayerScale, residuals, row remaps,
// fp32/bf16 stores) of the MFMA contraction kernels in gemm.hip and gemm256.hip.
#pragma once
#include "common.h"

namespace skimi {

struct GemmArgs {
    int M, N, K;
    const void* A;
    const void* W;
    long lda, ldw;
    int a_mode;
    int cN, cH, cW, cC, KH, KW, stride, pad, dil, OH, OW;
    const float* bias;
    const float* gamma;
    const void* resid;     // f32, bf16 or fp16 (resid_dtype)
    int resid_dtype;
    long ldr;
    int resid_rpb;
    long resid_bs;
    long resid_off;
    const void* resid2;    // second residual (same dtype), plain row m
    long ldr2;
    int out_rpb;           // output row remap (store_mode 0), like the residual's
    long out_bs;
    long out_off;
    int act;
    int post_act;          // activation applied after the residual adds
    void* out;
    void* out2;
    int out_dtype;
    long ldo, ldo2;
    int store_mode, ps_s, ps_C;
    // split-K
    int splitk;
    int k_per_split;   // multiple of BK
    float* partial;    // [M, N] fp32, zeroed ([splitk][M, N] when splitk_ordered)
    int splitk_ordered; // K splits write their own slab plane, the epilogue adds them in order (deterministic)
    int ntm, ntn;
    int vec4;          // epilogue may use 16-B accesses (N, ld*, pointers all 4-element aligned)
    int f16;           // 16-bit operands are fp16 and the MFMA is the f16 one (SKIMI_PREC_F16)
    int dbg;           // diagnostics only (gemm256 ablation: bit0 skip staging, bit1 skip MFMA phase)
    unsigned short* out_rec;   // or NULL: result rows also as bf16x3 records [M][N/32][hi 32 | lo 32]
    long rec_row;              // elements per record row = N/32 * 64
    // MXFP8 launches (gemm256w4_fp8_kernel): A / W are e4m3 payloads (lda / ldw in BYTES = Kp, K = Kp), scales E8M0
    const unsigned char* a_scales;
    const unsigned char* w_scales;
    long lsa, lsw;             // bytes per scale row = Kp / 32
    unsigned char* out_scales; // out_dtype SKIMI_FP8MX: E8M0 scales [M][N / 32] of the result (payload bytes in out, ldo = N)
};

// columns n..n+3 of row m into the records (8-byte stores; 8 lanes fill one 128-byte record)
__device__ __forceinline__ void store_rec4(const GemmArgs& p, long m, int n, float v0, float v1, float v2, float v3) {
    const float f[4] = {v0, v1, v2, v3};
    bf16x4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned short hb = f2bf(f[k]);
        h[k] = (short)hb;
        l[k] = (short)f2bf(f[k] - bf2f(hb));
    }
    unsigned short* q = p.out_rec + m * p.rec_row + (n >> 5) * 64 + (n & 31);
    *reinterpret_cast<bf16x4*>(q) = h;
    *reinterpret_cast<bf16x4*>(q + 32) = l;
}

// erf-GELU with erfc by Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7 absolute, i.e. fp32 rounding
// level on O(1) activations): branch-free, ~16 VALU ops.  libm's erff/expf inline to ~1 KB of
// branchy code per call site, which pushed the unrolled GEMM epilogues past the 64-KiB
// instruction cache (gemm256 bias+GELU epilogue: 110 KB of code, ~3x slower stores).
__device__ __forceinline__ float gelu_erf(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
    float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    q = __builtin_fmaf(t, q, 1.421413741f);
    q = __builtin_fmaf(t, q, -0.284496736f);
    q = __builtin_fmaf(t, q, 0.254829592f);
    q = q * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z) * 0.5f;   // 0.5 * erfc(|z|)
    return v >= 0.f ? __builtin_fmaf(-v, q, v) : v * q;
}
// The same formula on two values at once, written so that hipcc emits packed fp32 math
// (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth per issue slot): the GELU of a 256x256 tile is
// 256 values per lane on four waves, i.e. VALU time that nothing overlaps in a one-workgroup-
// per-CU GEMM.  With a = |v|:  t = 1 / (1 + 0.2316419 a),  q = (poly(t) / 2) t exp2(-(0.8493 a)^2)
// = erfc(a / sqrt 2) / 2,  gelu = max(v, 0) - a q.   10 packed ops + 2 rcp + 2 exp2 per pair.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t v) {
    const f32x2_t a = {fabsf(v.x), fabsf(v.y)};
    const f32x2_t d = a * 0.23164190165f + 1.f;                 // 0.3275911 / sqrt 2
    const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2_t w = a * 0.84932180028f;                       // sqrt(log2(e) / 2)
    const f32x2_t ww = w * w;
    const f32x2_t e = {__builtin_amdgcn_exp2f(-ww.x), __builtin_amdgcn_exp2f(-ww.y)};
    f32x2_t q = t * 0.5307027145f + (-0.7265760135f);           // A&S coefficients, halved
    q = t * q + 0.7107068705f;
    q = t * q + (-0.142248368f);
    q = t * q + 0.127414796f;
    q = q * t * e;
    const f32x2_t relu = {fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)};
    return relu - a * q;
}
// 1 / (1 + e^-v) on the hardware exp2 / rcp units (1 ulp each)
__device__ __forceinline__ float sigmoid_hw(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case SKIMI_ACT_RELU: return fmaxf(v, 0.f);
        case SKIMI_ACT_GELU: return gelu_erf(v);
        case SKIMI_ACT_SILU: return v * sigmoid_hw(v);
        case SKIMI_ACT_SIGMOID: return sigmoid_hw(v);
        default: return v;
    }
}

__device__ __forceinline__ float load_res(const void* r, int dt, long i) {
    return dt == SKIMI_F32 ? ((const float*)r)[i] : x16tof(((const unsigned short*)r)[i], dt == SKIMI_F16);
}
__device__ __forceinline__ float4 load_res4(const void* r, int dt, long i) {
    if (dt == SKIMI_F32) return *reinterpret_cast<const float4*>((const float*)r + i);
    const bf16x4 v = *reinterpret_cast<const bf16x4*>((const unsigned short*)r + i);
    const bool h = dt == SKIMI_F16;
    return make_float4(x16tof((unsigned short)v[0], h), x16tof((unsigned short)v[1], h), x16tof((unsigned short)v[2], h),
                       x16tof((unsigned short)v[3], h));
}

// everything that depends only on the output row m
struct RowMap {
    long out_off;   // mode 0: row*ldo; mode 1: top-left output pixel index of this input pixel
    long out2_off;  // same for out2
    long res_off;
    long res2_off;
    long row;       // m (index of the record row when the launch also writes records)
};

__device__ __forceinline__ RowMap row_map(const GemmArgs& p, int m) {
    RowMap r;
    r.row = m;
    if (p.store_mode == 0) {
        long mo = m;
        if (p.out_rpb > 0) {
            int b = m / p.out_rpb;
            mo = (long)b * p.out_bs + (m - b * p.out_rpb);
        }
        mo += p.out_off;
        r.out_off = mo * p.ldo;
        r.out2_off = mo * p.ldo2;
    } else {
        // m = (img, iy, ix) over [cN, cH, cW]
        int hw = p.cH * p.cW;
        int img = m / hw;
        int rem = m - img * hw;
        int iy = rem / p.cW;
        int ix = rem - iy * p.cW;
        long OWs = (long)p.cW * p.ps_s;
        r.out_off = (((long)img * p.cH * p.ps_s + (long)iy * p.ps_s) * OWs + (long)ix * p.ps_s);
        r.out2_off = r.out_off;
    }
    long mr = m;
    if (p.resid_rpb > 0) {
        int b = m / p.resid_rpb;
        mr = (long)b * p.resid_bs + (m - b * p.resid_rpb);
    }
    r.res_off = (mr + p.resid_off) * p.ldr;
    r.res2_off = (long)m * p.ldr2;
    return r;
}

__device__ __forceinline__ void store_one(const GemmArgs& p, const RowMap& rm, int n, float acc) {
    float v = acc;
    if (p.bias) v += p.bias[n];
    v = apply_act(v, p.act);
    if (p.gamma) v *= p.gamma[n];
    if (p.resid) v += load_res(p.resid, p.resid_dtype, rm.res_off + n);
    if (p.resid2) v += load_res(p.resid2, p.resid_dtype, rm.res2_off + n);
    v = apply_act(v, p.post_act);
    long o, o2;
    if (p.store_mode == 0) {
        o = rm.out_off + n;
        o2 = rm.out2_off + n;
    } else {
        int ab = n / p.ps_C;
        int co = n - ab * p.ps_C;
        int a = ab / p.ps_s;
        int b = ab - a * p.ps_s;
        long OWs = (long)p.cW * p.ps_s;
        long pix = rm.out_off + (long)a * OWs + b;
        o = pix * p.ldo + co;
        o2 = pix * p.ldo2 + co;
    }
    if (p.out_rec) {   // plain output rows only (checked at dispatch)
        const unsigned short hb = f2bf(v);
        unsigned short* q = p.out_rec + rm.row * p.rec_row + (n >> 5) * 64 + (n & 31);
        q[0] = hb;
        q[32] = f2bf(v - bf2f(hb));
    }
    if (!p.out) return;
    if (p.out_dtype == SKIMI_F32) {
        ((float*)p.out)[o] = v;
        if (p.out2) ((unsigned short*)p.out2)[o2] = f2bf(v);
    } else {
        ((unsigned short*)p.out)[o] = f2x16(v, p.out_dtype == SKIMI_F16);
        if (p.out2) ((float*)p.out2)[o2] = v;
    }
}

// four consecutive columns n..n+3 of one row (vectorised epilogue)
__device__ __forceinline__ void store_four(const GemmArgs& p, const RowMap& rm, int n, float4 acc) {
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], p.act);
    if (p.gamma) {
        const float4 g = *reinterpret_cast<const float4*>(p.gamma + n);
        v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
    }
    if (p.resid) {
        const float4 r = load_res4(p.resid, p.resid_dtype, rm.res_off + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    if (p.resid2) {
        const float4 r = load_res4(p.resid2, p.resid_dtype, rm.res2_off + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    if (p.post_act != SKIMI_ACT_NONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], p.post_act);
    }
    long o, o2;
    if (p.store_mode == 0) {
        o = rm.out_off + n;
        o2 = rm.out2_off + n;
    } else {
        int ab = n / p.ps_C;
        int co = n - ab * p.ps_C;
        int a = ab / p.ps_s;
        int b = ab - a * p.ps_s;
        long OWs = (long)p.cW * p.ps_s;
        long pix = rm.out_off + (long)a * OWs + b;
        o = pix * p.ldo + co;
        o2 = pix * p.ldo2 + co;
    }
    if (p.out_rec) store_rec4(p, rm.row, n, v[0], v[1], v[2], v[3]);
    if (!p.out) return;
    bf16x4 hb;   // the 16-bit copy: bf16, or fp16 when that is the output type (out2 of an fp32 result is always bf16)
    const bool h16 = p.out_dtype == SKIMI_F16;
#pragma unroll
    for (int k = 0; k < 4; ++k) hb[k] = (short)f2x16(v[k], h16);
    const float4 fv = make_float4(v[0], v[1], v[2], v[3]);
    if (p.out_dtype == SKIMI_F32) {
        *reinterpret_cast<float4*>((float*)p.out + o) = fv;
        if (p.out2) *reinterpret_cast<bf16x4*>((unsigned short*)p.out2 + o2) = hb;
    } else {
        *reinterpret_cast<bf16x4*>((unsigned short*)p.out + o) = hb;
        if (p.out2) *reinterpret_cast<float4*>((float*)p.out2 + o2) = fv;
    }
}

bool gemm256_eligible(const skimi_gemm_desc* d);
bool gemm_x3dma_eligible(const skimi_gemm_desc* d);
int gemm_x3dma_launch(GemmArgs& a, const skimi_gemm_desc* d, hipStream_t st);
size_t gemm_x3dma_scratch_bytes(const skimi_gemm_desc* d);
int split_planes_launch(const float* x, long ld, long rows, int C, void* hi, void* lo, hipStream_t st);
// fp32 [rows, C] -> records [rows][ceil(C/32)][hi 32 | lo 32] bf16 (operand form of the LDS-DMA bf16x3 kernel)
int split_records_launch(const float* x, long ld, long rows, int C, void* rec, hipStream_t st, void* zpage = nullptr);
int gemm256_launch(GemmArgs& a, hipStream_t st);
// conv_win.hip: 3 x 3 / stride 1 / pad 1 -> 128 channels on single-term 16-bit operands, halo window in LDS
bool conv_win_eligible(const skimi_gemm_desc* d);
int conv_win_launch(GemmArgs& a, hipStream_t st);
int gemm256_fp8_launch(GemmArgs& a, hipStream_t st);   // MXFP8 operands on the single-stream 256 x 256 loop; a.K = Kp (bytes per row)

}  // namespace skimi

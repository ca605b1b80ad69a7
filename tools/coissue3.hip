// Mini attention step without memory traffic: per iteration one 32x32 S tile by 4 chained MFMAs
// (K=64), the softmax VALU work on the PREVIOUS iteration's S tile (max, fma, exp2, sum, bf16
// pack), and 4 PV MFMAs with the previous P.  Do the VALU stream and the MFMA stream overlap?
//   MODE 0 both streams, 1 MFMA only, 2 VALU only.   2 workgroups of 4 waves per CU = 2 waves/SIMD.
// Measured (MI355X, ns per iteration, 2 waves/SIMD): MFMA-only 238, VALU-only 233, both 497 (= the sum);
// sched_group_barrier / iglp_opt pipelines 470-490; with the softmax input and P cut loose from the
// MFMAs (no cross-stream dependencies) 358; each op kind alone co-issues with the MFMA (coissue.hip).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/coissue3.hip -o tools/_coissue3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 b = (__bf16)x;   // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned short, b);
}

// AG: accumulators in AGPRs (inline-asm MFMA with "a" constraints) instead of arch VGPRs
// DEP: 0 full dependencies; 1 softmax input does not come from the MFMAs; 2 also P not fed to the PV MFMAs
template <int MODE, int GRP, int AG = 0, int DEP = 0>   // GRP > 0: sched_group_barrier pipeline of 1 MFMA + GRP VALU
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    bf16x8 kf[4], qf[4], vf[4], pf[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            kf[i][j] = (short)(0x3C00 + threadIdx.x + i + j);
            qf[i][j] = (short)(0x3C00 + threadIdx.x * 3 + i + j);
            vf[i][j] = (short)(0x3C00 + threadIdx.x * 5 + i + j);
        }
    for (int j = 0; j < 8; ++j) pf[0][j] = pf[1][j] = (short)0x3C00;
    f32x16 s, sp, o[2];
    for (int r = 0; r < 16; ++r) { s[r] = sp[r] = seed * r; o[0][r] = o[1][r] = 0.f; }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;
    const float c2 = 0.18f;
    if (MODE == 4 || MODE == 5 || MODE == 8) {   // MODE 8 = MODE 4 + the hand-off: the asm VALU ops consume copies of S made at the iteration end
        // the MFMA stream of MODE 3 (S chain with 4 different A/B pairs, then PV) with the VALU work
        // replaced by 8 independent asm ops per MFMA (MODE 4) or none (MODE 5): which side breaks co-issue?
        float w[8];
        for (int j = 0; j < 8; ++j) w[j] = seed + j;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < 4) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[c], c == 0 ? zero : s, 0, 0, 0);
                else o[c & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[c - 4], pf[(c - 4) >> 1], o[c & 1], 0, 0, 0);
                asm volatile("" : "+v"(s), "+v"(o[0]), "+v"(o[1]));
                if (MODE == 4 || MODE == 8) {
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        float& x = w[n];
                        switch (n) {
                            case 0: case 1: asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(seed)); break;
                            case 2: case 3: asm volatile("v_exp_f32 %0, %0" : "+v"(x)); break;
                            case 4: case 5: asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(seed)); break;
                            case 6: asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(seed)); break;
                            default: asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(seed)); break;
                        }
                    }
                }
            }
            if (MODE == 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = s[j] + s[j + 8];   // VALU reads of MFMA-written registers
            }
        }
        for (int j = 0; j < 8; ++j) l += w[j];
    } else
    if (MODE == 3 || MODE == 6 || MODE == 7) {   // MODE 7: S goes to the softmax registers through LDS (ds_write / ds_read) instead of v_mov   // MODE 6: the same with S and O accumulators in AGPRs (asm MFMA, "a" constraints)
        // hand-chunked: source order = issue order at chunk granularity (sched_barrier(0) between
        // chunks): chunk c = MFMA c of the iteration + its share of the previous tile's softmax
        for (int it = 0; it < iters; ++it) {
            bf16x8 pn[2] = {pf[0], pf[1]};
            float mloc, nmb, l0 = 0.f, l1 = 0.f;
#define MF(c)                                                                                               \
    if (MODE == 6) {                                                                                        \
        if ((c) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(s) : "v"(kf[c]), "v"(qf[c]));                \
        else if ((c) < 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(s) : "v"(kf[c]), "v"(qf[c]));           \
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[(c) & 1]) : "v"(vf[(c) - 4]), "v"(pf[((c) - 4) >> 1])); \
    } else if ((c) < 4) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[c], qf[c], (c) == 0 ? zero : s, 0, 0, 0);   \
    else o[(c) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[(c) - 4], pf[((c) - 4) >> 1], o[(c) & 1], 0, 0, 0);
#define EL(r)                                                                                               \
    { float v = __builtin_amdgcn_exp2f(__builtin_fmaf(sp[r], c2, nmb)); if ((r) & 1) l1 += v; else l0 += v; sp[r] = v; }
// every live value passes through an empty volatile asm: nothing can be moved across it
#define FENCE                                                                                               \
    if (MODE == 6) asm volatile("" : "+v"(sp), "+a"(s), "+a"(o[0]), "+a"(o[1]), "+v"(pn[0]), "+v"(pn[1]), "+v"(l0), "+v"(l1), "+v"(nmb)); \
    else asm volatile("" : "+v"(sp), "+v"(s), "+v"(o[0]), "+v"(o[1]), "+v"(pn[0]), "+v"(pn[1]), "+v"(l0), "+v"(l1), "+v"(nmb))
#define CV(ks, j0)                                                                                          \
    { pn[ks][j0] = (short)f2bf(sp[8 * (ks) + (j0)]); pn[ks][(j0) + 1] = (short)f2bf(sp[8 * (ks) + (j0) + 1]); }
            MF(0)
            mloc = sp[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = __builtin_fmaxf(mloc, sp[r]);
            m = __builtin_fmaxf(m, mloc);
            nmb = -(m * c2);
            FENCE;
            MF(1) EL(0) EL(1) CV(0, 0)
            FENCE;
            MF(2) EL(2) EL(3) CV(0, 2) EL(4)
            FENCE;
            MF(3) EL(5) CV(0, 4) EL(6) EL(7) CV(0, 6)
            FENCE;
            MF(4) EL(8) EL(9) CV(1, 0)
            FENCE;
            MF(5) EL(10) EL(11) CV(1, 2) EL(12)
            FENCE;
            MF(6) EL(13) CV(1, 4) EL(14)
            FENCE;
            MF(7) EL(15) CV(1, 6)
            l += l0 + l1;
            FENCE;
            pf[0] = pn[0]; pf[1] = pn[1];
            if (MODE == 7) {
                __shared__ __attribute__((aligned(16))) float xch[256 * 16];
                float4* slot = reinterpret_cast<float4*>(xch) + threadIdx.x;   // [4][256] float4, conflict-free
#pragma unroll
                for (int g = 0; g < 4; ++g) slot[g * 256] = make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = slot[g * 256];
                    sp[4 * g] = t.x; sp[4 * g + 1] = t.y; sp[4 * g + 2] = t.z; sp[4 * g + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sp[r] = s[r];
            }
#undef MF
#undef EL
#undef CV
#undef FENCE
        }
    } else
    for (int it = 0; it < iters; ++it) {
        // MFMA stream A: S = K Q^T
        if (MODE != 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (AG) {
                    if (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(s) : "v"(kf[ks]), "v"(qf[ks]));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(s) : "v"(kf[ks]), "v"(qf[ks]));
                } else {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks == 0 ? zero : s, 0, 0, 0);
                }
            }
        }
        // VALU stream: softmax of the previous S
        bf16x8 pn[2] = {pf[0], pf[1]};
        if (MODE != 1) {
            float mloc = sp[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = __builtin_fmaxf(mloc, sp[r]);
            const float mnew = __builtin_fmaxf(m, mloc);
            m = mnew;
            const float nmb = -(mnew * c2);
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sp[r], c2, nmb));
                float v1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sp[r + 1], c2, nmb));
                l0 += v0; l1 += v1;
                sp[r] = v0; sp[r + 1] = v1;
            }
            l += l0 + l1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) pn[ks][j] = (short)f2bf(sp[8 * ks + j]);
        }
        // MFMA stream B: O += V^T P^T with the previous P
        if (MODE != 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[dt]) : "v"(vf[2 * ks + dt]), "v"(pf[ks]));
                    else o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2 * ks + dt], pf[ks], o[dt], 0, 0, 0);
                }
        }
        if (GRP < 0 && MODE == 0) __builtin_amdgcn_iglp_opt(-GRP - 1);   // GRP -1,-2,-3,-4 -> iglp_opt 0..3
        if (GRP >= 100 && MODE == 0) {
            // shaped pipelines: the max chain (8 max3) under the first two MFMAs, the 16 exp chains under the other six
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define MFM SG(0x008, 1)
            if (GRP == 100) { MFM; SG(2, 5); MFM; SG(2, 5); MFM; SG(2, 11); MFM; SG(2, 11); MFM; SG(2, 11); MFM; SG(2, 11); MFM; SG(2, 11); MFM; SG(2, 11); }
            if (GRP == 101) { MFM; SG(2, 5); MFM; SG(2, 5); MFM; SG(0x400, 3); SG(2, 8); MFM; SG(0x400, 3); SG(2, 8); MFM; SG(0x400, 3); SG(2, 8);
                              MFM; SG(0x400, 3); SG(2, 8); MFM; SG(0x400, 2); SG(2, 8); MFM; SG(0x400, 2); SG(2, 8); }
            if (GRP == 102) { MFM; SG(2, 5); MFM; SG(2, 5); MFM; SG(2, 4); SG(0x400, 3); SG(2, 4); MFM; SG(2, 4); SG(0x400, 3); SG(2, 4); MFM; SG(2, 4); SG(0x400, 3); SG(2, 4);
                              MFM; SG(2, 4); SG(0x400, 3); SG(2, 4); MFM; SG(2, 4); SG(0x400, 2); SG(2, 4); MFM; SG(2, 4); SG(0x400, 2); SG(2, 4); }
            if (GRP == 103) { MFM; SG(2, 9); MFM; SG(2, 10); MFM; SG(2, 10); MFM; SG(2, 10); MFM; SG(2, 10); MFM; SG(2, 10); MFM; SG(2, 10); MFM; SG(2, 10); }
#undef SG
#undef MFM
        } else
        if (GRP > 0 && MODE == 0) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, GRP, 0);   // GRP VALU
            }
        }
        if (DEP < 2) { pf[0] = pn[0]; pf[1] = pn[1]; } else { l += (float)pn[0][0] + (float)pn[1][0]; }
        // hand S to the next iteration's softmax (MFMA-only mode keeps the dependency alive cheaply)
        if (MODE == 1 || DEP >= 1) { sp[0] += s[0] * 1e-30f; } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sp[r] = s[r];
        }
    }
    float acc = m + l;
    for (int r = 0; r < 16; ++r) acc += o[0][r] + o[1][r] + sp[r] + s[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + pf[0][0] + pf[1][0];
}

template <int MODE, int GRP, int AG = 0, int DEP = 0>
float run(float* out, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, GRP, AG, DEP><<<blocks, 256>>>(out, 100, 1.0f);
    (void)hipEventRecord(e0);
    k<MODE, GRP, AG, DEP><<<blocks, 256>>>(out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 512 * 256 * 4);
    for (int blocks : {256, 512}) {
        const float both = run<0, 0>(out, blocks), mf = run<1, 0>(out, blocks), va = run<2, 0>(out, blocks);
        printf("waves/SIMD=%d  ns per iteration (8 MFMA + softmax of 16 scores/lane): both %.1f  MFMA-only %.1f  VALU-only %.1f  sum %.1f\n",
               blocks / 256, both, mf, va, mf + va);
        printf("   sched_group_barrier 1 MFMA + N VALU:  N=6 %.1f  N=8 %.1f  N=10 %.1f  N=12 %.1f\n", run<0, 6>(out, blocks),
               run<0, 8>(out, blocks), run<0, 10>(out, blocks), run<0, 12>(out, blocks));
        // iglp_opt(1) is left out: hipcc (ROCm 7.2) runs out of memory on it for this loop
        printf("   MODE-3 MFMA stream + 8 independent asm VALU per MFMA: %.1f   that MFMA stream alone: %.1f   with S copied into the VALU registers each iteration: %.1f\n",
               run<4, 0>(out, blocks), run<5, 0>(out, blocks), run<8, 0>(out, blocks));
        printf("   strictly interleaved, S handed to the softmax through LDS: %.1f\n", run<7, 0>(out, blocks));
        printf("   strictly interleaved with the accumulators in AGPRs: %.1f\n", run<6, 0>(out, blocks));
        printf("   hand-chunked (sched_barrier between 1 MFMA + its VALU share): %.1f\n", run<3, 0>(out, blocks));
        printf("   shaped sched_group_barrier pipelines P0..P3: %.1f %.1f %.1f %.1f\n", run<0, 100>(out, blocks), run<0, 101>(out, blocks), run<0, 102>(out, blocks), run<0, 103>(out, blocks));
        printf("   iglp_opt(0), (2), (3): %.1f %.1f %.1f\n", run<0, -1>(out, blocks), run<0, -3>(out, blocks), run<0, -4>(out, blocks));
        printf("   softmax input independent of the MFMAs: both %.1f (grouped N=12: %.1f); P also not fed to PV: both %.1f (grouped %.1f)\n",
               run<0, 0, 0, 1>(out, blocks), run<0, 12, 0, 1>(out, blocks), run<0, 0, 0, 2>(out, blocks), run<0, 12, 0, 2>(out, blocks));
        printf("   accumulators in AGPRs (asm MFMA):  both %.1f  MFMA-only %.1f\n", run<0, 0, 1>(out, blocks), run<1, 0, 1>(out, blocks));
    }
    (void)hipDeviceSynchronize();
    return 0;
}

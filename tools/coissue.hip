// Does VALU work hide under v_mfma_f32_32x32x16_bf16 on gfx950?  One MFMA + NV VALU ops per
// iteration, 1 or 2 waves per SIMD; prints cycles per iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// KIND: 0 v_fma_f32, 1 v_exp_f32, 2 v_pk_fma_f32, 3 v_cvt_pk_bf16_f32, 4 v_max3_f32
template <int NV, int KIND, int NM, int NACC = 2>   // NACC 1: every MFMA accumulates into the same registers (dependent chain)
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = seed;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(threadIdx.x + j); b[j] = (short)(threadIdx.x * 3 + j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            if (NM) acc[mm & (NACC - 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mm & (NACC - 1)], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                float& x = v[n & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(seed));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (KIND == 2) {
                    f32x2& p = *reinterpret_cast<f32x2*>(&v[(2 * n) & 6]);
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(p));
                }
                if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
                if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(seed));
                if (KIND == 5) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(x) : "v"(seed));
                if (KIND == 6) {
                    f32x2& p = *reinterpret_cast<f32x2*>(&v[(2 * n) & 6]);
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
                }
                if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
                if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
                if (KIND == 10) {  // same mix as RAW chains: x -> fma -> exp -> (sum += x) -> cvt, two registers per MFMA
                    float& y = v[(n >> 2) & 1 ? 1 : 0];
                    switch (n & 3) {
                        case 0: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(seed)); break;
                        case 1: asm volatile("v_exp_f32 %0, %0" : "+v"(y)); break;
                        case 2: asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[7]) : "v"(y)); break;
                        default: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(v[6]) : "v"(y)); break;
                    }
                }
                if (KIND == 12) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(seed), "s"(0x07060302));
                if (KIND == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "s"(0));
                if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(seed));
                if (KIND == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
                if (KIND == 11) asm volatile("v_fmamk_f32 %0, %0, 0x3e3851ec, %1" : "+v"(x) : "v"(seed));
                if (KIND == 9) {   // the softmax mix per MFMA: 2 fma, 2 exp, 2 add, 1 max3, 1 cvt_pk on separate registers
                    switch (n & 7) {
                        case 0: case 1: asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(seed)); break;
                        case 2: case 3: asm volatile("v_exp_f32 %0, %0" : "+v"(x)); break;
                        case 4: case 5: asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(seed)); break;
                        case 6: asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(seed)); break;
                        default: asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(seed)); break;
                    }
                }
            }
        }
    }
    long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int NV, int KIND, int NM, int NACC = 2>
void run(float* out, int threads, const char* name) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, KIND, NM, NACC><<<256, threads>>>(out, 100, 1.0f);
    hipEventRecord(e0);
    k<NV, KIND, NM, NACC><<<256, threads>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s NM=%d NV=%2d waves/SIMD=%d : %.1f ns/iter  (%.1f ns per MFMA slot)\n", name, NM, NV, threads / 256,
           ms * 1e6 / iters, ms * 1e6 / iters / 4);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
#define ROW(KIND, NAME)                                                       \
    run<0, KIND, 4>(out, 256, NAME); run<0, KIND, 4>(out, 512, NAME);         \
    run<3, KIND, 4>(out, 256, NAME); run<3, KIND, 4>(out, 512, NAME);         \
    run<6, KIND, 4>(out, 256, NAME); run<6, KIND, 4>(out, 512, NAME);         \
    run<8, KIND, 4>(out, 256, NAME); run<8, KIND, 4>(out, 512, NAME);         \
    run<8, KIND, 0>(out, 256, NAME); run<8, KIND, 0>(out, 512, NAME);
#undef ROW
#define ROW(KIND, NAME)                                                       \
    run<0, KIND, 4>(out, 512, NAME); run<3, KIND, 4>(out, 512, NAME);         \
    run<6, KIND, 4>(out, 512, NAME); run<8, KIND, 0>(out, 512, NAME);
    ROW(0, "fma") ROW(2, "pk_fma") ROW(5, "dot2c_bf16") ROW(6, "pk_add") ROW(7, "add") ROW(8, "mul")
    run<8, 9, 4>(out, 256, "softmax-mix"); run<8, 9, 4>(out, 512, "softmax-mix");
    run<8, 9, 0>(out, 256, "softmax-mix"); run<8, 9, 0>(out, 512, "softmax-mix");
    run<8, 10, 4>(out, 256, "raw-chains"); run<8, 10, 4>(out, 512, "raw-chains");
    run<8, 10, 0>(out, 256, "raw-chains"); run<8, 10, 0>(out, 512, "raw-chains");
    run<8, 11, 4>(out, 256, "fmamk-lit"); run<8, 11, 4>(out, 512, "fmamk-lit");
    run<8, 11, 0>(out, 256, "fmamk-lit"); run<8, 11, 0>(out, 512, "fmamk-lit");
    run<8, 12, 4>(out, 256, "perm_b32"); run<8, 12, 4>(out, 512, "perm_b32"); run<8, 12, 0>(out, 256, "perm_b32"); run<8, 12, 0>(out, 512, "perm_b32");
    run<8, 13, 4>(out, 256, "cvt_pk_sgpr"); run<8, 13, 4>(out, 512, "cvt_pk_sgpr"); run<8, 13, 0>(out, 256, "cvt_pk_sgpr"); run<8, 13, 0>(out, 512, "cvt_pk_sgpr");
    run<8, 14, 4>(out, 256, "mov_b32"); run<8, 14, 4>(out, 512, "mov_b32"); run<8, 14, 0>(out, 256, "mov_b32"); run<8, 14, 0>(out, 512, "mov_b32");
    run<0, 9, 4, 1>(out, 256, "mix-1acc"); run<0, 9, 4, 1>(out, 512, "mix-1acc");
    run<8, 9, 4, 1>(out, 256, "mix-1acc"); run<8, 9, 4, 1>(out, 512, "mix-1acc");
    hipDeviceSynchronize();
    return 0;
}

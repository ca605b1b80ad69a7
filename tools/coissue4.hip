// Cross-wave co-execution: in a 512-thread workgroup (two waves per SIMD) waves 0-3 issue ONLY
// MFMAs and waves 4-7 ONLY VALU (the softmax op mix), no synchronisation.  Compare with each role
// alone.  This is the premise of a ping-pong attention schedule.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue4.hip -o tools/_coissue4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ROLE: 0 both roles present, 1 only the MFMA waves work, 2 only the VALU waves work
template <int ROLE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = seed;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(threadIdx.x + j); b[j] = (short)(threadIdx.x * 3 + j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x;
    if (wave < 4) {
        if (ROLE != 2)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) acc[mm & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mm & 1], 0, 0, 0);
            }
    } else {
        if (ROLE != 1)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int n = 0; n < 64; ++n) {   // 8 MFMA slots' worth: 16 fma, 16 exp, 16 add, 8 max3, 8 cvt
                    float& x = v[n & 7];
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
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ROLE>
float run(float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<ROLE><<<256, 512>>>(out, 100, 1.0f);
    (void)hipEventRecord(e0);
    k<ROLE><<<256, 512>>>(out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const float both = run<0>(out), mf = run<1>(out), va = run<2>(out);
    printf("ns per iteration (one wave: 8 MFMAs | the other wave on the SIMD: 64 softmax-mix VALU ops): both %.1f  MFMA wave alone %.1f  VALU wave alone %.1f\n", both, mf, va);
    (void)hipDeviceSynchronize();
    return 0;
}

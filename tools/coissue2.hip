// Co-issue of VALU with MFMA when the wave holds ~230 VGPRs (2 waves/SIMD from 2 workgroups/CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV, int NM>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = seed + a;
    bf16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (short)(threadIdx.x + j); fb[j] = (short)(threadIdx.x * 3 + j); }
    float v[64];
    for (int j = 0; j < 64; ++j) v[j] = seed + j + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) {
            if (NM) acc[mm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[mm], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                float& x = v[(mm * NV + n) & 63];
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(seed));
            }
        }
    }
    float s = 0;
    for (int a = 0; a < 8; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int j = 0; j < 64; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NM>
void run(float* out, int blocks) {
    const int iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, NM><<<blocks, 256>>>(out, 100, 1.0f);
    (void)hipEventRecord(e0);
    k<NV, NM><<<blocks, 256>>>(out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("big-VGPR fma NM=%d NV=%2d waves/SIMD=%d : %.1f ns per MFMA slot\n", NM, NV, blocks / 256, ms * 1e6 / iters / 8);
}
int main() {
    float* out;
    (void)hipMalloc(&out, 512 * 256 * 4);
    run<0, 1>(out, 256); run<0, 1>(out, 512);
    run<4, 1>(out, 256); run<4, 1>(out, 512);
    run<6, 1>(out, 256); run<6, 1>(out, 512);
    run<8, 1>(out, 256); run<8, 1>(out, 512);
    run<8, 0>(out, 256); run<8, 0>(out, 512);
    (void)hipDeviceSynchronize();
    return 0;
}

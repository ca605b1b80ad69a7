// Measured peaks of THIS MI355X (BASELINE.md §3 asks for them beside the datasheet 2.5 PFLOP/s / 8 TB/s that
// bench.py prices against): a register-only v_mfma_f32_32x32x16_bf16 loop (every SIMD busy, no memory) on zero
// and on random operands (the part is power-limited: MI355X_MICROARCH.md), and streaming reads / copies of a 4 GiB
// buffer at 16 B per lane.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/peaks.hip -o tools/_peaks && tools/_peaks > profiles/r02_peaks.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* in, float* out, int iters) {
    const bf16x8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[0] = s;   // keep the loop
}

// the same loop on the other shape / format: SHAPE 0 = 32x32x16, 1 = 16x16x32 (16 accumulators of 4 registers: the same
// 64 accumulator registers and the same FLOPs per iteration); F16: fp16 operands instead of bf16
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int SHAPE, bool F16>
__global__ __launch_bounds__(256) void mfma_loop2(const bf16x8* in, float* out, int iters);

template <bool F16>
__device__ __forceinline__ void loop32(const bf16x8 a, const bf16x8 b, float* out, int iters) {
    const f16x8 ah = __builtin_bit_cast(f16x8, a), bh = __builtin_bit_cast(f16x8, b);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // inline asm with the accumulators pinned in VGPRs: hipcc's allocator otherwise shuttles half of the 16x16
            // accumulators through v_accvgpr_mov every iteration
            if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(ah), "v"(bh));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[0] = s;
}
template <bool F16>
__device__ __forceinline__ void loop16(const bf16x8 a, const bf16x8 b, float* out, int iters) {
    const f16x8 ah = __builtin_bit_cast(f16x8, a), bh = __builtin_bit_cast(f16x8, b);
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(ah), "v"(bh));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}
template <int SHAPE, bool F16>
__global__ __launch_bounds__(256) void mfma_loop2(const bf16x8* in, float* out, int iters) {
    const bf16x8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    if constexpr (SHAPE == 0) loop32<F16>(a, b, out, iters);
    else loop16<F16>(a, b, out, iters);
}

__global__ __launch_bounds__(256) void stream_read(const f32x4* x, long n, float* out) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += __builtin_nontemporal_load(x + i);
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = s[0];
}
__global__ __launch_bounds__(256) void stream_copy(const f32x4* x, f32x4* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
}

template <typename F>
static double time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    std::vector<unsigned short> h(512 * 8);
    bf16x8* din;
    float* dout;
    hipMalloc(&din, 512 * 16);
    hipMalloc(&dout, 64);
    const int iters = 20000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    double tf[2];
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (auto& v : h) v = pass == 0 ? 0 : (unsigned short)(0x3C00 + (rand() & 0x3FF) + ((rand() & 1) << 15));   // |x| in [0.0078, 0.0156), random sign
        hipMemcpy(din, h.data(), 512 * 16, hipMemcpyHostToDevice);
        const double ms = time_ms([&] { mfma_loop<<<blocks, 256>>>(din, dout, iters); }, 3);
        tf[pass] = (double)blocks * 4 * iters * 4 * 32768.0 / (ms * 1e-3) / 1e12;
    }
    // round 3: MFMA shape x operand format on random operands (MI355X_MICROARCH.md "DVFS give-back" (7): the clock the chip
    // holds under load depends on the MFMA shape), 1 and 8 waves per SIMD, the four variants interleaved twice
    {
        srand(1);
        // values with random mantissas that are ordinary numbers in BOTH formats: bf16 0x3C00.. = 0.0078.., fp16 0x3C00.. = 1.0..
        for (auto& v : h) v = (unsigned short)(0x3C00 + (rand() & 0x3FF) + ((rand() & 1) << 15));
        hipMemcpy(din, h.data(), 512 * 16, hipMemcpyHostToDevice);
        printf("{\n \"mfma_shape_format_random_operands_TFLOPs\": {\n");
        for (int wps = 1; wps <= 8; wps *= 8) {
            const int nb = 256 * wps;
            double t[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                t[0] += time_ms([&] { mfma_loop2<0, false><<<nb, 256>>>(din, dout, iters); }, 3);
                t[1] += time_ms([&] { mfma_loop2<1, false><<<nb, 256>>>(din, dout, iters); }, 3);
                t[2] += time_ms([&] { mfma_loop2<0, true><<<nb, 256>>>(din, dout, iters); }, 3);
                t[3] += time_ms([&] { mfma_loop2<1, true><<<nb, 256>>>(din, dout, iters); }, 3);
            }
            const double fl = (double)nb * 4 * iters * 4 * 32768.0;
            printf("  \"%d_waves_per_simd\": {\"bf16_32x32x16\": %.0f, \"bf16_16x16x32\": %.0f, \"f16_32x32x16\": %.0f, \"f16_16x16x32\": %.0f}%s\n", wps,
                   fl / (t[0] / 2 * 1e-3) / 1e12, fl / (t[1] / 2 * 1e-3) / 1e12, fl / (t[2] / 2 * 1e-3) / 1e12, fl / (t[3] / 2 * 1e-3) / 1e12, wps == 1 ? "," : "");
        }
        printf(" }\n}\n");
    }
    const long n = (4l << 30) / 16;
    f32x4 *x, *y;
    hipMalloc(&x, n * 16);
    hipMalloc(&y, n * 16);
    hipMemset(x, 1, n * 16);
    const double rd = time_ms([&] { stream_read<<<256 * 16, 256>>>(x, n, dout); }, 5);
    const double cp = time_ms([&] { stream_copy<<<256 * 16, 256>>>(x, y, n); }, 5);
    printf("{\n \"device\": \"MI355X (gfx950)\",\n \"mfma_bf16_32x32x16_register_loop_TFLOPs\": {\"zero_operands\": %.0f, \"random_operands\": %.0f, \"datasheet_dense\": 2500},\n"
           " \"hbm_stream_4GiB_GBps\": {\"read\": %.0f, \"copy_read_plus_write\": %.0f, \"datasheet\": 8000},\n"
           " \"note\": \"bench.py prices roofline.frac against the datasheet peaks (MI355X_MICROARCH.md); these are what this box delivers on a loop with no memory traffic / no compute\"\n}\n",
           tf[0], tf[1], n * 16.0 / (rd * 1e-3) / 1e9, 2.0 * n * 16.0 / (cp * 1e-3) / 1e9);
    return 0;
}

// Shared host/device helpers for libskimi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/skimi.h"

namespace skimi {

// thread-local error text behind skimi_last_error()
void set_error(const char* fmt, ...);

#define SKIMI_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            ::skimi::set_error(__VA_ARGS__);  \
            return SKIMI_ERR_ARG;             \
        }                                     \
    } while (0)

#define SKIMI_HIP(call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ::skimi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                               __FILE__, __LINE__);                                      \
            return SKIMI_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)

#define SKIMI_LAUNCH_CHECK()                                                             \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            ::skimi::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),\
                               __FILE__, __LINE__);                                      \
            return SKIMI_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)

// Optional in-library kernel timer (bench.py's roofline leg): when armed for a kernel kind,
// every matching launch is bracketed by a hipEvent pair recorded on the launch stream.
enum ProfKind { PROF_NONE = 0, PROF_ATTN_BF16 = 1, PROF_GEMM = 2 };
bool prof_armed(int kind, long size_key);
void prof_before(hipStream_t st);
void prof_after(hipStream_t st, double flops, double bytes);

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator tile
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifdef __HIPCC__
// fp32 -> bf16, round to nearest even (plain cast: hipcc emits v_cvt_pk_bf16_f32, NaN-safe)
__device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short b) {
    return __builtin_bit_cast(float, ((unsigned int)b) << 16);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif

}  // namespace skimi

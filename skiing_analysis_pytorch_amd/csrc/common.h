// Shared host/device helpers for libskimi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
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

// Opt a kernel in to more than 64 KiB of dynamic LDS, once per (kernel, device): `done` is that kernel's own static
// mask (bit = device ordinal).  Thread-safe: two racing first calls both set the attribute, which is idempotent.
int lds_opt_in(std::atomic<unsigned>& done, const void* kernel, int bytes, const char* what);
#define SKIMI_LDS_OPT_IN(KERNEL, BYTES, WHAT)                                                            \
    do {                                                                                                 \
        static std::atomic<unsigned> done_{0};                                                           \
        const int rc_ = ::skimi::lds_opt_in(done_, reinterpret_cast<const void*>(KERNEL), (int)(BYTES), WHAT); \
        if (rc_ != SKIMI_OK) return rc_;                                                                 \
    } while (0)

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
// fp32 -> fp16, round to nearest even (v_cvt_f16_f32; NOT the round-toward-zero v_cvt_pkrtz), and back
__device__ __forceinline__ unsigned short f2h(float x) {
    _Float16 h = (_Float16)x;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float h2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
// fp32 -> the 16-bit operand format of a contraction: fp16 (SKIMI_F16) or bf16.  Both conversions are computed and
// one is selected (no branch: a branch in an epilogue costs a vmcnt(0) per store on gfx9)
__device__ __forceinline__ unsigned short f2x16(float x, bool f16) {
    const unsigned short a = f2h(x), b = f2bf(x);
    return f16 ? a : b;
}
// 16-bit activation element of either format -> fp32 (no branch: both conversions, one select)
__device__ __forceinline__ float x16tof(unsigned short b, bool f16) {
    const float a = h2f(b), c = bf2f(b);
    return f16 ? a : c;
}
// one k-step of a 32x32 output tile on 16-bit operands: fp16 (F16) or bf16 fragments, fp32 accumulate
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// MXFP8 (gemm_fp8.hip): E8M0 byte of a 32-element block = smallest power of two with amax / scale <= 448 (0: all zero),
// and its inverse as an exact power of two
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
    if (!(amax > 0.f)) return 0u;
    const float t = amax * (1.0f / 448.0f);
    const unsigned u = __float_as_uint(t);
    const int e = (int)((u >> 23) & 0xFF) - 127 + ((u & 0x7FFFFF) ? 1 : 0);
    return (unsigned)min(max(e + 127, 1), 254);
}
__device__ __forceinline__ float mx_inv_scale(unsigned sb) { return sb ? __uint_as_float((unsigned)(254 - (int)sb) << 23) : 0.f; }
// four values -> four e4m3 bytes (element 0 in the low byte)
__device__ __forceinline__ int mx_pack4(float a, float b, float c, float d, float inv) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a * inv, b * inv, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c * inv, d * inv, w, true);
    return w;
}
// maximum over the 8 lanes lane & ~7 .. lane | 7 (a 32-element block when every lane holds 4 consecutive elements)
// DPP lane permutes (quad_perm [1,0,3,2], [2,3,0,1], then row_half_mirror: lane i <-> 7 - i of each 8), no LDS traffic
__device__ __forceinline__ float max8(float v) {
    int x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, true)));
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, true)));
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, true)));
    return v;
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

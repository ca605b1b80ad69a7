// C-ABI surface of libskimi.so: error channel, version, generic ops.
#include <stdarg.h>

#include "common.h"
#include "kernels.h"

namespace skimi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace skimi

using namespace skimi;

extern "C" {

const char* skimi_last_error(void) { return g_err; }

int skimi_version(void) { return 100; }

int skimi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int skimi_gemm(const skimi_gemm_desc* d, void* stream) {
    if (!d) {
        set_error("skimi_gemm: null descriptor");
        return SKIMI_ERR_ARG;
    }
    return gemm_dispatch(d, (hipStream_t)stream, d->splitk_scratch, (size_t)d->splitk_scratch_bytes,
                         d->force_splitk);
}

int skimi_layernorm(const float* x, const float* x2, int64_t ldx, int64_t rows, int32_t C,
                    const float* gamma, const float* beta, float eps, void* out, int32_t out_dtype,
                    int64_t ldo, void* stream) {
    return layernorm_launch(x, x2, ldx, rows, C, gamma, beta, eps, out, out_dtype, ldo,
                            (hipStream_t)stream);
}

int skimi_qknorm_rope(void* qkv, int32_t dtype, int64_t tokens, int32_t heads, const float* qn_w,
                      const float* qn_b, const float* kn_w, const float* kn_b, float eps,
                      const int32_t* pos, const float* rope_cos, const float* rope_sin,
                      int32_t rope_npos, void* stream) {
    return qknorm_rope_launch(qkv, dtype, tokens, heads, qn_w, qn_b, kn_w, kn_b, eps, pos, rope_cos,
                              rope_sin, rope_npos, (hipStream_t)stream);
}

int skimi_attention(const void* qkv, void* out, int32_t dtype, int32_t batch, int32_t seq,
                    int32_t heads, int32_t head_dim, void* stream) {
    return attention_launch(qkv, out, dtype, batch, seq, heads, head_dim, (hipStream_t)stream);
}

}  // extern "C"

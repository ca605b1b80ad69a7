// C-ABI surface of libskimi.so: error channel, version, generic ops.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace skimi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int lds_opt_in(std::atomic<unsigned>& done, const void* kernel, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error("hipGetDevice failed (%s)", what);
        return SKIMI_ERR_HIP;
    }
    const unsigned bit = 1u << (dev & 31);
    if (done.load(std::memory_order_acquire) & bit) return SKIMI_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(%s, %d bytes of LDS) failed: %s", what, bytes, hipGetErrorString(e));
        return SKIMI_ERR_HIP;
    }
    done.fetch_or(bit, std::memory_order_release);
    return SKIMI_OK;
}

struct ProfState {
    int kind = PROF_NONE;
    long min_key = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double flops = 0, bytes = 0;
    hipEvent_t pending = nullptr;
};
static ProfState g_prof;
// launches may come from several host threads (one per stream): the event list is shared, the
// bracket of a launch is found again through the launching thread's own index
static std::mutex g_prof_mu;
static thread_local size_t tl_prof_idx = 0;

bool prof_armed(int kind, long size_key) { return g_prof.kind == kind && size_key >= g_prof.min_key; }
void prof_before(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::pair<hipEvent_t, hipEvent_t> e;
    if (!g_prof.pool.empty()) {
        e = g_prof.pool.back();
        g_prof.pool.pop_back();
    } else {
        (void)hipEventCreate(&e.first);
        (void)hipEventCreate(&e.second);
    }
    (void)hipEventRecord(e.first, st);
    g_prof.ev.push_back(e);
    tl_prof_idx = g_prof.ev.size() - 1;
}
void prof_after(hipStream_t st, double flops, double bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (tl_prof_idx >= g_prof.ev.size()) return;   // profile_stop ran in between
    (void)hipEventRecord(g_prof.ev[tl_prof_idx].second, st);
    g_prof.flops += flops;
    g_prof.bytes += bytes;
}

}  // namespace skimi

using namespace skimi;

extern "C" {

const char* skimi_last_error(void) { return g_err; }

int skimi_version(void) { return 100; }
int skimi_sizeof_gemm_desc(void) { return (int)sizeof(skimi_gemm_desc); }

int skimi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int skimi_profile_start(int32_t kind, int64_t min_size) {
    SKIMI_CHECK_ARG(kind == PROF_ATTN_BF16 || kind == PROF_GEMM, "skimi_profile_start: unknown kernel kind %d", kind);
    for (auto& e : g_prof.ev) g_prof.pool.push_back(e);
    g_prof.ev.clear();
    g_prof.flops = g_prof.bytes = 0;
    g_prof.kind = kind;
    g_prof.min_key = min_size;
    return SKIMI_OK;
}

int skimi_profile_stop(double* total_ms, int64_t* launches, double* flops, double* bytes) {
    double ms = 0;
    for (auto& e : g_prof.ev) {
        SKIMI_HIP(hipEventSynchronize(e.second));
        float t = 0;
        SKIMI_HIP(hipEventElapsedTime(&t, e.first, e.second));
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int64_t)g_prof.ev.size();
    if (flops) *flops = g_prof.flops;
    if (bytes) *bytes = g_prof.bytes;
    for (auto& e : g_prof.ev) g_prof.pool.push_back(e);
    g_prof.ev.clear();
    g_prof.kind = PROF_NONE;
    return SKIMI_OK;
}

int skimi_gemm(const skimi_gemm_desc* d, void* stream) {
    if (!d) {
        set_error("skimi_gemm: null descriptor");
        return SKIMI_ERR_ARG;
    }
    return gemm_dispatch(d, (hipStream_t)stream, d->splitk_scratch, (size_t)d->splitk_scratch_bytes,
                         d->force_splitk);
}

int skimi_resample_u8(const uint8_t* in, uint8_t* out, int64_t outer, int32_t n_in, int32_t n_out, int64_t inner,
                      const int32_t* kk, const int32_t* bounds, int32_t ksize, void* stream) {
    return resample_u8_launch(in, out, outer, n_in, n_out, inner, kk, bounds, ksize, (hipStream_t)stream);
}

int skimi_u8_hwc_to_f32_chw(const uint8_t* in, int32_t H, int32_t W, float* out, int32_t OH, int32_t OW, int32_t y_off,
                            int32_t x_off, float fill, void* stream) {
    return u8_hwc_to_f32_chw_launch(in, H, W, out, OH, OW, y_off, x_off, fill, (hipStream_t)stream);
}

int skimi_conv3x3_n32_pack(const float* w, void* packed, int32_t C, void* stream) {
    SKIMI_CHECK_ARG(w && packed && C > 0, "skimi_conv3x3_n32_pack: bad arguments");
    return conv_direct_pack_launch(w, (unsigned short*)packed, C, (hipStream_t)stream);
}

int skimi_conv3x3_n32(const void* in_hi, const void* in_lo, const void* packed_w, const float* bias, float* out,
                      int32_t F, int32_t H, int32_t W, int32_t C, int32_t relu, void* stream) {
    return conv_direct_n32_launch((const unsigned short*)in_hi, (const unsigned short*)in_lo,
                                  (const unsigned short*)packed_w, bias, out, F, H, W, C, relu, (hipStream_t)stream);
}

int skimi_split_planes(const float* x, int64_t ld, int64_t rows, int32_t C, void* hi, void* lo, void* stream) {
    SKIMI_CHECK_ARG(x && hi && lo && rows > 0 && C > 0, "skimi_split_planes: bad arguments");
    return split_planes_launch(x, (long)ld, (long)rows, C, hi, lo, (hipStream_t)stream);
}

int skimi_split_records(const float* x, int64_t ld, int64_t rows, int32_t C, void* records, void* stream) {
    SKIMI_CHECK_ARG(x && records && rows > 0 && C > 0, "skimi_split_records: bad arguments");
    return split_records_launch(x, (long)ld, (long)rows, C, records, (hipStream_t)stream);
}

int skimi_quant_mx(const void* x, int32_t dtype, int64_t ldx, int64_t rows, int32_t K, void* payload, void* scales,
                   void* stream) {
    return quant_mx_launch(x, dtype, (long)ldx, (long)rows, K, payload, scales, (hipStream_t)stream);
}

int skimi_layernorm_mx(const float* x, int64_t ldx, int64_t rows, int32_t C, const float* gamma, const float* beta,
                       float eps, void* payload, void* scales, void* stream) {
    return layernorm_mx_launch(x, ldx, rows, C, gamma, beta, eps, payload, scales, (hipStream_t)stream);
}

int skimi_gemm_fp8(const void* A, const void* A_scales, const void* W, const void* W_scales, int32_t M, int32_t N,
                   int32_t K, const float* bias, int32_t act, const float* gamma, const float* resid, int64_t ldr,
                   void* out, int32_t out_dtype, int64_t ldo, void* out_scales, void* stream) {
    return gemm_fp8_launch(A, A_scales, W, W_scales, M, N, K, bias, act, gamma, resid, (long)ldr, out, out_dtype, (long)ldo,
                           (hipStream_t)stream, out_scales);
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

int skimi_attention_out(const void* qkv, void* out, int32_t dtype, int32_t out_dtype, int32_t batch, int32_t seq,
                        int32_t heads, int32_t head_dim, void* stream) {
    if (out_dtype == SKIMI_FP8MX) {   // payload [tokens][Kp], then the scales [tokens][Kp / 32], Kp = heads * head_dim up to 128
        SKIMI_CHECK_ARG(out && batch > 0 && seq > 0 && attention_mx_output_ok(dtype, heads, head_dim),
                        "skimi_attention_out: MXFP8 rows need bf16 q / k / v, head_dim 64 and an even number of heads");
        const size_t Kp = align_up((size_t)heads * head_dim, 128);
        return attention_launch(qkv, nullptr, dtype, batch, seq, heads, head_dim, (hipStream_t)stream, 0, nullptr, 0, nullptr, 0,
                                out, (char*)out + (size_t)batch * seq * Kp);
    }
    SKIMI_CHECK_ARG(out_dtype == dtype || (dtype == SKIMI_BF16 && out_dtype == SKIMI_F16),
                    "skimi_attention_out: out_dtype is dtype, SKIMI_F16 or SKIMI_FP8MX for bf16 q / k / v (got %d -> %d)", dtype, out_dtype);
    return attention_launch(qkv, out, dtype, batch, seq, heads, head_dim, (hipStream_t)stream, 0, nullptr, 0, nullptr,
                            out_dtype == SKIMI_F16 && dtype == SKIMI_BF16);
}

}  // extern "C"

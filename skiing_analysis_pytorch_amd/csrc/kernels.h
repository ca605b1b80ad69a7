// Internal launch entry points shared between translation units of libskimi.
#pragma once
#include "common.h"

namespace skimi {

int gemm_dispatch(const skimi_gemm_desc* d, hipStream_t st, void* scratch, size_t scratch_bytes,
                  int force_splitk);

int layernorm_launch(const float* x, const float* x2, int64_t ldx, int64_t rows, int C,
                     const float* gamma, const float* beta, float eps, void* out, int out_dtype,
                     int64_t ldo, hipStream_t st, int64_t grp_rows = 0, int64_t grp_stride = 0,
                     int64_t grp_off = 0);

int qknorm_rope_launch(void* qkv, int dtype, int64_t tokens, int heads, const float* qn_w,
                       const float* qn_b, const float* kn_w, const float* kn_b, float eps,
                       const int32_t* pos, const float* rope_cos, const float* rope_sin, int rope_npos,
                       hipStream_t st, float q_scale = 1.f, int* q_scaled = nullptr);
// q_scale / q_scaled: the bf16 fast path can fold the softmax scale (x log2 e) into q BEFORE its one rounding
// to bf16 (attention_q64.hip then takes exp2 of the raw MFMA accumulator); *q_scaled says whether it did.

// general attention: q rows [batch, seq_q], k/v rows [batch, seq_k]; element strides
struct AttnArgs {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    long q_row, k_row, v_row, o_row;        // stride between consecutive tokens
    long q_batch, k_batch, v_batch, o_batch;  // stride between batches
    long q_head, k_head, v_head, o_head;    // stride between heads
    int batch, heads, seq_q, seq_k, head_dim;
    float scale;
    // optional two-level batch (fp32 kernel only): batch index g = go * batch_inner + gi sits at
    // go * *_batch2 + gi * *_batch (batch_inner = 0: one level, offset g * *_batch)
    int batch_inner = 0;
    long q_batch2 = 0, k_batch2 = 0, v_batch2 = 0, o_batch2 = 0;
    int q_prescaled = 0;   // bf16 kernels: q already carries scale * log2(e) (qknorm_rope_launch's q_scale)
    int out_f16 = 0;       // bf16 kernels: the output rows are written as fp16 (proj's operand under SKIMI_PREC_F16)
    // 64-query kernel only (attention_mx_output_ok): the output rows as MXFP8 instead of `out` -- payload [token][mx_row]
    // e4m3 bytes, one E8M0 byte per 32 channels in [token][mx_srow] (proj's A operand under SKIMI_PREC_FP8)
    void* out_mx = nullptr;
    void* out_mx_scales = nullptr;
    long mx_row = 0, mx_srow = 0;
};
// the bf16 attention launch can write its output as MXFP8 (one head = two 32-channel blocks, each inside one lane pair)
bool attention_mx_output_ok(int dtype, int heads, int head_dim);
int attention_f32_launch(const AttnArgs& a, hipStream_t st);
int attention_bf16_launch(const AttnArgs& a, hipStream_t st);
// preprocess.hip: Pillow-exact separable uint8 resample pass, uint8 HWC -> fp32 CHW / 255 with crop / pad
int resample_u8_launch(const unsigned char* in, unsigned char* out, long outer, int n_in, int n_out, long inner,
                       const int* kk, const int* bounds, int ksize, hipStream_t st);
int u8_hwc_to_f32_chw_launch(const unsigned char* in, int H, int W, float* out, int OH, int OW, int y_off, int x_off,
                             float fill, hipStream_t st);
// conv_direct.hip: 3x3, Cin % 32 == 0 -> 32 channels, pad 1, fp32-accurate, on bf16 hi/lo planes
int conv_direct_pack_launch(const float* w, unsigned short* out, int Cin, hipStream_t st);
int conv_direct_n32_launch(const unsigned short* in_hi, const unsigned short* in_lo, const unsigned short* w_packed,
                           const float* bias, float* out, int F, int H, int W, int C, int relu, hipStream_t st,
                           long px_stride = 0);   // 0: separate planes [.., C]; 2C: pixel records [hi C | lo C]
void attention_q64_dispatch(const AttnArgs& a, hipStream_t st);    // attention_q64.hip: 64 queries per wave
int attention_launch(const void* qkv, void* out, int dtype, int batch, int seq, int heads, int head_dim,
                     hipStream_t st, int q_prescaled = 0, void* x3_scratch = nullptr, size_t x3_scratch_bytes = 0,
                     int* out_records = nullptr,    // in: bf16x3 records wanted in `out` (fp32 mode); out: whether they were written
                     int out_f16 = 0,               // bf16 q / k / v, fp16 output rows (SKIMI_PREC_F16)
                     void* out_mx = nullptr, void* out_mx_scales = nullptr);   // attention_mx_output_ok(): MXFP8 rows instead of `out`
// attention_x3.hip: fp32-accurate attention (head_dim 64) on the bf16 matrix pipe, operands split hi + lo; needs
// scratch for the hi / lo planes of the packed qkv buffer (attention_launch uses it for fp32 inputs when given)
size_t attention_x3_scratch_bytes(long tokens, long row_elems);
int attention_x3_launch(const AttnArgs& a, long tokens, long row_elems, void* scratch, size_t scratch_bytes, hipStream_t st,
                        bool out_records = false);

// VideoPose3D expand-conv im2col: x [B, L, Cin] f32 -> A0 [B*(L-k+1), Kpad] f32 (zero padded)
int vp3d_im2col_launch(const float* x, float* a0, int B, int L, int Cin, int k, int Kpad, hipStream_t st);

// gemm_fp8.hip: MXFP8 operands (e4m3 + E8M0 per 32 K) and the scaled-MFMA contraction
int quant_mx_launch(const void* x, int dtype, long ldx, long rows, int K, void* q, void* scales, hipStream_t st);
int gemm_fp8_launch(const void* A, const void* As, const void* W, const void* Ws, int M, int N, int K, const float* bias, int act,
                    const float* gamma, const float* resid, long ldr, void* out, int out_dtype, long ldo, hipStream_t st,
                    void* out_scales = nullptr);
// LayerNorm straight into MXFP8 (C a multiple of 256): payload [rows][C] e4m3 + scales [rows][C / 32]
bool gemm_fp8_mx_output_ok(int M, int N);   // gemm_fp8_launch accepts out_dtype SKIMI_FP8MX for this shape (bias + GELU epilogue)
int layernorm_mx_launch(const float* x, int64_t ldx, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                        void* payload, void* scales, hipStream_t st);

// vp3d_stream.hip: the small-batch weight-streaming path of the TemporalModel (one launch per convolution)
int vp3d_expand_launch(const float* x, const float* w, int ldw, const float* bias, float* out_f32, void* out_rec, int B, int Lin,
                       int Cin, int taps, int C, hipStream_t st);
int vp3d_expand_mfma_launch(const float* x, const void* wfrag, int Kpad, const float* bias, float* out_f32, void* out_rec, int B,
                            int Lin, int Cin, int taps, int C, hipStream_t st);
int vp3d_mm_launch(const void* wrec, int Npad, const void* xrec, const float* bias, const float* resid, int resid_L,
                   int resid_off, float* out_f32, int ldo, void* out_rec, int B, int Lin, int C, int taps, int dil, int N,
                   int relu, hipStream_t st);

}  // namespace skimi

/*
 * skimi.h — C-ABI of libskimi.so, the MI355X (gfx950 / CDNA4) hot path of the
 * skiing multi-view 3D-pose pipeline.
 *
 * The reference (ChenKaiXuSan/Skiing_Analysis_PyTorch) has no FFI of its own: its
 * boundary for this path is two Python call sites,
 *     preds = self.vggt(imgs)                       vggt/vggt/infer.py:84
 *     predicted_3d_pos = model_pos(inputs_2d)       VideoPose3D/run.py:974
 * and two weight formats (flat state_dicts, vggt/vggt/infer.py:62-67 and
 * VideoPose3D/run.py:286-289).  Every entry point below is what a ctypes stub at
 * those call sites binds; each one names the reference
 * function it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary;
 *   - every pointer marked "dev" is a device (HBM) address, "host" a host address;
 *   - every call is asynchronous on the given hipStream_t (passed as void*; NULL =
 *     the null stream), never synchronises the host, never allocates in the launch
 *     path (handles allocate once at create/finalize time);
 *   - return value 0 = ok, negative = error; skimi_last_error() returns the text of
 *     the calling thread's last error;
 *   - activations are row-major, channels-last ("NHWC" for images / feature maps,
 *     [tokens, channels] for token streams).
 */
#ifndef SKIMI_H
#define SKIMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKIMI_OK 0
#define SKIMI_ERR_ARG (-1)      /* bad argument / shape mismatch */
#define SKIMI_ERR_HIP (-2)      /* HIP runtime error */
#define SKIMI_ERR_STATE (-3)    /* handle not finalized / missing weight */
#define SKIMI_ERR_WORKSPACE (-4)/* workspace too small */

/* element types of buffers that cross the ABI */
#define SKIMI_F32 0
#define SKIMI_BF16 1
#define SKIMI_BF16X3_REC 2 /* skimi_gemm_desc.a_dtype only: A already split into [hi 32 | lo 32] records */
#define SKIMI_FP8MX 3      /* skimi_gemm_fp8 out_dtype only: the result as MXFP8 (payload in out, E8M0 scales in out_scales) */
#define SKIMI_F16 4        /* IEEE binary16 (operands / activations of SKIMI_PREC_F16) */

/* arithmetic mode of the MFMA contractions
 *   SKIMI_PREC_BF16   : operands rounded to bf16, one v_mfma_f32_32x32x16_bf16 per
 *                       k-step, fp32 accumulate (the reference's GPU autocast mode,
 *                       vggt/vggt/infer.py:78-84)
 *   SKIMI_PREC_BF16X3 : operands kept in fp32 in HBM, split hi+lo into two bf16 while
 *                       staged to LDS, three MFMAs per k-step (hi*hi + hi*lo + lo*hi):
 *                       ~2^-17 relative operand error, the mode that meets the 1e-3
 *                       parity bar against the fp32 CPU reference */
#define SKIMI_PREC_BF16 0
#define SKIMI_PREC_BF16X3 1
/* SKIMI_PREC_FP8 (skimi_vggt_config.prec only; BASELINE config 5): as SKIMI_PREC_BF16, but the four Linear layers
 * (qkv, proj, fc1, fc2) of every DINOv2 / frame / global block (vggt/vggt/layers/block.py:77-98, mlp.py:34-40,
 * attention.py:50-72) run on the MXFP8 MFMA (skimi_gemm_fp8): weights quantised once at finalize, activations by
 * their producers (skimi_layernorm_mx; the attention kernel's output rows, skimi_attention_out with SKIMI_FP8MX;
 * fc1's epilogue with out_dtype SKIMI_FP8MX; skimi_quant_mx where the shape rules those out); the attention
 * products (bf16), the fp32 residual stream and the heads are unchanged. */
#define SKIMI_PREC_FP8 2
/* SKIMI_PREC_F16: the Linear layers of the DINOv2 / frame / global blocks and the patch embedding
 * (vggt/vggt/layers/block.py:77-98, mlp.py:34-40, attention.py:50-72, patch_embed.py:65-78) with fp16 operands on
 * v_mfma_f32_32x32x16_f16, fp32 accumulate: the same matrix rate as bf16 with three more mantissa bits (fp16 is the
 * reference's own autocast format on GPUs below compute capability 8, vggt/vggt/infer.py:77-82).  The operand
 * rounding of the Linears is what puts bf16 outside north_star's 1e-3 on the joints (profiles/r03_precision_ablation.md);
 * with fp16 there the joints are inside it.  LayerNorm, the attention epilogue and fc1's GELU epilogue write fp16;
 * the attention products (QK^T, PV: they average their rounding noise over all keys) keep bf16 q, k, v, P; residual
 * stream, LayerNorm, softmax and accumulation are fp32 as in every mode.  In skimi_gemm_desc: fp16 (or fp32, rounded
 * to fp16 while staged) A and W, a_dtype / w_dtype SKIMI_F16 / SKIMI_F32. */
#define SKIMI_PREC_F16 3

/* activation in the GEMM epilogue */
#define SKIMI_ACT_NONE 0
#define SKIMI_ACT_RELU 1
#define SKIMI_ACT_GELU 2   /* erf form, torch.nn.GELU() default (vggt/vggt/layers/mlp.py:26) */
#define SKIMI_ACT_SILU 3
#define SKIMI_ACT_SIGMOID 4

const char* skimi_last_error(void);
int skimi_version(void);
/* sizeof(skimi_gemm_desc) as this library was built: a binding checks its own struct against it */
int skimi_sizeof_gemm_desc(void);
/* number of HIP devices visible; does not initialise a context on any */
int skimi_device_count(void);

/* Measurement hook (bench.py roofline leg): while armed, every launch of the given kernel
 * kind (1 = bf16 flash attention with seq_k >= min_size, 2 = MFMA contraction with
 * M >= min_size) is bracketed by a hipEvent pair on its own stream.  skimi_profile_stop
 * synchronises those events and returns the summed device time, the launch count and the
 * algorithmic FLOPs / bytes of the bracketed launches. */
int skimi_profile_start(int32_t kind, int64_t min_size);
int skimi_profile_stop(double* total_ms, int64_t* launches, double* flops, double* bytes);

/* ------------------------------------------------------------------------- */
/* Generic fused contraction  out = epilogue( gather(A) . W^T )               */
/* Replaces torch.nn.Linear / Conv1d / Conv2d / ConvTranspose2d(k==s) on the  */
/* path (vggt/vggt/layers/{attention,mlp,patch_embed}.py, heads/dpt_head.py,  */
/* VideoPose3D/common/model.py:126-138).                                      */
/* ------------------------------------------------------------------------- */
typedef struct skimi_gemm_desc {
    int32_t M, N, K;          /* out rows, out cols, contraction length (K % 8 == 0) */
    const void* A;            /* dev; [rows, lda] f32 or bf16, channels-last */
    const void* W;            /* dev; [N, ldw] f32 (BF16X3) or bf16 (BF16): nn.Linear layout */
    int32_t a_dtype, w_dtype; /* SKIMI_F32 / SKIMI_BF16 (SKIMI_F16 under SKIMI_PREC_F16); a_dtype SKIMI_BF16X3_REC: see W_split */
    int64_t lda, ldw;         /* in elements */
    int32_t prec;             /* SKIMI_PREC_* */
    /* A gather: a_mode 0 = plain rows; 1 = implicit im2col of a channels-last image
     * [cN, cH, cW, cC] with a KH x KW window (tap-major K: k = (ky*KW+kx)*cC + c),
     * M = cN*OH*OW, K = KH*KW*cC, cC % BK == 0 (BK = 64 for BF16, 32 for BF16X3);
     * 2 = the same gather with slice-major K (BF16X3 only, cC % 32 == 0):
     * k = ((c / 32) * KH*KW + ky*KW + kx) * 32 + c % 32, i.e. weights [N][cC/32][KH][KW][32] */
    int32_t a_mode;
    int32_t cN, cH, cW, cC, KH, KW, stride, pad, dil, OH, OW;
    /* epilogue: v = acc + bias[n]; v = act(v); v *= gamma[n]; v += resid[m', n];
     * resid row m' = m + resid_row_off (+ (m / resid_rows_per_batch) * resid_batch_skip) */
    const float* bias;        /* dev [N] or NULL */
    const float* gamma;       /* dev [N] or NULL */
    const void* resid;        /* dev or NULL; f32 or bf16 per resid_dtype */
    int32_t resid_dtype;      /* SKIMI_F32 / SKIMI_BF16, applies to resid and resid2 */
    int64_t ldr;
    int32_t resid_rows_per_batch; /* 0 = no batching of the residual row map */
    int64_t resid_batch_stride;   /* rows between consecutive batches in resid */
    int64_t resid_row_off;
    int32_t act;
    /* then: v += resid2[m, n] (plain row m, leading dim ldr2); v = post_act(v) */
    const void* resid2;       /* dev or NULL */
    int64_t ldr2;
    int32_t post_act;
    /* store_mode 0 output row remap, same form as the residual's:
     * row = (m / out_rows_per_batch) * out_batch_stride + m % out_rows_per_batch + out_row_off
     * (out_rows_per_batch 0 = row m + out_row_off) */
    int32_t out_rows_per_batch;
    int64_t out_batch_stride;
    int64_t out_row_off;
    /* store: store_mode 0 = out[row*ldo + n]; 1 = ConvTranspose2d with kernel == stride
     * (ps_s): m = (img, iy, ix) over [cN, cH, cW], n = (a*ps_s + b)*ps_C + co,
     * out[((img*cH*ps_s + iy*ps_s + a)*cW*ps_s + ix*ps_s + b)*ldo + co] */
    void* out;                /* dev; f32, bf16 or fp16 */
    void* out2;               /* dev or NULL: second copy in the other dtype (same indexing, ldo2) */
    int32_t out_dtype;
    int64_t ldo, ldo2;
    int32_t store_mode, ps_s, ps_C;
    /* optional caller-owned fp32 scratch of >= M*N*4 bytes: lets skinny shapes (too few
     * output tiles to fill 256 CUs) run split-K; NULL = never split.  force_splitk > 0
     * pins the split count (tests). */
    void* splitk_scratch;
    uint64_t splitk_scratch_bytes;
    int32_t force_splitk;
    /* 1 = the caller guarantees the scratch is all zero on entry (every split-K launch leaves it
     * zeroed again), so no memset is issued; 0 = the launch zeroes what it needs first */
    int32_t splitk_scratch_zeroed;
    /* optional fast path of SKIMI_PREC_BF16X3 for large shapes (M >= 4096, N >= 96, enough 256-row
     * tiles to fill the chip): W_split = the same weights as bf16 records [N][ceil(K/32)][hi 32 | lo 32]
     * (skimi_split_records), and x3_scratch = caller-owned scratch of >= 4 bytes per element of the A
     * buffer the launch touches (rows of ceil(C/32)*32 elements) + 256, where A is split once into
     * the same records and from where both operands stream through LDS-DMA.  Both NULL, or a smaller
     * scratch = generic kernel.
     * a_dtype SKIMI_BF16X3_REC: A already IS those records (of the [rows, C] buffer the launch
     * touches; lda ignored) and x3_scratch points to >= 256 zero bytes that lie behind the records
     * within 4 GiB of A (padding taps read them); such a launch must qualify for the fast path
     * (skimi_gemm returns SKIMI_ERR_ARG otherwise). */
    const void* W_split;
    void* x3_scratch;
    uint64_t x3_scratch_bytes;
    /* dev or NULL: the result also (or, with out == NULL, only) as bf16x3 records
     * [M][ceil(N/32)][hi 32 | lo 32] followed by 256 zero bytes (4 * M * ceil(N/32) * 32 + 256 bytes,
     * 128-byte aligned; N % 32 == 0, plain output rows): the a_dtype SKIMI_BF16X3_REC operand of a
     * following contraction, written by this launch's epilogue instead of a separate split pass */
    void* out_records;
} skimi_gemm_desc;

/* fp32 [rows, C] (row stride ld elements) -> bf16 planes hi[rows, C], lo[rows, C]:
 * hi = bf16(x), lo = bf16(x - hi)  (operand form of SKIMI_PREC_BF16X3's fast path) */
int skimi_split_planes(const float* x, int64_t ld, int64_t rows, int32_t C, void* hi, void* lo, void* stream);

/* fp32 [rows, C] (row stride ld elements, C % 4 == 0) -> bf16 records [rows][ceil(C/32)][hi 32 | lo 32]
 * (4 * rows * ceil(C/32) * 32 bytes; a ragged last slice is zero-filled): the hi and lo halves of
 * a 32-element K-slice share one 128-byte line (operand form of skimi_gemm_desc.W_split) */
int skimi_split_records(const float* x, int64_t ld, int64_t rows, int32_t C, void* records, void* stream);

int skimi_gemm(const skimi_gemm_desc* d, void* stream);

/* OCP microscaling FP8 (MXFP8) operands of skimi_gemm_fp8: x [rows, K] (bf16 or fp32, row stride ldx elements,
 * 16-byte aligned rows) -> payload [rows][Kp] e4m3 bytes (Kp = K rounded up to 128, tail zero) and scales
 * [rows][Kp / 32] E8M0 bytes (value 2^(byte - 127); per 32-element block the smallest power of two with
 * amax / scale <= 448, so no element clips; 0 for an all-zero block). */
int skimi_quant_mx(const void* x, int32_t dtype, int64_t ldx, int64_t rows, int32_t K, void* payload, void* scales,
                   void* stream);
/* LayerNorm(C, affine, eps) of fp32 rows (vggt/vggt/layers/block.py:77-98's norm1 / norm2) written directly as the
 * MXFP8 operand of the following skimi_gemm_fp8: payload [rows][C], scales [rows][C / 32]; the same bytes as
 * skimi_quant_mx of the fp32 LayerNorm result.  C a multiple of 256 (<= 2048), ldx % 4 == 0. */
int skimi_layernorm_mx(const float* x, int64_t ldx, int64_t rows, int32_t C, const float* gamma, const float* beta,
                       float eps, void* payload, void* scales, void* stream);
/* out[m][n] = epilogue(sum_k A[m][k] W[n][k]) on v_mfma_scale_f32_32x32x64_f8f6f4, both operands as written by
 * skimi_quant_mx (nn.Linear layout for W: [N, K]).  Epilogue: + bias[n] (or NULL); act = SKIMI_ACT_NONE or
 * SKIMI_ACT_GELU; or, with gamma != NULL, gamma[n] * (. + bias[n]) + resid[m][n] (fp32, row stride ldr; may alias
 * out: block.py:77-98's LayerScale + residual).  out fp32 or bf16, row stride ldo; N, ldo, ldr multiples of 4.
 * out_dtype SKIMI_FP8MX (with out_scales != NULL; large shapes with the GELU epilogue only, N % 128 == 0): the
 * result is written directly in the operand form of the NEXT skimi_gemm_fp8 -- payload [M][N] bytes in out
 * (ldo = N) and scales [M][N / 32] in out_scales -- so the MLP's hidden activation never exists in bf16. */
int skimi_gemm_fp8(const void* A, const void* A_scales, const void* W, const void* W_scales, int32_t M, int32_t N,
                   int32_t K, const float* bias, int32_t act, const float* gamma, const float* resid, int64_t ldr,
                   void* out, int32_t out_dtype, int64_t ldo, void* out_scales, void* stream);

/* Direct 3x3 convolution (stride 1, pad 1) of a channels-last image to 32 output channels in the
 * fp32-accurate mode: the last conv of the DPT heads at full resolution
 * (vggt/vggt/heads/dpt_head.py:224-235, scratch.output_conv2[0]).  The input comes as the two bf16
 * planes of skimi_split_planes ([F, H, W, C] each, C % 32 == 0), the weights in the packed layout
 * that skimi_conv3x3_n32_pack makes of the reference's [32, C, 3, 3] tensor
 * (2 * 32 * C * 9 bf16).  out: fp32 [F, H, W, 32]; bias [32] or NULL; relu != 0 applies ReLU. */
int skimi_conv3x3_n32_pack(const float* w, void* packed, int32_t C, void* stream);
int skimi_conv3x3_n32(const void* in_hi, const void* in_lo, const void* packed_w, const float* bias, float* out,
                      int32_t F, int32_t H, int32_t W, int32_t C, int32_t relu, void* stream);

/* ------------------------------------------------------------------------- */
/* Image preprocessing on the device (vggt/load.py:38-183)                    */
/* ------------------------------------------------------------------------- */
/* One separable pass of Pillow's 8-bit resampler (what Image.resize(..., BICUBIC) runs on the host
 * in the reference): element (o, n, i) of `in` lives at (o * n_in + n) * inner + i; `kk` is
 * [n_out, ksize] int32 22-bit fixed-point coefficients, `bounds` [n_out, 2] = (first input index,
 * tap count), both built on the host as Pillow builds them (skiing_analysis_pytorch_amd/preprocess.py).
 * Horizontal pass of an HWC image: outer = H, inner = C; vertical pass: outer = 1, inner = W * C. */
int skimi_resample_u8(const uint8_t* in, uint8_t* out, int64_t outer, int32_t n_in, int32_t n_out, int64_t inner,
                      const int32_t* kk, const int32_t* bounds, int32_t ksize, void* stream);
/* uint8 HWC (3 channels) -> fp32 [3, OH, OW] = value / 255; output pixel (y, x) reads input pixel
 * (y + y_off, x + x_off), pixels outside the input are `fill` (centre crop / white padding). */
int skimi_u8_hwc_to_f32_chw(const uint8_t* in, int32_t H, int32_t W, float* out, int32_t OH, int32_t OW, int32_t y_off,
                            int32_t x_off, float fill, void* stream);

/* ------------------------------------------------------------------------- */
/* Row-wise ops on token streams                                              */
/* ------------------------------------------------------------------------- */
/* LayerNorm over the last dim C of x[rows, C] (optionally the concatenation of two
 * sources x and x2 of C/2 channels each: the [frame | global] intermediates of
 * vggt/vggt/models/aggregator.py:250-253).  out dtype f32 or bf16.
 * Replaces torch.nn.LayerNorm (block.py:49,66; dpt_head.py:56,223). gamma/beta may be
 * NULL (elementwise_affine=False, heads/camera_head.py:66). */
int skimi_layernorm(const float* x, const float* x2, int64_t ldx, int64_t rows, int32_t C,
                    const float* gamma, const float* beta, float eps,
                    void* out, int32_t out_dtype, int64_t ldo, void* stream);

/* q/k LayerNorm(head_dim) + 2D RoPE applied in place on a packed qkv buffer
 * [tokens, 3, heads, 64] (f32 or bf16).  pos is dev int32 [tokens, 2] (y, x);
 * rope_cos/rope_sin are dev f32 [rope_npos, 16] tables (cos/sin of pos * 1/base^(i/16),
 * built on the host as rope.py:86-117 does).  Replaces attention.py:54-58 +
 * rope.py:154-188.  qn_w/kn_w NULL = no norm; pos NULL = no rope. */
int skimi_qknorm_rope(void* qkv, int32_t dtype, int64_t tokens, int32_t heads,
                      const float* qn_w, const float* qn_b, const float* kn_w, const float* kn_b,
                      float eps, const int32_t* pos, const float* rope_cos, const float* rope_sin,
                      int32_t rope_npos, void* stream);

/* Scaled-dot-product attention, no mask, softmax scale 1/sqrt(head_dim)
 * (F.scaled_dot_product_attention, attention.py:60-61).
 * qkv: [batch*seq, 3, heads, head_dim]; out: [batch*seq, heads*head_dim], same dtype.
 * dtype bf16 + head_dim 64 runs the MFMA flash kernel; f32 runs the exact fp32 kernel. */
int skimi_attention(const void* qkv, void* out, int32_t dtype, int32_t batch, int32_t seq,
                    int32_t heads, int32_t head_dim, void* stream);
/* The same with the output type named: out_dtype == dtype, or SKIMI_F16 with dtype SKIMI_BF16 -- the form
 * SKIMI_PREC_F16 runs (bf16 q / k / v and probabilities, the result rows rounded once to fp16: the operand of
 * the proj Linear, attention.py:62-64) -- or SKIMI_FP8MX with dtype SKIMI_BF16, head_dim 64 and an even number of
 * heads: the form SKIMI_PREC_FP8 runs, the result rows as skimi_gemm_fp8's A operand, quantised from the fp32
 * quotient: `out` = e4m3 payload [batch*seq][Kp] followed by the E8M0 scales [batch*seq][Kp/32], Kp =
 * heads*head_dim rounded up to 128 (pad columns are not written). */
int skimi_attention_out(const void* qkv, void* out, int32_t dtype, int32_t out_dtype, int32_t batch, int32_t seq,
                        int32_t heads, int32_t head_dim, void* stream);

/* ------------------------------------------------------------------------- */
/* VideoPose3D TemporalModel lifter  (VideoPose3D/common/model.py:79-138)     */
/* ------------------------------------------------------------------------- */
typedef struct skimi_vp3d skimi_vp3d;

/* filter_widths: e.g. {3,3,3}; causal as TemporalModel(causal=...) */
skimi_vp3d* skimi_vp3d_create(int32_t joints_in, int32_t in_features, int32_t joints_out,
                              const int32_t* filter_widths, int32_t n_widths,
                              int32_t channels, int32_t causal);
void skimi_vp3d_destroy(skimi_vp3d*);
/* one state_dict entry, by its reference key name ("expand_conv.weight",
 * "layers_bn.0.running_var", "shrink.bias", ...; VideoPose3D/run.py:288-289).
 * data is a host fp32 array of n elements (num_batches_tracked is ignored). */
int skimi_vp3d_set_weight(skimi_vp3d*, const char* key, const float* host_data, int64_t n);
/* fold BatchNorm (eval mode, model.py:127,134-135) into conv weight + bias, repack
 * [Cout,Cin,k] -> [Cout,k,Cin], upload.  prec selects the MFMA mode. */
int skimi_vp3d_finalize(skimi_vp3d*, int32_t prec);
int32_t skimi_vp3d_receptive_field(const skimi_vp3d*);   /* model.py:41-48 */
size_t skimi_vp3d_workspace_bytes(const skimi_vp3d*, int32_t batch, int32_t frames_in);
/* x: dev f32 [batch, frames_in, joints_in, in_features]; out: dev f32
 * [batch, frames_in - rf + 1, joints_out, 3]  (model.py:63-77) */
int skimi_vp3d_forward(skimi_vp3d*, const float* x, float* out, int32_t batch, int32_t frames_in,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* VGGT forward  (vggt/vggt/models/vggt.py:29-96; call site infer.py:84)      */
/* ------------------------------------------------------------------------- */
typedef struct skimi_vggt skimi_vggt;

/* Shape parameters; the defaults of the reference are VGGT() = VGGT-1B
 * (vggt.py:18-27, aggregator.py:51-70, camera_head.py:26-37, dpt_head.py:40-53,
 * track_head.py:18-29).  head_dim = embed_dim / num_heads must be 64. */
typedef struct skimi_vggt_config {
    int32_t patch_size, embed_dim, depth, num_heads, num_register_tokens;
    int32_t use_dino;          /* 1: DINOv2 ViT patch embed ("dinov2_vit*14_reg"); 0: conv patch embed */
    int32_t dino_depth, dino_heads, dino_img_size;   /* dino_img_size: side the pos_embed grid was built for */
    int32_t cam_trunk_depth, cam_heads, cam_iters;
    int32_t dpt_features, dpt_out_channels[4], dpt_layers[4];
    int32_t track_features, track_hidden, track_corr_levels, track_corr_radius, track_iters, track_depth,
        track_heads, track_virtual;
    int32_t enable_camera, enable_depth, enable_point, enable_track;
    /* MFMA mode of the patch embedding and the DINOv2 / frame / global blocks: SKIMI_PREC_BF16 (the reference's
     * autocast, infer.py:78-84), SKIMI_PREC_F16 (fp16 operands: the joints stay within 1e-3 of the fp32 path),
     * SKIMI_PREC_BF16X3 (fp32-accurate) or SKIMI_PREC_FP8 */
    int32_t prec;
    /* MFMA mode of the depth / point DPT heads, which the reference runs in fp32 (torch.cuda.amp.autocast(enabled=False),
     * vggt.py:65): BF16X3 = faithful (the default everywhere); F16 / BF16 = 16-bit operands and activations, faster and
     * less accurate (fp16: depth ~1e-3 worst-case relative, profiles/r03_head_precision.json).  The camera head is
     * fp32-accurate in every mode. */
    int32_t head_prec;
} skimi_vggt_config;

skimi_vggt* skimi_vggt_create(const skimi_vggt_config* cfg);
void skimi_vggt_destroy(skimi_vggt*);
/* one entry of the reference state_dict by key ("aggregator.frame_blocks.3.attn.qkv.weight",
 * "depth_head.scratch.refinenet1.resConfUnit1.conv1.weight", ...; infer.py:62-67).
 * data: n fp32 elements on the host (on_device = 0) or already in HBM (on_device = 1). */
int skimi_vggt_set_weight(skimi_vggt*, const char* key, const float* data, int64_t n, int32_t on_device);
/* check every key of the configured model is present with the right size, repack
 * (conv taps, bf16 copies, padded K) and release the staged fp32 copies */
int skimi_vggt_finalize(skimi_vggt*);
/* DINOv2 positional embedding for an input size other than the one the model was built for:
 * pos_embed = [1 + (H/patch)*(W/patch), embed_dim] fp32 — row 0 the class position, the rest the
 * 37x37 grid resized with bicubic + antialias exactly as interpolate_pos_encoding does
 * (vggt/vggt/layers/vision_transformer.py:180-212).  A per-resolution constant, computed once by
 * the host side (skiing_analysis_pytorch_amd/vggt.py); without it such sizes are rejected. */
int skimi_vggt_set_pos_embed(skimi_vggt*, int32_t H, int32_t W, const float* pos_embed, int32_t on_device);
size_t skimi_vggt_workspace_bytes(skimi_vggt*, int32_t B, int32_t S, int32_t H, int32_t W, int32_t n_query);
/* The RoPE position table the forward uses for `frames` frames of H x W: dev int32 [frames, P, 2] (y, x), P = 1 +
 * num_register_tokens + (H/patch)*(W/patch); patches carry (row + 1, column + 1), the special tokens (0, 0)
 * (PositionGetter, vggt/vggt/layers/rope.py:39-59, + the offset of aggregator.py:219-228).  An index path: the
 * parity tests compare it bit for bit.  Synchronous copy; builds the handle's table for this shape if needed. */
int skimi_vggt_rope_positions(skimi_vggt*, int32_t frames, int32_t H, int32_t W, int32_t* positions);

/* device output buffers (fp32); a NULL pointer skips the store (a head whose outputs are all
 * NULL is not run).  Shapes as the reference's prediction dict (vggt.py:40-53). */
typedef struct skimi_vggt_outputs {
    float* pose_enc;           /* [B, S, 9] last iteration */
    float* pose_enc_list;      /* [cam_iters, B, S, 9] */
    float* depth;              /* [B, S, H, W, 1] */
    float* depth_conf;         /* [B, S, H, W] */
    float* world_points;       /* [B, S, H, W, 3] */
    float* world_points_conf;  /* [B, S, H, W] */
    float* track;              /* [B, S, N, 2] last iteration */
    float* vis;                /* [B, S, N] */
    float* conf;               /* [B, S, N] */
    float* tokens_last;        /* [B, S, P, 2*embed_dim]: aggregated_tokens_list[-1] (tests) */
} skimi_vggt_outputs;

/* images: dev f32 [B, S, 3, H, W] in [0, 1]; query_points: dev f32 [B, N, 2] pixels or NULL.
 * Errors mirror the reference: channels != 3 cannot be expressed (the layout fixes 3);
 * H or W not a multiple of patch_size -> SKIMI_ERR_ARG (patch_embed.py:69-70).
 * Concurrency: calls of any shapes (B, S, H, W) may run at the same time on one handle from several
 * host threads, each with its own workspace and stream (the handle's per-shape position / RoPE / UV
 * tables are a keyed cache of immutable entries, built under a lock on a shape's first call). */
int skimi_vggt_forward(skimi_vggt*, const float* images, const float* query_points, int32_t B, int32_t S,
                       int32_t H, int32_t W, int32_t n_query, const skimi_vggt_outputs* out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Geometry post-processing on device                                         */
/* ------------------------------------------------------------------------- */
/* pose_enc [rows, 9] (T, quat XYZW, fov_h, fov_w) -> extrinsic [rows, 3, 4] (cam-from-world,
 * OpenCV) and intrinsic [rows, 3, 3] (may be NULL) for an H x W image.
 * Replaces pose_encoding_to_extri_intri / quat_to_mat (vggt/vggt/utils/pose_enc.py:62-124,
 * rotation.py:14-44). */
int skimi_pose_to_cameras(const float* pose_enc, int64_t rows, int32_t H, int32_t W, float* extrinsic,
                          float* intrinsic, void* stream);
/* depth [frames, H, W] + cameras -> world points [frames, H, W, 3].
 * Replaces unproject_depth_map_to_point_map (vggt/vggt/utils/geometry.py:15-117), which the
 * reference runs in NumPy on the host after a D2H copy of the dense maps (infer.py:92-104). */
int skimi_unproject_depth(const float* depth, const float* extrinsic, const float* intrinsic,
                          float* world_points, int32_t frames, int32_t H, int32_t W, void* stream);
/* DLT triangulation: K [steps, views, 3, 3], R [steps, views, 3, 3], t [steps, views, 3],
 * keypoints [steps, views, joints, 2] pixels -> joints3d [steps, joints, 3].
 * Replaces triangulate_point / triangulate_one_frame (vggt/triangulate.py:13-71) for
 * views = 2 and generalises the same linear system to more views. */
int skimi_triangulate_dlt(const float* K, const float* R, const float* t, const float* keypoints,
                          float* joints3d, int64_t steps, int32_t views, int32_t joints, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SKIMI_H */

// Launch wrappers of vggt_kernels.hip (HBM-bound helpers of the VGGT forward).
#pragma once
#include "common.h"

namespace skimi {

int patch_gather_launch(const float* img, void* out, int out_dtype, int F, int H, int W, int p, int Kp,
                        hipStream_t st);
int special_tokens_launch(float* x, const float* table, int F, int S, int P, int n, int C, hipStream_t st);
int bilinear_ac_planes_launch(const float* in, unsigned short* out, int N, int h, int w, int H, int W, int C, hipStream_t st,
                              const float* tabx = nullptr, const float* taby = nullptr,
                              int slice_records = 0, void* zpage = nullptr);   // out: [N, H, W, hi C | lo C] bf16
int bilinear_ac_launch(const void* in, void* out, int dtype, int N, int h, int w, int H, int W, int C,
                       hipStream_t st, const float* tabx = nullptr, const float* taby = nullptr, int out_dtype = -1,   // out_dtype -1: the input's; SKIMI_F32 from a 16-bit map
                       const float* ln_g = nullptr, const float* ln_b = nullptr, float ln_eps = 0.f);   // C == 128, fp32 out: LayerNorm of every resized pixel in the same pass
int add_uv_pos_launch(void* x, int dtype, const float* tabx, const float* taby, int N, int H, int W, int C,
                      hipStream_t st);
int adaln_launch(const float* xn, const float* x, const float* mod, float* out, long rows, int D, hipStream_t st);
int pose_update_launch(const float* delta, float* pred_pad, float* act_out, long rows, int first, hipStream_t st);
int dpt_out_launch(const void* in, int dtype, const float* W, const float* b, int n_out, float* pts, float* conf,
                   long npix, int mode, hipStream_t st);
int f32_to_bf16_launch(const float* in, void* out, long n, hipStream_t st);
int f32_to_f16_launch(const float* in, void* out, long n, hipStream_t st);
int permute_conv_launch(const float* in, float* out, int Co, int Ci, int kh, int kw, hipStream_t st, int slice_major = 0);
int permute_convT_launch(const float* in, float* out, int Ci, int Co, int s, hipStream_t st);
int pad_cols_launch(const float* in, float* out, long rows, int K, int Kp, hipStream_t st);
int tile_vec_launch(const float* in, float* out, int n, int reps, hipStream_t st);

}  // namespace skimi

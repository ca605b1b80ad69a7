// Launch wrappers of track_kernels.hip (VGGT track head helpers).
#pragma once
#include "common.h"

namespace skimi {

int avgpool2_launch(const float* in, float* out, int N, int H, int W, int C, hipStream_t st);
int sample_border_launch(const float* fmap, long img_stride, const float* coords, long coord_stride, float* out, int B,
                         int N, int H, int W, int C, hipStream_t st);
int corr_sample_launch(const float* tgt, const float* fmap, const float* coords, float* out, long rows, int N, int S,
                       int H, int W, int C, int r, int level, long ldo, int out_off, hipStream_t st);
int pos_embed_sample_launch(const float* coords, long coord_stride, float* out, int BN, int H, int W, int D,
                            hipStream_t st);
int track_input_launch(const float* coords, const float* fcorr, const float* tfeat, const float* pos, const float* qrt,
                       float* x, long rows, int S, int L, long ldx, float max_scale, hipStream_t st);
int track_coord_update_launch(float* coords, const float* delta, long ldd, const float* query, float* pred, long rows,
                              int N, int S, float stride, hipStream_t st);
int track_init_launch(const float* q, float* coords, float* qs, long BN, int S, float stride, hipStream_t st);
int repeat_rows_launch(const float* src, float* dst, long BN, int S, int C, hipStream_t st);
int bns_to_bsn_launch(const float* in, float* out, int B, int N, int S, hipStream_t st);

}  // namespace skimi

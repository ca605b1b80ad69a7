// VideoPose3D TemporalModel forward (VideoPose3D/common/model.py:79-138) as a chain of fused
// MFMA contractions on channels-last activations [B, L, C]:
//   expand_conv(k) -> BN -> ReLU                    one GEMM over zero-padded windows (im2col of 34 ch)
//   per block i: conv(k, dilation 3^i) -> BN -> ReLU  implicit-gather GEMM, K = k*C (tap-major)
//                conv1x1 -> BN -> ReLU, + res         GEMM with the residual slice fused in the epilogue
//   shrink (1x1, bias)                              GEMM N = joints_out*3
// Eval-mode BatchNorm is affine, so it is folded into the conv weight and bias once at finalize
// (model.py:127,134-135); dropout is identity in eval.  Activations stay fp32 in HBM; the MFMA
// operands are bf16 (PREC_BF16) or split bf16 hi+lo (PREC_BF16X3, ~fp32 accuracy).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

using namespace skimi;

struct skimi_vp3d {
    int joints_in, in_features, joints_out, channels, causal;
    std::vector<int> fw;          // filter widths
    std::vector<int> pad;         // model.py:31,106
    std::vector<int> causal_shift;
    std::vector<int> dilation;    // dilation of block i's first conv (i >= 1)
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    int prec = SKIMI_PREC_BF16X3;
    int k0pad = 0;                // padded K of the expand GEMM
    // device weights (fp32 or bf16 depending on prec) + fp32 biases
    void* w_expand = nullptr;
    float* b_expand = nullptr;
    std::vector<void*> w_conv;    // 2 per block
    std::vector<void*> w_rec;     // fp32-accurate mode: the same matrices as bf16x3 records (LDS-DMA kernel, large batches)
    std::vector<float*> b_conv;
    void* w_shrink = nullptr;
    float* b_shrink = nullptr;
    // fp32-accurate mode, small batches (vp3d_stream.hip): every conv's weights pre-split (hi + lo bf16) in the
    // fragment-major order of that kernel; shrink rows padded to a multiple of 16 with zeros
    std::vector<void*> w_frag;      // 2 per block
    void* w_expand_frag = nullptr;  // [C][k0frag], K = taps * Cin zero-padded to a multiple of 32
    int k0frag = 0;
    void* w_shrink_frag = nullptr;
    int shrink_npad = 0;
    std::vector<void*> allocs;
};

static int upload(skimi_vp3d* h, const std::vector<float>& src, bool as_bf16, void** out) {
    void* d = nullptr;
    if (as_bf16) {
        std::vector<unsigned short> tmp(src.size());
        for (size_t i = 0; i < src.size(); ++i) {
            // round to nearest even, as the device cast does
            unsigned int u;
            memcpy(&u, &src[i], 4);
            unsigned int r = u + 0x7FFFu + ((u >> 16) & 1u);
            tmp[i] = (unsigned short)(r >> 16);
        }
        SKIMI_HIP(hipMalloc(&d, tmp.size() * 2));
        SKIMI_HIP(hipMemcpy(d, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    } else {
        SKIMI_HIP(hipMalloc(&d, src.size() * 4));
        SKIMI_HIP(hipMemcpy(d, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    }
    h->allocs.push_back(d);
    *out = d;
    return SKIMI_OK;
}

extern "C" {

skimi_vp3d* skimi_vp3d_create(int32_t joints_in, int32_t in_features, int32_t joints_out,
                              const int32_t* filter_widths, int32_t n_widths, int32_t channels,
                              int32_t causal) {
    if (joints_in <= 0 || in_features <= 0 || joints_out <= 0 || !filter_widths || n_widths <= 0 ||
        channels <= 0 || channels % 64 != 0) {
        set_error("skimi_vp3d_create: bad arguments (channels must be a multiple of 64)");
        return nullptr;
    }
    for (int i = 0; i < n_widths; ++i)
        if (filter_widths[i] % 2 == 0 || filter_widths[i] <= 0) {
            // model.py:20-21
            set_error("skimi_vp3d_create: Only odd filter widths are supported");
            return nullptr;
        }
    skimi_vp3d* h = new skimi_vp3d();
    h->joints_in = joints_in;
    h->in_features = in_features;
    h->joints_out = joints_out;
    h->channels = channels;
    h->causal = causal;
    h->fw.assign(filter_widths, filter_widths + n_widths);
    // model.py:31,105-110
    h->pad.push_back(h->fw[0] / 2);
    h->causal_shift.push_back(causal ? h->fw[0] / 2 : 0);
    h->dilation.push_back(1);
    int next_dilation = h->fw[0];
    for (int i = 1; i < n_widths; ++i) {
        h->pad.push_back((h->fw[i] - 1) * next_dilation / 2);
        h->causal_shift.push_back(causal ? (h->fw[i] / 2) * next_dilation : 0);
        h->dilation.push_back(next_dilation);
        next_dilation *= h->fw[i];
    }
    return h;
}

void skimi_vp3d_destroy(skimi_vp3d* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int skimi_vp3d_set_weight(skimi_vp3d* h, const char* key, const float* host_data, int64_t n) {
    SKIMI_CHECK_ARG(h && key && host_data && n > 0, "skimi_vp3d_set_weight: bad arguments");
    h->host[key].assign(host_data, host_data + n);
    h->finalized = false;
    return SKIMI_OK;
}

int32_t skimi_vp3d_receptive_field(const skimi_vp3d* h) {
    if (!h) return 0;
    int frames = 0;
    for (int p : h->pad) frames += p;
    return 1 + 2 * frames;
}

static int need(skimi_vp3d* h, const std::string& k, size_t n, const std::vector<float>** out) {
    auto it = h->host.find(k);
    if (it == h->host.end()) {
        set_error("skimi_vp3d_finalize: missing weight '%s'", k.c_str());
        return SKIMI_ERR_STATE;
    }
    if (it->second.size() != n) {
        set_error("skimi_vp3d_finalize: weight '%s' has %zu elements, expected %zu", k.c_str(),
                  it->second.size(), n);
        return SKIMI_ERR_STATE;
    }
    *out = &it->second;
    return SKIMI_OK;
}

// fold BN(eval) into a conv: W'[co][tap][ci] = W[co][ci][tap]*s[co]; b'[co] = beta - mean*s
static int fold(skimi_vp3d* h, const std::string& conv, const std::string& bn, int cout, int cin, int k,
                int kpad, std::vector<float>* w, std::vector<float>* b) {
    const std::vector<float>*cw, *g, *be, *mu, *var;
    int rc;
    if ((rc = need(h, conv + ".weight", (size_t)cout * cin * k, &cw))) return rc;
    if ((rc = need(h, bn + ".weight", cout, &g))) return rc;
    if ((rc = need(h, bn + ".bias", cout, &be))) return rc;
    if ((rc = need(h, bn + ".running_mean", cout, &mu))) return rc;
    if ((rc = need(h, bn + ".running_var", cout, &var))) return rc;
    w->assign((size_t)cout * kpad, 0.f);
    b->assign(cout, 0.f);
    for (int co = 0; co < cout; ++co) {
        const float s = (*g)[co] / sqrtf((*var)[co] + 1e-5f);   // BatchNorm1d eps default
        (*b)[co] = (*be)[co] - (*mu)[co] * s;
        for (int t = 0; t < k; ++t)
            for (int ci = 0; ci < cin; ++ci)
                (*w)[(size_t)co * kpad + (size_t)t * cin + ci] = (*cw)[((size_t)co * cin + ci) * k + t] * s;
    }
    return SKIMI_OK;
}

// Weights [N][K] fp32 -> hi / lo bf16 in the fragment-major order of vp3d_mm_kernel:
// [Npad / 16][K / 32][hi | lo][kg 0..3][row 0..15][8 elements]; rows N..Npad-1 are zero.
static unsigned short bf16_rne(float x) {
    unsigned int u;
    memcpy(&u, &x, 4);
    const unsigned int r = u + 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
static int upload_frag(skimi_vp3d* h, const std::vector<float>& w, int N, int Npad, int K, void** out) {
    const size_t S = (size_t)K / 32;
    std::vector<unsigned short> buf((size_t)Npad * K * 2, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float x = w[(size_t)n * K + k];
            const unsigned short hi = bf16_rne(x);
            unsigned int hu = (unsigned int)hi << 16;
            float hf;
            memcpy(&hf, &hu, 4);
            const size_t idx = (((size_t)(n >> 4) * S + (size_t)(k >> 5)) * 1024) + (size_t)((((k & 31) >> 3) * 16 + (n & 15)) * 8 + (k & 7));
            buf[idx] = hi;
            buf[idx + 512] = bf16_rne(x - hf);
        }
    void* d = nullptr;
    SKIMI_HIP(hipMalloc(&d, buf.size() * 2));
    h->allocs.push_back(d);
    SKIMI_HIP(hipMemcpy(d, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
    *out = d;
    return SKIMI_OK;
}

// fp32-accurate mode: a block's weight matrix [N, K] also as bf16x3 records, for the LDS-DMA kernel that
// gemm_dispatch picks once a batch fills the chip with 256-row tiles (gemm_x3dma.hip)
static int add_records(skimi_vp3d* h, const void* w_f32, int N, int K, bool bf16_mode) {
    void* rec = nullptr;
    if (!bf16_mode && K % 32 == 0) {
        SKIMI_HIP(hipMalloc(&rec, (size_t)N * K * 4));
        h->allocs.push_back(rec);
        int rc = split_records_launch((const float*)w_f32, K, N, K, rec, nullptr);
        if (rc) return rc;
        SKIMI_HIP(hipStreamSynchronize(nullptr));
    }
    h->w_rec.push_back(rec);
    return SKIMI_OK;
}

int skimi_vp3d_finalize(skimi_vp3d* h, int32_t prec) {
    SKIMI_CHECK_ARG(h, "skimi_vp3d_finalize: null handle");
    SKIMI_CHECK_ARG(prec == SKIMI_PREC_BF16 || prec == SKIMI_PREC_BF16X3, "skimi_vp3d_finalize: bad prec");
    for (void* p : h->allocs) (void)hipFree(p);
    h->allocs.clear();
    h->w_conv.clear();
    h->w_rec.clear();
    h->w_frag.clear();
    h->w_shrink_frag = nullptr;
    h->w_expand_frag = nullptr;
    h->b_conv.clear();
    h->prec = prec;
    const bool bf = prec == SKIMI_PREC_BF16;
    const int C = h->channels, cin0 = h->joints_in * h->in_features;
    std::vector<float> w, b;
    int rc;
    h->k0pad = (int)align_up((size_t)h->fw[0] * cin0, 8);
    if ((rc = fold(h, "expand_conv", "expand_bn", C, cin0, h->fw[0], h->k0pad, &w, &b))) return rc;
    if ((rc = upload(h, w, bf, &h->w_expand))) return rc;
    if ((rc = upload(h, b, false, (void**)&h->b_expand))) return rc;
    h->w_expand_frag = nullptr;
    if (!bf && C % 32 == 0) {
        h->k0frag = (int)align_up((size_t)h->fw[0] * cin0, 32);
        std::vector<float> wp((size_t)C * h->k0frag, 0.f);
        for (int co = 0; co < C; ++co)
            for (int k = 0; k < h->fw[0] * cin0; ++k) wp[(size_t)co * h->k0frag + k] = w[(size_t)co * h->k0pad + k];
        if ((rc = upload_frag(h, wp, C, C, h->k0frag, &h->w_expand_frag))) return rc;
    }
    for (size_t i = 1; i < h->fw.size(); ++i) {
        char cn[64], bn[64];
        void* dw;
        float* db;
        snprintf(cn, sizeof cn, "layers_conv.%zu", 2 * (i - 1));
        snprintf(bn, sizeof bn, "layers_bn.%zu", 2 * (i - 1));
        if ((rc = fold(h, cn, bn, C, C, h->fw[i], h->fw[i] * C, &w, &b))) return rc;
        if ((rc = upload(h, w, bf, &dw))) return rc;
        if ((rc = upload(h, b, false, (void**)&db))) return rc;
        h->w_conv.push_back(dw);
        h->b_conv.push_back(db);
        if ((rc = add_records(h, dw, C, h->fw[i] * C, bf))) return rc;
        if (!bf && C % 32 == 0) {
            void* fr;
            if ((rc = upload_frag(h, w, C, C, h->fw[i] * C, &fr))) return rc;
            h->w_frag.push_back(fr);
        }
        snprintf(cn, sizeof cn, "layers_conv.%zu", 2 * (i - 1) + 1);
        snprintf(bn, sizeof bn, "layers_bn.%zu", 2 * (i - 1) + 1);
        if ((rc = fold(h, cn, bn, C, C, 1, C, &w, &b))) return rc;
        if ((rc = upload(h, w, bf, &dw))) return rc;
        if ((rc = upload(h, b, false, (void**)&db))) return rc;
        h->w_conv.push_back(dw);
        h->b_conv.push_back(db);
        if ((rc = add_records(h, dw, C, C, bf))) return rc;
        if (!bf && C % 32 == 0) {
            void* fr;
            if ((rc = upload_frag(h, w, C, C, C, &fr))) return rc;
            h->w_frag.push_back(fr);
        }
    }
    const std::vector<float>*sw, *sb;
    const int nout = h->joints_out * 3;
    if ((rc = need(h, "shrink.weight", (size_t)nout * C, &sw))) return rc;
    if ((rc = need(h, "shrink.bias", nout, &sb))) return rc;
    if ((rc = upload(h, *sw, bf, &h->w_shrink))) return rc;
    if ((rc = upload(h, *sb, false, (void**)&h->b_shrink))) return rc;
    if (!bf && C % 32 == 0) {
        h->shrink_npad = (int)align_up((size_t)nout, 16);
        if ((rc = upload_frag(h, *sw, nout, h->shrink_npad, C, &h->w_shrink_frag))) return rc;
    }
    h->finalized = true;
    return SKIMI_OK;
}

// workspace = A0 [B*L0, k0pad] + three activation buffers [B*L0, C] + split-K slab of four [B*L0, C] planes (the
//             fp32-accurate mode sums its K splits in a fixed order, one plane per split: gemm.hip)
//           + two buffers of pre-split (hi + lo bf16) activations ([B*L0 rounded up to 16, C] + 256 each): records for the
//             LDS-DMA kernels, fragment-major tiles for the small-batch streaming kernels
size_t skimi_vp3d_workspace_bytes(const skimi_vp3d* h, int32_t batch, int32_t frames_in) {
    if (!h || batch <= 0 || frames_in < skimi_vp3d_receptive_field(h)) return 0;
    const size_t L0 = (size_t)frames_in - h->fw[0] + 1;
    const size_t rows = (size_t)batch * L0;
    const size_t k0 = align_up((size_t)h->fw[0] * h->joints_in * h->in_features, 8);
    return align_up(rows * k0 * 4, 256) + 7 * align_up(rows * h->channels * 4, 256) + 2 * align_up(align_up(rows, 16) * h->channels * 4 + 256, 256);
}

int skimi_vp3d_forward(skimi_vp3d* h, const float* x, float* out, int32_t batch, int32_t frames_in,
                       void* workspace, size_t workspace_bytes, void* stream) {
    SKIMI_CHECK_ARG(h && x && out && workspace, "skimi_vp3d_forward: null argument");
    if (!h->finalized) {
        set_error("skimi_vp3d_forward: weights not finalized");
        return SKIMI_ERR_STATE;
    }
    const int rf = skimi_vp3d_receptive_field(h);
    SKIMI_CHECK_ARG(batch > 0 && frames_in >= rf, "skimi_vp3d_forward: need frames_in >= receptive field %d", rf);
    const size_t wsneed = skimi_vp3d_workspace_bytes(h, batch, frames_in);
    if (workspace_bytes < wsneed) {
        set_error("skimi_vp3d_forward: workspace %zu < %zu", workspace_bytes, wsneed);
        return SKIMI_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int C = h->channels, cin0 = h->joints_in * h->in_features;
    const int L0 = frames_in - h->fw[0] + 1;
    const size_t rows0 = (size_t)batch * L0;
    char* ws = (char*)workspace;
    float* a0 = (float*)ws;
    ws += align_up(rows0 * h->k0pad * 4, 256);
    const size_t actb = align_up(rows0 * C * 4, 256);
    float* bufX = (float*)ws;
    float* bufY = (float*)(ws + actb);
    float* bufZ = (float*)(ws + 2 * actb);
    void* slab = ws + 3 * actb;
    const size_t recb = align_up(align_up(rows0, 16) * C * 4 + 256, 256);
    char* recX = ws + 7 * actb;     // records of bufX (block input), or split scratch when the chain is off
    char* recY = recX + recb;       // records of the dilated conv's output
    const int wdt = h->prec == SKIMI_PREC_BF16 ? SKIMI_BF16 : SKIMI_F32;

    int rc;
    // Small batches, fp32-accurate mode: the weight-streaming path -- one launch per convolution, no split-K,
    // no im2col, no slab (vp3d_stream.hip).  SKIMI_VP3D_STREAM=0 forces the generic GEMM chain (A/B timing);
    // SKIMI_VP3D_STREAM_ROWS moves the switch-over (rows of the first layer; larger batches are MFMA-bound and
    // run on the LDS-DMA kernels below).
    {
        static const int use_stream = getenv("SKIMI_VP3D_STREAM") ? atoi(getenv("SKIMI_VP3D_STREAM")) : 1;
        static const long max_rows = getenv("SKIMI_VP3D_STREAM_ROWS") ? atol(getenv("SKIMI_VP3D_STREAM_ROWS")) : 2048;
        const bool ok = use_stream && h->prec == SKIMI_PREC_BF16X3 && h->w_shrink_frag != nullptr && (long)rows0 <= max_rows &&
                        h->w_frag.size() == 2 * (h->fw.size() - 1);
        if (ok) {
            // expand_conv: on the MFMA kernel (bf16x3 like the other layers; SKIMI_VP3D_EXPAND_VALU=1: the exact-fp32 VALU kernel)
            static const int expand_valu = getenv("SKIMI_VP3D_EXPAND_VALU") ? atoi(getenv("SKIMI_VP3D_EXPAND_VALU")) : 0;
            if (expand_valu || h->w_expand_frag == nullptr)
                rc = vp3d_expand_launch(x, (const float*)h->w_expand, h->k0pad, h->b_expand, bufX, recX, batch, frames_in, cin0,
                                        h->fw[0], C, st);
            else
                rc = vp3d_expand_mfma_launch(x, h->w_expand_frag, h->k0frag, h->b_expand, bufX, recX, batch, frames_in, cin0,
                                             h->fw[0], C, st);
            if (rc) return rc;
            int L = L0;
#ifdef SKIMI_ABLATIONS
            int mm_left = getenv("SKIMI_VP3D_LAST") ? atoi(getenv("SKIMI_VP3D_LAST")) + 1 : 1 << 30;   // tools/vp3d_phases.py
#define SKIMI_VP3D_COUNT() if (--mm_left <= 0) return SKIMI_OK
#else
#define SKIMI_VP3D_COUNT()
#endif
            for (size_t i = 1; i < h->fw.size(); ++i) {
                const int k = h->fw[i], dil = h->dilation[i];
                const int Lo = L - (k - 1) * dil;
                SKIMI_CHECK_ARG(Lo > 0, "skimi_vp3d_forward: sequence too short");
                // conv k, dilated (+ BN + ReLU): records in, records out
                if ((rc = vp3d_mm_launch(h->w_frag[2 * (i - 1)], C, recX, h->b_conv[2 * (i - 1)], nullptr, 0, 0, nullptr, C, recY,
                                         batch, L, C, k, dil, C, 1, st))) return rc;
                SKIMI_VP3D_COUNT();
                // conv 1x1 (+ BN + ReLU) + res = x[:, pad+shift : L-pad+shift] (model.py:129-135): the block's
                // output as fp32 (residual of the next block) and as records (operand of the next layer)
                if ((rc = vp3d_mm_launch(h->w_frag[2 * (i - 1) + 1], C, recY, h->b_conv[2 * (i - 1) + 1], bufX, L,
                                         h->pad[i] + h->causal_shift[i], bufZ, C, recX, batch, Lo, C, 1, 1, C, 1, st))) return rc;
                SKIMI_VP3D_COUNT();
                float* t = bufX;
                bufX = bufZ;
                bufZ = t;
                L = Lo;
            }
            // shrink: 1x1 conv with bias, no activation
            return vp3d_mm_launch(h->w_shrink_frag, h->shrink_npad, recX, h->b_shrink, nullptr, 0, 0, out, h->joints_out * 3,
                                  nullptr, batch, L, C, 1, 1, h->joints_out * 3, 0, st);
        }
    }
    if ((rc = vp3d_im2col_launch(x, a0, batch, frames_in, cin0, h->fw[0], h->k0pad, st))) return rc;

    skimi_gemm_desc d;
    memset(&d, 0, sizeof d);
    d.prec = h->prec;
    d.a_dtype = SKIMI_F32;
    d.w_dtype = wdt;
    d.out_dtype = SKIMI_F32;
    d.splitk_scratch = slab;
    d.splitk_scratch_bytes = 4 * actb;
    d.splitk_scratch_zeroed = 1;   // zeroed once here; every split-K epilogue leaves it zero again
    SKIMI_HIP(hipMemsetAsync(slab, 0, 4 * actb, st));
    d.act = SKIMI_ACT_RELU;

    // Large batches (fp32-accurate mode): when every block GEMM qualifies for the LDS-DMA bf16x3 kernel with
    // records as its A operand, activations travel from layer to layer as records written by the producing
    // epilogue (no split pass; the dilated conv's output exists as records only).
    bool chain = h->prec == SKIMI_PREC_BF16X3 && !h->w_rec.empty() && h->w_rec[0] != nullptr;
    if (chain) {
        int Lc = L0;
        for (size_t i = 1; i < h->fw.size() && chain; ++i) {
            const int k = h->fw[i], dil = h->dilation[i];
            const int Lo = Lc - (k - 1) * dil;
            skimi_gemm_desc q = d;
            char* fake = (char*)(uintptr_t)0x10000000;
            q.a_dtype = SKIMI_BF16X3_REC; q.A = fake; q.lda = C; q.W = fake; q.out = fake;
            q.a_mode = 1; q.cN = batch; q.cH = 1; q.cW = Lc; q.cC = C; q.KH = 1; q.KW = k;
            q.stride = 1; q.pad = 0; q.dil = dil; q.OH = 1; q.OW = Lo;
            q.M = batch * Lo; q.N = C; q.K = k * C; q.W_split = h->w_rec[2 * (i - 1)];
            q.x3_scratch = fake + (size_t)batch * Lc * C * 4; q.x3_scratch_bytes = 256;
            chain = chain && Lo > 0 && gemm_x3dma_eligible(&q);
            q.a_mode = 0; q.K = C; q.W_split = h->w_rec[2 * (i - 1) + 1];
            q.x3_scratch = fake + (size_t)batch * Lo * C * 4;
            chain = chain && gemm_x3dma_eligible(&q);
            Lc = Lo;
        }
    }

    // expand: [B*L0, k0pad] x [C, k0pad]^T
    d.M = (int)rows0; d.N = C; d.K = h->k0pad;
    d.A = a0; d.lda = h->k0pad;
    d.W = h->w_expand; d.ldw = h->k0pad;
    d.bias = h->b_expand;
    d.out = bufX; d.ldo = C;
    d.out_records = chain ? recX : nullptr;
    if ((rc = gemm_dispatch(&d, st, d.splitk_scratch, d.splitk_scratch_bytes, 0))) return rc;
    d.out_records = nullptr;

    int L = L0;   // frames held by bufX, per batch element
    for (size_t i = 1; i < h->fw.size(); ++i) {
        const int k = h->fw[i], dil = h->dilation[i];
        const int Lo = L - (k - 1) * dil;
        SKIMI_CHECK_ARG(Lo > 0, "skimi_vp3d_forward: sequence too short");
        // conv k, dilated: gather over the [B, 1, L, C] "image"
        d.a_mode = 1;
        d.cN = batch; d.cH = 1; d.cW = L; d.cC = C; d.KH = 1; d.KW = k;
        d.stride = 1; d.pad = 0; d.dil = dil; d.OH = 1; d.OW = Lo;
        d.M = batch * Lo; d.N = C; d.K = k * C;
        d.lda = C;
        d.W = h->w_conv[2 * (i - 1)]; d.ldw = (int64_t)k * C;
        d.W_split = h->w_rec[2 * (i - 1)];   // qualifies (gemm_x3dma_eligible) from a few dozen clips per call on
        d.bias = h->b_conv[2 * (i - 1)];
        d.resid = nullptr;
        d.ldo = C;
        if (chain) {
            d.A = recX; d.a_dtype = SKIMI_BF16X3_REC;
            d.x3_scratch = recX + (size_t)batch * L * C * 4; d.x3_scratch_bytes = 256;
            d.out = nullptr; d.out_records = recY;
        } else {
            d.A = bufX; d.a_dtype = SKIMI_F32;
            d.x3_scratch = recX; d.x3_scratch_bytes = recb;
            d.out = bufY; d.out_records = nullptr;
        }
        if ((rc = gemm_dispatch(&d, st, d.splitk_scratch, d.splitk_scratch_bytes, 0))) return rc;
        // conv 1x1 + BN + ReLU, then + res = x[:, pad+shift : L-pad+shift]  (model.py:129-135)
        d.a_mode = 0;
        d.K = C;
        d.lda = C;
        if (chain) {
            d.A = recY;
            d.x3_scratch = recY + (size_t)batch * Lo * C * 4;
            d.out_records = recX;   // the block's output: fp32 (residual of the next block, shrink) and records
        } else {
            d.A = bufY;
        }
        d.W = h->w_conv[2 * (i - 1) + 1]; d.ldw = C;
        d.W_split = h->w_rec[2 * (i - 1) + 1];
        d.bias = h->b_conv[2 * (i - 1) + 1];
        d.resid = bufX; d.ldr = C;
        d.resid_rows_per_batch = Lo;
        d.resid_batch_stride = L;
        d.resid_row_off = h->pad[i] + h->causal_shift[i];
        // the epilogue reads residual rows written by other workgroups' inputs, so the
        // result goes to a third buffer and the buffers rotate
        d.out = bufZ;
        if ((rc = gemm_dispatch(&d, st, d.splitk_scratch, d.splitk_scratch_bytes, 0))) return rc;
        float* t = bufX;
        bufX = bufZ;
        bufZ = t;
        d.resid = nullptr;
        d.resid_rows_per_batch = 0;
        d.resid_batch_stride = 0;
        d.resid_row_off = 0;
        L = Lo;
    }
    // shrink: 1x1 conv with bias, no activation
    d.W_split = nullptr; d.x3_scratch = nullptr; d.x3_scratch_bytes = 0;
    d.a_dtype = SKIMI_F32; d.out_records = nullptr;
    d.a_mode = 0;
    d.act = SKIMI_ACT_NONE;
    d.M = batch * L; d.N = h->joints_out * 3; d.K = C;
    d.A = bufX; d.lda = C;
    d.W = h->w_shrink; d.ldw = C;
    d.bias = h->b_shrink;
    d.out = out; d.ldo = h->joints_out * 3;
    if ((rc = gemm_dispatch(&d, st, d.splitk_scratch, d.splitk_scratch_bytes, 0))) return rc;
    return SKIMI_OK;
}

}  // extern "C"

// VGGT forward on MI355X: the graph of fused MFMA contractions, flash attention and row-wise
// kernels that replaces `preds = self.vggt(imgs)` (vggt/vggt/infer.py:84).
//
// Data layout in HBM (all channels-last, row-major):
//   residual stream x      fp32 [F*P, C]          F = B*S frames, P = 1 + R + ph*pw tokens
//   GEMM operands          bf16 (PREC_BF16) or fp32 (PREC_BF16X3) [rows, C | 3C | 4C]
//   kept intermediates     fp32 [F*P, C] x {frame, global} for the 4 DPT layers + the last layer;
//                          the reference's torch.cat([frame, global], -1) (aggregator.py:250-253) is
//                          never materialised: consumers read the two halves through two pointers
//   DPT feature maps       [F, h, w, C] (NHWC), 3x3 / strided / transposed convs as implicit-gather GEMMs
// Weights are repacked once in skimi_vggt_finalize (conv taps -> [Cout, ky, kx, Cin], ConvTranspose
// -> pixel-shuffle GEMM, bf16 copies, K padded to 8).  The forward allocates nothing: every
// activation lives in the caller's workspace through a bump arena whose peak is computed by a
// dry run of the same code (skimi_vggt_workspace_bytes).
#include <math.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <atomic>
#include <mutex>
#include <vector>

#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"
#include "track_kernels.h"
#include "vggt_kernels.h"

using namespace skimi;

namespace {

struct Lin {              // out = A . W^T (+ b)
    void* w = nullptr;    // [N, ldw] bf16 or f32
    int wdt = SKIMI_F32;
    float* b = nullptr;
    int N = 0, K = 0;     // K as seen by the GEMM (padded)
    int prec = SKIMI_PREC_BF16X3;
    void* w_split = nullptr;   // BF16X3 only: bf16 [hi 32 | lo 32] records for the LDS-DMA kernel (wide 3x3 convs)
    void* wq = nullptr;        // SKIMI_PREC_FP8 blocks: the same matrix as MXFP8 payload [N][Kp] + scales [N][Kp / 32]
    void* wq_scales = nullptr;
};
struct LNw { float* g = nullptr; float* b = nullptr; };
struct BlockW {
    LNw n1, n2;
    Lin qkv, proj, fc1, fc2;
    float *ls1 = nullptr, *ls2 = nullptr;
    float *qn_w = nullptr, *qn_b = nullptr, *kn_w = nullptr, *kn_b = nullptr;
};
struct ConvW { Lin lin; int k = 1, stride = 1, pad = 0, cin = 0; int kmode = 1; };   // kmode: a_mode of the gather (1 tap-major, 2 slice-major K)
struct FusionW { Lin out_conv; ConvW r1c1, r1c2, r2c1, r2c2; bool has_r1 = false; };
struct DptW {
    LNw norm;
    Lin proj[4];
    Lin rs0, rs1;       // ConvTranspose k=s=4 / 2 as pixel-shuffle GEMMs (bias tiled s*s times)
    ConvW rs3;
    ConvW rn[4];
    FusionW ref[4];     // ref[0] = refinenet1 ... ref[3] = refinenet4
    ConvW oc1, oc2a;
    float *oc2b_w = nullptr, *oc2b_b = nullptr;
    unsigned short* oc2a_direct = nullptr;   // conv_direct.hip weight layout of oc2a (fp32-accurate heads)
    int n_out = 0, features = 0, oc[4] = {0, 0, 0, 0};
    bool feature_only = false, pos_embed = true;
    bool out_f32 = false;   // feature_only heads: the returned feature map is fp32 whatever the activation format (the tracker reads fp32)
    const LNw* out_ln = nullptr;   // feature_only heads with 128 features: LayerNorm applied by the last upsample (the tracker's fmap_norm)
    int down_ratio = 1;
};
struct CamW {
    std::vector<BlockW> trunk;
    LNw token_norm, trunk_norm;
    float* empty16 = nullptr;   // empty_pose_tokens padded to 16
    Lin embed_pose, poseLN, fc1, fc2;
};

struct Arena {
    char* base = nullptr;
    size_t off = 0, peak = 0, cap = 0;
    bool dry = false, overflow = false;
    void* alloc(size_t bytes) {
        off = align_up(off, 256);
        void* p = dry ? (void*)(uintptr_t)(0x1000 + off) : (void*)(base + off);
        off += bytes;
        if (off > peak) peak = off;
        if (!dry && off > cap) overflow = true;
        return p;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

struct UvTab { int w, h, C; float* tx; float* ty; };

// Per-shape constant tables (RoPE positions and cos/sin tables, the DPT heads' UV embeddings): immutable
// entries, built on first use and kept until the handle is destroyed, so that forwards of ANY shapes may be in
// flight on one handle at the same time.  Keyed by the frame size (H, W): the RoPE and UV tables do not depend on
// the number of frames at all, and the position table of F frames is a prefix of the table of more frames -- an
// entry serves every call of up to `frames` frames; a call with more builds a new entry of at least twice the
// capacity that SHARES the first entry's RoPE / UV tables, so a caller that varies its batch size holds at most
// 2 x the largest position table it ever asked for (the cache used to grow with every distinct F).
struct ShapeTabs {
    int frames = 0;                 // capacity of pos in frames
    int* pos = nullptr;             // int32 [frames*P, 2]
    float *rope_cos = nullptr, *rope_sin = nullptr;
    int rope_npos = 0;
    const float* dino_pos = nullptr;   // DINOv2 pos_embed of this frame size (the model's own or a registered resize)
    std::vector<UvTab> uv;
    std::vector<void*> owned;
    ~ShapeTabs() {
        for (void* p : owned) (void)hipFree(p);
    }
};

#define TRACK_PART 1
#include "track_impl.inc"
#undef TRACK_PART

}  // namespace

struct skimi_vggt {
    skimi_vggt_config cfg;
    std::map<std::string, std::pair<float*, int64_t>> raw;   // staged fp32 weights (device)
    std::vector<void*> owned;                                // packed device buffers
    bool finalized = false;
    // packed model
    Lin patch_proj;                 // [C, Kp] patchify GEMM (K = 3*p*p padded to 8)
    int patch_kp = 0;
    float* dino_pos = nullptr;      // pos_embed [1 + np0, C]
    std::map<std::pair<int, int>, float*> dino_pos_alt;   // (H, W) -> resized pos_embed [1 + ph*pw, C]
    float* dino_special = nullptr;  // [2][1+R][C] (cls + pos[0], registers), both selector rows equal
    float* agg_special = nullptr;   // [2][1+R][C] camera/register tokens for frame 0 / others
    std::vector<BlockW> dino, frame, global;
    LNw dino_norm;
    CamW cam;
    DptW depth, point, trackf;
    TrackW track;
    // per-shape tables: keyed cache, entries immutable once published (prepare())
    std::mutex prep_mu;             // guards the map; the entries themselves are read-only
    std::map<std::pair<int, int>, std::vector<std::unique_ptr<ShapeTabs>>> tabs;   // (H, W) -> entries of growing frame capacity
};

namespace {

// ------------------------------------------------------------------------------------------
// finalize helpers
// ------------------------------------------------------------------------------------------
struct Packer {
    skimi_vggt* h;
    int rc = SKIMI_OK;
    hipStream_t st = nullptr;

    float* raw(const std::string& key, int64_t n) {
        if (rc) return nullptr;
        auto it = h->raw.find(key);
        if (it == h->raw.end()) {
            set_error("skimi_vggt_finalize: missing weight '%s'", key.c_str());
            rc = SKIMI_ERR_STATE;
            return nullptr;
        }
        if (it->second.second != n) {
            set_error("skimi_vggt_finalize: weight '%s' has %ld elements, expected %ld", key.c_str(),
                      (long)it->second.second, (long)n);
            rc = SKIMI_ERR_STATE;
            return nullptr;
        }
        return it->second.first;
    }
    void* dmalloc(size_t bytes) {
        if (rc) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            set_error("skimi_vggt_finalize: hipMalloc(%zu) failed", bytes);
            rc = SKIMI_ERR_HIP;
            return nullptr;
        }
        h->owned.push_back(p);
        return p;
    }
    // a plain fp32 parameter used as is (LN affine, bias, LayerScale, tokens): private copy
    float* keep(const std::string& key, int64_t n) {
        float* src = raw(key, n);
        if (!src) return nullptr;
        float* dst = (float*)dmalloc(n * 4);
        if (dst && hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = SKIMI_ERR_HIP;
        return dst;
    }
    LNw ln(const std::string& p, int C) { return LNw{keep(p + ".weight", C), keep(p + ".bias", C)}; }

    // matrix [N, K] fp32 on device (owned by caller) -> Lin in the requested precision, K padded to 8
    Lin pack_matrix(float* src, int N, int K, int prec) {
        Lin L;
        L.N = N;
        L.K = (int)align_up((size_t)K, 8);
        L.prec = prec;
        if (rc) return L;
        float* m = src;
        if (L.K != K) {
            m = (float*)dmalloc((size_t)N * L.K * 4);
            if (m && (rc = pad_cols_launch(src, m, N, K, L.K, st))) return L;
        }
        if (prec == SKIMI_PREC_BF16 || prec == SKIMI_PREC_F16) {
            void* b = dmalloc((size_t)N * L.K * 2);
            if (b) rc = prec == SKIMI_PREC_F16 ? f32_to_f16_launch(m, b, (long)N * L.K, st) : f32_to_bf16_launch(m, b, (long)N * L.K, st);
            L.w = b;
            L.wdt = prec == SKIMI_PREC_F16 ? SKIMI_F16 : SKIMI_BF16;
        } else {
            if (m == src) {   // private copy: the staged buffer is released after finalize
                m = (float*)dmalloc((size_t)N * L.K * 4);
                if (m && hipMemcpyAsync(m, src, (size_t)N * L.K * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    rc = SKIMI_ERR_HIP;
            }
            L.w = m;
            L.wdt = SKIMI_F32;
        }
        return L;
    }
    Lin linear(const std::string& p, int N, int K, int prec, bool bias = true) {
        Lin L = pack_matrix(raw(p + ".weight", (int64_t)N * K), N, K, prec);
        if (bias) L.b = keep(p + ".bias", N);
        return L;
    }
    // nn.MultiheadAttention style names
    Lin linear_named(const std::string& wkey, const std::string& bkey, int N, int K, int prec) {
        Lin L = pack_matrix(raw(wkey, (int64_t)N * K), N, K, prec);
        L.b = keep(bkey, N);
        return L;
    }
    // bf16 [hi 32 | lo 32] records of a packed fp32 matrix, for the LDS-DMA bf16x3 kernel (which
    // gemm_x3dma_eligible then picks for launches that fill the chip with 256-row tiles)
    void add_records(Lin& L) {
        const size_t n = (size_t)L.N * ((L.K + 31) / 32 * 32);
        L.w_split = dmalloc(n * 4);
        if (L.w_split) rc = split_records_launch((const float*)L.w, L.K, L.N, L.K, L.w_split, st);
    }
    ConvW conv(const std::string& p, int Co, int Ci, int k, int stride, int pad, int prec, bool bias) {
        ConvW c;
        c.k = k; c.stride = stride; c.pad = pad; c.cin = Ci;
        float* src = raw(p + ".weight", (int64_t)Co * Ci * k * k);
        if (rc) return c;
        float* perm = src;
        float* tmp = nullptr;
        if (k > 1) {
            // fp32-accurate convs run on K-tiles of 32: order K slice-major there (see permute_conv_kernel)
            static const int force_tap_major = getenv("SKIMI_CONV_TAP_MAJOR") ? atoi(getenv("SKIMI_CONV_TAP_MAJOR")) : 0;   // A/B timing
            c.kmode = (prec == SKIMI_PREC_BF16X3 && Ci % 32 == 0 && !force_tap_major) ? 2 : 1;
            if (hipMalloc((void**)&tmp, (size_t)Co * Ci * k * k * 4) != hipSuccess) { rc = SKIMI_ERR_HIP; return c; }
            if ((rc = permute_conv_launch(src, tmp, Co, Ci, k, k, st, c.kmode == 2))) return c;
            perm = tmp;
        }
        c.lin = pack_matrix(perm, Co, Ci * k * k, prec);
        // fp32-accurate 3x3 convs, wide 1x1 projections and the ConvTranspose matrices also get bf16
        // hi|lo records: the LDS-DMA bf16x3 kernel (gemm_x3dma.hip) beats the generic one wherever
        // 256-row tiles fill the chip
        if (!rc && prec == SKIMI_PREC_BF16X3 && Ci % 32 == 0 && ((k == 3 && Co >= 96) || (k == 1 && Co >= 512)))
            add_records(c.lin);
        if (tmp) { (void)hipStreamSynchronize(st); (void)hipFree(tmp); }
        if (bias) c.lin.b = keep(p + ".bias", Co);
        return c;
    }
    // ConvTranspose2d(C, C, k = s, stride = s): [(a,b,co), Ci] weight, bias tiled s*s times
    Lin convT(const std::string& p, int C, int s, int prec) {
        Lin L;
        float* src = raw(p + ".weight", (int64_t)C * C * s * s);
        float* bsrc = raw(p + ".bias", C);
        if (rc) return L;
        float* tmp = nullptr;
        if (hipMalloc((void**)&tmp, (size_t)C * C * s * s * 4) != hipSuccess) { rc = SKIMI_ERR_HIP; return L; }
        if ((rc = permute_convT_launch(src, tmp, C, C, s, st))) return L;
        L = pack_matrix(tmp, s * s * C, C, prec);
        if (!rc && prec == SKIMI_PREC_BF16X3 && C % 32 == 0) add_records(L);
        (void)hipStreamSynchronize(st);
        (void)hipFree(tmp);
        L.b = (float*)dmalloc((size_t)s * s * C * 4);
        if (L.b) rc = rc ? rc : tile_vec_launch(bsrc, L.b, C, s * s, st);
        return L;
    }
    // MXFP8 copy of a packed Linear (from the staged fp32 weights: one rounding)
    void add_fp8(Lin& L, const std::string& p, int N, int K) {
        if (rc) return;
        const size_t Kp = align_up((size_t)K, 128);
        L.wq = dmalloc((size_t)N * Kp);
        L.wq_scales = dmalloc((size_t)N * (Kp / 32));
        if (L.wq && L.wq_scales) rc = quant_mx_launch(raw(p + ".weight", (int64_t)N * K), SKIMI_F32, K, N, K, L.wq, L.wq_scales, st);
    }
    BlockW block(const std::string& p, int C, int hidden, bool qk_norm, int hd, int prec_in) {
        BlockW b;
        const bool fp8 = prec_in == SKIMI_PREC_FP8 && C % 32 == 0 && hidden % 32 == 0;
        const int prec = prec_in == SKIMI_PREC_FP8 ? SKIMI_PREC_BF16 : prec_in;
        b.n1 = ln(p + ".norm1", C);
        b.n2 = ln(p + ".norm2", C);
        b.qkv = linear(p + ".attn.qkv", 3 * C, C, prec);
        b.proj = linear(p + ".attn.proj", C, C, prec);
        b.fc1 = linear(p + ".mlp.fc1", hidden, C, prec);
        b.fc2 = linear(p + ".mlp.fc2", C, hidden, prec);
        // fp32-accurate mode: the four Linears also as bf16x3 records, so that launches that fill the chip with 256-row
        // tiles (from one 8-view time step on) run on the LDS-DMA bf16x3 kernel (gemm_x3dma.hip) instead of the generic one
        if (!rc && prec == SKIMI_PREC_BF16X3 && C % 32 == 0 && hidden % 32 == 0) {
            add_records(b.qkv);
            add_records(b.proj);
            add_records(b.fc1);
            add_records(b.fc2);
        }
        if (fp8) {
            add_fp8(b.qkv, p + ".attn.qkv", 3 * C, C);
            // SKIMI_FP8_PROJ=0 (read when the weights are packed): proj stays on the bf16 kernel, as in round 2 (A/B timing)
            if (!(getenv("SKIMI_FP8_PROJ") && atoi(getenv("SKIMI_FP8_PROJ")) == 0)) add_fp8(b.proj, p + ".attn.proj", C, C);
            add_fp8(b.fc1, p + ".mlp.fc1", hidden, C);
            add_fp8(b.fc2, p + ".mlp.fc2", C, hidden);
        }
        b.ls1 = keep(p + ".ls1.gamma", C);
        b.ls2 = keep(p + ".ls2.gamma", C);
        if (qk_norm) {
            b.qn_w = keep(p + ".attn.q_norm.weight", hd);
            b.qn_b = keep(p + ".attn.q_norm.bias", hd);
            b.kn_w = keep(p + ".attn.k_norm.weight", hd);
            b.kn_b = keep(p + ".attn.k_norm.bias", hd);
        }
        return b;
    }
    DptW dpt(const std::string& p, int D, int features, const int* oc, int n_out, bool feature_only, int prec) {
        DptW d;
        d.features = features;
        d.n_out = n_out;
        d.feature_only = feature_only;
        for (int i = 0; i < 4; ++i) d.oc[i] = oc[i];
        d.norm = ln(p + ".norm", D);
        for (int i = 0; i < 4; ++i) {
            d.proj[i] = linear(p + ".projects." + std::to_string(i), oc[i], D, prec);
            if (!rc && prec == SKIMI_PREC_BF16X3 && oc[i] >= 512 && D % 32 == 0) add_records(d.proj[i]);
        }
        d.rs0 = convT(p + ".resize_layers.0", oc[0], 4, prec);
        d.rs1 = convT(p + ".resize_layers.1", oc[1], 2, prec);
        d.rs3 = conv(p + ".resize_layers.3", oc[3], oc[3], 3, 2, 1, prec, true);
        for (int i = 0; i < 4; ++i)
            d.rn[i] = conv(p + ".scratch.layer" + std::to_string(i + 1) + "_rn", features, oc[i], 3, 1, 1, prec, false);
        for (int r = 0; r < 4; ++r) {
            const std::string rp = p + ".scratch.refinenet" + std::to_string(r + 1);
            FusionW& f = d.ref[r];
            f.out_conv = linear(rp + ".out_conv", features, features, prec);
            if (!rc && prec == SKIMI_PREC_BF16X3 && features % 32 == 0) add_records(f.out_conv);
            f.has_r1 = r != 3;
            if (f.has_r1) {
                f.r1c1 = conv(rp + ".resConfUnit1.conv1", features, features, 3, 1, 1, prec, true);
                f.r1c2 = conv(rp + ".resConfUnit1.conv2", features, features, 3, 1, 1, prec, true);
            }
            f.r2c1 = conv(rp + ".resConfUnit2.conv1", features, features, 3, 1, 1, prec, true);
            f.r2c2 = conv(rp + ".resConfUnit2.conv2", features, features, 3, 1, 1, prec, true);
        }
        if (feature_only) {
            d.oc1 = conv(p + ".scratch.output_conv1", features, features, 3, 1, 1, prec, true);
        } else {
            d.oc1 = conv(p + ".scratch.output_conv1", features / 2, features, 3, 1, 1, prec, true);
            d.oc2a = conv(p + ".scratch.output_conv2.0", 32, features / 2, 3, 1, 1, prec, true);
            if (!rc && prec == SKIMI_PREC_BF16X3 && (features / 2) % 32 == 0) {
                // the full-resolution 3x3 -> 32 conv runs as a direct convolution (conv_direct.hip)
                const int Ci = features / 2;
                float* src = raw(p + ".scratch.output_conv2.0.weight", (int64_t)32 * Ci * 9);
                d.oc2a_direct = (unsigned short*)dmalloc((size_t)32 * Ci * 9 * 2 * 2);
                if (!rc && d.oc2a_direct) rc = conv_direct_pack_launch(src, d.oc2a_direct, Ci, st);
                if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = SKIMI_ERR_HIP;
            }
            d.oc2b_w = keep(p + ".scratch.output_conv2.2.weight", (int64_t)n_out * 32);
            d.oc2b_b = keep(p + ".scratch.output_conv2.2.bias", n_out);
        }
        return d;
    }
};

// ------------------------------------------------------------------------------------------
// forward context
// ------------------------------------------------------------------------------------------
struct Ctx {
    skimi_vggt* h;
    hipStream_t st;
    const ShapeTabs* tabs = nullptr;   // this call's shape tables (null in a dry run)
    Arena ar;
    int rc = SKIMI_OK;
    void* slab = nullptr;      // split-K scratch
    size_t slab_bytes = 0;

    bool dry() const { return ar.dry; }
    // operand format of the Linears' activations: fp32 (split hi + lo by the kernels), fp16, or bf16 (BF16 and FP8 modes)
    static int act_dt(int prec) { return prec == SKIMI_PREC_BF16X3 ? SKIMI_F32 : prec == SKIMI_PREC_F16 ? SKIMI_F16 : SKIMI_BF16; }
    static size_t esz(int dt) { return dt == SKIMI_F32 ? 4 : 2; }

    void gemm(skimi_gemm_desc& d) {
        // wide fp32-accurate 3x3 convs: lend arena scratch for the activation planes of the
        // LDS-DMA bf16x3 kernel (released right after the launch is enqueued: stream order keeps
        // later users of that memory behind it)
        const size_t mk = ar.mark();
        if (d.W_split != nullptr && d.a_dtype != SKIMI_BF16X3_REC) {
            d.x3_scratch_bytes = gemm_x3dma_scratch_bytes(&d);
            d.x3_scratch = ar.alloc(d.x3_scratch_bytes);
        }
        if (!rc && !dry()) rc = gemm_dispatch(&d, st, slab, slab_bytes, 0);
        ar.release(mk);
    }
    skimi_gemm_desc desc(const Lin& L, const void* A, int a_dt, long lda, int M, void* out, int out_dt, long ldo) {
        skimi_gemm_desc d;
        memset(&d, 0, sizeof d);
        d.M = M; d.N = L.N; d.K = L.K;
        d.A = A; d.a_dtype = a_dt; d.lda = lda;
        d.W = L.w; d.w_dtype = L.wdt; d.ldw = L.K;
        d.prec = L.prec;
        d.splitk_scratch_zeroed = 1;
        d.W_split = L.w_split;   // gemm_x3dma_eligible decides (shape, tile count)
        d.bias = L.b;
        d.out = out; d.out_dtype = out_dt; d.ldo = ldo;
        return d;
    }
    void conv_geom(skimi_gemm_desc& d, const ConvW& c, int N, int H, int W, int OH, int OW) {
        if (c.k == 1 && c.stride == 1) return;   // plain rows
        d.a_mode = c.kmode;
        d.cN = N; d.cH = H; d.cW = W; d.cC = c.cin; d.KH = c.k; d.KW = c.k;
        d.stride = c.stride; d.pad = c.pad; d.dil = 1; d.OH = OH; d.OW = OW;
    }
    void ln(const float* x, const float* x2, long ldx, long rows, int C, const LNw& w, float eps, void* out, int odt,
            long grp_rows = 0, long grp_stride = 0, long grp_off = 0) {
        if (rc || dry()) return;
        rc = layernorm_launch(x, x2, ldx, rows, C, w.g, w.b, eps, out, odt, C, st, grp_rows, grp_stride, grp_off);
    }
    void copy(void* dst, const void* src, size_t bytes) {
        if (rc || dry()) return;
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("hipMemcpyAsync failed");
            rc = SKIMI_ERR_HIP;
        }
    }
};

// one pre-LN transformer block on the fp32 residual stream x [batch*seq, C]
// (vggt/vggt/layers/block.py:77-98 + attention.py:50-72), scratch buffers supplied by the caller
struct BlockBufs { void* xn; void* qkv; void* ao; void* hid; void* q8 = nullptr; void* q8s = nullptr; };   // q8: MXFP8 scratch [M][4C] + scales

void run_block(Ctx& c, const BlockW& w, float* x, int batch, int seq, int C, int heads, float eps, bool rope,
               const BlockBufs& b) {
    const int M = batch * seq;
    const int prec = w.qkv.prec;
    const int adt = Ctx::act_dt(prec);
    // the packed q / k / v buffer: SKIMI_PREC_F16 keeps the attention products on bf16 operands (they average their rounding
    // noise over all keys: profiles/r03_precision_ablation.md), so qkv's epilogue writes bf16 and the attention kernel
    // converts its output to fp16 for proj
    const int qdt = adt == SKIMI_F16 ? SKIMI_BF16 : adt;
    const int hidden = w.fc1.N;
    const bool fp8 = w.qkv.wq != nullptr && b.q8 != nullptr;   // SKIMI_PREC_FP8: qkv, proj, fc1, fc2 on the MXFP8 MFMA
    // the quantisation rides in the producers where the shapes allow: LayerNorm writes MXFP8 directly (C % 256 == 0),
    // the attention kernel writes its output rows as MXFP8 (64-wide heads), and fc1's GELU epilogue writes the hidden
    // activation as MXFP8 (large launches on the single-stream loop)
    const bool ln_mx = fp8 && C % 256 == 0;
    const bool hid_mx = fp8 && gemm_fp8_mx_output_ok(M, hidden);
    const bool ao_mx = fp8 && w.proj.wq != nullptr && attention_mx_output_ok(qdt, heads, C / heads);
    // fp32-accurate mode: a Linear whose launch takes the LDS-DMA bf16x3 kernel reads its A operand as [hi 32 | lo 32]
    // records; where the producer can write them (LayerNorm, fc1's GELU epilogue) the fp32 copy and the split pass go
    auto as_records = [&](skimi_gemm_desc d, void* rec, size_t rec_bytes) {
        d.A = rec; d.a_dtype = SKIMI_BF16X3_REC;
        d.x3_scratch = (char*)rec + rec_bytes; d.x3_scratch_bytes = 256;   // the zero page behind the records
        return d;
    };
    const bool x3 = adt == SKIMI_F32 && C % 256 == 0;
    auto d_qkv = c.desc(w.qkv, b.xn, adt, C, M, b.qkv, qdt, 3 * C);
    bool rec_qkv = false;
    if (x3) {
        const auto q = as_records(d_qkv, b.xn, (size_t)M * C * 4);
        if ((rec_qkv = gemm_x3dma_eligible(&q))) d_qkv = q;
    }
    if (ln_mx) {
        if (!c.rc && !c.dry()) c.rc = layernorm_mx_launch(x, C, M, C, w.n1.g, w.n1.b, eps, b.q8, b.q8s, c.st);
    } else {
        c.ln(x, nullptr, C, M, C, w.n1, eps, b.xn, rec_qkv ? SKIMI_BF16X3_REC : adt);
    }
    if (fp8) {
        if (!ln_mx && !c.rc && !c.dry()) c.rc = quant_mx_launch(b.xn, SKIMI_BF16, C, M, C, b.q8, b.q8s, c.st);
        if (!c.rc && !c.dry())
            c.rc = gemm_fp8_launch(b.q8, b.q8s, w.qkv.wq, w.qkv.wq_scales, M, 3 * C, C, w.qkv.b, SKIMI_ACT_NONE, nullptr, nullptr, 0,
                                   b.qkv, SKIMI_BF16, 3 * C, c.st);
    } else {
        c.gemm(d_qkv);
    }
    int q_scaled = 0;
    if (!c.rc && !c.dry() && (w.qn_w || rope)) {
        // bf16 mode: the softmax scale (x log2 e) rides in q's one rounding to bf16, the attention kernel then
        // exponentiates the raw accumulator
        const float q_scale = (1.0f / sqrtf((float)(C / heads))) * 1.44269504088896340736f;
        c.rc = qknorm_rope_launch(b.qkv, qdt, M, heads, w.qn_w, w.qn_b, w.kn_w, w.kn_b, 1e-5f, rope ? c.tabs->pos : nullptr,
                                  c.tabs->rope_cos, c.tabs->rope_sin, c.tabs->rope_npos, c.st, q_scale,
                                  qdt == SKIMI_BF16 && C / heads == 64 ? &q_scaled : nullptr);
    }
    // fp32-accurate mode: the MLP's hidden buffer (M x hidden fp32, idle until fc1) lends the hi / lo planes of k and v; the
    // attention kernel writes proj's operand records when proj runs on the LDS-DMA kernel
    auto d_proj = c.desc(w.proj, b.ao, adt, C, M, x, SKIMI_F32, C);
    d_proj.gamma = w.ls1; d_proj.resid = x; d_proj.ldr = C;
    int ao_rec = 0;
    if (x3 && C / heads == 64) {
        const auto q = as_records(d_proj, b.ao, (size_t)M * C * 4);
        ao_rec = gemm_x3dma_eligible(&q) ? 1 : 0;
    }
    if (!c.rc && !c.dry())
        c.rc = attention_launch(b.qkv, b.ao, qdt, batch, seq, heads, C / heads, c.st, q_scaled, adt == SKIMI_F32 ? b.hid : nullptr,
                                adt == SKIMI_F32 ? (size_t)M * hidden * 4 : 0, &ao_rec, adt == SKIMI_F16,
                                ao_mx ? b.q8 : nullptr, ao_mx ? b.q8s : nullptr);
    if (ao_rec && !c.dry()) d_proj = as_records(d_proj, b.ao, (size_t)M * C * 4);
    if (fp8 && w.proj.wq) {
        if (!ao_mx && !c.rc && !c.dry()) c.rc = quant_mx_launch(b.ao, SKIMI_BF16, C, M, C, b.q8, b.q8s, c.st);
        if (!c.rc && !c.dry())
            c.rc = gemm_fp8_launch(b.q8, b.q8s, w.proj.wq, w.proj.wq_scales, M, C, C, w.proj.b, SKIMI_ACT_NONE, w.ls1, x, C, x,
                                   SKIMI_F32, C, c.st);
    } else {
        c.gemm(d_proj);
    }
    auto d_fc1 = c.desc(w.fc1, b.xn, adt, C, M, b.hid, adt, hidden);
    d_fc1.act = SKIMI_ACT_GELU;
    auto d_fc2 = c.desc(w.fc2, b.hid, adt, hidden, M, x, SKIMI_F32, C);
    d_fc2.gamma = w.ls2; d_fc2.resid = x; d_fc2.ldr = C;
    bool rec_fc1 = false;
    if (x3) {
        const auto q1 = as_records(d_fc1, b.xn, (size_t)M * C * 4);
        if ((rec_fc1 = gemm_x3dma_eligible(&q1))) d_fc1 = q1;
        const auto q2 = as_records(d_fc2, b.hid, (size_t)M * hidden * 4);
        if (hidden % 32 == 0 && gemm_x3dma_eligible(&q2)) {   // fc1 writes the hidden activation as fc2's records, and only so
            d_fc2 = q2;
            d_fc1.out = nullptr;
            d_fc1.out_records = b.hid;
        }
    }
    if (ln_mx) {
        if (!c.rc && !c.dry()) c.rc = layernorm_mx_launch(x, C, M, C, w.n2.g, w.n2.b, eps, b.q8, b.q8s, c.st);
    } else {
        c.ln(x, nullptr, C, M, C, w.n2, eps, b.xn, rec_fc1 ? SKIMI_BF16X3_REC : adt);
    }
    if (fp8) {
        if (!ln_mx && !c.rc && !c.dry()) c.rc = quant_mx_launch(b.xn, SKIMI_BF16, C, M, C, b.q8, b.q8s, c.st);
        // b.hid (M x hidden bf16 = 2 bytes per element) holds the MXFP8 hidden activation: payload M x hidden bytes,
        // then its scales
        unsigned char* hq = (unsigned char*)b.hid;
        unsigned char* hs = hq + (size_t)M * hidden;
        if (hid_mx) {
            if (!c.rc && !c.dry())
                c.rc = gemm_fp8_launch(b.q8, b.q8s, w.fc1.wq, w.fc1.wq_scales, M, hidden, C, w.fc1.b, SKIMI_ACT_GELU, nullptr, nullptr, 0,
                                       hq, SKIMI_FP8MX, hidden, c.st, hs);
        } else {
            if (!c.rc && !c.dry())
                c.rc = gemm_fp8_launch(b.q8, b.q8s, w.fc1.wq, w.fc1.wq_scales, M, hidden, C, w.fc1.b, SKIMI_ACT_GELU, nullptr, nullptr, 0,
                                       b.hid, SKIMI_BF16, hidden, c.st);
            if (!c.rc && !c.dry()) c.rc = quant_mx_launch(b.hid, SKIMI_BF16, hidden, M, hidden, b.q8, b.q8s, c.st);
            hq = (unsigned char*)b.q8;
            hs = (unsigned char*)b.q8s;
        }
        if (!c.rc && !c.dry())
            c.rc = gemm_fp8_launch(hq, hs, w.fc2.wq, w.fc2.wq_scales, M, C, hidden, w.fc2.b, SKIMI_ACT_NONE, w.ls2, x, C, x,
                                   SKIMI_F32, C, c.st);
        return;
    }
    c.gemm(d_fc1);
    c.gemm(d_fc2);
}

const UvTab* find_uv(const ShapeTabs* h, int w, int hh, int C) {
    if (!h) return nullptr;
    for (const auto& t : h->uv)
        if (t.w == w && t.h == hh && t.C == C) return &t;
    return nullptr;
}

// whether a feature_only head's last upsample also applies the consumer's LayerNorm (SKIMI_TRACK_LN_FUSED=0: separate pass)
bool dpt_fused_ln(const DptW& w, int f2, int udt) {
    static const bool off = getenv("SKIMI_TRACK_LN_FUSED") && !atoi(getenv("SKIMI_TRACK_LN_FUSED"));
    return !off && w.feature_only && w.out_ln != nullptr && w.out_ln->g != nullptr && w.out_ln->b != nullptr && f2 == 128 && udt == SKIMI_F32;
}

// DPT head (vggt/vggt/heads/dpt_head.py:172-291) on the kept intermediates.
// Returns the NHWC feature map pointer for feature_only heads; otherwise writes pts / conf.
void* run_dpt(Ctx& c, const DptW& w, float* const* sf, float* const* sg, int F, int P, int nsp, int ph, int pw, int C,
              int H, int W, float* pts, float* conf, int act_mode) {
    const int prec = w.proj[0].prec;
    const int adt = Ctx::act_dt(prec);
    const size_t es = Ctx::esz(adt);
    const int D = 2 * C, np = ph * pw, feat = w.features;
    const size_t mk = c.ar.mark();
    // resized pyramid sizes
    int hh[4] = {ph * 4, ph * 2, ph, (ph - 1) / 2 + 1};
    int ww[4] = {pw * 4, pw * 2, pw, (pw - 1) / 2 + 1};
    void* rn[4];
    for (int i = 0; i < 4; ++i) rn[i] = c.ar.alloc((size_t)F * hh[i] * ww[i] * feat * es);
    // Levels whose feat -> feat convs run on the LDS-DMA bf16x3 kernel hand their activations over as
    // operand records ([feat/32][hi 32 | lo 32] per pixel, + zero page) written by the producing
    // epilogue: no split pass, and no fp32 copy at all where the only reader is the next conv.
    auto rec_bytes = [&](int lvl) { return (size_t)F * hh[lvl] * ww[lvl] * feat * 4 + 256; };
    auto takes_records = [&](const Lin& L, const ConvW* cw, int lvl) {
        if (adt != SKIMI_F32 || feat % 32 != 0 || L.w_split == nullptr) return false;
        char* fake = (char*)(uintptr_t)0x10000000;   // eligibility looks at alignment and distances only
        auto d = c.desc(L, fake, SKIMI_BF16X3_REC, feat, F * hh[lvl] * ww[lvl], fake, adt, L.N);
        if (cw) c.conv_geom(d, *cw, F, hh[lvl], ww[lvl], hh[lvl], ww[lvl]);
        d.x3_scratch = fake + rec_bytes(lvl) - 256;
        d.x3_scratch_bytes = 256;
        return gemm_x3dma_eligible(&d);
    };
    bool use_rec[4];
    char* rn_rec[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
        use_rec[i] = takes_records(w.ref[i].r2c1.lin, &w.ref[i].r2c1, i);
        if (use_rec[i]) rn_rec[i] = (char*)c.ar.alloc(rec_bytes(i));
    }
    {
        const size_t mk2 = c.ar.mark();
        void* lnb = c.ar.alloc((size_t)F * np * D * es);
        void* lnb_rec = adt == SKIMI_F32 ? c.ar.alloc((size_t)F * np * D * 4 + 256) : nullptr;   // LayerNorm output as bf16x3 records
        for (int i = 0; i < 4; ++i) {
            const size_t mk3 = c.ar.mark();
            void* t0 = c.ar.alloc((size_t)F * np * w.oc[i] * es);
            {
                // wide projections run on the LDS-DMA bf16x3 kernel: the LayerNorm writes their operand records
                auto d = c.desc(w.proj[i], lnb, adt, D, F * np, t0, adt, w.oc[i]);
                bool rec_in = false;
                if (adt == SKIMI_F32 && D % 256 == 0 && w.proj[i].w_split != nullptr) {
                    auto q = d;
                    q.a_dtype = SKIMI_BF16X3_REC;
                    q.x3_scratch = (char*)lnb_rec + (size_t)F * np * D * 4;
                    q.x3_scratch_bytes = 256;
                    q.A = lnb_rec;
                    rec_in = gemm_x3dma_eligible(&q);
                    if (rec_in) d = q;
                }
                c.ln(sf[i], sg[i], C, (long)F * np, D, w.norm, 1e-5f, rec_in ? lnb_rec : lnb, rec_in ? SKIMI_BF16X3_REC : adt, np, P, nsp);
                c.gemm(d);
            }
            if (w.pos_embed && !c.rc && !c.dry()) {
                const UvTab* t = find_uv(c.tabs, pw, ph, w.oc[i]);
                if (!t) { set_error("uv table missing"); c.rc = SKIMI_ERR_STATE; }
                else c.rc = add_uv_pos_launch(t0, adt, t->tx, t->ty, F, ph, pw, w.oc[i], c.st);
            }
            void* t1 = t0;
            if (i == 0 || i == 1) {
                const int s = i == 0 ? 4 : 2;
                const Lin& L = i == 0 ? w.rs0 : w.rs1;
                t1 = c.ar.alloc((size_t)F * hh[i] * ww[i] * w.oc[i] * es);
                auto d = c.desc(L, t0, adt, w.oc[i], F * np, t1, adt, w.oc[i]);
                d.store_mode = 1; d.ps_s = s; d.ps_C = w.oc[i];
                d.cN = F; d.cH = ph; d.cW = pw;
                c.gemm(d);
            } else if (i == 3) {
                t1 = c.ar.alloc((size_t)F * hh[i] * ww[i] * w.oc[i] * es);
                auto d = c.desc(w.rs3.lin, t0, adt, w.oc[i], F * hh[i] * ww[i], t1, adt, w.oc[i]);
                c.conv_geom(d, w.rs3, F, ph, pw, hh[i], ww[i]);
                c.gemm(d);
            }
            // layerN_rn: 3x3, no bias; its only consumer is a ResidualConvUnit whose in-place ReLU
            // rewrites it (dpt_head.py:376), so the ReLU is applied here once
            {
                auto d = c.desc(w.rn[i].lin, t1, adt, w.oc[i], F * hh[i] * ww[i], rn[i], adt, feat);
                c.conv_geom(d, w.rn[i], F, hh[i], ww[i], hh[i], ww[i]);
                d.act = SKIMI_ACT_RELU;
                d.out_records = rn_rec[i];
                c.gemm(d);
            }
            c.ar.release(mk3);
        }
        c.ar.release(mk2);
    }
    // RefineNet fusion 4 -> 1 (dpt_head.py:261-291, 427-456)
    void* prev = nullptr;   // previous refinenet output at this level's resolution
    char* prev_rec = nullptr;   // ... or, after the last level, the same map as bf16x3 operand records
    for (int r = 3; r >= 0; --r) {
        const FusionW& f = w.ref[r];
        const int h0 = hh[r], w0 = ww[r];
        const int M = F * h0 * w0;
        const size_t bytes = (size_t)M * feat * es;
        const bool recs = use_rec[r];
        // a feat -> feat conv of this level: input as fp32 rows or as records
        auto conv_in = [&](const ConvW& cw, const void* in_f32, char* in_rec, void* out_f32, char* out_rec) {
            auto d = c.desc(cw.lin, recs ? (const void*)in_rec : in_f32, recs ? SKIMI_BF16X3_REC : adt, feat, M, out_f32, adt, feat);
            c.conv_geom(d, cw, F, h0, w0, h0, w0);
            if (recs) {
                d.x3_scratch = in_rec + rec_bytes(r) - 256;
                d.x3_scratch_bytes = 256;
            }
            d.out_records = out_rec;
            return d;
        };
        void* cur;                  // relu(input of resConfUnit2)
        char* cur_rec = nullptr;
        void* tmp = recs ? nullptr : c.ar.alloc(bytes);
        char* tmp_rec = recs ? (char*)c.ar.alloc(rec_bytes(r)) : nullptr;
        if (f.has_r1) {
            // res = RCU1(layer_rn): conv2(relu(conv1(relu(x)))) + relu(x); output = prev + res;
            // RCU2's in-place ReLU then rewrites output -> store relu(prev + res) directly
            cur = c.ar.alloc(bytes);
            if (recs) cur_rec = (char*)c.ar.alloc(rec_bytes(r));
            auto d1 = conv_in(f.r1c1, rn[r], rn_rec[r], tmp, tmp_rec);
            d1.act = SKIMI_ACT_RELU;
            c.gemm(d1);
            auto d2 = conv_in(f.r1c2, tmp, tmp_rec, cur, cur_rec);
            d2.resid = rn[r]; d2.ldr = feat; d2.resid_dtype = adt;
            d2.resid2 = prev; d2.ldr2 = feat;
            d2.post_act = SKIMI_ACT_RELU;
            c.gemm(d2);
        } else {
            cur = rn[r];
            cur_rec = rn_rec[r];
        }
        // RCU2(cur): conv2(relu(conv1(cur))) + cur; its result is read by the 1x1 out_conv only
        const bool u_recs = recs && takes_records(f.out_conv, nullptr, r);
        void* u = u_recs ? nullptr : c.ar.alloc(bytes);
        char* u_rec = u_recs ? (char*)c.ar.alloc(rec_bytes(r)) : nullptr;
        {
            auto d1 = conv_in(f.r2c1, cur, cur_rec, tmp, tmp_rec);
            d1.act = SKIMI_ACT_RELU;
            c.gemm(d1);
            auto d2 = conv_in(f.r2c2, tmp, tmp_rec, u, u_rec);
            d2.resid = cur; d2.ldr = feat; d2.resid_dtype = adt;
            c.gemm(d2);
        }
        // dpt_head.py:447-455 upsamples (bilinear, align_corners) to the next level's size (x2 for
        // refinenet1) and then applies the 1x1 out_conv.  Both are linear maps over different axes
        // (space / channels) and the interpolation weights sum to 1, so they commute exactly, bias
        // included; the 1x1 conv runs here on the 4x smaller map, fp32 rounding order aside.
        const int h1 = r > 0 ? hh[r - 1] : 2 * h0, w1 = r > 0 ? ww[r - 1] : 2 * w0;
        void* olow = c.ar.alloc(bytes);
        {
            auto d = c.desc(f.out_conv, u_recs ? (const void*)u_rec : u, u_recs ? SKIMI_BF16X3_REC : adt, feat, M, olow, adt, feat);
            if (u_recs) {
                d.x3_scratch = u_rec + rec_bytes(r) - 256;
                d.x3_scratch_bytes = 256;
            }
            c.gemm(d);
        }
        // The last level's upsampled map is read by output_conv1 only: when that conv runs on the
        // LDS-DMA bf16x3 kernel the upsample writes its operand records ([C/32][hi 32 | lo 32] per
        // pixel + the zero page) instead of fp32 (no fp32 round trip, no split pass).
        if (r == 0 && adt == SKIMI_F32 && w.oc1.lin.w_split != nullptr && feat % 32 == 0) {
            const size_t rec_bytes = (size_t)F * h1 * w1 * feat * 4;
            const size_t mk_rec = c.ar.mark();
            char* rec = (char*)c.ar.alloc(rec_bytes + 256);
            auto d = c.desc(w.oc1.lin, rec, SKIMI_BF16X3_REC, feat, F * h1 * w1, nullptr, adt, w.feature_only ? feat : feat / 2);
            c.conv_geom(d, w.oc1, F, h1, w1, h1, w1);
            d.x3_scratch = rec + rec_bytes;
            d.x3_scratch_bytes = 256;
            d.out = rec;   // placeholder for the eligibility query only
            if (gemm_x3dma_eligible(&d)) {   // pointer-independent up to alignment: the sizing pass agrees
                prev_rec = rec;
                if (!c.rc && !c.dry())
                    c.rc = bilinear_ac_planes_launch((const float*)olow, (unsigned short*)rec, F, h0, w0, h1, w1, feat, c.st,
                                                     nullptr, nullptr, 1, rec + rec_bytes);
                prev = nullptr;
                hh[r] = h1; ww[r] = w1;
                continue;
            }
            c.ar.release(mk_rec);
        }
        void* o = c.ar.alloc((size_t)F * h1 * w1 * feat * es);
        if (!c.rc && !c.dry()) c.rc = bilinear_ac_launch(olow, o, adt, F, h0, w0, h1, w1, feat, c.st);
        prev = o;
        hh[r] = h1; ww[r] = w1;   // resolution of `prev`
    }
    const int h1 = hh[0], w1 = ww[0];
    const int f2 = w.feature_only ? feat : feat / 2;
    void* c1 = c.ar.alloc((size_t)F * h1 * w1 * f2 * es);
    if (prev_rec != nullptr) {
        auto d = c.desc(w.oc1.lin, prev_rec, SKIMI_BF16X3_REC, feat, F * h1 * w1, c1, adt, f2);
        c.conv_geom(d, w.oc1, F, h1, w1, h1, w1);
        d.x3_scratch = prev_rec + (size_t)F * h1 * w1 * feat * 4;
        d.x3_scratch_bytes = 256;
        if (!c.rc && !c.dry()) c.rc = gemm_dispatch(&d, c.st, c.slab, c.slab_bytes, 0);
    } else {
        auto d = c.desc(w.oc1.lin, prev, adt, feat, F * h1 * w1, c1, adt, f2);
        c.conv_geom(d, w.oc1, F, h1, w1, h1, w1);
        c.gemm(d);
    }
    const int Ho = ph * c.h->cfg.patch_size / w.down_ratio, Wo = pw * c.h->cfg.patch_size / w.down_ratio;
    const UvTab* uvt = w.pos_embed ? find_uv(c.tabs, Wo, Ho, f2) : nullptr;
    if (w.pos_embed && !uvt && !c.rc && !c.dry()) { set_error("uv table missing"); c.rc = SKIMI_ERR_STATE; }
    static const int no_direct = getenv("SKIMI_CONV_DIRECT") ? !atoi(getenv("SKIMI_CONV_DIRECT")) : 0;   // A/B timing
    const bool direct = !w.feature_only && w.oc2a_direct != nullptr && adt == SKIMI_F32 && !no_direct;
    void* c2 = nullptr;
    if (direct) {
        // upsample (+ UV embedding) straight into bf16 hi / lo planes, then the direct 3x3 -> 32 conv
        const size_t pe = (size_t)F * Ho * Wo * f2;
        unsigned short* rec = (unsigned short*)c.ar.alloc(pe * 4);   // [pixel][hi f2 | lo f2]
        c2 = c.ar.alloc((size_t)F * Ho * Wo * 32 * es);
        if (!c.rc && !c.dry())
            c.rc = bilinear_ac_planes_launch((const float*)c1, rec, F, h1, w1, Ho, Wo, f2, c.st, uvt ? uvt->tx : nullptr,
                                             uvt ? uvt->ty : nullptr);
        if (!c.rc && !c.dry())
            c.rc = conv_direct_n32_launch(rec, rec + f2, w.oc2a_direct, w.oc2a.lin.b, (float*)c2, F, Ho, Wo, f2, 1, c.st,
                                          2L * f2);
    } else {
        const int udt = (w.feature_only && w.out_f32) ? SKIMI_F32 : adt;   // the tracker's feature map: fp32 out of the last upsample
        void* c1u = c.ar.alloc((size_t)F * Ho * Wo * f2 * Ctx::esz(udt));
        // upsample to the output size with the UV positional embedding added in the same pass
        if (!c.rc && !c.dry())
            c.rc = dpt_fused_ln(w, f2, udt)
                       ? bilinear_ac_launch(c1, c1u, adt, F, h1, w1, Ho, Wo, f2, c.st, uvt ? uvt->tx : nullptr, uvt ? uvt->ty : nullptr, udt,
                                            w.out_ln->g, w.out_ln->b, 1e-5f)
                       : bilinear_ac_launch(c1, c1u, adt, F, h1, w1, Ho, Wo, f2, c.st, uvt ? uvt->tx : nullptr, uvt ? uvt->ty : nullptr, udt);
        if (w.feature_only) return c1u;   // caller releases the arena
        c2 = c.ar.alloc((size_t)F * Ho * Wo * 32 * es);
        auto d = c.desc(w.oc2a.lin, c1u, adt, f2, F * Ho * Wo, c2, adt, 32);
        c.conv_geom(d, w.oc2a, F, Ho, Wo, Ho, Wo);
        d.act = SKIMI_ACT_RELU;
        c.gemm(d);
    }
    if (!c.rc && !c.dry())
        c.rc = dpt_out_launch(c2, adt, w.oc2b_w, w.oc2b_b, w.n_out, pts, conf, (long)F * Ho * Wo, act_mode, c.st);
    c.ar.release(mk);
    (void)H; (void)W;
    return nullptr;
}

#define TRACK_PART 2
#include "track_impl.inc"
#undef TRACK_PART

// camera head (vggt/vggt/heads/camera_head.py:73-141), fp32 activations
void run_camera(Ctx& c, const CamW& w, const float* sf, const float* sg, int B, int S, int P, int C, float* out_list,
                float* out_last) {
    const skimi_vggt_config& cfg = c.h->cfg;
    const int R = B * S, D = 2 * C;
    const size_t mk = c.ar.mark();
    float* pose_tokens = (float*)c.ar.alloc((size_t)R * D * 4);
    float* xn = (float*)c.ar.alloc((size_t)R * D * 4);
    float* xc = (float*)c.ar.alloc((size_t)R * D * 4);
    float* e = (float*)c.ar.alloc((size_t)R * D * 4);
    float* mod = (float*)c.ar.alloc((size_t)R * 3 * D * 4);
    float* pred16 = (float*)c.ar.alloc((size_t)R * 16 * 4);
    float* t2 = (float*)c.ar.alloc((size_t)R * (D / 2) * 4);
    float* delta = (float*)c.ar.alloc((size_t)R * 9 * 4);
    BlockBufs bb;
    bb.xn = c.ar.alloc((size_t)R * D * 4 + 256);     // + the zero page behind bf16x3 records (run_block)
    bb.qkv = c.ar.alloc((size_t)R * 3 * D * 4);
    bb.ao = c.ar.alloc((size_t)R * D * 4 + 256);
    bb.hid = c.ar.alloc((size_t)R * 4 * D * 4 + 256);
    // camera token = token 0 of every frame of the last [frame | global] intermediate
    c.ln(sf, sg, (long)P * C, R, D, w.token_norm, 1e-5f, pose_tokens, SKIMI_F32);
    LNw none;
    for (int it = 0; it < cfg.cam_iters; ++it) {
        if (it == 0 && !c.rc && !c.dry()) c.rc = tile_vec_launch(w.empty16, pred16, 16, R, c.st);
        {
            auto d = c.desc(w.embed_pose, pred16, SKIMI_F32, 16, R, e, SKIMI_F32, D);
            d.act = SKIMI_ACT_SILU;   // poseLN_modulation = Sequential(SiLU, Linear)
            c.gemm(d);
        }
        {
            auto d = c.desc(w.poseLN, e, SKIMI_F32, D, R, mod, SKIMI_F32, 3 * D);
            c.gemm(d);
        }
        c.ln(pose_tokens, nullptr, D, R, D, none, 1e-6f, xn, SKIMI_F32);
        if (!c.rc && !c.dry()) c.rc = adaln_launch(xn, pose_tokens, mod, xc, R, D, c.st);
        for (const BlockW& b : w.trunk) run_block(c, b, xc, B, S, D, cfg.cam_heads, 1e-5f, false, bb);
        c.ln(xc, nullptr, D, R, D, w.trunk_norm, 1e-5f, xn, SKIMI_F32);
        {
            auto d = c.desc(w.fc1, xn, SKIMI_F32, D, R, t2, SKIMI_F32, D / 2);
            d.act = SKIMI_ACT_GELU;
            c.gemm(d);
        }
        {
            auto d = c.desc(w.fc2, t2, SKIMI_F32, D / 2, R, delta, SKIMI_F32, 9);
            c.gemm(d);
        }
        float* dst = out_list ? out_list + (size_t)it * R * 9 : (it == cfg.cam_iters - 1 ? out_last : xn /*scratch*/);
        if (!c.rc && !c.dry()) c.rc = pose_update_launch(delta, pred16, dst, R, it == 0, c.st);
        if (it == cfg.cam_iters - 1 && out_list && out_last) c.copy(out_last, dst, (size_t)R * 9 * 4);
    }
    c.ar.release(mk);
}

// Looks up an (H, W) entry with room for F frames or builds one.  Called with h->prep_mu held.  A build that fails
// part-way publishes nothing (the half-built entry frees its device buffers on the way out).
int prepare(skimi_vggt* h, int F, int S, int H, int W, const ShapeTabs** out) {
    auto& entries = h->tabs[std::make_pair(H, W)];
    int largest = 0;
    for (const auto& e : entries) {
        if (e->frames >= F) {
            *out = e.get();
            return SKIMI_OK;
        }
        largest = std::max(largest, e->frames);
    }
    const ShapeTabs* first = entries.empty() ? nullptr : entries.front().get();
    const int cap = std::max(F, 2 * largest);
    (void)S;
    std::unique_ptr<ShapeTabs> tabs(new ShapeTabs());
    ShapeTabs* t_ = tabs.get();
    t_->frames = cap;
    const skimi_vggt_config& cfg = h->cfg;
    const int p = cfg.patch_size, ph = H / p, pw = W / p, nsp = 1 + cfg.num_register_tokens, P = nsp + ph * pw;
    auto up = [&](const void* src, size_t bytes, void** dst) -> int {
        SKIMI_HIP(hipMalloc(dst, bytes));
        t_->owned.push_back(*dst);
        SKIMI_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return SKIMI_OK;
    };
    int rc;
    if (cfg.use_dino) {
        const bool native = H == W && H == cfg.dino_img_size;
        t_->dino_pos = native ? h->dino_pos : h->dino_pos_alt.at({H, W});   // presence checked by check_shape
    }
    // positions: (y, x) + 1 for patches, 0 for the special tokens (rope.py:39-59, aggregator.py:219-228)
    std::vector<int> pos((size_t)cap * P * 2, 0);
    for (int f = 0; f < cap; ++f)
        for (int y = 0; y < ph; ++y)
            for (int x = 0; x < pw; ++x) {
                const size_t i = ((size_t)f * P + nsp + (size_t)y * pw + x) * 2;
                pos[i] = y + 1;
                pos[i + 1] = x + 1;
            }
    if ((rc = up(pos.data(), pos.size() * 4, (void**)&t_->pos))) return rc;
    if (first) {   // RoPE / UV tables depend on (H, W) only: shared with (and owned by) the first entry of this frame size
        t_->rope_cos = first->rope_cos; t_->rope_sin = first->rope_sin; t_->rope_npos = first->rope_npos;
        t_->uv = first->uv;
        *out = t_;
        entries.push_back(std::move(tabs));
        return SKIMI_OK;
    }
    // RoPE tables (rope.py:86-117), fp32 arithmetic as torch: 1/100^(i/16), pos*inv_freq, cos/sin
    const int npos = std::max(ph, pw) + 1;
    std::vector<float> cs((size_t)npos * 16), sn((size_t)npos * 16);
    for (int i = 0; i < 16; ++i) {
        const float expo = (float)(2 * i) / 32.0f;
        const float inv = 1.0f / powf(100.0f, expo);
        for (int q = 0; q < npos; ++q) {
            const float a = (float)q * inv;
            cs[(size_t)q * 16 + i] = cosf(a);
            sn[(size_t)q * 16 + i] = sinf(a);
        }
    }
    if ((rc = up(cs.data(), cs.size() * 4, (void**)&t_->rope_cos))) return rc;
    if ((rc = up(sn.data(), sn.size() * 4, (void**)&t_->rope_sin))) return rc;
    t_->rope_npos = npos;
    // UV sin/cos tables (heads/utils.py:11-109 via dpt_head.py:249-259), float64 then float, x 0.1
    auto add_uv = [&](int w, int hh, int C) -> int {
        if (find_uv(t_, w, hh, C)) return SKIMI_OK;
        const double aspect = (double)W / (double)H;
        const double diag = sqrt(aspect * aspect + 1.0);
        const double sx = aspect / diag, sy = 1.0 / diag;
        const int half = C / 2, quarter = C / 4;
        std::vector<float> tx((size_t)w * half), ty((size_t)hh * half);
        auto fill = [&](std::vector<float>& tab, int n, double span) {
            // torch.linspace(-span*(n-1)/n, span*(n-1)/n, n) in float32
            const float lo = (float)(-span * (n - 1) / n), hi = (float)(span * (n - 1) / n);
            const float step = n > 1 ? (hi - lo) / (float)(n - 1) : 0.f;
            for (int i = 0; i < n; ++i) {
                const float pv = i < n / 2 ? lo + step * (float)i : hi - step * (float)(n - 1 - i);
                for (int j = 0; j < quarter; ++j) {
                    const double omega = 1.0 / pow(100.0, (double)j / (double)quarter);
                    const double a = (double)pv * omega;
                    tab[(size_t)i * half + j] = (float)sin(a) * 0.1f;
                    tab[(size_t)i * half + quarter + j] = (float)cos(a) * 0.1f;
                }
            }
        };
        fill(tx, w, sx);
        fill(ty, hh, sy);
        UvTab t{w, hh, C, nullptr, nullptr};
        int r2;
        if ((r2 = up(tx.data(), tx.size() * 4, (void**)&t.tx))) return r2;
        if ((r2 = up(ty.data(), ty.size() * 4, (void**)&t.ty))) return r2;
        t_->uv.push_back(t);
        return SKIMI_OK;
    };
    if (cfg.enable_depth || cfg.enable_point) {
        for (int i = 0; i < 4; ++i)
            if ((rc = add_uv(pw, ph, cfg.dpt_out_channels[i]))) return rc;
        if ((rc = add_uv(pw * p, ph * p, cfg.dpt_features / 2))) return rc;
    }
    *out = t_;
    entries.push_back(std::move(tabs));
    return SKIMI_OK;
}

int forward_impl(skimi_vggt* h, Ctx& c, const float* images, const float* query, int B, int S, int H, int W, int nq,
                 const skimi_vggt_outputs* out) {
    const skimi_vggt_config& cfg = h->cfg;
    const int p = cfg.patch_size, C = cfg.embed_dim, ph = H / p, pw = W / p, np = ph * pw;
    const int nsp = 1 + cfg.num_register_tokens, P = nsp + np, F = B * S, M = F * P;
    const int prec = cfg.prec, adt = Ctx::act_dt(prec);
    const size_t es = Ctx::esz(adt);

    // split-K slab: large enough for the skinny camera-head / small-config GEMMs; the fp32-accurate contractions keep one
    // plane per K split (fixed summation order), so it holds several [M, N] planes of those
    c.slab_bytes = std::max<size_t>((size_t)F * 6 * C * 4 * 4 * 4, 4u << 20);
    c.slab = c.ar.alloc(c.slab_bytes);
    // zeroed once per forward; every split-K epilogue leaves the slab zero again
    if (!c.dry() && hipMemsetAsync(c.slab, 0, c.slab_bytes, c.st) != hipSuccess) c.rc = SKIMI_ERR_HIP;

    float* x = (float*)c.ar.alloc((size_t)M * C * 4);
    // kept intermediates
    std::vector<int> keep;
    auto want = [&](int l) { for (int k : keep) if (k == l) return; keep.push_back(l); };
    const bool run_depth = cfg.enable_depth && out && (out->depth || out->depth_conf);
    const bool run_point = cfg.enable_point && out && (out->world_points || out->world_points_conf);
    const bool run_cam = cfg.enable_camera && out && (out->pose_enc || out->pose_enc_list);
    const bool do_track = cfg.enable_track && out && out->track && query && nq > 0;
    if (run_depth || run_point || do_track) for (int i = 0; i < 4; ++i) want(cfg.dpt_layers[i]);
    if (run_cam || (out && out->tokens_last)) want(cfg.depth - 1);
    std::map<int, std::pair<float*, float*>> saved;
    for (int l : keep) {
        float* a = (float*)c.ar.alloc((size_t)M * C * 4);
        float* b = (float*)c.ar.alloc((size_t)M * C * 4);
        saved[l] = {a, b};
    }
    {
        const size_t mk = c.ar.mark();
        BlockBufs bb;
        bb.xn = c.ar.alloc((size_t)M * C * es + 256);     // + the zero page behind bf16x3 records (run_block)
        bb.qkv = c.ar.alloc((size_t)M * 3 * C * es);
        bb.ao = c.ar.alloc((size_t)M * C * es + 256);
        bb.hid = c.ar.alloc((size_t)M * 4 * C * es + 256);
        if (cfg.prec == SKIMI_PREC_FP8) {   // MXFP8 scratch of the widest quantised activation (the MLP hidden)
            const size_t kp = align_up((size_t)4 * C, 128);
            bb.q8 = c.ar.alloc((size_t)M * kp);
            bb.q8s = c.ar.alloc((size_t)M * (kp / 32));
        }
        // ---- patch embed (aggregator.py:195-208) ----
        void* pa = c.ar.alloc((size_t)F * np * h->patch_kp * es);
        if (!c.rc && !c.dry()) c.rc = patch_gather_launch(images, pa, adt, F, H, W, p, h->patch_kp, c.st);
        {
            auto d = c.desc(h->patch_proj, pa, adt, h->patch_kp, F * np, x, SKIMI_F32, C);
            d.out_rows_per_batch = np; d.out_batch_stride = P; d.out_row_off = nsp;
            if (cfg.use_dino) {
                // + pos_embed[1 + p] (vision_transformer.py:221), broadcast over frames
                const bool native = H == W && H == cfg.dino_img_size;
                (void)native;
                d.resid = c.tabs ? c.tabs->dino_pos : h->dino_pos; d.ldr = C;   // dry run: any non-null pointer
                d.resid_rows_per_batch = np; d.resid_batch_stride = 0; d.resid_row_off = 1;
            }
            c.gemm(d);
        }
        if (cfg.use_dino) {
            if (!c.rc && !c.dry()) c.rc = special_tokens_launch(x, h->dino_special, F, S, P, nsp, C, c.st);
            for (const BlockW& b : h->dino) run_block(c, b, x, F, P, C, cfg.dino_heads, 1e-6f, false, bb);
            // x_norm_patchtokens (vision_transformer.py:264-268): LayerNorm in place (row-local)
            c.ln(x, nullptr, C, M, C, h->dino_norm, 1e-6f, x, SKIMI_F32);
        }
        // camera / register tokens, frame 0 vs others (aggregator.py:210-217, 308-331)
        if (!c.rc && !c.dry()) c.rc = special_tokens_launch(x, h->agg_special, F, S, P, nsp, C, c.st);
        // ---- alternating attention (aggregator.py:237-253) ----
        for (int i = 0; i < cfg.depth && !c.rc; ++i) {
            run_block(c, h->frame[i], x, F, P, C, cfg.num_heads, 1e-5f, true, bb);
            auto it = saved.find(i);
            if (it != saved.end()) c.copy(it->second.first, x, (size_t)M * C * 4);
            run_block(c, h->global[i], x, B, S * P, C, cfg.num_heads, 1e-5f, true, bb);
            if (it != saved.end()) c.copy(it->second.second, x, (size_t)M * C * 4);
        }
        c.ar.release(mk);
    }
    if (out && out->tokens_last && !c.rc && !c.dry()) {
        // [B,S,P,2C] = cat(frame, global) of the last layer: two strided 2-D copies
        auto& sv = saved[cfg.depth - 1];
        if (hipMemcpy2DAsync(out->tokens_last, (size_t)2 * C * 4, sv.first, (size_t)C * 4, (size_t)C * 4, M,
                             hipMemcpyDeviceToDevice, c.st) != hipSuccess ||
            hipMemcpy2DAsync(out->tokens_last + C, (size_t)2 * C * 4, sv.second, (size_t)C * 4, (size_t)C * 4, M,
                             hipMemcpyDeviceToDevice, c.st) != hipSuccess) {
            set_error("hipMemcpy2DAsync failed");
            c.rc = SKIMI_ERR_HIP;
        }
    }
    if (run_cam) {
        auto& sv = saved[cfg.depth - 1];
        run_camera(c, h->cam, sv.first, sv.second, B, S, P, C, out->pose_enc_list, out->pose_enc);
    }
    float* sf[4];
    float* sg[4];
    if (run_depth || run_point || do_track)
        for (int i = 0; i < 4; ++i) {
            sf[i] = saved[cfg.dpt_layers[i]].first;
            sg[i] = saved[cfg.dpt_layers[i]].second;
        }
    if (run_depth) {
        float* pts = out->depth ? out->depth : (float*)c.ar.alloc((size_t)F * H * W * 4);
        float* cf = out->depth_conf ? out->depth_conf : (float*)c.ar.alloc((size_t)F * H * W * 4);
        run_dpt(c, h->depth, sf, sg, F, P, nsp, ph, pw, C, H, W, pts, cf, 0);
    }
    if (run_point) {
        float* pts = out->world_points ? out->world_points : (float*)c.ar.alloc((size_t)F * H * W * 3 * 4);
        float* cf = out->world_points_conf ? out->world_points_conf : (float*)c.ar.alloc((size_t)F * H * W * 4);
        run_dpt(c, h->point, sf, sg, F, P, nsp, ph, pw, C, H, W, pts, cf, 1);
    }
    if (do_track) run_track(c, sf, sg, B, S, P, nsp, ph, pw, C, H, W, query, nq, out->track, out->vis, out->conf);
    return c.rc;
}

}  // namespace

extern "C" {

skimi_vggt* skimi_vggt_create(const skimi_vggt_config* cfg) {
    if (!cfg) { set_error("skimi_vggt_create: null config"); return nullptr; }
    if (cfg->embed_dim <= 0 || cfg->num_heads <= 0 || cfg->embed_dim / cfg->num_heads != 64 ||
        cfg->embed_dim % cfg->num_heads != 0) {
        set_error("skimi_vggt_create: embed_dim / num_heads must be 64");
        return nullptr;
    }
    if (cfg->use_dino && cfg->embed_dim / cfg->dino_heads != 64) {
        set_error("skimi_vggt_create: DINOv2 head_dim must be 64");
        return nullptr;
    }
    if (cfg->patch_size <= 0 || cfg->depth <= 0 || cfg->num_register_tokens < 0) {
        set_error("skimi_vggt_create: bad sizes");
        return nullptr;
    }
    for (int i = 0; i < 4; ++i)
        if (cfg->dpt_layers[i] < 0 || cfg->dpt_layers[i] >= cfg->depth) {
            set_error("skimi_vggt_create: dpt_layers[%d] = %d outside [0, depth)", i, cfg->dpt_layers[i]);
            return nullptr;
        }
    if (cfg->prec < SKIMI_PREC_BF16 || cfg->prec > SKIMI_PREC_F16) {
        set_error("skimi_vggt_create: prec %d is not one of SKIMI_PREC_BF16 / BF16X3 / FP8 / F16", cfg->prec);
        return nullptr;
    }
    if (cfg->head_prec != SKIMI_PREC_BF16 && cfg->head_prec != SKIMI_PREC_BF16X3 && cfg->head_prec != SKIMI_PREC_F16) {
        set_error("skimi_vggt_create: head_prec %d: the depth / point heads run in SKIMI_PREC_BF16X3 (the reference's fp32) or, as a "
                  "faster, less accurate option, on SKIMI_PREC_F16 / SKIMI_PREC_BF16 operands", cfg->head_prec);
        return nullptr;
    }
    skimi_vggt* h = new skimi_vggt();
    h->cfg = *cfg;
    return h;
}

void skimi_vggt_destroy(skimi_vggt* h) {
    if (!h) return;
    for (auto& kv : h->raw) (void)hipFree(kv.second.first);
    for (void* p : h->owned) (void)hipFree(p);
    h->tabs.clear();
    for (auto& kv : h->dino_pos_alt) (void)hipFree(kv.second);
    delete h;
}

int skimi_vggt_set_weight(skimi_vggt* h, const char* key, const float* data, int64_t n, int32_t on_device) {
    SKIMI_CHECK_ARG(h && key && data && n > 0, "skimi_vggt_set_weight: bad arguments");
    float* d = nullptr;
    SKIMI_HIP(hipMalloc((void**)&d, (size_t)n * 4));
    hipError_t e = hipMemcpy(d, data, (size_t)n * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        set_error("skimi_vggt_set_weight: copy failed: %s", hipGetErrorString(e));
        return SKIMI_ERR_HIP;
    }
    auto it = h->raw.find(key);
    if (it != h->raw.end()) (void)hipFree(it->second.first);
    h->raw[key] = {d, n};
    h->finalized = false;
    return SKIMI_OK;
}

int skimi_vggt_set_pos_embed(skimi_vggt* h, int32_t H, int32_t W, const float* pos_embed, int32_t on_device) {
    SKIMI_CHECK_ARG(h && pos_embed && h->cfg.use_dino, "skimi_vggt_set_pos_embed: needs a DINOv2 model and a table");
    const int p = h->cfg.patch_size;
    SKIMI_CHECK_ARG(H > 0 && W > 0 && H % p == 0 && W % p == 0, "skimi_vggt_set_pos_embed: bad size %dx%d", H, W);
    const size_t n = ((size_t)(H / p) * (W / p) + 1) * h->cfg.embed_dim;
    float* d = nullptr;
    SKIMI_HIP(hipMalloc((void**)&d, n * 4));
    hipError_t e = hipMemcpy(d, pos_embed, n * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        set_error("skimi_vggt_set_pos_embed: copy failed: %s", hipGetErrorString(e));
        return SKIMI_ERR_HIP;
    }
    // published under the handle's table lock; a table registered again for the same size replaces the
    // CONTENTS of the existing buffer (its address stays valid for forwards already enqueued)
    std::lock_guard<std::mutex> lk(h->prep_mu);
    auto it = h->dino_pos_alt.find({H, W});
    if (it != h->dino_pos_alt.end()) {
        e = hipMemcpy(it->second, d, n * 4, hipMemcpyDeviceToDevice);
        (void)hipFree(d);
        if (e != hipSuccess) {
            set_error("skimi_vggt_set_pos_embed: copy failed: %s", hipGetErrorString(e));
            return SKIMI_ERR_HIP;
        }
        return SKIMI_OK;
    }
    h->dino_pos_alt[{H, W}] = d;
    return SKIMI_OK;
}

int skimi_vggt_finalize(skimi_vggt* h) {
    SKIMI_CHECK_ARG(h, "skimi_vggt_finalize: null handle");
    const skimi_vggt_config& cfg = h->cfg;
    for (void* p : h->owned) (void)hipFree(p);
    h->owned.clear();
    Packer pk{h};
    const int C = cfg.embed_dim, p = cfg.patch_size, R = cfg.num_register_tokens, nsp = 1 + R;
    const int hidden = 4 * C, D = 2 * C;
    const std::string A = "aggregator";
    const int kraw = 3 * p * p;
    // ---- patch embed ----
    const std::string pe = cfg.use_dino ? A + ".patch_embed.patch_embed.proj" : A + ".patch_embed.proj";
    h->patch_proj = pk.linear(pe, C, kraw, cfg.prec == SKIMI_PREC_FP8 ? SKIMI_PREC_BF16 : cfg.prec);   // FP8 mode: only the block Linears are MXFP8
    h->patch_kp = h->patch_proj.K;
    if (cfg.use_dino) {
        const std::string d = A + ".patch_embed";
        const int g = cfg.dino_img_size / p, np0 = g * g;
        h->dino_pos = pk.keep(d + ".pos_embed", (int64_t)(np0 + 1) * C);
        float* cls = pk.raw(d + ".cls_token", C);
        float* reg = R ? pk.raw(d + ".register_tokens", (int64_t)R * C) : nullptr;
        (void)pk.raw(d + ".mask_token", C);   // present in the state_dict, unused in inference
        h->dino_special = (float*)pk.dmalloc((size_t)2 * nsp * C * 4);
        if (!pk.rc) {
            // row 0 = cls + pos_embed[0] (vision_transformer.py:220-221), rows 1..R = registers
            std::vector<float> hc(C), hp(C), hr((size_t)R * C), tab((size_t)2 * nsp * C);
            SKIMI_HIP(hipMemcpy(hc.data(), cls, C * 4, hipMemcpyDeviceToHost));
            SKIMI_HIP(hipMemcpy(hp.data(), h->dino_pos, C * 4, hipMemcpyDeviceToHost));
            if (R) SKIMI_HIP(hipMemcpy(hr.data(), reg, (size_t)R * C * 4, hipMemcpyDeviceToHost));
            for (int s = 0; s < 2; ++s) {
                for (int c = 0; c < C; ++c) tab[((size_t)s * nsp) * C + c] = hc[c] + hp[c];
                for (int r = 0; r < R; ++r)
                    for (int c = 0; c < C; ++c) tab[((size_t)s * nsp + 1 + r) * C + c] = hr[(size_t)r * C + c];
            }
            SKIMI_HIP(hipMemcpy(h->dino_special, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        }
        h->dino.clear();
        for (int i = 0; i < cfg.dino_depth; ++i)
            h->dino.push_back(pk.block(d + ".blocks." + std::to_string(i), C, hidden, false, 64, cfg.prec));
        h->dino_norm = pk.ln(d + ".norm", C);
    }
    // camera_token [1,2,1,C], register_token [1,2,R,C] -> table [2][1+R][C]
    {
        float* cam = pk.raw(A + ".camera_token", (int64_t)2 * C);
        float* reg = R ? pk.raw(A + ".register_token", (int64_t)2 * R * C) : nullptr;
        h->agg_special = (float*)pk.dmalloc((size_t)2 * nsp * C * 4);
        if (!pk.rc) {
            for (int s = 0; s < 2; ++s) {
                SKIMI_HIP(hipMemcpy(h->agg_special + (size_t)s * nsp * C, cam + (size_t)s * C, C * 4, hipMemcpyDeviceToDevice));
                if (R) SKIMI_HIP(hipMemcpy(h->agg_special + ((size_t)s * nsp + 1) * C, reg + (size_t)s * R * C,
                                           (size_t)R * C * 4, hipMemcpyDeviceToDevice));
            }
        }
    }
    h->frame.clear();
    h->global.clear();
    for (int i = 0; i < cfg.depth; ++i) h->frame.push_back(pk.block(A + ".frame_blocks." + std::to_string(i), C, hidden, true, 64, cfg.prec));
    for (int i = 0; i < cfg.depth; ++i) h->global.push_back(pk.block(A + ".global_blocks." + std::to_string(i), C, hidden, true, 64, cfg.prec));
    // ---- camera head: fp32 (BF16X3) always ----
    if (cfg.enable_camera) {
        const std::string Hc = "camera_head";
        const int X3 = SKIMI_PREC_BF16X3;
        h->cam.trunk.clear();
        for (int i = 0; i < cfg.cam_trunk_depth; ++i)
            h->cam.trunk.push_back(pk.block(Hc + ".trunk." + std::to_string(i), D, 4 * D, false, 0, X3));
        h->cam.token_norm = pk.ln(Hc + ".token_norm", D);
        h->cam.trunk_norm = pk.ln(Hc + ".trunk_norm", D);
        float* ept = pk.raw(Hc + ".empty_pose_tokens", 9);
        h->cam.empty16 = (float*)pk.dmalloc(16 * 4);
        if (!pk.rc) pk.rc = pad_cols_launch(ept, h->cam.empty16, 1, 9, 16, pk.st);
        h->cam.embed_pose = pk.linear(Hc + ".embed_pose", D, 9, X3);   // K padded to 16
        h->cam.poseLN = pk.linear(Hc + ".poseLN_modulation.1", 3 * D, D, X3);
        h->cam.fc1 = pk.linear(Hc + ".pose_branch.fc1", D / 2, D, X3);
        h->cam.fc2 = pk.linear(Hc + ".pose_branch.fc2", 9, D / 2, X3);
    }
    if (cfg.enable_point) h->point = pk.dpt("point_head", D, cfg.dpt_features, cfg.dpt_out_channels, 4, false, cfg.head_prec);
    if (cfg.enable_depth) h->depth = pk.dpt("depth_head", D, cfg.dpt_features, cfg.dpt_out_channels, 2, false, cfg.head_prec);
    if (cfg.enable_track) pack_track(pk, h);
    if (pk.rc) return pk.rc;
    SKIMI_HIP(hipStreamSynchronize(pk.st));
    // release the staged fp32 copies
    for (auto it = h->raw.begin(); it != h->raw.end();) {
        (void)hipFree(it->second.first);
        it = h->raw.erase(it);
    }
    h->finalized = true;
    return SKIMI_OK;
}

static int check_shape(const skimi_vggt* h, int B, int S, int H, int W) {
    SKIMI_CHECK_ARG(B > 0 && S > 0, "skimi_vggt: B and S must be positive");
    const int p = h->cfg.patch_size;
    // patch_embed.py:69-70
    SKIMI_CHECK_ARG(H > 0 && H % p == 0, "Input image height %d is not a multiple of patch height %d", H, p);
    SKIMI_CHECK_ARG(W > 0 && W % p == 0, "Input image width %d is not a multiple of patch width: %d", W, p);
    if (h->cfg.use_dino && !(H == W && H == h->cfg.dino_img_size)) {
        std::lock_guard<std::mutex> lk(const_cast<skimi_vggt*>(h)->prep_mu);
        SKIMI_CHECK_ARG(h->dino_pos_alt.count({H, W}) != 0,
                        "DINOv2 pos_embed for %dx%d (model built for %d) has not been registered: call "
                        "skimi_vggt_set_pos_embed with the bicubic-antialias resized table first", H, W,
                        h->cfg.dino_img_size);
    }
    return SKIMI_OK;
}

size_t skimi_vggt_workspace_bytes(skimi_vggt* h, int32_t B, int32_t S, int32_t H, int32_t W, int32_t n_query) {
    if (!h || check_shape(h, B, S, H, W)) return 0;
    Ctx c{h, nullptr};
    c.ar.dry = true;
    skimi_vggt_outputs all;
    memset(&all, 0, sizeof all);
    // assume every enabled head runs and writes straight to caller buffers
    float* one = (float*)(uintptr_t)16;
    all.pose_enc = all.pose_enc_list = all.depth = all.depth_conf = all.world_points = all.world_points_conf = one;
    all.tokens_last = one;
    if (n_query > 0) all.track = all.vis = all.conf = one;
    forward_impl(h, c, nullptr, n_query > 0 ? one : nullptr, B, S, H, W, n_query, &all);
    return align_up(c.ar.peak, 256) + 256;
}

int skimi_vggt_rope_positions(skimi_vggt* h, int32_t frames, int32_t H, int32_t W, int32_t* positions) {
    SKIMI_CHECK_ARG(h && positions && frames > 0, "skimi_vggt_rope_positions: bad arguments");
    int rc;
    if ((rc = check_shape(h, 1, frames, H, W))) return rc;
    const ShapeTabs* tabs = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->prep_mu);
        if ((rc = prepare(h, frames, frames, H, W, &tabs))) return rc;
    }
    const int p = h->cfg.patch_size;
    const size_t P = 1 + h->cfg.num_register_tokens + (size_t)(H / p) * (W / p);
    SKIMI_HIP(hipMemcpy(positions, tabs->pos, (size_t)frames * P * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice));
    return SKIMI_OK;
}

int skimi_vggt_forward(skimi_vggt* h, const float* images, const float* query_points, int32_t B, int32_t S, int32_t H,
                       int32_t W, int32_t n_query, const skimi_vggt_outputs* out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    SKIMI_CHECK_ARG(h && images && out && workspace, "skimi_vggt_forward: null argument");
    if (!h->finalized) {
        set_error("skimi_vggt_forward: weights not finalized");
        return SKIMI_ERR_STATE;
    }
    int rc;
    if ((rc = check_shape(h, B, S, H, W))) return rc;
    // Calls of any shapes may run at the same time from several host threads (own workspace and stream
    // each): the per-shape tables are a keyed cache of immutable entries that live as long as the handle.
    const ShapeTabs* tabs = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->prep_mu);
        if ((rc = prepare(h, B * S, S, H, W, &tabs))) return rc;
    }
    Ctx c{h, (hipStream_t)stream};
    c.tabs = tabs;
    c.ar.base = (char*)workspace;
    c.ar.cap = workspace_bytes;
    // refuse before launching anything if the arena cannot hold the plan
    {
        Ctx dry{h, nullptr};
        dry.ar.dry = true;
        forward_impl(h, dry, images, query_points, B, S, H, W, n_query, out);
        if (dry.ar.peak > workspace_bytes) {
            set_error("skimi_vggt_forward: workspace %zu < %zu", workspace_bytes, dry.ar.peak);
            return SKIMI_ERR_WORKSPACE;
        }
    }
    return forward_impl(h, c, images, query_points, B, S, H, W, n_query, out);
}

}  // extern "C"

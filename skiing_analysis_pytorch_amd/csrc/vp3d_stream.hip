// VideoPose3D TemporalModel at small batch (B = 1 .. a few clips): the weight-streaming path.
//
// At B = 1 a layer is [M = 217..267 frames] x [K = 1024 or 3 x 1024] x [N = 1024 channels]: 4.2 / 12.6 MB of
// fp32 weights against 1 MB of activations and ~2 GFLOP -- the layer is bound by moving each weight byte
// through the chip ONCE (34.3 MB per RF-27 clip against 8 TB/s).  The generic GEMM path spent
// 109 us per clip in 13 dependent launches (split-K atomics + a reduce launch per layer, an im2col, a memset,
// `profiles/r01_vp3d_summary.md`).  Here a forward is ONE launch per convolution (6 for RF 27) and nothing else:
//
//   * vp3d_expand_kernel   expand_conv (k taps x 34 inputs -> C channels) in exact fp32 FMA, BN folded, ReLU;
//                          reads the 2D keypoints directly (no im2col), writes fp32 + bf16x3 records.
//   * vp3d_mm_kernel<CT,RT> one dilated / 1x1 / shrink convolution as a bf16x3 MFMA contraction
//                          (v_mfma_f32_16x16x32_bf16; x = hi + lo, acc += Wlo Xhi + Whi Xlo + Whi Xhi).
//
// Decomposition of a layer over the 256 CUs: (N / CT channel tiles) x (ms row splits), NO split-K across
// workgroups -- so no atomics, no reduce pass, no zeroed slab: a workgroup owns CT channels x R = ceil(M / ms)
// frames for the whole K.  The ms partners of a channel tile read the same CT x K weight slice; they get
// consecutive tile ids on ONE XCD, so the slice comes from HBM once and from that XCD's L2 for the others.
// (CT, ms) is chosen per layer to minimise the bytes a workgroup pulls through its L2 port,
// CT K + (R + halo) C elements: 16 x 4 for the dilated convs, 32 x 8 for the 1x1 convs at RF 27.
//
// Inside a workgroup (8 waves): K is split over the waves (a wave = a contiguous run of 32-element K
// slices), operands go global -> registers directly, one 16-byte load per lane and fragment half, no LDS
// staging and no conversion in the loop.  Both operands are stored pre-split (x = hi + lo, bf16 each:
// weights once at finalize, activations by the producing epilogue) in FRAGMENT-MAJOR order,
//     [row tile of 16][K slice of 32][hi | lo][kg 0..3][row 0..15][8 elements]        (2 KiB per tile and slice)
// i.e. exactly the register image of a 16 x 32 MFMA operand (lane = row + 16 kg holds elements
// 8 kg .. 8 kg + 7), so a wave's load is 1 KiB of consecutive bytes.  (The first version read row-major
// [row][slice][hi 32 | lo 32] records: every lane of a load then sits in another 128-byte line, the
// texture-address unit walks 64 lines per instruction, and the dilated convs took 43 us instead of ~10.)
// The dilated gather shifts the row window by tap * dilation: a fragment then straddles two stored tiles,
// still two runs of consecutive bytes.  2 (CT / 16 + RT) loads feed 3 (CT / 16) RT MFMAs per slice, three
// slices in flight per wave.  The eight K-partials meet in LDS once, then bias, ReLU, the residual slice
// (fp32, model.py:129-135) and the stores: fp32 (next block's residual / the output) and the next layer's
// operand in fragment-major order.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace skimi {

struct Vp3dMM {
    const char* wrec;     // [Npad / 16][S][2][4][16][8] bf16, S = taps * C / 32 slices (K order: tap-major)
    const char* xrec;     // [ceil(B * Lin / 16)][C / 32][2][4][16][8] bf16
    const float* bias;    // [N]
    const float* resid;   // fp32 [B * resid_L][C] or null: row b * resid_L + l + resid_off
    float* out_f32;       // [M][ldo] or null
    char* out_rec;        // [ceil(M / 16)][N / 32][2][4][16][8] bf16 or null (N % 32 == 0)
    int M, N, C, taps, dil, Lin, Lout, resid_L, resid_off, ldo, relu;
    int R, ms;            // rows per workgroup, row splits per channel tile
    int abl;              // timing ablations (SKIMI_ABLATIONS builds only; results are wrong): 1 no MFMA, 2 weights
                          // loaded once, 4 activations loaded once
};

typedef __attribute__((ext_vector_type(8))) short v8s;

#ifdef SKIMI_ABLATIONS
// phase timestamps (s_memtime, shader cycles) of wave 0 of every workgroup of the LAST vp3d_mm launch:
// [workgroup][8]: 0 start, 1 prologue done / first loads issued, 2 main loop done, 3 partials in LDS + barrier,
// 4 reduced + epilogue math, 5 stores issued, 6 stores drained
__device__ long long vp3d_ts[1024 * 8];
#define VP3D_TS(i)                                                                    \
    do {                                                                              \
        if (tid == 0 && blockIdx.x < 1024) vp3d_ts[blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define VP3D_TS(i)
#endif

template <int CT, int RT>
__global__ __launch_bounds__(512) void vp3d_mm_kernel(const Vp3dMM p) {
    constexpr int TA = CT / 16, D = 3, NT = TA * RT;
    __shared__ f32x4 red[8][NT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    VP3D_TS(0);
    // tile id: contiguous per XCD (blocks b and b + 8 share an XCD), so the ms partners of a channel tile
    // (consecutive ids) sit on one XCD whenever ms divides the per-XCD count
    int id;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
        const int q = nblk >> 3, r = nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ct = id / p.ms, rs = id - ct * p.ms;
    const int c0 = ct * CT, r0 = rs * p.R;
    const int SC = p.C >> 5;                 // slices per tap
    const int S = p.taps * SC;               // slices of K
    const int s0 = (int)((long)wave * S / 8), s1 = (int)((long)(wave + 1) * S / 8);
    // Every workgroup of an XCD walks the same activation rows and (its ms partners) the same weight slice: in
    // lockstep all of them miss on the same lines at the same moment and each pays the fabric latency on every
    // line.  Rotating the order in which a workgroup walks its slices by its position in the XCD lets whoever
    // comes first pull a line into the L2 and the others hit.
    const int nsl = s1 - s0;
    const int rot = nsl > 0 ? ((blockIdx.x >> 3) * 5) % nsl : 0;
    auto slice_at = [&](int i) {   // i-th slice of this wave's (rotated) walk, i clamped to the last
        int j = min(i, nsl - 1) + rot;
        if (j >= nsl) j -= nsl;
        return s0 + j;
    };

    const char* wb[TA];
    int xrow[RT];         // this lane's input row at tap 0
#pragma unroll
    for (int t = 0; t < TA; ++t) wb[t] = p.wrec + ((long)(c0 / 16 + t) * S) * 2048 + lane * 16;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int m = min(r0 + t * 16 + li, p.M - 1);
        const int b = m / p.Lout, l = m - b * p.Lout;
        xrow[t] = b * p.Lin + l;
    }
    const char* xbase = p.xrec + kg * 256;
    // epilogue operands of the tile this wave finishes (wave n finishes tile n): requested now, used after the loop
    float4 ebias = make_float4(0.f, 0.f, 0.f, 0.f), eres = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave < NT) {
        const int a = wave / RT, t = wave - a * RT;
        const int m = min(r0 + t * 16 + li, p.M - 1);
        const int c = c0 + a * 16 + 4 * kg;
        ebias.x = p.bias[min(c, p.N - 1)]; ebias.y = p.bias[min(c + 1, p.N - 1)];
        ebias.z = p.bias[min(c + 2, p.N - 1)]; ebias.w = p.bias[min(c + 3, p.N - 1)];
        if (p.resid) {
            const int b = m / p.Lout, l = m - b * p.Lout;
            eres = *reinterpret_cast<const float4*>(p.resid + ((long)b * p.resid_L + l + p.resid_off) * p.C + c);
        }
    }
    v8s ah[D][TA], al[D][TA], bh[D][RT], bl[D][RT];
    f32x4 acc[TA][RT];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load = [&](int d, int s) {
#ifdef SKIMI_ABLATIONS
        if ((p.abl & 2) && s != slice_at(0)) goto skip_w;
#endif
        {
        const long woff = (long)s * 2048;
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            ah[d][t] = *reinterpret_cast<const v8s*>(wb[t] + woff);
            al[d][t] = *reinterpret_cast<const v8s*>(wb[t] + woff + 1024);
        }
        }
#ifdef SKIMI_ABLATIONS
    skip_w:
        if ((p.abl & 4) && s != slice_at(0)) return;
#endif
        const int tap = s / SC, cs = s - tap * SC;
        const int shift = tap * p.dil;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int row = xrow[t] + shift;
            const char* xp = xbase + ((long)(row >> 4) * SC + cs) * 2048 + (row & 15) * 16;
            bh[d][t] = *reinterpret_cast<const v8s*>(xp);
            bl[d][t] = *reinterpret_cast<const v8s*>(xp + 1024);
        }
    };
    auto mma = [&](int d) {
#ifdef SKIMI_ABLATIONS
        if (p.abl & 1) {   // keep the operands live without the MFMAs
#pragma unroll
            for (int a = 0; a < TA; ++a) acc[a][0][0] += (float)(ah[d][a][0] + al[d][a][0]);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[0][t][1] += (float)(bh[d][t][0] + bl[d][t][0]);
            return;
        }
#endif
        // term-major: no MFMA waits for the accumulator of the one before it
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[d][a], bh[d][t], acc[a][t], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[d][a], bl[d][t], acc[a][t], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[d][a], bh[d][t], acc[a][t], 0, 0, 0);
    };

    // Software pipeline, D slices in flight.  Every load below is UNCONDITIONAL (slice index clamped to the
    // wave's last slice, a redundant re-read at the tail), so that the number of loads outstanding at each
    // MFMA group is a compile-time fact and hipcc emits counted s_waitcnt vmcnt(2 (D-1) (TA+RT)) instead of
    // draining the queue; the ragged tail (< D slices) is peeled.
    if (nsl > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) load(d, slice_at(d));
        VP3D_TS(1);
        int i = 0;
        for (; i + D <= nsl; i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                mma(d);
                load(d, slice_at(i + d + D));
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (i + d < nsl) mma(d);
    }

    VP3D_TS(2);
    // the eight K-partials of every 16 x 16 tile meet in LDS; tile n is finished by wave n
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int t = 0; t < RT; ++t) red[wave][a * RT + t][lane] = acc[a][t];
    __syncthreads();
    VP3D_TS(3);
    if (wave < NT) {
        f32x4 v = red[0][wave][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const f32x4 u = red[w][wave][lane];
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
        }
        const int a = wave / RT, t = wave - a * RT;
        // D[i][j]: i = channel (A row) = 4 (lane >> 4) + r, j = frame (B column) = lane & 15
        const int m = r0 + t * 16 + li;
        const int c = c0 + a * 16 + 4 * kg;
        if (m < min(r0 + p.R, p.M) && c < p.N) {
            VP3D_TS(4);
            const float eb[4] = {ebias.x, ebias.y, ebias.z, ebias.w};
            const float er[4] = {eres.x, eres.y, eres.z, eres.w};
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r] + eb[r];
                if (p.relu) x = fmaxf(x, 0.f);
                o[r] = x + er[r];       // x = res + relu(bn(conv(x))), model.py:129-135 (er = 0 without a residual)
            }
            if (p.out_f32) {
                float* op = p.out_f32 + (long)m * p.ldo + c;
                if ((p.ldo & 3) == 0 && c + 3 < p.N) {
                    *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < p.N) op[r] = o[r];
                }
            }
            if (p.out_rec) {
                bf16x4 h, lo4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned short hb = f2bf(o[r]);
                    h[r] = (short)hb;
                    lo4[r] = (short)f2bf(o[r] - bf2f(hb));
                }
                char* rp = p.out_rec + ((long)(m >> 4) * (p.N >> 5) + (c >> 5)) * 2048 + (((c & 31) >> 3) * 16 + (m & 15)) * 16 + (c & 7) * 2;
                *reinterpret_cast<bf16x4*>(rp) = h;
                *reinterpret_cast<bf16x4*>(rp + 1024) = lo4;
            }
        }
    }
    VP3D_TS(5);
#ifdef SKIMI_ABLATIONS
    __builtin_amdgcn_s_waitcnt(0x0F70);
    VP3D_TS(6);
#endif
}

#ifdef SKIMI_ABLATIONS
}  // namespace skimi
extern "C" int skimi_debug_vp3d_ts(long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(skimi::vp3d_ts), (size_t)n * 8);
}
namespace skimi {
#endif


// ---------------------------------------------------------------------------------------------------------
// vp3d_conv_kernel<CT, RT, TAPS>: the same contraction with every activation fragment loaded ONCE for all
// taps.  A workgroup's load path moves ~45-50 bytes per clock whatever the source (measured: the gather kernel
// above spends its main loop at 49 B/clk per CU with 1152 KB of requests per dilated conv, 960 KB of them the
// same activation rows fetched once per tap), so the lever is bytes per workgroup.  A dilated conv is
//     out[m] = sum_t W_t x[m + t d]  =  sum_t P_t[m + t d],   P_t = W_t x   (one plain product per tap),
// so the row shift can be applied to the OUTPUT: the K loop runs over channel slices only, a slice's RT row
// tiles (the workgroup's R rows + the 2 d halo, unshifted) are loaded once and feed the MFMAs of all TAPS
// accumulator sets, and the shift happens when the K-partials are summed out of LDS.  590 instead of 1152 KB
// per workgroup for the RF-27 dilated convs.  Row splits are per clip (a workgroup never straddles two clips).
struct Vp3dConv {
    const char* wrec;     // [Npad / 16][TAPS * C / 32][2][4][16][8] bf16 (K order: tap-major)
    const char* xrec;     // [ceil(B * Lin / 16)][C / 32][2][4][16][8] bf16
    const float* bias;
    const float* resid;   // fp32 [B * resid_L][C] or null: row b * resid_L + l + resid_off
    float* out_f32;       // [B * Lout][ldo] or null
    char* out_rec;        // fragment-major [ceil(B * Lout / 16)][N / 32]... or null
    int B, N, C, dil, Lin, Lout, resid_L, resid_off, ldo, relu;
    int R, msc;           // output rows per workgroup, row splits per clip
    int xstride, kvalid;  // XF32 mode (expand_conv): floats between consecutive input rows, valid K (the rest reads as zero)
};

// XF32 = 1: the activation operand is the raw fp32 input [B * Lin][xstride] (the 2D keypoints): row l's K vector is
// the `kvalid` consecutive floats starting at row l (= the taps x Cin window of expand_conv, model.py:103), split
// into hi + lo in registers; C is K rounded up to a multiple of 32.
// NW = waves per workgroup: 8 (K split over eight waves, one workgroup per CU: 120 KiB of K-partials for the dilated convs),
// or 4 -- half the partials, so TWO workgroups share a CU.  A launch of 257 .. 512 workgroups (a batch of two clips: the
// reference's own call, a clip and its flipped copy, VideoPose3D/run.py:1070-1083) then runs as ONE round of co-resident
// pairs that pay the in-kernel fixed cost (first loads out, K-partials, store drain: 4-9 us) once, instead of two rounds.
template <int CT, int RT, int TAPS, int XF32 = 0, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void vp3d_conv_kernel(const Vp3dConv p) {
    constexpr int TA = CT / 16, D = 2, NTL = TAPS * TA * RT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);          // [NW waves][NTL tiles][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    int id;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
        const int q = nblk >> 3, r = nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int per = p.B * p.msc;
    const int ct = id / per, rem = id - ct * per;
    const int b = rem / p.msc, rs = rem - b * p.msc;
    const int c0 = ct * CT, l0 = rs * p.R;
    const int SC = p.C >> 5;
    const int cs0 = (int)((long)wave * SC / NW), cs1 = (int)((long)(wave + 1) * SC / NW);
    const int ncs = cs1 - cs0;
    const int rot = ncs > 0 ? ((blockIdx.x >> 3) * 3) % ncs : 0;
    auto slice_at = [&](int i) {
        int j = min(i, ncs - 1) + rot;
        if (j >= ncs) j -= ncs;
        return cs0 + j;
    };
    const char* wb[TA];
    const char* xp[RT];
#pragma unroll
    for (int a = 0; a < TA; ++a) wb[a] = p.wrec + ((long)(c0 / 16 + a) * (TAPS * SC)) * 2048 + lane * 16;
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        if (XF32) {
            const int row = b * p.Lin + min(l0 + 16 * j + li, p.Lout - 1);
            xp[j] = p.xrec + ((long)row * p.xstride + kg * 8) * 4;
        } else {
            const int row = b * p.Lin + min(l0 + 16 * j + li, p.Lin - 1);
            xp[j] = p.xrec + ((long)(row >> 4) * SC) * 2048 + (row & 15) * 16 + kg * 256;
        }
    }
    v8s ah[D][TAPS][TA], al[D][TAPS][TA], bh[D][RT], bl[D][RT];
    f32x4 acc[TAPS][TA][RT];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int j = 0; j < RT; ++j) acc[t][a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load = [&](int d, int cs) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int a = 0; a < TA; ++a) {
                const char* w = wb[a] + (long)(t * SC + cs) * 2048;
                ah[d][t][a] = *reinterpret_cast<const v8s*>(w);
                al[d][t][a] = *reinterpret_cast<const v8s*>(w + 1024);
            }
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            if (XF32) {
                const float* xf = reinterpret_cast<const float*>(xp[j]) + cs * 32;
                const int k0 = cs * 32 + kg * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (k0 + e < p.kvalid) ? xf[e] : 0.f;
                    const unsigned short hb = f2bf(v);
                    bh[d][j][e] = (short)hb;
                    bl[d][j][e] = (short)f2bf(v - bf2f(hb));
                }
            } else {
                const char* x = xp[j] + (long)cs * 2048;
                bh[d][j] = *reinterpret_cast<const v8s*>(x);
                bl[d][j] = *reinterpret_cast<const v8s*>(x + 1024);
            }
        }
    };
    auto mma = [&](int d) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int j = 0; j < RT; ++j)
                    acc[t][a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[d][t][a], bh[d][j], acc[t][a][j], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int j = 0; j < RT; ++j)
                    acc[t][a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[d][t][a], bl[d][j], acc[t][a][j], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int j = 0; j < RT; ++j)
                    acc[t][a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[d][t][a], bh[d][j], acc[t][a][j], 0, 0, 0);
    };
    if (ncs > 0) {   // unconditional (clamped) loads: counted s_waitcnt, see vp3d_mm_kernel
#pragma unroll
        for (int d = 0; d < D; ++d) load(d, slice_at(d));
        int i = 0;
        for (; i + D <= ncs; i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                mma(d);
                load(d, slice_at(i + d + D));
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (i + d < ncs) mma(d);
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int j = 0; j < RT; ++j) red[(wave * NTL + (t * TA + a) * RT + j) * 64 + lane] = acc[t][a][j];
    __syncthreads();
    // out[l0 + m][c0 + 16 a + 4 q .. +3] = sum over taps and the NW K-partials of P_t[m + t d]; the accumulator of
    // row rho, channel quad q of a 16 x 16 tile sits in lane (rho & 15) + 16 q of row tile rho >> 4
    constexpr int Q = CT / 4;
    const int rv = min(p.R, p.Lout - l0);
    for (int o = tid; o < rv * Q; o += 64 * NW) {
        const int m = o / Q, q = o - m * Q;
        const int a = q >> 2, kq = q & 3;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int rho = m + t * p.dil;
            const int idx = ((t * TA + a) * RT + (rho >> 4)) * 64 + (rho & 15) + 16 * kq;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const f32x4 u = red[w * NTL * 64 + idx];
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
            }
        }
        const int l = l0 + m, c = c0 + a * 16 + 4 * kq;
        if (c >= p.N) continue;
        const long mg = (long)b * p.Lout + l;
        float ov[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = v[r] + p.bias[min(c + r, p.N - 1)];
            if (p.relu) x = fmaxf(x, 0.f);
            ov[r] = x;
        }
        if (p.resid) {   // x = res + relu(bn(conv(x))), model.py:129-135
            const float4 rv4 = *reinterpret_cast<const float4*>(p.resid + ((long)b * p.resid_L + l + p.resid_off) * p.C + c);
            ov[0] += rv4.x; ov[1] += rv4.y; ov[2] += rv4.z; ov[3] += rv4.w;
        }
        if (p.out_f32) {
            float* op = p.out_f32 + mg * p.ldo + c;
            if ((p.ldo & 3) == 0 && c + 3 < p.N) {
                *reinterpret_cast<float4*>(op) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < p.N) op[r] = ov[r];
            }
        }
        if (p.out_rec) {
            bf16x4 h, lo4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned short hb = f2bf(ov[r]);
                h[r] = (short)hb;
                lo4[r] = (short)f2bf(ov[r] - bf2f(hb));
            }
            char* rp = p.out_rec + ((mg >> 4) * (p.N >> 5) + (c >> 5)) * 2048 + (((c & 31) >> 3) * 16 + (int)(mg & 15)) * 16 + (c & 7) * 2;
            *reinterpret_cast<bf16x4*>(rp) = h;
            *reinterpret_cast<bf16x4*>(rp + 1024) = lo4;
        }
    }
}

template <int CT, int RT, int TAPS, int XF32 = 0, int NW = 8>
static int launch_conv(const Vp3dConv& p, int grid, hipStream_t st) {
    constexpr int lds = NW * TAPS * (CT / 16) * RT * 1024;
    static_assert(lds <= (NW == 4 ? 80 : 160) * 1024, "K-partials must fit the LDS (two workgroups per CU at NW = 4)");
    if (lds > 64 * 1024)   // > 64 KiB of dynamic LDS needs the opt-in, once per kernel and device
        SKIMI_LDS_OPT_IN((vp3d_conv_kernel<CT, RT, TAPS, XF32, NW>), lds, "vp3d_conv");
    hipLaunchKernelGGL((vp3d_conv_kernel<CT, RT, TAPS, XF32, NW>), dim3(grid), dim3(64 * NW), lds, st, p);
    return SKIMI_OK;
}

// expand_conv on the MFMA kernel: x [B][Lin][Cin] fp32, weights fragment-major [C][Kpad] (Kpad = taps * Cin rounded up to 32)
int vp3d_expand_mfma_launch(const float* x, const void* wfrag, int Kpad, const float* bias, float* out_f32, void* out_rec, int B,
                            int Lin, int Cin, int taps, int C, hipStream_t st) {
    SKIMI_CHECK_ARG(C % 32 == 0 && Kpad % 32 == 0 && Kpad >= taps * Cin, "vp3d_expand: bad shape");
    const int Lout = Lin - taps + 1;
    const int nt = C / 32;
    int msc = (int)std::max<long>(1, 256 / std::max<long>(1, (long)nt * B));
    msc = std::min(msc, Lout);
    while (cdiv(cdiv(Lout, msc), 16) > 5) ++msc;
    Vp3dConv q;
    q.wrec = (const char*)wfrag; q.xrec = (const char*)x; q.bias = bias; q.resid = nullptr;
    q.out_f32 = out_f32; q.out_rec = (char*)out_rec;
    q.B = B; q.N = C; q.C = Kpad; q.dil = 0; q.Lin = Lin; q.Lout = Lout;
    q.resid_L = 0; q.resid_off = 0; q.ldo = C; q.relu = 1;
    q.msc = msc; q.R = (int)cdiv(Lout, msc);
    q.xstride = Cin; q.kvalid = taps * Cin;
    const int rt = (int)cdiv(q.R, 16), grid = nt * B * msc;
    int rc;
    if (rt == 1) rc = launch_conv<32, 1, 1, 1>(q, grid, st);
    else if (rt == 2) rc = launch_conv<32, 2, 1, 1>(q, grid, st);
    else if (rt == 3) rc = launch_conv<32, 3, 1, 1>(q, grid, st);
    else if (rt == 4) rc = launch_conv<32, 4, 1, 1>(q, grid, st);
    else rc = launch_conv<32, 5, 1, 1>(q, grid, st);
    if (rc) return rc;
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

// expand_conv: out[m][c] = relu(bias[c] + sum_k W[c][k] x[b][l .. l + taps - 1][:] (k = tap * Cin + ci)), exact fp32.
// Workgroup = 32 channels x R rows.  The 32 weight rows (odd stride: conflict-free) and the R + taps - 1 input rows
// of the workgroup sit in LDS; the window of output row l is `taps * Cin` consecutive floats starting at row l.
// A thread owns one channel and up to RB rows (register blocking: one weight read feeds RB FMAs).
constexpr int VP3D_EXP_RB = 5;
__global__ __launch_bounds__(256) void vp3d_expand_kernel(const float* __restrict__ x, const float* __restrict__ w, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ out_f32,
                                                          char* __restrict__ out_rec, int M, int C, int Cin, int taps, int Lin,
                                                          int Lout, int R, int ms) {
    extern __shared__ float wsm[];
    constexpr int RB = VP3D_EXP_RB;
    const int K = taps * Cin, stride = K | 1;
    float* xs = wsm + 32 * stride;          // [R rows][K] windows, gathered (a window may cross a clip boundary in m, never in x)
    const int ct = blockIdx.x / ms, rs = blockIdx.x - ct * ms;
    const int c0 = ct * 32, r0 = rs * R;
    const int rend = min(r0 + R, M);
    // staging: 8 threads per row walk its K floats (no per-element division; the loads of a thread are
    // independent, the compiler batches them ahead of the LDS writes)
    {
        const int sr = threadIdx.x >> 3, sk = threadIdx.x & 7;
        const float* wrow = w + (long)(c0 + sr) * ldw;
#pragma unroll 4
        for (int k = sk; k < K; k += 8) wsm[sr * stride + k] = wrow[k];
        for (int r = sr; r < rend - r0; r += 32) {
            const int m = r0 + r, b = m / Lout, l = m - b * Lout;
            const float* xrow = x + ((long)b * Lin + l) * Cin;
#pragma unroll 4
            for (int k = sk; k < K; k += 8) xs[r * K + k] = xrow[k];
        }
    }
    __syncthreads();
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const float* wr = wsm + c * stride;
    const float bc = bias[c0 + c];
    float acc[RB];
    const float* xr[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        acc[j] = 0.f;
        xr[j] = xs + min(rg + 8 * j, max(rend - r0 - 1, 0)) * K;
    }
    for (int k = 0; k < K; ++k) {
        const float wv = wr[k];
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[j] = __builtin_fmaf(wv, xr[j][k], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int m = r0 + rg + 8 * j;
        if (m < rend) {
            const float v = fmaxf(acc[j] + bc, 0.f);
            out_f32[(long)m * C + c0 + c] = v;
            const unsigned short hb = f2bf(v);
            unsigned short* rp = reinterpret_cast<unsigned short*>(out_rec + ((long)(m >> 4) * (C >> 5) + ct) * 2048 +
                                                                   ((c >> 3) * 16 + (m & 15)) * 16) + (c & 7);
            rp[0] = hb;
            rp[512] = f2bf(v - bf2f(hb));
        }
    }
}

int vp3d_expand_launch(const float* x, const float* w, int ldw, const float* bias, float* out_f32, void* out_rec, int B, int Lin,
                       int Cin, int taps, int C, hipStream_t st) {
    SKIMI_CHECK_ARG(C % 32 == 0, "vp3d_expand: channels must be a multiple of 32");
    const int Lout = Lin - taps + 1, M = B * Lout;
    const int nt = C / 32;
    // rows per workgroup <= 8 * RB (8 row groups x RB rows per thread)
    int ms = (int)std::max<long>(1, std::min<long>(cdiv(M, 8), 256 / nt));
    ms = (int)std::max<long>(ms, cdiv(M, 8 * VP3D_EXP_RB));
    const int R = (int)cdiv(M, ms);
    const size_t lds = ((size_t)32 * ((taps * Cin) | 1) + (size_t)R * taps * Cin) * 4;
    SKIMI_CHECK_ARG(lds <= 64 * 1024, "vp3d_expand: receptive window too wide");
    hipLaunchKernelGGL(vp3d_expand_kernel, dim3(nt * ms), dim3(256), lds, st, x, w, ldw, bias, out_f32, (char*)out_rec, M, C, Cin,
                       taps, Lin, Lout, R, ms);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

template <int CT, int RT>
static void launch_mm(const Vp3dMM& p, int grid, hipStream_t st) {
    hipLaunchKernelGGL((vp3d_mm_kernel<CT, RT>), dim3(grid), dim3(512), 0, st, p);
}

// One convolution of the chain.  Npad = rows of the weight records (N rounded up to 16).
int vp3d_mm_launch(const void* wrec, int Npad, const void* xrec, const float* bias, const float* resid, int resid_L,
                   int resid_off, float* out_f32, int ldo, void* out_rec, int B, int Lin, int C, int taps, int dil, int N,
                   int relu, hipStream_t st) {
    const int Lout = Lin - (taps - 1) * dil, M = B * Lout;
    SKIMI_CHECK_ARG(Lout > 0 && C % 32 == 0 && Npad % 16 == 0 && Npad >= N, "vp3d_mm: bad shape");
    SKIMI_CHECK_ARG(out_rec == nullptr || N % 32 == 0, "vp3d_mm: records output needs N % 32 == 0");
    const long K = (long)taps * C, halo = (long)(taps - 1) * dil;
    // First choice: vp3d_conv_kernel (activations loaded once for all taps).  Candidates: taps = 1 -> CT 16 or 32, up to
    // 4 row tiles; taps = 3 -> CT 16, up to 5 row tiles INCLUDING the 2 d halo.  Cost = bytes a workgroup loads x rounds.
    static const int use_conv = getenv("SKIMI_VP3D_TAPREUSE") ? atoi(getenv("SKIMI_VP3D_TAPREUSE")) : 1;
    static const int use_pairs = getenv("SKIMI_VP3D_PAIRS") ? atoi(getenv("SKIMI_VP3D_PAIRS")) : 1;   // 0: 8-wave workgroups only (A/B timing)
    // cost of a candidate = rounds x (elements the workgroups of one CU load in a round + FIXED), FIXED = the in-kernel fixed cost
    // of a round (first loads out, K-partials through LDS, store drain: ~8 us against ~5.5 us for 131 K elements,
    // profiles/r02_vp3d_summary.md).  4-wave workgroups sit two to a CU: a round holds 512 of them and loads twice the elements.
    const double FIXED = 200e3;
    int cv_ct = 0, cv_msc = 0, cv_rt = 0, cv_nw = 8;
    double cv_cost = 1e30;
    if (use_conv && (taps == 1 || taps == 3)) {
        for (int nw : {8, 4}) {
            if (nw == 4 && !use_pairs) continue;
            for (int ct : {16, 32}) {
                if (Npad % ct || (taps == 3 && ct != 16)) continue;
                const int nt = Npad / ct, rtmax = 5;
                if (halo + 1 > 16 * rtmax) continue;
                const long cap = nw == 4 ? 512 : 256;
                int msc = (int)std::max<long>(1, cap / std::max<long>(1, (long)nt * B));
                msc = std::min(msc, Lout);
                while (cdiv(cdiv(Lout, msc) + halo, 16) > rtmax) ++msc;
                const int R = (int)cdiv(Lout, msc), rt = (int)cdiv(R + halo, 16);
                if (nw == 4 && taps * (ct / 16) * rt * 4 > 80) continue;   // K-partials of a pair: 2 x (4 waves x tiles) KiB <= 160
                const long grid = (long)nt * B * msc;
                const double rounds = (double)cdiv(grid, cap);
                const double per_cu = nw == 4 ? std::min<double>(2.0, (double)grid / 256.0) : 1.0;   // workgroups a CU runs side by side
                const double cost = rounds * (std::max(1.0, per_cu) * ((double)ct * K + (double)rt * 16 * C) + FIXED);
                if (cost < cv_cost || (use_pairs == 2 && nw == 4 && cv_nw == 8)) {   // use_pairs 2: pairs wherever they fit (experiments)
                    cv_cost = cost; cv_ct = ct; cv_msc = msc; cv_rt = rt; cv_nw = nw;
                }
            }
        }
    }
    // pick (CT, ms): bytes through one workgroup's L2 port, times the number of rounds over the 256 CUs
    int best_ct = 0, best_ms = 0, best_rt = 0;
    double best = 1e30;
    for (int ct : {16, 32}) {
        if (Npad % ct) continue;
        const int nt = Npad / ct, rtmax = ct == 16 ? 5 : 4;
        int ms = (int)std::max<long>(1, std::min<long>(256 / std::max(1, std::min(nt, 256)), cdiv(M, 1)));
        int R = (int)cdiv(M, ms);
        while (cdiv(R, 16) > rtmax) {
            ++ms;
            R = (int)cdiv(M, ms);
        }
        const double rounds = (double)cdiv((long)nt * ms, 256);
        const double cost = rounds * ((double)ct * K + (double)taps * cdiv(R, 16) * 16 * C + FIXED);   // every tap re-reads its row tiles
        if (cost < best) {
            best = cost; best_ct = ct; best_ms = ms; best_rt = (int)cdiv(R, 16);
        }
    }
    if (cv_ct != 0 && cv_cost <= best) {
        Vp3dConv q;
        q.wrec = (const char*)wrec; q.xrec = (const char*)xrec; q.bias = bias; q.resid = resid;
        q.out_f32 = out_f32; q.out_rec = (char*)out_rec;
        q.B = B; q.N = N; q.C = C; q.dil = dil; q.Lin = Lin; q.Lout = Lout;
        q.resid_L = resid_L; q.resid_off = resid_off; q.ldo = ldo; q.relu = relu;
        q.msc = cv_msc; q.R = (int)cdiv(Lout, cv_msc);
        q.xstride = 0; q.kvalid = 0;
        const int grid = (Npad / cv_ct) * B * cv_msc;
        int rc = SKIMI_ERR_ARG;
#define SKIMI_VP3D_CONV(CT, RT, TAPS)                                                                   \
    if (cv_ct == CT && cv_rt == RT && taps == TAPS)                                                     \
        rc = cv_nw == 4 ? launch_conv<CT, RT, TAPS, 0, 4>(q, grid, st) : launch_conv<CT, RT, TAPS>(q, grid, st); \
    else
        SKIMI_VP3D_CONV(16, 1, 3) SKIMI_VP3D_CONV(16, 2, 3) SKIMI_VP3D_CONV(16, 3, 3) SKIMI_VP3D_CONV(16, 4, 3) SKIMI_VP3D_CONV(16, 5, 3)
        SKIMI_VP3D_CONV(16, 1, 1) SKIMI_VP3D_CONV(16, 2, 1) SKIMI_VP3D_CONV(16, 3, 1) SKIMI_VP3D_CONV(16, 4, 1) SKIMI_VP3D_CONV(16, 5, 1)
        SKIMI_VP3D_CONV(32, 1, 1) SKIMI_VP3D_CONV(32, 2, 1) SKIMI_VP3D_CONV(32, 3, 1) SKIMI_VP3D_CONV(32, 4, 1) SKIMI_VP3D_CONV(32, 5, 1)
        { set_error("vp3d_conv: unsupported tiling %d x %d x %d", cv_ct, cv_rt, taps); }
#undef SKIMI_VP3D_CONV
        if (rc) return rc;
        SKIMI_LAUNCH_CHECK();
        return SKIMI_OK;
    }
    SKIMI_CHECK_ARG(best_ct != 0, "vp3d_mm: no tiling");
    Vp3dMM p;
    p.wrec = (const char*)wrec; p.xrec = (const char*)xrec; p.bias = bias; p.resid = resid;
    p.out_f32 = out_f32; p.out_rec = (char*)out_rec;
    p.M = M; p.N = N; p.C = C; p.taps = taps; p.dil = dil; p.Lin = Lin; p.Lout = Lout;
    p.resid_L = resid_L; p.resid_off = resid_off; p.ldo = ldo; p.relu = relu;
    p.ms = best_ms; p.R = (int)cdiv(M, best_ms);
    p.abl = 0;
#ifdef SKIMI_ABLATIONS
    p.abl = getenv("SKIMI_VP3D_ABL") ? atoi(getenv("SKIMI_VP3D_ABL")) : 0;
#endif
    const int grid = (Npad / best_ct) * best_ms;
#define SKIMI_VP3D_CASE(CT, RT) if (best_ct == CT && best_rt == RT) { launch_mm<CT, RT>(p, grid, st); } else
    SKIMI_VP3D_CASE(16, 1) SKIMI_VP3D_CASE(16, 2) SKIMI_VP3D_CASE(16, 3) SKIMI_VP3D_CASE(16, 4) SKIMI_VP3D_CASE(16, 5)
    SKIMI_VP3D_CASE(32, 1) SKIMI_VP3D_CASE(32, 2) SKIMI_VP3D_CASE(32, 3) SKIMI_VP3D_CASE(32, 4)
    {
        set_error("vp3d_mm: unsupported tiling %d x %d", best_ct, best_rt);
        return SKIMI_ERR_ARG;
    }
#undef SKIMI_VP3D_CASE
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

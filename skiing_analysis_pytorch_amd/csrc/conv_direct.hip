// Direct 3x3 convolution, Cin % 32 == 0 -> 32 output channels, stride 1, pad 1, fp32-accurate
// (bf16 hi/lo planes, 3 MFMAs per product), channels-last.  The DPT heads end in such a layer at the
// full 518 x 518 resolution (dpt_head.py:224-235 scratch.output_conv2[0]: 128 -> 32).
//
// As an implicit-gather GEMM this layer is the worst case of gemm.hip: N = 32 gives no reuse of a
// staged A tile across column tiles, so every input element is fetched nine times through the
// L1 / L2 into registers (measured 6.6 TB/s of gather traffic, 1.5 ms per 8 frames).  Here a
// workgroup owns a 16 x 16 pixel tile: the 18 x 18 halo window of one 32-channel slice is brought
// into LDS once by LDS-DMA and all nine taps read it from there; the 9 x 32 x 32 weight slice is
// staged the same way and shared by the 8 waves.
//   * 8 waves, wave w = output rows 2w, 2w+1 of the tile (32 pixels = one MFMA row block);
//   * per 32-channel slice: 9 taps x 2 k-steps x {a_lo.w_hi, a_hi.w_lo, a_hi.w_hi} = 54
//     v_mfma_f32_32x32x16_bf16 per wave on one 32 x 32 accumulator;
//   * LDS rows are 64 B (32 bf16) per pixel / per (tap, cout); the 16-B chunk is XOR-swizzled on the DMA source and
//     on the ds_read_b128: by (cout >> 2) & 3 for the weights, by ((hx + 2 hy) >> 2) & 3 for the window, whose second
//     tile row is read with its columns rotated by two (conflict-free at the 18-pixel pitch: see the fragment addresses);
//   * double-buffered slices: 2 x (2 planes x 21 KiB window + 2 planes x 18 KiB weights) = 156 KiB;
//   * the epilogue stores straight from the accumulator layout: for every register 32 lanes hold
//     the 32 output channels of one pixel = one full 128-B line.
#include "common.h"
#include "kernels.h"

namespace skimi {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ uint4 g_conv_zero_page[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

constexpr int CD_WIN = 18 * 18;            // halo window pixels
constexpr int CD_IN_PLANE = 21 * 1024;     // 324 px x 64 B rounded up to whole 1-KiB DMA pieces
constexpr int CD_W_PLANE = 9 * 32 * 64;    // 18 KiB
constexpr int CD_BUF = 2 * CD_IN_PLANE + 2 * CD_W_PLANE;

// weights [32][Cin][3][3] fp32 -> [Cin/32][plane hi|lo][tap][cout 32][32] bf16
__global__ void conv_direct_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cin) {
    const long n = 32L * Cin * 9;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c32 = (int)(i % 32);
        const int co = (int)((i / 32) % 32);
        const int tap = (int)((i / 1024) % 9);
        const int cs = (int)(i / 9216);
        const float v = w[((long)co * Cin + cs * 32 + c32) * 9 + tap];
        const unsigned short hi = f2bf(v);
        const unsigned short lo = f2bf(v - bf2f(hi));
        const long base = (long)cs * 2 * 9216 + tap * 1024 + co * 32 + c32;
        out[base] = hi;
        out[base + 9216] = lo;
    }
}

int conv_direct_pack_launch(const float* w, unsigned short* out, int Cin, hipStream_t st) {
    SKIMI_CHECK_ARG(Cin % 32 == 0, "conv_direct: Cin %% 32 != 0");
    hipLaunchKernelGGL(conv_direct_pack_kernel, dim3(64), dim3(256), 0, st, w, out, Cin);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

struct ConvDirectArgs {
    const unsigned short* in_hi;   // [F][H][W][C] bf16 planes of the fp32 input
    const unsigned short* in_lo;
    const unsigned short* w;       // conv_direct_pack_kernel layout
    const float* bias;             // [32] or null
    float* out;                    // [F][H][W][32]
    int F, H, W, C, relu, tiles_x, tiles_y;
    long px_stride;                // elements between pixels in a plane (C for separate planes, 2C for [hi | lo] pixel records)
};

__global__ __launch_bounds__(512, 2) void conv_direct_n32_kernel(const ConvDirectArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int f = bid / p.tiles_y;
    const int y0 = ty * 16, x0 = tx * 16;
    const int nsl = p.C / 32;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_conv_zero_page);

    // ---- DMA plan: window pieces (16 pixels x 64 B each) 0..20 per plane, weight pieces 0..17 per plane;
    //      piece q of a kind goes to wave q % 8 ----
    long in_off[3];     // element offset of this lane's source pixel (channel 0), -1 = outside the image
    int in_piece[3];
    int nin = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = wave + 8 * j;
        in_piece[j] = q;
        in_off[j] = -1;
        if (q < 21) {
            nin = j + 1;
            const int px = 16 * q + (lane >> 2);
            const int hy = px / 18, hx = px - hy * 18;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const int sc = (lane & 3) ^ (((hx + 2 * hy) >> 2) & 3);
            if (px < CD_WIN && y >= 0 && y < p.H && x >= 0 && x < p.W)
                in_off[j] = (((long)f * p.H + y) * p.W + x) * p.px_stride + sc * 8;
        }
    }
    int w_off[3], w_piece[3];
    int nw = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = wave + 8 * j;
        w_piece[j] = q;
        w_off[j] = 0;
        if (q < 18) {
            nw = j + 1;
            const int row = 16 * q + (lane >> 2);
            const int sc = (lane & 3) ^ ((row >> 2) & 3);
            w_off[j] = row * 32 + sc * 8;
        }
    }
    auto stage = [&](int buf, int s) {
        char* base = smem + buf * CD_BUF;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < nin) {   // wave-uniform
                const bool ok = in_off[j] >= 0;
                const unsigned short* sh = ok ? p.in_hi + in_off[j] + s * 32 : zero;
                const unsigned short* sl = ok ? p.in_lo + in_off[j] + s * 32 : zero;
                char* dst = base + in_piece[j] * 1024;
                __builtin_amdgcn_global_load_lds((gbl_void*)sh, (lds_void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)sl, (lds_void*)(dst + CD_IN_PLANE), 16, 0, 0);
            }
        }
        const unsigned short* ws = p.w + (long)s * 2 * 9216;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < nw) {
                char* dst = base + 2 * CD_IN_PLANE + w_piece[j] * 1024;
                __builtin_amdgcn_global_load_lds((gbl_void*)(ws + w_off[j]), (lds_void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(ws + 9216 + w_off[j]), (lds_void*)(dst + CD_W_PLANE), 16, 0, 0);
            }
        }
    };

    // ---- fragment addresses: lane (l31, lh) of wave w reads pixel (2w + (l31 >> 4) + dy, col(l31) + dx) ----
    // The window's row pitch is 18 pixels = 4.5 bank rows of 256 B, so the second tile row of a fragment sits two pixels off
    // the first inside a bank row.  Two measures make every ds_read_b128 conflict-free (checked against the instruction's
    // four 16-lane groups, MI355X_MICROARCH.md §LDS; the first version's key (px >> 2) & 3 was a 2-way conflict on every
    // A read): the chunk key follows u = hx + 2 hy, and the lanes of the second row take their columns rotated by two
    // (col = (l31 + 14) & 15), so that a lane group's sixteen u values are distinct mod 16.
    const int col = ((l31 & 15) + ((l31 >> 4) ? 14 : 0)) & 15;
    int a_off[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const int hy = 2 * wave + (l31 >> 4) + dy, hx = col + dx;
        const int px = hy * 18 + hx;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[tap][ks] = px * 64 + (((2 * ks + lh) ^ (((hx + 2 * hy) >> 2) & 3)) << 4);
    }
    int b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[ks] = l31 * 64 + (((2 * ks + lh) ^ ((l31 >> 2) & 3)) << 4);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    stage(0, 0);
    __syncthreads();   // drains the LDS-DMA (vmcnt(0)) ahead of the barrier
    for (int s = 0; s < nsl; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsl) stage(cur ^ 1, s + 1);
        const char* ih = smem + cur * CD_BUF;
        const char* wh = ih + 2 * CD_IN_PLANE;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(ih + a_off[tap][ks]);
                const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(ih + CD_IN_PLANE + a_off[tap][ks]);
                const bf16x8 w_hi = *reinterpret_cast<const bf16x8*>(wh + tap * 2048 + b_off[ks]);
                const bf16x8 w_lo = *reinterpret_cast<const bf16x8*>(wh + CD_W_PLANE + tap * 2048 + b_off[ks]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, w_hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, w_lo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, w_hi, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: register r of lane (cout = l31, lh) is MFMA row (r&3) + 8 (r>>2) + 4 lh of the wave's 32, i.e. pixel
    //      (row >> 4, column rotated back by two in the second tile row) ----
    const float bs = p.bias ? p.bias[l31] : 0.f;
    const bool interior = y0 + 16 <= p.H && x0 + 16 <= p.W;   // block-uniform
    float* ob = p.out + ((long)f * p.H * p.W) * 32 + l31;
    if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int y = y0 + 2 * wave + (pix >> 4), x = x0 + (((pix & 15) + ((pix >> 4) ? 14 : 0)) & 15);
            float v = acc[r] + bs;
            if (p.relu) v = fmaxf(v, 0.f);
            ob[((long)y * p.W + x) * 32] = v;
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
            const int pix = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int y = y0 + 2 * wave + (pix >> 4), x = x0 + (((pix & 15) + ((pix >> 4) ? 14 : 0)) & 15);
            float v = acc[r] + bs;
            if (p.relu) v = fmaxf(v, 0.f);
            if (y < p.H && x < p.W) ob[((long)y * p.W + x) * 32] = v;
        }
    }
}

int conv_direct_n32_launch(const unsigned short* in_hi, const unsigned short* in_lo, const unsigned short* w_packed,
                           const float* bias, float* out, int F, int H, int W, int C, int relu, hipStream_t st,
                           long px_stride) {
    SKIMI_CHECK_ARG(in_hi && in_lo && w_packed && out, "conv_direct: null buffer");
    SKIMI_CHECK_ARG(C % 32 == 0 && C >= 32 && F > 0 && H > 0 && W > 0, "conv_direct: bad shape");
    SKIMI_CHECK_ARG((((uintptr_t)in_hi | (uintptr_t)in_lo | (uintptr_t)w_packed) & 15) == 0, "conv_direct: 16-B alignment");
    constexpr size_t lds = 2ull * CD_BUF;
    SKIMI_LDS_OPT_IN(conv_direct_n32_kernel, lds, "conv_direct");
    ConvDirectArgs a;
    a.in_hi = in_hi; a.in_lo = in_lo; a.w = w_packed; a.bias = bias; a.out = out;
    a.F = F; a.H = H; a.W = W; a.C = C; a.relu = relu;
    a.px_stride = px_stride > 0 ? px_stride : C;
    a.tiles_x = (int)cdiv(W, 16);
    a.tiles_y = (int)cdiv(H, 16);
    const long nblk = (long)F * a.tiles_x * a.tiles_y;
    SKIMI_CHECK_ARG(nblk < (1l << 31), "conv_direct: grid too large");
    hipLaunchKernelGGL(conv_direct_n32_kernel, dim3((unsigned)nblk), dim3(512), lds, st, a);
    SKIMI_LAUNCH_CHECK();
    return SKIMI_OK;
}

}  // namespace skimi

// LDS-DMA stream micro-benchmark: is the L2 -> LDS operand stream of the 256x256 GEMM tile bound
// by bytes in flight (latency) or by the L2 / fabric bandwidth?  One 512-thread workgroup per CU
// walks the K dimension of its tile exactly as gemm256pp does (quarter tiles of 16 KiB = 128 rows
// x 128 B, two wave-instructions per wave, swizzled source chunks, XCD-patch tile order) into a
// ring of 16-KiB LDS slots, keeping DEPTH quarter tiles in flight (vmcnt(2*(DEPTH-1)) + barrier
// after each issue).  No MFMA, no LDS reads.  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/dma_depth.hip -o tools/_dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
#define VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))

__device__ __forceinline__ void tile_coords(int ntm, int ntn, int& tm, int& tn) {
    const int nblk = ntm * ntn, bid = blockIdx.x, xcd = bid & 7;
    const int q = nblk >> 3, r = nblk & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GM = 8;
    const int per_group = GM * ntn;
    const int group = id / per_group, within = id - group * per_group;
    const int rows = min(GM, ntm - group * GM);
    tn = within / rows;
    tm = group * GM + within - tn * rows;
}

// BAR: 1 = workgroup barrier after each wait (as the GEMM), 0 = waves run free
template <int DEPTH, int BAR, int NW>
__global__ __launch_bounds__(64 * NW) void k(const unsigned short* A, const unsigned short* W, int K, int ntm, int ntn,
                                            int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 10;               // 160 KiB ring
    constexpr int IPW = 16 / NW;         // wave-instructions per wave and quarter tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_coords(ntm, ntn, tm, tn);
    const unsigned short* src[4][IPW];
    int row0[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        row0[j] = (IPW * wave + j) * 8;
        const int row = row0[j] + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        src[0][j] = A + (long)(tm * 256 + row) * K + c * 8;
        src[1][j] = W + (long)(tn * 256 + row) * K + c * 8;
        src[2][j] = W + (long)(tn * 256 + 128 + row) * K + c * 8;
        src[3][j] = A + (long)(tm * 256 + 128 + row) * K + c * 8;
    }
    const int nkt = K / 64;
    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_global_load_lds((gbl_void*)(src[q][j] + kt * 64),
                                                 (lds_void*)(smem + slot * 16384 + row0[j] * 128), 16, 0, 0);
            slot = slot + 1 == NS ? 0 : slot + 1;
            VMCNT(IPW * (DEPTH - 1));
            if (BAR) __builtin_amdgcn_s_barrier();
        }
    }
    VMCNT(0);
    __syncthreads();
    if (sink && tid == 0) sink[blockIdx.x] = *reinterpret_cast<int*>(smem + 64);
}

// the bf16x3 kernels' staging shape: BK = 32 -> 64-B rows, a wave-instruction covers 16 rows x 64 B;
// a 16-KiB piece = one plane of 256 rows; pieces cycle A_hi, A_lo (two buffers), W_hi, W_lo
template <int DEPTH>
__global__ __launch_bounds__(256) void k64(const unsigned short* A, const unsigned short* A2, const unsigned short* W,
                                           const unsigned short* W2, int K, int ntm, int ntn, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 10;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_coords(ntm, ntn, tm, tn);
    const unsigned short* src[4][4];
    int row0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        row0[j] = (4 * wave + j) * 16;
        const int row = row0[j] + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        src[0][j] = A + (long)(tm * 256 + row) * K + c * 8;
        src[1][j] = A2 + (long)(tm * 256 + row) * K + c * 8;
        src[2][j] = W + (long)(tn * 256 + row) * K + c * 8;
        src[3][j] = W2 + (long)(tn * 256 + row) * K + c * 8;
    }
    const int nkt = K / 32;
    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((gbl_void*)(src[q][j] + kt * 32),
                                                 (lds_void*)(smem + slot * 16384 + row0[j] * 64), 16, 0, 0);
            slot = slot + 1 == NS ? 0 : slot + 1;
            VMCNT(4 * (DEPTH - 1));
            __builtin_amdgcn_s_barrier();
        }
    }
    VMCNT(0);
    __syncthreads();
    if (sink && tid == 0) sink[blockIdx.x] = *reinterpret_cast<int*>(smem + 64);
}

// element counts of the two buffers main() allocates: every case is checked against them before it is
// launched (an out-of-range case is a GPU memory fault, not a wrong number)
static const size_t A_ELEMS = (size_t)43968 * 4096, W_ELEMS = (size_t)8192 * 8192;
static bool in_range(size_t a_need, size_t w_need, const char* what) {
    if (a_need <= A_ELEMS && w_need <= W_ELEMS) return true;
    printf("%s: SKIPPED, needs %zu / %zu elements of A / W (have %zu / %zu)\n", what, a_need, w_need, A_ELEMS, W_ELEMS);
    return false;
}

template <int DEPTH>
void run64(const unsigned short* A, const unsigned short* W, int M, int N, int K, int* sink) {
    if (!in_range(2 * (size_t)M * K, 2 * (size_t)N * K, "run64")) return;
    const int ntm = M / 256, ntn = N / 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k64<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned short* A2 = A + (size_t)M * K;
    const unsigned short* W2 = W + (size_t)N * K;
    for (int i = 0; i < 2; ++i) k64<DEPTH><<<ntm * ntn, 256, 163840>>>(A, A2, W, W2, K, ntm, ntn, sink);
    hipEventRecord(e0);
    const int R = 10;
    for (int i = 0; i < R; ++i) k64<DEPTH><<<ntm * ntn, 256, 163840>>>(A, A2, W, W2, K, ntm, ntn, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / R;
    const double bytes = (double)ntm * ntn * (K / 32) * 65536.0;
    const double rounds = (ntm * ntn + 255) / 256;
    printf("64-B rows: M %5d N %5d K %5d depth %d: %8.1f us  %6.1f GB/s per CU  %5.2f TB/s  %.2f us per 64 KiB K-tile\n", M, N, K, DEPTH,
           us, bytes / 256 / us * 1e-3, bytes / us * 1e-6, us / rounds / (K / 32));
    fflush(stdout);
}

template <int DEPTH, int BAR, int NW>
void run(const unsigned short* A, const unsigned short* W, int M, int N, int K, int* sink) {
    if (!in_range((size_t)M * K, (size_t)N * K, "run")) return;
    const int ntm = M / 256, ntn = N / 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<DEPTH, BAR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        163840);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) k<DEPTH, BAR, NW><<<ntm * ntn, 64 * NW, 163840>>>(A, W, K, ntm, ntn, sink);
    hipEventRecord(e0);
    const int R = 10;
    for (int i = 0; i < R; ++i) k<DEPTH, BAR, NW><<<ntm * ntn, 64 * NW, 163840>>>(A, W, K, ntm, ntn, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    const double us = ms * 1e3 / R;
    const double bytes = (double)ntm * ntn * (K / 64) * 65536.0;
    const double rounds = (ntm * ntn + 255) / 256;
    printf("M %5d N %5d K %5d waves %d depth %d (%3d KiB in flight) bar %d: %8.1f us  %6.1f GB/s per CU  %5.2f TB/s  %.2f us per K-tile  %s\n",
           M, N, K, NW, DEPTH, DEPTH * 16, BAR, us, bytes / 256 / us * 1e-3, bytes / us * 1e-6,
           us / rounds / (K / 64), e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
}

// A is 43968 x 4096 and W 8192 x 8192 elements: run64 reads 2 * M * K (hi + lo) of A, run M * K; keep every
// case inside those sizes (an out-of-range case faults the GPU)
int main() {
    const int M = 8192, N = 8192, K = 8192;
    unsigned short *A, *W;
    int* sink;
    hipMalloc(&A, A_ELEMS * 2);
    hipMalloc(&W, W_ELEMS * 2);
    hipMalloc(&sink, 1 << 20);
    hipMemset(A, 0, A_ELEMS * 2);
    hipMemset(W, 0, W_ELEMS * 2);
    run64<2>(A, W, 8192, 2048, 2304, sink);
    run64<4>(A, W, 8192, 2048, 2304, sink);
    run64<6>(A, W, 8192, 2048, 2304, sink);
    run<1, 1, 8>(A, W, M, N, K, sink);
    run<2, 1, 8>(A, W, M, N, K, sink);
    run<3, 1, 8>(A, W, M, N, K, sink);
    run<4, 1, 8>(A, W, M, N, K, sink);
    run<5, 1, 8>(A, W, M, N, K, sink);
    run<6, 1, 8>(A, W, M, N, K, sink);
    run<7, 1, 8>(A, W, M, N, K, sink);
    run<8, 1, 8>(A, W, M, N, K, sink);
    run<9, 1, 8>(A, W, M, N, K, sink);
    run<4, 0, 8>(A, W, M, N, K, sink);
    run<8, 0, 8>(A, W, M, N, K, sink);
    run<4, 1, 4>(A, W, M, N, K, sink);
    run<8, 1, 4>(A, W, M, N, K, sink);
    // the fc1 shape of the bench (M = 43776 rows of 43968, N 4096, K 1024) and fc2 (N 1024, K 4096)
    run<4, 1, 8>(A, W, 43776, 4096, 1024, sink);
    run<8, 1, 8>(A, W, 43776, 4096, 1024, sink);
    run<4, 1, 8>(A, W, 43776, 1024, 4096, sink);
    run<8, 1, 8>(A, W, 43776, 1024, 4096, sink);
    return 0;
}

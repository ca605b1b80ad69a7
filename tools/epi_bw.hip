// Epilogue-burst micro-benchmark: 256x256 fp32 tiles, out = resid * a + b (read 4 B + write 4 B per
// element), one 256-thread workgroup per tile, 16 loads of 16 B in flight per lane.  How much of
// the chip's HBM bandwidth can ONE XCD pull on its own (workgroups of the other XCDs exit at
// once)?  Decides whether staggering the GEMM epilogues XCD by XCD could shorten the store bursts.
//   hipcc --offload-arch=gfx950 -O3 tools/epi_bw.hip -o tools/_epi_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k(const float* __restrict__ resid, float* __restrict__ out, int N, int ntn,
                                        int xcd_mask, int write) {
    if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) return;
    const int t = blockIdx.x >> 3;   // tile index within the XCD's share
    const int tile = t * 8 + (blockIdx.x & 7);
    const int tm = tile / ntn, tn = tile % ntn;
    const int tid = threadIdx.x;
    // 256 rows x 1 KiB: a wave covers 1 row (64 lanes x 16 B); 4 waves -> 4 rows per step, 64 steps
    const float* r = resid + (long)(tm * 256 + (tid >> 6)) * N + tn * 256 + (tid & 63) * 4;
    float* o = out + (long)(tm * 256 + (tid >> 6)) * N + tn * 256 + (tid & 63) * 4;
    for (int s = 0; s < 64; s += 16) {
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float4*>(r + (long)(s + i) * 4 * N);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i].x = v[i].x * 1.0001f + 1.f; v[i].y = v[i].y * 1.0001f + 1.f;
            v[i].z = v[i].z * 1.0001f + 1.f; v[i].w = v[i].w * 1.0001f + 1.f;
            if (write) *reinterpret_cast<float4*>(o + (long)(s + i) * 4 * N) = v[i];
        }
        if (!write && v[0].x == 123.f) o[0] = v[3].y;
    }
}

int main() {
    const int M = 43776, N = 1024;   // 171 x 4 tiles
    float *r, *o;
    hipMalloc(&r, (size_t)M * N * 4);
    hipMalloc(&o, (size_t)M * N * 4);
    hipMemset(r, 0, (size_t)M * N * 4);
    char* trash;
    hipMalloc(&trash, 1u << 30);
    const int ntn = N / 256, ntiles = (M / 256) * ntn;
    for (int write = 1; write >= 0; --write)
        for (int mask : {0xFF, 0x0F, 0x03, 0x01}) {
            int nx = __builtin_popcount(mask);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float tot = 0;
            const int R = 5;
            for (int it = 0; it < R; ++it) {
                hipMemsetAsync(trash, it, 1u << 30);   // flush L2 / Infinity Cache
                hipEventRecord(e0);
                k<<<(ntiles / 8) * 8, 256>>>(r, o, N, ntn, mask, write);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                tot += ms;
            }
            const double us = tot * 1e3 / R;
            const double bytes = (double)(ntiles / 8) * nx * 256 * 256 * 4 * (write ? 2 : 1);
            printf("%s  XCDs active %d: %8.1f us for %6.1f MB -> %6.2f TB/s (%5.1f GB/s per active CU)\n",
                   write ? "read+write" : "read only ", nx, us, bytes * 1e-6, bytes / us * 1e-6, bytes / us * 1e-3 / (32 * nx));
        }
    return 0;
}

// Store-pattern micro-benchmark: how fast can 512-thread workgroups write a bf16 [M][N] matrix
// tile by tile (256x256 tiles) with different lane->address maps?  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o gpurun_out/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) short s8;

template <int PAT>
__global__ __launch_bounds__(512) void k(unsigned short* out, int M, int N, int ntn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int m0 = tm * 256, n0 = tn * 256;
    if (PAT == 0) {   // linear: block writes a contiguous 128-KiB span
        s8 v = {1, 2, 3, 4, 5, 6, 7, 8};
        s8* o = reinterpret_cast<s8*>(out + (long)blockIdx.x * 65536);
        for (int i = 0; i < 16; ++i) o[i * 512 + tid] = v;
    } else if (PAT == 1 || PAT == 4) {   // gemm256 today: wave (wr,wc) owns 128 rows x 64 cols, 8 B/lane, 4 rows/instr
        const int wr = wave >> 2, wc = wave & 3;
        s4 v = {1, 2, 3, 4};
        for (int i = 0; i < 4; ++i)
            for (int it = 0; it < 8; ++it) {
                const int m = m0 + wr * 128 + i * 32 + it * 4 + (lane >> 4);
                s4* d = reinterpret_cast<s4*>(out + (long)m * N + n0 + wc * 64 + 4 * (lane & 15));
                if (PAT == 4) __builtin_nontemporal_store(v, d); else *d = v;
            }
    } else if (PAT == 2) {   // 8 B/lane, one full 512-B tile row per instruction
        s4 v = {1, 2, 3, 4};
        for (int r = 0; r < 32; ++r) {
            const int m = m0 + wave * 32 + r;
            *reinterpret_cast<s4*>(out + (long)m * N + n0 + 4 * lane) = v;
        }
    } else if (PAT == 3 || PAT == 5) {   // 16 B/lane, two 512-B rows per instruction
        s8 v = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + r * 2 + (lane >> 5);
            s8* d = reinterpret_cast<s8*>(out + (long)m * N + n0 + 8 * (lane & 31));
            if (PAT == 5) __builtin_nontemporal_store(v, d); else *d = v;
        }
    } else if (PAT == 6) {   // 16 B/lane, wave owns 64 cols: 8 rows x 128 B per instruction
        const int wr = wave >> 2, wc = wave & 3;
        s8 v = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wr * 128 + i * 8 + (lane >> 3);
            *reinterpret_cast<s8*>(out + (long)m * N + n0 + wc * 64 + 8 * (lane & 7)) = v;
        }
    }
}

template <int PAT>
void run(unsigned short* out, int M, int N) {
    const int ntn = N / 256, ntm = M / 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<PAT><<<ntm * ntn, 512>>>(out, M, N, ntn);
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; ++i) k<PAT><<<ntm * ntn, 512>>>(out, M, N, ntn);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("pat %d  M %d N %d  %.1f us  %.2f TB/s\n", PAT, M, N, ms * 1e3 / R, (double)M * N * 2 / (ms * 1e-3 / R) / 1e12);
}

int main() {
    const int sizes[3][2] = {{4096, 4096}, {16384, 4096}, {43776, 3072}};
    unsigned short* out;
    hipMalloc(&out, (size_t)43776 * 4096 * 2);
    for (auto& s : sizes) {
        run<0>(out, s[0], s[1]); run<1>(out, s[0], s[1]); run<2>(out, s[0], s[1]); run<3>(out, s[0], s[1]);
        run<4>(out, s[0], s[1]); run<5>(out, s[0], s[1]); run<6>(out, s[0], s[1]);
    }
    hipMemset(out, 0, 1024);
    hipDeviceSynchronize();
    return 0;
}

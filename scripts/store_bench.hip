// Store-path micro-benchmark (round 5): how fast can workgroups WRITE fp16 rows the way a conv epilogue does -- 16 B per lane, whole
// rows, every workgroup its own contiguous tile -- and how fast is a plain copy (16 B loads + 16 B stores)?
//   hipcc --offload-arch=gfx950 -O3 scripts/store_bench.hip -o scripts/bin/store_bench && scripts/bin/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: stores only (registers -> global), 1: copy (global -> registers -> global), 2: stores of a tile held in LDS
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int tile16) {
    __shared__ uint4 lds[2048];
    const size_t base = (size_t)blockIdx.x * tile16;
    if (MODE == 2) {
        for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < tile16; i += 256) {
        const size_t j = base + i;
        if (j >= n16) break;
        uint4 v;
        if (MODE == 1) v = src[j];
        else if (MODE == 2) v = lds[i & 2047];
        else v = make_uint4((unsigned)j, 1u, 2u, 3u);
        dst[j] = v;
    }
}

template <int MODE>
void run(const char* name, size_t bytes, int tile_bytes, uint4* a, uint4* b) {
    const size_t n16 = bytes / 16;
    const int tile16 = tile_bytes / 16;
    const unsigned grid = (unsigned)((n16 + tile16 - 1) / tile16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n16, tile16);
    CK(hipEventRecord(e0, 0));
    const int R = 20;
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n16, tile16);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / R;
    printf("%-34s %7.1f MB  tile %3d KB  grid %6u: %7.2f us  %5.2f TB/s written%s\n", name, bytes / 1048576.0, tile_bytes / 1024, grid, us,
           bytes / us / 1e6, MODE == 1 ? " (+ as much read)" : "");
}

int main() {
    uint4 *a, *b;
    const size_t cap = (size_t)512 << 20;
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap));
    CK(hipMemset(a, 1, cap)); CK(hipMemset(b, 0, cap));
    for (size_t mb : {12, 24, 48, 128, 512})
        for (int tile : {8192, 16384, 32768, 65536}) {
            run<0>("stores only (16 B / lane)", mb << 20, tile, a, b);
            run<2>("stores of an LDS-resident tile", mb << 20, tile, a, b);
            run<1>("copy (16 B loads -> 16 B stores)", mb << 20, tile, a, b);
        }
    return 0;
}

// Micro-benchmark (round 5): what does one CU sustain when it fills LDS from L2 / MALL / HBM --
//   mode 0: buffer_load_dwordx4 ... lds (LDS DMA, 1 KB per wave instruction, M0 = destination), counted vmcnt waits
//   mode 1: global_load_dwordx4 to VGPRs + ds_write_b128 (register staging), same ring depth
//   mode 2: global_load_dwordx4 to VGPRs only (no LDS write; the convs.hip way)
// as a function of waves per workgroup, pieces in flight per wave and the footprint the CUs walk.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/bin/dma_bench scripts/dma_bench.hip ; scripts/bin/dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// every wave streams `iters` rounds of P pieces (1 KB each); ring of D rounds in flight
template <int MODE, int P, int D>
__global__ __launch_bounds__(MODE == 0 ? 1024 : 512) void stream_kernel(const char* __restrict__ src, size_t span, int iters, float* sink, int shared_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    // workgroup b walks its own slice of the buffer (or, shared_rows: all workgroups of an XCD walk the same one)
    const size_t wg = shared_rows ? (blockIdx.x & 7) : blockIdx.x;
    const size_t slice = span / (shared_rows ? 8 : gridDim.x);
    const char* base = src + wg * slice;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
    char* ring = smem + w * (D * P * 1024);
    const unsigned per_round = (unsigned)nw * P * 1024u;          // bytes the workgroup moves per round
    const unsigned rounds_in_slice = (unsigned)(slice / per_round);
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 regs[D][P];
    auto issue = [&](int r, int slot) {
        const unsigned off = (unsigned)(r % rounds_in_slice) * per_round + (unsigned)w * P * 1024u + lane * 16u;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            if constexpr (MODE == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(ring + (slot * P + i) * 1024), 16, off + i * 1024u, 0, 0, 0);
            else
                regs[slot][i] = *reinterpret_cast<const uint4*>(base + off + i * 1024u);
        }
    };
    if constexpr (MODE == 0) {
#pragma unroll
        for (int r = 0; r < D - 1; ++r) issue(r, r);
        int slot = D - 1, cs = 0;
        for (int r = 0; r < iters; ++r) {
            wait_vm<(D - 2) * P>();
            issue(r + D - 1, slot);
            slot = slot + 1 == D ? 0 : slot + 1;
            const uint4 v = *reinterpret_cast<const uint4*>(ring + cs * P * 1024 + lane * 16);     // touch what landed
            acc.x ^= v.x;
            cs = cs + 1 == D ? 0 : cs + 1;
        }
        wait_vm<0>();
    } else {
#pragma unroll
        for (int r = 0; r < D - 1; ++r) issue(r, r);
        for (int r0 = 0; r0 < iters; r0 += D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {                 // static slot indices: registers, not scratch
                issue(r0 + k + D - 1, (k + D - 1) % D);
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    if constexpr (MODE == 1) *reinterpret_cast<uint4*>(ring + (k * P + i) * 1024 + lane * 16) = regs[k][i];
                    else { acc.x ^= regs[k][i].x; acc.y ^= regs[k][i].w; }
                }
                if constexpr (MODE == 1) {
                    const uint4 v = *reinterpret_cast<const uint4*>(ring + k * P * 1024 + ((lane * 16 + 512) & 1023));
                    acc.x ^= v.x;
                }
            }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 77u) sink[tid] = 1.f;
}

template <int MODE, int P, int D>
void run(const char* name, const char* buf, size_t span, int waves, int shared_rows, float* sink) {
    const int iters = 512;
    const size_t lds = (size_t)waves * D * P * 1024;
    if (lds > 160 * 1024 || (MODE != 0 && waves > 8)) return;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<MODE, P, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_kernel<MODE, P, D>), dim3(256), dim3(waves * 64), lds, 0, buf, span, iters, sink, shared_rows);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double bytes = 256.0 * waves * P * 1024.0 * iters;
    printf("%-28s waves/CU %2d  pieces/round %d  depth %d  span %6.1f MB%s: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", name, waves, P, D,
           span / 1048576.0, shared_rows ? " (8 shared slices)" : "", best * 1e3, bytes / best / 1e9, bytes / 256 / (best * 1e-3 * 2.4e9));
}

int main() {
    const size_t cap = (size_t)1 << 30;
    char* buf; float* sink;
    CK(hipMalloc(&buf, cap)); CK(hipMemset(buf, 1, cap)); CK(hipMalloc(&sink, 4096 * 4));
    for (size_t span : {(size_t)8 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
        for (int waves : {4, 8, 16}) {
            run<0, 4, 3>("lds-dma", buf, span, waves, 0, sink);
            run<0, 8, 3>("lds-dma", buf, span, waves, 0, sink);
            run<0, 8, 4>("lds-dma", buf, span, waves, 0, sink);
            run<1, 4, 3>("regs + ds_write_b128", buf, span, waves, 0, sink);
            run<1, 8, 2>("regs + ds_write_b128", buf, span, waves, 0, sink);
            run<2, 4, 3>("regs only", buf, span, waves, 0, sink);
            run<2, 8, 3>("regs only", buf, span, waves, 0, sink);
        }
    }
    // all workgroups of an XCD read the same 1 MB slice: L2 hits after the first touch (the conv's shared operands)
    for (int waves : {4, 8, 16}) {
        run<0, 8, 3>("lds-dma", buf, (size_t)8 << 20, waves, 1, sink);
        run<1, 4, 3>("regs + ds_write_b128", buf, (size_t)8 << 20, waves, 1, sink);
        run<2, 8, 3>("regs only", buf, (size_t)8 << 20, waves, 1, sink);
    }
    return 0;
}

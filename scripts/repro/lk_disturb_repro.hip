// Minimal stand-alone reproducer ATTEMPT for the LK disturbance of DESIGN.md section 5b (end of round 2).
// Result on MI355X: it does NOT reproduce -- 0 of 600 victim launches differ for every variant (gpurun_out c66) -- so the
// ingredients below (DPP sequential scans + byte gathers + data-dependent trip counts beside an MFMA / LDS loop with static
// or dynamic LDS and two or four accumulator sets) are not sufficient; kept as the starting point for the next attempt.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lk_repro scripts/repro/lk_disturb_repro.hip && /tmp/lk_repro [variant] [calls]
//
// victim:  one wavefront per "point": gathers 25 bytes of a constant image around a point-dependent position, forms integer
//          products and sums them with the sequential DPP scan of flow.hip (v_add_f32_dpp wave_shr:1), iterates a few times
//          with a data-dependent trip count -- the structure of lk_wave_kernel without the image pyramid.
// hammer:  an MFMA loop with configurable traits, launched over and over on a second stream:
//          variant 0  64c x 64p tile, operands staged through a STATIC LDS tile (like conv_igemm: harmless in the library)
//          variant 1  the same with the tile in DYNAMIC LDS (extern __shared__) and a second accumulator set
//                     (like convs_halo_kernel: disturbs in the library)
//          variant 2  no hammer
// The program reports how many victim launches differ from the idle reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float seq_step(float acc, float v) {
    const int left = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(float, left) + v;
}
__device__ __forceinline__ float seq_sum25(float v) {
    float acc = v;
#pragma unroll
    for (int k = 1; k < 25; ++k) acc = seq_step(acc, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 24));
}

__global__ __launch_bounds__(256) void victim(const unsigned char* __restrict__ img, int w, int h, int n, float* __restrict__ out) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x, pt = gidx >> 6, g = gidx & 63;
    if (pt >= n) return;
    const bool on = g < 25;
    const int wy = on ? g / 5 : 0, wx = on ? g % 5 : 0;
    float x = 8.f + (float)((pt * 37) % (w - 16)), y = 8.f + (float)((pt * 91) % (h - 16));
    float acc = 0.f;
    for (int it = 0; it < 12 + (pt & 7); ++it) {
        const int ix = (int)floorf(x), iy = (int)floorf(y);
        const float fx = x - ix, fy = y - iy;
        const int w00 = __float2int_rn((1.f - fx) * (1.f - fy) * 16384.f), w01 = __float2int_rn(fx * (1.f - fy) * 16384.f);
        const int w10 = __float2int_rn((1.f - fx) * fy * 16384.f), w11 = 16384 - w00 - w01 - w10;
        int v = 0;
        if (on) {
            const unsigned char* r0 = img + (size_t)min(max(iy + wy, 0), h - 2) * w;
            const unsigned char* r1 = r0 + w;
            const int c0 = min(max(ix + wx, 0), w - 2);
            v = (r0[c0] * w00 + r0[c0 + 1] * w01 + r1[c0] * w10 + r1[c0 + 1] * w11 + 256) >> 9;
        }
        const float s1 = seq_sum25((float)(v * (wx - 2))), s2 = seq_sum25((float)(v * (wy - 2)));
        x += s1 * 1e-6f;
        y += s2 * 1e-6f;
        acc += s1 - s2;
        if (fabsf(s1) + fabsf(s2) < 1.f) break;
    }
    if (g == 0) { out[3 * pt] = x; out[3 * pt + 1] = y; out[3 * pt + 2] = acc; }
}

template <bool DYN>
__global__ __launch_bounds__(512) void hammer(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    __shared__ __attribute__((aligned(16))) char stat[DYN ? 16 : 36864];
    _Float16* tile = reinterpret_cast<_Float16*>(DYN ? dyn : stat);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < 2304; i += 512)
            reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(b)[(blockIdx.x * 2304 + i + it * 64) & 0xffff];
        __syncthreads();
        const f16x8 av = *reinterpret_cast<const f16x8*>(a + ((size_t)(wave * 64 + lane) * 8 + it * 512) % 65536);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(tile + ((lane & 31) * 264 + u * 16 + (lane >> 5) * 8));
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(tile + (((lane & 31) + 32) * 264 + u * 16 + (lane >> 5) * 8));
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, acc1, 0, 0, 0);
            if (DYN) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, av, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, av, acc3, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 1, calls = argc > 2 ? atoi(argv[2]) : 400;
    const int w = 960, h = 540, n = 600;
    std::vector<unsigned char> img((size_t)w * h);
    unsigned s = 12345u;
    for (auto& p : img) { s = s * 1664525u + 1013904223u; p = (unsigned char)(s >> 24); }
    unsigned char* d_img; float *d_out, *d_h; _Float16 *d_a, *d_b;
    CHECK(hipMalloc(&d_img, img.size())); CHECK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, sizeof(float) * 3 * n)); CHECK(hipMalloc(&d_h, sizeof(float) * 512 * 1024));
    CHECK(hipMalloc(&d_a, 131072)); CHECK(hipMalloc(&d_b, 131072 * 2));
    CHECK(hipMemset(d_a, 0x3c, 131072)); CHECK(hipMemset(d_b, 0x38, 131072 * 2));
    hipStream_t sv, sh; CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(hammer<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    std::vector<float> ref(3 * n), got(3 * n);
    hipLaunchKernelGGL(victim, dim3((n * 64 + 255) / 256), dim3(256), 0, sv, d_img, w, h, n, d_out);
    CHECK(hipStreamSynchronize(sv)); CHECK(hipMemcpy(ref.data(), d_out, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    int bad_calls = 0, bad_pts = 0;
    for (int c = 0; c < calls; ++c) {
        if (variant == 0) hipLaunchKernelGGL(hammer<false>, dim3(200), dim3(512), 0, sh, d_a, d_b, d_h, 40);
        if (variant == 1) hipLaunchKernelGGL(hammer<true>, dim3(200), dim3(512), 65536, sh, d_a, d_b, d_h, 40);
        hipLaunchKernelGGL(victim, dim3((n * 64 + 255) / 256), dim3(256), 0, sv, d_img, w, h, n, d_out);
        CHECK(hipStreamSynchronize(sv)); CHECK(hipMemcpy(got.data(), d_out, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
        int b = 0;
        for (int i = 0; i < 3 * n; ++i) b += got[i] != ref[i];
        bad_calls += b > 0; bad_pts += b;
        if ((c & 15) == 15) CHECK(hipStreamSynchronize(sh));
    }
    CHECK(hipDeviceSynchronize());
    printf("variant %d: %d of %d victim launches differ from the idle reference (%d values)\n", variant, bad_calls, calls, bad_pts);
    return 0;
}

// Stem convolution: k x k conv over an RGB-like input (<= 4 real channels in the 8-channel NHWC
// pixel), Cout <= 32, + folded BN + activation.  YOLOv4 layer 0 (3x3 s1, 3 -> 32, mish) and OSNet conv1
// (7x7 s2, 3 -> 16, relu).
//
// The generic implicit-GEMM kernel gathers 16 B (8 channels, 5 of them padding) per tap and pixel from
// global memory: 49 gathers per output pixel for the 7x7 stem, ~320 MB of L1/L2 traffic for 50 crops.
// Here a workgroup stages the (15 s + k)^2 x 4-channel input patch of a 16x16 output tile in LDS once
// and builds the MFMA B operand from it: 8 consecutive K = 2 taps x 4 channels = one ds_read2_b64.
//   D[cout][pixel] = sum_k W[cout][k] X[pixel][k],  k = (kh, kw, c4), K = k*k*4 padded to 16
// A (weights, [32][KP] fp16) lives in registers for the whole workgroup.
#include "net.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void stem_conv_kernel(const f16* __restrict__ in, int in_cs, int in_coff,
                                                        f16* __restrict__ out, int out_cs, int out_coff,
                                                        const f16* __restrict__ w, const float* __restrict__ bias,
                                                        int H, int W, int Ho, int Wo, int pad, int act,
                                                        int cout_store) {
    constexpr int TAPS = KS * KS;
    constexpr int KP = (TAPS * 4 + 15) / 16 * 16;
    constexpr int NKS = KP / 16;
    constexpr int PW = 15 * STRIDE + KS, PH = PW;
    __shared__ __attribute__((aligned(16))) uint2 patch[PH * PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16;
    const long n = blockIdx.z;
    const f16* img = in + n * (long)H * W * in_cs + in_coff;
    for (int i = tid; i < PH * PW; i += 256) {
        const int iy = oy0 * STRIDE - pad + i / PW, ix = ox0 * STRIDE - pad + i % PW;
        uint2 v = make_uint2(0u, 0u);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint2*>(img + ((long)iy * W + ix) * in_cs);
        patch[i] = v;
    }
    const int frow = lane & 31, fh = lane >> 5;
    f16x8 afr[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        afr[ks] = *reinterpret_cast<const f16x8*>(w + (long)frow * KP + ks * 16 + fh * 8);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int mt = wave * 2 + t;                 // 8 M-tiles of 32 pixels = 2 tile rows each
        const int py = mt * 2 + (frow >> 4), px = frow & 15;
        const int base = (py * STRIDE) * PW + px * STRIDE;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            // taps 4 ks + 2 fh and + 1 (compile-time ks, fh in {0, 1}); taps >= TAPS have zero weights
            const int ta0 = 4 * ks, ta1 = 4 * ks + 2;
            const int t0 = fh ? ta1 : ta0;
            const int tA = t0 < TAPS ? t0 : TAPS - 1, tB = t0 + 1 < TAPS ? t0 + 1 : TAPS - 1;
            const uint2 a = patch[base + (tA / KS) * PW + tA % KS];
            const uint2 b = patch[base + (tB / KS) * PW + tB % KS];
            uint4 v = make_uint4(a.x, a.y, b.x, b.y);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks], *reinterpret_cast<f16x8*>(&v), acc, 0, 0, 0);
        }
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < Ho && ox < Wo) {
            f16* dst = out + ((n * Ho + oy) * (long)Wo + ox) * out_cs + out_coff;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = 8 * g + 4 * fh;
                if (co < cout_store) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + co);
                    f16x4 o;
                    float a4[4] = {acc[4 * g + 0] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z,
                                   acc[4 * g + 3] + bv.w};
                    apply_act_n<4>(a4, act);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16)a4[e];
                    *reinterpret_cast<f16x4*>(dst + co) = o;
                }
            }
        }
    }
}

}  // namespace

// w: fp16 [32][ceil16(k*k*4)], K order (kh, kw, c) with c < 4; bias f32[32]
int launch_stemconv(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, const f16* w,
                    const float* bias, int N, int H, int W, int Ho, int Wo, int k, int stride, int pad, int cout,
                    int act, hipStream_t s) {
    FM_CHECK_ARG(cout >= 1 && cout <= 32 && in_cs % 4 == 0 && in_coff % 4 == 0 && out_cs % 4 == 0 && out_coff % 4 == 0);
    const dim3 grid((Wo + 15) / 16, (Ho + 15) / 16, N), block(256);
    const int cs = (cout + 7) & ~7;
    if (k == 3 && stride == 1)
        hipLaunchKernelGGL((stem_conv_kernel<3, 1>), grid, block, 0, s, in, in_cs, in_coff, out, out_cs, out_coff, w,
                           bias, H, W, Ho, Wo, pad, act, cs);
    else if (k == 3 && stride == 2)
        hipLaunchKernelGGL((stem_conv_kernel<3, 2>), grid, block, 0, s, in, in_cs, in_coff, out, out_cs, out_coff, w,
                           bias, H, W, Ho, Wo, pad, act, cs);
    else if (k == 7 && stride == 2)
        hipLaunchKernelGGL((stem_conv_kernel<7, 2>), grid, block, 0, s, in, in_cs, in_coff, out, out_cs, out_coff, w,
                           bias, H, W, Ho, Wo, pad, act, cs);
    else {
        fm_set_error("stem conv: unsupported k=%d stride=%d", k, stride);
        return FM_ERR_ARG;
    }
    FM_HIP(hipGetLastError());
    return 0;
}

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
#include "pixel_source.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// SRC: where the patch comes from (pixel_source.h StemSrc): 0 the fp16 NHWC tensor `in`; 1 the detector's letterboxed
// resize of the u8 frame, 2 the extractor's crops -- computed on the fly with the front-end kernels' own functions, so
// the network input tensor is neither written nor read and the front-end launch disappears (round 6: the detector's
// preprocess kernel was 15 us between two passes of the detector stream, whose period is the step; the crop kernel
// wrote and the stem re-read 26 MB per 50 crops).  A patch position is computed by every tile it lies in (1.27x / 1.34x
// of the pixels).
template <int KS, int STRIDE, int SRC>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemSrc src, const f16* __restrict__ in, int in_cs, int in_coff,
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
    if (SRC != 0 && src.zero4 && tid < 4 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) src.zero4[tid] = 0;
    // crops: the normalisation (v / 255 - mean) / std costs two float64 divisions per channel; it is a function of the uint8
    // value alone, so the workgroup tabulates its 3 x 256 results once (crop_normalise: the values of the front-end kernel)
    __shared__ f16 norm_lut[SRC == 2 ? 3 * 256 : 2];
    if constexpr (SRC == 2) {
#pragma unroll
        for (int i = 0; i < 3; ++i) norm_lut[i * 256 + tid] = crop_normalise(tid, i);
        __syncthreads();
    }
    // (unrolled, loads unconditional on clamped coordinates, zeroing afterwards: all trips to memory of a thread's patch
    // positions are in flight together -- a load inside a divergent branch waits on the spot)
    constexpr int NIT = (PH * PW + 255) / 256;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256, ic = min(i, PH * PW - 1);
        const int iy = oy0 * STRIDE - pad + ic / PW, ix = ox0 * STRIDE - pad + ic % PW;
        const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
        uint2 v;
        if constexpr (SRC == 0) {
            v = *reinterpret_cast<const uint2*>(img + ((long)cy * W + cx) * in_cs);
        } else if constexpr (SRC == 1) {
            float rgb[3];
            det_input_pixel(src.frame, src.fw, src.fh, cx, cy, src.roi_x, src.roi_y, src.roi_w, src.roi_h, rgb);
            union { f16 h[4]; uint2 u; } pk;
            pk.h[0] = (f16)rgb[0]; pk.h[1] = (f16)rgb[1]; pk.h[2] = (f16)rgb[2]; pk.h[3] = (f16)0.f;
            v = pk.u;
        } else {
            int bgr[3];
            union { f16 h[4]; uint2 u; } pk;
            pk.u = make_uint2(0u, 0u);
            if (crop_u8_pixel(src.frame, src.fw, src.fh, src.boxes + n * 4, cx, cy, W, H, bgr)) {
                pk.h[0] = norm_lut[0 * 256 + bgr[2]]; pk.h[1] = norm_lut[1 * 256 + bgr[1]]; pk.h[2] = norm_lut[2 * 256 + bgr[0]];
            }
            v = pk.u;
        }
        if (!inside) v = make_uint2(0u, 0u);
        if (i < PH * PW) patch[i] = v;
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

template <int SRC>
static int launch_stem_kind(const StemSrc& src, const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff,
                            const f16* w, const float* bias, int N, int H, int W, int Ho, int Wo, int k, int stride, int pad,
                            int cout, int act, hipStream_t s) {
    const dim3 grid((Wo + 15) / 16, (Ho + 15) / 16, N), block(256);
    const int cs = (cout + 7) & ~7;
    if (k == 3 && stride == 1)
        hipLaunchKernelGGL((stem_conv_kernel<3, 1, SRC>), grid, block, 0, s, src, in, in_cs, in_coff, out, out_cs, out_coff, w,
                           bias, H, W, Ho, Wo, pad, act, cs);
    else if (k == 3 && stride == 2)
        hipLaunchKernelGGL((stem_conv_kernel<3, 2, SRC>), grid, block, 0, s, src, in, in_cs, in_coff, out, out_cs, out_coff, w,
                           bias, H, W, Ho, Wo, pad, act, cs);
    else if (k == 7 && stride == 2)
        hipLaunchKernelGGL((stem_conv_kernel<7, 2, SRC>), grid, block, 0, s, src, in, in_cs, in_coff, out, out_cs, out_coff, w,
                           bias, H, W, Ho, Wo, pad, act, cs);
    else {
        fm_set_error("stem conv: unsupported k=%d stride=%d", k, stride);
        return FM_ERR_ARG;
    }
    FM_HIP(hipGetLastError());
    return 0;
}

// w: fp16 [32][ceil16(k*k*4)], K order (kh, kw, c) with c < 4; bias f32[32]
int launch_stemconv_src(const StemSrc& src, const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff,
                        const f16* w, const float* bias, int N, int H, int W, int Ho, int Wo, int k, int stride, int pad,
                        int cout, int act, hipStream_t s) {
    FM_CHECK_ARG(cout >= 1 && cout <= 32 && in_cs % 4 == 0 && in_coff % 4 == 0 && out_cs % 4 == 0 && out_coff % 4 == 0);
    FM_CHECK_ARG(src.kind == 0 || (src.frame && src.fw > 0 && src.fh > 0 && (src.kind == 1 || (src.kind == 2 && src.boxes))));
    if (src.kind == 1) return launch_stem_kind<1>(src, in, in_cs, in_coff, out, out_cs, out_coff, w, bias, N, H, W, Ho, Wo, k, stride, pad, cout, act, s);
    if (src.kind == 2) return launch_stem_kind<2>(src, in, in_cs, in_coff, out, out_cs, out_coff, w, bias, N, H, W, Ho, Wo, k, stride, pad, cout, act, s);
    return launch_stem_kind<0>(src, in, in_cs, in_coff, out, out_cs, out_coff, w, bias, N, H, W, Ho, Wo, k, stride, pad, cout, act, s);
}

int launch_stemconv(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, const f16* w,
                    const float* bias, int N, int H, int W, int Ho, int Wo, int k, int stride, int pad, int cout,
                    int act, hipStream_t s) {
    return launch_stemconv_src(StemSrc{}, in, in_cs, in_coff, out, out_cs, out_coff, w, bias, N, H, W, Ho, Wo, k, stride, pad,
                               cout, act, s);
}

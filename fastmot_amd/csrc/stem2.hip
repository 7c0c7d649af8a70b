// The first two layers of a Darknet YOLO backbone in one launch (FM_OP_STEM2, round 6):
//   y = act2(conv3x3 stride 2 (act1(conv3x3 stride 1 (x, 3 -> 32)), 32 -> COUT))
// (yolov4.cfg layers 0 and 1; graph semantics scripts/yolo2onnx.py:558-705).  As two launches the stem wrote 27 MB of
// 608 x 608 x 32 activations and the stride-2 conv read them straight back: 32 us for two layers whose algorithmic
// traffic is the frame in and 12 MB out (profiles/r05_store_path_microbench.txt puts their practical
// memory floors at 6.5 + 7.3 us).  Here a workgroup (4 waves) owns an 8 x 8 tile of OUTPUT pixels:
//   patch    the 19 x 19 input pixels under the tile (4-channel fp16 pixels, or -- StemSrc kind 1 -- computed from the u8
//            frame with the detector's own resize function, pixel_source.h) -> LDS;
//   phase 1  the stem on the 17 x 17 positions the stride-2 conv reads: the MFMA sequence of stemconv.hip (same operand
//            fragments, same K order: the fp16 values are the ones the stem layer would have stored), + bias + act1,
//            zero outside the image (= the second conv's padding) -> LDS tile [289][32 + 8];
//   phase 2  the stride-2 3x3 conv as 9 taps x 2 MFMA steps per wave (one 32-cout x 32-pixel accumulator each), B
//            fragments from the LDS tile (row stride 80 B: conflict-free ds_read_b128), A fragments = the wave's 18 KB of
//            weights in fragment order, all requested at kernel entry; + bias + act2, NHWC stores.
// The halo makes phase 1 do 289 / 256 = 1.13x of the stem's work.
#include "pixel_source.h"

namespace {

// C3 > 0: a pointwise conv COUT -> C3 (+ bias + act3) over the tile follows in the same launch (the first CSP stage's merged
// 1x1 conv: the stride-2 conv's output is then never stored either).  w3 in fragment order [C3 / 32][COUT / 16][lane][8].
template <int COUT, int SRC, int C3>
__global__ __launch_bounds__(256) void stem2_kernel(const StemSrc src, const f16* __restrict__ in, int in_cs,
                                                    f16* __restrict__ out, int out_cs, int out_coff,
                                                    const f16* __restrict__ w1, const float* __restrict__ b1,
                                                    const f16* __restrict__ w2, const float* __restrict__ b2,
                                                    const f16* __restrict__ w3, const float* __restrict__ b3,
                                                    int H, int W, int Ho, int Wo, int act1, int act2, int act3) {
    constexpr int TO = 8, MW = 2 * TO + 1, NPOS = MW * MW, PW = MW + 2, M = 32, S = M + 8;
    constexpr int KP1 = 48, NKS1 = 3, TAPS = 9;
    constexpr int NCT = COUT / 32, NPT = TO * TO / 32, NT2 = NCT * NPT / 4;      // phase-2 tiles per wave
    constexpr int NP1 = (NPOS + 31) / 32;                                         // phase-1 position tiles
    static_assert(NCT * NPT % 4 == 0, "tile / wave split");
    __shared__ __attribute__((aligned(16))) uint2 patch[PW * PW];
    __shared__ __attribute__((aligned(16))) f16 mid[NPOS * S];
    constexpr int SD = COUT + 8;                                  // row stride of the phase-3 tile (conflict-free 16-byte reads)
    __shared__ __attribute__((aligned(16))) f16 dtile[C3 > 0 ? TO * TO * SD : 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const int ox0 = blockIdx.x * TO, oy0 = blockIdx.y * TO;
    const long n = blockIdx.z;
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;        // stem position of mid[0][0] (second conv: pad 1, stride 2)
    if (SRC != 0 && src.zero4 && tid < 4 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) src.zero4[tid] = 0;

    // ---- phase-2 weights of this wave's accumulator tiles: taps 0..2 are requested first and consumed last, taps 3..8 are
    // requested when phase 2 begins (all 18 fragments up front held 72 VGPRs through phase 1: 144 in all, three workgroups
    // per CU)
    constexpr int QA = 6;
    f16x8 fa2[NT2][18];
    const f16* wr2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int ct = (wave * NT2 + t) % NCT;
        wr2[t] = w2 + ((long)ct * 18 * 64 + lane) * 8;
#pragma unroll
        for (int q = 0; q < QA; ++q) fa2[t][q] = *reinterpret_cast<const f16x8*>(wr2[t] + q * 512);
    }
    f16x8 fa1[NKS1];
#pragma unroll
    for (int ks = 0; ks < NKS1; ++ks) fa1[ks] = *reinterpret_cast<const f16x8*>(w1 + (long)frow * KP1 + ks * 16 + fh * 8);

    // ---- input patch (stem: pad 1, stride 1): pixel (sy0 - 1 + i / PW, sx0 - 1 + i % PW)
    const f16* img = in + n * (long)H * W * in_cs;
    constexpr int NIT = (PW * PW + 255) / 256;      // (unrolled, unconditional clamped loads: see stemconv.hip)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256, ic = min(i, PW * PW - 1);
        const int iy = sy0 - 1 + ic / PW, ix = sx0 - 1 + ic % PW;
        const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
        uint2 v;
        if constexpr (SRC == 0) {
            v = *reinterpret_cast<const uint2*>(img + ((long)cy * W + cx) * in_cs);
        } else {
            float rgb[3];
            det_input_pixel(src.frame, src.fw, src.fh, cx, cy, src.roi_x, src.roi_y, src.roi_w, src.roi_h, rgb);
            union { f16 h[4]; uint2 u; } pk;
            pk.h[0] = (f16)rgb[0]; pk.h[1] = (f16)rgb[1]; pk.h[2] = (f16)rgb[2]; pk.h[3] = (f16)0.f;
            v = pk.u;
        }
        if (!inside) v = make_uint2(0u, 0u);
        if (i < PW * PW) patch[i] = v;
    }
    __syncthreads();

    // ---- phase 1: the stem on the MW x MW positions (stemconv.hip's operand construction: 2 taps x 4 channels per K half)
    for (int pt = wave; pt < NP1; pt += 4) {
        const int pos = pt * 32 + frow, pc = min(pos, NPOS - 1);
        const int py = pc / MW, px = pc - py * MW;
        const int base = py * PW + px;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS1; ++ks) {
            const int t0 = fh ? 4 * ks + 2 : 4 * ks;
            const int tA = t0 < TAPS ? t0 : TAPS - 1, tB = t0 + 1 < TAPS ? t0 + 1 : TAPS - 1;
            const uint2 a = patch[base + (tA / 3) * PW + tA % 3];
            const uint2 b = patch[base + (tB / 3) * PW + tB % 3];
            uint4 v = make_uint4(a.x, a.y, b.x, b.y);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[ks], *reinterpret_cast<f16x8*>(&v), acc, 0, 0, 0);
        }
        if (pos < NPOS) {
            const int sy = sy0 + py, sx = sx0 + px;
            const bool inside = sy >= 0 && sy < H && sx >= 0 && sx < W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = 8 * g + 4 * fh;
                const float4 bv = *reinterpret_cast<const float4*>(b1 + co);
                float a4[4] = {acc[4 * g + 0] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z, acc[4 * g + 3] + bv.w};
                apply_act_n<4>(a4, act1);
                union { f16 h[4]; uint2 u; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk.h[e] = (f16)(inside ? a4[e] : 0.f);
                *reinterpret_cast<uint2*>(&mid[pos * S + co]) = pk.u;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: 3x3 stride 2 over the LDS tile
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int q = QA; q < 18; ++q) fa2[t][q] = *reinterpret_cast<const f16x8*>(wr2[t] + q * 512);
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int tile = wave * NT2 + t, ct = tile % NCT, ptile = tile / NCT;
        const int pix = ptile * 32 + frow, py = pix / TO, px = pix % TO;
        const f16* bsrc = mid + ((2 * py) * MW + 2 * px) * S + fh * 8;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const f16* bt = bsrc + ((tap / 3) * MW + tap % 3) * S;
            const f16x8 fb0 = *reinterpret_cast<const f16x8*>(bt), fb1 = *reinterpret_cast<const f16x8*>(bt + 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa2[t][2 * tap], fb0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa2[t][2 * tap + 1], fb1, acc, 0, 0, 0);
        }
        const int oy = oy0 + py, ox = ox0 + px;
        if constexpr (C3 == 0) {
            if (oy < Ho && ox < Wo) {
                f16* dst = out + ((n * Ho + oy) * (long)Wo + ox) * out_cs + out_coff + ct * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 8 * g + 4 * fh;
                    const float4 bv = *reinterpret_cast<const float4*>(b2 + ct * 32 + co);
                    float a4[4] = {acc[4 * g + 0] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z, acc[4 * g + 3] + bv.w};
                    apply_act_n<4>(a4, act2);
                    f16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16)a4[e];
                    *reinterpret_cast<f16x4*>(dst + co) = o;
                }
            }
        } else {
            // the tile's fp16 values -- what the layer would have stored -- go to an LDS tile of their own for the pointwise
            // conv (`mid` is still being read by the other waves' phase 2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = 8 * g + 4 * fh;
                const float4 bv = *reinterpret_cast<const float4*>(b2 + ct * 32 + co);
                float a4[4] = {acc[4 * g + 0] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z, acc[4 * g + 3] + bv.w};
                apply_act_n<4>(a4, act2);
                union { f16 h[4]; uint2 u; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk.h[e] = (f16)a4[e];
                *reinterpret_cast<uint2*>(&dtile[pix * SD + ct * 32 + co]) = pk.u;
            }
        }
    }
    if constexpr (C3 > 0) {
        // ---- phase 3: pointwise conv over the 64-pixel tile: (C3 / 32) cout tiles x 2 pixel tiles over the 4 waves
        constexpr int NC3 = C3 / 32, NT3 = NC3 * NPT / 4, KS3 = COUT / 16;
        f16x8 fa3[NT3][KS3];
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            const int ct = (wave * NT3 + t) % NC3;
#pragma unroll
            for (int q = 0; q < KS3; ++q) fa3[t][q] = *reinterpret_cast<const f16x8*>(w3 + (((long)ct * KS3 + q) * 64 + lane) * 8);
        }
        __syncthreads();                                        // dtile complete
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
            const int tile = wave * NT3 + t, ct = tile % NC3, ptile = tile / NC3;
            const int pix = ptile * 32 + frow, py = pix / TO, px = pix % TO;
            const f16* bsrc = dtile + pix * SD + fh * 8;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < KS3; ++q)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa3[t][q], *reinterpret_cast<const f16x8*>(bsrc + q * 16), acc, 0, 0, 0);
            const int oy = oy0 + py, ox = ox0 + px;
            if (oy < Ho && ox < Wo) {
                f16* dst = out + ((n * Ho + oy) * (long)Wo + ox) * out_cs + out_coff + ct * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 8 * g + 4 * fh;
                    const float4 bv = *reinterpret_cast<const float4*>(b3 + ct * 32 + co);
                    float a4[4] = {acc[4 * g + 0] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z, acc[4 * g + 3] + bv.w};
                    apply_act_n<4>(a4, act3);
                    f16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16)a4[e];
                    *reinterpret_cast<f16x4*>(dst + co) = o;
                }
            }
        }
    }
}

template <int COUT, int C3>
int launch_cout(const StemSrc& src, const f16* in, int in_cs, f16* out, int out_cs, int out_coff, const f16* w1,
                const float* b1, const f16* w2, const float* b2, const f16* w3, const float* b3, int N, int H, int W, int Ho,
                int Wo, int act1, int act2, int act3, hipStream_t s) {
    const dim3 grid((Wo + 7) / 8, (Ho + 7) / 8, N), block(256);
    if (src.kind == 1)
        hipLaunchKernelGGL((stem2_kernel<COUT, 1, C3>), grid, block, 0, s, src, in, in_cs, out, out_cs, out_coff, w1, b1, w2, b2,
                           w3, b3, H, W, Ho, Wo, act1, act2, act3);
    else
        hipLaunchKernelGGL((stem2_kernel<COUT, 0, C3>), grid, block, 0, s, src, in, in_cs, out, out_cs, out_coff, w1, b1, w2, b2,
                           w3, b3, H, W, Ho, Wo, act1, act2, act3);
    FM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

bool stem2_supported(int mid, int cout) { return mid == 32 && (cout == 64 || cout == 128); }
bool stem3_supported(int cout2, int cout3) { return cout2 == 64 && (cout3 == 64 || cout3 == 128); }

// w1: the stem's weights as for FM_OP_STEMCONV ([32][48] fp16, K order (kh, kw, c4)), b1 f32[32]; w2: the second conv's in
// MFMA A-fragment order [cout / 32][288 / 16][lane][8] (K order (kh, kw, cin); Graph._pack_frag), b2 f32[cout]; cout3 > 0: a
// pointwise conv cout -> cout3 follows (w3 fragment order [cout3 / 32][cout / 16][lane][8], b3 f32[cout3]) and `out` has cout3 channels
int launch_stem2(const StemSrc& src, const f16* in, int in_cs, f16* out, int out_cs, int out_coff, const f16* w1,
                 const float* b1, const f16* w2, const float* b2, int N, int H, int W, int Ho, int Wo, int cout, int act1,
                 int act2, hipStream_t s, int cout3, const f16* w3, const float* b3, int act3) {
    FM_CHECK_ARG(stem2_supported(32, cout) && in_cs % 4 == 0 && out_cs % 4 == 0 && out_coff % 4 == 0);
    FM_CHECK_ARG(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
    FM_CHECK_ARG(src.kind == 0 || (src.kind == 1 && src.frame && src.fw > 0 && src.fh > 0));
    FM_CHECK_ARG(cout3 == 0 || (stem3_supported(cout, cout3) && w3 && b3));
#define STEM2_ARGS src, in, in_cs, out, out_cs, out_coff, w1, b1, w2, b2, w3, b3, N, H, W, Ho, Wo, act1, act2, act3, s
    if (cout3 == 64) return launch_cout<64, 64>(STEM2_ARGS);
    if (cout3 == 128) return launch_cout<64, 128>(STEM2_ARGS);
    if (cout == 64) return launch_cout<64, 0>(STEM2_ARGS);
    return launch_cout<128, 0>(STEM2_ARGS);
#undef STEM2_ARGS
}

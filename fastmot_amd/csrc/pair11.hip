// Two consecutive 1x1 convolutions around a concat in one launch (FM_OP_PAIR11, round 6):
//   t = act1(W1 xa + b1)                       (64 -> 64 channels)
//   y = act2(W2 [t | xc] + b2)                 (64 + 64 -> COUT channels, COUT in {64, 128})
// -- the tail of a CSPDarknet stage on the large maps (yolov4.cfg: the residual branch's last 1x1 conv, [route] with the
// stage's other branch, the stage's closing 1x1 conv; graph semantics scripts/yolo2onnx.py:558-705,782-803).  Both layers
// are bound by their memory traffic there (304 x 304 x 64: 23.7 + 35.5 MB in two launches, 10.8 + 10.3 us against a
// practical floor of 2.2 us + bytes / 6.9 TB/s each); fused, t is never stored: 35.5 MB, one launch.
//
// A workgroup (4 waves) owns 128 consecutive pixels.  xa and xc tiles -> LDS (coalesced 16-byte loads, rows padded by
// 16 B: conflict-free ds_read_b128 fragment reads); wave w multiplies pixel tile w (32 pixels) by all of W1 (fragment-
// ordered weights straight from L2, 8 KB), writes act1(.) as fp16 -- exactly what the unfused layer would have stored --
// over the xa tile; second conv: K = 128 = [t | xc] in the concat's channel order, one 32-cout tile at a time (8 weight
// fragments in registers, B fragments re-read from LDS); bias + act2, NHWC stores.  K order and MFMA sequence per output
// element are those of the unfused layers (16-wide steps, ascending), so the results are bit-identical to them.
#include "net.h"

namespace {

// PT: 32-pixel tiles per workgroup (4: every wave owns a tile and all channels; 2: two waves share a tile and split the
// mid / cout tiles -- twice the workgroups for maps whose 128-pixel tiles would not fill the chip, 152 x 152: 180 -> 361)
template <int COUT, int PT>
__global__ __launch_bounds__(256) void pair11_kernel(const f16* __restrict__ xa, int xa_cs, int xa_coff,
                                                     const f16* __restrict__ xc, int xc_cs, int xc_coff,
                                                     f16* __restrict__ out, int out_cs, int out_coff,
                                                     const f16* __restrict__ w1, const float* __restrict__ b1,
                                                     const f16* __restrict__ w2, const float* __restrict__ b2,
                                                     long P, int act1, int act2) {
    constexpr int C = 64, S = C + 8, BN = 32 * PT, NCT = COUT / 32, WPT = 4 / PT, NM = 2 / WPT, NC = NCT / WPT;
    constexpr int NLD = BN * 8 / 256;                            // 16-byte pieces per thread and tile
    __shared__ __attribute__((aligned(16))) f16 ta[BN * S];     // xa tile, then t
    __shared__ __attribute__((aligned(16))) f16 tc[BN * S];     // xc tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const int pt = wave % PT, part = wave / PT;                  // pixel tile, which share of the mid / cout tiles
    const long p0 = (long)blockIdx.x * BN;

    // ---- W1 fragments (this wave's mid tiles x 4 K steps) and the tiles' loads: everything requested before anything is used
    f16x8 fa1[NM][4];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            fa1[m][q] = *reinterpret_cast<const f16x8*>(w1 + (((long)(part * NM + m) * 4 + q) * 64 + lane) * 8);
    f16x8 va[NLD], vc[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * 256, row = e >> 3, c8 = (e & 7) * 8;       // 8 x 16 B per pixel row
        const long pix = min(p0 + row, P - 1);
        va[i] = *reinterpret_cast<const f16x8*>(xa + pix * xa_cs + xa_coff + c8);
        vc[i] = *reinterpret_cast<const f16x8*>(xc + pix * xc_cs + xc_coff + c8);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * 256, row = e >> 3, c8 = (e & 7) * 8;
        *reinterpret_cast<f16x8*>(&ta[row * S + c8]) = va[i];
        *reinterpret_cast<f16x8*>(&tc[row * S + c8]) = vc[i];
    }
    float4 bias1[NM][4];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias1[m][g] = *reinterpret_cast<const float4*>(b1 + (part * NM + m) * 32 + g * 8 + fh * 4);
    __syncthreads();

    // ---- first conv on this wave's 32 pixels (its share of the 64 mid channels)
    const f16* brow = ta + (pt * 32 + frow) * S + fh * 8;
    f32x16 acc1[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[m][r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f16x8 fb = *reinterpret_cast<const f16x8*>(brow + q * 16);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc1[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[m][q], fb, acc1[m], 0, 0, 0);
    }
    // t overwrites the xa tile: with one wave per pixel tile a wave touches only its own 32 rows (LDS is in order per wave);
    // two waves sharing a tile meet at a barrier on either side of the overwrite
    if constexpr (WPT > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = bias1[m][g];
            float a4[4] = {acc1[m][4 * g + 0] + bv.x, acc1[m][4 * g + 1] + bv.y, acc1[m][4 * g + 2] + bv.z,
                           acc1[m][4 * g + 3] + bv.w};
            apply_act_n<4>(a4, act1);
            union { f16 h[4]; uint2 u; } pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk.h[e] = (f16)a4[e];
            *reinterpret_cast<uint2*>(&ta[(pt * 32 + frow) * S + (part * NM + m) * 32 + g * 8 + fh * 4]) = pk.u;
        }
    if constexpr (WPT > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();

    // ---- second conv: K = [t (64) | xc (64)], one cout tile at a time
    const f16* crow = tc + (pt * 32 + frow) * S + fh * 8;
    const long pix = p0 + pt * 32 + frow;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ct = part * NC + c;
        f16x8 fa2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) fa2[q] = *reinterpret_cast<const f16x8*>(w2 + (((long)ct * 8 + q) * 64 + lane) * 8);
        float4 bias2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias2[g] = *reinterpret_cast<const float4*>(b2 + ct * 32 + g * 8 + fh * 4);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa2[q], *reinterpret_cast<const f16x8*>(brow + q * 16), acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa2[4 + q], *reinterpret_cast<const f16x8*>(crow + q * 16), acc, 0, 0, 0);
        if (pix < P) {
            f16* dst = out + pix * out_cs + out_coff + ct * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a4[4] = {acc[4 * g + 0] + bias2[g].x, acc[4 * g + 1] + bias2[g].y, acc[4 * g + 2] + bias2[g].z,
                               acc[4 * g + 3] + bias2[g].w};
                apply_act_n<4>(a4, act2);
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)a4[e];
                *reinterpret_cast<f16x4*>(dst + g * 8 + fh * 4) = o;
            }
        }
    }
}

template <int COUT, int PT>
int launch_pt(const f16* xa, int xa_cs, int xa_coff, const f16* xc, int xc_cs, int xc_coff, f16* out, int out_cs,
              int out_coff, const f16* w1, const float* b1, const f16* w2, const float* b2, long P, int act1, int act2,
              hipStream_t s) {
    const dim3 grid((unsigned)((P + 32 * PT - 1) / (32 * PT))), block(256);
    hipLaunchKernelGGL((pair11_kernel<COUT, PT>), grid, block, 0, s, xa, xa_cs, xa_coff, xc, xc_cs, xc_coff, out, out_cs, out_coff,
                       w1, b1, w2, b2, P, act1, act2);
    FM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

bool pair11_supported(int cin, int mid, int extra, int cout) { return cin == 64 && mid == 64 && extra == 64 && (cout == 64 || cout == 128); }

// w1 [64 x 64], w2 [cout x 128] in MFMA A-fragment order ([cout / 32][K / 16][lane][8], Graph._pack_frag); b1 f32[64], b2 f32[cout]
int launch_pair11(const f16* xa, int xa_cs, int xa_coff, const f16* xc, int xc_cs, int xc_coff, f16* out, int out_cs,
                  int out_coff, const f16* w1, const float* b1, const f16* w2, const float* b2, long P, int cout, int act1,
                  int act2, hipStream_t s) {
    FM_CHECK_ARG((cout == 64 || cout == 128) && xa_cs % 8 == 0 && xa_coff % 8 == 0 && xc_cs % 8 == 0 && xc_coff % 8 == 0 &&
                 out_cs % 4 == 0 && out_coff % 4 == 0 && P > 0);
    // 128-pixel tiles where they fill the chip (>= 256 workgroups), 64-pixel tiles below
    const bool small = P < 256 * 128;
#define PAIR11_ARGS xa, xa_cs, xa_coff, xc, xc_cs, xc_coff, out, out_cs, out_coff, w1, b1, w2, b2, P, act1, act2, s
    if (cout == 64) return small ? launch_pt<64, 2>(PAIR11_ARGS) : launch_pt<64, 4>(PAIR11_ARGS);
    return small ? launch_pt<128, 2>(PAIR11_ARGS) : launch_pt<128, 4>(PAIR11_ARGS);
#undef PAIR11_ARGS
}

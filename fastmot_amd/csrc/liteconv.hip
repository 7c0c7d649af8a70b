// Fused OSNet "LightConv3x3" (torchreid osnet.py LightConv3x3: 1x1 pointwise, linear, no bias ->
// depthwise 3x3 -> BN -> ReLU), one launch instead of two and no HBM round trip of the pointwise map.
//
// One workgroup (4 waves) produces a th x tw output tile of one sample:
//   phase A  pointwise GEMM on the matrix cores for the (th+2) x (tw+2) halo positions:
//            D[cout][pos] = W[cout][cin] * X[pos][cin] with v_mfma_f32_32x32x16_f16; every lane loads
//            its 8-channel operand straight from HBM/L2 (NHWC rows are the K-contiguous operand the
//            instruction wants, so the input never touches LDS); positions outside the image give 0,
//            which is exactly the zero padding of the depthwise stage because the pointwise has no bias.
//            The fp16-rounded result (the engine stores every activation as fp16) goes to LDS with a row
//            stride of an odd number of 16 B chunks, conflict-free for the 16 B reads of phase B.
//   phase B  depthwise 3x3 + folded-BN bias + activation from LDS, 16 B coalesced stores.
#include "net.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// blockIdx.y selects one of up to 4 independent LightConvs of the same geometry (the parallel streams
// of an OSNet block at the same depth): group g reads view in[g], writes channels
// [out_coff + g*C, +C) of `out` and uses the g-th slab of the stacked weights.
struct LiteGroups {
    const f16* in[4];
    int in_cs[4], in_coff[4];
};
template <typename T>
__device__ __forceinline__ T pick4(const T (&a)[4], int g) {
    return g == 0 ? a[0] : g == 1 ? a[1] : g == 2 ? a[2] : a[3];
}

template <int NT, int KS, int MAXPOS, int SMAX>
__global__ __launch_bounds__(256) void liteconv_kernel(
    const LiteGroups grp, f16* __restrict__ out, int out_cs, int out_coff_base,
    const f16* __restrict__ wpw_base, int kpad, const f16* __restrict__ wdw_base,
    const float* __restrict__ bias_base, int H, int W, int C, int th, int tw, int tiles_x, int tiles_y,
    int act, float* __restrict__ gap_out) {
    const int grp_id = blockIdx.y;
    const f16* __restrict__ in = pick4(grp.in, grp_id);
    const int in_cs = pick4(grp.in_cs, grp_id), in_coff = pick4(grp.in_coff, grp_id);
    const int out_coff = out_coff_base + grp_id * C;
    const f16* __restrict__ wpw = wpw_base + (size_t)grp_id * (32 * NT) * kpad;
    const f16* __restrict__ wdw = wdw_base + (size_t)grp_id * 9 * C;
    const float* __restrict__ bias = bias_base + (size_t)grp_id * C;
    __shared__ __attribute__((aligned(16))) f16 ys[MAXPOS * SMAX];
    __shared__ __attribute__((aligned(16))) f16 wd[9 * 32 * NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = C + (((C >> 3) & 1) ? 0 : 8);
    const int tile = blockIdx.x % (tiles_x * tiles_y);
    const long n = blockIdx.x / (tiles_x * tiles_y);
    const int ty0 = (tile / tiles_x) * th, tx0 = (tile % tiles_x) * tw;
    const int hw = tw + 2, npos = (th + 2) * hw;
    const f16* img = in + n * (long)H * W * in_cs + in_coff;

    for (int i = tid; i < 9 * C / 8; i += 256)
        *reinterpret_cast<uint4*>(&wd[i * 8]) = *reinterpret_cast<const uint4*>(wdw + i * 8);

    // ---- phase A
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    f16x8 afr[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            afr[nt][ks] = *reinterpret_cast<const f16x8*>(wpw + (long)(nt * 32 + frow) * kpad + ks * 16 + fk);
    const int mtiles = (npos + 31) / 32;
    for (int mt = wave; mt < mtiles; mt += 4) {
        const int pos = mt * 32 + frow;
        const int py = ty0 - 1 + pos / hw, px = tx0 - 1 + pos % hw;
        const bool inside = pos < npos && py >= 0 && py < H && px >= 0 && px < W;
        const f16* src = img + ((long)min(max(py, 0), H - 1) * W + min(max(px, 0), W - 1)) * in_cs;
        f16x8 bfr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kc = ks * 16 + fk;
            uint4 v = *reinterpret_cast<const uint4*>(src + (kc < C ? kc : 0));
            const bool ok = inside && kc < C;
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            bfr[ks] = *reinterpret_cast<f16x8*>(&v);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[nt][ks], bfr[ks], acc, 0, 0, 0);
            if (pos < npos) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = nt * 32 + g * 8 + (lane >> 5) * 4;
                    if (c0 < C) {
                        union { f16 h[4]; uint2 u; } pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk.h[e] = (f16)acc[g * 4 + e];
                        *reinterpret_cast<uint2*>(&ys[pos * S + c0]) = pk.u;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase B: a thread keeps ONE channel group (cg) and walks pixels, so that the per-channel sums
    // of the tile (the OSNet gate's average pool, fused here: phase C) accumulate in registers
    const int c8n = C / 8, lanes_px = 256 / c8n;          // pixel lanes; 256 % c8n threads idle
    const int cg = tid % c8n, pl = tid / c8n;
    const bool active = pl < lanes_px;
    f16* dst = out + n * (long)H * W * out_cs + out_coff;
    float gsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        float b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = bias[cg * 8 + e];
        float kf[9][8];                                    // this thread's depthwise taps, as floats
#pragma unroll
        for (int t = 0; t < 9; ++t) unpack8(*reinterpret_cast<const uint4*>(&wd[t * C + cg * 8]), kf[t]);
        const int tw_shift = tw == 16 ? 4 : 3;             // liteconv_tiling: tw is 16 or 8
        for (int pix = pl; pix < th * tw; pix += lanes_px) {
            const int oy = pix >> tw_shift, ox = pix & (tw - 1);
            const int gy = ty0 + oy, gx = tx0 + ox;
            if (gy >= H || gx >= W) continue;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = b8[e];
            const f16* yp = &ys[(oy * hw + ox) * S + cg * 8];
            uint4 raw[9];                                   // all nine taps requested before the first FMA
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    raw[dy * 3 + dx] = *reinterpret_cast<const uint4*>(yp + (dy * hw + dx) * S);
#pragma unroll
            for (int t = 0; t < 9; ++t) fma_mix_n<8>(reinterpret_cast<const uint32_t*>(&raw[t]), kf[t], acc);
            apply_act_n<8>(acc, act);
            const uint4 o = pack8(acc);
            *reinterpret_cast<uint4*>(dst + ((long)gy * W + gx) * out_cs + cg * 8) = o;
            if (gap_out) {   // sums of the STORED (fp16-rounded) activations, as a separate GAP would see them
                float r[8];
                unpack8(o, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) gsum[e] += r[e];
            }
        }
    }
    // ---- phase C (stream ends only: group 0): per-tile channel sums -> gap_out[n][tile][C], summed over
    // the pixel lanes in a fixed order (deterministic); the gated-sum kernel adds the tiles
    if (gap_out && grp_id == 0) {
        __syncthreads();                                  // ys is dead: reuse it as float red[lanes_px][C]
        float* red = reinterpret_cast<float*>(ys);
        if (active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[pl * C + cg * 8 + e] = gsum[e];
        }
        __syncthreads();
        if (tid < C) {
            float s = 0.f;
            for (int q = 0; q < lanes_px; ++q) s += red[q * C + tid];
            gap_out[((size_t)n * (tiles_x * tiles_y) + tile) * C + tid] = s;
        }
    }
}

}  // namespace

// in/out: NHWC fp16 with channel strides in_cs/out_cs and channel offsets; per group g < G: wpw packed
// pointwise weights [ceil32(C)][kpad] (same packing as launch_conv); wdw: [9][C]; bias: f32[C] (folded BN
// of the depthwise), the G slabs stacked contiguously.
void liteconv_tiling(int C, int W, int H, int* th, int* tw, int* tiles_x, int* tiles_y) {
    const int nt = (C + 31) / 32;
    if (nt == 1) { *th = 16; *tw = W > 8 ? 16 : 8; }
    else if (W > 8) { *th = 8; *tw = 16; }
    else { *th = 16; *tw = 8; }
    *tiles_x = (W + *tw - 1) / *tw;
    *tiles_y = (H + *th - 1) / *th;
}

// gap_out (optional): fp32 [N][tiles][C] per-tile channel sums of group 0's output (see phase C)
int launch_liteconv(int G, const f16* const* in, const int* in_cs, const int* in_coff, f16* out, int out_cs,
                    int out_coff, const f16* wpw, int kpad, const f16* wdw, const float* bias, int N, int H,
                    int W, int C, int act, float* gap_out, hipStream_t s) {
    FM_CHECK_ARG(G >= 1 && G <= 4 && C % 8 == 0 && C >= 8 && C <= 128 && out_cs % 8 == 0 && out_coff % 8 == 0);
    LiteGroups grp{};
    for (int g = 0; g < 4; ++g) {
        const int q = g < G ? g : 0;
        FM_CHECK_ARG(in_cs[q] % 8 == 0 && in_coff[q] % 8 == 0);
        grp.in[g] = in[q]; grp.in_cs[g] = in_cs[q]; grp.in_coff[g] = in_coff[q];
    }
    const int nt = (C + 31) / 32, ks = (C + 15) / 16;
    int th, tw, tiles_x, tiles_y;
    liteconv_tiling(C, W, H, &th, &tw, &tiles_x, &tiles_y);
    const dim3 grid((unsigned)((long)N * tiles_x * tiles_y), G), block(256);
#define LC_LAUNCH(NT_, KS_, MAXPOS_, SMAX_)                                                                  \
    hipLaunchKernelGGL((liteconv_kernel<NT_, KS_, MAXPOS_, SMAX_>), grid, block, 0, s, grp, out, out_cs,       \
                       out_coff, wpw, kpad, wdw, bias, H, W, C, th, tw, tiles_x, tiles_y, act, gap_out)
    if (ks == 1) LC_LAUNCH(1, 1, 324, 40);
    else if (nt == 1) LC_LAUNCH(1, 2, 324, 40);
    else if (nt == 2) { if (ks <= 3) LC_LAUNCH(2, 3, 180, 72); else LC_LAUNCH(2, 4, 180, 72); }
    else if (nt == 3) { if (ks <= 5) LC_LAUNCH(3, 5, 180, 104); else LC_LAUNCH(3, 6, 180, 104); }
    else { if (ks <= 7) LC_LAUNCH(4, 7, 180, 136); else LC_LAUNCH(4, 8, 180, 136); }
#undef LC_LAUNCH
    FM_HIP(hipGetLastError());
    return 0;
}

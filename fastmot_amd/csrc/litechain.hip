// The four parallel streams of an OSNet block (torchreid osnet.py OSBlock: conv2a..conv2d = chains of 1, 2, 3
// and 4 LightConv3x3 over the same input) in ONE launch.  liteconv.hip already runs one LightConv (1x1 on the
// matrix cores -> LDS -> depthwise 3x3 + BN + ReLU) per launch, the streams of equal depth side by side; a
// block still took 4 dependent launches of ~10 us for a few microseconds of work.  Here blockIdx.y is the
// stream and the workgroup walks the whole chain of its stream inside LDS:
//
//   level l of a depth-D stream works on the output tile grown by a halo of D - l pixels:
//     phase A  pointwise GEMM (MFMA) over the halo region; operand = the block input from HBM/L2 (l = 0) or the
//              previous level's output tile in LDS (l > 0); fp16-rounded result -> LDS tile `ys`
//     phase B  depthwise 3x3 + bias + activation over the region shrunk by one pixel; the result goes to the LDS
//              tile `zb` (zero outside the image: that is the next level's zero padding) or, at the last level,
//              to HBM, and its per-tile channel sums to the gate's partial-sum slot (as liteconv.hip's phase C).
//   Every intermediate is rounded to fp16 exactly where the per-level launches stored it, the MFMA K order is the
//   same: results are bit-identical to the per-level path (tests/test_conv_gpu.py).  The halo makes the deepest
//   stream recompute (24^2 + 22^2 + 20^2 + 18^2) / (4 * 18^2) = 1.38x of the pointwise work of a 16 x 16 tile.
#include "net.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ChainGaps {
    float* p[4];
};

template <int NT, int KS>
__global__ __launch_bounds__(256) void litechain_kernel(
    const f16* __restrict__ in, int in_cs, int in_coff, f16* __restrict__ out, int out_cs, int out_coff_base,
    const f16* __restrict__ wpw_base, int kpad, const f16* __restrict__ wdw_base,
    const float* __restrict__ bias_base, int H, int W, int C, int th, int tw, int tiles_x, int tiles_y, int act,
    ChainGaps gaps, int S, int ys_elems, int zb_elems) {
    extern __shared__ __attribute__((aligned(16))) f16 lds[];
    f16* ys = lds;                       // pointwise output of the current level (halo region)
    f16* zb = lds + ys_elems;            // depthwise output of the previous level = operand of this one
    f16* wd = zb + zb_elems;             // depthwise weights of the current level [9][C]
    const int stream = blockIdx.y, D = stream + 1, pset0 = stream * (stream + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const int tile = blockIdx.x % (tiles_x * tiles_y);
    const long n = blockIdx.x / (tiles_x * tiles_y);
    const int ty0 = (tile / tiles_x) * th, tx0 = (tile % tiles_x) * tw;
    const f16* img = in + n * (long)H * W * in_cs + in_coff;
    const int out_coff = out_coff_base + stream * C;
    f16* dst = out + n * (long)H * W * out_cs + out_coff;

    const int c8n = C / 8, lanes_px = 256 / c8n;          // phase B: a thread keeps one 8-channel group
    const int cg = tid % c8n, pl = tid / c8n;
    const bool active = pl < lanes_px;
    float gsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int lvl = 0; lvl < D; ++lvl) {
        const int hl = D - lvl;                           // halo of this level's pointwise region
        const int wp = tw + 2 * hl, hp = th + 2 * hl, npos = wp * hp;
        const f16* wpw = wpw_base + (size_t)(pset0 + lvl) * (32 * NT) * kpad;
        const f16* wdw = wdw_base + (size_t)(pset0 + lvl) * 9 * C;
        const float* bias = bias_base + (size_t)(pset0 + lvl) * C;
        for (int i = tid; i < 9 * C / 8; i += 256)        // (the previous level's phase B is behind a barrier)
            *reinterpret_cast<uint4*>(&wd[i * 8]) = *reinterpret_cast<const uint4*>(wdw + i * 8);

        // ---- phase A
        f16x8 afr[NT][KS];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                afr[nt][ks] = *reinterpret_cast<const f16x8*>(wpw + (long)(nt * 32 + frow) * kpad + ks * 16 + fk);
        const int mtiles = (npos + 31) / 32;
        for (int mt = wave; mt < mtiles; mt += 4) {
            const int pos = mt * 32 + frow, posc = min(pos, npos - 1);
            const int py = ty0 - hl + posc / wp, px = tx0 - hl + posc % wp;
            const bool inside = pos < npos && py >= 0 && py < H && px >= 0 && px < W;
            // operand rows: the block input in HBM/L2 at level 0, the previous level's LDS tile afterwards (it
            // covers exactly this region and is zero outside the image).  (Requesting all of a wave's level-0
            // tiles up front was measured SLOWER: 54 -> 62 us for the 64 x 32 stage; the levels are bound by
            // the depthwise phase, not by this round trip.)
            const f16* src = lvl == 0 ? img + ((long)min(max(py, 0), H - 1) * W + min(max(px, 0), W - 1)) * in_cs
                                      : nullptr;
            f16x8 bfr[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int kc = ks * 16 + fk, kcc = kc < C ? kc : 0;
                uint4 v = lvl == 0 ? *reinterpret_cast<const uint4*>(src + kcc)
                                   : *reinterpret_cast<const uint4*>(&zb[posc * S + kcc]);
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

        // ---- phase B
        const int wz = wp - 2, hz = hp - 2;
        const bool last = lvl == D - 1;
        if (active) {
            float b8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] = bias[cg * 8 + e];
            for (int pix = pl; pix < wz * hz; pix += lanes_px) {
                const int oy = pix / wz, ox = pix % wz;
                const int gy = ty0 - (hl - 1) + oy, gx = tx0 - (hl - 1) + ox;
                const bool in_img = gy >= 0 && gy < H && gx >= 0 && gx < W;
                if (last && !in_img) continue;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = b8[e];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        float v[8], k[8];
                        unpack8(*reinterpret_cast<const uint4*>(&ys[((oy + dy) * wp + ox + dx) * S + cg * 8]), v);
                        unpack8(*reinterpret_cast<const uint4*>(&wd[(dy * 3 + dx) * C + cg * 8]), k);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], k[e], acc[e]);
                    }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = apply_act(acc[e], act);
                const uint4 o = pack8(acc);
                if (!last) {
                    *reinterpret_cast<uint4*>(&zb[pix * S + cg * 8]) = in_img ? o : make_uint4(0, 0, 0, 0);
                } else {
                    *reinterpret_cast<uint4*>(dst + ((long)gy * W + gx) * out_cs + cg * 8) = o;
                    float r[8];     // sums of the STORED (fp16-rounded) activations, as a separate GAP would see them
                    unpack8(o, r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gsum[e] += r[e];
                }
            }
        }
        __syncthreads();
    }

    // ---- per-tile channel sums -> gaps.p[stream][n][tile][C], summed over the pixel lanes in a fixed order
    float* gap_out = gaps.p[stream];
    if (gap_out) {
        float* red = reinterpret_cast<float*>(ys);          // ys is dead (barrier above)
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

void liteconv_tiling(int C, int W, int H, int* th, int* tw, int* tiles_x, int* tiles_y);

// LDS bytes of the chain kernel (graph.py mirrors this to decide whether a block can use it)
size_t litechain_lds_bytes(int C, int W, int H) {
    int th, tw, tx, ty;
    liteconv_tiling(C, W, H, &th, &tw, &tx, &ty);
    const int S = C + (((C >> 3) & 1) ? 0 : 8), nt = (C + 31) / 32;
    const size_t ys = (size_t)(th + 8) * (tw + 8) * S, zb = (size_t)(th + 6) * (tw + 6) * S;
    const size_t red = (size_t)(256 / (C / 8)) * C * 2;                  // phase-C floats, in halfs
    return ((ys > red ? ys : red) + zb + 9 * 32 * nt) * sizeof(f16);
}

// in: the block's conv1 output (C channels); out: 4 C channels, stream s (depth s + 1) at [s C, (s + 1) C);
// weights: the 10 LightConv parameter sets in (stream, level) order -- wpw [10][ceil32(C)][kpad], wdw [10][9][C],
// bias f32 [10][C]; gap[s]: fp32 [N][tiles][C] per-tile channel sums of stream s (liteconv_tiling tiles)
int launch_litechain(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, const f16* wpw,
                     int kpad, const f16* wdw, const float* bias, int N, int H, int W, int C, int act,
                     float* const* gap, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && C >= 8 && C <= 128 && in_cs % 8 == 0 && in_coff % 8 == 0 && out_cs % 8 == 0 &&
                 out_coff % 8 == 0);
    const size_t shmem = litechain_lds_bytes(C, W, H);
    FM_CHECK_ARG(shmem <= 64 * 1024);
    int th, tw, tiles_x, tiles_y;
    liteconv_tiling(C, W, H, &th, &tw, &tiles_x, &tiles_y);
    const int S = C + (((C >> 3) & 1) ? 0 : 8), nt = (C + 31) / 32, ks = (C + 15) / 16;
    const size_t ys = (size_t)(th + 8) * (tw + 8) * S, red = (size_t)(256 / (C / 8)) * C * 2;
    const int ys_elems = (int)(ys > red ? ys : red), zb_elems = (th + 6) * (tw + 6) * S;
    ChainGaps gaps{};
    for (int i = 0; i < 4; ++i) gaps.p[i] = gap ? gap[i] : nullptr;
    const dim3 grid((unsigned)((long)N * tiles_x * tiles_y), 4), block(256);
#define LCH_LAUNCH(NT_, KS_)                                                                                      \
    hipLaunchKernelGGL((litechain_kernel<NT_, KS_>), grid, block, shmem, s, in, in_cs, in_coff, out, out_cs,      \
                       out_coff, wpw, kpad, wdw, bias, H, W, C, th, tw, tiles_x, tiles_y, act, gaps, S, ys_elems, \
                       zb_elems)
    if (ks == 1) LCH_LAUNCH(1, 1);
    else if (nt == 1) LCH_LAUNCH(1, 2);
    else if (nt == 2) { if (ks <= 3) LCH_LAUNCH(2, 3); else LCH_LAUNCH(2, 4); }
    else if (nt == 3) { if (ks <= 5) LCH_LAUNCH(3, 5); else LCH_LAUNCH(3, 6); }
    else { if (ks <= 7) LCH_LAUNCH(4, 7); else LCH_LAUNCH(4, 8); }
#undef LCH_LAUNCH
    FM_HIP(hipGetLastError());
    return 0;
}

extern "C" size_t fm_litechain_lds_bytes(int c, int w, int h) {
    return (c % 8 == 0 && c >= 8 && c <= 128 && w > 0 && h > 0) ? litechain_lds_bytes(c, w, h) : (size_t)-1;
}

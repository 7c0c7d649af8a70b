// Tail of an OSNet block in one launch (torchreid osnet.py OSBlock.forward: x2 = sum_t gate(x2_t) * x2_t;
// out = relu(conv3(x2) + downsample(x) | x)): the unified aggregation gate, the gated sum of the four streams and the
// 1x1 conv3 (+ the 1x1 downsample of a stage's first block as a second K segment, weights [W3 | Wd], or the identity
// shortcut).  Round 3 ran it as FM_OP_GATED_SUM + FM_OP_CONV: the gated sum went to HBM as fp16 and came back as the
// conv's operand -- six extra launches per ReID pass, each two dependent memory round trips long (7 - 13 us inside the
// pipeline for a few hundred KB of work).
//
// One workgroup = 128 pixels of one sample x up to 128 output channels (blockIdx.z: further output channel chunks; the
// x1.0 widths have up to 512).  Prologue: the gate MLP of the four streams from the per-tile channel sums the chain
// kernel left (same code as gated_sum_part_kernel: the gates are the same floats).  Main loop: a wavefront owns 32
// pixels; per 16-channel K step every lane builds its 8-channel operand in registers --
//   first segment : sum_t gate[t][k] * x_t[pixel][k] in fp32 (fmaf, t ascending), rounded to fp16: the value the unfused
//                   path stored and re-read
//   second segment: the block input x[pixel][k] as it is
// -- and feeds it to v_mfma_f32_32x32x16_f16 against the weight rows read from L2 (D[cout][pixel], K ascending: the
// conv engine's orientation and order).  NHWC rows are the K-contiguous operand the instruction wants, so no activation
// touches LDS.  Epilogue: + bias (+ identity shortcut) -> activation -> fp16, four channels per lane and pixel.
#include "net.h"

namespace {

struct GatedConvArgs {
    const f16* in[4];                     // the four streams, C channels each
    int in_cs[4], in_coff[4];
    const float* part[4];                 // per-tile channel sums of each stream [N][tiles][C]
    const f16* x2;                        // second K segment (nullable), c2 channels
    int x2_cs, x2_coff, c2;
    const f16* res;                       // identity shortcut (nullable), added before the activation
    int res_cs, res_coff;
    f16* out;
    int out_cs, out_coff;
    const f16* w;                         // [ceil32(cout)][kpad], K = C + c2
    const float* bias;
    int kpad;
    const f16* w1; const float* b1;       // gate MLP: fc1 [hid][C], fc2 [C][hid]
    const f16* w2; const float* b2;
    int HW, C, hid, tiles, cout, act;
};

constexpr int GC_PIX = 128;               // pixels per workgroup (32 per wavefront)

template <int NTW>
__global__ __launch_bounds__(256) void gatedconv_kernel(const GatedConvArgs a) {
    extern __shared__ float sm[];         // gap[4][C] | hidden[4][hid] | gate[4][C] | w1[hid][C] | w2[C][hid] | b1[hid] | b2[C] | bias[NTW * 32]
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, hid = a.hid, HW = a.HW;
    float* gap = sm;
    float* hidden = gap + 4 * C;
    float* gate = hidden + 4 * hid;
    float* sw1 = gate + 4 * C;
    float* sw2 = sw1 + hid * C;
    float* sb1 = sw2 + C * hid;
    float* sb2 = sb1 + hid;
    float* sbias = sm + ((8 * C + 4 * hid + 2 * hid * C + hid + C + 3) & ~3);   // [NTW * 32] this chunk's conv bias (16 B aligned)
    // Everything the launch reads before its epilogue is requested up front -- the first K step's four stream vectors and
    // weight rows, the first vector of the second segment, the shortcut's values -- beside the gate MLP's inputs, so that
    // the kernel is about two memory round trips long (the first version, which loaded as it went: 16 - 27 us per launch,
    // no faster than the two launches it replaced).
    const int pos = blockIdx.x * GC_PIX + wave * 32 + (lane & 31), fk = (lane >> 5) * 8;
    const bool pin = pos < HW;
    const size_t pix = (size_t)n * HW + min(pos, HW - 1);
    const int cbase = blockIdx.z * (NTW * 32);
    const int ntv = min(NTW, (a.cout + 31) / 32 - (int)blockIdx.z * NTW);   // tiles of this chunk that exist (uniform)
    const f16* wrow = a.w + (size_t)(cbase + (lane & 31)) * a.kpad + fk;
    uint4 first[4];
    f16x8 wfirst[NTW];
    {
        const int kc = min(fk, C - 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) first[t] = *reinterpret_cast<const uint4*>(a.in[t] + pix * a.in_cs[t] + a.in_coff[t] + kc);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            if (nt < ntv) wfirst[nt] = *reinterpret_cast<const f16x8*>(wrow + (size_t)nt * 32 * a.kpad);
    }
    uint4 x2first = make_uint4(0u, 0u, 0u, 0u);
    if (a.c2) x2first = *reinterpret_cast<const uint4*>(a.x2 + pix * a.x2_cs + a.x2_coff + min(fk, a.c2 - 8));
    f16x4 rpre[NTW][4];
    if (a.res) {
        const f16* rrow = a.res + pix * a.res_cs + a.res_coff;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = cbase + nt * 32 + g * 8 + (lane >> 5) * 4;
                rpre[nt][g] = *reinterpret_cast<const f16x4*>(rrow + min(c0, a.cout - 4));
            }
    }
    for (int i = tid; i < NTW * 32; i += 256) sbias[i] = a.bias[min(cbase + i, ((a.cout + 31) & ~31) - 1)];
    for (int i = tid; i < hid * C; i += 256) { sw1[i] = (float)a.w1[i]; sw2[i] = (float)a.w2[i]; }
    for (int i = tid; i < hid; i += 256) sb1[i] = a.b1[i];
    for (int i = tid; i < C; i += 256) sb2[i] = a.b2[i];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float* p = a.part[t] + (size_t)n * a.tiles * C;
        for (int c = tid; c < C; c += 256) {
            float s = 0.f;
            for (int q = 0; q < a.tiles; ++q) s += p[q * C + c];
            gap[t * C + c] = s / (float)HW;
        }
    }
    __syncthreads();
    for (int i = tid; i < 4 * hid; i += 256) {
        const int t = i / hid, h = i % hid;
        float s = sb1[h];
        for (int c = 0; c < C; ++c) s = fmaf(sw1[h * C + c], gap[t * C + c], s);
        hidden[t * hid + h] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 4 * C; i += 256) {
        const int t = i / C, c = i % C;
        float s = sb2[c];
        for (int h = 0; h < hid; ++h) s = fmaf(sw2[c * hid + h], hidden[t * hid + h], s);
        gate[t * C + c] = 1.f / (1.f + __expf(-s));
    }
    __syncthreads();

    f32x16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    // ---- main loop.  K is laid out in two segments of whole 16-channel steps (weights zero in the padding):
    // [0, ceil16(C)) the gated sum, [ceil16(C), ceil16(C) + ceil16(c2)) the block input -- every step is one kind for the
    // whole wavefront and its loads are unconditional (lanes in the padding re-read the last valid channels against
    // zero weights), so the next step's vectors and weight rows are in flight while this one is multiplied.
    const int C16 = (C + 15) & ~15, nk1 = C16 / 16, nk2 = (a.c2 + 15) / 16;
    auto load_streams = [&](int ks, uint4 (&dst)[4]) {
        const int kc = min(ks * 16 + fk, C - 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t] = *reinterpret_cast<const uint4*>(a.in[t] + pix * a.in_cs[t] + a.in_coff[t] + kc);
    };
    auto load_w = [&](int kstep, f16x8 (&dst)[NTW]) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            if (nt < ntv) dst[nt] = *reinterpret_cast<const f16x8*>(wrow + (size_t)nt * 32 * a.kpad + kstep * 16);
    };
    uint4 cur[4], nxt[4];
    f16x8 wc[NTW], wn[NTW];
    for (int ks = 0; ks < nk1; ++ks) {
        if (ks + 1 < nk1) { load_streams(ks + 1, nxt); load_w(ks + 1, wn); }
        const int kc = min(ks * 16 + fk, C - 8);
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v[8];
            unpack8(ks == 0 ? first[t] : cur[t], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(v[e], gate[t * C + kc + e], o[e]);
        }
        const uint4 bv = pack8(o);
        const f16x8 bf = *reinterpret_cast<const f16x8*>(&bv);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            if (nt < ntv) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks == 0 ? wfirst[nt] : wc[nt], bf, acc[nt], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) wc[nt] = wn[nt];
    }
    if (nk2) {
        auto load_x2 = [&](int ks) {
            return *reinterpret_cast<const uint4*>(a.x2 + pix * a.x2_cs + a.x2_coff + min(ks * 16 + fk, a.c2 - 8));
        };
        uint4 xc = x2first, xn = xc;
        load_w(nk1, wc);
        for (int ks = 0; ks < nk2; ++ks) {
            if (ks + 1 < nk2) { xn = load_x2(ks + 1); load_w(nk1 + ks + 1, wn); }
            const f16x8 bf = *reinterpret_cast<const f16x8*>(&xc);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                if (nt < ntv) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[nt], bf, acc[nt], 0, 0, 0);
            xc = xn;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) wc[nt] = wn[nt];
        }
    }
    if (!pin) return;
    f16* orow = a.out + pix * a.out_cs + a.out_coff;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = nt * 32 + g * 8 + (lane >> 5) * 4, c0 = cbase + cl;
            if (c0 >= a.cout) continue;
            const float4 b4 = *reinterpret_cast<const float4*>(sbias + cl);
            float v[4] = {acc[nt][g * 4 + 0] + b4.x, acc[nt][g * 4 + 1] + b4.y, acc[nt][g * 4 + 2] + b4.z,
                          acc[nt][g * 4 + 3] + b4.w};
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rpre[nt][g][e];
            }
            apply_act_n<4>(v, a.act);
            f16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (f16)v[e];
            *reinterpret_cast<f16x4*>(orow + c0) = o4;
        }
}

}  // namespace

// in: the four stream views (C channels each, NHWC fp16); parts: their per-tile channel sums [N][tiles][C]; x2: the second
// K segment (c2 channels, nullable); res: identity shortcut (nullable); w: [ceil32(cout)][kpad], columns [0, C) = the
// conv over the gated sum, [ceil16(C), ceil16(C) + c2) = the conv over x2, zero elsewhere; gate MLP as FM_OP_GATED_SUM's.
int launch_gatedconv(const f16* const* in, const int* in_cs, const int* in_coff, const float* const* parts, int tiles,
                     const f16* x2, int x2_cs, int x2_coff, int c2, const f16* res, int res_cs, int res_coff, f16* out,
                     int out_cs, int out_coff, const f16* w, const float* bias, int kpad, const f16* w1, const float* b1,
                     const f16* w2, const float* b2, int N, int HW, int C, int hid, int cout, int act, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && C >= 8 && C <= 128 && hid >= 1 && hid <= 16 && c2 % 8 == 0 && c2 >= 0 && cout % 8 == 0 &&
                 out_cs % 4 == 0 && out_coff % 4 == 0 && kpad >= ((C + 15) & ~15) + ((c2 + 15) & ~15) && tiles >= 1 &&
                 (!res || (res_cs % 4 == 0 && res_coff % 4 == 0)) && (c2 == 0 || (x2 && x2_cs % 8 == 0 && x2_coff % 8 == 0)));
    GatedConvArgs a{};
    for (int t = 0; t < 4; ++t) {
        FM_CHECK_ARG(in[t] && parts[t] && in_cs[t] % 8 == 0 && in_coff[t] % 8 == 0);
        a.in[t] = in[t]; a.in_cs[t] = in_cs[t]; a.in_coff[t] = in_coff[t]; a.part[t] = parts[t];
    }
    a.x2 = x2; a.x2_cs = x2_cs; a.x2_coff = x2_coff; a.c2 = c2;
    a.res = res; a.res_cs = res_cs; a.res_coff = res_coff;
    a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
    a.w = w; a.bias = bias; a.kpad = kpad;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.HW = HW; a.C = C; a.hid = hid; a.tiles = tiles; a.cout = cout; a.act = act;
    const size_t lds = sizeof(float) * (size_t)(((8 * C + 4 * hid + 2 * hid * C + hid + C + 3) & ~3) + 128);
    const int nt = (cout + 31) / 32;
    const int ntw = nt >= 4 ? 4 : nt;                      // output channel tiles per workgroup
    const dim3 grid((HW + GC_PIX - 1) / GC_PIX, N, (nt + ntw - 1) / ntw), block(256);
    switch (ntw) {
        case 1: hipLaunchKernelGGL(gatedconv_kernel<1>, grid, block, lds, s, a); break;
        case 2: hipLaunchKernelGGL(gatedconv_kernel<2>, grid, block, lds, s, a); break;
        case 3: hipLaunchKernelGGL(gatedconv_kernel<3>, grid, block, lds, s, a); break;
        default: hipLaunchKernelGGL(gatedconv_kernel<4>, grid, block, lds, s, a); break;
    }
    FM_HIP(hipGetLastError());
    return 0;
}

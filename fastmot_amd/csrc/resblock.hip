// Fused CSPDarknet residual unit:  y = x + act(conv3x3(act(conv1x1(x))))   (yolov4.cfg: [convolutional] 1x1,
// [convolutional] 3x3, [shortcut] from=-3; yolo2onnx.py:558-705,733-760).  One launch instead of three
// (1x1 conv, split-K 3x3 conv, reduce): on this part every dependent dispatch costs ~5 us, and the 23
// residual units are 40 % of YOLOv4's launches.
//
// One workgroup (4 waves) produces a th x tw tile of output pixels for all C output channels:
//   phase 1  1x1 conv + bias + activation on the (th+2) x (tw+2) halo positions with the matrix cores;
//            both operands come straight from HBM/L2 in fragment layout (NHWC pixels are K-contiguous),
//            the fp16-rounded result -- exactly what the unfused layer would have stored --
//            goes to LDS; positions outside the image are ZERO (they are the 3x3 conv's padding);
//   phase 2  3x3 conv as 9 taps x M/16 MFMA steps: B fragments from the LDS tile (row stride padded by
//            16 B: conflict-free ds_read_b128), A fragments (weights pre-packed in fragment order: 1 KB
//            contiguous per load) streamed from L2; + bias + activation + shortcut, 8-byte NHWC stores.
// The halo makes phase 1 do (th+2)(tw+2)/(th tw) = 1.56x (8x8) / 1.88x (4x8) of the 1x1 work; it is
// 1/9 of the total.
#include "net.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C: channels in/out, M: mid channels, TH x 8: pixel tile, NW: waves, CS: workgroups splitting the couts of
// one tile, KS: waves splitting the K range (taps) of one accumulator tile.  Every wave owns one 32-cout x
// 32-pixel accumulator tile in phase 2 ((C/32/CS) * (TH*8/32) * KS == NW); the KS partial tiles are summed
// through LDS.  CS and KS exist for the deep stages (38 x 38 x 256: 50 tiles): a workgroup that streams all
// of W2 (1.2 MB) through one CU's L1 is bound by that, so the stream is split over more CUs and more waves.  Both K loops are fully unrolled around a ring of PD weight-fragment
// chunks in flight: the kernel runs ~1 wave per SIMD on few workgroups, so nothing but the wave's own
// prefetch distance hides the ~1 us L2/HBM round trip of the streamed weights.
template <int C, int M, int TH, int NW, int CS, int KS>
__global__ __launch_bounds__(NW * 64) void resblock_kernel(
    const f16* __restrict__ x, int x_cs, int x_coff, f16* __restrict__ out, int out_cs, int out_coff,
    const f16* __restrict__ w1, const float* __restrict__ b1, const f16* __restrict__ w2,
    const float* __restrict__ b2, int H, int W, int tiles_x, int act1, int act2) {
    constexpr int TW = 8, HW = TW + 2, NPOS = (TH + 2) * HW, SX = C + 8, S = M + 8, NT = NW * 64;
    constexpr int NMT = M / 32, NPH = (NPOS + 31) / 32, NPAIR = NMT * NPH / NW;   // phase-1 tiles per wave
    constexpr int NCT = C / 32 / CS, NPT = TH * TW / 32;
    static_assert(NMT * NPH % NW == 0 && NCT * NPT * KS == NW, "tile / wave split");
    extern __shared__ __attribute__((aligned(16))) f16 lds[];
    f16* xt = lds;                      // [NPOS][SX]  input halo tile (zero outside the image)
    f16* mid = lds + NPOS * SX;         // [NPOS][S]   act1(conv1x1) on the halo (zero outside the image)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const int tile = blockIdx.x / CS, cs = blockIdx.x % CS;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const long n = blockIdx.y;
    const f16* img = x + n * (long)H * W * x_cs + x_coff;

    // ---- halo tile -> LDS, 16 B per lane, a position's channels contiguous (coalesced)
    {
        constexpr int VPP = C / 8, NV = NPOS * VPP, NIT = (NV + NT - 1) / NT;
        f16x8 v[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + i * NT, pos = e / VPP, c8 = (e % VPP) * 8;
            const int py = ty0 - 1 + pos / HW, px = tx0 - 1 + pos % HW;
            const bool ok = e < NV && py >= 0 && py < H && px >= 0 && px < W;
            const f16x8 ld = *reinterpret_cast<const f16x8*>(
                img + ((long)min(max(py, 0), H - 1) * W + min(max(px, 0), W - 1)) * x_cs + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = ok ? ld[j] : (f16)0.f;
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + i * NT;
            if (e < NV) *reinterpret_cast<f16x8*>(&xt[(e / VPP) * SX + (e % VPP) * 8]) = v[i];
        }
    }

    // ---- phase 1: mid = act1(W1 x + b1) on the halo; a wave's work items = (32 mid x 32 position tile,
    // chunk of CH1 MFMA steps), weights from global through the ring, pixels from the LDS tile
    {
        constexpr int ST = C / 16, CH1 = ST < 8 ? ST : 8, KC = ST / CH1, NI = NPAIR * KC, PD = NI < 3 ? NI : 3;
        f16x8 fa[PD][CH1];
        float4 bias1[NPAIR][4];         // loaded first: in-order returns, no vmcnt(0) drain of the ring later
#pragma unroll
        for (int p = 0; p < NPAIR; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bias1[p][g] = *reinterpret_cast<const float4*>((const float*)__builtin_assume_aligned(b1, 16) +
                                                               ((wave + p * NW) % NMT) * 32 + g * 8 + (lane >> 5) * 4);
        auto load_a = [&](int item, int slot) {
            const int mt = (wave + (item / KC) * NW) % NMT;
            const f16* wr = w1 + (((long)mt * ST + (item % KC) * CH1) * 64 + lane) * 8;
#pragma unroll
            for (int u = 0; u < CH1; ++u) fa[slot][u] = *reinterpret_cast<const f16x8*>(wr + u * 512);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) load_a(i, i);
        __syncthreads();                // halo tile complete
        f32x16 acc;
#pragma unroll
        for (int item = 0; item < NI; ++item) {
            const int p = item / KC, kc = (item % KC) * CH1 * 16;
            const int pos = ((wave + p * NW) / NMT) * 32 + frow;
            const f16* bsrc = xt + min(pos, NPOS - 1) * SX + fk + kc;
            f16x8 fb[CH1];
#pragma unroll
            for (int u = 0; u < CH1; ++u) fb[u] = *reinterpret_cast<const f16x8*>(bsrc + u * 16);
            if (item % KC == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < CH1; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[item % PD][u], fb[u], acc, 0, 0, 0);
            if (item + PD < NI) load_a(item + PD, item % PD);
            if (item % KC == KC - 1 && pos < NPOS) {
                const int py = ty0 - 1 + pos / HW, px = tx0 - 1 + pos % HW;
                const bool inside = py >= 0 && py < H && px >= 0 && px < W;
                const int mt = (wave + p * NW) % NMT;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = bias1[p][g];
                    union { f16 h[4]; uint2 u; } pk;
                    float a4[4] = {acc[g * 4 + 0] + bv.x, acc[g * 4 + 1] + bv.y, acc[g * 4 + 2] + bv.z,
                                   acc[g * 4 + 3] + bv.w};
                    apply_act_n<4>(a4, act1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk.h[e] = (f16)(inside ? a4[e] : 0.f);
                    *reinterpret_cast<uint2*>(&mid[pos * S + mt * 32 + g * 8 + (lane >> 5) * 4]) = pk.u;
                }
            }
        }
    }

    // ---- phase 2: 3x3 conv over the LDS tile; the weight fragments of a cout tile are contiguous over
    // (tap, m), so chunk q of CH MFMA steps starts CH KB after chunk q - 1.  Wave (ctl, pt, ks) runs chunks
    // [ks NQ / KS, (ks + 1) NQ / KS); a chunk never straddles a tap (CH divides M / 16 or equals it).
    const int ctl = wave % NCT, pt = (wave / NCT) % NPT, ks = wave / (NCT * NPT);
    const int ct = cs * NCT + ctl;
    const int pix = pt * 32 + frow;                          // pixel of the tile (row-major TH x TW)
    const int gy = ty0 + pix / TW, gx = tx0 + pix % TW;
    float4 bias2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        bias2[g] = *reinterpret_cast<const float4*>((const float*)__builtin_assume_aligned(b2, 16) + ct * 32 + g * 8 +
                                                    (lane >> 5) * 4);
    f32x16 acc;
    {
        constexpr int SPT = M / 16;
        constexpr int CH = (SPT % 8 == 0 && 9 * SPT % (8 * KS) == 0) ? 8 : (SPT % 4 == 0 && 9 * SPT % (4 * KS) == 0) ? 4 : 2;
        constexpr int NQ = 9 * SPT / CH / KS, PD = NQ < 24 / CH ? NQ : 24 / CH;      // 24 KB of weights in flight
        static_assert(SPT % CH == 0 && 9 * SPT % (CH * KS) == 0, "chunking");
        const int q0 = ks * NQ;
        const f16* wr = w2 + (((long)ct * (9 * SPT) + q0 * CH) * 64 + lane) * 8;
        f16x8 fa[PD][CH], fb[2][CH];
        auto load_a = [&](int q, int slot) {
#pragma unroll
            for (int u = 0; u < CH; ++u) fa[slot][u] = *reinterpret_cast<const f16x8*>(wr + (q * CH + u) * 512);
        };
        const int pbase = ((pix / TW) * HW + pix % TW) * S + fk;
        auto load_b = [&](int q, int slot) {
            const int step0 = (q0 + q) * CH, tap = step0 / SPT;          // wave-uniform
            const f16* src = mid + pbase + ((tap / 3) * HW + tap % 3) * S + (step0 % SPT) * 16;
#pragma unroll
            for (int u = 0; u < CH; ++u) fb[slot][u] = *reinterpret_cast<const f16x8*>(src + u * 16);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) load_a(i, i);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        __syncthreads();                                     // mid complete (weights already in flight)
        load_b(0, 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) load_b(q + 1, (q + 1) & 1);      // LDS fragments one chunk ahead of the MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CH; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[q % PD][u], fb[q & 1][u], acc, 0, 0, 0);
            if (q + PD < NQ) load_a(q + PD, q % PD);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (KS > 1) {                                  // sum the K-split partial tiles into the ks == 0 wave
        float4* red = reinterpret_cast<float4*>(mid);        // [(ks - 1)][ctl + NCT pt][g][lane] (resblock_lds)
        __syncthreads();                                     // every wave is done reading mid
        if (ks > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                red[(((ks - 1) * NCT * NPT + ctl + NCT * pt) * 4 + g) * 64 + lane] =
                    make_float4(acc[g * 4], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]);
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = red[(((k - 1) * NCT * NPT + ctl + NCT * pt) * 4 + g) * 64 + lane];
                acc[g * 4] += v.x; acc[g * 4 + 1] += v.y; acc[g * 4 + 2] += v.z; acc[g * 4 + 3] += v.w;
            }
    }

    // ---- epilogue: + bias, activation, + shortcut (from the LDS halo tile), store
    if (gy < H && gx < W) {
        const long gp = (n * H + gy) * (long)W + gx;
        const f16* rsrc = xt + ((pix / TW + 1) * HW + pix % TW + 1) * SX;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = ct * 32 + g * 8 + (lane >> 5) * 4;
            const f16x4 rv = *reinterpret_cast<const f16x4*>(rsrc + co);
            f16x4 o;
            float a4[4] = {acc[g * 4 + 0] + bias2[g].x, acc[g * 4 + 1] + bias2[g].y, acc[g * 4 + 2] + bias2[g].z,
                           acc[g * 4 + 3] + bias2[g].w};
            apply_act_n<4>(a4, act2);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)(a4[e] + (float)rv[e]);
            *reinterpret_cast<f16x4*>(out + gp * out_cs + out_coff + co) = o;
        }
    }
}

// LDS bytes: halo tile + max(mid tile, K-split partial tiles that reuse its space)
template <int C, int M, int TH, int NW, int KS>
constexpr size_t resblock_lds() {
    constexpr size_t npos = (TH + 2) * 10, midb = npos * (M + 8) * 2, redb = (size_t)(KS - 1) * (NW / KS) * 64 * 64;
    return npos * (C + 8) * 2 + (midb > redb ? midb : redb);
}

template <int C, int M, int TH, int NW, int CS, int KS>
int resblock_launch(const f16* x, int x_cs, int x_coff, f16* out, int out_cs, int out_coff, const f16* w1,
                    const float* b1, const f16* w2, const float* b2, int N, int H, int W, int act1, int act2,
                    hipStream_t s) {
    const int tiles_x = (W + 7) / 8, tiles_y = (H + TH - 1) / TH;
    constexpr size_t shmem = resblock_lds<C, M, TH, NW, KS>();
    static_assert(shmem <= 64 * 1024, "LDS tile");
    hipLaunchKernelGGL((resblock_kernel<C, M, TH, NW, CS, KS>), dim3(tiles_x * tiles_y * CS, N), dim3(NW * 64), shmem, s, x, x_cs,
                       x_coff, out, out_cs, out_coff, w1, b1, w2, b2, H, W, tiles_x, act1, act2);
    FM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// CSPDarknet53 uses (64, 32), (64, 64), (128, 128), (256, 256); the half-width variants cover the
// scaled-YOLOv4 bottlenecks
bool resblock_supported(int C, int M) {
    return (C == 64 || C == 128 || C == 256) && (M == C || M == C / 2);
}

// w1 (K = C) and w2 (K = 9 M, order kh, kw, m) are stored in MFMA A-fragment order:
// [cout tile of 32][K / 16 steps][lane 0..63][8 halfs], lane = (k / 8 % 2) * 32 + row -- one fragment load
// of a wave is one contiguous 1 KB block (row-major weights would touch 32 lines per load and lean on L1)
int launch_resblock(const f16* x, int x_cs, int x_coff, f16* out, int out_cs, int out_coff, const f16* w1,
                    const float* b1, const f16* w2, const float* b2, int N, int H, int W, int C, int M, int act1,
                    int act2, hipStream_t s) {
    FM_CHECK_ARG(resblock_supported(C, M) && x_cs % 8 == 0 && x_coff % 8 == 0 && out_cs % 4 == 0 && out_coff % 4 == 0);
#define RB(C_, M_, TH_, NW_, CS_, KS_)                                                                         \
    if (C == C_ && M == M_)                                                                                    \
        return resblock_launch<C_, M_, TH_, NW_, CS_, KS_>(x, x_cs, x_coff, out, out_cs, out_coff, w1, b1, w2, \
                                                           b2, N, H, W, act1, act2, s)
    // The 152^2 / 76^2 units split the taps of an accumulator tile over two waves (8 waves per workgroup; round 6,
    // scripts/resblock_sweep.py, HIP events around eager launches: 12.1 -> 11.75 us and 11.9 -> 11.35 us; an 8 x 8 pixel tile with
    // 8 waves 14.9, with 16 waves and the split 24.8 -- it spills).  FASTMOT_RB_VARIANT=0: one wave per tile, as before.
    static const int variant = getenv("FASTMOT_RB_VARIANT") ? atoi(getenv("FASTMOT_RB_VARIANT")) : 1;
    if (variant == 1) { RB(64, 64, 8, 8, 1, 2); RB(128, 128, 4, 8, 1, 2); }
    if (variant == 2) { RB(128, 128, 8, 8, 1, 1); }
    // measured per stage of YOLOv4-608 (rocprofv3, graph replay; unfused 1x1 + split-K 3x3 + reduce in brackets):
    RB(64, 32, 8, 4, 1, 1);       // 304^2: 8 x 8 pixels, 2 cout x 2 pixel tiles       20.9 us (29)
    RB(64, 64, 8, 4, 1, 1);       // 152^2                                             14 us (20)
    RB(128, 64, 4, 4, 1, 1);
    RB(128, 128, 4, 4, 1, 1);     // 76^2: 4 x 8 pixels, 4 cout tiles                  12.9 us (25)
    RB(256, 128, 4, 8, 4, 4);
    RB(256, 256, 4, 8, 4, 4);     // 38^2: 4 workgroups x 2 cout tiles x 4 K quarters  ~15 us (24.6); 21 us unsplit
#undef RB
    return FM_ERR_ARG;
}

extern "C" int fm_resblock_supported(int c, int mid) { return resblock_supported(c, mid) ? 1 : 0; }

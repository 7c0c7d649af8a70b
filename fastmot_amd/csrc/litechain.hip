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
#include <cstdlib>
#include <cstring>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ChainGaps {
    float* p[4];
};

// profiling build only (-DFM_LCH_TIMING, scripts/lch_timing.py): s_memtime stamps of workgroup (0, deepest stream)
#ifdef FM_LCH_TIMING
__device__ long long g_lch_stamps[32];
#define LCH_STAMP(i) if (record && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_lch_stamps[i] = __builtin_readcyclecounter();
#define LCH_WALL(i) if (record && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_lch_stamps[i] = wall_clock64();
#else
#define LCH_STAMP(i)
#define LCH_WALL(i)
#endif

// EPT halfs from LDS / to LDS or HBM as one access
template <int EPT> struct HalfVec;
template <> struct HalfVec<8> { typedef uint4 T; };
template <> struct HalfVec<4> { typedef uint2 T; };
template <> struct HalfVec<2> { typedef uint32_t T; };
template <int EPT>
__device__ __forceinline__ void unpack_n(const typename HalfVec<EPT>::T& v, float* f) {
    typedef f16 hv __attribute__((ext_vector_type(EPT)));
    const hv h = *reinterpret_cast<const hv*>(&v);
#pragma unroll
    for (int e = 0; e < EPT; ++e) f[e] = (float)h[e];
}
template <int EPT>
__device__ __forceinline__ typename HalfVec<EPT>::T pack_n(const float* f) {
    typedef f16 hv __attribute__((ext_vector_type(EPT)));
    hv h;
#pragma unroll
    for (int e = 0; e < EPT; ++e) h[e] = (f16)f[e];
    return *reinterpret_cast<const typename HalfVec<EPT>::T*>(&h);
}

// NTHR = 256 SUB threads: launches with few workgroups (small maps, small batches) leave most CUs idle while every
// workgroup walks its chain alone, so they run with 512 threads (1024 measured slower: the per-item index
// arithmetic is repeated by every thread that shares the item).  The pointwise tiles are dealt to NTHR / 64
// waves; in the depthwise phase SUB threads share a (pixel lane, 8-channel group) item, each taking 8 / SUB of its
// channels -- the pixel sequence of every channel, and with it the order of the gate's partial sums, is the same
// for every SUB: results do not depend on the launch shape.
template <int NT, int KS, int NTHR>
__global__ __launch_bounds__(NTHR, NTHR != 256 ? 1 : NT * KS == 1 ? 4 : NT * KS == 2 ? 3 : 1) void litechain_kernel(
    const f16* __restrict__ in, int in_cs, int in_coff, f16* __restrict__ out, int out_cs, int out_coff_base,
    const f16* __restrict__ wpw_base, int kpad, const f16* __restrict__ wdw_base,
    const float* __restrict__ bias_base, int H, int W, int C, int th, int tw, int tiles_x, int tiles_y, int act,
    ChainGaps gaps, int S, int ys_elems, int zb_elems, int record) {
    extern __shared__ __attribute__((aligned(16))) f16 lds[];
    f16* ys = lds;                       // pointwise output of the current level (halo region)
    f16* zb = lds + ys_elems;            // depthwise output of the previous level = operand of this one
    f16* wd = zb + zb_elems;             // depthwise weights of ALL levels of this stream [D][9][C] ...
    float* bs = reinterpret_cast<float*>(wd + 4 * 9 * 32 * NT);   // ... and their biases [D][C]
    // blockIdx.y = 0 is the DEEPEST stream: workgroups are dispatched in (x, y) order, and a launch with more workgroups
    // than the chip holds at once (400 x 4 against 256 CUs x 3) should start its four-level chains first and fill the
    // tail with the one-level ones (longest-processing-time order), not the other way round.
    const int stream = gridDim.y - 1 - blockIdx.y, D = stream + 1, pset0 = stream * (stream + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const int tile = blockIdx.x % (tiles_x * tiles_y);
    const long n = blockIdx.x / (tiles_x * tiles_y);
    const int ty0 = (tile / tiles_x) * th, tx0 = (tile % tiles_x) * tw;
    const f16* img = in + n * (long)H * W * in_cs + in_coff;
    const int out_coff = out_coff_base + stream * C;
    f16* dst = out + n * (long)H * W * out_cs + out_coff;

    constexpr int SUB = NTHR / 256, EPT = 8 / SUB, NWAVES = NTHR / 64;
    typedef typename HalfVec<EPT>::T hvec;
    const int c8n = C / 8, lanes_px = 256 / c8n;          // phase B: SUB threads keep one 8-channel group
    const int t256 = tid & 255, sub = tid >> 8;
    const int cg = t256 % c8n, pl = t256 / c8n, ch0 = cg * 8 + sub * EPT;
    const bool active = pl < lanes_px;
    float gsum[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) gsum[e] = 0.f;
    LCH_STAMP(0)
    LCH_WALL(30)

    // Every level used to begin with three dependent HBM/L2 round trips (depthwise weights -> LDS, pointwise
    // fragments, biases) in a workgroup that has nothing else to run meanwhile.  The depthwise weights and biases of
    // the whole chain (consecutive parameter sets) go to LDS once, next to the level-0 operand fetch; the pointwise
    // fragments of level l + 1 are requested while level l runs its depthwise phase.
    {
        const f16* wdw = wdw_base + (size_t)pset0 * 9 * C;
        for (int i = tid; i < D * 9 * C / 8; i += NTHR)
            *reinterpret_cast<uint4*>(&wd[i * 8]) = *reinterpret_cast<const uint4*>(wdw + i * 8);
        for (int i = tid; i < D * C; i += NTHR) bs[i] = bias_base[(size_t)pset0 * C + i];
    }
    constexpr bool PREFETCH = NT * KS <= 4;     // wider chains keep one fragment set (register budget)
    f16x8 afr[NT][KS], afr_next[PREFETCH ? NT : 1][PREFETCH ? KS : 1];
#define LCH_LOAD_FRAGS(DST, LVL)                                                                              \
    {                                                                                                         \
        const f16* wpw_ = wpw_base + (size_t)(pset0 + (LVL)) * (32 * NT) * kpad;                              \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                     \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                 \
                DST[nt][ks] = *reinterpret_cast<const f16x8*>(wpw_ + (long)(nt * 32 + frow) * kpad + ks * 16 + fk); \
    }
    LCH_LOAD_FRAGS(afr, 0)

    for (int lvl = 0; lvl < D; ++lvl) {
        const int hl = D - lvl;                           // halo of this level's pointwise region
        // The region [X0, X1) x [Y0, Y1): the tile grown by the halo, CLIPPED to the image plus one ring of padding.  Nothing
        // further out is ever needed -- it is zero at every level -- and for the tiles of OSNet's maps most of the halo
        // lies out there: a 16 x 8 map is one tile (level 0: 18 x 10 positions instead of 24 x 16), the 32 x 16 and 64 x 32
        // maps are two tiles wide (round 5: 1112 -> 720 pointwise and 856 -> 668 depthwise positions per depth-4 chain of
        // the 16 x 8 stage; the values, and the order of the gate's sums, are unchanged).
        const int X0 = max(tx0 - hl, -1), X1 = min(tx0 + tw + hl, W + 1);
        const int Y0 = max(ty0 - hl, -1), Y1 = min(ty0 + th + hl, H + 1);
        const int wp = X1 - X0, hp = Y1 - Y0, npos = wp * hp;
        // the depthwise output region: the next level's pointwise region, or at the last level the tile itself (unclipped:
        // its pixel enumeration is the order of the gate's partial sums, the same as in liteconv.hip)
        const bool last = lvl == D - 1;
        const int X0n = last ? tx0 : max(tx0 - hl + 1, -1), X1n = last ? tx0 + tw : min(tx0 + tw + hl - 1, W + 1);
        const int Y0n = last ? ty0 : max(ty0 - hl + 1, -1), Y1n = last ? ty0 + th : min(ty0 + th + hl - 1, H + 1);
        const int wz = X1n - X0n, hz = Y1n - Y0n;
        const int ax = X0n - 1 - X0, ay = Y0n - 1 - Y0;   // top-left tap of output (0, 0) in this level's region (-1 only where
                                                          // the output lies outside the image and reads nothing)
        // floor(i / d) = (i * ceil(2^16 / d)) >> 16 for i < 2048, d <= 32 (regions are at most 24 x 24)
        const unsigned rcp_wp = (65536u + wp - 1) / wp, rcp_wz = (65536u + wz - 1) / wz;

        // ---- phase A
        const int mtiles = (npos + 31) / 32;
        for (int mt = wave; mt < mtiles; mt += NWAVES) {
            const int pos = mt * 32 + frow, posc = min(pos, npos - 1);
            const int prow = (int)(((unsigned)posc * rcp_wp) >> 16);
            const int py = Y0 + prow, px = X0 + (posc - prow * wp);
            const bool inside = pos < npos && py >= 0 && py < H && px >= 0 && px < W;
            // operand rows: the block input in HBM/L2 at level 0, the previous level's LDS tile afterwards (it
            // covers exactly this region and is zero outside the image).  (Requesting all of a wave's level-0
            // tiles up front was measured SLOWER: 54 -> 62 us for the 64 x 32 stage; the levels are bound by
            // the depthwise phase, not by this round trip.  Round 5 staged the whole level-0 region into `ys` in
            // one trip before the loop and ran level 0 in place from LDS: bit-identical, 383.2 / 381.3 -> 383.1 /
            // 383.4 us per 50-crop pass, i.e. nothing -- the 7.7 k cycles of a workgroup's first phase are the
            // cold start of a launch whose 1024 resident workgroups all miss at once, not these L2 hits.)
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
        LCH_STAMP(16 + lvl)
        __syncthreads();
        LCH_STAMP(1 + 2 * lvl)
        if constexpr (PREFETCH) {
            if (lvl + 1 < D) LCH_LOAD_FRAGS(afr_next, lvl + 1)    // lands during phase B
        }

        // ---- phase B
        const f16* wdl = wd + lvl * 9 * C;
        if (active) {
            // this thread's bias and depthwise taps of the level; the taps stay packed halfs (v_fma_mix_f32 extends both
            // factors): 36 VGPRs less than float copies, which is what lets four workgroups share a CU
            float b8[EPT];
            hvec kh[9];
#pragma unroll
            for (int e = 0; e < EPT; ++e) b8[e] = bs[lvl * C + ch0 + e];
#pragma unroll
            for (int t = 0; t < 9; ++t) kh[t] = *reinterpret_cast<const hvec*>(&wdl[t * C + ch0]);
            for (int pix = pl; pix < wz * hz; pix += lanes_px) {
                const int oy = (int)(((unsigned)pix * rcp_wz) >> 16), ox = pix - oy * wz;
                const int gy = Y0n + oy, gx = X0n + ox;
                const bool in_img = gy >= 0 && gy < H && gx >= 0 && gx < W;
                if (!in_img) {                              // outside the image: the next level's zero padding, nothing to read
                    if (!last) {
                        hvec zero;
                        memset(&zero, 0, sizeof(zero));
                        *reinterpret_cast<hvec*>(&zb[pix * S + ch0]) = zero;
                    }
                    continue;
                }
                float acc[EPT];
#pragma unroll
                for (int e = 0; e < EPT; ++e) acc[e] = b8[e];
                const f16* yp = &ys[((oy + ay) * wp + ox + ax) * S + ch0];
                hvec raw[9];                                // all nine taps requested before the first FMA
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        raw[dy * 3 + dx] = *reinterpret_cast<const hvec*>(yp + (dy * wp + dx) * S);
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    fma_mix_nh<EPT>(reinterpret_cast<const uint32_t*>(&raw[t]), reinterpret_cast<const uint32_t*>(&kh[t]), acc);
                apply_act_n<EPT>(acc, act);
                hvec o = pack_n<EPT>(acc);
                if (!last) {
                    *reinterpret_cast<hvec*>(&zb[pix * S + ch0]) = o;
                } else {
                    *reinterpret_cast<hvec*>(dst + ((long)gy * W + gx) * out_cs + ch0) = o;
                    float r[EPT];   // sums of the STORED (fp16-rounded) activations, as a separate GAP would see them
                    unpack_n<EPT>(o, r);
#pragma unroll
                    for (int e = 0; e < EPT; ++e) gsum[e] += r[e];
                }
            }
        }
        LCH_STAMP(20 + lvl)
        __syncthreads();
        LCH_STAMP(2 + 2 * lvl)
        LCH_WALL(31)
        if (lvl + 1 < D) {
            if constexpr (PREFETCH) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) afr[nt][ks] = afr_next[nt][ks];
            } else {
                LCH_LOAD_FRAGS(afr, lvl + 1)
            }
        }
    }
#undef LCH_LOAD_FRAGS

    // ---- per-tile channel sums -> gaps.p[stream][n][tile][C], summed over the pixel lanes in a fixed order
    float* gap_out = gaps.p[stream];
    if (gap_out) {
        float* red = reinterpret_cast<float*>(ys);          // ys is dead (barrier above)
        if (active) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) red[pl * C + ch0 + e] = gsum[e];
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

// Pixel stride of the LDS tiles, in halfs.  C + 8 where C / 8 is even keeps the 16-byte accesses of 32 consecutive pixels
// off each other's banks; C = 16 goes unpadded: its 32-byte pixels are conflict-free for the 16-byte reads of both phases
// (two-way only for phase A's 8-byte stores), and 35 KB instead of 51 KB per workgroup puts four of them on a CU where
// the 64 x 32 stage of a 50-crop pass has 1600 to run.
static int lds_stride(int C) {
    if (C == 16) return 16;
    return C + (((C >> 3) & 1) ? 0 : 8);
}

// LDS bytes of the chain kernel (graph.py mirrors this to decide whether a block can use it)
size_t litechain_lds_bytes(int C, int W, int H) {
    int th, tw, tx, ty;
    liteconv_tiling(C, W, H, &th, &tw, &tx, &ty);
    const int S = lds_stride(C), nt = (C + 31) / 32;
    const size_t ys = (size_t)(th + 8) * (tw + 8) * S, zb = (size_t)(th + 6) * (tw + 6) * S;
    const size_t red = (size_t)(256 / (C / 8)) * C * 2;                  // phase-C floats, in halfs
    // + depthwise weights [4][9][32 nt] (halfs) and biases [4][32 nt] (floats) of the whole chain
    return ((ys > red ? ys : red) + zb + 4 * 9 * 32 * nt + 4 * 2 * 32 * nt) * sizeof(f16);
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
    const int S = lds_stride(C), nt = (C + 31) / 32, ks = (C + 15) / 16;
    const size_t ys = (size_t)(th + 8) * (tw + 8) * S, red = (size_t)(256 / (C / 8)) * C * 2;
    const int ys_elems = (int)(ys > red ? ys : red), zb_elems = (th + 6) * (tw + 6) * S;
    ChainGaps gaps{};
    for (int i = 0; i < 4; ++i) gaps.p[i] = gap ? gap[i] : nullptr;
    const long wgs = (long)N * tiles_x * tiles_y * 4;
    const dim3 grid((unsigned)(wgs / 4), 4);
    // few workgroups: wider ones (see the kernel)
    const int nthr = wgs <= 384 ? 512 : 256;     // (800: 396.7 vs 397.8 us per 50-crop pass, profiles/r05_osnet_launch_shapes_ab.txt)
    int record = 0;                              // (profiling build: this launch keeps its stamps)
#ifdef FM_LCH_TIMING
    // the stamps of the FASTMOT_LCH_TIMING_LAUNCH-th chain launch of every six (OSNet has six) are the ones kept
    static int launches = 0;
    static const int pick = getenv("FASTMOT_LCH_TIMING_LAUNCH") ? atoi(getenv("FASTMOT_LCH_TIMING_LAUNCH")) : 5;
    if (launches++ % 6 == pick) record = 1;
#endif
#define LCH_LAUNCH_T(NT_, KS_, NTHR_)                                                                             \
    hipLaunchKernelGGL((litechain_kernel<NT_, KS_, NTHR_>), grid, dim3(NTHR_), shmem, s, in, in_cs, in_coff, out, \
                       out_cs, out_coff, wpw, kpad, wdw, bias, H, W, C, th, tw, tiles_x, tiles_y, act, gaps, S,   \
                       ys_elems, zb_elems, record)
#define LCH_LAUNCH(NT_, KS_)                                                                                      \
    {                                                                                                             \
        bool done_ = false;                                                                                       \
        if constexpr ((NT_) * (KS_) <= 4) {                                                                       \
            if (!done_ && nthr >= 512) { LCH_LAUNCH_T(NT_, KS_, 512); done_ = true; }                             \
        }                                                                                                         \
        if (!done_) LCH_LAUNCH_T(NT_, KS_, 256);                                                                  \
    }
    if (ks == 1) LCH_LAUNCH(1, 1)
    else if (nt == 1) LCH_LAUNCH(1, 2)
    else if (nt == 2) { if (ks <= 3) LCH_LAUNCH(2, 3) else LCH_LAUNCH(2, 4) }
    else if (nt == 3) { if (ks <= 5) LCH_LAUNCH(3, 5) else LCH_LAUNCH(3, 6) }
    else { if (ks <= 7) LCH_LAUNCH(4, 7) else LCH_LAUNCH(4, 8) }
#undef LCH_LAUNCH
#undef LCH_LAUNCH_T
    FM_HIP(hipGetLastError());
    return 0;
}

#ifdef FM_LCH_TIMING
extern "C" int fm_debug_lch_stamps(long long* out32) {
    FM_HIP(hipDeviceSynchronize());
    FM_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_lch_stamps), sizeof(long long) * 32));
    return 0;
}
#endif

extern "C" size_t fm_litechain_lds_bytes(int c, int w, int h) {
    return (c % 8 == 0 && c >= 8 && c <= 128 && w > 0 && h > 0) ? litechain_lds_bytes(c, w, h) : (size_t)-1;
}

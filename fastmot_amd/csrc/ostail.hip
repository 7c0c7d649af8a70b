// OSNet x0.25's last stage and head as ONE launch, one workgroup per sample (FM_OP_OSTAIL, round 6; VERDICT r5 item 3,
// DESIGN section 11 3b): transition pool -> OSBlock (96 -> 128, with downsample) -> OSBlock (128 -> 128) -> conv5 -> global
// average pool -> fc (+ BN1d) -> ReLU -> L2 normalise.  torchreid osnet.py OSNet.featuremaps conv4 / conv5 + forward;
// reference call site fastmot/feature_extractor.py:62-74, fastmot/models/reid.py:48-109.
//
// On the 16 x 8 maps of that stage a sample is 128 pixels x <= 128 channels = 32 KB: eleven dependent launches (pool, conv1,
// four-stream chain, gate, conv3 [+ downsample], the same again, conv5, head) of 5-13 us each did a few hundred nanoseconds
// of arithmetic apiece, 76 of the 375 us of a 50-crop pass.  Here a 1024-thread workgroup keeps the sample in LDS from the pool
// to the embedding:
//   * every pointwise conv is a set of (32 pixel x 32 cout) MFMA tiles, one per wave (16 tiles for the 128-cout layers), the
//     weight fragments straight from L2 in fragment order (coalesced 1 KB per K step), the operand rows from the LDS tile;
//   * the four LightConv chains advance level by level together: level l runs the pointwise conv of the 4 - l streams that
//     are that deep on 4 waves each, then their depthwise 3x3 + BN + ReLU with all 1024 threads on (pixel, 4 channels) items
//     over a tile that carries its ring of zero padding (no halo recompute: the whole map is the tile);
//   * the gate (GAP -> fc1 -> ReLU -> fc2 -> sigmoid, shared by the streams), the gated sum, the residual / downsample path
//     and the head read what the previous phase left in LDS.
// Every tensor the unfused layers stored is rounded to fp16 at the same point, the MFMA K order is the unfused kernels'
// order, pool / depthwise / gate / head arithmetic is the unfused kernels' arithmetic; what differs is the order of the fp32
// sums of the two average pools (gate, head), i.e. the last bits of an fp32 mean (tests/test_conv_gpu.py::test_osnet_tail_*).
#include "net.h"
#include <atomic>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// profiling build only (-DFM_OST_TIMING, scripts/ost_timing.py): wall-clock stamps (100 MHz) of workgroup 0, thread 0
#ifdef FM_OST_TIMING
__device__ long long g_ost_stamps[48];
#define OST_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_ost_stamps[i] = wall_clock64();
#else
#define OST_STAMP(i)
#endif

constexpr int MH = 16, MW = 8, HW = MH * MW;          // the stage's map
constexpr int CI = 96, MID = 32, CO = 128, FD = 512;  // stage input / block mid / stage output channels, embedding size
constexpr int TS = CO + 8;                            // row stride (halfs) of the 128-channel LDS tiles: conflict-free ds_read_b128
constexpr int S = MID + 8;                            // ... of the 32-channel tiles
constexpr int RW = MW + 2, YPOS = (MH + 2) * RW;      // a stream's pointwise output with its ring of zero padding
constexpr int FRAG = 512;                             // halfs of one (32 cout x 16 k) weight fragment

// parameter blobs (Graph.fuse_ostail writes them in this order; fm_layer.stride / .pad carry the totals as a check)
constexpr int H_W1A = 0;                              // block A conv1   [6 k steps][64 lanes][8]
constexpr int H_PWA = H_W1A + 6 * FRAG;               // block A chain   [10 sets][2][64][8]
constexpr int H_DWA = H_PWA + 20 * FRAG;              //                 [10][9][32]
constexpr int H_W3A = H_DWA + 10 * 9 * MID;           // block A conv3 | downsample over [x2 | x]  [4 cout tiles][8][64][8]
constexpr int H_W1B = H_W3A + 32 * FRAG;              // block B conv1   [8][64][8]
constexpr int H_PWB = H_W1B + 8 * FRAG;
constexpr int H_DWB = H_PWB + 20 * FRAG;
constexpr int H_W3B = H_DWB + 10 * 9 * MID;           // block B conv3   [4][2][64][8]
constexpr int H_W5 = H_W3B + 8 * FRAG;                // conv5           [4][8][64][8]
constexpr int H_FC = H_W5 + 32 * FRAG;                // fc              [512][128]
constexpr int H_TOTAL = H_FC + FD * CO;
constexpr int GATE_F = 2 * MID + 4 + 2 * MID + MID;   // w1 [2][32] | b1 [2] + 2 pad | w2 [32][2] | b2 [32]
constexpr int F_B1A = 0, F_BCA = F_B1A + MID, F_GA = F_BCA + 10 * MID, F_B3A = F_GA + GATE_F, F_B1B = F_B3A + CO,
              F_BCB = F_B1B + MID, F_GB = F_BCB + 10 * MID, F_B3B = F_GB + GATE_F, F_B5 = F_B3B + CO, F_FC = F_B5 + CO,
              F_TOTAL = F_FC + FD;

// LDS (halfs): R1 | EX | R2 | ZB | dw taps, then floats.  R1 / R2: 128-channel tiles; the four streams' padded pointwise outputs
// (YS) need more than one of them and overlay the one that is dead during a chain plus EX: block A's chain uses EX + R2 (R1
// holds the concat [x2 | x] the downsample path still needs), block B's chain R1 + EX (R2 holds the residual).
constexpr int L_R1 = 0, L_EX = L_R1 + HW * TS, L_R2 = L_R1 + 4 * YPOS * S, L_ZB = L_R2 + HW * TS, L_DW = L_ZB + 4 * HW * S,
              L_HALFS = L_DW + 10 * 9 * MID;
constexpr int LF_BC = 0, LF_GP = LF_BC + 10 * MID, LF_GAP = LF_GP + GATE_F, LF_HID = LF_GAP + 4 * MID, LF_GATE = LF_HID + 8, LF_HGAP = LF_GATE + 4 * MID,
              LF_FEAT = LF_HGAP + CO, LF_RED = LF_FEAT + FD, LF_PART = LF_RED + 16, L_FLOATS = LF_PART + 4 * 32 * MID;
constexpr size_t LDS_BYTES = (size_t)L_HALFS * 2 + (size_t)L_FLOATS * 4;
static_assert(HW * TS * 2 >= 64 * CO * 4, "the head's partial sums live in R2");
static_assert(L_HALFS % 8 == 0 && LDS_BYTES <= 160 * 1024, "LDS budget");

// One wave: D[32 cout][32 pixels] = W[32][16 KS] X[16 KS][32].  wf: the cout tile's fragments [KS][64][8] (global: one
// coalesced 1 KB request per K step, issued a phase ahead of their use -- a workgroup that is alone on its CU has nothing
// else to run during an L2 round trip); xrow: this lane's operand row (pixel lane & 31) + (lane >> 5) * 8 in an LDS tile.
template <int KS>
__device__ __forceinline__ void load_frags(f16x8 (&a)[KS], const f16* __restrict__ wf, int lane) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const f16x8*>(wf + ks * FRAG + lane * 8);
}
template <int KS>
__device__ __forceinline__ f32x16 mm_tile(const f16x8 (&a)[KS], const f16* xrow) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], *reinterpret_cast<const f16x8*>(xrow + ks * 16), acc, 0, 0, 0);
    return acc;
}

// the four bias quads of a lane's outputs, requested BEFORE the next phase's weight fragments: loads return in order, and an
// epilogue that asked for its bias after them would wait for all of them
__device__ __forceinline__ void load_bias(float4 (&b)[4], const float* __restrict__ bias, int fh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const float4*>(bias + g * 8 + fh * 4);
}

// acc (+ bias) (+ residual already in the destination) -> act -> fp16 -> drow[c0 .. c0 + 4), c0 = 8 g + 4 (lane >> 5)
template <bool BIAS, bool RES, bool RELU>
__device__ __forceinline__ void store_tile(const f32x16& acc, const float4 (&bias)[4], f16* drow, int fh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = g * 8 + fh * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[g * 4 + e];
        if constexpr (BIAS) {
            v[0] += bias[g].x; v[1] += bias[g].y; v[2] += bias[g].z; v[3] += bias[g].w;
        }
        union { f16 h[4]; uint2 u; } pk;
        if constexpr (RES) {
            pk.u = *reinterpret_cast<const uint2*>(drow + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)pk.h[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pk.h[e] = (f16)(RELU ? fmaxf(v[e], 0.f) : v[e]);
        *reinterpret_cast<uint2*>(drow + c0) = pk.u;
    }
}

// Sum over groups of N = 8 / 16 adjacent lanes, every lane ending with its group's total: DPP adds (quad swaps, then the
// half-row and row mirrors pair the 4- and 8-lane groups) instead of __shfl_xor's ds_bpermute round trips through the LDS
// crossbar (~100 cycles each, and these reductions are chains of three to five).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float t) {
    return t + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), CTRL, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float lanes_sum(float t) {
    t = dpp_add<0xB1>(t);                      // quad_perm [1, 0, 3, 2]
    t = dpp_add<0x4E>(t);                      // quad_perm [2, 3, 0, 1]
    if constexpr (N >= 8) t = dpp_add<0x141>(t);    // row_half_mirror
    if constexpr (N >= 16) t = dpp_add<0x140>(t);   // row_mirror
    return t;
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() carries a release fence, i.e. s_waitcnt vmcnt(0): every weight
// request in flight -- issued a phase ahead precisely so that it is NOT waited for here -- would be drained at each of the
// kernel's ~30 barriers.  Nothing in this kernel exchanges data through global memory.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void zero_halfs(f16* p, int n8, int tid) {
    for (int i = tid; i < n8; i += 1024) reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Depthwise 3x3 + BN + ReLU of the `nact` streams alive at level lvl.  An item = (stream, PXW consecutive pixels of a row, 4
// channels): its 3 x (PXW + 2) input positions and nine taps are read once for PXW outputs (per output pixel 18 LDS reads
// -> 6.75 at PXW = 4; one-pixel items measured 8.8 us per chain, four-pixel items 6.9; two-pixel items for the last two
// levels, where four-pixel items leave a SIMD one or two waves, were slower: the phase is bound by its instruction count).  A stream that ends at this level
// leaves the channel sums of its 32 half rows (of the STORED fp16 values, as a separate average pool would see them) in
// part[s][32][32]: a four-pixel item is a half row; the two two-pixel items of one are 8 lanes apart and add by a shuffle.
template <int PXW>
__device__ __forceinline__ void dw_items(const f16* ys, f16* zb, const f16* dwl, const float* bcl, float* part, int lvl, int nact,
                                         int tid) {
    constexpr int IPS = HW / PXW * 8;          // items per stream: 256 / 512
    if (tid < nact * IPS) {
        const int s = lvl + tid / IPS, r = tid % IPS, ch0 = (r & 7) * 4, strip = r >> 3;
        const int y = strip / (MW / PXW), x0 = (strip % (MW / PXW)) * PXW;
        const int pset = s * (s + 1) / 2 + lvl;
        const f16* yp = ys + (s * YPOS + y * RW + x0) * S + ch0;             // tap (-1, -1) of the item's first pixel
        uint2 raw[3][PXW + 2], kh[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < PXW + 2; ++dx) raw[dy][dx] = *reinterpret_cast<const uint2*>(yp + (dy * RW + dx) * S);
#pragma unroll
        for (int t = 0; t < 9; ++t) kh[t] = *reinterpret_cast<const uint2*>(dwl + (pset * 9 + t) * MID + ch0);
        const float4 bias = *reinterpret_cast<const float4*>(bcl + pset * MID + ch0);
        float gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < PXW; ++p) {
            float acc[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    fma_mix_nh<4>(reinterpret_cast<const uint32_t*>(&raw[dy][p + dx]),
                                  reinterpret_cast<const uint32_t*>(&kh[dy * 3 + dx]), acc);
            union { f16 h[4]; uint2 u; } pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk.h[e] = (f16)fmaxf(acc[e], 0.f);
            *reinterpret_cast<uint2*>(zb + (s * HW + y * MW + x0 + p) * S + ch0) = pk.u;
            if (lvl == s) {                    // (wave-uniform)
#pragma unroll
                for (int e = 0; e < 4; ++e) gs[e] += (float)pk.h[e];
            }
        }
        if (lvl == s) {                        // the stream ends here: 32 half-row sums per channel for the gate's average pool
            if constexpr (PXW == 2) {          // the other two pixels of the half row: the item 8 lanes up (strip ^ 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = __shfl_xor(gs[e], 8);
                    gs[e] = (strip & 1) ? o + gs[e] : gs[e] + o;
                }
            }
            if (PXW == 4 || !(strip & 1))
                *reinterpret_cast<float4*>(part + (s * 32 + strip * PXW / 4) * MID + ch0) = make_float4(gs[0], gs[1], gs[2], gs[3]);
        }
    }
}

// The four streams of an OSBlock over y1 (= zb stream 0's tile on entry) -> zb[s] = stream s (depth s + 1) and part[s][32][32] =
// its per-channel sums over the 32 half rows.  ys must be zero
// on its rings; dwl / bcl: this block's depthwise taps [10][9][32] and biases [10][32] in LDS; a0: the level-0 pointwise
// fragments of this wave's stream (pw + pset * 2 * FRAG, pset = s (s + 1) / 2, s = wave / 4), requested by the caller.
__device__ __forceinline__ void os_chain(const f16* __restrict__ pw, f16x8 (&a0)[2], const f16* dwl, const float* bcl, f16* ys,
                                         f16* zb, float* part, int tid, int stamp0) {
    const int lane = tid & 63, wave = tid >> 6, frow = lane & 31, fh = lane >> 5;
    (void)stamp0;
    const float4 nob[4] = {};
    f16x8 an[2];
    for (int lvl = 0; lvl < 4; ++lvl) {
        const int nact = 4 - lvl;
        if (wave < nact * 4) {
            const int s = lvl + (wave >> 2), mt = wave & 3, px = mt * 32 + frow;
            const f16* xrow = zb + ((lvl == 0 ? 0 : s * HW) + px) * S + fh * 8;
            const f32x16 acc = mm_tile<2>(a0, xrow);
            store_tile<false, false, false>(acc, nob, ys + (s * YPOS + ((px >> 3) + 1) * RW + (px & 7) + 1) * S, fh);
        }
        lds_barrier();
        OST_STAMP(stamp0 + 2 * lvl)
        if (wave < (nact - 1) * 4) {          // the next level's fragments land during the depthwise phase
            const int s = lvl + 1 + (wave >> 2);
            load_frags<2>(an, pw + (s * (s + 1) / 2 + lvl + 1) * 2 * FRAG, lane);
        }
        dw_items<4>(ys, zb, dwl, bcl, part, lvl, nact, tid);
        a0[0] = an[0];
        a0[1] = an[1];
        lds_barrier();
        OST_STAMP(stamp0 + 2 * lvl + 1)
    }
}

// The aggregation gate (GAP -> fc1 -> ReLU -> fc2 -> sigmoid, shared by the streams) and the gated sum of zb's four streams
// -> x2[pixel * TS + c], c < 32.  gp: fc1 w [2][32] | b [2] + 2 pad | fc2 w [32][2] | b [32] (LDS copy).
__device__ __forceinline__ void os_gate(const float* gp, const f16* zb, const float* part, float* fs, f16* x2, int tid) {
    // average pool of the stored (fp16) stream outputs: the chain left 32 half-row sums per (stream, channel) in `part`
    float* gap = fs + LF_GAP;
    float* gate = fs + LF_GATE;
    float* hidden = fs + LF_HID;
    {   // average: 8 lanes per (stream, channel), four half rows each, added across the lanes
        const int s = tid >> 8, c = (tid >> 3) & 31, j = tid & 7;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) t += part[(s * 32 + j * 4 + g) * MID + c];
        t = lanes_sum<8>(t);
        if (j == 0) gap[s * MID + c] = t / (float)HW;
    }
    lds_barrier();
    if (tid < 256) {   // fc1 + ReLU: 32 lanes per (stream, hidden unit), one product each
        const int s = tid >> 6, h = (tid >> 5) & 1, k = tid & 31;
        float t = gp[h * MID + k] * gap[s * MID + k];
        t = lanes_sum<16>(t);
        t += __shfl_xor(t, 16);
        t += gp[2 * MID + h];
        if (k == 0) hidden[s * 2 + h] = t > 0.f ? t : 0.f;
    }
    lds_barrier();
    if (tid < 4 * MID) {   // fc2 + sigmoid
        const int s = tid >> 5, c = tid & 31;
        float t = gp[2 * MID + 4 + 2 * MID + c];
#pragma unroll
        for (int h = 0; h < 2; ++h) t = fmaf(gp[2 * MID + 4 + c * 2 + h], hidden[s * 2 + h], t);
        gate[tid] = 1.f / (1.f + __expf(-t));
    }
    lds_barrier();
    {
        const int px = tid >> 3, q = tid & 7;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            union { f16 h[4]; uint2 u; } pk;
            pk.u = *reinterpret_cast<const uint2*>(zb + (s * HW + px) * S + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf((float)pk.h[e], gate[s * MID + q * 4 + e], o[e]);
        }
        union { f16 h[4]; uint2 u; } pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk.h[e] = (f16)o[e];
        *reinterpret_cast<uint2*>(x2 + px * TS + q * 4) = pk.u;
    }
    lds_barrier();
}

__global__ __launch_bounds__(1024) void ostail_kernel(const f16* __restrict__ in, int in_cs, int in_coff,
                                                      const f16* __restrict__ ph, const float* __restrict__ pf,
                                                      float* __restrict__ out, float* __restrict__ raw_out,
                                                      float* __restrict__ mirror) {
    extern __shared__ __attribute__((aligned(16))) f16 lds[];
    f16* R1 = lds + L_R1;
    f16* EX = lds + L_EX;
    f16* R2 = lds + L_R2;
    f16* ZB = lds + L_ZB;
    f16* DW = lds + L_DW;
    float* fs = reinterpret_cast<float*>(lds + L_HALFS);
    float* part = fs + LF_PART;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 31, fh = lane >> 5;
    const int mt = wave & 3, nt = wave >> 2, px = mt * 32 + frow;       // this wave's tile of the 16-tile pointwise convs
    const int cs = wave >> 2;                                            // ... and its stream at a chain's level 0
    const long n = blockIdx.x;
    f16x8 a2[2], a6[6], a8[8], b8[8];
    OST_STAMP(0)
    float4 bb[4];

    // ---- transition pool (AvgPool2d(2, 2) of the 32 x 16 x 96 map) -> x = R1[:, 32 .. 128), as ops.hip pool_kernel
    {
        const f16* img = in + n * (4 * HW) * (long)in_cs + in_coff;
        uint4 v[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = min(tid + it * 1024, HW * (CI / 8) - 1), cg = i % (CI / 8), p = i / (CI / 8);
            const f16* src = img + ((long)(2 * (p >> 3)) * (2 * MW) + 2 * (p & 7)) * in_cs + cg * 8;
#pragma unroll
            for (int d = 0; d < 4; ++d) v[it][d] = *reinterpret_cast<const uint4*>(src + ((d >> 1) * (2 * MW) + (d & 1)) * (long)in_cs);
        }
        if (wave < 4) load_frags<6>(a6, ph + H_W1A, lane);
        load_bias(bb, pf + F_B1A, fh);
        load_frags<2>(a2, ph + H_PWA + cs * (cs + 1) / 2 * 2 * FRAG, lane);
        zero_halfs(EX, 4 * YPOS * S / 8, tid);                      // block A's YS = EX + R2
        for (int i = tid; i < 10 * 9 * MID / 8; i += 1024)
            reinterpret_cast<uint4*>(DW)[i] = *reinterpret_cast<const uint4*>(ph + H_DWA + i * 8);
        for (int i = tid; i < 10 * MID + GATE_F; i += 1024) fs[LF_BC + i] = pf[F_BCA + i];      // chain biases | gate parameters
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int i = tid + it * 1024;
            if (i < HW * (CI / 8)) {
                const int cg = i % (CI / 8), p = i / (CI / 8);
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    float f[8];
                    unpack8(v[it][d], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += f[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] *= 0.25f;
                *reinterpret_cast<uint4*>(R1 + p * TS + MID + cg * 8) = pack8(acc);
            }
        }
    }
    lds_barrier();
    OST_STAMP(1)
    // ---- block A: conv1 (96 -> 32) -> y1 = ZB stream 0
    if (wave < 4) {
        const int p = wave * 32 + frow;
        const f32x16 acc = mm_tile<6>(a6, R1 + p * TS + MID + fh * 8);
        store_tile<true, false, true>(acc, bb, ZB + p * S, fh);
    }
    lds_barrier();
    OST_STAMP(2)
    os_chain(ph + H_PWA, a2, DW, fs + LF_BC, EX, ZB, part, tid, 16);
    OST_STAMP(3)
    load_frags<8>(a8, ph + H_W3A + nt * 8 * FRAG, lane);            // conv3 | downsample, under the gate
    load_bias(bb, pf + F_B3A + nt * 32, fh);
    os_gate(fs + LF_GP, ZB, part, fs, R1, tid);
    OST_STAMP(4)
    // ---- conv3 + downsample: relu([W3 | Wd] [x2 | x] + b) -> R2
    {
        const f32x16 acc = mm_tile<8>(a8, R1 + px * TS + fh * 8);
        __builtin_amdgcn_sched_barrier(0);                          // (the requests below reuse the registers of a8)
        if (wave < 4) load_frags<8>(b8, ph + H_W1B, lane);          // block B's conv1 and level-0 pointwise fragments
        load_frags<2>(a2, ph + H_PWB + cs * (cs + 1) / 2 * 2 * FRAG, lane);
        store_tile<true, false, true>(acc, bb, R2 + px * TS + nt * 32, fh);
        load_bias(bb, pf + F_B1B, fh);
    }
    for (int i = tid; i < 10 * 9 * MID / 8; i += 1024)          // block B's depthwise taps, biases and gate (block A's are dead)
        reinterpret_cast<uint4*>(DW)[i] = *reinterpret_cast<const uint4*>(ph + H_DWB + i * 8);
    for (int i = tid; i < 10 * MID + GATE_F; i += 1024) fs[LF_BC + i] = pf[F_BCB + i];
    lds_barrier();
    OST_STAMP(5)
    // ---- block B: conv1 (128 -> 32) -> y1; its YS = R1 + EX is cleared meanwhile (the concat is dead)
    if (wave < 4) {
        const int p = wave * 32 + frow;
        const f32x16 acc = mm_tile<8>(b8, R2 + p * TS + fh * 8);
        store_tile<true, false, true>(acc, bb, ZB + p * S, fh);
    }
    zero_halfs(R1, 4 * YPOS * S / 8, tid);
    lds_barrier();
    OST_STAMP(6)
    os_chain(ph + H_PWB, a2, DW, fs + LF_BC, R1, ZB, part, tid, 24);
    OST_STAMP(7)
    load_frags<2>(a2, ph + H_W3B + nt * 2 * FRAG, lane);            // conv3 and conv5, under the gate
    load_bias(bb, pf + F_B3B + nt * 32, fh);
    load_frags<8>(a8, ph + H_W5 + nt * 8 * FRAG, lane);
    float4 b5[4];
    load_bias(b5, pf + F_B5 + nt * 32, fh);
    os_gate(fs + LF_GP, ZB, part, fs, R1, tid);
    OST_STAMP(8)
    // ---- conv3 (32 -> 128) + identity -> relu, in place in R2; this thread's pieces of the fc matrix are requested here, two
    // phases ahead (128 KB per workgroup, cold: it takes microseconds to arrive).  The matrix is read the way it lies in
    // memory -- iteration j, thread t: 16 bytes at halfs (j * 1024 + t) * 8, a wave = four whole rows, 1 KB contiguous.
    uint4 wrow[8];
    float fcb[8];
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) wrow[j] = *reinterpret_cast<const uint4*>(ph + H_FC + (long)(j * 1024 + tid) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) fcb[j] = pf[F_FC + j * 64 + (tid >> 4)];
        const f32x16 acc = mm_tile<2>(a2, R1 + px * TS + fh * 8);
        store_tile<true, true, true>(acc, bb, R2 + px * TS + nt * 32, fh);
    }
    lds_barrier();
    OST_STAMP(9)
    // ---- conv5 (128 -> 128) -> R1
    {
        const f32x16 acc = mm_tile<8>(a8, R2 + px * TS + fh * 8);
        store_tile<true, false, true>(acc, b5, R1 + px * TS + nt * 32, fh);
    }
    lds_barrier();
    OST_STAMP(10)
    // ---- head: global average pool -> fc + BN1d -> relu -> L2 normalise (ops.hip head_kernel)
    float* hpart = reinterpret_cast<float*>(R2);
    float* hgap = fs + LF_HGAP;
    float* feat = fs + LF_FEAT;
    float* red = fs + LF_RED;
    {
        const int cg = tid & 15, pg = tid >> 4;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(R1 + (pg + 64 * j) * TS + cg * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) hpart[pg * CO + cg * 8 + e] = a[e];
    }
    lds_barrier();
    {   // 8 lanes per channel, eight of the 64 partial sums each
        const int c = tid >> 3, j = tid & 7;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += hpart[(j * 8 + g) * CO + c];
        t = lanes_sum<8>(t);
        if (j == 0) hgap[c] = t / (float)HW;
    }
    lds_barrier();
    // fc: lane (tid & 15) holds K chunk [8 (tid & 15), + 8) of row j * 64 + tid / 16; the 16 lanes of a row add up by shuffles
    float nrm = 0.f;
    {
        const int kc = (tid & 15) * 8;
        float g8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g8[e] = hgap[kc + e];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float k[8], t = 0.f;
            unpack8(wrow[j], k);
#pragma unroll
            for (int e = 0; e < 8; ++e) t = fmaf(k[e], g8[e], t);
            t = lanes_sum<16>(t);
            const int d = j * 64 + (tid >> 4);
            t += fcb[j];
            t = t > 0.f ? t : 0.f;
            if ((tid & 15) == 0) {
                feat[d] = t;
                nrm += t * t;
            }
        }
    }
    nrm = lanes_sum<16>(nrm);
    nrm += __shfl_xor(nrm, 16);
    nrm += __shfl_xor(nrm, 32);
    if (lane == 0) red[wave] = nrm;
    lds_barrier();
    float total = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) total += red[w];
    const float inv = 1.f / sqrtf(total);
    if (tid < FD) {
        const float f = feat[tid];
        if (raw_out) raw_out[n * FD + tid] = f;
        const float v = f * inv;
        out[n * FD + tid] = v;
        if (mirror) mirror[n * FD + tid] = v;
    }
    OST_STAMP(11)
}

}  // namespace

bool ostail_supported(int in_h, int in_w, int cin, int mid, int cout, int feat, int n_halfs, int n_floats) {
    return in_h == 2 * MH && in_w == 2 * MW && cin == CI && mid == MID && cout == CO && feat == FD && n_halfs == H_TOTAL &&
           n_floats == F_TOTAL;
}

// in: the last transition's conv output [N][32][16][>= 96] (fp16 NHWC view); ph / pf: the parameter blobs above;
// out / mirror: [N][512] fp32 L2-normalised embeddings (device / page-locked host copy, may be null); raw_out: before the norm
int launch_ostail(const f16* in, int in_cs, int in_coff, const f16* ph, const float* pf, int N, float* out, float* raw_out,
                  float* mirror, hipStream_t s) {
    FM_CHECK_ARG(in && ph && pf && out && N >= 1 && in_cs % 8 == 0 && in_coff % 8 == 0 && in_coff + CI <= in_cs);
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    FM_HIP(hipGetDevice(&dev));
    if (dev >= 64 || !(configured.load(std::memory_order_relaxed) >> dev & 1)) {
        FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ostail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)LDS_BYTES));
        if (dev < 64) configured.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(ostail_kernel, dim3(N), dim3(1024), LDS_BYTES, s, in, in_cs, in_coff, ph, pf, out, raw_out, mirror);
    FM_HIP(hipGetLastError());
    return 0;
}

#ifdef FM_OST_TIMING
extern "C" int fm_debug_ost_stamps(long long* out48) {
    FM_HIP(hipDeviceSynchronize());
    FM_HIP(hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_ost_stamps), sizeof(long long) * 48));
    return 0;
}
#endif

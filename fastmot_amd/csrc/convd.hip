// Implicit-GEMM convolution with BOTH operands moved global -> LDS by the DMA path (buffer_load_dwordx4 ... lds) and a
// multi-accumulator tile per wavefront (FM_OP_CONVD).
//
// Replaces the TensorRT engines of the reference like conv.hip does (YOLOv4: fastmot/models/yolo.py:106-151, OSNet:
// fastmot/models/reid.py:48-92; graph semantics scripts/yolo2onnx.py:558-705): Conv + folded BatchNorm + activation
// (+ shortcut, + concat by channel offset, + fused x2 upsample, fp32 heads) in ONE launch.
//
// Why a second tiled kernel: conv.hip stages operands global -> 32 named VGPRs ->
// ds_write, gives every wave ONE 32x32 accumulator (two ds_read_b128 per MFMA: the LDS read port alone caps it at ~50 %
// of the matrix pipe) and needs a separate reduce launch to split K.  Measured there: ~900 cycles per 64-deep K step
// for 128 cycles of MFMA, SQ_WAIT_ANY > SQ_ACTIVE_INST_ANY on every instance.  Here:
//   * a wave owns MC x MP accumulators of 32 x 32 (up to 2 x 2 = a 64 x 64 register tile: one ds_read_b128 per MFMA),
//     a workgroup's four waves a (WC*MC*32) couts x (WP*MP*32) pixels tile;
//   * operands never pass through VGPRs: one buffer_load_dwordx4 ... lds moves 1 KB (8 rows of 128 B = 64 halfs of K)
//     per wave instruction straight into the LDS ring.  Weights are pre-packed per (32-cout block, K step) as the exact
//     4 KB LDS image (models/graph.py _pack_tile64); a pixel row of a K step is 128 contiguous bytes of the NHWC input
//     (Cin % 64 == 0: a step never straddles a (kh, kw) tap).  Taps outside the image are NOT selected away after the
//     load: the lane's offset is replaced by one beyond the buffer descriptor's range and the hardware writes zeros;
//   * LDS image: row = 128 B, its eight 16-byte slots XOR-swizzled with (row / 2) % 8 -- a 16-lane group of
//     ds_read_b128 (rows r .. r+15, one K chunk) then covers all 64 banks exactly once.  The DMA writes lane-linear,
//     so the swizzle is applied to the SOURCE: the weights are stored swizzled, and a lane fetches the input chunk
//     slot ^ swizzle of its pixel (same 128-byte line: coalescing is unchanged);
//   * a ring of NS (2..4) stages per K group with COUNTED waits: s_waitcnt vmcnt((NS - 2) * pieces) + ONE s_barrier per
//     K step; the step that frees a slot is followed at once by the DMA that refills it;
//   * K groups (KG = 1, 2, 4): KG x 4 waves per workgroup, group g walks the g-th contiguous part of the K range with
//     its own ring and accumulators (two or four waves per SIMD where a batch-1 layer has too few tiles for that), the
//     partial tiles are summed through LDS in a fixed order (deterministic) -- no workspace, no reduce launch.
// Epilogue as in conv.hip: the fp32 tile is transposed through LDS and leaves as 16-byte pieces of whole NHWC rows.
//
// Roofline per layer: max(2 K Cout P / 2.5 PFLOP/s, (in + out + weights) * 2 B / 8 TB/s); per workgroup the L2 -> LDS
// fill path (~64 B/clk/CU) bounds a tile at (BM + BN) * K * 2 B / 64 cycles.
#include "net.h"
#include <atomic>
#include <cstdlib>

int launch_convd(const ConvParams& p, hipStream_t s);

namespace {

int g_convd_cfg = 0;     // fm_ctx option "convd_cfg": bm | bn << 8 | kg << 16 | ns << 20 forces one configuration (A/B runs)

// profiling build only (-DFM_CONVD_TIMING, scripts/convd_timing.py): cycle stamps of wave 0 of two workgroups, and
// ablations (g_convd_abl bit 0: no DMA inside the K loop, bit 1: no fragment reads / MFMAs, bit 2: no output phase)
#ifdef FM_CONVD_TIMING
__device__ long long g_convd_stamps[2][264];
int g_convd_abl = 0;
// (stamps go to LDS behind the ring and leave at the end: a global store would count on vmcnt like the DMA does)
#define CONVD_STAMP(i) if (stamp_slot >= 0 && (i) < 264) reinterpret_cast<long long*>(smem + stamp_off)[(i)] = __builtin_readcyclecounter();
#define CONVD_ABL(bit) (abl & (bit))
#define CONVD_EXTRA_PARAM , const int abl, const int stamp_off
#define CONVD_EXTRA_ARG , g_convd_abl, (int)lds
#define CONVD_EXTRA_LDS 4096
#else
#define CONVD_STAMP(i)
#define CONVD_ABL(bit) false
#define CONVD_EXTRA_PARAM
#define CONVD_EXTRA_ARG
#define CONVD_EXTRA_LDS 0
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr;

// TAPS: 0 = 1x1 conv without padding (no tap walk, no masks), 3 = 3x3 conv (any stride / padding)
// ROLE: 1 = the workgroup gets as many LOADER waves as it has MFMA waves.  The loaders own the DMA (address state, issue,
//       counted waits), the MFMA waves only read fragments and multiply: a loader's ~50-70 issue cycles per 1 KB piece then
//       overlap its SIMD partner's MFMAs instead of preceding them in the same instruction stream.
// SPB:  K steps per barrier (1 or 2).  With one 32 x 32 accumulator (or two) per wave a step is only 4 (8) MFMAs: the
//       barrier, the counted wait and the latency of the first fragment reads then cost more than the multiplications;
//       two steps per ring slot halve them (the ring holds nslots slots of SPB steps).
template <int WC, int WP, int MC, int MP, int KG, int TAPS, int ROLE, int SPB>
__global__ __launch_bounds__(256 * KG * (1 + ROLE), (KG == 1 && !ROLE) ? 2 : 1) void convd_kernel(const ConvParams p, const int ns, const int nslots CONVD_EXTRA_PARAM) {
    static_assert(WC * WP == 4, "4 waves per K group");
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass has no buffer-resource type: it only needs the launch stub)
    constexpr int BM = WC * MC * 32, BN = WP * MP * 32;
    constexpr int NPA = BM / 32, NPB = BN / 32;     // DMA pieces (1 KB = 8 rows) per wave and K step
    constexpr int PPS = NPA + NPB;
    constexpr int SS = (BM + BN) * 128;             // bytes of one ring stage: [BM weight rows][BN pixel rows] x 128 B
    constexpr int LDO = BM + 4;
    constexpr int T = 256 * KG * (1 + ROLE);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane((tid >> 8) % KG);    // K group
    const int w = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);     // wave inside the group
    // (launch bounds: with one wave per SIMD hipcc put the accumulators into AGPRs and copied all of them to and from
    // VGPRs around every K step's MFMAs -- 128 v_accvgpr moves per step; a budget of 256 registers keeps them in place)
    const bool loader = ROLE ? __builtin_amdgcn_readfirstlane(tid >> 8) >= KG : true;
    const bool mfma_wave = ROLE ? !loader : true;
    const int wc = w / WP, wp = w % WP;

    // ---- XCD-aware tile order (see conv.hip): XCD i gets the i-th contiguous chunk of an order in which the heavier
    // operand's slice is private to it
    int tile_p, tile_c;
    {
        const int total = p.grid_p * p.grid_c;
        const int chunk = (total + 7) >> 3;
        const int logical = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if (logical >= total) return;
        if (p.weight_major) {              // logical = tile_c * grid_p + tile_p
            tile_c = idiv_small(logical, p.grid_p, 1.f / (float)p.grid_p);
            tile_p = logical - tile_c * p.grid_p;
        } else {                           // logical = tile_p * grid_c + tile_c
            tile_p = idiv_small(logical, p.grid_c, 1.f / (float)p.grid_c);
            tile_c = logical - tile_p * p.grid_c;
        }
    }
    const int c0 = tile_c * BM, p0 = tile_p * BN;
#ifdef FM_CONVD_TIMING
    const int stamp_slot = tid == 0 ? (blockIdx.x == 0 ? 0 : (blockIdx.x == 8 * 5 + 3 ? 1 : -1)) : -1;
#endif
    CONVD_STAMP(0)
#ifdef FM_CONVD_TIMING
    if (stamp_slot >= 0) reinterpret_cast<long long*>(smem + stamp_off)[262] = __builtin_amdgcn_s_memrealtime();   // 100 MHz wall clock
#endif

    // ---- K range of this group
    const int nk = p.Kpad >> 6;
    const int per = (nk + KG - 1) / KG;
    const int s0 = g * per;
    const int nkg = max(0, min(per, nk - s0));

    // ---- buffer descriptors.  num_records is huge on purpose: only the out-of-image marker is ever out of range.
    // The input descriptor starts (pad, pad) pixels BEFORE the tensor so that every lane offset is non-negative (the range
    // check looks at the lane offset alone); those bytes are never touched -- their taps are masked.
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, 0x7fffffff, 0x00020000);
    const char* inb = reinterpret_cast<const char*>(p.in) - (size_t)(p.pad * p.W + p.pad) * p.in_cs * 2;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), 0, 0x7fffffff, 0x00020000);

    // weights: image [cout block of 32][K step][32 rows][128 B]; wave w moves the w-th KB of block i of the tile
    const unsigned va = (unsigned)w * 1024u + (unsigned)lane * 16u;
    unsigned sa_blk[NPA];
    {
        const int nblk = (p.Cout + 31) >> 5;
#pragma unroll
        for (int i = 0; i < NPA; ++i) sa_blk[i] = (unsigned)min(tile_c * NPA + i, nblk - 1) * (unsigned)nk * 4096u;
    }
    // pixels: lane -> (pixel 8 * piece + lane / 8, 16-byte slot lane % 8); slot s of row r holds K chunk s ^ ((r / 2) % 8)
    const int r8 = lane >> 3;
    const unsigned chunk = (unsigned)((lane & 7) ^ (((w & 1) << 2) | (r8 >> 1)));
    unsigned vb[NPB], vm[NPB];
    {
        const int hw = p.Ho * p.Wo;
        const float inv_hw = 1.f / (float)hw, inv_wo = 1.f / (float)p.Wo;
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int pix = p0 + (w + 4 * i) * 8 + r8;
            const bool ok = pix < p.P;
            const int pc = min(pix, p.P - 1);
            const int n = idiv_small(pc, hw, inv_hw), rem = pc - n * hw;
            const int oy = idiv_small(rem, p.Wo, inv_wo), ox = rem - oy * p.Wo;
            vb[i] = ((unsigned)((n * p.H + oy * p.stride) * p.W + ox * p.stride) * (unsigned)p.in_cs + (unsigned)p.in_coff) * 2u +
                    chunk * 16u;
            vm[i] = 0;
            if constexpr (TAPS) {
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                unsigned colok = 0, m = 0;
#pragma unroll
                for (int k = 0; k < TAPS; ++k) colok |= ((unsigned)(ix0 + k) < (unsigned)p.W ? 1u : 0u) << k;
#pragma unroll
                for (int k = 0; k < TAPS; ++k) m |= ((unsigned)(iy0 + k) < (unsigned)p.H ? colok : 0u) << (k * TAPS);
                vm[i] = ok ? m : 0u;
            } else {
                if (!ok) vb[i] = OOB;
            }
        }
    }

    // ---- issue state of this wave: (kh, kw, c0) of the next K step to request, its scalar offsets, its ring slot
    int i_kh = 0, i_kw = 0, i_c0 = s0 * 64;
    if constexpr (TAPS) {
        const int tap = i_c0 / p.Cin;
        i_c0 -= tap * p.Cin;
        i_kh = tap / TAPS;
        i_kw = tap - i_kh * TAPS;
    }
    const bool ragged = TAPS == 0 && (p.Cin & 63) != 0;
    unsigned i_soff = (unsigned)((i_kh * p.W + i_kw) * p.in_cs + i_c0) * 2u;
    unsigned i_astep = (unsigned)s0 * 4096u;
    int i_stage = 0, i_step = 0;
    char* const ring = smem + g * nslots * (SPB * SS);
    // requests the SPB steps of the next ring slot.  A slot always receives SPB * PPS pieces, so that the counted waits
    // stay exact: when the group's last slot has a step too few (odd K range), the last real step is fetched once more
    // into the unused half (never multiplied).
    auto issue = [&]() {
#pragma unroll
        for (int u = 0; u < SPB; ++u) {
            char* la = ring + (i_stage * SPB + u) * SS;
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(la + (i * 4 + w) * 1024), 16, va, sa_blk[i] + i_astep, 0, 0);
            char* lb = la + BM * 128;
            const int t = i_kh * TAPS + i_kw;
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                unsigned vo = vb[i];
                if constexpr (TAPS) vo = ((vm[i] >> t) & 1u) ? vo : OOB;
                // 1x1 with cin % 64 != 0 (OSNet's 16 .. 96-channel layers): the K range is padded to whole steps with
                // zero weights; the chunks of a step that lie beyond the pixel's channels are not fetched (zeros)
                else if (ragged) vo = (int)chunk < ((p.Cin - i_c0 + 7) >> 3) ? vo : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(lb + (w + 4 * i) * 1024), 16, vo, i_soff, 0, 0);
            }
            if (SPB > 1 && i_step + 1 >= nkg) continue;       // (the group's last step: stay on it)
            ++i_step;
            i_astep += 4096u;
            i_soff += 128u;
            if constexpr (TAPS) {
                i_c0 += 64;
                if (i_c0 >= p.Cin) {
                    i_c0 = 0;
                    if (++i_kw == TAPS) { i_kw = 0; ++i_kh; }
                    i_soff = (unsigned)((i_kh * p.W + i_kw) * p.in_cs) * 2u;
                }
            } else {
                i_c0 += 64;
            }
        }
        i_stage = i_stage + 1 == nslots ? 0 : i_stage + 1;
    };

    f32x16 acc[MC][MP];
#pragma unroll
    for (int mi = 0; mi < MC; ++mi)
#pragma unroll
        for (int pi = 0; pi < MP; ++pi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][pi][r] = 0.f;

    // fragment addresses: lane (row lane % 32, K half lane / 32) reads chunk 2 j + half of its row for MFMA step j
    const int frow = lane & 31, fh = lane >> 5, swz = (lane >> 1) & 7;
    unsigned koff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) koff[j] = (unsigned)(((2 * j + fh) ^ swz) << 4);
    const unsigned aoff = (unsigned)(wc * MC * 32 + frow) * 128u;
    const unsigned boff = (unsigned)BM * 128u + (unsigned)(wp * MP * 32 + frow) * 128u;

    const int psup = (per + SPB - 1) / SPB, nsup = (nkg + SPB - 1) / SPB;      // ring slots (SPB steps each) to walk / of this group
    const int npro = min(ns - 1, nsup);
    CONVD_STAMP(1)
    if (loader)
        for (int s = 0; s < npro; ++s) issue();
    CONVD_STAMP(2)
    int issued = npro, c_stage = 0;
    for (int it = 0; it < psup; ++it) {
        if (loader) {
            // slots requested and not yet consumed (this iteration's included); everything but the oldest may stay in flight
            const int rem = min(ns - 1, nsup - it);
            if (rem >= 3) wait_vmcnt<2 * SPB * PPS>();
            else if (rem == 2) wait_vmcnt<SPB * PPS>();
            else wait_vmcnt<0>();
        }
        CONVD_STAMP(8 + 4 * it)
        __builtin_amdgcn_s_barrier();       // every wave's pieces of this step have landed; the previous step's slot is free
        asm volatile("" ::: "memory");
        CONVD_STAMP(9 + 4 * it)
        if (loader && issued < nsup) {
            if (!CONVD_ABL(1)) issue();
            ++issued;
        }
        CONVD_STAMP(10 + 4 * it)
        if (mfma_wave && it < nsup && !CONVD_ABL(2)) {
#pragma unroll
            for (int u = 0; u < SPB; ++u) {
                if (SPB > 1 && it * SPB + u >= nkg) break;     // (odd K range: the last slot's second half is a repeat)
                const char* sa = ring + (c_stage * SPB + u) * SS;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f16x8 af[MC], bf[MP];
#pragma unroll
                    for (int mi = 0; mi < MC; ++mi) af[mi] = *reinterpret_cast<const f16x8*>(sa + aoff + mi * 4096 + koff[j]);
#pragma unroll
                    for (int pi = 0; pi < MP; ++pi) bf[pi] = *reinterpret_cast<const f16x8*>(sa + boff + pi * 4096 + koff[j]);
#pragma unroll
                    for (int mi = 0; mi < MC; ++mi)
#pragma unroll
                        for (int pi = 0; pi < MP; ++pi)
                            acc[mi][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[pi], acc[mi][pi], 0, 0, 0);
                }
            }
            c_stage = c_stage + 1 == nslots ? 0 : c_stage + 1;
        }
        CONVD_STAMP(11 + 4 * it)
    }
    CONVD_STAMP(3)

    // ---- epilogue
    // K groups: the partial tiles of groups 1.. meet group 0's in LDS (fragment layout, conflict-free float4 per lane)
    if constexpr (KG > 1) {
        __syncthreads();                             // every wave is done with the ring (no DMA is in flight any more)
        float4* part = reinterpret_cast<float4*>(smem + BN * LDO * 4);
        if (mfma_wave && g > 0) {
#pragma unroll
            for (int mi = 0; mi < MC; ++mi)
#pragma unroll
                for (int pi = 0; pi < MP; ++pi)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        part[(((g - 1) * MC * MP + mi * MP + pi) * 4 + q) * 256 + w * 64 + lane] =
                            make_float4(acc[mi][pi][4 * q + 0], acc[mi][pi][4 * q + 1], acc[mi][pi][4 * q + 2], acc[mi][pi][4 * q + 3]);
        }
        __syncthreads();
        if (mfma_wave && g == 0) {
            for (int gg = 1; gg < KG; ++gg)          // fixed order: deterministic sums
#pragma unroll
                for (int mi = 0; mi < MC; ++mi)
#pragma unroll
                    for (int pi = 0; pi < MP; ++pi)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 t4 = part[(((gg - 1) * MC * MP + mi * MP + pi) * 4 + q) * 256 + w * 64 + lane];
                            acc[mi][pi][4 * q + 0] += t4.x; acc[mi][pi][4 * q + 1] += t4.y;
                            acc[mi][pi][4 * q + 2] += t4.z; acc[mi][pi][4 * q + 3] += t4.w;
                        }
        }
    }
    CONVD_STAMP(4)
    // Through LDS ([0, BN * LDO * 4): the transposed fp32 tile): every lane then owns 8 consecutive couts of a pixel row and
    // outputs leave as 16-byte pieces of whole NHWC rows.  (Measured and rejected, profiles/r05_convd_kscan_*: stores straight
    // from the MFMA fragments -- v_permlane32_swap pairs, 16 bytes per lane to 32 different rows per instruction, no barrier
    // -- cost +0.8 us per launch on 128 x 128 tiles: the store path takes ~8 B/clk/CU either way, and the loader waves of a
    // ROLE workgroup help with the row-wise form.)
    float* so = reinterpret_cast<float*>(smem);
    if constexpr (KG == 1) __syncthreads();          // every wave is done with the ring
    if (mfma_wave && g == 0) {
        // D fragment: lane owns pixel lane % 32 and couts 8 q + 4 (lane / 32) + {0..3} of each 32 x 32 block
#pragma unroll
        for (int pi = 0; pi < MP; ++pi)
#pragma unroll
            for (int mi = 0; mi < MC; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(&so[((wp * MP + pi) * 32 + frow) * LDO + (wc * MC + mi) * 32 + 8 * q + 4 * fh]) =
                        make_float4(acc[mi][pi][4 * q + 0], acc[mi][pi][4 * q + 1], acc[mi][pi][4 * q + 2], acc[mi][pi][4 * q + 3]);
    }
    __syncthreads();
    constexpr int CH = BM / 8;                       // 16 B chunks per pixel row of the tile
    constexpr int ROWS = T / CH;
    const int och = tid % CH, orow = tid / CH;
    const int co = c0 + och * 8;
    if (co < p.cout_store && !CONVD_ABL(4)) {
        const int hw_out = p.Ho * p.Wo;
        const float inv_hw = 1.f / (float)hw_out, inv_wo = 1.f / (float)p.Wo;
        float bias8[8];
        *reinterpret_cast<float4*>(&bias8[0]) = *reinterpret_cast<const float4*>(p.bias + co);
        *reinterpret_cast<float4*>(&bias8[4]) = *reinterpret_cast<const float4*>(p.bias + co + 4);
        constexpr int NPASS = (BN + ROWS - 1) / ROWS;
        // all passes' tile reads (and shortcut loads) are requested before the first activation: with one or two waves per
        // SIMD a pass-by-pass loop paid the LDS / L2 latency once per pass (7.5 k cycles for a 128 x 128 tile)
        float v[NPASS][8], r[NPASS][8];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = min(orow + ps * ROWS, BN - 1);
            *reinterpret_cast<float4*>(&v[ps][0]) = *reinterpret_cast<const float4*>(&so[row * LDO + och * 8]);
            *reinterpret_cast<float4*>(&v[ps][4]) = *reinterpret_cast<const float4*>(&so[row * LDO + och * 8 + 4]);
        }
        if (p.res_mode != RES_NONE) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int pix = min(p0 + orow + ps * ROWS, p.P - 1);
                unpack8(*reinterpret_cast<const uint4*>(p.res + (size_t)pix * p.res_cs + p.res_coff + co), r[ps]);
            }
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = orow + ps * ROWS;
            const int pix = p0 + row;
            if (row >= BN || pix >= p.P) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[ps][e] += bias8[e];
                if (p.res_mode == RES_BEFORE_ACT) v[ps][e] += r[ps][e];
            }
            apply_act_n<8>(v[ps], p.act);
            if (p.res_mode == RES_AFTER_ACT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[ps][e] += r[ps][e];
            }
            if (p.out32) {
                float* dst = p.out32 + (size_t)pix * p.out_cs + p.out_coff + co;
                *reinterpret_cast<float4*>(dst) = make_float4(v[ps][0], v[ps][1], v[ps][2], v[ps][3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[ps][4], v[ps][5], v[ps][6], v[ps][7]);
            } else {
                const uint4 o = pack8(v[ps]);
                if (p.up == 2) {   // fused nearest x2 upsample: replicate to the 2x2 block
                    const int nn = idiv_small(pix, hw_out, inv_hw), rem = pix - nn * hw_out;
                    const int ry = idiv_small(rem, p.Wo, inv_wo), rx = rem - ry * p.Wo;
                    const size_t o00 = ((size_t)nn * 2 * p.Ho + 2 * ry) * (2 * p.Wo) + 2 * rx;
                    f16* dst = p.out + o00 * p.out_cs + p.out_coff + co;
                    *reinterpret_cast<uint4*>(dst) = o;
                    *reinterpret_cast<uint4*>(dst + p.out_cs) = o;
                    *reinterpret_cast<uint4*>(dst + (size_t)2 * p.Wo * p.out_cs) = o;
                    *reinterpret_cast<uint4*>(dst + (size_t)(2 * p.Wo + 1) * p.out_cs) = o;
                } else {
                    *reinterpret_cast<uint4*>(p.out + (size_t)pix * p.out_cs + p.out_coff + co) = o;
                }
            }
        }
    }
    CONVD_STAMP(5)
#ifdef FM_CONVD_TIMING
    if (stamp_slot >= 0) {
        long long* st = reinterpret_cast<long long*>(smem + stamp_off);
        st[263] = __builtin_amdgcn_s_memrealtime();
        st[6] = per;
        st[7] = BM * 1000000 + BN * 1000 + KG * 100 + SPB * 10 + ns;
        for (int i = 0; i < 264; ++i) g_convd_stamps[stamp_slot][i] = st[i];
    }
#endif
#endif
}

constexpr int LDS_MAX = 160 * 1024;

struct Cfg {
    int bm, bn, kg, ns, role, spb;
};

template <int WC, int WP, int MC, int MP, int KG, int TAPS, int ROLE, int SPB>
int launch_inst(const ConvParams& p, int ns, hipStream_t s) {
    constexpr int BM = WC * MC * 32, BN = WP * MP * 32, SS = (BM + BN) * 128 * SPB;      // one ring slot
    const int cout_pad = (p.Cout + 31) & ~31;
    const int nk = p.Kpad >> 6, per = ((nk + KG - 1) / KG + SPB - 1) / SPB;            // slots a K group walks
    const int nslots = ns < per ? ns : (per < 1 ? 1 : per);
    ConvParams q = p;
    q.grid_p = (p.P + BN - 1) / BN;
    q.grid_c = (cout_pad + BM - 1) / BM;
    q.grid_z = 1;
    q.weight_major = (size_t)cout_pad * p.Kpad > (size_t)p.N * p.H * p.W * p.Cin ? 1 : 0;
    const size_t ring = (size_t)KG * nslots * SS;
    const size_t epi = (size_t)BN * (BM + 4) * 4 + (size_t)(KG - 1) * BM * BN * 4;
    const size_t lds = ring > epi ? ring : epi;
    FM_CHECK_ARG(lds + CONVD_EXTRA_LDS <= (size_t)LDS_MAX);
    // the opt-in to more than 64 KB of dynamic LDS belongs to (device, instantiation): one flag each (a process may hold
    // contexts on several GPUs; set from any of their threads -- setting it twice is harmless, the flag only saves the call)
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    FM_HIP(hipGetDevice(&dev));
    if (dev >= 64 || !(configured.load(std::memory_order_relaxed) >> dev & 1)) {
        FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(convd_kernel<WC, WP, MC, MP, KG, TAPS, ROLE, SPB>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
        if (dev < 64) configured.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
    const int total = q.grid_p * q.grid_c;
    hipLaunchKernelGGL((convd_kernel<WC, WP, MC, MP, KG, TAPS, ROLE, SPB>), dim3(((total + 7) / 8) * 8), dim3(256 * KG * (1 + ROLE)),
                       lds + CONVD_EXTRA_LDS, s, q, ns, nslots CONVD_EXTRA_ARG);
    FM_HIP(hipGetLastError());
    return 0;
}

template <int MC, int MP, int KG, int ROLE, int SPB>
int launch_taps(const ConvParams& p, int ns, hipStream_t s) {
    if (p.KH == 1) return launch_inst<2, 2, MC, MP, KG, 0, ROLE, SPB>(p, ns, s);
    return launch_inst<2, 2, MC, MP, KG, 3, ROLE, SPB>(p, ns, s);
}

// instances: tiles 128 x 128, 128 x 64, 64 x 64; 1 or 2 K groups with or without loader waves; 4 K groups (16 MFMA waves)
// for the 64 x 64 tile only (two accumulators spill at the 128 registers a 1024-thread workgroup leaves a lane)
template <int MC, int MP, int SPB>
int launch_role(const ConvParams& p, const Cfg& c, hipStream_t s) {
    if constexpr (MC * MP == 1 && SPB == 1) {
        if (c.kg == 4) return launch_taps<MC, MP, 4, 0, 1>(p, c.ns, s);
    }
    if (c.kg >= 2) return c.role ? launch_taps<MC, MP, 2, 1, SPB>(p, c.ns, s) : launch_taps<MC, MP, 2, 0, SPB>(p, c.ns, s);
    return c.role ? launch_taps<MC, MP, 1, 1, SPB>(p, c.ns, s) : launch_taps<MC, MP, 1, 0, SPB>(p, c.ns, s);
}

// two steps per barrier only for the tiles with one or two accumulators per wave (the 2 x 2 tile has 16 MFMAs a step)
template <int MC, int MP>
int launch_kg(const ConvParams& p, const Cfg& c, hipStream_t s) {
    if constexpr (MC * MP < 4) {
        if (c.spb == 2) return launch_role<MC, MP, 2>(p, c, s);
    }
    return launch_role<MC, MP, 1>(p, c, s);
}

// ring slots: as many as the K range of a group (`per` slots of spb steps) and `lds_budget` allow (<= 4), at least 2
int stages_for(int bm, int bn, int kg, int spb, int per, int lds_budget) {
    const int ss = (bm + bn) * 128 * spb;
    int ns = lds_budget / (kg * ss);
    ns = ns > 4 ? 4 : ns;
    ns = ns > per ? per : ns;
    return ns < 2 ? 2 : ns;
}

// Tile / K-group / role choice, from the single-layer sweeps and K scans of round 5 (profiles/r05_convd_sweep_*.txt,
// r05_convd_kscan_*.txt; every shape of YOLOv4 @ 608, -CSP @ 640 and -P6 @ 1280 with cin % 64 == 0):
//   * the largest of 128 x 128, 128 x 64, 64 x 64 that still gives ~180 workgroups (0.7 per CU): the fill path and the
//     store path (~8 B/clk/CU) both want every CU, the LDS read port wants the large register tile;
//   * K ranges of one or two steps (1x1 convs on the large maps) are bound by their stores: 64 x 64 tiles, several small
//     workgroups per CU at different phases, no loader waves (they would idle);
//   * 128 x 128: one K group with loader waves; 128 x 64: the same with two K steps per barrier from 8 steps on (12.5 vs
//     13.2 us for two K groups on the 76 x 76 3x3 layers); 64 x 64: two K groups with loader waves from 8 steps on;
//   * ring depth 4 where one workgroup per CU is all the layer has, 2 (half the LDS) where a second workgroup can be
//     resident beside it (more than 256 tiles).
Cfg choose(const ConvParams& p) {
    const int cout_pad = (p.Cout + 31) & ~31, nk = p.Kpad >> 6;
    auto tiles = [&](int bm, int bn) { return (long)((p.P + bn - 1) / bn) * ((cout_pad + bm - 1) / bm); };
    Cfg c{64, 64, 1, 2, 0, 1};
    if (nk > 2) {
        if (cout_pad % 128 == 0 && tiles(128, 128) >= 180) c = Cfg{128, 128, 1, 2, 1, 1};
        else if (cout_pad % 128 == 0 && tiles(128, 64) >= 180) c = Cfg{128, 64, 1, 2, 1, nk >= 8 ? 2 : 1};
        else c.role = 1;
        if (c.bm == 64 && nk >= 8) c.kg = 2;
    }
    const long nt = tiles(c.bm, c.bn);
    c.ns = stages_for(c.bm, c.bn, c.kg, c.spb, ((nk + c.kg - 1) / c.kg + c.spb - 1) / c.spb, nt > 256 ? LDS_MAX / 2 : LDS_MAX);
    return c;
}

}  // namespace

void convd_set_cfg(int code) { g_convd_cfg = code; }

#ifdef FM_CONVD_TIMING
extern "C" int fm_debug_convd_stamps(long long* out528, int set_abl) {
    FM_HIP(hipDeviceSynchronize());
    FM_HIP(hipMemcpyFromSymbol(out528, HIP_SYMBOL(g_convd_stamps), sizeof(long long) * 528));
    g_convd_abl = set_abl;
    return 0;
}
#endif

// p.w: tile-image weights (header); p.K = KH * KW * Cin; 3x3: Cin % 64 == 0 (Kpad == K); 1x1: Cin % 8 == 0, the image is
// zero-padded to Kpad = ceil64(K)
int launch_convd(const ConvParams& p, hipStream_t s) {
    FM_CHECK_ARG(p.in_cs % 8 == 0 && p.in_coff % 8 == 0 && p.K == p.KH * p.KW * p.Cin && p.Kpad == ((p.K + 63) & ~63));
    FM_CHECK_ARG(p.KH == 1 ? p.Cin % 8 == 0 : p.Cin % 64 == 0);
    FM_CHECK_ARG(p.out_cs % 8 == 0 && p.out_coff % 8 == 0 && p.KH == p.KW && ((p.KH == 1 && p.pad == 0) || p.KH == 3));
    FM_CHECK_ARG(p.res_mode == RES_NONE || (p.res_cs % 8 == 0 && p.res_coff % 8 == 0));
    FM_CHECK_ARG(p.P < (1 << 22) && p.grid_p >= 0);
    // 32-bit byte offsets below 2^31 (the marker for "outside the image" is 2^31): input view incl. the (pad, pad) shift
    // and the last tap, weight image
    FM_CHECK_ARG(((long)p.N * p.H * p.W + (long)(p.KH + p.pad) * p.W + p.KW + p.pad) * p.in_cs * 2 < (1L << 31));
    FM_CHECK_ARG((long)((p.Cout + 31) & ~31) * p.Kpad * 2 < (1L << 31));
    Cfg c = choose(p);
    if (g_convd_cfg) {      // forced: bm | bn << 8 | kg << 16 | ns << 20 | role << 24 | (spb == 2) << 25 (ns 0: as deep as LDS allows)
        Cfg f{g_convd_cfg & 255, (g_convd_cfg >> 8) & 255, (g_convd_cfg >> 16) & 15, (g_convd_cfg >> 20) & 15, (g_convd_cfg >> 24) & 1,
              1 + ((g_convd_cfg >> 25) & 1)};
        const int nk = p.Kpad >> 6;
        if (f.kg > nk) f.kg = nk >= 2 ? 2 : 1;
        if (f.kg == 4 && (f.bm != 64 || f.bn != 64)) f.kg = 2;
        if (f.bm * f.bn >= 128 * 128) f.spb = 1;
        if (f.kg == 4) f.role = 0, f.spb = 1;
        if (f.ns == 1) f.ns = 2;          // (a ring of one slot would be read while its DMA is in flight)
        const int slot = (f.bm + f.bn) * 128 * f.spb;
        if (f.ns == 0) f.ns = stages_for(f.bm, f.bn, f.kg, f.spb, ((nk + f.kg - 1) / f.kg + f.spb - 1) / f.spb, LDS_MAX);
        while (f.ns > 2 && (size_t)f.kg * f.ns * slot > (size_t)LDS_MAX) --f.ns;
        if ((size_t)f.kg * 2 * slot <= (size_t)LDS_MAX) c = f;
    }
    if (c.bm == 128 && c.bn == 128) return launch_kg<2, 2>(p, c, s);
    if (c.bm == 128 && c.bn == 64) return launch_kg<2, 1>(p, c, s);
    if (c.bm == 64 && c.bn == 64) return launch_kg<1, 1>(p, c, s);
    fm_set_error("convd: no %d x %d tile", c.bm, c.bn);
    return FM_ERR_ARG;
}

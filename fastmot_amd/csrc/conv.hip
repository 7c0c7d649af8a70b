// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16).
//
// Replaces the TensorRT engines of the reference (YOLOv4: fastmot/models/yolo.py:106-151,
// OSNet: fastmot/models/reid.py:48-92; graph semantics scripts/yolo2onnx.py:558-705): every
// Conv + folded BatchNorm + activation (+ shortcut add, + concat by output-channel offset) is ONE
// launch of this kernel (plus a tiny reduce launch when the layer is split along K).
//
// GEMM view   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (kh, kw, cin)
//   A operand = weights  (M = output channels), pre-packed [cout_pad32][Kpad64] fp16
//   B operand = im2col of the NHWC fp16 input, gathered on the fly (16 B = 8 channels per load,
//               zero fill outside the image), never materialised in HBM
//   both are K-contiguous per lane, which is exactly the 32x32x16 f16 fragment layout
//   D fragment: lane owns pixel (lane&31) and 4 consecutive output channels per register quad
//               -> 8-byte NHWC stores, epilogue (bias, activation, residual) fused in registers.
//
// What shapes the kernel is that the detector runs at batch 1: a layer has 0.2-1.5 M outputs, i.e.
// only 48-700 tiles of 64x64 for 256 CUs, and K is long (up to 4608).  Measured with one K step
// of prefetch and K=32 steps the loop ran at ~1.2 k cycles per step (latency bound: global load ->
// LDS store -> barrier -> LDS read -> 2 MFMAs per wave).  Hence:
//   * K step 64 (4 MFMAs per wave between barriers), LDS rows padded to 144 B (conflict-free
//     ds_read_b128 fragment reads), LDS double buffered, one barrier per step;
//   * 4-stage software pipeline in registers with UNCONDITIONAL, clamped loads (a load inside an
//     exec-masked branch makes hipcc emit s_waitcnt vmcnt(0)); out-of-image taps are zeroed when
//     the stage is written to LDS, so the waits are counted (vmcnt(N > 0)) and 3 steps stay in flight;
//   * split-K over gridDim.z when a layer has fewer than ~2 workgroups per CU: fp32 partial tiles
//     go to a workspace, a second small kernel sums them in a fixed order (deterministic) and
//     applies the epilogue.  (Tried and rejected, round 1: letting the last-arriving workgroup of a
//     tile do the reduction in the same launch.  The partials cross XCDs, whose L2s are not coherent
//     with each other: agent-scope fences write back + invalidate a whole L2 per workgroup (network
//     3.6x slower), and fence-free sc1 write-through stores / sc1 loads made split layers 1.6x slower
//     than conv + reduce launch, which only pays the ~5 us launch boundary.)
//
// Roofline: per layer max(2*K*Cout*P / 2.5 PFLOP/s, (in + out + weights) * 2 B / 8 TB/s);
// YOLOv4 @608 = 128.4 GFLOP, 618 MB -> 0.089 ms/frame lower bound.
#include "net.h"
#include <cstdlib>

namespace {

constexpr int BK = 64;    // K elements per step
constexpr int LDK = 72;   // padded LDS row (halves): 144 B

// TAPMODE: how the K walk crosses (kh, kw) taps -- 0: 1x1 conv (never), 1: Cin >= BK (at most one tap per
// K step: branch-free selects), 2: generic (divergent loop; compiled as exec-masked loops that split the
// pipelined K loop into many basic blocks, which is why the common cases get their own instances)
template <int WC, int WP, int MC, int MP, int MINB, int TAPMODE>
__global__ __launch_bounds__(256, MINB) void conv_igemm_kernel(const ConvParams p, float* __restrict__ ws) {
    static_assert(WC * WP == 4, "4 waves per block");
    static_assert(WC * MC <= 4 && WP * MP <= 4, "at most 4 chunks per thread and operand");
    constexpr int BMC = WC * MC * 32;
    constexpr int BNP = WP * MP * 32;
    constexpr int A_IT = (BMC + 31) / 32;
    constexpr int B_IT = (BNP + 31) / 32;
    // operand staging (double buffered); the same bytes are reused by the epilogue as an fp32
    // [pixel][cout + 4] tile so that outputs leave in 16 B pieces of whole NHWC rows
    constexpr int LDO = BMC + 4;
    constexpr int SMEM_OPER = 2 * (BMC + BNP) * LDK * 2, SMEM_OUT = BNP * LDO * 4;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_OPER > SMEM_OUT ? SMEM_OPER : SMEM_OUT];
    f16 (*sA)[BMC * LDK] = reinterpret_cast<f16 (*)[BMC * LDK]>(smem);
    f16 (*sB)[BNP * LDK] = reinterpret_cast<f16 (*)[BNP * LDK]>(smem + 2 * BMC * LDK * 2);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wc = wv / WP, wp = wv % WP;
    // ---- XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its
    // own 4 MB L2; with the natural (x, y, z) order every XCD touches every weight row and a whole layer's
    // weights are fetched 8 times (measured: 1.5 GB of L2 fills per frame for 0.37 GB of operands).  The
    // launch is 1-D; XCD i gets the i-th CONTIGUOUS chunk of a tile order in which its operand slice is
    // private: (cout tile, K split) slowest for weight-heavy layers, pixel tile slowest for input-heavy
    // ones.  Only performance depends on the round-robin assumption, never the result.
    int tile_p, tile_c, split;
    {
        const int total = p.grid_p * p.grid_c * p.grid_z;
        const int chunk = (total + 7) >> 3;
        const int logical = p.weight_major == 2 ? (int)blockIdx.x   // natural order (A/B experiments only)
                                                : (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if (logical >= total) return;
        // (idiv_small: float-reciprocal division, exact below 2^22 -- the integer sequences made up a third of the
        // ~900 instructions a 1x1 layer's thread executes)
        const float inv_gp = 1.f / (float)p.grid_p, inv_gc = 1.f / (float)p.grid_c, inv_gz = 1.f / (float)p.grid_z;
        if (p.weight_major == 2) {
            const int r = idiv_small(logical, p.grid_p, inv_gp);
            tile_p = logical - r * p.grid_p;
            split = idiv_small(r, p.grid_c, inv_gc);
            tile_c = r - split * p.grid_c;
        } else if (p.weight_major) {          // logical = (tile_c * grid_z + split) * grid_p + tile_p
            const int r = idiv_small(logical, p.grid_p, inv_gp);
            tile_p = logical - r * p.grid_p;
            tile_c = idiv_small(r, p.grid_z, inv_gz);
            split = r - tile_c * p.grid_z;
        } else {                       // logical = (tile_p * grid_c + tile_c) * grid_z + split
            const int r = idiv_small(logical, p.grid_z, inv_gz);
            split = logical - r * p.grid_z;
            tile_p = idiv_small(r, p.grid_c, inv_gc);
            tile_c = r - tile_p * p.grid_c;
        }
    }
    const int nsplit = p.grid_z;
    const int c0 = tile_c * BMC, p0 = tile_p * BNP;
    const int lrow = tid >> 3, lchunk = tid & 7;
    const int cout_pad = (p.Cout + 31) & ~31;

    // K range of this split
    const int nk_total = p.Kpad / BK;
    const int per = (nk_total + nsplit - 1) / nsplit;
    const int k_begin = split * per;
    const int nk = min(per, nk_total - k_begin);     // >= 1 by construction of the launch

    // ---- per-thread im2col state of the pixels this thread stages
    const f16* pbase[B_IT];
    int phi0[B_IT], pwi0[B_IT];
    const int hw_out = p.Ho * p.Wo;
    const float inv_hw = 1.f / (float)hw_out, inv_wo = 1.f / (float)p.Wo;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int pix = min(p0 + lrow + 32 * i, p.P - 1);    // clamped pixels are never stored
        const int hw = hw_out;
        const int n = idiv_small(pix, hw, inv_hw), rem = pix - n * hw;
        const int ho = idiv_small(rem, p.Wo, inv_wo), wo = rem - ho * p.Wo;
        phi0[i] = ho * p.stride - p.pad;
        pwi0[i] = wo * p.stride - p.pad;
        pbase[i] = p.in + (size_t)n * p.H * p.W * p.in_cs + p.in_coff;
    }
    // K walk of this thread's 8-channel chunk: k = (k_begin + step) * 64 + lchunk*8 -> (kh, kw, c)
    int kk = k_begin * BK + lchunk * 8;
    int tap = idiv_small(kk, p.Cin, 1.f / (float)p.Cin);
    int kc = kk - tap * p.Cin;
    int kh = idiv_small(tap, p.KW, 1.f / (float)p.KW), kw = tap - kh * p.KW;

    // Stage registers.  Everything below uses macros with literal stage names: passing stage
    // arrays to lambdas by pointer/reference kept them in scratch memory (no SROA).
    // (plain scalars, no arrays: hipcc left two-element register arrays in scratch memory)
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    uint4 ra0_0 = z4, ra0_1 = z4, ra0_2 = z4, ra0_3 = z4, ra1_0 = z4, ra1_1 = z4, ra1_2 = z4, ra1_3 = z4;
    uint4 ra2_0 = z4, ra2_1 = z4, ra2_2 = z4, ra2_3 = z4, ra3_0 = z4, ra3_1 = z4, ra3_2 = z4, ra3_3 = z4;
    uint4 rb0_0 = z4, rb0_1 = z4, rb0_2 = z4, rb0_3 = z4, rb1_0 = z4, rb1_1 = z4, rb1_2 = z4, rb1_3 = z4;
    uint4 rb2_0 = z4, rb2_1 = z4, rb2_2 = z4, rb2_3 = z4, rb3_0 = z4, rb3_1 = z4, rb3_2 = z4, rb3_3 = z4;
    unsigned okm0 = 0, okm1 = 0, okm2 = 0, okm3 = 0;   // bit i: tap of rb?[i] lies inside the image
    const int arow_max = cout_pad - 1;
    const f16* wbase = p.w + (size_t)k_begin * BK + lchunk * 8;
    // (loops over the per-thread chunk index are unrolled by hand with `if constexpr`: a
    // `_Pragma("unroll") for` inside these macros was not unrolled and sent the arrays to scratch)
#define CONV_LOAD_A(S, I, KS)                                                                               \
    if constexpr ((I) < A_IT) {                                                                             \
        const int row = min(c0 + lrow + 32 * (I), arow_max);                                                \
        ra##S##_##I = *reinterpret_cast<const uint4*>(wbase + (__umul24(row, p.Kpad) + (unsigned)((KS) * BK)));  \
    }
#define CONV_LOAD_B(S, I)                                                                                   \
    if constexpr ((I) < B_IT) {                                                                             \
        const int hi = phi0[I] + kh, wi = pwi0[I] + kw;                                                     \
        const bool ok = kk < p.K && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;                             \
        m_ |= ok ? (1u << (I)) : 0u;                                                                        \
        const int hc = min(max(hi, 0), p.H - 1), wcl = min(max(wi, 0), p.W - 1);                            \
        const int kcc = kk < p.K ? kc : 0;                                                                  \
        rb##S##_##I = *reinterpret_cast<const uint4*>(                                                         \
            pbase[I] + (__umul24(__umul24(hc, p.W) + wcl, p.in_cs) + (unsigned)kcc));                       \
    }
#define CONV_LOAD_STAGE(S, KS)                                                                              \
    {                                                                                                       \
        CONV_LOAD_A(S, 0, KS) CONV_LOAD_A(S, 1, KS) CONV_LOAD_A(S, 2, KS) CONV_LOAD_A(S, 3, KS)             \
        unsigned m_ = 0;                                                                                    \
        CONV_LOAD_B(S, 0) CONV_LOAD_B(S, 1) CONV_LOAD_B(S, 2) CONV_LOAD_B(S, 3)                             \
        okm##S = m_;                                                                                        \
        kk += BK;                                                                                           \
        kc += BK;                                                                                           \
        if constexpr (TAPMODE == 1) {                                                                       \
            const bool wr_ = kc >= p.Cin;                                                                   \
            kc -= wr_ ? p.Cin : 0;                                                                          \
            kw += wr_ ? 1 : 0;                                                                              \
            const bool wr2_ = kw == p.KW;                                                                   \
            kw = wr2_ ? 0 : kw;                                                                             \
            kh += wr2_ ? 1 : 0;                                                                             \
        } else if constexpr (TAPMODE == 2) {                                                                \
            while (kc >= p.Cin) {                                                                           \
                kc -= p.Cin;                                                                                \
                if (++kw == p.KW) { kw = 0; ++kh; }                                                         \
            }                                                                                               \
        }                                                                                                   \
    }
#define CONV_STORE_A(S, I, BUF)                                                                             \
    if constexpr ((I) < A_IT) {                                                                             \
        if (lrow + 32 * (I) < BMC)                                                                          \
            *reinterpret_cast<uint4*>(&sA[BUF][(lrow + 32 * (I)) * LDK + lchunk * 8]) = ra##S##_##I;           \
    }
#define CONV_STORE_B(S, I, BUF)                                                                             \
    if constexpr ((I) < B_IT) {                                                                             \
        if (lrow + 32 * (I) < BNP) {                                                                        \
            uint4 v = rb##S##_##I;                                                                             \
            const bool ok = (okm##S >> (I)) & 1u;                                                           \
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;             \
            *reinterpret_cast<uint4*>(&sB[BUF][(lrow + 32 * (I)) * LDK + lchunk * 8]) = v;                  \
        }                                                                                                   \
    }
#define CONV_STORE_STAGE(S, BUF)                                                                            \
    {                                                                                                       \
        CONV_STORE_A(S, 0, BUF) CONV_STORE_A(S, 1, BUF) CONV_STORE_A(S, 2, BUF) CONV_STORE_A(S, 3, BUF)     \
        CONV_STORE_B(S, 0, BUF) CONV_STORE_B(S, 1, BUF) CONV_STORE_B(S, 2, BUF) CONV_STORE_B(S, 3, BUF)     \
    }
#define CONV_COMPUTE(BUF)                                                                                   \
    {                                                                                                       \
        _Pragma("unroll") for (int k16 = 0; k16 < BK / 16; ++k16) {                                         \
            f16x8 af[MC], bf[MP];                                                                           \
            _Pragma("unroll") for (int mi = 0; mi < MC; ++mi)                                               \
                af[mi] = *reinterpret_cast<const f16x8*>(&sA[BUF][((wc * MC + mi) * 32 + frow) * LDK + k16 * 16 + fk]); \
            _Pragma("unroll") for (int pi = 0; pi < MP; ++pi)                                               \
                bf[pi] = *reinterpret_cast<const f16x8*>(&sB[BUF][((wp * MP + pi) * 32 + frow) * LDK + k16 * 16 + fk]); \
            _Pragma("unroll") for (int mi = 0; mi < MC; ++mi)                                               \
                _Pragma("unroll") for (int pi = 0; pi < MP; ++pi)                                           \
                    acc[mi][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[pi], acc[mi][pi], 0, 0, 0); \
        }                                                                                                   \
    }

    f32x16 acc[MC][MP];
#pragma unroll
    for (int mi = 0; mi < MC; ++mi)
#pragma unroll
        for (int pi = 0; pi < MP; ++pi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][pi][r] = 0.f;

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    CONV_LOAD_STAGE(0, 0)
    if (1 < nk) CONV_LOAD_STAGE(1, 1)
    if (2 < nk) CONV_LOAD_STAGE(2, 2)
    if (3 < nk) CONV_LOAD_STAGE(3, 3)
    CONV_STORE_STAGE(0, 0)
    __syncthreads();
    int ks0 = 0;
    // steady state: each step refills the stage it has just retired, 4 steps ahead; no guards
    for (; ks0 + 8 <= nk; ks0 += 4) {
        CONV_LOAD_STAGE(0, ks0 + 4) CONV_COMPUTE(0) CONV_STORE_STAGE(1, 1) __syncthreads();
        CONV_LOAD_STAGE(1, ks0 + 5) CONV_COMPUTE(1) CONV_STORE_STAGE(2, 0) __syncthreads();
        CONV_LOAD_STAGE(2, ks0 + 6) CONV_COMPUTE(0) CONV_STORE_STAGE(3, 1) __syncthreads();
        CONV_LOAD_STAGE(3, ks0 + 7) CONV_COMPUTE(1) CONV_STORE_STAGE(0, 0) __syncthreads();
    }
    // drain: the last (up to 7) steps, guarded
    for (; ks0 < nk; ks0 += 4) {
        if (ks0 + 4 < nk) CONV_LOAD_STAGE(0, ks0 + 4)
        CONV_COMPUTE(0)
        if (ks0 + 1 < nk) CONV_STORE_STAGE(1, 1)
        __syncthreads();
        if (ks0 + 1 >= nk) break;
        if (ks0 + 5 < nk) CONV_LOAD_STAGE(1, ks0 + 5)
        CONV_COMPUTE(1)
        if (ks0 + 2 < nk) CONV_STORE_STAGE(2, 0)
        __syncthreads();
        if (ks0 + 2 >= nk) break;
        if (ks0 + 6 < nk) CONV_LOAD_STAGE(2, ks0 + 6)
        CONV_COMPUTE(0)
        if (ks0 + 3 < nk) CONV_STORE_STAGE(3, 1)
        __syncthreads();
        if (ks0 + 3 >= nk) break;
        if (ks0 + 7 < nk) CONV_LOAD_STAGE(3, ks0 + 7)
        CONV_COMPUTE(1)
        if (ks0 + 4 < nk) CONV_STORE_STAGE(0, 0)
        __syncthreads();
    }
#undef CONV_LOAD_STAGE
#undef CONV_LOAD_A
#undef CONV_LOAD_B
#undef CONV_STORE_A
#undef CONV_STORE_B
#undef CONV_STORE_STAGE
#undef CONV_COMPUTE

    // ---- epilogue
    if (nsplit > 1) {   // split-K: raw fp32 partial sums, reduced by splitk_reduce_kernel
#pragma unroll
        for (int pi = 0; pi < MP; ++pi) {
            const int pix = p0 + (wp * MP + pi) * 32 + (lane & 31);
            if (pix >= p.P) continue;
#pragma unroll
            for (int mi = 0; mi < MC; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = c0 + (wc * MC + mi) * 32 + 8 * g + 4 * (lane >> 5);
                    if (co >= p.cout_store) continue;
                    *reinterpret_cast<float4*>(ws + ((size_t)split * p.P + pix) * cout_pad + co) =
                        make_float4(acc[mi][pi][4 * g + 0], acc[mi][pi][4 * g + 1], acc[mi][pi][4 * g + 2],
                                    acc[mi][pi][4 * g + 3]);
                }
        }
        return;
    }
    // The MFMA D fragment gives a lane 4 channels of one pixel (8 B, 32 different NHWC rows per
    // store instruction).  Transposing through LDS makes every lane own 8 consecutive channels and
    // consecutive lanes consecutive 16 B of the same row: full-line stores and residual loads.
    float* so = reinterpret_cast<float*>(smem);
    __syncthreads();                                 // every wave is done with the operand tiles
#pragma unroll
    for (int pi = 0; pi < MP; ++pi)
#pragma unroll
        for (int mi = 0; mi < MC; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&so[((wp * MP + pi) * 32 + (lane & 31)) * LDO +
                                               (wc * MC + mi) * 32 + 8 * g + 4 * (lane >> 5)]) =
                    make_float4(acc[mi][pi][4 * g + 0], acc[mi][pi][4 * g + 1], acc[mi][pi][4 * g + 2],
                                acc[mi][pi][4 * g + 3]);
    __syncthreads();
    constexpr int CH = BMC / 8;                      // 16 B chunks per pixel row of the tile
    constexpr int ROWS = 256 / CH;
    const int och = tid % CH, orow = tid / CH;
    const int co = c0 + och * 8;
    if (co < p.cout_store) {
        float bias8[8];
        *reinterpret_cast<float4*>(&bias8[0]) = *reinterpret_cast<const float4*>(p.bias + co);
        *reinterpret_cast<float4*>(&bias8[4]) = *reinterpret_cast<const float4*>(p.bias + co + 4);
#pragma unroll
        for (int it = 0; it < BNP / ROWS; ++it) {
            const int row = it * ROWS + orow;
            const int pix = p0 + row;
            if (pix >= p.P) break;
            float v[8];
            *reinterpret_cast<float4*>(&v[0]) = *reinterpret_cast<const float4*>(&so[row * LDO + och * 8]);
            *reinterpret_cast<float4*>(&v[4]) = *reinterpret_cast<const float4*>(&so[row * LDO + och * 8 + 4]);
            float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.res_mode != RES_NONE)
                unpack8(*reinterpret_cast<const uint4*>(p.res + (size_t)pix * p.res_cs + p.res_coff + co), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += bias8[e];
                if (p.res_mode == RES_BEFORE_ACT) v[e] += r[e];
            }
            apply_act_n<8>(v, p.act);
            if (p.res_mode == RES_AFTER_ACT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (p.out32) {
                float* dst = p.out32 + (size_t)pix * p.out_cs + p.out_coff + co;
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                const uint4 o = pack8(v);
                if (p.up == 2) {   // fused nearest x2 upsample: replicate to the 2x2 block
                    const int hw = hw_out, nn = idiv_small(pix, hw, inv_hw), rem = pix - nn * hw;
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
}

// sum of the split-K partials (fixed order z = 0..S-1) + bias + activation (+ residual)
__global__ void splitk_reduce_kernel(const ConvParams p, const float* __restrict__ ws, int S) {
    const int cout_pad = (p.Cout + 31) & ~31;
    const int c4n = p.cout_store / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)p.P * c4n) return;
    const int co = (int)(idx % c4n) * 4;
    const long pix = idx / c4n;
    float4 a = *reinterpret_cast<const float4*>(ws + (size_t)pix * cout_pad + co);
    for (int z = 1; z < S; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(ws + ((size_t)z * p.P + pix) * cout_pad + co);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
    float v[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.res_mode != RES_NONE) {
        const f16x4 rv = *reinterpret_cast<const f16x4*>(p.res + (size_t)pix * p.res_cs + p.res_coff + co);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (float)rv[e];
    }
    if (p.res_mode == RES_BEFORE_ACT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    apply_act_n<4>(v, p.act);
    if (p.res_mode == RES_AFTER_ACT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    if (p.out32) {
        *reinterpret_cast<float4*>(p.out32 + (size_t)pix * p.out_cs + p.out_coff + co) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
        store_out(p, pix, co, o);
    }
}

template <int WC, int WP, int MC, int MP, int MINB, int TAPMODE>
int launch_cfg_tap(const ConvParams& p, int S, float* ws, hipStream_t s) {
    constexpr int BMC = WC * MC * 32, BNP = WP * MP * 32;
    const int cout_pad = (p.Cout + 31) & ~31;
    ConvParams q = p;
    q.grid_p = (p.P + BNP - 1) / BNP;
    q.grid_c = (cout_pad + BMC - 1) / BMC;
    q.grid_z = S;
    // operand bytes: weights vs. (im2col-free) input; the heavier one gets the XCD-private slice
    q.weight_major = (size_t)cout_pad * p.Kpad > (size_t)p.N * p.H * p.W * p.Cin ? 1 : 0;
    constexpr int force = -1;
    if (force >= 0) q.weight_major = force;
    const int total = q.grid_p * q.grid_c * q.grid_z;
    dim3 grid(((total + 7) / 8) * 8);
    hipLaunchKernelGGL((conv_igemm_kernel<WC, WP, MC, MP, MINB, TAPMODE>), grid, dim3(256), 0, s, q, ws);
    if (S > 1) {
        const long total = (long)p.P * (p.cout_store / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, ws, S);
    }
    FM_HIP(hipGetLastError());
    return 0;
}

template <int WC, int WP, int MC, int MP, int MINB>
int launch_cfg(const ConvParams& p, int S, float* ws, hipStream_t s) {
    if (p.KH * p.KW == 1) return launch_cfg_tap<WC, WP, MC, MP, MINB, 0>(p, S, ws, s);
    if (p.Cin >= BK) return launch_cfg_tap<WC, WP, MC, MP, MINB, 1>(p, S, ws, s);
    return launch_cfg_tap<WC, WP, MC, MP, MINB, 2>(p, S, ws, s);
}

}  // namespace

// Tile + split selection.  Tiles: 32c x 128p for narrow layers (OSNet x0.25), otherwise 64c x 64p.
// (A 64c x 128p tile for the large early layers was measured 1.5-1.7x SLOWER on them -- 55 KB of LDS
// and 256 VGPRs leave 2 workgroups per CU to hide the load -> LDS -> MFMA -> store chain of a layer
// with only 1-5 K-steps; the 64 x 64 tile runs 4 per CU.)  Split-K brings the launch to ~2
// workgroups per CU as long as every split keeps >= 4 K-steps.
int launch_conv(const ConvParams& p, float* ws, size_t ws_floats, hipStream_t s) {
    FM_CHECK_ARG(p.Cin % 8 == 0 && p.in_cs % 8 == 0 && p.in_coff % 8 == 0);
    FM_CHECK_ARG(p.out_cs % 8 == 0 && p.out_coff % 8 == 0 && p.Kpad % BK == 0);
    FM_CHECK_ARG(p.res_mode == RES_NONE || (p.res_cs % 8 == 0 && p.res_coff % 8 == 0));
    // 24-bit multiplies (full-rate VALU) build the operand offsets: element offsets inside one sample / the
    // weight matrix must fit 32 bits and their factors 24 bits
    FM_CHECK_ARG((long)p.H * p.W < (1L << 24) && p.in_cs < (1 << 24) && (long)p.H * p.W * p.in_cs < (1L << 32));
    FM_CHECK_ARG(p.Kpad < (1 << 24) && (long)((p.Cout + 31) & ~31) * p.Kpad < (1L << 32));
    FM_CHECK_ARG(p.P < (1 << 22) && p.Kpad < (1 << 22));      // idiv_small (prologue index decompositions)
    const int cout_pad = (p.Cout + 31) & ~31;
    auto tiles = [&](int bmc, int bnp) { return (long)((p.P + bnp - 1) / bnp) * ((cout_pad + bmc - 1) / bmc); };
    const int nk = p.Kpad / BK;
    // a split costs a second (reduce) launch, ~5 us: only worth it when it removes >= ~12 K-steps
    constexpr int split_min_nk = 16;
    constexpr int split_target = 384;
    constexpr int split_min_steps = 4;
    auto split_for = [&](long t) {
        int S = 1;
        if (t < 256 && nk >= split_min_nk) {
            S = (int)((split_target + t - 1) / t);
            S = S < nk / split_min_steps ? S : nk / split_min_steps;
            S = S > 16 ? 16 : S;
            if (S < 1) S = 1;
            // every split must own at least one step
            while (S > 1 && (long)((nk + S - 1) / S) * (S - 1) >= nk) --S;
            while (S > 1 && (size_t)S * p.P * cout_pad > ws_floats) --S;
        }
        return S;
    };
    constexpr int minb = 2;
    if (cout_pad <= 32) return launch_cfg<1, 4, 1, 1, 2>(p, 1, ws, s);                   //  32c x 128p
    const long t = tiles(64, 64);
    // many short workgroups (early, memory-bound layers): 128 VGPRs -> 4 workgroups per CU
    if (minb == 4 || (minb == 3 && t >= 1024)) return launch_cfg<2, 2, 1, 1, 4>(p, split_for(t), ws, s);
    return launch_cfg<2, 2, 1, 1, 2>(p, split_for(t), ws, s);                            //  64c x  64p
}

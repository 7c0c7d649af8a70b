// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16).
//
// Replaces the TensorRT engines of the reference (YOLOv4: fastmot/models/yolo.py:106-151,
// OSNet: fastmot/models/reid.py:48-92; graph semantics scripts/yolo2onnx.py:558-705): every
// Conv + folded BatchNorm + activation (+ shortcut add, + concat by output-channel offset) is ONE
// launch of this kernel.
//
// GEMM view   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (kh, kw, cin)
//   A operand = weights  (M = output channels), pre-packed [cout_pad32][Kpad32] fp16
//   B operand = im2col of the NHWC fp16 input, gathered on the fly (16 B = 8 channels per load,
//               zero fill outside the image), never materialised in HBM
//   both are K-contiguous per lane, which is exactly the 32x32x16 f16 fragment layout
//   D fragment: lane owns pixel (lane&31) and 4 consecutive output channels per register quad
//               -> 8-byte NHWC stores, epilogue (bias, activation, residual) fused in registers.
// Tiling: 256 threads = 4 waves; block tile (WC*MC*32 channels) x (WP*MP*32 pixels) x 32 (K);
// LDS rows padded to 80 B (conflict-free ds_read_b128 for the fragment reads, see
// cdna_hip_programming.md section 2), double buffered, global loads for step k+1 in flight
// during the MFMAs of step k, one barrier per K step.
//
// Roofline: per layer max(2*K*Cout*P / 2.5 PFLOP/s, (in + out + weights) * 2 B / 8 TB/s);
// YOLOv4 @608 = 128.4 GFLOP, 618 MB -> 0.089 ms/frame lower bound.
#include "net.h"

namespace {

constexpr int BK = 32;    // K elements per step
constexpr int LDK = 40;   // padded LDS row (halves): 80 B

template <int WC, int WP, int MC, int MP>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    static_assert(WC * WP == 4, "4 waves per block");
    constexpr int BMC = WC * MC * 32;
    constexpr int BNP = WP * MP * 32;
    constexpr int A_IT = (BMC + 63) / 64;
    constexpr int B_IT = (BNP + 63) / 64;
    __shared__ __attribute__((aligned(16))) f16 sA[2][BMC * LDK];
    __shared__ __attribute__((aligned(16))) f16 sB[2][BNP * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wc = wv / WP, wp = wv % WP;
    const int c0 = blockIdx.y * BMC, p0 = blockIdx.x * BNP;
    const int lrow = tid >> 2, lchunk = tid & 3;
    const int cout_pad = (p.Cout + 31) & ~31;

    // ---- per-thread im2col state of the pixels this thread stages
    const f16* pbase[B_IT];
    int phi0[B_IT], pwi0[B_IT];
    bool pvalid[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = lrow + 64 * i;
        const int pix = p0 + row;
        pvalid[i] = (row < BNP) && (pix < p.P);
        const int pp = min(pix, p.P - 1);                        // clamped pixels are never stored
        const int hw = p.Ho * p.Wo;
        const int n = pp / hw, rem = pp - n * hw;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        phi0[i] = ho * p.stride - p.pad;
        pwi0[i] = wo * p.stride - p.pad;
        pbase[i] = p.in + (size_t)n * p.H * p.W * p.in_cs + p.in_coff;
    }
    // K walk of this thread's 8-channel chunk: k = kbase + lchunk*8 -> (kh, kw, c)
    int kk = lchunk * 8;
    int tap = kk / p.Cin;
    int kc = kk - tap * p.Cin;
    int kh = tap / p.KW, kw = tap - kh * p.KW;

    // ---- software pipeline: DEPTH K-steps of global loads in flight (registers), LDS double buffered.
    // Batch-1 layers launch few workgroups (often < 1 per CU), so nothing else hides the ~1-2k cycle
    // L2/HBM latency of a K-step: with one step of prefetch the loop ran at ~1.2k cycles per step.
    constexpr int DEPTH = 4;   // even: the LDS buffer of a step is then a compile-time constant
    // Stage registers.  Everything below is written with macros and literal stage indices: passing
    // the stage arrays to lambdas by pointer/reference kept them in scratch memory (no SROA).
    uint4 ra0[A_IT], ra1[A_IT], ra2[A_IT], ra3[A_IT], rb0[B_IT], rb1[B_IT], rb2[B_IT], rb3[B_IT];
    unsigned okm0 = 0, okm1 = 0, okm2 = 0, okm3 = 0;    // bit i: tap of rb[.][i] is inside the image.  The zeroing select is
                            // deferred to the LDS store; right after the load it would force vmcnt(0).
    const int arow_max = cout_pad - 1;
    // Loads are UNCONDITIONAL (addresses clamped into the tensor): a load inside an exec-masked
    // branch makes hipcc fall back to s_waitcnt vmcnt(0) around it, which serialises the pipeline.
#define CONV_LOAD_STAGE(S, KS)                                                                              \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                  \
            const int row = min(c0 + lrow + 64 * i, arow_max);                                              \
            ra##S[i] = *reinterpret_cast<const uint4*>(p.w + (size_t)row * p.Kpad + (KS) * BK + lchunk * 8); \
        }                                                                                                   \
        unsigned m_ = 0;                                                                                    \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                                  \
            const int hi = phi0[i] + kh, wi = pwi0[i] + kw;                                                 \
            const bool ok = kk < p.K && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;                         \
            m_ |= ok ? (1u << i) : 0u;                                                                      \
            const int hc = min(max(hi, 0), p.H - 1), wcl = min(max(wi, 0), p.W - 1);                        \
            const int kcc = kk < p.K ? kc : 0;                                                              \
            rb##S[i] = *reinterpret_cast<const uint4*>(pbase[i] + ((size_t)hc * p.W + wcl) * p.in_cs + kcc); \
        }                                                                                                   \
        okm##S = m_;                                                                                        \
        kk += BK;                                                                                           \
        kc += BK;                                                                                           \
        while (kc >= p.Cin) {                                                                               \
            kc -= p.Cin;                                                                                    \
            if (++kw == p.KW) { kw = 0; ++kh; }                                                             \
        }                                                                                                   \
    }
#define CONV_STORE_STAGE(S, BUF)                                                                            \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                                    \
            if (lrow + 64 * i < BMC)                                                                        \
                *reinterpret_cast<uint4*>(&sA[BUF][(lrow + 64 * i) * LDK + lchunk * 8]) = ra##S[i];         \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                                    \
            if (lrow + 64 * i < BNP) {                                                                      \
                uint4 v = rb##S[i];                                                                         \
                const bool ok = (okm##S >> i) & 1u;                                                         \
                v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;         \
                *reinterpret_cast<uint4*>(&sB[BUF][(lrow + 64 * i) * LDK + lchunk * 8]) = v;                \
            }                                                                                               \
    }
#define CONV_COMPUTE(BUF)                                                                                   \
    {                                                                                                       \
        _Pragma("unroll") for (int k16 = 0; k16 < 2; ++k16) {                                               \
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

    const int nk = p.Kpad / BK;
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    CONV_LOAD_STAGE(0, 0)
    if (1 < nk) CONV_LOAD_STAGE(1, 1)
    if (2 < nk) CONV_LOAD_STAGE(2, 2)
    if (3 < nk) CONV_LOAD_STAGE(3, 3)
    CONV_STORE_STAGE(0, 0)
    __syncthreads();
    int ks0 = 0;
    // steady state: each step refills the stage it has just retired, DEPTH steps ahead; no guards
    for (; ks0 + 2 * DEPTH <= nk; ks0 += DEPTH) {
        CONV_LOAD_STAGE(0, ks0 + 4) CONV_COMPUTE(0) CONV_STORE_STAGE(1, 1) __syncthreads();
        CONV_LOAD_STAGE(1, ks0 + 5) CONV_COMPUTE(1) CONV_STORE_STAGE(2, 0) __syncthreads();
        CONV_LOAD_STAGE(2, ks0 + 6) CONV_COMPUTE(0) CONV_STORE_STAGE(3, 1) __syncthreads();
        CONV_LOAD_STAGE(3, ks0 + 7) CONV_COMPUTE(1) CONV_STORE_STAGE(0, 0) __syncthreads();
    }
    // drain: the last (up to 2*DEPTH - 1) steps, guarded
    for (; ks0 < nk; ks0 += DEPTH) {
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
#undef CONV_STORE_STAGE
#undef CONV_COMPUTE

    // ---- epilogue: bias + activation (+ residual), 4 consecutive channels per 8-byte store
#pragma unroll
    for (int pi = 0; pi < MP; ++pi) {
        const int pix = p0 + (wp * MP + pi) * 32 + (lane & 31);
        if (pix >= p.P) continue;
#pragma unroll
        for (int mi = 0; mi < MC; ++mi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = c0 + (wc * MC + mi) * 32 + 8 * g + 4 * (lane >> 5);
                if (co >= p.cout_store) continue;
                const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
                float v[4] = {acc[mi][pi][4 * g + 0] + b.x, acc[mi][pi][4 * g + 1] + b.y,
                              acc[mi][pi][4 * g + 2] + b.z, acc[mi][pi][4 * g + 3] + b.w};
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.res_mode != RES_NONE) {
                    const f16x4 rv = *reinterpret_cast<const f16x4*>(p.res + (size_t)pix * p.res_cs + p.res_coff + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = (float)rv[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.res_mode == RES_BEFORE_ACT) v[e] += r[e];
                    v[e] = apply_act(v[e], p.act);
                    if (p.res_mode == RES_AFTER_ACT) v[e] += r[e];
                }
                if (p.out32) {
                    *reinterpret_cast<float4*>(p.out32 + (size_t)pix * p.out_cs + p.out_coff + co) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    f16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
                    *reinterpret_cast<f16x4*>(p.out + (size_t)pix * p.out_cs + p.out_coff + co) = o;
                }
            }
        }
    }
}

template <int WC, int WP, int MC, int MP>
int launch_cfg(const ConvParams& p, hipStream_t s) {
    constexpr int BMC = WC * MC * 32, BNP = WP * MP * 32;
    const int cout_pad = (p.Cout + 31) & ~31;
    dim3 grid((p.P + BNP - 1) / BNP, (cout_pad + BMC - 1) / BMC);
    hipLaunchKernelGGL((conv_igemm_kernel<WC, WP, MC, MP>), grid, dim3(256), 0, s, p);
    FM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Tile selection: the largest tile that still fills the 256 CUs; small-channel layers use the
// 32- or 64-channel tiles (OSNet x0.25 mid = 16..32, YOLO stem 32/64).
int launch_conv(const ConvParams& p, hipStream_t s) {
    FM_CHECK_ARG(p.Cin % 8 == 0 && p.in_cs % 8 == 0 && p.in_coff % 8 == 0);
    FM_CHECK_ARG(p.out_cs % 4 == 0 && p.out_coff % 4 == 0 && p.Kpad % BK == 0);
    const int cout_pad = (p.Cout + 31) & ~31;
    auto nwg = [&](int bmc, int bnp) {
        return (long)((p.P + bnp - 1) / bnp) * ((cout_pad + bmc - 1) / bmc);
    };
    if (cout_pad <= 32) return launch_cfg<1, 4, 1, 1>(p, s);                  //  32c x 128p
    if (cout_pad <= 64) {
        if (nwg(64, 128) >= 256) return launch_cfg<2, 2, 1, 2>(p, s);         //  64c x 128p
        return launch_cfg<2, 2, 1, 1>(p, s);                                  //  64c x  64p
    }
    if (nwg(128, 128) >= 384) return launch_cfg<2, 2, 2, 2>(p, s);            // 128c x 128p
    if (nwg(128, 64) >= 256) return launch_cfg<2, 2, 2, 1>(p, s);             // 128c x  64p
    return launch_cfg<2, 2, 1, 1>(p, s);                                      //  64c x  64p
}

// Conv-engine internals (NHWC fp16 activations, fp32 accumulate) shared by conv.hip/ops.hip/net.hip.
#pragma once
#include "common.h"
#include <hip/hip_fp16.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_LINEAR = 0, ACT_LEAKY = 1, ACT_MISH = 2, ACT_RELU = 3, ACT_LOGISTIC = 4, ACT_SWISH = 5 };
enum { RES_NONE = 0, RES_AFTER_ACT = 1, RES_BEFORE_ACT = 2 };

struct ConvParams {
    const f16* in;  int in_cs, in_coff;     // pixel stride (elements) / channel offset of the input view
    const f16* w;                           // [cout_pad32][Kpad] fp16, K = (kh, kw, cin)
    const float* bias;                      // [cout_pad32] (BN folded)
    f16* out;       int out_cs, out_coff;   // fp16 output view (may be null when out32 is set)
    float* out32;                           // optional fp32 output (YOLO heads), same view geometry
    const f16* res; int res_cs, res_coff;   // residual view (nullable)
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int K, Kpad, P;                         // K = KH*KW*Cin, Kpad = ceil64(K), P = N*Ho*Wo
    int cout_store;                         // ceil8(Cout): channels written
    int act, res_mode;
    int grid_p, grid_c, grid_z, weight_major;   // tile grid + XCD-aware order (filled by launch_conv)
    int up;                                 // 2: every output pixel is stored to its 2x2 block of a (2Ho, 2Wo) view
};

int launch_conv(const ConvParams& p, float* ws, size_t ws_floats, hipStream_t s);

struct NetState {
    int which = 0, max_batch = 0;
    std::vector<fm_tensor> tensors;
    std::vector<void*> bufs;
    std::vector<fm_layer> layers;
    char* weights = nullptr;
    size_t weight_bytes = 0;
    float* gates = nullptr;
    int n_gates = 0, gate_c = 0;
    hipStream_t stream = nullptr;
    char* arena = nullptr;    // shared activation arena (tensors with offset >= 0)
    float* ws = nullptr;      // split-K partial sums (fp32)
    size_t ws_floats = 0;
    std::vector<std::pair<long, hipGraphExec_t>> graphs;   // (batch, emb_offset) -> captured layer sequence
    bool use_graphs = true;
    int emb_offset = 0;   // row offset of FM_OP_HEAD outputs in ctx->emb (batched extraction)
    int batch_offset = 0; // sample offset into the input tensor for chunked runs
    int first = 0;        // first layer fm_net_run executes: 1 when the caller has run layer 0 itself (fm_net_run_stem_from)
};

NetState* fm_net_get(fm_ctx* ctx, int which);
int fm_net_run_internal(fm_ctx* ctx, int which, int batch);

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const f16x8 h = *reinterpret_cast<const f16x8*>(&v);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (f16)f[e];
    return *reinterpret_cast<const uint4*>(&h);
}

// floor(i / d) for 0 <= i, i < 2^22 and d >= 1 through the float reciprocal inv = 1.f / d (one multiply + convert instead of
// the ~35-instruction integer division sequence; exact in that range: the quotient's error q * 2^-23 stays below the
// 0.5 / d margin of (i + 0.5) / d).  Index decompositions in kernel prologues run once per lane, but a 5 us kernel has
// only a few hundred instructions in total.
__device__ __forceinline__ int idiv_small(int i, int d, float inv) {
    (void)d;
    return (int)(((float)i + 0.5f) * inv);
}

// acc + (float)h * k, h = the low (HI = 0) or high half of a packed fp16 pair: v_fma_mix_f32 extends the half exactly
// inside the FMA -- the same value as convert + fmaf, one VALU instruction instead of two (hipcc picks it or not
// depending on the surrounding code; the depthwise loops of the LightConv kernels are 72 of these per 8 outputs).
template <int HI>
__device__ __forceinline__ float fma_mix_h(uint32_t packed, float k, float acc) {
    float d;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(k), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(k), "v"(acc));
    return d;
}
// the same with the factor k a packed fp16 pair too (half of the registers of a float copy of the weights; a half
// extends to the same float either way, so the FMA's result is the same bit pattern)
template <int HI>
__device__ __forceinline__ float fma_mix_hh(uint32_t packed, uint32_t kpacked, float acc) {
    float d;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(packed), "v"(kpacked), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(packed), "v"(kpacked), "v"(acc));
    return d;
}
template <int N>
__device__ __forceinline__ void fma_mix_nh(const uint32_t* raw, const uint32_t* k, float* acc) {
#pragma unroll
    for (int e = 0; e < N; e += 2) {
        acc[e] = fma_mix_hh<0>(raw[e >> 1], k[e >> 1], acc[e]);
        acc[e + 1] = fma_mix_hh<1>(raw[e >> 1], k[e >> 1], acc[e + 1]);
    }
}
// acc[e] += (float)h[e] * k[e] for the N halfs packed in `raw` (N / 2 dwords)
template <int N>
__device__ __forceinline__ void fma_mix_n(const uint32_t* raw, const float* k, float* acc) {
    // (convert + fmaf, which hipcc pairs into v_pk_fma_f32, measured slower here: 8000 vs 6240 cycles for the first
    // depthwise phase of a 16 x 8 chain, scripts/lch_timing.py)
#pragma unroll
    for (int e = 0; e < N; e += 2) {
        acc[e] = fma_mix_h<0>(raw[e >> 1], k[e], acc[e]);
        acc[e + 1] = fma_mix_h<1>(raw[e >> 1], k[e + 1], acc[e + 1]);
    }
}

// fp16 store of 4 output channels of output pixel `pix`; with p.up == 2 the pixel is replicated to its
// 2x2 block of the (2Ho, 2Wo) destination view: the nearest x2 [upsample] layer (yolo2onnx.py:806-836)
// folded into its producer, one launch fewer per PAN level.
__device__ __forceinline__ void store_out(const ConvParams& p, long pix, int co, f16x4 o) {
    if (p.up == 2) {
        const int hw = p.Ho * p.Wo, rem = (int)(pix % hw);
        const size_t o00 = ((size_t)(pix / hw) * 2 * p.Ho + 2 * (rem / p.Wo)) * (2 * p.Wo) + 2 * (rem % p.Wo);
        f16* dst = p.out + o00 * p.out_cs + p.out_coff + co;
        *reinterpret_cast<f16x4*>(dst) = o;
        *reinterpret_cast<f16x4*>(dst + p.out_cs) = o;
        *reinterpret_cast<f16x4*>(dst + (size_t)2 * p.Wo * p.out_cs) = o;
        *reinterpret_cast<f16x4*>(dst + (size_t)(2 * p.Wo + 1) * p.out_cs) = o;
    } else {
        *reinterpret_cast<f16x4*>(p.out + (size_t)pix * p.out_cs + p.out_coff + co) = o;
    }
}

// Activations.  The divisions are v_rcp_f32 (1 ulp) instead of IEEE divides: ~10 VALU less per element, far below
// the fp16 rounding of every stored activation (Mish runs on up to 11.8 M outputs per layer of YOLOv4).
__device__ __forceinline__ float act_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float act_leaky(float x) { return x > 0.f ? x : 0.1f * x; }
__device__ __forceinline__ float act_relu(float x) { return x > 0.f ? x : 0.f; }
__device__ __forceinline__ float act_mish(float x) {   // x * tanh(softplus(x)),  tanh(log(1+e^x)) = t/(t+2), t = n^2+2n, n = e^x
    const float n = __expf(x);
    const float t = n * (n + 2.f);
    return x > 20.f ? x : x * (t * act_rcp(t + 2.f));
}
__device__ __forceinline__ float act_logistic(float x) { return act_rcp(1.f + __expf(-x)); }
__device__ __forceinline__ float act_swish(float x) { return x * act_rcp(1.f + __expf(-x)); }

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_LEAKY: return act_leaky(x);
        case ACT_MISH: return act_mish(x);
        case ACT_RELU: return act_relu(x);
        case ACT_LOGISTIC: return act_logistic(x);
        case ACT_SWISH: return act_swish(x);
        default: return x;
    }
}

// N values at once: ONE (wave-uniform) branch on the activation per group.  With apply_act() per element the
// compiler kept a switch per element (litechain.hip's depthwise phase: 8 x ~25 scalar/branch instructions per
// 8 outputs, more than the 72 FMAs they followed).
template <int N>
__device__ __forceinline__ void apply_act_n(float* v, int act) {
    switch (act) {
        case ACT_LEAKY:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = act_leaky(v[e]);
            break;
        case ACT_MISH:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = act_mish(v[e]);
            break;
        case ACT_RELU:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = act_relu(v[e]);
            break;
        case ACT_LOGISTIC:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = act_logistic(v[e]);
            break;
        case ACT_SWISH:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = act_swish(v[e]);
            break;
        default: break;
    }
}

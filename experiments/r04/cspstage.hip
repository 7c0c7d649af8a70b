// Fused first CSP stage of CSPDarknet53 (round 4): everything between the stage's stride-2 conv and its output in ONE
// launch.  yolov4.cfg sections (yolo2onnx.py:558-760 semantics):
//
//     d (64)  --1x1-->  A (64) ------------------------------------------------+
//        \--1x1-->  b (64) --1x1--> r (32) --3x3--> + b = b' (64) --1x1--> c (64) --[c | A] 1x1--> out (64)
//
// Unfused this is 4 launches on the 304 x 304 map (merged sibling 1x1 64->128, fused residual unit, 1x1 64->64, 1x1 128->64:
// 17.9 + 17.2 + 12.8 + 11.2 = 59 us, rows 2-5 of profiles/r03_yolo_layer_roofline.txt) that move 118 MB through HBM / L2 for
// 23.7 MB of algorithmic traffic (d in, out out): every intermediate is written and read back.  Here a workgroup owns an
// 8 x 8 pixel tile (+ 1 pixel of halo for the 3x3) and keeps every intermediate in LDS, rounded to fp16 exactly where the
// unfused layers store it, zero outside the image exactly where the 3x3 pads:
//
//   P0  d halo (10 x 10 positions x 64 ch)                       global -> LDS   (zero outside the image)
//   P1  [b | A] = act(W2 d)      128 couts x 100 positions, K 64 ; b for the halo, A for the interior
//   P2  r = act(W3a b)            32 couts x 100 positions, K 64 ; ZERO outside the image (the 3x3's padding)
//   P3  b' = b + act(W3b * r)     64 couts x 64 pixels, K 9 x 32  (3x3 taps read r at shifted halo positions)
//   P4  c = act(W4 b')            64 x 64, K 64
//   P5  out = act(W5 [c | A])     64 x 64, K 128                  -> LDS -> 16-byte NHWC stores
//
// All five weight matrices (80 KB, MFMA A-fragment order) are fetched into REGISTERS in the prologue, beside the halo load:
// one exposed round trip per workgroup, none between the phases; the B operands come from the LDS tiles (rows padded by
// 16 B: conflict-free ds_read_b128).  4 waves; a phase's (cout tile, pixel tile) items are dealt to the waves.
// Arithmetic is the unfused layers': v_mfma_f32_32x32x16_f16 over ascending K, fp32 bias + activation (+ shortcut), one
// rounding to fp16 per layer -- the stage output is bit-identical to the four launches (tests/test_conv_gpu.py).
//
// MEASURED (round 4, profiles/r04_cspstage_ab.txt): 73 us against the four launches' 59 us -- the table builder therefore uses
// it only with FASTMOT_CSPSTAGE=1.  HBM traffic was not what bounded these layers: a thread of this kernel executes ~6000
// instructions (5 epilogues with Mish, fp16 packing, LDS staging, index arithmetic around 50 MFMAs) with two workgroups =
// 2 wavefronts per SIMD resident (74 KB of LDS, 200 registers: the weights), the unfused kernels run the same instruction
// volume at 16 wavefronts per CU.  At 304 x 304 these layers are bound by instruction issue per output element (~56
// instructions per output in the tiled kernel), not by bytes: the lever is a leaner epilogue, not fewer round trips.
#include "net.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CS_TH = 8, CS_TW = 8, CS_HW = CS_TW + 2, CS_NPOS = (CS_TH + 2) * CS_HW;   // 100 halo positions
constexpr int CS_ROWS = 128;                 // positions padded to 4 MFMA pixel tiles
constexpr int CS_C = 64, CS_M = 32;          // channels of d / b / A / c / out, bottleneck width
constexpr int CS_LD64 = CS_C + 8, CS_LD32 = CS_M + 8, CS_LD128 = 2 * CS_C + 8;
constexpr size_t CS_LDS = (size_t)(CS_ROWS * CS_LD64 * 2      // Dt: d on the halo (later: output staging)
                                   + CS_ROWS * CS_LD64       // Bt: b on the halo
                                   + 64 * CS_LD128           // CA: [c | A] on the interior
                                   + CS_ROWS * CS_LD32       // Rt: r on the halo
                                   + 64 * CS_LD64) * 2       // Bp: b' on the interior
                        + 352 * 4;                                    // the five bias vectors (float32)

struct CspArgs {
    const f16* x; int x_cs, x_coff;          // d
    f16* out; int out_cs, out_coff;
    const f16 *w2, *w3a, *w3b, *w4, *w5;     // A-fragment order [cout/32][K/16][lane][8]
    const float *b2, *b3a, *b3b, *b4, *b5;
    int H, W, tiles_x, act;
};

__global__ __launch_bounds__(256, 2) void cspstage1_kernel(const CspArgs a) {
    extern __shared__ __attribute__((aligned(16))) f16 cs_lds[];
    f16* Dt = cs_lds;
    f16* Bt = Dt + CS_ROWS * CS_LD64;
    f16* CA = Bt + CS_ROWS * CS_LD64;
    f16* Rt = CA + 64 * CS_LD128;
    f16* Bp = Rt + CS_ROWS * CS_LD32;
    float* sbias = reinterpret_cast<float*>(Bp + 64 * CS_LD64);      // [b2 128 | b3a 32 | b3b 64 | b4 64 | b5 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fk = (lane >> 5) * 8, hi4 = 4 * (lane >> 5);
    const int tile = blockIdx.x;
    const int ty0 = (tile / a.tiles_x) * CS_TH, tx0 = (tile % a.tiles_x) * CS_TW;
    const long n = blockIdx.y;
    const f16* img = a.x + n * (long)a.H * a.W * a.x_cs + a.x_coff;

    // ---- weights of every phase -> registers (fragment loads: 1 KB contiguous per wave instruction)
    auto frag = [&](const f16* w, int ct, int ksteps, int u) {
        return *reinterpret_cast<const f16x8*>(w + (((long)ct * ksteps + u) * 64 + lane) * 8);
    };
    f16x8 w2f[4], w3af[4], w3bf[18], w4f[4], w5f[8];
    const int ct2 = wave & 1, pt2 = wave >> 1;       // (cout tile, pixel tile) of the 64 x 64 phases
#pragma unroll
    for (int u = 0; u < 4; ++u) w2f[u] = frag(a.w2, wave, 4, u);          // P1: wave = cout tile of [b | A]
#pragma unroll
    for (int u = 0; u < 4; ++u) w3af[u] = frag(a.w3a, 0, 4, u);
#pragma unroll
    for (int u = 0; u < 18; ++u) w3bf[u] = frag(a.w3b, ct2, 18, u);
#pragma unroll
    for (int u = 0; u < 4; ++u) w4f[u] = frag(a.w4, ct2, 4, u);
#pragma unroll
    for (int u = 0; u < 8; ++u) w5f[u] = frag(a.w5, ct2, 8, u);

    // ---- P0: halo of d -> LDS (16 B per lane, a position's 64 channels contiguous), zero outside the image / past NPOS;
    // the biases go to LDS as well: a global load inside a phase's epilogue would put an L2 round trip on the workgroup's
    // critical path per use (the first version did: 32 of them in a row, 78 us for the stage against 59 us unfused)
    {
        constexpr int NIT = CS_ROWS * 8 / 256;
        f16x8 ld[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + i * 256, pos = e >> 3, c8 = (e & 7) * 8;
            const int py = ty0 - 1 + pos / CS_HW, px = tx0 - 1 + pos % CS_HW;
            ld[i] = *reinterpret_cast<const f16x8*>(
                img + ((long)min(max(py, 0), a.H - 1) * a.W + min(max(px, 0), a.W - 1)) * a.x_cs + c8);
        }
        if (tid < 88) *reinterpret_cast<float4*>(sbias + 4 * tid) = *reinterpret_cast<const float4*>(a.b2 + 4 * tid);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + i * 256, pos = e >> 3, c8 = (e & 7) * 8;
            const int py = ty0 - 1 + pos / CS_HW, px = tx0 - 1 + pos % CS_HW;
            const bool ok = pos < CS_NPOS && py >= 0 && py < a.H && px >= 0 && px < a.W;
            f16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? ld[i][j] : (f16)0.f;
            *reinterpret_cast<f16x8*>(&Dt[pos * CS_LD64 + c8]) = v;
        }
    }
    __syncthreads();

    // epilogue helper: acc (32 couts x 32 positions) + bias -> activation -> 4 x (4 consecutive couts) per lane
    auto finish = [&](const f32x16& acc, int bias_off, int ct, float (&v)[4][4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4*>(sbias + bias_off + ct * 32 + 8 * g + hi4);
            v[g][0] = acc[4 * g + 0] + b.x; v[g][1] = acc[4 * g + 1] + b.y;
            v[g][2] = acc[4 * g + 2] + b.z; v[g][3] = acc[4 * g + 3] + b.w;
            apply_act_n<4>(v[g], a.act);
        }
    };
    auto store4 = [&](f16* dst, const float (&v)[4]) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
        *reinterpret_cast<f16x4*>(dst) = o;
    };

    // ---- P1: [b | A] = act(W2 d) -- wave w: cout tile w (0, 1: b -> Bt on every position; 2, 3: A -> CA on the interior)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const f16* bsrc = Dt + (pt * 32 + frow) * CS_LD64 + fk;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[u], *reinterpret_cast<const f16x8*>(bsrc + u * 16), acc, 0, 0, 0);
        float v[4][4];
        finish(acc, 0, wave, v);
        const int pos = pt * 32 + frow, py = pos / CS_HW, px = pos % CS_HW;
        if (wave < 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) store4(&Bt[pos * CS_LD64 + wave * 32 + 8 * g + hi4], v[g]);
        } else if (pos < CS_NPOS && py >= 1 && py <= CS_TH && px >= 1 && px <= CS_TW) {
            const int idx = (py - 1) * CS_TW + (px - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) store4(&CA[idx * CS_LD128 + CS_C + (wave - 2) * 32 + 8 * g + hi4], v[g]);
        }
    }
    __syncthreads();

    // ---- P2: r = act(W3a b) on the halo, zero outside the image -- wave w: position tile w
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const f16* bsrc = Bt + (wave * 32 + frow) * CS_LD64 + fk;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3af[u], *reinterpret_cast<const f16x8*>(bsrc + u * 16), acc, 0, 0, 0);
        float v[4][4];
        finish(acc, 128, 0, v);
        const int pos = wave * 32 + frow;
        const int py = ty0 - 1 + pos / CS_HW, px = tx0 - 1 + pos % CS_HW;
        const bool inside = pos < CS_NPOS && py >= 0 && py < a.H && px >= 0 && px < a.W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (!inside) { v[g][0] = v[g][1] = v[g][2] = v[g][3] = 0.f; }
            store4(&Rt[pos * CS_LD32 + 8 * g + hi4], v[g]);
        }
    }
    __syncthreads();

    const int pix = pt2 * 32 + frow;                          // interior pixel of this lane in the 64 x 64 phases
    const int iy = pix / CS_TW, ix = pix % CS_TW;
    // ---- P3: b' = b + act(W3b * r): 9 taps x 2 MFMA steps, B fragments from Rt at the shifted halo position
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f16* bsrc = Rt + ((iy + t / 3) * CS_HW + ix + t % 3) * CS_LD32 + fk;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3bf[t * 2 + u], *reinterpret_cast<const f16x8*>(bsrc + u * 16), acc, 0, 0, 0);
        }
        float v[4][4];
        finish(acc, 160, ct2, v);
        const f16* rsrc = Bt + ((iy + 1) * CS_HW + ix + 1) * CS_LD64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = ct2 * 32 + 8 * g + hi4;
            const f16x4 rv = *reinterpret_cast<const f16x4*>(rsrc + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[g][e] += (float)rv[e];
            store4(&Bp[pix * CS_LD64 + co], v[g]);
        }
    }
    __syncthreads();

    // ---- P4: c = act(W4 b') -> CA[:, 0..63]
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const f16* bsrc = Bp + pix * CS_LD64 + fk;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4f[u], *reinterpret_cast<const f16x8*>(bsrc + u * 16), acc, 0, 0, 0);
        float v[4][4];
        finish(acc, 224, ct2, v);
#pragma unroll
        for (int g = 0; g < 4; ++g) store4(&CA[pix * CS_LD128 + ct2 * 32 + 8 * g + hi4], v[g]);
    }
    __syncthreads();

    // ---- P5: out = act(W5 [c | A]) -> staging tile (the space of Dt) -> 16-byte stores of whole NHWC rows
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const f16* bsrc = CA + pix * CS_LD128 + fk;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w5f[u], *reinterpret_cast<const f16x8*>(bsrc + u * 16), acc, 0, 0, 0);
        float v[4][4];
        finish(acc, 288, ct2, v);
#pragma unroll
        for (int g = 0; g < 4; ++g) store4(&Dt[pix * CS_LD64 + ct2 * 32 + 8 * g + hi4], v[g]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 64 * 8 / 256; ++i) {
        const int e = tid + i * 256, p = e >> 3, c8 = (e & 7) * 8;
        const int gy = ty0 + p / CS_TW, gx = tx0 + p % CS_TW;
        if (gy < a.H && gx < a.W)
            *reinterpret_cast<uint4*>(a.out + ((n * a.H + gy) * (long)a.W + gx) * a.out_cs + a.out_coff + c8) =
                *reinterpret_cast<const uint4*>(&Dt[p * CS_LD64 + c8]);
    }
}

}  // namespace

bool cspstage_supported(int c, int mid) { return c == CS_C && mid == CS_M; }

// weights: w_off -> [W2 (128 x 64) | W3a (32 x 64) | W3b (64 x 288) | W4 (64 x 64) | W5 (64 x 128)] in A-fragment order,
// b_off -> [b2 (128) | b3a (32) | b3b (64) | b4 (64) | b5 (64)] float32
int launch_cspstage(const f16* x, int x_cs, int x_coff, f16* out, int out_cs, int out_coff, const f16* w, const float* b,
                    int N, int H, int W, int C, int M, int act, hipStream_t s) {
    FM_CHECK_ARG(cspstage_supported(C, M) && x_cs % 8 == 0 && x_coff % 8 == 0 && out_cs % 8 == 0 && out_coff % 8 == 0);
    static bool configured = false;
    if (!configured) {
        FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cspstage1_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS_LDS));
        configured = true;
    }
    CspArgs a{};
    a.x = x; a.x_cs = x_cs; a.x_coff = x_coff;
    a.out = out; a.out_cs = out_cs; a.out_coff = out_coff;
    a.w2 = w;
    a.w3a = a.w2 + 128 * 64;
    a.w3b = a.w3a + 32 * 64;
    a.w4 = a.w3b + 64 * 288;
    a.w5 = a.w4 + 64 * 64;
    a.b2 = b; a.b3a = b + 128; a.b3b = b + 160; a.b4 = b + 224; a.b5 = b + 288;
    a.H = H; a.W = W; a.tiles_x = (W + CS_TW - 1) / CS_TW; a.act = act;
    const int tiles_y = (H + CS_TH - 1) / CS_TH;
    hipLaunchKernelGGL(cspstage1_kernel, dim3(a.tiles_x * tiles_y, N), dim3(256), CS_LDS, s, a);
    FM_HIP(hipGetLastError());
    return 0;
}

extern "C" int fm_cspstage_supported(int c, int mid) { return cspstage_supported(c, mid) ? 1 : 0; }

// Streaming 1x1 convolution for the MEMORY-BOUND pointwise layers (round 4).
//
// YOLOv4@608 has ~14 stride-1 1x1 layers with 64..256 input channels on 304^2..76^2 maps (the CSP stages' entry / exit
// convs, the PAN path's lateral convs, the 76^2 head): 1-4 K steps of work per output tile, 23-35 MB of activations each.
// Their roofline is HBM (0.4-4.4 us per layer), the LDS-tiled kernel (conv.hip) runs them at 1.3-3.1 TB/s (5-18 us,
// profiles/r03_yolo_layer_roofline.txt rows 2, 4, 5, 7, 10, 11, 13, 22, 23, 63-70): there a workgroup lives for ONE 64 x 64
// tile -- load operands -> LDS -> barrier -> 4-16 MFMAs -> fp32 tile through LDS -> store, nothing overlaps inside it, and the
// weights (8-128 KB, identical for every tile) are fetched again by each of the 700-2900 workgroups of a launch.
//
// Here a launch is (at most) two PERSISTENT workgroups per CU that walk the pixel tiles:
//   * the layer's weights live in REGISTERS as MFMA A fragments for the whole launch (each of the 4 waves keeps the
//     cout tiles it owns: <= 128 VGPRs);
//   * pixel tiles (64 or 128 pixels x Cin) are read with full-row 16-byte loads, one tile AHEAD: the loads of tile i+1
//     are in flight while tile i is multiplied and stored; they reach the MFMA fragment layout through LDS (double
//     buffered, rows padded by 16 B: conflict-free ds_read_b128);
//   * the fp16 (heads: fp32) output tile goes through LDS once, so that every lane stores 16 B of whole NHWC rows.
// Two barriers per tile.  Same arithmetic as conv.hip (v_mfma_f32_32x32x16_f16 over ascending K, fp32 bias + activation,
// one rounding to fp16): the outputs are BIT-IDENTICAL to the tiled kernel's (tests/test_conv_gpu.py checks exactly that),
// which stays the path for everything this kernel does not take (3x3, strides, residual adds, x2 upsampling, odd channel
// counts, small maps).
//
// MEASURED (round 4, profiles/r04_conv1x1_stream_ab.txt: per-layer times of YOLOv4@608 from rocprofv3 graph-replay traces):
// bit-identical, and SLOWER on 13 of its 14 layers (304^2 64->128: 17.9 -> 21.0 us; 76^2 256->256: 8.3 -> 11.5; the 76^2
// head 6.5 -> 10.8; only 304^2 64->64 gains, 12.8 -> 11.8).  Two persistent workgroups per CU hold 8 wavefronts and one tile
// of loads each in flight, the tiled kernel's four short-lived workgroups per CU hold 16 and four; with 90-1444 tiles per
// layer the persistent grid also quantises badly (722 tiles on 512 workgroups = two rounds, the second 41 % full).  The
// kernel therefore ships switched OFF (fm_ctx option "conv1x1_stream" = 1 enables it; the bit-identity test does).
#include "net.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int g_conv1x1_stream = 0;          // fm_ctx option "conv1x1_stream": OFF by default, see the measurement above

template <int CIN, int CT, int PXT, bool F32OUT>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const ConvParams p, int ntiles) {
    constexpr int WC = CT < 4 ? CT : 4, WP = 4 / WC, MC = CT / WC, MP = PXT / 32 / WP;
    static_assert(WC * WP == 4 && MC * WC == CT && MP * WP * 32 == PXT && MP >= 1, "wave split");
    constexpr int KS = CIN / 16;                       // MFMA steps
    constexpr int LDB = CIN + 8;                       // halfs per staged pixel row (+16 B)
    constexpr int COUT = CT * 32;
    constexpr int LDO = F32OUT ? COUT + 4 : COUT + 8;  // elements per output row of the staging tile
    constexpr int NL = PXT * CIN / 8 / 256;            // 16-byte loads per thread and tile
    constexpr int CPR = CIN / 8;                       // 16-byte chunks per pixel row
    static_assert(PXT * CIN / 8 % 256 == 0, "tile loads");
    extern __shared__ __attribute__((aligned(16))) char c1_sm[];
    f16* sB = reinterpret_cast<f16*>(c1_sm);                                   // [2][PXT][LDB]
    char* sO = c1_sm + (size_t)2 * PXT * LDB * 2;                              // [PXT][LDO] f16 / f32
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wv / WP, wp = wv % WP;
    const int frow = lane & 31, fk = (lane >> 5) * 8;

    // ---- the wave's weights and biases: registers, once
    f16x8 af[MC][KS];
    float4 bias[MC][4];
#pragma unroll
    for (int mi = 0; mi < MC; ++mi) {
        const f16* wr = p.w + (size_t)((wc * MC + mi) * 32 + frow) * p.Kpad + fk;
#pragma unroll
        for (int k = 0; k < KS; ++k) af[mi][k] = *reinterpret_cast<const f16x8*>(wr + k * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias[mi][g] = *reinterpret_cast<const float4*>(p.bias + (wc * MC + mi) * 32 + 8 * g + 4 * (lane >> 5));
    }
    const f16* src = p.in + p.in_coff;
    // (named scalars, no array: a register array handed to a lambda ends up in scratch memory -- conv.hip has the story)
    static_assert(NL <= 8, "stage registers");
    uint4 st0, st1, st2, st3, st4, st5, st6, st7;
    st0 = st1 = st2 = st3 = st4 = st5 = st6 = st7 = make_uint4(0, 0, 0, 0);
#define C1_LOAD(I, T)                                                                                   \
    if constexpr ((I) < NL) {                                                                           \
        const int e = tid + 256 * (I), row = e / CPR, ch = e % CPR;                                     \
        const long pix = min((long)(T) * PXT + row, (long)p.P - 1);   /* clamped rows are never stored */ \
        st##I = *reinterpret_cast<const uint4*>(src + pix * p.in_cs + ch * 8);                           \
    }
#define C1_STORE(I, BUF)                                                                                \
    if constexpr ((I) < NL) {                                                                           \
        const int e = tid + 256 * (I), row = e / CPR, ch = e % CPR;                                     \
        *reinterpret_cast<uint4*>(sB + ((size_t)(BUF) * PXT + row) * LDB + ch * 8) = st##I;             \
    }
#define load_tile(T) { C1_LOAD(0, T) C1_LOAD(1, T) C1_LOAD(2, T) C1_LOAD(3, T) C1_LOAD(4, T) C1_LOAD(5, T) C1_LOAD(6, T) C1_LOAD(7, T) }
#define store_tile(BUF) { C1_STORE(0, BUF) C1_STORE(1, BUF) C1_STORE(2, BUF) C1_STORE(3, BUF) C1_STORE(4, BUF) C1_STORE(5, BUF) C1_STORE(6, BUF) C1_STORE(7, BUF) }
    int t = blockIdx.x;
    if (t >= ntiles) return;
    load_tile(t)
    store_tile(0)
    __syncthreads();
    int cur = 0;
    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < ntiles) load_tile(tn)                                         // next tile: in flight under everything below
        // ---- multiply
        f32x16 acc[MC][MP];
#pragma unroll
        for (int mi = 0; mi < MC; ++mi)
#pragma unroll
            for (int pi = 0; pi < MP; ++pi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][pi][r] = 0.f;
        const f16* bt = sB + (size_t)cur * PXT * LDB;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            f16x8 bf[MP];
#pragma unroll
            for (int pi = 0; pi < MP; ++pi)
                bf[pi] = *reinterpret_cast<const f16x8*>(bt + ((wp * MP + pi) * 32 + frow) * LDB + k * 16 + fk);
#pragma unroll
            for (int mi = 0; mi < MC; ++mi)
#pragma unroll
                for (int pi = 0; pi < MP; ++pi)
                    acc[mi][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][k], bf[pi], acc[mi][pi], 0, 0, 0);
        }
        // ---- bias + activation, tile to LDS in NHWC order
#pragma unroll
        for (int mi = 0; mi < MC; ++mi)
#pragma unroll
            for (int pi = 0; pi < MP; ++pi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[mi][pi][4 * g + 0] + bias[mi][g].x, acc[mi][pi][4 * g + 1] + bias[mi][g].y,
                                  acc[mi][pi][4 * g + 2] + bias[mi][g].z, acc[mi][pi][4 * g + 3] + bias[mi][g].w};
                    apply_act_n<4>(v, p.act);
                    const int row = (wp * MP + pi) * 32 + frow, col = (wc * MC + mi) * 32 + 8 * g + 4 * (lane >> 5);
                    if constexpr (F32OUT) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(sO) + (size_t)row * LDO + col) =
                            make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        f16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
                        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(sO) + (size_t)row * LDO + col) = o;
                    }
                }
        __syncthreads();                 // output tile complete; every wave is done reading sB[cur]
        // ---- 16-byte stores of whole rows
        {
            constexpr int EPC = F32OUT ? 4 : 8;                                // elements per 16-byte chunk
            constexpr int CH = COUT / EPC, ROWS = 256 / CH;
            static_assert(256 % CH == 0 && PXT % ROWS == 0, "store mapping");
            const int och = tid % CH, orow = tid / CH, co = och * EPC;
            if (co < p.cout_store) {
#pragma unroll
                for (int it = 0; it < PXT / ROWS; ++it) {
                    const int row = it * ROWS + orow;
                    const long pix = (long)t * PXT + row;
                    if (pix >= p.P) break;
                    if constexpr (F32OUT) {
                        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(sO) + (size_t)row * LDO + co);
                        *reinterpret_cast<float4*>(p.out32 + pix * p.out_cs + p.out_coff + co) = v;
                    } else {
                        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const f16*>(sO) + (size_t)row * LDO + co);
                        *reinterpret_cast<uint4*>(p.out + pix * p.out_cs + p.out_coff + co) = v;
                    }
                }
            }
        }
        if (tn < ntiles) store_tile(cur ^ 1)
        cur ^= 1;
        __syncthreads();                 // next operand tile in place; the output tile may be overwritten
    }
#undef load_tile
#undef store_tile
#undef C1_LOAD
#undef C1_STORE
}

template <int CIN, int CT, int PXT, bool F32OUT>
int launch_c1(const ConvParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * PXT * (CIN + 8) * 2 + (size_t)PXT * (F32OUT ? (CT * 32 + 4) * 4 : (CT * 32 + 8) * 2);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool configured = false;
    if (!configured) {
        FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_stream_kernel<CIN, CT, PXT, F32OUT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const int ntiles = (p.P + PXT - 1) / PXT;
    const int per_cu = lds <= 76 * 1024 ? 2 : 1;
    const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    hipLaunchKernelGGL((conv1x1_stream_kernel<CIN, CT, PXT, F32OUT>), dim3(grid), dim3(256), lds, s, p, ntiles);
    FM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

void conv1x1_stream_enable(int on) { g_conv1x1_stream = on; }

// Takes the layer if it is one of the shapes above; *taken says whether it did.
int launch_conv1x1_stream(const ConvParams& p, hipStream_t s, bool* taken) {
    *taken = false;
    if (!g_conv1x1_stream) return 0;
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.up == 2 || p.res_mode != RES_NONE) return 0;
    if (p.Ho != p.H || p.Wo != p.W || p.K != p.Cin || p.Kpad != p.Cin) return 0;
    if (p.P < 4096) return 0;                        // small maps: too few tiles to stream (conv.hip / convs.hip)
    if (p.in_cs % 8 || p.in_coff % 8 || p.out_cs % 8 || p.out_coff % 8) return 0;
    const int ct = (p.Cout + 31) / 32;
    const bool f32 = p.out32 != nullptr;
    if (p.cout_store > ct * 32) return 0;
#define C1(CIN_, CT_, PXT_)                                                                    \
    if (p.Cin == CIN_ && ct == CT_) {                                                          \
        *taken = true;                                                                         \
        return f32 ? launch_c1<CIN_, CT_, PXT_, true>(p, s) : launch_c1<CIN_, CT_, PXT_, false>(p, s); \
    }
    C1(64, 2, 128) C1(64, 4, 128) C1(128, 2, 128) C1(128, 4, 128) C1(256, 4, 64) C1(256, 8, 64) C1(128, 8, 64)
#undef C1
    return 0;
}

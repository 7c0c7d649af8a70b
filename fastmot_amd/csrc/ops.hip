// Memory-bound layers of the detector / ReID networks (NHWC fp16, 8 channels = 16 B per lane).
// Graph semantics: scripts/yolo2onnx.py:733-863 (route/upsample/maxpool) and the torchreid OSNet
// definition summarised in SURVEY.md appendix A (LightConv3x3, channel gate, pooling, head).
// All of these are HBM/L2-bandwidth bound: one pass, 16-byte coalesced accesses, no re-reads.
#include "net.h"
#include <atomic>

namespace {

// depthwise 3x3, stride 1, pad 1, + bias (folded BN) + activation.  w: [9][C] fp16, bias f32[C]
__global__ void dwconv3_kernel(const f16* __restrict__ in, int in_cs, int in_coff,
                               f16* __restrict__ out, int out_cs, int out_coff,
                               const f16* __restrict__ w, const float* __restrict__ bias,
                               int N, int H, int W, int C, int act) {
    const int c8n = C / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * H * W * c8n;
    if (idx >= total) return;
    const int c = (int)(idx % c8n) * 8;
    const long pix = idx / c8n;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long n = pix / ((long)W * H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias[c + e];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            float v[8], k[8];
            unpack8(*reinterpret_cast<const uint4*>(in + ((n * H + yy) * W + xx) * in_cs + in_coff + c), v);
            unpack8(*reinterpret_cast<const uint4*>(w + ((dy + 1) * 3 + (dx + 1)) * C + c), k);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], k[e], acc[e]);
        }
    }
    apply_act_n<8>(acc, act);
    *reinterpret_cast<uint4*>(out + pix * out_cs + out_coff + c) = pack8(acc);
}

// darknet SPP block in one launch: stride-1 max pools k = 5, 9, 13 (window clipped at the border) of
// one (sample, 8-channel group) per workgroup, entirely in LDS.  pool9 = pool5(pool5), pool13 =
// pool5(pool9) and each pool5 is a row pass + a column pass: 6 passes of 5 taps instead of
// 25 + 81 + 169 taps, one read of the input and three launches fewer.
//   out channel offsets: +0 -> k13, +C -> k9, +2C -> k5  (yolov4.cfg route -1,-3,-5,-6 order)
constexpr int SPP_MAX_HW = 2048;
__device__ __forceinline__ f16x8 hmax8(f16x8 a, f16x8 b) { return __builtin_elementwise_max(a, b); }
__global__ __launch_bounds__(256) void spp_kernel(const f16* __restrict__ in, int in_cs, int in_coff,
                                                  f16* __restrict__ out, int out_cs, int out_coff, int H,
                                                  int W, int C) {
    __shared__ f16x8 a[SPP_MAX_HW], b[SPP_MAX_HW];
    const int c8n = C / 8, cg = blockIdx.x % c8n;
    const long n = blockIdx.x / c8n;
    const int HW = H * W;
    const f16* src = in + n * HW * in_cs + in_coff + cg * 8;
    f16* dst = out + n * HW * out_cs + out_coff + cg * 8;
    for (int i = threadIdx.x; i < HW; i += 256) a[i] = *reinterpret_cast<const f16x8*>(src + (long)i * in_cs);
    __syncthreads();
    for (int level = 2; level >= 0; --level) {
        for (int i = threadIdx.x; i < HW; i += 256) {   // row pass a -> b
            const int x = i % W, row = i - x;
            f16x8 m = a[i];
            for (int d = -2; d <= 2; ++d) {
                const int xx = min(max(x + d, 0), W - 1);
                m = hmax8(m, a[row + xx]);
            }
            b[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < HW; i += 256) {   // column pass b -> a (+ store)
            const int y = i / W, x = i - y * W;
            f16x8 m = b[i];
            for (int d = -2; d <= 2; ++d) {
                const int yy = min(max(y + d, 0), H - 1);
                m = hmax8(m, b[yy * W + x]);
            }
            a[i] = m;
            *reinterpret_cast<f16x8*>(dst + (long)i * out_cs + level * C) = m;
        }
        __syncthreads();
    }
}

// max / average pooling, window k, stride s, pad (window clipped at the border)
__global__ void pool_kernel(const f16* __restrict__ in, int in_cs, int in_coff, f16* __restrict__ out,
                            int out_cs, int out_coff, int N, int H, int W, int C, int Ho, int Wo,
                            int k, int s, int pad, int avg) {
    const int c8n = C / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * Ho * Wo * c8n;
    if (idx >= total) return;
    const int c = (int)(idx % c8n) * 8;
    const long pix = idx / c8n;
    const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho);
    const long n = pix / ((long)Wo * Ho);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = avg ? 0.f : -65504.f;
    const int y0 = yo * s - pad, x0 = xo * s - pad;
    for (int dy = 0; dy < k; ++dy) {
        const int yy = y0 + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = x0 + dx;
            if (xx < 0 || xx >= W) continue;
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(in + ((n * H + yy) * W + xx) * in_cs + in_coff + c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = avg ? acc[e] + v[e] : fmaxf(acc[e], v[e]);
        }
    }
    if (avg) {
        const float inv = 1.f / (float)(k * k);   // AvgPool2d(2, 2): full windows only
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= inv;
    }
    *reinterpret_cast<uint4*>(out + pix * out_cs + out_coff + c) = pack8(acc);
}

// nearest-neighbour x2 upsample into a channel slice (yolo2onnx.py:806-836)
__global__ void upsample2_kernel(const f16* __restrict__ in, int in_cs, int in_coff,
                                 f16* __restrict__ out, int out_cs, int out_coff, int N, int H, int W,
                                 int C) {
    const int c8n = C / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * (2 * H) * (2 * W) * c8n;
    if (idx >= total) return;
    const int c = (int)(idx % c8n) * 8;
    const long pix = idx / c8n;
    const int xo = (int)(pix % (2 * W)), yo = (int)((pix / (2 * W)) % (2 * H));
    const long n = pix / ((long)4 * W * H);
    const uint4 v = *reinterpret_cast<const uint4*>(in + ((n * H + (yo >> 1)) * W + (xo >> 1)) * in_cs + in_coff + c);
    *reinterpret_cast<uint4*>(out + pix * out_cs + out_coff + c) = v;
}

// channel slice copy (route of a tensor that already lives elsewhere)
__global__ void copy_kernel(const f16* __restrict__ in, int in_cs, int in_coff, f16* __restrict__ out,
                            int out_cs, int out_coff, long npix, int C) {
    const int c8n = C / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c8n) return;
    const int c = (int)(idx % c8n) * 8;
    const long pix = idx / c8n;
    *reinterpret_cast<uint4*>(out + pix * out_cs + out_coff + c) =
        *reinterpret_cast<const uint4*>(in + pix * in_cs + in_coff + c);
}

// element-wise sum of two channel slices ([shortcut] whose operand is also read elsewhere, so that it
// cannot be folded into the producing conv's epilogue; yolo2onnx.py _make_shortcut_node)
__global__ void add_kernel(const f16* __restrict__ a, int a_cs, int a_coff, const f16* __restrict__ b, int b_cs,
                           int b_coff, f16* __restrict__ out, int out_cs, int out_coff, long npix, int C) {
    const int c8n = C / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * c8n) return;
    const int c = (int)(idx % c8n) * 8;
    const long pix = idx / c8n;
    float x[8], y[8];
    unpack8(*reinterpret_cast<const uint4*>(a + pix * a_cs + a_coff + c), x);
    unpack8(*reinterpret_cast<const uint4*>(b + pix * b_cs + b_coff + c), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    *reinterpret_cast<uint4*>(out + pix * out_cs + out_coff + c) = pack8(x);
}

// OSNet channel gate: gate[n][c] = sigmoid(fc2(relu(fc1(GAP(x[n])))))   (one block per sample)
// w1: [hid][C] f16, b1: f32[hid], w2: [C][hid] f16, b2: f32[C]; gate out f32 [N][C]
__global__ __launch_bounds__(256) void gate_kernel(const f16* __restrict__ in, int in_cs, int in_coff,
                                                   int HW, int C, int hid, const f16* __restrict__ w1,
                                                   const float* __restrict__ b1,
                                                   const f16* __restrict__ w2,
                                                   const float* __restrict__ b2,
                                                   float* __restrict__ gate) {
    extern __shared__ float sm[];            // [256 / c8n][C] partial sums | gap[C] | hidden[hid]
    const int n = blockIdx.x, tid = threadIdx.x;
    const int c8n = C / 8;
    const int groups = 256 / c8n;            // pixel groups (C <= 2048/8)
    const int cg = tid % c8n, pg = tid / c8n;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pg < groups) {
        const f16* base = in + (size_t)n * HW * in_cs + in_coff + cg * 8;
        for (int px = pg; px < HW; px += groups) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(base + (size_t)px * in_cs), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[pg * C + cg * 8 + e] = acc[e];
    }
    __syncthreads();
    float* gap = sm + groups * C;
    float* hidden = gap + C;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += sm[g * C + c];
        gap[c] = s / (float)HW;
    }
    __syncthreads();
    for (int h = tid; h < hid; h += 256) {
        float s = b1[h];
        for (int c = 0; c < C; ++c) s = fmaf((float)w1[h * C + c], gap[c], s);
        hidden[h] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = b2[c];
        for (int h = 0; h < hid; ++h) s = fmaf((float)w2[c * hid + h], hidden[h], s);
        gate[(size_t)n * C + c] = 1.f / (1.f + __expf(-s));
    }
}

// x2 = sum_T s_T * gate_T  (OSNet: four gated streams summed)
struct GateSumArgs {
    const f16* in[4];
    int in_cs[4], in_coff[4];
    const float* gate[4];
    int nstreams;
};
__global__ void gate_sum_kernel(GateSumArgs a, f16* __restrict__ out, int out_cs, int out_coff, int N,
                                int HW, int C) {
    const int c8n = C / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * HW * c8n;
    if (idx >= total) return;
    const int c = (int)(idx % c8n) * 8;
    const long pix = idx / c8n;
    const long n = pix / HW;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < a.nstreams; ++t) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(a.in[t] + pix * a.in_cs[t] + a.in_coff[t] + c), v);
        const float* g = a.gate[t] + n * C + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], g[e], acc[e]);
    }
    *reinterpret_cast<uint4*>(out + pix * out_cs + out_coff + c) = pack8(acc);
}

// OSNet unified aggregation gate, whole block tail in ONE launch (one workgroup of 512 threads per sample):
//   gap_t = GAP(x_t);  g_t = sigmoid(fc2(relu(fc1(gap_t))));  out = sum_t x_t * g_t     (t < nstreams <= 4)
// The second pass re-reads x_t from L2 (a sample's four streams are <= 1 MB).  Same arithmetic as
// gate_kernel + gate_sum_kernel (fp32 sums, fmaf in stream order); only the GAP partial grouping differs.
constexpr int GS_THREADS = 512;
__global__ __launch_bounds__(GS_THREADS) void gated_sum_kernel(GateSumArgs a, int HW, int C, int hid,
                                                               const f16* __restrict__ w1,
                                                               const float* __restrict__ b1,
                                                               const f16* __restrict__ w2,
                                                               const float* __restrict__ b2,
                                                               f16* __restrict__ out, int out_cs, int out_coff) {
    extern __shared__ float sm[];            // part[groups][C] | gap[4][C] | hidden[4][hid] | gate[4][C]
    const int n = blockIdx.x, tid = threadIdx.x;
    const int c8n = C / 8;
    const int groups = GS_THREADS / c8n;
    const int cg = tid % c8n, pg = tid / c8n;
    float* gap = sm + groups * C;
    float* hidden = gap + 4 * C;
    float* gate = hidden + 4 * hid;
    const f16* base[4];
    int cs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = t < a.nstreams ? t : 0;
        cs[t] = a.in_cs[q];
        base[t] = a.in[q] + (size_t)n * HW * cs[t] + a.in_coff[q] + cg * 8;
    }
    float acc[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    if (pg < groups) {
        for (int px = pg; px < HW; px += groups) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < a.nstreams) {
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(base[t] + (size_t)px * cs[t]), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[t][e] += v[e];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= a.nstreams) break;
        if (pg < groups) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[pg * C + cg * 8 + e] = acc[t][e];
        }
        __syncthreads();
        for (int c = tid; c < C; c += GS_THREADS) {
            float s = 0.f;
            for (int g = 0; g < groups; ++g) s += sm[g * C + c];
            gap[t * C + c] = s / (float)HW;
        }
        __syncthreads();
    }
    for (int i = tid; i < a.nstreams * hid; i += GS_THREADS) {
        const int t = i / hid, h = i % hid;
        float s = b1[h];
        for (int c = 0; c < C; ++c) s = fmaf((float)w1[h * C + c], gap[t * C + c], s);
        hidden[t * hid + h] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < a.nstreams * C; i += GS_THREADS) {
        const int t = i / C, c = i % C;
        float s = b2[c];
        for (int h = 0; h < hid; ++h) s = fmaf((float)w2[c * hid + h], hidden[t * hid + h], s);
        gate[t * C + c] = 1.f / (1.f + __expf(-s));
    }
    __syncthreads();
    if (pg < groups) {
        f16* dst = out + (size_t)n * HW * out_cs + out_coff + cg * 8;
        for (int px = pg; px < HW; px += groups) {
            float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < a.nstreams) {
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(base[t] + (size_t)px * cs[t]), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = fmaf(v[e], gate[t * C + cg * 8 + e], o[e]);
                }
            }
            *reinterpret_cast<uint4*>(dst + (size_t)px * out_cs) = pack8(o);
        }
    }
}

// Same operation when the producers already left per-tile channel sums (liteconv.hip phase C):
// part[t] = fp32 [N][tiles][C].  The average pool is then a sum of <= 32 numbers per channel, every
// workgroup recomputes the tiny gate MLP for its sample and the gated sum itself runs with full
// parallelism over pixels (grid = pixel chunks x samples) instead of one workgroup per sample.
struct GatedSumPartArgs {
    const f16* in[4];
    int in_cs[4], in_coff[4];
    const float* part[4];
    int nstreams;
};
// Pixels per workgroup: with 256 / (C / 8) (at least 64) every thread owns ONE (pixel, 8-channel group) item whose four
// stream vectors are requested at kernel entry, under the gate MLP's three LDS round trips, and the launch has 2-4x the
// workgroups (x0.25, 64 x 32 stage: 800 instead of 400); the per-workgroup MLP is a few hundred FMAs.
__global__ __launch_bounds__(256) void gated_sum_part_kernel(GatedSumPartArgs a, int HW, int C, int hid, int tiles,
                                                             int GS2_PIX,
                                                             const f16* __restrict__ w1,
                                                             const float* __restrict__ b1,
                                                             const f16* __restrict__ w2,
                                                             const float* __restrict__ b2,
                                                             f16* __restrict__ out, int out_cs, int out_coff) {
    extern __shared__ float sm[];            // gap[4][C] | hidden[4][hid] | gate[4][C] | w1[hid][C] | w2[C][hid] | b1[hid] | b2[C] | parts
    const int n = blockIdx.y, tid = threadIdx.x;
    float* gap = sm;
    float* hidden = gap + 4 * C;
    float* gate = hidden + 4 * hid;
    float* sw1 = gate + 4 * C;
    float* sw2 = sw1 + hid * C;
    float* sb1 = sw2 + C * hid;
    float* sb2 = sb1 + hid;
    // Everything the launch reads before its last phase is requested up front -- the gate MLP's weights go to LDS with
    // the tile sums, and the first pixel group's four stream vectors wait in registers -- so that the kernel is two
    // memory round trips long instead of four (11.6 -> ~6 us per launch; six launches per ReID pass).
    for (int i = tid; i < hid * C; i += 256) { sw1[i] = (float)w1[i]; sw2[i] = (float)w2[i]; }
    for (int i = tid; i < hid; i += 256) sb1[i] = b1[i];
    for (int i = tid; i < C; i += 256) sb2[i] = b2[i];
    const int c8n = C / 8;
    const int p0 = blockIdx.x * GS2_PIX;
    uint4 first[4] = {};
    bool first_ok = false;
    {
        const int px = p0 + tid / c8n, cg = tid % c8n;
        if (tid < GS2_PIX * c8n && px < HW) {
            first_ok = true;
            const size_t pix = (size_t)n * HW + px;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < a.nstreams) first[t] = *reinterpret_cast<const uint4*>(a.in[t] + pix * a.in_cs[t] + a.in_coff[t] + cg * 8);
        }
    }
    // the tile sums of the four streams: all of them requested at once by all threads and parked in LDS, then added per
    // (stream, channel) in tile order -- the order a lone thread walking global memory used, one dependent round trip
    // per tile (8-16 of them in a kernel that is otherwise four round trips long)
    float* parts = sb2 + C;                  // [nstreams][tiles][C]
    const int tc = tiles * C;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= a.nstreams) break;
        const float* p = a.part[t] + (size_t)n * tc;
        for (int i = tid; i < tc; i += 256) parts[t * tc + i] = p[i];
    }
    __syncthreads();
    for (int i = tid; i < a.nstreams * C; i += 256) {
        const int t = i / C, c = i - t * C;
        float s = 0.f;
        for (int q = 0; q < tiles; ++q) s += parts[t * tc + q * C + c];
        gap[i] = s / (float)HW;
    }
    __syncthreads();
    for (int i = tid; i < a.nstreams * hid; i += 256) {
        const int t = i / hid, h = i % hid;
        float s = sb1[h];
        for (int c = 0; c < C; ++c) s = fmaf(sw1[h * C + c], gap[t * C + c], s);
        hidden[t * hid + h] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < a.nstreams * C; i += 256) {
        const int t = i / C, c = i % C;
        float s = sb2[c];
        for (int h = 0; h < hid; ++h) s = fmaf(sw2[c * hid + h], hidden[t * hid + h], s);
        gate[t * C + c] = 1.f / (1.f + __expf(-s));
    }
    __syncthreads();
    for (int idx = tid; idx < GS2_PIX * c8n; idx += 256) {
        const int px = p0 + idx / c8n, cg = idx % c8n;
        if (px >= HW) break;
        const size_t pix = (size_t)n * HW + px;
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < a.nstreams) {
                float v[8];
                const uint4 raw = (idx == tid && first_ok) ? first[t]
                                                           : *reinterpret_cast<const uint4*>(a.in[t] + pix * a.in_cs[t] + a.in_coff[t] + cg * 8);
                unpack8(raw, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(v[e], gate[t * C + cg * 8 + e], o[e]);
            }
        }
        *reinterpret_cast<uint4*>(out + pix * out_cs + out_coff + cg * 8) = pack8(o);
    }
}

// OSNet head: global average pool -> Linear(C -> D) + folded BN1d + ReLU -> L2 normalise
// (models/reid.py OUTPUT_LAYOUT = 512; feature_extractor.py:73).  One block per sample.
__global__ __launch_bounds__(256) void head_kernel(const f16* __restrict__ in, int in_cs, int in_coff,
                                                   int HW, int C, int D, const f16* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ out,
                                                   float* __restrict__ raw_out, float* __restrict__ mirror) {
    extern __shared__ float sm[];   // [groups][C] | gap[C] | feat[D] | red[256]
    const int n = blockIdx.x, tid = threadIdx.x;
    const int c8n = C / 8;
    const int groups = 256 / c8n > 0 ? 256 / c8n : 1;
    const int cg = tid % c8n, pg = tid / c8n;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8n <= 256 && pg < groups) {
        const f16* base = in + (size_t)n * HW * in_cs + in_coff + cg * 8;
        for (int px = pg; px < HW; px += groups) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(base + (size_t)px * in_cs), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[pg * C + cg * 8 + e] = acc[e];
    }
    __syncthreads();
    float* gap = sm + groups * C;
    float* feat = gap + C;
    float* red = feat + D;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += sm[g * C + c];
        gap[c] = s / (float)HW;
    }
    __syncthreads();
    float nrm = 0.f;
    for (int d = tid; d < D; d += 256) {
        float s = b[d];
        const f16* wr = w + (size_t)d * C;
        for (int c = 0; c < C; c += 8) {
            float k[8];
            unpack8(*reinterpret_cast<const uint4*>(wr + c), k);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(k[e], gap[c + e], s);
        }
        s = s > 0.f ? s : 0.f;
        feat[d] = s;
        nrm += s * s;
    }
    red[tid] = nrm;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    const float inv = 1.f / sqrtf(red[0]);
    for (int d = tid; d < D; d += 256) {
        if (raw_out) raw_out[(size_t)n * D + d] = feat[d];
        const float v = feat[d] * inv;
        out[(size_t)n * D + d] = v;
        if (mirror) mirror[(size_t)n * D + d] = v;      // the same row in page-locked host memory (no export launch behind the network)
    }
}

inline dim3 grid1d(long total, int block = 256) { return dim3((unsigned)((total + block - 1) / block)); }

}  // namespace

int launch_dwconv3(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, const f16* w,
                   const float* bias, int N, int H, int W, int C, int act, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && in_cs % 8 == 0 && in_coff % 8 == 0 && out_cs % 8 == 0 && out_coff % 8 == 0);
    const long total = (long)N * H * W * (C / 8);
    hipLaunchKernelGGL(dwconv3_kernel, grid1d(total), dim3(256), 0, s, in, in_cs, in_coff, out, out_cs,
                       out_coff, w, bias, N, H, W, C, act);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_spp(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, int N, int H, int W,
               int C, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && in_cs % 8 == 0 && in_coff % 8 == 0 && out_cs % 8 == 0 && out_coff % 8 == 0);
    FM_CHECK_ARG(H * W <= SPP_MAX_HW);
    hipLaunchKernelGGL(spp_kernel, dim3((unsigned)((long)N * (C / 8))), dim3(256), 0, s, in, in_cs, in_coff, out,
                       out_cs, out_coff, H, W, C);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_pool(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, int N, int H,
                int W, int C, int Ho, int Wo, int k, int stride, int pad, int avg, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && in_cs % 8 == 0 && in_coff % 8 == 0 && out_cs % 8 == 0 && out_coff % 8 == 0);
    const long total = (long)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(pool_kernel, grid1d(total), dim3(256), 0, s, in, in_cs, in_coff, out, out_cs, out_coff,
                       N, H, W, C, Ho, Wo, k, stride, pad, avg);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_upsample2(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, int N, int H,
                     int W, int C, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && in_cs % 8 == 0 && in_coff % 8 == 0 && out_cs % 8 == 0 && out_coff % 8 == 0);
    const long total = (long)N * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2_kernel, grid1d(total), dim3(256), 0, s, in, in_cs, in_coff, out, out_cs,
                       out_coff, N, H, W, C);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_add(const f16* a, int a_cs, int a_coff, const f16* b, int b_cs, int b_coff, f16* out, int out_cs,
               int out_coff, long npix, int C, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && a_cs % 8 == 0 && a_coff % 8 == 0 && b_cs % 8 == 0 && b_coff % 8 == 0 &&
                 out_cs % 8 == 0 && out_coff % 8 == 0);
    hipLaunchKernelGGL(add_kernel, grid1d(npix * (C / 8)), dim3(256), 0, s, a, a_cs, a_coff, b, b_cs, b_coff, out,
                       out_cs, out_coff, npix, C);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_copy(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, long npix, int C,
                hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && in_cs % 8 == 0 && in_coff % 8 == 0 && out_cs % 8 == 0 && out_coff % 8 == 0);
    hipLaunchKernelGGL(copy_kernel, grid1d(npix * (C / 8)), dim3(256), 0, s, in, in_cs, in_coff, out, out_cs,
                       out_coff, npix, C);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_gate(const f16* in, int in_cs, int in_coff, int N, int HW, int C, int hid, const f16* w1,
                const float* b1, const f16* w2, const float* b2, float* gate, hipStream_t s) {
    FM_CHECK_ARG(C % 8 == 0 && C / 8 <= 256 && in_cs % 8 == 0 && in_coff % 8 == 0);
    const int groups = 256 / (C / 8);
    const size_t shmem = sizeof(float) * ((size_t)groups * C + C + hid);
    hipLaunchKernelGGL(gate_kernel, dim3(N), dim3(256), shmem, s, in, in_cs, in_coff, HW, C, hid, w1, b1, w2,
                       b2, gate);
    FM_HIP(hipGetLastError());
    return 0;
}

// part (optional): per stream, fp32 [N][tiles][C] channel sums left by the producing liteconv launches
int launch_gated_sum(int nstreams, const f16* const* in, const int* in_cs, const int* in_coff, int N, int HW,
                     int C, int hid, const f16* w1, const float* b1, const f16* w2, const float* b2, f16* out,
                     int out_cs, int out_coff, const float* const* part, int tiles, hipStream_t s) {
    FM_CHECK_ARG(nstreams >= 1 && nstreams <= 4 && C % 8 == 0 && C / 8 <= GS_THREADS && out_cs % 8 == 0 &&
                 out_coff % 8 == 0 && hid >= 1);
    // the tile sums of all streams are staged in LDS (round 5): shapes whose sums do not fit take the kernel that
    // reduces the maps itself (same gate; the order of its fp32 sums differs)
    const size_t part_lds = ((size_t)8 * C + 4 * hid + 2 * (size_t)hid * C + hid + C + (size_t)nstreams * tiles * C) * sizeof(float);
    if (part && part_lds > 150 * 1024) part = nullptr;
    if (part) {
        GatedSumPartArgs a{};
        a.nstreams = nstreams;
        for (int t = 0; t < nstreams; ++t) {
            FM_CHECK_ARG(in_cs[t] % 8 == 0 && in_coff[t] % 8 == 0 && part[t]);
            a.in[t] = in[t]; a.in_cs[t] = in_cs[t]; a.in_coff[t] = in_coff[t]; a.part[t] = part[t];
        }
        const size_t shmem = part_lds;
        if (shmem > 64 * 1024) {          // opt-in per device (GATE_SLOT_TILES = 32 tiles x 128 channels x 4 streams is 64 KB + the MLP)
            static std::atomic<unsigned long long> configured{0};
            int dev = 0;
            FM_HIP(hipGetDevice(&dev));
            if (dev >= 64 || !(configured.load(std::memory_order_relaxed) >> dev & 1)) {
                FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gated_sum_part_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                if (dev < 64) configured.fetch_or(1ull << dev, std::memory_order_relaxed);
            }
        }
        const int pix = 256 / (C / 8) > 64 ? 256 / (C / 8) : 64;
        hipLaunchKernelGGL(gated_sum_part_kernel, dim3((HW + pix - 1) / pix, N), dim3(256), shmem, s, a, HW,
                           C, hid, tiles, pix, w1, b1, w2, b2, out, out_cs, out_coff);
        FM_HIP(hipGetLastError());
        return 0;
    }
    GateSumArgs a{};
    a.nstreams = nstreams;
    for (int t = 0; t < nstreams; ++t) {
        FM_CHECK_ARG(in_cs[t] % 8 == 0 && in_coff[t] % 8 == 0);
        a.in[t] = in[t]; a.in_cs[t] = in_cs[t]; a.in_coff[t] = in_coff[t];
    }
    const int groups = GS_THREADS / (C / 8);
    const size_t shmem = ((size_t)groups * C + 8 * C + 4 * hid) * sizeof(float);
    FM_CHECK_ARG(shmem <= 64 * 1024);
    hipLaunchKernelGGL(gated_sum_kernel, dim3(N), dim3(GS_THREADS), shmem, s, a, HW, C, hid, w1, b1, w2, b2, out,
                       out_cs, out_coff);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_gate_sum(int nstreams, const f16* const* in, const int* in_cs, const int* in_coff,
                    const float* const* gate, f16* out, int out_cs, int out_coff, int N, int HW, int C,
                    hipStream_t s) {
    FM_CHECK_ARG(nstreams >= 1 && nstreams <= 4 && C % 8 == 0);
    GateSumArgs a{};
    a.nstreams = nstreams;
    for (int t = 0; t < nstreams; ++t) {
        a.in[t] = in[t];
        a.in_cs[t] = in_cs[t];
        a.in_coff[t] = in_coff[t];
        a.gate[t] = gate[t];
    }
    const long total = (long)N * HW * (C / 8);
    hipLaunchKernelGGL(gate_sum_kernel, grid1d(total), dim3(256), 0, s, a, out, out_cs, out_coff, N, HW, C);
    FM_HIP(hipGetLastError());
    return 0;
}

int launch_head(const f16* in, int in_cs, int in_coff, int N, int HW, int C, int D, const f16* w,
                const float* b, float* out, float* raw_out, hipStream_t s, float* mirror) {
    FM_CHECK_ARG(C % 8 == 0 && C / 8 <= 256 && in_cs % 8 == 0 && in_coff % 8 == 0);
    const int groups = 256 / (C / 8);
    const size_t shmem = sizeof(float) * ((size_t)groups * C + C + D + 256);
    hipLaunchKernelGGL(head_kernel, dim3(N), dim3(256), shmem, s, in, in_cs, in_coff, HW, C, D, w, b, out,
                       raw_out, mirror);
    FM_HIP(hipGetLastError());
    return 0;
}

// KLT front end on the device: gray / resize / Gaussian pyramid / Scharr / pyramidal LK /
// min-eigenvalue corners (GFTT) / FAST-9, and the foreground-mask bookkeeping of Flow.predict.
//
// Replaces the OpenCV calls and Numba helpers of fastmot/flow.py:121-213:
//   cv2.cvtColor(BGR2GRAY) :129,153   cv2.resize :130,154,187-189   cv2.goodFeaturesToTrack :171-173
//   FastFeatureDetector.detect :190   cv2.calcOpticalFlowPyrLK :205-207
//   mask_area (utils/numba.py:33-39)  _rect_filter :282-295  _ellipse_filter :297-306
// OpenCV is not under /root/reference and not installable: the kernels restate the published
// OpenCV algorithms (imgproc/color, resize, pyramids; video/lkpyramid.cpp; imgproc/featureselect.cpp,
// corner.cpp; features2d/fast.cpp) and are checked against the numpy restatement in
// oracle/cv_oracle.py -- parity of this stage is UNPINNED by the reference.
//
// The reference's foreground mask is a full-frame u8 image zeroed rectangle by rectangle in
// closest-first track order; here "mask(p) at track k" is evaluated analytically as
// "p is inside no rectangle j < k", which makes every per-track quantity independent and lets all
// tracks run in one launch (no 2 MB mask fills, no per-track sequencing).
// Roofline: all kernels are HBM/L2 bound pixel passes (gray: 6.2 MB in, 2.1 MB out; pyramid
// 0.69 MB/img; LK <= 3 KB gathered per point and level, L2 resident).
#include "common.h"
#include <cstdlib>
#include <cmath>
#include <chrono>

namespace {
constexpr int MAX_LEVELS = 8;
}

// wall-clock split of the two synchronising KLT calls (ms, accumulated; printed by fm_flow_timing when
// FASTMOT_FLOW_TIMING_VERBOSE is set): prepare {host, first sync, enqueue, last sync}, lk {first sync, enqueue, last sync}
double g_flow_sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static inline double fm_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct FlowState {
    fm_flow_cfg cfg{};
    int W = 0, H = 0;
    int levels = 0;                       // pyramid levels actually built (maxLevel + 1)
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    uint8_t* gray[2] = {nullptr, nullptr};          // full resolution
    uint8_t* pyr[2][MAX_LEVELS] = {};               // level 0 = optical-flow frame
    int16_t* deriv[2][MAX_LEVELS] = {};             // Scharr of each pyramid (dx,dy interleaved)
    uint8_t* bg_img = nullptr;
    int prev = 0;                                   // index of the "previous" set
    // targets of the current predict
    int nT = 0, rect_cap = 0;
    int32_t* rects = nullptr;                       // [nT][4] inclusive integer rects
    DevBuf tgt_in, tgt_out, det_in, det_out, lk_in, lk_out, bg_out;
    unsigned long long* eig = nullptr;              // GFTT scratch: local-maximum keys, 1024 slots per tile (eig_cap tiles)
    uint2* tile_stat = nullptr;                     // per tile: maximum bits, key count (eig_cap entries)
    size_t eig_cap = 0;                             // capacity in tiles
    int32_t* ov_idx = nullptr;                      // overlap lists (earlier rects intersecting rect k)
    int32_t* ov_off = nullptr;
    int ov_cap = 0;
    // what the kernels read: the owned buffers above (staged calls) or the packed upload of fm_flow_prepare
    const int32_t* v_rects = nullptr;
    const int32_t* v_ov_idx = nullptr;
    const int32_t* v_ov_off = nullptr;
    int32_t* bg_flags = nullptr;                    // FAST score / flags
    // diagnostic LK variants (ctx option "lk_variant"): counters, capture records + per-point header
    int* lk_diag = nullptr;
    int* lk_cap = nullptr;
    int* lk_cap_hdr = nullptr;
    int lk_cap_pts = 0;
};

void fm_flow_free(FlowState* f) {
    if (!f) return;
    for (int s = 0; s < 2; ++s) {
        if (f->gray[s]) (void)hipFree(f->gray[s]);
        for (int l = 0; l < MAX_LEVELS; ++l)
            if (f->pyr[s][l]) (void)hipFree(f->pyr[s][l]);
    }
    for (int st = 0; st < 2; ++st)
        for (int l = 0; l < MAX_LEVELS; ++l)
            if (f->deriv[st][l]) (void)hipFree(f->deriv[st][l]);
    for (void* p : {(void*)f->bg_img, (void*)f->rects, (void*)f->eig, (void*)f->tile_stat, (void*)f->ov_idx, (void*)f->ov_off, (void*)f->bg_flags,
                    (void*)f->lk_diag, (void*)f->lk_cap, (void*)f->lk_cap_hdr})
        if (p) (void)hipFree(p);
    for (DevBuf* b : {&f->tgt_in, &f->tgt_out, &f->det_in, &f->det_out, &f->lk_in, &f->lk_out, &f->bg_out})
        b->release();
    delete f;
}

namespace {

// FM_SGPR_CAP: scalar register budget of the KLT kernels.  A round-2 measure against the LK results that differed under
// load (their rate followed the budget); round 3 found the cause elsewhere (packed-fp32 arithmetic, see lk_wave_body
// and build.py FILE_FLAGS) -- the cap stays because the kernels were tuned with it (occupancy 8 either way).
#define FM_SGPR_CAP __attribute__((amdgpu_num_sgpr(48)))

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- cvtColor BGR2GRAY, 8 bit: OpenCV's RGB2Gray<uchar>, (B cb + G cg + R cr + (1 << (bits - 1))) >> bits with the
// 14-bit coefficients 1868 / 9617 / 4899 (up to the 4.2 era: the reference pins 4.1.1) or the 15-bit ones
// 3735 / 19235 / 9798 (later 4.x) -- fm_flow_cfg.gray_coeff_bits
struct GrayCoeffs { int cb, cg, cr, bits; };
__host__ __device__ inline GrayCoeffs gray_coeffs(int bits) {
    return bits == 15 ? GrayCoeffs{3735, 19235, 9798, 15} : GrayCoeffs{1868, 9617, 4899, 14};
}
__device__ __forceinline__ int to_gray(const uint8_t* p, GrayCoeffs c) {
    return (p[0] * c.cb + p[1] * c.cg + p[2] * c.cr + (1 << (c.bits - 1))) >> c.bits;
}

__global__ void gray_kernel(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, int n, GrayCoeffs c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gray[i] = (uint8_t)to_gray(bgr + (size_t)i * 3, c);
}

struct LinCoef { int s0, s1; int a0, a1; };
__device__ __forceinline__ LinCoef lin_coef(int d, double scale, int ssize) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    LinCoef c;
    c.s0 = s;
    c.s1 = min(s + 1, ssize - 1);
    c.a0 = __float2int_rn((1.f - f) * 2048.f);
    c.a1 = __float2int_rn(f * 2048.f);
    return c;
}

// ---- cv2.resize 8UC1: INTER_LINEAR fixed point; exact 2x decimation -> INTER_AREA (2x2 mean)
__global__ void resize_linear_kernel(const uint8_t* __restrict__ src, int sw, int sh,
                                     uint8_t* __restrict__ dst, int dw, int dh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    if (sw == 2 * dw && sh == 2 * dh) {
        const uint8_t* r0 = src + (size_t)(2 * y) * sw + 2 * x;
        const uint8_t* r1 = r0 + sw;
        dst[(size_t)y * dw + x] = (uint8_t)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
        return;
    }
    const LinCoef cx = lin_coef(x, (double)sw / dw, sw), cy = lin_coef(y, (double)sh / dh, sh);
    const uint8_t* r0 = src + (size_t)cy.s0 * sw;
    const uint8_t* r1 = src + (size_t)cy.s1 * sw;
    const int S0 = r0[cx.s0] * cx.a0 + r0[cx.s1] * cx.a1;
    const int S1 = r1[cx.s0] * cx.a0 + r1[cx.s1] * cx.a1;
    const int v = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
    dst[(size_t)y * dw + x] = (uint8_t)min(max(v, 0), 255);
}

// ---- cv::pyrDown 8UC1: separable [1 4 6 4 1]/16, BORDER_REFLECT_101, (sum + 128) >> 8
__global__ void pyrdown_kernel(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst,
                               int dw, int dh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const int wk[5] = {1, 4, 6, 4, 1};
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, sh) * sw;
        int rs = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) rs += wk[i] * row[reflect101(2 * x + i - 2, sw)];
        sum += wk[j] * rs;
    }
    dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
}

// ---- calcSharrDeriv (video/lkpyramid.cpp): int16 (dx, dy), borders mirror row/col 1 and n-2
__global__ void scharr_kernel(const uint8_t* __restrict__ src, int w, int h, int16_t* __restrict__ d) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const int y0 = y > 0 ? y - 1 : (h > 1 ? 1 : 0), y2 = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
    const uint8_t* r0 = src + (size_t)y0 * w;
    const uint8_t* r1 = src + (size_t)y * w;
    const uint8_t* r2 = src + (size_t)y2 * w;
    auto t0 = [&](int c) { return (r0[c] + r2[c]) * 3 + r1[c] * 10; };   // vertical smoothing
    auto t1 = [&](int c) { return r2[c] - r0[c]; };                       // vertical difference
    const int dx = t0(xp) - t0(xm);
    const int dy = (t1(xp) + t1(xm)) * 3 + t1(x) * 10;
    d[((size_t)y * w + x) * 2] = (int16_t)dx;
    d[((size_t)y * w + x) * 2 + 1] = (int16_t)dy;
}

// ---- fused pyramid build (4 launches instead of 13).  Same arithmetic, pixel for pixel, as the stand-alone
// kernels above (which stay for the general-scale path and the image tests).
__device__ __forceinline__ uint8_t pyrdown_px(const uint8_t* __restrict__ src, int sw, int sh, int x, int y) {
    const int wk[5] = {1, 4, 6, 4, 1};
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, sh) * sw;
        int rs = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) rs += wk[i] * row[reflect101(2 * x + i - 2, sw)];
        sum += wk[j] * rs;
    }
    return (uint8_t)((sum + 128) >> 8);
}

__device__ __forceinline__ int scharr_px(const uint8_t* __restrict__ src, int w, int h, int x, int y) {
    const int y0 = y > 0 ? y - 1 : (h > 1 ? 1 : 0), y2 = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
    const uint8_t* r0 = src + (size_t)y0 * w;
    const uint8_t* r1 = src + (size_t)y * w;
    const uint8_t* r2 = src + (size_t)y2 * w;
    auto t0 = [&](int c) { return (r0[c] + r2[c]) * 3 + r1[c] * 10; };
    auto t1 = [&](int c) { return r2[c] - r0[c]; };
    const int dx = t0(xp) - t0(xm);
    const int dy = (t1(xp) + t1(xm)) * 3 + t1(x) * 10;
    return (int)(((unsigned)dx & 0xffffu) | ((unsigned)dy << 16));   // int16 pair (dx, dy) as one 32-bit store
}

// BGR frame -> gray (full resolution) + the half-resolution optical-flow image (cv2.resize's exact-2x path =
// INTER_AREA 2x2 mean of the gray pixels): one thread per 2x2 block, the frame is read once.
__global__ __launch_bounds__(256) void gray_half_kernel(const uint8_t* __restrict__ bgr, int W, int H,
                                                        uint8_t* __restrict__ gray, uint8_t* __restrict__ half,
                                                        GrayCoeffs c) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;      // half-resolution coordinates
    const int hw = W >> 1;
    if (x >= hw) return;
    int g[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint8_t* p = bgr + ((size_t)(2 * y + j) * W + 2 * x) * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            g[j][i] = to_gray(p + 3 * i, c);
        *reinterpret_cast<uint16_t*>(gray + (size_t)(2 * y + j) * W + 2 * x) = (uint16_t)(g[j][0] | (g[j][1] << 8));
    }
    half[(size_t)y * hw + x] = (uint8_t)((g[0][0] + g[0][1] + g[1][0] + g[1][1] + 2) >> 2);
}

// one pyramid level: Scharr derivatives of `src` AND its pyrDown, both read the same image (blockIdx.y < sh:
// derivative rows, then dh rows of the next level)
__global__ __launch_bounds__(256) void pyr_level_kernel(const uint8_t* __restrict__ src, int sw, int sh,
                                                        int* __restrict__ deriv, uint8_t* __restrict__ dst, int dw,
                                                        int dh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (y < sh) {
        if (x < sw) deriv[(size_t)y * sw + x] = scharr_px(src, sw, sh, x, y);
        return;
    }
    y -= sh;
    if (x < dw && y < dh) dst[(size_t)y * dw + x] = pyrdown_px(src, sw, sh, x, y);
}

// the small tail of the pyramid (level `first` and above, <= ~33 k pixels at 1080p) in ONE workgroup: each level
// depends on the complete previous one, so separate launches cost a dependent-launch boundary per level for a
// few microseconds of work.  Global stores of one workgroup are visible to its own waves after __syncthreads().
struct PyrTail {
    uint8_t* img[MAX_LEVELS];
    int* deriv[MAX_LEVELS];
    int w[MAX_LEVELS], h[MAX_LEVELS];
    int first, levels;
};

__global__ __launch_bounds__(1024) void pyr_tail_kernel(PyrTail t) {
    const int tid = threadIdx.x;
    for (int l = t.first; l < t.levels; ++l) {
        const int w = t.w[l], h = t.h[l];
        // level l is complete here (previous launch, or the pyrDown below + barrier)
        for (int i = tid; i < w * h; i += 1024) t.deriv[l][i] = scharr_px(t.img[l], w, h, i % w, i / w);
        if (l + 1 < t.levels) {
            const int dw = t.w[l + 1], dh = t.h[l + 1];
            for (int i = tid; i < dw * dh; i += 1024) t.img[l + 1][i] = pyrdown_px(t.img[l], w, h, i % dw, i / dw);
        }
        __syncthreads();
    }
}

// ---- pyramidal Lucas-Kanade (video/lkpyramid.cpp LKTrackerInvoker), one thread per point
struct LKArgs {
    const uint8_t* I[MAX_LEVELS];
    const uint8_t* J[MAX_LEVELS];
    const int16_t* D[MAX_LEVELS];
    int w[MAX_LEVELS], h[MAX_LEVELS];
    int levels, win, max_count;
    float eps2, min_eig_thresh;
};

#define LK_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// one image byte.  -DFM_LK_DWORD_LOADS: through an ALIGNED 32-bit load, the byte shifted out in registers -- experiment of
// late round 2: what the disturbed LK kernel gets wrong looks like single window samples being off (err changes by one
// or two quanta of 1/800, positions by 1e-4 .. 0.15 px: scripts/stress_lk7.py), which pointed at the scattered sub-dword
// loads; but the disturbance is the same with dword loads (34 vs 42 differing results in 400 calls), so they are not it.
__device__ __forceinline__ int lk_px(const uint8_t* __restrict__ row, int c) {
#ifndef FM_LK_DWORD_LOADS
    return row[c];
#else
    const uintptr_t a = reinterpret_cast<uintptr_t>(row) + (uintptr_t)c;
    const uint32_t w = *reinterpret_cast<const uint32_t*>(a & ~uintptr_t(3));
    return (int)((w >> ((a & 3) * 8)) & 0xffu);
#endif
}

// float32 sum of lanes 0..N-1 in lane order, starting from 0.f like the scalar loops of lkpyramid.cpp.
// Sequential inclusive scan over the lanes: every step adds the left neighbour's running value (DPP wave_shr:1,
// lane 0 reads 0) to the lane's own term, a(s)[i] = a(s-1)[i-1] + v[i].  By induction lane i holds
// ((v0 + v1) + ...) + vi after step i and keeps it, so N-1 steps leave the scalar-order sum in lane N-1: the
// same N-1 dependent float32 adds as the scalar loop, one VALU instruction each (round 2's first version walked
// the lanes with v_readlane: 2 instructions + an SGPR round trip per term).
__device__ __forceinline__ float seq_step(float acc, float v) {
    const int left = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    return __builtin_bit_cast(float, left) + v;
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
template <int N>
__device__ __forceinline__ float seq_sum(float v, float* pfx = nullptr) {
    float acc = v;
#pragma unroll
    for (int k = 1; k < N; ++k) acc = seq_step(acc, v);
    if (pfx) *pfx = acc;
    return lane_value(acc, N - 1);
}
// two / three independent sums, their scans interleaved step by step (a DPP read of a VGPR written by the previous
// VALU instruction costs wait states; the neighbouring chain fills them).  `pfx` (diagnostic builds): the lanes' running
// sums after the scan, lane i = v0 + .. + vi.
template <int N>
__device__ __forceinline__ void seq_sum2(float v0, float v1, float& s0, float& s1, float* pfx = nullptr) {
    float a0 = v0, a1 = v1;
#pragma unroll
    for (int k = 1; k < N; ++k) { a0 = seq_step(a0, v0); a1 = seq_step(a1, v1); }
    if (pfx) { pfx[0] = a0; pfx[1] = a1; }
    s0 = lane_value(a0, N - 1);
    s1 = lane_value(a1, N - 1);
}
template <int N>
__device__ __forceinline__ void seq_sum3(float v0, float v1, float v2, float& s0, float& s1, float& s2, float* pfx = nullptr) {
    float a0 = v0, a1 = v1, a2 = v2;
#pragma unroll
    for (int k = 1; k < N; ++k) { a0 = seq_step(a0, v0); a1 = seq_step(a1, v1); a2 = seq_step(a2, v2); }
    if (pfx) { pfx[0] = a0; pfx[1] = a1; pfx[2] = a2; }
    s0 = lane_value(a0, N - 1);
    s1 = lane_value(a1, N - 1);
    s2 = lane_value(a2, N - 1);
}

// The same sums WITHOUT any cross-lane VALU operation: every lane parks its term in the wavefront's LDS slab, then
// every lane reads all N terms back (same address in all lanes: broadcast reads) and adds them in scalar order in its
// own registers -- N - 1 dependent float32 adds per sum again, the result is in every lane by construction.
// LDS operations of one wavefront execute in order; the fences keep the compiler from moving them.
template <int N>
__device__ __forceinline__ float lds_seq(const float* __restrict__ slab) {
    const float4* q = reinterpret_cast<const float4*>(slab);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
        const float4 t = q[k];
        acc += t.x; acc += t.y; acc += t.z; acc += t.w;
    }
#pragma unroll
    for (int k = N / 4 * 4; k < N; ++k) acc += slab[k];
    return acc;
}
template <int N, int K>
__device__ __forceinline__ void lds_sums(float* slab, int lane, const float (&v)[K], float (&s)[K]) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 32) {
#pragma unroll
        for (int k = 0; k < K; ++k) slab[32 * k + lane] = v[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = lds_seq<N>(slab + 32 * k);
    __builtin_amdgcn_wave_barrier();
}

// One WAVEFRONT per point: lane g < win*win owns window pixel (g / win, g % win) -- the samples of a window are taken
// in parallel, and every window sum (A11, A12, A22, b1, b2, the error) is accumulated in the sequential (y, x) order of
// LKTrackerInvoker's scalar loops (seq_sum*: 24 dependent float32 adds): bit-identical to
// oracle/cv_oracle.calc_optical_flow_pyr_lk.  (A butterfly reduction agrees with the scalar order to ~1e-4 px only --
// enough to flip an inlier decision now and then and let the keypoint sets of the two implementations drift apart over
// a clip, tests/test_e2e_parity_gpu.py.)  Everything after the sums is wave-uniform arithmetic that every lane
// repeats; the control flow (levels skipped, iteration counts) is therefore uniform and compiled as scalar branches on
// "any lane" conditions -- which is why ONE wrong lane changes the result of the point.
//
// The wrong lane (rounds 1-2: "LK results differ from call to call while the ReID network runs").  Round 3
// captured every iteration of disturbed calls (lk_diag_kernel below, scripts/lk_bisect.py, profiles/r03_lk_bisect.txt):
// samples, running sums and broadcast sums were always right; what was wrong was dx of the position update
//     dx = (A12 * b2 - A22 * b1) * Dt,   dy = (A12 * b1 - A11 * b2) * Dt
// in lanes 48..63 only.  The SLP vectoriser had packed the two expressions into v_pk_mul_f32 / v_pk_add_f32, the cross
// terms through `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (low result from the HIGH register of a source pair);
// on MI355X that instruction form returns a wrong low half in the last 16 lanes of the wavefront now and then while
// wavefronts of the fused LightConv kernels (or of two streamed-conv variants) are resident on the same CU -- never
// when idle, never with unpacked v_mul_f32 / v_sub_f32 (csrc/diag.hip reproduces it stand alone: 0 of 72 M evaluations
// idle, ~25 k wrong lanes beside litechain_kernel, all in lanes 48..63, all in the low half).  A wrong dx in lanes
// 48..63 then ends (or prolongs) the iteration through the any-lane branch: positions off by 1e-4 .. 6 px.
// Consequence: this file is compiled with -fno-slp-vectorize (build.py FILE_FLAGS: no packed fp32 in the KLT kernels; 0
// disagreeing lanes in 5.1 M iterations under the same load), tests/test_isa_lint.py refuses the instruction form in
// every kernel of the library, and the launch needs neither whole CUs nor an ordering against the ReID network.
template <int WINC>
__device__ __forceinline__ void lk_wave_body(const LKArgs& a, int n, const float* __restrict__ prev_pts,
                                                 float* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                 float* __restrict__ err) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int pt = gidx >> 6, g = gidx & 63;
    if (pt >= n) return;                       // the whole wavefront exits together
    constexpr int win = WINC;
    const bool lane_on = g < win * win;
    const int wy = lane_on ? g / win : 0, wx = lane_on ? g % win : 0;
    const float half = (win - 1) * 0.5f;
    const float px0 = prev_pts[2 * pt], py0 = prev_pts[2 * pt + 1];
    float nx = 0.f, ny = 0.f;
    bool st = true;
    float er = 0.f;
    const float FLT_SCALE = 1.f / (1 << 20);
    auto weights = [](float fa, float fb, int& iw00, int& iw01, int& iw10, int& iw11) {
        iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << 14));
        iw01 = __float2int_rn(fa * (1.f - fb) * (1 << 14));
        iw10 = __float2int_rn((1.f - fa) * fb * (1 << 14));
        iw11 = (1 << 14) - iw00 - iw01 - iw10;
    };
    for (int level = a.levels - 1; level >= 0; --level) {
        const int w = a.w[level], h = a.h[level];
        const uint8_t* I = a.I[level];
        const uint8_t* J = a.J[level];
        const int16_t* D = a.D[level];
        const float sc = 1.f / (float)(1 << level);
        float ppx = px0 * sc, ppy = py0 * sc;
        if (level == a.levels - 1) { nx = ppx; ny = ppy; }
        else { nx *= 2.f; ny *= 2.f; }
        ppx -= half; ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
            if (level == 0) { st = false; er = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        weights(ppx - ipx, ppy - ipy, iw00, iw01, iw10, iw11);
        int ival = 0, ixval = 0, iyval = 0;
        if (lane_on) {
            const int xx0 = ipx + wx, xx1 = xx0 + 1, yy0 = ipy + wy, yy1 = yy0 + 1;
            const uint8_t* r0 = I + (size_t)reflect101(yy0, h) * w;
            const uint8_t* r1 = I + (size_t)reflect101(yy1, h) * w;
            const int c0 = reflect101(xx0, w), c1 = reflect101(xx1, w);
            ival = LK_DESCALE(lk_px(r0, c0) * iw00 + lk_px(r0, c1) * iw01 + lk_px(r1, c0) * iw10 + lk_px(r1, c1) * iw11, 14 - 5);
            // derivative image is zero outside (BORDER_CONSTANT), lkpyramid.cpp
            auto dv = [&](int xx, int yy) -> int2 {
                if (xx < 0 || xx >= w || yy < 0 || yy >= h) return make_int2(0, 0);
                const int v = *reinterpret_cast<const int*>(D + ((size_t)yy * w + xx) * 2);
                return make_int2((int)(short)(v & 0xffff), (int)(short)(v >> 16));
            };
            const int2 d00 = dv(xx0, yy0), d01 = dv(xx1, yy0), d10 = dv(xx0, yy1), d11 = dv(xx1, yy1);
            ixval = LK_DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
            iyval = LK_DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
        }
        float A11, A12, A22;
        seq_sum3<WINC * WINC>((float)(ixval * ixval), (float)(ixval * iyval), (float)(iyval * iyval), A11, A12, A22);
        A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
        float Dt = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < a.min_eig_thresh || Dt < 1.1920929e-07f) {
            if (level == 0) st = false;
            continue;
        }
        Dt = 1.f / Dt;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        float outx = nx + half, outy = ny + half;
        bool running = true;
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (running && (inx < -win || inx >= w || iny < -win || iny >= h)) {
                if (level == 0) st = false;
                running = false;
            }
            if (!running) break;                 // uniform within the 32-lane group
            weights(nx - inx, ny - iny, iw00, iw01, iw10, iw11);
            int diff = 0;
            if (lane_on) {
                const uint8_t* r0 = J + (size_t)reflect101(iny + wy, h) * w;
                const uint8_t* r1 = J + (size_t)reflect101(iny + wy + 1, h) * w;
                const int c0 = reflect101(inx + wx, w), c1 = reflect101(inx + wx + 1, w);
                diff = LK_DESCALE(lk_px(r0, c0) * iw00 + lk_px(r0, c1) * iw01 + lk_px(r1, c0) * iw10 + lk_px(r1, c1) * iw11, 14 - 5) - ival;
            }
            float b1, b2;
            seq_sum2<WINC * WINC>((float)(diff * ixval), (float)(diff * iyval), b1, b2);
            b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * Dt, dy = (A12 * b1 - A11 * b2) * Dt;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            if (dx * dx + dy * dy <= a.eps2) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        nx = outx; ny = outy;
        if (st && level == 0) {
            const float ex = nx - half, ey = ny - half;
            const int inx = (int)floorf(ex), iny = (int)floorf(ey);
            if (inx < -win || inx >= w || iny < -win || iny >= h) { st = false; continue; }
            weights(ex - inx, ey - iny, iw00, iw01, iw10, iw11);
            int diff = 0;
            if (lane_on) {
                const uint8_t* r0 = J + (size_t)reflect101(iny + wy, h) * w;
                const uint8_t* r1 = J + (size_t)reflect101(iny + wy + 1, h) * w;
                const int c0 = reflect101(inx + wx, w), c1 = reflect101(inx + wx + 1, w);
                diff = LK_DESCALE(lk_px(r0, c0) * iw00 + lk_px(r0, c1) * iw01 + lk_px(r1, c0) * iw10 + lk_px(r1, c1) * iw11, 14 - 5) - ival;
            }
            er = seq_sum<WINC * WINC>(fabsf((float)diff)) * 1.f / (32 * win * win);
        }
    }
    if (g == 0) {
        next_pts[2 * pt] = nx;
        next_pts[2 * pt + 1] = ny;
        status[pt] = st ? 1 : 0;
        err[pt] = er;
    }
}


template <int WINC>
__global__ __launch_bounds__(1024) FM_SGPR_CAP void lk_wave_kernel(LKArgs a, int n, const float* __restrict__ prev_pts,
                                                      float* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                      float* __restrict__ err) {
    lk_wave_body<WINC>(a, n, prev_pts, next_pts, status, err);
}

// ---- two points per wavefront.  Lanes 0..24 carry the window of point 2 w, lanes 32..56 the window of point 2 w + 1;
// the seven lanes behind each window LEAVE the kernel at once.  That is what lets the DPP scans of the two windows run
// in the same instructions: a lane whose left neighbour is masked off reads 0 (bound_ctrl), exactly like lane 0 whose
// neighbour does not exist, so the scan restarts at lane 32 by itself.  The broadcast of a sum comes from lane 24 of
// the own half (ds_bpermute instead of v_readlane), the arithmetic behind it is per lane as before, and the control
// flow -- levels skipped, iteration counts -- is now divergent between the halves (the compiler masks; nothing is
// wave-uniform any more except the level loop).  Same operations in the same order per point: bit-identical to the
// one-point kernel (tests/test_flow_gpu.py).  Half the wavefronts hold the registers half as long per point: the
// kernel's footprint beside the other streams' kernels is what the pipeline pays for.
__device__ __forceinline__ float half_value(float v, int lane_in_half, int half_base) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((half_base + lane_in_half) << 2, __builtin_bit_cast(int, v)));
}

// PATCH (round 5): the iteration's samples of the NEW frame come from a 16 x 16 byte patch of the level
// in LDS, filled once per (point, level) around the start position (+-5 px of travel inside a level; a window that
// leaves the patch falls back to the global loads).  An iteration then waits for four ds_read_u8 (~100 cycles) instead
// of four global byte loads (~500-700 cycles beside the networks); the values, and with them every result, are the same.
// The patch is private to a half-wavefront: a wavefront's LDS operations execute in program order, no barrier is needed
// (and none is possible: lanes leave early and the two halves diverge).
constexpr int LK_PW = 16, LK_PR = 5;

template <int WINC, bool PATCH>
__global__ __launch_bounds__(256) FM_SGPR_CAP void lk_pair_kernel(LKArgs a, int n, const float* __restrict__ prev_pts,
                                                                  float* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                                  float* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) uint8_t patch_all[PATCH ? 8 * LK_PW * LK_PW : 16];
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = gidx & 63, g = l & 31, hb = l & 32;
    const int pt = (gidx >> 6) * 2 + (l >> 5);
    constexpr int win = WINC, NW = WINC * WINC;
    if (pt >= n || g >= NW) return;
    uint8_t* const patch = patch_all + (PATCH ? ((threadIdx.x >> 5) & 7) * (LK_PW * LK_PW) : 0);
    int sx0 = 0, sy0 = 0;                       // top-left corner of the patch in level coordinates
    bool have_patch = false;
    const int wy = g / win, wx = g % win;
    const float half = (win - 1) * 0.5f;
    const float px0 = prev_pts[2 * pt], py0 = prev_pts[2 * pt + 1];
    float nx = 0.f, ny = 0.f;
    bool st = true;
    float er = 0.f;
    const float FLT_SCALE = 1.f / (1 << 20);
    auto weights = [](float fa, float fb, int& iw00, int& iw01, int& iw10, int& iw11) {
        iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << 14));
        iw01 = __float2int_rn(fa * (1.f - fb) * (1 << 14));
        iw10 = __float2int_rn((1.f - fa) * fb * (1 << 14));
        iw11 = (1 << 14) - iw00 - iw01 - iw10;
    };
    auto sample = [&](const uint8_t* img, int w, int h, int bx, int by, int iw00, int iw01, int iw10, int iw11) -> int {
        const uint8_t* r0 = img + (size_t)reflect101(by + wy, h) * w;
        const uint8_t* r1 = img + (size_t)reflect101(by + wy + 1, h) * w;
        const int c0 = reflect101(bx + wx, w), c1 = reflect101(bx + wx + 1, w);
        return LK_DESCALE(lk_px(r0, c0) * iw00 + lk_px(r0, c1) * iw01 + lk_px(r1, c0) * iw10 + lk_px(r1, c1) * iw11, 14 - 5);
    };
    // the same sample of the new frame J, from the patch when the (win + 1)^2 pixels of the window lie inside it
    auto sample_j = [&](const uint8_t* img, int w, int h, int bx, int by, int iw00, int iw01, int iw10, int iw11) -> int {
        if (PATCH && have_patch && bx >= sx0 && by >= sy0 && bx + win + 1 <= sx0 + LK_PW && by + win + 1 <= sy0 + LK_PW) {
            const uint8_t* q = patch + (by - sy0 + wy) * LK_PW + (bx - sx0 + wx);
            return LK_DESCALE((int)q[0] * iw00 + (int)q[1] * iw01 + (int)q[LK_PW] * iw10 + (int)q[LK_PW + 1] * iw11, 14 - 5);
        }
        return sample(img, w, h, bx, by, iw00, iw01, iw10, iw11);
    };
    // patch[r][c] = J[reflect101(y0 + r)][reflect101(x0 + c)]: 64 dwords, three per lane.  Inside the image a dword is two
    // ALIGNED loads and a byte alignment; at the borders four byte loads with the reflection sample() applies.
    auto fill_patch = [&](const uint8_t* img, int w, int h, int x0, int y0) {
        sx0 = x0; sy0 = y0;
        const bool inside = x0 >= 0 && x0 + LK_PW + 4 <= w;
        for (int item = g; item < LK_PW * LK_PW / 4; item += NW) {
            const int r = item >> 2, c = (item & 3) * 4;
            const uint8_t* row = img + (size_t)reflect101(y0 + r, h) * w;
            uint32_t v;
            if (inside) {
                const uintptr_t ad = reinterpret_cast<uintptr_t>(row) + (uintptr_t)(x0 + c);
                const uint32_t* al = reinterpret_cast<const uint32_t*>(ad & ~uintptr_t(3));
                v = __builtin_amdgcn_alignbyte(al[1], al[0], (uint32_t)(ad & 3));
            } else {
                v = (uint32_t)row[reflect101(x0 + c, w)] | (uint32_t)row[reflect101(x0 + c + 1, w)] << 8 |
                    (uint32_t)row[reflect101(x0 + c + 2, w)] << 16 | (uint32_t)row[reflect101(x0 + c + 3, w)] << 24;
            }
            *reinterpret_cast<uint32_t*>(patch + r * LK_PW + c) = v;
        }
        have_patch = true;
    };
    for (int level = a.levels - 1; level >= 0; --level) {
        const int w = a.w[level], h = a.h[level];
        const uint8_t* I = a.I[level];
        const uint8_t* J = a.J[level];
        const int16_t* D = a.D[level];
        have_patch = false;
        const float sc = 1.f / (float)(1 << level);
        float ppx = px0 * sc, ppy = py0 * sc;
        if (level == a.levels - 1) { nx = ppx; ny = ppy; }
        else { nx *= 2.f; ny *= 2.f; }
        ppx -= half; ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
            if (level == 0) { st = false; er = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        weights(ppx - ipx, ppy - ipy, iw00, iw01, iw10, iw11);
        const int ival = sample(I, w, h, ipx, ipy, iw00, iw01, iw10, iw11);
        int ixval, iyval;
        {
            const int xx0 = ipx + wx, xx1 = xx0 + 1, yy0 = ipy + wy, yy1 = yy0 + 1;
            // derivative image is zero outside (BORDER_CONSTANT), lkpyramid.cpp
            auto dv = [&](int xx, int yy) -> int2 {
                if (xx < 0 || xx >= w || yy < 0 || yy >= h) return make_int2(0, 0);
                const int v = *reinterpret_cast<const int*>(D + ((size_t)yy * w + xx) * 2);
                return make_int2((int)(short)(v & 0xffff), (int)(short)(v >> 16));
            };
            const int2 d00 = dv(xx0, yy0), d01 = dv(xx1, yy0), d10 = dv(xx0, yy1), d11 = dv(xx1, yy1);
            ixval = LK_DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
            iyval = LK_DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
        }
        float A11, A12, A22;
        {
            const float v0 = (float)(ixval * ixval), v1 = (float)(ixval * iyval), v2 = (float)(iyval * iyval);
            float a0 = v0, a1 = v1, a2 = v2;
#pragma unroll
            for (int k = 1; k < NW; ++k) { a0 = seq_step(a0, v0); a1 = seq_step(a1, v1); a2 = seq_step(a2, v2); }
            A11 = half_value(a0, NW - 1, hb); A12 = half_value(a1, NW - 1, hb); A22 = half_value(a2, NW - 1, hb);
        }
        A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
        float Dt = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < a.min_eig_thresh || Dt < 1.1920929e-07f) {
            if (level == 0) st = false;
            continue;
        }
        Dt = 1.f / Dt;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        float outx = nx + half, outy = ny + half;
        if (PATCH) {
            const int fx = (int)floorf(nx), fy = (int)floorf(ny);
            if (fx >= -win && fx < w && fy >= -win && fy < h) fill_patch(J, w, h, fx - LK_PR, fy - LK_PR);
        }
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= w || iny < -win || iny >= h) {
                if (level == 0) st = false;
                break;
            }
            weights(nx - inx, ny - iny, iw00, iw01, iw10, iw11);
            const int diff = sample_j(J, w, h, inx, iny, iw00, iw01, iw10, iw11) - ival;
            float b1, b2;
            {
                const float v0 = (float)(diff * ixval), v1 = (float)(diff * iyval);
                float a0 = v0, a1 = v1;
#pragma unroll
                for (int k = 1; k < NW; ++k) { a0 = seq_step(a0, v0); a1 = seq_step(a1, v1); }
                b1 = half_value(a0, NW - 1, hb); b2 = half_value(a1, NW - 1, hb);
            }
            b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * Dt, dy = (A12 * b1 - A11 * b2) * Dt;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            if (dx * dx + dy * dy <= a.eps2) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        nx = outx; ny = outy;
        if (st && level == 0) {
            const float ex = nx - half, ey = ny - half;
            const int inx = (int)floorf(ex), iny = (int)floorf(ey);
            if (inx < -win || inx >= w || iny < -win || iny >= h) { st = false; continue; }
            weights(ex - inx, ey - iny, iw00, iw01, iw10, iw11);
            const int diff = sample_j(J, w, h, inx, iny, iw00, iw01, iw10, iw11) - ival;
            const float v0 = fabsf((float)diff);
            float a0 = v0;
#pragma unroll
            for (int k = 1; k < NW; ++k) a0 = seq_step(a0, v0);
            er = half_value(a0, NW - 1, hb) * 1.f / (32 * win * win);
        }
    }
    if (g == 0) {
        next_pts[2 * pt] = nx;
        next_pts[2 * pt + 1] = ny;
        status[pt] = st ? 1 : 0;
        err[pt] = er;
    }
}

// ---- round 6: the same two-points-per-wavefront kernel with everything that does NOT depend on the tracked position taken
// out of the level loop.  In calcOpticalFlowPyrLK a level's template terms -- the window's samples of the previous frame
// I, its Scharr derivatives, the matrix sums A11 / A12 / A22 -- are functions of the ORIGINAL point and the level alone;
// only the iterations follow the position found on the coarser level.  lk_pair_kernel fetched and summed them level by
// level: per level two dependent trips to memory (~500-700 cycles each beside the networks) and three 24-step serial
// scans in front of the first iteration, six times in a row.  Here all levels' loads are requested at once and the 3 x
// levels scans run interleaved (18 independent chains: the DPP adds issue back to back instead of waiting for each other);
// the level loop keeps the patch fill and the iterations.  Every value is computed by the same operations in the same
// order as before: bit-identical (tests/test_flow_gpu.py, the e2e parity tests).  Up to LK_PRE levels (the reference's
// maxLevel = 5 gives 6); deeper pyramids take lk_pair_kernel.
constexpr int LK_PRE = 6;

template <int WINC>
__global__ __launch_bounds__(256) FM_SGPR_CAP void lk_pair_pre_kernel(LKArgs a, int n, const float* __restrict__ prev_pts,
                                                                      float* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                                      float* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) uint8_t patch_all[8 * LK_PW * LK_PW];
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = gidx & 63, g = l & 31, hb = l & 32;
    const int pt = (gidx >> 6) * 2 + (l >> 5);
    constexpr int win = WINC, NW = WINC * WINC;
    if (pt >= n || g >= NW) return;
    uint8_t* const patch = patch_all + ((threadIdx.x >> 5) & 7) * (LK_PW * LK_PW);
    int sx0 = 0, sy0 = 0;
    bool have_patch = false;
    const int wy = g / win, wx = g % win;
    const float half = (win - 1) * 0.5f;
    const float px0 = prev_pts[2 * pt], py0 = prev_pts[2 * pt + 1];
    float nx = 0.f, ny = 0.f;
    bool st = true;
    float er = 0.f;
    const float FLT_SCALE = 1.f / (1 << 20);
    auto weights = [](float fa, float fb, int& iw00, int& iw01, int& iw10, int& iw11) {
        iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << 14));
        iw01 = __float2int_rn(fa * (1.f - fb) * (1 << 14));
        iw10 = __float2int_rn((1.f - fa) * fb * (1 << 14));
        iw11 = (1 << 14) - iw00 - iw01 - iw10;
    };
    auto sample = [&](const uint8_t* img, int w, int h, int bx, int by, int iw00, int iw01, int iw10, int iw11) -> int {
        const uint8_t* r0 = img + (size_t)reflect101(by + wy, h) * w;
        const uint8_t* r1 = img + (size_t)reflect101(by + wy + 1, h) * w;
        const int c0 = reflect101(bx + wx, w), c1 = reflect101(bx + wx + 1, w);
        return LK_DESCALE(lk_px(r0, c0) * iw00 + lk_px(r0, c1) * iw01 + lk_px(r1, c0) * iw10 + lk_px(r1, c1) * iw11, 14 - 5);
    };
    auto sample_j = [&](const uint8_t* img, int w, int h, int bx, int by, int iw00, int iw01, int iw10, int iw11) -> int {
        if (have_patch && bx >= sx0 && by >= sy0 && bx + win + 1 <= sx0 + LK_PW && by + win + 1 <= sy0 + LK_PW) {
            const uint8_t* q = patch + (by - sy0 + wy) * LK_PW + (bx - sx0 + wx);
            return LK_DESCALE((int)q[0] * iw00 + (int)q[1] * iw01 + (int)q[LK_PW] * iw10 + (int)q[LK_PW + 1] * iw11, 14 - 5);
        }
        return sample(img, w, h, bx, by, iw00, iw01, iw10, iw11);
    };
    auto fill_patch = [&](const uint8_t* img, int w, int h, int x0, int y0) {
        sx0 = x0; sy0 = y0;
        const bool inside = x0 >= 0 && x0 + LK_PW + 4 <= w;
        for (int item = g; item < LK_PW * LK_PW / 4; item += NW) {
            const int r = item >> 2, c = (item & 3) * 4;
            const uint8_t* row = img + (size_t)reflect101(y0 + r, h) * w;
            uint32_t v;
            if (inside) {
                const uintptr_t ad = reinterpret_cast<uintptr_t>(row) + (uintptr_t)(x0 + c);
                const uint32_t* al = reinterpret_cast<const uint32_t*>(ad & ~uintptr_t(3));
                v = __builtin_amdgcn_alignbyte(al[1], al[0], (uint32_t)(ad & 3));
            } else {
                v = (uint32_t)row[reflect101(x0 + c, w)] | (uint32_t)row[reflect101(x0 + c + 1, w)] << 8 |
                    (uint32_t)row[reflect101(x0 + c + 2, w)] << 16 | (uint32_t)row[reflect101(x0 + c + 3, w)] << 24;
            }
            *reinterpret_cast<uint32_t*>(patch + r * LK_PW + c) = v;
        }
        have_patch = true;
    };

    // ---- the template terms of every level
    int ivalL[LK_PRE], ixL[LK_PRE], iyL[LK_PRE];
    bool posok[LK_PRE];
#pragma unroll
    for (int lv = 0; lv < LK_PRE; ++lv) {
        ivalL[lv] = ixL[lv] = iyL[lv] = 0;
        posok[lv] = false;
        if (lv < a.levels) {
            const int w = a.w[lv], h = a.h[lv];
            const float sc = 1.f / (float)(1 << lv);
            const float ppx = px0 * sc - half, ppy = py0 * sc - half;
            const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
            posok[lv] = !(ipx < -win || ipx >= w || ipy < -win || ipy >= h);
            // (a position outside the level is never used: its loads go to a clamped one)
            const int cx = min(max(ipx, -win), w - 1), cy = min(max(ipy, -win), h - 1);
            int iw00, iw01, iw10, iw11;
            weights(ppx - ipx, ppy - ipy, iw00, iw01, iw10, iw11);
            ivalL[lv] = sample(a.I[lv], w, h, cx, cy, iw00, iw01, iw10, iw11);
            const int16_t* D = a.D[lv];
            const int xx0 = cx + wx, xx1 = xx0 + 1, yy0 = cy + wy, yy1 = yy0 + 1;
            auto dv = [&](int xx, int yy) -> int2 {      // derivative image is zero outside (BORDER_CONSTANT), lkpyramid.cpp
                if (xx < 0 || xx >= w || yy < 0 || yy >= h) return make_int2(0, 0);
                const int v = *reinterpret_cast<const int*>(D + ((size_t)yy * w + xx) * 2);
                return make_int2((int)(short)(v & 0xffff), (int)(short)(v >> 16));
            };
            const int2 d00 = dv(xx0, yy0), d01 = dv(xx1, yy0), d10 = dv(xx0, yy1), d11 = dv(xx1, yy1);
            ixL[lv] = LK_DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
            iyL[lv] = LK_DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
        }
    }
    float A11L[LK_PRE], A12L[LK_PRE], A22L[LK_PRE];
    {
        float v0[LK_PRE], v1[LK_PRE], v2[LK_PRE], s0[LK_PRE], s1[LK_PRE], s2[LK_PRE];
#pragma unroll
        for (int lv = 0; lv < LK_PRE; ++lv) {
            s0[lv] = v0[lv] = (float)(ixL[lv] * ixL[lv]);
            s1[lv] = v1[lv] = (float)(ixL[lv] * iyL[lv]);
            s2[lv] = v2[lv] = (float)(iyL[lv] * iyL[lv]);
        }
#pragma unroll
        for (int k = 1; k < NW; ++k)
#pragma unroll
            for (int lv = 0; lv < LK_PRE; ++lv) {
                s0[lv] = seq_step(s0[lv], v0[lv]); s1[lv] = seq_step(s1[lv], v1[lv]); s2[lv] = seq_step(s2[lv], v2[lv]);
            }
#pragma unroll
        for (int lv = 0; lv < LK_PRE; ++lv) {
            A11L[lv] = half_value(s0[lv], NW - 1, hb); A12L[lv] = half_value(s1[lv], NW - 1, hb); A22L[lv] = half_value(s2[lv], NW - 1, hb);
        }
    }

    // ---- coarse to fine: what follows the position
#pragma unroll
    for (int lv = LK_PRE - 1; lv >= 0; --lv) {
        if (lv >= a.levels) continue;
        const int level = lv;
        const int w = a.w[lv], h = a.h[lv];
        const uint8_t* J = a.J[lv];
        have_patch = false;
        const float sc = 1.f / (float)(1 << level);
        if (level == a.levels - 1) { nx = px0 * sc; ny = py0 * sc; }
        else { nx *= 2.f; ny *= 2.f; }
        if (!posok[lv]) {
            if (level == 0) { st = false; er = 0.f; }
            continue;
        }
        const int ival = ivalL[lv], ixval = ixL[lv], iyval = iyL[lv];
        float A11 = A11L[lv] * FLT_SCALE, A12 = A12L[lv] * FLT_SCALE, A22 = A22L[lv] * FLT_SCALE;
        float Dt = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < a.min_eig_thresh || Dt < 1.1920929e-07f) {
            if (level == 0) st = false;
            continue;
        }
        Dt = 1.f / Dt;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        float outx = nx + half, outy = ny + half;
        {
            const int fx = (int)floorf(nx), fy = (int)floorf(ny);
            if (fx >= -win && fx < w && fy >= -win && fy < h) fill_patch(J, w, h, fx - LK_PR, fy - LK_PR);
        }
        int iw00, iw01, iw10, iw11;
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= w || iny < -win || iny >= h) {
                if (level == 0) st = false;
                break;
            }
            weights(nx - inx, ny - iny, iw00, iw01, iw10, iw11);
            const int diff = sample_j(J, w, h, inx, iny, iw00, iw01, iw10, iw11) - ival;
            float b1, b2;
            {
                const float v0 = (float)(diff * ixval), v1 = (float)(diff * iyval);
                float a0 = v0, a1 = v1;
#pragma unroll
                for (int k = 1; k < NW; ++k) { a0 = seq_step(a0, v0); a1 = seq_step(a1, v1); }
                b1 = half_value(a0, NW - 1, hb); b2 = half_value(a1, NW - 1, hb);
            }
            b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * Dt, dy = (A12 * b1 - A11 * b2) * Dt;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            if (dx * dx + dy * dy <= a.eps2) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        nx = outx; ny = outy;
        if (st && level == 0) {
            const float ex = nx - half, ey = ny - half;
            const int inx = (int)floorf(ex), iny = (int)floorf(ey);
            if (inx < -win || inx >= w || iny < -win || iny >= h) { st = false; continue; }
            weights(ex - inx, ey - iny, iw00, iw01, iw10, iw11);
            const int diff = sample_j(J, w, h, inx, iny, iw00, iw01, iw10, iw11) - ival;
            const float v0 = fabsf((float)diff);
            float a0 = v0;
#pragma unroll
            for (int k = 1; k < NW; ++k) a0 = seq_step(a0, v0);
            er = half_value(a0, NW - 1, hb) * 1.f / (32 * win * win);
        }
    }
    if (g == 0) {
        next_pts[2 * pt] = nx;
        next_pts[2 * pt + 1] = ny;
        status[pt] = st ? 1 : 0;
        err[pt] = er;
    }
}

#ifdef FM_DIAG      // diagnostic builds only (FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_DIAG, include/fastmot_hip_diag.h)
// ---- diagnostic variants of the LK kernel (round 3: bisect of the results that differ under load).
// MODE 0: window sums as DPP scans (the production arithmetic), 1: through LDS (no cross-lane VALU operation),
// 2: both, compared, re-evaluated on a mismatch (counters say which of the two changed its mind).
// CHK: every image sample is loaded twice (second time through a laundered pointer the compiler cannot merge) and
//      the wave-uniform position update is compared across the lanes.
// CAP: every (level, iteration) appends a record of 8 rows x 64 lanes to `cap`: the lanes' samples, the lanes'
//      running sums after the scan, the broadcast sums, every lane's copy of the position; header = HW_ID / XCC_ID.
// diag[]: 0 sum mismatches (MODE 2), 1 DPP value changed on re-evaluation, 2 LDS value changed, 3 still different
//      after 3 retries, 4 duplicate-load mismatches, 5 lanes disagree on the position, 6 iterations evaluated.
constexpr int LK_CAP_ROWS = 12, LK_CAP_REC = LK_CAP_ROWS * 64, LK_CAP_MAXREC = 80;

__device__ __forceinline__ const uint8_t* launder(const uint8_t* p) {
    asm volatile("" : "+v"(p));
    return p;
}

template <int WINC, int MODE, bool CHK, bool CAP>
__global__ __launch_bounds__(1024) FM_SGPR_CAP void lk_diag_kernel(LKArgs a, int n, const float* __restrict__ prev_pts,
                                                                   float* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                                   float* __restrict__ err, int* __restrict__ diag,
                                                                   int* __restrict__ cap, int* __restrict__ cap_hdr) {
    __shared__ float slab_all[MODE == 0 ? 1 : 16 * 96];
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int pt = gidx >> 6, g = gidx & 63;
    if (pt >= n) return;
    float* slab = slab_all + (MODE == 0 ? 0 : (threadIdx.x >> 6) * 96);
    constexpr int win = WINC;
    constexpr int NW = WINC * WINC;
    const bool lane_on = g < NW;
    const int wy = lane_on ? g / win : 0, wx = lane_on ? g % win : 0;
    const float half = (win - 1) * 0.5f;
    const float px0 = prev_pts[2 * pt], py0 = prev_pts[2 * pt + 1];
    float nx = 0.f, ny = 0.f;
    bool st = true;
    float er = 0.f;
    const float FLT_SCALE = 1.f / (1 << 20);
    int nrec = 0;
    int* rec0 = CAP ? cap + (size_t)pt * LK_CAP_MAXREC * LK_CAP_REC : nullptr;
    auto put = [&](int row, int v) { if (CAP && nrec < LK_CAP_MAXREC) rec0[(size_t)nrec * LK_CAP_REC + row * 64 + g] = v; };
    auto putf = [&](int row, float v) { put(row, __builtin_bit_cast(int, v)); };
    auto weights = [](float fa, float fb, int& iw00, int& iw01, int& iw10, int& iw11) {
        iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << 14));
        iw01 = __float2int_rn(fa * (1.f - fb) * (1 << 14));
        iw10 = __float2int_rn((1.f - fa) * fb * (1 << 14));
        iw11 = (1 << 14) - iw00 - iw01 - iw10;
    };
    // bilinear sample of image `img` at window pixel (wy, wx) of the window whose corner is (bx, by)
    auto sample = [&](const uint8_t* img, int w, int h, int bx, int by, int iw00, int iw01, int iw10, int iw11) -> int {
        const uint8_t* r0 = img + (size_t)reflect101(by + wy, h) * w;
        const uint8_t* r1 = img + (size_t)reflect101(by + wy + 1, h) * w;
        const int c0 = reflect101(bx + wx, w), c1 = reflect101(bx + wx + 1, w);
        const int v = LK_DESCALE(lk_px(r0, c0) * iw00 + lk_px(r0, c1) * iw01 + lk_px(r1, c0) * iw10 + lk_px(r1, c1) * iw11, 14 - 5);
        if (CHK) {
            const uint8_t* q0 = launder(r0);
            const uint8_t* q1 = launder(r1);
            const int v2 = LK_DESCALE(lk_px(q0, c0) * iw00 + lk_px(q0, c1) * iw01 + lk_px(q1, c0) * iw10 + lk_px(q1, c1) * iw11, 14 - 5);
            if (v2 != v) atomicAdd(&diag[4], 1);
        }
        return v;
    };
    // K window sums in scalar order; pfx = the lanes' running sums of the DPP scan (capture)
    auto sums2 = [&](float v0, float v1, float& s0, float& s1, float* pfx) {
        if (MODE == 0) { seq_sum2<NW>(v0, v1, s0, s1, pfx); return; }
        const float v[2] = {v0, v1};
        float l[2];
        lds_sums<NW, 2>(slab, g, v, l);
        if (MODE == 1) { s0 = l[0]; s1 = l[1]; return; }
        float d0, d1;
        seq_sum2<NW>(v0, v1, d0, d1, pfx);
        int tries = 0;
        while ((__builtin_bit_cast(int, d0) != __builtin_bit_cast(int, l[0]) || __builtin_bit_cast(int, d1) != __builtin_bit_cast(int, l[1])) && tries < 3) {
            if (g == 0 && tries == 0) atomicAdd(&diag[0], 1);
            float e0, e1, m[2];
            seq_sum2<NW>(v0, v1, e0, e1, nullptr);
            lds_sums<NW, 2>(slab, g, v, m);
            if (g == 0) {
                if (__builtin_bit_cast(int, e0) != __builtin_bit_cast(int, d0) || __builtin_bit_cast(int, e1) != __builtin_bit_cast(int, d1)) atomicAdd(&diag[1], 1);
                if (__builtin_bit_cast(int, m[0]) != __builtin_bit_cast(int, l[0]) || __builtin_bit_cast(int, m[1]) != __builtin_bit_cast(int, l[1])) atomicAdd(&diag[2], 1);
            }
            d0 = e0; d1 = e1; l[0] = m[0]; l[1] = m[1];
            ++tries;
        }
        if (tries == 3 && g == 0) atomicAdd(&diag[3], 1);
        s0 = l[0]; s1 = l[1];
    };
    for (int level = a.levels - 1; level >= 0; --level) {
        const int w = a.w[level], h = a.h[level];
        const uint8_t* I = a.I[level];
        const uint8_t* J = a.J[level];
        const int16_t* D = a.D[level];
        const float sc = 1.f / (float)(1 << level);
        float ppx = px0 * sc, ppy = py0 * sc;
        if (level == a.levels - 1) { nx = ppx; ny = ppy; }
        else { nx *= 2.f; ny *= 2.f; }
        ppx -= half; ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
            if (level == 0) { st = false; er = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        weights(ppx - ipx, ppy - ipy, iw00, iw01, iw10, iw11);
        int ival = 0, ixval = 0, iyval = 0;
        if (lane_on) {
            const int xx0 = ipx + wx, xx1 = xx0 + 1, yy0 = ipy + wy, yy1 = yy0 + 1;
            ival = sample(I, w, h, ipx, ipy, iw00, iw01, iw10, iw11);
            auto dv = [&](int xx, int yy) -> int2 {
                if (xx < 0 || xx >= w || yy < 0 || yy >= h) return make_int2(0, 0);
                const int v = *reinterpret_cast<const int*>(D + ((size_t)yy * w + xx) * 2);
                return make_int2((int)(short)(v & 0xffff), (int)(short)(v >> 16));
            };
            const int2 d00 = dv(xx0, yy0), d01 = dv(xx1, yy0), d10 = dv(xx0, yy1), d11 = dv(xx1, yy1);
            ixval = LK_DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
            iyval = LK_DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
        }
        float A11, A12, A22;
        float pfx[3] = {0.f, 0.f, 0.f};
        {
            float s01, s02;
            // (three sums = the two-sum helper twice in the LDS / checked modes; the DPP mode keeps the production shape)
            if (MODE == 0) seq_sum3<NW>((float)(ixval * ixval), (float)(ixval * iyval), (float)(iyval * iyval), A11, A12, A22, CAP ? pfx : nullptr);
            else {
                sums2((float)(ixval * ixval), (float)(ixval * iyval), A11, A12, pfx);
                sums2((float)(iyval * iyval), (float)(ixval * iyval), A22, s01, pfx + 1);
                (void)s02;
            }
        }
        put(0, ival); put(1, ixval); put(2, iyval); putf(3, pfx[0]); putf(4, pfx[1]); putf(5, pfx[2]);
        putf(6, A11); put(7, (level << 8) | 1);
        ++nrec;
        A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
        float Dt = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < a.min_eig_thresh || Dt < 1.1920929e-07f) {
            if (level == 0) st = false;
            continue;
        }
        Dt = 1.f / Dt;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        float outx = nx + half, outy = ny + half;
        bool running = true;
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (running && (inx < -win || inx >= w || iny < -win || iny >= h)) {
                if (level == 0) st = false;
                running = false;
            }
            if (!running) break;
            weights(nx - inx, ny - iny, iw00, iw01, iw10, iw11);
            int diff = 0;
            if (lane_on) diff = sample(J, w, h, inx, iny, iw00, iw01, iw10, iw11) - ival;
            float b1, b2;
            float pb[2] = {0.f, 0.f};
            sums2((float)(diff * ixval), (float)(diff * iyval), b1, b2, CAP ? pb : nullptr);
            put(0, diff); putf(1, pb[0]); putf(2, pb[1]); putf(5, b1); putf(6, b2); put(7, (level << 8) | (j << 16) | 2);
            b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * Dt, dy = (A12 * b1 - A11 * b2) * Dt;
            putf(8, nx); putf(9, b1);
            nx += dx; ny += dy;
            putf(3, nx); putf(4, ny); putf(10, dx); putf(11, dy);
            ++nrec;
            if (CHK) {
                if (g == 0) atomicAdd(&diag[6], 1);
                const int fx = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, nx));
                const int fy = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ny));
                if (__builtin_bit_cast(int, nx) != fx || __builtin_bit_cast(int, ny) != fy) {
                    atomicAdd(&diag[5], 1);
                    atomicAdd(&diag[8 + (g >> 4)], 1);          // which quarter of the wavefront
                }
            }
            outx = nx + half; outy = ny + half;
            if (dx * dx + dy * dy <= a.eps2) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        nx = outx; ny = outy;
        if (st && level == 0) {
            const float ex = nx - half, ey = ny - half;
            const int inx = (int)floorf(ex), iny = (int)floorf(ey);
            if (inx < -win || inx >= w || iny < -win || iny >= h) { st = false; continue; }
            weights(ex - inx, ey - iny, iw00, iw01, iw10, iw11);
            int diff = 0;
            if (lane_on) diff = sample(J, w, h, inx, iny, iw00, iw01, iw10, iw11) - ival;
            float pe = 0.f;
            if (MODE == 0) er = seq_sum<NW>(fabsf((float)diff), CAP ? &pe : nullptr) * 1.f / (32 * win * win);
            else {
                float e1, e2;
                sums2(fabsf((float)diff), 0.f, e1, e2, nullptr);
                er = e1 * 1.f / (32 * win * win);
            }
            put(0, diff); putf(1, pe); putf(5, er); put(7, 3);
            ++nrec;
        }
    }
    if (g == 0) {
        next_pts[2 * pt] = nx;
        next_pts[2 * pt + 1] = ny;
        status[pt] = st ? 1 : 0;
        err[pt] = er;
        if (CAP) {
            cap_hdr[4 * pt] = (int)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
            cap_hdr[4 * pt + 1] = (int)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
            cap_hdr[4 * pt + 2] = nrec;
            cap_hdr[4 * pt + 3] = (int)blockIdx.x;
        }
    }
}

#endif  // FM_DIAG

// ---- foreground-mask bookkeeping: rect k sees pixel p as foreground iff no rect j<k covers p.
// Only earlier rects that intersect rect k can cover its pixels: the host passes that (short) list.
struct Overlaps { const int32_t* idx; const int32_t* off; };   // idx[off[k] .. off[k+1]) = j < k intersecting k

__device__ __forceinline__ bool covered_by(const int32_t* rects, const int32_t* list, int cnt, int x, int y) {
    for (int q = 0; q < cnt; ++q) {
        const int32_t* r = rects + 4 * list[q];
        if (x >= r[0] && x <= r[2] && y >= r[1] && y <= r[3]) return true;
    }
    return false;
}

__device__ __forceinline__ bool covered_any(const int32_t* rects, int n, int x, int y) {
    for (int j = 0; j < n; ++j) {
        const int32_t* r = rects + 4 * j;
        if (x >= r[0] && x <= r[2] && y >= r[1] && y <= r[3]) return true;
    }
    return false;
}

// i / w for 0 <= i < 2^23 without the ~40-instruction integer division: float estimate + one correction step
// (the estimate is off by at most one for any w the frame sizes allow; the correction makes it exact).
__device__ __forceinline__ int fast_div(int i, int w, float inv_w) {
    int q = (int)(((float)i + 0.5f) * inv_w);
    const int r = i - q * w;
    q += r >= w ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

constexpr int OV_LDS = 32;     // overlapping earlier rects of one track kept in LDS (more: global lists)

// (one 16-byte broadcast read per rect and no early exit: the short-circuit version was four dependent LDS reads and
// four branches per rect, in the per-pixel loops of the bookkeeping and GFTT kernels)
__device__ __forceinline__ bool covered_lds(const int* s_ov, int cnt, int x, int y) {
    bool c = false;
    for (int q = 0; q < cnt; ++q) {
        const int4 r = *reinterpret_cast<const int4*>(s_ov + 4 * q);
        c |= (x >= r.x) & (x <= r.z) & (y >= r.y) & (y <= r.w);
    }
    return c;
}

__global__ __launch_bounds__(256) void target_area_kernel(const int32_t* __restrict__ rects, Overlaps ov,
                                                          int32_t* __restrict__ area) {
    const int k = blockIdx.x;
    const int32_t* r = rects + 4 * k;
    const int w = r[2] - r[0] + 1, h = r[3] - r[1] + 1;
    const int32_t* list = ov.idx + ov.off[k];
    const int cnt = ov.off[k + 1] - ov.off[k];
    int c = 0;
    if (cnt == 0) {
        c = threadIdx.x == 0 ? w * h : 0;
    } else {
        for (int i = threadIdx.x; i < w * h; i += 256)
            c += covered_by(rects, list, cnt, r[0] + i % w, r[1] + i / w) ? 0 : 1;
    }
    __shared__ int red[256];
    red[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) area[k] = red[0];
}

// _rect_filter (flow.py:282-295): rounded point inside rect k and still foreground
__global__ void kp_filter_kernel(const int32_t* __restrict__ rects, Overlaps ov, const float* __restrict__ kps,
                                 const int32_t* __restrict__ kp_track, int n, uint8_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = kp_track[i];
    const int32_t* r = rects + 4 * k;
    const int x = (int)rintf(kps[2 * i]), y = (int)rintf(kps[2 * i + 1]);
    const bool inside = x >= r[0] && x <= r[2] && y >= r[1] && y <= r[3];
    keep[i] = (inside && !covered_by(rects, ov.idx + ov.off[k], ov.off[k + 1] - ov.off[k], x, y)) ? 1 : 0;
}

// fused bookkeeping of flow.py:163-169 for one track per block: mask area, _rect_filter of the
// propagated keypoints, the "too few keypoints" decision and the GFTT minDistance (flow.py:267-271)
// (1024 threads per track: a lone wavefront per SIMD issues one instruction every ~8 cycles; 256 threads took 49 us on the
// benchmark's crops)
constexpr int PREP_BLK = 1024;
__global__ __launch_bounds__(PREP_BLK) FM_SGPR_CAP void prepare_kernel(const int32_t* __restrict__ rects, Overlaps ov,
                                                      const float* __restrict__ kps,
                                                      const int32_t* __restrict__ kp_off, double feat_density,
                                                      double feat_dist_factor, int32_t* __restrict__ area,
                                                      uint8_t* __restrict__ keep, uint8_t* __restrict__ needy,
                                                      int32_t* __restrict__ min_dist,
                                                      uint8_t* __restrict__ needy_host) {
    const int k = blockIdx.x, tid = threadIdx.x;
    const int32_t* r = rects + 4 * k;
    const int r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    const int w = r2 - r0 + 1, h = r3 - r1 + 1;
    const int32_t* list = ov.idx + ov.off[k];
    const int cnt = ov.off[k + 1] - ov.off[k];
    __shared__ __attribute__((aligned(16))) int s_ov[4 * OV_LDS];
    const bool in_lds = cnt <= OV_LDS;
    if (in_lds && tid < 4 * cnt) s_ov[tid] = rects[4 * list[tid >> 2] + (tid & 3)];
    __syncthreads();
    auto covered = [&](int x, int y) { return in_lds ? covered_lds(s_ov, cnt, x, y) : covered_by(rects, list, cnt, x, y); };
    int c = 0, kept = 0;
    if (cnt == 0) c = tid == 0 ? w * h : 0;
    else if (in_lds) {
        // uncovered pixels, 64 columns of one row per step: OR of the column ranges of the rects that cross the row
        // (the per-pixel loop below is h * w * cnt rect tests: 95 us for the 300 tracks of a 4K stream)
        const int nwords = (w + 63) >> 6;
        const float inv_nw = 1.f / (float)nwords;
        for (int q = tid; q < h * nwords; q += PREP_BLK) {
            const int yy = fast_div(q, nwords, inv_nw), wd = q - yy * nwords;
            const int y = r1 + yy, xa = r0 + wd * 64;
            const int nbits = min(64, w - wd * 64);
            unsigned long long bits = 0ull;
            for (int j = 0; j < cnt; ++j) {
                const int4 rr = *reinterpret_cast<const int4*>(s_ov + 4 * j);
                const int lo = max(rr.x - xa, 0), hi = min(rr.z - xa, 63);          // inclusive bit range
                if (y >= rr.y && y <= rr.w && lo <= hi) bits |= (~0ull >> (63 - hi)) & (~0ull << lo);
            }
            const unsigned long long valid = nbits == 64 ? ~0ull : ((1ull << nbits) - 1ull);
            c += __popcll(~bits & valid);
        }
    } else {
        const float inv_w = 1.f / (float)w;
        for (int i = tid; i < w * h; i += PREP_BLK) {
            const int y = fast_div(i, w, inv_w), x = i - y * w;
            c += covered(r0 + x, r1 + y) ? 0 : 1;
        }
    }
    for (int i = kp_off[k] + tid; i < kp_off[k + 1]; i += PREP_BLK) {
        const int x = (int)rintf(kps[2 * i]), y = (int)rintf(kps[2 * i + 1]);
        const bool ok = x >= r0 && x <= r2 && y >= r1 && y <= r3 && !covered(x, y);
        keep[i] = ok ? 1 : 0;
        kept += ok ? 1 : 0;
    }
    __shared__ int red[PREP_BLK], red2[PREP_BLK];
    red[tid] = c;
    red2[tid] = kept;
    __syncthreads();
    for (int off = PREP_BLK / 2; off > 0; off >>= 1) {
        if (tid < off) { red[tid] += red[tid + off]; red2[tid] += red2[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        area[k] = red[0];
        const uint8_t nd = (double)red2[0] < feat_density * (double)red[0] ? 1 : 0;
        needy[k] = nd;
        if (needy_host) needy_host[k] = nd;
        const int md = (int)rint(sqrt((double)red[0]) * feat_dist_factor);
        min_dist[k] = md > 1 ? md : 1;
    }
}

// ---- goodFeaturesToTrack pieces (imgproc/featureselect.cpp, corner.cpp)
// min-eigenvalue map of one crop: Sobel 3x3 (scale 1/(4*block*255)), products, 3x3 box sum,
// borders REFLECT_101 at the crop edge (the crop is an isolated Mat)
struct CropArgs { int x0, y0, w, h, k; size_t eig_off; };

__device__ __forceinline__ void sobel_at(const uint8_t* img, int stride, const CropArgs& c, int x, int y,
                                         float scale, float& dx, float& dy) {
    int v[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int yy = c.y0 + reflect101(y + j - 1, c.h);
#pragma unroll
        for (int i = 0; i < 3; ++i) v[j][i] = img[(size_t)yy * stride + c.x0 + reflect101(x + i - 1, c.w)];
    }
    const int gx = (v[0][2] + 2 * v[1][2] + v[2][2]) - (v[0][0] + 2 * v[1][0] + v[2][0]);
    const int gy = (v[2][0] + 2 * v[2][1] + v[2][2]) - (v[0][0] + 2 * v[0][1] + v[0][2]);
    dx = gx * scale;
    dy = gy * scale;
}

// One pass over a crop does everything of goodFeaturesToTrack that is per pixel: the min-eigenvalue map (never stored),
// its masked maximum and the 3 x 3 local maxima among the unmasked pixels -- the only pixels the selection below can
// pick, whatever the threshold (quality * maximum) turns out to be.  A workgroup owns 32 x 32 pixels (four per lane; with
// 32 x 8 the kernel was bound by the chain of dependent loads every workgroup starts with -- crop, overlap list, rects,
// pixels -- times the 60000 workgroups of a 4K frame with 300 tracks: 213 us): Sobel pair once per
// position of the haloed tile into LDS, box sums for the tile and a one-pixel ring (the local-maximum test), then per
// tile (tile T of the crop is tile c.eig_off + T of the call):
//   tile_stat[..].x   float bits of the tile's masked maximum (0 if none: the reference's maximum starts at 0 too)
//   tile_stat[..].y   number of local maxima, their keys (value bits << 32 | raster index) in cand[1024 * tile + 0 ..]
// (the order of the keys does not matter: they are unique and get sorted).  No atomics: a first version appended to one
// list per crop and kept one maximum per crop with device-scope atomics -- 1600 of them on the same two addresses per
// crop, 640 us for the kernel.  Round 3 computed nine Sobel pairs per pixel (81 loads, 54 border reflections), stored
// the map, and the selection kernel scanned it twice with one workgroup per track: at 4K / 300 tracks 311 us + ~190 of
// the selection's 413 us.  Same values, same order of additions (rows inside, then down): bit-identical maps, so the
// same corners.
constexpr int EIG_TW = 32, EIG_TH = 32, EIG_TPX = EIG_TW * EIG_TH;
template <int R>
__global__ __launch_bounds__(256) FM_SGPR_CAP void eig_cand_kernel(const uint8_t* __restrict__ img, int stride,
                                                                   const CropArgs* __restrict__ crops,
                                                                   const int32_t* __restrict__ rects, Overlaps ov,
                                                                   const uint8_t* __restrict__ needy,
                                                                   uint2* __restrict__ tile_stat,
                                                                   unsigned long long* __restrict__ cand) {
    const int t = blockIdx.y, tid = threadIdx.x;
    const CropArgs c = crops[t];
    if (needy && !needy[c.k]) return;
    const int tiles_x = (c.w + EIG_TW - 1) / EIG_TW, tiles_y = (c.h + EIG_TH - 1) / EIG_TH;
    if ((int)blockIdx.x >= tiles_x * tiles_y) return;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * EIG_TW, y0 = ty * EIG_TH;
    constexpr int EW = EIG_TW + 2, EH = EIG_TH + 2, HW = EW + 2 * R, HH = EH + 2 * R, BS = 2 * R + 1;
    __shared__ float2 s_d[HH * HW];
    __shared__ float s_e[EH * EW];
    __shared__ __attribute__((aligned(16))) int s_ov[4 * OV_LDS];
    const int32_t* list = ov.idx + ov.off[c.k];
    const int lcnt = ov.off[c.k + 1] - ov.off[c.k];
    const bool ov_lds = lcnt <= OV_LDS;
    if (ov_lds && tid < 4 * lcnt) s_ov[tid] = rects[4 * list[tid >> 2] + (tid & 3)];
    // map values wanted: the tile and one ring around it, inside the crop; Sobel pairs: R further (reflected at the crop edge)
    const int ex_lo = max(x0 - 1, 0), ex_hi = min(x0 + EIG_TW + 1, c.w);
    const int ey_lo = max(y0 - 1, 0), ey_hi = min(y0 + EIG_TH + 1, c.h);
    const float scale = 1.f / (4.f * BS * 255.f);
    for (int q = tid; q < HH * HW; q += 256) {
        const int hy = q / HW, hx = q - hy * HW;
        const int px = x0 - 1 - R + hx, py = y0 - 1 - R + hy;
        if (px >= ex_lo - R && px < ex_hi + R && py >= ey_lo - R && py < ey_hi + R) {
            float dx, dy;
            if (px >= 1 && px < c.w - 1 && py >= 1 && py < c.h - 1) {
                // inside the crop nothing is reflected (eight reflections with their loops were two thirds of the
                // instructions of a position)
                const uint8_t* r0 = img + (size_t)(c.y0 + py - 1) * stride + c.x0 + px - 1;
                const uint8_t* r1 = r0 + stride;
                const uint8_t* r2 = r1 + stride;
                const int v00 = r0[0], v01 = r0[1], v02 = r0[2], v10 = r1[0], v12 = r1[2], v20 = r2[0], v21 = r2[1], v22 = r2[2];
                dx = (float)((v02 + 2 * v12 + v22) - (v00 + 2 * v10 + v20)) * scale;
                dy = (float)((v20 + 2 * v21 + v22) - (v00 + 2 * v01 + v02)) * scale;
            } else {
                sobel_at(img, stride, c, reflect101(px, c.w), reflect101(py, c.h), scale, dx, dy);
            }
            s_d[q] = make_float2(dx, dy);
        }
    }
    __syncthreads();
    // the mask of the tile, one 32-bit word per row (bit = covered by an earlier track's rect): one lane per row walks
    // the rect list once instead of every pixel walking it
    __shared__ uint32_t s_rowmask[EIG_TH];
    if (ov_lds && lcnt && tid >= 256 - EIG_TH) {
        const int row = tid - (256 - EIG_TH), yy = c.y0 + y0 + row, xa = c.x0 + x0;
        uint32_t bits = 0u;
        for (int j = 0; j < lcnt; ++j) {
            const int4 rr = *reinterpret_cast<const int4*>(s_ov + 4 * j);
            const int lo = max(rr.x - xa, 0), hi = min(rr.z - xa, EIG_TW - 1);      // inclusive bit range
            if (yy >= rr.y && yy <= rr.w && lo <= hi) bits |= (~0u >> (31 - hi)) & (~0u << lo);
        }
        s_rowmask[row] = bits;
    }
    for (int q = tid; q < EH * EW; q += 256) {
        const int ey = q / EW, ex = q - ey * EW;
        const int cx = x0 - 1 + ex, cy = y0 - 1 + ey;
        if (cx < ex_lo || cx >= ex_hi || cy < ey_lo || cy >= ey_hi) continue;
        float sxx = 0.f, sxy = 0.f, syy = 0.f;
#pragma unroll
        for (int j = 0; j < BS; ++j) {
            float rxx = 0.f, rxy = 0.f, ryy = 0.f;
#pragma unroll
            for (int ii = 0; ii < BS; ++ii) {
                const float2 d = s_d[(ey + j) * HW + ex + ii];
                rxx += d.x * d.x; rxy += d.x * d.y; ryy += d.y * d.y;
            }
            sxx += rxx; sxy += rxy; syy += ryy;
        }
        const float a = sxx * 0.5f, b = sxy, cc = syy * 0.5f;
        s_e[q] = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
    }
    __syncthreads();
    // a lane's pixels: column lx of rows ly, ly + 8, ...
    constexpr int NR = EIG_TH / 8;
    const int lane = tid & 63, wave = tid >> 6;
    const int lx = tid & (EIG_TW - 1), ly = tid / EIG_TW;
    const int x = x0 + lx;
    __shared__ float s_wmax[4];
    __shared__ int s_wcnt[NR * 4];
    float m = 0.f;
    float val[NR];
    unsigned long long bal[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int yl = ly + 8 * k, y = y0 + yl;
        const bool valid = x < c.w && y < c.h;
        const float* ec = s_e + (yl + 1) * EW + lx + 1;
        const float v = valid ? *ec : 0.f;
        bool open = valid && v > 0.f;               // (the masked maximum starts at 0; a candidate is > threshold >= 0)
        if (open && lcnt)
            open = !(ov_lds ? (bool)((s_rowmask[yl] >> lx) & 1u) : covered_by(rects, list, lcnt, c.x0 + x, c.y0 + y));
        m = fmaxf(m, open ? v : 0.f);
        // val >= its 8 neighbours (dilate inside the crop), 1-px border skipped
        bool emit = open && x >= 1 && y >= 1 && x < c.w - 1 && y < c.h - 1;
        if (emit) {
#pragma unroll
            for (int j = -1; j <= 1; ++j)
#pragma unroll
                for (int ii = -1; ii <= 1; ++ii)
                    if (ec[j * EW + ii] > v) emit = false;
        }
        val[k] = emit ? v : 0.f;                    // (an emitted value is > 0)
        bal[k] = __ballot(emit);
        if (lane == 0) s_wcnt[k * 4 + wave] = __popcll(bal[k]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) s_wmax[wave] = m;
    __syncthreads();
    const size_t tile = c.eig_off + blockIdx.x;
    int before = 0;                                 // keys of the (row group, wavefront) pairs ahead of this one
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        for (int q = 0; q < wave; ++q) before += s_wcnt[k * 4 + q];
        if (val[k] > 0.f)
            cand[tile * EIG_TPX + before + __popcll(bal[k] & ((1ull << lane) - 1ull))] =
                ((unsigned long long)__float_as_uint(val[k]) << 32) | (unsigned)((y0 + ly + 8 * k) * c.w + x);
        for (int q = wave; q < 4; ++q) before += s_wcnt[k * 4 + q];
    }
    if (tid == 0)
        tile_stat[tile] = make_uint2(__float_as_uint(fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]))),
                                     (unsigned)before);
}

// tiles of a crop / of the largest crop (grid.x of eig_cand_kernel)
static int eig_tiles(const CropArgs& c) { return ((c.w + EIG_TW - 1) / EIG_TW) * ((c.h + EIG_TH - 1) / EIG_TH); }
static int eig_max_tiles(const std::vector<CropArgs>& crops) {
    int m = 0;
    for (const CropArgs& c : crops) m = std::max(m, eig_tiles(c));
    return m;
}
static void launch_eig(hipStream_t s, const uint8_t* img, int stride, const CropArgs* d_crops, int n, int max_tiles,
                       const int32_t* rects, Overlaps ov, int block_size, const uint8_t* needy, uint2* tile_stat,
                       unsigned long long* cand) {
    if (block_size == 3)
        hipLaunchKernelGGL(eig_cand_kernel<1>, dim3(max_tiles, n), dim3(256), 0, s, img, stride, d_crops, rects, ov, needy,
                           tile_stat, cand);
    else   // (fm_flow_configure admits 3 and 5)
        hipLaunchKernelGGL(eig_cand_kernel<2>, dim3(max_tiles, n), dim3(256), 0, s, img, stride, d_crops, rects, ov, needy,
                           tile_stat, cand);
}

// one block per needy track: threshold (quality * the crop's masked maximum) over the crop's local maxima -- both
// from eig_cand_kernel -> sort in LDS -> min-distance selection on a cell grid (cell = minDistance, <= 4 accepted corners
// per cell, the same 3x3-cell neighbourhood test as featureselect.cpp) -> ellipse filter.
#ifdef FM_GFTT_TIMING
__device__ long long g_gftt_stamps[64][8];
#define GFTT_STAMP(i) if (threadIdx.x == 0 && blockIdx.x < 64) g_gftt_stamps[blockIdx.x][i] = __builtin_readcyclecounter();
#else
#define GFTT_STAMP(i)
#endif
constexpr int GFTT_MAX_CAND = 4096;      // sorted in LDS (32 KB)
constexpr int GFTT_MAX_CELLS = 1536;     // x 4 slots x 4 B = 24 KB
constexpr int GFTT_BLK = 1024;

__global__ __launch_bounds__(GFTT_BLK) FM_SGPR_CAP void gftt_select_kernel(const CropArgs* __restrict__ crops,
                                                               const uint2* __restrict__ tile_stat,
                                                               const unsigned long long* __restrict__ cand, float quality,
                                                               int max_corners, const int32_t* __restrict__ min_dist,
                                                               const double* __restrict__ full_tlbr,
                                                               float* __restrict__ pts_out, int cap,
                                                               int32_t* __restrict__ counts,
                                                               const uint8_t* __restrict__ needy,
                                                               int32_t* __restrict__ compact_total,
                                                               int32_t* __restrict__ compact_off) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const CropArgs c = crops[t];
    if (needy && !needy[c.k]) {            // enough propagated keypoints: nothing to detect
        if (tid == 0) {
            counts[t] = 0;
            if (compact_off) compact_off[t] = 0;
        }
        return;
    }
    GFTT_STAMP(0)
    __shared__ int s_n;
    __shared__ unsigned long long keys[GFTT_MAX_CAND];
    __shared__ __attribute__((aligned(16))) int cells[GFTT_MAX_CELLS * 4];
    __shared__ float s_red[GFTT_BLK / 64];
    __shared__ int s_tcnt[GFTT_BLK];
    const float inv_w = 1.f / (float)c.w;
    const int ntiles = ((c.w + EIG_TW - 1) / EIG_TW) * ((c.h + EIG_TH - 1) / EIG_TH);   // (tiles of eig_cand_kernel)
    const uint2* ts = tile_stat + c.eig_off;
    // (minMaxLoc with mask) * qualityLevel
    {
        float m = 0.f;
        for (int q = tid; q < ntiles; q += GFTT_BLK) m = fmaxf(m, __uint_as_float(ts[q].x));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((tid & 63) == 0) s_red[tid >> 6] = m;
        if (tid == 0) s_n = 0;
        __syncthreads();
    }
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < GFTT_BLK / 64; ++q) mx = fmaxf(mx, s_red[q]);
    const float thr = mx * quality;
    GFTT_STAMP(1)
    // candidates: local maxima with val > thr.  key = (float bits of val << 32) | raster index : descending 64-bit
    // order == (val desc, index desc), the order of std::sort(..., greaterThanPtr) in featureselect.cpp
    for (int base = 0; base < ntiles; base += GFTT_BLK) {
        __syncthreads();
        s_tcnt[tid] = base + tid < ntiles ? (int)ts[base + tid].y : 0;
        __syncthreads();
        const int nt = min(GFTT_BLK, ntiles - base);
        for (int q = tid; q < nt * 64; q += GFTT_BLK) {                   // 64 slots per tile and step; a tile rarely has more
            const int tl = q >> 6, cnt = s_tcnt[tl];
            const unsigned long long* cl = cand + (c.eig_off + base + tl) * EIG_TPX;
            for (int e = q & 63; e < cnt; e += 64) {
                const unsigned long long k = cl[e];
                if (!(__uint_as_float((unsigned)(k >> 32)) > thr)) continue;
                const int slot = atomicAdd(&s_n, 1);
                if (slot < GFTT_MAX_CAND) keys[slot] = k;
            }
        }
    }
    __syncthreads();
    GFTT_STAMP(2)
    const int n = min(s_n, GFTT_MAX_CAND);
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += GFTT_BLK) keys[i] = 0ull;     // padding sorts last
    __syncthreads();
    if (n <= GFTT_BLK) {
        // rank sort: keys are unique (the raster index is part of them), so a key's position is the number of larger
        // keys -- n broadcast LDS reads per thread, no dependent passes (the 55 passes of the bitonic network for
        // ~550 candidates took 36 k cycles, each one an LDS round trip).  Beyond one key per thread the network wins:
        // ranking up to four keys per thread in one sweep took 290 k cycles for 2200 candidates, the network 143 k.
        const unsigned long long mine = tid < n ? keys[tid] : 0ull;
        int rank = 0, q = 0;
        for (; q + 8 <= n; q += 8) {                  // eight reads in flight (a lone wavefront per SIMD has no other
            unsigned long long k8[8];                 // way to hide the LDS latency)
#pragma unroll
            for (int u = 0; u < 8; ++u) k8[u] = keys[q + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += k8[u] > mine ? 1 : 0;
        }
        for (; q < n; ++q) rank += keys[q] > mine ? 1 : 0;
        __syncthreads();
        if (tid < n) keys[rank] = mine;
        __syncthreads();
    } else {
        for (int k2 = 2; k2 <= np2; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < np2; i += GFTT_BLK) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a0 = keys[i], a1 = keys[l];
                        const bool desc = (i & k2) == 0;
                        if (desc ? a0 < a1 : a0 > a1) { keys[i] = a1; keys[l] = a0; }
                    }
                }
                __syncthreads();
            }
    }
    GFTT_STAMP(3)
    // Greedy min-distance selection (featureselect.cpp: candidates in sorted order, one is accepted iff no already
    // accepted corner is closer than minDistance).
    const int md = min_dist[needy ? c.k : t];
    const int gw = (c.w + md - 1) / md, gh = (c.h + md - 1) / md;
    const bool use_grid = gw * gh <= GFTT_MAX_CELLS;
    __shared__ short acc_x[1024], acc_y[1024];
    __shared__ int s_acc;
    __shared__ int s_surv[GFTT_BLK];
    __shared__ int s_wcnt[GFTT_BLK / 64];
    // a grid slot holds a corner as two int16 (y << 16 | x, crop coordinates < 16384); an empty slot holds a point far
    // outside every crop, so that the distance test needs no "slot in use" branch
    constexpr int GFTT_EMPTY = (int)0xC000C000u;          // (-16384, -16384)
    if (use_grid)
        for (int i = tid; i < gw * gh * 4; i += GFTT_BLK) cells[i] = GFTT_EMPTY;
    if (tid == 0) s_acc = 0;
    __syncthreads();
    const int md2 = md * md;
    const float inv_md = 1.f / (float)md;
    const int limit = min(max_corners, 1024);
    // is an accepted corner closer than minDistance?  (grid cells; the accepted list [0, nacc) when the grid does not fit LDS)
    auto blocked = [&](int x, int y, int nacc) -> bool {
        bool hit = false;
        if (use_grid) {
            // the 3 x 3 neighbourhood as nine independent 16-byte reads (clamped, masked afterwards): one LDS round trip
            // instead of up to 36 dependent ones (that loop was 57 % of the kernel: 15.7 k cycles per batch of 64
            // candidates, scripts/gftt_timing.py)
            const int cx0 = fast_div(x, md, inv_md), cy0 = fast_div(y, md, inv_md);
            int4 cv[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int cyn = min(max(cy0 + j / 3 - 1, 0), gh - 1), cxn = min(max(cx0 + j % 3 - 1, 0), gw - 1);
                cv[j] = *reinterpret_cast<const int4*>(cells + (cyn * gw + cxn) * 4);
            }
            // 36 slots, four instructions each: packed int16 difference, dot product with itself, compare, or.  (A clamped
            // neighbour outside the grid repeats a cell that is tested anyway.)
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const int me = (y << 16) | x;
            const s16x2 mev = *reinterpret_cast<const s16x2*>(&me);
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int pv[4] = {cv[j].x, cv[j].y, cv[j].z, cv[j].w};
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const s16x2 d = mev - *reinterpret_cast<const s16x2*>(&pv[sidx]);
                    hit |= __builtin_amdgcn_sdot2(d, d, 0, false) < md2;
                }
            }
        } else {
            for (int j = 0; j < nacc && !hit; ++j) {
                const int dx = x - acc_x[j], dy = y - acc_y[j];
                if (dx * dx + dy * dy < md2) hit = true;
            }
        }
        return hit;
    };
    // The sorted list is walked in spans of 64, 128, 256, ... candidates.  Per span:
    //   1. ALL wavefronts test the span's candidates against the corners accepted before the span -- whoever is blocked
    //      now stays blocked (corners are only added) -- and the survivors are compacted in order;
    //   2. the first wavefront takes the survivors 64 at a time: every lane re-tests its candidate (corners accepted
    //      earlier in this span), then the batch is resolved in lane (= sorted) order with register traffic only: the
    //      first survivor is accepted and knocks out the later ones closer than minDistance, repeat; every accepted lane
    //      files its own corner (list slot = its rank, grid slot by compare-and-swap).
    // Once the crop is covered nearly every candidate dies in step 1, on sixteen wavefronts instead of one: round 3 walked
    // all candidates in step 2, ~6.5 k cycles per 64 whatever their fate (120 k cycles for 1000 candidates, 220 k for 2200:
    // the longest phase of the kernel at 4K).
    int nacc = 0;                                          // (first wavefront: corners accepted so far)
    for (int start = 0, span = 64; start < n; start += span, span = min(span * 2, GFTT_BLK)) {
        const int cnt = min(span, n - start);
        int x = 0, y = 0;
        bool alive = false;
        if (tid < cnt) {
            const int ri = (int)(keys[start + tid] & 0xffffffffu);
            y = fast_div(ri, c.w, inv_w);
            x = ri - y * c.w;
            alive = start == 0 || !blocked(x, y, s_acc);
        }
        const unsigned long long sb = __ballot(alive);
        if ((tid & 63) == 0) s_wcnt[tid >> 6] = __popcll(sb);
        __syncthreads();
        int nsurv = 0, before = 0;
#pragma unroll
        for (int q = 0; q < GFTT_BLK / 64; ++q) {
            const int v = s_wcnt[q];
            before += q < (tid >> 6) ? v : 0;
            nsurv += v;
        }
        if (alive) s_surv[before + __popcll(sb & ((1ull << (tid & 63)) - 1ull))] = start + tid;
        __syncthreads();
        if (tid < 64) {
            for (int base = 0; base < nsurv && nacc < limit; base += 64) {
                bool live = base + tid < nsurv;
                if (live) {
                    const int ri = (int)(keys[s_surv[base + tid]] & 0xffffffffu);
                    y = fast_div(ri, c.w, inv_w);
                    x = ri - y * c.w;
                    live = !blocked(x, y, nacc);
                }
                unsigned long long mask = __ballot(live);
                unsigned long long accepted = 0ull;
                const int nacc0 = nacc;
                while (mask != 0ull && nacc < limit) {
                    const int first = __ffsll((long long)mask) - 1;
                    const int fx = __builtin_amdgcn_readlane(x, first), fy = __builtin_amdgcn_readlane(y, first);
                    accepted |= 1ull << first;
                    ++nacc;
                    if (live) {
                        const int dx = x - fx, dy = y - fy;
                        if (tid == first || dx * dx + dy * dy < md2) live = false;
                    }
                    mask = __ballot(live);
                }
                if ((accepted >> tid) & 1ull) {
                    const int slot = nacc0 + __popcll(accepted & ((1ull << tid) - 1ull));
                    acc_x[slot] = (short)x;
                    acc_y[slot] = (short)y;
                    if (use_grid) {
                        int* cell = cells + (fast_div(y, md, inv_md) * gw + fast_div(x, md, inv_md)) * 4;
                        const int pk = (y << 16) | x;
                        for (int sidx = 0; sidx < 4; ++sidx)
                            if (atomicCAS(&cell[sidx], GFTT_EMPTY, pk) == GFTT_EMPTY) break;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();          // the grid / list writes are visible to the next batch
            }
            if (tid == 0) s_acc = nacc;
        }
        __syncthreads();
        if (s_acc >= limit) break;               // (uniform)
    }
    GFTT_STAMP(4)
#ifdef FM_GFTT_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 64) { g_gftt_stamps[blockIdx.x][6] = n; g_gftt_stamps[blockIdx.x][7] = ((long long)(c.w * c.h) << 20) | s_acc; }
#endif
    // _ellipse_filter (flow.py:297-306): pts (f32) + offset (f32), then float64 ellipse test -- flags in
    // parallel, order-preserving compaction by ballot ranks
    const int nacc_all = s_acc;
    unsigned char* inside = reinterpret_cast<unsigned char*>(keys);       // the sort keys are dead
    {
        const double* b = full_tlbr + 4 * t;
        const double cx = (b[0] + b[2]) / 2, cy = (b[1] + b[3]) / 2;
        const double ax = (b[2] - b[0] + 1) * 0.5, ay = (b[3] - b[1] + 1) * 0.5;
        for (int q = tid; q < nacc_all; q += GFTT_BLK) {
            const float px = (float)acc_x[q] + (float)c.x0, py = (float)acc_y[q] + (float)c.y0;
            const double ux = ((double)px - cx) / ax, uy = ((double)py - cy) / ay;
            inside[q] = ux * ux + uy * uy <= 1. ? 1 : 0;
        }
    }
    __syncthreads();
    if (tid < 64) {
        int m = 0;
        for (int b0 = 0; b0 < nacc_all; b0 += 64)
            m += __popcll(__ballot(b0 + tid < nacc_all && inside[b0 + tid]));
        int base = 0;
        if (compact_total) {               // compacted output: reserve m slots, clip at the capacity
            if (tid == 0) {
                base = atomicAdd(compact_total, m);
                compact_off[t] = base;
            }
            base = __builtin_amdgcn_readfirstlane(base);
            m = max(0, min(m, cap - base));
        } else {
            base = t * cap;
            m = min(m, cap);
        }
        int w_ = 0;
        for (int b0 = 0; b0 < nacc_all && w_ < m; b0 += 64) {
            const int q = b0 + tid;
            const bool in = q < nacc_all && inside[q];
            const unsigned long long bal = __ballot(in);
            const int rank = w_ + __popcll(bal & ((1ull << tid) - 1ull));
            if (in && rank < m) {
                pts_out[((size_t)base + rank) * 2] = (float)acc_x[q] + (float)c.x0;
                pts_out[((size_t)base + rank) * 2 + 1] = (float)acc_y[q] + (float)c.y0;
            }
            w_ += __popcll(bal);
        }
        if (tid == 0) counts[t] = m;
    }
    GFTT_STAMP(5)
}

// ---- FAST-9/16 (features2d/fast.cpp), score = max threshold keeping the pixel a corner
__constant__ int c_fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int c_fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

__global__ void fast_score_kernel(const uint8_t* __restrict__ img, int w, int h, int thr,
                                  int32_t* __restrict__ score) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    int sc = 0;
    if (x >= 3 && y >= 3 && x < w - 3 && y < h - 3) {
        const int v = img[(size_t)y * w + x];
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = v - (int)img[(size_t)(y + c_fast_dy[k]) * w + x + c_fast_dx[k]];
        int best = 0;   // max over 9-arcs of min(d) (dark ring) and min(-d) (bright ring)
        for (int s = 0; s < 16; ++s) {
            int mn = 255, mxn = 255;
            for (int k = 0; k < 9; ++k) {
                const int dv = d[(s + k) & 15];
                mn = min(mn, dv);
                mxn = min(mxn, -dv);
            }
            best = max(best, max(mn, mxn));
        }
        if (best > thr) sc = best - 1;   // cornerScore: largest threshold for which it is a corner
    }
    score[(size_t)y * w + x] = sc;
}

// NMS + mask (INTER_NEAREST sample of the final foreground mask at the keypoint) -> flag per pixel, and the number
// of flags of every 256-pixel segment of the raster (one workgroup each).  The track rects are staged in LDS: with 300
// tracks every keypoint candidate walks all of them.
constexpr int FAST_SEG = 256, FAST_RECT_LDS = 1024;
__global__ __launch_bounds__(FAST_SEG) void fast_flag_kernel(const int32_t* __restrict__ score, int w, int h,
                                                             const int32_t* __restrict__ rects, int nT, int full_w,
                                                             int full_h, uint8_t* __restrict__ flag,
                                                             int32_t* __restrict__ seg_cnt) {
    __shared__ __attribute__((aligned(16))) int s_rect[4 * FAST_RECT_LDS];
    __shared__ int s_wave[FAST_SEG / 64];
    const bool in_lds = nT <= FAST_RECT_LDS;
    if (in_lds)
        for (int q = threadIdx.x; q < nT; q += FAST_SEG)
            *reinterpret_cast<int4*>(s_rect + 4 * q) = *reinterpret_cast<const int4*>(rects + 4 * q);
    __syncthreads();
    const int i = blockIdx.x * FAST_SEG + threadIdx.x;
    bool kp = false;
    if (i < w * h) {
        const int y = i / w, x = i - y * w;
        const int s = score[i];
        if (s > 0 && x >= 3 && y >= 3 && x < w - 3 && y < h - 3) {
            kp = s > score[i - 1] && s > score[i + 1] && s > score[i - w - 1] && s > score[i - w] &&
                 s > score[i - w + 1] && s > score[i + w - 1] && s > score[i + w] && s > score[i + w + 1];
            if (kp) {
                const int fx = min((int)floor((double)x * ((double)full_w / w)), full_w - 1);
                const int fy = min((int)floor((double)y * ((double)full_h / h)), full_h - 1);
                if (in_lds ? covered_lds(s_rect, nT, fx, fy) : covered_any(rects, nT, fx, fy)) kp = false;
            }
        }
        flag[i] = kp ? 1 : 0;
    }
    const int n = __popcll(__ballot(kp));
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) seg_cnt[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// raster-order compaction, one workgroup per segment: its first output slot is the sum of the earlier segments' counts
// (a few hundred values, summed by every workgroup for itself), inside the segment ballot ranks.  (One workgroup walking
// the whole raster with a block-wide scan took 250 us on the 384 x 216 background image of a 4K stream, most of it
// single stores into the pinned result block.)
__global__ __launch_bounds__(FAST_SEG) void fast_compact_kernel(const uint8_t* __restrict__ flag, int w, int h,
                                                                const int32_t* __restrict__ seg_cnt,
                                                                float* __restrict__ pts, int cap,
                                                                int32_t* __restrict__ n_out,
                                                                const int32_t* __restrict__ new_total,
                                                                int32_t* __restrict__ totals_host) {
    __shared__ int s_red[FAST_SEG / 64], s_wave[FAST_SEG / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int part = 0;
    for (int q = tid; q < (int)blockIdx.x; q += FAST_SEG) part += seg_cnt[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    const int i = blockIdx.x * FAST_SEG + tid;
    const bool on = i < w * h && flag[i];
    const unsigned long long bal = __ballot(on);
    if (lane == 0) { s_red[wave] = part; s_wave[wave] = __popcll(bal); }
    __syncthreads();
    int pos = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    for (int q = 0; q < wave; ++q) pos += s_wave[q];
    pos += __popcll(bal & ((1ull << lane) - 1ull));
    if (on && pos < cap) {
        const int y = i / w;
        *reinterpret_cast<float2*>(pts + 2 * (size_t)pos) = make_float2((float)(i - y * w), (float)y);
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const int total = s_red[0] + s_red[1] + s_red[2] + s_red[3] + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        *n_out = total;
        if (totals_host) {            // [0] = new keypoints (gftt_select_kernel, earlier on this stream), [1] = background
            if (new_total) totals_host[0] = new_total[0];
            totals_host[1] = total;
        }
    }
}

// scratch of the background branch inside FlowState::bg_flags: score map | 8 counters | segment counts | flag bytes
struct FastBufs { int32_t* score; int32_t* counters; int32_t* seg_cnt; uint8_t* flag; int nseg; };
static size_t fast_bufs_bytes(int bw, int bh) {
    const size_t npx = (size_t)bw * bh, nseg = (npx + FAST_SEG - 1) / FAST_SEG;
    return sizeof(int32_t) * (npx + 8 + nseg) + npx;
}
static FastBufs fast_bufs(int32_t* base, int bw, int bh) {
    const size_t npx = (size_t)bw * bh, nseg = (npx + FAST_SEG - 1) / FAST_SEG;
    FastBufs b;
    b.score = base;
    b.counters = base + npx;
    b.seg_cnt = b.counters + 8;
    b.flag = reinterpret_cast<uint8_t*>(b.seg_cnt + nseg);
    b.nseg = (int)nseg;
    return b;
}

__global__ void copy_total_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst) { dst[0] = src[0]; }

// The KLT stream, for every launch that may read the pyramid of the newest frame: that pyramid is built on s_flow2
// (fm_flow_begin) so that the keypoint preparation -- which only reads the PREVIOUS frame's images -- runs beside
// it; the first such reader makes s_flow wait for it.
int flow_stream(fm_ctx* ctx, hipStream_t* out) {
    if (ctx->pyr_pending) {
        FM_HIP(hipStreamWaitEvent(ctx->s_flow, ctx->ev_pyr, 0));
        ctx->pyr_pending = false;
    }
    *out = ctx->s_flow;
    return 0;
}

int build_pyramid(fm_ctx* ctx, FlowState* f, int set, hipStream_t s) {
    const int n = ctx->frame_w * ctx->frame_h;
    if (f->W == 2 * f->lw[0] && f->H == 2 * f->lh[0] && (f->W & 1) == 0) {
        // gray + half-resolution image in one pass over the frame; the large levels as one launch each
        // (derivatives + next level); the small ones in one workgroup
        fm_trace_mark(ctx, s, 42);
        hipLaunchKernelGGL(gray_half_kernel, dim3((f->lw[0] + 255) / 256, f->lh[0]), dim3(256), 0, s, ctx->frame_cur,
                           f->W, f->H, f->gray[set], f->pyr[set][0], gray_coeffs(f->cfg.gray_coeff_bits));
        // the levels of at most ~10 k pixels share one workgroup (1080p: from level 3; 4K: from level 4 -- level 3 of a
        // 4K frame, 32 k pixels, in the tail made it 53 us long), the larger ones get a launch each
        int tail = 0;
        while (tail < f->levels && f->lw[tail] * f->lh[tail] > 10000) ++tail;
        for (int l = 0; l < tail; ++l) {
            const bool has_next = l + 1 < f->levels;
            hipLaunchKernelGGL(pyr_level_kernel, dim3((f->lw[l] + 255) / 256, f->lh[l] + (has_next ? f->lh[l + 1] : 0)),
                               dim3(256), 0, s, f->pyr[set][l], f->lw[l], f->lh[l],
                               reinterpret_cast<int*>(f->deriv[set][l]), has_next ? f->pyr[set][l + 1] : nullptr,
                               has_next ? f->lw[l + 1] : 0, has_next ? f->lh[l + 1] : 0);
        }
        if (f->levels > tail) {
            PyrTail t{};
            for (int l = 0; l < f->levels; ++l) {
                t.img[l] = f->pyr[set][l];
                t.deriv[l] = reinterpret_cast<int*>(f->deriv[set][l]);
                t.w[l] = f->lw[l];
                t.h[l] = f->lh[l];
            }
            t.first = tail;
            t.levels = f->levels;
            hipLaunchKernelGGL(pyr_tail_kernel, dim3(1), dim3(1024), 0, s, t);
        }
        FM_HIP(hipGetLastError());
        fm_trace_mark(ctx, s, 43);
        return 0;
    }
    hipLaunchKernelGGL(gray_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ctx->frame_cur, f->gray[set], n,
                       gray_coeffs(f->cfg.gray_coeff_bits));
    hipLaunchKernelGGL(resize_linear_kernel, dim3((f->lw[0] + 255) / 256, f->lh[0]), dim3(256), 0, s,
                       f->gray[set], f->W, f->H, f->pyr[set][0], f->lw[0], f->lh[0]);
    for (int l = 1; l < f->levels; ++l)
        hipLaunchKernelGGL(pyrdown_kernel, dim3((f->lw[l] + 255) / 256, f->lh[l]), dim3(256), 0, s,
                           f->pyr[set][l - 1], f->lw[l - 1], f->lh[l - 1], f->pyr[set][l], f->lw[l], f->lh[l]);
    // Scharr derivatives of this pyramid: consumed when it has become the "previous" one
    for (int l = 0; l < f->levels; ++l)
        hipLaunchKernelGGL(scharr_kernel, dim3((f->lw[l] + 255) / 256, f->lh[l]), dim3(256), 0, s, f->pyr[set][l],
                           f->lw[l], f->lh[l], f->deriv[set][l]);
    FM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" int fm_flow_configure(fm_ctx* ctx, const fm_flow_cfg* cfg) {
    FM_CHECK_ARG(ctx && cfg && ctx->frame_w > 0);
    FM_CHECK_ARG((cfg->win_size == 3 || cfg->win_size == 5) && cfg->max_level >= 0 && cfg->max_level < MAX_LEVELS);   // LK window instances
    FM_CHECK_ARG(cfg->block_size == 3 || cfg->block_size == 5);
    FM_CHECK_ARG(ctx->frame_w <= 16384 && ctx->frame_h <= 16384);     // corner coordinates are packed as int16 pairs (gftt_select_kernel)
    FM_CHECK_ARG(cfg->gray_coeff_bits == 0 || cfg->gray_coeff_bits == 14 || cfg->gray_coeff_bits == 15);
    FM_HIP(hipDeviceSynchronize());
    if (ctx->flow) fm_flow_free(ctx->flow);
    FlowState* f = new FlowState();
    ctx->flow = f;
    f->cfg = *cfg;
    f->W = ctx->frame_w;
    f->H = ctx->frame_h;
    // buildOpticalFlowPyramid: stop when a level is not larger than the window
    int w = cfg->small_w, h = cfg->small_h;
    f->levels = 0;
    for (int l = 0; l <= cfg->max_level; ++l) {
        if (l > 0 && (w <= cfg->win_size || h <= cfg->win_size)) break;
        f->lw[l] = w;
        f->lh[l] = h;
        f->levels = l + 1;
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    for (int s = 0; s < 2; ++s) {
        FM_HIP(hipMalloc(&f->gray[s], (size_t)f->W * f->H));
        for (int l = 0; l < f->levels; ++l) FM_HIP(hipMalloc(&f->pyr[s][l], (size_t)f->lw[l] * f->lh[l]));
    }
    for (int st = 0; st < 2; ++st)
        for (int l = 0; l < f->levels; ++l) FM_HIP(hipMalloc(&f->deriv[st][l], (size_t)f->lw[l] * f->lh[l] * 4));
    FM_HIP(hipMalloc(&f->bg_img, (size_t)cfg->bg_w * cfg->bg_h));
    FM_HIP(hipMalloc(&f->bg_flags, fast_bufs_bytes(cfg->bg_w, cfg->bg_h)));
    f->eig_cap = (size_t)4 * f->W * f->H / EIG_TPX;
    FM_HIP(hipMalloc(&f->eig, sizeof(unsigned long long) * EIG_TPX * f->eig_cap));
    FM_HIP(hipMalloc(&f->tile_stat, sizeof(uint2) * f->eig_cap));
    return 0;
}

extern "C" int fm_flow_init(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->flow && ctx->frame_cur);
    FlowState* f = ctx->flow;
    f->prev = 0;
    int rc = build_pyramid(ctx, f, 0, ctx->s_flow);
    if (rc) return rc;
    f->nT = 0;
    return 0;
}

extern "C" int fm_flow_begin(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->flow && ctx->frame_cur);
    hipStream_t s;
    int rc = flow_stream(ctx, &s);          // (a pyramid nobody read: order this one behind it)
    if (rc) return rc;
    // the side stream starts behind whatever s_flow still does with the buffers of this set (nothing inside
    // fm_flow_predict, which ends synchronised; the step functions used by tests may have work queued)
    FM_HIP(hipEventRecord(ctx->ev_pyr, ctx->s_flow));
    FM_HIP(hipStreamWaitEvent(ctx->s_flow2, ctx->ev_pyr, 0));
    if ((rc = build_pyramid(ctx, ctx->flow, ctx->flow->prev ^ 1, ctx->s_flow2))) return rc;
    FM_HIP(hipEventRecord(ctx->ev_pyr, ctx->s_flow2));
    ctx->pyr_pending = true;
    return 0;
}

extern "C" int fm_flow_swap(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->flow);
    hipStream_t s;
    int rc = flow_stream(ctx, &s);          // from now on the new pyramid is the "previous" one every launch reads
    if (rc) return rc;
    ctx->flow->prev ^= 1;
    return 0;
}

extern "C" int fm_flow_targets(fm_ctx* ctx, int nT, const double* inside_tlbr, const float* kps,
                               const int32_t* kp_off, int32_t* area_out, uint8_t* keep_out) {
    FM_CHECK_ARG(ctx && ctx->flow && nT >= 0);
    FlowState* f = ctx->flow;
    hipStream_t s;
    { int rc_s_ = flow_stream(ctx, &s); if (rc_s_) return rc_s_; }
    f->nT = nT;
    if (nT == 0) return 0;
    FM_CHECK_ARG(inside_tlbr && kp_off && area_out);
    const int nk = kp_off[nT];
    FM_CHECK_ARG(nk == 0 || (kps && keep_out));
    if (nT > f->rect_cap) {
        FM_HIP(hipStreamSynchronize(s));
        if (f->rects) FM_HIP(hipFree(f->rects));
        f->rects = nullptr;
        int cap = f->rect_cap ? f->rect_cap : 64;
        while (cap < nT) cap *= 2;
        FM_HIP(hipMalloc(&f->rects, sizeof(int32_t) * 4 * cap));
        f->rect_cap = cap;
    }
    // overlap lists: earlier rects (closer tracks) that intersect rect k
    std::vector<int32_t> irect(4 * (size_t)nT), ov_off(nT + 1, 0), ov_idx;
    for (int k = 0; k < nT; ++k)
        for (int e = 0; e < 4; ++e) irect[4 * k + e] = (int32_t)inside_tlbr[4 * k + e];   // crop(): int() truncation
    for (int k = 0; k < nT; ++k) {
        const int32_t* a = &irect[4 * k];
        for (int j = 0; j < k; ++j) {
            const int32_t* b = &irect[4 * j];
            if (b[0] <= a[2] && b[2] >= a[0] && b[1] <= a[3] && b[3] >= a[1]) ov_idx.push_back(j);
        }
        ov_off[k + 1] = (int32_t)ov_idx.size();
    }
    const int n_ov = (int)ov_idx.size();
    if (nT + 1 > f->ov_cap || n_ov > f->ov_cap * 8) {
        FM_HIP(hipStreamSynchronize(s));
        if (f->ov_idx) FM_HIP(hipFree(f->ov_idx));
        if (f->ov_off) FM_HIP(hipFree(f->ov_off));
        f->ov_idx = f->ov_off = nullptr;
        int cap = f->ov_cap ? f->ov_cap : 64;
        while (cap < nT + 1 || cap * 8 < n_ov) cap *= 2;
        FM_HIP(hipMalloc(&f->ov_off, sizeof(int32_t) * cap));
        FM_HIP(hipMalloc(&f->ov_idx, sizeof(int32_t) * cap * 8));
        f->ov_cap = cap;
    }
    // packed upload: rects i32[nT][4] | kp_track i32[nk] | kps f32[nk][2] | ov_off i32[nT+1] | ov_idx i32[n_ov]
    const size_t o_rect = 0, o_trk = sizeof(int32_t) * 4 * nT, o_kps = o_trk + sizeof(int32_t) * nk;
    const size_t o_ovoff = o_kps + sizeof(float) * 2 * nk, o_ovidx = o_ovoff + sizeof(int32_t) * (nT + 1);
    const size_t in_bytes = o_ovidx + sizeof(int32_t) * n_ov;
    int rc = f->tgt_in.reserve(in_bytes + 16);
    if (rc) return rc;
    if ((rc = f->tgt_out.reserve(sizeof(int32_t) * nT + nk + 16))) return rc;
    FM_HIP(hipStreamSynchronize(s));
    char* hbuf = f->tgt_in.host<char>();
    memcpy(hbuf + o_rect, irect.data(), sizeof(int32_t) * 4 * nT);
    memcpy(hbuf + o_ovoff, ov_off.data(), sizeof(int32_t) * (nT + 1));
    if (n_ov) memcpy(hbuf + o_ovidx, ov_idx.data(), sizeof(int32_t) * n_ov);
    int32_t* ht = reinterpret_cast<int32_t*>(hbuf + o_trk);
    for (int k = 0; k < nT; ++k)
        for (int i = kp_off[k]; i < kp_off[k + 1]; ++i) ht[i] = k;
    if (nk) memcpy(hbuf + o_kps, kps, sizeof(float) * 2 * nk);
    FM_HIP(hipMemcpyAsync(f->tgt_in.d, hbuf, in_bytes, hipMemcpyHostToDevice, s));
    FM_HIP(hipMemcpyAsync(f->rects, f->tgt_in.dev<char>() + o_rect, sizeof(int32_t) * 4 * nT, hipMemcpyDeviceToDevice, s));
    FM_HIP(hipMemcpyAsync(f->ov_off, f->tgt_in.dev<char>() + o_ovoff, sizeof(int32_t) * (nT + 1), hipMemcpyDeviceToDevice, s));
    if (n_ov) FM_HIP(hipMemcpyAsync(f->ov_idx, f->tgt_in.dev<char>() + o_ovidx, sizeof(int32_t) * n_ov, hipMemcpyDeviceToDevice, s));
    f->v_rects = f->rects; f->v_ov_idx = f->ov_idx; f->v_ov_off = f->ov_off;
    const Overlaps ov{f->ov_idx, f->ov_off};
    int32_t* d_area = f->tgt_out.dev<int32_t>();
    uint8_t* d_keep = reinterpret_cast<uint8_t*>(d_area + nT);
    hipLaunchKernelGGL(target_area_kernel, dim3(nT), dim3(256), 0, s, f->rects, ov, d_area);
    if (nk)
        hipLaunchKernelGGL(kp_filter_kernel, dim3((nk + 255) / 256), dim3(256), 0, s, f->rects, ov,
                           reinterpret_cast<const float*>(f->tgt_in.dev<char>() + o_kps),
                           reinterpret_cast<const int32_t*>(f->tgt_in.dev<char>() + o_trk), nk, d_keep);
    FM_HIP(hipGetLastError());
    FM_HIP(hipMemcpyAsync(f->tgt_out.h, f->tgt_out.d, sizeof(int32_t) * nT + nk, hipMemcpyDeviceToHost, s));
    FM_HIP(hipStreamSynchronize(s));
    memcpy(area_out, f->tgt_out.h, sizeof(int32_t) * nT);
    if (nk) memcpy(keep_out, f->tgt_out.host<char>() + sizeof(int32_t) * nT, nk);
    return 0;
}

extern "C" int fm_flow_detect(fm_ctx* ctx, int n, const int32_t* track_idx, const double* track_tlbr,
                              const int32_t* min_dist, int cap, float* pts_out, int32_t* counts_out) {
    FM_CHECK_ARG(ctx && ctx->flow && n >= 0 && cap > 0);
    if (n == 0) return 0;
    FM_CHECK_ARG(track_idx && track_tlbr && min_dist && pts_out && counts_out);
    FlowState* f = ctx->flow;
    hipStream_t s;
    { int rc_s_ = flow_stream(ctx, &s); if (rc_s_) return rc_s_; }
    FM_HIP(hipStreamSynchronize(s));
    const int32_t* hrects = reinterpret_cast<const int32_t*>(f->tgt_in.host<char>());
    std::vector<CropArgs> crops(n);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        FM_CHECK_ARG(track_idx[i] >= 0 && track_idx[i] < f->nT);
        const int32_t* r = hrects + 4 * track_idx[i];
        CropArgs c;
        c.x0 = r[0]; c.y0 = r[1]; c.w = r[2] - r[0] + 1; c.h = r[3] - r[1] + 1; c.k = track_idx[i];
        c.eig_off = off;
        off += (size_t)eig_tiles(c);
        crops[i] = c;
    }
    if (off > f->eig_cap) {     // many small crops (one tile at least each): grow like fm_flow_prepare does (stream idle: synchronised above)
        f->eig_cap = 0;         // (a failing allocation below must not leave a capacity behind null pointers)
        FM_HIP(hipFree(f->eig));
        f->eig = nullptr;
        FM_HIP(hipFree(f->tile_stat));
        f->tile_stat = nullptr;
        FM_HIP(hipMalloc(&f->eig, sizeof(unsigned long long) * EIG_TPX * off * 2));
        FM_HIP(hipMalloc(&f->tile_stat, sizeof(uint2) * off * 2));
        f->eig_cap = off * 2;
    }
    const size_t o_crop = 0, o_md = sizeof(CropArgs) * n, o_box = (o_md + sizeof(int32_t) * n + 15) & ~size_t(15);
    const size_t in_bytes = o_box + sizeof(double) * 4 * n;
    int rc = f->det_in.reserve(in_bytes);
    if (rc) return rc;
    const size_t out_bytes = sizeof(float) * 2 * (size_t)n * cap + sizeof(int32_t) * n;
    if ((rc = f->det_out.reserve(out_bytes))) return rc;
    char* hb = f->det_in.host<char>();
    memcpy(hb + o_crop, crops.data(), sizeof(CropArgs) * n);
    memcpy(hb + o_md, min_dist, sizeof(int32_t) * n);
    memcpy(hb + o_box, track_tlbr, sizeof(double) * 4 * n);
    FM_HIP(hipMemcpyAsync(f->det_in.d, hb, in_bytes, hipMemcpyHostToDevice, s));
    char* db = f->det_in.dev<char>();
    launch_eig(s, f->gray[f->prev], f->W, reinterpret_cast<const CropArgs*>(db + o_crop), n, eig_max_tiles(crops),
               f->v_rects, Overlaps{f->v_ov_idx, f->v_ov_off}, f->cfg.block_size, nullptr, f->tile_stat, f->eig);
    float* d_pts = f->det_out.dev<float>();
    int32_t* d_cnt = reinterpret_cast<int32_t*>(d_pts + 2 * (size_t)n * cap);
    hipLaunchKernelGGL(gftt_select_kernel, dim3(n), dim3(GFTT_BLK), 0, s, reinterpret_cast<const CropArgs*>(db + o_crop),
                       f->tile_stat, f->eig, (float)f->cfg.quality_level,
                       f->cfg.max_corners, reinterpret_cast<const int32_t*>(db + o_md),
                       reinterpret_cast<const double*>(db + o_box), d_pts, cap, d_cnt, nullptr, nullptr, nullptr);
    FM_HIP(hipGetLastError());
    FM_HIP(hipMemcpyAsync(f->det_out.h, f->det_out.d, out_bytes, hipMemcpyDeviceToHost, s));
    FM_HIP(hipStreamSynchronize(s));
    memcpy(pts_out, f->det_out.h, sizeof(float) * 2 * (size_t)n * cap);
    memcpy(counts_out, f->det_out.host<char>() + sizeof(float) * 2 * (size_t)n * cap, sizeof(int32_t) * n);
    return 0;
}

extern "C" int fm_flow_background(fm_ctx* ctx, int cap, float* pts_out, int* n_out) {
    FM_CHECK_ARG(ctx && ctx->flow && cap > 0 && pts_out && n_out);
    FlowState* f = ctx->flow;
    hipStream_t s;
    { int rc_s_ = flow_stream(ctx, &s); if (rc_s_) return rc_s_; }
    const int bw = f->cfg.bg_w, bh = f->cfg.bg_h;
    int rc = f->bg_out.reserve(sizeof(float) * 2 * cap + 16);
    if (rc) return rc;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((bw + 255) / 256, bh), dim3(256), 0, s, f->gray[f->prev], f->W,
                       f->H, f->bg_img, bw, bh);
    hipLaunchKernelGGL(fast_score_kernel, dim3((bw + 63) / 64, bh), dim3(64), 0, s, f->bg_img, bw, bh,
                       f->cfg.fast_thresh, f->bg_flags);
    const FastBufs fb = fast_bufs(f->bg_flags, bw, bh);
    int32_t* d_n = fb.counters;
    hipLaunchKernelGGL(fast_flag_kernel, dim3(fb.nseg), dim3(FAST_SEG), 0, s, fb.score, bw, bh, f->v_rects, f->nT, f->W,
                       f->H, fb.flag, fb.seg_cnt);
    hipLaunchKernelGGL(fast_compact_kernel, dim3(fb.nseg), dim3(FAST_SEG), 0, s, fb.flag, bw, bh, fb.seg_cnt,
                       f->bg_out.dev<float>(), cap, d_n, nullptr, nullptr);
    FM_HIP(hipGetLastError());
    FM_HIP(hipMemcpyAsync(f->bg_out.host<char>() + sizeof(float) * 2 * cap, d_n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    FM_HIP(hipMemcpyAsync(f->bg_out.h, f->bg_out.d, sizeof(float) * 2 * cap, hipMemcpyDeviceToHost, s));
    FM_HIP(hipStreamSynchronize(s));
    int n = *reinterpret_cast<int32_t*>(f->bg_out.host<char>() + sizeof(float) * 2 * cap);
    if (n > cap) {
        fm_set_error("background keypoint capacity %d < %d", cap, n);
        return FM_ERR_ARG;
    }
    memcpy(pts_out, f->bg_out.h, sizeof(float) * 2 * n);
    *n_out = n;
    return 0;
}

extern "C" int fm_flow_lk(fm_ctx* ctx, int n, const float* prev_pts, float* next_pts, uint8_t* status,
                          float* err) {
    FM_CHECK_ARG(ctx && ctx->flow && n >= 0);
    FlowState* f = ctx->flow;
    hipStream_t s;
    { int rc_s_ = flow_stream(ctx, &s); if (rc_s_) return rc_s_; }
    if (n > 0) {
        FM_CHECK_ARG(prev_pts && next_pts && status && err);
        int rc = f->lk_in.reserve(sizeof(float) * 2 * n);
        if (rc) return rc;
        const size_t o_st = sizeof(float) * 2 * n, o_err = (o_st + n + 15) & ~size_t(15);
        const size_t out_bytes = o_err + sizeof(float) * n;
        if ((rc = f->lk_out.reserve(out_bytes))) return rc;
        double tl0 = fm_now_ms();
        FM_HIP(hipStreamSynchronize(s));
        g_flow_sub[4] += fm_now_ms() - tl0; tl0 = fm_now_ms();
        // points in and results out through device-mapped pinned host memory (each wavefront reads 8 bytes and writes
        // 13: no blit copies around the kernel; measured 0.127 ms per call against 0.147 ms with H2D + D2H copies)
        memcpy(f->lk_in.h, prev_pts, sizeof(float) * 2 * n);
        const int p = f->prev, c = p ^ 1;
        LKArgs a{};
        for (int l = 0; l < f->levels; ++l) {
            a.I[l] = f->pyr[p][l];
            a.J[l] = f->pyr[c][l];
            a.D[l] = f->deriv[p][l];
            a.w[l] = f->lw[l];
            a.h[l] = f->lh[l];
        }
        a.levels = f->levels;
        a.win = f->cfg.win_size;
        a.max_count = std::min(std::max(f->cfg.max_count, 0), 100);
        const double eps = std::min(std::max(f->cfg.epsilon, 0.), 10.);
        a.eps2 = (float)(eps * eps);
        a.min_eig_thresh = 1e-4f;
        char* o = f->lk_out.host<char>();
        {
            float* o_pts = reinterpret_cast<float*>(o);
            uint8_t* o_stat = reinterpret_cast<uint8_t*>(o + o_st);
            float* o_errp = reinterpret_cast<float*>(o + o_err);
            const float* in_pts = f->lk_in.host<float>();
            // ordinary 4-point workgroups beside whatever the other streams run (rounds 1-2 gave this launch whole CUs
            // or ordered the ReID network behind it; see the note at lk_wave_body for what was actually wrong)
            const int threads = 256, lds_req = 0;
            const dim3 grid((unsigned)(((size_t)n * 64 + threads - 1) / threads));
#ifdef FM_DIAG
            if (ctx->opt_lk_variant > 0 && a.win == 5) {
                const int v = ctx->opt_lk_variant;
                if (!f->lk_diag) {
                    FM_HIP(hipMalloc(&f->lk_diag, sizeof(int) * 16));
                    FM_HIP(hipMemset(f->lk_diag, 0, sizeof(int) * 16));
                }
                if ((v & 32) && f->lk_cap_pts < n) {
                    if (f->lk_cap) { (void)hipFree(f->lk_cap); (void)hipFree(f->lk_cap_hdr); }
                    FM_HIP(hipMalloc(&f->lk_cap, sizeof(int) * (size_t)n * LK_CAP_MAXREC * LK_CAP_REC));
                    FM_HIP(hipMalloc(&f->lk_cap_hdr, sizeof(int) * 4 * (size_t)n));
                    f->lk_cap_pts = n;
                }
#define FM_LK_DIAG_LAUNCH(MODE, CHK, CAP) \
    hipLaunchKernelGGL((lk_diag_kernel<5, MODE, CHK, CAP>), grid, dim3(threads), lds_req, s, a, n, in_pts, o_pts, o_stat, o_errp, \
                       f->lk_diag, f->lk_cap, f->lk_cap_hdr)
                if (v == 128) {                                              // one point per wavefront (the round-2 kernel)
                    hipLaunchKernelGGL(lk_wave_kernel<5>, grid, dim3(threads), lds_req, s, a, n, in_pts, o_pts, o_stat, o_errp);
                } else
                switch (v) {
                case 1 | 64: FM_LK_DIAG_LAUNCH(0, false, false); break;      // DPP sums, diagnostic body (control)
                case 1 | 32: FM_LK_DIAG_LAUNCH(0, false, true); break;       // DPP sums + capture
                case 1 | 16: FM_LK_DIAG_LAUNCH(0, true, false); break;       // DPP sums + duplicate loads / lane agreement
                case 2: FM_LK_DIAG_LAUNCH(1, false, false); break;           // LDS sums
                case 2 | 16: FM_LK_DIAG_LAUNCH(1, true, false); break;
                case 3: FM_LK_DIAG_LAUNCH(2, false, false); break;           // both, compared
                case 3 | 16: FM_LK_DIAG_LAUNCH(2, true, false); break;
                default: fm_set_error("unknown lk_variant %d", v); return FM_ERR_ARG;
                }
#undef FM_LK_DIAG_LAUNCH
            } else {
#else
            {
#endif
                const dim3 grid2((unsigned)((((size_t)n + 1) / 2 * 64 + threads - 1) / threads));     // two points per wavefront
                fm_trace_mark(ctx, s, 40);
                // (PATCH = false, the global-load sampling of rounds 3-4, measured equal within the spread:
                // profiles/r05_lk_patch_and_ring_depth_ab.txt)
                // (lk_pair_pre_kernel against lk_pair_kernel: 77 vs 90 us inside the pipeline at 1080p, config[4] 185 vs 176
                // frames/s, profiles/r06_lk_template_terms_ab.txt)
                if (a.levels <= LK_PRE) {      // template terms of all levels up front (bit-identical; see the kernel)
                    if (a.win == 5)
                        hipLaunchKernelGGL((lk_pair_pre_kernel<5>), grid2, dim3(threads), 0, s, a, n, in_pts, o_pts, o_stat, o_errp);
                    else
                        hipLaunchKernelGGL((lk_pair_pre_kernel<3>), grid2, dim3(threads), 0, s, a, n, in_pts, o_pts, o_stat, o_errp);
                } else if (a.win == 5)
                    hipLaunchKernelGGL((lk_pair_kernel<5, true>), grid2, dim3(threads), 0, s, a, n, in_pts, o_pts, o_stat, o_errp);
                else
                    hipLaunchKernelGGL((lk_pair_kernel<3, true>), grid2, dim3(threads), 0, s, a, n, in_pts, o_pts, o_stat, o_errp);
                fm_trace_mark(ctx, s, 41);
            }
        }
        FM_HIP(hipGetLastError());
        g_flow_sub[5] += fm_now_ms() - tl0; tl0 = fm_now_ms();
        FM_HIP(hipStreamSynchronize(s));
        g_flow_sub[6] += fm_now_ms() - tl0;
        const char* ho = f->lk_out.host<char>();
        memcpy(next_pts, ho, sizeof(float) * 2 * n);
        memcpy(status, ho + o_st, n);
        memcpy(err, ho + o_err, sizeof(float) * n);
    } else {
        FM_HIP(hipStreamSynchronize(s));
    }
    f->prev ^= 1;   // save preprocessed frame buffers for the next prediction (flow.py:212-213)
    return 0;
}

#ifdef FM_DIAG
// diagnostic read-out of the LK variants: 16 counters (reset afterwards), and for the capture variant the per-point
// headers [n][4] and records [n][LK_CAP_MAXREC][8][64] of the last call
extern "C" int fm_flow_lk_diag_read(fm_ctx* ctx, int32_t* counters, int n, int32_t* hdr, int32_t* records) {
    FM_CHECK_ARG(ctx && ctx->flow && counters);
    FlowState* f = ctx->flow;
    FM_HIP(hipDeviceSynchronize());
    if (!f->lk_diag) { memset(counters, 0, sizeof(int32_t) * 16); return 0; }
    FM_HIP(hipMemcpy(counters, f->lk_diag, sizeof(int32_t) * 16, hipMemcpyDeviceToHost));
    FM_HIP(hipMemset(f->lk_diag, 0, sizeof(int) * 16));
    if (hdr && records) {
        FM_CHECK_ARG(n > 0 && n <= f->lk_cap_pts);
        FM_HIP(hipMemcpy(hdr, f->lk_cap_hdr, sizeof(int32_t) * 4 * (size_t)n, hipMemcpyDeviceToHost));
        FM_HIP(hipMemcpy(records, f->lk_cap, sizeof(int32_t) * (size_t)n * LK_CAP_MAXREC * LK_CAP_REC, hipMemcpyDeviceToHost));
    }
    return 0;
}

#endif  // FM_DIAG

extern "C" int fm_flow_read_image(fm_ctx* ctx, int which, uint8_t* out, int* w, int* h) {
    FM_CHECK_ARG(ctx && ctx->flow && out && w && h);
    FlowState* f = ctx->flow;
    FM_HIP(hipDeviceSynchronize());
    const uint8_t* src = nullptr;
    if (which == 0 || which == 1) {
        src = f->gray[which == 0 ? f->prev : f->prev ^ 1];
        *w = f->W; *h = f->H;
    } else if (which >= 2 && which < 2 + f->levels) {
        src = f->pyr[f->prev][which - 2];
        *w = f->lw[which - 2]; *h = f->lh[which - 2];
    } else if (which >= 10 && which < 10 + f->levels) {
        src = f->pyr[f->prev ^ 1][which - 10];
        *w = f->lw[which - 10]; *h = f->lh[which - 10];
    } else if (which == 20) {
        src = f->bg_img;
        *w = f->cfg.bg_w; *h = f->cfg.bg_h;
    } else {
        fm_set_error("unknown image id %d", which);
        return FM_ERR_ARG;
    }
    FM_HIP(hipMemcpy(out, src, (size_t)(*w) * (*h), hipMemcpyDeviceToHost));
    return 0;
}

// One-sync version of flow.py:159-200 for all tracks: mask bookkeeping, neediness, GFTT for the
// needy tracks, background FAST keypoints.  Variable-length outputs are written by the kernels
// straight into pinned host memory (only the produced bytes cross PCIe).
extern "C" int fm_flow_prepare(fm_ctx* ctx, int nT, const double* inside_tlbr, const double* full_tlbr,
                               const float* kps, const int32_t* kp_off, double feat_density,
                               double feat_dist_factor, int32_t* area_out, uint8_t* keep_out,
                               uint8_t* needy_out, int pts_cap, float* new_pts_out, int32_t* new_off_out,
                               int32_t* new_cnt_out, int* n_new_out, int bg_cap, float* bg_pts_out,
                               int* n_bg_out) {
    FM_CHECK_ARG(ctx && ctx->flow && nT >= 0 && pts_cap > 0 && bg_cap > 0 && n_new_out && n_bg_out && bg_pts_out);
    FlowState* f = ctx->flow;
    hipStream_t s = ctx->s_flow;            // (reads only the previous frame: no wait for the new pyramid)
    f->nT = nT;
    const int nk = nT ? kp_off[nT] : 0;
    double tp0 = fm_now_ms();
    // ---- host side: integer rects, overlap lists, crop table
    std::vector<int32_t> irect(4 * (size_t)nT), ov_off(nT + 1, 0), ov_idx;
    std::vector<CropArgs> crops(nT);
    size_t eig_total = 0;
    for (int k = 0; k < nT; ++k) {
        for (int e = 0; e < 4; ++e) irect[4 * k + e] = (int32_t)inside_tlbr[4 * k + e];
        const int32_t* a = &irect[4 * k];
        for (int j = 0; j < k; ++j) {
            const int32_t* b = &irect[4 * j];
            if (b[0] <= a[2] && b[2] >= a[0] && b[1] <= a[3] && b[3] >= a[1]) ov_idx.push_back(j);
        }
        ov_off[k + 1] = (int32_t)ov_idx.size();
        CropArgs c;
        c.x0 = a[0]; c.y0 = a[1]; c.w = a[2] - a[0] + 1; c.h = a[3] - a[1] + 1; c.k = k; c.eig_off = eig_total;
        eig_total += (size_t)eig_tiles(c);
        crops[k] = c;
    }
    const int n_ov = (int)ov_idx.size();
    g_flow_sub[0] += fm_now_ms() - tp0; tp0 = fm_now_ms();
    // (no synchronisation here: the staging block was consumed by the previous call, which ended with one, and the
    // kernels of fm_flow_begin that may still be running touch none of the buffers below -- only a reallocation
    // needs an idle stream)
    if (eig_total > f->eig_cap || nT > f->rect_cap || nT + 1 > f->ov_cap || n_ov > f->ov_cap * 8)
        FM_HIP(hipStreamSynchronize(s));
    g_flow_sub[1] += fm_now_ms() - tp0; tp0 = fm_now_ms();
    if (eig_total > f->eig_cap) {
        f->eig_cap = 0;
        FM_HIP(hipFree(f->eig));
        f->eig = nullptr;
        FM_HIP(hipFree(f->tile_stat));
        f->tile_stat = nullptr;
        FM_HIP(hipMalloc(&f->eig, sizeof(unsigned long long) * EIG_TPX * eig_total * 2));
        FM_HIP(hipMalloc(&f->tile_stat, sizeof(uint2) * eig_total * 2));
        f->eig_cap = eig_total * 2;
    }
    if (nT > f->rect_cap) {
        if (f->rects) FM_HIP(hipFree(f->rects));
        f->rects = nullptr;
        int cap = f->rect_cap ? f->rect_cap : 64;
        while (cap < nT) cap *= 2;
        FM_HIP(hipMalloc(&f->rects, sizeof(int32_t) * 4 * cap));
        f->rect_cap = cap;
    }
    if (nT + 1 > f->ov_cap || n_ov > f->ov_cap * 8) {
        if (f->ov_idx) FM_HIP(hipFree(f->ov_idx));
        if (f->ov_off) FM_HIP(hipFree(f->ov_off));
        f->ov_idx = f->ov_off = nullptr;
        int cap = f->ov_cap ? f->ov_cap : 64;
        while (cap < nT + 1 || cap * 8 < n_ov) cap *= 2;
        FM_HIP(hipMalloc(&f->ov_off, sizeof(int32_t) * cap));
        FM_HIP(hipMalloc(&f->ov_idx, sizeof(int32_t) * cap * 8));
        f->ov_cap = cap;
    }
    // ---- packed upload
    auto al = [](size_t v) { return (v + 15) & ~size_t(15); };
    const size_t o_rect = 0, o_kpoff = al(o_rect + sizeof(int32_t) * 4 * nT), o_kps = al(o_kpoff + sizeof(int32_t) * (nT + 1));
    const size_t o_ovoff = al(o_kps + sizeof(float) * 2 * nk), o_ovidx = al(o_ovoff + sizeof(int32_t) * (nT + 1));
    const size_t o_crop = al(o_ovidx + sizeof(int32_t) * n_ov), o_box = al(o_crop + sizeof(CropArgs) * nT);
    const size_t o_tot = al(o_box + sizeof(double) * 4 * nT);          // two zeroed counters, uploaded with the rest
    const size_t in_bytes = o_tot + 16;
    // device-side scratch behind the upload: needy flags + min distances (read by the eig / select kernels)
    const size_t o_dneedy = in_bytes, o_dmd = al(o_dneedy + nT);
    int rc = f->tgt_in.reserve(al(o_dmd + sizeof(int32_t) * nT));
    if (rc) return rc;
    // outputs (device-visible pinned host memory): area | md | counts | off | total,n_bg | needy | keep | pts | bg
    const size_t q_area = 0, q_md = al(q_area + 4 * (size_t)nT), q_cnt = al(q_md + 4 * (size_t)nT);
    const size_t q_off = al(q_cnt + 4 * (size_t)nT), q_tot = al(q_off + 4 * (size_t)nT), q_needy = al(q_tot + 16);
    const size_t q_keep = al(q_needy + nT), q_pts = al(q_keep + nk), q_bg = al(q_pts + sizeof(float) * 2 * pts_cap);
    const size_t out_bytes = al(q_bg + sizeof(float) * 2 * bg_cap);
    if ((rc = f->tgt_out.reserve(out_bytes))) return rc;
    char* hb = f->tgt_in.host<char>();
    char* db = f->tgt_in.dev<char>();
    if (nT) {
        memcpy(hb + o_rect, irect.data(), sizeof(int32_t) * 4 * nT);
        memcpy(hb + o_kpoff, kp_off, sizeof(int32_t) * (nT + 1));
        if (nk) memcpy(hb + o_kps, kps, sizeof(float) * 2 * nk);
        memcpy(hb + o_ovoff, ov_off.data(), sizeof(int32_t) * (nT + 1));
        if (n_ov) memcpy(hb + o_ovidx, ov_idx.data(), sizeof(int32_t) * n_ov);
        memcpy(hb + o_crop, crops.data(), sizeof(CropArgs) * nT);
        memcpy(hb + o_box, full_tlbr, sizeof(double) * 4 * nT);
    }
    memset(hb + o_tot, 0, 16);
    // ONE upload; the kernels read rects / overlap lists / counters straight from it (no device-to-device
    // copies, no memset)
    FM_HIP(hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, s));
    FM_HIP(hipEventRecord(ctx->ev_prep, s));           // the background branch (side stream) needs the rects only
    f->v_rects = reinterpret_cast<const int32_t*>(db + o_rect);
    f->v_ov_off = reinterpret_cast<const int32_t*>(db + o_ovoff);
    f->v_ov_idx = reinterpret_cast<const int32_t*>(db + o_ovidx);
    // results: written by the kernels straight into the pinned, device-mapped block (a device block + one D2H copy
    // measured the same; the consumers of needy / min-distance read device copies either way)
    char* ho = f->tgt_out.host<char>();      // pinned, device accessible
    char* dbo = ho;
    int32_t* tot_host = reinterpret_cast<int32_t*>(ho + q_tot);
    int32_t* tot = reinterpret_cast<int32_t*>(db + o_tot);             // device counters: [0] new pts, [1] bg pts
    if (nT) {
        const Overlaps ov{f->v_ov_idx, f->v_ov_off};
        // needy flags / min distances are consumed by the eig / select kernels: device copies (behind the upload)
        uint8_t* d_needy = reinterpret_cast<uint8_t*>(db + o_dneedy);
        int32_t* d_md = reinterpret_cast<int32_t*>(db + o_dmd);
        fm_trace_mark(ctx, s, 44);
        hipLaunchKernelGGL(prepare_kernel, dim3(nT), dim3(PREP_BLK), 0, s, f->v_rects, ov,
                           reinterpret_cast<const float*>(db + o_kps), reinterpret_cast<const int32_t*>(db + o_kpoff),
                           feat_density, feat_dist_factor, reinterpret_cast<int32_t*>(dbo + q_area),
                           reinterpret_cast<uint8_t*>(dbo + q_keep), d_needy, d_md,
                           reinterpret_cast<uint8_t*>(dbo + q_needy));
        launch_eig(s, f->gray[f->prev], f->W, reinterpret_cast<const CropArgs*>(db + o_crop), nT, eig_max_tiles(crops),
                   f->v_rects, ov, f->cfg.block_size, d_needy, f->tile_stat, f->eig);
        hipLaunchKernelGGL(gftt_select_kernel, dim3(nT), dim3(GFTT_BLK), 0, s,
                           reinterpret_cast<const CropArgs*>(db + o_crop), f->tile_stat, f->eig,
                           (float)f->cfg.quality_level, f->cfg.max_corners, d_md,
                           reinterpret_cast<const double*>(db + o_box), reinterpret_cast<float*>(ho + q_pts), pts_cap,
                           reinterpret_cast<int32_t*>(dbo + q_cnt), d_needy, tot, reinterpret_cast<int32_t*>(dbo + q_off));
        fm_trace_mark(ctx, s, 45);
    }
    // background keypoints under the final mask: four small dependent launches that share nothing with the per-track
    // branch above except the uploaded rects -- they run on the side stream (behind the new frame's pyramid, which is
    // shorter than the per-track branch) and join at the end of the call
    hipStream_t sb = ctx->s_flow2;
    const int bw = f->cfg.bg_w, bh = f->cfg.bg_h;
    fm_trace_mark(ctx, sb, 46);
    // (the fork event is recorded behind the upload only: the branch must not wait for the per-track kernels)
    hipLaunchKernelGGL(resize_linear_kernel, dim3((bw + 255) / 256, bh), dim3(256), 0, sb, f->gray[f->prev], f->W,
                       f->H, f->bg_img, bw, bh);
    hipLaunchKernelGGL(fast_score_kernel, dim3((bw + 63) / 64, bh), dim3(64), 0, sb, f->bg_img, bw, bh,
                       f->cfg.fast_thresh, f->bg_flags);
    const FastBufs fb = fast_bufs(f->bg_flags, bw, bh);
    FM_HIP(hipStreamWaitEvent(sb, ctx->ev_prep, 0));
    hipLaunchKernelGGL(fast_flag_kernel, dim3(fb.nseg), dim3(FAST_SEG), 0, sb, fb.score, bw, bh, f->v_rects, nT, f->W,
                       f->H, fb.flag, fb.seg_cnt);
    // the last kernels of the two branches put the totals into the result block
    hipLaunchKernelGGL(fast_compact_kernel, dim3(fb.nseg), dim3(FAST_SEG), 0, sb, fb.flag, bw, bh, fb.seg_cnt,
                       reinterpret_cast<float*>(ho + q_bg), bg_cap, tot + 1, nullptr,
                       reinterpret_cast<int32_t*>(dbo + q_tot));
    fm_trace_mark(ctx, sb, 47);
    FM_HIP(hipEventRecord(ctx->ev_bg, sb));
    hipLaunchKernelGGL(copy_total_kernel, dim3(1), dim3(1), 0, s, tot, reinterpret_cast<int32_t*>(dbo + q_tot));
    FM_HIP(hipGetLastError());
    g_flow_sub[2] += fm_now_ms() - tp0; tp0 = fm_now_ms();
    FM_HIP(hipStreamSynchronize(s));
    FM_HIP(hipEventSynchronize(ctx->ev_bg));
    g_flow_sub[3] += fm_now_ms() - tp0;
    if (nT) {
        memcpy(area_out, ho + q_area, 4 * (size_t)nT);
        memcpy(needy_out, ho + q_needy, nT);
        if (nk) memcpy(keep_out, ho + q_keep, nk);
        memcpy(new_cnt_out, ho + q_cnt, 4 * (size_t)nT);
        memcpy(new_off_out, ho + q_off, 4 * (size_t)nT);
    }
    const int n_new = tot_host[0], n_bg = tot_host[1];
    if (n_new > pts_cap || n_bg > bg_cap) {
        fm_set_error("keypoint capacity exceeded (%d/%d new, %d/%d background)", n_new, pts_cap, n_bg, bg_cap);
        return FM_ERR_ARG;
    }
    if (n_new) memcpy(new_pts_out, ho + q_pts, sizeof(float) * 2 * n_new);
    if (n_bg) memcpy(bg_pts_out, ho + q_bg, sizeof(float) * 2 * n_bg);
    *n_new_out = n_new;
    *n_bg_out = n_bg;
    return 0;
}

#ifdef FM_GFTT_TIMING
extern "C" int fm_debug_gftt_stamps(long long* out512) {
    FM_HIP(hipDeviceSynchronize());
    FM_HIP(hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_gftt_stamps), sizeof(long long) * 512));
    return 0;
}
#endif

#ifdef FM_DIAG
// ---------------------------------------------------------------------------------------------------
// Diagnostic kernel (scripts/stress_spin.py): a long, fully deterministic computation on the flow stream,
// used to tell a bug in a kernel of this library from a platform issue when kernels on other streams run
// concurrently.  mode bit 0: 32-lane butterfly (ds_bpermute) every iteration; bit 1: byte loads from `img`.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void spin_kernel(int iters, int mode, const uint8_t* __restrict__ img, int img_n,
                                                   unsigned* __restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned x = 2463534242u ^ (unsigned)gid;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        if (mode & 2) x += img[(x >> 8) % (unsigned)img_n];
        if (mode & 1) {
            float v = (float)(x & 0xffff);
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
            acc += v * 1e-6f;
            if ((x & 1023u) == 0u && (gid & 32)) break;      // the two 32-lane halves of a wave diverge
        }
    }
    out[gid] = x ^ __float_as_uint(acc);
}
}  // namespace

extern "C" int fm_debug_spin(fm_ctx* ctx, int blocks, int iters, int mode, unsigned* out_host) {
    FM_CHECK_ARG(ctx && ctx->flow && blocks > 0 && out_host);
    FlowState* f = ctx->flow;
    hipStream_t s;
    { int rc_s_ = flow_stream(ctx, &s); if (rc_s_) return rc_s_; }
    const size_t bytes = sizeof(unsigned) * 256 * (size_t)blocks;
    int rc = f->lk_out.reserve(bytes);
    if (rc) return rc;
    FM_HIP(hipStreamSynchronize(s));
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s, iters, mode, f->gray[f->prev], f->W * f->H,
                       f->lk_out.dev<unsigned>());
    FM_HIP(hipGetLastError());
    FM_HIP(hipMemcpyAsync(f->lk_out.h, f->lk_out.d, bytes, hipMemcpyDeviceToHost, s));
    FM_HIP(hipStreamSynchronize(s));
    memcpy(out_host, f->lk_out.h, bytes);
    return 0;
}
#endif  // FM_DIAG

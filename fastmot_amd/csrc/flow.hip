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
#include <cmath>

namespace {
constexpr int MAX_LEVELS = 8;
}

struct FlowState {
    fm_flow_cfg cfg{};
    int W = 0, H = 0;
    int levels = 0;                       // pyramid levels actually built (maxLevel + 1)
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    uint8_t* gray[2] = {nullptr, nullptr};          // full resolution
    uint8_t* pyr[2][MAX_LEVELS] = {};               // level 0 = optical-flow frame
    int16_t* deriv[MAX_LEVELS] = {};                // Scharr of the previous pyramid (dx,dy interleaved)
    uint8_t* bg_img = nullptr;
    int prev = 0;                                   // index of the "previous" set
    // targets of the current predict
    int nT = 0, rect_cap = 0;
    int32_t* rects = nullptr;                       // [nT][4] inclusive integer rects
    DevBuf tgt_in, tgt_out, det_in, det_out, lk_in, lk_out, bg_out;
    float* eig = nullptr;                           // scratch for GFTT
    size_t eig_cap = 0;
    float* cand = nullptr;                          // [tracks][cand_cap][2] (value, raster index as float bits)
    int cand_tracks = 0;
    int32_t* bg_flags = nullptr;                    // FAST score / flags
};

void fm_flow_free(FlowState* f) {
    if (!f) return;
    for (int s = 0; s < 2; ++s) {
        if (f->gray[s]) (void)hipFree(f->gray[s]);
        for (int l = 0; l < MAX_LEVELS; ++l)
            if (f->pyr[s][l]) (void)hipFree(f->pyr[s][l]);
    }
    for (int l = 0; l < MAX_LEVELS; ++l)
        if (f->deriv[l]) (void)hipFree(f->deriv[l]);
    for (void* p : {(void*)f->bg_img, (void*)f->rects, (void*)f->eig, (void*)f->cand, (void*)f->bg_flags})
        if (p) (void)hipFree(p);
    for (DevBuf* b : {&f->tgt_in, &f->tgt_out, &f->det_in, &f->det_out, &f->lk_in, &f->lk_out, &f->bg_out})
        b->release();
    delete f;
}

namespace {

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- cvtColor BGR2GRAY, 8 bit (OpenCV 4.x RGB2Gray<uchar>: 15-bit fixed point)
__global__ void gray_kernel(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = bgr + (size_t)i * 3;
    gray[i] = (uint8_t)((p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15);
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
constexpr int LK_MAX_WIN = 7;

__global__ void lk_kernel(LKArgs a, int n, const float* __restrict__ prev_pts, float* __restrict__ next_pts,
                          uint8_t* __restrict__ status, float* __restrict__ err) {
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const int win = a.win;
    const float half = (win - 1) * 0.5f;
    const float px0 = prev_pts[2 * pt], py0 = prev_pts[2 * pt + 1];
    float nx = 0.f, ny = 0.f;
    bool st = true;
    float er = 0.f;
    short Ipatch[LK_MAX_WIN * LK_MAX_WIN], dIx[LK_MAX_WIN * LK_MAX_WIN], dIy[LK_MAX_WIN * LK_MAX_WIN];
    const float FLT_SCALE = 1.f / (1 << 20);
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
        float fa = ppx - ipx, fb = ppy - ipy;
        int iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << 14));
        int iw01 = __float2int_rn(fa * (1.f - fb) * (1 << 14));
        int iw10 = __float2int_rn((1.f - fa) * fb * (1 << 14));
        int iw11 = (1 << 14) - iw00 - iw01 - iw10;
        float A11 = 0.f, A12 = 0.f, A22 = 0.f;
        for (int y = 0; y < win; ++y) {
            const int yy0 = ipy + y, yy1 = yy0 + 1;
            const uint8_t* r0 = I + (size_t)reflect101(yy0, h) * w;
            const uint8_t* r1 = I + (size_t)reflect101(yy1, h) * w;
            for (int x = 0; x < win; ++x) {
                const int xx0 = ipx + x, xx1 = xx0 + 1;
                const int c0 = reflect101(xx0, w), c1 = reflect101(xx1, w);
                const int ival = LK_DESCALE(r0[c0] * iw00 + r0[c1] * iw01 + r1[c0] * iw10 + r1[c1] * iw11, 14 - 5);
                // derivative image is zero outside (BORDER_CONSTANT), lkpyramid.cpp
                auto dv = [&](int xx, int yy, int ch) -> int {
                    return (xx < 0 || xx >= w || yy < 0 || yy >= h) ? 0 : (int)D[((size_t)yy * w + xx) * 2 + ch];
                };
                const int ixval = LK_DESCALE(dv(xx0, yy0, 0) * iw00 + dv(xx1, yy0, 0) * iw01 +
                                             dv(xx0, yy1, 0) * iw10 + dv(xx1, yy1, 0) * iw11, 14);
                const int iyval = LK_DESCALE(dv(xx0, yy0, 1) * iw00 + dv(xx1, yy0, 1) * iw01 +
                                             dv(xx0, yy1, 1) * iw10 + dv(xx1, yy1, 1) * iw11, 14);
                Ipatch[y * win + x] = (short)ival;
                dIx[y * win + x] = (short)ixval;
                dIy[y * win + x] = (short)iyval;
                A11 += (float)(ixval * ixval);
                A12 += (float)(ixval * iyval);
                A22 += (float)(iyval * iyval);
            }
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
        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= w || iny < -win || iny >= h) {
                if (level == 0) st = false;
                break;
            }
            fa = nx - inx; fb = ny - iny;
            iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << 14));
            iw01 = __float2int_rn(fa * (1.f - fb) * (1 << 14));
            iw10 = __float2int_rn((1.f - fa) * fb * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            float b1 = 0.f, b2 = 0.f;
            for (int y = 0; y < win; ++y) {
                const uint8_t* r0 = J + (size_t)reflect101(iny + y, h) * w;
                const uint8_t* r1 = J + (size_t)reflect101(iny + y + 1, h) * w;
                for (int x = 0; x < win; ++x) {
                    const int c0 = reflect101(inx + x, w), c1 = reflect101(inx + x + 1, w);
                    const int diff = LK_DESCALE(r0[c0] * iw00 + r0[c1] * iw01 + r1[c0] * iw10 + r1[c1] * iw11, 14 - 5) -
                                     Ipatch[y * win + x];
                    b1 += (float)(diff * dIx[y * win + x]);
                    b2 += (float)(diff * dIy[y * win + x]);
                }
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
            fa = ex - inx; fb = ey - iny;
            iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << 14));
            iw01 = __float2int_rn(fa * (1.f - fb) * (1 << 14));
            iw10 = __float2int_rn((1.f - fa) * fb * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            float errval = 0.f;
            for (int y = 0; y < win; ++y) {
                const uint8_t* r0 = J + (size_t)reflect101(iny + y, h) * w;
                const uint8_t* r1 = J + (size_t)reflect101(iny + y + 1, h) * w;
                for (int x = 0; x < win; ++x) {
                    const int c0 = reflect101(inx + x, w), c1 = reflect101(inx + x + 1, w);
                    const int diff = LK_DESCALE(r0[c0] * iw00 + r0[c1] * iw01 + r1[c0] * iw10 + r1[c1] * iw11, 14 - 5) -
                                     Ipatch[y * win + x];
                    errval += fabsf((float)diff);
                }
            }
            er = errval * 1.f / (32 * win * win);
        }
    }
    next_pts[2 * pt] = nx;
    next_pts[2 * pt + 1] = ny;
    status[pt] = st ? 1 : 0;
    err[pt] = er;
}

// ---- foreground-mask bookkeeping: rect k sees pixel p as foreground iff no rect j<k covers p
__device__ __forceinline__ bool covered_before(const int32_t* rects, int k, int x, int y) {
    for (int j = 0; j < k; ++j) {
        const int32_t* r = rects + 4 * j;
        if (x >= r[0] && x <= r[2] && y >= r[1] && y <= r[3]) return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void target_area_kernel(const int32_t* __restrict__ rects, int nT,
                                                          int32_t* __restrict__ area) {
    const int k = blockIdx.x;
    const int32_t* r = rects + 4 * k;
    const int w = r[2] - r[0] + 1, h = r[3] - r[1] + 1;
    int cnt = 0;
    for (int i = threadIdx.x; i < w * h; i += 256) {
        const int x = r[0] + i % w, y = r[1] + i / w;
        cnt += covered_before(rects, k, x, y) ? 0 : 1;
    }
    __shared__ int red[256];
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) area[k] = red[0];
}

// _rect_filter (flow.py:282-295): rounded point inside rect k and still foreground
__global__ void kp_filter_kernel(const int32_t* __restrict__ rects, const float* __restrict__ kps,
                                 const int32_t* __restrict__ kp_track, int n, uint8_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = kp_track[i];
    const int32_t* r = rects + 4 * k;
    const int x = (int)rintf(kps[2 * i]), y = (int)rintf(kps[2 * i + 1]);
    const bool inside = x >= r[0] && x <= r[2] && y >= r[1] && y <= r[3];
    keep[i] = (inside && !covered_before(rects, k, x, y)) ? 1 : 0;
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

__global__ void eig_kernel(const uint8_t* __restrict__ img, int stride, const CropArgs* __restrict__ crops,
                           float* __restrict__ eig, int block_size) {
    const CropArgs c = crops[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.w * c.h) return;
    const int x = i % c.w, y = i / c.w;
    const float scale = 1.f / (4.f * block_size * 255.f);
    float sxx = 0.f, sxy = 0.f, syy = 0.f;
    const int r = block_size / 2;
    for (int j = -r; j <= r; ++j) {
        float rxx = 0.f, rxy = 0.f, ryy = 0.f;
        for (int ii = -r; ii <= r; ++ii) {
            float dx, dy;
            sobel_at(img, stride, c, reflect101(x + ii, c.w), reflect101(y + j, c.h), scale, dx, dy);
            rxx += dx * dx; rxy += dx * dy; ryy += dy * dy;
        }
        sxx += rxx; sxy += rxy; syy += ryy;
    }
    const float a = sxx * 0.5f, b = sxy, cc = syy * 0.5f;
    eig[c.eig_off + i] = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
}

// one block per needy track: masked max -> threshold -> 3x3 local maxima -> sort -> min-distance
// selection -> ellipse filter
__global__ __launch_bounds__(256) void gftt_select_kernel(const CropArgs* __restrict__ crops,
                                                          const int32_t* __restrict__ rects,
                                                          const float* __restrict__ eig, float quality,
                                                          int max_corners, const int32_t* __restrict__ min_dist,
                                                          const double* __restrict__ full_tlbr,
                                                          float* __restrict__ cand, int cand_cap,
                                                          float* __restrict__ pts_out, int cap,
                                                          int32_t* __restrict__ counts) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const CropArgs c = crops[t];
    const float* e = eig + c.eig_off;
    __shared__ float red[256];
    __shared__ int s_n;
    // masked maximum (minMaxLoc with mask)
    float mx = 0.f;
    for (int i = tid; i < c.w * c.h; i += 256) {
        const int x = i % c.w, y = i / c.w;
        if (!covered_before(rects, c.k, c.x0 + x, c.y0 + y)) mx = fmaxf(mx, e[i]);
    }
    red[tid] = mx;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
        __syncthreads();
    }
    const float thr = red[0] * quality;
    __syncthreads();
    // candidates: val > thr, val >= 8 neighbours (dilate inside the crop), mask set, 1-px border skipped
    float* cval = cand + (size_t)t * cand_cap * 4;                       // [cand_cap] values
    int* cidx = reinterpret_cast<int*>(cval + cand_cap);                  // [cand_cap] raster indices
    int* order = reinterpret_cast<int*>(cval + 2 * (size_t)cand_cap);     // [cand_cap] sorted raster indices
    for (int i = tid; i < c.w * c.h; i += 256) {
        const int x = i % c.w, y = i / c.w;
        if (x < 1 || y < 1 || x >= c.w - 1 || y >= c.h - 1) continue;
        const float v = e[i];
        if (!(v > thr) || v == 0.f) continue;
        bool is_max = true;
        for (int j = -1; j <= 1 && is_max; ++j)
            for (int ii = -1; ii <= 1; ++ii)
                if (e[(y + j) * c.w + x + ii] > v) { is_max = false; break; }
        if (!is_max || covered_before(rects, c.k, c.x0 + x, c.y0 + y)) continue;
        const int slot = atomicAdd(&s_n, 1);
        if (slot < cand_cap) {
            cval[slot] = v;
            cidx[slot] = i;
        }
    }
    __syncthreads();
    const int n = min(s_n, cand_cap);
    // rank sort: value descending, ties by larger raster index first (greaterThanPtr)
    for (int i = tid; i < n; i += 256) {
        const float vi = cval[i];
        const int ri = cidx[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float vj = cval[j];
            const int rj = cidx[j];
            rank += (vj > vi || (vj == vi && rj > ri)) ? 1 : 0;
        }
        order[rank] = ri;
    }
    __syncthreads();
    // greedy min-distance selection by the first wavefront (accepted points in LDS)
    __shared__ short acc_x[1024], acc_y[1024];
    __shared__ int s_acc;
    if (tid == 0) s_acc = 0;
    __syncthreads();
    if (tid < 64) {
        const int md = min_dist[t];
        const int md2 = md * md;
        const int limit = min(max_corners, 1024);
        int nacc = 0;
        for (int q = 0; q < n && nacc < limit; ++q) {
            const int ri = order[q];
            const int x = ri % c.w, y = ri / c.w;
            bool bad = false;
            for (int j = tid; j < nacc; j += 64) {
                const int dx = x - acc_x[j], dy = y - acc_y[j];
                if (dx * dx + dy * dy < md2) bad = true;
            }
            bad = __any(bad);
            if (!bad) {
                if (tid == 0) { acc_x[nacc] = (short)x; acc_y[nacc] = (short)y; }
                ++nacc;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (tid == 0) s_acc = nacc;
    }
    __syncthreads();
    // _ellipse_filter (flow.py:297-306) in float32/float64 mixed like numpy: pts f32 + offset f32
    if (tid == 0) {
        const double* b = full_tlbr + 4 * t;
        const double cx = (b[0] + b[2]) / 2, cy = (b[1] + b[3]) / 2;
        const double ax = (b[2] - b[0] + 1) * 0.5, ay = (b[3] - b[1] + 1) * 0.5;
        int m = 0;
        for (int q = 0; q < s_acc && m < cap; ++q) {
            const float px = (float)acc_x[q] + (float)c.x0, py = (float)acc_y[q] + (float)c.y0;
            const double ux = ((double)px - cx) / ax, uy = ((double)py - cy) / ay;
            if (ux * ux + uy * uy <= 1.) {
                pts_out[((size_t)t * cap + m) * 2] = px;
                pts_out[((size_t)t * cap + m) * 2 + 1] = py;
                ++m;
            }
        }
        counts[t] = m;
    }
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

// NMS + mask + raster-order compaction in ONE block (the background image is ~20 k pixels)
__global__ __launch_bounds__(1024) void fast_collect_kernel(const int32_t* __restrict__ score, int w, int h,
                                                            const int32_t* __restrict__ rects, int nT,
                                                            int full_w, int full_h, float* __restrict__ pts,
                                                            int cap, int32_t* __restrict__ n_out) {
    __shared__ int s_cnt[1024];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int total = w * h;
    for (int start = 0; start < total; start += 1024) {
        const int i = start + tid;
        bool kp = false;
        int x = 0, y = 0;
        if (i < total) {
            x = i % w; y = i / w;
            const int s = score[i];
            if (s > 0 && x >= 3 && y >= 3 && x < w - 3 && y < h - 3) {
                kp = s > score[i - 1] && s > score[i + 1] && s > score[i - w - 1] && s > score[i - w] &&
                     s > score[i - w + 1] && s > score[i + w - 1] && s > score[i + w] && s > score[i + w + 1];
                if (kp) {
                    // INTER_NEAREST sample of the full-resolution foreground mask at the keypoint
                    const int fx = min((int)floor((double)x * ((double)full_w / w)), full_w - 1);
                    const int fy = min((int)floor((double)y * ((double)full_h / h)), full_h - 1);
                    if (covered_before(rects, nT, fx, fy)) kp = false;
                }
            }
        }
        s_cnt[tid] = kp ? 1 : 0;
        __syncthreads();
        // inclusive scan (Hillis-Steele)
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = tid >= off ? s_cnt[tid - off] : 0;
            __syncthreads();
            s_cnt[tid] += v;
            __syncthreads();
        }
        const int pos = s_base + s_cnt[tid] - 1;
        if (kp && pos < cap) {
            pts[2 * pos] = (float)x;
            pts[2 * pos + 1] = (float)y;
        }
        __syncthreads();
        if (tid == 1023) s_base += s_cnt[1023];
        __syncthreads();
    }
    if (tid == 0) *n_out = s_base;
}

int build_pyramid(fm_ctx* ctx, FlowState* f, int set) {
    hipStream_t s = ctx->s_flow;
    const int n = ctx->frame_w * ctx->frame_h;
    hipLaunchKernelGGL(gray_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ctx->frame_cur, f->gray[set], n);
    hipLaunchKernelGGL(resize_linear_kernel, dim3((f->lw[0] + 255) / 256, f->lh[0]), dim3(256), 0, s,
                       f->gray[set], f->W, f->H, f->pyr[set][0], f->lw[0], f->lh[0]);
    for (int l = 1; l < f->levels; ++l)
        hipLaunchKernelGGL(pyrdown_kernel, dim3((f->lw[l] + 255) / 256, f->lh[l]), dim3(256), 0, s,
                           f->pyr[set][l - 1], f->lw[l - 1], f->lh[l - 1], f->pyr[set][l], f->lw[l], f->lh[l]);
    FM_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" int fm_flow_configure(fm_ctx* ctx, const fm_flow_cfg* cfg) {
    FM_CHECK_ARG(ctx && cfg && ctx->frame_w > 0);
    FM_CHECK_ARG(cfg->win_size >= 3 && cfg->win_size <= LK_MAX_WIN && cfg->max_level >= 0 && cfg->max_level < MAX_LEVELS);
    FM_CHECK_ARG(cfg->block_size == 3 || cfg->block_size == 5);
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
    for (int l = 0; l < f->levels; ++l) FM_HIP(hipMalloc(&f->deriv[l], (size_t)f->lw[l] * f->lh[l] * 4));
    FM_HIP(hipMalloc(&f->bg_img, (size_t)cfg->bg_w * cfg->bg_h));
    FM_HIP(hipMalloc(&f->bg_flags, sizeof(int32_t) * ((size_t)cfg->bg_w * cfg->bg_h + 4)));
    f->eig_cap = (size_t)4 * f->W * f->H;
    FM_HIP(hipMalloc(&f->eig, sizeof(float) * f->eig_cap));
    return 0;
}

extern "C" int fm_flow_init(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->flow && ctx->frame_cur);
    FlowState* f = ctx->flow;
    f->prev = 0;
    int rc = build_pyramid(ctx, f, 0);
    if (rc) return rc;
    f->nT = 0;
    return 0;
}

extern "C" int fm_flow_begin(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->flow && ctx->frame_cur);
    return build_pyramid(ctx, ctx->flow, ctx->flow->prev ^ 1);
}

extern "C" int fm_flow_swap(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->flow);
    ctx->flow->prev ^= 1;
    return 0;
}

extern "C" int fm_flow_targets(fm_ctx* ctx, int nT, const double* inside_tlbr, const float* kps,
                               const int32_t* kp_off, int32_t* area_out, uint8_t* keep_out) {
    FM_CHECK_ARG(ctx && ctx->flow && nT >= 0);
    FlowState* f = ctx->flow;
    hipStream_t s = ctx->s_flow;
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
    // packed upload: rects i32[nT][4] | kp_track i32[nk] | kps f32[nk][2]
    const size_t o_rect = 0, o_trk = sizeof(int32_t) * 4 * nT, o_kps = o_trk + sizeof(int32_t) * nk;
    const size_t in_bytes = o_kps + sizeof(float) * 2 * nk;
    int rc = f->tgt_in.reserve(in_bytes + 16);
    if (rc) return rc;
    if ((rc = f->tgt_out.reserve(sizeof(int32_t) * nT + nk + 16))) return rc;
    FM_HIP(hipStreamSynchronize(s));
    char* hbuf = f->tgt_in.host<char>();
    int32_t* hr = reinterpret_cast<int32_t*>(hbuf + o_rect);
    for (int k = 0; k < nT; ++k)
        for (int e = 0; e < 4; ++e) hr[4 * k + e] = (int32_t)inside_tlbr[4 * k + e];   // crop(): int() truncation
    int32_t* ht = reinterpret_cast<int32_t*>(hbuf + o_trk);
    for (int k = 0; k < nT; ++k)
        for (int i = kp_off[k]; i < kp_off[k + 1]; ++i) ht[i] = k;
    if (nk) memcpy(hbuf + o_kps, kps, sizeof(float) * 2 * nk);
    FM_HIP(hipMemcpyAsync(f->tgt_in.d, hbuf, in_bytes, hipMemcpyHostToDevice, s));
    FM_HIP(hipMemcpyAsync(f->rects, f->tgt_in.dev<char>() + o_rect, sizeof(int32_t) * 4 * nT, hipMemcpyDeviceToDevice, s));
    int32_t* d_area = f->tgt_out.dev<int32_t>();
    uint8_t* d_keep = reinterpret_cast<uint8_t*>(d_area + nT);
    hipLaunchKernelGGL(target_area_kernel, dim3(nT), dim3(256), 0, s, f->rects, nT, d_area);
    if (nk)
        hipLaunchKernelGGL(kp_filter_kernel, dim3((nk + 255) / 256), dim3(256), 0, s, f->rects,
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
    hipStream_t s = ctx->s_flow;
    FM_HIP(hipStreamSynchronize(s));
    const int32_t* hrects = reinterpret_cast<const int32_t*>(f->tgt_in.host<char>());
    std::vector<CropArgs> crops(n);
    size_t off = 0;
    int max_area = 0;
    for (int i = 0; i < n; ++i) {
        FM_CHECK_ARG(track_idx[i] >= 0 && track_idx[i] < f->nT);
        const int32_t* r = hrects + 4 * track_idx[i];
        CropArgs c;
        c.x0 = r[0]; c.y0 = r[1]; c.w = r[2] - r[0] + 1; c.h = r[3] - r[1] + 1; c.k = track_idx[i];
        c.eig_off = off;
        off += (size_t)c.w * c.h;
        max_area = std::max(max_area, c.w * c.h);
        crops[i] = c;
    }
    if (off > f->eig_cap) {
        fm_set_error("GFTT scratch too small (%zu > %zu px)", off, f->eig_cap);
        return FM_ERR_STATE;
    }
    const int cand_cap = 8192;   // per track: [cand_cap][2] values + order area
    if (n > f->cand_tracks) {
        if (f->cand) FM_HIP(hipFree(f->cand));
        f->cand = nullptr;
        FM_HIP(hipMalloc(&f->cand, sizeof(float) * (size_t)n * cand_cap * 4));   // vals | idx | order | spare
        f->cand_tracks = n;
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
    hipLaunchKernelGGL(eig_kernel, dim3((max_area + 255) / 256, n), dim3(256), 0, s, f->gray[f->prev], f->W,
                       reinterpret_cast<const CropArgs*>(db + o_crop), f->eig, f->cfg.block_size);
    float* d_pts = f->det_out.dev<float>();
    int32_t* d_cnt = reinterpret_cast<int32_t*>(d_pts + 2 * (size_t)n * cap);
    hipLaunchKernelGGL(gftt_select_kernel, dim3(n), dim3(256), 0, s, reinterpret_cast<const CropArgs*>(db + o_crop),
                       f->rects, f->eig, (float)f->cfg.quality_level, f->cfg.max_corners,
                       reinterpret_cast<const int32_t*>(db + o_md), reinterpret_cast<const double*>(db + o_box),
                       f->cand, cand_cap, d_pts, cap, d_cnt);
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
    hipStream_t s = ctx->s_flow;
    const int bw = f->cfg.bg_w, bh = f->cfg.bg_h;
    int rc = f->bg_out.reserve(sizeof(float) * 2 * cap + 16);
    if (rc) return rc;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((bw + 255) / 256, bh), dim3(256), 0, s, f->gray[f->prev], f->W,
                       f->H, f->bg_img, bw, bh);
    hipLaunchKernelGGL(fast_score_kernel, dim3((bw + 63) / 64, bh), dim3(64), 0, s, f->bg_img, bw, bh,
                       f->cfg.fast_thresh, f->bg_flags);
    int32_t* d_n = f->bg_flags + (size_t)bw * bh;
    hipLaunchKernelGGL(fast_collect_kernel, dim3(1), dim3(1024), 0, s, f->bg_flags, bw, bh, f->rects, f->nT, f->W,
                       f->H, f->bg_out.dev<float>(), cap, d_n);
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
    hipStream_t s = ctx->s_flow;
    if (n > 0) {
        FM_CHECK_ARG(prev_pts && next_pts && status && err);
        int rc = f->lk_in.reserve(sizeof(float) * 2 * n);
        if (rc) return rc;
        const size_t o_st = sizeof(float) * 2 * n, o_err = (o_st + n + 15) & ~size_t(15);
        const size_t out_bytes = o_err + sizeof(float) * n;
        if ((rc = f->lk_out.reserve(out_bytes))) return rc;
        FM_HIP(hipStreamSynchronize(s));
        memcpy(f->lk_in.h, prev_pts, sizeof(float) * 2 * n);
        FM_HIP(hipMemcpyAsync(f->lk_in.d, f->lk_in.h, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
        const int p = f->prev, c = p ^ 1;
        LKArgs a{};
        for (int l = 0; l < f->levels; ++l) {
            hipLaunchKernelGGL(scharr_kernel, dim3((f->lw[l] + 255) / 256, f->lh[l]), dim3(256), 0, s, f->pyr[p][l],
                               f->lw[l], f->lh[l], f->deriv[l]);
            a.I[l] = f->pyr[p][l];
            a.J[l] = f->pyr[c][l];
            a.D[l] = f->deriv[l];
            a.w[l] = f->lw[l];
            a.h[l] = f->lh[l];
        }
        a.levels = f->levels;
        a.win = f->cfg.win_size;
        a.max_count = std::min(std::max(f->cfg.max_count, 0), 100);
        const double eps = std::min(std::max(f->cfg.epsilon, 0.), 10.);
        a.eps2 = (float)(eps * eps);
        a.min_eig_thresh = 1e-4f;
        char* o = f->lk_out.dev<char>();
        hipLaunchKernelGGL(lk_kernel, dim3((n + 63) / 64), dim3(64), 0, s, a, n, f->lk_in.dev<float>(),
                           reinterpret_cast<float*>(o), reinterpret_cast<uint8_t*>(o + o_st),
                           reinterpret_cast<float*>(o + o_err));
        FM_HIP(hipGetLastError());
        FM_HIP(hipMemcpyAsync(f->lk_out.h, f->lk_out.d, out_bytes, hipMemcpyDeviceToHost, s));
        FM_HIP(hipStreamSynchronize(s));
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

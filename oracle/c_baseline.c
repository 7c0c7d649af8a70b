/* TEST INFRASTRUCTURE ONLY -- compiled (plain C, -O3 -march=x86-64-v3, single thread) restatement of the routines the
 * reference's CPU path runs as COMPILED code: the OpenCV calls of fastmot/flow.py (cvtColor :129,153, resize
 * :130,154,187, goodFeaturesToTrack :171-173, FastFeatureDetector :190, calcOpticalFlowPyrLK :205-207,
 * findHomography :220-224, estimateAffinePartial2D :243-247).  The Python orchestration around them (Flow.predict,
 * MultiTracker, Numba-jitted Kalman / distance helpers -> numpy in oracle/np_oracle.py) stays Python, as in the
 * reference.  Used by oracle/c_baseline.py: bench.py's `cpu_baseline_compiled` ("Numba-class proxy")
 * and tests/test_c_baseline.py, which pins every function here against oracle/cv_oracle.py (bit-exact for the
 * image / LK / corner routines, 1e-9 for the double-precision model fits).  Never linked into the product library.
 *
 * Algorithms restated from the published OpenCV sources (imgproc/color, resize, pyramids, corner, featureselect;
 * features2d/fast; video/lkpyramid; calib3d/ptsetreg, fundam, levmarq; core/rand) -- the same statements as
 * cv_oracle.py, which cites the details.  Compiled with -ffp-contract=off: float32 arithmetic must not be fused. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

/* ---------------------------------------------------------------- images */
/* bits: 14 (1868 / 9617 / 4899, OpenCV up to the 4.2 era) or 15 (3735 / 19235 / 9798); cv_oracle.bgr2gray */
void cb_bgr2gray(const uint8_t* bgr, int n, uint8_t* gray, int bits) {
    const int cb = bits == 15 ? 3735 : 1868, cg = bits == 15 ? 19235 : 9617, cr = bits == 15 ? 9798 : 4899;
    const int sh = bits == 15 ? 15 : 14;
    for (int i = 0; i < n; ++i)
        gray[i] = (uint8_t)((bgr[3 * i] * cb + bgr[3 * i + 1] * cg + bgr[3 * i + 2] * cr + (1 << (sh - 1))) >> sh);
}

static void lin_coef(int d, double scale, int ssize, int* s0, int* s1, int* a0, int* a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    *s0 = s;
    *s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    *a0 = (int)lrintf((1.f - f) * 2048.f);
    *a1 = (int)lrintf(f * 2048.f);
}

/* cv2.resize 8UC1 INTER_LINEAR (exact 2x decimation = INTER_AREA) */
void cb_resize(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    if (sw == 2 * dw && sh == 2 * dh) {
        for (int y = 0; y < dh; ++y) {
            const uint8_t* r0 = src + (size_t)(2 * y) * sw;
            const uint8_t* r1 = r0 + sw;
            for (int x = 0; x < dw; ++x)
                dst[(size_t)y * dw + x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
        }
        return;
    }
    int* xs = (int*)malloc(sizeof(int) * 4 * dw);
    for (int x = 0; x < dw; ++x) lin_coef(x, (double)sw / dw, sw, &xs[4 * x], &xs[4 * x + 1], &xs[4 * x + 2], &xs[4 * x + 3]);
    for (int y = 0; y < dh; ++y) {
        int y0, y1, b0, b1;
        lin_coef(y, (double)sh / dh, sh, &y0, &y1, &b0, &b1);
        const uint8_t* r0 = src + (size_t)y0 * sw;
        const uint8_t* r1 = src + (size_t)y1 * sw;
        for (int x = 0; x < dw; ++x) {
            const int* c = xs + 4 * x;
            const int S0 = r0[c[0]] * c[2] + r0[c[1]] * c[3], S1 = r1[c[0]] * c[2] + r1[c[1]] * c[3];
            int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)y * dw + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(xs);
}

void cb_resize_nearest(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    for (int y = 0; y < dh; ++y) {
        int sy = (int)floor(y * ((double)sh / dh));
        if (sy > sh - 1) sy = sh - 1;
        for (int x = 0; x < dw; ++x) {
            int sx = (int)floor(x * ((double)sw / dw));
            if (sx > sw - 1) sx = sw - 1;
            dst[(size_t)y * dw + x] = src[(size_t)sy * sw + sx];
        }
    }
}

void cb_pyr_down(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    static const int wk[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            int sum = 0;
            for (int j = 0; j < 5; ++j) {
                const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, sh) * sw;
                int rs = 0;
                for (int i = 0; i < 5; ++i) rs += wk[i] * row[reflect101(2 * x + i - 2, sw)];
                sum += wk[j] * rs;
            }
            dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
        }
}

void cb_scharr(const uint8_t* src, int w, int h, int16_t* d) {
    for (int y = 0; y < h; ++y) {
        const int y0 = y > 0 ? y - 1 : (h > 1 ? 1 : 0), y2 = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
        const uint8_t *r0 = src + (size_t)y0 * w, *r1 = src + (size_t)y * w, *r2 = src + (size_t)y2 * w;
        for (int x = 0; x < w; ++x) {
            const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
#define T0(c) ((r0[c] + r2[c]) * 3 + r1[c] * 10)
#define T1(c) (r2[c] - r0[c])
            d[((size_t)y * w + x) * 2] = (int16_t)(T0(xp) - T0(xm));
            d[((size_t)y * w + x) * 2 + 1] = (int16_t)((T1(xp) + T1(xm)) * 3 + T1(x) * 10);
#undef T0
#undef T1
        }
    }
}

/* ---------------------------------------------------------------- pyramidal LK (5x5 default, any odd win <= 15) */
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
typedef struct { const uint8_t* I; const uint8_t* J; const int16_t* D; int w, h; } cb_level;

void cb_lk(const cb_level* lv, int levels, int win, int max_count, float eps2, float min_eig, int n, const float* pts,
           float* out, uint8_t* status, float* err) {
    const float half = (win - 1) * 0.5f, FLT_SCALE = 1.f / (1 << 20);
    const int w2 = win * win;
    int Iv[225], Ix[225], Iy[225];
    for (int p = 0; p < n; ++p) {
        float nx = 0.f, ny = 0.f, er = 0.f;
        int st = 1;
        for (int level = levels - 1; level >= 0; --level) {
            const cb_level* L = lv + level;
            const int w = L->w, h = L->h;
            const float sc = 1.f / (float)(1 << level);
            float ppx = pts[2 * p] * sc, ppy = pts[2 * p + 1] * sc;
            if (level == levels - 1) { nx = ppx; ny = ppy; } else { nx *= 2.f; ny *= 2.f; }
            ppx -= half; ppy -= half;
            const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
            if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
                if (level == 0) { st = 0; er = 0.f; }
                continue;
            }
            float fa = ppx - (float)ipx, fb = ppy - (float)ipy;
            int iw00 = (int)lrintf((1.f - fa) * (1.f - fb) * (1 << 14)), iw01 = (int)lrintf(fa * (1.f - fb) * (1 << 14));
            int iw10 = (int)lrintf((1.f - fa) * fb * (1 << 14)), iw11 = (1 << 14) - iw00 - iw01 - iw10;
            float A11 = 0.f, A12 = 0.f, A22 = 0.f;
            for (int y = 0; y < win; ++y)
                for (int x = 0; x < win; ++x) {
                    const int xx0 = ipx + x, xx1 = xx0 + 1, yy0 = ipy + y, yy1 = yy0 + 1;
                    const uint8_t* r0 = L->I + (size_t)reflect101(yy0, h) * w;
                    const uint8_t* r1 = L->I + (size_t)reflect101(yy1, h) * w;
                    const int c0 = reflect101(xx0, w), c1 = reflect101(xx1, w);
                    const int ival = DESCALE(r0[c0] * iw00 + r0[c1] * iw01 + r1[c0] * iw10 + r1[c1] * iw11, 14 - 5);
                    int dx[4], dy[4];
                    const int xs[4] = {xx0, xx1, xx0, xx1}, ys[4] = {yy0, yy0, yy1, yy1};
                    for (int q = 0; q < 4; ++q) {
                        if (xs[q] < 0 || xs[q] >= w || ys[q] < 0 || ys[q] >= h) { dx[q] = dy[q] = 0; continue; }
                        const int16_t* dp = L->D + ((size_t)ys[q] * w + xs[q]) * 2;
                        dx[q] = dp[0]; dy[q] = dp[1];
                    }
                    const int ixv = DESCALE(dx[0] * iw00 + dx[1] * iw01 + dx[2] * iw10 + dx[3] * iw11, 14);
                    const int iyv = DESCALE(dy[0] * iw00 + dy[1] * iw01 + dy[2] * iw10 + dy[3] * iw11, 14);
                    Iv[y * win + x] = ival; Ix[y * win + x] = ixv; Iy[y * win + x] = iyv;
                    A11 += (float)(ixv * ixv); A12 += (float)(ixv * iyv); A22 += (float)(iyv * iyv);
                }
            A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
            float Dt = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if (minEig < min_eig || Dt < FLT_EPSILON) {
                if (level == 0) st = 0;
                continue;
            }
            Dt = 1.f / Dt;
            float cx = nx - half, cy = ny - half, pdx = 0.f, pdy = 0.f;
            float ox = cx + half, oy = cy + half;
            for (int j = 0; j < max_count; ++j) {
                const int inx = (int)floorf(cx), iny = (int)floorf(cy);
                if (inx < -win || inx >= w || iny < -win || iny >= h) {
                    if (level == 0) st = 0;
                    break;
                }
                fa = cx - (float)inx; fb = cy - (float)iny;
                iw00 = (int)lrintf((1.f - fa) * (1.f - fb) * (1 << 14)); iw01 = (int)lrintf(fa * (1.f - fb) * (1 << 14));
                iw10 = (int)lrintf((1.f - fa) * fb * (1 << 14)); iw11 = (1 << 14) - iw00 - iw01 - iw10;
                float b1 = 0.f, b2 = 0.f;
                for (int y = 0; y < win; ++y) {
                    const uint8_t* r0 = L->J + (size_t)reflect101(iny + y, h) * w;
                    const uint8_t* r1 = L->J + (size_t)reflect101(iny + y + 1, h) * w;
                    for (int x = 0; x < win; ++x) {
                        const int c0 = reflect101(inx + x, w), c1 = reflect101(inx + x + 1, w);
                        const int diff = DESCALE(r0[c0] * iw00 + r0[c1] * iw01 + r1[c0] * iw10 + r1[c1] * iw11, 14 - 5) -
                                         Iv[y * win + x];
                        b1 += (float)(diff * Ix[y * win + x]);
                        b2 += (float)(diff * Iy[y * win + x]);
                    }
                }
                b1 *= FLT_SCALE; b2 *= FLT_SCALE;
                const float dxv = (A12 * b2 - A22 * b1) * Dt, dyv = (A12 * b1 - A11 * b2) * Dt;
                cx += dxv; cy += dyv;
                ox = cx + half; oy = cy + half;
                if (dxv * dxv + dyv * dyv <= eps2) break;
                if (j > 0 && fabsf(dxv + pdx) < 0.01f && fabsf(dyv + pdy) < 0.01f) {
                    ox -= dxv * 0.5f; oy -= dyv * 0.5f;
                    break;
                }
                pdx = dxv; pdy = dyv;
            }
            nx = ox; ny = oy;
            if (st && level == 0) {
                const float ex = nx - half, ey = ny - half;
                const int inx = (int)floorf(ex), iny = (int)floorf(ey);
                if (inx < -win || inx >= w || iny < -win || iny >= h) { st = 0; continue; }
                fa = ex - (float)inx; fb = ey - (float)iny;
                iw00 = (int)lrintf((1.f - fa) * (1.f - fb) * (1 << 14)); iw01 = (int)lrintf(fa * (1.f - fb) * (1 << 14));
                iw10 = (int)lrintf((1.f - fa) * fb * (1 << 14)); iw11 = (1 << 14) - iw00 - iw01 - iw10;
                float e = 0.f;
                for (int y = 0; y < win; ++y) {
                    const uint8_t* r0 = L->J + (size_t)reflect101(iny + y, h) * w;
                    const uint8_t* r1 = L->J + (size_t)reflect101(iny + y + 1, h) * w;
                    for (int x = 0; x < win; ++x) {
                        const int c0 = reflect101(inx + x, w), c1 = reflect101(inx + x + 1, w);
                        const int diff = DESCALE(r0[c0] * iw00 + r0[c1] * iw01 + r1[c0] * iw10 + r1[c1] * iw11, 14 - 5) -
                                         Iv[y * win + x];
                        e += fabsf((float)diff);
                    }
                }
                er = e * 1.f / (float)(32 * win * win);
            }
        }
        (void)w2;
        out[2 * p] = nx; out[2 * p + 1] = ny;
        status[p] = (uint8_t)st;
        err[p] = er;
    }
}

/* ---------------------------------------------------------------- goodFeaturesToTrack on an isolated crop */
static void min_eig_map(const uint8_t* img, int stride, int w, int h, int block, float* eig) {
    const float scale = 1.f / (4.f * (float)block * 255.f);
    float* dx = (float*)malloc(sizeof(float) * 2 * (size_t)w * h);
    float* dy = dx + (size_t)w * h;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v[3][3];
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i)
                    v[j][i] = img[(size_t)reflect101(y + j - 1, h) * stride + reflect101(x + i - 1, w)];
            const int gx = (v[0][2] + 2 * v[1][2] + v[2][2]) - (v[0][0] + 2 * v[1][0] + v[2][0]);
            const int gy = (v[2][0] + 2 * v[2][1] + v[2][2]) - (v[0][0] + 2 * v[0][1] + v[0][2]);
            dx[(size_t)y * w + x] = (float)gx * scale;
            dy[(size_t)y * w + x] = (float)gy * scale;
        }
    const int r = block / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float sxx = 0.f, sxy = 0.f, syy = 0.f;
            for (int j = -r; j <= r; ++j) {
                float rxx = 0.f, rxy = 0.f, ryy = 0.f;
                const int yy = reflect101(y + j, h);
                for (int i = -r; i <= r; ++i) {
                    const int xx = reflect101(x + i, w);
                    const float a = dx[(size_t)yy * w + xx], b = dy[(size_t)yy * w + xx];
                    rxx += a * a; rxy += a * b; ryy += b * b;
                }
                sxx += rxx; sxy += rxy; syy += ryy;
            }
            const float a = sxx * 0.5f, b = sxy, c = syy * 0.5f;
            eig[(size_t)y * w + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        }
    free(dx);
}

typedef struct { float v; int idx; } cb_cand;
static int cand_cmp(const void* a, const void* b) {
    const cb_cand *x = (const cb_cand*)a, *y = (const cb_cand*)b;
    if (x->v != y->v) return x->v > y->v ? -1 : 1;
    return x->idx > y->idx ? -1 : (x->idx < y->idx ? 1 : 0);
}

/* img / mask: crops with row strides; -> up to max_corners (x, y) float pairs, strongest first */
int cb_gftt(const uint8_t* img, int img_stride, const uint8_t* mask, int mask_stride, int w, int h, int max_corners,
            float quality, int min_dist, int block, float* out_xy) {
    float* eig = (float*)malloc(sizeof(float) * (size_t)w * h);
    min_eig_map(img, img_stride, w, h, block, eig);
    float mx = 0.f;
    int any = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            if (mask[(size_t)y * mask_stride + x]) {
                const float v = eig[(size_t)y * w + x];
                if (!any || v > mx) mx = v;
                any = 1;
            }
    int n_out = 0;
    if (any) {
        const float thr = mx * quality;
        cb_cand* c = (cb_cand*)malloc(sizeof(cb_cand) * (size_t)w * h);
        int nc = 0;
        for (int y = 1; y < h - 1; ++y)
            for (int x = 1; x < w - 1; ++x) {
                const float v = eig[(size_t)y * w + x];
                if (!(v > thr) || v == 0.f || !mask[(size_t)y * mask_stride + x]) continue;
                int is_max = 1;
                for (int j = -1; j <= 1 && is_max; ++j)
                    for (int i = -1; i <= 1; ++i)
                        if (eig[(size_t)(y + j) * w + x + i] > v) { is_max = 0; break; }
                if (is_max) { c[nc].v = v; c[nc].idx = y * w + x; ++nc; }
            }
        qsort(c, nc, sizeof(cb_cand), cand_cmp);
        const int md2 = min_dist * min_dist;
        for (int q = 0; q < nc && n_out < max_corners; ++q) {
            const int y = c[q].idx / w, x = c[q].idx - y * w;
            int ok = 1;
            for (int k = 0; k < n_out; ++k) {
                const int ddx = x - (int)out_xy[2 * k], ddy = y - (int)out_xy[2 * k + 1];
                if (ddx * ddx + ddy * ddy < md2) { ok = 0; break; }
            }
            if (ok) { out_xy[2 * n_out] = (float)x; out_xy[2 * n_out + 1] = (float)y; ++n_out; }
        }
        free(c);
    }
    free(eig);
    return n_out;
}

/* ---------------------------------------------------------------- FAST-9/16 with non-maximum suppression */
int cb_fast(const uint8_t* img, int w, int h, int thr, float* out_xy, int cap) {
    static const int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    static const int DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    int* score = (int*)calloc((size_t)w * h, sizeof(int));
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            const int v = img[(size_t)y * w + x];
            int d[16];
            for (int k = 0; k < 16; ++k) d[k] = v - (int)img[(size_t)(y + DY[k]) * w + x + DX[k]];
            int best = 0;
            for (int s = 0; s < 16; ++s) {
                int mn = 255, mxn = 255;
                for (int k = 0; k < 9; ++k) {
                    const int dv = d[(s + k) & 15];
                    if (dv < mn) mn = dv;
                    if (-dv < mxn) mxn = -dv;
                }
                const int m = mn > mxn ? mn : mxn;
                if (m > best) best = m;
            }
            if (best > thr) score[(size_t)y * w + x] = best - 1;
        }
    int n = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int s = score[(size_t)y * w + x];
            if (s <= 0) continue;
            int is_max = 1;
            for (int j = -1; j <= 1 && is_max; ++j)
                for (int i = -1; i <= 1; ++i) {
                    if (!i && !j) continue;
                    const int yy = y + j, xx = x + i;
                    const int o = (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0 : score[(size_t)yy * w + xx];
                    if (!(s > o)) { is_max = 0; break; }
                }
            if (is_max && n < cap) { out_xy[2 * n] = (float)x; out_xy[2 * n + 1] = (float)y; ++n; }
        }
    free(score);
    return n;
}

/* ---------------------------------------------------------------- RANSAC + LM model fits */
typedef struct { float x, y; } cb_pt;

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9): w ascending, v columns */
static void jacobi_eig(int n, double* A, double* w, double* V) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0., diag = 0.;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (i == j) diag += A[i * n + j] * A[i * n + j];
                else off += A[i * n + j] * A[i * n + j];
            }
        if (off <= 1e-30 * diag || off < 1e-300) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2. * apq);
                const double t = (theta >= 0 ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                const double c = 1. / sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; ++i) {       /* ascending order */
        int m = i;
        for (int j = i + 1; j < n; ++j)
            if (w[j] < w[m]) m = j;
        if (m != i) {
            double t = w[i]; w[i] = w[m]; w[m] = t;
            for (int k = 0; k < n; ++k) { t = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = t; }
        }
    }
}

static int have_collinear(const cb_pt* p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = (double)p[j].x - p[i].x, dy1 = (double)p[j].y - p[i].y;
        for (int k = 0; k < j; ++k) {
            const double dx2 = (double)p[k].x - p[i].x, dy2 = (double)p[k].y - p[i].y;
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
        }
    }
    return 0;
}

static double det3(const cb_pt* p, int a, int b, int c) {
    return p[a].x * ((double)p[b].y - p[c].y) - p[a].y * ((double)p[b].x - p[c].x) +
           ((double)p[b].x * p[c].y - (double)p[b].y * p[c].x);
}

/* model: 0 = homography (4 points, 8 parameters), 1 = partial affine (2 points, 4 parameters) */
static int check_subset(int model, const cb_pt* a, const cb_pt* b, int count) {
    if (model == 1) return !have_collinear(a, count);
    if (have_collinear(a, count) || have_collinear(b, count)) return 0;
    if (count == 4) {
        static const int T[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int neg = 0;
        for (int t = 0; t < 4; ++t) neg += det3(a, T[t][0], T[t][1], T[t][2]) * det3(b, T[t][0], T[t][1], T[t][2]) < 0;
        if (neg != 0 && neg != 4) return 0;
    }
    return 1;
}

static int run_kernel(int model, const cb_pt* M, const cb_pt* m, int n, double* H) {
    if (model == 1) {
        const double x1 = M[0].x, y1 = M[0].y, x2 = M[1].x, y2 = M[1].y;
        const double X1 = m[0].x, Y1 = m[0].y, X2 = m[1].x, Y2 = m[1].y;
        const double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
        const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
        const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
        const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
        const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
        H[0] = S0; H[1] = -S1; H[2] = S2; H[3] = S1; H[4] = S0; H[5] = S3; H[6] = 0; H[7] = 0; H[8] = 1;
        return 1;
    }
    double cmx = 0, cmy = 0, cMx = 0, cMy = 0;
    for (int i = 0; i < n; ++i) { cmx += m[i].x; cmy += m[i].y; cMx += M[i].x; cMy += M[i].y; }
    cmx /= n; cmy /= n; cMx /= n; cMy /= n;
    double smx = 0, smy = 0, sMx = 0, sMy = 0;
    for (int i = 0; i < n; ++i) {
        smx += fabs(m[i].x - cmx); smy += fabs(m[i].y - cmy);
        sMx += fabs(M[i].x - cMx); sMy += fabs(M[i].y - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
    smx = n / smx; smy = n / smy; sMx = n / sMx; sMy = n / sMy;
    double LtL[81];
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < n; ++i) {
        const double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
        const double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; ++j)
            for (int k = 0; k < 9; ++k) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    double w[9], V[81];
    jacobi_eig(9, LtL, w, V);
    double h0[9];
    for (int k = 0; k < 9; ++k) h0[k] = V[k * 9 + 0];
    const double inv[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double nrm[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    double t[9], r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += inv[i * 3 + k] * h0[k * 3 + j];
            t[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += t[i * 3 + k] * nrm[k * 3 + j];
            r[i * 3 + j] = s;
        }
    if (fabs(r[8]) < DBL_MIN) return 0;
    for (int k = 0; k < 9; ++k) {
        H[k] = r[k] / r[8];
        if (!isfinite(H[k])) return 0;
    }
    return 1;
}

static void compute_error(int model, const cb_pt* M, const cb_pt* m, int n, const double* H, float* err) {
    float F[9];
    for (int k = 0; k < 9; ++k) F[k] = (float)H[k];
    for (int i = 0; i < n; ++i) {
        if (model == 0) {
            const float ww = 1.f / (F[6] * M[i].x + F[7] * M[i].y + 1.f);
            const float dx = (F[0] * M[i].x + F[1] * M[i].y + F[2]) * ww - m[i].x;
            const float dy = (F[3] * M[i].x + F[4] * M[i].y + F[5]) * ww - m[i].y;
            err[i] = dx * dx + dy * dy;
        } else {
            const float a = F[0] * M[i].x + F[1] * M[i].y + F[2] - m[i].x;
            const float b = F[3] * M[i].x + F[4] * M[i].y + F[5] - m[i].y;
            err[i] = a * a + b * b;
        }
    }
}

static int update_iters(double p, double ep, int mp, int max_iters) {
    p = p < 0 ? 0 : p > 1 ? 1 : p;
    ep = ep < 0 ? 0 : ep > 1 ? 1 : ep;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, mp);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* RANSACPointSetRegistrator::run; returns 1 and fills H[9], mask[n] on success */
int cb_ransac(int model, const float* p1, const float* p2, int count, double threshold, double confidence,
              int max_iters, double* H, uint8_t* mask_out) {
    const cb_pt* m1 = (const cb_pt*)p1;
    const cb_pt* m2 = (const cb_pt*)p2;
    const int mp = model == 0 ? 4 : 2;
    memset(mask_out, 0, count);
    if (count < mp) return 0;
    if (count == mp) {
        if (!run_kernel(model, m1, m2, count, H)) return 0;
        memset(mask_out, 1, count);
        return 1;
    }
    uint64_t state = 0xffffffffffffffffull;
    int niters = max_iters > 1 ? max_iters : 1, max_good = 0;
    const float thr2 = (float)(threshold * threshold);
    float* err = (float*)malloc(sizeof(float) * count);
    uint8_t* mask = (uint8_t*)malloc(count);
    for (int it = 0; it < niters; ++it) {
        int idx[4], found = 0;
        cb_pt s1[4], s2[4];
        for (int attempt = 0; attempt < 1000 && !found; ++attempt) {
            for (int i = 0; i < mp; ++i) {
                for (;;) {
                    state = (state & 0xffffffffull) * 4164903690ull + (state >> 32);
                    const int v = (int)((unsigned)state % (unsigned)count);
                    int dup = 0;
                    for (int j = 0; j < i; ++j) dup |= idx[j] == v;
                    if (!dup) { idx[i] = v; break; }
                }
                s1[i] = m1[idx[i]]; s2[i] = m2[idx[i]];
            }
            found = check_subset(model, s1, s2, mp);
        }
        if (!found) {
            if (it == 0) { free(err); free(mask); return 0; }
            break;
        }
        double M[9];
        if (!run_kernel(model, s1, s2, mp, M)) continue;
        compute_error(model, m1, m2, count, M, err);
        int good = 0;
        for (int i = 0; i < count; ++i) { mask[i] = err[i] <= thr2; good += mask[i]; }
        if (good > (max_good > mp - 1 ? max_good : mp - 1)) {
            memcpy(mask_out, mask, count);
            memcpy(H, M, sizeof(M));
            max_good = good;
            niters = update_iters(confidence, (double)(count - good) / count, mp, niters);
        }
    }
    free(err); free(mask);
    return max_good > 0;
}

/* LM refinement (LMSolverImpl::run, eps = FLT_EPSILON) of the 8 / 4 parameters over the inlier set */
static int n_params(int model) { return model == 0 ? 8 : 4; }

static void to_params(int model, const double* M, double* x) {
    if (model == 0) memcpy(x, M, sizeof(double) * 8);
    else { x[0] = M[0]; x[1] = M[3]; x[2] = M[2]; x[3] = M[5]; }
}

static void from_params(int model, const double* x, double* M) {
    if (model == 0) { memcpy(M, x, sizeof(double) * 8); M[8] = 1.; }
    else { M[0] = x[0]; M[1] = -x[1]; M[2] = x[2]; M[3] = x[1]; M[4] = x[0]; M[5] = x[3]; M[6] = 0; M[7] = 0; M[8] = 1; }
}

/* residuals r[2n] and (optionally) the normal equations A = J^T J, v = J^T r; returns |r|^2, *rinf = max |r| */
static double normal_eq(int model, const cb_pt* a, const cb_pt* b, int n, const double* h, double* A, double* v,
                        double* rinf) {
    const int lx = n_params(model);
    if (A) { memset(A, 0, sizeof(double) * lx * lx); memset(v, 0, sizeof(double) * lx); }
    double S = 0, mx = 0;
    for (int i = 0; i < n; ++i) {
        double r0, r1, j0[8], j1[8];
        const double X = a[i].x, Y = a[i].y;
        if (model == 0) {
            double ww = h[6] * X + h[7] * Y + 1.;
            ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0.;
            const double xi = (h[0] * X + h[1] * Y + h[2]) * ww, yi = (h[3] * X + h[4] * Y + h[5]) * ww;
            r0 = xi - b[i].x; r1 = yi - b[i].y;
            j0[0] = X * ww; j0[1] = Y * ww; j0[2] = ww; j0[3] = j0[4] = j0[5] = 0; j0[6] = -X * ww * xi; j0[7] = -Y * ww * xi;
            j1[0] = j1[1] = j1[2] = 0; j1[3] = X * ww; j1[4] = Y * ww; j1[5] = ww; j1[6] = -X * ww * yi; j1[7] = -Y * ww * yi;
        } else {
            r0 = h[0] * X - h[1] * Y + h[2] - b[i].x;
            r1 = h[1] * X + h[0] * Y + h[3] - b[i].y;
            j0[0] = X; j0[1] = -Y; j0[2] = 1; j0[3] = 0;
            j1[0] = Y; j1[1] = X; j1[2] = 0; j1[3] = 1;
        }
        S += r0 * r0 + r1 * r1;
        if (fabs(r0) > mx) mx = fabs(r0);
        if (fabs(r1) > mx) mx = fabs(r1);
        if (A)
            for (int p = 0; p < lx; ++p) {
                v[p] += j0[p] * r0 + j1[p] * r1;
                for (int q = 0; q < lx; ++q) A[p * lx + q] += j0[p] * j0[q] + j1[p] * j1[q];
            }
    }
    if (rinf) *rinf = mx;
    return S;
}

static void sym_solve(int n, const double* A, const double* b, double* x, double* inv_diag) {
    double M[64], w[8], V[64];
    memcpy(M, A, sizeof(double) * n * n);
    jacobi_eig(n, M, w, V);
    double wmax = 0;
    for (int i = 0; i < n; ++i) wmax = fabs(w[i]) > wmax ? fabs(w[i]) : wmax;
    const double thr = DBL_EPSILON * 2 * wmax * n;
    double inv[8];
    for (int i = 0; i < n; ++i) inv[i] = fabs(w[i]) > thr ? 1. / w[i] : 0.;
    for (int k = 0; k < n; ++k) {
        double s = 0;
        for (int i = 0; i < n; ++i) {
            double vb = 0;
            for (int j = 0; j < n; ++j) vb += V[j * n + i] * b[j];
            s += V[k * n + i] * inv[i] * vb;
        }
        x[k] = s;
        if (inv_diag) {
            double d = 0;
            for (int i = 0; i < n; ++i) d += V[k * n + i] * inv[i] * V[k * n + i];
            inv_diag[k] = d;
        }
    }
}

void cb_lm_refine(int model, const float* pa, const float* pb, int n, double* M, int max_iters) {
    const cb_pt *a = (const cb_pt*)pa, *b = (const cb_pt*)pb;
    const int lx = n_params(model);
    double x[8], xd[8], d[8], A[64], v[8], Ap[64], D[8], zero[8] = {0};
    to_params(model, M, x);
    double rinf = 0;
    double S = normal_eq(model, a, b, n, x, A, v, &rinf);
    for (int i = 0; i < lx; ++i) D[i] = A[i * lx + i];
    double lam = 1., lc = 0.75;
    const double eps = FLT_EPSILON;
    for (int it = 0;;) {
        memcpy(Ap, A, sizeof(double) * lx * lx);
        for (int i = 0; i < lx; ++i) Ap[i * lx + i] += lam * D[i];
        sym_solve(lx, Ap, v, d, NULL);
        for (int i = 0; i < lx; ++i) xd[i] = x[i] - d[i];
        const double Sd = normal_eq(model, a, b, n, xd, NULL, NULL, NULL);
        double dS = 0;
        for (int i = 0; i < lx; ++i) {
            double s = 0;
            for (int k = 0; k < lx; ++k) s += A[i * lx + k] * d[k];
            dS += d[i] * (2 * v[i] - s);
        }
        const double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > 0.75) {
            lam *= 0.5;
            if (lam < lc) lam = 0;
        } else if (R < 0.25) {
            double t = 0;
            for (int i = 0; i < lx; ++i) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = nu < 2. ? 2. : nu > 10. ? 10. : nu;
            if (lam == 0) {
                double inv_diag[8], dummy[8], mxv = 0;
                sym_solve(lx, A, zero, dummy, inv_diag);
                for (int i = 0; i < lx; ++i) mxv = fabs(inv_diag[i]) > mxv ? fabs(inv_diag[i]) : mxv;
                lam = lc = 1. / (mxv > DBL_EPSILON ? mxv : DBL_EPSILON);
                nu *= 0.5;
            }
            lam *= nu;
        }
        if (Sd < S) {
            memcpy(x, xd, sizeof(double) * lx);
            S = normal_eq(model, a, b, n, x, A, v, &rinf);
        }
        ++it;
        double dinf = 0;
        for (int i = 0; i < lx; ++i) dinf = fabs(d[i]) > dinf ? fabs(d[i]) : dinf;
        if (!(it < max_iters && dinf >= eps && rinf >= eps)) break;
    }
    from_params(model, x, M);
}

/* homography from all inliers (the re-fit findHomography does before LM) */
int cb_homography_fit(const float* p1, const float* p2, int n, double* H) {
    return run_kernel(0, (const cb_pt*)p1, (const cb_pt*)p2, n, H);
}

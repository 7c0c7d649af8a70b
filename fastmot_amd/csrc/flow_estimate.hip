// Host-side half of Flow.predict: robust camera-motion and per-track motion estimation.
//
// Replaces (fastmot/flow.py:215-263): cv2.findHomography(RANSAC, maxIters, confidence),
// cv2.estimateAffinePartial2D(RANSAC, maxIters, confidence) and the Numba helpers _get_good_match
// (:347-352), _fg_filter (:308-323), _estimate_bbox (:273-280), _get_inliers (:354-357).
//
// Why host C++ and not a kernel: each RANSAC loop is a short, strictly serial chain (OpenCV's RNG
// stream, data-dependent sample rejection, adaptive iteration count: typically 3-10 hypotheses over
// ~100 points, i.e. a few microseconds) and the per-track loop is ordered through the foreground
// mask; a device round trip (>= 20 us) per dependent step would dominate.  The pixel work that
// feeds it (pyramids, LK, corners) is on the GPU (flow.hip).  Times are reported, not rooflined
// ("latency-bound serial algorithms -- report us").
//
// Restated from OpenCV's published algorithms (calib3d/ptsetreg.cpp RANSACPointSetRegistrator,
// fundam.cpp HomographyEstimatorCallback / HomographyRefineCallback, ptsetreg.cpp
// AffinePartial2DEstimatorCallback / RefineCallback, levmarq.cpp LMSolverImpl, core rand.cpp RNG).
// OpenCV is not available: parity of this stage is UNPINNED; the numpy
// restatement in oracle/cv_oracle.py is the checker.
#include "common.h"
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>

namespace {

struct Pt { float x, y; };

struct CvRNG {   // cv::RNG (multiply-with-carry), seeded with (uint64)-1 like RANSACPointSetRegistrator
    uint64_t state = 0xffffffffffffffffULL;
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// symmetric eigen-decomposition (cyclic Jacobi), eigenvalues descending, rows of V = eigenvectors
void jacobi_eigen(int n, std::vector<double>& A, std::vector<double>& W, std::vector<double>& V) {
    V.assign((size_t)n * n, 0.);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0., diag = 0.;
        for (int i = 0; i < n; ++i) {
            diag += A[(size_t)i * n + i] * A[(size_t)i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        }
        if (off <= 1e-30 * diag || off < 1e-300) break;   // converged to round-off
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2. * apq);
                const double t = (theta >= 0 ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
                const double c = 1. / std::sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vpk = V[(size_t)p * n + k], vqk = V[(size_t)q * n + k];
                    V[(size_t)p * n + k] = c * vpk - s * vqk;
                    V[(size_t)q * n + k] = s * vpk + c * vqk;
                }
            }
    }
    W.resize(n);
    for (int i = 0; i < n; ++i) W[i] = A[(size_t)i * n + i];
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return W[a] > W[b]; });
    std::vector<double> W2(n), V2((size_t)n * n);
    for (int i = 0; i < n; ++i) {
        W2[i] = W[idx[i]];
        for (int k = 0; k < n; ++k) V2[(size_t)i * n + k] = V[(size_t)idx[i] * n + k];
    }
    W.swap(W2);
    V.swap(V2);
}

// x = pinv(A) b for symmetric A (cv::solve / cv::invert with DECOMP_EIGEN)
void sym_solve(int n, const std::vector<double>& A, const double* b, double* x, std::vector<double>* inv_diag) {
    std::vector<double> M(A), W, V;
    jacobi_eigen(n, M, W, V);
    double wmax = 0;
    for (double w : W) wmax = std::max(wmax, std::fabs(w));
    const double thr = DBL_EPSILON * 2 * wmax * n;
    std::vector<double> coef(n);
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += V[(size_t)i * n + k] * b[k];
        coef[i] = std::fabs(W[i]) > thr ? s / W[i] : 0.;
    }
    for (int k = 0; k < n; ++k) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += V[(size_t)i * n + k] * coef[i];
        x[k] = s;
    }
    if (inv_diag) {
        inv_diag->assign(n, 0.);
        for (int k = 0; k < n; ++k) {
            double s = 0;
            for (int i = 0; i < n; ++i)
                if (std::fabs(W[i]) > thr) s += V[(size_t)i * n + k] * V[(size_t)i * n + k] / W[i];
            (*inv_diag)[k] = s;
        }
    }
}

// ---------------------------------------------------------------- model callbacks
struct Model {
    virtual ~Model() {}
    virtual int model_points() const = 0;
    virtual int n_params() const = 0;
    virtual bool check_subset(const Pt* a, const Pt* b, int count) const = 0;
    virtual bool run_kernel(const Pt* a, const Pt* b, int count, double* M) const = 0;   // M: 9 doubles
    virtual void compute_error(const Pt* a, const Pt* b, int count, const double* M, float* err) const = 0;
    // LM refinement residuals / Jacobian on the parameter vector
    virtual void to_params(const double* M, double* h) const = 0;
    virtual void from_params(const double* h, double* M) const = 0;
    virtual void residuals(const Pt* a, const Pt* b, int count, const double* h, double* r, double* J) const = 0;
    // one pass over the points: A = J^T J (row-major lx*lx), v = J^T r, returns |r|^2 and max |r|
    // (exploits the sparsity of the two Jacobian rows of a point; J is never stored)
    virtual double normal_eq(const Pt* a, const Pt* b, int count, const double* h, double* A, double* v,
                             double* rinf) const = 0;
};

bool have_collinear(const Pt* p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = p[j].x - p[i].x, dy1 = p[j].y - p[i].y;
        for (int k = 0; k < j; ++k) {
            const double dx2 = p[k].x - p[i].x, dy2 = p[k].y - p[i].y;
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <=
                FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
                return true;
        }
    }
    return false;
}

double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

struct Homography : Model {
    double normal_eq(const Pt* M, const Pt* m, int count, const double* h, double* A, double* v,
                     double* rinf) const override {
        double G[6] = {0, 0, 0, 0, 0, 0};        // sum g g^T (upper: 00 01 02 11 12 22)
        double X[6] = {0, 0, 0, 0, 0, 0};        // sum -xi g * (g0, g1): [g0*g0, g0*g1, g1*g0.., ] stored as 3x2
        double Y[6] = {0, 0, 0, 0, 0, 0};
        double Q[3] = {0, 0, 0};                 // sum (xi^2 + yi^2) [g0 g0, g0 g1, g1 g1]
        double vv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double S = 0, ri = 0;
        for (int i = 0; i < count; ++i) {
            const double Mx = M[i].x, My = M[i].y;
            double ww = h[6] * Mx + h[7] * My + 1.;
            ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww, yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
            const double rx = xi - m[i].x, ry = yi - m[i].y;
            const double g0 = Mx * ww, g1 = My * ww, g2 = ww;
            G[0] += g0 * g0; G[1] += g0 * g1; G[2] += g0 * g2; G[3] += g1 * g1; G[4] += g1 * g2; G[5] += g2 * g2;
            X[0] -= xi * g0 * g0; X[1] -= xi * g0 * g1; X[2] -= xi * g1 * g0; X[3] -= xi * g1 * g1;
            X[4] -= xi * g2 * g0; X[5] -= xi * g2 * g1;
            Y[0] -= yi * g0 * g0; Y[1] -= yi * g0 * g1; Y[2] -= yi * g1 * g0; Y[3] -= yi * g1 * g1;
            Y[4] -= yi * g2 * g0; Y[5] -= yi * g2 * g1;
            const double q = xi * xi + yi * yi;
            Q[0] += q * g0 * g0; Q[1] += q * g0 * g1; Q[2] += q * g1 * g1;
            vv[0] += g0 * rx; vv[1] += g1 * rx; vv[2] += g2 * rx;
            vv[3] += g0 * ry; vv[4] += g1 * ry; vv[5] += g2 * ry;
            const double w = -(xi * rx + yi * ry);
            vv[6] += w * g0; vv[7] += w * g1;
            S += rx * rx + ry * ry;
            ri = std::max(ri, std::max(std::fabs(rx), std::fabs(ry)));
        }
        for (int i = 0; i < 64; ++i) A[i] = 0.;
        const double Gm[3][3] = {{G[0], G[1], G[2]}, {G[1], G[3], G[4]}, {G[2], G[4], G[5]}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                A[i * 8 + j] = Gm[i][j];
                A[(i + 3) * 8 + j + 3] = Gm[i][j];
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j) {
                A[i * 8 + 6 + j] = A[(6 + j) * 8 + i] = X[i * 2 + j];
                A[(i + 3) * 8 + 6 + j] = A[(6 + j) * 8 + i + 3] = Y[i * 2 + j];
            }
        A[6 * 8 + 6] = Q[0]; A[6 * 8 + 7] = A[7 * 8 + 6] = Q[1]; A[7 * 8 + 7] = Q[2];
        for (int i = 0; i < 8; ++i) v[i] = vv[i];
        *rinf = ri;
        return S;
    }
    int model_points() const override { return 4; }
    int n_params() const override { return 8; }
    bool check_subset(const Pt* a, const Pt* b, int count) const override {
        if (have_collinear(a, count) || have_collinear(b, count)) return false;
        if (count == 4) {   // orientation consistency of the 4 triangles
            static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
            int negative = 0;
            for (int i = 0; i < 4; ++i) {
                const int* t = tt[i];
                const double A[3][3] = {{a[t[0]].x, a[t[0]].y, 1.}, {a[t[1]].x, a[t[1]].y, 1.}, {a[t[2]].x, a[t[2]].y, 1.}};
                const double B[3][3] = {{b[t[0]].x, b[t[0]].y, 1.}, {b[t[1]].x, b[t[1]].y, 1.}, {b[t[2]].x, b[t[2]].y, 1.}};
                negative += det3(A) * det3(B) < 0;
            }
            if (negative != 0 && negative != 4) return false;
        }
        return true;
    }
    // normalised DLT, smallest eigenvector of L^T L (fundam.cpp HomographyEstimatorCallback::runKernel)
    bool run_kernel(const Pt* M, const Pt* m, int count, double* H) const override {
        double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
        for (int i = 0; i < count; ++i) { cmx += m[i].x; cmy += m[i].y; cMx += M[i].x; cMy += M[i].y; }
        cmx /= count; cmy /= count; cMx /= count; cMy /= count;
        double smx = 0, smy = 0, sMx = 0, sMy = 0;
        for (int i = 0; i < count; ++i) {
            smx += std::fabs(m[i].x - cmx); smy += std::fabs(m[i].y - cmy);
            sMx += std::fabs(M[i].x - cMx); sMy += std::fabs(M[i].y - cMy);
        }
        if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON ||
            std::fabs(sMy) < DBL_EPSILON)
            return false;
        smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
        std::vector<double> LtL(81, 0.);
        for (int i = 0; i < count; ++i) {
            const double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
            const double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
            const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
            const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
            for (int j = 0; j < 9; ++j)
                for (int k = j; k < 9; ++k) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
        }
        for (int j = 0; j < 9; ++j)
            for (int k = 0; k < j; ++k) LtL[j * 9 + k] = LtL[k * 9 + j];
        std::vector<double> W, V;
        jacobi_eigen(9, LtL, W, V);
        const double* h0 = &V[8 * 9];
        const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
        const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
        double T[9], R[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += invHnorm[i * 3 + k] * h0[k * 3 + j];
                T[i * 3 + j] = s;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += T[i * 3 + k] * Hnorm2[k * 3 + j];
                R[i * 3 + j] = s;
            }
        if (std::fabs(R[8]) < DBL_MIN) return false;
        for (int i = 0; i < 9; ++i) H[i] = R[i] / R[8];
        for (int i = 0; i < 9; ++i)
            if (!std::isfinite(H[i])) return false;
        return true;
    }
    void compute_error(const Pt* M, const Pt* m, int count, const double* H, float* err) const override {
        const float Hf[8] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
        for (int i = 0; i < count; ++i) {
            const float ww = 1.f / (Hf[6] * M[i].x + Hf[7] * M[i].y + 1.f);
            const float dx = (Hf[0] * M[i].x + Hf[1] * M[i].y + Hf[2]) * ww - m[i].x;
            const float dy = (Hf[3] * M[i].x + Hf[4] * M[i].y + Hf[5]) * ww - m[i].y;
            err[i] = dx * dx + dy * dy;
        }
    }
    void to_params(const double* M, double* h) const override { for (int i = 0; i < 8; ++i) h[i] = M[i]; }
    void from_params(const double* h, double* M) const override { for (int i = 0; i < 8; ++i) M[i] = h[i]; M[8] = 1.; }
    void residuals(const Pt* M, const Pt* m, int count, const double* h, double* r, double* J) const override {
        for (int i = 0; i < count; ++i) {
            const double Mx = M[i].x, My = M[i].y;
            double ww = h[6] * Mx + h[7] * My + 1.;
            ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww, yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
            r[2 * i] = xi - m[i].x;
            r[2 * i + 1] = yi - m[i].y;
            if (J) {
                double* j0 = J + (size_t)(2 * i) * 8;
                double* j1 = j0 + 8;
                j0[0] = Mx * ww; j0[1] = My * ww; j0[2] = ww; j0[3] = j0[4] = j0[5] = 0.;
                j0[6] = -Mx * ww * xi; j0[7] = -My * ww * xi;
                j1[0] = j1[1] = j1[2] = 0.; j1[3] = Mx * ww; j1[4] = My * ww; j1[5] = ww;
                j1[6] = -Mx * ww * yi; j1[7] = -My * ww * yi;
            }
        }
    }
};

struct AffinePartial : Model {
    double normal_eq(const Pt* f, const Pt* t, int count, const double* h, double* A, double* v,
                     double* rinf) const override {
        double m2 = 0, sx = 0, sy = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, S = 0, ri = 0;
        for (int i = 0; i < count; ++i) {
            const double Mx = f[i].x, My = f[i].y;
            const double rx = h[0] * Mx - h[1] * My + h[2] - t[i].x;
            const double ry = h[1] * Mx + h[0] * My + h[3] - t[i].y;
            m2 += Mx * Mx + My * My; sx += Mx; sy += My;
            v0 += Mx * rx + My * ry; v1 += -My * rx + Mx * ry; v2 += rx; v3 += ry;
            S += rx * rx + ry * ry;
            ri = std::max(ri, std::max(std::fabs(rx), std::fabs(ry)));
        }
        const double n = count;
        const double Am[16] = {m2, 0, sx, sy, 0, m2, -sy, sx, sx, -sy, n, 0, sy, sx, 0, n};
        for (int i = 0; i < 16; ++i) A[i] = Am[i];
        v[0] = v0; v[1] = v1; v[2] = v2; v[3] = v3;
        *rinf = ri;
        return S;
    }
    int model_points() const override { return 2; }
    int n_params() const override { return 4; }
    bool check_subset(const Pt* a, const Pt*, int count) const override { return !have_collinear(a, count); }
    bool run_kernel(const Pt* f, const Pt* t, int, double* M) const override {
        const double x1 = f[0].x, y1 = f[0].y, x2 = f[1].x, y2 = f[1].y;
        const double X1 = t[0].x, Y1 = t[0].y, X2 = t[1].x, Y2 = t[1].y;
        const double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
        const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
        const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
        const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
        const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
        M[0] = S0; M[1] = -S1; M[2] = S2; M[3] = S1; M[4] = S0; M[5] = S3; M[6] = 0; M[7] = 0; M[8] = 1;
        return true;
    }
    void compute_error(const Pt* f, const Pt* t, int count, const double* M, float* err) const override {
        const float F0 = (float)M[0], F1 = (float)M[1], F2 = (float)M[2], F3 = (float)M[3], F4 = (float)M[4], F5 = (float)M[5];
        for (int i = 0; i < count; ++i) {
            const float a = F0 * f[i].x + F1 * f[i].y + F2 - t[i].x;
            const float b = F3 * f[i].x + F4 * f[i].y + F5 - t[i].y;
            err[i] = a * a + b * b;
        }
    }
    void to_params(const double* M, double* h) const override { h[0] = M[0]; h[1] = M[3]; h[2] = M[2]; h[3] = M[5]; }
    void from_params(const double* h, double* M) const override {
        M[0] = h[0]; M[1] = -h[1]; M[2] = h[2]; M[3] = h[1]; M[4] = h[0]; M[5] = h[3]; M[6] = 0; M[7] = 0; M[8] = 1;
    }
    void residuals(const Pt* f, const Pt* t, int count, const double* h, double* r, double* J) const override {
        for (int i = 0; i < count; ++i) {
            const double Mx = f[i].x, My = f[i].y;
            r[2 * i] = h[0] * Mx - h[1] * My + h[2] - t[i].x;
            r[2 * i + 1] = h[1] * Mx + h[0] * My + h[3] - t[i].y;
            if (J) {
                double* j0 = J + (size_t)(2 * i) * 4;
                j0[0] = Mx; j0[1] = -My; j0[2] = 1.; j0[3] = 0.;
                j0[4] = My; j0[5] = Mx; j0[6] = 0.; j0[7] = 1.;
            }
        }
    }
};

int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    const double num0 = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    const double num = std::log(num0);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::nearbyint(num / denom);
}

// RANSACPointSetRegistrator::run.  Returns true and fills M (9) + mask when a model was found.
bool ransac_run(const Model& cb, const Pt* m1, const Pt* m2, int count, double threshold, double confidence,
                int max_iters, double* M, std::vector<uint8_t>& best_mask) {
    const int mp = cb.model_points();
    best_mask.assign(count, 0);
    if (count < mp) return false;
    if (count == mp) {
        if (!cb.run_kernel(m1, m2, count, M)) return false;
        std::fill(best_mask.begin(), best_mask.end(), 1);
        return true;
    }
    CvRNG rng;
    int niters = std::max(max_iters, 1);
    std::vector<float> err(count);
    std::vector<uint8_t> mask(count);
    std::vector<Pt> s1(mp), s2(mp);
    std::vector<int> idx(mp);
    int max_good = 0;
    const float thr2 = (float)(threshold * threshold);
    for (int iter = 0; iter < niters; ++iter) {
        // getSubset
        bool found = false;
        int attempts = 0;
        const int max_attempts = 1000;
        for (; attempts < max_attempts; ++attempts) {
            int i = 0;
            for (; i < mp; ++i) {
                int v;
                for (;;) {
                    v = rng.uniform(0, count);
                    bool dup = false;
                    for (int j = 0; j < i; ++j) dup |= idx[j] == v;
                    if (!dup) break;
                }
                idx[i] = v;
                s1[i] = m1[v];
                s2[i] = m2[v];
            }
            if (!cb.check_subset(s1.data(), s2.data(), mp)) continue;
            found = true;
            break;
        }
        if (!found) {
            if (iter == 0) return false;
            break;
        }
        double model[9];
        if (!cb.run_kernel(s1.data(), s2.data(), mp, model)) continue;
        cb.compute_error(m1, m2, count, model, err.data());
        int good = 0;
        for (int i = 0; i < count; ++i) {
            mask[i] = err[i] <= thr2 ? 1 : 0;
            good += mask[i];
        }
        if (good > std::max(max_good, mp - 1)) {
            best_mask = mask;
            memcpy(M, model, sizeof(double) * 9);
            max_good = good;
            niters = ransac_update_iters(confidence, (double)(count - good) / count, mp, niters);
        }
    }
    return max_good > 0;
}

// LMSolverImpl::run (levmarq.cpp), maxIters iterations, eps = FLT_EPSILON
void lm_refine(const Model& cb, const Pt* a, const Pt* b, int count, double* M, int max_iters) {
    const int lx = cb.n_params();
    double x[8], xd[8], d[8], temp_d[8], D[8], v[8], vn[8];
    std::vector<double> A((size_t)lx * lx), An((size_t)lx * lx), Ap, rd(2 * (size_t)count);
    cb.to_params(M, x);
    double rinf = 0;
    double S = cb.normal_eq(a, b, count, x, A.data(), v, &rinf);
    for (int i = 0; i < lx; ++i) D[i] = A[(size_t)i * lx + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    const double eps = FLT_EPSILON;
    for (int iter = 0;;) {
        Ap = A;
        for (int i = 0; i < lx; ++i) Ap[(size_t)i * lx + i] += lambda * D[i];
        sym_solve(lx, Ap, v, d, nullptr);
        for (int i = 0; i < lx; ++i) xd[i] = x[i] - d[i];
        // trial point: the normal equations at xd also give |r(xd)|^2; reused if the step is accepted
        double rinf_d = 0;
        const double Sd = cb.normal_eq(a, b, count, xd, An.data(), vn, &rinf_d);
        for (int i = 0; i < lx; ++i) {   // temp_d = 2 v - A d
            double s = 0;
            for (int k = 0; k < lx; ++k) s += A[(size_t)i * lx + k] * d[k];
            temp_d[i] = 2 * v[i] - s;
        }
        double dS = 0;
        for (int i = 0; i < lx; ++i) dS += d[i] * temp_d[i];
        const double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < lx; ++i) t += d[i] * v[i];
            double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                std::vector<double> inv_diag;
                double dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xx[8];
                sym_solve(lx, A, dummy, xx, &inv_diag);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; ++i) maxval = std::max(maxval, std::fabs(inv_diag[i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            rinf = rinf_d;
            for (int i = 0; i < lx; ++i) { x[i] = xd[i]; v[i] = vn[i]; }
            A.swap(An);
        }
        ++iter;
        double dinf = 0;
        for (int i = 0; i < lx; ++i) dinf = std::max(dinf, std::fabs(d[i]));
        if (!(iter < max_iters && dinf >= eps && rinf >= eps)) break;
    }
    cb.from_params(x, M);
}

double round_half_even(double v) { return std::nearbyint(v); }

}  // namespace


// ---- a small persistent worker pool for the host side of Flow.predict: the camera-motion RANSAC and the per-track
// RANSACs of one frame are independent jobs (OpenCV seeds a fresh RNG per findHomography / estimateAffinePartial2D
// call, so a track's estimate does not depend on which thread computes it or when).
//
// Wake-up latency matters at this scale (~0.25 ms of work per frame): fm_flow_predict calls prewake() BEFORE it
// waits for the GPU's LK kernel, the workers then spin (bounded) until the jobs arrive, and go back to sleep on the
// condition variable afterwards -- no permanently spinning cores.
namespace {
inline void cpu_relax() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __asm__ __volatile__("pause");
#endif
}

struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<void(int)> fn;
    std::atomic<int> next{0};
    std::atomic<uint64_t> gen{0};          // bumped when a job set is published
    std::atomic<uint64_t> wake{0};         // bumped by prewake()
    int n_jobs = 0;
    std::atomic<int> active{0};
    bool quit = false;

    explicit Pool(int workers) {
        for (int i = 0; i < workers; ++i) th.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        for (auto& t : th)
            if (t.joinable()) t.join();
    }
    void work() {
        for (;;) {
            const int j = next.fetch_add(1, std::memory_order_relaxed);
            if (j >= n_jobs) break;
            fn(j);
        }
    }
    void loop() {
        uint64_t seen_gen = 0, seen_wake = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || gen.load() != seen_gen || wake.load() != seen_wake; });
                if (quit) return;
            }
            seen_wake = wake.load();
            // spin (bounded) for the job set announced by prewake()
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(3);
            while (gen.load(std::memory_order_acquire) == seen_gen && std::chrono::steady_clock::now() < t_end)
                cpu_relax();
            if (gen.load(std::memory_order_acquire) == seen_gen) continue;       // nothing came: sleep again
            seen_gen = gen.load(std::memory_order_acquire);
            work();
            if (active.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(m);
                cv_done.notify_all();
            }
        }
    }
    void prewake() {
        {
            std::lock_guard<std::mutex> lk(m);
            wake.fetch_add(1);
        }
        cv.notify_all();
    }
    int workers() const { return (int)th.size(); }
    // hands fn(0..n-1) to the WORKERS (jobs are taken in index order) and returns at once; the caller does its own
    // part and then calls finish()
    void start(int n, std::function<void(int)> f) {
        {
            std::lock_guard<std::mutex> lk(m);
            fn = std::move(f);
            n_jobs = n;
            next.store(0);
            active.store((int)th.size());
            gen.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
    }
    void finish() {
        work();                                   // (normally nothing is left)
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return active.load() == 0; });
    }
};

Pool& pool() {
    static Pool p([] {
        if (const char* e = getenv("FASTMOT_FLOW_THREADS")) return std::max(0, atoi(e) - 1);
        const int hw = (int)std::thread::hardware_concurrency();
        return std::max(0, std::min(hw / 4, 6));          // + the calling thread
    }());
    return p;
}
}  // namespace

// accumulated wall time of the stages of fm_flow_predict (ms): begin, prepare, lk, estimate; [4] = calls;
// [5], [6]: camera-motion and per-track parts of fm_flow_estimate
static double g_flow_times[8] = {0, 0, 0, 0, 0, 0, 0, 0};

extern "C" int fm_flow_estimate(fm_ctx* ctx, int n_pts, const float* prev_pts, const float* cur_pts,
                                const uint8_t* status, int nT, const int32_t* begins, const int32_t* ends,
                                int bg_begin, int bg_end, const double* track_tlbr, int frame_w, int frame_h,
                                int ransac_max_iter, double ransac_conf, int inlier_thresh, double* H_out,
                                int* ok_out, int32_t* result_out, double* est_tlbr_out, int32_t* n_matched_out,
                                uint8_t* inlier_out) {
    FM_CHECK_ARG(ctx && n_pts >= 0 && nT >= 0 && H_out && ok_out && inlier_out);
    FM_CHECK_ARG(bg_begin >= 0 && bg_begin <= bg_end && bg_end <= n_pts);
    memset(inlier_out, 0, n_pts);
    *ok_out = 0;
    const auto te0 = std::chrono::steady_clock::now();
    for (int k = 0; k < nT; ++k) {
        result_out[k] = 0;
        n_matched_out[k] = 0;
    }
    const Pt* P = reinterpret_cast<const Pt*>(prev_pts);
    const Pt* C = reinterpret_cast<const Pt*>(cur_pts);
    // ---- camera motion: background matches [bg_begin, bg_end) with status (flow.py:216-232).  It only
    // decides whether the frame is usable and shares no data with the per-track estimates below: job 0 of the pool.
    double H[9];
    bool cam_ok = false;
    auto camera_motion = [&]() {
        std::vector<Pt> ba, bb;
        std::vector<int> bidx;
        std::vector<uint8_t> bmask;
        for (int i = bg_begin; i < bg_end; ++i)
            if (status[i]) { ba.push_back(P[i]); bb.push_back(C[i]); bidx.push_back(i); }
        if ((int)ba.size() < 4) return;
        Homography hcb;
        bool ok = ransac_run(hcb, ba.data(), bb.data(), (int)ba.size(), 3.0, ransac_conf, ransac_max_iter, H, bmask);
        int n_in = 0;
        if (ok) {
            std::vector<Pt> ia, ib;
            for (size_t i = 0; i < ba.size(); ++i)
                if (bmask[i]) { ia.push_back(ba[i]); ib.push_back(bb[i]); }
            n_in = (int)ia.size();
            if (ba.size() > 4 && n_in > 0) {
                if (hcb.run_kernel(ia.data(), ib.data(), n_in, H)) lm_refine(hcb, ia.data(), ib.data(), n_in, H, 10);
            }
        }
        if (!ok || n_in < inlier_thresh) return;
        for (size_t i = 0; i < ba.size(); ++i)
            if (bmask[i]) inlier_out[bidx[i]] = 1;          // background indices only: disjoint from the tracks'
        cam_ok = true;
    };

    // ---- per-track motion (flow.py:235-263), closest-first order; the foreground mask is the set of predicted
    // boxes of the tracks accepted so far.  That mask is the ONLY coupling between tracks (a keypoint that has
    // moved under an already predicted box is dropped before the fit), and it rarely bites: every track is first
    // estimated speculatively with no mask, in parallel; the sequential pass below then accepts a speculative
    // result iff none of the track's candidate points is covered by the boxes accepted before it -- the point set,
    // hence the RANSAC draw sequence and the result, are then exactly those of the sequential algorithm --
    // and recomputes the track in place otherwise.
    struct TrackFit {
        std::vector<int> cand;        // indices of the candidate points the fit saw
        std::vector<int> inl;         // indices of the inliers
        double est[4] = {0, 0, 0, 0};
        int n = 0, result = 0;
        bool fitted = false;          // est / inl valid
    };
    auto fit_track = [&](int k, const std::vector<double>* boxes, TrackFit& r) {
        static thread_local std::vector<Pt> a, b;
        static thread_local std::vector<uint8_t> mask;
        a.clear(); b.clear();
        r.cand.clear(); r.inl.clear();
        r.result = 0; r.fitted = false;
        for (int i = begins[k]; i < ends[k]; ++i) {
            if (!status[i]) continue;
            // _fg_filter: inside the frame and not under an already predicted box
            const int x = (int)std::nearbyint(C[i].x), y = (int)std::nearbyint(C[i].y);
            if (x < 0 || y < 0 || x >= frame_w || y >= frame_h) continue;
            if (boxes) {
                bool covered = false;
                for (size_t q = 0; q < boxes->size(); q += 4)
                    if (x >= (*boxes)[q] && x <= (*boxes)[q + 2] && y >= (*boxes)[q + 1] && y <= (*boxes)[q + 3]) { covered = true; break; }
                if (covered) continue;
            }
            a.push_back(P[i]); b.push_back(C[i]); r.cand.push_back(i);
        }
        const int n = (int)a.size();
        r.n = n;
        if (n < 3) return;
        double M[9];
        AffinePartial acb;
        if (!ransac_run(acb, a.data(), b.data(), n, 3.0, ransac_conf, ransac_max_iter, M, mask)) return;
        std::vector<Pt> ia, ib;
        for (int i = 0; i < n; ++i)
            if (mask[i]) { ia.push_back(a[i]); ib.push_back(b[i]); r.inl.push_back(r.cand[i]); }
        if (n > 2 && !ia.empty()) lm_refine(acb, ia.data(), ib.data(), (int)ia.size(), M, 10);
        // _estimate_bbox (flow.py:273-280)
        const double* tb = track_tlbr + 4 * k;
        const double tlx = tb[0] * M[0] + tb[1] * M[1] + M[2], tly = tb[0] * M[3] + tb[1] * M[4] + M[5];
        double scale = std::sqrt(M[0] * M[0] + M[3] * M[3]);
        if (scale < 0.9 || scale > 1.1) scale = 1.;
        const double w = tb[2] - tb[0] + 1, h = tb[3] - tb[1] + 1;
        r.est[0] = round_half_even(tlx); r.est[1] = round_half_even(tly);
        r.est[2] = round_half_even(tlx + w * scale - 1.); r.est[3] = round_half_even(tly + h * scale - 1.);
        r.fitted = true;
        const double ix1 = std::max(r.est[0], 0.), iy1 = std::max(r.est[1], 0.);
        const double ix2 = std::min(r.est[2], (double)frame_w - 1), iy2 = std::min(r.est[3], (double)frame_h - 1);
        const bool outside = ix2 < ix1 || iy2 < iy1;
        // 2: estimated but rejected (prev_keypoints updated, keypoints cleared); 1: accepted
        r.result = (outside || (int)r.inl.size() < inlier_thresh) ? 2 : 1;
    };
    static thread_local std::vector<TrackFit> fits;
    static thread_local std::vector<std::atomic<int>> ready;
    if ((int)fits.size() < nT) {
        fits.resize(nT);
        ready = std::vector<std::atomic<int>>(nT);
    }
    TrackFit* fp = fits.data();        // (a thread_local named inside a worker's lambda would be the WORKER's instance)
    std::atomic<int>* rdy = ready.data();
    Pool& pl = pool();
    const bool parallel = pl.workers() > 0 && nT > 0;
    if (parallel) {
        for (int k = 0; k < nT; ++k) rdy[k].store(0, std::memory_order_relaxed);
        // job 0 = camera motion (the longest job, started first), jobs 1..nT = speculative per-track fits in
        // closest-first order; this thread validates / commits them in the same order as they become ready
        pl.start(nT + 1, [&, fp, rdy](int job) {
            if (job == 0) { camera_motion(); return; }
            fit_track(job - 1, nullptr, fp[job - 1]);
            rdy[job - 1].store(1, std::memory_order_release);
        });
    } else {
        camera_motion();
    }
    {
    std::vector<double> boxes;   // accepted est_tlbr, crop() semantics
    int n_redone = 0;
    for (int k = 0; k < nT; ++k) {
        TrackFit& r = fp[k];
        if (!parallel) {
            fit_track(k, &boxes, r);               // the plain sequential algorithm
        } else {
            while (rdy[k].load(std::memory_order_acquire) == 0) cpu_relax();
            bool valid = true;
            if (!boxes.empty() && !r.cand.empty()) {
                // bounding box of the candidate points first: most accepted boxes are nowhere near this track
                int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN;
                for (int i : r.cand) {
                    const int x = (int)std::nearbyint(C[i].x), y = (int)std::nearbyint(C[i].y);
                    bx0 = std::min(bx0, x); bx1 = std::max(bx1, x); by0 = std::min(by0, y); by1 = std::max(by1, y);
                }
                for (size_t q = 0; q < boxes.size() && valid; q += 4) {
                    if (boxes[q] > bx1 || boxes[q + 2] < bx0 || boxes[q + 1] > by1 || boxes[q + 3] < by0) continue;
                    for (int i : r.cand) {
                        const int x = (int)std::nearbyint(C[i].x), y = (int)std::nearbyint(C[i].y);
                        if (x >= boxes[q] && x <= boxes[q + 2] && y >= boxes[q + 1] && y <= boxes[q + 3]) { valid = false; break; }
                    }
                }
            }
            if (!valid) {
                fit_track(k, &boxes, r);
                ++n_redone;
            }
        }
        n_matched_out[k] = r.n;
        if (!r.fitted) continue;
        for (int i : r.inl) inlier_out[i] = 1;
        memcpy(est_tlbr_out + 4 * k, r.est, sizeof(r.est));
        result_out[k] = r.result;
        if (r.result != 1) continue;
        // crop(fg_mask, est_tlbr)[:] = 0 : int truncation, clamp at 0 (utils/rect.py:83-89)
        boxes.push_back(std::max((double)(int)r.est[0], 0.));
        boxes.push_back(std::max((double)(int)r.est[1], 0.));
        boxes.push_back(std::max((double)(int)r.est[2], 0.));
        boxes.push_back(std::max((double)(int)r.est[3], 0.));
    }
    g_flow_times[5] += n_redone;
    }
    if (parallel) pl.finish();         // the camera-motion job (and nothing else) may still be running
    g_flow_times[6] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te0).count();
    if (!cam_ok) {                            // flow.py:227-231: nothing of this frame is used
        memset(inlier_out, 0, n_pts);
        for (int k = 0; k < nT; ++k) {
            result_out[k] = 0;
            n_matched_out[k] = 0;
        }
        return 0;
    }
    memcpy(H_out, H, sizeof(double) * 9);
    *ok_out = 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Flow.predict (flow.py:135-264) in ONE call: pyramid of the new frame, keypoint bookkeeping /
// detection / background keypoints, pyramidal LK, camera motion + per-track boxes.  The Python side
// only orders the tracks and scatters the results; the glue arithmetic below reproduces the NumPy
// expressions of the reference bit for bit (float32 products, see the comments).
// ---------------------------------------------------------------------------------------------------
extern double g_flow_sub[8];

// ---- The KLT + Kalman chain of a step on a native worker thread (fm_track_predict_async / _wait).
// fastmot_amd/mot.py used to hand Flow.predict + apply_kalman to a second PYTHON thread: its marshalling and result
// scattering then shared the interpreter lock with the main thread's own Python (every return from a C call had to win
// the lock back), which stretched both threads by 0.1-0.2 ms per step.  The worker below runs fm_flow_predict and the
// Kalman launch (fm_trk_step, with the KLT boxes and multipliers MultiTracker.apply_kalman derives from the
// prediction: tracker.py:164-183) without touching Python; the caller marshals before and scatters after, on its own
// thread, while the GPU is busy anyway.  The worker spins (bounded, 2 ms) for the next job after finishing one and
// sleeps on a condition variable otherwise.
namespace {
struct TrackPredictJob {
    fm_ctx* ctx = nullptr;
    int nT = 0, pts_cap = 0, nK = 0;
    const double *inside = nullptr, *full = nullptr;
    const float* kps = nullptr;
    const int32_t* kp_off = nullptr;
    fm_flow_predict_params prm{};
    float *prev_out = nullptr, *cur_out = nullptr;
    int32_t *trk_off_out = nullptr, *bg_range_out = nullptr, *result_out = nullptr, *n_matched_out = nullptr;
    double *H_out = nullptr, *est_out = nullptr;
    const int32_t *slots = nullptr, *ages = nullptr, *sorted_idx = nullptr;
    double age_penalty = 1.;
    double* tlbr_out = nullptr;
    uint8_t* lost_out = nullptr;
    int status = FM_FLOW_NO_BACKGROUND, kalman_done = 0, rc = 0;
    char err[512] = {0};
};

struct PredictWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::atomic<uint64_t> submitted{0}, completed{0};
    TrackPredictJob job;
    bool quit = false;

    PredictWorker() { th = std::thread([this] { loop(); }); }
    ~PredictWorker() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    static void run(TrackPredictJob& j) {
        fm_ctx* ctx = j.ctx;
        j.rc = 0; j.kalman_done = 0; j.err[0] = 0;
        j.status = FM_FLOW_NO_BACKGROUND;
        if (hipSetDevice(ctx->device) != hipSuccess) { j.rc = FM_ERR_HIP; snprintf(j.err, sizeof(j.err), "hipSetDevice failed"); }
        if (!j.rc)
            j.rc = fm_flow_predict(ctx, j.nT, j.inside, j.full, j.kps, j.kp_off, &j.prm, j.pts_cap, j.prev_out, j.cur_out,
                                   j.trk_off_out, j.bg_range_out, j.H_out, &j.status, j.result_out, j.est_out, j.n_matched_out);
        if (!j.rc && j.status == FM_FLOW_OK && j.nK > 0) {
            // MultiTracker.apply_kalman: a track whose KLT box was estimated gets it as a measurement, with a large
            // uncertainty for occluded tracks (large age / low inlier ratio)
            std::vector<double> klt(4 * (size_t)j.nK, 0.), mult(j.nK, 1.);
            std::vector<uint8_t> has(j.nK, 0);
            for (int i = 0; i < j.nK; ++i) {
                const int k = j.sorted_idx[i];
                if (k < 0 || j.result_out[k] == 0 || j.result_out[k] == 2) continue;
                for (int e = 0; e < 4; ++e) klt[4 * i + e] = j.est_out[4 * k + e];
                has[i] = 1;
                const double inlier_ratio = (double)(j.trk_off_out[k + 1] - j.trk_off_out[k]) / (double)j.n_matched_out[k];
                const double a = j.age_penalty * (double)j.ages[i];
                mult[i] = (a > 1. ? a : 1.) / inlier_ratio;
            }
            j.rc = fm_trk_step(ctx, j.nK, j.slots, j.H_out, klt.data(), has.data(), mult.data(), j.tlbr_out, j.lost_out);
            j.kalman_done = j.rc == 0;
        }
        if (j.rc) snprintf(j.err, sizeof(j.err), "%s", fm_last_error());
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            // bounded spin for the next job (a pipeline submits one per frame), then sleep
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
            while (submitted.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < t_end) cpu_relax();
            if (submitted.load(std::memory_order_acquire) == seen) {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || submitted.load() != seen; });
                if (quit) return;
            }
            seen = submitted.load(std::memory_order_acquire);
            run(job);
            {
                std::lock_guard<std::mutex> lk(m);
                completed.store(seen, std::memory_order_release);
            }
            cv_done.notify_all();
        }
    }
};

// one worker per context (a second fm_ctx, or a second pipeline thread on another context, has its own job slot)
PredictWorker& predict_worker(fm_ctx* ctx) {
    if (!ctx->predict_worker) ctx->predict_worker = new PredictWorker();
    return *static_cast<PredictWorker*>(ctx->predict_worker);
}
}  // namespace

void fm_predict_worker_free(fm_ctx* ctx) {
    delete static_cast<PredictWorker*>(ctx->predict_worker);
    ctx->predict_worker = nullptr;
}

extern "C" int fm_track_predict_async(fm_ctx* ctx, int nT, const double* inside_tlbr, const double* full_tlbr,
                                      const float* kps, const int32_t* kp_off, const fm_flow_predict_params* prm,
                                      int pts_cap, float* prev_out, float* cur_out, int32_t* trk_off_out,
                                      int32_t* bg_range_out, double* H_out, int32_t* result_out, double* est_tlbr_out,
                                      int32_t* n_matched_out, int nK, const int32_t* slots, const int32_t* ages,
                                      const int32_t* sorted_idx, double age_penalty, double* tlbr_out,
                                      uint8_t* lost_out) {
    FM_CHECK_ARG(ctx && ctx->flow && nT >= 0 && nK >= 0 && prm && pts_cap > 0 && prev_out && cur_out && trk_off_out &&
                 bg_range_out && H_out);
    FM_CHECK_ARG(nT == 0 || (inside_tlbr && full_tlbr && kp_off && result_out && est_tlbr_out && n_matched_out));
    FM_CHECK_ARG(nK == 0 || (slots && ages && sorted_idx && tlbr_out && lost_out));
    PredictWorker& w = predict_worker(ctx);
    if (w.completed.load(std::memory_order_acquire) != w.submitted.load(std::memory_order_acquire)) {
        fm_set_error("a track prediction is already in flight (fm_track_predict_wait first)");
        return FM_ERR_STATE;
    }
    TrackPredictJob& j = w.job;
    j.ctx = ctx; j.nT = nT; j.inside = inside_tlbr; j.full = full_tlbr; j.kps = kps; j.kp_off = kp_off; j.prm = *prm;
    j.pts_cap = pts_cap; j.prev_out = prev_out; j.cur_out = cur_out; j.trk_off_out = trk_off_out;
    j.bg_range_out = bg_range_out; j.H_out = H_out; j.result_out = result_out; j.est_out = est_tlbr_out;
    j.n_matched_out = n_matched_out; j.nK = nK; j.slots = slots; j.ages = ages; j.sorted_idx = sorted_idx;
    j.age_penalty = age_penalty; j.tlbr_out = tlbr_out; j.lost_out = lost_out;
    {
        std::lock_guard<std::mutex> lk(w.m);
        w.submitted.fetch_add(1, std::memory_order_release);
    }
    w.cv.notify_all();
    return 0;
}

// blocks until the job of fm_track_predict_async has finished; status_out: FM_FLOW_*; kalman_done_out: 1 when the
// Kalman step ran (status OK and nK > 0) and tlbr_out / lost_out are valid
extern "C" int fm_track_predict_wait(fm_ctx* ctx, int* status_out, int* kalman_done_out) {
    FM_CHECK_ARG(ctx && status_out && kalman_done_out);
    PredictWorker& w = predict_worker(ctx);
    const uint64_t want = w.submitted.load(std::memory_order_acquire);
    for (int spins = 0; spins < 20000 && w.completed.load(std::memory_order_acquire) != want; ++spins) cpu_relax();
    if (w.completed.load(std::memory_order_acquire) != want) {
        std::unique_lock<std::mutex> lk(w.m);
        w.cv_done.wait(lk, [&] { return w.completed.load() == want; });
    }
    const TrackPredictJob& j = w.job;
    *status_out = j.status;
    *kalman_done_out = j.kalman_done;
    if (j.rc) {
        fm_set_error("%s", j.err);
        return j.rc;
    }
    return 0;
}

extern "C" int fm_flow_timing(double* out5, int reset) {
    for (int i = 0; i < 5; ++i) out5[i] = g_flow_times[i];
    if (getenv("FASTMOT_FLOW_TIMING_VERBOSE")) {
        const double nc = g_flow_times[4] > 0 ? g_flow_times[4] : 1;
        fprintf(stderr, "flow sub-stages (ms/call): prepare host %.3f sync0 %.3f enqueue %.3f sync1 %.3f | lk sync0 %.3f enqueue %.3f sync1 %.3f\n",
                g_flow_sub[0] / nc, g_flow_sub[1] / nc, g_flow_sub[2] / nc, g_flow_sub[3] / nc, g_flow_sub[4] / nc,
                g_flow_sub[5] / nc, g_flow_sub[6] / nc);
    }
    if (getenv("FASTMOT_FLOW_TIMING_VERBOSE"))
        fprintf(stderr, "flow_estimate: %.3f tracks re-fitted under the mask, %.3f ms per call\n",
                g_flow_times[5] / (g_flow_times[4] > 0 ? g_flow_times[4] : 1), g_flow_times[6] / (g_flow_times[4] > 0 ? g_flow_times[4] : 1));
    if (reset) {
        for (double& v : g_flow_times) v = 0;
        for (double& v : g_flow_sub) v = 0;
    }
    return 0;
}

extern "C" int fm_flow_predict(fm_ctx* ctx, int nT, const double* inside_tlbr, const double* full_tlbr,
                               const float* kps, const int32_t* kp_off, const fm_flow_predict_params* prm,
                               int pts_cap, float* prev_out, float* cur_out, int32_t* trk_off_out,
                               int32_t* bg_range_out, double* H_out, int* status_out, int32_t* result_out,
                               double* est_tlbr_out, int32_t* n_matched_out) {
    FM_CHECK_ARG(ctx && ctx->flow && nT >= 0 && prm && pts_cap > 0 && prev_out && cur_out && trk_off_out &&
                 bg_range_out && H_out && status_out);
    FM_CHECK_ARG(nT == 0 || (inside_tlbr && full_tlbr && kp_off && result_out && est_tlbr_out && n_matched_out));
    *status_out = FM_FLOW_NO_BACKGROUND;
    for (int k = 0; k <= nT; ++k) trk_off_out[k] = 0;
    bg_range_out[0] = bg_range_out[1] = 0;
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto lap = [&](int i) {
        const auto t1 = clk::now();
        g_flow_times[i] += std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
    };
    g_flow_times[4] += 1.0;
    int rc = fm_flow_begin(ctx);
    if (rc) return rc;
    lap(0);

    // ---- keypoint bookkeeping + detection (flow.py:156-200)
    const int n_kps = nT ? kp_off[nT] : 0;
    static thread_local std::vector<int32_t> area, new_off, new_cnt, begins, ends;
    static thread_local std::vector<uint8_t> keep, needy, status, inl;
    static thread_local std::vector<float> new_pts, bg_pts, prev, scaled, cur, err;
    const int bg_cap = 8192;
    area.assign(nT, 0); new_off.assign(nT, 0); new_cnt.assign(nT, 0); needy.assign(nT, 0);
    keep.assign(n_kps > 0 ? n_kps : 1, 0);
    new_pts.resize((size_t)pts_cap * 2);
    bg_pts.resize((size_t)bg_cap * 2);
    int n_new = 0, n_bg = 0;
    const int32_t zero_off = 0;
    rc = fm_flow_prepare(ctx, nT, inside_tlbr, full_tlbr, kps, nT ? kp_off : &zero_off, prm->feat_density,
                         prm->feat_dist_factor, area.data(), keep.data(), needy.data(), pts_cap, new_pts.data(),
                         new_off.data(), new_cnt.data(), &n_new, bg_cap, bg_pts.data(), &n_bg);
    if (rc) return rc;
    lap(1);
    prev.clear();
    begins.assign(nT, 0); ends.assign(nT, 0);
    for (int k = 0; k < nT; ++k) {
        begins[k] = (int32_t)(prev.size() / 2);
        if (needy[k]) {   // only detect new keypoints when too few are propagated
            const float* p = new_pts.data() + 2 * (size_t)new_off[k];
            prev.insert(prev.end(), p, p + 2 * (size_t)new_cnt[k]);
        } else {
            for (int i = kp_off[k]; i < kp_off[k + 1]; ++i)
                if (keep[i]) { prev.push_back(kps[2 * i]); prev.push_back(kps[2 * i + 1]); }
        }
        ends[k] = (int32_t)(prev.size() / 2);
    }
    if (n_bg == 0) {   // flow.py:191-196
        return fm_flow_swap(ctx);
    }
    const int bg_begin = (int)(prev.size() / 2);
    // keypoints = keypoints * (1 / bg_scale): float32 reciprocal, float32 product
    const float ibx = 1.0f / prm->bg_scale[0], iby = 1.0f / prm->bg_scale[1];
    for (int i = 0; i < n_bg; ++i) {
        prev.push_back(bg_pts[2 * i] * ibx);
        prev.push_back(bg_pts[2 * i + 1] * iby);
    }
    const int n_pts = (int)(prev.size() / 2);
    FM_CHECK_ARG(n_pts <= pts_cap);

    // ---- optical flow on the scaled images (flow.py:202-213)
    scaled.resize(prev.size()); cur.resize(prev.size()); err.resize(n_pts); status.resize(n_pts); inl.resize(n_pts);
    for (int i = 0; i < n_pts; ++i) {
        scaled[2 * i] = prev[2 * i] * prm->opt_scale[0];
        scaled[2 * i + 1] = prev[2 * i + 1] * prm->opt_scale[1];
    }
    pool().prewake();          // the RANSAC workers wake up while this thread waits for the LK kernel
    rc = fm_flow_lk(ctx, n_pts, scaled.data(), cur.data(), status.data(), err.data());
    if (rc) return rc;
    lap(2);
    const float iox = 1.0f / prm->opt_scale[0], ioy = 1.0f / prm->opt_scale[1];
    const float max_err = (float)prm->max_error;
    for (int i = 0; i < n_pts; ++i) {
        status[i] = (status[i] && err[i] < max_err) ? 1 : 0;
        if (status[i]) { cur[2 * i] *= iox; cur[2 * i + 1] *= ioy; }
    }

    // ---- camera motion + per-track boxes (flow.py:215-263)
    int ok = 0;
    const int bg_end = n_pts - 1 > bg_begin ? n_pts - 1 : bg_begin;
    rc = fm_flow_estimate(ctx, n_pts, prev.data(), cur.data(), status.data(), nT, begins.data(), ends.data(),
                          bg_begin, bg_end, full_tlbr, prm->frame_w, prm->frame_h, prm->ransac_max_iter,
                          prm->ransac_conf, prm->inlier_thresh, H_out, &ok, result_out, est_tlbr_out, n_matched_out,
                          inl.data());
    if (rc) return rc;
    lap(3);
    if (!ok) {
        *status_out = FM_FLOW_NO_HOMOGRAPHY;
        return 0;
    }
    // ---- compact the inlier keypoints: per track (result != 0), then the background
    int w = 0;
    for (int k = 0; k < nT; ++k) {
        trk_off_out[k] = w;
        if (result_out[k] == 0) continue;
        for (int i = begins[k]; i < ends[k]; ++i)
            if (inl[i]) {
                prev_out[2 * w] = prev[2 * i]; prev_out[2 * w + 1] = prev[2 * i + 1];
                cur_out[2 * w] = cur[2 * i]; cur_out[2 * w + 1] = cur[2 * i + 1];
                ++w;
            }
    }
    trk_off_out[nT] = w;
    bg_range_out[0] = w;
    for (int i = bg_begin; i < n_pts; ++i)
        if (inl[i]) {
            prev_out[2 * w] = prev[2 * i]; prev_out[2 * w + 1] = prev[2 * i + 1];
            cur_out[2 * w] = cur[2 * i]; cur_out[2 * w + 1] = cur[2 * i + 1];
            ++w;
        }
    bg_range_out[1] = w;
    *status_out = FM_FLOW_OK;
    return 0;
}

// Batched Kalman filter on the device-resident track table (fp64).
//
// Replaces the per-track Python loop of MultiTracker.apply_kalman (fastmot/tracker.py:164-183)
// and its Numba bodies KalmanFilter.warp/_predict/_project/_update
// (fastmot/kalman_filter.py:227-292, 308-345) with ONE launch over all tracks:
// one 64-lane wavefront per track, lane (r,c) owns covariance element P[r][c]; the 8x8
// products run through LDS (3 x 512 B per wave).  The 4x4 innovation solve uses partial
// pivoting like LAPACK gesv behind np.linalg.solve (kalman_filter.py:341).
//
// Roofline: 576 B of state read+written and ~5 kFLOP fp64 per track => launch-latency bound
// (T=50: 58 KB); the design goal is a single ~5 us launch instead of 150 Numba calls.
#include "common.h"

namespace {

struct Hmat { double h[9]; };
struct Rect { double r[4]; };

constexpr int WAVES = 4;   // tracks per 256-thread block

__device__ inline double ios_frame(const double* b, const double* f) {
    // utils/rect.py:101-109
    const double iw = fmin(b[2], f[2]) - fmax(b[0], f[0]) + 1;
    const double ih = fmin(b[3], f[3]) - fmax(b[1], f[1]) + 1;
    if (iw <= 0 || ih <= 0) return 0.;
    const double w = b[2] - b[0] + 1, h = b[3] - b[1] + 1;
    const double area = (w <= 0 || h <= 0) ? 0. : w * h;
    return iw * ih / area;
}

// Measurement update shared by the KLT (FLOW) and detector paths.
// sm: mean[8] in LDS, sP: cov[64] in LDS, sK: scratch[64] (K in [r*4+j], r<8), sS: scratch.
__device__ inline void kf_update_wave(int lane, double* sm, double* sP, double* sK, double* sS,
                                      const double* z, double fac_w, double fac_h, double min_w,
                                      double min_h, double mult, bool apply) {
    const int r = lane >> 3, c = lane & 7;
    // _project (kalman_filter.py:321-336): R from the CURRENT mean's box size
    const double w = sm[2] - sm[0] + 1, h = sm[3] - sm[1] + 1;
    const double sw = fmax(fac_w * w, min_w) * mult, sh = fmax(fac_h * h, min_h) * mult;
    if (lane < 16) {
        const int i = lane >> 2, j = lane & 3;
        double s = sP[i * 8 + j];
        if (i == j) s += (i & 1) ? sh * sh : sw * sw;
        sS[lane] = s;
    }
    __syncthreads();
    // K = solve(S, (P H^T)^T)^T : lane r<8 solves S x = P[r, :4]^T  (kalman_filter.py:341)
    if (lane < 8) {
        double A[4][4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) A[i][j] = sS[i * 4 + j];
            b[i] = sP[lane * 8 + i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int piv = k;
            double best = fabs(A[k][k]);
#pragma unroll
            for (int i = k + 1; i < 4; ++i)
                if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); piv = i; }
#pragma unroll
            for (int i = k + 1; i < 4; ++i)
                if (i == piv) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { double t = A[k][j]; A[k][j] = A[i][j]; A[i][j] = t; }
                    double t = b[k]; b[k] = b[i]; b[i] = t;
                }
#pragma unroll
            for (int i = k + 1; i < 4; ++i) {
                const double f = A[i][k] / A[k][k];
#pragma unroll
                for (int j = k; j < 4; ++j) A[i][j] -= f * A[k][j];
                b[i] -= f * b[k];
            }
        }
        double x[4];
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            double s = b[i];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) s -= A[i][j] * x[j];
            x[i] = s / A[i][i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sK[lane * 4 + j] = x[j];
    }
    __syncthreads();
    // KS[r][j] = sum_i K[r][i] S[i][j]  -> store in sK[32 + r*4 + j]
    if (lane < 32) {
        const int rr = lane >> 2, j = lane & 3;
        double s = 0.;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += sK[rr * 4 + i] * sS[i * 4 + j];
        sK[32 + lane] = s;
    }
    __syncthreads();
    // P -= (K S) K^T ; mean += innovation @ K^T   (kalman_filter.py:342-345)
    double acc = 0.;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += sK[32 + r * 4 + j] * sK[c * 4 + j];
    const double pnew = sP[lane] - acc;
    double mnew = 0.;
    if (lane < 8) {
        double s = 0.;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (z[j] - sm[j]) * sK[lane * 4 + j];
        mnew = sm[lane] + s;
    }
    __syncthreads();
    if (apply) {   // wave-uniform; every wave of the block runs the barriers above
        sP[lane] = pnew;
        if (lane < 8) sm[lane] = mnew;
    }
    __syncthreads();
}

__device__ inline void write_box(int lane, const double* sm, const Rect& fr, double* tlbr_out,
                                 uint8_t* lost_out, int w) {
    if (lane == 0) {
        double b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b[i] = rint(sm[i]);   // as_tlbr: round half to even (utils/rect.py:6-13)
            tlbr_out[(size_t)w * 4 + i] = b[i];
        }
        lost_out[w] = ios_frame(b, fr.r) < 0.5 ? 1 : 0;
    }
}

__global__ __launch_bounds__(64 * WAVES) void kf_step_kernel(
    int n, const int32_t* __restrict__ slots, Hmat Hm, const double* __restrict__ klt,
    const uint8_t* __restrict__ has_klt, const double* __restrict__ mult, double* __restrict__ mean,
    double* __restrict__ cov, KFConst kf, Rect fr, double* __restrict__ tlbr_out,
    uint8_t* __restrict__ lost_out, int ops) {
    __shared__ double lds[WAVES][3 * 64 + 32];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * WAVES + wv;
    const bool live = w < n;
    const int slot = live ? slots[w] : 0;
    double* sP = lds[wv];
    double* sA = sP + 64;
    double* sB = sA + 64;
    double* sm = sB + 64;        // mean[8]
    double* sS = sm + 8;         // 16 + spare
    const int r = lane >> 3, c = lane & 7;

    sP[lane] = cov[(size_t)slot * 64 + lane];
    if (lane < 8) sm[lane] = mean[(size_t)slot * 8 + lane];
    __syncthreads();

    // ---------------- warp by homography (kalman_filter.py:227-292)
    if (ops & FM_KF_WARP) {
        const double* H = Hm.h;
        const int rb = r >> 2, cb = c >> 2;             // 0 = position block, 1 = velocity block
        const int rc_ = (r & 3) >> 1, cc_ = (c & 3) >> 1;   // corner (0 tl, 1 br)
        const int i = r & 1, j = c & 1;
        const double px = sm[2 * rc_], py = sm[2 * rc_ + 1];
        const double vx = sm[4 + 2 * rc_], vy = sm[4 + 2 * rc_ + 1];
        const double a = H[6] * px + H[7] * py + 1.;
        const double b = H[6] * vx + H[7] * vy;
        const double qx = H[0] * px + H[1] * py + H[2];
        const double qy = H[3] * px + H[4] * py + H[5];
        const double hvx = H[0] * vx + H[1] * vy;
        const double hvy = H[3] * vx + H[4] * vy;
        const double qi = i ? qy : qx, hvi = i ? hvy : hvx;
        const double h1 = H[i * 3 + j], h3j = H[6 + j];
        const double a2 = a * a;
        double jv = 0.;
        if (rc_ == cc_) {
            if (rb == cb) jv = h1 / a - qi * h3j / a2;
            else if (rb == 1 && cb == 0)
                jv = -(hvi * h3j + b * h1) / a2 + 2. * b * qi * h3j / (a2 * a);
        }
        sA[lane] = jv;   // J
        double mnew = 0.;
        if (c == 0) mnew = rb == 0 ? qi / a : hvi / a - b * qi / a2;   // lanes r*8: new mean[r]
        __syncthreads();
        double t = 0.;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sA[r * 8 + k] * sP[k * 8 + c];   // J P
        sB[lane] = t;
        __syncthreads();
        double p1 = 0.;
#pragma unroll
        for (int k = 0; k < 8; ++k) p1 += sB[r * 8 + k] * sA[c * 8 + k];  // (J P) J^T
        if (c == 0) sm[r] = mnew;
        sP[lane] = p1;
        __syncthreads();
    }

    // ---------------- predict (kalman_filter.py:308-319)
    if (ops & FM_KF_PREDICT) {
        const double bw = sm[2] - sm[0] + 1, bh = sm[3] - sm[1] + 1;
        const double std = kf.std_factor_acc * fmax(bw, bh) + kf.std_offset_acc;
        const double s2 = std * std;
        // F P
        double fp;
        if (r < 4) fp = sP[r * 8 + c] + kf.F_pos_self * sP[(r + 4) * 8 + c] +
                        kf.F_pos_other * sP[(((r + 2) & 3) + 4) * 8 + c];
        else fp = kf.F_vel * sP[r * 8 + c];
        double mnew = 0.;
        if (lane < 4) mnew = sm[lane] + kf.F_pos_self * sm[lane + 4] + kf.F_pos_other * sm[((lane + 2) & 3) + 4];
        else if (lane < 8) mnew = kf.F_vel * sm[lane];
        sA[lane] = fp;
        __syncthreads();
        double p2;
        if (c < 4) p2 = sA[r * 8 + c] + kf.F_pos_self * sA[r * 8 + c + 4] +
                        kf.F_pos_other * sA[r * 8 + ((c + 2) & 3) + 4];
        else p2 = kf.F_vel * sA[r * 8 + c];
        double q = 0.;
        if (r == c) q = r < 4 ? kf.q_pp : kf.q_vv;
        else if (r == c + 4 || c == r + 4) q = kf.q_pv;
        p2 += q * s2;
        sB[lane] = p2;
        if (lane < 8) sm[lane] = mnew;
        __syncthreads();
        sP[lane] = 0.5 * (sB[r * 8 + c] + sB[c * 8 + r]);   // ensure symmetry (:318)
        __syncthreads();
    }

    // ---------------- KLT measurement update (tracker.py:171-176), wave-uniform branch
    const bool upd = live && has_klt[live ? w : 0];
    if (ops & FM_KF_UPDATE_KLT) {
        double z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = upd ? klt[(size_t)w * 4 + i] : 0.;
        kf_update_wave(lane, sm, sP, sA, sS, z, kf.fac_klt[0], kf.fac_klt[1], kf.min_klt[0],
                       kf.min_klt[1], upd ? mult[w] : 1.0, upd);
    }
    if (live) {
        cov[(size_t)slot * 64 + lane] = sP[lane];
        if (lane < 8) mean[(size_t)slot * 8 + lane] = sm[lane];
        write_box(lane, sm, fr, tlbr_out, lost_out, w);
    }
}

__global__ __launch_bounds__(64 * WAVES) void kf_update_det_kernel(
    int n, const int32_t* __restrict__ slots, const double* __restrict__ det,
    double* __restrict__ mean, double* __restrict__ cov, KFConst kf, Rect fr,
    double* __restrict__ tlbr_out, uint8_t* __restrict__ lost_out) {
    __shared__ double lds[WAVES][3 * 64 + 32];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * WAVES + wv;
    const bool live = w < n;
    const int slot = live ? slots[w] : 0;
    double* sP = lds[wv];
    double* sA = sP + 64;
    double* sm = sA + 128;
    double* sS = sm + 8;
    sP[lane] = cov[(size_t)slot * 64 + lane];
    if (lane < 8) sm[lane] = mean[(size_t)slot * 8 + lane];
    __syncthreads();
    double z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = live ? det[(size_t)w * 4 + i] : 0.;
    // all waves of the block run the update (uniform barriers); dead waves discard the result
    kf_update_wave(lane, sm, sP, sA, sS, z, kf.fac_det[0], kf.fac_det[1], kf.min_det[0],
                   kf.min_det[1], 1.0, live);
    if (live) {
        cov[(size_t)slot * 64 + lane] = sP[lane];
        if (lane < 8) mean[(size_t)slot * 8 + lane] = sm[lane];
        write_box(lane, sm, fr, tlbr_out, lost_out, w);
    }
}

// KalmanFilter.create (kalman_filter.py:96-126): one thread per new track
__global__ void kf_create_kernel(int n, const int32_t* __restrict__ slots,
                                 const double* __restrict__ det, double* __restrict__ mean,
                                 double* __restrict__ cov, KFConst kf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i];
    const double* b = det + (size_t)i * 4;
    double* m = mean + (size_t)slot * 8;
    double* P = cov + (size_t)slot * 64;
    const double w = b[2] - b[0] + 1, h = b[3] - b[1] + 1;
    const double spw = fmax(kf.init_pos_weight * kf.fac_det[0] * w, kf.min_det[0]);
    const double sph = fmax(kf.init_pos_weight * kf.fac_det[1] * h, kf.min_det[1]);
    const double svw = fmax(kf.init_vel_weight * kf.fac_det[0] * w, kf.min_det[0]);
    const double svh = fmax(kf.init_vel_weight * kf.fac_det[1] * h, kf.min_det[1]);
    const double std[8] = {spw, sph, spw, sph, svw, svh, svw, svh};
    for (int k = 0; k < 64; ++k) P[k] = 0.;
    for (int k = 0; k < 8; ++k) {
        m[k] = k < 4 ? b[k] : 0.;
        P[k * 9] = std[k] * std[k];
    }
}

int check_slots(fm_ctx* ctx, int n, const int32_t* slots) {
    int m = -1;
    for (int i = 0; i < n; ++i) {
        if (slots[i] < 0) { fm_set_error("negative slot"); return FM_ERR_ARG; }
        if (slots[i] > m) m = slots[i];
    }
    return fm_ensure_slots(ctx, m + 1);
}

}  // namespace

extern "C" int fm_trk_create(fm_ctx* ctx, int n, const int32_t* slots, const double* det_tlbr) {
    FM_CHECK_ARG(ctx && n >= 0);
    if (n == 0) return 0;
    FM_CHECK_ARG(slots && det_tlbr && ctx->kf_set);
    int rc = check_slots(ctx, n, slots);
    if (rc) return rc;
    const size_t bs = sizeof(int32_t) * n, bb = sizeof(double) * 4 * n;
    const size_t off = (bs + 15) & ~size_t(15);
    if ((rc = ctx->io0.reserve(off + bb))) return rc;
    FM_HIP(hipStreamSynchronize(ctx->s_main));   // staging buffer reuse
    char* h = ctx->io0.host<char>();
    memcpy(h, slots, bs);
    memcpy(h + off, det_tlbr, bb);
    FM_HIP(hipMemcpyAsync(ctx->io0.d, h, off + bb, hipMemcpyHostToDevice, ctx->s_main));
    char* d = ctx->io0.dev<char>();
    hipLaunchKernelGGL(kf_create_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->s_main, n,
                       (const int32_t*)d, (const double*)(d + off), ctx->mean, ctx->cov, ctx->kf);
    FM_HIP(hipGetLastError());
    return 0;
}

extern "C" int fm_trk_step(fm_ctx* ctx, int n, const int32_t* slots, const double* H,
                           const double* klt_tlbr, const uint8_t* has_klt, const double* mult,
                           double* tlbr_out, uint8_t* lost_out) {
    return fm_trk_step_ops(ctx, FM_KF_WARP | FM_KF_PREDICT | FM_KF_UPDATE_KLT, n, slots, H, klt_tlbr,
                           has_klt, mult, tlbr_out, lost_out);
}

extern "C" int fm_trk_step_ops(fm_ctx* ctx, int ops, int n, const int32_t* slots, const double* H,
                               const double* klt_tlbr, const uint8_t* has_klt, const double* mult,
                               double* tlbr_out, uint8_t* lost_out) {
    FM_CHECK_ARG(ctx && n >= 0);
    if (n == 0) return 0;
    FM_CHECK_ARG(slots && H && klt_tlbr && has_klt && mult && tlbr_out && lost_out && ctx->kf_set);
    int rc = check_slots(ctx, n, slots);
    if (rc) return rc;
    // packed input: klt[n][4] f64 | mult[n] f64 | slots[n] i32 | has[n] u8
    const size_t o_klt = 0, o_mult = o_klt + sizeof(double) * 4 * n, o_slots = o_mult + sizeof(double) * n;
    const size_t o_has = o_slots + sizeof(int32_t) * n, in_bytes = o_has + n;
    const size_t o_lost = sizeof(double) * 4 * n, out_bytes = o_lost + n;
    if ((rc = ctx->io0.reserve(in_bytes))) return rc;
    if ((rc = ctx->io1.reserve(out_bytes))) return rc;
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    char* h = ctx->io0.host<char>();
    memcpy(h + o_klt, klt_tlbr, sizeof(double) * 4 * n);
    memcpy(h + o_mult, mult, sizeof(double) * n);
    memcpy(h + o_slots, slots, sizeof(int32_t) * n);
    memcpy(h + o_has, has_klt, n);
    // small batches: the kernel reads the pinned staging buffer and writes the pinned result buffer
    // directly (one wave per track touches each input once) -- no blit copies, one synchronise
    const bool zc = n <= FM_ZERO_COPY_TRACKS;
    if (!zc) FM_HIP(hipMemcpyAsync(ctx->io0.d, h, in_bytes, hipMemcpyHostToDevice, ctx->s_main));
    Hmat Hm;
    memcpy(Hm.h, H, sizeof(double) * 9);
    Rect fr;
    memcpy(fr.r, ctx->frame_rect, sizeof(double) * 4);
    char* d = zc ? h : ctx->io0.dev<char>();
    char* o = zc ? ctx->io1.host<char>() : ctx->io1.dev<char>();
    fm_trace_mark(ctx, ctx->s_main, 50);
    hipLaunchKernelGGL(kf_step_kernel, dim3((n + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, ctx->s_main, n,
                       (const int32_t*)(d + o_slots), Hm, (const double*)(d + o_klt),
                       (const uint8_t*)(d + o_has), (const double*)(d + o_mult), ctx->mean, ctx->cov,
                       ctx->kf, fr, (double*)o, (uint8_t*)(o + o_lost), ops);
    FM_HIP(hipGetLastError());
    fm_trace_mark(ctx, ctx->s_main, 51);
    if (!zc) FM_HIP(hipMemcpyAsync(ctx->io1.h, ctx->io1.d, out_bytes, hipMemcpyDeviceToHost, ctx->s_main));
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    memcpy(tlbr_out, ctx->io1.host<char>(), sizeof(double) * 4 * n);
    memcpy(lost_out, ctx->io1.host<char>() + o_lost, n);
    return 0;
}

extern "C" int fm_trk_update_det(fm_ctx* ctx, int n, const int32_t* slots, const double* det_tlbr,
                                 double* tlbr_out, uint8_t* lost_out) {
    FM_CHECK_ARG(ctx && n >= 0);
    if (n == 0) return 0;
    FM_CHECK_ARG(slots && det_tlbr && tlbr_out && lost_out && ctx->kf_set);
    int rc = check_slots(ctx, n, slots);
    if (rc) return rc;
    const size_t o_det = 0, o_slots = sizeof(double) * 4 * n, in_bytes = o_slots + sizeof(int32_t) * n;
    const size_t o_lost = sizeof(double) * 4 * n, out_bytes = o_lost + n;
    if ((rc = ctx->io0.reserve(in_bytes))) return rc;
    if ((rc = ctx->io1.reserve(out_bytes))) return rc;
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    char* h = ctx->io0.host<char>();
    memcpy(h + o_det, det_tlbr, sizeof(double) * 4 * n);
    memcpy(h + o_slots, slots, sizeof(int32_t) * n);
    const bool zc = n <= FM_ZERO_COPY_TRACKS;
    if (!zc) FM_HIP(hipMemcpyAsync(ctx->io0.d, h, in_bytes, hipMemcpyHostToDevice, ctx->s_main));
    Rect fr;
    memcpy(fr.r, ctx->frame_rect, sizeof(double) * 4);
    char* d = zc ? h : ctx->io0.dev<char>();
    char* o = zc ? ctx->io1.host<char>() : ctx->io1.dev<char>();
    fm_trace_mark(ctx, ctx->s_main, 52);
    hipLaunchKernelGGL(kf_update_det_kernel, dim3((n + WAVES - 1) / WAVES), dim3(64 * WAVES), 0,
                       ctx->s_main, n, (const int32_t*)(d + o_slots), (const double*)(d + o_det),
                       ctx->mean, ctx->cov, ctx->kf, fr, (double*)o, (uint8_t*)(o + o_lost));
    FM_HIP(hipGetLastError());
    fm_trace_mark(ctx, ctx->s_main, 53);
    if (!zc) FM_HIP(hipMemcpyAsync(ctx->io1.h, ctx->io1.d, out_bytes, hipMemcpyDeviceToHost, ctx->s_main));
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    memcpy(tlbr_out, ctx->io1.host<char>(), sizeof(double) * 4 * n);
    memcpy(lost_out, ctx->io1.host<char>() + o_lost, n);
    return 0;
}

// Association kernels: pairwise terms, stage cost matrices, LAP / greedy solvers (all fp64).
//
// Replaces MultiTracker._matching_cost/_iou_cost/_reid_cost (fastmot/tracker.py:314-366),
// utils/distance.py cdist/iou_dist (:17-108), KalmanFilter.motion_distance
// (kalman_filter.py:206-225,347-353), utils/matching.py fuse_motion/gate_cost (:101-116),
// scipy.optimize.linear_sum_assignment (called at utils/matching.py:27) and _greedy_match
// (utils/matching.py:74-97), find_occluded (utils/rect.py:143-157).
//
// Design: fm_assoc_prepare computes every (track, detection) term ONCE per detector frame in
// a single launch (block per track: Cholesky of the 4x4 innovation covariance in LDS, wave
// reductions over the 512-d features); the association cascade then only gathers/gates
// sub-matrices and solves them with a single-wavefront shortest-augmenting-path kernel that
// reproduces SciPy's (Crouse) scan order and tie-breaking exactly, so matched index arrays are
// identical to the reference's.  Compiled with -ffp-contract=off: cost arithmetic uses
// separate IEEE mul/add like NumPy, decisions (>, <=) see the same doubles.
//
// Roofline: T=D=50, M=512 -> 0.33 MB and 7.7 MFLOP fp64 per frame: launch-latency bound.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace {

constexpr double INF_COST = 1e5;            // utils/matching.py:7
constexpr double CHI_SQ_INV_95 = 9.4877;    // utils/matching.py:6

__device__ inline double box_area(const double* b) {   // utils/rect.py:28-32
    const double w = b[2] - b[0] + 1, h = b[3] - b[1] + 1;
    return (w <= 0 || h <= 0) ? 0. : w * h;
}

__device__ inline double iou_dist_pair(const double* a, double area_a, const double* b) {
    // utils/distance.py:98-107
    const double iw = fmin(a[2], b[2]) - fmax(a[0], b[0]) + 1;
    const double ih = fmin(a[3], b[3]) - fmax(a[1], b[1]) + 1;
    if (iw > 0 && ih > 0) {
        const double inter = iw * ih;
        const double uni = area_a + box_area(b) - inter;
        return 1. - inter / uni;
    }
    return 1.;
}

// (the boxes arrive through device-mapped HOST memory for small n: every lane walking all boxes there cost one PCIe
// read per box and lane, 29 us for 50 boxes; they are staged in LDS 256 at a time by one coalesced read instead)
__global__ __launch_bounds__(64) void occluded_kernel(int n, const double* __restrict__ tlbr, double thresh,
                                                      uint8_t* __restrict__ out) {
    __shared__ double box[256][4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    double a[4] = {0, 0, 0, 0};
    if (live) {
        const double4 v = *reinterpret_cast<const double4*>(tlbr + (size_t)i * 4);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    }
    const double area_self = box_area(a);
    uint8_t occ = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int m = min(256, n - j0);
        for (int e = threadIdx.x; e < m * 4; e += 64) (&box[0][0])[e] = tlbr[(size_t)j0 * 4 + e];
        __syncthreads();
        for (int t = 0; t < m && !occ; ++t) {
            if (j0 + t == i) continue;
            const double* b = box[t];
            const double iw = fmin(a[2], b[2]) - fmax(a[0], b[0]) + 1;
            const double ih = fmin(a[3], b[3]) - fmax(a[1], b[1]) + 1;
            if (iw > 0 && ih > 0 && iw * ih / area_self >= thresh) occ = 1;
        }
        __syncthreads();
    }
    if (live) out[i] = occ;
}

__global__ void iou_dist_kernel(int na, const double* __restrict__ a, int nb,
                                const double* __restrict__ b, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= na * nb) return;
    const int i = idx / nb, j = idx % nb;
    const double* pa = a + (size_t)i * 4;
    out[idx] = iou_dist_pair(pa, box_area(pa), b + (size_t)j * 4);
}

// ---------------------------------------------------------------------------------------
// pairwise terms: one 256-thread block per track row
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pairwise_kernel(
    int nT, int nD, int metric, int dim, const int32_t* __restrict__ slots,
    const double* __restrict__ trk_tlbr, const double* __restrict__ det_tlbr,
    const double* __restrict__ mean, const double* __restrict__ cov,
    const float* __restrict__ favg, const int32_t* __restrict__ fcnt,
    const float* __restrict__ emb, KFConst kf, const uint8_t* __restrict__ row_f32,
    double* __restrict__ feat, double* __restrict__ maha, double* __restrict__ iou,
    uint8_t* __restrict__ row_has_feat, char* __restrict__ mirror) {
    // mirror (may be null): the same four arrays at the same offsets in page-locked host memory -- the host cascade
    // (fm_assoc_cascade) reads them there, no copy is enqueued
    const size_t mat = (size_t)nT * nD;
    double* const mfeat = reinterpret_cast<double*>(mirror);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sa = reinterpret_cast<float*>(smem);                         // [dim]
    double* sd = reinterpret_cast<double*>(smem + (size_t)dim * 4);     // L[16], pm[4], tb[4], area
    const int t = blockIdx.x, tid = threadIdx.x;
    const int slot = slots[t];
    const int cnt = fcnt[slot];
    for (int k = tid; k < dim; k += 256) sa[k] = favg[(size_t)slot * dim + k];
    if (tid == 0) {
        // project (kalman_filter.py:321-336, DETECTOR) + Cholesky (kalman_filter.py:350)
        const double* m = mean + (size_t)slot * 8;
        const double* P = cov + (size_t)slot * 64;
        const double w = m[2] - m[0] + 1, h = m[3] - m[1] + 1;
        const double sw = fmax(kf.fac_det[0] * w, kf.min_det[0]);
        const double sh = fmax(kf.fac_det[1] * h, kf.min_det[1]);
        double S[4][4], L[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                S[i][j] = P[i * 8 + j];
                L[i][j] = 0.;
            }
        S[0][0] += sw * sw; S[1][1] += sh * sh; S[2][2] += sw * sw; S[3][3] += sh * sh;
        for (int j = 0; j < 4; ++j) {
            double s = S[j][j];
            for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
            L[j][j] = sqrt(s);
            for (int i = j + 1; i < 4; ++i) {
                double v = S[i][j];
                for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
                L[i][j] = v / L[j][j];
            }
        }
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < 4; ++j) sd[i * 4 + j] = L[i][j];
            sd[16 + i] = m[i];
            sd[20 + i] = trk_tlbr[(size_t)t * 4 + i];
        }
        sd[24] = box_area(trk_tlbr + (size_t)t * 4);
        row_has_feat[t] = cnt > 0 ? 1 : 0;
        if (mirror) reinterpret_cast<uint8_t*>(mfeat + 3 * mat)[t] = cnt > 0 ? 1 : 0;
    }
    __syncthreads();
    for (int d = tid; d < nD; d += 256) {
        const double* z = det_tlbr + (size_t)d * 4;
        // forward substitution  y = L^-1 (z - Hx)   (kalman_filter.py:349-353)
        double y[4];
        for (int i = 0; i < 4; ++i) {
            double s = z[i] - sd[16 + i];
            for (int k = 0; k < i; ++k) s -= sd[i * 4 + k] * y[k];
            y[i] = s / sd[i * 4 + i];
        }
        const double md = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
        const double io = iou_dist_pair(sd + 20, sd[24], z);
        maha[(size_t)t * nD + d] = md;
        iou[(size_t)t * nD + d] = io;
        if (mirror) {
            mfeat[mat + (size_t)t * nD + d] = md;
            mfeat[2 * mat + (size_t)t * nD + d] = io;
        }
    }
    // feature distance: one wave per detection column (distance.py:47-87).  Accumulators are
    // f64; each term has the type the reference's operands give it: track feature f64 x embedding
    // f32 in _matching_cost (b_norm terms stay f32 products), f32 x f32 in _reid_cost (row_f32).
    const int wv = tid >> 6, lane = tid & 63;
    const bool f32row = row_f32[t] != 0;
    if (cnt > 0) {
        for (int d = wv; d < nD; d += 4) {
            const float* b = emb + (size_t)d * dim;
            double acc0 = 0., acc1 = 0., acc2 = 0.;
            for (int k4 = lane; k4 < dim / 4; k4 += 64) {
                const float4 bv = *reinterpret_cast<const float4*>(b + k4 * 4);
                const float4 av = *reinterpret_cast<const float4*>(sa + k4 * 4);
                const float af[4] = {av.x, av.y, av.z, av.w};
                const float bf[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (metric == FM_METRIC_COSINE) {
                        acc2 += (double)(bf[e] * bf[e]);
                        if (f32row) {
                            acc0 += (double)(af[e] * bf[e]);
                            acc1 += (double)(af[e] * af[e]);
                        } else {
                            const double a = af[e];
                            acc0 += a * (double)bf[e];
                            acc1 += a * a;
                        }
                    } else if (f32row) {
                        const float df = af[e] - bf[e];
                        acc0 += (double)(df * df);
                    } else {
                        const double dd = (double)af[e] - (double)bf[e];
                        acc0 += dd * dd;
                    }
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                acc0 += __shfl_xor(acc0, off);
                acc1 += __shfl_xor(acc1, off);
                acc2 += __shfl_xor(acc2, off);
            }
            if (lane == 0) {
                double v;
                if (metric == FM_METRIC_COSINE) v = 1. - acc0 / (sqrt(acc1) * sqrt(acc2));
                else v = sqrt(acc0);
                feat[(size_t)t * nD + d] = v;
                if (mirror) mfeat[(size_t)t * nD + d] = v;
            }
        }
    } else {
        for (int d = tid; d < nD; d += 256) {
            feat[(size_t)t * nD + d] = 1.;
            if (mirror) mfeat[(size_t)t * nD + d] = 1.;
        }
    }
}

// ---------------------------------------------------------------------------------------
// stage cost matrix (gather + fuse_motion + gate_cost)
// ---------------------------------------------------------------------------------------
__global__ void stage_cost_kernel(int stage, int nr, int nc, int nD, const int32_t* __restrict__ rows,
                                  const int32_t* __restrict__ cols, const int64_t* __restrict__ rlab,
                                  const int64_t* __restrict__ dlab, const uint8_t* __restrict__ docc,
                                  const uint8_t* __restrict__ row_has_feat,
                                  const double* __restrict__ feat, const double* __restrict__ maha,
                                  const double* __restrict__ iou, double motion_weight,
                                  double max_cost, double fill_val, double* __restrict__ cost) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nr * nc) return;
    const int i = idx / nc, j = idx % nc;
    const int t = rows[i], d = cols[j];
    const size_t p = (size_t)t * nD + d;
    const bool lab_bad = rlab[i] != dlab[d];
    double c;
    if (stage == FM_STAGE_MATCHING) {
        // tracker.py:326-340 ; utils/matching.py:101-107
        const bool empty = (!row_has_feat[t]) || docc[d];
        const double f = empty ? fill_val : feat[p];
        const double md = maha[p];
        const double norm_factor = 1. / CHI_SQ_INV_95;
        const double f_weight = 1. - motion_weight;
        c = f_weight * f + (motion_weight * norm_factor) * md;
        if (md > CHI_SQ_INV_95) c = INF_COST;
        if (lab_bad || c > max_cost) c = INF_COST;
    } else if (stage == FM_STAGE_IOU) {
        c = iou[p];
        if (lab_bad || c > max_cost) c = INF_COST;   // tracker.py:351-352
    } else {
        c = feat[p];
        if (lab_bad) c = INF_COST;                   // tracker.py:363-365 (no threshold)
    }
    cost[idx] = c;
}

// ---------------------------------------------------------------------------------------
// Rectangular LAP, one wavefront.  Follows SciPy's rectangular_lsap (Crouse 2016):
// rows are added one at a time; a Dijkstra-like scan over the `remaining` column list finds
// the shortest augmenting path; ties on the path cost prefer an unassigned column (the
// LAST one in scan order), otherwise the FIRST column in scan order.  The `remaining` list
// starts reversed (nc-1 .. 0) and is compacted by swap-with-last, exactly as in SciPy, so
// the result (not just its cost) is identical.
// cost element (i, j) = cost[i * rs + j * cs]; requires nr <= nc (host transposes).
// ---------------------------------------------------------------------------------------
constexpr int LAP_CODE_BIG = 1 << 24;

// lds_mode: 2 = cost + work arrays in LDS, 1 = work arrays in LDS, 0 = everything in global.
__global__ __launch_bounds__(64) void lap_kernel(const double* __restrict__ gcost, int nr, int nc,
                                                 long rs, long cs, double* __restrict__ gwd,
                                                 int32_t* __restrict__ gwi,
                                                 int32_t* __restrict__ col4row_out,
                                                 double* __restrict__ mcost_out, int lds_mode) {
    extern __shared__ __attribute__((aligned(16))) char lap_smem[];
    const int lane = threadIdx.x;
    const size_t wd_elems = (size_t)nr + 2 * (size_t)nc;
    double* wd = lds_mode ? reinterpret_cast<double*>(lap_smem) : gwd;
    int32_t* wi = lds_mode ? reinterpret_cast<int32_t*>(lap_smem + wd_elems * 8) : gwi;
    const double* cost = gcost;
    if (lds_mode == 2) {
        const size_t wi_bytes = (4 * (size_t)nc + 2 * (size_t)nr) * 4;
        double* lc = reinterpret_cast<double*>(lap_smem + ((wd_elems * 8 + wi_bytes + 15) & ~size_t(15)));
        // re-pack as [nr][nc] row-major (handles the transposed view)
        for (int idx = lane; idx < nr * nc; idx += 64) {
            const int i = idx / nc, j = idx % nc;
            lc[idx] = gcost[(long)i * rs + (long)j * cs];
        }
        cost = lc;
        rs = nc;
        cs = 1;
    }
    double* u = wd;            // [nr]
    double* v = u + nr;        // [nc]
    double* spc = v + nc;      // [nc]
    int32_t* path = wi;        // [nc]
    int32_t* col4row = path + nc;   // [nr]
    int32_t* row4col = col4row + nr;  // [nc]
    int32_t* remaining = row4col + nc;  // [nc]
    int32_t* SR = remaining + nc;   // [nr]
    int32_t* SC = SR + nr;          // [nc]
    for (int i = lane; i < nr; i += 64) { u[i] = 0.; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.; row4col[j] = -1; path[j] = -1; }
    __syncthreads();
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    for (int cur = 0; cur < nr; ++cur) {
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        for (int j = lane; j < nc; j += 64) { SC[j] = 0; spc[j] = INF; remaining[j] = nc - j - 1; }
        __syncthreads();
        int num_remaining = nc, sink = -1, i = cur;
        double minVal = 0.;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            double best = INF;
            int bcode = -1;
            const double ui = u[i];
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = ((minVal + cost[(long)i * rs + (long)j * cs]) - ui) - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const int code = (row4col[j] == -1) ? (LAP_CODE_BIG + it) : (LAP_CODE_BIG - 1 - it);
                if (s < best || (s == best && code > bcode)) { best = s; bcode = code; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oc = __shfl_xor(bcode, off);
                if (ob < best || (ob == best && oc > bcode)) { best = ob; bcode = oc; }
            }
            // best/bcode are now wave-uniform
            if (bcode < 0 || best == INF) { sink = -2; break; }   // infeasible
            minVal = best;
            const int index = bcode >= LAP_CODE_BIG ? bcode - LAP_CODE_BIG : LAP_CODE_BIG - 1 - bcode;
            const int j = remaining[index];
            const int r4c = row4col[j];
            if (r4c == -1) sink = j; else i = r4c;
            __syncthreads();
            if (lane == 0) {
                SC[j] = 1;
                remaining[index] = remaining[num_remaining - 1];
            }
            --num_remaining;
            __syncthreads();
        }
        if (sink < 0) {   // cannot happen for finite costs; flag and stop
            if (lane == 0) col4row_out[0] = -2;
            return;
        }
        // dual update
        for (int r = lane; r < nr; r += 64)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        if (lane == 0) {
            u[cur] += minVal;
            int j = sink;
            while (true) {   // augment
                const int r = path[j];
                row4col[j] = r;
                const int tmp = col4row[r];
                col4row[r] = j;
                j = tmp;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    for (int r = lane; r < nr; r += 64) {
        col4row_out[r] = col4row[r];
        mcost_out[r] = cost[(long)r * rs + (long)col4row[r] * cs];
    }
}

// Register-resident variant for nr <= nc <= 64 (the common frame: a few dozen tracks / detections):
// lane j owns column j (v, shortest path cost, path, row4col, visited flag, its position in SciPy's
// `remaining` list -- the scan position decides ties) and lane r owns row r (u, col4row, visited flag).
// No work arrays in memory, no barriers; the argmin of every search step is a DPP reduction
// (row_shr 1/2/4/8, row_bcast 15/31 -> lane 63) and every indexed access a v_readlane with a uniform
// index -- ds_bpermute shuffles made the step ~2000 cycles.  Same arithmetic, scan order and
// tie-breaking as lap_kernel, i.e. as scipy.optimize.linear_sum_assignment.
__device__ __forceinline__ int lane_read(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_read(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// (best, bcode) <- lexicographic min over (value ascending, code descending) with the lane selected by
// the DPP control; lanes without a source see their own value (idempotent)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void lap_min_step(double& best, int& bcode) {
    const long long b = __double_as_longlong(best);
    const int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const int oc = __builtin_amdgcn_update_dpp(bcode, bcode, CTRL, ROW_MASK, 0xf, false);
    const double ob = __longlong_as_double((long long)(((unsigned long long)ohi << 32) | olo));
    if (ob < best || (ob == best && oc > bcode)) { best = ob; bcode = oc; }
}

__global__ __launch_bounds__(64) void lap64_kernel(const double* __restrict__ gcost, int nr, int nc, long rs,
                                                   long cs, int32_t* __restrict__ col4row_out,
                                                   double* __restrict__ mcost_out) {
    __shared__ double lc[64 * 64];
    const int lane = threadIdx.x;
    for (int idx = lane; idx < nr * nc; idx += 64) {
        const int i = idx / nc, j = idx % nc;
        lc[idx] = gcost[(long)i * rs + (long)j * cs];
    }
    __syncthreads();
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    const bool is_col = lane < nc;
    double u = 0., v = 0.;
    int col4row = -1, row4col = -1, path = -1;
    for (int cur = 0; cur < nr; ++cur) {
        double spc = INF;
        int SC = 0, SR = 0, pos = nc - lane - 1;
        int num_remaining = nc, sink = -1, i = cur;
        double minVal = 0.;
        while (sink == -1) {
            if (lane == i) SR = 1;
            const double ui = lane_read(u, i);
            double best = INF;
            int bcode = -1;
            if (is_col && !SC) {
                const double r = ((minVal + lc[i * nc + lane]) - ui) - v;
                if (r < spc) { path = i; spc = r; }
                best = spc;
                bcode = (row4col == -1) ? (LAP_CODE_BIG + pos) : (LAP_CODE_BIG - 1 - pos);
            }
            lap_min_step<0x111, 0xf>(best, bcode);   // row_shr:1
            lap_min_step<0x112, 0xf>(best, bcode);   // row_shr:2
            lap_min_step<0x114, 0xf>(best, bcode);   // row_shr:4
            lap_min_step<0x118, 0xf>(best, bcode);   // row_shr:8  -> lane 15 of every row
            lap_min_step<0x142, 0xa>(best, bcode);   // row_bcast:15 into rows 1, 3
            lap_min_step<0x143, 0xc>(best, bcode);   // row_bcast:31 into rows 2, 3 -> lane 63
            best = lane_read(best, 63);
            bcode = lane_read(bcode, 63);
            if (bcode < 0 || best == INF) { sink = -2; break; }   // infeasible
            minVal = best;
            const int index = bcode >= LAP_CODE_BIG ? bcode - LAP_CODE_BIG : LAP_CODE_BIG - 1 - bcode;
            const unsigned long long at_index = __ballot(is_col && !SC && pos == index);
            const unsigned long long at_last = __ballot(is_col && !SC && pos == num_remaining - 1);
            const int j = __ffsll((long long)at_index) - 1;
            const int jl = __ffsll((long long)at_last) - 1;
            const int r4c = lane_read(row4col, j);
            if (r4c == -1) sink = j; else i = r4c;
            // remaining[index] = remaining[num_remaining - 1]
            if (lane == jl && jl != j) pos = index;
            if (lane == j) SC = 1;
            --num_remaining;
        }
        if (sink < 0) {   // cannot happen for finite costs; flag and stop
            if (lane == 0) col4row_out[0] = -2;
            return;
        }
        // dual update (col4row still the assignment before augmentation)
        const double spc_c = __shfl(spc, col4row < 0 ? 0 : col4row);
        if (lane < nr && SR && lane != cur) u += minVal - spc_c;
        if (is_col && SC) v -= minVal - spc;
        if (lane == cur) u += minVal;
        int j = sink;
        while (true) {   // augment
            const int r = lane_read(path, j);
            const int tmp = lane_read(col4row, r);
            if (lane == j) row4col = r;
            if (lane == r) col4row = j;
            j = tmp;
            if (r == cur) break;
        }
    }
    if (lane < nr) {
        col4row_out[lane] = col4row;
        mcost_out[lane] = lc[lane * nc + col4row];
    }
}

// ---------------------------------------------------------------------------------------
// greedy matching (utils/matching.py:74-97): repeated first-minimum argmin over the alive
// sub-matrix; per-row minima are cached and only invalidated rows are rescanned.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void greedy_kernel(const double* __restrict__ cost, int nr, int nc,
                                                     double max_cost, double* __restrict__ rmin,
                                                     int32_t* __restrict__ rarg,
                                                     int32_t* __restrict__ alive,   // [nr] rows | [nc] cols
                                                     int32_t* __restrict__ m_rows,
                                                     int32_t* __restrict__ m_cols,
                                                     int32_t* __restrict__ n_match) {
    __shared__ double s_val[256];
    __shared__ int s_row[256];
    __shared__ int s_state[3];
    const int tid = threadIdx.x;
    int32_t* row_alive = alive;
    int32_t* col_alive = alive + nr;
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    for (int i = tid; i < nr; i += 256) { row_alive[i] = 1; rarg[i] = -2; }
    for (int j = tid; j < nc; j += 256) col_alive[j] = 1;
    if (tid == 0) s_state[0] = 0;
    __syncthreads();
    const int iters = nr < nc ? nr : nc;
    for (int it = 0; it < iters; ++it) {
        // refresh stale row minima
        for (int i = tid; i < nr; i += 256) {
            if (!row_alive[i]) continue;
            const int a = rarg[i];
            if (a == -2 || !col_alive[a]) {
                double bv = INF;
                int bj = -1;
                for (int j = 0; j < nc; ++j)
                    if (col_alive[j]) {
                        const double c = cost[(size_t)i * nc + j];
                        if (bj < 0 || c < bv) { bv = c; bj = j; }
                    }
                rmin[i] = bv;
                rarg[i] = bj;
            }
        }
        __syncthreads();
        double bv = INF;
        int bi = -1;
        for (int i = tid; i < nr; i += 256)
            if (row_alive[i] && (bi < 0 || rmin[i] < bv)) { bv = rmin[i]; bi = i; }
        s_val[tid] = bv;
        s_row[tid] = bi;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                const double ov = s_val[tid + off];
                const int oi = s_row[tid + off];
                const int mi = s_row[tid];
                if (oi >= 0 && (mi < 0 || ov < s_val[tid] || (ov == s_val[tid] && oi < mi))) {
                    s_val[tid] = ov;
                    s_row[tid] = oi;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int i = s_row[0];
            int stop = 1;
            if (i >= 0 && s_val[0] <= max_cost) {
                const int j = rarg[i];
                const int k = s_state[0];
                m_rows[k] = i;
                m_cols[k] = j;
                s_state[0] = k + 1;
                row_alive[i] = 0;
                col_alive[j] = 0;
                stop = 0;
            }
            s_state[1] = stop;
        }
        __syncthreads();
        if (s_state[1]) break;
    }
    if (tid == 0) *n_match = s_state[0];
}

// Stages host inputs for a kernel and returns the pointer the kernel must read (`src`).  Small inputs
// (zero_copy) stay in the pinned, device-mapped staging buffer and are read by the kernel over PCIe:
// a blit H2D copy costs more (~10 us of runtime overhead + a copy kernel) than the few uncached reads.
int upload(fm_ctx* ctx, DevBuf& buf, const std::vector<std::pair<const void*, size_t>>& parts,
           std::vector<size_t>& offs, char** src = nullptr, bool zero_copy = false) {
    size_t total = 0;
    offs.clear();
    for (auto& p : parts) {
        offs.push_back(total);
        total += (p.second + 15) & ~size_t(15);
    }
    int rc = buf.reserve(total ? total : 16);
    if (rc) return rc;
    FM_HIP(hipStreamSynchronize(ctx->s_main));   // host mirror reuse
    char* h = buf.host<char>();
    for (size_t i = 0; i < parts.size(); ++i)
        if (parts[i].second) memcpy(h + offs[i], parts[i].first, parts[i].second);
    if (src) *src = zero_copy ? h : buf.dev<char>();
    if (total && !(zero_copy && src)) FM_HIP(hipMemcpyAsync(buf.d, h, total, hipMemcpyHostToDevice, ctx->s_main));
    return 0;
}

// Runs the LAP kernel on a device cost matrix [nr][nc]; outputs SciPy-ordered (rows, cols).
// Outputs (col4row, cost of every assignment) are written by the kernel directly into the pinned host
// buffer: one stream synchronise, no D2H blit.  m_cost (optional): cost[m_rows[k]][m_cols[k]].
// h_cost (optional): the same matrix in pinned host memory, already complete (stream synchronised) --
// small problems are then solved by fm_lap_host (see lap_host.hip for why).
int run_lap(fm_ctx* ctx, const double* d_cost, int nr, int nc, int32_t* m_rows, int32_t* m_cols,
            int* n_match, double* m_cost = nullptr, const double* h_cost = nullptr) {
    *n_match = 0;
    if (nr == 0 || nc == 0) return 0;
    const bool transpose = nc < nr;   // SciPy works on the transposed problem when nc < nr
    const int R = transpose ? nc : nr, C = transpose ? nr : nc;
    const long rs = transpose ? 1 : nc, cs = transpose ? nc : 1;
    if (h_cost) {
        std::vector<int32_t> c4r(R);
        if (!fm_lap_host(h_cost, R, C, rs, cs, c4r.data())) {
            fm_set_error("cost matrix is infeasible");
            return FM_ERR_STATE;
        }
        if (!transpose) {
            for (int i = 0; i < R; ++i) {
                m_rows[i] = i; m_cols[i] = c4r[i];
                if (m_cost) m_cost[i] = h_cost[(size_t)i * nc + c4r[i]];
            }
        } else {
            std::vector<int> order(R);
            for (int i = 0; i < R; ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return c4r[a] < c4r[b]; });
            for (int k = 0; k < R; ++k) {
                m_rows[k] = c4r[order[k]]; m_cols[k] = order[k];
                if (m_cost) m_cost[k] = h_cost[(size_t)m_rows[k] * nc + m_cols[k]];
            }
        }
        *n_match = R;
        return 0;
    }
    const size_t wd_bytes = sizeof(double) * (R + 2 * (size_t)C);
    const size_t wi_bytes = sizeof(int32_t) * (4 * (size_t)C + 2 * (size_t)R);
    int rc = ctx->as_work.reserve(wd_bytes + wi_bytes + 64);
    if (rc) return rc;
    const size_t mc_off = (sizeof(int32_t) * (size_t)(R + 8) + 15) & ~size_t(15);
    if ((rc = ctx->as_out.reserve(mc_off + sizeof(double) * R))) return rc;
    const bool zc = FM_ZERO_COPY_TRACKS > 0;
    char* o_base = zc ? ctx->as_out.host<char>() : ctx->as_out.dev<char>();
    int32_t* o_c4r = reinterpret_cast<int32_t*>(o_base);
    double* o_mc = reinterpret_cast<double*>(o_base + mc_off);
    char* w = ctx->as_work.dev<char>();
    const size_t work_lds = ((wd_bytes + wi_bytes + 15) & ~size_t(15));
    const size_t cost_lds = sizeof(double) * (size_t)R * C;
    const size_t lds_limit = 150 * 1024;
    int lds_mode = 0;
    size_t shmem = 0;
    if (work_lds + cost_lds <= lds_limit) { lds_mode = 2; shmem = work_lds + cost_lds; }
    else if (work_lds <= lds_limit) { lds_mode = 1; shmem = work_lds; }
    if (shmem > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lap_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
            attr_set = true;
        }
    }
    if (C <= 64)
        hipLaunchKernelGGL(lap64_kernel, dim3(1), dim3(64), 0, ctx->s_main, d_cost, R, C, rs, cs, o_c4r, o_mc);
    else
        hipLaunchKernelGGL(lap_kernel, dim3(1), dim3(64), shmem, ctx->s_main, d_cost, R, C, rs, cs, (double*)w,
                           (int32_t*)(w + ((wd_bytes + 15) & ~size_t(15))), o_c4r, o_mc, lds_mode);
    FM_HIP(hipGetLastError());
    if (!zc)
        FM_HIP(hipMemcpyAsync(ctx->as_out.h, ctx->as_out.d, mc_off + sizeof(double) * R, hipMemcpyDeviceToHost,
                              ctx->s_main));
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    const int32_t* c4r = ctx->as_out.host<int32_t>();
    o_mc = reinterpret_cast<double*>(ctx->as_out.host<char>() + mc_off);
    if (c4r[0] == -2) {
        fm_set_error("cost matrix is infeasible");
        return FM_ERR_STATE;
    }
    if (!transpose) {
        for (int i = 0; i < R; ++i) {
            m_rows[i] = i; m_cols[i] = c4r[i];
            if (m_cost) m_cost[i] = o_mc[i];
        }
    } else {
        // rows of the original problem ascending: argsort(col4row)
        std::vector<int> order(R);
        for (int i = 0; i < R; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return c4r[a] < c4r[b]; });
        for (int k = 0; k < R; ++k) {
            m_rows[k] = c4r[order[k]]; m_cols[k] = order[k];
            if (m_cost) m_cost[k] = o_mc[order[k]];
        }
    }
    *n_match = R;
    return 0;
}

int run_greedy(fm_ctx* ctx, const double* d_cost, int nr, int nc, double max_cost, int32_t* m_rows,
               int32_t* m_cols, int* n_match) {
    *n_match = 0;
    if (nr == 0 || nc == 0) return 0;
    const int mn = nr < nc ? nr : nc;
    const size_t o_rarg = sizeof(double) * nr, o_alive = o_rarg + sizeof(int32_t) * nr;
    int rc = ctx->as_work.reserve(o_alive + sizeof(int32_t) * ((size_t)nr + nc) + 64);
    if (rc) return rc;
    const size_t out_bytes = sizeof(int32_t) * (2 * (size_t)mn + 4);
    if ((rc = ctx->as_out.reserve(out_bytes))) return rc;
    char* w = ctx->as_work.dev<char>();
    const bool zc = FM_ZERO_COPY_TRACKS > 0;   // pinned, written by the kernel directly
    int32_t* o = zc ? ctx->as_out.host<int32_t>() : ctx->as_out.dev<int32_t>();
    hipLaunchKernelGGL(greedy_kernel, dim3(1), dim3(256), 0, ctx->s_main, d_cost, nr, nc, max_cost,
                       (double*)w, (int32_t*)(w + o_rarg), (int32_t*)(w + o_alive), o + 4, o + 4 + mn, o);
    FM_HIP(hipGetLastError());
    if (!zc) FM_HIP(hipMemcpyAsync(ctx->as_out.h, ctx->as_out.d, out_bytes, hipMemcpyDeviceToHost, ctx->s_main));
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    const int32_t* ho = ctx->as_out.host<int32_t>();
    const int n = ho[0];
    memcpy(m_rows, ho + 4, sizeof(int32_t) * n);
    memcpy(m_cols, ho + 4 + mn, sizeof(int32_t) * n);
    *n_match = n;
    return 0;
}

}  // namespace

extern "C" int fm_find_occluded(fm_ctx* ctx, int n, const double* tlbr, double thresh, uint8_t* out) {
    FM_CHECK_ARG(ctx && n >= 0);
    if (n == 0) return 0;
    FM_CHECK_ARG(tlbr && out);
    std::vector<size_t> offs;
    char* src = nullptr;
    const bool small = n <= FM_ZERO_COPY_TRACKS / 2;
    int rc = upload(ctx, ctx->occ_in, {{tlbr, sizeof(double) * 4 * n}}, offs, &src, small);
    if (rc) return rc;
    if ((rc = ctx->occ_out.reserve(n))) return rc;
    hipLaunchKernelGGL(occluded_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->s_main, n,
                       (const double*)src, thresh, small ? ctx->occ_out.host<uint8_t>() : ctx->occ_out.dev<uint8_t>());
    FM_HIP(hipGetLastError());
    if (!small) FM_HIP(hipMemcpyAsync(ctx->occ_out.h, ctx->occ_out.d, n, hipMemcpyDeviceToHost, ctx->s_main));
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    memcpy(out, ctx->occ_out.h, n);
    return 0;
}

extern "C" int fm_iou_dist(fm_ctx* ctx, int na, const double* a, int nb, const double* b, double* out) {
    FM_CHECK_ARG(ctx && na >= 0 && nb >= 0);
    if (na == 0 || nb == 0) return 0;
    FM_CHECK_ARG(a && b && out);
    std::vector<size_t> offs;
    int rc = upload(ctx, ctx->io0, {{a, sizeof(double) * 4 * na}, {b, sizeof(double) * 4 * nb}}, offs);
    if (rc) return rc;
    const size_t ob = sizeof(double) * (size_t)na * nb;
    if ((rc = ctx->io1.reserve(ob))) return rc;
    char* d = ctx->io0.dev<char>();
    hipLaunchKernelGGL(iou_dist_kernel, dim3((na * nb + 255) / 256), dim3(256), 0, ctx->s_main, na,
                       (const double*)(d + offs[0]), nb, (const double*)(d + offs[1]), ctx->io1.dev<double>());
    FM_HIP(hipGetLastError());
    FM_HIP(hipMemcpyAsync(ctx->io1.h, ctx->io1.d, ob, hipMemcpyDeviceToHost, ctx->s_main));
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    memcpy(out, ctx->io1.h, ob);
    return 0;
}


// after_extractor: the embeddings are the batch fm_extract_async enqueued last -- the pairwise kernel is ordered behind
// the ReID network's last launch ON THE DEVICE (ev_ext_net), so the call may be made while that network still runs.
static int assoc_prepare_impl(fm_ctx* ctx, int metric, int nT, const int32_t* slots,
                              const double* trk_tlbr, const int64_t* trk_label, int nD,
                              const double* det_tlbr, const int64_t* det_label,
                              const uint8_t* det_occluded, const uint8_t* trk_feat_f32, bool after_extractor) {
    FM_CHECK_ARG(ctx && nT >= 0 && nD >= 0 && ctx->kf_set);
    FM_CHECK_ARG(metric == FM_METRIC_EUCLIDEAN || metric == FM_METRIC_COSINE);
    ctx->as_nT = nT;
    ctx->as_nD = nD;
    ctx->as_metric = metric;
    ctx->as_mirror = false;
    if (nT == 0 || nD == 0) return 0;
    FM_CHECK_ARG(slots && trk_tlbr && trk_label && det_tlbr && det_label && det_occluded && trk_feat_f32);
    FM_CHECK_ARG(nD <= ctx->emb_n);
    if (after_extractor && !ctx->ext_net_recorded) {
        fm_set_error("fm_assoc_prepare2(after_extractor): the embeddings on the device do not come from fm_extract_async");
        return FM_ERR_STATE;
    }
    for (int i = 0; i < nT; ++i) FM_CHECK_ARG(slots[i] >= 0 && slots[i] < ctx->slot_cap);
    // small problems: the terms also go to the buffer's page-locked mirror, where the host cascade reads them
    const bool mirror = FM_ZERO_COPY_TRACKS > 0 && (size_t)nT * nD <= (size_t)ctx->opt_host_lap_elems;
    // An early launch goes onto the ReID stream itself, behind the network in stream order, and reads its (few KB of)
    // inputs from the page-locked staging buffer: no cross-stream wait is parked anywhere.  Round 6 first enqueued it on
    // s_main behind hipStreamWaitEvent(ev_ext_net); with one more stream in the context that pending barrier packet made
    // every dependent launch of the streams sharing its hardware queue slower (ReID stage 0.54 -> 0.75 ms,
    // profiles/r06_schedule_patch_retest_ab.txt).  Only small problems launch early: the stage kernels of the large
    // ones run on s_main and order behind the event as before.
    const bool early = after_extractor && mirror;
    hipStream_t st = early ? ctx->s_ext : ctx->s_main;
    std::vector<size_t> offs;
    char* in = nullptr;
    int rc = upload(ctx, ctx->as_in,
                    {{slots, sizeof(int32_t) * nT}, {trk_tlbr, sizeof(double) * 4 * nT},
                     {trk_label, sizeof(int64_t) * nT}, {det_tlbr, sizeof(double) * 4 * nD},
                     {det_label, sizeof(int64_t) * nD}, {det_occluded, (size_t)nD},
                     {trk_feat_f32, (size_t)nT}}, offs, &in, early);
    if (rc) return rc;
    for (int i = 0; i < 6; ++i) ctx->as_off[i] = offs[i];
    const size_t mat = sizeof(double) * (size_t)nT * nD;
    if ((rc = ctx->as_pair.reserve(3 * mat + nT + 64))) return rc;
    char* pr = ctx->as_pair.dev<char>();
    const size_t shmem = (size_t)ctx->feat_dim * 4 + 32 * sizeof(double);
    if (after_extractor && !early) FM_HIP(hipStreamWaitEvent(ctx->s_main, ctx->ev_ext_net, 0));
    fm_trace_mark(ctx, st, 54);
    hipLaunchKernelGGL(pairwise_kernel, dim3(nT), dim3(256), shmem, st, nT, nD, metric,
                       ctx->feat_dim, (const int32_t*)(in + offs[0]), (const double*)(in + offs[1]),
                       (const double*)(in + offs[3]), ctx->mean, ctx->cov, ctx->feat_avg, ctx->feat_cnt,
                       ctx->emb, ctx->kf, (const uint8_t*)(in + offs[6]), (double*)pr, (double*)(pr + mat),
                       (double*)(pr + 2 * mat),
                       (uint8_t*)(pr + 3 * mat), mirror ? ctx->as_pair.host<char>() : nullptr);
    FM_HIP(hipGetLastError());
    fm_trace_mark(ctx, st, 55);
    if (mirror) FM_HIP(hipEventRecord(ctx->ev_pair, st));
    ctx->as_mirror = mirror;
    ctx->as_in_device = !early;
    ctx->as_in_bytes = offs[6] + (size_t)nT;
    return 0;
}

extern "C" int fm_assoc_prepare(fm_ctx* ctx, int metric, int nT, const int32_t* slots,
                                const double* trk_tlbr, const int64_t* trk_label, int nD,
                                const double* det_tlbr, const int64_t* det_label,
                                const uint8_t* det_occluded, const uint8_t* trk_feat_f32) {
    return assoc_prepare_impl(ctx, metric, nT, slots, trk_tlbr, trk_label, nD, det_tlbr, det_label, det_occluded,
                              trk_feat_f32, false);
}

extern "C" int fm_assoc_prepare2(fm_ctx* ctx, int metric, int nT, const int32_t* slots,
                                 const double* trk_tlbr, const int64_t* trk_label, int nD,
                                 const double* det_tlbr, const int64_t* det_label,
                                 const uint8_t* det_occluded, const uint8_t* trk_feat_f32, int after_extractor,
                                 int* host_cascade) {
    int rc = assoc_prepare_impl(ctx, metric, nT, slots, trk_tlbr, trk_label, nD, det_tlbr, det_label, det_occluded,
                                trk_feat_f32, after_extractor != 0);
    if (host_cascade) *host_cascade = (rc == 0 && ctx->as_mirror) ? 1 : 0;
    return rc;
}

// ---------------------------------------------------------------------------------------
// The association cascade of MultiTracker.update (tracker.py:198-248) in ONE call, on the host, for the problem sizes
// whose assignment is solved on the host anyway (fm_lap_host: measured ~5x faster than one wavefront at every size,
// profiles/r02_lap_crossover.txt).  The pairwise terms -- the arithmetic of the stage -- come from pairwise_kernel through
// as_pair's page-locked mirror; what runs here is what stage_cost_kernel does per stage (a gather, one multiply-add and
// the gates: the same IEEE doubles, this file is built with -ffp-contract=off for host and device), the solver, and the
// reference's list bookkeeping between the stages, whose ORDER decides track IDs (utils/matching.py:58-70 with Numba's
// set iteration order, utils/setorder.py).  Per-stage device launches + host wake-ups + Python round trips cost
// ~25 us per stage on a GPU the two networks keep busy (profiles/r05_pipeline_trace.txt: 214 us between the embeddings
// and the end of the step for 46 us of kernels); larger problems keep the per-stage device path (fm_assoc_stage).
// ---------------------------------------------------------------------------------------
namespace {

// list(set(range(n)) - set(matched)) as Numba's integer set iterates it (fastmot_amd/utils/setorder.py)
void unmatched_order(int n, const std::vector<uint8_t>& gone, std::vector<int>& out) {
    out.clear();
    for (int k = 0; k < n; ++k)
        if (!gone[k]) out.push_back(k);
    int size = 16;
    while (size < 2 * n) size <<= 1;
    const int min_entries = std::max(2 * (int)out.size(), 16);
    if (4 * min_entries > size || size <= 16) return;
    while ((size >> 1) >= min_entries) size >>= 1;
    const int mask = size - 1;
    if (out.empty() || out.back() <= mask) return;
    std::vector<int> table(size, -1);
    for (int k : out) {
        unsigned index = k & mask, perturb = k;
        int probes = 0;
        for (; probes < 3; ++probes) {
            if (table[index] < 0) break;
            index = (index + 1) & mask;
        }
        if (probes == 3)
            while (table[index] >= 0) {
                perturb >>= 5;
                index = (index * 5 + 1 + perturb) & mask;
            }
        table[index] = k;
    }
    out.clear();
    for (int k : table)
        if (k >= 0) out.push_back(k);
}

struct Pairwise {
    const double *feat, *maha, *iou;
    const uint8_t* has_feat;
    const int64_t *tlab, *dlab;
    const uint8_t* docc;
    int nD;
};

// one linear-assignment stage (utils/matching.py:10-30,58-70): rows[] = pairwise rows, cols = detection ids.
// matches: (index into rows[], detection id); u_rows: indices into rows[]; u_cols: detection ids.
int lap_stage(fm_ctx* ctx, const Pairwise& P, int stage, const int32_t* rows, int nr, const std::vector<int>& cols,
              double motion_weight, double max_cost, double fill_val, std::vector<double>& cost,
              std::vector<std::pair<int, int>>& matches, std::vector<int>& u_rows, std::vector<int>& u_cols) {
    const int nc = (int)cols.size();
    matches.clear();
    std::vector<uint8_t> rgone(nr, 0), cgone(nc, 0);
    std::vector<int32_t> m_rows, m_cols;
    std::vector<double> m_cost;
    int n_match = 0;
    if (nr > 0 && nc > 0) {
        cost.resize((size_t)nr * nc);
        const double norm_factor = 1. / CHI_SQ_INV_95, f_weight = 1. - motion_weight;
        for (int i = 0; i < nr; ++i) {
            const int t = rows[i];
            for (int j = 0; j < nc; ++j) {
                const int d = cols[j];
                const size_t p = (size_t)t * P.nD + d;
                const bool lab_bad = P.tlab[t] != P.dlab[d];
                double c;
                if (stage == FM_STAGE_MATCHING) {        // tracker.py:326-340 ; utils/matching.py:101-107
                    const bool empty = (!P.has_feat[t]) || P.docc[d];
                    const double f = empty ? fill_val : P.feat[p];
                    const double md = P.maha[p];
                    c = f_weight * f + (motion_weight * norm_factor) * md;
                    if (md > CHI_SQ_INV_95) c = INF_COST;
                    if (lab_bad || c > max_cost) c = INF_COST;
                } else {                                 // tracker.py:351-352
                    c = P.iou[p];
                    if (lab_bad || c > max_cost) c = INF_COST;
                }
                cost[(size_t)i * nc + j] = c;
            }
        }
        const int mn = nr < nc ? nr : nc;
        m_rows.resize(mn); m_cols.resize(mn); m_cost.resize(mn);
        int rc = run_lap(ctx, nullptr, nr, nc, m_rows.data(), m_cols.data(), &n_match, m_cost.data(), cost.data());
        if (rc) return rc;
        for (int k = 0; k < n_match; ++k) { rgone[m_rows[k]] = 1; cgone[m_cols[k]] = 1; }
    }
    std::vector<int> uc;
    unmatched_order(nr, rgone, u_rows);
    unmatched_order(nc, cgone, uc);
    u_cols.clear();
    for (int c : uc) u_cols.push_back(cols[c]);
    for (int k = 0; k < n_match; ++k) {
        if (m_cost[k] < INF_COST) matches.emplace_back(m_rows[k], cols[m_cols[k]]);
        else { u_rows.push_back(m_rows[k]); u_cols.push_back(cols[m_cols[k]]); }
    }
    return 0;
}

}  // namespace

static int cascade_run(const Pairwise& P, int nT, int nD, const fm_cascade_in* in, int32_t* out, int out_cap) {
    fm_ctx* ctx = nullptr;          // (run_lap's host branch does not touch the context)
    FM_CHECK_ARG(in && out && in->n_groups >= 0 && in->n_unconf >= 0 && in->n_hist >= 0 && nT > 0 && nD > 0);
    FM_CHECK_ARG(in->group_off && in->det_conf);
    const int n_conf = in->group_off[in->n_groups];
    FM_CHECK_ARG(n_conf >= 0 && (n_conf == 0 || (in->conf_rows && in->conf_active)));
    FM_CHECK_ARG(in->n_unconf == 0 || in->unconf_rows);
    FM_CHECK_ARG(in->n_hist == 0 || (in->hist_rows && in->hist_labels));
    for (int i = 0; i < n_conf; ++i) FM_CHECK_ARG(in->conf_rows[i] >= 0 && in->conf_rows[i] < nT);
    for (int i = 0; i < in->n_unconf; ++i) FM_CHECK_ARG(in->unconf_rows[i] >= 0 && in->unconf_rows[i] < nT);
    for (int i = 0; i < in->n_hist; ++i) FM_CHECK_ARG(in->hist_rows[i] >= 0 && in->hist_rows[i] < nT);
    FM_CHECK_ARG(out_cap >= FM_CASCADE_HEADER + 3 * (n_conf + in->n_unconf) + 3 * nD);
    std::vector<double> cost;
    std::vector<std::pair<int, int>> m;
    std::vector<int> u_rows, u_det(nD), nxt;
    for (int d = 0; d < nD; ++d) u_det[d] = d;
    int32_t* hdr = out;
    int32_t* w = out + FM_CASCADE_HEADER;
    // ---- 1st association: motion + embeddings, depth by depth (tracker.py:205-218)
    std::vector<int> u1;          // indices into conf_rows
    int n1 = 0;
    for (int g = 0; g < in->n_groups; ++g) {
        const int b = in->group_off[g], e = in->group_off[g + 1];
        if (u_det.empty()) {
            for (int i = b; i < n_conf; ++i) u1.push_back(i);
            break;
        }
        if (e == b) continue;
        int rc = lap_stage(ctx, P, FM_STAGE_MATCHING, in->conf_rows + b, e - b, u_det, in->motion_weight,
                           in->max_assoc_cost, in->fill_val, cost, m, u_rows, nxt);
        if (rc) return rc;
        for (auto& mm : m) { *w++ = in->conf_rows[b + mm.first]; *w++ = mm.second; ++n1; }
        for (int r : u_rows) u1.push_back(b + r);
        u_det.swap(nxt);
    }
    // ---- 2nd association with IoU: the active ones among the unmatched (tracker.py:220-226)
    std::vector<int32_t> act_rows;
    std::vector<int> inact;
    for (int i : u1) {
        if (in->conf_active[i]) act_rows.push_back(in->conf_rows[i]);
        else inact.push_back(in->conf_rows[i]);
    }
    int rc = lap_stage(ctx, P, FM_STAGE_IOU, act_rows.data(), (int)act_rows.size(), u_det, 0., in->max_iou_cost, 1., cost,
                       m, u_rows, nxt);
    if (rc) return rc;
    const int n2 = (int)m.size();
    for (auto& mm : m) { *w++ = act_rows[mm.first]; *w++ = mm.second; }
    std::vector<int> u2;
    for (int r : u_rows) u2.push_back(act_rows[r]);
    u_det.swap(nxt);
    // ---- 3rd association with unconfirmed tracks (tracker.py:228-231)
    rc = lap_stage(ctx, P, FM_STAGE_IOU, in->unconf_rows, in->n_unconf, u_det, 0., in->max_iou_cost, 1., cost, m, u_rows, nxt);
    if (rc) return rc;
    const int n3 = (int)m.size();
    for (auto& mm : m) { *w++ = in->unconf_rows[mm.first]; *w++ = mm.second; }
    std::vector<int> u3;
    for (int r : u_rows) u3.push_back(in->unconf_rows[r]);
    u_det.swap(nxt);
    for (int r : inact) *w++ = r;
    for (int r : u2) *w++ = r;
    for (int r : u3) *w++ = r;
    // ---- reID with the track history (tracker.py:233-240,355-366; greedy: utils/matching.py:74-97)
    std::vector<int> valid, invalid;
    for (int d : u_det) {
        if (!(in->det_conf[d] >= in->conf_thresh)) continue;
        (P.docc[d] ? invalid : valid).push_back(d);
    }
    const int nh = in->n_hist, nv = (int)valid.size();
    int n_reid = 0;
    std::vector<uint8_t> taken(nv, 0);
    if (nh > 0 && nv > 0) {
        cost.resize((size_t)nh * nv);
        for (int i = 0; i < nh; ++i)
            for (int j = 0; j < nv; ++j) {
                const int d = valid[j];
                cost[(size_t)i * nv + j] = in->hist_labels[i] != P.dlab[d] ? INF_COST : P.feat[(size_t)in->hist_rows[i] * nD + d];
            }
        std::vector<uint8_t> ralive(nh, 1);
        const int iters = nh < nv ? nh : nv;
        for (int it = 0; it < iters; ++it) {
            int bi = -1, bj = -1;
            double bv = 0.;
            for (int i = 0; i < nh; ++i) {           // np.argmin of the remaining sub-matrix: first minimum in row-major order
                if (!ralive[i]) continue;
                for (int j = 0; j < nv; ++j) {
                    if (taken[j]) continue;
                    const double c = cost[(size_t)i * nv + j];
                    if (bi < 0 || c < bv) { bv = c; bi = i; bj = j; }
                }
            }
            if (bi < 0 || !(bv <= in->max_reid_cost)) break;
            *w++ = bi; *w++ = valid[bj];
            ++n_reid;
            ralive[bi] = 0;
            taken[bj] = 1;
        }
    }
    for (int d : invalid) *w++ = d;
    int n_rest = 0;
    for (int j = 0; j < nv; ++j)
        if (!taken[j]) { *w++ = valid[j]; ++n_rest; }
    hdr[0] = n1; hdr[1] = n2; hdr[2] = n3;
    hdr[3] = (int)inact.size(); hdr[4] = (int)u2.size(); hdr[5] = (int)u3.size();
    hdr[6] = n_reid; hdr[7] = (int)invalid.size(); hdr[8] = n_rest;
    hdr[9] = (int)(w - out);
    return 0;
}

extern "C" int fm_assoc_cascade(fm_ctx* ctx, const fm_cascade_in* in, int32_t* out, int out_cap) {
    FM_CHECK_ARG(ctx);
    const int nT = ctx->as_nT, nD = ctx->as_nD;
    if (!ctx->as_mirror || nT == 0 || nD == 0) {
        fm_set_error("fm_assoc_cascade: no host-side pairwise terms (fm_assoc_prepare2 on a small problem first)");
        return FM_ERR_STATE;
    }
    FM_HIP(hipEventSynchronize(ctx->ev_pair));
    const size_t mat = (size_t)nT * nD;
    const double* pm = ctx->as_pair.host<double>();
    const char* hin = ctx->as_in.host<char>();
    Pairwise P{pm, pm + mat, pm + 2 * mat, reinterpret_cast<const uint8_t*>(pm + 3 * mat),
               reinterpret_cast<const int64_t*>(hin + ctx->as_off[2]), reinterpret_cast<const int64_t*>(hin + ctx->as_off[4]),
               reinterpret_cast<const uint8_t*>(hin + ctx->as_off[5]), nD};
    return cascade_run(P, nT, nD, in, out, out_cap);
}

extern "C" int fm_cascade_host(int nT, int nD, const double* feat, const double* maha, const double* iou,
                               const uint8_t* row_has_feat, const int64_t* trk_label, const int64_t* det_label,
                               const uint8_t* det_occluded, const fm_cascade_in* in, int32_t* out, int out_cap) {
    FM_CHECK_ARG(feat && maha && iou && row_has_feat && trk_label && det_label && det_occluded);
    Pairwise P{feat, maha, iou, row_has_feat, trk_label, det_label, det_occluded, nD};
    return cascade_run(P, nT, nD, in, out, out_cap);
}

extern "C" int fm_assoc_get_pairwise(fm_ctx* ctx, double* feat, double* maha, double* iou) {
    FM_CHECK_ARG(ctx);
    const size_t mat = sizeof(double) * (size_t)ctx->as_nT * ctx->as_nD;
    if (mat == 0) return 0;
    FM_HIP(hipStreamSynchronize(ctx->s_main));
    if (!ctx->as_in_device) FM_HIP(hipStreamSynchronize(ctx->s_ext));   // (an early launch ran there)
    char* pr = ctx->as_pair.dev<char>();
    if (feat) FM_HIP(hipMemcpy(feat, pr, mat, hipMemcpyDeviceToHost));
    if (maha) FM_HIP(hipMemcpy(maha, pr + mat, mat, hipMemcpyDeviceToHost));
    if (iou) FM_HIP(hipMemcpy(iou, pr + 2 * mat, mat, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int fm_assoc_stage(fm_ctx* ctx, int stage, int solver, int nr, const int32_t* rows, int nc,
                              const int32_t* cols, double motion_weight, double max_cost,
                              double fill_val, const int64_t* row_label_override, int32_t* m_rows,
                              int32_t* m_cols, uint8_t* match_gated, int* n_match, double* cost_out) {
    FM_CHECK_ARG(ctx && n_match && nr >= 0 && nc >= 0);
    FM_CHECK_ARG(stage >= FM_STAGE_MATCHING && stage <= FM_STAGE_REID && (solver == 0 || solver == 1));
    *n_match = 0;
    if (nr == 0 || nc == 0) return 0;
    FM_CHECK_ARG(rows && cols && m_rows && m_cols);
    const int nT = ctx->as_nT, nD = ctx->as_nD;
    for (int i = 0; i < nr; ++i) FM_CHECK_ARG(rows[i] >= 0 && rows[i] < nT);
    for (int j = 0; j < nc; ++j) FM_CHECK_ARG(cols[j] >= 0 && cols[j] < nD);
    // row labels: override or the labels given at prepare time (gathered on the host: tiny)
    std::vector<int64_t> rlab(nr);
    if (row_label_override) memcpy(rlab.data(), row_label_override, sizeof(int64_t) * nr);
    else {
        const int64_t* tl = reinterpret_cast<const int64_t*>(ctx->as_in.host<char>() + ctx->as_off[2]);
        for (int i = 0; i < nr; ++i) rlab[i] = tl[rows[i]];
    }
    std::vector<size_t> offs;
    char* si = nullptr;
    int rc = upload(ctx, ctx->as_stage_in,
                    {{rows, sizeof(int32_t) * nr}, {cols, sizeof(int32_t) * nc},
                     {rlab.data(), sizeof(int64_t) * nr}}, offs, &si,
                    FM_ZERO_COPY_TRACKS > 0 && (size_t)nr * nc <= 65536);
    if (rc) return rc;
    const size_t cbytes = sizeof(double) * (size_t)nr * nc;
    if ((rc = ctx->as_cost.reserve(cbytes))) return rc;
    // small LAP problems: the cost matrix goes straight to pinned host memory and is solved there
    const bool host_lap = solver == 0 && FM_ZERO_COPY_TRACKS > 0 && (size_t)nr * nc <= (size_t)ctx->opt_host_lap_elems;
    double* cost_dst = host_lap ? ctx->as_cost.host<double>() : ctx->as_cost.dev<double>();
    const size_t mat = sizeof(double) * (size_t)nT * nD;
    if (!ctx->as_in_device) {
        // the terms came from an early launch on the ReID stream (fm_assoc_prepare2): its inputs were read from the staging
        // buffer -- the labels / occlusion flags go to the device now, and this stream orders behind that launch
        FM_HIP(hipMemcpyAsync(ctx->as_in.d, ctx->as_in.h, ctx->as_in_bytes, hipMemcpyHostToDevice, ctx->s_main));
        FM_HIP(hipStreamWaitEvent(ctx->s_main, ctx->ev_pair, 0));
        ctx->as_in_device = true;
    }
    char* in = ctx->as_in.dev<char>();
    char* pr = ctx->as_pair.dev<char>();
    fm_trace_mark(ctx, ctx->s_main, 56);
    hipLaunchKernelGGL(stage_cost_kernel, dim3((nr * nc + 255) / 256), dim3(256), 0, ctx->s_main, stage,
                       nr, nc, nD, (const int32_t*)(si + offs[0]), (const int32_t*)(si + offs[1]),
                       (const int64_t*)(si + offs[2]), (const int64_t*)(in + ctx->as_off[4]),
                       (const uint8_t*)(in + ctx->as_off[5]), (const uint8_t*)(pr + 3 * mat),
                       (const double*)pr, (const double*)(pr + mat), (const double*)(pr + 2 * mat),
                       motion_weight, max_cost, fill_val, cost_dst);
    FM_HIP(hipGetLastError());
    fm_trace_mark(ctx, ctx->s_main, 57);
    if (host_lap) FM_HIP(hipStreamSynchronize(ctx->s_main));
    else if (cost_out)   // tests / debugging only
        FM_HIP(hipMemcpyAsync(ctx->as_cost.h, ctx->as_cost.d, cbytes, hipMemcpyDeviceToHost, ctx->s_main));
    std::vector<double> mcost(solver == 0 && match_gated ? (size_t)(nr < nc ? nr : nc) : 0);
    if (solver == 0)
        rc = run_lap(ctx, ctx->as_cost.dev<double>(), nr, nc, m_rows, m_cols, n_match,
                     mcost.empty() ? nullptr : mcost.data(), host_lap ? ctx->as_cost.host<double>() : nullptr);
    else rc = run_greedy(ctx, ctx->as_cost.dev<double>(), nr, nc, max_cost, m_rows, m_cols, n_match);
    if (rc) return rc;
    if (solver == 0 && match_gated)
        for (int k = 0; k < *n_match; ++k)   // utils/matching.py:65
            match_gated[k] = mcost[k] < INF_COST ? 0 : 1;
    if (cost_out) memcpy(cost_out, ctx->as_cost.host<double>(), cbytes);
    return 0;
}

extern "C" int fm_lap(fm_ctx* ctx, const double* cost, int nr, int nc, int32_t* m_rows,
                      int32_t* m_cols, int* n_match) {
    FM_CHECK_ARG(ctx && n_match && nr >= 0 && nc >= 0);
    *n_match = 0;
    if (nr == 0 || nc == 0) return 0;
    FM_CHECK_ARG(cost && m_rows && m_cols);
    if ((size_t)nr * nc <= (size_t)ctx->opt_host_lap_elems)
        return run_lap(ctx, nullptr, nr, nc, m_rows, m_cols, n_match, nullptr, cost);
    std::vector<size_t> offs;
    int rc = upload(ctx, ctx->as_cost, {{cost, sizeof(double) * (size_t)nr * nc}}, offs);
    if (rc) return rc;
    return run_lap(ctx, ctx->as_cost.dev<double>(), nr, nc, m_rows, m_cols, n_match);
}

extern "C" int fm_greedy(fm_ctx* ctx, const double* cost, int nr, int nc, double max_cost,
                         int32_t* m_rows, int32_t* m_cols, int* n_match) {
    FM_CHECK_ARG(ctx && n_match && nr >= 0 && nc >= 0);
    *n_match = 0;
    if (nr == 0 || nc == 0) return 0;
    FM_CHECK_ARG(cost && m_rows && m_cols);
    std::vector<size_t> offs;
    int rc = upload(ctx, ctx->as_cost, {{cost, sizeof(double) * (size_t)nr * nc}}, offs);
    if (rc) return rc;
    return run_greedy(ctx, ctx->as_cost.dev<double>(), nr, nc, max_cost, m_rows, m_cols, n_match);
}

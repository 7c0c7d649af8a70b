// Detector front/back end on the device: frame residency, preprocessing, YOLO head decode,
// score/class filter, per-class DIoU-NMS, box filter.
//
// Replaces (reference file:line, relative to /root/reference)
//   YOLODetector._preprocess        fastmot/detector.py:289-300  (CuPy zoom order=1 mode='opencv'
//                                   grid_mode=True in uint8, BGR->RGB, HWC->CHW, * 1/255)
//   YOLODetector._create_letterbox  fastmot/detector.py:302-320  (ROI, pad value 0.5)
//   CalDetection / _NewCoords       fastmot/plugins/yolo_layer.cu:127-230
//   YOLODetector._filter_dets       fastmot/detector.py:322-365
//   diou_nms                        fastmot/utils/rect.py:199-244
//
// Differences by design: the reference copies ALL candidates (22 743 x 28 B @608) to the host and
// filters there; here decode + threshold + compaction are one kernel, NMS runs on the device
// (pair matrix as bit masks + one-wave scan) and only the final detections (48 B each) cross
// PCIe.  Candidate order: the reference sorts with unstable quicksorts (ties undefined);
// the device sort is deterministic: (class asc, box_conf desc, original index asc).
// Arithmetic: decode in fp32 with fast exp like the plugin; NMS terms in the types Numba gives
// them (areas f32, box corners / IoU / DIoU f64); rounding half-to-even.
// Roofline: HBM bound -- reads (5+C)*A*sum(HW)*4 B of head tensors (7.7 MB @608/80 classes),
// frame 6.2 MB in, 608*608*8*2 B out.
#include "pixel_source.h"
#include <cmath>
#include <mutex>
#include <utility>

struct DetState {
    fm_yolo_cfg cfg{};
    bool configured = false;
    int cap = 8192;
    // Results and the post-processing buffers are double buffered and completed by events: the pass on the NEXT frame
    // can be queued behind this one (MOT.step with next_frame: the detector stream never idles) while the host still
    // has to collect this frame's detections.  Passes are collected in enqueue order.
    static constexpr int NSLOT = 2;
    float* cand[NSLOT] = {nullptr, nullptr};        // [cap][8] : x y w h box_conf class cls_prob orig_idx(as float bits)
    float* sorted[NSLOT] = {nullptr, nullptr};      // [cap][8]
    int32_t* counters[NSLOT] = {nullptr, nullptr};  // [0]=n_cand [1]=overflow [2]=n_det
    uint64_t* mask[NSLOT] = {nullptr, nullptr};     // [cap/64][cap]
    fm_det48* dets[NSLOT] = {nullptr, nullptr};     // [cap]
    bool used[NSLOT] = {false, false};
    hipEvent_t ev_dec[NSLOT] = {nullptr, nullptr};  // candidates of the pass complete (stream that produced them)
    int post_pending = -1;                          // slot whose sort + NMS has not been enqueued yet (flush_post)
    bool general_post = false;                      // the three-kernel sort / bit matrix / scan path (more than 4096 candidates,
                                                    // fm_ctx option "nms_path" = 1); default: nms_greedy_kernel
    static constexpr int GREEDY_RETRY = 1024;       // passes on the general path before the greedy kernel gets another try
    int general_passes = 0;
    int greedy_kmax = 2048, greedy_shrink = 0;      // LDS capacity (candidates) of the next greedy launches; passes since it was too large
    bool sorted_valid[NSLOT] = {false, false};      // d->sorted[slot] holds the pass's sorted rows (general path / test hook)
    static constexpr int PREFIX = 2048;        // detections copied back with the pass (more: synchronous fallback)
    fm_det48* dets_host[NSLOT] = {nullptr, nullptr};
    int32_t* counters_host[NSLOT] = {nullptr, nullptr};
    hipEvent_t ev_done[NSLOT] = {nullptr, nullptr};
    hipStream_t s_fallback = nullptr;          // redo of a pass the greedy kernel declined (collect): NOT s_up, see there
    hipEvent_t ev0[NSLOT] = {nullptr, nullptr}, ev1[NSLOT] = {nullptr, nullptr};   // bracket the network launches
    bool timed[NSLOT] = {false, false};        // ... of the passes that were timed (fm_ctx option "net_timing")
    long n_passes = 0;
    int wr = 0, rd = 0, pending = 0, last = -1;   // slot written next / collected next / passes in flight / last collected
    uint8_t* label_mask = nullptr;
    float* rows_in = nullptr;     // test hook upload
    int rows_cap = 0;
};

void fm_det_free(DetState* d) {
    if (!d) return;
    for (int i = 0; i < DetState::NSLOT; ++i) {
        for (void* q : {(void*)d->cand[i], (void*)d->sorted[i], (void*)d->counters[i], (void*)d->mask[i], (void*)d->dets[i]})
            if (q) (void)hipFree(q);
    }
    for (void* p : {(void*)d->label_mask, (void*)d->rows_in})
        if (p) (void)hipFree(p);
    if (d->s_fallback) (void)hipStreamDestroy(d->s_fallback);
    for (int i = 0; i < DetState::NSLOT; ++i) {
        if (d->dets_host[i]) (void)hipHostFree(d->dets_host[i]);
        if (d->counters_host[i]) (void)hipHostFree(d->counters_host[i]);
        for (hipEvent_t e : {d->ev_done[i], d->ev_dec[i], d->ev0[i], d->ev1[i]})
            if (e) (void)hipEventDestroy(e);
    }
    delete d;
}

namespace {

// ------------------------------------------------------------------------------------ frames
// bilinear resize in uint8 with half-pixel centres and edge clamp (cupyx zoom mode='opencv',
// grid_mode=True: src = (dst + 0.5) * (in/out) - 0.5; affine_transform order=1, mode='nearest'),
// result rounded to uint8 (rint), then BGR->RGB, * 1/255 (fp32) and stored as fp16 NHWC with the
// channel dimension padded to 8 (zeros).  Outside the letterbox ROI the input is 0.5.
__global__ void preprocess_kernel(const uint8_t* __restrict__ frame, int fw, int fh,
                                  f16* __restrict__ inp, int in_w, int in_h, int cs, int roi_x, int roi_y,
                                  int roi_w, int roi_h, int32_t* __restrict__ counters) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    // candidate counters of this frame's decode (one memset node fewer on the detector stream; the previous
    // frame's counters were copied to the host earlier on the same stream)
    if (counters && x < 4 && y == 0) counters[x] = 0;
    if (x >= in_w || y >= in_h) return;
    float rgb[3];
    det_input_pixel(frame, fw, fh, x, y, roi_x, roi_y, roi_w, roi_h, rgb);      // (pixel_source.h: shared with the fused stem)
    f16x8 o;
    o[0] = (f16)rgb[0]; o[1] = (f16)rgb[1]; o[2] = (f16)rgb[2];
#pragma unroll
    for (int e = 3; e < 8; ++e) o[e] = (f16)0.f;
    *reinterpret_cast<f16x8*>(inp + ((size_t)y * in_w + x) * cs) = o;
}

// ------------------------------------------------------------------------------------ decode
struct HeadArgs {
    const float* data;   // NHWC fp32 [H][W][cs]
    int cs, gw, gh, na, base_index;
    float anchors[2 * FM_MAX_ANCHORS];
    float scale_xy;
};

__device__ __forceinline__ float sigmoid_fast(float x) { return 1.0f / (1.0f + __expf(-x)); }

struct FilterArgs {
    const uint8_t* label_mask;
    int num_classes;
    float conf_thresh;
    double size[2], offset[2];
    float* cand;
    int32_t* counters;
    int cap;
};

// appends one candidate row if it survives detector.py:331-341 (class mask, score threshold),
// scaled to pixels: det[:4] *= (size, size); det[:2] -= offset  (in-place float32 arithmetic)
__device__ __forceinline__ void emit_candidate(const FilterArgs& fa, float bx, float by, float bw, float bh,
                                               float box_conf, int cls, float cls_prob, int orig) {
    if (cls < 0 || cls >= 128 || !fa.label_mask[cls]) return;
    const float score = box_conf * cls_prob;
    if (!(score >= fa.conf_thresh)) return;
    const int slot = atomicAdd(fa.counters, 1);
    if (slot >= fa.cap) {
        fa.counters[1] = 1;
        return;
    }
    float* r = fa.cand + (size_t)slot * 8;
    float x = (float)((double)bx * fa.size[0]);
    float y = (float)((double)by * fa.size[1]);
    r[2] = (float)((double)bw * fa.size[0]);
    r[3] = (float)((double)bh * fa.size[1]);
    r[0] = (float)((double)x - fa.offset[0]);
    r[1] = (float)((double)y - fa.offset[1]);
    r[4] = box_conf;
    r[5] = (float)cls;
    r[6] = cls_prob;
    r[7] = __int_as_float(orig);
}

struct HeadSet {
    HeadArgs h[FM_MAX_HEADS];
    int first_block[FM_MAX_HEADS + 1];   // head i owns blocks [first_block[i], first_block[i + 1])
};

// one thread per (head, anchor, cell), all heads in one launch: plugins/yolo_layer.cu:127-173 (classic) /
// :185-230 (new_coords)
__global__ void decode_kernel(HeadSet hs, FilterArgs fa, int in_w, int in_h, int new_coords) {
    int hi = 0;
#pragma unroll
    for (int i = 1; i < FM_MAX_HEADS; ++i) hi += (int)blockIdx.x >= hs.first_block[i] ? 1 : 0;
    const HeadArgs& h = hs.h[hi];
    const int idx = ((int)blockIdx.x - hs.first_block[hi]) * blockDim.x + threadIdx.x;
    const int cells = h.gw * h.gh;
    if (idx >= cells * h.na) return;
    const int a = idx / cells, cell = idx - a * cells;
    const int row = cell / h.gw, col = cell - row * h.gw;
    const int info = 5 + fa.num_classes;
    const float* p = h.data + (size_t)cell * h.cs + a * info;
    int cls = 0;
    float best = -INFINITY;
    // four class logits per (4-byte aligned) 16-byte load, scanned in ascending order with the strict comparison of the
    // reference kernel (the first maximum wins): a quarter of the dependent load -> compare steps of the one-by-one loop
    typedef float4 f4u __attribute__((aligned(4)));
    int i = 5;
    for (; i + 4 <= info; i += 4) {
        const float4 v = *reinterpret_cast<const f4u*>(p + i);
        if (v.x > best) { best = v.x; cls = i - 5; }
        if (v.y > best) { best = v.y; cls = i - 4; }
        if (v.z > best) { best = v.z; cls = i - 3; }
        if (v.w > best) { best = v.w; cls = i - 2; }
    }
    for (; i < info; ++i) {
        const float l = p[i];
        if (l > best) { best = l; cls = i - 5; }
    }
    float bx, by, bw, bh, box_prob, cls_prob;
    const float s = h.scale_xy;
    if (!new_coords) {
        cls_prob = sigmoid_fast(best);
        box_prob = sigmoid_fast(p[4]);
        bx = (col + (s * sigmoid_fast(p[0]) - (s - 1.0f) * 0.5f)) / h.gw;
        by = (row + (s * sigmoid_fast(p[1]) - (s - 1.0f) * 0.5f)) / h.gh;
        bw = __expf(p[2]) * h.anchors[2 * a + 0] / in_w;
        bh = __expf(p[3]) * h.anchors[2 * a + 1] / in_h;
    } else {
        cls_prob = best;
        box_prob = p[4];
        bx = (col + (s * p[0] - (s - 1.0f) * 0.5f)) / h.gw;
        by = (row + (s * p[1] - (s - 1.0f) * 0.5f)) / h.gh;
        bw = p[2] * p[2] * 4 * h.anchors[2 * a + 0] / in_w;
        bh = p[3] * p[3] * 4 * h.anchors[2 * a + 1] / in_h;
    }
    bx -= bw / 2;   // centre -> top-left
    by -= bh / 2;
    emit_candidate(fa, bx, by, bw, bh, box_prob, cls, cls_prob, h.base_index + idx);
}

// test hook: candidate rows given by the host ([n][7], fractions of the frame)
__global__ void rows_filter_kernel(const float* __restrict__ rows, int n, FilterArgs fa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rows + (size_t)i * 7;
    emit_candidate(fa, r[0], r[1], r[2], r[3], r[4], (int)r[5], r[6], i);
}

// ------------------------------------------------------------------------------------ sort
// rank sort by (class asc, box_conf desc, original index asc); K is read from the device counter.  Every workgroup
// ranks 256 candidates against all K: the sort keys of 256 candidates at a time are staged in LDS as ONE 64-bit integer
// each -- class (8 bits: emit_candidate admits classes < 128) | descending-order key of box_conf (32 bits) | original
// index (24 bits: fm_detect_configure / fm_filter_dets refuse more rows than that) -- and read by all lanes at the same
// address (broadcast): K^2 integer comparisons from LDS instead of K^2 dependent global reads (K = 1500: 1.24 ms ->
// ~10 us).  The confidence key is the usual order-preserving map of IEEE bits (negative values: all bits flipped,
// others: sign bit set), inverted for the descending order, so a negative or -0.0 confidence (a NEW_COORDS head without
// its logistic activation, rows from the test hook) ranks where the float comparison puts it.
constexpr int FM_SORT_INDEX_BITS = 24;
__device__ __forceinline__ uint64_t sort_key(const float* r) {
    const uint64_t cls = (uint64_t)(uint32_t)(int)r[5] & 0xffull;
    const uint32_t b = __float_as_uint(r[4] + 0.0f);                       // (-0.0 + 0.0 = +0.0: equal to +0.0, as it compares)
    const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    const uint64_t inv_score = (uint32_t)~asc;
    const uint64_t ord = (uint64_t)(uint32_t)__float_as_int(r[7]) & ((1ull << FM_SORT_INDEX_BITS) - 1);
    return (cls << 56) | (inv_score << FM_SORT_INDEX_BITS) | ord;
}

__global__ __launch_bounds__(256) void rank_sort_kernel(const float* __restrict__ cand, float* __restrict__ sorted,
                                                        const int32_t* __restrict__ counters, int cap) {
    const int K = min(counters[0], cap);
    if ((int)blockIdx.x * 256 >= K) return;
    __shared__ uint64_t keys[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < K;
    const float* ri = cand + (size_t)(live ? i : 0) * 8;
    const uint64_t ki = sort_key(ri);
    int rank = 0;
    for (int j0 = 0; j0 < K; j0 += 256) {
        const int j = j0 + threadIdx.x;
        keys[threadIdx.x] = j < K ? sort_key(cand + (size_t)j * 8) : ~0ull;      // (padding keys rank after everything)
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < 256; ++t) rank += keys[t] < ki ? 1 : 0;
        __syncthreads();
    }
    if (!live) return;
    float* o = sorted + (size_t)rank * 8;
    const float4 r0 = *reinterpret_cast<const float4*>(ri), r1 = *reinterpret_cast<const float4*>(ri + 4);
    *reinterpret_cast<float4*>(o) = r0;
    *reinterpret_cast<float4*>(o + 4) = r1;
}

// The same rank sort with ALL K keys staged in LDS at once (dynamic LDS: 8 bytes per candidate of capacity; used
// whenever that fits, i.e. up to 16384 candidates): the chunked version above pays one dependent global round trip and
// two barriers per 256 keys -- six in a row for the ~1500 candidates of the benchmark, on a GPU whose memory system is
// busy with the detector and ReID networks -- this one pays a single round trip with every load of the workgroup in
// flight, then compares out of LDS two keys per ds_read_b128 (broadcast: every lane reads the same address).
__global__ __launch_bounds__(256) void rank_sort_lds_kernel(const float* __restrict__ cand, float* __restrict__ sorted,
                                                            const int32_t* __restrict__ counters, int cap) {
    __builtin_amdgcn_s_setprio(3);         // (latency-critical side work beside the networks' bulk wavefronts)
    extern __shared__ uint64_t all_keys[];
    const int K = min(counters[0], cap);
    // a workgroup ranks 64 candidates, four lanes each: every lane counts the smaller keys in its quarter of the list
    // (four times the wavefronts of one-lane-per-candidate for the same K^2 comparisons: the chain of a lane -- LDS read,
    // two 64-bit compares, add -- is latency bound, more wavefronts hide it)
    if ((int)blockIdx.x * 64 >= K) return;
    const int Kp = (K + 7) & ~7;                                  // (padding keys rank last)
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    // this lane's own row is fetched together with the keys (one memory round trip for both, the kernel is a chain of them)
    const float* ri = cand + (size_t)min(i, K - 1) * 8;
    const float4 r0 = *reinterpret_cast<const float4*>(ri), r1 = *reinterpret_cast<const float4*>(ri + 4);
    for (int j = threadIdx.x; j < Kp; j += 256) all_keys[j] = j < K ? sort_key(cand + (size_t)j * 8) : ~0ull;
    __syncthreads();
    const uint64_t ki = all_keys[min(i, Kp - 1)];
    int rank = 0;
    const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(all_keys) + q * (Kp / 8);
#pragma unroll 8
    for (int t = 0; t < Kp / 8; ++t) {
        const ulonglong2 k = k2[t];
        rank += (k.x < ki ? 1 : 0) + (k.y < ki ? 1 : 0);
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    if (i >= K || q != 0) return;
    float* o = sorted + (size_t)rank * 8;
    *reinterpret_cast<float4*>(o) = r0;
    *reinterpret_cast<float4*>(o + 4) = r1;
}

// ------------------------------------------------------------------------------------ NMS
// suppression test of utils/rect.py:216-241 for pair (i keeps, j candidate), same class.
__device__ __forceinline__ bool diou_suppresses(const float* a, const float* b, double thresh) {
    const float area_a = a[2] * a[3], area_b = b[2] * b[3];          // float32 products
    const double abr_x = (double)(a[0] + a[2]) - 1, abr_y = (double)(a[1] + a[3]) - 1;
    const double bbr_x = (double)(b[0] + b[2]) - 1, bbr_y = (double)(b[1] + b[3]) - 1;
    const double acx = ((double)a[0] + abr_x) / 2, acy = ((double)a[1] + abr_y) / 2;
    const double bcx = ((double)b[0] + bbr_x) / 2, bcy = ((double)b[1] + bbr_y) / 2;
    const double ixmin = fmaxf(a[0], b[0]), iymin = fmaxf(a[1], b[1]);
    const double ixmax = fmin(abr_x, bbr_x), iymax = fmin(abr_y, bbr_y);
    const double iw = fmax(0., ixmax - ixmin + 1), ih = fmax(0., iymax - iymin + 1);
    const double inter = iw * ih;
    const double uni = (double)(area_a + area_b) - inter;
    const double iou = inter / uni;
    // (d / c)^0.6 >= 0, so DIoU <= IoU: a pair whose IoU does not exceed the threshold is never suppressed -- the
    // float64 pow (hundreds of instructions) is only evaluated for the few pairs that overlap that much.  A NaN IoU
    // (degenerate boxes) fails this test and takes the full path, like the reference's arithmetic.
    if (iou <= thresh) return false;
    const double exmin = fminf(a[0], b[0]), eymin = fminf(a[1], b[1]);
    const double exmax = fmax(abr_x, bbr_x), eymax = fmax(abr_y, bbr_y);
    const double ew = exmax - exmin + 1, eh = eymax - eymin + 1;
    const double c = ew * ew + eh * eh;
    const double d = (acx - bcx) * (acx - bcx) + (acy - bcy) * (acy - bcy);
    const double diou = iou - pow(d / c, 0.6);
    return !(diou <= thresh);
}

// mask[w][i] bit b = candidate j = 64*w + b (j > i, same class) is suppressed by i.  A task is one 64 x 64 block of the
// upper triangle (rows rb*64.., column word w >= rb); a workgroup takes tasks blockIdx.x, + gridDim.x, ...  (the grid
// is fixed, K is only known on the device).  Inside a task lane = row, and each of the four wavefronts tests 16 of the 64
// columns (staged in LDS once per task) and writes its quarter of the mask word: the per-lane loop over the columns is a
// dependent chain of ~60 instructions per column with divergent exits -- a quarter of it per wavefront and four times
// the wavefronts (1300 for the benchmark's 1555 candidates, the chip has 1024 SIMDs) took the kernel from 24 to 15 us
// inside the pipeline.
constexpr int NMS_MASK_GRID = 1024;
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ sorted,
                                                       const int32_t* __restrict__ counters, int cap,
                                                       double thresh, uint64_t* __restrict__ mask) {
    __builtin_amdgcn_s_setprio(3);
    const int K = min(counters[0], cap);
    const int kw = (K + 63) / 64;
    const int ntask = kw * (kw + 1) / 2;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    __shared__ float cols[64][8];
    const float cut = (float)thresh - 1e-3f;            // slack far above the float32 error of the estimate below
    for (int t = blockIdx.x; t < ntask; t += gridDim.x) {
        // t = w (w + 1) / 2 + rb, 0 <= rb <= w
        int w = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
        while (w * (w + 1) / 2 > t) --w;
        while ((w + 1) * (w + 2) / 2 <= t) ++w;
        const int rb = t - w * (w + 1) / 2;
        const int i = rb * 64 + lane;
        float a[8];
        {
            const int ii = min(i, K - 1);
            const float4 r0 = *reinterpret_cast<const float4*>(sorted + (size_t)ii * 8);
            const float4 r1 = *reinterpret_cast<const float4*>(sorted + (size_t)ii * 8 + 4);
            a[0] = r0.x; a[1] = r0.y; a[2] = r0.z; a[3] = r0.w; a[4] = r1.x; a[5] = r1.y; a[6] = r1.z; a[7] = r1.w;
        }
        if (threadIdx.x < 128) {                        // 64 columns x two float4
            const int j = min(w * 64 + (threadIdx.x >> 1), K - 1);
            *reinterpret_cast<float4*>(&cols[threadIdx.x >> 1][(threadIdx.x & 1) * 4]) =
                *reinterpret_cast<const float4*>(sorted + (size_t)j * 8 + (threadIdx.x & 1) * 4);
        }
        __syncthreads();
        uint32_t bits = 0;
        const int b0 = q * 16, nb = min(16, K - w * 64 - b0);
        const float a_x1 = a[0] + a[2], a_y1 = a[1] + a[3], a_area = a[2] * a[3];
        for (int bb = 0; bb < nb; ++bb) {
            const int b = b0 + bb;
            if (w * 64 + b <= i) continue;
            const float* cb = cols[b];
            if (cb[5] != a[5]) continue;
            // float32 estimate of the IoU (the +1 pixel convention of rect.py included): DIoU <= IoU, so a pair whose
            // estimate stays clearly below the threshold cannot be suppressed; only the rest pays for the exact float64
            // arithmetic (and its pow) of the reference
            const float iw = fminf(a_x1, cb[0] + cb[2]) - fmaxf(a[0], cb[0]);
            const float ih = fminf(a_y1, cb[1] + cb[3]) - fmaxf(a[1], cb[1]);
            if (iw <= 0.f || ih <= 0.f) continue;
            const float inter = iw * ih;
            const float uni = a_area + cb[2] * cb[3] - inter;
            if (inter <= cut * uni) continue;
            // second stage, still float32: the whole DIoU = IoU - (d / c)^0.6 with the hardware log2 / exp2 (centre
            // distance d and enclosing diagonal c as in rect.py:231-240; the -1 / +1 of the pixel convention cancel in
            // both).  Its error is a few 1e-6 absolute; a pair is decided here when the estimate is 1e-4 or more away
            // from the threshold, and only the rest -- a handful per frame -- pays for the reference's float64 arithmetic
            // and its pow().  In the dense regime of the benchmark (1500 candidates) nearly every pair that passes the
            // IoU pre-test used to take that path: ~10^6 float64 pow() per frame.
            {
                const float b_x1 = cb[0] + cb[2], b_y1 = cb[1] + cb[3];
                const float ew = fmaxf(a_x1, b_x1) - fminf(a[0], cb[0]), eh = fmaxf(a_y1, b_y1) - fminf(a[1], cb[1]);
                const float dx = 0.5f * ((a[0] + a_x1) - (cb[0] + b_x1)), dy = 0.5f * ((a[1] + a_y1) - (cb[1] + b_y1));
                const float qq = (dx * dx + dy * dy) / (ew * ew + eh * eh);
                const float est = inter / uni - (qq > 0.f ? __builtin_amdgcn_exp2f(0.6f * __builtin_amdgcn_logf(qq)) : 0.f);
                if (est > (float)thresh + 1e-4f) { bits |= (1u << bb); continue; }
                if (est < (float)thresh - 1e-4f) continue;
                // (a NaN estimate -- degenerate boxes -- fails both comparisons and takes the exact path)
            }
            if (diou_suppresses(a, cb, thresh)) bits |= (1u << bb);
        }
        // word-major (the scan reads a word of many rows at once); this wavefront's 16 bits of the word
        if (i < K) reinterpret_cast<uint16_t*>(mask)[((size_t)w * cap + i) * 4 + q] = (uint16_t)bits;
        __syncthreads();                                // (cols is reused by the next task)
    }
}

// greedy scan in sorted order + final box filter (detector.py:356-364), one workgroup.  Two chunks of 64 candidates
// (c0 = 2j, c1 = 2j + 1) per iteration j.  The bits "removed by an earlier survivor" of a chunk are the OR of its mask
// word over all SURVIVING rows before the chunk -- a gather over the word-major mask.
//   wavefronts 1..15  do all the memory work, in three groups of five (320 lanes) that own the iterations j = g (mod 3).
//                     The loads of iteration j are issued (unconditionally, row indices clamped) three iterations ahead,
//                     right after the group has published its previous one, and nothing else of that wavefront touches
//                     memory in between: the compiler's own wait (everything outstanding) costs nothing and the round
//                     trip has three iterations' time.  (One group with alternating register sets did not get there:
//                     hipcc's wait counts across the loop's back edge fall to "everything outstanding".)
//                     The group publishes iteration j WHILE wavefront 0 resolves iteration j - 1, so it only knows the
//                     survivors before chunk p0 = 2j - 2:
//                       rem[0], rem[1]   OR of mask word c0 / c1 over the surviving rows before chunk p0,
//                       seven 64 x 64 blocks: (rows p0 | p1) x (word c0 | c1), and (c0, c0), (c0, c1), (c1, c1).
//   wavefront 0       touches LDS only: it folds the rows of the previous iteration's survivors (its own registers) and
//                     of c0's survivors into rem, and resolves each chunk in scalar registers (lane b holds row b's
//                     word of the diagonal block, one step per survivor that removes anything).
// One barrier per 128 candidates, no memory round trip and no gather on the critical path.
// History: 5.1 ms -> 68 us in round 3 (inside the pipeline; every chunk still waited for the loads it had just issued, in
// both roles: profiles/r03_bench_kernel_stats.txt) -> round 4: 107 us at the benchmark's ~1000 survivors of 1555 -> 41 us
// with gather and resolve in turns (two barriers per 128) -> this structure.
// The survivors are then filtered and written in order (prefix counts over the keep bits).
__global__ __launch_bounds__(1024) void nms_scan_kernel(const float* __restrict__ sorted,
                                                       int32_t* __restrict__ counters, int cap,
                                                       const uint64_t* __restrict__ mask, double max_area,
                                                       double min_ar, fm_det48* __restrict__ dets,
                                                       fm_det48* __restrict__ dets_host, int prefix,
                                                       int32_t* __restrict__ counters_host) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ uint64_t keep[];       // [cap/64] survivors
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = min(counters[0], cap);
    const int kw = (K + 63) / 64;
    constexpr int GU = 8, GROWS = 320 * GU;  // rows whose loads are pipelined: 2560; beyond that (rare) they are fetched at use
    const int grp = wave > 0 ? (wave - 1) % 3 : 0, gl = wave > 0 ? ((wave - 1) / 3) * 64 + lane : 0;
    const int niter = (kw + 1) / 2;
    // the rows the final filter will want (candidates tid and tid + 1024) are fetched now: one memory round trip less
    // behind the scan
    float4 pre[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float* r = sorted + (size_t)min(u * 1024 + tid, max(K - 1, 0)) * 8;
        pre[u][0] = *reinterpret_cast<const float4*>(r);
        pre[u][1] = *reinterpret_cast<const float4*>(r + 4);
    }
    struct Set { uint64_t g0[GU], g1[GU]; uint64_t blk[7]; };
    auto issue = [&](int j, Set& t) {
        const int c0 = 2 * j, c1 = min(2 * j + 1, kw - 1);
        const uint64_t* col0 = mask + (size_t)c0 * cap;
        const uint64_t* col1 = mask + (size_t)c1 * cap;
        const int last = max((c0 - 2) * 64 - 1, 0);
#pragma unroll
        for (int u = 0; u < GU; ++u)
            if (u == 0 || u * 320 <= last) {                                   // (uniform: only the rows that exist)
                const int i = min(u * 320 + gl, last);
                t.g0[u] = col0[i];
                t.g1[u] = col1[i];
            }
        if (wave <= 3) {                                   // the group's first wavefront publishes the blocks
            // (rows beyond K, or before the list for j = 0, are never looked at: their survivor bits are 0)
            const int rp0 = min(max((c0 - 2) * 64 + lane, 0), K - 1), rp1 = min(max((c0 - 1) * 64 + lane, 0), K - 1);
            const int r0 = min(c0 * 64 + lane, K - 1), r1 = min(c1 * 64 + lane, K - 1);
            t.blk[0] = col0[rp0]; t.blk[1] = col1[rp0];
            t.blk[2] = col0[rp1]; t.blk[3] = col1[rp1];
            t.blk[4] = col0[r0]; t.blk[5] = col1[r0]; t.blk[6] = col1[r1];
        }
    };
    auto alive = [&](int i, int lim) -> uint64_t {         // all ones iff row i < lim survived (branch-free)
        return 0ull - (uint64_t)((i < lim ? 1u : 0u) & (uint32_t)((keep[min(i, lim - 1 < 0 ? 0 : lim - 1) >> 6] >> (i & 63)) & 1ull));
    };
    // barrier that waits for this wave's LDS traffic only: __syncthreads() would also drain the global loads in flight
    // (its fence waits for vmcnt(0)) and expose their latency in every iteration
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    __shared__ unsigned long long rem_word[2][2];          // [iteration parity][c0, c1]: removed bits, double buffered
    __shared__ uint64_t blocks[2][7][64];                  // [iteration parity][block][row]
    if (tid < 4) rem_word[tid >> 1][tid & 1] = 0;
    __syncthreads();
    if (wave > 0) {
        Set mine;
        // reduce the rows whose fate is known (before chunk 2j - 2) and hand iteration j to wavefront 0
        auto publish = [&](int j) {
            const int lim = (2 * j - 2) * 64;
            uint64_t r0 = 0, r1 = 0;                       // (the first use waits for the loads issued three iterations ago)
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                if (u * 320 >= lim) break;                 // (uniform)
                const uint64_t on = alive(u * 320 + gl, lim);
                r0 |= mine.g0[u] & on;
                r1 |= mine.g1[u] & on;
            }
            for (int i0 = GROWS; i0 < lim; i0 += 320) {                        // more than 2560 candidates: not pipelined
                const int i = i0 + gl;
                if (i < lim && ((keep[i >> 6] >> (i & 63)) & 1ull)) {
                    r0 |= mask[(size_t)(2 * j) * cap + i];
                    r1 |= mask[(size_t)min(2 * j + 1, kw - 1) * cap + i];
                }
            }
            if (r0) atomicOr(&rem_word[j & 1][0], (unsigned long long)r0);
            if (r1) atomicOr(&rem_word[j & 1][1], (unsigned long long)r1);
            if (wave <= 3) {
#pragma unroll
                for (int q = 0; q < 7; ++q) blocks[j & 1][q][lane] = mine.blk[q];
            }
        };
        if (grp < niter) issue(grp, mine);
        int turn = grp;
        if (grp == 0 && niter > 0) {
            publish(0);
            if (3 < niter) issue(3, mine);
            turn = 3;
        }
        lds_barrier();
        for (int j = 0; j < niter; ++j) {
            if (j + 1 == turn && turn < niter) {           // (wave-uniform) iteration j + 1 beside wavefront 0's iteration j
                publish(turn);
                if (turn + 3 < niter) issue(turn + 3, mine);
                turn += 3;
            }
            lds_barrier();
        }
    } else {
        // the lowest candidate not removed yet survives and removes its row's bits; repeat until none is left.  Only
        // survivors whose row removes something INSIDE the chunk need a step of the serial loop (two v_readlane and a
        // dependent scalar chain, ~100 cycles): the survivors between two of them are decided in one mask operation --
        // where most candidates survive (the benchmark's scripted heads: ~1000 of 1555) that is 40 -> ~10 steps a chunk
        auto resolve = [&](uint64_t rem, uint64_t dg) -> uint64_t {
            const uint64_t acts = __ballot(dg != 0);                           // rows that remove anything in this chunk
            uint64_t kept = 0;
            uint64_t avail = ~rem;
            for (;;) {
                const uint64_t cand = avail & acts;
                if (!cand) break;
                const int b = __builtin_ctzll(cand);
                const uint64_t row = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dg, b) |
                                     ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dg >> 32), b) << 32);
                kept |= avail & ((2ull << b) - 1ull);                          // b and the row-less survivors below it
                avail &= ~row & (~1ull << b);                                  // rows only carry bits above b
            }
            return kept | avail;
        };
        auto uniform64 = [](uint64_t v) -> uint64_t {                          // (scalar registers: wave-uniform bookkeeping)
            return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
                   ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
        };
        uint64_t kp0 = 0, kp1 = 0;                                             // survivors of the previous iteration's chunks
        lds_barrier();
        for (int j = 0; j < niter; ++j) {
            const int c0 = 2 * j, c1 = 2 * j + 1;
            unsigned long long* rw = rem_word[j & 1];
            const uint64_t (*bk)[64] = blocks[j & 1];
            // rows of the previous iteration's survivors: not known yet when the helpers reduced
            if ((kp0 >> lane) & 1ull) {
                const uint64_t a0 = bk[0][lane], a1 = bk[1][lane];
                if (a0) atomicOr(&rw[0], (unsigned long long)a0);
                if (a1) atomicOr(&rw[1], (unsigned long long)a1);
            }
            if ((kp1 >> lane) & 1ull) {
                const uint64_t b0 = bk[2][lane], b1 = bk[3][lane];
                if (b0) atomicOr(&rw[0], (unsigned long long)b0);
                if (b1) atomicOr(&rw[1], (unsigned long long)b1);
            }
            const uint64_t d0 = bk[4][lane], od = bk[5][lane], d1 = bk[6][lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint64_t rem0 = uniform64(rw[0]);
            if (c0 == kw - 1 && (K & 63)) rem0 |= ~0ull << (K & 63);           // rows beyond K do not exist
            const uint64_t kept0 = resolve(rem0, d0);
            uint64_t kept1 = 0;
            if (c1 < kw) {
                // chunk c0's survivors remove their rows' bits of word c1
                if (((kept0 >> lane) & 1ull) && od) atomicOr(&rw[1], (unsigned long long)od);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                uint64_t rem1 = uniform64(rw[1]);
                if (c1 == kw - 1 && (K & 63)) rem1 |= ~0ull << (K & 63);
                kept1 = resolve(rem1, d1);
            }
            if (lane == 0) {
                keep[c0] = kept0;
                if (c1 < kw) keep[c1] = kept1;
                rw[0] = 0;
                rw[1] = 0;
            }
            kp0 = kept0;
            kp1 = kept1;
            lds_barrier();
        }
    }
    // final filter of the survivors, in order
    __shared__ int base;
    __shared__ int wave_cnt[16];
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < K; i0 += 1024) {
        const int i = i0 + tid;
        bool ok = false;
        double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        float r[8];
        {
            float4 q0, q1;
            if (i0 == 0) { q0 = pre[0][0]; q1 = pre[0][1]; }
            else if (i0 == 1024) { q0 = pre[1][0]; q1 = pre[1][1]; }
            else {
                const float* rp = sorted + (size_t)min(i, K - 1) * 8;
                q0 = *reinterpret_cast<const float4*>(rp);
                q1 = *reinterpret_cast<const float4*>(rp + 4);
            }
            r[0] = q0.x; r[1] = q0.y; r[2] = q0.z; r[3] = q0.w; r[4] = q1.x; r[5] = q1.y; r[6] = q1.z; r[7] = q1.w;
        }
        if (i < K && ((keep[i >> 6] >> (i & 63)) & 1ull)) {
            // to_tlbr (utils/rect.py:49-57) on float64 copies of the float32 row
            const double xmin = r[0], ymin = r[1];
            t0 = rint(xmin); t1 = rint(ymin);
            t2 = rint(xmin + (double)r[2] - 1.); t3 = rint(ymin + (double)r[3] - 1.);
            const double bw = t2 - t0 + 1, bh = t3 - t1 + 1;
            const double area = (bw <= 0 || bh <= 0) ? 0. : bw * bh;
            const double ar = bw > 0 ? bh / bw : 0.;
            ok = area > 0 && area <= max_area && ar >= min_ar;
        }
        const uint64_t bal = __ballot(ok);
        if (lane == 0) wave_cnt[tid >> 6] = __builtin_popcountll(bal);
        __syncthreads();
        int off = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        for (int wv = 0; wv < (tid >> 6); ++wv) off += wave_cnt[wv];
        if (ok) {
            fm_det48& d = dets[off];
            d.tlbr[0] = t0; d.tlbr[1] = t1; d.tlbr[2] = t2; d.tlbr[3] = t3;
            d.label = (int64_t)r[5];
            d.conf = (double)(r[4] * r[6]);     // float32 product (detector.py:362)
            if (off < prefix) dets_host[off] = d;
        }
        __syncthreads();
        if (tid == 0)
            for (int wv = 0; wv < 16; ++wv) base += wave_cnt[wv];
        __syncthreads();
    }
    if (tid == 0) {
        counters[2] = base;
        counters_host[0] = counters[0]; counters_host[1] = counters[1]; counters_host[2] = base; counters_host[3] = 0;   // (final: never "declined")
    }
}

// (out of line for the greedy kernel below: the float64 pow() inside costs ~100 registers, inlined it set the register
// count of the whole kernel although a handful of pairs per frame reach it)
__device__ __attribute__((noinline)) bool diou_suppresses_call(float a0, float a1, float a2, float a3, float b0, float b1,
                                                               float b2, float b3, double thresh) {
    const float a[4] = {a0, a1, a2, a3}, b[4] = {b0, b1, b2, b3};
    return diou_suppresses(a, b, thresh);
}

// ------------------------------------------------------------------------------------ greedy NMS, survivor by survivor
// Round 4.  The bit matrix + chunked scan above cost 10 + 20 us on an idle GPU and 42 + 88 us inside the pipelined step:
// every dependent trip to memory takes microseconds there (the detector and ReID networks keep the memory system
// saturated), the scan makes one per 64 candidates, and the bit matrix does K^2 / 2 pair tests whatever the outcome.
// This kernel is the reference's greedy loop itself (rect.py:199-244) on the SORTED candidate list, one round per
// SURVIVOR, for up to 4096 candidates in one workgroup of modest footprint (8 wavefronts, <= 128 registers, a few KB .. 64 KB of LDS: it finds room on a
// busy GPU; a 1024-thread / 113 KB first version with the sort inside waited a millisecond for an empty CU):
//   * thread t keeps the sorted candidates t, t + 256, ... in registers (box, class) and their alive bits;
//   * a round: the survivor's box is broadcast from LDS, every thread tests its own alive candidates of that class
//     behind it (float32 estimates first, the reference's float64 DIoU for the close calls -- diou_suppresses), the
//     alive bits go to a 64-word bitmap by ballot, wavefront 0 finds the next alive candidate: two barriers;
//   * the survivors are filtered (to_tlbr, area, aspect ratio) and written in order at the end.
// Memory is touched three times: sorted rows in, the survivors' confidences in, detections out.  Cost ~ (survivors) x
// ~0.2 us instead of K^2 / 2 pair tests + K / 64 memory round trips: what a real detector produces (tens of objects, a few
// hundred candidates) and the benchmark's dense regime (1500 -> a handful) finish in a few us; all-disjoint boxes degrade
// linearly.  More candidates than the launch provided LDS for (kmax, chosen from the previous pass): the kernel reports
// "not handled" (counters[3]) and collect() runs the bit matrix + scan for that pass.
constexpr int NG_T = 512, NG_CPT = 8, NG_NW = NG_T / 64, NG_KMAX = NG_T * NG_CPT;
static inline size_t ng_lds_bytes(int kmax) { return (size_t)kmax * 20 + 64 * 8 + 64; }
constexpr int NG_MAX_ROUNDS = 96;      // survivors the greedy kernel resolves before it hands the pass to the bit matrix + scan

__global__ __launch_bounds__(NG_T, 4) void nms_greedy_kernel(const float* __restrict__ sorted, int32_t* __restrict__ counters,
                                                          int cap, int kmax, double thresh, double max_area, double min_ar,
                                                          fm_det48* __restrict__ dets, fm_det48* __restrict__ dets_host,
                                                          int prefix, int32_t* __restrict__ counters_host) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char ng_sm[];
    float4* sbox = reinterpret_cast<float4*>(ng_sm);                              // [kmax] boxes in sorted order
    float* scls = reinterpret_cast<float*>(ng_sm + (size_t)kmax * 16);            // [kmax] their classes
    uint64_t* alive_w = reinterpret_cast<uint64_t*>(ng_sm + (size_t)kmax * 20);   // [64] bitmap of the sorted list
    int32_t* misc = reinterpret_cast<int32_t*>(ng_sm + (size_t)kmax * 20 + 512);  // [0] current survivor, [1] base, [2 .. 2 + NG_NW) wave counts
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = min(counters[0], cap);
    if (K > kmax) {                                      // (uniform) not handled here: the host runs the general path
        if (tid == 0) {
            counters[3] = 1;
            counters_host[0] = counters[0]; counters_host[1] = counters[1]; counters_host[2] = 0; counters_host[3] = 1;
        }
        return;
    }
    // ---- the thread's share of the sorted list: candidates tid + NG_T u; wavefront w owns bitmap words w + NG_NW u
    float4 box[NG_CPT];
    float cls[NG_CPT];
    unsigned alive = 0, kept = 0;                        // bit u: candidate tid + NG_T u is alive / is a survivor
#pragma unroll
    for (int u = 0; u < NG_CPT; ++u) {
        const int j = tid + NG_T * u;
        const float* r = sorted + (size_t)min(j, max(K - 1, 0)) * 8;
        box[u] = *reinterpret_cast<const float4*>(r);
        cls[u] = r[5];
    }
#pragma unroll
    for (int u = 0; u < NG_CPT; ++u) {
        const int j = tid + NG_T * u;
        alive |= (j < K ? 1u : 0u) << u;
        if (j < K) { sbox[j] = box[u]; scls[j] = cls[u]; }
        const uint64_t bal = __ballot(j < K);
        if (lane == 0) alive_w[wave + NG_NW * u] = bal;
    }
    if (tid == 0) misc[0] = K > 0 ? 0 : -1;
    const float cut = (float)thresh - 1e-3f;
    __syncthreads();
    // ---- one round per survivor (at most NG_MAX_ROUNDS: a frame whose candidates mostly SURVIVE -- hundreds of rounds --
    // is cheaper through the bit matrix + scan; the pass is then flagged like one with too many candidates)
    for (int round = 0;; ++round) {
        const int s = __builtin_amdgcn_readfirstlane(misc[0]);
        if (s < 0) break;
        if (round >= NG_MAX_ROUNDS) {                    // (uniform)
            if (tid == 0) {
                counters[3] = 2;
                counters_host[0] = counters[0]; counters_host[1] = counters[1]; counters_host[2] = 0; counters_host[3] = 2;
            }
            return;
        }
        const float4 a4 = sbox[s];
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        const float a_x1 = a[0] + a[2], a_y1 = a[1] + a[3], a_area = a[2] * a[3];
        const float a_cls = scls[s];
        // float32 stages for the thread's candidates (unrolled), the close calls are collected in `need` and get the
        // reference's float64 arithmetic one at a time below: ONE copy of that code (pow() included)
        unsigned need = 0;
#pragma unroll
        for (int u = 0; u < NG_CPT; ++u) {
            const int j = tid + NG_T * u;
            if (j == s) kept |= 1u << u;
            if (((alive >> u) & 1u) && j > s && cls[u] == a_cls) {
                const float4 c = box[u];
                const float b_x1 = c.x + c.z, b_y1 = c.y + c.w;
                const float iw = fminf(a_x1, b_x1) - fmaxf(a[0], c.x);
                const float ih = fminf(a_y1, b_y1) - fmaxf(a[1], c.y);
                if (iw > 0.f && ih > 0.f) {
                    const float inter = iw * ih;
                    const float uni = a_area + c.z * c.w - inter;
                    if (!(inter <= cut * uni)) {
                        const float ew = fmaxf(a_x1, b_x1) - fminf(a[0], c.x), eh = fmaxf(a_y1, b_y1) - fminf(a[1], c.y);
                        const float dx = 0.5f * ((a[0] + a_x1) - (c.x + b_x1)), dy = 0.5f * ((a[1] + a_y1) - (c.y + b_y1));
                        const float q = (dx * dx + dy * dy) / (ew * ew + eh * eh);
                        const float est = inter / uni - (q > 0.f ? __builtin_amdgcn_exp2f(0.6f * __builtin_amdgcn_logf(q)) : 0.f);
                        if (est > (float)thresh + 1e-4f) alive &= ~(1u << u);
                        else if (!(est < (float)thresh - 1e-4f)) need |= 1u << u;      // (NaN lands here too)
                    }
                }
            }
        }
        while (need) {
            const int u = __builtin_ctz(need);
            need &= need - 1;
            const float4 c = sbox[tid + NG_T * u];       // (from LDS: indexing the register copies by a run-time u
                                                         //  would move them to scratch memory)
            if (diou_suppresses_call(a[0], a[1], a[2], a[3], c.x, c.y, c.z, c.w, thresh)) alive &= ~(1u << u);
        }
#pragma unroll
        for (int u = 0; u < NG_CPT; ++u) {
            const uint64_t bal = __ballot((alive >> u) & 1u);
            if (lane == 0) alive_w[wave + NG_NW * u] = bal;
        }
        __syncthreads();
        if (wave == 0) {
            // next alive candidate behind s: lane l looks at bitmap word l
            uint64_t w = alive_w[lane];
            const int ws = s >> 6;
            if (lane < ws) w = 0;
            else if (lane == ws) w &= (s & 63) == 63 ? 0ull : (~0ull << ((s & 63) + 1));
            const uint64_t nz = __ballot(w != 0);
            int next = -1;
            if (nz) {
                const int fl = __builtin_ctzll(nz);
                const uint64_t ww = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, fl) |
                                    ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), fl) << 32);
                next = fl * 64 + __builtin_ctzll(ww);
            }
            if (lane == 0) misc[0] = next;
        }
        __syncthreads();
    }
    // ---- final filter of the survivors, in sorted order (to_tlbr, area, aspect ratio: detector.py:356-364)
    if (tid == 0) misc[1] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NG_CPT; ++u) {
        if ((int)NG_T * u >= K) break;                   // (uniform)
        const int j = tid + NG_T * u;
        bool ok = false;
        double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if ((kept >> u) & 1u) {
            const float4 b = box[u];
            const double xmin = b.x, ymin = b.y;
            t0 = rint(xmin); t1 = rint(ymin);
            t2 = rint(xmin + (double)b.z - 1.); t3 = rint(ymin + (double)b.w - 1.);
            const double bw = t2 - t0 + 1, bh = t3 - t1 + 1;
            const double area = (bw <= 0 || bh <= 0) ? 0. : bw * bh;
            const double ar = bw > 0 ? bh / bw : 0.;
            ok = area > 0 && area <= max_area && ar >= min_ar;
        }
        const uint64_t bal = __ballot(ok);
        if (lane == 0) misc[2 + wave] = __builtin_popcountll(bal);
        __syncthreads();
        const int base = misc[1];
        int off = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        for (int wv = 0; wv < wave; ++wv) off += misc[2 + wv];
        if (ok) {
            const float* r = sorted + (size_t)j * 8;
            fm_det48 d;
            d.tlbr[0] = t0; d.tlbr[1] = t1; d.tlbr[2] = t2; d.tlbr[3] = t3;
            d.label = (int64_t)r[5];
            d.conf = (double)(r[4] * r[6]);     // float32 product (detector.py:362)
            dets[off] = d;
            if (off < prefix) dets_host[off] = d;
        }
        __syncthreads();
        if (tid == 0) {
            int n = base;
            for (int wv = 0; wv < NG_NW; ++wv) n += misc[2 + wv];
            misc[1] = n;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int n = misc[1];
        counters[2] = n;
        counters[3] = 0;
        counters_host[0] = counters[0]; counters_host[1] = counters[1]; counters_host[2] = n; counters_host[3] = 0;
    }
}

int ensure_det(fm_ctx* ctx) {
    if (ctx->det) return 0;
    ctx->det = new DetState();
    return 0;
}

int alloc_post(DetState* d, int cap) {
    cap = (cap + 63) & ~63;
    if (d->cand[0] && cap == d->cap) return 0;
    for (int i = 0; i < DetState::NSLOT; ++i) {
        for (void* p : {(void*)d->cand[i], (void*)d->sorted[i], (void*)d->mask[i], (void*)d->dets[i]})
            if (p) (void)hipFree(p);
        if (d->dets_host[i]) (void)hipHostFree(d->dets_host[i]);
        d->dets_host[i] = nullptr;
        d->cand[i] = d->sorted[i] = nullptr; d->mask[i] = nullptr; d->dets[i] = nullptr;
        d->used[i] = false;
    }
    d->cap = cap;
    d->wr = d->rd = d->pending = 0;
    d->last = -1;
    d->post_pending = -1;
    for (int i = 0; i < DetState::NSLOT; ++i) {
        FM_HIP(hipMalloc(&d->cand[i], sizeof(float) * 8 * cap));
        FM_HIP(hipMalloc(&d->sorted[i], sizeof(float) * 8 * cap));
        FM_HIP(hipMalloc(&d->mask[i], sizeof(uint64_t) * (size_t)cap * (cap / 64)));
        FM_HIP(hipMalloc(&d->dets[i], sizeof(fm_det48) * cap));
        FM_HIP(hipHostMalloc(&d->dets_host[i], sizeof(fm_det48) * cap, hipHostMallocDefault));
    }
    if (!d->counters[0]) {
        for (int i = 0; i < DetState::NSLOT; ++i) {
            FM_HIP(hipMalloc(&d->counters[i], sizeof(int32_t) * 4));
            FM_HIP(hipHostMalloc(&d->counters_host[i], sizeof(int32_t) * 4, hipHostMallocDefault));
            FM_HIP(hipEventCreateWithFlags(&d->ev_done[i], hipEventDisableTiming));
            FM_HIP(hipEventCreateWithFlags(&d->ev_dec[i], hipEventDisableTiming));
            FM_HIP(hipEventCreate(&d->ev0[i]));
            FM_HIP(hipEventCreate(&d->ev1[i]));
        }
        FM_HIP(hipMalloc(&d->label_mask, 128));
    }
    return 0;
}

FilterArgs filter_args(DetState* d, int slot) {
    FilterArgs fa{};
    fa.label_mask = d->label_mask;
    fa.num_classes = d->cfg.num_classes;
    fa.conf_thresh = (float)d->cfg.conf_thresh;
    fa.size[0] = d->cfg.size[0]; fa.size[1] = d->cfg.size[1];
    fa.offset[0] = d->cfg.offset[0]; fa.offset[1] = d->cfg.offset[1];
    fa.cand = d->cand[slot];
    fa.counters = d->counters[slot];
    fa.cap = d->cap;
    return fa;
}

// the general three-kernel path: sort, suppression bit matrix, chunked scan (any number of candidates up to the capacity)
static int launch_rank_sort(DetState* d, int slot, hipStream_t sp);
static int enqueue_general_post(fm_ctx* ctx, DetState* d, int slot, hipStream_t sp, bool sorted_already = false) {
    const int cap = d->cap;
    if (!sorted_already) {
        int rc = launch_rank_sort(d, slot, sp);
        if (rc) return rc;
    }
    fm_trace_mark(ctx, sp, 22);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(std::min(NMS_MASK_GRID, (cap / 64) * (cap / 64 + 1) / 2)), dim3(256), 0, sp, d->sorted[slot],
                       d->counters[slot], cap, d->cfg.nms_thresh, d->mask[slot]);
    fm_trace_mark(ctx, sp, 23);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), sizeof(uint64_t) * (cap / 64), sp, d->sorted[slot],
                       d->counters[slot], cap, d->mask[slot], d->cfg.max_area, d->cfg.min_aspect_ratio, d->dets[slot],
                       d->dets_host[slot], cap < DetState::PREFIX ? cap : DetState::PREFIX, d->counters_host[slot]);
    FM_HIP(hipGetLastError());
    return 0;
}

static int launch_rank_sort(DetState* d, int slot, hipStream_t sp) {
    const int cap = d->cap;
    static bool configured = false;
    if (!configured) {
        FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rank_sort_lds_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint64_t) * (16384 + 8))));
        configured = true;
    }
    if (cap <= 16384)
        hipLaunchKernelGGL(rank_sort_lds_kernel, dim3(cap / 64 + 1), dim3(256), sizeof(uint64_t) * (cap + 8), sp,
                           d->cand[slot], d->sorted[slot], d->counters[slot], cap);
    else
        hipLaunchKernelGGL(rank_sort_kernel, dim3(cap / 256 + 1), dim3(256), 0, sp, d->cand[slot], d->sorted[slot],
                           d->counters[slot], cap);
    FM_HIP(hipGetLastError());
    d->sorted_valid[slot] = true;
    return 0;
}

// sort + the greedy kernel (LDS for `greedy_kmax` candidates, chosen from the candidate counts of the passes collected so
// far; a pass with more is flagged by the kernel and collect() runs the bit matrix + scan for it)
static int enqueue_greedy_post(fm_ctx* ctx, DetState* d, int slot, hipStream_t sp) {
    static bool configured = false;
    if (!configured) {
        FM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_greedy_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)ng_lds_bytes(NG_KMAX)));
        configured = true;
    }
    int rc = launch_rank_sort(d, slot, sp);
    if (rc) return rc;
    fm_trace_mark(ctx, sp, 22);
    const int kmax = d->greedy_kmax < d->cap ? d->greedy_kmax : (d->cap < NG_KMAX ? d->cap : NG_KMAX);
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(1), dim3(NG_T), ng_lds_bytes(kmax), sp, d->sorted[slot], d->counters[slot],
                       d->cap, kmax, d->cfg.nms_thresh, d->cfg.max_area, d->cfg.min_aspect_ratio, d->dets[slot],
                       d->dets_host[slot], d->cap < DetState::PREFIX ? d->cap : DetState::PREFIX, d->counters_host[slot]);
    FM_HIP(hipGetLastError());
    return 0;
}

// sort + NMS + final filter + read-back of the result of the pass in `slot`, on a low-priority stream of their own
// (s_up) behind the event that marks the pass's candidates complete.  The NMS scan is one workgroup for tens of
// microseconds: on the detector stream it kept the next frame's network waiting (1500 candidates per frame: 603 -> 667
// frames/s in alternating runs with the post-processing moved off it).  The post-processing is enqueued by whichever
// comes first of: fm_frame_upload_next (behind the next frame's copy), the next pass, a collection -- in a pipelined
// run that is the collection of the PREVIOUS pass, so it sits in its stream waiting for its network (enqueueing it
// with the pass itself: 837 against 863 frames/s and a wider spread).  Which HIP streams share a hardware queue decides a lot here (the
// context drives more streams than the runtime has queues): one more stream for this work measured 612 -> 430, the
// idle streams of the extra ReID instances 400 / 215 / 215, the ReID stream itself 480, s_up at high priority 450
// (profiles/r03_pipeline_order_ab.txt).
static int flush_post(fm_ctx* ctx, DetState* d) {
    const int slot = d->post_pending;
    if (slot < 0) return 0;
    d->post_pending = -1;
    const int cap = d->cap;
    hipStream_t sp = ctx->s_up;
    FM_HIP(hipStreamWaitEvent(sp, d->ev_dec[slot], 0));
    fm_trace_mark(ctx, sp, 20);
    int rc_p = (d->general_post || ctx->opt_nms_general) ? enqueue_general_post(ctx, d, slot, sp) : enqueue_greedy_post(ctx, d, slot, sp);
    if (rc_p) return rc_p;
    // (The counters and a bounded prefix of the detections -- they are few; the rest, rare, is fetched at collection
    // time -- are written to page-locked host memory by the scan kernel itself.  As two hipMemcpyAsync they were handed
    // to a copy engine as soon as they were enqueued, i.e. while the pass they wait for still had a millisecond to run,
    // and every later device-to-host copy that landed on that engine -- the embeddings of the frame being tracked, the
    // KLT results -- waited behind them: scripts/trace_pipeline.py showed the main thread receiving its embeddings
    // 0.6 ms after the ReID network had finished, exactly when the NEXT pass's post-processing ended.  Whether it
    // happened depended on the engine the runtime picked: runs of the same binary fell into 650 or 860 frames/s.)
    FM_HIP(hipEventRecord(d->ev_done[slot], sp));
    fm_trace_mark(ctx, sp, 21);
    return 0;
}

// the candidates of slot d->wr are complete on stream `s` (real path: decode on the detector stream; test hook: the
// row filter): book the pass and leave its post-processing pending
int enqueue_post(fm_ctx* ctx, DetState* d, hipStream_t s) {
    int rc = flush_post(ctx, d);              // (a pass whose post-processing nobody flushed yet: keep the order)
    if (rc) return rc;
    if (d->pending >= DetState::NSLOT) {      // never collected (a caller that only ever enqueues): drop the oldest
        d->rd = (d->rd + 1) % DetState::NSLOT;
        --d->pending;
    }
    const int slot = d->wr;
    FM_HIP(hipEventRecord(d->ev_dec[slot], s));
    d->post_pending = slot;
    d->used[slot] = true;
    d->wr = (slot + 1) % DetState::NSLOT;
    ++d->pending;
    return 0;
}

// a pass is about to write the candidate buffers of slot d->wr on stream `s`: the post-processing of the pass that
// used the slot before (two passes ago) must be through with them
static int acquire_slot(DetState* d, hipStream_t s) {
    // (in steady state that post-processing ended a step ago: the host can see it, and a wait that is not enqueued is one
    // packet fewer on the stream whose period is the step -- every packet there costs microseconds, r06_net_timing_events_ab.txt)
    if (d->used[d->wr] && hipEventQuery(d->ev_done[d->wr]) != hipSuccess) {
        (void)hipGetLastError();                     // (hipErrorNotReady is not an error)
        FM_HIP(hipStreamWaitEvent(s, d->ev_done[d->wr], 0));
    }
    return 0;
}

int collect(fm_ctx* ctx, DetState* d, hipStream_t s, fm_det48* out, int cap_out, int* n) {
    if (d->pending == 0) {
        fm_set_error("no detector pass in flight (fm_detect_async first)");
        return FM_ERR_STATE;
    }
    const int slot = d->rd;
    int rc_f = flush_post(ctx, d);
    if (rc_f) return rc_f;
    FM_HIP(hipEventSynchronize(d->ev_done[slot]));
    if (d->counters_host[slot][3] != 0 && !d->counters_host[slot][1]) {
        // the greedy kernel did not take the pass -- more candidates than it was given LDS for (1), or more survivors than
        // its round budget (2: most candidates survive, the bit matrix + scan is the cheaper tool): the pass (sorted
        // already, candidates untouched) goes through the general path now, and so do the following ones -- for good
        // beyond NG_KMAX candidates, otherwise until the greedy kernel is tried again GREEDY_RETRY passes later
        const int why = d->counters_host[slot][3];
        // on a stream of its own: s_up may already hold the NEXT pass's post-processing, which waits for that pass's whole
        // detector network (ev_dec) -- behind it this redo would stall for a network pass.  Every buffer it
        // touches belongs to `slot`, whose earlier work is complete (ev_done above).
        if (!d->s_fallback) FM_HIP(hipStreamCreateWithFlags(&d->s_fallback, hipStreamNonBlocking));
        hipStream_t sp = d->s_fallback;
        int rc_g = enqueue_general_post(ctx, d, slot, sp, d->sorted_valid[slot]);
        if (rc_g) return rc_g;
        FM_HIP(hipStreamSynchronize(sp));
        if (why == 2 || d->counters_host[slot][0] > NG_KMAX) {
            d->general_post = true;
            d->general_passes = 0;
        }
    } else if (d->general_post && d->counters_host[slot][0] <= NG_KMAX * 3 / 4 && ++d->general_passes >= DetState::GREEDY_RETRY) {
        d->general_post = false;
    }
    {   // LDS of the next greedy launches: room for 1.25 x the largest of the recent counts, in steps of 512
        const int k = d->counters_host[slot][0];
        int want = ((k + k / 4 + 511) / 512) * 512;
        if (want < 512) want = 512;
        if (want > NG_KMAX) want = NG_KMAX;
        if (want > d->greedy_kmax) d->greedy_kmax = want;
        else if (want < d->greedy_kmax && ++d->greedy_shrink >= 64) { d->greedy_kmax = want; d->greedy_shrink = 0; }
        else if (want == d->greedy_kmax) d->greedy_shrink = 0;
    }
    d->rd = (slot + 1) % DetState::NSLOT;
    --d->pending;
    d->last = slot;
    if (d->counters_host[slot][1]) {
        fm_set_error("candidate list overflow (%d > %d): raise max_candidates", d->counters_host[slot][0], d->cap);
        return FM_ERR_STATE;
    }
    const int nd = d->counters_host[slot][2];
    if (nd > DetState::PREFIX) {
        // (the slot's device list is not reused before the pass after next is enqueued, i.e. not before this returns)
        FM_HIP(hipMemcpy(d->dets_host[slot], d->dets[slot], sizeof(fm_det48) * nd, hipMemcpyDeviceToHost));
    }
    if (nd > cap_out) {
        fm_set_error("output capacity %d < %d detections", cap_out, nd);
        return FM_ERR_ARG;
    }
    memcpy(out, d->dets_host[slot], sizeof(fm_det48) * nd);
    *n = nd;
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------- frames
extern "C" int fm_frame_configure(fm_ctx* ctx, int width, int height, int ring_size) {
    FM_CHECK_ARG(ctx && width > 0 && height > 0 && ring_size >= 0);
    FM_HIP(hipDeviceSynchronize());
    for (void* p : {(void*)ctx->frame_own, (void*)ctx->frame_own2, (void*)ctx->frame_ring})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)ctx->frame_pinned, (void*)ctx->frame_pinned2})
        if (p) (void)hipHostFree(p);
    ctx->frame_own = ctx->frame_own2 = ctx->frame_ring = ctx->frame_pinned = ctx->frame_pinned2 = nullptr;
    ctx->frame_next = nullptr;
    const size_t bytes = (size_t)width * height * 3;
    FM_HIP(hipMalloc(&ctx->frame_own, bytes + FM_FRAME_SLACK));        // (pixel_source.h load_px2 reads 8 bytes at a pixel)
    FM_HIP(hipMalloc(&ctx->frame_own2, bytes + FM_FRAME_SLACK));
    FM_HIP(hipHostMalloc(&ctx->frame_pinned, bytes, hipHostMallocDefault));
    FM_HIP(hipHostMalloc(&ctx->frame_pinned2, bytes, hipHostMallocDefault));
    if (ring_size > 0) FM_HIP(hipMalloc(&ctx->frame_ring, bytes * ring_size + FM_FRAME_SLACK));
    ctx->frame_w = width;
    ctx->frame_h = height;
    ctx->ring_size = ring_size;
    ctx->frame_cur = ctx->frame_own;
    return 0;
}

// ---- page-locked frame buffers handed to the caller (process-wide registry of their ranges)
namespace {
std::mutex g_host_mu;
std::vector<std::pair<const uint8_t*, size_t>> g_host_ranges;

bool is_pinned_range(const uint8_t* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (auto& r : g_host_ranges)
        if (p >= r.first && p + bytes <= r.first + r.second) return true;
    return false;
}
}  // namespace

extern "C" int fm_host_alloc(size_t bytes, void** out) {
    FM_CHECK_ARG(out && bytes > 0);
    void* p = nullptr;
    FM_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        g_host_ranges.emplace_back((const uint8_t*)p, bytes);
    }
    *out = p;
    return 0;
}

extern "C" int fm_host_free(void* p) {
    if (!p) return 0;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        for (size_t i = 0; i < g_host_ranges.size(); ++i)
            if (g_host_ranges[i].first == (const uint8_t*)p) {
                g_host_ranges.erase(g_host_ranges.begin() + i);
                break;
            }
    }
    FM_HIP(hipHostFree(p));
    return 0;
}

// H2D copy of a frame from page-locked memory (the copy engine; a copy KERNEL measured no faster in round 2)
static int enqueue_frame_copy(uint8_t* dst, const uint8_t* src_pinned, size_t bytes, hipStream_t s) {
    FM_HIP(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

extern "C" int fm_frame_upload(fm_ctx* ctx, const uint8_t* bgr) {
    FM_CHECK_ARG(ctx && bgr && ctx->frame_own);
    const size_t bytes = (size_t)ctx->frame_w * ctx->frame_h * 3;
    // every consumer of the previous frame must be done before it is overwritten
    FM_HIP(hipStreamSynchronize(ctx->s_det));
    FM_HIP(hipStreamSynchronize(ctx->s_ext));
    FM_HIP(hipStreamSynchronize(ctx->s_flow));
    FM_HIP(hipStreamSynchronize(ctx->s_flow2));
    const uint8_t* src = bgr;
    if (!is_pinned_range(bgr, bytes)) {
        memcpy(ctx->frame_pinned, bgr, bytes);
        src = ctx->frame_pinned;
    }
    int rc_copy = enqueue_frame_copy(ctx->frame_own, src, bytes, ctx->s_det);
    if (rc_copy) return rc_copy;
    FM_HIP(hipStreamSynchronize(ctx->s_det));   // the other streams read the frame too
    ctx->frame_cur = ctx->frame_own;
    return 0;
}

// ---- next-frame prefetch: the detector may be started on frame t+1 while frame t is still being tracked
// (MOT.step(frame, next_frame)).  The next frame lives in the second upload slot (or the ring) and becomes
// the current one with fm_frame_promote_next -- no second upload.
extern "C" int fm_frame_upload_next(fm_ctx* ctx, const uint8_t* bgr) {
    FM_CHECK_ARG(ctx && bgr && ctx->frame_own2);
    const size_t bytes = (size_t)ctx->frame_w * ctx->frame_h * 3;
    const uint8_t* src = bgr;
    if (!is_pinned_range(bgr, bytes)) {
        // previous H2D copy out of a staging buffer: its event, not the stream (a detector pass may be running)
        if (ctx->ev_next_upload) FM_HIP(hipEventSynchronize(ctx->ev_next_upload));
        memcpy(ctx->frame_pinned2, bgr, bytes);
        src = ctx->frame_pinned2;
    }
    // The previous readers of frame_own2 -- every stage of the step before the last promote, its detector pass
    // included -- are done (fm_frame_promote_next synchronised the ReID / KLT streams, that pass was collected).  The
    // copy goes to the ReID stream: that stream is idle at this point of a step (its network starts once this frame's
    // detections have been collected, long after a 6 MB copy), it is a high-priority stream, and the pass on the new
    // frame waits for the copy's event only.  On the low-priority stream that carries the post-processing the copy was
    // held back while the KLT / ReID kernels of the running step kept the high-priority queues busy, and the detector
    // -- the longest chain of a step -- started late every frame: 662 -> 780 frames/s for this move alone, 872 together
    // with MOT.step enqueueing the prefetch before it starts the KLT job (config[1]; config[4] 100 -> 158;
    // profiles/r03_pipeline_order_ab.txt holds the whole matrix, the tracker stream and a high-priority upload stream
    // included: 550-600 and 450).
    hipStream_t cs = ctx->s_ext;
    fm_trace_mark(ctx, cs, 30);
    int rc_copy = enqueue_frame_copy(ctx->frame_own2, src, bytes, cs);
    if (rc_copy) return rc_copy;
    fm_trace_mark(ctx, cs, 31);
    if (!ctx->ev_next_upload) FM_HIP(hipEventCreateWithFlags(&ctx->ev_next_upload, hipEventDisableTiming));
    FM_HIP(hipEventRecord(ctx->ev_next_upload, cs));
    if (ctx->det && (rc_copy = flush_post(ctx, ctx->det))) return rc_copy;   // see flush_post
    ctx->frame_next = ctx->frame_own2;
    return 0;
}

extern "C" int fm_frame_ring_select_next(fm_ctx* ctx, int index) {
    FM_CHECK_ARG(ctx && index >= 0 && index < ctx->ring_size);
    ctx->frame_next = ctx->frame_ring + (size_t)ctx->frame_w * ctx->frame_h * 3 * index;
    return 0;
}

extern "C" int fm_frame_promote_next(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->frame_next);
    if (ctx->frame_next == ctx->frame_own2) {
        // the upload of the prefetched frame was enqueued on the ReID stream (fm_frame_upload_next); every stream that
        // reads the frame from now on waits for that copy's event
        FM_HIP(hipStreamSynchronize(ctx->s_ext));
        FM_HIP(hipStreamSynchronize(ctx->s_flow));
        FM_HIP(hipStreamSynchronize(ctx->s_flow2));
        // (the copy ran a step ago: when the host already sees its event complete, four barrier packets -- one in front of
        // the next frame's copy on the ReID stream -- need not be enqueued at all)
        if (hipEventQuery(ctx->ev_next_upload) != hipSuccess) {
            (void)hipGetLastError();
            FM_HIP(hipStreamWaitEvent(ctx->s_ext, ctx->ev_next_upload, 0));
            FM_HIP(hipStreamWaitEvent(ctx->s_flow, ctx->ev_next_upload, 0));
            FM_HIP(hipStreamWaitEvent(ctx->s_flow2, ctx->ev_next_upload, 0));
            FM_HIP(hipStreamWaitEvent(ctx->s_main, ctx->ev_next_upload, 0));
        }
        std::swap(ctx->frame_own, ctx->frame_own2);
        std::swap(ctx->frame_pinned, ctx->frame_pinned2);
        ctx->frame_cur = ctx->frame_own;
    } else {
        ctx->frame_cur = ctx->frame_next;
    }
    ctx->frame_next = nullptr;
    return 0;
}

extern "C" int fm_frame_ring_store(fm_ctx* ctx, int index, const uint8_t* bgr) {
    FM_CHECK_ARG(ctx && bgr && index >= 0 && index < ctx->ring_size);
    const size_t bytes = (size_t)ctx->frame_w * ctx->frame_h * 3;
    FM_HIP(hipMemcpy(ctx->frame_ring + bytes * index, bgr, bytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int fm_frame_ring_select(fm_ctx* ctx, int index) {
    FM_CHECK_ARG(ctx && index >= 0 && index < ctx->ring_size);
    ctx->frame_cur = ctx->frame_ring + (size_t)ctx->frame_w * ctx->frame_h * 3 * index;
    return 0;
}

extern "C" int fm_frame_read(fm_ctx* ctx, uint8_t* bgr) {
    FM_CHECK_ARG(ctx && bgr && ctx->frame_cur);
    FM_HIP(hipDeviceSynchronize());
    FM_HIP(hipMemcpy(bgr, ctx->frame_cur, (size_t)ctx->frame_w * ctx->frame_h * 3, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------- detector
extern "C" int fm_detect_configure(fm_ctx* ctx, const fm_yolo_cfg* cfg) {
    FM_CHECK_ARG(ctx && cfg);
    FM_CHECK_ARG(cfg->n_heads >= 0 && cfg->n_heads <= FM_MAX_HEADS && cfg->num_classes > 0 && cfg->num_classes <= 128);
    {
        // the candidate sort packs a row's original index into FM_SORT_INDEX_BITS bits of its key
        long long rows = 0;
        for (int i = 0; i < cfg->n_heads; ++i) {
            FM_CHECK_ARG(cfg->n_anchors[i] > 0 && cfg->n_anchors[i] <= FM_MAX_ANCHORS && cfg->grid_w[i] > 0 && cfg->grid_h[i] > 0);
            rows += (long long)cfg->grid_w[i] * cfg->grid_h[i] * cfg->n_anchors[i];
        }
        FM_CHECK_ARG(rows < (1ll << FM_SORT_INDEX_BITS));
    }
    int rc = ensure_det(ctx);
    if (rc) return rc;
    DetState* d = ctx->det;
    FM_HIP(hipStreamSynchronize(ctx->s_det));
    FM_HIP(hipStreamSynchronize(ctx->s_up));
    d->post_pending = -1;
    d->rd = d->wr;            // a new detector: nothing of the previous one is collected any more
    d->pending = 0;
    d->cfg = *cfg;
    d->general_post = ctx->opt_nms_general != 0;
    if ((rc = alloc_post(d, cfg->max_candidates > 0 ? cfg->max_candidates : 8192))) return rc;
    FM_HIP(hipMemcpy(d->label_mask, cfg->label_mask, 128, hipMemcpyHostToDevice));
    d->configured = true;
    return 0;
}

static int enqueue_preprocess(fm_ctx* ctx, DetState* d, NetState* net, const uint8_t* frame) {
    const fm_yolo_cfg& c = d->cfg;
    FM_CHECK_ARG(frame != nullptr);
    FM_CHECK_ARG(c.input_tensor >= 0 && c.input_tensor < (int)net->tensors.size());
    const fm_tensor& t = net->tensors[c.input_tensor];
    FM_CHECK_ARG(t.h == c.in_h && t.w == c.in_w && !t.f32);
    hipLaunchKernelGGL(preprocess_kernel, dim3((c.in_w + 255) / 256, c.in_h), dim3(256), 0, ctx->s_det,
                       frame, ctx->frame_w, ctx->frame_h, (f16*)net->bufs[c.input_tensor], c.in_w,
                       c.in_h, t.c, c.roi_x, c.roi_y, c.roi_w, c.roi_h, filter_args(d, d->wr).counters);
    FM_HIP(hipGetLastError());
    return 0;
}

extern "C" int fm_detect_preprocess_only(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->det && ctx->det->configured && ctx->det_net);
    return enqueue_preprocess(ctx, ctx->det, ctx->det_net, ctx->frame_cur);
}

static int detect_async_on(fm_ctx* ctx, const uint8_t* frame);

extern "C" int fm_detect_async(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx);
    return detect_async_on(ctx, ctx->frame_cur);
}

// detector on the prefetched next frame (fm_frame_upload_next / fm_frame_ring_select_next)
extern "C" int fm_detect_async_next(fm_ctx* ctx) {
    FM_CHECK_ARG(ctx && ctx->frame_next);
    return detect_async_on(ctx, ctx->frame_next);
}

static int detect_async_on(fm_ctx* ctx, const uint8_t* frame) {
    FM_CHECK_ARG(ctx && ctx->det && ctx->det->configured && ctx->det_net);
    DetState* d = ctx->det;
    NetState* net = ctx->det_net;
    const fm_yolo_cfg& c = d->cfg;
    hipStream_t s = ctx->s_det;
    fm_trace_mark(ctx, s, 10);
    if (frame == ctx->frame_own2 && ctx->ev_next_upload) FM_HIP(hipStreamWaitEvent(s, ctx->ev_next_upload, 0));
    int rc = acquire_slot(d, s);
    if (rc) return rc;
    fm_trace_mark(ctx, s, 14);
    // Round 6: when the network begins with a stem convolution over its input tensor, that convolution computes the
    // resized / normalised pixels itself while it stages its patch (stemconv.hip, pixel_source.h): the preprocess launch
    // (15 us inside the pipeline, between two passes of the stream whose period is the step) and the input tensor's
    // write + read disappear.  Same functions, same values (tests/test_detect_gpu.py compares the head tensors).
    const bool fused = ctx->opt_fused_input && fm_net_stem_fusable(net, c.input_tensor);
    // the event pair around the network is measurement, not product: two more packets on the stream whose period is the
    // step cost 1.2 % of the frame rate (profiles/r06_net_timing_events_ab.txt) -- recorded on every N-th pass only when
    // a caller asked for it (option "net_timing" = N; bench.py samples every 4th pass)
    const bool timing = ctx->opt_net_timing > 0 && d->n_passes++ % ctx->opt_net_timing == 0;
    d->timed[d->wr] = timing;
    if (fused) {
        if (timing) FM_HIP(hipEventRecord(d->ev0[d->wr], s));
        fm_trace_mark(ctx, s, 11);
        StemSrc src{};
        src.kind = 1; src.frame = frame; src.fw = ctx->frame_w; src.fh = ctx->frame_h;
        src.roi_x = c.roi_x; src.roi_y = c.roi_y; src.roi_w = c.roi_w; src.roi_h = c.roi_h;
        src.zero4 = filter_args(d, d->wr).counters;
        if ((rc = fm_net_run_stem_from(ctx, net, src, 1))) return rc;
        fm_trace_mark(ctx, s, 15);
        net->first = 1;
        rc = fm_net_run_internal(ctx, FM_NET_DETECTOR, 1);
        net->first = 0;
        if (rc) return rc;
    } else {
        if ((rc = enqueue_preprocess(ctx, d, net, frame))) return rc;
        if (timing) FM_HIP(hipEventRecord(d->ev0[d->wr], s));
        fm_trace_mark(ctx, s, 11);
        if ((rc = fm_net_run_internal(ctx, FM_NET_DETECTOR, 1))) return rc;
    }
    if (timing) FM_HIP(hipEventRecord(d->ev1[d->wr], s));
    fm_trace_mark(ctx, s, 12);
    FilterArgs fa = filter_args(d, d->wr);      // (counters were zeroed by this frame's preprocess kernel)
    HeadSet hs{};
    int base = 0, blocks = 0;
    for (int i = 0; i < FM_MAX_HEADS + 1; ++i) hs.first_block[i] = 0x7fffffff;
    for (int i = 0; i < c.n_heads; ++i) {
        FM_CHECK_ARG(c.head_tensor[i] >= 0 && c.head_tensor[i] < (int)net->tensors.size());
        const fm_tensor& t = net->tensors[c.head_tensor[i]];
        FM_CHECK_ARG(t.f32 && t.h == c.grid_h[i] && t.w == c.grid_w[i] && c.n_anchors[i] <= FM_MAX_ANCHORS);
        HeadArgs& h = hs.h[i];
        h.data = (const float*)net->bufs[c.head_tensor[i]];
        h.cs = t.c; h.gw = c.grid_w[i]; h.gh = c.grid_h[i]; h.na = c.n_anchors[i];
        h.base_index = base;
        memcpy(h.anchors, c.anchors[i], sizeof(float) * 2 * FM_MAX_ANCHORS);
        h.scale_xy = c.scale_xy[i];
        const int n = h.gw * h.gh * h.na;
        hs.first_block[i] = blocks;
        blocks += (n + 255) / 256;
        base += n;
    }
    if (blocks) hipLaunchKernelGGL(decode_kernel, dim3(blocks), dim3(256), 0, s, hs, fa, c.in_w, c.in_h, c.new_coords);
    FM_HIP(hipGetLastError());
    fm_trace_mark(ctx, s, 13);
    return enqueue_post(ctx, d, s);
}

extern "C" int fm_detect_sync(fm_ctx* ctx, fm_det48* out, int cap, int* n) {
    FM_CHECK_ARG(ctx && ctx->det && ctx->det->configured && out && n);
    return collect(ctx, ctx->det, ctx->s_det, out, cap, n);
}

extern "C" int fm_filter_dets(fm_ctx* ctx, const float* rows, int n, fm_det48* out, int cap, int* n_out) {
    FM_CHECK_ARG(ctx && ctx->det && ctx->det->configured && n >= 0 && out && n_out);
    FM_CHECK_ARG(n < (1 << FM_SORT_INDEX_BITS));          // (the sort key's index field)
    DetState* d = ctx->det;
    hipStream_t s = ctx->s_det;
    FM_HIP(hipStreamSynchronize(s));
    FM_HIP(hipStreamSynchronize(ctx->s_up));
    d->post_pending = -1;
    d->rd = d->wr;            // (test hook: passes nobody collected are dropped; the streams are idle here)
    d->pending = 0;
    if (n > d->rows_cap) {
        if (d->rows_in) FM_HIP(hipFree(d->rows_in));
        d->rows_in = nullptr;
        FM_HIP(hipMalloc(&d->rows_in, sizeof(float) * 7 * (size_t)n));
        d->rows_cap = n;
    }
    if (n) FM_HIP(hipMemcpyAsync(d->rows_in, rows, sizeof(float) * 7 * (size_t)n, hipMemcpyHostToDevice, s));
    FM_HIP(hipMemsetAsync(d->counters[d->wr], 0, sizeof(int32_t) * 4, s));
    if (n) hipLaunchKernelGGL(rows_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d->rows_in, n, filter_args(d, d->wr));
    int rc = enqueue_post(ctx, d, s);
    if (rc) return rc;
    return collect(ctx, d, s, out, cap, n_out);
}

// candidates that passed the confidence threshold / detections that survived NMS and the box filters in the pass that
// fm_detect_sync collected last (what the sort + NMS kernels of that pass worked on)
extern "C" int fm_detect_last_counts(fm_ctx* ctx, int* n_candidates, int* n_detections) {
    FM_CHECK_ARG(ctx && ctx->det && n_candidates && n_detections);
    const DetState* d = ctx->det;
    if (d->last < 0) {
        *n_candidates = *n_detections = 0;
        return 0;
    }
    *n_candidates = d->counters_host[d->last][0];
    *n_detections = d->counters_host[d->last][2];
    return 0;
}

extern "C" int fm_detect_raw_candidates(fm_ctx* ctx, float* rows, int cap, int* n) {
    FM_CHECK_ARG(ctx && ctx->det && rows && n);
    DetState* d = ctx->det;
    int rc_f = flush_post(ctx, d);
    if (rc_f) return rc_f;
    FM_HIP(hipStreamSynchronize(ctx->s_det));
    FM_HIP(hipStreamSynchronize(ctx->s_up));
    const int slot = (d->wr + DetState::NSLOT - 1) % DetState::NSLOT;       // the pass enqueued last
    int32_t cnt[4];
    FM_HIP(hipMemcpy(cnt, d->counters[slot], sizeof(cnt), hipMemcpyDeviceToHost));
    const int k = cnt[0] < d->cap ? cnt[0] : d->cap;
    FM_CHECK_ARG(k <= cap);
    FM_HIP(hipMemcpy(rows, d->sorted[slot], sizeof(float) * 8 * k, hipMemcpyDeviceToHost));
    *n = k;
    return 0;
}

// HIP-event time (ms) of the network launches of the last fm_detect_async, measured on the
// detector stream itself (bench.py roofline: conv FLOPs / this time).
extern "C" int fm_detect_net_ms(fm_ctx* ctx, float* ms) {
    FM_CHECK_ARG(ctx && ctx->det && ms && ctx->det->last >= 0);      // the pass collected last
    const int slot = ctx->det->last;
    if (!ctx->det->timed[slot]) { *ms = -1.f; return 0; }       // (this pass was not timed: option "net_timing")
    FM_HIP(hipEventSynchronize(ctx->det->ev1[slot]));
    FM_HIP(hipEventElapsedTime(ms, ctx->det->ev0[slot], ctx->det->ev1[slot]));
    return 0;
}

/*
 * fastmot_hip.h -- C ABI of libfastmot_hip.so, the MI355X (gfx950) implementation of the
 * FastMOT per-frame hot path (MOT.step and everything under it).
 *
 * The reference (GeekAlexis/FastMOT) is Python; its only native boundary is a TensorRT
 * plugin loaded with ctypes (fastmot/utils/inference.py:49-53).  TensorRT does not exist
 * on ROCm, so this header DEFINES the boundary: every entry point below replaces the
 * numeric body of one reference routine (cited as file:line, relative to /root/reference)
 * and is bound from Python with ctypes (fastmot_amd/_lib.py; INTEGRATION.md shows the stub
 * a reference maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; fm_last_error() gives the message.
 *   - the caller owns every host pointer it passes; the library owns all device memory.
 *   - a ctx is bound to one GPU and is NOT thread safe (one video stream = one ctx = one
 *     process, mirroring the reference's single-threaded MOT.step, mot.py:125-168).
 *   - "slot" = index of a track in the device-resident track table (state mean f64[8],
 *     covariance f64[8][8], running-mean ReID feature f32[512]); slot bookkeeping (which
 *     slot belongs to which track id) stays in Python, like the reference's dict of Tracks.
 *   - boxes are tlbr f64[4] with inclusive corners (utils/rect.py:17-57).
 */
#ifndef FASTMOT_HIP_H
#define FASTMOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fm_ctx fm_ctx;

/* ---------------------------------------------------------------- runtime ------------- */
/* replaces TRTInference.__init__ stream/buffer setup (utils/inference.py:44-94) */
int fm_ctx_create(int device, fm_ctx** out);
int fm_ctx_destroy(fm_ctx* ctx);
const char* fm_last_error(void);
/* number of visible HIP devices (<0 on error) */
int fm_device_count(void);
/* PCI address ("0000:c1:00.0") of HIP device `device`: the host side uses it to find the GPU's NUMA node
 * (/sys/bus/pci/devices/<id>/numa_node) and binds the stream's process to it before any page-locked memory is
 * allocated (fastmot_amd/runtime.py) -- the per-frame H2D copy of 6.2 MB reads that memory. */
int fm_device_pci_bus_id(int device, char* buf, int buflen);
/* blocks until every stream of the ctx is idle (TRTInference.synchronize, inference.py:119-121) */
int fm_ctx_synchronize(fm_ctx* ctx);
/* makes the context's device current for the calling host thread; every additional host thread that
 * drives the context calls it once (HIP's current device is per thread) */
int fm_ctx_bind_thread(fm_ctx* ctx);
/* tunables: "zero_copy_tracks" (default 2048; batches up to this many tracks / boxes exchange kernel
 * inputs and outputs through pinned device-mapped host memory instead of blit copies; 0 disables),
 * "host_lap_elems" (default 262144; LAP cost matrices up to this many elements are solved by the host
 * solver of the library, larger ones by the device kernels; 0 = always device), "nms_path" (default 0: the fused sort + greedy
 * DIoU-NMS kernel for up to 4096 candidates per frame, the three-kernel sort / bit-matrix / scan path beyond; 1 = always
 * the latter), "convd_cfg" (default 0 = chosen per layer; bm | bn << 8 | kg << 16 | ns << 20 | role << 24 | (spb == 2) << 25 forces
 * one tile / K-group / ring depth (0 = as deep as LDS allows, otherwise >= 2) / loader-wave / steps-per-barrier configuration
 * of FM_OP_CONVD for A/B measurements; read at every launch that is not a graph replay),
 * "fused_input" (default 1: a network that begins with a stem convolution lets that convolution compute its input pixels from the
 * frame -- detector resize / ReID crops -- instead of running the front-end kernel into the input tensor; 0 for A/B and tests),
 * "net_timing" (default 0; N > 0: every N-th detector pass carries the HIP-event pair fm_detect_net_ms reads),
 * "use_graphs" (default 1;
 * 0 launches the network layers one by one instead of replaying hipGraphs), "lk_variant" (diagnostic builds only, include/fastmot_hip_diag.h;
 * 0 is the only value the shipped library accepts).  Initial values can be set with the environment
 * variables FASTMOT_ZERO_COPY / FASTMOT_HOST_LAP / FASTMOT_GRAPHS. */
int fm_ctx_set_option(fm_ctx* ctx, const char* key, int value);
/* Event trace of a pipelined run (diagnostics; scripts/trace_pipeline.py).  fm_trace_start arms `cap` timed events and
 * returns the host's CLOCK_MONOTONIC time (ns) that corresponds to the trace's zero; from then on the library records
 * one event per stage boundary on the stage's own stream (tags: 10-13 detector pass reached / inputs ready / network done
 * / decode done, 20-21 post-processing, 30-31 next frame's H2D copy, 32-33 ReID network, 40-41 LK launch).
 * fm_trace_read synchronises the device, returns tags[i] / ms[i] (GPU time since the trace's zero) and disarms. */
int fm_trace_start(fm_ctx* ctx, int cap, int64_t* host_ns);
int fm_trace_read(fm_ctx* ctx, int cap, int32_t* tags, float* ms, int* n);
/* writes "name:gcnArch:CUs:clockMHz:hbmBytes" of the ctx device */
int fm_device_info(fm_ctx* ctx, char* buf, int buflen);

/* ---------------------------------------------------------------- Kalman filter ------- */
/* KalmanFilter.__init__/reset_dt tunables (kalman_filter.py:13-94, _init_mat :294-306) */
typedef struct fm_kf_params {
    double dt;
    double std_factor_acc, std_offset_acc;
    double std_factor_det[2], std_factor_klt[2];
    double min_std_det[2], min_std_klt[2];
    double init_pos_weight, init_vel_weight;
    double vel_coupling, vel_half_life;
} fm_kf_params;

int fm_kf_configure(fm_ctx* ctx, const fm_kf_params* p);
/* frame rectangle used for the "ios(box, frame) < 0.5 => lost" test (tracker.py:180,263) */
int fm_set_frame_rect(fm_ctx* ctx, const double tlbr[4]);

/* KalmanFilter.create for n new tracks (kalman_filter.py:96-126): state written to slots */
int fm_trk_create(fm_ctx* ctx, int n, const int32_t* slots, const double* det_tlbr);

/* MultiTracker.apply_kalman body for n tracks in ONE launch (tracker.py:164-183):
 *   warp(H) -> predict -> [update(klt box, FLOW, mult)] -> as_tlbr -> ios(frame) test.
 * H: 3x3 row-major homography.  klt_tlbr[n][4], has_klt[n], mult[n] (std multiplier,
 * tracker.py:175).  Outputs: tlbr_out[n][4] (rounded half-even), lost_out[n] (ios<0.5). */
int fm_trk_step(fm_ctx* ctx, int n, const int32_t* slots, const double* H,
                const double* klt_tlbr, const uint8_t* has_klt, const double* mult,
                double* tlbr_out, uint8_t* lost_out);

/* Same kernel with an explicit stage mask, so that KalmanFilter.warp / predict / update(FLOW)
 * (kalman_filter.py:128-204,227-292) remain individually callable through the mirror class. */
enum { FM_KF_WARP = 1, FM_KF_PREDICT = 2, FM_KF_UPDATE_KLT = 4 };
int fm_trk_step_ops(fm_ctx* ctx, int ops, int n, const int32_t* slots, const double* H,
                    const double* klt_tlbr, const uint8_t* has_klt, const double* mult,
                    double* tlbr_out, uint8_t* lost_out);

/* KalmanFilter.update(..., MeasType.DETECTOR) for n matched (slot, detection box) pairs
 * (tracker.py:259-262).  Same outputs as fm_trk_step. */
int fm_trk_update_det(fm_ctx* ctx, int n, const int32_t* slots, const double* det_tlbr,
                      double* tlbr_out, uint8_t* lost_out);

/* Track.state accessors (track.py:134) -- off the hot path (tests, visualisation). */
int fm_trk_get_state(fm_ctx* ctx, int n, const int32_t* slots, double* mean, double* cov);
int fm_trk_set_state(fm_ctx* ctx, int n, const int32_t* slots, const double* mean, const double* cov);
/* state[dst] = state[src]  (Track.merge_continuation, track.py:204-208) */
int fm_trk_copy_state(fm_ctx* ctx, int dst_slot, int src_slot);

/* ---------------------------------------------------------------- ReID features ------- */
/* embedding dimension of the running-mean feature table (default 512) */
int fm_feat_configure(fm_ctx* ctx, int dim);
/* upload this frame's L2-normalised embeddings [n][dim] f32 (FeatureExtractor.postprocess
 * output, feature_extractor.py:62-74) so that association can use them on the device.
 * If emb == NULL the embeddings already produced on the device by fm_extract_* are used. */
int fm_emb_upload(fm_ctx* ctx, int n, const float* emb);
/* AverageFeature.update for n (slot, embedding row) pairs (track.py:106-112,119-126) */
int fm_feat_update(fm_ctx* ctx, int n, const int32_t* slots, const int32_t* emb_rows);
/* AverageFeature.merge: dst absorbs src (track.py:114-117) */
int fm_feat_merge(fm_ctx* ctx, int dst_slot, int src_slot);
/* clears the feature state of n slots (new Track, track.py:142-143) */
int fm_feat_reset(fm_ctx* ctx, int n, const int32_t* slots);
int fm_feat_get(fm_ctx* ctx, int slot, float* sum, float* avg, int32_t* count);
/* batched read / seed of the running-mean features (cross-stream gallery exchange; not in the
 * reference, opt-in, see fastmot_amd/gallery.py) */
int fm_feat_read(fm_ctx* ctx, int n, const int32_t* slots, float* avg_out, int32_t* count_out);
int fm_feat_write(fm_ctx* ctx, int n, const int32_t* slots, const float* avg, const int32_t* count);

/* ---------------------------------------------------------------- association --------- */
enum { FM_METRIC_EUCLIDEAN = 0, FM_METRIC_COSINE = 1 }; /* utils/distance.py:12-14 */

/* find_occluded (utils/rect.py:143-157) */
int fm_find_occluded(fm_ctx* ctx, int n, const double* tlbr, double thresh, uint8_t* out);

/* All pairwise terms of MultiTracker.update in one launch, kept on the device:
 *   feat[t][d]  cdist(avg_feat[slot t], emb[d])            (utils/distance.py:17-87)
 *   maha[t][d]  KalmanFilter.motion_distance               (kalman_filter.py:206-225,347-353)
 *   iou [t][d]  iou_dist(track box, det box)               (utils/distance.py:91-108)
 * rows = nT tracks (slots[], rounded boxes trk_tlbr[], labels), cols = nD detections
 * (boxes, labels, occluded mask).  Rows without a valid feature use the fill value at
 * stage time (tracker.py:328-330).  trk_feat_f32[t]=1 marks rows whose feature enters cdist as
 * float32 (history tracks in _reid_cost, tracker.py:360-362) instead of the float64 copy used by
 * _matching_cost (tracker.py:321-326): the products are then formed in f32 like the reference. */
int fm_assoc_prepare(fm_ctx* ctx, int metric,
                     int nT, const int32_t* slots, const double* trk_tlbr, const int64_t* trk_label,
                     int nD, const double* det_tlbr, const int64_t* det_label,
                     const uint8_t* det_occluded, const uint8_t* trk_feat_f32);

/* downloads the prepared [nT][nD] f64 matrices (any pointer may be NULL); KalmanFilter.
 * motion_distance / cdist / iou_dist parity tests read them. */
int fm_assoc_get_pairwise(fm_ctx* ctx, double* feat, double* maha, double* iou);

enum {
    FM_STAGE_MATCHING = 0, /* _matching_cost: 0.8 feat + 0.2 maha/chi2, gates (tracker.py:314-341) */
    FM_STAGE_IOU      = 1, /* _iou_cost: gate 1-iou_thresh                     (tracker.py:343-353) */
    FM_STAGE_REID     = 2  /* _reid_cost: feat, label gate only                (tracker.py:355-366) */
};

/* Builds the gated cost matrix of one association stage from the prepared pairwise terms
 * for the given row / column subsets (indices into the fm_assoc_prepare arrays), then
 * solves it on the device:
 *   solver 0 = rectangular LAP (scipy.optimize.linear_sum_assignment semantics incl.
 *              tie-breaking; utils/matching.py:10-30) followed by the INF_COST un-matching
 *              of utils/matching.py:58-70: match_gated[k]=1 when cost>=1e5.
 *   solver 1 = greedy argmin matching while cost <= max_cost (utils/matching.py:74-97).
 * row_label_override: _reid_cost quirk (tracker.py:364) -- labels used for the rows; NULL
 * = labels given to fm_assoc_prepare.
 * Outputs: m_rows/m_cols (local indices into rows[]/cols[], in the solver's output order),
 * n_match.  cost_out (optional, may be NULL): the [nr][nc] f64 cost matrix. */
int fm_assoc_stage(fm_ctx* ctx, int stage, int solver,
                   int nr, const int32_t* rows, int nc, const int32_t* cols,
                   double motion_weight, double max_cost, double fill_val,
                   const int64_t* row_label_override,
                   int32_t* m_rows, int32_t* m_cols, uint8_t* match_gated, int* n_match,
                   double* cost_out);

/* fm_assoc_prepare with two additions.  after_extractor = 1: the embeddings are the batch fm_extract_async is still
 * computing -- the pairwise kernel is ordered behind the ReID network's last launch on the device, so
 * MultiTracker.update_begin can enqueue it while that network runs and the terms are complete a kernel's length after
 * the embeddings (no host wake-up in between).  *host_cascade = 1 when the problem is small enough (nT * nD <=
 * host_lap_elems, zero_copy_tracks > 0) for the terms to be written to page-locked host memory as well, where
 * fm_assoc_cascade reads them. */
int fm_assoc_prepare2(fm_ctx* ctx, int metric,
                      int nT, const int32_t* slots, const double* trk_tlbr, const int64_t* trk_label,
                      int nD, const double* det_tlbr, const int64_t* det_label,
                      const uint8_t* det_occluded, const uint8_t* trk_feat_f32, int after_extractor, int* host_cascade);

/* The association cascade of MultiTracker.update (tracker.py:198-248) in one call, for problems whose assignment is
 * solved on the host anyway (host_lap_elems; after fm_assoc_prepare2 reported host_cascade): the three
 * linear-assignment stages -- _matching_cost depth by depth (tracker.py:205-218,314-341), _iou_cost for the remaining
 * active and then for the unconfirmed tracks (:220-231,343-353) -- with the unmatched lists in the order of
 * utils/matching.py:58-70 under Numba's set iteration, the confidence / occlusion split of the remaining detections
 * (:233-235) and the greedy re-identification against the history (:236-240,355-366, utils/matching.py:74-97).
 * Rows are indices into the fm_assoc_prepare arrays.  `out` (int32, capacity out_cap >= FM_CASCADE_HEADER +
 * 3 * (n_conf + n_unconf) + 3 * nD): header [n1, n2, n3, nu1, nu2, nu3, n_reid, n_invalid, n_rest, used], then
 * (row, det) pairs of the three stages, the unmatched rows of the three stages (stage 1: its inactive tracks, :221),
 * (index into hist_rows, det) pairs of the re-identification, the occluded unmatched detections, the others. */
#define FM_CASCADE_HEADER 16
typedef struct fm_cascade_in {
    int32_t n_groups;              /* depth groups of the confirmed tracks (tracker.py:219-233: age // 2) */
    int32_t n_unconf, n_hist;
    int32_t reserved;
    const int32_t* group_off;      /* [n_groups + 1] offsets into conf_rows */
    const int32_t* conf_rows;      /* rows of the confirmed tracks, group by group */
    const uint8_t* conf_active;    /* Track.active of each of them */
    const int32_t* unconf_rows;
    const int32_t* hist_rows;      /* history rows (then foreign gallery rows) */
    const int64_t* hist_labels;    /* labels used for them (tracker.py:364 takes the first n of ALL history tracks) */
    const double* det_conf;        /* [nD] */
    double motion_weight, max_assoc_cost, fill_val, max_iou_cost, conf_thresh, max_reid_cost;
} fm_cascade_in;
int fm_assoc_cascade(fm_ctx* ctx, const fm_cascade_in* in, int32_t* out, int out_cap);
/* The host half of fm_assoc_cascade on caller-provided pairwise terms ([nT][nD] f64 each; row_has_feat, labels and the
 * occlusion mask as fm_assoc_prepare takes them): no context, no device -- what the CPU test suite checks against the
 * oracle's restatement of utils/matching.py (tests/test_cascade_host.py).  Not a fallback: the terms themselves only
 * ever come from pairwise_kernel. */
int fm_cascade_host(int nT, int nD, const double* feat, const double* maha, const double* iou,
                    const uint8_t* row_has_feat, const int64_t* trk_label, const int64_t* det_label,
                    const uint8_t* det_occluded, const fm_cascade_in* in, int32_t* out, int out_cap);

/* Stand-alone solvers on a host cost matrix [nr][nc] f64 (device kernels; used by
 * _rectify_matches' greedy_match, tracker.py:384, and by the parity tests). */
int fm_lap(fm_ctx* ctx, const double* cost, int nr, int nc,
           int32_t* m_rows, int32_t* m_cols, int* n_match);
int fm_greedy(fm_ctx* ctx, const double* cost, int nr, int nc, double max_cost,
              int32_t* m_rows, int32_t* m_cols, int* n_match);
/* iou_dist on host boxes (utils/distance.py:91-108) -- used by _rectify_matches */
int fm_iou_dist(fm_ctx* ctx, int na, const double* a, int nb, const double* b, double* out);

/* ---------------------------------------------------------------- conv engine --------- */
/* Replaces the TensorRT engines (fastmot/utils/inference.py:39-125; engine build
 * fastmot/models/yolo.py:106-151, fastmot/models/reid.py:48-92).  A network is a layer table over
 * NHWC fp16 tensors (channels padded to 8); Conv+BN+activation(+shortcut) layers run on the MFMA
 * implicit-GEMM kernel, concat/route is expressed by channel offsets into shared tensors. */
enum { FM_NET_DETECTOR = 0, FM_NET_EXTRACTOR = 1,
       FM_NET_EXTRACTOR_B = 2 /* 2, 3, 4: optional further instances of the ReID network (same layer table, own
                               * buffers and streams): fm_extract_async then runs a batch as 2-4 concurrent parts */ };
#define FM_MAX_EXTRA_EXTRACTORS 3
enum {
    FM_OP_CONV = 0,      /* conv k x k, stride, pad + bias + act (+ residual)                    */
    FM_OP_DWCONV3 = 1,   /* depthwise 3x3 s1 p1 + bias + act (OSNet LightConv3x3)                */
    FM_OP_MAXPOOL = 2,   /* k, stride, pad (yolo2onnx.py:838-863; OSNet k3 s2 p1)               */
    FM_OP_AVGPOOL = 3,   /* k = stride (OSNet transition 2x2)                                   */
    FM_OP_UPSAMPLE2 = 4, /* nearest x2 (yolo2onnx.py:806-836)                                   */
    FM_OP_COPY = 5,      /* channel-slice copy                                                  */
    FM_OP_GATE = 6,      /* OSNet channel gate: GAP -> fc1 -> ReLU -> fc2 -> sigmoid -> gate[0] */
    FM_OP_GATE_SUM = 7,  /* out = sum_i in[i] * gate[i]                                         */
    FM_OP_HEAD = 8,      /* GAP -> Linear(+BN1d) -> ReLU -> L2 normalise -> ctx embeddings      */
    FM_OP_SPP = 10,      /* darknet SPP block: stride-1 max pools k = 13, 9, 5 of in[0] (cin channels)
                          * written to out at channel offsets out_coff + {0, cin, 2 cin}; one launch,
                          * pool9 = pool5 o pool5, pool13 = pool5 o pool9, separable, in LDS      */
    FM_OP_LITECONV = 9,  /* fused OSNet LightConv3x3: 1x1 linear (w_off, no bias) -> depthwise 3x3
                          * (w2_off) + bias (b_off) + act; cin == cout <= 128.  n_in = G <= 4
                          * independent LightConvs of equal geometry in one launch: group g reads
                          * in[g]/in_coff[g], writes out channels [out_coff + g*cin, +cin) and uses
                          * the g-th slab of the stacked weights                                */
    FM_OP_STEMCONV = 12, /* k x k conv (k,stride in {3/1, 3/2, 7/2}) of an input with <= 4 real channels,
                          * cout <= 32, + bias + act: w = fp16 [32][ceil16(k*k*4)] (K order kh,kw,c<4),
                          * b = f32[32]; patch staged in LDS (stemconv.hip)                     */
    FM_OP_ADD = 13,      /* out = in[0] + in[1] (stand-alone [shortcut])                          */
    FM_OP_RESBLOCK = 14, /* fused darknet residual unit (1x1 conv, 3x3 conv, [shortcut] from=-3;
                          * yolo2onnx.py:558-760): out = in[0] + act(conv3x3(act(conv1x1(in[0]))));
                          * cin = cout in {64,128,256} channels, hid = mid channels (% 32, <= 256);
                          * w_off/b_off = the 1x1 conv, w2_off/b2_off = the 3x3 conv; weights in MFMA
                          * fragment order [cout/32][K/16][lane = (k/8%2)*32 + cout%32][k%8], K order
                          * (kh, kw, cin) (resblock.hip)                                          */
    FM_OP_CONVS = 15,    /* FM_OP_CONV for layers with few output pixels and a long reduction (19 x 19 YOLO
                          * levels): K split across the waves of a workgroup, no workspace / reduce launch
                          * (convs.hip).  Same fields and semantics as FM_OP_CONV; cin % 64 == 0; weights in
                          * MFMA fragment order [ceil32(cout)/32][k*k*cin/16][lane][8] as FM_OP_RESBLOCK   */
    FM_OP_LITECHAIN = 16,/* the four streams of an OSNet block (chains of 1..4 FM_OP_LITECONV over in[0]) in one
                          * launch (litechain.hip): stream s writes out channels [out_coff + s*cin, +cin);
                          * w_off / w2_off / b_off = the 10 parameter sets stacked in (stream, level) order,
                          * each laid out as for FM_OP_LITECONV; gate[s] = GAP partial slot of stream s       */
    FM_OP_CONVD = 17,    /* FM_OP_CONV with both operands moved global -> LDS by the DMA path and up to 2 x 2 MFMA accumulators
                          * per wavefront, K optionally split across wave groups of the workgroup (convd.hip).  Same fields
                          * and semantics as FM_OP_CONV; 3x3: cin % 64 == 0; 1x1: cin % 8 == 0 and cin >= 16 (ragged K: the
                          * image is zero-padded to ceil64(K), chunks of a step beyond cin are not fetched); weights as LDS
                          * tile images: [ceil32(cout)/32][ceil64(k*k*cin)/64][32 rows][8 slots][8 halfs], K order (kh, kw, cin),
                          * slot s of row r holding K chunk s ^ ((r / 2) % 8) of that row's 64-wide K step              */
    FM_OP_STEM2 = 18,    /* the first two layers of a Darknet YOLO backbone in one launch (stem2.hip): conv 3x3 s1 (<= 4 real input
                          * channels -> hid = 32, activation gate[0]) then conv 3x3 s2 pad 1 (32 -> cout in {64, 128}, activation act);
                          * in[0] = the network input; w_off / b_off = the first conv as for FM_OP_STEMCONV, w2_off / b2_off = the second
                          * in MFMA fragment order [cout/32][288/16][lane][8] (K order kh, kw, cin) + f32 bias                          */
    FM_OP_PAIR11 = 19,   /* two 1x1 convs around a concat in one launch (pair11.hip): t = act_gate[0](W1 in[0] + b1) (64 -> hid = 64), out =
                          * act(W2 [t | in[1]] + b2) (64 + 64 -> cout in {64, 128}); weights in MFMA fragment order (w_off / b_off, w2_off / b2_off) */
    FM_OP_OSTAIL = 20,   /* OSNet x0.25's last stage and head in one launch, one workgroup per sample (ostail.hip): AvgPool2d(2, 2) of
                          * in[0] (32 x 16 x cin = 96) -> OSBlock (96 -> k = 128, hid = 32 mid channels, with downsample) -> OSBlock
                          * (128 -> 128) -> conv5 -> GAP -> fc (cout = 512, BN folded) -> ReLU -> L2 normalise -> ctx embeddings (as
                          * FM_OP_HEAD).  w_off: the fp16 parameters (`stride` halfs), b_off: the fp32 ones (`pad` floats) in the
                          * order ostail.hip documents (pointwise weights in MFMA fragment order)                                      */
    FM_OP_GATED_SUM = 11 /* OSNet unified aggregation gate in one launch: out = sum_i in[i] *
                          * sigmoid(fc2(relu(fc1(GAP(in[i]))))) with shared fc weights
                          * (w_off, b_off, w2_off, b2_off, hid) -- FM_OP_GATE x n_in + FM_OP_GATE_SUM */
};
enum { FM_ACT_LINEAR = 0, FM_ACT_LEAKY = 1, FM_ACT_MISH = 2, FM_ACT_RELU = 3, FM_ACT_LOGISTIC = 4,
       FM_ACT_SWISH = 5 };
enum { FM_RES_NONE = 0, FM_RES_AFTER_ACT = 1, FM_RES_BEFORE_ACT = 2 };

typedef struct fm_tensor {
    int32_t h, w, c;     /* per-sample geometry, c = channel stride (multiple of 8) */
    int32_t f32;         /* 1: fp32 storage (YOLO head outputs), 0: fp16 */
    int64_t offset;      /* byte offset in the network's activation arena (256 B aligned), or -1 for a
                          * private buffer.  Tensors whose live ranges do not overlap may share bytes:
                          * keeping the working set small keeps it resident in the 256 MB Infinity Cache */
} fm_tensor;

typedef struct fm_layer {
    int32_t op;
    int32_t n_in;
    int32_t in[4], in_coff[4];
    int32_t out, out_coff;
    int32_t res, res_coff, res_mode;
    int32_t cin, cout, k, stride, pad, act;
    int32_t hid;
    int32_t up;          /* CONV: nearest-neighbour upsampling factor of the stored output (1 or 2):
                          * the [upsample] layer of yolo2onnx.py:806-836 folded into its producer */
    int32_t gate[4];
    int64_t w_off, b_off, w2_off, b2_off;   /* byte offsets into the weight blob (16 B aligned) */
} fm_layer;

/* weights: CONV  w = fp16 [ceil32(cout)][ceil64(k*k*cin)] (K order kh,kw,cin), b = f32[ceil32(cout)]
 *          DWCONV3 w = fp16 [9][c], b = f32[c];  GATE w=[hid][c] b=[hid] w2=[c][hid] b2=[c];
 *          HEAD  w = fp16 [cout][cin], b = f32[cout]  (BN folded everywhere). */
int fm_net_create(fm_ctx* ctx, int which, int max_batch, int n_tensors, const fm_tensor* tensors,
                  int n_layers, const fm_layer* layers, const void* weights, size_t weight_bytes,
                  int n_gates, int gate_channels, size_t arena_bytes);
int fm_net_destroy(fm_ctx* ctx, int which);
/* enqueues every layer for `batch` samples on the network's stream (no host sync) */
int fm_net_run(fm_ctx* ctx, int which, int batch);
/* host <-> tensor copies (synchronous; tests and weight-free smoke runs) */
int fm_net_tensor_write(fm_ctx* ctx, int which, int tensor, const void* host, size_t bytes);
int fm_net_tensor_read(fm_ctx* ctx, int which, int tensor, void* host, size_t bytes);
/* copies the embeddings produced by FM_OP_HEAD ([n][feat_dim] f32, L2-normalised) to the host
 * after synchronising the extractor stream (FeatureExtractor.postprocess, feature_extractor.py:62-74) */
int fm_net_read_embeddings(fm_ctx* ctx, int n, float* host);
/* total FLOPs (2*MAC) and minimal HBM bytes of one run at the given batch: the numbers the
 * bench's roofline uses (SURVEY.md section 8d formulas) */
int fm_net_cost(fm_ctx* ctx, int which, int batch, double* flops, double* bytes);
/* average duration in ms of the conv (MFMA) launches of the last fm_net_run measured with HIP
 * events on the network's own stream; enable != 0 switches per-layer timing on (slow path). */
int fm_net_profile(fm_ctx* ctx, int which, int batch, int iters, double* conv_ms, double* other_ms,
                   int* n_conv, int* n_other);

/* per-layer HIP-event times in ms (out[n_layers]); tuning aid */
int fm_net_profile_layers(fm_ctx* ctx, int which, int batch, int iters, double* out);

/* ---------------------------------------------------------------- frames -------------- */
/* Frames are BGR u8 HxWx3, C-contiguous (what VideoIO.read() hands to MOT.step, app.py:85).
 * fm_frame_upload copies a host frame to the ctx's current device frame (pinned staging + async
 * H2D on the detector stream; replaces cp.asarray(frame), detector.py:292).  A ring of frames can
 * also be made resident in HBM up front (bench: inputs resident before the timed region). */
int fm_frame_configure(fm_ctx* ctx, int width, int height, int ring_size);
int fm_frame_upload(fm_ctx* ctx, const uint8_t* bgr);
/* Page-locked host buffers for frames (the reference preallocates pinned HostDeviceMem buffers,
 * utils/inference.py:7-36, flow.py:100-118).  A frame that lies inside a buffer obtained from
 * fm_host_alloc is copied to the device straight from where it is (no staging copy); it must stay
 * unmodified until the step that uses it has returned (next-frame prefetch: until it has been the
 * current frame).  Any other host pointer is staged through the ctx's own pinned buffer first. */
int fm_host_alloc(size_t bytes, void** out);
int fm_host_free(void* p);
/* Next-frame prefetch (no counterpart in the reference, whose detector is synchronous per step): the
 * detector network can be started on frame t+1 while frame t is still in the ReID / association stages.
 * fm_frame_upload_next copies a host frame into the second upload slot (asynchronously; the detector pass on it waits
 * for the copy's event),
 * fm_frame_ring_select_next points at a resident frame; fm_detect_async_next = fm_detect_async on that
 * frame; fm_frame_promote_next makes it the current frame of the next step without another upload. */
int fm_frame_upload_next(fm_ctx* ctx, const uint8_t* bgr);
int fm_frame_ring_select_next(fm_ctx* ctx, int index);
int fm_frame_promote_next(fm_ctx* ctx);
int fm_detect_async_next(fm_ctx* ctx);
int fm_frame_ring_store(fm_ctx* ctx, int index, const uint8_t* bgr);
int fm_frame_ring_select(fm_ctx* ctx, int index);
int fm_frame_read(fm_ctx* ctx, uint8_t* bgr);   /* current device frame -> host (tests) */

/* ---------------------------------------------------------------- detector ------------ */
#define FM_MAX_HEADS 4
#define FM_MAX_ANCHORS 6   /* yolo_layer.h:11 */
typedef struct fm_det48 {  /* DET_DTYPE, detector.py:18-23 (aligned, 48 B) */
    double tlbr[4];
    int64_t label;
    double conf;
} fm_det48;

typedef struct fm_yolo_cfg {
    int32_t in_w, in_h;                 /* network input (INPUT_SHAPE) */
    int32_t roi_x, roi_y, roi_w, roi_h; /* letterbox ROI inside the input (whole input if !LETTERBOX) */
    int32_t input_tensor;               /* tensor id of the network input */
    int32_t n_heads;
    int32_t head_tensor[FM_MAX_HEADS];  /* fp32 NHWC head tensors, channel = anchor*(5+C) + attr */
    int32_t grid_w[FM_MAX_HEADS], grid_h[FM_MAX_HEADS], n_anchors[FM_MAX_HEADS];
    float anchors[FM_MAX_HEADS][2 * FM_MAX_ANCHORS];
    float scale_xy[FM_MAX_HEADS];
    int32_t num_classes, new_coords;
    uint8_t label_mask[128];            /* class ids to keep (detector.py:262-266) */
    double conf_thresh, nms_thresh, max_area, min_aspect_ratio;
    double size[2], offset[2];          /* upscaled_sz / bbox_offset (detector.py:302-320) */
    int32_t max_candidates;             /* capacity of the candidate list (default 8192) */
} fm_yolo_cfg;

int fm_detect_configure(fm_ctx* ctx, const fm_yolo_cfg* cfg);
/* YOLODetector.detect_async (detector.py:270-273): preprocess (bilinear resize in u8, BGR->RGB,
 * /255, detector.py:289-300) -> network -> head decode (plugins/yolo_layer.cu:127-230) ->
 * score/class filter -> per-class DIoU-NMS -> box filter (detector.py:322-365); network and decode on the detector
 * stream, sort / NMS / filter behind them on a stream of their own; only the surviving detections reach the host
 * (written to page-locked memory by the last kernel). */
int fm_detect_async(fm_ctx* ctx);
/* YOLODetector.postprocess (detector.py:275-287): waits for the stream, returns detections sorted
 * by class id.  Returns FM_ERR_STATE if the candidate list overflowed. */
int fm_detect_sync(fm_ctx* ctx, fm_det48* out, int cap, int* n);
/* candidates over conf_thresh / detections after NMS + box filters of the pass fm_detect_sync collected last */
int fm_detect_last_counts(fm_ctx* ctx, int* n_candidates, int* n_detections);
/* test hooks: run only the preprocessing / only the filter+NMS stage on host-provided YOLO
 * candidate rows [n][7] = x, y, w, h, box_conf, class_id, class_prob (yolo_layer.h:34-39) */
int fm_detect_preprocess_only(fm_ctx* ctx);
int fm_filter_dets(fm_ctx* ctx, const float* rows, int n, fm_det48* out, int cap, int* n_out);
int fm_detect_raw_candidates(fm_ctx* ctx, float* rows, int cap, int* n);
/* HIP-event duration (ms) of the network launches of the pass fm_detect_sync collected last, recorded on the detector
 * stream (the bench's live roofline measurement); -1 when that pass carried no events (option "net_timing") */
int fm_detect_net_ms(fm_ctx* ctx, float* ms);

/* ---------------------------------------------------------------- feature extractor --- */
/* FeatureExtractor.extract_async (feature_extractor.py:48-60): for n boxes crop the current
 * device frame (multi_crop, utils/rect.py:93-97: int truncation, clamp >= 0, inclusive br),
 * resize to INPUT_SHAPE with OpenCV's fixed-point INTER_LINEAR (cv2.resize, feature_extractor.py:85),
 * BGR->RGB, /255, ImageNet mean/std (feature_extractor.py:88-98) -> network input, then run the
 * ReID network in batches of max_batch; FM_OP_HEAD leaves the L2-normalised embeddings
 * (feature_extractor.py:73) in the ctx embedding table, where association reads them. */
int fm_extract_configure(fm_ctx* ctx, int input_tensor, int in_w, int in_h);
int fm_extract_async(fm_ctx* ctx, int n, const double* tlbr);
/* FeatureExtractor.postprocess (feature_extractor.py:62-74): sync + copy [n][dim] f32 to host */
int fm_extract_sync(fm_ctx* ctx, int n, float* emb);
/* test hook: the preprocessed crops [n][in_h][in_w][3] as f32 (RGB, normalised) */
int fm_extract_read_input(fm_ctx* ctx, int n, float* out);

/* ---------------------------------------------------------------- optical flow (KLT) --- */
/* Flow.__init__ buffers / parameters (flow.py:17-119).  Image sizes: full frame, optical-flow
 * frame (opt_flow_scale_factor) and background-feature frame (bg_feat_scale_factor). */
typedef struct fm_flow_cfg {
    int32_t small_w, small_h;     /* round(opt_flow_scale_factor * size) */
    int32_t bg_w, bg_h;           /* round(bg_feat_scale_factor * size)  */
    int32_t win_size, max_level, max_count;   /* cv2.calcOpticalFlowPyrLK winSize / maxLevel / criteria */
    double epsilon;
    int32_t fast_thresh;          /* bg_feat_thresh */
    int32_t max_corners, block_size;
    double quality_level;         /* obj_feat_params */
    int32_t gray_coeff_bits;      /* cv2.cvtColor(BGR2GRAY) fixed point: 14 = B 1868, G 9617, R 4899 (OpenCV <= 4.2-era
                                   * RGB2Gray<uchar>, what the reference's pinned 4.1.1 computes as far as it can be
                                   * established offline), 15 = 3735 / 19235 / 9798 (later 4.x); 0 = 14.  DESIGN.md 7 */
} fm_flow_cfg;
int fm_flow_configure(fm_ctx* ctx, const fm_flow_cfg* cfg);
/* Flow.init (flow.py:121-133): BGR->gray + optical-flow resize (+pyramid) of the current device frame
 * become the "previous" images */
int fm_flow_init(fm_ctx* ctx);
/* first part of Flow.predict (flow.py:153-154): gray / small / pyramid of the current device
 * frame, enqueued on the flow stream (overlaps the detector) */
int fm_flow_begin(fm_ctx* ctx);
/* mask bookkeeping of flow.py:159-181 for nT tracks in closest-first order:
 * area_out[k]  = mask_area(crop(fg_mask, rect_k)) with every earlier rect already zeroed,
 * keep_out[i]  = _rect_filter verdict of propagated keypoint i (kp_off[k]..kp_off[k+1]) */
int fm_flow_targets(fm_ctx* ctx, int nT, const double* inside_tlbr, const float* kps, const int32_t* kp_off,
                    int32_t* area_out, uint8_t* keep_out);
/* cv2.goodFeaturesToTrack on the previous gray crop of each listed track (flow.py:169-178) with
 * the foreground mask of fm_flow_targets, followed by _ellipse_filter (flow.py:297-306).
 * track_idx: indices into the fm_flow_targets arrays; track_tlbr: full (unclipped) boxes;
 * min_dist: minDistance per track.  pts_out: [n][cap][2] f32 frame coordinates, counts_out[n]. */
int fm_flow_detect(fm_ctx* ctx, int n, const int32_t* track_idx, const double* track_tlbr,
                   const int32_t* min_dist, int cap, float* pts_out, int32_t* counts_out);
/* background keypoints (flow.py:187-200): INTER_LINEAR resize of the previous gray frame,
 * INTER_NEAREST resize of the final foreground mask, FAST-9/16 + NMS, mask filter.
 * pts_out: [cap][2] f32 in background-frame coordinates (not yet unscaled). */
int fm_flow_background(fm_ctx* ctx, int cap, float* pts_out, int* n_out);
/* fm_flow_targets + fm_flow_detect (for the tracks that turn out to need new keypoints,
 * flow.py:167-178: len(kept) < feat_density * area, minDistance = max(round(sqrt(area) *
 * feat_dist_factor), 1)) + fm_flow_background in ONE stream round trip.  New keypoints are returned
 * compacted: track k owns new_pts[new_off[k] .. new_off[k] + new_cnt[k]). */
int fm_flow_prepare(fm_ctx* ctx, int nT, const double* inside_tlbr, const double* full_tlbr,
                    const float* kps, const int32_t* kp_off, double feat_density, double feat_dist_factor,
                    int32_t* area_out, uint8_t* keep_out, uint8_t* needy_out, int pts_cap,
                    float* new_pts_out, int32_t* new_off_out, int32_t* new_cnt_out, int* n_new_out,
                    int bg_cap, float* bg_pts_out, int* n_bg_out);
/* cv2.calcOpticalFlowPyrLK(prev_small, cur_small, pts) (flow.py:205-207): Scharr derivatives +
 * pyramidal LK for n points; then the frame buffers are swapped (flow.py:212-213). */
int fm_flow_lk(fm_ctx* ctx, int n, const float* prev_pts, float* next_pts, uint8_t* status, float* err);
/* swap without LK (failure paths of flow.py:191-196) */
int fm_flow_swap(fm_ctx* ctx);
/* second half of Flow.predict on the host side of the library (flow.py:215-263): camera motion by
 * cv2.findHomography(RANSAC) over the background matches, then per track _fg_filter ->
 * cv2.estimateAffinePartial2D(RANSAC) -> _estimate_bbox -> inlier bookkeeping.  Latency-bound
 * serial work (a few microseconds per hypothesis loop): kept in C++ on the host, see DESIGN.md.
 * Inputs: prev/cur points (frame coordinates), status, target ranges [begin,end) per track
 * (closest-first), bg range, boxes.  Outputs: H (3x3), ok flag, per track result code
 * (0 = skipped, 1 = box estimated), est_tlbr, n_matched, inlier flags over all points. */
int fm_flow_estimate(fm_ctx* ctx, int n_pts, const float* prev_pts, const float* cur_pts,
                     const uint8_t* status, int nT, const int32_t* begins, const int32_t* ends,
                     int bg_begin, int bg_end, const double* track_tlbr, int frame_w, int frame_h,
                     int ransac_max_iter, double ransac_conf, int inlier_thresh,
                     double* H_out, int* ok_out, int32_t* result_out, double* est_tlbr_out,
                     int32_t* n_matched_out, uint8_t* inlier_out);
/* Flow.predict (flow.py:135-264) as ONE call = fm_flow_begin + fm_flow_prepare + the keypoint list
 * assembly + fm_flow_lk + fm_flow_estimate, with the NumPy glue arithmetic of the reference (float32
 * scale products) done in the library.  Inputs as fm_flow_prepare (tracks closest-first).
 * Outputs: status (FM_FLOW_OK / NO_BACKGROUND: no background keypoints, buffers swapped, flow.py:191-196 /
 * NO_HOMOGRAPHY: flow.py:227-231); H (3x3); per track result code (0 skipped, 1 box estimated, 2 estimated
 * but rejected), est_tlbr, n_matched; the RANSAC-inlier keypoints compacted in prev_out / cur_out
 * ([pts_cap][2] f32, frame coordinates): track k owns [trk_off[k], trk_off[k+1]) (empty when result 0),
 * the background inliers [bg_range[0], bg_range[1]). */
typedef struct fm_flow_predict_params {
    double feat_density, feat_dist_factor;
    float opt_scale[2], bg_scale[2];      /* opt_flow_scale_factor, bg_feat_scale_factor as float32 */
    double max_error;                     /* compared as float32, like NumPy's err < max_error */
    int32_t ransac_max_iter;
    double ransac_conf;
    int32_t inlier_thresh;
    int32_t frame_w, frame_h;
} fm_flow_predict_params;
enum { FM_FLOW_OK = 0, FM_FLOW_NO_BACKGROUND = 1, FM_FLOW_NO_HOMOGRAPHY = 2 };
int fm_flow_predict(fm_ctx* ctx, int nT, const double* inside_tlbr, const double* full_tlbr, const float* kps,
                    const int32_t* kp_off, const fm_flow_predict_params* prm, int pts_cap, float* prev_out,
                    float* cur_out, int32_t* trk_off_out, int32_t* bg_range_out, double* H_out, int* status_out,
                    int32_t* result_out, double* est_tlbr_out, int32_t* n_matched_out);
/* The KLT + Kalman chain of one step on the library's own worker thread: fm_flow_predict followed by fm_trk_step with
 * the KLT measurements MultiTracker.apply_kalman derives from it (tracker.py:150-183).  nT tracks in prediction order
 * (as fm_flow_predict), nK tracks in table order for the Kalman step: slots, ages and sorted_idx[i] = position of
 * track i among the nT predicted ones or -1; multiplier = max(age_penalty * age, 1) / inlier_ratio.  All pointers
 * must stay valid until fm_track_predict_wait returns; one job at a time PER CONTEXT (every fm_ctx owns its worker).
 * fm_track_predict_wait reports the prediction status (FM_FLOW_*) and whether the Kalman step ran. */
int fm_track_predict_async(fm_ctx* ctx, int nT, const double* inside_tlbr, const double* full_tlbr, const float* kps,
                           const int32_t* kp_off, const fm_flow_predict_params* prm, int pts_cap, float* prev_out,
                           float* cur_out, int32_t* trk_off_out, int32_t* bg_range_out, double* H_out,
                           int32_t* result_out, double* est_tlbr_out, int32_t* n_matched_out, int nK,
                           const int32_t* slots, const int32_t* ages, const int32_t* sorted_idx, double age_penalty,
                           double* tlbr_out, uint8_t* lost_out);
int fm_track_predict_wait(fm_ctx* ctx, int* status_out, int* kalman_done_out);
/* LDS bytes FM_OP_LITECHAIN needs for c channels on h x w maps (<= 65536 to be launchable): the layer-table
 * builder decides with the same formula whether an OSNet block can use the chain kernel (no device needed) */
size_t fm_litechain_lds_bytes(int c, int w, int h);
/* 1 when FM_OP_RESBLOCK supports c in/out channels with mid hidden channels (no device needed) */
int fm_resblock_supported(int c, int mid);
/* profiling hook: accumulated host wall time (ms) of the stages of fm_flow_predict -- out5 = {begin, prepare,
 * lk, estimate, number of calls}; reset != 0 clears the accumulators */
int fm_flow_timing(double* out5, int reset);
/* test hooks: read the device images (which: 0 prev gray, 1 cur gray, 2.. pyramid levels of prev
 * (2+l) and cur (10+l), 20 bg image) */
int fm_flow_read_image(fm_ctx* ctx, int which, uint8_t* out, int* w, int* h);

/* ------------------------------------------------------------------------------------------------------------------
 * Cross-stream ReID-gallery all-gather (multi-GPU: one process and one fm_ctx per GPU / video stream).  NOT in the
 * reference, which tracks one stream in one process; the exchanged rows extend the history side of _reid_cost
 * (tracker.py:355-366).  RCCL (librccl.so) is loaded on the first call.  fm_gallery_unique_id: rank 0 creates the
 * 128-byte communicator id and the application distributes it; fm_gallery_init (collective) joins it;
 * fm_gallery_allgather_async (collective) enqueues H2D + ncclAllGather + D2H of one fixed-size row per rank on a
 * side stream and returns; fm_gallery_allgather_wait copies out the world * row_bytes gathered rows, rank-major.
 * channel: a context owns two independent communicators, 0 = the gallery, 1 = small control messages of the
 * application (barrier / max-over-ranks of a benchmark); one exchange in flight per channel.
 * In-process constraint: RCCL resolves the HSA runtime by its bare library name; a process that has ALSO loaded
 * another ROCm copy (import torch) must run its collectives through that copy's RCCL (gallery.py: TorchComm). */
int fm_gallery_unique_id(char* out128);
int fm_gallery_init(fm_ctx* ctx, int channel, int world, int rank, const char* id128, size_t row_bytes);
int fm_gallery_allgather_async(fm_ctx* ctx, int channel, const void* send_row);
int fm_gallery_allgather_wait(fm_ctx* ctx, int channel, void* recv_rows, float* stream_ms_out);
int fm_gallery_destroy(fm_ctx* ctx, int channel);

#ifdef __cplusplus
}
#endif
#endif /* FASTMOT_HIP_H */

// Internal definitions shared by the libfastmot_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/fastmot_hip.h"
#ifdef FM_DIAG
#include "../../include/fastmot_hip_diag.h"
#endif

#define FM_ERR_HIP (-1)
#define FM_ERR_ARG (-2)
#define FM_ERR_STATE (-3)

void fm_set_error(const char* fmt, ...);

#define FM_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            fm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return FM_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define FM_CHECK_ARG(cond)                                                     \
    do {                                                                       \
        if (!(cond)) {                                                         \
            fm_set_error("%s:%d bad argument: %s", __FILE__, __LINE__, #cond); \
            return FM_ERR_ARG;                                                 \
        }                                                                      \
    } while (0)

// Growable device buffer with a pinned host mirror (HostDeviceMem of utils/inference.py:7-36).
// batches up to this many tracks exchange their (tiny) kernel inputs / outputs through pinned,
// device-mapped host memory instead of blit copies
bool fm_lap_host(const double* cost, int nr, int nc, long rs, long cs, int32_t* col4row_out);
#define FM_ZERO_COPY_TRACKS (ctx->opt_zero_copy_tracks)

struct DevBuf {
    void* d = nullptr;
    void* h = nullptr;   // pinned
    size_t cap = 0;
    bool pinned_mirror = true;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        size_t ncap = cap ? cap : 4096;
        while (ncap < bytes) ncap *= 2;
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        d = h = nullptr;
        cap = 0;
        FM_HIP(hipMalloc(&d, ncap));
        if (pinned_mirror) FM_HIP(hipHostMalloc(&h, ncap, hipHostMallocDefault));
        cap = ncap;
        return 0;
    }
    void release() {
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        d = h = nullptr;
        cap = 0;
    }
    template <typename T> T* dev() { return reinterpret_cast<T*>(d); }
    template <typename T> T* host() { return reinterpret_cast<T*>(h); }
};

struct KFConst {           // constants derived from fm_kf_params, passed by value to kernels
    double F_pos_self;     // vel_coupling * dt
    double F_pos_other;    // (1 - vel_coupling) * dt
    double F_vel;          // 0.5^(dt / half_life)
    double q_pp, q_pv, q_vv;   // dt^4/4, dt^3/2, dt^2
    double std_factor_acc, std_offset_acc;
    double fac_det[2], fac_klt[2], min_det[2], min_klt[2];
    double init_pos_weight, init_vel_weight;
};

struct NetState;     // conv engine (net.hip)
struct DetState;     // detector pre/post (detect.hip)
struct ExtState;     // extractor pre (extract.hip)
struct FlowState;    // KLT (flow.hip)
struct GalleryState; // cross-stream ReID-gallery all-gather over RCCL (gallery.hip)
constexpr int FM_GALLERY_CHANNELS = 2;

struct fm_ctx {
    int device = 0;
    // tunables (fm_ctx_set_option; initial values from the environment)
    int opt_zero_copy_tracks = 2048;   // FASTMOT_ZERO_COPY: 0 = always blit copies
    int opt_host_lap_elems = 262144;   // FASTMOT_HOST_LAP: cost matrices up to this size use lap_host.hip (measured: the
                                       // host solver is ~5x faster at every size up to 400 x 400, profiles/r02_lap_crossover.txt)
    int opt_use_graphs = 1;            // FASTMOT_GRAPHS: 0 = launch network layers one by one (no hipGraph)
    int opt_fused_input = 1;           // "fused_input" / FASTMOT_FUSED_INPUT: the networks' stem convolutions compute their input pixels from the
                                       // frame themselves (pixel_source.h); 0 = front-end kernel + input tensor (tests compare the two, A/B)
    int opt_net_timing = 0;            // "net_timing" = N: HIP events around the detector network on every N-th pass (fm_detect_net_ms); 0 = never
    int opt_nms_general = 0;           // "nms_path" = 1: always the three-kernel sort / bit matrix / scan path (tests, A/B)
    int opt_lk_variant = 0;            // "lk_variant": diagnostic variants of the LK kernel (flow.hip lk_diag_kernel)
    void* predict_worker = nullptr;    // native KLT + Kalman worker thread of this context (flow_estimate.hip)
    hipStream_t s_main = nullptr;   // tracker kernels
    hipStream_t s_det = nullptr;    // detector network
    hipStream_t s_up = nullptr;     // detector post-processing (sort, NMS, D2H), off the detector stream (detect.hip flush_post)
    hipStream_t s_ext = nullptr;    // ReID network
    hipStream_t s_ext_x[FM_MAX_EXTRA_EXTRACTORS] = {};   // streams of the extra ReID instances
    hipEvent_t ev_ext_in = nullptr, ev_ext_x_done[FM_MAX_EXTRA_EXTRACTORS] = {};
    hipStream_t s_flow = nullptr;   // KLT
    hipStream_t s_flow2 = nullptr;  // KLT: pyramid of the new frame (independent of the keypoint preparation on s_flow)
    hipEvent_t ev_pyr = nullptr;    // completion of that pyramid; s_flow waits on it before its first reader (LK)
    bool pyr_pending = false;
    hipEvent_t ev_prep = nullptr, ev_bg = nullptr;   // fork / join of the background-keypoint branch of fm_flow_prepare
    hipEvent_t ev_feat = nullptr;   // last reader of ctx->emb on s_main (fm_feat_update); s_ext waits on it
    hipEvent_t ev_ext_net = nullptr;   // the ReID network's last launch of a batch on s_ext (embeddings complete on the device):
                                       // fm_assoc_prepare2 orders the pairwise kernel behind it without the host
    hipEvent_t ev_ext_done = nullptr;  // embeddings exported to page-locked memory (fm_extract_sync waits for this, not for the stream)
    hipEvent_t ev_pair = nullptr;      // pairwise kernel of fm_assoc_prepare / fm_assoc_prepare2 done (its pinned mirror is complete)
    bool ext_net_recorded = false;     // ev_ext_net belongs to the batch ctx->emb holds
    size_t as_in_bytes = 0;
    bool as_in_device = true;          // as_in's device copy holds this frame's inputs (not for an early launch: fm_assoc_stage refuses)
    bool as_mirror = false;            // the pairwise terms of this frame also lie in as_pair's pinned mirror (host cascade)

    // ---- device-resident track table
    int slot_cap = 0;
    double* mean = nullptr;      // [cap][8]
    double* cov = nullptr;       // [cap][64]
    int feat_dim = 512;
    float* feat_sum = nullptr;   // [cap][dim]
    float* feat_avg = nullptr;   // [cap][dim]
    int32_t* feat_cnt = nullptr; // [cap]
    KFConst kf{};
    bool kf_set = false;
    double frame_rect[4] = {0, 0, 0, 0};

    // ---- per-frame embeddings on the device [n][dim] f32
    float* emb = nullptr;
    float* emb_host = nullptr;   // page-locked mirror [emb_cap][dim]: the ReID head writes a row to both (no export launch, round 6)
    int emb_cap = 0;
    int emb_n = 0;

    // ---- association scratch
    int as_nT = 0, as_nD = 0, as_metric = 0;
    size_t as_off[6] = {0, 0, 0, 0, 0, 0};   // slots|trk_tlbr|trk_label|det_tlbr|det_label|det_occ
    DevBuf as_in;       // packed inputs of fm_assoc_prepare
    DevBuf as_pair;     // feat | maha | iou  [3][nT][nD] f64
    DevBuf as_stage_in; // rows, cols, labels
    DevBuf as_cost;     // [nr][nc] f64
    DevBuf as_work;     // LAP work arrays
    DevBuf as_out;      // matches
    DevBuf io0, io1;    // generic staging (kalman etc.)
    DevBuf occ_in, occ_out;   // fm_find_occluded (own buffers: may run concurrently with the Kalman thread)
    DevBuf feat_in;     // fm_feat_update staging (own buffer: the call returns without synchronising)

    // ---- frames (BGR u8) resident on the device
    int frame_w = 0, frame_h = 0, ring_size = 0;
    uint8_t* frame_cur = nullptr;          // points into frame_own or the ring
    uint8_t* frame_own = nullptr;          // upload slot of the current frame
    uint8_t* frame_own2 = nullptr;         // second upload slot (prefetched next frame); the two swap roles
    uint8_t* frame_next = nullptr;         // frame the detector was prefetched on (fm_frame_*_next)
    uint8_t* frame_pinned2 = nullptr;
    hipEvent_t ev_next_upload = nullptr;   // completion of the prefetched frame's H2D copy (enqueued on the ReID stream)
    uint8_t* frame_ring = nullptr;
    uint8_t* frame_pinned = nullptr;

    DetState* det = nullptr;
    ExtState* ext = nullptr;
    NetState* det_net = nullptr;
    NetState* ext_net = nullptr;
    NetState* ext_net_x[FM_MAX_EXTRA_EXTRACTORS] = {};   // FM_NET_EXTRACTOR_B + i: further parts of a split batch
    FlowState* flow = nullptr;
    GalleryState* gallery[2] = {nullptr, nullptr};   // [FM_GALLERY_CHANNELS]

    // ---- event trace of the pipeline (fm_trace_start / fm_trace_read, scripts/trace_pipeline.py); empty = off
    std::vector<hipEvent_t> trace_ev;
    std::vector<int> trace_tag;
    std::atomic<int> trace_n{0};
    std::atomic<bool> trace_on{false};   // set once the two vectors are in place, cleared before they are taken away
    std::atomic<int> trace_busy{0};      // fm_trace_mark calls in flight (the prediction worker marks too): fm_trace_read
                                         // takes the vectors away only when none is left
    hipEvent_t trace_base = nullptr;
};

// one timed event on stream `s` (no-op unless a trace is running; both host threads of a context may call it)
inline void fm_trace_mark(fm_ctx* ctx, hipStream_t s, int tag) {
    if (!ctx->trace_on.load(std::memory_order_acquire)) return;
    ctx->trace_busy.fetch_add(1);                               // (seq_cst on both sides of the handshake)
    if (ctx->trace_on.load()) {       // (re-checked: a reader that disarmed waits for busy == 0)
        const int i = ctx->trace_n.fetch_add(1);
        if (i < (int)ctx->trace_ev.size()) {
            ctx->trace_tag[i] = tag;
            (void)hipEventRecord(ctx->trace_ev[i], s);
        }
    }
    ctx->trace_busy.fetch_sub(1);
}

int fm_ensure_slots(fm_ctx* ctx, int max_slot_plus_1);
void fm_ext_invalidate_export(fm_ctx* ctx);
void fm_predict_worker_free(fm_ctx* ctx);
void fm_gallery_free(fm_ctx* ctx);

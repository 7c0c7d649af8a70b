// ReID front end on the device: crop -> OpenCV-style fixed-point bilinear resize -> normalise.
//
// Replaces FeatureExtractor.extract_async/_preprocess/_normalize (fastmot/feature_extractor.py:
// 48-60,84-98) and multi_crop (fastmot/utils/rect.py:93-97).  The reference resizes every crop
// on the CPU (ThreadPool + cv2.resize) and uploads 393 KB per crop; here the crops are read from
// the frame that is already resident in HBM and written straight into the network input.
//
// cv2.resize(INTER_LINEAR) on 8-bit images is restated from OpenCV's published algorithm
// (imgproc/resize.cpp, 4.1.1 pinned by the reference Dockerfile:5): source coordinate
// (d + 0.5) * scale - 0.5, 11-bit fixed-point coefficients (INTER_RESIZE_COEF_SCALE = 2048),
// horizontal pass to int32, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
// Parity for this stage is pinned only against the restatement in oracle/cv_oracle.py (OpenCV is
// not available to run; SURVEY.md section 8c "parity unpinned").
// Roofline: HBM bound; per crop reads <= w*h*3 B of frame, writes 256*128*8*2 B = 512 KB.
#include "pixel_source.h"
#include <cmath>

struct ExtState {
    int input_tensor = -1, in_w = 0, in_h = 0;
    double* boxes = nullptr;       // device [cap][4]
    double* boxes_host = nullptr;  // pinned
    int cap = 0;
    DevBuf emb_out;                // pinned read-back buffer of fm_extract_sync
    int exported_n = -1;           // rows of ctx->emb that reached page-locked memory behind the last network pass
    bool mirror_export = false;    // ... in ctx->emb_host (written by the head layer itself) instead of emb_out.h (export_kernel)
};

// ctx->emb was written by something other than the extractor (fm_emb_upload): the exported copy is stale
void fm_ext_invalidate_export(fm_ctx* ctx) {
    if (ctx->ext) ctx->ext->exported_n = -1;
}

void fm_ext_free(ExtState* e) {
    if (!e) return;
    e->emb_out.release();
    if (e->boxes) (void)hipFree(e->boxes);
    if (e->boxes_host) (void)hipHostFree(e->boxes_host);
    delete e;
}

namespace {

// embeddings -> page-locked host memory, by a kernel behind the network instead of a device-to-host copy at collection
// time: no copy engine involved (see detect.hip flush_post for what a queued engine copy did to this pipeline), and
// the rows are on the host when the stream is
__global__ __launch_bounds__(256) void export_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n4) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) dst[i] = src[i];
}

__global__ void crop_resize_kernel(const uint8_t* __restrict__ frame, int fw, int fh,
                                   const double* __restrict__ boxes, int n, f16* __restrict__ out,
                                   int ow, int oh, int cs) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x >= ow) return;
    f16x8 o;
    crop_input_pixel(frame, fw, fh, boxes + (size_t)b * 4, x, y, ow, oh, o);     // (pixel_source.h: shared with the fused stem)
    *reinterpret_cast<f16x8*>(out + (((size_t)b * oh + y) * ow + x) * cs) = o;
}

}  // namespace

extern "C" int fm_extract_configure(fm_ctx* ctx, int input_tensor, int in_w, int in_h) {
    FM_CHECK_ARG(ctx && ctx->ext_net && input_tensor >= 0 && input_tensor < (int)ctx->ext_net->tensors.size());
    const fm_tensor& t = ctx->ext_net->tensors[input_tensor];
    FM_CHECK_ARG(t.h == in_h && t.w == in_w && !t.f32);
    if (!ctx->ext) ctx->ext = new ExtState();
    ctx->ext->input_tensor = input_tensor;
    ctx->ext->in_w = in_w;
    ctx->ext->in_h = in_h;
    return 0;
}

int fm_emb_reserve(fm_ctx* ctx, int n);

static int export_embeddings(fm_ctx* ctx, ExtState* e, int n, hipStream_t s) {
    const size_t bytes = sizeof(float) * (size_t)n * ctx->feat_dim;
    e->exported_n = -1;
    if (ctx->emb_host) {
        // round 6: the head layer wrote every row to ctx->emb_host as well (ops.hip head_kernel): nothing to launch
        e->exported_n = n;
        e->mirror_export = true;
        FM_HIP(hipEventRecord(ctx->ev_ext_done, s));
        return 0;
    }
    e->mirror_export = false;
    if (bytes > e->emb_out.cap) {             // (growing frees the old buffers: nothing may be in flight on them)
        FM_HIP(hipStreamSynchronize(s));
        int rc = e->emb_out.reserve(bytes);
        if (rc) return rc;
    }
    const int n4 = (int)(bytes / 16);
    hipLaunchKernelGGL(export_kernel, dim3((n4 + 255) / 256 < 64 ? (n4 + 255) / 256 : 64), dim3(256), 0, s,
                       (const float4*)ctx->emb, (float4*)e->emb_out.h, n4);
    FM_HIP(hipGetLastError());
    e->exported_n = n;
    // what fm_extract_sync waits for: the stream may go on with work of the association (the early pairwise launch)
    FM_HIP(hipEventRecord(ctx->ev_ext_done, s));
    return 0;
}

// crops [off, off + b) of the uploaded boxes -> the network's first activation.  Round 6: when the network begins with
// a stem convolution over its input tensor, that convolution computes crop -> resize -> normalise itself while it stages
// its patch (stemconv.hip, pixel_source.h; sets ni->first = 1 for the fm_net_run that follows): the crop launch and the
// input tensor -- 26 MB written and read back per 50 crops -- disappear.  Otherwise crop_resize_kernel fills the tensor.
static int front_end(fm_ctx* ctx, ExtState* e, NetState* ni, hipStream_t si, int off, int b, int cs) {
    ni->first = 0;
    if (ctx->opt_fused_input && fm_net_stem_fusable(ni, e->input_tensor)) {
        StemSrc src{};
        src.kind = 2; src.frame = ctx->frame_cur; src.fw = ctx->frame_w; src.fh = ctx->frame_h;
        src.boxes = e->boxes_host + (size_t)off * 4;
        int rc = fm_net_run_stem_from(ctx, ni, src, b);
        if (rc) return rc;
        ni->first = 1;
        return 0;
    }
    hipLaunchKernelGGL(crop_resize_kernel, dim3((e->in_w + 127) / 128, e->in_h, b), dim3(128), 0, si,
                       ctx->frame_cur, ctx->frame_w, ctx->frame_h, e->boxes + (size_t)off * 4, b,
                       (f16*)ni->bufs[e->input_tensor], e->in_w, e->in_h, cs);
    FM_HIP(hipGetLastError());
    return 0;
}

extern "C" int fm_extract_async(fm_ctx* ctx, int n, const double* tlbr) {
    FM_CHECK_ARG(ctx && ctx->ext && ctx->ext_net && n >= 0 && ctx->frame_cur);
    ctx->emb_n = 0;
    ctx->ext_net_recorded = false;
    int rc_exp = 0;
    if (ctx->ext) ctx->ext->exported_n = -1;
    if (n == 0) return 0;
    FM_CHECK_ARG(tlbr);
    ExtState* e = ctx->ext;
    NetState* net = ctx->ext_net;
    hipStream_t s = ctx->s_ext;
    if (n > e->cap) {
        FM_HIP(hipStreamSynchronize(s));
        if (e->boxes) FM_HIP(hipFree(e->boxes));
        if (e->boxes_host) FM_HIP(hipHostFree(e->boxes_host));
        e->boxes = e->boxes_host = nullptr;
        int cap = e->cap ? e->cap : 64;
        while (cap < n) cap *= 2;
        FM_HIP(hipMalloc(&e->boxes, sizeof(double) * 4 * cap));
        FM_HIP(hipHostMalloc(&e->boxes_host, sizeof(double) * 4 * cap, hipHostMallocDefault));
        e->cap = cap;
    }
    // fm_feat_update (s_main, asynchronous) may still read the previous frame's embeddings
    if (hipEventQuery(ctx->ev_feat) != hipSuccess) {     // (long complete in a running pipeline: no barrier packet then)
        (void)hipGetLastError();
        FM_HIP(hipStreamWaitEvent(s, ctx->ev_feat, 0));
    }
    if (n > ctx->emb_cap) {
        // association (s_main) may still read the previous embeddings
        FM_HIP(hipStreamSynchronize(ctx->s_main));
        FM_HIP(hipStreamSynchronize(s));
        int rc = fm_emb_reserve(ctx, n);
        if (rc) return rc;
    }
    FM_HIP(hipStreamSynchronize(s));   // boxes_host reuse
    memcpy(e->boxes_host, tlbr, sizeof(double) * 4 * n);
    // The fused stem reads the boxes straight from this page-locked buffer (a workgroup's four doubles, one trip over PCIe,
    // ~1.5 us): a blit copy in front of the network cost ~10 us on the chain detections -> embeddings.  The front-end
    // kernel of the unfused path reads every box from every thread: it keeps the device copy.
    const bool zero_copy_boxes = ctx->opt_fused_input && fm_net_stem_fusable(net, e->input_tensor);
    if (!zero_copy_boxes)
        FM_HIP(hipMemcpyAsync(e->boxes, e->boxes_host, sizeof(double) * 4 * n, hipMemcpyHostToDevice, s));
    fm_trace_mark(ctx, s, 32);
    const fm_tensor& t = net->tensors[e->input_tensor];
    // Several instances of the network (FM_NET_EXTRACTOR_B + i): the batch is cut into parts that run
    // concurrently on their own streams.  The ~30 dependent launches of OSNet are latency bound, so part-size
    // chains side by side finish sooner than one full-size chain; rows of ctx->emb are written by their owners.
    int parts = 1;
    while (parts <= FM_MAX_EXTRA_EXTRACTORS && ctx->ext_net_x[parts - 1]) ++parts;
    while (parts > 1 && n < 4 * parts) --parts;                 // at least 4 crops per part
    if (parts > 1 && (n + parts - 1) / parts <= net->max_batch) {
        FM_HIP(hipEventRecord(ctx->ev_ext_in, s));              // boxes uploaded, ev_feat honoured
        int off = 0;
        for (int i = 0; i < parts; ++i) {
            const int b = n / parts + (i < n % parts ? 1 : 0);
            NetState* ni = i == 0 ? net : ctx->ext_net_x[i - 1];
            hipStream_t si = i == 0 ? s : ctx->s_ext_x[i - 1];
            FM_CHECK_ARG(b <= ni->max_batch);
            if (i) FM_HIP(hipStreamWaitEvent(si, ctx->ev_ext_in, 0));
            int rc = front_end(ctx, e, ni, si, off, b, t.c);
            if (rc) return rc;
            ni->emb_offset = off;
            rc = fm_net_run_internal(ctx, i == 0 ? FM_NET_EXTRACTOR : FM_NET_EXTRACTOR_B + i - 1, b);
            ni->emb_offset = 0;
            ni->first = 0;
            if (rc) return rc;
            if (i) {                                            // everything downstream orders after s_ext only
                FM_HIP(hipEventRecord(ctx->ev_ext_x_done[i - 1], si));
                FM_HIP(hipStreamWaitEvent(s, ctx->ev_ext_x_done[i - 1], 0));
            }
            off += b;
        }
        FM_HIP(hipEventRecord(ctx->ev_ext_net, s));
        ctx->ext_net_recorded = true;
        if ((rc_exp = export_embeddings(ctx, e, n, s))) return rc_exp;
        fm_trace_mark(ctx, s, 33);
        ctx->emb_n = n;
        return 0;
    }
    for (int off = 0; off < n; off += net->max_batch) {
        const int b = n - off < net->max_batch ? n - off : net->max_batch;
        int rc = front_end(ctx, e, net, s, off, b, t.c);
        if (rc) return rc;
        fm_trace_mark(ctx, s, 34);
        net->emb_offset = off;
        rc = fm_net_run_internal(ctx, FM_NET_EXTRACTOR, b);
        net->emb_offset = 0;
        net->first = 0;
        if (rc) return rc;
    }
    fm_trace_mark(ctx, s, 35);
    FM_HIP(hipEventRecord(ctx->ev_ext_net, s));
    ctx->ext_net_recorded = true;
    if ((rc_exp = export_embeddings(ctx, e, n, s))) return rc_exp;
    fm_trace_mark(ctx, s, 33);
    ctx->emb_n = n;
    return 0;
}

extern "C" int fm_extract_sync(fm_ctx* ctx, int n, float* emb) {
    FM_CHECK_ARG(ctx && n >= 0 && n <= ctx->emb_cap);
    if (n == 0) {
        FM_HIP(hipStreamSynchronize(ctx->s_ext));
        return 0;
    }
    FM_CHECK_ARG(emb && ctx->ext);
    // read back on the extractor's own stream through a pinned buffer: a synchronous hipMemcpy runs on the
    // legacy NULL stream, which must not be mixed with the other host thread's asynchronous work
    const size_t bytes = sizeof(float) * (size_t)n * ctx->feat_dim;
    if (ctx->ext->exported_n != n) {          // (rows that did not come from fm_extract_async: plain copy)
        int rc = ctx->ext->emb_out.reserve(bytes);
        if (rc) return rc;
        FM_HIP(hipMemcpyAsync(ctx->ext->emb_out.h, ctx->emb, bytes, hipMemcpyDeviceToHost, ctx->s_ext));
        FM_HIP(hipStreamSynchronize(ctx->s_ext));
    } else {
        FM_HIP(hipEventSynchronize(ctx->ev_ext_done));   // the rows are in page-locked memory
        if (ctx->ext->mirror_export) {
            memcpy(emb, ctx->emb_host, bytes);
            return 0;
        }
    }
    memcpy(emb, ctx->ext->emb_out.h, bytes);
    return 0;
}

extern "C" int fm_extract_read_input(fm_ctx* ctx, int n, float* out) {
    FM_CHECK_ARG(ctx && ctx->ext && ctx->ext_net && n > 0 && n <= ctx->ext_net->max_batch && out);
    ExtState* e = ctx->ext;
    NetState* net = ctx->ext_net;
    const fm_tensor& t = net->tensors[e->input_tensor];
    if (ctx->opt_fused_input && fm_net_stem_fusable(net, e->input_tensor)) {
        // the fused stem never wrote the tensor: fill it now with the front-end kernel (same pixel function) from the
        // boxes of the last fm_extract_async, which are still on the device
        FM_CHECK_ARG(ctx->frame_cur && n <= e->cap);
        hipLaunchKernelGGL(crop_resize_kernel, dim3((e->in_w + 127) / 128, e->in_h, n), dim3(128), 0, ctx->s_ext,
                           ctx->frame_cur, ctx->frame_w, ctx->frame_h, e->boxes_host, n, (f16*)net->bufs[e->input_tensor],
                           e->in_w, e->in_h, t.c);
        FM_HIP(hipGetLastError());
    }
    FM_HIP(hipStreamSynchronize(ctx->s_ext));
    std::vector<f16> tmp((size_t)n * t.h * t.w * t.c);
    FM_HIP(hipMemcpy(tmp.data(), net->bufs[e->input_tensor], tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < (size_t)n * t.h * t.w; ++p)
        for (int c = 0; c < 3; ++c) out[p * 3 + c] = (float)tmp[p * t.c + c];
    return 0;
}

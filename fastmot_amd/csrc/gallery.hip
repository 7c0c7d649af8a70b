// Cross-stream ReID-gallery all-gather behind the C ABI (fm_gallery_allgather).  NOT in the reference
// (single process, single stream): with one video stream per GPU the only exchange between the ranks is every rank's
// lost-track gallery -- one fixed-size row per rank (header + {id, label, count} + average features, ~104 KB), see
// fastmot_amd/gallery.py for the wire format and the protocol.
//
// RCCL is bound at run time (dlopen librccl.so on the first fm_gallery_* call): a single-GPU deployment never loads
// it, and the library has no link-time dependency on it.  The collective runs on its own stream: H2D of the local row
// from pinned staging, ncclAllGather over xGMI, D2H of all rows, completion event -- fm_gallery_allgather_async returns
// after the enqueue, fm_gallery_allgather_wait publishes the rows (normally long finished: the tracker asks for them
// one detector frame later).  The payload is latency bound (<= 832 KB for 8 ranks): one collective per exchange.
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>

namespace {

constexpr int kUniqueIdBytes = 128;                 // NCCL_UNIQUE_ID_BYTES (rccl.h)
struct UniqueId { char internal[kUniqueIdBytes]; };
typedef void* Comm;

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int /*ncclDataType_t*/, Comm, hipStream_t) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string why;                              // the load error, kept: dlerror() hands a message out once
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            const char* e = dlerror();
            why = e ? e : "dlopen failed";
        }
        if (r.handle) {
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
            if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) why = "missing symbols";
        }
    });
    if (!r.handle || !r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
        fm_set_error("librccl.so could not be loaded: %s", why.c_str());
        return nullptr;
    }
    return &r;
}

#define FM_RCCL(call)                                                                                   \
    do {                                                                                                \
        const int rc_ = (call);                                                                         \
        if (rc_ != 0) {                                                                                 \
            fm_set_error("%s:%d RCCL error %d: %s", __FILE__, __LINE__, rc_,                            \
                         r->GetErrorString ? r->GetErrorString(rc_) : "?");                             \
            return FM_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

}  // namespace

struct GalleryState {
    Comm comm = nullptr;
    int world = 0, rank = 0;
    size_t row_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *send_host = nullptr, *recv_host = nullptr;      // pinned
    void *send_dev = nullptr, *recv_dev = nullptr;
    bool pending = false;
};

static void gallery_state_free(GalleryState* g) {
    if (!g) return;
    if (g->pending) (void)hipEventSynchronize(g->ev1);
    if (g->comm) {
        Rccl* r = rccl();
        if (r) (void)r->CommDestroy(g->comm);
    }
    for (void* p : {g->send_dev, g->recv_dev})
        if (p) (void)hipFree(p);
    for (void* p : {g->send_host, g->recv_host})
        if (p) (void)hipHostFree(p);
    for (hipEvent_t e : {g->ev0, g->ev1})
        if (e) (void)hipEventDestroy(e);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

static void gallery_free_channel(fm_ctx* ctx, int channel) {
    gallery_state_free(ctx->gallery[channel]);
    ctx->gallery[channel] = nullptr;
}

void fm_gallery_free(fm_ctx* ctx) {
    for (int c = 0; c < FM_GALLERY_CHANNELS; ++c) gallery_free_channel(ctx, c);
}

static int gallery_state_build(GalleryState* g, Rccl* r, const char* id128) {
    FM_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    FM_HIP(hipEventCreate(&g->ev0));
    FM_HIP(hipEventCreate(&g->ev1));
    FM_HIP(hipHostMalloc(&g->send_host, g->row_bytes, hipHostMallocDefault));
    FM_HIP(hipHostMalloc(&g->recv_host, g->row_bytes * g->world, hipHostMallocDefault));
    FM_HIP(hipMalloc(&g->send_dev, g->row_bytes));
    FM_HIP(hipMalloc(&g->recv_dev, g->row_bytes * g->world));
    UniqueId id;
    memcpy(id.internal, id128, kUniqueIdBytes);
    FM_RCCL(r->CommInitRank(&g->comm, g->world, id, g->rank));
    return 0;
}

// rank 0 creates the communicator id; the application hands the 128 bytes to the other ranks (any channel)
extern "C" int fm_gallery_unique_id(char* out128) {
    FM_CHECK_ARG(out128);
    Rccl* r = rccl();
    if (!r) return FM_ERR_STATE;
    UniqueId id;
    FM_RCCL(r->GetUniqueId(&id));
    memcpy(out128, id.internal, kUniqueIdBytes);
    return 0;
}

// collective over all ranks: joins the communicator `id128` as rank `rank` of `world` on the context's device.
// channel: a context owns FM_GALLERY_CHANNELS independent communicators (0 = the ReID gallery; 1 = small control
// messages of the application, e.g. the barrier and the max-over-ranks of a benchmark) -- one exchange in flight each
extern "C" int fm_gallery_init(fm_ctx* ctx, int channel, int world, int rank, const char* id128, size_t row_bytes) {
    FM_CHECK_ARG(ctx && id128 && world >= 1 && rank >= 0 && rank < world && row_bytes > 0 && row_bytes % 8 == 0);
    FM_CHECK_ARG(channel >= 0 && channel < FM_GALLERY_CHANNELS);
    if (ctx->gallery[channel]) gallery_free_channel(ctx, channel);
    Rccl* r = rccl();
    if (!r) return FM_ERR_STATE;
    FM_HIP(hipSetDevice(ctx->device));
    // built aside and published to the context only when every step -- the communicator last -- has succeeded: a
    // failed init leaves the channel empty, never a half-built state with a null communicator
    GalleryState* g = new GalleryState();
    g->world = world; g->rank = rank; g->row_bytes = row_bytes;
    const int rc = gallery_state_build(g, r, id128);
    if (rc) {
        gallery_state_free(g);
        return rc;
    }
    ctx->gallery[channel] = g;
    return 0;
}

__global__ __launch_bounds__(256) void rows_to_host_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes) {
    const size_t n16 = bytes / 16, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (size_t i = n16 * 16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < bytes; i += stride) dst[i] = src[i];
}

// enqueues one all-gather of this rank's row (row_bytes, copied before the call returns); collective
extern "C" int fm_gallery_allgather_async(fm_ctx* ctx, int channel, const void* send_row) {
    FM_CHECK_ARG(ctx && channel >= 0 && channel < FM_GALLERY_CHANNELS && ctx->gallery[channel] && send_row);
    GalleryState* g = ctx->gallery[channel];
    if (g->pending) {
        fm_set_error("an all-gather is already in flight (fm_gallery_allgather_wait first)");
        return FM_ERR_STATE;
    }
    Rccl* r = rccl();
    if (!r) return FM_ERR_STATE;
    FM_CHECK_ARG(g->comm != nullptr);
    memcpy(g->send_host, send_row, g->row_bytes);
    FM_HIP(hipEventRecord(g->ev0, g->stream));
    FM_HIP(hipMemcpyAsync(g->send_dev, g->send_host, g->row_bytes, hipMemcpyHostToDevice, g->stream));
    FM_RCCL(r->AllGather(g->send_dev, g->recv_dev, g->row_bytes, 1 /* ncclUint8 */, g->comm, g->stream));
    // the gathered rows go to page-locked host memory by a kernel, not by a copy-engine transfer: a device-to-host copy
    // queued here would occupy its engine until the collective -- i.e. the slowest rank -- is through, and hold up the
    // tracker's own read-backs that land on the same engine (detect.hip flush_post has the measurement)
    {
        const size_t bytes = g->row_bytes * (size_t)g->world;
        hipLaunchKernelGGL(rows_to_host_kernel, dim3(bytes / 16 / 256 + 1 < 64 ? bytes / 16 / 256 + 1 : 64), dim3(256), 0,
                           g->stream, (const uint8_t*)g->recv_dev, (uint8_t*)g->recv_host, bytes);
        FM_HIP(hipGetLastError());
    }
    FM_HIP(hipEventRecord(g->ev1, g->stream));
    g->pending = true;
    return 0;
}

// waits for the in-flight all-gather and copies the world * row_bytes rows (rank-major) out; stream_ms_out (optional):
// time the exchange occupied its stream
extern "C" int fm_gallery_allgather_wait(fm_ctx* ctx, int channel, void* recv_rows, float* stream_ms_out) {
    FM_CHECK_ARG(ctx && channel >= 0 && channel < FM_GALLERY_CHANNELS && ctx->gallery[channel] && recv_rows);
    GalleryState* g = ctx->gallery[channel];
    if (!g->pending) {
        fm_set_error("no all-gather in flight");
        return FM_ERR_STATE;
    }
    FM_HIP(hipEventSynchronize(g->ev1));
    g->pending = false;
    memcpy(recv_rows, g->recv_host, g->row_bytes * g->world);
    if (stream_ms_out) FM_HIP(hipEventElapsedTime(stream_ms_out, g->ev0, g->ev1));
    return 0;
}

extern "C" int fm_gallery_destroy(fm_ctx* ctx, int channel) {
    FM_CHECK_ARG(ctx && channel >= 0 && channel < FM_GALLERY_CHANNELS);
    gallery_free_channel(ctx, channel);
    return 0;
}

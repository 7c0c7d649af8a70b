// Layer-table executor of the conv engine (detector / ReID networks).
// Replaces TRTInference (fastmot/utils/inference.py:39-125): buffers are allocated once, a run
// enqueues every layer on the network's own HIP stream; nothing synchronises until the caller
// asks for results (detect_async/postprocess protocol of fastmot/detector.py:26-42).
#include "pixel_source.h"
#include <memory>

int launch_dwconv3(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, const f16* w,
                   const float* bias, int N, int H, int W, int C, int act, hipStream_t s);
int launch_liteconv(int G, const f16* const* in, const int* in_cs, const int* in_coff, f16* out, int out_cs,
                    int out_coff, const f16* wpw, int kpad, const f16* wdw, const float* bias, int N, int H,
                    int W, int C, int act, float* gap_out, hipStream_t s);
void liteconv_tiling(int C, int W, int H, int* th, int* tw, int* tiles_x, int* tiles_y);
int launch_litechain(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, const f16* wpw,
                     int kpad, const f16* wdw, const float* bias, int N, int H, int W, int C, int act,
                     float* const* gap, hipStream_t s);
int launch_gated_sum(int nstreams, const f16* const* in, const int* in_cs, const int* in_coff, int N, int HW,
                     int C, int hid, const f16* w1, const float* b1, const f16* w2, const float* b2, f16* out,
                     int out_cs, int out_coff, const float* const* part, int tiles, hipStream_t s);
// one gate slot: [max_batch][GATE_SLOT_TILES][gate_c] fp32 (gate values use the first [max_batch][gate_c])
constexpr int GATE_SLOT_TILES = 32;
int launch_pair11(const f16* xa, int xa_cs, int xa_coff, const f16* xc, int xc_cs, int xc_coff, f16* out, int out_cs,
                  int out_coff, const f16* w1, const float* b1, const f16* w2, const float* b2, long P, int cout, int act1,
                  int act2, hipStream_t s);
bool pair11_supported(int cin, int mid, int extra, int cout);
int launch_ostail(const f16* in, int in_cs, int in_coff, const f16* ph, const float* pf, int N, float* out, float* raw_out,
                  float* mirror, hipStream_t s);
bool ostail_supported(int in_h, int in_w, int cin, int mid, int cout, int feat, int n_halfs, int n_floats);
int launch_stemconv(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, const f16* w,
                    const float* bias, int N, int H, int W, int Ho, int Wo, int k, int stride, int pad, int cout,
                    int act, hipStream_t s);
int launch_spp(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, int N, int H, int W,
               int C, hipStream_t s);
int launch_pool(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, int N, int H,
                int W, int C, int Ho, int Wo, int k, int stride, int pad, int avg, hipStream_t s);
int launch_upsample2(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, int N, int H,
                     int W, int C, hipStream_t s);
int launch_conv_streamed(const ConvParams& p, hipStream_t s);
int launch_convd(const ConvParams& p, hipStream_t s);
bool resblock_supported(int C, int M);
int launch_resblock(const f16* x, int x_cs, int x_coff, f16* out, int out_cs, int out_coff, const f16* w1,
                    const float* b1, const f16* w2, const float* b2, int N, int H, int W, int C, int M, int act1,
                    int act2, hipStream_t s);
int launch_add(const f16* a, int a_cs, int a_coff, const f16* b, int b_cs, int b_coff, f16* out, int out_cs,
               int out_coff, long npix, int C, hipStream_t s);
int launch_copy(const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff, long npix, int C,
                hipStream_t s);
int launch_gate(const f16* in, int in_cs, int in_coff, int N, int HW, int C, int hid, const f16* w1,
                const float* b1, const f16* w2, const float* b2, float* gate, hipStream_t s);
int launch_gate_sum(int nstreams, const f16* const* in, const int* in_cs, const int* in_coff,
                    const float* const* gate, f16* out, int out_cs, int out_coff, int N, int HW, int C,
                    hipStream_t s);
int launch_head(const f16* in, int in_cs, int in_coff, int N, int HW, int C, int D, const f16* w,
                const float* b, float* out, float* raw_out, hipStream_t s, float* mirror = nullptr);
int fm_emb_reserve(fm_ctx* ctx, int n);


void fm_net_free(NetState* n) {
    if (!n) return;
    for (size_t i = 0; i < n->bufs.size(); ++i)
        if (n->bufs[i] && n->tensors[i].offset < 0) (void)hipFree(n->bufs[i]);
    if (n->arena) (void)hipFree(n->arena);
    if (n->weights) (void)hipFree(n->weights);
    if (n->gates) (void)hipFree(n->gates);
    if (n->ws) (void)hipFree(n->ws);
    for (auto& g : n->graphs) (void)hipGraphExecDestroy(g.second);
    delete n;
}

static NetState*& net_slot(fm_ctx* ctx, int which) {
    return which == FM_NET_DETECTOR ? ctx->det_net : which == FM_NET_EXTRACTOR ? ctx->ext_net
                                                    : ctx->ext_net_x[which - FM_NET_EXTRACTOR_B];
}
NetState* fm_net_get(fm_ctx* ctx, int which) { return net_slot(ctx, which); }
int fm_net_run_internal(fm_ctx* ctx, int which, int batch) { return fm_net_run(ctx, which, batch); }

// captured graphs bake the pointers their layers were launched with: whoever re-allocates one of those drops the graphs
void fm_net_drop_graphs(NetState* net) {
    if (!net) return;
    (void)hipStreamSynchronize(net->stream);
    for (auto& g : net->graphs) (void)hipGraphExecDestroy(g.second);
    net->graphs.clear();
}

static size_t elem_size(const fm_tensor& t) { return t.f32 ? 4 : 2; }

extern "C" int fm_net_destroy(fm_ctx* ctx, int which) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS));
    NetState*& slot = net_slot(ctx, which);
    if (slot) {
        FM_HIP(hipStreamSynchronize(slot->stream));
        fm_net_free(slot);
        slot = nullptr;
    }
    return 0;
}

extern "C" int fm_net_create(fm_ctx* ctx, int which, int max_batch, int n_tensors, const fm_tensor* tensors,
                             int n_layers, const fm_layer* layers, const void* weights, size_t weight_bytes,
                             int n_gates, int gate_channels, size_t arena_bytes) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS) && max_batch > 0 && n_tensors > 0 && n_layers > 0);
    FM_CHECK_ARG(tensors && layers && weights && weight_bytes > 0);
    int rc = fm_net_destroy(ctx, which);
    if (rc) return rc;
    std::unique_ptr<NetState, void (*)(NetState*)> net(new NetState(), fm_net_free);
    net->which = which;
    net->max_batch = max_batch;
    net->stream = which == FM_NET_DETECTOR ? ctx->s_det : which == FM_NET_EXTRACTOR ? ctx->s_ext
                                                        : ctx->s_ext_x[which - FM_NET_EXTRACTOR_B];
    net->tensors.assign(tensors, tensors + n_tensors);
    net->layers.assign(layers, layers + n_layers);
    if (arena_bytes) {
        FM_HIP(hipMalloc(&net->arena, arena_bytes));
        FM_HIP(hipMemset(net->arena, 0, arena_bytes));
    }
    for (const fm_tensor& t : net->tensors) {
        FM_CHECK_ARG(t.h > 0 && t.w > 0 && t.c > 0 && t.c % 8 == 0);
        void* b = nullptr;
        const size_t bytes = (size_t)max_batch * t.h * t.w * t.c * elem_size(t);
        if (t.offset >= 0) {
            FM_CHECK_ARG(t.offset % 256 == 0 && (size_t)t.offset + bytes <= arena_bytes);
            b = net->arena + t.offset;
        } else {
            FM_HIP(hipMalloc(&b, bytes));
            FM_HIP(hipMemset(b, 0, bytes));
        }
        net->bufs.push_back(b);
    }
    for (const fm_layer& L : net->layers) {
        FM_CHECK_ARG(L.out >= 0 && L.out < n_tensors && L.n_in >= 1 && L.n_in <= 4);
        for (int i = 0; i < L.n_in; ++i) FM_CHECK_ARG(L.in[i] >= 0 && L.in[i] < n_tensors);
        FM_CHECK_ARG(L.res_mode == FM_RES_NONE || (L.res >= 0 && L.res < n_tensors));
        FM_CHECK_ARG(L.w_off >= 0 && (size_t)L.w_off <= weight_bytes && L.w_off % 16 == 0 && L.b_off % 16 == 0);
        if (L.op == FM_OP_OSTAIL)       // its two parameter blobs carry their sizes
            FM_CHECK_ARG(L.stride > 0 && L.pad > 0 && L.b_off >= 0 && (size_t)L.w_off + (size_t)L.stride * 2 <= weight_bytes &&
                         (size_t)L.b_off + (size_t)L.pad * 4 <= weight_bytes);
    }
    FM_HIP(hipMalloc(&net->weights, weight_bytes));
    FM_HIP(hipMemcpy(net->weights, weights, weight_bytes, hipMemcpyHostToDevice));
    net->weight_bytes = weight_bytes;
    net->n_gates = n_gates;
    net->gate_c = gate_channels;
    net->ws_floats = (size_t)16 << 20;   // 64 MB of fp32 split-K partials
    FM_HIP(hipMalloc(&net->ws, net->ws_floats * sizeof(float)));
    if (n_gates > 0) {
        FM_CHECK_ARG(gate_channels > 0);
        FM_HIP(hipMalloc(&net->gates, sizeof(float) * (size_t)n_gates * max_batch * gate_channels * GATE_SLOT_TILES));
    }
    if (which != FM_NET_DETECTOR) {
        FM_HIP(hipStreamSynchronize(ctx->s_main));
        if ((rc = fm_emb_reserve(ctx, max_batch))) return rc;
    }
    net_slot(ctx, which) = net.release();
    return 0;
}

static int run_layer(fm_ctx* ctx, NetState* net, const fm_layer& L, int B) {
    const fm_tensor& ti = net->tensors[L.in[0]];
    const fm_tensor& to = net->tensors[L.out];
    hipStream_t s = net->stream;
    float* ws = net->ws;
    const f16* in0 = (const f16*)net->bufs[L.in[0]];
    f16* out = (f16*)net->bufs[L.out];
    switch (L.op) {
        case FM_OP_CONV:
        case FM_OP_CONVS:
        case FM_OP_CONVD: {
            ConvParams p{};
            p.in = in0; p.in_cs = ti.c; p.in_coff = L.in_coff[0];
            p.w = (const f16*)(net->weights + L.w_off);
            p.bias = (const float*)(net->weights + L.b_off);
            p.out = to.f32 ? nullptr : out;
            p.out32 = to.f32 ? (float*)net->bufs[L.out] : nullptr;
            p.out_cs = to.c; p.out_coff = L.out_coff;
            if (L.res_mode != FM_RES_NONE) {
                p.res = (const f16*)net->bufs[L.res];
                p.res_cs = net->tensors[L.res].c;
                p.res_coff = L.res_coff;
            }
            p.up = L.up == 2 ? 2 : 1;
            FM_CHECK_ARG(L.up <= 2 && (p.up == 1 || (!to.f32 && to.h % 2 == 0 && to.w % 2 == 0)));
            p.N = B; p.H = ti.h; p.W = ti.w; p.Cin = L.cin; p.Ho = to.h / p.up; p.Wo = to.w / p.up; p.Cout = L.cout;
            p.KH = p.KW = L.k; p.stride = L.stride; p.pad = L.pad;
            p.K = L.k * L.k * L.cin; p.Kpad = (p.K + 63) & ~63; p.P = B * p.Ho * p.Wo;
            p.cout_store = (L.cout + 7) & ~7;
            p.act = L.act; p.res_mode = L.res_mode;
            FM_CHECK_ARG((ti.h + 2 * L.pad - L.k) / L.stride + 1 == p.Ho);
            FM_CHECK_ARG(L.out_coff + p.cout_store <= to.c && L.in_coff[0] + L.cin <= ti.c);
            if (L.op == FM_OP_CONVS) return launch_conv_streamed(p, s);
            if (L.op == FM_OP_CONVD) return launch_convd(p, s);
            return launch_conv(p, ws, net->ws_floats, s);
        }
        case FM_OP_DWCONV3:
            return launch_dwconv3(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff,
                                  (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off),
                                  B, ti.h, ti.w, L.cin, L.act, s);
        case FM_OP_LITECONV: {
            FM_CHECK_ARG(L.cin == ((L.cout + 7) & ~7) && L.n_in >= 1 && L.n_in <= 4);
            FM_CHECK_ARG(L.out_coff + L.n_in * L.cin <= to.c);
            const f16* ins[4];
            int cs[4], co[4];
            for (int i = 0; i < L.n_in; ++i) {
                const fm_tensor& tg = net->tensors[L.in[i]];
                FM_CHECK_ARG(tg.h == to.h && tg.w == to.w && L.in_coff[i] + L.cin <= tg.c);
                ins[i] = (const f16*)net->bufs[L.in[i]];
                cs[i] = tg.c;
                co[i] = L.in_coff[i];
            }
            float* gap_out = nullptr;
            if (L.gate[0] >= 0) {   // group 0 ends a stream: leave per-tile channel sums for FM_OP_GATED_SUM
                int th, tw, tx, ty;
                liteconv_tiling(L.cin, ti.w, ti.h, &th, &tw, &tx, &ty);
                FM_CHECK_ARG(L.gate[0] < net->n_gates && L.cin <= net->gate_c && tx * ty <= GATE_SLOT_TILES);
                gap_out = net->gates + (size_t)L.gate[0] * net->max_batch * net->gate_c * GATE_SLOT_TILES;
            }
            return launch_liteconv(L.n_in, ins, cs, co, out, to.c, L.out_coff,
                                   (const f16*)(net->weights + L.w_off), (L.cin + 63) & ~63,
                                   (const f16*)(net->weights + L.w2_off), (const float*)(net->weights + L.b_off),
                                   B, ti.h, ti.w, L.cin, L.act, gap_out, s);
        }
        case FM_OP_LITECHAIN: {
            FM_CHECK_ARG(L.cin == ((L.cout + 7) & ~7) && to.h == ti.h && to.w == ti.w && L.in_coff[0] + L.cin <= ti.c &&
                         L.out_coff + 4 * L.cin <= to.c);
            int th, tw, tx, ty;
            liteconv_tiling(L.cin, ti.w, ti.h, &th, &tw, &tx, &ty);
            FM_CHECK_ARG(L.cin <= net->gate_c && tx * ty <= GATE_SLOT_TILES);
            float* gaps[4];
            for (int i = 0; i < 4; ++i) {
                FM_CHECK_ARG(L.gate[i] >= 0 && L.gate[i] < net->n_gates);
                gaps[i] = net->gates + (size_t)L.gate[i] * net->max_batch * net->gate_c * GATE_SLOT_TILES;
            }
            return launch_litechain(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff,
                                    (const f16*)(net->weights + L.w_off), (L.cin + 63) & ~63,
                                    (const f16*)(net->weights + L.w2_off), (const float*)(net->weights + L.b_off),
                                    B, ti.h, ti.w, L.cin, L.act, gaps, s);
        }
        case FM_OP_GATED_SUM: {
            const f16* ins[4];
            int cs[4], co[4];
            FM_CHECK_ARG(L.n_in >= 1 && L.n_in <= 4);
            for (int i = 0; i < L.n_in; ++i) {
                ins[i] = (const f16*)net->bufs[L.in[i]];
                cs[i] = net->tensors[L.in[i]].c;
                co[i] = L.in_coff[i];
            }
            const float* parts[4] = {nullptr, nullptr, nullptr, nullptr};
            int tiles = 0;
            if (L.gate[0] >= 0) {   // producers left per-tile channel sums (FM_OP_LITECONV with gate[0] >= 0)
                int th, tw, tx, ty;
                liteconv_tiling(L.cin, ti.w, ti.h, &th, &tw, &tx, &ty);
                tiles = tx * ty;
                for (int i = 0; i < L.n_in; ++i) {
                    FM_CHECK_ARG(L.gate[i] >= 0 && L.gate[i] < net->n_gates);
                    parts[i] = net->gates + (size_t)L.gate[i] * net->max_batch * net->gate_c * GATE_SLOT_TILES;
                }
            }
            return launch_gated_sum(L.n_in, ins, cs, co, B, ti.h * ti.w, L.cin, L.hid,
                                    (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off),
                                    (const f16*)(net->weights + L.w2_off), (const float*)(net->weights + L.b2_off),
                                    out, to.c, L.out_coff, L.gate[0] >= 0 ? parts : nullptr, tiles, s);
        }
        case FM_OP_STEMCONV:
            FM_CHECK_ARG(!to.f32 && (ti.h + 2 * L.pad - L.k) / L.stride + 1 == to.h &&
                         (ti.w + 2 * L.pad - L.k) / L.stride + 1 == to.w && L.out_coff + ((L.cout + 7) & ~7) <= to.c);
            return launch_stemconv(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff,
                                   (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off),
                                   B, ti.h, ti.w, to.h, to.w, L.k, L.stride, L.pad, L.cout, L.act, s);
        case FM_OP_PAIR11: {
            FM_CHECK_ARG(L.n_in == 2 && !to.f32 && pair11_supported(L.cin, L.hid, L.cin, L.cout));
            const fm_tensor& tc = net->tensors[L.in[1]];
            FM_CHECK_ARG(tc.h == ti.h && tc.w == ti.w && to.h == ti.h && to.w == ti.w && L.in_coff[0] + 64 <= ti.c &&
                         L.in_coff[1] + 64 <= tc.c && L.out_coff + L.cout <= to.c);
            return launch_pair11(in0, ti.c, L.in_coff[0], (const f16*)net->bufs[L.in[1]], tc.c, L.in_coff[1], out, to.c, L.out_coff,
                                 (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off),
                                 (const f16*)(net->weights + L.w2_off), (const float*)(net->weights + L.b2_off),
                                 (long)B * ti.h * ti.w, L.cout, L.gate[0], L.act, s);
        }
        case FM_OP_STEM2:
            return launch_stem2_layer(L, StemSrc{}, net, B, s);
        case FM_OP_SPP:
            FM_CHECK_ARG(to.h == ti.h && to.w == ti.w && L.out_coff + 3 * L.cin <= to.c);
            return launch_spp(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff, B, ti.h, ti.w, L.cin, s);
        case FM_OP_MAXPOOL:
        case FM_OP_AVGPOOL:
            return launch_pool(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff, B, ti.h, ti.w, L.cin, to.h, to.w,
                               L.k, L.stride, L.pad, L.op == FM_OP_AVGPOOL, s);
        case FM_OP_UPSAMPLE2:
            FM_CHECK_ARG(to.h == 2 * ti.h && to.w == 2 * ti.w);
            return launch_upsample2(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff, B, ti.h, ti.w, L.cin, s);
        case FM_OP_RESBLOCK:
            FM_CHECK_ARG(!to.f32 && to.h == ti.h && to.w == ti.w && L.cin == L.cout &&
                         L.in_coff[0] + L.cin <= ti.c && L.out_coff + L.cout <= to.c);
            return launch_resblock(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff,
                                   (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off),
                                   (const f16*)(net->weights + L.w2_off), (const float*)(net->weights + L.b2_off),
                                   B, ti.h, ti.w, L.cin, L.hid, L.act, L.act, s);
        case FM_OP_ADD: {
            FM_CHECK_ARG(L.n_in == 2);
            const fm_tensor& tb = net->tensors[L.in[1]];
            FM_CHECK_ARG(tb.h == ti.h && tb.w == ti.w && to.h == ti.h && to.w == ti.w);
            return launch_add(in0, ti.c, L.in_coff[0], (const f16*)net->bufs[L.in[1]], tb.c, L.in_coff[1], out, to.c,
                              L.out_coff, (long)B * ti.h * ti.w, L.cin, s);
        }
        case FM_OP_COPY:
            return launch_copy(in0, ti.c, L.in_coff[0], out, to.c, L.out_coff, (long)B * ti.h * ti.w, L.cin, s);
        case FM_OP_GATE:
            FM_CHECK_ARG(L.gate[0] >= 0 && L.gate[0] < net->n_gates && L.cin <= net->gate_c);
            return launch_gate(in0, ti.c, L.in_coff[0], B, ti.h * ti.w, L.cin, L.hid,
                               (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off),
                               (const f16*)(net->weights + L.w2_off), (const float*)(net->weights + L.b2_off),
                               net->gates + (size_t)L.gate[0] * net->max_batch * net->gate_c * GATE_SLOT_TILES, s);
        case FM_OP_GATE_SUM: {
            const f16* ins[4];
            int cs[4], co[4];
            const float* gs[4];
            for (int i = 0; i < L.n_in; ++i) {
                ins[i] = (const f16*)net->bufs[L.in[i]];
                cs[i] = net->tensors[L.in[i]].c;
                co[i] = L.in_coff[i];
                FM_CHECK_ARG(L.gate[i] >= 0 && L.gate[i] < net->n_gates);
                gs[i] = net->gates + (size_t)L.gate[i] * net->max_batch * net->gate_c * GATE_SLOT_TILES;
            }
            return launch_gate_sum(L.n_in, ins, cs, co, gs, out, to.c, L.out_coff, B, ti.h * ti.w, L.cin, s);
        }
        case FM_OP_HEAD:
            FM_CHECK_ARG(L.cout == ctx->feat_dim && net->emb_offset + B <= ctx->emb_cap);
            return launch_head(in0, ti.c, L.in_coff[0], B, ti.h * ti.w, L.cin, L.cout,
                               (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off),
                               ctx->emb + (size_t)net->emb_offset * ctx->feat_dim, nullptr, s,
                               ctx->emb_host ? ctx->emb_host + (size_t)net->emb_offset * ctx->feat_dim : nullptr);
        case FM_OP_OSTAIL:
            FM_CHECK_ARG(L.cout == ctx->feat_dim && net->emb_offset + B <= ctx->emb_cap && L.in_coff[0] + L.cin <= ti.c &&
                         ostail_supported(ti.h, ti.w, L.cin, L.hid, L.k, L.cout, L.stride, L.pad));
            return launch_ostail(in0, ti.c, L.in_coff[0], (const f16*)(net->weights + L.w_off),
                                 (const float*)(net->weights + L.b_off), B, ctx->emb + (size_t)net->emb_offset * ctx->feat_dim,
                                 nullptr, ctx->emb_host ? ctx->emb_host + (size_t)net->emb_offset * ctx->feat_dim : nullptr, s);
        default:
            fm_set_error("unknown layer op %d", L.op);
            return FM_ERR_ARG;
    }
}

// gate kernels index gate[n*C + c] with C = the gated channel count; buffers are spaced by
// max_batch*gate_c so any C <= gate_c fits.
static int run_layers_eager(fm_ctx* ctx, NetState* net, int batch) {
    for (size_t i = (size_t)net->first; i < net->layers.size(); ++i) {
        int rc = run_layer(ctx, net, net->layers[i], batch);
        if (rc) return rc;
    }
    return 0;
}

// Layer 0 is a stem convolution over the network's input tensor (and nothing else reads that tensor): the detector /
// extractor front ends can then let the stem compute its input pixels from the frame itself (pixel_source.h).
bool fm_net_stem_fusable(const NetState* net, int input_tensor) {
    if (!net || net->layers.empty()) return false;
    const fm_layer& L = net->layers[0];
    if ((L.op != FM_OP_STEMCONV && L.op != FM_OP_STEM2) || L.n_in != 1 || L.in[0] != input_tensor || L.in_coff[0] != 0) return false;
    for (size_t i = 1; i < net->layers.size(); ++i) {
        const fm_layer& M = net->layers[i];
        for (int k = 0; k < M.n_in; ++k)
            if (M.in[k] == input_tensor) return false;
        if (M.res_mode != FM_RES_NONE && M.res == input_tensor) return false;
        if (M.out == input_tensor) return false;
    }
    return true;
}

int launch_stem2_layer(const fm_layer& L, const StemSrc& src, const NetState* net, int batch, hipStream_t s) {
    const fm_tensor& ti = net->tensors[L.in[0]];
    const fm_tensor& to = net->tensors[L.out];
    const bool three = L.gate[1] > 0;
    const int cout2 = three ? L.gate[1] : L.cout, act2 = three ? L.gate[2] : L.act;
    FM_CHECK_ARG(!to.f32 && L.hid == 32 && stem2_supported(L.hid, cout2) && L.in_coff[0] == 0 && L.out_coff + L.cout <= to.c);
    FM_CHECK_ARG(!three || stem3_supported(cout2, L.cout));
    const char* w2 = net->weights + L.w2_off;
    const char* b2 = net->weights + L.b2_off;
    return launch_stem2(src, (const f16*)net->bufs[L.in[0]], ti.c, (f16*)net->bufs[L.out], to.c, L.out_coff,
                        (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off), (const f16*)w2,
                        (const float*)b2, batch, ti.h, ti.w, to.h / 1, to.w / 1, cout2, L.gate[0], act2, s, three ? L.cout : 0,
                        three ? (const f16*)(w2 + (size_t)cout2 * 288 * 2) : nullptr,
                        three ? (const float*)(b2 + (size_t)cout2 * 4) : nullptr, L.act);
}

// Layer 0 of `net` on `src` instead of the input tensor, launched eagerly on the network's stream (its arguments -- the
// frame, the boxes -- change from call to call: not part of the captured graph); fm_net_run then starts at layer 1
// (net->first, which the caller sets around its fm_net_run call).
int fm_net_run_stem_from(fm_ctx* ctx, NetState* net, const StemSrc& src, int batch) {
    FM_CHECK_ARG(ctx && net && !net->layers.empty() && batch >= 1 && batch <= net->max_batch);
    const fm_layer& L = net->layers[0];
    FM_CHECK_ARG(L.op == FM_OP_STEMCONV || L.op == FM_OP_STEM2);
    const fm_tensor& ti = net->tensors[L.in[0]];
    const fm_tensor& to = net->tensors[L.out];
    if (L.op == FM_OP_STEM2) {
        FM_CHECK_ARG(src.kind != 2);
        return launch_stem2_layer(L, src, net, batch, net->stream);
    }
    return launch_stemconv_src(src, (const f16*)net->bufs[L.in[0]], ti.c, 0, (f16*)net->bufs[L.out], to.c, L.out_coff,
                               (const f16*)(net->weights + L.w_off), (const float*)(net->weights + L.b_off), batch, ti.h,
                               ti.w, to.h, to.w, L.k, L.stride, L.pad, L.cout, L.act, net->stream);
}

// The layer sequence of one (batch, embedding offset) is captured once into a hipGraph and
// replayed afterwards: ~160 kernel launches cost ~0.5 ms of host time per frame when issued
// one by one, which sits on the (serial) host critical path of MOT.step.
extern "C" int fm_net_run(fm_ctx* ctx, int which, int batch) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS));
    NetState* net = net_slot(ctx, which);
    FM_CHECK_ARG(net != nullptr && batch >= 0 && batch <= net->max_batch);
    if (batch == 0) return 0;
    if (!net->use_graphs || !ctx->opt_use_graphs) return run_layers_eager(ctx, net, batch);
    const long key = (((long)batch << 32) | (unsigned)net->emb_offset) ^ ((long)net->first << 60);
    for (auto& g : net->graphs)
        if (g.first == key) {
            FM_HIP(hipGraphLaunch(g.second, net->stream));
            return 0;
        }
    // first use: validate eagerly once (argument errors surface here), then capture
    int rc = run_layers_eager(ctx, net, batch);
    if (rc) return rc;
    FM_HIP(hipStreamSynchronize(net->stream));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(net->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        net->use_graphs = false;
        return 0;
    }
    rc = run_layers_eager(ctx, net, batch);
    const hipError_t ec = hipStreamEndCapture(net->stream, &graph);
    if (rc != 0 || ec != hipSuccess || graph == nullptr ||
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        net->use_graphs = false;   // stay eager (the eager run above already produced this step's result)
        return rc;
    }
    (void)hipGraphDestroy(graph);
    net->graphs.emplace_back(key, exec);
    return 0;   // this call's work was done by the eager validation run
}

extern "C" int fm_net_tensor_write(fm_ctx* ctx, int which, int tensor, const void* host, size_t bytes) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS) && host);
    NetState* net = net_slot(ctx, which);
    FM_CHECK_ARG(net && tensor >= 0 && tensor < (int)net->tensors.size());
    const fm_tensor& t = net->tensors[tensor];
    FM_CHECK_ARG(bytes <= (size_t)net->max_batch * t.h * t.w * t.c * elem_size(t));
    FM_HIP(hipStreamSynchronize(net->stream));
    FM_HIP(hipMemcpy(net->bufs[tensor], host, bytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int fm_net_tensor_read(fm_ctx* ctx, int which, int tensor, void* host, size_t bytes) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS) && host);
    NetState* net = net_slot(ctx, which);
    FM_CHECK_ARG(net && tensor >= 0 && tensor < (int)net->tensors.size());
    const fm_tensor& t = net->tensors[tensor];
    FM_CHECK_ARG(bytes <= (size_t)net->max_batch * t.h * t.w * t.c * elem_size(t));
    FM_HIP(hipStreamSynchronize(net->stream));
    FM_HIP(hipMemcpy(host, net->bufs[tensor], bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int fm_net_read_embeddings(fm_ctx* ctx, int n, float* host) {
    FM_CHECK_ARG(ctx && n >= 0 && n <= ctx->emb_cap);
    if (n == 0) return 0;
    FM_CHECK_ARG(host);
    FM_HIP(hipStreamSynchronize(ctx->s_ext));
    FM_HIP(hipMemcpy(host, ctx->emb, sizeof(float) * (size_t)n * ctx->feat_dim, hipMemcpyDeviceToHost));
    ctx->emb_n = n;
    return 0;
}

static void layer_cost(const NetState* net, const fm_layer& L, int B, double* flops, double* bytes) {
    const fm_tensor& ti = net->tensors[L.in[0]];
    const fm_tensor& to = net->tensors[L.out];
    const double pin = (double)B * ti.h * ti.w, pout = (double)B * to.h * to.w;
    *flops = 0;
    *bytes = 0;
    switch (L.op) {
        case FM_OP_CONV:
        case FM_OP_CONVS:
        case FM_OP_CONVD:
            *flops = 2.0 * L.k * L.k * L.cin * L.cout * pout;
            *bytes = pin * L.cin * 2 + pout * L.cout * (to.f32 ? 4 : 2) + (double)L.k * L.k * L.cin * L.cout * 2 +
                     (L.res_mode != FM_RES_NONE ? pout * L.cout * 2 : 0);
            break;
        case FM_OP_DWCONV3:
            *flops = 2.0 * 9 * L.cin * pout;
            *bytes = (pin + pout) * L.cin * 2;
            break;
        case FM_OP_LITECONV:
            *flops = 2.0 * (L.cin + 9) * L.cout * pout * L.n_in;
            *bytes = ((pin + pout) * L.cin * 2 + (double)L.cin * L.cout * 2) * L.n_in;
            break;
        case FM_OP_LITECHAIN:
            *flops = 2.0 * (L.cin + 9) * L.cout * pout * 10;
            *bytes = (pin + 4 * pout) * L.cin * 2 + 10.0 * L.cin * L.cout * 2;
            break;
        case FM_OP_GATED_SUM: *bytes = (pin * L.n_in + pout) * L.cin * 2; break;
        case FM_OP_STEMCONV:
            *flops = 2.0 * L.k * L.k * 3 * L.cout * pout;
            *bytes = pin * 8 + pout * L.cout * 2;
            break;
        case FM_OP_PAIR11:      // the two layers it replaces (the network's algorithmic work does not change with a fusion)
            *flops = 2.0 * L.cin * L.hid * pout + 2.0 * (L.hid + L.cin) * L.cout * pout;
            *bytes = (pin * L.cin + pout * L.hid + (double)L.cin * L.hid) * 2 + (pout * (L.hid + L.cin) + pout * L.cout + (double)(L.hid + L.cin) * L.cout) * 2;
            break;
        case FM_OP_STEM2:       // both convs' FLOPs; the bytes of the fused pair: input in, second conv's output out, weights
            *flops = 2.0 * 9 * 3 * L.hid * pin + 2.0 * 9 * L.hid * (L.gate[1] > 0 ? L.gate[1] : L.cout) * pout +
                     (L.gate[1] > 0 ? 2.0 * L.gate[1] * L.cout * pout : 0.);
            *bytes = pin * 8 + pout * L.cout * 2 + 9.0 * L.hid * (L.gate[1] > 0 ? L.gate[1] : L.cout) * 2;
            break;
        case FM_OP_RESBLOCK:
            *flops = 2.0 * 10 * L.cin * L.hid * pout;
            *bytes = (pin + pout) * L.cin * 2 + 10.0 * L.cin * L.hid * 2;
            break;
        case FM_OP_SPP: *bytes = (pin + 3 * pout) * L.cin * 2; break;
        case FM_OP_GATE: *bytes = pin * L.cin * 2; break;
        case FM_OP_GATE_SUM: *bytes = (pin * L.n_in + pout) * L.cin * 2; break;
        case FM_OP_OSTAIL: {    // the eleven layers it replaces: 128 pixels per sample, cin -> (hid | k) -> k -> k, head k -> cout
            const double px = 128.0 * B, m = L.hid, c = L.k, lite = 10.0 * (m + 9) * m;
            *flops = 2.0 * px * (L.cin * m + lite + (m + L.cin) * c + c * m + lite + m * c + c * c) + 2.0 * c * L.cout * B;
            *bytes = pin * L.cin * 2 + ((double)L.stride) * 2 + (double)B * L.cout * 4;
            break;
        }
        case FM_OP_HEAD:
            *flops = 2.0 * L.cin * L.cout * B;
            *bytes = pin * L.cin * 2 + (double)L.cin * L.cout * 2;
            break;
        default: *bytes = (pin + pout) * L.cin * 2; break;
    }
}

extern "C" int fm_net_cost(fm_ctx* ctx, int which, int batch, double* flops, double* bytes) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS) && flops && bytes);
    NetState* net = net_slot(ctx, which);
    FM_CHECK_ARG(net != nullptr);
    double f = 0, b = 0;
    for (const fm_layer& L : net->layers)
        if (L.op == FM_OP_CONV || L.op == FM_OP_CONVS || L.op == FM_OP_CONVD || L.op == FM_OP_RESBLOCK || L.op == FM_OP_PAIR11) {
            double lf, lb;
            layer_cost(net, L, batch, &lf, &lb);
            f += lf;
            b += lb;
        } else if (L.op == FM_OP_STEM2) {
            // the pair's second conv as the stand-alone layer it replaces (the stem itself was never part of this sum): the
            // algorithmic work of a network does not change with the fusions of its layer table
            const fm_tensor& ti = net->tensors[L.in[0]];
            const fm_tensor& to = net->tensors[L.out];
            const double pmid = (double)batch * ti.h * ti.w, pout = (double)batch * to.h * to.w;
            const int c2 = L.gate[1] > 0 ? L.gate[1] : L.cout;
            f += 2.0 * 9 * L.hid * c2 * pout;
            b += pmid * L.hid * 2 + pout * c2 * 2 + 9.0 * L.hid * c2 * 2;
            if (L.gate[1] > 0) {        // ... and the pointwise conv behind it
                f += 2.0 * c2 * L.cout * pout;
                b += pout * c2 * 2 + pout * L.cout * 2 + (double)c2 * L.cout * 2;
            }
        }
    *flops = f;
    *bytes = b;
    return 0;
}

extern "C" int fm_net_profile(fm_ctx* ctx, int which, int batch, int iters, double* conv_ms,
                              double* other_ms, int* n_conv, int* n_other) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS) && iters > 0);
    NetState* net = net_slot(ctx, which);
    FM_CHECK_ARG(net != nullptr && batch > 0 && batch <= net->max_batch);
    hipEvent_t e0, e1;
    FM_HIP(hipEventCreate(&e0));
    FM_HIP(hipEventCreate(&e1));
    double tc = 0, to = 0;
    int nc = 0, no = 0;
    for (int it = 0; it < iters; ++it)
        for (const fm_layer& L : net->layers) {
            FM_HIP(hipEventRecord(e0, net->stream));
            int rc = run_layer(ctx, net, L, batch);
            if (rc) return rc;
            FM_HIP(hipEventRecord(e1, net->stream));
            FM_HIP(hipEventSynchronize(e1));
            float ms = 0;
            FM_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (L.op == FM_OP_CONV || L.op == FM_OP_CONVS || L.op == FM_OP_CONVD || L.op == FM_OP_RESBLOCK || L.op == FM_OP_STEM2 || L.op == FM_OP_PAIR11) { tc += ms; ++nc; } else { to += ms; ++no; }
        }
    FM_HIP(hipEventDestroy(e0));
    FM_HIP(hipEventDestroy(e1));
    if (conv_ms) *conv_ms = tc / iters;
    if (other_ms) *other_ms = to / iters;
    if (n_conv) *n_conv = nc / iters;
    if (n_other) *n_other = no / iters;
    return 0;
}

// per-layer HIP-event times (ms), averaged over iters; out[n_layers]
extern "C" int fm_net_profile_layers(fm_ctx* ctx, int which, int batch, int iters, double* out) {
    FM_CHECK_ARG(ctx && (which >= 0 && which < FM_NET_EXTRACTOR_B + FM_MAX_EXTRA_EXTRACTORS) && iters > 0 && out);
    NetState* net = net_slot(ctx, which);
    FM_CHECK_ARG(net != nullptr && batch > 0 && batch <= net->max_batch);
    hipEvent_t e0, e1;
    FM_HIP(hipEventCreate(&e0));
    FM_HIP(hipEventCreate(&e1));
    const size_t n = net->layers.size();
    for (size_t i = 0; i < n; ++i) out[i] = 0;
    for (int it = 0; it < iters; ++it)
        for (size_t i = 0; i < n; ++i) {
            FM_HIP(hipEventRecord(e0, net->stream));
            int rc = run_layer(ctx, net, net->layers[i], batch);
            if (rc) return rc;
            FM_HIP(hipEventRecord(e1, net->stream));
            FM_HIP(hipEventSynchronize(e1));
            float ms = 0;
            FM_HIP(hipEventElapsedTime(&ms, e0, e1));
            out[i] += ms / iters;
        }
    FM_HIP(hipEventDestroy(e0));
    FM_HIP(hipEventDestroy(e1));
    return 0;
}

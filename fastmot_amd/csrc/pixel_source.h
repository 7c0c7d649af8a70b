// Where a network's first layer can take its input pixels from (round 6): the standalone front-end kernels
// (preprocess_kernel in detect.hip, crop_resize_kernel in extract.hip) and the stem convolution that computes the
// same pixels on the fly while it stages its LDS patch (stemconv.hip, StemSrc) share these functions, so that the
// fused path is bit-identical to kernel + tensor + stem by construction.
#pragma once
#include "net.h"

// YOLODetector._preprocess (fastmot/detector.py:289-300): bilinear resize in uint8 with half-pixel centres and edge
// clamp (cupyx zoom mode='opencv', grid_mode=True: src = (dst + 0.5) * (in/out) - 0.5; affine_transform order=1,
// mode='nearest'), rounded to uint8 (rint), BGR -> RGB, * 1/255 (fp32), fp16.  Outside the letterbox ROI: 0.5.
// (x, y): pixel of the network input.  Returns the three channels as floats already rounded through u8.
// two neighbouring BGR pixels = 6 consecutive bytes as ONE (unaligned) 8-byte load instead of six byte loads: the resize
// functions below are bound by their dependent trips to memory (round 6: the stem computing its own input took 31 us inside
// the pipeline).  Reads up to 7 bytes past a pixel: frame buffers are allocated with FM_FRAME_SLACK spare bytes.
#define FM_FRAME_SLACK 16
typedef uint64_t fm_u64_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ uint64_t load_px2(const uint8_t* p) { return *reinterpret_cast<const fm_u64_unaligned*>(p); }

__device__ __forceinline__ void det_input_pixel(const uint8_t* __restrict__ frame, int fw, int fh, int x, int y,
                                                int roi_x, int roi_y, int roi_w, int roi_h, float rgb[3]) {
    // (branch-free: positions outside the letterbox ROI compute a clamped one and select 0.5 at the end, so that a caller's
    // unrolled loop can have all its loads in flight at once)
    const int rx0 = x - roi_x, ry0 = y - roi_y;
    const bool in_roi = !(rx0 < 0 || ry0 < 0 || rx0 >= roi_w || ry0 >= roi_h);
    const int rx = min(max(rx0, 0), roi_w - 1), ry = min(max(ry0, 0), roi_h - 1);
    const double zy = (double)fh / roi_h, zx = (double)fw / roi_w;
    const double sy = ry * zy + (zy - 1.) / 2., sx = rx * zx + (zx - 1.) / 2.;
    const double fy = floor(sy), fx = floor(sx);
    const double wy = sy - fy, wx = sx - fx;
    int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    y0 = min(max(y0, 0), fh - 1); y1 = min(max(y1, 0), fh - 1);
    x0 = min(max(x0, 0), fw - 1); x1 = min(max(x1, 0), fw - 1);
    // (after the clamps x1 is x0 + 1 or x0)
    const uint64_t q0 = load_px2(frame + ((size_t)y0 * fw + x0) * 3), q1 = load_px2(frame + ((size_t)y1 * fw + x0) * 3);
    const int sh = x1 == x0 ? 0 : 24;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int p00 = (int)((q0 >> (8 * c)) & 255), p01 = (int)((q0 >> (sh + 8 * c)) & 255);
        const int p10 = (int)((q1 >> (8 * c)) & 255), p11 = (int)((q1 >> (sh + 8 * c)) & 255);
        const double top = (1. - wx) * p00 + wx * p01;
        const double bot = (1. - wx) * p10 + wx * p11;
        const double v = rint((1. - wy) * top + wy * bot);
        const double u8 = fmin(fmax(v, 0.), 255.);
        rgb[2 - c] = in_roi ? (float)(u8 * (1. / 255.)) : 0.5f;      // BGR -> RGB
    }
}

struct ResizeCoef { int s; short a0, a1; };

// OpenCV resize coordinate + coefficient computation for one output index (imgproc/resize.cpp, INTER_LINEAR, 8-bit)
__device__ __forceinline__ ResizeCoef lin_coef(int d, double scale, int ssize) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    ResizeCoef c;
    c.s = s;
    // saturate_cast<short>(v * 2048) with cvRound (round half to even)
    c.a0 = (short)__float2int_rn((1.f - f) * 2048.f);
    c.a1 = (short)__float2int_rn(f * 2048.f);
    return c;
}

// FeatureExtractor._preprocess / _normalize (fastmot/feature_extractor.py:84-98) + multi_crop (utils/rect.py:93-97):
// pixel (x, y) of the ow x oh network input of the crop `bx` (tlbr, doubles): crop (astype(int) truncation,
// maximum(., 0), inclusive bottom-right, numpy slice clamp) -> cv2.resize INTER_LINEAR (an exact 2x decimation in both
// axes goes to INTER_AREA: rounded 2x2 mean) -> BGR -> RGB -> (v / 255 - mean) / std.  o: 8 halfs, channels 3..7 zero.
// the resized crop's pixel as uint8 BGR (bgr[c]); false when the crop is empty (the network input is zero then)
__device__ __forceinline__ bool crop_u8_pixel(const uint8_t* __restrict__ frame, int fw, int fh,
                                              const double* __restrict__ bx, int x, int y, int ow, int oh, int bgr[3]) {
    int x1 = max((int)bx[0], 0), y1 = max((int)bx[1], 0);
    int x2 = max((int)bx[2], 0), y2 = max((int)bx[3], 0);
    x2 = min(x2 + 1, fw); y2 = min(y2 + 1, fh);
    const int cw = x2 - x1, ch = y2 - y1;
    if (!(cw > 0 && ch > 0)) return false;
    const ResizeCoef cx = lin_coef(x, (double)cw / ow, cw);
    const ResizeCoef cy = lin_coef(y, (double)ch / oh, ch);
    const int sx1 = min(cx.s + 1, cw - 1), sy1 = min(cy.s + 1, ch - 1);
    const bool area2 = cw == 2 * ow && ch == 2 * oh;
    // both branches read two neighbouring pixels of two rows: one 8-byte load per row (load_px2)
    const int ax = area2 ? 2 * x : cx.s, ay0 = area2 ? 2 * y : cy.s, ay1 = area2 ? 2 * y + 1 : sy1;
    const uint64_t q0 = load_px2(frame + ((size_t)(y1 + ay0) * fw + x1 + ax) * 3);
    const uint64_t q1 = load_px2(frame + ((size_t)(y1 + ay1) * fw + x1 + ax) * 3);
    const int sh = (area2 || sx1 != cx.s) ? 24 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int a0 = (int)((q0 >> (8 * c)) & 255), a1 = (int)((q0 >> (sh + 8 * c)) & 255);
        const int b0 = (int)((q1 >> (8 * c)) & 255), b1 = (int)((q1 >> (sh + 8 * c)) & 255);
        if (area2) {
            bgr[c] = (a0 + a1 + b0 + b1 + 2) >> 2;
        } else {
            const int S0 = a0 * cx.a0 + a1 * cx.a1;
            const int S1 = b0 * cx.a0 + b1 * cx.a1;
            const int v = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
            bgr[c] = min(max(v, 0), 255);
        }
    }
    return true;
}

// (v / 255 - mean) / std of RGB channel rc for a uint8 value (feature_extractor.py:88-98), rounded to fp16 through float.
// Two float64 divisions: the fused stem tabulates the 3 x 256 values once per workgroup instead of dividing per pixel.
__device__ __forceinline__ f16 crop_normalise(int u8, int rc) {
    const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
    return (f16)(float)(((double)u8 / 255. - mean[rc]) / stdv[rc]);
}

__device__ __forceinline__ void crop_input_pixel(const uint8_t* __restrict__ frame, int fw, int fh,
                                                 const double* __restrict__ bx, int x, int y, int ow, int oh, f16x8& o) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)0.f;
    int bgr[3];
    if (crop_u8_pixel(frame, fw, fh, bx, x, y, ow, oh, bgr)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[2 - c] = crop_normalise(bgr[c], 2 - c);      // BGR -> RGB
    }
}

// Input of a stem convolution (stemconv.hip).  kind 0: the fp16 NHWC tensor; 1: the detector's letterboxed resize of
// the frame; 2: the extractor's crops of the frame.
struct StemSrc {
    int kind;
    const uint8_t* frame;
    int fw, fh;
    int roi_x, roi_y, roi_w, roi_h;   // kind 1
    const double* boxes;              // kind 2: [N][4] tlbr on the device
    int32_t* zero4;                   // optional: four int32 the launch sets to zero (the decode's candidate counters)
};

int launch_stemconv_src(const StemSrc& src, const f16* in, int in_cs, int in_coff, f16* out, int out_cs, int out_coff,
                        const f16* w, const float* bias, int N, int H, int W, int Ho, int Wo, int k, int stride, int pad,
                        int cout, int act, hipStream_t s);
int launch_stem2(const StemSrc& src, const f16* in, int in_cs, f16* out, int out_cs, int out_coff, const f16* w1,
                 const float* b1, const f16* w2, const float* b2, int N, int H, int W, int Ho, int Wo, int cout, int act1,
                 int act2, hipStream_t s, int cout3 = 0, const f16* w3 = nullptr, const float* b3 = nullptr, int act3 = 0);
bool stem2_supported(int mid, int cout);
bool stem3_supported(int cout2, int cout3);
// FM_OP_STEM2 of a layer table: two stages (gate[1] < 0: cout / act are the second conv's) or three (gate[1] = the second
// conv's channels, gate[2] its activation, cout / act the pointwise conv's; its weights / bias lie behind the second conv's)
int launch_stem2_layer(const fm_layer& L, const StemSrc& src, const NetState* net, int batch, hipStream_t s);
// runs layer 0 of `net` -- a stem convolution over the network's input tensor -- on `src` instead of that tensor
bool fm_net_stem_fusable(const NetState* net, int input_tensor);
int fm_net_run_stem_from(fm_ctx* ctx, NetState* net, const StemSrc& src, int batch);

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
__device__ __forceinline__ void det_input_pixel(const uint8_t* __restrict__ frame, int fw, int fh, int x, int y,
                                                int roi_x, int roi_y, int roi_w, int roi_h, float rgb[3]) {
    const int rx = x - roi_x, ry = y - roi_y;
    if (rx < 0 || ry < 0 || rx >= roi_w || ry >= roi_h) {
        rgb[0] = rgb[1] = rgb[2] = 0.5f;
        return;
    }
    const double zy = (double)fh / roi_h, zx = (double)fw / roi_w;
    const double sy = ry * zy + (zy - 1.) / 2., sx = rx * zx + (zx - 1.) / 2.;
    const double fy = floor(sy), fx = floor(sx);
    const double wy = sy - fy, wx = sx - fx;
    int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    y0 = min(max(y0, 0), fh - 1); y1 = min(max(y1, 0), fh - 1);
    x0 = min(max(x0, 0), fw - 1); x1 = min(max(x1, 0), fw - 1);
    const uint8_t* p00 = frame + ((size_t)y0 * fw + x0) * 3;
    const uint8_t* p01 = frame + ((size_t)y0 * fw + x1) * 3;
    const uint8_t* p10 = frame + ((size_t)y1 * fw + x0) * 3;
    const uint8_t* p11 = frame + ((size_t)y1 * fw + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double top = (1. - wx) * p00[c] + wx * p01[c];
        const double bot = (1. - wx) * p10[c] + wx * p11[c];
        const double v = rint((1. - wy) * top + wy * bot);
        const double u8 = fmin(fmax(v, 0.), 255.);
        rgb[2 - c] = (float)(u8 * (1. / 255.));      // BGR -> RGB
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
__device__ __forceinline__ void crop_input_pixel(const uint8_t* __restrict__ frame, int fw, int fh,
                                                 const double* __restrict__ bx, int x, int y, int ow, int oh, f16x8& o) {
    int x1 = max((int)bx[0], 0), y1 = max((int)bx[1], 0);
    int x2 = max((int)bx[2], 0), y2 = max((int)bx[3], 0);
    x2 = min(x2 + 1, fw); y2 = min(y2 + 1, fh);
    const int cw = x2 - x1, ch = y2 - y1;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)0.f;
    if (cw > 0 && ch > 0) {
        const ResizeCoef cx = lin_coef(x, (double)cw / ow, cw);
        const ResizeCoef cy = lin_coef(y, (double)ch / oh, ch);
        const int sx1 = min(cx.s + 1, cw - 1), sy1 = min(cy.s + 1, ch - 1);
        const uint8_t* r0 = frame + ((size_t)(y1 + cy.s) * fw + x1) * 3;
        const uint8_t* r1 = frame + ((size_t)(y1 + sy1) * fw + x1) * 3;
        const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
        const bool area2 = cw == 2 * ow && ch == 2 * oh;
        const uint8_t* q0 = frame + ((size_t)(y1 + 2 * y) * fw + x1 + 2 * x) * 3;
        const uint8_t* q1 = q0 + (size_t)fw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int u8;
            if (area2) {
                u8 = (q0[c] + q0[3 + c] + q1[c] + q1[3 + c] + 2) >> 2;
            } else {
                const int S0 = r0[cx.s * 3 + c] * cx.a0 + r0[sx1 * 3 + c] * cx.a1;
                const int S1 = r1[cx.s * 3 + c] * cx.a0 + r1[sx1 * 3 + c] * cx.a1;
                const int v = (((cy.a0 * (S0 >> 4)) >> 16) + ((cy.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
                u8 = min(max(v, 0), 255);
            }
            const int rc = 2 - c;    // BGR -> RGB
            o[rc] = (f16)(float)(((double)u8 / 255. - mean[rc]) / stdv[rc]);
        }
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
// runs layer 0 of `net` -- a stem convolution over the network's input tensor -- on `src` instead of that tensor
bool fm_net_stem_fusable(const NetState* net, int input_tensor);
int fm_net_run_stem_from(fm_ctx* ctx, NetState* net, const StemSrc& src, int batch);

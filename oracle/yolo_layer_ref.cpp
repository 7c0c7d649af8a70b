// TEST INFRASTRUCTURE ONLY -- host build of the reference's YOLO decode kernels.
//
// fastmot/plugins/yolo_layer.cu holds two CUDA kernels, CalDetection and CalDetection_NewCoords (one thread per grid cell
// and anchor).  Their bodies are plain C apart from the thread index and __expf; oracle/yolo_layer_ref.py (the recipe)
// cuts the device functions out of the reference file WHERE IT LIES (from "inline __device__ float sigmoidGPU" to the
// line before "void YoloLayerPlugin::forwardGpu") into oracle/_ref/yolo_layer_kernels.inc -- a build product, git-ignored,
// never committed -- and this file compiles them for the host behind the shims below and runs the "grid" as a loop.
// __expf, CUDA's fast exponential, becomes expf: the index arithmetic, the output layout, the class arg-max and the
// box formulas are the reference's own statements; only the last bits of exp() are the host library's.
#include <cmath>
#include <cstring>
#include <limits>

#define __global__
#define __device__
#define CUDART_INF_F std::numeric_limits<float>::infinity()
#define __expf(x) expf(x)      // (glibc declares a __expf of its own: a macro, not a function)
struct Dim3 { int x; };
static thread_local Dim3 threadIdx, blockDim, blockIdx;

namespace Yolo {
struct alignas(float) Detection {          // fastmot/plugins/yolo_layer.h:34-39
    float bbox[4];
    float det_confidence;
    float class_id;
    float class_confidence;
};
}
using namespace Yolo;

namespace nvinfer1 {
#include "_ref/yolo_layer_kernels.inc"
}

extern "C" void ref_yolo_decode(const float* input, float* output, int yolo_width, int yolo_height, int num_anchors,
                                const float* anchors, int num_classes, int input_w, int input_h, float scale_x_y,
                                int new_coords) {
    const int n = yolo_width * yolo_height * num_anchors;
    blockDim.x = 64;
    for (int idx = 0; idx < ((n + 63) / 64) * 64; ++idx) {       // (whole blocks: the kernels guard idx themselves)
        blockIdx.x = idx / 64;
        threadIdx.x = idx % 64;
        if (new_coords)
            nvinfer1::CalDetection_NewCoords(input, output, 1, yolo_width, yolo_height, num_anchors, anchors, num_classes,
                                             input_w, input_h, scale_x_y);
        else
            nvinfer1::CalDetection(input, output, 1, yolo_width, yolo_height, num_anchors, anchors, num_classes, input_w,
                                   input_h, scale_x_y);
    }
}

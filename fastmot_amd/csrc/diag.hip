// Diagnostic kernels (no counterpart in the reference).
//
// fm_diag_pkhaz: stand-alone reproducer of what round 3's bisect of the LK kernel found (DESIGN 5b,
// profiles/r03_lk_bisect.txt): the position update of the LK iteration,
//     dx = (A12 * b2 - A22 * b1) * Dt,   dy = (A12 * b1 - A11 * b2) * Dt,
// is compiled (SLP vectoriser) into packed-fp32 VALU instructions,
//     v_pk_mul_f32 P, A12A12, B                     ; P  = (A12 * b2, A12 * b1)
//     v_pk_mul_f32 B, A22A11, B op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]   ; B = (-A22 * b1, -A11 * b2), IN PLACE, halves swapped
//     v_pk_add_f32 B, P, B
//     v_pk_mul_f32 D, DtDt, B
// and while VALU-heavy wavefronts of another kernel share the CU, lanes 48..63 of the wavefront now and then get a
// wrong LOW half of D (dx); every other lane and the high half are right.  Every wavefront here runs that chain on
// wave-uniform operands and counts the lanes whose result differs from lane 0's.
// variant 0: the chain as compiled; 1: the swapped multiply writes a fresh register pair; 2: one more wait state
// (s_nop) between the two multiplies; 3: s_nop 3 between every pair; 4: plain (unpacked) v_mul / v_fma arithmetic.
#ifdef FM_DIAG      // compiled to an empty object in the shipped library (build with FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_DIAG)
#include "common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(256) void pkhaz_kernel(const float* __restrict__ in, int iters, int* __restrict__ out) {
    const int g = threadIdx.x & 63;
    // wave-uniform operands, one copy per lane (like the LK kernel's A11 / A12 / A22 / Dt after the broadcasts)
    const float A11 = in[0], A12 = in[1], A22 = in[2], Dt = in[3];
    float b1 = in[4], b2 = in[5];
    const v2f a12a12 = {A12, A12}, a22a11 = {A22, A11}, dtdt = {Dt, Dt};
    int bad_lo = 0, bad_hi = 0;
    float accx = 0.f, accy = 0.f;
    for (int it = 0; it < iters; ++it) {
        v2f b = {b2, b1}, p, d;
        if (VARIANT == 0) {
            asm volatile(
                "v_pk_mul_f32 %[p], %[a12], %[b]\n\t"
                "v_pk_mul_f32 %[b], %[a22a11], %[b] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "s_nop 0\n\t"
                "v_pk_add_f32 %[b], %[p], %[b]\n\t"
                "s_nop 0\n\t"
                "v_pk_mul_f32 %[d], %[dt], %[b]\n\t"
                "s_nop 1"
                : [p] "=&v"(p), [b] "+v"(b), [d] "=&v"(d) : [a12] "v"(a12a12), [a22a11] "v"(a22a11), [dt] "v"(dtdt));
        } else if (VARIANT == 1) {
            v2f q;
            asm volatile(
                "v_pk_mul_f32 %[p], %[a12], %[b]\n\t"
                "v_pk_mul_f32 %[q], %[a22a11], %[b] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "s_nop 0\n\t"
                "v_pk_add_f32 %[q], %[p], %[q]\n\t"
                "s_nop 0\n\t"
                "v_pk_mul_f32 %[d], %[dt], %[q]\n\t"
                "s_nop 1"
                : [p] "=&v"(p), [q] "=&v"(q), [d] "=&v"(d) : [b] "v"(b), [a12] "v"(a12a12), [a22a11] "v"(a22a11), [dt] "v"(dtdt));
        } else if (VARIANT == 2) {
            asm volatile(
                "v_pk_mul_f32 %[p], %[a12], %[b]\n\t"
                "s_nop 0\n\t"
                "v_pk_mul_f32 %[b], %[a22a11], %[b] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "s_nop 0\n\t"
                "v_pk_add_f32 %[b], %[p], %[b]\n\t"
                "s_nop 0\n\t"
                "v_pk_mul_f32 %[d], %[dt], %[b]\n\t"
                "s_nop 1"
                : [p] "=&v"(p), [b] "+v"(b), [d] "=&v"(d) : [a12] "v"(a12a12), [a22a11] "v"(a22a11), [dt] "v"(dtdt));
        } else if (VARIANT == 3) {
            asm volatile(
                "v_pk_mul_f32 %[p], %[a12], %[b]\n\t"
                "s_nop 3\n\t"
                "v_pk_mul_f32 %[b], %[a22a11], %[b] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "s_nop 3\n\t"
                "v_pk_add_f32 %[b], %[p], %[b]\n\t"
                "s_nop 3\n\t"
                "v_pk_mul_f32 %[d], %[dt], %[b]\n\t"
                "s_nop 3"
                : [p] "=&v"(p), [b] "+v"(b), [d] "=&v"(d) : [a12] "v"(a12a12), [a22a11] "v"(a22a11), [dt] "v"(dtdt));
        } else {
            float t0, t1, dx, dy;
            asm volatile(
                "v_mul_f32 %[t0], %[A12], %[b2]\n\t"
                "v_mul_f32 %[t1], %[A12], %[b1]\n\t"
                "v_mul_f32 %[dx], %[A22], %[b1]\n\t"
                "v_mul_f32 %[dy], %[A11], %[b2]\n\t"
                "v_sub_f32 %[dx], %[t0], %[dx]\n\t"
                "v_sub_f32 %[dy], %[t1], %[dy]\n\t"
                "v_mul_f32 %[dx], %[Dt], %[dx]\n\t"
                "v_mul_f32 %[dy], %[Dt], %[dy]"
                : [t0] "=&v"(t0), [t1] "=&v"(t1), [dx] "=&v"(dx), [dy] "=&v"(dy)
                : [A12] "v"(A12), [A22] "v"(A22), [A11] "v"(A11), [Dt] "v"(Dt), [b1] "v"(b1), [b2] "v"(b2));
            d.x = dx; d.y = dy;
        }
        const float fdx = d.x, fdy = d.y;        // (plain floats: __builtin_bit_cast of a vector ELEMENT reads element 0)
        const int dxb = __builtin_bit_cast(int, fdx), dyb = __builtin_bit_cast(int, fdy);
        bad_lo += dxb != __builtin_amdgcn_readfirstlane(dxb);
        bad_hi += dyb != __builtin_amdgcn_readfirstlane(dyb);
        // next operands: uniform, data dependent (from lane 0's results so that a wrong lane does not feed itself)
        const float ux = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(dxb));
        const float uy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(dyb));
        accx += ux; accy += uy;
        b1 = in[4] + 1e-3f * (float)(it & 255) + 1e-4f * ux;
        b2 = in[5] - 2e-3f * (float)(it & 127) + 1e-4f * uy;
    }
    if (bad_lo) atomicAdd(&out[g >> 4], bad_lo);
    if (bad_hi) atomicAdd(&out[4 + (g >> 4)], bad_hi);
    if (accx == 12345.678f && accy == 1.f) out[15] = 1;      // keeps the accumulators alive
}


// ---- which packed instruction is hit, and by which neighbour?  One packed instruction per evaluation on PER-LANE data,
// checked in the same lane against the same arithmetic done with unpacked instructions; beside a synthetic neighbour
// kernel that issues one instruction class in a loop.
template <int V>
__global__ __launch_bounds__(256) void pkvictim_kernel(const float* __restrict__ in, int iters, int* __restrict__ out) {
    const int g = threadIdx.x & 63;
    const float s0 = in[8 + g], s1 = in[72 + g], s2 = in[136 + g], s3 = in[200 + g];
    int bad_lo = 0, bad_hi = 0;
    float t = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float k = (float)(it & 63) * 0.03125f;
        const v2f a = {s0 + k, s1 - k}, b = {s2 - 0.5f * k, s3 + 0.25f * k}, c = {s1, s0};
        v2f d;
        float e0, e1;
        if (V == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b));
            e0 = a.x * b.x; e1 = a.y * b.y;
        } else if (V == 1) {
            asm volatile("v_pk_add_f32 %0, %1, %2\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b));
            e0 = a.x + b.x; e1 = a.y + b.y;
        } else if (V == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            e0 = __builtin_fmaf(a.x, b.x, c.x); e1 = __builtin_fmaf(a.y, b.y, c.y);
        } else if (V == 3) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b));
            e0 = a.x * b.y; e1 = a.y * b.x;
        } else if (V == 4) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b));
            e0 = a.x * -b.x; e1 = a.y * -b.y;
        } else {
            // two packed instructions back to back, the second consuming the first (as in the LK chain)
            v2f p;
            asm volatile("v_pk_mul_f32 %1, %2, %3\n\ts_nop 0\n\tv_pk_add_f32 %0, %1, %4\n\ts_nop 1" : "=&v"(d), "=&v"(p) : "v"(a), "v"(b), "v"(c));
            e0 = a.x * b.x + c.x; e1 = a.y * b.y + c.y;
        }
        const float d0 = d.x, d1 = d.y;
        bad_lo += __builtin_bit_cast(int, d0) != __builtin_bit_cast(int, e0);
        bad_hi += __builtin_bit_cast(int, d1) != __builtin_bit_cast(int, e1);
        t += e0 + e1;
    }
    if (bad_lo) atomicAdd(&out[g >> 4], bad_lo);
    if (bad_hi) atomicAdd(&out[4 + (g >> 4)], bad_hi);
    if (t == 12345.678f) out[15] = 1;
}

// neighbour kernels: `iters` x 16 instructions of one class on private registers
template <int A>
__global__ __launch_bounds__(256) void pkneighbour_kernel(const float* __restrict__ in, int iters, float* __restrict__ sink) {
    const int g = threadIdx.x & 63;
    float x0 = in[8 + g], x1 = in[72 + g], x2 = in[136 + g], x3 = in[200 + g];
    uint32_t h0 = __builtin_bit_cast(uint32_t, x0), h1 = __builtin_bit_cast(uint32_t, x1);
    v2f p0 = {x0, x1}, p1 = {x2, x3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (A == 0) {          // v_fma_mix_f32 taking the HIGH half of a packed fp16 pair (the LightConv kernels' inner loop)
                asm volatile("v_fma_mix_f32 %0, %2, %3, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mix_f32 %1, %4, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x0), "+v"(x1) : "v"(h0), "v"(x2), "v"(h1));
            } else if (A == 1) {   // ... the LOW half
                asm volatile("v_fma_mix_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mix_f32 %1, %4, %3, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x0), "+v"(x1) : "v"(h0), "v"(x2), "v"(h1));
            } else if (A == 2) {   // sdwa converts
                asm volatile("v_cvt_f32_f16_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                             "v_cvt_f32_f16_sdwa %1, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x0), "=v"(x1) : "v"(h0), "v"(h1));
            } else if (A == 3) {   // packed fp16
                asm volatile("v_pk_fma_f16 %0, %0, %2, %1\n\tv_pk_fma_f16 %1, %1, %2, %0" : "+v"(h0), "+v"(h1) : "v"(x2));
            } else if (A == 4) {   // packed fp32
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_mul_f32 %1, %1, %0" : "+v"(p0), "+v"(p1));
            } else if (A == 5) {   // plain fp32
                asm volatile("v_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %1, %1, %2, %0" : "+v"(x0), "+v"(x1) : "v"(x2));
            } else {               // plain fp32 with the VOP3 op_sel-free mix of mul / add
                asm volatile("v_mul_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %0" : "+v"(x0), "+v"(x1) : "v"(x2));
            }
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + p0.x + p0.y + p1.x + p1.y + __builtin_bit_cast(float, h0) + __builtin_bit_cast(float, h1);
}

}  // namespace

// Runs the reproducer on the KLT stream: `waves` wavefronts x `iters` evaluations of the update; out[0..3] = lanes with
// a wrong low half (dx) per quarter of the wavefront, out[4..7] = wrong high half (dy).
extern "C" int fm_diag_pkhaz(fm_ctx* ctx, int variant, int waves, int iters, int32_t* out8) {
    FM_CHECK_ARG(ctx && out8 && waves > 0 && iters > 0 && variant >= 0 && variant <= 4);
    static float* d_in = nullptr;
    static int* d_out = nullptr;
    if (!d_in) {
        FM_HIP(hipMalloc(&d_in, sizeof(float) * 8));
        FM_HIP(hipMalloc(&d_out, sizeof(int) * 16));
        const float h[8] = {0.83f, -0.127f, 1.21f, 1.037f, 0.0123f, -0.0456f, 0.f, 0.f};
        FM_HIP(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));
    }
    hipStream_t s = ctx->s_flow;
    FM_HIP(hipMemsetAsync(d_out, 0, sizeof(int) * 16, s));
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    switch (variant) {
    case 0: hipLaunchKernelGGL(pkhaz_kernel<0>, grid, block, 0, s, d_in, iters, d_out); break;
    case 1: hipLaunchKernelGGL(pkhaz_kernel<1>, grid, block, 0, s, d_in, iters, d_out); break;
    case 2: hipLaunchKernelGGL(pkhaz_kernel<2>, grid, block, 0, s, d_in, iters, d_out); break;
    case 3: hipLaunchKernelGGL(pkhaz_kernel<3>, grid, block, 0, s, d_in, iters, d_out); break;
    default: hipLaunchKernelGGL(pkhaz_kernel<4>, grid, block, 0, s, d_in, iters, d_out); break;
    }
    FM_HIP(hipGetLastError());
    FM_HIP(hipMemcpyAsync(out8, d_out, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, s));
    FM_HIP(hipStreamSynchronize(s));
    return 0;
}

// victim (one packed instruction class, 0..5) on the KLT stream, `launches` times, while neighbour class `aggressor`
// (0..6, -1 = none) runs on the ReID stream.  out8 as fm_diag_pkhaz.
extern "C" int fm_diag_pkhaz2(fm_ctx* ctx, int victim, int aggressor, int launches, int32_t* out8) {
    FM_CHECK_ARG(ctx && out8 && victim >= 0 && victim <= 5 && aggressor >= -1 && aggressor <= 6 && launches > 0);
    static float* d_in = nullptr;
    static float* d_sink = nullptr;
    static int* d_out = nullptr;
    const int agg_blocks = 256 * 6;
    if (!d_in) {
        FM_HIP(hipMalloc(&d_in, sizeof(float) * 512));
        FM_HIP(hipMalloc(&d_sink, sizeof(float) * 256 * agg_blocks));
        FM_HIP(hipMalloc(&d_out, sizeof(int) * 16));
        float h[512];
        unsigned r = 12345u;
        for (float& v : h) { r = r * 1664525u + 1013904223u; v = (float)((r >> 8) & 0xffff) / 65536.f + 0.25f; }
        FM_HIP(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));
    }
    FM_HIP(hipMemset(d_out, 0, sizeof(int) * 16));
    const int agg_iters = 6000;
#define FM_AGG(A) hipLaunchKernelGGL(pkneighbour_kernel<A>, dim3(agg_blocks), dim3(256), 0, ctx->s_ext, d_in, agg_iters, d_sink)
#define FM_VIC(V) hipLaunchKernelGGL(pkvictim_kernel<V>, dim3(150), dim3(256), 0, ctx->s_flow, d_in, 2000, d_out)
    for (int l = 0; l < launches; ++l) {
        if (l % 4 == 0) {
            switch (aggressor) {
            case 0: FM_AGG(0); break; case 1: FM_AGG(1); break; case 2: FM_AGG(2); break; case 3: FM_AGG(3); break;
            case 4: FM_AGG(4); break; case 5: FM_AGG(5); break; case 6: FM_AGG(6); break; default: break;
            }
        }
        switch (victim) {
        case 0: FM_VIC(0); break; case 1: FM_VIC(1); break; case 2: FM_VIC(2); break; case 3: FM_VIC(3); break;
        case 4: FM_VIC(4); break; default: FM_VIC(5); break;
        }
    }
#undef FM_AGG
#undef FM_VIC
    FM_HIP(hipGetLastError());
    FM_HIP(hipStreamSynchronize(ctx->s_flow));
    FM_HIP(hipStreamSynchronize(ctx->s_ext));
    FM_HIP(hipMemcpy(out8, d_out, sizeof(int32_t) * 8, hipMemcpyDeviceToHost));
    return 0;
}
#endif  // FM_DIAG

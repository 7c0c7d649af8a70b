/* fastmot_hip_diag.h -- diagnostic entry points of libfastmot_hip.so.  NOT part of the product ABI: they exist only
 * in a library built with -DFM_DIAG (FASTMOT_EXTRA_HIPCC_FLAGS=-DFM_DIAG python -m fastmot_amd.build); the shipped
 * library does not export them (tests/test_abi.py).  No counterpart in the reference: they are the apparatus of round 3's
 * bisect of the LK results that differed under load (scripts/lk_bisect.py, pkhaz.py, pkhaz2.py). */
#ifndef FASTMOT_HIP_DIAG_H
#define FASTMOT_HIP_DIAG_H
#include "fastmot_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics of the LK kernel (option "lk_variant" > 0 selects an instrumented variant of the kernel for
 * fm_flow_lk; no counterpart in the reference): 16 event counters, reset by the call; with hdr / records non-null the
 * capture of the last call, hdr [n][4] = HW_ID, XCC_ID, records written, workgroup; records [n][80][12][64]. */
int fm_flow_lk_diag_read(fm_ctx* ctx, int32_t* counters, int n, int32_t* hdr, int32_t* records);
/* Stand-alone reproducer of the packed-fp32 mis-execution that round 3's bisect found in the LK kernel (csrc/diag.hip,
 * DESIGN 5b): `waves` wavefronts x `iters` evaluations of the LK position update on the KLT stream;
 * out8[0..3] = lanes (per quarter of the wavefront) whose low result differed from lane 0's, out8[4..7] = high. */
int fm_diag_pkhaz(fm_ctx* ctx, int variant, int waves, int iters, int32_t* out8);
/* ... one packed instruction class (victim 0..5) checked per lane against unpacked arithmetic, `launches` launches on
 * the KLT stream while a synthetic neighbour kernel of instruction class `aggressor` (0..6, -1 = none) occupies the
 * ReID stream (csrc/diag.hip). */
int fm_diag_pkhaz2(fm_ctx* ctx, int victim, int aggressor, int launches, int32_t* out8);
/* diagnostic: a long deterministic kernel on the flow stream (mode bit 0: 32-lane butterflies, bit 1: byte
 * loads from the previous gray image); out_host: 256 * blocks words.  See scripts/stress_spin.py. */
int fm_debug_spin(fm_ctx* ctx, int blocks, int iters, int mode, unsigned* out_host);

#ifdef __cplusplus
}
#endif
#endif

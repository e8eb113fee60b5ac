// Shared host/device helpers for libraftcorr_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/raftcorr.h"

// ---------------------------------------------------------------- host side: errors
void rc_set_error(const char* fmt, ...);
int rc_check_launch(const char* what);

#define RC_REQUIRE(cond, code, ...)          \
  do {                                       \
    if (!(cond)) {                           \
      rc_set_error(__VA_ARGS__);             \
      return (code);                         \
    }                                        \
  } while (0)

// ---------------------------------------------------------------- element types
typedef _Float16 rc_half;
typedef __bf16 rc_bf16;

template <typename T> struct rc_traits;
template <> struct rc_traits<float> { static constexpr int id = RC_F32; };
template <> struct rc_traits<rc_half> { static constexpr int id = RC_F16; };
template <> struct rc_traits<rc_bf16> { static constexpr int id = RC_BF16; };

__device__ __forceinline__ float rc_to_float(float v) { return v; }
__device__ __forceinline__ float rc_to_float(rc_half v) { return (float)v; }
__device__ __forceinline__ float rc_to_float(rc_bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T rc_from_float(float v);
template <> __device__ __forceinline__ float rc_from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ rc_half rc_from_float<rc_half>(float v) { return (rc_half)v; }
template <> __device__ __forceinline__ rc_bf16 rc_from_float<rc_bf16>(float v) { return (rc_bf16)v; }

// round through the storage type (identity for float)
template <typename T> __device__ __forceinline__ float rc_round_as(float v) {
  return rc_to_float(rc_from_float<T>(v));
}

// ---------------------------------------------------------------- pyramid layout on device
// A pixel's column of level i is cut into groups of RB = 16 / sizeof(element) consecutive w2
// (one 16-byte block each).  Group g of pixel position q = w1 >> i is stored in block row
//     sb = (g - (q / RB) + NB - RC_WRAP_ROWS / RB) mod NB,      NB = widthp / RB,
// so the wrap-around of a column sits at disparities of about -RC_WRAP_ROWS level-i pixels.
constexpr int RC_WRAP_ROWS = 32;

struct rc_dev_layout {
  int32_t L;
  int32_t pitch;
  int32_t H, W1, W2;
  int32_t B;
  int32_t width[RC_MAX_LEVELS];
  int32_t widthp[RC_MAX_LEVELS];
  int32_t bias[RC_MAX_LEVELS];  // lookup only (set by its launcher): keeps the block-row argument >= 0
  int64_t offset[RC_MAX_LEVELS];
};

static inline rc_dev_layout rc_to_dev(const rc_layout* lay) {
  rc_dev_layout d;
  d.L = lay->L;
  d.pitch = lay->pitch;
  d.H = (int32_t)lay->H;
  d.W1 = (int32_t)lay->W1;
  d.W2 = (int32_t)lay->W2;
  d.B = (int32_t)lay->B;
  for (int i = 0; i < RC_MAX_LEVELS; ++i) {
    d.width[i] = lay->width[i];
    d.widthp[i] = lay->widthp[i];
    d.bias[i] = 0;
    d.offset[i] = lay->offset[i];
  }
  return d;
}

int rc_validate_layout(const rc_layout* lay);

// ---------------------------------------------------------------- tap coordinates
// Non-negative (a mod m) for m > 0.
__device__ __forceinline__ int rc_pmod(int a, int m) {
  int r = a % m;
  return r < 0 ? r + m : r;
}

// Block row of group g for pixel position q (both already divided down to level i); nb = block rows.
template <int RB>
__host__ __device__ __forceinline__ int rc_block_row(int g, int q, int nb) {
  int r = (g - q / RB + nb - RC_WRAP_ROWS / RB) % nb;
  return r < 0 ? r + nb : r;
}

// Element offset of (row bh, pixel w1, level-i column w2i) (RB = elements per 16-byte block).
template <int RB>
__device__ __forceinline__ int64_t rc_cell(const rc_dev_layout& lay, int i, int64_t bh, int w1, int w2i) {
  const int nb = lay.widthp[i] / RB;
  const int sb = rc_block_row<RB>(w2i / RB, w1 >> i, nb);
  return lay.offset[i] + ((bh * nb + sb) * lay.pitch + w1) * RB + (w2i % RB);
}

// float -> int that maps nan / huge values to a far-out-of-range index (all taps zero)
__device__ __forceinline__ int rc_safe_int(float f) {
  return (fabsf(f) < 1.0e8f) ? (int)f : -(1 << 28);
}

// EXACT coordinates (sampler_kernel.cu:42-57): base tap index and lerp weight for one level.
__device__ __forceinline__ void rc_exact_coords(float x, float inv_scale, int radius, int* base,
                                                float* a) {
  const float xi = x * inv_scale;  // exact: division by a power of two
  const float xf = floorf(xi);
  *a = xi - xf;
  *base = rc_safe_int(xf) - radius;
}

// GRIDSAMPLE coordinates of tap k (core/corr.py:137, core/utils/utils.py:63, grid_sample with
// align_corners=True).  Every operation rounds in fp32 exactly as the reference's tensors do;
// no contraction is possible here (no multiply feeds an add).
__device__ __forceinline__ void rc_gridsample_coords(float xi, int k_minus_r, float wm1, int* i0,
                                                     float* w0, float* w1) {
  const float p = __fadd_rn(xi, (float)k_minus_r);
  const float g = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, p), wm1), 1.0f);
  const float ix = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), wm1);
  const float x0 = floorf(ix);
  *w1 = __fsub_rn(ix, x0);
  *w0 = __fsub_rn(__fadd_rn(x0, 1.0f), ix);
  const bool ok = fabsf(x0) < 1.0e8f;
  *i0 = ok ? (int)x0 : -(1 << 28);
  if (!ok) {
    *w0 = 0.f;
    *w1 = 0.f;
  }
}

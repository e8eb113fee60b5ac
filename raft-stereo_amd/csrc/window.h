// Tap-window addressing over the group-skewed blocked pyramid, shared by the lookup kernels
// (lookup.hip) and the fused lookup + 1x1 convolution (fused.hip).
#ifndef RC_WINDOW_H
#define RC_WINDOW_H
#include "rc_common.h"

namespace {

template <typename VT> struct vt_info {
  static constexpr int RB = 16 / (int)sizeof(VT);
  static constexpr int LG = sizeof(VT) == 4 ? 2 : 3;  // log2(RB)
};

// geometry of a tap window: entries wbase .. wbase + NW - 1, wbase = (first tap) - EXT.  In
// grid_sample mode the window is one entry wider on both sides, because the reference's fp32
// coordinate round trip can move a tap across an integer.
template <int RB, int R, int MODE> struct win_info {
  static constexpr int NT = 2 * R + 1;
  static constexpr int EXT = MODE == RC_COORD_GRIDSAMPLE ? 1 : 0;
  static constexpr int NW = NT + 1 + 2 * EXT;
  static constexpr int NBLK = (RB - 1 + NW + RB - 1) / RB;  // 16-byte blocks a window can touch
  static constexpr int ND = NBLK * 4;
};

__device__ __forceinline__ float half_bits_to_float(uint32_t h, rc_half) {
  return (float)__builtin_bit_cast(rc_half, (uint16_t)h);
}
__device__ __forceinline__ float half_bits_to_float(uint32_t h, rc_bf16) { return __uint_as_float(h << 16); }

// An offset the compiler must keep in scalar registers (the value is wave-uniform by
// construction), so that base + uniform(offset) stays a scalar address and the lane index goes
// into the 32-bit vector offset of the memory instruction
__device__ __forceinline__ int64_t uniform(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// (t mod nb) for 0 <= t < bias + 2 nb, bias a multiple of nb chosen by set_row_bias
__device__ __forceinline__ uint32_t row_mod(uint32_t t, uint32_t nb, uint32_t bias) {
  if (bias <= 2 * nb) {  // wave-uniform; always, unless a level is only a few pixels wide
    t = min(t, t - 2 * nb);
    return min(t, t - nb);
  }
  return t % nb;
}

// Block row of the k-th block after row sb0 (< nb), wrapping around the column
__device__ __forceinline__ uint32_t row_step(uint32_t sb0, uint32_t k, uint32_t nb, bool tiny) {
  const uint32_t s = sb0 + k;
  return tiny ? s % nb : min(s, s - nb);  // tiny (nb < blocks per window): wave-uniform
}

// Per-level window position of one pixel
struct win_pos {
  int wbase;      // w2 of window entry 0
  int o;          // its position inside the first block
  int g0;         // group (block) of window entry 0
  uint32_t sb0;   // block row of that group in this pixel's column
  int nbk;        // index of the last block the window touches
  bool any_ok;    // the window overlaps the image
  bool edge;      // some entry is outside the image
};

template <int RB, int LG, int NW>
__device__ __forceinline__ win_pos locate_window(int wbase_raw, int w1, int i, int Wi, int nb, int bias) {
  win_pos p;
  p.any_ok = (wbase_raw + NW - 1 >= 0) && (wbase_raw < Wi);
  p.edge = (wbase_raw < 0) || (wbase_raw + NW - 1 >= Wi);
  p.wbase = wbase_raw;
  const int wb = p.any_ok ? wbase_raw : 0;  // keeps the group arithmetic small; such lanes output zeros
  p.g0 = wb >> LG;
  p.o = wb & (RB - 1);
  p.nbk = (p.o + NW - 1) >> LG;
  p.sb0 = row_mod((uint32_t)(p.g0 - ((w1 >> i) >> LG) + nb - RC_WRAP_ROWS / RB + bias), (uint32_t)nb,
                  (uint32_t)bias);
  return p;
}

// v[j] = element (o + j) of the window held in the dwords d[0..ND): shift down by `o` elements
// with two select stages (+ one byte-align for 16-bit elements)
template <typename VT, int NW, int ND>
__device__ __forceinline__ void decode_window(uint32_t (&d)[ND], int o, float (&v)[NW]) {
  const int sh = (sizeof(VT) == 4) ? o : (o >> 1);  // shift in dwords, 0..3
  // (both operands are read into temporaries first: a select between array ELEMENTS would be
  // turned into a dynamically indexed, scratch-backed array by the compiler)
  const bool s2 = (sh & 2) != 0, s1 = (sh & 1) != 0;
#pragma unroll
  for (int k = 0; k + 2 < ND; ++k) {
    const uint32_t lo = d[k], hi = d[k + 2];
    d[k] = s2 ? hi : lo;
  }
#pragma unroll
  for (int k = 0; k + 3 < ND; ++k) {
    const uint32_t lo = d[k], hi = d[k + 1];
    d[k] = s1 ? hi : lo;
  }
  if constexpr (sizeof(VT) == 4) {
#pragma unroll
    for (int j = 0; j < NW; ++j) v[j] = __uint_as_float(d[j]);
  } else {
    const uint32_t bsh = (o & 1) * 2;
#pragma unroll
    for (int k = 0; 2 * k < NW; ++k) {
      const uint32_t t = __builtin_amdgcn_alignbyte(d[k + 1], d[k], bsh);
      v[2 * k] = half_bits_to_float(t & 0xffffu, VT());
      if (2 * k + 1 < NW) v[2 * k + 1] = half_bits_to_float(t >> 16, VT());
    }
  }
}

// bias[i]: the smallest multiple of the level's block-row count that keeps
// g0 - (w1 >> i) / RB + nb - RC_WRAP_ROWS / RB non-negative for every window that overlaps the image
inline void set_row_bias(rc_dev_layout& d, int rb, int nw) {
  for (int i = 0; i < d.L; ++i) {
    const int nb = d.widthp[i] / rb;
    const int gneg = (nw - 1 + rb - 1) / rb;  // -(lowest group index of an overlapping window)
    const int lo = gneg + (((d.W1 - 1) >> i) / rb) + RC_WRAP_ROWS / rb - nb;
    d.bias[i] = lo > 0 ? (lo + nb - 1) / nb * nb : 0;
  }
}

// Argument checks common to every lookup entry point (a = the pyramid, b, c = two more required pointers)
inline int check_common(const rc_layout* lay, const void* a, const void* b, const void* c, int cc, int radius,
                 int mode) {
  RC_REQUIRE(lay && a && b && c, RC_EINVAL, "null pointer argument");
  int rc = rc_validate_layout(lay);
  if (rc) return rc;
  RC_REQUIRE(cc >= 1, RC_EINVAL, "coords must have at least one channel (got %d)", cc);
  RC_REQUIRE(radius >= 1 && radius <= RC_MAX_RADIUS, RC_EUNSUPPORTED, "radius %d outside 1..%d", radius,
             RC_MAX_RADIUS);
  RC_REQUIRE(mode == RC_COORD_EXACT || mode == RC_COORD_GRIDSAMPLE, RC_EINVAL, "bad coord_mode %d", mode);
  if (mode == RC_COORD_GRIDSAMPLE)
    RC_REQUIRE(lay->width[lay->L - 1] >= 2, RC_EINVAL,
               "grid_sample coordinates need every level at least 2 wide (W2=%lld, L=%d)",
               (long long)lay->W2, lay->L);
  RC_REQUIRE((int64_t)lay->B * lay->H * ((lay->W1 + 63) / 64) < (1ll << 31), RC_EINVAL, "too many pixels");
  RC_REQUIRE(lay->pitch < (1 << 20) && (int64_t)(lay->widthp[0] / lay->rb + 1) * lay->pitch * 16 < (1ll << 31),
             RC_EINVAL, "one image row of the pyramid exceeds the 32-bit lane offset range");
  RC_REQUIRE((uintptr_t)a % 16 == 0, RC_EINVAL, "pyramid pointer must be 16-byte aligned");
  return RC_OK;
}

}  // namespace

#endif  // RC_WINDOW_H

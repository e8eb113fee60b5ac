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

// ---- merged windows of a forward pyramid: level i (stored) read together with level i + 1 (derived).
// Entry j of level i+1 is the mean of entries 2j, 2j+1 of level i, so the NW entries of level i+1 starting
// at s_hi lie over the 2 NW entries of level i starting at 2 s_hi (an EVEN element: no sub-dword shift for
// 16-bit types), and because floor(x / 2^i) = 2 floor(x / 2^(i+1)) + bit, the level-i window starts T0 + bit
// entries into that run.
template <int RB, int R, int MODE> struct pair_info {
  using WI = win_info<RB, R, MODE>;
  static constexpr int NT = WI::NT, EXT = WI::EXT, NW = WI::NW;
  static constexpr int NW2 = 2 * NW;
  static constexpr int T0 = R + EXT;
  static constexpr int NBLK = (RB - 2 + NW2 + RB - 1) / RB;  // 16-byte blocks the run can touch
  static constexpr int ND = NBLK * 4;
};

// d[0..ND): the blocks of the run; o: (even) position of its first entry inside the first block.
// lo[m] = entry T0 + bit + m of the run (the level-i window), hi[j] = mean of entries 2j, 2j+1 rounded to the
// storage type (the level-(i+1) window): the arithmetic of the build's epilogue and of avg_pool2d on a
// 16-bit volume (fp32 mean of the stored values, one rounding).
template <typename VT, int NW, int ND, int T0>
__device__ __forceinline__ void decode_pair(uint32_t (&d)[ND], int o, int bit, float (&lo)[NW], float (&hi)[NW]) {
  if constexpr (sizeof(VT) == 4) {
    static_assert(2 * NW + 2 <= ND, "run does not fit its blocks");
    const bool s2 = (o & 2) != 0;  // o is 0 or 2
#pragma unroll
    for (int k = 0; k < 2 * NW; ++k) {
      const uint32_t a = d[k], b = d[k + 2];
      d[k] = s2 ? b : a;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) hi[j] = 0.5f * (__uint_as_float(d[2 * j]) + __uint_as_float(d[2 * j + 1]));
    const bool b1 = bit != 0;
#pragma unroll
    for (int m = 0; m < NW; ++m) {
      const uint32_t a = d[T0 + m], b = d[T0 + 1 + m];
      lo[m] = __uint_as_float(b1 ? b : a);
    }
  } else {
    static_assert(NW + 3 <= ND, "run does not fit its blocks");
    const int sh = o >> 1;  // shift in dwords, 0..3
    const bool s2 = (sh & 2) != 0, s1 = (sh & 1) != 0;
#pragma unroll
    for (int k = 0; k <= NW; ++k) {
      const uint32_t a = d[k], b = d[k + 2];
      d[k] = s2 ? b : a;
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const uint32_t a = d[k], b = d[k + 1];
      d[k] = s1 ? b : a;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j)
      hi[j] = rc_round_as<VT>(0.5f * (half_bits_to_float(d[j] & 0xffffu, VT()) + half_bits_to_float(d[j] >> 16, VT())));
    // the level-i window starts at element T0 + bit of the run: a byte permute of two neighbouring dwords
    constexpr int K0 = T0 / 2;
    constexpr uint32_t SEL0 = (T0 & 1) ? 0x05040302u : 0x03020100u, SEL1 = (T0 & 1) ? 0x07060504u : 0x05040302u;
    const uint32_t sel = bit ? SEL1 : SEL0;
#pragma unroll
    for (int p = 0; 2 * p < NW; ++p) {
      const uint32_t t = __builtin_amdgcn_perm(d[K0 + p + 1], d[K0 + p], sel);
      lo[2 * p] = half_bits_to_float(t & 0xffffu, VT());
      lo[2 * p + 1] = half_bits_to_float(t >> 16, VT());
    }
  }
}

// The 2r+1 taps of one level from its window entries v[0..NW) (entry m is column wbase + m of the level,
// already zero outside the image): floor + lerp (sampler_kernel.cu:42-57) or the reference's grid_sample
// coordinate round trip (core/corr.py:137, core/utils/utils.py:63), whose fp32 rounding can move a tap across
// an integer -- the window is one entry wider on both sides for that.
// Every tap is product + fma in a FIXED order (spelled out: of `a*b + c*d` the compiler contracts whichever
// product it likes, differently from one instantiation to the next, and rc_lookup_fwd_vt's taps must be
// rc_lookup_fwd's rounded).
template <int R, int MODE, int NW>
__device__ __forceinline__ void interp_taps(const float (&v)[NW], int wbase, int Wi, float xi, float frac,
                                            float (&tap)[2 * R + 1]) {
  constexpr int NT = 2 * R + 1, EXT = MODE == RC_COORD_GRIDSAMPLE ? 1 : 0;
  if constexpr (MODE == RC_COORD_EXACT) {
    const float a = frac, bw = 1.0f - a;
#pragma unroll
    for (int k = 0; k < NT; ++k) tap[k] = __builtin_fmaf(v[k + 1], a, v[k] * bw);
  } else {
    const float wm1 = (float)(Wi - 1);
    int dlt[NT];  // where grid_sample's floor landed relative to tap k: -1, 0 or +1
    float w0[NT], w1w[NT];
    bool odd = false;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      int i0k;
      rc_gridsample_coords(xi, k - R, wm1, &i0k, &w0[k], &w1w[k]);
      dlt[k] = i0k - (wbase + EXT + k);
      odd |= (dlt[k] != 0);
    }
    if (__builtin_expect(__any(odd), 0)) {
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        // (the operands are read into temporaries and pinned first: a select between array ELEMENTS is turned
        // into a dynamically indexed, scratch-backed array by the compiler -- 112 bytes of scratch per lane
        // and a third of the f32 lookup's speed when that happened)
        float a0 = v[k], a1 = v[k + 1], a2 = v[k + 2], a3 = v[k + 3];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        const bool lt = dlt[k] < 0, gt = dlt[k] > 0;
        const float va = lt ? a0 : (gt ? a2 : a1);
        const float vb = lt ? a1 : (gt ? a3 : a2);
        tap[k] = __builtin_fmaf(vb, w1w[k], va * w0[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NT; ++k) tap[k] = __builtin_fmaf(v[k + EXT + 1], w1w[k], v[k + EXT] * w0[k]);
    }
  }
}

// Loads of one window / run: NBLK 16-byte blocks of level `lvl_base` (wave-uniform base of the level's slab
// of this IMAGE, `slab_bytes` long), branch-free.  They are buffer loads: a block no entry of which is inside the
// image (the window hangs over the border, ends before block k, or misses the image altogether -- on the bench's
// disparity field 12 % of the pixels look left of the image) gets the offset `slab_bytes`, which the buffer's
// range check turns into zeros WITHOUT a memory access.  Plain loads of "some valid block" instead cost 18 % more
// lines from memory than the pixels needed.  `h` is the lane's image row: a wave owns 64
// consecutive pixels of the flattened image, so it may straddle two rows.  `ngroups`: blocks per column that
// hold image entries.
template <int NBLK>
__device__ __forceinline__ void load_blocks(uint32_t (&d)[NBLK * 4], const char* lvl_base, uint32_t slab_bytes,
                                            const win_pos& pos, int nb, int ngroups, uint32_t blk_bytes, int w1, int h) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const uint32_t row0 = (uint32_t)h * (uint32_t)nb;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(lvl_base), 0, (int)slab_bytes, 0x00020000);
  // blocks klo..khi of the window hold image entries (none: khi < klo)
  const int klo = max(0, -pos.g0);
  const int khi = pos.any_ok ? min(pos.nbk, ngroups - 1 - pos.g0) : -1;
  uint32_t sb = pos.sb0;  // < nb
#pragma unroll
  for (int k = 0; k < NBLK; ++k) {
    if (k > 0) {  // next block row of the column, wrapping around (exact for every nb >= 1)
      sb += 1;
      sb = sb == (uint32_t)nb ? 0u : sb;
    }
    uint32_t a = __umul24(row0 + sb, blk_bytes) + (uint32_t)w1 * 16u;
    asm volatile("" : "+v"(a));  // computed for every lane: the select below stays a v_cndmask, not a branch
    const uint32_t off = (k >= klo && k <= khi) ? a : slab_bytes;
    const u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    d[4 * k + 0] = q.x;
    d[4 * k + 1] = q.y;
    d[4 * k + 2] = q.z;
    d[4 * k + 3] = q.w;
  }
}

// Everything one pixel needs from levels i (stored) and i + 1 (derived) of a forward pyramid
template <typename VT, int R, int MODE> struct pair_fetch {
  static constexpr int RB = vt_info<VT>::RB, LG = vt_info<VT>::LG;
  using PI = pair_info<RB, R, MODE>;
  uint32_t d[PI::ND];
  win_pos pos;        // of the run at level i
  int s_lo, s_hi;     // first window entry (column) at level i / i + 1
  float x_lo, x_hi;   // the level coordinates
  float f_lo, f_hi;   // their fractional parts

  // b: image (wave-uniform); h, w1: the lane's pixel
  __device__ __forceinline__ void issue(const VT* vol, const rc_dev_layout& lay, int i, float x, int w1, int h, int b,
                                        uint32_t blk_bytes) {
    const int nb = lay.widthp[i] >> LG;
    x_lo = x * (1.0f / (float)(1 << i));
    x_hi = x * (0.5f / (float)(1 << i));
    const float fl = floorf(x_lo), fh = floorf(x_hi);
    f_lo = x_lo - fl;
    f_hi = x_hi - fh;
    s_lo = rc_safe_int(fl) - R - PI::EXT;
    s_hi = rc_safe_int(fh) - R - PI::EXT;
    pos = locate_window<RB, LG, PI::NW2>(2 * s_hi, w1, i, lay.width[i], nb, lay.bias[i]);
    const uint32_t slab = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)lay.H * (uint32_t)nb * blk_bytes);
    const char* lvl = reinterpret_cast<const char*>(vol) +
                      uniform(lay.offset[i] * (int64_t)sizeof(VT) + (int64_t)b * slab);
    load_blocks<PI::NBLK>(d, lvl, slab, pos, nb, (lay.width[i] + RB - 1) >> LG, blk_bytes, w1, h);
  }

  // window entries of both levels, zero outside the image
  __device__ __forceinline__ void windows(const rc_dev_layout& lay, int i, float (&lo)[PI::NW], float (&hi)[PI::NW]) {
    const int Wlo = lay.width[i], Whi = lay.width[i + 1];
    decode_pair<VT, PI::NW, PI::ND, PI::T0>(d, pos.o, (s_lo - 2 * s_hi - PI::T0) & 1, lo, hi);
    const bool edge = (s_lo < 0) || (s_lo + PI::NW > Wlo) || (s_hi < 0) || (s_hi + PI::NW > Whi);
    if (__any(edge)) {  // interior waves skip the zero padding
#pragma unroll
      for (int m = 0; m < PI::NW; ++m) {
        lo[m] = ((unsigned)(s_lo + m) < (unsigned)Wlo) ? lo[m] : 0.f;
        hi[m] = ((unsigned)(s_hi + m) < (unsigned)Whi) ? hi[m] : 0.f;
      }
    }
  }
};

// The same for a stored level read alone (the last level of a pyramid with an odd level count)
template <typename VT, int R, int MODE> struct single_fetch {
  static constexpr int RB = vt_info<VT>::RB, LG = vt_info<VT>::LG;
  using WI = win_info<RB, R, MODE>;
  uint32_t d[WI::ND];
  win_pos pos;
  float xi, frac;

  __device__ __forceinline__ void issue(const VT* vol, const rc_dev_layout& lay, int i, float x, int w1, int h, int b,
                                        uint32_t blk_bytes) {
    const int nb = lay.widthp[i] >> LG;
    xi = x * (1.0f / (float)(1 << i));
    const float xf = floorf(xi);
    frac = xi - xf;
    pos = locate_window<RB, LG, WI::NW>(rc_safe_int(xf) - R - WI::EXT, w1, i, lay.width[i], nb, lay.bias[i]);
    const uint32_t slab = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)lay.H * (uint32_t)nb * blk_bytes);
    const char* lvl = reinterpret_cast<const char*>(vol) +
                      uniform(lay.offset[i] * (int64_t)sizeof(VT) + (int64_t)b * slab);
    load_blocks<WI::NBLK>(d, lvl, slab, pos, nb, (lay.width[i] + RB - 1) >> LG, blk_bytes, w1, h);
  }

  __device__ __forceinline__ void window(const rc_dev_layout& lay, int i, float (&v)[WI::NW]) {
    decode_window<VT, WI::NW, WI::ND>(d, pos.o, v);
    if (__any(pos.edge)) {
#pragma unroll
      for (int m = 0; m < WI::NW; ++m) v[m] = ((unsigned)(pos.wbase + m) < (unsigned)lay.width[i]) ? v[m] : 0.f;
    }
  }
};

// A wave's share of a lookup: 64 consecutive pixels of the flattened image b (pixel = h * W1 + w1), so
// that every channel plane of the output [B, C, H, W1] is written in whole, aligned 256-byte pieces whatever
// W1 is (row-aligned chunks leave every row's last wave part empty and share cache lines between waves).
struct px_item {
  int b;        // image (wave-uniform)
  int p0;       // first flattened pixel (wave-uniform, a multiple of 64)
  int h, w1;    // the lane's pixel (the last pixel of the image for lanes past its end)
  bool live;    // the lane has a pixel
};
__device__ __forceinline__ px_item locate_item(int item, int chunks_per_image, int H, int W1, int lane) {
  px_item it;
  it.b = item / chunks_per_image;
  it.p0 = (item - it.b * chunks_per_image) * 64;
  const int HW = H * W1;
  const int p = it.p0 + lane;
  it.live = p < HW;
  const int pc = min(p, HW - 1);
  it.h = pc / W1;
  it.w1 = pc - it.h * W1;
  return it;
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
  RC_REQUIRE((int64_t)lay->B * ((lay->H * lay->W1 + 63) / 64) < (1ll << 31) && lay->H * lay->W1 < (1ll << 30), RC_EINVAL,
             "too many pixels");
  RC_REQUIRE(lay->pitch < (1 << 20) && lay->H * (int64_t)(lay->widthp[0] / lay->rb + 1) < (1ll << 24) &&
                 lay->H * (int64_t)(lay->widthp[0] / lay->rb + 1) * lay->pitch * 16 < (1ll << 32),
             RC_EINVAL, "one image's level 0 exceeds the 32-bit lane offset range (4 GiB)");
  RC_REQUIRE((uintptr_t)a % 16 == 0, RC_EINVAL, "pyramid pointer must be 16-byte aligned");
  return RC_OK;
}

}  // namespace

#endif  // RC_WINDOW_H

// Per-iteration correlation lookup, forward and adjoint, over the group-skewed blocked pyramid.
//
// Replaces core/corr.py:127-146 (reg), core/corr.py:44-51 + sampler/sampler_kernel.cu:19-105
// (reg_cuda) of the reference.  One launch covers all levels.  A wave owns 64 consecutive
// pixels of one image row, one pixel per lane.  In the rc_layout a pixel's column of a level is
// cut into aligned groups of 8 (f16/bf16) or 4 (f32) consecutive w2, one 16-byte block each,
// and consecutive groups of a pixel sit in consecutive block rows; a 2r+2 tap window is the
// 2-3 (16-bit) or 3-4 (f32) blocks that cover it.  The lane loads them with 16-byte loads and
// shifts the taps into place in registers.  Pixels with similar disparity use the same block
// rows, so each load instruction of the wave reads a few long contiguous runs.  Everything that
// does not depend on the lane (row, chunk, level bases, output planes) is kept wave-uniform, so
// loads and stores use scalar base + 32-bit lane offset addressing.  Outputs are written per
// channel plane, 256 B per wave store.
#include "rc_common.h"

namespace {

constexpr int LK_THREADS = 256;  // 4 waves, each its own 64-pixel chunk

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

// LV > 0 is a compile-time level count: the 16-byte loads of all levels are issued back to back
// (no branch in between: lanes that do not need a block re-read one they do need), then decoded,
// interpolated and stored.  LV == 0 takes the level count from the layout and handles one level
// at a time.
template <typename VT, int R, int MODE, int LV>
__global__ __launch_bounds__(LK_THREADS) void lookup_fwd_kernel(const VT* __restrict__ vol,
                                                                rc_dev_layout lay,
                                                                const float* __restrict__ coords,
                                                                int coords_channels,
                                                                float* __restrict__ out,
                                                                int chunks_per_row, int n_items) {
  constexpr int RB = vt_info<VT>::RB, LG = vt_info<VT>::LG;
  using WI = win_info<RB, R, MODE>;
  constexpr int NT = WI::NT, EXT = WI::EXT, NW = WI::NW, NBLK = WI::NBLK, ND = WI::ND;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * (LK_THREADS / 64) + wave;  // (row, 64-pixel chunk), wave-uniform
  if (item >= n_items) return;
  const int bh = item / chunks_per_row;
  const int c0 = (item - bh * chunks_per_row) * 64;
  const int w1 = c0 + lane;
  if (w1 >= lay.W1) return;
  const int b = bh / lay.H, h = bh - b * lay.H;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  constexpr int LB = LV ? LV : 1;  // levels in flight together
  const int L = LV ? LV : lay.L;
  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;  // distance between block rows of a pixel

  const float x = (coords + uniform(((int64_t)b * coords_channels) * HW + (int64_t)h * lay.W1 + c0))[lane];
  const int64_t out_u = ((int64_t)b * L * NT) * HW + (int64_t)h * lay.W1 + c0;  // wave-uniform

  for (int i0 = 0; i0 < L; i0 += LB) {
    uint32_t d[LB][ND];
    win_pos pos[LB];
    float frac[LB];

    // ---- phase 1: addresses and loads of LB levels
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int i = i0 + j;
      const int Wi = lay.width[i], nb = lay.widthp[i] >> LG;
      const float xi = x * (1.0f / (float)(1 << i));
      const float xf = floorf(xi);
      frac[j] = xi - xf;
      pos[j] = locate_window<RB, LG, NW>(rc_safe_int(xf) - R - EXT, w1, i, Wi, nb, lay.bias[i]);
      const char* lvl = reinterpret_cast<const char*>(vol) +
                        uniform(lay.offset[i] * (int64_t)sizeof(VT) + (int64_t)bh * nb * blk_bytes);
      const bool tiny = nb < NBLK;
#pragma unroll
      for (int k = 0; k < NBLK; ++k) {
        const uint32_t sb = row_step(pos[j].sb0, (uint32_t)min(k, pos[j].nbk), (uint32_t)nb, tiny);
        const uint4 q = *reinterpret_cast<const uint4*>(lvl + (__umul24(sb, blk_bytes) + (uint32_t)w1 * 16u));
        d[j][4 * k + 0] = q.x;
        d[j][4 * k + 1] = q.y;
        d[j][4 * k + 2] = q.z;
        d[j][4 * k + 3] = q.w;
      }
    }

    // ---- phase 2: decode, interpolate, store
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int i = i0 + j;
      const int Wi = lay.width[i];
      const int64_t op = out_u + (int64_t)i * NT * HW;  // wave-uniform plane base
      float v[NW];
      decode_window<VT, NW, ND>(d[j], pos[j].o, v);
      if (__any(pos[j].edge)) {  // interior waves skip the zero padding
#pragma unroll
        for (int m = 0; m < NW; ++m) v[m] = ((unsigned)(pos[j].wbase + m) < (unsigned)Wi) ? v[m] : 0.f;
      }

      if constexpr (MODE == RC_COORD_EXACT) {
        const float a = frac[j], bw = 1.0f - a;
#pragma unroll
        for (int k = 0; k < NT; ++k) (out + uniform(op + k * HW))[lane] = v[k] * bw + v[k + 1] * a;
      } else {
        const float xi = x * (1.0f / (float)(1 << i));
        const float wm1 = (float)(Wi - 1);
        int dlt[NT];  // where grid_sample's floor landed relative to tap k: -1, 0 or +1
        float w0[NT], w1w[NT];
        bool odd = false;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          int i0k;
          rc_gridsample_coords(xi, k - R, wm1, &i0k, &w0[k], &w1w[k]);
          dlt[k] = i0k - (pos[j].wbase + EXT + k);
          odd |= (dlt[k] != 0);
        }
        if (__builtin_expect(__any(odd), 0)) {
          // the fp32 round trip moved some tap across an integer: take it from the widened window
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            const float va = dlt[k] < 0 ? v[k] : (dlt[k] > 0 ? v[k + 2] : v[k + 1]);
            const float vb = dlt[k] < 0 ? v[k + 1] : (dlt[k] > 0 ? v[k + 3] : v[k + 2]);
            (out + uniform(op + k * HW))[lane] = va * w0[k] + vb * w1w[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < NT; ++k)
            (out + uniform(op + k * HW))[lane] = v[k + EXT] * w0[k] + v[k + EXT + 1] * w1w[k];
        }
      }
    }
  }
}

// Adjoint: grad_vol (float pyramid, RB = 4) += scatter of grad_out.  The pixel owns its column
// of every level, so plain read-modify-write of its 16-byte blocks is race free.  Per level the
// gradient planes and the blocks are loaded first, then updated and stored; blocks of groups that
// do not exist (window hanging over the image border) are loaded from a valid address and not
// stored.
template <int R, int MODE>
__global__ __launch_bounds__(LK_THREADS) void lookup_bwd_kernel(const float* __restrict__ coords,
                                                                int coords_channels,
                                                                const float* __restrict__ gout,
                                                                float* __restrict__ gvol,
                                                                rc_dev_layout lay, int chunks_per_row,
                                                                int n_items) {
  constexpr int RB = 4, LG = 2;
  using WI = win_info<RB, R, MODE>;
  constexpr int NT = WI::NT, EXT = WI::EXT, NW = WI::NW, NBLK = WI::NBLK, ND = WI::ND;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * (LK_THREADS / 64) + wave;
  if (item >= n_items) return;
  const int bh = item / chunks_per_row;
  const int c0 = (item - bh * chunks_per_row) * 64;
  const int w1 = c0 + lane;
  if (w1 >= lay.W1) return;
  const int b = bh / lay.H, h = bh - b * lay.H;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;

  const float x = (coords + uniform(((int64_t)b * coords_channels) * HW + (int64_t)h * lay.W1 + c0))[lane];
  const int64_t g_u = ((int64_t)b * lay.L * NT) * HW + (int64_t)h * lay.W1 + c0;  // wave-uniform

  for (int i = 0; i < lay.L; ++i) {
    const int Wi = lay.width[i], nb = lay.widthp[i] >> LG;
    const float xi = x * (1.0f / (float)(1 << i));
    const float xf = floorf(xi);
    const float a = xi - xf;
    const win_pos pos = locate_window<RB, LG, NW>(rc_safe_int(xf) - R - EXT, w1, i, Wi, nb, lay.bias[i]);
    char* lvl = reinterpret_cast<char*>(gvol) +
                uniform(lay.offset[i] * (int64_t)sizeof(float) + (int64_t)bh * nb * blk_bytes);
    const bool tiny = nb < NBLK;

    float g[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) g[k] = (gout + uniform(g_u + ((int64_t)i * NT + k) * HW))[lane];

    // the blocks of the window: block k holds group g0 + k; `own[k]` = the group exists and this
    // lane's window reaches it
    uint32_t off[NBLK];
    bool own[NBLK];
    float4 q[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      const int gk = pos.g0 + k;
      own[k] = pos.any_ok && k <= pos.nbk && gk >= 0 && gk < nb;
      const uint32_t sb = row_step(pos.sb0, (uint32_t)(own[k] ? k : 0), (uint32_t)nb, tiny);
      off[k] = __umul24(sb, blk_bytes) + (uint32_t)w1 * 16u;
      q[k] = *reinterpret_cast<const float4*>(lvl + off[k]);
    }

    // t[m] = gradient of window entry m (w2 = wbase + m)
    float t[ND];
#pragma unroll
    for (int m = 0; m < ND; ++m) t[m] = 0.f;
    if constexpr (MODE == RC_COORD_EXACT) {
      const float bw = 1.0f - a;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        t[k] += g[k] * bw;
        t[k + 1] += g[k] * a;
      }
    } else {
      const float wm1 = (float)(Wi - 1);
      int dlt[NT];
      float w0[NT], w1w[NT];
      bool odd = false;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        int i0k;
        rc_gridsample_coords(xi, k - R, wm1, &i0k, &w0[k], &w1w[k]);
        dlt[k] = i0k - (pos.wbase + EXT + k);
        odd |= (dlt[k] != 0);
      }
      if (__builtin_expect(__any(odd), 0)) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const float a0 = g[k] * w0[k], a1 = g[k] * w1w[k];
          t[k] += dlt[k] < 0 ? a0 : 0.f;
          t[k + 1] += dlt[k] < 0 ? a1 : (dlt[k] == 0 ? a0 : 0.f);
          t[k + 2] += dlt[k] == 0 ? a1 : (dlt[k] > 0 ? a0 : 0.f);
          t[k + 3] += dlt[k] > 0 ? a1 : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          t[k + EXT] += g[k] * w0[k];
          t[k + EXT + 1] += g[k] * w1w[k];
        }
      }
    }
#pragma unroll
    for (int m = 0; m < NW; ++m) t[m] = ((unsigned)(pos.wbase + m) < (unsigned)Wi) ? t[m] : 0.f;

    // move window entry m to slot o + m of the blocks
    {
      const bool s1 = (pos.o & 1) != 0, s2 = (pos.o & 2) != 0;
#pragma unroll
      for (int m = ND - 1; m >= 1; --m) {
        const float lo = t[m - 1], hi = t[m];
        t[m] = s1 ? lo : hi;
      }
      t[0] = s1 ? 0.f : t[0];
#pragma unroll
      for (int m = ND - 1; m >= 2; --m) {
        const float lo = t[m - 2], hi = t[m];
        t[m] = s2 ? lo : hi;
      }
      t[1] = s2 ? 0.f : t[1];
      t[0] = s2 ? 0.f : t[0];
    }
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      if (own[k]) {
        float4 r = q[k];
        r.x += t[4 * k + 0];
        r.y += t[4 * k + 1];
        r.z += t[4 * k + 2];
        r.w += t[4 * k + 3];
        *reinterpret_cast<float4*>(lvl + off[k]) = r;
      }
    }
  }
}

// bias[i]: the smallest multiple of the level's block-row count that keeps
// g0 - (w1 >> i) / RB + nb - RC_WRAP_ROWS / RB non-negative for every window that overlaps the image
void set_row_bias(rc_dev_layout& d, int rb, int nw) {
  for (int i = 0; i < d.L; ++i) {
    const int nb = d.widthp[i] / rb;
    const int gneg = (nw - 1 + rb - 1) / rb;  // -(lowest group index of an overlapping window)
    const int lo = gneg + (((d.W1 - 1) >> i) / rb) + RC_WRAP_ROWS / rb - nb;
    d.bias[i] = lo > 0 ? (lo + nb - 1) / nb * nb : 0;
  }
}

template <typename VT, int R, int MODE>
int launch_fwd_mode(const void* vol, rc_dev_layout d, const float* coords, int cc, float* out, hipStream_t st) {
  using WI = win_info<vt_info<VT>::RB, R, MODE>;
  set_row_bias(d, vt_info<VT>::RB, WI::NW);
  const int chunks = (d.W1 + 63) / 64;
  const int n_items = (int)((int64_t)d.B * d.H * chunks);
  const int blocks = (n_items + (LK_THREADS / 64) - 1) / (LK_THREADS / 64);
  if (R == 4 && d.L == 4)  // the model's configuration: all levels' loads in flight together
    hipLaunchKernelGGL((lookup_fwd_kernel<VT, R, MODE, (R == 4 ? 4 : 0)>), dim3(blocks), dim3(LK_THREADS), 0, st,
                       (const VT*)vol, d, coords, cc, out, chunks, n_items);
  else
    hipLaunchKernelGGL((lookup_fwd_kernel<VT, R, MODE, 0>), dim3(blocks), dim3(LK_THREADS), 0, st, (const VT*)vol,
                       d, coords, cc, out, chunks, n_items);
  return rc_check_launch("rc_lookup_fwd");
}

template <typename VT, int R>
int launch_fwd(const void* vol, const rc_dev_layout& d, const float* coords, int cc, float* out, int mode,
               hipStream_t st) {
  return mode == RC_COORD_EXACT ? launch_fwd_mode<VT, R, RC_COORD_EXACT>(vol, d, coords, cc, out, st)
                                : launch_fwd_mode<VT, R, RC_COORD_GRIDSAMPLE>(vol, d, coords, cc, out, st);
}

template <typename VT>
int dispatch_fwd_radius(const void* vol, const rc_dev_layout& d, const float* coords, int cc, float* out,
                        int radius, int mode, hipStream_t st) {
  switch (radius) {
    case 1: return launch_fwd<VT, 1>(vol, d, coords, cc, out, mode, st);
    case 2: return launch_fwd<VT, 2>(vol, d, coords, cc, out, mode, st);
    case 3: return launch_fwd<VT, 3>(vol, d, coords, cc, out, mode, st);
    case 4: return launch_fwd<VT, 4>(vol, d, coords, cc, out, mode, st);
    case 5: return launch_fwd<VT, 5>(vol, d, coords, cc, out, mode, st);
    case 6: return launch_fwd<VT, 6>(vol, d, coords, cc, out, mode, st);
    case 7: return launch_fwd<VT, 7>(vol, d, coords, cc, out, mode, st);
    case 8: return launch_fwd<VT, 8>(vol, d, coords, cc, out, mode, st);
  }
  rc_set_error("radius %d outside 1..%d", radius, RC_MAX_RADIUS);
  return RC_EUNSUPPORTED;
}

template <int R, int MODE>
int launch_bwd_mode(const float* coords, int cc, const float* gout, float* gvol, rc_dev_layout d, hipStream_t st) {
  using WI = win_info<4, R, MODE>;
  set_row_bias(d, 4, WI::NW);
  const int chunks = (d.W1 + 63) / 64;
  const int n_items = (int)((int64_t)d.B * d.H * chunks);
  const int blocks = (n_items + (LK_THREADS / 64) - 1) / (LK_THREADS / 64);
  hipLaunchKernelGGL((lookup_bwd_kernel<R, MODE>), dim3(blocks), dim3(LK_THREADS), 0, st, coords, cc, gout, gvol,
                     d, chunks, n_items);
  return rc_check_launch("rc_lookup_bwd");
}

template <int R>
int launch_bwd(const float* coords, int cc, const float* gout, float* gvol, const rc_dev_layout& d,
               int mode, hipStream_t st) {
  return mode == RC_COORD_EXACT ? launch_bwd_mode<R, RC_COORD_EXACT>(coords, cc, gout, gvol, d, st)
                                : launch_bwd_mode<R, RC_COORD_GRIDSAMPLE>(coords, cc, gout, gvol, d, st);
}

int check_common(const rc_layout* lay, const void* a, const void* b, const void* c, int cc, int radius,
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

extern "C" int rc_lookup_fwd(const void* vol, const rc_layout* lay, const float* coords, int coords_channels,
                             float* out, int radius, int coord_mode, void* stream) {
  int rc = check_common(lay, vol, coords, out, coords_channels, radius, coord_mode);
  if (rc) return rc;
  const rc_dev_layout d = rc_to_dev(lay);
  hipStream_t st = (hipStream_t)stream;
  switch (lay->dtype) {
    case RC_F32: return dispatch_fwd_radius<float>(vol, d, coords, coords_channels, out, radius, coord_mode, st);
    case RC_F16: return dispatch_fwd_radius<rc_half>(vol, d, coords, coords_channels, out, radius, coord_mode, st);
    case RC_BF16: return dispatch_fwd_radius<rc_bf16>(vol, d, coords, coords_channels, out, radius, coord_mode, st);
  }
  rc_set_error("unknown pyramid dtype %d", lay->dtype);
  return RC_EUNSUPPORTED;
}

extern "C" int rc_lookup_bwd(const float* coords, int coords_channels, const float* grad_out,
                             float* grad_vol, const rc_layout* glay, int radius, int coord_mode,
                             void* stream) {
  int rc = check_common(glay, grad_vol, coords, grad_out, coords_channels, radius, coord_mode);
  if (rc) return rc;
  RC_REQUIRE(glay->dtype == RC_F32, RC_EINVAL, "the gradient pyramid must use an RC_F32 layout");
  const rc_dev_layout d = rc_to_dev(glay);
  hipStream_t st = (hipStream_t)stream;
  switch (radius) {
    case 1: return launch_bwd<1>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 2: return launch_bwd<2>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 3: return launch_bwd<3>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 4: return launch_bwd<4>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 5: return launch_bwd<5>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 6: return launch_bwd<6>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 7: return launch_bwd<7>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 8: return launch_bwd<8>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
  }
  return RC_EUNSUPPORTED;
}

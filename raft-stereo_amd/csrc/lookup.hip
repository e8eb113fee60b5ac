// Per-iteration correlation lookup, forward and adjoint, over the blocked-skew pyramid.
//
// Replaces core/corr.py:127-146 (reg), core/corr.py:44-51 + sampler/sampler_kernel.cu:19-105
// (reg_cuda) of the reference.  One launch covers all levels.  A wave owns 64 consecutive
// pixels of one image row, one pixel per lane.  In the rc_layout a pixel's 2r+2 taps of a
// level are consecutive skew rows of its own column, stored as aligned 16-byte blocks; the
// lane loads the 2-3 (f16/bf16) or 3-4 (f32) blocks that cover its window with 16-byte loads
// and shifts the taps into place in registers.  Pixels with similar disparity use the same row
// blocks, so each load instruction of the wave reads one contiguous 1 KiB run.  Outputs are
// written per channel plane, 256 B per wave store.
#include "rc_common.h"

namespace {

constexpr int LK_THREADS = 256;  // 4 waves, each its own 64-pixel chunk

template <typename VT> struct vt_info {
  static constexpr int RB = 16 / (int)sizeof(VT);
  static constexpr int LG = sizeof(VT) == 4 ? 2 : 3;  // log2(RB)
};

__device__ __forceinline__ float half_bits_to_float(uint32_t h, rc_half) {
  return (float)__builtin_bit_cast(rc_half, (uint16_t)h);
}
__device__ __forceinline__ float half_bits_to_float(uint32_t h, rc_bf16) { return __uint_as_float(h << 16); }

// (t mod m) in [0, m) for t + bias >= 0, bias a multiple of m (set_skew_bias): two conditional
// subtractions when t + bias < 4 m (always, unless a level is only a few pixels wide)
__device__ __forceinline__ int skew_pmod(int t_plus_bias, int m, int bias) {
  uint32_t r = (uint32_t)t_plus_bias;
  if (bias <= 3 * m) {  // wave-uniform
    r = min(r, r - 2u * (uint32_t)m);
    r = min(r, r - (uint32_t)m);
    return (int)r;
  }
  return (int)(r % (uint32_t)m);
}

// An offset the compiler must keep in scalar registers (the value is wave-uniform by
// construction), so that base + uniform(offset) stays a scalar address and the lane index goes
// into the 32-bit vector offset of the memory instruction
__device__ __forceinline__ int64_t uniform(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// v[j] = element (o + j) of the window held in the dwords d[0..ND): shift down by `o` elements
// with two select stages (+ one byte-align for 16-bit elements)
template <typename VT, int NT, int ND>
__device__ __forceinline__ void decode_window(uint32_t (&d)[ND], int o, float (&v)[NT + 1]) {
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
    for (int j = 0; j <= NT; ++j) v[j] = __uint_as_float(d[j]);
  } else {
    const uint32_t bsh = (o & 1) * 2;
#pragma unroll
    for (int k = 0; 2 * k <= NT; ++k) {
      const uint32_t t = __builtin_amdgcn_alignbyte(d[k + 1], d[k], bsh);
      v[2 * k] = half_bits_to_float(t & 0xffffu, VT());
      if (2 * k + 1 <= NT) v[2 * k + 1] = half_bits_to_float(t >> 16, VT());
    }
  }
}

// A wave owns 64 consecutive pixels of one image row.  Everything that does not depend on the
// lane (row, chunk, level bases, output planes) is kept wave-uniform so that loads and stores use
// scalar base + 32-bit lane offset addressing.  LV > 0 is a compile-time level count: the 16-byte
// loads of all levels are issued back to back (no branch in between: lanes that do not need a
// block re-read one they do need), then decoded, interpolated and stored.  LV == 0 takes the level
// count from the layout and handles one level at a time.
template <typename VT, int R, int MODE, int LV>
__global__ __launch_bounds__(LK_THREADS) void lookup_fwd_kernel(const VT* __restrict__ vol,
                                                                rc_dev_layout lay,
                                                                const float* __restrict__ coords,
                                                                int coords_channels,
                                                                float* __restrict__ out,
                                                                int chunks_per_row, int n_items) {
  constexpr int NT = 2 * R + 1;
  constexpr int RB = vt_info<VT>::RB, LG = vt_info<VT>::LG;
  constexpr int NBLK = (RB + NT + RB - 1) / RB;  // 16-byte blocks a window can touch
  constexpr int ND = NBLK * 4;
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
  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;  // distance between row blocks of a pixel

  const float x = (coords + uniform(((int64_t)b * coords_channels) * HW + (int64_t)h * lay.W1 + c0))[lane];
  const int64_t out_u = ((int64_t)b * L * NT) * HW + (int64_t)h * lay.W1 + c0;  // wave-uniform

  for (int i0 = 0; i0 < L; i0 += LB) {
    uint32_t d[LB][ND];
    int base[LB], s0[LB], o[LB];
    float frac[LB];
    bool slow[LB], edge[LB];

    // ---- phase 1: addresses and loads of LB levels
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int i = i0 + j;
      {
        const int Wi = lay.width[i], WiP = lay.widthp[i];
        const float xi = x * (1.0f / (float)(1 << i));
        const float xf = floorf(xi);
        frac[j] = xi - xf;
        base[j] = rc_safe_int(xf) - R;
        const bool any_ok = (base[j] + NT >= 0) && (base[j] < Wi);
        edge[j] = (base[j] < 0) || (base[j] + NT >= Wi);  // some tap is outside the image
        const int bias = lay.bias[i];
        const int t = (any_ok ? base[j] - (w1 >> i) - RC_SKEW : 0) + bias;
        s0[j] = skew_pmod(t, WiP, bias);
        const bool fast = any_ok && (s0[j] + NT < WiP);
        slow[j] = any_ok && !fast;  // the window wraps around the modulus (disparity about -12..-21)
        const int s0c = fast ? s0[j] : 0;
        o[j] = s0c & (RB - 1);
        const int blk0 = s0c >> LG;
        const int nb = fast ? ((o[j] + NT) >> LG) : 0;  // last block this lane needs
        const char* lvl = reinterpret_cast<const char*>(vol) +
                          uniform(lay.offset[i] * (int64_t)sizeof(VT) + (int64_t)bh * (WiP >> LG) * blk_bytes);
        const uint32_t voff = __umul24((uint32_t)blk0, blk_bytes) + (uint32_t)w1 * 16u;
#pragma unroll
        for (int k = 0; k < NBLK; ++k) {
          const uint32_t kk = (uint32_t)min(k, nb);
          const uint4 q = *reinterpret_cast<const uint4*>(lvl + (voff + __umul24(kk, blk_bytes)));
          d[j][4 * k + 0] = q.x;
          d[j][4 * k + 1] = q.y;
          d[j][4 * k + 2] = q.z;
          d[j][4 * k + 3] = q.w;
        }
      }
    }

    // ---- phase 2: decode, interpolate, store
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int i = i0 + j;
      {
        const int Wi = lay.width[i], WiP = lay.widthp[i];
        const int64_t op = out_u + (int64_t)i * NT * HW;  // wave-uniform plane base
        float v[NT + 1];
        decode_window<VT, NT, ND>(d[j], o[j], v);
        if (__builtin_expect(__any(slow[j]), 0)) {
          if (slow[j]) {
            const VT* col = vol + lay.offset[i] + ((int64_t)bh * (WiP >> LG) * lay.pitch + w1) * RB;
#pragma unroll
            for (int m = 0; m <= NT; ++m) {
              const int s = (s0[j] + m) % WiP;  // levels narrower than the window wrap more than once
              v[m] = rc_to_float(col[(int64_t)(s >> LG) * lay.pitch * RB + (s & (RB - 1))]);
            }
          }
        }
        if (__any(edge[j])) {  // interior waves skip the zero padding
#pragma unroll
          for (int m = 0; m <= NT; ++m) v[m] = ((unsigned)(base[j] + m) < (unsigned)Wi) ? v[m] : 0.f;
        }

        if (MODE == RC_COORD_EXACT) {
          const float a = frac[j], bw = 1.0f - a;
#pragma unroll
          for (int k = 0; k < NT; ++k) (out + uniform(op + k * HW))[lane] = v[k] * bw + v[k + 1] * a;
        } else {
          const float xi = x * (1.0f / (float)(1 << i));
          const float wm1 = (float)(Wi - 1);
          const int qm = w1 >> i;
          int i0k[NT];
          float w0[NT], w1w[NT];
          bool odd = false;
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            rc_gridsample_coords(xi, k - R, wm1, &i0k[k], &w0[k], &w1w[k]);
            odd |= (i0k[k] != base[j] + k);
          }
          if (__builtin_expect(__any(odd), 0)) {
            // the fp32 round trip moved some tap across an integer: fetch those taps directly
            const VT* col = vol + lay.offset[i] + ((int64_t)bh * (WiP >> LG) * lay.pitch + w1) * RB;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
              float va = v[k], vb = v[k + 1];
              if (i0k[k] != base[j] + k) {
                const int wa = i0k[k], wb = i0k[k] + 1;
                const int sa = rc_pmod(wa - qm - RC_SKEW, WiP), sb = rc_pmod(wb - qm - RC_SKEW, WiP);
                va = ((unsigned)wa < (unsigned)Wi)
                         ? rc_to_float(col[(int64_t)(sa >> LG) * lay.pitch * RB + (sa & (RB - 1))]) : 0.f;
                vb = ((unsigned)wb < (unsigned)Wi)
                         ? rc_to_float(col[(int64_t)(sb >> LG) * lay.pitch * RB + (sb & (RB - 1))]) : 0.f;
              }
              (out + uniform(op + k * HW))[lane] = va * w0[k] + vb * w1w[k];
            }
          } else {
#pragma unroll
            for (int k = 0; k < NT; ++k) (out + uniform(op + k * HW))[lane] = v[k] * w0[k] + v[k + 1] * w1w[k];
          }
        }
      }
    }
  }
}

// Adjoint: grad_vol (float pyramid, RB = 4) += scatter of grad_out.  The pixel owns its column
// of every level, so plain read-modify-write of its 16-byte blocks is race free.
template <int R, int MODE>
__global__ __launch_bounds__(LK_THREADS) void lookup_bwd_kernel(const float* __restrict__ coords,
                                                                int coords_channels,
                                                                const float* __restrict__ gout,
                                                                float* __restrict__ gvol,
                                                                rc_dev_layout lay, int chunks_per_row,
                                                                int n_items) {
  constexpr int NT = 2 * R + 1;
  constexpr int RB = 4;
  constexpr int NBLK = (RB + NT + RB - 1) / RB;
  constexpr int ND = NBLK * 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * (LK_THREADS / 64) + wave;
  if (item >= n_items) return;
  const int bh = item / chunks_per_row;
  const int w1 = (item - bh * chunks_per_row) * 64 + lane;
  if (w1 >= lay.W1) return;
  const int b = bh / lay.H, h = bh - b * lay.H;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const int64_t bstride = (int64_t)lay.pitch * RB;

  const float x = coords[((int64_t)b * coords_channels) * HW + (int64_t)h * lay.W1 + w1];
  const float* gp = gout + ((int64_t)b * lay.L * NT) * HW + (int64_t)h * lay.W1 + w1;

  float inv = 1.0f;
  for (int i = 0; i < lay.L; ++i) {
    const int Wi = lay.width[i], WiP = lay.widthp[i];
    const int qm = (w1 >> i) % WiP;
    float* col = gvol + lay.offset[i] + ((int64_t)bh * (WiP / RB) * lay.pitch + w1) * RB;
    const float xi = x * inv;
    inv *= 0.5f;
    const float xf = floorf(xi);
    const float a = xi - xf;
    const int base = rc_safe_int(xf) - R;

    float g[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) g[k] = gp[((int64_t)i * NT + k) * HW];

    float t[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) t[j] = 0.f;

    if (MODE == RC_COORD_EXACT) {
      const float bw = 1.0f - a;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        t[k] += g[k] * bw;
        t[k + 1] += g[k] * a;
      }
    } else {
      const float wm1 = (float)(Wi - 1);
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        int i0;
        float w0, w1w;
        rc_gridsample_coords(xi, k - R, wm1, &i0, &w0, &w1w);
        if (i0 == base + k) {
          t[k] += g[k] * w0;
          t[k + 1] += g[k] * w1w;
        } else {
          // rare: this tap landed on a neighbouring integer; update its two cells directly
          const int wa = i0, wb = i0 + 1;
          if ((unsigned)wa < (unsigned)Wi) {
            const int s = rc_pmod(wa - qm - RC_SKEW, WiP);
            col[(int64_t)(s / RB) * bstride + (s % RB)] += g[k] * w0;
          }
          if ((unsigned)wb < (unsigned)Wi) {
            const int s = rc_pmod(wb - qm - RC_SKEW, WiP);
            col[(int64_t)(s / RB) * bstride + (s % RB)] += g[k] * w1w;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j <= NT; ++j) t[j] = ((unsigned)(base + j) < (unsigned)Wi) ? t[j] : 0.f;

    const bool any_ok = (base + NT >= 0) && (base < Wi);
    if (!any_ok) continue;
    const int s0 = rc_pmod(base - qm - RC_SKEW, WiP);
    if (s0 + NT < WiP) {
      const int blk0 = s0 / RB, o = s0 - blk0 * RB;
      // move tap j to slot o + j
#pragma unroll
      for (int m = ND - 1; m >= 1; --m) t[m] = (o & 1) ? t[m - 1] : t[m];
      t[0] = (o & 1) ? 0.f : t[0];
#pragma unroll
      for (int m = ND - 1; m >= 2; --m) t[m] = (o & 2) ? t[m - 2] : t[m];
      t[1] = (o & 2) ? 0.f : t[1];
      t[0] = (o & 2) ? 0.f : t[0];
      float4* p = reinterpret_cast<float4*>(col + (int64_t)blk0 * bstride);
      const int64_t bs16 = bstride / 4;
#pragma unroll
      for (int k = 0; k < NBLK; ++k) {
        if (k * RB <= o + NT) {
          float4 q = p[k * bs16];
          q.x += t[4 * k + 0];
          q.y += t[4 * k + 1];
          q.z += t[4 * k + 2];
          q.w += t[4 * k + 3];
          p[k * bs16] = q;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j <= NT; ++j) {
        if ((unsigned)(base + j) < (unsigned)Wi) {
          const int s = (s0 + j) % WiP;
          col[(int64_t)(s / RB) * bstride + (s % RB)] += t[j];
        }
      }
    }
  }
}

// bias[i]: the smallest multiple of widthp[i] with base - (w1 >> i) - RC_SKEW + bias >= 0 for every
// in-range window (base >= -(2r+1))
void set_skew_bias(rc_dev_layout& d, int radius) {
  for (int i = 0; i < d.L; ++i) {
    const int lo = (2 * radius + 1) + ((d.W1 - 1) >> i) + RC_SKEW;
    d.bias[i] = (lo + d.widthp[i] - 1) / d.widthp[i] * d.widthp[i];
  }
}

template <typename VT, int R>
int launch_fwd(const void* vol, const rc_dev_layout& d_in, const float* coords, int cc, float* out,
               int mode, hipStream_t st) {
  rc_dev_layout d = d_in;
  set_skew_bias(d, R);
  const int chunks = (d.W1 + 63) / 64;
  const int64_t items64 = (int64_t)d.B * d.H * chunks;
  const int n_items = (int)items64;
  const int blocks = (n_items + (LK_THREADS / 64) - 1) / (LK_THREADS / 64);
#define RC_LAUNCH(MODE, LV)                                                                                \
  hipLaunchKernelGGL((lookup_fwd_kernel<VT, R, MODE, LV>), dim3(blocks), dim3(LK_THREADS), 0, st, (const VT*)vol, \
                     d, coords, cc, out, chunks, n_items)
  if (R == 4 && d.L == 4) {  // the model's configuration: all levels' loads in flight together
    if (mode == RC_COORD_EXACT) RC_LAUNCH(RC_COORD_EXACT, 4); else RC_LAUNCH(RC_COORD_GRIDSAMPLE, 4);
  } else {
    if (mode == RC_COORD_EXACT) RC_LAUNCH(RC_COORD_EXACT, 0); else RC_LAUNCH(RC_COORD_GRIDSAMPLE, 0);
  }
#undef RC_LAUNCH
  return rc_check_launch("rc_lookup_fwd");
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

template <int R>
int launch_bwd(const float* coords, int cc, const float* gout, float* gvol, const rc_dev_layout& d,
               int mode, hipStream_t st) {
  const int chunks = (d.W1 + 63) / 64;
  const int n_items = (int)((int64_t)d.B * d.H * chunks);
  const int blocks = (n_items + (LK_THREADS / 64) - 1) / (LK_THREADS / 64);
  if (mode == RC_COORD_EXACT)
    hipLaunchKernelGGL((lookup_bwd_kernel<R, RC_COORD_EXACT>), dim3(blocks), dim3(LK_THREADS), 0, st, coords,
                       cc, gout, gvol, d, chunks, n_items);
  else
    hipLaunchKernelGGL((lookup_bwd_kernel<R, RC_COORD_GRIDSAMPLE>), dim3(blocks), dim3(LK_THREADS), 0, st,
                       coords, cc, gout, gvol, d, chunks, n_items);
  return rc_check_launch("rc_lookup_bwd");
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
  RC_REQUIRE((int64_t)(lay->widthp[0] / lay->rb + 1) * lay->pitch * 16 < (1ll << 31), RC_EINVAL,
             "one image row of the pyramid exceeds the 32-bit lane offset range");
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

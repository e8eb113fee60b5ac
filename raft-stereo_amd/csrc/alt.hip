// On-the-fly ("alt") correlation lookup: the volume is never materialised.
//
// Replaces core/corr.py:64-107 (PytorchAlternateCorrBlock1D) of the reference, which runs 36
// full-feature grid_samples per GRU iteration and re-pools fmap2 on every call.
//
// rc_alt_prepare (once per image pair) writes channel-last copies: f1t [B,H,W1,Dp] and the
// pooled f2 pyramid f2t_i [B,H,W2>>i,Dp], so a pixel's feature vector is one contiguous run and
// MFMA operands are K-major without any transposition.
// rc_alt_fwd (every iteration): a wave owns 16 consecutive pixels of one image row.  Per level it
// takes the union of the pixels' tap windows, and for each 16-wide block of right-image
// positions in it forms the 16x16 product tile  f2t_i[block] . f1t[pixels]^T  on the matrix
// cores (K = Dp); MFMA rows are w2, so a lane holds 4 consecutive w2 of one pixel and folds them
// straight into that pixel's 2r+1 interpolated taps.  With locally smooth disparity a level
// needs 1-3 blocks; arbitrary coords stay correct, only slower.
#include "rc_common.h"

#include <math.h>
#include <string.h>
#include <type_traits>

namespace {

typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int DMAX = 256;

template <typename T> __device__ __forceinline__ f4v mfma16(s8v a, s8v b, f4v c);
template <> __device__ __forceinline__ f4v mfma16<rc_half>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f4v mfma16<rc_bf16>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

struct alt_dev_layout {
  int B, H, W1, W2, D, Dp, L;
  int width[RC_MAX_LEVELS];
  int64_t f2_offset[RC_MAX_LEVELS];
};

alt_dev_layout to_dev(const rc_alt_layout* a) {
  alt_dev_layout d;
  d.B = (int)a->B; d.H = (int)a->H; d.W1 = (int)a->W1; d.W2 = (int)a->W2;
  d.D = (int)a->D; d.Dp = (int)a->Dp; d.L = a->L;
  for (int i = 0; i < RC_MAX_LEVELS; ++i) { d.width[i] = a->width[i]; d.f2_offset[i] = a->f2_offset[i]; }
  return d;
}

// NCHW [B,D,H,W] -> channel-last [B,H,W>>level,Dp], averaging 2^level source columns (pairwise
// tree, the order of the avg_pool2d chain, corr.py:104) and zero padding D..Dp.
template <typename T>
__global__ __launch_bounds__(256) void to_channel_last_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                              int D, int Dp, int H, int W, int level) {
  __shared__ float tile[64][65];
  const int Wl = W >> level;
  const int w0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
  const int bh = blockIdx.z, b = bh / H, h = bh - b * H;
  {
    const int wc = threadIdx.x & 63, dr = threadIdx.x >> 6;
    const int w = w0 + wc;
    for (int dd = dr; dd < 64; dd += 4) {
      const int d = d0 + dd;
      float v = 0.f;
      if (d < D && w < Wl) {
        const T* p = src + (((int64_t)b * D + d) * H + h) * W + ((int64_t)w << level);
        float t[1 << (RC_MAX_LEVELS - 1)];
#pragma unroll
        for (int e = 0; e < (1 << (RC_MAX_LEVELS - 1)); ++e) t[e] = (e < (1 << level)) ? rc_to_float(p[e]) : 0.f;
        for (int n = 1 << level; n > 1; n >>= 1)
#pragma unroll
          for (int e = 0; e < (1 << (RC_MAX_LEVELS - 2)); ++e)
            if (e < n / 2) t[e] = 0.5f * (t[2 * e] + t[2 * e + 1]);
        v = t[0];
      }
      tile[dd][wc] = v;
    }
  }
  __syncthreads();
  {
    const int dc = threadIdx.x & 63, wr = threadIdx.x >> 6;
    const int d = d0 + dc;
    for (int ww = wr; ww < 64; ww += 4) {
      const int w = w0 + ww;
      if (w < Wl && d < Dp) dst[((int64_t)bh * Wl + w) * Dp + d] = rc_from_float<T>(tile[dc][ww]);
    }
  }
}

template <typename T, int R, int MODE>
__global__ __launch_bounds__(256) void alt_fwd_kernel(const T* __restrict__ f1t, const T* __restrict__ f2t,
                                                      alt_dev_layout lay, const float* __restrict__ coords,
                                                      int coords_channels, float* __restrict__ out,
                                                      float scale, int groups_per_row, int n_groups) {
  constexpr bool IS16 = !std::is_same<T, float>::value;
  constexpr int NT = 2 * R + 1;
  constexpr int KCH = IS16 ? 32 : 16;          // K elements consumed per operand load
  constexpr int NKC = DMAX / KCH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int grp = blockIdx.x * 4 + wave;        // (row, 16-pixel group)
  if (grp >= n_groups) return;                  // wave-uniform
  const int bh = grp / groups_per_row;
  const int w1 = (grp - bh * groups_per_row) * 16 + li;
  const int b = bh / lay.H, h = bh - b * lay.H;
  const bool live = w1 < lay.W1;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const int nkc = lay.Dp / KCH;

  // this pixel's feature vector as the MFMA B operand (columns = pixels)
  uint4 bfrag[NKC];
  {
    const T* p = f1t + ((int64_t)bh * lay.W1 + (live ? w1 : 0)) * lay.Dp + (IS16 ? 8 : 4) * g;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
      bfrag[kc] = (live && kc < nkc) ? *reinterpret_cast<const uint4*>(p + kc * KCH) : make_uint4(0, 0, 0, 0);
  }
  const float x = live ? coords[((int64_t)b * coords_channels) * HW + (int64_t)h * lay.W1 + w1] : 0.f;

  float inv = 1.0f;
  for (int i = 0; i < lay.L; ++i) {
    const int Wi = lay.width[i];
    const float xi = x * inv;
    inv *= 0.5f;
    int i0[NT];
    float w0[NT], wt1[NT];
    int lo = 1 << 30, hi = -(1 << 30);
    if (MODE == RC_COORD_EXACT) {
      const float xf = floorf(xi);
      const float a = xi - xf;
      const int base = rc_safe_int(xf) - R;
#pragma unroll
      for (int k = 0; k < NT; ++k) { i0[k] = base + k; w0[k] = 1.0f - a; wt1[k] = a; }
    } else {
      const float wm1 = (float)(Wi - 1);
#pragma unroll
      for (int k = 0; k < NT; ++k) rc_gridsample_coords(xi, k - R, wm1, &i0[k], &w0[k], &wt1[k]);
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < NT; ++k) { lo = min(lo, i0[k]); hi = max(hi, i0[k] + 1); }
    }
    // union of the 16 pixels' windows (all 64 lanes take part; the 4 groups hold copies)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      lo = min(lo, __shfl_xor(lo, o));
      hi = max(hi, __shfl_xor(hi, o));
    }
    lo = max(lo, 0);
    hi = min(hi, Wi - 1);

    float o_[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) o_[k] = 0.f;

    const T* f2l = f2t + lay.f2_offset[i] + ((int64_t)bh * Wi) * lay.Dp;
    for (int cb = lo >> 4; cb <= (hi >> 4); ++cb) {  // empty when lo > hi
      const int wrow = cb * 16 + li;
      const bool rok = wrow < Wi;
      const T* ap = f2l + (int64_t)(rok ? wrow : 0) * lay.Dp + (IS16 ? 8 : 4) * g;
      f4v acc = (f4v)(0.f);
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        if (kc < nkc) {
          const uint4 av = rok ? *reinterpret_cast<const uint4*>(ap + kc * KCH) : make_uint4(0, 0, 0, 0);
          if constexpr (IS16) {
            acc = mfma16<T>(__builtin_bit_cast(s8v, av), __builtin_bit_cast(s8v, bfrag[kc]), acc);
          } else {
            const f4v a4 = __builtin_bit_cast(f4v, av), b4 = __builtin_bit_cast(f4v, bfrag[kc]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[e], acc, 0, 0, 0);
          }
        }
      }
      // lane holds C[w2 = cb*16 + 4g + r][this pixel]; fold into the interpolated taps
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int w2 = cb * 16 + 4 * g + r;
        const float v = acc[r];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const float wgt = (w2 == i0[k] ? w0[k] : 0.f) + (w2 == i0[k] + 1 ? wt1[k] : 0.f);
          o_[k] = fmaf(wgt, v, o_[k]);
        }
      }
    }
    // sum the four w2 sub-ranges held by the lane groups, then spread the stores over the groups
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      float s = o_[k];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (live && (k & 3) == g) out[((int64_t)b * lay.L * NT + i * NT + k) * HW + (int64_t)h * lay.W1 + w1] = s * scale;
    }
  }
}

template <typename T, int R>
int launch_alt(const void* f1t, const void* f2t, const alt_dev_layout& d, const float* coords, int cc,
               float* out, int mode, hipStream_t st) {
  const int gpr = (d.W1 + 15) / 16;
  const int64_t ng = (int64_t)d.B * d.H * gpr;
  RC_REQUIRE(ng < (1ll << 31), RC_EINVAL, "too many pixels");
  const float scale = (float)(1.0 / sqrt((double)d.D));
  const unsigned blocks = (unsigned)((ng + 3) / 4);
  if (mode == RC_COORD_EXACT)
    hipLaunchKernelGGL((alt_fwd_kernel<T, R, RC_COORD_EXACT>), dim3(blocks), dim3(256), 0, st, (const T*)f1t,
                       (const T*)f2t, d, coords, cc, out, scale, gpr, (int)ng);
  else
    hipLaunchKernelGGL((alt_fwd_kernel<T, R, RC_COORD_GRIDSAMPLE>), dim3(blocks), dim3(256), 0, st,
                       (const T*)f1t, (const T*)f2t, d, coords, cc, out, scale, gpr, (int)ng);
  return rc_check_launch("rc_alt_fwd");
}

template <typename T>
int dispatch_alt(const void* f1t, const void* f2t, const alt_dev_layout& d, const float* coords, int cc,
                 float* out, int radius, int mode, hipStream_t st) {
  switch (radius) {
    case 1: return launch_alt<T, 1>(f1t, f2t, d, coords, cc, out, mode, st);
    case 2: return launch_alt<T, 2>(f1t, f2t, d, coords, cc, out, mode, st);
    case 3: return launch_alt<T, 3>(f1t, f2t, d, coords, cc, out, mode, st);
    case 4: return launch_alt<T, 4>(f1t, f2t, d, coords, cc, out, mode, st);
    case 5: return launch_alt<T, 5>(f1t, f2t, d, coords, cc, out, mode, st);
    case 6: return launch_alt<T, 6>(f1t, f2t, d, coords, cc, out, mode, st);
  }
  rc_set_error("alt: radius %d outside 1..6", radius);
  return RC_EUNSUPPORTED;
}

int validate_alt(const rc_alt_layout* a) {
  RC_REQUIRE(a != nullptr, RC_EINVAL, "null rc_alt_layout");
  rc_alt_layout ref;
  int rc = rc_alt_layout_init(a->B, a->D, a->H, a->W1, a->W2, a->L, &ref);
  if (rc) return rc;
  RC_REQUIRE(memcmp(&ref, a, sizeof(ref)) == 0, RC_EINVAL, "rc_alt_layout was not produced by rc_alt_layout_init");
  return RC_OK;
}

template <typename T>
int run_prepare(const void* f1, const void* f2, void* f1t, void* f2t, const rc_alt_layout* a, hipStream_t st) {
  const int D = (int)a->D, Dp = (int)a->Dp, H = (int)a->H;
  const unsigned BH = (unsigned)(a->B * a->H);
  hipLaunchKernelGGL((to_channel_last_kernel<T>), dim3((unsigned)((a->W1 + 63) / 64), (Dp + 63) / 64, BH), dim3(256),
                     0, st, (const T*)f1, (T*)f1t, D, Dp, H, (int)a->W1, 0);
  int rc = rc_check_launch("rc_alt_prepare(f1)");
  for (int i = 0; i < a->L && !rc; ++i) {
    hipLaunchKernelGGL((to_channel_last_kernel<T>), dim3((unsigned)((a->width[i] + 63) / 64), (Dp + 63) / 64, BH),
                       dim3(256), 0, st, (const T*)f2, (T*)f2t + a->f2_offset[i], D, Dp, H, (int)a->W2, i);
    rc = rc_check_launch("rc_alt_prepare(f2)");
  }
  return rc;
}

}  // namespace

extern "C" int rc_alt_layout_init(int64_t B, int64_t D, int64_t H, int64_t W1, int64_t W2, int L,
                                  rc_alt_layout* out) {
  RC_REQUIRE(out != nullptr, RC_EINVAL, "rc_alt_layout_init: null output");
  RC_REQUIRE(B > 0 && D > 0 && H > 0 && W1 > 0 && W2 > 0, RC_EINVAL, "rc_alt_layout_init: non-positive size");
  RC_REQUIRE(L >= 1 && L <= RC_MAX_LEVELS, RC_EUNSUPPORTED, "num_levels %d outside 1..%d", L, RC_MAX_LEVELS);
  RC_REQUIRE((W2 >> (L - 1)) >= 1, RC_EINVAL, "W2=%lld is too small for %d levels", (long long)W2, L);
  RC_REQUIRE(D <= DMAX, RC_EUNSUPPORTED, "alt: feature depth %lld above %d", (long long)D, DMAX);
  RC_REQUIRE(B * H < 65536, RC_EINVAL, "alt: B*H=%lld exceeds the grid z range; split the batch", (long long)(B * H));
  memset(out, 0, sizeof(*out));
  out->B = B; out->H = H; out->W1 = W1; out->W2 = W2; out->D = D; out->L = L;
  out->Dp = (D + 31) / 32 * 32;
  out->f1_elems = B * H * W1 * out->Dp;
  int64_t off = 0;
  for (int i = 0; i < L; ++i) {
    out->width[i] = (int32_t)(W2 >> i);
    out->f2_offset[i] = off;
    off += B * H * (int64_t)out->width[i] * out->Dp;
  }
  out->f2_elems = off;
  return RC_OK;
}

extern "C" int rc_alt_prepare(const void* f1, const void* f2, void* f1t, void* f2t, const rc_alt_layout* lay,
                              int in_dtype, void* stream) {
  RC_REQUIRE(f1 && f2 && f1t && f2t, RC_EINVAL, "null pointer argument");
  int rc = validate_alt(lay);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  switch (in_dtype) {
    case RC_F32: return run_prepare<float>(f1, f2, f1t, f2t, lay, st);
    case RC_F16: return run_prepare<rc_half>(f1, f2, f1t, f2t, lay, st);
    case RC_BF16: return run_prepare<rc_bf16>(f1, f2, f1t, f2t, lay, st);
  }
  rc_set_error("unknown in_dtype %d", in_dtype);
  return RC_EUNSUPPORTED;
}

extern "C" int rc_alt_fwd(const void* f1t, const void* f2t, const rc_alt_layout* lay, int in_dtype,
                          const float* coords, int coords_channels, float* out, int radius, int coord_mode,
                          void* stream) {
  RC_REQUIRE(f1t && f2t && coords && out, RC_EINVAL, "null pointer argument");
  int rc = validate_alt(lay);
  if (rc) return rc;
  RC_REQUIRE(coords_channels >= 1, RC_EINVAL, "coords must have at least one channel");
  RC_REQUIRE(coord_mode == RC_COORD_EXACT || coord_mode == RC_COORD_GRIDSAMPLE, RC_EINVAL, "bad coord_mode");
  if (coord_mode == RC_COORD_GRIDSAMPLE)
    RC_REQUIRE(lay->width[lay->L - 1] >= 2, RC_EINVAL, "grid_sample coordinates need every level at least 2 wide");
  RC_REQUIRE((uintptr_t)f1t % 16 == 0 && (uintptr_t)f2t % 16 == 0, RC_EINVAL, "working copies must be 16-byte aligned");
  const alt_dev_layout d = to_dev(lay);
  hipStream_t st = (hipStream_t)stream;
  switch (in_dtype) {
    case RC_F32: return dispatch_alt<float>(f1t, f2t, d, coords, coords_channels, out, radius, coord_mode, st);
    case RC_F16: return dispatch_alt<rc_half>(f1t, f2t, d, coords, coords_channels, out, radius, coord_mode, st);
    case RC_BF16: return dispatch_alt<rc_bf16>(f1t, f2t, d, coords, coords_channels, out, radius, coord_mode, st);
  }
  rc_set_error("unknown in_dtype %d", in_dtype);
  return RC_EUNSUPPORTED;
}

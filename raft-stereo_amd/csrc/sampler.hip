// Drop-in kernels for the reference's `corr_sampler` extension (sampler/sampler.cpp:24-51,
// sampler/sampler_kernel.cu:19-166): one level, the CALLER's row-major volume [B,H,W1,W2].
//
// The row-major layout is part of that interface: a pixel's 2r+2 taps are contiguous while neighbouring
// pixels are a whole volume row apart, so every pixel costs one or two cache lines whatever the kernel does.
// Forward: one pixel per lane (2r+2 predicated element loads, the outputs leave coalesced per channel plane);
// a quarter-wave per pixel, round 1's shape, made sixteen times the waves for the same lines (14.8 us at the
// KITTI size against 6).  Backward: the gradient volume is written exactly once, as a stream of 16-byte
// pieces (a lane owns 8 or 4 consecutive entries of the flat array, zero unless the pixel's window crosses
// them): the reference zero-fills it and then adds 2r+2 entries per pixel (sampler_kernel.cu:148, :102).
// The arithmetic follows the reference: products and sums are rounded to the volume's type
// (sampler_kernel.cu:50-56, :94-102).
#include "rc_common.h"

namespace {

constexpr int SP_THREADS = 256;
constexpr int SP_MAXTAPS = 2 * RC_SAMPLER_MAX_RADIUS + 2;

template <typename T, int R>
__global__ __launch_bounds__(SP_THREADS) void sampler_fwd_kernel(const T* __restrict__ volume,
                                                                 const float* __restrict__ coords,
                                                                 int coords_channels, T* __restrict__ corr,
                                                                 int H, int W1, int W2, int64_t n_pix) {
  const int64_t pix = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x;
  if (pix >= n_pix) return;
  const int64_t HW = (int64_t)H * W1;
  const int64_t n = pix / HW, q = pix - n * HW;  // image, pixel inside it
  const float x0 = coords[(n * coords_channels) * HW + q];
  const float fx = floorf(x0);
  const float dx = x0 - fx;
  const int base = rc_safe_int(fx) - R;
  constexpr int rd = 2 * R + 1;
  const T* row = volume + pix * W2;
  // all 2r+2 loads unconditional, from a clamped column (a load under the bounds test is a branch, and the compiler
  // then waits for every one before issuing the next: 2r+2 memory latencies per pixel for the 16-bit types)
  // (fp32: the compiler issues the conditional loads back to back by itself, and skipping the columns outside the
  // image saves their lines: 13.4 against 14.8 us at the KITTI size)
  float sv[rd + 1];
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int j = 0; j <= rd; ++j) {
      const int x1 = base + j;
      sv[j] = (unsigned)x1 < (unsigned)W2 ? rc_to_float(row[x1]) : 0.f;
    }
  } else {
    T raw[rd + 1];
#pragma unroll
    for (int j = 0; j <= rd; ++j) raw[j] = row[min(max(base + j, 0), W2 - 1)];
#pragma unroll
    for (int j = 0; j <= rd; ++j) sv[j] = (unsigned)(base + j) < (unsigned)W2 ? rc_to_float(raw[j]) : 0.f;
  }
  // corr[j] = s_j * (1 - dx) + s_{j+1} * dx, each product and the sum rounded to T
  const float wa = rc_round_as<T>(1.0f - dx), wb = rc_round_as<T>(dx);
  T* o = corr + (n * rd) * HW + q;
#pragma unroll
  for (int j = 0; j < rd; ++j)
    o[(int64_t)j * HW] = rc_from_float<T>(rc_round_as<T>(sv[j] * wa) + rc_round_as<T>(sv[j + 1] * wb));
}

// W2 % VEC == 0 (VEC entries = 16 bytes): a wave writes SP_RPW whole pixel rows, one 16-byte piece per lane and
// row (rows longer than 64 pieces take more turns); the pixel's coordinate is one address for the whole wave,
// and nothing is divided per lane (a flat piece index per lane cost two 64-bit divisions for 16 bytes of
// output: 86 us at the KITTI size against 66 for element stores)
constexpr int SP_RPW = 8;

template <typename T, int VEC>
__global__ __launch_bounds__(SP_THREADS) void sampler_bwd_vec_kernel(const float* __restrict__ coords,
                                                                     int coords_channels,
                                                                     const T* __restrict__ corr_grad,
                                                                     T* __restrict__ volume_grad, int HW, int W2,
                                                                     int r, uint32_t n_pix) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (SP_THREADS / 64) + (threadIdx.x >> 6));
  const uint32_t pix0 = wave * SP_RPW;
  if (pix0 >= n_pix) return;
  const int ppr = W2 / VEC;  // pieces per pixel row
  const int rd = 2 * r + 1;
  uint32_t n = pix0 / (uint32_t)HW, q = pix0 - n * (uint32_t)HW;  // wave-uniform
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  float xs[SP_RPW], gs[SP_RPW];
  T graw[SP_RPW];
  const int tap = lane < rd ? lane : 0;
#pragma unroll
  for (int k = 0; k < SP_RPW; ++k) {  // the rows' coordinates and gradients first: 2 SP_RPW loads in flight, all
    // unconditional (rows past the end re-read the last one; lanes past the taps read tap 0): a load under a
    // test is a branch, and the compiler waits for each one where it joins
    xs[k] = coords[((int64_t)n * coords_channels) * HW + q];
    graw[k] = corr_grad[((int64_t)n * rd + tap) * HW + q];
    if (pix0 + k + 1 < n_pix && ++q == (uint32_t)HW) {
      q = 0;
      ++n;
    }
  }
#pragma unroll
  for (int k = 0; k < SP_RPW; ++k)  // lane j < 2r+1 holds the pixel's gradient of tap j
    gs[k] = (pix0 + k < n_pix && lane < rd) ? rc_to_float(graw[k]) : 0.f;
#pragma unroll
  for (int k = 0; k < SP_RPW; ++k) {
    if (pix0 + k >= n_pix) break;  // wave-uniform
    const float fx = floorf(xs[k]);
    const float dx = xs[k] - fx;
    const int base = rc_safe_int(fx) - r;
    const float wa = rc_round_as<T>(1.0f - dx), wb = rc_round_as<T>(dx);
    // lane j <= 2r+1: what column base + j receives (sampler_kernel.cu:94-102, the same order of additions)
    const float gprev = __shfl_up(gs[k], 1);
    float acc = 0.f;
    if (lane > 0 && lane <= rd) acc = rc_round_as<T>(acc + rc_round_as<T>(gprev * wb));
    if (lane < rd) acc = rc_round_as<T>(acc + rc_round_as<T>(gs[k] * wa));
    T* row = volume_grad + (int64_t)(pix0 + k) * W2;
    for (int p0 = 0; p0 < ppr; p0 += 64) {  // every lane takes every turn: the exchanges below need lanes 0..2r+1
      const int w0 = (p0 + lane) * VEC;
      alignas(16) T v[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int j = w0 + e - base;  // tap index that lands on this column
        const float t = __shfl(acc, j & 63);
        v[e] = rc_from_float<T>((unsigned)j <= (unsigned)rd ? t : 0.f);
      }
      if (p0 + lane < ppr)
        __builtin_nontemporal_store(*reinterpret_cast<const u4*>(v), reinterpret_cast<u4*>(row + w0));
    }
  }
}

// any W2: one wave per pixel row, element stores
template <typename T>
__global__ __launch_bounds__(SP_THREADS) void sampler_bwd_kernel(const float* __restrict__ coords,
                                                                 int coords_channels,
                                                                 const T* __restrict__ corr_grad,
                                                                 T* __restrict__ volume_grad, int B, int H,
                                                                 int W1, int W2, int r, int64_t n_pix) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = ((int64_t)blockIdx.x * SP_THREADS + threadIdx.x) >> 6;
  if (pix >= n_pix) return;
  const int x = (int)(pix % W1);
  const int64_t t = pix / W1;
  const int y = (int)(t % H);
  const int n = (int)(t / H);
  const int64_t HW = (int64_t)H * W1;
  const int rd = 2 * r + 1;

  const float x0 = coords[((int64_t)n * coords_channels) * HW + (int64_t)y * W1 + x];
  const float fx = floorf(x0);
  const float dx = x0 - fx;
  const int base = rc_safe_int(fx) - r;
  const T* g = corr_grad + ((int64_t)n * rd * H + y) * W1 + x;
  T* row = volume_grad + pix * W2;
  for (int w = lane; w < W2; w += 64) {
    const int j = w - base;  // tap index that lands on column w
    float acc = 0.f;
    if (j >= 0 && j <= rd) {
      if (j > 0) acc = rc_round_as<T>(acc + rc_round_as<T>(rc_to_float(g[(int64_t)(j - 1) * HW]) * rc_round_as<T>(dx)));
      if (j < rd) acc = rc_round_as<T>(acc + rc_round_as<T>(rc_to_float(g[(int64_t)j * HW]) * rc_round_as<T>(1.0f - dx)));
    }
    row[w] = rc_from_float<T>(acc);
  }
}

int check_args(const void* a, const void* b, const void* c, int cc, int64_t B, int64_t H, int64_t W1,
               int64_t W2, int radius) {
  RC_REQUIRE(a && b && c, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(B > 0 && H > 0 && W1 > 0 && W2 > 0, RC_EINVAL, "non-positive size");
  RC_REQUIRE(cc >= 1, RC_EINVAL, "coords must have at least one channel");
  RC_REQUIRE(radius >= 1 && radius <= RC_SAMPLER_MAX_RADIUS, RC_EUNSUPPORTED, "corr_sampler: radius %d outside 1..%d", radius,
             RC_SAMPLER_MAX_RADIUS);
  RC_REQUIRE(B * H * W1 < (1ll << 40) && W2 < (1ll << 30), RC_EINVAL, "sizes exceed index range");
  return RC_OK;
}

}  // namespace

extern "C" int rc_sampler_fwd(const void* volume, const float* coords, int coords_channels, void* corr,
                              int64_t B, int64_t H, int64_t W1, int64_t W2, int radius, int dtype,
                              void* stream) {
  int rc = check_args(volume, coords, corr, coords_channels, B, H, W1, W2, radius);
  if (rc) return rc;
  const int64_t n_pix = B * H * W1;
  const unsigned blocks = (unsigned)((n_pix + SP_THREADS - 1) / SP_THREADS);
  hipStream_t st = (hipStream_t)stream;
#define RC_CASE_R(T, RR)                                                                                       \
  case RR:                                                                                                     \
    hipLaunchKernelGGL((sampler_fwd_kernel<T, RR>), dim3(blocks), dim3(SP_THREADS), 0, st, (const T*)volume,   \
                       coords, coords_channels, (T*)corr, (int)H, (int)W1, (int)W2, n_pix);                    \
    break;
#define RC_CASE(ID, T)                                                                                         \
  case ID:                                                                                                     \
    switch (radius) {                                                                                          \
      RC_CASE_R(T, 1) RC_CASE_R(T, 2) RC_CASE_R(T, 3) RC_CASE_R(T, 4) RC_CASE_R(T, 5) RC_CASE_R(T, 6)         \
      RC_CASE_R(T, 7)                                                                                          \
    }                                                                                                          \
    break;
  static_assert(RC_SAMPLER_MAX_RADIUS == 7, "one case per radius");
  switch (dtype) {
    RC_CASE(RC_F32, float)
    RC_CASE(RC_F16, rc_half)
    RC_CASE(RC_BF16, rc_bf16)
    default:
      rc_set_error("unknown dtype %d", dtype);
      return RC_EUNSUPPORTED;
  }
#undef RC_CASE
#undef RC_CASE_R
  return rc_check_launch("rc_sampler_fwd");
}

extern "C" int rc_sampler_bwd(const float* coords, int coords_channels, const void* corr_grad,
                              void* volume_grad, int64_t B, int64_t H, int64_t W1, int64_t W2, int radius,
                              int dtype, void* stream) {
  int rc = check_args(coords, corr_grad, volume_grad, coords_channels, B, H, W1, W2, radius);
  if (rc) return rc;
  const int64_t n_pix = B * H * W1;
  hipStream_t st = (hipStream_t)stream;
  const int esz = dtype == RC_F32 ? 4 : 2, vec = 16 / esz;
  const bool vector = W2 % vec == 0 && (uintptr_t)volume_grad % 16 == 0 && n_pix < (1ll << 31) && H * W1 < (1ll << 31);
  const int64_t n_waves = (n_pix + SP_RPW - 1) / SP_RPW;
  const unsigned blocks = vector ? (unsigned)((n_waves * 64 + SP_THREADS - 1) / SP_THREADS)
                                 : (unsigned)((n_pix * 64 + SP_THREADS - 1) / SP_THREADS);
#define RC_CASE(ID, T)                                                                                         \
  case ID:                                                                                                     \
    if (vector)                                                                                                \
      hipLaunchKernelGGL((sampler_bwd_vec_kernel<T, 16 / (int)sizeof(T)>), dim3(blocks), dim3(SP_THREADS), 0,  \
                         st, coords, coords_channels, (const T*)corr_grad, (T*)volume_grad, (int)(H * W1),     \
                         (int)W2, radius, (uint32_t)n_pix);                                                    \
    else                                                                                                       \
      hipLaunchKernelGGL((sampler_bwd_kernel<T>), dim3(blocks), dim3(SP_THREADS), 0, st, coords,               \
                         coords_channels, (const T*)corr_grad, (T*)volume_grad, (int)B, (int)H, (int)W1,       \
                         (int)W2, radius, n_pix);                                                              \
    break;
  switch (dtype) {
    RC_CASE(RC_F32, float)
    RC_CASE(RC_F16, rc_half)
    RC_CASE(RC_BF16, rc_bf16)
    default:
      rc_set_error("unknown dtype %d", dtype);
      return RC_EUNSUPPORTED;
  }
#undef RC_CASE
  return rc_check_launch("rc_sampler_bwd");
}

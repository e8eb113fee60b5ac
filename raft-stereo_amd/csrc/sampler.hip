// Drop-in kernels for the reference's `corr_sampler` extension (sampler/sampler.cpp:24-51,
// sampler/sampler_kernel.cu:19-166): one level, the CALLER's row-major volume [B,H,W1,W2].
//
// The row-major layout is part of that interface, so a pixel's 2r+2 taps are contiguous while
// neighbouring pixels are a whole volume row apart.  Instead of one thread per pixel (10 strided
// scalar loads each) a quarter-wave of 16 lanes serves one pixel: lane j < 2r+2 loads tap j, so
// every load instruction of the wave fetches four contiguous 2r+2-element windows, and the
// lerp pairs are exchanged with a DPP-style shuffle.  The arithmetic follows the reference:
// products and sums are rounded to the volume's type (sampler_kernel.cu:50-56, :94-102).
#include "rc_common.h"

namespace {

constexpr int SP_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(SP_THREADS) void sampler_fwd_kernel(const T* __restrict__ volume,
                                                                 const float* __restrict__ coords,
                                                                 int coords_channels, T* __restrict__ corr,
                                                                 int B, int H, int W1, int W2, int r,
                                                                 int64_t n_pix) {
  const int64_t pix = ((int64_t)blockIdx.x * SP_THREADS + threadIdx.x) >> 4;
  const int j = threadIdx.x & 15;  // tap index served by this lane
  const bool live = pix < n_pix;
  const int64_t p = live ? pix : 0;
  const int x = (int)(p % W1);
  const int64_t t = p / W1;
  const int y = (int)(t % H);
  const int n = (int)(t / H);
  const int64_t HW = (int64_t)H * W1;

  const float x0 = coords[((int64_t)n * coords_channels) * HW + (int64_t)y * W1 + x];
  const float fx = floorf(x0);
  const float dx = x0 - fx;
  const int x1 = rc_safe_int(fx) - r + j;
  const int rd = 2 * r + 1;
  float s = 0.f;
  if (live && j <= rd && (unsigned)x1 < (unsigned)W2) s = rc_to_float(volume[(p * W2) + x1]);
  // corr[j] = s_j * (1 - dx) + s_{j+1} * dx, each product and the sum rounded to T
  const float a = rc_round_as<T>(s * rc_round_as<T>(1.0f - dx));
  const float snext = __shfl_down(s, 1, 16);
  const float bq = rc_round_as<T>(snext * rc_round_as<T>(dx));
  if (live && j < rd) corr[(((int64_t)n * rd + j) * H + y) * W1 + x] = rc_from_float<T>(a + bq);
}

template <typename T>
__global__ __launch_bounds__(SP_THREADS) void sampler_bwd_kernel(const float* __restrict__ coords,
                                                                 int coords_channels,
                                                                 const T* __restrict__ corr_grad,
                                                                 T* __restrict__ volume_grad, int B, int H,
                                                                 int W1, int W2, int r, int64_t n_pix) {
  // one wave per pixel row of the volume: zero-fills the row (the reference allocates a fresh
  // zeros_like, sampler_kernel.cu:148) and drops the 2r+2 tap gradients in.
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
  const unsigned blocks = (unsigned)((n_pix * 16 + SP_THREADS - 1) / SP_THREADS);
  hipStream_t st = (hipStream_t)stream;
#define RC_CASE(ID, T)                                                                                   \
  case ID:                                                                                               \
    hipLaunchKernelGGL((sampler_fwd_kernel<T>), dim3(blocks), dim3(SP_THREADS), 0, st, (const T*)volume, \
                       coords, coords_channels, (T*)corr, (int)B, (int)H, (int)W1, (int)W2, radius, n_pix); \
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
  return rc_check_launch("rc_sampler_fwd");
}

extern "C" int rc_sampler_bwd(const float* coords, int coords_channels, const void* corr_grad,
                              void* volume_grad, int64_t B, int64_t H, int64_t W1, int64_t W2, int radius,
                              int dtype, void* stream) {
  int rc = check_args(coords, corr_grad, volume_grad, coords_channels, B, H, W1, W2, radius);
  if (rc) return rc;
  const int64_t n_pix = B * H * W1;
  const unsigned blocks = (unsigned)((n_pix * 64 + SP_THREADS - 1) / SP_THREADS);
  hipStream_t st = (hipStream_t)stream;
#define RC_CASE(ID, T)                                                                                  \
  case ID:                                                                                              \
    hipLaunchKernelGGL((sampler_bwd_kernel<T>), dim3(blocks), dim3(SP_THREADS), 0, st, coords,          \
                       coords_channels, (const T*)corr_grad, (T*)volume_grad, (int)B, (int)H, (int)W1,  \
                       (int)W2, radius, n_pix);                                                         \
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

// Experiment bench for the lookup kernel (NOT part of the library): variants of the product kernel's
// structure on a KITTI-sized f16 pyramid with the bench's coordinates, per-launch HIP-event timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/lab_lookup2 tools/lab_lookup2.hip raft-stereo_amd/csrc/core.hip
#include "../raft-stereo_amd/csrc/rc_common.h"
#include "../raft-stereo_amd/csrc/window.h"

#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

namespace {
// VAR bits: 1 = nontemporal stores, 2 = no stores (results kept live), 4 = no block loads (stores of fake taps)
//           8 = nontemporal block loads
template <int NT, int VAR>
__device__ __forceinline__ void store_taps(float* __restrict__ out, int64_t plane0, int64_t HW, int lane,
                                           const float (&tap)[NT]) {
  if constexpr (VAR & 2) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NT; ++k) acc += tap[k];
    if (acc == 12345.678f) (out + uniform(plane0))[lane] = acc;
    return;
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if constexpr (VAR & 1) __builtin_nontemporal_store(tap[k], (out + uniform(plane0 + k * HW)) + lane);
    else (out + uniform(plane0 + k * HW))[lane] = tap[k];
  }
}

template <typename VT, int R, int MODE, int VAR, int IPW>   // IPW = items (64-pixel chunks) per wave
__global__ __launch_bounds__(256) void lab_kernel(const VT* __restrict__ vol, rc_dev_layout lay,
                                                  const float* __restrict__ coords, int coords_channels,
                                                  float* __restrict__ out, int chunks_per_row, int n_items) {
  using PF = pair_fetch<VT, R, MODE>;
  constexpr int NT = 2 * R + 1, NW = PF::PI::NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item0 = (blockIdx.x * 4 + wave) * IPW;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;
  PF p0[IPW], p2[IPW];
  int64_t out_u[IPW];
  bool live[IPW];
#pragma unroll
  for (int t = 0; t < IPW; ++t) {
    const int item = min(item0 + t, n_items - 1);
    const px_item it = locate_item(item, chunks_per_row, lay.H, lay.W1, lane);
    live[t] = (item0 + t < n_items) && it.live;
    const int b = it.b;
    float x = (coords + uniform(((int64_t)b * coords_channels) * HW))[min(it.p0 + lane, (int)HW - 1)];
    out_u[t] = ((int64_t)b * 4 * NT) * HW + it.p0;
    if constexpr (VAR & 4) {
      p0[t].s_lo = p0[t].s_hi = p2[t].s_lo = p2[t].s_hi = (int)x;
      p0[t].x_lo = x;
    } else {
      p0[t].issue(vol, lay, 0, x, it.w1, it.h, b, blk_bytes);
      p2[t].issue(vol, lay, 2, x, it.w1, it.h, b, blk_bytes);
    }
  }
#pragma unroll
  for (int t = 0; t < IPW; ++t) {
    if (!live[t]) continue;
    float lo[NW], hi[NW], tap[NT];
    if constexpr (VAR & 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < NT; ++k) tap[k] = p0[t].x_lo + (float)(k + i);
        store_taps<NT, VAR>(out, out_u[t] + (int64_t)i * NT * HW, HW, lane, tap);
      }
    } else {
      p0[t].windows(lay, 0, lo, hi);
      interp_taps<R, MODE, NW>(lo, p0[t].s_lo, lay.width[0], p0[t].x_lo, p0[t].f_lo, tap);
      store_taps<NT, VAR>(out, out_u[t], HW, lane, tap);
      interp_taps<R, MODE, NW>(hi, p0[t].s_hi, lay.width[1], p0[t].x_hi, p0[t].f_hi, tap);
      store_taps<NT, VAR>(out, out_u[t] + (int64_t)NT * HW, HW, lane, tap);
      p2[t].windows(lay, 2, lo, hi);
      interp_taps<R, MODE, NW>(lo, p2[t].s_lo, lay.width[2], p2[t].x_lo, p2[t].f_lo, tap);
      store_taps<NT, VAR>(out, out_u[t] + (int64_t)2 * NT * HW, HW, lane, tap);
      interp_taps<R, MODE, NW>(hi, p2[t].s_hi, lay.width[3], p2[t].x_hi, p2[t].f_hi, tap);
      store_taps<NT, VAR>(out, out_u[t] + (int64_t)3 * NT * HW, HW, lane, tap);
    }
  }
}
}  // namespace

struct Ctx {
  void* vol;
  rc_layout lay;
  float* coords;
  float* out;
  int64_t n_pix;
};

template <int VAR, int IPW> static void launch(Ctx* c) {
  rc_dev_layout d = rc_to_dev(&c->lay);
  set_row_bias(d, 8, pair_info<8, 4, RC_COORD_EXACT>::NW2);
  const int chunks = (d.H * d.W1 + 63) / 64;
  const int n_items = (int)((int64_t)d.B * chunks);
  const int waves = (n_items + IPW - 1) / IPW;
  hipLaunchKernelGGL((lab_kernel<rc_half, 4, RC_COORD_EXACT, VAR, IPW>), dim3((waves + 3) / 4), dim3(256), 0, 0,
                     (const rc_half*)c->vol, d, c->coords, 2, c->out, chunks, n_items);
}

template <int VAR, int IPW> static void bench(Ctx* c, const char* name, double bytes) {
  const int N = 60;
  std::vector<hipEvent_t> ev(N + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < 5; ++i) launch<VAR, IPW>(c);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(ev[0], 0));
  for (int i = 0; i < N; ++i) {
    launch<VAR, IPW>(c);
    CK(hipEventRecord(ev[i + 1], 0));
  }
  CK(hipEventSynchronize(ev[N]));
  std::vector<float> us(N);
  for (int i = 0; i < N; ++i) {
    float ms;
    CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    us[i] = ms * 1e3f;
  }
  float tot;
  CK(hipEventElapsedTime(&tot, ev[0], ev[N]));
  std::vector<float> s = us;
  std::sort(s.begin(), s.end());
  printf("%-44s avg %8.2f us  min %8.2f  med %8.2f  max %8.2f   %7.1f GB/s alg\n", name, tot * 1e3 / N, s[0], s[N / 2],
         s[N - 1], bytes / (tot * 1e3 / N) / 1e3);
  for (auto& e : ev) CK(hipEventDestroy(e));
}

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 8;
  const int H = 96, W = 312, L = 4;
  Ctx c;
  if (rc_pyramid_layout(B, H, W, W, L, RC_F16, &c.lay)) { fprintf(stderr, "layout\n"); return 1; }
  c.n_pix = (int64_t)B * H * W;
  const size_t vol_bytes = (size_t)c.lay.total * 2;
  CK(hipMalloc(&c.vol, vol_bytes));
  CK(hipMemset(c.vol, 0x3c, vol_bytes));  // halves 0x3c3c ~ 1.06
  std::vector<float> hc((size_t)B * 2 * H * W);
  srand(1);
  for (int b = 0; b < B; ++b) {
    float grid[8][8];
    for (auto& r : grid) for (float& g : r) g = (float)rand() / RAND_MAX * (W / 3.0f);
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        const float gy = 7.0f * h / (H - 1), gx = 7.0f * w / (W - 1);
        const int y0 = fminf(floorf(gy), 6), x0 = fminf(floorf(gx), 6);
        const float fy = gy - y0, fx = gx - x0;
        float d = grid[y0][x0] * (1 - fy) * (1 - fx) + grid[y0][x0 + 1] * (1 - fy) * fx +
                  grid[y0 + 1][x0] * fy * (1 - fx) + grid[y0 + 1][x0 + 1] * fy * fx;
        d += (float)rand() / RAND_MAX - 0.5f;
        d = fminf(fmaxf(d, 0.f), W / 2.0f);
        hc[(((size_t)b * 2 + 0) * H + h) * W + w] = w - d;
        hc[(((size_t)b * 2 + 1) * H + h) * W + w] = h;
      }
  }
  CK(hipMalloc(&c.coords, hc.size() * 4));
  CK(hipMemcpy(c.coords, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&c.out, (size_t)c.n_pix * 36 * 4));
  const double alg = (double)c.n_pix * 228;
  bench<0, 1>(&c, "product structure", alg);
  bench<1, 1>(&c, "nontemporal stores", alg);
  bench<2, 1>(&c, "no stores (loads + decode)", alg);
  bench<4, 1>(&c, "no block loads (coords + stores)", alg);
  bench<0, 1>(&c, "product structure (again)", alg);
  return 0;
}

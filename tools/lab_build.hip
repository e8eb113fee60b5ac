// Experiment bench for the 16-bit build kernel (NOT part of the library): times rc_build_pyramid on a
// KITTI-sized f16 batch; built several times with -DRC_LAB_ABLATE=n to see which part bounds it.
//   for n in 0 1 2 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRC_LAB_ABLATE=$n -o tools/lab_build_$n \
//        tools/lab_build.hip raft-stereo_amd/csrc/build.hip raft-stereo_amd/csrc/core.hip; done
#include "../raft-stereo_amd/csrc/rc_common.h"

#ifndef RC_LAB_ABLATE
#define RC_LAB_ABLATE 0
#endif

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

// reads `n` 16-byte pieces and keeps nothing: empties the caches of the build's data without leaving dirty lines
__global__ void lab_sweep(const uint4* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

#ifdef RC_LAB_STAMPS
extern "C" int rc_lab_read_stamps(unsigned long long*, size_t);
#endif

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8;
  const int H = 96, W = argc > 2 ? atoi(argv[2]) : 312, D = 256, L = 4;
  rc_layout lay;
  if (rc_pyramid_layout(B, H, W, W, L, RC_F16, &lay)) return 1;
  const size_t fe = (size_t)B * D * H * W;
  std::vector<uint16_t> hf(fe);
  srand(3);
  for (auto& v : hf) v = (uint16_t)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));  // random halves ~ +-[0.1, 1)
  void *f1, *f2, *vol;
  CK(hipMalloc(&f1, fe * 2));
  CK(hipMalloc(&f2, fe * 2));
  CK(hipMalloc(&vol, (size_t)lay.total * 2));
  CK(hipMemcpy(f1, hf.data(), fe * 2, hipMemcpyHostToDevice));
  std::reverse(hf.begin(), hf.end());
  CK(hipMemcpy(f2, hf.data(), fe * 2, hipMemcpyHostToDevice));
  const int N = 30;
  // RC_LAB_COLD=1: a 1 GiB fill between the launches, so that every build finds the caches as it does inside a
  // step (its inputs were produced long ago, 32 lookups ran in between), not as the previous build left them
  // RC_LAB_COLD=2: a read-only sweep of 1 GiB instead (cold caches, no dirty lines: what a build finds inside a step)
  const int cold = getenv("RC_LAB_COLD") ? atoi(getenv("RC_LAB_COLD")) : 0;
  void* flush = nullptr;
  if (cold) {
    CK(hipMalloc(&flush, ((size_t)1 << 30) + 64));
    CK(hipMemset(flush, 1, ((size_t)1 << 30) + 64));
  }
  std::vector<hipEvent_t> ev(2 * N + 2);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < 3; ++i)
    if (rc_build_pyramid(f1, f2, vol, &lay, D, RC_F16, 0)) { fprintf(stderr, "%s\n", rc_last_error_string()); return 1; }
  CK(hipDeviceSynchronize());
  for (int i = 0; i < N; ++i) {
    if (cold == 1) CK(hipMemsetAsync(flush, i, (size_t)1 << 30, 0));
    if (cold == 2) lab_sweep<<<2048, 256>>>((const uint4*)flush, ((size_t)1 << 30) / 16, (uint32_t*)((char*)flush + ((size_t)1 << 30)));
    CK(hipEventRecord(ev[2 * i], 0));
    rc_build_pyramid(f1, f2, vol, &lay, D, RC_F16, 0);
    CK(hipEventRecord(ev[2 * i + 1], 0));
  }
  CK(hipEventSynchronize(ev[2 * N - 1]));
  std::vector<float> us(N);
  for (int i = 0; i < N; ++i) {
    float ms;
    CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    us[i] = ms * 1e3f;
  }
  std::sort(us.begin(), us.end());
  const double alg = 2.0 * fe * 2 + (double)B * H * W * W * 2 * 1.25, flop = 2.0 * B * H * (double)W * W * D;
  printf("%sablate=%d B=%d W=%d: min %.1f med %.1f max %.1f us   %.1f GB/s alg, %.1f TFLOP/s (median)\n", cold == 1 ? "cold+dirty " : cold == 2 ? "cold " : "", RC_LAB_ABLATE, B, W,
         us[0], us[N / 2], us[N - 1], alg / us[N / 2] / 1e3, flop / us[N / 2] / 1e6);
#ifdef RC_LAB_STAMPS
  {
    // phase durations (shader cycles) of the last launch, per wave: median over the workgroups' waves 0 and 7
    const int NS = 32, n_wg = std::min(4096, (B * H + 7) / 8 * 8 * ((W + 127) / 128));
    std::vector<unsigned long long> st((size_t)4096 * 8 * NS);
    if (rc_lab_read_stamps(st.data(), st.size())) return 1;
    const int n_chunks = (W + 63) / 64;
    auto med = [&](int a, int b, int wave) {
      std::vector<double> d;
      for (int g = 0; g < n_wg; ++g) {
        const unsigned long long* p = &st[((size_t)g * 8 + wave) * NS];
        if (p[a] && p[b] && p[b] >= p[a]) d.push_back((double)(p[b] - p[a]));
      }
      if (d.empty()) return -1.0;
      std::sort(d.begin(), d.end());
      return d[d.size() / 2];
    };
    // every stamped slot in order; the median time from the previous stamped slot is printed under the slot's number
    // (slot meanings: see the RC_STAMP calls of the kernel that ran: build16_kernel or build16g_kernel in build.hip)
    for (int wave : {0, 3, 4, 7}) {
      printf("wave %d, cycles from the previous stamp (median over %d workgroups):", wave, n_wg);
      int prev = -1;
      for (int sl = 0; sl < NS; ++sl) {
        bool any = false;
        for (int g = 0; g < n_wg && !any; ++g) any = st[((size_t)g * 8 + wave) * NS + sl] != 0;
        if (!any) continue;
        if (prev >= 0) printf("  [%d] %.0f", sl, med(prev, sl, wave));
        prev = sl;
      }
      printf("   | lifetime %.0f\n", med(0, prev, wave));
    }
  }
#endif
  return 0;
}

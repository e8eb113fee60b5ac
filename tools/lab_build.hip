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
  std::vector<hipEvent_t> ev(N + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < 3; ++i)
    if (rc_build_pyramid(f1, f2, vol, &lay, D, RC_F16, 0)) { fprintf(stderr, "%s\n", rc_last_error_string()); return 1; }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(ev[0], 0));
  for (int i = 0; i < N; ++i) {
    rc_build_pyramid(f1, f2, vol, &lay, D, RC_F16, 0);
    CK(hipEventRecord(ev[i + 1], 0));
  }
  CK(hipEventSynchronize(ev[N]));
  std::vector<float> us(N);
  for (int i = 0; i < N; ++i) {
    float ms;
    CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    us[i] = ms * 1e3f;
  }
  std::sort(us.begin(), us.end());
  const double alg = 2.0 * fe * 2 + (double)B * H * W * W * 2 * 1.875, flop = 2.0 * B * H * (double)W * W * D;
  printf("ablate=%d B=%d W=%d: min %.1f med %.1f max %.1f us   %.1f GB/s alg, %.1f TFLOP/s (median)\n", RC_LAB_ABLATE, B, W,
         us[0], us[N / 2], us[N - 1], alg / us[N / 2] / 1e3, flop / us[N / 2] / 1e6);
  return 0;
}

// lab: how fast can a kernel write channel planes [B][C][HW] in pieces of different sizes?  (the fused
// lookup + convc1 kernel writes 64 planes x 128 B per wave; the plain lookup 36 planes x 256 B)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/lab_store_patterns tools/lab_store_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// PIECE: bytes a wave (or a block of 4 waves, COOP) writes contiguously into one plane per visit.
// C planes of HW2 bytes each per image.  A wave owns PIECE bytes of every plane (64 px x 2 B = 128 ...).
template <int PIECE, bool NT>
__global__ __launch_bounds__(256) void planes_kernel(unsigned char* out, int C, int64_t plane_bytes, int pieces_per_image,
                                                     int n_items) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= n_items) return;
  const int b = item / pieces_per_image, p = item - b * pieces_per_image;
  unsigned char* base = out + ((int64_t)b * C) * plane_bytes + (int64_t)p * PIECE;
  constexpr int LPP = PIECE / 16;      // lanes per plane per instruction
  constexpr int PPI = 64 / LPP;        // planes per instruction
  const u4 v = {(uint32_t)item, (uint32_t)lane, 1u, 2u};
  for (int c0 = 0; c0 < C; c0 += PPI) {
    const int c = c0 + lane / LPP;
    if (c < C && (int64_t)p * PIECE + (lane % LPP) * 16 < plane_bytes) {
      u4* dst = reinterpret_cast<u4*>(base + (int64_t)c * plane_bytes + (lane % LPP) * 16);
      if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
  }
}

// the plain lookup's pattern: one plane per instruction, 64 lanes x 4 B
template <bool NT>
__global__ __launch_bounds__(256) void dword_kernel(float* out, int C, int64_t plane_elems, int pieces_per_image, int n_items) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= n_items) return;
  const int b = item / pieces_per_image, p = item - b * pieces_per_image;
  float* base = out + ((int64_t)b * C) * plane_elems + (int64_t)p * 64 + lane;
  if ((int64_t)p * 64 + lane >= plane_elems) return;
  for (int c = 0; c < C; ++c) {
    if (NT) __builtin_nontemporal_store((float)item, base + (int64_t)c * plane_elems); else base[(int64_t)c * plane_elems] = (float)item;
  }
}

template <typename F> void timeit(const char* name, double bytes, F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  std::vector<float> t;
  for (int i = 0; i < 30; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f); }
  std::sort(t.begin(), t.end());
  printf("%-52s med %8.2f us  min %8.2f   %7.1f GB/s\n", name, t[15], t[0], bytes / (t[15] * 1e-6) / 1e9);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const int HW = 94 * 312;  // KITTI at 1/4
  unsigned char* buf; CK(hipMalloc(&buf, (size_t)B * 64 * HW * 4));
  CK(hipMemset(buf, 0, (size_t)B * 64 * HW * 4));
  printf("B = %d pairs, HW = %d\n", B, HW);
  {
    const int64_t pb = (int64_t)HW * 2; const double bytes = (double)B * 64 * pb;
#define RUN(PIECE, NT) { const int ppi = (int)((pb + PIECE - 1) / PIECE), n = B * ppi; \
      timeit("16-bit 64 planes, piece " #PIECE " B" #NT, bytes, [&] { hipLaunchKernelGGL((planes_kernel<PIECE, NT>), dim3((n + 3) / 4), dim3(256), 0, 0, buf, 64, pb, ppi, n); }); }
    RUN(128, false) RUN(128, true) RUN(256, false) RUN(256, true) RUN(512, false) RUN(512, true) RUN(1024, false) RUN(1024, true)
  }
  {
    const double bytes = (double)B * 36 * HW * 4; const int ppi = (HW + 63) / 64, n = B * ppi;
    timeit("f32 36 planes, 256 B pieces (lookup)", bytes, [&] { hipLaunchKernelGGL((dword_kernel<false>), dim3((n + 3) / 4), dim3(256), 0, 0, (float*)buf, 36, (int64_t)HW, ppi, n); });
    timeit("f32 36 planes, 256 B pieces (lookup) nt", bytes, [&] { hipLaunchKernelGGL((dword_kernel<true>), dim3((n + 3) / 4), dim3(256), 0, 0, (float*)buf, 36, (int64_t)HW, ppi, n); });
  }
  return 0;
}

// Lab: how fast can the fused adjoint's INPUT be read at all?  Its input is n = 22 tensors of tap gradients,
// [B][36 planes][H][W] fp32 each (16.6 MB at B 8, H 80, W 180: 365 MB together, more than the Infinity Cache).
// The kernel's workgroup owns TP pixels of one image row and reads, lookup after lookup, TP * 4 bytes of each
// of the 36 planes: pieces of 128 bytes (TP 32), 57.6 KB apart, not line-aligned (a row is 720 bytes).  This
// program reads exactly those bytes with nothing else going on, in several shapes of workgroup and orders of
// launch, with the same number of loads in flight per lane (4 lookups x 9 planes) and the same LDS per
// workgroup as the kernel (so the same number of workgroups share a CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <unistd.h>
static int g_gap_us = 0;

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int NL = 22, NPL = 36, NT = 9;
struct lists {
  const float* g[NL];
};

// ORDER 0: workgroup id -> (row, tile), tiles of a row next to each other (the kernel's order)
// ORDER 1: XCD-aware: the 8 XCDs (id % 8) each take a contiguous eighth of the (row, tile) list
// ORDER 2: like 0 but tile-major: all rows of tile 0, then tile 1, ...
template <int TP, int ORDER>
__global__ __launch_bounds__(4 * TP) void read_planes(lists ls, int B, int H, int W, float* sink) {
  extern __shared__ float lds[];
  const int tiles = (W + TP - 1) / TP;
  const int nblk = B * H * tiles;
  int id = blockIdx.x;
  if (ORDER == 1) {
    const int per = (nblk + 7) / 8;
    id = (id % 8) * per + id / 8;
    if (id >= nblk) return;
  }
  int bh, tile;
  if (ORDER == 2) {
    tile = id / (B * H);
    bh = id - tile * (B * H);
  } else {
    bh = id / tiles;
    tile = id - bh * tiles;
  }
  const int b = bh / H, h = bh - b * H;
  const int p = threadIdx.x % TP, lvl = threadIdx.x / TP;
  const int w = tile * TP + p;
  const size_t HW = (size_t)H * W;
  float acc = 0.f;
  if (w < W) {
    const size_t base = ((size_t)b * NPL + (size_t)lvl * NT) * HW + (size_t)h * W + w;
    float v[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < NT; ++k) v[j][k] = ls.g[j][base + k * HW];
    for (int t0 = 0; t0 < NL; t0 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < NT; ++k) acc += v[j][k];
        const int t = min(t0 + 4 + j, NL - 1);
#pragma unroll
        for (int k = 0; k < NT; ++k) v[j][k] = ls.g[t][base + k * HW];
      }
    }
  }
  if (acc == 12345.678f) sink[threadIdx.x] = acc + lds[threadIdx.x];
}

__global__ void sweep(const uint4* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int TP, int ORDER>
static void run(const lists& ls, int B, int H, int W, size_t lds_bytes, const uint4* flush, float* sink,
                const char* what) {
  const int tiles = (W + TP - 1) / TP;
  const int nblk = B * H * tiles;
  const int grid = ORDER == 1 ? (nblk + 7) / 8 * 8 : nblk;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&read_planes<TP, ORDER>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
  const int N = 9;
  std::vector<hipEvent_t> ev(2 * N);
  for (auto& e : ev) CHECK(hipEventCreate(&e));
  for (int i = 0; i < N; ++i) {
    sweep<<<2048, 256>>>(flush, ((size_t)1 << 30) / 16, (uint32_t*)sink);
    if (g_gap_us > 0) {  // the GPU idles before the launch, as it does in an eager training step
      CHECK(hipDeviceSynchronize());
      usleep(g_gap_us);
    }
    CHECK(hipEventRecord(ev[2 * i]));
    read_planes<TP, ORDER><<<grid, 4 * TP, lds_bytes>>>(ls, B, H, W, sink);
    CHECK(hipEventRecord(ev[2 * i + 1]));
  }
  CHECK(hipEventSynchronize(ev[2 * N - 1]));
  std::vector<float> us(N);
  for (int i = 0; i < N; ++i) {
    float ms;
    CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    us[i] = ms * 1e3f;
  }
  std::sort(us.begin(), us.end());
  const double bytes = (double)NL * B * NPL * H * W * 4;
  printf("%-64s %5d wg, %3zu KB LDS  median %7.1f us  %7.1f GB/s\n", what, nblk, lds_bytes >> 10, us[N / 2],
         bytes / us[N / 2] / 1e3);
}

int main(int argc, char** argv) {
  const int B = 8, H = 80, W = argc > 1 ? atoi(argv[1]) : 180;
  const size_t n = (size_t)B * NPL * H * W;
  // the tensors are carved from one allocation, `spacing` bytes apart (argv[2]; default: the size rounded up to
  // 2 MiB, which is how torch's caching allocator places tensors of this size) plus `skew` bytes per tensor (argv[3])
  const size_t two_mb = (size_t)2 << 20;
  const size_t spacing = argc > 2 && atol(argv[2]) > 0 ? (size_t)atol(argv[2]) : (n * 4 + two_mb - 1) / two_mb * two_mb;
  const size_t skew = argc > 3 ? (size_t)atol(argv[3]) : 0;
  g_gap_us = argc > 4 ? atoi(argv[4]) : 0;
  printf("idle gap before each launch: %d us\n", g_gap_us);
  char* pool;
  CHECK(hipMalloc(&pool, NL * (spacing + skew) + two_mb));
  CHECK(hipMemset(pool, 0, NL * (spacing + skew) + two_mb));
  lists ls;
  for (int t = 0; t < NL; ++t) ls.g[t] = reinterpret_cast<const float*>(pool + t * (spacing + skew));
  printf("tensors %zu bytes apart (size %zu)\n", spacing + skew, n * 4);
  uint4* flush;
  float* sink;
  CHECK(hipMalloc(&flush, (size_t)1 << 30));
  CHECK(hipMemset(flush, 2, (size_t)1 << 30));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipDeviceSynchronize());
  printf("W = %d (a row of a plane is %d bytes)\n", W, W * 4);
  run<32, 0>(ls, B, H, W, 52 << 10, flush, sink, "32 pixels per workgroup (128-byte pieces), the kernel's order");
  run<32, 1>(ls, B, H, W, 52 << 10, flush, sink, "32 pixels, each XCD a contiguous eighth of the rows");
  CHECK(hipDeviceSynchronize());
  return 0;
}

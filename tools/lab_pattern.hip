// Lab: how fast can the build's INPUT access pattern be read at all?  The feature maps are NCHW, so one image
// row's [256 k][W] slice is 256 pieces of W * 2 bytes, one per k plane, 60 KB apart; a build workgroup reads them
// in pieces of 128 bytes (a 64-column chunk of f2) or 256 bytes (its 128-column tile of f1).  This kernel reads the
// same pieces in the same order with nothing else going on (no LDS, no MFMA, no stores): one wave per (image row,
// piece width P bytes, range of columns), 8 k rows x 128 bytes per load instruction as the build's DMA does, 16
// loads in flight per lane.  Between launches a read-only sweep of 1 GiB empties the caches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void sweep(const uint4* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// One workgroup (WAVES waves) reads the [256 k][CW columns] slice of image row `row` that starts at column c0:
// CW * 2 bytes per k row, in 128-byte pieces (8 lanes x 16 bytes), 8 k rows per wave instruction.
// ORDER 0: the build's XCD-aware order (XCD x takes the rows = x mod 8); 1: XCD x takes a contiguous eighth of
// the rows (neighbouring rows, which share a line where one ends and the next begins, meet in one L2)
template <int CW, int ORDER>
__global__ __launch_bounds__(512) void read_slices(const uint16_t* __restrict__ f, int H, int W, int n_rows,
                                                   int tiles, uint32_t* sink) {
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int row = ORDER == 0 ? (j / tiles) * 8 + (id & 7) : (id & 7) * ((n_rows + 7) / 8) + j / tiles;
  const int tile = j % tiles;
  if (row >= n_rows) return;
  const int b = row / H, h = row - b * H;
  const int c0 = tile * CW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kr = lane >> 3, seg = lane & 7;
  const size_t ks = (size_t)H * W;
  const uint16_t* base = f + ((size_t)b * 256 * H + h) * W;
  uint32_t acc = 0;
  constexpr int PIECES = CW / 64;  // 128-byte pieces per k row
  // 32 groups of 8 k rows; wave w takes groups w, w + 8, ...
  uint4 v[4 * PIECES];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int k = (p * 8 + wave) * 8 + kr;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
      int col = c0 + q * 64 + seg * 8;
      col = col >= W ? 0 : col;
      v[p * PIECES + q] = *reinterpret_cast<const uint4*>(base + (size_t)k * ks + col);
    }
  }
#pragma unroll
  for (int e = 0; e < 4 * PIECES; ++e) acc += v[e].x ^ v[e].y ^ v[e].z ^ v[e].w;
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int CW, int ORDER = 0>
static double run(const uint16_t* f, int B, int H, int W, const uint4* flush, uint32_t* sink, const char* what) {
  const int n_rows = B * H, tiles = (W + CW - 1) / CW;
  const int blocks = (n_rows + 7) / 8 * 8 * tiles;
  const int N = 15;
  std::vector<hipEvent_t> ev(2 * N);
  for (auto& e : ev) CHECK(hipEventCreate(&e));
  for (int i = 0; i < N; ++i) {
    sweep<<<2048, 256>>>(flush, ((size_t)1 << 30) / 16, sink);
    CHECK(hipEventRecord(ev[2 * i]));
    read_slices<CW, ORDER><<<blocks, 512>>>(f, H, W, n_rows, tiles, sink);
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
  const double bytes = (double)B * 256 * H * W * 2;
  printf("%-44s %4d blocks  median %7.1f us  %7.1f GB/s of the tensor's %.1f MB\n", what, blocks, us[N / 2],
         bytes / us[N / 2] / 1e3, bytes / 1e6);
  return us[N / 2];
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = 96, W = 312;
  const size_t n = (size_t)B * 256 * H * W;
  uint16_t* f;
  uint4* flush;
  uint32_t* sink;
  CHECK(hipMalloc(&f, n * 2 + 4096));
  CHECK(hipMemset(f, 1, n * 2 + 4096));
  CHECK(hipMalloc(&flush, (size_t)1 << 30));
  CHECK(hipMemset(flush, 2, (size_t)1 << 30));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipDeviceSynchronize());
  run<64>(f, B, H, W, flush, sink, "128-byte pieces (an f2 chunk per workgroup)");
  run<128>(f, B, H, W, flush, sink, "256-byte pieces (the build's f1 tile)");
  run<320>(f, B, H, W, flush, sink, "whole 624-byte rows (one workgroup per image row)");
  run<64, 1>(f, B, H, W, flush, sink, "128-byte pieces, XCD-contiguous rows");
  run<128, 1>(f, B, H, W, flush, sink, "256-byte pieces, XCD-contiguous rows");
  run<320, 1>(f, B, H, W, flush, sink, "whole rows, XCD-contiguous rows");
  CHECK(hipDeviceSynchronize());
  return 0;
}

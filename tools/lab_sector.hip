// Lab: what does a partial read of a 128-byte line cost on the HBM side of MI355X?
// Every lane reads 16 bytes; groups of (BYTES / 16) lanes read the first BYTES bytes of consecutive 128-byte
// lines of a buffer far larger than the Infinity Cache.  If the memory side fetches 64-byte pieces, reading
// 64 of every 128 bytes takes half the time per LINE of a full read; if it fetches whole lines, the same time.
// Built here (tools/lab_sector.sh), run on the GPU box, alone and under rocprofv3 --pmc (TCC_EA0_RDREQ_sum,
// TCC_BUBBLE_sum, TCC_EA0_RDREQ_DRAM_32B_sum) to calibrate what the counters tally for partial lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// BYTES of each line are read, starting at byte OFS of the line (OFS + BYTES <= 128)
template <int BYTES, int OFS>
__global__ __launch_bounds__(256) void read_part(const uint4* __restrict__ buf, uint32_t* __restrict__ sink,
                                                 uint32_t n_lines) {
  constexpr int LPL = BYTES / 16;  // lanes per line
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  uint32_t acc = 0;
  // 8 loads in flight per lane, each over the whole grid's stride
  const uint32_t stride = gridDim.x * 256u;
#pragma unroll 1
  for (uint32_t base = g; base / LPL < n_lines; base += 8 * stride) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t t = base + u * stride;
      const uint32_t line = t / LPL, within = t % LPL;
      v[u] = line < n_lines ? buf[(size_t)line * 8 + OFS / 16 + within] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[g & 1023] = acc;  // never true in practice; keeps the loads
}

template <int BYTES, int OFS>
static void run(const uint4* buf, uint32_t* sink, uint32_t n_lines, const char* what) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const int blocks = 256 * 8;
  for (int rep = 0; rep < 2; ++rep) read_part<BYTES, OFS><<<blocks, 256>>>(buf, sink, n_lines);
  CHECK(hipEventRecord(a));
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) read_part<BYTES, OFS><<<blocks, 256>>>(buf, sink, n_lines);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  printf("%-34s %8.1f us  lines/us %9.1f  useful %7.1f GB/s  whole lines %7.1f GB/s\n", what, ms * 1e3,
         n_lines / (ms * 1e3), (double)n_lines * BYTES / (ms * 1e-3) / 1e9, (double)n_lines * 128 / (ms * 1e-3) / 1e9);
}

int main() {
  const size_t bytes = (size_t)2 << 30;  // 2 GiB: nothing survives in the 256 MiB Infinity Cache between passes
  const uint32_t n_lines = (uint32_t)(bytes / 128);
  uint4* buf;
  uint32_t* sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  run<128, 0>(buf, sink, n_lines, "128 of 128 bytes");
  run<64, 0>(buf, sink, n_lines, "first 64 of 128 bytes");
  run<64, 64>(buf, sink, n_lines, "second 64 of 128 bytes");
  run<32, 0>(buf, sink, n_lines, "first 32 of 128 bytes");
  run<16, 0>(buf, sink, n_lines, "first 16 of 128 bytes");
  run<16, 48>(buf, sink, n_lines, "bytes 48..63 of 128");
  CHECK(hipDeviceSynchronize());
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}

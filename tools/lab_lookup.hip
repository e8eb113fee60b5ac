// Experiment bench for the lookup kernel (NOT part of the library): times the product entry point
// and plain streaming kernels with the same byte mix on a KITTI-sized f16 pyramid to see which part bounds it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/lab_lookup tools/lab_lookup.hip \
//         raft-stereo_amd/csrc/core.hip raft-stereo_amd/csrc/lookup.hip
#include "../raft-stereo_amd/csrc/rc_common.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// ---- streaming kernels with the lookup's byte mix: per pixel RDB bytes read as 16-byte pieces,
// 36 dword plane stores
template <int NLOAD, int NSTORE>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, float* __restrict__ out,
                                                     int64_t n_pix, int64_t src_pix_stride16, int64_t plane) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pix) return;
  uint4 q[NLOAD > 0 ? NLOAD : 1];
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) q[k] = src[p + k * src_pix_stride16];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) acc += __uint_as_float(q[k].x ^ q[k].y ^ q[k].z ^ q[k].w);
  if (NSTORE == 0) {
    if (acc == 12345.678f) out[p] = acc;
  }
#pragma unroll
  for (int k = 0; k < NSTORE; ++k) out[p + k * plane] = acc + (float)k;
}

// store-shape experiments: the same 144 bytes per pixel written in different instruction shapes
// MODE 0: 36 dword stores (plane-major, 256 B per wave instruction)
// MODE 1: 9 dwordx4 stores, 1 KiB contiguous per wave instruction
// MODE 2: 9 dwordx4 stores, lane (q,j) writes pixels 4q..4q+3 of plane 4t+j: four 256-B runs per instruction
// MODE 3: 36 dword non-temporal stores
// MODE 4: 18 dwordx2 stores, 512 B contiguous per wave instruction
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, int64_t n_pix, int64_t plane) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pix) return;
  const float acc = (float)threadIdx.x;
  if (MODE == 0) {
#pragma unroll
    for (int k = 0; k < 36; ++k) out[p + k * plane] = acc + (float)k;
  } else if (MODE == 3) {
#pragma unroll
    for (int k = 0; k < 36; ++k) __builtin_nontemporal_store(acc + (float)k, out + p + k * plane);
  } else if (MODE == 1) {
    // wave w of the grid owns pixels [64 w', 64 w'+64) of 4 planes at a time -> emulate: thread writes
    // float4 at (plane group k, its own 4-pixel slot): 9 instructions, each 1 KiB contiguous
    const int64_t wave0 = p & ~(int64_t)63;
    const int lane = (int)(p & 63);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      // planes 4k..4k+3; lane -> (plane 4k + lane/16, pixels wave0 + 4*(lane%16))
      float4 v = make_float4(acc, acc + 1, acc + 2, acc + (float)k);
      *reinterpret_cast<float4*>(out + (int64_t)(4 * k + (lane >> 4)) * plane + wave0 + 4 * (lane & 15)) = v;
    }
  } else if (MODE == 2) {
    const int64_t wave0 = p & ~(int64_t)63;
    const int lane = (int)(p & 63);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float4 v = make_float4(acc, acc + 1, acc + 2, acc + (float)k);
      *reinterpret_cast<float4*>(out + (int64_t)(4 * k + (lane & 3)) * plane + wave0 + 4 * (lane >> 2)) = v;
    }
  } else if (MODE == 4) {
    const int64_t wave0 = p & ~(int64_t)63;
    const int lane = (int)(p & 63);
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      float2 v = make_float2(acc, acc + (float)k);
      *reinterpret_cast<float2*>(out + (int64_t)(2 * k + (lane >> 5)) * plane + wave0 + 2 * (lane & 31)) = v;
    }
  }
}
template <int MODE> static void launch_store(void* c_);

static double time_launches(int n, void (*launch)(void*), void* ctx) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) launch(ctx);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < n; ++i) launch(ctx);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / n;
}

struct Ctx {
  void* vol;
  rc_layout lay;
  float* coords;
  float* out;
  int64_t n_pix;
  int variant;
};

static void launch_product(void* c_) {
  Ctx* c = (Ctx*)c_;
  int rc = rc_lookup_fwd(c->vol, &c->lay, c->coords, 2, c->out, 4, RC_COORD_EXACT, 0);
  if (rc) { fprintf(stderr, "rc_lookup_fwd: %s\n", rc_last_error_string()); exit(1); }
}
template <int NL, int NS> static void launch_stream(void* c_) {
  Ctx* c = (Ctx*)c_;
  hipLaunchKernelGGL((stream_kernel<NL, NS>), dim3((unsigned)((c->n_pix + 255) / 256)), dim3(256), 0, 0,
                     (const uint4*)c->vol, c->out, c->n_pix, c->n_pix, c->n_pix);
}

template <int MODE> static void launch_store(void* c_) {
  Ctx* c = (Ctx*)c_;
  hipLaunchKernelGGL((store_kernel<MODE>), dim3((unsigned)((c->n_pix + 255) / 256)), dim3(256), 0, 0, c->out,
                     c->n_pix, c->n_pix);
}

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 8;
  const int H = 96, W = 312, L = 4, R = 4;
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
  const int N = 200;
  auto report = [&](const char* name, double us, double bytes) {
    printf("%-44s %8.2f us  %7.1f GB/s (of %.1f MB)\n", name, us, bytes / us / 1e3, bytes / 1e6);
  };
  report("product rc_lookup_fwd", time_launches(N, launch_product, &c), alg);
  report("stream: 8x16B loads + 36 dword stores", time_launches(N, launch_stream<8, 36>, &c), (double)c.n_pix * (128 + 144));
  report("stream: 8x16B loads only", time_launches(N, launch_stream<8, 0>, &c), (double)c.n_pix * 128);
  report("stream: 36 dword stores only", time_launches(N, launch_stream<0, 36>, &c), (double)c.n_pix * 144);
  report("stream: 10x16B loads + 36 dword stores", time_launches(N, launch_stream<10, 36>, &c), (double)c.n_pix * (160 + 144));
  report("stores: 36 x dword", time_launches(N, launch_store<0>, &c), (double)c.n_pix * 144);
  report("stores: 9 x dwordx4, 1 KiB runs", time_launches(N, launch_store<1>, &c), (double)c.n_pix * 144);
  report("stores: 9 x dwordx4, 4 x 256 B runs", time_launches(N, launch_store<2>, &c), (double)c.n_pix * 144);
  report("stores: 36 x dword nontemporal", time_launches(N, launch_store<3>, &c), (double)c.n_pix * 144);
  report("stores: 18 x dwordx2, 512 B runs", time_launches(N, launch_store<4>, &c), (double)c.n_pix * 144);
  return 0;
}

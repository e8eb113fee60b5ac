// Adjoint of the fused lookup + convc1 + ReLU (SURVEY.md section 8 f, rank 1, training):
//   act = relu(W corr + b),  corr = lookup(coords)            core/raft_stereo.py:110 -> core/update.py:71,78
// autograd of those lines gives, with gm = grad_act where act > 0 (else 0):
//   grad_corr[k] = sum_o W[o][k] gm[o]     (then into the lookups' fused adjoint, rc_lookup_bwd_dense)
//   grad_W[o][k] = sum_pixels gm[o] corr[k],   grad_b[o] = sum_pixels gm[o]
// The forward never stored the L*(2r+1)-channel tensor; the caller recomputes it with rc_lookup_fwd right
// before this call (one lookup kernel; no 36-channel tensor is kept alive across the GRU iterations).
//
// convc1_bwd_taps_mfma_kernel: the tap gradients on the matrix cores for 16-bit tensors (see there).
// convc1_bwd_taps_kernel (f32, or planes that are not whole 16-byte segments): one pixel per lane, its 64 masked
// gradients in registers; the weights, transposed to [k][o] in the workspace, arrive through scalar loads
// (the index is wave-uniform), so the 64 K multiply-adds per pixel are plain v_fmac with a scalar operand.
// convc1_bwd_weight_mfma_kernel: the weight and bias sums on the matrix cores (see there);
// convc1_bwd_weight_kernel is the plain-VALU route for images whose planes are not whole 16-byte segments
// (blocks stride over chunks of 64 pixels staged in LDS; thread (o, k mod 4) keeps its dW entries in
// registers).  Either way float block partials are added in double in a fixed order by
// convc1_bwd_reduce_kernel (no atomics, reproducible run to run).
#include "rc_common.h"

#include <mutex>

namespace {

constexpr int CB_OUT = RC_CONVC1_OUT;                              // 64 output channels
constexpr int CB_KMAX = RC_MAX_LEVELS * (2 * RC_MAX_RADIUS + 1);   // 68 input channels at most
constexpr int CB_KJ = (CB_KMAX + 3) / 4;                           // dW entries per thread
constexpr int CB_BLOCKS = 256;                                     // pixel blocks (partials per entry)
constexpr int CB_PIX = 64;                                         // pixels per staged chunk
constexpr int CB_PITCH = CB_PIX + 1;

__global__ void convc1_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over [k][o]
  if (i < K * CB_OUT) wt[i] = w[(i % CB_OUT) * K + i / CB_OUT];
}

template <typename T>
__global__ __launch_bounds__(256) void convc1_bwd_taps_kernel(const T* __restrict__ act, const T* __restrict__ gact,
                                                              const float* __restrict__ wt, float* __restrict__ gcorr,
                                                              int K, int64_t HW, int relu) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int64_t b = blockIdx.y;
  const T* a0 = act + b * CB_OUT * HW + p;
  const T* g0 = gact + b * CB_OUT * HW + p;
  const T* am = relu ? a0 : g0;  // both planes are read unconditionally (see convc1_bwd_weight_kernel)
  float gm[CB_OUT];
#pragma unroll
  for (int o = 0; o < CB_OUT; ++o) {
    const float g = rc_to_float(g0[(int64_t)o * HW]);
    const float a = rc_to_float(am[(int64_t)o * HW]);
    gm[o] = (!relu || a > 0.f) ? g : 0.f;
  }
  float* dst = gcorr + b * K * HW + p;
  for (int k = 0; k < K; ++k) {
    const float* wk = wt + k * CB_OUT;  // wave-uniform: scalar loads
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int o = 0; o < CB_OUT; o += 2) {
      acc0 = fmaf(wk[o], gm[o], acc0);
      acc1 = fmaf(wk[o + 1], gm[o + 1], acc1);
    }
    dst[(int64_t)k * HW] = acc0 + acc1;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void convc1_bwd_weight_kernel(const float* __restrict__ corr,
                                                                const T* __restrict__ act, const T* __restrict__ gact,
                                                                float* __restrict__ part, int K, int64_t HW,
                                                                int64_t chunks_per_image, int64_t n_chunks, int relu) {
  __shared__ float Gs[CB_OUT][CB_PITCH];
  __shared__ float Cs[CB_KMAX][CB_PITCH];
  const int t = threadIdx.x;
  const int o = t >> 2, kq = t & 3;
  const int nj = (K - kq + 3) >> 2;  // entries k = kq + 4 j < K
  float acc[CB_KJ], accb = 0.f;
#pragma unroll
  for (int j = 0; j < CB_KJ; ++j) acc[j] = 0.f;
  const int px = t & 63, r0 = t >> 6;  // staging: row r0 + 4 i, pixel px
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int64_t b = c / chunks_per_image, p = (c - b * chunks_per_image) * CB_PIX + px;
    const bool in = p < HW;
    __syncthreads();  // the previous chunk has been consumed
    // unconditional loads from a clamped pixel, selected after (a load under `if (in)` / `if (relu)` is a branch and
    // the compiler waits for each one where it joins: 32 memory latencies in a row per chunk)
    const int64_t pc = in ? p : 0;
    const T* amask = relu ? act : gact;
    T graw[CB_OUT / 4], araw[CB_OUT / 4];
#pragma unroll
    for (int i = 0; i < CB_OUT / 4; ++i) {
      const int64_t e = (b * CB_OUT + r0 + 4 * i) * HW + pc;
      graw[i] = gact[e];
      araw[i] = amask[e];
    }
#pragma unroll
    for (int i = 0; i < CB_OUT / 4; ++i) {
      float g = rc_to_float(graw[i]);
      if (relu && !(rc_to_float(araw[i]) > 0.f)) g = 0.f;
      Gs[r0 + 4 * i][px] = in ? g : 0.f;
    }
    for (int kk = r0; kk < K; kk += 4) {
      const float cv = corr[(b * K + kk) * HW + pc];
      Cs[kk][px] = in ? cv : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < CB_PIX; ++q) {
      const float g = Gs[o][q];
      accb += g;
#pragma unroll
      for (int j = 0; j < CB_KJ; ++j)
        if (j < nj) acc[j] = fmaf(g, Cs[kq + 4 * j][q], acc[j]);
    }
  }
  // partial [block][o*K + k], then [block][64*K + o]
  float* mine = part + (int64_t)blockIdx.x * (CB_OUT * K + CB_OUT);
#pragma unroll
  for (int j = 0; j < CB_KJ; ++j)
    if (j < nj) mine[o * K + kq + 4 * j] = acc[j];
  if (kq == 0) mine[CB_OUT * K + o] = accb;
}

// ---- the same sums on the matrix cores (whole 16-byte pixel segments: HW % 8 == 0 for 16-bit, % 4 for f32).
// dW[o][k] = sum_px gm[o][px] corr[k][px] is a [64 x P] x [P x K] product with the pixels as the contraction
// index: an MFMA lane wants 8 (16-bit) or 4 (f32) consecutive pixels of ONE channel plane, which is one
// 16-byte global load -- no LDS staging.  A wave owns groups of 32 (16) pixels and keeps the 4 x NT output
// tiles in registers; column K of the last tile is fed ones, so it collects db.  The 4 waves of a block add
// their tiles through LDS one after the other and the block writes one partial.
typedef short cb_s8v __attribute__((ext_vector_type(8)));
typedef _Float16 cb_h8v __attribute__((ext_vector_type(8)));
typedef __bf16 cb_b8v __attribute__((ext_vector_type(8)));
typedef float cb_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t cb_u4v __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ cb_f4v cb_mfma16(cb_s8v a, cb_s8v b, cb_f4v c);
template <> __device__ __forceinline__ cb_f4v cb_mfma16<rc_half>(cb_s8v a, cb_s8v b, cb_f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cb_h8v, a), __builtin_bit_cast(cb_h8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ cb_f4v cb_mfma16<rc_bf16>(cb_s8v a, cb_s8v b, cb_f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cb_b8v, a), __builtin_bit_cast(cb_b8v, b), c, 0, 0, 0);
}

// keep the 16-bit halves of g whose counterpart in a is a positive number (sign-magnitude: 0x0001..0x7fff)
__device__ __forceinline__ uint32_t cb_mask_pos16(uint32_t g, uint32_t a) {
  const uint32_t lo = a & 0xffffu, hi = a >> 16;
  const uint32_t mlo = (lo - 1u) < 0x7fffu ? 0xffffu : 0u;
  const uint32_t mhi = (hi - 1u) < 0x7fffu ? 0xffff0000u : 0u;
  return g & (mlo | mhi);
}

template <typename T> __device__ __forceinline__ uint32_t cb_pack2(float lo, float hi) {
  typedef T v2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, v2));
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void convc1_bwd_weight_mfma_kernel(const float* __restrict__ corr,
                                                                     const T* __restrict__ act,
                                                                     const T* __restrict__ gact,
                                                                     float* __restrict__ part, int K, int64_t HW,
                                                                     int64_t groups_per_image, int64_t n_groups,
                                                                     int relu) {
  constexpr bool IS16 = sizeof(T) == 2;
  constexpr int SEG = IS16 ? 8 : 4;    // pixels per lane segment (16 bytes of T)
  constexpr int GP = 4 * SEG;          // pixels per group
  constexpr int LD = NT * 16;          // padded row of the block partial
  __shared__ float red[CB_OUT * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  cb_f4v acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (cb_f4v)(0.f);

  const T* amask = relu ? act : gact;  // block-uniform
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t b = grp / groups_per_image;
    const int64_t p = (grp - b * groups_per_image) * GP + g * SEG;  // this lane's pixel segment
    const bool in = p < HW;
    const int64_t pc = in ? p : 0;
    // A fragments: masked gradients of channel 16 mt + i
    // (every load of a group is unconditional and issued before the first use: a load under `if (relu)` is a
    // branch, and the compiler then waits for each one where the branches join -- 5 memory latencies per group
    // instead of 1.  Without the ReLU the mask operand is the gradient itself, read twice, and ignored.)
    cb_u4v ga[4], am[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int64_t e = (b * CB_OUT + 16 * mt + i) * HW + pc;
      ga[mt] = *reinterpret_cast<const cb_u4v*>(gact + e);
      am[mt] = *reinterpret_cast<const cb_u4v*>(amask + e);
    }
    cb_f4v craw[NT][IS16 ? 2 : 1];  // all of the group's corr segments in flight together, like the gradients
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int k = 16 * nt + i;
      const float* src = corr + (b * K + (k < K ? k : 0)) * HW + pc;
      craw[nt][0] = *reinterpret_cast<const cb_f4v*>(src);
      if constexpr (IS16) craw[nt][1] = *reinterpret_cast<const cb_f4v*>(src + 4);
    }
    __builtin_amdgcn_sched_barrier(0);  // or the scheduler sinks the later loads behind the first MFMAs
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      cb_u4v gv = ga[mt];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint32_t m;
        if constexpr (IS16) m = cb_mask_pos16(gv[d], am[mt][d]);
        else m = __uint_as_float(am[mt][d]) > 0.f ? gv[d] : 0u;
        gv[d] = relu ? m : gv[d];
      }
      ga[mt] = in ? gv : (cb_u4v)(0u);
    }
    // B fragments: corr of channel 16 nt + i (ones in column K: the bias gradient), zero beyond
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int k = 16 * nt + i;
      const bool real = in && k < K;
      const float fill = (in && k == K) ? 1.f : 0.f;
      if constexpr (IS16) {
        const cb_f4v c0 = craw[nt][0], c1 = craw[nt][1];
        cb_u4v bv;
        bv[0] = cb_pack2<T>(real ? c0[0] : fill, real ? c0[1] : fill);
        bv[1] = cb_pack2<T>(real ? c0[2] : fill, real ? c0[3] : fill);
        bv[2] = cb_pack2<T>(real ? c1[0] : fill, real ? c1[1] : fill);
        bv[3] = cb_pack2<T>(real ? c1[2] : fill, real ? c1[3] : fill);
        const cb_s8v bf = __builtin_bit_cast(cb_s8v, bv);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = cb_mfma16<T>(__builtin_bit_cast(cb_s8v, ga[mt]), bf, acc[mt][nt]);
      } else {
        const cb_f4v c0 = craw[nt][0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // contraction slot g of step q = pixel 4 g + q, for A and B alike
          const float bq = real ? c0[q] : fill;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ga[mt][q]), bq, acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
  // lane (g, i) holds rows o = 16 mt + 4 g + r of column k = 16 nt + i; the waves add up in turn
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* cell = red + (16 * mt + 4 * g + r) * LD + 16 * nt + i;
            *cell = (w == 0 ? 0.f : *cell) + acc[mt][nt][r];
          }
    }
    __syncthreads();
  }
  float* mine = part + (int64_t)blockIdx.x * (CB_OUT * K + CB_OUT);
  for (int e = threadIdx.x; e < CB_OUT * (K + 1); e += 256) {
    const int o = e / (K + 1), k = e - o * (K + 1);
    const float v = red[o * LD + k];
    if (k < K) mine[o * K + k] = v;
    else mine[CB_OUT * K + o] = v;
  }
}

// ---- tap gradients on the matrix cores (16-bit types, whole 16-byte segments):
// grad_corr[k][px] = sum_o W[o][k] gm[o][px] is [K x 64] x [64 x P].  A block stages the masked gradients of 64
// pixels in LDS as [o][px] (16-byte loads along the planes), each wave takes 16 of the pixels and reads its B
// fragments (8 consecutive o of one pixel) back with 16-bit LDS reads; the A fragments are the weights, packed
// once per call by convc1_pack_bwd_kernel.  Results leave as 64-byte runs of each tap plane.
constexpr int CB_GP = 66;  // LDS row pitch in 16-bit elements: 33 dwords, so the 4 lane rows of a read hit 4 bank groups

template <typename T>
__global__ void convc1_pack_bwd_kernel(const float* __restrict__ w, T* __restrict__ wp, int K) {
  // wp[((nt*2 + s)*64 + lane)*8 + j] = W[o = 32 s + 8 g + j][k = 16 nt + i],  lane = 16 g + i
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int nt_total = K / 16 + 1;
  if (e >= nt_total * 2 * 64 * 8) return;
  const int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 1, nt = e >> 10;
  const int k = 16 * nt + (lane & 15), o = 32 * s + 8 * (lane >> 4) + j;
  wp[e] = rc_from_float<T>(k < K ? w[o * K + k] : 0.f);
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void convc1_bwd_taps_mfma_kernel(const T* __restrict__ act, const T* __restrict__ gact,
                                                                   const T* __restrict__ wp, float* __restrict__ gcorr,
                                                                   int K, int64_t HW, int relu) {
  __shared__ uint16_t Gs[CB_OUT * CB_GP];
  const int64_t b = blockIdx.y, p0 = (int64_t)blockIdx.x * 64;
  {
    const int seg = threadIdx.x & 7;
    const int64_t p = p0 + 8 * seg;
    const bool in = p < HW;
    cb_u4v gvs[2], avs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int o = (threadIdx.x >> 3) + 32 * h;
      // unconditional loads from a clamped address (see convc1_bwd_weight_mfma_kernel)
      const int64_t e = (b * CB_OUT + o) * HW + (in ? p : 0);
      gvs[h] = *reinterpret_cast<const cb_u4v*>(gact + e);
      avs[h] = *reinterpret_cast<const cb_u4v*>((relu ? act : gact) + e);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int o = (threadIdx.x >> 3) + 32 * h;
      cb_u4v gv = gvs[h];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t m = cb_mask_pos16(gv[d], avs[h][d]);
        gv[d] = in ? (relu ? m : gv[d]) : 0u;
      }
      uint32_t* dst = reinterpret_cast<uint32_t*>(Gs + o * CB_GP + 8 * seg);
#pragma unroll
      for (int d = 0; d < 4; ++d) dst[d] = gv[d];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  cb_f4v acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (cb_f4v)(0.f);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    cb_u4v bv;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t lo = Gs[(32 * s + 8 * g + 2 * d) * CB_GP + 16 * wave + i];
      const uint32_t hi = Gs[(32 * s + 8 * g + 2 * d + 1) * CB_GP + 16 * wave + i];
      bv[d] = lo | (hi << 16);
    }
    const cb_s8v bf = __builtin_bit_cast(cb_s8v, bv);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const cb_u4v av = *reinterpret_cast<const cb_u4v*>(wp + ((nt * 2 + s) * 64 + lane) * 8);
      acc[nt] = cb_mfma16<T>(__builtin_bit_cast(cb_s8v, av), bf, acc[nt]);
    }
  }
  const int64_t p = p0 + 16 * wave + i;
  if (p < HW) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * nt + 4 * g + r;
        if (k < K) gcorr[(b * K + k) * HW + p] = acc[nt][r];
      }
  }
}

// dW, db = sum of the block partials, in double, in a fixed order: 64 entries x 16 slices per block
__global__ __launch_bounds__(1024) void convc1_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int K,
                                                                 float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ double sl[16][64];
  const int n = CB_OUT * K + CB_OUT;
  const int e = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  double sum = 0.0;
  if (i < n) {
#pragma unroll 4
    for (int b = s; b < nblk; b += 16) sum += (double)part[(int64_t)b * n + i];
  }
  sl[s][e] = sum;
  __syncthreads();
  if (s == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sl[q][e];
    if (i < CB_OUT * K) gw[i] = (float)t;
    else gb[i - CB_OUT * K] = (float)t;
  }
}

template <typename T>
int launch_convc1_bwd(const float* corr, const void* act, const void* gact, const float* weight, float* gcorr,
                      float* gw, float* gb, float* work, int64_t B, int K, int64_t HW, int relu, hipStream_t st) {
  float* wt = work;                                  // [K][64] floats, or the MFMA-packed 16-bit weights
  float* part = work + CB_KMAX * CB_OUT;
  constexpr int SEG = 16 / (int)sizeof(T), GP = 4 * SEG;
  const bool whole = HW % SEG == 0 && (uintptr_t)corr % 16 == 0 && (uintptr_t)act % 16 == 0 && (uintptr_t)gact % 16 == 0;
  int rc;
  if constexpr (sizeof(T) == 2) {
    if (whole) {
      const int NTt = K / 16 + 1;
      hipLaunchKernelGGL((convc1_pack_bwd_kernel<T>), dim3((NTt * 1024 + 255) / 256), dim3(256), 0, st, weight, (T*)wt, K);
      rc = rc_check_launch("rc_convc1_bwd(pack)");
      if (rc) return rc;
      const dim3 grid((unsigned)((HW + 63) / 64), (unsigned)B);
#define RC_CB_TAPS(N)                                                                                              \
  case N:                                                                                                          \
    hipLaunchKernelGGL((convc1_bwd_taps_mfma_kernel<T, N>), grid, dim3(256), 0, st, (const T*)act, (const T*)gact, \
                       (const T*)wt, gcorr, K, HW, relu);                                                          \
    break;
      switch (NTt) {
        RC_CB_TAPS(1)
        RC_CB_TAPS(2)
        RC_CB_TAPS(3)
        RC_CB_TAPS(4)
        RC_CB_TAPS(5)
      }
#undef RC_CB_TAPS
    }
  }
  if (sizeof(T) != 2 || !whole) {
    hipLaunchKernelGGL(convc1_transpose_kernel, dim3((K * CB_OUT + 255) / 256), dim3(256), 0, st, weight, wt, K);
    rc = rc_check_launch("rc_convc1_bwd(transpose)");
    if (rc) return rc;
    hipLaunchKernelGGL((convc1_bwd_taps_kernel<T>), dim3((unsigned)((HW + 255) / 256), (unsigned)B), dim3(256), 0, st,
                       (const T*)act, (const T*)gact, wt, gcorr, K, HW, relu);
  }
  rc = rc_check_launch("rc_convc1_bwd(taps)");
  if (rc) return rc;
  int nblk;
  if (whole) {
    const int64_t gpi = (HW + GP - 1) / GP, n_groups = B * gpi;
    nblk = (int)((n_groups + 7) / 8 < CB_BLOCKS ? (n_groups + 7) / 8 : CB_BLOCKS);
    const int NT = K / 16 + 1;
#define RC_CB_LAUNCH(N)                                                                                          \
  case N:                                                                                                        \
    hipLaunchKernelGGL((convc1_bwd_weight_mfma_kernel<T, N>), dim3(nblk), dim3(256), 0, st, corr, (const T*)act, \
                       (const T*)gact, part, K, HW, gpi, n_groups, relu);                                        \
    break;
    switch (NT) {
      RC_CB_LAUNCH(1)
      RC_CB_LAUNCH(2)
      RC_CB_LAUNCH(3)
      RC_CB_LAUNCH(4)
      RC_CB_LAUNCH(5)
    }
#undef RC_CB_LAUNCH
  } else {
    const int64_t cpi = (HW + CB_PIX - 1) / CB_PIX, n_chunks = B * cpi;
    nblk = (int)(n_chunks < CB_BLOCKS ? n_chunks : CB_BLOCKS);
    hipLaunchKernelGGL((convc1_bwd_weight_kernel<T>), dim3(nblk), dim3(256), 0, st, corr, (const T*)act,
                       (const T*)gact, part, K, HW, cpi, n_chunks, relu);
  }
  rc = rc_check_launch("rc_convc1_bwd(weight)");
  if (rc) return rc;
  const int n = CB_OUT * K + CB_OUT;
  hipLaunchKernelGGL(convc1_bwd_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, part, nblk, K, gw, gb);
  return rc_check_launch("rc_convc1_bwd(reduce)");
}

}  // namespace

extern "C" int64_t rc_convc1_bwd_work_bytes(void) {
  return ((int64_t)CB_KMAX * CB_OUT + (int64_t)CB_BLOCKS * (CB_OUT * CB_KMAX + CB_OUT)) * (int64_t)sizeof(float);
}

extern "C" int rc_convc1_bwd(const float* corr, const void* act, const void* grad_act, const float* weight,
                             float* grad_corr, float* grad_weight, float* grad_bias, void* work, int64_t B,
                             int in_channels, int64_t HW, int dtype, int relu, void* stream) {
  RC_REQUIRE(corr && act && grad_act && weight && grad_corr && grad_weight && grad_bias && work, RC_EINVAL,
             "null pointer argument");
  RC_REQUIRE(B > 0 && B < 65536 && HW > 0 && HW < (1ll << 31), RC_EINVAL, "bad sizes B=%lld HW=%lld", (long long)B,
             (long long)HW);
  RC_REQUIRE(in_channels >= 1 && in_channels <= CB_KMAX, RC_EUNSUPPORTED, "in_channels %d outside 1..%d",
             in_channels, CB_KMAX);
  RC_REQUIRE((uintptr_t)work % 16 == 0, RC_EINVAL, "work must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  float* wk = (float*)work;
  switch (dtype) {
    case RC_F32:
      return launch_convc1_bwd<float>(corr, act, grad_act, weight, grad_corr, grad_weight, grad_bias, wk, B,
                                      in_channels, HW, relu, st);
    case RC_F16:
      return launch_convc1_bwd<rc_half>(corr, act, grad_act, weight, grad_corr, grad_weight, grad_bias, wk, B,
                                        in_channels, HW, relu, st);
    case RC_BF16:
      return launch_convc1_bwd<rc_bf16>(corr, act, grad_act, weight, grad_corr, grad_weight, grad_bias, wk, B,
                                        in_channels, HW, relu, st);
  }
  rc_set_error("unknown dtype %d", dtype);
  return RC_EUNSUPPORTED;
}

// ---- the same adjoint for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses
// that forbid hipMalloc): the workspace is a file-scope device array.  Every GPU the library runs on holds
// its own copy of the array at its own address until the process ends; the address is looked up once per
// device, with that device current, and from then on only that pointer is used on the host side (the
// array's name is a host-side stand-in, not a device address).
namespace {
constexpr size_t CB_SCRATCH_BYTES =
    ((size_t)CB_KMAX * CB_OUT + (size_t)CB_BLOCKS * (CB_OUT * CB_KMAX + CB_OUT)) * sizeof(float);  // 4 539 392 B
constexpr int CB_MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[CB_SCRATCH_BYTES];
static void* g_scratch_ptr[CB_MAX_DEVICES];     // per device, set once (under g_scratch_lock)
static std::mutex g_scratch_lock;

extern "C" int rc_convc1_bwd_static(const float* corr, const void* act, const void* grad_act, const float* weight,
                                    float* grad_corr, float* grad_weight, float* grad_bias, int64_t B,
                                    int in_channels, int64_t HW, int dtype, int relu, void* stream) {
  // the scratch array is sized for the widest convolution input the library supports; anything wider would
  // overrun it, so it is refused here, before any device work
  RC_REQUIRE(in_channels >= 1 && in_channels <= CB_KMAX, RC_EUNSUPPORTED,
             "rc_convc1_bwd_static: in_channels %d exceeds the static scratch buffer's maximum of %d "
             "(%zu bytes per device)", in_channels, CB_KMAX, CB_SCRATCH_BYTES);
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= CB_MAX_DEVICES) {
    rc_set_error("rc_convc1_bwd_static: no current HIP device (or its index is >= %d)", CB_MAX_DEVICES);
    return RC_ENODEVICE;
  }
  void* ptr;
  {
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
      void* p = nullptr;
      e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // this device's copy (the device is current)
      if (e != hipSuccess || !p) {
        rc_set_error("rc_convc1_bwd_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
        return e != hipSuccess ? (int)e : RC_ENODEVICE;
      }
      g_scratch_ptr[dev] = p;
    }
    ptr = g_scratch_ptr[dev];
  }
  // nothing in the buffer has to start at zero: the pack / transpose kernels write the weights' part and
  // every partial-sum slot the reduction reads is written by this call's own blocks first
  return rc_convc1_bwd(corr, act, grad_act, weight, grad_corr, grad_weight, grad_bias, ptr, B, in_channels, HW,
                       dtype, relu, stream);
}

// Pointwise work of one GRU iteration that surrounds the correlation lookup (SURVEY.md section 8 f,
// ranks 3 and 4): the ConvGRU gate arithmetic and the convex upsampling of the disparity field.
//
// Replaces, in the reference:
//   rc_gru_gate_r / rc_gru_update     core/update.py:27-31   (the sigmoid / tanh / blend around the
//                                     three convolutions of ConvGRU.forward; the convolutions stay
//                                     with the caller)
//   rc_upsample_flow_fwd              core/raft_stereo.py:55-67  (softmax over the 9 neighbours,
//                                     unfold of factor*flow, weighted sum, permute)
// All three are streaming kernels (HBM bound): every operand is read once with the widest load the
// alignment allows and every result is written once.
#include "rc_common.h"

#include <math.h>

namespace {

constexpr int PW_THREADS = 256;

// ------------------------------------------------------------------------------------------ GRU
// What a chain of torch ops on 16-bit tensors computes: every op evaluates in fp32 and rounds its
// result to the tensor type.  rnd<T>() marks each of those roundings (identity for float).
template <typename T> __device__ __forceinline__ float rnd(float v) { return rc_round_as<T>(v); }

// f32 tensors take the accurate library functions; for 16-bit tensors, whose results are rounded to 11 or 8
// bits anyway, v_exp_f32 / v_rcp_f32 (about 1 ulp of fp32 each) keep the kernels off the VALU roof
template <typename T> __device__ __forceinline__ float sigmoid_t(float x) {
  if constexpr (sizeof(T) == 4) return 1.0f / (1.0f + expf(-x));
  else return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
template <typename T> __device__ __forceinline__ float tanh_t(float x) {
  if constexpr (sizeof(T) == 4) return tanhf(x);
  else return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

template <typename T, int VEC> struct pw_vec {
  typedef T type __attribute__((ext_vector_type(VEC)));
};
template <typename T> struct pw_vec<T, 1> {
  struct type {
    T v;
    __device__ __forceinline__ T& operator[](int) { return v; }
    __device__ __forceinline__ const T& operator[](int) const { return v; }
  };
};

template <typename T, int VEC>
__device__ __forceinline__ typename pw_vec<T, VEC>::type pw_load(const T* p) {
  return *reinterpret_cast<const typename pw_vec<T, VEC>::type*>(p);
}

// r*h of core/update.py:28-29:  r = sigmoid(convr(hx) + cr);  out = r * h
template <typename T, int VEC>
__global__ __launch_bounds__(PW_THREADS) void gru_gate_r_kernel(const T* __restrict__ convr, int64_t convr_bs,
                                                                const T* __restrict__ cr, int64_t cr_bs,
                                                                const T* __restrict__ h,
                                                                T* __restrict__ rh, int64_t rh_bs, int64_t per_batch,
                                                                int64_t total) {
  const int64_t i = ((int64_t)blockIdx.x * PW_THREADS + threadIdx.x) * VEC;  // inside image b = blockIdx.y
  if (i >= per_batch) return;
  const int64_t b = blockIdx.y, e = b * per_batch + i;
  const auto a = pw_load<T, VEC>(convr + b * convr_bs + i);
  const auto c = pw_load<T, VEC>(cr + b * cr_bs + i);
  const auto hv = pw_load<T, VEC>(h + e);
  typename pw_vec<T, VEC>::type o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float s = rnd<T>(rc_to_float(a[j]) + rc_to_float(c[j]));
    const float r = rnd<T>(sigmoid_t<T>(s));
    o[j] = rc_from_float<T>(r * rc_to_float(hv[j]));
  }
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(rh + b * rh_bs + i) = o;
}

// core/update.py:27,29,31:  z = sigmoid(convz(hx) + cz);  q = tanh(convq([r*h, x]) + cq);  h = (1-z)*h + z*q
template <typename T, int VEC>
__global__ __launch_bounds__(PW_THREADS) void gru_update_kernel(const T* __restrict__ convz, int64_t convz_bs,
                                                                const T* __restrict__ cz, int64_t cz_bs,
                                                                const T* __restrict__ convq, int64_t convq_bs,
                                                                const T* __restrict__ cq, int64_t cq_bs,
                                                                const T* h, T* h_out, int64_t per_batch,
                                                                int64_t total) {
  const int64_t i = ((int64_t)blockIdx.x * PW_THREADS + threadIdx.x) * VEC;  // inside image b = blockIdx.y
  if (i >= per_batch) return;
  const int64_t b = blockIdx.y, e = b * per_batch + i;
  const auto az = pw_load<T, VEC>(convz + b * convz_bs + i);
  const auto bz = pw_load<T, VEC>(cz + b * cz_bs + i);
  const auto aq = pw_load<T, VEC>(convq + b * convq_bs + i);
  const auto bq = pw_load<T, VEC>(cq + b * cq_bs + i);
  const auto hv = pw_load<T, VEC>(h + e);
  typename pw_vec<T, VEC>::type o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float z = rnd<T>(sigmoid_t<T>(rnd<T>(rc_to_float(az[j]) + rc_to_float(bz[j]))));
    const float q = rnd<T>(tanh_t<T>(rnd<T>(rc_to_float(aq[j]) + rc_to_float(bq[j]))));
    const float keep = rnd<T>(rnd<T>(1.0f - z) * rc_to_float(hv[j]));
    const float take = rnd<T>(z * q);
    o[j] = rc_from_float<T>(keep + take);
  }
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(h_out + e) = o;
}

// Adjoint of gru_update_kernel (autograd of core/update.py:27,29,31), gates recomputed from the saved
// pre-activations:  dz_pre = g (q - h) z (1-z),  dq_pre = g z (1 - q^2),  dh = g (1-z)
template <typename T, int VEC>
__global__ __launch_bounds__(PW_THREADS) void gru_update_bwd_kernel(
    const T* __restrict__ convz, const T* __restrict__ cz, int64_t cz_bs, const T* __restrict__ convq,
    const T* __restrict__ cq, int64_t cq_bs, const T* __restrict__ h, const T* __restrict__ gout,
    T* __restrict__ dzpre, T* __restrict__ dqpre, T* __restrict__ dh, int64_t per_batch, int64_t total) {
  const int64_t i = ((int64_t)blockIdx.x * PW_THREADS + threadIdx.x) * VEC;  // inside image b = blockIdx.y
  if (i >= per_batch) return;
  const int64_t b = blockIdx.y, e = b * per_batch + i;
  const auto az = pw_load<T, VEC>(convz + e);
  const auto bz = pw_load<T, VEC>(cz + b * cz_bs + i);
  const auto aq = pw_load<T, VEC>(convq + e);
  const auto bq = pw_load<T, VEC>(cq + b * cq_bs + i);
  const auto hv = pw_load<T, VEC>(h + e);
  const auto gv = pw_load<T, VEC>(gout + e);
  typename pw_vec<T, VEC>::type oz, oq, oh;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float z = rnd<T>(sigmoid_t<T>(rnd<T>(rc_to_float(az[j]) + rc_to_float(bz[j]))));
    const float q = rnd<T>(tanh_t<T>(rnd<T>(rc_to_float(aq[j]) + rc_to_float(bq[j]))));
    const float g = rc_to_float(gv[j]), hh = rc_to_float(hv[j]);
    oz[j] = rc_from_float<T>(g * (q - hh) * z * (1.0f - z));
    oq[j] = rc_from_float<T>(g * z * (1.0f - q * q));
    oh[j] = rc_from_float<T>(g * (1.0f - z));
  }
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(dzpre + e) = oz;
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(dqpre + e) = oq;
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(dh + e) = oh;
}

// Adjoint of gru_gate_r_kernel:  dr_pre = g h r (1-r);  dh (+)= g r   (dh_accum != 0: added to what dh holds,
// which is how the two uses of h in core/update.py:29,31 meet)
template <typename T, int VEC>
__global__ __launch_bounds__(PW_THREADS) void gru_gate_r_bwd_kernel(const T* __restrict__ convr,
                                                                    const T* __restrict__ cr, int64_t cr_bs,
                                                                    const T* __restrict__ h, const T* __restrict__ grh,
                                                                    int64_t grh_bs, T* __restrict__ drpre, T* dh,
                                                                    int dh_accum, int64_t per_batch, int64_t total) {
  const int64_t i = ((int64_t)blockIdx.x * PW_THREADS + threadIdx.x) * VEC;  // inside image b = blockIdx.y
  if (i >= per_batch) return;
  const int64_t b = blockIdx.y, e = b * per_batch + i;
  const auto a = pw_load<T, VEC>(convr + e);
  const auto c = pw_load<T, VEC>(cr + b * cr_bs + i);
  const auto hv = pw_load<T, VEC>(h + e);
  const auto gv = pw_load<T, VEC>(grh + b * grh_bs + i);
  typename pw_vec<T, VEC>::type prev;
  if (dh_accum) prev = pw_load<T, VEC>(dh + e);
  typename pw_vec<T, VEC>::type orr, oh;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float r = rnd<T>(sigmoid_t<T>(rnd<T>(rc_to_float(a[j]) + rc_to_float(c[j]))));
    const float g = rc_to_float(gv[j]);
    orr[j] = rc_from_float<T>(g * rc_to_float(hv[j]) * r * (1.0f - r));
    const float d = g * r;
    oh[j] = rc_from_float<T>(dh_accum ? rc_to_float(prev[j]) + d : d);
  }
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(drpre + e) = orr;
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(dh + e) = oh;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

inline int pw_blocks(int64_t N, int64_t per_batch, int vec, dim3* blocks) {
  const int64_t n = (per_batch / vec + PW_THREADS - 1) / PW_THREADS;
  RC_REQUIRE(n < (1ll << 31) && N < 65536, RC_EINVAL, "too many elements (N=%lld, per_batch=%lld)", (long long)N,
             (long long)per_batch);
  *blocks = dim3((unsigned)n, (unsigned)N);
  return RC_OK;
}

inline int pw_check(int64_t N, int64_t per_batch, int dtype) {
  RC_REQUIRE(N > 0 && per_batch > 0 && N < (1ll << 31) && per_batch < (1ll << 40), RC_EINVAL,
             "bad sizes N=%lld per_batch=%lld", (long long)N, (long long)per_batch);
  RC_REQUIRE(dtype == RC_F32 || dtype == RC_F16 || dtype == RC_BF16, RC_EUNSUPPORTED, "unknown dtype %d", dtype);
  return RC_OK;
}

template <typename T>
int launch_gate_r(const void* convr, int64_t convr_bs, const void* cr, int64_t cr_bs, const void* h, void* rh,
                  int64_t rh_bs,
                  int64_t N, int64_t per_batch, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t total = N * per_batch;
  const bool vec = per_batch % V == 0 && convr_bs % V == 0 && cr_bs % V == 0 && rh_bs % V == 0 && aligned16(convr) && aligned16(cr) &&
                   aligned16(h) && aligned16(rh);
  dim3 blocks;
  int rc = pw_blocks(N, per_batch, vec ? V : 1, &blocks);
  if (rc) return rc;
  if (vec)
    hipLaunchKernelGGL((gru_gate_r_kernel<T, V>), blocks, dim3(PW_THREADS), 0, st, (const T*)convr, convr_bs,
                       (const T*)cr, cr_bs, (const T*)h, (T*)rh, rh_bs, per_batch, total);
  else
    hipLaunchKernelGGL((gru_gate_r_kernel<T, 1>), blocks, dim3(PW_THREADS), 0, st, (const T*)convr, convr_bs,
                       (const T*)cr, cr_bs, (const T*)h, (T*)rh, rh_bs, per_batch, total);
  return rc_check_launch("rc_gru_gate_r");
}

template <typename T>
int launch_update(const void* convz, int64_t convz_bs, const void* cz, int64_t cz_bs, const void* convq,
                  int64_t convq_bs, const void* cq, int64_t cq_bs, const void* h, void* h_out, int64_t N, int64_t per_batch, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t total = N * per_batch;
  const bool vec = per_batch % V == 0 && convz_bs % V == 0 && convq_bs % V == 0 && cz_bs % V == 0 && cq_bs % V == 0 && aligned16(convz) && aligned16(cz) &&
                   aligned16(convq) && aligned16(cq) && aligned16(h) && aligned16(h_out);
  dim3 blocks;
  int rc = pw_blocks(N, per_batch, vec ? V : 1, &blocks);
  if (rc) return rc;
  if (vec)
    hipLaunchKernelGGL((gru_update_kernel<T, V>), blocks, dim3(PW_THREADS), 0, st, (const T*)convz, convz_bs,
                       (const T*)cz, cz_bs, (const T*)convq, convq_bs, (const T*)cq, cq_bs, (const T*)h, (T*)h_out, per_batch,
                       total);
  else
    hipLaunchKernelGGL((gru_update_kernel<T, 1>), blocks, dim3(PW_THREADS), 0, st, (const T*)convz, convz_bs,
                       (const T*)cz, cz_bs, (const T*)convq, convq_bs, (const T*)cq, cq_bs, (const T*)h, (T*)h_out, per_batch,
                       total);
  return rc_check_launch("rc_gru_update");
}

template <typename T>
int launch_update_bwd(const void* convz, const void* cz, int64_t cz_bs, const void* convq, const void* cq,
                      int64_t cq_bs, const void* h, const void* gout, void* dz, void* dq, void* dh, int64_t N,
                      int64_t per_batch, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t total = N * per_batch;
  const bool vec = per_batch % V == 0 && cz_bs % V == 0 && cq_bs % V == 0 && aligned16(convz) && aligned16(cz) &&
                   aligned16(convq) && aligned16(cq) && aligned16(h) && aligned16(gout) && aligned16(dz) &&
                   aligned16(dq) && aligned16(dh);
  dim3 blocks;
  int rc = pw_blocks(N, per_batch, vec ? V : 1, &blocks);
  if (rc) return rc;
  if (vec)
    hipLaunchKernelGGL((gru_update_bwd_kernel<T, V>), blocks, dim3(PW_THREADS), 0, st, (const T*)convz,
                       (const T*)cz, cz_bs, (const T*)convq, (const T*)cq, cq_bs, (const T*)h, (const T*)gout,
                       (T*)dz, (T*)dq, (T*)dh, per_batch, total);
  else
    hipLaunchKernelGGL((gru_update_bwd_kernel<T, 1>), blocks, dim3(PW_THREADS), 0, st, (const T*)convz,
                       (const T*)cz, cz_bs, (const T*)convq, (const T*)cq, cq_bs, (const T*)h, (const T*)gout,
                       (T*)dz, (T*)dq, (T*)dh, per_batch, total);
  return rc_check_launch("rc_gru_update_bwd");
}

template <typename T>
int launch_gate_r_bwd(const void* convr, const void* cr, int64_t cr_bs, const void* h, const void* grh,
                      int64_t grh_bs, void* dr, void* dh, int dh_accum, int64_t N, int64_t per_batch,
                      hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t total = N * per_batch;
  const bool vec = per_batch % V == 0 && cr_bs % V == 0 && grh_bs % V == 0 && aligned16(convr) && aligned16(cr) &&
                   aligned16(h) && aligned16(grh) && aligned16(dr) && aligned16(dh);
  dim3 blocks;
  int rc = pw_blocks(N, per_batch, vec ? V : 1, &blocks);
  if (rc) return rc;
  if (vec)
    hipLaunchKernelGGL((gru_gate_r_bwd_kernel<T, V>), blocks, dim3(PW_THREADS), 0, st, (const T*)convr,
                       (const T*)cr, cr_bs, (const T*)h, (const T*)grh, grh_bs, (T*)dr, (T*)dh, dh_accum, per_batch,
                       total);
  else
    hipLaunchKernelGGL((gru_gate_r_bwd_kernel<T, 1>), blocks, dim3(PW_THREADS), 0, st, (const T*)convr,
                       (const T*)cr, cr_bs, (const T*)h, (const T*)grh, grh_bs, (T*)dr, (T*)dh, dh_accum, per_batch,
                       total);
  return rc_check_launch("rc_gru_gate_r_bwd");
}

// ------------------------------------------------------------------------------------- upsampling
// A wave owns 64 consecutive low-resolution pixels of one row and walks the F x F output pixels each
// expands into, G = min(F,4) output columns (fx) of one output row (fy) per step: the lane loads its
// 9 G mask logits (every load of the wave is one contiguous run of a mask plane; the next step's loads are
// issued before this step's arithmetic), normalises them and writes G consecutive output pixels per flow
// channel as one 8/16-byte store: the wave's store is one contiguous run of an output row.
// MT is the mask type: softmax evaluates in fp32 and rounds to MT, as torch.softmax on an MT tensor
// does (core/raft_stereo.py:60; the mask is fp16 under autocast), the products and the sum are fp32.
template <int G> struct up_vec;
template <> struct up_vec<1> { typedef float type; };
template <> struct up_vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct up_vec<4> { typedef float type __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ int64_t up_uniform(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <typename MT, int F, int C>
__global__ __launch_bounds__(PW_THREADS) void upsample_kernel(const float* __restrict__ flow, int64_t flow_bs,
                                                              const MT* __restrict__ mask, float* __restrict__ out,
                                                              int H, int W, int chunks_per_row, int n_items) {
  constexpr int G = F < 4 ? F : 4;    // output columns per step
  constexpr int S = F * (F / G);      // steps: (fy, column group) pairs
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * (PW_THREADS / 64) + wave;  // (n*H + h)*chunks + chunk, wave-uniform
  if (item >= n_items) return;
  const int nh = item / chunks_per_row;
  const int c0 = (item - nh * chunks_per_row) * 64;
  const int n = nh / H, h = nh - n * H;
  const int w = c0 + lane;
  if (w >= W) return;
  const int64_t HW = (int64_t)H * W;

  // mask[n, k, fy, fx, h, w]: plane (k*F + fy)*F + fx of image n.  Step s covers planes k*F*F + s*G + j.
  const MT* mrow = mask + up_uniform((int64_t)n * 9 * F * F * HW + (int64_t)h * W + c0);
  const uint32_t HWu = (uint32_t)HW;
  MT buf[2][9][G];  // the next step's logits are in flight while this step's are normalised
  auto fetch = [&](int s, MT (&m)[9][G]) {
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int j = 0; j < G; ++j) m[k][j] = mrow[(uint32_t)(k * F * F + s * G + j) * HWu + (uint32_t)lane];
  };
  fetch(0, buf[0]);

  // factor * flow at the 9 neighbours (F.unfold(factor*flow, [3,3], padding=1): zeros outside)
  float v[C][9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int hh = h + ky - 1;
    const bool rok = (unsigned)hh < (unsigned)H;  // wave-uniform
    const float* frow = flow + up_uniform((int64_t)n * flow_bs + (int64_t)(rok ? hh : h) * W);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ww = w + kx - 1;
      const bool ok = rok && (unsigned)ww < (unsigned)W;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float f = (frow + (int64_t)c * HW)[ok ? ww : w];
        v[c][ky * 3 + kx] = ok ? (float)F * f : 0.f;
      }
    }
  }

  float* orow = out + up_uniform(((int64_t)n * C * H + h) * F * (int64_t)F * W + (int64_t)F * c0);
  constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) fetch(s + 1, buf[(s + 1) & 1]);
    const MT(&m)[9][G] = buf[s & 1];
    const int fy = s / (F / G), g0 = (s % (F / G)) * G;
    float res[C][G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      float mx = rc_to_float(m[0][j]);
#pragma unroll
      for (int k = 1; k < 9; ++k) mx = fmaxf(mx, rc_to_float(m[k][j]));
      float e[9], sum = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if constexpr (sizeof(MT) == 4) e[k] = expf(m[k][j] - mx);
        else e[k] = __builtin_amdgcn_exp2f(fmaf(rc_to_float(m[k][j]), LOG2E, -mx * LOG2E));
        sum += e[k];
      }
      const float inv = sizeof(MT) == 4 ? 1.0f / sum : __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int c = 0; c < C; ++c) res[c][j] = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float p = rc_round_as<MT>(e[k] * inv);
#pragma unroll
        for (int c = 0; c < C; ++c) res[c][j] += p * v[c][k];
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      typename up_vec<G>::type o;
      if constexpr (G == 1) {
        o = res[c][0];
      } else {
#pragma unroll
        for (int j = 0; j < G; ++j) o[j] = res[c][j];
      }
      float* dst = orow + up_uniform(((int64_t)c * H * F + fy) * (int64_t)F * W) + (uint32_t)(lane * F + g0);
      __builtin_nontemporal_store(o, reinterpret_cast<typename up_vec<G>::type*>(dst));
    }
  }
}

// ---- adjoint of the upsampling (autograd of core/raft_stereo.py:59-67).  With g = d out, v_k = f * flow at
// neighbour k and p = softmax(mask):
//   a_k = sum_c g[c] v_k[c]  (rounded to the mask type: it is the gradient of the MT tensor `p`),
//   d mask_k = p_k (a_k - sum_j p_j a_j),        t[c][k] = sum over (fy,fx) of p_k g[c]
// Kernel A: a wave owns 64 low-resolution pixels of a row, walks their f x f output pixels, writes d mask
// (same plane-wise runs as the forward's loads) and the 9 C sums t of each pixel into `work` [N,C,9,H,W].
// Kernel B: d flow[c,h,w] = f * sum_k t[c][k] at the pixel whose k-th neighbour is (h,w): a gather, no atomics.
template <typename MT, int F, int C>
__global__ __launch_bounds__(PW_THREADS) void upsample_bwd_kernel(const float* __restrict__ flow, int64_t flow_bs,
                                                                  const MT* __restrict__ mask,
                                                                  const float* __restrict__ gout,
                                                                  MT* __restrict__ dmask, float* __restrict__ work,
                                                                  int H, int W, int chunks_per_row, int n_items) {
  constexpr int G = F < 4 ? F : 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * (PW_THREADS / 64) + wave;  // (n*H + h)*chunks + chunk, wave-uniform
  if (item >= n_items) return;
  const int nh = item / chunks_per_row;
  const int c0 = (item - nh * chunks_per_row) * 64;
  const int n = nh / H, h = nh - n * H;
  const int w = c0 + lane;
  if (w >= W) return;
  const int64_t HW = (int64_t)H * W;

  float v[C][9], t[C][9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int hh = h + ky - 1;
    const bool rok = (unsigned)hh < (unsigned)H;
    const float* frow = flow + up_uniform((int64_t)n * flow_bs + (int64_t)(rok ? hh : h) * W);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ww = w + kx - 1;
      const bool ok = rok && (unsigned)ww < (unsigned)W;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float f = (frow + (int64_t)c * HW)[ok ? ww : w];
        v[c][ky * 3 + kx] = ok ? (float)F * f : 0.f;
        t[c][ky * 3 + kx] = 0.f;
      }
    }
  }

  const int64_t mbase = up_uniform((int64_t)n * 9 * F * F * HW + (int64_t)h * W + c0);
  const MT* mrow = mask + mbase;
  MT* drow = dmask + mbase;
  const float* grow = gout + up_uniform(((int64_t)n * C * H + h) * F * (int64_t)F * W + (int64_t)F * c0);
  const uint32_t HWu = (uint32_t)HW;
#pragma unroll 1
  for (int fy = 0; fy < F; ++fy) {
#pragma unroll 1
    for (int g0 = 0; g0 < F; g0 += G) {
      float m[9][G];
      const uint32_t pl0 = (uint32_t)(fy * F + g0);
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < G; ++j) m[k][j] = rc_to_float(mrow[(uint32_t)(k * F * F + pl0 + j) * HWu + (uint32_t)lane]);
      float gg[C][G];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float* src = grow + up_uniform(((int64_t)c * H * F + fy) * (int64_t)F * W) + (uint32_t)(lane * F + g0);
        const auto q = *reinterpret_cast<const typename up_vec<G>::type*>(src);
        if constexpr (G == 1) {
          gg[c][0] = q;
        } else {
#pragma unroll
          for (int j = 0; j < G; ++j) gg[c][j] = q[j];
        }
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        float mx = m[0][j];
#pragma unroll
        for (int k = 1; k < 9; ++k) mx = fmaxf(mx, m[k][j]);
        float e[9], s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if constexpr (sizeof(MT) == 4) e[k] = expf(m[k][j] - mx);
          else e[k] = __expf(m[k][j] - mx);
          s += e[k];
        }
        const float inv = sizeof(MT) == 4 ? 1.0f / s : __builtin_amdgcn_rcpf(s);
        float a[9], dot = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          e[k] = rc_round_as<MT>(e[k] * inv);  // p_k
          float ak = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            ak += gg[c][j] * v[c][k];
            t[c][k] += e[k] * gg[c][j];
          }
          a[k] = rc_round_as<MT>(ak);
          dot += e[k] * a[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
          drow[(uint32_t)(k * F * F + pl0 + j) * HWu + (uint32_t)lane] = rc_from_float<MT>(e[k] * (a[k] - dot));
      }
    }
  }
  float* wrow = work + up_uniform((int64_t)n * C * 9 * HW + (int64_t)h * W + c0);
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wrow[(uint32_t)(c * 9 + k) * HWu + (uint32_t)lane] = t[c][k];
}

template <int C>
__global__ __launch_bounds__(PW_THREADS) void upsample_bwd_flow_kernel(const float* __restrict__ work,
                                                                       float* __restrict__ dflow, int64_t dflow_bs,
                                                                       int H, int W, float factor, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;  // (n, h, w)
  if (idx >= total) return;
  const int w = (int)(idx % W);
  const int64_t r = idx / W;
  const int h = (int)(r % H);
  const int64_t n = r / H;
  const int64_t HW = (int64_t)H * W;
  // all 9 C loads unconditional, from a clamped neighbour (a load under the bounds test is a branch, and the
  // compiler waits for each one where it joins); what lies outside the image is dropped after
  float v[C][9];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int hs = min(max(h - (k / 3) + 1, 0), H - 1), ws = min(max(w - (k % 3) + 1, 0), W - 1);
      v[c][k] = work[((n * C + c) * 9 + k) * HW + (int64_t)hs * W + ws];
    }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int hs = h - (k / 3) + 1, ws = w - (k % 3) + 1;  // the pixel whose k-th neighbour is (h, w)
      if ((unsigned)hs < (unsigned)H && (unsigned)ws < (unsigned)W) acc += v[c][k];
    }
    dflow[n * dflow_bs + (int64_t)c * HW + (int64_t)h * W + w] = factor * acc;
  }
}

template <typename MT, int F, int C>
int launch_upsample_bwd(const float* flow, int64_t flow_bs, const void* mask, const float* gout, void* dmask,
                        float* dflow, int64_t dflow_bs, float* work, int64_t N, int64_t H, int64_t W,
                        hipStream_t st) {
  const int chunks = (int)((W + 63) / 64);
  const int64_t items = N * H * chunks;
  RC_REQUIRE(items < (1ll << 31), RC_EINVAL, "too many pixels");
  const int blocks = (int)((items + PW_THREADS / 64 - 1) / (PW_THREADS / 64));
  hipLaunchKernelGGL((upsample_bwd_kernel<MT, F, C>), dim3(blocks), dim3(PW_THREADS), 0, st, flow, flow_bs,
                     (const MT*)mask, gout, (MT*)dmask, work, (int)H, (int)W, chunks, (int)items);
  int rc = rc_check_launch("rc_upsample_flow_bwd");
  if (rc) return rc;
  const int64_t total = N * H * W;
  hipLaunchKernelGGL((upsample_bwd_flow_kernel<C>), dim3((unsigned)((total + PW_THREADS - 1) / PW_THREADS)),
                     dim3(PW_THREADS), 0, st, work, dflow, dflow_bs, (int)H, (int)W, (float)F, total);
  return rc_check_launch("rc_upsample_flow_bwd(flow)");
}

template <typename MT, int F>
int launch_upsample_bwd_c(const float* flow, int64_t flow_bs, const void* mask, const float* gout, void* dmask,
                          float* dflow, int64_t dflow_bs, float* work, int64_t N, int C, int64_t H, int64_t W,
                          hipStream_t st) {
  return C == 1 ? launch_upsample_bwd<MT, F, 1>(flow, flow_bs, mask, gout, dmask, dflow, dflow_bs, work, N, H, W, st)
                : launch_upsample_bwd<MT, F, 2>(flow, flow_bs, mask, gout, dmask, dflow, dflow_bs, work, N, H, W, st);
}

template <typename MT>
int launch_upsample_bwd_f(const float* flow, int64_t flow_bs, const void* mask, const float* gout, void* dmask,
                          float* dflow, int64_t dflow_bs, float* work, int64_t N, int C, int64_t H, int64_t W,
                          int factor, hipStream_t st) {
  switch (factor) {
    case 1: return launch_upsample_bwd_c<MT, 1>(flow, flow_bs, mask, gout, dmask, dflow, dflow_bs, work, N, C, H, W, st);
    case 2: return launch_upsample_bwd_c<MT, 2>(flow, flow_bs, mask, gout, dmask, dflow, dflow_bs, work, N, C, H, W, st);
    case 4: return launch_upsample_bwd_c<MT, 4>(flow, flow_bs, mask, gout, dmask, dflow, dflow_bs, work, N, C, H, W, st);
    case 8: return launch_upsample_bwd_c<MT, 8>(flow, flow_bs, mask, gout, dmask, dflow, dflow_bs, work, N, C, H, W, st);
  }
  rc_set_error("upsampling factor %d is not one of 1, 2, 4, 8", factor);
  return RC_EUNSUPPORTED;
}

template <typename MT, int F, int C>
int launch_upsample(const float* flow, int64_t flow_bs, const void* mask, float* out, int64_t N, int64_t H,
                    int64_t W, hipStream_t st) {
  const int chunks = (int)((W + 63) / 64);
  const int64_t items = N * H * chunks;
  RC_REQUIRE(items < (1ll << 31), RC_EINVAL, "too many pixels");
  const int blocks = (int)((items + PW_THREADS / 64 - 1) / (PW_THREADS / 64));
  hipLaunchKernelGGL((upsample_kernel<MT, F, C>), blocks, dim3(PW_THREADS), 0, st, flow, flow_bs,
                     (const MT*)mask, out, (int)H, (int)W, chunks, (int)items);
  return rc_check_launch("rc_upsample_flow_fwd");
}

template <typename MT, int F>
int launch_upsample_c(const float* flow, int64_t flow_bs, const void* mask, float* out, int64_t N, int C,
                      int64_t H, int64_t W, hipStream_t st) {
  return C == 1 ? launch_upsample<MT, F, 1>(flow, flow_bs, mask, out, N, H, W, st)
                : launch_upsample<MT, F, 2>(flow, flow_bs, mask, out, N, H, W, st);
}

template <typename MT>
int launch_upsample_f(const float* flow, int64_t flow_bs, const void* mask, float* out, int64_t N, int C,
                      int64_t H, int64_t W, int factor, hipStream_t st) {
  switch (factor) {
    case 1: return launch_upsample_c<MT, 1>(flow, flow_bs, mask, out, N, C, H, W, st);
    case 2: return launch_upsample_c<MT, 2>(flow, flow_bs, mask, out, N, C, H, W, st);
    case 4: return launch_upsample_c<MT, 4>(flow, flow_bs, mask, out, N, C, H, W, st);
    case 8: return launch_upsample_c<MT, 8>(flow, flow_bs, mask, out, N, C, H, W, st);
  }
  rc_set_error("upsampling factor %d is not one of 1, 2, 4, 8", factor);
  return RC_EUNSUPPORTED;
}

}  // namespace

#define RC_PW_DISPATCH(dtype, CALL)                           \
  switch (dtype) {                                            \
    case RC_F32: { using T = float; return CALL; }            \
    case RC_F16: { using T = rc_half; return CALL; }          \
    case RC_BF16: { using T = rc_bf16; return CALL; }         \
  }                                                           \
  return RC_EUNSUPPORTED;

extern "C" int rc_gru_gate_r(const void* convr_out, int64_t convr_batch_stride, const void* cr,
                             int64_t cr_batch_stride, const void* h, void* rh, int64_t rh_batch_stride, int64_t N, int64_t per_batch, int dtype,
                             void* stream) {
  RC_REQUIRE(convr_out && cr && h && rh, RC_EINVAL, "null pointer argument");
  int rc = pw_check(N, per_batch, dtype);
  if (rc) return rc;
  RC_REQUIRE(convr_batch_stride >= per_batch && cr_batch_stride >= per_batch && rh_batch_stride >= per_batch,
             RC_EINVAL, "a batch stride is shorter than one image (%lld)", (long long)per_batch);
  hipStream_t st = (hipStream_t)stream;
  RC_PW_DISPATCH(dtype, (launch_gate_r<T>(convr_out, convr_batch_stride, cr, cr_batch_stride, h, rh, rh_batch_stride, N, per_batch, st)))
}

extern "C" int rc_gru_update(const void* convz_out, int64_t convz_batch_stride, const void* cz,
                             int64_t cz_batch_stride, const void* convq_out, int64_t convq_batch_stride,
                             const void* cq, int64_t cq_batch_stride, const void* h, void* h_out, int64_t N,
                             int64_t per_batch, int dtype, void* stream) {
  RC_REQUIRE(convz_out && cz && convq_out && cq && h && h_out, RC_EINVAL, "null pointer argument");
  int rc = pw_check(N, per_batch, dtype);
  if (rc) return rc;
  RC_REQUIRE(cz_batch_stride >= per_batch && cq_batch_stride >= per_batch && convz_batch_stride >= per_batch &&
                 convq_batch_stride >= per_batch, RC_EINVAL,
             "a batch stride is shorter than one image (%lld)", (long long)per_batch);
  hipStream_t st = (hipStream_t)stream;
  RC_PW_DISPATCH(dtype, (launch_update<T>(convz_out, convz_batch_stride, cz, cz_batch_stride, convq_out,
                                          convq_batch_stride, cq, cq_batch_stride, h, h_out,
                                          N, per_batch, st)))
}

extern "C" int rc_gru_update_bwd(const void* convz_out, const void* cz, int64_t cz_batch_stride,
                                 const void* convq_out, const void* cq, int64_t cq_batch_stride, const void* h,
                                 const void* grad_h_out, void* grad_z_pre, void* grad_q_pre, void* grad_h, int64_t N,
                                 int64_t per_batch, int dtype, void* stream) {
  RC_REQUIRE(convz_out && cz && convq_out && cq && h && grad_h_out && grad_z_pre && grad_q_pre && grad_h, RC_EINVAL,
             "null pointer argument");
  int rc = pw_check(N, per_batch, dtype);
  if (rc) return rc;
  RC_REQUIRE(cz_batch_stride >= per_batch && cq_batch_stride >= per_batch, RC_EINVAL,
             "a batch stride is shorter than one image (%lld)", (long long)per_batch);
  hipStream_t st = (hipStream_t)stream;
  RC_PW_DISPATCH(dtype, (launch_update_bwd<T>(convz_out, cz, cz_batch_stride, convq_out, cq, cq_batch_stride, h,
                                              grad_h_out, grad_z_pre, grad_q_pre, grad_h, N, per_batch, st)))
}

extern "C" int rc_gru_gate_r_bwd(const void* convr_out, const void* cr, int64_t cr_batch_stride, const void* h,
                                 const void* grad_rh, int64_t grad_rh_batch_stride, void* grad_r_pre, void* grad_h,
                                 int accumulate_grad_h, int64_t N, int64_t per_batch, int dtype, void* stream) {
  RC_REQUIRE(convr_out && cr && h && grad_rh && grad_r_pre && grad_h, RC_EINVAL, "null pointer argument");
  int rc = pw_check(N, per_batch, dtype);
  if (rc) return rc;
  RC_REQUIRE(cr_batch_stride >= per_batch && grad_rh_batch_stride >= per_batch, RC_EINVAL,
             "a batch stride is shorter than one image (%lld)", (long long)per_batch);
  hipStream_t st = (hipStream_t)stream;
  RC_PW_DISPATCH(dtype, (launch_gate_r_bwd<T>(convr_out, cr, cr_batch_stride, h, grad_rh, grad_rh_batch_stride,
                                              grad_r_pre, grad_h, accumulate_grad_h, N, per_batch, st)))
}

extern "C" int rc_upsample_flow_fwd(const float* flow, int64_t flow_batch_stride, const void* mask, float* out,
                                    int64_t N, int channels, int64_t H, int64_t W, int factor, int mask_dtype,
                                    void* stream) {
  RC_REQUIRE(flow && mask && out, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(N > 0 && H > 0 && W > 0, RC_EINVAL, "non-positive size N=%lld H=%lld W=%lld", (long long)N,
             (long long)H, (long long)W);
  RC_REQUIRE(channels == 1 || channels == 2, RC_EUNSUPPORTED, "channels must be 1 or 2 (got %d)", channels);
  RC_REQUIRE(flow_batch_stride >= (int64_t)channels * H * W, RC_EINVAL, "flow batch stride shorter than %d planes",
             channels);
  RC_REQUIRE(factor >= 1 && (int64_t)9 * factor * factor * H * W < (1ll << 30), RC_EINVAL,
             "one image's mask exceeds the 32-bit lane offset range");
  RC_REQUIRE(((uintptr_t)out & 15) == 0, RC_EINVAL, "out must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (mask_dtype) {
    case RC_F32: return launch_upsample_f<float>(flow, flow_batch_stride, mask, out, N, channels, H, W, factor, st);
    case RC_F16: return launch_upsample_f<rc_half>(flow, flow_batch_stride, mask, out, N, channels, H, W, factor, st);
    case RC_BF16: return launch_upsample_f<rc_bf16>(flow, flow_batch_stride, mask, out, N, channels, H, W, factor, st);
  }
  rc_set_error("unknown mask dtype %d", mask_dtype);
  return RC_EUNSUPPORTED;
}

extern "C" int64_t rc_upsample_flow_bwd_work_bytes(int64_t N, int channels, int64_t H, int64_t W) {
  if (N <= 0 || H <= 0 || W <= 0 || (channels != 1 && channels != 2)) return -1;
  return N * channels * 9 * H * W * (int64_t)sizeof(float);
}

extern "C" int rc_upsample_flow_bwd(const float* flow, int64_t flow_batch_stride, const void* mask,
                                    const float* grad_out, void* grad_mask, float* grad_flow,
                                    int64_t grad_flow_batch_stride, void* work, int64_t N, int channels, int64_t H,
                                    int64_t W, int factor, int mask_dtype, void* stream) {
  RC_REQUIRE(flow && mask && grad_out && grad_mask && grad_flow && work, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(N > 0 && H > 0 && W > 0, RC_EINVAL, "non-positive size N=%lld H=%lld W=%lld", (long long)N,
             (long long)H, (long long)W);
  RC_REQUIRE(channels == 1 || channels == 2, RC_EUNSUPPORTED, "channels must be 1 or 2 (got %d)", channels);
  RC_REQUIRE(flow_batch_stride >= (int64_t)channels * H * W && grad_flow_batch_stride >= (int64_t)channels * H * W,
             RC_EINVAL, "a flow batch stride is shorter than %d planes", channels);
  RC_REQUIRE(factor >= 1 && (int64_t)9 * factor * factor * H * W < (1ll << 30), RC_EINVAL,
             "one image's mask exceeds the 32-bit lane offset range");
  RC_REQUIRE(((uintptr_t)grad_out & 15) == 0 && ((uintptr_t)work & 15) == 0, RC_EINVAL,
             "grad_out and work must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  float* wk = (float*)work;
  switch (mask_dtype) {
    case RC_F32:
      return launch_upsample_bwd_f<float>(flow, flow_batch_stride, mask, grad_out, grad_mask, grad_flow,
                                          grad_flow_batch_stride, wk, N, channels, H, W, factor, st);
    case RC_F16:
      return launch_upsample_bwd_f<rc_half>(flow, flow_batch_stride, mask, grad_out, grad_mask, grad_flow,
                                            grad_flow_batch_stride, wk, N, channels, H, W, factor, st);
    case RC_BF16:
      return launch_upsample_bwd_f<rc_bf16>(flow, flow_batch_stride, mask, grad_out, grad_mask, grad_flow,
                                            grad_flow_batch_stride, wk, N, channels, H, W, factor, st);
  }
  rc_set_error("unknown mask dtype %d", mask_dtype);
  return RC_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------------- sequence loss
// train_stereo.py:35-70 (sequence_loss): one pass over all n predictions instead of ~8 torch ops and a
// boolean-mask gather (a host synchronisation) per prediction.
//   valid' = (valid >= 0.5) & (|flow_gt| < max_flow)            (one flow channel: stereo)
//   loss   = sum_i w_i * mean over valid' of |pred_i - flow_gt|
//   epe    = |pred_last - flow_gt| over valid':  mean, fraction < 1, < 3, < 5
// Kernel 1: grid (pixel blocks, chunks of SL_P predictions); a block strides over pixels and reduces its
// SL_P sums (and, in the chunks that own them, the pixel count and the metrics of the last prediction) to
// one float partial each.  Kernel 2: one wave per quantity adds the partials in double in a fixed order
// (reproducible run to run), then the loss is formed.  results (double):
//   [0,n) sum_i |pred_i - gt|, [n] count, [n+1] sum epe, [n+2..n+4] #epe<1,<3,<5, [n+5] loss, [n+6] non-finite
//   values seen in the predictions (the reference asserts there are none, train_stereo.py:51-52)
namespace {

constexpr int SL_P = 8;            // predictions per block
constexpr int SL_MAX_BLOCKS = 512; // pixel blocks (partials per quantity)
constexpr int SL_EXTRA = 5;        // count, epe sum, <1, <3, <5

struct sl_ptrs {
  const float* pred[RC_MAX_FUSED_LOOKUPS];
  float* grad[RC_MAX_FUSED_LOOKUPS];
  float weight[RC_MAX_FUSED_LOOKUPS];
};

__device__ __forceinline__ float sl_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the block's 4 waves; the result is valid in thread 0
__device__ __forceinline__ float sl_block_sum(float v, float* red) {
  v = sl_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int VEC>
__global__ __launch_bounds__(PW_THREADS) void seqloss_partial_kernel(sl_ptrs ptrs, int n, int64_t pred_bs,
                                                                     const float* __restrict__ gt,
                                                                     const float* __restrict__ valid, float max_flow,
                                                                     float* __restrict__ partial, int64_t HW,
                                                                     int64_t total) {
  __shared__ float red[4];
  const int chunk = blockIdx.y, i0 = chunk * SL_P;
  const int nblk = gridDim.x;
  float acc[SL_P], cnt = 0.f, epe = 0.f, n1 = 0.f, n3 = 0.f, n5 = 0.f, bad = 0.f;
#pragma unroll
  for (int j = 0; j < SL_P; ++j) acc[j] = 0.f;
  const int jlast = n - 1 - i0;  // slot of the last prediction, if this chunk holds it
  // VEC consecutive pixels per thread and step (whole 16-byte pieces of one image when VEC = 4)
  typedef typename pw_vec<float, VEC>::type fv;
  for (int64_t p = ((int64_t)blockIdx.x * PW_THREADS + threadIdx.x) * VEC; p < total;
       p += (int64_t)nblk * PW_THREADS * VEC) {
    const int64_t b = p / HW, q = p - b * HW;
    const fv g = pw_load<float, VEC>(gt + p), va = pw_load<float, VEC>(valid + p);
    bool ok[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      ok[e] = (va[e] >= 0.5f) && (sqrtf(g[e] * g[e]) < max_flow);
      cnt += ok[e] ? 1.f : 0.f;
    }
    // all of the chunk's predictions first (the loads of later ones do not move above the branch that consumes an
    // earlier one by themselves: one memory latency per prediction instead of one per pixel step)
    fv xs[SL_P];
#pragma unroll
    for (int j = 0; j < SL_P; ++j) {
      const int i = min(i0 + j, n - 1);  // past the end: re-read the last one, not accumulated
      xs[j] = pw_load<float, VEC>(ptrs.pred[i] + b * pred_bs + q);
    }
#pragma unroll
    for (int j = 0; j < SL_P; ++j) {
      const fv x = xs[j];
      if (i0 + j < n) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          acc[j] += ok[e] ? fabsf(x[e] - g[e]) : 0.f;
          bad += (fabsf(x[e]) <= 3.0e38f) ? 0.f : 1.f;  // false for inf and nan
          if (j == jlast && ok[e]) {
            const float er = sqrtf((x[e] - g[e]) * (x[e] - g[e]));
            epe += er;
            n1 += er < 1.f ? 1.f : 0.f;
            n3 += er < 3.f ? 1.f : 0.f;
            n5 += er < 5.f ? 1.f : 0.f;
          }
        }
      }
    }
  }
  // partial[q][block]: quantity q of this pixel block
  const int nq = n + SL_EXTRA + gridDim.y;
#pragma unroll
  for (int j = 0; j < SL_P; ++j) {
    const float s = sl_block_sum(acc[j], red);
    if (threadIdx.x == 0 && i0 + j < n) partial[(int64_t)(i0 + j) * nblk + blockIdx.x] = s;
  }
  if (chunk == 0) {
    const float s = sl_block_sum(cnt, red);
    if (threadIdx.x == 0) partial[(int64_t)n * nblk + blockIdx.x] = s;
  }
  if (jlast >= 0 && jlast < SL_P) {
    const float s0 = sl_block_sum(epe, red), s1 = sl_block_sum(n1, red), s3 = sl_block_sum(n3, red),
                s5 = sl_block_sum(n5, red);
    if (threadIdx.x == 0) {
      partial[(int64_t)(n + 1) * nblk + blockIdx.x] = s0;
      partial[(int64_t)(n + 2) * nblk + blockIdx.x] = s1;
      partial[(int64_t)(n + 3) * nblk + blockIdx.x] = s3;
      partial[(int64_t)(n + 4) * nblk + blockIdx.x] = s5;
    }
  }
  {
    const float s = sl_block_sum(bad, red);
    if (threadIdx.x == 0) partial[(int64_t)(n + SL_EXTRA + chunk) * nblk + blockIdx.x] = s;
  }
  (void)nq;
}

__global__ __launch_bounds__(1024) void seqloss_final_kernel(const float* __restrict__ partial, int nblk, int n,
                                                             int nchunks, sl_ptrs ptrs, double* __restrict__ results) {
  __shared__ double q[RC_MAX_FUSED_LOOKUPS + SL_EXTRA + RC_MAX_FUSED_LOOKUPS / SL_P + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int nq = n + SL_EXTRA + nchunks;
  for (int k = wave; k < nq; k += nw) {
    double s = 0.0;
    for (int t = lane; t < nblk; t += 64) s += (double)partial[(int64_t)k * nblk + t];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) q[k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double loss = 0.0, bad = 0.0;
    for (int i = 0; i < n; ++i) {
      results[i] = q[i];
      loss += (double)ptrs.weight[i] * (q[i] / q[n]);  // mean over the valid pixels; 0/0 = nan like torch's empty mean
    }
    for (int k = 0; k < SL_EXTRA; ++k) results[n + k] = q[n + k];
    for (int c = 0; c < nchunks; ++c) bad += q[n + SL_EXTRA + c];
    results[n + 5] = loss;
    results[n + 6] = bad;
  }
}

// d loss / d pred_i = grad_loss * w_i * sign(pred_i - gt) / count on valid' pixels, 0 elsewhere
template <int VEC>
__global__ __launch_bounds__(PW_THREADS) void seqloss_bwd_kernel(sl_ptrs ptrs, int n, int64_t pred_bs, int64_t grad_bs,
                                                                 const float* __restrict__ gt,
                                                                 const float* __restrict__ valid, float max_flow,
                                                                 const double* __restrict__ results,
                                                                 const float* __restrict__ grad_loss, int64_t HW,
                                                                 int64_t total) {
  typedef typename pw_vec<float, VEC>::type fv;
  const int64_t p = ((int64_t)blockIdx.x * PW_THREADS + threadIdx.x) * VEC;
  if (p >= total) return;
  const int64_t b = p / HW, q = p - b * HW;
  const fv g = pw_load<float, VEC>(gt + p), va = pw_load<float, VEC>(valid + p);
  const float scale = grad_loss[0] / (float)results[n];
  float m[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) m[e] = ((va[e] >= 0.5f) && (sqrtf(g[e] * g[e]) < max_flow)) ? scale : 0.f;
  for (int i = 0; i < n; ++i) {  // i is uniform: pointers and weights come through scalar loads
    const fv x = pw_load<float, VEC>(ptrs.pred[i] + b * pred_bs + q);
    const float w = ptrs.weight[i];
    fv o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = x[e] - g[e];
      o[e] = m[e] * w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    *reinterpret_cast<fv*>(ptrs.grad[i] + b * grad_bs + q) = o;
  }
}

inline int sl_blocks(int64_t total) {   // total = work items (pixels or 4-pixel vectors)
  const int64_t nb = (total + PW_THREADS - 1) / PW_THREADS;
  return (int)(nb < SL_MAX_BLOCKS ? nb : SL_MAX_BLOCKS);
}

}  // namespace

extern "C" int64_t rc_sequence_loss_work_bytes(int n_predictions) {
  if (n_predictions < 1 || n_predictions > RC_MAX_FUSED_LOOKUPS) return -1;
  const int nchunks = (n_predictions + SL_P - 1) / SL_P;
  return (int64_t)(n_predictions + SL_EXTRA + nchunks) * SL_MAX_BLOCKS * (int64_t)sizeof(float);
}

extern "C" int rc_sequence_loss_fwd(const float* const* preds, const float* weights, int n_predictions,
                                    int64_t pred_batch_stride, const float* flow_gt, const float* valid,
                                    float max_flow, double* results, void* work, int64_t N, int64_t HW,
                                    void* stream) {
  RC_REQUIRE(preds && weights && flow_gt && valid && results && work, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(n_predictions >= 1 && n_predictions <= RC_MAX_FUSED_LOOKUPS, RC_EUNSUPPORTED,
             "n_predictions %d outside 1..%d", n_predictions, RC_MAX_FUSED_LOOKUPS);
  RC_REQUIRE(N > 0 && HW > 0 && N * HW < (1ll << 40) && pred_batch_stride >= HW, RC_EINVAL,
             "bad sizes N=%lld HW=%lld stride=%lld", (long long)N, (long long)HW, (long long)pred_batch_stride);
  sl_ptrs ptrs;
  for (int i = 0; i < RC_MAX_FUSED_LOOKUPS; ++i) {
    const int s = i < n_predictions ? i : 0;
    RC_REQUIRE(preds[s], RC_EINVAL, "null pointer in the prediction list");
    ptrs.pred[i] = preds[s];
    ptrs.grad[i] = nullptr;
    ptrs.weight[i] = i < n_predictions ? weights[i] : 0.f;
  }
  const int64_t total = N * HW;
  bool vec = HW % 4 == 0 && pred_batch_stride % 4 == 0 && aligned16(flow_gt) && aligned16(valid);
  for (int i = 0; i < n_predictions; ++i) vec = vec && aligned16(preds[i]);
  const int nblk = sl_blocks(vec ? total / 4 : total), nchunks = (n_predictions + SL_P - 1) / SL_P;
  hipStream_t st = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL(seqloss_partial_kernel<4>, dim3(nblk, nchunks), dim3(PW_THREADS), 0, st, ptrs, n_predictions,
                       pred_batch_stride, flow_gt, valid, max_flow, (float*)work, HW, total);
  else
    hipLaunchKernelGGL(seqloss_partial_kernel<1>, dim3(nblk, nchunks), dim3(PW_THREADS), 0, st, ptrs, n_predictions,
                       pred_batch_stride, flow_gt, valid, max_flow, (float*)work, HW, total);
  int rc = rc_check_launch("rc_sequence_loss_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(seqloss_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)work, nblk, n_predictions,
                     nchunks, ptrs, results);
  return rc_check_launch("rc_sequence_loss_fwd(final)");
}

extern "C" int rc_sequence_loss_bwd(const float* const* preds, float* const* grad_preds, const float* weights,
                                    int n_predictions, int64_t pred_batch_stride, int64_t grad_batch_stride,
                                    const float* flow_gt, const float* valid, float max_flow, const double* results,
                                    const float* grad_loss, int64_t N, int64_t HW, void* stream) {
  RC_REQUIRE(preds && grad_preds && weights && flow_gt && valid && results && grad_loss, RC_EINVAL,
             "null pointer argument");
  RC_REQUIRE(n_predictions >= 1 && n_predictions <= RC_MAX_FUSED_LOOKUPS, RC_EUNSUPPORTED,
             "n_predictions %d outside 1..%d", n_predictions, RC_MAX_FUSED_LOOKUPS);
  RC_REQUIRE(N > 0 && HW > 0 && N * HW < (1ll << 40) && pred_batch_stride >= HW && grad_batch_stride >= HW, RC_EINVAL,
             "bad sizes N=%lld HW=%lld", (long long)N, (long long)HW);
  sl_ptrs ptrs;
  for (int i = 0; i < RC_MAX_FUSED_LOOKUPS; ++i) {
    const int s = i < n_predictions ? i : 0;
    RC_REQUIRE(preds[s] && grad_preds[s], RC_EINVAL, "null pointer in the prediction lists");
    ptrs.pred[i] = preds[s];
    ptrs.grad[i] = grad_preds[s];
    ptrs.weight[i] = i < n_predictions ? weights[i] : 0.f;
  }
  const int64_t total = N * HW;
  bool vec = HW % 4 == 0 && pred_batch_stride % 4 == 0 && grad_batch_stride % 4 == 0 && aligned16(flow_gt) &&
             aligned16(valid);
  for (int i = 0; i < n_predictions; ++i) vec = vec && aligned16(preds[i]) && aligned16(grad_preds[i]);
  const int64_t items = vec ? total / 4 : total;
  const int64_t nb = (items + PW_THREADS - 1) / PW_THREADS;
  RC_REQUIRE(nb < (1ll << 31), RC_EINVAL, "too many pixels");
  if (vec)
    hipLaunchKernelGGL(seqloss_bwd_kernel<4>, dim3((unsigned)nb), dim3(PW_THREADS), 0, (hipStream_t)stream, ptrs,
                       n_predictions, pred_batch_stride, grad_batch_stride, flow_gt, valid, max_flow, results,
                       grad_loss, HW, total);
  else
    hipLaunchKernelGGL(seqloss_bwd_kernel<1>, dim3((unsigned)nb), dim3(PW_THREADS), 0, (hipStream_t)stream, ptrs,
                       n_predictions, pred_batch_stride, grad_batch_stride, flow_gt, valid, max_flow, results,
                       grad_loss, HW, total);
  return rc_check_launch("rc_sequence_loss_bwd");
}

// ------------------------------------------------------------------------ between-scale glue
// core/update.py:87-95: pool2x = avg_pool2d(x, 3, stride=2, padding=1) (count_include_pad: always / 9) feeds the
// coarser GRU, interp = bilinear resize with align_corners=True feeds the finer one.  Streaming kernels, one
// output element per thread along W (coalesced), fp32 arithmetic, result rounded to the tensor type; the output
// may be a channel slice of the buffer the GRU's convolutions read (batch stride argument).
namespace {

// One thread = VEC consecutive outputs of one output row (VEC * sizeof(T) = 16 bytes when rows are whole
// vectors, else 1): a flat index over all output vectors keeps every block full whatever the row length.
template <typename T, int VEC>
__global__ __launch_bounds__(PW_THREADS) void pool2x_kernel(const T* __restrict__ x, T* __restrict__ out, int H, int W,
                                                            int Ho, int Wo, int C, int64_t out_bs, int64_t total,
                                                            int vec_in) {
  const int64_t v = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;
  if (v >= total) return;
  const int wv = Wo / VEC;
  const int wo0 = (int)(v % wv) * VEC;
  const int64_t r = v / wv;
  const int ho = (int)(r % Ho);
  const int64_t nc = r / Ho, n = nc / C, c = nc - n * C;
  const T* src = x + nc * (int64_t)H * W;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  // Every load is unconditional, from a clamped row / column, and what lies outside the image is zeroed after:
  // a load under a bounds test is a branch, and the compiler waits for each one where the branches join (one
  // memory latency per load instead of one per thread).
  float in[3][2 * VEC + 1];
  if (vec_in) {  // launch-uniform: input rows are whole, aligned vectors of 2 VEC elements
    typename pw_vec<T, 2 * VEC>::type vr[3];
    T left[3];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const T* row = src + (int64_t)min(max(2 * ho + dy, 0), H - 1) * W;
      vr[dy + 1] = pw_load<T, 2 * VEC>(row + 2 * wo0);
      left[dy + 1] = row[max(2 * wo0 - 1, 0)];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      in[d][0] = wo0 > 0 ? rc_to_float(left[d]) : 0.f;
#pragma unroll
      for (int q = 0; q < 2 * VEC; ++q) in[d][q + 1] = rc_to_float(vr[d][q]);
    }
  } else {
    T raw[3][2 * VEC + 1];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const T* row = src + (int64_t)min(max(2 * ho + dy, 0), H - 1) * W;
#pragma unroll
      for (int q = 0; q < 2 * VEC + 1; ++q) raw[dy + 1][q] = row[min(max(2 * wo0 - 1 + q, 0), W - 1)];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int q = 0; q < 2 * VEC + 1; ++q)
        in[d][q] = (unsigned)(2 * wo0 - 1 + q) < (unsigned)W ? rc_to_float(raw[d][q]) : 0.f;
  }
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    if ((unsigned)(2 * ho + dy) >= (unsigned)H) continue;  // (no loads inside: a plain skip of three additions)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] += (in[dy + 1][2 * j] + in[dy + 1][2 * j + 1]) + in[dy + 1][2 * j + 2];
  }
  typename pw_vec<T, VEC>::type o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) o[j] = rc_from_float<T>(acc[j] * (1.0f / 9.0f));
  *reinterpret_cast<typename pw_vec<T, VEC>::type*>(out + n * out_bs + (c * Ho + ho) * (int64_t)Wo + wo0) = o;
}

// One thread = VEC consecutive outputs of INTERP_ROWS consecutive output rows.  The horizontal blends of an input
// row are kept in registers and reused by every output row between the same two input rows (when upsampling by
// two, 4 output rows read 3 input rows instead of 8); each value is computed by the same expression whichever
// output row asks first, so the result does not depend on the grouping.
constexpr int INTERP_ROWS = 4;

template <typename T, int VEC>
__global__ __launch_bounds__(PW_THREADS) void interp_kernel(const T* __restrict__ x, T* __restrict__ out, int H, int W,
                                                            int Ho, int Wo, int C, int64_t out_bs, float sy, float sx,
                                                            int64_t total) {
  const int64_t v = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;
  if (v >= total) return;
  const int wv = Wo / VEC, hg = (Ho + INTERP_ROWS - 1) / INTERP_ROWS;
  const int wo0 = (int)(v % wv) * VEC;
  const int64_t r = v / wv;
  const int ho0 = (int)(r % hg) * INTERP_ROWS;
  const int64_t nc = r / hg, n = nc / C, c = nc - n * C;
  const T* src = x + nc * (int64_t)H * W;
  // align_corners=True: source position = destination index * (in - 1) / (out - 1), in fp32 like torch
  int x0[VEC], x1[VEC];
  float lx[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float fx = sx * (float)(wo0 + j);
    x0[j] = min((int)fx, W - 1);
    x1[j] = min(x0[j] + 1, W - 1);
    lx[j] = fx - (float)x0[j];
  }
  auto blend_row = [&](int y, float* h) {
    const T* row = src + (int64_t)y * W;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float hx = 1.0f - lx[j];
      h[j] = hx * rc_to_float(row[x0[j]]) + lx[j] * rc_to_float(row[x1[j]]);
    }
  };
  float top[VEC], bot[VEC];
  int ty = -1, by = -1;  // the input rows `top` and `bot` hold
  T* dst = out + n * out_bs + c * (int64_t)Ho * Wo + wo0;
#pragma unroll
  for (int k = 0; k < INTERP_ROWS; ++k) {
    const int ho = ho0 + k;
    if (ho >= Ho) break;
    const float fy = sy * (float)ho;
    const int y0 = min((int)fy, H - 1), y1 = min(y0 + 1, H - 1);
    const float ly = fy - (float)y0, hy = 1.0f - ly;
    if (y0 != ty) {
      if (y0 == by) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) top[j] = bot[j];
      } else {
        blend_row(y0, top);
      }
      ty = y0;
    }
    if (y1 != by) {
      if (y1 == ty) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) bot[j] = top[j];
      } else {
        blend_row(y1, bot);
      }
      by = y1;
    }
    typename pw_vec<T, VEC>::type o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = rc_from_float<T>(hy * top[j] + ly * bot[j]);
    *reinterpret_cast<typename pw_vec<T, VEC>::type*>(dst + (int64_t)ho * Wo) = o;
  }
}

template <typename T, int V>
int launch_resample_v(int which, const void* x, void* out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Ho,
                      int64_t Wo, int64_t out_bs, hipStream_t st) {
  const int64_t rows = which == 0 ? Ho : (Ho + INTERP_ROWS - 1) / INTERP_ROWS;  // interp: a thread walks 4 rows
  const int64_t total = N * C * rows * (Wo / V);
  const int64_t nb = (total + PW_THREADS - 1) / PW_THREADS;
  RC_REQUIRE(nb < (1ll << 31), RC_EINVAL, "too many elements");
  const dim3 grid((unsigned)nb);
  if (which == 0) {
    const int vec_in = W % (2 * V) == 0 && (uintptr_t)x % (2 * V * sizeof(T)) == 0;
    hipLaunchKernelGGL((pool2x_kernel<T, V>), grid, dim3(PW_THREADS), 0, st, (const T*)x, (T*)out, (int)H, (int)W,
                       (int)Ho, (int)Wo, (int)C, out_bs, total, vec_in);
    return rc_check_launch("rc_pool2x");
  }
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  hipLaunchKernelGGL((interp_kernel<T, V>), grid, dim3(PW_THREADS), 0, st, (const T*)x, (T*)out, (int)H, (int)W,
                     (int)Ho, (int)Wo, (int)C, out_bs, sy, sx, total);
  return rc_check_launch("rc_interp_bilinear");
}

// the widest store (16, 8 or 4 bytes of consecutive outputs) that output rows and the output pointer allow
template <typename T>
int launch_resample(int which, const void* x, void* out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Ho,
                    int64_t Wo, int64_t out_bs, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  const uintptr_t addr = (uintptr_t)out;
  auto fits = [&](int v) { return Wo % v == 0 && out_bs % v == 0 && addr % (v * sizeof(T)) == 0; };
  if (fits(V)) return launch_resample_v<T, V>(which, x, out, N, C, H, W, Ho, Wo, out_bs, st);
  if (fits(V / 2)) return launch_resample_v<T, V / 2>(which, x, out, N, C, H, W, Ho, Wo, out_bs, st);
  if constexpr (V >= 8) {
    if (fits(V / 4)) return launch_resample_v<T, V / 4>(which, x, out, N, C, H, W, Ho, Wo, out_bs, st);
  }
  return launch_resample_v<T, 1>(which, x, out, N, C, H, W, Ho, Wo, out_bs, st);
}

int resample_any(int which, const void* x, void* out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Ho,
                 int64_t Wo, int64_t out_bs, int dtype, void* stream) {
  RC_REQUIRE(x && out, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && H < (1 << 24) && W < (1 << 24), RC_EINVAL,
             "bad sizes");
  RC_REQUIRE(out_bs >= C * Ho * Wo, RC_EINVAL, "output batch stride shorter than one image");
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case RC_F32: return launch_resample<float>(which, x, out, N, C, H, W, Ho, Wo, out_bs, st);
    case RC_F16: return launch_resample<rc_half>(which, x, out, N, C, H, W, Ho, Wo, out_bs, st);
    case RC_BF16: return launch_resample<rc_bf16>(which, x, out, N, C, H, W, Ho, Wo, out_bs, st);
  }
  rc_set_error("unknown dtype %d", dtype);
  return RC_EUNSUPPORTED;
}

}  // namespace

extern "C" int rc_pool2x(const void* x, void* out, int64_t N, int64_t C, int64_t H, int64_t W,
                         int64_t out_batch_stride, int dtype, void* stream) {
  return resample_any(0, x, out, N, C, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, out_batch_stride, dtype, stream);
}

extern "C" int rc_interp_bilinear(const void* x, void* out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Ho,
                                  int64_t Wo, int64_t out_batch_stride, int dtype, void* stream) {
  return resample_any(1, x, out, N, C, H, W, Ho, Wo, out_batch_stride, dtype, stream);
}

// ---- convolution epilogue (inference): x = act(x + bias[c]) in place.  torch's MIOpen path adds a convolution's
// bias as a separate broadcast add_ (a strided-iterator kernel) and the ReLU after it as another pass; this is both
// in one pass of 16-byte accesses with the same roundings: the bias is rounded to x's type first (what autocast's
// cast of the parameter gives), the sum is rounded to x's type, ReLU keeps NaN.
namespace {

template <typename T, typename BT, int VEC>
__global__ __launch_bounds__(PW_THREADS) void bias_act_kernel(T* __restrict__ x, const BT* __restrict__ bias, int C,
                                                              int64_t hwv, int relu, int64_t total) {
  const int64_t v = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;
  if (v >= total) return;
  const int c = (int)((v / hwv) % C);
  const float b = rnd<T>(rc_to_float(bias[c]));
  auto* p = reinterpret_cast<typename pw_vec<T, VEC>::type*>(x) + v;
  auto val = *p;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float f = rnd<T>(rc_to_float(val[j]) + b);
    if (relu) f = f < 0.f ? 0.f : f;
    val[j] = rc_from_float<T>(f);
  }
  *p = val;
}

template <typename T, typename BT, int V>
int launch_bias_act_v(void* x, const void* bias, int64_t N, int64_t C, int64_t HW, int relu, hipStream_t st) {
  const int64_t total = N * C * (HW / V);
  const int64_t nb = (total + PW_THREADS - 1) / PW_THREADS;
  RC_REQUIRE(nb < (1ll << 31), RC_EINVAL, "too many elements");
  hipLaunchKernelGGL((bias_act_kernel<T, BT, V>), dim3((unsigned)nb), dim3(PW_THREADS), 0, st, (T*)x, (const BT*)bias,
                     (int)C, HW / V, relu, total);
  return rc_check_launch("rc_bias_act");
}

template <typename T, typename BT>
int launch_bias_act(void* x, const void* bias, int64_t N, int64_t C, int64_t HW, int relu, hipStream_t st) {
  constexpr int V = 16 / (int)sizeof(T);
  auto fits = [&](int v) { return HW % v == 0 && (uintptr_t)x % (v * sizeof(T)) == 0; };
  if (fits(V)) return launch_bias_act_v<T, BT, V>(x, bias, N, C, HW, relu, st);
  if (fits(V / 2)) return launch_bias_act_v<T, BT, V / 2>(x, bias, N, C, HW, relu, st);
  if constexpr (V >= 8) {
    if (fits(V / 4)) return launch_bias_act_v<T, BT, V / 4>(x, bias, N, C, HW, relu, st);
  }
  return launch_bias_act_v<T, BT, 1>(x, bias, N, C, HW, relu, st);
}

}  // namespace

extern "C" int rc_bias_act(void* x, const void* bias, int bias_dtype, int64_t N, int64_t C, int64_t HW, int dtype,
                           int relu, void* stream) {
  RC_REQUIRE(x && bias, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(N > 0 && C > 0 && C < (1 << 30) && HW > 0, RC_EINVAL, "bad sizes");
  RC_REQUIRE(bias_dtype == dtype || bias_dtype == RC_F32, RC_EUNSUPPORTED,
             "the bias must have x's type or be RC_F32 (got %d for x of type %d)", bias_dtype, dtype);
  hipStream_t st = (hipStream_t)stream;
  const bool bf = bias_dtype == RC_F32;
  switch (dtype) {
    case RC_F32: return launch_bias_act<float, float>(x, bias, N, C, HW, relu, st);
    case RC_F16:
      return bf ? launch_bias_act<rc_half, float>(x, bias, N, C, HW, relu, st)
                : launch_bias_act<rc_half, rc_half>(x, bias, N, C, HW, relu, st);
    case RC_BF16:
      return bf ? launch_bias_act<rc_bf16, float>(x, bias, N, C, HW, relu, st)
                : launch_bias_act<rc_bf16, rc_bf16>(x, bias, N, C, HW, relu, st);
  }
  rc_set_error("unknown dtype %d", dtype);
  return RC_EUNSUPPORTED;
}

// ---- adjoints of the between-scale glue (training): gathers, no atomics.  One thread per INPUT element.
namespace {

template <typename T>
__global__ __launch_bounds__(PW_THREADS) void pool2x_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int H, int W,
                                                                int Ho, int Wo, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;
  if (idx >= total) return;
  const int w = (int)(idx % W);
  const int64_t r = idx / W;
  const int h = (int)(r % H);
  const int64_t nc = r / H;
  const T* src = g + nc * (int64_t)Ho * Wo;
  // outputs whose 3x3 window (centre 2 ho, 2 wo) covers (h, w): |2 ho - h| <= 1
  const int ho0 = h >> 1, ho1 = (h & 1) ? ho0 + 1 : ho0;
  const int wo0 = w >> 1, wo1 = (w & 1) ? wo0 + 1 : wo0;
  float acc = 0.f;
  for (int ho = ho0; ho <= ho1; ++ho) {
    if (ho >= Ho) continue;
    for (int wo = wo0; wo <= wo1; ++wo)
      if (wo < Wo) acc += rc_to_float(src[(int64_t)ho * Wo + wo]);
  }
  dx[idx] = rc_from_float<T>(acc * (1.0f / 9.0f));
}

// weight of input index `i` in output index `o` along one axis (the forward's own fp32 arithmetic)
__device__ __forceinline__ float interp_axis_weight(int o, int i, float scale, int n_in) {
  const float f = scale * (float)o;
  const int i0 = min((int)f, n_in - 1), i1 = min(i0 + 1, n_in - 1);
  const float l = f - (float)i0;
  return (i0 == i ? 1.0f - l : 0.f) + (i1 == i ? l : 0.f);
}

template <typename T>
__global__ __launch_bounds__(PW_THREADS) void interp_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int H, int W,
                                                                int Ho, int Wo, float sy, float sx, float isy, float isx,
                                                                int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * PW_THREADS + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  const int64_t r = idx / W;
  const int y = (int)(r % H);
  const int64_t nc = r / H;
  const T* src = g + nc * (int64_t)Ho * Wo;
  // outputs that can touch input row y: source position in (y - 1, y + 1); one index of slack on both sides,
  // the exact weight decides
  const int ho_lo = sy > 0.f ? max(0, (int)floorf(((float)y - 1.0f) * isy) - 1) : 0;
  const int ho_hi = sy > 0.f ? min(Ho - 1, (int)ceilf(((float)y + 1.0f) * isy) + 1) : Ho - 1;
  const int wo_lo = sx > 0.f ? max(0, (int)floorf(((float)x - 1.0f) * isx) - 1) : 0;
  const int wo_hi = sx > 0.f ? min(Wo - 1, (int)ceilf(((float)x + 1.0f) * isx) + 1) : Wo - 1;
  float acc = 0.f;
  for (int ho = ho_lo; ho <= ho_hi; ++ho) {
    const float wy = interp_axis_weight(ho, y, sy, H);
    if (wy == 0.f) continue;
    float row = 0.f;
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      const float wx = interp_axis_weight(wo, x, sx, W);
      if (wx != 0.f) row += wx * rc_to_float(src[(int64_t)ho * Wo + wo]);
    }
    acc += wy * row;
  }
  dx[idx] = rc_from_float<T>(acc);
}

template <typename T>
int launch_resample_bwd(int which, const void* g, void* dx, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Ho,
                        int64_t Wo, hipStream_t st) {
  const int64_t total = N * C * H * W;
  const int64_t nb = (total + PW_THREADS - 1) / PW_THREADS;
  RC_REQUIRE(nb < (1ll << 31), RC_EINVAL, "too many elements");
  if (which == 0) {
    hipLaunchKernelGGL((pool2x_bwd_kernel<T>), dim3((unsigned)nb), dim3(PW_THREADS), 0, st, (const T*)g, (T*)dx, (int)H,
                       (int)W, (int)Ho, (int)Wo, total);
    return rc_check_launch("rc_pool2x_bwd");
  }
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  hipLaunchKernelGGL((interp_bwd_kernel<T>), dim3((unsigned)nb), dim3(PW_THREADS), 0, st, (const T*)g, (T*)dx, (int)H,
                     (int)W, (int)Ho, (int)Wo, sy, sx, sy > 0.f ? 1.0f / sy : 0.f, sx > 0.f ? 1.0f / sx : 0.f, total);
  return rc_check_launch("rc_interp_bilinear_bwd");
}

int resample_bwd_any(int which, const void* g, void* dx, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Ho,
                     int64_t Wo, int dtype, void* stream) {
  RC_REQUIRE(g && dx, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && H < (1 << 24) && W < (1 << 24), RC_EINVAL,
             "bad sizes");
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case RC_F32: return launch_resample_bwd<float>(which, g, dx, N, C, H, W, Ho, Wo, st);
    case RC_F16: return launch_resample_bwd<rc_half>(which, g, dx, N, C, H, W, Ho, Wo, st);
    case RC_BF16: return launch_resample_bwd<rc_bf16>(which, g, dx, N, C, H, W, Ho, Wo, st);
  }
  rc_set_error("unknown dtype %d", dtype);
  return RC_EUNSUPPORTED;
}

}  // namespace

extern "C" int rc_pool2x_bwd(const void* grad_out, void* grad_x, int64_t N, int64_t C, int64_t H, int64_t W,
                             int dtype, void* stream) {
  return resample_bwd_any(0, grad_out, grad_x, N, C, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, dtype, stream);
}

extern "C" int rc_interp_bilinear_bwd(const void* grad_out, void* grad_x, int64_t N, int64_t C, int64_t H, int64_t W,
                                      int64_t Ho, int64_t Wo, int dtype, void* stream) {
  return resample_bwd_any(1, grad_out, grad_x, N, C, H, W, Ho, Wo, dtype, stream);
}

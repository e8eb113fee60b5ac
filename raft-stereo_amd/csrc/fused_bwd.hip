// Adjoint of the fused lookup + convc1 + ReLU (SURVEY.md section 8 f, rank 1, training):
//   act = relu(W corr + b),  corr = lookup(coords)            core/raft_stereo.py:110 -> core/update.py:71,78
// autograd of those lines gives, with gm = grad_act where act > 0 (else 0):
//   grad_corr[k] = sum_o W[o][k] gm[o]     (then into the lookups' fused adjoint, rc_lookup_bwd_dense)
//   grad_W[o][k] = sum_pixels gm[o] corr[k],   grad_b[o] = sum_pixels gm[o]
// The forward never stored the L*(2r+1)-channel tensor; the caller recomputes it with rc_lookup_fwd right
// before this call (one lookup kernel; no 36-channel tensor is kept alive across the GRU iterations).
//
// convc1_bwd_taps_kernel: one pixel per lane, its 64 masked gradients in registers; the weights, transposed
// to [k][o] in the workspace, arrive through scalar loads (the index is wave-uniform), so the 64 K
// multiply-adds per pixel are plain v_fmac with a scalar operand.
// convc1_bwd_weight_kernel: blocks stride over chunks of 64 pixels staged in LDS; thread (o, k mod 4) keeps
// its dW entries in registers; float block partials are added in double in a fixed order by
// convc1_bwd_reduce_kernel (no atomics, reproducible run to run).
#include "rc_common.h"

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
  float gm[CB_OUT];
#pragma unroll
  for (int o = 0; o < CB_OUT; ++o) {
    const float g = rc_to_float(g0[(int64_t)o * HW]);
    gm[o] = (!relu || rc_to_float(a0[(int64_t)o * HW]) > 0.f) ? g : 0.f;
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
#pragma unroll
    for (int i = 0; i < CB_OUT / 4; ++i) {
      const int oo = r0 + 4 * i;
      float g = 0.f;
      if (in) {
        const int64_t e = (b * CB_OUT + oo) * HW + p;
        g = rc_to_float(gact[e]);
        if (relu && !(rc_to_float(act[e]) > 0.f)) g = 0.f;
      }
      Gs[oo][px] = g;
    }
    for (int kk = r0; kk < K; kk += 4) Cs[kk][px] = in ? corr[(b * K + kk) * HW + p] : 0.f;
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

__global__ void convc1_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int K, float* __restrict__ gw,
                                         float* __restrict__ gb) {
  const int n = CB_OUT * K + CB_OUT;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)part[(int64_t)b * n + i];
  if (i < CB_OUT * K) gw[i] = (float)s;
  else gb[i - CB_OUT * K] = (float)s;
}

template <typename T>
int launch_convc1_bwd(const float* corr, const void* act, const void* gact, const float* weight, float* gcorr,
                      float* gw, float* gb, float* work, int64_t B, int K, int64_t HW, int relu, hipStream_t st) {
  float* wt = work;
  float* part = work + CB_KMAX * CB_OUT;
  hipLaunchKernelGGL(convc1_transpose_kernel, dim3((K * CB_OUT + 255) / 256), dim3(256), 0, st, weight, wt, K);
  int rc = rc_check_launch("rc_convc1_bwd(transpose)");
  if (rc) return rc;
  hipLaunchKernelGGL((convc1_bwd_taps_kernel<T>), dim3((unsigned)((HW + 255) / 256), (unsigned)B), dim3(256), 0, st,
                     (const T*)act, (const T*)gact, wt, gcorr, K, HW, relu);
  rc = rc_check_launch("rc_convc1_bwd(taps)");
  if (rc) return rc;
  const int64_t cpi = (HW + CB_PIX - 1) / CB_PIX, n_chunks = B * cpi;
  const int nblk = (int)(n_chunks < CB_BLOCKS ? n_chunks : CB_BLOCKS);
  hipLaunchKernelGGL((convc1_bwd_weight_kernel<T>), dim3(nblk), dim3(256), 0, st, corr, (const T*)act, (const T*)gact,
                     part, K, HW, cpi, n_chunks, relu);
  rc = rc_check_launch("rc_convc1_bwd(weight)");
  if (rc) return rc;
  const int n = CB_OUT * K + CB_OUT;
  hipLaunchKernelGGL(convc1_bwd_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, part, nblk, K, gw, gb);
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

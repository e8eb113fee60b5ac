// Adjoint of the pyramid build: grad pyramid (float, skewed) -> d fmap1, d fmap2.
//
// Autograd of core/corr.py:119-125 (avg_pool2d chain) and :154-156 (einsum, / sqrt(D)).
// Two steps on one stream:
//   1. unskew_fold: reads every 16-byte block once (coalesced), folds levels 1..L-1 into level 0
//      (each pooled cell spreads 2^-i of its gradient over its 2^i children), scales by
//      1/sqrt(D), and writes the dense dC[b,h,w1,w2] and its transpose dCt[b,h,w2,w1] in the
//      compute type, zero padded to a multiple of 32 along the contraction axis.
//   2. two batched "NT" MFMA GEMMs with K-contiguous operands:
//        df1[b,:,h,:] = f2[b,:,h,:] (D x W2)  *  dC [b,h]^T (W2 x W1)
//        df2[b,:,h,:] = f1[b,:,h,:] (D x W1)  *  dCt[b,h]^T (W1 x W2)
#include "rc_common.h"

#include <math.h>
#include <type_traits>

namespace {

typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int UT = 64;  // unskew tile edge

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

template <typename T>
__global__ __launch_bounds__(256) void unskew_fold_kernel(const float* __restrict__ gvol, rc_dev_layout lay,
                                                          T* __restrict__ dC, T* __restrict__ dCt, int W1p,
                                                          int W2p, float scale) {
  __shared__ float tile[UT][UT + 1];  // [w1 - c0][w2 - n0]; pitch 65 keeps the diagonal scatter conflict free
  const int c0 = blockIdx.x * UT, n0 = blockIdx.y * UT;
  const int64_t bh = blockIdx.z;
  const int W1 = lay.W1, W2 = lay.W2;
  // (the gradient pyramid is float: 4 skew rows per 16-byte block)
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int w1 = c0 + col;

  for (int e = threadIdx.x; e < UT * (UT + 1); e += 256) (&tile[0][0])[e] = 0.f;
  __syncthreads();

  float wgt = scale;
  for (int i = 0; i < lay.L; ++i) {
    const int Wi = lay.width[i], nb = lay.widthp[i] / 4;
    const int nlo = n0 >> i, nhi = min((n0 + UT - 1) >> i, Wi - 1);   // pooled columns of this tile
    const float* lvl = gvol + lay.offset[i] + (bh * nb) * lay.pitch * 4;
    if (w1 < W1) {
      // this pixel's groups of 4 pooled columns that touch the tile, one 16-byte block each
      for (int g = (nlo >> 2) + rg; g <= (nhi >> 2); g += 4) {
        const int sb = rc_block_row<4>(g, w1 >> i, nb);
        const float4 v4 = *reinterpret_cast<const float4*>(lvl + ((int64_t)sb * lay.pitch + w1) * 4);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int w2i = 4 * g + r;
          if (w2i >= nlo && w2i <= nhi) {
            const int base = (w2i << i) - n0;
            for (int e = 0; e < (1 << i); ++e) tile[col][base + e] += v[r] * wgt;
          }
        }
      }
    }
    wgt *= 0.5f;
    __syncthreads();
  }

  // dC[bh][w1][w2] (row = w1) and dCt[bh][w2][w1] (row = w2), zero padded up to W2p / W1p
  for (int e = threadIdx.x; e < UT * UT; e += 256) {
    const int r = e >> 6, cc = e & 63;
    {
      const int a = c0 + r, bq = n0 + cc;  // tile[r][cc]
      if (a < W1 && bq < W2p) dC[(bh * W1 + a) * W2p + bq] = rc_from_float<T>(tile[r][cc]);
    }
    {
      const int bq = n0 + r, a = c0 + cc;  // tile[cc][r]
      if (bq < W2 && a < W1p) dCt[(bh * W2 + bq) * W1p + a] = rc_from_float<T>(tile[cc][r]);
    }
  }
}

template <typename T> __device__ __forceinline__ f4v mfma16(s8v a, s8v b, f4v c);
template <> __device__ __forceinline__ f4v mfma16<rc_half>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f4v mfma16<rc_bf16>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

// C[m][n] = sum_k A[m][k] * Bt[n][k] per (b,h); 64x64 tile, 4 waves of 32x32.
//   A  : fmap  [B, M=D, H, K]  ->  A[(b*M*H + h)*K + m*H*K + k]      (K-contiguous rows, unpadded)
//   Bt : dC    [B*H, N, Kp]     (rows zero padded to Kp, always 16-byte aligned)
//   C  : dfmap [B, M=D, H, N]
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ A, const T* __restrict__ Bt,
                                                      T* __restrict__ C, int M, int N, int K, int Kp, int H,
                                                      bool vecA) {
  constexpr bool IS16 = !std::is_same<T, float>::value;
  constexpr int BK = IS16 ? 32 : 16;
  constexpr int EPS = 16 / (int)sizeof(T);
  constexpr int PITCH = BK + EPS;  // elements; rows stay 16-byte aligned
  __shared__ __attribute__((aligned(16))) T As[64 * PITCH];
  __shared__ __attribute__((aligned(16))) T Bs[64 * PITCH];

  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  const int bh = blockIdx.z, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave & 1, wn = wave >> 1;
  const int lrow = threadIdx.x >> 2, lseg = threadIdx.x & 3;

  const T* Arow = A + ((int64_t)b * M * H + h) * K + (int64_t)(m0 + lrow) * H * K;
  const T* Brow = Bt + ((int64_t)bh * N + (n0 + lrow)) * Kp;
  const bool a_ok = (m0 + lrow) < M, b_ok = (n0 + lrow) < N;

  auto load_a = [&](int k0) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    const int k = k0 + lseg * EPS;
    if (a_ok && k < K) {
      if (vecA) {
        v = *reinterpret_cast<const uint4*>(Arow + k);
      } else {
        alignas(16) T tmp[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) tmp[e] = (k + e < K) ? Arow[k + e] : (T)0.0f;
        v = *reinterpret_cast<uint4*>(tmp);
      }
    }
    return v;
  };
  auto load_b = [&](int k0) -> uint4 {
    const int k = k0 + lseg * EPS;
    return (b_ok && k < Kp) ? *reinterpret_cast<const uint4*>(Brow + k) : make_uint4(0, 0, 0, 0);
  };

  f4v acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = (f4v)(0.f);

  uint4 ra = load_a(0), rb = load_b(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    *reinterpret_cast<uint4*>(As + lrow * PITCH + lseg * EPS) = ra;
    *reinterpret_cast<uint4*>(Bs + lrow * PITCH + lseg * EPS) = rb;
    __syncthreads();
    if (k0 + BK < K) {
      ra = load_a(k0 + BK);
      rb = load_b(k0 + BK);
    }
    if constexpr (IS16) {
      s8v af[2], bf[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        af[x] = *reinterpret_cast<const s8v*>(As + (wm * 32 + x * 16 + li) * PITCH + 8 * g);
        bf[x] = *reinterpret_cast<const s8v*>(Bs + (wn * 32 + x * 16 + li) * PITCH + 8 * g);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mfma16<T>(af[x], bf[y], acc[x][y]);
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float af[2], bf[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          af[x] = (float)As[(wm * 32 + x * 16 + li) * PITCH + kk * 4 + g];
          bf[x] = (float)Bs[(wn * 32 + x * 16 + li) * PITCH + kk * 4 + g];
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[x], bf[y], acc[x][y], 0, 0, 0);
      }
    }
  }

  T* Cb = C + ((int64_t)b * M * H + h) * N;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + x * 16 + 4 * g + r, n = n0 + wn * 32 + y * 16 + li;
        if (m < M && n < N) Cb[(int64_t)m * H * N + n] = rc_from_float<T>(acc[x][y][r]);
      }
}

template <typename T>
int run_bwd(const float* gvol, const rc_layout* lay, const void* f1, const void* f2, void* df1, void* df2,
            int D, void* work, hipStream_t st) {
  const rc_dev_layout d = rc_to_dev(lay);
  const int W1 = d.W1, W2 = d.W2, H = d.H;
  const int W1p = (int)round_up(W1, 32), W2p = (int)round_up(W2, 32);
  const int64_t BH = lay->B * lay->H;
  T* dC = reinterpret_cast<T*>(work);
  T* dCt = dC + round_up(BH * W1 * W2p, 64);
  const float scale = (float)(1.0 / sqrt((double)D));
  RC_REQUIRE(BH < 65536, RC_EINVAL, "B*H=%lld exceeds the grid z range; split the batch", (long long)BH);

  hipLaunchKernelGGL((unskew_fold_kernel<T>), dim3((W1 + UT - 1) / UT, (W2 + UT - 1) / UT, (unsigned)BH),
                     dim3(256), 0, st, gvol, d, dC, dCt, W1p, W2p, scale);
  int rc = rc_check_launch("rc_build_bwd(unskew)");
  if (rc) return rc;

  constexpr int EPS = 16 / (int)sizeof(T);
  const bool v2 = (W2 % EPS == 0) && ((uintptr_t)f2 % 16 == 0);
  const bool v1 = (W1 % EPS == 0) && ((uintptr_t)f1 % 16 == 0);
  // df1 = f2 * dC^T : M = D, N = W1, K = W2
  hipLaunchKernelGGL((gemm_nt_kernel<T>), dim3((W1 + 63) / 64, (D + 63) / 64, (unsigned)BH), dim3(256), 0, st,
                     (const T*)f2, (const T*)dC, (T*)df1, D, W1, W2, W2p, H, v2);
  rc = rc_check_launch("rc_build_bwd(df1)");
  if (rc) return rc;
  // df2 = f1 * dCt^T : M = D, N = W2, K = W1
  hipLaunchKernelGGL((gemm_nt_kernel<T>), dim3((W2 + 63) / 64, (D + 63) / 64, (unsigned)BH), dim3(256), 0, st,
                     (const T*)f1, (const T*)dCt, (T*)df2, D, W2, W1, W1p, H, v1);
  return rc_check_launch("rc_build_bwd(df2)");
}

int64_t elem_size(int dtype) { return dtype == RC_F32 ? 4 : 2; }

}  // namespace

extern "C" int64_t rc_build_bwd_work_bytes(const rc_layout* lay, int in_dtype) {
  if (!lay || rc_validate_layout(lay)) return -1;
  if (in_dtype != RC_F32 && in_dtype != RC_F16 && in_dtype != RC_BF16) return -1;
  const int64_t BH = lay->B * lay->H;
  const int64_t W1p = round_up(lay->W1, 32), W2p = round_up(lay->W2, 32);
  return (round_up(BH * lay->W1 * W2p, 64) + round_up(BH * lay->W2 * W1p, 64)) * elem_size(in_dtype);
}

extern "C" int rc_build_bwd(const float* grad_vol, const rc_layout* lay, const void* f1, const void* f2,
                            void* df1, void* df2, int64_t D, int in_dtype, void* work, void* stream) {
  RC_REQUIRE(grad_vol && lay && f1 && f2 && df1 && df2 && work, RC_EINVAL, "null pointer argument");
  int rc = rc_validate_layout(lay);
  if (rc) return rc;
  RC_REQUIRE(lay->dtype == RC_F32, RC_EINVAL, "the gradient pyramid must use an RC_F32 layout");
  RC_REQUIRE(D >= 1 && D < (1 << 20), RC_EINVAL, "bad feature depth D=%lld", (long long)D);
  RC_REQUIRE((uintptr_t)work % 16 == 0, RC_EINVAL, "work buffer must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (in_dtype) {
    case RC_F32: return run_bwd<float>(grad_vol, lay, f1, f2, df1, df2, (int)D, work, st);
    case RC_F16: return run_bwd<rc_half>(grad_vol, lay, f1, f2, df1, df2, (int)D, work, st);
    case RC_BF16: return run_bwd<rc_bf16>(grad_vol, lay, f1, f2, df1, df2, (int)D, work, st);
  }
  rc_set_error("unknown in_dtype %d", in_dtype);
  return RC_EUNSUPPORTED;
}

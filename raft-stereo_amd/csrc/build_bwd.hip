// Adjoint of the pyramid build: grad pyramid (float, skewed) -> d fmap1, d fmap2.
//
// Autograd of core/corr.py:119-125 (avg_pool2d chain) and :154-156 (einsum, / sqrt(D)).
// Two steps on one stream:
//   1. unskew_fold: reads every 16-byte block once (coalesced), folds levels 1..L-1 into level 0
//      (each pooled cell spreads 2^-i of its gradient over its 2^i children), scales by
//      1/sqrt(D), and writes the dense dC[b,h,w1,w2] in the compute type, rows zero padded to a
//      multiple of 32.
//   2. two MFMA contractions per image row, feature rows straight from global memory, the dC slab
//      staged through LDS:
//        df1[b,:,h,:] = f2[b,:,h,:] (D x W2)  *  dC[b,h]^T (W2 x W1)     (plain LDS image)
//        df2[b,:,h,:] = f1[b,:,h,:] (D x W1)  *  dC[b,h]   (W1 x W2)     (transposing LDS reads)
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
                                                          T* __restrict__ dC, int W2p, float scale) {
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

  // dC[bh][w1][w2], zero padded up to W2p (cells outside the image were never touched: still zero)
  for (int e = threadIdx.x; e < UT * UT; e += 256) {
    const int r = e >> 6, cc = e & 63;
    const int a = c0 + r, bq = n0 + cc;
    if (a < W1 && bq < W2p) dC[(bh * W1 + a) * W2p + bq] = rc_from_float<T>(tile[r][cc]);
  }
}

template <typename T> __device__ __forceinline__ f4v mfma16(s8v a, s8v b, f4v c);
template <> __device__ __forceinline__ f4v mfma16<rc_half>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f4v mfma16<rc_bf16>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4v_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef short s4v __attribute__((ext_vector_type(4)));

constexpr int GT = 512;  // threads: 8 waves x 32 feature rows
constexpr int TS = 64;   // output columns per workgroup

template <typename T> struct gemm_cfg {
  static constexpr bool IS16 = !std::is_same<T, float>::value;
  static constexpr int EPS = 16 / (int)sizeof(T);
  static constexpr int KC = IS16 ? 256 : 64;      // contraction chunk held in LDS
  static constexpr int KSTEP = IS16 ? 32 : 16;    // contraction elements per operand load
  static constexpr int PN = KC + EPS;             // pitch of the [n][k] image (rows stay 16-byte aligned,
                                                  // and 16 consecutive rows fall on different bank quads)
  static constexpr int PT = TS + 16;              // pitch of the [k][n] image (as the build's staged tile)
  static constexpr int LDS_ELEMS = (TS * PN > KC * PT) ? TS * PN : KC * PT;
};

// 16 bytes (one operand piece of 8 / 4 consecutive k) of a feature row, zero past the end of the row
template <typename T>
__device__ __forceinline__ u4v load_feat(const T* __restrict__ row, int k, int K, bool vec) {
  constexpr int EPS = 16 / (int)sizeof(T);
  if (vec && k + EPS <= K) return *reinterpret_cast<const u4v_a4*>(row + k);
  alignas(16) T tmp[EPS];
#pragma unroll
  for (int e = 0; e < EPS; ++e) tmp[e] = (k + e < K) ? row[k + e] : (T)0.0f;
  return *reinterpret_cast<u4v*>(tmp);
}

// 8 bytes (4 consecutive k of a 16-bit feature row), zero past the end of the row
template <typename T>
__device__ __forceinline__ uint2 load_feat4(const T* __restrict__ row, int k, int K, bool vec) {
  typedef uint32_t u2v_a4 __attribute__((ext_vector_type(2), aligned(4)));
  if (vec && k + 4 <= K) {
    const u2v_a4 v = *reinterpret_cast<const u2v_a4*>(row + k);
    return make_uint2(v[0], v[1]);
  }
  alignas(8) T tmp[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) tmp[e] = (k + e < K) ? row[k + e] : (T)0.0f;
  return *reinterpret_cast<uint2*>(tmp);
}

// Out[b, m, h, n] = sum_k Feat[b, m, h, k] * G(k, n) for one image row (b,h) and 64 columns n per
// workgroup; M feature channels, 32 per wave.
//   TRANS = false (d fmap1):  n = w1, k = w2, G(k,n) = dC[bh][n][k]  -> LDS image [n][k], plain 16-byte reads
//   TRANS = true  (d fmap2):  n = w2, k = w1, G(k,n) = dC[bh][k][n]  -> LDS image [k][n], transposing reads
// dC rows are ldc (a multiple of 32) long and zero beyond W2.  The MFMA rows are n and its columns are
// the feature channels, so a lane ends up with 4 consecutive n of one channel: one 8-byte store.
// Feature rows go from global memory straight into MFMA operands (each element is used once per
// workgroup); the 64 x K slab of dC is staged through LDS in chunks of KC.
template <typename T, bool TRANS>
__global__ __launch_bounds__(GT) void gemm_rows_kernel(const T* __restrict__ feat, const T* __restrict__ dC,
                                                       T* __restrict__ out, int M, int N, int K, int H, int W1,
                                                       int ldc, bool vec) {
  using C = gemm_cfg<T>;
  __shared__ __attribute__((aligned(16))) T lds[C::LDS_ELEMS];
  const int n0 = blockIdx.x * TS;
  const int bh = blockIdx.z, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.y * 256 + wave * 32;

  // this lane's two feature rows (clamped: rows past M are computed and dropped)
  const T* frow[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int m = min(m0 + 16 * x + li, M - 1);
    frow[x] = feat + (((int64_t)b * M + m) * H + h) * K;
  }
  const T* dCrow = dC + (int64_t)bh * W1 * ldc;

  f4v acc[4][2];
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    acc[y][0] = (f4v)(0.f);
    acc[y][1] = (f4v)(0.f);
  }

  constexpr int NKS = C::KC / C::KSTEP;  // operand steps per chunk
  for (int kc0 = 0; kc0 < K; kc0 += C::KC) {
    // ---- this chunk's feature pieces: all loads issued up front, in flight while the slab is staged.
    // 16-bit, plain image: lane (g, li) holds k = kk + 8 g .. + 7 of feature row li and of slab column
    // li; transposed image: the transposing read hands group g the k = kk + 4 g + {0..3} and
    // kk + 16 + 4 g + {0..3}, so the feature piece is gathered to match.  f32: 16 k per step as 4
    // exact-f32 MFMAs, lane (g, li) supplies k = kk + 4 g + e in sub-step e.
    u4v fa[NKS][2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int k = kc0 + ks * C::KSTEP;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (k >= K) {  // wave-uniform
          fa[ks][x] = (u4v)(0u);
        } else if constexpr (!C::IS16) {
          fa[ks][x] = load_feat<T>(frow[x], k + 4 * g, K, vec);
        } else if constexpr (!TRANS) {
          fa[ks][x] = load_feat<T>(frow[x], k + 8 * g, K, vec);
        } else {
          const uint2 lo = load_feat4<T>(frow[x], k + 4 * g, K, vec);
          const uint2 hi = load_feat4<T>(frow[x], k + 16 + 4 * g, K, vec);
          fa[ks][x] = (u4v){lo.x, lo.y, hi.x, hi.y};
        }
      }
    }

    __syncthreads();
    // ---- stage the slab chunk, zero filled outside dC
    if constexpr (!TRANS) {
      // image [n][k]: row n0 + r of dC, columns kc0 .. kc0 + KC
      constexpr int SEGS = C::KC / C::EPS;
      for (int e = threadIdx.x; e < TS * SEGS; e += GT) {
        const int r = e / SEGS, sgm = e - r * SEGS;
        const int n = n0 + r, k = kc0 + sgm * C::EPS;
        u4v v = (u4v)(0u);
        if (n < N && k < ldc) v = *reinterpret_cast<const u4v*>(dCrow + (int64_t)n * ldc + k);
        *reinterpret_cast<u4v*>(lds + r * C::PN + sgm * C::EPS) = v;
      }
    } else {
      // image [k][n]: row kc0 + r of dC (a w1), columns n0 .. n0 + 64
      constexpr int SEGS = TS / C::EPS;
      for (int e = threadIdx.x; e < C::KC * SEGS; e += GT) {
        const int r = e / SEGS, sgm = e - r * SEGS;
        const int k = kc0 + r, n = n0 + sgm * C::EPS;
        u4v v = (u4v)(0u);
        if (k < K && n < ldc) v = *reinterpret_cast<const u4v*>(dCrow + (int64_t)k * ldc + n);
        *reinterpret_cast<u4v*>(lds + r * C::PT + sgm * C::EPS) = v;
      }
    }
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int kk = ks * C::KSTEP;  // offset inside the chunk
      if (kc0 + kk >= K) break;      // wave-uniform
      if constexpr (C::IS16) {
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          s8v gb;
          if constexpr (!TRANS) {
            gb = *reinterpret_cast<const s8v*>(lds + (16 * y + li) * C::PN + kk + 8 * g);
          } else {
            typedef s4v __attribute__((address_space(3))) * lds_s4p;
            const short* base = reinterpret_cast<const short*>(lds);
            const int e = (kk + 4 * g + (li >> 2)) * C::PT + 16 * y + 4 * (li & 3);
            const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(base + e));
            const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(base + e + 16 * C::PT));
            gb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int x = 0; x < 2; ++x) acc[y][x] = mfma16<T>(gb, __builtin_bit_cast(s8v, fa[ks][x]), acc[y][x]);
        }
      } else {
        const float* lf = reinterpret_cast<const float*>(lds);
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          f4v gb;
          if constexpr (!TRANS) {
            gb = *reinterpret_cast<const f4v*>(lf + (16 * y + li) * C::PN + kk + 4 * g);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) gb[e] = lf[(kk + 4 * g + e) * C::PT + 16 * y + li];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int x = 0; x < 2; ++x)
              acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(gb[e], __builtin_bit_cast(f4v, fa[ks][x])[e],
                                                               acc[y][x], 0, 0, 0);
        }
      }
    }
  }

  // lane (g, li): columns n0 + 16 y + 4 g + {0..3} of channel m0 + 16 x + li
  const bool vec_out = (N % 4 == 0);
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int m = m0 + 16 * x + li;
    if (m >= M) continue;
    T* orow = out + (((int64_t)b * M + m) * H + h) * N;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int n = n0 + 16 * y + 4 * g;
      if (vec_out && n + 3 < N) {
        if constexpr (C::IS16) {
          typedef T t4v __attribute__((ext_vector_type(4)));
          *reinterpret_cast<t4v*>(orow + n) = __builtin_convertvector(acc[y][x], t4v);
        } else {
          *reinterpret_cast<f4v*>(orow + n) = acc[y][x];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) orow[n + r] = rc_from_float<T>(acc[y][x][r]);
      }
    }
  }
}

template <typename T>
int run_bwd(const float* gvol, const rc_layout* lay, const void* f1, const void* f2, void* df1, void* df2,
            int D, void* work, hipStream_t st) {
  const rc_dev_layout d = rc_to_dev(lay);
  const int W1 = d.W1, W2 = d.W2, H = d.H;
  const int W2p = (int)round_up(W2, 32);
  const int64_t BH = lay->B * lay->H;
  T* dC = reinterpret_cast<T*>(work);
  const float scale = (float)(1.0 / sqrt((double)D));
  RC_REQUIRE(BH < 65536, RC_EINVAL, "B*H=%lld exceeds the grid z range; split the batch", (long long)BH);

  hipLaunchKernelGGL((unskew_fold_kernel<T>), dim3((W1 + UT - 1) / UT, (W2p + UT - 1) / UT, (unsigned)BH),
                     dim3(256), 0, st, gvol, d, dC, W2p, scale);
  int rc = rc_check_launch("rc_build_bwd(unskew)");
  if (rc) return rc;

  const unsigned my = (unsigned)((D + 255) / 256);
  const bool v2 = (W2 * sizeof(T)) % 4 == 0 && (uintptr_t)f2 % 4 == 0;
  const bool v1 = (W1 * sizeof(T)) % 4 == 0 && (uintptr_t)f1 % 4 == 0;
  // d fmap1 = f2 . dC^T : n = w1, k = w2
  hipLaunchKernelGGL((gemm_rows_kernel<T, false>), dim3((W1 + TS - 1) / TS, my, (unsigned)BH), dim3(GT), 0, st,
                     (const T*)f2, (const T*)dC, (T*)df1, D, W1, W2, H, W1, W2p, v2);
  rc = rc_check_launch("rc_build_bwd(df1)");
  if (rc) return rc;
  // d fmap2 = f1 . dC : n = w2, k = w1
  hipLaunchKernelGGL((gemm_rows_kernel<T, true>), dim3((W2 + TS - 1) / TS, my, (unsigned)BH), dim3(GT), 0, st,
                     (const T*)f1, (const T*)dC, (T*)df2, D, W2, W1, H, W1, W2p, v1);
  return rc_check_launch("rc_build_bwd(df2)");
}

int64_t elem_size(int dtype) { return dtype == RC_F32 ? 4 : 2; }

}  // namespace

extern "C" int64_t rc_build_bwd_work_bytes(const rc_layout* lay, int in_dtype) {
  if (!lay || rc_validate_layout(lay)) return -1;
  if (in_dtype != RC_F32 && in_dtype != RC_F16 && in_dtype != RC_BF16) return -1;
  const int64_t BH = lay->B * lay->H;
  const int64_t W2p = round_up(lay->W2, 32);
  return round_up(BH * lay->W1 * W2p, 64) * elem_size(in_dtype);
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

// Correlation-volume build with the pooled pyramid fused into the epilogue.
//
// Replaces core/corr.py:111-125,148-156 (CorrBlock1D) and :32-42,53-61 (CorrBlockFast1D) of the
// reference: einsum('aijk,aijh->ajkh') / sqrt(D), reshape, and the avg_pool2d([1,2]) chain.
//
// One workgroup (4 waves) owns one image row (b,h) and a tile of 64 left-image pixels (w1) and
// sweeps the right-image positions (w2) in chunks of TN.  Per chunk: the f2 chunk [K][TN] is
// staged global -> registers -> LDS (the next chunk's loads are in flight during the MFMAs),
// transposed on the way out of LDS by ds_read_b64_tr_b16 (16-bit) so that both MFMA operands
// are K-major; the f1 tile stays in registers for the whole sweep.  MFMA rows are w2 and MFMA
// columns are w1, so a lane ends up with 4 consecutive w2 of one pixel: levels 1 and 2 of the
// pyramid are in-register means, level 3 needs one cross-lane exchange.  All levels are
// scattered into LDS rings indexed by the unwrapped skew row u = w2_i - (w1 >> i) - 17, stored
// like the pyramid itself ([row block][pixel][rows of the block]); once the rows of a block are
// complete for all 64 pixels of the tile the block is flushed with 16-byte stores, 1 KiB
// contiguous per wave.
#include "rc_common.h"

#include <math.h>
#include <type_traits>

namespace {

constexpr int BT = 256;    // threads per workgroup
constexpr int TM = 64;     // w1 tile
constexpr int KMAX = 256;  // feature depth held on chip

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ f4v mfma16(s8v a, s8v b, f4v c);
template <> __device__ __forceinline__ f4v mfma16<rc_half>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f4v mfma16<rc_bf16>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

template <typename T, typename VT, int TN> struct build_cfg {
  static constexpr bool IS16 = !std::is_same<T, float>::value;
  static constexpr int ESZ = sizeof(T);
  static constexpr int EPS = 16 / ESZ;            // elements per 16-byte segment
  static_assert(TN * ESZ == 128, "a staged row is 128 bytes");
  static constexpr int PF = TN + (IS16 ? 16 : 16);  // LDS row pitch (elements) of the staged tile
  static constexpr int KSTEP = IS16 ? 32 : 4;
  static constexpr int NKS = KMAX / KSTEP;        // k-steps: 8 (16-bit) or 64 (f32)
  static constexpr int AT = TN / 32;              // 16-row (w2) MFMA tiles per wave
  static constexpr int RB = 16 / (int)sizeof(VT);  // skew rows per 16-byte block of the pyramid
  static constexpr int RING_ROWS = 128 + 64 + 32 + 16;
  static constexpr int LDSF_BYTES = KMAX * PF * ESZ;
  static constexpr int RING_BYTES = RING_ROWS * TM * (int)sizeof(VT);
  static constexpr int LDS_BYTES = LDSF_BYTES + RING_BYTES;
};

__device__ __forceinline__ int ring_row0(int level) {  // first row of ring `level` in the ring area
  return level == 0 ? 0 : level == 1 ? 128 : level == 2 ? 192 : 224;
}

// ---- staging: one [K][128-byte] tile, global -> registers (8 x 16 B per thread) -> LDS
template <typename T>
__device__ __forceinline__ void stage_load(uint4 (&regs)[8], const T* __restrict__ rowbase, int64_t kstride,
                                           int W, int n0, int D, bool vec) {
  constexpr int EPS = 16 / sizeof(T);
  const int seg = threadIdx.x & 7, kr = threadIdx.x >> 3;
  const int col = n0 + seg * EPS;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int k = p * 32 + kr;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < D) {
      const T* src = rowbase + (int64_t)k * kstride + col;
      if (vec) {
        if (col < W) v = *reinterpret_cast<const uint4*>(src);
      } else {
        alignas(16) T tmp[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) tmp[e] = (col + e < W) ? src[e] : (T)0.0f;
        v = *reinterpret_cast<uint4*>(tmp);
      }
    }
    regs[p] = v;
  }
}

template <typename T, int PF>
__device__ __forceinline__ void stage_store(const uint4 (&regs)[8], T* ldsF) {
  constexpr int EPS = 16 / sizeof(T);
  const int seg = threadIdx.x & 7, kr = threadIdx.x >> 3;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int k = p * 32 + kr;
    *reinterpret_cast<uint4*>(ldsF + k * PF + seg * EPS) = regs[p];
  }
}

// ---- 16-bit MFMA operand out of the [K][n] LDS image (K-major fragments by transposed read).
// Lane l = 16 g + i supplies row (i >> 2) / columns 4 (i & 3).. of its group's 4-row block and
// receives column i.  Group g takes k = 32 kb + 4 g + {0..3} and 32 kb + 16 + 4 g + {0..3}; the
// same assignment is used for both operands, so the contraction is unchanged while the two
// groups of a half-wave read 8 consecutive rows (conflict-free at a 160-byte pitch).
template <int PF>
__device__ __forceinline__ s8v tr_frag(const short* ldsF, int kb, int col_off) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i = lane & 15;
  const int e = (kb * 32 + 4 * g + (i >> 2)) * PF + col_off + 4 * (i & 3);
  typedef s4v __attribute__((address_space(3))) * lds_s4p;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(ldsF + e));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(ldsF + e + 16 * PF));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename T, typename VT, int TN>
__global__ __launch_bounds__(BT) void build_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                   VT* __restrict__ vol, rc_dev_layout lay, int D, float scale,
                                                   int n_tiles, int n_rows, bool vec1, bool vec2) {
  using C = build_cfg<T, VT, TN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ldsF = reinterpret_cast<T*>(smem);
  VT* ring = reinterpret_cast<VT*>(smem + C::LDSF_BYTES);

  // XCD-aware mapping: the tiles of one image row get block ids congruent mod 8 (one XCD, one
  // L2), since they all re-read that row's f2.
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int row = (j / n_tiles) * 8 + (id & 7);
  const int tile = j % n_tiles;
  if (row >= n_rows) return;
  const int b = row / lay.H, h = row - b * lay.H;
  const int c0 = tile * TM;
  const int W1 = lay.W1, W2 = lay.W2, L = lay.L;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave & 1, wn = wave >> 1;  // wave's w1 half / w2 half
  const int nk = (D + C::KSTEP - 1) / C::KSTEP;

  const T* f1row = f1 + ((int64_t)b * D * lay.H + h) * W1;
  const T* f2row = f2 + ((int64_t)b * D * lay.H + h) * W2;
  const int64_t ks1 = (int64_t)lay.H * W1, ks2 = (int64_t)lay.H * W2;

  uint4 regs[8];

  // ---- f1 tile -> per-wave register-resident B operand
  s8v bfrag16[C::IS16 ? C::NKS : 1][2];
  float bfrag32[C::IS16 ? 1 : C::NKS][2];
  if constexpr (C::IS16) {
    stage_load<T>(regs, f1row, ks1, W1, c0, D, vec1);
    stage_store<T, C::PF>(regs, ldsF);
    stage_load<T>(regs, f2row, ks2, W2, 0, D, vec2);  // chunk 0 of f2, in flight
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < C::NKS; ++kb) {
      if (kb < nk) {
        bfrag16[kb][0] = tr_frag<C::PF>(reinterpret_cast<const short*>(ldsF), kb, wm * 32);
        bfrag16[kb][1] = tr_frag<C::PF>(reinterpret_cast<const short*>(ldsF), kb, wm * 32 + 16);
      } else {
        bfrag16[kb][0] = (s8v)(0);
        bfrag16[kb][1] = (s8v)(0);
      }
    }
  } else {
    stage_load<T>(regs, f2row, ks2, W2, 0, D, vec2);
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        const int k = ks * 4 + g, w1 = c0 + wm * 32 + bt * 16 + li;
        bfrag32[ks][bt] = (k < D && w1 < W1) ? (float)f1row[(int64_t)k * ks1 + w1] : 0.f;
      }
    }
  }

  const int n_chunks = (W2 + TN - 1) / TN;
  const int64_t rowblk = (int64_t)row;

  // ring cell of unwrapped skew row u (level i), tile column w1l
  auto ring_at = [&](int i, int u, int w1l) -> VT* {
    const int rs = u & ((128 >> i) - 1);
    return ring + ((ring_row0(i) + (rs & ~(C::RB - 1))) * TM + w1l * C::RB + (rs & (C::RB - 1)));
  };

  // flush `nblk` row blocks of level i starting at skew row ulo (a multiple of RB) to global memory
  auto flush = [&](int i, int ulo, int nblk) {
    const int Wi = lay.width[i], WiP = lay.widthp[i];
    VT* lvl = vol + lay.offset[i] + (rowblk * (WiP / C::RB)) * lay.pitch * C::RB;
    for (int it = threadIdx.x; it < nblk * TM; it += BT) {
      const int blk = it >> 6, w1l = it & 63;
      const int u0 = ulo + blk * C::RB;
      const int w20 = u0 + ((c0 + w1l) >> i) + RC_SKEW;  // w2 of the block's first row for this pixel
      if (w20 + C::RB - 1 < 0 || w20 >= Wi) continue;     // no cell of this block exists
      const VT* src = ring_at(i, u0, w1l);
      VT* dst = lvl + ((int64_t)(rc_pmod(u0, WiP) / C::RB) * lay.pitch + c0 + w1l) * C::RB;
      if (w20 >= 0 && w20 + C::RB - 1 < Wi) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < C::RB; ++e)
          if ((unsigned)(w20 + e) < (unsigned)Wi) dst[e] = src[e];
      }
    }
  };

  for (int c = 0; c < n_chunks; ++c) {
    const int n0 = c * TN;
    __syncthreads();  // previous readers of ldsF and of the rings are done
    stage_store<T, C::PF>(regs, ldsF);
    __syncthreads();
    if (c + 1 < n_chunks) stage_load<T>(regs, f2row, ks2, W2, n0 + TN, D, vec2);

    f4v acc[C::AT][2];
#pragma unroll
    for (int at = 0; at < C::AT; ++at) {
      acc[at][0] = (f4v)(0.f);
      acc[at][1] = (f4v)(0.f);
    }
    if constexpr (C::IS16) {
#pragma unroll
      for (int kb = 0; kb < C::NKS; ++kb) {
        if (kb < nk) {
#pragma unroll
          for (int at = 0; at < C::AT; ++at) {
            const s8v a = tr_frag<C::PF>(reinterpret_cast<const short*>(ldsF), kb, wn * (TN / 2) + at * 16);
            acc[at][0] = mfma16<T>(a, bfrag16[kb][0], acc[at][0]);
            acc[at][1] = mfma16<T>(a, bfrag16[kb][1], acc[at][1]);
          }
        }
      }
    } else {
      const float* lf = reinterpret_cast<const float*>(ldsF);
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        if (ks < nk) {
#pragma unroll
          for (int at = 0; at < C::AT; ++at) {
            const float a = lf[(ks * 4 + g) * C::PF + wn * (TN / 2) + at * 16 + li];
            acc[at][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag32[ks][0], acc[at][0], 0, 0, 0);
            acc[at][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag32[ks][1], acc[at][1], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue: scale, round to the storage type, pool, scatter into the rings
#pragma unroll
    for (int at = 0; at < C::AT; ++at) {
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        const int w1l = wm * 32 + bt * 16 + li;      // column inside the tile
        const int w1 = c0 + w1l;
        const int w2b = n0 + wn * (TN / 2) + at * 16 + 4 * g;  // first of this lane's 4 w2
        float v0[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] = rc_round_as<VT>(acc[at][bt][r] * scale);
          *ring_at(0, w2b + r - w1 - RC_SKEW, w1l) = rc_from_float<VT>(v0[r]);
        }
        if (L > 1) {
          const float p10 = rc_round_as<VT>(0.5f * (v0[0] + v0[1]));
          const float p11 = rc_round_as<VT>(0.5f * (v0[2] + v0[3]));
          const int u1 = (w2b >> 1) - (w1 >> 1) - RC_SKEW;
          *ring_at(1, u1, w1l) = rc_from_float<VT>(p10);
          *ring_at(1, u1 + 1, w1l) = rc_from_float<VT>(p11);
          if (L > 2) {
            const float p2 = rc_round_as<VT>(0.5f * (p10 + p11));
            *ring_at(2, (w2b >> 2) - (w1 >> 2) - RC_SKEW, w1l) = rc_from_float<VT>(p2);
            if (L > 3) {
              const float other = __shfl_xor(p2, 16);
              const float p3 = rc_round_as<VT>(0.5f * (p2 + other));
              if ((g & 1) == 0) *ring_at(3, (w2b >> 3) - (w1 >> 3) - RC_SKEW, w1l) = rc_from_float<VT>(p3);
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- flush the row blocks completed by this chunk (they start on a block boundary)
    for (int i = 0; i < L; ++i) {
      const int Mi = TM >> i, TNi = TN >> i;
      flush(i, (n0 >> i) - (c0 >> i) - Mi - (RC_SKEW - 1), TNi / C::RB);
    }
  }
  // rows that never complete inside the tile (their other part belongs to the wrapped range)
  const int n0l = (n_chunks - 1) * TN;
  for (int i = 0; i < L; ++i) {
    const int Mi = TM >> i, TNi = TN >> i;
    flush(i, (n0l >> i) + TNi - (c0 >> i) - Mi - (RC_SKEW - 1), (Mi + C::RB - 1) / C::RB);
  }
}

// ------------------------------------------------------------------ straightforward fallback
// (any D, one thread per level-0 element; used for D > 256 and as an on-device cross-check)
template <typename T, typename VT>
__global__ void naive_corr_kernel(const T* __restrict__ f1, const T* __restrict__ f2, VT* __restrict__ vol,
                                  rc_dev_layout lay, int D, float scale, int64_t total) {
  constexpr int RB = 16 / (int)sizeof(VT);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int W1 = lay.W1, W2 = lay.W2;
  const int w1 = (int)(idx % W1);
  const int64_t t = idx / W1;
  const int w2 = (int)(t % W2);
  const int64_t bh = t / W2;
  const int b = (int)(bh / lay.H), h = (int)(bh % lay.H);
  const T* p1 = f1 + ((int64_t)b * D * lay.H + h) * W1 + w1;
  const T* p2 = f2 + ((int64_t)b * D * lay.H + h) * W2 + w2;
  float acc = 0.f;
  for (int k = 0; k < D; ++k)
    acc = fmaf(rc_to_float(p1[(int64_t)k * lay.H * W1]), rc_to_float(p2[(int64_t)k * lay.H * W2]), acc);
  vol[rc_cell<RB>(lay, 0, bh, rc_pmod(w2 - w1 - RC_SKEW, lay.widthp[0]), w1)] = rc_from_float<VT>(acc * scale);
}

template <typename VT>
__global__ void naive_pool_kernel(VT* __restrict__ vol, rc_dev_layout lay, int level, int64_t total) {
  constexpr int RB = 16 / (int)sizeof(VT);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int W1 = lay.W1, Wi = lay.width[level];
  const int w1 = (int)(idx % W1);
  const int64_t t = idx / W1;
  const int w2 = (int)(t % Wi);
  const int64_t bh = t / Wi;
  const int qp = w1 >> (level - 1), WpP = lay.widthp[level - 1];
  const float a = rc_to_float(vol[rc_cell<RB>(lay, level - 1, bh, rc_pmod(2 * w2 - qp - RC_SKEW, WpP), w1)]);
  const float bb = rc_to_float(vol[rc_cell<RB>(lay, level - 1, bh, rc_pmod(2 * w2 + 1 - qp - RC_SKEW, WpP), w1)]);
  vol[rc_cell<RB>(lay, level, bh, rc_pmod(w2 - (w1 >> level) - RC_SKEW, lay.widthp[level]), w1)] =
      rc_from_float<VT>(0.5f * (a + bb));
}

template <typename T, typename VT>
int launch_naive(const void* f1, const void* f2, void* vol, const rc_layout* lay, int D, float scale,
                 hipStream_t st) {
  const rc_dev_layout d = rc_to_dev(lay);
  int64_t total = lay->B * lay->H * lay->W2 * lay->W1;
  hipLaunchKernelGGL((naive_corr_kernel<T, VT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     (const T*)f1, (const T*)f2, (VT*)vol, d, D, scale, total);
  int rc = rc_check_launch("rc_build_pyramid(naive corr)");
  for (int i = 1; i < lay->L && !rc; ++i) {
    total = lay->B * lay->H * (int64_t)lay->width[i] * lay->W1;
    hipLaunchKernelGGL((naive_pool_kernel<VT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (VT*)vol, d, i, total);
    rc = rc_check_launch("rc_build_pyramid(naive pool)");
  }
  return rc;
}

template <typename T, typename VT, int TN>
int launch_build(const void* f1, const void* f2, void* vol, const rc_layout* lay, int D, float scale,
                 hipStream_t st) {
  using C = build_cfg<T, VT, TN>;
  static bool attr_set[64] = {};  // per device; raising the limit is idempotent, races are harmless
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    rc_set_error("no current HIP device");
    return RC_ENODEVICE;
  }
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build_kernel<T, VT, TN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      rc_set_error("hipFuncSetAttribute(%d bytes LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
      return (int)e;
    }
    attr_set[dev] = true;
  }
  const rc_dev_layout d = rc_to_dev(lay);
  const int n_tiles = (int)((lay->W1 + TM - 1) / TM);
  const int n_rows = (int)(lay->B * lay->H);
  const int64_t blocks = (int64_t)((n_rows + 7) / 8) * 8 * n_tiles;
  RC_REQUIRE(blocks < (1ll << 31), RC_EINVAL, "grid too large");
  constexpr int EPS = 16 / (int)sizeof(T);
  const bool vec1 = (lay->W1 % EPS == 0) && (((uintptr_t)f1) % 16 == 0);
  const bool vec2 = (lay->W2 % EPS == 0) && (((uintptr_t)f2) % 16 == 0);
  hipLaunchKernelGGL((build_kernel<T, VT, TN>), dim3((unsigned)blocks), dim3(BT), C::LDS_BYTES, st,
                     (const T*)f1, (const T*)f2, (VT*)vol, d, D, scale, n_tiles, n_rows, vec1, vec2);
  return rc_check_launch("rc_build_pyramid");
}

}  // namespace

static int build_dispatch(const void* f1, const void* f2, void* vol, const rc_layout* lay, int64_t D,
                          int in_dtype, void* stream, bool force_naive) {
  RC_REQUIRE(f1 && f2 && vol && lay, RC_EINVAL, "null pointer argument");
  int rc = rc_validate_layout(lay);
  if (rc) return rc;
  RC_REQUIRE(D >= 1 && D < (1 << 20), RC_EINVAL, "bad feature depth D=%lld", (long long)D);
  RC_REQUIRE((uintptr_t)vol % 16 == 0, RC_EINVAL, "pyramid pointer must be 16-byte aligned");
  const float scale = (float)(1.0 / sqrt((double)D));
  hipStream_t st = (hipStream_t)stream;
  const bool naive = force_naive || D > KMAX;
  const int iD = (int)D;
  const int vol_dtype = lay->dtype;
#define RC_BUILD_CASE(IN, VOL, T, VT, TN)                                            \
  if (in_dtype == IN && vol_dtype == VOL)                                            \
    return naive ? launch_naive<T, VT>(f1, f2, vol, lay, iD, scale, st)              \
                 : launch_build<T, VT, TN>(f1, f2, vol, lay, iD, scale, st);
  RC_BUILD_CASE(RC_F32, RC_F32, float, float, 32)
  RC_BUILD_CASE(RC_F16, RC_F16, rc_half, rc_half, 64)
  RC_BUILD_CASE(RC_BF16, RC_BF16, rc_bf16, rc_bf16, 64)
  RC_BUILD_CASE(RC_F16, RC_F32, rc_half, float, 64)
  RC_BUILD_CASE(RC_BF16, RC_F32, rc_bf16, float, 64)
#undef RC_BUILD_CASE
  rc_set_error("unsupported (in_dtype=%d, pyramid dtype=%d) combination", in_dtype, vol_dtype);
  return RC_EUNSUPPORTED;
}

extern "C" int rc_build_pyramid(const void* f1, const void* f2, void* vol, const rc_layout* lay, int64_t D,
                                int in_dtype, void* stream) {
  return build_dispatch(f1, f2, vol, lay, D, in_dtype, stream, false);
}

// Diagnostic twin of rc_build_pyramid that always takes the one-thread-per-element kernels.
extern "C" int rc_build_pyramid_naive(const void* f1, const void* f2, void* vol, const rc_layout* lay,
                                      int64_t D, int in_dtype, void* stream) {
  return build_dispatch(f1, f2, vol, lay, D, in_dtype, stream, true);
}

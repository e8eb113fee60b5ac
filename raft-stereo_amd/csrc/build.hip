// Correlation-volume build with the pooled pyramid fused into the epilogue.
//
// Replaces core/corr.py:111-125,148-156 (CorrBlock1D) and :32-42,53-61 (CorrBlockFast1D) of the
// reference: einsum('aijk,aijh->ajkh') / sqrt(D), reshape, and the avg_pool2d([1,2]) chain.
//
// One workgroup (4 waves) owns one image row (b,h) and a tile of 128 left-image pixels (w1), 32
// per wave, and sweeps the right-image positions (w2) in chunks of TN (64 for 16-bit inputs, 32
// for f32).  Per chunk the f2 chunk [K][TN] is staged global -> registers -> LDS (the next
// chunk's loads are in flight during the MFMAs) and transposed on the way out of LDS by
// ds_read_b64_tr_b16 (16-bit) so that both MFMA operands are K-major; the f1 tile stays in
// registers for the whole sweep.  MFMA rows are w2 and MFMA columns are w1, so a lane ends up
// with 4 consecutive, 4-aligned w2 of one pixel: half (16-bit) or all (f32) of one 16-byte block
// of the rc_layout.  Halves are paired, and the pooled levels are assembled, with
// v_permlane16_swap / v_permlane32_swap (no LDS), and every level is written straight from
// registers: each store instruction of a wave writes whole, aligned 128-byte lines.
#include "rc_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// Ablation switches for tools/lab_build.hip only (the library is built with 0): 1 = no pyramid stores,
// 2 = no MFMA, 4 = no f2 loads after the first two chunks
#ifndef RC_LAB_ABLATE
#define RC_LAB_ABLATE 0
#endif

// Phase stamps for tools/lab_build.hip only (the library is built without): s_memtime at the phase boundaries
// of every wave of build16_kernel, into a buffer of their own that nothing else reads
#ifdef RC_LAB_STAMPS
constexpr int RC_LAB_NSTAMP = 32;
__device__ unsigned long long rc_lab_stamps[4096 * 8 * RC_LAB_NSTAMP];
#define RC_STAMP(slot)                                                                                              \
  do {                                                                                                              \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && (slot) < RC_LAB_NSTAMP)                                     \
      rc_lab_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * RC_LAB_NSTAMP + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int rc_lab_read_stamps(unsigned long long* dst, size_t n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rc_lab_stamps), n * sizeof(unsigned long long));
}
#else
#define RC_STAMP(slot) do {} while (0)
#endif

namespace {

constexpr int BT = 256;    // threads per workgroup
constexpr int TW = 32;     // w1 columns per wave
constexpr int TM = 128;    // w1 tile of a workgroup
constexpr int KMAX = 256;  // feature depth held on chip

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ f4v mfma16(s8v a, s8v b, f4v c);
template <> __device__ __forceinline__ f4v mfma16<rc_half>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f4v mfma16<rc_bf16>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

template <typename T> struct build_cfg {
  static constexpr bool IS16 = !std::is_same<T, float>::value;
  static constexpr int ESZ = sizeof(T);
  static constexpr int EPS = 16 / ESZ;            // elements per 16-byte segment
  static constexpr int TN = 128 / ESZ;            // w2 per chunk: a staged row is 128 bytes
  static constexpr int PF = TN + 16;              // LDS row pitch (elements) of the staged tile
  static constexpr int KSTEP = IS16 ? 32 : 4;
  static constexpr int NKS = KMAX / KSTEP;        // k-steps: 8 (16-bit) or 64 (f32)
  static constexpr int NRT = TN / 16;             // 16-row (w2) MFMA tiles per chunk: 4 or 2
  static constexpr int LDS_BYTES = KMAX * PF * ESZ;
};

// ---- staging: one [K][128-byte] tile, global -> registers (8 x 16 B per thread) -> LDS.
// FAST > 0 (rows 4-byte aligned, D == KMAX): no zero fill and one scalar base, so the 8 loads issue
// back to back; the 16-byte loads only assume dword alignment.  A segment wholly past the end of
// the row re-reads the row start: what it produces only ever lands in padding cells of the pyramid
// (w2 >= W2) or in columns that are not stored (w1 >= W1).  The one segment that straddles the
// end of a row is read element by element.
typedef uint32_t u4v_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <typename T, int FAST, int NP = 8>
__device__ __forceinline__ void stage_load(u4v (&regs)[NP], const T* __restrict__ rowbase, int64_t kstride,
                                           int W, int n0, int D) {
  constexpr int EPS = 16 / sizeof(T);
  constexpr int RPP = KMAX / NP;  // rows per pass = threads / 8
  const int seg = threadIdx.x & 7, kr = threadIdx.x >> 3;
  const int col = n0 + seg * EPS;
  if constexpr (FAST) {
    if (FAST == 2 || col + EPS <= W || col >= W) {  // FAST == 2: W is a multiple of the segment
      const uint32_t voff = ((uint32_t)kr * (uint32_t)kstride + (uint32_t)(col >= W ? 0 : col)) * (uint32_t)sizeof(T);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const char* src = reinterpret_cast<const char*>(rowbase) + (int64_t)p * RPP * kstride * (int64_t)sizeof(T);
        regs[p] = *reinterpret_cast<const u4v_a4*>(src + voff);
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const T* src = rowbase + (int64_t)(p * RPP + kr) * kstride + col;
        alignas(16) T tmp[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) tmp[e] = (col + e < W) ? src[e] : (T)0.0f;
        regs[p] = *reinterpret_cast<u4v*>(tmp);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int k = p * RPP + kr;
      u4v v = (u4v)(0u);
      if (k < D) {
        const T* src = rowbase + (int64_t)k * kstride + col;
        alignas(16) T tmp[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) tmp[e] = (col + e < W) ? src[e] : (T)0.0f;
        v = *reinterpret_cast<u4v*>(tmp);
      }
      regs[p] = v;
    }
  }
}

template <typename T, int PF, int NP = 8>
__device__ __forceinline__ void stage_store(const u4v (&regs)[NP], T* ldsF) {
  constexpr int EPS = 16 / sizeof(T);
  const int seg = threadIdx.x & 7, kr = threadIdx.x >> 3;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int k = p * (KMAX / NP) + kr;
    *reinterpret_cast<u4v*>(ldsF + k * PF + seg * EPS) = regs[p];
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

// ---- cross-lane exchanges between the four 16-lane rows of a wave (gfx950)
// swap16: rows 1 and 3 of x trade places with rows 0 and 2 of y;  swap32: rows 2,3 of x with rows 0,1 of y
__device__ __forceinline__ void swap16(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}
__device__ __forceinline__ void swap32(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}
// 4x4 transpose over lane rows: afterwards lane row g holds in r[p] what lane row p held in r[g]
__device__ __forceinline__ void transpose_rows(uint32_t (&r)[4]) {
  swap16(r[0], r[1]);
  swap16(r[2], r[3]);
  swap32(r[0], r[2]);
  swap32(r[1], r[3]);
}

typedef float f2v __attribute__((ext_vector_type(2)));
// two floats -> one dword of two 16-bit values (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round to nearest even)
template <typename VT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef VT v2 __attribute__((ext_vector_type(2)));
  const f2v f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, v2));
}
template <typename VT> __device__ __forceinline__ float unpack_lo(uint32_t d) {
  return rc_to_float(__builtin_bit_cast(VT, (uint16_t)(d & 0xffffu)));
}
template <typename VT> __device__ __forceinline__ float unpack_hi(uint32_t d) {
  return rc_to_float(__builtin_bit_cast(VT, (uint16_t)(d >> 16)));
}
template <> __device__ __forceinline__ float unpack_lo<rc_bf16>(uint32_t d) { return __uint_as_float(d << 16); }
template <> __device__ __forceinline__ float unpack_hi<rc_bf16>(uint32_t d) { return __uint_as_float(d & 0xffff0000u); }

// Where a wave's blocks go.  Per level: a wave-uniform base (this image row's slab) plus a 32-bit
// lane offset (block row * pitch + pixel) * 16 bytes.
template <typename VT> struct out_map {
  static constexpr int RB = 16 / (int)sizeof(VT);
  static constexpr int LG = sizeof(VT) == 4 ? 2 : 3;
  char* base[RC_MAX_LEVELS];     // wave-uniform
  uint32_t nb[RC_MAX_LEVELS];    // block rows
  uint32_t kb[RC_MAX_LEVELS];    // bias - RC_WRAP_ROWS / RB
  bool slowmod[RC_MAX_LEVELS];   // level only a few pixels wide: general modulus
  uint32_t blk_bytes;            // pitch * 16
  int W1;

  // byte offset of the block of group G of pixel `col` at level i
  __device__ __forceinline__ uint32_t offset(int i, int col, int G) const {
    uint32_t t = (uint32_t)(G - ((col >> i) >> LG)) + kb[i];  // >= 0 by choice of bias
    if (!slowmod[i]) {  // wave-uniform
      t = min(t, t - 2 * nb[i]);
      t = min(t, t - nb[i]);
    } else {
      t %= nb[i];
    }
    return __umul24(t, blk_bytes) + (uint32_t)col * 16u;
  }
  __device__ __forceinline__ bool ok(int i, int col, int G) const { return col < W1 && (uint32_t)G < nb[i]; }
  __device__ __forceinline__ void store16(int i, int col, int G, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const {
    if constexpr (RC_LAB_ABLATE & 1) {
      asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(offset(i, col, G)));
      return;
    }
    if (ok(i, col, G)) {
      const u4v v = {a, b, c, d};
      __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(base[i] + offset(i, col, G)));
    }
  }
  __device__ __forceinline__ void store8(int i, int col, int G, int half, uint32_t a, uint32_t b) const {
    if (ok(i, col, G)) *reinterpret_cast<uint2*>(base[i] + (offset(i, col, G) + 8u * half)) = make_uint2(a, b);
  }
};

// ---- epilogue, 16-bit pyramid: a 64 (w2) x 32 (w1) wave tile, acc[rt][ct], lane (g, li) holds rows
// n0 + 16 rt + 4 g + {0..3} of column cw + 16 ct + li.  Levels >= 1 are means of the already
// rounded lower level (what avg_pool2d on a 16-bit volume does).
template <typename VT>
__device__ __forceinline__ void epilogue16(const f4v (&acc)[4][2], float scale, const out_map<VT>& om, int L,
                                           int n0, int cw) {
  const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int col_sw = cw + 16 * (g & 1) + li;  // column a lane serves after a swap16 between the column tiles
  uint32_t q1[4][2];  // level 1: two values per (row tile, column tile)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    uint32_t x[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      x[ct][0] = pack2<VT>(acc[rt][ct][0] * scale, acc[rt][ct][1] * scale);
      x[ct][1] = pack2<VT>(acc[rt][ct][2] * scale, acc[rt][ct][3] * scale);
      q1[rt][ct] = pack2<VT>(0.5f * (unpack_lo<VT>(x[ct][0]) + unpack_hi<VT>(x[ct][0])),
                             0.5f * (unpack_lo<VT>(x[ct][1]) + unpack_hi<VT>(x[ct][1])));
    }
    // lane rows 0,2 collect the 8-row blocks of column tile 0, lane rows 1,3 those of column tile 1
    swap16(x[0][0], x[1][0]);
    swap16(x[0][1], x[1][1]);
    om.store16(0, col_sw, ((n0 + 16 * rt) >> 3) + (g >> 1), x[0][0], x[0][1], x[1][0], x[1][1]);
  }
  if (L < 3) return;  // level 1 is not stored (the lookup derives it from level 0); level 2 is its pairwise mean
  float p2[4][2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) p2[rt][ct] = 0.5f * (unpack_lo<VT>(q1[rt][ct]) + unpack_hi<VT>(q1[rt][ct]));
  uint32_t w[4];  // w[ct + 2 m] = level-2 values of row tiles 2m, 2m+1: block elements g and 4 + g
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = pack2<VT>(p2[2 * (j >> 1)][j & 1], p2[2 * (j >> 1) + 1][j & 1]);
  transpose_rows(w);  // lane row j: w[p] = elements (p, 4 + p) of the block of (column tile j & 1, m = j >> 1)
  const uint32_t d0 = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u), d1 = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u);
  const uint32_t d2 = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u), d3 = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
  om.store16(2, col_sw, (n0 >> 5) + (g >> 1), d0, d1, d2, d3);  // level 3 is derived from this one
}

// ---- epilogue, f32 pyramid: a 32 (w2) x 32 (w1) wave tile (two row tiles), n0 a multiple of 32.
// A lane's 4 rows are one whole block of level 0.
__device__ __forceinline__ void epilogue32(const f4v& a00, const f4v& a01, const f4v& a10, const f4v& a11,
                                           float scale, const out_map<float>& om, int L, int n0, int cw) {
  const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int col_sw = cw + 16 * (g & 1) + li;
  const f4v* acc[2][2] = {{&a00, &a01}, {&a10, &a11}};
  float p1[2][2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (*acc[rt][ct])[r] * scale;
      om.store16(0, cw + 16 * ct + li, ((n0 + 16 * rt) >> 2) + g, __float_as_uint(v[0]), __float_as_uint(v[1]),
                 __float_as_uint(v[2]), __float_as_uint(v[3]));
      p1[rt][ct][0] = 0.5f * (v[0] + v[1]);
      p1[rt][ct][1] = 0.5f * (v[2] + v[3]);
    }
  if (L < 3) return;  // levels 1 and 3 are not stored
  float p2[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    p2[rt][0] = 0.5f * (p1[rt][0][0] + p1[rt][0][1]);
    p2[rt][1] = 0.5f * (p1[rt][1][0] + p1[rt][1][1]);
  }
  uint32_t r[4];  // r[ct + 2 rt]
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = __float_as_uint(p2[j >> 1][j & 1]);
  transpose_rows(r);  // lane row j: the level-2 block of (column tile j & 1, row tile j >> 1)
  om.store16(2, col_sw, (n0 >> 4) + (g >> 1), r[0], r[1], r[2], r[3]);
}

// ---- epilogue of a 64 (w2) x 16 (w1) wave tile of a 16-bit pyramid (build16_kernel): acc[rt], lane (g, li) holds
// rows n0 + 16 rt + 4 g + {0..3} of column cw + li: eight level-0 blocks and two level-2 blocks per column.
template <typename VT>
__device__ __forceinline__ void epilogue16_col(const f4v (&acc)[4], float scale, const out_map<VT>& om, int L, int n0,
                                               int cw) {
  const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int col = cw + li;
  float p2[4];  // level 2: one value per row tile (level 1, its two parents, is not stored)
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    uint32_t x[2][2];  // [row tile 2m + e][half]: 4 rows = half a level-0 block
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const f4v& a = acc[2 * m + e];
      x[e][0] = pack2<VT>(a[0] * scale, a[1] * scale);
      x[e][1] = pack2<VT>(a[2] * scale, a[3] * scale);
      const uint32_t q1 = pack2<VT>(0.5f * (unpack_lo<VT>(x[e][0]) + unpack_hi<VT>(x[e][0])),
                                    0.5f * (unpack_lo<VT>(x[e][1]) + unpack_hi<VT>(x[e][1])));
      p2[2 * m + e] = 0.5f * (unpack_lo<VT>(q1) + unpack_hi<VT>(q1));
    }
    // lane rows (0,1) and (2,3) pair their 4-row halves: afterwards lane row g holds the whole block
    // (row tile 2m + (g & 1), rows 8 (g >> 1) .. + 8) in x[0][0], x[0][1], x[1][0], x[1][1]
    swap16(x[0][0], x[1][0]);
    swap16(x[0][1], x[1][1]);
    om.store16(0, col, ((n0 + 16 * (2 * m + (g & 1))) >> 3) + (g >> 1), x[0][0], x[0][1], x[1][0], x[1][1]);
  }
  if (L < 3) return;
  // level 2: block m holds the values of row tiles 2m, 2m + 1; w[m] = (element g, element 4 + g)
  uint32_t w[4];
  w[0] = pack2<VT>(p2[0], p2[1]);
  w[1] = pack2<VT>(p2[2], p2[3]);
  w[2] = w[3] = 0u;
  transpose_rows(w);  // lane row j < 2: w[p] = elements (p, 4 + p) of block j
  const uint32_t d0 = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u), d1 = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u);
  const uint32_t d2 = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u), d3 = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
  if (g < 2) om.store16(2, col, (n0 >> 5) + g, d0, d1, d2, d3);
}

// ---- 16-bit feature maps, 16-bit pyramid (the mixed-precision configuration).  One workgroup = 8 waves owns
// one image row x 128 w1, 16 columns per wave, and sweeps w2 in chunks of 64.  Two staged chunks are in flight
// in registers while a third is multiplied out of LDS, so the HBM latency of the f2 stream is covered from
// inside the workgroup (with one chunk in flight the waves spent half their time waiting: rocprof,
// profiles/r01).  A wave's f1 fragments are 32 registers, which leaves room for the second chunk at 4 waves
// per SIMD; the price, every f2 fragment feeding one MFMA instead of two, is LDS bandwidth the kernel has to
// spare (it is bound by HBM).
constexpr int BT16 = 512;
constexpr int TW16 = 16;

template <typename T, int FAST>
__global__ __launch_bounds__(BT16, 4) void build16_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                          T* __restrict__ vol, rc_dev_layout lay, int D, float scale,
                                                          int n_tiles, int n_rows, int rows_per_xcd) {
  using C = build_cfg<T>;
  constexpr int NP = 4;  // staging passes of 64 rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ldsF = reinterpret_cast<T*>(smem);

  // XCD-aware mapping: the tiles of one image row get block ids congruent mod 8 (one XCD, one L2)
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int row = rows_per_xcd ? (id & 7) * rows_per_xcd + j / n_tiles : (j / n_tiles) * 8 + (id & 7);
  const int tile = j % n_tiles;
  if (row >= n_rows) return;
  const int b = row / lay.H, h = row - b * lay.H;
  const int c0 = tile * TM;
  const int W1 = lay.W1, W2 = lay.W2, L = lay.L;

  const int wave = threadIdx.x >> 6;
  const int cw = c0 + wave * TW16;  // first w1 column of this wave
  const int nk = FAST ? C::NKS : (D + C::KSTEP - 1) / C::KSTEP;

  const T* f1row = f1 + ((int64_t)b * D * lay.H + h) * W1;
  const T* f2row = f2 + ((int64_t)b * D * lay.H + h) * W2;
  const int64_t ks1 = (int64_t)lay.H * W1, ks2 = (int64_t)lay.H * W2;
  const int n_chunks = (W2 + C::TN - 1) / C::TN;

  T* const buf0 = ldsF;
  T* const buf1 = ldsF + C::LDS_BYTES / (int)sizeof(T);
  u4v regs[2][NP];
  s8v bfrag[C::NKS];
  RC_STAMP(0);
  {
    // both halves (64 columns each) of the f1 tile and the first two f2 chunks are requested before anything is
    // waited for; the halves go through the two staging buffers side by side
    u4v fa[NP], fb[NP];
    stage_load<T, FAST, NP>(fa, f1row, ks1, W1, c0, D);
    stage_load<T, FAST, NP>(fb, f1row, ks1, W1, c0 + 64, D);
    stage_load<T, FAST, NP>(regs[0], f2row, ks2, W2, 0, D);
    if (n_chunks > 1) stage_load<T, FAST, NP>(regs[1], f2row, ks2, W2, C::TN, D);
    RC_STAMP(1);
    stage_store<T, C::PF, NP>(fa, buf0);
    stage_store<T, C::PF, NP>(fb, buf1);
    RC_STAMP(2);
    __syncthreads();
    RC_STAMP(3);
    const short* src = reinterpret_cast<const short*>((wave >> 2) ? buf1 : buf0);
#pragma unroll
    for (int kb = 0; kb < C::NKS; ++kb)
      bfrag[kb] = kb < nk ? tr_frag<C::PF>(src, kb, (wave & 3) * TW16) : (s8v)(0);
    __syncthreads();  // the fragments have been read: the buffers are free for f2
    RC_STAMP(4);
    stage_store<T, C::PF, NP>(regs[0], buf0);
    if (n_chunks > 2) stage_load<T, FAST, NP>(regs[0], f2row, ks2, W2, 2 * C::TN, D);
    RC_STAMP(5);
  }

  out_map<T> om;
  om.blk_bytes = (uint32_t)lay.pitch * 16u;
  om.W1 = W1;
#pragma unroll
  for (int i = 0; i < RC_MAX_LEVELS; ++i) {
    const int nb = lay.widthp[i < L ? i : 0] >> out_map<T>::LG;
    om.nb[i] = (uint32_t)nb;
    om.kb[i] = (uint32_t)(lay.bias[i < L ? i : 0] - RC_WRAP_ROWS / out_map<T>::RB);
    om.slowmod[i] = lay.bias[i < L ? i : 0] > 3 * nb;
    om.base[i] = reinterpret_cast<char*>(vol) +
                 (lay.offset[i < L ? i : 0] * (int64_t)sizeof(T) + (int64_t)row * nb * om.blk_bytes);
  }

  // Chunk c is multiplied out of staging buffer c & 1 while chunk c + 1 (register set `nxt`) moves into the
  // other one and chunk c + 3 is requested into the registers it frees: one barrier per chunk, and neither
  // the LDS stores nor the wait for the global loads sit between the barrier and the MFMAs.
  auto chunk = [&](u4v (&nxt)[NP], int c, const T* cur_buf, T* nxt_buf) {
    const int n0 = c * C::TN;
    __syncthreads();  // chunk c is in cur_buf; everybody is done reading nxt_buf (chunk c - 1)
    RC_STAMP(6 + 4 * c);
    const bool active = cw < W1;  // wave-uniform: a wave past the row end only helps with the staging
    f4v acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = (f4v)(0.f);
    if (active) {
#pragma unroll
      for (int kb = 0; kb < C::NKS; ++kb) {
        if (kb < nk) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            if constexpr (RC_LAB_ABLATE & 2) {
              if (kb == 0) acc[rt][0] += (float)reinterpret_cast<const short*>(cur_buf)[threadIdx.x + rt] + (float)bfrag[rt][0];
            } else {
              acc[rt] = mfma16<T>(tr_frag<C::PF>(reinterpret_cast<const short*>(cur_buf), kb, rt * 16), bfrag[kb], acc[rt]);
            }
          }
        }
      }
    }
    RC_STAMP(7 + 4 * c);
    if (c + 1 < n_chunks) {
      stage_store<T, C::PF, NP>(nxt, nxt_buf);
      if (c + 3 < n_chunks && !(RC_LAB_ABLATE & 4)) stage_load<T, FAST, NP>(nxt, f2row, ks2, W2, n0 + 3 * C::TN, D);
    }
    RC_STAMP(8 + 4 * c);
    if (active) epilogue16_col<T>(acc, scale, om, L, n0, cw);
    RC_STAMP(9 + 4 * c);
  };
  for (int c = 0; c < n_chunks; c += 2) {
    chunk(regs[1], c, buf0, buf1);
    if (c + 1 < n_chunks) chunk(regs[0], c + 1, buf1, buf0);
  }
}

// ---- 16-bit feature maps whose rows are whole 16-byte segments (W1, W2 multiples of 8, D = 256): the f2 stream and
// the f1 tile go global -> LDS by LDS-DMA (global_load_lds_dwordx4), no staging registers and no ds_write pass.
#ifndef RC_LAB_BUILD_DMA_POLICY  // lab: cache policy of the DMA loads of f2 ("" plain, " nt", " sc1", ...)
#define RC_LAB_BUILD_DMA_POLICY ""
#endif
// f1 (glds16<true>): non-temporal.  A workgroup's f1 tile has no other reader, and left in the Infinity Cache its
// 123 MB (KITTI, 8 pairs) push out pyramid lines the lookups are about to read: with " nt" the build itself takes
// 108.8 instead of 111 us in a step and each of the 32 lookups 9.2-9.4 instead of 9.5 us.  (The same policy on f2,
// which the three tiles of a row all read, costs the build 30 us: its re-reads then miss.)
#ifndef RC_LAB_BUILD_F1_POLICY
#define RC_LAB_BUILD_F1_POLICY " nt"
#endif
// That frees the registers for 32-column wave tiles: 8 waves = 4 column groups x 2 halves of the 64-row chunk,
// every f2 fragment read from LDS feeds TWO MFMAs (the 16-column kernel above reads every fragment for one: its
// MFMA phase is bound by LDS bandwidth, 2 200 cycles per chunk where the matrix pipe needs 1 000:
// profiles/r03/lab_build_stamps.txt).  A DMA instruction writes 1 KiB of LDS in lane order, so a staged tile is
// [256 k][128 bytes] without padding; bank conflicts of the transposed reads are avoided by an XOR swizzle of
// the 16-byte segments, applied on the SOURCE side of the DMA (each lane fetches the segment its slot holds):
//     physical segment = segment ^ (((k >> 1) & 3) << 1).
// Stores are range-checked buffer stores (cells outside the image get an out-of-range offset and are dropped):
// every wave issues the same number of store instructions per chunk, which is what lets a COUNTED s_waitcnt
// vmcnt(n) wait for the chunk's DMA without waiting for the stores issued after it.
constexpr int G_ROW = 128;           // bytes per LDS row: 64 elements
constexpr int G_BUF = KMAX * G_ROW;  // one staged tile: 32 KiB

// one DMA instruction: lane l fetches 16 bytes from gsrc, they land at lds_dst + 16 l (lds_dst wave-uniform)
// ONCE: the data has one reader (a workgroup's own f1 tile)
template <bool ONCE = false>
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  if constexpr (ONCE)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" RC_LAB_BUILD_F1_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" RC_LAB_BUILD_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// [256 k][64 columns from n0] of one image row -> the LDS tile at byte address lds_buf: 4 DMA instructions per
// wave, instruction p of wave w fills rows 64 p + 8 w .. + 7.  Segments past the end of the row re-read its start
// (what they produce lands in padding cells or in columns that are not stored).
template <typename T, bool ONCE = false>
__device__ __forceinline__ void dma_tile(const T* rowbase, int64_t kstride, int W, int n0, uint32_t lds_buf, int wave) {
  const int lane = threadIdx.x & 63;
  const int kr = lane >> 3;
  const int ls = (lane & 7) ^ (((kr >> 1) & 3) << 1);  // the logical segment this lane's slot holds
  int col = n0 + ls * 8;
  col = (unsigned)col >= (unsigned)W ? 0 : col;  // (also the columns before the row of a shifted f1 tile)
  const T* src = rowbase + (int64_t)(wave * 8 + kr) * kstride + col;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    glds16<ONCE>(src + (int64_t)p * 64 * kstride, lds_buf + (uint32_t)((p * 64 + wave * 8) * G_ROW));
}

// byte offset inside a swizzled tile of this lane's transposed-read address for the 16 columns from `co`
// (a multiple of 16), k block 0; k block kb adds 4096 kb, the second half of a fragment 2048
__device__ __forceinline__ uint32_t frag_off(int co) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i = lane & 15;
  const int kl = 4 * g + (i >> 2);
  const int sw = ((kl >> 1) & 3) << 1;
  const int seg = (co >> 3) + ((i >> 1) & 1);
  return (uint32_t)(kl * G_ROW + ((seg ^ sw) << 4) + (i & 1) * 8);
}
__device__ __forceinline__ s8v tr_frag_g(const unsigned char* tile, uint32_t off, int kb) {
  typedef s4v __attribute__((address_space(3))) * lds_s4p;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(tile + off + kb * 4096));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(tile + off + kb * 4096 + 2048));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// pooled pair: round(0.5 * (lo(x0) + hi(x0))) | round(0.5 * (lo(x1) + hi(x1))) << 16 -- the fp32 mean of two stored
// values rounded once, what avg_pool2d does on a 16-bit volume.  fp16: the mixed-precision FMA reads the halves
// directly (sum = lo * 1.0 + hi in fp32, one instruction instead of two conversions and an add) and v_fma_mixlo/hi
// scale, round and pack in one (0.5 * sum is exact, so this is the same single rounding)
template <typename VT> __device__ __forceinline__ uint32_t mean_pair(uint32_t x0, uint32_t x1) {
  return pack2<VT>(0.5f * (unpack_lo<VT>(x0) + unpack_hi<VT>(x0)), 0.5f * (unpack_lo<VT>(x1) + unpack_hi<VT>(x1)));
}
template <> __device__ __forceinline__ uint32_t mean_pair<rc_half>(uint32_t x0, uint32_t x1) {
  float s0, s1;
  uint32_t r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(s0) : "v"(x0));
  asm("v_fma_mix_f32 %0, %1, 1.0, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(s1) : "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, 0.5, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, 0.5, 0 op_sel_hi:[0,0,0]"
      : "=&v"(r)
      : "v"(s0), "v"(s1));
  return r;
}

// one stored level of one image row, as a range-checked buffer
struct out_slab {
  __amdgpu_buffer_rsrc_t rs;
  uint32_t bytes, nb, kb;
};
// 16-byte block of group G of pixel `col` at the slab's level (shift `i`); cells outside the image are dropped
// KEEP: plain stores, the pyramid stays in the caches for the lookups (see build16g_kernel); else non-temporal
template <int I, bool KEEP>
__device__ __forceinline__ void slab_store16(const out_slab& sl, uint32_t blk_bytes, int W1, int col, int G, uint32_t a,
                                             uint32_t b, uint32_t c, uint32_t d) {
  uint32_t t = (uint32_t)(G - ((col >> I) >> 3)) + sl.kb;  // >= 0 by choice of bias
  t = min(t, t - 2 * sl.nb);
  t = min(t, t - sl.nb);
  uint32_t off = __umul24(t, blk_bytes) + (uint32_t)col * 16u;
  asm volatile("" : "+v"(off));  // computed for every lane: the select stays a v_cndmask
  off = ((uint32_t)col < (uint32_t)W1 && (uint32_t)G < sl.nb) ? off : sl.bytes;  // (col < 0: shifted first tile)
  const u4v v = {a, b, c, d};
  if constexpr (RC_LAB_ABLATE & 1) {
    asm volatile("" ::"v"(v), "v"(off));
    return;
  }
  if constexpr (KEEP) __builtin_amdgcn_raw_buffer_store_b128(v, sl.rs, (int)off, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(v, sl.rs, (int)off, 0, 2 /* nt */);
}

// epilogue of a 32 (w2) x 32 (w1) wave tile: acc[rt][ct], lane (g, li) holds rows n0 + 16 rt + 4 g + {0..3} of
// column cw + 16 ct + li.  Exactly 2 (L < 3) or 3 store instructions.
template <typename VT, bool KEEP>
__device__ __forceinline__ void epilogue16_g(const f4v (&acc)[2][2], float scale, const out_slab& s0, const out_slab& s2,
                                             uint32_t blk_bytes, int W1, int L, int n0, int cw) {
  const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int col_sw = cw + 16 * (g & 1) + li;  // column a lane serves after a swap16 between the column tiles
  uint32_t q1[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    uint32_t x[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      x[ct][0] = pack2<VT>(acc[rt][ct][0] * scale, acc[rt][ct][1] * scale);
      x[ct][1] = pack2<VT>(acc[rt][ct][2] * scale, acc[rt][ct][3] * scale);
      q1[rt][ct] = mean_pair<VT>(x[ct][0], x[ct][1]);
    }
    swap16(x[0][0], x[1][0]);
    swap16(x[0][1], x[1][1]);
    slab_store16<0, KEEP>(s0, blk_bytes, W1, col_sw, ((n0 + 16 * rt) >> 3) + (g >> 1), x[0][0], x[0][1], x[1][0], x[1][1]);
  }
  if (L < 3) return;  // level 1 is not stored; level 2 is its pairwise mean
  uint32_t w[4];      // w[ct] = level-2 values of row tiles 0, 1: block elements g and 4 + g
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) w[ct] = mean_pair<VT>(q1[0][ct], q1[1][ct]);
  w[2] = w[3] = 0u;
  transpose_rows(w);  // lane row j < 2: w[p] = elements (p, 4 + p) of the block of column tile j
  const uint32_t d0 = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u), d1 = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u);
  const uint32_t d2 = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u), d3 = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
  slab_store16<2, KEEP>(s2, blk_bytes, g < 2 ? W1 : 0, col_sw, n0 >> 5, d0, d1, d2, d3);
}

// F1_ONCE: the pyramid can stay in the 256 MiB Infinity Cache for the lookups (the launcher decides by its size).
// The f1 tile, which has one reader, is then fetched non-temporally (glds16), and the pyramid is written with plain
// stores instead of non-temporal ones: the build itself is 10 us slower for it (KITTI, 8 pairs: 102 -> 112 us; the
// dirty pyramid lines compete with f2 in the L2), but every lookup finds the pyramid in the cache from the first
// one on, 8.8 instead of 9.3 us on average over a step's 32 (after non-temporal stores the first lookup takes 14 us,
// the second 11, and it is the eighth before they are down at 9): 0.416 -> 0.410 ms per step.  Level 2 alone
// written plain: 0.413.  A larger pyramid is streamed out as before.
template <typename T, bool F1_ONCE>
__global__ __launch_bounds__(BT16, 4) void build16g_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                           T* __restrict__ vol, rc_dev_layout lay, float scale,
                                                           int n_tiles, int n_rows, int rows_per_xcd,
                                                           int line_tiles) {
  using C = build_cfg<T>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // XCD-aware mapping: the tiles of an image row on one XCD as in build16_kernel, and (rows_per_xcd > 0) every XCD
  // a run of consecutive rows instead of every eighth: rows of 624 bytes (W = 312) do not end on 128-byte lines,
  // and the line two rows share is then fetched into one L2 by two workgroups that follow each other, not into two
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int row = rows_per_xcd ? (id & 7) * rows_per_xcd + j / n_tiles : (j / n_tiles) * 8 + (id & 7);
  const int tile = j % n_tiles;
  if (row >= n_rows) return;
  const int b = row / lay.H, h = row - b * lay.H;
  const int W1 = lay.W1, W2 = lay.W2, L = lay.L;
  constexpr int D = KMAX;
  const T* f1row = f1 + ((int64_t)b * D * lay.H + h) * W1;
  const T* f2row = f2 + ((int64_t)b * D * lay.H + h) * W2;
  // line_tiles (the k planes are a whole number of 128-byte lines apart): the tiles of f1 start on 128-byte lines
  // of memory instead of at the row's first column -- a row of 624 bytes starts anywhere in a line, a 256-byte
  // piece of it then straddles three lines, and the two it shares with the neighbouring tiles were fetched twice
  // (the f1 stream is non-temporal: a line does not wait in the L2 for its second reader).  The first tile is
  // narrower by the row's offset in its line, the launcher's tile count covers the widest case, and a tile that
  // starts past the row has nothing to do.
  const int shift = line_tiles ? (int)((reinterpret_cast<uintptr_t>(f1row) >> 1) & 63) : 0;  // columns, multiple of 8
  const int c0 = tile * TM - shift;
  if (c0 >= W1) return;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wave & 3, rh = wave >> 2;  // column group (32 columns), half of the chunk's rows
  const int cw = c0 + cg * 32;
  const bool active = cw < W1 && cw + 32 > 0;  // wave-uniform: a wave outside the row only helps with the staging

  const int64_t ks1 = (int64_t)lay.H * W1, ks2 = (int64_t)lay.H * W2;
  const int n_chunks = (W2 + C::TN - 1) / C::TN;
  typedef unsigned char __attribute__((address_space(3))) * lds_p;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_p)smem;  // LDS byte address of the dynamic allocation

  RC_STAMP(0);
  // ---- the first f2 chunk is requested into registers (the LDS buffers are busy with f1 until its fragments have
  // been read), then the f1 tile: both halves (64 columns each) through the two buffers.  One memory round trip
  // covers both.
  u4v c0regs[4];
  const int sseg = threadIdx.x & 7, skr = threadIdx.x >> 3;  // segment and row (of 64 per pass) this thread stages
  {
    int col = sseg * 8;
    col = col >= W2 ? 0 : col;
    const T* src = f2row + (int64_t)skr * ks2 + col;
#pragma unroll
    for (int p = 0; p < 4; ++p) c0regs[p] = *reinterpret_cast<const u4v*>(src + (int64_t)p * 64 * ks2);
  }
  dma_tile<T, F1_ONCE>(f1row, ks1, W1, c0, lds0, wave);
  dma_tile<T, F1_ONCE>(f1row, ks1, W1, c0 + 64, lds0 + G_BUF, wave);
  RC_STAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  RC_STAMP(2);
  s8v bfrag[C::NKS][2];
  {
    const unsigned char* src = smem + (cg >> 1) * G_BUF;
    const uint32_t o0 = frag_off((cg & 1) * 32), o1 = frag_off((cg & 1) * 32 + 16);
#pragma unroll
    for (int kb = 0; kb < C::NKS; ++kb) {
      bfrag[kb][0] = tr_frag_g(src, o0, kb);
      bfrag[kb][1] = tr_frag_g(src, o1, kb);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // the fragments have been read: the buffers are free for f2
  RC_STAMP(3);
  {
    // chunk 0 from its registers into buffer 0 (same swizzle as the DMA's), then chunk 1 by DMA into buffer 1
    unsigned char* dst = smem + skr * G_ROW + ((sseg ^ (((skr >> 1) & 3) << 1)) << 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u4v*>(dst + p * 64 * G_ROW) = c0regs[p];
  }
  if (n_chunks > 1) dma_tile<T>(f2row, ks2, W2, C::TN, lds0 + G_BUF, wave);
  RC_STAMP(4);

  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;
  out_slab s0, s2;
  {
    const int nb0 = lay.widthp[0] >> 3, nb2 = lay.widthp[L > 2 ? 2 : 0] >> 3;
    s0.nb = (uint32_t)nb0;
    s0.kb = (uint32_t)(lay.bias[0] - RC_WRAP_ROWS / 8);
    s0.bytes = (uint32_t)nb0 * blk_bytes;
    s0.rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(vol) + (lay.offset[0] * (int64_t)sizeof(T) + (int64_t)row * s0.bytes), 0, (int)s0.bytes,
        0x00020000);
    s2.nb = (uint32_t)nb2;
    s2.kb = (uint32_t)(lay.bias[L > 2 ? 2 : 0] - RC_WRAP_ROWS / 8);
    s2.bytes = (uint32_t)nb2 * blk_bytes;
    s2.rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(vol) + (lay.offset[L > 2 ? 2 : 0] * (int64_t)sizeof(T) + (int64_t)row * s2.bytes), 0,
        (int)s2.bytes, 0x00020000);
  }
  const uint32_t oa0 = frag_off(32 * rh), oa1 = frag_off(32 * rh + 16);
  const bool three = L >= 3;  // store instructions per chunk of an active wave: 3 or 2

  // The two waves that share a SIMD (w and w + 4: the row halves of one column group) run the same program between
  // the same barriers; left alone they multiply together and convert together, and the matrix pipe idles through
  // every epilogue.  Waves 4-7 therefore run half a chunk behind: they keep a chunk's sums across the barrier and
  // write them out while their partner multiplies the next chunk, then multiply while the partner writes out
  // (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Nothing else is live across the barrier in this kernel,
  // so the deferral costs no registers.  MEASURED (profiles/r03/lab_build_stagger.txt): no gain here -- the late
  // wave's epilogue, beside its partner's MFMAs, takes as much longer (1 400 -> 2 800 cycles) as its MFMA phase gets
  // shorter (2 050 -> 950): the chunk takes 3 900 cycles either way, so the stagger is left switched off.
#ifndef RC_BUILD_STAGGER
#define RC_BUILD_STAGGER 0
#endif
  const bool late = RC_BUILD_STAGGER && rh != 0;
  f4v acc[2][2];
  auto multiply = [&](int c) {
    const unsigned char* cur = smem + (c & 1) * G_BUF;
    if constexpr (RC_LAB_ABLATE & 2) {  // lab: no LDS reads, no MFMA
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e >> 1][e & 1] = (f4v)((float)cur[threadIdx.x + e] + (float)bfrag[e][0][0]);
      return;
    }
#pragma unroll
    for (int kb = 0; kb < C::NKS; ++kb) {
      const s8v a0 = tr_frag_g(cur, oa0, kb), a1 = tr_frag_g(cur, oa1, kb);
      if (kb == 0) {
        acc[0][0] = mfma16<T>(a0, bfrag[0][0], (f4v)(0.f));
        acc[0][1] = mfma16<T>(a0, bfrag[0][1], (f4v)(0.f));
        acc[1][0] = mfma16<T>(a1, bfrag[0][0], (f4v)(0.f));
        acc[1][1] = mfma16<T>(a1, bfrag[0][1], (f4v)(0.f));
      } else {
        acc[0][0] = mfma16<T>(a0, bfrag[kb][0], acc[0][0]);
        acc[0][1] = mfma16<T>(a0, bfrag[kb][1], acc[0][1]);
        acc[1][0] = mfma16<T>(a1, bfrag[kb][0], acc[1][0]);
        acc[1][1] = mfma16<T>(a1, bfrag[kb][1], acc[1][1]);
      }
    }
  };
  for (int c = 0; c < n_chunks; ++c) {
    // chunk c >= 1 has landed once all but the vector-memory operations issued after its DMA are done: the stores
    // of the epilogue that followed it (none yet for a late wave at c = 1, none ever for an inactive wave)
    if (c == 0) {  // chunk 0 came through registers: this wave's LDS stores of it must have landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (!active || (late && c == 1)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (three) {
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // everybody's share of chunk c is in LDS; everybody is done reading chunk c - 1
    RC_STAMP(6 + 4 * c);
    if (c >= 1 && c + 1 < n_chunks) dma_tile<T>(f2row, ks2, W2, (c + 1) * C::TN, lds0 + ((c + 1) & 1) * G_BUF, wave);
    if (active) {
      if (late) {
        if (c >= 1) epilogue16_g<T, F1_ONCE>(acc, scale, s0, s2, blk_bytes, W1, L, (c - 1) * C::TN + 32 * rh, cw);
        RC_STAMP(7 + 4 * c);
        multiply(c);
      } else {
        multiply(c);
        RC_STAMP(7 + 4 * c);
        epilogue16_g<T, F1_ONCE>(acc, scale, s0, s2, blk_bytes, W1, L, c * C::TN + 32 * rh, cw);
      }
      RC_STAMP(9 + 4 * c);
    }
  }
  if (active && late) epilogue16_g<T, F1_ONCE>(acc, scale, s0, s2, blk_bytes, W1, L, (n_chunks - 1) * C::TN + 32 * rh, cw);
}

template <typename T, typename VT, int FAST>
__global__ __launch_bounds__(BT) void build_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                   VT* __restrict__ vol, rc_dev_layout lay, int D, float scale,
                                                   int n_tiles, int n_rows, int rows_per_xcd) {
  using C = build_cfg<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ldsF = reinterpret_cast<T*>(smem);

  // XCD-aware mapping: the tiles of one image row get block ids congruent mod 8 (one XCD, one
  // L2), since they all re-read that row's f2.
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int row = rows_per_xcd ? (id & 7) * rows_per_xcd + j / n_tiles : (j / n_tiles) * 8 + (id & 7);
  const int tile = j % n_tiles;
  if (row >= n_rows) return;
  const int b = row / lay.H, h = row - b * lay.H;
  const int c0 = tile * TM;
  const int W1 = lay.W1, W2 = lay.W2, L = lay.L;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int cw = c0 + wave * TW;  // first w1 column of this wave
  const int nk = FAST ? C::NKS : (D + C::KSTEP - 1) / C::KSTEP;  // FAST: D == KMAX, loops fully unrolled

  const T* f1row = f1 + ((int64_t)b * D * lay.H + h) * W1;
  const T* f2row = f2 + ((int64_t)b * D * lay.H + h) * W2;
  const int64_t ks1 = (int64_t)lay.H * W1, ks2 = (int64_t)lay.H * W2;

  u4v regs[8];

  // ---- f1 tile -> per-wave register-resident B operand
  s8v bfrag16[C::IS16 ? C::NKS : 1][2];
  float bfrag32[C::IS16 ? 1 : C::NKS][2];
  if constexpr (C::IS16) {
    // the 128 columns pass through the staging tile in two halves of 64
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      stage_load<T, FAST>(regs, f1row, ks1, W1, c0 + 64 * half, D);
      if (half) __syncthreads();  // the first half's fragments have been read
      stage_store<T, C::PF>(regs, ldsF);
      __syncthreads();
      if ((wave >> 1) == half) {
#pragma unroll
        for (int kb = 0; kb < C::NKS; ++kb) {
          if (kb < nk) {
            bfrag16[kb][0] = tr_frag<C::PF>(reinterpret_cast<const short*>(ldsF), kb, (wave & 1) * 32);
            bfrag16[kb][1] = tr_frag<C::PF>(reinterpret_cast<const short*>(ldsF), kb, (wave & 1) * 32 + 16);
          } else {
            bfrag16[kb][0] = (s8v)(0);
            bfrag16[kb][1] = (s8v)(0);
          }
        }
      }
    }
    stage_load<T, FAST>(regs, f2row, ks2, W2, 0, D);  // chunk 0 of f2, in flight
  } else {
    stage_load<T, FAST>(regs, f2row, ks2, W2, 0, D);
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int k = ks * 4 + g, w1 = cw + ct * 16 + li;
        bfrag32[ks][ct] = (k < D && w1 < W1) ? (float)f1row[(int64_t)k * ks1 + w1] : 0.f;
      }
    }
  }

  out_map<VT> om;
  om.blk_bytes = (uint32_t)lay.pitch * 16u;
  om.W1 = W1;
#pragma unroll
  for (int i = 0; i < RC_MAX_LEVELS; ++i) {
    const int nb = lay.widthp[i < L ? i : 0] >> out_map<VT>::LG;
    om.nb[i] = (uint32_t)nb;
    om.kb[i] = (uint32_t)(lay.bias[i < L ? i : 0] - RC_WRAP_ROWS / out_map<VT>::RB);
    om.slowmod[i] = lay.bias[i < L ? i : 0] > 3 * nb;
    om.base[i] = reinterpret_cast<char*>(vol) +
                 (lay.offset[i < L ? i : 0] * (int64_t)sizeof(VT) + (int64_t)row * nb * om.blk_bytes);
  }

  const int n_chunks = (W2 + C::TN - 1) / C::TN;
  for (int c = 0; c < n_chunks; ++c) {
    const int n0 = c * C::TN;
    __syncthreads();  // previous readers of ldsF are done
    stage_store<T, C::PF>(regs, ldsF);
    __syncthreads();
    if (c + 1 < n_chunks) stage_load<T, FAST>(regs, f2row, ks2, W2, n0 + C::TN, D);
    if (cw >= W1) continue;  // wave-uniform: a wave past the row end only helps with the staging

    f4v acc[C::NRT][2];
#pragma unroll
    for (int rt = 0; rt < C::NRT; ++rt) {
      acc[rt][0] = (f4v)(0.f);
      acc[rt][1] = (f4v)(0.f);
    }
    if constexpr (C::IS16) {
#pragma unroll
      for (int kb = 0; kb < C::NKS; ++kb) {
        if (kb < nk) {
#pragma unroll
          for (int rt = 0; rt < C::NRT; ++rt) {
            const s8v a = tr_frag<C::PF>(reinterpret_cast<const short*>(ldsF), kb, rt * 16);
            acc[rt][0] = mfma16<T>(a, bfrag16[kb][0], acc[rt][0]);
            acc[rt][1] = mfma16<T>(a, bfrag16[kb][1], acc[rt][1]);
          }
        }
      }
    } else {
      const float* lf = reinterpret_cast<const float*>(ldsF);
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        if (ks < nk) {
#pragma unroll
          for (int rt = 0; rt < C::NRT; ++rt) {
            const float a = lf[(ks * 4 + g) * C::PF + rt * 16 + li];
            acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag32[ks][0], acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfrag32[ks][1], acc[rt][1], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue: scale, round to the storage type, pool, write every level from registers
    if constexpr (std::is_same<VT, float>::value) {
#pragma unroll
      for (int hh = 0; hh < C::NRT / 2; ++hh)
        epilogue32(acc[2 * hh][0], acc[2 * hh][1], acc[2 * hh + 1][0], acc[2 * hh + 1][1], scale, om, L,
                   n0 + 32 * hh, cw);
    } else {
      epilogue16<VT>(acc, scale, om, L, n0, cw);
    }
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
  vol[rc_cell<RB>(lay, 0, bh, w1, w2)] = rc_from_float<VT>(acc * scale);
}

// stored level `level` (even, >= 2) from stored level `level - 2`: two pooling steps, each the fp32 mean of
// two values of the storage type, rounded to it (avg_pool2d on a 16-bit volume, core/corr.py:42)
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
  float m[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float a = rc_to_float(vol[rc_cell<RB>(lay, level - 2, bh, w1, 4 * w2 + 2 * e)]);
    const float bb = rc_to_float(vol[rc_cell<RB>(lay, level - 2, bh, w1, 4 * w2 + 2 * e + 1)]);
    m[e] = rc_round_as<VT>(0.5f * (a + bb));
  }
  vol[rc_cell<RB>(lay, level, bh, w1, w2)] = rc_from_float<VT>(0.5f * (m[0] + m[1]));
}

template <typename T, typename VT>
int launch_naive(const void* f1, const void* f2, void* vol, const rc_layout* lay, int D, float scale,
                 hipStream_t st) {
  const rc_dev_layout d = rc_to_dev(lay);
  int64_t total = lay->B * lay->H * lay->W2 * lay->W1;
  hipLaunchKernelGGL((naive_corr_kernel<T, VT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     (const T*)f1, (const T*)f2, (VT*)vol, d, D, scale, total);
  int rc = rc_check_launch("rc_build_pyramid(naive corr)");
  for (int i = 2; i < lay->L && !rc; i += 2) {
    total = lay->B * lay->H * (int64_t)lay->width[i] * lay->W1;
    hipLaunchKernelGGL((naive_pool_kernel<VT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (VT*)vol, d, i, total);
    rc = rc_check_launch("rc_build_pyramid(naive pool)");
  }
  return rc;
}

// bias[i] for the build's block-row arithmetic: the smallest multiple of the level's block-row
// count with G - (col >> i) / RB - RC_WRAP_ROWS / RB + bias >= 0 for every group G >= 0
void set_build_bias(rc_dev_layout& d, int rb) {
  for (int i = 0; i < d.L; ++i) {
    const int nb = d.widthp[i] / rb;
    const int lo = (((d.pitch + TM) >> i) / rb) + RC_WRAP_ROWS / rb;
    d.bias[i] = (lo + nb - 1) / nb * nb;
  }
}

template <typename T, typename VT>
int launch_build(const void* f1, const void* f2, void* vol, const rc_layout* lay, int D, float scale,
                 hipStream_t st) {
  using C = build_cfg<T>;
  static bool attr_set[64] = {};  // per device; raising the limit is idempotent, races are harmless
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    rc_set_error("no current HIP device");
    return RC_ENODEVICE;
  }
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build_kernel<T, VT, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build_kernel<T, VT, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build_kernel<T, VT, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if constexpr (C::IS16 && std::is_same<T, VT>::value) {
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build16_kernel<T, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::LDS_BYTES);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build16_kernel<T, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::LDS_BYTES);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build16_kernel<T, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::LDS_BYTES);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build16g_kernel<T, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_BUF);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&build16g_kernel<T, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_BUF);
    }
    if (e != hipSuccess) {
      rc_set_error("hipFuncSetAttribute(%d bytes LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
      return (int)e;
    }
    attr_set[dev] = true;
  }
  rc_dev_layout d = rc_to_dev(lay);
  set_build_bias(d, 16 / (int)sizeof(VT));
  const int n_tiles = (int)((lay->W1 + TM - 1) / TM);
  const int n_rows = (int)(lay->B * lay->H);
  const int64_t blocks = (int64_t)((n_rows + 7) / 8) * 8 * n_tiles;
  RC_REQUIRE(blocks < (1ll << 31), RC_EINVAL, "grid too large");
  constexpr int EPS = 16 / (int)sizeof(T);
  // the branch-free staging path: dword-aligned rows, full depth, 32-bit byte offsets inside one image
  const bool fast = (lay->W1 * sizeof(T)) % 4 == 0 && ((uintptr_t)f1) % 4 == 0 && (lay->W2 * sizeof(T)) % 4 == 0 &&
                    ((uintptr_t)f2) % 4 == 0 && lay->W1 >= EPS && lay->W2 >= EPS && D == KMAX &&
                    (int64_t)D * lay->H * (lay->W1 > lay->W2 ? lay->W1 : lay->W2) * (int64_t)sizeof(T) < (1ll << 31);
  const bool whole = lay->W1 % EPS == 0 && lay->W2 % EPS == 0;  // no segment straddles the end of a row
  // rows of one XCD: a run of consecutive ones for the 16-bit kernels (see build16g_kernel: KITTI 110.2 -> 108.8 us
  // with 13 % fewer bytes read, SceneFlow 59.8 -> 58.1 us), every eighth row for fp32 (0.444 against 0.452 ms);
  // RC_BUILD_ROW_ORDER=interleaved|contiguous overrides
  const char* order = getenv("RC_BUILD_ROW_ORDER");
  const bool contiguous = order ? !strcmp(order, "contiguous") : sizeof(T) == 2;
  const int rows_per_xcd = contiguous ? (n_rows + 7) / 8 : 0;
#define RC_LAUNCH_BUILD(F)                                                                               \
  hipLaunchKernelGGL((build_kernel<T, VT, F>), dim3((unsigned)blocks), dim3(BT), C::LDS_BYTES, st, (const T*)f1, \
                     (const T*)f2, (VT*)vol, d, D, scale, n_tiles, n_rows, rows_per_xcd)
#define RC_LAUNCH_BUILD16(F)                                                                             \
  hipLaunchKernelGGL((build16_kernel<T, F>), dim3((unsigned)blocks), dim3(BT16), 2 * C::LDS_BYTES, st, (const T*)f1, \
                     (const T*)f2, (T*)vol, d, D, scale, n_tiles, n_rows, rows_per_xcd)
  if constexpr (C::IS16 && std::is_same<T, VT>::value) {
    // rows of whole, 16-byte aligned segments and levels wide enough for the branch-free block-row arithmetic:
    // the LDS-DMA kernel
    bool dma = fast && whole && lay->W1 % 8 == 0 && lay->W2 % 8 == 0 && ((uintptr_t)f1) % 16 == 0 &&
               ((uintptr_t)f2) % 16 == 0 && !getenv("RC_BUILD_NO_DMA");
    for (int i = 0; i < lay->L; i += 2) dma = dma && d.bias[i] <= 3 * (lay->widthp[i] / 8);
    // a pyramid that can stay in the 256 MiB Infinity Cache for the lookups is worth protecting from the f1 stream
    // (see glds16); a larger one is not, and there the non-temporal f1 costs the build 8 % (64 KITTI pairs)
    // (RC_BUILD_F1_POLICY=stream|plain overrides the size rule: tests, measurements)
    const char* policy = getenv("RC_BUILD_F1_POLICY");
    const bool f1_once = policy ? !strcmp(policy, "stream") : lay->total * (int64_t)sizeof(T) <= (200ll << 20);
    // f1 tiles on the 128-byte lines of memory (see the kernel) where that means the same for every k plane and
    // the rows do not start on lines anyway; one more tile per row if the widest shifted row needs it
    const char* tiles = getenv("RC_BUILD_F1_TILES");  // "row" / "line": overrides (tests, measurements)
    const bool planes_on_lines = ((int64_t)lay->H * lay->W1 * (int64_t)sizeof(T)) % 128 == 0;
    const bool rows_on_lines = (lay->W1 * (int64_t)sizeof(T)) % 128 == 0 && ((uintptr_t)f1) % 128 == 0;
    const int line_tiles = tiles ? !strcmp(tiles, "line") : (planes_on_lines && !rows_on_lines);
    const int n_tiles_g = line_tiles ? (int)((lay->W1 + 56 + TM - 1) / TM) : n_tiles;
    const int64_t blocks_g = (int64_t)((n_rows + 7) / 8) * 8 * n_tiles_g;
    RC_REQUIRE(blocks_g < (1ll << 31), RC_EINVAL, "grid too large");
    if (dma && f1_once)
      hipLaunchKernelGGL((build16g_kernel<T, true>), dim3((unsigned)blocks_g), dim3(BT16), 2 * G_BUF, st, (const T*)f1,
                         (const T*)f2, (T*)vol, d, scale, n_tiles_g, n_rows, rows_per_xcd, line_tiles);
    else if (dma)
      hipLaunchKernelGGL((build16g_kernel<T, false>), dim3((unsigned)blocks_g), dim3(BT16), 2 * G_BUF, st, (const T*)f1,
                         (const T*)f2, (T*)vol, d, scale, n_tiles_g, n_rows, rows_per_xcd, line_tiles);
    else if (fast && whole) RC_LAUNCH_BUILD16(2);
    else if (fast) RC_LAUNCH_BUILD16(1);
    else RC_LAUNCH_BUILD16(0);
  } else {
    if (fast && whole) RC_LAUNCH_BUILD(2);
    else if (fast) RC_LAUNCH_BUILD(1);
    else RC_LAUNCH_BUILD(0);
  }
#undef RC_LAUNCH_BUILD
#undef RC_LAUNCH_BUILD16
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
  RC_REQUIRE(!lay->full, RC_EINVAL, "rc_build_pyramid takes a forward pyramid (rc_pyramid_layout)");
  const float scale = (float)(1.0 / sqrt((double)D));
  hipStream_t st = (hipStream_t)stream;
  const bool naive = force_naive || D > KMAX;
  const int iD = (int)D;
  const int vol_dtype = lay->dtype;
#define RC_BUILD_CASE(IN, VOL, T, VT)                                                \
  if (in_dtype == IN && vol_dtype == VOL)                                            \
    return naive ? launch_naive<T, VT>(f1, f2, vol, lay, iD, scale, st)              \
                 : launch_build<T, VT>(f1, f2, vol, lay, iD, scale, st);
  RC_BUILD_CASE(RC_F32, RC_F32, float, float)
  RC_BUILD_CASE(RC_F16, RC_F16, rc_half, rc_half)
  RC_BUILD_CASE(RC_BF16, RC_BF16, rc_bf16, rc_bf16)
  RC_BUILD_CASE(RC_F16, RC_F32, rc_half, float)
  RC_BUILD_CASE(RC_BF16, RC_F32, rc_bf16, float)
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

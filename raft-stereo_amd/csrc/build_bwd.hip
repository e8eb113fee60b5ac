// Adjoint of the pyramid build: gradient of the volume -> d fmap1, d fmap2.
//
// Autograd of core/corr.py:119-125 (avg_pool2d chain) and :154-156 (einsum, / sqrt(D)).
//   rc_build_bwd_dense takes the dense gradient of level 0, dC[b,h,w1,w2] in the feature type
//   (rows zero padded to a multiple of 32; rc_lookup_bwd_dense in lookup.hip writes it with the
//   levels already folded and 1/sqrt(D) applied), and runs two MFMA contractions per image row,
//   both operands staged through LDS with coalesced loads:
//        df1[b,:,h,:] = f2[b,:,h,:] (D x W2)  *  dC[b,h]^T (W2 x W1)     (plain LDS image)
//        df2[b,:,h,:] = f1[b,:,h,:] (D x W1)  *  dC[b,h]   (W1 x W2)     (transposing LDS reads)
//   rc_build_bwd takes the float gradient pyramid (group-skewed layout) that rc_lookup_bwd
//   accumulates: unskew_fold reads every 16-byte block once, folds levels 1..L-1 into level 0
//   (a pooled cell spreads 2^-i of its gradient over its 2^i children), scales by 1/sqrt(D) and
//   writes that dense dC into the caller's workspace; then the same two contractions.
#include "rc_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
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
constexpr int TMF = 256; // feature rows (channels) per workgroup

template <typename T> struct gemm_cfg {
  static constexpr bool IS16 = !std::is_same<T, float>::value;
  static constexpr int EPS = 16 / (int)sizeof(T);
  static constexpr int KC = 128 / (int)sizeof(T);  // contraction chunk: a staged row is 128 bytes
  static constexpr int KSTEP = IS16 ? 32 : 16;     // contraction elements per operand piece
  static constexpr int PA = KC + EPS;              // pitch of the feature image [m][k] and of the plain slab
                                                   // image [n][k]: 16 consecutive rows hit different bank quads
  static constexpr int PT = TS + 16;               // pitch of the transposed slab image [k][n]
  static constexpr int FEAT_ELEMS = TMF * PA;
  static constexpr int SLAB_ELEMS = (TS * PA > KC * PT) ? TS * PA : KC * PT;
};

// 16 bytes of a feature row starting at element k, zero past the end of the row
template <typename T>
__device__ __forceinline__ u4v load_feat(const T* __restrict__ row, int k, int K, bool vec) {
  constexpr int EPS = 16 / (int)sizeof(T);
  if (vec && k + EPS <= K) return *reinterpret_cast<const u4v_a4*>(row + k);
  alignas(16) T tmp[EPS];
#pragma unroll
  for (int e = 0; e < EPS; ++e) tmp[e] = (k + e < K) ? row[k + e] : (T)0.0f;
  return *reinterpret_cast<u4v*>(tmp);
}

// Out[b, m, h, n] = sum_k Feat[b, m, h, k] * G(k, n) for one image row (b,h), 64 columns n and 256
// channels m per workgroup (32 per wave).
//   TRANS = false (d fmap1):  n = w1, k = w2, G(k,n) = dC[bh][n][k]  -> LDS image [n][k], plain 16-byte reads
//   TRANS = true  (d fmap2):  n = w2, k = w1, G(k,n) = dC[bh][k][n]  -> LDS image [k][n], transposing reads
// dC rows are ldc (a multiple of 32) long and zero beyond W2.  Both operands are staged through
// LDS in chunks of KC along k with coalesced 16-byte loads (the next chunk is in flight during the
// MFMAs).  The MFMA rows are n and its columns are the channels, so a lane ends up with 4
// consecutive n of one channel: one 8-byte store.
template <typename T, bool TRANS, bool VEC>
__global__ __launch_bounds__(GT) void gemm_rows_kernel(const T* __restrict__ feat, const T* __restrict__ dC,
                                                       T* __restrict__ out, int M, int N, int K, int H, int W1,
                                                       int ldc, int n_slabs, int n_mgroups, int n_rows,
                                                       int rows_per_xcd) {
  constexpr bool vec = VEC;  // feature rows on 4-byte boundaries and at least one 16-byte piece long
  using C = gemm_cfg<T>;
  __shared__ __attribute__((aligned(16))) T featI[C::FEAT_ELEMS];
  __shared__ __attribute__((aligned(16))) T slabI[C::SLAB_ELEMS];
  // XCD-aware mapping: the column slabs (and channel groups) of one image row get block ids
  // congruent mod 8 (one XCD, one L2), since they all re-read that row's features / dC
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int per_row = n_slabs * n_mgroups;
  // (rows_per_xcd > 0: an XCD takes a run of consecutive image rows -- feature rows do not end on 128-byte lines,
  // and the line two rows share, read here and written in the output, then stays in one L2)
  const int bh = rows_per_xcd ? (id & 7) * rows_per_xcd + j / per_row : (j / per_row) * 8 + (id & 7);
  if (bh >= n_rows) return;
  const int sub = j % per_row;
  const int n0 = (sub % n_slabs) * TS;
  const int mb = (sub / n_slabs) * TMF;
  const int b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, li = lane & 15;

  // staging roles: thread -> (row, 16-byte segment) of the 128-byte rows
  const int srow = threadIdx.x >> 3, sseg = threadIdx.x & 7;
  const T* dCrow = dC + (int64_t)bh * W1 * ldc;

  // A chunk's five 16-byte loads per thread are buffer loads at an offset that is either the piece's or past the
  // buffer's end (which reads zeros and touches no memory): no load sits under a bounds test.  Written with
  // `if (m < M) load`, every load was a branch of its own and the compiler waited for each one where the
  // branches joined -- five memory latencies in a row per chunk, and none of them behind the MFMAs.  A feature
  // row's last, partial piece (K not a multiple of the piece) is read as the 16 bytes that END at the row's end
  // and shifted into place when the chunk is stored to LDS (the shift is the same for the whole launch).
  const int64_t img_elems = (int64_t)M * H * K;
  const uint32_t feat_bytes = (uint32_t)(img_elems * (int64_t)sizeof(T));
  const uint32_t dc_bytes = (uint32_t)((int64_t)W1 * ldc * (int64_t)sizeof(T));
  const __amdgpu_buffer_rsrc_t featR =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(feat + (int64_t)b * img_elems), 0, (int)feat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dCrow), 0, (int)dc_bytes, 0x00020000);
  const int tail_shift_bits = (C::EPS - K % C::EPS) * 8 * (int)sizeof(T);  // used when K % EPS != 0 only
  u4v rf[4], rs;
  auto load_chunk = [&](int kc0) {
    const int kf = kc0 + sseg * C::EPS;
    if (vec) {  // launch-uniform: rows start on 4-byte boundaries
      const int kl = kf + C::EPS <= K ? kf : K - C::EPS;  // the partial piece: the row's last 16 bytes instead
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mb + srow + 64 * j;
        const uint32_t off = (uint32_t)((m * H + h) * K + kl) * (uint32_t)sizeof(T);
        rf[j] = __builtin_amdgcn_raw_buffer_load_b128(featR, (int)((m < M && kf < K) ? off : feat_bytes), 0, 0);
      }
    } else {  // element by element
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mb + srow + 64 * j;
        rf[j] = (u4v)(0u);
        if (m < M) rf[j] = load_feat<T>(feat + (((int64_t)b * M + m) * H + h) * K, kf, K, false);
      }
    }
    if constexpr (!TRANS) {
      // image [n][k]: row n0 + srow of dC, columns kc0 .. kc0 + KC
      const int n = n0 + srow, k = kc0 + sseg * C::EPS;
      const uint32_t off = (uint32_t)(n * ldc + k) * (uint32_t)sizeof(T);
      rs = __builtin_amdgcn_raw_buffer_load_b128(dcR, (int)((n < N && k < ldc) ? off : dc_bytes), 0, 0);
    } else {
      // image [k][n]: row kc0 + r of dC (a w1), columns n0 .. n0 + 64 (a 64-column row is TS / EPS segments)
      constexpr int SEGS = TS / C::EPS;
      const int r = threadIdx.x / SEGS, sg = threadIdx.x - r * SEGS;
      const int k = kc0 + r, n = n0 + sg * C::EPS;
      const uint32_t off = (uint32_t)(k * ldc + n) * (uint32_t)sizeof(T);
      rs = __builtin_amdgcn_raw_buffer_load_b128(dcR, (int)((k < K && n < ldc) ? off : dc_bytes), 0, 0);
    }
  };
  // the partial piece of chunk kc0, loaded as the row's last 16 bytes: move its first K - kf elements to the front
  auto fix_tail = [&](int kc0) {
    const int kf = kc0 + sseg * C::EPS;
    const bool tail = vec && kf < K && kf + C::EPS > K;
    if (!__any(tail)) return;  // wave-uniform; no loads inside
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t lo = (uint64_t)rf[j][0] | ((uint64_t)rf[j][1] << 32), hi = (uint64_t)rf[j][2] | ((uint64_t)rf[j][3] << 32);
      const int sb = tail_shift_bits;  // 1 .. 127, block-uniform
      const uint64_t nlo = sb >= 64 ? hi >> (sb - 64) : (lo >> sb) | (hi << (64 - sb));
      const uint64_t nhi = sb >= 64 ? 0ull : hi >> sb;
      const u4v sh = {(uint32_t)nlo, (uint32_t)(nlo >> 32), (uint32_t)nhi, (uint32_t)(nhi >> 32)};
      rf[j] = tail ? sh : rf[j];
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<u4v*>(featI + (srow + 64 * j) * C::PA + sseg * C::EPS) = rf[j];
    if constexpr (!TRANS) {
      *reinterpret_cast<u4v*>(slabI + srow * C::PA + sseg * C::EPS) = rs;
    } else {
      constexpr int SEGS = TS / C::EPS;
      const int r = threadIdx.x / SEGS, sg = threadIdx.x - r * SEGS;
      *reinterpret_cast<u4v*>(slabI + r * C::PT + sg * C::EPS) = rs;
    }
  };

  f4v acc[4][2];
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    acc[y][0] = (f4v)(0.f);
    acc[y][1] = (f4v)(0.f);
  }

  load_chunk(0);
  for (int kc0 = 0; kc0 < K; kc0 += C::KC) {
    __syncthreads();  // previous chunk's readers are done
    fix_tail(kc0);
    store_chunk();
    __syncthreads();
    if (kc0 + C::KC < K) load_chunk(kc0 + C::KC);

    const T* frow0 = featI + (wave * 32 + li) * C::PA;  // this lane's channel rows: + 16 x rows
#pragma unroll
    for (int ks = 0; ks < C::KC / C::KSTEP; ++ks) {
      const int kk = ks * C::KSTEP;  // offset inside the chunk
      if (kc0 + kk >= K) break;      // wave-uniform
      if constexpr (C::IS16) {
        // plain image: lane (g, li) holds k = kk + 8 g .. + 7 of channel row li and of slab column li;
        // transposed image: the transposing read hands group g the k = kk + 4 g + {0..3} and
        // kk + 16 + 4 g + {0..3}, so the channel piece is gathered to match
        s8v fa[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const T* fr = frow0 + 16 * x * C::PA + kk;
          if constexpr (!TRANS) {
            fa[x] = *reinterpret_cast<const s8v*>(fr + 8 * g);
          } else {
            const s4v lo = *reinterpret_cast<const s4v*>(fr + 4 * g);
            const s4v hi = *reinterpret_cast<const s4v*>(fr + 16 + 4 * g);
            fa[x] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          s8v gb;
          if constexpr (!TRANS) {
            gb = *reinterpret_cast<const s8v*>(slabI + (16 * y + li) * C::PA + kk + 8 * g);
          } else {
            typedef s4v __attribute__((address_space(3))) * lds_s4p;
            const short* base = reinterpret_cast<const short*>(slabI);
            const int e = (kk + 4 * g + (li >> 2)) * C::PT + 16 * y + 4 * (li & 3);
            const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(base + e));
            const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(base + e + 16 * C::PT));
            gb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int x = 0; x < 2; ++x) acc[y][x] = mfma16<T>(gb, fa[x], acc[y][x]);
        }
      } else {
        // f32: 16 k per step as 4 exact-f32 MFMAs; lane (g, li) supplies k = kk + 4 g + e in sub-step e
        const float* lf = reinterpret_cast<const float*>(slabI);
        f4v fa[2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
          fa[x] = *reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(frow0) + 16 * x * C::PA + kk + 4 * g);
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          f4v gb;
          if constexpr (!TRANS) {
            gb = *reinterpret_cast<const f4v*>(lf + (16 * y + li) * C::PA + kk + 4 * g);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) gb[e] = lf[(kk + 4 * g + e) * C::PT + 16 * y + li];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int x = 0; x < 2; ++x)
              acc[y][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(gb[e], fa[x][e], acc[y][x], 0, 0, 0);
        }
      }
    }
  }

  // lane (g, li): columns n0 + 16 y + 4 g + {0..3} of channel mb + 32 wave + 16 x + li
  if constexpr (C::IS16) {
    if ((N & 1) == 0) {  // rows of the result start on dword boundaries
      // 16-bit results go through LDS (the feature image is free now) so that they leave as 16-byte
      // pieces, 8 lanes per 128-byte run of a channel row, instead of 8-byte pieces of 16 rows at once
      static_assert(TS + 8 <= C::PA && TMF * C::PA <= C::FEAT_ELEMS, "the output tile reuses the feature image");
      typedef T t4v __attribute__((ext_vector_type(4)));
      __syncthreads();  // every wave has finished reading featI
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          *reinterpret_cast<t4v*>(featI + (wave * 32 + 16 * x + li) * C::PA + 16 * y + 4 * g) =
              __builtin_convertvector(acc[y][x], t4v);
      __syncthreads();
      const int n = n0 + sseg * C::EPS;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = srow + 64 * j, m = mb + r;
        if (m >= M || n >= N) continue;
        T* dst = out + (((int64_t)b * M + m) * H + h) * N + n;
        const T* src = featI + r * C::PA + sseg * C::EPS;
        if (n + C::EPS <= N) {
          *reinterpret_cast<u4v_a4*>(dst) = *reinterpret_cast<const u4v*>(src);
        } else {
          for (int e = 0; n + e < N; ++e) dst[e] = src[e];
        }
      }
      return;
    }
  }
  const bool vec_out = (N % 4 == 0);
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int m = mb + wave * 32 + 16 * x + li;
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

// the two contractions from a dense dC [BH][W1][W2p]
template <typename T>
int run_gemms(const void* dCv, const rc_layout* lay, const void* f1, const void* f2, void* df1, void* df2, int D,
              hipStream_t st) {
  const int W1 = (int)lay->W1, W2 = (int)lay->W2, H = (int)lay->H;
  const int W2p = (int)round_up(W2, 32);
  const int64_t BH = lay->B * lay->H;
  const T* dC = reinterpret_cast<const T*>(dCv);
  const int my = (D + TMF - 1) / TMF;
  const int64_t rows8 = (BH + 7) / 8 * 8;
  // 16-byte pieces: rows on 4-byte boundaries and at least one piece long (a partial last piece is read as the
  // row's last 16 bytes)
  constexpr int EPS = 16 / (int)sizeof(T);
  const bool v2 = (W2 * sizeof(T)) % 4 == 0 && (uintptr_t)f2 % 4 == 0 && W2 >= EPS;
  const bool v1 = (W1 * sizeof(T)) % 4 == 0 && (uintptr_t)f1 % 4 == 0 && W1 >= EPS;
  const int ns1 = (W1 + TS - 1) / TS, ns2 = (W2 + TS - 1) / TS;
  RC_REQUIRE(rows8 * (ns1 > ns2 ? ns1 : ns2) * my < (1ll << 31), RC_EINVAL, "grid too large");
  // the kernel addresses one image's feature maps and one row's dC through 32-bit buffer offsets
  RC_REQUIRE((int64_t)D * H * (W1 > W2 ? W1 : W2) * (int64_t)sizeof(T) < (1ll << 31) &&
                 (int64_t)W1 * W2p * (int64_t)sizeof(T) < (1ll << 31),
             RC_EUNSUPPORTED, "one image's feature maps (or one row's dense gradient) exceed 2 GiB");
  const char* order = getenv("RC_BUILD_ROW_ORDER");  // as in rc_build_pyramid
  const int rows_per_xcd = order && !strcmp(order, "interleaved") ? 0 : (int)(rows8 / 8);
  // d fmap1 = f2 . dC^T : n = w1, k = w2
#define RC_GEMM_ROWS(TRANS, VEC, feat, out, N, K, ns)                                                             \
  hipLaunchKernelGGL((gemm_rows_kernel<T, TRANS, VEC>), dim3((unsigned)(rows8 * (ns) * my), 1, 1), dim3(GT), 0, st, \
                     (const T*)(feat), dC, (T*)(out), D, N, K, H, W1, W2p, ns, my, (int)BH, rows_per_xcd)
  if (v2) RC_GEMM_ROWS(false, true, f2, df1, W1, W2, ns1);
  else RC_GEMM_ROWS(false, false, f2, df1, W1, W2, ns1);
  int rc = rc_check_launch("rc_build_bwd(df1)");
  if (rc) return rc;
  // d fmap2 = f1 . dC : n = w2, k = w1
  if (v1) RC_GEMM_ROWS(true, true, f1, df2, W2, W1, ns2);
  else RC_GEMM_ROWS(true, false, f1, df2, W2, W1, ns2);
#undef RC_GEMM_ROWS
  return rc_check_launch("rc_build_bwd(df2)");
}

template <typename T>
int run_bwd(const float* gvol, const rc_layout* lay, const void* f1, const void* f2, void* df1, void* df2,
            int D, void* work, hipStream_t st) {
  const rc_dev_layout d = rc_to_dev(lay);
  const int W1 = d.W1, W2 = d.W2;
  const int W2p = (int)round_up(W2, 32);
  const int64_t BH = lay->B * lay->H;
  T* dC = reinterpret_cast<T*>(work);
  const float scale = (float)(1.0 / sqrt((double)D));
  RC_REQUIRE(BH < 65536, RC_EINVAL, "B*H=%lld exceeds the grid z range; split the batch", (long long)BH);
  hipLaunchKernelGGL((unskew_fold_kernel<T>), dim3((W1 + UT - 1) / UT, (W2p + UT - 1) / UT, (unsigned)BH),
                     dim3(256), 0, st, gvol, d, dC, W2p, scale);
  int rc = rc_check_launch("rc_build_bwd(unskew)");
  if (rc) return rc;
  return run_gemms<T>(dC, lay, f1, f2, df1, df2, D, st);
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

extern "C" int rc_build_bwd_dense(const void* dense_grad, const rc_layout* lay, const void* f1, const void* f2,
                                  void* df1, void* df2, int64_t D, int in_dtype, void* stream) {
  RC_REQUIRE(dense_grad && lay && f1 && f2 && df1 && df2, RC_EINVAL, "null pointer argument");
  int rc = rc_validate_layout(lay);
  if (rc) return rc;
  RC_REQUIRE(D >= 1 && D < (1 << 20), RC_EINVAL, "bad feature depth D=%lld", (long long)D);
  RC_REQUIRE((uintptr_t)dense_grad % 16 == 0, RC_EINVAL, "dense gradient must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (in_dtype) {
    case RC_F32: return run_gemms<float>(dense_grad, lay, f1, f2, df1, df2, (int)D, st);
    case RC_F16: return run_gemms<rc_half>(dense_grad, lay, f1, f2, df1, df2, (int)D, st);
    case RC_BF16: return run_gemms<rc_bf16>(dense_grad, lay, f1, f2, df1, df2, (int)D, st);
  }
  rc_set_error("unknown in_dtype %d", in_dtype);
  return RC_EUNSUPPORTED;
}

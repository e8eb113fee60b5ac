// On-the-fly ("alt") correlation lookup: the volume is never materialised.
//
// Replaces core/corr.py:64-107 (PytorchAlternateCorrBlock1D) of the reference, which runs 36
// full-feature grid_samples per GRU iteration and re-pools fmap2 on every call.
//
// rc_alt_prepare (once per image pair) writes the working copies f1t and the pooled f2 pyramid
// f2t_i in MFMA-fragment order: positions in blocks of 16, channels in chunks of KCH (32 for
// 16-bit, 16 for f32); chunk kc of position block pb is the 64 x 16-byte operand of one MFMA,
// lane (g, li) = channels kc*KCH + [g*KCH/4, (g+1)*KCH/4) of position 16*pb + li:
//   element offset = ((row*nblk + pb)*nkc + kc)*64*EL + (g*16 + li)*EL + e,  EL = 16 B / elemsize.
// An operand load is then one contiguous 1 KiB run per wave instruction (8 cache lines instead of 64).
// rc_alt_fwd (every iteration): a workgroup owns 16 consecutive pixels of one image row, one wave
// per pyramid level.  The wave takes the union of the pixels' tap windows at its level, and for each 16-wide block of right-image
// positions in it forms the 16x16 product tile  f2t_i[block] . f1t[pixels]^T  on the matrix
// cores (K = Dp); MFMA rows are w2, so a lane holds 4 consecutive w2 of one pixel; it drops those
// inside the pixel's own window into LDS, from where the pixel's 2r+1 taps are interpolated.
// With locally smooth disparity a level needs 1-3 blocks; arbitrary coords stay correct, only slower.
#include "rc_common.h"

#include <math.h>
#include <string.h>
#include <type_traits>

namespace {

typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int DMAX = 256;

template <typename T> __device__ __forceinline__ f4v mfma16(s8v a, s8v b, f4v c);
template <> __device__ __forceinline__ f4v mfma16<rc_half>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f4v mfma16<rc_bf16>(s8v a, s8v b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
}

struct alt_dev_layout {
  int B, H, W1, W2, D, Dp, L;
  int f1_once;  // host side only: picks the kernel that fetches the pixels' own features non-temporally
  int width[RC_MAX_LEVELS];
  int64_t f2_offset[RC_MAX_LEVELS];
};

alt_dev_layout to_dev(const rc_alt_layout* a) {
  alt_dev_layout d;
  d.B = (int)a->B; d.H = (int)a->H; d.W1 = (int)a->W1; d.W2 = (int)a->W2;
  d.D = (int)a->D; d.Dp = (int)a->Dp; d.L = a->L;
  for (int i = 0; i < RC_MAX_LEVELS; ++i) { d.width[i] = a->width[i]; d.f2_offset[i] = a->f2_offset[i]; }
  d.f1_once = 0;
  return d;
}

// NCHW [B,D,H,W] -> fragment order (see the file header) of levels 0..nlev-1 in one pass over the
// source: a workgroup reads a tile of 128 positions x 64 channels, pools it 2:1 per level in LDS
// (pairwise tree in fp32, the order of the avg_pool2d chain, corr.py:104) and writes every level's
// blocks.  Zero beyond a level's width (blocks are padded to 16 positions) and beyond D (padded to Dp).
struct frag_levels {
  int nlev;
  int width[RC_MAX_LEVELS];
  int64_t offset[RC_MAX_LEVELS];  // < 0: the level is only a step of the pooling chain and is not written
};

template <typename T>
__global__ __launch_bounds__(256) void to_fragments_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                           int D, int Dp, int H, int W, frag_levels lv) {
  constexpr int EL = 16 / (int)sizeof(T), KCH = 4 * EL;
  constexpr int KPT = 64 / KCH;      // channel chunks per tile
  constexpr int PITCH = 128 + 64 + 32 + 16 + 1;
  __shared__ float tile[64][PITCH];  // per channel: level 0 at [0,128), level 1 at [128,192), ...
  const int nkc = Dp / KCH;
  const int w0 = blockIdx.x * 128, d0 = blockIdx.y * 64;
  const int bh = blockIdx.z, b = bh / H, h = bh - b * H;
  {
    // 16 loads in flight per thread; addresses are clamped instead of predicated (a predicated load
    // would be waited for on its own) and padding channels are zeroed afterwards
    const int wc = threadIdx.x & 127, dr = threadIdx.x >> 7;
    const int w = min(w0 + wc, W - 1);
    const T* col = src + ((int64_t)b * D * H + h) * W + w;
    const int64_t dstride = (int64_t)H * W;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      T v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = col[min(d0 + pass * 32 + 2 * u + dr, D - 1) * dstride];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int dd = pass * 32 + 2 * u + dr;
        tile[dd][wc] = d0 + dd < D ? rc_to_float(v[u]) : 0.f;
      }
    }
  }
  int from = 0;
  for (int i = 1; i < lv.nlev; ++i) {
    __syncthreads();
    const int n = 128 >> i, to = from + (256 >> i);
    for (int e = threadIdx.x; e < 64 * n; e += 256) {
      const int dd = e / n, q = e - dd * n;
      tile[dd][to + q] = 0.5f * (tile[dd][from + 2 * q] + tile[dd][from + 2 * q + 1]);
    }
    from = to;
  }
  __syncthreads();
  // one 16-byte fragment piece per thread and step; consecutive threads write consecutive pieces
  int col0 = 0;
  for (int i = 0; i < lv.nlev; ++i) {
    const int Wl = lv.width[i], nblk = (Wl + 15) >> 4;
    const int blocks_here = 8 >> i;  // 16-position blocks of this level inside the tile
    T* base = dst + lv.offset[i];
    for (int f = threadIdx.x; lv.offset[i] >= 0 && f < blocks_here * KPT * 64; f += 256) {
      const int li = f & 15, g = (f >> 4) & 3, kcl = (f >> 6) % KPT, blk = (f >> 6) / KPT;
      const int pb = (w0 >> (4 + i)) + blk, kc = d0 / KCH + kcl;
      if (pb >= nblk || kc >= nkc) continue;
      const bool inside = pb * 16 + li < Wl;
      __attribute__((aligned(16))) T v[EL];
#pragma unroll
      for (int e = 0; e < EL; ++e)
        v[e] = rc_from_float<T>(inside ? tile[kcl * KCH + g * EL + e][col0 + blk * 16 + li] : 0.f);
      T* q = base + ((((int64_t)bh * nblk + pb) * nkc + kc) * 64 + g * 16 + li) * EL;
      *reinterpret_cast<uint4*>(q) = *reinterpret_cast<const uint4*>(v);
    }
    col0 += 128 >> i;
  }
}

// The same working copies for 16-bit features whose rows are whole 16-byte segments (W % 8 == 0, 16-byte aligned):
// 16-byte loads (a wave fetches 4 channels x 256 contiguous bytes per instruction instead of 128 bytes of 2-byte
// elements), a 16-bit tile in LDS, and gfx950's transposing LDS read (ds_read_b64_tr_b16: a group of 16 lanes reads
// 4 channel rows x 16 positions and every lane receives the 4 channels of its own position), so a fragment piece
// costs 2 LDS reads instead of 8 and no conversion.  Level 2 is pooled in registers, in fp32 and in the order of
// the element kernel's tree (means of pairs, then means of those, rounded once): the same bits.
constexpr int TF_P0 = 256 + 32;  // bytes per channel row of the level-0 tile (128 positions); + 32: the 4 rows of a
constexpr int TF_P2 = 64 + 32;   // transposing read start 8 banks apart.  Level 2: 32 positions.

template <typename T>
__global__ __launch_bounds__(256) void to_fragments16_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                             int D, int Dp, int H, int W, frag_levels lv,
                                                             int n_rows, int rows_per_xcd) {
  static_assert(sizeof(T) == 2, "16-bit features");
  typedef short s4v __attribute__((ext_vector_type(4)));
  typedef s4v __attribute__((address_space(3))) * lds_s4p;
  __shared__ __attribute__((aligned(16))) unsigned char t0[64 * TF_P0];
  __shared__ __attribute__((aligned(16))) unsigned char t2[64 * TF_P2];
  const int nkc = Dp / 32;
  // one-dimensional grid, XCD-aware: an XCD works through a run of consecutive image rows, and through the tiles
  // of one row one after the other -- a tile's 256-byte pieces do not start on 128-byte lines, and the lines it
  // shares with its neighbours (and the row with the next row) then meet in one L2
  const int xt = (W + 127) >> 7, dt = (Dp + 63) >> 6, per_row = xt * dt;
  const int slot = blockIdx.x >> 3, rl = slot / per_row, t = slot - rl * per_row;
  const int bh = (blockIdx.x & 7) * rows_per_xcd + rl;
  if (bh >= n_rows) return;
  const int w0 = (t % xt) * 128, d0 = (t / xt) * 64;
  const int b = bh / H, h = bh - b * H;
  const bool pooled = lv.nlev >= 3;  // level 2 is wanted (level 1 is only a step on the way)
  {
    const int seg = threadIdx.x & 15, cr = threadIdx.x >> 4;  // 8 positions of one channel; 16 channels per pass
    const int w = w0 + seg * 8;
    const bool in_w = w < W;  // whole segments: W % 8 == 0
    const T* col = src + ((int64_t)b * D * H + h) * W + (in_w ? w : 0);
    const int64_t dstride = (int64_t)H * W;
    uint4 v[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass)  // clamped, not predicated; zeroed below
      v[pass] = *reinterpret_cast<const uint4*>(col + (int64_t)min(d0 + pass * 16 + cr, D - 1) * dstride);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int dd = pass * 16 + cr;
      if (!in_w || d0 + dd >= D) v[pass] = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(t0 + dd * TF_P0 + seg * 16) = v[pass];
      if (pooled) {
        const uint32_t x[4] = {v[pass].x, v[pass].y, v[pass].z, v[pass].w};
        float l1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          T e[2];
          __builtin_memcpy(e, &x[j], 4);
          l1[j] = 0.5f * (rc_to_float(e[0]) + rc_to_float(e[1]));
        }
        const T l2[2] = {rc_from_float<T>(0.5f * (l1[0] + l1[1])), rc_from_float<T>(0.5f * (l1[2] + l1[3]))};
        uint32_t packed;
        __builtin_memcpy(&packed, l2, 4);
        *reinterpret_cast<uint32_t*>(t2 + dd * TF_P2 + seg * 4) = packed;
      }
    }
  }
  __syncthreads();
  // one 16-byte fragment piece per thread and step: lane (g, li) of a wave = channels 8 g .. 8 g + 7 of position li
  // of one block; (block, channel chunk) are wave-uniform, so the bounds tests below are, too
  const int li = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
  const int lane_off = (8 * g + (li >> 2));  // channel row this lane addresses in a transposing read
  auto piece = [&](const unsigned char* tile, int pitch, int kcl, int blk) {
    const unsigned char* a = tile + (kcl * 32 + lane_off) * pitch + blk * 32 + (li & 3) * 8;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)a);
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(a + 4 * pitch));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  if (lv.offset[0] >= 0) {
    const int nblk = (lv.width[0] + 15) >> 4;
    T* base = dst + lv.offset[0];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int f = (int)threadIdx.x + 256 * s;
      const int kcl = (f >> 6) & 1, blk = f >> 7;
      const int pb = (w0 >> 4) + blk, kc = d0 / 32 + kcl;
      const s8v v = piece(t0, TF_P0, kcl, blk);
      if (pb < nblk && kc < nkc)
        *reinterpret_cast<s8v*>(base + ((((int64_t)bh * nblk + pb) * nkc + kc) * 64 + g * 16 + li) * 8) = v;
    }
  }
  if (pooled && lv.offset[2] >= 0) {
    const int nblk = (lv.width[2] + 15) >> 4;
    T* base = dst + lv.offset[2];
    const int f = (int)threadIdx.x;
    const int kcl = (f >> 6) & 1, blk = f >> 7;
    const int pb = (w0 >> 6) + blk, kc = d0 / 32 + kcl;
    const s8v v = piece(t2, TF_P2, kcl, blk);
    if (pb < nblk && kc < nkc)
      *reinterpret_cast<s8v*>(base + ((((int64_t)bh * nblk + pb) * nkc + kc) * 64 + g * 16 + li) * 8) = v;
  }
}

// FULL: Dp == DMAX, every operand load unconditional (the real model's D = 256).
// A wave serves a PAIR of levels (i, i + 1), i even, of 16 pixels: pooling is linear, so the correlation with the
// pooled right image is the mean of two correlations one level down,
//     C_{i+1}[w1][j] = (C_i[w1][2j] + C_i[w1][2j+1]) / 2,
// and because floor(x / 2^i) = 2 floor(x / 2^(i+1)) + bit, level i's window lies inside the run of 2 NS level-i
// positions under level (i+1)'s window (the lookup of the materialised pyramid reads its levels the same way,
// window.h).  The copies of f2 at the odd levels are never read: a third fewer operand bytes per lookup, and the
// odd levels are means of fp32 sums instead of sums over features pooled and rounded to 16 bits -- closer to
// the reference, which pools in fp32 (core/corr.py:104).
#ifndef RC_LAB_ALT_POLICY  // lab: 1 = f1 fragments non-temporal, 2 = outputs non-temporal, 3 = both
#define RC_LAB_ALT_POLICY 0
#endif
#if RC_LAB_ALT_POLICY & 2
#define ALT_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define ALT_STORE(ptr, val) (*(ptr) = (val))
#endif

template <typename T, int R, int MODE, bool FULL, bool F1_ONCE>
__global__ __launch_bounds__(256) void alt_fwd_kernel(const T* __restrict__ f1t, const T* __restrict__ f2t,
                                                      alt_dev_layout lay, const float* __restrict__ coords,
                                                      int coords_channels, float* __restrict__ out,
                                                      float scale, int blocks_per_row, int n_rows) {
  constexpr bool IS16 = !std::is_same<T, float>::value;
  constexpr int NT = 2 * R + 1;
  constexpr int EXT = MODE == RC_COORD_GRIDSAMPLE ? 1 : 0;  // grid_sample's fp32 round trip can move a tap by one
  constexpr int NS = NT + 1 + 2 * EXT;                      // correlation values a pixel's taps can touch
  constexpr int NR = 2 * NS;                                // ... of level i under the window of level i + 1
  constexpr int PITCH = NR | 1;
  constexpr int EL = 16 / (int)sizeof(T);
  constexpr int KCH = 4 * EL;                               // K elements consumed per operand load
  constexpr int NKC = DMAX / KCH;
  // a pixel's run of correlation values at the wave's even level: slot j <-> w2 = rs + j
  __shared__ float win[4][16][PITCH];

  // (wave through readfirstlane: the level-indexed kernel arguments below are then scalar loads; indexed by a
  // value the compiler takes for per-lane they were vector loads issued after the feature loads, and the first
  // block of f2 waited for all of them plus one more round trip for its offset)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  // A workgroup owns 32 consecutive pixels of one image row: wave = (block of 16 pixels, pair of levels).  A run
  // needs 2-4 operand blocks, the two waves of a pixel block share its operand through the L1.  Workgroup ids
  // that agree mod 8 run on the same XCD: they get the same rows, so a row of fmap2 goes through one L2 only.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / blocks_per_row) * 8 + xcd;
  const int pb1 = (slot % blocks_per_row) * 2 + (wave >> 1);  // the wave's block of 16 pixels
  const int i = 2 * (wave & 1);                               // its even level
  const int w1 = pb1 * 16 + li;
  if (bh >= n_rows || pb1 * 16 >= lay.W1 || i >= lay.L) return;  // wave-uniform
  const bool has_hi = i + 1 < lay.L;                             // wave-uniform
  const int b = bh / lay.H, h = bh - b * lay.H;
  const bool live = w1 < lay.W1;
  const int w1c = live ? w1 : lay.W1 - 1;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const int nkc = FULL ? NKC : lay.Dp / KCH;
  const int64_t blk_elems = (int64_t)nkc * 64 * EL;  // one block of 16 positions, all channels

  // the coordinate first: loads return in order, and the first block of f2 can be asked for as soon as x is here,
  // while the pixels' own features are still on their way
  const float x = coords[((int64_t)b * coords_channels) * HW + (int64_t)h * lay.W1 + w1c];
  __builtin_amdgcn_sched_barrier(0);
  // the pixels' feature vectors as MFMA B operands (columns = pixels)
  uint4 bfrag[NKC];
  {
    const T* p = f1t + ((int64_t)bh * ((lay.W1 + 15) >> 4) + pb1) * blk_elems + lane * EL;
    // Both operands together (Middlebury frame: 194 + 243 MB) do not fit the 256 MiB Infinity Cache, and every
    // lookup reads all of both: with plain loads each evicts the other and everything comes from HBM every time.
    // f1's fragments have no reader outside their workgroup, so when the two do not fit (F1_ONCE, decided by
    // the launcher) they are fetched non-temporally and f2 stays in the cache from one lookup to the next:
    // 100.9 -> 94.1 us.  (Outputs non-temporal as well: 92.5, left plain for their consumer; f2's coarse copy
    // non-temporal: nothing.)
    typedef uint32_t alt_u4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if constexpr (F1_ONCE) {  // (a template argument: under a run-time flag the compiler folds the two sets of
                                // loads into one and drops the policy they do not share)
        const alt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const alt_u4*>(p + kc * 64 * EL));
        bfrag[kc] = make_uint4(v[0], v[1], v[2], v[3]);
      } else if constexpr (FULL) {
        bfrag[kc] = *reinterpret_cast<const uint4*>(p + kc * 64 * EL);
      } else {
        bfrag[kc] = kc < nkc ? *reinterpret_cast<const uint4*>(p + kc * 64 * EL) : make_uint4(0, 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  const int Wlo = lay.width[i], Whi = has_hi ? lay.width[i + 1] : 0;
  const float x_lo = x / (float)(1 << i), x_hi = x / (float)(2 << i);
  const float f_lo = floorf(x_lo), f_hi = floorf(x_hi);
  const int base_lo = rc_safe_int(f_lo) - R - EXT, base_hi = rc_safe_int(f_hi) - R - EXT;
  const int rs = has_hi ? 2 * base_hi : base_lo;  // first position of the run
  const int nr = has_hi ? NR : NS;
  // where level i's window starts inside the run: R + EXT or one more; coordinates so large that the integer
  // conversion saturates break that relation, and then every tap is outside the image anyway
  const int off_lo = min(max(base_lo - rs, 0), nr - NS);
  float* mywin = &win[wave][li][0];

  // union of the 16 pixels' runs (all 64 lanes take part; the 4 groups hold copies)
  int lo = live ? rs : (1 << 30), hi = live ? rs + nr - 1 : -(1 << 30);
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  lo = max(lo, 0);
  hi = min(hi, Wlo - 1);

  const T* f2l = f2t + lay.f2_offset[i] + ((int64_t)bh * ((Wlo + 15) >> 4)) * blk_elems + lane * EL;
  // the next block of positions is requested before the current one is multiplied (the trip count is 2-4 and
  // not known to the compiler, which otherwise exposes one memory latency per block)
  const int cb_end = __builtin_amdgcn_readfirstlane(hi >> 4);
  int cb = __builtin_amdgcn_readfirstlane(lo >> 4);
  uint4 av[NKC], nx[NKC];
  auto fetch = [&](uint4 (&dstv)[NKC], int blk) {
    const T* ap = f2l + blk * blk_elems;
#if RC_LAB_ALT_POLICY & 4  // lab: the coarse pair's operand (a fifth of f2's bytes) non-temporal
    if (i == 2) {
      typedef uint32_t alt_u4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc)
        if (FULL || kc < nkc) {
          const alt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const alt_u4*>(ap + kc * 64 * EL));
          dstv[kc] = make_uint4(v[0], v[1], v[2], v[3]);
        }
      return;
    }
#endif
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
      if (FULL || kc < nkc) dstv[kc] = *reinterpret_cast<const uint4*>(ap + kc * 64 * EL);
  };
  // 16 right-image positions x 16 pixels on the matrix cores
  auto multiply = [&](const uint4 (&a)[NKC], int blk) {
    f4v acc = (f4v)(0.f);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (FULL || kc < nkc) {
        if constexpr (IS16) {
          acc = mfma16<T>(__builtin_bit_cast(s8v, a[kc]), __builtin_bit_cast(s8v, bfrag[kc]), acc);
        } else {
          const f4v a4 = __builtin_bit_cast(f4v, a[kc]), b4 = __builtin_bit_cast(f4v, bfrag[kc]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[e], acc, 0, 0, 0);
        }
      }
    }
    // lane holds C[w2 = blk*16 + 4g + r][this pixel]: keep what falls into the pixel's own run
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w2 = blk * 16 + 4 * g + r;
      const int j = w2 - rs;
      if ((unsigned)j < (unsigned)nr && w2 < Wlo) mywin[j] = acc[r];
    }
  };
  if constexpr (!IS16) {  // f32 operands are twice the registers: one block at a time
    for (; cb <= cb_end; ++cb) {
      fetch(av, cb);
      multiply(av, cb);
    }
  } else if (cb <= cb_end) {  // (nothing to do when lo > hi)
    fetch(av, cb);
    while (true) {  // two register sets take turns: no copies
      if (cb < cb_end) fetch(nx, cb + 1);
      multiply(av, cb);
      if (++cb > cb_end) break;
      if (cb < cb_end) fetch(av, cb + 1);
      multiply(nx, cb);
      if (++cb > cb_end) break;
    }
  }
  // the four lane groups of a pixel exchange through LDS: one wave, in-order LDS, no workgroup barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // lane group g interpolates taps g, g+4, g+8, ... of both levels; a tap position outside the image counts 0
  float* orow = out + ((int64_t)b * lay.L * NT) * HW + (int64_t)h * lay.W1 + w1c;
#pragma unroll
  for (int k0 = 0; k0 < NT; k0 += 4) {
    const int k = k0 + g;
    {  // level i: slots off_lo ..
      int t0;
      float wa, wb;
      if (MODE == RC_COORD_EXACT) {
        t0 = base_lo + k;
        wb = x_lo - f_lo;
        wa = 1.0f - wb;
      } else {
        rc_gridsample_coords(x_lo, k - R, (float)(Wlo - 1), &t0, &wa, &wb);
      }
      const int j0 = off_lo + min(max(t0 - base_lo, 0), NS - 2);
      const float v0 = mywin[j0], v1 = mywin[j0 + 1];
      const float s = ((unsigned)t0 < (unsigned)Wlo ? wa * v0 : 0.f) + ((unsigned)(t0 + 1) < (unsigned)Wlo ? wb * v1 : 0.f);
      if (live && k < NT) ALT_STORE(orow + (int64_t)(i * NT + k) * HW, s * scale);
    }
    if (has_hi) {  // level i + 1: slot pairs (2j, 2j + 1) of the run
      int t0;
      float wa, wb;
      if (MODE == RC_COORD_EXACT) {
        t0 = base_hi + k;
        wb = x_hi - f_hi;
        wa = 1.0f - wb;
      } else {
        rc_gridsample_coords(x_hi, k - R, (float)(Whi - 1), &t0, &wa, &wb);
      }
      const int j0 = 2 * min(max(t0 - base_hi, 0), NS - 2);
      const float v0 = 0.5f * (mywin[j0] + mywin[j0 + 1]), v1 = 0.5f * (mywin[j0 + 2] + mywin[j0 + 3]);
      const float s = ((unsigned)t0 < (unsigned)Whi ? wa * v0 : 0.f) + ((unsigned)(t0 + 1) < (unsigned)Whi ? wb * v1 : 0.f);
      if (live && k < NT) ALT_STORE(orow + (int64_t)((i + 1) * NT + k) * HW, s * scale);
    }
  }
}

template <typename T, int R>
int launch_alt(const void* f1t, const void* f2t, const alt_dev_layout& d, const float* coords, int cc,
               float* out, int mode, hipStream_t st) {
  const int bpr = (d.W1 + 31) / 32;  // workgroups per image row: 32 pixels each
  const int n_rows = d.B * d.H;
  const int64_t nb = (int64_t)((n_rows + 7) / 8) * 8 * bpr;
  RC_REQUIRE(nb < (1ll << 31), RC_EINVAL, "too many pixels");
  const float scale = (float)(1.0 / sqrt((double)d.D));
  const unsigned blocks = (unsigned)nb;
#define RC_ALT(MODE, FULL, ONCE)                                                                            \
  hipLaunchKernelGGL((alt_fwd_kernel<T, R, MODE, FULL, ONCE>), dim3(blocks), dim3(256), 0, st, (const T*)f1t, \
                     (const T*)f2t, d, coords, cc, out, scale, bpr, n_rows)
  const bool full = d.Dp == DMAX;
  // (the streamed-f1 variant exists for full-depth features only: operands too large for the cache at fewer
  // channels take the plain loads)
  if (mode == RC_COORD_EXACT) {
    if (full && d.f1_once) RC_ALT(RC_COORD_EXACT, true, true);
    else if (full) RC_ALT(RC_COORD_EXACT, true, false);
    else RC_ALT(RC_COORD_EXACT, false, false);
  } else {
    if (full && d.f1_once) RC_ALT(RC_COORD_GRIDSAMPLE, true, true);
    else if (full) RC_ALT(RC_COORD_GRIDSAMPLE, true, false);
    else RC_ALT(RC_COORD_GRIDSAMPLE, false, false);
  }
#undef RC_ALT
  return rc_check_launch("rc_alt_fwd");
}

template <typename T>
int dispatch_alt(const void* f1t, const void* f2t, const alt_dev_layout& d, const float* coords, int cc,
                 float* out, int radius, int mode, hipStream_t st) {
  switch (radius) {
    case 1: return launch_alt<T, 1>(f1t, f2t, d, coords, cc, out, mode, st);
    case 2: return launch_alt<T, 2>(f1t, f2t, d, coords, cc, out, mode, st);
    case 3: return launch_alt<T, 3>(f1t, f2t, d, coords, cc, out, mode, st);
    case 4: return launch_alt<T, 4>(f1t, f2t, d, coords, cc, out, mode, st);
    case 5: return launch_alt<T, 5>(f1t, f2t, d, coords, cc, out, mode, st);
    case 6: return launch_alt<T, 6>(f1t, f2t, d, coords, cc, out, mode, st);
  }
  rc_set_error("alt: radius %d outside 1..%d", radius, RC_ALT_MAX_RADIUS);
  return RC_EUNSUPPORTED;
}

int validate_alt(const rc_alt_layout* a) {
  RC_REQUIRE(a != nullptr, RC_EINVAL, "null rc_alt_layout");
  rc_alt_layout ref;
  int rc = rc_alt_layout_init(a->B, a->D, a->H, a->W1, a->W2, a->L, &ref);
  if (rc) return rc;
  RC_REQUIRE(memcmp(&ref, a, sizeof(ref)) == 0, RC_EINVAL, "rc_alt_layout was not produced by rc_alt_layout_init");
  return RC_OK;
}

template <typename T>
int run_prepare(const void* f1, const void* f2, void* f1t, void* f2t, const rc_alt_layout* a, hipStream_t st) {
  const int D = (int)a->D, Dp = (int)a->Dp, H = (int)a->H;
  const unsigned BH = (unsigned)(a->B * a->H);
  frag_levels one = {};
  one.nlev = 1;
  one.width[0] = (int)a->W1;
  // 16-bit rows of whole, aligned 16-byte segments: the vector kernel (RC_ALT_PREPARE=elements: the other one)
  const char* how = getenv("RC_ALT_PREPARE");
  const bool elements = sizeof(T) != 2 || (how && !strcmp(how, "elements"));
  auto launch = [&](const void* src, void* dst, int W, const frag_levels& lv) {
    const dim3 grid((unsigned)((W + 127) / 128), (Dp + 63) / 64, BH);
    if constexpr (sizeof(T) == 2) {
      if (!elements && W % 8 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0) {
        const int rows_per_xcd = ((int)BH + 7) / 8;
        const int64_t nb = (int64_t)rows_per_xcd * 8 * grid.x * grid.y;
        if (nb < (1ll << 31)) {
          hipLaunchKernelGGL((to_fragments16_kernel<T>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)src, (T*)dst, D,
                             Dp, H, W, lv, (int)BH, rows_per_xcd);
          return;
        }
      }
    }
    hipLaunchKernelGGL((to_fragments_kernel<T>), grid, dim3(256), 0, st, (const T*)src, (T*)dst, D, Dp, H, W, lv);
  };
  launch(f1, f1t, (int)a->W1, one);
  int rc = rc_check_launch("rc_alt_prepare(f1)");
  if (rc) return rc;
  frag_levels all = {};
  all.nlev = ((a->L - 1) & ~1) + 1;  // the last level that has a copy
  for (int i = 0; i < all.nlev; ++i) {
    all.width[i] = a->width[i];
    all.offset[i] = (i & 1) ? -1 : a->f2_offset[i];
  }
  launch(f2, f2t, (int)a->W2, all);
  return rc_check_launch("rc_alt_prepare(f2)");
}

}  // namespace

extern "C" int rc_alt_layout_init(int64_t B, int64_t D, int64_t H, int64_t W1, int64_t W2, int L,
                                  rc_alt_layout* out) {
  RC_REQUIRE(out != nullptr, RC_EINVAL, "rc_alt_layout_init: null output");
  RC_REQUIRE(B > 0 && D > 0 && H > 0 && W1 > 0 && W2 > 0, RC_EINVAL, "rc_alt_layout_init: non-positive size");
  RC_REQUIRE(L >= 1 && L <= RC_MAX_LEVELS, RC_EUNSUPPORTED, "num_levels %d outside 1..%d", L, RC_MAX_LEVELS);
  RC_REQUIRE((W2 >> (L - 1)) >= 1, RC_EINVAL, "W2=%lld is too small for %d levels", (long long)W2, L);
  RC_REQUIRE(D <= DMAX, RC_EUNSUPPORTED, "alt: feature depth %lld above %d", (long long)D, DMAX);
  RC_REQUIRE(B * H < 65536, RC_EINVAL, "alt: B*H=%lld exceeds the grid z range; split the batch", (long long)(B * H));
  memset(out, 0, sizeof(*out));
  out->B = B; out->H = H; out->W1 = W1; out->W2 = W2; out->D = D; out->L = L;
  out->Dp = (D + 31) / 32 * 32;
  out->f1_elems = B * H * ((W1 + 15) / 16 * 16) * out->Dp;  // positions padded to blocks of 16
  int64_t off = 0;
  for (int i = 0; i < L; ++i) {
    out->width[i] = (int32_t)(W2 >> i);
    if (i & 1) continue;  // odd levels have no copy: the lookup derives them from the level below (offset 0)
    out->f2_offset[i] = off;
    off += B * H * (int64_t)((out->width[i] + 15) / 16 * 16) * out->Dp;
  }
  out->f2_elems = off;
  return RC_OK;
}

extern "C" int rc_alt_prepare(const void* f1, const void* f2, void* f1t, void* f2t, const rc_alt_layout* lay,
                              int in_dtype, void* stream) {
  RC_REQUIRE(f1 && f2 && f1t && f2t, RC_EINVAL, "null pointer argument");
  int rc = validate_alt(lay);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  switch (in_dtype) {
    case RC_F32: return run_prepare<float>(f1, f2, f1t, f2t, lay, st);
    case RC_F16: return run_prepare<rc_half>(f1, f2, f1t, f2t, lay, st);
    case RC_BF16: return run_prepare<rc_bf16>(f1, f2, f1t, f2t, lay, st);
  }
  rc_set_error("unknown in_dtype %d", in_dtype);
  return RC_EUNSUPPORTED;
}

extern "C" int rc_alt_fwd(const void* f1t, const void* f2t, const rc_alt_layout* lay, int in_dtype,
                          const float* coords, int coords_channels, float* out, int radius, int coord_mode,
                          void* stream) {
  RC_REQUIRE(f1t && f2t && coords && out, RC_EINVAL, "null pointer argument");
  int rc = validate_alt(lay);
  if (rc) return rc;
  RC_REQUIRE(coords_channels >= 1, RC_EINVAL, "coords must have at least one channel");
  RC_REQUIRE(coord_mode == RC_COORD_EXACT || coord_mode == RC_COORD_GRIDSAMPLE, RC_EINVAL, "bad coord_mode");
  if (coord_mode == RC_COORD_GRIDSAMPLE)
    RC_REQUIRE(lay->width[lay->L - 1] >= 2, RC_EINVAL, "grid_sample coordinates need every level at least 2 wide");
  RC_REQUIRE((uintptr_t)f1t % 16 == 0 && (uintptr_t)f2t % 16 == 0, RC_EINVAL, "working copies must be 16-byte aligned");
  alt_dev_layout d = to_dev(lay);
  // operands that cannot both stay in the 256 MiB Infinity Cache between lookups: f1 is streamed past it
  const int64_t esize = in_dtype == RC_F32 ? 4 : 2;
  const char* policy = getenv("RC_ALT_F1_POLICY");  // "stream" / "plain": overrides the size rule (tests, measurements)
  d.f1_once = policy ? !strcmp(policy, "stream") : (lay->f1_elems + lay->f2_elems) * esize > (200ll << 20);
  hipStream_t st = (hipStream_t)stream;
  switch (in_dtype) {
    case RC_F32: return dispatch_alt<float>(f1t, f2t, d, coords, coords_channels, out, radius, coord_mode, st);
    case RC_F16: return dispatch_alt<rc_half>(f1t, f2t, d, coords, coords_channels, out, radius, coord_mode, st);
    case RC_BF16: return dispatch_alt<rc_bf16>(f1t, f2t, d, coords, coords_channels, out, radius, coord_mode, st);
  }
  rc_set_error("unknown in_dtype %d", in_dtype);
  return RC_EUNSUPPORTED;
}

// Correlation lookup fused with the motion encoder's first layer:
//     out = relu(convc1(lookup(coords)))        convc1 = 1x1 convolution, L*(2r+1) -> 64 channels
//
// Replaces core/corr.py:127-146 / :44-51 (the lookup) followed by core/update.py:71,78
// (`cor = F.relu(self.convc1(corr))`) of the reference: the L*(2r+1)-channel fp32 tensor between
// the two (144 B written + 144 B read back per pixel and iteration) never exists.
//
// A wave owns 64 consecutive pixels of the flattened image.  Gather: one pixel per lane, exactly the plain
// lookup (two runs of 16-byte blocks per pixel, levels 1 and 3 derived in registers).  The taps are rounded to
// the pyramid's type and packed, K slot = 16 level + tap.  A 4x4 transpose over the lane rows
// (v_permlane16/32_swap) per packed dword then gives lane (g, li) the slots of LEVEL g of the pixel that lane
// (s, li) gathered, in register s: the A-operand layout of the MFMA (row = pixel, K = 8 (lane / 16) + j within a
// 32-slot step), so
//     C[pixel][out channel] += taps[pixel][K] * Wp[K][out channel],   K = 4 levels x 16 tap slots
// (unused slots carry zero weights and zero taps) runs once per 16 pixels.  Weights come pre-arranged in
// fragment order (rc_convc1_pack); the accumulators start from the bias.  MFMA rows are pixels, so a lane ends
// up with 4 pixels of one output channel per pass; the lane -> pixel assignment of the gather,
//     pixel = 32 (s / 2) + 8 (li / 4) + 4 (s % 2) + li % 4          (s = lane / 16, li = lane % 16),
// makes those 4 x 4 pixels of a lane 16 consecutive ones, so the activations leave as 16-byte stores, 64
// contiguous bytes per channel plane and instruction, with no second transpose (f32 outputs, odd sizes).
// 16-bit outputs made of whole 16-byte pieces take one more hop, through a wave-private LDS tile
// [channel][64 pixels]: every store instruction then writes whole 128-byte lines (8 planes x 64 pixels), which
// is also what lets outputs beyond the Infinity Cache leave through non-temporal stores; the gather is in
// natural order there (pixel = lane).
//   16-bit pyramid: taps rounded to the pyramid's type, v_mfma_f32_16x16x32 (f32 accumulate),
//                   output in that type (what autocast makes of convc1);
//   f32 pyramid:    v_mfma_f32_16x16x4_f32, f32 output.
#include "rc_common.h"
#include "window.h"

#include <math.h>

namespace {

typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr int FC_OUT = RC_CONVC1_OUT;  // output channels of convc1 (core/update.py:71)
constexpr int FC_SLOTS = 16;           // tap slots per level in the packed weights
constexpr int FC_THREADS = 256;

typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
// 16 pixels x 16 output channels += [16 x 16 tap slots] . [16 slots x 16]: a lane gives 4 slots of each operand
template <typename T> __device__ __forceinline__ f4v mfma16(uint2 a, uint2 b, f4v c);
template <> __device__ __forceinline__ f4v mfma16<rc_half>(uint2 a, uint2 b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4v, a), __builtin_bit_cast(h4v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f4v mfma16<rc_bf16>(uint2 a, uint2 b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4v, a), __builtin_bit_cast(s4v, b), c, 0, 0, 0);
}

template <typename VT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef VT v2 __attribute__((ext_vector_type(2)));
  const f2v f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, v2));
}

// one level's taps, two per dword (the odd one out paired with zero)
template <typename VT, int NT, bool PIN>
__device__ __forceinline__ void pack_level(const float (&tap)[NT], uint32_t (&pk)[(NT + 1) / 2]) {
#pragma unroll
  for (int j = 0; j < (NT + 1) / 2; ++j) {
    pk[j] = pack2<VT>(tap[2 * j], 2 * j + 1 < NT ? tap[2 * j + 1] : 0.f);
    if (PIN) asm volatile("" : "+v"(pk[j]));
  }
}

// max(v, floor) of two values, packed to the 16-bit type (floor = 0: ReLU; -inf: none).  f16 has a packed max, and
// rounding is monotone, so the maximum can be taken after the conversion: one instruction per pair
template <typename VT> __device__ __forceinline__ uint32_t act_pack2(float lo, float hi, float floor) {
  if constexpr (sizeof(VT) == 2 && !__is_same(VT, rc_bf16)) {
    typedef VT v2 __attribute__((ext_vector_type(2)));
    const f2v f = {lo, hi};
    const v2 fl = {(VT)floor, (VT)floor};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_convertvector(f, v2), fl));
  } else {
    return pack2<VT>(__builtin_amdgcn_fmed3f(lo, floor, __builtin_inff()), __builtin_amdgcn_fmed3f(hi, floor, __builtin_inff()));
  }
}

__device__ __forceinline__ void swap16(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap32(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

template <typename A> __device__ __forceinline__ A pick4(const A (&a)[RC_MAX_LEVELS], int i) {
  return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

// Packed weights (at most FC_OUT * 64 elements of the pyramid's type), t = out / 16, lane = 16 g + li,
// out = 16 t + li, tap slot of level g; NM = ceil(NT / 4) MFMA steps of 16 slots (4 per level):
//   16-bit: [((t*NM + m)*64 + lane)*4 + e]   <- W[out][g*NT + 4*m + e]
//   f32:    [(t*16 + e)*64 + lane]           <- W[out][g*NT + e]
// zero where the level or the tap does not exist.
template <typename T>
__global__ __launch_bounds__(256) void convc1_pack_kernel(const float* __restrict__ w, T* __restrict__ packed, int L,
                                                          int NT) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= FC_OUT * 64) return;
  int t, lane, tap;
  if (sizeof(T) == 2) {
    const int NM = (NT + 3) / 4, e = idx & 3, tm = idx >> 8;
    lane = (idx >> 2) & 63;
    t = tm / NM;
    tap = 4 * (tm - t * NM) + e;
    if (t >= 4) return;  // beyond the 4 NM fragments in use
  } else {
    lane = idx & 63;
    tap = (idx >> 6) & 15;
    t = idx >> 10;
  }
  const int li = lane & 15, g = lane >> 4;
  const float v = (g < L && tap < NT) ? w[(16 * t + li) * (L * NT) + g * NT + tap] : 0.f;
  packed[idx] = rc_from_float<T>(v);
}

__device__ __forceinline__ void swap16u(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}
__device__ __forceinline__ void swap32u(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}
// 4x4 transpose over lane rows: afterwards lane row g holds in r[p] what lane row p held in r[g]
__device__ __forceinline__ void transpose_rows(uint32_t (&r)[4]) {
  swap16u(r[0], r[1]);
  swap16u(r[2], r[3]);
  swap32u(r[0], r[2]);
  swap32u(r[1], r[3]);
}

// Optional coordinate update in front of the fused lookup (rc_lookup_convc1_fwd_update): coords_out == null for a
// plain rc_lookup_convc1_fwd
struct fc_update {
  const void* delta;   // [B,2,H,W1] in delta_dtype, channel 0 is added to x (null: nothing is added)
  int delta_dtype;     // RC_F32 / RC_F16 / RC_BF16
  float* coords_out;   // [B,2,H,W1]
  float* flow_out;     // [B,2,H,W1]
  int wgs_per_xcd;     // set by the launcher: > 0 = each XCD takes that many consecutive workgroups' pixels
};

// TILE: 16-bit outputs made of whole 16-byte pieces (decided by the launcher) leave through the LDS tile
template <typename VT, int R, int MODE, bool TILE>
__global__ __launch_bounds__(FC_THREADS) void lookup_convc1_kernel(const VT* __restrict__ vol, rc_dev_layout lay,
                                                                   const float* __restrict__ coords,
                                                                   int coords_channels,
                                                                   const VT* __restrict__ wpack,
                                                                   const float* __restrict__ bias,
                                                                   VT* __restrict__ out, int chunks_per_image,
                                                                   int n_items, int relu, int nts, fc_update upd) {
  constexpr bool IS16 = sizeof(VT) == 2;
  using PF = pair_fetch<VT, R, MODE>;
  using SF = single_fetch<VT, R, MODE>;
  constexpr int NT = 2 * R + 1, NW = PF::PI::NW;
  constexpr int NDW = (NT + 1) / 2;  // packed dwords per level (16-bit)
  constexpr int NM = (NT + 3) / 4;   // MFMA steps (16-bit): 16 tap slots each, 4 per level
  static_assert(NT <= FC_SLOTS, "a level's taps must fit its slots");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (image, 64-pixel chunk), wave-uniform.  A block's chunks are 256 consecutive pixels of ONE image (the
  // launcher counts chunks_per_image in whole blocks): chunks past the end of the image gather its last pixel,
  // store nothing, and still reach the block's barriers
  // (a launch that is resident all at once gives each XCD a contiguous eighth of the pixels, as the plain lookup
  // does: lookup.hip)
  int wg = blockIdx.x;
  if (upd.wgs_per_xcd > 0) {
    wg = (int)(blockIdx.x & 7) * upd.wgs_per_xcd + (int)(blockIdx.x >> 3);
    if (wg * (FC_THREADS / 64) >= n_items) return;  // the whole workgroup: n_items counts whole workgroups
  }
  const int item = wg * (FC_THREADS / 64) + wave;
  if (!TILE && item >= n_items) return;
  const int64_t HW_ = (int64_t)lay.H * lay.W1;
  // the lane's pixel inside the chunk (see the header): rows of the later MFMA land 16 consecutive pixels per lane
  // (16-bit outputs in whole 16-byte pieces go through an LDS tile instead and gather in natural order)
  static_assert(!TILE || IS16, "the LDS tile path is for 16-bit outputs");
  const bool vec = TILE || ((HW_ * (int64_t)sizeof(VT)) % 16 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);  // wave-uniform
  constexpr bool natural = TILE;
  const int poff = natural ? lane : 32 * (lane >> 5) + 8 * ((lane & 15) >> 2) + 4 * ((lane >> 4) & 1) + (lane & 3);
  const px_item it = locate_item(item, chunks_per_image, lay.H, lay.W1, poff);
  const int b = it.b, h = it.h, w1g = it.w1;  // lanes past the end of the image repeat its last pixel (never stored)
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const int g = lane >> 4;
  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;
  const int L = lay.L;

  // ---- phase 1: the plain lookup's gather, one pixel per lane: taps of every level in registers
  const int pidx = min(it.p0 + poff, (int)HW - 1);
  float x = (coords + uniform(((int64_t)b * coords_channels) * HW))[pidx];
  if (upd.coords_out) {  // wave-uniform: the GRU iteration's coordinate glue (core/raft_stereo.py:109-111, :120-123)
    const int64_t q0 = uniform(((int64_t)b * 2) * HW);
    // x, the delta and y are requested together and converted after (a value used inside the branch that loads
    // it is waited for there: three memory latencies in a row before the pyramid's loads could start)
    uint32_t draw = 0;
    if (upd.delta) {
      if (upd.delta_dtype == RC_F32) draw = (reinterpret_cast<const uint32_t*>(upd.delta) + q0)[pidx];
      else draw = (reinterpret_cast<const uint16_t*>(upd.delta) + q0)[pidx];
    }
    float y = (float)h;
    if (coords_channels > 1) y = (coords + uniform(((int64_t)b * coords_channels + 1) * HW))[pidx];
    if (upd.delta) {  // coords1 = coords1 + (delta_flow[:,0], 0): the fp32 add of the reference (delta promoted)
      float dlt;
      if (upd.delta_dtype == RC_F32) dlt = __uint_as_float(draw);
      else if (upd.delta_dtype == RC_F16) dlt = (float)__builtin_bit_cast(rc_half, (uint16_t)draw);
      else dlt = __uint_as_float(draw << 16);
      x += dlt;
    }
    if (it.p0 + poff < (int)HW) {  // lanes past the end of the image repeat its last pixel and store nothing
      (upd.coords_out + q0)[pidx] = x;
      (upd.coords_out + uniform(q0 + HW))[pidx] = y;
      (upd.flow_out + q0)[pidx] = x - (float)w1g;   // flow = coords1 - coords0, coords0 = the pixel grid
      (upd.flow_out + uniform(q0 + HW))[pidx] = y - (float)h;
    }
  }
  float tap[RC_MAX_LEVELS][NT];
  uint32_t pk[RC_MAX_LEVELS][NDW];  // 16-bit pyramids: the taps rounded to the pyramid's type, two per dword
#pragma unroll
  for (int i = 0; i < RC_MAX_LEVELS; ++i)
#pragma unroll
    for (int k = 0; k < NT; ++k) tap[i][k] = 0.f;
  if (L == 4) {  // both runs in flight together
    PF p0, p2;
    p0.issue(vol, lay, 0, x, w1g, h, b, blk_bytes);
    p2.issue(vol, lay, 2, x, w1g, h, b, blk_bytes);
    float lo[NW], hi[NW];
    p0.windows(lay, 0, lo, hi);
    interp_taps<R, MODE, NW>(lo, p0.s_lo, lay.width[0], p0.x_lo, p0.f_lo, tap[0]);
    interp_taps<R, MODE, NW>(hi, p0.s_hi, lay.width[1], p0.x_hi, p0.f_hi, tap[1]);
    if constexpr (IS16) {  // packed (and pinned: 10 registers instead of 18) while the second run is decoded
      pack_level<VT, NT, true>(tap[0], pk[0]);
      pack_level<VT, NT, true>(tap[1], pk[1]);
    }
    p2.windows(lay, 2, lo, hi);
    interp_taps<R, MODE, NW>(lo, p2.s_lo, lay.width[2], p2.x_lo, p2.f_lo, tap[2]);
    interp_taps<R, MODE, NW>(hi, p2.s_hi, lay.width[3], p2.x_hi, p2.f_hi, tap[3]);
    if constexpr (IS16) {
      pack_level<VT, NT, false>(tap[2], pk[2]);
      pack_level<VT, NT, false>(tap[3], pk[3]);
    }
  } else {
    if (L >= 2) {
      PF pf;
      pf.issue(vol, lay, 0, x, w1g, h, b, blk_bytes);
      float lo[NW], hi[NW];
      pf.windows(lay, 0, lo, hi);
      interp_taps<R, MODE, NW>(lo, pf.s_lo, lay.width[0], pf.x_lo, pf.f_lo, tap[0]);
      interp_taps<R, MODE, NW>(hi, pf.s_hi, lay.width[1], pf.x_hi, pf.f_hi, tap[1]);
    }
    if (L == 1) {
      SF sf;
      sf.issue(vol, lay, 0, x, w1g, h, b, blk_bytes);
      float v[NW];
      sf.window(lay, 0, v);
      interp_taps<R, MODE, NW>(v, sf.pos.wbase, lay.width[0], sf.xi, sf.frac, tap[0]);
    }
    if (L == 3) {
      SF sf;
      sf.issue(vol, lay, 2, x, w1g, h, b, blk_bytes);
      float v[NW];
      sf.window(lay, 2, v);
      interp_taps<R, MODE, NW>(v, sf.pos.wbase, lay.width[2], sf.xi, sf.frac, tap[2]);
    }
    if constexpr (IS16) {
#pragma unroll
      for (int i = 0; i < RC_MAX_LEVELS; ++i) pack_level<VT, NT, false>(tap[i], pk[i]);
    }
  }

  const int li = lane & 15;
  VT* obase = out + uniform(((int64_t)b * FC_OUT) * HW + it.p0);

  // ---- 16-bit pyramids: lane rows <-> levels (the MFMA A operand), shared by both output paths
  uint32_t aq[4][2 * NM];  // [pass s][dword of the level's 4 NM slots] of lane (g, li): level g of the pixel of lane (s, li)
  if constexpr (IS16) {
#pragma unroll
    for (int j = 0; j < 2 * NM; ++j) {
      if (j < NDW) {
        uint32_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pk[i][j];
        transpose_rows(r);
#pragma unroll
        for (int s = 0; s < 4; ++s) aq[s][j] = r[s];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) aq[s][j] = 0u;
      }
    }
    if constexpr (TILE) {
      // ---- the main path (16-bit outputs in whole 16-byte pieces): 16 output channels at a time -- 8 MFMAs,
      // activation on the packed pairs -- into an LDS tile [32 channels][the BLOCK's 256 pixels] (pitch 528 B:
      // the 16 lanes of a lane row write 16 different bank groups), from which every store instruction writes
      // 512 contiguous bytes of 2 channel planes.  The size of these pieces is what outputs beyond the Infinity
      // Cache pay for: their stores are non-temporal (they go straight to memory), and a stream of 128-byte
      // pieces -- what one wave's 64 pixels make -- into 64 planes reaches 2.4 TB/s where 512-byte pieces
      // reach 3.6 (tools/lab_store_patterns.hip).  Straight from the accumulators a store would cover 64 bytes of
      // 16 planes.
      constexpr int PITCH = FC_THREADS * 2 + 16;
      __shared__ __attribute__((aligned(16))) unsigned char tile[(FC_OUT / 2) * PITCH];
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const float act_floor = relu ? 0.f : -__builtin_inff();
      uint2 wq[4][NM];
      float bq[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int m = 0; m < NM; ++m) wq[t][m] = reinterpret_cast<const uint2*>(wpack)[(t * NM + m) * 64 + lane];
        bq[t] = bias[16 * t + li];
      }
      const int p0b = it.p0 - 64 * wave;  // the block's first pixel
      VT* oblk = out + uniform(((int64_t)b * FC_OUT) * HW + p0b);
      constexpr int WPB = FC_THREADS / 64, LPR = FC_THREADS / 8;  // waves per block; lanes per tile row (16 B each)
      constexpr int CPW = (FC_OUT / 2) / WPB, PPI = 64 / LPR;     // channels per wave and half; planes per instruction
      const int seg = lane & (LPR - 1);
      const bool inside = p0b + 8 * seg < (int)HW;  // HW is a multiple of 8 here: a piece is inside or outside
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * hf + tt;
          const float bvt = bq[t];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            f4v a = {bvt, bvt, bvt, bvt};
#pragma unroll
            for (int m = 0; m < NM; ++m) a = mfma16<VT>(make_uint2(aq[s][2 * m], aq[s][2 * m + 1]), wq[t][m], a);
            const int px = 64 * wave + 16 * s + 4 * g;  // natural gather order: pass s, row 4 g + r is pixel 16 s + 4 g + r
            *reinterpret_cast<uint2*>(tile + (16 * tt + li) * PITCH + px * 2) =
                make_uint2(act_pack2<VT>(a[0], a[1], act_floor), act_pack2<VT>(a[2], a[3], act_floor));
          }
        }
        if (hf == 0) {
          // the stores below are written in assembly and the compiler does not count them: a wait for the second
          // half's weights after them would wait for them too.  Naming the registers puts it here.
#pragma unroll
          for (int t = 2; t < 4; ++t) {
#pragma unroll
            for (int m = 0; m < NM; ++m) asm volatile("" : "+v"(wq[t][m].x), "+v"(wq[t][m].y));
            asm volatile("" : "+v"(bq[t]));
          }
        }
        __syncthreads();
        u4 w[CPW / PPI];
        u4* dst[CPW / PPI];
#pragma unroll
        for (int i = 0; i < CPW / PPI; ++i) {
          const int cl = CPW * wave + PPI * i + lane / LPR, ch = 32 * hf + cl;
          w[i] = *reinterpret_cast<const u4*>(tile + cl * PITCH + seg * 16);
          dst[i] = reinterpret_cast<u4*>(oblk + ((uint32_t)ch * (uint32_t)HW + (uint32_t)(8 * seg)));
        }
        if (inside) {
          // Spelled out: of `if (nts) __builtin_nontemporal_store(..) else plain store` the compiler makes ONE
          // plain store (it merges the two arms and the hint is lost); otherwise agent-scope stores, as in the
          // plain lookup (store_taps, lookup.hip)
#ifdef RC_LAB_STORE_POLICY
#pragma unroll
          for (int i = 0; i < CPW / PPI; ++i)
            asm volatile("global_store_dwordx4 %0, %1, off " RC_LAB_STORE_POLICY ::"v"(dst[i]), "v"(w[i]) : "memory");
#else
          if (nts) {  // wave-uniform
#pragma unroll
            for (int i = 0; i < CPW / PPI; ++i)
              asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst[i]), "v"(w[i]) : "memory");
          } else {
#pragma unroll
            for (int i = 0; i < CPW / PPI; ++i)
              asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst[i]), "v"(w[i]) : "memory");
          }
#endif
        }
        if (hf == 0) __syncthreads();  // reads done before the second half overwrites the tile
      }
      return;
    }
  }

  // ---- the general path (f32 pyramids, odd sizes): all 64 channels' sums in registers
  // weights (B operand: column = out channel 16 t + li, K slots of level g) and bias (C column li)
  uint2 wa16[4][NM];
  float wa32[4][NT];
  f4v acc[4][4];  // [pass][t]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if constexpr (IS16) {
#pragma unroll
      for (int m = 0; m < NM; ++m) wa16[t][m] = reinterpret_cast<const uint2*>(wpack)[(t * NM + m) * 64 + lane];
    } else {
#pragma unroll
      for (int e = 0; e < NT; ++e) wa32[t][e] = reinterpret_cast<const float*>(wpack)[(t * 16 + e) * 64 + lane];
    }
    const float bvt = bias[16 * t + li];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s][t] = (f4v){bvt, bvt, bvt, bvt};
  }
  if constexpr (IS16) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < NM; ++m)
          acc[s][t] = mfma16<VT>(make_uint2(aq[s][2 * m], aq[s][2 * m + 1]), wa16[t][m], acc[s][t]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < NT; ++e) {
      uint32_t r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = __float_as_uint(tap[i][e]);
      transpose_rows(r);  // r[s] of lane (g, li): tap e of level g of the pixel of lane (s, li)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r[s]), wa32[t][e], acc[s][t], 0, 0, 0);
    }
  }

  // ---- phase 3: activation and stores.  acc[s][t][r] = channel 16 t + li of pixel 32 (s / 2) + 8 g + 4 (s % 2) + r:
  // per (t, s / 2) the lane holds 8 consecutive pixels of one channel plane
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t plane = (uint32_t)(16 * t + li) * (uint32_t)HW;
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = acc[2 * hs + e][t][r];
          v[4 * e + r] = relu ? fmaxf(a, 0.f) : a;
        }
      const int q = 32 * hs + 8 * g;  // first of the 8 pixels, inside the chunk
      VT* dst = obase + (plane + (uint32_t)q);
      if (vec) {  // whole 16-byte pieces: HW is a multiple of the piece, so a piece is inside the image or outside
        if (it.p0 + q < (int)HW) {
          if constexpr (IS16) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 w = {pack2<VT>(v[0], v[1]), pack2<VT>(v[2], v[3]), pack2<VT>(v[4], v[5]), pack2<VT>(v[6], v[7])};
            *reinterpret_cast<u4*>(dst) = w;
          } else {
            const f4v w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f4v*>(dst) = w0;
            *reinterpret_cast<f4v*>(dst + 4) = w1;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (it.p0 + q + e < (int)HW) dst[e] = rc_from_float<VT>(v[e]);
      }
    }
  }
}

template <typename VT, int R>
int launch_fused(const void* vol, rc_dev_layout d, const float* coords, int cc, const void* wpack,
                 const float* bias, void* out, int mode, int relu, hipStream_t st, const fc_update& upd) {
  constexpr int WPB = FC_THREADS / 64;
  const int chunks = (d.H * d.W1 + FC_THREADS - 1) / FC_THREADS * WPB;  // whole blocks per image
  const int n_items = (int)((int64_t)d.B * chunks);
  int blocks = (n_items + (FC_THREADS / 64) - 1) / (FC_THREADS / 64);
  fc_update updx = upd;
  updx.wgs_per_xcd = 0;
  if (blocks <= 1024) {  // resident all at once (256 CUs x 4 of these workgroups)
    updx.wgs_per_xcd = (blocks + 7) / 8;
    blocks = updx.wgs_per_xcd * 8;
  }
  // outputs that cannot stay in the Infinity Cache until their consumer runs leave through non-temporal stores
  // (the same rule and threshold as the plain lookup, lookup.hip); the LDS-tile path only: its stores cover
  // 512 contiguous bytes
  const int nts = sizeof(VT) == 2 && (int64_t)d.B * FC_OUT * d.H * d.W1 * (int64_t)sizeof(VT) > (160ll << 20);
  constexpr bool IS16 = sizeof(VT) == 2;
  const bool tile = IS16 && ((int64_t)d.H * d.W1) % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
#define RC_LAUNCH(MODE, TILE)                                                                                  \
  do {                                                                                                         \
    set_row_bias(d, vt_info<VT>::RB, pair_info<vt_info<VT>::RB, R, MODE>::NW2);                               \
    hipLaunchKernelGGL((lookup_convc1_kernel<VT, R, MODE, TILE>), dim3(blocks), dim3(FC_THREADS), 0, st,       \
                       (const VT*)vol, d, coords, cc, (const VT*)wpack, bias, (VT*)out, chunks, n_items, relu, \
                       nts, updx);                                                                             \
  } while (0)
  if constexpr (IS16) {
    if (tile) {
      if (mode == RC_COORD_EXACT) RC_LAUNCH(RC_COORD_EXACT, true);
      else RC_LAUNCH(RC_COORD_GRIDSAMPLE, true);
      return rc_check_launch("rc_lookup_convc1_fwd");
    }
  }
  if (mode == RC_COORD_EXACT) RC_LAUNCH(RC_COORD_EXACT, false);
  else RC_LAUNCH(RC_COORD_GRIDSAMPLE, false);
#undef RC_LAUNCH
  return rc_check_launch("rc_lookup_convc1_fwd");
}

template <typename VT>
int dispatch_fused(const void* vol, const rc_dev_layout& d, const float* coords, int cc, const void* wpack,
                   const float* bias, void* out, int radius, int mode, int relu, hipStream_t st, const fc_update& upd) {
  switch (radius) {
    case 1: return launch_fused<VT, 1>(vol, d, coords, cc, wpack, bias, out, mode, relu, st, upd);
    case 2: return launch_fused<VT, 2>(vol, d, coords, cc, wpack, bias, out, mode, relu, st, upd);
    case 3: return launch_fused<VT, 3>(vol, d, coords, cc, wpack, bias, out, mode, relu, st, upd);
    case 4: return launch_fused<VT, 4>(vol, d, coords, cc, wpack, bias, out, mode, relu, st, upd);
    case 5: return launch_fused<VT, 5>(vol, d, coords, cc, wpack, bias, out, mode, relu, st, upd);
    case 6: return launch_fused<VT, 6>(vol, d, coords, cc, wpack, bias, out, mode, relu, st, upd);
    case 7: return launch_fused<VT, 7>(vol, d, coords, cc, wpack, bias, out, mode, relu, st, upd);
  }
  rc_set_error("fused lookup + convc1: radius %d outside 1..%d", radius, RC_CONVC1_MAX_RADIUS);
  return RC_EUNSUPPORTED;
}

}  // namespace

extern "C" int rc_convc1_pack(const float* weight, int num_levels, int radius, int dtype, void* packed,
                              void* stream) {
  RC_REQUIRE(weight && packed, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(num_levels >= 1 && num_levels <= RC_MAX_LEVELS, RC_EUNSUPPORTED, "num_levels %d outside 1..%d",
             num_levels, RC_MAX_LEVELS);
  RC_REQUIRE(radius >= 1 && radius <= RC_CONVC1_MAX_RADIUS, RC_EUNSUPPORTED,
             "fused lookup + convc1: radius %d outside 1..%d", radius, RC_CONVC1_MAX_RADIUS);
  RC_REQUIRE((uintptr_t)packed % 16 == 0, RC_EINVAL, "packed weights must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int NT = 2 * radius + 1, blocks = FC_OUT * 64 / 256;
  switch (dtype) {
    case RC_F32:
      hipLaunchKernelGGL((convc1_pack_kernel<float>), dim3(blocks), dim3(256), 0, st, weight, (float*)packed,
                         num_levels, NT);
      break;
    case RC_F16:
      hipLaunchKernelGGL((convc1_pack_kernel<rc_half>), dim3(blocks), dim3(256), 0, st, weight, (rc_half*)packed,
                         num_levels, NT);
      break;
    case RC_BF16:
      hipLaunchKernelGGL((convc1_pack_kernel<rc_bf16>), dim3(blocks), dim3(256), 0, st, weight, (rc_bf16*)packed,
                         num_levels, NT);
      break;
    default: rc_set_error("unknown dtype %d", dtype); return RC_EUNSUPPORTED;
  }
  return rc_check_launch("rc_convc1_pack");
}

static int fused_any(const void* vol, const rc_layout* lay, const float* coords, int coords_channels,
                     const void* packed_weight, const float* bias, void* out, int radius, int coord_mode, int relu,
                     void* stream, const fc_update& upd) {
  RC_REQUIRE(packed_weight && bias, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(radius <= RC_CONVC1_MAX_RADIUS, RC_EUNSUPPORTED, "fused lookup + convc1: radius %d outside 1..%d",
             radius, RC_CONVC1_MAX_RADIUS);
  int rc = check_common(lay, vol, coords, out, coords_channels, radius, coord_mode);
  if (rc) return rc;
  RC_REQUIRE(!lay->full, RC_EINVAL, "rc_lookup_convc1_fwd takes a forward pyramid (rc_pyramid_layout)");
  RC_REQUIRE((int64_t)RC_CONVC1_OUT * lay->H * lay->W1 < (1ll << 29), RC_EINVAL,
             "one image's output planes exceed the 32-bit lane offset range");
  RC_REQUIRE((uintptr_t)packed_weight % 16 == 0 && (uintptr_t)bias % 16 == 0, RC_EINVAL,
             "packed weights and bias must be 16-byte aligned");
  const rc_dev_layout d = rc_to_dev(lay);
  hipStream_t st = (hipStream_t)stream;
  switch (lay->dtype) {
    case RC_F32:
      return dispatch_fused<float>(vol, d, coords, coords_channels, packed_weight, bias, out, radius, coord_mode,
                                   relu, st, upd);
    case RC_F16:
      return dispatch_fused<rc_half>(vol, d, coords, coords_channels, packed_weight, bias, out, radius, coord_mode,
                                     relu, st, upd);
    case RC_BF16:
      return dispatch_fused<rc_bf16>(vol, d, coords, coords_channels, packed_weight, bias, out, radius, coord_mode,
                                     relu, st, upd);
  }
  rc_set_error("unknown pyramid dtype %d", lay->dtype);
  return RC_EUNSUPPORTED;
}

extern "C" int rc_lookup_convc1_fwd(const void* vol, const rc_layout* lay, const float* coords,
                                    int coords_channels, const void* packed_weight, const float* bias, void* out,
                                    int radius, int coord_mode, int relu, void* stream) {
  const fc_update none = {nullptr, RC_F32, nullptr, nullptr};
  return fused_any(vol, lay, coords, coords_channels, packed_weight, bias, out, radius, coord_mode, relu, stream, none);
}

extern "C" int rc_lookup_convc1_fwd_update(const void* vol, const rc_layout* lay, const float* coords,
                                           int coords_channels, const void* delta_flow, int delta_dtype,
                                           float* coords_out, float* flow_out, const void* packed_weight,
                                           const float* bias, void* out, int radius, int coord_mode, int relu,
                                           void* stream) {
  RC_REQUIRE(coords_out && flow_out, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(delta_dtype == RC_F32 || delta_dtype == RC_F16 || delta_dtype == RC_BF16, RC_EUNSUPPORTED,
             "unknown delta_flow dtype %d", delta_dtype);
  const fc_update upd = {delta_flow, delta_dtype, coords_out, flow_out};
  return fused_any(vol, lay, coords, coords_channels, packed_weight, bias, out, radius, coord_mode, relu, stream, upd);
}

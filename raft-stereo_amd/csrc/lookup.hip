// Per-iteration correlation lookup, forward and adjoint, over the group-skewed blocked pyramid.
//
// Replaces core/corr.py:127-146 (reg), core/corr.py:44-51 + sampler/sampler_kernel.cu:19-105
// (reg_cuda) of the reference.  One launch covers all levels.  A wave owns 64 consecutive
// pixels of one image row, one pixel per lane.  In the rc_layout a pixel's column of a level is
// cut into aligned groups of 8 (f16/bf16) or 4 (f32) consecutive w2, one 16-byte block each,
// and consecutive groups of a pixel sit in consecutive block rows; a 2r+2 tap window is the
// 2-3 (16-bit) or 3-4 (f32) blocks that cover it.  The lane loads them with 16-byte loads and
// shifts the taps into place in registers.  Pixels with similar disparity use the same block
// rows, so each load instruction of the wave reads a few long contiguous runs.  Everything that
// does not depend on the lane (row, chunk, level bases, output planes) is kept wave-uniform, so
// loads and stores use scalar base + 32-bit lane offset addressing.  Outputs are written per
// channel plane, 256 B per wave store.
#include "rc_common.h"
#include "window.h"

#include <math.h>

namespace {

constexpr int LK_THREADS = 256;  // 4 waves, each its own 64-pixel chunk

// Optional coordinate update in front of a lookup (rc_lookup_fwd_update); all null for a plain lookup
struct lk_update {
  const float* delta;  // [B,2,H,W1]; channel 0 is added to x, channel 1 is ignored (stereo: epipolar projection)
  float* coords_out;   // [B,2,H,W1]
  float* flow_out;     // [B,2,H,W1]
  int wgs_per_xcd;     // set by the launcher: > 0 = each XCD takes that many consecutive workgroups' pixels
};
// a launch of at most this many workgroups is resident all at once (256 CUs x 4 of these 4-wave workgroups)
constexpr int LK_ONE_ROUND_WGS = 1024;

// Stores the 2r+1 taps of one level: one aligned 256-byte piece per channel plane and wave.
//   NTS:  non-temporal stores, for outputs far larger than the caches (they are read once, by the next kernel);
//   else: agent-scope stores (`sc1`): written through the XCD's L2 as the kernel runs.  Plain stores stay in
//         that L2 as dirty lines and are written back when the kernel ends -- a tail with nothing to overlap
//         (KITTI, 8 pairs: 11.1 us against 9.0); the kernel that reads the taps next finds them in the
//         Infinity Cache either way (tools/lab_consumer.py: lookup + cast 20.0 us with both).
template <int NT, bool NTS, typename OT>
__device__ __forceinline__ void store_taps(OT* __restrict__ out, int64_t plane0, int64_t HW, int lane,
                                           const float (&tap)[NT]) {
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    OT* base = out + uniform(plane0 + k * HW);  // scalar
    if constexpr (sizeof(OT) == 4) {
#ifdef RC_LAB_STORE_POLICY
      asm volatile("global_store_dword %0, %1, %2 " RC_LAB_STORE_POLICY ::"v"(lane * 4), "v"(tap[k]), "s"(base) : "memory");
#else
      // (spelled out so that the plane's base stays in scalar registers: the builtin takes a 64-bit vector address)
      if constexpr (NTS) asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(lane * 4), "v"(tap[k]), "s"(base) : "memory");
      else asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(lane * 4), "v"(tap[k]), "s"(base) : "memory");
#endif
    } else {
      // output in the volume's 16-bit type (rc_lookup_fwd_vt): the fp32 tap rounded once, 128 bytes per plane
      // (the fp32 value pinned in a register first: left alone, the compiler folds the interpolation's last
      // fma and the conversion into one v_fma_mixlo_f16, which rounds the unrounded sum -- ties then fall
      // differently from rc_lookup_fwd's taps converted afterwards)
      float t32 = tap[k];
      asm volatile("" : "+v"(t32));
      const uint32_t bits = (uint32_t)__builtin_bit_cast(uint16_t, rc_from_float<OT>(t32));
      if constexpr (NTS) asm volatile("global_store_short %0, %1, %2 nt" ::"v"(lane * 2), "v"(bits), "s"(base) : "memory");
      else asm volatile("global_store_short %0, %1, %2 sc1" ::"v"(lane * 2), "v"(bits), "s"(base) : "memory");
    }
  }
}

// A forward pyramid keeps its even levels; levels (i, i+1) are served by one run of level i (pair_fetch in
// window.h).  ALL4 (the model's configuration, 4 levels): the 16-byte loads of both runs are issued back to
// back, then decoded, interpolated and stored.  Otherwise one pair (or the odd last level) at a time.
template <typename VT, int R, int MODE, bool ALL4, bool NTS, typename OT>
__global__ __launch_bounds__(LK_THREADS) void lookup_fwd_kernel(const VT* __restrict__ vol,
                                                                rc_dev_layout lay,
                                                                const float* __restrict__ coords,
                                                                int coords_channels,
                                                                OT* __restrict__ out,
                                                                int chunks_per_image, int n_items, lk_update upd) {
  using PF = pair_fetch<VT, R, MODE>;
  using SF = single_fetch<VT, R, MODE>;
  constexpr int NT = 2 * R + 1, NW = PF::PI::NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroups go to the 8 XCDs round-robin.  When all of them are resident at once (upd.wgs_per_xcd > 0: the launch
  // is one wave of workgroups), each XCD takes a contiguous eighth of the pixels (KITTI, 8 pairs: 9.7 -> 9.5 us);
  // over several rounds of workgroups the plain order is the faster one (64 pairs: 86.9 against 88.6 us).
  int wg = blockIdx.x;
  if (upd.wgs_per_xcd > 0) wg = (int)(blockIdx.x & 7) * upd.wgs_per_xcd + (int)(blockIdx.x >> 3);
  const int item = wg * (int)(blockDim.x >> 6) + wave;  // (image, 64-pixel chunk), wave-uniform; 1 or 4 waves per workgroup
  if (item >= n_items) return;
  const px_item it = locate_item(item, chunks_per_image, lay.H, lay.W1, lane);
  if (!it.live) return;
  const int b = it.b, h = it.h, w1 = it.w1;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const int L = ALL4 ? 4 : lay.L;
  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;  // distance between block rows of a pixel

  // (non-temporal: the lookup is the coordinates' one reader here, and 32 iterations' worth of them, 61 MB at
  // KITTI with 8 pairs, would otherwise sit in the Infinity Cache beside the pyramid: 8.63 -> 8.56 us per lookup)
  float x = __builtin_nontemporal_load(coords + uniform(((int64_t)b * coords_channels) * HW + it.p0) + lane);
  if (upd.delta) {  // wave-uniform: the GRU iteration's coordinate update folded into this lookup
    // core/raft_stereo.py:120-123 (delta_flow[:,1] = 0; coords1 = coords1 + delta_flow) and :111
    // (flow = coords1 - coords0, coords0 = the pixel grid of core/utils/utils.py coords_grid)
    const int64_t q0 = uniform(((int64_t)b * 2) * HW + it.p0);
    const float dlt = (upd.delta + q0)[lane];  // x, the delta and y are requested together
    float y = (float)h;
    if (coords_channels > 1) y = (coords + uniform(((int64_t)b * coords_channels + 1) * HW + it.p0))[lane];
    x += dlt;
    (upd.coords_out + q0)[lane] = x;
    (upd.coords_out + uniform(q0 + HW))[lane] = y;
    (upd.flow_out + q0)[lane] = x - (float)w1;
    (upd.flow_out + uniform(q0 + HW))[lane] = y - (float)h;
  }
  const int64_t out_u = ((int64_t)b * L * NT) * HW + it.p0;  // wave-uniform

  auto emit_pair = [&](PF& pf, int i, const float (&lo)[NW], const float (&hi)[NW]) {
    float tap[NT];
    interp_taps<R, MODE, NW>(lo, pf.s_lo, lay.width[i], pf.x_lo, pf.f_lo, tap);
    store_taps<NT, NTS, OT>(out, out_u + (int64_t)i * NT * HW, HW, lane, tap);
    interp_taps<R, MODE, NW>(hi, pf.s_hi, lay.width[i + 1], pf.x_hi, pf.f_hi, tap);
    store_taps<NT, NTS, OT>(out, out_u + (int64_t)(i + 1) * NT * HW, HW, lane, tap);
  };
  auto finish_pair = [&](PF& pf, int i) {
    float lo[NW], hi[NW];
    pf.windows(lay, i, lo, hi);
    emit_pair(pf, i, lo, hi);
  };

  if constexpr (ALL4) {
    PF p0, p2;
    p0.issue(vol, lay, 0, x, w1, h, b, blk_bytes);
    p2.issue(vol, lay, 2, x, w1, h, b, blk_bytes);
    float lo[NW], hi[NW];
    p0.windows(lay, 0, lo, hi);
    // The stores are written in assembly, so the compiler does not count them: its wait for the second run's
    // loads after them would be a wait for them too (a store's acknowledgement takes far longer than a load
    // that was issued long before).  Naming the registers here puts that wait before the first store.
    // (Directly, not through a pointer or a reference parameter: that turns the array into scratch memory.)
#pragma unroll
    for (int k = 0; k < PF::PI::ND; ++k) asm volatile("" : "+v"(p2.d[k]));
    emit_pair(p0, 0, lo, hi);
    finish_pair(p2, 2);
  } else {
#pragma unroll
    for (int i = 0; i + 1 < RC_MAX_LEVELS; i += 2) {
      if (i + 1 < L) {  // wave-uniform
        PF pf;
        pf.issue(vol, lay, i, x, w1, h, b, blk_bytes);
        finish_pair(pf, i);
      }
    }
    if (L & 1) {  // the last level stands alone
      const int i = L - 1;
      SF sf;
      sf.issue(vol, lay, i, x, w1, h, b, blk_bytes);
      float v[NW], tap[NT];
      sf.window(lay, i, v);
      interp_taps<R, MODE, NW>(v, sf.pos.wbase, lay.width[i], sf.xi, sf.frac, tap);
      store_taps<NT, NTS, OT>(out, out_u + (int64_t)i * NT * HW, HW, lane, tap);
    }
  }
}

// Adjoint: grad_vol (float pyramid, RB = 4) += scatter of grad_out.  The pixel owns its column
// of every level, so plain read-modify-write of its 16-byte blocks is race free.  Per level the
// gradient planes and the blocks are loaded first, then updated and stored; blocks of groups that
// do not exist (window hanging over the image border) are loaded from a valid address and not
// stored.
template <int R, int MODE>
__global__ __launch_bounds__(LK_THREADS) void lookup_bwd_kernel(const float* __restrict__ coords,
                                                                int coords_channels,
                                                                const float* __restrict__ gout,
                                                                float* __restrict__ gvol,
                                                                rc_dev_layout lay, int chunks_per_row,
                                                                int n_items) {
  constexpr int RB = 4, LG = 2;
  using WI = win_info<RB, R, MODE>;
  constexpr int NT = WI::NT, EXT = WI::EXT, NW = WI::NW, NBLK = WI::NBLK, ND = WI::ND;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * (LK_THREADS / 64) + wave;
  if (item >= n_items) return;
  const int bh = item / chunks_per_row;
  const int c0 = (item - bh * chunks_per_row) * 64;
  const int w1 = c0 + lane;
  if (w1 >= lay.W1) return;
  const int b = bh / lay.H, h = bh - b * lay.H;
  const int64_t HW = (int64_t)lay.H * lay.W1;
  const uint32_t blk_bytes = (uint32_t)lay.pitch * 16u;

  const float x = (coords + uniform(((int64_t)b * coords_channels) * HW + (int64_t)h * lay.W1 + c0))[lane];
  const int64_t g_u = ((int64_t)b * lay.L * NT) * HW + (int64_t)h * lay.W1 + c0;  // wave-uniform

  for (int i = 0; i < lay.L; ++i) {
    const int Wi = lay.width[i], nb = lay.widthp[i] >> LG;
    const float xi = x * (1.0f / (float)(1 << i));
    const float xf = floorf(xi);
    const float a = xi - xf;
    const win_pos pos = locate_window<RB, LG, NW>(rc_safe_int(xf) - R - EXT, w1, i, Wi, nb, lay.bias[i]);
    char* lvl = reinterpret_cast<char*>(gvol) +
                uniform(lay.offset[i] * (int64_t)sizeof(float) + (int64_t)bh * nb * blk_bytes);
    const bool tiny = nb < NBLK;

    float g[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) g[k] = (gout + uniform(g_u + ((int64_t)i * NT + k) * HW))[lane];

    // the blocks of the window: block k holds group g0 + k; `own[k]` = the group exists and this
    // lane's window reaches it
    uint32_t off[NBLK];
    bool own[NBLK];
    float4 q[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      const int gk = pos.g0 + k;
      own[k] = pos.any_ok && k <= pos.nbk && gk >= 0 && gk < nb;
      const uint32_t sb = row_step(pos.sb0, (uint32_t)(own[k] ? k : 0), (uint32_t)nb, tiny);
      off[k] = __umul24(sb, blk_bytes) + (uint32_t)w1 * 16u;
      q[k] = *reinterpret_cast<const float4*>(lvl + off[k]);
    }

    // t[m] = gradient of window entry m (w2 = wbase + m)
    float t[ND];
#pragma unroll
    for (int m = 0; m < ND; ++m) t[m] = 0.f;
    if constexpr (MODE == RC_COORD_EXACT) {
      const float bw = 1.0f - a;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        t[k] += g[k] * bw;
        t[k + 1] += g[k] * a;
      }
    } else {
      const float wm1 = (float)(Wi - 1);
      int dlt[NT];
      float w0[NT], w1w[NT];
      bool odd = false;
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        int i0k;
        rc_gridsample_coords(xi, k - R, wm1, &i0k, &w0[k], &w1w[k]);
        dlt[k] = i0k - (pos.wbase + EXT + k);
        odd |= (dlt[k] != 0);
      }
      if (__builtin_expect(__any(odd), 0)) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const float a0 = g[k] * w0[k], a1 = g[k] * w1w[k];
          t[k] += dlt[k] < 0 ? a0 : 0.f;
          t[k + 1] += dlt[k] < 0 ? a1 : (dlt[k] == 0 ? a0 : 0.f);
          t[k + 2] += dlt[k] == 0 ? a1 : (dlt[k] > 0 ? a0 : 0.f);
          t[k + 3] += dlt[k] > 0 ? a1 : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          t[k + EXT] += g[k] * w0[k];
          t[k + EXT + 1] += g[k] * w1w[k];
        }
      }
    }
#pragma unroll
    for (int m = 0; m < NW; ++m) t[m] = ((unsigned)(pos.wbase + m) < (unsigned)Wi) ? t[m] : 0.f;

    // move window entry m to slot o + m of the blocks
    {
      const bool s1 = (pos.o & 1) != 0, s2 = (pos.o & 2) != 0;
#pragma unroll
      for (int m = ND - 1; m >= 1; --m) {
        const float lo = t[m - 1], hi = t[m];
        t[m] = s1 ? lo : hi;
      }
      t[0] = s1 ? 0.f : t[0];
#pragma unroll
      for (int m = ND - 1; m >= 2; --m) {
        const float lo = t[m - 2], hi = t[m];
        t[m] = s2 ? lo : hi;
      }
      t[1] = s2 ? 0.f : t[1];
      t[0] = s2 ? 0.f : t[0];
    }
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      if (own[k]) {
        float4 r = q[k];
        r.x += t[4 * k + 0];
        r.y += t[4 * k + 1];
        r.z += t[4 * k + 2];
        r.w += t[4 * k + 3];
        *reinterpret_cast<float4*>(lvl + off[k]) = r;
      }
    }
  }
}

// One lookup's contribution to one (level, pixel) gradient row held in LDS: the 2r+1 tap gradients g
// spread over the window entries around xi (the level's coordinate).  The row belongs to the calling
// thread alone; its cells are S floats apart (the rows of a workgroup's S pixels are interleaved, so lane
// p is always in bank p whatever its window), and it has ADJ_PAD cells on either side of the image's Wi:
// cell w2 is row[(w2 + PAD) * S].  A window that touches the image at all lies inside the padded row, so
// every lookup is the same ten-odd read-add-writes at constant offsets from one address: no per-entry
// clamp, no second code path for windows at the image's edge (at the coarse levels most windows are).
template <int R, int MODE>
struct adj_pad {
  static constexpr int value = win_info<4, R, MODE>::NW - 1;
};

template <int R, int MODE, int S>
__device__ __forceinline__ void adjoint_accumulate(float* row, int Wi, float xi, float wm1,
                                                   const float (&g)[2 * R + 1]) {
  using WI = win_info<4, R, MODE>;
  constexpr int NT = WI::NT, EXT = WI::EXT, NW = WI::NW, PAD = adj_pad<R, MODE>::value;
  const float xf = floorf(xi);
  const float a = xi - xf;
  const int wbase = rc_safe_int(xf) - R - EXT;
  if (wbase + NW - 1 < 0 || wbase >= Wi) return;  // window outside the image

  // tw[m] = gradient of window entry m (w2 of this level = wbase + m)
  float tw[NW + 1];
#pragma unroll
  for (int m = 0; m <= NW; ++m) tw[m] = 0.f;
  if constexpr (MODE == RC_COORD_EXACT) {
    const float bw = 1.0f - a;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      tw[k] += g[k] * bw;
      tw[k + 1] += g[k] * a;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      int i0k;
      float w0, w1w;
      rc_gridsample_coords(xi, k - R, wm1, &i0k, &w0, &w1w);
      const int dlt = i0k - (wbase + EXT + k);  // -1, 0 or +1 (the fp32 round trip can cross an integer)
      const float a0 = g[k] * w0, a1 = g[k] * w1w;
      tw[k] += dlt < 0 ? a0 : 0.f;
      tw[k + 1] += dlt < 0 ? a1 : (dlt == 0 ? a0 : 0.f);
      tw[k + 2] += dlt == 0 ? a1 : (dlt > 0 ? a0 : 0.f);
      tw[k + 3] += dlt > 0 ? a1 : 0.f;
    }
  }
  float cur[NW];
  float* win = row + (wbase + PAD) * S;
#pragma unroll
  for (int m = 0; m < NW; ++m) cur[m] = win[m * S];
#pragma unroll
  for (int m = 0; m < NW; ++m) win[m * S] = cur[m] + tw[m];
}

// ---------------------------------------------------------------------------------------------
// Fused adjoint of up to RC_MAX_FUSED_LOOKUPS lookups, written as the DENSE gradient of level 0:
//   dC[bh][w1][w2] = 1/sqrt(D) * sum over lookups, levels i, taps of  2^-i * (tap gradient landing on w2 >> i)
// in the feature type, rows `ldc` long (zero beyond W2).  This is what rc_lookup_bwd (once per
// lookup, into the f32 gradient pyramid) followed by the unskew/fold step of rc_build_bwd
// produces, without the pyramid: a workgroup owns TP pixels of one image row and keeps their
// per-level gradient rows in LDS (f32).  Thread (level i, pixel p) owns row (i, p): it walks all
// lookups, adds each one's 2r+2 window entries into its row with plain LDS read-add-write (no
// atomics: nobody else touches the row; the next lookup's gradient planes are in flight
// meanwhile), then all threads fold the levels and write each dense row once.
#ifdef RC_LAB_ADJ_STAMPS  // lab: s_memtime per wave at the kernel's phases (tools/lab_adj_stamps.py)
}  // namespace
constexpr int RC_ADJ_NSTAMP = 16;
__device__ unsigned long long rc_lab_adj_stamps[8192 * 4 * RC_ADJ_NSTAMP];
extern "C" int rc_lab_adj_read_stamps(unsigned long long* dst, size_t n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rc_lab_adj_stamps), n * sizeof(unsigned long long));
}
namespace {
#define RC_ASTAMP(slot)                                                                      \
  do {                                                                                       \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192 && (slot) < RC_ADJ_NSTAMP)              \
      rc_lab_adj_stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RC_ADJ_NSTAMP + (slot)] = \
          __builtin_amdgcn_s_memtime();                                                      \
  } while (0)
#else
#define RC_ASTAMP(slot) do {} while (0)
#endif

struct adj_lists {
  const float* coords[RC_MAX_FUSED_LOOKUPS];
  const float* gout[RC_MAX_FUSED_LOOKUPS];
};

struct adj_dims {
  int B, H, W1, W2, L, ldc, n, cc;
  int width[RC_MAX_LEVELS];
  int base[RC_MAX_LEVELS];  // LDS offset of level i (floats) = TP * sum over the lower levels of (width + 2 * PAD)
};

typedef const float __attribute__((address_space(1)))* rc_gptr;  // a pointer known to be global memory

template <typename T, int R, int MODE, int TP>
__global__ __launch_bounds__(RC_MAX_LEVELS * TP) void adjoint_dense_kernel(adj_lists lists, adj_dims d,
                                                                           T* __restrict__ dC, float scale) {
  using WI = win_info<4, R, MODE>;
  constexpr int NT = WI::NT, EXT = WI::EXT, NW = WI::NW;
  constexpr int NTHR = RC_MAX_LEVELS * TP;
  extern __shared__ __attribute__((aligned(16))) float acc[];

  // Workgroups go to the 8 XCDs round-robin: each XCD takes a contiguous eighth of the (row, tile) list, so the
  // tiles of an image row, whose 128-byte pieces of a plane share lines (a row is not a whole number of
  // lines), meet in one L2, and what an XCD asks of memory is runs of neighbouring rows.  Reading this input
  // alone: 3.7 TB/s in plain order, 5.2 TB/s in this one (tools/lab_adj_pattern.hip).
  const int tiles = (d.W1 + TP - 1) / TP;
  const int n_tiles = d.B * d.H * tiles;
#ifdef RC_LAB_ADJ_PLAIN_ORDER
  const int id = blockIdx.x;
#else
  const int id = (int)(blockIdx.x & 7) * ((n_tiles + 7) >> 3) + (int)(blockIdx.x >> 3);
#endif
  if (id >= n_tiles) return;
  const int bh = id / tiles;
  const int c0 = (id - bh * tiles) * TP;
  const int b = bh / d.H, h = bh - b * d.H;
  const int p = threadIdx.x % TP, lvl = threadIdx.x / TP;
  const int w1 = c0 + p;
  const int64_t HW = (int64_t)d.H * d.W1;
  const int L = d.L;
  constexpr int PAD = adj_pad<R, MODE>::value;
  const int total = d.base[L - 1] + TP * (d.width[L - 1] + 2 * PAD);  // floats of LDS in use

  RC_ASTAMP(0);
  // selects, not d.width[lvl]: indexing the kernel argument by a lane's level is a vector memory load, and the
  // compiler then waits for ALL loads in flight wherever the loop below reads the value
  // (the empty asm keeps the compiler from turning the selects back into that load)
  auto of_level = [&](const int (&a)[RC_MAX_LEVELS]) {
    int v = 0;
#pragma unroll
    for (int i = 0; i < RC_MAX_LEVELS; ++i) {
      int ai = a[i];
      asm volatile("" : "+s"(ai));
      v = lvl == i ? ai : v;
    }
    return v;
  };
  const bool active = w1 < d.W1 && lvl < L;
  const int Wi = of_level(d.width);
  const float inv = 1.0f / (float)(1 << lvl);
  const float wm1 = (float)(Wi - 1);
  float* row = acc + of_level(d.base) + p;  // cell w2 of this thread's row is row[(w2 + PAD) * TP]
  // wave-uniform bases + 32-bit lane offsets (computed once): a fetch costs no address arithmetic
  const int64_t cu = uniform(((int64_t)b * d.cc) * HW + (int64_t)h * d.W1 + c0);
  const int64_t gu = uniform(((int64_t)b * L * NT) * HW + (int64_t)h * d.W1 + c0);
  uint32_t voff[NT];  // in bytes (< 2^31, checked by the caller): uniform base + 32-bit lane offset is one instruction
#pragma unroll
  for (int k = 0; k < NT; ++k) voff[k] = ((uint32_t)(lvl * NT + k) * (uint32_t)HW + (uint32_t)p) * 4u;
  // PD lookups' planes are in flight at any time: with the pixel tile bounded by LDS only a few
  // waves fit on a CU, so the latency has to be covered from inside the wave
#ifndef RC_LAB_ADJ_PD
#define RC_LAB_ADJ_PD 4
#endif
#ifndef RC_LAB_ADJ_ABLATE
#define RC_LAB_ADJ_ABLATE 0
#endif
#ifndef RC_LAB_ADJ_LOAD_AUX  // lab: cache policy of the gradient-plane loads (0 plain, 2 non-temporal, 1 / 3 with sc0)
#define RC_LAB_ADJ_LOAD_AUX 0
#endif
  constexpr int PD = RC_LAB_ADJ_PD;
  float xs[PD], gs[PD][NT];
  // buffer loads: descriptor from the uniform base (scalar work), the lane's precomputed byte offset as it is -- a
  // global load would spend a 64-bit vector add per plane on the address.  Every fetch is unconditional (a load
  // under a branch would make the compiler wait for all loads at every join); past the last lookup the
  // descriptor's size is 0, so the loads touch no memory and return at once.
  const int last = d.n - 1;
  auto fetch = [&](int t, int slot) {  // t is block-uniform: the list entries come through scalar loads
    const int size = t <= last ? 0x7fffffff : 0;
    const int tc = min(t, last);
    const __amdgpu_buffer_rsrc_t rc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lists.coords[tc] + cu), 0, size, 0x00020000);
    xs[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)((uint32_t)p * 4u), 0, 0));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lists.gout[tc] + gu), 0, size, 0x00020000);
#pragma unroll
    for (int k = 0; k < NT; ++k)
      gs[slot][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff[k], 0, RC_LAB_ADJ_LOAD_AUX));
  };
  float lab_sum = 0.f;
  // the first PD lookups are asked for before the rows are cleared: their latency passes behind the clearing
  if (active) {
#pragma unroll
    for (int j = 0; j < PD; ++j) {  // in the loop's order (slot by slot), or the loop's first wait has to be for everything
      fetch(j, j);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  typedef float f4z __attribute__((ext_vector_type(4)));
  for (int e = threadIdx.x; e < (total + 3) >> 2; e += NTHR) reinterpret_cast<f4z*>(acc)[e] = (f4z){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  RC_ASTAMP(1);

  if (active) {
    for (int t0 = 0; t0 < d.n; t0 += PD) {
      RC_ASTAMP(2 + t0 / PD);
#pragma unroll
      for (int j = 0; j < PD; ++j) {
        const int t = t0 + j;
        // consume the slot, THEN refill it: issued the other way round, the new planes land in fresh registers
        // that are copied into the slot's at the end of every pass of the loop, after a wait for all of them
#if RC_LAB_ADJ_ABLATE & 1  // lab: the loads alone (their sum goes to the sink cell once)
        for (int k = 0; k < NT; ++k) lab_sum += gs[j][k] + xs[j];
#else
        if (t < d.n) adjoint_accumulate<R, MODE, TP>(row, Wi, xs[j] * inv, wm1, gs[j]);
#endif
        __builtin_amdgcn_sched_barrier(0);
#if RC_LAB_ADJ_ABLATE & 2  // lab: no loads of the gradient planes
        xs[j] = (((rc_gptr)lists.coords[min(t + PD, last)]) + cu)[(uint32_t)p];
#else
        fetch(t + PD, j);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if RC_LAB_ADJ_ABLATE & 1
    row[0] = lab_sum;
#endif
  }
  RC_ASTAMP(12);
  __syncthreads();
  RC_ASTAMP(13);

  // fold the levels (a pooled cell spreads 2^-i of its gradient over its 2^i children) and write the
  // dense rows, 8 consecutive w2 per thread
  const int segs = d.ldc >> 3;
#if RC_LAB_ADJ_ABLATE & 4  // lab: no fold, no stores
  if (scale != 12345.f) return;
#endif
  for (int e = threadIdx.x; e < TP * segs; e += NTHR) {
    const int pp = e % TP, w20 = (e / TP) * 8;  // neighbouring lanes: neighbouring pixels (distinct LDS banks)
    if (c0 + pp >= d.W1) continue;
    // 8 + 4 + 2 + 1 independent LDS reads; levels the pyramid does not have come with width 0
    float part[RC_MAX_LEVELS][8];
#pragma unroll
    for (int i = 0; i < RC_MAX_LEVELS; ++i) {
      const float* rw = acc + d.base[i] + pp;
      const int wi0 = w20 >> i;
      const float wgt = scale / (float)(1 << i);
#pragma unroll
      for (int r = 0; r < (8 >> i); ++r) {
        const int wi = wi0 + r;
        const float val = rw[(min(wi, d.width[i] - 1) + PAD) * TP];
        part[i][r] = wi < d.width[i] ? wgt * val : 0.f;
      }
    }
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float sum = ((part[0][r] + part[1][r >> 1]) + part[2][r >> 2]) + part[3][r >> 3];
      v[r] = w20 + r < d.W2 ? sum : 0.f;
    }
    T* dst = dC + ((int64_t)bh * d.W1 + c0 + pp) * d.ldc + w20;
    if constexpr (sizeof(T) == 4) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      reinterpret_cast<f4v*>(dst)[0] = (f4v){v[0], v[1], v[2], v[3]};
      reinterpret_cast<f4v*>(dst)[1] = (f4v){v[4], v[5], v[6], v[7]};
    } else {
      typedef float f8v __attribute__((ext_vector_type(8)));
      typedef T t8v __attribute__((ext_vector_type(8)));
      const f8v f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
      *reinterpret_cast<t8v*>(dst) = __builtin_convertvector(f, t8v);
    }
  }
  RC_ASTAMP(14);
}

template <typename T, int R, int MODE>
int launch_adjoint_dense(const adj_lists& lists, adj_dims d, void* dC, float scale, hipStream_t st) {
  // the largest pixel tile whose per-level rows fit in 80 KiB of LDS
  constexpr int PAD = adj_pad<R, MODE>::value;
  int sum_pitch = 0;  // cells per pixel: every level's row and its padding
  for (int i = 0; i < d.L; ++i) sum_pitch += d.width[i] + 2 * PAD;
  int tp = 64;
  while (tp > 8 && (int64_t)tp * sum_pitch * 4 > 80 * 1024) tp >>= 1;
  const int64_t lds = ((int64_t)tp * sum_pitch * 4 + 15) & ~(int64_t)15;
  RC_REQUIRE(lds <= 159 * 1024, RC_EUNSUPPORTED, "W2=%d is too wide for the fused adjoint", d.W2);
  int off = 0;
  for (int i = 0; i < d.L; ++i) {
    d.base[i] = off;
    off += tp * (d.width[i] + 2 * PAD);
  }
  const int64_t blocks = ((int64_t)d.B * d.H * ((d.W1 + tp - 1) / tp) + 7) / 8 * 8;  // a whole number per XCD
  RC_REQUIRE(blocks < (1ll << 31), RC_EINVAL, "too many pixels");
#define RC_ADJ(TPV)                                                                                       \
  case TPV: {                                                                                             \
    static bool attr_set[64] = {};                                                                        \
    int dev = 0;                                                                                          \
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {                                       \
      rc_set_error("no current HIP device");                                                              \
      return RC_ENODEVICE;                                                                                \
    }                                                                                                     \
    if (!attr_set[dev]) {                                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&adjoint_dense_kernel<T, R, MODE, TPV>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);         \
      if (e != hipSuccess) {                                                                              \
        rc_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));                                    \
        return (int)e;                                                                                    \
      }                                                                                                   \
      attr_set[dev] = true;                                                                               \
    }                                                                                                     \
    hipLaunchKernelGGL((adjoint_dense_kernel<T, R, MODE, TPV>), dim3((unsigned)blocks),                   \
                       dim3(RC_MAX_LEVELS * TPV), (size_t)lds, st, lists, d, (T*)dC, scale);              \
    break;                                                                                                \
  }
  switch (tp) {
    RC_ADJ(64)
    RC_ADJ(32)
    RC_ADJ(16)
    RC_ADJ(8)
  }
#undef RC_ADJ
  return rc_check_launch("rc_lookup_bwd_dense");
}

template <typename T, int R>
int launch_adjoint_dense_mode(const adj_lists& lists, const adj_dims& d, void* dC, float scale, int mode,
                              hipStream_t st) {
  return mode == RC_COORD_EXACT ? launch_adjoint_dense<T, R, RC_COORD_EXACT>(lists, d, dC, scale, st)
                                : launch_adjoint_dense<T, R, RC_COORD_GRIDSAMPLE>(lists, d, dC, scale, st);
}

template <typename T>
int dispatch_adjoint_dense(const adj_lists& lists, const adj_dims& d, void* dC, float scale, int radius, int mode,
                           hipStream_t st) {
  switch (radius) {
    case 1: return launch_adjoint_dense_mode<T, 1>(lists, d, dC, scale, mode, st);
    case 2: return launch_adjoint_dense_mode<T, 2>(lists, d, dC, scale, mode, st);
    case 3: return launch_adjoint_dense_mode<T, 3>(lists, d, dC, scale, mode, st);
    case 4: return launch_adjoint_dense_mode<T, 4>(lists, d, dC, scale, mode, st);
    case 5: return launch_adjoint_dense_mode<T, 5>(lists, d, dC, scale, mode, st);
    case 6: return launch_adjoint_dense_mode<T, 6>(lists, d, dC, scale, mode, st);
    case 7: return launch_adjoint_dense_mode<T, 7>(lists, d, dC, scale, mode, st);
    case 8: return launch_adjoint_dense_mode<T, 8>(lists, d, dC, scale, mode, st);
  }
  rc_set_error("radius %d outside 1..%d", radius, RC_MAX_RADIUS);
  return RC_EUNSUPPORTED;
}

// Outputs beyond this size cannot stay in the 256 MiB Infinity Cache until their consumer reads them: they
// are written with non-temporal stores (KITTI, 64 pairs: 104 -> 94 us).  Non-temporal stores make the kernel
// faster on smaller outputs too (8 pairs: 11.1 -> 9.7 us) but the consumer then pays more than that
// (lookup + cast: 20.0 -> 20.9 us): those use agent-scope stores instead (store_taps).
#ifndef RC_LAB_LK_NT_BYTES  // lab builds move the threshold (0: always non-temporal; huge: never)
#define RC_LAB_LK_NT_BYTES (160ll << 20)
#endif
constexpr int64_t LK_NT_STORE_BYTES = RC_LAB_LK_NT_BYTES;

template <typename VT, int R, int MODE, typename OT>
int launch_fwd_mode(const void* vol, rc_dev_layout d, const float* coords, int cc, OT* out, hipStream_t st,
                    const lk_update& upd) {
  using PI = pair_info<vt_info<VT>::RB, R, MODE>;
  set_row_bias(d, vt_info<VT>::RB, PI::NW2);
  const int chunks = (d.H * d.W1 + 63) / 64;
  const int n_items = (int)((int64_t)d.B * chunks);
  int blocks = (n_items + (LK_THREADS / 64) - 1) / (LK_THREADS / 64);
  int threads = LK_THREADS;
  lk_update updx = upd;
  updx.wgs_per_xcd = 0;
  if (blocks <= LK_ONE_ROUND_WGS) {  // see the kernel: an eighth of the pixels per XCD
    // ... in workgroups of ONE wave: 3 744 of them spread over 256 CUs more evenly than 936 of four (KITTI, 8 pairs:
    // 8.56 -> 8.48 us); a launch of several rounds is faster in fours (64 pairs: 86.5 against 90.1 us)
    threads = 64;
    updx.wgs_per_xcd = (n_items + 7) / 8;
    blocks = updx.wgs_per_xcd * 8;
  }
  const bool nts = (int64_t)d.B * d.L * (2 * R + 1) * d.H * d.W1 * (int64_t)sizeof(OT) > LK_NT_STORE_BYTES;
#define RC_LK(ALL4, NTS)                                                                                       \
  hipLaunchKernelGGL((lookup_fwd_kernel<VT, R, MODE, ALL4, NTS, OT>), dim3(blocks), dim3(threads), 0, st,     \
                     (const VT*)vol, d, coords, cc, out, chunks, n_items, updx)
  if (R == 4 && d.L == 4) {  // the model's configuration: all levels' loads in flight together
    if (nts) RC_LK((R == 4), true);
    else RC_LK((R == 4), false);
  } else {
    if (nts) RC_LK(false, true);
    else RC_LK(false, false);
  }
#undef RC_LK
  return rc_check_launch("rc_lookup_fwd");
}

template <typename VT, int R, typename OT>
int launch_fwd(const void* vol, const rc_dev_layout& d, const float* coords, int cc, OT* out, int mode,
               hipStream_t st, const lk_update& upd) {
  return mode == RC_COORD_EXACT ? launch_fwd_mode<VT, R, RC_COORD_EXACT, OT>(vol, d, coords, cc, out, st, upd)
                                : launch_fwd_mode<VT, R, RC_COORD_GRIDSAMPLE, OT>(vol, d, coords, cc, out, st, upd);
}

template <typename VT, typename OT>
int dispatch_fwd_radius(const void* vol, const rc_dev_layout& d, const float* coords, int cc, OT* out,
                        int radius, int mode, hipStream_t st, const lk_update& upd) {
  switch (radius) {
    case 1: return launch_fwd<VT, 1, OT>(vol, d, coords, cc, out, mode, st, upd);
    case 2: return launch_fwd<VT, 2, OT>(vol, d, coords, cc, out, mode, st, upd);
    case 3: return launch_fwd<VT, 3, OT>(vol, d, coords, cc, out, mode, st, upd);
    case 4: return launch_fwd<VT, 4, OT>(vol, d, coords, cc, out, mode, st, upd);
    case 5: return launch_fwd<VT, 5, OT>(vol, d, coords, cc, out, mode, st, upd);
    case 6: return launch_fwd<VT, 6, OT>(vol, d, coords, cc, out, mode, st, upd);
    case 7: return launch_fwd<VT, 7, OT>(vol, d, coords, cc, out, mode, st, upd);
    case 8: return launch_fwd<VT, 8, OT>(vol, d, coords, cc, out, mode, st, upd);
  }
  rc_set_error("radius %d outside 1..%d", radius, RC_MAX_RADIUS);
  return RC_EUNSUPPORTED;
}

template <int R, int MODE>
int launch_bwd_mode(const float* coords, int cc, const float* gout, float* gvol, rc_dev_layout d, hipStream_t st) {
  using WI = win_info<4, R, MODE>;
  set_row_bias(d, 4, WI::NW);
  const int chunks = (d.W1 + 63) / 64;
  const int n_items = (int)((int64_t)d.B * d.H * chunks);
  const int blocks = (n_items + (LK_THREADS / 64) - 1) / (LK_THREADS / 64);
  hipLaunchKernelGGL((lookup_bwd_kernel<R, MODE>), dim3(blocks), dim3(LK_THREADS), 0, st, coords, cc, gout, gvol,
                     d, chunks, n_items);
  return rc_check_launch("rc_lookup_bwd");
}

template <int R>
int launch_bwd(const float* coords, int cc, const float* gout, float* gvol, const rc_dev_layout& d,
               int mode, hipStream_t st) {
  return mode == RC_COORD_EXACT ? launch_bwd_mode<R, RC_COORD_EXACT>(coords, cc, gout, gvol, d, st)
                                : launch_bwd_mode<R, RC_COORD_GRIDSAMPLE>(coords, cc, gout, gvol, d, st);
}

}  // namespace

// out_vt: the output is in the volume's own type (rc_lookup_fwd_vt) instead of fp32
static int lookup_fwd_any(const void* vol, const rc_layout* lay, const float* coords, int coords_channels,
                          void* out, bool out_vt, int radius, int coord_mode, void* stream, const lk_update& upd) {
  int rc = check_common(lay, vol, coords, out, coords_channels, radius, coord_mode);
  if (rc) return rc;
  RC_REQUIRE(!lay->full, RC_EINVAL, "rc_lookup_fwd takes a forward pyramid (rc_pyramid_layout)");
  const rc_dev_layout d = rc_to_dev(lay);
  hipStream_t st = (hipStream_t)stream;
  switch (lay->dtype) {
    case RC_F32:
      return dispatch_fwd_radius<float, float>(vol, d, coords, coords_channels, (float*)out, radius, coord_mode, st, upd);
    case RC_F16:
      if (out_vt)
        return dispatch_fwd_radius<rc_half, rc_half>(vol, d, coords, coords_channels, (rc_half*)out, radius, coord_mode, st, upd);
      return dispatch_fwd_radius<rc_half, float>(vol, d, coords, coords_channels, (float*)out, radius, coord_mode, st, upd);
    case RC_BF16:
      if (out_vt)
        return dispatch_fwd_radius<rc_bf16, rc_bf16>(vol, d, coords, coords_channels, (rc_bf16*)out, radius, coord_mode, st, upd);
      return dispatch_fwd_radius<rc_bf16, float>(vol, d, coords, coords_channels, (float*)out, radius, coord_mode, st, upd);
  }
  rc_set_error("unknown pyramid dtype %d", lay->dtype);
  return RC_EUNSUPPORTED;
}

extern "C" int rc_lookup_fwd(const void* vol, const rc_layout* lay, const float* coords, int coords_channels,
                             float* out, int radius, int coord_mode, void* stream) {
  return lookup_fwd_any(vol, lay, coords, coords_channels, out, false, radius, coord_mode, stream, lk_update{});
}

extern "C" int rc_lookup_fwd_vt(const void* vol, const rc_layout* lay, const float* coords, int coords_channels,
                                void* out, int radius, int coord_mode, void* stream) {
  return lookup_fwd_any(vol, lay, coords, coords_channels, out, true, radius, coord_mode, stream, lk_update{});
}

extern "C" int rc_lookup_fwd_update(const void* vol, const rc_layout* lay, const float* coords,
                                    int coords_channels, const float* delta_flow, float* coords_out,
                                    float* flow_out, float* out, int radius, int coord_mode, void* stream) {
  RC_REQUIRE(delta_flow && coords_out && flow_out, RC_EINVAL, "null pointer argument");
  const lk_update upd = {delta_flow, coords_out, flow_out};
  return lookup_fwd_any(vol, lay, coords, coords_channels, out, false, radius, coord_mode, stream, upd);
}

extern "C" int rc_lookup_bwd(const float* coords, int coords_channels, const float* grad_out,
                             float* grad_vol, const rc_layout* glay, int radius, int coord_mode,
                             void* stream) {
  int rc = check_common(glay, grad_vol, coords, grad_out, coords_channels, radius, coord_mode);
  if (rc) return rc;
  RC_REQUIRE(glay->dtype == RC_F32 && glay->full, RC_EINVAL,
             "the gradient pyramid must use an rc_grad_pyramid_layout");
  const rc_dev_layout d = rc_to_dev(glay);
  hipStream_t st = (hipStream_t)stream;
  switch (radius) {
    case 1: return launch_bwd<1>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 2: return launch_bwd<2>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 3: return launch_bwd<3>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 4: return launch_bwd<4>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 5: return launch_bwd<5>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 6: return launch_bwd<6>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 7: return launch_bwd<7>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
    case 8: return launch_bwd<8>(coords, coords_channels, grad_out, grad_vol, d, coord_mode, st);
  }
  return RC_EUNSUPPORTED;
}

extern "C" int64_t rc_dense_grad_ld(const rc_layout* glay) {
  if (!glay || rc_validate_layout(glay)) return -1;
  return (glay->W2 + 31) / 32 * 32;
}

extern "C" int rc_lookup_bwd_dense(const float* const* coords_list, const float* const* grad_out_list,
                                   int n_lookups, int coords_channels, void* dense_grad, const rc_layout* glay,
                                   int64_t D, int out_dtype, int radius, int coord_mode, void* stream) {
  RC_REQUIRE(coords_list && grad_out_list && dense_grad && glay, RC_EINVAL, "null pointer argument");
  RC_REQUIRE(n_lookups >= 1 && n_lookups <= RC_MAX_FUSED_LOOKUPS, RC_EUNSUPPORTED,
             "n_lookups %d outside 1..%d", n_lookups, RC_MAX_FUSED_LOOKUPS);
  int rc = check_common(glay, dense_grad, coords_list[0], grad_out_list[0], coords_channels, radius, coord_mode);
  if (rc) return rc;
  RC_REQUIRE(D >= 1 && D < (1 << 20), RC_EINVAL, "bad feature depth D=%lld", (long long)D);
  RC_REQUIRE((int64_t)glay->L * (2 * radius + 1) * glay->H * glay->W1 < (1ll << 29), RC_EINVAL,
             "one image's gradient planes exceed the 32-bit lane offset range");
  adj_lists lists;
  for (int t = 0; t < RC_MAX_FUSED_LOOKUPS; ++t) {
    const int s = t < n_lookups ? t : 0;
    RC_REQUIRE(coords_list[s] && grad_out_list[s], RC_EINVAL, "null pointer in the lookup lists");
    lists.coords[t] = coords_list[s];
    lists.gout[t] = grad_out_list[s];
  }
  adj_dims d;
  d.B = (int)glay->B; d.H = (int)glay->H; d.W1 = (int)glay->W1; d.W2 = (int)glay->W2; d.L = glay->L;
  d.ldc = (int)((glay->W2 + 31) / 32 * 32);
  d.n = n_lookups;
  d.cc = coords_channels;
  for (int i = 0; i < RC_MAX_LEVELS; ++i) {
    d.width[i] = i < glay->L ? glay->width[i] : 0;  // width 0: the fold skips the level
    d.base[i] = 0;
  }
  const float scale = (float)(1.0 / sqrt((double)D));
  hipStream_t st = (hipStream_t)stream;
  switch (out_dtype) {
    case RC_F32: return dispatch_adjoint_dense<float>(lists, d, dense_grad, scale, radius, coord_mode, st);
    case RC_F16: return dispatch_adjoint_dense<rc_half>(lists, d, dense_grad, scale, radius, coord_mode, st);
    case RC_BF16: return dispatch_adjoint_dense<rc_bf16>(lists, d, dense_grad, scale, radius, coord_mode, st);
  }
  rc_set_error("unknown out_dtype %d", out_dtype);
  return RC_EUNSUPPORTED;
}

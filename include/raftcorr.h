/*
 * raftcorr.h -- C-ABI of libraftcorr_hip.so: the MI355X (gfx950) correlation hot path of
 * RAFT-Stereo (volume build + pooled pyramid, per-iteration lookup fwd/bwd, build adjoint,
 * on-the-fly "alt" lookup) and the pointwise work of a GRU iteration around it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates,
 *     frees or keeps a pointer past the call.  Its only state is a thread-local error string
 *     and, per device, the flags "this kernel's dynamic-LDS limit has been raised"
 *     (hipFuncSetAttribute is idempotent, so racing first calls are harmless);
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *   - every function returns 0 on success, a negative RC_E* code for argument errors, or a
 *     positive hipError_t for launch errors; rc_last_error_string() explains the last
 *     failure of the calling thread;
 *   - the functions are re-entrant and may be called concurrently from several host
 *     threads (nn.DataParallel-style) on different devices/streams; the caller makes the
 *     right device current before the call.
 *
 * Reference interfaces replaced (paths inside the upstream RAFT-Stereo tree):
 *   rc_build_pyramid   core/corr.py:111-125 + :148-156   CorrBlock1D.__init__ / .corr
 *                      core/corr.py:32-42  + :53-61      CorrBlockFast1D.__init__ / .corr
 *   rc_lookup_fwd      core/corr.py:127-146 (reg, grid_sample coordinates) and
 *                      core/corr.py:44-51 -> sampler/sampler.cpp:24-32 ->
 *                      sampler/sampler_kernel.cu:19-60,107-136 (reg_cuda), all levels at once
 *   rc_lookup_fwd_vt   the same with the taps in the volume's type: what core/corr.py:44-51 returns under
 *                      mixed precision (sampler/sampler_kernel.cu:122-124 allocates volume.options())
 *   rc_lookup_bwd      core/corr.py:24-29 -> sampler/sampler.cpp:34-45 ->
 *                      sampler/sampler_kernel.cu:63-105,138-166
 *   rc_build_bwd       autograd of core/corr.py:119-125 (avg_pool2d chain) and :154-156
 *   rc_sampler_fwd/bwd sampler/sampler.cpp:48-51 `corr_sampler.forward/backward` on the
 *                      caller's own row-major [B,H,W1,W2] single-level volume
 *   rc_alt_prepare,
 *   rc_alt_fwd         core/corr.py:64-107 PytorchAlternateCorrBlock1D (volume never built)
 * and, on either side of the lookup (SURVEY.md section 8 f):
 *   rc_lookup_convc1_fwd, rc_convc1_bwd     core/update.py:71,78 on core/raft_stereo.py:110
 *   rc_lookup_fwd_update,
 *   rc_lookup_convc1_fwd_update             core/raft_stereo.py:109-111,120-123
 *   rc_gru_gate_r, rc_gru_update (+ _bwd)   core/update.py:23-32
 *   rc_pool2x, rc_interp_bilinear           core/update.py:87-95
 *   rc_upsample_flow_fwd/bwd                core/raft_stereo.py:55-67
 *   rc_sequence_loss_fwd/bwd                train_stereo.py:35-70
 */
#ifndef RAFTCORR_H
#define RAFTCORR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_VERSION 200          /* 0.2.0: forward pyramids keep the even levels only */
#define RC_MAX_LEVELS 4
#define RC_MAX_RADIUS 8
#define RC_MAX_FUSED_LOOKUPS 32  /* lookups one rc_lookup_bwd_dense call can take */
#define RC_CONVC1_OUT 64         /* output channels of the motion encoder's convc1 (core/update.py:71) */
#define RC_CONVC1_MAX_RADIUS 7   /* rc_lookup_convc1_fwd: a level's 2r+1 taps must fit 16 MFMA K slots */
#define RC_SAMPLER_MAX_RADIUS 7  /* rc_sampler_fwd/bwd: one kernel instantiation per radius (forward: 2r+2 volume entries per lane in registers) */
#define RC_ALT_MAX_RADIUS 6      /* rc_alt_fwd: the union of 16 pixels' tap windows must fit the wave's MFMA tiles */

/* element types of feature maps / volumes */
enum { RC_F32 = 0, RC_F16 = 1, RC_BF16 = 2 };

/* how a tap's sampling position is derived from coords (see oracle/corr_oracle.py tap_coords) */
enum {
  RC_COORD_EXACT = 0,      /* floor + lerp of sampler_kernel.cu:42-57 (reg_cuda semantics)   */
  RC_COORD_GRIDSAMPLE = 1  /* reg semantics: 2x/(W-1)-1 then grid_sample(align_corners=True)  */
};

/* error codes (negative); positive return values are hipError_t */
enum {
  RC_OK = 0,
  RC_EINVAL = -1,       /* bad shape / null pointer / misaligned pointer */
  RC_EUNSUPPORTED = -2, /* dtype / radius / level count outside the compiled range */
  RC_ENODEVICE = -3
};

/*
 * Private layout of the correlation pyramid ("group-skewed blocks, even levels stored"):
 * the column of pixel (b,h,w1) at level i -- its width[i] correlations against w2 = 0.. --
 * is cut into groups of rb = 16 / element-size consecutive w2, one 16-byte block each.  Group
 * g = w2 / rb of that pixel is stored in block row
 *     sb = (g - ((w1 >> i) / rb) + nb - 32 / rb) mod nb,          nb = widthp[i] / rb,
 * and element (b,h,w1,w2) of level i lives at
 *     offset[i] + (((b*H + h) * nb + sb) * pitch + w1) * rb + w2 % rb.
 * Consequences: (1) the taps of a lookup are a few whole blocks, and neighbouring pixels
 * with similar disparity read neighbouring blocks of the same block rows (16 bytes per lane,
 * long contiguous runs per wave); (2) the 4 consecutive w2 an MFMA lane owns after the build are
 * half (16-bit) or all (f32) of one block, so the build writes the pyramid straight from
 * registers in full 128-byte lines; (3) the wrap-around of a column sits at disparities of
 * about -32 level-i pixels.
 * A forward pyramid (rc_pyramid_layout) keeps only the EVEN levels in memory (stored[i] != 0):
 * entry j of level i+1 is the mean of entries 2j, 2j+1 of level i (core/corr.py:124), and the
 * 2r+2 entries of level i+1 a lookup needs lie over 4r+4 consecutive entries of level i that
 * contain the lookup's own level-i window, so the lookup kernels read that one run and derive
 * level i+1 in registers with the build's arithmetic (fp32 mean of the stored values, rounded to
 * the storage type).  The pyramid is 1.25 x level 0 instead of 1.875 x, and a lookup touches
 * two runs per pixel instead of four.  A gradient pyramid (rc_grad_pyramid_layout, always
 * RC_F32) stores every level.  Nothing outside this library and its Python wrapper may index
 * the pyramid.
 */
typedef struct rc_layout {
  int64_t B, H, W1, W2;
  int32_t L;                        /* levels of the pyramid (= levels ever read, corr.py:133-134) */
  int32_t pitch;                    /* W1 rounded up to a multiple of 64 pixels */
  int32_t rb;                       /* w2 per 16-byte block: 8 (f16/bf16) or 4 (f32) */
  int32_t dtype;                    /* element type the layout was made for */
  int32_t width[RC_MAX_LEVELS];     /* W2 >> i */
  int32_t widthp[RC_MAX_LEVELS];    /* width rounded up to a multiple of 8 */
  int32_t stored[RC_MAX_LEVELS];    /* level i has memory (forward pyramids: even levels only) */
  int32_t full;                     /* made by rc_grad_pyramid_layout: every level stored */
  int32_t reserved;
  int64_t offset[RC_MAX_LEVELS];    /* element offset of level i (0 when not stored) */
  int64_t total;                    /* elements in the whole pyramid */
} rc_layout;

int rc_version(void);
const char* rc_last_error_string(void);

/* Fills *out for a forward pyramid of `dtype` elements (even levels stored); RC_EINVAL if
 * W2 >> (L-1) < 1 or any size is non-positive. */
int rc_pyramid_layout(int64_t B, int64_t H, int64_t W1, int64_t W2, int L, int dtype,
                      rc_layout* out);
/* The float gradient pyramid rc_lookup_bwd accumulates into and rc_build_bwd folds (every level
 * stored, dtype RC_F32); also what rc_lookup_bwd_dense / rc_build_bwd_dense take for the geometry. */
int rc_grad_pyramid_layout(int64_t B, int64_t H, int64_t W1, int64_t W2, int L, rc_layout* out);

/*
 * Build all levels of the pyramid: C = f1^T f2 / sqrt(D) per image row, then the pairwise
 * mean chain, written in the private layout.  f1: [B,D,H,W1], f2: [B,D,H,W2] contiguous
 * NCHW of `in_dtype`; vol: lay->total elements of lay->dtype.
 * Arithmetic: RC_F32 inputs use the exact-f32 MFMA (v_mfma_f32_16x16x4_f32), RC_F16/RC_BF16
 * inputs the 16x16x32 f16/bf16 MFMA with f32 accumulation; levels >= 1 are means of the
 * already-rounded lower level (what avg_pool2d on a 16-bit volume does, corr.py:42).
 * Supported: (in,vol) = (F32,F32), (F16,F16), (BF16,BF16), (F16|BF16, F32).
 * Environment (read per call; cache policy and kernel choice only, the results are the same bits):
 * RC_BUILD_F1_POLICY=stream|plain -- stream: the pyramid is meant to stay in the Infinity Cache for the lookups
 * (f1 tiles fetched past the cache, pyramid written with plain stores); plain: the pyramid is streamed out
 * (default: by the pyramid's size, stream up to 200 MB); RC_BUILD_NO_DMA=1 -- the kernel without LDS-DMA;
 * RC_BUILD_ROW_ORDER=contiguous|interleaved -- image rows of one XCD (default: contiguous for 16-bit features).
 * RC_BUILD_F1_TILES=line|row -- f1 tiles start on 128-byte lines of memory or at the row's first column
 *   (default: line where the k planes are whole lines apart and the rows are not).
 */
int rc_build_pyramid(const void* f1, const void* f2, void* vol, const rc_layout* lay,
                     int64_t D, int in_dtype, void* stream);
/* Diagnostic twin (no counterpart in the reference): same contract, but always runs the
 * one-thread-per-element kernels that rc_build_pyramid only falls back to for D > 256.
 * Used by the GPU tests as an on-device cross-check of the MFMA kernel. */
int rc_build_pyramid_naive(const void* f1, const void* f2, void* vol, const rc_layout* lay,
                           int64_t D, int in_dtype, void* stream);

/*
 * Lookup, all levels in one launch.
 * coords: float [B, coords_channels, H, W1] (channel 0 = x; other channels ignored);
 * out:    float [B, L*(2r+1), H, W1], channel = level*(2r+1) + tap  (corr.py:145-146).
 */
int rc_lookup_fwd(const void* vol, const rc_layout* lay, const float* coords,
                  int coords_channels, float* out, int radius, int coord_mode, void* stream);
/*
 * The same lookup with `out` in the VOLUME's type (lay->dtype; for an fp32 pyramid identical to
 * rc_lookup_fwd): what the reference's mixed-precision block returns -- CorrBlockFast1D's
 * CorrSampler allocates its result with volume.options() (core/corr.py:44-51,
 * sampler/sampler_kernel.cu:122-124), fp16 under autocast, where CorrBlock1D ends in .float()
 * (core/corr.py:146).  Every tap is computed in fp32 as in rc_lookup_fwd and rounded once.
 */
int rc_lookup_fwd_vt(const void* vol, const rc_layout* lay, const float* coords,
                     int coords_channels, void* out, int radius, int coord_mode, void* stream);

/*
 * SURVEY.md section 8(f) rank 2 -- the GRU iteration's coordinate glue folded into the lookup
 * (core/raft_stereo.py:120-123 of the previous iteration and :109-111 of this one):
 *     coords_out = coords + (delta_flow[:,0], 0)      delta_flow: float [B,2,H,W1], channel 1 ignored
 *     flow_out   = coords_out - coords0               coords0 = pixel grid (x = w1, y = h)
 *     out        = lookup(coords_out)
 * coords_out/flow_out: float [B,2,H,W1]; coords_out may alias coords when coords_channels == 2.
 * With coords_channels == 1 the y row of coords_out is the row index h.  Arithmetic is the
 * reference's fp32 add/subtract, so results are bit-identical to the separate steps.
 */
int rc_lookup_fwd_update(const void* vol, const rc_layout* lay, const float* coords,
                         int coords_channels, const float* delta_flow, float* coords_out,
                         float* flow_out, float* out, int radius, int coord_mode, void* stream);

/*
 * Adjoint of rc_lookup_fwd w.r.t. the pyramid.  grad_vol is a float pyramid (glay->dtype ==
 * RC_F32) that is ACCUMULATED into (+=): zero it once per forward pass, then call this once
 * per lookup.  Calls on one stream are ordered; no atomics are used (a pixel owns its column).
 */
int rc_lookup_bwd(const float* coords, int coords_channels, const float* grad_out,
                  float* grad_vol, const rc_layout* glay, int radius, int coord_mode,
                  void* stream);

/*
 * Adjoint of rc_build_pyramid: folds the level gradients into level 0 and contracts with the
 * feature maps: df1[b,d,h,w1] = sum_w2 dC f2 / sqrt(D), df2[b,d,h,w2] = sum_w1 dC f1 / sqrt(D).
 * df1/df2 have the dtype and shape of f1/f2.  `work` is scratch of rc_build_bwd_work_bytes().
 */
int64_t rc_build_bwd_work_bytes(const rc_layout* glay, int in_dtype);
int rc_build_bwd(const float* grad_vol, const rc_layout* glay, const void* f1, const void* f2,
                 void* df1, void* df2, int64_t D, int in_dtype, void* work, void* stream);

/*
 * The same adjoint for a whole forward pass at once.  rc_lookup_bwd_dense takes the coords and
 * grad_out of up to RC_MAX_FUSED_LOOKUPS lookups (HOST arrays of device pointers, shapes as in
 * rc_lookup_fwd) and writes the dense gradient of level 0, pooled levels folded in and scaled by
 * 1/sqrt(D): dense_grad[b*H + h][w1][w2] of `out_dtype`, rows rc_dense_grad_ld(glay) elements
 * long (zero beyond W2), B*H*W1*ld elements in all.  It replaces one rc_lookup_bwd per lookup
 * (autograd of corr.py:127-146 / sampler_kernel.cu:63-105 called once per GRU iteration) plus the
 * pyramid fold of rc_build_bwd, without ever touching a gradient pyramid.  rc_build_bwd_dense
 * then contracts it with the feature maps (autograd of corr.py:154-156); df1/df2 and dense_grad
 * have the feature dtype.  A thread owns the per-level gradient row of its pixel (in LDS, f32) and
 * adds the lookups in list order: no atomics, results are reproducible run to run.
 * Limits (RC_EUNSUPPORTED / RC_EINVAL beyond them): one image's gradient planes L*(2r+1)*H*W1 < 2^29
 * elements (rc_lookup_bwd_dense); one image's feature maps D*H*max(W1,W2) and one row's dense gradient
 * W1*ld below 2 GiB (rc_build_bwd, rc_build_bwd_dense: 32-bit buffer offsets inside an image).
 */
int64_t rc_dense_grad_ld(const rc_layout* glay);
int rc_lookup_bwd_dense(const float* const* coords_list, const float* const* grad_out_list,
                        int n_lookups, int coords_channels, void* dense_grad, const rc_layout* glay,
                        int64_t D, int out_dtype, int radius, int coord_mode, void* stream);
int rc_build_bwd_dense(const void* dense_grad, const rc_layout* glay, const void* f1, const void* f2,
                       void* df1, void* df2, int64_t D, int in_dtype, void* stream);

/*
 * SURVEY.md section 8(f) rank 1 -- lookup fused with the first layer of the motion encoder:
 *     out = relu(convc1(lookup(coords))),  convc1 = nn.Conv2d(L*(2r+1), 64, 1)  (core/update.py:71,78
 * applied to the result of core/raft_stereo.py:110).  The L*(2r+1)-channel tensor is never written.
 * rc_convc1_pack arranges the convolution weight -- float [64, L*(2r+1)] as nn.Conv2d stores it,
 * input channel = level*(2r+1) + tap -- into MFMA fragment order for a pyramid of `dtype`:
 * `packed` holds RC_CONVC1_OUT*64 elements of that type (call again whenever the weight changes).
 * rc_lookup_convc1_fwd: vol/lay/coords as in rc_lookup_fwd; bias float [64]; out [B,64,H,W1] in the
 * pyramid's dtype (RC_F16/RC_BF16: taps rounded to that type, f32 accumulation -- what autocast makes
 * of the convolution; RC_F32: f32 throughout); relu != 0 applies the activation.  radius <= 7.
 */
int rc_convc1_pack(const float* weight, int num_levels, int radius, int dtype, void* packed,
                   void* stream);
int rc_lookup_convc1_fwd(const void* vol, const rc_layout* lay, const float* coords,
                         int coords_channels, const void* packed_weight, const float* bias,
                         void* out, int radius, int coord_mode, int relu, void* stream);
/*
 * SURVEY.md section 8(f) ranks 1 + 2 in one launch -- a whole GRU iteration's glue in front of the fused
 * lookup (core/raft_stereo.py:120-123 of the previous iteration, :109-111 of this one, then core/update.py:78):
 *     coords_out = coords + (delta_flow[:,0], 0)      delta_flow: [B,2,H,W1] of delta_dtype (RC_F32/F16/BF16:
 *                                                     what the flow head returns, under autocast 16 bits),
 *                                                     promoted to fp32 as torch's `+` does; NULL: nothing added
 *     flow_out   = coords_out - coords0               coords0 = pixel grid (x = w1, y = h)
 *     out        = relu(convc1(lookup(coords_out)))   as rc_lookup_convc1_fwd
 * coords_out/flow_out: float [B,2,H,W1]; coords_out may alias coords when coords_channels == 2.  Bit-identical
 * to rc_lookup_fwd_update's coordinates followed by rc_lookup_convc1_fwd.
 */
int rc_lookup_convc1_fwd_update(const void* vol, const rc_layout* lay, const float* coords,
                                int coords_channels, const void* delta_flow, int delta_dtype,
                                float* coords_out, float* flow_out, const void* packed_weight,
                                const float* bias, void* out, int radius, int coord_mode, int relu,
                                void* stream);

/*
 * Adjoint of rc_lookup_convc1_fwd (autograd of core/update.py:78 on core/raft_stereo.py:110), so the fused
 * layer also trains.  The caller recomputes corr = rc_lookup_fwd(coords) (float [B,K,H,W], K = L*(2r+1))
 * right before the call -- the forward never stored it -- and passes the forward's output `act` (for the
 * ReLU mask; `relu` as in the forward) and its gradient, both [B,64,H,W] of `dtype`, with the float weight
 * [64,K] as nn.Conv2d stores it:
 *     grad_corr   float [B,K,H,W]  = W^T (grad_act where act > 0)   -> the `grad_out` of rc_lookup_bwd(_dense)
 *     grad_weight float [64,K], grad_bias float [64]                 (sums over all pixels)
 * `work`: rc_convc1_bwd_work_bytes() of scratch (transposed weights + float block partials, which are added
 * in double in a fixed order: no atomics, reproducible run to run).
 */
int64_t rc_convc1_bwd_work_bytes(void);
int rc_convc1_bwd(const float* corr, const void* act, const void* grad_act, const float* weight,
                  float* grad_corr, float* grad_weight, float* grad_bias, void* work, int64_t B,
                  int in_channels, int64_t HW, int dtype, int relu, void* stream);

/*
 * rc_convc1_bwd for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as
 * rc_convc1_bwd minus `work`: the scratch (transposed weights + block partials) is a file-scope device array
 * of the library, 4 539 392 bytes, which every GPU the library runs on holds until the process ends.  The
 * call allocates nothing (no hipMalloc / hipFree, no allocating library routine).  Limit: in_channels <= 68
 * (= RC_MAX_LEVELS * (2 * RC_MAX_RADIUS + 1), what the array is sized for); a wider input is refused with
 * RC_EUNSUPPORTED.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams,
 * two host threads) race; calls on one stream, or on different GPUs, do not.
 */
int rc_convc1_bwd_static(const float* corr, const void* act, const void* grad_act, const float* weight,
                         float* grad_corr, float* grad_weight, float* grad_bias, int64_t B, int in_channels,
                         int64_t HW, int dtype, int relu, void* stream);

/*
 * Drop-in for the reference's `corr_sampler` extension (sampler/sampler.cpp:48-51): one
 * level, the caller's row-major volume [B,H,W1,W2] of `dtype`, coords [B,coords_channels,H,W1]
 * float, corr / corr_grad [B,2r+1,H,W1] of `dtype`.  rc_sampler_bwd OVERWRITES volume_grad
 * (the reference returns a fresh zeros_like, sampler_kernel.cu:148).
 */
int rc_sampler_fwd(const void* volume, const float* coords, int coords_channels, void* corr,
                   int64_t B, int64_t H, int64_t W1, int64_t W2, int radius, int dtype,
                   void* stream);
int rc_sampler_bwd(const float* coords, int coords_channels, const void* corr_grad,
                   void* volume_grad, int64_t B, int64_t H, int64_t W1, int64_t W2, int radius,
                   int dtype, void* stream);

/*
 * On-the-fly ("alt") correlation.  rc_alt_prepare converts the feature maps once per image
 * pair into channel-last working copies: f1t [B,H,W1,Dp] and the pooled f2 pyramid
 * f2t level i [B,H,W2>>i,Dp] for the EVEN levels i (Dp = D rounded up to 32, zero padded),
 * both of `in_dtype`.  rc_alt_fwd then computes the same [B, L*(2r+1), H, W1] float output as
 * rc_lookup_fwd from banded MFMA products, without ever materialising the volume; an odd
 * level's correlations are the means of two of the level below (pooling is linear), so the
 * odd levels of f2 have no copy (f2_offset 0, nothing counted in f2_elems).
 * Environment (read per call; cache policy only, the results are the same bits):
 * RC_ALT_F1_POLICY=stream|plain -- rc_alt_fwd fetches f1t past the Infinity Cache or not (default: by
 * size, stream when f1t + f2t exceed 200 MB, so that f2t stays cached from one lookup to the next).
 * RC_ALT_PREPARE=elements -- rc_alt_prepare's kernel for arbitrary widths also where the 16-byte kernel applies.
 */
typedef struct rc_alt_layout {
  int64_t B, H, W1, W2, D, Dp;
  int32_t L;
  int32_t width[RC_MAX_LEVELS];
  int64_t f1_elems;                 /* elements of f1t */
  int64_t f2_offset[RC_MAX_LEVELS]; /* element offset of (even) level i inside f2t */
  int64_t f2_elems;                 /* elements of f2t: its even levels */
} rc_alt_layout;

int rc_alt_layout_init(int64_t B, int64_t D, int64_t H, int64_t W1, int64_t W2, int L,
                       rc_alt_layout* out);
int rc_alt_prepare(const void* f1, const void* f2, void* f1t, void* f2t,
                   const rc_alt_layout* lay, int in_dtype, void* stream);
int rc_alt_fwd(const void* f1t, const void* f2t, const rc_alt_layout* lay, int in_dtype,
               const float* coords, int coords_channels, float* out, int radius,
               int coord_mode, void* stream);

/*
 * SURVEY.md section 8(f) rank 3 -- the pointwise arithmetic of ConvGRU.forward (core/update.py:23-32).
 * The three convolutions stay with the caller; the library does what surrounds them, reading every
 * operand once:
 *     rc_gru_gate_r:  rh = sigmoid(convr_out + cr) * h                                (update.py:28-29)
 *     rc_gru_update:  h_out = (1 - z) * h + z * q,  z = sigmoid(convz_out + cz),
 *                                                   q = tanh(convq_out + cq)          (update.py:27,29,31)
 * All tensors are [N, C, H, W] of `dtype` with per_batch = C*H*W elements per image.  h and h_out are
 * contiguous; in the forward calls convX_out may be channel slices of a wider tensor (convz and convr read the
 * same input, so a caller may run them as one stacked convolution), cz/cr/cq likewise (the `.split` views of
 * core/raft_stereo.py:88), and rh may be a channel slice of the buffer the next convolution reads
 * ([r*h, x], update.py:29): those take the distance between images in elements.  h_out may alias h.
 * 16-bit types round after every elementary operation, as the chain of torch ops under autocast does.
 * The _bwd twins are the adjoints (gates recomputed from the saved pre-activations):
 *     rc_gru_update_bwd: grad_z_pre (for convz_out and cz), grad_q_pre (for convq_out and cq), grad_h = g (1-z)
 *     rc_gru_gate_r_bwd: grad_r_pre (for convr_out and cr), grad_h (+)= grad_rh * r  (accumulate_grad_h != 0
 *                        adds to what grad_h holds: h is used by both kernels)
 */
int rc_gru_gate_r(const void* convr_out, int64_t convr_batch_stride, const void* cr,
                  int64_t cr_batch_stride, const void* h, void* rh, int64_t rh_batch_stride, int64_t N,
                  int64_t per_batch, int dtype, void* stream);
int rc_gru_update(const void* convz_out, int64_t convz_batch_stride, const void* cz,
                  int64_t cz_batch_stride, const void* convq_out, int64_t convq_batch_stride,
                  const void* cq, int64_t cq_batch_stride, const void* h, void* h_out, int64_t N,
                  int64_t per_batch, int dtype, void* stream);
int rc_gru_update_bwd(const void* convz_out, const void* cz, int64_t cz_batch_stride,
                      const void* convq_out, const void* cq, int64_t cq_batch_stride, const void* h,
                      const void* grad_h_out, void* grad_z_pre, void* grad_q_pre, void* grad_h,
                      int64_t N, int64_t per_batch, int dtype, void* stream);
int rc_gru_gate_r_bwd(const void* convr_out, const void* cr, int64_t cr_batch_stride, const void* h,
                      const void* grad_rh, int64_t grad_rh_batch_stride, void* grad_r_pre,
                      void* grad_h, int accumulate_grad_h, int64_t N, int64_t per_batch, int dtype,
                      void* stream);

/*
 * The glue between the GRU scales (core/update.py:87-95, run every iteration): pool2x(x) =
 * avg_pool2d(x, 3, stride=2, padding=1) -> [N, C, (H-1)/2+1, (W-1)/2+1] (padding counted: always / 9), and
 * interp(x, dest) = bilinear resize to Ho x Wo with align_corners=True.  x: [N,C,H,W] contiguous of `dtype`;
 * out: images out_batch_stride elements apart (a channel slice of the GRU's input buffer works); fp32
 * arithmetic, result rounded to `dtype`.
 */
int rc_pool2x(const void* x, void* out, int64_t N, int64_t C, int64_t H, int64_t W,
              int64_t out_batch_stride, int dtype, void* stream);
int rc_interp_bilinear(const void* x, void* out, int64_t N, int64_t C, int64_t H, int64_t W,
                       int64_t Ho, int64_t Wo, int64_t out_batch_stride, int dtype, void* stream);
/* Their adjoints (gathers, no atomics): grad_out is the contiguous gradient of the forward's output
 * ([N,C,Ho,Wo], Ho/Wo of pool2x as above), grad_x [N,C,H,W] is overwritten; both of `dtype`. */
int rc_pool2x_bwd(const void* grad_out, void* grad_x, int64_t N, int64_t C, int64_t H, int64_t W,
                  int dtype, void* stream);
int rc_interp_bilinear_bwd(const void* grad_out, void* grad_x, int64_t N, int64_t C, int64_t H,
                           int64_t W, int64_t Ho, int64_t Wo, int dtype, void* stream);

/*
 * Epilogue of the update block's dense convolutions in inference (core/update.py:13,77-84,133-137: every
 * `conv(x)` adds its bias, most are followed by F.relu): x = act(x + bias[c]) in place, x [N,C,HW] contiguous
 * of `dtype` (the output of a bias-free convolution), bias [C] of bias_dtype (x's type, or RC_F32: the
 * fp32 parameter under autocast, rounded to x's type first as autocast's cast does); relu != 0 applies
 * max(., 0) keeping NaN.  Replaces the separate broadcast add_ and ReLU passes torch's MIOpen path runs,
 * with their roundings (sum rounded to `dtype`).
 */
int rc_bias_act(void* x, const void* bias, int bias_dtype, int64_t N, int64_t C, int64_t HW, int dtype,
                int relu, void* stream);

/*
 * SURVEY.md section 8(f) rank 4 -- convex upsampling, RAFTStereo.upsample_flow (core/raft_stereo.py:55-67):
 *     out[n, c, f*h + fy, f*w + fx] = sum_k softmax_k(mask[n, k, fy, fx, h, w]) * f * flow[n, c, h + k/3 - 1, w + k%3 - 1]
 * (zeros outside the image, F.unfold padding=1).  flow: float [N, >=channels, H, W], the first `channels`
 * (1 or 2; the caller keeps only channel 0, raft_stereo.py:134) planes of each image are used and images are
 * flow_batch_stride elements apart; mask: [N, 9*f*f, H, W] of mask_dtype, contiguous; out: float
 * [N, channels, f*H, f*W].  factor f in {1, 2, 4, 8} (= 2**n_downsample).  The softmax is evaluated in fp32 and
 * rounded to mask_dtype (what torch.softmax on the autocast mask returns); products and sum are fp32.
 */
int rc_upsample_flow_fwd(const float* flow, int64_t flow_batch_stride, const void* mask, float* out,
                         int64_t N, int channels, int64_t H, int64_t W, int factor, int mask_dtype,
                         void* stream);
/*
 * Its adjoint (autograd of core/raft_stereo.py:59-67, run once per GRU iteration in training):
 * grad_out float [N, channels, f*H, f*W] -> grad_mask [N, 9*f*f, H, W] of mask_dtype (softmax backward,
 * rounded where the reference's 16-bit tensors round) and grad_flow float, `channels` planes per image,
 * images grad_flow_batch_stride elements apart (planes beyond `channels` are not touched).  `work` is
 * scratch of rc_upsample_flow_bwd_work_bytes() (the per-pixel, per-neighbour sums that the second kernel
 * gathers into grad_flow: no atomics, reproducible run to run).
 */
int64_t rc_upsample_flow_bwd_work_bytes(int64_t N, int channels, int64_t H, int64_t W);
int rc_upsample_flow_bwd(const float* flow, int64_t flow_batch_stride, const void* mask,
                         const float* grad_out, void* grad_mask, float* grad_flow,
                         int64_t grad_flow_batch_stride, void* work, int64_t N, int channels,
                         int64_t H, int64_t W, int factor, int mask_dtype, void* stream);

/*
 * SURVEY.md section 8(f) rank 4, second half -- sequence_loss (train_stereo.py:35-70) over all n GRU
 * iterations' predictions in one pass (the reference runs ~8 tensor ops and a boolean-mask gather, i.e. a
 * host synchronisation, per prediction):
 *     valid' = (valid >= 0.5) & (|flow_gt| < max_flow)
 *     loss   = sum_i weights[i] * mean over valid' of |pred_i - flow_gt|        weights[i] = gamma'^(n-i-1)
 *     epe    = |pred_last - flow_gt| on valid'
 * preds: HOST array of n (<= RC_MAX_FUSED_LOOKUPS) device pointers, each float [N,1,H,W] whose images are
 * pred_batch_stride elements apart (flow_up[:, :1] of a 2-channel tensor is fine); weights: HOST float[n];
 * flow_gt float [N,1,H,W], valid float [N,H,W], HW = H*W.  results: DEVICE double[n+7] =
 *     [0,n) sum_i, [n] #valid', [n+1] sum epe, [n+2..n+4] #epe<1, <3, <5, [n+5] loss, [n+6] number of
 *     non-finite prediction values (the reference asserts there are none, train_stereo.py:51-52).
 * Block partials (float) go to `work` (rc_sequence_loss_work_bytes) and are added in double in a fixed order:
 * no atomics, reproducible run to run.  rc_sequence_loss_bwd writes
 *     grad_preds[i] = grad_loss[0] * weights[i] * sign(pred_i - flow_gt) / #valid'  on valid', 0 elsewhere,
 * reading #valid' from `results` and the upstream gradient from the DEVICE scalar grad_loss (no host sync).
 */
int64_t rc_sequence_loss_work_bytes(int n_predictions);
int rc_sequence_loss_fwd(const float* const* preds, const float* weights, int n_predictions,
                         int64_t pred_batch_stride, const float* flow_gt, const float* valid,
                         float max_flow, double* results, void* work, int64_t N, int64_t HW,
                         void* stream);
int rc_sequence_loss_bwd(const float* const* preds, float* const* grad_preds, const float* weights,
                         int n_predictions, int64_t pred_batch_stride, int64_t grad_batch_stride,
                         const float* flow_gt, const float* valid, float max_flow,
                         const double* results, const float* grad_loss, int64_t N, int64_t HW,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAFTCORR_H */

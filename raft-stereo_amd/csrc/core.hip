// Library-level plumbing: version, thread-local error string, pyramid layout.
#include "rc_common.h"

#include <string.h>

static thread_local char g_err[512] = "";

void rc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int rc_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rc_set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return RC_OK;
}

extern "C" int rc_version(void) { return RC_VERSION; }

extern "C" const char* rc_last_error_string(void) { return g_err; }

static int make_layout(int64_t B, int64_t H, int64_t W1, int64_t W2, int L, int dtype, bool full, rc_layout* out) {
  RC_REQUIRE(out != nullptr, RC_EINVAL, "rc_pyramid_layout: null output");
  RC_REQUIRE(B > 0 && H > 0 && W1 > 0 && W2 > 0, RC_EINVAL,
             "rc_pyramid_layout: non-positive size B=%lld H=%lld W1=%lld W2=%lld", (long long)B, (long long)H,
             (long long)W1, (long long)W2);
  RC_REQUIRE(L >= 1 && L <= RC_MAX_LEVELS, RC_EUNSUPPORTED, "num_levels %d outside 1..%d", L, RC_MAX_LEVELS);
  RC_REQUIRE(dtype == RC_F32 || dtype == RC_F16 || dtype == RC_BF16, RC_EUNSUPPORTED, "unknown dtype %d", dtype);
  RC_REQUIRE((W2 >> (L - 1)) >= 1, RC_EINVAL, "W2=%lld is too small for %d levels (needs W2 >= %d)",
             (long long)W2, L, 1 << (L - 1));
  RC_REQUIRE(W1 < (1 << 24) && W2 < (1 << 24) && B * H < (1ll << 31), RC_EINVAL, "sizes exceed index range");
  memset(out, 0, sizeof(*out));
  out->B = B;
  out->H = H;
  out->W1 = W1;
  out->W2 = W2;
  out->L = L;
  out->pitch = (int32_t)((W1 + 63) / 64 * 64);
  out->rb = dtype == RC_F32 ? 4 : 8;
  out->dtype = dtype;
  out->full = full ? 1 : 0;
  int64_t off = 0;
  for (int i = 0; i < L; ++i) {
    out->width[i] = (int32_t)(W2 >> i);
    out->widthp[i] = (out->width[i] + 7) / 8 * 8;
    out->stored[i] = (full || (i & 1) == 0) ? 1 : 0;  // odd levels of a forward pyramid are derived in registers
    if (!out->stored[i]) continue;
    out->offset[i] = off;
    off += B * H * (int64_t)out->widthp[i] * out->pitch;
  }
  out->total = off;
  return RC_OK;
}

extern "C" int rc_pyramid_layout(int64_t B, int64_t H, int64_t W1, int64_t W2, int L, int dtype,
                                 rc_layout* out) {
  return make_layout(B, H, W1, W2, L, dtype, false, out);
}

extern "C" int rc_grad_pyramid_layout(int64_t B, int64_t H, int64_t W1, int64_t W2, int L, rc_layout* out) {
  return make_layout(B, H, W1, W2, L, RC_F32, true, out);
}

int rc_validate_layout(const rc_layout* lay) {
  rc_layout ref;
  int rc = make_layout(lay->B, lay->H, lay->W1, lay->W2, lay->L, lay->dtype, lay->full != 0, &ref);
  if (rc) return rc;
  RC_REQUIRE(memcmp(&ref, lay, sizeof(ref)) == 0, RC_EINVAL,
             "rc_layout was not produced by rc_pyramid_layout / rc_grad_pyramid_layout");
  return RC_OK;
}

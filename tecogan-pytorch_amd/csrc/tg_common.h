// Shared helpers for libtecogan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <stdlib.h>

#include "../../include/tecogan_hip.h"

// Tuning / ablation switches read from the environment exist only in lab builds
// (tools/build_lab_libs.sh, -DTG_LAB=1).  The shipped library reads exactly three variables,
// all documented in INTEGRATION.md: TG_CONV_WINO, TG_WINO_CHAIN and TG_WINO_RES (kernel-form selection
// for A/B runs; every setting produces reference-parity results).  The Python mirror has switches of its own
// (TG_FNET_BATCH, TG_FNET_FIRST_BATCH, TG_WINO_RES_CT, TG_CONV4_DIRECT, TECOGAN_COMM, TECOGAN_HIP_LIB): INTEGRATION.md.
#ifndef TG_LAB
#define TG_LAB 0
#endif
#if TG_LAB
#define TG_LAB_ENV(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define TG_LAB_ENV(name, dflt) (dflt)
#endif

namespace tg {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return TG_E_HIP;
  }
  return TG_OK;
}

#define TG_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) {                  \
      tg::set_error(__VA_ARGS__);   \
      return (code);                \
    }                               \
  } while (0)

// whether tg_convt3x3s2_z_fwd_form(form) runs as two launches (whole rounds of 4-row workgroups + a 2-row tail)
bool convt_z_split_rule(int n, int h, int w, int form);

// split-K halves of tg_conv3x3_splitk_fwd (the plan accounts them as separate kernel classes)
int conv3x3_splitk_conv(const float* x, int64_t x_nstride, int c1, const float* x2,
                        int64_t x2_nstride, const float* w_packed, int ocb, int n, int cin,
                        int cout, int h, int w, int ksplit, float* partials, tg_stream_t stream);
int conv3x3_splitk_finalize(const float* partials, int ksplit, const float* bias, int act, int pool,
                            float* y, int n, int cout, int h, int w, tg_stream_t stream);

// rows of the image one conv3x3 workgroup covers (selects the kernel variant):
// 64-oc blocks use the 4-row / 2-tile-per-wave variant on big images and the
// 2-row variant (twice the workgroups) where tile quantisation dominates.
inline int conv3x3_rows_per_wg(int ocb, long long pixels) {
  if (ocb == 32) return 4;
  return pixels >= 200000 ? 4 : 2;
}

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// 64-channel-group layers with at most half as many 2-row tiles as CUs run the in-workgroup
// K-split variant of the conv kernel (1-row tiles, two wave groups over alternating chunks).
bool conv3x3_uses_wg_ksplit(int n, int cin, int cout, int h, int w);
bool conv3x3_uses_oneshot(int n, int cin, int cout, int h, int w);
// Winograd F(2x2,3x3) form (tg_conv3x3_wino.hip); arguments as tg_conv3x3_wino_fwd, unchecked
int conv3x3_wino_launch(const float* x, int64_t x_ns, int c1, const float* x2, int64_t x2_ns, const float* u,
                        const float* bias, const float* res, int64_t res_ns, const float* mask,
                        int64_t mask_ns, float* y, int64_t y_ns, int n, int cin, int cout, int h, int w,
                        int act, tg_stream_t stream);

// several dependent Winograd layers in one launch (tg_conv3x3_wino.hip); err: int32 fault counter in
// device or pinned host memory, poll_limit < 0 injects a fault into every waiting workgroup
constexpr int TG_CHAIN_POLL_LIMIT_DEFAULT = 1 << 21;
int conv3x3_wino_chain_launch(const tg_wino_layer* layers, int n_layers, int n, int cout, int h, int w,
                              int32_t* flags, int32_t* err, unsigned epoch, int poll_limit, tg_stream_t stream);

// SRNet's conv_in + residual-block convs of one frame as ONE launch of persistent, LDS-resident workgroups
// (tg_conv3x3_wino_res.hip): supported shapes, workspace (exchange buffer + flags + 256 bytes), launch
bool conv3x3_wino_resident_ok(int n, int cout, int h, int w);
int64_t conv3x3_wino_resident_ws_bytes(int h, int w);
int conv3x3_wino_resident_launch(const tg_wino_layer* layers, int n_layers, int cout, int h, int w, void* ws,
                                 int32_t* err, unsigned base, int poll_limit, tg_stream_t stream,
                                 const tg_wres_convt* ct = nullptr);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// input-channel chunk streamed through LDS per K-step group
constexpr int CK = 8;

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case TG_ACT_RELU: return v > 0.f ? v : 0.f;
    case TG_ACT_LRELU02: return v >= 0.f ? v : v * 0.2f;
    case TG_ACT_TANH24: return tanhf(v) * 24.f;
    default: return v;
  }
}

// Negative-side slope of the piecewise-linear activations: x >= 0 ? x : x*slope
// (1 = identity, 0 = ReLU, 0.2 = LeakyReLU(0.2)); wave-uniform, branch free.
__device__ __forceinline__ float act_slope(int act) {
  return act == TG_ACT_RELU ? 0.f : (act == TG_ACT_LRELU02 ? 0.2f : 1.f);
}

inline float act_slope_host(int act) {
  return act == TG_ACT_RELU ? 0.f : (act == TG_ACT_LRELU02 ? 0.2f : 1.f);
}

// ---- shared sampling arithmetic (device) ---------------------------------
// BicubicUpsampler weights (net_utils.py:113-127), a = -0.75, fp32, evaluated
// exactly as cubic @ [1, s, s^2, s^3] with s = d/f.
__device__ __forceinline__ void bicubic_w(int d, int f, float k[4]) {
  const float a = -0.75f;
  float s = (float)d / (float)f;  // 1.0*d/f in double then fp32 in the reference; d/f exact for f in {2,4}
  float s2 = s * s, s3 = s2 * s;
  // rows of the Keys matrix times [1, s, s2, s3], left-to-right accumulation
  k[0] = ((0.f * 1.f + a * s) + (-2.f * a) * s2) + a * s3;
  k[1] = ((1.f * 1.f + 0.f * s) + (-(a + 3.f)) * s2) + (a + 2.f) * s3;
  k[2] = ((0.f * 1.f + (-a) * s) + (2.f * a + 3.f) * s2) + (-(a + 2.f)) * s3;
  k[3] = ((0.f * 1.f + 0.f * s) + a * s2) + (-a) * s3;
}

// F.interpolate(bilinear, align_corners=False), integer scale: source taps.
__device__ __forceinline__ void bilinear_src(int dst, int scale, int in_size,
                                             int& i0, int& i1, float& l0,
                                             float& l1) {
  float src = ((float)dst + 0.5f) * (1.0f / (float)scale) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)floorf(src);
  i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

// fp32 torch.linspace(-1, 1, n)[i] (ATen CPU: two fused multiply-adds)
// Branch- and select-free form (round 4: v_cndmask_b32 costs ~23 cycles per wave instruction on gfx950,
// tools/valu_lab.hip): with j = min(i, n-1-i) the second half is -fmaf(step, j, -1) -- round-to-nearest is
// sign symmetric, so that is fmaf(-step, j, 1) bit for bit (but for the sign of an exact zero, which the
// callers add 1 to) -- and the sign bit comes from the integer n/2 - 1 - i.
__device__ __forceinline__ float linspace_m1p1(int i, int n, float step) {
  const int mirror = n - 1 - i;
  const int j = i < mirror ? i : mirror;                                   // v_min_i32
  const unsigned flip = (unsigned)(n / 2 - 1 - i) & 0x80000000u;           // i >= n/2
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __builtin_fmaf(step, (float)j, -1.0f)) ^ flip);
}

// Sampling position of backward_warp along one axis (net_utils.py:62-78 +
// grid_sample align_corners=True, padding_mode='border').
__device__ __forceinline__ float warp_coord(int i, int n, float flow) {
  float step = 2.0f / (float)(n - 1);
  float half = (float)(n - 1) / 2.0f;
  float g = linspace_m1p1(i, n, step) + flow / half;
  float p = (g + 1.0f) * half;
  p = __builtin_fmaxf(p, 0.f);                       // clip_coordinates (v_max / v_min: no conditional moves)
  p = __builtin_fminf(p, (float)(n - 1));
  return p;
}

}  // namespace tg

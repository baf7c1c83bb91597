// Gather-side kernels of the recurrent loop (gfx950, HBM/L2-bound):
//   * fused  reflect-pad -> scale*upsample(flow) -> backward_warp -> space_to_depth
//   * stand-alone backward_warp, space_to_depth, upsample, maxpool2, quantise.
//
// The fused kernel never materialises the HR flow, the meshgrid, the warped
// frame or the permuted copy the reference creates
// (codes/utils/net_utils.py:62-72 + :36-47): per HR frame it reads the LR flow
// (0.3 MB) and hr_prev (8.2 MB) and writes the 48-channel SRNet input slice
// (8.2 MB) -- 16.8 MB of algorithmic traffic instead of ~75 MB.
//
// Thread mapping: one thread = one LR column `ox` of one HR row `hy`, i.e. the
// `s` horizontally adjacent HR pixels that space_to_depth scatters to `s`
// different channel planes.  For a fixed (sy, sx, c) plane consecutive lanes
// write consecutive `ox`: every store instruction is a contiguous 256-byte
// row segment; the vertical bicubic partial sums of the flow are shared by
// the s pixels of a thread.
#include "tg_common.h"

namespace tg {

// row / column of the reflect-padded (bottom/right) flow -> source index.
// F.pad(..., 'reflect') at tecogan_nets.py:241: padded index fh+k mirrors fh-2-k.
__device__ __forceinline__ int reflect_src(int i, int f) { return i < f ? i : 2 * f - 2 - i; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// bilinear tap gather with grid_sample's out-of-bounds-is-zero rule
__device__ __forceinline__ float warp_sample(const float* __restrict__ img, int h, int w,
                                             float px, float py) {
  float fx0 = floorf(px), fy0 = floorf(py);
  float wx1 = px - fx0, wx0 = 1.0f - wx1;
  float wy1 = py - fy0, wy0 = 1.0f - wy1;
  int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  // px in [0, w-1] so x0,y0 are always inside; x1/y1 may equal w/h (weight 0).
  float v00 = img[y0 * w + x0];
  float v01 = (x1 <= w - 1) ? img[y0 * w + x1] : 0.f;
  float v10 = (y1 <= h - 1) ? img[y1 * w + x0] : 0.f;
  float v11 = (x1 <= w - 1 && y1 <= h - 1) ? img[y1 * w + x1] : 0.f;
  return ((v00 * (wy0 * wx0) + v01 * (wy0 * wx1)) + v10 * (wy1 * wx0)) + v11 * (wy1 * wx1);
}

struct FusedArgs {
  const float* lr_flow;  // (n,2,fh,fw)
  const float* hr_prev;  // (n,c,S*h,S*w)
  float* out;            // (n, S*S*c, h, w) via out_ns
  float* hr_flow_out;    // optional (n,2,S*h,S*w)
  long long out_ns;
  int n, c, h, w, fh, fw, up_mode;
};

template <int S, int C>
__global__ __launch_bounds__(256) void flowup_warp_s2d_kernel(FusedArgs a) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int hy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  const int HH = S * a.h, WW = S * a.w;
  if (ox >= a.w || hy >= HH) return;
  const int oy = hy / S, sy = hy - oy * S;
  const float* f0 = a.lr_flow + (long long)n * 2 * a.fh * a.fw;
  const float* f1 = f0 + a.fh * a.fw;

  float fxv[S], fyv[S];
  if (a.up_mode == TG_UP_BICUBIC) {
    float ky[4];
    bicubic_w(sy, S, ky);
    int rr[4], cc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      rr[p] = reflect_src(clampi(oy - 1 + p, 0, a.h - 1), a.fh);
      cc[p] = reflect_src(clampi(ox - 1 + p, 0, a.w - 1), a.fw);
    }
    float vx[4], vy[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sx_ = 0.f, sy_ = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        sx_ += ky[p] * f0[rr[p] * a.fw + cc[q]];
        sy_ += ky[p] * f1[rr[p] * a.fw + cc[q]];
      }
      vx[q] = sx_; vy[q] = sy_;
    }
#pragma unroll
    for (int d = 0; d < S; ++d) {
      float kx[4];
      bicubic_w(d, S, kx);
      float ax = 0.f, ay = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) { ax += kx[q] * vx[q]; ay += kx[q] * vy[q]; }
      fxv[d] = (float)S * ax;
      fyv[d] = (float)S * ay;
    }
  } else {
    int y0, y1; float ly0, ly1;
    bilinear_src(hy, S, a.h, y0, y1, ly0, ly1);
    y0 = reflect_src(y0, a.fh); y1 = reflect_src(y1, a.fh);
#pragma unroll
    for (int d = 0; d < S; ++d) {
      int x0, x1; float lx0, lx1;
      bilinear_src(ox * S + d, S, a.w, x0, x1, lx0, lx1);
      x0 = reflect_src(x0, a.fw); x1 = reflect_src(x1, a.fw);
      float tx_ = lx0 * f0[y0 * a.fw + x0] + lx1 * f0[y0 * a.fw + x1];
      float bx_ = lx0 * f0[y1 * a.fw + x0] + lx1 * f0[y1 * a.fw + x1];
      float ty_ = lx0 * f1[y0 * a.fw + x0] + lx1 * f1[y0 * a.fw + x1];
      float by_ = lx0 * f1[y1 * a.fw + x0] + lx1 * f1[y1 * a.fw + x1];
      fxv[d] = (float)S * (ly0 * tx_ + ly1 * bx_);
      fyv[d] = (float)S * (ly0 * ty_ + ly1 * by_);
    }
  }

  if (a.hr_flow_out) {
    float* fo = a.hr_flow_out + (long long)n * 2 * HH * WW + (long long)hy * WW + ox * S;
#pragma unroll
    for (int d = 0; d < S; ++d) { fo[d] = fxv[d]; fo[(long long)HH * WW + d] = fyv[d]; }
  }

  const float* img = a.hr_prev + (long long)n * C * HH * WW;
  float* ob = a.out + (long long)n * a.out_ns + (long long)oy * a.w + ox;
  const long long lrhw = (long long)a.h * a.w;
#pragma unroll
  for (int d = 0; d < S; ++d) {
    const int hx = ox * S + d;
    float px = warp_coord(hx, WW, fxv[d]);
    float py = warp_coord(hy, HH, fyv[d]);
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      float v = warp_sample(img + (long long)ch * HH * WW, HH, WW, px, py);
      ob[(long long)((sy * S + d) * C + ch) * lrhw] = v;
    }
  }
}

__global__ __launch_bounds__(256) void backward_warp_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ flow,
                                                            float* __restrict__ y, int n, int c,
                                                            int h, int w) {
  const int px_ = blockIdx.x * 64 + (threadIdx.x & 63);
  const int py_ = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (px_ >= w || py_ >= h) return;
  const long long hw = (long long)h * w;
  const float* fl = flow + (long long)b * 2 * hw + (long long)py_ * w + px_;
  float sx = warp_coord(px_, w, fl[0]);
  float sy = warp_coord(py_, h, fl[hw]);
  for (int ch = 0; ch < c; ++ch) {
    const float* img = x + ((long long)b * c + ch) * hw;
    y[((long long)b * c + ch) * hw + (long long)py_ * w + px_] = warp_sample(img, h, w, sx, sy);
  }
}

__global__ void space_to_depth_kernel(const float* __restrict__ x, float* __restrict__ y,
                                      long long y_ns, int n, int c, int h, int w, int s) {
  const int oh = h / s, ow = w / s;
  const long long total = (long long)n * s * s * c * oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int ox = (int)(i % ow); long long t = i / ow;
    int oy = (int)(t % oh); t /= oh;
    int k = (int)(t % (s * s * c)); int b = (int)(t / (s * s * c));
    int ch = k % c, sxy = k / c, sx = sxy % s, sy = sxy / s;
    y[(long long)b * y_ns + ((long long)k * oh + oy) * ow + ox] =
        x[(((long long)b * c + ch) * h + (oy * s + sy)) * w + ox * s + sx];
  }
}

__global__ void upsample_kernel(const float* __restrict__ x, float* __restrict__ y, int nc, int h,
                                int w, int s, int mode, float mul) {
  const int oh = h * s, ow = w * s;
  const long long total = (long long)nc * oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int oxx = (int)(i % ow); long long t = i / ow;
    int oyy = (int)(t % oh); int p = (int)(t / oh);
    const float* src = x + (long long)p * h * w;
    float v;
    if (mode == TG_UP_BICUBIC) {
      int ii = oyy / s, dy = oyy - ii * s, jj = oxx / s, dx = oxx - jj * s;
      float ky[4], kx[4];
      bicubic_w(dy, s, ky);
      bicubic_w(dx, s, kx);
      v = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int cq = clampi(jj - 1 + q, 0, w - 1);
        float vq = 0.f;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) vq += ky[pp] * src[clampi(ii - 1 + pp, 0, h - 1) * w + cq];
        v += kx[q] * vq;
      }
    } else {
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      bilinear_src(oyy, s, h, y0, y1, ly0, ly1);
      bilinear_src(oxx, s, w, x0, x1, lx0, lx1);
      float top = lx0 * src[y0 * w + x0] + lx1 * src[y0 * w + x1];
      float bot = lx0 * src[y1 * w + x0] + lx1 * src[y1 * w + x1];
      v = ly0 * top + ly1 * bot;
    }
    y[i] = mul * v;
  }
}

__global__ void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int nc, int h,
                                int w) {
  const int oh = h / 2, ow = w / 2;
  const long long total = (long long)nc * oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int ox = (int)(i % ow); long long t = i / ow;
    int oy = (int)(t % oh); int p = (int)(t / oh);
    const float* s = x + ((long long)p * h + 2 * oy) * w + 2 * ox;
    y[i] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[w], s[w + 1]));
  }
}

__global__ void quantize_u8_hwc_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, int c,
                                       int h, int w) {
  const long long hw = (long long)h * w;
  const long long total = hw * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int ch = (int)(i % c); long long p = i / c;
    float v = rintf(x[(long long)ch * hw + p] * 255.0f);  // round-half-even like np.round
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    y[i] = (uint8_t)v;
  }
}

static inline int grid1d(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace tg

using namespace tg;

extern "C" int tg_flowup_warp_s2d_fwd(const float* lr_flow, int fh, int fw, const float* hr_prev,
                                      float* out, int64_t out_nstride, float* hr_flow_out, int n,
                                      int c, int h, int w, int scale, int up_mode,
                                      tg_stream_t stream) {
  TG_REQUIRE(lr_flow && hr_prev && out, TG_E_ARG, "flowup_warp_s2d: null pointer");
  TG_REQUIRE(n > 0 && c == 3 && h > 0 && w > 0 && (scale == 2 || scale == 4), TG_E_SHAPE,
             "flowup_warp_s2d: n=%d c=%d (3) h=%d w=%d scale=%d (2|4)", n, c, h, w, scale);
  TG_REQUIRE(fh > 0 && fw > 0 && fh <= h && fw <= w && h - fh < fh && w - fw < fw, TG_E_SHAPE,
             "flowup_warp_s2d: flow %dx%d vs lr %dx%d (reflect pad needs pad < size)", fh, fw, h, w);
  TG_REQUIRE(up_mode == TG_UP_BICUBIC || up_mode == TG_UP_BILINEAR, TG_E_ARG,
             "flowup_warp_s2d: up_mode=%d", up_mode);
  TG_REQUIRE(scale * h >= 2 && scale * w >= 2, TG_E_SHAPE, "flowup_warp_s2d: degenerate size");
  FusedArgs a{lr_flow, hr_prev, out, hr_flow_out, out_nstride, n, c, h, w, fh, fw, up_mode};
  dim3 g(cdiv(w, 64), cdiv(scale * h, 4), n), t(256);
  hipStream_t s = (hipStream_t)stream;
  if (scale == 4) hipLaunchKernelGGL((flowup_warp_s2d_kernel<4, 3>), g, t, 0, s, a);
  else hipLaunchKernelGGL((flowup_warp_s2d_kernel<2, 3>), g, t, 0, s, a);
  return check_launch("flowup_warp_s2d");
}

extern "C" int tg_backward_warp_fwd(const float* x, const float* flow, float* y, int n, int c,
                                    int h, int w, tg_stream_t stream) {
  TG_REQUIRE(x && flow && y, TG_E_ARG, "backward_warp: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && h >= 2 && w >= 2, TG_E_SHAPE, "backward_warp: n=%d c=%d h=%d w=%d",
             n, c, h, w);
  dim3 g(cdiv(w, 64), cdiv(h, 4), n), t(256);
  hipLaunchKernelGGL(backward_warp_kernel, g, t, 0, (hipStream_t)stream, x, flow, y, n, c, h, w);
  return check_launch("backward_warp");
}

extern "C" int tg_space_to_depth(const float* x, float* y, int64_t y_nstride, int n, int c, int h,
                                 int w, int scale, tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "space_to_depth: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && scale >= 1 && h >= scale && w >= scale, TG_E_SHAPE,
             "space_to_depth: n=%d c=%d h=%d w=%d s=%d", n, c, h, w, scale);
  long long total = (long long)n * scale * scale * c * (h / scale) * (w / scale);
  hipLaunchKernelGGL(space_to_depth_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                     x, y, (long long)y_nstride, n, c, h, w, scale);
  return check_launch("space_to_depth");
}

extern "C" int tg_upsample_fwd(const float* x, float* y, int nc, int h, int w, int scale,
                               int up_mode, float mul, tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "upsample: null pointer");
  TG_REQUIRE(nc > 0 && h > 0 && w > 0 && scale >= 1, TG_E_SHAPE, "upsample: nc=%d h=%d w=%d s=%d",
             nc, h, w, scale);
  TG_REQUIRE(up_mode == TG_UP_BICUBIC || up_mode == TG_UP_BILINEAR, TG_E_ARG, "upsample: mode=%d",
             up_mode);
  long long total = (long long)nc * h * scale * w * scale;
  hipLaunchKernelGGL(upsample_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, x, y,
                     nc, h, w, scale, up_mode, mul);
  return check_launch("upsample");
}

extern "C" int tg_maxpool2_fwd(const float* x, float* y, int nc, int h, int w,
                               tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "maxpool2: null pointer");
  TG_REQUIRE(nc > 0 && h >= 2 && w >= 2, TG_E_SHAPE, "maxpool2: nc=%d h=%d w=%d", nc, h, w);
  long long total = (long long)nc * (h / 2) * (w / 2);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, x, y,
                     nc, h, w);
  return check_launch("maxpool2");
}

extern "C" int tg_quantize_u8_hwc(const float* x, uint8_t* y, int c, int h, int w,
                                  tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "quantize_u8: null pointer");
  TG_REQUIRE(c > 0 && h > 0 && w > 0, TG_E_SHAPE, "quantize_u8: c=%d h=%d w=%d", c, h, w);
  long long total = (long long)c * h * w;
  hipLaunchKernelGGL(quantize_u8_hwc_kernel, dim3(grid1d(total)), dim3(256), 0,
                     (hipStream_t)stream, x, y, c, h, w);
  return check_launch("quantize_u8");
}

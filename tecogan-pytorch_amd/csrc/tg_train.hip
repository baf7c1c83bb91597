// Training-side element-wise / gather / reduction kernels (gfx950, HBM-bound):
// activation backward, bias gradient, maxpool / upsample / warp / space-to-depth
// transposes, BatchNorm (train) forward+backward, Linear(->1), the two loss
// functions with their gradients, fused Adam, axpy.  The matrix-shaped backward
// work (conv dgrad / wgrad) lives in tg_conv3x3_mfma.hip / tg_wgrad_mfma.hip.
//
// Reference ops replaced (autograd of): LeakyReLU/ReLU/tanh*24
// (tecogan_nets.py:24-80), MaxPool2d (:28-42), F.interpolate / BicubicUpsampler
// (net_utils.py:85-156), backward_warp (net_utils.py:50-82), space_to_depth
// (:36-47), BatchNorm2d (tecogan_nets.py:324-339), Linear (:375), CharbonnierLoss
// / VanillaGANLoss (optim/losses.py:6-50), optim.Adam (vsrgan_model.py:76-87).
#include "tg_common.h"

namespace tg {

static inline int grid_for(long long total, int cap = 4096) {
  long long b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
// one block per channel: 16 waves when the channel plane is large (early discriminator
// layers: 12 x 128 x 128 elements per channel), 4 otherwise
static inline int bn_threads(int n, long long hw) { return (long long)n * hw >= 32768 ? 1024 : 256; }
// slices of the batch a BatchNorm reduction is split into (1 = one block per channel): equal slices,
// at least 32k elements each, aiming at ~512 blocks
static inline int bn_slices(int n, int c, long long hw) {
  for (int ns = 8; ns >= 2; --ns)
    if (n % ns == 0 && (long long)(n / ns) * hw >= 32768 && c * ns <= 1024 && (long long)n * hw * c >= (long long)ns * 2 * c)
      return ns;
  return 1;
}
__global__ void bn_local_stats_kernel(const float* __restrict__ x, int n, int c, int hw, float* __restrict__ stats2c);
__global__ void bn_merge_stats_kernel(const float* __restrict__ gathered, int world, float cnt_r, float eps,
                                      float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                      float* __restrict__ run_mean, float* __restrict__ run_var, int c);
#define TG_GRID_STRIDE(i, total)                                                   \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); \
       i += (long long)gridDim.x * blockDim.x)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// block-wide sum (blockDim.x a multiple of 64, sm holds one float per wave); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
  __syncthreads();
  return r;
}

// dx = dy * act'(.) expressed through the OUTPUT y of the activation
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                               float* __restrict__ dx, long long n, int act) {
  TG_GRID_STRIDE(i, n) {
    float g = dy[i], o = y[i], r;
    if (act == TG_ACT_RELU) r = o > 0.f ? g : 0.f;
    else if (act == TG_ACT_LRELU02) r = o > 0.f ? g : g * 0.2f;
    else if (act == TG_ACT_TANH24) r = g * (24.f - o * o * (1.f / 24.f));  // 24*(1 - tanh^2)
    else r = g;
    dx[i] = r;
  }
}

// db[c] += sum over n, h, w of dy[n][c][h][w]; grid = (channel, slice): each block reduces a
// slice of the (n, hw) range and adds its partial with one atomic (db is zeroed first when
// not accumulating), so large batched gradients (19 frames x HR) fill the GPU.
struct BiasGradSegs { const float* seg[64]; };
__global__ __launch_bounds__(256) void bias_grad_kernel(BiasGradSegs dy, int n_per_seg,
                                                        float* __restrict__ db, int n, int c,
                                                        int hw, int nslice) {
  __shared__ float sm[4];
  const int ch = blockIdx.x, sl = blockIdx.y;
  const long long total = (long long)n * hw;
  const long long per = ((total + nslice - 1) / nslice + 3) & ~3ll;    // multiple of 4: slices stay 16-byte aligned
  const long long lo = (long long)sl * per;
  long long hi = lo + per; if (hi > total) hi = total;
  float s = 0.f;
  // walk the slice image by image: no per-element division, block-uniform plane pointer
  int b = (int)(lo / hw);
  long long r0 = lo - (long long)b * hw;
  for (long long left = hi - lo; left > 0; ++b, r0 = 0) {
    const int sg = b / n_per_seg, lb = b - sg * n_per_seg;
    const float* __restrict__ pl = dy.seg[sg] + ((long long)lb * c + ch) * hw;
    const long long cnt = (hw - r0 < left) ? hw - r0 : left;
    if (((hw | r0 | cnt) & 3) == 0 && (((uintptr_t)pl) & 15) == 0) {     // 16-byte loads (every slice of an hw % 4 == 0 plane)
      const f32x4* p4 = reinterpret_cast<const f32x4*>(pl + r0);
      for (long long r = threadIdx.x; r < (cnt >> 2); r += 256) { const f32x4 v = p4[r]; s += (v[0] + v[1]) + (v[2] + v[3]); }
    } else {
      for (long long r = threadIdx.x; r < cnt; r += 256) s += pl[r0 + r];
    }
    left -= cnt;
  }
  float r = block_sum(s, sm);
  if (threadIdx.x == 0) atomicAdd(db + ch, r);
}

// The bias gradients of all the layers of the chained SRNet body in one launch: layer L = 0..nlayer-1
// reads base[f] + L * lstride for every frame f; grid = (channel, slice, layer).
struct BiasBodyArgs { const float* base[64]; float* db[25]; };
__global__ __launch_bounds__(256) void bias_grad_body_kernel(BiasBodyArgs a, int nlayer, long long lstride,
                                                             int nframes, int n_per, int c, int hw, int nslice) {
  __shared__ float sm[4];
  const int ch = blockIdx.x, sl = blockIdx.y, L = blockIdx.z;
  const long long total = (long long)nframes * n_per * hw;
  const long long per = ((total + nslice - 1) / nslice + 3) & ~3ll;
  const long long lo = (long long)sl * per;
  long long hi = lo + per; if (hi > total) hi = total;
  float s = 0.f;
  int b = (int)(lo / hw);
  long long r0 = lo - (long long)b * hw;
  for (long long left = hi - lo; left > 0; ++b, r0 = 0) {
    const int f = b / n_per, lb = b - f * n_per;
    const float* src = a.base[f] + (long long)L * lstride;
    const float* __restrict__ pl = src + ((long long)lb * c + ch) * hw;
    const long long cnt = (hw - r0 < left) ? hw - r0 : left;
    if (((hw | r0 | cnt) & 3) == 0 && (((uintptr_t)pl) & 15) == 0) {
      const f32x4* p4 = reinterpret_cast<const f32x4*>(pl + r0);
      for (long long r = threadIdx.x; r < (cnt >> 2); r += 256) { const f32x4 v = p4[r]; s += (v[0] + v[1]) + (v[2] + v[3]); }
    } else {
      for (long long r = threadIdx.x; r < cnt; r += 256) s += pl[r0 + r];
    }
    left -= cnt;
  }
  float r = block_sum(s, sm);
  if (threadIdx.x == 0) atomicAdd(a.db[L] + ch, r);
}

// gradient of MaxPool2d(2,2) floor mode: goes to the FIRST maximal element of the window
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    float* __restrict__ dx, int nc, int h, int w) {
  const int oh = h / 2, ow = w / 2;
  const long long total = (long long)nc * h * w;
  TG_GRID_STRIDE(i, total) {
    int xx = (int)(i % w); long long t = i / w;
    int yy = (int)(t % h); int p = (int)(t / h);
    int oy = yy / 2, ox = xx / 2;
    float g = 0.f;
    if (oy < oh && ox < ow) {
      const float* s = x + ((long long)p * h + 2 * oy) * w + 2 * ox;
      float v0 = s[0], v1 = s[1], v2 = s[w], v3 = s[w + 1];
      int arg = 0; float m = v0;
      if (v1 > m) { m = v1; arg = 1; }
      if (v2 > m) { m = v2; arg = 2; }
      if (v3 > m) { m = v3; arg = 3; }
      if (arg == (yy - 2 * oy) * 2 + (xx - 2 * ox)) g = dy[((long long)p * oh + oy) * ow + ox];
    }
    dx[i] = g;
  }
}

// transpose of tg_upsample_fwd in GATHER form (deterministic, no atomics):
//   dx[r][c] = mul * sum over the outputs (oy, ox) whose stencil touches input (r, c).
// bicubic: output row oy = s*i + d uses inputs clamp(i-1+p), p = 0..3, so input r is reached only
// from i in [r-2, r+1] (the clamped border taps fall in the same window); bilinear: output oy uses
// y0 = floor(src), y1 = min(y0+1, h-1), so input r is reached from oy in [s*(r-1), s*(r+1)+s).
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int nc,
                                    int h, int w, int s, int mode, float mul) {
  const int oh = h * s, ow = w * s;
  const long long total = (long long)nc * h * w;
  TG_GRID_STRIDE(idx, total) {
    int c = (int)(idx % w); long long t = idx / w;
    int r = (int)(t % h); int p = (int)(t / h);
    const float* g = dy + (long long)p * oh * ow;
    float acc = 0.f;
    if (mode == TG_UP_BICUBIC) {
      // per-axis weight of output o on this input: sum of the taps that clamp onto it
      for (int i = r - 2; i <= r + 1; ++i) {
        if (i < 0 || i >= h) continue;
        for (int d = 0; d < s; ++d) {
          float ky[4];
          bicubic_w(d, s, ky);
          float wy = 0.f;
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) {
            int rp = i - 1 + pp; rp = rp < 0 ? 0 : (rp > h - 1 ? h - 1 : rp);
            if (rp == r) wy += ky[pp];
          }
          if (wy == 0.f) continue;
          const float* grow = g + (long long)(i * s + d) * ow;
          float rowacc = 0.f;
          for (int j = c - 2; j <= c + 1; ++j) {
            if (j < 0 || j >= w) continue;
            for (int e = 0; e < s; ++e) {
              float kx[4];
              bicubic_w(e, s, kx);
              float wx = 0.f;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                int cq = j - 1 + q; cq = cq < 0 ? 0 : (cq > w - 1 ? w - 1 : cq);
                if (cq == c) wx += kx[q];
              }
              if (wx != 0.f) rowacc += wx * grow[j * s + e];
            }
          }
          acc += wy * rowacc;
        }
      }
    } else {
      int oy_lo = s * (r - 1), oy_hi = s * (r + 2);
      int ox_lo = s * (c - 1), ox_hi = s * (c + 2);
      oy_lo = oy_lo < 0 ? 0 : oy_lo; oy_hi = oy_hi > oh ? oh : oy_hi;
      ox_lo = ox_lo < 0 ? 0 : ox_lo; ox_hi = ox_hi > ow ? ow : ox_hi;
      for (int oy = oy_lo; oy < oy_hi; ++oy) {
        int y0, y1; float ly0, ly1;
        bilinear_src(oy, s, h, y0, y1, ly0, ly1);
        float wy = (y0 == r ? ly0 : 0.f) + (y1 == r ? ly1 : 0.f);
        if (wy == 0.f) continue;
        float rowacc = 0.f;
        for (int ox = ox_lo; ox < ox_hi; ++ox) {
          int x0, x1; float lx0, lx1;
          bilinear_src(ox, s, w, x0, x1, lx0, lx1);
          float wx = (x0 == c ? lx0 : 0.f) + (x1 == c ? lx1 : 0.f);
          if (wx != 0.f) rowacc += wx * g[(long long)oy * ow + ox];
        }
        acc += wy * rowacc;
      }
    }
    dx[idx] = mul * acc;
  }
}

// backward of backward_warp: dflow (n,2,h,w) and (optionally) dimg scatter.  dimg pre-zeroed.
__global__ __launch_bounds__(256) void backward_warp_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ flow, const float* __restrict__ dy,
    float* __restrict__ dimg, float* __restrict__ dflow, int n, int c, int h, int w, int s2d) {
  const int px_ = blockIdx.x * 64 + (threadIdx.x & 63);
  const int py_ = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (px_ >= w || py_ >= h) return;
  const long long hw = (long long)h * w;
  const long long pix = (long long)py_ * w + px_;
  const float fx = flow[(long long)b * 2 * hw + pix], fy = flow[((long long)b * 2 + 1) * hw + pix];
  // recompute the forward sampling position, tracking whether the clip was active
  float halfx = (float)(w - 1) / 2.0f, halfy = (float)(h - 1) / 2.0f;
  float gx = linspace_m1p1(px_, w, 2.0f / (float)(w - 1)) + fx / halfx;
  float gy = linspace_m1p1(py_, h, 2.0f / (float)(h - 1)) + fy / halfy;
  float ux = (gx + 1.0f) * halfx, uy = (gy + 1.0f) * halfy;
  // clip_coordinates_set_grad: gradient passes only strictly inside (0, size-1)
  float mx = (ux <= 0.f || ux >= (float)(w - 1)) ? 0.f : 1.f;
  float my = (uy <= 0.f || uy >= (float)(h - 1)) ? 0.f : 1.f;
  float sx = ux < 0.f ? 0.f : (ux > (float)(w - 1) ? (float)(w - 1) : ux);
  float sy = uy < 0.f ? 0.f : (uy > (float)(h - 1) ? (float)(h - 1) : uy);
  float fx0 = floorf(sx), fy0 = floorf(sy);
  float wx1 = sx - fx0, wx0 = 1.f - wx1, wy1 = sy - fy0, wy0 = 1.f - wy1;
  int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  bool okx1 = x1 <= w - 1, oky1 = y1 <= h - 1;
  float gsx = 0.f, gsy = 0.f;
  // s2d > 1: dy is the gradient of space_to_depth(warp(x), s2d) (plane (sy s + sx) c + ch, h / s x w / s)
  const int oy_ = py_ / s2d, ox_ = px_ / s2d, ph_ = (py_ - oy_ * s2d) * s2d + (px_ - ox_ * s2d);
  const long long ohw = hw / (s2d * s2d);
  const float* dyo = dy + ((long long)b * c * s2d * s2d + (long long)ph_ * c) * ohw + (long long)oy_ * (w / s2d) + ox_;
  for (int ch = 0; ch < c; ++ch) {
    const long long plane = ((long long)b * c + ch) * hw;
    const float g = dyo[ch * ohw];
    const float* img = x + plane;
    float v00 = img[y0 * w + x0];
    float v01 = okx1 ? img[y0 * w + x1] : 0.f;
    float v10 = oky1 ? img[y1 * w + x0] : 0.f;
    float v11 = (okx1 && oky1) ? img[y1 * w + x1] : 0.f;
    // d out / d sx, d out / d sy
    gsx += g * ((v01 - v00) * wy0 + (v11 - v10) * wy1);
    gsy += g * ((v10 - v00) * wx0 + (v11 - v01) * wx1);
    if (dimg) {
      float* d = dimg + plane;
      atomicAdd(d + y0 * w + x0, g * wy0 * wx0);
      if (okx1) atomicAdd(d + y0 * w + x1, g * wy0 * wx1);
      if (oky1) atomicAdd(d + y1 * w + x0, g * wy1 * wx0);
      if (okx1 && oky1) atomicAdd(d + y1 * w + x1, g * wy1 * wx1);
    }
  }
  if (dflow) {
    // s = (g + 1) * half, g = lin + f / half  =>  ds/df = 1 (where not clipped)
    dflow[(long long)b * 2 * hw + pix] = gsx * mx;
    dflow[((long long)b * 2 + 1) * hw + pix] = gsy * my;
  }
}

// inverse of space_to_depth: x (n, s*s*c, h, w) -> y (n, c, s*h, s*w)
__global__ void depth_to_space_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                      int c, int h, int w, int s) {
  const int oh = h * s, ow = w * s;
  const long long total = (long long)n * c * oh * ow;
  TG_GRID_STRIDE(i, total) {
    int ox = (int)(i % ow); long long t = i / ow;
    int oy = (int)(t % oh); t /= oh;
    int ch = (int)(t % c); int b = (int)(t / c);
    int sy = oy % s, sx = ox % s;
    int k = (sy * s + sx) * c + ch;
    y[i] = x[(((long long)b * s * s * c + k) * h + oy / s) * w + ox / s];
  }
}

// vector form (S in {2, 4}, output rows of 4 S pixels, 16-byte aligned): S 16-byte loads (one from
// each sub-pixel plane) -> 4 S consecutive output pixels as S 16-byte stores
// ym != null: the result is also the gradient of an activation output ym (same layout as y): it is
// multiplied by act'(.) on the way out (ym > 0 ? 1 : slope) -- the separate act_bwd pass (read g, read
// ym, write dz: 600 MB for the critic's first layer at crop 256) disappears.
template <int S>
__global__ __launch_bounds__(256) void depth_to_space_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int n, int c, int h, int w,
                                                                const float* __restrict__ ym = nullptr, float slope = 0.f) {
  const int oh = h * S, ow = w * S, groups = w / 4;
  const long long total = (long long)n * c * oh * groups;
  TG_GRID_STRIDE(i, total) {
    const int gq = (int)(i % groups); long long t = i / groups;
    const int oy = (int)(t % oh); t /= oh;
    const int ch = (int)(t % c); const int b = (int)(t / c);
    const int iy = oy / S, sy = oy - iy * S;
    float v[4 * S];
#pragma unroll
    for (int sx = 0; sx < S; ++sx) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(
          x + (((long long)b * S * S * c + (sy * S + sx) * c + ch) * h + iy) * w + gq * 4);
      v[sx] = q[0]; v[S + sx] = q[1]; v[2 * S + sx] = q[2]; v[3 * S + sx] = q[3];
    }
    const long long doff = (((long long)b * c + ch) * oh + oy) * ow + (long long)gq * 4 * S;
    f32x4* dst = reinterpret_cast<f32x4*>(y + doff);
    if (ym) {
      const f32x4* msk = reinterpret_cast<const f32x4*>(ym + doff);
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const f32x4 m = msk[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * k + e] = m[e] > 0.f ? v[4 * k + e] : v[4 * k + e] * slope;
      }
    }
#pragma unroll
    for (int k = 0; k < S; ++k) dst[k] = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
  }
}

// Charbonnier: loss += scale * sum sqrt(d^2+eps); dx = gscale * d / sqrt(d^2+eps)
__global__ __launch_bounds__(256) void charbonnier_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ y,
                                                          long long n, float eps, float scale,
                                                          float* __restrict__ loss,
                                                          float gscale, float* __restrict__ dx) {
  __shared__ float sm[4];
  float s = 0.f;
  TG_GRID_STRIDE(i, n) {
    float d = x[i] - y[i];
    float r = sqrtf(d * d + eps);
    s += r;
    if (dx) dx[i] = gscale * d / r;
  }
  float r = block_sum(s, sm);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, r * scale);
}

// ---- feature (VGG) / feature-matching losses ---------------------------------------------
// y = (x - mean[c]) / std[c]  (VGGFeatureExtractor.forward, vgg_nets.py:29).
// mean == nullptr -> 0, which is also the op's backward: dx = dy / std[c].
__global__ __launch_bounds__(256) void channel_norm_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ stdv,
                                                           float* __restrict__ y, long long total,
                                                           int c, long long hw) {
  TG_GRID_STRIDE(i, total) {
    int ch = (int)((i / hw) % c);
    float m = mean ? mean[ch] : 0.f;
    y[i] = (x[i] - m) / stdv[ch];
  }
}

// CosineSimilarityLoss (optim/losses.py:53-62) over dim=1 of (n,c,h,w):
//   cos_p = sum_c (a/max(|a|,eps)) (b/max(|b|,eps));  loss += scale * sum_p (1 - cos_p)
//   da = -gscale * dcos/da,  dcos/da_c = b_c/(|a|'|b|') - [|a| > eps] cos a_c/|a|^2
// (clamp_min passes no gradient to the norm below eps, as autograd does).
// One thread per pixel; consecutive threads read consecutive addresses of each channel plane.
__global__ __launch_bounds__(256) void cosine_loss_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b, long long npix,
                                                          int c, long long hw, float eps, float scale,
                                                          float* __restrict__ loss, float gscale,
                                                          float* __restrict__ da) {
  __shared__ float sm[4];
  float s = 0.f;
  TG_GRID_STRIDE(p, npix) {
    const long long base = (p / hw) * c * hw + (p % hw);
    float dot = 0.f, aa = 0.f, bb = 0.f;
    for (int ch = 0; ch < c; ++ch) {
      float av = a[base + ch * hw], bv = b[base + ch * hw];
      dot += av * bv; aa += av * av; bb += bv * bv;
    }
    const float na = sqrtf(aa), nb = sqrtf(bb);
    const float nac = fmaxf(na, eps), nbc = fmaxf(nb, eps);
    const float inv = 1.0f / (nac * nbc);
    const float cs = dot * inv;
    s += 1.0f - cs;
    if (da) {
      const float k = na > eps ? cs / (nac * nac) : 0.f;
      for (int ch = 0; ch < c; ++ch) {
        float av = a[base + ch * hw], bv = b[base + ch * hw];
        da[base + ch * hw] = -gscale * (bv * inv - k * av);
      }
    }
  }
  float r = block_sum(s, sm);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, r * scale);
}

// nn.L1Loss / nn.MSELoss (optim/__init__.py:10-14): mode 1 = |d|, 2 = d^2
__global__ __launch_bounds__(256) void pixel_loss_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ y, long long n,
                                                         int mode, float scale, float* __restrict__ loss,
                                                         float gscale, float* __restrict__ dx) {
  __shared__ float sm[4];
  float s = 0.f;
  TG_GRID_STRIDE(i, n) {
    float d = x[i] - y[i];
    if (mode == 1) {
      s += fabsf(d);
      if (dx) dx[i] = gscale * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    } else {
      s += d * d;
      if (dx) dx[i] = gscale * 2.f * d;
    }
  }
  float r = block_sum(s, sm);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, r * scale);
}

// BCE-with-logits against a constant target t; also mean(x) and mean(log(sigmoid(x)+1e-8))
// stats[0] += scale*sum(loss), stats[1] += scale*sum(x), stats[2] += scale*sum(log(sig+1e-8))
// LS = true: LSGANLoss (optim/losses.py:17-28), (x - t)^2 with dx = 2 (x - t); the logged statistics are the same
template <bool LS>
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, long long n,
                                                         float target, float scale,
                                                         float* __restrict__ stats, float gscale,
                                                         float* __restrict__ dx) {
  __shared__ float sm[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  TG_GRID_STRIDE(i, n) {
    float v = x[i];
    float sig = 1.f / (1.f + expf(-v));
    if (LS) s0 += (v - target) * (v - target);
    else s0 += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    s1 += v;
    s2 += logf(sig + 1e-8f);
    if (dx) dx[i] = LS ? gscale * (2.f * (v - target)) : gscale * (sig - target);
  }
  float r0 = block_sum(s0, sm), r1 = block_sum(s1, sm), r2 = block_sum(s2, sm);
  if (threadIdx.x == 0 && stats) {
    atomicAdd(stats + 0, r0 * scale);
    atomicAdd(stats + 1, r1 * scale);
    atomicAdd(stats + 2, r2 * scale);
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                            float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                            const float* __restrict__ skip) {
  // fail-safe of the chained launches: a fault recorded during this step (possibly on another rank: the
  // slot travels with the gradient bucket) turns the update into a no-op -- weights and moments untouched
  if (skip && *skip != 0.f) return;
  TG_GRID_STRIDE(i, n) {
    float gi = g[i];
    if (wd != 0.f) gi += wd * p[i];
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a,
                            long long n) {
  TG_GRID_STRIDE(i, n) y[i] += a * x[i];
}

__global__ void div_scalar_kernel(float* __restrict__ y, const float* __restrict__ x, float d, long long n) {
  TG_GRID_STRIDE(i, n) y[i] = x[i] / d;        // IEEE division: what DDP's `grad / world_size` computes
}

// ---- BatchNorm2d (train) + LeakyReLU(0.2) fused ---------------------------------
// stats: one block per channel -> mean, invstd (biased var), running stats update
// One pass over channel `ch` of n images (plane hw) of NA arrays that share the element offset:
// acc(t) is called once per element with t[a] = arr[a][element].  Four images at a time, 16-byte loads
// when the plane allows: 4 NA independent loads in flight per thread instead of one dependent load per
// iteration (the per-channel BatchNorm reductions were chains of n * hw / threads round trips: 16 us for 3 MB).
template <int NA, class Acc>
__device__ __forceinline__ void bn_sweep(const float* const (&arr)[NA], int n, int c, int ch, int hw, Acc&& acc) {
  bool vec = (hw & 3) == 0;
#pragma unroll
  for (int a = 0; a < NA; ++a) vec = vec && (reinterpret_cast<uintptr_t>(arr[a]) & 15) == 0;
  if (vec) {
    const int hw4 = hw >> 2;
    int b = 0;
    for (; b + 3 < n; b += 4) {
      for (int i = threadIdx.x; i < hw4; i += blockDim.x) {
        f32x4 v[4][NA];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int a = 0; a < NA; ++a)
            v[u][a] = reinterpret_cast<const f32x4*>(arr[a] + ((long long)(b + u) * c + ch) * hw)[i];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) t[a] = v[u][a][e];
            acc(t);
          }
      }
    }
    for (; b < n; ++b) {
      for (int i = threadIdx.x; i < hw4; i += blockDim.x) {
        f32x4 v[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) v[a] = reinterpret_cast<const f32x4*>(arr[a] + ((long long)b * c + ch) * hw)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t[NA];
#pragma unroll
          for (int a = 0; a < NA; ++a) t[a] = v[a][e];
          acc(t);
        }
      }
    }
    return;
  }
  for (int b = 0; b < n; ++b) {
    const long long off = ((long long)b * c + ch) * hw;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) {
      float t[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) t[a] = arr[a][off + i];
      acc(t);
    }
  }
}

__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* __restrict__ x, int n, int c,
                                                       int hw, float eps, float momentum,
                                                       float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd,
                                                       float* __restrict__ run_mean,
                                                       float* __restrict__ run_var) {
  __shared__ float sm[16];
  __shared__ float s_mean;
  int ch = blockIdx.x;
  float s = 0.f;
  const float* const xs[1] = {x};
  bn_sweep<1>(xs, n, c, ch, hw, [&](const float (&t)[1]) { s += t[0]; });
  float tot = block_sum(s, sm);
  const float cnt = (float)n * (float)hw;
  if (threadIdx.x == 0) s_mean = tot / cnt;
  __syncthreads();
  const float mean = s_mean;
  float q = 0.f;
  bn_sweep<1>(xs, n, c, ch, hw, [&](const float (&t)[1]) { const float d = t[0] - mean; q += d * d; });
  float sq = block_sum(q, sm);
  if (threadIdx.x == 0) {
    float var = sq / cnt;
    save_mean[ch] = mean;
    save_invstd[ch] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
      run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * mean;
      run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (var * cnt / (cnt - 1.f));
    }
  }
}

// y = lrelu((x - mean) * invstd * gamma + beta)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y,
                                long long total, int c, int hw, float slope) {
  if ((hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const long long total4 = total >> 2;              // 16-byte groups never straddle a channel plane
    const int hw4 = hw >> 2;
    TG_GRID_STRIDE(i, total4) {
      const int ch = (int)((i / hw4) % c);
      const float is = invstd[ch], ga = gamma[ch], mu = mean[ch], be = beta[ch];
      const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = (xv[e] - mu) * is * ga + be; o[e] = v >= 0.f ? v : v * slope; }     // (the scalar form's order)
      reinterpret_cast<f32x4*>(y)[i] = o;
    }
    return;
  }
  TG_GRID_STRIDE(i, total) {
    int ch = (int)((i / hw) % c);
    float v = (x[i] - mean[ch]) * invstd[ch] * gamma[ch] + beta[ch];
    y[i] = v >= 0.f ? v : v * slope;
  }
}

// backward of (BN train + lrelu): per-channel reductions then the input gradient.
//   dz = dy * lrelu'(y);  dbeta = sum dz;  dgamma = sum dz * xhat
//   dx = gamma*invstd * (dz - dbeta/N - xhat * dgamma/N)
__global__ __launch_bounds__(1024) void bn_bwd_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ invstd, int n, int c, int hw,
    float slope, float* __restrict__ sum_dz, float* __restrict__ sum_dz_xhat) {
  __shared__ float sm[16];
  int ch = blockIdx.x;
  float a = 0.f, bq = 0.f;
  const float mu = mean[ch], is = invstd[ch];
  // blockIdx.y = slice of n images (gridDim.y > 1: partial sums at + slice * 2c, bn_sum_slices_kernel adds them)
  const long long base = (long long)blockIdx.y * n * c * hw;
  sum_dz += (long long)blockIdx.y * 2 * c;
  sum_dz_xhat += (long long)blockIdx.y * 2 * c;
  const float* const arrs[3] = {dy + base, y + base, x + base};
  bn_sweep<3>(arrs, n, c, ch, hw, [&](const float (&t)[3]) {
    const float dz = t[1] > 0.f ? t[0] : t[0] * slope;
    a += dz;
    bq += dz * (t[2] - mu) * is;
  });
  float r0 = block_sum(a, sm), r1 = block_sum(bq, sm);
  if (threadIdx.x == 0) { sum_dz[ch] = r0; sum_dz_xhat[ch] = r1; }
}
// out[j] = sum over slices (fixed order) of part[s * 2c + j], j < 2c
__global__ void bn_sum_slices_kernel(const float* __restrict__ part, int ns, int c2, float* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= c2) return;
  float v = 0.f;
  for (int s_ = 0; s_ < ns; ++s_) v += part[(long long)s_ * c2 + j];
  out[j] = v;
}
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                    const float* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ invstd,
                                    const float* __restrict__ gamma,
                                    const float* __restrict__ sum_dz,
                                    const float* __restrict__ sum_dz_xhat, float* __restrict__ dx,
                                    long long total, int c, int hw, float slope, float inv_cnt) {
  if ((hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
    const long long total4 = total >> 2;
    const int hw4 = hw >> 2;
    TG_GRID_STRIDE(i, total4) {
      const int ch = (int)((i / hw4) % c);
      const float mu = mean[ch], is = invstd[ch], gs = gamma[ch] * is;
      const float m0 = sum_dz[ch] * inv_cnt, sx = sum_dz_xhat[ch];
      const f32x4 gv = reinterpret_cast<const f32x4*>(dy)[i], yv = reinterpret_cast<const f32x4*>(y)[i],
                  xv = reinterpret_cast<const f32x4*>(x)[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = yv[e] > 0.f ? gv[e] : gv[e] * slope;
        const float xhat = (xv[e] - mu) * is;
        o[e] = gs * (dz - m0 - xhat * sx * inv_cnt);              // (the scalar form's order)
      }
      reinterpret_cast<f32x4*>(dx)[i] = o;
    }
    return;
  }
  TG_GRID_STRIDE(i, total) {
    int ch = (int)((i / hw) % c);
    float g = dy[i];
    float dz = y[i] > 0.f ? g : g * slope;
    float xhat = (x[i] - mean[ch]) * invstd[ch];
    dx[i] = gamma[ch] * invstd[ch] * (dz - sum_dz[ch] * inv_cnt - xhat * sum_dz_xhat[ch] * inv_cnt);
  }
}

// Linear(K -> 1): y[r] = dot(x[r,:], w) + b ; one block per row
__global__ __launch_bounds__(256) void linear1_fwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ y, int k) {
  __shared__ float sm[4];
  const float* xr = x + (long long)blockIdx.x * k;
  float s = 0.f;
  // (the critic's last layer: 12 rows of 65536 at crop 256 -- one element per thread and iteration was a
  //  chain of 256 dependent round trips, 92 us; 16-byte loads, eight in flight)
  if ((k & 3) == 0 && ((reinterpret_cast<uintptr_t>(xr) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xr);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    const int k4 = k >> 2;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    int i = threadIdx.x;
    for (; i + 7 * 256 < k4; i += 8 * 256) {
      f32x4 a[8], c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = x4[i + u * 256]; c[u] = w4[i + u * 256]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u & 3] += (a[u][0] * c[u][0] + a[u][1] * c[u][1]) + (a[u][2] * c[u][2] + a[u][3] * c[u][3]);
    }
    for (; i < k4; i += 256) { const f32x4 a = x4[i], c = w4[i]; p[0] += (a[0] * c[0] + a[1] * c[1]) + (a[2] * c[2] + a[3] * c[3]); }
    s = (p[0] + p[1]) + (p[2] + p[3]);
  } else {
    for (int i = threadIdx.x; i < k; i += 256) s += xr[i] * w[i];
  }
  float r = block_sum(s, sm);
  if (threadIdx.x == 0) y[blockIdx.x] = r + (b ? b[0] : 0.f);
}
// dx[r,:] = dy[r] * w ; dw (+)= sum_r dy[r] * x[r,:] ; db (+)= sum_r dy[r]
__global__ void linear1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                   const float* __restrict__ dy, float* __restrict__ dx,
                                   float* __restrict__ dw, float* __restrict__ db, int rows, int k,
                                   int accumulate) {
  TG_GRID_STRIDE(i, k) {
    float acc = 0.f;
    float wi = w[i];
    for (int r = 0; r < rows; ++r) {
      float g = dy[r];
      if (dx) dx[(long long)r * k + i] = g * wi;
      acc += g * x[(long long)r * k + i];
    }
    if (dw) dw[i] = accumulate ? dw[i] + acc : acc;
    if (i == 0 && db) {
      float s = 0.f;
      for (int r = 0; r < rows; ++r) s += dy[r];
      db[0] = accumulate ? db[0] + s : s;
    }
  }
}

}  // namespace tg

using namespace tg;
#define ST ((hipStream_t)stream)

extern "C" int tg_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act,
                          tg_stream_t stream) {
  TG_REQUIRE(dy && y && dx && n > 0, TG_E_ARG, "act_bwd: bad argument");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, dy, y, dx, (long long)n, act);
  return check_launch("act_bwd");
}

static int bias_grad_launch(const float* const* dy_list, int nseg, float* db, int n_per_seg, int c,
                            int hw, int accumulate, tg_stream_t stream) {
  TG_REQUIRE(dy_list && db && nseg >= 1 && nseg <= 64 && n_per_seg > 0 && c > 0 && hw > 0, TG_E_ARG,
             "bias_grad: bad argument");
  BiasGradSegs segs{};
  for (int i = 0; i < nseg; ++i) {
    TG_REQUIRE(dy_list[i], TG_E_ARG, "bias_grad: null segment %d", i);
    segs.seg[i] = dy_list[i];
  }
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(db, 0, (size_t)c * sizeof(float), ST);
    TG_REQUIRE(e == hipSuccess, TG_E_HIP, "bias_grad: memset: %s", hipGetErrorString(e));
  }
  const int n = nseg * n_per_seg;
  long long total = (long long)n * hw;
  // slices of 4096 elements (a multiple of 4: the 16-byte path stays aligned), at most ~8192 blocks
  int nslice = (int)((total + 4095) / 4096);
  if (nslice < 1) nslice = 1;
  if (nslice * c > 8192) nslice = 8192 / c > 0 ? 8192 / c : 1;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(c, nslice), dim3(256), 0, ST, segs, n_per_seg, db, n, c, hw,
                     nslice);
  return check_launch("bias_grad");
}

extern "C" int tg_bias_grad_body(const float* const* dz_bases, int nframes,
                                 int64_t layer_stride, int nlayers, float* const* dbs, int n_per_frame, int c,
                                 int hw, tg_stream_t stream) {
  TG_REQUIRE(dz_bases && dbs && nframes >= 1 && nframes <= 64 && nlayers >= 1 && nlayers <= 25 &&
                 n_per_frame > 0 && c > 0 && hw > 0, TG_E_ARG, "bias_grad_body: bad argument");
  BiasBodyArgs a{};
  for (int i = 0; i < nframes; ++i) {
    TG_REQUIRE(dz_bases[i], TG_E_ARG, "bias_grad_body: null frame %d", i);
    a.base[i] = dz_bases[i];
  }
  for (int i = 0; i < nlayers; ++i) {
    TG_REQUIRE(dbs[i], TG_E_ARG, "bias_grad_body: null gradient %d", i);
    a.db[i] = dbs[i];
  }
  const long long total = (long long)nframes * n_per_frame * hw;
  int nslice = (int)((total + 4095) / 4096);
  if (nslice < 1) nslice = 1;
  if ((long long)nslice * c * nlayers > 16384) nslice = (int)(16384 / ((long long)c * nlayers)) > 0 ? (int)(16384 / ((long long)c * nlayers)) : 1;
  hipLaunchKernelGGL(bias_grad_body_kernel, dim3(c, nslice, nlayers), dim3(256), 0, ST, a, nlayers,
                     (long long)layer_stride, nframes, n_per_frame, c, hw, nslice);
  return check_launch("bias_grad_body");
}

extern "C" int tg_bias_grad(const float* dy, float* db, int n, int c, int hw, int accumulate,
                            tg_stream_t stream) {
  TG_REQUIRE(dy, TG_E_ARG, "bias_grad: null pointer");
  return bias_grad_launch(&dy, 1, db, n, c, hw, accumulate, stream);
}

extern "C" int tg_bias_grad_multi(const float* const* dy_list, int nseg, float* db, int n_per_seg,
                                  int c, int hw, int accumulate, tg_stream_t stream) {
  return bias_grad_launch(dy_list, nseg, db, n_per_seg, c, hw, accumulate, stream);
}

extern "C" int tg_maxpool2_bwd(const float* x, const float* dy, float* dx, int nc, int h, int w,
                               tg_stream_t stream) {
  TG_REQUIRE(x && dy && dx && nc > 0 && h >= 2 && w >= 2, TG_E_ARG, "maxpool2_bwd: bad argument");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((long long)nc * h * w)), dim3(256), 0, ST, x,
                     dy, dx, nc, h, w);
  return check_launch("maxpool2_bwd");
}

extern "C" int tg_upsample_bwd(const float* dy, float* dx, int nc, int h, int w, int scale,
                               int up_mode, float mul, tg_stream_t stream) {
  TG_REQUIRE(dy && dx && nc > 0 && h > 0 && w > 0 && scale >= 1, TG_E_ARG, "upsample_bwd: bad argument");
  TG_REQUIRE(up_mode == TG_UP_BICUBIC || up_mode == TG_UP_BILINEAR, TG_E_ARG, "upsample_bwd: mode");
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for((long long)nc * h * w, 8192)), dim3(256), 0,
                     ST, dy, dx, nc, h, w, scale, up_mode, mul);
  return check_launch("upsample_bwd");
}

static int warp_bwd_impl(const float* x, const float* flow, const float* dy, float* dimg,
                         float* dflow, int n, int c, int h, int w, tg_stream_t stream, bool zero_img, int s2d = 1) {
  TG_REQUIRE(x && flow && dy && (dimg || dflow), TG_E_ARG, "backward_warp_bwd: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && h >= 2 && w >= 2 && s2d >= 1 && h % s2d == 0 && w % s2d == 0, TG_E_SHAPE,
             "backward_warp_bwd: shape");
  if (dimg && zero_img) {
    hipError_t e = hipMemsetAsync(dimg, 0, (size_t)n * c * h * w * sizeof(float), ST);
    TG_REQUIRE(e == hipSuccess, TG_E_HIP, "backward_warp_bwd: memset: %s", hipGetErrorString(e));
  }
  dim3 g(cdiv(w, 64), cdiv(h, 4), n), t(256);
  hipLaunchKernelGGL(backward_warp_bwd_kernel, g, t, 0, ST, x, flow, dy, dimg, dflow, n, c, h, w, s2d);
  return check_launch("backward_warp_bwd");
}

extern "C" int tg_backward_warp_bwd(const float* x, const float* flow, const float* dy, float* dimg,
                                    float* dflow, int n, int c, int h, int w, tg_stream_t stream) {
  return warp_bwd_impl(x, flow, dy, dimg, dflow, n, c, h, w, stream, true);
}

// the image gradient is ADDED to what dimg holds (the scatter uses atomic adds anyway): the frame a
// warp reads usually has a gradient already (its own loss term) -- no memset, no separate accumulation pass
extern "C" int tg_backward_warp_bwd_acc(const float* x, const float* flow, const float* dy, float* dimg_acc,
                                        float* dflow, int n, int c, int h, int w, tg_stream_t stream) {
  TG_REQUIRE(dimg_acc, TG_E_ARG, "backward_warp_bwd_acc: null pointer");
  return warp_bwd_impl(x, flow, dy, dimg_acc, dflow, n, c, h, w, stream, false);
}

// dy_s2d: the gradient of space_to_depth(backward_warp(x, flow), scale), (n, scale^2 c, h / scale, w / scale);
// accumulate != 0: the image gradient is ADDED to dimg (tg_backward_warp_bwd_acc), else dimg is overwritten
extern "C" int tg_backward_warp_s2d_bwd(const float* x, const float* flow, const float* dy_s2d, float* dimg,
                                        int accumulate, float* dflow, int n, int c, int h, int w, int scale,
                                        tg_stream_t stream) {
  TG_REQUIRE(!accumulate || dimg, TG_E_ARG, "backward_warp_s2d_bwd: accumulate without dimg");
  return warp_bwd_impl(x, flow, dy_s2d, dimg, dflow, n, c, h, w, stream, !accumulate, scale);
}

extern "C" int tg_depth_to_space(const float* x, float* y, int n, int c, int h, int w, int scale,
                                 tg_stream_t stream) {
  TG_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0 && scale >= 1, TG_E_ARG, "depth_to_space");
  long long total = (long long)n * c * h * scale * w * scale;
  if ((scale == 2 || scale == 4) && w % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    const long long items = (long long)n * c * h * scale * (w / 4);
    if (scale == 2)
      hipLaunchKernelGGL(depth_to_space_vec_kernel<2>, dim3(grid_for(items, 8192)), dim3(256), 0, ST, x, y, n, c, h, w,
                         (const float*)nullptr, 0.f);
    else
      hipLaunchKernelGGL(depth_to_space_vec_kernel<4>, dim3(grid_for(items, 8192)), dim3(256), 0, ST, x, y, n, c, h, w,
                         (const float*)nullptr, 0.f);
    return check_launch("depth_to_space");
  }
  hipLaunchKernelGGL(depth_to_space_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, ST, x, y, n, c,
                     h, w, scale);
  return check_launch("depth_to_space");
}

extern "C" int tg_depth_to_space_act_bwd_supported(const float* x, const float* act_y, const float* y, int w, int scale) {
  return (scale == 2 || scale == 4) && w % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)act_y) & 15) == 0;
}

extern "C" int tg_depth_to_space_act_bwd(const float* x, const float* act_y, int act, float* y, int n, int c, int h,
                                         int w, int scale, tg_stream_t stream) {
  TG_REQUIRE(x && y && act_y && n > 0 && c > 0 && h > 0 && w > 0, TG_E_ARG, "depth_to_space_act_bwd: bad argument");
  TG_REQUIRE(act == TG_ACT_RELU || act == TG_ACT_LRELU02, TG_E_ARG, "depth_to_space_act_bwd: act=%d (relu | lrelu)", act);
  TG_REQUIRE(tg_depth_to_space_act_bwd_supported(x, act_y, y, w, scale), TG_E_ARG,
             "depth_to_space_act_bwd: scale 2 | 4, w %% 4 == 0, 16-byte aligned tensors");
  const float slope = act == TG_ACT_RELU ? 0.f : 0.2f;
  const long long items = (long long)n * c * h * scale * (w / 4);
  if (scale == 2)
    hipLaunchKernelGGL(depth_to_space_vec_kernel<2>, dim3(grid_for(items, 8192)), dim3(256), 0, ST, x, y, n, c, h, w, act_y, slope);
  else
    hipLaunchKernelGGL(depth_to_space_vec_kernel<4>, dim3(grid_for(items, 8192)), dim3(256), 0, ST, x, y, n, c, h, w, act_y, slope);
  return check_launch("depth_to_space_act_bwd");
}

extern "C" int tg_charbonnier(const float* x, const float* y, int64_t n, float eps, float loss_scale,
                              float* loss_accum, float grad_scale, float* dx, tg_stream_t stream) {
  TG_REQUIRE(x && y && n > 0 && (loss_accum || dx), TG_E_ARG, "charbonnier: bad argument");
  hipLaunchKernelGGL(charbonnier_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, x, y,
                     (long long)n, eps, loss_scale, loss_accum, grad_scale, dx);
  return check_launch("charbonnier");
}

extern "C" int tg_channel_norm(const float* x, const float* mean, const float* stdv, float* y, int n,
                               int c, int64_t hw, tg_stream_t stream) {
  TG_REQUIRE(x && stdv && y, TG_E_ARG, "channel_norm: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && hw > 0, TG_E_SHAPE, "channel_norm: n=%d c=%d hw=%lld", n, c, (long long)hw);
  long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(channel_norm_kernel, dim3(grid_for(total)), dim3(256), 0, ST, x, mean, stdv, y,
                     total, c, (long long)hw);
  return check_launch("channel_norm");
}

extern "C" int tg_cosine_loss(const float* a, const float* b, int n, int c, int64_t hw, float eps,
                              float loss_scale, float* loss_accum, float grad_scale, float* da,
                              tg_stream_t stream) {
  TG_REQUIRE(a && b && (loss_accum || da), TG_E_ARG, "cosine_loss: bad argument");
  TG_REQUIRE(n > 0 && c > 0 && hw > 0, TG_E_SHAPE, "cosine_loss: n=%d c=%d hw=%lld", n, c, (long long)hw);
  long long npix = (long long)n * hw;
  hipLaunchKernelGGL(cosine_loss_kernel, dim3(grid_for(npix, 2048)), dim3(256), 0, ST, a, b, npix, c,
                     (long long)hw, eps, loss_scale, loss_accum, grad_scale, da);
  return check_launch("cosine_loss");
}

extern "C" int tg_pixel_loss(const float* x, const float* y, int64_t n, int mode, float loss_scale,
                             float* loss_accum, float grad_scale, float* dx, tg_stream_t stream) {
  TG_REQUIRE(x && y && n > 0 && (loss_accum || dx), TG_E_ARG, "pixel_loss: bad argument");
  TG_REQUIRE(mode == TG_LOSS_L1 || mode == TG_LOSS_MSE, TG_E_ARG, "pixel_loss: mode=%d", mode);
  hipLaunchKernelGGL(pixel_loss_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, x, y, (long long)n,
                     mode, loss_scale, loss_accum, grad_scale, dx);
  return check_launch("pixel_loss");
}

extern "C" int tg_bce_logits(const float* x, int64_t n, float target, float scale, float* stats3,
                             float grad_scale, float* dx, tg_stream_t stream) {
  TG_REQUIRE(x && n > 0 && (stats3 || dx), TG_E_ARG, "bce_logits: bad argument");
  hipLaunchKernelGGL(bce_logits_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, ST, x, (long long)n,
                     target, scale, stats3, grad_scale, dx);
  return check_launch("bce_logits");
}

extern "C" int tg_lsgan_loss(const float* x, int64_t n, float target, float scale, float* stats3,
                             float grad_scale, float* dx, tg_stream_t stream) {
  TG_REQUIRE(x && n > 0 && (stats3 || dx), TG_E_ARG, "lsgan_loss: bad argument");
  hipLaunchKernelGGL(bce_logits_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, ST, x, (long long)n,
                     target, scale, stats3, grad_scale, dx);
  return check_launch("lsgan_loss");
}

extern "C" int tg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step,
                            tg_stream_t stream) {
  TG_REQUIRE(p && g && m && v && n > 0 && step >= 1, TG_E_ARG, "adam_step: bad argument");
  float bc1 = 1.f - powf(beta1, (float)step);
  float sbc2 = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, g, m, v, (long long)n, lr,
                     beta1, beta2, eps, weight_decay, bc1, sbc2, (const float*)nullptr);
  return check_launch("adam_step");
}

extern "C" int tg_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int step,
                                    const float* skip_if_nonzero, tg_stream_t stream) {
  TG_REQUIRE(p && g && m && v && n > 0 && step >= 1, TG_E_ARG, "adam_step_guarded: bad argument");
  float bc1 = 1.f - powf(beta1, (float)step);
  float sbc2 = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, g, m, v, (long long)n, lr,
                     beta1, beta2, eps, weight_decay, bc1, sbc2, skip_if_nonzero);
  return check_launch("adam_step_guarded");
}

__global__ void fault_to_slot_kernel(const int32_t* __restrict__ err, float* __restrict__ slot) {
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) *slot += 1.0f;
}

extern "C" int tg_fault_to_slot(const int32_t* fault_counter, float* slot, tg_stream_t stream) {
  TG_REQUIRE(fault_counter && slot, TG_E_ARG, "fault_to_slot: null pointer");
  hipLaunchKernelGGL(fault_to_slot_kernel, dim3(1), dim3(1), 0, ST, fault_counter, slot);
  return check_launch("fault_to_slot");
}

extern "C" int tg_div_scalar(float* y, const float* x, float d, int64_t n, tg_stream_t stream) {
  TG_REQUIRE(y && x && n > 0 && d != 0.f, TG_E_ARG, "div_scalar: bad argument");
  hipLaunchKernelGGL(div_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, ST, y, x, d, (long long)n);
  return check_launch("div_scalar");
}

extern "C" int tg_axpy(float* y, const float* x, float a, int64_t n, tg_stream_t stream) {
  TG_REQUIRE(y && x && n > 0, TG_E_ARG, "axpy: bad argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, ST, y, x, a, (long long)n);
  return check_launch("axpy");
}

extern "C" int tg_bn_lrelu_train_fwd(const float* x, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum,
                                     float eps, float slope, float* y, float* save_mean,
                                     float* save_invstd, int n, int c, int hw, tg_stream_t stream) {
  TG_REQUIRE(x && gamma && beta && y && save_mean && save_invstd, TG_E_ARG, "bn_fwd: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && hw > 0 && (long long)n * hw > 1, TG_E_SHAPE, "bn_fwd: shape");
  // One block per channel leaves most of the device idle when c is small and the planes are large (the
  // critic's 64-channel blocks on 24 HR clips: 64 blocks for 100 MB, 110 us).  Then: equal slices of the
  // batch per block, (mean, M2) pairs merged with Chan's formula in slice order (deterministic).  The
  // partials live in `y`, which the apply kernel overwrites afterwards.
  const int ns = bn_slices(n, c, hw);
  if (ns > 1) {
    hipLaunchKernelGGL(bn_local_stats_kernel, dim3(c, ns), dim3(bn_threads(n / ns, hw)), 0, ST, x, n / ns, c, hw, y);
    hipLaunchKernelGGL(bn_merge_stats_kernel, dim3(cdiv(c, 256)), dim3(256), 0, ST, (const float*)y, ns,
                       (float)(n / ns) * (float)hw, eps, momentum, save_mean, save_invstd, running_mean, running_var, c);
  } else {
    hipLaunchKernelGGL(bn_stats_kernel, dim3(c), dim3(bn_threads(n, hw)), 0, ST, x, n, c, hw, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var);
  }
  long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, ST, x, save_mean,
                     save_invstd, gamma, beta, y, total, c, hw, slope);
  return check_launch("bn_lrelu_train_fwd");
}

extern "C" int tg_bn_lrelu_train_bwd(const float* x, const float* y, const float* dy,
                                     const float* gamma, const float* save_mean,
                                     const float* save_invstd, float slope, float* dx,
                                     float* dgamma, float* dbeta, int accumulate, float* scratch2c,
                                     int n, int c, int hw, tg_stream_t stream) {
  TG_REQUIRE(x && y && dy && gamma && save_mean && save_invstd && scratch2c, TG_E_ARG,
             "bn_bwd: null pointer");
  float* s0 = scratch2c;
  float* s1 = scratch2c + c;
  const int ns = dx ? bn_slices(n, c, hw) : 1;      // (the partial sums live in dx until the apply kernel overwrites it)
  if (ns > 1) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(c, ns), dim3(bn_threads(n / ns, hw)), 0, ST, x, y, dy, save_mean,
                       save_invstd, n / ns, c, hw, slope, dx, dx + c);
    hipLaunchKernelGGL(bn_sum_slices_kernel, dim3(cdiv(2 * c, 256)), dim3(256), 0, ST, (const float*)dx, ns, 2 * c, s0);
  } else {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(c), dim3(bn_threads(n, hw)), 0, ST, x, y, dy, save_mean, save_invstd,
                       n, c, hw, slope, s0, s1);
  }
  long long total = (long long)n * c * hw;
  if (dx)
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, ST, x, y, dy, save_mean,
                       save_invstd, gamma, s0, s1, dx, total, c, hw, slope,
                       1.0f / ((float)n * (float)hw));
  if (dgamma) {
    if (accumulate) {
      hipLaunchKernelGGL(axpy_kernel, dim3(1), dim3(256), 0, ST, dgamma, (const float*)s1, 1.0f, (long long)c);
      hipLaunchKernelGGL(axpy_kernel, dim3(1), dim3(256), 0, ST, dbeta, (const float*)s0, 1.0f, (long long)c);
    } else {
      (void)hipMemcpyAsync(dgamma, s1, c * sizeof(float), hipMemcpyDeviceToDevice, ST);
      (void)hipMemcpyAsync(dbeta, s0, c * sizeof(float), hipMemcpyDeviceToDevice, ST);
    }
  }
  return check_launch("bn_lrelu_train_bwd");
}

extern "C" int tg_linear1_fwd(const float* x, const float* w, const float* b, float* y, int rows,
                              int k, tg_stream_t stream) {
  TG_REQUIRE(x && w && y && rows > 0 && k > 0, TG_E_ARG, "linear1_fwd: bad argument");
  hipLaunchKernelGGL(linear1_fwd_kernel, dim3(rows), dim3(256), 0, ST, x, w, b, y, k);
  return check_launch("linear1_fwd");
}

extern "C" int tg_linear1_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                              float* db, int rows, int k, int accumulate, tg_stream_t stream) {
  TG_REQUIRE(x && w && dy && rows > 0 && k > 0, TG_E_ARG, "linear1_bwd: bad argument");
  hipLaunchKernelGGL(linear1_bwd_kernel, dim3(grid_for(k)), dim3(256), 0, ST, x, w, dy, dx, dw, db, rows,
                     k, accumulate);
  return check_launch("linear1_bwd");
}

// ---- BD degradation: per-channel Gaussian blur + stride-s decimation -------------
// Replaces downsample_bd (codes/utils/data_utils.py:30-53; create_kernel :11-27):
// y[p][oy][ox] = sum_{i,j} k[i][j] * x[p][oy*s + i - pad_t][ox*s + j - pad_l]
// pad = 0: valid conv (training, base_model.py:75); pad = 1: 'reflect' padding of
// (k-1)//2 before / k-1-(k-1)//2 after (testing, data_utils.py:40-48).
namespace tg {
__global__ void downsample_bd_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                     float* __restrict__ y, int nc, int h, int w, int ks, int s,
                                     int pad, int oh, int ow) {
  const long long total = (long long)nc * oh * ow;
  const int pb = pad ? (ks - 1) / 2 : 0;
  TG_GRID_STRIDE(i, total) {
    int ox = (int)(i % ow); long long t = i / ow;
    int oy = (int)(t % oh); int p = (int)(t / oh);
    const float* src = x + (long long)p * h * w;
    float acc = 0.f;
    for (int a = 0; a < ks; ++a) {
      int yy = oy * s + a - pb;
      yy = yy < 0 ? -yy : (yy > h - 1 ? 2 * (h - 1) - yy : yy);
      for (int b = 0; b < ks; ++b) {
        int xx = ox * s + b - pb;
        xx = xx < 0 ? -xx : (xx > w - 1 ? 2 * (w - 1) - xx : xx);
        acc += k[a * ks + b] * src[(long long)yy * w + xx];
      }
    }
    y[i] = acc;
  }
}
}  // namespace tg

extern "C" int tg_downsample_bd(const float* x, const float* kernel2d, float* y, int nc, int h,
                                int w, int ksize, int scale, int pad, tg_stream_t stream) {
  TG_REQUIRE(x && kernel2d && y, TG_E_ARG, "downsample_bd: null pointer");
  TG_REQUIRE(nc > 0 && ksize >= 1 && scale >= 1 && h >= ksize && w >= ksize, TG_E_SHAPE,
             "downsample_bd: nc=%d h=%d w=%d k=%d s=%d", nc, h, w, ksize, scale);
  int oh = pad ? (h - 1) / scale + 1 : (h - ksize) / scale + 1;
  int ow = pad ? (w - 1) / scale + 1 : (w - ksize) / scale + 1;
  hipLaunchKernelGGL(tg::downsample_bd_kernel, dim3(tg::grid_for((long long)nc * oh * ow)), dim3(256),
                     0, ST, x, kernel2d, y, nc, h, w, ksize, scale, pad, oh, ow);
  return tg::check_launch("downsample_bd");
}

// ---- SyncBatchNorm building blocks (nn.SyncBatchNorm.convert_sync_batchnorm,
// codes/models/base_model.py:133): the per-channel reductions are exposed separately so
// the host can exchange them over RCCL between the two halves: forward = all-gather of the
// per-rank (mean, centred M2) pairs merged with Chan's formula (what torch's SyncBatchNorm
// does with mean / invstd / count), backward = all-reduce of (sum dz, sum dz*xhat).
// One small collective per BN layer and direction.
namespace tg {
// local statistics of this rank's slice of the batch: mean and CENTRED sum of squares
// (two passes over the plane, like bn_stats_kernel) -> stats2c = [mean | M2]
__global__ __launch_bounds__(1024) void bn_local_stats_kernel(const float* __restrict__ x, int n, int c,
                                                             int hw, float* __restrict__ stats2c) {
  // blockIdx.y = slice of the batch (gridDim.y equal slices of n images each; 1 in the per-rank form):
  // slice s writes its (mean, M2) pair at stats2c + s * 2c -- the layout bn_merge_stats_kernel merges
  __shared__ float sm[16];
  __shared__ float s_mean;
  int ch = blockIdx.x;
  x += (long long)blockIdx.y * n * c * hw;
  stats2c += (long long)blockIdx.y * 2 * c;
  float s = 0.f;
  const float* const xs[1] = {x};
  bn_sweep<1>(xs, n, c, ch, hw, [&](const float (&t)[1]) { s += t[0]; });
  float tot = block_sum(s, sm);
  if (threadIdx.x == 0) s_mean = tot / ((float)n * (float)hw);
  __syncthreads();
  const float mean = s_mean;
  float q = 0.f;
  bn_sweep<1>(xs, n, c, ch, hw, [&](const float (&t)[1]) { const float d = t[0] - mean; q += d * d; });
  float sq = block_sum(q, sm);
  if (threadIdx.x == 0) { stats2c[ch] = mean; stats2c[c + ch] = sq; }
}
// Chan et al. merge of `world` equally sized partitions (DistributedSampler: equal per-rank
// batches): mean = avg(mean_r); M2 = sum M2_r + cnt_r * sum (mean_r - mean)^2, ranks in
// index order on every rank (bit-identical statistics everywhere).  No E[x^2]-mean^2
// cancellation, no clamp: with world = 1 this is exactly bn_stats_kernel's result.
__global__ void bn_merge_stats_kernel(const float* __restrict__ gathered, int world, float cnt_r,
                                      float eps, float momentum, float* __restrict__ mean,
                                      float* __restrict__ invstd, float* __restrict__ run_mean,
                                      float* __restrict__ run_var, int c) {
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  float m = 0.f;
  for (int r = 0; r < world; ++r) m += gathered[(long long)r * 2 * c + ch];
  m = world == 1 ? m : m / (float)world;
  float m2 = 0.f;
  for (int r = 0; r < world; ++r) {
    float d = gathered[(long long)r * 2 * c + ch] - m;
    m2 += gathered[(long long)r * 2 * c + c + ch] + cnt_r * d * d;
  }
  const float cnt = cnt_r * (float)world;
  float var = m2 / cnt;
  mean[ch] = m;
  invstd[ch] = 1.0f / sqrtf(var + eps);
  if (run_mean) {
    run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * m;
    run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (var * cnt / (cnt - 1.f));
  }
}
}  // namespace tg

extern "C" int tg_bn_local_stats(const float* x, float* stats2c, int n, int c, int hw,
                                 tg_stream_t stream) {
  TG_REQUIRE(x && stats2c && n > 0 && c > 0 && hw > 0, TG_E_ARG, "bn_local_stats: bad argument");
  hipLaunchKernelGGL(tg::bn_local_stats_kernel, dim3(c), dim3(tg::bn_threads(n, hw)), 0, ST, x, n, c, hw, stats2c);
  return tg::check_launch("bn_local_stats");
}

extern "C" int tg_bn_merge_stats(const float* gathered, int world, float count_per_rank, float eps,
                                 float momentum, float* mean, float* invstd, float* running_mean,
                                 float* running_var, int c, tg_stream_t stream) {
  TG_REQUIRE(gathered && mean && invstd && c > 0 && world >= 1 && count_per_rank * world > 1.f, TG_E_ARG,
             "bn_merge_stats: bad argument");
  hipLaunchKernelGGL(tg::bn_merge_stats_kernel, dim3(tg::cdiv(c, 256)), dim3(256), 0, ST, gathered, world,
                     count_per_rank, eps, momentum, mean, invstd, running_mean, running_var, c);
  return tg::check_launch("bn_merge_stats");
}

extern "C" int tg_bn_lrelu_apply(const float* x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, float slope, float* y,
                                 int n, int c, int hw, tg_stream_t stream) {
  TG_REQUIRE(x && mean && invstd && gamma && beta && y, TG_E_ARG, "bn_lrelu_apply: null pointer");
  long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(tg::bn_apply_kernel, dim3(tg::grid_for(total)), dim3(256), 0, ST, x, mean, invstd,
                     gamma, beta, y, total, c, hw, slope);
  return tg::check_launch("bn_lrelu_apply");
}

extern "C" int tg_bn_lrelu_bwd_reduce(const float* x, const float* y, const float* dy,
                                      const float* mean, const float* invstd, float slope,
                                      float* sums2c, int n, int c, int hw, tg_stream_t stream) {
  TG_REQUIRE(x && y && dy && mean && invstd && sums2c, TG_E_ARG, "bn_lrelu_bwd_reduce: null pointer");
  hipLaunchKernelGGL(tg::bn_bwd_reduce_kernel, dim3(c), dim3(tg::bn_threads(n, hw)), 0, ST, x, y, dy, mean, invstd, n, c,
                     hw, slope, sums2c, sums2c + c);
  return tg::check_launch("bn_lrelu_bwd_reduce");
}

extern "C" int tg_bn_lrelu_bwd_apply(const float* x, const float* y, const float* dy,
                                     const float* mean, const float* invstd, const float* gamma,
                                     const float* sums2c, float slope, float inv_count, float* dx,
                                     int n, int c, int hw, tg_stream_t stream) {
  TG_REQUIRE(x && y && dy && mean && invstd && gamma && sums2c && dx, TG_E_ARG,
             "bn_lrelu_bwd_apply: null pointer");
  long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(tg::bn_bwd_apply_kernel, dim3(tg::grid_for(total)), dim3(256), 0, ST, x, y, dy,
                     mean, invstd, gamma, sums2c, sums2c + c, dx, total, c, hw, slope, inv_count);
  return tg::check_launch("bn_lrelu_bwd_apply");
}

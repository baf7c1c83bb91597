// Data-movement kernels of the training step that replace chains of ATen slice / flip / cat /
// zero-fill launches (21 % of the step's launches were at::native copy / fill / flip /
// cat kernels).  Each is one launch, reads every source element once and writes every
// destination element once.
//
//   tg_time_gather      out[n][k] = x[n][idx[k]] (0 if idx<0) ping-pong augmentation
//                                                           (vsrgan_model.py:112-119), the forward /
//                                                           reversed halves of the ping-pong loss
//                                                           (:246-247), `data[:, :t]` (tecogan_nets.py:437)
//   tg_pingpong_grad    gradient of the ping-pong loss routed onto the 2*te-1 frames: +g on the
//                       first te-1, 0 on the middle one, -flip(g) on the last te-1
//   tg_d_assemble_fwd   the discriminator's 27-channel input (tecogan_nets.py:440-463): frame
//                       triplets in rrrgggbbb order, the warped triplets centre-cropped and
//                       zero-padded, the bicubic condition
//   tg_d_assemble_bwd   its adjoint: gradient w.r.t. the frames (n, T, c, H, W; zero for the
//                       frames beyond t) and w.r.t. the warped frames (n*t, c, H, W)
#include "tg_common.h"

namespace tg {

struct TimeIdx { int idx[64]; };

__global__ __launch_bounds__(256) void time_gather_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          TimeIdx ti, int T, int K, long long inner4,
                                                          long long total4) {
  // float4 granularity (inner % 4 == 0 is checked by the launcher)
  const f32x4* xs = reinterpret_cast<const f32x4*>(x);
  f32x4* yd = reinterpret_cast<f32x4*>(y);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i % inner4, r = i / inner4;
    const int k = (int)(r % K);
    const long long n = r / K;
    const int f = ti.idx[k];
    yd[i] = f >= 0 ? xs[(n * T + f) * inner4 + e] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// y (b, a, inner) = x (a, b, inner) transposed over the two leading axes
__global__ __launch_bounds__(256) void transpose01_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int a, int b, long long inner4, long long total4) {
  const f32x4* xs = reinterpret_cast<const f32x4*>(x);
  f32x4* yd = reinterpret_cast<f32x4*>(y);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i % inner4, r = i / inner4;
    const int ia = (int)(r % a);
    const long long ib = r / a;
    yd[i] = xs[((long long)ia * b + ib) * inner4 + e];
  }
}

struct StackSrc { const float* p[64]; };

// y (n, k, inner): y[:, j] = src_j (n, inner)
__global__ __launch_bounds__(256) void stack_time_kernel(StackSrc src, float* __restrict__ y, int k,
                                                         long long inner4, long long total4) {
  f32x4* yd = reinterpret_cast<f32x4*>(y);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i % inner4, r = i / inner4;
    const int j = (int)(r % k);
    const long long n = r / k;
    yd[i] = reinterpret_cast<const f32x4*>(src.p[j])[n * inner4 + e];
  }
}

__global__ __launch_bounds__(256) void pingpong_grad_kernel(const float* __restrict__ g, float* __restrict__ out,
                                                            int te, long long inner, long long total) {
  const int T = 2 * te - 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i % inner, r = i / inner;
    const int f = (int)(r % T);
    const long long n = r / T;
    float v = 0.f;
    if (f < te - 1) v = g[(n * (te - 1) + f) * inner + e];
    else if (f >= te) v = -g[(n * (te - 1) + (2 * te - 2 - f)) * inner + e];
    out[i] = v;
  }
}

// x (n_clip, 9c, H, W); clip k = (n, j) holds frames f = 3j .. 3j+2 of clip n.
// channel layout: section s in {frames, warped, cond} x (ch*3 + m)   [m = frame within the triplet]
__global__ __launch_bounds__(256) void d_assemble_fwd_kernel(const float* __restrict__ data, int t_data,
                                                             const float* __restrict__ warped,
                                                             const float* __restrict__ cond, int t_cond,
                                                             float* __restrict__ x, int n, int t, int c,
                                                             int H, int W, int pad, int crop,
                                                             long long total) {
  const long long hw = (long long)H * W;
  const int ncl = t / 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W); long long r = i / W;
    const int yy = (int)(r % H); r /= H;
    const int c27 = (int)(r % (9 * c));
    const long long k = r / (9 * c);
    const int s = c27 / (3 * c), q = c27 - s * 3 * c;
    const int ch = q / 3, m = q - ch * 3;
    const long long nn = k / ncl;
    const int f = (int)(k - nn * ncl) * 3 + m;
    const long long p = (long long)yy * W + xx;
    float v;
    if (s == 0) v = data[((nn * t_data + f) * c + ch) * hw + p];
    else if (s == 2) v = cond[((nn * t_cond + f) * c + ch) * hw + p];
    else {
      const bool in = yy >= pad && yy < pad + crop && xx >= pad && xx < pad + crop;
      v = in ? warped[((nn * t + f) * c + ch) * hw + p] : 0.f;
    }
    x[i] = v;
  }
}

// g (n_clip, 9c, H, W) -> g_data (n, t_data, c, H, W) [frames >= t: 0], g_warped (n*t, c, H, W)
__global__ __launch_bounds__(256) void d_assemble_bwd_kernel(const float* __restrict__ g,
                                                             float* __restrict__ g_data, int t_data,
                                                             float* __restrict__ g_warped, int n, int t,
                                                             int c, int H, int W, int pad, int crop,
                                                             long long total) {
  const long long hw = (long long)H * W;
  const int ncl = t / 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // i walks g_data: (n, t_data, c, H, W)
    const long long p = i % hw; long long r = i / hw;
    const int ch = (int)(r % c); r /= c;
    const int f = (int)(r % t_data);
    const long long nn = r / t_data;
    if (f >= t) { g_data[i] = 0.f; continue; }
    const long long k = nn * ncl + f / 3;
    const int m = f % 3;
    const float* gk = g + (k * 9 * c) * hw + p;
    g_data[i] = gk[(long long)(ch * 3 + m) * hw];
    const int yy = (int)(p / W), xx = (int)(p - (long long)yy * W);
    const bool in = yy >= pad && yy < pad + crop && xx >= pad && xx < pad + crop;
    g_warped[((nn * t + f) * c + ch) * hw + p] = in ? gk[(long long)(3 * c + ch * 3 + m) * hw] : 0.f;
  }
}

// out[i] = (accumulate ? out[i] : 0) + (idx[i] < n_src ? src[idx[i]] : 0)
__global__ __launch_bounds__(256) void index_gather_kernel(const float* __restrict__ src,
                                                           const long long* __restrict__ idx,
                                                           float* __restrict__ out, long long n_out,
                                                           long long n_src, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out;
       i += (long long)gridDim.x * blockDim.x) {
    const long long j = idx[i];
    const float v = (j >= 0 && j < n_src) ? src[j] : 0.f;
    out[i] = accumulate ? out[i] + v : v;
  }
}

static inline unsigned grid_for_n(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace tg

using namespace tg;

extern "C" int tg_time_gather(const float* x, float* y, const int* idx_host, int n, int t_in, int k,
                              int64_t inner, tg_stream_t stream) {
  TG_REQUIRE(x && y && idx_host, TG_E_ARG, "time_gather: null pointer");
  TG_REQUIRE(n > 0 && t_in > 0 && k > 0 && k <= 64 && inner > 0 && inner % 4 == 0, TG_E_SHAPE,
             "time_gather: n=%d t=%d k=%d (<=64) inner=%lld (%%4)", n, t_in, k, (long long)inner);
  TG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, TG_E_ARG, "time_gather: 16-byte alignment");
  TimeIdx ti{};
  for (int i = 0; i < k; ++i) {
    TG_REQUIRE(idx_host[i] < t_in, TG_E_ARG, "time_gather: idx[%d]=%d of %d", i, idx_host[i], t_in);
    ti.idx[i] = idx_host[i];
  }
  const long long inner4 = inner / 4, total4 = (long long)n * k * inner4;
  hipLaunchKernelGGL(time_gather_kernel, dim3(grid_for_n(total4)), dim3(256), 0, (hipStream_t)stream, x, y, ti,
                     t_in, k, inner4, total4);
  return check_launch("time_gather");
}

extern "C" int tg_pingpong_grad(const float* g, float* out, int n, int te, int64_t inner, tg_stream_t stream) {
  TG_REQUIRE(g && out && n > 0 && te >= 2 && inner > 0, TG_E_ARG, "pingpong_grad: bad argument");
  const long long total = (long long)n * (2 * te - 1) * inner;
  hipLaunchKernelGGL(pingpong_grad_kernel, dim3(grid_for_n(total)), dim3(256), 0, (hipStream_t)stream, g, out, te,
                     (long long)inner, total);
  return check_launch("pingpong_grad");
}

extern "C" int tg_d_assemble_fwd(const float* data, int t_data, const float* warped, const float* cond,
                                 int t_cond, float* x, int n, int t, int c, int h, int w, int pad, int crop,
                                 tg_stream_t stream) {
  TG_REQUIRE(data && warped && cond && x, TG_E_ARG, "d_assemble_fwd: null pointer");
  TG_REQUIRE(n > 0 && t > 0 && t % 3 == 0 && t <= t_data && t <= t_cond && c > 0 && h > 0 && w > 0 && pad >= 0 &&
                 crop >= 0 && pad + crop <= h && pad + crop <= w, TG_E_SHAPE,
             "d_assemble_fwd: n=%d t=%d (%%3, <= %d, %d) c=%d h=%d w=%d pad=%d crop=%d", n, t, t_data, t_cond, c, h, w, pad, crop);
  const long long total = (long long)n * (t / 3) * 9 * c * h * w;
  hipLaunchKernelGGL(d_assemble_fwd_kernel, dim3(grid_for_n(total)), dim3(256), 0, (hipStream_t)stream, data, t_data,
                     warped, cond, t_cond, x, n, t, c, h, w, pad, crop, total);
  return check_launch("d_assemble_fwd");
}

extern "C" int tg_d_assemble_bwd(const float* g, float* g_data, int t_data, float* g_warped, int n, int t, int c,
                                 int h, int w, int pad, int crop, tg_stream_t stream) {
  TG_REQUIRE(g && g_data && g_warped, TG_E_ARG, "d_assemble_bwd: null pointer");
  TG_REQUIRE(n > 0 && t > 0 && t % 3 == 0 && t <= t_data && c > 0 && h > 0 && w > 0 && pad >= 0 && crop >= 0 &&
                 pad + crop <= h && pad + crop <= w, TG_E_SHAPE, "d_assemble_bwd: shape");
  const long long total = (long long)n * t_data * c * h * w;
  hipLaunchKernelGGL(d_assemble_bwd_kernel, dim3(grid_for_n(total)), dim3(256), 0, (hipStream_t)stream, g, g_data,
                     t_data, g_warped, n, t, c, h, w, pad, crop, total);
  return check_launch("d_assemble_bwd");
}

extern "C" int tg_transpose01(const float* x, float* y, int a, int b, int64_t inner, tg_stream_t stream) {
  TG_REQUIRE(x && y && a > 0 && b > 0 && inner > 0 && inner % 4 == 0, TG_E_ARG, "transpose01: a=%d b=%d inner=%lld (%%4)",
             a, b, (long long)inner);
  TG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, TG_E_ARG, "transpose01: 16-byte alignment");
  const long long inner4 = inner / 4, total4 = (long long)a * b * inner4;
  hipLaunchKernelGGL(transpose01_kernel, dim3(grid_for_n(total4)), dim3(256), 0, (hipStream_t)stream, x, y, a, b,
                     inner4, total4);
  return check_launch("transpose01");
}

extern "C" int tg_stack_time(const float* const* src_host, int k, float* y, int n, int64_t inner,
                             tg_stream_t stream) {
  TG_REQUIRE(src_host && y && k > 0 && k <= 64 && n > 0 && inner > 0 && inner % 4 == 0, TG_E_ARG,
             "stack_time: k=%d (<=64) n=%d inner=%lld (%%4)", k, n, (long long)inner);
  StackSrc s{};
  for (int j = 0; j < k; ++j) {
    TG_REQUIRE(src_host[j] && ((uintptr_t)src_host[j] % 16) == 0, TG_E_ARG, "stack_time: source %d null / unaligned", j);
    s.p[j] = src_host[j];
  }
  const long long inner4 = inner / 4, total4 = (long long)n * k * inner4;
  hipLaunchKernelGGL(stack_time_kernel, dim3(grid_for_n(total4)), dim3(256), 0, (hipStream_t)stream, s, y, k, inner4,
                     total4);
  return check_launch("stack_time");
}

extern "C" int tg_index_gather(const float* src, const int64_t* idx, float* out, int64_t n_out, int64_t n_src,
                               int accumulate, tg_stream_t stream) {
  TG_REQUIRE(src && idx && out && n_out > 0 && n_src > 0, TG_E_ARG, "index_gather: bad argument");
  hipLaunchKernelGGL(index_gather_kernel, dim3(grid_for_n(n_out)), dim3(256), 0, (hipStream_t)stream, src,
                     (const long long*)idx, out, (long long)n_out, (long long)n_src, accumulate);
  return check_launch("index_gather");
}

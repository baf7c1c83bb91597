// 3x3 / stride 1 / pad 1 convolution as an fp32-MFMA implicit GEMM for gfx950.
//
//   D[oc][pixel] = sum_k  W[oc][k] * X[k][pixel],   k = (tap, cin)
//
// One wave owns a 32-pixel row segment x (NT x 32) output channels and issues
// v_mfma_f32_32x32x2_f32 with A = weights (rows = oc), B = input (cols =
// pixels).  With that orientation each accumulator register holds one output
// channel for 32 consecutive pixels, so the NCHW epilogue store is a 128-byte
// contiguous segment per half-wave -- no transpose, no LDS round trip.
//
// Workgroup = WM x WN waves: WM consecutive image rows, WN groups of NT*32
// output channels.  Input channels are streamed through LDS in chunks of 8
// (double buffered, one barrier per chunk): the chunk's halo patch
// [8][WM+2][34] stays planar (NCHW order), which is exactly the B-operand
// order (lane&31 -> consecutive x, lane>>5 -> next channel), so every
// ds_read_b32 is conflict free; weights sit as [tap][8][OCB] so the A operand
// (lane&31 -> consecutive oc) is conflict free too.  fp32 MFMA consumes one A
// and one B VGPR per 64 cycles, so plain 4-byte LDS reads are 4x below the
// LDS roof; the kernel is MFMA-issue bound by construction.
//
// Reference ops replaced: nn.Conv2d(.,.,3,1,1)+bias+act(+residual) at
// codes/models/networks/tecogan_nets.py:23-65, 92-98, 111-113, 367-369 and the
// torch.cat at :71 / :141 (two-source input).
#include "tg_common.h"

namespace tg {

constexpr int TW = 32;          // pixels per row segment (MFMA N)
constexpr int PW = TW + 2;      // patch width incl. halo
constexpr int RS = 36;          // LDS row stride (floats)

struct Conv3x3Args {
  const float* x;
  const float* x2;
  const float* wpk;
  const float* bias;
  const float* res;
  float* y;
  long long x_ns, x2_ns, res_ns, y_ns;
  int c1, cin, cout, h, w, act;
  int tiles_x, tiles_y, nchunk, nocg;
};

// pack OIHW (or IOHW for transposed convs) -> [ocg][chunk][tap][CK][OCB]
__global__ void pack3x3_kernel(const float* __restrict__ w, float* __restrict__ out,
                               int cin, int cout, int ocb, int nchunk, int nocg,
                               int transposed) {
  int total = nocg * nchunk * 9 * CK * ocb;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    int o = i % ocb;
    int t = i / ocb;
    int c = t % CK; t /= CK;
    int tap = t % 9; t /= 9;
    int ch = t % nchunk;
    int g = t / nchunk;
    int oc = g * ocb + o, ci = ch * CK + c;
    float v = 0.f;
    if (oc < cout && ci < cin) {
      v = transposed ? w[((size_t)ci * cout + oc) * 9 + tap]
                     : w[((size_t)oc * cin + ci) * 9 + tap];
    }
    out[i] = v;
  }
}

template <int WM, int WN, int NT>
__global__ __launch_bounds__(WM* WN * 64) void conv3x3_mfma_kernel(Conv3x3Args a) {
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int OCB = WN * NT * 32;
  constexpr int PH = WM + 2;
  constexpr int SLOTS = PH * PW;                 // patch positions per channel
  constexpr int IN_FLOATS = CK * PH * RS;
  constexpr int W_FLOATS = 9 * CK * OCB;
  constexpr int W_VEC4 = W_FLOATS / 4;
  constexpr int W_PER_T = (W_VEC4 + NTHREADS - 1) / NTHREADS;
  constexpr int G = NTHREADS / SLOTS;            // channel groups staged in parallel
  static_assert(G >= 1, "tile too large for the staging scheme");
  constexpr int C_PER_T = (CK + G - 1) / G;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                        // [2][IN_FLOATS]
  float* s_w = smem + 2 * IN_FLOATS;         // [2][W_FLOATS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;

  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int ocg = b % a.nocg;
  const int n = b / a.nocg;
  const int x0 = tx * TW, y0 = ty * WM;

  // ---- staging assignment: thread -> (patch slot, channel group) ----------
  const int sg = tid / SLOTS;
  const int ss = tid - sg * SLOTS;
  const int sr = ss / PW, sc = ss - sr * PW;
  const int gy = y0 - 1 + sr, gx = x0 - 1 + sc;
  const bool s_active = (sg < G);
  const bool s_inimg = s_active && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
  const long long hw = (long long)a.h * a.w;
  const long long pix = (long long)gy * a.w + gx;
  const float* xb1 = a.x + (long long)n * a.x_ns + pix;
  const float* xb2 = a.x2 ? a.x2 + (long long)n * a.x2_ns + pix : nullptr;
  const int lds_slot = sr * RS + sc;

  const f32x4* wsrc = reinterpret_cast<const f32x4*>(
      a.wpk + (size_t)ocg * a.nchunk * W_FLOATS);

  float rin[C_PER_T];
  f32x4 rw[W_PER_T];

  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int i = 0; i < C_PER_T; ++i) {
      int cl = sg + i * G;
      int c = ch * CK + cl;
      float v = 0.f;
      if (s_inimg && cl < CK && c < a.cin) {
        v = (c < a.c1) ? xb1[(long long)c * hw] : xb2[(long long)(c - a.c1) * hw];
      }
      rin[i] = v;
    }
    const f32x4* ws = wsrc + (size_t)ch * W_VEC4;
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
      int idx = tid + i * NTHREADS;
      rw[i] = ws[idx < W_VEC4 ? idx : W_VEC4 - 1];
    }
  };
  auto store_chunk = [&](int buf) {
    float* si = s_in + buf * IN_FLOATS;
    if (s_active) {
#pragma unroll
      for (int i = 0; i < C_PER_T; ++i) {
        int cl = sg + i * G;
        if (cl < CK) si[cl * (PH * RS) + lds_slot] = rin[i];
      }
    }
    f32x4* sw = reinterpret_cast<f32x4*>(s_w + buf * W_FLOATS);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
      int idx = tid + i * NTHREADS;
      if (idx < W_VEC4) sw[idx] = rw[i];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int lh = lane >> 5, ll = lane & 31;
  // per-lane LDS offsets (floats)
  const int b_off = lh * (PH * RS) + wm * RS + ll;
  const int a_off = lh * OCB + wn * (NT * 32) + ll;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int ch = 0; ch < a.nchunk; ++ch) {
    const int buf = ch & 1;
    const bool more = (ch + 1 < a.nchunk);
    if (more) load_chunk(ch + 1);

    const float* si = s_in + buf * IN_FLOATS + b_off;
    const float* sw = s_w + buf * W_FLOATS + a_off;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int kk = 0; kk < CK / 2; ++kk) {
        float bv = si[(2 * kk) * (PH * RS) + ky * RS + kx];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float av = sw[(tap * CK + 2 * kk) * OCB + t * 32];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, activation, residual, NCHW store --------------------
  const int px = x0 + ll, py = y0 + wm;
  if (px < a.w && py < a.h) {
    const long long opix = (long long)py * a.w + px;
    float* yb = a.y + (long long)n * a.y_ns + opix;
    const float* rb = a.res ? a.res + (long long)n * a.res_ns + opix : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int oc = ocg * OCB + wn * (NT * 32) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (oc < a.cout) {
          float v = acc[t][r] + (a.bias ? a.bias[oc] : 0.f);
          v = apply_act(v, a.act);
          if (rb) v += rb[(long long)oc * hw];
          yb[(long long)oc * hw] = v;
        }
      }
    }
  }
}

template <int WM, int WN, int NT>
static int launch_conv(const Conv3x3Args& a0, int n, hipStream_t stream) {
  Conv3x3Args a = a0;
  constexpr int OCB = WN * NT * 32;
  a.tiles_x = cdiv(a.w, TW);
  a.tiles_y = cdiv(a.h, WM);
  a.nocg = cdiv(a.cout, OCB);
  a.nchunk = cdiv(a.cin, CK);
  size_t lds = 2 * (size_t)(CK * (WM + 2) * RS + 9 * CK * OCB) * sizeof(float);
  long long blocks = (long long)a.tiles_x * a.tiles_y * a.nocg * n;
  TG_REQUIRE(blocks > 0 && blocks < (1ll << 31), TG_E_SHAPE, "conv3x3: grid %lld", blocks);
  hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, NT>), dim3((unsigned)blocks),
                     dim3(WM * WN * 64), lds, stream, a);
  return check_launch("conv3x3_mfma");
}

}  // namespace tg

using namespace tg;

extern "C" int tg_conv3x3_pick_ocb(int cout) { return cout <= 32 ? 32 : 64; }

extern "C" size_t tg_conv3x3_packed_floats(int cin, int cout, int ocb) {
  if (cin <= 0 || cout <= 0 || (ocb != 32 && ocb != 64)) return 0;
  return (size_t)cdiv(cout, ocb) * cdiv(cin, CK) * 9 * CK * ocb;
}

extern "C" int tg_conv3x3_pack(const float* w, float* w_packed, int cin, int cout,
                               int ocb, int transposed, tg_stream_t stream) {
  TG_REQUIRE(w && w_packed, TG_E_ARG, "conv3x3_pack: null pointer");
  TG_REQUIRE(cin > 0 && cout > 0 && (ocb == 32 || ocb == 64), TG_E_SHAPE,
             "conv3x3_pack: cin=%d cout=%d ocb=%d", cin, cout, ocb);
  int nchunk = cdiv(cin, CK), nocg = cdiv(cout, ocb);
  int total = nocg * nchunk * 9 * CK * ocb;
  int blocks = cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack3x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     w_packed, cin, cout, ocb, nchunk, nocg, transposed);
  return check_launch("pack3x3");
}

extern "C" int tg_conv3x3_fwd(const float* x, int64_t x_nstride, int c1, const float* x2,
                              int64_t x2_nstride, const float* w_packed, int ocb,
                              const float* bias, const float* res, int64_t res_nstride,
                              float* y, int64_t y_nstride, int n, int cin, int cout, int h,
                              int w, int act, tg_stream_t stream) {
  TG_REQUIRE(x && w_packed && y, TG_E_ARG, "conv3x3_fwd: null pointer");
  TG_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, TG_E_SHAPE,
             "conv3x3_fwd: n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, w);
  TG_REQUIRE(c1 > 0 && c1 <= cin && (c1 == cin || x2), TG_E_ARG,
             "conv3x3_fwd: c1=%d cin=%d needs a second source", c1, cin);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_TANH24, TG_E_ARG, "conv3x3_fwd: act=%d", act);
  TG_REQUIRE(ocb == 32 || ocb == 64, TG_E_ARG, "conv3x3_fwd: ocb=%d", ocb);
  Conv3x3Args a{};
  a.x = x; a.x2 = (c1 < cin) ? x2 : nullptr; a.wpk = w_packed; a.bias = bias; a.res = res;
  a.y = y; a.x_ns = x_nstride; a.x2_ns = x2_nstride; a.res_ns = res_nstride; a.y_ns = y_nstride;
  a.c1 = c1; a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = act;
  hipStream_t s = (hipStream_t)stream;
  if (ocb == 32) return launch_conv<4, 1, 1>(a, n, s);
  // 64 output channels per workgroup.  Small images get the 2-row tile so that
  // more workgroups exist (tile quantisation dominates there).
  if (conv3x3_rows_per_wg(ocb, (long long)n * h * w) == 4) return launch_conv<4, 1, 2>(a, n, s);
  return launch_conv<2, 2, 1>(a, n, s);
}

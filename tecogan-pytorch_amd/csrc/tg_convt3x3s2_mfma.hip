// ConvTranspose2d(cin, cout, k3, stride 2, pad 1, output_padding 1) as four
// sub-pixel phase GEMMs on fp32 MFMA (gfx950).
//
//   out[2y+py][2x+px] = bias + sum_{ci} sum_{taps of phase (py,px)}
//                         in[ci][y+dy][x+dx] * W[ci][co][ky][kx]
//   with oy = 2*iy - 1 + ky  =>  py=0: (dy=0,ky=1);  py=1: (dy=0,ky=2),(dy=1,ky=0)
//   (same along x).  9 taps in total over the 4 phases -> exactly the MAC
//   count of the input-resolution convention (model_summary.py:47-48), no
//   multiplications by inserted zeros.
//
// One wave = one 32-pixel INPUT row segment x 32 output channels x 4 phases
// (4 accumulator tiles of 32x32; a workgroup is WM rows x 2 oc halves).  The four distinct B operands (dy,dx in
// {0,1}^2) are read once per K-step and shared by the taps that use them.
// The epilogue pairs the px=0/px=1 accumulators into float2 stores, so each
// half-wave writes 256 contiguous bytes of an output row.
//
// Replaces SRNet.conv_up (codes/models/networks/tecogan_nets.py:119-126).
#include "tg_common.h"

namespace tg {

constexpr int TTW = 32;
constexpr int TPW = TTW + 1;   // patch width: x .. x+32
constexpr int TRS = 36;        // LDS row stride
constexpr int TOCB = 64;

struct ConvTArgs {
  const float* x;
  const float* wpk;
  const float* bias;
  float* y;
  long long x_ns, y_ns;
  int cin, cout, h, w, act;
  int tiles_x, tiles_y, nchunk, nocg;
};

template <int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void convt3x3s2_mfma_kernel(ConvTArgs a) {
  static_assert(WN * 32 == TOCB, "WN waves x 32 oc must cover the 64-oc block");
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int PH = WM + 1;
  constexpr int SLOTS = PH * TPW;
  constexpr int IN_FLOATS = CK * PH * TRS;
  constexpr int W_FLOATS = 9 * CK * TOCB;
  constexpr int W_VEC4 = W_FLOATS / 4;
  constexpr int W_PER_T = (W_VEC4 + NTHREADS - 1) / NTHREADS;
  constexpr int G = NTHREADS / SLOTS;
  static_assert(G >= 1, "tile too large");
  constexpr int C_PER_T = (CK + G - 1) / G;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + 2 * IN_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int ocg = b % a.nocg;
  const int n = b / a.nocg;
  const int x0 = tx * TTW, y0 = ty * WM;

  const int sg = tid / SLOTS;
  const int ss = tid - sg * SLOTS;
  const int sr = ss / TPW, sc = ss - sr * TPW;
  const int gy = y0 + sr, gx = x0 + sc;
  const bool s_active = sg < G;
  const bool s_inimg = s_active && gy < a.h && gx < a.w;
  const long long hw = (long long)a.h * a.w;
  const float* xb = a.x + (long long)n * a.x_ns + (long long)gy * a.w + gx;
  const int lds_slot = sr * TRS + sc;
  const f32x4* wsrc =
      reinterpret_cast<const f32x4*>(a.wpk + (size_t)ocg * a.nchunk * W_FLOATS);

  float rin[C_PER_T];
  f32x4 rw[W_PER_T];
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int i = 0; i < C_PER_T; ++i) {
      int cl = sg + i * G, c = ch * CK + cl;
      rin[i] = (s_inimg && cl < CK && c < a.cin) ? xb[(long long)c * hw] : 0.f;
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
        if (cl < CK) si[cl * (PH * TRS) + lds_slot] = rin[i];
      }
    }
    f32x4* sw = reinterpret_cast<f32x4*>(s_w + buf * W_FLOATS);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
      int idx = tid + i * NTHREADS;
      if (idx < W_VEC4) sw[idx] = rw[i];
    }
  };

  // acc[phase = py*2+px]: 32 oc x 32 input pixels each
  f32x16 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  const int lh = lane >> 5, ll = lane & 31;
  const int b_off = lh * (PH * TRS) + wm * TRS + ll;
  const int a_off = lh * TOCB + wn * 32 + ll;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

#define TG_CT_MFMA(P, TAP, BV)                                                   \
  {                                                                              \
    float av = sw[((TAP)*CK + 2 * kk) * TOCB];                                   \
    acc[P] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, BV, acc[P], 0, 0, 0);      \
  }

  for (int ch = 0; ch < a.nchunk; ++ch) {
    const int buf = ch & 1;
    const bool more = ch + 1 < a.nchunk;
    if (more) load_chunk(ch + 1);
    const float* si = s_in + buf * IN_FLOATS + b_off;
    const float* sw = s_w + buf * W_FLOATS + a_off;
#pragma unroll
    for (int kk = 0; kk < CK / 2; ++kk) {
      const float* sc_ = si + (2 * kk) * (PH * TRS);
      float b00 = sc_[0], b01 = sc_[1], b10 = sc_[TRS], b11 = sc_[TRS + 1];
      // tap index = ky*3 + kx
      TG_CT_MFMA(0, 4, b00)   // (py0,px0): ky1,kx1 in[y][x]
      TG_CT_MFMA(1, 3, b01)   // (py0,px1): ky1,kx0 in[y][x+1]
      TG_CT_MFMA(1, 5, b00)   //            ky1,kx2 in[y][x]
      TG_CT_MFMA(2, 1, b10)   // (py1,px0): ky0,kx1 in[y+1][x]
      TG_CT_MFMA(2, 7, b00)   //            ky2,kx1 in[y][x]
      TG_CT_MFMA(3, 0, b11)   // (py1,px1): ky0,kx0 in[y+1][x+1]
      TG_CT_MFMA(3, 2, b10)   //            ky0,kx2 in[y+1][x]
      TG_CT_MFMA(3, 6, b01)   //            ky2,kx0 in[y][x+1]
      TG_CT_MFMA(3, 8, b00)   //            ky2,kx2 in[y][x]
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }
#undef TG_CT_MFMA

  const int px = x0 + ll, py = y0 + wm;
  if (px < a.w && py < a.h) {
    const int ow = 2 * a.w;
    const long long ohw = 4ll * hw;
    float* yb = a.y + (long long)n * a.y_ns + (long long)(2 * py) * ow + 2 * px;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int oc = ocg * TOCB + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (oc < a.cout) {
        float bv = a.bias ? a.bias[oc] : 0.f;
        float* yo = yb + (long long)oc * ohw;
        float2 v0, v1;
        v0.x = apply_act(acc[0][r] + bv, a.act);
        v0.y = apply_act(acc[1][r] + bv, a.act);
        v1.x = apply_act(acc[2][r] + bv, a.act);
        v1.y = apply_act(acc[3][r] + bv, a.act);
        *reinterpret_cast<float2*>(yo) = v0;
        *reinterpret_cast<float2*>(yo + ow) = v1;
      }
    }
  }
}

}  // namespace tg

using namespace tg;

extern "C" int tg_convt3x3s2_fwd(const float* x, int64_t x_nstride, const float* w_packed,
                                 const float* bias, float* y, int64_t y_nstride, int n,
                                 int cin, int cout, int h, int w, int act,
                                 tg_stream_t stream) {
  TG_REQUIRE(x && w_packed && y, TG_E_ARG, "convt3x3s2_fwd: null pointer");
  TG_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, TG_E_SHAPE,
             "convt3x3s2_fwd: n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, w);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_TANH24, TG_E_ARG, "convt: act=%d", act);
  TG_REQUIRE((y_nstride % 2) == 0 && ((uintptr_t)y % 8) == 0, TG_E_ARG,
             "convt3x3s2_fwd: output must be 8-byte aligned");
  ConvTArgs a{};
  a.x = x; a.wpk = w_packed; a.bias = bias; a.y = y; a.x_ns = x_nstride; a.y_ns = y_nstride;
  a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = act;
  constexpr int WM = 4, WN = 2;
  a.tiles_x = cdiv(w, TTW);
  a.tiles_y = cdiv(h, WM);
  a.nocg = cdiv(cout, TOCB);
  a.nchunk = cdiv(cin, CK);
  size_t lds = 2 * (size_t)(CK * (WM + 1) * TRS + 9 * CK * TOCB) * sizeof(float);
  long long blocks = (long long)a.tiles_x * a.tiles_y * a.nocg * n;
  TG_REQUIRE(blocks > 0 && blocks < (1ll << 31), TG_E_SHAPE, "convt: grid %lld", blocks);
  hipLaunchKernelGGL((convt3x3s2_mfma_kernel<WM, WN>), dim3((unsigned)blocks), dim3(WM * WN * 64), lds,
                     (hipStream_t)stream, a);
  return check_launch("convt3x3s2_mfma");
}

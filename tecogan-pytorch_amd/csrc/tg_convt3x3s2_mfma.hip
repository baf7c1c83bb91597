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
// (4 accumulator tiles of 32x32; a workgroup is WM rows x 2 oc halves).
// Same operand layouts as tg_conv3x3_mfma.hip (K permuted so a tap's four
// k-steps are one ds_read_b128 per operand); the four distinct B operands
// (dy,dx in {0,1}^2) are read once per chunk-half and shared by the taps that
// use them.  The epilogue pairs the px=0/px=1 accumulators into float2
// stores, so each half-wave writes 256 contiguous bytes of an output row.
//
// Replaces SRNet.conv_up (codes/models/networks/tecogan_nets.py:119-126).
#include "tg_common.h"
#include <cstdlib>
#include <type_traits>

namespace tg {

constexpr int TTW = 32;
constexpr int TPW = TTW + 1;   // patch width: x .. x+32
constexpr int TRS = 34;        // LDS row stride (16-byte slots)
constexpr int TOCB = 64;
constexpr unsigned TOOB = 0x80000000u;

struct ConvTArgs {
  const float* x;
  const float* wpk;
  const float* bias;
  float* y;
  long long x_ns, y_ns;
  int cin, cout, h, w, act;
  int tiles_x, tiles_y, nchunk, nocg;
  int y_base;          // first input row of this launch's tile grid (0 but for the split Z launch, round 6)
  // Z mode (the LAST up-sampling layer of SRNet, inference): instead of the 64-channel HR
  // tensor the kernel emits the 9*cz "tap planes" of the following 3x3 output conv,
  //   z[tap*cz + o][Y][X] = sum_oc  Wout[o][oc][tap] * act(convT(x)[oc][Y][X] + bias[oc]),
  // i.e. the output conv's channel contraction done where the operand already sits in registers.
  const float* wz;     // [2 oc-halves][16 steps][64 lanes] A operand of the 1x1 contraction
  float* z;            // (n, 32, 2h, 2w) planes; the first zrows are written
  long long z_ns;
  int zrows;           // 9 * cz <= 32
};

template <int WM, int WN, bool ZMODE = false>
__global__ __launch_bounds__(WM* WN * 64, ZMODE ? 4 : 1) void convt3x3s2_mfma_kernel(ConvTArgs a) {      // (4 waves per SIMD: without the second argument the 2-row form takes 118 + 64 AGPRs and runs 2)
  static_assert(WN * 32 == TOCB, "WN waves x 32 oc must cover the 64-oc block");
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int PH = WM + 1;
  constexpr int IN_ITEMS = PH * 2 * TPW;
  constexpr int IN_FLOATS = PH * 2 * TRS * 4;
  constexpr int W_FLOATS = 9 * CK * TOCB;
  constexpr int W_VEC4 = W_FLOATS / 4;
  constexpr int W_PER_T = (W_VEC4 + NTHREADS - 1) / NTHREADS;
  constexpr int I_PER_T = (IN_ITEMS + NTHREADS - 1) / NTHREADS;

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
  const int x0 = tx * TTW, y0 = a.y_base + ty * WM;
  const int hw = a.h * a.w;

  unsigned voff[I_PER_T];
  int lds_item[I_PER_T];
#pragma unroll
  for (int i = 0; i < I_PER_T; ++i) {
    int q = tid + i * NTHREADS;
    int r = q / (2 * TPW), rem = q - r * (2 * TPW);
    int hf = rem / TPW, col = rem - hf * TPW;
    int gy = y0 + r, gx = x0 + col;
    bool ok = q < IN_ITEMS && gy < a.h && gx < a.w;
    voff[i] = ok ? (unsigned)((4 * hf * hw + gy * a.w + gx) * 4) : TOOB;
    lds_item[i] = q < IN_ITEMS ? ((r * 2 + hf) * TRS + col) * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.cin * hw * 4, 0x00020000);
  const unsigned plane = (unsigned)hw * 4u;
  const f32x4* wsrc =
      reinterpret_cast<const f32x4*>(a.wpk + (size_t)ocg * a.nchunk * W_FLOATS);

  f32x4 rin[I_PER_T];
  f32x4 rw[W_PER_T];
  auto load_chunk = [&](int ch) {
    const unsigned cbase = (unsigned)(ch * CK) * plane;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        rin[i][j] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(
                       rs1, (int)(voff[i] + cbase + (unsigned)j * plane), 0, 0));
    const f32x4* ws = wsrc + (size_t)ch * W_VEC4;
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
      int idx = tid + i * NTHREADS;
      rw[i] = ws[idx < W_VEC4 ? idx : W_VEC4 - 1];
    }
  };
  auto store_chunk = [&](int buf) {
    float* si = s_in + buf * IN_FLOATS;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i)
      if (lds_item[i] >= 0) *reinterpret_cast<f32x4*>(si + lds_item[i]) = rin[i];
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
  const int b_off = ((wm * 2 + lh) * TRS + ll) * 4;
  const int a_off = (lh * TOCB + wn * 32 + ll) * 4;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

#define TG_CT_TAP(P, TAP, BV)                                                           \
  {                                                                                     \
    f32x4 av = *reinterpret_cast<const f32x4*>(sw + (TAP) * (2 * TOCB * 4));            \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) acc[P] =                           \
        __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], BV[kk], acc[P], 0, 0, 0);          \
  }

  for (int ch = 0; ch < a.nchunk; ++ch) {
    const int buf = ch & 1;
    const bool more = ch + 1 < a.nchunk;
    if (more) load_chunk(ch + 1);
    const float* si = s_in + buf * IN_FLOATS + b_off;
    const float* sw = s_w + buf * W_FLOATS + a_off;
    const f32x4 b00 = *reinterpret_cast<const f32x4*>(si);
    const f32x4 b01 = *reinterpret_cast<const f32x4*>(si + 4);
    const f32x4 b10 = *reinterpret_cast<const f32x4*>(si + 2 * TRS * 4);
    const f32x4 b11 = *reinterpret_cast<const f32x4*>(si + 2 * TRS * 4 + 4);
    // tap index = ky*3 + kx
    TG_CT_TAP(0, 4, b00)   // (py0,px0): ky1,kx1 in[y][x]
    TG_CT_TAP(1, 3, b01)   // (py0,px1): ky1,kx0 in[y][x+1]
    TG_CT_TAP(1, 5, b00)   //            ky1,kx2 in[y][x]
    TG_CT_TAP(2, 1, b10)   // (py1,px0): ky0,kx1 in[y+1][x]
    TG_CT_TAP(2, 7, b00)   //            ky2,kx1 in[y][x]
    TG_CT_TAP(3, 0, b11)   // (py1,px1): ky0,kx0 in[y+1][x+1]
    TG_CT_TAP(3, 2, b10)   //            ky0,kx2 in[y+1][x]
    TG_CT_TAP(3, 6, b01)   //            ky2,kx0 in[y][x+1]
    TG_CT_TAP(3, 8, b00)   //            ky2,kx2 in[y][x]
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }
#undef TG_CT_TAP

  const int px = x0 + ll, py = y0 + wm;
  const float slope = act_slope(a.act);
  const int ocb0 = ocg * TOCB + wn * 32 + 4 * lh;
  float bv[16];
  if constexpr (ZMODE) {
    // (a channel past cout reads 0 here where the other branch reads bias[cout - 1]: its accumulator is 0 and its row of
    // the contraction operand is 0, the contribution is 0 either way)
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias : a.wz), 0, a.bias ? a.cout * 4 : 0, 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (int)((unsigned)ocb0 * 4u + (unsigned)((r & 3) + 8 * (r >> 2)) * 4u), 0, 0));
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int oc = ocb0 + (r & 3) + 8 * (r >> 2);
      bv[r] = a.bias ? a.bias[oc < a.cout ? oc : a.cout - 1] : 0.f;
    }
  }
  if constexpr (ZMODE) {
    // The accumulator layout IS the B operand of the contraction over oc: register r of lane l
    // holds channel wn*32 + (r&3) + 8*(r>>2) + 4*(l>>5) of pixel column l&31, i.e. MFMA step r
    // consumes the channel pair (c_r, c_r + 4).  The A operand (32 tap-plane rows x that pair)
    // was packed in exactly this order (tg_convt_pack_wz).  16 MFMAs per phase and wave; the two
    // oc-half waves of a row are summed through LDS (staging buffers are dead: the K loop ended
    // on a barrier), the wn == 0 wave stores the planes.
    // Round 6: this epilogue was ~750 VALU instructions per wave and tile against 352 MFMAs -- and fp32 MFMA and
    // VALU time ADD on a gfx950 SIMD (EXPERIMENTS.md).  Now: bias + activation in place (ReLU as one v_max), operands
    // through buffer resources with immediate offsets (no 64-bit address arithmetic), the reduction buffer addressed
    // by one base + immediates, 8-byte buffer stores with the plane as the scalar offset and an out-of-range offset
    // instead of a branch for rows >= zrows.  Same arithmetic in the same order: bit-identical to the round-2 form.
    const __amdgpu_buffer_rsrc_t rwz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wz), 0, 2 * 16 * 64 * 4, 0x00020000);
    const unsigned wzo = (unsigned)(wn * 16 * 64 + lane) * 4u;
    float az[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      az[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rwz, (int)(wzo + (unsigned)r * 256u), 0, 0));
    const bool relu = slope == 0.f;              // launch-uniform
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = acc[p][r] + bv[r];
        acc[p][r] = relu ? fmaxf(t, 0.f) : (t >= 0.f ? t : t * slope + 0.f);
      }
    float* const red = smem + (wm * 16) * 64 + lane;      // [2 phases][WM][16][64]: + (q * WM * 16 + r) * 64
    const bool inimg = px < a.w && py < a.h;
    const int ow = 2 * a.w;
    const unsigned ohw = 4u * (unsigned)hw;
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
        a.z + (long long)n * a.z_ns, 0, (int)(32u * ohw * 4u), 0x00020000);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {            // phase pairs (py = pp; px = 0, 1)
      f32x16 z[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[q][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          z[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(az[r], acc[pp * 2 + q][r], z[q], 0, 0, 0);
      }
      if (wn == 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(q * WM * 16 + r) * 64] = z[q][r];
      }
      __syncthreads();
      if (wn == 0) {
        const unsigned zo = inimg ? ((unsigned)(4 * lh) * ohw + (unsigned)((2 * py + pp) * ow + 2 * px)) * 4u : TOOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mb = (r & 3) + 8 * (r >> 2);
          if (mb >= 27) continue;                // (step 15 holds rows 27 and 31: zrows <= 27, never stored)
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          // (locals first: __builtin_bit_cast of a vector ELEMENT expression reads element 0 whatever the index)
          float e0 = z[0][r] + red[(0 * WM * 16 + r) * 64];
          asm volatile("" : "+v"(e0));           // (keeps the SLP vectorizer from pairing the two adds: v_pk_add_f32 + 3 v_mov)
          const float e1 = z[1][r] + red[(1 * WM * 16 + r) * 64];
          const u32x2 v = {__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
          if (mb + 4 < a.zrows) {                // uniform: both lane halves' rows exist (12 of the 15 steps at zrows = 27)
            __builtin_amdgcn_raw_buffer_store_b64(v, rz, (int)zo, (int)((unsigned)mb * ohw * 4u), 0);
          } else {
            const unsigned off = (mb + 4 * lh < a.zrows) ? zo : TOOB;
            __builtin_amdgcn_raw_buffer_store_b64(v, rz, (int)off, (int)((unsigned)mb * ohw * 4u), 0);
          }
        }
      }
      if (pp == 0) __syncthreads();             // the second pair re-uses the reduction buffer
    }
    return;
  }
  if (px < a.w && py < a.h) {
    const int ow = 2 * a.w;
    const long long ohw = 4ll * hw;
    float* yb = a.y + (long long)n * a.y_ns + (long long)(2 * py) * ow + 2 * px;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int oc = ocb0 + (r & 3) + 8 * (r >> 2);
      float* yo = yb + (long long)oc * ohw;
      float2 v0, v1;
      float t0 = acc[0][r] + bv[r], t1 = acc[1][r] + bv[r];
      float t2 = acc[2][r] + bv[r], t3 = acc[3][r] + bv[r];
      v0.x = t0 >= 0.f ? t0 : t0 * slope + 0.f;
      v0.y = t1 >= 0.f ? t1 : t1 * slope + 0.f;
      v1.x = t2 >= 0.f ? t2 : t2 * slope + 0.f;
      v1.y = t3 >= 0.f ? t3 : t3 * slope + 0.f;
      if (oc < a.cout) {
        *reinterpret_cast<float2*>(yo) = v0;
        *reinterpret_cast<float2*>(yo + ow) = v1;
      }
    }
  }
}

// ---- small frames (the training unroll: 2 x 32 x 32 .. 2 x 64 x 64 inputs per launch) ----------------
// The kernel above gives 32-128 workgroups there, each wave walking the 8 channel chunks of its row one
// after the other (288 dependent MFMAs = 8.8 us, 15-16 us per launch with 12 % of the SIMDs busy).  The
// one-shot scheme of tg_conv3x3_mfma.hip instead: a workgroup = ONE input row x 32 pixels x 64 channels
// x 4 phases, all 8 chunks of the 2-row patch staged at once (12 loads per thread, all in flight), the 8
// waves = 2 oc halves x 4 K groups of two chunks (72 MFMAs each, weights straight from L2 into
// registers), the K groups meet in LDS in a fixed order and every wave finishes one (output row, half
// of the accumulator registers) pair: bias, activation, 8-byte stores.
constexpr int TOS_CH_FLOATS = 2 * 2 * TRS * 4;        // one chunk of the patch: 2 rows x 2 halves x 34 slots x 4
__global__ __launch_bounds__(512) void convt3x3s2_oneshot_kernel(ConvTArgs a) {
  constexpr int KG = 4, W_FLOATS = 9 * CK * TOCB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                                 // [8 chunks][TOS_CH_FLOATS]
  float* red = smem + 8 * TOS_CH_FLOATS;              // [2 oc halves][4 K groups][4 phases][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int lh = lane >> 5, ll = lane & 31;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * TTW, y0 = ty;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;
  const int nchunk = a.nchunk;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.cin * hw * 4, 0x00020000);
  // ---- this wave's weights: chunks 2 wk, 2 wk + 1
  const f32x4* wl = reinterpret_cast<const f32x4*>(a.wpk) + (lh * TOCB + wn * 32 + ll);
  f32x4 aw[2][9];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int c = wk * 2 + ci;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < nchunk) v = wl[(size_t)c * (W_FLOATS / 4) + tap * (2 * TOCB)];
      aw[ci][tap] = v;
    }
  }
  // ---- the patch: item q = (chunk, row r, half hf, col) -> 4 channel planes, all loads before the first store
  constexpr int ITEMS_PER_CH = 2 * 2 * TPW;            // 132
  constexpr int MAXQ = (8 * ITEMS_PER_CH + 511) / 512;  // 3
  const int total = nchunk * ITEMS_PER_CH;
  f32x4 v[MAXQ];
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const int q = tid + k * 512;
    const int ch = q / ITEMS_PER_CH, rem = q - ch * ITEMS_PER_CH;
    const int r = rem / (2 * TPW), rem2 = rem - r * (2 * TPW);
    const int hf = rem2 / TPW, col = rem2 - hf * TPW;
    const int gy = y0 + r, gx = x0 + col;
    const bool ok = q < total && gy < a.h && gx < a.w;
    const unsigned base = ok ? (unsigned)(((ch * CK + 4 * hf) * hw + gy * a.w + gx) * 4) : TOOB;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, (int)(base + (unsigned)j * plane), 0, 0));
  }
#pragma unroll
  for (int k = 0; k < MAXQ; ++k) {
    const int q = tid + k * 512;
    const int ch = q / ITEMS_PER_CH, rem = q - ch * ITEMS_PER_CH;
    const int r = rem / (2 * TPW), rem2 = rem - r * (2 * TPW);
    const int hf = rem2 / TPW, col = rem2 - hf * TPW;
    if (q < total) *reinterpret_cast<f32x4*>(s_in + ch * TOS_CH_FLOATS + ((r * 2 + hf) * TRS + col) * 4) = v[k];
  }
  __syncthreads();
  // ---- MFMAs: acc[phase = py * 2 + px], 32 oc x 32 input pixels each
  f32x16 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
#define TG_CTO_TAP(P, TAP, BV)                                                             \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) acc[P] =                                \
      __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ci][TAP][kk], BV[kk], acc[P], 0, 0, 0);
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int c = wk * 2 + ci;
    if (c < nchunk) {
      const float* si = s_in + c * TOS_CH_FLOATS + (lh * TRS + ll) * 4;
      const f32x4 b00 = *reinterpret_cast<const f32x4*>(si);
      const f32x4 b01 = *reinterpret_cast<const f32x4*>(si + 4);
      const f32x4 b10 = *reinterpret_cast<const f32x4*>(si + 2 * TRS * 4);
      const f32x4 b11 = *reinterpret_cast<const f32x4*>(si + 2 * TRS * 4 + 4);
      TG_CTO_TAP(0, 4, b00)   // (py0,px0): ky1,kx1 in[y][x]
      TG_CTO_TAP(1, 3, b01)   // (py0,px1): ky1,kx0 in[y][x+1]
      TG_CTO_TAP(1, 5, b00)   //            ky1,kx2 in[y][x]
      TG_CTO_TAP(2, 1, b10)   // (py1,px0): ky0,kx1 in[y+1][x]
      TG_CTO_TAP(2, 7, b00)   //            ky2,kx1 in[y][x]
      TG_CTO_TAP(3, 0, b11)   // (py1,px1): ky0,kx0 in[y+1][x+1]
      TG_CTO_TAP(3, 2, b10)   //            ky0,kx2 in[y+1][x]
      TG_CTO_TAP(3, 6, b01)   //            ky2,kx0 in[y][x+1]
      TG_CTO_TAP(3, 8, b00)   //            ky2,kx2 in[y][x]
    }
  }
#undef TG_CTO_TAP
  // ---- the K groups meet in LDS; wave (wn, wk) finishes output row py = wk & 1, registers [8 (wk >> 1), +8)
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(((wn * KG + wk) * 4 + p) * 16 + r) * 64 + lane] = acc[p][r];
  __syncthreads();
  const int fpy = wk & 1, r0 = 8 * (wk >> 1);
  const int px = x0 + ll, py = y0;
  if (px < a.w && py < a.h) {
    const float slope = act_slope(a.act);
    const int ow = 2 * a.w;
    const long long ohw = 4ll * hw;
    float* yb = a.y + (long long)n * a.y_ns + (long long)(2 * py + fpy) * ow + 2 * px;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = r0 + rr;
      const int oc = wn * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int g = 0; g < KG; ++g) {                   // fixed order
        s0 += red[(((wn * KG + g) * 4 + fpy * 2 + 0) * 16 + r) * 64 + lane];
        s1 += red[(((wn * KG + g) * 4 + fpy * 2 + 1) * 16 + r) * 64 + lane];
      }
      if (oc < a.cout) {
        const float bb = a.bias ? a.bias[oc] : 0.f;
        float2 o;
        s0 += bb; s1 += bb;
        o.x = s0 >= 0.f ? s0 : s0 * slope + 0.f;
        o.y = s1 >= 0.f ? s1 : s1 * slope + 0.f;
        *reinterpret_cast<float2*>(yb + (long long)oc * ohw) = o;
      }
    }
  }
}


// ---- Z mode on a large frame (inference's last up-sampling layer, 268x640 -> 536x1280): STREAMING form (round 6) ----
// The tiled kernel above re-stages all 147 KB of weights through LDS for every 4-row x 32-pixel tile (1 340 times at
// 268x640), meets at a barrier per 8-channel chunk, reduces the two output-channel halves through LDS behind two more
// barriers, and quantises to 5.23 -> 6 tiles per CU: 149 us = 0.64 of the fp32-MFMA peak (22 % of the frame).
// Here the 64x64x9 weights are LDS-RESIDENT for the whole launch (147 456 B of the 160 KB, copied once per CU, then
// read-only: no barrier after the first) and every WAVE is an autonomous worker:
//   * a work item = one input row x 32 pixels x ALL 64 output channels x one output-row parity py (the phase pair
//     px = 0, 1): 4 accumulator tiles of 32x32 as before, but the two channel halves sit in the SAME wave, so the
//     output conv's contraction (Z mode) needs no cross-wave reduction;
//   * the B operand (input pixels) goes global -> registers (lanes 0-31 read 128 contiguous bytes of one channel,
//     lanes 32-63 of the channel 4 above: the K permutation of the packed weights), prefetched one 8-channel chunk
//     ahead; the A operand is one ds_read_b128 per (tap, channel half) of the resident weights, as in the tiled kernel;
//   * items are pulled from an atomic counter, the 6-tap items (py = 1) first, the 3-tap items (py = 0) last: the
//     tail of the launch is one small item, and a workgroup that starts late (LDS held by a co-running kernel of the
//     flow stream) simply finds less work -- 10 720 items on 3 072 workers at 268x640;
//   * per phase the same taps in the same order over the same ascending chunks as the tiled kernel, the two halves'
//     contraction sums added in its order: results are BIT-IDENTICAL to convt3x3s2_mfma_kernel<.,.,true>.
// The counter pair {next item, finished waves} lives in a device-global slot the host rotates per launch; the last wave
// to finish puts the slot back to zero.
#ifndef ZS_ABL
#define ZS_ABL 0     // lab builds (TG_LAB), timing only: 1 no B loads after an item's first chunk, 2 no contraction / stores, 4 no A reads after the first tap
#endif
#if !TG_LAB && ZS_ABL
#error "tg_convt3x3s2_mfma.hip: ZS_ABL needs -DTG_LAB=1 (lab builds only; the ablated kernel computes wrong results)"
#endif
constexpr int ZS_WAVES = 12;
constexpr int ZS_THREADS = ZS_WAVES * 64;
constexpr int ZS_W_FLOATS = 9 * 64 * 64;                  // the packed weights, verbatim: [chunk 8][tap 9][half 2][oc 64][4]
constexpr int ZS_WZ_OFF = ZS_W_FLOATS;                    // [oc half 2][step 16][lane 64]
constexpr int ZS_BIAS_OFF = ZS_WZ_OFF + 2 * 16 * 64;
constexpr size_t ZS_LDS_BYTES = (size_t)(ZS_BIAS_OFF + 64) * sizeof(float);    // 155 904 of 163 840 (+ 2.2 KB static: the batch ring)
constexpr int ZS_RING = 256;                              // batches a wave may lag behind the newest one
constexpr int ZS_SLOTS = 64;
__device__ unsigned g_zs_slot[ZS_SLOTS][2];

template <int PY>
__device__ __forceinline__ void zs_item(const ConvTArgs& a, const float* s_w, int n, int y, int x0, int lane) {
  constexpr int NB = PY ? 4 : 2;                           // B vectors per chunk: b00 b01 (b10 b11)
  const int lh = lane >> 5, ll = lane & 31;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.cin * hw * 4, 0x00020000);
  // per-lane byte offsets of the window rows inside channel 4 lh of chunk 0 (out of the image: a zero).  Pixels x and
  // x + 1 of a row come as ONE 8-byte load (b00 | b01, b10 | b11): half the load instructions of the first form; the
  // x + 1 value of the image's last column would be the next row's first pixel and is zeroed by hand.
  constexpr int NR = NB / 2;
  unsigned vo[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int gy = y + i, gx = x0 + ll;
    vo[i] = (gy < a.h && gx < a.w) ? (unsigned)(4 * lh) * plane + (unsigned)(gy * a.w + gx) * 4u : TOOB;
  }
  const bool edge = x0 + ll + 1 >= a.w;                    // this lane's x + 1 is outside the row
  const bool edge_tile = x0 + TTW >= a.w;                  // uniform: only the last tile of a row has such lanes
  f32x4 bq[2][NB];                                        // two chunks of the B operand in flight (see the K loop)
  auto load_b = [&](int ch, f32x4 (&b)[NB]) {
    const unsigned cb = (unsigned)(ch * CK) * plane;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(vo[i] + (unsigned)kk * plane), (int)cb, 0);
        const unsigned v0 = v[0], v1 = v[1];
        b[2 * i][kk] = __builtin_bit_cast(float, v0);
        b[2 * i + 1][kk] = __builtin_bit_cast(float, v1);
      }
  };
  auto fix_edge = [&](f32x4 (&b)[NB]) {
    if (edge_tile) {
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b[2 * i + 1][kk] = edge ? 0.f : b[2 * i + 1][kk];
    }
  };
  f32x16 acc[4];                                           // [phase px 0 | 1][oc half 0 | 1]
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  const float* sa = s_w + (lh * TOCB + ll) * 4;
  // The A operand of tap t + 1 is requested before the MFMAs of tap t (two register sets): the compiler left to itself
  // hoists all 12 reads of a chunk (48 registers) and spills.
  auto lda = [&](const float* sw, int tap, f32x4 (&av)[2]) {
    av[0] = *reinterpret_cast<const f32x4*>(sw + tap * (2 * TOCB * 4));
    av[1] = *reinterpret_cast<const f32x4*>(sw + tap * (2 * TOCB * 4) + 32 * 4);
  };
  auto mm = [&](int q, const f32x4 (&av)[2], const f32x4& bv) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc[q * 2 + hf] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[hf][kk], bv[kk], acc[q * 2 + hf], 0, 0, 0);
  };
  f32x4 ab[2][2];                                          // [register set][oc half]
  // (q, tap, B vector) in the tiled kernel's order per phase
  constexpr int NT = PY ? 6 : 3;
  constexpr int TQ[6] = {0, PY ? 0 : 1, 1, 1, 1, 1};
  constexpr int TT[6] = {PY ? 1 : 4, PY ? 7 : 3, PY ? 0 : 5, 2, 6, 8};
  constexpr int TB[6] = {PY ? 2 : 0, PY ? 0 : 1, PY ? 3 : 0, 2, 1, 0};
  // PAR: the register set that holds this chunk's first tap on entry
  auto chunk = [&](int ch, const f32x4 (&b)[NB], bool last, auto par) {
    constexpr int PAR = decltype(par)::value;
    const float* sw = sa + ch * (9 * CK * TOCB);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (ZS_ABL & 4) { if (ch == 0 && t == 0) lda(sw, TT[1], ab[1]); }
      else if (t + 1 < NT) lda(sw, TT[t + 1], ab[(PAR + t + 1) & 1]);
      else if (!last) lda(sw + 9 * CK * TOCB, TT[0], ab[(PAR + t + 1) & 1]);      // the next chunk's first tap
      mm(TQ[t], ab[(PAR + t) & 1], b[TB[t]]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // The K loop, fully unrolled over the (at most 8) chunks with compile-time register sets; the B operand is requested
  // one chunk ahead (two chunks ahead -- three register sets -- measured +-0 and spilled: tools/abl_convtz.sh, round 6).
  load_b(0, bq[0]);
  if ((ZS_ABL & 1)) load_b(1, bq[1]);
  lda(sa, TT[0], ab[0]);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < a.nchunk) {
      if (c + 1 < a.nchunk && !(ZS_ABL & 1)) load_b(c + 1, bq[(c + 1) % 2]);
      fix_edge(bq[c % 2]);
      if ((c * NT) & 1) chunk(c, bq[c % 2], c + 1 >= a.nchunk, std::integral_constant<int, 1>{});
      else chunk(c, bq[c % 2], c + 1 >= a.nchunk, std::integral_constant<int, 0>{});
    }
  }
  // ---- bias + activation, the output conv's contraction over the 64 channels, 8-byte stores of the tap planes ----
  const float slope = act_slope(a.act);
  const float* s_wz = s_w + ZS_WZ_OFF + lane;
  const float* s_b = s_w + ZS_BIAS_OFF + 4 * lh;
  const int px = x0 + ll;
  const bool inimg = px < a.w;
  const int ow = 2 * a.w;
  const int ohw = 4 * hw;
  // One contraction chain per (phase, channel half) -- the halves added in the tiled kernel's order, wn 0 + wn 1 -- with
  // the two chains of a phase interleaved (a chain's MFMAs depend on each other) and the operands of four MFMAs
  // prepared ahead of them (VALU -> MFMA hazard distance).  Round-6 anatomy (tools/abl_convtz.sh): this epilogue cost
  // 29 us of the launch's 145 for 18 us of MFMA work in its first form (one chain at a time, 4 VALU per element,
  // sixteen divergent branches around the stores).
  const bool relu = slope == 0.f;                          // launch-uniform
  auto actv = [&](float t) { return relu ? fmaxf(t, 0.f) : (t >= 0.f ? t : t * slope + 0.f); };
  // bias + activation in place (no new registers), then the chains straight out of the accumulators
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = actv(acc[p][r] + s_b[(p & 1) * 32 + (r & 3) + 8 * (r >> 2)]);
  __builtin_amdgcn_sched_barrier(0);
  f32x16 zq[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    f32x16 za, zb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { za[r] = 0.f; zb[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      za = __builtin_amdgcn_mfma_f32_32x32x2f32(s_wz[r * 64], acc[q * 2][r], za, 0, 0, 0);
      zb = __builtin_amdgcn_mfma_f32_32x32x2f32(s_wz[(16 + r) * 64], acc[q * 2 + 1][r], zb, 0, 0, 0);
    }
    zq[q] = za + zb;
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    // 32-bit buffer offsets (the planes of one image are 32 x 4 hw floats < 4 GiB: checked by the launcher): the lane part
    // once, the plane of step r as the instruction's scalar offset -- 64-bit addresses per plane cost 32 registers.  Rows
    // past zrows and pixels past the image edge get an offset beyond the descriptor's range: the store is dropped.
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
        a.z + (long long)n * a.z_ns, 0, (int)(32u * (unsigned)ohw * 4u), 0x00020000);
    const unsigned zo = inimg ? ((unsigned)(4 * lh) * (unsigned)ohw + (unsigned)((2 * y + PY) * ow + 2 * px)) * 4u : TOOB;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mb = (r & 3) + 8 * (r >> 2);
      if (mb >= 27) continue;                              // (step 15 holds rows 27 and 31: zrows <= 27, never stored)
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      // (locals first: __builtin_bit_cast of a vector ELEMENT expression reads element 0 whatever the index -- found
      // in the ISA: sixteen stores of the same pair)
      const float e0 = zq[0][r], e1 = zq[1][r];
      const u32x2 v = {__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
      const unsigned off = (mb + 4 * lh < a.zrows) ? zo : TOOB;
      __builtin_amdgcn_raw_buffer_store_b64(v, rz, (int)off, (int)((unsigned)mb * (unsigned)ohw * 4u), 0);
    }
  }
}

__global__ __launch_bounds__(ZS_THREADS) void convt3x3s2_z_stream_kernel(ConvTArgs a, int n_img, int slot, int* err,
                                                                         int poll_limit) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  __shared__ unsigned s_ring[ZS_RING][2];
  __shared__ unsigned s_ready, s_done, s_prog[ZS_WAVES];
  const int tid = threadIdx.x, lane = tid & 63;
  // the workgroup's first batch of items is requested before anything else: the round trip to the device-wide counter
  // (microseconds when 256 workgroups start together) passes under the copy of the weights
  unsigned first_gs = 0u;
  if (slot >= 0 && tid == 0) {
    unsigned size = (unsigned)(2 * n_img * a.h * a.tiles_x) / (3u * gridDim.x);
    size = size < 4u ? 4u : (size > 24u ? 24u : size);
    first_gs = (atomicAdd(g_zs_slot[slot], size) << 5) | size;
  }
  // ---- the resident operands: weights (zero-filled past nchunk), contraction operand, bias ----
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.wpk);
    f32x4* dst = reinterpret_cast<f32x4*>(s_w);
    const int nv = a.nchunk * (9 * CK * TOCB) / 4;
    for (int i = tid; i < nv; i += ZS_THREADS) dst[i] = src[i];
    for (int i = tid; i < 2 * 16 * 64; i += ZS_THREADS) s_w[ZS_WZ_OFF + i] = a.wz[i];
    if (tid < 64) s_w[ZS_BIAS_OFF + tid] = (a.bias && tid < a.cout) ? a.bias[tid] : 0.f;
    if (tid == 64) { s_ready = 0u; s_done = 0u; }
    if (tid >= 128 && tid < 128 + ZS_WAVES) s_prog[tid - 128] = 0u;
  }
  __syncthreads();
  const int per_par = n_img * a.h * a.tiles_x;             // items of one parity
  const int nitems = 2 * per_par;
  const int nwg = gridDim.x;
  auto run = [&](int item) {
    const int py = item < per_par ? 1 : 0;                 // the 6-tap items first
    int r = item - (py ? 0 : per_par);
    const int tx = r % a.tiles_x; r /= a.tiles_x;
    const int y = r % a.h;
    const int n = r / a.h;
    if (py) zs_item<1>(a, s_w, n, y, tx * TTW, lane);
    else zs_item<0>(a, s_w, n, y, tx * TTW, lane);
  };
  if (slot < 0) {
    // static list, balanced per SIMD: the three waves of a SIMD share its matrix pipe, so what has to be equal is the
    // SIMD's total.  6-tap items (7 cost units with their contraction) are dealt round-robin over the S SIMDs; the
    // SIMDs that got one fewer of them receive one 3-tap item (4 units) each ahead of the round-robin deal of the
    // rest, which also starts with them.  268x640 on 1024 SIMDs: 3712 / 3776 / 3520 MFMAs per SIMD (average 3685)
    // instead of 3968 / 3776 / 3520 from a plain round-robin; a SIMD's list is dealt to its waves in turn.
    const int S = nwg * 4, sig = blockIdx.x * 4 + ((tid >> 6) & 3), slot3 = tid >> 8;
    const int nbig = per_par, nsmall = per_par;
    const int X = nbig % S;                                // SIMDs [0, X) hold one 6-tap item more
    const int comp = X ? S - X : 0;                        // compensation items, one per light SIMD
    const int nb = (nbig - sig + S - 1) / S;               // this SIMD's 6-tap items
    const bool light = X && sig >= X;
    for (int p = slot3;; p += 3) {
      int item;
      if (p < nb) item = sig + S * p;
      else {
        int q = p - nb, j;
        if (light && q == 0) j = sig - X;
        else {
          if (light) q -= 1;
          j = comp + q * S + (sig - X + S) % S;
        }
        if (j >= nsmall) break;
        item = nbig + j;
      }
      run(item);
    }
    return;
  }
  // ---- dynamic list.  A device-wide atomic costs microseconds on this part (measured: one per item, 10 720 of them,
  // adds 55 us to the launch), so the counter hands out BATCHES of items to workgroups -- ~900 atomics per launch.
  // Inside a workgroup a batch is dealt round-robin to the 12 waves (the deal continues across batches, so every
  // wave gets the same number of items +- 1), without a barrier: batch k is published in an LDS ring by the wave
  // (k mod 12), one batch AHEAD of its use (nobody waits for the round trip in steady state); a wave that reaches
  // batch k before it is published polls the publication count.  Batch sizes shrink with the items left (guided
  // self-scheduling, 4..24).  Every poll is bounded: on a time-out the wave counts a fault in pinned host memory and
  // leaves (the launcher then reports TG_E_HIP at the next call and keeps to the tiled form).
  const int wv = tid >> 6;
  unsigned* const ctr = g_zs_slot[slot];
  auto request = [&](unsigned lastg) -> unsigned {         // one batch from the device-wide counter: (start << 5) | size
    const unsigned left = (unsigned)nitems > lastg ? (unsigned)nitems - lastg : 0u;
    unsigned size = left / (3u * (unsigned)nwg);
    size = size < 4u ? 4u : (size > 24u ? 24u : size);
    return (atomicAdd(ctr, size) << 5) | size;
  };
  // ring entry k % ZS_RING: {start, size | (first wave of the deal) << 8 | terminal << 16}
  auto publish = [&](int k, unsigned gs, unsigned rot) {
    const unsigned g = gs >> 5, size = gs & 31u;
    const bool term = g >= (unsigned)nitems;
    const unsigned n_here = term ? 0u : ((g + size < (unsigned)nitems) ? size : (unsigned)nitems - g);
    s_ring[k % ZS_RING][0] = g;
    s_ring[k % ZS_RING][1] = n_here | (rot << 8) | ((term ? 1u : 0u) << 16);
    __hip_atomic_store(&s_ready, (unsigned)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (tid == 0) publish(0, first_gs, 0u);
  __syncthreads();
  bool fault = false;
  for (int k = 0; !fault; ++k) {
    // batch k (published at the latest by the wave that is about to ask for batch k + 1 ... or by the prologue)
    if (lane == 0) {
      int polls = 0;
      while (__hip_atomic_load(&s_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= (unsigned)k) {
        __builtin_amdgcn_s_sleep(4);
        if (++polls > poll_limit) { fault = true; break; }
      }
    }
    fault = __builtin_amdgcn_readfirstlane((int)fault) != 0;
    if (fault) break;
    const unsigned g = s_ring[k % ZS_RING][0], meta = s_ring[k % ZS_RING][1];
    const int n_here = (int)(meta & 255u), rot = (int)((meta >> 8) & 255u);
    if (meta >> 16) break;                                 // the list is exhausted
    // the wave whose turn it is asks for batch k + 1 before it works on batch k
    if (wv == (k + 1) % ZS_WAVES) {
      if (k + 1 >= ZS_RING && lane == 0) {                 // (never in practice: nobody may still be ZS_RING batches behind)
        int polls = 0;
        for (;;) {
          unsigned lo = 0xFFFFFFFFu;
          for (int w2 = 0; w2 < ZS_WAVES; ++w2) { const unsigned pv = __hip_atomic_load(&s_prog[w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); lo = pv < lo ? pv : lo; }
          if ((unsigned)(k + 1) < lo + (unsigned)ZS_RING) break;
          __builtin_amdgcn_s_sleep(8);
          if (++polls > poll_limit) { fault = true; break; }
        }
      }
      fault = __builtin_amdgcn_readfirstlane((int)fault) != 0;
      if (fault) break;
      unsigned gs = 0u;
      if (lane == 0) gs = request(g);
      gs = (unsigned)__builtin_amdgcn_readfirstlane((int)gs);
      if (lane == 0) publish(k + 1, gs, (unsigned)((rot + n_here) % ZS_WAVES));
    }
    if (lane == 0) __hip_atomic_store(&s_prog[wv], (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // this wave's share of batch k: positions p with (rot + p) % 12 == wv
    for (int pos = (wv - rot + ZS_WAVES) % ZS_WAVES; pos < n_here; pos += ZS_WAVES) run((int)g + pos);
  }
  if (fault && lane == 0 && err) __hip_atomic_fetch_add(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // the last wave of the launch hands the counter slot back clean (one count per workgroup)
  if (lane == 0 && __hip_atomic_fetch_add(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == ZS_WAVES - 1) {
    __threadfence();
    if (atomicAdd(ctr + 1, 1u) == (unsigned)nwg - 1u) {
      atomicExch(ctr, 0u);
      atomicExch(ctr + 1, 0u);
    }
  }
}

// A operand of the Z-mode contraction: wz[(half*16 + r)*64 + l] = Wout[o][c][tap] for tap-plane
// row m = l & 31 (m = tap*cz + o < 9*cz, else 0) and channel c = half*32 + (r&3) + 8*(r>>2) +
// 4*(l>>5) (< nf, else 0).  Wout is the output conv's OIHW weight (cz, nf, 3, 3).
__global__ void convt_pack_wz_kernel(const float* __restrict__ wout, float* __restrict__ wz, int cz,
                                     int nf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * 16 * 64) return;
  const int l = i & 63, r = (i >> 6) & 15, half = i >> 10;
  const int m = l & 31, c = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
  float v = 0.f;
  if (m < 9 * cz && c < nf) {
    const int tap = m / cz, o = m - tap * cz;
    v = wout[((long long)o * nf + c) * 9 + tap];
  }
  wz[i] = v;
}

// out[o][Y][X] = bias[o] + sum_tap z[tap*cz + o][Y + ky - 1][X + kx - 1]  (zero outside the image:
// the output conv's zero padding) + upsample_func(lr_curr) (tecogan_nets.py:145), and the uint8
// HWC frame (data_utils.py:80-87).  One thread per HR pixel; every z element is read once.
template <int CZ>
__global__ __launch_bounds__(256) void convout_tail_kernel(const float* __restrict__ z, long long z_ns,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ up, int up_mode,
                                                           int up_scale, float* __restrict__ y,
                                                           long long y_ns, uint8_t* __restrict__ u8,
                                                           int n, int h, int w) {
  const int X = blockIdx.x * 64 + (threadIdx.x & 63);
  const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (X >= w || Y >= h) return;
  const long long hw = (long long)h * w;
  const float* zb = z + (long long)b * z_ns;
  float v[CZ];
#pragma unroll
  for (int o = 0; o < CZ; ++o) v[o] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = Y + ky - 1;
    if (yy < 0 || yy >= h) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = X + kx - 1;
      if (xx < 0 || xx >= w) continue;
      const float* p = zb + (long long)((ky * 3 + kx) * CZ) * hw + (long long)yy * w + xx;
#pragma unroll
      for (int o = 0; o < CZ; ++o) v[o] += p[(long long)o * hw];
    }
  }
#pragma unroll
  for (int o = 0; o < CZ; ++o) v[o] += bias ? bias[o] : 0.f;
  if (up) {
    const int lh = h / up_scale, lw = w / up_scale;
    const float* src = up + (long long)b * CZ * lh * lw;
    if (up_mode == TG_UP_BICUBIC) {
      const int i = Y / up_scale, dy = Y - i * up_scale, j = X / up_scale, dx = X - j * up_scale;
      float kyw[4], kxw[4];
      bicubic_w(dy, up_scale, kyw);
      bicubic_w(dx, up_scale, kxw);
      int ri[4], ci[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        int r = i - 1 + p; ri[p] = r < 0 ? 0 : (r > lh - 1 ? lh - 1 : r);
        int c = j - 1 + p; ci[p] = c < 0 ? 0 : (c > lw - 1 ? lw - 1 : c);
      }
#pragma unroll
      for (int o = 0; o < CZ; ++o) {
        const float* s = src + (long long)o * lh * lw;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float vq = 0.f;
#pragma unroll
          for (int p = 0; p < 4; ++p) vq += kyw[p] * s[ri[p] * lw + ci[q]];
          acc += kxw[q] * vq;
        }
        v[o] += acc;
      }
    } else {
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      bilinear_src(Y, up_scale, lh, y0, y1, ly0, ly1);
      bilinear_src(X, up_scale, lw, x0, x1, lx0, lx1);
#pragma unroll
      for (int o = 0; o < CZ; ++o) {
        const float* s = src + (long long)o * lh * lw;
        float top = lx0 * s[y0 * lw + x0] + lx1 * s[y0 * lw + x1];
        float bot = lx0 * s[y1 * lw + x0] + lx1 * s[y1 * lw + x1];
        v[o] += ly0 * top + ly1 * bot;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CZ; ++o) y[(long long)b * y_ns + (long long)o * hw + (long long)Y * w + X] = v[o];
  if (u8) {
#pragma unroll
    for (int o = 0; o < CZ; ++o) {
      float r = rintf(v[o] * 255.0f);
      r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
      u8[(((long long)b * h + Y) * w + X) * CZ + o] = (uint8_t)r;      // (n, h, w, cz)
    }
  }
}


// The same tail, FOUR horizontally adjacent HR pixels per thread (round 6; w % 4 == 0 and 16-byte aligned planes): a tap
// plane is read as one (unaligned) 16-byte load instead of four 4-byte ones -- 27 loads per thread instead of 108 per
// four threads (the texture path's cost is per lane and instruction, not per byte) --, the bicubic residual's vertical
// pass is shared by the four pixels of an LR column, the fp32 frame leaves as 16-byte stores and the uint8 frame as one
// 12-byte store.  Every output is accumulated in the scalar kernel's order: bit-identical.
template <int CZ>
__global__ __launch_bounds__(256) void convout_tail4_kernel(const float* __restrict__ z, long long z_ns,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ up, int up_mode,
                                                            int up_scale, float* __restrict__ y,
                                                            long long y_ns, uint8_t* __restrict__ u8,
                                                            int n, int h, int w) {
  const int X = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
  const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (X >= w || Y >= h) return;
  const unsigned hw = (unsigned)h * (unsigned)w;
  // one image's planes: 32-bit offsets against a buffer resource (out-of-range reads at the two ends of the buffer: 0)
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(z + (long long)b * z_ns), 0, (int)(9u * CZ * hw * 4u), 0x00020000);
  float v[CZ][4];
#pragma unroll
  for (int o = 0; o < CZ; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[o][e] = 0.f;
  const bool left = X == 0, right = X + 4 >= w;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = Y + ky - 1;
    if (yy < 0 || yy >= h) continue;                       // (uniform: a wave covers one row)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned base = ((unsigned)yy * (unsigned)w + (unsigned)(X + kx - 1)) * 4u;     // X = 0, kx = 0: wraps to 0xFFFFFFFC -> out of range -> 0, fixed below anyway
#pragma unroll
      for (int o = 0; o < CZ; ++o) {
        const unsigned pl = (unsigned)((ky * 3 + kx) * CZ + o) * hw * 4u;
        f32x4 t;
        if (kx == 0 && left) {                             // the row's first thread: elements X .. X + 2 only
          t[0] = 0.f;
#pragma unroll
          for (int e = 1; e < 4; ++e)
            t[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, (int)(base + 4u * e + pl), 0, 0));
        } else {
          t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)(base + pl), 0, 0));
          if (kx == 2 && right) t[3] = 0.f;                // X + 4 is the next row's first pixel: the conv's zero padding
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[o][e] += t[e];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CZ; ++o) {
    const float bb = bias ? bias[o] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[o][e] += bb;
  }
  if (up) {
    const int lh = h / up_scale, lw = w / up_scale;
    const float* src = up + (long long)b * CZ * lh * lw;
    if (up_mode == TG_UP_BICUBIC) {
      const int i = Y / up_scale, dy = Y - i * up_scale;
      float kyw[4];
      bicubic_w(dy, up_scale, kyw);
      int ri[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) { int r = i - 1 + p; ri[p] = r < 0 ? 0 : (r > lh - 1 ? lh - 1 : r); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = (X + e) / up_scale, dx = (X + e) - j * up_scale;
        float kxw[4];
        bicubic_w(dx, up_scale, kxw);
        int ci[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { int c = j - 1 + p; ci[p] = c < 0 ? 0 : (c > lw - 1 ? lw - 1 : c); }
#pragma unroll
        for (int o = 0; o < CZ; ++o) {
          const float* s_ = src + (long long)o * lh * lw;
          float acc = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float vq = 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p) vq += kyw[p] * s_[ri[p] * lw + ci[q]];
            acc += kxw[q] * vq;
          }
          v[o][e] += acc;
        }
      }
    } else {
      int y0, y1; float ly0, ly1;
      bilinear_src(Y, up_scale, lh, y0, y1, ly0, ly1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int x0, x1; float lx0, lx1;
        bilinear_src(X + e, up_scale, lw, x0, x1, lx0, lx1);
#pragma unroll
        for (int o = 0; o < CZ; ++o) {
          const float* s_ = src + (long long)o * lh * lw;
          float top = lx0 * s_[y0 * lw + x0] + lx1 * s_[y0 * lw + x1];
          float bot = lx0 * s_[y1 * lw + x0] + lx1 * s_[y1 * lw + x1];
          v[o][e] += ly0 * top + ly1 * bot;
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CZ; ++o)
    *reinterpret_cast<f32x4*>(y + (long long)b * y_ns + (long long)o * hw + (long long)Y * w + X) =
        f32x4{v[o][0], v[o][1], v[o][2], v[o][3]};
  if (u8) {
    uint8_t q[4 * CZ];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 0; o < CZ; ++o) {
        float r = rintf(v[o][e] * 255.0f);
        r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
        q[e * CZ + o] = (uint8_t)r;
      }
    uint8_t* dst = u8 + (((long long)b * h + Y) * w + X) * CZ;      // (n, h, w, cz): 4 * cz contiguous bytes
    if constexpr (CZ == 3) {
      unsigned w3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        w3[k] = (unsigned)q[4 * k] | ((unsigned)q[4 * k + 1] << 8) | ((unsigned)q[4 * k + 2] << 16) | ((unsigned)q[4 * k + 3] << 24);
      unsigned* d32 = reinterpret_cast<unsigned*>(dst);              // X % 4 == 0 -> 12-byte steps: 4-byte aligned
      d32[0] = w3[0]; d32[1] = w3[1]; d32[2] = w3[2];
    } else {
#pragma unroll
      for (int k = 0; k < 4 * CZ; ++k) dst[k] = q[k];
    }
  }
}


// The Z-mode launcher's split-tail rule (also asked by the frame plan's launch accounting): 4-row workgroups, one image,
// at least one whole round of the 2-per-CU resident slots, and a last round between a quarter and 95 % full (form 3: always
// when a whole round exists; form 0: never).
bool convt_z_split_rule(int n, int h, int w, int form) {
  if (n != 1 || !(form == 3 || form == -1)) return false;
  const long long tiles_x = cdiv(w, TTW);
  const long long wg4 = tiles_x * cdiv(h, 4) * n;
  if (wg4 < 512) return false;                       // (the launcher uses 2-row workgroups there)
  static int ncu_s = 0;
  if (!ncu_s) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu_s, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); ncu_s = 256; } }
  const long long slots = 2ll * ncu_s, rem = wg4 % slots;
  if (wg4 < slots) return false;
  const long long ty4 = (wg4 / slots) * slots / tiles_x;
  if (ty4 < 1 || ty4 >= cdiv(h, 4)) return false;
  return form == 3 || (4 * rem >= slots && 100 * rem <= 95 * slots);
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
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_LRELU02, TG_E_ARG, "convt: act=%d (none|relu|lrelu)", act);
  TG_REQUIRE((y_nstride % 2) == 0 && ((uintptr_t)y % 8) == 0, TG_E_ARG,
             "convt3x3s2_fwd: output must be 8-byte aligned");
  TG_REQUIRE((long long)(cin + CK) * h * w * 4 < (1ll << 31), TG_E_SHAPE,
             "convt3x3s2_fwd: one batch item must be < 2 GiB");
  ConvTArgs a{};
  a.x = x; a.wpk = w_packed; a.bias = bias; a.y = y; a.x_ns = x_nstride; a.y_ns = y_nstride;
  a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = act;
  constexpr int WN = 2;
  a.tiles_x = cdiv(w, TTW);
  a.nocg = cdiv(cout, TOCB);
  a.nchunk = cdiv(cin, CK);
  // at most one one-row tile per CU (the training frames): everything in flight at once
  if (cin <= 64 && cout <= 64 && (long long)a.tiles_x * h * n <= 256) {
    a.tiles_y = h;
    const size_t lds1 = (size_t)(8 * TOS_CH_FLOATS + 2 * 4 * 4 * 16 * 64) * sizeof(float);      // 148 KB
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convt3x3s2_oneshot_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      attr_set = true;
    }
    hipLaunchKernelGGL(convt3x3s2_oneshot_kernel, dim3((unsigned)(a.tiles_x * h * n)), dim3(512), lds1,
                       (hipStream_t)stream, a);
    return check_launch("convt3x3s2_oneshot");
  }
  // 4-row workgroups (8 waves) unless that leaves fewer than two workgroups per CU: a 134x320
  // input is 340 of them = 1.33 per CU (the CUs with two finish last: 66 % balance); 2-row
  // workgroups (670 = 2.62 per CU, 87 %) take the small frames.  TG_CONVT_ROWS overrides (lab).
  static const int rows_env = TG_LAB_ENV("TG_CONVT_ROWS", 0);
  const long long wg4 = (long long)a.tiles_x * cdiv(h, 4) * a.nocg * n;
  const int rows = rows_env == 2 || rows_env == 4 ? rows_env : (wg4 < 512 ? 2 : 4);
  a.tiles_y = cdiv(h, rows);
  size_t lds = 2 * (size_t)((rows + 1) * 2 * TRS * 4 + 9 * CK * TOCB) * sizeof(float);
  long long blocks = (long long)a.tiles_x * a.tiles_y * a.nocg * n;
  TG_REQUIRE(blocks > 0 && blocks < (1ll << 31), TG_E_SHAPE, "convt: grid %lld", blocks);
  if (rows == 2)
    hipLaunchKernelGGL((convt3x3s2_mfma_kernel<2, WN>), dim3((unsigned)blocks), dim3(2 * WN * 64), lds,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((convt3x3s2_mfma_kernel<4, WN>), dim3((unsigned)blocks), dim3(4 * WN * 64), lds,
                       (hipStream_t)stream, a);
  return check_launch("convt3x3s2_mfma");
}

extern "C" int tg_convt_pack_wz(const float* w_out_oihw, float* wz, int cz, int nf, tg_stream_t stream) {
  TG_REQUIRE(w_out_oihw && wz, TG_E_ARG, "convt_pack_wz: null pointer");
  TG_REQUIRE(cz >= 1 && 9 * cz <= 32 && nf >= 1 && nf <= 64, TG_E_SHAPE, "convt_pack_wz: cz=%d (<=3) nf=%d (<=64)", cz, nf);
  hipLaunchKernelGGL(convt_pack_wz_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, w_out_oihw, wz, cz, nf);
  return check_launch("convt_pack_wz");
}

extern "C" int tg_convt3x3s2_z_fwd(const float* x, int64_t x_nstride, const float* w_packed,
                                   const float* bias, const float* wz, int cz, float* z,
                                   int64_t z_nstride, int n, int cin, int cout, int h, int w, int act,
                                   tg_stream_t stream) {
  static const int form_env = TG_LAB_ENV("TG_CONVTZ_FORM", -1);     // lab builds: A/B of the forms through the frame plan
  return tg_convt3x3s2_z_fwd_form(x, x_nstride, w_packed, bias, wz, cz, z, z_nstride, n, cin, cout, h, w, act, form_env, stream);
}

extern "C" int tg_convt3x3s2_z_fwd_form(const float* x, int64_t x_nstride, const float* w_packed,
                                        const float* bias, const float* wz, int cz, float* z,
                                        int64_t z_nstride, int n, int cin, int cout, int h, int w, int act,
                                        int form, tg_stream_t stream) {
  TG_REQUIRE(form >= -1 && form <= 3, TG_E_ARG, "convt3x3s2_z_fwd_form: form=%d (-1 auto, 0 tiled, 1 streaming, 2 streaming with a static item list, 3 tiled with a split tail)", form);
  TG_REQUIRE(x && w_packed && wz && z, TG_E_ARG, "convt3x3s2_z_fwd: null pointer");
  TG_REQUIRE(n > 0 && cin > 0 && cout > 0 && cout <= 64 && h > 0 && w > 0 && cz >= 1 && 9 * cz <= 32,
             TG_E_SHAPE, "convt3x3s2_z_fwd: n=%d cin=%d cout=%d (<=64) h=%d w=%d cz=%d (<=3)", n, cin, cout, h, w, cz);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_LRELU02, TG_E_ARG, "convt_z: act=%d", act);
  TG_REQUIRE((z_nstride % 2) == 0 && ((uintptr_t)z % 8) == 0, TG_E_ARG, "convt3x3s2_z_fwd: z must be 8-byte aligned");
  TG_REQUIRE((long long)(cin + CK) * h * w * 4 < (1ll << 31) && 32ll * 4 * h * w * 4 < (1ll << 31), TG_E_SHAPE,
             "convt3x3s2_z_fwd: one batch item (input and the 32 planes) must be < 2 GiB");
  ConvTArgs a{};
  a.x = x; a.wpk = w_packed; a.bias = bias; a.y = nullptr; a.x_ns = x_nstride; a.y_ns = 0;
  a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = act;
  a.wz = wz; a.z = z; a.z_ns = z_nstride; a.zrows = 9 * cz;
  constexpr int WN = 2;
  static const int rows_env = TG_LAB_ENV("TG_CONVTZ_ROWS", 0);
  a.tiles_x = cdiv(w, TTW); a.nocg = 1; a.nchunk = cdiv(cin, CK);
  // the streaming form (weights LDS-resident, autonomous waves) wherever every wave of the chip finds a few items
  {
    const int stream_env = form == 3 ? 0 : form;
    static int ncu = -1;
    static bool attr_ok = false;
    if (ncu < 0) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ncu = v;
      else { (void)hipGetLastError(); ncu = 0; }
      attr_ok = ncu > 0 && hipFuncSetAttribute(reinterpret_cast<const void*>(convt3x3s2_z_stream_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZS_LDS_BYTES) == hipSuccess;
      if (!attr_ok) (void)hipGetLastError();
    }
    // fail-safe of the streaming form's polls: a fault counter in pinned host memory, looked at on entry
    static int* zs_err = nullptr;
    static bool zs_off = false;
    if (form >= 1 && attr_ok && !zs_err && !zs_off) {      // (only a caller that asks for the streaming form pays for its 64 pinned bytes)
      void* hp = nullptr;
      if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hp) { zs_err = static_cast<int*>(hp); *zs_err = 0; }
      else { (void)hipGetLastError(); zs_off = true; }
    }
    if (zs_err && __atomic_load_n(zs_err, __ATOMIC_RELAXED) != 0) {
      const int f = __atomic_exchange_n(zs_err, 0, __ATOMIC_RELAXED);
      zs_off = true;
      tg::set_error("convt3x3s2_z (streaming form): %d wave(s) timed out waiting for a batch of work items; the planes written "
                    "by launches since the previous call are INVALID.  The tiled form is used from now on", f);
      return TG_E_HIP;
    }
    const long long items = 2ll * n * h * a.tiles_x;
    const bool fits = attr_ok && !zs_off && (zs_err || form < 1) && cin <= 64 && cout <= 64 && items < (1ll << 30) && 32ll * 4 * h * w * 4 < (1ll << 31);
    TG_REQUIRE(form < 1 || fits, TG_E_SHAPE, "convt3x3s2_z_fwd_form: the streaming form needs cin, cout <= 64 and 160 KB of LDS");
    // The rule keeps the TILED form (round 6, EXPERIMENTS.md): stand-alone the streaming form with the static item list
    // is 3 % faster at 268x640 (143.5 vs 148 us), through the whole frame it is +-0 (1463-1467 vs 1462 frames/s), and
    // its dynamic list -- the form that is robust against a co-running kernel holding CUs -- is 12 % slower (the
    // device-wide counter).  Both stay selectable (form 1 / 2) and are tested bit-identical to the tiled form.
    const bool want = stream_env > 0;
    (void)ncu;
    if (fits && want) {
      static unsigned seq = 0;
      const int slot = form == 2 ? -1 : (int)(seq++ % ZS_SLOTS);
      const int grid = (int)(items / ZS_WAVES < ncu ? cdiv((int)items, ZS_WAVES) : ncu);
      hipLaunchKernelGGL(convt3x3s2_z_stream_kernel, dim3((unsigned)grid), dim3(ZS_THREADS), ZS_LDS_BYTES,
                         (hipStream_t)stream, a, n, slot, zs_err, 1 << 20);
      return check_launch("convt3x3s2_z_stream");
    }
  }
  // same balance rule as tg_convt3x3s2_fwd (at 268x640, 1340 four-row workgroups, the two-row form
  // measured 153 vs 148 us: more workgroups than slots are balanced by the dispatcher anyway)
  const long long wg4 = (long long)a.tiles_x * cdiv(h, 4) * n;
  const int rows = rows_env == 2 || rows_env == 4 ? rows_env : (wg4 < 512 ? 2 : 4);
  a.tiles_y = cdiv(h, rows);
  size_t lds = 2 * (size_t)((rows + 1) * 2 * TRS * 4 + 9 * CK * TOCB) * sizeof(float);   // >= 2*rows*16*64*4 B
  long long blocks = (long long)a.tiles_x * a.tiles_y * n;
  TG_REQUIRE(blocks > 0 && blocks < (1ll << 31), TG_E_SHAPE, "convt_z: grid %lld", blocks);
  // Split tail (round 6): 1 340 four-row workgroups are 2.62 rounds of the 512 resident slots, i.e. 3 -- so the whole
  // rounds run as four-row workgroups and the REMAINING rows as two-row workgroups (4 resident per CU: a finer last
  // round) in a second launch behind the first.  Same kernels, same arithmetic per output: bit-identical; 144.7 -> 141.2 us
  // at 268x640 (tools/convtz_lab.py).  The rule takes it when at least one whole round exists and the last round is
  // between a quarter and 95 % full; form 0 never splits, form 3 always does when a whole round exists.
  const bool split = convt_z_split_rule(n, h, w, form);
  if (split) {
    int ncu3 = 256;
    { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu3, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); ncu3 = 256; } }
    const int slots = 2 * ncu3;
    const int full_rounds = (int)(blocks / slots);
    const int ty4 = full_rounds * slots / a.tiles_x;             // tile rows of the first launch
    if (full_rounds >= 1 && ty4 >= 1 && ty4 < a.tiles_y) {
      ConvTArgs a4 = a; a4.tiles_y = ty4; a4.y_base = 0;
      hipLaunchKernelGGL((convt3x3s2_mfma_kernel<4, WN, true>), dim3((unsigned)(a.tiles_x * ty4)), dim3(4 * WN * 64), lds,
                         (hipStream_t)stream, a4);
      ConvTArgs a2 = a; a2.y_base = 4 * ty4; a2.tiles_y = cdiv(h - 4 * ty4, 2);
      const size_t lds2 = 2 * (size_t)((2 + 1) * 2 * TRS * 4 + 9 * CK * TOCB) * sizeof(float);
      hipLaunchKernelGGL((convt3x3s2_mfma_kernel<2, WN, true>), dim3((unsigned)(a.tiles_x * a2.tiles_y)), dim3(2 * WN * 64), lds2,
                         (hipStream_t)stream, a2);
      return check_launch("convt3x3s2_z(split)");
    }
  }
  if (rows == 2)
    hipLaunchKernelGGL((convt3x3s2_mfma_kernel<2, WN, true>), dim3((unsigned)blocks), dim3(2 * WN * 64), lds,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((convt3x3s2_mfma_kernel<4, WN, true>), dim3((unsigned)blocks), dim3(4 * WN * 64), lds,
                       (hipStream_t)stream, a);
  return check_launch("convt3x3s2_z");
}

extern "C" int tg_convout_tail(const float* z, int64_t z_nstride, int cz, const float* bias,
                               const float* up_src, int up_mode, int up_scale, float* y,
                               int64_t y_nstride, uint8_t* u8_out, int n, int h, int w,
                               tg_stream_t stream) {
  return tg_convout_tail_form(z, z_nstride, cz, bias, up_src, up_mode, up_scale, y, y_nstride, u8_out, n, h, w, -1, stream);
}

extern "C" int tg_convout_tail_form(const float* z, int64_t z_nstride, int cz, const float* bias,
                                    const float* up_src, int up_mode, int up_scale, float* y,
                                    int64_t y_nstride, uint8_t* u8_out, int n, int h, int w, int form,
                                    tg_stream_t stream) {
  TG_REQUIRE(z && y, TG_E_ARG, "convout_tail: null pointer");
  TG_REQUIRE(n > 0 && h > 0 && w > 0 && cz >= 1 && cz <= 3, TG_E_SHAPE, "convout_tail: n=%d h=%d w=%d cz=%d", n, h, w, cz);
  TG_REQUIRE(form >= -1 && form <= 1, TG_E_ARG, "convout_tail: form=%d (-1 the rule, 0 one pixel per thread, 1 four)", form);
  if (up_src)
    TG_REQUIRE((up_mode == TG_UP_BICUBIC || up_mode == TG_UP_BILINEAR) && up_scale >= 1 && h % up_scale == 0 &&
                   w % up_scale == 0, TG_E_SHAPE, "convout_tail: up_mode=%d up_scale=%d", up_mode, up_scale);
  hipStream_t s = (hipStream_t)stream;
  // four pixels per thread wherever rows and planes are 16-byte aligned (and the uint8 rows 4-byte aligned for cz = 3)
  const bool vec_ok = (w % 4) == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)y % 16) == 0 && (z_nstride % 4) == 0 &&
                      (y_nstride % 4) == 0 && (!u8_out || ((uintptr_t)u8_out % 4) == 0) &&
                      9ll * cz * h * w * 4 < (1ll << 31);
  TG_REQUIRE(form != 1 || vec_ok, TG_E_SHAPE, "convout_tail: the four-pixel form needs w %% 4 == 0 and 16-byte aligned planes");
  if (form == 1 || (form == -1 && vec_ok)) {
    dim3 g4(cdiv(w / 4, 64), cdiv(h, 4), n), t4(256);
    switch (cz) {
      case 1: hipLaunchKernelGGL(convout_tail4_kernel<1>, g4, t4, 0, s, z, (long long)z_nstride, bias, up_src, up_mode, up_scale, y, (long long)y_nstride, u8_out, n, h, w); break;
      case 2: hipLaunchKernelGGL(convout_tail4_kernel<2>, g4, t4, 0, s, z, (long long)z_nstride, bias, up_src, up_mode, up_scale, y, (long long)y_nstride, u8_out, n, h, w); break;
      default: hipLaunchKernelGGL(convout_tail4_kernel<3>, g4, t4, 0, s, z, (long long)z_nstride, bias, up_src, up_mode, up_scale, y, (long long)y_nstride, u8_out, n, h, w); break;
    }
    return check_launch("convout_tail4");
  }
  dim3 g(cdiv(w, 64), cdiv(h, 4), n), t(256);
  switch (cz) {
    case 1: hipLaunchKernelGGL(convout_tail_kernel<1>, g, t, 0, s, z, (long long)z_nstride, bias, up_src, up_mode, up_scale, y, (long long)y_nstride, u8_out, n, h, w); break;
    case 2: hipLaunchKernelGGL(convout_tail_kernel<2>, g, t, 0, s, z, (long long)z_nstride, bias, up_src, up_mode, up_scale, y, (long long)y_nstride, u8_out, n, h, w); break;
    default: hipLaunchKernelGGL(convout_tail_kernel<3>, g, t, 0, s, z, (long long)z_nstride, bias, up_src, up_mode, up_scale, y, (long long)y_nstride, u8_out, n, h, w); break;
  }
  return check_launch("convout_tail");
}

// Conv2d(ci, co, k4, stride 2, pad 1, no bias) -- the discriminator blocks (tecogan_nets.py:322-340) -- and its data
// gradient (a 4x4 / stride-2 transposed convolution) as implicit GEMMs on the fp32 matrix cores (gfx950),
// WITHOUT the space-to-depth copy and the phase-masked 3x3 embedding of rounds 2-3 (tg_conv3x3_fwd_phased: the
// embedded kernel stages a 3-row patch per 8-channel chunk and uses 4 of its 9 taps -- 16 MFMAs of 64 cycles per
// staging round, 0.45 of peak on the 128x128 -> 64x64 block).
//
// Forward:   y[n][oc][oy][ox] = sum_ci sum_{ky,kx} w[oc][ci][ky][kx] x[n][ci][2 oy - 1 + ky][2 ox - 1 + kx]
//   K = 16 ci.  A workgroup (4 waves) owns TR output rows x 32 TXH output columns x 64 output channels; a wave one
//   row segment of 32 pixels x 64 channels (two 32 x 32 accumulators of v_mfma_f32_32x32x2_f32).  Per chunk of 4
//   input channels the raw (2 TR + 2) x (64 TXH + 2) patch goes into LDS DE-INTERLEAVED BY COLUMN PARITY -- the
//   stride-2 pixel access of a tap then is a unit-stride 8-byte read per lane (conflict free) -- and the 16 x 64 x 4
//   weights of the chunk as the A operands in consumption order: 64 MFMAs (4 096 cycles) per wave and barrier
//   instead of 16.
// Data gradient:   dx[n][ci][iy][ix] = sum_co sum_{ky = iy + 1 - 2 oy, kx = ix + 1 - 2 ox} w[co][ci][ky][kx] g[n][co][oy][ox]
//   four sub-pixel phases of 2 x 2 taps (K = 4 co each).  A wave owns one dx row (row parity fixed) x 64 columns
//   = 32 even + 32 odd pixels x 64 input channels (four accumulators); the 6 distinct B operands of a chunk (2 rows
//   x 3 column shifts of g) are shared by the two column parities; the epilogue pairs even / odd pixels into 8-byte
//   stores and (optionally) multiplies by act'(x) of the layer below (what tg_depth_to_space_act_bwd did on the way
//   out of the embedding).
#include "tg_common.h"
#include <cstdlib>

namespace tg {

constexpr unsigned C4_OOB = 0x80000000u;
constexpr int C4_CK = 4;      // input channels per chunk (forward): two K steps of v_mfma_f32_32x32x2_f32
constexpr int C4_WCH = 16 * 2 * 64 * 2;   // packed weight floats per (64-oc group, chunk): [tap][ocb][lane][ks]

struct Conv4Args {
  const float* x;
  const float* wpk;
  float* y;
  long long x_ns, y_ns;
  int ci, co, h, w;         // h, w: INPUT size (even); output h / 2 x w / 2
  int tiles_x, tiles_y, nocg, nchunk;
};

// OIHW (co, ci, 4, 4) -> [ocg][chunk][tap][ocb][lane][ks]: lane (m = lane & 31, k = lane >> 5) of K step ks holds
// W[oc = 64 ocg + 32 ocb + m][ci = 4 chunk + 2 ks + k][tap]
__global__ void conv4_pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int ci, int co) {
  const int nchunk = ci / C4_CK;
  const long long total = (long long)(co / 64) * nchunk * C4_WCH;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t = (int)(i % C4_WCH); const int gc = (int)(i / C4_WCH);
    const int ks = t & 1; t >>= 1;
    const int lane = t & 63; t >>= 6;
    const int ocb = t & 1; const int tap = t >> 1;
    const int chunk = gc % nchunk, ocg = gc / nchunk;
    const int oc = 64 * ocg + 32 * ocb + (lane & 31), c = C4_CK * chunk + 2 * ks + (lane >> 5);
    out[i] = w[((size_t)oc * ci + c) * 16 + tap];
  }
}

__device__ __forceinline__ float c4_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

template <int TXH>
__global__ __launch_bounds__(256) void conv4x4s2_mfma_kernel(Conv4Args a) {
  constexpr int TR = 4 / TXH;                 // output rows of the workgroup
  constexpr int TCW = 32 * TXH;               // output columns
  constexpr int PR = 2 * TR + 2, PC = 2 * TCW + 2;   // input patch
  constexpr int CP = TCW + 1;                 // columns per parity plane
  constexpr int CPS = 2 * CP + 2;             // plane stride in floats ([col][ks 2] + pad; 132 / 68: 8-byte aligned)
  constexpr int IN_FLOATS = PR * 2 * 2 * CPS; // [row][k][parity][col][ks]
  constexpr int ITEMS = PR * PC * 2;          // 8-byte items: (row, col, k) -> channels {k, 2 + k}
  constexpr int I_PER_T = (ITEMS + 255) / 256;
  constexpr int W_PER_T = C4_WCH / 4 / 256;   // 16-byte pieces per thread = 8

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                    // [2][IN_FLOATS]
  float* s_w = smem + 2 * IN_FLOATS;     // [2][C4_WCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int ocg = b % a.nocg;
  const int n = b / a.nocg;
  const int oh = a.h >> 1, ow = a.w >> 1;
  const int oy0 = ty * TR, ox0 = tx * TCW;
  const int hw = a.h * a.w;

  // ---- staging assignment ----------------------------------------------------------------------
  unsigned voff[I_PER_T];
  int lds_item[I_PER_T];
#pragma unroll
  for (int i = 0; i < I_PER_T; ++i) {
    const int q = tid + i * 256;
    const int k = q / (PR * PC), rem = q - k * (PR * PC);
    const int r = rem / PC, c = rem - r * PC;
    const int gy = 2 * oy0 - 1 + r, gx = 2 * ox0 - 1 + c;
    const bool ok = q < ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    voff[i] = ok ? (unsigned)((k * hw + gy * a.w + gx) * 4) : C4_OOB;
    lds_item[i] = q < ITEMS ? ((r * 2 + k) * 2 + (c & 1)) * CPS + (c >> 1) * 2 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, (unsigned)(a.ci * hw * 4), 0x00020000);
  const unsigned plane = (unsigned)hw * 4u;
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk + (size_t)ocg * a.nchunk * C4_WCH);

  float2 rin[I_PER_T];
  f32x4 rw[W_PER_T];
  auto load_chunk = [&](int ch) {
    const unsigned cbase = (unsigned)(ch * C4_CK) * plane;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i) {
      rin[i].x = c4_load(rs, voff[i] + cbase);                 // channel 4 ch + k      (K step 0)
      rin[i].y = c4_load(rs, voff[i] + cbase + 2u * plane);    // channel 4 ch + 2 + k  (K step 1)
    }
    const f32x4* ws = wsrc + (size_t)ch * (C4_WCH / 4);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) rw[i] = ws[tid + i * 256];
  };
  auto store_chunk = [&](int buf) {
    float* si = s_in + buf * IN_FLOATS;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i)
      if (lds_item[i] >= 0) *reinterpret_cast<float2*>(si + lds_item[i]) = rin[i];
    f32x4* sw = reinterpret_cast<f32x4*>(s_w + buf * C4_WCH);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) sw[tid + i * 256] = rw[i];
  };

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int lk = lane >> 5, ln = lane & 31;
  const int wr = TXH == 2 ? (wave >> 1) : wave;        // output row of the wave inside the tile
  const int wc = TXH == 2 ? (wave & 1) * 32 : 0;       // first output column of the wave inside the tile
  // B operand of tap (ky, kx): patch row 2 wr + ky, patch column 2 (wc + ln) + kx -> parity kx & 1, index wc + ln + (kx >> 1)
  const int b_off = ((2 * wr) * 2 + lk) * 2 * CPS + (wc + ln) * 2;
  const int a_off = lane * 2;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < a.nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < a.nchunk) load_chunk(ch + 1);
    const float* si = s_in + buf * IN_FLOATS + b_off;
    const float* sw = s_w + buf * C4_WCH + a_off;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float2 bv = *reinterpret_cast<const float2*>(si + (ky * 2 * 2 + (kx & 1)) * CPS + (kx >> 1) * 2);
        const float2 a0 = *reinterpret_cast<const float2*>(sw + ((ky * 4 + kx) * 2 + 0) * 128);
        const float2 a1 = *reinterpret_cast<const float2*>(sw + ((ky * 4 + kx) * 2 + 1) * 128);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bv.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bv.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bv.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bv.y, acc[1], 0, 0, 0);
      }
    if (ch + 1 < a.nchunk) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: D row m = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of the 32-channel block, column = lane & 31 ----
  const int oy = oy0 + wr, ox = ox0 + wc + ln;
  if (oy < oh && ox < ow) {
    float* yo = a.y + (long long)n * a.y_ns + (size_t)(64 * ocg) * oh * ow + (size_t)oy * ow + ox;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * t + 8 * (r >> 2) + 4 * lk + (r & 3);
        yo[(size_t)m * oh * ow] = acc[t][r];
      }
  }
}

template <int TXH> static constexpr int c4_lds_bytes() {
  return (2 * ((2 * (4 / TXH) + 2) * 2 * 2 * (2 * (32 * TXH + 1) + 2)) + 2 * C4_WCH) * 4;
}

// ------------------------------------------------------------------------------------------------------------
// data gradient
// ------------------------------------------------------------------------------------------------------------
constexpr int D4_CK = 8;                          // output channels (of the conv) per chunk: four K steps
constexpr int D4_WCH = 16 * 2 * 64 * 4;           // [tap][cib][lane][ks 4]
constexpr int D4_PC = 34;                         // patch columns: ox = j0 - 1 .. j0 + 32
constexpr int D4_RS = D4_PC * 4 + 4;              // floats per (row, k): [col][ks 4] + pad (140)
constexpr int D4_IN = 4 * 2 * D4_RS;              // [row 4][k 2]

struct Conv4DgradArgs {
  const float* g;      // (n, co, h / 2, w / 2)
  const float* wpk;
  const float* act_y;  // optional: dx *= act'(act_y) (act_y > 0 ? 1 : slope), same layout as dx
  float* dx;           // (n, ci, h, w)
  long long g_ns, dx_ns, act_ns;
  float slope;
  int ci, co, h, w;
  int tiles_x, tiles_y, ncig, nchunk;
};

// OIHW (co, ci, 4, 4) -> [cig][chunk][tap][cib][lane][ks]: lane (m, k) of K step ks holds
// W[co = 8 chunk + 2 ks + k][ci = 64 cig + 32 cib + m][tap]
__global__ void conv4_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int ci, int co) {
  const int nchunk = co / D4_CK;
  const long long total = (long long)(ci / 64) * nchunk * D4_WCH;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int t = (int)(i % D4_WCH); const int gc = (int)(i / D4_WCH);
    const int ks = t & 3; t >>= 2;
    const int lane = t & 63; t >>= 6;
    const int cib = t & 1; const int tap = t >> 1;
    const int chunk = gc % nchunk, cig = gc / nchunk;
    const int c = 64 * cig + 32 * cib + (lane & 31), o = D4_CK * chunk + 2 * ks + (lane >> 5);
    out[i] = w[((size_t)o * ci + c) * 16 + tap];
  }
}

// Workgroup: 4 consecutive dx rows 4 ty .. 4 ty + 3 (wave = row) x 64 columns 64 tx .. x 64 input channels.
// Rows of g needed: oy = 2 ty - 1 .. 2 ty + 2 (4), columns ox = 32 tx - 1 .. 32 tx + 32 (34).
template <bool ACT>
__global__ __launch_bounds__(256) void conv4x4s2_dgrad_mfma_kernel(Conv4DgradArgs a) {
  constexpr int ITEMS = 4 * D4_PC * 2;             // 16-byte items: (row, col, k) -> channels {k, 2 + k, 4 + k, 6 + k}
  constexpr int I_PER_T = (ITEMS + 255) / 256;     // 2
  constexpr int W_PER_T = D4_WCH / 4 / 256;        // 8
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                    // [2][D4_IN]
  float* s_w = smem + 2 * D4_IN;         // [2][D4_WCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int cig = b % a.ncig;
  const int n = b / a.ncig;
  const int oh = a.h >> 1, ow = a.w >> 1, ohw = oh * ow;
  const int oyb = 2 * ty - 1, oxb = 32 * tx - 1;

  unsigned voff[I_PER_T];
  int lds_item[I_PER_T];
#pragma unroll
  for (int i = 0; i < I_PER_T; ++i) {
    const int q = tid + i * 256;
    const int k = q / (4 * D4_PC), rem = q - k * (4 * D4_PC);
    const int r = rem / D4_PC, c = rem - r * D4_PC;
    const int gy = oyb + r, gx = oxb + c;
    const bool ok = q < ITEMS && gy >= 0 && gy < oh && gx >= 0 && gx < ow;
    voff[i] = ok ? (unsigned)((k * ohw + gy * ow + gx) * 4) : C4_OOB;
    lds_item[i] = q < ITEMS ? (r * 2 + k) * D4_RS + c * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.g + (long long)n * a.g_ns), 0, (unsigned)(a.co * ohw * 4), 0x00020000);
  const unsigned plane = (unsigned)ohw * 4u;
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk + (size_t)cig * a.nchunk * D4_WCH);

  f32x4 rin[I_PER_T];
  f32x4 rw[W_PER_T];
  auto load_chunk = [&](int ch) {
    const unsigned cbase = (unsigned)(ch * D4_CK) * plane;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) rin[i][j] = c4_load(rs, voff[i] + cbase + (unsigned)(2 * j) * plane);
    const f32x4* ws = wsrc + (size_t)ch * (D4_WCH / 4);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) rw[i] = ws[tid + i * 256];
  };
  auto store_chunk = [&](int buf) {
    float* si = s_in + buf * D4_IN;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i)
      if (lds_item[i] >= 0) *reinterpret_cast<f32x4*>(si + lds_item[i]) = rin[i];
    f32x4* sw = reinterpret_cast<f32x4*>(s_w + buf * D4_WCH);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) sw[tid + i * 256] = rw[i];
  };

  f32x16 acc[2][2];        // [column parity][channel block]
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;

  const int lk = lane >> 5, ln = lane & 31;
  const int iy = 4 * ty + wave, py = iy & 1;
  // row parity 0 (iy = 2 m): (oy = m, ky = 1), (oy = m - 1, ky = 3); parity 1: (oy = m + 1, ky = 0), (oy = m, ky = 2).
  // patch row of oy = m is (m - oyb) = (wave >> 1) + 1.
  const int prow_m = (wave >> 1) + 1;
  // dx column ix = 2 j + px, j = 32 tx + ln: px 0: (ox = j, kx = 1), (ox = j - 1, kx = 3); px 1: (ox = j + 1, kx = 0), (ox = j, kx = 2)
  // patch column of ox = j is ln + 1.
  const int b_lane = lk * D4_RS + (ln + 1) * 4;
  const int a_off = lane * 4;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < a.nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < a.nchunk) load_chunk(ch + 1);
    const float* si = s_in + buf * D4_IN + b_lane;
    const float* sw = s_w + buf * D4_WCH + a_off;
#pragma unroll
    for (int e = 0; e < 2; ++e) {                       // the two g rows of this dx row
      const int prow = py == 0 ? prow_m - e : prow_m + 1 - e;      // py 0: m, m - 1;  py 1: m + 1, m
      const int ky = py == 0 ? 1 + 2 * e : 2 * e;                  //       1, 3            0, 2
      f32x4 bv[3];
#pragma unroll
      for (int d = 0; d < 3; ++d)      // column shifts ox = j - 1, j, j + 1
        bv[d] = *reinterpret_cast<const f32x4*>(si + prow * 2 * D4_RS + (d - 1) * 4);
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int d = px == 0 ? 1 - f : 2 - f;        // px 0: ox = j (kx 1), j - 1 (kx 3); px 1: j + 1 (kx 0), j (kx 2)
          const int kx = px == 0 ? 1 + 2 * f : 2 * f;
          const float* wt = sw + (size_t)((ky * 4 + kx) * 2) * 256;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(wt);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(wt + 256);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            acc[px][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[ks], bv[d][ks], acc[px][0], 0, 0, 0);
            acc[px][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ks], bv[d][ks], acc[px][1], 0, 0, 0);
          }
        }
    }
    if (ch + 1 < a.nchunk) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: even / odd pixels of a lane are neighbours -> 8-byte stores ----------------------------------
  // Buffer instructions with the channel offset in an SGPR: no 64-bit address per (channel, lane).  act'(.): ALL 32
  // activation reads first, then the stores -- interleaved, every read waits for the store before it (dx and act_y may
  // alias as far as the compiler knows): 455 -> 635 us on the 24 x 64 x 256 x 256 block.
  const int ix = 64 * tx + 2 * ln;
  const unsigned hw4 = (unsigned)(a.h * a.w) * 4u;
  const bool live = iy < a.h && ix < a.w;
  const unsigned vo = live ? (unsigned)(iy * a.w + ix) * 4u + (unsigned)(4 * lk) * hw4 : C4_OOB;
  const unsigned recs = 64u * hw4;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef float c4f2 __attribute__((ext_vector_type(2)));
  c4f2 yv[2][16];
  if constexpr (ACT) {
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.act_y + (long long)n * a.act_ns + (size_t)(64 * cig) * a.h * a.w), 0, recs, 0x00020000);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        yv[t][r] = __builtin_bit_cast(c4f2, __builtin_amdgcn_raw_buffer_load_b64(
            ra, (int)vo, (int)((unsigned)(32 * t + 8 * (r >> 2) + (r & 3)) * hw4), 0));
  }
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      a.dx + (long long)n * a.dx_ns + (size_t)(64 * cig) * a.h * a.w, 0, recs, 0x00020000);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v0 = acc[0][t][r], v1 = acc[1][t][r];
      if constexpr (ACT) {
        v0 = yv[t][r].x > 0.f ? v0 : v0 * a.slope;
        v1 = yv[t][r].y > 0.f ? v1 : v1 * a.slope;
      }
      const u32x2 d = {__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)};
      __builtin_amdgcn_raw_buffer_store_b64(d, rd, (int)vo, (int)((unsigned)(32 * t + 8 * (r >> 2) + (r & 3)) * hw4), 0);
    }
}

constexpr int d4_lds_bytes() { return (2 * D4_IN + 2 * D4_WCH) * 4; }


// ------------------------------------------------------------------------------------------------------------
// small maps (output 16 or 8 pixels wide: the critic's deeper blocks, 12-24 clips): a 32-pixel MFMA column block is
// 32 / OW output rows; the workgroup's 4 waves = 2 pixel blocks x 2 channel blocks (one 32 x 32 accumulator each);
// the input channels are split over `ksplit` workgroups that write partial sums, added in a fixed order by
// conv4_splitk_sum_kernel -- 48-192 tiles alone would leave most CUs idle (the embedded split-K form ran these layers
// on 32-pixel-WIDE tiles: 50-75 % of every tile outside the map, 54-75 us per launch).
// ------------------------------------------------------------------------------------------------------------
struct Conv4SmallArgs {
  const float* x;
  const float* wpk;
  float* y;              // ksplit == 1: the output; else partials [ksplit][n][co][oh][ow]
  long long x_ns, y_ns, part_ss;
  int ci, co, h, w;
  int tiles_y, nocg, nchunk, ksplit;
};

template <int OW>
__global__ __launch_bounds__(256) void conv4x4s2_small_kernel(Conv4SmallArgs a) {
  constexpr int F = 32 / OW;                  // output rows of a wave's pixel block
  constexpr int TRW = 2 * F;                  // output rows of the workgroup
  constexpr int PR = 2 * TRW + 2, PC = 2 * OW + 2;
  constexpr int CP = OW + 1;
  constexpr int CPS = 2 * CP + 2;
  constexpr int IN_FLOATS = PR * 2 * 2 * CPS;
  constexpr int ITEMS = PR * PC * 2;
  constexpr int I_PER_T = (ITEMS + 255) / 256;
  constexpr int W_PER_T = C4_WCH / 4 / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + 2 * IN_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int ocg = b % a.nocg; b /= a.nocg;
  const int ks = b % a.ksplit;
  const int n = b / a.ksplit;
  const int oh = a.h >> 1;
  const int oy0 = ty * TRW;
  const int hw = a.h * a.w;
  const int ch_begin = (int)((long long)ks * a.nchunk / a.ksplit), ch_end = (int)((long long)(ks + 1) * a.nchunk / a.ksplit);

  unsigned voff[I_PER_T];
  int lds_item[I_PER_T];
#pragma unroll
  for (int i = 0; i < I_PER_T; ++i) {
    const int q = tid + i * 256;
    const int k = q / (PR * PC), rem = q - k * (PR * PC);
    const int r = rem / PC, c = rem - r * PC;
    const int gy = 2 * oy0 - 1 + r, gx = c - 1;
    const bool ok = q < ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    voff[i] = ok ? (unsigned)((k * hw + gy * a.w + gx) * 4) : C4_OOB;
    lds_item[i] = q < ITEMS ? ((r * 2 + k) * 2 + (c & 1)) * CPS + (c >> 1) * 2 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, (unsigned)(a.ci * hw * 4), 0x00020000);
  const unsigned plane = (unsigned)hw * 4u;
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk + (size_t)ocg * a.nchunk * C4_WCH);

  float2 rin[I_PER_T];
  f32x4 rw[W_PER_T];
  auto load_chunk = [&](int ch) {
    const unsigned cbase = (unsigned)(ch * C4_CK) * plane;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i) {
      rin[i].x = c4_load(rs, voff[i] + cbase);
      rin[i].y = c4_load(rs, voff[i] + cbase + 2u * plane);
    }
    const f32x4* ws = wsrc + (size_t)ch * (C4_WCH / 4);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) rw[i] = ws[tid + i * 256];
  };
  auto store_chunk = [&](int buf) {
    float* si = s_in + buf * IN_FLOATS;
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i)
      if (lds_item[i] >= 0) *reinterpret_cast<float2*>(si + lds_item[i]) = rin[i];
    f32x4* sw = reinterpret_cast<f32x4*>(s_w + buf * C4_WCH);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) sw[tid + i * 256] = rw[i];
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int lk = lane >> 5, ln = lane & 31;
  const int pt = wave & 1, ocb = wave >> 1;
  const int lr = ln / OW, lc = ln - lr * OW;
  const int wr = pt * F + lr;                          // output row of the lane inside the tile
  const int b_off = ((2 * wr) * 2 + lk) * 2 * CPS + lc * 2;
  const int a_off = ocb * 128 + lane * 2;

  load_chunk(ch_begin);
  store_chunk(0);
  __syncthreads();
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int buf = (ch - ch_begin) & 1;
    if (ch + 1 < ch_end) load_chunk(ch + 1);
    const float* si = s_in + buf * IN_FLOATS + b_off;
    const float* sw = s_w + buf * C4_WCH + a_off;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float2 bv = *reinterpret_cast<const float2*>(si + (ky * 2 * 2 + (kx & 1)) * CPS + (kx >> 1) * 2);
        const float2 av = *reinterpret_cast<const float2*>(sw + ((ky * 4 + kx) * 2) * 128);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      }
    if (ch + 1 < ch_end) store_chunk(buf ^ 1);
    __syncthreads();
  }

  const int oy = oy0 + wr;
  if (oy < oh) {
    float* yo = a.y + (long long)ks * a.part_ss + (long long)n * a.y_ns + (size_t)(64 * ocg + 32 * ocb) * oh * OW +
                (size_t)oy * OW + lc;
#pragma unroll
    for (int r = 0; r < 16; ++r) yo[(size_t)(8 * (r >> 2) + 4 * lk + (r & 3)) * oh * OW] = acc[r];
  }
}

template <int OW> static constexpr int c4s_lds_bytes() {
  return (2 * ((2 * 2 * (32 / OW) + 2) * 2 * 2 * (2 * (OW + 1) + 2)) + 2 * C4_WCH) * 4;
}

// y[i] = sum_s part[s][i] (fixed order: bit-reproducible), optionally times act'(act_y[i])
__global__ void conv4_splitk_sum_kernel(const float* part, float* y, int S, long long total4,      // S == 1: y may be part
                                        const float* __restrict__ act_y, float slope) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
    for (int s = 1; s < S; ++s) v += reinterpret_cast<const f32x4*>(part)[(long long)s * total4 + i];
    if (act_y) {
      const f32x4 ya = reinterpret_cast<const f32x4*>(act_y)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ya[e] > 0.f ? v[e] : v[e] * slope;
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

// Data gradient on small maps (dx 32 or 16 pixels wide): a wave owns 64 / W dx rows of ONE parity (iy, iy + 2, ...):
// lanes = (row f, column pair j); the workgroup's 4 waves = 2 parities x 2 row groups = 4 * 64 / W consecutive rows.
struct Conv4DgradSmallArgs {
  const float* g;
  const float* wpk;
  float* dx;             // ksplit == 1: the gradient; else partials [ksplit][n][ci][h][w]
  long long g_ns, dx_ns, part_ss;
  int ci, co, h, w;
  int tiles_y, ncig, nchunk, ksplit;
};

template <int W>
__global__ __launch_bounds__(256) void conv4x4s2_dgrad_small_kernel(Conv4DgradSmallArgs a) {
  constexpr int F = 64 / W;                 // dx rows per wave
  constexpr int HC = W / 2;                 // column pairs per row
  constexpr int PRW = 2 * F + 2;            // patch rows of g
  constexpr int PCW = HC + 2;               // patch columns
  constexpr int RS = PCW * 4 + 4;           // floats per (row, k)
  constexpr int IN = PRW * 2 * RS;
  constexpr int ITEMS = PRW * PCW * 2;
  static_assert(ITEMS <= 256, "one staging item per thread");
  constexpr int W_PER_T = D4_WCH / 4 / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + 2 * IN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int cig = b % a.ncig; b /= a.ncig;
  const int ks = b % a.ksplit;
  const int n = b / a.ksplit;
  const int oh = a.h >> 1, ow = a.w >> 1, ohw = oh * ow;
  const int base = ty * 4 * F;              // first dx row of the workgroup
  const int oyb = base / 2 - 1;
  const int ch_begin = (int)((long long)ks * a.nchunk / a.ksplit), ch_end = (int)((long long)(ks + 1) * a.nchunk / a.ksplit);

  unsigned voff; int lds_item;
  {
    const int q = tid;
    const int k = q / (PRW * PCW), rem = q - k * (PRW * PCW);
    const int r = rem / PCW, c = rem - r * PCW;
    const int gy = oyb + r, gx = c - 1;
    const bool ok = q < ITEMS && gy >= 0 && gy < oh && gx >= 0 && gx < ow;
    voff = ok ? (unsigned)((k * ohw + gy * ow + gx) * 4) : C4_OOB;
    lds_item = q < ITEMS ? (r * 2 + k) * RS + c * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.g + (long long)n * a.g_ns), 0, (unsigned)(a.co * ohw * 4), 0x00020000);
  const unsigned plane = (unsigned)ohw * 4u;
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk + (size_t)cig * a.nchunk * D4_WCH);

  f32x4 rin;
  f32x4 rw[W_PER_T];
  auto load_chunk = [&](int ch) {
    const unsigned cbase = (unsigned)(ch * D4_CK) * plane;
#pragma unroll
    for (int j = 0; j < 4; ++j) rin[j] = c4_load(rs, voff + cbase + (unsigned)(2 * j) * plane);
    const f32x4* ws = wsrc + (size_t)ch * (D4_WCH / 4);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) rw[i] = ws[tid + i * 256];
  };
  auto store_chunk = [&](int buf) {
    if (lds_item >= 0) *reinterpret_cast<f32x4*>(s_in + buf * IN + lds_item) = rin;
    f32x4* sw = reinterpret_cast<f32x4*>(s_w + buf * D4_WCH);
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) sw[tid + i * 256] = rw[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][t][r] = 0.f;

  const int lk = lane >> 5, ln = lane & 31;
  const int py = wave & 1, gq = wave >> 1;
  const int f = ln / HC, jl = ln - f * HC;
  const int iy = base + 2 * F * gq + py + 2 * f;
  const int prow_m = F * gq + f + 1;        // patch row of oy = iy >> 1
  const int b_lane = lk * RS + (jl + 1) * 4;
  const int a_off = lane * 4;

  load_chunk(ch_begin);
  store_chunk(0);
  __syncthreads();
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int buf = (ch - ch_begin) & 1;
    if (ch + 1 < ch_end) load_chunk(ch + 1);
    const float* si = s_in + buf * IN + b_lane;
    const float* sw = s_w + buf * D4_WCH + a_off;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int prow = py == 0 ? prow_m - e : prow_m + 1 - e;
      const int ky = py == 0 ? 1 + 2 * e : 2 * e;
      f32x4 bv[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv[d] = *reinterpret_cast<const f32x4*>(si + prow * 2 * RS + (d - 1) * 4);
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int ff = 0; ff < 2; ++ff) {
          const int d = px == 0 ? 1 - ff : 2 - ff;
          const int kx = px == 0 ? 1 + 2 * ff : 2 * ff;
          const float* wt = sw + (size_t)((ky * 4 + kx) * 2) * 256;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(wt);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(wt + 256);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            acc[px][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s4], bv[d][s4], acc[px][0], 0, 0, 0);
            acc[px][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s4], bv[d][s4], acc[px][1], 0, 0, 0);
          }
        }
    }
    if (ch + 1 < ch_end) store_chunk(buf ^ 1);
    __syncthreads();
  }

  const unsigned hw4 = (unsigned)(a.h * a.w) * 4u;
  const bool live = iy < a.h;
  const unsigned vo = live ? (unsigned)(iy * a.w + 2 * jl) * 4u + (unsigned)(4 * lk) * hw4 : C4_OOB;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      a.dx + (long long)ks * a.part_ss + (long long)n * a.dx_ns + (size_t)(64 * cig) * a.h * a.w, 0, 64u * hw4, 0x00020000);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v0 = acc[0][t][r], v1 = acc[1][t][r];   // (scalars first: __builtin_bit_cast of a vector ELEMENT expression reads element 0)
      const u32x2 d = {__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)};
      __builtin_amdgcn_raw_buffer_store_b64(d, rd, (int)vo, (int)((unsigned)(32 * t + 8 * (r >> 2) + (r & 3)) * hw4), 0);
    }
}

template <int W> static constexpr int d4s_lds_bytes() {
  return (2 * ((2 * (64 / W) + 2) * 2 * ((W / 2 + 2) * 4 + 4)) + 2 * D4_WCH) * 4;
}

static bool c4_attr_done = false;
static void c4_set_attrs() {
  if (c4_attr_done) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      c4_lds_bytes<1>());
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      c4_lds_bytes<2>());
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_dgrad_mfma_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, d4_lds_bytes());
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_dgrad_mfma_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, d4_lds_bytes());
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_dgrad_small_kernel<32>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, d4s_lds_bytes<32>());
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_dgrad_small_kernel<16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, d4s_lds_bytes<16>());
  c4_attr_done = true;
}

}  // namespace tg

using namespace tg;

static bool c4_small(int h, int w) {      // the folded-row forms: whole workgroup tiles (4 or 8 output rows / 8 or 16 dx rows)
  return (w == 32 && h % 8 == 0) || (w == 16 && h % 16 == 0);
}

extern "C" int tg_conv4x4s2_supported(int n, int ci, int co, int h, int w) {
  // whole 32-pixel output row segments (or the small-map forms), 64-channel blocks both ways, tensors below 2 GiB per
  // item (32-bit buffer offsets)
  return n > 0 && ci % 64 == 0 && co % 64 == 0 && h >= 2 && (h & 1) == 0 && (w % 64 == 0 || c4_small(h, w)) &&
         (long long)ci * h * w * 4 < (1ll << 31) && (long long)co * (h / 2) * (w / 2) * 4 < (1ll << 31);
}

// Split factor of the small-map forms (1 for the large ones) and the floats of partial sums a call needs
// (0: none): the caller passes a workspace of that size.
static int c4_pick_ksplit(int base_wgs, int nchunk) {
  if (const int e = TG_LAB_ENV("TG_C4_KSPLIT", 0)) return nchunk % e == 0 ? e : 1;   // lab builds only
  int ks = 1;
  while (base_wgs * ks < 320 && ks * 2 <= nchunk / 2 && nchunk % (ks * 2) == 0) ks *= 2;
  return ks;
}
static int c4_fwd_ksplit(int n, int ci, int co, int h, int w) {
  if (w % 64 == 0) return 1;
  const int trw = w == 32 ? 4 : 8;
  return c4_pick_ksplit((h / 2 / trw) * (co / 64) * n, ci / C4_CK);
}
static int c4_dgrad_ksplit(int n, int ci, int co, int h, int w) {
  if (w % 64 == 0) return 1;
  const int rows = w == 32 ? 8 : 16;
  return c4_pick_ksplit((h / rows) * (ci / 64) * n, co / D4_CK);
}
extern "C" size_t tg_conv4x4s2_workspace_floats(int n, int ci, int co, int h, int w, int dgrad) {
  if (!tg_conv4x4s2_supported(n, ci, co, h, w)) return 0;
  const int ks = dgrad ? c4_dgrad_ksplit(n, ci, co, h, w) : c4_fwd_ksplit(n, ci, co, h, w);
  if (ks == 1) return 0;
  return (size_t)ks * n * (dgrad ? (size_t)ci * h * w : (size_t)co * (h / 2) * (w / 2));
}

extern "C" size_t tg_conv4x4s2_packed_floats(int ci, int co) { return (size_t)ci * co * 16; }

extern "C" int tg_conv4x4s2_pack(const float* w, float* w_fwd, float* w_dgrad, int ci, int co, tg_stream_t stream) {
  TG_REQUIRE(w && (w_fwd || w_dgrad), TG_E_ARG, "conv4x4s2_pack: null pointer");
  TG_REQUIRE(ci % 64 == 0 && co % 64 == 0 && ci > 0 && co > 0, TG_E_SHAPE, "conv4x4s2_pack: ci=%d co=%d (multiples of 64)", ci, co);
  const long long total = (long long)ci * co * 16;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (w_fwd) hipLaunchKernelGGL(conv4_pack_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, w_fwd, ci, co);
  if (w_dgrad) hipLaunchKernelGGL(conv4_pack_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, w_dgrad, ci, co);
  return check_launch("conv4x4s2_pack");
}

extern "C" int tg_conv4x4s2_fwd(const float* x, const float* w_fwd, float* y, float* workspace, int n, int ci, int co,
                                int h, int w, tg_stream_t stream) {
  TG_REQUIRE(x && w_fwd && y, TG_E_ARG, "conv4x4s2_fwd: null pointer");
  TG_REQUIRE(tg_conv4x4s2_supported(n, ci, co, h, w), TG_E_SHAPE, "conv4x4s2_fwd: unsupported n=%d ci=%d co=%d h=%d w=%d", n, ci,
             co, h, w);
  c4_set_attrs();
  if (w % 64 != 0) {
    Conv4SmallArgs sa;
    const int ks = c4_fwd_ksplit(n, ci, co, h, w);
    TG_REQUIRE(ks == 1 || workspace, TG_E_ARG, "conv4x4s2_fwd: this shape needs tg_conv4x4s2_workspace_floats of workspace");
    const long long out_floats = (long long)n * co * (h / 2) * (w / 2);
    sa.x = x; sa.wpk = w_fwd; sa.y = ks == 1 ? y : workspace;
    sa.x_ns = (long long)ci * h * w; sa.y_ns = (long long)co * (h / 2) * (w / 2); sa.part_ss = out_floats;
    sa.ci = ci; sa.co = co; sa.h = h; sa.w = w;
    sa.nocg = co / 64; sa.nchunk = ci / C4_CK; sa.ksplit = ks;
    if (w == 32) {
      sa.tiles_y = h / 2 / 4;
      hipLaunchKernelGGL(conv4x4s2_small_kernel<16>, dim3((unsigned)(sa.tiles_y * sa.nocg * ks * n)), dim3(256),
                         c4s_lds_bytes<16>(), (hipStream_t)stream, sa);
    } else {
      sa.tiles_y = h / 2 / 8;
      hipLaunchKernelGGL(conv4x4s2_small_kernel<8>, dim3((unsigned)(sa.tiles_y * sa.nocg * ks * n)), dim3(256),
                         c4s_lds_bytes<8>(), (hipStream_t)stream, sa);
    }
    if (ks > 1) {
      const long long t4 = out_floats / 4;
      hipLaunchKernelGGL(conv4_splitk_sum_kernel, dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const float*)workspace, y, ks, t4, (const float*)nullptr, 0.f);
    }
    return check_launch("conv4x4s2_fwd (small maps)");
  }
  Conv4Args a;
  a.x = x; a.wpk = w_fwd; a.y = y;
  a.x_ns = (long long)ci * h * w; a.y_ns = (long long)co * (h / 2) * (w / 2);
  a.ci = ci; a.co = co; a.h = h; a.w = w;
  a.nocg = co / 64; a.nchunk = ci / C4_CK;
  const int oh = h / 2, ow = w / 2;
  if (ow % 64 == 0) {
    a.tiles_x = ow / 64; a.tiles_y = (oh + 1) / 2;
    hipLaunchKernelGGL(conv4x4s2_mfma_kernel<2>, dim3((unsigned)(a.tiles_x * a.tiles_y * a.nocg * n)), dim3(256),
                       c4_lds_bytes<2>(), (hipStream_t)stream, a);
  } else {
    a.tiles_x = ow / 32; a.tiles_y = (oh + 3) / 4;
    hipLaunchKernelGGL(conv4x4s2_mfma_kernel<1>, dim3((unsigned)(a.tiles_x * a.tiles_y * a.nocg * n)), dim3(256),
                       c4_lds_bytes<1>(), (hipStream_t)stream, a);
  }
  return check_launch("conv4x4s2_fwd");
}

extern "C" int tg_conv4x4s2_dgrad(const float* g, const float* w_dgrad, const float* act_y, int act, float* dx,
                                  float* workspace, int n, int ci, int co, int h, int w, tg_stream_t stream) {
  TG_REQUIRE(g && w_dgrad && dx, TG_E_ARG, "conv4x4s2_dgrad: null pointer");
  TG_REQUIRE(tg_conv4x4s2_supported(n, ci, co, h, w), TG_E_SHAPE, "conv4x4s2_dgrad: unsupported n=%d ci=%d co=%d h=%d w=%d", n,
             ci, co, h, w);
  TG_REQUIRE(!act_y || act == TG_ACT_RELU || act == TG_ACT_LRELU02, TG_E_ARG, "conv4x4s2_dgrad: act=%d (relu | lrelu)", act);
  c4_set_attrs();
  if (w % 64 != 0) {
    Conv4DgradSmallArgs sa;
    const int ks = c4_dgrad_ksplit(n, ci, co, h, w);
    const bool two_pass = ks > 1 || act_y;      // act'(.) rides on the summing pass
    TG_REQUIRE(ks == 1 || workspace, TG_E_ARG, "conv4x4s2_dgrad: this shape needs tg_conv4x4s2_workspace_floats of workspace");
    const long long out_floats = (long long)n * ci * h * w;
    sa.g = g; sa.wpk = w_dgrad; sa.dx = ks == 1 ? dx : workspace;
    sa.g_ns = (long long)co * (h / 2) * (w / 2); sa.dx_ns = (long long)ci * h * w; sa.part_ss = out_floats;
    sa.ci = ci; sa.co = co; sa.h = h; sa.w = w;
    sa.ncig = ci / 64; sa.nchunk = co / D4_CK; sa.ksplit = ks;
    if (w == 32) {
      sa.tiles_y = h / 8;
      hipLaunchKernelGGL(conv4x4s2_dgrad_small_kernel<32>, dim3((unsigned)(sa.tiles_y * sa.ncig * ks * n)), dim3(256),
                         d4s_lds_bytes<32>(), (hipStream_t)stream, sa);
    } else {
      sa.tiles_y = h / 16;
      hipLaunchKernelGGL(conv4x4s2_dgrad_small_kernel<16>, dim3((unsigned)(sa.tiles_y * sa.ncig * ks * n)), dim3(256),
                         d4s_lds_bytes<16>(), (hipStream_t)stream, sa);
    }
    if (two_pass) {
      const long long t4 = out_floats / 4;
      hipLaunchKernelGGL(conv4_splitk_sum_kernel, dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const float*)(ks == 1 ? dx : workspace), dx, ks, t4, act_y, act == TG_ACT_LRELU02 ? 0.2f : 0.f);
    }
    return check_launch("conv4x4s2_dgrad (small maps)");
  }
  Conv4DgradArgs a;
  a.g = g; a.wpk = w_dgrad; a.act_y = act_y; a.dx = dx;
  a.g_ns = (long long)co * (h / 2) * (w / 2); a.dx_ns = (long long)ci * h * w; a.act_ns = a.dx_ns;
  a.slope = act == TG_ACT_LRELU02 ? 0.2f : 0.f;
  a.ci = ci; a.co = co; a.h = h; a.w = w;
  a.tiles_x = w / 64; a.tiles_y = (h + 3) / 4; a.ncig = ci / 64; a.nchunk = co / D4_CK;
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.ncig * n));
  if (act_y)
    hipLaunchKernelGGL(conv4x4s2_dgrad_mfma_kernel<true>, grid, dim3(256), d4_lds_bytes(), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(conv4x4s2_dgrad_mfma_kernel<false>, grid, dim3(256), d4_lds_bytes(), (hipStream_t)stream, a);
  return check_launch("conv4x4s2_dgrad");
}

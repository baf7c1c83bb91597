// Weight gradient of a 3x3 / stride 1 / pad 1 convolution on fp32 MFMA (gfx950):
//
//   G[a][b][tap] = sum_{n,y,x}  P[n][a][y][x] * Q[n][b][y+ky-1][x+kx-1]      (zero padded)
//
// With P = dZ (gradient w.r.t. the conv's pre-activation, `a` = cout) and Q = X
// (the conv's input, `b` = cin) this is dW of nn.Conv2d(cin,cout,3,1,1)
// (tecogan_nets.py:23-65,92-98,111-113).  The transposed convs and the
// discriminator's 4x4/s2 convs reach the same kernel through their
// space-to-depth embedding (see models/train_ops.py).
//
// GEMM view: M = a (32 per wave), N = b (32 per wave), K = pixels.  A lane
// holds channel (lane&31); the K order is permuted as in the forward kernel so
// a lane's four k-steps are four CONSECUTIVE pixels of a row: the A operand is
// one ds_read_b128 from a planar [channel][pixel] LDS tile -- NCHW rows are
// staged as they are.  The 9 taps of a pixel group share 18 shifted B values
// (3 rows x 6 columns), so per 36 MFMAs a wave issues 1 b128 + 18 b32 LDS reads.
// Each wave keeps 9 tap accumulators (32x32 each); a workgroup = 2x2 waves
// covers a 64x64 channel block.  Workgroups stride over the pixel tiles
// (K split) and write raw partial sums; wgrad_reduce_kernel adds the splits in
// a fixed order (deterministic) and accumulates into the gradient tensor.
#include "tg_common.h"

namespace tg {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int WG_R = 2;                 // rows per pixel tile
constexpr int WG_TW = 32;               // cols per pixel tile
constexpr int WG_CSA = WG_R * WG_TW + 4;         // 68: A channel stride (floats), 17 slots (odd)
constexpr int WG_RSB = WG_TW + 3;                // 35: B row stride
constexpr int WG_CSB = (WG_R + 2) * WG_RSB + 1;  // 141: B channel stride (odd -> conflict free)
constexpr int WG_A_FLOATS = 64 * WG_CSA;
constexpr int WG_B_FLOATS = 64 * WG_CSB;
constexpr unsigned WG_OOB = 0x80000000u;

constexpr int WG_MAXSEG = 64;
struct WgradArgs {
  // The batch may be split over up to WG_MAXSEG separately allocated segments of n_per_seg
  // images each (the unrolled frames of a recurrent layer): image i lives in segment
  // i / n_per_seg.  pseg = unshifted operand (dZ) (n_per_seg, ca, h, w); qseg = shifted (X).
  const float* pseg[WG_MAXSEG];
  const float* qseg[WG_MAXSEG];
  int n_per_seg;
  float* part;      // [nsplit][ca][cb_total][9] raw partial sums
  long long p_ns, q_ns;
  int ca, cb, cb_total, cb_off;   // G is written at columns [cb_off, cb_off+cb) of a cb_total-wide matrix
  int n, h, w;
  int tiles_x, tiles_y, ntiles, nsplit, nab, nbb;
  // (ablation builds, -DWG_ABL=bits: 1 no global loads after the first tile, 2 no LDS stores after the
  //  first, 4 no MFMAs, 8 no LDS operand reads)
  // phase-restricted taps (see tg_conv3x3_mfma.hip): the b channels come in 4 sub-pixel phases
  // of cphase channels; phase coordinate v along an axis uses tap set rowsets[v].  Taps outside
  // the set are not computed (their gradient entries are written as 0; the embedding drops them).
  int cphase;
  unsigned char rowsets[2];
  // Layered mode (nlayer > 0; tg_wgrad3x3_body): ONE launch computes the gradients of nlayer
  // consecutive equally shaped layers whose operands lie in per-frame blocks of tensors `lstride`
  // floats apart (the chained SRNet body keeps a frame's activations / gradients that way):
  //   layer L = 1..nlayer:  P = pseg[seg] + L * lstride,  Q = qseg[seg] + (L - 1) * lstride
  // blockIdx also enumerates the layers; partial sums of layer L start at part + (L - 1) * nsplit * ca * cb_total * 9.
  // (Kept to two scalars: a larger argument struct is no longer promoted out of private memory by
  //  the compiler and every access turns into a scratch load -- measured 4x slower.)
  int nlayer;
  long long lstride;
  // K split inside the workgroup (un-phased 2 x 32 launches): with ca <= 32 (bit 0) / cb <= 32 (bit 1) the
  // wave pairs that would multiply zero padding take alternate groups of 8 pixels instead and write
  // their sums as extra splits: partial index = split * KS + kid, KS = 2 or 4.
  int kmode;
  // stride-2 form (dW of a transposed conv) with the vector staging: the sums of dZ over the pixels -- the
  // layer's bias gradient -- are taken from the staged Q values on their way to LDS (every HR pixel belongs
  // to exactly one tile's "own" rows / columns): bias_part[split][cb], added by the tail blocks of wgrad_reduce_kernel.
  float* bias_part;
};

enum { WTAPS_ALL = 0, WTAPS_01 = 1, WTAPS_12 = 2, WTAPS_1 = 3 };
__host__ __device__ constexpr bool wtap_on(int code, int k) {
  return code == WTAPS_ALL || (code == WTAPS_01 && k <= 1) || (code == WTAPS_12 && k >= 1) ||
         (code == WTAPS_1 && k == 1);
}

// Tile geometry of one workgroup step: R rows x TW columns of P pixels (K of the GEMM) and the Q patch
// they meet.  <2, 32>: the general form.  <4, 16> / <8, 8>: the same 64 pixels folded for narrow maps
// (FNet's 16 x 16 and 8 x 8 levels waste 50-75 % of a 32-column tile).  S2: Q is sampled with stride 2,
//   G[a][b][ky][kx] = sum P[n][a][y][x] * Q[n][b][2y - 1 + ky][2x - 1 + kx]
// -- the weight gradient of ConvTranspose2d(k3, s2, p1, op1) (tecogan_nets.py:119-126; P = its input,
// Q = dZ at twice the resolution) taken straight from dZ: one row of 32 pixels meets a 3 x 65 patch.
template <int R_, int TW_, int S2_>
struct WgGeo {
  static constexpr int R = R_, TW = TW_, S2 = S2_;
  static constexpr int PIX = R * TW;
  static constexpr int CSA = PIX + 4;                       // A channel stride (floats)
  static constexpr int BR = S2 ? 2 * R + 1 : R + 2;         // Q patch rows / columns
  static constexpr int BC = S2 ? 2 * TW + 1 : TW + 2;
  static constexpr int RSB = S2 ? BC + 2 : TW + 3;          // B row stride
  static constexpr int CSB = (BR * RSB) | 1;                // odd channel stride -> conflict free
  static constexpr int A_FLOATS = 64 * CSA, B_FLOATS = 64 * CSB;
  static constexpr size_t LDS_BYTES = 2 * (size_t)(A_FLOATS + B_FLOATS) * sizeof(float);
};
using WgGeoStd = WgGeo<WG_R, WG_TW, 0>;
static_assert(WgGeoStd::CSA == WG_CSA && WgGeoStd::RSB == WG_RSB && WgGeoStd::CSB == WG_CSB, "geometry");

template <int RY, int RX, class G = WgGeoStd, bool VEC = false, int KS = 1>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                          // [2][A_FLOATS]
  float* sB = smem + 2 * G::A_FLOATS;        // [2][B_FLOATS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cot = wave & 1, cit = wave >> 1;
  int kid = 0;
  if (KS > 1) {                              // (wave-uniform)
    if (a.kmode & 1) { kid = cot; cot = 0; }
    if (a.kmode & 2) { kid += cit * ((a.kmode & 1) ? 2 : 1); cit = 0; }
  }
  int b = blockIdx.x;
  const int split = b % a.nsplit; b /= a.nsplit;
  const int bb = b % a.nbb; b /= a.nbb;
  const int ab = b % a.nab;
  const int li = b / a.nab;                  // layer index (0 outside the layered mode)
  const int a0 = ab * 64, b0 = bb * 64;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;
  const int hq = G::S2 ? 2 * a.h : a.h, wq = G::S2 ? 2 * a.w : a.w;
  const unsigned planeq = (unsigned)(hq * wq) * 4u;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int lh = lane >> 5, ll = lane & 31;
  const int a_rd = (cot * 32 + ll) * G::CSA + 4 * lh + 8 * kid;
  const int b_rd = (cit * 32 + ll) * G::CSB + (G::S2 ? 8 : 4) * lh + (G::S2 ? 16 : 8) * kid;

  // ---- staging, element-wise form (any width / alignment) ------------------------------------
  constexpr int A_PER_T = (64 * G::PIX) / 256;                       // 16 (8 for the stride-2 form)
  constexpr int B_ELEMS = 64 * G::BR * G::BC;                        // 8704 for <2, 32>
  constexpr int B_PER_T = (B_ELEMS + 255) / 256;                     // 34
  constexpr bool B_EXACT = B_ELEMS % 256 == 0;
  // ---- staging, vector form (w % 4 == 0, 16-byte aligned planes): the rows of a tile are read as
  // aligned 16-byte groups (8-byte pairs in the stride-2 form, whose patch starts at an even
  // column), position (row, group) is a per-THREAD constant and the loop walks channels -- one add
  // per load and immediate LDS offsets instead of ~20 index instructions per element (the staging
  // skeleton was ~19 % of the launch), and 3.4x fewer memory instructions.
  //   A: PIX / 4 groups per channel, thread -> group tid % (PIX/4), channels tid / (PIX/4) + CSTEP * i
  //   B: KG = (TW + 8) / 4 groups per row from column x0 - 4, padded to a power of two KGP; thread ->
  //      (k = tid % KGP, r = tid / KGP % (64 / KGP)), its wave owns 16 channels
  //   B, stride 2: 33 pairs per row from column 2 x0 - 2; thread -> position tid % 128 of the 3 x 33,
  //      its half of the workgroup owns 32 channels
  constexpr int VW = G::S2 ? 2 : 4;                                  // floats per B load
  constexpr int AV_PER_T = G::PIX / 16, A_CSTEP = 1024 / G::PIX;     // 4 loads, channels 16 apart (2 / 32)
  constexpr int KG = G::S2 ? 33 : (G::TW + 8) / 4;
  constexpr int KGP = KG <= 4 ? 4 : (KG <= 8 ? 8 : 16);
  constexpr int BV_PER_T = G::S2 ? 32 : 16;
  static_assert(G::S2 || 64 / KGP >= G::BR, "patch rows must fit the 64 positions of a wave");
  constexpr int RA_N = VEC ? AV_PER_T * 4 : A_PER_T, RB_N = VEC ? BV_PER_T * VW : B_PER_T;
  float ra[RA_N], rb[RB_N];
  // per-thread constants of the vector form
  const int va_k = tid % (G::TW / 4), va_r = (tid / (G::TW / 4)) % G::R, va_c = tid / (G::PIX / 4);
  int vb_k, vb_r, vb_c;
  bool vb_on;
  if (G::S2) {
    const int pos = tid & 127;
    vb_r = pos / 33; vb_k = pos - vb_r * 33; vb_c = (tid >> 7) * 32; vb_on = pos < 99;
  } else {
    vb_k = tid & (KGP - 1); vb_r = (tid / KGP) & (64 / KGP - 1); vb_c = (tid >> 6) * 16;
    vb_on = vb_k < KG && vb_r < G::BR;
  }

  auto load_tile = [&](int tile) {
    int n = tile / (a.tiles_x * a.tiles_y);
    int rem = tile - n * (a.tiles_x * a.tiles_y);
    int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    int x0 = tx * G::TW, y0 = ty * G::R;
    const int seg = __builtin_amdgcn_readfirstlane(n / a.n_per_seg);
    const int ln = __builtin_amdgcn_readfirstlane(n - seg * a.n_per_seg);
    const float* pbase = a.pseg[seg];
    const float* qbase = a.qseg[seg];
    if (a.nlayer) {
      pbase += (long long)(li + 1) * a.lstride;
      qbase += (long long)li * a.lstride;
    }
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pbase + (long long)ln * a.p_ns), 0, a.ca * hw * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(qbase + (long long)ln * a.q_ns), 0, a.cb * hq * wq * 4, 0x00020000);
    if constexpr (VEC) {
      {
        const int gy = y0 + va_r, gx = x0 + 4 * va_k;
        const bool ok = gy < a.h && gx < a.w;                  // w % 4 == 0: a group is inside or outside
        const unsigned off0 = ok ? ((unsigned)(a0 + va_c) * plane + (unsigned)(gy * a.w + gx) * 4u) : WG_OOB;
#pragma unroll
        for (int i = 0; i < AV_PER_T; ++i) {                   // (channel tail: past num_records -> 0)
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              rp, (int)(off0 + (unsigned)(i * A_CSTEP) * plane), 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[4 * i + e] = v[e];
        }
      }
      {
        const int gy = (G::S2 ? 2 * y0 : y0) - 1 + vb_r;
        const int gx = G::S2 ? 2 * x0 - 2 + 2 * vb_k : x0 - 4 + 4 * vb_k;
        const bool ok = vb_on && gy >= 0 && gy < hq && gx >= 0 && gx < wq;
        const unsigned off0 = ok ? ((unsigned)(b0 + vb_c) * planeq + (unsigned)(gy * wq + gx) * 4u) : WG_OOB;
#pragma unroll
        for (int i = 0; i < BV_PER_T; ++i) {
          const unsigned off = off0 + (unsigned)i * planeq;
          if constexpr (G::S2) {
            const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rq, (int)off, 0, 0));
            rb[2 * i] = v[0]; rb[2 * i + 1] = v[1];
          } else {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, (int)off, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[4 * i + e] = v[e];
          }
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / G::PIX, rc = idx % G::PIX, r = rc / G::TW, col = rc % G::TW;
      int gy = y0 + r, gx = x0 + col;
      bool ok = gy < a.h && gx < a.w;      // channel tail handled by num_records
      unsigned off = ok ? ((unsigned)(a0 + c) * plane + (unsigned)(gy * a.w + gx) * 4u) : WG_OOB;
      ra[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / (G::BR * G::BC);
      int rem2 = idx - c * (G::BR * G::BC);
      int r = rem2 / G::BC, col = rem2 - r * G::BC;
      int gy = (G::S2 ? 2 * y0 : y0) - 1 + r, gx = (G::S2 ? 2 * x0 : x0) - 1 + col;
      bool ok = (B_EXACT || idx < B_ELEMS) && gy >= 0 && gy < hq && gx >= 0 && gx < wq;
      unsigned off = ok ? ((unsigned)(b0 + c) * planeq + (unsigned)(gy * wq + gx) * 4u) : WG_OOB;
      rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, (int)off, 0, 0));
    }
    }
  };
  // Bias gradients out of the staged values (vector form; a.bias_part != null):
  //   stride-2 form (dW of a transposed conv, db = sum of Q = dZ): rows 1, 2 of the 3-row patch and pairs
  //     1..32 of a row are this tile's own HR pixels;
  //   every other form (db = sum of P = dZ): the P tile has no halo, every staged value counts once
  //     (the workgroups of b-channel block 0 do it).
  constexpr bool BIAS = VEC && G::S2 != 0;
  constexpr bool BIAS_A = VEC && G::S2 == 0;
  constexpr int NBS = BIAS ? BV_PER_T : (BIAS_A ? AV_PER_T : 1);
  float bsum[NBS];
#pragma unroll
  for (int i = 0; i < NBS; ++i) bsum[i] = 0.f;
  const bool own = BIAS && a.bias_part && ab == 0 && vb_on && vb_r >= 1 && vb_k >= 1 && vb_k <= 32;
  const bool own_a = BIAS_A && a.bias_part && bb == 0;
  auto store_tile = [&](int buf) {
    float* pa = sA + buf * G::A_FLOATS;
    float* pb = sB + buf * G::B_FLOATS;
    if constexpr (BIAS) {
      if (own) {
#pragma unroll
        for (int i = 0; i < BV_PER_T; ++i) bsum[i] += rb[2 * i] + rb[2 * i + 1];
      }
    }
    if constexpr (BIAS_A) {
      if (own_a) {
#pragma unroll
        for (int i = 0; i < AV_PER_T; ++i) bsum[i] += (ra[4 * i] + ra[4 * i + 1]) + (ra[4 * i + 2] + ra[4 * i + 3]);
      }
    }
    if constexpr (VEC) {
      float* qa = pa + va_c * G::CSA + va_r * G::TW + 4 * va_k;               // 16-byte aligned (CSA % 4 == 0)
#pragma unroll
      for (int i = 0; i < AV_PER_T; ++i)
        *reinterpret_cast<f32x4*>(qa + i * A_CSTEP * G::CSA) = f32x4{ra[4 * i], ra[4 * i + 1], ra[4 * i + 2], ra[4 * i + 3]};
      // patch column of element e of the group: stride 1: 4k - 3 + e (columns x0 - 4 .. x0 - 2 are not part
      // of the patch); stride 2: 2k - 1 + e (column 2 x0 - 2 is not)
      const int col0 = G::S2 ? 2 * vb_k - 1 : 4 * vb_k - 3;
      float* qb = pb + vb_c * G::CSB + vb_r * G::RSB + col0;
      if (vb_on) {
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          if (col0 + e >= 0 && col0 + e < G::BC) {
#pragma unroll
            for (int i = 0; i < BV_PER_T; ++i) qb[i * G::CSB + e] = rb[VW * i + e];
          }
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / G::PIX, rc = idx % G::PIX;
      pa[c * G::CSA + rc] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / (G::BR * G::BC);
      int rem2 = idx - c * (G::BR * G::BC);
      int r = rem2 / G::BC, col = rem2 - r * G::BC;
      if (B_EXACT || idx < B_ELEMS) pb[c * G::CSB + r * G::RSB + col] = rb[i];
    }
    }
  };

  int tile = split;
  int it = 0;
  if (tile < a.ntiles) {
    load_tile(tile);
    store_tile(0);
  }
  __syncthreads();
#ifndef WG_ABL
#define WG_ABL 0      // compile-time ablation bits (tools/build_lab_libs.sh builds one library per value)
#endif
#define WGABL(bit) ((WG_ABL & (bit)) != 0)
#if !TG_LAB && WG_ABL
#error "tg_wgrad_mfma.hip: WG_ABL needs -DTG_LAB=1 (lab builds only; the ablated kernels compute wrong results)"
#endif
  constexpr int NBV = G::S2 ? 9 : 6;         // shifted Q values one (row, group of 4 pixels) needs per tap row
  constexpr int QS = G::S2 ? 2 : 1;          // Q step per P pixel
  for (; tile < a.ntiles; tile += a.nsplit, ++it) {
    const int buf = it & 1;
    const bool more = tile + a.nsplit < a.ntiles;
    if (more && !WGABL(1)) load_tile(tile + a.nsplit);
    const float* pa = sA + buf * G::A_FLOATS + a_rd;
    const float* pb = sB + buf * G::B_FLOATS + b_rd;
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
#pragma unroll
      for (int g = 0; g < G::TW / 8; g += KS) {     // (K split: this wave's groups are kid, kid + KS, ...)
        f32x4 av = {1.f, 2.f, 3.f, 4.f};
        float bv[3][NBV];
        if (!WGABL(8)) {
          av = *reinterpret_cast<const f32x4*>(pa + r * G::TW + 8 * g);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int c6 = 0; c6 < NBV; ++c6) bv[ky][c6] = pb[(QS * r + ky) * G::RSB + QS * 8 * g + c6];
        } else {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int c6 = 0; c6 < NBV; ++c6) bv[ky][c6] = (float)(ky + c6 + r + g);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if (wtap_on(RY, ky) && wtap_on(RX, kx)) {
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                if (WGABL(4)) acc[ky * 3 + kx][kk] += av[kk] * bv[ky][QS * kk + kx];
                else
                acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[ky][QS * kk + kx],
                                                                        acc[ky * 3 + kx], 0, 0, 0);
              }
            }
      }
    }
    if (more && !WGABL(2)) store_tile(buf ^ 1);
    __syncthreads();
  }

  // D[i = a-channel][j = b-channel]: lane holds j = ll, registers walk i
  float* out = a.part + (((long long)li * a.nsplit + split) * KS + kid) * a.ca * a.cb_total * 9;
  const int bj = b0 + cit * 32 + ll;
  if (bj < a.cb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int ai = a0 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ai < a.ca) {
        float* o = out + ((long long)ai * a.cb_total + a.cb_off + bj) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t] = acc[t][r];
      }
    }
  }
  if constexpr (BIAS) {
    if (a.bias_part && ab == 0) {          // (block-uniform; the loop's last barrier has passed: LDS is free)
      float* sb = smem;                    // [256 threads][33]
#pragma unroll
      for (int i = 0; i < BV_PER_T; ++i) sb[tid * 33 + i] = bsum[i];
      __syncthreads();
      if (tid < 64) {
        const int grp = tid >> 5, i = tid & 31;
        float v = 0.f;
        for (int j = 0; j < 128; ++j) v += sb[(grp * 128 + j) * 33 + i];      // fixed order
        if (b0 + tid < a.cb) a.bias_part[(long long)split * a.cb + b0 + tid] = v;
      }
    }
  }
  if constexpr (BIAS_A) {
    if (own_a) {                           // (block-uniform)
      float* sb = smem;                    // [256 threads][AV_PER_T + 1]
      __syncthreads();
#pragma unroll
      for (int i = 0; i < AV_PER_T; ++i) sb[tid * (AV_PER_T + 1) + i] = bsum[i];
      __syncthreads();
      if (tid < 64) {                      // channel tid = vc + A_CSTEP * i was staged by the PIX / 4 threads of vc
        const int vc = tid % A_CSTEP, i = tid / A_CSTEP;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < G::PIX / 4; ++j) v += sb[(vc * (G::PIX / 4) + j) * (AV_PER_T + 1) + i];     // fixed order
        if (a0 + tid < a.ca) a.bias_part[((long long)li * a.nsplit + split) * a.ca + a0 + tid] = v;
      }
    }
  }
}

// (the bias partials [split][c] of an un-layered launch are added by the tail blocks of wgrad_reduce_kernel)
// the same for every layer of a layered launch: blockIdx.y = layer; part: [layer][split][c]
struct WgradLayerBias { float* b[24]; };
__global__ __launch_bounds__(256) void bias_part_reduce_layers_kernel(const float* __restrict__ part, WgradLayerBias bl,
                                                                      int nsplit, int cb, int accumulate) {
  __shared__ float sm[4][64];
  const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cb) {
    const float* p = part + (long long)blockIdx.y * nsplit * cb + c;
    int k = sg;
    for (; k + 12 < nsplit; k += 16) {
      s0 += p[(long long)k * cb]; s1 += p[(long long)(k + 4) * cb];
      s2 += p[(long long)(k + 8) * cb]; s3 += p[(long long)(k + 12) * cb];
    }
    for (; k < nsplit; k += 4) s0 += p[(long long)k * cb];
  }
  sm[sg][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  float* db = bl.b[blockIdx.y];
  if (sg == 0 && c < cb) db[c] = (accumulate ? db[c] : 0.f) + ((sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]));
}

// the un-phased, un-layered launches in the other geometries
template <class G, bool VEC>
__global__ __launch_bounds__(256) void wgrad3x3_mfma_geo_kernel(WgradArgs a) {
  wgrad_body<WTAPS_ALL, WTAPS_ALL, G, VEC>(a);
}

template <bool VEC>
__device__ __forceinline__ void wgrad_std_dispatch(const WgradArgs& a) {
  if (a.cphase == 0) { wgrad_body<WTAPS_ALL, WTAPS_ALL, WgGeoStd, VEC>(a); return; }
  // block-uniform: the 64 b channels of this block belong to one sub-pixel phase
  const int bb = ((int)blockIdx.x / a.nsplit) % a.nbb;      // (phased launches are never layered)
  const int ph = (bb * 64) / a.cphase;
  const int ry = a.rowsets[(ph >> 1) & 1], rx = a.rowsets[ph & 1];
  switch (ry * 4 + rx) {
    case WTAPS_01 * 4 + WTAPS_01: wgrad_body<WTAPS_01, WTAPS_01, WgGeoStd, VEC>(a); break;
    case WTAPS_01 * 4 + WTAPS_12: wgrad_body<WTAPS_01, WTAPS_12, WgGeoStd, VEC>(a); break;
    case WTAPS_12 * 4 + WTAPS_01: wgrad_body<WTAPS_12, WTAPS_01, WgGeoStd, VEC>(a); break;
    case WTAPS_12 * 4 + WTAPS_12: wgrad_body<WTAPS_12, WTAPS_12, WgGeoStd, VEC>(a); break;
    case WTAPS_01 * 4 + WTAPS_1: wgrad_body<WTAPS_01, WTAPS_1, WgGeoStd, VEC>(a); break;
    case WTAPS_1 * 4 + WTAPS_01: wgrad_body<WTAPS_1, WTAPS_01, WgGeoStd, VEC>(a); break;
    case WTAPS_1 * 4 + WTAPS_1: wgrad_body<WTAPS_1, WTAPS_1, WgGeoStd, VEC>(a); break;
    default: wgrad_body<WTAPS_ALL, WTAPS_ALL, WgGeoStd, VEC>(a); break;
  }
}
// un-phased 2 x 32 launches with 2 or 4 waves sharing the pixels of a tile (WgradArgs::kmode)
template <bool VEC, int KS>
__global__ __launch_bounds__(256) void wgrad3x3_mfma_ks_kernel(WgradArgs a) {
  wgrad_body<WTAPS_ALL, WTAPS_ALL, WgGeoStd, VEC, KS>(a);
}
__global__ __launch_bounds__(256) void wgrad3x3_mfma_kernel(WgradArgs a) { wgrad_std_dispatch<false>(a); }
// the same with the vector staging (w % 4 == 0, 16-byte aligned operands)
__global__ __launch_bounds__(256) void wgrad3x3_mfma_vec_kernel(WgradArgs a) { wgrad_std_dispatch<true>(a); }

// g[e] = (accumulate ? g[e] : 0) + sum_s part[s][e]  over the written column range.
// A block reduces 64 consecutive outputs; its 4 waves take interleaved quarters of the splits
// (coalesced 256-byte rows, 4 independent chains per thread) and are combined through LDS in a
// fixed order, so the result does not depend on scheduling.
// swapped (a launch with <= 4 SHIFTED channels, run by the small-ca kernel with the operands exchanged):
//   G[a][b][ky][kx] = sum P[a](y, x) Q[b](y + ky - 1, x + kx - 1) seen from Q's side is G'[b][a][2 - ky][2 - kx];
//   the partials are [split][cb][ca][9] and entry (b, a, 8 - t) goes to grad[a][cb_off + b][t].
// The blocks past `gblocks` add the bias-gradient partials of the same launch ([bsplit][bn] -> db): one launch
// less per layer.
struct BiasTail { const float* part; float* db; int nsplit, n, gblocks; };
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part,
                                                           float* __restrict__ g, int nsplit, int ca,
                                                           int cb, int cb_total, int cb_off,
                                                           int accumulate, int swapped, BiasTail bt) {
  __shared__ float sm[4][64];
  if (bt.part && (int)blockIdx.x >= bt.gblocks) {
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int c = ((int)blockIdx.x - bt.gblocks) * 64 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < bt.n) {
      const float* p = bt.part + c;
      int k = sg;
      for (; k + 12 < bt.nsplit; k += 16) {
        s0 += p[(long long)k * bt.n]; s1 += p[(long long)(k + 4) * bt.n];
        s2 += p[(long long)(k + 8) * bt.n]; s3 += p[(long long)(k + 12) * bt.n];
      }
      for (; k < bt.nsplit; k += 4) s0 += p[(long long)k * bt.n];
    }
    sm[sg][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sg == 0 && c < bt.n) bt.db[c] = (accumulate ? bt.db[c] : 0.f) + ((sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]));
    return;
  }
  const long long stride = swapped ? (long long)cb * ca * 9 : (long long)ca * cb_total * 9;
  const long long total = (long long)ca * cb * 9;
  const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + o;
  long long e = 0;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < total) {
    int t = (int)(i % 9); long long r = i / 9;
    int bj = (int)(r % cb); int ai = (int)(r / cb);
    e = ((long long)ai * cb_total + cb_off + bj) * 9 + t;
    const float* p = part + (swapped ? ((long long)bj * ca + ai) * 9 + (8 - t) : e);
    int k = sg;
    for (; k + 12 < nsplit; k += 16) {
      s0 += p[(long long)k * stride];
      s1 += p[(long long)(k + 4) * stride];
      s2 += p[(long long)(k + 8) * stride];
      s3 += p[(long long)(k + 12) * stride];
    }
    for (; k < nsplit; k += 4) s0 += p[(long long)k * stride];
  }
  sm[sg][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sg == 0 && i < total) {
    float s = (sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]);
    g[e] = accumulate ? g[e] + s : s;
  }
}

// ---- ca <= 4: the weight gradient of an OUTPUT conv (conv_out 64 -> 3, tecogan_nets.py:131; flow[2]
// 32 -> 2, :65).  On the MFMA kernel the 3 gradient channels are padded to a 64-row tile: 95 % of the
// matrix work multiplies zeros, and on the HR frames of the unroll (19 x 2 x 256 x 256 pixels) that
// was the single most expensive launch of the training step (1.75 ms).  Here: plain FMAs.
//   block = 64 b-channels x 4 rows; tile = 4 rows x 32 columns of one image; the Q tile (64 x 6 x 34)
//   sits in LDS with an odd channel stride (conflict-free across the 64 lanes of a wave), the P
//   tile is read as LDS broadcasts; a thread walks its row with a sliding 3x3 window of Q and keeps
//   ca x 9 sums in registers across ALL the tiles its block visits (persistent blocks), the 4 row
//   groups meet in LDS at the end and the block writes ONE partial [ca][cb][9]; wgrad_reduce_kernel
//   adds the blocks' partials in a fixed order.
constexpr int SC_TW = 32, SC_TR = 4;
constexpr int SC_QRS = SC_TW + 2;                   // 34
constexpr int SC_QCS = (SC_TR + 2) * SC_QRS + 1;    // 205 (odd)
struct WgradSmallArgs {
  const float* pseg[WG_MAXSEG];
  const float* qseg[WG_MAXSEG];
  int n_per_seg;
  float* part;           // [nblk][ca][cb_total][9]
  long long p_ns, q_ns;
  int ca, cb, cb_total, cb_off, n, h, w, tiles_x, tiles_y, ntiles, nblk, nbb;
};
// CA = gradient rows actually computed (a.ca <= CA): conv_out's three rows cost three quarters of four
template <bool VEC, int CA>
__global__ __launch_bounds__(256) void wgrad3x3_smallca_kernel(WgradSmallArgs a) {
  __shared__ float sQ[64 * SC_QCS];                 // 52.5 KB
  __shared__ __attribute__((aligned(16))) float sP[4 * SC_TR * SC_TW];
  const int tid = threadIdx.x, bl = tid & 63, g = tid >> 6;
  const int bb = blockIdx.x % a.nbb, blk = blockIdx.x / a.nbb;
  const int b0 = bb * 64;
  const int hw = a.h * a.w;
  float acc[CA][9];
#pragma unroll
  for (int i = 0; i < CA; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = 0.f;
  // staging through registers: ALL loads of a tile are issued before the first LDS store, and the
  // loads of the block's NEXT tile are in flight while this one is being accumulated
  constexpr int QN = 64 * (SC_TR + 2) * SC_QRS, QPT = (QN + 255) / 256;      // 51 per thread
  constexpr int PN = 4 * SC_TR * SC_TW, PPT = PN / 256;                      // 2 per thread
  // vector form (w % 4 == 0, aligned planes; see wgrad_body): the Q rows as 16-byte groups from column
  // x0 - 4 -- position (row, group) = tid % 64 of the 6 x 10, the thread's wave walks channels w, w + 4, ...;
  // the P tile as 128 groups, one per thread of the first two waves
  constexpr int QV = 16;
  float rqv[VEC ? QV * 4 : QPT], rpv[VEC ? 4 : PPT];
  const int vq_pos = tid & 63, vq_r = vq_pos / 10, vq_k = vq_pos - vq_r * 10, vq_c = tid >> 6;
  const bool vq_on = vq_pos < 60;
  const int vp_c = tid >> 5, vp_r = (tid >> 3) & 3, vp_k = tid & 7;
  auto issue = [&](int tile) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int rem = tile - n * (a.tiles_x * a.tiles_y);
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int x0 = tx * SC_TW, y0 = ty * SC_TR;
    const int seg = n / a.n_per_seg, ln = n - seg * a.n_per_seg;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.qseg[seg] + (long long)ln * a.q_ns), 0, a.cb * hw * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.pseg[seg] + (long long)ln * a.p_ns), 0, a.ca * hw * 4, 0x00020000);
    if constexpr (VEC) {
      const int gy = y0 - 1 + vq_r, gx = x0 - 4 + 4 * vq_k;
      const bool ok = vq_on && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      const unsigned off0 = ok ? (unsigned)(((b0 + vq_c) * hw + gy * a.w + gx) * 4) : WG_OOB;
#pragma unroll
      for (int i = 0; i < QV; ++i) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rq, (int)(off0 + (unsigned)(4 * i) * (unsigned)hw * 4u), 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) rqv[4 * i + e] = v[e];
      }
      const int py = y0 + vp_r, px = x0 + 4 * vp_k;
      const bool pok = tid < 128 && py < a.h && px < a.w;
      const unsigned poff = pok ? (unsigned)((vp_c * hw + py * a.w + px) * 4) : WG_OOB;
      const f32x4 pv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)poff, 0, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) rpv[e] = pv[e];
      return;
    }
#pragma unroll
    for (int k = 0; k < (VEC ? 0 : QPT); ++k) {
      const int i = tid + k * 256;
      const int c = i % SC_QRS, r = (i / SC_QRS) % (SC_TR + 2), ch = i / (SC_QRS * (SC_TR + 2));
      const int gy = y0 - 1 + r, gx = x0 - 1 + c;
      const bool ok = i < QN && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      const unsigned off = ok ? (unsigned)(((b0 + ch) * hw + gy * a.w + gx) * 4) : WG_OOB;
      rqv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, (int)off, 0, 0));
    }
#pragma unroll
    for (int k = 0; k < (VEC ? 0 : PPT); ++k) {
      const int i = tid + k * 256;
      const int c = i % SC_TW, r = (i / SC_TW) % SC_TR, ch = i / (SC_TW * SC_TR);
      const int gy = y0 + r, gx = x0 + c;
      const bool ok = gy < a.h && gx < a.w;          // channels >= ca are past num_records: 0
      const unsigned off = ok ? (unsigned)((ch * hw + gy * a.w + gx) * 4) : WG_OOB;
      rpv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)off, 0, 0));
    }
  };
  auto commit = [&]() {
    if constexpr (VEC) {
      const int col0 = 4 * vq_k - 3;                     // patch column of element 0 of the group
      float* q = sQ + vq_c * SC_QCS + vq_r * SC_QRS + col0;
      if (vq_on) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (col0 + e >= 0 && col0 + e < SC_QRS) {
#pragma unroll
            for (int i = 0; i < QV; ++i) q[4 * i * SC_QCS + e] = rqv[4 * i + e];
          }
        }
      }
      if (tid < 128) *reinterpret_cast<f32x4*>(sP + tid * 4) = f32x4{rpv[0], rpv[1], rpv[2], rpv[3]};
      return;
    }
#pragma unroll
    for (int k = 0; k < (VEC ? 0 : QPT); ++k) {
      const int i = tid + k * 256;
      const int c = i % SC_QRS, r = (i / SC_QRS) % (SC_TR + 2), ch = i / (SC_QRS * (SC_TR + 2));
      if (i < QN) sQ[ch * SC_QCS + r * SC_QRS + c] = rqv[k];
    }
#pragma unroll
    for (int k = 0; k < (VEC ? 0 : PPT); ++k) sP[tid + k * 256] = rpv[k];
  };
  if (blk < a.ntiles) issue(blk);
  for (int tile = blk; tile < a.ntiles; tile += a.nblk) {
    __syncthreads();                                 // the previous tile's reads are done
    commit();
    __syncthreads();
    if (tile + a.nblk < a.ntiles) issue(tile + a.nblk);
    const float* q = sQ + bl * SC_QCS + g * SC_QRS;  // window rows g, g+1, g+2 of the tile (image rows y-1, y, y+1)
    float w0[3], w1[3], w2[3];                        // columns x-1, x, x+1
#pragma unroll
    for (int k = 0; k < 3; ++k) { w0[k] = q[k * SC_QRS]; w1[k] = q[k * SC_QRS + 1]; }
#pragma unroll 4
    for (int x = 0; x < SC_TW; ++x) {
#pragma unroll
      for (int k = 0; k < 3; ++k) w2[k] = q[k * SC_QRS + x + 2];
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const float p = sP[(i * SC_TR + g) * SC_TW + x];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc[i][k * 3 + 0] = __builtin_fmaf(p, w0[k], acc[i][k * 3 + 0]);
          acc[i][k * 3 + 1] = __builtin_fmaf(p, w1[k], acc[i][k * 3 + 1]);
          acc[i][k * 3 + 2] = __builtin_fmaf(p, w2[k], acc[i][k * 3 + 2]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) { w0[k] = w1[k]; w1[k] = w2[k]; }
    }
  }
  // the 4 row groups -> group 0 (fixed order), then the block's partial
  __syncthreads();
  float* red = sQ;                                    // [3][64][36]
  if (g > 0) {
#pragma unroll
    for (int i = 0; i < CA; ++i)
#pragma unroll
      for (int t = 0; t < 9; ++t) red[((g - 1) * 64 + bl) * 36 + i * 9 + t] = acc[i][t];
  }
  __syncthreads();
  if (g == 0 && b0 + bl < a.cb) {
    float* out = a.part + (long long)blk * a.ca * a.cb_total * 9;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      if (i >= a.ca) break;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float v = acc[i][t];
#pragma unroll
        for (int gg = 0; gg < 3; ++gg) v += red[(gg * 64 + bl) * 36 + i * 9 + t];
        out[((long long)i * a.cb_total + a.cb_off + b0 + bl) * 9 + t] = v;
      }
    }
  }
}

// the same reduction for every layer of a layered launch: blockIdx.y = layer
struct WgradLayerGrads { float* g[24]; };
__global__ __launch_bounds__(256) void wgrad_reduce_layers_kernel(const float* __restrict__ part, WgradLayerGrads gl,
                                                                  int nsplit, int ca, int cb, int accumulate) {
  __shared__ float sm[4][64];
  const long long stride = (long long)ca * cb * 9;
  const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + o;
  const float* p = part + (long long)blockIdx.y * nsplit * stride + i;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < stride) {
    int k = sg;
    for (; k + 12 < nsplit; k += 16) {
      s0 += p[(long long)k * stride];
      s1 += p[(long long)(k + 4) * stride];
      s2 += p[(long long)(k + 8) * stride];
      s3 += p[(long long)(k + 12) * stride];
    }
    for (; k < nsplit; k += 4) s0 += p[(long long)k * stride];
  }
  sm[sg][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sg == 0 && i < stride) {
    float s = (sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]);
    float* g = gl.g[blockIdx.y];
    g[i] = accumulate ? g[i] + s : s;
  }
}

}  // namespace tg

using namespace tg;

constexpr int SC_MAXBLK = 768;      // persistent blocks of the small-ca kernel (3 per CU)
// in-workgroup K split (WgradArgs::kmode): the workspace holds kmode's extra partials when cb_total <= 64
static int wgrad_kmode(int ca, int cb, int cb_total) {
  return (ca <= 32 ? 1 : 0) | (cb <= 32 && cb_total <= 64 ? 2 : 0);
}
// tile geometry of an un-phased launch: 0 = 2 x 32, 1 = 4 x 16 (maps up to 16 wide), 2 = 8 x 8 (up to 8 wide)
static int wgrad_geo(int h, int w) { return w <= 8 && h > 2 ? 2 : (w <= 16 && h > 2 ? 1 : 0); }
static void wgrad_tile(int geo, int* r, int* tw) {
  *r = geo == 2 ? 8 : (geo == 1 ? 4 : (geo == 3 ? 1 : WG_R));
  *tw = geo == 2 ? 8 : (geo == 1 ? 16 : WG_TW);
}
static int wgrad_nsplit(int n, int h, int w, int ca, int cb, int geo = 0) {
  if (ca <= 4 && geo != 3) {
    // persistent blocks: at most 3 per CU, and at least ~3 tiles each (a block's final reduction and
    // its partial cost about one tile)
    const int nt = n * cdiv(h, SC_TR) * cdiv(w, SC_TW), per_b = SC_MAXBLK / cdiv(cb, 64);
    int nb = nt < per_b ? nt : (per_b > 0 ? per_b : 1);
    if (nb > 256 / cdiv(cb, 64) && nb * 3 > nt) nb = nt / 3 > 256 / cdiv(cb, 64) ? nt / 3 : 256 / cdiv(cb, 64);
    return nb > 0 ? nb : 1;
  }
  int r, tw;
  wgrad_tile(geo, &r, &tw);
  int ntiles = n * cdiv(h, r) * cdiv(w, tw);
  int blocks_ch = cdiv(ca, 64) * cdiv(cb, 64);
  int s = 512 / blocks_ch;
  if (s < 1) s = 1;
  if (s > ntiles) s = ntiles;
  if (s > 256) s = 256;
  return s;
}

extern "C" size_t tg_wgrad3x3_workspace_floats(int n, int ca, int cb_total, int h, int w) {
  if (n <= 0 || ca <= 0 || cb_total <= 0 || h <= 0 || w <= 0) return 0;
  // (the un-phased launch may fold the tile for a narrow map; phased launches use 2 x 32: the larger count)
  const int s0 = wgrad_nsplit(n, h, w, ca, cb_total), s1 = wgrad_nsplit(n, h, w, ca, cb_total, wgrad_geo(h, w));
  const int ks = (ca <= 32 ? 2 : 1) * (cb_total <= 64 ? 2 : 1);        // room for wgrad_kmode's extra partials
  size_t need = (size_t)(s0 > s1 ? s0 : s1) * ks * ca * cb_total * 9 + (size_t)256 * ca;      // + bias partials
  if (ca > 4) {        // a launch over <= 4 of the cb_total columns goes to the small-ca kernel, operands exchanged
    const size_t sw = (size_t)wgrad_nsplit(n, h, w, 4, ca) * 4 * ca * 9;
    if (sw > need) need = sw;
  }
  return need;
}

extern "C" size_t tg_wgrad3x3_convt_workspace_floats(int n, int ci, int co, int h, int w) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w <= 0) return 0;
  const size_t ns = (size_t)wgrad_nsplit(n, h, w, ci, co, 3);
  return ns * ci * co * 9 + ns * co;                 // + the bias-gradient partials
}

template <class G, bool VEC>
static void launch_geo2(const WgradArgs& a, unsigned blocks, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_mfma_geo_kernel<G, VEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad3x3_mfma_geo_kernel<G, VEC>), dim3(blocks), dim3(256), G::LDS_BYTES, s, a);
}
template <class G>
static void launch_geo(const WgradArgs& a, unsigned blocks, hipStream_t s, bool vec) {
  if (vec) launch_geo2<G, true>(a, blocks, s); else launch_geo2<G, false>(a, blocks, s);
}
template <bool VEC, int KS>
static void launch_ks(const WgradArgs& a, unsigned blocks, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_mfma_ks_kernel<VEC, KS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgGeoStd::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad3x3_mfma_ks_kernel<VEC, KS>), dim3(blocks), dim3(256), WgGeoStd::LDS_BYTES, s, a);
}
static void launch_std(const WgradArgs& a, unsigned blocks, hipStream_t s, bool vec) {
  if (a.kmode == 3) { if (vec) launch_ks<true, 4>(a, blocks, s); else launch_ks<false, 4>(a, blocks, s); return; }
  if (a.kmode) { if (vec) launch_ks<true, 2>(a, blocks, s); else launch_ks<false, 2>(a, blocks, s); return; }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_mfma_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgGeoStd::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_mfma_vec_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgGeoStd::LDS_BYTES);
    attr_set = true;
  }
  if (vec) hipLaunchKernelGGL(wgrad3x3_mfma_vec_kernel, dim3(blocks), dim3(256), WgGeoStd::LDS_BYTES, s, a);
  else hipLaunchKernelGGL(wgrad3x3_mfma_kernel, dim3(blocks), dim3(256), WgGeoStd::LDS_BYTES, s, a);
}
static void launch_smallca(const WgradSmallArgs& sa, int nseg, hipStream_t s) {
  bool vec = sa.w % 4 == 0 && sa.p_ns % 4 == 0 && sa.q_ns % 4 == 0;
  for (int i = 0; i < nseg && vec; ++i)
    if (((uintptr_t)sa.pseg[i] | (uintptr_t)sa.qseg[i]) & 15) vec = false;
  static const int novec = TG_LAB_ENV("TG_WGRAD_NOVEC", 0);
  const dim3 g((unsigned)(sa.nblk * sa.nbb)), t(256);
  const bool v = vec && !novec;
  switch (sa.ca) {
    case 1: if (v) hipLaunchKernelGGL((wgrad3x3_smallca_kernel<true, 1>), g, t, 0, s, sa); else hipLaunchKernelGGL((wgrad3x3_smallca_kernel<false, 1>), g, t, 0, s, sa); break;
    case 2: if (v) hipLaunchKernelGGL((wgrad3x3_smallca_kernel<true, 2>), g, t, 0, s, sa); else hipLaunchKernelGGL((wgrad3x3_smallca_kernel<false, 2>), g, t, 0, s, sa); break;
    case 3: if (v) hipLaunchKernelGGL((wgrad3x3_smallca_kernel<true, 3>), g, t, 0, s, sa); else hipLaunchKernelGGL((wgrad3x3_smallca_kernel<false, 3>), g, t, 0, s, sa); break;
    default: if (v) hipLaunchKernelGGL((wgrad3x3_smallca_kernel<true, 4>), g, t, 0, s, sa); else hipLaunchKernelGGL((wgrad3x3_smallca_kernel<false, 4>), g, t, 0, s, sa); break;
  }
}
// the vector staging needs whole 16-byte groups: widths that are multiples of 4 and aligned planes
static bool wgrad_vec_ok(const WgradArgs& a, int nseg) {
  if (a.w % 4 != 0 || a.p_ns % 4 != 0 || a.q_ns % 4 != 0 || a.lstride % 4 != 0) return false;
  for (int i = 0; i < nseg; ++i)
    if (((uintptr_t)a.pseg[i] | (uintptr_t)a.qseg[i]) & 15) return false;
  return true;
}

static int wgrad_launch(const float* const* p_list, const float* const* q_list, int nseg,
                        int64_t p_nstride, int64_t q_nstride, float* grad, float* workspace,
                        int n_per_seg, int ca, int cb, int cb_total, int cb_off, int h, int w,
                        int accumulate, tg_stream_t stream, int cphase = 0, int set_p0 = 0,
                        int set_p1 = 0, int stride2 = 0, float* bias_grad = nullptr) {
  TG_REQUIRE(p_list && q_list && grad && workspace, TG_E_ARG, "wgrad3x3: null pointer");
  TG_REQUIRE(nseg >= 1 && nseg <= WG_MAXSEG, TG_E_ARG, "wgrad3x3: %d segments (1..%d)", nseg, WG_MAXSEG);
  TG_REQUIRE(n_per_seg > 0 && ca > 0 && cb > 0 && h > 0 && w > 0 && cb_off >= 0 &&
                 cb_off + cb <= cb_total, TG_E_SHAPE,
             "wgrad3x3: n=%dx%d ca=%d cb=%d (+%d of %d) h=%d w=%d", nseg, n_per_seg, ca, cb, cb_off,
             cb_total, h, w);
  TG_REQUIRE((long long)(ca > (stride2 ? 4 * cb : cb) ? ca : (stride2 ? 4 * cb : cb)) * h * w * 4 < (1ll << 31), TG_E_SHAPE,
             "wgrad3x3: one batch item must be < 2 GiB");
  const int geo = stride2 ? 3 : (cphase ? 0 : wgrad_geo(h, w));
  if (ca <= 4 && !cphase && !stride2) {
    WgradSmallArgs sa{};
    for (int i = 0; i < nseg; ++i) {
      TG_REQUIRE(p_list[i] && q_list[i], TG_E_ARG, "wgrad3x3: null segment %d", i);
      sa.pseg[i] = p_list[i]; sa.qseg[i] = q_list[i];
    }
    const int n = nseg * n_per_seg;
    sa.n_per_seg = n_per_seg; sa.part = workspace; sa.p_ns = p_nstride; sa.q_ns = q_nstride;
    sa.ca = ca; sa.cb = cb; sa.cb_total = cb_total; sa.cb_off = cb_off; sa.n = n; sa.h = h; sa.w = w;
    sa.tiles_x = cdiv(w, SC_TW); sa.tiles_y = cdiv(h, SC_TR); sa.ntiles = n * sa.tiles_x * sa.tiles_y;
    sa.nbb = cdiv(cb, 64);
    sa.nblk = wgrad_nsplit(n, h, w, ca, cb_total);        // = what the workspace was sized with (cb <= cb_total)
    hipStream_t s = (hipStream_t)stream;
    launch_smallca(sa, nseg, s);
    int rc = check_launch("wgrad3x3_smallca");
    if (rc != TG_OK) return rc;
    const long long total = (long long)ca * cb * 9;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, workspace, grad,
                       sa.nblk, ca, cb, cb_total, cb_off, accumulate, 0, BiasTail{});
    rc = check_launch("wgrad_reduce");
    if (rc != TG_OK || !bias_grad) return rc;
    return tg_bias_grad_multi(p_list, nseg, bias_grad, n_per_seg, ca, h * w, accumulate, stream);
  }
  if (cb <= 4 && !cphase && !stride2) {
    // few SHIFTED channels: the small-ca kernel with the operands exchanged (wgrad_reduce_swapped_kernel)
    WgradSmallArgs sa{};
    for (int i = 0; i < nseg; ++i) {
      TG_REQUIRE(p_list[i] && q_list[i], TG_E_ARG, "wgrad3x3: null segment %d", i);
      sa.pseg[i] = q_list[i]; sa.qseg[i] = p_list[i];
    }
    const int n = nseg * n_per_seg;
    sa.n_per_seg = n_per_seg; sa.part = workspace; sa.p_ns = q_nstride; sa.q_ns = p_nstride;
    sa.ca = cb; sa.cb = ca; sa.cb_total = ca; sa.cb_off = 0; sa.n = n; sa.h = h; sa.w = w;
    sa.tiles_x = cdiv(w, SC_TW); sa.tiles_y = cdiv(h, SC_TR); sa.ntiles = n * sa.tiles_x * sa.tiles_y;
    sa.nbb = cdiv(ca, 64);
    sa.nblk = wgrad_nsplit(n, h, w, cb, ca);               // the small-ca rule with the roles exchanged
    hipStream_t s = (hipStream_t)stream;
    launch_smallca(sa, nseg, s);
    int rc = check_launch("wgrad3x3_smallca(swapped)");
    if (rc != TG_OK) return rc;
    const long long total = (long long)ca * cb * 9;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, workspace,
                       grad, sa.nblk, ca, cb, cb_total, cb_off, accumulate, 1, BiasTail{});
    rc = check_launch("wgrad_reduce(swapped)");
    if (rc != TG_OK || !bias_grad) return rc;
    return tg_bias_grad_multi(p_list, nseg, bias_grad, n_per_seg, ca, h * w, accumulate, stream);
  }
  WgradArgs a{};
  for (int i = 0; i < nseg; ++i) {
    TG_REQUIRE(p_list[i] && q_list[i], TG_E_ARG, "wgrad3x3: null segment %d", i);
    a.pseg[i] = p_list[i]; a.qseg[i] = q_list[i];
  }
  const int n = nseg * n_per_seg;
  a.n_per_seg = n_per_seg;
  a.part = workspace; a.p_ns = p_nstride; a.q_ns = q_nstride;
  a.ca = ca; a.cb = cb; a.cb_total = cb_total; a.cb_off = cb_off; a.n = n; a.h = h; a.w = w;
  int tile_r, tile_w;
  wgrad_tile(geo, &tile_r, &tile_w);
  a.tiles_x = cdiv(w, tile_w); a.tiles_y = cdiv(h, tile_r);
  a.ntiles = n * a.tiles_x * a.tiles_y;
  a.nab = cdiv(ca, 64); a.nbb = cdiv(cb, 64);
  a.nsplit = wgrad_nsplit(n, h, w, ca, cb_total, geo);   // at most what the workspace was sized with
  static const int maxwg = TG_LAB_ENV("TG_WGRAD_MAXWG", 0);     // lab: cap the K split (workgroups per channel block)
  if (maxwg > 0 && a.nsplit > maxwg) a.nsplit = maxwg;
  a.cphase = cphase; a.rowsets[0] = (unsigned char)set_p0; a.rowsets[1] = (unsigned char)set_p1;
  if (cphase) {
    TG_REQUIRE(cphase % 64 == 0 && cb == 4 * cphase && cb_off == 0 && set_p0 >= 0 && set_p0 <= 3 &&
                   set_p1 >= 0 && set_p1 <= 3, TG_E_ARG,
               "wgrad3x3: phased taps need cb = 4 phases of a multiple of 64 channels (cb=%d cphase=%d)", cb,
               cphase);
  }
  unsigned blocks = (unsigned)(a.nab * a.nbb * a.nsplit);
  hipStream_t s = (hipStream_t)stream;
  static const int novec = TG_LAB_ENV("TG_WGRAD_NOVEC", 0);      // lab: A/B of the two staging forms
  const bool vec = !novec && wgrad_vec_ok(a, nseg);
  if (geo == 0 && !cphase) a.kmode = wgrad_kmode(ca, cb, cb_total);
  const int ks = a.kmode == 3 ? 4 : (a.kmode ? 2 : 1);
  // bias gradient (db = sum of P = dZ over images and pixels): out of the P values the vector form stages
  const bool fuse_a = bias_grad && !stride2 && vec && !cphase;
  if (fuse_a) a.bias_part = workspace + (size_t)a.nsplit * ks * ca * cb_total * 9;
  if (geo == 0) {
    launch_std(a, blocks, s, vec);
  } else if (geo == 1) {
    launch_geo<WgGeo<4, 16, 0>>(a, blocks, s, vec);
  } else if (geo == 2) {
    launch_geo<WgGeo<8, 8, 0>>(a, blocks, s, vec);
  } else {
    // bias gradient of the transposed conv: out of the staged dZ when the vector form runs (and cb fits one block)
    const bool fuse_b = bias_grad && vec && a.nbb == 1;
    a.bias_part = fuse_b ? workspace + (size_t)a.nsplit * ca * cb_total * 9 : nullptr;
    launch_geo<WgGeo<1, 32, 1>>(a, blocks, s, vec);
    if (bias_grad && !fuse_b) {      // element-wise staging / several channel blocks: the stand-alone reduction over the HR tensors
      int rb_ = check_launch("wgrad3x3_convt");
      if (rb_ != TG_OK) return rb_;
      rb_ = tg_bias_grad_multi(q_list, nseg, bias_grad, n_per_seg, cb, 4 * h * w, accumulate, stream);
      if (rb_ != TG_OK) return rb_;
    }
  }
  int rc = check_launch("wgrad3x3_mfma");
  if (rc != TG_OK) return rc;
  long long total = (long long)ca * cb * 9;
  // the bias partials of the launch (db over ca: P = dZ; stride-2 form: over cb, Q = dZ) are added by the tail
  // blocks of the same reduce launch
  BiasTail bt{};
  const int gblocks = (int)((total + 63) / 64);
  if (a.bias_part) { bt.part = a.bias_part; bt.db = bias_grad; bt.nsplit = a.nsplit; bt.n = stride2 ? cb : ca; bt.gblocks = gblocks; }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(gblocks + (a.bias_part ? cdiv(bt.n, 64) : 0))), dim3(256), 0, s,
                     workspace, grad, a.nsplit * ks, ca, cb, cb_total, cb_off, accumulate, 0, bt);
  rc = check_launch("wgrad_reduce");
  if (rc != TG_OK || !bias_grad || stride2 || fuse_a) return rc;
  return tg_bias_grad_multi(p_list, nseg, bias_grad, n_per_seg, ca, h * w, accumulate, stream);
}

// K split of a layered launch: ~5 rounds of workgroups over the device, whole tiles per workgroup
static int body_nsplit(int nframes, int n_per_frame, int nlayers, int c, int h, int w) {
  const int ntiles = nframes * n_per_frame * cdiv(h, WG_R) * cdiv(w, WG_TW);
  const int blocks_ch = cdiv(c, 64) * cdiv(c, 64);
  int s = 1280 / (nlayers * blocks_ch);
  if (s < 1) s = 1;
  if (s > ntiles) s = ntiles;
  if (s > 256) s = 256;
  return s;
}

extern "C" size_t tg_wgrad3x3_body_workspace_floats(int nframes, int n_per_frame, int nlayers, int c, int h, int w) {
  if (nframes <= 0 || n_per_frame <= 0 || nlayers <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
  const size_t ns = (size_t)body_nsplit(nframes, n_per_frame, nlayers, c, h, w);
  return (size_t)nlayers * ns * c * c * 9 + (size_t)nlayers * ns * c;          // + bias partials
}

static int wgrad_body_impl(const float* const* dz_bases, const float* const* act_bases,
                           int nframes, int64_t layer_stride, int nlayers,
                           float* const* grads, float* const* dbs, float* workspace, int n_per_frame, int c, int h, int w,
                           int accumulate, tg_stream_t stream) {
  TG_REQUIRE(dz_bases && act_bases && grads && workspace, TG_E_ARG, "wgrad3x3_body: null pointer");
  TG_REQUIRE(nframes >= 1 && nframes <= WG_MAXSEG && nlayers >= 1 && nlayers <= 24, TG_E_ARG,
             "wgrad3x3_body: %d frames (1..%d), %d layers (1..24)", nframes, WG_MAXSEG, nlayers);
  TG_REQUIRE(n_per_frame > 0 && c > 0 && h > 0 && w > 0 && layer_stride >= (int64_t)n_per_frame * c * h * w,
             TG_E_SHAPE, "wgrad3x3_body: n=%d c=%d h=%d w=%d stride=%lld", n_per_frame, c, h, w, (long long)layer_stride);
  TG_REQUIRE((long long)c * h * w * 4 < (1ll << 31), TG_E_SHAPE, "wgrad3x3_body: one batch item must be < 2 GiB");
  WgradArgs a{};
  WgradLayerGrads gl{};
  for (int i = 0; i < nframes; ++i) {
    TG_REQUIRE(dz_bases[i] && act_bases[i], TG_E_ARG, "wgrad3x3_body: null frame %d", i);
    a.pseg[i] = dz_bases[i]; a.qseg[i] = act_bases[i];
  }
  for (int i = 0; i < nlayers; ++i) {
    TG_REQUIRE(grads[i], TG_E_ARG, "wgrad3x3_body: null gradient %d", i);
    gl.g[i] = grads[i];
  }
  const int n = nframes * n_per_frame;
  a.n_per_seg = n_per_frame; a.part = workspace;
  a.p_ns = a.q_ns = (long long)c * h * w;
  a.ca = a.cb = a.cb_total = c; a.cb_off = 0; a.n = n; a.h = h; a.w = w;
  a.tiles_x = cdiv(w, WG_TW); a.tiles_y = cdiv(h, WG_R);
  a.ntiles = n * a.tiles_x * a.tiles_y;
  a.nab = a.nbb = cdiv(c, 64);
  a.nsplit = body_nsplit(nframes, n_per_frame, nlayers, c, h, w);
  a.nlayer = nlayers; a.lstride = layer_stride;
  hipStream_t s = (hipStream_t)stream;
  static const int novec = TG_LAB_ENV("TG_WGRAD_NOVEC", 0);
  const bool vec = !novec && wgrad_vec_ok(a, nframes);
  WgradLayerBias bl{};
  if (dbs) {
    for (int i = 0; i < nlayers; ++i) {
      TG_REQUIRE(dbs[i], TG_E_ARG, "wgrad3x3_body: null bias gradient %d", i);
      bl.b[i] = dbs[i];
    }
    if (vec) a.bias_part = workspace + (size_t)nlayers * a.nsplit * c * c * 9;
  }
  launch_std(a, (unsigned)(nlayers * a.nab * a.nbb * a.nsplit), s, vec);
  int rc = check_launch("wgrad3x3_body");
  if (rc != TG_OK) return rc;
  const long long total = (long long)c * c * 9;
  hipLaunchKernelGGL(wgrad_reduce_layers_kernel, dim3((unsigned)((total + 63) / 64), (unsigned)nlayers), dim3(256), 0, s,
                     workspace, gl, a.nsplit, c, c, accumulate);
  rc = check_launch("wgrad_reduce_layers");
  if (rc != TG_OK || !dbs) return rc;
  if (a.bias_part) {
    hipLaunchKernelGGL(bias_part_reduce_layers_kernel, dim3((unsigned)cdiv(c, 64), (unsigned)nlayers), dim3(256), 0, s,
                       a.bias_part, bl, a.nsplit, c, accumulate);
    return check_launch("bias_part_reduce_layers");
  }
  // element-wise staging: the stand-alone reduction, layer by layer (dZ of layer L = 1..nlayers at base + L * stride)
  const float* segs[WG_MAXSEG];
  for (int L = 1; L <= nlayers; ++L) {
    for (int f = 0; f < nframes; ++f) segs[f] = dz_bases[f] + (size_t)L * layer_stride;
    rc = tg_bias_grad_multi(segs, nframes, dbs[L - 1], n_per_frame, c, h * w, accumulate, stream);
    if (rc != TG_OK) return rc;
  }
  return TG_OK;
}

extern "C" int tg_wgrad3x3_body(const float* const* dz_bases, const float* const* act_bases,
                                int nframes, int64_t layer_stride, int nlayers,
                                float* const* grads, float* workspace, int n_per_frame, int c, int h, int w,
                                int accumulate, tg_stream_t stream) {
  return wgrad_body_impl(dz_bases, act_bases, nframes, layer_stride, nlayers, grads, nullptr, workspace, n_per_frame, c,
                         h, w, accumulate, stream);
}

// ... and the bias gradients of the same layers (dbs[L - 1] (+)= sum of dZ of layer L = 1..nlayers) from the dZ
// values the launch stages anyway -- tg_bias_grad_body then only has the chain's first layer left to do.
extern "C" int tg_wgrad3x3_body_bias(const float* const* dz_bases, const float* const* act_bases,
                                     int nframes, int64_t layer_stride, int nlayers,
                                     float* const* grads, float* const* dbs, float* workspace, int n_per_frame, int c,
                                     int h, int w, int accumulate, tg_stream_t stream) {
  TG_REQUIRE(dbs, TG_E_ARG, "wgrad3x3_body_bias: null pointer");
  return wgrad_body_impl(dz_bases, act_bases, nframes, layer_stride, nlayers, grads, dbs, workspace, n_per_frame, c,
                         h, w, accumulate, stream);
}

extern "C" int tg_wgrad3x3(const float* p, int64_t p_nstride, const float* q, int64_t q_nstride,
                           float* grad, float* workspace, int n, int ca, int cb, int cb_total,
                           int cb_off, int h, int w, int accumulate, tg_stream_t stream) {
  TG_REQUIRE(p && q, TG_E_ARG, "wgrad3x3: null pointer");
  return wgrad_launch(&p, &q, 1, p_nstride, q_nstride, grad, workspace, n, ca, cb, cb_total, cb_off, h,
                      w, accumulate, stream);
}

extern "C" int tg_wgrad3x3_multi_phased(const float* const* p_list, const float* const* q_list, int nseg,
                                        int64_t p_nstride, int64_t q_nstride, float* grad,
                                        float* workspace, int n_per_seg, int ca, int cb, int h, int w,
                                        int accumulate, int cphase, int taps_phase0, int taps_phase1,
                                        tg_stream_t stream) {
  TG_REQUIRE(cphase > 0, TG_E_ARG, "wgrad3x3_multi_phased: cphase=%d", cphase);
  return wgrad_launch(p_list, q_list, nseg, p_nstride, q_nstride, grad, workspace, n_per_seg, ca, cb,
                      cb, 0, h, w, accumulate, stream, cphase, taps_phase0, taps_phase1);
}

// tg_wgrad3x3_multi that also takes the layer's bias gradient (bias_grad (ca,) (+)= sum of P over images and
// pixels) from the P values it stages anyway; forms without the vector staging run the stand-alone reduction.
extern "C" int tg_wgrad3x3_multi_bias(const float* const* p_list, const float* const* q_list, int nseg,
                                      int64_t p_nstride, int64_t q_nstride, float* grad, float* bias_grad,
                                      float* workspace, int n_per_seg, int ca, int cb, int cb_total, int cb_off,
                                      int h, int w, int accumulate, tg_stream_t stream) {
  return wgrad_launch(p_list, q_list, nseg, p_nstride, q_nstride, grad, workspace, n_per_seg, ca, cb,
                      cb_total, cb_off, h, w, accumulate, stream, 0, 0, 0, 0, bias_grad);
}

// dW of ConvTranspose2d(ci, co, 3, 2, 1, output_padding 1): x_list[i] (n_per_seg, ci, h, w) = the layer's
// inputs, dz_list[i] (n_per_seg, co, 2h, 2w) = the gradients of its pre-activation; grad (ci, co, 3, 3)
// in the layer's own weight layout.  No space-to-depth copy of dZ, the nine taps in one balanced pass.
extern "C" int tg_wgrad3x3_convt_multi(const float* const* x_list, const float* const* dz_list, int nseg,
                                       float* grad, float* bias_grad, float* workspace, int n_per_seg, int ci, int co,
                                       int h, int w, int accumulate, tg_stream_t stream) {
  return wgrad_launch(x_list, dz_list, nseg, (int64_t)ci * h * w, (int64_t)co * 4 * h * w, grad, workspace, n_per_seg,
                      ci, co, co, 0, h, w, accumulate, stream, 0, 0, 0, 1, bias_grad);
}

extern "C" int tg_wgrad3x3_multi(const float* const* p_list, const float* const* q_list, int nseg,
                                 int64_t p_nstride, int64_t q_nstride, float* grad, float* workspace,
                                 int n_per_seg, int ca, int cb, int cb_total, int cb_off, int h,
                                 int w, int accumulate, tg_stream_t stream) {
  return wgrad_launch(p_list, q_list, nseg, p_nstride, q_nstride, grad, workspace, n_per_seg, ca, cb,
                      cb_total, cb_off, h, w, accumulate, stream);
}

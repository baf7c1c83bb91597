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
};

enum { WTAPS_ALL = 0, WTAPS_01 = 1, WTAPS_12 = 2, WTAPS_1 = 3 };
__host__ __device__ constexpr bool wtap_on(int code, int k) {
  return code == WTAPS_ALL || (code == WTAPS_01 && k <= 1) || (code == WTAPS_12 && k >= 1) ||
         (code == WTAPS_1 && k == 1);
}

template <int RY, int RX>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                          // [2][WG_A_FLOATS]
  float* sB = smem + 2 * WG_A_FLOATS;        // [2][WG_B_FLOATS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cot = wave & 1, cit = wave >> 1;
  int b = blockIdx.x;
  const int split = b % a.nsplit; b /= a.nsplit;
  const int bb = b % a.nbb; b /= a.nbb;
  const int ab = b % a.nab;
  const int li = b / a.nab;                  // layer index (0 outside the layered mode)
  const int a0 = ab * 64, b0 = bb * 64;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int lh = lane >> 5, ll = lane & 31;
  const int a_rd = (cot * 32 + ll) * WG_CSA + 4 * lh;
  const int b_rd = (cit * 32 + ll) * WG_CSB + 4 * lh;

  constexpr int A_PER_T = (64 * WG_R * WG_TW) / 256;                 // 16
  constexpr int B_ELEMS = 64 * (WG_R + 2) * (WG_TW + 2);            // 8704
  constexpr int B_PER_T = B_ELEMS / 256;                             // 34
  float ra[A_PER_T], rb[B_PER_T];

  auto load_tile = [&](int tile) {
    int n = tile / (a.tiles_x * a.tiles_y);
    int rem = tile - n * (a.tiles_x * a.tiles_y);
    int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    int x0 = tx * WG_TW, y0 = ty * WG_R;
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
        const_cast<float*>(qbase + (long long)ln * a.q_ns), 0, a.cb * hw * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx >> 6, r = (idx >> 5) & 1, col = idx & 31;
      int gy = y0 + r, gx = x0 + col;
      bool ok = gy < a.h && gx < a.w;      // channel tail handled by num_records
      unsigned off = ok ? ((unsigned)(a0 + c) * plane + (unsigned)(gy * a.w + gx) * 4u) : WG_OOB;
      ra[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / ((WG_R + 2) * (WG_TW + 2));
      int rem2 = idx - c * ((WG_R + 2) * (WG_TW + 2));
      int r = rem2 / (WG_TW + 2), col = rem2 - r * (WG_TW + 2);
      int gy = y0 - 1 + r, gx = x0 - 1 + col;
      bool ok = gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      unsigned off = ok ? ((unsigned)(b0 + c) * plane + (unsigned)(gy * a.w + gx) * 4u) : WG_OOB;
      rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, (int)off, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
    float* pa = sA + buf * WG_A_FLOATS;
    float* pb = sB + buf * WG_B_FLOATS;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx >> 6, rc = idx & 63;
      pa[c * WG_CSA + rc] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / ((WG_R + 2) * (WG_TW + 2));
      int rem2 = idx - c * ((WG_R + 2) * (WG_TW + 2));
      int r = rem2 / (WG_TW + 2), col = rem2 - r * (WG_TW + 2);
      pb[c * WG_CSB + r * WG_RSB + col] = rb[i];
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
  for (; tile < a.ntiles; tile += a.nsplit, ++it) {
    const int buf = it & 1;
    const bool more = tile + a.nsplit < a.ntiles;
    if (more && !WGABL(1)) load_tile(tile + a.nsplit);
    const float* pa = sA + buf * WG_A_FLOATS + a_rd;
    const float* pb = sB + buf * WG_B_FLOATS + b_rd;
#pragma unroll
    for (int r = 0; r < WG_R; ++r) {
#pragma unroll
      for (int g = 0; g < WG_TW / 8; ++g) {
        f32x4 av = {1.f, 2.f, 3.f, 4.f};
        float bv[3][6];
        if (!WGABL(8)) {
          av = *reinterpret_cast<const f32x4*>(pa + r * WG_TW + 8 * g);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) bv[ky][c6] = pb[(r + ky) * WG_RSB + 8 * g + c6];
        } else {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) bv[ky][c6] = (float)(ky + c6 + r + g);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if (wtap_on(RY, ky) && wtap_on(RX, kx)) {
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                if (WGABL(4)) acc[ky * 3 + kx][kk] += av[kk] * bv[ky][kk + kx];
                else
                acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[ky][kk + kx],
                                                                        acc[ky * 3 + kx], 0, 0, 0);
              }
            }
      }
    }
    if (more && !WGABL(2)) store_tile(buf ^ 1);
    __syncthreads();
  }

  // D[i = a-channel][j = b-channel]: lane holds j = ll, registers walk i
  float* out = a.part + ((long long)li * a.nsplit + split) * a.ca * a.cb_total * 9;
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
}

__global__ __launch_bounds__(256) void wgrad3x3_mfma_kernel(WgradArgs a) {
  if (a.cphase == 0) { wgrad_body<WTAPS_ALL, WTAPS_ALL>(a); return; }
  // block-uniform: the 64 b channels of this block belong to one sub-pixel phase
  const int bb = ((int)blockIdx.x / a.nsplit) % a.nbb;      // (phased launches are never layered)
  const int ph = (bb * 64) / a.cphase;
  const int ry = a.rowsets[(ph >> 1) & 1], rx = a.rowsets[ph & 1];
  switch (ry * 4 + rx) {
    case WTAPS_01 * 4 + WTAPS_01: wgrad_body<WTAPS_01, WTAPS_01>(a); break;
    case WTAPS_01 * 4 + WTAPS_12: wgrad_body<WTAPS_01, WTAPS_12>(a); break;
    case WTAPS_12 * 4 + WTAPS_01: wgrad_body<WTAPS_12, WTAPS_01>(a); break;
    case WTAPS_12 * 4 + WTAPS_12: wgrad_body<WTAPS_12, WTAPS_12>(a); break;
    case WTAPS_01 * 4 + WTAPS_1: wgrad_body<WTAPS_01, WTAPS_1>(a); break;
    case WTAPS_1 * 4 + WTAPS_01: wgrad_body<WTAPS_1, WTAPS_01>(a); break;
    case WTAPS_1 * 4 + WTAPS_1: wgrad_body<WTAPS_1, WTAPS_1>(a); break;
    default: wgrad_body<WTAPS_ALL, WTAPS_ALL>(a); break;
  }
}

// g[e] = (accumulate ? g[e] : 0) + sum_s part[s][e]  over the written column range.
// A block reduces 64 consecutive outputs; its 4 waves take interleaved quarters of the splits
// (coalesced 256-byte rows, 4 independent chains per thread) and are combined through LDS in a
// fixed order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part,
                                                           float* __restrict__ g, int nsplit, int ca,
                                                           int cb, int cb_total, int cb_off,
                                                           int accumulate) {
  __shared__ float sm[4][64];
  const long long stride = (long long)ca * cb_total * 9;
  const long long total = (long long)ca * cb * 9;
  const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + o;
  long long e = 0;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < total) {
    int t = (int)(i % 9); long long r = i / 9;
    int bj = (int)(r % cb); int ai = (int)(r / cb);
    e = ((long long)ai * cb_total + cb_off + bj) * 9 + t;
    const float* p = part + e;
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
__global__ __launch_bounds__(256) void wgrad3x3_smallca_kernel(WgradSmallArgs a) {
  __shared__ float sQ[64 * SC_QCS];                 // 52.5 KB
  __shared__ float sP[4 * SC_TR * SC_TW];
  const int tid = threadIdx.x, bl = tid & 63, g = tid >> 6;
  const int bb = blockIdx.x % a.nbb, blk = blockIdx.x / a.nbb;
  const int b0 = bb * 64;
  const int hw = a.h * a.w;
  float acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = 0.f;
  // staging through registers: ALL loads of a tile are issued before the first LDS store, and the
  // loads of the block's NEXT tile are in flight while this one is being accumulated
  constexpr int QN = 64 * (SC_TR + 2) * SC_QRS, QPT = (QN + 255) / 256;      // 51 per thread
  constexpr int PN = 4 * SC_TR * SC_TW, PPT = PN / 256;                      // 2 per thread
  float rqv[QPT], rpv[PPT];
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
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
      const int i = tid + k * 256;
      const int c = i % SC_QRS, r = (i / SC_QRS) % (SC_TR + 2), ch = i / (SC_QRS * (SC_TR + 2));
      const int gy = y0 - 1 + r, gx = x0 - 1 + c;
      const bool ok = i < QN && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      const unsigned off = ok ? (unsigned)(((b0 + ch) * hw + gy * a.w + gx) * 4) : WG_OOB;
      rqv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, (int)off, 0, 0));
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int i = tid + k * 256;
      const int c = i % SC_TW, r = (i / SC_TW) % SC_TR, ch = i / (SC_TW * SC_TR);
      const int gy = y0 + r, gx = x0 + c;
      const bool ok = gy < a.h && gx < a.w;          // channels >= ca are past num_records: 0
      const unsigned off = ok ? (unsigned)((ch * hw + gy * a.w + gx) * 4) : WG_OOB;
      rpv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)off, 0, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
      const int i = tid + k * 256;
      const int c = i % SC_QRS, r = (i / SC_QRS) % (SC_TR + 2), ch = i / (SC_QRS * (SC_TR + 2));
      if (i < QN) sQ[ch * SC_QCS + r * SC_QRS + c] = rqv[k];
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) sP[tid + k * 256] = rpv[k];
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
      for (int i = 0; i < 4; ++i) {
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
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 9; ++t) red[((g - 1) * 64 + bl) * 36 + i * 9 + t] = acc[i][t];
  }
  __syncthreads();
  if (g == 0 && b0 + bl < a.cb) {
    float* out = a.part + (long long)blk * a.ca * a.cb_total * 9;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
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
static int wgrad_nsplit(int n, int h, int w, int ca, int cb) {
  if (ca <= 4) {
    const int nt = n * cdiv(h, SC_TR) * cdiv(w, SC_TW), per_b = SC_MAXBLK / cdiv(cb, 64);
    return nt < per_b ? nt : (per_b > 0 ? per_b : 1);
  }
  int ntiles = n * cdiv(h, WG_R) * cdiv(w, WG_TW);
  int blocks_ch = cdiv(ca, 64) * cdiv(cb, 64);
  int s = 512 / blocks_ch;
  if (s < 1) s = 1;
  if (s > ntiles) s = ntiles;
  if (s > 256) s = 256;
  return s;
}

extern "C" size_t tg_wgrad3x3_workspace_floats(int n, int ca, int cb_total, int h, int w) {
  if (n <= 0 || ca <= 0 || cb_total <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)wgrad_nsplit(n, h, w, ca, cb_total) * ca * cb_total * 9;
}

static int wgrad_launch(const float* const* p_list, const float* const* q_list, int nseg,
                        int64_t p_nstride, int64_t q_nstride, float* grad, float* workspace,
                        int n_per_seg, int ca, int cb, int cb_total, int cb_off, int h, int w,
                        int accumulate, tg_stream_t stream, int cphase = 0, int set_p0 = 0,
                        int set_p1 = 0) {
  TG_REQUIRE(p_list && q_list && grad && workspace, TG_E_ARG, "wgrad3x3: null pointer");
  TG_REQUIRE(nseg >= 1 && nseg <= WG_MAXSEG, TG_E_ARG, "wgrad3x3: %d segments (1..%d)", nseg, WG_MAXSEG);
  TG_REQUIRE(n_per_seg > 0 && ca > 0 && cb > 0 && h > 0 && w > 0 && cb_off >= 0 &&
                 cb_off + cb <= cb_total, TG_E_SHAPE,
             "wgrad3x3: n=%dx%d ca=%d cb=%d (+%d of %d) h=%d w=%d", nseg, n_per_seg, ca, cb, cb_off,
             cb_total, h, w);
  TG_REQUIRE((long long)(ca > cb ? ca : cb) * h * w * 4 < (1ll << 31), TG_E_SHAPE,
             "wgrad3x3: one batch item must be < 2 GiB");
  if (ca <= 4 && !cphase) {
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
    hipLaunchKernelGGL(wgrad3x3_smallca_kernel, dim3((unsigned)(sa.nblk * sa.nbb)), dim3(256), 0, s, sa);
    int rc = check_launch("wgrad3x3_smallca");
    if (rc != TG_OK) return rc;
    const long long total = (long long)ca * cb * 9;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, workspace, grad,
                       sa.nblk, ca, cb, cb_total, cb_off, accumulate);
    return check_launch("wgrad_reduce");
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
  a.tiles_x = cdiv(w, WG_TW); a.tiles_y = cdiv(h, WG_R);
  a.ntiles = n * a.tiles_x * a.tiles_y;
  a.nab = cdiv(ca, 64); a.nbb = cdiv(cb, 64);
  a.nsplit = wgrad_nsplit(n, h, w, ca, cb_total);   // same value the workspace was sized with
  static const int maxwg = TG_LAB_ENV("TG_WGRAD_MAXWG", 0);     // lab: cap the K split (workgroups per channel block)
  if (maxwg > 0 && a.nsplit > maxwg) a.nsplit = maxwg;
  a.cphase = cphase; a.rowsets[0] = (unsigned char)set_p0; a.rowsets[1] = (unsigned char)set_p1;
  if (cphase) {
    TG_REQUIRE(cphase % 64 == 0 && cb == 4 * cphase && cb_off == 0 && set_p0 >= 0 && set_p0 <= 3 &&
                   set_p1 >= 0 && set_p1 <= 3, TG_E_ARG,
               "wgrad3x3: phased taps need cb = 4 phases of a multiple of 64 channels (cb=%d cphase=%d)", cb,
               cphase);
  }
  size_t lds = 2 * (size_t)(WG_A_FLOATS + WG_B_FLOATS) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_mfma_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  unsigned blocks = (unsigned)(a.nab * a.nbb * a.nsplit);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad3x3_mfma_kernel, dim3(blocks), dim3(256), lds, s, a);
  int rc = check_launch("wgrad3x3_mfma");
  if (rc != TG_OK) return rc;
  long long total = (long long)ca * cb * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, workspace,
                     grad, a.nsplit, ca, cb, cb_total, cb_off, accumulate);
  return check_launch("wgrad_reduce");
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
  return (size_t)nlayers * body_nsplit(nframes, n_per_frame, nlayers, c, h, w) * c * c * 9;
}

extern "C" int tg_wgrad3x3_body(const float* const* dz_bases, const float* const* act_bases,
                                int nframes, int64_t layer_stride, int nlayers,
                                float* const* grads, float* workspace, int n_per_frame, int c, int h, int w,
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
  size_t lds = 2 * (size_t)(WG_A_FLOATS + WG_B_FLOATS) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_mfma_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad3x3_mfma_kernel, dim3((unsigned)(nlayers * a.nab * a.nbb * a.nsplit)), dim3(256), lds, s, a);
  int rc = check_launch("wgrad3x3_body");
  if (rc != TG_OK) return rc;
  const long long total = (long long)c * c * 9;
  hipLaunchKernelGGL(wgrad_reduce_layers_kernel, dim3((unsigned)((total + 63) / 64), (unsigned)nlayers), dim3(256), 0, s,
                     workspace, gl, a.nsplit, c, c, accumulate);
  return check_launch("wgrad_reduce_layers");
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

extern "C" int tg_wgrad3x3_multi(const float* const* p_list, const float* const* q_list, int nseg,
                                 int64_t p_nstride, int64_t q_nstride, float* grad, float* workspace,
                                 int n_per_seg, int ca, int cb, int cb_total, int cb_off, int h,
                                 int w, int accumulate, tg_stream_t stream) {
  return wgrad_launch(p_list, q_list, nseg, p_nstride, q_nstride, grad, workspace, n_per_seg, ca, cb,
                      cb_total, cb_off, h, w, accumulate, stream);
}

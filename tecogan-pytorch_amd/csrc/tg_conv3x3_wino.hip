// 3x3 / stride 1 / pad 1 convolution in the Winograd F(2x2, 3x3) form on the fp32 matrix cores.
//
//   Y = A^T [ (G g G^T) .  (B^T d B) ] A        per 2x2 output tile and (oc, ic) pair
//
// 16 multiplies per tile instead of 36: 2.25x fewer MFMA passes than the direct form
// (tg_conv3x3_mfma.hip), all of them fp32 -- B and A hold only 0 / +-1, G only 0 / +-1 / +-1/2,
// so the transforms add no rounding beyond fp32 adds (measured through SRNet's 20 residual
// layers the result is as close to the fp64 truth as the direct fp32 form).
// This is the arithmetic the reference's conv layers (tecogan_nets.py:85-100,116; every
// nn.Conv2d(k3, s1, p1)) get from their GPU backend as well.
//
// Work decomposition (one workgroup = 4 waves):
//   * 16 tiles of one tile row (2 image rows x 32 pixels) x 64 output channels;
//     wave v owns output channels [16v, 16v+16) for all 16 Winograd positions:
//     16 accumulators of v_mfma_f32_16x16x4_f32 (M = 16 oc, N = 16 tiles, K = 4 ic).
//   * K loop over stages of 16 input channels: the raw 4 x 34 patch of every channel is staged
//     into LDS, each thread transforms one (channel, tile) 4x4 window (32 adds) into
//     V[ic][tile][16 positions] in LDS, all waves read V as the MFMA B operand.
//   * the transformed weights U[p][oc][ic] (packed by wino_pack_kernel in exactly the order a
//     lane consumes them) go from L2 straight into registers -- each wave owns a distinct slice,
//     LDS would add nothing.
//   * the inverse transform runs in registers: a lane holds all 16 positions of its
//     (4 oc x 1 tile), adds bias / activation / residual and stores 2x2 pixels per channel.
// 670 workgroups for 134 x 320 (2.6 per CU, all resident at once, 3 waves per SIMD).  Measured: 23-24 us
// per 64 -> 64 layer against 31 us for the direct form; the matrix pipe is ~35 % busy -- about 10 us of
// a launch are fixed cost (launch, first loads, the final 11 MB of stores) that identical workgroups
// pay in lockstep; several clips per launch overlap it (DESIGN.md sections 4 and 10).
#include <stdlib.h>

#include "tg_common.h"

#ifndef TG_CHAIN_FENCES
#define TG_CHAIN_FENCES 0   // 1: release / acquire fences around the flag of the chained launch (measurement build)
#endif
#ifndef TG_WINO_LAB
#define TG_WINO_LAB 0   // 1: ablation switches of tools/wino_abl.sh (TG_WINO_ABL) compiled in
#endif
#define WABL(bit) (TG_WINO_LAB && (a.abl & (bit)))
#if !TG_LAB && TG_WINO_LAB
#error "tg_conv3x3_wino.hip: TG_WINO_LAB needs -DTG_LAB=1 (lab builds only; csrc/build.sh refuses it for the in-tree library)"
#endif

namespace tg {

struct WinoArgs {
  const float* x;
  const float* x2;     // channels [c1, cin) (or null)
  const float* u;      // packed transformed weights
  const float* bias;
  const float* res;
  const float* mask;   // optional ReLU-backward mask applied last (see tg_conv3x3_fwd_masked)
  float* y;
  long long x_ns, x2_ns, res_ns, mask_ns, y_ns;
  int c1, cin, cout, h, w, act;
  int tiles_x, tiles_y, nstage, nocg, nocb;
  int nblocks;         // > 0: XCD-banded block order
  int vec_ok;          // float2 stores allowed (w even, 8-byte aligned planes)
  int abl;             // lab builds only (TG_WINO_LAB, env TG_WINO_ABL): 1 no weight loads, 2 no input loads, 4 no stores, 16 no MFMA, 32 weights from L1, 64 empty launch, 128 no main loop
};

#ifndef W_BRANCHY_U
#define W_BRANCHY_U 0   // 1: the round 2-4 form of the weight prefetch, for A/B
#endif
constexpr int W_ICS = 16;             // input channels per stage
constexpr int W_RS = 40;              // LDS row stride of the raw patch (floats); 4 rows = 160 = 32 mod 64 banks
constexpr int W_ICSTR = 4 * W_RS;
constexpr int W_VS = 20;              // floats per (ic, tile) in V: 16 positions + 4 pad -> 16-byte reads of 16 lanes hit 64 distinct banks

// U layout: [K step of 4 ic][oc block of 16][j 0..3][lane 0..63][e 0..3]; register p = 4j + e of a
// lane holds U[p][oc = 16*ocb + (lane & 15)][ic = 4*kstep + (lane >> 4)], U[p = 4a + b] =
// (G g G^T)[a][b].  Zero padded in ic (to a multiple of 16) and oc (to a multiple of 64).
// transposed: 0 = OIHW weights, 2 = data gradient of a conv (swap channel roles, rotate 180).
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout,
                                 int nchunk, int nocb, int transposed) {
  const int total = nchunk * nocb * 4 * 64 * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 3, lane = (i >> 2) & 63, j = (i >> 8) & 3;
    const int rest = i >> 10;
    const int ocb = rest % nocb, chunk = rest / nocb;
    const int p = 4 * j + e;
    const int oc = 16 * ocb + (lane & 15), ic = 4 * chunk + (lane >> 4);
    float v = 0.f;
    if (oc < cout && ic < cin) {
      float g[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
          g[a][b] = transposed == 2 ? w[((size_t)ic * cout + oc) * 9 + (8 - (a * 3 + b))]
                                    : w[((size_t)oc * cin + ic) * 9 + a * 3 + b];
      // rows of G: [1,0,0], [.5,.5,.5], [.5,-.5,.5], [0,0,1]
      const int pi = p >> 2, pj = p & 3;
      float col[3];   // (G g)[pi][b]
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        col[b] = pi == 0 ? g[0][b]
               : pi == 1 ? 0.5f * ((g[0][b] + g[1][b]) + g[2][b])
               : pi == 2 ? 0.5f * ((g[0][b] - g[1][b]) + g[2][b])
                         : g[2][b];
      }
      v = pj == 0 ? col[0]
        : pj == 1 ? 0.5f * ((col[0] + col[1]) + col[2])
        : pj == 2 ? 0.5f * ((col[0] - col[1]) + col[2])
                  : col[2];
    }
    out[i] = v;
  }
}

template <bool COH> __device__ __forceinline__ float2 ld2(const float* p) {
  if constexpr (COH) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (unsigned)v), __builtin_bit_cast(float, (unsigned)(v >> 32)));
  } else {
    return *reinterpret_cast<const float2*>(p);
  }
}
template <bool COH> __device__ __forceinline__ float ld1(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH> __device__ __forceinline__ void st2(float* p, float v0, float v1) {
  if constexpr (COH) {
    const unsigned long long v = (unsigned long long)__builtin_bit_cast(unsigned, v0) | ((unsigned long long)__builtin_bit_cast(unsigned, v1) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *reinterpret_cast<float2*>(p) = make_float2(v0, v1);
  }
}
template <bool COH> __device__ __forceinline__ void st1(float* p, float v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// One workgroup's tile.  COH (chained launches, below): activations move with agent-scope (sc1)
// loads / stores -- they bypass the XCD's non-coherent L2 copy -- because producer and consumer
// workgroups of consecutive layers run in the same launch, possibly behind different L2s.
constexpr int AUX_SC1 = 16;       // cache-policy bit of the buffer instructions on gfx940+

// TR: tile rows of the workgroup's 16 Winograd tiles (TR x 16/TR tiles = 2 TR x 32/TR pixels).  1: one tile row of 32
// pixels (the form everything was tuned on); 2 / 4: 4 x 16 / 8 x 8 pixels for maps narrower than a 32-pixel tile
// (FNet's 17x40 and 34x80 maps at inference, its 16x16 maps on the training frames: 17-50 % of every 32-pixel-wide
// tile lay outside the map).  Every output's arithmetic and its order are the same in all three: bit-identical.
template <int TR> struct WGeo3 {
  static constexpr int TC = 16 / TR;                 // tile columns
  static constexpr int PR = 2 * TR + 2, PC = 2 * TC + 2;   // raw patch rows / columns per input channel
  static constexpr int PE = PR * PC;
  static constexpr int RS = TR == 1 ? W_RS : (TR == 2 ? 24 : 12);   // LDS row stride: the float2 reads of 32 lanes
                                                                     // (2 channels x 16 tiles) hit 64 distinct banks
  static_assert(PR * RS <= W_ICSTR, "patch of one channel fits the channel stride");
};

template <bool COH, int TR = 1>
__device__ __forceinline__ void wino_tile(const WinoArgs& a, int tx, int ty, int ocg, int n) {
  using G = WGeo3<TR>;
  __shared__ __attribute__((aligned(16))) float s_raw[W_ICS * W_ICSTR];   // 10 KB
  __shared__ __attribute__((aligned(16))) float s_v[2 * W_ICS * 16 * W_VS];   // 2 x 20 KB: [buffer][ic][tile][16 positions + pad]

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int x0 = tx * (2 * G::TC), y0 = ty * (2 * TR);
  const int hw = a.h * a.w;

  // ---- raw patch staging: element e = t + 256 k = (ic, row, col) of the 16 x 4 x 34 patch ------
  constexpr int RAW_ELEMS = W_ICS * G::PE;
  constexpr int RAW_PER_T = (RAW_ELEMS + 255) / 256;   // 9
  // byte offset of the element inside a channel plane, or OOB: the buffer bounds check then
  // returns 0 for the zero padding, for channels past cin (K padded to a multiple of 16) and
  // for the channels that belong to the other source tensor
  constexpr unsigned OOB = 0x80000000u;
  unsigned roff[RAW_PER_T];
#pragma unroll
  for (int k = 0; k < RAW_PER_T; ++k) {
    const int e = t + 256 * k;
    const int ic = e / G::PE, rem = e - ic * G::PE, r = rem / G::PC, c = rem - r * G::PC;
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    roff[k] = (e < RAW_ELEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w)
                  ? (unsigned)(ic * hw + gy * a.w + gx) * 4u : OOB;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (size_t)n * a.x_ns), 0, (unsigned)a.c1 * hw * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x2 ? a.x2 + (size_t)n * a.x2_ns : a.x), 0,
      a.x2 ? (unsigned)(a.cin - a.c1) * hw * 4u : 0u, 0x00020000);
  const bool dual = a.x2 != nullptr;
  auto load_raw = [&](int s, float (&reg)[RAW_PER_T]) {
    if (WABL(2) && s > 0) return;
    const unsigned so = (unsigned)(s * W_ICS) * hw * 4u;
#pragma unroll
    for (int k = 0; k < RAW_PER_T; ++k) {
      float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(roff[k] + so), 0, COH ? AUX_SC1 : 0));
      if (dual)   // offsets below c1 wrap to > 2^31 and read 0
        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                 rx2, (int)(roff[k] + so - (unsigned)a.c1 * hw * 4u), 0, COH ? AUX_SC1 : 0));
      reg[k] = v;
    }
  };
  auto store_raw = [&](const float (&reg)[RAW_PER_T]) {
#pragma unroll
    for (int k = 0; k < RAW_PER_T; ++k) {
      const int e = t + 256 * k;
      const int ic = e / G::PE, rem = e - ic * G::PE, r = rem / G::PC, c = rem - r * G::PC;
      if (e < RAW_ELEMS) s_raw[ic * W_ICSTR + r * G::RS + c] = reg[k];
    }
  };

  // ---- transformed weights: 4 x 16-byte loads per K step, perfectly coalesced ---------------
  const f32x4* ug = reinterpret_cast<const f32x4*>(a.u) + ((size_t)(ocg * 4 + wv) * 4) * 64 + l;
  const size_t ustep = (size_t)a.nocb * 4 * 64;
  const int ktotal = 4 * a.nstage;
  // one half (positions 8*half .. 8*half+7) of the weights of K step `kstep`
  auto load_uh = [&](int kstep, int half, f32x4 (&u)[4]) {
#if W_BRANCHY_U
    if (kstep >= ktotal || (WABL(1) && kstep > 1)) return;
#else
    // BRANCH-FREE (round 5): behind `if (kstep < ktotal)` the compiler's s_waitcnt insertion merges both paths at the
    // join assuming the FEWER loads in flight, and every K step waited `vmcnt(1) / vmcnt(0)` -- for the half-block
    // requested a moment ago as well as for its own.  Past the last K step the last block is requested again (an L1
    // hit, never used); the waits then carry the exact counts (vmcnt(7) .. (4) in the ISA).
    if (WABL(1) && kstep > 1) return;
    kstep = kstep < ktotal ? kstep : ktotal - 1;
#endif
    if (WABL(32)) kstep &= 1;
    const f32x4* p = ug + (size_t)kstep * ustep + half * 128;
    u[2 * half] = p[0];
    u[2 * half + 1] = p[64];
  };

  f32x4 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transform assignment: thread -> (ic = t >> 4, tile = t & 15)
  constexpr int VBUF4 = W_ICS * 16 * W_VS / 4;          // one V buffer in 16-byte units
  const float* traw = s_raw + (t >> 4) * W_ICSTR + 2 * ((t & 15) / G::TC) * G::RS + 2 * ((t & 15) % G::TC);
  f32x4* tv = reinterpret_cast<f32x4*>(s_v + t * W_VS);                                        // 4 x 16 bytes per buffer
  const f32x4* bv = reinterpret_cast<const f32x4*>(s_v + ((l >> 4) * 16 + (l & 15)) * W_VS);   // + ks * 4*16*W_VS floats

  // 8 MFMAs: positions 8*half .. 8*half+7 of K step ks (B operand from V[buf])
  auto mfma8 = [&](const f32x4 (&u)[4], int buf, int ks, int half) {
    f32x4 bq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bq[j] = bv[buf * VBUF4 + ks * (4 * 16 * W_VS / 4) + 2 * half + j];
    if (WABL(16)) { acc[ks][0] += bq[0][0] + bq[1][1] + u[2 * half][0] + u[2 * half + 1][0]; return; }
#pragma unroll
    for (int p = 0; p < 8; ++p)
      acc[8 * half + p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[2 * half + (p >> 2)][p & 3], bq[p >> 2][p & 3],
                                                               acc[8 * half + p], 0, 0, 0);
  };
  // K step k of the running count (ks = k & 3 inside its stage): as soon as a half of its weight
  // block has been consumed, the same half of K step k+2 is requested into it -- 1.5 K steps
  // (24 MFMAs of this wave, ~3x that with three waves per SIMD) of distance on two blocks
  auto kstep = [&](int k, f32x4 (&u)[4], int buf, auto&& between) {
    mfma8(u, buf, k & 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_uh(k + 2, 0, u);
    between();
    mfma8(u, buf, k & 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_uh(k + 2, 1, u);
  };
  // B^T d B of this thread's 4x4 window, two output rows per call (hp = 0: rows 0, 1 from raw
  // rows 0..2; hp = 1: rows 2, 3 from raw rows 1..3) -> V[buf]
  auto transform_half = [&](int buf, int hp) {
    float d[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float2 p0 = *reinterpret_cast<const float2*>(traw + (r + hp) * G::RS);
      const float2 p1 = *reinterpret_cast<const float2*>(traw + (r + hp) * G::RS + 2);
      d[r][0] = p0.x; d[r][1] = p0.y; d[r][2] = p1.x; d[r][3] = p1.y;
    }
    float qa[4], qb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qa[c] = hp == 0 ? d[0][c] - d[2][c] : d[1][c] - d[0][c];    // rows 0 / 2 of B^T d
      qb[c] = hp == 0 ? d[1][c] + d[2][c] : d[0][c] - d[2][c];    // rows 1 / 3
    }
    tv[buf * VBUF4 + 2 * hp] = f32x4{qa[0] - qa[2], qa[1] + qa[2], qa[2] - qa[1], qa[1] - qa[3]};
    tv[buf * VBUF4 + 2 * hp + 1] = f32x4{qb[0] - qb[2], qb[1] + qb[2], qb[2] - qb[1], qb[1] - qb[3]};
  };

  // epilogue geometry (needed early: the residual is fetched under the last stage's MFMAs)
  const int ox = x0 + 2 * ((l & 15) % G::TC);
  const int oy0 = y0 + 2 * ((l & 15) / G::TC);      // first output row of the lane's tile
  const int oc_base = ocg * 64 + wv * 16 + 4 * (l >> 4);
  const float* rn = a.res ? a.res + (size_t)n * a.res_ns : nullptr;
  const bool res_pre = rn && a.vec_ok;

  // ---- prologue: stage 0 transformed, stage 1's patch in LDS (not yet published) -------------
  float rawreg[RAW_PER_T];
  f32x4 u0[4], u1[4];
  const int last = a.nstage - 1;
  load_raw(0, rawreg);
  load_uh(0, 0, u0); load_uh(0, 1, u0);
  load_uh(1, 0, u1); load_uh(1, 1, u1);
  store_raw(rawreg);
  __syncthreads();
  if (last > 0) load_raw(1, rawreg);
  transform_half(0, 0);
  transform_half(0, 1);
  __syncthreads();                           // V[0] visible, raw patch free
  if (last > 0) store_raw(rawreg);

  // ---- main loop: the MFMAs of stage s (64 per wave: 4 K steps x 16 positions) run with the
  // input transform of stage s+1 and the global loads of stage s+2 in their shadow.  (The
  // scheduling fences keep the compiler from hoisting all operand reads to the top, which
  // would cost the third wave per SIMD.)
  auto nothing = [] {};
  for (int s = 0; s < (WABL(128) ? 0 : last); ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    kstep(4 * s + 0, u0, cur, nothing);
    __syncthreads();                         // patch of stage s+1 visible
    if (s + 2 <= last) load_raw(s + 2, rawreg);    // (a branch-free form of THIS prefetch measured +-0: its loads are consumed
                                                   // in the same iteration, round 5)
    kstep(4 * s + 1, u1, cur, [&] { transform_half(nxt, 0); });
    kstep(4 * s + 2, u0, cur, [&] { transform_half(nxt, 1); });
    kstep(4 * s + 3, u1, cur, nothing);
    __syncthreads();                         // V[nxt] visible, V[cur] and the patch free
    if (s + 2 <= last) store_raw(rawreg);
  }
  // ---- last stage: the residual tile is fetched under its MFMAs
  float2 rpre[4][2];
  {
    const int cur = last & 1;
    kstep(4 * last + 0, u0, cur, nothing);
    kstep(4 * last + 1, u1, cur, nothing);
    kstep(4 * last + 2, u0, cur, nothing);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rpre[r][i] = make_float2(0.f, 0.f);
        if (res_pre && oc_base + r < a.cout && ox < a.w && oy0 + i < a.h)
          rpre[r][i] = ld2<COH>(rn + (size_t)(oc_base + r) * hw + (size_t)(oy0 + i) * a.w + ox);
      }
    kstep(4 * last + 3, u1, cur, nothing);
  }

  // ---- inverse transform A^T m A, epilogue -------------------------------------------------
  const float slope = act_slope(a.act);
  float* yn = a.y + (size_t)n * a.y_ns;
  const float* mn = a.mask ? a.mask + (size_t)n * a.mask_ns : nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oc = oc_base + r;
    float sr[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sr[0][j] = (acc[0 + j][r] + acc[4 + j][r]) + acc[8 + j][r];
      sr[1][j] = (acc[4 + j][r] - acc[8 + j][r]) - acc[12 + j][r];
    }
    if (oc >= a.cout || ox >= a.w || (WABL(4) && sr[0][0] != 123.f)) continue;
    const float bz = a.bias ? a.bias[oc] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = oy0 + i;
      if (oy >= a.h) continue;
      float v0 = ((sr[i][0] + sr[i][1]) + sr[i][2]) + bz;
      float v1 = ((sr[i][1] - sr[i][2]) - sr[i][3]) + bz;
      if (a.act == TG_ACT_TANH24) { v0 = apply_act(v0, a.act); v1 = apply_act(v1, a.act); }
      else {      // max(x, 0) + slope * min(x, 0): the piecewise-linear activations without v_cndmask (see tg_conv3x3_wino_res.hip)
        v0 = __builtin_fmaxf(v0, slope * v0);      // (round 6: 2 VALU instead of max(x, 0) + slope * min(x, 0); slope in [0, 1])
        v1 = __builtin_fmaxf(v1, slope * v1);
      }
      const size_t o = (size_t)oc * hw + (size_t)oy * a.w + ox;
      const bool two = ox + 1 < a.w;
      if (a.vec_ok) {
        if (rn) { v0 += rpre[r][i].x; v1 += rpre[r][i].y; }
        if (mn) { const float2 mm = *reinterpret_cast<const float2*>(mn + o); v0 = mm.x > 0.f ? v0 : 0.f; v1 = mm.y > 0.f ? v1 : 0.f; }
        st2<COH>(yn + o, v0, v1);
      } else {
        if (rn) { v0 += ld1<COH>(rn + o); if (two) v1 += ld1<COH>(rn + o + 1); }
        if (mn) { v0 = mn[o] > 0.f ? v0 : 0.f; if (two) v1 = mn[o + 1] > 0.f ? v1 : 0.f; }
        st1<COH>(yn + o, v0);
        if (two) st1<COH>(yn + o + 1, v1);
      }
    }
  }
}

template <int TR>
__global__ __launch_bounds__(256, 3) void conv3x3_wino_kernel(WinoArgs a) {
  if (WABL(64)) return;                      // empty launch
  int b = blockIdx.x;
  if (a.nblocks > 0) {      // XCD x gets the contiguous band of tiles [x*per, (x+1)*per)
    const int per = (a.nblocks + 7) >> 3;
    b = (b & 7) * per + (b >> 3);
    if (b >= a.nblocks) return;
  }
  const int tx = __builtin_amdgcn_readfirstlane(b % a.tiles_x); b /= a.tiles_x;
  const int ty = __builtin_amdgcn_readfirstlane(b % a.tiles_y); b /= a.tiles_y;
  const int ocg = __builtin_amdgcn_readfirstlane(b % a.nocg);
  const int n = __builtin_amdgcn_readfirstlane(b / a.nocg);
  wino_tile<false, TR>(a, tx, ty, ocg, n);
}

// ---- several dependent layers in ONE launch ----------------------------------------------------
// Every launch of the kernel above pays ~10 us that nothing overlaps (launch, first loads, the
// final stores: all workgroups are resident at once and walk the same timeline).  Here the
// workgroups of layer k+1 are dispatched right behind those of layer k (same grid) and start as
// soon as the 3x3 tile neighbourhood they read has been written:
//   * flag[layer][tile] = epoch is stored (agent scope) after the tile's stores have left the
//     workgroup; a consumer polls the <= 9 flags of its neighbourhood (s_sleep between polls);
//   * that neighbourhood also covers the write-after-read hazards of SRNet's buffers (in-place
//     residual sum, two ping-pong tensors): whoever still reads the old content of tile T is one
//     of the tiles T's writer waits for;
//   * activations use sc1 loads / stores (COH above): no L2 invalidation between layers, the
//     weights stay cached.
// Memory model: the producer's data stores are agent-scope write-through (sc1); `s_waitcnt 0`
// returns once every one of them has been acknowledged at the coherence point, the workgroup
// barrier extends that to all four waves, and only then is the flag stored -- the flag therefore
// cannot become visible before the data (release).  The consumer's data loads are sc1 as well:
// they never hit a stale line of its own XCD's L2 / L1, and they are issued after the flag was
// seen and a workgroup barrier (acquire).  No cache invalidation is needed on either side.
// HARDWARE REQUIREMENT (forward progress): a workgroup only ever waits for workgroups with a
// smaller blockIdx.  The launch relies on the dispatcher starting workgroups in blockIdx order
// per XCD (what every CDNA part does, but HIP does not promise it).  The kernel is therefore
// FAIL-SAFE rather than trusting: a poll limit (~0.2 s) ends the wait, the workgroup counts a
// fault in *err (system scope: the plan points it at pinned host memory) and carries on so the
// launch always terminates; the owner of the plan sees the fault on its next call, reports
// TG_E_HIP and falls back to one launch per layer for good (tg_api.hip: chain_poll).
struct WinoLayer {
  const float *x, *x2, *u, *bias, *res;
  float* y;
  long long x_ns, x2_ns, res_ns, y_ns;
  int c1, cin, act, nstage;
};
constexpr int W_MAX_CHAIN = 24;
struct WinoChainArgs {
  WinoLayer L[W_MAX_CHAIN];
  int nlayer, bpl;           // blocks per layer in the grid (tiles rounded up to a multiple of 8)
  int cout, h, w, tiles_x, tiles_y, ntile, vec_ok;
  int* flags;                // [nlayer][ntile]
  int* err;                  // fault counter: device memory (flag buffer tail) or pinned host memory (plan)
  unsigned epoch;
  int poll_limit;            // polls of ~64 cycles before a waiter gives up; < 0: every waiter faults at once (fault-injection tests)
};

template <bool COHX>
__global__ __launch_bounds__(256, 3) void conv3x3_wino_chain_kernel(WinoChainArgs c) {
  const int layer = __builtin_amdgcn_readfirstlane((int)blockIdx.x / c.bpl);
  int b = (int)blockIdx.x - layer * c.bpl;
  {
    const int per = c.bpl >> 3;              // XCD x gets the band of tiles [x*per, (x+1)*per) of every layer
    b = (b & 7) * per + (b >> 3);
    if (b >= c.ntile) return;
  }
  const int tile = b;
  const int tx = __builtin_amdgcn_readfirstlane(b % c.tiles_x); b /= c.tiles_x;
  const int ty = __builtin_amdgcn_readfirstlane(b % c.tiles_y);
  const int n = __builtin_amdgcn_readfirstlane(b / c.tiles_y);
  const WinoLayer& L = c.L[layer];
  WinoArgs a;
  a.x = L.x; a.x2 = L.x2; a.u = L.u; a.bias = L.bias; a.res = L.res; a.mask = nullptr; a.y = L.y;
  a.x_ns = L.x_ns; a.x2_ns = L.x2_ns; a.res_ns = L.res_ns; a.mask_ns = 0; a.y_ns = L.y_ns;
  a.c1 = L.c1; a.cin = L.cin; a.cout = c.cout; a.h = c.h; a.w = c.w; a.act = L.act;
  a.tiles_x = c.tiles_x; a.tiles_y = c.tiles_y; a.nstage = L.nstage; a.nocg = 1; a.nocb = 4;
  a.nblocks = 0; a.vec_ok = c.vec_ok; a.abl = 0;
  if (layer > 0) {                            // wait for the producers of the 3x3 neighbourhood
    const int t = threadIdx.x;
    if (t < 9) {
      const int ny = ty - 1 + t / 3, nx = tx - 1 + t % 3;
      if (ny >= 0 && ny < c.tiles_y && nx >= 0 && nx < c.tiles_x) {
        const unsigned* f = reinterpret_cast<const unsigned*>(c.flags) + (size_t)(layer - 1) * c.ntile + (n * c.tiles_y + ny) * c.tiles_x + nx;
        int polls = 0;
        bool fault = c.poll_limit < 0;
        while (!fault && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != c.epoch) {
          __builtin_amdgcn_s_sleep(4);
          fault = ++polls > c.poll_limit;
        }
        if (fault) __hip_atomic_fetch_add(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#if TG_CHAIN_FENCES
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // measured variant, see the note below
#endif
    }
    __syncthreads();
  }
  wino_tile<COHX>(a, tx, ty, 0, n);
  __builtin_amdgcn_s_waitcnt(0);              // this wave's stores have been acknowledged
  __syncthreads();
  // Memory-model note.  The formally portable hand-over is
  //   producer: data stores; fence(release, agent); relaxed flag store
  //   consumer: relaxed flag poll; fence(acquire, agent); data loads
  // On gfx950 the release fence is `buffer_wbl2 sc1; s_waitcnt vmcnt(0)` and the acquire fence `buffer_inv sc1`
  // (MI355X_MICROARCH.md): with agent-scope (sc1, write-through) data stores that were waited for there is
  // nothing left to write back, and with sc1 data loads there is no L1 line to invalidate -- the fences
  // add nothing but their cost.  Built with -DTG_CHAIN_FENCES=1 the launch measured +X % (tools/chain_fence_lab.sh;
  // DESIGN.md section 10c has the number), so the shipped form keeps sc1 stores + acknowledged waitcnt +
  // relaxed agent-scope flag, and the property is held by tests/test_hip_soak.py (200 launches per production
  // shape on fresh data under a concurrent memory stream, bit-compared with per-layer launches).
#if TG_CHAIN_FENCES
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#endif
  if (threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<unsigned*>(c.flags) + (size_t)layer * c.ntile + tile, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tg

using namespace tg;

extern "C" int64_t tg_conv3x3_wino_packed_floats(int cin, int cout) {
  if (cin <= 0 || cout <= 0) return -1;
  const int64_t nchunk = 4 * ((cin + 15) / 16), nocb = 4 * ((cout + 63) / 64);
  return nchunk * nocb * 4 * 64 * 4;
}

// Measured on MI355X (tools/wino_lab.py): at 134x320 / 64 -> 64 the Winograd form takes 0.70x the
// time of the direct kernel, at 67x160 (170 workgroups) 0.75x; with 128 workgroups and fewer (2 x 64 x 64
// training frames, FNet's low-resolution middle at batch 1: 33x80, 16x40) the one-shot / split-K
// kernels win by 5-30 %.
// TG_CONV_WINO=0/1 overrides (lab / A-B).
extern "C" int tg_conv3x3_prefers_wino(int n, int cin, int cout, int h, int w) {
  static const int env = [] { const char* e = getenv("TG_CONV_WINO"); return e && *e ? atoi(e) : -1; }();
  if (env == 0) return 0;
  if (n <= 0 || cin < 16 || cout <= 0 || cout % 64 != 0 || h < 2 || w < 2) return 0;
  const long long wgs = (long long)cdiv(w, 32) * cdiv(h, 2) * (cout / 64) * n;
  // (narrow images waste the columns of a 2 x 32 pixel workgroup, but those of the direct kernel's
  // 32-pixel tiles just the same: at 16x16 x 36 frames the Winograd form still takes 0.6x the time)
  return env == 1 ? 1 : (wgs >= 160 ? 1 : 0);
}

extern "C" int tg_pack_conv3x3_wino(const float* w, float* out, int cin, int cout, int transposed,
                                    tg_stream_t stream) {
  TG_REQUIRE(w && out, TG_E_ARG, "pack_conv3x3_wino: null pointer");
  TG_REQUIRE(cin > 0 && cout > 0 && (transposed == 0 || transposed == 2), TG_E_ARG,
             "pack_conv3x3_wino: cin=%d cout=%d transposed=%d (0 or 2)", cin, cout, transposed);
  const int nchunk = 4 * cdiv(cin, 16), nocb = 4 * cdiv(cout, 64);
  const int total = nchunk * nocb * 4 * 64 * 4;
  hipLaunchKernelGGL(wino_pack_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w, out, cin, cout, nchunk, nocb, transposed);
  return check_launch("pack_conv3x3_wino");
}

namespace tg {
int conv3x3_wino_launch(const float* x, int64_t x_ns, int c1, const float* x2, int64_t x2_ns, const float* u,
                        const float* bias, const float* res, int64_t res_ns, const float* mask,
                        int64_t mask_ns, float* y, int64_t y_ns, int n, int cin, int cout, int h, int w,
                        int act, tg_stream_t stream) {
  WinoArgs a{};
  a.x = x; a.x2 = x2; a.u = u; a.bias = bias; a.res = res; a.mask = mask; a.y = y;
  a.x_ns = x_ns; a.x2_ns = x2_ns; a.res_ns = res_ns; a.mask_ns = mask_ns; a.y_ns = y_ns;
  a.c1 = x2 ? c1 : cin; a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = act;
  // tile arrangement of a workgroup's 16 tiles: the one that covers the map with the fewest workgroups
  // (ties: the widest -- longest coalesced rows); lab builds: TG_WINO_TR forces one
  int tr = 1;
  {
    long long best = (long long)cdiv(w, 32) * cdiv(h, 2);
    for (int cand = 2; cand <= 4; cand *= 2) {
      const long long c = (long long)cdiv(w, 32 / cand) * cdiv(h, 2 * cand);
      if (c < best) { best = c; tr = cand; }
    }
    static const int tr_env = TG_LAB_ENV("TG_WINO_TR", 0);
    if (tr_env == 1 || tr_env == 2 || tr_env == 4) tr = tr_env;
  }
  a.tiles_x = cdiv(w, 32 / tr); a.tiles_y = cdiv(h, 2 * tr);
  a.nstage = cdiv(cin, 16); a.nocg = cdiv(cout, 64); a.nocb = 4 * a.nocg;
  auto al8 = [](const void* p, int64_t ns) { return ((uintptr_t)p % 8) == 0 && ns % 2 == 0; };
  a.vec_ok = (w % 2 == 0) && ((int64_t)h * w) % 2 == 0 && al8(y, y_ns) && (!res || al8(res, res_ns)) &&
             (!mask || al8(mask, mask_ns));
  static const int abl_env = TG_LAB_ENV("TG_WINO_ABL", 0);
  a.abl = abl_env;
  const long long blocks = (long long)a.tiles_x * a.tiles_y * a.nocg * n;
  static const int xcd_env = TG_LAB_ENV("TG_WINO_XCD", -1);   // lab
  const bool xcd = xcd_env >= 0 ? xcd_env != 0 : blocks >= 512;
  a.nblocks = xcd ? (int)blocks : 0;
  const unsigned grid = xcd ? (unsigned)(8 * ((blocks + 7) / 8)) : (unsigned)blocks;
  if (tr == 1) hipLaunchKernelGGL(conv3x3_wino_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (tr == 2) hipLaunchKernelGGL(conv3x3_wino_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv3x3_wino_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("conv3x3_wino");
}
}  // namespace tg

extern "C" int tg_conv3x3_wino_fwd(const float* x, int64_t x_nstride, int c1, const float* x2,
                                   int64_t x2_nstride, const float* u_packed, const float* bias,
                                   const float* res, int64_t res_nstride, const float* relu_mask,
                                   int64_t mask_nstride, float* y, int64_t y_nstride, int n, int cin,
                                   int cout, int h, int w, int act, tg_stream_t stream) {
  TG_REQUIRE(x && u_packed && y, TG_E_ARG, "conv3x3_wino: null pointer");
  TG_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, TG_E_SHAPE, "conv3x3_wino: n=%d cin=%d cout=%d h=%d w=%d",
             n, cin, cout, h, w);
  TG_REQUIRE(!x2 || (c1 > 0 && c1 < cin), TG_E_ARG, "conv3x3_wino: c1=%d of cin=%d", c1, cin);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_TANH24, TG_E_ARG, "conv3x3_wino: act=%d", act);
  TG_REQUIRE((long long)cin * h * w < (1ll << 29), TG_E_SHAPE, "conv3x3_wino: image too large");
  return conv3x3_wino_launch(x, x_nstride, c1, x2, x2_nstride, u_packed, bias, res, res_nstride, relu_mask,
                             mask_nstride, y, y_nstride, n, cin, cout, h, w, act, stream);
}

#if TG_WINO_LAB
// lab only (tools/wino_split_probe.py): `layers` dependent 64->64 layers on ONE stream, or the same
// on two independent half-height chains whose launches alternate between two streams (all enqueued
// from C so that the host is not the bottleneck)
extern "C" int tg_lab_wino_chains(float* a0, float* c0, float* a1, float* c1, const float* u, const float* bias,
                                  int layers, int h, int w, tg_stream_t s0, tg_stream_t s1, int two) {
  float *pa[2] = {a0, a1}, *pc[2] = {c0, c1};
  tg_stream_t st[2] = {s0, s1};
  const int64_t ns = (int64_t)64 * h * w;
  for (int l = 0; l < layers; ++l)
    for (int k = 0; k < (two ? 2 : 1); ++k) {
      int rc = conv3x3_wino_launch(pa[k], ns, 64, nullptr, 0, u, bias, nullptr, 0, nullptr, 0, pc[k], ns, 1, 64, 64, h, w,
                                   TG_ACT_RELU, st[k]);
      if (rc != TG_OK) return rc;
      float* t = pa[k]; pa[k] = pc[k]; pc[k] = t;
    }
  return TG_OK;
}
#endif

extern "C" int64_t tg_conv3x3_wino_chain_flag_ints(int n_layers, int n, int h, int w) {
  if (n_layers <= 0 || n_layers > W_MAX_CHAIN || n <= 0 || h <= 0 || w <= 0) return -1;
  return (int64_t)n_layers * n * cdiv(h, 2) * cdiv(w, 32) + 16;     // + the error counter (last 16 ints)
}

namespace tg {
int conv3x3_wino_chain_launch(const tg_wino_layer* layers, int n_layers, int n, int cout, int h, int w,
                              int32_t* flags, int32_t* err, unsigned epoch, int poll_limit, tg_stream_t stream) {
  TG_REQUIRE(layers && flags && err, TG_E_ARG, "conv3x3_wino_chain: null pointer");
  TG_REQUIRE(n_layers >= 1 && n_layers <= W_MAX_CHAIN, TG_E_ARG, "conv3x3_wino_chain: %d layers (1..%d)", n_layers, W_MAX_CHAIN);
  TG_REQUIRE(n > 0 && h > 0 && w > 0 && cout > 0 && cout <= 64, TG_E_SHAPE,
             "conv3x3_wino_chain: n=%d cout=%d (<= 64: one output-channel group) h=%d w=%d", n, cout, h, w);
  TG_REQUIRE(epoch != 0, TG_E_ARG, "conv3x3_wino_chain: epoch 0 is the cleared state of the flags");
  WinoChainArgs c{};
  c.nlayer = n_layers; c.cout = cout; c.h = h; c.w = w;
  c.tiles_x = cdiv(w, 32); c.tiles_y = cdiv(h, 2);
  const long long ntile = (long long)c.tiles_x * c.tiles_y * n;
  TG_REQUIRE(ntile * n_layers < (1ll << 30), TG_E_SHAPE, "conv3x3_wino_chain: too many tiles");
  c.ntile = (int)ntile;
  c.bpl = (int)(8 * ((ntile + 7) / 8));
  c.flags = flags; c.err = err; c.epoch = epoch; c.poll_limit = poll_limit;
  bool vec = (w % 2 == 0) && ((int64_t)h * w) % 2 == 0;
  auto al8 = [](const void* p, int64_t ns) { return ((uintptr_t)p % 8) == 0 && ns % 2 == 0; };
  for (int i = 0; i < n_layers; ++i) {
    const tg_wino_layer& l = layers[i];
    TG_REQUIRE(l.x && l.u_packed && l.y, TG_E_ARG, "conv3x3_wino_chain: layer %d: null pointer", i);
    TG_REQUIRE(l.cin > 0 && (!l.x2 || (l.c1 > 0 && l.c1 < l.cin)), TG_E_ARG, "conv3x3_wino_chain: layer %d: cin=%d c1=%d", i, l.cin, l.c1);
    TG_REQUIRE((long long)l.cin * h * w < (1ll << 29), TG_E_SHAPE, "conv3x3_wino_chain: layer %d too large", i);
    TG_REQUIRE(l.act >= TG_ACT_NONE && l.act <= TG_ACT_TANH24, TG_E_ARG, "conv3x3_wino_chain: layer %d: act=%d", i, l.act);
    WinoLayer& d = c.L[i];
    d.x = l.x; d.x2 = l.x2; d.u = l.u_packed; d.bias = l.bias; d.res = l.res; d.y = l.y;
    d.x_ns = l.x_nstride; d.x2_ns = l.x2_nstride; d.res_ns = l.res_nstride; d.y_ns = l.y_nstride;
    d.c1 = l.x2 ? l.c1 : l.cin; d.cin = l.cin; d.act = l.act; d.nstage = cdiv(l.cin, 16);
    vec = vec && al8(l.y, l.y_nstride) && (!l.res || al8(l.res, l.res_nstride));
  }
  c.vec_ok = vec ? 1 : 0;
  const unsigned grid = (unsigned)(c.bpl * n_layers);
#if TG_LAB
  static const int noncoh = TG_LAB_ENV("TG_WINO_CHAIN_NONCOH", 0);   // timing only: results may be stale
  static const int pad_lds = TG_LAB_ENV("TG_WINO_CHAIN_LDS", 0);     // dynamic LDS to cap residency
  if (noncoh) {
    hipLaunchKernelGGL(conv3x3_wino_chain_kernel<false>, dim3(grid), dim3(256), pad_lds, (hipStream_t)stream, c);
    return check_launch("conv3x3_wino_chain");
  }
  hipLaunchKernelGGL(conv3x3_wino_chain_kernel<true>, dim3(grid), dim3(256), pad_lds, (hipStream_t)stream, c);
#else
  hipLaunchKernelGGL(conv3x3_wino_chain_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, c);
#endif
  return check_launch("conv3x3_wino_chain");
}
}  // namespace tg

extern "C" int tg_conv3x3_wino_chain(const tg_wino_layer* layers, int n_layers, int n, int cout, int h, int w,
                                     int32_t* flags, int epoch, tg_stream_t stream) {
  TG_REQUIRE(flags && n_layers >= 1 && n > 0 && h > 0 && w > 0, TG_E_ARG, "conv3x3_wino_chain: bad argument");
  const long long ntile = (long long)cdiv(w, 32) * cdiv(h, 2) * n;
  return conv3x3_wino_chain_launch(layers, n_layers, n, cout, h, w, flags, flags + (size_t)n_layers * ntile,
                                   (unsigned)epoch, TG_CHAIN_POLL_LIMIT_DEFAULT, stream);
}

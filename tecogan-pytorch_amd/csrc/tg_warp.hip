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
// Thread mapping: a block is one 256-pixel HR row segment with lane <-> HR x, so the
// bilinear gathers of a wave read ~contiguous addresses; the LR-flow terms of the segment
// are staged once in LDS, and the warped values are transposed through LDS so that each
// wave stores 64 consecutive LR columns of one (sy, sx, c) plane (256 contiguous bytes).
#include "tg_common.h"
#include <cstdlib>

namespace tg {

// row / column of the reflect-padded (bottom/right) flow -> source index.
// F.pad(..., 'reflect') at tecogan_nets.py:241: padded index fh+k mirrors fh-2-k.
// (min / max forms: integer v_min / v_max instead of compare + conditional move)
__device__ __forceinline__ int reflect_src(int i, int f) { const int m = 2 * f - 2 - i; return i < m ? i : m; }   // i < f <=> i <= m
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return max(lo, min(v, hi)); }

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
  // per-axis constants of the sampling grid, computed once on the host with the same IEEE
  // fp32 operations the reference performs per call: step = 2/(N-1), half = (N-1)/2 and
  // rhalf = RN(1/half) for the division below.
  float step_x, step_y, half_x, half_y, rhalf_x, rhalf_y;
  int in_aligned, out_aligned;   // 16-byte alignment of hr_prev / out (base and clip stride)
};

// x / d for a loop-invariant d with r = RN(1/d): the refinement tail of the IEEE division
// expansion (two residual corrections), correctly rounded for the normal-range operands
// of this path -- 5 FMA-class instructions instead of the ~12 of a generic fp32 divide.
// tools/div_lab.hip checks it bit-for-bit against `x / d` on the device.
__device__ __forceinline__ float div_const(float x, float d, float r) {
  float q = x * r;
  float e = __builtin_fmaf(-d, q, x);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-d, q, x);
  return __builtin_fmaf(e, r, q);
}

template <int AUX = 0>
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void bstore(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voff, (int)soff, AUX);
}

// warp_coord (tg_common.h) with the loop-invariant pieces hoisted
__device__ __forceinline__ float warp_coord_c(int i, int n, float flow, float step, float half,
                                              float rhalf) {
  float g = linspace_m1p1(i, n, step) + div_const(flow, half, rhalf);
  float p = (g + 1.0f) * half;
  p = __builtin_fmaxf(p, 0.f);
  p = __builtin_fminf(p, (float)(n - 1));
  return p;
}

// The kernel is bound by the texture-address unit, which spends ~16 cycles on every wave
// memory instruction whatever its width (rocprofv3: TA busy 75-95 % of the kernel with
// dword gathers and dword stores), so the design minimises the NUMBER of wave memory
// instructions per pixel:
//
// Block = R consecutive HR rows of one LR row (R divides S) x a 256-pixel HR column
// segment; each thread owns 2 horizontally adjacent pixels x R/2 rows.
//   1. the LR-flow patch of the block (4 source rows x NV columns x 2 channels, reflect pad
//      + replicate clamp folded into the index) is loaded ONCE into LDS; the vertical 4-tap
//      sums per (HR row, LR column) are formed from LDS (bilinear: the two source rows of
//      every HR row are loaded directly); S threads stage the horizontal bicubic weights,
//   2. every thread finishes its flows with the horizontal taps from LDS, evaluates the
//      sampling positions exactly as the reference does, and gathers, per channel and
//      source row, ONE 16-byte lane starting at pixel a's x0: it holds a's (x0, x0+1) taps
//      and -- when b samples the same source row 0..2 elements further right, which a
//      smooth flow makes the usual case -- b's as well.  Lanes where it does not load b's
//      taps separately (8-byte pairs).  The TA cost of a gather is per active lane,
//      independent of the lane width (measured), so sharing halves it.  A tap at N
//      (x0+1 == W or y0+1 == H) has weight exactly 0 because positions are clamped to
//      N-1: the lane may read the neighbouring element (or the buffer's out-of-range 0)
//      and the row offset is simply dropped -- no predication,
//   3. the R x C x 256 results are transposed through LDS and leave as dwordx4 stores:
//      16 lanes x 16 B = 256 contiguous bytes of one (sy, sx, c) plane.
// Measured alternatives (tools/warp_lab.py, 8 clips per launch): dword gathers +20 %;
// staging a flow-shifted window of the previous frame in LDS and sampling from LDS +-0 at
// zero flow and slower as soon as the flow varies inside a block; XCD-banded block order
// +4 %; nontemporal stores +-0; nontemporal gathers +30 %.
// All offsets are 32-bit against block-uniform buffer resources.
#ifndef TG_WARP_ABL
#define TG_WARP_ABL 0   // lab only (tools/warp_lab.py): 1 no stores, 2 one tap row instead of two, 4 no flow loads
#endif
#if !TG_LAB && TG_WARP_ABL
#error "tg_warp.hip: TG_WARP_ABL needs -DTG_LAB=1 (lab builds only; the ablated kernels compute wrong results)"
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef TG_WARP_PF
#define TG_WARP_PF 0    // 1: speculative touch of the previous frame under the flow loads for one-frame launches (see PF below);
                        // measured +-0 on MI355X (6.3 - 7.8 us per launch with and without, round 5): off
#endif
template <int S, int C, int R, int RPT, int PF = 0>
__global__ __launch_bounds__(128 * (R / RPT)) void flowup_warp_s2d_kernel(FusedArgs a) {
  constexpr int SEG = 256;
  constexpr int NT = 128 * (R / RPT);      // threads: 128 pixel pairs x R/RPT row groups
  constexpr int NV = SEG / S + 3;          // LR columns touched by the segment
  constexpr int OXB = SEG / S;             // LR columns produced by the segment
  constexpr int OUT_ITEMS = R * S * C * (OXB / 4);
  constexpr int PS = OXB + 16;             // staged plane stride: the S sub-pixel phases land in disjoint banks
  static_assert(S % R == 0 && R % RPT == 0, "rows of a block share one LR row; two row groups");
  __shared__ float s_raw[2][4][NV + 1];    // [flow channel][source row][LR column]
  __shared__ float2 s_fl[R][2][NV + 1];    // [row][slot][LR column] -> (flow x, flow y)
  __shared__ __attribute__((aligned(16))) float s_kx[S][4];
  __shared__ __attribute__((aligned(16))) float s_out[R * S * C * PS];

  const int t = threadIdx.x;
  const int HH = S * a.h, WW = S * a.w;
  const int nseg = (WW + SEG - 1) / SEG, nrb = HH / R;
  int tile = blockIdx.x;
  if (tile >= nseg * nrb * a.n) return;
  // integer division runs on the VALU; readfirstlane tells the compiler the results are
  // wave-uniform (otherwise every buffer access is wrapped in a waterfall loop)
  const int seg = __builtin_amdgcn_readfirstlane(tile % nseg);
  tile /= nseg;
  const int rb = __builtin_amdgcn_readfirstlane(tile % nrb);
  const int n = __builtin_amdgcn_readfirstlane(tile / nrb);
  const int x0 = seg * SEG;
  const int hy0 = rb * R;
  const int oy = hy0 / S, sy0 = hy0 - oy * S;
  const unsigned fhw = (unsigned)(a.fh * a.fw);
  const unsigned hrhw = (unsigned)(HH * WW);
  const unsigned lrhw = (unsigned)(a.h * a.w);
  const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.lr_flow + (size_t)n * 2 * fhw), 0, 2 * fhw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.hr_prev + (size_t)n * C * hrhw), 0, C * hrhw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
      a.out + (size_t)n * a.out_ns, 0, S * S * C * lrhw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
      a.hr_flow_out ? a.hr_flow_out + (size_t)n * 2 * hrhw : a.out, 0, 2 * hrhw * 4, 0x00020000);
  const int jbase = x0 / S - 1;
  const bool bicubic = a.up_mode == TG_UP_BICUBIC;
  const bool vec_out = (a.w & 3) == 0 && a.out_aligned;   // dwordx4 stores need 16-byte rows

  // PF (round 5, one-frame launches: every block is resident at once and the kernel's duration is ONE block's chain
  // of dependent round trips -- flow patch, gathers, stores).  The gathers cannot start before the flow is known, but
  // the flow only moves a sample by a few pixels: the 16 bytes at the thread's OWN position are requested at once, so
  // the previous frame's lines travel towards this CU's L2 / L1 while the flow patch is in flight and the real gathers
  // find them there.  The values are never used (the asm at the end only keeps the loads alive).
  // (requested BEHIND the flow loads: vmcnt retires in order, so the flow patch must not queue behind them)
  f32x4 pf[C];
  auto issue_pf = [&]() {
    if constexpr (PF != 0) {
      const int ptp = t & 127, pr = (t >> 7) * RPT;
      const unsigned po = ((unsigned)(hy0 + pr) * (unsigned)WW + (unsigned)(x0 + 2 * ptp)) * 4u;
#pragma unroll
      for (int ch = 0; ch < C; ++ch)
        pf[ch] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (int)po, (int)(ch * hrhw * 4u), 0));
    }
  };

  // ---- 1. LR-flow terms
  if (bicubic) {
    {   // every load of the patch is issued before the first LDS store: a loop of load -> store
        // iterations would pay the L2 latency once per iteration on the block's critical path
      constexpr int FPT = (2 * 4 * NV + NT - 1) / NT;
      float fv[FPT];
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int it = t + i * NT;
        const int k = it % NV, p = (it / NV) & 3, ch = it / (4 * NV);
        const unsigned cc = (unsigned)reflect_src(clampi(jbase + k, 0, a.w - 1), a.fw);
        const unsigned o = ((unsigned)reflect_src(clampi(oy - 1 + p, 0, a.h - 1), a.fh) * a.fw + cc) * 4u;
        fv[i] = (it < 2 * 4 * NV && !(TG_WARP_ABL & 4)) ? bload(rf, o, (ch & 1) * fhw * 4u) : 0.01f;
      }
      issue_pf();
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int it = t + i * NT;
        const int k = it % NV, p = (it / NV) & 3, ch = it / (4 * NV);
        if (it < 2 * 4 * NV) s_raw[ch][p][k] = fv[i];
      }
    }
    if (t >= NT - S) {
      float k[4];
      bicubic_w(t - (NT - S), S, k);
#pragma unroll
      for (int q = 0; q < 4; ++q) s_kx[t - (NT - S)][q] = k[q];
    }
    __syncthreads();
    for (int it = t; it < R * NV; it += NT) {
      const int r = it / NV, k = it - r * NV;
      float ky[4];
      bicubic_w(sy0 + r, S, ky);
      float sx_ = 0.f, sy_ = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) { sx_ += ky[p] * s_raw[0][p][k]; sy_ += ky[p] * s_raw[1][p][k]; }
      s_fl[r][0][k] = make_float2(sx_, sy_);
    }
  } else {
    constexpr int BPT = (R * NV + NT - 1) / NT;
    float2 b0[BPT], b1[BPT];
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int it = t + i * NT;
      const int r = it / NV, k = it - r * NV;
      const unsigned cc = (unsigned)reflect_src(clampi(jbase + k, 0, a.w - 1), a.fw);
      int y0, y1; float ly0, ly1;
      bilinear_src(hy0 + (r < R ? r : 0), S, a.h, y0, y1, ly0, ly1);
      unsigned o0 = ((unsigned)reflect_src(y0, a.fh) * a.fw + cc) * 4u;
      unsigned o1 = ((unsigned)reflect_src(y1, a.fh) * a.fw + cc) * 4u;
      if (it < R * NV) {
        b0[i] = make_float2(bload(rf, o0, 0), bload(rf, o0, fhw * 4u));
        b1[i] = make_float2(bload(rf, o1, 0), bload(rf, o1, fhw * 4u));
      }
    }
    issue_pf();
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int it = t + i * NT;
      const int r = it / NV, k = it - r * NV;
      if (it < R * NV) { s_fl[r][0][k] = b0[i]; s_fl[r][1][k] = b1[i]; }
    }
  }
  __syncthreads();

  // ---- 2. flows, sampling positions, gathers.  Thread -> 2 horizontally adjacent pixels
  //         (a, b) x RPT rows; the tile is 256 columns x R rows.
  const int tp = t & 127, r0 = (t >> 7) * RPT;
  const int hxa = x0 + 2 * tp;
  if (hxa < WW) {
    const bool live_b = hxa + 1 < WW;
    float fx[2][RPT], fy[2][RPT];
    if (bicubic) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k0 = (2 * tp + e) / S, dx = (2 * tp + e) - k0 * S;   // x0 is a multiple of S
        const f32x4 kx = *reinterpret_cast<const f32x4*>(s_kx[dx]);
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
          float ax = 0.f, ay = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float2 v = s_fl[r0 + j][0][k0 + q];
            ax += kx[q] * v.x; ay += kx[q] * v.y;
          }
          fx[e][j] = (float)S * ax; fy[e][j] = (float)S * ay;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int c0, c1; float lx0, lx1;
        bilinear_src(hxa + e, S, a.w, c0, c1, lx0, lx1);     // indices into the padded LR row
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
          int y0, y1; float ly0, ly1;
          bilinear_src(hy0 + r0 + j, S, a.h, y0, y1, ly0, ly1);
          const float2 t0 = s_fl[r0 + j][0][c0 - jbase], t1 = s_fl[r0 + j][0][c1 - jbase];
          const float2 b0 = s_fl[r0 + j][1][c0 - jbase], b1 = s_fl[r0 + j][1][c1 - jbase];
          float tx_ = lx0 * t0.x + lx1 * t1.x, bx_ = lx0 * b0.x + lx1 * b1.x;
          float ty_ = lx0 * t0.y + lx1 * t1.y, by_ = lx0 * b0.y + lx1 * b1.y;
          fx[e][j] = (float)S * (ly0 * tx_ + ly1 * bx_);
          fy[e][j] = (float)S * (ly0 * ty_ + ly1 * by_);
        }
      }
    }
    if (a.hr_flow_out) {
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const unsigned o = ((unsigned)(hy0 + r0 + j) * WW + hxa) * 4u;
        bstore(fx[0][j], ro, o, 0); bstore(fy[0][j], ro, o, hrhw * 4u);
        if (live_b) { bstore(fx[1][j], ro, o + 4u, 0); bstore(fy[1][j], ro, o + 4u, hrhw * 4u); }
      }
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int r = r0 + j;
      // positions, tap offsets and weights of both pixels
      unsigned o0[2], o1[2]; int ixy[2][2];
      float w00[2], w01[2], w10[2], w11[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float px = warp_coord_c(hxa + e, WW, fx[e][j], a.step_x, a.half_x, a.rhalf_x);
        const float py = warp_coord_c(hy0 + r, HH, fy[e][j], a.step_y, a.half_y, a.rhalf_y);
        const float fx0 = floorf(px), fy0 = floorf(py);
        const float wx1 = px - fx0, wx0 = 1.0f - wx1;
        const float wy1 = py - fy0, wy0 = 1.0f - wy1;
        const int ix = (int)fx0, iy = (int)fy0;
        ixy[e][0] = ix; ixy[e][1] = iy;
        o0[e] = ((unsigned)iy * WW + ix) * 4u;
        o1[e] = o0[e] + (unsigned)(min(iy + 1, HH - 1) - iy) * ((unsigned)WW * 4u);   // next row, or the same one at the border (weight 0)
        w00[e] = wy0 * wx0; w01[e] = wy0 * wx1; w10[e] = wy1 * wx0; w11[e] = wy1 * wx1;
      }
      // pixel b's taps lie inside pixel a's 16-byte lanes when both sample the same source
      // row and b starts 0..2 elements to the right (the usual case for a smooth flow)
      const int d = ixy[1][0] - ixy[0][0];
      const bool shared = ixy[1][1] == ixy[0][1] && (unsigned)d <= 2u;
      // Element d, d + 1 of the shared 16-byte lane by BIT masks, not by conditional moves: v_cndmask_b32
      // costs ~23 cycles per wave instruction on gfx950 against ~5 for v_bfi_b32 (tools/valu_lab.hip), and
      // the 12 three-way selects per channel pair were half of this kernel's VALU time (round 4).
      // m1 / m2 = all ones when d == 1 / d == 2 (d outside 0..2: the lane is not shared, vb is redone below)
      unsigned m1 = 0u - ((unsigned)d & 1u), m2 = 0u - (((unsigned)d >> 1) & 1u);
      asm volatile("" : "+v"(m1), "+v"(m2));          // opaque: the compiler turns known 0 / ~0 masks back into v_cndmask
      auto pick = [&](float e0, float e1, float e2) {     // v_bfi_b32 d, m, a, b = (m & a) | (~m & b): one instruction per step
        float r;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m1), "v"(e1), "v"(e0));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m2), "v"(e2), "v"(r));
        return r;
      };
      float va[C], vb[C];
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const unsigned pl = ch * hrhw * 4u;
        const f32x4 t4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (int)o0[0], (int)pl, 0));
        const f32x4 b4 = (TG_WARP_ABL & 2) ? t4 : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (int)o1[0], (int)pl, 0));
        va[ch] = ((t4[0] * w00[0] + t4[1] * w01[0]) + b4[0] * w10[0]) + b4[1] * w11[0];
        const float t0 = pick(t4[0], t4[1], t4[2]);
        const float t1 = pick(t4[1], t4[2], t4[3]);
        const float b0 = pick(b4[0], b4[1], b4[2]);
        const float b1 = pick(b4[1], b4[2], b4[3]);
        vb[ch] = ((t0 * w00[1] + t1 * w01[1]) + b0 * w10[1]) + b1 * w11[1];
      }
      if (!shared) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
          const unsigned pl = ch * hrhw * 4u;
          const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ri, (int)o0[1], (int)pl, 0));
          const f32x2 b2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ri, (int)o1[1], (int)pl, 0));
          vb[ch] = ((t2[0] * w00[1] + t2[1] * w01[1]) + b2[0] * w10[1]) + b2[1] * w11[1];
        }
      }
      const int oxa = (2 * tp) / S, sxa = 2 * tp - oxa * S;
      const int oxb = (2 * tp + 1) / S, sxb = 2 * tp + 1 - oxb * S;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        s_out[((r * S + sxa) * C + ch) * PS + oxa] = va[ch];
        s_out[((r * S + sxb) * C + ch) * PS + oxb] = vb[ch];
      }
    }
  }
  __syncthreads();
  if constexpr (PF != 0) {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) asm volatile("" ::"v"(pf[ch]));
  }

  // ---- 3. space_to_depth: staged as s_out[(r, sx, ch)][ox]
  if (vec_out) {
    // item -> (plane = (r, sx, ch), quad of 4 LR columns); 16 lanes cover one plane row
#pragma unroll
    for (int i = 0; i < (OUT_ITEMS + NT - 1) / NT; ++i) {
      const int it = t + i * NT;
      const int q = it % (OXB / 4), pl = it / (OXB / 4);
      const int ox = x0 / S + 4 * q;
      if (it < OUT_ITEMS && ox < a.w) {
        const int r = pl / (S * C), rem = pl - r * (S * C);   // rem = sx * C + ch
        const f32x4 v = *reinterpret_cast<const f32x4*>(&s_out[pl * PS + 4 * q]);
        const unsigned o = ((unsigned)((sy0 + r) * S * C + rem) * lrhw + (unsigned)oy * a.w + ox) * 4u;
        if ((TG_WARP_ABL & 1) && v[0] != 12345.678f) continue;
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(int)))) int, v), rout, (int)o, 0, 0);
      }
    }
  } else {
    for (int it = t; it < S * OXB; it += NT) {
      const int sx = it / OXB, oxl = it - sx * OXB;
      const int ox = x0 / S + oxl;
      if (ox >= a.w) continue;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const unsigned o = ((unsigned)(((sy0 + r) * S + sx) * C) * lrhw + (unsigned)oy * a.w + ox) * 4u;
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
          bstore(s_out[((r * S + sx) * C + ch) * PS + oxl], rout, o, ch * lrhw * 4u);
      }
    }
  }
}

__global__ __launch_bounds__(256) void backward_warp_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ flow,
                                                            float* __restrict__ y, int n, int c,
                                                            int h, int w, int s2d) {
  const int px_ = blockIdx.x * 64 + (threadIdx.x & 63);
  const int py_ = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (px_ >= w || py_ >= h) return;
  const long long hw = (long long)h * w;
  const float* fl = flow + (long long)b * 2 * hw + (long long)py_ * w + px_;
  float sx = warp_coord(px_, w, fl[0]);
  float sy = warp_coord(py_, h, fl[hw]);
  // s2d > 1: the result leaves in space_to_depth(., s2d) layout (net_utils.py:36-47: plane (sy s + sx) c + ch of
  // the h / s x w / s image) -- the training unroll's warp -> space_to_depth pair as one launch
  const int oy = py_ / s2d, ox = px_ / s2d, ph = (py_ - oy * s2d) * s2d + (px_ - ox * s2d);
  const long long ohw = hw / (s2d * s2d);
  float* yo = y + ((long long)b * c * s2d * s2d + (long long)ph * c) * ohw + (long long)oy * (w / s2d) + ox;
  for (int ch = 0; ch < c; ++ch) {
    const float* img = x + ((long long)b * c + ch) * hw;
    yo[ch * ohw] = warp_sample(img, h, w, sx, sy);
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

// Vector form for S in {2, 4}, w % (4 S) == 0 and 16-byte aligned rows: a thread reads 4 S
// consecutive pixels of one input row (S 16-byte loads) and writes one 16-byte vector into each
// of the S sub-pixel planes of that row -- every byte moves once in full sectors (the scalar
// form above reads with stride S: 67 MB of an HR training frame took 215 us, this takes ~15).
template <int S>
__global__ __launch_bounds__(256) void space_to_depth_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                long long y_ns, int n, int c, int h, int w) {
  const int oh = h / S, ow = w / S, groups = w / (4 * S);
  const long long total = (long long)n * c * h * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int gq = (int)(i % groups); long long t = i / groups;
    const int iy = (int)(t % h); t /= h;
    const int ch = (int)(t % c); const int b = (int)(t / c);
    const int oy = iy / S, sy = iy - oy * S;
    const f32x4* src = reinterpret_cast<const f32x4*>(x + (((long long)b * c + ch) * h + iy) * w + (long long)gq * 4 * S);
    float v[4 * S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const f32x4 q = src[k];
      v[4 * k] = q[0]; v[4 * k + 1] = q[1]; v[4 * k + 2] = q[2]; v[4 * k + 3] = q[3];
    }
#pragma unroll
    for (int sx = 0; sx < S; ++sx) {
      const f32x4 o = {v[sx], v[S + sx], v[2 * S + sx], v[3 * S + sx]};
      float* dst = y + (long long)b * y_ns + ((long long)((sy * S + sx) * c + ch) * oh + oy) * ow + gq * 4;
      *reinterpret_cast<f32x4*>(dst) = o;
    }
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

// Bilinear x2 (the three decoder stages of FNet, tecogan_nets.py:49-61; F.interpolate(scale_factor=2, mode='bilinear',
// align_corners=False)): one thread = 2 input columns x 1 input row -> a 2 x 4 output patch from 12 loads, two
// 16-byte stores.  Same weights (bilinear_src) and the same expression as upsample_kernel: bit-identical, without
// its per-element 64-bit index divisions and 4-byte stores (57 -> ~25 us on the 8 x 64 x 68 x 160 stage).
__global__ __launch_bounds__(256) void upsample_bilinear2x_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int nc, int h, int w, float mul) {
  const int hw2 = w >> 1, ow = 2 * w;
  const int total = nc * h * hw2;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int m = idx % hw2; const int t = idx / hw2;
    const int i = t % h, p = t / h;
    const float* src = x + (size_t)p * h * w;
    const int r[3] = {max(i - 1, 0), i, min(i + 1, h - 1)};
    const int c[4] = {max(2 * m - 1, 0), 2 * m, 2 * m + 1, min(2 * m + 2, w - 1)};
    float v[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) v[a][b] = src[r[a] * w + c[b]];
    float lx0[4], lx1[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) { int d0, d1; bilinear_src(4 * m + o, 2, w, d0, d1, lx0[o], lx1[o]); }
    float* dst = y + ((size_t)p * 2 * h + 2 * i) * ow + 4 * m;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int d0, d1; float ly0, ly1;
      bilinear_src(2 * i + e, 2, h, d0, d1, ly0, ly1);
      float tv[4], bv[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        tv[b] = e == 0 ? v[0][b] : v[1][b];                        // row y0
        bv[b] = e == 0 ? (i ? v[1][b] : v[2][b]) : v[2][b];        // row y1 (row 0: y0 = 0, y1 = min(1, h - 1))
      }
      float o4[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        // columns (x0, x1) of output column 4 m + o inside c[]: (0, 1 | 2 at m = 0), (1, 2), (1, 2), (2, 3)
        const int ka = o == 0 ? 0 : (o == 3 ? 2 : 1);
        const float ta = tv[ka], ba = bv[ka];
        const float tb = o == 0 ? (m ? tv[1] : tv[2]) : (o == 3 ? tv[3] : tv[2]);
        const float bb = o == 0 ? (m ? bv[1] : bv[2]) : (o == 3 ? bv[3] : bv[2]);
        float top = lx0[o] * ta + lx1[o] * tb;
        float bot = lx0[o] * ba + lx1[o] * bb;
        float vv = ly0 * top + ly1 * bot;
        o4[o] = mul * vv;
      }
      *reinterpret_cast<float4*>(dst + (size_t)e * ow) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
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

// uint8 HWC frames -> fp32 CHW in [0,1]: `gt.permute(0,3,1,2).float() / 255.0`
// (base_model.py:112).  x (n,h,w,c) uint8 -> y (n,c,h,w), true division.
__global__ void dequantize_u8_hwc_kernel(const uint8_t* __restrict__ x, float* __restrict__ y,
                                         int c, int h, int w, long long total) {
  const long long hw = (long long)h * w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long p = i % hw; long long t = i / hw;
    int ch = (int)(t % c); long long f = t / c;
    y[i] = (float)x[(f * hw + p) * c + ch] / 255.0f;
  }
}

// Y of rgb_to_ycbcr (data_utils.py:56-77): uint8(round(clip(r*T00 + g*T10 + b*T20 + 16)))
// in float64, products and sums rounded separately, left to right (numpy's matmul loop);
// rint = round-half-even like np.round.
__device__ __forceinline__ int luma_u8(int r, int g, int b) {
  double v = __dadd_rn(__dadd_rn(__dmul_rn((double)r, 0.256788235294118),
                                 __dmul_rn((double)g, 0.504129411764706)),
                       __dmul_rn((double)b, 0.097905882352941));
  v = __dadd_rn(v, 16.0);
  v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
  return (int)rint(v);
}

// compute_PSNR's sum of squared differences (metric_calculator.py:228-240) per frame, exact
// (integers): y_only = 1 -> on the Y channel, 0 -> over the three RGB channels.
__global__ __launch_bounds__(256) void psnr_sse_u8_kernel(const uint8_t* __restrict__ a,
                                                          const uint8_t* __restrict__ b,
                                                          unsigned long long* __restrict__ sse,
                                                          long long hw, int y_only) {
  __shared__ unsigned long long sm[4];
  const int f = blockIdx.y;
  const uint8_t* pa = a + (long long)f * hw * 3;
  const uint8_t* pb = b + (long long)f * hw * 3;
  unsigned long long s = 0;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw;
       p += (long long)gridDim.x * blockDim.x) {
    int ar = pa[3 * p], ag = pa[3 * p + 1], ab = pa[3 * p + 2];
    int br = pb[3 * p], bg = pb[3 * p + 1], bb = pb[3 * p + 2];
    if (y_only) {
      int d = luma_u8(ar, ag, ab) - luma_u8(br, bg, bb);
      s += (unsigned long long)(d * d);
    } else {
      int d0 = ar - br, d1 = ag - bg, d2 = ab - bb;
      s += (unsigned long long)(d0 * d0 + d1 * d1 + d2 * d2);
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sse + f, sm[0] + sm[1] + sm[2] + sm[3]);
}

// test hook for luma_u8: y[i] = Y(rgb[i])
__global__ void luma_u8_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = (uint8_t)luma_u8(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
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
  TG_REQUIRE((long long)scale * scale * c * h * w < (1ll << 31), TG_E_SHAPE,
             "flowup_warp_s2d: frame too large for 32-bit offsets");
  const float nx = (float)(scale * w - 1), ny = (float)(scale * h - 1);
  FusedArgs a{lr_flow, hr_prev, out, hr_flow_out, out_nstride, n, c, h, w, fh, fw, up_mode,
              2.0f / nx, 2.0f / ny, nx / 2.0f, ny / 2.0f, 1.0f / (nx / 2.0f), 1.0f / (ny / 2.0f),
              ((uintptr_t)hr_prev & 15) == 0 && (n == 1 || ((long long)c * scale * scale * h * w) % 4 == 0),
              ((uintptr_t)out & 15) == 0 && (n == 1 || out_nstride % 4 == 0)};
  const int rows = scale;
  const int tiles = cdiv(scale * w, 256) * (scale * h / rows) * n;
  hipStream_t s = (hipStream_t)stream;
  // rows per thread.  One frame (670 tiles at 134x320 LR) cannot fill the chip, and the kernel
  // is then bound by the per-wave dependency chain: 1 row per thread = twice the waves, half
  // the chain (7.1 vs 8.0 us).  Many clips per launch are throughput-bound and prefer fewer,
  // longer waves (46 vs 49 us at 8 clips).  TG_WARP_RPT overrides (lab).
  static const int rpt_env = TG_LAB_ENV("TG_WARP_RPT", 0);
  const int rpt = rpt_env ? rpt_env : (tiles <= 2048 ? 1 : 2);
  if (scale == 4) {
    if (rpt == 2) hipLaunchKernelGGL((flowup_warp_s2d_kernel<4, 3, 4, 2>), dim3(tiles), dim3(256), 0, s, a);
    else if (TG_WARP_PF) hipLaunchKernelGGL((flowup_warp_s2d_kernel<4, 3, 4, 1, 1>), dim3(tiles), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((flowup_warp_s2d_kernel<4, 3, 4, 1>), dim3(tiles), dim3(512), 0, s, a);
  } else {
    hipLaunchKernelGGL((flowup_warp_s2d_kernel<2, 3, 2, 1>), dim3(tiles), dim3(256), 0, s, a);
  }
  return check_launch("flowup_warp_s2d");
}

// Same-bytes copy ceiling of the fused warp kernel (bench.py: roofline_warp*.copy_ceiling): a float4 grid-stride copy
// with the warp launch's own grid, so that `frac` can also be read against what THIS part delivers for a launch of
// that size (MI355X_MICROARCH.md: 6.29 TB/s of the 8 TB/s peak only for large copies).
__global__ __launch_bounds__(512) void copy_f32x4_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

extern "C" int tg_copy_ceiling(const void* src, void* dst, int64_t bytes, int blocks, int threads, tg_stream_t stream) {
  TG_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0, TG_E_ARG, "copy_ceiling: bytes=%lld (a positive multiple of 16)", (long long)bytes);
  TG_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, TG_E_ARG, "copy_ceiling: 16-byte aligned buffers");
  TG_REQUIRE(blocks > 0 && threads > 0 && threads <= 512 && threads % 64 == 0, TG_E_ARG, "copy_ceiling: grid %d x %d", blocks, threads);
  hipLaunchKernelGGL(copy_f32x4_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)stream,
                     static_cast<const f32x4*>(src), static_cast<f32x4*>(dst), (long long)(bytes / 16));
  return check_launch("copy_ceiling");
}

extern "C" int tg_backward_warp_fwd(const float* x, const float* flow, float* y, int n, int c,
                                    int h, int w, tg_stream_t stream) {
  TG_REQUIRE(x && flow && y, TG_E_ARG, "backward_warp: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && h >= 2 && w >= 2, TG_E_SHAPE, "backward_warp: n=%d c=%d h=%d w=%d",
             n, c, h, w);
  dim3 g(cdiv(w, 64), cdiv(h, 4), n), t(256);
  hipLaunchKernelGGL(backward_warp_kernel, g, t, 0, (hipStream_t)stream, x, flow, y, n, c, h, w, 1);
  return check_launch("backward_warp");
}

extern "C" int tg_backward_warp_s2d_fwd(const float* x, const float* flow, float* y, int n, int c, int h, int w,
                                        int scale, tg_stream_t stream) {
  TG_REQUIRE(x && flow && y, TG_E_ARG, "backward_warp_s2d: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && h >= 2 && w >= 2 && scale >= 1 && h % scale == 0 && w % scale == 0, TG_E_SHAPE,
             "backward_warp_s2d: n=%d c=%d h=%d w=%d scale=%d", n, c, h, w, scale);
  dim3 g(cdiv(w, 64), cdiv(h, 4), n), t(256);
  hipLaunchKernelGGL(backward_warp_kernel, g, t, 0, (hipStream_t)stream, x, flow, y, n, c, h, w, scale);
  return check_launch("backward_warp_s2d");
}

extern "C" int tg_space_to_depth(const float* x, float* y, int64_t y_nstride, int n, int c, int h,
                                 int w, int scale, tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "space_to_depth: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && scale >= 1 && h >= scale && w >= scale, TG_E_SHAPE,
             "space_to_depth: n=%d c=%d h=%d w=%d s=%d", n, c, h, w, scale);
  long long total = (long long)n * scale * scale * c * (h / scale) * (w / scale);
  const bool vec = (scale == 2 || scale == 4) && h % scale == 0 && w % (4 * scale) == 0 &&
                   (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && y_nstride % 4 == 0;
  if (vec) {
    const long long items = (long long)n * c * h * (w / (4 * scale));
    if (scale == 2)
      hipLaunchKernelGGL(space_to_depth_vec_kernel<2>, dim3(grid1d(items)), dim3(256), 0, (hipStream_t)stream, x, y,
                         (long long)y_nstride, n, c, h, w);
    else
      hipLaunchKernelGGL(space_to_depth_vec_kernel<4>, dim3(grid1d(items)), dim3(256), 0, (hipStream_t)stream, x, y,
                         (long long)y_nstride, n, c, h, w);
    return check_launch("space_to_depth");
  }
  hipLaunchKernelGGL(space_to_depth_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                     x, y, (long long)y_nstride, n, c, h, w, scale);
  return check_launch("space_to_depth");
}

// 16-byte form of maxpool2_kernel (w % 4 == 0, 16-byte aligned planes): two outputs per thread from two float4 loads
__global__ __launch_bounds__(256) void maxpool2_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int nc,
                                                           int h, int w) {
  const int oh = h / 2, ow2 = w / 4;
  const int total = nc * oh * ow2;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int m = idx % ow2; const int t = idx / ow2;
    const int oy = t % oh, p = t / oh;
    const float* s = x + ((size_t)p * h + 2 * oy) * w + 4 * m;
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + w);
    *reinterpret_cast<float2*>(y + ((size_t)p * oh + oy) * (w / 2) + 2 * m) =
        make_float2(fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y)), fmaxf(fmaxf(a.z, a.w), fmaxf(b.z, b.w)));
  }
}

extern "C" int tg_upsample_fwd(const float* x, float* y, int nc, int h, int w, int scale,
                               int up_mode, float mul, tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "upsample: null pointer");
  TG_REQUIRE(nc > 0 && h > 0 && w > 0 && scale >= 1, TG_E_SHAPE, "upsample: nc=%d h=%d w=%d s=%d",
             nc, h, w, scale);
  TG_REQUIRE(up_mode == TG_UP_BICUBIC || up_mode == TG_UP_BILINEAR, TG_E_ARG, "upsample: mode=%d",
             up_mode);
  long long total = (long long)nc * h * scale * w * scale;
  if (up_mode == TG_UP_BILINEAR && scale == 2 && (w & 1) == 0 && total < (1ll << 31) &&
      (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    const long long th = (long long)nc * h * (w / 2);
    hipLaunchKernelGGL(upsample_bilinear2x_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, nc, h, w, mul);
    return check_launch("upsample");
  }
  hipLaunchKernelGGL(upsample_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, x, y,
                     nc, h, w, scale, up_mode, mul);
  return check_launch("upsample");
}

extern "C" int tg_maxpool2_fwd(const float* x, float* y, int nc, int h, int w,
                               tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "maxpool2: null pointer");
  TG_REQUIRE(nc > 0 && h >= 2 && w >= 2, TG_E_SHAPE, "maxpool2: nc=%d h=%d w=%d", nc, h, w);
  long long total = (long long)nc * (h / 2) * (w / 2);
  if ((w & 3) == 0 && total < (1ll << 31) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    hipLaunchKernelGGL(maxpool2_vec_kernel, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, nc, h, w);
    return check_launch("maxpool2");
  }
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

extern "C" int tg_dequantize_u8_hwc(const uint8_t* x, float* y, int n, int c, int h, int w,
                                    tg_stream_t stream) {
  TG_REQUIRE(x && y, TG_E_ARG, "dequantize_u8: null pointer");
  TG_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, TG_E_SHAPE, "dequantize_u8: n=%d c=%d h=%d w=%d", n, c, h, w);
  long long total = (long long)n * c * h * w;
  hipLaunchKernelGGL(dequantize_u8_hwc_kernel, dim3(grid1d(total)), dim3(256), 0,
                     (hipStream_t)stream, x, y, c, h, w, total);
  return check_launch("dequantize_u8");
}

extern "C" int tg_psnr_sse_u8(const uint8_t* true_hwc, const uint8_t* pred_hwc, uint64_t* sse,
                              int frames, int h, int w, int y_only, tg_stream_t stream) {
  TG_REQUIRE(true_hwc && pred_hwc && sse, TG_E_ARG, "psnr_sse_u8: null pointer");
  TG_REQUIRE(frames > 0 && frames < 65536 && h > 0 && w > 0, TG_E_SHAPE,
             "psnr_sse_u8: frames=%d h=%d w=%d", frames, h, w);
  const long long hw = (long long)h * w;
  if (hipMemsetAsync(sse, 0, sizeof(uint64_t) * frames, (hipStream_t)stream) != hipSuccess)
    return check_launch("psnr_sse_u8 memset");
  long long bx = (hw + 1023) / 1024;
  hipLaunchKernelGGL(psnr_sse_u8_kernel, dim3((unsigned)(bx > 512 ? 512 : bx), frames), dim3(256), 0,
                     (hipStream_t)stream, true_hwc, pred_hwc, (unsigned long long*)sse, hw, y_only);
  return check_launch("psnr_sse_u8");
}

extern "C" int tg_luma_u8(const uint8_t* rgb, uint8_t* y, int64_t n, tg_stream_t stream) {
  TG_REQUIRE(rgb && y && n > 0, TG_E_ARG, "luma_u8: bad argument");
  hipLaunchKernelGGL(luma_u8_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, rgb, y,
                     (long long)n);
  return check_launch("luma_u8");
}

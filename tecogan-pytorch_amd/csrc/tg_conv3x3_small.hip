// 3x3 conv with <= 4 output channels: direct fp32 VALU kernel for gfx950.
//
// An MFMA tile has at least 16 (32) output rows; with cout = 2 or 3 more than
// 80-90 % of the matrix core would multiply padding, so this layer type runs
// on the vector ALU instead: each lane owns 4 consecutive pixels x COUT
// channels (<= 12 accumulators); per (cin, ky) it reads 6 input floats from
// the LDS halo patch (one b128 + one b64) and issues 3*4*COUT FMAs against
// wave-uniform weights (scalar registers / LDS broadcast).
//
// Fused epilogue: bias, activation (tanh*24 for the flow head), and the
// `out += upsample_func(lr_curr)` residual of SRNet.forward
// (codes/models/networks/tecogan_nets.py:145) evaluated per output pixel from
// the low-resolution source (bicubic: net_utils.py:133-156, bilinear: :86-89).
//
// Replaces FNet.flow[2] (tecogan_nets.py:65,80) and SRNet.conv_out (:131,145).
#include "tg_common.h"

namespace tg {

constexpr int S_TH = 16;          // tile rows
constexpr int S_TWT = 16;         // thread columns
constexpr int S_PXT = 4;          // pixels per thread along x
constexpr int S_TW = S_TWT * S_PXT;   // 64
constexpr int S_PH = S_TH + 2;
constexpr int S_PW = S_TW + 2;    // 66
constexpr int S_RS = 68;          // LDS row stride (16-byte aligned rows)
constexpr int S_CK = 4;           // cin chunk

struct SmallArgs {
  const float* x;
  const float* wt;     // OIHW
  const float* bias;
  const float* up;     // (n, cout, h/us, w/us) or null
  float* y;
  long long x_ns, y_ns;
  int cin, cout, h, w, act, up_mode, up_scale;
  int tiles_x, tiles_y;
  uint8_t* u8;         // optional (n == 1): the same result as HWC uint8, float32_to_uint8
                       // (codes/utils/data_utils.py:80-87) fused into the epilogue
  const float* res;    // optional (n, cout, h, w) residual added after the activation (training: the
  long long res_ns;    // bicubic frame computed once per step); conv3x3_small_ks_kernel only
};

template <int COUT>
__device__ __forceinline__ void add_upsampled(const SmallArgs& a, int n, int py, int px,
                                              float v[COUT]) {
  const int us = a.up_scale;
  const int lh = a.h / us, lw = a.w / us;
  const float* src = a.up + (long long)n * COUT * lh * lw;
  if (a.up_mode == TG_UP_BICUBIC) {
    const int i = py / us, dy = py - i * us, j = px / us, dx = px - j * us;
    float ky[4], kx[4];
    bicubic_w(dy, us, ky);
    bicubic_w(dx, us, kx);
    int ri[4], ci[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = i - 1 + p; ri[p] = r < 0 ? 0 : (r > lh - 1 ? lh - 1 : r);
      int c = j - 1 + p; ci[p] = c < 0 ? 0 : (c > lw - 1 ? lw - 1 : c);
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      const float* s = src + (long long)o * lh * lw;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float vq = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) vq += ky[p] * s[ri[p] * lw + ci[q]];
        acc += kx[q] * vq;
      }
      v[o] += acc;
    }
  } else {  // bilinear, align_corners=False
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilinear_src(py, us, lh, y0, y1, ly0, ly1);
    bilinear_src(px, us, lw, x0, x1, lx0, lx1);
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      const float* s = src + (long long)o * lh * lw;
      float top = lx0 * s[y0 * lw + x0] + lx1 * s[y0 * lw + x1];
      float bot = lx0 * s[y1 * lw + x0] + lx1 * s[y1 * lw + x1];
      v[o] += ly0 * top + ly1 * bot;
    }
  }
}

template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(SmallArgs a) {
  __shared__ __attribute__((aligned(16))) float s_in[2][S_CK][S_PH][S_RS];

  const int tid = threadIdx.x;
  const int tcx = tid % S_TWT, tcy = tid / S_TWT;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * S_TW, y0 = ty * S_TH;
  const long long hw = (long long)a.h * a.w;
  const float* xb = a.x + (long long)n * a.x_ns;

  // wave-uniform weights through the constant address space = scalar loads (see the v2 kernel)
  const __attribute__((address_space(4))) float* wk =
      (const __attribute__((address_space(4))) float*)a.wt;           // OIHW

  constexpr int PATCH = S_PH * S_PW;                       // 1188 per channel
  constexpr int PER_T = (S_CK * PATCH + 255) / 256;        // 19
  float rin[PER_T];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / PATCH, rem = idx - c * PATCH;
      int r = rem / S_PW, col = rem - r * S_PW;
      int gy = y0 - 1 + r, gx = x0 - 1 + col;
      float v = 0.f;
      if (c < S_CK && c0 + c < a.cin && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w)
        v = xb[(long long)(c0 + c) * hw + (long long)gy * a.w + gx];
      rin[i] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      int idx = tid + i * 256;
      int c = idx / PATCH, rem = idx - c * PATCH;
      int r = rem / S_PW, col = rem - r * S_PW;
      if (c < S_CK) s_in[buf][c][r][col] = rin[i];
    }
  };

  float acc[COUT][S_PXT];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int p = 0; p < S_PXT; ++p) acc[o][p] = 0.f;

  const int nchunk = cdiv(a.cin, S_CK);
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    const bool more = ch + 1 < nchunk;
    if (more) load_chunk((ch + 1) * S_CK);
#pragma unroll
    for (int c = 0; c < S_CK; ++c) {
      const int cg = ch * S_CK + c;
      if (cg < a.cin) {
        float wreg[COUT][9];
#pragma unroll
        for (int o = 0; o < COUT; ++o)
#pragma unroll
          for (int t = 0; t < 9; ++t) wreg[o][t] = wk[(o * a.cin + cg) * 9 + t];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* row = &s_in[buf][c][tcy + ky][tcx * S_PXT];
          float4 v4 = *reinterpret_cast<const float4*>(row);
          float2 v2 = *reinterpret_cast<const float2*>(row + 4);
          float in6[6] = {v4.x, v4.y, v4.z, v4.w, v2.x, v2.y};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
              float wv = wreg[o][ky * 3 + kx];
#pragma unroll
              for (int p = 0; p < S_PXT; ++p) acc[o][p] += wv * in6[p + kx];
            }
          }
        }
      }
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

  const int py = y0 + tcy;
  if (py < a.h) {
#pragma unroll
    for (int p = 0; p < S_PXT; ++p) {
      const int px = x0 + tcx * S_PXT + p;
      if (px < a.w) {
        float v[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o)
          v[o] = apply_act(acc[o][p] + (a.bias ? a.bias[o] : 0.f), a.act);
        if (a.up) add_upsampled<COUT>(a, n, py, px, v);
#pragma unroll
        for (int o = 0; o < COUT; ++o)
          a.y[(long long)n * a.y_ns + (long long)o * hw + (long long)py * a.w + px] = v[o];
      }
    }
  }
}


// ---- v2: 16-byte staging (needs w % 4 == 0 and 16-byte aligned planes) -----------------
// The LDS patch of a channel covers image columns [x0-4, x0+68): whole float4s, so every
// global access is one bounds-checked buffer_load_dwordx4 (out-of-image quads read as 0)
// written back with one ds_write_b128; a thread's six taps of a row are two b128 + one b32.
constexpr int V_Q = 18;                 // float4 per patch row (72 columns)
constexpr int V_RS = 76;                // LDS row stride in floats (19 slots: odd)
constexpr int V_CK = 4;
constexpr int V_ITEMS = V_CK * S_PH * V_Q;          // 1296 float4 per chunk
constexpr int V_PER_T = (V_ITEMS + 255) / 256;      // 6
constexpr unsigned V_OOB = 0x80000000u;

template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_v2_kernel(SmallArgs a) {
  // ONE LDS buffer (21.9 KB => 7 workgroups per CU instead of 3): the two chunks in flight live
  // in registers, and with 28 waves per CU the second barrier per chunk costs less than the
  // occupancy the double buffer took.
  __shared__ __attribute__((aligned(16))) float s_in[1][V_CK][S_PH][V_RS];
  // The weights are wave-uniform: read through the constant address space they become scalar
  // loads (s_load_dwordx*) and feed the FMAs as SGPR operands.  Staging them in LDS cost 27
  // broadcast ds_reads per input channel and made the kernel LDS-issue-bound (PMC:
  // SQ_WAIT_INST_LDS 37 % of the wave cycles, 32 % of the LDS cycles bank conflicts).
  const __attribute__((address_space(4))) float* wk =
      (const __attribute__((address_space(4))) float*)a.wt;           // OIHW

  const int tid = threadIdx.x;
  const int tcx = tid % S_TWT, tcy = tid / S_TWT;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * S_TW, y0 = ty * S_TH;
  const int hw = a.h * a.w;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.cin * hw * 4, 0x00020000);

  // per-thread staging slots (fixed for the whole kernel)
  unsigned voff[V_PER_T];
  int lds_off[V_PER_T];
#pragma unroll
  for (int i = 0; i < V_PER_T; ++i) {
    int idx = tid + i * 256;
    int c = idx / (S_PH * V_Q), rem = idx - c * (S_PH * V_Q);
    int r = rem / V_Q, q = rem - r * V_Q;
    int gy = y0 - 1 + r, gx = x0 - 4 + 4 * q;
    bool ok = idx < V_ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    voff[i] = ok ? (unsigned)((c * hw + gy * a.w + gx) * 4) : V_OOB;
    lds_off[i] = idx < V_ITEMS ? (c * S_PH + r) * V_RS + 4 * q : -1;
  }
  const unsigned plane = (unsigned)hw * 4u;
  // Two chunks are in flight: the loop is bound by the latency of the global loads (4
  // channels = 0.4 us of FMAs per chunk against ~2 us to fetch the next one), so chunk
  // ch + 2 is requested before chunk ch is consumed; rinA / rinB alternate (loop unrolled x2).
  f32x4 rinA[V_PER_T], rinB[V_PER_T];
  auto load_chunk = [&](f32x4 (&rin)[V_PER_T], int c0) {
#pragma unroll
    for (int i = 0; i < V_PER_T; ++i) {
      // channels past cin fall beyond num_records and read as 0
      rin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                             rsrc, (int)(voff[i] + (unsigned)c0 * plane), 0, 0));
    }
  };
  auto store_chunk = [&](const f32x4 (&rin)[V_PER_T], int buf) {
    float* base = &s_in[buf][0][0][0];
#pragma unroll
    for (int i = 0; i < V_PER_T; ++i)
      if (lds_off[i] >= 0) *reinterpret_cast<f32x4*>(base + lds_off[i]) = rin[i];
  };

  float acc[COUT][S_PXT];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int p = 0; p < S_PXT; ++p) acc[o][p] = 0.f;

  auto compute = [&](int ch, int buf) {
#pragma unroll
    for (int c = 0; c < V_CK; ++c) {
      const int cg = ch * V_CK + c;
      if (cg < a.cin) {
        float wreg[COUT][9];
#pragma unroll
        for (int o = 0; o < COUT; ++o)
#pragma unroll
          for (int t = 0; t < 9; ++t) wreg[o][t] = wk[(o * a.cin + cg) * 9 + t];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* row = &s_in[buf][c][tcy + ky][tcx * S_PXT];
          f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
          f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 4);
          float v2 = row[8];
          float in6[6] = {v0[3], v1[0], v1[1], v1[2], v1[3], v2};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
              float wv = wreg[o][ky * 3 + kx];
#pragma unroll
              for (int p = 0; p < S_PXT; ++p) acc[o][p] += wv * in6[p + kx];
            }
          }
        }
      }
    }
  };

  const int nchunk = cdiv(a.cin, V_CK);
  load_chunk(rinA, 0);
  store_chunk(rinA, 0);
  __syncthreads();
  if (nchunk > 1) load_chunk(rinB, V_CK);            // chunk 1 -> B
  // invariant at the top of an even iteration ch: LDS = chunk ch, B = chunk ch + 1 (in flight)
  for (int ch = 0; ch < nchunk; ch += 2) {
    if (ch + 2 < nchunk) load_chunk(rinA, (ch + 2) * V_CK);
    compute(ch, 0);
    __syncthreads();                                 // everyone has consumed chunk ch
    if (ch + 1 < nchunk) {
      store_chunk(rinB, 0);
      __syncthreads();
      if (ch + 3 < nchunk) load_chunk(rinB, (ch + 3) * V_CK);
      compute(ch + 1, 0);
      __syncthreads();
      if (ch + 2 < nchunk) {
        store_chunk(rinA, 0);
        __syncthreads();
      }
    }
  }

  const int py = y0 + tcy;
  const int px0 = x0 + tcx * S_PXT;
  if (py < a.h && px0 < a.w) {          // w % 4 == 0: the four pixels are all inside
    float v[S_PXT][COUT];
#pragma unroll
    for (int p = 0; p < S_PXT; ++p) {
#pragma unroll
      for (int o = 0; o < COUT; ++o)
        v[p][o] = apply_act(acc[o][p] + (a.bias ? a.bias[o] : 0.f), a.act);
      if (a.up) add_upsampled<COUT>(a, n, py, px0 + p, v[p]);
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      f32x4 ov = {v[0][o], v[1][o], v[2][o], v[3][o]};
      *reinterpret_cast<f32x4*>(a.y + (long long)n * a.y_ns + (long long)o * hw +
                                (long long)py * a.w + px0) = ov;
    }
    if (a.u8) {
      // uint8(clip(round_half_even(x * 255), 0, 255)), HWC: the thread's 4 pixels x COUT channels
      // are 4 * COUT consecutive bytes (px0 % 4 == 0 => 4-byte aligned)
      uint8_t q[S_PXT * COUT];
#pragma unroll
      for (int p = 0; p < S_PXT; ++p)
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          float r = rintf(v[p][o] * 255.0f);
          r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
          q[p * COUT + o] = (uint8_t)r;
        }
      uint32_t* dst = reinterpret_cast<uint32_t*>(a.u8 + ((long long)py * a.w + px0) * COUT);
#pragma unroll
      for (int k = 0; k < COUT; ++k)
        dst[k] = (uint32_t)q[4 * k] | ((uint32_t)q[4 * k + 1] << 8) | ((uint32_t)q[4 * k + 2] << 16) |
                 ((uint32_t)q[4 * k + 3] << 24);
    }
  }
}

// ---- small frames (the training unroll: 2 x 128 x 128 / 2 x 256 x 256 HR pixels per launch) -------
// The 16 x 64 tiles above give 32 / 128 workgroups there, each walking 16 channel chunks with a
// global-load round trip per chunk: 39 us for 0.1 GFLOP (the step used the MFMA kernel instead, 3 of 32
// output columns alive: 16 / 43 us).  Here a tile is 4 rows x 64 columns and the FOUR WAVES of a
// workgroup split the input channels (wave k: chunks k, k + 4, ...), each staging its chunks into its
// own slice of LDS -- no workgroup barrier inside the loop -- and the partial sums meet in LDS once at
// the end (fixed order: deterministic).  4x the workgroups, 4x fewer dependent round trips each.
constexpr int K_TH = 4, K_PH = K_TH + 2;
constexpr int K_ITEMS = V_CK * K_PH * V_Q;          // 432 float4 per chunk
constexpr int K_PER_T = (K_ITEMS + 63) / 64;        // 7 per lane
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_ks_kernel(SmallArgs a) {
  __shared__ __attribute__((aligned(16))) float s_in[4][V_CK][K_PH][V_RS];      // 29.2 KB; later the partial sums
  const __attribute__((address_space(4))) float* wk =
      (const __attribute__((address_space(4))) float*)a.wt;           // OIHW, wave-uniform: scalar loads
  const int tid = threadIdx.x, lane = tid & 63, kg = tid >> 6;
  const int tcx = lane & 15, tcy = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * S_TW, y0 = ty * K_TH;
  const int hw = a.h * a.w;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.cin * hw * 4, 0x00020000);
  unsigned voff[K_PER_T];
  int lds_off[K_PER_T];
#pragma unroll
  for (int i = 0; i < K_PER_T; ++i) {
    const int idx = lane + i * 64;
    const int c = idx / (K_PH * V_Q), rem = idx - c * (K_PH * V_Q);
    const int r = rem / V_Q, q = rem - r * V_Q;
    const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * q;
    const bool ok = idx < K_ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    voff[i] = ok ? (unsigned)((c * hw + gy * a.w + gx) * 4) : V_OOB;
    lds_off[i] = idx < K_ITEMS ? (c * K_PH + r) * V_RS + 4 * q : -1;
  }
  const unsigned plane = (unsigned)hw * 4u;
  float* mine = &s_in[kg][0][0][0];
  f32x4 rinA[K_PER_T], rinB[K_PER_T];
  auto load_chunk = [&](f32x4 (&rin)[K_PER_T], int ch) {
#pragma unroll
    for (int i = 0; i < K_PER_T; ++i)       // channels past cin fall beyond num_records and read as 0
      rin[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                             rsrc, (int)(voff[i] + (unsigned)(ch * V_CK) * plane), 0, 0));
  };
  auto store_chunk = [&](const f32x4 (&rin)[K_PER_T]) {
#pragma unroll
    for (int i = 0; i < K_PER_T; ++i)
      if (lds_off[i] >= 0) *reinterpret_cast<f32x4*>(mine + lds_off[i]) = rin[i];
  };
  float acc[COUT][S_PXT];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int p = 0; p < S_PXT; ++p) acc[o][p] = 0.f;
  auto compute = [&](int ch) {
#pragma unroll
    for (int c = 0; c < V_CK; ++c) {
      const int cg = ch * V_CK + c;
      if (cg < a.cin) {
        float wreg[COUT][9];
#pragma unroll
        for (int o = 0; o < COUT; ++o)
#pragma unroll
          for (int t = 0; t < 9; ++t) wreg[o][t] = wk[(o * a.cin + cg) * 9 + t];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* row = mine + (c * K_PH + tcy + ky) * V_RS + tcx * S_PXT;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 4);
          const float v2 = row[8];
          const float in6[6] = {v0[3], v1[0], v1[1], v1[2], v1[3], v2};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
              const float wv = wreg[o][ky * 3 + kx];
#pragma unroll
              for (int p = 0; p < S_PXT; ++p) acc[o][p] += wv * in6[p + kx];
            }
        }
      }
    }
  };
  // this wave's chunks kg, kg + 4, ...: one chunk in LDS, the next two in registers (A / B alternate).
  // A wave's LDS operations execute in program order, so its own writes / reads need no barrier.
  const int nchunk = cdiv(a.cin, V_CK);
  int ch = kg;
  if (ch < nchunk) {
    load_chunk(rinA, ch);
    if (ch + 4 < nchunk) load_chunk(rinB, ch + 4);
    store_chunk(rinA);
    for (;;) {
      if (ch + 8 < nchunk) load_chunk(rinA, ch + 8);
      compute(ch);
      ch += 4;
      if (ch >= nchunk) break;
      store_chunk(rinB);
      if (ch + 8 < nchunk) load_chunk(rinB, ch + 8);
      compute(ch);
      ch += 4;
      if (ch >= nchunk) break;
      store_chunk(rinA);
    }
  }
  // ---- the four waves' sums meet in LDS: red[kg][o * 4 + p][lane]
  __syncthreads();
  float* red = &s_in[0][0][0][0];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int p = 0; p < S_PXT; ++p) red[(kg * COUT * S_PXT + o * S_PXT + p) * 64 + lane] = acc[o][p];
  __syncthreads();
  if (tid < 64 * COUT) {
    const int o = tid >> 6;
    const int py = y0 + tcy, px0 = x0 + tcx * S_PXT;      // (lane = tid & 63: the same pixel quad)
    if (py < a.h && px0 < a.w) {
      const float bb = a.bias ? a.bias[o] : 0.f;
      f32x4 ov;
#pragma unroll
      for (int p = 0; p < S_PXT; ++p) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) sum += red[(g * COUT * S_PXT + o * S_PXT + p) * 64 + lane];
        ov[p] = apply_act(sum + bb, a.act);
      }
      const long long off = (long long)o * hw + (long long)py * a.w + px0;
      if (a.res) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + (long long)n * a.res_ns + off);
#pragma unroll
        for (int p = 0; p < S_PXT; ++p) ov[p] += rv[p];
      }
      *reinterpret_cast<f32x4*>(a.y + (long long)n * a.y_ns + off) = ov;
    }
  }
}

// ---- the mirror image: cin <= 4, many output channels -- the data gradient of a small-cout head
// (conv_out 64 -> 3: dX = conv(dZ (3 ch), rot180 weights) to 64 channels, masked by the ReLU of the
// layer below).  On the MFMA kernel K = 27 is padded to a 72-deep chunk and the launch is bound by its
// epilogue: 31.7 us for 2 x 256 x 256 (33.5 MB written, 33.5 MB of mask read: ~13 us of traffic).
// Here: tile 4 rows x 64 columns, the 3 x 6 x 72 patch in LDS once, a thread keeps its 3 x 3 x 6 input
// window in registers and walks its wave's output channels (wave k: channels 16 k .. 16 k + 15 of every
// 64) with the 27 weights of a channel as scalar operands: 108 FMAs, one mask load, one 16-byte store.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_fewin_kernel(SmallArgs a) {
  __shared__ __attribute__((aligned(16))) float s_in[CIN][K_PH][V_RS];
  const __attribute__((address_space(4))) float* wk =
      (const __attribute__((address_space(4))) float*)a.wt;           // (cout, CIN, 3, 3)
  const int tid = threadIdx.x, lane = tid & 63, kg = tid >> 6;
  const int tcx = lane & 15, tcy = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * S_TW, y0 = ty * K_TH;
  const int hw = a.h * a.w;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, CIN * hw * 4, 0x00020000);
  constexpr int ITEMS = CIN * K_PH * V_Q;                  // <= 432 float4
#pragma unroll
  for (int i = 0; i < (ITEMS + 255) / 256; ++i) {
    const int idx = tid + i * 256;
    const int c = idx / (K_PH * V_Q), rem = idx - c * (K_PH * V_Q);
    const int r = rem / V_Q, q = rem - r * V_Q;
    const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * q;
    const bool ok = idx < ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rsrc, ok ? (int)((c * hw + gy * a.w + gx) * 4) : (int)V_OOB, 0, 0));
    if (idx < ITEMS) *reinterpret_cast<f32x4*>(&s_in[c][r][4 * q]) = v;
  }
  __syncthreads();
  float in[CIN][3][6];
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* row = &s_in[c][tcy + ky][tcx * S_PXT];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 4);
      in[c][ky][0] = v0[3]; in[c][ky][1] = v1[0]; in[c][ky][2] = v1[1]; in[c][ky][3] = v1[2]; in[c][ky][4] = v1[3];
      in[c][ky][5] = row[8];
    }
  const int py = y0 + tcy, px0 = x0 + tcx * S_PXT;
  if (py >= a.h || px0 >= a.w) return;                    // (w % 4 == 0: the four pixels are all inside or outside)
  const float* mn = a.res ? a.res + (long long)n * a.res_ns : nullptr;      // (the ReLU mask travels in `res`)
  float* yn = a.y + (long long)n * a.y_ns;
  for (int og = kg * 16; og < a.cout; og += 64) {
#pragma unroll 4
    for (int oo = 0; oo < 16; ++oo) {
      const int o = og + oo;
      if (o >= a.cout) break;
      float acc[S_PXT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float wv = wk[((o * CIN + c) * 3 + ky) * 3 + kx];
#pragma unroll
            for (int p = 0; p < S_PXT; ++p) acc[p] += wv * in[c][ky][p + kx];
          }
      const long long off = (long long)o * hw + (long long)py * a.w + px0;
      f32x4 ov = {acc[0], acc[1], acc[2], acc[3]};
      if (mn) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mn + off);
#pragma unroll
        for (int p = 0; p < S_PXT; ++p) ov[p] = m[p] > 0.f ? ov[p] : 0.f;
      }
      *reinterpret_cast<f32x4*>(yn + off) = ov;
    }
  }
}

}  // namespace tg

using namespace tg;

// whether tg_conv3x3_small_fwd_u8 can emit the uint8 frame for this launch (else: tg_quantize_u8_hwc)
extern "C" int tg_conv3x3_small_can_fuse_u8(const float* x, int64_t x_nstride, const float* y,
                                            int64_t y_nstride, int n, int cin, int h, int w) {
  return n == 1 && (w % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
         (x_nstride % 4 == 0) && (y_nstride % 4 == 0) && ((long long)(cin + 4) * h * w * 4 < (1ll << 31));
}

extern "C" int tg_conv3x3_small_fwd(const float* x, int64_t x_nstride, const float* w_oihw,
                                    const float* bias, const float* up_src, int up_mode,
                                    int up_scale, float* y, int64_t y_nstride, int n, int cin,
                                    int cout, int h, int w, int act, tg_stream_t stream) {
  return tg_conv3x3_small_fwd_u8(x, x_nstride, w_oihw, bias, up_src, up_mode, up_scale, y, y_nstride,
                                 nullptr, n, cin, cout, h, w, act, stream);
}

static int small_launch(const float* x, int64_t x_nstride, const float* w_oihw,
                        const float* bias, const float* up_src, int up_mode,
                        int up_scale, float* y, int64_t y_nstride, uint8_t* u8_out,
                        int n, int cin, int cout, int h, int w, int act,
                        tg_stream_t stream, const float* res, int64_t res_nstride) {
  TG_REQUIRE(x && w_oihw && y, TG_E_ARG, "conv3x3_small_fwd: null pointer");
  TG_REQUIRE(n > 0 && cin > 0 && cin <= 64 && cout >= 1 && cout <= 4 && h > 0 && w > 0,
             TG_E_SHAPE, "conv3x3_small_fwd: n=%d cin=%d (<=64) cout=%d (<=4) h=%d w=%d", n, cin,
             cout, h, w);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_TANH24, TG_E_ARG, "conv3x3_small: act=%d", act);
  if (up_src) {
    TG_REQUIRE((up_mode == TG_UP_BICUBIC || up_mode == TG_UP_BILINEAR) && up_scale >= 1 &&
                   h % up_scale == 0 && w % up_scale == 0,
               TG_E_SHAPE, "conv3x3_small_fwd: up_mode=%d up_scale=%d", up_mode, up_scale);
  }
  SmallArgs a{};
  a.x = x; a.wt = w_oihw; a.bias = bias; a.up = up_src; a.y = y; a.x_ns = x_nstride;
  a.y_ns = y_nstride; a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = act;
  a.up_mode = up_mode; a.up_scale = up_scale;
  a.u8 = u8_out; a.res = res; a.res_ns = res_nstride;
  a.tiles_x = cdiv(w, S_TW); a.tiles_y = cdiv(h, S_TH);
  long long blocks = (long long)a.tiles_x * a.tiles_y * n;
  TG_REQUIRE(blocks > 0 && blocks < (1ll << 31), TG_E_SHAPE, "conv3x3_small: grid %lld", blocks);
  dim3 g((unsigned)blocks), t(256);
  hipStream_t s = (hipStream_t)stream;
  const bool vec_ok = (w % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
                      (x_nstride % 4 == 0) && (y_nstride % 4 == 0) &&
                      ((long long)(cin + 4) * h * w * 4 < (1ll << 31));
  TG_REQUIRE(!u8_out || (n == 1 && vec_ok && ((uintptr_t)u8_out % 4 == 0)), TG_E_ARG,
             "conv3x3_small_fwd_u8: the fused uint8 output needs n == 1, w %% 4 == 0 and aligned planes "
             "(tg_conv3x3_small_can_fuse_u8)");
  TG_REQUIRE(!res || (vec_ok && !up_src && !u8_out && ((uintptr_t)res % 16 == 0) && res_nstride % 4 == 0), TG_E_ARG,
             "conv3x3_small_fwd_res: the residual form needs w %% 4 == 0, aligned planes, no up_src / uint8 output");
  // few 16-row tiles (the training frames): 4-row tiles, the four waves of a workgroup split the channels
  if (vec_ok && !up_src && !u8_out && (res || blocks < 512)) {
    a.tiles_y = cdiv(h, K_TH);
    const long long kb = (long long)a.tiles_x * a.tiles_y * n;
    TG_REQUIRE(kb < (1ll << 31), TG_E_SHAPE, "conv3x3_small: grid %lld", kb);
    dim3 kgrid((unsigned)kb);
    switch (cout) {
      case 1: hipLaunchKernelGGL(conv3x3_small_ks_kernel<1>, kgrid, t, 0, s, a); break;
      case 2: hipLaunchKernelGGL(conv3x3_small_ks_kernel<2>, kgrid, t, 0, s, a); break;
      case 3: hipLaunchKernelGGL(conv3x3_small_ks_kernel<3>, kgrid, t, 0, s, a); break;
      default: hipLaunchKernelGGL(conv3x3_small_ks_kernel<4>, kgrid, t, 0, s, a); break;
    }
    return check_launch("conv3x3_small_ks");
  }
  if (vec_ok) {
    switch (cout) {
      case 1: hipLaunchKernelGGL(conv3x3_small_v2_kernel<1>, g, t, 0, s, a); break;
      case 2: hipLaunchKernelGGL(conv3x3_small_v2_kernel<2>, g, t, 0, s, a); break;
      case 3: hipLaunchKernelGGL(conv3x3_small_v2_kernel<3>, g, t, 0, s, a); break;
      default: hipLaunchKernelGGL(conv3x3_small_v2_kernel<4>, g, t, 0, s, a); break;
    }
    return check_launch("conv3x3_small_v2");
  }
  switch (cout) {
    case 1: hipLaunchKernelGGL(conv3x3_small_kernel<1>, g, t, 0, s, a); break;
    case 2: hipLaunchKernelGGL(conv3x3_small_kernel<2>, g, t, 0, s, a); break;
    case 3: hipLaunchKernelGGL(conv3x3_small_kernel<3>, g, t, 0, s, a); break;
    default: hipLaunchKernelGGL(conv3x3_small_kernel<4>, g, t, 0, s, a); break;
  }
  return check_launch("conv3x3_small");
}

extern "C" int tg_conv3x3_small_fwd_u8(const float* x, int64_t x_nstride, const float* w_oihw,
                                       const float* bias, const float* up_src, int up_mode,
                                       int up_scale, float* y, int64_t y_nstride, uint8_t* u8_out,
                                       int n, int cin, int cout, int h, int w, int act,
                                       tg_stream_t stream) {
  return small_launch(x, x_nstride, w_oihw, bias, up_src, up_mode, up_scale, y, y_nstride, u8_out, n, cin, cout, h, w,
                      act, stream, nullptr, 0);
}

// y = act(conv3x3(x) + bias) + res  (res: (n, cout, h, w), e.g. the bicubic frame of `out += upsample_func(lr)`,
// tecogan_nets.py:145, computed once per training step); w % 4 == 0, 16-byte aligned planes.
extern "C" int tg_conv3x3_small_fwd_res(const float* x, int64_t x_nstride, const float* w_oihw, const float* bias,
                                        const float* res, int64_t res_nstride, float* y, int64_t y_nstride, int n,
                                        int cin, int cout, int h, int w, int act, tg_stream_t stream) {
  TG_REQUIRE(res, TG_E_ARG, "conv3x3_small_fwd_res: null residual");
  return small_launch(x, x_nstride, w_oihw, bias, nullptr, 0, 1, y, y_nstride, nullptr, n, cin, cout, h, w, act, stream,
                      res, res_nstride);
}

// y = relu_mask > 0 ? conv3x3(x) : 0 for cin <= 4 and any number of output channels (no bias / activation):
// the data gradient of a cout <= 4 head, w_oihw = (cout, cin, 3, 3) of THIS op (for a gradient: the layer's
// weights with the channel roles swapped and the taps rotated).  w % 4 == 0, 16-byte aligned planes.
extern "C" int tg_conv3x3_fewin_fwd(const float* x, int64_t x_nstride, const float* w_oihw, const float* relu_mask,
                                    int64_t mask_nstride, float* y, int64_t y_nstride, int n, int cin, int cout,
                                    int h, int w, tg_stream_t stream) {
  TG_REQUIRE(x && w_oihw && y, TG_E_ARG, "conv3x3_fewin_fwd: null pointer");
  TG_REQUIRE(n > 0 && cin >= 1 && cin <= 4 && cout >= 1 && h > 0 && w > 0, TG_E_SHAPE,
             "conv3x3_fewin_fwd: n=%d cin=%d (<= 4) cout=%d h=%d w=%d", n, cin, cout, h, w);
  TG_REQUIRE((w % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && (x_nstride % 4 == 0) &&
                 (y_nstride % 4 == 0) && (!relu_mask || (((uintptr_t)relu_mask % 16 == 0) && mask_nstride % 4 == 0)) &&
                 ((long long)cout * h * w * 4 < (1ll << 31)),
             TG_E_ARG, "conv3x3_fewin_fwd: needs w %% 4 == 0 and 16-byte aligned planes");
  SmallArgs a{};
  a.x = x; a.wt = w_oihw; a.y = y; a.x_ns = x_nstride; a.y_ns = y_nstride; a.cin = cin; a.cout = cout; a.h = h; a.w = w;
  a.res = relu_mask; a.res_ns = mask_nstride;
  a.tiles_x = cdiv(w, S_TW); a.tiles_y = cdiv(h, K_TH);
  const long long blocks = (long long)a.tiles_x * a.tiles_y * n;
  TG_REQUIRE(blocks > 0 && blocks < (1ll << 31), TG_E_SHAPE, "conv3x3_fewin: grid %lld", blocks);
  dim3 g((unsigned)blocks), t(256);
  hipStream_t s = (hipStream_t)stream;
  switch (cin) {
    case 1: hipLaunchKernelGGL(conv3x3_fewin_kernel<1>, g, t, 0, s, a); break;
    case 2: hipLaunchKernelGGL(conv3x3_fewin_kernel<2>, g, t, 0, s, a); break;
    case 3: hipLaunchKernelGGL(conv3x3_fewin_kernel<3>, g, t, 0, s, a); break;
    default: hipLaunchKernelGGL(conv3x3_fewin_kernel<4>, g, t, 0, s, a); break;
  }
  return check_launch("conv3x3_fewin");
}

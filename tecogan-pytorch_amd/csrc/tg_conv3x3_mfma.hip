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
// output channels.  Input channels stream through LDS in chunks of 8 (double
// buffered, one barrier per chunk).  Inside a chunk the K order is permuted so
// that MFMA k-step kk of a tap consumes channel 4*(lane>>5) + kk: a lane's four
// k-steps then need four CONSECUTIVE channels, and both operands of a whole tap
// (4 MFMAs) come from one ds_read_b128 each:
//     weights  LDS [tap][half][oc][4]     (A: lane -> oc,  16 B = 4 channels)
//     input    LDS [row][half][col][4]    (B: lane -> col, 16 B = 4 channels)
// Consecutive lanes read consecutive 16-byte slots: conflict free.  Operands
// of tap t+1 are fetched into a second register set while tap t's MFMAs issue.
//
// Staging is branch free: bounds-checked raw buffer loads (out-of-image halo
// pixels and channel tails get an out-of-range offset and read as 0), four
// channel planes of one pixel per thread -> one ds_write_b128.
//
// Reference ops replaced: nn.Conv2d(.,.,3,1,1)+bias+act(+residual) at
// codes/models/networks/tecogan_nets.py:23-65, 92-98, 111-113, 367-369 and the
// torch.cat at :71 / :141 (two-source input).
#include "tg_common.h"
#include <cstdlib>

namespace tg {

constexpr int TW = 32;          // pixels per row segment (MFMA N)
constexpr int PW = TW + 2;      // patch width incl. halo
constexpr int RS = 34;          // LDS patch row stride (16-byte slots)

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
  // split-K: each workgroup reduces chunks [ks*nchunk/ksplit, (ks+1)*nchunk/ksplit) and
  // writes RAW partial sums to part + ks*part_ss (n, cout, h, w); splitk_finalize_kernel
  // adds them in a fixed order (deterministic) with bias / activation / pooling.
  int ksplit;
  float* part;
  long long part_ss;
  int vec_ok;   // w % 4 == 0 and 16-byte aligned y / res planes: float4 epilogue allowed
  long long* dbg;   // lab instrumentation (ABL & 16): 8 cycle stamps per workgroup
  int nblocks;      // > 0: XCD-banded block order over nblocks tiles (grid = 8 * ceil(nblocks / 8))
  // optional ReLU-backward mask applied last: y = mask > 0 ? y : 0  (same layout as y).  Lets the
  // data-gradient conv of a layer deliver dZ of the PREVIOUS layer directly (no act_bwd pass).
  const float* mask;
  long long mask_ns;
  // Phase-restricted taps (strided convs through their space-to-depth embedding, see
  // tap_rows()): tapsel 0 = all nine taps; 1 = the phase of an INPUT channel chunk selects the
  // taps (cphase channels per phase); 2 = the phase of the OUTPUT channel block does.
  // rowsets[py] / colsets[px] name the tap rows / columns in use (TAPS_* below).
  int tapsel, cphase;
  unsigned char rowsets[2];
};

// pack OIHW (or IOHW for transposed convs) -> [ocg][chunk][tap][half][ocb][4]
// (channel within chunk = 4*half + j), zero padded in cin and cout.
__global__ void pack3x3_kernel(const float* __restrict__ w, float* __restrict__ out,
                               int cin, int cout, int ocb, int nchunk, int nocg,
                               int transposed) {
  int total = nocg * nchunk * 9 * CK * ocb;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    int j = i & 3;
    int t = i >> 2;
    int o = t % ocb; t /= ocb;
    int half = t & 1; t >>= 1;
    int tap = t % 9; t /= 9;
    int ch = t % nchunk;
    int g = t / nchunk;
    int oc = g * ocb + o, ci = ch * CK + 4 * half + j;
    float v = 0.f;
    if (oc < cout && ci < cin) {
      if (transposed == 2) v = w[((size_t)ci * cout + oc) * 9 + (8 - tap)];   // dgrad: swap + rot180
      else if (transposed == 1) v = w[((size_t)ci * cout + oc) * 9 + tap];
      else v = w[((size_t)oc * cin + ci) * 9 + tap];
    }
    out[i] = v;
  }
}

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

constexpr unsigned OOB = 0x80000000u;   // >= any num_records we build (tensors < 2 GiB per item)

// Tap subsets of the embedded strided convolutions.  A Conv2d(k4, s2, p1) on x equals a 3x3
// conv on space_to_depth(x, 2) whose sub-pixel phase (py, px) only owns the kernel rows
// {ky : 2*oy - 1 + ky = 2*(oy + ty - 1) + py}: py = 1 -> ty in {0, 1}, py = 0 -> ty in {1, 2}
// (same along x); the transposed conv's data gradient on space_to_depth(dY, 2) owns
// py = 1 -> {0, 1}, py = 0 -> {1}.  The embedded 3x3 weights are zero elsewhere: 4/9 (resp.
// 9/36) of the MFMAs of the dense kernel are useful, the rest multiplied zeros.  A set is
// named by a code: 0 = {0,1,2}, 1 = {0,1}, 2 = {1,2}, 3 = {1}.
enum { TAPS_ALL = 0, TAPS_01 = 1, TAPS_12 = 2, TAPS_1 = 3 };
__host__ __device__ constexpr int taps_first(int code) { return (code == TAPS_12 || code == TAPS_1) ? 1 : 0; }
__host__ __device__ constexpr int taps_count(int code) { return code == TAPS_ALL ? 3 : (code == TAPS_1 ? 1 : 2); }

// The MFMAs of one channel chunk over the tap rows RY x tap columns RX (all static): operands
// of the next tap are fetched into the other register set while this tap's MFMAs issue.
template <int NT, int OCB, int RY, int RX>
__device__ __forceinline__ void chunk_taps(const float* si, const float* sw, f32x16 (&acc)[NT]) {
  constexpr int Y0 = taps_first(RY), NY = taps_count(RY), X0 = taps_first(RX), NX = taps_count(RX);
  constexpr int N = NY * NX;
  f32x4 bq[2], aq[2][NT];
  auto fetch = [&](int i, int slot) {
    const int ky = Y0 + i / NX, kx = X0 + i % NX;
    bq[slot] = *reinterpret_cast<const f32x4*>(si + (ky * 2 * 34 + kx) * 4);      // RS = 34
#pragma unroll
    for (int t = 0; t < NT; ++t)
      aq[slot][t] = *reinterpret_cast<const f32x4*>(sw + (ky * 3 + kx) * (2 * OCB * 4) + t * 128);
  };
  fetch(0, 0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int cur = i & 1;
    if (i + 1 < N) fetch(i + 1, cur ^ 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][t][kk], bq[cur][kk], acc[t], 0, 0, 0);
    }
  }
}

template <int NT, int OCB>
__device__ __forceinline__ void chunk_taps_sel(int ry, int rx, const float* si, const float* sw,
                                               f32x16 (&acc)[NT]) {
  // wave-uniform dispatch to the statically unrolled variant
  switch (ry * 4 + rx) {
    case TAPS_01 * 4 + TAPS_01: chunk_taps<NT, OCB, TAPS_01, TAPS_01>(si, sw, acc); break;
    case TAPS_01 * 4 + TAPS_12: chunk_taps<NT, OCB, TAPS_01, TAPS_12>(si, sw, acc); break;
    case TAPS_12 * 4 + TAPS_01: chunk_taps<NT, OCB, TAPS_12, TAPS_01>(si, sw, acc); break;
    case TAPS_12 * 4 + TAPS_12: chunk_taps<NT, OCB, TAPS_12, TAPS_12>(si, sw, acc); break;
    case TAPS_01 * 4 + TAPS_1: chunk_taps<NT, OCB, TAPS_01, TAPS_1>(si, sw, acc); break;
    case TAPS_1 * 4 + TAPS_01: chunk_taps<NT, OCB, TAPS_1, TAPS_01>(si, sw, acc); break;
    case TAPS_1 * 4 + TAPS_1: chunk_taps<NT, OCB, TAPS_1, TAPS_1>(si, sw, acc); break;
    case TAPS_12 * 4 + TAPS_1: chunk_taps<NT, OCB, TAPS_12, TAPS_1>(si, sw, acc); break;
    case TAPS_1 * 4 + TAPS_12: chunk_taps<NT, OCB, TAPS_1, TAPS_12>(si, sw, acc); break;
    default: chunk_taps<NT, OCB, TAPS_ALL, TAPS_ALL>(si, sw, acc); break;
  }
}

// ABL: ablation bits for tools/conv_lab.hip only (0 in the product):
//   1 = no re-staging inside the chunk loop, 2 = no barrier in the loop,
//   4 = no epilogue stores, 8 = no LDS operand reads.
// OPT bits (tuning switches, selected by the launcher): 1 = epilogue through LDS with
// 16-byte stores (needs a.vec_ok), 2 = weight chunks staged by LDS-DMA (global_load_lds).
// KS = 2: the workgroup's waves are also split over K -- wave group wk takes every KS-th
// channel chunk (both chunks of a pair are staged together), the partial accumulators are
// added through LDS in a fixed order.  For problems with fewer 32x32 tiles than SIMDs
// (training: 2 x 64 x 64 LR frames) this doubles the busy SIMDs without a second launch.
template <int WM, int WN, int NT, bool DUAL, int ABL = 0, int OPT = 0, int STAGGER = 12, int KS = 1>
__global__ __launch_bounds__(WM* WN* KS * 64) void conv3x3_mfma_kernel(Conv3x3Args a) {
  static_assert(KS == 1 || (KS == 2 && (OPT & 2)), "the K-split variant stages weights by LDS-DMA");
  constexpr int NTHREADS = WM * WN * KS * 64;
  constexpr int OCB = WN * NT * 32;
  constexpr int PH = WM + 2;
  constexpr int IN_ITEMS = PH * 2 * PW;            // 16-byte items (pixel x 4 channels) per chunk
  constexpr int IN_FLOATS = PH * 2 * RS * 4;
  constexpr int W_FLOATS = 9 * CK * OCB;
  constexpr int W_VEC4 = W_FLOATS / 4;
  constexpr int W_PER_T = (W_VEC4 + NTHREADS - 1) / NTHREADS;
  constexpr int I_PER_T = (IN_ITEMS + NTHREADS - 1) / NTHREADS;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                        // [2][KS][IN_FLOATS]
  float* s_w = smem + 2 * KS * IN_FLOATS;    // [2][KS][W_FLOATS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = (wave / WM) % WN;
  const int wk = wave / (WM * WN);           // K group (0 when KS == 1)
  // double-buffer slot of the chunk (pair) that starts at channel chunk `ch`
  auto bufof = [](int ch) { return KS == 1 ? (ch & 1) : ((ch >> 1) & 1); };

  int b = blockIdx.x;
  if (a.nblocks > 0) {
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Give XCD x the
    // contiguous band of tiles [x*per, (x+1)*per): vertically adjacent tiles then share their
    // halo rows through one L2 instead of each fetching them from HBM / Infinity Cache.
    const int per = (a.nblocks + 7) >> 3;
    b = (b & 7) * per + (b >> 3);
    if (b >= a.nblocks) return;
  }
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int ocg = b % a.nocg; b /= a.nocg;
  const int ks = b % a.ksplit;
  const int n = b / a.ksplit;
  const int x0 = tx * TW, y0 = ty * WM;
  const int hw = a.h * a.w;
  const int ch_begin = (int)((long long)ks * a.nchunk / a.ksplit);
  const int ch_end = (int)((long long)(ks + 1) * a.nchunk / a.ksplit);

  // ---- staging assignment: item q = (row r, half hf, col) -------------------
  unsigned voff[I_PER_T];     // byte offset of channel (4*hf) at this pixel, or OOB
  int lds_item[I_PER_T];      // float index of the 16-byte LDS slot, -1 if none
#pragma unroll
  for (int i = 0; i < I_PER_T; ++i) {
    int q = tid + i * NTHREADS;
    int r = q / (2 * PW), rem = q - r * (2 * PW);
    int hf = rem / PW, col = rem - hf * PW;
    int gy = y0 - 1 + r, gx = x0 - 1 + col;
    bool ok = q < IN_ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
    voff[i] = ok ? (unsigned)((4 * hf * hw + gy * a.w + gx) * 4) : OOB;
    lds_item[i] = q < IN_ITEMS ? ((r * 2 + hf) * RS + col) * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.c1 * hw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(DUAL ? a.x2 + (long long)n * a.x2_ns : a.x), 0,
      DUAL ? (a.cin - a.c1) * hw * 4 : 0, 0x00020000);
  const unsigned plane = (unsigned)hw * 4u;

  const f32x4* wsrc =
      reinterpret_cast<const f32x4*>(a.wpk + (size_t)ocg * a.nchunk * W_FLOATS);

  f32x4 rin[KS][I_PER_T];
  f32x4 rw[W_PER_T];

  auto load_chunk = [&](int ch) {
    if (!((ABL & 32) && ch != ch_begin))      // lab: skip input re-staging
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const unsigned cbase = (unsigned)((ch + k) * CK) * plane;
#pragma unroll
      for (int i = 0; i < I_PER_T; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // channel ch*8 + 4*hf + j.  Beyond c1 (or cin) the offset is past
          // num_records and the load returns 0; the second source then supplies it.
          unsigned o1 = voff[i] + cbase + (unsigned)j * plane;
          float v = buf_load(rs1, o1);
          if (DUAL) v += buf_load(rs2, o1 - (unsigned)a.c1 * plane);
          rin[k][i][j] = v;
        }
      }
    }
    const f32x4* ws = wsrc + (size_t)ch * W_VEC4;
    if ((ABL & 64) && ch != ch_begin) return;   // lab: skip weight re-staging
    if constexpr (OPT & 2) {
      // LDS-DMA: each wave instruction moves 1 KiB (64 lanes x 16 B) of the packed chunk
      // straight into the other weight buffer; no VGPR round trip, no ds_write.
      constexpr int PIECES = W_FLOATS * 4 / 1024;
      constexpr int NWAVES = NTHREADS / 64;
      // chunk ch + k lives in slot bufof(ch) * KS + k; the packed chunks of a pair are
      // contiguous in global memory and in LDS, so the pair is one run of KS * PIECES pieces
      const char* src = reinterpret_cast<const char*>(ws) + lane * 16;
      char* dst = reinterpret_cast<char*>(s_w + bufof(ch) * KS * W_FLOATS);
      const int valid = (ch_end - ch < KS ? ch_end - ch : KS) * PIECES;   // never read past the last chunk
#pragma unroll
      for (int i = 0; i < (KS * PIECES + NWAVES - 1) / NWAVES; ++i) {
        int piece = wave + i * NWAVES;
        if (piece < valid)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(src + piece * 1024),
              (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < W_PER_T; ++i) {
        int idx = tid + i * NTHREADS;
        rw[i] = ws[idx < W_VEC4 ? idx : W_VEC4 - 1];
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      float* si = s_in + (buf * KS + k) * IN_FLOATS;
#pragma unroll
      for (int i = 0; i < I_PER_T; ++i)
        if (lds_item[i] >= 0) *reinterpret_cast<f32x4*>(si + lds_item[i]) = rin[k][i];
    }
    if constexpr (!(OPT & 2)) {
      f32x4* sw = reinterpret_cast<f32x4*>(s_w + buf * W_FLOATS);
#pragma unroll
      for (int i = 0; i < W_PER_T; ++i) {
        int idx = tid + i * NTHREADS;
        if (idx < W_VEC4) sw[idx] = rw[i];
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int lh = lane >> 5, ll = lane & 31;
  // per-lane LDS offsets (floats) of the 16-byte operand slots
  const int b_off = ((wm * 2 + lh) * RS + ll) * 4;
  const int a_off = (lh * OCB + wn * (NT * 32) + ll) * 4;

  if constexpr (OPT & 4) {
    // De-synchronise the workgroups that share a CU: they run the same instruction stream
    // and would otherwise hit their staging / barrier points at the same moments, leaving the
    // MFMA pipe idle.  Offset each residency "layer" by a fraction of a chunk period.
    const int layer = (blockIdx.x / 256) % 3;
    for (int i = 0; i < layer; ++i) __builtin_amdgcn_s_sleep(STAGGER);
  }
  long long tk0 = 0, tk1 = 0, t_mfma = 0, t_sync = 0, t_issue = 0;
  if constexpr (ABL & 16) tk0 = clock64();
  load_chunk(ch_begin);
  store_chunk(bufof(ch_begin));
  __syncthreads();
  if constexpr (ABL & 16) tk1 = clock64();

  for (int ch = ch_begin; ch < ch_end; ch += KS) {
    long long ta = 0, tb = 0, tc = 0;
    if constexpr (ABL & 16) ta = clock64();
    const int buf = bufof(ch);
    const bool more = (ch + KS < ch_end) && !(ABL & 1);
    if (more) load_chunk(ch + KS);

    if constexpr (ABL & 16) tb = clock64();
    const float* si = s_in + (buf * KS + wk) * IN_FLOATS + b_off;
    const float* sw = s_w + (buf * KS + wk) * W_FLOATS + a_off;
    if (KS == 1 || ch + wk < ch_end) {       // odd chunk count: the last pair has one member
    if (a.tapsel) {
      // phase of this chunk's input channels (1) or of this block's output channels (2)
      const int ph = a.tapsel == 1 ? ((ch + wk) * CK) / a.cphase : (ocg * OCB) / a.cphase;
      chunk_taps_sel<NT, OCB>(a.rowsets[(ph >> 1) & 1], a.rowsets[ph & 1], si, sw, acc);
    } else {
    f32x4 bq[2], aq[2][NT];
    bq[0] = *reinterpret_cast<const f32x4*>(si);
#pragma unroll
    for (int t = 0; t < NT; ++t) aq[0][t] = *reinterpret_cast<const f32x4*>(sw + t * 128);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1, nxt = cur ^ 1;
      if (tap + 1 < 9 && !(ABL & 8)) {
        const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
        bq[nxt] = *reinterpret_cast<const f32x4*>(si + (ky * 2 * RS + kx) * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          aq[nxt][t] = *reinterpret_cast<const f32x4*>(sw + (tap + 1) * (2 * OCB * 4) + t * 128);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][t][kk], bq[cur][kk], acc[t], 0, 0, 0);
      }
    }
    }
    }
    if constexpr (ABL & 16) {
      // wait for the accumulator (i.e. for the last MFMA) before stamping
      asm volatile("s_nop 0" ::"v"(acc[0][0]));
      tc = clock64();
    }
    if (more) store_chunk(buf ^ 1);
    if (!(ABL & 2)) __syncthreads();
    if constexpr (ABL & 16) {
      long long td = clock64();
      t_issue += tb - ta; t_mfma += tc - tb; t_sync += td - tc;
    }
  }
  if constexpr (ABL & 16) {
    if (a.dbg && tid == 0) {
      long long* d = a.dbg + (long long)blockIdx.x * 8;
      d[0] = tk0; d[1] = tk1 - tk0; d[2] = t_issue; d[3] = t_mfma; d[4] = t_sync; d[5] = clock64();
      d[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID[3:0]
    }
  }

  if constexpr (KS == 2) {
    // all staging buffers are dead (the loop ends on a barrier): K group 1 hands its
    // partial sums to K group 0, which owns the epilogue.  acc = group0 + group1, fixed order.
    float* red = smem;
    if (wk == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          red[(((wn * WM + wm) * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[t][r] += red[(((wn * WM + wm) * NT + t) * 16 + r) * 64 + lane];
    }
    __syncthreads();          // the epilogue re-uses the same LDS
  }
  const bool do_ep = wk == 0;
  // ---- epilogue: bias, activation, residual, NCHW store --------------------
  // All loads are issued before any use (independent, one wait), and the
  // activation is a select on a wave-uniform slope: no branches per element.
  const int px = x0 + ll, py = y0 + wm;
  const bool inimg = px < a.w && py < a.h;
  const long long opix = (long long)py * a.w + px;
  const int ocb0 = ocg * OCB + wn * (NT * 32) + 4 * lh;
  if (a.ksplit > 1) {   // raw partial sums, wave-uniform branch
    if (inimg) {
      float* pb = a.part + (long long)ks * a.part_ss + (long long)n * a.cout * hw + opix;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int oc = ocb0 + t * 32 + (r & 3) + 8 * (r >> 2);
          if (oc < a.cout) pb[(long long)oc * hw] = acc[t][r];
        }
    }
    return;
  }
  const float slope = act_slope(a.act);
  if constexpr (OPT & 1) {
    if (a.vec_ok) {      // wave-uniform
      // accumulators -> LDS [oc][32 px] (stride 36) -> each lane re-reads 4 consecutive
      // pixels of one channel and issues 16-byte stores: 4x fewer store instructions.
      constexpr int ES = 36;
      float* ep = smem + (wave % (WM * WN)) * (NT * 32 * ES);
      if (do_ep) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ep[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ES + ll] = acc[t][r];
      }
      __syncthreads();
      const int ocw = ocg * OCB + wn * (NT * 32);
      if (do_ep && py < a.h) {
#pragma unroll
        for (int j = 0; j < NT * 4; ++j) {
          int idx4 = j * 64 + lane;
          int ol = idx4 >> 3, p4 = (idx4 & 7) * 4;
          int oc = ocw + ol;
          int gx = x0 + p4;
          if (oc < a.cout && gx < a.w) {
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + ol * ES + p4);
            float bb = a.bias ? a.bias[oc] : 0.f;
            long long off = (long long)oc * hw + (long long)py * a.w + gx;
            f32x4 rr = {0.f, 0.f, 0.f, 0.f};
            if (a.res) rr = *reinterpret_cast<const f32x4*>(a.res + (long long)n * a.res_ns + off);
            f32x4 mm = {1.f, 1.f, 1.f, 1.f};
            if (a.mask) mm = *reinterpret_cast<const f32x4*>(a.mask + (long long)n * a.mask_ns + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float q = v[e] + bb;
              q = (q >= 0.f ? q : q * slope + 0.f) + rr[e];
              v[e] = mm[e] > 0.f ? q : 0.f;
            }
            *reinterpret_cast<f32x4*>(a.y + (long long)n * a.y_ns + off) = v;
          }
        }
      }
      return;
    }
  }
  if (!do_ep) return;        // no barrier below
  float bv[NT][16], rv[NT][16];
  unsigned dead[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) dead[t] = 0u;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int oc = ocb0 + t * 32 + (r & 3) + 8 * (r >> 2);
      int occ = oc < a.cout ? oc : a.cout - 1;
      bv[t][r] = a.bias ? a.bias[occ] : 0.f;
      rv[t][r] = (a.res && inimg) ? a.res[(long long)n * a.res_ns + opix + (long long)occ * hw] : 0.f;
      if (a.mask && inimg && a.mask[(long long)n * a.mask_ns + opix + (long long)occ * hw] <= 0.f)
        dead[t] |= 1u << r;
    }
  if (inimg) {
    float* yb = a.y + (long long)n * a.y_ns + opix;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int oc = ocb0 + t * 32 + (r & 3) + 8 * (r >> 2);
        float v = acc[t][r] + bv[t][r];
        v = (v >= 0.f ? v : v * slope + 0.f) + rv[t][r];
        if (dead[t] & (1u << r)) v = 0.f;
        if (oc < a.cout && (!(ABL & 4) || v == 12345.678f)) yb[(long long)oc * hw] = v;
      }
  }
}

// ---------------------------------------------------------------------------------------
// One-shot variant for the training frames (2 x 64 x 64 LR: 256 one-row tiles for 256 CUs) and
// cin <= 64.  The K-split variant above still walks the channel chunks in four double-buffered
// iterations, each ending on a barrier; with ONE workgroup per CU nothing hides its prologue,
// its barriers or its reduce, and a launch takes 13.4 us for 4.4 us of MFMAs (800 such launches
// per training step).  Here the whole K range is in flight at once:
//   * 8 waves = 2 oc halves x 4 K groups; every wave requests ITS weights (<= 2 chunks x 9 taps,
//     one coalesced 16-byte load per tap: the packed layout is the A operand) straight into
//     registers and its share of the input patch (all chunks, 3 rows x 34 px) for LDS -- one
//     round of global loads, one barrier;
//   * <= 72 MFMAs per wave, no loop-carried staging;
//   * K groups 1..3 hand their accumulators to group 0 through LDS (fixed order: deterministic),
//     the result is transposed through LDS and leaves as one 16-byte store per thread with
//     bias / activation / residual / ReLU-mask applied -- three barriers per launch in all.
template <bool DUAL>
__global__ __launch_bounds__(512) void conv3x3_oneshot_kernel(Conv3x3Args a) {
  constexpr int OCB = 64, KG = 4, MAXCH = 8;
  constexpr int IN_FLOATS = 3 * 2 * RS * 4;            // one chunk's patch: 3 rows x 2 halves x 34 slots
  constexpr int ITEMS_PER_CH = 3 * 2 * PW;             // 204 16-byte items
  constexpr int W_VEC4 = 9 * CK * OCB / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                                  // [MAXCH][IN_FLOATS]; later the epilogue stage [64][36]
  float* red = smem + MAXCH * IN_FLOATS;               // [2][KG-1][16][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int lh = lane >> 5, ll = lane & 31;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * TW, y0 = ty;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;
  const int nchunk = a.nchunk;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.c1 * hw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(DUAL ? a.x2 + (long long)n * a.x2_ns : a.x), 0,
      DUAL ? (a.cin - a.c1) * hw * 4 : 0, 0x00020000);

  // ---- weights of this wave's chunks -> registers (issued first: longest latency)
  const int cpw = (nchunk + KG - 1) / KG;              // chunks per K group (<= 2)
  const int c0 = wk * cpw;
  const f32x4* wlane = reinterpret_cast<const f32x4*>(a.wpk) + (lh * OCB + wn * 32 + ll);
  f32x4 aw[2][9];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int c = c0 + ci;
    const bool on = ci < cpw && c < nchunk;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (on) v = wlane[(size_t)c * W_VEC4 + tap * (2 * OCB)];
      aw[ci][tap] = v;
    }
  }
  // ---- the input patch of every chunk -> LDS
  const int total = nchunk * ITEMS_PER_CH;
  // two items per thread in flight: all their loads are issued before the first LDS store (a loop of
  // load -> store iterations pays the memory latency once per iteration)
  constexpr int MAXQ = (MAXCH * ITEMS_PER_CH + 511) / 512;     // 4
#pragma unroll
  for (int k0 = 0; k0 < MAXQ; k0 += 2) {
    f32x4 v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + (k0 + k) * 512;
      const int ch = q / ITEMS_PER_CH, rem = q - ch * ITEMS_PER_CH;
      const int r = rem / (2 * PW), rem2 = rem - r * (2 * PW);
      const int hf = rem2 / PW, col = rem2 - hf * PW;
      const int gy = y0 - 1 + r, gx = x0 - 1 + col;
      const bool ok = q < total && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      const unsigned base = ok ? (unsigned)(((ch * CK + 4 * hf) * hw + gy * a.w + gx) * 4) : OOB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned o1 = base + (unsigned)j * plane;
        float t = buf_load(rs1, o1);
        if (DUAL) t += buf_load(rs2, o1 - (unsigned)a.c1 * plane);
        v[k][j] = t;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + (k0 + k) * 512;
      const int ch = q / ITEMS_PER_CH, rem = q - ch * ITEMS_PER_CH;
      const int r = rem / (2 * PW), rem2 = rem - r * (2 * PW);
      const int hf = rem2 / PW, col = rem2 - hf * PW;
      if (q < total) *reinterpret_cast<f32x4*>(s_in + ch * IN_FLOATS + ((r * 2 + hf) * RS + col) * 4) = v[k];
    }
  }
  __syncthreads();

  // ---- MFMAs of this wave's K range
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int c = c0 + ci;
    if (ci < cpw && c < nchunk) {
      const float* si = s_in + c * IN_FLOATS + (lh * RS + ll) * 4;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(si + (ky * 2 * RS + kx) * 4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ci][tap][kk], bq[kk], acc, 0, 0, 0);
      }
    }
  }
  // ---- K groups 1..3 -> group 0 (fixed order), then the tile through LDS as [oc 64][36]
  if (wk > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wn * (KG - 1) + wk - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();                                      // all input reads are done as well
  constexpr int ES = 36;
  float* ep = smem;
  if (wk == 0) {
#pragma unroll
    for (int g = 0; g < KG - 1; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[((wn * (KG - 1) + g) * 16 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) ep[(wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ES + ll] = acc[r];
  }
  __syncthreads();
  const float slope = act_slope(a.act);
  const int py = y0;
  if (a.vec_ok) {
    const int oc = tid >> 3, p4 = (tid & 7) * 4;
    const int gx = x0 + p4;
    if (oc < a.cout && gx < a.w) {
      f32x4 v = *reinterpret_cast<const f32x4*>(ep + oc * ES + p4);
      const float bb = a.bias ? a.bias[oc] : 0.f;
      const long long off = (long long)oc * hw + (long long)py * a.w + gx;
      f32x4 rr = {0.f, 0.f, 0.f, 0.f};
      if (a.res) rr = *reinterpret_cast<const f32x4*>(a.res + (long long)n * a.res_ns + off);
      f32x4 mm = {1.f, 1.f, 1.f, 1.f};
      if (a.mask) mm = *reinterpret_cast<const f32x4*>(a.mask + (long long)n * a.mask_ns + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float q = v[e] + bb;
        q = (q >= 0.f ? q : q * slope + 0.f) + rr[e];
        v[e] = mm[e] > 0.f ? q : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.y + (long long)n * a.y_ns + off) = v;
    }
  } else {
    for (int i = tid; i < 64 * TW; i += 512) {
      const int oc = i >> 5, p = i & 31;
      const int gx = x0 + p;
      if (oc < a.cout && gx < a.w) {
        const long long off = (long long)oc * hw + (long long)py * a.w + gx;
        float q = ep[oc * ES + p] + (a.bias ? a.bias[oc] : 0.f);
        q = (q >= 0.f ? q : q * slope + 0.f) + (a.res ? a.res[(long long)n * a.res_ns + off] : 0.f);
        if (a.mask && a.mask[(long long)n * a.mask_ns + off] <= 0.f) q = 0.f;
        a.y[(long long)n * a.y_ns + off] = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// The one-shot scheme for a STRIDE-2 3x3 conv: y(oy, ox) = sum_k w[k] x(2 oy - 1 + ky, 2 ox - 1 + kx)
// (zero outside) -- the data gradient of ConvTranspose2d(k3, s2, p1, op1) (tecogan_nets.py:119-126)
// on the training frames.  Through its space-to-depth embedding that gradient is a 256-channel
// phased conv walking 32 channel chunks (27 us per launch for 2 x 32 x 32 outputs); directly it is a
// K = 576 contraction like any body layer: the patch is 3 input rows x 66 input columns per
// 8-channel chunk, the B operand of output pixel ox and tap kx sits at column 2 ox + kx.
constexpr int RS2 = 2 * TW + 2;                        // 66 patch columns
__global__ __launch_bounds__(512) void conv3x3s2_oneshot_kernel(Conv3x3Args a) {
  constexpr int OCB = 64, KG = 4, MAXCH = 8;
  constexpr int IN_FLOATS = 3 * 2 * RS2 * 4;           // one chunk's patch
  constexpr int ITEMS_PER_CH = 3 * 2 * RS2;            // 396 16-byte items
  constexpr int W_VEC4 = 9 * CK * OCB / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                                  // [MAXCH][IN_FLOATS]; later the epilogue stage [64][36]
  float* red = smem + MAXCH * IN_FLOATS;               // [2][KG-1][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int lh = lane >> 5, ll = lane & 31;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * TW, y0 = ty;
  const int hw = a.h * a.w;                            // OUTPUT plane
  const int ih = 2 * a.h, iw = 2 * a.w, ihw = ih * iw; // input plane
  const unsigned plane = (unsigned)ihw * 4u;
  const int nchunk = a.nchunk;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (long long)n * a.x_ns), 0, a.cin * ihw * 4, 0x00020000);
  const int cpw = (nchunk + KG - 1) / KG;
  const int c0 = wk * cpw;
  const f32x4* wlane = reinterpret_cast<const f32x4*>(a.wpk) + (lh * OCB + wn * 32 + ll);
  f32x4 aw[2][9];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int c = c0 + ci;
    const bool on = ci < cpw && c < nchunk;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (on) v = wlane[(size_t)c * W_VEC4 + tap * (2 * OCB)];
      aw[ci][tap] = v;
    }
  }
  const int total = nchunk * ITEMS_PER_CH;
  constexpr int MAXQ = (MAXCH * ITEMS_PER_CH + 511) / 512;     // 7
#pragma unroll
  for (int k0 = 0; k0 < MAXQ; k0 += 2) {
    f32x4 v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + (k0 + k) * 512;
      const int ch = q / ITEMS_PER_CH, rem = q - ch * ITEMS_PER_CH;
      const int r = rem / (2 * RS2), rem2 = rem - r * (2 * RS2);
      const int hf = rem2 / RS2, col = rem2 - hf * RS2;
      const int gy = 2 * y0 - 1 + r, gx = 2 * x0 - 1 + col;
      const bool ok = (k0 + k) < MAXQ && q < total && gy >= 0 && gy < ih && gx >= 0 && gx < iw;
      const unsigned base = ok ? (unsigned)(((ch * CK + 4 * hf) * ihw + gy * iw + gx) * 4) : OOB;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[k][j] = buf_load(rs1, base + (unsigned)j * plane);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + (k0 + k) * 512;
      const int ch = q / ITEMS_PER_CH, rem = q - ch * ITEMS_PER_CH;
      const int r = rem / (2 * RS2), rem2 = rem - r * (2 * RS2);
      const int hf = rem2 / RS2, col = rem2 - hf * RS2;
      if ((k0 + k) < MAXQ && q < total) *reinterpret_cast<f32x4*>(s_in + ch * IN_FLOATS + ((r * 2 + hf) * RS2 + col) * 4) = v[k];
    }
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int c = c0 + ci;
    if (ci < cpw && c < nchunk) {
      const float* si = s_in + c * IN_FLOATS + (lh * RS2 + 2 * ll) * 4;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(si + (ky * 2 * RS2 + kx) * 4);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ci][tap][kk], bq[kk], acc, 0, 0, 0);
      }
    }
  }
  if (wk > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wn * (KG - 1) + wk - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  constexpr int ES = 36;
  float* ep = smem;
  if (wk == 0) {
#pragma unroll
    for (int g = 0; g < KG - 1; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[((wn * (KG - 1) + g) * 16 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) ep[(wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ES + ll] = acc[r];
  }
  __syncthreads();
  const float slope = act_slope(a.act);
  for (int i = tid; i < 64 * TW; i += 512) {
    const int oc = i >> 5, p = i & 31;
    const int gx = x0 + p;
    if (oc < a.cout && gx < a.w) {
      const long long off = (long long)oc * hw + (long long)y0 * a.w + gx;
      float q = ep[oc * ES + p] + (a.bias ? a.bias[oc] : 0.f);
      q = (q >= 0.f ? q : q * slope + 0.f) + (a.res ? a.res[(long long)n * a.res_ns + off] : 0.f);
      if (a.mask && a.mask[(long long)n * a.mask_ns + off] <= 0.f) q = 0.f;
      a.y[(long long)n * a.y_ns + off] = q;
    }
  }
}

// 3 = LDS-transposed 16-byte epilogue + LDS-DMA weight staging (measured +2 % on the frame,
// parity suite green); bit 4 (start-up stagger of co-resident workgroups) measured null.
#ifndef TG_CONV_OPT
#define TG_CONV_OPT 3
#endif

template <int WM, int WN, int NT, int KS = 1>
static int launch_conv(const Conv3x3Args& a0, int n, hipStream_t stream) {
  Conv3x3Args a = a0;
  a.vec_ok = (a.w % 4 == 0) && ((uintptr_t)a.y % 16 == 0) && (a.y_ns % 4 == 0) &&
             (!a.res || (((uintptr_t)a.res % 16 == 0) && (a.res_ns % 4 == 0))) &&
             (!a.mask || (((uintptr_t)a.mask % 16 == 0) && (a.mask_ns % 4 == 0)));
  constexpr int OCB = WN * NT * 32;
  a.tiles_x = cdiv(a.w, TW);
  a.tiles_y = cdiv(a.h, WM);
  a.nocg = cdiv(a.cout, OCB);
  a.nchunk = cdiv(a.cin, CK);
  size_t lds = 2 * KS * (size_t)((WM + 2) * 2 * RS * 4 + 9 * CK * OCB) * sizeof(float);
  if (a.ksplit < 1) a.ksplit = 1;
  TG_REQUIRE(KS == 1 || a.ksplit == 1, TG_E_ARG, "conv3x3: in-workgroup K split excludes ksplit");
  if (KS > 1) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(
                              conv3x3_mfma_kernel<WM, WN, NT, true, 0, TG_CONV_OPT, 12, KS>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(
                              conv3x3_mfma_kernel<WM, WN, NT, false, 0, TG_CONV_OPT, 12, KS>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
  }
  long long blocks = (long long)a.tiles_x * a.tiles_y * a.nocg * n * a.ksplit;
  TG_REQUIRE(blocks > 0 && blocks < (1ll << 31), TG_E_SHAPE, "conv3x3: grid %lld", blocks);
  // XCD banding pays when a band is many tile rows deep; TG_CONV_XCD=0/1 overrides (lab)
  static const int xcd_env = TG_LAB_ENV("TG_CONV_XCD", -1);
  const bool xcd = xcd_env >= 0 ? xcd_env != 0 : blocks >= 512;
  a.nblocks = xcd ? (int)blocks : 0;
  const unsigned grid = xcd ? (unsigned)(8 * ((blocks + 7) / 8)) : (unsigned)blocks;
  if (a.x2)
    hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, NT, true, 0, TG_CONV_OPT, 12, KS>),
                       dim3(grid), dim3(WM * WN * KS * 64), lds, stream, a);
  else
    hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, NT, false, 0, TG_CONV_OPT, 12, KS>),
                       dim3(grid), dim3(WM * WN * KS * 64), lds, stream, a);
  return check_launch("conv3x3_mfma");
}

// out = act(sum_s part[s] + bias), optionally followed by MaxPool2d(2,2) (floor).
// The sum runs s = 0..S-1 in order: bit-reproducible.
__global__ void splitk_finalize_kernel(const float* __restrict__ part, int S, long long part_ss,
                                       const float* __restrict__ bias, float slope, int pool,
                                       float* __restrict__ y, int n, int c, int h, int w) {
  const int oh = pool ? h / 2 : h, ow = pool ? w / 2 : w;
  const long long total = (long long)n * c * oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int ox = (int)(i % ow); long long t = i / ow;
    int oy = (int)(t % oh); t /= oh;
    int ch = (int)(t % c);
    long long plane = t * h * (long long)w;   // (n*c + ch) plane offset
    float b = bias ? bias[ch] : 0.f;
    auto at = [&](int yy, int xx) {
      const float* p = part + plane + (long long)yy * w + xx;
      float v = 0.f;
      for (int s = 0; s < S; ++s) v += p[(long long)s * part_ss];
      v += b;
      return v >= 0.f ? v : v * slope + 0.f;
    };
    float v;
    if (pool) {
      v = fmaxf(fmaxf(at(2 * oy, 2 * ox), at(2 * oy, 2 * ox + 1)),
                fmaxf(at(2 * oy + 1, 2 * ox), at(2 * oy + 1, 2 * ox + 1)));
    } else {
      v = at(oy, ox);
    }
    y[i] = v;
  }
}

}  // namespace tg

namespace tg {
// the one-shot kernel: the K-split variant's shapes with cin <= 64 (<= 8 chunks) and one oc block
bool conv3x3_uses_oneshot(int n, int cin, int cout, int h, int w) {
  static const int env = TG_LAB_ENV("TG_CONV_ONESHOT", 1);
  return env && cout <= 64 && cdiv(cin, CK) <= 8 && conv3x3_uses_wg_ksplit(n, cin, cout, h, w);
}

bool conv3x3_uses_wg_ksplit(int n, int cin, int cout, int h, int w) {
  static const int ks_env = TG_LAB_ENV("TG_CONV_WG_KSPLIT", 1);
  if (!ks_env || tg_conv3x3_pick_ocb(cout) != 64 || conv3x3_rows_per_wg(64, (long long)n * h * w) != 2)
    return false;
  const long long wgs2 = (long long)cdiv(w, TW) * cdiv(h, 2) * cdiv(cout, 64) * n;
  return wgs2 <= 128 && cdiv(cin, CK) >= 6;
}
}  // namespace tg

using namespace tg;

// Split factor for layers whose tile count cannot fill 256 CUs (FNet's low-resolution,
// many-channel middle).  1 = no split.  Cost model fitted to rocprofv3 durations (us):
//   t(ks) = 4 + 1.15 * (nchunk / ks) * max(1, wgs * ks / 256)  [+ 4.5 for the finalize launch]
// -- a workgroup needs 1.15 us per 8-channel chunk (36 fp32 MFMAs at 2 GHz), workgroups beyond
// one per CU share the MFMA pipes, fixed costs are launch ramp + prologue + epilogue.  A split
// must win by 15 %: it also costs ks x the output size in partial-sum traffic.
extern "C" int tg_conv3x3_pick_ksplit(int n, int cin, int cout, int h, int w) {
  int ocb = tg_conv3x3_pick_ocb(cout);
  int rows = conv3x3_rows_per_wg(ocb, (long long)n * h * w);
  const double wgs = (double)cdiv(w, TW) * cdiv(h, rows) * cdiv(cout, ocb) * n;
  const int nchunk = cdiv(cin, CK);
  // the one-shot kernel already has the whole K range in flight inside ONE launch: 10.8 us against
  // 9.0 + 4.2 us for a 4-way split + finalize on the 2 x 32 x 32 training frames (rocprofv3, round 3)
  if (conv3x3_uses_oneshot(n, cin, cout, h, w)) return 1;
  static const int legacy = TG_LAB_ENV("TG_KSPLIT_LEGACY", 0);
  if (legacy) {   // lab: the round-1 rule (fill ~640 workgroup slots)
    if (wgs >= 400 || nchunk < 4) return 1;
    int ks = 1;
    while (ks < 8 && wgs * ks < 640 && nchunk / (ks * 2) >= 2) ks *= 2;
    return ks;
  }
  auto cost = [&](int ks) {
    double share = wgs * ks / 256.0;
    return 4.0 + 1.15 * ((double)nchunk / ks) * (share > 1.0 ? share : 1.0) + (ks > 1 ? 4.5 : 0.0);
  };
  int best = 1;
  double tbest = cost(1);
  for (int ks = 2; ks <= 8 && nchunk / ks >= 2; ks *= 2) {
    double t = cost(ks);
    if (t < 0.85 * tbest) { best = ks; tbest = t; }
  }
  return best;
}

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

static int conv3x3_impl(const float* x, int64_t x_nstride, int c1, const float* x2,
                        int64_t x2_nstride, const float* w_packed, int ocb, const float* bias,
                        const float* res, int64_t res_nstride, float* y, int64_t y_nstride, int n,
                        int cin, int cout, int h, int w, int act, int ksplit, float* partials,
                        tg_stream_t stream, const float* mask = nullptr, int64_t mask_nstride = 0,
                        int tapsel = 0, int cphase = 0, int set_p0 = 0, int set_p1 = 0) {
  TG_REQUIRE(x && w_packed && y, TG_E_ARG, "conv3x3_fwd: null pointer");
  TG_REQUIRE(!mask || ksplit <= 1, TG_E_ARG, "conv3x3_fwd: the ReLU mask is applied in the epilogue (no split-K)");
  TG_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, TG_E_SHAPE,
             "conv3x3_fwd: n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, w);
  TG_REQUIRE(c1 > 0 && c1 <= cin && (c1 == cin || x2), TG_E_ARG,
             "conv3x3_fwd: c1=%d cin=%d needs a second source", c1, cin);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_LRELU02, TG_E_ARG,
             "conv3x3_fwd: act=%d (none|relu|lrelu; tanh*24 only in tg_conv3x3_small_fwd)", act);
  TG_REQUIRE(ocb == 32 || ocb == 64, TG_E_ARG, "conv3x3_fwd: ocb=%d", ocb);
  TG_REQUIRE((long long)(cin + CK) * h * w * 4 < (1ll << 31), TG_E_SHAPE,
             "conv3x3_fwd: one batch item must be < 2 GiB (cin=%d h=%d w=%d)", cin, h, w);
  Conv3x3Args a{};
  a.x = x; a.x2 = (c1 < cin) ? x2 : nullptr; a.wpk = w_packed; a.bias = bias; a.res = res;
  a.y = y; a.x_ns = x_nstride; a.x2_ns = x2_nstride; a.res_ns = res_nstride; a.y_ns = y_nstride;
  a.c1 = c1; a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = act;
  a.ksplit = ksplit; a.part = partials; a.part_ss = (long long)n * cout * h * w;
  a.mask = mask; a.mask_ns = mask_nstride;
  a.tapsel = tapsel; a.cphase = cphase; a.rowsets[0] = (unsigned char)set_p0; a.rowsets[1] = (unsigned char)set_p1;
  if (tapsel) {
    TG_REQUIRE((tapsel == 1 || tapsel == 2) && cphase > 0 && cphase % CK == 0 && set_p0 >= 0 && set_p0 <= 3 &&
                   set_p1 >= 0 && set_p1 <= 3, TG_E_ARG, "conv3x3: tapsel=%d cphase=%d sets %d/%d", tapsel,
               cphase, set_p0, set_p1);
    TG_REQUIRE(tapsel == 1 ? cin == 4 * cphase : (cout == 4 * cphase && cphase % 64 == 0), TG_E_SHAPE,
               "conv3x3: phased taps need 4 phases of %d channels (cin=%d cout=%d)", cphase, cin, cout);
  }
  hipStream_t s = (hipStream_t)stream;
  if (ocb == 32) return launch_conv<4, 1, 1>(a, n, s);
  // 64 output channels per workgroup.  Small images get the 2-row tile so that
  // more workgroups exist (tile quantisation dominates there).
  if (conv3x3_rows_per_wg(ocb, (long long)n * h * w) == 4) return launch_conv<4, 1, 2>(a, n, s);
  // At most half as many 2-row tiles as CUs (e.g. a training batch of 2 x 64 x 64): 1-row tiles
  // with the channel chunks split over two wave groups keep every SIMD busy in ONE launch.
  if (a.ksplit <= 1 && !tapsel && conv3x3_uses_oneshot(n, cin, cout, h, w)) {
    a.vec_ok = (a.w % 4 == 0) && ((uintptr_t)a.y % 16 == 0) && (a.y_ns % 4 == 0) &&
               (!a.res || (((uintptr_t)a.res % 16 == 0) && (a.res_ns % 4 == 0))) &&
               (!a.mask || (((uintptr_t)a.mask % 16 == 0) && (a.mask_ns % 4 == 0)));
    a.tiles_x = cdiv(a.w, TW); a.tiles_y = a.h; a.nocg = 1; a.nchunk = cdiv(a.cin, CK);
    const size_t lds = (size_t)(8 * 3 * 2 * RS * 4 + 2 * 3 * 16 * 64) * sizeof(float);    // 50.7 KB
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_oneshot_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_oneshot_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    const unsigned grid = (unsigned)(a.tiles_x * a.tiles_y * n);
    if (a.x2) hipLaunchKernelGGL(conv3x3_oneshot_kernel<true>, dim3(grid), dim3(512), lds, s, a);
    else hipLaunchKernelGGL(conv3x3_oneshot_kernel<false>, dim3(grid), dim3(512), lds, s, a);
    return check_launch("conv3x3_oneshot");
  }
  if (a.ksplit <= 1 && conv3x3_uses_wg_ksplit(n, cin, cout, h, w)) return launch_conv<1, 2, 1, 2>(a, n, s);
  return launch_conv<2, 2, 1>(a, n, s);
}

extern "C" int tg_conv3x3_fwd(const float* x, int64_t x_nstride, int c1, const float* x2,
                              int64_t x2_nstride, const float* w_packed, int ocb,
                              const float* bias, const float* res, int64_t res_nstride,
                              float* y, int64_t y_nstride, int n, int cin, int cout, int h,
                              int w, int act, tg_stream_t stream) {
  return conv3x3_impl(x, x_nstride, c1, x2, x2_nstride, w_packed, ocb, bias, res, res_nstride, y,
                      y_nstride, n, cin, cout, h, w, act, 1, nullptr, stream);
}

extern "C" int tg_conv3x3_fwd_phased(const float* x, int64_t x_nstride, const float* w_packed, int ocb,
                                     const float* bias, float* y, int64_t y_nstride, int n, int cin,
                                     int cout, int h, int w, int act, int tapsel, int cphase,
                                     int taps_phase0, int taps_phase1, tg_stream_t stream) {
  TG_REQUIRE(tapsel == 1 || tapsel == 2, TG_E_ARG, "conv3x3_fwd_phased: tapsel=%d (1 input | 2 output phases)", tapsel);
  return conv3x3_impl(x, x_nstride, cin, nullptr, 0, w_packed, ocb, bias, nullptr, 0, y, y_nstride, n, cin,
                      cout, h, w, act, 1, nullptr, stream, nullptr, 0, tapsel, cphase, taps_phase0,
                      taps_phase1);
}

extern "C" int tg_conv3x3_fwd_phased_masked(const float* x, int64_t x_nstride, const float* w_packed, int ocb,
                                            const float* bias, const float* relu_mask, int64_t mask_nstride,
                                            float* y, int64_t y_nstride, int n, int cin, int cout, int h,
                                            int w, int act, int tapsel, int cphase, int taps_phase0,
                                            int taps_phase1, tg_stream_t stream) {
  TG_REQUIRE(tapsel == 1 || tapsel == 2, TG_E_ARG, "conv3x3_fwd_phased: tapsel=%d (1 input | 2 output phases)", tapsel);
  return conv3x3_impl(x, x_nstride, cin, nullptr, 0, w_packed, ocb, bias, nullptr, 0, y, y_nstride, n, cin,
                      cout, h, w, act, 1, nullptr, stream, relu_mask, mask_nstride, tapsel, cphase, taps_phase0,
                      taps_phase1);
}

// Split factor for a phased launch that cannot fill the device (the critic's deeper 4x4 / s2 blocks:
// 48-192 workgroups walking 32-64 channel chunks, each chunk a full load -> LDS -> MFMA round trip
// with nothing else resident on the CU -- 67 us for 0.8 GFLOP at crop 128).  1 = no split.
extern "C" int tg_conv3x3_phased_pick_ksplit(int n, int cin, int cout, int h, int w, int ocb) {
  if (n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || (ocb != 32 && ocb != 64)) return 1;
  const int rows = conv3x3_rows_per_wg(ocb, (long long)n * h * w);
  const long long wgs = (long long)cdiv(w, TW) * cdiv(h, rows) * cdiv(cout, ocb) * n;
  const int nchunk = cdiv(cin, CK);
  if (wgs >= 384 || nchunk < 16) return 1;
  int ks = 2;
  while (ks < 8 && wgs * ks * 2 <= 1024 && nchunk / (ks * 2) >= 4) ks *= 2;
  return ks;
}

extern "C" int tg_conv3x3_fwd_phased_splitk(const float* x, int64_t x_nstride, const float* w_packed, int ocb,
                                            const float* bias, float* y, int n, int cin, int cout, int h, int w,
                                            int act, int tapsel, int cphase, int taps_phase0, int taps_phase1,
                                            int ksplit, float* partials, tg_stream_t stream) {
  TG_REQUIRE(y && partials, TG_E_ARG, "conv3x3_fwd_phased_splitk: null pointer");
  TG_REQUIRE(tapsel == 1 || tapsel == 2, TG_E_ARG, "conv3x3_fwd_phased_splitk: tapsel=%d", tapsel);
  TG_REQUIRE(ksplit >= 2 && ksplit <= 16 && ksplit <= cdiv(cin, CK), TG_E_ARG,
             "conv3x3_fwd_phased_splitk: ksplit=%d for cin=%d", ksplit, cin);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_LRELU02, TG_E_ARG, "conv3x3_fwd_phased_splitk: act=%d", act);
  int rc = conv3x3_impl(x, x_nstride, cin, nullptr, 0, w_packed, ocb, nullptr, nullptr, 0, partials,
                        (int64_t)cout * h * w, n, cin, cout, h, w, TG_ACT_NONE, ksplit, partials, stream, nullptr, 0,
                        tapsel, cphase, taps_phase0, taps_phase1);
  if (rc != TG_OK) return rc;
  return conv3x3_splitk_finalize(partials, ksplit, bias, act, 0, y, n, cout, h, w, stream);
}

// shapes the stride-2 one-shot kernel takes: one 64-channel block, cin <= 64, at most 1024 one-row tiles
// (two workgroups per CU: two rounds; measured against the phased form on s2d(x) in tools/time_ops.py)
static const int S2_MAX_TILES = 1024;
extern "C" int tg_conv3x3s2_supported(int n, int cin, int cout, int h_out, int w_out) {
  if (n <= 0 || cin <= 0 || cin > 64 || cout <= 0 || cout > 64 || h_out <= 0 || w_out <= 0) return 0;
  return (long long)n * h_out * cdiv(w_out, TW) <= S2_MAX_TILES ? 1 : 0;
}

extern "C" int tg_conv3x3s2_fwd(const float* x, int64_t x_nstride, const float* w_packed, const float* bias,
                                const float* relu_mask, int64_t mask_nstride, float* y, int64_t y_nstride, int n,
                                int cin, int cout, int h_out, int w_out, int act, tg_stream_t stream) {
  TG_REQUIRE(x && w_packed && y, TG_E_ARG, "conv3x3s2_fwd: null pointer");
  TG_REQUIRE(tg_conv3x3s2_supported(n, cin, cout, h_out, w_out), TG_E_SHAPE,
             "conv3x3s2_fwd: n=%d cin=%d cout=%d out %dx%d (cin, cout <= 64, <= 1024 row tiles)", n, cin, cout, h_out, w_out);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_LRELU02, TG_E_ARG, "conv3x3s2_fwd: act=%d", act);
  TG_REQUIRE((long long)(cin + CK) * 4 * h_out * w_out * 4 < (1ll << 31), TG_E_SHAPE, "conv3x3s2_fwd: item too large");
  Conv3x3Args a{};
  a.x = x; a.wpk = w_packed; a.bias = bias; a.y = y; a.x_ns = x_nstride; a.y_ns = y_nstride;
  a.mask = relu_mask; a.mask_ns = mask_nstride;
  a.c1 = cin; a.cin = cin; a.cout = cout; a.h = h_out; a.w = w_out; a.act = act;
  a.tiles_x = cdiv(w_out, TW); a.tiles_y = h_out; a.nocg = 1; a.nchunk = cdiv(cin, CK);
  const size_t lds = (size_t)(8 * 3 * 2 * RS2 * 4 + 2 * 3 * 16 * 64) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3s2_oneshot_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(conv3x3s2_oneshot_kernel, dim3((unsigned)(a.tiles_x * a.tiles_y * n)), dim3(512), lds,
                     (hipStream_t)stream, a);
  return check_launch("conv3x3s2_oneshot");
}

extern "C" int tg_conv3x3_fwd_masked(const float* x, int64_t x_nstride, int c1, const float* x2,
                                     int64_t x2_nstride, const float* w_packed, int ocb,
                                     const float* bias, const float* res, int64_t res_nstride,
                                     const float* relu_mask, int64_t mask_nstride, float* y,
                                     int64_t y_nstride, int n, int cin, int cout, int h, int w,
                                     int act, tg_stream_t stream) {
  return conv3x3_impl(x, x_nstride, c1, x2, x2_nstride, w_packed, ocb, bias, res, res_nstride, y,
                      y_nstride, n, cin, cout, h, w, act, 1, nullptr, stream, relu_mask, mask_nstride);
}

namespace tg {
int conv3x3_splitk_conv(const float* x, int64_t x_nstride, int c1, const float* x2,
                        int64_t x2_nstride, const float* w_packed, int ocb, int n, int cin,
                        int cout, int h, int w, int ksplit, float* partials, tg_stream_t stream) {
  return conv3x3_impl(x, x_nstride, c1, x2, x2_nstride, w_packed, ocb, nullptr, nullptr, 0,
                      partials, (int64_t)cout * h * w, n, cin, cout, h, w, TG_ACT_NONE, ksplit,
                      partials, stream);
}
int conv3x3_splitk_finalize(const float* partials, int ksplit, const float* bias, int act, int pool,
                            float* y, int n, int cout, int h, int w, tg_stream_t stream) {
  long long total = (long long)n * cout * (pool ? (h / 2) * (w / 2) : h * w);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     partials, ksplit, (long long)n * cout * h * w, bias, act_slope_host(act), pool,
                     y, n, cout, h, w);
  return check_launch("splitk_finalize");
}
}  // namespace tg

extern "C" int tg_conv3x3_splitk_fwd(const float* x, int64_t x_nstride, int c1, const float* x2,
                                     int64_t x2_nstride, const float* w_packed, int ocb,
                                     const float* bias, float* y, int n, int cin, int cout, int h,
                                     int w, int act, int ksplit, float* partials, int pool,
                                     tg_stream_t stream) {
  TG_REQUIRE(y && partials, TG_E_ARG, "conv3x3_splitk_fwd: null pointer");
  TG_REQUIRE(ksplit >= 2 && ksplit <= 16 && ksplit <= cdiv(cin, CK), TG_E_ARG,
             "conv3x3_splitk_fwd: ksplit=%d for cin=%d", ksplit, cin);
  TG_REQUIRE(act >= TG_ACT_NONE && act <= TG_ACT_LRELU02, TG_E_ARG, "conv3x3_splitk_fwd: act=%d", act);
  TG_REQUIRE(!pool || (h >= 2 && w >= 2), TG_E_SHAPE, "conv3x3_splitk_fwd: pool needs h,w >= 2");
  int rc = conv3x3_splitk_conv(x, x_nstride, c1, x2, x2_nstride, w_packed, ocb, n, cin, cout, h, w,
                               ksplit, partials, stream);
  if (rc != TG_OK) return rc;
  return conv3x3_splitk_finalize(partials, ksplit, bias, act, pool, y, n, cout, h, w, stream);
}

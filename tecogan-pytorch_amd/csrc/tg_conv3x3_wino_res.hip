// SRNet's conv_in + residual-block convs (tecogan_nets.py:85-100, 108-116, 141-143) of ONE
// 134x320-class frame as ONE launch of PERSISTENT, LDS-RESIDENT workgroups (round 4).
//
// Why: the per-layer Winograd kernel (tg_conv3x3_wino.hip) spends ~10 of its ~23 us per layer on
// cost every one of its 670 co-resident workgroups pays in lockstep -- launch, first loads of the
// 11 MB input, the 11 MB of stores nothing overlaps, drain.  A 64-channel
// 134x320 activation is 11 MB = 43 KB per CU: it FITS in the LDS of the 256 CUs.  So here
//   * one workgroup per CU owns a block of 8 x 24 pixels (4 x 12 Winograd tiles = three MFMA groups
//     of 16 tiles) for ALL the layers; the block (+ a one-pixel ring) of the current and of the next
//     layer live in two LDS buffers of 64 x 10 x 28 floats (ping-pong; the residual input of a
//     ResidualBlock is what the destination buffer still holds: x -> conv1 -> other buffer ->
//     conv2 -> += x in place);
//   * between two layers only the block's outermost ring of pixels travels: 64 pixels x 64 channels
//     = 16 KB per workgroup through an exchange buffer in global memory (agent-scope 16-byte stores,
//     one monotonic flag per workgroup, 16-byte agent-scope loads of the <= 8 neighbours' rings)
//     instead of 11 MB written + 11 MB read per layer;
//   * 12 waves = 3 tile groups x 4 output-channel blocks of 16 (the three waves that share a SIMD
//     share a weight slice, so their loads hit L1); a wave builds its MFMA B operand (the
//     transformed input window B^T d B) in REGISTERS straight from the resident block -- no V tensor
//     in LDS, no barrier inside the K loop (the per-layer kernel has 8);
//   * same arithmetic in the same order as conv3x3_wino_kernel (same packed U, K ascending, same
//     transform / inverse-transform expressions): results are BIT-IDENTICAL to the per-layer
//     launches (tests/test_hip_parity.py::test_wino_resident_*).
// Forward progress needs every workgroup resident at once (neighbours wait for each other in both
// directions): the launcher refuses grids above the device's CU count minus a margin, and the kernel
// is FAIL-SAFE like the other chained launches (poll limit -> fault counter in pinned host memory
// -> the plan reports TG_E_HIP and runs one launch per layer for good).
#include <type_traits>

#include "tg_common.h"

#ifndef TG_WRES_LAB
#define TG_WRES_LAB 0   // 1: ablation switches (env TG_WRES_ABL) compiled in -- tools/build_lab_libs.sh, timing only
#endif
#define RABL(bit) (TG_WRES_LAB && (a.abl & (bit)))

namespace tg {

constexpr int WR_TH = 4, WR_TW = 12;             // Winograd tiles per block
constexpr int WR_BH = 2 * WR_TH, WR_BW = 2 * WR_TW;   // 8 x 24 pixels
constexpr int WR_RS = 28;                         // LDS row stride (floats): ring + 24 + ring + 2 pad
constexpr int WR_CS = 288;                        // channel stride: 10 rows x 28 = 280 -> 288 = 32 mod 64:
                                                  // the 32 lanes of a ds_read_b64 group (2 channels x 16 tiles) hit 64 distinct banks
constexpr int WR_NC = 64;                         // channels (nf)
constexpr int WR_THREADS = 768;                   // 12 waves: 3 per SIMD
constexpr int WR_MAXL = 24;
constexpr int WR_SLOTS = 64;                      // published ring pixels of a block: top 24, bottom 24, left 8, right 8
constexpr int WR_SC1 = 16;                        // agent-scope cache policy bit of the buffer instructions
constexpr size_t WR_LDS_BYTES = (size_t)2 * WR_NC * WR_CS * sizeof(float);   // 147 456

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#if TG_WRES_LAB
// lab: s_memtime stamps of every wave of workgroup 100, 8 per layer (tools/wino_res_lab.py --stamps)
__device__ long long g_wres_dbg[12 * 24 * 8];
#define RSTAMP(k) do { if (blockIdx.x == 100 && l == 0) g_wres_dbg[(wv * 24 + L) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define RSTAMP(k) do { } while (0)
#endif

struct WResLayer {
  const float* u;      // tg_pack_conv3x3_wino form
  const float* bias;
  int act;
  int res;             // + the destination buffer's old content (residual input) after the activation
  int nks;             // K steps of 4 input channels (4 * ceil(cin / 16))
  int pad;
};
struct WResArgs {
  WResLayer L[WR_MAXL];
  const float* x;      // first layer's input, channels [0, c1)
  const float* x2;     // channels [c1, cin0) or null
  float* y;            // last layer's output (64 x h x w)
  float* xbuf;         // exchange: [parity 2][block][slot 64][channel 64]
  unsigned* flags;     // [block], monotonic: base + layers finished
  int* err;            // fault counter (pinned host memory or device memory)
  unsigned base;       // flag value before this launch's first layer
  int poll_limit;
  int nlayer, h, w, nbx, nby, c1, cin0;
  int abl;             // lab builds only: 1 no flag wait / ring loads, 2 no MFMA, 4 no weight loads, 8 no ring stores, 16 no window reads, 32 no hand-over at all
};

__global__ __launch_bounds__(WR_THREADS, 3) void conv3x3_wino_resident_kernel(WResArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_act[];    // [2][64][WR_CS]
  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q = wv & 3;                       // output channels [16q, 16q + 16); waves q, q+4, q+8 share a SIMD and a weight slice
  const int g = wv >> 2;                      // tile group
  const int wg = blockIdx.x;
  const int bx = __builtin_amdgcn_readfirstlane(wg % a.nbx), by = __builtin_amdgcn_readfirstlane(wg / a.nbx);
  const int X0 = bx * WR_BW, Y0 = by * WR_BH;
  const int hw = a.h * a.w;
  const int nwg = a.nbx * a.nby;

  // ---- zero both buffers (rings at the image border and pixels outside the image stay 0) ----
  {
    f32x4* z = reinterpret_cast<f32x4*>(s_act);
    for (int i = t; i < 2 * WR_NC * WR_CS / 4; i += WR_THREADS) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  // ---- first layer's input block (+ ring) from global memory into buffer 0 ---------------
  // (bounds-checked buffer loads: zero padding, and the channels of the other source tensor, read 0;
  // every load of a batch is in flight before the first LDS store)
  {
    constexpr int WIN = (WR_BH + 2) * (WR_BW + 2);       // 260 pixels per channel
    constexpr unsigned OOB = 0x80000000u;
    const int total = a.cin0 * WIN;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)a.c1 * hw * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x2 ? a.x2 : a.x), 0, a.x2 ? (unsigned)(a.cin0 - a.c1) * hw * 4u : 0u, 0x00020000);
    constexpr int BATCH = 11;
    for (int e0 = 0; e0 < total; e0 += BATCH * WR_THREADS) {
      float v[BATCH];
      int lo[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int e = e0 + k * WR_THREADS + t;
        const int ic = e / WIN, rem = e - ic * WIN, r = rem / (WR_BW + 2), c = rem - r * (WR_BW + 2);
        const int gy = Y0 - 1 + r, gx = X0 - 1 + c;
        const bool in = e < total && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
        const unsigned px = (unsigned)(gy * a.w + gx) * 4u;
        const unsigned o1 = (in && ic < a.c1) ? (unsigned)ic * hw * 4u + px : OOB;
        const unsigned o2 = (in && ic >= a.c1) ? (unsigned)(ic - a.c1) * hw * 4u + px : OOB;
        const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (int)o1, 0, 0));
        const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, (int)o2, 0, 0));
        v[k] = ic < a.c1 ? v1 : v2;
        lo[k] = e < total ? ic * WR_CS + r * WR_RS + c : -1;
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k)
        if (lo[k] >= 0) s_act[lo[k]] = v[k];
    }
  }
  __syncthreads();

  // ---- per-lane geometry ---------------------------------------------------------------------
  const int T = 16 * g + (l & 15);            // tile of the block: row-major 4 x 12
  const int ty = T / WR_TW, tx = T - ty * WR_TW;
  const int kk = l >> 4;                      // K index inside a K step (B operand) / output-channel quad (D)
  const int rb = kk * WR_CS + (2 * ty) * WR_RS + 2 * tx;        // window origin in the resident block (floats)
  const int oc_base = 16 * q + 4 * kk;
  const int wb = oc_base * WR_CS + (2 * ty + 1) * WR_RS + 2 * tx + 1;   // own 2x2 pixels, channel oc_base
  const int gy0 = Y0 + 2 * ty, gx0 = X0 + 2 * tx;               // image position of the tile
  const bool live = gy0 < a.h && gx0 < a.w;                     // h, w even: a tile is inside or outside as a whole
  const bool e_top = ty == 0, e_bot = ty == WR_TH - 1, e_lft = tx == 0, e_rgt = tx == WR_TW - 1;
  const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc(
      a.xbuf, 0, (unsigned)(2u * nwg * WR_SLOTS * WR_NC * 4u), 0x00020000);

  const size_t ulane = (size_t)(q * 4) * 64 + l;               // this lane's 16 bytes inside a K step's block
  constexpr size_t USTEP = (size_t)4 * 4 * 64;                  // f32x4 per K step (64 output channels)
  auto load_u = [&](const f32x4* ub, int ks, int nks, f32x4 (&u)[4]) {
    if (ks >= nks || (RABL(4) && ks > 1)) return;
    const f32x4* p = ub + (size_t)ks * USTEP;
    u[0] = p[0]; u[1] = p[64]; u[2] = p[128]; u[3] = p[192];
  };

  f32x4 u0[4], u1[4];
  {
    const f32x4* ub = reinterpret_cast<const f32x4*>(a.L[0].u) + ulane;
    load_u(ub, 0, a.L[0].nks, u0);
    load_u(ub, 1, a.L[0].nks, u1);
  }

  for (int L = 0; L < a.nlayer; ++L) {
    const WResLayer& lay = a.L[L];
    const float* src = s_act + (L & 1) * (WR_NC * WR_CS);
    float* dst = s_act + ((L & 1) ^ 1) * (WR_NC * WR_CS);
    const f32x4* ub = reinterpret_cast<const f32x4*>(lay.u) + ulane;
    const int nks = lay.nks;

    RSTAMP(0);
    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one K step: window of channel 4 ks + kk -> B^T d B in registers -> 16 MFMAs.  (fp32 MFMAs and VALU
    // instructions of the waves of a SIMD do NOT overlap on gfx950 -- tools/valu_lab.hip: 16 MFMAs + 32
    // adds take exactly the sum of both -- so the order inside a step matters little; a burst form with
    // the next window prefetched measured slower: 24.4 vs 21.5 us per layer, it costs registers.)
    auto kstep = [&](int ks, const f32x4 (&u)[4]) {
      const float* sp = src + rb + (RABL(16) ? 0 : ks * (4 * WR_CS));
      float d[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float2 p0 = *reinterpret_cast<const float2*>(sp + r * WR_RS);
        const float2 p1 = *reinterpret_cast<const float2*>(sp + r * WR_RS + 2);
        d[r][0] = p0.x; d[r][1] = p0.y; d[r][2] = p1.x; d[r][3] = p1.y;
      }
      f32x4 bq[4];
      {
        float qa[4], qb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { qa[c] = d[0][c] - d[2][c]; qb[c] = d[1][c] + d[2][c]; }
        bq[0] = f32x4{qa[0] - qa[2], qa[1] + qa[2], qa[2] - qa[1], qa[1] - qa[3]};
        bq[1] = f32x4{qb[0] - qb[2], qb[1] + qb[2], qb[2] - qb[1], qb[1] - qb[3]};
#pragma unroll
        for (int c = 0; c < 4; ++c) { qa[c] = d[2][c] - d[1][c]; qb[c] = d[1][c] - d[3][c]; }
        bq[2] = f32x4{qa[0] - qa[2], qa[1] + qa[2], qa[2] - qa[1], qa[1] - qa[3]};
        bq[3] = f32x4{qb[0] - qb[2], qb[1] + qb[2], qb[2] - qb[1], qb[1] - qb[3]};
      }
      if (RABL(2)) { acc[0] += bq[0] + bq[1] + bq[2] + bq[3] + u[0] + u[1] + u[2] + u[3]; return; }
#pragma unroll
      for (int p = 0; p < 16; ++p)
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[p >> 2][p & 3], bq[p >> 2][p & 3], acc[p], 0, 0, 0);
    };

    for (int ks = 0; ks < nks; ks += 2) {
      kstep(ks, u0);
      __builtin_amdgcn_sched_barrier(0);
      load_u(ub, ks + 2, nks, u0);
      kstep(ks + 1, u1);
      __builtin_amdgcn_sched_barrier(0);
      load_u(ub, ks + 3, nks, u1);
    }
    RSTAMP(1);
    // ---- inverse transform A^T m A, bias / activation / residual -----------------------------
    const bool last = L + 1 == a.nlayer;
    const float slope = act_slope(lay.act);
    float bz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bz[r] = lay.bias[oc_base + r];
    float v[4][2][2];                         // [channel r][row i][column j]
    auto epilogue = [&](auto has_res) {
      constexpr bool RES = decltype(has_res)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sr[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sr[0][j] = (acc[0 + j][r] + acc[4 + j][r]) + acc[8 + j][r];
          sr[1][j] = (acc[4 + j][r] - acc[8 + j][r]) - acc[12 + j][r];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v0 = ((sr[i][0] + sr[i][1]) + sr[i][2]) + bz[r];
          float v1 = ((sr[i][1] - sr[i][2]) - sr[i][3]) + bz[r];
          v0 = v0 >= 0.f ? v0 : v0 * slope;
          v1 = v1 >= 0.f ? v1 : v1 * slope;
          if constexpr (RES) {                // the residual input: what the destination buffer still holds
            v0 += dst[wb + r * WR_CS + i * WR_RS];
            v1 += dst[wb + r * WR_CS + i * WR_RS + 1];
          }
          v[r][i][0] = v0; v[r][i][1] = v1;
        }
      }
    };
    if (lay.res) epilogue(std::true_type{}); else epilogue(std::false_type{});
    // the next layer's first weights travel under the hand-over
    if (!last) {
      const f32x4* un = reinterpret_cast<const f32x4*>(a.L[L + 1].u) + ulane;
      load_u(un, 0, a.L[L + 1].nks, u0);
      load_u(un, 1, a.L[L + 1].nks, u1);
    }
    if (live) {
      if (last) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            *reinterpret_cast<float2*>(a.y + (size_t)(oc_base + r) * hw + (size_t)(gy0 + i) * a.w + gx0) =
                make_float2(v[r][i][0], v[r][i][1]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            dst[wb + r * WR_CS + i * WR_RS] = v[r][i][0];
            dst[wb + r * WR_CS + i * WR_RS + 1] = v[r][i][1];
          }
        // publish the block's outermost pixels: slot pixel-major, 4 consecutive channels = one 16-byte store
        const unsigned pb = ((unsigned)((L & 1) * nwg + wg) * WR_SLOTS * WR_NC + oc_base) * 4u;
        auto pub = [&](int slot, int i, int j) {
          const u32x4 dd = {__builtin_bit_cast(unsigned, v[0][i][j]), __builtin_bit_cast(unsigned, v[1][i][j]),
                            __builtin_bit_cast(unsigned, v[2][i][j]), __builtin_bit_cast(unsigned, v[3][i][j])};
          if (RABL(8) && dd[0] != 0x12345678u) return;
          __builtin_amdgcn_raw_buffer_store_b128(dd, rxb, (int)(pb + (unsigned)slot * (WR_NC * 4u)), 0, WR_SC1);
        };
        if (e_top) { pub(2 * tx, 0, 0); pub(2 * tx + 1, 0, 1); }
        if (e_bot) { pub(WR_BW + 2 * tx, 1, 0); pub(WR_BW + 2 * tx + 1, 1, 1); }
        if (e_lft) { pub(2 * WR_BW + 2 * ty, 0, 0); pub(2 * WR_BW + 2 * ty + 1, 1, 0); }
        if (e_rgt) { pub(2 * WR_BW + WR_BH + 2 * ty, 0, 1); pub(2 * WR_BW + WR_BH + 2 * ty + 1, 1, 1); }
      }
    }
    RSTAMP(2);
    if (last) break;
    if (RABL(32)) { __syncthreads(); continue; }

    // ---- hand-over: ring stores acknowledged (every wave), flag, neighbours' flags, their rings ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RSTAMP(3);
    __syncthreads();                          // also: every read of src and every write of dst of this layer is done
    RSTAMP(4);
    const unsigned target = a.base + (unsigned)(L + 1);
    if (t == 0) __hip_atomic_store(a.flags + wg, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t >= 64 && t < 72 && !RABL(1)) {
      const int d8 = t - 64, dd = d8 + (d8 >= 4);          // 0..8 without the centre
      const int ny = by - 1 + dd / 3, nx = bx - 1 + dd % 3;
      if (ny >= 0 && ny < a.nby && nx >= 0 && nx < a.nbx) {
        const unsigned* f = a.flags + ny * a.nbx + nx;
        int polls = 0;
        bool fault = a.poll_limit < 0;
        while (!fault && (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          __builtin_amdgcn_s_sleep(2);
          fault = ++polls > a.poll_limit;
        }
        if (fault) __hip_atomic_fetch_add(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();
    RSTAMP(5);
    // ring of dst: 68 pixels x 16 channel quads, one 16-byte agent-scope load each
    {
      constexpr int NRING = 2 * (WR_BW + 2) + 2 * WR_BH;           // 68
      constexpr int ITEMS = NRING * (WR_NC / 4);                   // 1088
      constexpr int PER_T = (ITEMS + WR_THREADS - 1) / WR_THREADS; // 2
      f32x4 hv[PER_T];
      int ho[PER_T];
      int tt = t;
      asm volatile("" : "+v"(tt));            // keeps this geometry out of the K loop's register budget (it would be hoisted and spilled)
#pragma unroll
      for (int k = 0; k < PER_T; ++k) {
        const int item = tt + k * WR_THREADS;
        const int pi = item >> 4, c4 = item & 15;
        int ry, rx;
        if (pi < WR_BW + 2) { ry = 0; rx = pi; }
        else if (pi < 2 * (WR_BW + 2)) { ry = WR_BH + 1; rx = pi - (WR_BW + 2); }
        else if (pi < 2 * (WR_BW + 2) + WR_BH) { ry = pi - 2 * (WR_BW + 2) + 1; rx = 0; }
        else { ry = pi - 2 * (WR_BW + 2) - WR_BH + 1; rx = WR_BW + 1; }
        const int gy = Y0 - 1 + ry, gx = X0 - 1 + rx;
        ho[k] = -1;
        hv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (item < ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w && !RABL(1)) {
          const int sby = gy / WR_BH, sbx = gx / WR_BW;
          const int ly = gy - sby * WR_BH, lx = gx - sbx * WR_BW;
          // which of the owner's published rows / columns holds the pixel
          const int slot = ry == 0 ? WR_BW + lx : (ry == WR_BH + 1 ? lx : (rx == 0 ? 2 * WR_BW + WR_BH + ly : 2 * WR_BW + ly));
          const unsigned off = ((unsigned)(((L & 1) * nwg + sby * a.nbx + sbx) * WR_SLOTS + slot) * WR_NC + 4u * c4) * 4u;
          hv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxb, (int)off, 0, WR_SC1));
          ho[k] = (4 * c4) * WR_CS + ry * WR_RS + rx;
        }
      }
#pragma unroll
      for (int k = 0; k < PER_T; ++k)
        if (ho[k] >= 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[ho[k] + e * WR_CS] = hv[k][e];
        }
    }
    RSTAMP(6);
    __syncthreads();
    RSTAMP(7);
  }
}

static int wres_capacity() {
  static int cap = -1;
  if (cap >= 0) return cap;
  const void* fn = reinterpret_cast<const void*>(conv3x3_wino_resident_kernel);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    return cap = 0;
  }
  int per_cu = 0, dev = 0, ncu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, WR_THREADS, WR_LDS_BYTES) != hipSuccess ||
      hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return cap = 0;
  }
  return cap = per_cu * ncu;
}

static long long wres_blocks(int h, int w) { return (long long)cdiv(h, WR_BH) * cdiv(w, WR_BW); }

// one workgroup per CU, and a few CUs left free so that a co-running launch (the flow estimator on
// its own stream) cannot keep the last workgroups of this one from ever starting
bool conv3x3_wino_resident_ok(int n, int cout, int h, int w) {
  if (n != 1 || cout != WR_NC || h < 2 || w < 2 || (h & 1) || (w & 1)) return false;
  const int cap = wres_capacity();
  return cap > 0 && wres_blocks(h, w) <= cap - 8;
}

int64_t conv3x3_wino_resident_ws_bytes(int h, int w) {
  const long long nb = wres_blocks(h, w);
  return 2 * nb * WR_SLOTS * WR_NC * 4 + ((nb * 4 + 255) / 256) * 256 + 256;   // exchange + flags + fault counter
}

int conv3x3_wino_resident_launch(const tg_wino_layer* layers, int n_layers, int cout, int h, int w, void* ws,
                                 int32_t* err, unsigned base, int poll_limit, tg_stream_t stream) {
  TG_REQUIRE(layers && ws && err, TG_E_ARG, "conv3x3_wino_resident: null pointer");
  TG_REQUIRE(n_layers >= 1 && n_layers <= WR_MAXL, TG_E_ARG, "conv3x3_wino_resident: %d layers (1..%d)", n_layers, WR_MAXL);
  TG_REQUIRE(conv3x3_wino_resident_ok(1, cout, h, w), TG_E_SHAPE,
             "conv3x3_wino_resident: cout=%d h=%d w=%d not supported (cout 64, even h and w, one 8x24 block per CU)", cout, h, w);
  TG_REQUIRE(((uintptr_t)ws % 256) == 0, TG_E_ARG, "conv3x3_wino_resident: workspace must be 256-byte aligned");
  WResArgs a{};
  a.nlayer = n_layers; a.h = h; a.w = w; a.nbx = cdiv(w, WR_BW); a.nby = cdiv(h, WR_BH);
  const long long nb = (long long)a.nbx * a.nby;
  for (int i = 0; i < n_layers; ++i) {
    const tg_wino_layer& l = layers[i];
    TG_REQUIRE(l.x && l.u_packed && l.y && l.bias, TG_E_ARG, "conv3x3_wino_resident: layer %d: null pointer", i);
    TG_REQUIRE(l.act == TG_ACT_NONE || l.act == TG_ACT_RELU || l.act == TG_ACT_LRELU02, TG_E_ARG,
               "conv3x3_wino_resident: layer %d: act=%d", i, l.act);
    if (i == 0) {
      TG_REQUIRE(l.cin > 0 && l.cin <= WR_NC && (!l.x2 || (l.c1 > 0 && l.c1 < l.cin)) && !l.res, TG_E_ARG,
                 "conv3x3_wino_resident: first layer: cin=%d (<= 64) c1=%d, no residual", l.cin, l.c1);
    } else {
      // the chain pattern of the reference's SRNet: a layer reads what the previous one wrote; a residual
      // input is the tensor the previous layer read (= what the destination LDS buffer still holds)
      TG_REQUIRE(l.cin == cout && !l.x2 && l.x == layers[i - 1].y, TG_E_ARG,
                 "conv3x3_wino_resident: layer %d must read layer %d's output", i, i - 1);
      TG_REQUIRE(!l.res || l.res == layers[i - 1].x, TG_E_ARG,
                 "conv3x3_wino_resident: layer %d: the residual input must be layer %d's input", i, i - 1);
    }
    WResLayer& d = a.L[i];
    d.u = l.u_packed; d.bias = l.bias; d.act = l.act; d.res = l.res ? 1 : 0; d.nks = 4 * cdiv(l.cin, 16);
  }
  a.x = layers[0].x; a.x2 = layers[0].x2; a.c1 = layers[0].x2 ? layers[0].c1 : layers[0].cin; a.cin0 = layers[0].cin;
  a.y = layers[n_layers - 1].y;
  a.xbuf = static_cast<float*>(ws);
  a.flags = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + 2 * nb * WR_SLOTS * WR_NC * 4);
  a.err = err; a.base = base; a.poll_limit = poll_limit;
#if TG_WRES_LAB
  { static const int abl = [] { const char* e = getenv("TG_WRES_ABL"); return e ? atoi(e) : 0; }(); a.abl = abl; }
#endif
  hipLaunchKernelGGL(conv3x3_wino_resident_kernel, dim3((unsigned)nb), dim3(WR_THREADS), WR_LDS_BYTES,
                     (hipStream_t)stream, a);
  return check_launch("conv3x3_wino_resident");
}

}  // namespace tg

using namespace tg;

#if TG_WRES_LAB
extern "C" int tg_lab_wres_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wres_dbg), sizeof(long long) * 12 * 24 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int tg_conv3x3_wino_resident_supported(int n, int cout, int h, int w) {
  return conv3x3_wino_resident_ok(n, cout, h, w) ? 1 : 0;
}

extern "C" int64_t tg_conv3x3_wino_resident_ws_bytes(int h, int w) {
  if (h <= 0 || w <= 0) return -1;
  return conv3x3_wino_resident_ws_bytes(h, w);
}

extern "C" int tg_conv3x3_wino_resident(const tg_wino_layer* layers, int n_layers, int cout, int h, int w,
                                        void* workspace, int epoch, tg_stream_t stream) {
  TG_REQUIRE(workspace && epoch > 0, TG_E_ARG, "conv3x3_wino_resident: bad argument (epoch counts from 1)");
  const int64_t bytes = conv3x3_wino_resident_ws_bytes(h, w);
  int32_t* err = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) + bytes - 256);
  return conv3x3_wino_resident_launch(layers, n_layers, cout, h, w, workspace, err, (unsigned)epoch * 32u,
                                      TG_CHAIN_POLL_LIMIT_DEFAULT, stream);
}

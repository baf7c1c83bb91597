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
//
// TG_FILE_FLAGS: -fno-slp-vectorize
// (build.sh passes this line's flags to hipcc for THIS file.  With the K loop one straight-line block the SLP
// vectorizer pairs the transform's adds into v_pk_add_f32 and pays for it with ~12 v_mov_b32 per K step -- every VALU
// instruction adds to the fp32-MFMA time on this pipe: 21.2 vs 18.9 us per layer, round 5.)
#include <type_traits>

#include "tg_common.h"

// ---- lab switches (A/B and ablation builds, tools/build_lab_libs.sh).  Every one of them needs -DTG_LAB=1, which
// csrc/build.sh refuses for the in-tree library: the shipped translation unit is the DEFAULT column below and nothing
// else (WR_FAKEBANK computes wrong results by design, timing only). ----
#ifndef WR_POLL_SLEEP
#define WR_POLL_SLEEP 2   // s_sleep units (64 cycles) between two polls of the ring
#endif
#ifndef WR_PRIO_I
#define WR_PRIO_I 0     // s_setprio of the interior waves
#endif
#ifndef WR_PRIO_B
#define WR_PRIO_B 0     // s_setprio of the boundary waves (the interior waves stay at 0)
#endif
#ifndef WR_USETS
#define WR_USETS 2      // weight blocks (K steps) in flight per wave: 2 or 3
#endif
#ifndef TG_WRES_LAB
#define TG_WRES_LAB 0   // 1: ablation switches (env TG_WRES_ABL) compiled in -- tools/build_lab_libs.sh, timing only
#endif
#define RABL(bit) (TG_WRES_LAB && (a.abl & (bit)))
#ifndef WR_FAKEBANK
#define WR_FAKEBANK 0   // lab, TIMING ONLY (wrong results): window origins spread over distinct bank pairs -- the ceiling of any layout fix
#endif
#ifndef WR_RDFORM
#define WR_RDFORM 0     // 0: the compiler's window reads (it merges them into ds_read2_b64); 1: eight ds_read_b64 (inline asm)
#endif
#ifndef WR_PREF
#define WR_PREF 1       // 1: K step ks + 1's window is requested before K step ks's MFMAs (second register set)
#endif
#ifndef WR_PK
#define WR_PK 0         // 1: the input transform on v_pk_add_f32 (measured)
#endif
#ifndef WR_UASM
#define WR_UASM 1       // hand-written weight requests + exact vmcnt waits (round 5); 0: compiler-visible loads, for A/B
#endif
#ifndef WR_BRANCHY_U
#define WR_BRANCHY_U 0
#endif
#define WR_LAB_BITS ((TG_WRES_LAB ? 1 : 0) | (WR_FAKEBANK ? 2 : 0) | (WR_RDFORM ? 4 : 0) | (WR_PREF != 1 ? 8 : 0) | \
                     (WR_PK ? 16 : 0) | (WR_UASM != 1 ? 32 : 0) | (WR_BRANCHY_U ? 64 : 0) | (WR_USETS != 2 ? 128 : 0) | \
                     (WR_PRIO_I || WR_PRIO_B ? 256 : 0) | (WR_POLL_SLEEP != 2 ? 512 : 0))
#if !TG_LAB && WR_LAB_BITS
#error "tg_conv3x3_wino_res.hip: a lab switch is set without -DTG_LAB=1 (the in-tree library ships the defaults only)"
#endif
#if WR_UASM && WR_PK
#error "WR_PK's K step never waits for the hand-requested weight block (wait_u): -DWR_PK=1 needs -DWR_UASM=0"
#endif
#if WR_UASM && WR_USETS != 2
#error "the hand-written waits assume two weight blocks in flight"
#endif

namespace tg {

int wres_lab_bits() { return WR_LAB_BITS | (TG_LAB ? 1024 : 0); }    // tg_build_info(): 0 in the shipped library

typedef float v2f __attribute__((ext_vector_type(2)));

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
constexpr int WR_ACT_FLOATS = 2 * WR_NC * WR_CS;   // the two activation buffers: 147 456 bytes
constexpr int WR_BIAS_OFF = WR_ACT_FLOATS;          // [layer][64] biases (read once per launch)
constexpr int WR_SYNC_OFF = WR_BIAS_OFF + WR_MAXL * WR_NC;   // 16 bytes: arrival counter of the boundary waves
constexpr size_t WR_LDS_BYTES = (size_t)(WR_SYNC_OFF + 4) * sizeof(float);   // 153 616 of 163 840

// Tile (ty * 12 + tx) of lane n = 0..15 of group g.  Group 2 holds INTERIOR tiles only (rows 1-2, columns
// 1..10: their 4x4 windows never touch the ring), so its four waves start the next layer right behind the
// workgroup barrier while the waves of groups 0 and 1 wait for the neighbours' rings.  Chosen (brute force)
// so that the 32 lanes of a ds_read_b64 group hit distinct bank pairs but for four two-way collisions.
__constant__ unsigned char WR_TILE[48] = {
    0, 1, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 24, 39,
    2, 8, 16, 23, 35, 36, 37, 38, 40, 41, 42, 43, 44, 45, 46, 47,
    17, 18, 19, 20, 21, 22, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34};

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
  float* xbuf;         // exchange: [parity 2][block][slot 64][channel pair 32]{value, tag, value, tag}
  int* err;            // fault counter (pinned host memory or device memory)
  unsigned base;       // tag of layer l of this launch = base + l + 1 (strictly increasing over launches)
  int poll_limit;
  int nlayer, h, w, nbx, nby, c1, cin0;
  // optional tail: ConvTranspose2d(64, 64, 3, 2, 1, 1) + act of the last layer's output (SRNet's first up-sampling
  // layer, tecogan_nets.py:119-126) on the resident blocks; ct_u == null: the last layer's output goes to y
  const float* ct_u;   // tg_conv3x3_wino_resident_ct_pack form
  const float* ct_bias;
  float* ct_y;         // (64, 2h, 2w)
  int ct_act;
  int abl;             // lab builds only: 1 no flag wait / ring loads, 2 no MFMA, 4 no weight loads, 8 no ring stores, 16 no window reads, 32 no hand-over at all, 64 no input transform, 128 weights from L1, 256 half the weight bytes, 512 half the window bytes
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
    for (int i = t; i < WR_ACT_FLOATS / 4; i += WR_THREADS) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < a.nlayer * WR_NC; i += WR_THREADS) s_act[WR_BIAS_OFF + i] = a.L[i >> 6].bias[i & 63];
    if (t == 0) *reinterpret_cast<unsigned*>(s_act + WR_SYNC_OFF) = 0u;
  }
  unsigned* const s_cnt = reinterpret_cast<unsigned*>(s_act + WR_SYNC_OFF);
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
  const int T = WR_TILE[16 * g + (l & 15)];   // tile of the block (row-major 4 x 12)
  const int ty = T / WR_TW, tx = T - ty * WR_TW;
  const bool interior = g == 2;               // wave-uniform
  if (WR_PRIO_B > 0 && !interior) __builtin_amdgcn_s_setprio(WR_PRIO_B);
  if (WR_PRIO_I > 0 && interior) __builtin_amdgcn_s_setprio(WR_PRIO_I);
  const int kk = l >> 4;                      // K index inside a K step (B operand) / output-channel quad (D)
#if WR_FAKEBANK
  const int rb = kk * WR_CS + 2 * (l & 15);
#else
  const int rb = kk * WR_CS + (2 * ty) * WR_RS + 2 * tx;        // window origin in the resident block (floats)
#endif
  const int oc_base = 16 * q + 4 * kk;
  const int wb = oc_base * WR_CS + (2 * ty + 1) * WR_RS + 2 * tx + 1;   // own 2x2 pixels, channel oc_base
  const int gy0 = Y0 + 2 * ty, gx0 = X0 + 2 * tx;               // image position of the tile
  const bool live = gy0 < a.h && gx0 < a.w;                     // h, w even: a tile is inside or outside as a whole
  const bool e_top = ty == 0, e_bot = ty == WR_TH - 1, e_lft = tx == 0, e_rgt = tx == WR_TW - 1;
  const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc(
      a.xbuf, 0, (unsigned)(2u * nwg * WR_SLOTS * WR_NC * 8u), 0x00020000);

  const size_t ulane = (size_t)(q * 4) * 64 + l;               // this lane's 16 bytes inside a K step's block
  constexpr size_t USTEP = (size_t)4 * 4 * 64;                  // f32x4 per K step (64 output channels)
  // The weight prefetch (round 5).  Two things kept a K step from ever running ahead of its weights:
  //  * a prefetch behind `if (ks < nks)` makes the compiler's s_waitcnt insertion merge the two paths at the join and
  //    assume the FEWER loads in flight: every K step waited `vmcnt(3..0)`, i.e. for the block requested a moment ago
  //    as well as for its own;
  //  * branch-free (past the last K step the last block is requested again: an L1 hit, never used) the same pass
  //    still emitted `vmcnt(0)` in this loop nest.
  // So the loads and their waits are written out (WR_UASM): the compiler does not see these loads, the waits carry the
  // exact counts -- a block's four loads are always followed by the four loads of the other block before it is used,
  // so `vmcnt(7..4)` releases it (at a layer's first K steps more has been issued in between: the wait is then merely
  // stronger than needed).  The compiler's own waits (ring, stores) count only its own operations and can only be
  // stronger than needed as well.  The stream of blocks runs ACROSS the layers: the last two K steps of a layer
  // request the first two blocks of the next one (they arrive under the epilogue and the hand-over), so exactly two
  // blocks are in flight at every point of the launch and the epilogue issues no request of its own.
  // (WR_UASM = 0: compiler-visible loads, the round-4 form with -DWR_BRANCHY_U=1; lab builds only.  Because the compiler
  // cannot see these loads, csrc/build.sh FAILS the build if this kernel ever reports scratch or spilled registers.)
  const f32x4* un = nullptr;                   // WR_UASM: the next layer's blocks (the K loop runs on into them)
  auto load_u = [&](const f32x4* ub, int ks, int nks, f32x4 (&u)[4]) {
#if WR_BRANCHY_U
    if (ks >= nks || (RABL(4) && ks > 1)) return;
#elif WR_UASM
    if (RABL(4) && ks > 1) return;
    if (ks >= nks) { ub = un; ks -= nks; }      // uniform: a scalar select, no branch
#else
    if (RABL(4) && ks > 1) return;
    ks = ks < nks ? ks : nks - 1;
#endif
    const f32x4* p = ub + (size_t)(RABL(128) ? (ks & 1) : ks) * USTEP;     // lab 128: always the same two blocks (L1 hits)
#if WR_UASM
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(u[0]) : "v"(p));
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(u[1]) : "v"(p));
    asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(u[2]) : "v"(p));
    asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(u[3]) : "v"(p));
#else
    u[0] = p[0]; u[1] = p[64];
    if (RABL(256)) return;                     // lab 256: half the weight bytes
    u[2] = p[128]; u[3] = p[192];
#endif
  };
  // Behind the K loop two blocks nobody uses are still in flight (the clamped requests of the last two K steps): the
  // compiler takes their registers for dead and would hand them to the next instruction -- a landing load then
  // overwrites, say, the address of the epilogue's first request (a memory fault, found on the GPU).  The drain keeps
  // the eight registers allocated until every load has landed.
  auto drain_u = [&](f32x4 (&ua)[4], f32x4 (&ub_)[4]) {
#if WR_UASM
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ua[0]), "+v"(ua[1]), "+v"(ua[2]), "+v"(ua[3]),
                                        "+v"(ub_[0]), "+v"(ub_[1]), "+v"(ub_[2]), "+v"(ub_[3]));
#endif
  };
  // the block `u` was requested one block ago (see above): release it quarter by quarter
  auto wait_u = [&](f32x4 (&u)[4]) {
#if WR_UASM
    asm volatile("s_waitcnt vmcnt(7)" : "+v"(u[0]));
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(u[1]));
    asm volatile("s_waitcnt vmcnt(5)" : "+v"(u[2]));
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(u[3]));
#endif
  };

  f32x4 u0[4], u1[4];
#if WR_USETS == 3
  f32x4 u2[4];
#endif
  {
    const f32x4* ub = reinterpret_cast<const f32x4*>(a.L[0].u) + ulane;
    load_u(ub, 0, a.L[0].nks, u0);
    load_u(ub, 1, a.L[0].nks, u1);
#if WR_USETS == 3
    load_u(ub, 2, a.L[0].nks, u2);
#endif
  }

  const bool fold = a.ct_u != nullptr;        // launch-uniform
  for (int L = 0; L < a.nlayer; ++L) {
    const WResLayer& lay = a.L[L];
    const float* src = s_act + (L & 1) * (WR_NC * WR_CS);
    float* dst = s_act + ((L & 1) ^ 1) * (WR_NC * WR_CS);
    const f32x4* ub = reinterpret_cast<const f32x4*>(lay.u) + ulane;
    const int nks = lay.nks;
    // (behind the last layer its own first blocks are requested once more; drain_u retires them)
    un = reinterpret_cast<const f32x4*>(a.L[L + 1 < a.nlayer ? L + 1 : L].u) + ulane;

    RSTAMP(0);
    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one K step: window of channel 4 ks + kk -> B^T d B in registers -> 16 MFMAs.  (fp32 MFMAs and VALU
    // instructions of the waves of a SIMD do NOT overlap on gfx950 -- tools/valu_lab.hip: 16 MFMAs + 32
    // adds take exactly the sum of both -- so the order inside a step matters little; a burst form with
    // the next window prefetched measured slower: 24.4 vs 21.5 us per layer, it costs registers.)
    // the 4 x 4 window of channel 4 ks + kk at this lane's tile
    auto win_load = [&](int ks, float (&d)[4][4]) {
      const float* sp = src + rb + (RABL(16) ? 0 : ks * (4 * WR_CS));
#if WR_RDFORM == 1
      // eight ds_read_b64 (2 x 32 lanes, 256 B/clk) instead of the four ds_read2_b64 the compiler merges them into
      // (4 x 16 lanes per access, 128 B/clk -- MI355X_MICROARCH.md section LDS)
      const unsigned ad = (unsigned)(uintptr_t)sp;      // LDS byte address = low 32 bits of the generic pointer
      v2f p[8];
#define WR_DSR(i, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(p[i]) : "v"(ad))
      WR_DSR(0, 0); WR_DSR(1, 8); WR_DSR(2, 112); WR_DSR(3, 120); WR_DSR(4, 224); WR_DSR(5, 232); WR_DSR(6, 336); WR_DSR(7, 344);
#undef WR_DSR
      static_assert(WR_RS == 28, "the byte offsets above are rows of 28 floats");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
#pragma unroll
      for (int r = 0; r < 4; ++r) { d[r][0] = p[2 * r].x; d[r][1] = p[2 * r].y; d[r][2] = p[2 * r + 1].x; d[r][3] = p[2 * r + 1].y; }
#else
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (RABL(512) && r >= 2) {             // lab 512: half the window bytes
          d[r][0] = d[r - 2][1]; d[r][1] = d[r - 2][0]; d[r][2] = d[r - 2][3]; d[r][3] = d[r - 2][2];
          continue;
        }
        const float2 p0 = *reinterpret_cast<const float2*>(sp + r * WR_RS);
        const float2 p1 = *reinterpret_cast<const float2*>(sp + r * WR_RS + 2);
        d[r][0] = p0.x; d[r][1] = p0.y; d[r][2] = p1.x; d[r][3] = p1.y;
      }
#endif
    };
    // one K step: window -> B^T d B in registers -> 16 MFMAs.  (fp32 MFMAs and VALU
    // instructions of the waves of a SIMD do NOT overlap on gfx950 -- tools/valu_lab.hip: 16 MFMAs + 32
    // adds take exactly the sum of both -- so the order inside a step matters little; a burst form with
    // the next window prefetched measured slower: 24.4 vs 21.5 us per layer, it costs registers.)
    auto kcompute = [&](const float (&d)[4][4], f32x4 (&u)[4]) {
      wait_u(u);
      f32x4 bq[4];
      if (RABL(64)) {                          // lab: no input transform (wrong results; what the 32 adds cost)
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[r] = f32x4{d[r][0], d[r][1], d[r][2], d[r][3]};
      } else {
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
#if WR_PK
    auto kstep = [&](int ks, const f32x4 (&u)[4]) {
      const float* sp = src + rb + (RABL(16) ? 0 : ks * (4 * WR_CS));
      // B^T d B on packed fp32 (v_pk_add_f32: the two columns of an 8-byte LDS read are one operand): 8 packed
      // row combinations, then per combination {q0 - q2, q1 - q3} (packed) and {q1 + q2, q2 - q1} -- ONE
      // v_pk_add_f32 with op_sel / neg_hi picking the halves (the compiler needs three instructions for that
      // shuffle, hence the asm).  16 VALU instructions per window instead of 32, the same fp32 sums bit for bit
      // (q2 - q1 is evaluated as -q1 + q2).  The s_nop closes the VALU-write -> MFMA-read distance (2 wait
      // states on gfx950) that the compiler cannot see through the asm.
      v2f lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lo[r] = *reinterpret_cast<const v2f*>(sp + r * WR_RS);
        hi[r] = *reinterpret_cast<const v2f*>(sp + r * WR_RS + 2);
      }
      f32x4 bq[4];
      auto cols = [&](const v2f ql0, const v2f qh0, const v2f ql1, const v2f qh1, f32x4& b0, f32x4& b1) {
        const v2f e0 = ql0 - qh0, e1 = ql1 - qh1;
        v2f f0, f1;
        asm("v_pk_add_f32 %0, %2, %3 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
            "v_pk_add_f32 %1, %4, %5 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
            "s_nop 1"
            : "=&v"(f0), "=&v"(f1) : "v"(ql0), "v"(qh0), "v"(ql1), "v"(qh1));
        b0 = f32x4{e0.x, f0.x, f0.y, e0.y};
        b1 = f32x4{e1.x, f1.x, f1.y, e1.y};
      };
      cols(lo[0] - lo[2], hi[0] - hi[2], lo[1] + lo[2], hi[1] + hi[2], bq[0], bq[1]);
      cols(lo[2] - lo[1], hi[2] - hi[1], lo[1] - lo[3], hi[1] - hi[3], bq[2], bq[3]);
      if (RABL(2)) { acc[0] += bq[0] + bq[1] + bq[2] + bq[3] + u[0] + u[1] + u[2] + u[3]; return; }
#pragma unroll
      for (int p = 0; p < 16; ++p)
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[p >> 2][p & 3], bq[p >> 2][p & 3], acc[p], 0, 0, 0);
    };
#else
    auto kstep = [&](int ks, f32x4 (&u)[4]) {
      float d[4][4];
      win_load(ks, d);
      kcompute(d, u);
    };
#endif

#if WR_PREF
    {   // window of K step ks + 1 in flight under the MFMAs of K step ks (the LDS latency of a K step is otherwise
        // exposed at its head: a wave that is alone on its SIMD -- the interior waves during the hand-over -- idles there)
      float dA[4][4], dB[4][4];
      win_load(0, dA);
      for (int ks = 0; ks < nks; ks += 2) {
        win_load(ks + 1, dB);
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the reads to the end of the K step)
        kcompute(dA, u0);
        __builtin_amdgcn_sched_barrier(0);
        load_u(ub, ks + 2, nks, u0);
        win_load(ks + 2 < nks ? ks + 2 : ks, dA);
        __builtin_amdgcn_sched_barrier(0);
        kcompute(dB, u1);
        __builtin_amdgcn_sched_barrier(0);
        load_u(ub, ks + 3, nks, u1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#elif WR_USETS == 3
    {   // three weight blocks in flight (two K steps of distance): a wave that runs alone on its SIMD -- the
        // interior waves during the hand-over -- otherwise waits for L2 every K step
      int ks = 0;
      for (; ks + 3 <= nks; ks += 3) {
        kstep(ks, u0);
        __builtin_amdgcn_sched_barrier(0);
        load_u(ub, ks + 3, nks, u0);
        kstep(ks + 1, u1);
        __builtin_amdgcn_sched_barrier(0);
        load_u(ub, ks + 4, nks, u1);
        kstep(ks + 2, u2);
        __builtin_amdgcn_sched_barrier(0);
        load_u(ub, ks + 5, nks, u2);
      }
      if (ks < nks) kstep(ks, u0);
      if (ks + 1 < nks) kstep(ks + 1, u1);
    }
#else
    for (int ks = 0; ks < nks; ks += 2) {
      kstep(ks, u0);
      __builtin_amdgcn_sched_barrier(0);
      load_u(ub, ks + 2, nks, u0);
      __builtin_amdgcn_sched_barrier(0);        // (straight-line code now: without it the scheduler sinks the loads into the next K step)
      kstep(ks + 1, u1);
      __builtin_amdgcn_sched_barrier(0);
      load_u(ub, ks + 3, nks, u1);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    RSTAMP(1);
    // ---- inverse transform A^T m A, bias / activation / residual -----------------------------
    const bool last = L + 1 == a.nlayer && !fold;   // with the transposed-conv tail the last layer hands over like any other
    const float slope = act_slope(lay.act);
    float bz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bz[r] = s_act[WR_BIAS_OFF + L * WR_NC + oc_base + r];
    float v[4][2][2];                         // [channel r][row i][column j]
    // ACTK (round 6): 0 = no activation (conv2 of a residual block: nothing to do), 2 = the generic form, 2 VALU per
    // output: max(x, slope * x) = x >= 0 ? x : x * slope for every slope in [0, 1] (ReLU 0, LeakyReLU 0.2, none 1) -- the
    // round-4 form max(x, 0) + slope * min(x, 0) was 3; on this SIMD VALU time adds to the matrix time.  (Further
    // instantiations -- ReLU as one v_max, nothing at all for act none -- spill: 165 of 168 registers are taken, and this
    // kernel must not spill: csrc/build.sh.)
    auto epilogue = [&](auto has_res, auto act_k) {
      constexpr bool RES = decltype(has_res)::value;
      constexpr int ACTK = decltype(act_k)::value;
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
          // x >= 0 ? x : x * slope without a conditional move (v_cndmask_b32: ~23 cycles per wave instruction
          // on gfx950, tools/valu_lab.hip): max(x, 0) + slope * min(x, 0) -- same value for every finite x
          if constexpr (ACTK == 2) {
            v0 = __builtin_fmaxf(v0, slope * v0);
            v1 = __builtin_fmaxf(v1, slope * v1);
          }
          if constexpr (RES) {                // the residual input: what the destination buffer still holds
            v0 += dst[wb + r * WR_CS + i * WR_RS];
            v1 += dst[wb + r * WR_CS + i * WR_RS + 1];
          }
          v[r][i][0] = v0; v[r][i][1] = v1;
        }
      }
    };
    {
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
      (void)sizeof(I1);
      (void)sizeof(I0);
      if (lay.res) epilogue(std::true_type{}, I2{}); else epilogue(std::false_type{}, I2{});      // layer-uniform
    }
    // the next layer's first weights travel under the hand-over
#if !WR_UASM
    {   // (branch-free for the same reason as load_u: behind the last layer the last layer's blocks are requested again)
      const int Ln = L + 1 < a.nlayer ? L + 1 : L;
      const f32x4* un2 = reinterpret_cast<const f32x4*>(a.L[Ln].u) + ulane;
      load_u(un2, 0, a.L[Ln].nks, u0);
      load_u(un2, 1, a.L[Ln].nks, u1);
#if WR_USETS == 3
      load_u(un2, 2, a.L[Ln].nks, u2);
#endif
    }
#endif
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
        // publish the block's outermost pixels as SELF-VALIDATING granules: {value, tag} pairs, two pairs
        // (channels c, c + 1) per 16-byte agent-scope store; tag = base + layer + 1 is unique per launch and
        // layer, so the reader needs no flag and the writer no acknowledgement (RCCL's LL idea; a 16-byte
        // sc1 store is observed untorn on gfx950).  Slot pixel-major: [slot 64][channel pair 32][4 dwords].
        const unsigned tag = a.base + (unsigned)(L + 1);
        const unsigned pb = ((unsigned)((L & 1) * nwg + wg) * WR_SLOTS * (WR_NC / 2) + (unsigned)(oc_base >> 1)) * 16u;
        auto pub = [&](int slot, int i, int j) {
          if (RABL(8)) return;
          const u32x4 d0 = {__builtin_bit_cast(unsigned, v[0][i][j]), tag, __builtin_bit_cast(unsigned, v[1][i][j]), tag};
          const u32x4 d1 = {__builtin_bit_cast(unsigned, v[2][i][j]), tag, __builtin_bit_cast(unsigned, v[3][i][j]), tag};
          const unsigned o = pb + (unsigned)slot * (WR_NC / 2 * 16u);
          __builtin_amdgcn_raw_buffer_store_b128(d0, rxb, (int)o, 0, WR_SC1);
          __builtin_amdgcn_raw_buffer_store_b128(d1, rxb, (int)(o + 16u), 0, WR_SC1);
        };
        if (e_top) { pub(2 * tx, 0, 0); pub(2 * tx + 1, 0, 1); }
        if (e_bot) { pub(WR_BW + 2 * tx, 1, 0); pub(WR_BW + 2 * tx + 1, 1, 1); }
        if (e_lft) { pub(2 * WR_BW + 2 * ty, 0, 0); pub(2 * WR_BW + 2 * ty + 1, 1, 0); }
        if (e_rgt) { pub(2 * WR_BW + WR_BH + 2 * ty, 0, 1); pub(2 * WR_BW + WR_BH + 2 * ty + 1, 1, 1); }
      }
    }
    RSTAMP(2);
    if (last) { drain_u(u0, u1); break; }
    if (RABL(32)) { __syncthreads(); continue; }

    // ---- hand-over -------------------------------------------------------------------------------
    // Every wave: the workgroup barrier (all of dst's own pixels are written, all reads of src are
    // done).  The four INTERIOR waves go straight on to the next layer: their windows never touch the
    // ring.  The eight BOUNDARY waves fetch the ring -- 68 pixels x 32 channel pairs, one 16-byte
    // agent-scope load per item, re-issued until both tags of the item carry this layer's value --
    // store it into dst and meet at an LDS arrival counter, while the interior waves already keep the
    // matrix pipe busy.
    RSTAMP(3);
    __syncthreads();
    RSTAMP(4);
    if (interior) continue;
    {
      constexpr int NRING = 2 * (WR_BW + 2) + 2 * WR_BH;           // 68
      constexpr int ITEMS = NRING * (WR_NC / 2);                   // 2176
      constexpr int NBT = 512;
      constexpr int PER_T = (ITEMS + NBT - 1) / NBT;               // 5 (4.25 on average)
      const unsigned target = a.base + (unsigned)(L + 1);
      unsigned off[PER_T];
      int ho[PER_T];
      unsigned pend = 0;
      int tt = t;
      asm volatile("" : "+v"(tt));            // keeps this geometry out of the K loop's register budget (it would be hoisted and spilled)
#pragma unroll
      for (int k = 0; k < PER_T; ++k) {
        const int item = tt + k * NBT;
        const int pi = item >> 5, c2 = item & 31;
        int ry, rx;
        if (pi < WR_BW + 2) { ry = 0; rx = pi; }
        else if (pi < 2 * (WR_BW + 2)) { ry = WR_BH + 1; rx = pi - (WR_BW + 2); }
        else if (pi < 2 * (WR_BW + 2) + WR_BH) { ry = pi - 2 * (WR_BW + 2) + 1; rx = 0; }
        else { ry = pi - 2 * (WR_BW + 2) - WR_BH + 1; rx = WR_BW + 1; }
        const int gy = Y0 - 1 + ry, gx = X0 - 1 + rx;
        off[k] = 0u; ho[k] = 0;
        if (item < ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w && !RABL(1)) {
          const int sby = gy / WR_BH, sbx = gx / WR_BW;
          const int ly = gy - sby * WR_BH, lx = gx - sbx * WR_BW;
          // which of the owner's published rows / columns holds the pixel
          const int slot = ry == 0 ? WR_BW + lx : (ry == WR_BH + 1 ? lx : (rx == 0 ? 2 * WR_BW + WR_BH + ly : 2 * WR_BW + ly));
          off[k] = ((unsigned)(((L & 1) * nwg + sby * a.nbx + sbx) * WR_SLOTS + slot) * (WR_NC / 2) + (unsigned)c2) * 16u;
          ho[k] = (2 * c2) * WR_CS + ry * WR_RS + rx;
          pend |= 1u << k;
        }
      }
      // (plain scalars, initialised: with a vector array left undefined for items that are not pending the
      // compiler stored element 0 of an item into BOTH channels of its pair -- found on the GPU, round 4)
      unsigned q0[PER_T], q1[PER_T], q2[PER_T], q3[PER_T];
#pragma unroll
      for (int k = 0; k < PER_T; ++k) { q0[k] = 0u; q1[k] = 0u; q2[k] = 0u; q3[k] = 0u; }
      // While the neighbours are still computing, a thread polls ONE of its items (512 16-byte loads per
      // round and workgroup instead of 2176: the poll traffic queued in front of the interior waves'
      // weight loads was several times the weight traffic itself); once that one carries the tag the
      // rest is fetched in one go.
      int polls = 0;
      bool fault = a.poll_limit < 0;
      bool probe = true;
      while (pend != 0u && !fault) {
        const unsigned want_mask = probe ? (pend & (0u - pend)) : pend;      // lowest pending item | all of them
#pragma unroll
        for (int k = 0; k < PER_T; ++k)
          if (want_mask & (1u << k)) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rxb, (int)off[k], 0, WR_SC1);
            q0[k] = q[0]; q1[k] = q[1]; q2[k] = q[2]; q3[k] = q[3];
          }
        const unsigned before = pend;
#pragma unroll
        for (int k = 0; k < PER_T; ++k)
          if ((want_mask & (1u << k)) && q1[k] == target && q3[k] == target) {
            dst[ho[k]] = __builtin_bit_cast(float, q0[k]);
            dst[ho[k] + WR_CS] = __builtin_bit_cast(float, q2[k]);
            pend &= ~(1u << k);
          }
        if (probe && pend != before) probe = false;
        else if (pend != 0u) {
          __builtin_amdgcn_s_sleep(WR_POLL_SLEEP);
          fault = ++polls > a.poll_limit;
        }
      }
      if (fault && pend != 0u) __hip_atomic_fetch_add(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    RSTAMP(6);
    // arrival counter of the eight boundary waves (monotonic over the launch): a wave's LDS stores are
    // ordered before its increment, so whoever reads the full count sees every ring pixel
    if (l == 0) __hip_atomic_fetch_add(s_cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    {
      const unsigned want = 8u * (unsigned)(L + 1);
      while ((int)(__hip_atomic_load(s_cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - want) < 0)
        __builtin_amdgcn_s_sleep(1);
    }
    RSTAMP(7);
  }
  if (!fold) return;
  drain_u(u0, u1);                            // (the requests behind the last layer's epilogue)

  // ---- tail: ConvTranspose2d(64, 64, 3, stride 2, pad 1, output_padding 1) + act on the resident block ------------
  //   out[2y + py][2x + px] = bias + sum_ic sum_taps in[y + dy][x + dx] W[ic][oc][ky][kx],
  //   py = 0: (dy 0, ky 1);  py = 1: (dy 0, ky 2), (dy 1, ky 0)   (the same along x; 9 taps over the 4 phases).
  // The wave keeps its (tile group, 16-channel block): MFMA column n = Winograd tile n of the group, four column
  // blocks = the tile's 2 x 2 pixels (a, b), so a lane's 3 x 3 window at its tile origin holds every operand:
  // 9 LDS reads and 36 MFMAs per K step, 16 accumulators (4 pixels x 4 phases).  Weights straight from L2 (two
  // K steps in flight), output as 16-byte stores of 4 consecutive columns.  Direct fp32 products: as a launch of
  // its own this layer ran at 0.51 of peak on 670 workgroups and cost a kernel boundary in the frame's serial chain.
  __syncthreads();                            // the ring of the last layer is complete for EVERY wave
  {
    // (the tail's geometry is derived from an opaque copy of the lane id: computed ahead of the layer loop it cost the
    // loop a register and a scratch slot)
    int l2 = l;
    asm volatile("" : "+v"(l2));
    const int T2 = WR_TILE[16 * g + (l2 & 15)];
    const int ty = T2 / WR_TW, tx = T2 - ty * WR_TW, kk = l2 >> 4;
    const int oc_base = 16 * q + 4 * kk;
    const int gy0 = Y0 + 2 * ty, gx0 = X0 + 2 * tx;
    const bool live = gy0 < a.h && gx0 < a.w;
    const float* src = s_act + (a.nlayer & 1) * (WR_NC * WR_CS);
    const int cb = kk * WR_CS + (2 * ty + 1) * WR_RS + (2 * tx + 1);
    f32x4 ca[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) ca[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* cu = reinterpret_cast<const f32x4*>(a.ct_u) + (size_t)(q * 3) * 64 + l2;
    constexpr size_t CSTEP = (size_t)4 * 3 * 64;
    constexpr int CT_NKS = WR_NC / 4;
    // (weights: the same hand-written request / wait pairs as the layers' -- see load_u; three 16-byte loads per block)
    auto load_w = [&](int ks, f32x4 (&w)[3]) {
#if WR_UASM
      const f32x4* pw = cu + (size_t)(ks < CT_NKS ? ks : CT_NKS - 1) * CSTEP;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[0]) : "v"(pw));
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(w[1]) : "v"(pw));
      asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(w[2]) : "v"(pw));
#else
      if (ks >= CT_NKS) return;
      const f32x4* pw = cu + (size_t)ks * CSTEP;
      w[0] = pw[0]; w[1] = pw[64]; w[2] = pw[128];
#endif
    };
    auto cstep = [&](int ks, f32x4 (&w)[3]) {
#if WR_UASM
      asm volatile("s_waitcnt vmcnt(5)" : "+v"(w[0]));
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(w[1]));
      asm volatile("s_waitcnt vmcnt(3)" : "+v"(w[2]));
#endif
      const float* sp = src + cb + ks * (4 * WR_CS);
      float d[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) d[i][j] = sp[i * WR_RS + j];
#pragma unroll
      for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int pb2 = 0; pb2 < 2; ++pb2)
#pragma unroll
          for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
              for (int ey = 0; ey <= py; ++ey)
#pragma unroll
                for (int ex = 0; ex <= px; ++ex) {
                  // phase coordinate 0: the single tap (d 0, k 1); coordinate 1: (d 0, k 2), (d 1, k 0)
                  const int dy = py ? ey : 0, ky = py ? (ey ? 0 : 2) : 1;
                  const int dx = px ? ex : 0, kx = px ? (ex ? 0 : 2) : 1;
                  const int tap = ky * 3 + kx, ai = (pa * 2 + pb2) * 4 + py * 2 + px;
                  ca[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap >> 2][tap & 3], d[pa + dy][pb2 + dx], ca[ai], 0, 0, 0);
                }
    };
    f32x4 w0[3], w1[3];
    load_w(0, w0);
    load_w(1, w1);
    for (int ks = 0; ks < CT_NKS; ks += 2) {
      cstep(ks, w0);
      __builtin_amdgcn_sched_barrier(0);
      load_w(ks + 2, w0);
      __builtin_amdgcn_sched_barrier(0);
      cstep(ks + 1, w1);
      __builtin_amdgcn_sched_barrier(0);
      load_w(ks + 3, w1);
      __builtin_amdgcn_sched_barrier(0);
    }
#if WR_UASM
    // (the last two requests are never used: keep their registers until they have landed -- see drain_u)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]));
#endif
    if (live) {
      const float cslope = act_slope(a.ct_act);
      const size_t ohw = (size_t)4 * hw;
      const int ow = 2 * a.w;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bz = a.ct_bias[oc_base + r];
        float* yo = a.ct_y + (size_t)(oc_base + r) * ohw + (size_t)(2 * gy0) * ow + 2 * gx0;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            float o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {     // columns 4 tx' + e = (pixel b = e >> 1, phase px = e & 1)
              float v = ca[(pa * 2 + (e >> 1)) * 4 + py * 2 + (e & 1)][r] + bz;
              o4[e] = __builtin_fmaxf(v, cslope * v);
            }
            *reinterpret_cast<float4*>(yo + (size_t)(2 * pa + py) * ow) = make_float4(o4[0], o4[1], o4[2], o4[3]);
          }
      }
    }
  }
}

// SRNet's first ConvTranspose2d weights (cin 64, cout 64, 3, 3) -> the tail's A operands:
// [K step 16][channel block 4][j 3][lane 64][e 4]: element p = 4 j + e < 9 of lane (oc = 16 q + (lane & 15),
// ic = 4 ks + (lane >> 4)) is W[ic][oc][p / 3][p % 3]; p >= 9 is padding.
__global__ void wres_ct_pack_kernel(const float* __restrict__ w, float* __restrict__ out) {
  const int total = 16 * 4 * 3 * 64 * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 3, lane = (i >> 2) & 63;
    int t = i >> 8;
    const int j = t % 3; t /= 3;
    const int qq = t & 3, ks = t >> 2;
    const int p = 4 * j + e;
    const int oc = 16 * qq + (lane & 15), ic = 4 * ks + (lane >> 4);
    out[i] = p < 9 ? w[((size_t)ic * WR_NC + oc) * 9 + p] : 0.f;
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
  return 2 * nb * WR_SLOTS * WR_NC * 8 + 256;   // exchange granules ({value, tag} pairs) + fault counter
}

int conv3x3_wino_resident_launch(const tg_wino_layer* layers, int n_layers, int cout, int h, int w, void* ws,
                                 int32_t* err, unsigned base, int poll_limit, tg_stream_t stream,
                                 const tg_wres_convt* ct) {
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
  a.err = err; a.base = base; a.poll_limit = poll_limit;
  if (ct) {
    TG_REQUIRE(ct->u_packed && ct->bias && ct->y, TG_E_ARG, "conv3x3_wino_resident: transposed-conv tail: null pointer");
    TG_REQUIRE(ct->act == TG_ACT_NONE || ct->act == TG_ACT_RELU || ct->act == TG_ACT_LRELU02, TG_E_ARG,
               "conv3x3_wino_resident: transposed-conv tail: act=%d", ct->act);
    TG_REQUIRE(n_layers + 1 < 32 && ((uintptr_t)ct->y % 16) == 0, TG_E_ARG, "conv3x3_wino_resident: transposed-conv tail: layers / alignment");
    a.ct_u = ct->u_packed; a.ct_bias = ct->bias; a.ct_y = ct->y; a.ct_act = ct->act;
  }
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
  return tg_conv3x3_wino_resident_ct(layers, n_layers, cout, h, w, workspace, epoch, nullptr, stream);
}

extern "C" int tg_conv3x3_wino_resident_ct(const tg_wino_layer* layers, int n_layers, int cout, int h, int w,
                                           void* workspace, int epoch, const tg_wres_convt* convt, tg_stream_t stream) {
  TG_REQUIRE(workspace && epoch > 0, TG_E_ARG, "conv3x3_wino_resident: bad argument (epoch counts from 1)");
  const int64_t bytes = conv3x3_wino_resident_ws_bytes(h, w);
  int32_t* err = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) + bytes - 256);
  return conv3x3_wino_resident_launch(layers, n_layers, cout, h, w, workspace, err, (unsigned)epoch * 32u,
                                      TG_CHAIN_POLL_LIMIT_DEFAULT, stream, convt);
}

extern "C" size_t tg_conv3x3_wino_resident_ct_floats(void) { return (size_t)16 * 4 * 3 * 64 * 4; }

extern "C" int tg_conv3x3_wino_resident_ct_pack(const float* w_iohw, float* out, tg_stream_t stream) {
  TG_REQUIRE(w_iohw && out, TG_E_ARG, "conv3x3_wino_resident_ct_pack: null pointer");
  hipLaunchKernelGGL(wres_ct_pack_kernel, dim3(96), dim3(256), 0, (hipStream_t)stream, w_iohw, out);
  return check_launch("conv3x3_wino_resident_ct_pack");
}

// Several DEPENDENT 3x3 / stride 1 / pad 1 layers of the training frames in ONE launch.
//
// The training unroll (tecogan_nets.py:174-225) applies SRNet's conv_in + 2*nb residual-block
// convs (:108-116, :141-143) to 19 tiny frames per step (2 x 32 x 32 or 2 x 64 x 64 LR pixels),
// and the reverse sweep runs the matching 2*nb + 1 data-gradient convs per frame: ~800 launches
// per step of the one-shot kernel (tg_conv3x3_mfma.hip), each 11 us for 2-4 us of MFMA -- launch,
// first-load latency and drain of a 64..256-workgroup grid dominate.  Here ONE persistent
// workgroup per output tile walks all the layers:
//
//   * tile = one image row x 32 pixels x (64 or 32) output channels; 8 waves = (2 or 1) oc halves
//     x (4 or 8) K groups, v_mfma_f32_32x32x2_f32, whole K range in flight (the one-shot scheme);
//     the 32-channel form doubles the workgroups for the smallest frames (2 x 32 x 32: 128
//     workgroups instead of 64 -- half the MFMA time per layer);
//   * layer l+1 of a tile needs rows y-1..y+1 of layer l: the workgroups exchange them through
//     global memory with agent-scope (sc1) stores / loads and one flag per (layer, tile, oc
//     part): a producer stores its flag after its data stores were acknowledged (s_waitcnt 0 +
//     workgroup barrier = release), a consumer polls the <= 9 x parts flags of its 3x3 tile
//     neighbourhood before its first load (acquire).  That neighbourhood also covers the
//     write-after-read hazards of ping-pong / in-place residual buffers (see
//     tg_conv3x3_wino.hip); the training chains write every layer to its own tensor anyway
//     (the activations are kept for the backward pass);
//   * the weights of a layer (packed as tg_conv3x3_pack, ocb = 64) are requested into
//     registers BEFORE the flags are polled: their L2 latency hides under the wait;
//   * K groups meet in LDS (fixed order: deterministic), all 512 threads finish the tile
//     (bias, activation, residual, ReLU mask of the NEXT gradient) with 8- / 16-byte stores.
//
// FORWARD PROGRESS: every workgroup of the grid must be resident at the same time (a workgroup
// waits for flags of its neighbours, which may have any block index).  The launcher therefore
// refuses grids larger than HALF of what the device can hold at once (two chains of two
// processes still fit), and the kernel is fail-safe on top: a poll limit ends a wait, counts a
// fault in *err (system scope; the host mirror points it at pinned host memory) and carries on,
// so the launch always terminates and the owner can fall back to one launch per layer.
#include <type_traits>

#include "tg_common.h"

namespace tg {

constexpr int RC_MAXL = 24;
constexpr int RC_TW = 32, RC_PW = 34, RC_RS = 34;
constexpr int RC_IN_FLOATS = 3 * 2 * RC_RS * 4;     // one 8-channel chunk of the patch: 3 rows x 2 halves x 34 slots x 4
constexpr int RC_ITEMS = 3 * 2 * RC_PW;             // 204 16-byte items per chunk
constexpr unsigned RC_OOB = 0x80000000u;
constexpr int RC_SC1 = 16;                          // cache-policy bit of the buffer instructions on gfx940+: agent scope

struct RCLayer {
  const float *x, *x2, *wpk, *bias, *res, *mask;
  float* y;
  long long x_ns, x2_ns, res_ns, mask_ns, y_ns;
  int c1, cin, cout, act;
};
struct RowChainArgs {
  RCLayer L[RC_MAXL];
  int nlayer, n, h, w, tiles_x, ntile;
  unsigned* flags;      // [nlayer][ntile][parts]
  int* err;
  unsigned epoch;
  int poll_limit;
  long long* dbg;       // lab builds (TG_LAB): 8 s_memtime stamps per (layer) of workgroup 0
};
#if TG_LAB
#define RC_STAMP(k) do { if (a.dbg && blockIdx.x == RC_DBG_BLOCK && tid == 0) a.dbg[l * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define RC_DBG_BLOCK 37
#else
#define RC_STAMP(k) do { } while (0)
#endif

template <bool COH> __device__ __forceinline__ float rc_ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, COH ? RC_SC1 : 0));
}

// KG = K groups per workgroup (8 waves): 4 -> 2 oc halves of 32 (one workgroup per tile),
//                                        8 -> 1 oc half (two workgroups per tile).
template <int KG>
__global__ __launch_bounds__(512, 4) void conv3x3_rowchain_kernel(RowChainArgs a) {
  constexpr int NOH = 8 / KG;            // oc halves per workgroup
  constexpr int PARTS = 2 / NOH;         // workgroups per tile
  constexpr int CPW = 8 / KG;            // channel chunks per wave (<= 8 chunks = 64 input channels)
  constexpr int OUTS = NOH * 32 * RC_TW; // outputs of the workgroup's tile
  constexpr int PER = OUTS / 512;        // per thread: 2 or 4 consecutive pixels of one channel
  constexpr int INFLIGHT = KG == 8 ? 4 : 2;   // patch items (16 bytes x 4 channel planes) in flight per thread (register budget)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                               // [8][RC_IN_FLOATS]
  float* red = smem + 8 * RC_IN_FLOATS;             // [KG][NOH * 32][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % NOH, wk = wave / NOH;
  const int lh = lane >> 5, ll = lane & 31;
  int b = blockIdx.x;
  const int part = b % PARTS; b /= PARTS;
  const int tile = b;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.h;
  const int n = b / a.h;
  const int x0 = tx * RC_TW, y0 = ty;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;
  const int ochalf = part * NOH + wn;               // which 32-channel half of the 64 this wave computes

  // epilogue assignment: PER consecutive pixels of one output channel
  const int e_oc = (tid * PER) / RC_TW;             // 0 .. NOH*32-1 (local)
  const int e_px = (tid * PER) % RC_TW;
  const int g_oc = part * NOH * 32 + e_oc;          // channel in [0, 64)
  const int gx = x0 + e_px;
  const bool vec = (a.w % PER == 0);                // rows start PER-aligned: 8-/16-byte accesses are aligned

  for (int l = 0; l < a.nlayer; ++l) {
    const RCLayer& L = a.L[l];
    const int nchunk = (L.cin + CK - 1) / CK;
    RC_STAMP(0);
    // ---- this wave's weights -> registers (before the wait: independent of the previous layer)
    const int c0 = wk * CPW;
    const f32x4* wlane = reinterpret_cast<const f32x4*>(L.wpk) + (lh * 64 + ochalf * 32 + ll);
    f32x4 aw[CPW][9];
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci) {
      const bool on = c0 + ci < nchunk;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (on) v = wlane[(size_t)(c0 + ci) * (9 * CK * 64 / 4) + tap * (2 * 64)];
        aw[ci][tap] = v;
      }
    }
    // (the bias of this thread's output channel too: a global load in the epilogue would sit on the
    //  layer's critical path)
    const float bb = (L.bias && g_oc < L.cout) ? L.bias[g_oc] : 0.f;
    // ---- wait for the producers of the 3x3 tile neighbourhood of the previous layer
    if (l > 0) {
      if (tid < 9 * PARTS) {
        const int nb_ = tid / PARTS, p = tid % PARTS;
        const int ny = ty - 1 + nb_ / 3, nx = tx - 1 + nb_ % 3;
        if (ny >= 0 && ny < a.h && nx >= 0 && nx < a.tiles_x) {
          const unsigned* f = a.flags + ((size_t)(l - 1) * a.ntile + (size_t)(n * a.h + ny) * a.tiles_x + nx) * PARTS + p;
          int polls = 0;
          bool fault = a.poll_limit < 0;
          while (!fault && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
            __builtin_amdgcn_s_sleep(2);
            fault = ++polls > a.poll_limit;
          }
          if (fault) __hip_atomic_fetch_add(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      __syncthreads();
    }
    RC_STAMP(1);
    // ---- the input patch of every chunk -> LDS (layer 0 reads tensors of earlier launches:
    // ordinary loads; later layers read what other workgroups of THIS launch wrote: sc1)
    {
      const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(L.x + (long long)n * L.x_ns), 0, L.c1 * hw * 4, 0x00020000);
      const bool dual = L.x2 != nullptr;
      const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(dual ? L.x2 + (long long)n * L.x2_ns : L.x), 0,
          dual ? (L.cin - L.c1) * hw * 4 : 0, 0x00020000);
      const int total = nchunk * RC_ITEMS;
      // all the loads of a batch of items are issued before the first LDS store (a loop of
      // load -> store iterations pays the memory latency once per iteration: 2.2 us per layer)
      auto stage = [&](auto coh) {
        constexpr bool COH = decltype(coh)::value;
        constexpr int MAXQ = (8 * RC_ITEMS + 511) / 512;       // 4 items per thread at 64 input channels
#pragma unroll
        for (int k0 = 0; k0 < MAXQ; k0 += INFLIGHT) {
          f32x4 v[INFLIGHT];
#pragma unroll
          for (int k = 0; k < INFLIGHT; ++k) {
            const int q = tid + (k0 + k) * 512;
            const int ch = q / RC_ITEMS, rem = q - ch * RC_ITEMS;
            const int r = rem / (2 * RC_PW), rem2 = rem - r * (2 * RC_PW);
            const int hf = rem2 / RC_PW, col = rem2 - hf * RC_PW;
            const int gy = y0 - 1 + r, px = x0 - 1 + col;
            const bool ok = q < total && gy >= 0 && gy < a.h && px >= 0 && px < a.w;
            const unsigned base = ok ? (unsigned)(((ch * CK + 4 * hf) * hw + gy * a.w + px) * 4) : RC_OOB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned o1 = base + (unsigned)j * plane;
              float t = rc_ld<COH>(rs1, o1);
              if (dual) t += rc_ld<COH>(rs2, o1 - (unsigned)L.c1 * plane);
              v[k][j] = t;
            }
          }
#pragma unroll
          for (int k = 0; k < INFLIGHT; ++k) {
            const int q = tid + (k0 + k) * 512;
            const int ch = q / RC_ITEMS, rem = q - ch * RC_ITEMS;
            const int r = rem / (2 * RC_PW), rem2 = rem - r * (2 * RC_PW);
            const int hf = rem2 / RC_PW, col = rem2 - hf * RC_PW;
            if (q < total)
              *reinterpret_cast<f32x4*>(s_in + ch * RC_IN_FLOATS + ((r * 2 + hf) * RC_RS + col) * 4) = v[k];
          }
        }
      };
      if (l > 0) stage(std::true_type{}); else stage(std::false_type{});
    }
    // the residual / mask values of this thread's outputs: requested now, used in the epilogue
    const long long eoff = (long long)g_oc * hw + (long long)y0 * a.w + gx;
    float rr[PER], mm[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) { rr[e] = 0.f; mm[e] = 1.f; }
    const bool live = g_oc < L.cout && gx < a.w;
    if (live && L.res) {
      const float* rp = L.res + (long long)n * L.res_ns + eoff;
      if (l > 0) {
#pragma unroll
        for (int e = 0; e < PER; ++e)
          if (gx + e < a.w) rr[e] = __hip_atomic_load(rp + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
#pragma unroll
        for (int e = 0; e < PER; ++e)
          if (gx + e < a.w) rr[e] = rp[e];
      }
    }
    if (live && L.mask) {
      const float* mp = L.mask + (long long)n * L.mask_ns + eoff;
#pragma unroll
      for (int e = 0; e < PER; ++e)
        if (gx + e < a.w) mm[e] = mp[e];
    }
    __syncthreads();
    RC_STAMP(2);

    // ---- MFMAs of this wave's K range
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci) {
      const int c = c0 + ci;
      if (c < nchunk) {
        const float* si = s_in + c * RC_IN_FLOATS + (lh * RC_RS + ll) * 4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          const f32x4 bq = *reinterpret_cast<const f32x4*>(si + (ky * 2 * RC_RS + kx) * 4);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ci][tap][kk], bq[kk], acc, 0, 0, 0);
        }
      }
    }
    RC_STAMP(3);
    // ---- K groups meet in LDS: red[wk][oc local][px]
#pragma unroll
    for (int r = 0; r < 16; ++r)
      red[(wk * NOH * 32 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * RC_TW + ll] = acc[r];
    __syncthreads();
    {
      float v[PER];
#pragma unroll
      for (int e = 0; e < PER; ++e) v[e] = 0.f;
#pragma unroll
      for (int g = 0; g < KG; ++g) {        // fixed order
        const float* p = red + (g * NOH * 32 + e_oc) * RC_TW + e_px;
        if constexpr (PER == 4) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(p);
          v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
        } else {
          const float2 q = *reinterpret_cast<const float2*>(p);
          v[0] += q.x; v[1] += q.y;
        }
      }
      if (live) {
        const float slope = act_slope(L.act);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
          float q = v[e] + bb;
          q = (q >= 0.f ? q : q * slope + 0.f) + rr[e];
          v[e] = mm[e] > 0.f ? q : 0.f;
        }
        float* yp = L.y + (long long)n * L.y_ns;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yp, 0, L.cout * hw * 4, 0x00020000);
        const unsigned yo = (unsigned)eoff * 4u;
        if (vec && ((reinterpret_cast<uintptr_t>(yp) & 15) == 0) && (hw % 4 == 0) && (L.y_ns % 4 == 0)) {
          if constexpr (PER == 4) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 d = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                       __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(d, ry, (int)yo, 0, RC_SC1);
          } else {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 d = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1])};
            __builtin_amdgcn_raw_buffer_store_b64(d, ry, (int)yo, 0, RC_SC1);
          }
        } else {
#pragma unroll
          for (int e = 0; e < PER; ++e)
            if (gx + e < a.w)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), ry, (int)(yo + 4u * e), 0, RC_SC1);
        }
      }
    }
    // ---- publish the tile: data acknowledged (this wave), then all waves, then the flag
    RC_STAMP(4);
    __builtin_amdgcn_s_waitcnt(0);
    RC_STAMP(5);
    __syncthreads();                     // also: every read of `red` / `s_in` of this layer is done
    RC_STAMP(6);
    if (tid == 0)
      __hip_atomic_store(a.flags + ((size_t)l * a.ntile + tile) * PARTS + part, a.epoch, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
}


// ---- the 16 x 16 x 4 form: FOUR workgroups per tile (2 pixel halves x 2 channel halves) -----------
// On the smallest frames (2 x 32 x 32: 64 row tiles) even the two-workgroup form keeps only half of
// the SIMDs busy (128 workgroups x 8 waves on 256 CUs: 2.1 us of MFMA per layer of a 5.6 us layer).
// v_mfma_f32_16x16x4_f32 tiles (16 channels x 16 pixels x 4 input channels) make a workgroup of
// 16 pixels x 32 channels possible: 256 workgroups, one per CU, 36 MFMAs of 32 cycles per wave.
//   A (weights): lane holds W[oc = 16 mb + (lane & 15)][ic = 8 chunk + 4 ks + (lane >> 4)][tap]; packed by
//     pack3x3_m16_kernel as [oc half][chunk][tap][lane][mb * 2 + ks]: 9 16-byte loads per wave and layer;
//   B (pixels): lane holds X[ic = 8 chunk + 4 ks + (lane >> 4)][row][px = (lane & 15) + kx]; the patch
//     sits in LDS as [chunk][row 3][k 4][col 18][ks 2], so both k steps of a tap are ONE 8-byte read,
//     conflict-free (16 consecutive columns per k);
//   D: lane holds channels 16 mb + 4 (lane >> 4) + r, r = 0..3, of pixel (lane & 15).
constexpr int R16_PW = 18;                                  // 16 pixels + halo
constexpr int R16_CH_FLOATS = 3 * 4 * R16_PW * 2;           // 432 floats per 8-channel chunk

// OIHW (transposed = 0) or the data gradient (2: channel roles swapped, taps rotated) -> the A layout
__global__ void pack3x3_m16_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout,
                                   int transposed) {
  const int total = 2 * 8 * 9 * 64 * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 3, lane = (i >> 2) & 63;
    int t = i >> 8;
    const int tap = t % 9; t /= 9;
    const int chunk = t & 7, och = t >> 3;
    const int mb = e >> 1, ks = e & 1;
    const int oc = och * 32 + mb * 16 + (lane & 15), ic = chunk * 8 + ks * 4 + (lane >> 4);
    float v = 0.f;
    if (oc < cout && ic < cin)
      v = transposed == 2 ? w[((size_t)ic * cout + oc) * 9 + (8 - tap)] : w[((size_t)oc * cin + ic) * 9 + tap];
    out[i] = v;
  }
}

__global__ __launch_bounds__(512, 4) void conv3x3_rowchain16_kernel(RowChainArgs a) {
  constexpr int PARTS = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                                 // [8][R16_CH_FLOATS]
  float* red = smem + 8 * R16_CH_FLOATS;              // [8 K groups][32 oc][16 px]
  const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
  const int lk = lane >> 4, lp = lane & 15;
  int b = blockIdx.x;
  const int part = b & 3; b >>= 2;
  const int pxh = part & 1, och = part >> 1;
  const int tile = b;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.h;
  const int n = b / a.h;
  const int x0 = tx * RC_TW + pxh * 16, y0 = ty;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;
  const int g_half = tx * 2 + pxh, n_half = a.tiles_x * 2;      // this workgroup's 16-pixel column block
  // epilogue assignment: one output per thread
  const int e_oc = tid >> 4, e_px = tid & 15;
  const int g_oc = och * 32 + e_oc, gx = x0 + e_px;

  for (int l = 0; l < a.nlayer; ++l) {
    const RCLayer& L = a.L[l];
    const int nchunk = (L.cin + CK - 1) / CK;
    RC_STAMP(0);
    // ---- this wave's weights (its 8-channel chunk, 9 taps) and the thread's bias
    f32x4 aw[9];
    {
      const f32x4* wl = reinterpret_cast<const f32x4*>(L.wpk) + ((size_t)(och * 8 + wk) * 9) * 64 + lane;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) aw[tap] = wk < nchunk ? wl[tap * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool live = g_oc < L.cout && gx < a.w;
    const float bb = (L.bias && g_oc < L.cout) ? L.bias[g_oc] : 0.f;
    // ---- wait: rows y-1..y+1 x column blocks g-1..g+1 x both channel halves of the previous layer
    if (l > 0) {
      if (tid < 18) {
        const int p = tid & 1, nb_ = tid >> 1;
        const int ny = ty - 1 + nb_ / 3, ng = g_half - 1 + nb_ % 3;
        if (ny >= 0 && ny < a.h && ng >= 0 && ng < n_half) {
          const int ntile_ = (n * a.h + ny) * a.tiles_x + (ng >> 1);
          const unsigned* f = a.flags + ((size_t)(l - 1) * a.ntile + ntile_) * PARTS + (p * 2 + (ng & 1));
          int polls = 0;
          bool fault = a.poll_limit < 0;
          while (!fault && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
            __builtin_amdgcn_s_sleep(2);
            fault = ++polls > a.poll_limit;
          }
          if (fault) __hip_atomic_fetch_add(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      __syncthreads();
    }
    RC_STAMP(1);
    // ---- the patch: (channel c, row r, column) -> LDS [c >> 3][r][c & 3][col][(c >> 2) & 1]
    {
      const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(L.x + (long long)n * L.x_ns), 0, L.c1 * hw * 4, 0x00020000);
      const bool dual = L.x2 != nullptr;
      const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(dual ? L.x2 + (long long)n * L.x2_ns : L.x), 0,
          dual ? (L.cin - L.c1) * hw * 4 : 0, 0x00020000);
      const int total = nchunk * 8 * 3 * R16_PW;
      auto stage = [&](auto coh) {
        constexpr bool COH = decltype(coh)::value;
        constexpr int MAXQ = (64 * 3 * R16_PW + 511) / 512;     // 7 per thread at 64 input channels
        float v[MAXQ];
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
          const int q = tid + k * 512;
          const int col = q % R16_PW, t2 = q / R16_PW;
          const int r = t2 % 3, c = t2 / 3;
          const int gy = y0 - 1 + r, px = x0 - 1 + col;
          const bool ok = q < total && gy >= 0 && gy < a.h && px >= 0 && px < a.w;
          const unsigned o1 = ok ? (unsigned)((c * hw + gy * a.w + px) * 4) : RC_OOB;
          float t = rc_ld<COH>(rs1, o1);
          if (dual) t += rc_ld<COH>(rs2, o1 - (unsigned)L.c1 * plane);
          v[k] = t;
        }
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
          const int q = tid + k * 512;
          const int col = q % R16_PW, t2 = q / R16_PW;
          const int r = t2 % 3, c = t2 / 3;
          if (q < total) s_in[(c >> 3) * R16_CH_FLOATS + (((r * 4 + (c & 3)) * R16_PW + col) << 1) + ((c >> 2) & 1)] = v[k];
        }
      };
      if (l > 0) stage(std::true_type{}); else stage(std::false_type{});
    }
    const long long eoff = (long long)g_oc * hw + (long long)y0 * a.w + gx;
    float rr = 0.f, mm = 1.f;
    if (live && L.res) {
      const float* rp = L.res + (long long)n * L.res_ns + eoff;
      rr = l > 0 ? __hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *rp;
    }
    if (live && L.mask) mm = L.mask[(long long)n * L.mask_ns + eoff];
    __syncthreads();
    RC_STAMP(2);
    // ---- MFMAs: this wave = input channels [8 wk, 8 wk + 8)
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (wk < nchunk) {
      const float* si = s_in + wk * R16_CH_FLOATS + ((lk * R16_PW + lp) << 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        const float2 bq = *reinterpret_cast<const float2*>(si + ((ky * 4 * R16_PW + kx) << 1));
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[tap][0], bq.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[tap][2], bq.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[tap][1], bq.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[tap][3], bq.y, acc[1], 0, 0, 0);
      }
    }
    RC_STAMP(3);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wk * 32 + mb * 16 + 4 * lk + r) << 4) + lp] = acc[mb][r];
    __syncthreads();
    {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += red[((g * 32 + e_oc) << 4) + e_px];      // fixed order
      if (live) {
        const float slope = act_slope(L.act);
        float q = v + bb;
        q = (q >= 0.f ? q : q * slope + 0.f) + rr;
        q = mm > 0.f ? q : 0.f;
        float* yp = L.y + (long long)n * L.y_ns;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yp, 0, L.cout * hw * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q), ry, (int)((unsigned)eoff * 4u), 0, RC_SC1);
      }
    }
    RC_STAMP(4);
    __builtin_amdgcn_s_waitcnt(0);
    RC_STAMP(5);
    __syncthreads();
    RC_STAMP(6);
    if (tid == 0)
      __hip_atomic_store(a.flags + ((size_t)l * a.ntile + tile) * PARTS + part, a.epoch, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int KG> static size_t rc_lds_bytes() {
  return (size_t)(8 * RC_IN_FLOATS + KG * (8 / KG) * 32 * RC_TW) * sizeof(float);
}

// workgroups of conv3x3_rowchain_kernel<KG> the device can hold at once (0 on error)
template <int KG> static int rc_capacity() {
  static int cap = -1;
  if (cap >= 0) return cap;
  const void* fn = reinterpret_cast<const void*>(conv3x3_rowchain_kernel<KG>);
  const size_t lds = rc_lds_bytes<KG>();
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return cap = 0; }
  int per_cu = 0, dev = 0, ncu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, lds) != hipSuccess ||
      hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return cap = 0;
  }
  return cap = per_cu * ncu;
}

static size_t rc16_lds_bytes() { return (size_t)(8 * R16_CH_FLOATS + 8 * 32 * 16) * sizeof(float); }
static int rc16_capacity() {
  static int cap = -1;
  if (cap >= 0) return cap;
  const void* fn = reinterpret_cast<const void*>(conv3x3_rowchain16_kernel);
  int per_cu = 0, dev = 0, ncu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, rc16_lds_bytes()) != hipSuccess ||
      hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return cap = 0;
  }
  return cap = per_cu * ncu;
}

}  // namespace tg

using namespace tg;

extern "C" int64_t tg_conv3x3_chain_flag_ints(int n_layers, int n, int h, int w) {
  if (n_layers <= 0 || n_layers > RC_MAXL || n <= 0 || h <= 0 || w <= 0) return -1;
  return (int64_t)n_layers * n * h * cdiv(w, RC_TW) * 4;
}

extern "C" size_t tg_conv3x3_pack16_floats(void) { return (size_t)2 * 8 * 9 * 64 * 4; }

extern "C" int tg_conv3x3_pack16(const float* w, float* w_packed, int cin, int cout, int transposed,
                                 tg_stream_t stream) {
  TG_REQUIRE(w && w_packed, TG_E_ARG, "conv3x3_pack16: null pointer");
  TG_REQUIRE(cin > 0 && cin <= 64 && cout > 0 && cout <= 64 && (transposed == 0 || transposed == 2), TG_E_SHAPE,
             "conv3x3_pack16: cin=%d cout=%d (<= 64) transposed=%d (0 | 2)", cin, cout, transposed);
  hipLaunchKernelGGL(pack3x3_m16_kernel, dim3(144), dim3(256), 0, (hipStream_t)stream, w, w_packed, cin, cout, transposed);
  return check_launch("pack3x3_m16");
}

// Every layer of a chain packed by ONE launch (the training step re-packs 2 x 21 weight tensors after each
// optimiser step: 42 launches of ~3 us otherwise).  Item i: source tensor (O, i_total, 3, 3), of which the
// input-channel slice [i_off, i_off + I) is used; transposed = 0: the layer itself (cin = I, cout = O),
// 2: its data gradient (cin = O, cout = I, taps rotated).  blockIdx.y = item.
struct PackItems { tg_pack_item it[RC_MAXL]; };
__global__ void chain_pack_kernel(PackItems p, int layout) {
  const tg_pack_item& q = p.it[blockIdx.y];
  const int cin = q.cin, cout = q.cout;
  const int total = layout == 16 ? 2 * 8 * 9 * 64 * 4 : cdiv(cin, 8) * 9 * 8 * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int oc, ic, tap;
    if (layout == 16) {                      // (pack3x3_m16_kernel)
      const int e = i & 3, lane = (i >> 2) & 63;
      int t = i >> 8;
      tap = t % 9; t /= 9;
      const int chunk = t & 7, och = t >> 3;
      oc = och * 32 + (e >> 1) * 16 + (lane & 15); ic = chunk * 8 + (e & 1) * 4 + (lane >> 4);
    } else {                                 // (pack3x3_kernel, one 64-channel block)
      const int j = i & 3;
      int t = i >> 2;
      const int o = t & 63; t >>= 6;
      const int half = t & 1; t >>= 1;
      tap = t % 9;
      oc = o; ic = (t / 9) * 8 + 4 * half + j;
    }
    float v = 0.f;
    if (oc < cout && ic < cin)
      v = q.transposed == 2 ? q.w[((size_t)ic * q.i_total + q.i_off + oc) * 9 + (8 - tap)]
                            : q.w[((size_t)oc * q.i_total + q.i_off + ic) * 9 + tap];
    q.out[i] = v;
  }
}

extern "C" size_t tg_conv3x3_chain_packed_floats(int pack_layout, int cin) {
  return pack_layout == 16 ? (size_t)2 * 8 * 9 * 64 * 4 : (size_t)cdiv(cin, 8) * 9 * 8 * 64;
}

extern "C" int tg_conv3x3_chain_pack(const tg_pack_item* items, int n_items, int pack_layout, tg_stream_t stream) {
  TG_REQUIRE(items && n_items >= 1 && n_items <= RC_MAXL, TG_E_ARG, "conv3x3_chain_pack: %d items (1..%d)", n_items, RC_MAXL);
  TG_REQUIRE(pack_layout == 16 || pack_layout == 64, TG_E_ARG, "conv3x3_chain_pack: layout %d (16 | 64)", pack_layout);
  PackItems p{};
  for (int i = 0; i < n_items; ++i) {
    const tg_pack_item& q = items[i];
    TG_REQUIRE(q.w && q.out && q.cin > 0 && q.cin <= 64 && q.cout > 0 && q.cout <= 64 &&
                   (q.transposed == 0 || q.transposed == 2) && q.i_off >= 0 &&
                   q.i_off + (q.transposed == 2 ? q.cout : q.cin) <= q.i_total, TG_E_ARG,
               "conv3x3_chain_pack: item %d cin=%d cout=%d transposed=%d slice %d of %d", i, q.cin, q.cout, q.transposed,
               q.i_off, q.i_total);
    p.it[i] = q;
  }
  hipLaunchKernelGGL(chain_pack_kernel, dim3(144, (unsigned)n_items), dim3(256), 0, (hipStream_t)stream, p, pack_layout);
  return check_launch("chain_pack");
}

// 0: the shape cannot run as a chained launch on this device (too many tiles to be resident at
// once, channels > 64); else the number of workgroups per tile the launcher will use: 4 = the
// 16 x 16 x 4 form (weights packed by tg_conv3x3_pack16), 2 / 1 = the 32 x 32 x 2 forms
// (tg_conv3x3_pack, ocb 64).
extern "C" int tg_conv3x3_chain_supported(int n, int h, int w, int cmax) {
  if (n <= 0 || h <= 0 || w <= 0 || cmax <= 0 || cmax > 64) return 0;
  const long long ntile = (long long)n * h * cdiv(w, RC_TW);
  int ncu = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  // four workgroups per tile while that is at most one workgroup per CU (and resident twice over)
  if (4 * ntile <= ncu && 2 * 4 * ntile <= rc16_capacity()) return 4;
  if (2 * ntile * 2 <= rc_capacity<8>()) return 2;
  if (2 * ntile <= rc_capacity<4>()) return 1;
  return 0;
}

extern "C" int tg_conv3x3_chain(const tg_chain_layer* layers, int n_layers, int n, int h, int w, int pack_layout,
                                int32_t* flags, int32_t* err, uint32_t epoch, int poll_limit, tg_stream_t stream) {
  TG_REQUIRE(layers && flags && err, TG_E_ARG, "conv3x3_chain: null pointer");
  TG_REQUIRE(n_layers >= 1 && n_layers <= RC_MAXL, TG_E_ARG, "conv3x3_chain: %d layers (1..%d)", n_layers, RC_MAXL);
  TG_REQUIRE(n > 0 && h > 0 && w > 0, TG_E_SHAPE, "conv3x3_chain: n=%d h=%d w=%d", n, h, w);
  TG_REQUIRE(epoch != 0, TG_E_ARG, "conv3x3_chain: epoch 0 is the cleared state of the flags");
  RowChainArgs a{};
  a.nlayer = n_layers; a.n = n; a.h = h; a.w = w; a.tiles_x = cdiv(w, RC_TW);
  const long long ntile = (long long)n * h * a.tiles_x;
  a.ntile = (int)ntile;
  a.flags = reinterpret_cast<unsigned*>(flags); a.err = err; a.epoch = epoch; a.poll_limit = poll_limit;
#if TG_LAB
  a.dbg = reinterpret_cast<long long*>(flags + tg_conv3x3_chain_flag_ints(RC_MAXL, n, h, w));   // lab: the tail of an over-sized flag buffer
#else
  a.dbg = nullptr;
#endif
  int cmax = 1;
  for (int i = 0; i < n_layers; ++i) {
    const tg_chain_layer& l = layers[i];
    TG_REQUIRE(l.x && l.w_packed && l.y, TG_E_ARG, "conv3x3_chain: layer %d: null pointer", i);
    TG_REQUIRE(l.cin > 0 && l.cin <= 64 && l.cout > 0 && l.cout <= 64, TG_E_SHAPE,
               "conv3x3_chain: layer %d: cin=%d cout=%d (<= 64)", i, l.cin, l.cout);
    TG_REQUIRE(!l.x2 || (l.c1 > 0 && l.c1 < l.cin), TG_E_ARG, "conv3x3_chain: layer %d: c1=%d of cin=%d", i, l.c1, l.cin);
    TG_REQUIRE(l.act >= TG_ACT_NONE && l.act <= TG_ACT_LRELU02, TG_E_ARG, "conv3x3_chain: layer %d: act=%d", i, l.act);
    TG_REQUIRE((long long)(l.cin + CK) * h * w * 4 < (1ll << 31), TG_E_SHAPE, "conv3x3_chain: layer %d too large", i);
    RCLayer& d = a.L[i];
    d.x = l.x; d.x2 = l.x2; d.wpk = l.w_packed; d.bias = l.bias; d.res = l.res; d.mask = l.relu_mask; d.y = l.y;
    d.x_ns = l.x_nstride; d.x2_ns = l.x2_nstride; d.res_ns = l.res_nstride; d.mask_ns = l.mask_nstride; d.y_ns = l.y_nstride;
    d.c1 = l.x2 ? l.c1 : l.cin; d.cin = l.cin; d.cout = l.cout; d.act = l.act;
    if (l.cout > cmax) cmax = l.cout;
  }
  const int parts = tg_conv3x3_chain_supported(n, h, w, cmax);
  TG_REQUIRE(parts > 0, TG_E_SHAPE,
             "conv3x3_chain: %lld tiles cannot all be resident on this device (see tg_conv3x3_chain_supported)", ntile);
  TG_REQUIRE(pack_layout == (parts == 4 ? 16 : 64), TG_E_ARG,
             "conv3x3_chain: weights packed in layout %d, this shape runs with %d workgroups per tile and needs layout %d "
             "(tg_conv3x3_chain_supported: 4 -> tg_conv3x3_pack16, 1 | 2 -> tg_conv3x3_pack with ocb 64)",
             pack_layout, parts, parts == 4 ? 16 : 64);
  hipStream_t s = (hipStream_t)stream;
  if (parts == 4) {
    hipLaunchKernelGGL(conv3x3_rowchain16_kernel, dim3((unsigned)(ntile * 4)), dim3(512), rc16_lds_bytes(), s, a);
    return check_launch("conv3x3_chain16");
  }
  if (parts == 2)
    hipLaunchKernelGGL(conv3x3_rowchain_kernel<8>, dim3((unsigned)(ntile * 2)), dim3(512), rc_lds_bytes<8>(), s, a);
  else
    hipLaunchKernelGGL(conv3x3_rowchain_kernel<4>, dim3((unsigned)ntile), dim3(512), rc_lds_bytes<4>(), s, a);
  return check_launch("conv3x3_chain");
}

// ---- SRNet's conv_in + residual blocks on one training frame (tecogan_nets.py:108-116, :141-143)
// and the matching reverse sweep, each as ONE chained launch.  `acts` / `dz` hold 1 + 2*nb
// tensors (n, nf, h, w) back to back:
//   acts[0] = relu(conv_in(cat[lr, tran]));  acts[1+2b] = relu(conv1_b(acts[2b]));
//   acts[2+2b] = conv2_b(acts[1+2b]) + acts[2b]                      (block b's output)
//   dz[2nb]   is the INPUT: the caller places the gradient of the body's output there (so that every
//             (dZ, X) operand pair of the weight gradients lies in the two blocks, tg_wgrad3x3_body);
//   dz[1+2b]  = relu'(acts[1+2b]) . dgrad(conv2_b)(g_b)             g_b = dz[2+2b]
//   dz[2b]    = dgrad(conv1_b)(dz[1+2b]) + g_b                        for b >= 1: gradient of block b-1's output
//   dz[0]     = relu'(acts[0]) . (dgrad(conv1_0)(dz[1]) + g_0)        dZ of conv_in
//   d_tran    = dgrad(conv_in, channels [c_lr, c_lr + c_tran))(dz[0])
// so (dz[i], input of layer i) are exactly the operand pairs of the deferred weight gradients.
static int body_common(int nb, int n, int nf, int h, int w, const void* a, const void* b_, const void* c) {
  TG_REQUIRE(a && b_ && c, TG_E_ARG, "srnet_body: null pointer");
  TG_REQUIRE(nb >= 1 && 1 + 2 * nb + 1 <= RC_MAXL, TG_E_ARG, "srnet_body: nb=%d (1..%d)", nb, (RC_MAXL - 2) / 2);
  TG_REQUIRE(n > 0 && nf > 0 && nf <= 64 && h > 0 && w > 0, TG_E_SHAPE, "srnet_body: n=%d nf=%d h=%d w=%d", n, nf, h, w);
  return TG_OK;
}

extern "C" int tg_srnet_body_fwd(const tg_packed_layer* layers, int pack_layout, int nb, const float* lr, int c_lr,
                                 const float* tran, int c_tran, float* acts, int n, int nf, int h, int w,
                                 int32_t* flags, int32_t* err, uint32_t epoch, int poll_limit, tg_stream_t stream) {
  if (int rc = body_common(nb, n, nf, h, w, layers, lr, acts)) return rc;
  TG_REQUIRE(tran && c_lr > 0 && c_tran > 0 && c_lr + c_tran <= 64, TG_E_ARG, "srnet_body_fwd: c_lr=%d c_tran=%d", c_lr, c_tran);
  const int64_t hw = (int64_t)h * w, ns = (int64_t)nf * hw, ts = (int64_t)n * ns;
  tg_chain_layer cl[RC_MAXL];
  for (int i = 0; i < 1 + 2 * nb; ++i) {
    TG_REQUIRE(layers[i].w && layers[i].b, TG_E_ARG, "srnet_body_fwd: layer %d null", i);
    tg_chain_layer& d = cl[i];
    d = tg_chain_layer{};
    const bool first = i == 0, conv2 = !first && (i % 2 == 0);
    d.x = first ? lr : acts + (size_t)(i - 1) * ts;
    d.x2 = first ? tran : nullptr;
    d.w_packed = layers[i].w; d.bias = layers[i].b;
    d.res = conv2 ? acts + (size_t)(i - 2) * ts : nullptr;
    d.y = acts + (size_t)i * ts;
    d.x_nstride = first ? (int64_t)c_lr * hw : ns; d.x2_nstride = first ? (int64_t)c_tran * hw : 0;
    d.res_nstride = ns; d.mask_nstride = 0; d.y_nstride = ns;
    d.c1 = first ? c_lr : nf; d.cin = first ? c_lr + c_tran : nf; d.cout = nf;
    d.act = conv2 ? TG_ACT_NONE : TG_ACT_RELU;
  }
  return tg_conv3x3_chain(cl, 1 + 2 * nb, n, h, w, pack_layout, flags, err, epoch, poll_limit, stream);
}

extern "C" int tg_srnet_body_bwd(const tg_packed_layer* dgrad, int pack_layout, int nb, const float* acts, float* dz,
                                 float* d_tran, int c_tran, int n, int nf, int h, int w, int32_t* flags, int32_t* err,
                                 uint32_t epoch, int poll_limit, tg_stream_t stream) {
  if (int rc = body_common(nb, n, nf, h, w, dgrad, dz, acts)) return rc;
  TG_REQUIRE(d_tran && c_tran > 0 && c_tran <= 64, TG_E_ARG, "srnet_body_bwd: null pointer / c_tran=%d", c_tran);
  const int64_t hw = (int64_t)h * w, ns = (int64_t)nf * hw, ts = (int64_t)n * ns;
  tg_chain_layer cl[RC_MAXL];
  int k = 0;
  for (int b = nb - 1; b >= 0; --b) {
    const float* g = dz + (size_t)(2 + 2 * b) * ts;
    TG_REQUIRE(dgrad[1 + 2 * b].w && dgrad[2 + 2 * b].w, TG_E_ARG, "srnet_body_bwd: block %d null", b);
    {   // dz[1+2b] = relu'(acts[1+2b]) . dgrad(conv2_b)(g)
      tg_chain_layer& d = cl[k++];
      d = tg_chain_layer{};
      d.x = g; d.w_packed = dgrad[2 + 2 * b].w; d.relu_mask = acts + (size_t)(1 + 2 * b) * ts; d.y = dz + (size_t)(1 + 2 * b) * ts;
      d.x_nstride = ns; d.mask_nstride = ns; d.y_nstride = ns; d.c1 = nf; d.cin = nf; d.cout = nf; d.act = TG_ACT_NONE;
    }
    {   // dz[2b] = dgrad(conv1_b)(dz[1+2b]) + g   (b == 0: masked by relu'(acts[0]) -> dZ of conv_in)
      tg_chain_layer& d = cl[k++];
      d = tg_chain_layer{};
      d.x = dz + (size_t)(1 + 2 * b) * ts; d.w_packed = dgrad[1 + 2 * b].w; d.res = g; d.y = dz + (size_t)(2 * b) * ts;
      d.relu_mask = b == 0 ? acts : nullptr;
      d.x_nstride = ns; d.res_nstride = ns; d.mask_nstride = ns; d.y_nstride = ns; d.c1 = nf; d.cin = nf; d.cout = nf;
      d.act = TG_ACT_NONE;
    }
  }
  {     // d_tran = dgrad(conv_in w.r.t. the warped-frame channels)(dz[0])
    TG_REQUIRE(dgrad[0].w, TG_E_ARG, "srnet_body_bwd: conv_in data-gradient pack null");
    tg_chain_layer& d = cl[k++];
    d = tg_chain_layer{};
    d.x = dz; d.w_packed = dgrad[0].w; d.y = d_tran;
    d.x_nstride = ns; d.y_nstride = (int64_t)c_tran * hw; d.c1 = nf; d.cin = nf; d.cout = c_tran; d.act = TG_ACT_NONE;
  }
  return tg_conv3x3_chain(cl, k, n, h, w, pack_layout, flags, err, epoch, poll_limit, stream);
}

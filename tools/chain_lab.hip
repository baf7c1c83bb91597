// Lab: can consecutive "layers" live in one launch on MI355X, synchronised by per-tile flags, with
// agent-scope (sc1) loads / stores instead of L2 invalidation?  Layer l, tile t writes 16 KB =
// f(l, t, epoch) and raises flag[l][t]; layer l+1, tile t waits for tiles t-1, t, t+1 of layer l,
// reads their data back (checks every value) and writes its own.  Tiles are dealt to XCDs round-robin,
// so neighbours live behind different L2s.  Reports mismatches, bail-outs (a poll limit: the kernel
// cannot hang) and the time per layer against the same work as one launch per layer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/chain_lab tools/chain_lab.hip && tools/chain_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int TILE_F = 4096;      // floats per tile (16 KB)
constexpr int AUX_SC1 = 16;       // gfx940+ cache policy: sc1 = agent scope

__device__ __forceinline__ float val(int layer, int tile, int i, int epoch) {
  return (float)((layer * 131 + tile * 7 + i + epoch * 3) & 1023);
}

template <int COHERENT>
__global__ __launch_bounds__(256) void chain_kernel(float* buf0, float* buf1, int* flags, int ntile, int layer0,
                                                    int nlayer, int epoch, int* err) {
  const int b = blockIdx.x;
  const int layer = layer0 + b / ntile, tile = b % ntile;
  const int t = threadIdx.x;
  float* src = (layer & 1) ? buf0 : buf1;      // written by layer - 1
  float* dst = (layer & 1) ? buf1 : buf0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(src, 0, (unsigned)ntile * TILE_F * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (unsigned)ntile * TILE_F * 4u, 0x00020000);
  if (layer > 0) {
    if (COHERENT && b / ntile > 0) {           // same launch: wait for the three producer tiles
      if (t < 3) {
        const int nt = tile - 1 + t;
        if (nt >= 0 && nt < ntile) {
          int polls = 0;
          while (__hip_atomic_load(flags + (layer - 1) * ntile + nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++polls > (1 << 20)) { atomicAdd(err + 1, 1); break; }
          }
        }
      }
      __syncthreads();
    }
    // read back the neighbours' tiles
    int bad = 0;
    for (int k = -1; k <= 1; ++k) {
      const int nt = tile + k;
      if (nt < 0 || nt >= ntile) continue;
      for (int i = t; i < TILE_F; i += 256) {
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (nt * TILE_F + i) * 4, 0, COHERENT ? AUX_SC1 : 0));
        bad += v != val(layer - 1, nt, i, epoch);
      }
    }
    if (bad) atomicAdd(err, bad);
  }
  for (int i = t; i < TILE_F; i += 256)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val(layer, tile, i, epoch)), rd, (tile * TILE_F + i) * 4, 0,
                                          COHERENT ? AUX_SC1 : 0);
  if (COHERENT) {
    __builtin_amdgcn_s_waitcnt(0);             // this wave's stores have left
    __syncthreads();
    if (t == 0) __hip_atomic_store(flags + layer * ntile + tile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int ntile = 670, nlayer = 20, reps = 50;
  float *b0, *b1; int *flags, *err;
  CK(hipMalloc(&b0, (size_t)ntile * TILE_F * 4)); CK(hipMalloc(&b1, (size_t)ntile * TILE_F * 4));
  CK(hipMalloc(&flags, (size_t)nlayer * ntile * 4)); CK(hipMalloc(&err, 8));
  CK(hipMemset(flags, 0, (size_t)nlayer * ntile * 4)); CK(hipMemset(err, 0, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int epoch = 0; float ms; int herr[2];
  // (a) one launch per layer, plain cached accesses (kernel boundaries do the coherence)
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
      ++epoch;
      for (int l = 0; l < nlayer; ++l)
        hipLaunchKernelGGL(chain_kernel<0>, dim3(ntile), dim3(256), 0, 0, b0, b1, flags, ntile, l, nlayer, epoch, err);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
  printf("per-layer launches : %.2f us per layer, mismatches %d\n", 1e3 * ms / (reps * nlayer), herr[0]);
  CK(hipMemset(err, 0, 8));
  // (b) all layers in one launch, flags + sc1 accesses
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
      ++epoch;
      hipLaunchKernelGGL(chain_kernel<1>, dim3(ntile * nlayer), dim3(256), 0, 0, b0, b1, flags, ntile, 0, nlayer, epoch, err);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
  printf("one chained launch : %.2f us per layer, mismatches %d, poll bail-outs %d\n", 1e3 * ms / (reps * nlayer), herr[0], herr[1]);
  return 0;
}

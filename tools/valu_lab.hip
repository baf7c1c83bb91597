// Lab: how do fp32 MFMA, VALU and LDS instructions of the waves of one SIMD overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/valu_lab.hip -o tools/valu_lab && tools/valu_lab
// Every workgroup = W waves per SIMD x 4 SIMDs; 256 workgroups (one per CU).  Per iteration a wave issues
// M fp32 MFMAs (16x16x4, 16 independent accumulators) and V VALU ops (mode: plain add / packed add / fma /
// packed fma / cndmask) and D ds_read_b64.  Reports ns per iteration and the implied cycles at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int M, int V, int MODE, int D>
__global__ void k(float* out, int iters) {
  __shared__ float lds[4096];
  const int t = threadIdx.x;
  lds[t] = (float)t;
  __syncthreads();
  f32x4 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = (float)t * 1e-3f, b = 1.0001f;
  float v[16];
  f32x2 w[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = a + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = f32x2{a + i, a - i};
  float dsum = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < M; ++i)
      acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 15], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(b));
      if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i & 7]) : "v"(w[(i + 1) & 7]));
      if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 15]) : "v"(b));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i & 7]) : "v"(w[(i + 1) & 7]));
      if (MODE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i & 15]) : "v"(b));
      if (MODE == 6) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i & 15]) : "v"(b), "v"(a));
      if (MODE == 7) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[i & 15]) : "v"(b), "v"(a));
      if (MODE == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i & 15]) : "v"(b));
      if (MODE == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i & 15]) : "v"(b));
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      f32x2 r;
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"((t & 63) * 8), "i"(i * 512));
      asm volatile("s_waitcnt lgkmcnt(0)");
      dsum += r[0];
    }
  }
  float s = dsum;
#pragma unroll
  for (int p = 0; p < 16; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += w[i][0] + w[i][1];
  out[blockIdx.x * blockDim.x + t] = s;
}

template <int M, int V, int MODE, int D>
void run(const char* name, float* out) {
  const int iters = 2000;
  for (int wps = 1; wps <= 3; ++wps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<M, V, MODE, D>), dim3(256), dim3(256 * wps), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<M, V, MODE, D>), dim3(256), dim3(256 * wps), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    printf("%-34s waves/SIMD %d: %8.1f ns/iter = %7.0f cycles@2.4GHz  (per wave-iter %6.0f)\n", name, wps, ns, ns * 2.4, ns * 2.4 / wps);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 768 * 4);
  run<0, 32, 0, 0>("32 v_add", out);
  run<0, 32, 4, 0>("32 v_cndmask vcc (e32)", out);
  run<0, 32, 6, 0>("32 v_and_or_b32", out);
  run<0, 32, 7, 0>("32 v_bfi_b32", out);
  run<0, 32, 8, 0>("32 v_mul_f32", out);
  run<0, 32, 9, 0>("32 v_mov_b32", out);
  run<0, 32, 2, 0>("32 v_fma", out);
  return 0;
}

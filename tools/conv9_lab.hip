// Lab prototype: 3x3 conv 64 -> 64 with fp32 operands split EXACTLY into three bf16 terms
// each (x = h + m + l, 8+8+8 significand bits) and the 9 cross products accumulated on the
// bf16 MFMA pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Every bf16 x bf16 product is
// exact in fp32, so the result is an fp32 dot product with a different summation order --
// not a reduced-precision convolution.  MFMA cycles per 16 input channels x 9 taps of one
// 32x32 tile: 81 x 32 = 2592 (bf16 x 9) against 72 x 64 = 4608 (v_mfma_f32_32x32x2f32).
//
// Activations travel between layers as "S3": [n][h][w][chunk 4][split 3][16 ch] bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/conv9_lab tools/conv9_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;

constexpr int C = 64;
constexpr int TR = 2, TC = 32, HR = TR + 2, HC = TC + 2, NPX = HR * HC;   // 136 halo pixels
constexpr int IN_SLOTS = 3 * 2 * NPX;                 // 816 16-byte slots per 16-channel chunk
constexpr int IN_DMA = (IN_SLOTS + 63) / 64;          // 13 wave DMAs
constexpr int IN_BYTES = IN_DMA * 1024;               // 13312
constexpr int W_SLOTS = 3 * 3 * 2 * 64;               // (kx, split, k-group, oc) per (chunk, ky)
constexpr int W_DMA = W_SLOTS / 64;                   // 18
constexpr int W_BYTES = W_SLOTS * 16;                 // 18432
constexpr unsigned OOB = 0x80000000u;
constexpr int PIX_BYTES = 4 * 3 * 16 * 2;             // 384

__host__ __device__ inline uint16_t bf16_rne(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__host__ __device__ inline void split3(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  h = bf16_rne(x); float r = x - bf16_f(h);
  m = bf16_rne(r); r = r - bf16_f(m);
  l = bf16_rne(r);
}

struct Args {
  const void* x;     // S3 input
  const void* wp;    // packed weights [chunk][ky][W_SLOTS][16 B]
  const float* bias;
  const void* res;   // optional S3 residual
  void* y;           // S3 output
  int h, w, relu;
  int tiles_x, tiles_y;
  unsigned long long* dbg;   // lab: per-wave s_memtime stamps of block 0
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, void* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
}

#ifndef BLOCKS_PER_CU
#define BLOCKS_PER_CU 3
#endif
#ifndef APF
#define APF 3            // weight prefetch distance in (ky,kx) steps
#endif
constexpr int OUT_PIX = 400;                 // padded pixel stride of the output staging (bytes)
constexpr int RES_BYTES = 24 * 64 * 16;      // residual block, [part 24][px 64] 16-byte slots

// Weights do not go through LDS: the A operand of a (chunk, ky, kx, split, oc-half) is one
// perfectly coalesced 1 KB global load (pre-packed in lane order, L1/L2 resident), prefetched
// APF steps ahead into a register ring.  The pixel operand, which needs the shifted halo, is
// staged in LDS by LDS-DMA (one 16-channel chunk ahead) and read one step ahead.  The
// residual block is DMA'd into LDS at kernel start.  The epilogue transposes through LDS so
// that the S3 output leaves as contiguous 16-byte stores (2 rows x 12 KB per block).
__global__ __launch_bounds__(256, BLOCKS_PER_CU) void conv9_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char s_in[2][IN_BYTES];
  __shared__ __attribute__((aligned(16))) char s_res[RES_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int x0 = tx * TC, y0 = ty * TR;
  const size_t img = (size_t)a.h * a.w * PIX_BYTES;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (char*)const_cast<void*>(a.x) + n * img, 0, (int)img, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.wp), 0, 12 * W_BYTES, 0x00020000);
  unsigned long long st[8];
  st[0] = __builtin_amdgcn_s_memtime();

  // ---- residual block -> LDS (async), bias -> registers
  const int col = lane & 31, kg = lane >> 5;
  if (a.res) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        (char*)const_cast<void*>(a.res) + n * img, 0, (int)img, 0x00020000);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int part = wave + 4 * i;                    // 24 parts, one wave DMA each
      const int gy = y0 + (lane >> 5), gx = x0 + (lane & 31);
      unsigned o = (gy < a.h && gx < a.w) ? (unsigned)((gy * a.w + gx) * PIX_BYTES + part * 16) : OOB;
      dma16(rr, s_res + part * 1024, o);
    }
  }
  float bias[16];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[4 * g + e] = a.bias[wm * 32 + 8 * g + 4 * kg + e];

  unsigned in_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = wave + 4 * i, slot = j * 64 + lane;
    unsigned o = OOB;
    if (j < IN_DMA && slot < IN_SLOTS) {
      const int sk = slot / NPX, p = slot - sk * NPX;
      const int hr = p / HC, hc = p - hr * HC;
      const int gy = y0 - 1 + hr, gx = x0 - 1 + hc;
      if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.w)
        o = (unsigned)((gy * a.w + gx) * PIX_BYTES + (sk >> 1) * 32 + (sk & 1) * 16);
    }
    in_off[i] = o;
  }
  auto stage_in = [&](int c4, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wave + 4 * i;
      if (j < IN_DMA) dma16(rx, s_in[buf] + j * 1024, in_off[i] + (unsigned)(c4 * 96));
    }
  };
  const unsigned w_lane = (unsigned)(wm * 1024 + lane * 16);
  auto load_w = [&](int step, bf16x8 (&A)[3]) {        // step = (chunk*3 + ky)*3 + kx
#pragma unroll
    for (int s = 0; s < 3; ++s)
      A[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
          rw, (int)(w_lane + (unsigned)((step * 3 + s) * 2048)), 0, 0));
  };
  auto load_b = [&](const char* pin, int ky, int kx, bf16x8 (&B)[3]) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
      B[s] = *reinterpret_cast<const bf16x8*>(pin + (((s * 2 + kg) * NPX) + (wn + ky) * HC + col + kx) * 16);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  bf16x8 A[APF + 1][3], B[2][3];
  stage_in(0, 0);
#pragma unroll
  for (int p = 0; p < APF; ++p) load_w(p, A[p]);
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
  __syncthreads();
  st[1] = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    if (c4 + 1 < 4) stage_in(c4 + 1, (c4 + 1) & 1);
    const char* pin = s_in[c4 & 1];
    load_b(pin, 0, 0, B[0]);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int step = c4 * 9 + q;
      if (step + APF < 36) load_w(step + APF, A[(step + APF) % (APF + 1)]);
      if (q + 1 < 9) load_b(pin, (q + 1) / 3, (q + 1) % 3, B[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetches ahead of this step's MFMAs
      bf16x8 (&Aa)[3] = A[step % (APF + 1)];
      bf16x8 (&Bb)[3] = B[q & 1];
      // smallest terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[2], Bb[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[2], Bb[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[1], Bb[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[1], Bb[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[2], Bb[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[0], Bb[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[1], Bb[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[0], Bb[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa[0], Bb[0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c4 + 1 < 4) {
      // next chunk's pixels must have landed; the weight ring may stay in flight
      // in-order return: <= 3*APF loads outstanding means everything older than the weight
      // ring (the pixel DMA of the next chunk) has landed
      __builtin_amdgcn_s_waitcnt(0x0070 | ((3 * APF) & 15) | (((3 * APF) >> 4) << 14));
      __syncthreads();
    }
    st[2 + c4] = __builtin_amdgcn_s_memtime();
  }
  __syncthreads();                       // all waves are done with s_in: re-use it for the output

  // ---- epilogue: bias, ReLU, residual, exact 3-way split, transpose through LDS
  char* s_out = s_in[0];
  const int px = wn * 32 + col;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int oc = wm * 32 + 8 * g + 4 * kg;
    const int part = (oc >> 4) * 6 + ((oc >> 3) & 1);          // + 2*split
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = acc[4 * g + e] + bias[4 * g + e];
      if (a.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    if (a.res) {
      uint16_t rh[4], rm[4], rl[4];
      const char* pr = s_res + px * 16 + (oc & 7) * 2;
      *reinterpret_cast<uint2*>(rh) = *reinterpret_cast<const uint2*>(pr + (part + 0) * 1024);
      *reinterpret_cast<uint2*>(rm) = *reinterpret_cast<const uint2*>(pr + (part + 2) * 1024);
      *reinterpret_cast<uint2*>(rl) = *reinterpret_cast<const uint2*>(pr + (part + 4) * 1024);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (bf16_f(rh[e]) + bf16_f(rm[e])) + bf16_f(rl[e]);
    }
    uint16_t oh[4], om[4], ol[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(v[e], oh[e], om[e], ol[e]);
    char* po = s_out + px * OUT_PIX + (oc & 7) * 2;
    *reinterpret_cast<uint2*>(po + (part + 0) * 16) = *reinterpret_cast<uint2*>(oh);
    *reinterpret_cast<uint2*>(po + (part + 2) * 16) = *reinterpret_cast<uint2*>(om);
    *reinterpret_cast<uint2*>(po + (part + 4) * 16) = *reinterpret_cast<uint2*>(ol);
  }
  __syncthreads();
  {
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (char*)a.y + n * img, 0, (int)img, 0x00020000);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int q = tid + 256 * i;                 // 64 px x 24 parts
      const int p = q / 24, part = q - p * 24;
      const int gy = y0 + (p >> 5), gx = x0 + (p & 31);
      if (gy < a.h && gx < a.w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + p * OUT_PIX + part * 16);
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(int)))) int, v), ry,
            (gy * a.w + gx) * PIX_BYTES + part * 16, 0, 0);
      }
    }
  }
  st[6] = __builtin_amdgcn_s_memtime();
  if (a.dbg && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 7; ++i) a.dbg[wave * 8 + i] = st[i];
}

// ---- host side ---------------------------------------------------------------------------
static void to_s3(const std::vector<float>& x, int n, int h, int w, std::vector<uint16_t>& s3) {
  s3.assign((size_t)n * h * w * 192, 0);
  for (int b = 0; b < n; ++b)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < h * w; ++p) {
        uint16_t hh, mm, ll;
        split3(x[((size_t)b * C + c) * h * w + p], hh, mm, ll);
        size_t base = ((size_t)b * h * w + p) * 192 + (c >> 4) * 48 + (c & 15);
        s3[base] = hh; s3[base + 16] = mm; s3[base + 32] = ll;
      }
}
static void from_s3(const std::vector<uint16_t>& s3, int n, int h, int w, std::vector<float>& x) {
  x.assign((size_t)n * C * h * w, 0.f);
  for (int b = 0; b < n; ++b)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < h * w; ++p) {
        size_t base = ((size_t)b * h * w + p) * 192 + (c >> 4) * 48 + (c & 15);
        x[((size_t)b * C + c) * h * w + p] = (bf16_f(s3[base]) + bf16_f(s3[base + 16])) + bf16_f(s3[base + 32]);
      }
}
static void pack_w(const std::vector<float>& wt, std::vector<uint16_t>& wp) {   // wt[oc][ci][ky][kx]
  wp.assign((size_t)12 * W_SLOTS * 8, 0);
  for (int c4 = 0; c4 < 4; ++c4)
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx)
        for (int wm = 0; wm < 2; ++wm)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              int oc = wm * 32 + (lane & 31), ci = c4 * 16 + (lane >> 5) * 8 + e;
              uint16_t sp[3];
              split3(wt[((oc * C + ci) * 3 + ky) * 3 + kx], sp[0], sp[1], sp[2]);
              for (int s = 0; s < 3; ++s) {
                size_t slot = ((((size_t)(c4 * 3 + ky) * 3 + kx) * 3 + s) * 2 + wm) * 64 + lane;
                wp[slot * 8 + e] = sp[s];
              }
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float frand(uint32_t& s) { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.0f; }

int main(int argc, char** argv) {
  int h = argc > 1 ? atoi(argv[1]) : 134, w = argc > 2 ? atoi(argv[2]) : 320, n = argc > 3 ? atoi(argv[3]) : 1;
  int reps = argc > 4 ? atoi(argv[4]) : 50;
  uint32_t seed = 12345;
  std::vector<float> x((size_t)n * C * h * w), wt((size_t)C * C * 9), bias(C), res(x.size());
  for (auto& v : x) v = frand(seed) * 2.f - 0.5f;
  for (auto& v : res) v = frand(seed) - 0.5f;
  for (auto& v : wt) v = (frand(seed) - 0.5f) * 0.12f;
  for (auto& v : bias) v = (frand(seed) - 0.5f) * 0.1f;
  std::vector<uint16_t> xs, rs, wp;
  to_s3(x, n, h, w, xs); to_s3(res, n, h, w, rs); pack_w(wt, wp);
  // the split is exact
  { std::vector<float> back; from_s3(xs, n, h, w, back); size_t bad = 0;
    for (size_t i = 0; i < x.size(); ++i) bad += back[i] != x[i];
    printf("split3 round trip: %zu of %zu values differ\n", bad, x.size()); }
  void *dx, *dr, *dw, *dy; float* db;
  CK(hipMalloc(&dx, xs.size() * 2)); CK(hipMalloc(&dr, rs.size() * 2)); CK(hipMalloc(&dy, xs.size() * 2));
  CK(hipMalloc(&dw, wp.size() * 2)); CK(hipMalloc(&db, C * 4));
  CK(hipMemcpy(dx, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dr, rs.data(), rs.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice));
  unsigned long long* ddbg; CK(hipMalloc(&ddbg, 32 * 8)); CK(hipMemset(ddbg, 0, 32 * 8));
  Args a{dx, dw, db, dr, dy, h, w, 0, (w + TC - 1) / TC, (h + TR - 1) / TR, ddbg};
  dim3 grid(a.tiles_x * a.tiles_y * n);
  hipLaunchKernelGGL(conv9_kernel, grid, dim3(256), 0, 0, a);
  CK(hipDeviceSynchronize());
  if ((size_t)h * w * n <= 64 * 64) {
    std::vector<uint16_t> ys(xs.size());
    CK(hipMemcpy(ys.data(), dy, ys.size() * 2, hipMemcpyDeviceToHost));
    std::vector<float> y; from_s3(ys, n, h, w, y);
    double maxerr = 0, maxref = 0, max32 = 0;
    for (int b = 0; b < n; ++b) for (int oc = 0; oc < C; ++oc) for (int yy = 0; yy < h; ++yy) for (int xx = 0; xx < w; ++xx) {
      double s = bias[oc]; float s32 = 0.f;
      for (int ci = 0; ci < C; ++ci) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
        int iy = yy + ky - 1, ix = xx + kx - 1;
        if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
        float xv = x[((size_t)b * C + ci) * h * w + iy * w + ix], wv = wt[((oc * C + ci) * 3 + ky) * 3 + kx];
        s += (double)xv * wv; s32 = fmaf(xv, wv, s32);
      }
      s += res[((size_t)b * C + oc) * h * w + yy * w + xx];
      double r32 = (double)(s32 + bias[oc] + res[((size_t)b * C + oc) * h * w + yy * w + xx]);
      double got = y[((size_t)b * C + oc) * h * w + yy * w + xx];
      maxerr = fmax(maxerr, fabs(got - s)); maxref = fmax(maxref, fabs(s)); max32 = fmax(max32, fabs(r32 - s));
    }
    printf("check %dx%dx%d: max |bf16x9 - fp64| = %.3e, max |fp32 fma chain - fp64| = %.3e, max |ref| = %.3f\n",
           n, h, w, maxerr, max32, maxref);
  }
  { unsigned long long hd[32]; CK(hipMemcpy(hd, ddbg, sizeof(hd), hipMemcpyDeviceToHost));
    for (int wv = 0; wv < 4; ++wv) { printf("wave %d cycles:", wv); for (int i = 1; i < 7; ++i) printf(" %llu", hd[wv * 8 + i] - hd[wv * 8 + i - 1]); printf("  (prologue, chunk0..3, epilogue; ideal chunk = 2592)\n"); } }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(conv9_kernel, grid, dim3(256), 0, 0, a);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv9_kernel, grid, dim3(256), 0, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = 1e3 * ms / reps, gflop = 2.0 * C * C * 9 * (double)n * h * w / 1e9;
  printf("%dx%dx%d: %.2f us/launch, %.1f algorithmic TFLOP/s (fp32 MFMA kernel: 29.2 us at 1x134x320)\n",
         n, h, w, us, gflop / us * 1e-3 * 1e3 / 1e3 * 1e3);
  return 0;
}

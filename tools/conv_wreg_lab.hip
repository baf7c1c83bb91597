// Lab: weights-stationary variant of the 64 -> 64 conv3x3 -- every wave keeps its 32 x 576
// weight block in VGPRs (288 registers), workgroups are persistent (one per CU, tiles strided
// over the grid), LDS holds only the double-buffered input chunk.  Question: does removing
// the per-workgroup weight re-staging beat the loss of 3-workgroup co-residency?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/conv_wreg_lab.hip -o tools/conv_wreg_lab
#include "../tecogan-pytorch_amd/csrc/tg_conv3x3_mfma.hip"
#include <vector>
namespace tg { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);} }
using namespace tg;

struct WArgs { const float* x; const float* wpk; const float* bias; float* y; int h, w, tiles_x, ntiles, relu; };

#ifndef EPI
#define EPI 1     // 0: strided dword stores, 1: LDS-transposed 16-byte stores
#endif

__global__ __launch_bounds__(256, 1) void conv_wreg_kernel(WArgs a) {
  constexpr int WM = 2, OCB = 64, PH = WM + 2;
  constexpr int IN_ITEMS = PH * 2 * PW;
  constexpr int IN_FLOATS = PH * 2 * RS * 4;
  constexpr int I_PER_T = (IN_ITEMS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float s_in[2][IN_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_ep[4 * 32 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, lh = lane >> 5, ll = lane & 31;
  const int hw = a.h * a.w;
  const unsigned plane = (unsigned)hw * 4u;

  // ---- this wave's weights: [chunk 8][tap 9] x (4 k-steps) ----
  f32x4 wreg[8][9];
  {
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpk);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        wreg[ch][tap] = wp[((ch * 9 + tap) * 2 + lh) * OCB + wn * 32 + ll];
  }
  float bias[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = a.bias[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 64 * hw * 4, 0x00020000);
  const int b_off = ((wm * 2 + lh) * RS + ll) * 4;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x0 = tx * TW, y0 = ty * WM;
    unsigned voff[I_PER_T]; int lds_item[I_PER_T];
#pragma unroll
    for (int i = 0; i < I_PER_T; ++i) {
      int q = tid + i * 256;
      int r = q / (2 * PW), rem = q - r * (2 * PW);
      int hf = rem / PW, col = rem - hf * PW;
      int gy = y0 - 1 + r, gx = x0 - 1 + col;
      bool ok = q < IN_ITEMS && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
      voff[i] = ok ? (unsigned)((4 * hf * hw + gy * a.w + gx) * 4) : OOB;
      lds_item[i] = q < IN_ITEMS ? ((r * 2 + hf) * RS + col) * 4 : -1;
    }
    f32x4 rin[I_PER_T];
    auto load_chunk = [&](int ch) {
      const unsigned cbase = (unsigned)(ch * CK) * plane;
#pragma unroll
      for (int i = 0; i < I_PER_T; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rin[i][j] = buf_load(rs, voff[i] + cbase + (unsigned)j * plane);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
      for (int i = 0; i < I_PER_T; ++i)
        if (lds_item[i] >= 0) *reinterpret_cast<f32x4*>(s_in[buf] + lds_item[i]) = rin[i];
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int buf = ch & 1;
      if (ch + 1 < 8) load_chunk(ch + 1);
      const float* si = s_in[buf] + b_off;
      f32x4 bq[2];
      bq[0] = *reinterpret_cast<const f32x4*>(si);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = tap & 1, nxt = cur ^ 1;
        if (tap + 1 < 9) {
          const int ky = (tap + 1) / 3, kx = (tap + 1) % 3;
          bq[nxt] = *reinterpret_cast<const f32x4*>(si + (ky * 2 * RS + kx) * 4);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ch][tap][kk], bq[cur][kk], acc, 0, 0, 0);
      }
      if (ch + 1 < 8) store_chunk(buf ^ 1);
      __syncthreads();
    }
    // ---- epilogue ----
    const int px = x0 + ll, py = y0 + wm;
#if EPI == 0
    if (px < a.w && py < a.h) {
      float* yb = a.y + (long long)py * a.w + px;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int oc = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[r] + bias[r];
        yb[(long long)oc * hw] = (a.relu && v < 0.f) ? 0.f : v;
      }
    }
#else
    float* ep = s_ep + wave * (32 * 36);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bias[r];
      ep[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + ll] = (a.relu && v < 0.f) ? 0.f : v;
    }
    // each wave re-reads only its own region: no block barrier needed (wave-synchronous LDS)
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    if (py < a.h) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int idx4 = j * 64 + lane;
        int ol = idx4 >> 3, p4 = (idx4 & 7) * 4;
        int gx = x0 + p4;
        if (gx < a.w)
          *reinterpret_cast<f32x4*>(a.y + (long long)(wn * 32 + ol) * hw + (long long)py * a.w + gx) =
              *reinterpret_cast<const f32x4*>(ep + ol * 36 + p4);
      }
    }
#endif
  }
}

int main(int argc, char** argv) {
  int cin = 64, cout = 64, h = 134, w = 320;
  if (argc > 2) { h = atoi(argv[1]); w = atoi(argv[2]); }
  int reps = 50;
  size_t xn = (size_t)cin * h * w, yn = (size_t)cout * h * w;
  std::vector<float> hx(xn), hwt((size_t)cout * cin * 9), hb(cout);
  srand(1);
  for (auto& v : hx) v = rand() / (float)RAND_MAX * 2 - 1;
  for (auto& v : hwt) v = (rand() / (float)RAND_MAX * 2 - 1) / 24.f;
  for (auto& v : hb) v = rand() / (float)RAND_MAX - 0.5f;
  float *x, *y, *yref, *wraw, *wp, *bias;
  hipMalloc(&x, xn * 4); hipMalloc(&y, yn * 4); hipMalloc(&yref, yn * 4); hipMalloc(&wraw, hwt.size() * 4);
  hipMalloc(&bias, cout * 4);
  hipMemcpy(x, hx.data(), xn * 4, hipMemcpyHostToDevice);
  hipMemcpy(wraw, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hb.data(), cout * 4, hipMemcpyHostToDevice);
  hipMalloc(&wp, tg_conv3x3_packed_floats(cin, cout, 64) * 4);
  tg_conv3x3_pack(wraw, wp, cin, cout, 64, 0, 0);
  // reference: the product kernel
  tg_conv3x3_fwd(x, (int64_t)cin * h * w, cin, nullptr, 0, wp, 64, bias, nullptr, 0, yref, (int64_t)cout * h * w, 1, cin, cout, h, w, TG_ACT_RELU, 0);
  WArgs a{x, wp, bias, y, h, w, cdiv(w, TW), cdiv(w, TW) * cdiv(h, 2), 1};
  int grid = a.ntiles < 256 ? a.ntiles : 256;
  hipLaunchKernelGGL(conv_wreg_kernel, dim3(grid), dim3(256), 0, 0, a);
  hipDeviceSynchronize();
  printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
  std::vector<float> o(yn), r(yn);
  hipMemcpy(o.data(), y, yn * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), yref, yn * 4, hipMemcpyDeviceToHost);
  double md = 0; for (size_t i = 0; i < yn; ++i) md = fmax(md, fabs(o[i] - r[i]));
  printf("max |wreg - product| = %.3e\n", md);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_wreg_kernel, dim3(grid), dim3(256), 0, 0, a);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_wreg_kernel, dim3(grid), dim3(256), 0, 0, a);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = 1e3 * ms / reps, gflop = 2.0 * cin * 9 * cout * (double)h * w / 1e9;
  printf("wreg persistent: %.2f us  %.1f TF/s\n", us, gflop / us * 1e3);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i)
    tg_conv3x3_fwd(x, (int64_t)cin * h * w, cin, nullptr, 0, wp, 64, bias, nullptr, 0, yref, (int64_t)cout * h * w, 1, cin, cout, h, w, TG_ACT_RELU, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("product kernel : %.2f us  %.1f TF/s\n", 1e3 * ms / reps, gflop / (1e3 * ms / reps) * 1e3);
  return 0;
}

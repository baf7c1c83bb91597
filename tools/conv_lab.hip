// Stand-alone ablation / tile-shape lab for the conv3x3 MFMA kernel (no torch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_lab.hip -o tools/conv_lab
#include "../tecogan-pytorch_amd/csrc/tg_conv3x3_mfma.hip"
#include <vector>
#include <type_traits>
#include <cstdlib>
namespace tg { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);} }
using namespace tg;

template <int WM, int WN, int NT, int ABL, int OPT = 0, int STG = 12>
static float run(const char* name, Conv3x3Args a, int n, int reps, double gflop) {
  a.vec_ok = (a.w % 4 == 0);
  a.ksplit = 1;
  constexpr int OCB = WN * NT * 32;
  a.tiles_x = cdiv(a.w, TW); a.tiles_y = cdiv(a.h, WM); a.nocg = cdiv(a.cout, OCB); a.nchunk = cdiv(a.cin, CK);
  size_t lds = 2 * (size_t)((WM + 2) * 2 * RS * 4 + 9 * CK * OCB) * sizeof(float);
  unsigned blocks = a.tiles_x * a.tiles_y * a.nocg * n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, NT, false, ABL, OPT, STG>), dim3(blocks), dim3(WM * WN * 64), lds, 0, a);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, NT, false, ABL, OPT, STG>), dim3(blocks), dim3(WM * WN * 64), lds, 0, a);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float us = 1e3f * ms / reps;
  printf("%-28s WG=%4u x%3d lds=%6zu  %8.2f us  %7.2f TF/s  (%s)\n", name, blocks, WM * WN * 64, lds, us,
         gflop / us * 1e3, hipGetErrorString(hipGetLastError()));   // GFLOP/us = PFLOP/s; x1000 = TFLOP/s
  return us;
}

int main(int argc, char** argv) {
  int cin = 64, cout = 64, h = 134, w = 320, n = 1;
  if (argc > 4) { cin = atoi(argv[1]); cout = atoi(argv[2]); h = atoi(argv[3]); w = atoi(argv[4]); }
  size_t xn = (size_t)n * cin * h * w, yn = (size_t)n * cout * h * w;
  std::vector<float> hx(xn), hw_((size_t)cout * cin * 9);
  srand(1);
  for (auto& v : hx) v = rand() / (float)RAND_MAX * 2 - 1;
  for (auto& v : hw_) v = (rand() / (float)RAND_MAX * 2 - 1) / 24.f;
  float *x, *y, *wraw, *wp64, *wp32, *bias;
  hipMalloc(&x, xn * 4); hipMalloc(&y, yn * 4); hipMalloc(&wraw, hw_.size() * 4);
  hipMalloc(&bias, cout * 4); hipMemset(bias, 0, cout * 4);
  hipMemcpy(x, hx.data(), xn * 4, hipMemcpyHostToDevice);
  hipMemcpy(wraw, hw_.data(), hw_.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&wp64, tg_conv3x3_packed_floats(cin, cout, 64) * 4);
  hipMalloc(&wp32, tg_conv3x3_packed_floats(cin, cout, 32) * 4);
  tg_conv3x3_pack(wraw, wp64, cin, cout, 64, 0, 0);
  tg_conv3x3_pack(wraw, wp32, cin, cout, 32, 0, 0);
  Conv3x3Args a{};
  a.x = x; a.wpk = wp64; a.bias = bias; a.y = y; a.x_ns = (long long)cin * h * w; a.y_ns = (long long)cout * h * w;
  a.c1 = cin; a.cin = cin; a.cout = cout; a.h = h; a.w = w; a.act = TG_ACT_RELU;
  double gflop = 2.0 * cin * 9 * cout * (double)n * h * w / 1e9;
  printf("conv3x3 %d->%d @%dx%d  %.3f GFLOP\n", cin, cout, h, w, gflop);
  {
    int nb = 0;
    size_t lds = 2 * (size_t)((2 + 2) * 2 * RS * 4 + 9 * CK * 64) * sizeof(float);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)conv3x3_mfma_kernel<2, 2, 1, false, 0, 3>, 256, lds);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)conv3x3_mfma_kernel<2, 2, 1, false, 0, 3>);
    printf("occupancy API: %d blocks/CU for <2,2,1> OPT3 (dyn LDS %zu B, numRegs %d, static LDS %zu)\n", nb, lds, fa.numRegs, (size_t)fa.sharedSizeBytes);
    for (size_t l = 16384; l <= 65536; l += 8192) {
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)conv3x3_mfma_kernel<2, 2, 1, false, 0, 3>, 256, l);
      printf("   dyn LDS %6zu -> %d blocks/CU\n", l, nb);
    }
  }
  const int R = 50;
  run<2, 2, 1, 0>("<2,2,1> base", a, n, R, gflop);
  auto anatomy = [&](auto tag, const char* name) {
    constexpr int OPTV = decltype(tag)::value;
    long long* dbg; unsigned nb = 670;
    hipMalloc(&dbg, nb * 8 * sizeof(long long)); hipMemset(dbg, 0, nb * 8 * sizeof(long long));
    Conv3x3Args ad = a; ad.dbg = dbg;
    run<2, 2, 1, 16, OPTV>(name, ad, n, 5, gflop);
    std::vector<long long> h(nb * 8);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double pro = 0, iss = 0, mf = 0, sy = 0, life = 0;
    for (unsigned i = 0; i < nb; ++i) { pro += h[i*8+1]; iss += h[i*8+2]; mf += h[i*8+3]; sy += h[i*8+4]; life += h[i*8+5] - h[i*8]; }
    printf("  cycles/WG avg: prologue %.0f | load-issue %.0f  mfma-block %.0f  store+barrier %.0f | lifetime %.0f (ideal mfma 8x36x64 = 18432)\n",
           pro / nb, iss / nb, mf / nb, sy / nb, life / nb);
    // per-XCD view (s_memtime is only comparable inside one XCD)
    for (int x = 0; x < 8; ++x) {
      long long lo = -1, hi = 0; double lf = 0; int cnt = 0; long long first_end = -1;
      for (unsigned i = 0; i < nb; ++i) if ((h[i*8+6] & 15) == x) {
        if (lo < 0 || h[i*8] < lo) lo = h[i*8];
        if (h[i*8+5] > hi) hi = h[i*8+5];
        if (first_end < 0 || h[i*8+5] < first_end) first_end = h[i*8+5];
        lf += h[i*8+5] - h[i*8]; ++cnt;
      }
      long long last_start = 0; for (unsigned i = 0; i < nb; ++i) if ((h[i*8+6] & 15) == x && h[i*8] > last_start) last_start = h[i*8];
      if (cnt) printf("   XCD %d: %3d WGs  span %6lld  avg lifetime %6.0f  last start +%lld  first end +%lld\n", x, cnt, hi - lo, lf / cnt, last_start - lo, first_end - lo);
    }
    hipFree(dbg);
  };
  run<3, 2, 1, 0, 3>("<3,2,1> OPT3 (6 waves)", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3", a, n, R, gflop);
  run<3, 2, 1, 0, 3>("<3,2,1> OPT3 (6 waves)", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3", a, n, R, gflop);
  run<3, 2, 1, 0, 3>("<3,2,1> OPT3 (6 waves)", a, n, R, gflop);
  run<5, 2, 1, 0, 3>("<5,2,1> OPT3 (10 waves)", a, n, R, gflop);
  run<2, 2, 1, 32, 3>("<2,2,1> OPT3 no input restage", a, n, R, gflop);
  run<2, 2, 1, 64, 3>("<2,2,1> OPT3 no weight restage", a, n, R, gflop);
  run<2, 2, 1, 96, 3>("<2,2,1> OPT3 neither", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3 full", a, n, R, gflop);
  run<3, 2, 1, 0, 3>("<3,2,1> OPT3 (6 waves)", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3", a, n, R, gflop);
  run<3, 2, 1, 0, 3>("<3,2,1> OPT3 (6 waves)", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3", a, n, R, gflop);
  run<3, 2, 1, 0, 3>("<3,2,1> OPT3 (6 waves)", a, n, R, gflop);
  run<5, 2, 1, 0, 3>("<5,2,1> OPT3 (10 waves)", a, n, R, gflop);
  run<2, 2, 1, 32, 3>("<2,2,1> OPT3 no input restage", a, n, R, gflop);
  run<2, 2, 1, 64, 3>("<2,2,1> OPT3 no weight restage", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3 full", a, n, R, gflop);
  anatomy(std::integral_constant<int, 0>{}, "<2,2,1> OPT0 instrumented");
  anatomy(std::integral_constant<int, 1>{}, "<2,2,1> OPT1 instrumented");
  anatomy(std::integral_constant<int, 3>{}, "<2,2,1> OPT3 instrumented");
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3", a, n, R, gflop);
  run<2, 2, 1, 0, 7, 6>("<2,2,1> OPT7 stagger 6", a, n, R, gflop);
  run<2, 2, 1, 0, 7, 12>("<2,2,1> OPT7 stagger 12", a, n, R, gflop);
  run<2, 2, 1, 0, 7, 24>("<2,2,1> OPT7 stagger 24", a, n, R, gflop);
  run<2, 2, 1, 0, 7, 48>("<2,2,1> OPT7 stagger 48", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3 again", a, n, R, gflop);
  run<2, 2, 1, 0, 7, 12>("<2,2,1> OPT7 stagger 12 again", a, n, R, gflop);
  run<2, 2, 1, 0, 1>("<2,2,1> OPT1 lds-epilogue", a, n, R, gflop);
  run<2, 2, 1, 0, 2>("<2,2,1> OPT2 dma-weights", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3 both", a, n, R, gflop);
  run<2, 2, 1, 0, 0>("<2,2,1> base (again)", a, n, R, gflop);
  run<2, 2, 1, 0, 3>("<2,2,1> OPT3 both (again)", a, n, R, gflop);
  run<4, 1, 2, 0, 3>("<4,1,2> OPT3 both", a, n, R, gflop);
  run<2, 2, 1, 1>("<2,2,1> no-restage", a, n, R, gflop);
  run<2, 2, 1, 3>("<2,2,1> no-restage no-bar", a, n, R, gflop);
  run<2, 2, 1, 4>("<2,2,1> no-epilogue-store", a, n, R, gflop);
  run<2, 2, 1, 7>("<2,2,1> mfma+lds only", a, n, R, gflop);
  run<2, 2, 1, 15>("<2,2,1> mfma only", a, n, R, gflop);
  run<4, 1, 2, 0>("<4,1,2> base", a, n, R, gflop);
  run<4, 1, 2, 7>("<4,1,2> mfma+lds only", a, n, R, gflop);
  run<4, 2, 1, 0>("<4,2,1> base (8 waves)", a, n, R, gflop);
  run<4, 2, 1, 7>("<4,2,1> mfma+lds only", a, n, R, gflop);
  run<2, 1, 2, 0>("<2,1,2> base (2 waves)", a, n, R, gflop);
  run<1, 2, 1, 0>("<1,2,1> base (2 waves)", a, n, R, gflop);
  run<1, 1, 2, 0>("<1,1,2> base (1 wave)", a, n, R, gflop);
  run<2, 2, 1, 0>("<2,2,1> base again", a, n, R, gflop);
  a.wpk = wp32;
  run<4, 1, 1, 0>("<4,1,1> ocb32", a, n, R, gflop);
  run<2, 1, 1, 0>("<2,1,1> ocb32 (2 waves)", a, n, R, gflop);
  return 0;
}

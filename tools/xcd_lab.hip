// Lab: what does it take for two workgroups of ONE launch to exchange data through the XCD's L2 instead of
// through memory?  Ping-pong latency and message-passing correctness for every (store policy, load policy)
// pair between a workgroup and (a) a partner on the same XCD, (b) a partner on another XCD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/xcd_lab tools/xcd_lab.hip && ./tools/xcd_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Args {
  unsigned* flag_a;   // [pairs] ping
  unsigned* flag_b;   // [pairs] pong
  unsigned* payload;  // [pairs][64]
  unsigned* ids;      // [grid][2] xcc id, hw id
  long long* out;     // [pairs][4] ticks, polls exhausted, payload errors, rounds
  int rounds, st_pol, ld_pol, inv, limit, partner_step, payload_pol, payload_ld_pol;
};

template <int POL> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 4, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(v, r, 0, 0, POL);
}
template <int POL> __device__ __forceinline__ unsigned ld(const unsigned* p) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, 4, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, POL);
}
__device__ __forceinline__ void st_pol(unsigned* p, unsigned v, int pol) {
  switch (pol) { case 0: st<0>(p, v); break; case 1: st<1>(p, v); break; case 16: st<16>(p, v); break; default: st<17>(p, v); }
}
__device__ __forceinline__ unsigned ld_pol(const unsigned* p, int pol) {
  switch (pol) { case 0: return ld<0>(p); case 1: return ld<1>(p); case 16: return ld<16>(p); default: return ld<17>(p); }
}
__device__ __forceinline__ void inv(int kind) {
  if (kind == 1) asm volatile("buffer_inv sc0" ::: "memory");
  else if (kind == 2) asm volatile("buffer_inv sc1" ::: "memory");
  else if (kind == 3) asm volatile("buffer_inv sc0 sc1" ::: "memory");
}

// grid: workgroup b pings, workgroup b + partner_step pongs, for b < npairs ... we use ONE pair per launch:
// pinger = block 0, ponger = block partner_step; all other blocks idle (they exist so that the dispatcher's
// round-robin places block partner_step where we want it).
__global__ void pingpong(Args a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  unsigned xcc = 0, hwid = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (lane == 0) { a.ids[2 * b] = xcc; a.ids[2 * b + 1] = hwid; }
  if (b != 0 && b != a.partner_step) return;
  const bool pinger = b == 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  long long exhausted = 0, perr = 0;
  int done = 0;
  for (int i = 1; i <= a.rounds && !exhausted; ++i) {
    done = i;
    if (pinger) {
      // payload (64 lanes x 4 bytes), then release, then the flag
      st_pol(a.payload + lane, (unsigned)i * 64u + lane, a.payload_pol);
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_s_barrier();
      if (lane == 0) st_pol(a.flag_a, (unsigned)i, a.st_pol);
      // wait for the pong
      if (lane == 0) {
        int polls = 0;
        for (;;) {
          inv(a.inv);
          if (ld_pol(a.flag_b, a.ld_pol) == (unsigned)i) break;
          if (++polls > a.limit) { ++exhausted; break; }
        }
      }
      __builtin_amdgcn_s_barrier();
    } else {
      if (lane == 0) {
        int polls = 0;
        for (;;) {
          inv(a.inv);
          if (ld_pol(a.flag_a, a.ld_pol) == (unsigned)i) break;
          if (++polls > a.limit) { ++exhausted; break; }
        }
      }
      __builtin_amdgcn_s_barrier();
      inv(a.inv);
      const unsigned v = ld_pol(a.payload + lane, a.payload_ld_pol);
      if (v != (unsigned)i * 64u + lane) ++perr;
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_s_barrier();
      if (lane == 0) st_pol(a.flag_b, (unsigned)i, a.st_pol);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  // wave-reduce the payload errors
  for (int o = 32; o > 0; o >>= 1) perr += __shfl_down(perr, o);
  if (lane == 0) {
    long long* o = a.out + (pinger ? 0 : 4);
    o[0] = t1 - t0; o[1] = exhausted; o[2] = perr; o[3] = done;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  const int grid = 64;
  unsigned *fa, *fb, *pl, *ids; long long* out;
  CK(hipMalloc(&fa, 256)); CK(hipMalloc(&fb, 256)); CK(hipMalloc(&pl, 1024)); CK(hipMalloc(&ids, grid * 8)); CK(hipMalloc(&out, 64));
  std::vector<unsigned> hid(grid * 2);
  const char* pname[] = {"plain", "sc0", "?", "?"};
  auto pn = [](int p) { return p == 0 ? "plain " : p == 1 ? "sc0   " : p == 16 ? "sc1   " : "sc0sc1"; };
  const int pols[] = {0, 1, 16, 17};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int partner : {8, 1}) {      // same XCD (block 8 -> XCD 0) / other XCD (block 1 -> XCD 1)
    printf("---- partner block %d\n", partner);
    for (int sp : pols) for (int lp : pols) for (int iv : {0, 1, 2}) for (int pp : {0, 16}) {
      const int plp = lp;            // payload loads with the flag's policy
      Args a{fa, fb, pl, ids, out, 2000, sp, lp, iv, 4000, partner, pp, plp};
      CK(hipMemset(fa, 0, 256)); CK(hipMemset(fb, 0, 256)); CK(hipMemset(pl, 0, 1024)); CK(hipMemset(out, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(pingpong, dim3(grid), dim3(64), 0, 0, a);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long ho[8]; CK(hipMemcpy(ho, out, 64, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hid.data(), ids, grid * 8, hipMemcpyDeviceToHost));
      printf("flag st %s ld %s inv %d payload st %s | xcc %u/%u cu %08x/%08x | %7.1f ns/round  exhausted %lld/%lld  payload errors %lld\n",
             pn(sp), pn(lp), iv, pn(pp), hid[0], hid[2 * partner], hid[1], hid[2 * partner + 1], ms * 1e6 / (ho[3] > 0 ? ho[3] : 1), ho[1], ho[5], ho[6]);
      fflush(stdout);
    }
  }
  (void)pname;
  return 0;
}

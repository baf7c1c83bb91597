// Error channel, version, and the whole-frame FRNet.step plan.
#include <new>
#include <vector>

#include "tg_common.h"

namespace tg {
static thread_local char g_err[512] = "ok";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tg

extern "C" int tg_version(void) { return TG_ABI_MAJOR * 100 + 1; }
extern "C" const char* tg_last_error_string(void) { return tg::g_err; }

#ifndef TG_BUILD_FLAGS_STR
#define TG_BUILD_FLAGS_STR "unknown (not built by csrc/build.sh)"
#endif
#ifndef TG_FILE_FLAGS_STR
#define TG_FILE_FLAGS_STR ""
#endif
namespace tg { int wres_lab_bits(); }
extern "C" const char* tg_build_info(void) {
  static char s[1024];
  if (!s[0])
    snprintf(s, sizeof(s), "lab=%d wres_lab_bits=%d flags=%s file_flags=%s", (int)TG_LAB, tg::wres_lab_bits(),
             TG_BUILD_FLAGS_STR, TG_FILE_FLAGS_STR);
  return s;
}

// ---------------------------------------------------------------------------
// FRNet.step plan (codes/models/networks/tecogan_nets.py:227-252): the launch
// list of one recurrent frame, resolved once per (shape, weights).  Owns no
// device memory: activations live in the caller's workspace.
// ---------------------------------------------------------------------------
struct tg_frnet_plan {
  tg_frnet_cfg cfg;
  std::vector<tg_layer_weights> L;
  float *A, *B, *FLOW, *S2D, *U1, *U2, *PART;
  float *FA, *FB, *FPART, *FLOW2;   // FNet's own buffers (phase 1 may overlap phase 2 of the previous frame)
  float* WZ;                        // packed output-conv weights for the fused HR stage
  bool wz_ready;
  int32_t* CHAINF;                  // per-tile flags of the chained SRNet launch (tg_conv3x3_wino_chain)
  bool chain_ready;                 // flags zeroed
  unsigned epoch;                   // one per chained launch
  int chain_layers;                 // layers in the chained launch of this plan's shape (0: none)
  void* RESWS;                      // workspace of the LDS-resident SRNet body launch (exchange buffer + flags)
  bool res_ready;                   // its flags zeroed
  // fail-safe of the chained launch: fault counter in pinned host memory (the kernel adds to it
  // with system scope), looked at by every later call on the plan
  int32_t* chain_err;               // hipHostMalloc, 64 bytes; null when the allocation failed (chain then off)
  bool chain_disabled;              // a fault was reported: one launch per layer until the plan re-arms (below)
  int chain_faults;                 // faults reported so far
  int chain_poll_limit;
  // Recovery from a TRANSIENT fault (a co-tenant that kept a workgroup off the GPU for a moment):
  // after `rearm_wait` clean per-layer frames the one-launch body is tried again; a fault of the re-armed body doubles
  // the wait (exponential back-off, capped), a permanent co-tenant therefore costs one faulted clip every 2^k * first
  // frames and never more.  rearm_first == 0: never re-arm (the round-5 behaviour).
  int rearm_first, rearm_wait, clean_frames, rearms;
  hipEvent_t rearm_fence;           // recorded behind the first per-layer frame after a report: the re-arm waits until the
  bool fence_set;                   // GPU has passed it, so a late fault of a PRE-report launch is never blamed on the re-armed body
  int fh, fw, launches;
  int st_launch[24];
  double st_flops[24], st_bytes[24];
};

// phase bits: 1 = FNet (lr_curr, lr_prev -> flow slot), 2 = warp + SRNet (flow slot, hr_prev -> hr_out)
static int step_impl(tg_frnet_plan* p, const float* lr_curr, const float* lr_prev,
                     const float* hr_prev, float* hr_out, uint8_t* u8_out, tg_stream_t st,
                     unsigned mask, bool dry, int phases = 3, int slot = 0,
                     const float* flow_ext = nullptr);

// Fail-safe of the chained launch: a host read of the pinned fault counter.  The first call that
// sees faults reports them (TG_E_HIP) and turns the chain off for the rest of the plan's life.
static int chain_poll(tg_frnet_plan* p) {
  if (!p->chain_err) return TG_OK;
  const int pending = __atomic_load_n(p->chain_err, __ATOMIC_RELAXED);
  if (pending == 0) return TG_OK;
  __atomic_fetch_sub(p->chain_err, pending, __ATOMIC_RELAXED);
  p->chain_faults += pending;
  // Faults that arrive after the report come from chained launches that were already enqueued when
  // the host noticed the first one (the frames of that period were declared invalid then): counted,
  // not reported again -- everything enqueued since the report uses one launch per layer.
  if (p->chain_disabled) return TG_OK;
  p->chain_disabled = true;
  p->clean_frames = 0;
  p->rearm_wait = p->rearm_wait ? (p->rearm_wait < (1 << 20) ? 2 * p->rearm_wait : p->rearm_wait) : p->rearm_first;
  p->fence_set = false;
  tg::set_error("chained SRNet launch: %d workgroup(s) timed out waiting for a producer tile; the frames "
                "enqueued on this plan since the previous successful check are INVALID.  The plan now runs "
                "one launch per layer (TG_WINO_CHAIN=0 selects that from the start)%s", pending,
                p->rearm_first > 0 ? " and tries the one-launch body again after a back-off of clean frames" : "");
  return TG_E_HIP;
}

// Called once per frame enqueued through phase 2 while the one-launch body is off: counts clean frames, and re-arms the
// body once the back-off has passed, no new fault has arrived, and the GPU is past the first per-layer frame.
static void chain_rearm_tick(tg_frnet_plan* p, tg_stream_t st) {
  if (!p->chain_disabled || !p->chain_err || p->rearm_first <= 0) return;
  if (!p->fence_set) {
    if (!p->rearm_fence && hipEventCreateWithFlags(&p->rearm_fence, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError(); p->rearm_fence = nullptr; p->rearm_first = 0; return;     // no fence -> never re-arm
    }
    if (hipEventRecord(p->rearm_fence, (hipStream_t)st) != hipSuccess) { (void)hipGetLastError(); return; }
    p->fence_set = true;
    return;
  }
  if (++p->clean_frames < p->rearm_wait) return;
  if (hipEventQuery(p->rearm_fence) != hipSuccess) { (void)hipGetLastError(); return; }   // not there yet: ask again next frame
  if (__atomic_load_n(p->chain_err, __ATOMIC_RELAXED) != 0) return;                        // (chain_poll counts it first)
  p->chain_disabled = false;
  p->res_ready = false;             // exchange buffers / flags are zeroed again in front of the next one-launch body
  p->chain_ready = false;
  p->rearms += 1;
}

static size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

// largest split-K partial buffer any layer of this shape asks for
static size_t fnet_partial_floats(const tg_frnet_cfg* c) {
  size_t best = 0;
  int hh = c->h, ww = c->w, cin = 2 * c->in_nc;
  auto consider = [&](int ci, int co) {
    int ks = tg_conv3x3_pick_ksplit(c->n, ci, co, hh, ww);
    if (ks > 1) { size_t v = (size_t)ks * c->n * co * hh * ww; if (v > best) best = v; }
  };
  const int enc[3] = {32, 64, 128}, dec[3] = {256, 128, 64};
  for (int e = 0; e < 3; ++e) { consider(cin, enc[e]); consider(enc[e], enc[e]); cin = enc[e]; hh /= 2; ww /= 2; }
  for (int d = 0; d < 3; ++d) { consider(cin, dec[d]); consider(dec[d], dec[d]); cin = dec[d]; hh *= 2; ww *= 2; }
  consider(cin, 32);
  // SRNet layers pick split-K too when the frame itself is tiny
  hh = c->h; ww = c->w;
  consider((c->scale * c->scale + 1) * c->in_nc, c->nf);
  consider(c->nf, c->nf);
  return best;
}

static const int CHAIN_MAX_LAYERS = 24;
static void carve(const tg_frnet_cfg* c, size_t off[15]) {
  size_t hw = (size_t)c->h * c->w, n = c->n;
  size_t o = 0;
  const size_t sr = c->fnet_only ? 0 : 1;                      // an FNet-only plan has no SRNet regions
  off[0] = o; o += sr * align64(n * 64 * hw);                  // A
  off[1] = o; o += sr * align64(n * 64 * hw);                  // B
  off[2] = o; o += align64(n * 2 * hw);                        // FLOW
  off[3] = o; o += sr * align64(n * c->scale * c->scale * c->in_nc * hw);  // S2D
  off[4] = o; o += sr * align64(n * c->nf * 4 * hw);           // U1
  off[5] = o; o += (c->scale == 4) ? sr * align64(n * c->nf * 16 * hw) : 0;  // U2
  off[6] = o; o += sr * align64(fnet_partial_floats(c));       // PART (split-K partial sums, SRNet)
  off[7] = o; o += align64(n * 64 * hw);                       // FA   (FNet ping)
  off[8] = o; o += align64(n * 64 * hw);                       // FB   (FNet pong)
  off[9] = o; o += align64(fnet_partial_floats(c));            // FPART (split-K partial sums, FNet)
  off[10] = o; o += align64(n * 2 * hw);                       // FLOW2 (second flow slot)
  off[11] = o; o += sr * 2048;                                 // WZ (A operand of the fused output-conv contraction)
  off[12] = o;                                                 // CHAINF (int32 flags: 24 layers x 16-tile workgroups + error counter)
  o += sr * align64((size_t)CHAIN_MAX_LAYERS * n * ((c->h + 1) / 2) * ((c->w + 31) / 32) + 16);
  off[13] = o;                                                 // RESWS (tg_conv3x3_wino_res.hip: exchange buffer + flags; n == 1 only)
  {
    const int64_t rb = (sr && n == 1) ? tg::conv3x3_wino_resident_ws_bytes(c->h, c->w) : 0;
    o += align64((size_t)(rb + 3) / 4);
  }
  off[14] = o;
}

static int cfg_ok(const tg_frnet_cfg* c) {
  return c && c->in_nc == 3 && c->out_nc >= 1 && c->out_nc <= 4 && c->nf >= 1 && c->nf <= 64 &&
         c->nb >= 0 && (c->scale == 2 || c->scale == 4) &&
         (c->up_mode == TG_UP_BICUBIC || c->up_mode == TG_UP_BILINEAR) && c->n >= 1 &&
         c->h >= 8 && c->w >= 8;
}

extern "C" size_t tg_frnet_workspace_floats(const tg_frnet_cfg* cfg) {
  if (!cfg_ok(cfg)) return 0;
  size_t off[15];
  carve(cfg, off);
  return off[14];
}

extern "C" int tg_frnet_plan_create(const tg_frnet_cfg* cfg, const tg_layer_weights* layers,
                                    int n_layers, float* workspace, tg_frnet_plan** out) {
  TG_REQUIRE(cfg && layers && workspace && out, TG_E_ARG, "frnet_plan_create: null pointer");
  TG_REQUIRE(cfg_ok(cfg), TG_E_SHAPE,
             "frnet_plan_create: unsupported cfg (in_nc=3, out_nc<=4, nf<=64, scale 2|4, h,w>=8)");
  int nup = cfg->scale == 4 ? 2 : 1;
  int need = 14 + 1 + 2 * cfg->nb + nup + 1;
  TG_REQUIRE(n_layers == need, TG_E_ARG, "frnet_plan_create: %d layers given, %d expected",
             n_layers, need);
  for (int i = 0; i < n_layers; ++i)
    TG_REQUIRE(layers[i].w && layers[i].b, TG_E_ARG, "frnet_plan_create: layer %d null", i);
  tg_frnet_plan* p = new (std::nothrow) tg_frnet_plan();
  TG_REQUIRE(p, TG_E_ARG, "frnet_plan_create: out of host memory");
  p->cfg = *cfg;
  p->L.assign(layers, layers + n_layers);
  size_t off[15];
  carve(cfg, off);
  p->WZ = workspace + off[11]; p->wz_ready = false;
  p->RESWS = workspace + off[13]; p->res_ready = false;
  p->CHAINF = reinterpret_cast<int32_t*>(workspace + off[12]); p->chain_ready = false; p->epoch = 0; p->chain_layers = 0;
  p->chain_err = nullptr; p->chain_disabled = false; p->chain_faults = 0; p->chain_poll_limit = tg::TG_CHAIN_POLL_LIMIT_DEFAULT;
  p->rearm_first = 64; p->rearm_wait = 0; p->clean_frames = 0; p->rearms = 0; p->rearm_fence = nullptr; p->fence_set = false;
  if (!cfg->fnet_only) {
    void* hp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hp) {
      p->chain_err = static_cast<int32_t*>(hp);
      for (int i = 0; i < 16; ++i) p->chain_err[i] = 0;
    } else {
      (void)hipGetLastError();      // no fault channel -> no chained launch (per-layer launches are always safe)
    }
  }
  p->FA = workspace + off[7]; p->FB = workspace + off[8]; p->FPART = workspace + off[9];
  p->FLOW2 = workspace + off[10];
  p->PART = workspace + off[6];
  p->A = workspace + off[0]; p->B = workspace + off[1]; p->FLOW = workspace + off[2];
  p->S2D = workspace + off[3]; p->U1 = workspace + off[4];
  p->U2 = cfg->scale == 4 ? workspace + off[5] : nullptr;
  p->fh = cfg->h / 8 * 8; p->fw = cfg->w / 8 * 8;
  for (int k = 0; k < 24; ++k) { p->st_launch[k] = 0; p->st_flops[k] = 0; p->st_bytes[k] = 0; }
  // dry run: per-kernel-class launch counts, algorithmic flops and bytes of one frame
  // (dummy non-null pointers; nothing is dereferenced or launched)
  static float dummy;
  step_impl(p, &dummy, &dummy, &dummy, &dummy, nullptr, nullptr, 0, true);
  p->launches = 0;
  for (int k = 0; k < 24; ++k) p->launches += p->st_launch[k];
  p->launches -= p->st_launch[8];  // quantise only runs when a u8 output is requested
  *out = p;
  return TG_OK;
}

extern "C" void tg_frnet_plan_destroy(tg_frnet_plan* plan) {
  if (!plan) return;
  // (a chained launch still in flight may yet add to the counter: the caller synchronises before
  // destroying a plan, as for the workspace it owns)
  if (plan->chain_err) (void)hipHostFree(plan->chain_err);
  if (plan->rearm_fence) (void)hipEventDestroy(plan->rearm_fence);
  delete plan;
}
extern "C" int tg_frnet_plan_launches(const tg_frnet_plan* plan) { return plan ? plan->launches : 0; }

// Launch classes = distinct kernel symbols (what rocprofv3 --stats groups by).
enum {
  K_CONV64_R2 = 0,  // conv3x3_mfma_kernel<2,2,1>
  K_CONV64_R4 = 1,  // conv3x3_mfma_kernel<4,1,2>
  K_CONV32 = 2,     // conv3x3_mfma_kernel<4,1,1>
  K_CONVT = 3,      // convt3x3s2_mfma_kernel<2|4,2>
  K_SMALL = 4,      // conv3x3_small_kernel<COUT>
  K_WARP = 5,       // flowup_warp_s2d_kernel<S,3>
  K_POOL = 6,
  K_UPSAMPLE = 7,
  K_QUANT = 8,
  K_FINAL = 9,      // splitk_finalize_kernel
  K_CONV64_KS = 10, // conv3x3_mfma_kernel<1,2,1,...,2>: in-workgroup K split (few tiles)
  K_CONVT_Z = 11,   // convt3x3s2_mfma_kernel<4,2,true>: last up-sampling layer + output-conv contraction
  K_TAIL = 12,      // convout_tail_kernel: 9-tap shift-add + residual + uint8
  K_CONV_ONESHOT = 13,  // conv3x3_oneshot_kernel: few tiles, cin <= 64, whole K range in flight
  K_CONV_WINO = 14,     // conv3x3_wino_kernel: Winograd F(2x2,3x3) form of the large 64-channel-group layers
  K_WINO_CHAIN = 15,    // conv3x3_wino_chain_kernel: SRNet's conv_in + residual blocks as ONE launch
  K_WINO_RES = 16,      // conv3x3_wino_resident_kernel: the same layers on persistent, LDS-resident workgroups (one 134x320-class frame)
  K_COUNT = 17
};

// The HR stage as two launches instead of three and without the 64-channel HR tensor: the last
// ConvTranspose2d emits the 27 tap planes of conv_out (tecogan_nets.py:119-131), a streaming
// kernel shift-adds them.  TG_HR_FUSE=0 selects the unfused launches (lab / A-B).
static bool hr_fuse_enabled() {
  static const int v = TG_LAB_ENV("TG_HR_FUSE", 1);
  return v != 0;
}
// SRNet's 1 + 2*nb full-resolution layers as one chained launch (tg_conv3x3_wino_chain) when a layer
// has more 16-tile workgroups than fit on the GPU at once (768) but only a few rounds of them:
// measured against one launch per layer (tools/wino_chain_probe.py) -15 % at 2 clips of 134x320,
// -6.5 % at 4, -4.6 % at 268x640, +-0 at 8 clips, and +3..15 % (slower) for a single 134x320 clip, whose
// 670 workgroups are all resident at once and therefore cannot pipeline.  TG_WINO_CHAIN=0 / 1: never / whenever eligible.
static bool chain_wanted(long long ntile) {
  static const int v = [] { const char* e = getenv("TG_WINO_CHAIN"); return e && *e ? atoi(e) : -1; }();
  if (v >= 0) return v != 0;
  return ntile > 768 && ntile <= 3000;
}
// The LDS-resident form (tg_conv3x3_wino_res.hip) whenever the frame fits one 8x24 block per CU and is large
// enough for the Winograd form to be the per-layer choice (a single 134x320-class frame).  TG_WINO_RES=0
// selects the per-layer / chained launches, 1 the resident launch wherever it is supported (A/B runs, tests).
static bool resident_wanted(bool layer_prefers_wino) {
  static const int v = [] { const char* e = getenv("TG_WINO_RES"); return e && *e ? atoi(e) : -1; }();
  return v >= 0 ? v != 0 : layer_prefers_wino;     // 1: whenever supported (tests run small frames through it)
}
// the fused tail writes (n, H, W, c) uint8 frames for any n; the unfused quantise pass only n == 1
static bool plan_u8_ok(const tg_frnet_plan* p) {
  return p->cfg.n == 1 || (hr_fuse_enabled() && p->cfg.out_nc <= 3 && p->cfg.nf <= 64);
}

static int step_impl(tg_frnet_plan* p, const float* lr_curr, const float* lr_prev,
                     const float* hr_prev, float* hr_out, uint8_t* u8_out, tg_stream_t st,
                     unsigned mask, bool dry, int phases, int slot, const float* flow_ext) {
  const tg_frnet_cfg& c = p->cfg;
  // phase 2 may read the flow of its frame from another plan's batched FNet pass
  float* const flow_buf = flow_ext ? const_cast<float*>(flow_ext) : (slot ? p->FLOW2 : p->FLOW);
  bool active = (phases & 1) != 0;     // which phase the launches being issued belong to
  float* part_buf = p->FPART;
  const int n = c.n, h = c.h, w = c.w, s = c.scale;
  const int64_t hw = (int64_t)h * w;
  int li = 0;
  float *A = p->FA, *B = p->FB;
  int rc = TG_OK;
  if (!dry && (phases & 2) && mask == 0xFFFFFFFFu) chain_rearm_tick(p, st);
  // account + (unless dry / masked out) launch
  auto go = [&](int kind, double flops, double bytes, auto&& fn) {
    if (rc != TG_OK) return;
    if (dry) {
      p->st_launch[kind] += 1; p->st_flops[kind] += flops; p->st_bytes[kind] += bytes;
      return;
    }
    if (active && (mask & (1u << kind))) rc = fn();
  };
  // conv3x3 (+bias+act) [+ MaxPool2d(2,2) when pool]; y receives the final tensor.
  auto conv = [&](const float* x, int64_t xns, int c1, const float* x2, int64_t x2ns, int cin,
                  int cout, int hh, int ww, int act, const float* res, int64_t rns, float* y,
                  int64_t yns, bool pool = false, float* pool_tmp = nullptr) {
    const tg_layer_weights lw = p->L[li++];
    int ocb = tg_conv3x3_pick_ocb(cout);
    int kind = ocb == 32 ? K_CONV32
                         : (tg::conv3x3_rows_per_wg(ocb, (long long)n * hh * ww) == 4 ? K_CONV64_R4
                                                                                       : K_CONV64_R2);
    double px = (double)n * hh * ww;
    double fl = 2.0 * cin * 9 * cout * px;
    double by = 4.0 * px * (cin + cout + (res ? cout : 0)) + 4.0 * 9 * cin * cout;
    if (lw.u && tg_conv3x3_prefers_wino(n, cin, cout, hh, ww)) {
      float* yw = pool ? pool_tmp : y;
      go(K_CONV_WINO, fl, by, [&] {
        return tg::conv3x3_wino_launch(x, xns, c1, x2, x2ns, lw.u, lw.b, res, rns, nullptr, 0, yw,
                                       pool ? (int64_t)cout * hh * ww : yns, n, cin, cout, hh, ww, act, st);
      });
      if (pool)
        go(K_POOL, 0, 4.0 * n * cout * (hh * ww + (hh / 2) * (ww / 2)),
           [&] { return tg_maxpool2_fwd(yw, y, n * cout, hh, ww, st); });
      return;
    }
    int ks = res ? 1 : tg_conv3x3_pick_ksplit(n, cin, cout, hh, ww);
    if (ks == 1 && kind == K_CONV64_R2 && tg::conv3x3_uses_wg_ksplit(n, cin, cout, hh, ww))
      kind = tg::conv3x3_uses_oneshot(n, cin, cout, hh, ww) ? K_CONV_ONESHOT : K_CONV64_KS;   // launcher's rule
    if (ks > 1) {
      go(kind, fl, by, [&] {
        return tg::conv3x3_splitk_conv(x, xns, c1, x2, x2ns, lw.w, ocb, n, cin, cout, hh, ww, ks,
                                       part_buf, st);
      });
      go(K_FINAL, 0, 4.0 * px * cout * (ks + 1), [&] {
        return tg::conv3x3_splitk_finalize(part_buf, ks, lw.b, act, pool ? 1 : 0, y, n, cout, hh, ww,
                                           st);
      });
      return;
    }
    float* yc = pool ? pool_tmp : y;
    go(kind, fl, by, [&] {
      return tg_conv3x3_fwd(x, xns, c1, x2, x2ns, lw.w, ocb, lw.b, res, rns, yc,
                            pool ? (int64_t)cout * hh * ww : yns, n, cin, cout, hh, ww, act, st);
    });
    if (pool)
      go(K_POOL, 0, 4.0 * n * cout * (hh * ww + (hh / 2) * (ww / 2)),
         [&] { return tg_maxpool2_fwd(yc, y, n * cout, hh, ww, st); });
  };
  // ---- FNet (tecogan_nets.py:67-82) ------------------------------------------
  int hh = h, ww = w;
  const int enc[3] = {32, 64, 128};
  int cin = 2 * c.in_nc;
  const float* src = nullptr;
  for (int e = 0; e < 3; ++e) {
    int co = enc[e];
    if (e == 0)
      conv(lr_curr, c.in_nc * hw, c.in_nc, lr_prev, c.in_nc * hw, cin, co, hh, ww, TG_ACT_LRELU02,
           nullptr, 0, A, (int64_t)co * hh * ww);
    else
      conv(src, (int64_t)cin * hh * ww, cin, nullptr, 0, cin, co, hh, ww, TG_ACT_LRELU02, nullptr,
           0, A, (int64_t)co * hh * ww);
    // second conv of the block + MaxPool2d: pooled result lands in A (B is scratch)
    conv(A, (int64_t)co * hh * ww, co, nullptr, 0, co, co, hh, ww, TG_ACT_LRELU02, nullptr, 0, A,
         (int64_t)co * (hh / 2) * (ww / 2), true, B);
    hh /= 2; ww /= 2; cin = co;
    float* t = A; A = B; B = t;   // pooled result now in B
    src = B;
  }
  const int dec[3] = {256, 128, 64};
  for (int d = 0; d < 3; ++d) {
    int co = dec[d];
    conv(src, (int64_t)cin * hh * ww, cin, nullptr, 0, cin, co, hh, ww, TG_ACT_LRELU02, nullptr, 0,
         A, (int64_t)co * hh * ww);
    conv(A, (int64_t)co * hh * ww, co, nullptr, 0, co, co, hh, ww, TG_ACT_LRELU02, nullptr, 0, B,
         (int64_t)co * hh * ww);
    {
      float *pi = B, *po = A; int ph = hh, pw = ww;
      go(K_UPSAMPLE, 0, 4.0 * n * co * ph * pw * 5.0,
         [&] { return tg_upsample_fwd(pi, po, n * co, ph, pw, 2, TG_UP_BILINEAR, 1.0f, st); });
    }
    hh *= 2; ww *= 2; cin = co;
    float* t = A; A = B; B = t;
    src = B;
  }
  conv(src, (int64_t)cin * hh * ww, cin, nullptr, 0, cin, 32, hh, ww, TG_ACT_LRELU02, nullptr, 0, A,
       (int64_t)32 * hh * ww);
  {
    const tg_layer_weights lw = p->L[li++];
    float* ai = A; int fh_ = hh, fw_ = ww;
    go(K_SMALL, 2.0 * 32 * 9 * 2 * n * fh_ * fw_, 4.0 * n * fh_ * fw_ * 34, [&] {
      return tg_conv3x3_small_fwd(ai, (int64_t)32 * fh_ * fw_, lw.w, lw.b, nullptr, TG_UP_NONE, 1,
                                  flow_buf, (int64_t)2 * fh_ * fw_, n, 32, 2, fh_, fw_,
                                  TG_ACT_TANH24, st);
    });
  }
  // ---- pad + upsample + warp + space_to_depth (tecogan_nets.py:238-250) -------
  active = (phases & 2) != 0;
  part_buf = p->PART;
  const int s2dc = s * s * c.in_nc;
  go(K_WARP, 0,
     4.0 * n * ((double)c.in_nc * s * s * hw * 2 + 2.0 * p->fh * p->fw), [&] {
       return tg_flowup_warp_s2d_fwd(flow_buf, p->fh, p->fw, hr_prev, p->S2D, s2dc * hw, nullptr, n,
                                     c.in_nc, h, w, s, c.up_mode, st);
     });
  // ---- SRNet (tecogan_nets.py:136-147) ----------------------------------------
  A = p->A; B = p->B;
  const int nf = c.nf;
  // conv_in joins the chain when it has a Winograd form itself (cin = 3 + 48 at 4x; 3 + 12 at 2x is
  // below the kernel's 16-channel stage and runs the direct form as its own launch)
  const bool in_wino = p->L[li].u && tg_conv3x3_prefers_wino(n, c.in_nc + s2dc, nf, h, w);
  const int skip = in_wino ? 0 : 1;
  const int nchain = 1 + 2 * c.nb - skip;
  bool chain = p->chain_err && !p->chain_disabled && nf <= 64 && nchain >= 2 && nchain <= CHAIN_MAX_LAYERS &&
               chain_wanted((long long)n * tg::cdiv(h, 2) * tg::cdiv(w, 32)) &&
               tg_conv3x3_prefers_wino(n, nf, nf, h, w);
  for (int i = skip; i < 1 + 2 * c.nb && chain; ++i) chain = p->L[li + i].u != nullptr;
  bool resident = p->chain_err && !p->chain_disabled && n == 1 && nchain >= 2 && nchain <= CHAIN_MAX_LAYERS &&
                  resident_wanted(tg_conv3x3_prefers_wino(n, nf, nf, h, w) != 0) && (skip || c.in_nc + s2dc <= 64) && tg::conv3x3_wino_resident_ok(n, nf, h, w);
  for (int i = skip; i < 1 + 2 * c.nb && resident; ++i) resident = p->L[li + i].u != nullptr && p->L[li + i].b != nullptr;
  if (resident) chain = true;
  bool ct_fold = false;
  if (chain) {
    if (skip)
      conv(lr_curr, c.in_nc * hw, c.in_nc, p->S2D, s2dc * hw, c.in_nc + s2dc, nf, h, w, TG_ACT_RELU,
           nullptr, 0, A, nf * hw);
    tg_wino_layer cl[CHAIN_MAX_LAYERS];
    double fl = 0, by = 0;
    const double px = (double)n * h * w;
    for (int k = 0; k < nchain; ++k) {
      const int i = k + skip;                 // layer index inside SRNet: 0 = conv_in, odd = conv1, even = conv2
      const tg_layer_weights lw = p->L[li + k];
      tg_wino_layer& d = cl[k];
      const bool first = i == 0, conv2 = !first && (i % 2 == 0);
      d.x = first ? lr_curr : (conv2 ? B : A);
      d.x2 = first ? p->S2D : nullptr;
      d.u_packed = lw.u; d.bias = lw.b;
      d.res = conv2 ? A : nullptr;
      d.y = (first || conv2) ? A : B;
      d.x_nstride = first ? c.in_nc * hw : nf * hw;
      d.x2_nstride = first ? s2dc * hw : 0;
      d.res_nstride = nf * hw; d.y_nstride = nf * hw;
      d.c1 = first ? c.in_nc : nf;
      d.cin = first ? c.in_nc + s2dc : nf;
      d.act = conv2 ? TG_ACT_NONE : TG_ACT_RELU;
      fl += 2.0 * d.cin * 9 * nf * px;
      by += 4.0 * px * (d.cin + nf + (conv2 ? nf : 0)) + 4.0 * 9 * d.cin * nf;
    }
    li += nchain;
    p->chain_layers = nchain;
    if (resident) {
      // SRNet's first up-sampling layer rides on the same launch (tg_conv3x3_wino_res.hip: transposed-conv tail) when
      // the plan holds its weights in that form (4x only: at 2x the single transposed conv is the Z-mode launch)
      ct_fold = s == 4 && p->L[li].u != nullptr && p->L[li].b != nullptr;   // (the caller opts in by supplying the tail's pack)
      if (ct_fold) { fl += 2.0 * nf * 9 * nf * px; by += 4.0 * px * nf * 4.0 + 4.0 * 9 * nf * nf; }
      go(K_WINO_RES, fl, by, [&] {
        if (!p->res_ready) {
          if (hipMemsetAsync(p->RESWS, 0, (size_t)tg::conv3x3_wino_resident_ws_bytes(h, w), (hipStream_t)st) != hipSuccess)
            return tg::check_launch("wino_resident workspace memset");
          p->res_ready = true;
        }
        if (++p->epoch == 0) p->epoch = 1;
        tg_wres_convt ct{p->L[li].u, p->L[li].b, p->U1, TG_ACT_RELU};
        return tg::conv3x3_wino_resident_launch(cl, nchain, nf, h, w, p->RESWS, p->chain_err, p->epoch << 5,
                                                p->chain_poll_limit, st, ct_fold ? &ct : nullptr);
      });
    } else
    go(K_WINO_CHAIN, fl, by, [&] {
      if (!p->chain_ready) {
        const size_t ints = (size_t)tg_conv3x3_wino_chain_flag_ints(nchain, n, h, w);
        if (hipMemsetAsync(p->CHAINF, 0, ints * sizeof(int32_t), (hipStream_t)st) != hipSuccess)
          return tg::check_launch("wino_chain flags memset");
        p->chain_ready = true;
      }
      if (++p->epoch == 0) p->epoch = 1;
      return tg::conv3x3_wino_chain_launch(cl, nchain, n, nf, h, w, p->CHAINF, p->chain_err, p->epoch,
                                           p->chain_poll_limit, st);
    });
  } else {
    conv(lr_curr, c.in_nc * hw, c.in_nc, p->S2D, s2dc * hw, c.in_nc + s2dc, nf, h, w, TG_ACT_RELU,
         nullptr, 0, A, nf * hw);
    for (int b = 0; b < c.nb; ++b) {
      conv(A, nf * hw, nf, nullptr, 0, nf, nf, h, w, TG_ACT_RELU, nullptr, 0, B, nf * hw);
      conv(B, nf * hw, nf, nullptr, 0, nf, nf, h, w, TG_ACT_NONE, A, nf * hw, A, nf * hw);
    }
  }
  const bool fuse = hr_fuse_enabled() && c.out_nc <= 3 && nf <= 64;
  const tg_layer_weights lw_up1 = p->L[li++];
  const tg_layer_weights lw_up2 = s == 4 ? p->L[li++] : tg_layer_weights{nullptr, nullptr, nullptr};
  const tg_layer_weights lw_out = p->L[li++];
  const double hpx = (double)n * s * s * hw;
  if (fuse) {
    // WZ is derived from conv_out's weights once per plan (plans are rebuilt when weights change)
    if (!dry && !p->wz_ready && (phases & 2)) {
      rc = tg_convt_pack_wz(lw_out.w, p->WZ, c.out_nc, nf, st);
      p->wz_ready = rc == TG_OK;
    }
    float* zbuf = s == 4 ? p->U2 : p->U1;          // the 64-channel tensor is never written: its space holds the planes
    const int zh = s * h / 2, zw = s * w / 2;      // input size of the last up-sampling layer
    const float* zin = A;
    if (s == 4) {
      float* ai = A;
      if (!ct_fold)
        go(K_CONVT, 2.0 * nf * 9 * nf * n * hw, 4.0 * n * hw * nf * 5.0, [&] {
          return tg_convt3x3s2_fwd(ai, nf * hw, lw_up1.w, lw_up1.b, p->U1, nf * 4 * hw, n, nf, nf, h, w,
                                   TG_ACT_RELU, st);
        });
      zin = p->U1;
    }
    const tg_layer_weights lw_last = s == 4 ? lw_up2 : lw_up1;
    if (dry && tg::convt_z_split_rule(n, zh, zw, -1)) p->st_launch[K_CONVT_Z] += 1;   // (two launches: split tail, round 6)
    go(K_CONVT_Z, 2.0 * nf * 9 * nf * n * zh * zw + 2.0 * nf * 9 * c.out_nc * hpx,
       4.0 * n * zh * zw * nf + 4.0 * hpx * 9 * c.out_nc, [&] {
         return tg_convt3x3s2_z_fwd(zin, (int64_t)nf * zh * zw, lw_last.w, lw_last.b, p->WZ, c.out_nc, zbuf,
                                    (int64_t)32 * s * s * hw, n, nf, nf, zh, zw, TG_ACT_RELU, st);
       });
    go(K_TAIL, 0, 4.0 * hpx * (9 * c.out_nc + c.out_nc), [&] {
      return tg_convout_tail(zbuf, (int64_t)32 * s * s * hw, c.out_nc, lw_out.b, lr_curr, c.up_mode, s, hr_out,
                             (int64_t)c.out_nc * s * s * hw, u8_out, n, s * h, s * w, st);
    });
    return rc;
  }
  if (!ct_fold) {
    float* ai = A;
    go(K_CONVT, 2.0 * nf * 9 * nf * n * hw, 4.0 * n * hw * nf * 5.0, [&] {
      return tg_convt3x3s2_fwd(ai, nf * hw, lw_up1.w, lw_up1.b, p->U1, nf * 4 * hw, n, nf, nf, h, w,
                               TG_ACT_RELU, st);
    });
  }
  const float* top = p->U1;
  if (s == 4) {
    go(K_CONVT, 2.0 * nf * 9 * nf * n * 4 * hw, 4.0 * n * 4 * hw * nf * 5.0, [&] {
      return tg_convt3x3s2_fwd(p->U1, nf * 4 * hw, lw_up2.w, lw_up2.b, p->U2, nf * 16 * hw, n, nf, nf,
                               2 * h, 2 * w, TG_ACT_RELU, st);
    });
    top = p->U2;
  }
  bool u8_fused = false;
  {
    u8_fused = !dry && u8_out &&
               tg_conv3x3_small_can_fuse_u8(top, (int64_t)nf * s * s * hw, hr_out,
                                            (int64_t)c.out_nc * s * s * hw, n, nf, s * h, s * w);
    go(K_SMALL, 2.0 * nf * 9 * c.out_nc * hpx, 4.0 * hpx * (nf + c.out_nc), [&] {
      return tg_conv3x3_small_fwd_u8(top, (int64_t)nf * s * s * hw, lw_out.w, lw_out.b, lr_curr, c.up_mode, s,
                                     hr_out, (int64_t)c.out_nc * s * s * hw, u8_fused ? u8_out : nullptr,
                                     n, nf, c.out_nc, s * h, s * w, TG_ACT_NONE, st);
    });
  }
  if ((u8_out && !u8_fused) || dry) {
    // only when the output conv cannot emit the uint8 frame itself (w % 4 != 0, unaligned planes)
    go(K_QUANT, 0, 5.0 * c.out_nc * s * s * hw,
       [&] { return tg_quantize_u8_hwc(hr_out, u8_out, c.out_nc, s * h, s * w, st); });
  }
  return rc;
}

extern "C" int tg_frnet_step(tg_frnet_plan* p, const float* lr_curr, const float* lr_prev,
                             const float* hr_prev, float* hr_out, uint8_t* u8_out,
                             tg_stream_t st) {
  return tg_frnet_step_masked(p, lr_curr, lr_prev, hr_prev, hr_out, u8_out, 0xFFFFFFFFu, st);
}

extern "C" int tg_frnet_step_masked(tg_frnet_plan* p, const float* lr_curr, const float* lr_prev,
                                    const float* hr_prev, float* hr_out, uint8_t* u8_out,
                                    unsigned kind_mask, tg_stream_t st) {
  TG_REQUIRE(p && lr_curr && lr_prev && hr_prev && hr_out, TG_E_ARG, "frnet_step: null pointer");
  TG_REQUIRE(!u8_out || plan_u8_ok(p), TG_E_ARG, "frnet_step: u8 output needs n == 1 (or the fused HR stage)");
  TG_REQUIRE(!p->cfg.fnet_only, TG_E_ARG, "frnet_step: the plan was created FNet-only");
  if (int rc = chain_poll(p)) return rc;
  return step_impl(p, lr_curr, lr_prev, hr_prev, hr_out, u8_out, st, kind_mask, false);
}

extern "C" float* tg_frnet_plan_flow(tg_frnet_plan* p, int flow_slot) {
  if (!p || (flow_slot != 0 && flow_slot != 1)) return nullptr;
  return flow_slot ? p->FLOW2 : p->FLOW;
}

extern "C" int tg_frnet_step_srnet(tg_frnet_plan* p, const float* lr_flow, const float* lr_curr,
                                   const float* hr_prev, float* hr_out, uint8_t* u8_out,
                                   tg_stream_t st) {
  TG_REQUIRE(p && lr_flow && lr_curr && hr_prev && hr_out, TG_E_ARG, "frnet_step_srnet: null pointer");
  TG_REQUIRE(!p->cfg.fnet_only, TG_E_ARG, "frnet_step_srnet: the plan was created FNet-only");
  TG_REQUIRE(!u8_out || plan_u8_ok(p), TG_E_ARG, "frnet_step_srnet: u8 output needs n == 1 (or the fused HR stage)");
  if (int rc = chain_poll(p)) return rc;
  return step_impl(p, lr_curr, nullptr, hr_prev, hr_out, u8_out, st, 0xFFFFFFFFu, false, 2, 0, lr_flow);
}

extern "C" int tg_frnet_step_phase(tg_frnet_plan* p, int phases, int flow_slot, const float* lr_curr,
                                   const float* lr_prev, const float* hr_prev, float* hr_out,
                                   uint8_t* u8_out, tg_stream_t st) {
  TG_REQUIRE(p && lr_curr, TG_E_ARG, "frnet_step_phase: null pointer");
  TG_REQUIRE(phases >= 1 && phases <= 3 && (flow_slot == 0 || flow_slot == 1), TG_E_ARG,
             "frnet_step_phase: phases=%d slot=%d", phases, flow_slot);
  TG_REQUIRE(!(phases & 1) || lr_prev, TG_E_ARG, "frnet_step_phase: phase 1 needs lr_prev");
  TG_REQUIRE(!(phases & 2) || (hr_prev && hr_out), TG_E_ARG, "frnet_step_phase: phase 2 needs hr_prev/hr_out");
  TG_REQUIRE(!u8_out || plan_u8_ok(p), TG_E_ARG, "frnet_step_phase: u8 output needs n == 1 (or the fused HR stage)");
  TG_REQUIRE(!(phases & 2) || !p->cfg.fnet_only, TG_E_ARG, "frnet_step_phase: the plan was created FNet-only");
  if (int rc = chain_poll(p)) return rc;
  return step_impl(p, lr_curr, lr_prev, hr_prev, hr_out, u8_out, st, 0xFFFFFFFFu, false, phases,
                   flow_slot);
}

// measurement helper: enqueue the masked launch list `reps` times from C (no host round
// trip per replay, so single-launch kernel classes are timed GPU-bound)
extern "C" int tg_frnet_replay(tg_frnet_plan* p, const float* lr_curr, const float* lr_prev,
                               const float* hr_prev, float* hr_out, unsigned kind_mask, int reps,
                               tg_stream_t st) {
  TG_REQUIRE(p && lr_curr && lr_prev && hr_prev && hr_out && reps > 0 && !p->cfg.fnet_only, TG_E_ARG,
             "frnet_replay: bad argument");
  if (int rc = chain_poll(p)) return rc;
  for (int i = 0; i < reps; ++i) {
    int rc = step_impl(p, lr_curr, lr_prev, hr_prev, hr_out, nullptr, st, kind_mask, false);
    if (rc != TG_OK) return rc;
  }
  return TG_OK;
}

extern "C" int tg_frnet_plan_chain_status(tg_frnet_plan* p, int* faults_total, int* chain_active) {
  TG_REQUIRE(p, TG_E_ARG, "frnet_plan_chain_status: null plan");
  const int rc = chain_poll(p);
  if (faults_total) *faults_total = p->chain_faults;
  if (chain_active) *chain_active = (p->chain_layers > 0 && !p->chain_disabled) ? 1 : 0;
  return rc;
}

extern "C" int tg_frnet_plan_set_chain_poll_limit(tg_frnet_plan* p, int poll_limit) {
  TG_REQUIRE(p, TG_E_ARG, "frnet_plan_set_chain_poll_limit: null plan");
  p->chain_poll_limit = poll_limit;
  return TG_OK;
}

extern "C" int tg_frnet_plan_set_chain_rearm(tg_frnet_plan* p, int first_after_frames) {
  TG_REQUIRE(p && first_after_frames >= 0, TG_E_ARG, "frnet_plan_set_chain_rearm: bad argument");
  p->rearm_first = first_after_frames;
  if (p->chain_disabled && p->rearm_wait == 0) p->rearm_wait = first_after_frames;
  return TG_OK;
}

extern "C" int tg_frnet_plan_chain_rearms(const tg_frnet_plan* p, int* rearms, int* current_wait_frames) {
  TG_REQUIRE(p, TG_E_ARG, "frnet_plan_chain_rearms: null plan");
  if (rearms) *rearms = p->rearms;
  if (current_wait_frames) *current_wait_frames = p->rearm_wait;
  return TG_OK;
}

extern "C" int tg_frnet_plan_kinds(void) { return K_COUNT; }

extern "C" const char* tg_frnet_kind_name(int kind) {
  static const char* names[K_COUNT] = {
      "conv3x3_mfma_kernel<2,2,1>", "conv3x3_mfma_kernel<4,1,2>", "conv3x3_mfma_kernel<4,1,1>",
      "convt3x3s2_mfma_kernel",      "conv3x3_small_kernel",       "flowup_warp_s2d_kernel",
      "maxpool2_kernel",             "upsample_kernel",            "quantize_u8_hwc_kernel",
      "splitk_finalize_kernel",      "conv3x3_mfma_kernel<1,2,1,KS=2>",
      "convt3x3s2_mfma_kernel<Z>",   "convout_tail_kernel",        "conv3x3_oneshot_kernel",
      "conv3x3_wino_kernel",         "conv3x3_wino_chain_kernel",  "conv3x3_wino_resident_kernel"};
  return (kind >= 0 && kind < K_COUNT) ? names[kind] : "?";
}

extern "C" int tg_frnet_plan_kind_stats(const tg_frnet_plan* p, int kind, int* launches,
                                        double* flops, double* bytes) {
  TG_REQUIRE(p && kind >= 0 && kind < K_COUNT, TG_E_ARG, "plan_kind_stats: bad argument");
  if (launches) *launches = p->st_launch[kind];
  if (flops) *flops = p->st_flops[kind];
  if (bytes) *bytes = p->st_bytes[kind];
  return TG_OK;
}

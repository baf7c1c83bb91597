/*
 * tecogan_hip.h -- C ABI of libtecogan_hip.so (MI355X / gfx950 only).
 *
 * The upstream reference (skycrapers/TecoGAN-PyTorch) has NO native code and
 * therefore no FFI of its own: its hot path is Python calling stock ATen ops.
 * This header is the boundary a maintainer binds instead of those ATen calls;
 * each entry cites the reference call site it replaces (paths relative to the
 * upstream repo root).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - all tensors are fp32, contiguous NCHW unless a batch stride is given;
 *     pointers are DEVICE pointers owned by the caller (PyTorch allocates);
 *     the library never retains them past the call and never allocates device memory
 *     (a frame plan owns one 64-byte pinned-host fault counter, see tg_frnet_plan_chain_status);
 *   - `*_nstride` = distance in floats between consecutive batch items, so a
 *     channel-slice of a larger buffer can be read / written in place (this is
 *     how every torch.cat on the path is folded away);
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues
 *     kernels on it (no hidden synchronisation, hipGraph-capturable);
 *   - every entry returns 0 (TG_OK) or a negative TG_E_* code;
 *     tg_last_error_string() describes the last failure on this thread.
 *     No C++ exception crosses the boundary.
 */
#ifndef TECOGAN_HIP_H
#define TECOGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tg_stream_t;

enum {
  TG_OK = 0,
  TG_E_SHAPE = -1, /* unsupported / inconsistent shape */
  TG_E_ARG = -2,   /* null pointer, bad enum */
  TG_E_HIP = -3    /* HIP runtime error at launch */
};

/* fused epilogue activation of the conv kernels */
enum {
  TG_ACT_NONE = 0,
  TG_ACT_RELU = 1,    /* nn.ReLU                 tecogan_nets.py:96,113,121,125 */
  TG_ACT_LRELU02 = 2, /* nn.LeakyReLU(0.2)       tecogan_nets.py:24-64, 369   */
  TG_ACT_TANH24 = 3   /* torch.tanh(.) * 24      tecogan_nets.py:80           */
};

/* up-sampling flavour: get_upsampling_func, codes/utils/net_utils.py:85-97 */
enum {
  TG_UP_NONE = 0,
  TG_UP_BICUBIC = 1, /* BicubicUpsampler (BD), net_utils.py:101-156 */
  TG_UP_BILINEAR = 2 /* F.interpolate bilinear align_corners=False (BI), :86-89 */
};

/* ABI version, major * 100 + minor.  The major number changes whenever a struct declared here
 * changes size or an entry changes its signature (a host compiled against another major must not
 * call in: check tg_version() / 100 == TG_ABI_MAJOR after dlopen).
 *   1xx: round-1 ABI.   2xx (minor 1: + chained training launches, phased_masked): tg_layer_weights gained `u` (24 bytes, was 16);
 *   tg_frnet_plan_chain_error_index replaced by tg_frnet_plan_chain_status. */
#define TG_ABI_MAJOR 2
int tg_version(void);
const char* tg_last_error_string(void);
/* How this library was built (a static string): "lab=<0|1> wres_lab_bits=<n> flags=<hipcc flags of csrc/build.sh>
 * file_flags=<per-file TG_FILE_FLAGS>".  The in-tree library is lab=0 wres_lab_bits=0 by construction (build.sh
 * refuses EXTRA_FLAGS / TG_LAB for it, the sources #error on a lab switch without TG_LAB); tests/test_abi_cpu.py
 * asserts exactly that, so an ablation build can never pass for the product. */
const char* tg_build_info(void);

/* ------------------------------------------------------------------------
 * 3x3 stride-1 pad-1 convolution, fp32 MFMA implicit GEMM.
 * Replaces every nn.Conv2d(.,.,3,1,1) on the path:
 *   FNet tecogan_nets.py:23-65, SRNet conv_in :111-113, ResidualBlock :92-98,
 *   D conv_in :367-369.
 * Fused: channel-concat of two sources (torch.cat :71, :141), bias,
 * activation, residual add (`self.conv(x) + x` :98).
 *
 * Weights must be pre-packed with tg_conv3x3_pack (layout private to the
 * kernel: [oc-group][cin-chunk][tap][8 cin][ocb oc], zero padded).
 *   ocb: output channels per workgroup, 32 or 64 (use tg_conv3x3_pick_ocb).
 * ---------------------------------------------------------------------- */
int tg_conv3x3_pick_ocb(int cout);
size_t tg_conv3x3_packed_floats(int cin, int cout, int ocb);
/* w_oihw: (cout, cin, 3, 3) as nn.Conv2d.weight.  transposed=1: w is
 * (cin, cout, 3, 3) as nn.ConvTranspose2d.weight (used by tg_convt3x3s2_fwd).
 * transposed=2: data-gradient packing of a Conv2d weight: the op maps the conv's
 * cout -> cin, i.e. call with cin := conv.cout, cout := conv.cin and w the
 * ORIGINAL (conv.cout, conv.cin, 3, 3) tensor; taps are rotated by 180 degrees. */
int tg_conv3x3_pack(const float* w, float* w_packed, int cin, int cout, int ocb,
                    int transposed, tg_stream_t stream);

int tg_conv3x3_fwd(
    const float* x, int64_t x_nstride, int c1,   /* channels [0,c1) from x          */
    const float* x2, int64_t x2_nstride,         /* channels [c1,cin) from x2 (or NULL) */
    const float* w_packed, int ocb, const float* bias /* may be NULL */,
    const float* res, int64_t res_nstride,       /* optional residual, added after act */
    float* y, int64_t y_nstride,
    int n, int cin, int cout, int h, int w, int act, tg_stream_t stream);

/* conv3x3 with PHASE-RESTRICTED TAPS: the strided layers run through their space-to-depth
 * embeddings -- Conv2d(k4, s2, p1) of the discriminator blocks (tecogan_nets.py:322-340) is a 3x3
 * conv on space_to_depth(x, 2) with 4*ci input channels, its data gradient a 3x3 conv onto
 * 4*ci output channels, and the data gradient of ConvTranspose2d(k3, s2, p1, op1) (:119-126) a
 * 3x3 conv on space_to_depth(dY, 2).  In those embedded weights every sub-pixel phase
 * (py, px) owns only some tap rows / columns; the others are zeros.  This entry skips them:
 *   tapsel 1: the phase of the INPUT channel chunk ((channel / cphase): py = phase >> 1,
 *             px = phase & 1) selects the taps; tapsel 2: the phase of the OUTPUT block does;
 *   taps_phase0 / taps_phase1: tap set used by phase coordinate 0 / 1 along each axis,
 *             0 = {0,1,2}, 1 = {0,1}, 2 = {1,2}, 3 = {1}.
 * Results equal tg_conv3x3_fwd on the same (zero-padded) weights up to summation order. */
int tg_conv3x3_fwd_phased(const float* x, int64_t x_nstride, const float* w_packed, int ocb,
                          const float* bias, float* y, int64_t y_nstride, int n, int cin,
                          int cout, int h, int w, int act, int tapsel, int cphase,
                          int taps_phase0, int taps_phase1, tg_stream_t stream);
/* nn.Conv2d(ci, co, 4, 2, 1, bias=False) -- the discriminator blocks (tecogan_nets.py:322-340) -- taken directly
 * (K = 16 ci implicit GEMM on the fp32 matrix cores, no space-to-depth copy, no phase masks) and its data gradient
 * (the 4x4 / stride-2 transposed convolution; optionally multiplied by act'(act_y) of the layer below on the way
 * out: act_y (n, ci, h, w) the activation OUTPUT that was this conv's input, act = TG_ACT_RELU | TG_ACT_LRELU02).
 * x, dx: (n, ci, h, w); y, g: (n, co, h/2, w/2), contiguous.  Weights: tg_conv4x4s2_pack from OIHW (co, ci, 4, 4)
 * into two buffers of tg_conv4x4s2_packed_floats(ci, co) floats (either may be NULL).
 * tg_conv4x4s2_supported: ci, co multiples of 64, h even, w a multiple of 64 -- or the small maps of the deeper
 * blocks, w = 32 (h % 8 == 0) / w = 16 (h % 16 == 0): 32 pixels of the matrix tile are then 2 / 4 rows, the input
 * channels are split over several workgroups and a second launch adds the partial sums in a fixed order; such calls
 * need tg_conv4x4s2_workspace_floats(.., dgrad) floats of workspace (0: none, workspace may be NULL).  Other shapes
 * keep the embedded form (tg_conv3x3_fwd_phased on tg_space_to_depth(x, 2)), and so does the weight gradient. */
int tg_conv4x4s2_supported(int n, int ci, int co, int h, int w);
size_t tg_conv4x4s2_packed_floats(int ci, int co);
int tg_conv4x4s2_pack(const float* w, float* w_fwd, float* w_dgrad, int ci, int co, tg_stream_t stream);
size_t tg_conv4x4s2_workspace_floats(int n, int ci, int co, int h, int w, int dgrad);
int tg_conv4x4s2_fwd(const float* x, const float* w_fwd, float* y, float* workspace, int n, int ci, int co, int h, int w,
                     tg_stream_t stream);
int tg_conv4x4s2_dgrad(const float* g, const float* w_dgrad, const float* act_y, int act, float* dx, float* workspace,
                       int n, int ci, int co, int h, int w, tg_stream_t stream);
/* STRIDE-2 3x3 conv on small frames: y(oy, ox) = act(sum_k w[k] x(2 oy - 1 + ky, 2 ox - 1 + kx) + bias),
 * zero outside x (n, cin, 2 h_out, 2 w_out) -- the data gradient of ConvTranspose2d(k3, s2, p1, op1)
 * (tecogan_nets.py:119-126) taken directly (K = 9 cin) instead of through the 4 cin-channel phased
 * embedding; w_packed = tg_conv3x3_pack(W as (cout = ci, cin = co), ocb 64).  relu_mask as in
 * tg_conv3x3_fwd_masked.  tg_conv3x3s2_supported: cin, cout <= 64 and <= 1024 one-row tiles. */
int tg_conv3x3s2_supported(int n, int cin, int cout, int h_out, int w_out);
int tg_conv3x3s2_fwd(const float* x, int64_t x_nstride, const float* w_packed, const float* bias,
                     const float* relu_mask, int64_t mask_nstride, float* y, int64_t y_nstride, int n,
                     int cin, int cout, int h_out, int w_out, int act, tg_stream_t stream);
/* the same with a ReLU-backward mask in the epilogue (see tg_conv3x3_fwd_masked): the data gradient
 * of a transposed conv delivers dZ of the ReLU layer below it directly (relu_mask may be NULL) */
int tg_conv3x3_fwd_phased_masked(const float* x, int64_t x_nstride, const float* w_packed, int ocb,
                                 const float* bias, const float* relu_mask, int64_t mask_nstride,
                                 float* y, int64_t y_nstride, int n, int cin, int cout, int h, int w,
                                 int act, int tapsel, int cphase, int taps_phase0, int taps_phase1,
                                 tg_stream_t stream);
/* The same launch with the channel chunks split over `ksplit` workgroup sets (deterministic: partial sums
 * [ksplit][n][cout][h][w] in `partials`, added in a fixed order by the finalize launch, which also applies
 * bias / act) -- for the critic's deeper blocks, whose 48-192 tiles cannot fill 256 CUs.
 * tg_conv3x3_phased_pick_ksplit returns the recommended factor (1 = use tg_conv3x3_fwd_phased). */
int tg_conv3x3_phased_pick_ksplit(int n, int cin, int cout, int h, int w, int ocb);
int tg_conv3x3_fwd_phased_splitk(const float* x, int64_t x_nstride, const float* w_packed, int ocb,
                                 const float* bias, float* y, int n, int cin, int cout, int h, int w,
                                 int act, int tapsel, int cphase, int taps_phase0, int taps_phase1,
                                 int ksplit, float* partials, tg_stream_t stream);
/* tg_conv3x3_fwd followed by a ReLU-backward mask in the same epilogue:
 *   y = relu_mask > 0 ? y : 0      (relu_mask: (n,cout,h,w) fp32, e.g. a ReLU layer's output)
 * Used by the training tape: the data-gradient conv of a layer (weights packed with
 * transposed = 2) then delivers dZ of the PRECEDING ReLU layer directly, without a separate
 * activation-backward pass (torch.autograd's threshold_backward). */
int tg_conv3x3_fwd_masked(const float* x, int64_t x_nstride, int c1, const float* x2,
                          int64_t x2_nstride, const float* w_packed, int ocb,
                          const float* bias, const float* res, int64_t res_nstride,
                          const float* relu_mask, int64_t mask_nstride, float* y,
                          int64_t y_nstride, int n, int cin, int cout, int h, int w, int act,
                          tg_stream_t stream);

/* The same convolution (nn.Conv2d(k3, s1, p1), tecogan_nets.py:85-100,116) in the Winograd
 * F(2x2, 3x3) form: 16 instead of 36 fp32 MFMA multiplies per 2x2 output tile.  All arithmetic is
 * fp32 (the transform matrices hold only 0, +-1, +-1/2); results equal tg_conv3x3_fwd up to fp32
 * summation order.  `u_packed` comes from tg_pack_conv3x3_wino (tg_conv3x3_wino_packed_floats(cin,
 * cout) floats; transposed = 0 for OIHW weights, 2 for the data gradient: channel roles swapped,
 * taps rotated by 180 degrees).  x2 / res / relu_mask as in tg_conv3x3_fwd / _masked (may be NULL). */
int64_t tg_conv3x3_wino_packed_floats(int cin, int cout);
/* 1 when the Winograd form is the faster one for this layer shape on an MI355X (at least 160 16-tile
 * workgroups, cout a multiple of 64, cin >= 16); the frame plan uses it to pick
 * the form of each layer whose tg_layer_weights.u is set. */
int tg_conv3x3_prefers_wino(int n, int cin, int cout, int h, int w);
int tg_pack_conv3x3_wino(const float* w, float* out, int cin, int cout, int transposed,
                         tg_stream_t stream);
int tg_conv3x3_wino_fwd(const float* x, int64_t x_nstride, int c1, const float* x2,
                        int64_t x2_nstride, const float* u_packed, const float* bias,
                        const float* res, int64_t res_nstride, const float* relu_mask,
                        int64_t mask_nstride, float* y, int64_t y_nstride, int n, int cin,
                        int cout, int h, int w, int act, tg_stream_t stream);

/* Several DEPENDENT 3x3 layers (layer i+1 reads what layer i writes: SRNet's conv_in and residual
 * blocks, tecogan_nets.py:108-116,141-143) in ONE launch.  Every separate launch of the Winograd
 * kernel pays ~10 us that nothing overlaps (launch, first loads, final stores); here the workgroups
 * of layer i+1 are dispatched behind those of layer i and start as soon as the 3x3 tile
 * neighbourhood they read has been written (per-tile flags, agent-scope loads / stores).  A
 * workgroup only waits for workgroups with a smaller block index; forward progress relies on the
 * dispatcher starting workgroups in block order (true of every CDNA part, not promised by HIP), so
 * the kernel is fail-safe: a poll limit turns a lost flag into a fault count instead of a hang.
 *   layers[i]: as tg_conv3x3_wino_fwd (x2 / bias / res may be NULL; res is added after the
 *              activation; y may alias res: in-place residual sum); cout <= 64 for every layer.
 *   flags:     tg_conv3x3_wino_chain_flag_ints(n_layers, n, h, w) int32, caller owned, zeroed ONCE
 *              before the first call; its last 16 ints are a fault counter the caller MUST read
 *              after synchronising (non-zero: a workgroup gave up waiting -- never seen; the
 *              results are then undefined.  The frame plan does this for its own chain, see
 *              tg_frnet_plan_chain_status).
 *   epoch:     any non-zero value not used before with these flags (a frame counter).
 * Buffers may be reused along the chain only in the patterns of the reference's SRNet: a layer may
 * overwrite a tensor that the PREVIOUS layer read, or its own residual input. */
typedef struct {
  const float* x; const float* x2; const float* u_packed; const float* bias; const float* res;
  float* y;
  int64_t x_nstride, x2_nstride, res_nstride, y_nstride;
  int c1, cin, act;
} tg_wino_layer;
int64_t tg_conv3x3_wino_chain_flag_ints(int n_layers, int n, int h, int w);
int tg_conv3x3_wino_chain(const tg_wino_layer* layers, int n_layers, int n, int cout, int h, int w,
                          int32_t* flags, int epoch, tg_stream_t stream);

/* The same DEPENDENT layers of ONE frame (n = 1) on persistent, LDS-RESIDENT workgroups (round 4;
 * tg_conv3x3_wino_res.hip): one workgroup per CU owns an 8 x 24 pixel block of the frame for ALL the
 * layers; the 64-channel block (+ a one-pixel ring) of the current and of the next layer live in LDS,
 * and between two layers only the block's outermost pixels travel through an exchange buffer in
 * global memory (16 KB per workgroup instead of 11 MB written + 11 MB read per layer at 134x320).
 * Same arithmetic in the same order as tg_conv3x3_wino_fwd: BIT-IDENTICAL results.
 *   supported: n == 1, cout == 64, even h and w, ceil(h/8) * ceil(w/24) <= CUs - 8 (every workgroup must
 *              be resident at once: neighbours wait for each other).  134x320 -> 238 workgroups.
 *   layers:    as tg_conv3x3_wino_chain with the buffer pattern of the reference's SRNet made a
 *              REQUIREMENT: layers[i].x == layers[i-1].y, layers[i].res in {NULL, layers[i-1].x},
 *              cin == 64 for i > 0, bias non-NULL.  Only layers[0].x / x2 are read and only
 *              layers[n_layers-1].y is written: the intermediate tensors never reach memory.
 *   workspace: tg_conv3x3_wino_resident_ws_bytes(h, w) bytes, 256-byte aligned, caller owned, zeroed ONCE
 *              before the first call; its last 256 bytes hold a fault counter (int32) the caller MUST
 *              read after synchronising (non-zero: a workgroup gave up waiting for a neighbour; the
 *              results are then undefined.  The frame plan does this for its own launch).
 *   epoch:     1, 2, 3, ... strictly increasing per call on one workspace. */
int tg_conv3x3_wino_resident_supported(int n, int cout, int h, int w);
int64_t tg_conv3x3_wino_resident_ws_bytes(int h, int w);
int tg_conv3x3_wino_resident(const tg_wino_layer* layers, int n_layers, int cout, int h, int w,
                             void* workspace, int epoch, tg_stream_t stream);
/* The same launch with SRNet's FIRST up-sampling layer (nn.ConvTranspose2d(64, 64, 3, 2, 1, output_padding=1) + ReLU,
 * tecogan_nets.py:119-126) as its tail: after the last conv layer's ring exchange every workgroup applies the
 * transposed convolution to its resident 8x24 block (+1 ring pixel right / below) and writes convt->y
 * (64, 2h, 2w); layers[n_layers-1].y is then NOT written.  Direct fp32 MFMA products (no Winograd): the result
 * equals tg_convt3x3s2_fwd on the same input up to the summation order (about 1e-6 relative).
 * u_packed: tg_conv3x3_wino_resident_ct_pack of the layer's (cin 64, cout 64, 3, 3) weights
 * (tg_conv3x3_wino_resident_ct_floats() floats).  convt == NULL: tg_conv3x3_wino_resident. */
typedef struct tg_wres_convt {
  const float* u_packed;
  const float* bias;
  float* y;
  int act;
} tg_wres_convt;
size_t tg_conv3x3_wino_resident_ct_floats(void);
int tg_conv3x3_wino_resident_ct_pack(const float* w_iohw, float* out, tg_stream_t stream);
int tg_conv3x3_wino_resident_ct(const tg_wino_layer* layers, int n_layers, int cout, int h, int w,
                                void* workspace, int epoch, const tg_wres_convt* convt, tg_stream_t stream);

/* Dependent 3x3 layers of SMALL frames (the training unroll: 2 x 32 x 32 / 2 x 64 x 64 LR pixels per
 * frame, tecogan_nets.py:174-225) in ONE launch: one persistent workgroup per (image row, 32-pixel
 * segment[, 32-channel half]) walks all the layers and exchanges halo rows with its neighbours
 * through agent-scope memory + per-tile flags (tg_conv3x3_chain.hip).  Direct fp32-MFMA form.  The
 * launcher picks the workgroups per tile from the shape (tg_conv3x3_chain_supported) and the weights must be
 * packed for that choice -- pack_layout 64: tg_conv3x3_pack with ocb = 64 (transposed = 0 forward, 2 data
 * gradient) for 1 or 2 workgroups per tile (32 x 32 x 2 MFMAs); pack_layout 16: tg_conv3x3_pack16
 * (tg_conv3x3_pack16_floats() floats per layer) for 4 workgroups per tile (16 x 16 x 4 MFMAs, the
 * smallest frames); a mismatch is TG_E_ARG.
 *   layers[i]: x (+ x2: channels [c1, cin)) -> y = relu_mask > 0 ? act(conv + bias) + res : 0;
 *              cin, cout <= 64; bias / res / relu_mask / x2 may be NULL; buffers may be reused along the
 *              chain in SRNet's patterns only (ping-pong, in-place residual sum).
 *   flags:     tg_conv3x3_chain_flag_ints(n_layers, n, h, w) int32, caller owned, zeroed ONCE.
 *   err:       int32 fault counter, device OR pinned host memory (the kernel adds with system scope):
 *              a workgroup that exhausts poll_limit polls (~32 cycles each; < 0: at once = fault
 *              injection) counts a fault and carries on -- the launch always ends; a non-zero count
 *              after synchronisation means the output is undefined and the caller must fall back to
 *              one launch per layer.
 *   epoch:     non-zero, different from the previous call on these flags.
 * Every workgroup must be resident at once: tg_conv3x3_chain_supported returns 0 when the grid would
 * exceed half of what the device holds (the call then fails with TG_E_SHAPE), else the number of
 * workgroups per tile it will use (1, 2 or 4). */
typedef struct {
  const float* x; const float* x2; const float* w_packed; const float* bias; const float* res;
  const float* relu_mask;
  float* y;
  int64_t x_nstride, x2_nstride, res_nstride, mask_nstride, y_nstride;
  int c1, cin, cout, act;
} tg_chain_layer;
int64_t tg_conv3x3_chain_flag_ints(int n_layers, int n, int h, int w);
int tg_conv3x3_chain_supported(int n, int h, int w, int cmax);
size_t tg_conv3x3_pack16_floats(void);
int tg_conv3x3_pack16(const float* w, float* w_packed, int cin, int cout, int transposed, tg_stream_t stream);
/* All the layers of a chain packed by one launch.  Item: source weights (O, i_total, 3, 3), of which the
 * input-channel slice [i_off, i_off + I) is used; transposed 0: the layer (cin = I, cout = O), 2: its data
 * gradient (cin = O, cout = I); out: tg_conv3x3_chain_packed_floats(pack_layout, cin) floats, in the layout
 * of tg_conv3x3_pack16 (16) / tg_conv3x3_pack with ocb 64 (64). */
typedef struct tg_pack_item {
  const float* w;
  float* out;
  int cin, cout, transposed, i_total, i_off;
} tg_pack_item;
size_t tg_conv3x3_chain_packed_floats(int pack_layout, int cin);
int tg_conv3x3_chain_pack(const tg_pack_item* items, int n_items, int pack_layout, tg_stream_t stream);
int tg_conv3x3_chain(const tg_chain_layer* layers, int n_layers, int n, int h, int w, int pack_layout,
                     int32_t* flags, int32_t* err, uint32_t epoch, int poll_limit, tg_stream_t stream);
/* SRNet's conv_in + nb residual blocks on one training frame (tecogan_nets.py:108-116, :141-143) and
 * the matching reverse sweep, each as one tg_conv3x3_chain launch.  acts / dz: 1 + 2*nb tensors
 * (n, nf, h, w) back to back, acts[0] = conv_in's output, acts[1+2b] / acts[2+2b] = block b's inner
 * activation / output; dz[i] = gradient w.r.t. the pre-activation of layer i -- dz[2nb], the gradient of
 * the body's output, is the INPUT the caller fills before the call -- d_tran = gradient w.r.t. the
 * warped-frame input channels (n, c_tran, h, w).
 * layers[i] / dgrad[i]: forward / data-gradient packs of layer i (dgrad[0]: conv_in restricted to
 * input channels [c_lr, c_lr + c_tran), bias unused). */
typedef struct { const float* w; const float* b; } tg_packed_layer;
int tg_srnet_body_fwd(const tg_packed_layer* layers, int pack_layout, int nb, const float* lr, int c_lr,
                      const float* tran, int c_tran, float* acts, int n, int nf, int h, int w, int32_t* flags,
                      int32_t* err, uint32_t epoch, int poll_limit, tg_stream_t stream);
int tg_srnet_body_bwd(const tg_packed_layer* dgrad, int pack_layout, int nb, const float* acts, float* dz,
                      float* d_tran, int c_tran, int n, int nf, int h, int w, int32_t* flags, int32_t* err,
                      uint32_t epoch, int poll_limit, tg_stream_t stream);

/* Weight / bias gradients of the chained body's layers for ALL unrolled frames in one launch each.
 * dz_bases[f] / act_bases[f]: the dz / acts blocks of frame f (tg_srnet_body_bwd / _fwd),
 * layer_stride = n_per_frame * c * h * w.
 *   tg_wgrad3x3_body: grads[L - 1] (+)= dW of layer L = 1 .. nlayers (= 2 * nb; conv_in, whose input has
 *     other channel counts, goes through tg_wgrad3x3_multi); workspace: tg_wgrad3x3_body_workspace_floats.
 *   tg_bias_grad_body: dbs[L] += db of layer L = 0 .. nlayers - 1 (= 1 + 2 * nb layers, conv_in included). */
size_t tg_wgrad3x3_body_workspace_floats(int nframes, int n_per_frame, int nlayers, int c, int h, int w);
int tg_wgrad3x3_body(const float* const* dz_bases, const float* const* act_bases, int nframes, int64_t layer_stride, int nlayers, float* const* grads, float* workspace,
                     int n_per_frame, int c, int h, int w, int accumulate, tg_stream_t stream);
/* ... and their bias gradients in the same pass: dbs[L - 1] (c floats) (+)= sum of dZ of layer L = 1..nlayers. */
int tg_wgrad3x3_body_bias(const float* const* dz_bases, const float* const* act_bases, int nframes,
                          int64_t layer_stride, int nlayers, float* const* grads, float* const* dbs,
                          float* workspace, int n_per_frame, int c, int h, int w, int accumulate,
                          tg_stream_t stream);
int tg_bias_grad_body(const float* const* dz_bases, int nframes,
                      int64_t layer_stride, int nlayers, float* const* dbs, int n_per_frame, int c, int hw,
                      tg_stream_t stream);

/* Split-K variant for layers whose output tile count cannot fill the GPU (FNet's
 * low-resolution many-channel middle, tecogan_nets.py:37-60): `ksplit` groups of
 * input channels are reduced by different workgroups into `partials`
 * (ksplit * n*cout*h*w floats, caller owned), then a finalize pass adds them in a
 * fixed order (bit-reproducible) and applies bias + activation and, when
 * pool != 0, the following nn.MaxPool2d(2,2) (y is then (n,cout,h/2,w/2)).
 * tg_conv3x3_pick_ksplit returns the recommended factor (1 = use tg_conv3x3_fwd). */
int tg_conv3x3_pick_ksplit(int n, int cin, int cout, int h, int w);
int tg_conv3x3_splitk_fwd(const float* x, int64_t x_nstride, int c1, const float* x2,
                          int64_t x2_nstride, const float* w_packed, int ocb,
                          const float* bias, float* y, int n, int cin, int cout, int h,
                          int w, int act, int ksplit, float* partials, int pool,
                          tg_stream_t stream);

/* ------------------------------------------------------------------------
 * ConvTranspose2d(cin, cout, 3, stride 2, padding 1, output_padding 1) + bias
 * + activation, as 4 sub-pixel phase GEMMs (no zero MACs), fp32 MFMA.
 * Replaces SRNet.conv_up, tecogan_nets.py:119-126.   y is (n, cout, 2h, 2w).
 * Weights packed with tg_conv3x3_pack(..., ocb=64, transposed=1).
 * ---------------------------------------------------------------------- */
int tg_convt3x3s2_fwd(const float* x, int64_t x_nstride, const float* w_packed,
                      const float* bias, float* y, int64_t y_nstride, int n,
                      int cin, int cout, int h, int w, int act,
                      tg_stream_t stream);
/* The HR stage without the 64-channel HR tensor (inference; SRNet.forward tecogan_nets.py:119-131,
 * 145): the LAST ConvTranspose2d + ReLU, run in "Z mode", contracts its output over the channels
 * with conv_out's weights while the values are still in the MFMA accumulators and stores the
 * 9*cz tap planes  z[tap*cz + o] = sum_oc Wout[o][oc][tap] * relu(convT(x)[oc] + b[oc])
 * (cz = out_nc <= 3; z is (n, 32, 2h, 2w), the first 9*cz planes written); tg_convout_tail
 * shift-adds them (zero padding of conv_out), adds conv_out's bias and upsample_func(lr_curr),
 * and optionally emits the uint8 HWC frame.  Same result as tg_convt3x3s2_fwd +
 * tg_conv3x3_small_fwd(_u8) up to summation order; 74 MB written + read instead of 176 MB.
 * tg_convt_pack_wz packs conv_out's OIHW weights (cz, nf, 3, 3) into the 2048-float operand. */
int tg_convt_pack_wz(const float* w_out_oihw, float* wz, int cz, int nf, tg_stream_t stream);
int tg_convt3x3s2_z_fwd(const float* x, int64_t x_nstride, const float* w_packed,
                        const float* bias, const float* wz, int cz, float* z,
                        int64_t z_nstride, int n, int cin, int cout, int h, int w, int act,
                        tg_stream_t stream);
/* The same with the kernel form chosen by the caller: -1 = the library's rule (what tg_convt3x3s2_z_fwd does: a tiled
 * form, 0 or 3), 0 = the tiled form (one workgroup per 4-row x 32-pixel tile, weights staged through LDS chunk by chunk),
 * 1 / 2 = the streaming form of round 6 (weights LDS-resident for the whole launch, every wave an autonomous worker
 * on (row, 32 pixels, row parity) items; 1: items handed out in batches from a device-wide counter, polls bounded,
 * a time-out is reported as TG_E_HIP by the NEXT call and turns the form off; 2: a static, per-SIMD balanced item
 * list; needs cin, cout <= 64), 3 = tiled with a split tail (whole rounds of four-row workgroups, the remaining rows as
 * two-row workgroups in a second launch; what the rule picks for one 268x640-class frame).  All forms produce
 * BIT-IDENTICAL planes (same taps in the same order per phase).
 * Measured at 268x640 (MI355X): tiled 148 us, streaming static 143.5 us, streaming dynamic 164 us; through the
 * frame +-0, hence the rule (EXPERIMENTS.md, round 6). */
int tg_convt3x3s2_z_fwd_form(const float* x, int64_t x_nstride, const float* w_packed,
                             const float* bias, const float* wz, int cz, float* z,
                             int64_t z_nstride, int n, int cin, int cout, int h, int w, int act,
                             int form, tg_stream_t stream);
int tg_convout_tail(const float* z, int64_t z_nstride, int cz, const float* bias,
                    const float* up_src, int up_mode, int up_scale, float* y,
                    int64_t y_nstride, uint8_t* u8_out, int n, int h, int w,
                    tg_stream_t stream);
/* The same with the kernel form chosen by the caller: -1 the library's rule (four pixels per thread wherever w % 4 == 0
 * and the planes are 16-byte aligned), 0 one HR pixel per thread, 1 four (TG_E_SHAPE if the shape does not allow it).
 * Both forms accumulate every output in the same order: bit-identical (tests/test_hip_parity.py). */
int tg_convout_tail_form(const float* z, int64_t z_nstride, int cz, const float* bias,
                         const float* up_src, int up_mode, int up_scale, float* y,
                         int64_t y_nstride, uint8_t* u8_out, int n, int h, int w, int form,
                         tg_stream_t stream);

/* ------------------------------------------------------------------------
 * 3x3 conv with a tiny output-channel count (cout <= 4), direct fp32 VALU
 * kernel (an MFMA tile would be >90 % padding).  Replaces FNet.flow[2]
 * (32->2, +tanh*24, tecogan_nets.py:65,80) and SRNet.conv_out (64->3,
 * tecogan_nets.py:131) with the `out += upsample_func(lr_curr)` residual
 * (:145) fused: up_mode/up_scale describe how `up_src` (n, cout, h/up_scale,
 * w/up_scale) is up-sampled and added.  w_oihw is the plain (cout,cin,3,3).
 * ---------------------------------------------------------------------- */
int tg_conv3x3_small_fwd(const float* x, int64_t x_nstride, const float* w_oihw,
                         const float* bias, const float* up_src, int up_mode,
                         int up_scale, float* y, int64_t y_nstride, int n,
                         int cin, int cout, int h, int w, int act,
                         tg_stream_t stream);
/* The same launch additionally writing the result as an (h, w, cout) uint8 frame --
 * float32_to_uint8 (codes/utils/data_utils.py:80-87: round-half-even, clip) of the fp32 value
 * it stores in y -- so FRNet.infer_sequence's per-frame quantise pass and its re-read of the
 * HR frame disappear.  Needs n == 1, w % 4 == 0 and 16-byte aligned planes:
 * tg_conv3x3_small_can_fuse_u8 tells; otherwise call tg_quantize_u8_hwc on y. */
int tg_conv3x3_small_fwd_u8(const float* x, int64_t x_nstride, const float* w_oihw,
                            const float* bias, const float* up_src, int up_mode,
                            int up_scale, float* y, int64_t y_nstride, uint8_t* u8_out,
                            int n, int cin, int cout, int h, int w, int act,
                            tg_stream_t stream);
int tg_conv3x3_small_can_fuse_u8(const float* x, int64_t x_nstride, const float* y,
                                 int64_t y_nstride, int n, int cin, int h, int w);
/* y = act(conv3x3(x) + bias) + res with an explicit residual tensor (n, cout, h, w) -- the training step
 * computes the bicubic frame of `out += upsample_func(lr_curr)` (tecogan_nets.py:145) once per step instead
 * of 16 taps per pixel in every epilogue.  w % 4 == 0, 16-byte aligned planes.  (Launches of a few
 * workgroups -- the training frames -- run as 4-row tiles with the waves of a workgroup splitting the input
 * channels, with or without a residual: tg_conv3x3_small_fwd picks that form by itself.) */
/* The mirror image: cin <= 4, any number of output channels, no bias / activation, optional ReLU mask
 * (y = relu_mask > 0 ? conv : 0) -- the data gradient of a cout <= 4 head (conv_out 64 -> 3, the flow head
 * 32 -> 2): w_oihw = (cout, cin, 3, 3) of THIS op, i.e. the layer's weights with the channel roles
 * swapped and the taps rotated by 180 degrees.  w % 4 == 0, 16-byte aligned planes. */
int tg_conv3x3_fewin_fwd(const float* x, int64_t x_nstride, const float* w_oihw, const float* relu_mask,
                         int64_t mask_nstride, float* y, int64_t y_nstride, int n, int cin, int cout,
                         int h, int w, tg_stream_t stream);
int tg_conv3x3_small_fwd_res(const float* x, int64_t x_nstride, const float* w_oihw, const float* bias,
                             const float* res, int64_t res_nstride, float* y, int64_t y_nstride, int n,
                             int cin, int cout, int h, int w, int act, tg_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused: reflect-pad(bottom/right) -> scale * upsample -> backward_warp ->
 * space_to_depth.   Replaces tecogan_nets.py:238-250 (step) / :189,203,208
 * (forward_sequence):
 *   lr_flow (n, 2, fh, fw) with fh = h/8*8, fw = w/8*8 (FNet output);
 *   hr_prev (n, c, s*h, s*w);
 *   out[n, (sy*s+sx)*c + ch, oy, ox] = warp(hr_prev, s*up(pad(lr_flow)))[n, ch, oy*s+sy, ox*s+sx]
 * written at out + n*out_nstride (so it can land in the channel slice of the
 * SRNet input buffer).  hr_flow_out (n,2,s*h,s*w) may be NULL; when given the
 * up-sampled flow is also stored (training needs it for D and for backward).
 * ---------------------------------------------------------------------- */
int tg_flowup_warp_s2d_fwd(const float* lr_flow, int fh, int fw,
                           const float* hr_prev, float* out,
                           int64_t out_nstride, float* hr_flow_out, int n,
                           int c, int h, int w, int scale, int up_mode,
                           tg_stream_t stream);
/* Measurement aid (bench.py roofline_warp*.copy_ceiling; no counterpart in the reference): a float4 grid-stride copy of
 * `bytes` (read once, written once) launched with the given grid -- the fused warp kernel's own -- so that its HBM
 * fraction can also be read against what a plain copy of the same size reaches on this part. */
int tg_copy_ceiling(const void* src, void* dst, int64_t bytes, int blocks, int threads, tg_stream_t stream);

/* backward_warp(x, flow): codes/utils/net_utils.py:50-82 (grid_sample bilinear,
 * border, align_corners=True; flow ch0 = x, ch1 = y, in pixels). */
int tg_backward_warp_fwd(const float* x, const float* flow, float* y, int n,
                         int c, int h, int w, tg_stream_t stream);

/* space_to_depth(x, s): net_utils.py:36-47.  x (n,c,h,w) -> y (n,s*s*c,h/s,w/s) */
int tg_space_to_depth(const float* x, float* y, int64_t y_nstride, int n, int c,
                      int h, int w, int scale, tg_stream_t stream);

/* upsample_func: BicubicUpsampler.forward net_utils.py:133-156 /
 * F.interpolate bilinear :86-89.  x (nc,h,w) -> y (nc, s*h, s*w), y = mul * up(x). */
int tg_upsample_fwd(const float* x, float* y, int nc, int h, int w, int scale,
                    int up_mode, float mul, tg_stream_t stream);

/* nn.MaxPool2d(2,2) floor mode: tecogan_nets.py:28,35,42.  y (nc, h/2, w/2) */
int tg_maxpool2_fwd(const float* x, float* y, int nc, int h, int w,
                    tg_stream_t stream);

/* float32_to_uint8 + CHW->HWC: codes/utils/data_utils.py:80-87 and the
 * transpose at tecogan_nets.py:281.  x (c,h,w) fp32 -> y (h,w,c) uint8,
 * y = uint8(clip(rint(x*255), 0, 255)), rint = round-half-even. */
int tg_quantize_u8_hwc(const float* x, uint8_t* y, int c, int h, int w,
                       tg_stream_t stream);

/* uint8 HWC frames -> fp32 CHW in [0,1], the `gt.permute(0,3,1,2).float() / 255.0` of
 * BaseModel.prepare_inference_data (codes/models/base_model.py:112).
 * x (n,h,w,c) uint8 -> y (n,c,h,w) fp32. */
int tg_dequantize_u8_hwc(const uint8_t* x, float* y, int n, int c, int h, int w,
                         tg_stream_t stream);

/* MetricCalculator.compute_PSNR (codes/metrics/metric_calculator.py:228-244) on uint8 HWC
 * frames resident on the device: sse[f] = sum of squared differences of frame f, exact
 * (integers), on the Y channel of rgb_to_ycbcr (codes/utils/data_utils.py:56-77) when
 * y_only, else over the three RGB channels.  PSNR = 20 log10(255 / sqrt(sse / count)). */
int tg_psnr_sse_u8(const uint8_t* true_hwc, const uint8_t* pred_hwc, uint64_t* sse,
                   int frames, int h, int w, int y_only, tg_stream_t stream);
/* Y channel of rgb_to_ycbcr for n RGB triples (n,3) uint8 -> (n) uint8; exposed so the
 * conversion can be checked exhaustively against numpy. */
int tg_luma_u8(const uint8_t* rgb, uint8_t* y, int64_t n, tg_stream_t stream);

/* ========================================================================
 * Training side: backward kernels.
 * Conv data gradients reuse tg_conv3x3_fwd with weights packed by
 * tg_conv3x3_pack(..., transposed = 2) (rot180 + in/out swap).
 * ====================================================================== */

/* dW of Conv2d(cin,cout,3,1,1):  G[a][b][tap] (+)= sum_{n,y,x} p[n][a][y][x] * q[n][b][y+ky-1][x+kx-1]
 * with p = dZ (a = cout), q = X (b = cin); fp32 MFMA, deterministic split reduction.
 * `grad` is the (ca, cb_total, 3, 3) gradient tensor; columns [cb_off, cb_off+cb) are
 * written (two-source convs call it once per source).  workspace:
 * tg_wgrad3x3_workspace_floats(...) floats, caller owned. */
size_t tg_wgrad3x3_workspace_floats(int n, int ca, int cb_total, int h, int w);
int tg_wgrad3x3(const float* p, int64_t p_nstride, const float* q, int64_t q_nstride,
                float* grad, float* workspace, int n, int ca, int cb, int cb_total,
                int cb_off, int h, int w, int accumulate, tg_stream_t stream);

/* tg_wgrad3x3 / tg_bias_grad over a batch that lives in `nseg` (<= 64) separately allocated
 * segments of `n_per_seg` images each -- the per-frame tensors of a layer that is applied
 * once per unrolled frame (FRNet.forward_sequence, tecogan_nets.py:174-225).  p_list / q_list /
 * dy_list are HOST arrays of device pointers; they are read during the call only. */
int tg_wgrad3x3_multi(const float* const* p_list, const float* const* q_list, int nseg,
                      int64_t p_nstride, int64_t q_nstride, float* grad, float* workspace,
                      int n_per_seg, int ca, int cb, int cb_total, int cb_off, int h, int w,
                      int accumulate, tg_stream_t stream);
/* tg_wgrad3x3_multi that also delivers the layer's bias gradient: bias_grad (ca floats) (+)= sum of p over
 * images and pixels, taken from the p values the kernel stages anyway (vector staging: w % 4 == 0, aligned
 * planes; other forms run tg_bias_grad_multi themselves) -- no second pass over dZ. */
int tg_wgrad3x3_multi_bias(const float* const* p_list, const float* const* q_list, int nseg,
                           int64_t p_nstride, int64_t q_nstride, float* grad, float* bias_grad,
                           float* workspace, int n_per_seg, int ca, int cb, int cb_total, int cb_off,
                           int h, int w, int accumulate, tg_stream_t stream);
/* tg_wgrad3x3_multi for a space-to-depth embedded strided conv (see tg_conv3x3_fwd_phased): the
 * q channels come in 4 sub-pixel phases of `cphase` (multiple of 64) channels and only the taps
 * the phase owns are computed; the other entries of grad are written as 0. */
int tg_wgrad3x3_multi_phased(const float* const* p_list, const float* const* q_list, int nseg,
                             int64_t p_nstride, int64_t q_nstride, float* grad,
                             float* workspace, int n_per_seg, int ca, int cb, int h, int w,
                             int accumulate, int cphase, int taps_phase0, int taps_phase1,
                             tg_stream_t stream);
/* dW of nn.ConvTranspose2d(ci, co, 3, 2, 1, output_padding = 1) (tecogan_nets.py:119-126) straight from
 * the gradient of its output:
 *   grad[a][b][ky][kx] (+)= sum x[n][a][y][x] * dz[n][b][2y - 1 + ky][2x - 1 + kx]
 * x_list[i]: (n_per_seg, ci, h, w) inputs of the layer, dz_list[i]: (n_per_seg, co, 2h, 2w); grad in the
 * layer's own (ci, co, 3, 3) layout.  No space-to-depth copy of dZ, all nine taps in one balanced pass
 * (the phased form above spends 1 / 2 / 2 / 4 taps on its four channel blocks).  bias_grad (co floats, may be
 * NULL): the layer's bias gradient (+)= sum of dZ over images and pixels, taken from the dZ values the kernel
 * stages anyway (no second pass over the HR tensors: 637 MB per step at crop 256).  Workspace:
 * tg_wgrad3x3_convt_workspace_floats(total images, ci, co, h, w). */
size_t tg_wgrad3x3_convt_workspace_floats(int n, int ci, int co, int h, int w);
int tg_wgrad3x3_convt_multi(const float* const* x_list, const float* const* dz_list, int nseg, float* grad,
                            float* bias_grad, float* workspace, int n_per_seg, int ci, int co, int h, int w,
                            int accumulate, tg_stream_t stream);
int tg_bias_grad_multi(const float* const* dy_list, int nseg, float* db, int n_per_seg, int c,
                       int hw, int accumulate, tg_stream_t stream);
/* dx = dy * act'(.), expressed through the activation OUTPUT y (ReLU, LeakyReLU(0.2),
 * tanh*24).  dx may alias dy. */
int tg_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act,
               tg_stream_t stream);
/* db[c] (+)= sum_{n,h,w} dy */
int tg_bias_grad(const float* dy, float* db, int n, int c, int hw, int accumulate,
                 tg_stream_t stream);
/* MaxPool2d(2,2) backward; x is the pool INPUT (nc,h,w), dy (nc,h/2,w/2) */
int tg_maxpool2_bwd(const float* x, const float* dy, float* dx, int nc, int h, int w,
                    tg_stream_t stream);
/* transpose of tg_upsample_fwd: dx (nc,h,w) = up^T(mul * dy) */
int tg_upsample_bwd(const float* dy, float* dx, int nc, int h, int w, int scale,
                    int up_mode, float mul, tg_stream_t stream);
/* backward_warp backward (grid_sample bilinear/border/align_corners=True autograd):
 * dimg (n,c,h,w) and/or dflow (n,2,h,w); either may be NULL. */
int tg_backward_warp_bwd(const float* x, const float* flow, const float* dy, float* dimg,
                         float* dflow, int n, int c, int h, int w, tg_stream_t stream);
/* the same, but the image gradient is ADDED to what dimg_acc already holds (no zeroing, no separate
 * accumulation pass: the frame a warp reads usually has a gradient of its own loss term already). */
int tg_backward_warp_bwd_acc(const float* x, const float* flow, const float* dy, float* dimg_acc, float* dflow,
                             int n, int c, int h, int w, tg_stream_t stream);
/* The training unroll's pair backward_warp (net_utils.py:50-82) -> space_to_depth (net_utils.py:36-47;
 * tecogan_nets.py:208-212) as ONE launch each way: y (n, scale^2 c, h/scale, w/scale) =
 * space_to_depth(backward_warp(x, flow), scale), bit-identical to the two separate calls; backward takes the
 * gradient in that layout (dy_s2d), accumulate != 0 adds the image gradient to dimg instead of overwriting it. */
int tg_backward_warp_s2d_fwd(const float* x, const float* flow, float* y, int n, int c, int h, int w, int scale,
                             tg_stream_t stream);
int tg_backward_warp_s2d_bwd(const float* x, const float* flow, const float* dy_s2d, float* dimg, int accumulate,
                             float* dflow, int n, int c, int h, int w, int scale, tg_stream_t stream);
/* inverse of tg_space_to_depth: x (n, s*s*c, h, w) -> y (n, c, s*h, s*w) */
int tg_depth_to_space(const float* x, float* y, int n, int c, int h, int w, int scale,
                      tg_stream_t stream);
/* tg_depth_to_space whose result is the gradient of an activation OUTPUT act_y (same shape as y): the
 * result is multiplied by act'(.) on the way out (relu | lrelu 0.2, through the output as tg_act_bwd) -- one pass
 * instead of depth_to_space + act_bwd.  scale 2 | 4, w % 4 == 0, 16-byte aligned tensors
 * (tg_depth_to_space_act_bwd_supported). */
int tg_depth_to_space_act_bwd_supported(const float* x, const float* act_y, const float* y, int w, int scale);
int tg_depth_to_space_act_bwd(const float* x, const float* act_y, int act, float* y, int n, int c, int h,
                              int w, int scale, tg_stream_t stream);
/* CharbonnierLoss (optim/losses.py:31-50): *loss_accum += loss_scale * sum sqrt(d^2+eps),
 * dx = grad_scale * d / sqrt(d^2+eps)  (d = x - y); loss_accum or dx may be NULL. */
int tg_charbonnier(const float* x, const float* y, int64_t n, float eps, float loss_scale,
                   float* loss_accum, float grad_scale, float* dx, tg_stream_t stream);
/* VGGFeatureExtractor input normalisation (codes/models/networks/vgg_nets.py:29):
 * y = (x - mean[c]) / std[c];  mean == NULL means 0 (the op's backward: dx = dy / std[c]). */
int tg_channel_norm(const float* x, const float* mean, const float* std, float* y, int n,
                    int c, int64_t hw, tg_stream_t stream);
/* CosineSimilarityLoss (codes/models/optim/losses.py:53-62; F.cosine_similarity dim=1) of two
 * (n,c,h,w) feature maps, as used for the perceptual loss at vsrgan_model.py:226-241:
 * *loss_accum += loss_scale * sum_pixels(1 - cos);  da = grad_scale * d(sum_pixels(1 - cos))/da.
 * Either output may be NULL. */
int tg_cosine_loss(const float* a, const float* b, int n, int c, int64_t hw, float eps,
                   float loss_scale, float* loss_accum, float grad_scale, float* da,
                   tg_stream_t stream);
/* nn.L1Loss / nn.MSELoss selected by define_criterion (codes/models/optim/__init__.py:10-14):
 * *loss_accum += loss_scale * sum(v), dx = grad_scale * dv/dx, v = |x-y| or (x-y)^2. */
#define TG_LOSS_L1 1
#define TG_LOSS_MSE 2
int tg_pixel_loss(const float* x, const float* y, int64_t n, int mode, float loss_scale,
                  float* loss_accum, float grad_scale, float* dx, tg_stream_t stream);
/* VanillaGANLoss (optim/losses.py:6-14) against a constant target, plus the statistics
 * VSRGANModel.train logs (vsrgan_model.py:163-164,194-195):
 * stats3[0] += scale*sum(bce), [1] += scale*sum(x), [2] += scale*sum(log(sigmoid(x)+1e-8));
 * dx = grad_scale * (sigmoid(x) - target). */
int tg_bce_logits(const float* x, int64_t n, float target, float scale, float* stats3,
                  float grad_scale, float* dx, tg_stream_t stream);
/* LSGANLoss (optim/losses.py:17-28): MSE against the constant 1 / 0 target; same statistics layout
 * (stats3[0] += scale*sum((x-t)^2), [1], [2] as above), dx = grad_scale * 2 (x - target). */
int tg_lsgan_loss(const float* x, int64_t n, float target, float scale, float* stats3,
                  float grad_scale, float* dx, tg_stream_t stream);
/* torch.optim.Adam step (vsrgan_model.py:76-87), in place, `step` = 1-based count */
int tg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int step,
                 tg_stream_t stream);
/* The same step behind a device-side guard: a no-op (weights and both moments untouched) when
 * *skip_if_nonzero != 0.  The training step points it at the FAULT SLOT of the network's flat gradient
 * buffer: tg_fault_to_slot adds 1 to that float when the pinned fault counter of the chained launches
 * (tg_srnet_body_fwd / _bwd) is non-zero, BEFORE the data-parallel all-reduce of the buffer -- so a fault
 * on any rank drops the update on every rank, instead of applying gradients built on stale tiles
 * (the reference has no counterpart: its layers are separate launches). */
int tg_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int step,
                         const float* skip_if_nonzero, tg_stream_t stream);
int tg_fault_to_slot(const int32_t* fault_counter, float* slot, tg_stream_t stream);
int tg_axpy(float* y, const float* x, float a, int64_t n, tg_stream_t stream);
/* y[i] = x[i] / d (y may alias x): the mean of an all-reduced gradient bucket, with the IEEE division
 * DDP's `bucket / world_size` performs (base_model.py:130-136), exact for any world size. */
int tg_div_scalar(float* y, const float* x, float d, int64_t n, tg_stream_t stream);
/* BatchNorm2d (train mode, batch statistics, running stats updated with the unbiased
 * variance) + LeakyReLU(slope), tecogan_nets.py:322-340; and its backward. */
int tg_bn_lrelu_train_fwd(const float* x, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum,
                          float eps, float slope, float* y, float* save_mean,
                          float* save_invstd, int n, int c, int hw, tg_stream_t stream);
int tg_bn_lrelu_train_bwd(const float* x, const float* y, const float* dy,
                          const float* gamma, const float* save_mean,
                          const float* save_invstd, float slope, float* dx,
                          float* dgamma, float* dbeta, int accumulate, float* scratch2c,
                          int n, int c, int hw, tg_stream_t stream);
/* SyncBatchNorm halves (base_model.py:133): per-channel reductions exposed so the host can
 * exchange the packed vectors over RCCL in between.
 *   forward : tg_bn_local_stats -> stats2c = [mean | centred sum of squares] of THIS rank's
 *             slice; all-gather (world x 2c floats); tg_bn_merge_stats merges equally sized
 *             partitions with Chan's formula (no E[x^2]-mean^2 cancellation; world = 1
 *             reproduces tg_bn_lrelu_train_fwd's statistics exactly) and updates the
 *             running stats with the unbiased global variance;
 *   backward: tg_bn_lrelu_bwd_reduce -> sums2c = [sum dz | sum dz*xhat]; all-reduce(sum);
 *             tg_bn_lrelu_bwd_apply with inv_count = 1 / GLOBAL element count. */
int tg_bn_local_stats(const float* x, float* stats2c, int n, int c, int hw, tg_stream_t stream);
int tg_bn_merge_stats(const float* gathered, int world, float count_per_rank, float eps,
                      float momentum, float* mean, float* invstd, float* running_mean,
                      float* running_var, int c, tg_stream_t stream);
int tg_bn_lrelu_apply(const float* x, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, float slope, float* y,
                      int n, int c, int hw, tg_stream_t stream);
int tg_bn_lrelu_bwd_reduce(const float* x, const float* y, const float* dy,
                           const float* mean, const float* invstd, float slope,
                           float* sums2c, int n, int c, int hw, tg_stream_t stream);
int tg_bn_lrelu_bwd_apply(const float* x, const float* y, const float* dy,
                          const float* mean, const float* invstd, const float* gamma,
                          const float* sums2c, float slope, float inv_count, float* dx,
                          int n, int c, int hw, tg_stream_t stream);
/* nn.Linear(k, 1) (tecogan_nets.py:375): y[r] = x[r,:].w + b, and backward */
int tg_linear1_fwd(const float* x, const float* w, const float* b, float* y, int rows,
                   int k, tg_stream_t stream);
int tg_linear1_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                   float* db, int rows, int k, int accumulate, tg_stream_t stream);

/* downsample_bd (codes/utils/data_utils.py:30-53): per-plane 2-D Gaussian (kernel2d:
 * ksize*ksize floats on the device, create_kernel :11-27) + stride-`scale` decimation.
 * pad = 0: valid conv (training);  pad = 1: reflect padding (testing).
 * y is (nc, oh, ow) with oh = pad ? (h-1)/scale+1 : (h-ksize)/scale+1. */
int tg_downsample_bd(const float* x, const float* kernel2d, float* y, int nc, int h,
                     int w, int ksize, int scale, int pad, tg_stream_t stream);

/* ------------------------------------------------------------------------
 * Data movement of the training step, one launch each (they replace chains of ATen slice / flip /
 * cat / zero-fill kernels):
 *   tg_time_gather     y[n][k] = x[n][idx[k]] over (n, t, inner) tensors, idx a HOST array of k <= 64
 *                      frame numbers (a negative entry writes a zero frame): the ping-pong augmentation cat(x, x.flip(1)[:, 1:])
 *                      (vsrgan_model.py:112-119), the halves hr[:, :te-1] / hr[:, te:].flip(1) of the
 *                      ping-pong loss (:246-247), data[:, :t] (tecogan_nets.py:437);
 *   tg_transpose01     y (b, a, inner) = x (a, b, inner): clip-major (n, t, ...) <-> frame-major (t, n, ...),
 *                      so that the recurrent unroll (tecogan_nets.py:196-214) reads / writes each
 *                      time step as a contiguous slice instead of copying it out;
 *   tg_stack_time      y (n, k, inner) with y[:, j] = src_host[j] (n, inner): torch.stack(frames, 1) of
 *                      the k <= 64 per-frame outputs (tecogan_nets.py:216), src_host a HOST array of
 *                      device pointers;
 *   tg_index_gather    out[i] (+)= src[idx[i]], 0 where idx[i] is outside [0, n_src): the weight
 *                      re-layouts that express ConvTranspose2d(k3,s2) backward and Conv2d(k4,s2) as
 *                      3x3 convolutions over space_to_depth tensors, and their inverse on the gradient;
 *   tg_pingpong_grad   the ping-pong loss gradient g (n, te-1, inner) routed onto the 2*te-1 frames:
 *                      +g | 0 | -flip(g);
 *   tg_d_assemble_fwd  SpatioTemporalDiscriminator's input (tecogan_nets.py:440-463): x (n*t/3, 9c, h, w)
 *                      = [frame triplets rrrgggbbb | warped triplets centre-cropped (crop) and
 *                      zero-padded (pad) | bicubic condition]; data / cond are (n, t_data|t_cond, c, h, w)
 *                      of which the first t frames are used, warped is (n*t, c, h, w);
 *   tg_d_assemble_bwd  its adjoint: g -> g_data (n, t_data, c, h, w; zero beyond t), g_warped (n*t, c, h, w).
 * ---------------------------------------------------------------------- */
int tg_time_gather(const float* x, float* y, const int* idx_host, int n, int t_in, int k,
                   int64_t inner, tg_stream_t stream);
int tg_transpose01(const float* x, float* y, int a, int b, int64_t inner, tg_stream_t stream);
int tg_stack_time(const float* const* src_host, int k, float* y, int n, int64_t inner,
                  tg_stream_t stream);
int tg_index_gather(const float* src, const int64_t* idx, float* out, int64_t n_out, int64_t n_src,
                    int accumulate, tg_stream_t stream);
int tg_pingpong_grad(const float* g, float* out, int n, int te, int64_t inner, tg_stream_t stream);
int tg_d_assemble_fwd(const float* data, int t_data, const float* warped, const float* cond,
                      int t_cond, float* x, int n, int t, int c, int h, int w, int pad, int crop,
                      tg_stream_t stream);
int tg_d_assemble_bwd(const float* g, float* g_data, int t_data, float* g_warped, int n, int t, int c,
                      int h, int w, int pad, int crop, tg_stream_t stream);

/* ------------------------------------------------------------------------
 * Training-batch assembly from an HBM-resident uint8 training set (the decoded LMDB of
 * scripts/create_lmdb.py:57: raw RGB HWC frames).  Replaces the per-sample CPU work of
 * UnpairedLMDBDataset.__getitem__ (codes/data/unpaired_lmdb_dataset.py:55-89: frame windows
 * incl. "moving first frame", crop_sequence :95-109, augment_sequence :112-129, /255) and the
 * H2D copy of the fp32 batch.  Geometry is the caller's (drawn with the reference's random
 * streams):
 *   geo (n, t, 4) int64: byte offset of the stored frame in `store`, frame width, window
 *                        row0, col0;   aug (n, 3) int32: flip axis (0 | 2 rows | 3 columns,
 *                        numpy axes of the tchw stack), temporal flip (0|1), np.rot90 count;
 *   out (n, t, c, size, size) fp32 = rot90(flip_t(flip_s(windows))) / 255.
 * ---------------------------------------------------------------------- */
int tg_gather_clips_u8(const uint8_t* store, const int64_t* geo, const int32_t* aug,
                       float* out, int n, int t, int c, int size, tg_stream_t stream);

/* ------------------------------------------------------------------------
 * RCCL exchange of the data-parallel training step (one process per GPU, xGMI).
 * Replaces DistributedDataParallel's gradient all-reduce (base_model.py:130-136), the
 * SyncBatchNorm statistics exchange (:133) and dist.all_reduce of the adaptive-D scalars
 * (vsrgan_model.py:166-173) for a host WITHOUT torch.distributed: rank 0 calls
 * tg_comm_get_unique_id and ships the 128 bytes to the other ranks over any channel it has
 * (the reference's launcher: env:// TCP, utils/dist_utils.py:8-24); every rank then calls
 * tg_comm_init_rank (collective, current HIP device).  Collectives only enqueue on `stream`.
 * RCCL is bound with dlopen at the first tg_comm_* call (the copy already mapped in the
 * process wins); a missing RCCL is TG_E_HIP, never a silent single-rank run.
 * The Python host mirror keeps the reference's own transport by default -- torch.distributed,
 * whose "nccl" backend IS this same RCCL -- because the process group's lifetime and
 * rendezvous belong to the host application (utils/dist_utils.py:init_dist); it switches to
 * these entry points with TECOGAN_COMM=c_abi.
 * ---------------------------------------------------------------------- */
#define TG_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct tg_comm tg_comm;
int tg_comm_get_unique_id(uint8_t id[TG_COMM_ID_BYTES]);
int tg_comm_init_rank(const uint8_t id[TG_COMM_ID_BYTES], int world, int rank, tg_comm** out);
int tg_comm_destroy(tg_comm* comm);
int tg_comm_world(const tg_comm* comm);
int tg_comm_rank(const tg_comm* comm);
/* what RCCL ITSELF reports for the communicator (ncclCommCount, ncclCommUserRank) -- the reference reads the same from
 * torch.distributed (dist.get_world_size() / get_rank(), utils/dist_utils.py:27-34); a launcher / library mismatch shows
 * here before the first collective.  TG_E_HIP when the bound librccl lacks the two queries. */
int tg_comm_query(const tg_comm* comm, int* ranks_seen, int* rank_seen);
const char* tg_comm_library_origin(void); /* which librccl was bound ("" = none yet) */
/* in-place SUM over ranks (flat gradient bucket: 10.4 MB G / 3.3 MB D; BN backward sums) */
int tg_allreduce_sum_f32(tg_comm* comm, float* buf, int64_t count, tg_stream_t stream);
/* recv = [rank 0 | rank 1 | ...], count_per_rank floats each (SyncBN forward statistics) */
int tg_allgather_f32(tg_comm* comm, const float* send, float* recv, int64_t count_per_rank,
                     tg_stream_t stream);

/* ------------------------------------------------------------------------
 * Whole-frame plan: one call = FRNet.step (tecogan_nets.py:227-252).  The plan
 * holds only launch geometry and pointers into a caller-owned workspace and
 * caller-owned packed weights; it owns no device memory.
 * ---------------------------------------------------------------------- */
typedef struct tg_frnet_plan tg_frnet_plan;

typedef struct {
  int in_nc, out_nc, nf, nb, scale, up_mode; /* FRNet ctor, tecogan_nets.py:154 */
  int n, h, w;                               /* LR batch / size */
  int fnet_only;                             /* 1: plan and workspace for phase 1 (FNet) only */
} tg_frnet_cfg;

/* Weight table handed to the plan: device pointers to packed weights / biases
 * in the order documented in tecogan-pytorch_amd/models/networks/tecogan_nets.py
 * (FNet 14 convs, SRNet conv_in, 2*nb resblock convs, 1-2 conv_up, conv_out). */
typedef struct {
  const float* w;
  const float* b;
  const float* u;   /* tg_pack_conv3x3_wino form of the same weights, or NULL: the plan then runs the
                       layer in the direct form whatever its shape (see tg_conv3x3_prefers_wino) */
} tg_layer_weights;

size_t tg_frnet_workspace_floats(const tg_frnet_cfg* cfg);
int tg_frnet_plan_create(const tg_frnet_cfg* cfg, const tg_layer_weights* layers,
                         int n_layers, float* workspace, tg_frnet_plan** out);
void tg_frnet_plan_destroy(tg_frnet_plan* plan);
/* Fail-safe of the chained SRNet launch (tg_conv3x3_wino_chain inside the plan).  A workgroup that
 * gives up waiting for a producer tile counts a fault in a pinned-host counter the plan owns (64
 * bytes of hipHostMalloc memory, the only allocation a plan makes) and carries on, so the launch
 * always ends.  EVERY later tg_frnet_step* / tg_frnet_replay call on the plan looks at that counter
 * first (a host read, no synchronisation): the first call that sees it non-zero returns TG_E_HIP
 * ("frames since the fault are invalid") and switches the plan to one launch per layer (until it re-arms: see below);
 * calls after that run normally on the fallback.  tg_frnet_plan_chain_status does the same check
 * on demand (call it after a synchronisation: it then covers everything enqueued so far):
 * returns TG_OK / TG_E_HIP as above, *faults_total = faults since creation, *chain_active = 1
 * while the plan's shape still uses the chained launch.
 * tg_frnet_plan_set_chain_poll_limit: polls (~64 shader cycles each) a waiter makes before it
 * gives up; default 2^21 (~0.2 s).  A negative limit makes every waiter fault immediately
 * (fault injection for tests of the host's error path). */
int tg_frnet_plan_chain_status(tg_frnet_plan* plan, int* faults_total, int* chain_active);
int tg_frnet_plan_set_chain_poll_limit(tg_frnet_plan* plan, int poll_limit);
/* Recovery from a transient fault (round 6).  "For good" above is the round-5 behaviour and what
 * first_after_frames = 0 selects.  By default (64) the plan counts the frames it enqueues on the per-layer
 * fallback; after that many -- and once the GPU has passed the first of them (a hipEvent the plan creates on its
 * first fault and owns) with no new fault counted -- the one-launch body is armed again (its flags / exchange
 * buffers are re-zeroed in stream order).  A fault of the re-armed body is reported like the first one and
 * DOUBLES the wait (capped at 2^20 frames): a permanent co-tenant costs one invalid clip per back-off period,
 * a transient one costs 64 slower frames.  Results are bit-identical on either path.
 * tg_frnet_plan_chain_rearms: *rearms = how often the body was armed again, *current_wait_frames = the
 * back-off in force (0 before the first fault). */
int tg_frnet_plan_set_chain_rearm(tg_frnet_plan* plan, int first_after_frames);
int tg_frnet_plan_chain_rearms(const tg_frnet_plan* plan, int* rearms, int* current_wait_frames);
/* hr_out may alias nothing else; lr_curr/lr_prev (n,c,h,w), hr_prev/hr_out (n,c,s*h,s*w).
 * u8_out (optional): (n, s*h, s*w, c) uint8 quantised frames (n > 1 needs the fused HR stage:
 * out_nc <= 3, nf <= 64). */
int tg_frnet_step(tg_frnet_plan* plan, const float* lr_curr, const float* lr_prev,
                  const float* hr_prev, float* hr_out, uint8_t* u8_out,
                  tg_stream_t stream);
/* The frame in two phases, for clip inference: FNet depends only on the LR frames, so
 * phase 1 of frame t+1 may run on a second stream while phase 2 of frame t is in flight.
 *   phases & 1: FNet(lr_curr, lr_prev) -> internal flow slot `flow_slot` (0|1)
 *   phases & 2: pad/upsample/warp/s2d + SRNet reading that slot -> hr_out (+ u8_out)
 * The two phases use disjoint workspace regions; the caller orders
 * phase1(slot) -> phase2(slot) -> next phase1(slot) with stream events. */
int tg_frnet_step_phase(tg_frnet_plan* plan, int phases, int flow_slot, const float* lr_curr,
                        const float* lr_prev, const float* hr_prev, float* hr_out,
                        uint8_t* u8_out, tg_stream_t stream);
/* Clip inference with a BATCHED flow estimator: FNet needs only the LR frames, so an
 * FNet-only plan (cfg.fnet_only = 1, cfg.n = frames per batch) estimates the flows of n
 * consecutive frame pairs in one pass -- as FRNet.forward_sequence does for training
 * (tecogan_nets.py:183-196) -- and the per-frame plan consumes them one by one:
 *   tg_frnet_step_phase(fnet_plan, 1, slot, lr[i0..i0+n), lr[i0-1..i0+n-1), ...)
 *   tg_frnet_step_srnet(frame_plan, tg_frnet_plan_flow(fnet_plan, slot) + j*2*fh*fw, lr[i0+j], ...)
 * with fh = h/8*8, fw = w/8*8. */
float* tg_frnet_plan_flow(tg_frnet_plan* plan, int flow_slot);
int tg_frnet_step_srnet(tg_frnet_plan* plan, const float* lr_flow, const float* lr_curr,
                        const float* hr_prev, float* hr_out, uint8_t* u8_out,
                        tg_stream_t stream);
/* number of kernel launches one tg_frnet_step enqueues (for reporting) */
int tg_frnet_plan_launches(const tg_frnet_plan* plan);

/* Measurement support: launches are grouped in `kinds` = distinct kernel symbols
 * (what `rocprofv3 --kernel-trace --stats` groups by).  tg_frnet_step_masked
 * enqueues only the launches whose kind bit is set in kind_mask (buffers keep
 * the contents of the last full step), so bench.py can bracket one kernel
 * class with HIP events; tg_frnet_plan_kind_stats returns that class's launch
 * count and ALGORITHMIC flops / bytes per frame. */
int tg_frnet_plan_kinds(void);
const char* tg_frnet_kind_name(int kind);
int tg_frnet_plan_kind_stats(const tg_frnet_plan* plan, int kind, int* launches,
                             double* flops, double* bytes);
int tg_frnet_replay(tg_frnet_plan* plan, const float* lr_curr, const float* lr_prev,
                    const float* hr_prev, float* hr_out, unsigned kind_mask, int reps,
                    tg_stream_t stream);
int tg_frnet_step_masked(tg_frnet_plan* plan, const float* lr_curr, const float* lr_prev,
                         const float* hr_prev, float* hr_out, uint8_t* u8_out,
                         unsigned kind_mask, tg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TECOGAN_HIP_H */

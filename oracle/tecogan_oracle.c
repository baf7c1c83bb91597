/*
 * Plain-C restatement of the FRVSR/TecoGAN inference frame (FRNet.step).
 *
 * *** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***  Built by oracle/Makefile
 * into oracle/_build/liboracle_c.so and used only by tests/ (second,
 * torch-independent check of the HIP path) -- never by the product.
 *
 * Every routine is a direct loop nest over the defining formula; citations
 * are to the upstream reference (skycrapers/TecoGAN-PyTorch @ v1):
 *   conv3x3 / convT        codes/models/networks/tecogan_nets.py:23-65, 92-98, 111-131
 *   maxpool / bilinear x2  tecogan_nets.py:28,35,42 / :74-79
 *   bicubic / bilinear up  codes/utils/net_utils.py:101-156 / :86-89
 *   backward_warp          codes/utils/net_utils.py:50-82
 *   space_to_depth         codes/utils/net_utils.py:36-47
 *   float32_to_uint8       codes/utils/data_utils.py:80-87
 *   FRNet.step             tecogan_nets.py:227-252
 * Parity pinning: checked against the committed tests/golden vectors (outputs of the reference
 * itself) in tests/test_oracle_c.py.  Accumulation is fp32 in a fixed order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* act: 0 none, 1 relu, 2 lrelu(0.2), 3 tanh*24 */
static inline float actf(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v >= 0.f ? v : v * 0.2f;
    case 3: return tanhf(v) * 24.f;
    default: return v;
  }
}

/* y[n][co][h][w] = act(b[co] + sum x[n][ci][y+ky-1][x+kx-1] * w[co][ci][ky][kx]) (+ res) */
void orc_conv3x3(const float* x, const float* w, const float* b, const float* res, float* y,
                 int n, int cin, int cout, int h, int wd, int act) {
#pragma omp parallel for collapse(2)
  for (int in = 0; in < n; ++in)
    for (int co = 0; co < cout; ++co) {
      float* yd = y + ((size_t)in * cout + co) * h * wd;
      /* accumulate in a private plane so that y may alias res (resblock tail) */
      float* yo = (float*)malloc((size_t)h * wd * sizeof(float));
      for (int i = 0; i < h * wd; ++i) yo[i] = b ? b[co] : 0.f;
      for (int ci = 0; ci < cin; ++ci) {
        const float* xi = x + ((size_t)in * cin + ci) * h * wd;
        const float* wk = w + ((size_t)co * cin + ci) * 9;
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            float wv = wk[ky * 3 + kx];
            int y0 = ky == 0 ? 1 : 0, y1 = ky == 2 ? h - 1 : h;
            int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? wd - 1 : wd;
            for (int yy = y0; yy < y1; ++yy) {
              const float* xr = xi + (size_t)(yy + ky - 1) * wd + (kx - 1);
              float* yr = yo + (size_t)yy * wd;
              for (int xx = x0; xx < x1; ++xx) yr[xx] += wv * xr[xx];
            }
          }
      }
      for (int i = 0; i < h * wd; ++i) {
        float v = actf(yo[i], act);
        if (res) v += res[((size_t)in * cout + co) * h * wd + i];
        yd[i] = v;
      }
      free(yo);
    }
}

/* ConvTranspose2d(k3,s2,p1,op1): y[n][co][2h][2w]; w is (cin, cout, 3, 3); oy = 2*iy - 1 + ky */
void orc_convt3x3s2(const float* x, const float* w, const float* b, float* y, int n, int cin,
                    int cout, int h, int wd, int act) {
  int oh = 2 * h, ow = 2 * wd;
#pragma omp parallel for collapse(2)
  for (int in = 0; in < n; ++in)
    for (int co = 0; co < cout; ++co) {
      float* yo = y + ((size_t)in * cout + co) * oh * ow;
      for (int i = 0; i < oh * ow; ++i) yo[i] = b ? b[co] : 0.f;
      for (int ci = 0; ci < cin; ++ci) {
        const float* xi = x + ((size_t)in * cin + ci) * h * wd;
        const float* wk = w + ((size_t)ci * cout + co) * 9;
        for (int iy = 0; iy < h; ++iy)
          for (int ix = 0; ix < wd; ++ix) {
            float xv = xi[(size_t)iy * wd + ix];
            for (int ky = 0; ky < 3; ++ky) {
              int oy = 2 * iy - 1 + ky;
              if (oy < 0 || oy >= oh) continue;
              for (int kx = 0; kx < 3; ++kx) {
                int ox = 2 * ix - 1 + kx;
                if (ox < 0 || ox >= ow) continue;
                yo[(size_t)oy * ow + ox] += xv * wk[ky * 3 + kx];
              }
            }
          }
      }
      for (int i = 0; i < oh * ow; ++i) yo[i] = actf(yo[i], act);
    }
}

void orc_maxpool2(const float* x, float* y, int nc, int h, int w) {
  int oh = h / 2, ow = w / 2;
  for (int p = 0; p < nc; ++p)
    for (int oy = 0; oy < oh; ++oy)
      for (int ox = 0; ox < ow; ++ox) {
        const float* s = x + ((size_t)p * h + 2 * oy) * w + 2 * ox;
        float a = s[0] > s[1] ? s[0] : s[1], c = s[w] > s[w + 1] ? s[w] : s[w + 1];
        y[((size_t)p * oh + oy) * ow + ox] = a > c ? a : c;
      }
}

static void bicubic_w(int d, int f, float k[4]) {
  const float a = -0.75f;
  float s = (float)d / (float)f, s2 = s * s, s3 = s2 * s;
  k[0] = a * s + (-2.f * a) * s2 + a * s3;
  k[1] = 1.f + (-(a + 3.f)) * s2 + (a + 2.f) * s3;
  k[2] = (-a) * s + (2.f * a + 3.f) * s2 + (-(a + 2.f)) * s3;
  k[3] = a * s2 + (-a) * s3;
}

static void bilinear_src(int dst, int scale, int in_size, int* i0, int* i1, float* l0, float* l1) {
  float src = ((float)dst + 0.5f) * (1.0f / (float)scale) - 0.5f;
  if (src < 0.f) src = 0.f;
  *i0 = (int)floorf(src);
  *i1 = *i0 + 1 < in_size ? *i0 + 1 : in_size - 1;
  *l1 = src - (float)*i0;
  *l0 = 1.0f - *l1;
}

/* mode 1 bicubic (BD), 2 bilinear align_corners=False (BI); y = mul * up(x) */
void orc_upsample(const float* x, float* y, int nc, int h, int w, int s, int mode, float mul) {
  int oh = h * s, ow = w * s;
  for (int p = 0; p < nc; ++p) {
    const float* src = x + (size_t)p * h * w;
    for (int oy = 0; oy < oh; ++oy)
      for (int ox = 0; ox < ow; ++ox) {
        float v;
        if (mode == 1) {
          int i = oy / s, dy = oy - i * s, j = ox / s, dx = ox - j * s;
          float ky[4], kx[4];
          bicubic_w(dy, s, ky);
          bicubic_w(dx, s, kx);
          v = 0.f;
          for (int q = 0; q < 4; ++q) {
            int cq = clampi(j - 1 + q, 0, w - 1);
            float vq = 0.f;
            for (int pp = 0; pp < 4; ++pp) vq += ky[pp] * src[(size_t)clampi(i - 1 + pp, 0, h - 1) * w + cq];
            v += kx[q] * vq;
          }
        } else {
          int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
          bilinear_src(oy, s, h, &y0, &y1, &ly0, &ly1);
          bilinear_src(ox, s, w, &x0, &x1, &lx0, &lx1);
          float top = lx0 * src[(size_t)y0 * w + x0] + lx1 * src[(size_t)y0 * w + x1];
          float bot = lx0 * src[(size_t)y1 * w + x0] + lx1 * src[(size_t)y1 * w + x1];
          v = ly0 * top + ly1 * bot;
        }
        y[((size_t)p * oh + oy) * ow + ox] = mul * v;
      }
  }
}

static float linspace_m1p1(int i, int n, float step) {
  return (i < n / 2) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
}
static float warp_coord(int i, int n, float flow) {
  float step = 2.0f / (float)(n - 1), half = (float)(n - 1) / 2.0f;
  float g = linspace_m1p1(i, n, step) + flow / half;
  float p = (g + 1.0f) * half;
  if (!(p > 0.f)) p = 0.f;
  if (p > (float)(n - 1)) p = (float)(n - 1);
  return p;
}

void orc_backward_warp(const float* x, const float* flow, float* y, int n, int c, int h, int w) {
  size_t hw = (size_t)h * w;
  for (int b = 0; b < n; ++b)
    for (int py = 0; py < h; ++py)
      for (int px = 0; px < w; ++px) {
        float sx = warp_coord(px, w, flow[((size_t)b * 2) * hw + (size_t)py * w + px]);
        float sy = warp_coord(py, h, flow[((size_t)b * 2 + 1) * hw + (size_t)py * w + px]);
        float fx0 = floorf(sx), fy0 = floorf(sy);
        float wx1 = sx - fx0, wx0 = 1.f - wx1, wy1 = sy - fy0, wy0 = 1.f - wy1;
        int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        for (int ch = 0; ch < c; ++ch) {
          const float* img = x + ((size_t)b * c + ch) * hw;
          float v00 = img[(size_t)y0 * w + x0];
          float v01 = x1 <= w - 1 ? img[(size_t)y0 * w + x1] : 0.f;
          float v10 = y1 <= h - 1 ? img[(size_t)y1 * w + x0] : 0.f;
          float v11 = (x1 <= w - 1 && y1 <= h - 1) ? img[(size_t)y1 * w + x1] : 0.f;
          y[((size_t)b * c + ch) * hw + (size_t)py * w + px] =
              ((v00 * (wy0 * wx0) + v01 * (wy0 * wx1)) + v10 * (wy1 * wx0)) + v11 * (wy1 * wx1);
        }
      }
}

void orc_space_to_depth(const float* x, float* y, int n, int c, int h, int w, int s) {
  int oh = h / s, ow = w / s;
  for (int b = 0; b < n; ++b)
    for (int sy = 0; sy < s; ++sy)
      for (int sx = 0; sx < s; ++sx)
        for (int ch = 0; ch < c; ++ch)
          for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox)
              y[(((size_t)b * s * s * c + (sy * s + sx) * c + ch) * oh + oy) * ow + ox] =
                  x[(((size_t)b * c + ch) * h + oy * s + sy) * w + ox * s + sx];
}

/* F.pad(x, (0, pw, 0, ph), 'reflect'): padded index f+k mirrors f-2-k */
void orc_reflect_pad_br(const float* x, float* y, int nc, int h, int w, int ph, int pw) {
  int oh = h + ph, ow = w + pw;
  for (int p = 0; p < nc; ++p)
    for (int oy = 0; oy < oh; ++oy)
      for (int ox = 0; ox < ow; ++ox) {
        int sy = oy < h ? oy : 2 * h - 2 - oy, sx = ox < w ? ox : 2 * w - 2 - ox;
        y[((size_t)p * oh + oy) * ow + ox] = x[((size_t)p * h + sy) * w + sx];
      }
}

/* (c,h,w) fp32 -> (h,w,c) uint8, round-half-even */
void orc_quantize_u8_hwc(const float* x, uint8_t* y, int c, int h, int w) {
  for (int p = 0; p < h * w; ++p)
    for (int ch = 0; ch < c; ++ch) {
      float v = rintf(x[(size_t)ch * h * w + p] * 255.0f);
      v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
      y[(size_t)p * c + ch] = (uint8_t)v;
    }
}

/*
 * FRNet.step.  Weights in the layer order of the product plan: FNet 14 convs
 * (w,b pairs), SRNet conv_in, 2*nb resblock convs, 1-2 conv_up (cin,cout,3,3),
 * conv_out.  up_mode: 1 bicubic / 2 bilinear.  Returns 0, or -1 on alloc failure.
 */
int orc_frnet_step(const float* const* wts, const float* const* bss, int nb, int nf, int scale,
                   int up_mode, const float* lr_curr, const float* lr_prev, const float* hr_prev,
                   float* hr_out, int n, int h, int w) {
  const int c = 3, s = scale;
  size_t hw = (size_t)h * w;
  size_t big = (size_t)n * 256 * hw;
  float* A = (float*)malloc(big * sizeof(float));
  float* B = (float*)malloc(big * sizeof(float));
  float* cat = (float*)malloc((size_t)n * (c + s * s * c) * hw * sizeof(float));
  float* hrf = (float*)malloc((size_t)n * 2 * s * s * hw * sizeof(float));
  float* warped = (float*)malloc((size_t)n * c * s * s * hw * sizeof(float));
  float* U1 = (float*)malloc((size_t)n * nf * 4 * hw * sizeof(float));
  float* U2 = (float*)malloc((size_t)n * nf * 16 * hw * sizeof(float));
  if (!A || !B || !cat || !hrf || !warped || !U1 || !U2) return -1;
  int li = 0;
  /* FNet: cat(x1, x2) */
  for (int b = 0; b < n; ++b) {
    memcpy(A + (size_t)b * 2 * c * hw, lr_curr + (size_t)b * c * hw, c * hw * sizeof(float));
    memcpy(A + ((size_t)b * 2 * c + c) * hw, lr_prev + (size_t)b * c * hw, c * hw * sizeof(float));
  }
  int hh = h, ww = w, cin = 2 * c;
  const int enc[3] = {32, 64, 128}, dec[3] = {256, 128, 64};
  for (int e = 0; e < 3; ++e) {
    orc_conv3x3(A, wts[li], bss[li], NULL, B, n, cin, enc[e], hh, ww, 2); ++li;
    orc_conv3x3(B, wts[li], bss[li], NULL, A, n, enc[e], enc[e], hh, ww, 2); ++li;
    orc_maxpool2(A, B, n * enc[e], hh, ww);
    hh /= 2; ww /= 2; cin = enc[e];
    float* t = A; A = B; B = t;
  }
  for (int d = 0; d < 3; ++d) {
    orc_conv3x3(A, wts[li], bss[li], NULL, B, n, cin, dec[d], hh, ww, 2); ++li;
    orc_conv3x3(B, wts[li], bss[li], NULL, A, n, dec[d], dec[d], hh, ww, 2); ++li;
    orc_upsample(A, B, n * dec[d], hh, ww, 2, 2, 1.0f);
    hh *= 2; ww *= 2; cin = dec[d];
    float* t = A; A = B; B = t;
  }
  orc_conv3x3(A, wts[li], bss[li], NULL, B, n, 64, 32, hh, ww, 2); ++li;
  orc_conv3x3(B, wts[li], bss[li], NULL, A, n, 32, 2, hh, ww, 3); ++li;     /* lr_flow in A */
  /* pad, upsample * scale, warp, space_to_depth */
  orc_reflect_pad_br(A, B, n * 2, hh, ww, h - hh, w - ww);
  orc_upsample(B, hrf, n * 2, h, w, s, up_mode, (float)s);
  orc_backward_warp(hr_prev, hrf, warped, n, c, s * h, s * w);
  for (int b = 0; b < n; ++b)
    memcpy(cat + (size_t)b * (c + s * s * c) * hw, lr_curr + (size_t)b * c * hw, c * hw * sizeof(float));
  {
    float* tmp = (float*)malloc((size_t)n * s * s * c * hw * sizeof(float));
    if (!tmp) return -1;
    orc_space_to_depth(warped, tmp, n, c, s * h, s * w, s);
    for (int b = 0; b < n; ++b)
      memcpy(cat + ((size_t)b * (c + s * s * c) + c) * hw, tmp + (size_t)b * s * s * c * hw,
             (size_t)s * s * c * hw * sizeof(float));
    free(tmp);
  }
  /* SRNet */
  orc_conv3x3(cat, wts[li], bss[li], NULL, A, n, c + s * s * c, nf, h, w, 1); ++li;
  for (int b = 0; b < nb; ++b) {
    orc_conv3x3(A, wts[li], bss[li], NULL, B, n, nf, nf, h, w, 1); ++li;
    orc_conv3x3(B, wts[li], bss[li], A, A, n, nf, nf, h, w, 0); ++li;
  }
  orc_convt3x3s2(A, wts[li], bss[li], U1, n, nf, nf, h, w, 1); ++li;
  const float* top = U1;
  if (s == 4) { orc_convt3x3s2(U1, wts[li], bss[li], U2, n, nf, nf, 2 * h, 2 * w, 1); ++li; top = U2; }
  orc_conv3x3(top, wts[li], bss[li], NULL, hr_out, n, nf, c, s * h, s * w, 0); ++li;
  orc_upsample(lr_curr, warped, n * c, h, w, s, up_mode, 1.0f);
  for (size_t i = 0; i < (size_t)n * c * s * s * hw; ++i) hr_out[i] += warped[i];
  free(A); free(B); free(cat); free(hrf); free(warped); free(U1); free(U2);
  return 0;
}

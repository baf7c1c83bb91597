// Training-batch assembly from the HBM-resident uint8 training set.
//
// Replaces, per sample, the CPU work of UnpairedLMDBDataset.__getitem__
// (codes/data/unpaired_lmdb_dataset.py:55-89): frame selection / "moving first frame" windows,
// crop_sequence (:95-109), augment_sequence (:112-129: spatial flip, temporal flip, np.rot90)
// and `torch.FloatTensor(pats) / 255.0` -- and the pinned-memory H2D copy of the fp32 batch.
// The geometry is drawn on the host with the reference's random streams; this kernel only moves
// bytes: 1 byte read + 4 bytes written per output element, one launch per batch.
#include "tg_common.h"

namespace tg {

// geo[(n*t + j)*4 + {0,1,2,3}] = byte offset of the stored frame, its width, window row0, col0
// aug[n*3 + {0,1,2}]           = flip axis (0 | 2 rows | 3 columns), temporal flip, rot90 count
__global__ __launch_bounds__(256) void gather_clips_u8_kernel(const uint8_t* __restrict__ store,
                                                              const long long* __restrict__ geo,
                                                              const int* __restrict__ aug,
                                                              float* __restrict__ out, int t, int c,
                                                              int s, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % s); long long r = i / s;
    const int y = (int)(r % s); r /= s;
    const int ch = (int)(r % c); r /= c;
    const int tt = (int)(r % t);
    const int n = (int)(r / t);
    const int flip_axis = aug[n * 3], flip_t = aug[n * 3 + 1], k = aug[n * 3 + 2] & 3;
    // inverse of np.rot90(A, k, axes=(2, 3)):  R[y][x] = A[y1][x1]
    int y1, x1;
    if (k == 0) { y1 = y; x1 = x; }
    else if (k == 1) { y1 = x; x1 = s - 1 - y; }
    else if (k == 2) { y1 = s - 1 - y; x1 = s - 1 - x; }
    else { y1 = s - 1 - x; x1 = y; }
    const int ts = flip_t ? t - 1 - tt : tt;                 // np.flip(pats, 0)
    if (flip_axis == 2) y1 = s - 1 - y1;                     // np.flip(pats, 2)
    else if (flip_axis == 3) x1 = s - 1 - x1;                // np.flip(pats, 3)
    const long long* g = geo + ((long long)n * t + ts) * 4;
    const long long off = g[0], w = g[1];
    const uint8_t v = store[off + ((g[2] + y1) * w + (g[3] + x1)) * c + ch];    // HWC frame
    out[i] = (float)v / 255.0f;                              // torch.FloatTensor(pats) / 255.0
  }
}

}  // namespace tg

extern "C" int tg_gather_clips_u8(const uint8_t* store, const int64_t* geo, const int32_t* aug,
                                  float* out, int n, int t, int c, int size, tg_stream_t stream) {
  TG_REQUIRE(store && geo && aug && out, TG_E_ARG, "gather_clips_u8: null pointer");
  TG_REQUIRE(n > 0 && t > 0 && c > 0 && size > 0, TG_E_SHAPE, "gather_clips_u8: n=%d t=%d c=%d size=%d",
             n, t, c, size);
  const long long total = (long long)n * t * c * size * size;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tg::gather_clips_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     store, (const long long*)geo, aug, out, t, c, size, total);
  return tg::check_launch("gather_clips_u8");
}

// M9 (model side): the decode head's final 1x1 convolution for SMALL class counts (PASCAL-VOC's 21; reference
// semseg/models/uperforseg.py:262, `self.classifier = nn.Conv2d(channels, num_labels, 1)`), forward and input gradient.
//
//   forward   logits[b][k][p] = bias[k] + sum_c W[k][c] * y[b][p][c]        y: dense channels_last (B, P, Cin), logits: NCHW
//   backward  gy[b][p][c]     = sum_k g[b][k][p] * W[k][c]
//
// At 21 classes this is a 268 MB read (or write) against 2.9 GFLOP: HBM-bound streaming work, not a GEMM for the matrix
// cores (a 128-wide MFMA tile would run 6x the useful products).  hipBLASLt's kernels for the shape (MT32x128) ran it at
// 1.7-1.9 TB/s (160 / 142 us per step at B = 8, 128^2); these stream at the copy ceiling:
//   forward : a lane owns a pixel; the 256 x 32 activation tile goes through LDS (coalesced 128-byte rows in, conflict-free
//             16-byte reads out, next tile prefetched into registers under the FMAs), the weights are wave-uniform scalar
//             loads (s_load: no LDS, no vector registers), plane-contiguous coalesced stores;
//   backward: a lane owns a channel and keeps its W column in registers; the gradient planes of 64 pixels go through LDS
//             and are read back as broadcasts; 1 KB contiguous stores per wave.
// fp32 FMA chains in a fixed order: bitwise reproducible.
#include "sea_common.h"

namespace sea {

typedef float cf4 __attribute__((ext_vector_type(4)));

constexpr int CLS_TILE = 256;   // pixels per block and K chunk of the forward
constexpr int CLS_KC = 32;      // channels per K chunk
constexpr int CLS_LD = 36;      // LDS row stride in floats: 16-byte reads of 16 consecutive rows hit 16 distinct bank quads

template <int CP>
__global__ __launch_bounds__(256) void classifier_fwd_kernel(const float* __restrict__ y, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out, int P,
                                                             int Cin, int cls, int64_t total) {
  __shared__ __attribute__((aligned(16))) float tile[CLS_TILE * CLS_LD];
  const int tid = threadIdx.x;
  const int ntiles = (int)((total + CLS_TILE - 1) / CLS_TILE);
  const int nk = Cin / CLS_KC;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t p0 = (int64_t)t * CLS_TILE;
    float acc[CP];
#pragma unroll
    for (int k = 0; k < CP; ++k) acc[k] = (bias != nullptr && k < cls) ? bias[k] : 0.f;
    cf4 pre[8];
    auto fetch = [&](int kc) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j, row = idx >> 3, c4 = idx & 7;
        int64_t gp = p0 + row;
        gp = gp < total ? gp : total - 1;   // tail rows: read something valid, never stored
        pre[j] = __builtin_nontemporal_load((const cf4*)(y + gp * Cin + kc * CLS_KC + 4 * c4));
      }
    };
    fetch(0);
    for (int kc = 0; kc < nk; ++kc) {
      __syncthreads();   // the previous chunk has been read by every lane
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j, row = idx >> 3, c4 = idx & 7;
        *(cf4*)(tile + row * CLS_LD + 4 * c4) = pre[j];
      }
      __syncthreads();
      if (kc + 1 < nk) fetch(kc + 1);
      cf4 yv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = *(const cf4*)(tile + tid * CLS_LD + 4 * j);
      const float* __restrict__ w = W + kc * CLS_KC;   // wave-uniform: the compiler turns w[..] into scalar loads
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        if (k < cls) {
          float a = acc[k];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            a = fmaf(w[(int64_t)k * Cin + 4 * j + 0], yv[j][0], a);
            a = fmaf(w[(int64_t)k * Cin + 4 * j + 1], yv[j][1], a);
            a = fmaf(w[(int64_t)k * Cin + 4 * j + 2], yv[j][2], a);
            a = fmaf(w[(int64_t)k * Cin + 4 * j + 3], yv[j][3], a);
          }
          acc[k] = a;
        }
      }
    }
    const int64_t gp = p0 + tid;
    if (gp < total) {
      const int64_t b = gp / P, pp = gp - b * P;
#pragma unroll
      for (int k = 0; k < CP; ++k)
        if (k < cls) out[(b * cls + k) * P + pp] = acc[k];
    }
  }
}

// one block: 64 pixels (of one image) x 256 channels
template <int CP>
__global__ __launch_bounds__(256) void classifier_bwd_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                             float* __restrict__ gy, int P, int Cin, int cls, int B) {
  __shared__ __attribute__((aligned(16))) float gt[CP * 64];
  const int tid = threadIdx.x;
  const int c = blockIdx.y * 256 + tid;
  float wr[CP];
#pragma unroll
  for (int k = 0; k < CP; ++k) wr[k] = (k < cls && c < Cin) ? W[(int64_t)k * Cin + c] : 0.f;
  const int tiles_per_img = (P + 63) / 64;
  const int64_t ntiles = (int64_t)B * tiles_per_img;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = (int)(t / tiles_per_img);
    const int p0 = (int)(t - (int64_t)b * tiles_per_img) * 64;
    __syncthreads();
    for (int i = tid; i < CP * 64; i += 256) {
      const int k = i >> 6, pp = p0 + (i & 63);
      gt[i] = (k < cls && pp < P) ? __builtin_nontemporal_load(g + ((int64_t)b * cls + k) * P + pp) : 0.f;
    }
    __syncthreads();
    if (c < Cin) {
      const int np = (P - p0) < 64 ? (P - p0) : 64;
      float* dst = gy + ((int64_t)b * P + p0) * Cin + c;
      for (int pp = 0; pp < np; pp += 4) {
        cf4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CP; ++k) {
          const cf4 gv = *(const cf4*)(gt + k * 64 + pp);   // the same address in every lane: an LDS broadcast
          s += gv * wr[k];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (pp + e < np) dst[(int64_t)(pp + e) * Cin] = s[e];
      }
    }
  }
}

}  // namespace sea

using namespace sea;

extern "C" int sea_classifier_fwd(const float* y, const float* W, const float* bias, float* out, int B, int P, int Cin, int cls,
                                  void* stream) {
  SEA_CHECK_ARG(y && W && out && B > 0 && P > 0 && cls > 0 && cls <= 32 && Cin > 0 && (Cin % CLS_KC) == 0);
  SEA_CHECK_ARG((((uintptr_t)y) & 15) == 0);
  const int64_t total = (int64_t)B * P;
  const int grid = (int)((total + CLS_TILE - 1) / CLS_TILE < 4 * 256 ? (total + CLS_TILE - 1) / CLS_TILE : 4 * 256);
  if (cls <= 24)
    hipLaunchKernelGGL(classifier_fwd_kernel<24>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, W, bias, out, P, Cin, cls, total);
  else
    hipLaunchKernelGGL(classifier_fwd_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, W, bias, out, P, Cin, cls, total);
  SEA_RETURN_LAST();
}

extern "C" int sea_classifier_bwd(const float* g, const float* W, float* gy, int B, int P, int Cin, int cls, void* stream) {
  SEA_CHECK_ARG(g && W && gy && B > 0 && P > 0 && cls > 0 && cls <= 32 && Cin > 0);
  const int64_t ntiles = (int64_t)B * ((P + 63) / 64);
  const int cb = (Cin + 255) / 256;
  int gx = (int)(ntiles < (kMaxGridX * 2) / cb ? ntiles : (kMaxGridX * 2) / cb);
  if (gx < 1) gx = 1;
  if (cls <= 24)
    hipLaunchKernelGGL(classifier_bwd_kernel<24>, dim3(gx, cb), dim3(256), 0, (hipStream_t)stream, g, W, gy, P, Cin, cls, B);
  else
    hipLaunchKernelGGL(classifier_bwd_kernel<32>, dim3(gx, cb), dim3(256), 0, (hipStream_t)stream, g, W, gy, P, Cin, cls, B);
  SEA_RETURN_LAST();
}

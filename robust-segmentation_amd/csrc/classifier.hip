// M10: the decode head's final 1 x 1 convolution (reference semseg/models/uperforseg.py:262, `cls_seg` -> conv_seg) for a
// SMALL number of classes (<= 32: PASCAL-VOC's 21), forward and input gradient, frozen weights.
//
//   forward   logits[b][c][p] = sum_k W[c][k] y[b][p][k] + bias[c]     y: NHWC rows (B P, K), logits: NCHW (B, cls, P)
//   backward  gy[b][p][k]     = sum_c g[b][c][p] W[c][k]               g: NCHW,            gy: NHWC rows
//
// Why a kernel of our own: with 21 output rows the product is a stream over y (268 MB at 8 x 128 x 128 x 512: 47 us at the
// copy ceiling) with 2.8 GFLOP attached; the library's kernels for the shape (hipBLASLt MT32x128) take 158 us forward and
// 138 us backward -- three times their HBM time, 0.3 ms of a 14.5 ms attack step.  Here the classes sit on one 32-wide
// MFMA dimension of v_mfma_f32_32x32x2_f32 (f32 operands: EXACT fp32 products, fp32 accumulate -- no operand split, so the
// logits carry no M8 rounding), the pixels on the other, and both kernels read / write every activation byte once:
//   forward:  D[class][pixel]: lane = pixel, registers = classes -> a wave-store writes 32 consecutive pixels of one class
//             plane (the NCHW layout K2 wants).  The k index a lane feeds to MFMA step e of an 8-wide chunk is kk + 4 h + e: a
//             lane's four operands of a chunk are ONE 16-byte load of its pixel row (and one ds_read_b128 of W).
//   backward: D[pixel][k]: lane = k column, registers = pixels -> a wave-store writes 128 contiguous bytes of two pixel rows;
//             the class pair {2 s, 2 s + 1} of MFMA step s comes straight from the NCHW gradient (32 consecutive pixels per
//             half wave).
// Fixed summation order (k ascending within a lane half; class pairs ascending): run-to-run and batch-composition
// independent bits.
#include "sea_common.h"

namespace sea {

typedef float cf32x16 __attribute__((ext_vector_type(16)));
typedef float cf32x4 __attribute__((ext_vector_type(4)));

constexpr int CLS_MAX = 32;

// grid: blocks of 256 threads = 4 waves, each wave strides over 32-pixel tiles.  LDS: W as [32][K + 4] floats (the 16-byte
// pad turns the 32 rows a half wave reads at one k into 16 distinct bank slots per service group).
__global__ __launch_bounds__(256, 2) void classifier_fwd_kernel(const float* __restrict__ y, const float* __restrict__ W,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int tiles, int P, int K, int cls) {
  extern __shared__ __attribute__((aligned(16))) char cls_smem[];
  const int ldw = K + 4;
  float* const Ws = (float*)cls_smem;
  for (int i = threadIdx.x; i < CLS_MAX * (K / 4); i += 256) {
    const int c = i / (K / 4), k4 = i - c * (K / 4);
    const cf32x4 v = c < cls ? *(const cf32x4*)(W + (int64_t)c * K + 4 * k4) : cf32x4{0.f, 0.f, 0.f, 0.f};
    *(cf32x4*)(Ws + c * ldw + 4 * k4) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const float* const wrow = Ws + r * ldw + 4 * h;
  const int tiles_per_img = P / 32;
  for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
    const float* const yrow = y + ((int64_t)t * 32 + r) * K + 4 * h;
    cf32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // eight chunks (64 k) per trip, the next trip's loads in flight over this trip's 32 MFMAs
    cf32x4 cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = *(const cf32x4*)(yrow + 8 * i);
    for (int kk = 0; kk < K; kk += 64) {
      if (kk + 64 < K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = *(const cf32x4*)(yrow + kk + 64 + 8 * i);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const cf32x4 a = *(const cf32x4*)(wrow + kk + 8 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], cur[i][e], acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
    const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 32;
    float* const ob = out + (int64_t)b * cls * P + p0 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
      if (c < cls) ob[(int64_t)c * P] = acc[e] + (bias ? bias[c] : 0.f);
    }
  }
}

// LDS: W as [2 ceil(cls / 2)][K + 32] floats (rows >= cls zero; the 128-byte pad puts the two class rows a wave reads at
// once on different bank halves)
// GATE: gy = gate > 0 ? gy * scale[k] : 0 on the way out -- the backward of the ReLU(scale z + shift) that produced the
// classifier's input (the FPN bottleneck's folded BatchNorm + ReLU, uperforseg.py:296-304), bit for bit what the separate
// sea_gate_scale pass gives on the stored gradient (one rounding of the product either way)
template <bool GATE>
__global__ __launch_bounds__(256, 2) void classifier_bwd_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                                float* __restrict__ gy, int tiles, int P, int K, int cls,
                                                                const float* __restrict__ gate, const float* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) char cls_smem[];
  const int ldw = K + 32;
  const int pairs = (cls + 1) / 2;
  float* const Ws = (float*)cls_smem;
  for (int i = threadIdx.x; i < 2 * pairs * (K / 4); i += 256) {
    const int c = i / (K / 4), k4 = i - c * (K / 4);
    const cf32x4 v = c < cls ? *(const cf32x4*)(W + (int64_t)c * K + 4 * k4) : cf32x4{0.f, 0.f, 0.f, 0.f};
    *(cf32x4*)(Ws + c * ldw + 4 * k4) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_per_img = P / 32;
  for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
    const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 32;
    const float* const gb = g + (int64_t)b * cls * P + p0 + r;
    float a[CLS_MAX / 2];
#pragma unroll
    for (int s = 0; s < CLS_MAX / 2; ++s) {
      const int c = 2 * s + h;
      a[s] = (s < pairs && c < cls) ? gb[(int64_t)c * P] : 0.f;
    }
    float* const orow = gy + ((int64_t)t * 32 + 4 * h) * K + r;
    const float* const grow = GATE ? gate + ((int64_t)t * 32 + 4 * h) * K + r : nullptr;
    for (int n0 = 0; n0 < K; n0 += 128) {      // four 32-column tiles per trip (64 accumulator registers)
      cf32x16 acc[4];
      float gt[GATE ? 4 : 1][GATE ? 16 : 1];     // the gate values of this trip, requested before its MFMAs
      float sc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
        if constexpr (GATE) {
          if (n0 + 32 * nt < K) {
            sc[nt] = scale[n0 + 32 * nt + r];
#pragma unroll
            for (int e = 0; e < 16; ++e) gt[nt][e] = grow[(int64_t)((e & 3) + 8 * (e >> 2)) * K + n0 + 32 * nt];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < CLS_MAX / 2; ++s) {
        if (s < pairs) {
          const float* const wr = Ws + (2 * s + h) * ldw + n0 + r;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            if (n0 + 32 * nt < K) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wr[32 * nt], acc[nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (n0 + 32 * nt < K) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t o = (int64_t)((e & 3) + 8 * (e >> 2)) * K + n0 + 32 * nt;
            if constexpr (GATE)
              orow[o] = gt[nt][e] > 0.f ? acc[nt][e] * sc[nt] : 0.f;
            else
              orow[o] = acc[nt][e];
          }
        }
      }
    }
  }
}

}  // namespace sea

using namespace sea;

// 1 when the pair of kernels takes the shape: whole 32-pixel tiles per image, K a multiple of 32 (backward: 32-column tiles;
// forward: 64-wide trips handle any multiple of 8, kept to one rule), at most 32 classes, W within the LDS budget
extern "C" int sea_classifier_supported(int P, int K, int cls) {
  return (P > 0 && (P % 32) == 0 && K >= 64 && (K % 64) == 0 && K <= 1024 && cls >= 1 && cls <= CLS_MAX) ? 1 : 0;
}

static int classifier_grid(int tiles) {
  const int blocks = (tiles + 3) / 4;
  return blocks < 512 ? blocks : 512;          // two blocks per CU, each wave strides over its tiles
}

extern "C" int sea_classifier_fwd(const float* y, const float* W, const float* bias, float* out, int B, int P, int K, int cls,
                                  void* stream) {
  SEA_CHECK_ARG(y && W && out && B > 0 && sea_classifier_supported(P, K, cls));
  SEA_CHECK_ARG(((((uintptr_t)y) | ((uintptr_t)W)) & 15) == 0 && (int64_t)B * P / 32 < (1ll << 30));
  const int tiles = (int)((int64_t)B * P / 32);
  const int lds = CLS_MAX * (K + 4) * 4;
  static bool attr_set_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set_dev[dev & 63]) {
    (void)hipFuncSetAttribute((const void*)classifier_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CLS_MAX * (1024 + 4) * 4);
    attr_set_dev[dev & 63] = true;
  }
  hipLaunchKernelGGL(classifier_fwd_kernel, dim3(classifier_grid(tiles)), dim3(256), (size_t)lds, (hipStream_t)stream, y, W, bias, out,
                     tiles, P, K, cls);
  SEA_RETURN_LAST();
}

extern "C" int sea_classifier_bwd(const float* g, const float* W, float* gy, int B, int P, int K, int cls, const float* gate,
                                  const float* gate_scale, void* stream) {
  SEA_CHECK_ARG(g && W && gy && B > 0 && sea_classifier_supported(P, K, cls) && ((gate == nullptr) == (gate_scale == nullptr)));
  SEA_CHECK_ARG((((uintptr_t)W) & 15) == 0 && (int64_t)B * P / 32 < (1ll << 30));
  const int tiles = (int)((int64_t)B * P / 32);
  const int lds = 2 * ((cls + 1) / 2) * (K + 32) * 4;
  static bool attr_set_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set_dev[dev & 63]) {
    (void)hipFuncSetAttribute((const void*)classifier_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CLS_MAX * (1024 + 32) * 4);
    (void)hipFuncSetAttribute((const void*)classifier_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CLS_MAX * (1024 + 32) * 4);
    attr_set_dev[dev & 63] = true;
  }
  if (gate)
    hipLaunchKernelGGL(classifier_bwd_kernel<true>, dim3(classifier_grid(tiles)), dim3(256), (size_t)lds, (hipStream_t)stream, g, W, gy,
                       tiles, P, K, cls, gate, gate_scale);
  else
    hipLaunchKernelGGL(classifier_bwd_kernel<false>, dim3(classifier_grid(tiles)), dim3(256), (size_t)lds, (hipStream_t)stream, g, W, gy,
                       tiles, P, K, cls, gate, gate_scale);
  SEA_RETURN_LAST();
}

// M3 (model side): layout changes of the ConvNeXt block, fused with the per-channel layer scale and the
// residual add.
//
// The ConvNeXt block (reference convnext_orig.py:73-86) runs its depthwise conv in NCHW and its
// LayerNorm / pointwise MLP in NHWC; in PyTorch the two permutes become strided element-wise kernels
// (`elementwise_kernel_manual_unroll`, ~116 launches and ~7 ms per APGD step at B=8, 512x512, i.e. ~1.4
// TB/s).  Both directions are plain (C x HW) <-> (HW x C) matrix transposes per image and run at HBM
// speed when tiled through LDS:
//   sea_nchw_to_nhwc : out[b,p,c] = scale[c] * in[b,c,p]                      (scale optional)
//   sea_nhwc_to_nchw : out[b,c,p] = residual[b,c,p] + scale[c] * in[b,p,c]    (scale, residual optional)
// 64x64 tiles, LDS row stride 65 (conflict-free transposed reads), 16-byte global accesses on both sides.
// Requires C % 4 == 0 and HW % 4 == 0 (the Python wrapper falls back to PyTorch otherwise).
#include "sea_common.h"

namespace sea {

constexpr int TT = 64;

// in: (B, R, S) row-major (S contiguous); out: (B, S, R).  scale indexed by `r` when SCALE_ON_ROWS
// (NCHW->NHWC: rows are channels) or by `s` otherwise (NHWC->NCHW: input columns are channels).
template <bool SCALE_ROWS, bool HAS_RES>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                                        const float* __restrict__ res, float* __restrict__ out, int R,
                                                        int S) {
  __shared__ float tile[TT][TT + 1];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * TT, s0 = blockIdx.x * TT;
  const float* ip = in + (int64_t)b * R * S;
  float* op = out + (int64_t)b * R * S;
  const int q = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;
  // load: rows r0+rr+16k, cols s0+q..q+3
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + rr + 16 * k, s = s0 + q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && s < S) v = *reinterpret_cast<const float4*>(ip + (int64_t)r * S + s);  // S % 4 == 0
    if (SCALE_ROWS && scale != nullptr && r < R) {
      const float sc = scale[r];
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    }
    tile[rr + 16 * k][q + 0] = v.x;
    tile[rr + 16 * k][q + 1] = v.y;
    tile[rr + 16 * k][q + 2] = v.z;
    tile[rr + 16 * k][q + 3] = v.w;
  }
  __syncthreads();
  // store: output rows s0+rr+16k (input columns), output cols r0+q..q+3 (input rows)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int s = s0 + rr + 16 * k, r = r0 + q;
    if (s < S && r < R) {  // R % 4 == 0
      float4 v = make_float4(tile[q + 0][rr + 16 * k], tile[q + 1][rr + 16 * k], tile[q + 2][rr + 16 * k],
                             tile[q + 3][rr + 16 * k]);
      if (!SCALE_ROWS && scale != nullptr) {
        const float sc = scale[s];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      }
      float* dst = op + (int64_t)s * R + r;
      if (HAS_RES) {
        const float4 a = *reinterpret_cast<const float4*>(res + (int64_t)b * R * S + (int64_t)s * R + r);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      *reinterpret_cast<float4*>(dst) = v;
    }
  }
}

}  // namespace sea

using namespace sea;

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int sea_nchw_to_nhwc(const float* in, const float* scale, float* out, int B, int C, int64_t HW, void* stream) {
  SEA_CHECK_ARG(in && out && B > 0 && B <= 65535 && C > 0 && HW > 0 && (C % 4) == 0 && (HW % 4) == 0);
  SEA_CHECK_ARG(al16(in) && al16(out) && HW <= 2147483647);
  dim3 grid((unsigned)((HW + TT - 1) / TT), (C + TT - 1) / TT, B);
  hipLaunchKernelGGL((transpose_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, in, scale,
                     (const float*)nullptr, out, C, (int)HW);
  SEA_RETURN_LAST();
}

extern "C" int sea_nhwc_to_nchw(const float* in, const float* scale, const float* residual, float* out, int B, int C,
                                int64_t HW, void* stream) {
  SEA_CHECK_ARG(in && out && B > 0 && B <= 65535 && C > 0 && HW > 0 && (C % 4) == 0 && (HW % 4) == 0);
  SEA_CHECK_ARG(al16(in) && al16(out) && (!residual || al16(residual)) && HW <= 2147483647);
  // input matrix is (HW x C): rows = pixels, columns = channels; scale is per input column
  dim3 grid((C + TT - 1) / TT, (unsigned)((HW + TT - 1) / TT), B);
  if (residual)
    hipLaunchKernelGGL((transpose_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, in, scale, residual,
                       out, (int)HW, C);
  else
    hipLaunchKernelGGL((transpose_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, in, scale,
                       (const float*)nullptr, out, (int)HW, C);
  SEA_RETURN_LAST();
}

// ---- 2x2 patch gather / scatter for the trunk's down-sampling convolutions (reference convnext_orig.py:118-124) ------------
// (B,H,W,C) channels_last pixels <-> (B*H/2*W/2, 4*C) patch rows in (di, dj, c) order: the 2x2 / stride-2 convolution
// is then ONE GEMM on the patch matrix.  Pure data movement, 16 bytes per lane, both sides contiguous C-vectors.
namespace sea {
template <bool INVERSE>
__global__ __launch_bounds__(256) void patch2x2_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int H, int W,
                                                       int C4, int64_t n4) {
  const int Wh = W / 2, Hh = H / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int q = (int)(r & 3);
    r >>= 2;
    const int pj = (int)(r % Wh);
    r /= Wh;
    const int pi = (int)(r % Hh);
    const int64_t b = r / Hh;
    const int64_t pix = ((b * H + 2 * pi + (q >> 1)) * W + 2 * pj + (q & 1)) * C4 + c;
    if (INVERSE)
      dst[pix] = src[i];
    else
      dst[i] = src[pix];
  }
}
}  // namespace sea

// inverse == 0: pixels (B,H,W,C) -> patches (B*H/2*W/2, 4C); inverse != 0: patches -> pixels.  C % 4 == 0, H, W even.
extern "C" int sea_patch2x2(const float* src, float* dst, int B, int H, int W, int C, int inverse, void* stream) {
  SEA_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && (H % 2) == 0 && (W % 2) == 0);
  SEA_CHECK_ARG(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0);
  const int64_t n4 = (int64_t)B * H * W * (C / 4);
  const int grid = sea::grid_for(n4, 256);
  if (inverse)
    hipLaunchKernelGGL((sea::patch2x2_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                       (float4*)dst, H, W, C / 4, n4);
  else
    hipLaunchKernelGGL((sea::patch2x2_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                       (float4*)dst, H, W, C / 4, n4);
  SEA_RETURN_LAST();
}

// M5 (model side): LayerNorm over the last (channel) dimension of an NHWC tensor with few channels
// (ConvNeXt: C = 48 / 96 / 192 / 384 / 768, eps = 1e-6; convnext_orig.py:19-40, 75-80), forward and the
// input gradient.  ATen launches one workgroup per row; with 48-96 channels that leaves most lanes idle and
// reaches ~0.4-0.8 TB/s.  Here a row is owned by LPR = 16 / 32 / 64 lanes (float4 each, K float4 per lane for
// C > 256), 64/LPR rows per wave, statistics by xor-shuffles inside the lane group: one streaming pass,
// 16-byte accesses, no LDS.
//   forward :  y = (x - mean) * rstd * w + b          (mean, rstd saved per row; two-pass variance in registers)
//   backward:  dx = rstd * (gw - mean_c(gw) - xhat * mean_c(gw * xhat)),  gw = g * w,  xhat = (x - mean) * rstd
#include "sea_common.h"

namespace sea {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR, int K>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ w,
                                                     const float4* __restrict__ b, float4* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                     int NV, float eps) {
  constexpr int RPB = 256 / LPR;  // rows per block
  const int lane = threadIdx.x % LPR;
  const float inv_c = 1.f / (float)(NV * 4);
  float4 wv[K], bv[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int v = lane + k * LPR;
    wv[k] = v < NV ? w[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    bv[k] = v < NV ? b[v] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float4* xr = x + r * NV;
    float4 xv[K];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int v = lane + k * LPR;
      xv[k] = v < NV ? xr[v] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
    }
    const float mu = group_sum<LPR>(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int v = lane + k * LPR;
      if (v < NV) {
        const float dx = xv[k].x - mu, dy = xv[k].y - mu, dz = xv[k].z - mu, dw = xv[k].w - mu;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    const float rs = rsqrtf(group_sum<LPR>(q) * inv_c + eps);
    float4* yr = y + r * NV;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int v = lane + k * LPR;
      if (v < NV) {
        float4 o;
        o.x = fmaf((xv[k].x - mu) * rs, wv[k].x, bv[k].x);
        o.y = fmaf((xv[k].y - mu) * rs, wv[k].y, bv[k].y);
        o.z = fmaf((xv[k].z - mu) * rs, wv[k].z, bv[k].z);
        o.w = fmaf((xv[k].w - mu) * rs, wv[k].w, bv[k].w);
        yr[v] = o;
      }
    }
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
  }
}

template <int LPR, int K>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float4* __restrict__ g, const float4* __restrict__ x,
                                                     const float4* __restrict__ w, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float4* __restrict__ dx, int64_t rows,
                                                     int NV) {
  constexpr int RPB = 256 / LPR;
  const int lane = threadIdx.x % LPR;
  const float inv_c = 1.f / (float)(NV * 4);
  float4 wv[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int v = lane + k * LPR;
    wv[k] = v < NV ? w[v] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float mu = mean[r], rs = rstd[r];
    float4 gw[K], xh[K];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int v = lane + k * LPR;
      if (v < NV) {
        const float4 gv = g[r * NV + v], xv = x[r * NV + v];
        gw[k] = make_float4(gv.x * wv[k].x, gv.y * wv[k].y, gv.z * wv[k].z, gv.w * wv[k].w);
        xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        s1 += (gw[k].x + gw[k].y) + (gw[k].z + gw[k].w);
        s2 += (gw[k].x * xh[k].x + gw[k].y * xh[k].y) + (gw[k].z * xh[k].z + gw[k].w * xh[k].w);
      } else {
        gw[k] = xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float m1 = group_sum<LPR>(s1) * inv_c, m2 = group_sum<LPR>(s2) * inv_c;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int v = lane + k * LPR;
      if (v < NV) {
        float4 o;
        o.x = rs * (gw[k].x - m1 - xh[k].x * m2);
        o.y = rs * (gw[k].y - m1 - xh[k].y * m2);
        o.z = rs * (gw[k].z - m1 - xh[k].z * m2);
        o.w = rs * (gw[k].w - m1 - xh[k].w * m2);
        dx[r * NV + v] = o;
      }
    }
  }
}

template <int LPR, int K>
static void launch_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int64_t rows,
                       int C, float eps, hipStream_t st) {
  const int rpb = 256 / LPR;
  hipLaunchKernelGGL((ln_fwd_kernel<LPR, K>), dim3(grid_for((rows + rpb - 1) / rpb, 1)), dim3(256), 0, st,
                     (const float4*)x, (const float4*)w, (const float4*)b, (float4*)y, mean, rstd, rows, C / 4, eps);
}
template <int LPR, int K>
static void launch_bwd(const float* g, const float* x, const float* w, const float* mean, const float* rstd, float* dx,
                       int64_t rows, int C, hipStream_t st) {
  const int rpb = 256 / LPR;
  hipLaunchKernelGGL((ln_bwd_kernel<LPR, K>), dim3(grid_for((rows + rpb - 1) / rpb, 1)), dim3(256), 0, st,
                     (const float4*)g, (const float4*)x, (const float4*)w, mean, rstd, (float4*)dx, rows, C / 4);
}

}  // namespace sea

using namespace sea;

#define LN_DISPATCH(FN, ...)                          \
  do {                                                \
    const int nv = C / 4;                             \
    if (nv <= 16) FN<16, 1>(__VA_ARGS__);             \
    else if (nv <= 32) FN<32, 1>(__VA_ARGS__);        \
    else if (nv <= 64) FN<64, 1>(__VA_ARGS__);        \
    else if (nv <= 128) FN<64, 2>(__VA_ARGS__);       \
    else if (nv <= 192) FN<64, 3>(__VA_ARGS__);       \
    else FN<64, 4>(__VA_ARGS__);                      \
  } while (0)

// x, y (rows, C) contiguous fp32; w, b (C); mean, rstd (rows) written.  C % 4 == 0, C <= 1024.
extern "C" int sea_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                                 int64_t rows, int C, float eps, void* stream) {
  SEA_CHECK_ARG(x && w && b && y && mean && rstd && rows > 0 && C >= 4 && (C % 4) == 0 && C <= 1024);
  SEA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w) | ((uintptr_t)b)) & 15) == 0);
  LN_DISPATCH(launch_fwd, x, w, b, y, mean, rstd, rows, C, eps, (hipStream_t)stream);
  SEA_RETURN_LAST();
}

// dx = d loss / d x given g = d loss / d y (w frozen: no parameter gradients)
extern "C" int sea_layernorm_bwd(const float* g, const float* x, const float* w, const float* mean, const float* rstd,
                                 float* dx, int64_t rows, int C, void* stream) {
  SEA_CHECK_ARG(g && x && w && mean && rstd && dx && rows > 0 && C >= 4 && (C % 4) == 0 && C <= 1024);
  SEA_CHECK_ARG(((((uintptr_t)g) | ((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)dx)) & 15) == 0);
  LN_DISPATCH(launch_bwd, g, x, w, mean, rstd, dx, rows, C, (hipStream_t)stream);
  SEA_RETURN_LAST();
}

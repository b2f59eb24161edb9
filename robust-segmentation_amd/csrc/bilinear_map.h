// ATen's bilinear (align_corners=False) source-index rule, shared by the up-sampling kernels (M2) and the fused
// FPN-bottleneck kernels (M6).
#pragma once
#include <hip/hip_runtime.h>

namespace sea {

struct AxisMapU {
  int i0, i1;
  float lam;
};

__device__ __forceinline__ AxisMapU axis_map_u(int dst, float r, int n_in) {
  float src = r * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  AxisMapU m;
  m.i0 = (int)src;
  if (m.i0 > n_in - 1) m.i0 = n_in - 1;
  m.i1 = m.i0 + ((m.i0 < n_in - 1) ? 1 : 0);
  m.lam = src - (float)m.i0;
  return m;
}

__device__ __forceinline__ int first_dst_ge(int t, float r, int n_in, int n_out) {
  if (t <= 0) return 0;
  if (t > n_in - 1) return n_out;
  int d = (int)ceilf(((float)t + 0.5f) / r - 0.5f);
  d = d < 0 ? 0 : (d > n_out ? n_out : d);
  while (d > 0 && axis_map_u(d - 1, r, n_in).i0 >= t) --d;
  while (d < n_out && axis_map_u(d, r, n_in).i0 < t) ++d;
  return d;
}

// weight of source index `src` in the interpolation of destination index `dst`
__device__ __forceinline__ float axis_coef(int dst, int src, float r, int n_in) {
  const AxisMapU m = axis_map_u(dst, r, n_in);
  return ((m.i0 == src) ? (1.f - m.lam) : 0.f) + ((m.i1 == src) ? m.lam : 0.f);
}

}  // namespace sea

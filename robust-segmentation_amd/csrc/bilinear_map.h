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

// Power-of-two factor S in this axis (n_out = S * n_in): with t = dst + S/2 the source pair is i0 = (t >> log2 S) - 1
// (negative: clamped to 0 with lambda = 0, exactly what the clamp of src does) and lambda = ((t & (S-1)) + 0.5) / S.
// r = 1/S and every intermediate of axis_map_u are exact in float for such factors: identical results, integer ops only.
// S = 0 selects the general float rule.
template <int S>
__device__ __forceinline__ AxisMapU axis_map_p2(int dst, float r, int n_in) {
  if constexpr (S == 0) {
    return axis_map_u(dst, r, n_in);
  } else {
    static_assert(S == 2 || S == 4 || S == 8 || S == 16, "supported factors");
    constexpr int LOG = S == 2 ? 1 : S == 4 ? 2 : S == 8 ? 3 : 4;
    const int t = dst + S / 2;
    AxisMapU m;
    m.i0 = (t >> LOG) - 1;
    m.lam = ((float)(t & (S - 1)) + 0.5f) * (1.f / (float)S);
    if (m.i0 < 0) {
      m.i0 = 0;
      m.lam = 0.f;
    }
    m.i1 = min(m.i0 + 1, n_in - 1);
    return m;
  }
}

// weight of source index `src` in the interpolation of destination index `dst`
__device__ __forceinline__ float axis_coef(int dst, int src, float r, int n_in) {
  const AxisMapU m = axis_map_u(dst, r, n_in);
  return ((m.i0 == src) ? (1.f - m.lam) : 0.f) + ((m.i1 == src) ? m.lam : 0.f);
}

}  // namespace sea

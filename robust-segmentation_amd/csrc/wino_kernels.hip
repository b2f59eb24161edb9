// M4 (model side): Winograd F(m x m, 3 x 3) transforms for the 3x3 / stride-1 / pad-1 convolutions of the
// UperNet head (semseg/models/uperforseg.py:200-215, 255-262: fpn_convs, fpn_bottleneck), m = 2 or 4.
//
// The head's fpn_bottleneck (2048 -> 512 channels at 128 x 128, batch 8) is 2.47 TFLOP per direction: at the
// ~157 TFLOP/s fp32 MFMA peak a direct (implicit-GEMM) convolution cannot go below ~16 ms, and MIOpen's igemm
// kernels already sit at 139 TFLOP/s.  Winograd does the same convolution with 2.25x (m=2) / 4x (m=4) fewer
// multiplications:   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A
// This file holds the three HBM-bound transforms; the (m+2)^2 independent GEMMs  M[k] = V[k] (T x Cin) @ U[k]
// (Cin x Cout) in between are a plain strided-batched fp32 GEMM and go to hipBLASLt through torch.bmm
// (measured 147 TFLOP/s on these shapes).
//
// Layouts (all fp32):  x, y  NHWC dense (B,H,W,C);  tiles t = (b, ty, tx), T = B * ceil(H/m) * ceil(W/m);
//   V (A*A, T, Cin)   input tiles in the Winograd domain, A = m + 2, k = i*A + j
//   U (A*A, Cin, Cout) filters in the Winograd domain (or (A*A, Cout, Cin) of the flipped filters for the
//                      input-gradient convolution)
//   M (A*A, T, Cout)  products, turned back into (B,H,W,Cout) by the output transform (+ optional bias)
// Lanes run along channels everywhere, so every global access is a coalesced 8/16-byte access.
#include "sea_common.h"

namespace sea {

__host__ __device__ constexpr float wino_bt(int m, int i, int k) {
  if (m == 2) {
    constexpr float T[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
    return T[i][k];
  }
  constexpr float T[6][6] = {{4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                             {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  return T[i][k];
}
__host__ __device__ constexpr float wino_g(int m, int i, int k) {
  if (m == 2) {
    constexpr float T[4][3] = {{1, 0, 0}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0, 0, 1}};
    return T[i][k];
  }
  constexpr float T[6][3] = {{1.f / 4, 0, 0},           {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                             {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6},  {0, 0, 1}};
  return T[i][k];
}
__host__ __device__ constexpr float wino_at(int m, int i, int k) {
  if (m == 2) {
    constexpr float T[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
    return T[i][k];
  }
  constexpr float T[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
  return T[i][k];
}

// acc += c * v with the compile-time coefficient folded (0 skipped, +-1 as add / sub)
template <int VEC>
__device__ __forceinline__ void axpy(float c, const float (&v)[VEC], float (&acc)[VEC], bool& first) {
  if (c == 0.f) return;
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = first ? c * v[e] : fmaf(c, v[e], acc[e]);
  first = false;
}

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<2> {
  using type = float2;
};

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float (&v)[VEC]) {
  const typename VecT<VEC>::type r = *(const typename VecT<VEC>::type*)p;
  const float* rp = (const float*)&r;
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] = rp[e];
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&v)[VEC]) {
  typename VecT<VEC>::type r;
  float* rp = (float*)&r;
#pragma unroll
  for (int e = 0; e < VEC; ++e) rp[e] = v[e];
  *(typename VecT<VEC>::type*)p = r;
}

// ---- input transform: V[k][t][c] = (B^T d B)[k], d = the (m+2)^2 patch of tile t (zero outside the image) ----
// Optional prologue (the backward of a fused scale/shift + ReLU epilogue): d = gate > 0 ? x * scale[c] : 0.
template <int M, int VEC>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                         const float* __restrict__ scale, float* __restrict__ V, int C,
                                                         int H, int W, int nTh, int nTw, int64_t T, int64_t total,
                                                         int64_t xps, int64_t vts, int xcd, Divs3 dv,
                                                         uint32_t* __restrict__ amax_out) {
  constexpr int A = M + 2;
  const bool fast = total < kFastIndexLimit;
  // amax_out != NULL: amax_out[t] receives the float bits of max |V[.][t][.]| of tile t (one word per row of the Winograd-
  // domain GEMMs = the row's fp16 x 2 scale; a tile's scale depends on that tile only)
  // plain order by default: the transform writes 2.25x what it reads, and the XCD-contiguous order measured 4-9 %
  // slower (profiles/r2_xcd_order_ab.log); SEA_XCD_ORDER=2 turns it on for this kernel
  const IndexRange rg = xcd_range(total, xcd);
  for (int64_t idx = rg.begin; idx < rg.end; idx += rg.stride) {
    const Index4 ix = split_index(idx, dv, fast);  // (cg, tx, ty, b)
    const int cg = ix.c0, tx = ix.c1, ty = ix.c2;
    const int b = (int)ix.c3;
    const int64_t t = ((int64_t)b * nTh + ty) * nTw + tx;
    const int y0 = ty * M - 1, x0 = tx * M - 1;
    const int64_t base = (int64_t)b * H * W * C + (int64_t)cg * VEC;
    const float* xb = x + (int64_t)b * H * W * xps + (int64_t)cg * VEC;  // x may be a channel slice: pixel stride xps
    float sc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) sc[e] = scale ? scale[cg * VEC + e] : 1.f;
    float d[A][A][VEC];
#pragma unroll
    for (int i = 0; i < A; ++i) {
      const int yy = y0 + i;
#pragma unroll
      for (int j = 0; j < A; ++j) {
        const int xx = x0 + j;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          vload<VEC>(xb + ((int64_t)yy * W + xx) * xps, d[i][j]);
          if (gate) {
            float gt[VEC];
            vload<VEC>(gate + base + ((int64_t)yy * W + xx) * C, gt);
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[i][j][e] = gt[e] > 0.f ? d[i][j][e] * sc[e] : 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) d[i][j][e] = 0.f;
        }
      }
    }
    // rows: tmp = B^T d
    float tmp[A][A][VEC];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy<VEC>(wino_bt(M, i, k), d[k][j], tmp[i][j], first);
      }
    // columns: v = tmp B, streamed out as soon as each element is ready
    float* vb = V + t * vts + (int64_t)cg * VEC;  // V may be a channel slice of a wider (A*A, T, vts) tensor
    uint32_t vmax = 0;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        float v[VEC];
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy<VEC>(wino_bt(M, j, k), tmp[i][k], v, first);
        vstore<VEC>(vb + (int64_t)(i * A + j) * T * vts, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const uint32_t b = __float_as_uint(v[e]) & 0x7fffffffu;
          vmax = b > vmax ? b : vmax;
        }
      }
    if (amax_out != nullptr) {   // kernel-uniform
      // the 64 lanes of a wave are 64 consecutive channel groups: one tile when C / VEC is a multiple of 64 (every layer of
      // the UperNet head) -> one atomic per wave; otherwise (or in a partially active wave) one per lane
      const int ti = (int)t;
      const bool whole = __ballot(1) == ~0ull && __all(ti == __shfl(ti, 0, 64));
      if (whole) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t other = (uint32_t)__shfl_xor((int)vmax, o, 64);
          vmax = other > vmax ? other : vmax;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(amax_out + t, vmax);
      } else {
        atomicMax(amax_out + t, vmax);
      }
    }
  }
}

// The same transform with FOUR channels per lane (16-byte accesses: a wave-load covers 1 KB instead of 512 B of the vector
// memory path, which bounds this kernel -- most of all with the gate prologue, whose second input doubles the loads).  The
// patch is consumed column by column (six loads, then that column of tmp = B^T d) so that only tmp (36 x 4 registers) stays
// live; same axpy order per element as above: same bits.
template <int M, int VEC>
__global__ __launch_bounds__(256) void wino_input_cols_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                              const float* __restrict__ scale, float* __restrict__ V, int C,
                                                              int H, int W, int nTh, int nTw, int64_t T, int64_t total,
                                                              int64_t xps, int64_t vts, int xcd, Divs3 dv,
                                                              uint32_t* __restrict__ amax_out) {
  constexpr int A = M + 2;
  const bool fast = total < kFastIndexLimit;
  const IndexRange rg = xcd_range(total, xcd);
  for (int64_t idx = rg.begin; idx < rg.end; idx += rg.stride) {
    const Index4 ix = split_index(idx, dv, fast);  // (cg, tx, ty, b)
    const int cg = ix.c0, tx = ix.c1, ty = ix.c2;
    const int b = (int)ix.c3;
    const int64_t t = ((int64_t)b * nTh + ty) * nTw + tx;
    const int y0 = ty * M - 1, x0 = tx * M - 1;
    const int64_t base = (int64_t)b * H * W * C + (int64_t)cg * VEC;
    const float* xb = x + (int64_t)b * H * W * xps + (int64_t)cg * VEC;
    float sc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) sc[e] = scale ? scale[cg * VEC + e] : 1.f;
    float tmp[A][A][VEC];
#pragma unroll
    for (int j = 0; j < A; ++j) {
      const int xx = x0 + j;
      float col[A][VEC];
#pragma unroll
      for (int i = 0; i < A; ++i) {
        const int yy = y0 + i;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          vload<VEC>(xb + ((int64_t)yy * W + xx) * xps, col[i]);
          if (gate) {
            float gt[VEC];
            vload<VEC>(gate + base + ((int64_t)yy * W + xx) * C, gt);
#pragma unroll
            for (int e = 0; e < VEC; ++e) col[i][e] = gt[e] > 0.f ? col[i][e] * sc[e] : 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) col[i][e] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < A; ++i) {
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy<VEC>(wino_bt(M, i, k), col[k], tmp[i][j], first);
      }
    }
    float* vb = V + t * vts + (int64_t)cg * VEC;
    uint32_t vmax = 0;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        float v[VEC];
        bool first = true;
#pragma unroll
        for (int k = 0; k < A; ++k) axpy<VEC>(wino_bt(M, j, k), tmp[i][k], v, first);
        vstore<VEC>(vb + (int64_t)(i * A + j) * T * vts, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const uint32_t b = __float_as_uint(v[e]) & 0x7fffffffu;
          vmax = b > vmax ? b : vmax;
        }
      }
    if (amax_out != nullptr) {   // kernel-uniform
      const int ti = (int)t;
      const bool whole = __ballot(1) == ~0ull && __all(ti == __shfl(ti, 0, 64));
      if (whole) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t other = (uint32_t)__shfl_xor((int)vmax, o, 64);
          vmax = other > vmax ? other : vmax;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(amax_out + t, vmax);
      } else {
        atomicMax(amax_out + t, vmax);
      }
    }
  }
}

// ---- output transform: y tile = act(scale[c] * (A^T m A) + bias[c]), m[k] = M[k][t][c]; act = ReLU or identity ---
template <int M, int VEC>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mx, const float* __restrict__ addend,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ bias, int relu,
                                                          float* __restrict__ y, int C, int H, int W, int nTh, int nTw,
                                                          int64_t T, int64_t total, Divs3 dv) {
  constexpr int A = M + 2;
  const bool fast = total < kFastIndexLimit;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const Index4 ix = split_index(idx, dv, fast);  // (cg, tx, ty, b)
    const int cg = ix.c0, tx = ix.c1, ty = ix.c2;
    const int b = (int)ix.c3;
    const int64_t t = ((int64_t)b * nTh + ty) * nTw + tx;
    const float* mb = Mx + t * C + (int64_t)cg * VEC;
    // rows first, one Winograd-domain row at a time: tmp[p][j] = sum_i At[p][i] m[i][j]
    float tmp[M][A][VEC];
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float col[A][VEC];
#pragma unroll
      for (int i = 0; i < A; ++i) vload<VEC>(mb + (int64_t)(i * A + j) * T * C, col[i]);
#pragma unroll
      for (int p = 0; p < M; ++p) {
        bool first = true;
#pragma unroll
        for (int i = 0; i < A; ++i) axpy<VEC>(wino_at(M, p, i), col[i], tmp[p][j], first);
      }
    }
    float bv[VEC], sv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      bv[e] = bias ? bias[cg * VEC + e] : 0.f;
      sv[e] = scale ? scale[cg * VEC + e] : 1.f;
    }
    float* yb = y + (int64_t)b * H * W * C + (int64_t)cg * VEC;
#pragma unroll
    for (int p = 0; p < M; ++p) {
      const int yy = ty * M + p;
#pragma unroll
      for (int q = 0; q < M; ++q) {
        const int xx = tx * M + q;
        float o[VEC];
        bool first = true;
#pragma unroll
        for (int j = 0; j < A; ++j) axpy<VEC>(wino_at(M, q, j), tmp[p][j], o, first);
        if (addend && yy < H && xx < W) {  // contribution of the coarse inputs (M6), before scale / shift
          float ad[VEC];
          vload<VEC>(addend + (int64_t)b * H * W * C + ((int64_t)yy * W + xx) * C + (int64_t)cg * VEC, ad);
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] += ad[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          o[e] = scale ? fmaf(o[e], sv[e], bv[e]) : o[e] + bv[e];
          if (relu) o[e] = fmaxf(o[e], 0.f);
        }
        if (yy < H && xx < W) vstore<VEC>(yb + ((int64_t)yy * W + xx) * C, o);
      }
    }
  }
}

// ---- filter transform: U[k] = (G g G^T)[k]; w is (Cout, Cin, 3, 3) contiguous ----------------------------------
// flip = 0: U (A*A, Cin, Cout) for the forward convolution
// flip = 1: U (A*A, Cout, Cin) of the 180-degree rotated filters: the input-gradient convolution
template <int M>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                          int Cin, int flip) {
  constexpr int A = M + 2;
  const int64_t n = (int64_t)Cout * Cin;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cin);
    const int o = (int)(idx / Cin);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int bq = 0; bq < 3; ++bq) g[a][bq] = w[idx * 9 + (flip ? (2 - a) * 3 + (2 - bq) : a * 3 + bq)];
    float tmp[A][3];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (wino_g(M, i, k) != 0.f) s = fmaf(wino_g(M, i, k), g[k][j], s);
        tmp[i][j] = s;
      }
    const int64_t off = flip ? (int64_t)o * Cin + c : (int64_t)c * Cout + o;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (wino_g(M, j, k) != 0.f) s = fmaf(wino_g(M, j, k), tmp[i][k], s);
        U[(int64_t)(i * A + j) * n + off] = s;
      }
  }
}

}  // namespace sea

using namespace sea;

// A/B (env SEA_WINO_IN_VEC4, read per call): 1 = the F(4,3) input transform with four channels per lane
static inline int wino_in_vec4() {
  const char* e = getenv("SEA_WINO_IN_VEC4");
  return e ? atoi(e) : 1;
}

static bool wino_dims(int B, int C, int H, int W, int m, int vec, int* nTh, int* nTw, int64_t* T) {
  if (!(B > 0 && C > 0 && H > 0 && W > 0 && (m == 2 || m == 4) && (C % vec) == 0)) return false;
  *nTh = (H + m - 1) / m;
  *nTw = (W + m - 1) / m;
  *T = (int64_t)B * *nTh * *nTw;
  return true;
}

extern "C" int64_t sea_wino_tiles(int B, int H, int W, int m) {
  if (!(B > 0 && H > 0 && W > 0 && (m == 2 || m == 4))) return 0;
  return (int64_t)B * ((H + m - 1) / m) * ((W + m - 1) / m);
}

static int wino_input_impl(const float* x, int64_t x_pixel_stride, const float* gate, const float* scale, float* V,
                           int64_t v_tile_stride, int B, int C, int H, int W, int m, uint32_t* amax_out, void* stream);

extern "C" int sea_wino_input_transform(const float* x, int64_t x_pixel_stride, const float* gate, const float* scale,
                                        float* V, int64_t v_tile_stride, int B, int C, int H, int W, int m,
                                        void* stream) {
  return wino_input_impl(x, x_pixel_stride, gate, scale, V, v_tile_stride, B, C, H, W, m, nullptr, stream);
}

// same, and the float bits of max |V| are max-accumulated into *amax_out (a pre-zeroed device word; several calls that fill
// channel slices of one V may share it): the activation scale of sea_gemm_split_f16 without another pass over V
extern "C" int sea_wino_input_transform_amax(const float* x, int64_t x_pixel_stride, const float* gate, const float* scale,
                                             float* V, int64_t v_tile_stride, int B, int C, int H, int W, int m,
                                             uint32_t* amax_out, void* stream) {
  SEA_CHECK_ARG(amax_out != nullptr);
  return wino_input_impl(x, x_pixel_stride, gate, scale, V, v_tile_stride, B, C, H, W, m, amax_out, stream);
}

static int wino_input_impl(const float* x, int64_t x_pixel_stride, const float* gate, const float* scale, float* V,
                           int64_t v_tile_stride, int B, int C, int H, int W, int m, uint32_t* amax_out, void* stream) {
  int nTh, nTw;
  int64_t T;
  SEA_CHECK_ARG(x && V && wino_dims(B, C, H, W, m, 4, &nTh, &nTw, &T));
  SEA_CHECK_ARG(x_pixel_stride >= C && (x_pixel_stride % 4) == 0);
  SEA_CHECK_ARG(v_tile_stride >= C && (v_tile_stride % 4) == 0);
  SEA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)V) | ((uintptr_t)gate)) & 15) == 0);
  if (m == 2) {
    const int64_t total = T * (C / 4);
    hipLaunchKernelGGL((wino_input_kernel<2, 4>), dim3(grid_for_xcd(total, 256)), dim3(256), 0, (hipStream_t)stream, x, gate,
                       scale, V, C, H, W, nTh, nTw, T, total, x_pixel_stride, v_tile_stride, xcd_order_enabled() == 2, divs3(C / 4, nTw, nTh),
                       amax_out);
  } else if (wino_in_vec4() & 1) {
    const int64_t total = T * (C / 4);
    hipLaunchKernelGGL((wino_input_cols_kernel<4, 4>), dim3(grid_for_xcd(total, 256)), dim3(256), 0, (hipStream_t)stream, x, gate,
                       scale, V, C, H, W, nTh, nTw, T, total, x_pixel_stride, v_tile_stride, xcd_order_enabled() == 2, divs3(C / 4, nTw, nTh),
                       amax_out);
  } else {
    const int64_t total = T * (C / 2);
    hipLaunchKernelGGL((wino_input_kernel<4, 2>), dim3(grid_for_xcd(total, 256)), dim3(256), 0, (hipStream_t)stream, x, gate,
                       scale, V, C, H, W, nTh, nTw, T, total, x_pixel_stride, v_tile_stride, xcd_order_enabled() == 2, divs3(C / 2, nTw, nTh),
                       amax_out);
  }
  SEA_RETURN_LAST();
}

extern "C" int sea_wino_output_transform(const float* Mx, const float* addend, const float* scale, const float* bias,
                                         int relu, float* y, int B, int C, int H, int W, int m, void* stream) {
  int nTh, nTw;
  int64_t T;
  SEA_CHECK_ARG(Mx && y && wino_dims(B, C, H, W, m, 4, &nTh, &nTw, &T));
  SEA_CHECK_ARG(((((uintptr_t)Mx) | ((uintptr_t)y) | ((uintptr_t)addend)) & 15) == 0);
  if (m == 2) {
    const int64_t total = T * (C / 4);
    hipLaunchKernelGGL((wino_output_kernel<2, 4>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, Mx,
                       addend, scale, bias, relu, y, C, H, W, nTh, nTw, T, total, divs3(C / 4, nTw, nTh));
  } else if (wino_in_vec4() & 2) {     // (A/B bit 2: four channels per lane in the output transform too)
    const int64_t total = T * (C / 4);
    hipLaunchKernelGGL((wino_output_kernel<4, 4>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, Mx,
                       addend, scale, bias, relu, y, C, H, W, nTh, nTw, T, total, divs3(C / 4, nTw, nTh));
  } else {
    const int64_t total = T * (C / 2);
    hipLaunchKernelGGL((wino_output_kernel<4, 2>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, Mx,
                       addend, scale, bias, relu, y, C, H, W, nTh, nTw, T, total, divs3(C / 2, nTw, nTh));
  }
  SEA_RETURN_LAST();
}

extern "C" int sea_wino_filter_transform(const float* w, float* U, int Cout, int Cin, int m, int flip, void* stream) {
  SEA_CHECK_ARG(w && U && Cout > 0 && Cin > 0 && (m == 2 || m == 4));
  const int64_t n = (int64_t)Cout * Cin;
  if (m == 2)
    hipLaunchKernelGGL((wino_filter_kernel<2>), dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, w, U, Cout,
                       Cin, flip);
  else
    hipLaunchKernelGGL((wino_filter_kernel<4>), dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, w, U, Cout,
                       Cin, flip);
  SEA_RETURN_LAST();
}

// M2 (model side): bilinear up-sampling, align_corners=False, fp32 NCHW planes - forward and backward.
//
// UperNet's FPN up-samples 512-channel maps three times per forward (and again top-down), plus the final
// logits (semseg/models/uperforseg.py:236-262, 416-418).  ATen's upsample_bilinear2d kernels run these at
// ~0.3 TB/s (0.9 ms for a 268 MB output; ~8 % of an APGD step forward+backward) although the op is a pure
// HBM stream: write the output once (forward) / read the output gradient once (backward).
//
// Forward: one lane writes 4 consecutive output pixels (float4 store); its <= 2x5 input values come from
// L1/L2 (the input is s^2 times smaller than the output).
// Backward: gather, deterministic (ATen scatters with atomicAdd): a workgroup owns a TIxTI tile of INPUT
// pixels of one plane, stages the output-gradient region that touches it in LDS with coalesced row
// reads, and every input pixel sums its footprint in a fixed order, cell by cell (same bookkeeping as
// loss_upsampled.hip).
// Source-index rule = ATen: src = r*(dst+0.5)-0.5 clamped at 0, i0=floor(src), i1=min(i0+1,n-1).
#include "sea_common.h"
#include "bilinear_map.h"

namespace sea {

// grid = (ceil(W/64), ceil(H/16), planes); block 256 = 16 rows x 16 strips of 4 pixels
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int h,
                                                           int w, int H, int W, float rh, float rw) {
  const int plane = blockIdx.z;
  const int Y = blockIdx.y * 16 + (threadIdx.x >> 4);
  const int X0 = (blockIdx.x * 16 + (threadIdx.x & 15)) * 4;
  if (Y >= H || X0 >= W) return;
  const float* xp = x + (int64_t)plane * h * w;
  const AxisMapU my = axis_map_u(Y, rh, h);
  const float* r0 = xp + (int64_t)my.i0 * w;
  const float* r1 = xp + (int64_t)my.i1 * w;
  float out[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int X = min(X0 + j, W - 1);
    const AxisMapU mx = axis_map_u(X, rw, w);
    const float top = (1.f - mx.lam) * r0[mx.i0] + mx.lam * r0[mx.i1];
    const float bot = (1.f - mx.lam) * r1[mx.i0] + mx.lam * r1[mx.i1];
    out[j] = (1.f - my.lam) * top + my.lam * bot;
  }
  float* yp = y + ((int64_t)plane * H + Y) * W + X0;
  if (X0 + 3 < W && ((((uintptr_t)yp) & 15) == 0)) {
    *reinterpret_cast<float4*>(yp) = make_float4(out[0], out[1], out[2], out[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (X0 + j < W) yp[j] = out[j];
  }
}

// grid = (tiles_x, tiles_y, planes); dynamic LDS: go region [RMAX][RLD] + axis tables + cell tables + row sums
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int h,
                                                           int w, int H, int W, float rh, float rw, int TI, int RMAX) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int RLD = RMAX + 1;
  float* reg = smem;                         // RMAX * RLD
  int* r_i1 = (int*)(reg + RMAX * RLD);      // RMAX (local index of the bottom source row)
  float* r_lam = (float*)(r_i1 + RMAX);
  int* c_i1 = (int*)(r_lam + RMAX);
  float* c_lam = (float*)(c_i1 + RMAX);
  int* rbeg = (int*)(c_lam + RMAX);          // TI + 3
  int* cbeg = rbeg + (TI + 3);
  float* tmp = (float*)(cbeg + (TI + 3));    // RMAX * (TI + 1): row-reduced partial sums
  const int plane = blockIdx.z;
  const int ya = blockIdx.y * TI, xa = blockIdx.x * TI;
  const int yb = min(ya + TI, h), xb = min(xa + TI, w);
  // cell boundaries first (TI+3 lanes do the float work once), region bounds are read back from them
  if (threadIdx.x < TI + 3) {
    const int t = threadIdx.x;
    rbeg[t] = first_dst_ge(min(ya - 1 + t, yb), rh, h, H);
    cbeg[t] = first_dst_ge(min(xa - 1 + t, xb), rw, w, W);
  }
  __syncthreads();
  const int Y0 = rbeg[0], Y1 = rbeg[TI + 2], X0 = cbeg[0], X1 = cbeg[TI + 2];
  const int RH = Y1 - Y0, RW = X1 - X0;
  const float* gp = gy + (int64_t)plane * H * W;
  // stage the region: 8 independent loads in flight per lane before the first LDS write (a plain
  // load->store loop serialises on the HBM latency: measured 10x slower than the roofline)
  {
    const int total = RH * RW;
    for (int base = 0; base < total; base += 256 * 8) {
      float v[8];
      int dst[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + k * 256 + threadIdx.x;
        const int ri = idx / RW, ci = idx - ri * RW;
        dst[k] = idx < total ? ri * RLD + ci : -1;
        v[k] = idx < total ? gp[(int64_t)(Y0 + ri) * W + X0 + ci] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (dst[k] >= 0) reg[dst[k]] = v[k];
    }
  }
  for (int i = threadIdx.x; i < RH; i += 256) {
    const AxisMapU m = axis_map_u(Y0 + i, rh, h);
    r_i1[i] = m.i1 - (ya - 1);
    r_lam[i] = m.lam;
  }
  for (int i = threadIdx.x; i < RW; i += 256) {
    const AxisMapU m = axis_map_u(X0 + i, rw, w);
    c_i1[i] = m.i1 - (xa - 1);
    c_lam[i] = m.lam;
  }
  __syncthreads();
  const int nly = yb - ya, nlx = xb - xa;
  // separable gather: first along x (per region row), then along y
  const int TLD = TI + 1;
  for (int item = threadIdx.x; item < RH * nlx; item += 256) {
    const int ri = item / nlx, tx = item - ri * nlx;
    const int xl = tx + 1;
    float rowacc = 0.f;
    for (int qx = 0; qx < 2; ++qx) {
      const int cx = xl - 1 + qx;
      const int j_lo = cbeg[cx] - X0, j_hi = cbeg[cx + 1] - X0;
      if (j_lo >= j_hi) continue;
      const int cx1 = c_i1[j_lo];
      for (int ci = j_lo; ci < j_hi; ++ci) {
        const float lx = c_lam[ci];
        const float wx = ((cx == xl) ? (1.f - lx) : 0.f) + ((cx1 == xl) ? lx : 0.f);
        rowacc = fmaf(wx, reg[ri * RLD + ci], rowacc);
      }
    }
    tmp[ri * TLD + tx] = rowacc;
  }
  __syncthreads();
  float* op = gx + (int64_t)plane * h * w;
  for (int item = threadIdx.x; item < nly * nlx; item += 256) {
    const int ty = item / nlx, tx = item - ty * nlx;
    const int yl = ty + 1;
    float acc = 0.f;
    for (int qy = 0; qy < 2; ++qy) {
      const int cy = yl - 1 + qy;
      const int i_lo = rbeg[cy] - Y0, i_hi = rbeg[cy + 1] - Y0;
      if (i_lo >= i_hi) continue;
      const int cy1 = r_i1[i_lo];
      for (int ri = i_lo; ri < i_hi; ++ri) {
        const float ly = r_lam[ri];
        const float wy = ((cy == yl) ? (1.f - ly) : 0.f) + ((cy1 == yl) ? ly : 0.f);
        acc = fmaf(wy, tmp[ri * TLD + tx], acc);
      }
    }
    op[(int64_t)(ya + ty) * w + (xa + tx)] = acc;
  }
}

// ---- power-of-two integer factors (the final logit up-sampling: x4 UperNet, x16 Segmenter) -------------------------
// With H = S*h, W = S*w and S a power of two, ATen's source index r*(dst+0.5)-0.5 (r = 1/S) is exact in float:
// with t = dst + S/2,  i0 = (t >> log2 S) - 1 (clamped to 0 with lambda = 0 when negative, exactly what the clamp of
// src does),  lambda = ((t & (S-1)) + 0.5) / S.  No float index arithmetic, no search for footprint boundaries, no LDS.
template <int N>
struct FVec;
template <>
struct FVec<1> {
  using type = float;
};
template <>
struct FVec<2> {
  using type = float2;
};
template <>
struct FVec<4> {
  using type = float4;
};

template <int S>
struct Pow2 {
  static constexpr int LOG = S == 2 ? 1 : S == 4 ? 2 : S == 8 ? 3 : 4;
  static_assert(S == 2 || S == 4 || S == 8 || S == 16, "supported factors");
};

// One lane = one float4 of the output.  Groups of G = min(4, S/2) consecutive pixels share their source columns:
// 4 loads per group (16 for x2, 8 for x4, 4 for x8 / x16) instead of 16 per float4; values identical to
// upsample_fwd_kernel bit for bit (same expression tree).
template <int S>
__device__ __forceinline__ void upsample_fwd_pow2_item(const float* __restrict__ x, float* __restrict__ y, int h, int w,
                                                       int x4, int Y, int64_t plane, int64_t i) {
  constexpr int LOG = Pow2<S>::LOG;
  constexpr int G = (S / 2 >= 4) ? 4 : S / 2;
  constexpr float inv = 1.f / (float)S;
  const int ty = Y + S / 2;
  int r0i = (ty >> LOG) - 1;
  float ly = ((float)(ty & (S - 1)) + 0.5f) * inv;
  if (r0i < 0) {
    r0i = 0;
    ly = 0.f;
  }
  const int r1i = min(r0i + 1, h - 1);
  const float* r0 = x + (plane * h + r0i) * w;
  const float* r1 = x + (plane * h + r1i) * w;
  float out[4];
#pragma unroll
  for (int g = 0; g < 4 / G; ++g) {
    const int tx = x4 * 4 + g * G + S / 2;
    int c0 = (tx >> LOG) - 1;
    const bool clamped = c0 < 0;
    c0 = clamped ? 0 : c0;
    const int c1 = min(c0 + 1, w - 1);
    const float a0 = r0[c0], a1 = r0[c1], b0 = r1[c0], b1 = r1[c1];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const float lx = clamped ? 0.f : ((float)((tx + j) & (S - 1)) + 0.5f) * inv;
      const float top = (1.f - lx) * a0 + lx * a1;
      const float bot = (1.f - lx) * b0 + lx * b1;
      out[g * G + j] = (1.f - ly) * top + ly * bot;
    }
  }
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 o;
  o.x = out[0];
  o.y = out[1];
  o.z = out[2];
  o.w = out[3];
  __builtin_nontemporal_store(o, reinterpret_cast<f4*>(y + i * 4));  // written once, read by the next kernel from HBM
}

template <int S>
__global__ __launch_bounds__(256) void upsample_fwd_pow2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int h, int w, int64_t total, int xcd, FastDiv fW4,
                                                                FastDiv fH) {
  const int H = h * S, W4 = (w * S) >> 2;
  const IndexRange rg = xcd_range(total, xcd);
  if (total < kFastIndexLimit) {
    const uint32_t end = (uint32_t)rg.end, stride = (uint32_t)rg.stride;
    for (uint32_t i = (uint32_t)rg.begin; i < end; i += stride) {
      const uint32_t r = fdiv(i, fW4), plane = fdiv(r, fH);
      upsample_fwd_pow2_item<S>(x, y, h, w, (int)(i - r * W4), (int)(r - plane * H), plane, i);
    }
  } else {
    for (int64_t i = rg.begin; i < rg.end; i += rg.stride) {
      const int64_t r = i / W4;
      upsample_fwd_pow2_item<S>(x, y, h, w, (int)(i % W4), (int)(r % H), r / H, i);
    }
  }
}

// Cell variant of the forward (the default): one lane = one float4 column of one "cell row" c in [-1, h-1], i.e. the S
// output rows Y = S*c + S/2 + k that interpolate between input rows c and c+1 (c = -1 / h-1: the clamped half cells).
// The <= 8 source values and the 8 horizontal interpolants are computed once and feed S float4 stores (x4: 2 loads per
// store instead of 8, x16: 0.25 instead of 4); every store instruction is still a coalesced row segment.  Same
// expression tree as upsample_fwd_kernel: identical values.
template <int S>
__device__ __forceinline__ void upsample_fwd_cell_item(const float* __restrict__ x, float* __restrict__ y, int h, int w,
                                                       int x4, int c, int64_t plane, bool nt) {
  constexpr int LOG = Pow2<S>::LOG;
  constexpr int G = (S / 2 >= 4) ? 4 : S / 2;
  constexpr float inv = 1.f / (float)S;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int H = h * S, W = w * S;
  const int r0i = c < 0 ? 0 : c, r1i = min(r0i + 1, h - 1);
  const float* r0 = x + (plane * h + r0i) * w;
  const float* r1 = x + (plane * h + r1i) * w;
  float top[4], bot[4];
#pragma unroll
  for (int g = 0; g < 4 / G; ++g) {
    const int tx = x4 * 4 + g * G + S / 2;
    int c0 = (tx >> LOG) - 1;
    const bool clamped = c0 < 0;
    c0 = clamped ? 0 : c0;
    const int c1 = min(c0 + 1, w - 1);
    const float a0 = r0[c0], a1 = r0[c1], b0 = r1[c0], b1 = r1[c1];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const float lx = clamped ? 0.f : ((float)((tx + j) & (S - 1)) + 0.5f) * inv;
      top[g * G + j] = (1.f - lx) * a0 + lx * a1;
      bot[g * G + j] = (1.f - lx) * b0 + lx * b1;
    }
  }
  float* yp = y + plane * H * W + x4 * 4;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    const int Y = S * c + S / 2 + k;
    if (Y < 0 || Y >= H) continue;
    const float ly = c < 0 ? 0.f : ((float)k + 0.5f) * inv;
    f4 o;
    o.x = (1.f - ly) * top[0] + ly * bot[0];
    o.y = (1.f - ly) * top[1] + ly * bot[1];
    o.z = (1.f - ly) * top[2] + ly * bot[2];
    o.w = (1.f - ly) * top[3] + ly * bot[3];
    // non-temporal only for outputs that cannot stay in the 256 MB Infinity Cache anyway: a 176 MB logit tensor
    // (C=21) is read back by the loss kernel right away and should still be on die then
    if (nt)
      __builtin_nontemporal_store(o, reinterpret_cast<f4*>(yp + (int64_t)Y * W));
    else
      *reinterpret_cast<f4*>(yp + (int64_t)Y * W) = o;
  }
}

template <int S>
__global__ __launch_bounds__(256) void upsample_fwd_cells_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int h, int w, int64_t total, int xcd, FastDiv fW4,
                                                                 FastDiv fC, int nt) {
  const int W4 = (w * S) >> 2, NC = h + 1;
  const IndexRange rg = xcd_range(total, xcd);
  if (total < kFastIndexLimit) {
    const uint32_t end = (uint32_t)rg.end, stride = (uint32_t)rg.stride;
    for (uint32_t i = (uint32_t)rg.begin; i < end; i += stride) {
      const uint32_t r = fdiv(i, fW4), plane = fdiv(r, fC);
      upsample_fwd_cell_item<S>(x, y, h, w, (int)(i - r * W4), (int)(r - plane * NC) - 1, plane, nt != 0);
    }
  } else {
    for (int64_t i = rg.begin; i < rg.end; i += rg.stride) {
      const int64_t r = i / W4;
      upsample_fwd_cell_item<S>(x, y, h, w, (int)(i % W4), (int)(r % NC) - 1, r / NC, nt != 0);
    }
  }
}

// Gather, deterministic: one lane = one INPUT pixel; its footprint is the 2S x 2S output window starting at
// (S*yq - S/2, S*xq - S/2) with the separable weights c[t] = (t+0.5)/S (t < S: the pixel is the right/bottom neighbour
// i1) and (2S-t-0.5)/S (t >= S: it is i0).  Edges: window parts outside the image are skipped; where the source index
// is clamped (first / last half cell) both interpolation weights fall on the edge pixel: weight 1.  Row sums first
// (ascending x), rows ascending: a fixed order.  The window start is a multiple of S/2 floats: vector loads.
template <int S>
__device__ __forceinline__ void upsample_bwd_pow2_item(const float* __restrict__ gy, float* __restrict__ gx, int h, int w,
                                                       int xq, int yq, int64_t plane, int64_t i) {
  constexpr int T = 2 * S;
  constexpr int VL = (S / 2 >= 4) ? 4 : S / 2;
  constexpr float inv = 1.f / (float)S;
  const int H = h * S, W = w * S;
  const bool eL = xq == 0, eR = xq == w - 1, eT = yq == 0, eB = yq == h - 1;
  float wx[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float c = (t < S) ? ((float)t + 0.5f) * inv : ((float)(2 * S - t) - 0.5f) * inv;
    if (t < S / 2)
      wx[t] = eL ? 0.f : c;  // outside (never loaded)
    else if (t < S)
      wx[t] = eL ? 1.f : c;  // clamped at the left edge
    else if (t < 3 * S / 2)
      wx[t] = eR ? 1.f : c;  // clamped at the right edge
    else
      wx[t] = eR ? 0.f : c;  // outside
  }
  const int X0 = S * xq - S / 2, Y0 = S * yq - S / 2;
  const float* gp = gy + plane * H * W + X0;
  float acc = 0.f;
#pragma unroll 2
  for (int ty = 0; ty < T; ++ty) {
    const int Y = Y0 + ty;
    if (Y < 0 || Y >= H) continue;
    float cy = (ty < S) ? ((float)ty + 0.5f) * inv : ((float)(2 * S - ty) - 0.5f) * inv;
    cy = ((eT && ty < S) || (eB && ty >= S)) ? 1.f : cy;
    const float* row = gp + (int64_t)Y * W;
    float v[T];
#pragma unroll
    for (int k = 0; k < T / VL; ++k) {
      const bool outside = (eL && k * VL < S / 2) || (eR && k * VL >= 3 * S / 2);
      typename FVec<VL>::type q;
      float* qf = reinterpret_cast<float*>(&q);
#pragma unroll
      for (int e = 0; e < VL; ++e) qf[e] = 0.f;
      if (!outside) q = *reinterpret_cast<const typename FVec<VL>::type*>(row + k * VL);
#pragma unroll
      for (int e = 0; e < VL; ++e) v[k * VL + e] = qf[e];
    }
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) rs = fmaf(wx[t], v[t], rs);
    acc = fmaf(cy, rs, acc);
  }
  gx[i] = acc;
}

template <int S>
__global__ __launch_bounds__(256) void upsample_bwd_pow2_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                int h, int w, int64_t total, int xcd, FastDiv fw,
                                                                FastDiv fh) {
  const IndexRange rg = xcd_range(total, xcd);
  if (total < kFastIndexLimit) {
    const uint32_t end = (uint32_t)rg.end, stride = (uint32_t)rg.stride;
    for (uint32_t i = (uint32_t)rg.begin; i < end; i += stride) {
      const uint32_t r = fdiv(i, fw), plane = fdiv(r, fh);
      upsample_bwd_pow2_item<S>(gy, gx, h, w, (int)(i - r * w), (int)(r - plane * h), plane, i);
    }
  } else {
    for (int64_t i = rg.begin; i < rg.end; i += rg.stride) {
      const int64_t r = i / w;
      upsample_bwd_pow2_item<S>(gy, gx, h, w, (int)(i % w), (int)(r % h), r / h, i);
    }
  }
}

// Row-streaming gather for the same case (the default when the output width is 128 ... 1024): the per-pixel gather
// above reads S-strided 16-byte pieces (x16: a wave-load touches 32 cache lines for 1 KB), this one reads whole output
// rows with coalesced float4 loads.  A block is (W/4 columns) x (256 / (W/4) sub-bands); a sub-band owns RP consecutive
// input rows and walks the RP + 1 "cells" of S output rows that touch them (cell c: rows with source pair (c, c+1),
// i.e. Y in [S*c + S/2, S*c + 3S/2); cell -1 / h-1 are the clamped half cells whose whole weight falls on the edge row).
// Per cell every lane adds its float4 of each row into two running sums: (1-lambda) for input row c (complete after
// this cell), lambda for row c+1.  The finished row goes to LDS (double buffered: one barrier per cell) and the first
// w lanes of the sub-band reduce it horizontally with the same separable weights.  Rows ascending, then columns
// ascending: a fixed order, no atomics.
template <int S>
__global__ __launch_bounds__(256) void upsample_bwd_rows_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                int h, int w, int lcols, int RP, int bands, int units,
                                                                int per_xcd) {
  __shared__ __attribute__((aligned(16))) float V[2][1024];
  constexpr float inv = 1.f / (float)S;
  const int u = per_xcd ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (u >= units) return;  // whole block
  const int W = w * S, H = h * S, cols = 1 << lcols;
  const int sub = threadIdx.x >> lcols, col = threadIdx.x & (cols - 1), nsub = 256 >> lcols;
  const int plane = u / bands, band = u - plane * bands;
  const int ra = (band * nsub + sub) * RP, rb = min(ra + RP, h);
  const float* gp = gy + (int64_t)plane * H * W + col * 4;
  float* op = gx + (int64_t)plane * h * w;
  float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
  for (int ci = 0; ci <= RP; ++ci) {
    const int c = ra - 1 + ci;
    const bool active = ra < h && c < rb;  // cells ra-1 .. rb-1 (none when the sub-band starts below the image)
    const bool to_c = active && c >= ra, to_n = active && c + 1 < rb;
    if (active) {
      const float wa = (c == h - 1) ? 1.f : 0.f, wb = (c == -1) ? 1.f : 0.f;  // clamped half cells: weight 1
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const int Y = S * c + S / 2 + k;
        if (Y >= 0 && Y < H) {
          const float4 v = *reinterpret_cast<const float4*>(gp + (int64_t)Y * W);
          const float lam = ((float)k + 0.5f) * inv;
          const float a = wa != 0.f ? 1.f : 1.f - lam, b = wb != 0.f ? 1.f : lam;
          if (to_c) {
            cur.x = fmaf(a, v.x, cur.x);
            cur.y = fmaf(a, v.y, cur.y);
            cur.z = fmaf(a, v.z, cur.z);
            cur.w = fmaf(a, v.w, cur.w);
          }
          if (to_n) {
            nxt.x = fmaf(b, v.x, nxt.x);
            nxt.y = fmaf(b, v.y, nxt.y);
            nxt.z = fmaf(b, v.z, nxt.z);
            nxt.w = fmaf(b, v.w, nxt.w);
          }
        }
      }
    }
    float* vr = &V[ci & 1][sub * W];
    if (to_c) *reinterpret_cast<float4*>(vr + col * 4) = cur;
    __syncthreads();
    if (to_c) {
      for (int xq = col; xq < w; xq += cols) {
        const int X0 = S * xq - S / 2;
        const bool eL = xq == 0, eR = xq == w - 1;
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * S; ++t) {
          const int X = X0 + t;
          const float cw = (t < S) ? ((float)t + 0.5f) * inv : ((float)(2 * S - t) - 0.5f) * inv;
          const float wgt = ((eL && t < S) || (eR && t >= S)) ? 1.f : cw;
          if (X >= 0 && X < W) rs = fmaf(wgt, vr[X], rs);
        }
        op[(int64_t)c * w + xq] = rs;
      }
    }
    cur = nxt;
    nxt = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// integer power-of-two factor in both axes (2, 4, 8, 16), 0 otherwise
static inline int pow2_factor(int h, int w, int H, int W) {
  for (int S = 2; S <= 16; S *= 2)
    if ((int64_t)h * S == H && (int64_t)w * S == W) return S;
  return 0;
}

static inline int upsample_general_only() {
  static const int on = [] {
    const char* e = getenv("SEA_UPSAMPLE_GENERAL");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return on;
}

// ---- channels_last variants: x (B,h,w,C), y (B,H,W,C), C % 4 == 0 ------------------------------------------
// Lanes run along the channel dimension (16-byte accesses, perfectly coalesced); no LDS.  The UperNet head
// is channels_last end to end on ROCm (MIOpen's NHWC igemm kernels return that layout), so these variants
// remove the layout copies around every up-sampling.
// S > 0: power-of-two factor in both axes, integer source map (identical values, no float index arithmetic)
template <int S>
__device__ __forceinline__ AxisMapU nhwc_axis(int dst, float r, int n_in) {
  if constexpr (S == 0) {
    return axis_map_u(dst, r, n_in);
  } else {
    constexpr int LOG = Pow2<S>::LOG;
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

template <int S>
__global__ __launch_bounds__(256) void upsample_nhwc_fwd_kernel(const float4* __restrict__ x,
                                                                const float4* __restrict__ res, float4* __restrict__ y,
                                                                int CG, int h, int w, int H, int W, float rh, float rw,
                                                                int64_t total, int64_t ypg, int xcd, Divs3 dv) {
  const IndexRange rg = xcd_range(total, xcd);
  const bool fast = total < kFastIndexLimit;
  for (int64_t i = rg.begin; i < rg.end; i += rg.stride) {
    const Index4 ix = split_index(i, dv, fast);  // (cg, X, Y, b)
    const int cg = ix.c0, X = ix.c1, Y = ix.c2;
    const int b = (int)ix.c3;
    const int64_t opix = ((int64_t)b * H + Y) * W + X;
    const AxisMapU my = nhwc_axis<S>(Y, rh, h), mx = nhwc_axis<S>(X, rw, w);
    const float4* xb = x + (int64_t)b * h * w * CG + cg;
    const float4 v00 = xb[((int64_t)my.i0 * w + mx.i0) * CG], v01 = xb[((int64_t)my.i0 * w + mx.i1) * CG];
    const float4 v10 = xb[((int64_t)my.i1 * w + mx.i0) * CG], v11 = xb[((int64_t)my.i1 * w + mx.i1) * CG];
    const float lx = mx.lam, ly = my.lam, ux = 1.f - lx, uy = 1.f - ly;
    float4 o;
    o.x = uy * (ux * v00.x + lx * v01.x) + ly * (ux * v10.x + lx * v11.x);
    o.y = uy * (ux * v00.y + lx * v01.y) + ly * (ux * v10.y + lx * v11.y);
    o.z = uy * (ux * v00.z + lx * v01.z) + ly * (ux * v10.z + lx * v11.z);
    o.w = uy * (ux * v00.w + lx * v01.w) + ly * (ux * v10.w + lx * v11.w);
    if (res) {  // fused top-down add of the FPN: y = residual + up(x), residual dense (B,H,W,C)
      const float4 r = res[i];
      o.x += r.x;
      o.y += r.y;
      o.z += r.z;
      o.w += r.w;
    }
    y[opix * ypg + cg] = o;
  }
}

// Cell variant for power-of-two factors (the default there): one lane = 4 channels of one cell (cy, cx) in
// [-1, h-1] x [-1, w-1]: the S x S output pixels that interpolate between the same four source pixels.  4 loads and 2S
// horizontal interpolants feed up to S*S coalesced 16-byte stores (the per-pixel kernel spends 4 loads and a full index
// decomposition per store).  Same expression tree: identical values.
template <int S>
__global__ __launch_bounds__(256) void upsample_nhwc_fwd_cells_kernel(const float4* __restrict__ x,
                                                                      const float4* __restrict__ res,
                                                                      float4* __restrict__ y, int CG, int h, int w,
                                                                      int64_t total, int64_t ypg, int xcd, Divs3 dv) {
  constexpr float inv = 1.f / (float)S;
  const int H = h * S, W = w * S;
  const IndexRange rg = xcd_range(total, xcd);
  const bool fast = total < kFastIndexLimit;
  for (int64_t i = rg.begin; i < rg.end; i += rg.stride) {
    const Index4 ix = split_index(i, dv, fast);  // (cg, cx + 1, cy + 1, b)
    const int cg = ix.c0, cx = ix.c1 - 1, cy = ix.c2 - 1;
    const int b = (int)ix.c3;
    const int r0 = cy < 0 ? 0 : cy, r1 = min(r0 + 1, h - 1), c0 = cx < 0 ? 0 : cx, c1 = min(c0 + 1, w - 1);
    const float4* xb = x + (int64_t)b * h * w * CG + cg;
    const float4 v00 = xb[((int64_t)r0 * w + c0) * CG], v01 = xb[((int64_t)r0 * w + c1) * CG];
    const float4 v10 = xb[((int64_t)r1 * w + c0) * CG], v11 = xb[((int64_t)r1 * w + c1) * CG];
    float4 t0[S], t1[S];
#pragma unroll
    for (int kx = 0; kx < S; ++kx) {
      const float lx = cx < 0 ? 0.f : ((float)kx + 0.5f) * inv, ux = 1.f - lx;
      t0[kx] = make_float4(ux * v00.x + lx * v01.x, ux * v00.y + lx * v01.y, ux * v00.z + lx * v01.z,
                           ux * v00.w + lx * v01.w);
      t1[kx] = make_float4(ux * v10.x + lx * v11.x, ux * v10.y + lx * v11.y, ux * v10.z + lx * v11.z,
                           ux * v10.w + lx * v11.w);
    }
    const int Yb = S * cy + S / 2, Xb = S * cx + S / 2;
#pragma unroll
    for (int ky = 0; ky < S; ++ky) {
      const int Y = Yb + ky;
      if (Y < 0 || Y >= H) continue;
      const float ly = cy < 0 ? 0.f : ((float)ky + 0.5f) * inv, uy = 1.f - ly;
#pragma unroll
      for (int kx = 0; kx < S; ++kx) {
        const int X = Xb + kx;
        if (X < 0 || X >= W) continue;
        const int64_t opix = ((int64_t)b * H + Y) * W + X;
        float4 o = make_float4(uy * t0[kx].x + ly * t1[kx].x, uy * t0[kx].y + ly * t1[kx].y,
                               uy * t0[kx].z + ly * t1[kx].z, uy * t0[kx].w + ly * t1[kx].w);
        if (res) {
          const float4 r = res[opix * CG + cg];
          o.x += r.x;
          o.y += r.y;
          o.z += r.z;
          o.w += r.w;
        }
        y[opix * ypg + cg] = o;
      }
    }
  }
}

// gather: one lane = (input pixel, 4 channels); footprint rows/cols from ATen's source-index rule
__global__ __launch_bounds__(256) void upsample_nhwc_bwd_kernel(const float4* __restrict__ gy, float4* __restrict__ gx,
                                                                int CG, int h, int w, int H, int W, float rh, float rw,
                                                                int64_t total, int64_t gpg, int xcd, Divs3 dv) {
  const IndexRange rg = xcd_range(total, xcd);
  const bool fast = total < kFastIndexLimit;
  for (int64_t i = rg.begin; i < rg.end; i += rg.stride) {
    const Index4 ix = split_index(i, dv, fast);  // (cg, xq, yq, b)
    const int cg = ix.c0, xq = ix.c1, yq = ix.c2;
    const int b = (int)ix.c3;
    const int Ylo = first_dst_ge(yq - 1, rh, h, H), Yhi = first_dst_ge(yq + 1, rh, h, H);
    const int Xlo = first_dst_ge(xq - 1, rw, w, W), Xhi = first_dst_ge(xq + 1, rw, w, W);
    const float4* gb = gy + (int64_t)b * H * W * gpg + cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int Y = Ylo; Y < Yhi; ++Y) {
      const AxisMapU my = axis_map_u(Y, rh, h);
      const float wy = ((my.i0 == yq) ? (1.f - my.lam) : 0.f) + ((my.i1 == yq) ? my.lam : 0.f);
      if (wy == 0.f) continue;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int X = Xlo; X < Xhi; ++X) {
        const AxisMapU mx = axis_map_u(X, rw, w);
        const float wx = ((mx.i0 == xq) ? (1.f - mx.lam) : 0.f) + ((mx.i1 == xq) ? mx.lam : 0.f);
        const float4 g = gb[((int64_t)Y * W + X) * gpg];
        r.x = fmaf(wx, g.x, r.x);
        r.y = fmaf(wx, g.y, r.y);
        r.z = fmaf(wx, g.z, r.z);
        r.w = fmaf(wx, g.w, r.w);
      }
      acc.x = fmaf(wy, r.x, acc.x);
      acc.y = fmaf(wy, r.y, acc.y);
      acc.z = fmaf(wy, r.z, acc.z);
      acc.w = fmaf(wy, r.w, acc.w);
    }
    gx[i] = acc;
  }
}

// The same gather for SMALL inputs (the pyramid-pooling maps of the head: 1 x 1 ... 6 x 6 coarse pixels under a 16 x 16 output,
// uperforseg.py:171-177): one lane per (coarse pixel, 4 channels) leaves 6 ... 216 blocks looping over footprints of up to
// 16 x 16 pixels (71 us for 6 MB).  Here a block owns one coarse pixel x 16 channel groups and its 16 row-lanes take the
// footprint's rows Ylo + r, Ylo + r + 16, ...; the 16 partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void upsample_nhwc_bwd_small_kernel(const float4* __restrict__ gy, float4* __restrict__ gx,
                                                                      int CG, int h, int w, int H, int W, float rh, float rw,
                                                                      int64_t gpg) {
  __shared__ float4 part[16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int cg = blockIdx.y * 16 + cl;
  const int pix = blockIdx.x;               // (b, yq, xq)
  const int xq = pix % w, yq = (pix / w) % h, b = pix / (w * h);
  const int Ylo = first_dst_ge(yq - 1, rh, h, H), Yhi = first_dst_ge(yq + 1, rh, h, H);
  const int Xlo = first_dst_ge(xq - 1, rw, w, W), Xhi = first_dst_ge(xq + 1, rw, w, W);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cg < CG) {
    const float4* gb = gy + (int64_t)b * H * W * gpg + cg;
    for (int Y = Ylo + rl; Y < Yhi; Y += 16) {
      const AxisMapU my = axis_map_u(Y, rh, h);
      const float wy = ((my.i0 == yq) ? (1.f - my.lam) : 0.f) + ((my.i1 == yq) ? my.lam : 0.f);
      if (wy == 0.f) continue;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int X = Xlo; X < Xhi; ++X) {
        const AxisMapU mx = axis_map_u(X, rw, w);
        const float wx = ((mx.i0 == xq) ? (1.f - mx.lam) : 0.f) + ((mx.i1 == xq) ? mx.lam : 0.f);
        const float4 g = gb[((int64_t)Y * W + X) * gpg];
        r.x = fmaf(wx, g.x, r.x);
        r.y = fmaf(wx, g.y, r.y);
        r.z = fmaf(wx, g.z, r.z);
        r.w = fmaf(wx, g.w, r.w);
      }
      acc.x = fmaf(wy, r.x, acc.x);
      acc.y = fmaf(wy, r.y, acc.y);
      acc.z = fmaf(wy, r.z, acc.z);
      acc.w = fmaf(wy, r.w, acc.w);
    }
  }
  part[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && cg < CG) {
    float4 t = part[0][cl];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const float4 o = part[r][cl];
      t.x += o.x, t.y += o.y, t.z += o.z, t.w += o.w;
    }
    gx[(int64_t)pix * CG + cg] = t;
  }
}

// Power-of-two factor: the footprint of input pixel (yq, xq) is the 2S x 2S window at (S*yq - S/2, S*xq - S/2) with the
// separable constant weights of upsample_bwd_pow2_kernel; the rows are unrolled (2S independent 16-byte loads in flight
// per lane instead of one dependent load per tap), no float index arithmetic.
template <int S>
__global__ __launch_bounds__(256) void upsample_nhwc_bwd_pow2_kernel(const float4* __restrict__ gy,
                                                                     float4* __restrict__ gx, int CG, int h, int w,
                                                                     int64_t total, int64_t gpg, int xcd, Divs3 dv) {
  constexpr int T = 2 * S;
  constexpr float inv = 1.f / (float)S;
  const int H = h * S, W = w * S;
  const IndexRange rg = xcd_range(total, xcd);
  const bool fast = total < kFastIndexLimit;
  for (int64_t i = rg.begin; i < rg.end; i += rg.stride) {
    const Index4 ix = split_index(i, dv, fast);  // (cg, xq, yq, b)
    const int cg = ix.c0, xq = ix.c1, yq = ix.c2;
    const int b = (int)ix.c3;
    const bool eL = xq == 0, eR = xq == w - 1, eT = yq == 0, eB = yq == h - 1;
    const int X0 = S * xq - S / 2, Y0 = S * yq - S / 2;
    const float4* gb = gy + ((int64_t)b * H * W + X0) * gpg + cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
    for (int ty = 0; ty < T; ++ty) {
      const int Y = Y0 + ty;
      if (Y < 0 || Y >= H) continue;
      float cy = (ty < S) ? ((float)ty + 0.5f) * inv : ((float)(2 * S - ty) - 0.5f) * inv;
      cy = ((eT && ty < S) || (eB && ty >= S)) ? 1.f : cy;
      const float4* row = gb + (int64_t)Y * W * gpg;
      float4 v[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const bool outside = (eL && t < S / 2) || (eR && t >= 3 * S / 2);
        v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!outside) v[t] = row[(int64_t)t * gpg];
      }
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float c = (t < S) ? ((float)t + 0.5f) * inv : ((float)(2 * S - t) - 0.5f) * inv;
        const float wx = ((eL && t < S) || (eR && t >= S)) ? 1.f : c;
        r.x = fmaf(wx, v[t].x, r.x);
        r.y = fmaf(wx, v[t].y, r.y);
        r.z = fmaf(wx, v[t].z, r.z);
        r.w = fmaf(wx, v[t].w, r.w);
      }
      acc.x = fmaf(cy, r.x, acc.x);
      acc.y = fmaf(cy, r.y, acc.y);
      acc.z = fmaf(cy, r.z, acc.z);
      acc.w = fmaf(cy, r.w, acc.w);
    }
    gx[i] = acc;
  }
}

static bool plan_bwd(int h, int w, int H, int W, int* TI, int* RMAX, size_t* lds) {
  const double sh = (double)H / h, sw = (double)W / w;
  const double s = sh > sw ? sh : sw;
  // power-of-two input tiles (feature maps are powers of two: no ragged border tiles), region <= 80x80
  // full-res pixels (~25 KB of LDS -> 6 workgroups per CU)
  for (int t = 32; t >= 1; t >>= 1) {
    const int rm = (int)((t + 1) * s) + 4;
    const size_t b = sizeof(float) * ((size_t)rm * (rm + 1) + 4 * (size_t)rm + 2 * (size_t)(t + 3) + (size_t)rm * (t + 1));
    if (rm <= 80 && b <= 48 * 1024) {
      *TI = t;
      *RMAX = rm;
      *lds = b;
      return true;
    }
  }
  return false;
}

}  // namespace sea

using namespace sea;

// x: (planes, h, w) -> y: (planes, H, W), H >= h, W >= w
extern "C" int sea_upsample_bilinear_fwd(const float* x, float* y, int64_t planes, int h, int w, int H, int W,
                                         void* stream) {
  SEA_CHECK_ARG(x && y && planes > 0 && h > 0 && w > 0 && H >= h && W >= w);
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;
  hipStream_t s = (hipStream_t)stream;
  const int S = upsample_general_only() ? 0 : pow2_factor(h, w, H, W);
  if (S && (((uintptr_t)y) & 15) == 0) {  // float4 stores: W % 4 == 0 and a 16-byte aligned base
    if ((W & 3) == 0) {
      static const int lanes_only = [] {
        const char* e = getenv("SEA_UPSAMPLE_FWD");
        return (e && e[0] == 'l') ? 1 : 0;  // "lane": one float4 per lane (A/B against the cell variant)
      }();
      if (S >= 4 && !lanes_only) {
        const int64_t total = planes * (h + 1) * (W / 4);
        const dim3 grid(grid_for_xcd(total, 256)), block(256);
        const int xo = xcd_order_enabled() == 2;
        const FastDiv fW4 = fast_div((uint32_t)(W / 4)), fC = fast_div((uint32_t)(h + 1));
        static const int nt_mode = [] {  // SEA_UPSAMPLE_NT=0 / 1: never / always; default: by output size
          const char* e = getenv("SEA_UPSAMPLE_NT");
          return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : 2;
        }();
        const int nt = nt_mode == 2 ? (planes * H * W * 4 > (192ll << 20)) : nt_mode;
        switch (S) {
          case 4: hipLaunchKernelGGL(upsample_fwd_cells_kernel<4>, grid, block, 0, s, x, y, h, w, total, xo, fW4, fC, nt); break;
          case 8: hipLaunchKernelGGL(upsample_fwd_cells_kernel<8>, grid, block, 0, s, x, y, h, w, total, xo, fW4, fC, nt); break;
          default: hipLaunchKernelGGL(upsample_fwd_cells_kernel<16>, grid, block, 0, s, x, y, h, w, total, xo, fW4, fC, nt); break;
        }
        SEA_RETURN_LAST();
      }
      const int64_t total = planes * H * (W / 4);
      const dim3 grid(grid_for_xcd(total, 256)), block(256);
      // a pure output stream (the input is S^2 times smaller): the plain block order measured 10 % faster than the
      // XCD-contiguous one (151 planes x 8, 128 -> 512: 341 vs 379 us); SEA_XCD_ORDER=2 forces the latter
      const int xo = xcd_order_enabled() == 2;
      const FastDiv fW4 = fast_div((uint32_t)(W / 4)), fH = fast_div((uint32_t)H);
      switch (S) {
        case 2: hipLaunchKernelGGL(upsample_fwd_pow2_kernel<2>, grid, block, 0, s, x, y, h, w, total, xo, fW4, fH); break;
        case 4: hipLaunchKernelGGL(upsample_fwd_pow2_kernel<4>, grid, block, 0, s, x, y, h, w, total, xo, fW4, fH); break;
        case 8: hipLaunchKernelGGL(upsample_fwd_pow2_kernel<8>, grid, block, 0, s, x, y, h, w, total, xo, fW4, fH); break;
        default: hipLaunchKernelGGL(upsample_fwd_pow2_kernel<16>, grid, block, 0, s, x, y, h, w, total, xo, fW4, fH); break;
      }
      SEA_RETURN_LAST();
    }
  }
  for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
    const int np = (int)((planes - p0) < 65535 ? (planes - p0) : 65535);
    dim3 grid((W + 63) / 64, (H + 15) / 16, np);
    hipLaunchKernelGGL(upsample_fwd_kernel, grid, dim3(256), 0, s, x + p0 * h * w, y + p0 * H * W, h, w, H, W, rh, rw);
  }
  SEA_RETURN_LAST();
}

// gy: (planes, H, W) -> gx: (planes, h, w): gradient of sea_upsample_bilinear_fwd w.r.t. its input
extern "C" int sea_upsample_bilinear_bwd(const float* gy, float* gx, int64_t planes, int h, int w, int H, int W,
                                         void* stream) {
  SEA_CHECK_ARG(gy && gx && planes > 0 && h > 0 && w > 0 && H >= h && W >= w);
  {
    const int S = upsample_general_only() ? 0 : pow2_factor(h, w, H, W);
    // vector loads of min(4, S/2) floats at multiples of S/2 floats from the row start: rows must keep that alignment
    const int VL = S / 2 >= 4 ? 4 : S / 2;
    static const int gather_only = [] {
      const char* e = getenv("SEA_UPSAMPLE_BWD");
      return (e && e[0] == 'g') ? 1 : 0;
    }();
    const int cols = W / 4;
    if (S && !gather_only && (W & 3) == 0 && (cols == 32 || cols == 64 || cols == 128 || cols == 256) &&
        (((uintptr_t)gy) & 15) == 0 && planes < (1 << 24)) {
      int lcols = 5;
      while ((1 << lcols) < cols) ++lcols;
      const int nsub = 256 / cols;
      const int RP = 16 / S > 2 ? 16 / S : 2;                  // input rows per sub-band: ~5 cells of S rows each
      const int bands = (h + nsub * RP - 1) / (nsub * RP);
      const int64_t units64 = planes * bands;
      if (units64 < (1ll << 30)) {
        const int units = (int)units64;
        const int per_xcd = xcd_order_enabled() ? (units + 7) / 8 : 0;
        const dim3 grid(per_xcd ? per_xcd * 8 : units), block(256);
        hipStream_t s = (hipStream_t)stream;
        switch (S) {
          case 2: hipLaunchKernelGGL(upsample_bwd_rows_kernel<2>, grid, block, 0, s, gy, gx, h, w, lcols, RP, bands, units, per_xcd); break;
          case 4: hipLaunchKernelGGL(upsample_bwd_rows_kernel<4>, grid, block, 0, s, gy, gx, h, w, lcols, RP, bands, units, per_xcd); break;
          case 8: hipLaunchKernelGGL(upsample_bwd_rows_kernel<8>, grid, block, 0, s, gy, gx, h, w, lcols, RP, bands, units, per_xcd); break;
          default: hipLaunchKernelGGL(upsample_bwd_rows_kernel<16>, grid, block, 0, s, gy, gx, h, w, lcols, RP, bands, units, per_xcd); break;
        }
        SEA_RETURN_LAST();
      }
    }
    if (S && (((uintptr_t)gy) & (VL * 4 - 1)) == 0 && (W % (VL > 1 ? VL : 1)) == 0) {
      const int64_t total = planes * h * w;
      const dim3 grid(grid_for_xcd(total, 256)), block(256);
      const int xo = xcd_order_enabled() == 2;  // plain order measured faster (386 vs 411 us at x4, 151 planes x 8)
      const FastDiv fw = fast_div((uint32_t)w), fh = fast_div((uint32_t)h);
      hipStream_t s = (hipStream_t)stream;
      switch (S) {
        case 2: hipLaunchKernelGGL(upsample_bwd_pow2_kernel<2>, grid, block, 0, s, gy, gx, h, w, total, xo, fw, fh); break;
        case 4: hipLaunchKernelGGL(upsample_bwd_pow2_kernel<4>, grid, block, 0, s, gy, gx, h, w, total, xo, fw, fh); break;
        case 8: hipLaunchKernelGGL(upsample_bwd_pow2_kernel<8>, grid, block, 0, s, gy, gx, h, w, total, xo, fw, fh); break;
        default: hipLaunchKernelGGL(upsample_bwd_pow2_kernel<16>, grid, block, 0, s, gy, gx, h, w, total, xo, fw, fh); break;
      }
      SEA_RETURN_LAST();
    }
  }
  int TI, RMAX;
  size_t lds;
  SEA_CHECK_ARG(plan_bwd(h, w, H, W, &TI, &RMAX, &lds));
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;
  hipStream_t s = (hipStream_t)stream;
  for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
    const int np = (int)((planes - p0) < 65535 ? (planes - p0) : 65535);
    dim3 grid((w + TI - 1) / TI, (h + TI - 1) / TI, np);
    hipLaunchKernelGGL(upsample_bwd_kernel, grid, dim3(256), lds, s, gy + p0 * H * W, gx + p0 * h * w, h, w, H, W, rh,
                       rw, TI, RMAX);
  }
  SEA_RETURN_LAST();
}

// channels_last: x (B,h,w,C) -> y (B,H,W,C) and the gradient w.r.t. x; C % 4 == 0, 16-byte aligned.
// residual (nullable, dense (B,H,W,C)) is added to the up-sampled map.  y / gy may be a channel slice of a wider NHWC tensor: *_pixel_stride is the distance in floats between
// consecutive pixels (>= C, % 4 == 0), so the op can write into / read from a concatenation buffer in place.
extern "C" int sea_upsample_bilinear_nhwc_fwd(const float* x, const float* residual, float* y, int B, int C, int h, int w,
                                              int H, int W, int64_t y_pixel_stride, void* stream) {
  SEA_CHECK_ARG(x && y && B > 0 && C > 0 && (C % 4) == 0 && h > 0 && w > 0 && H >= h && W >= w);
  SEA_CHECK_ARG(y_pixel_stride >= C && (y_pixel_stride % 4) == 0);
  SEA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)residual)) & 15) == 0);
  const int64_t total = (int64_t)B * H * W * (C / 4);
  const int S = upsample_general_only() ? 0 : pow2_factor(h, w, H, W);
#define SEA_LAUNCH_NHWC_FWD(SS)                                                                                         \
  hipLaunchKernelGGL(upsample_nhwc_fwd_kernel<SS>, dim3(grid_for_xcd(total, 256 * 2)), dim3(256), 0,                   \
                     (hipStream_t)stream, (const float4*)x, (const float4*)residual, (float4*)y, C / 4, h, w, H, W,     \
                     (float)h / (float)H, (float)w / (float)W, total, y_pixel_stride / 4, xcd_order_enabled(),          \
                     divs3(C / 4, W, H))
  static const int lanes_only = [] {
    const char* e = getenv("SEA_UPSAMPLE_FWD");
    return (e && e[0] == 'l') ? 1 : 0;
  }();
  if ((S == 2 || S == 4 || S == 8) && !lanes_only) {
    const int64_t cells = (int64_t)B * (h + 1) * (w + 1) * (C / 4);
#define SEA_LAUNCH_NHWC_FWD_CELLS(SS)                                                                                   \
  hipLaunchKernelGGL(upsample_nhwc_fwd_cells_kernel<SS>, dim3(grid_for_xcd(cells, 256)), dim3(256), 0,                  \
                     (hipStream_t)stream, (const float4*)x, (const float4*)residual, (float4*)y, C / 4, h, w, cells,    \
                     y_pixel_stride / 4, xcd_order_enabled(), divs3(C / 4, w + 1, h + 1))
    if (S == 2) {
      SEA_LAUNCH_NHWC_FWD_CELLS(2);
    } else if (S == 4) {
      SEA_LAUNCH_NHWC_FWD_CELLS(4);
    } else {
      SEA_LAUNCH_NHWC_FWD_CELLS(8);
    }
#undef SEA_LAUNCH_NHWC_FWD_CELLS
    SEA_RETURN_LAST();
  }
  switch (S) {
    case 2: SEA_LAUNCH_NHWC_FWD(2); break;
    case 4: SEA_LAUNCH_NHWC_FWD(4); break;
    case 8: SEA_LAUNCH_NHWC_FWD(8); break;
    case 16: SEA_LAUNCH_NHWC_FWD(16); break;
    default: SEA_LAUNCH_NHWC_FWD(0); break;
  }
#undef SEA_LAUNCH_NHWC_FWD
  SEA_RETURN_LAST();
}

// ---- adaptive average pooling of an NHWC map to oh x ow bins (the pyramid pooling of the head, uperforseg.py:150-177:
// 16 x 16 -> 1, 2, 3, 6), ATen's bin rule: rows [floor(i H / oh), ceil((i + 1) H / oh)).  ATen's NHWC kernel runs these on
// 8 blocks (52 us for 6 MB); here a block owns one bin x 16 channel groups, 16 pixel-lanes stride through the bin's pixels
// and the partial sums meet in LDS in a fixed order.
namespace sea {
__device__ __forceinline__ int bin_start(int i, int n_in, int n_out) { return (int)(((int64_t)i * n_in) / n_out); }
__device__ __forceinline__ int bin_end(int i, int n_in, int n_out) { return (int)(((int64_t)(i + 1) * n_in + n_out - 1) / n_out); }

__global__ __launch_bounds__(256) void adaptive_pool_nhwc_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ out,
                                                                     int CG, int H, int W, int oh, int ow) {
  __shared__ float4 part[16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int cg = blockIdx.y * 16 + cl;
  const int bin = blockIdx.x;               // (b, i, j)
  const int j = bin % ow, i = (bin / ow) % oh, b = bin / (ow * oh);
  const int y0 = bin_start(i, H, oh), y1 = bin_end(i, H, oh), x0 = bin_start(j, W, ow), x1 = bin_end(j, W, ow);
  const int bw = x1 - x0, n = (y1 - y0) * bw;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cg < CG) {
    const float4* xb = x + (int64_t)b * H * W * CG + cg;
    for (int k = rl; k < n; k += 16) {
      const int yy = y0 + k / bw, xx = x0 + k % bw;
      const float4 v = xb[((int64_t)yy * W + xx) * CG];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
  }
  part[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && cg < CG) {
    float4 t = part[0][cl];
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const float4 o = part[r][cl];
      t.x += o.x, t.y += o.y, t.z += o.z, t.w += o.w;
    }
    const float inv = 1.f / (float)n;
    out[(int64_t)bin * CG + cg] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
  }
}

// dx[b,y,x,:] = sum over the bins (i, j) that contain (y, x) of g[b,i,j,:] / |bin|   (at most 2 x 2 bins overlap a pixel)
__global__ __launch_bounds__(256) void adaptive_pool_nhwc_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ dx,
                                                                     int CG, int H, int W, int oh, int ow, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const int64_t pix = idx / CG;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    // candidate bins: the last one starting at or before the coordinate, and its predecessor
    int i1 = (int)((((int64_t)yy + 1) * oh - 1) / H);
    i1 = i1 > oh - 1 ? oh - 1 : i1;
    int j1 = (int)((((int64_t)xx + 1) * ow - 1) / W);
    j1 = j1 > ow - 1 ? ow - 1 : j1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = (i1 > 0 ? i1 - 1 : 0); i <= i1; ++i) {
      const int y0 = bin_start(i, H, oh), y1 = bin_end(i, H, oh);
      if (yy < y0 || yy >= y1) continue;
      for (int j = (j1 > 0 ? j1 - 1 : 0); j <= j1; ++j) {
        const int x0 = bin_start(j, W, ow), x1 = bin_end(j, W, ow);
        if (xx < x0 || xx >= x1) continue;
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
        const float4 v = g[(((int64_t)b * oh + i) * ow + j) * CG + cg];
        acc.x = fmaf(v.x, inv, acc.x), acc.y = fmaf(v.y, inv, acc.y), acc.z = fmaf(v.z, inv, acc.z), acc.w = fmaf(v.w, inv, acc.w);
      }
    }
    dx[idx] = acc;
  }
}
}  // namespace sea

// x (B,H,W,C) NHWC fp32 dense -> out (B,oh,ow,C): adaptive average pooling (ATen's bins); oh <= H, ow <= W
extern "C" int sea_adaptive_avg_pool_nhwc_fwd(const float* x, float* out, int B, int C, int H, int W, int oh, int ow,
                                              void* stream) {
  SEA_CHECK_ARG(x && out && B > 0 && C > 0 && (C % 4) == 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && oh <= H && ow <= W);
  SEA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0 && (int64_t)B * oh * ow < (1ll << 31));
  hipLaunchKernelGGL(adaptive_pool_nhwc_fwd_kernel, dim3(B * oh * ow, (C / 4 + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x, (float4*)out, C / 4, H, W, oh, ow);
  SEA_RETURN_LAST();
}

// g (B,oh,ow,C) -> dx (B,H,W,C): the input gradient of the pooling above
extern "C" int sea_adaptive_avg_pool_nhwc_bwd(const float* g, float* dx, int B, int C, int H, int W, int oh, int ow,
                                              void* stream) {
  SEA_CHECK_ARG(g && dx && B > 0 && C > 0 && (C % 4) == 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && oh <= H && ow <= W);
  SEA_CHECK_ARG(((((uintptr_t)g) | ((uintptr_t)dx)) & 15) == 0);
  const int64_t total = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(adaptive_pool_nhwc_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)g, (float4*)dx, C / 4, H, W, oh, ow, total);
  SEA_RETURN_LAST();
}

extern "C" int sea_upsample_bilinear_nhwc_bwd(const float* gy, float* gx, int B, int C, int h, int w, int H, int W,
                                              int64_t gy_pixel_stride, void* stream) {
  SEA_CHECK_ARG(gy && gx && B > 0 && C > 0 && (C % 4) == 0 && h > 0 && w > 0 && H >= h && W >= w);
  SEA_CHECK_ARG(gy_pixel_stride >= C && (gy_pixel_stride % 4) == 0);
  SEA_CHECK_ARG(((((uintptr_t)gy) | ((uintptr_t)gx)) & 15) == 0);
  const int64_t total = (int64_t)B * h * w * (C / 4);
  const int S = upsample_general_only() ? 0 : pow2_factor(h, w, H, W);
#define SEA_LAUNCH_NHWC_BWD(SS)                                                                                         \
  hipLaunchKernelGGL(upsample_nhwc_bwd_pow2_kernel<SS>, dim3(grid_for_xcd(total, 256)), dim3(256), 0,                   \
                     (hipStream_t)stream, (const float4*)gy, (float4*)gx, C / 4, h, w, total, gy_pixel_stride / 4,     \
                     xcd_order_enabled(), divs3(C / 4, w, h))
  if (S == 2) {
    SEA_LAUNCH_NHWC_BWD(2);
  } else if (S == 4) {
    SEA_LAUNCH_NHWC_BWD(4);
  } else if (S == 8) {
    SEA_LAUNCH_NHWC_BWD(8);
  } else if (total < 65536 && (int64_t)B * h * w < 65536) {
    hipLaunchKernelGGL(upsample_nhwc_bwd_small_kernel, dim3(B * h * w, (C / 4 + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)gy, (float4*)gx, C / 4, h, w, H, W, (float)h / (float)H, (float)w / (float)W,
                       gy_pixel_stride / 4);
  } else {
    hipLaunchKernelGGL(upsample_nhwc_bwd_kernel, dim3(grid_for_xcd(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)gy, (float4*)gx, C / 4, h, w, H, W, (float)h / (float)H, (float)w / (float)W,
                       total, gy_pixel_stride / 4, xcd_order_enabled(), divs3(C / 4, w, h));
  }
#undef SEA_LAUNCH_NHWC_BWD
  SEA_RETURN_LAST();
}

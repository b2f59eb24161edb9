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

// ---- channels_last variants: x (B,h,w,C), y (B,H,W,C), C % 4 == 0 ------------------------------------------
// Lanes run along the channel dimension (16-byte accesses, perfectly coalesced); no LDS.  The UperNet head
// is channels_last end to end on ROCm (MIOpen's NHWC igemm kernels return that layout), so these variants
// remove the layout copies around every up-sampling.
__global__ __launch_bounds__(256) void upsample_nhwc_fwd_kernel(const float4* __restrict__ x,
                                                                const float4* __restrict__ res, float4* __restrict__ y,
                                                                int CG, int h, int w, int H, int W, float rh, float rw,
                                                                int64_t total, int64_t ypg) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    int64_t p = i / CG;
    const int64_t opix = p;
    const int X = (int)(p % W);
    p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    const AxisMapU my = axis_map_u(Y, rh, h), mx = axis_map_u(X, rw, w);
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

// gather: one lane = (input pixel, 4 channels); footprint rows/cols from ATen's source-index rule
__global__ __launch_bounds__(256) void upsample_nhwc_bwd_kernel(const float4* __restrict__ gy, float4* __restrict__ gx,
                                                                int CG, int h, int w, int H, int W, float rh, float rw,
                                                                int64_t total, int64_t gpg) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % CG);
    int64_t p = i / CG;
    const int xq = (int)(p % w);
    p /= w;
    const int yq = (int)(p % h);
    const int b = (int)(p / h);
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
  hipLaunchKernelGGL(upsample_nhwc_fwd_kernel, dim3(grid_for(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x, (const float4*)residual, (float4*)y, C / 4, h, w, H, W, (float)h / (float)H, (float)w / (float)W, total,
                     y_pixel_stride / 4);
  SEA_RETURN_LAST();
}

extern "C" int sea_upsample_bilinear_nhwc_bwd(const float* gy, float* gx, int B, int C, int h, int w, int H, int W,
                                              int64_t gy_pixel_stride, void* stream) {
  SEA_CHECK_ARG(gy && gx && B > 0 && C > 0 && (C % 4) == 0 && h > 0 && w > 0 && H >= h && W >= w);
  SEA_CHECK_ARG(gy_pixel_stride >= C && (gy_pixel_stride % 4) == 0);
  SEA_CHECK_ARG(((((uintptr_t)gy) | ((uintptr_t)gx)) & 15) == 0);
  const int64_t total = (int64_t)B * h * w * (C / 4);
  hipLaunchKernelGGL(upsample_nhwc_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)gy, (float4*)gx, C / 4, h, w, H, W, (float)h / (float)H, (float)w / (float)W, total,
                     gy_pixel_stride / 4);
  SEA_RETURN_LAST();
}

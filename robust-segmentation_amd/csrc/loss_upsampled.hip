// K2u: the SEA loss kernel fused with the model's final bilinear upsample.
//
// UperNet computes logits at 1/4 resolution and Segmenter at 1/16, then both call
// F.interpolate(logits, size=input, mode="bilinear", align_corners=False) (uperforseg.py:416-418,
// segmenter.py:228).  The (B,C,H,W) upsampled logits and their gradient exist only to be consumed by
// the loss: they are the two largest tensors of an attack step (176 MB each at C=21, 1.27 GB at
// C=151).  This kernel takes the LOW-RES logits, interpolates on the fly out of LDS, evaluates the
// loss/argmax/accuracy per full-resolution pixel exactly like K2, and returns the gradient w.r.t.
// the low-res logits.  HBM traffic drops from 2*C*4 bytes per full-res pixel to 2*C*4 bytes per
// LOW-res pixel (16x / 256x less) plus the label read and argmax write.
//
// Determinism: the gradient of a low-res logit is a sum over the ~(2s)^2 full-res pixels whose
// bilinear footprint touches it.  It is evaluated as a GATHER (one accumulator per (low-res pixel,
// class), fixed summation order) instead of a scatter with float atomics, so results are bitwise
// reproducible.  Work is partitioned by low-res tiles: a workgroup owns TLxTL low-res pixels (it alone
// writes their gradient) and the full-res pixels whose top-left source index falls in the tile (it
// alone counts their loss/accuracy/argmax); it additionally recomputes the softmax statistics of the
// one-cell halo of full-res pixels that contribute to its tile's gradient.
//
// Interpolation follows ATen's upsample_bilinear2d (align_corners=False): src = r*(dst+0.5)-0.5
// clamped at 0, i0 = floor(src), i1 = min(i0+1, n-1), lambda = src-i0,
// val = (1-ly)*((1-lx)*v00 + lx*v01) + ly*((1-lx)*v10 + lx*v11).  Any (non-integer) scale works.
#include "sea_common.h"

namespace sea {

constexpr float kLn2u = 0.69314718055994530942f;

struct __attribute__((aligned(16))) BlockPartialU {
  float loss, track;
  int n_correct, pad;
};

struct AxisMap {
  int i0, i1;
  float lam;
};

__device__ __forceinline__ AxisMap axis_map(int dst, float r, int n_in) {
  float src = r * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  AxisMap m;
  m.i0 = (int)src;
  if (m.i0 > n_in - 1) m.i0 = n_in - 1;
  m.i1 = m.i0 + ((m.i0 < n_in - 1) ? 1 : 0);
  m.lam = src - (float)m.i0;
  return m;
}

// smallest dst in [0, n_out] whose i0 >= t  (i0 is non-decreasing in dst)
__device__ __forceinline__ int first_dst_with_i0_ge(int t, float r, int n_in, int n_out) {
  if (t <= 0) return 0;
  if (t > n_in - 1) return n_out;
  int d = (int)ceilf(((float)t + 0.5f) / r - 0.5f);
  d = d < 0 ? 0 : (d > n_out ? n_out : d);
  while (d > 0 && axis_map(d - 1, r, n_in).i0 >= t) --d;
  while (d < n_out && axis_map(d, r, n_in).i0 < t) ++d;
  return d;
}

__device__ __forceinline__ float lerp2(float v00, float v01, float v10, float v11, float lx, float ly) {
  const float top = (1.f - lx) * v00 + lx * v01;
  const float bot = (1.f - lx) * v10 + lx * v11;
  return (1.f - ly) * top + ly * bot;
}

__device__ __forceinline__ float loss_value_u(int mode, bool valid, bool correct, float ce, float logp, float py,
                                              float l1p, float wy) {
  switch (mode) {
    case SEA_MODE_MASK_CE: return correct ? ce : 0.f;
    case SEA_MODE_MASK_CE_BAL: return correct ? wy * ce : 0.f;
    case SEA_MODE_JS: return valid ? (kLn2u + 0.5f * (py * logp - (1.f + py) * l1p)) : 0.f;
    default: return valid ? ce : 0.f;
  }
}

// grid = (tiles_x, tiles_y, B), block = 256.  Dynamic LDS layout (floats unless noted):
//   lowt  [C][TLP][TLP]      low-res tile incl. 1-pixel halo, TLP = TL + 2
//   rmap  [RMAX] x {i0,i1 (local), lam}  row / column interpolation tables of the full-res region
//   pm, pA, pK [RMAX*RMAX], plab int [RMAX*RMAX]   per full-res pixel softmax statistics
//   rbeg, cbeg int [TL+3]   first region row/col whose source cell index i0 is >= ya-1+t (cell t = rows
//                           [rbeg[t], rbeg[t+1]))
template <bool GRAD>
__global__ __launch_bounds__(256) void loss_upsampled_kernel(
    const float* __restrict__ low, const void* __restrict__ y, int y_bytes, const float* __restrict__ w, int mode,
    int track_mode, int C, int h, int wl, int H, int W, float rh, float rw, float gscale, int TL, int RMAX,
    float* __restrict__ dlow, void* __restrict__ pred, int pred_bytes, BlockPartialU* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int TLP = TL + 2;
  float* lowt = smem;                                   // C*TLP*TLP
  int* r_i0 = (int*)(lowt + C * TLP * TLP);            // RMAX
  int* r_i1 = r_i0 + RMAX;
  float* r_lam = (float*)(r_i1 + RMAX);
  int* c_i0 = (int*)(r_lam + RMAX);
  int* c_i1 = c_i0 + RMAX;
  float* c_lam = (float*)(c_i1 + RMAX);
  float* pm = c_lam + RMAX;                             // RMAX*RMAX each
  float* pA = pm + RMAX * RMAX;
  float* pK = pA + RMAX * RMAX;
  int* plab = (int*)(pK + RMAX * RMAX);
  int* rbeg = plab + RMAX * RMAX;                       // TL+3
  int* cbeg = rbeg + (TL + 3);                          // TL+3

  const int b = blockIdx.z;
  const int ya = blockIdx.y * TL, xa = blockIdx.x * TL;             // owned low-res tile origin
  const int yb = min(ya + TL, h), xb = min(xa + TL, wl);
  // full-res region: rows whose i0 in [ya-1, yb) ; owned rows: i0 in [ya, yb)
  const int Y0e = first_dst_with_i0_ge(ya - 1, rh, h, H), Y0o = first_dst_with_i0_ge(ya, rh, h, H);
  const int Y1 = first_dst_with_i0_ge(yb, rh, h, H);
  const int X0e = first_dst_with_i0_ge(xa - 1, rw, wl, W), X0o = first_dst_with_i0_ge(xa, rw, wl, W);
  const int X1 = first_dst_with_i0_ge(xb, rw, wl, W);
  const int RH = Y1 - Y0e, RW = X1 - X0e;  // <= RMAX by construction of the launcher

  // ---- phase 0: low-res tile (rows ya-1 .. yb, clamped into the image) and axis tables -------------
  const float* lowb = low + (int64_t)b * C * h * wl;
  for (int i = threadIdx.x; i < C * TLP * TLP; i += 256) {
    const int c = i / (TLP * TLP), rem = i - c * TLP * TLP;
    const int ly = rem / TLP, lx = rem - ly * TLP;
    int gy = ya - 1 + ly, gx = xa - 1 + lx;
    gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
    gx = gx < 0 ? 0 : (gx > wl - 1 ? wl - 1 : gx);
    lowt[i] = lowb[((int64_t)c * h + gy) * wl + gx];
  }
  for (int i = threadIdx.x; i < RH; i += 256) {
    const AxisMap m = axis_map(Y0e + i, rh, h);
    r_i0[i] = m.i0 - (ya - 1);
    r_i1[i] = m.i1 - (ya - 1);
    r_lam[i] = m.lam;
  }
  for (int i = threadIdx.x; i < RW; i += 256) {
    const AxisMap m = axis_map(X0e + i, rw, wl);
    c_i0[i] = m.i0 - (xa - 1);
    c_i1[i] = m.i1 - (xa - 1);
    c_lam[i] = m.lam;
  }
  if (threadIdx.x < TL + 3) {
    const int t = threadIdx.x;  // local cell index: global source index ya-1+t
    rbeg[t] = min(max(first_dst_with_i0_ge(ya - 1 + t, rh, h, H), Y0e), Y1) - Y0e;
    cbeg[t] = min(max(first_dst_with_i0_ge(xa - 1 + t, rw, wl, W), X0e), X1) - X0e;
  }
  __syncthreads();

  // ---- phase A: per full-res pixel of the region: max / argmax / z_y / sum-exp / loss ---------------
  float lsum = 0.f, tsum = 0.f;
  int ncorr = 0;
  const int P2 = TLP * TLP;
  for (int p = threadIdx.x; p < RH * RW; p += 256) {
    const int ri = p / RW, ci = p - ri * RW;
    const int Y = Y0e + ri, X = X0e + ci;
    const bool owned = (Y >= Y0o) && (X >= X0o);
    int lab = load_label_rt(y, y_bytes, ((int64_t)b * H + Y) * W + X);
    lab = (lab < 0 || lab >= C) ? -1 : lab;
    // ONE pass over the classes with an online soft-max (running max + rescaled sum), and the two
    // horizontally adjacent corner values fetched as a pair (ds_read2_b32): 2 LDS instructions per class
    // instead of 8.  The tile rows are TLP wide, so index+1 is always inside the row buffer.
    const int o0 = r_i0[ri] * TLP + c_i0[ci], o1 = r_i1[ri] * TLP + c_i0[ci];
    const bool same_col = (c_i1[ci] == c_i0[ci]);
    const float ly = r_lam[ri], lx = c_lam[ci];
    float m = -INFINITY, zy = 0.f, s = 0.f;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
      const float* t = lowt + c * P2;
      const float a0 = t[o0], a1 = t[o0 + 1], b0 = t[o1], b1 = t[o1 + 1];
      const float z = lerp2(a0, same_col ? a0 : a1, b0, same_col ? b0 : b1, lx, ly);
      zy = (lab == c) ? z : zy;
      const float e = __expf(-fabsf(z - m));  // exp(-inf) = 0 on the first class
      if (!(z <= m) && !(m != m)) {             // z > m (strict: first maximum wins), or z is the first NaN (torch.max)
        s = s * e + 1.f;
        m = z;
        arg = c;
      } else {
        s += e;
      }
    }
    const bool valid = lab >= 0;
    const bool correct = valid && (arg == lab);
    const float lse = m + __logf(s);
    const float ce = lse - zy, logp = zy - lse;
    const float py = __expf(logp), l1p = __logf(1.f + py);
    const bool need_w = (mode == SEA_MODE_MASK_CE_BAL) || (track_mode == SEA_MODE_MASK_CE_BAL);
    const float wy = (need_w && valid) ? w[lab] : 1.f;
    if (owned) {
      const float lv = loss_value_u(mode, valid, correct, ce, logp, py, l1p, wy);
      lsum += lv;
      tsum += (track_mode == mode) ? lv : loss_value_u(track_mode, valid, correct, ce, logp, py, l1p, wy);
      ncorr += correct ? 1 : 0;
      if (pred != nullptr) store_index_rt(pred, pred_bytes, ((int64_t)b * H + Y) * W + X, arg);
    }
    if (GRAD) {
      float k;
      if (mode == SEA_MODE_JS)
        k = valid ? (-0.5f * (logp - l1p) * py) : 0.f;
      else if (mode == SEA_MODE_CE)
        k = valid ? 1.f : 0.f;
      else
        k = correct ? wy : 0.f;
      k *= gscale;
      pm[p] = m;
      pK[p] = k;
      pA[p] = k / s;
      plab[p] = lab;
    }
  }

  // ---- phase B: gather the gradient of every owned (low-res pixel, class) ---------------------------
  // A thread owns one low-res pixel and a chunk of CK classes.  The full-res pixels that touch the
  // low-res pixel (yl,xl) live in the four source cells (yl-1|yl) x (xl-1|xl); inside one cell all pixels
  // interpolate between the same four low-res values, which therefore sit in registers (per class) while
  // the per-pixel softmax statistics (max, K/sum, K, label) are read once and reused for the whole chunk.
  if (GRAD) {
    __syncthreads();
    constexpr int CK = 8;
    const int nlx = xb - xa, nly = yb - ya;
    const int nchunk = (C + CK - 1) / CK;
    float* dlb = dlow + (int64_t)b * C * h * wl;
    for (int item = threadIdx.x; item < nchunk * nly * nlx; item += 256) {
      const int ck = item / (nly * nlx), rem = item - ck * nly * nlx;
      const int ty = rem / nlx, tx = rem - ty * nlx;
      const int yl = ty + 1, xl = tx + 1;  // local low-res coordinates inside lowt
      const int c0 = ck * CK;
      float acc[CK];
#pragma unroll
      for (int k = 0; k < CK; ++k) acc[k] = 0.f;
      for (int qy = 0; qy < 2; ++qy) {
        const int cy = yl - 1 + qy;                       // source cell row (local)
        const int i_lo = rbeg[cy], i_hi = rbeg[cy + 1];
        if (i_lo >= i_hi) continue;
        const int cy1 = r_i1[i_lo];                       // constant inside the cell
        for (int qx = 0; qx < 2; ++qx) {
          const int cx = xl - 1 + qx;
          const int j_lo = cbeg[cx], j_hi = cbeg[cx + 1];
          if (j_lo >= j_hi) continue;
          const int cx1 = c_i1[j_lo];
          float v00[CK], v01[CK], v10[CK], v11[CK];
#pragma unroll
          for (int k = 0; k < CK; ++k) {
            const int c = min(c0 + k, C - 1);
            const float* t = lowt + c * P2;
            v00[k] = t[cy * TLP + cx];
            v01[k] = t[cy * TLP + cx1];
            v10[k] = t[cy1 * TLP + cx];
            v11[k] = t[cy1 * TLP + cx1];
          }
          for (int ri = i_lo; ri < i_hi; ++ri) {
            const float ly = r_lam[ri];
            const float wyv = ((cy == yl) ? (1.f - ly) : 0.f) + ((cy1 == yl) ? ly : 0.f);
            if (wyv == 0.f) continue;
            for (int ci = j_lo; ci < j_hi; ++ci) {
              const float lx = c_lam[ci];
              const float wv = wyv * (((cx == xl) ? (1.f - lx) : 0.f) + ((cx1 == xl) ? lx : 0.f));
              const int p = ri * RW + ci;
              const float kk = pK[p];
              if (wv == 0.f || kk == 0.f) continue;
              const float mm = pm[p], wa = wv * pA[p], wk = wv * kk;
              const int lb = plab[p] - c0;
#pragma unroll
              for (int k = 0; k < CK; ++k) {
                const float z = lerp2(v00[k], v01[k], v10[k], v11[k], lx, ly);
                acc[k] = fmaf(wa, __expf(z - mm), acc[k]);
                acc[k] -= (lb == k) ? wk : 0.f;
              }
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < CK; ++k)
        if (c0 + k < C) dlb[((int64_t)(c0 + k) * h + (ya + ty)) * wl + (xa + tx)] = acc[k];
    }
  }

  // ---- block reduction of the owned-pixel sums -----------------------------------------------------------
  __shared__ float s_l[4], s_t[4];
  __shared__ int s_n[4];
  lsum = wave_sum(lsum);
  tsum = wave_sum(tsum);
  ncorr = wave_sum_i(ncorr);
  if ((threadIdx.x & 63) == 0) {
    s_l[threadIdx.x >> 6] = lsum;
    s_t[threadIdx.x >> 6] = tsum;
    s_n[threadIdx.x >> 6] = ncorr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    BlockPartialU o;
    o.loss = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
    o.track = (s_t[0] + s_t[1]) + (s_t[2] + s_t[3]);
    o.n_correct = s_n[0] + s_n[1] + s_n[2] + s_n[3];
    o.pad = 0;
    partials[((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = o;
  }
}

__global__ __launch_bounds__(256) void loss_upsampled_finalize(const BlockPartialU* __restrict__ partials, int tiles,
                                                               float* __restrict__ loss_sum,
                                                               float* __restrict__ track_sum,
                                                               int32_t* __restrict__ n_correct) {
  __shared__ double s_l[256], s_t[256];
  __shared__ int s_n[256];
  const int b = blockIdx.x;
  double l = 0.0, t = 0.0;
  int n = 0;
  for (int i = threadIdx.x; i < tiles; i += 256) {
    const BlockPartialU p = partials[(int64_t)b * tiles + i];
    l += (double)p.loss;
    t += (double)p.track;
    n += p.n_correct;
  }
  s_l[threadIdx.x] = l;
  s_t[threadIdx.x] = t;
  s_n[threadIdx.x] = n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s_l[threadIdx.x] += s_l[threadIdx.x + o];
      s_t[threadIdx.x] += s_t[threadIdx.x + o];
      s_n[threadIdx.x] += s_n[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss_sum[b] = (float)s_l[0];
    track_sum[b] = (float)s_t[0];
    n_correct[b] = s_n[0];
  }
}

struct UpsPlan {
  int TL, RMAX, tiles_x, tiles_y;
  size_t lds;
};

// choose the low-res tile edge: as large as the LDS budget and a 48-pixel full-res region allow
static bool plan_upsampled(int C, int h, int wl, int H, int W, UpsPlan* out) {
  const double sh = (double)H / h, sw = (double)W / wl;
  const double s = sh > sw ? sh : sw;
  for (int TL = 8; TL >= 1; --TL) {
    const int RMAX = (int)((TL + 1) * s) + 4;
    const size_t TLP = TL + 2;
    const size_t lds = sizeof(float) * ((size_t)C * TLP * TLP + 6 * (size_t)RMAX + 4 * (size_t)RMAX * RMAX + 2 * (size_t)(TL + 3));
    if (lds <= 96 * 1024 && RMAX <= 72) {
      out->TL = TL;
      out->RMAX = RMAX;
      out->tiles_x = (wl + TL - 1) / TL;
      out->tiles_y = (h + TL - 1) / TL;
      out->lds = lds;
      return true;
    }
  }
  return false;
}

}  // namespace sea

using namespace sea;

extern "C" size_t sea_loss_upsampled_workspace_bytes(int B, int C, int h, int w, int H, int W) {
  UpsPlan p;
  if (B <= 0 || !plan_upsampled(C, h, w, H, W, &p)) return 0;
  return (size_t)B * p.tiles_x * p.tiles_y * sizeof(BlockPartialU);
}

extern "C" int sea_loss_fwd_bwd_upsampled(const float* low, const void* y, int y_bytes, const float* w, int mode,
                                          int track_mode, int B, int C, int h, int wl, int H, int W, float grad_scale,
                                          float* dlow, void* pred, int pred_bytes, void* workspace,
                                          size_t workspace_bytes, float* loss_sum, float* track_sum,
                                          int32_t* n_correct, void* stream) {
  SEA_CHECK_ARG(low && y && workspace && loss_sum && track_sum && n_correct);
  SEA_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && h > 0 && wl > 0 && H >= h && W >= wl);
  SEA_CHECK_ARG(mode >= 0 && mode <= 3 && track_mode >= 0 && track_mode <= 3);
  SEA_CHECK_ARG(!((mode == SEA_MODE_MASK_CE_BAL || track_mode == SEA_MODE_MASK_CE_BAL) && w == nullptr));
  SEA_CHECK_ARG(y_bytes == 8 || y_bytes == 4 || y_bytes == 2 || y_bytes == 1);
  SEA_CHECK_ARG(pred == nullptr || pred_bytes == 8 || pred_bytes == 4 || pred_bytes == 2 || pred_bytes == 1);
  SEA_CHECK_ARG(!(pred && pred_bytes == 1 && C > 255) && !(y_bytes == 1 && C > 255));
  UpsPlan p;
  SEA_CHECK_ARG(plan_upsampled(C, h, wl, H, W, &p));
  SEA_CHECK_ARG(workspace_bytes >= (size_t)B * p.tiles_x * p.tiles_y * sizeof(BlockPartialU));
  SEA_CHECK_ARG((((uintptr_t)workspace) & 15) == 0);
  const float rh = (float)h / (float)H, rw = (float)wl / (float)W;  // ATen: area_pixel_compute_scale
  dim3 grid(p.tiles_x, p.tiles_y, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dlow) {
    auto k = loss_upsampled_kernel<true>;
    if (p.lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(k, grid, block, p.lds, s, low, y, y_bytes, w, mode, track_mode, C, h, wl, H, W, rh, rw,
                       grad_scale, p.TL, p.RMAX, dlow, pred, pred_bytes, (BlockPartialU*)workspace);
  } else {
    auto k = loss_upsampled_kernel<false>;
    if (p.lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(k, grid, block, p.lds, s, low, y, y_bytes, w, mode, track_mode, C, h, wl, H, W, rh, rw,
                       grad_scale, p.TL, p.RMAX, (float*)nullptr, pred, pred_bytes, (BlockPartialU*)workspace);
  }
  hipLaunchKernelGGL(loss_upsampled_finalize, dim3(B), dim3(256), 0, s, (const BlockPartialU*)workspace,
                     p.tiles_x * p.tiles_y, loss_sum, track_sum, n_correct);
  SEA_RETURN_LAST();
}

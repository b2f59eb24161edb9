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


// ====================================================================================================================
// K2u for power-of-two ratios (round 6): x4 (UperNet, uperforseg.py:416-418) and x16 (Segmenter, segmenter.py:228).
//
// The general kernel above gathers: it evaluates the soft-max statistics per full-resolution pixel with lanes = pixels
// (four LDS reads, an interpolation and an exponential per class) and then, per low-resolution pixel, re-interpolates
// and re-exponentiates every pixel of its footprint -- ~45 lane operations per (pixel, class), one wave per SIMD behind
// 96 KB of LDS: 1.9 ms at C = 151 where up-sample + K2 + up-sample-backward take 1.0 (profiles/r5_cold_kernel_roofline.md).
// For a power-of-two ratio S the source cell of a pixel is integer arithmetic (t = Y + S/2: cell (t >> log2 S) - 1, lambda
// = ((t & (S - 1)) + 0.5) / S, exact in fp32), the S x S pixels of a CELL interpolate between the same four corner vectors,
// and the transposed assignment becomes natural:
//
//   * lanes = CLASSES (c = lane + 64 slot, up to three slots): a lane holds its classes' four corner logits in registers
//     for the whole cell, so z = lerp2(corners) costs two fused operations per class and pixel after the two horizontal
//     interpolants of a pixel column (shared by the S pixels of the column); max / sum over the classes are one wave
//     reduction each per pixel (DPP within rows of 16, three scalar operations across rows); arg-max is a ballot; the
//     exponentials are kept for the gradient; everything per pixel (loss, K, weights) is wave-uniform scalar work;
//   * the gradient is a SCATTER into registers: d z_c = K (p_c - [c == y]) goes to the lane's four corner accumulators
//     with the four bilinear weights -- no re-interpolation, no second exponential, and a pixel with K = 0 (masked losses:
//     every misclassified pixel) skips the gradient work altogether (wave-uniform branch);
//   * a wave walks a SEGMENT of one cell row left to right: the right corners of a cell are the left corners of the next,
//     so their accumulators are carried and a low-resolution column is flushed once per cell row -- as a "top" or a "bottom"
//     partial vector (classes contiguous: coalesced).  A second kernel adds, per low-resolution pixel and in a fixed order,
//     the bottom partial of the cell row above, the top partial of its own (plus the leading partials of segments that start
//     at its column, plus the clamped border rows) and transposes to the NCHW planes through LDS.  No atomics: bitwise
//     reproducible.  Labels and arg-max bytes of a cell row travel as one wave access (lane = pixel of the row).
//   * no LDS in the main kernel, ~70 registers: eight waves per SIMD; corner vectors are gathered straight from the planes
//     (64 lines per wave access, 2 x NS accesses per cell of S^2 pixels, prefetched one cell ahead).
// Interpolation arithmetic is ATen's expression tree (lerp2 above), so the logits equal F.interpolate's bit for bit.
__device__ __forceinline__ float dpp_max16(float v) {   // max over the 16 lanes of a DPP row, in every lane
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));  // row_mirror
  return v;
}
__device__ __forceinline__ float dpp_sum16(float v) {   // sum over the 16 lanes of a DPP row (lanes of a row agree up to order:
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // the caller
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // reads lane 0
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // of each row)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ float lane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float wave_max_u(float v) {
  v = dpp_max16(v);
  return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ float wave_sum_u(float v) {   // one value for the whole wave, fixed order
  v = dpp_sum16(v);
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}

constexpr int ups2_seg_cells(int S) { return S == 4 ? 8 : 2; }   // cells per wave (a segment of a cell row)

struct Ups2Args {
  const float* low;
  const void* y;
  const float* w;
  float* main_part;   // [B][2 roles][h + 1][wl][C]
  float* lead_part;   // [B][2 roles][h + 1][nseg][C]
  void* pred;
  BlockPartialU* partials;   // one record per wave: [B][(h + 1) * nseg]
  int y_bytes, pred_bytes, mode, track_mode, C, h, wl, nseg, B;
  float gscale;
};

template <int S, int NS, bool GRAD>
__global__ __launch_bounds__(256) void loss_upsampled_pow2_kernel(const Ups2Args p) {
  constexpr int SEGC = ups2_seg_cells(S);
  constexpr int LOG = S == 4 ? 2 : 4;
  const int lane = threadIdx.x & 63;
  const int C = p.C, h = p.h, wl = p.wl, H = h * S, W = wl * S;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // one wave = one (image, cell row, segment)
  const int per_img = (h + 1) * p.nseg;
  if (wid >= (int64_t)p.B * per_img) return;
  const int b = (int)(wid / per_img), rem = (int)(wid - (int64_t)b * per_img);
  const int ci_idx = rem / p.nseg, seg = rem - ci_idx * p.nseg;
  const int ci = ci_idx - 1;                                                 // cell row: source rows (ci, ci + 1), clamped
  const int r0 = ci < 0 ? 0 : ci, r1 = ci + 1 > h - 1 ? h - 1 : ci + 1;
  const int py_lo = ci < 0 ? S / 2 : 0, py_hi = ci == h - 1 ? S / 2 : S;
  const int cj_a = seg * SEGC - 1;
  int cj_b = cj_a + SEGC;
  cj_b = cj_b > wl ? wl : cj_b;                                              // cells cj_a .. cj_b - 1 of [-1, wl - 1]

  bool cv[NS];
  const float* pl0[NS];
  const float* pl1[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c = lane + 64 * s;
    cv[s] = c < C;
    const int cc = cv[s] ? c : C - 1;
    pl0[s] = p.low + (((int64_t)b * C + cc) * h + r0) * wl;
    pl1[s] = p.low + (((int64_t)b * C + cc) * h + r1) * wl;
  }
  float v00[NS], v10[NS], v01[NS], v11[NS], accT[NS], accB[NS];
  {
    const int c0 = cj_a < 0 ? 0 : cj_a;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      v00[s] = pl0[s][c0];
      v10[s] = pl1[s][c0];
      accT[s] = 0.f;
      accB[s] = 0.f;
    }
  }
  float* const mainT = GRAD ? p.main_part + ((((int64_t)b * 2 + 0) * (h + 1) + ci_idx) * wl) * C : nullptr;
  float* const mainB = GRAD ? p.main_part + ((((int64_t)b * 2 + 1) * (h + 1) + ci_idx) * wl) * C : nullptr;
  float* const leadT = GRAD ? p.lead_part + ((((int64_t)b * 2 + 0) * (h + 1) + ci_idx) * p.nseg + seg) * C : nullptr;
  float* const leadB = GRAD ? p.lead_part + ((((int64_t)b * 2 + 1) * (h + 1) + ci_idx) * p.nseg + seg) * C : nullptr;
  bool leading = true;                                                      // the next flush is the segment's first column
  auto flush = [&](int col) __attribute__((always_inline)) {
    if constexpr (GRAD) {
      float* const t = leading ? leadT : mainT + (int64_t)col * C;
      float* const bo = leading ? leadB : mainB + (int64_t)col * C;
#pragma unroll
      for (int s = 0; s < NS; ++s)
        if (cv[s]) {
          t[lane + 64 * s] = accT[s];
          bo[lane + 64 * s] = accB[s];
        }
      leading = false;
    }
  };

  float lsum = 0.f, tsum = 0.f;
  int ncorr = 0;
  const bool need_w = (p.mode == SEA_MODE_MASK_CE_BAL) || (p.track_mode == SEA_MODE_MASK_CE_BAL);
  // (the right corners of the NEXT cell are fetched while this cell is computed: a corner gather touches 64 lines)
  float n01[NS], n11[NS];
  {
    const int c1 = cj_a + 1 > wl - 1 ? wl - 1 : cj_a + 1;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      n01[s] = pl0[s][c1];
      n11[s] = pl1[s][c1];
    }
  }
  for (int cj = cj_a; cj < cj_b; ++cj) {
    const int c0 = cj < 0 ? 0 : cj, c1 = cj + 1 > wl - 1 ? wl - 1 : cj + 1;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      v01[s] = n01[s];
      v11[s] = n11[s];
    }
    if (cj + 1 < cj_b) {
      const int c2 = cj + 2 > wl - 1 ? wl - 1 : cj + 2;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        n01[s] = pl0[s][c2];
        n11[s] = pl1[s][c2];
      }
    }
    const int px_lo = cj < 0 ? S / 2 : 0, px_hi = cj == wl - 1 ? S / 2 : S;
    const int X0 = S * cj + S / 2;                                           // X = X0 + px, Y = Y0 + py
    const int Y0 = S * ci + S / 2;
    float nT[NS], nB[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      nT[s] = 0.f;
      nB[s] = 0.f;
    }
    // A GROUP of 16 pixels (x4: the cell, lane q = py * 4 + px; x16: one pixel row of the cell, lane q = px) in three phases:
    //   1  per pixel (lanes = classes): interpolated logits, max / arg-max / sum of exponentials / z_y over the classes -- two
    //      wave reductions and a ballot; the four statistics are written into LANE q of four registers, the exponentials kept;
    //   2  once per group (lanes = pixels): everything that is scalar per pixel -- log-sum-exp, CE / JS loss and tracking
    //      loss, correctness, the gradient coefficient K and K / sum -- as ONE vector pass over the 16 pixels instead of
    //      sixteen wave-uniform scalar chains on the vector ALU (that was three quarters of the instructions);
    //   3  per pixel with K != 0 (lanes = classes): d z_c = K (p_c - [c == y]) scattered into the four corner accumulators.
    auto group = [&](int64_t pix, bool in, auto geom) __attribute__((always_inline)) {
      // geom(q) -> {pixel q exists (wave-uniform), lx, ly}; pix / in: this LANE's pixel address and whether it exists
      int labv = -1;
      if (in) labv = load_label_rt(p.y, p.y_bytes, pix);
      labv = (labv < 0 || labv >= C) ? -1 : labv;
      float Mv = 0.f, Sv = 1.f, Zv = 0.f, E[GRAD ? 16 : 1][NS];
      int Av = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        bool ex;
        float lx, ly;
        geom(q, ex, lx, ly);
        if (!ex) continue;                                                  // (wave-uniform)
        float z[NS], e[NS];
        float mloc = -INFINITY;
        bool anynan = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          z[s] = lerp2(v00[s], v01[s], v10[s], v11[s], lx, ly);
          if (!cv[s]) z[s] = -INFINITY;
          mloc = fmaxf(mloc, z[s]);
          anynan = anynan || (z[s] != z[s]);
        }
        float m = wave_max_u(mloc);
        int arg = 0;
        if (__builtin_expect(__ballot(anynan) != 0, 0)) {                   // torch.max: the first NaN wins (cold path)
          arg = C;
#pragma unroll
          for (int s = NS - 1; s >= 0; --s) {
            const unsigned long long bm = __ballot(cv[s] && (z[s] != z[s]));
            if (bm) arg = 64 * s + __builtin_ctzll(bm);
          }
          m = __builtin_nanf("");
        } else {                                                            // the first class whose z equals the maximum
#pragma unroll
          for (int s = NS - 1; s >= 0; --s) {
            const unsigned long long bm = __ballot(cv[s] && z[s] == m);
            if (bm) arg = 64 * s + __builtin_ctzll(bm);
          }
        }
        float sloc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          e[s] = __expf(z[s] - m);            // exp(-inf) = 0 for the padding lanes
          sloc += e[s];
          if constexpr (GRAD) E[q][s] = e[s];
        }
        const float ssum = wave_sum_u(sloc);
        const int lab = __builtin_amdgcn_readlane(labv, q);
        float zy = 0.f;
        if (lab >= 0) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
            if ((lab >> 6) == s) zy = lane_f(z[s], lab & 63);
        }
        const bool mine = lane == q;                                        // (one compare + four selects per pixel)
        Mv = mine ? m : Mv;
        Sv = mine ? ssum : Sv;
        Zv = mine ? zy : Zv;
        Av = mine ? arg : Av;
        __builtin_amdgcn_sched_barrier(0);   // (pixels one after the other: interleaved by the scheduler they need 256 registers)
      }
      // ---- phase 2: lane = pixel
      const bool valid = in && labv >= 0;
      const bool correct = valid && (Av == labv);
      const float lse = Mv + __logf(Sv);
      const float ce = lse - Zv, logp = Zv - lse;
      const float pyv = __expf(logp), l1p = __logf(1.f + pyv);
      const float wy = (need_w && valid) ? p.w[labv] : 1.f;
      const float lv = loss_value_u(p.mode, valid, correct, ce, logp, pyv, l1p, wy);
      const float tv = (p.track_mode == p.mode) ? lv : loss_value_u(p.track_mode, valid, correct, ce, logp, pyv, l1p, wy);
      if (in) {
        lsum += lv;
        tsum += tv;
        ncorr += correct ? 1 : 0;
        if (p.pred != nullptr) store_index_rt(p.pred, p.pred_bytes, pix, Av);
      }
      if constexpr (GRAD) {
        float k;
        if (p.mode == SEA_MODE_JS)
          k = valid ? (-0.5f * (logp - l1p) * pyv) : 0.f;
        else if (p.mode == SEA_MODE_CE)
          k = valid ? 1.f : 0.f;
        else
          k = correct ? wy : 0.f;
        k *= p.gscale;
        if (!in) k = 0.f;
        const float av = k / Sv;
        // ---- phase 3: lanes = classes again
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          bool ex;
          float lx, ly;
          geom(q, ex, lx, ly);
          if (!ex) continue;
          const float kq = lane_f(k, q);
          if (kq == 0.f) continue;                                          // (wave-uniform: e.g. every misclassified pixel of a masked loss)
          const float aq = lane_f(av, q);
          const int lab = __builtin_amdgcn_readlane(labv, q);
          const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            float dz = aq * E[q][s];
            if ((lab >> 6) == s && lane == (lab & 63)) dz -= kq;
            accT[s] = fmaf(w00, dz, accT[s]);
            nT[s] = fmaf(w01, dz, nT[s]);
            accB[s] = fmaf(w10, dz, accB[s]);
            nB[s] = fmaf(w11, dz, nB[s]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if constexpr (S == 4) {
      const int lpy = lane >> 2, lpx = lane & 3;
      const bool in = lane < 16 && lpy >= py_lo && lpy < py_hi && lpx >= px_lo && lpx < px_hi;
      const int64_t pix = ((int64_t)b * H + (Y0 + lpy)) * W + (X0 + lpx);
      group(pix, in, [&](int q, bool& ex, float& lx, float& ly) {
        const int py = q >> 2, px = q & 3;
        ex = py >= py_lo && py < py_hi && px >= px_lo && px < px_hi;
        lx = cj < 0 ? 0.f : ((float)px + 0.5f) * 0.25f;
        ly = ci < 0 ? 0.f : ((float)py + 0.5f) * 0.25f;
      });
    } else {
      // 16 x 16 pixels per cell: sixteen 4 x 4 groups, lane q = (py & 3) * 4 + (px & 3) like the x4 cell
#pragma clang loop unroll(disable)
      for (int g16 = 0; g16 < 16; ++g16) {
        const int by = 4 * (g16 >> 2), bx = 4 * (g16 & 3);
        if (by + 4 <= py_lo || by >= py_hi || bx + 4 <= px_lo || bx >= px_hi) continue;
        const int lpy = by + (lane >> 2), lpx = bx + (lane & 3);
        const bool in = lane < 16 && lpy >= py_lo && lpy < py_hi && lpx >= px_lo && lpx < px_hi;
        const int64_t pix = ((int64_t)b * H + (Y0 + lpy)) * W + (X0 + lpx);
        group(pix, in, [&](int q, bool& ex, float& lx, float& ly) {
          const int py = by + (q >> 2), px = bx + (q & 3);
          ex = py >= py_lo && py < py_hi && px >= px_lo && px < px_hi;
          lx = cj < 0 ? 0.f : ((float)px + 0.5f) * (1.0f / 16);
          ly = ci < 0 ? 0.f : ((float)py + 0.5f) * (1.0f / 16);
        });
      }
    }
    if (c1 == c0) {                        // a clamped border cell: both columns are the same low-resolution pixel
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        accT[s] += nT[s];
        accB[s] += nB[s];
      }
    } else {
      flush(c0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        accT[s] = nT[s];
        accB[s] = nB[s];
        v00[s] = v01[s];
        v10[s] = v11[s];
      }
    }
  }
  {
    const int last = cj_b - 1;
    flush(last + 1 > wl - 1 ? wl - 1 : last + 1);
  }
  // (lanes 0-15 hold the sums of "their" pixels of every group: one record per wave, fixed order)
  lsum = wave_sum_u(lsum);
  tsum = wave_sum_u(tsum);
  ncorr = wave_sum_i(ncorr);
  if (lane == 0) {
    BlockPartialU o;
    o.loss = lsum;
    o.track = tsum;
    o.n_correct = ncorr;
    o.pad = 0;
    p.partials[wid] = o;
  }
}

// d low[b][c][i][j] = sum, in a fixed order, of the partial vectors that belong to low-resolution pixel (i, j):
//   top of cell row -1 (i = 0 only), bottom of cell row i - 1, top of cell row i, bottom of cell row h - 1 (i = h - 1 only);
//   each = the main partial of column j (j > 0) + the leading partial of the segment that starts at column j
template <int S>
__global__ __launch_bounds__(256) void loss_upsampled_pow2_combine(const float* __restrict__ main_part,
                                                                   const float* __restrict__ lead_part, int B, int C, int h,
                                                                   int wl, int nseg, float* __restrict__ dlow) {
  constexpr int SEGC = ups2_seg_cells(S);
  constexpr int TJ = 32;
  extern __shared__ float tile[];        // [C][TJ + 1]
  const int jt = blockIdx.x * TJ, i = blockIdx.y, b = blockIdx.z;
  auto term = [&](int role, int ci_idx, int j, int c) -> float {
    float v = 0.f;
    if (j > 0) v = main_part[((((int64_t)b * 2 + role) * (h + 1) + ci_idx) * wl + j) * C + c];
    // column j leads segment g when j == max(g SEGC - 1, 0)
    const int g = (j + 1) / SEGC;
    if ((j == 0) || ((j + 1) % SEGC == 0 && g < nseg)) {
      const int gg = j == 0 ? 0 : g;
      v += lead_part[((((int64_t)b * 2 + role) * (h + 1) + ci_idx) * nseg + gg) * C + c];
    }
    return v;
  };
  for (int idx = threadIdx.x; idx < TJ * C; idx += 256) {
    const int jl = idx / C, c = idx - jl * C, j = jt + jl;
    if (j >= wl) continue;
    float v = 0.f;
    if (i == 0) v += term(0, 0, j, c);                 // top of cell row -1
    v += term(1, i, j, c);                             // bottom of cell row i - 1  (index i)
    v += term(0, i + 1, j, c);                         // top of cell row i        (index i + 1)
    if (i == h - 1) v += term(1, h, j, c);             // bottom of cell row h - 1
    tile[c * (TJ + 1) + jl] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < TJ * C; idx += 256) {
    const int c = idx / TJ, jl = idx - c * TJ, j = jt + jl;
    if (j < wl) dlow[(((int64_t)b * C + c) * h + i) * wl + j] = tile[c * (TJ + 1) + jl];
  }
}

static int ups2_scale(int h, int wl, int H, int W, int C) {
  static const bool on = [] {
    const char* e = getenv("SEA_K2U_POW2");
    return !(e && e[0] == '0');
  }();
  if (!on || C > 192 || h < 2 || wl < 2) return 0;
  if (H == 4 * h && W == 4 * wl) return 4;
  if (H == 16 * h && W == 16 * wl) return 16;
  return 0;
}
static int ups2_nseg(int wl, int S) { return (wl + 1 + ups2_seg_cells(S) - 1) / ups2_seg_cells(S); }
// workspace: [wave records][main partials][leading partials]
static size_t ups2_rec_bytes(int B, int h, int wl, int S) { return ((size_t)B * (h + 1) * ups2_nseg(wl, S) * sizeof(BlockPartialU) + 255) / 256 * 256; }
static size_t ups2_main_bytes(int B, int C, int h, int wl) { return ((size_t)B * 2 * (h + 1) * wl * C * 4 + 255) / 256 * 256; }
static size_t ups2_lead_bytes(int B, int C, int h, int wl, int S) { return (size_t)B * 2 * (h + 1) * ups2_nseg(wl, S) * C * 4; }

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
  const size_t v1 = (size_t)B * p.tiles_x * p.tiles_y * sizeof(BlockPartialU);
  const int S = ups2_scale(h, w, H, W, C);
  if (!S) return v1;
  const size_t v2 = ups2_rec_bytes(B, h, w, S) + ups2_main_bytes(B, C, h, w) + ups2_lead_bytes(B, C, h, w, S);
  return v2 > v1 ? v2 : v1;
}

template <int S, int NS>
static void ups2_launch(const Ups2Args& a, bool grad, float* dlow, float* loss_sum, float* track_sum, int32_t* n_correct,
                        hipStream_t s) {
  const int64_t waves = (int64_t)a.B * (a.h + 1) * a.nseg;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (grad)
    hipLaunchKernelGGL((loss_upsampled_pow2_kernel<S, NS, true>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((loss_upsampled_pow2_kernel<S, NS, false>), grid, block, 0, s, a);
  if (grad) {
    const size_t lds = (size_t)a.C * 33 * sizeof(float);
    hipLaunchKernelGGL((loss_upsampled_pow2_combine<S>), dim3((a.wl + 31) / 32, a.h, a.B), dim3(256), lds, s, a.main_part,
                       a.lead_part, a.B, a.C, a.h, a.wl, a.nseg, dlow);
  }
  hipLaunchKernelGGL(loss_upsampled_finalize, dim3(a.B), dim3(256), 0, s, (const BlockPartialU*)a.partials, (a.h + 1) * a.nseg,
                     loss_sum, track_sum, n_correct);
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
  if (const int S = ups2_scale(h, wl, H, W, C)) {
    const size_t need = ups2_rec_bytes(B, h, wl, S) + ups2_main_bytes(B, C, h, wl) + ups2_lead_bytes(B, C, h, wl, S);
    if (workspace_bytes >= need) {       // (a caller with the general kernel's smaller workspace keeps the general kernel)
      Ups2Args a;
      a.low = low; a.y = y; a.w = w; a.pred = pred; a.y_bytes = y_bytes; a.pred_bytes = pred_bytes; a.mode = mode;
      a.track_mode = track_mode; a.C = C; a.h = h; a.wl = wl; a.B = B; a.nseg = ups2_nseg(wl, S); a.gscale = grad_scale;
      a.partials = (BlockPartialU*)workspace;
      a.main_part = (float*)((char*)workspace + ups2_rec_bytes(B, h, wl, S));
      a.lead_part = (float*)((char*)workspace + ups2_rec_bytes(B, h, wl, S) + ups2_main_bytes(B, C, h, wl));
      const hipStream_t st = (hipStream_t)stream;
      const int NS = (C + 63) / 64;
#define SEA_UPS2(SS)                                                                                         \
  do {                                                                                                       \
    if (NS == 1) ups2_launch<SS, 1>(a, dlow != nullptr, dlow, loss_sum, track_sum, n_correct, st);           \
    else if (NS == 2) ups2_launch<SS, 2>(a, dlow != nullptr, dlow, loss_sum, track_sum, n_correct, st);      \
    else ups2_launch<SS, 3>(a, dlow != nullptr, dlow, loss_sum, track_sum, n_correct, st);                   \
  } while (0)
      if (S == 4) SEA_UPS2(4); else SEA_UPS2(16);
#undef SEA_UPS2
      SEA_RETURN_LAST();
    }
  }
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

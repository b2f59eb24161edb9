// K2: fused per-pixel SEA loss forward + logit gradient + tracking loss + accuracy + argmax.
//
// Roofline: HBM.  Algorithmic bytes per pixel (fp32, with gradient): 2*C*4 (read logits, write
// dlogits) + label + pred.  One pass: every logit is loaded from HBM exactly once, kept in
// registers (class vector of a pixel = CPAD registers per lane, VEC pixels per lane), and its
// gradient is written exactly once.
//
// Layout NCHW ("class planes"): lane l of a wave owns VEC consecutive pixels; for every class c the
// wave reads one contiguous 64*VEC*4-byte segment of plane c -> perfectly coalesced
// global_load_dwordx4 (VEC=4) with C independent loads in flight per lane (latency hiding by ILP,
// no LDS needed because there is no reuse across lanes: the class reduction is within a lane).
//
// Per-image sums (loss, tracking loss, #correct) are reduced wave -> block with shuffles + LDS and
// written as one record per block; a second tiny kernel sums the records of an image in a fixed
// order in double precision.  No float atomics => bitwise run-to-run determinism.
#include "loss_common.h"

namespace sea {

// ---- NCHW, class vector in registers ---------------------------------------------------------
// grid = (tiles per image, B); block = 256 threads; tile = 256*VEC consecutive pixels of one image.
// CPAD = number of class registers; EXACT means C == CPAD (no per-class guards at all).
//
// Codegen notes (each one was a measured register-file problem):
//  * planes are addressed as ONE wave-uniform 64-bit pointer (SGPR pair, global address space)
//    advanced by HW per class, plus a 32-bit per-lane byte offset -> `global_load_dwordx4 v, v_off,
//    s[ptr:ptr+1]`.  The empty asm keeps the running pointer opaque so that the compiler does not
//    materialise C separate 64-bit plane addresses.
//  * argmax is computed as max-chain followed by a descending "first index equal to the max" scan;
//    the textbook `if (z > m) { m = z; arg = c; }` makes the compiler keep C 64-bit lane masks alive
//    (one per compare) and spill them.
//  * the label compares of the gradient pass use an opaque copy of the label so they are not CSE'd
//    with those of the z_y select (same reason).
// TUNE bits (benchmark variants): 1 = non-temporal gradient stores, 2 = non-temporal logit loads,
// 4 = ask for 4 waves/SIMD (<= 128 VGPRs)
template <typename T, int CPAD, int VEC, bool GRAD, bool EXACT, int TUNE = 0>
__global__ __launch_bounds__(256, (TUNE & 4) ? 4 : 1) void loss_nchw_reg(const T* __restrict__ logits, const void* __restrict__ y,
                                                     int y_bytes, const float* __restrict__ w, int mode,
                                                     int track_mode, int C, int64_t HW, float gscale,
                                                     T* __restrict__ dlogits, void* __restrict__ pred,
                                                     int pred_bytes, float* __restrict__ loss_px,
    BlockPartial* __restrict__ partials) {
  using R = typename Elem<T>::raw;
  using P = typename RawVec<R, VEC>::type;
  const int b = blockIdx.y;
  const int64_t px0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  const bool active = px0 < HW;  // HW % VEC == 0 is guaranteed by the launcher
  const uint32_t lane_off = (uint32_t)(threadIdx.x * VEC * sizeof(T));
  if (EXACT) C = CPAD;

  float z[CPAD][VEC];
  int lab[VEC];
#pragma unroll
  for (int c = 0; c < CPAD; ++c)
#pragma unroll
    for (int v = 0; v < VEC; ++v) z[c][v] = -INFINITY;
#pragma unroll
  for (int v = 0; v < VEC; ++v) lab[v] = -1;

  if (active) {
    // labels first: their latency hides under the C plane loads issued right after
    load_labels<VEC>(y, y_bytes, (int64_t)b * HW + px0, lab);
    gptr<char> plane = (gptr<char>)(logits + (int64_t)b * C * HW + (int64_t)blockIdx.x * 256 * VEC);
    const int64_t plane_bytes = HW * (int64_t)sizeof(T);
#pragma unroll
    for (int c = 0; c < CPAD; ++c) {
      if (EXACT || c < C) {
        P p;
        if (TUNE & 2)
          p = __builtin_nontemporal_load(reinterpret_cast<gptr<P>>(plane + lane_off));
        else
          p = *reinterpret_cast<gptr<P>>(plane + lane_off);
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[c][v] = Elem<T>::to_f(vec_get<R, VEC>(p, v));
      }
      plane += plane_bytes;
      asm volatile("" : "+s"(plane));
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) lab[v] = (lab[v] < 0 || lab[v] >= C) ? -1 : lab[v];
  }

  float lsum = 0.f, tsum = 0.f;
  int ncorr = 0;
  float K[VEC];  // gradient = K * (p_c - [c == y])
  float A[VEC];  // K / sum_exp
  int amax[VEC];
  float lpx[VEC];

#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float m = z[0][v];
#pragma unroll
    for (int c = 1; c < CPAD; ++c) m = fmaxf(m, z[c][v]);
    // first index attaining the maximum (torch.max tie-break): descending scan, last write wins
    int arg = 0;
    float zy = 0.f;
#pragma unroll
    for (int c = CPAD - 1; c >= 0; --c) {
      const float zc = z[c][v];
      arg = (zc == m) ? c : arg;
      zy = (lab[v] == c) ? zc : zy;
    }
    // Order fence (no instruction): the exp pass below overwrites z in place, so the scan above must
    // be complete first; without it the compiler runs exp early into fresh registers and keeps
    // BOTH z and exp(z-m) alive (2x the register file, half the occupancy).
    asm volatile("" : "+v"(m) : "v"(arg), "v"(zy));
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPAD; ++c) {
      const float e = __expf(z[c][v] - m);  // padded classes: exp(-inf) = 0
      z[c][v] = e;
      s += e;
    }
    // torch.max semantics for non-finite logits (reference attacker.py:145, 370, 485): the first NaN wins, else the
    // first maximum.  The fmaxf chain above ignores NaNs, but then exp(z - m) is NaN exactly at the NaN logits (and
    // at the +inf logits when the maximum is +inf, and everywhere when all logits are -inf), so the sum flags the
    // case and the first NaN among the e's is torch's index in all of them.  Never taken for finite logits.
    if (__builtin_expect(s != s, 0)) {
#pragma unroll
      for (int c = CPAD - 1; c >= 0; --c) arg = (z[c][v] != z[c][v]) ? c : arg;
    }
    const bool valid = active && lab[v] >= 0;
    const bool correct = valid && (arg == lab[v]);
    const float lse = m + __logf(s);
    const float ce = lse - zy;
    const float logp = zy - lse;
    const bool need_js = (mode == SEA_MODE_JS) || (track_mode == SEA_MODE_JS);
    float py = 0.f, l1p = 0.f;
    if (need_js) {
      py = __expf(logp);
      l1p = __logf(1.f + py);
    }
    const bool need_w = (mode == SEA_MODE_MASK_CE_BAL) || (track_mode == SEA_MODE_MASK_CE_BAL);
    const float wy = (need_w && valid) ? w[lab[v]] : 1.f;
    const float lv = loss_value(mode, valid, correct, ce, logp, py, l1p, wy);
    lsum += lv;
    lpx[v] = lv;
    tsum += (track_mode == mode) ? lv : loss_value(track_mode, valid, correct, ce, logp, py, l1p, wy);
    ncorr += correct ? 1 : 0;
    amax[v] = arg;
    if (GRAD) {
      K[v] = grad_coef(mode, valid, correct, logp, py, l1p, wy) * gscale;
      A[v] = K[v] / s;
    }
  }

  if (active) {
    if (pred != nullptr) {
      store_indices<VEC>(pred, pred_bytes, (int64_t)b * HW + px0, amax);
    }
    if (loss_px != nullptr) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) loss_px[(int64_t)b * HW + px0 + v] = lpx[v];
    }
    if (GRAD) {
      gptr_w<char> gplane = (gptr_w<char>)(dlogits + (int64_t)b * C * HW + (int64_t)blockIdx.x * 256 * VEC);
      const int64_t plane_bytes = HW * (int64_t)sizeof(T);
      int lab2[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        lab2[v] = lab[v];
        asm volatile("" : "+v"(lab2[v]));
      }
#pragma unroll
      for (int c = 0; c < CPAD; ++c) {
        if (EXACT || c < C) {
          P p;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            // K*(e_c/s - [c==y]) = A*e_c - (c==y ? K : 0)
            const float g = A[v] * z[c][v];
            vec_set<R, VEC>(p, v, Elem<T>::from_f((lab2[v] == c) ? g - K[v] : g));
          }
          if (TUNE & 1)
            __builtin_nontemporal_store(p, reinterpret_cast<gptr_w<P>>(gplane + lane_off));
          else
            *reinterpret_cast<gptr_w<P>>(gplane + lane_off) = p;
        }
        gplane += plane_bytes;
        asm volatile("" : "+s"(gplane));
      }
    }
  }
  block_reduce_store(lsum, tsum, ncorr, partials);
}

// ---- NCHW, any C: two passes over the class planes (second pass re-reads; used only when the class
// vector does not fit the register file) ---------------------------------------------------------------
template <typename T, bool GRAD>
__global__ __launch_bounds__(256) void loss_nchw_stream(const T* __restrict__ logits, const void* __restrict__ y,
                                                        int y_bytes, const float* __restrict__ w, int mode,
                                                        int track_mode, int C, int64_t HW, float gscale,
                                                        T* __restrict__ dlogits, void* __restrict__ pred,
                                                        int pred_bytes, float* __restrict__ loss_px,
    BlockPartial* __restrict__ partials) {
  const int b = blockIdx.y;
  const int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool active = px < HW;
  using R = typename Elem<T>::raw;
  const R* lbase = reinterpret_cast<const R*>(logits) + (int64_t)b * C * HW + px;
  float lsum = 0.f, tsum = 0.f;
  int ncorr = 0;
  if (active) {
    int lab = load_label_rt(y, y_bytes, (int64_t)b * HW + px);
    lab = (lab < 0 || lab >= C) ? -1 : lab;
    // online max / sum-exp: one read of every plane
    float m = Elem<T>::to_f(lbase[0]);
    int arg = 0;
    float s = 1.f;
    float zy = (lab == 0) ? m : 0.f;
    for (int c = 1; c < C; ++c) {
      const float zc = Elem<T>::to_f(lbase[(int64_t)c * HW]);
      zy = (lab == c) ? zc : zy;
      if (!(zc <= m) && !(m != m)) {  // zc > m, or zc is the first NaN (torch.max: a NaN wins and stays)
        s = s * __expf(m - zc) + 1.f;
        m = zc;
        arg = c;
      } else {
        s += __expf(zc - m);
      }
    }
    const bool valid = lab >= 0;
    const bool correct = valid && arg == lab;
    const float lse = m + __logf(s);
    const float ce = lse - zy, logp = zy - lse;
    const float py = __expf(logp), l1p = __logf(1.f + py);
    const bool need_w = (mode == SEA_MODE_MASK_CE_BAL) || (track_mode == SEA_MODE_MASK_CE_BAL);
    const float wy = (need_w && valid) ? w[lab] : 1.f;
    lsum = loss_value(mode, valid, correct, ce, logp, py, l1p, wy);
    tsum = (track_mode == mode) ? lsum : loss_value(track_mode, valid, correct, ce, logp, py, l1p, wy);
    ncorr = correct ? 1 : 0;
    if (pred != nullptr) store_index_rt(pred, pred_bytes, (int64_t)b * HW + px, arg);
    if (loss_px != nullptr) loss_px[(int64_t)b * HW + px] = lsum;
    if (GRAD) {
      const float k = grad_coef(mode, valid, correct, logp, py, l1p, wy) * gscale;
      R* gbase = reinterpret_cast<R*>(dlogits) + (int64_t)b * C * HW + px;
      if (k == 0.f) {
        for (int c = 0; c < C; ++c) gbase[(int64_t)c * HW] = Elem<T>::from_f(0.f);
      } else {
        for (int c = 0; c < C; ++c) {
          const float pc = __expf(Elem<T>::to_f(lbase[(int64_t)c * HW]) - lse);
          gbase[(int64_t)c * HW] = Elem<T>::from_f(k * (pc - ((lab == c) ? 1.f : 0.f)));
        }
      }
    }
  }
  block_reduce_store(lsum, tsum, ncorr, partials);
}

// ---- NHWC (channels_last logits): the class vector of a pixel is contiguous -----------------------
// A block stages 256 pixels x C classes through LDS with coalesced accesses, each lane then walks its
// pixel's row in LDS (row stride padded to an odd number of dwords -> bank-conflict free).
template <typename T, bool GRAD>
__global__ __launch_bounds__(256) void loss_nhwc_lds(const T* __restrict__ logits, const void* __restrict__ y,
                                                     int y_bytes, const float* __restrict__ w, int mode,
                                                     int track_mode, int C, int CS /*row stride, floats*/,
                                                     int64_t HW, float gscale, T* __restrict__ dlogits,
                                                     void* __restrict__ pred, int pred_bytes, float* __restrict__ loss_px,
                                                     BlockPartial* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [256][CS]
  const int b = blockIdx.y;
  const int64_t p0 = (int64_t)blockIdx.x * 256;
  const int npx = (int)min((int64_t)256, HW - p0);
  using R = typename Elem<T>::raw;
  const R* src = reinterpret_cast<const R*>(logits) + ((int64_t)b * HW + p0) * C;
  const int total = npx * C;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int p = i / C, c = i - p * C;
    tile[p * CS + c] = Elem<T>::to_f(src[i]);
  }
  __syncthreads();
  float lsum = 0.f, tsum = 0.f;
  int ncorr = 0;
  const int t = threadIdx.x;
  if (t < npx) {
    float* row = tile + t * CS;
    int lab = load_label_rt(y, y_bytes, (int64_t)b * HW + p0 + t);
    lab = (lab < 0 || lab >= C) ? -1 : lab;
    float m = row[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float zc = row[c];
      if (!(zc <= m) && !(m != m)) {  // zc > m, or zc is the first NaN (torch.max semantics)
        m = zc;
        arg = c;
      }
    }
    const float zy = lab >= 0 ? row[lab] : 0.f;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const float e = __expf(row[c] - m);
      row[c] = e;
      s += e;
    }
    const bool valid = lab >= 0;
    const bool correct = valid && arg == lab;
    const float lse = m + __logf(s);
    const float ce = lse - zy, logp = zy - lse;
    const float py = __expf(logp), l1p = __logf(1.f + py);
    const bool need_w = (mode == SEA_MODE_MASK_CE_BAL) || (track_mode == SEA_MODE_MASK_CE_BAL);
    const float wy = (need_w && valid) ? w[lab] : 1.f;
    lsum = loss_value(mode, valid, correct, ce, logp, py, l1p, wy);
    tsum = (track_mode == mode) ? lsum : loss_value(track_mode, valid, correct, ce, logp, py, l1p, wy);
    ncorr = correct ? 1 : 0;
    if (pred != nullptr) store_index_rt(pred, pred_bytes, (int64_t)b * HW + p0 + t, arg);
    if (loss_px != nullptr) loss_px[(int64_t)b * HW + p0 + t] = lsum;
    if (GRAD) {
      const float k = grad_coef(mode, valid, correct, logp, py, l1p, wy) * gscale;
      const float a = k / s;
      for (int c = 0; c < C; ++c) {
        const float g = a * row[c];
        row[c] = (lab == c) ? g - k : g;
      }
    }
  }
  if (GRAD) {
    __syncthreads();
    R* dst = reinterpret_cast<R*>(dlogits) + ((int64_t)b * HW + p0) * C;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int p = i / C, c = i - p * C;
      dst[i] = Elem<T>::from_f(tile[p * CS + c]);
    }
  }
  block_reduce_store(lsum, tsum, ncorr, partials);
}

// ---- second stage: fixed-order sum of the per-block records of each image ------------------------
__global__ __launch_bounds__(256) void loss_finalize(const BlockPartial* __restrict__ partials, int tiles,
                                                     float* __restrict__ loss_sum, float* __restrict__ track_sum,
                                                     int32_t* __restrict__ n_correct) {
  __shared__ double s_l[256], s_t[256];
  __shared__ int s_n[256];
  const int b = blockIdx.x;
  double l = 0.0, t = 0.0;
  int n = 0;
  for (int i = threadIdx.x; i < tiles; i += 256) {
    const BlockPartial p = partials[1 + (int64_t)b * tiles + i];
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

// ---- dispatch --------------------------------------------------------------------------------------
template <typename T, int CPAD, int VEC>
static void launch_reg(const LossArgs& a) {
  dim3 grid(tiles_for(a.HW, VEC), a.B), block(256);
#define SEA_GO(G, E)                                                                                    \
  hipLaunchKernelGGL((loss_nchw_reg<T, CPAD, VEC, G, E>), grid, block, 0, a.s, (const T*)a.logits, a.y, \
                     a.y_bytes, a.w, a.mode, a.track_mode, a.C, a.HW, a.gscale, (T*)a.dlogits, a.pred,  \
                     a.pred_bytes, a.loss_px, a.partials)
  const bool exact = (a.C == CPAD);
  if (a.dlogits) {
    if (exact)
      SEA_GO(true, true);
    else
      SEA_GO(true, false);
  } else {
    if (exact)
      SEA_GO(false, true);
    else
      SEA_GO(false, false);
  }
#undef SEA_GO
}

template <typename T>
static void launch_stream(const LossArgs& a) {
  dim3 grid(tiles_for(a.HW, 1), a.B), block(256);
  if (a.dlogits)
    hipLaunchKernelGGL((loss_nchw_stream<T, true>), grid, block, 0, a.s, (const T*)a.logits, a.y, a.y_bytes, a.w,
                       a.mode, a.track_mode, a.C, a.HW, a.gscale, (T*)a.dlogits, a.pred, a.pred_bytes, a.loss_px, a.partials);
  else
    hipLaunchKernelGGL((loss_nchw_stream<T, false>), grid, block, 0, a.s, (const T*)a.logits, a.y, a.y_bytes, a.w,
                       a.mode, a.track_mode, a.C, a.HW, a.gscale, (T*)nullptr, a.pred, a.pred_bytes, a.loss_px, a.partials);
}

template <typename T>
static int launch_nhwc(const LossArgs& a) {
  const int CS = a.C | 1;  // odd row stride (in dwords): lanes hit distinct banks
  const size_t lds = (size_t)256 * CS * sizeof(float);
  if (lds > 160 * 1024 - 64) return SEA_ERR_ARG;
  dim3 grid(tiles_for(a.HW, 1), a.B), block(256);
  if (a.dlogits) {
    auto k = loss_nhwc_lds<T, true>;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, a.s, (const T*)a.logits, a.y, a.y_bytes, a.w, a.mode, a.track_mode, a.C,
                       CS, a.HW, a.gscale, (T*)a.dlogits, a.pred, a.pred_bytes, a.loss_px, a.partials);
  } else {
    auto k = loss_nhwc_lds<T, false>;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, a.s, (const T*)a.logits, a.y, a.y_bytes, a.w, a.mode, a.track_mode, a.C,
                       CS, a.HW, a.gscale, (T*)nullptr, a.pred, a.pred_bytes, a.loss_px, a.partials);
  }
  return 0;
}

// choose the register-resident variant: exact-C instantiations for the datasets of the reference
// (VOC 21, ADE 151/150, Cityscapes 19), otherwise the smallest CPAD >= C.
template <typename T>
static int dispatch_nchw(const LossArgs& a, bool vec4_ok, bool vec2_ok, int* vec_used) {
  const int C = a.C;
#define SEA_REG(CP, V)       \
  do {                       \
    launch_reg<T, CP, V>(a); \
    *vec_used = V;           \
    return 0;                \
  } while (0)
  // force_vec: low 4 bits 0 = heuristic, 1/2/4 = pixels per lane; bits 4.. = TUNE variant (fp32 C=21/151 only)
  const int fv = a.force_vec & 15;
  int tune = (a.force_vec >> 4) & 15;
  // measured defaults (kernel_bench, MI355X): C=21 fp32 runs best with non-temporal loads+stores at 4
  // waves/SIMD (74 % of 8 TB/s vs 65 %), C=151 with non-temporal loads (69 % vs 67.6 %)
  if ((a.force_vec & 0xfff) == 0 && sizeof(T) == 4 && a.dlogits) tune = (C == 21 && vec4_ok) ? 7 : (C == 151 ? 2 : 0);
  if ((a.force_vec & 0xfff) == 0 && sizeof(T) == 4 && !a.dlogits && C == 21 && vec4_ok) tune = 6;
  if (tune == 15) tune = 0;  // explicit "no tuning" for A/B runs
  if (tune == 6 && !a.dlogits && sizeof(T) == 4 && C == 21 && vec4_ok) {  // no-gradient: nt loads, 4 waves/SIMD
    dim3 grid(tiles_for(a.HW, 4), a.B), block(256);
    hipLaunchKernelGGL((loss_nchw_reg<T, 21, 4, false, true, 6>), grid, block, 0, a.s, (const T*)a.logits, a.y,
                       a.y_bytes, a.w, a.mode, a.track_mode, a.C, a.HW, a.gscale, (T*)nullptr, a.pred, a.pred_bytes,
                       a.loss_px, a.partials);
    *vec_used = 4;
    return 0;
  }
  if (tune && a.dlogits && sizeof(T) == 4) {
    dim3 block(256);
#define SEA_TUNED(CP, V, TU)                                                                                         \
  do {                                                                                                               \
    dim3 grid(tiles_for(a.HW, V), a.B);                                                                              \
    hipLaunchKernelGGL((loss_nchw_reg<T, CP, V, true, true, TU>), grid, block, 0, a.s, (const T*)a.logits, a.y,      \
                       a.y_bytes, a.w, a.mode, a.track_mode, a.C, a.HW, a.gscale, (T*)a.dlogits, a.pred,             \
                       a.pred_bytes, a.loss_px, a.partials);                                                         \
    *vec_used = V;                                                                                                   \
    return 0;                                                                                                        \
  } while (0)
    if (C == 21 && vec4_ok) {
      if (tune == 1) SEA_TUNED(21, 4, 1);
      if (tune == 2) SEA_TUNED(21, 4, 2);
      if (tune == 3) SEA_TUNED(21, 4, 3);
      if (tune == 4) SEA_TUNED(21, 4, 4);
      if (tune == 5) SEA_TUNED(21, 4, 5);
      if (tune == 7) SEA_TUNED(21, 4, 7);
    }
    if (C == 151) {
      if (tune == 1) SEA_TUNED(151, 1, 1);
      if (tune == 2) SEA_TUNED(151, 1, 2);
      if (tune == 3) SEA_TUNED(151, 1, 3);
    }
#undef SEA_TUNED
  }
  if (vec4_ok && (fv == 0 || fv == 4)) {
    if (C <= 8) SEA_REG(8, 4);
    if (C <= 16) SEA_REG(16, 4);
    if (C == 19) SEA_REG(19, 4);
    if (C == 21) SEA_REG(21, 4);
    if (C <= 24) SEA_REG(24, 4);
    if (C <= 32) SEA_REG(32, 4);
  }
  if (vec2_ok && (fv == 0 || fv == 2 || fv == 4)) {
    if (C <= 8) SEA_REG(8, 2);
    if (C <= 16) SEA_REG(16, 2);
    if (C == 19) SEA_REG(19, 2);
    if (C == 21) SEA_REG(21, 2);
    if (C <= 24) SEA_REG(24, 2);
    if (C <= 32) SEA_REG(32, 2);
    if (C <= 48) SEA_REG(48, 2);
    if (C <= 64) SEA_REG(64, 2);
  }
  if (C <= 8) SEA_REG(8, 1);
  if (C <= 16) SEA_REG(16, 1);
  if (C == 19) SEA_REG(19, 1);
  if (C == 21) SEA_REG(21, 1);
  if (C <= 24) SEA_REG(24, 1);
  if (C <= 32) SEA_REG(32, 1);
  if (C <= 48) SEA_REG(48, 1);
  if (C <= 64) SEA_REG(64, 1);
  if (C <= 96) SEA_REG(96, 1);
  if (C <= 128) SEA_REG(128, 1);
  if (C == 150) SEA_REG(150, 1);
  if (C == 151) SEA_REG(151, 1);
  if (C <= 160) SEA_REG(160, 1);
  if (C <= 192) SEA_REG(192, 1);
#undef SEA_REG
  launch_stream<T>(a);
  *vec_used = 1;
  return 0;
}

template <typename T>
static int dispatch_dtype(const LossArgs& a, int layout, int* tiles_used) {
  const int y_bytes = a.y_bytes;
  if (y_bytes != 8 && y_bytes != 4 && y_bytes != 2 && y_bytes != 1) return SEA_ERR_ARG;
  if (layout == SEA_LAYOUT_NHWC) {
    *tiles_used = tiles_for(a.HW, 1);
    return launch_nhwc<T>(a);
  }
  if (layout != SEA_LAYOUT_NCHW) return SEA_ERR_ARG;
  // vector width: every plane start (b*C+c)*HW + px0 must be VEC-element aligned
  auto al = [&](int vec) {
    const uintptr_t bytes = sizeof(T) * vec;
    return (a.HW % vec) == 0 && (((uintptr_t)a.logits) % bytes) == 0 &&
           (!a.dlogits || (((uintptr_t)a.dlogits) % bytes) == 0);
  };
  int vec = 1;
  // force_vec bit 12: legacy register kernels only (A/B runs); bits 8..11: variant of the streaming kernel
  const bool legacy = (a.force_vec & 0x1000) != 0;
  constexpr int V16 = 16 / (int)sizeof(T);
  // no gradient: beyond 32 classes the class vector no longer fits the registers at a useful occupancy -> streaming
  // kernel (loss_stream.hip).  Measured cold (tools/k2_lab.py): C=151 fp32 200 us vs 225 us register kernel, bf16
  // 130 us vs 165 us; at C=21 the register kernel (all 21 plane loads of a lane in flight at once) is as fast (fp32)
  // or faster (16-bit: 28.5 vs 32.7 us) than the chunk pipeline, which pays one memory round trip per chunk.
  const bool want_stream = a.C > 32 || ((a.force_vec >> 8) & 15) != 0;
  if (!legacy && !a.dlogits && al(V16) && (a.force_vec & 15) == 0 && want_stream) {
    launch_fwd<T>(a, (a.force_vec >> 8) & 15);
    *tiles_used = tiles_for(a.HW, V16);
    return 0;
  }
  // ADE-sized class vectors with gradient: split over the wave halves (loss_split.hip); variant 15 = skip (A/B runs)
  if (!legacy && a.dlogits && (a.force_vec & 15) == 0 && ((a.force_vec >> 8) & 15) != 15 &&
      dispatch_split<T>(a, tiles_used))
    return 0;
  const int rc = dispatch_nchw<T>(a, al(4), al(2), &vec);
  *tiles_used = tiles_for(a.HW, vec);
  return rc;
}

static int loss_fwd_bwd_impl(const void* logits, int dtype, int layout, const void* y, int y_bytes, const float* w,
                             int mode, int track_mode, int B, int C, int64_t HW, float grad_scale, void* dlogits,
                             void* pred, int pred_bytes, float* loss_px, void* workspace, size_t workspace_bytes,
                             float* loss_sum, float* track_sum, int32_t* n_correct, void* stream, int force_vec) {
  SEA_CHECK_ARG(logits && y && workspace);
  // loss_sum == track_sum == n_correct == NULL: leave the per-block records in the workspace for
  // sea_apgd_track (one launch less in the APGD loop)
  const bool deferred = !loss_sum && !track_sum && !n_correct;
  SEA_CHECK_ARG(deferred || (loss_sum && track_sum && n_correct));
  SEA_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && HW > 0);
  SEA_CHECK_ARG(mode >= 0 && mode <= 3 && track_mode >= 0 && track_mode <= 3);
  SEA_CHECK_ARG(!((mode == SEA_MODE_MASK_CE_BAL || track_mode == SEA_MODE_MASK_CE_BAL) && w == nullptr));
  SEA_CHECK_ARG(pred == nullptr || pred_bytes == 8 || pred_bytes == 4 || pred_bytes == 2 || pred_bytes == 1);
  SEA_CHECK_ARG(!(pred && pred_bytes == 1 && C > 255) && !(pred && pred_bytes == 2 && C > 32767));
  SEA_CHECK_ARG(!(y_bytes == 1 && C > 255));
  SEA_CHECK_ARG(workspace_bytes >= sea_loss_workspace_bytes(B, HW));
  SEA_CHECK_ARG((((uintptr_t)workspace) & 15) == 0);
  LossArgs a{logits, y, y_bytes, w, mode, track_mode, B, C, HW, grad_scale, dlogits, pred, pred_bytes,
             loss_px, (BlockPartial*)workspace, (hipStream_t)stream, force_vec};
  int tiles = 0, rc;
  switch (dtype) {
    case SEA_DTYPE_F32: rc = dispatch_dtype<float>(a, layout, &tiles); break;
    case SEA_DTYPE_BF16: rc = dispatch_dtype<__hip_bfloat16>(a, layout, &tiles); break;
    case SEA_DTYPE_F16: rc = dispatch_dtype<__half>(a, layout, &tiles); break;
    default: return SEA_ERR_ARG;
  }
  if (rc) return rc;
  if (!deferred)
    hipLaunchKernelGGL(loss_finalize, dim3(B), dim3(256), 0, a.s, (const BlockPartial*)workspace, tiles, loss_sum,
                       track_sum, n_correct);
  SEA_RETURN_LAST();
}

}  // namespace sea

using namespace sea;

extern "C" size_t sea_loss_workspace_bytes(int B, int64_t HW) {
  if (B <= 0 || HW <= 0) return 0;
  return ((size_t)B * (size_t)max_tiles(HW) + 1) * sizeof(BlockPartial);
}

extern "C" int sea_loss_fwd_bwd(const void* logits, int dtype, int layout, const void* y, int y_bytes,
                                const float* w, int mode, int track_mode, int B, int C, int64_t HW,
                                float grad_scale, void* dlogits, void* pred, int pred_bytes, float* loss_px,
                                void* workspace, size_t workspace_bytes, float* loss_sum, float* track_sum, int32_t* n_correct,
                                void* stream) {
  return loss_fwd_bwd_impl(logits, dtype, layout, y, y_bytes, w, mode, track_mode, B, C, HW, grad_scale, dlogits,
                           pred, pred_bytes, loss_px, workspace, workspace_bytes, loss_sum, track_sum, n_correct, stream, 0);
}

// benchmark hook: same as sea_loss_fwd_bwd but pins the pixels-per-lane of the register kernel
extern "C" int sea_loss_fwd_bwd_tuned(const void* logits, int dtype, int layout, const void* y, int y_bytes,
                                      const float* w, int mode, int track_mode, int B, int C, int64_t HW,
                                      float grad_scale, void* dlogits, void* pred, int pred_bytes, float* loss_px,
                                      void* workspace, size_t workspace_bytes, float* loss_sum, float* track_sum,
                                      int32_t* n_correct, void* stream, int force_vec) {
  return loss_fwd_bwd_impl(logits, dtype, layout, y, y_bytes, w, mode, track_mode, B, C, HW, grad_scale, dlogits,
                           pred, pred_bytes, loss_px, workspace, workspace_bytes, loss_sum, track_sum, n_correct, stream,
                           force_vec);
}

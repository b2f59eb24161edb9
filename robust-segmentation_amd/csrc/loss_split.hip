// K2 for ADE-sized class vectors with gradient (C = 150 / 151): the class vector of a pixel is SPLIT over the two
// 32-lane halves of a wave.
//
// The register-resident kernel of loss_kernels.hip keeps all C logits of a pixel in one lane: 151 VGPRs (+ temporaries
// = 246) leave 2 waves per SIMD, and with every wave in the same phase (load everything -> compute -> store
// everything) loads, arithmetic and stores barely overlap: 63-66 % of 8 TB/s in fp32, 54 % with 16-bit logits.
// Here lane l (< 32) of a wave holds classes [0, CH) of word column l and lane l + 32 holds classes [CH, C) of the
// SAME column (a word = one fp32 pixel or two 16-bit pixels), CH = ceil(C / 2): 76 class registers per lane, 4-5
// waves per SIMD.  The two partial soft-max statistics are merged with four cross-half shuffles per pixel
// (max / first-argmax / z_y / sum-exp); every lane then writes the gradient of its own classes.  A wave-instruction
// touches two 128-byte runs (class c and class c + CH of 32 consecutive words).
//
// fp32 keeps e = exp(z - m_half) in place of z and rescales by exp(m_half - m) through the per-pixel factor, so exp
// runs once per logit; 16-bit logits stay packed and exp is evaluated again in the gradient pass (see loss_stream.hip).
#include "loss_common.h"

namespace sea {

template <typename T, int C, int WAVES>
__global__ __launch_bounds__(256, WAVES) void loss_nchw_split(const T* __restrict__ logits, const void* __restrict__ y,
                                                              int y_bytes, const float* __restrict__ w, int mode,
                                                              int track_mode, int64_t HW, float gscale,
                                                              T* __restrict__ dlogits, void* __restrict__ pred,
                                                              int pred_bytes, float* __restrict__ loss_px,
                                                              BlockPartial* __restrict__ partials) {
  constexpr int PPW = Word<T>::PPW;
  constexpr int CH = (C + 1) / 2;           // classes per half (the upper half has C - CH of them)
  constexpr bool ODD = (C & 1) != 0;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 5;                // 0: classes [0, CH), 1: classes [CH, C)
  const int64_t col = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);  // word column inside the image
  const int64_t px0 = col * PPW;
  const bool active = px0 < HW;            // HW % PPW == 0 is guaranteed by the launcher
  const int64_t plane_bytes = HW * (int64_t)sizeof(T);
  // 32-bit lane offset from the (wave-uniform, SGPR) plane pointer: own half's first class + own column
  const uint32_t lane_off = (uint32_t)(sub * (int64_t)CH * plane_bytes + col * 4);
  const int cbase = sub * CH;

  uint32_t raw[CH];
  int lab[PPW];
#pragma unroll
  for (int h = 0; h < PPW; ++h) lab[h] = -1;
#pragma unroll
  for (int j = 0; j < CH; ++j) raw[j] = Word<T>::neg_inf();

  if (active) {
    load_labels<PPW>(y, y_bytes, (int64_t)b * HW + px0, lab);
    gptr<char> plane = (gptr<char>)(logits + (int64_t)b * C * HW);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (!(ODD && j == CH - 1) || sub == 0) {  // the upper half has one class less when C is odd
        uint32_t t[1];
        load_words<1>(plane + lane_off, t);
        raw[j] = t[0];
      }
      plane += plane_bytes;
      asm volatile("" : "+s"(plane));
    }
#pragma unroll
    for (int h = 0; h < PPW; ++h) lab[h] = (lab[h] < 0 || lab[h] >= C) ? -1 : lab[h];
  }

  float lsum = 0.f, tsum = 0.f;
  int ncorr = 0;
  float Kc[PPW], Ac[PPW], Mc[PPW];  // per pixel: K, K/sum (times the half's rescale in fp32), max used by the gradient
  int amax[PPW];
  const bool need_js = (mode == SEA_MODE_JS) || (track_mode == SEA_MODE_JS);
  const bool need_w = (mode == SEA_MODE_MASK_CE_BAL) || (track_mode == SEA_MODE_MASK_CE_BAL);
#pragma unroll
  for (int h = 0; h < PPW; ++h) {
    // ---- own half: max, first arg-max, z_y, sum of exp --------------------------------------------------------
#pragma unroll
    for (int j = 0; j < CH; ++j) fence_word(raw[j]);
    float m = Word<T>::get(raw[0], h);
#pragma unroll
    for (int j = 1; j < CH; ++j) {
      m = fmaxf(m, Word<T>::get(raw[j], h));
      if ((j & 7) == 7) {
        asm volatile("" : "+v"(m));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) fence_word(raw[j]);
    int argj = 0;
    float zy = 0.f;
    const int rel = lab[h] - cbase;  // own-half index of the label (outside [0, CH) when the other half has it)
#pragma unroll
    for (int j = CH - 1; j >= 0; --j) {
      const float zc = Word<T>::get(raw[j], h);
      argj = (zc == m) ? j : argj;   // descending scan: the first maximum wins (torch.max)
      zy = (rel == j) ? zc : zy;
      if ((j & 7) == 0) {
        asm volatile("" : "+v"(argj), "+v"(zy));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) fence_word(raw[j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float e = __expf(Word<T>::get(raw[j], h) - m);
      if constexpr (PPW == 1) raw[j] = __float_as_uint(e);  // fp32: keep exp(z - m_half) in place of z
      s += e;
      if ((j & 7) == 7) {
        asm volatile("" : "+v"(s));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    int arg = cbase + argj;
    // NaN / +-inf logits: the first NaN among the e's is torch.max's index.  (Inactive tail lanes hold -inf everywhere,
    // exp(-inf - -inf) = NaN: they must not drag the wave into the cold path.)
    if (__builtin_expect(active && s != s, 0)) {
      int an = 0x7fffffff;
#pragma unroll
      for (int j = CH - 1; j >= 0; --j) {
        const float e = (PPW == 1) ? __uint_as_float(raw[j]) : __expf(Word<T>::get(raw[j], h) - m);
        an = (e != e) ? cbase + j : an;
      }
      arg = an != 0x7fffffff ? an : arg;
    }
    // ---- merge the two halves (lane l <-> lane l + 32) ------------------------------------------------------------
    const float m_o = __shfl_xor(m, 32, 64);
    const float s_o = __shfl_xor(s, 32, 64);
    const int arg_o = __shfl_xor(arg, 32, 64);
    const float zy_o = __shfl_xor(zy, 32, 64);
    const bool s_bad = (s != s), so_bad = (s_o != s_o);
    // torch.max over both halves: a half holding a NaN wins (the lower class index if both do); otherwise the larger
    // maximum, the lower index on ties
    bool other;
    if (s_bad || so_bad)
      other = so_bad && (!s_bad || arg_o < arg);
    else
      other = (m_o > m) || (m_o == m && arg_o < arg);
    const int arg_t = other ? arg_o : arg;
    const float M = fmaxf(m, m_o);
    const float me = (M == -INFINITY) ? -3.0e38f : M;                    // all logits -inf: keep exp(-inf - -inf) out
    const float r_own = __expf(((m == -INFINITY) ? -3.0e38f : m) - me);   // exp(m_half - M) <= 1
    const float r_oth = __expf(((m_o == -INFINITY) ? -3.0e38f : m_o) - me);
    const float s_t = s * r_own + s_o * r_oth;
    const bool own_has = (rel >= 0) && (rel < CH);
    const float zy_t = own_has ? zy : zy_o;

    const bool valid = active && lab[h] >= 0;
    const bool correct = valid && (arg_t == lab[h]);
    const float lse = M + __logf(s_t);
    const float ce = lse - zy_t;
    const float logp = zy_t - lse;
    float py = 0.f, l1p = 0.f;
    if (need_js) {
      py = __expf(logp);
      l1p = __logf(1.f + py);
    }
    const float wy = (need_w && valid) ? w[lab[h]] : 1.f;
    const float lv = loss_value(mode, valid, correct, ce, logp, py, l1p, wy);
    if (sub == 0) {  // both halves hold the same per-pixel values: count them once
      lsum += lv;
      tsum += (track_mode == mode) ? lv : loss_value(track_mode, valid, correct, ce, logp, py, l1p, wy);
      ncorr += correct ? 1 : 0;
      if (active && loss_px != nullptr) loss_px[(int64_t)b * HW + px0 + h] = lv;
    }
    amax[h] = arg_t;
    Kc[h] = grad_coef(mode, valid, correct, logp, py, l1p, wy) * gscale;
    Ac[h] = (PPW == 1) ? (Kc[h] / s_t) * r_own : Kc[h] / s_t;
    Mc[h] = M;
    __builtin_amdgcn_sched_barrier(0);
  }

  if (active) {
    if (pred != nullptr && sub == 0) {
#pragma unroll
      for (int h = 0; h < PPW; ++h) store_index_rt(pred, pred_bytes, (int64_t)b * HW + px0 + h, amax[h]);
    }
    gptr_w<char> gplane = (gptr_w<char>)(dlogits + (int64_t)b * C * HW);
    int rel2[PPW];  // opaque copies: keep the compares below from being CSE'd with those of the z_y select
#pragma unroll
    for (int h = 0; h < PPW; ++h) {
      rel2[h] = lab[h] - cbase;
      asm volatile("" : "+v"(rel2[h]));
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (!(ODD && j == CH - 1) || sub == 0) {
        fence_word(raw[j]);
        float g[2] = {0.f, 0.f};
#pragma unroll
        for (int h = 0; h < PPW; ++h) {
          const float e = (PPW == 1) ? __uint_as_float(raw[j]) : __expf(Word<T>::get(raw[j], h) - Mc[h]);
          const float t = Ac[h] * e;
          g[h] = (rel2[h] == j) ? t - Kc[h] : t;  // K * (p_c - [c == y])
        }
        uint32_t out[1] = {Word<T>::pack(g[0], g[1])};
        store_words<1>(gplane + lane_off, out);
      }
      gplane += plane_bytes;
      asm volatile("" : "+s"(gplane));
      if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  }
  block_reduce_store(lsum, tsum, ncorr, partials);
}

template <typename T>
bool dispatch_split(const LossArgs& a, int* tiles_used) {
  constexpr int PPW = Word<T>::PPW;
  if (!a.dlogits || (a.C != 150 && a.C != 151) || (a.HW % PPW) != 0) return false;
  if ((((uintptr_t)a.logits) | ((uintptr_t)a.dlogits)) & 3) return false;
  if ((int64_t)a.C * a.HW * (int64_t)sizeof(T) >= (int64_t)1 << 31) return false;  // 32-bit lane offsets
  const int64_t cols = a.HW / PPW;
  const int tiles = (int)((cols + 127) / 128);
  dim3 grid(tiles, a.B), block(256);
  // waves per SIMD asked of the compiler.  Measured cold (tools/k2_lab.py, 8 x 151 x 512 x 512): fp32 4 waves 464 us
  // (3: 464, 5: 470; register kernel 485); bf16 3 waves 256 us (4 and 5 spill: 430 / 551 us; register kernel 293).
  // variant (A/B runs): 1 = 5 waves, 2 = 3 waves, 3 = 4 waves
  int variant = (a.force_vec >> 8) & 15;
  if (variant == 0) variant = sizeof(T) == 2 ? 2 : 3;
#define SEA_SPLIT(CC, WV)                                                                                          \
  hipLaunchKernelGGL((loss_nchw_split<T, CC, WV>), grid, block, 0, a.s, (const T*)a.logits, a.y, a.y_bytes, a.w,  \
                     a.mode, a.track_mode, a.HW, a.gscale, (T*)a.dlogits, a.pred, a.pred_bytes, a.loss_px, a.partials)
  if (a.C == 151) {
    if (variant == 1)
      SEA_SPLIT(151, 5);
    else if (variant == 2)
      SEA_SPLIT(151, 3);
    else
      SEA_SPLIT(151, 4);
  } else {
    if (variant == 1)
      SEA_SPLIT(150, 5);
    else if (variant == 2)
      SEA_SPLIT(150, 3);
    else
      SEA_SPLIT(150, 4);
  }
#undef SEA_SPLIT
  *tiles_used = tiles;
  return true;
}

template bool dispatch_split<float>(const LossArgs&, int*);
template bool dispatch_split<__hip_bfloat16>(const LossArgs&, int*);
template bool dispatch_split<__half>(const LossArgs&, int*);

}  // namespace sea

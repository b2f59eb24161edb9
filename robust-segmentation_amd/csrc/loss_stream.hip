// K2 without gradient, streaming variant (see loss_kernels.hip for the register-resident kernel and the C ABI,
// loss_split.hip for the ADE-sized gradient kernel).
//
// Logits sit in registers as raw 32-bit words (one fp32 pixel, or two 16-bit pixels per word) and are converted on
// use.  Two codegen rules, each of which cost hundreds of spilled VGPRs before it was applied:
//   * `fence_words`: an empty asm that makes the words opaque between passes, otherwise the compiler keeps the
//     CONVERTED floats of a whole chunk / class vector alive across passes instead of re-converting (1 VALU op);
//   * `__builtin_amdgcn_sched_barrier(0)` after each pixel group, otherwise the scheduler interleaves all groups of
//     a lane for ILP and their temporaries are live at the same time.
#include "loss_common.h"

namespace sea {

// torch.max over the classes of ONE pixel read straight from memory (first NaN wins, else first maximum): the
// cold path behind a NaN soft-max sum.
template <typename T>
__device__ __noinline__ int slow_torch_argmax(const T* __restrict__ px_base, int C, int64_t HW) {
  using R = typename Elem<T>::raw;
  const R* p = reinterpret_cast<const R*>(px_base);
  float m = Elem<T>::to_f(p[0]);
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float z = Elem<T>::to_f(p[(int64_t)c * HW]);
    if (!(z <= m) && !(m != m)) {
      m = z;
      arg = c;
    }
  }
  return arg;
}

// ---- NCHW, no gradient: streaming class loop ------------------------------------------------------------------
// Without a gradient nothing has to survive the class loop, so the class vector is NOT kept in registers: the
// planes stream through a double-buffered chunk of CH 16-byte loads per lane while the previous chunk is folded
// into an online soft-max (running max, rescaled running sum, first-maximum index, z_y).  Register use does not
// depend on C (the register-resident kernel needs C*VEC: 246 at C=151), loads are in flight all the time, and every
// dtype gets one 16-byte access per lane and plane (VEC = 4 fp32 / 8 bf16,f16 pixels per lane).  Used by every
// evaluation pass (clean / adversarial predict) and by the last iteration of each APGD run (attacker.py:467).
template <typename T, int CH, int WAVES>
__global__ __launch_bounds__(256, WAVES) void loss_nchw_fwd(const T* __restrict__ logits, const void* __restrict__ y,
                                                            int y_bytes, const float* __restrict__ w, int mode,
                                                            int track_mode, int C, int64_t HW, void* __restrict__ pred,
                                                            int pred_bytes, float* __restrict__ loss_px,
                                                            BlockPartial* __restrict__ partials) {
  constexpr int PPW = Word<T>::PPW, NW = 4, VEC = NW * PPW;
  const int b = blockIdx.y;
  const int64_t px0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  const bool active = px0 < HW;  // HW % VEC == 0 is guaranteed by the launcher
  const uint32_t lane_off = (uint32_t)threadIdx.x * 16u;
  const int64_t plane_bytes = HW * (int64_t)sizeof(T);

  float m[VEC], s[VEC], zy[VEC];
  int arg[VEC], lab[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    m[v] = -INFINITY;
    s[v] = 0.f;
    zy[v] = 0.f;
    arg[v] = 0;
    lab[v] = -1;
  }
  float lsum = 0.f, tsum = 0.f;
  int ncorr = 0;

  if (active) {
    load_labels<VEC>(y, y_bytes, (int64_t)b * HW + px0, lab);
#pragma unroll
    for (int v = 0; v < VEC; ++v) lab[v] = (lab[v] < 0 || lab[v] >= C) ? -1 : lab[v];
    gptr<char> plane = (gptr<char>)(logits + (int64_t)b * C * HW + (int64_t)blockIdx.x * 256 * VEC);

    // issue the plane loads of one chunk (n = classes in it, wave-uniform; only the last chunk has n < CH)
    auto load_chunk = [&](uint32_t(&buf)[CH][NW], int n) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (j < n) load_words<NW>(plane + lane_off, buf[j]);
        plane += plane_bytes;
        asm volatile("" : "+s"(plane));
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // fold one chunk into the running statistics, one word column (= PPW pixels) at a time
    auto reduce_chunk = [&](uint32_t(&buf)[CH][NW], int c0, int n) {
#pragma unroll
      for (int k = 0; k < NW; ++k) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (j >= n) buf[j][k] = Word<T>::neg_inf();
          fence_word(buf[j][k]);
        }
#pragma unroll
        for (int h = 0; h < PPW; ++h) {
          const int v = k * PPW + h;
          const float m_old = m[v];
          const int rel = lab[v] - c0;
          float mm = m_old, zz = zy[v];
          int aa = arg[v];
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const float z = Word<T>::get(buf[j][k], h);
            const bool take = z > mm;  // strict: the first maximum wins (NaNs: cold path at the end)
            mm = take ? z : mm;
            aa = take ? c0 + j : aa;
            zz = (rel == j) ? z : zz;
          }
          // a running maximum of -inf (nothing finite seen yet) must not turn exp(-inf - -inf) into NaN
          const float me = (mm == -INFINITY) ? -3.0e38f : mm;
          const float mo = (m_old == -INFINITY) ? -3.0e38f : m_old;
          float acc = s[v] * __expf(mo - me);
#pragma unroll
          for (int j = 0; j < CH; ++j) acc += __expf(Word<T>::get(buf[j][k], h) - me);
          m[v] = mm;
          arg[v] = aa;
          zy[v] = zz;
          s[v] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    uint32_t bufA[CH][NW], bufB[CH][NW];
    const int nfull = C / CH, ntail = C - nfull * CH;
    int c0 = 0, k = 0;
    if (nfull > 0) load_chunk(bufA, CH);
    // two chunks per trip so that both buffers are statically named (runtime-indexed register arrays go to scratch)
#pragma unroll 1
    for (; k + 2 <= nfull; k += 2) {
      load_chunk(bufB, CH);
      reduce_chunk(bufA, c0, CH);
      c0 += CH;
      if (k + 2 < nfull)
        load_chunk(bufA, CH);
      else if (ntail)
        load_chunk(bufA, ntail);
      reduce_chunk(bufB, c0, CH);
      c0 += CH;
    }
    if (k < nfull) {  // one full chunk left in bufA
      if (ntail) load_chunk(bufB, ntail);
      reduce_chunk(bufA, c0, CH);
      c0 += CH;
      if (ntail) reduce_chunk(bufB, c0, ntail);
    } else if (ntail) {  // the tail sits in bufA (not loaded yet when C < CH)
      if (nfull == 0) load_chunk(bufA, ntail);
      reduce_chunk(bufA, c0, ntail);
    }

    const bool need_js = (mode == SEA_MODE_JS) || (track_mode == SEA_MODE_JS);
    const bool need_w = (mode == SEA_MODE_MASK_CE_BAL) || (track_mode == SEA_MODE_MASK_CE_BAL);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (__builtin_expect(s[v] != s[v], 0))  // NaN / +-inf logits: torch.max semantics from memory (cold)
        arg[v] = slow_torch_argmax<T>(logits + (int64_t)b * C * HW + px0 + v, C, HW);
      const bool valid = lab[v] >= 0;
      const bool correct = valid && (arg[v] == lab[v]);
      const float lse = m[v] + __logf(s[v]);
      const float ce = lse - zy[v];
      const float logp = zy[v] - lse;
      float py = 0.f, l1p = 0.f;
      if (need_js) {
        py = __expf(logp);
        l1p = __logf(1.f + py);
      }
      const float wy = (need_w && valid) ? w[lab[v]] : 1.f;
      const float lv = loss_value(mode, valid, correct, ce, logp, py, l1p, wy);
      lsum += lv;
      tsum += (track_mode == mode) ? lv : loss_value(track_mode, valid, correct, ce, logp, py, l1p, wy);
      ncorr += correct ? 1 : 0;
      if (loss_px != nullptr) loss_px[(int64_t)b * HW + px0 + v] = lv;
    }
    if (pred != nullptr) {
      store_indices<VEC>(pred, pred_bytes, (int64_t)b * HW + px0, arg);
    }
  }
  block_reduce_store(lsum, tsum, ncorr, partials);
}

// streaming no-gradient kernel; variant: 0 = default, 1..3 = alternatives kept for A/B runs (tools/kernel_bench.py)
template <typename T>
void launch_fwd(const LossArgs& a, int variant) {
  constexpr int VEC = 16 / (int)sizeof(T);
  dim3 grid(tiles_for(a.HW, VEC), a.B), block(256);
#define SEA_FWD(CH, WV)                                                                                          \
  hipLaunchKernelGGL((loss_nchw_fwd<T, CH, WV>), grid, block, 0, a.s, (const T*)a.logits, a.y, a.y_bytes, a.w,   \
                     a.mode, a.track_mode, a.C, a.HW, a.pred, a.pred_bytes, a.loss_px, a.partials)
  if (variant == 1)
    SEA_FWD(4, 5);
  else if (variant == 2)
    SEA_FWD(8, 3);
  else if (variant == 3)
    SEA_FWD(6, 4);
  else if (variant == 4)
    SEA_FWD(2, 8);
  else if (sizeof(T) == 4)
    SEA_FWD(4, 5);   // 93 VGPRs, 5 waves/SIMD
  else
    SEA_FWD(4, 4);   // 16-bit: 8 pixels per lane, 127 VGPRs
#undef SEA_FWD
}

template void launch_fwd<float>(const LossArgs&, int);
template void launch_fwd<__hip_bfloat16>(const LossArgs&, int);
template void launch_fwd<__half>(const LossArgs&, int);

}  // namespace sea

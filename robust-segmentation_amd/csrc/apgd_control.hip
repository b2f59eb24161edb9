// K4 + K7: device-resident APGD bookkeeping.
//
// sea_apgd_track is one tiny launch (one thread per image) that replaces the per-iteration host
// logic of the reference (semseg/attacker.py:485-495, 520-551, 568-569), whose `.nonzero()`,
// `.sum() > 0` and `.cpu()` calls each force a host<->device round trip.  All predicates stay on
// the device as per-image flags; sea_select_copy then performs every conditional bulk copy of the
// iteration in a single HBM pass.
#include "sea_common.h"

namespace sea {

// must match BlockPartial of loss_kernels.hip
struct __attribute__((aligned(16))) LossRecord {
  float loss, track;
  int n_correct, pad;
};

__global__ __launch_bounds__(1024) void apgd_track_kernel(
    const float* __restrict__ loss_sum, const float* __restrict__ track_sum, const int32_t* __restrict__ n_correct,
    const int32_t* __restrict__ n_ignored, int B, int64_t HW, int iter, int n_iter, int check_k, int early_stop,
    int init, int32_t* __restrict__ acc_cnt, float* __restrict__ acc, float* __restrict__ loss_best,
    float* __restrict__ loss_best_last, float* __restrict__ reduced_last, float* __restrict__ step,
    float* __restrict__ loss_steps, uint8_t* __restrict__ flags, int32_t* __restrict__ done,
    const LossRecord* __restrict__ records, int32_t* __restrict__ iter_dev, const int32_t* __restrict__ check_table,
    const int32_t* __restrict__ n_iter_dev = nullptr) {
  // replayable form (HIP-graph mode): the loop index lives in *iter_dev, the checkpoint window of iteration i in
  // check_table[i]; the counter is advanced at the very end, after every use (K1 of this step has read it already).
  // With n_iter_dev the run length is device state too: one captured graph serves stages of any length.
  if (n_iter_dev != nullptr) n_iter = *n_iter_dev;
  if (iter_dev != nullptr) {
    iter = *iter_dev;
    iter = iter < 0 ? 0 : (iter >= n_iter ? n_iter - 1 : iter);
    check_k = check_table[iter];
  }
  __shared__ int s_any_nonzero;
  __shared__ float s_track[1024];
  __shared__ int s_corr[1024];
  if (threadIdx.x == 0) s_any_nonzero = 0;
  // Deferred K2 reduction: sum the per-block records of every image here (fixed order, double), so the
  // loop needs no separate finalize launch.  records[0] is the header {.,., tiles, images}.
  if (records != nullptr) {
    // Every image's records are split over wpi waves (B = 8 at C = 151: 2048 records per image, 16 waves -> 2 per
    // image; round 2 summed them with one wave per image, 4 images in flight: 32 us).  Lanes stride over a wave's
    // contiguous chunk, shuffle-reduce in double, and the wave partials are added in wave order: a fixed order, so the
    // sums are reproducible run to run.
    const int tiles = records[0].n_correct;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wpi = (B >= nw) ? 1 : nw / B;
    const int groups = nw / wpi;
    __shared__ double s_pt[16];
    __shared__ int s_pn[16];
    for (int b0 = 0; b0 < B && b0 < 1024; b0 += groups) {
      const int b = b0 + wave / wpi, part = wave % wpi;
      double t = 0.0;
      int n = 0;
      if (b < B && b < 1024 && wave < groups * wpi) {
        const int chunk = (tiles + wpi - 1) / wpi;
        const int i1 = (part + 1) * chunk < tiles ? (part + 1) * chunk : tiles;
        for (int i = part * chunk + lane; i < i1; i += 64) {
          const LossRecord r = records[1 + (int64_t)b * tiles + i];
          t += (double)r.track;
          n += r.n_correct;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          t += __shfl_down(t, o, 64);
          n += __shfl_down(n, o, 64);
        }
      }
      if (lane == 0) {
        s_pt[wave] = t;
        s_pn[wave] = n;
      }
      __syncthreads();
      if (part == 0 && lane == 0 && b < B && b < 1024 && wave < groups * wpi) {
        double tt = 0.0;
        int nn = 0;
        for (int w = 0; w < wpi; ++w) {
          tt += s_pt[wave + w];
          nn += s_pn[wave + w];
        }
        s_track[b] = (float)tt;
        s_corr[b] = nn;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  const bool frozen = (*done != 0);
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    uint8_t f_adv = 0, f_best = 0, f_restart = 0;
    if (!frozen) {
      // per-image mean over ALL pixels (pixel_to_img_loss, attacker.py:237-240)
      const float track = (records ? s_track[b] : track_sum[b]) / (float)HW;
      if (init) {
        // step 0 (attacker.py:370-383): ignored pixels count as wrong, everything is "best so far"
        const int cnt = records ? s_corr[b] : n_correct[b];
        acc_cnt[b] = cnt;
        acc[b] = (float)cnt / (float)HW;
        loss_best[b] = track;
        loss_best_last[b] = track;
        reduced_last[b] = 1.f;
      } else {
        // best-adv tracking (attacker.py:485-495): ignored pixels count as correct; <= keeps the latest
        const int cnt = (records ? s_corr[b] : n_correct[b]) + n_ignored[b];
        const int best = acc_cnt[b];
        if (cnt <= best) {
          f_adv = 1;
          acc_cnt[b] = cnt;
          acc[b] = (float)cnt / (float)HW;
        }
        // best-loss tracking on the tracking loss, strict > (attacker.py:520-526)
        loss_steps[(int64_t)iter * B + b] = track;
        float lb = loss_best[b];
        if (track > lb) {
          f_best = 1;
          lb = track;
          loss_best[b] = lb;
        }
        if (check_k > 0) {
          // oscillation check over the last k steps (attacker.py:243-248); rows wrap like Python
          // negative indices
          int t = 0;
          for (int c = 0; c < check_k; ++c) {
            int r1 = (iter - c) % n_iter;
            if (r1 < 0) r1 += n_iter;
            int r0 = (iter - c - 1) % n_iter;
            if (r0 < 0) r0 += n_iter;
            const float a1 = (r1 == iter) ? track : loss_steps[(int64_t)r1 * B + b];
            const float a0 = (r0 == iter) ? track : loss_steps[(int64_t)r0 * B + b];
            t += (a1 > a0) ? 1 : 0;
          }
          const float osc = ((float)t <= (float)((double)check_k * 0.75)) ? 1.f : 0.f;
          const float no_impr = (1.f - reduced_last[b]) * ((loss_best_last[b] >= lb) ? 1.f : 0.f);
          const float fl = fmaxf(osc, no_impr);
          reduced_last[b] = fl;
          loss_best_last[b] = lb;
          if (fl > 0.f) {
            step[b] = step[b] / 2.f;
            f_restart = 1;
          }
        }
      }
      if (acc_cnt[b] != 0) atomicOr(&s_any_nonzero, 1);
    }
    flags[b] = f_adv;
    flags[B + b] = f_best;
    flags[2 * B + b] = f_restart;
  }
  __syncthreads();
  // early stop: every image has zero pixel accuracy (attacker.py:568-569).  Not evaluated at
  // step 0: the reference only tests inside the loop.
  if (threadIdx.x == 0 && !frozen && early_stop && !init && s_any_nonzero == 0) *done = 1;
  if (threadIdx.x == 0 && iter_dev != nullptr) *iter_dev = iter + 1;
}

// one image per blockIdx.y; flags are wave-uniform scalars
__global__ __launch_bounds__(256) void select_copy_v4(const uint8_t* __restrict__ flags, float4* __restrict__ x_adv,
                                                      float4* __restrict__ grad, float4* __restrict__ x_best,
                                                      float4* __restrict__ grad_best,
                                                      float4* __restrict__ x_best_adv, int B, int64_t n4) {
  const int b = blockIdx.y;
  const bool f_adv = flags[b] != 0, f_best = flags[B + b] != 0, f_restart = flags[2 * B + b] != 0;
  if (!(f_adv || f_best || f_restart)) return;
  const int64_t base = (int64_t)b * n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = base + i;
    if (f_adv || f_best) {
      const float4 v = x_adv[k];
      if (f_adv) x_best_adv[k] = v;
      if (f_best) {
        const float4 g = grad[k];
        x_best[k] = v;
        grad_best[k] = g;
        // restart target == the values just read: nothing to write back
      }
    }
    if (f_restart && !f_best) {
      x_adv[k] = x_best[k];
      grad[k] = grad_best[k];
    }
  }
}

__global__ __launch_bounds__(256) void select_copy_v1(const uint8_t* __restrict__ flags, float* __restrict__ x_adv,
                                                      float* __restrict__ grad, float* __restrict__ x_best,
                                                      float* __restrict__ grad_best, float* __restrict__ x_best_adv,
                                                      int B, int64_t n) {
  const int b = blockIdx.y;
  const bool f_adv = flags[b] != 0, f_best = flags[B + b] != 0, f_restart = flags[2 * B + b] != 0;
  if (!(f_adv || f_best || f_restart)) return;
  const int64_t base = (int64_t)b * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = base + i;
    if (f_adv || f_best) {
      const float v = x_adv[k];
      if (f_adv) x_best_adv[k] = v;
      if (f_best) {
        x_best[k] = v;
        grad_best[k] = grad[k];
      }
    }
    if (f_restart && !f_best) {
      x_adv[k] = x_best[k];
      grad[k] = grad_best[k];
    }
  }
}

// pred -> pred_best for images with flag[0]; byte-wise copy of (HW * pred_bytes) per image
__global__ __launch_bounds__(256) void select_copy_pred(const uint8_t* __restrict__ flags,
                                                        const unsigned char* __restrict__ pred,
                                                        unsigned char* __restrict__ pred_best, int64_t bytes_per_img) {
  const int b = blockIdx.y;
  if (flags[b] == 0) return;
  const int64_t base = (int64_t)b * bytes_per_img;
  if ((bytes_per_img % 16) == 0 && ((((uintptr_t)pred) | ((uintptr_t)pred_best)) & 15) == 0) {
    const uint4* s = (const uint4*)(pred + base);
    uint4* d = (uint4*)(pred_best + base);
    const int64_t n16 = bytes_per_img / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
      d[i] = s[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes_per_img;
         i += (int64_t)gridDim.x * blockDim.x)
      pred_best[base + i] = pred[base + i];
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace sea

using namespace sea;

// deferred mode (K7 sums K2's records itself): 16 waves; with reduced inputs the kernel is a few per-image scalars
static inline int track_block(const float* track_sum, const int32_t* n_correct) {
  return (track_sum && n_correct) ? 256 : 1024;
}

extern "C" int sea_apgd_track(const float* loss_sum, const float* track_sum, const int32_t* n_correct,
                              const int32_t* n_ignored, int B, int64_t HW, int iter, int n_iter, int check_k,
                              int early_stop, int init, int32_t* acc_cnt, float* acc, float* loss_best,
                              float* loss_best_last, float* reduced_last, float* step, float* loss_steps,
                              uint8_t* flags, int32_t* done, const void* loss_workspace, void* stream) {
  SEA_CHECK_ARG(acc_cnt && acc && loss_best && loss_best_last && reduced_last && step && flags && done && B > 0 &&
                HW > 0);
  // either the reduced K2 outputs or the K2 workspace holding the per-block records (deferred mode)
  SEA_CHECK_ARG((track_sum && n_correct) || (loss_workspace && B <= 1024));
  SEA_CHECK_ARG(init || (n_ignored && loss_steps && iter >= 0 && n_iter > 0 && iter < n_iter));
  SEA_CHECK_ARG(check_k >= 0);
  hipLaunchKernelGGL(apgd_track_kernel, dim3(1), dim3(track_block(track_sum, n_correct)), 0, (hipStream_t)stream, loss_sum,
                     track_sum, n_correct, n_ignored, B, HW, iter, n_iter, check_k, early_stop, init, acc_cnt, acc, loss_best,
                     loss_best_last, reduced_last, step, loss_steps, flags, done,
                     (track_sum && n_correct) ? (const LossRecord*)nullptr : (const LossRecord*)loss_workspace,
                     (int32_t*)nullptr, (const int32_t*)nullptr);
  SEA_RETURN_LAST();
}

static int track_graph_impl(const float* loss_sum, const float* track_sum, const int32_t* n_correct, const int32_t* n_ignored, int B,
                            int64_t HW, int32_t* iter_dev, const int32_t* check_table, int n_iter, const int32_t* n_iter_dev,
                            int early_stop, int32_t* acc_cnt, float* acc, float* loss_best, float* loss_best_last,
                            float* reduced_last, float* step, float* loss_steps, uint8_t* flags, int32_t* done,
                            const void* loss_workspace, void* stream) {
  SEA_CHECK_ARG(acc_cnt && acc && loss_best && loss_best_last && reduced_last && step && flags && done && B > 0 &&
                HW > 0 && iter_dev && check_table && n_ignored && loss_steps && n_iter > 0);
  SEA_CHECK_ARG((track_sum && n_correct) || (loss_workspace && B <= 1024));
  hipLaunchKernelGGL(apgd_track_kernel, dim3(1), dim3(track_block(track_sum, n_correct)), 0, (hipStream_t)stream, loss_sum,
                     track_sum, n_correct, n_ignored, B, HW, 0, n_iter, 0, early_stop, 0, acc_cnt, acc, loss_best, loss_best_last, reduced_last,
                     step, loss_steps, flags, done,
                     (track_sum && n_correct) ? (const LossRecord*)nullptr : (const LossRecord*)loss_workspace, iter_dev,
                     check_table, n_iter_dev);
  SEA_RETURN_LAST();
}

extern "C" int sea_apgd_track_graph(const float* loss_sum, const float* track_sum, const int32_t* n_correct,
                                    const int32_t* n_ignored, int B, int64_t HW, int32_t* iter_dev,
                                    const int32_t* check_table, int n_iter, int early_stop, int32_t* acc_cnt, float* acc,
                                    float* loss_best, float* loss_best_last, float* reduced_last, float* step,
                                    float* loss_steps, uint8_t* flags, int32_t* done, const void* loss_workspace,
                                    void* stream) {
  return track_graph_impl(loss_sum, track_sum, n_correct, n_ignored, B, HW, iter_dev, check_table, n_iter, nullptr, early_stop,
                          acc_cnt, acc, loss_best, loss_best_last, reduced_last, step, loss_steps, flags, done, loss_workspace,
                          stream);
}

// the same with the run length read from device memory (check_table and loss_steps sized for the longest run the caller
// will replay): a captured graph then serves runs of any length
extern "C" int sea_apgd_track_graph_dev(const float* loss_sum, const float* track_sum, const int32_t* n_correct,
                                        const int32_t* n_ignored, int B, int64_t HW, int32_t* iter_dev,
                                        const int32_t* check_table, const int32_t* n_iter_dev, int early_stop, int32_t* acc_cnt,
                                        float* acc, float* loss_best, float* loss_best_last, float* reduced_last, float* step,
                                        float* loss_steps, uint8_t* flags, int32_t* done, const void* loss_workspace,
                                        void* stream) {
  SEA_CHECK_ARG(n_iter_dev != nullptr);
  return track_graph_impl(loss_sum, track_sum, n_correct, n_ignored, B, HW, iter_dev, check_table, 1, n_iter_dev, early_stop,
                          acc_cnt, acc, loss_best, loss_best_last, reduced_last, step, loss_steps, flags, done, loss_workspace,
                          stream);
}

extern "C" int sea_select_copy(const uint8_t* flags, float* x_adv, float* grad, float* x_best, float* grad_best,
                               float* x_best_adv, const void* pred, void* pred_best, int pred_bytes, int B,
                               int64_t n_per_img, int64_t HW, void* stream) {
  SEA_CHECK_ARG(flags && x_adv && grad && x_best && grad_best && x_best_adv && B > 0 && B <= 65535 && n_per_img > 0);
  hipStream_t s = (hipStream_t)stream;
  int cap = kMaxGridX / B;
  if (cap < 1) cap = 1;
  if ((n_per_img % 4) == 0 && aligned16(x_adv) && aligned16(grad) && aligned16(x_best) && aligned16(grad_best) &&
      aligned16(x_best_adv)) {
    int gx = grid_for(n_per_img / 4, 256);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(select_copy_v4, dim3(gx, B), dim3(256), 0, s, flags, (float4*)x_adv, (float4*)grad,
                       (float4*)x_best, (float4*)grad_best, (float4*)x_best_adv, B, n_per_img / 4);
  } else {
    int gx = grid_for(n_per_img, 256);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(select_copy_v1, dim3(gx, B), dim3(256), 0, s, flags, x_adv, grad, x_best, grad_best,
                       x_best_adv, B, n_per_img);
  }
  if (pred && pred_best) {
    SEA_CHECK_ARG(HW > 0 && (pred_bytes == 8 || pred_bytes == 4 || pred_bytes == 2 || pred_bytes == 1));
    const int64_t bytes = HW * pred_bytes;
    int gx = grid_for(bytes / 16 + 1, 256);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(select_copy_pred, dim3(gx, B), dim3(256), 0, s, flags, (const unsigned char*)pred,
                       (unsigned char*)pred_best, bytes);
  }
  SEA_RETURN_LAST();
}

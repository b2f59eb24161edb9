// K3: per-class integer statistics (intersection / prediction / target counts, confusion matrix)
// and the ignored-pixel count.  HBM-bound byte/integer work: pred + label read once, histograms
// privatised in LDS per workgroup, flushed with integer global atomics (order independent => exact
// and deterministic).  Outputs are int64 and are accumulated into (+=).
#include "sea_common.h"

namespace sea {

// grid = (chunks, B).  LDS: 3*C int32 (inter | pred_cnt | tgt_cnt).
__global__ __launch_bounds__(256) void class_counts_kernel(const void* __restrict__ pred, int pred_bytes,
                                                           const void* __restrict__ y, int y_bytes, int C,
                                                           int64_t HW, int mask_pred, int per_image,
                                                           unsigned long long* __restrict__ inter,
                                                           unsigned long long* __restrict__ pred_cnt,
                                                           unsigned long long* __restrict__ tgt_cnt) {
  extern __shared__ __attribute__((aligned(16))) int hist[];
  int* h_int = hist;
  int* h_prd = hist + C;
  int* h_tgt = hist + 2 * C;
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = load_label_rt(y, y_bytes, base + i);   // -1 = ignored
    const int p = load_label_rt(pred, pred_bytes, base + i);
    const bool tv = (t >= 0 && t < C);
    if (tv) {
      atomicAdd(&h_tgt[t], 1);
      if (p == t) atomicAdd(&h_int[t], 1);
    }
    // ignored = the label equals the ignore value (-1); out-of-range labels are not "ignored"
    const bool count_pred = (p >= 0 && p < C) && !(mask_pred && t == -1);
    if (count_pred) atomicAdd(&h_prd[p], 1);
  }
  __syncthreads();
  const int64_t ob = per_image ? (int64_t)b * C : 0;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    if (h_int[c]) atomicAdd(&inter[ob + c], (unsigned long long)h_int[c]);
    if (h_prd[c]) atomicAdd(&pred_cnt[ob + c], (unsigned long long)h_prd[c]);
    if (h_tgt[c]) atomicAdd(&tgt_cnt[ob + c], (unsigned long long)h_tgt[c]);
  }
}

// confusion matrix, LDS-privatised when C*C int32 fits, otherwise straight global atomics
template <bool USE_LDS>
__global__ __launch_bounds__(256) void confusion_kernel(const void* __restrict__ pred, int pred_bytes,
                                                        const void* __restrict__ y, int y_bytes, int64_t n, int C,
                                                        unsigned long long* __restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) int lh[];
  const int CC = C * C;
  if (USE_LDS) {
    for (int i = threadIdx.x; i < CC; i += blockDim.x) lh[i] = 0;
    __syncthreads();
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = load_label_rt(y, y_bytes, i);
    const int p = load_label_rt(pred, pred_bytes, i);
    if (t >= 0 && t < C && p >= 0 && p < C) {
      if (USE_LDS)
        atomicAdd(&lh[t * C + p], 1);
      else
        atomicAdd(&hist[t * C + p], 1ull);
    }
  }
  if (USE_LDS) {
    __syncthreads();
    for (int i = threadIdx.x; i < CC; i += blockDim.x)
      if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
  }
}

// n_ignored[b] = #{y[b,:] == ignore}; one block per image chunk, int atomics
__global__ __launch_bounds__(256) void count_ignored_kernel(const void* __restrict__ y, int y_bytes, int64_t HW,
                                                            int32_t* __restrict__ n_ignored) {
  __shared__ int s[4];
  const int b = blockIdx.y;
  int n = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x)
    n += (load_label_rt(y, y_bytes, (int64_t)b * HW + i) == -1) ? 1 : 0;
  n = wave_sum_i(n);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = s[0] + s[1] + s[2] + s[3];
    if (t) atomicAdd(&n_ignored[b], t);
  }
}

static inline bool width_ok(int b) { return b == 8 || b == 4 || b == 2 || b == 1; }

}  // namespace sea

using namespace sea;

extern "C" int sea_class_counts(const void* pred, int pred_bytes, const void* y, int y_bytes, int B, int C,
                                int64_t HW, int mask_pred, int per_image, int64_t* inter, int64_t* pred_cnt,
                                int64_t* tgt_cnt, void* stream) {
  SEA_CHECK_ARG(pred && y && inter && pred_cnt && tgt_cnt && B > 0 && B <= 65535 && C > 0 && HW > 0);
  SEA_CHECK_ARG(width_ok(pred_bytes) && width_ok(y_bytes));
  const size_t lds = (size_t)3 * C * sizeof(int);
  SEA_CHECK_ARG(lds <= 64 * 1024);
  int gx = grid_for(HW, 256 * 16);  // >= 16 pixels per thread so the LDS flush amortises
  int cap = kMaxGridX / B;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(class_counts_kernel, dim3(gx, B), dim3(256), lds, (hipStream_t)stream, pred, pred_bytes, y,
                     y_bytes, C, HW, mask_pred, per_image, (unsigned long long*)inter,
                     (unsigned long long*)pred_cnt, (unsigned long long*)tgt_cnt);
  SEA_RETURN_LAST();
}

extern "C" int sea_confusion(const void* pred, int pred_bytes, const void* y, int y_bytes, int64_t n, int C,
                             int64_t* hist, void* stream) {
  SEA_CHECK_ARG(pred && y && hist && n > 0 && C > 0 && C <= 32767);
  SEA_CHECK_ARG(width_ok(pred_bytes) && width_ok(y_bytes));
  const size_t lds = (size_t)C * C * sizeof(int);
  const int g = grid_for(n, 256 * 32);
  if (lds <= 150 * 1024) {  // C <= 195: the K x K histogram is privatised in LDS (ADE20K's 151 classes need 91 KB)
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)confusion_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
    hipLaunchKernelGGL(confusion_kernel<true>, dim3(lds > 64 * 1024 ? (g > 512 ? 512 : g) : g), dim3(256), lds,
                       (hipStream_t)stream, pred, pred_bytes, y, y_bytes, n, C, (unsigned long long*)hist);
  } else
    hipLaunchKernelGGL(confusion_kernel<false>, dim3(g), dim3(256), 0, (hipStream_t)stream, pred, pred_bytes, y,
                       y_bytes, n, C, (unsigned long long*)hist);
  SEA_RETURN_LAST();
}

extern "C" int sea_count_ignored(const void* y, int y_bytes, int B, int64_t HW, int32_t* n_ignored, void* stream) {
  SEA_CHECK_ARG(y && n_ignored && B > 0 && B <= 65535 && HW > 0 && width_ok(y_bytes));
  hipError_t e = hipMemsetAsync(n_ignored, 0, sizeof(int32_t) * B, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  int gx = grid_for(HW, 256 * 8);
  int cap = kMaxGridX / B;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(count_ignored_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, y, y_bytes, HW,
                     n_ignored);
  SEA_RETURN_LAST();
}

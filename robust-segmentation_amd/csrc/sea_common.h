// Shared helpers for the gfx950 kernels of libsea_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sea_hip.h"

#define SEA_ERR_ARG 1  // == hipErrorInvalidValue

#define SEA_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return SEA_ERR_ARG; \
  } while (0)

#define SEA_RETURN_LAST()                \
  do {                                   \
    hipError_t e__ = hipGetLastError();  \
    return (int)e__;                     \
  } while (0)

namespace sea {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kMaxGridX = 256 * 8; // memory-bound launches: <= 8 blocks per CU, grid-stride beyond

static inline int grid_for(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > kMaxGridX) g = kMaxGridX;
  return (int)g;
}

// ---- XCD-aware work order for grid-stride kernels with spatial reuse -------------------------------------------------
// The dispatcher deals consecutive block ids round-robin to the 8 XCDs, each with a private 4 MB L2.  A kernel whose
// neighbouring work items read overlapping input (stencils, bilinear footprints, Winograd tiles) therefore fetches every
// shared pixel once per XCD from the fabric.  xcd_range() gives XCD k the k-th contiguous eighth of the linear index
// space (one image at B = 8) and lets the blocks that landed on that XCD stride through it: overlapping reads meet in
// one L2.  The launch must use grid_for_xcd() (a multiple of 8 blocks).  SEA_XCD_ORDER=0 restores the plain order.
static inline int xcd_order_enabled() {
  static const int on = [] {
    const char* e = getenv("SEA_XCD_ORDER");
    return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
  }();
  return on;
}

static inline int grid_for_xcd(int64_t work_items, int block) {
  const int g = grid_for(work_items, block);
  return (g + 7) / 8 * 8;
}

// ---- division by a launch-time constant: 3 instructions instead of the ~100 of a 64-bit (or ~25 of a 32-bit) integer
// division.  Granlund-Montgomery round-up method with the 33-bit magic 2^32 + m, s = ceil(log2 d); exact for every
// dividend below 2^31 (the sum below cannot wrap).  Kernels that decompose a linear work index use it whenever the
// index space is smaller than 2^31 (kFastIndexLimit) and fall back to 64-bit arithmetic otherwise.
struct FastDiv {
  uint32_t d, m, s;
};

static inline FastDiv fast_div(uint32_t d) {  // d >= 1
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.s = s;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv f) { return (__umulhi(n, f.m) + n) >> f.s; }

// linear index -> (c0, c1, c2, c3) with i = ((c3 * n2 + c2) * n1 + c1) * n0 + c0
struct Divs3 {
  FastDiv f0, f1, f2;
};

static inline Divs3 divs3(int64_t n0, int64_t n1, int64_t n2) {
  Divs3 d;
  d.f0 = fast_div((uint32_t)n0);
  d.f1 = fast_div((uint32_t)n1);
  d.f2 = fast_div((uint32_t)n2);
  return d;
}

struct Index4 {
  int c0, c1, c2;
  int64_t c3;
};

__device__ __forceinline__ Index4 split_index(int64_t i, const Divs3 d, bool fast) {
  Index4 r;
  if (fast) {  // wave-uniform: the whole index space is below 2^31
    const uint32_t n = (uint32_t)i, q0 = fdiv(n, d.f0), q1 = fdiv(q0, d.f1), q2 = fdiv(q1, d.f2);
    r.c0 = (int)(n - q0 * d.f0.d);
    r.c1 = (int)(q0 - q1 * d.f1.d);
    r.c2 = (int)(q1 - q2 * d.f2.d);
    r.c3 = q2;
  } else {
    int64_t p = i;
    r.c0 = (int)(p % d.f0.d);
    p /= d.f0.d;
    r.c1 = (int)(p % d.f1.d);
    p /= d.f1.d;
    r.c2 = (int)(p % d.f2.d);
    r.c3 = p / d.f2.d;
  }
  return r;
}

constexpr int64_t kFastIndexLimit = (1ll << 31) - (1ll << 24);  // room for the rounding of xcd_range chunks

struct IndexRange {
  int64_t begin, end, stride;
};

__device__ __forceinline__ IndexRange xcd_range(int64_t total, int on) {
  IndexRange r;
  if (!on) {
    r.begin = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    r.end = total;
    r.stride = (int64_t)gridDim.x * blockDim.x;
    return r;
  }
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int64_t chunk = (((total + 7) >> 3) + blockDim.x - 1) / blockDim.x * blockDim.x;
  const int64_t b0 = xcd * chunk;
  r.begin = b0 + (int64_t)local * blockDim.x + threadIdx.x;
  r.end = b0 + chunk < total ? b0 + chunk : total;
  r.stride = (int64_t)per * blockDim.x;
  return r;
}

// label loads: any integer width; ignore label = -1 (255 for uint8); returns -1 for ignored
template <int BYTES>
__device__ __forceinline__ int load_label(const void* y, int64_t i);
template <>
__device__ __forceinline__ int load_label<8>(const void* y, int64_t i) {
  long long v = ((const long long*)y)[i];
  return (v < 0 || v > 0x7fffffffLL) ? -1 : (int)v;
}
template <>
__device__ __forceinline__ int load_label<4>(const void* y, int64_t i) {
  return ((const int*)y)[i];
}
template <>
__device__ __forceinline__ int load_label<2>(const void* y, int64_t i) {
  return (int)((const short*)y)[i];
}
template <>
__device__ __forceinline__ int load_label<1>(const void* y, int64_t i) {
  int v = (int)((const unsigned char*)y)[i];
  return v == 255 ? -1 : v;
}

__device__ __forceinline__ int load_label_rt(const void* y, int bytes, int64_t i) {
  switch (bytes) {
    case 8: return load_label<8>(y, i);
    case 4: return load_label<4>(y, i);
    case 2: return load_label<2>(y, i);
    default: return load_label<1>(y, i);
  }
}

// VEC consecutive labels starting at element i (one wave-uniform switch on the width).  One-byte labels (what the
// attack uses: labels are compacted to uint8 once per run) come as whole dwords when VEC % 4 == 0: one
// global_load_dword(x2) per lane instead of VEC global_load_ubyte (8 extra memory instructions per lane next to
// 21 plane loads were the largest fixed cost of the 16-bit kernels at C=21).
template <int VEC>
__device__ __forceinline__ void load_labels(const void* y, int bytes, int64_t i, int (&lab)[VEC]) {
  if (bytes == 8) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) lab[v] = load_label<8>(y, i + v);
  } else if (bytes == 4) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) lab[v] = load_label<4>(y, i + v);
  } else if (bytes == 2) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) lab[v] = load_label<2>(y, i + v);
  } else {
    if constexpr (VEC % 4 == 0) {
      // i % VEC == 0 for every caller (tile origins and H*W are multiples of VEC), so the alignment of the access
      // is that of the base pointer: a wave-uniform test, no divergent fallback
      if ((((uintptr_t)y) & 3) == 0) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(y) + i);
#pragma unroll
        for (int g = 0; g < VEC / 4; ++g) {
          const uint32_t w = p[g];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int v = (int)((w >> (8 * k)) & 0xffu);
            lab[4 * g + k] = v == 255 ? -1 : v;
          }
        }
        return;
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) lab[v] = load_label<1>(y, i + v);
  }
}

__device__ __forceinline__ void store_index_rt(void* p, int bytes, int64_t i, int v) {
  switch (bytes) {
    case 8: ((long long*)p)[i] = (long long)v; break;
    case 4: ((int*)p)[i] = v; break;
    case 2: ((short*)p)[i] = (short)v; break;
    default: ((unsigned char*)p)[i] = (unsigned char)v; break;
  }
}

// VEC consecutive indices (argmax map); one-byte outputs are packed into dword stores when VEC % 4 == 0
template <int VEC>
__device__ __forceinline__ void store_indices(void* p, int bytes, int64_t i, const int (&val)[VEC]) {
  if constexpr (VEC % 4 == 0) {
    if (bytes == 1 && (((uintptr_t)p) & 3) == 0) {  // i % VEC == 0 for every caller
      uint32_t* q = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(p) + i);
#pragma unroll
      for (int g = 0; g < VEC / 4; ++g)
        q[g] = (uint32_t)(val[4 * g] & 0xff) | ((uint32_t)(val[4 * g + 1] & 0xff) << 8) |
               ((uint32_t)(val[4 * g + 2] & 0xff) << 16) | ((uint32_t)(val[4 * g + 3] & 0xff) << 24);
      return;
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) store_index_rt(p, bytes, i + v, val[v]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

}  // namespace sea

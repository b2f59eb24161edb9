// Shared helpers for the gfx950 kernels of libsea_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

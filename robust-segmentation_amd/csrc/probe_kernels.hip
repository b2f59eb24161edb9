// Measurement probes (not on the product path): the HBM copy ceiling of THIS device, measured with the library's
// own streaming idiom so that bench.py can quote the loss kernels against something they could actually reach
// (MI355X_MICROARCH.md: 6.29 TB/s for a float4 copy, 79 % of the 8 TB/s spec peak).
#include "sea_common.h"

namespace sea {

typedef float f4 __attribute__((ext_vector_type(4)));

// 16 B per lane per access, UNROLL independent accesses in flight per lane.  Every block streams ONE contiguous
// slice of the buffer (consecutive 4 KiB rows of 256 lanes x 16 B): DRAM pages are opened once and consumed whole.
// NT = non-temporal loads and stores (streamed once: do not displace other lines from L2 / the Infinity Cache).
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, int64_t n4) {
  const int64_t per = ((n4 + gridDim.x - 1) / gridDim.x + 256 * UNROLL - 1) / (256 * UNROLL) * (256 * UNROLL);
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < n4 ? lo + per : n4;
  int64_t i = lo + threadIdx.x;
  for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT)
        __builtin_nontemporal_store(v[u], dst + i + u * 256);
      else
        dst[i + u * 256] = v[u];
    }
  }
  for (; i < hi; i += 256) dst[i] = src[i];
}

// read-only probe: sums every float4 (the store below practically never executes; it keeps the loads alive)
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read_kernel(const f4* __restrict__ src, float* __restrict__ sink, int64_t n4) {
  const int64_t per = ((n4 + gridDim.x - 1) / gridDim.x + 256 * UNROLL - 1) / (256 * UNROLL) * (256 * UNROLL);
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < n4 ? lo + per : n4;
  int64_t i = lo + threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  for (; i < hi; i += 256) acc += src[i];
  float s = wave_sum(acc.x + acc.y + acc.z + acc.w);
  if ((threadIdx.x & 63) == 0 && s == 12345.678f) sink[blockIdx.x & 2047] = s;
}

}  // namespace sea

using namespace sea;

extern "C" int sea_probe_stream_copy(const void* src, void* dst, size_t bytes, int non_temporal, void* stream) {
  SEA_CHECK_ARG(src && dst && bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
  const int64_t n4 = (int64_t)(bytes / 16);
  // non_temporal: bit 0 = nt accesses; bits 8.. = blocks per CU (default 16: 5.64 TB/s on MI355X; 8: 5.51, 32: 5.60)
  const int per_cu = (non_temporal >> 8) > 0 ? (non_temporal >> 8) : 16;
  int64_t g = (n4 + 256 * 8 - 1) / (256 * 8);
  const int grid = (int)(g < 256 * per_cu ? (g < 1 ? 1 : g) : 256 * per_cu);
  if (non_temporal & 1)
    hipLaunchKernelGGL((stream_copy_kernel<8, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f4*)src, (f4*)dst, n4);
  else
    hipLaunchKernelGGL((stream_copy_kernel<8, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f4*)src, (f4*)dst, n4);
  SEA_RETURN_LAST();
}

extern "C" int sea_probe_stream_read(const void* src, float* sink, size_t bytes, void* stream) {
  SEA_CHECK_ARG(src && sink && bytes % 16 == 0 && (((uintptr_t)src) & 15) == 0);
  const int64_t n4 = (int64_t)(bytes / 16);
  hipLaunchKernelGGL((stream_read_kernel<8>), dim3(grid_for(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const f4*)src, sink, n4);
  SEA_RETURN_LAST();
}

// The magic numbers the kernels divide work indices with (sea_common.h: FastDiv), for the host-side unit test:
// for every n < 2^31:  n / d == (mulhi32(n, *m) + n) >> *s.
extern "C" int sea_fastdiv_magic(uint32_t d, uint32_t* m, uint32_t* s) {
  SEA_CHECK_ARG(d >= 1 && m && s);
  const sea::FastDiv f = sea::fast_div(d);
  *m = f.m;
  *s = f.s;
  return 0;
}

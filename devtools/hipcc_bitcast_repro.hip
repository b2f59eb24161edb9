// hipcc 7.2 (-O3 --offload-arch=gfx950): bit_cast<f16x2>(hi2[1]) of a uint32_t 2-vector is compiled to the halves of hi2[0].
//   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only devtools/hipcc_bitcast_repro.hip -o - | grep -E "v_cvt_f32_f16|v_pk_fma|v_fma"
// Expected: four v_cvt_f32_f16 (both packed words are converted back); observed: two, and the second v_pk_fma_f32 subtracts the
// first word's halves again.  With `uint32_t h0, h1` instead of the 2-vector the code is right (what the product kernels do).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__global__ void k(const f32x4* in, u32x2* out, float sc) {
  f32x4 v = in[threadIdx.x];
  for (int e = 0; e < 4; ++e) v[e] *= sc;
  u32x2 hi2 = u32x2{pack_f16(v[0], v[1]), pack_f16(v[2], v[3])};
  f32x4 sv = v;
  const f32x2 f0 = __builtin_convertvector(__builtin_bit_cast(f16x2, hi2[0]), f32x2);
  const f32x2 f1 = __builtin_convertvector(__builtin_bit_cast(f16x2, hi2[1]), f32x2);
  sv[0] -= f0[0]; sv[1] -= f0[1]; sv[2] -= f1[0]; sv[3] -= f1[1];
  out[2 * threadIdx.x] = hi2;
  out[2 * threadIdx.x + 1] = u32x2{pack_f16(sv[0], sv[1]), pack_f16(sv[2], sv[3])};
}

// Probe (round 6): (1) is v_mfma_f32_32x32x16_f16 symmetric under an operand swap, bit for bit (D(A,B)^T == D(B,A))?  The fused
// MLP kernel computes the hidden tile transposed; (2) semantics of v_permlane32_swap as hipcc's builtin exposes them.
//   hipcc --offload-arch=gfx950 -O3 devtools/probes/mfma_swap_probe.hip -o /tmp/mfma_swap_probe && /tmp/mfma_swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: 32 x 16 (row-major), B: 32 x 16 (row n, k) -> D[m][n] = sum_k A[m][k] B[n][k], chained over `steps` k-blocks
__global__ void k_mfma(const _Float16* A, const _Float16* B, int steps, float* D1, float* D2) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 c1 = {}, c2 = {};
  for (int s = 0; s < steps; ++s) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
      a[e] = A[(s * 32 + r) * 16 + 8 * h + e];
      b[e] = B[(s * 32 + r) * 16 + 8 * h + e];
    }
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);   // D1[i][j]: i = rows of A
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0);   // D2[i][j]: i = rows of B  (= D1^T ?)
  }
  for (int e = 0; e < 16; ++e) {
    const int i = (e & 3) + 8 * (e >> 2) + 4 * h;   // accumulator: lane & 31 = column j, register -> row i
    D1[i * 32 + r] = c1[e];
    D2[i * 32 + r] = c2[e];
  }
}
__global__ void k_swap(unsigned* out) {
  const unsigned x = 1000 + threadIdx.x, y = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  const int steps = 24;
  std::vector<_Float16> A(steps * 32 * 16), B(steps * 32 * 16);
  srand(7);
  for (auto& v : A) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 8.f);
  for (auto& v : B) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.37f);
  _Float16 *dA, *dB; float *d1, *d2; unsigned* ds;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&d1, 4096); hipMalloc(&d2, 4096); hipMalloc(&ds, 512);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, steps, d1, d2);
  k_swap<<<1, 64>>>(ds);
  float h1[1024], h2[1024]; unsigned hs[128];
  hipMemcpy(h1, d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 4096, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 512, hipMemcpyDeviceToHost);
  int diff = 0; double maxd = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    if (memcmp(&h1[i * 32 + j], &h2[j * 32 + i], 4)) { ++diff; double d = fabs((double)h1[i * 32 + j] - h2[j * 32 + i]); if (d > maxd) maxd = d; }
  }
  printf("mfma operand swap: %d of 1024 elements differ bitwise (max |d| %.3g); sample %g %g\n", diff, maxd, h1[5], h2[5 * 32]);
  printf("permlane32_swap(x=1000+l, y=2000+l): r0[0]=%u r0[31]=%u r0[32]=%u r0[63]=%u | r1[0]=%u r1[31]=%u r1[32]=%u r1[63]=%u\n", hs[0], hs[31],
         hs[32], hs[63], hs[64], hs[95], hs[96], hs[127]);
  printf("expected if r0 = {x.lo, y.lo}, r1 = {x.hi, y.hi}: 1000 1031 2000 2031 | 1032 1063 2032 2063\n");
  return 0;
}

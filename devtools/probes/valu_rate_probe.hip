// Probe (round 6): what one SIMD of gfx950 sustains in fp32 vector instructions per cycle -- v_fma_f32 vs v_pk_fma_f32, one and
// two waves per SIMD (s_memtime around a long independent stream).  The fused MLP kernel is bound by its ~45 vector instructions
// per hidden element; whether packed fp32 halves that is decided here.
//   hipcc --offload-arch=gfx950 -O3 devtools/probes/valu_rate_probe.hip -o devtools/probes/valu_rate_probe && devtools/probes/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int reps) {
  float a[8]; f32x2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 0.5f}; }
  const float c = 1.0001f, d = 0.0003f;
  const f32x2 c2 = {c, c}, d2 = {d, d};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
        if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        if (MODE == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (MODE == 5) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800123" : "+v"(a[i]) : "v"(c));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
void run(const char* name, int waves) {
  float* out; unsigned long long* t;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&t, 256 * 8 * 8);
  const int reps = 2000;
  k<MODE><<<256, waves * 64>>>(out, t, reps);
  k<MODE><<<256, waves * 64>>>(out, t, reps);
  hipDeviceSynchronize();
  unsigned long long h[2048];
  hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; int n = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { s += h[b * 8 + w]; ++n; }
  const double per = s / n / (reps * 64.0);
  printf("%-16s %d wave(s) per SIMD: %.2f cycles per instruction per wave -> %.2f cycles per instruction per SIMD\n", name, waves / 4, per,
         per / (waves / 4));
  hipFree(out); hipFree(t);
}
int main() {
  run<0>("v_fma_f32", 4); run<0>("v_fma_f32", 8);
  run<1>("v_pk_fma_f32", 4); run<1>("v_pk_fma_f32", 8);
  run<3>("v_pk_mul_f32", 4); run<3>("v_pk_mul_f32", 8);
  run<5>("v_fmaak_f32", 4); run<5>("v_fmaak_f32", 8);
  run<4>("v_cvt_pk_f16_f32", 4); run<4>("v_cvt_pk_f16_f32", 8);
  run<2>("v_exp_f32", 4); run<2>("v_exp_f32", 8);
  return 0;
}

// K1 / K5 / K6: element-wise L-inf arithmetic of the attack loop (gfx950).
//
// HBM-bound streaming kernels: 16 B per lane per access (global_load/store_dwordx4), grid capped
// at 8 blocks per CU with a grid-stride loop, no LDS.  This translation unit is compiled with
// -ffp-contract=off: every multiply/add below rounds exactly like the separate float32 ATen ops
// of the reference, which is what makes the outputs bit-identical to it.
#include "sea_common.h"

namespace sea {

__device__ __forceinline__ float sgn(float g) { return (float)((g > 0.f) - (g < 0.f)); }

// min(max(z, lo), hi) then clamp to [0,1]: torch.clamp(torch.min(torch.max(z, x-eps), x+eps), 0, 1)
__device__ __forceinline__ float box(float z, float x, float eps) {
  float lo = x - eps, hi = x + eps;
  z = fminf(fmaxf(z, lo), hi);
  return fminf(fmaxf(z, 0.f), 1.f);
}

__device__ __forceinline__ float apgd_elem(float x, float xa, float xo, float g, float st, float eps,
                                           float a, float one_minus_a) {
  float g2 = xa - xo;
  float z = xa + st * sgn(g);
  z = box(z, x, eps);
  float t = (z - xa) * a;
  t = xa + t;
  t = t + g2 * one_minus_a;
  return box(t, x, eps);
}

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 apgd_elem4(f4 vx, f4 va, f4 vo, f4 vg, float st, float eps, float a, float oma) {
  f4 r;
  r.x = apgd_elem(vx.x, va.x, vo.x, vg.x, st, eps, a, oma);
  r.y = apgd_elem(vx.y, va.y, vo.y, vg.y, st, eps, a, oma);
  r.z = apgd_elem(vx.z, va.z, vo.z, vg.z, st, eps, a, oma);
  r.w = apgd_elem(vx.w, va.w, vo.w, vg.w, st, eps, a, oma);
  return r;
}

// one image per blockIdx.y so the per-image step size is a scalar (SGPR) load.  Two grid-stride positions per trip:
// 8 independent 16-byte loads in flight per lane.  x_adv, x_old and grad are consumed exactly once per iteration
// (non-temporal loads: they need not displace the clean image x, which every iteration re-reads, or the output,
// which the model's first convolution reads next).
__global__ __launch_bounds__(256) void apgd_linf_step_v4(const f4* __restrict__ x, const f4* __restrict__ xadv,
                                                         const f4* __restrict__ xold, const f4* __restrict__ grad,
                                                         const float* __restrict__ step_b, float eps, float a, float oma,
                                                         f4* __restrict__ out, int64_t n4_per_img) {
  const int b = blockIdx.y;
  const float st = step_b[b];
  const int64_t base = (int64_t)b * n4_per_img;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_img; i += 2 * stride) {
    const int64_t j = i + stride;
    const bool hj = j < n4_per_img;
    const int64_t jj = hj ? j : i;
    const f4 vx0 = x[base + i], va0 = __builtin_nontemporal_load(xadv + base + i),
             vo0 = __builtin_nontemporal_load(xold + base + i), vg0 = __builtin_nontemporal_load(grad + base + i);
    const f4 vx1 = x[base + jj], va1 = __builtin_nontemporal_load(xadv + base + jj),
             vo1 = __builtin_nontemporal_load(xold + base + jj), vg1 = __builtin_nontemporal_load(grad + base + jj);
    out[base + i] = apgd_elem4(vx0, va0, vo0, vg0, st, eps, a, oma);
    if (hj) out[base + j] = apgd_elem4(vx1, va1, vo1, vg1, st, eps, a, oma);
  }
}

__global__ __launch_bounds__(256) void apgd_linf_step_v1(const float* __restrict__ x,
                                                         const float* __restrict__ xadv,
                                                         const float* __restrict__ xold,
                                                         const float* __restrict__ grad,
                                                         const float* __restrict__ step_b, float eps,
                                                         float a, float oma, float* __restrict__ out,
                                                         int64_t n_per_img) {
  const int b = blockIdx.y;
  const float st = step_b[b];
  const int64_t base = (int64_t)b * n_per_img;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_img;
       i += (int64_t)gridDim.x * blockDim.x)
    out[base + i] = apgd_elem(x[base + i], xadv[base + i], xold[base + i], grad[base + i], st, eps, a, oma);
}

// Replayable form of K1 (HIP-graph mode of the attack loop): every per-iteration scalar comes from device memory and
// the iterate buffers keep their addresses.  *iter_dev is the loop index i (a = 1 for i = 0, else 0.75; K7 advances it);
// the update is IN PLACE: x_old <- x_adv (the reference's `x_adv_old = x_adv.clone()`, attacker.py:390) and
// x_adv <- new iterate, element by element (each element is read before it is written, by the same lane).
__global__ __launch_bounds__(256) void apgd_linf_step_inplace_v4(const f4* __restrict__ x, f4* __restrict__ xadv,
                                                                 f4* __restrict__ xold, const f4* __restrict__ grad,
                                                                 const float* __restrict__ step_b, float eps_val,
                                                                 const int32_t* __restrict__ iter_dev,
                                                                 int64_t n4_per_img, const float* __restrict__ eps_dev) {
  const int b = blockIdx.y;
  const float eps = eps_dev ? *eps_dev : eps_val;   // (the radius as run-invariant device state: one captured graph serves every stage)
  const float st = step_b[b];
  const bool first = *iter_dev <= 0;
  const float a = first ? 1.0f : 0.75f;
  const float oma = first ? (float)(1.0 - 1.0) : (float)(1.0 - 0.75);  // (1 - a) in double, like the reference
  const int64_t base = (int64_t)b * n4_per_img;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_img; i += (int64_t)gridDim.x * blockDim.x) {
    const f4 vx = x[base + i], va = xadv[base + i], vo = xold[base + i], vg = __builtin_nontemporal_load(grad + base + i);
    xold[base + i] = va;
    xadv[base + i] = apgd_elem4(vx, va, vo, vg, st, eps, a, oma);
  }
}

// the same for images whose element count is not a multiple of 4 (e.g. PASCAL-VOC's 3 x 473 x 473: a per-image base is then
// not 16-byte aligned): one float per lane and trip
__global__ __launch_bounds__(256) void apgd_linf_step_inplace_v1(const float* __restrict__ x, float* __restrict__ xadv,
                                                                 float* __restrict__ xold, const float* __restrict__ grad,
                                                                 const float* __restrict__ step_b, float eps_val,
                                                                 const int32_t* __restrict__ iter_dev, int64_t n_per_img,
                                                                 const float* __restrict__ eps_dev) {
  const int b = blockIdx.y;
  const float eps = eps_dev ? *eps_dev : eps_val;
  const float st = step_b[b];
  const bool first = *iter_dev <= 0;
  const float a = first ? 1.0f : 0.75f;
  const float oma = first ? (float)(1.0 - 1.0) : (float)(1.0 - 0.75);
  const int64_t base = (int64_t)b * n_per_img;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_img; i += (int64_t)gridDim.x * blockDim.x) {
    const float va = xadv[base + i];
    const float r = apgd_elem(x[base + i], va, xold[base + i], grad[base + i], st, eps, a, oma);
    xold[base + i] = va;
    xadv[base + i] = r;
  }
}

// generic 2-input -> 1-output element-wise kernels, OP selected at compile time
enum { OP_RANDOM_START = 0, OP_PROJECT = 1 };

template <int OP>
__device__ __forceinline__ float ew2(float p, float q, float eps) {
  if (OP == OP_RANDOM_START) {  // p = x, q = u : clip(x + eps*(2u-1), 0, 1)
    float t = 2.f * q - 1.f;
    float z = p + eps * t;
    return fminf(fmaxf(z, 0.f), 1.f);
  } else {  // p = z, q = x : clip(x + clip(z-x, -eps, eps), 0, 1)
    float d = p - q;
    d = fminf(fmaxf(d, -eps), eps);
    float z = q + d;
    return fminf(fmaxf(z, 0.f), 1.f);
  }
}

// two grid-stride positions per trip (four independent 16-byte loads in flight per lane, like K1); the first operand is
// consumed once (non-temporal), the clean image stays cacheable for the model's first layer
template <int OP>
__global__ __launch_bounds__(256) void ew2_v4(const f4* __restrict__ p, const f4* __restrict__ q, float eps,
                                              f4* __restrict__ out, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
    const int64_t j = i + stride;
    const bool hj = j < n4;
    const int64_t jj = hj ? j : i;
    const f4 a0 = __builtin_nontemporal_load(p + i), b0 = q[i];
    const f4 a1 = __builtin_nontemporal_load(p + jj), b1 = q[jj];
    f4 r0, r1;
    r0.x = ew2<OP>(a0.x, b0.x, eps);
    r0.y = ew2<OP>(a0.y, b0.y, eps);
    r0.z = ew2<OP>(a0.z, b0.z, eps);
    r0.w = ew2<OP>(a0.w, b0.w, eps);
    r1.x = ew2<OP>(a1.x, b1.x, eps);
    r1.y = ew2<OP>(a1.y, b1.y, eps);
    r1.z = ew2<OP>(a1.z, b1.z, eps);
    r1.w = ew2<OP>(a1.w, b1.w, eps);
    out[i] = r0;
    if (hj) out[j] = r1;
  }
}
template <int OP>
__global__ __launch_bounds__(256) void ew2_v1(const float* __restrict__ p, const float* __restrict__ q,
                                              float eps, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = ew2<OP>(p[i], q[i], eps);
}

__device__ __forceinline__ void pgd_elem(float X, float d, float g, float alpha, float eps, int clamp_in,
                                         float& d_out, float& x_in) {
  if (clamp_in) {
    // the reference differentiates through clamp(X+delta, 0, 1) (val.py:150): zero gradient where the
    // unclamped input left [0,1] (possible by one ulp after the previous projection)
    const float xi0 = X + d;
    if (xi0 < 0.f || xi0 > 1.f) g = 0.f;
  }
  float t = d + alpha * sgn(g);
  float z = X + t;
  z = fminf(fmaxf(z, 0.f), 1.f);
  t = z - X;
  t = fminf(fmaxf(t, -eps), eps);
  d_out = t;
  float xi = X + t;
  x_in = clamp_in ? fminf(fmaxf(xi, 0.f), 1.f) : xi;
}

template <bool WRITE_X>
__global__ __launch_bounds__(256) void pgd_linf_step_v4(const float4* __restrict__ X,
                                                        const float4* __restrict__ delta,
                                                        const float4* __restrict__ grad, float alpha,
                                                        float eps, int clamp_in, float4* __restrict__ dout,
                                                        float4* __restrict__ xin, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 vx = X[i], vd = delta[i], vg = grad[i], rd, rx;
    pgd_elem(vx.x, vd.x, vg.x, alpha, eps, clamp_in, rd.x, rx.x);
    pgd_elem(vx.y, vd.y, vg.y, alpha, eps, clamp_in, rd.y, rx.y);
    pgd_elem(vx.z, vd.z, vg.z, alpha, eps, clamp_in, rd.z, rx.z);
    pgd_elem(vx.w, vd.w, vg.w, alpha, eps, clamp_in, rd.w, rx.w);
    dout[i] = rd;
    if (WRITE_X) xin[i] = rx;
  }
}
template <bool WRITE_X>
__global__ __launch_bounds__(256) void pgd_linf_step_v1(const float* __restrict__ X,
                                                        const float* __restrict__ delta,
                                                        const float* __restrict__ grad, float alpha,
                                                        float eps, int clamp_in, float* __restrict__ dout,
                                                        float* __restrict__ xin, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float rd, rx;
    pgd_elem(X[i], delta[i], grad[i], alpha, eps, clamp_in, rd, rx);
    dout[i] = rd;
    if (WRITE_X) xin[i] = rx;
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace sea

using namespace sea;

extern "C" int sea_apgd_linf_step(const float* x, const float* x_adv, const float* x_old,
                                  const float* grad, const float* step_b, float eps, float a, float* out,
                                  int B, int64_t n_per_img, void* stream) {
  SEA_CHECK_ARG(x && x_adv && x_old && grad && step_b && out && B > 0 && n_per_img > 0 && B <= 65535);
  hipStream_t s = (hipStream_t)stream;
  // (1 - a) is evaluated in double by the Python reference and then rounded to float32
  const float oma = (float)(1.0 - (double)a);
  if ((n_per_img % 4) == 0 && aligned16(x) && aligned16(x_adv) && aligned16(x_old) && aligned16(grad) &&
      aligned16(out)) {
    int64_t n4 = n_per_img / 4;
    int gx = grid_for(n4, 256);
    int cap = kMaxGridX / B;
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(apgd_linf_step_v4, dim3(gx, B), dim3(256), 0, s, (const f4*)x, (const f4*)x_adv,
                       (const f4*)x_old, (const f4*)grad, step_b, eps, a, oma, (f4*)out, n4);
  } else {
    int gx = grid_for(n_per_img, 256);
    int cap = kMaxGridX / B;
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(apgd_linf_step_v1, dim3(gx, B), dim3(256), 0, s, x, x_adv, x_old, grad, step_b, eps,
                       a, oma, out, n_per_img);
  }
  SEA_RETURN_LAST();
}

static int linf_step_graph_impl(const float* x, float* x_adv, float* x_old, const float* grad, const float* step_b, float eps,
                                const float* eps_dev, const int32_t* iter_dev, int B, int64_t n_per_img, void* stream) {
  SEA_CHECK_ARG(x && x_adv && x_old && grad && step_b && iter_dev && B > 0 && n_per_img > 0 && B <= 65535);
  int cap = kMaxGridX / B;
  if (cap < 1) cap = 1;
  if ((n_per_img % 4) != 0 || !(aligned16(x) && aligned16(x_adv) && aligned16(x_old) && aligned16(grad))) {
    int gx1 = grid_for(n_per_img, 256);
    if (gx1 > cap) gx1 = cap;
    hipLaunchKernelGGL(apgd_linf_step_inplace_v1, dim3(gx1, B), dim3(256), 0, (hipStream_t)stream, x, x_adv, x_old, grad, step_b,
                       eps, iter_dev, n_per_img, eps_dev);
    SEA_RETURN_LAST();
  }
  const int64_t n4 = n_per_img / 4;
  int gx = grid_for(n4, 256);
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(apgd_linf_step_inplace_v4, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, (const f4*)x, (f4*)x_adv,
                     (f4*)x_old, (const f4*)grad, step_b, eps, iter_dev, n4, eps_dev);
  SEA_RETURN_LAST();
}

extern "C" int sea_apgd_linf_step_graph(const float* x, float* x_adv, float* x_old, const float* grad,
                                        const float* step_b, float eps, const int32_t* iter_dev, int B,
                                        int64_t n_per_img, void* stream) {
  return linf_step_graph_impl(x, x_adv, x_old, grad, step_b, eps, nullptr, iter_dev, B, n_per_img, stream);
}

// the same with the radius read from device memory (one float): a captured graph then serves runs of any radius
extern "C" int sea_apgd_linf_step_graph_dev(const float* x, float* x_adv, float* x_old, const float* grad,
                                            const float* step_b, const float* eps_dev, const int32_t* iter_dev, int B,
                                            int64_t n_per_img, void* stream) {
  SEA_CHECK_ARG(eps_dev != nullptr);
  return linf_step_graph_impl(x, x_adv, x_old, grad, step_b, 0.f, eps_dev, iter_dev, B, n_per_img, stream);
}

template <int OP>
static int launch_ew2(const float* p, const float* q, float eps, float* out, int64_t n, void* stream) {
  SEA_CHECK_ARG(p && q && out && n > 0);
  hipStream_t s = (hipStream_t)stream;
  if ((n % 4) == 0 && aligned16(p) && aligned16(q) && aligned16(out)) {
    hipLaunchKernelGGL(ew2_v4<OP>, dim3(grid_for(n / 4, 256)), dim3(256), 0, s, (const f4*)p, (const f4*)q, eps, (f4*)out,
                       n / 4);
  } else {
    hipLaunchKernelGGL(ew2_v1<OP>, dim3(grid_for(n, 256)), dim3(256), 0, s, p, q, eps, out, n);
  }
  SEA_RETURN_LAST();
}

extern "C" int sea_linf_random_start(const float* x, const float* u, float eps, float* out, int64_t n,
                                     void* stream) {
  return launch_ew2<OP_RANDOM_START>(x, u, eps, out, n, stream);
}

extern "C" int sea_linf_project(const float* z, const float* x, float eps, float* out, int64_t n,
                                void* stream) {
  return launch_ew2<OP_PROJECT>(z, x, eps, out, n, stream);
}

extern "C" int sea_pgd_linf_step(const float* X, const float* delta, const float* grad, float alpha,
                                 float eps, float* delta_out, float* x_in_out, int clamp_input, int64_t n,
                                 void* stream) {
  SEA_CHECK_ARG(X && delta && grad && delta_out && n > 0);
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = (n % 4) == 0 && aligned16(X) && aligned16(delta) && aligned16(grad) &&
                  aligned16(delta_out) && (!x_in_out || aligned16(x_in_out));
  if (v4) {
    int g = grid_for(n / 4, 256);
    if (x_in_out)
      hipLaunchKernelGGL(pgd_linf_step_v4<true>, dim3(g), dim3(256), 0, s, (const float4*)X,
                         (const float4*)delta, (const float4*)grad, alpha, eps, clamp_input,
                         (float4*)delta_out, (float4*)x_in_out, n / 4);
    else
      hipLaunchKernelGGL(pgd_linf_step_v4<false>, dim3(g), dim3(256), 0, s, (const float4*)X,
                         (const float4*)delta, (const float4*)grad, alpha, eps, clamp_input,
                         (float4*)delta_out, (float4*)nullptr, n / 4);
  } else {
    int g = grid_for(n, 256);
    if (x_in_out)
      hipLaunchKernelGGL(pgd_linf_step_v1<true>, dim3(g), dim3(256), 0, s, X, delta, grad, alpha, eps,
                         clamp_input, delta_out, x_in_out, n);
    else
      hipLaunchKernelGGL(pgd_linf_step_v1<false>, dim3(g), dim3(256), 0, s, X, delta, grad, alpha, eps,
                         clamp_input, delta_out, (float*)nullptr, n);
  }
  SEA_RETURN_LAST();
}

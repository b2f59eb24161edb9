// K1' (round 6): the L2 branch of apgd_train -- reference semseg/attacker.py:412-436 (the step), with the per-image norm of
// autoattack.other_utils.L2_norm (attacker.py:6): ``(t ** 2).view(B, -1).sum(-1).sqrt()``:
//
//   z  = x_adv + step * grad / (||grad|| + 1e-12)
//   z  = clamp(x + (z - x) / (||z - x|| + 1e-12) * min(eps, ||z - x||), 0, 1)
//   z  = x_adv + (z - x_adv) * a + (x_adv - x_old) * (1 - a)
//   out = clamp(x + (z - x) / (||z - x|| + 1e-12) * min(eps, ||z - x||), 0, 1)
//
// Three per-image norms, each of a tensor that depends on the norm before: four passes, no intermediate tensor -- pass k
// RECOMPUTES the element-wise chain from x, x_adv, x_old, grad and the norms it already has, and either reduces the next
// tensor's squares (passes 1-3) or writes the result (pass 4).  A block owns a fixed slice of one image and writes ONE partial
// sum; every later pass adds an image's partials in index order (double): no atomics, run-to-run bitwise reproducible.
// Compiled with -ffp-contract=off: every element-wise multiply / add / divide rounds like the separate float32 ATen ops of the
// reference, so given the same norms the iterate is the reference's bit for bit; the norms themselves are sums in another order
// than ATen's (last-bit differences: the iterate agrees with the reference to ~1e-7, tests/test_l2_gpu.py).
// No shipped entry point of the reference reaches this branch: it exists so that the drop-in surface raises
// nothing the reference does not raise.  HBM-bound streaming work, 4 x (4 reads) + 1 write of the image tensors.
#include "sea_common.h"

namespace sea {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int L2_BLOCKS = 64;    // partial sums per image and pass

struct L2Args {
  const float* x;
  const float* x_adv;
  const float* x_old;
  const float* grad;
  const float* step_b;
  float eps, a, oma;
  float* out;
  double* part;                  // [3][B][L2_BLOCKS]
  int B;
  int64_t n;                     // elements per image
};

__device__ __forceinline__ float norm_of(const double* part) {   // fixed order; sqrt of the float32 sum like the reference's
  double s = 0.0;
  for (int i = 0; i < L2_BLOCKS; ++i) s += part[i];
  return sqrtf((float)s);
}

// the projection's factor:  1 / (n + 1e-12) * min(eps, n)  applied as  d / (n + 1e-12) * min(eps, n)
__device__ __forceinline__ float project1(float xv, float z, float n, float eps) {
  const float d = z - xv;
  float v = xv + d / (n + 1e-12f) * fminf(eps, n);
  v = fminf(fmaxf(v, 0.f), 1.f);
  return v;
}

template <int PASS>
__global__ __launch_bounds__(256) void apgd_l2_pass_kernel(const L2Args p) {
  const int b = blockIdx.y;
  const int64_t n = p.n, base = (int64_t)b * n;
  const float* x = p.x + base;
  const float* xa = p.x_adv + base;
  const float* xo = p.x_old + base;
  const float* g = p.grad + base;
  const float step = p.step_b[b];
  float ng = 0.f, n1 = 0.f, n2 = 0.f;
  if (PASS >= 2) ng = norm_of(p.part + ((int64_t)0 * p.B + b) * L2_BLOCKS);
  if (PASS >= 3) n1 = norm_of(p.part + ((int64_t)1 * p.B + b) * L2_BLOCKS);
  if (PASS >= 4) n2 = norm_of(p.part + ((int64_t)2 * p.B + b) * L2_BLOCKS);
  // a block's slice: consecutive elements (the same slice in every pass, so a partial is a pure function of the inputs)
  const int64_t per = (n + L2_BLOCKS - 1) / L2_BLOCKS;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float gv = g[i];
    float sq = 0.f;
    if (PASS == 1) {
      sq = gv * gv;
    } else {
      const float xv = x[i], av = xa[i];
      const float z0 = av + step * gv / (ng + 1e-12f);
      if (PASS == 2) {
        const float d = z0 - xv;
        sq = d * d;
      } else {
        const float z1 = project1(xv, z0, n1, p.eps);
        const float z2 = av + (z1 - av) * p.a + (av - xo[i]) * p.oma;
        if (PASS == 3) {
          const float d = z2 - xv;
          sq = d * d;
        } else {
          p.out[base + i] = project1(xv, z2, n2, p.eps);
        }
      }
    }
    if (PASS < 4) acc += (double)sq;
  }
  if (PASS < 4) {
    // lane -> wave -> block, fixed order
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.part[((int64_t)(PASS - 1) * p.B + b) * L2_BLOCKS + blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
  }
}

}  // namespace sea

using namespace sea;

extern "C" int64_t sea_apgd_l2_workspace_bytes(int B) { return B > 0 ? (int64_t)3 * B * L2_BLOCKS * 8 : -1; }

extern "C" int sea_apgd_l2_step(const float* x, const float* x_adv, const float* x_old, const float* grad, const float* step_b,
                                float eps, float a, float* out, void* workspace, int B, int64_t n_per_img, void* stream) {
  SEA_CHECK_ARG(x && x_adv && x_old && grad && step_b && out && workspace && B > 0 && B <= 65535 && n_per_img > 0);
  SEA_CHECK_ARG((((uintptr_t)workspace) & 7) == 0);
  L2Args p;
  p.x = x; p.x_adv = x_adv; p.x_old = x_old; p.grad = grad; p.step_b = step_b; p.eps = eps; p.a = a;
  p.oma = (float)(1.0 - (double)a);     // (1 - a) is evaluated in double by the Python reference and then rounded to float32
  p.out = out; p.part = (double*)workspace; p.B = B; p.n = n_per_img;
  const dim3 grid(L2_BLOCKS, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(apgd_l2_pass_kernel<1>, grid, block, 0, s, p);
  hipLaunchKernelGGL(apgd_l2_pass_kernel<2>, grid, block, 0, s, p);
  hipLaunchKernelGGL(apgd_l2_pass_kernel<3>, grid, block, 0, s, p);
  hipLaunchKernelGGL(apgd_l2_pass_kernel<4>, grid, block, 0, s, p);
  SEA_RETURN_LAST();
}

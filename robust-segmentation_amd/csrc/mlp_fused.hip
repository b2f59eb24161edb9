// M8f (round 6): the MLP of a ConvNeXt block -- y = res + W2 GELU(W1 x + b1) + b2, reference
// semseg/models/backbones/convnext_orig.py:77-79 (pwconv1 -> act -> pwconv2, layer scale folded into W2) -- as ONE kernel per
// direction in which the 4C-wide hidden tensor never leaves the compute unit.
//
// Why.  At the 128^2 and 64^2 stages (C = 96 / 192, M = 131072 / 32768 rows) the two-GEMM form is bound by the hidden tensor's
// round trips through HBM, not by the matrix cores: t = W1 x is 201 MB at C = 96, written by the first product, read by the
// second, and again -- with the equally large g W2 -- by the two input-gradient products: 1.2 GB per block against 0.25 GB of
// x / y / g / dx (profiles/r5_grid_by_grid.txt: 65 + 78 us forward, 65 + 116 us backward per block at C = 96).  Here a wave owns
// 32 rows of x for the whole MLP and walks the hidden dimension in tiles of 32:
//
//   forward    T^T (32 hidden x 32 rows) = W1[tile] x^T         x as fp16 x 2 B-fragments in REGISTERS for the whole kernel
//              h = GELU(T / scales + b1), split into fp16 x 2, turned into the next product's operand IN REGISTERS:
//              the accumulator holds hidden (e & 3) + 8 (e >> 2) + 4 half on lane = row, the operand wants 8 consecutive
//              hidden values per lane: one v_permlane32_swap per packed pair exchanges the middle quads of the two halves
//              y^T (C x 32 rows) += W2[:, tile] h^T             accumulators in registers for the whole kernel
//   backward   T recomputed from x (nothing but x is kept from the forward: the 4C-wide t is never written at all),
//              U^T = W2^T[tile] g^T, v = U GELU'(T), dx^T += W1^T[:, tile] v^T; the per-row maxima of g (the fp16 x 2 scales of
//              a gradient operand) are taken from the registers that hold g: no rowmax launch.
//
// Only the weights go through LDS (a 32-wide hidden tile of each packed weight: C / 4 KB forward, 3 C / 8 KB backward, double
// buffered, one barrier per hidden tile); they are the packed fp16 x 2 images sea_gemm_split_pack(terms = 22) writes, unchanged.
//
// SAME BITS as the two sea_gemm_split launches it replaces: the same operand split (split4_f16), the same power-of-two scales,
// the same three v_mfma_f32_32x32x16_f16 products per 16-deep step in the same order along K for every accumulator element.
// The products are issued with the operands exchanged (weights as A, activations as B): that transposes the result and nothing
// else (devtools/probes/mfma_swap_probe.hip: D(A, B)^T == D(B, A) bit for bit on gfx950), which is what puts a row's whole
// hidden vector on one lane pair.  tests/test_mlp_fused_gpu.py compares against the unfused pair with torch.equal.
#include "gemm_split.h"
#include <type_traits>
#include <utility>

namespace sea {

struct MlpArgs {
  const float* x;        // (M, C) LayerNorm output, row stride ldx
  const float* g;        // backward: (M, C) gradient w.r.t. the MLP output, row stride ldg
  int64_t ldx, ldg;
  const char* Wa;        // pack(w1: N = H, K = C), terms 22           [C/32][2][H][32] fp16 + inverse scales
  const char* Wu;        // backward: pack(w2^T: N = H, K = C)         same shape
  const char* Wb;        // forward: pack(w2: N = C, K = H); backward: pack(w1^T: N = C, K = H)    [H/32][2][Npad][32]
  const float* b1;       // (H)
  const float* b2;       // forward: (C) or null
  const float* res;      // forward: (M, C) residual or null, row stride ldres
  int64_t ldres;
  float* y;              // forward: y; backward: dx.  (M, C), row stride ldy
  int64_t ldy;
  int M, H, Npad;
  const uint32_t* amax1; // float bits of a bound of max|x| (one word: the LayerNorm bound)
  const uint32_t* amax2; // forward: float bits of a bound of max|GELU(t)| (one word)
  unsigned long long* dbg; // diagnostic builds only (STAMP): per wave 8 words of accumulated s_memtime ticks
  const float* ln_w;     // optional: x is the INPUT of a LayerNorm over its C channels (affine ln_w, ln_b, eps ln_eps) whose output
  const float* ln_b;     // feeds the MLP: normalised in the prologue (forward and backward), differentiated in the backward's epilogue
  float ln_eps;
  const float* amax_mul; // backward: ONE float, rowmax(g[r]) * this bounds |(g W2)[r][:] GELU'| (SeaGemmEpilogue.a_amax_mul_dev)
};

// erff without its branch.  The device library's erff is `if (|x| < 1) polynomial in x^2 else 1 - exp(-polynomial in |x|)`: per
// element a divergent branch that both sides of a wave take anyway, sixteen scalar branch regions per hidden tile that nothing
// can be scheduled across (the MFMAs of the next tile's first product in particular).  Both sides are evaluated here and
// selected: the same operations on the same constants (ROCm 7.2 ocml erfF: llvm.fmuladd chains that the backend contracts to
// fma, fma(ax, p, ax), llvm.exp.f32, copysign), hence the same bits -- for EVERY float: sea_probe_gelu_mismatches compares
// gelu / gelu' built on it with gelu_f / gelu_grad_f over all 2^32 inputs (tests/test_mlp_fused_gpu.py).
// (in stages, so that the kernel can place matrix instructions between them; gelu_nb / gelu_grad_nb below are these stages)
__device__ __forceinline__ float erf_small(float x) {          // |x| < 1: fma(|x|, P(x^2), |x|)
  const float ax = __builtin_fabsf(x);
  const float t = x * x;
  float p = __builtin_fmaf(t, -0x1.268bc2p-11f, 0x1.420828p-8f);
  p = __builtin_fmaf(t, p, -0x1.b5937p-6f);
  p = __builtin_fmaf(t, p, 0x1.ce077cp-4f);
  p = __builtin_fmaf(t, p, -0x1.81266p-2f);
  p = __builtin_fmaf(t, p, 0x1.06eba0p-3f);
  return __builtin_fmaf(ax, p, ax);
}
__device__ __forceinline__ float erf_large_arg(float x) {      // |x| >= 1: erf = 1 - exp(-this)
  const float ax = __builtin_fabsf(x);
  float q = __builtin_fmaf(ax, 0x1.1d3156p-16f, -0x1.8d129p-12f);
  q = __builtin_fmaf(ax, q, 0x1.f9a6d2p-9f);
  q = __builtin_fmaf(ax, q, -0x1.8c3164p-6f);
  q = __builtin_fmaf(ax, q, 0x1.b4e9c8p-4f);
  q = __builtin_fmaf(ax, q, 0x1.4515fap-1f);
  q = __builtin_fmaf(ax, q, 0x1.078e50p-3f);
  return __builtin_fmaf(ax, q, ax);
}
__device__ __forceinline__ float erf_exp(float q) { return __builtin_expf(-q); }
__device__ __forceinline__ float erf_finish(float x, float small, float ex) {
  return __builtin_copysignf(__builtin_fabsf(x) < 1.0f ? small : 1.0f - ex, x);
}
constexpr float kRsqrt2 = 0.70710678118654752440f;
__device__ __forceinline__ float gelu_finish(float x, float er) { return 0.5f * x * (1.f + er); }
__device__ __forceinline__ float gelu_grad_finish(float x, float er) {
  const float cdf = 0.5f * (1.f + er);
  const float pdf = __expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}
__device__ __forceinline__ float erf_nb(float x) { return erf_finish(x, erf_small(x), erf_exp(erf_large_arg(x))); }
__device__ __forceinline__ float gelu_nb(float x) { return gelu_finish(x, erf_nb(x * kRsqrt2)); }
__device__ __forceinline__ float gelu_grad_nb(float x) { return gelu_grad_finish(x, erf_nb(x * kRsqrt2)); }

// every float: bit patterns of gelu_nb / gelu_grad_nb against gelu_f / gelu_grad_f (NaN results count as equal to NaN)
__global__ __launch_bounds__(256) void gelu_compare_kernel(unsigned long long* out) {
  unsigned long long bad0 = 0, bad1 = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * 256) {
    const float x = __uint_as_float((uint32_t)i);
    const float a = gelu_f(x), b = gelu_nb(x), c = gelu_grad_f(x), d = gelu_grad_nb(x);
    bad0 += (__float_as_uint(a) != __float_as_uint(b)) && !(a != a && b != b);
    bad1 += (__float_as_uint(c) != __float_as_uint(d)) && !(c != c && d != d);
  }
  if (bad0) atomicAdd(out, bad0);
  if (bad1) atomicAdd(out + 1, bad1);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N - 1>): the step index is a constant inside f by construction
// (a `#pragma unroll` loop of 41 large steps was only partially unrolled at C = 192, and the remainder loop indexed the
// accumulators and fragments at run time, through scratch)
template <class Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}


// probe: the LayerNorm of the fused kernels' prologue on its own (tests: against sea_layernorm_fwd bit for bit)
template <int C>
__global__ __launch_bounds__(256) void ln_rows_probe_kernel(const float* x, const float* ln_w, const float* ln_b, float eps, int M,
                                                            float* yn, float* mean, float* rstd) {
  constexpr int KS = C / 16;
  constexpr int LN_KSP = C <= 128 ? 8 : 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  int row = (blockIdx.x * 4 + wave) * 32 + r;
  const bool live = row < M;
  row = live ? row : M - 1;
  f32x4 xa[KS][2];
  const float* const xr = x + (int64_t)row * C + 8 * h;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    xa[ks][0] = *(const f32x4*)(xr + 16 * ks);
    xa[ks][1] = *(const f32x4*)(xr + 16 * ks + 4);
  }
  const float ln_inv_c = 1.f / (float)C;
  auto ln_tree = [&](float (&t)[2][LN_KSP]) __attribute__((always_inline)) -> float {
#pragma unroll
    for (int o = LN_KSP / 2; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < o; ++i) {
        t[0][i] += t[0][i + o];
        t[1][i] += t[1][i + o];
      }
    const float a0 = t[0][0] + __shfl_xor(t[0][0], 32, 64), a1 = t[1][0] + __shfl_xor(t[1][0], 32, 64);
    return a0 + a1;
  };
  float t[2][LN_KSP];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < LN_KSP; ++i) t[j][i] = i < KS ? (xa[i < KS ? i : 0][j][0] + xa[i < KS ? i : 0][j][1]) + (xa[i < KS ? i : 0][j][2] + xa[i < KS ? i : 0][j][3]) : 0.f;
  float mu;
  {
#pragma clang fp contract(off)
    mu = ln_tree(t) * ln_inv_c;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < LN_KSP; ++i) {
        if (i < KS) {
          const float dx = xa[i][j][0] - mu, dy = xa[i][j][1] - mu, dz = xa[i][j][2] - mu, dw = xa[i][j][3] - mu;
          t[j][i] = fmaf(dx, dx, dy * dy) + fmaf(dz, dz, dw * dw);
        } else {
          t[j][i] = 0.f;
        }
      }
  }
  const float rs = rsqrtf(fmaf(ln_tree(t), ln_inv_c, eps));
  if (live) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 wv = *(const f32x4*)(ln_w + 16 * ks + 8 * h + 4 * j), bv = *(const f32x4*)(ln_b + 16 * ks + 8 * h + 4 * j);
        f32x4 o;
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaf((xa[ks][j][e] - mu) * rs, wv[e], bv[e]);
        }
        *(f32x4*)(yn + (int64_t)row * C + 16 * ks + 8 * h + 4 * j) = o;
      }
    if (h == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

__device__ __forceinline__ f32x16 mfma3(const u32x4 w_hi, const u32x4 w_mid, const u32x4 a_hi, const u32x4 a_mid, f32x16 c) {
  // sea_gemm_split's chain, smallest first: (activation mid, weight hi), (hi, mid), (hi, hi) -- weights as the A operand
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w_hi), __builtin_bit_cast(f16x8, a_mid), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w_mid), __builtin_bit_cast(f16x8, a_hi), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w_hi), __builtin_bit_cast(f16x8, a_hi), c, 0, 0, 0);
  return c;
}

// 8 consecutive floats of a row -> the fp16 x 2 fragments (hi, mid) of one 16-deep step, scaled by the row's power of two
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi4, float sc, u32x4& f_hi, u32x4& f_mid) {
  u32x2 a[2], b[2];
  split4_f16(lo, sc, a);
  split4_f16(hi4, sc, b);
  f_hi = u32x4{a[0][0], a[0][1], b[0][0], b[0][1]};
  f_mid = u32x4{a[1][0], a[1][1], b[1][0], b[1][1]};
}

// LDS of a block: two weight stages and the epilogue constants; the epilogue's 32-row patches reuse everything from offset 0
constexpr int mlp_stage_bytes(int C, bool bwd) { return (bwd ? 3 : 2) * 128 * C; }
constexpr int mlp_cst_bytes(int C, bool bwd) { return (bwd ? 3 : 2) * 4 * C * 4; }
constexpr int mlp_blocks_per_cu(int C, int waves) { return C <= 96 ? 8 / waves : 1; }   // (C = 192: 512 registers, one wave per SIMD)
constexpr int mlp_lds_bytes(int C, int waves, bool bwd) {
  const int a = 2 * mlp_stage_bytes(C, bwd) + mlp_cst_bytes(C, bwd);
  const int b = waves * 32 * (C + 4) * 4;
  return a > b ? a : b;
}

template <int C, int WAVES, bool BWD, bool PIPE, bool STAMP = false, int MLP_EPS = 4>
__global__ __launch_bounds__(WAVES * 64, (C <= 96 ? 8 : 4) / 4) void mlp_fused_kernel(const MlpArgs p) {
  constexpr int T = WAVES * 64;
  constexpr int KS = C / 16;             // 16-deep steps of the products that contract over C
  constexpr int NT = C / 32;             // 32-row output tiles of the products that contract over the hidden dimension
  constexpr int ARR = 128 * C;           // bytes of one staged array: (C/32) x 2 images of 32 rows x 64 B == 2 images of C rows x 64 B
  constexpr int NARR = BWD ? 3 : 2;      // [W1 tile j + 1] ([W2^T tile j + 1]) [second-product K block j]
  constexpr int STAGE = NARR * ARR;
  constexpr int PIECES = STAGE / 16;
  constexpr int PER = (PIECES + T - 1) / T;
  static_assert(PIECES % 64 == 0, "whole waves of 16-byte pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const uint64_t t_entry = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int r = lane & 31, h = lane >> 5;
  const int M = p.M, H = p.H;
  const int NTILE = H / 32;
  const int m0 = ((int)blockIdx.x * WAVES + wave) * 32;
  int row = m0 + r;
  const bool live = row < M;
  row = live ? row : M - 1;

  // ---- weight staging by LDS-DMA (global_load_lds_dwordx4: no staging registers; a wave-instruction fills 1 KB of LDS in lane
  // order, so the 16-byte chunk swizzle of the images is applied to the SOURCE address).  Piece pc = 64 wave + T i + lane of a
  // stage lands at byte 16 pc = (array, image, row, chunk position) and takes chunk position ^ ((row >> 2) & 3) of that row of
  // the packed weights.  Arrays and images are whole multiples of 64 pieces, so everything but ONE lane offset is scalar:
  // 16 rows x 4 chunks per instruction, row = 16 k + (lane >> 2) with (row >> 2) & 3 == (lane >> 4) & 3.
  const uint32_t lane_off = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
  // stage s(j) = {row tiles ja of the first products' weights, K block jb of the second product's}
  auto dma = [&](char* st, int ja, int jb) __attribute__((always_inline)) {
    ja = ja < NTILE ? ja : NTILE - 1;    // (past the end: a valid tile again, never read)
    jb = jb < NTILE ? jb : NTILE - 1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int pcw = wave_u * 64 + T * i;                       // first piece of this wave's instruction
      if (PIECES % T != 0 && pcw >= PIECES) continue;            // (wave-uniform)
      const int arr = pcw / (8 * C), qw = pcw - arr * (8 * C);
      const char* src;
      if (arr < NARR - 1)                // a 32-row tile of a [C/32][2][H][32] pack: image qw >> 7, 2 KB each
        src = (arr == 0 ? p.Wa : p.Wu) + ((int64_t)(qw >> 7) * H + 32 * ja) * 64 + (qw & 127) * 16;
      else {                             // K block jb of a [H/32][2][Npad][32] pack, rows 0 .. C - 1: image qw / (4 C)
        const int t = qw / (4 * C);
        src = p.Wb + ((int64_t)(2 * jb + t) * p.Npad) * 64 + (qw - t * (4 * C)) * 16;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane_off),
                                       (__attribute__((address_space(3))) void*)(st + pcw * 16), 16, 0, 0);
    }
  };
  char* const st0 = smem;
  char* const st1 = smem + STAGE;
  // per-hidden-unit constants of the first products' epilogue (inverse weight scales, bias): into LDS once -- a global load
  // inside the loop would be waited for with vmcnt, which is in order: it would also wait for the tile's DMA
  float* const cst = (float*)(smem + 2 * STAGE);      // [wi_a (H)] [b1 (H)] ([wi_u (H)])
  {
    const float* const wa_g = (const float*)(p.Wa + (int64_t)(C / 32) * 2 * H * 64);   // inverse weight scales behind the images
    const float* const wu_g = BWD ? (const float*)(p.Wu + (int64_t)(C / 32) * 2 * H * 64) : nullptr;
    for (int i = tid; i < H; i += T) {
      cst[i] = wa_g[i];
      cst[H + i] = p.b1[i];
      if constexpr (BWD) cst[2 * H + i] = wu_g[i];
    }
  }
  // ---- prologue: tile 0 of the first products' weights goes into stage 1 (read once, before the loop), s(0) into stage 0
  dma(st1, 0, 0);
  dma(st0, 1, 0);

  // ---- this lane's half of its row, as fp16 x 2 B-fragments (one power-of-two scale per row).  Read straight from global: 16
  // bytes per lane at a row stride of 4 C bytes.  (A coalesced stream through a wave-private LDS patch was built and measured:
  // no faster -- the prologue is bound by HBM, every block of a round loading its rows at the same time, not by the access
  // pattern: 11.7 k vs 12.5 k cycles forward, 24.6 k vs 22.7 k backward at C = 96 -- and its LDS cost the second block per CU.)
  constexpr int LDP = (C + 4) * 4;       // row stride of the epilogue's LDS patch in bytes
  constexpr int NF4 = 32 * C / 4 / 64;   // 16-byte accesses per lane for the wave's 32 rows
  auto load_rows = [&](const float* base, int64_t ld, f32x4 (&out)[KS][2]) __attribute__((always_inline)) {
    const float* const xr = base + (int64_t)row * ld + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      out[ks][0] = *(const f32x4*)(xr + 16 * ks);
      out[ks][1] = *(const f32x4*)(xr + 16 * ks + 4);
    }
  };
  float sc1, inv1;
  pow2_scale(*p.amax1, sc1, inv1);
  u32x4 xf[KS][2];
  // LayerNorm in front of the MLP (ConvNeXt: norm -> pwconv1, convnext_orig.py:75-77), optional.  The sums over the channels
  // are taken in the order of ln_fwd_kernel / ln_bwd_kernel (csrc/ln_kernels.hip: one float4 per lane of a 32- or 64-lane
  // group, (a + b) + (c + d) per float4, then an xor-shuffle tree from the highest index bit down), so that fusing the
  // LayerNorm in changes no bit: here float4 v = 4 ks + 2 h + j sits in lane half h, and the tree's levels are the bits of
  // ks (in the lane), then h (one exchange between the two halves), then j (in the lane).
  constexpr int LN_KSP = C <= 128 ? 8 : 16;      // float4 slots per (h, j) of the kernel's lane group (32 lanes: 8, 64 lanes: 16)
  const float ln_inv_c = 1.f / (float)C;
  auto ln_tree = [&](float (&t)[2][LN_KSP]) __attribute__((always_inline)) -> float {
#pragma unroll
    for (int o = LN_KSP / 2; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < o; ++i) {
        t[0][i] += t[0][i + o];
        t[1][i] += t[1][i + o];
      }
    const float a0 = t[0][0] + __shfl_xor(t[0][0], 32, 64), a1 = t[1][0] + __shfl_xor(t[1][0], 32, 64);
    return a0 + a1;
  };
  float ln_mu = 0.f, ln_rs = 1.f;
  {
    f32x4 xa[KS][2];
    load_rows(p.x, p.ldx, xa);
    if (p.ln_w != nullptr) {
      float t[2][LN_KSP];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < LN_KSP; ++i) t[j][i] = i < KS ? (xa[i < KS ? i : 0][j][0] + xa[i < KS ? i : 0][j][1]) + (xa[i < KS ? i : 0][j][2] + xa[i < KS ? i : 0][j][3]) : 0.f;
      // (contraction off, fused operations written out: the LayerNorm kernel's COMPILED arithmetic -- fma(dx, dx, dy dy) +
      // fma(dz, dz, dw dw) per float4, fma(sum, 1 / C, eps) -- not whatever fusion this context would invite: without the
      // pragma hipcc turns x - sum * (1 / C) into one v_fmamk here)
      {
#pragma clang fp contract(off)
        ln_mu = ln_tree(t) * ln_inv_c;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < LN_KSP; ++i) {
            if (i < KS) {
              const float dx = xa[i][j][0] - ln_mu, dy = xa[i][j][1] - ln_mu, dz = xa[i][j][2] - ln_mu, dw = xa[i][j][3] - ln_mu;
              t[j][i] = fmaf(dx, dx, dy * dy) + fmaf(dz, dz, dw * dw);
            } else {
              t[j][i] = 0.f;
            }
          }
      }
      ln_rs = rsqrtf(fmaf(ln_tree(t), ln_inv_c, p.ln_eps));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 wv = *(const f32x4*)(p.ln_w + 16 * ks + 8 * h + 4 * j), bv = *(const f32x4*)(p.ln_b + 16 * ks + 8 * h + 4 * j);
          {
#pragma clang fp contract(off)
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[ks][j][e] = fmaf((xa[ks][j][e] - ln_mu) * ln_rs, wv[e], bv[e]);
          }
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) split8(xa[ks][0], xa[ks][1], sc1, xf[ks][0], xf[ks][1]);
  }
  float sc2 = 1.f, inv2 = 1.f;           // scale of the second product's activation operand (forward: one word; backward: per row)
  float invg = 1.f;
  u32x4 gf[BWD ? KS : 1][2];
  if constexpr (BWD) {
    f32x4 ga[KS][2];
    load_rows(p.g, p.ldg, ga);
    uint32_t mx = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t b0 = __float_as_uint(ga[ks][0][e]) & 0x7fffffffu, b1 = __float_as_uint(ga[ks][1][e]) & 0x7fffffffu;
        mx = b0 > mx ? b0 : mx;
        mx = b1 > mx ? b1 : mx;
      }
    }
    const uint32_t other = (uint32_t)__shfl_xor((int)mx, 32, 64);   // the other half of the row
    mx = other > mx ? other : mx;        // = sea_absmax_bits(rows_per_word = 1): float bits of max_k |g[row][k]|
    float scg;
    pow2_scale(mx, scg, invg);
    const float mul = *p.amax_mul;
    uint32_t word = mx;
    if (mul != 1.f) word = __float_as_uint(__uint_as_float(word) * mul) & 0x7fffffffu;
    pow2_scale(word, sc2, inv2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) split8(ga[ks][0], ga[ks][1], scg, gf[ks][0], gf[ks][1]);
  } else {
    pow2_scale(*p.amax2, sc2, inv2);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const uint32_t rd = (uint32_t)(r * 64 + swz<false>(r, h));      // fragment read offset inside a 32-row image; step s: ^ 32
  // first products of hidden tile `from`: T^T (and U^T) = W[tile] . act^T, contraction over C
  auto first = [&](const char* st, f32x16& Tt, f32x16& Ut) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Tt[e] = 0.f;
    if constexpr (BWD) {
#pragma unroll
      for (int e = 0; e < 16; ++e) Ut[e] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint32_t o = (uint32_t)((ks >> 1) * 4096) + (rd ^ (uint32_t)(32 * (ks & 1)));
      const u32x4 w0 = *(const u32x4*)(st + o), w1 = *(const u32x4*)(st + o + 2048);
      Tt = mfma3(w0, w1, xf[ks][0], xf[ks][1], Tt);
      if constexpr (BWD) {
        const u32x4 u0 = *(const u32x4*)(st + ARR + o), u1 = *(const u32x4*)(st + ARR + o + 2048);
        Ut = mfma3(u0, u1, gf[ks][0], gf[ks][1], Ut);
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;

  f32x16 Tc, Uc, Tn, Un;
  first(st1, Tc, Uc);
  __syncthreads();                       // stage 1 is free for s(1)

  const float* const wi_a = cst;
  const float* const b1_l = cst + H;
  const float* const wi_u = cst + 2 * H;
  constexpr int SECOND = (NARR - 1) * ARR;

  // ---- the loop is a three-stage software pipeline over the hidden tiles.  Iteration j holds, independent of each other,
  //   F  the first products of tile j + 1                (matrix pipe; weights: row tiles j + 1 of stage s(j))
  //   E  the element-wise work of tile j                 (vector ALU: scales, bias, GELU / GELU', fp16 x 2 split, lane exchange)
  //   S  the second product of tile j - 1                (matrix pipe; weights: K block j - 1 of stage s(j))
  // so a wave's own MFMAs run in the shadow of its own VALU work (one per VPER vector instructions, placed with
  // sched_group_barrier: left alone the scheduler issues the 18 + 18 MFMAs in two clusters around ~700 vector instructions and
  // the two waves of a SIMD end up taking turns instead of overlapping -- 20 % MFMA-busy, 44 % of the wave cycles waiting,
  // profiles/r6_mlp_fused_pmc.txt).  PIPE = false (the backward at C = 96: 256 registers at two waves per SIMD) keeps F behind
  // E in the iteration: no second pair of first-product accumulators.
  // stage s(j) = {first-product row tiles j + 1, second-product K block j - 1}; s(j + 1) lands by DMA during iteration j.
  u32x4 hp[2][2] = {};                   // operand fragments of tile j - 1: [16-deep step][term]
  uint64_t tick[4] = {0, 0, 0, 0};
  const uint64_t t_begin = STAMP ? __builtin_amdgcn_s_memtime() : 0;
#define SEA_PIN() __builtin_amdgcn_sched_barrier(0)
  auto body = [&](auto FF, auto SS, auto EE, int j) __attribute__((always_inline)) {
    constexpr bool F = decltype(FF)::value, S = decltype(SS)::value, E = decltype(EE)::value;
    char* const cur = (j & 1) ? st1 : st0;
    char* const nxt = (j & 1) ? st0 : st1;
    uint64_t c0 = 0, c2 = 0, c3 = 0, c4 = 0;
    if constexpr (STAMP) {
      c0 = __builtin_amdgcn_s_memtime();
      SEA_PIN();
    }
    constexpr bool LAG = PIPE;           // the second product runs one tile behind, beside the next tile's element-wise work
    if (j + 1 < NTILE + (LAG ? 1 : 0)) dma(nxt, j + 2, LAG ? j : j + 1);
    // ---- the matrix work beside the element-wise work, as a list of TRIPLES (the three products of one 16-deep step into one
    // accumulator; two weight fragments each, read one triple ahead): first products of tile j + 1 (T, and U alternating with
    // it in the backward), then the second product of tile j - 1
    constexpr int NTF = (F && PIPE) ? (BWD ? 2 : 1) * KS : 0;
    constexpr int NTS = (S && LAG) ? 2 * NT : 0;
    constexpr int NTR = NTF + NTS, NM = 3 * NTR;
    u32x4 wf[2][2];
    auto rd_frags = [&](auto TR) __attribute__((always_inline)) {
      constexpr int tr = decltype(TR)::value;
      if constexpr (tr < NTF) {
        constexpr int ks = BWD ? tr >> 1 : tr;
        const uint32_t o = (uint32_t)(((BWD && (tr & 1)) ? ARR : 0) + (ks >> 1) * 4096) + (rd ^ (uint32_t)(32 * (ks & 1)));
        wf[tr & 1][0] = *(const u32x4*)(cur + o);
        wf[tr & 1][1] = *(const u32x4*)(cur + o + 2048);
      } else {
        constexpr int t2 = tr - NTF, ss = t2 / NT, n = t2 - ss * NT;
        const uint32_t o = (uint32_t)(SECOND + n * 2048) + (rd ^ (uint32_t)(32 * ss));
        wf[tr & 1][0] = *(const u32x4*)(cur + o);
        wf[tr & 1][1] = *(const u32x4*)(cur + o + C * 64);
      }
    };
    auto mfma_m = [&](auto MC) __attribute__((always_inline)) {
      constexpr int m = decltype(MC)::value, tr = m / 3, pr = m - 3 * tr;
      if constexpr (pr == 0 && tr + 1 < NTR) rd_frags(std::integral_constant<int, tr + 1>{});
      // sea_gemm_split's chain, smallest first: (activation mid, weight hi), (hi, mid), (hi, hi) -- weights as the A operand
      const f16x8 w = __builtin_bit_cast(f16x8, wf[tr & 1][pr == 1 ? 1 : 0]);
      constexpr int at = pr == 0 ? 1 : 0;
      if constexpr (tr < NTF) {
        constexpr int ks = BWD ? tr >> 1 : tr;
        if constexpr (BWD && (tr & 1)) {
          if constexpr (tr == 1 && pr == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) Un[e] = 0.f;
          }
          Un = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, __builtin_bit_cast(f16x8, gf[BWD ? ks : 0][at]), Un, 0, 0, 0);
        } else {
          if constexpr (tr == 0 && pr == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) Tn[e] = 0.f;
          }
          Tn = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, __builtin_bit_cast(f16x8, xf[ks][at]), Tn, 0, 0, 0);
        }
      } else {
        constexpr int t2 = tr - NTF, ss = t2 / NT, n = t2 - ss * NT;
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, __builtin_bit_cast(f16x8, hp[ss][at]), acc[n], 0, 0, 0);
      }
    };
    if constexpr (NTR > 0) rd_frags(std::integral_constant<int, 0>{});
    u32x4 hf[2][2];
    if constexpr (E) {
      // ---- hidden tile j, element-wise: scales, bias, GELU / GELU', the fp16 x 2 split and the lane exchange that makes it the
      // next product's operand, cut into NSTEP micro-steps of ~20 vector instructions (a quad's pre-activations; per PAIR of
      // elements: erf's small-argument polynomial, its large-argument polynomial, the exponential, the finish; a quad's split;
      // the exchange), each pinned behind its share of the MFMAs: a wave's matrix instructions run in the shadow of its own
      // vector work instead of in a phase of their own (left to itself the scheduler issues them in two clusters around ~700
      // vector instructions, and the two waves of a SIMD take turns: profiles/r6_mlp_fused_stamps.log)
      constexpr int EPS = MLP_EPS;         // elements per micro-step (2 or 4 independent dependency chains)
      constexpr int SPQ = 2 + 4 * (4 / EPS);   // micro-steps per quad
      constexpr int NSTEP = 4 * SPQ + 1;
      float ev[16], ek[16], ep[16], eq[16];
      uint32_t hq[4][2], mq[4][2];       // packed fp16 pairs of quad q = hidden 8 q + 4 h + {0, 1}, {2, 3}: first and second term
      f32x4 wa, bb, wu;
      auto cst_rd = [&](int q) __attribute__((always_inline)) {
        const int hid = 32 * j + 8 * q + 4 * h;
        wa = *(const f32x4*)(wi_a + hid);
        bb = *(const f32x4*)(b1_l + hid);
        if constexpr (BWD) wu = *(const f32x4*)(wi_u + hid);
      };
      cst_rd(0);
      static_for<NSTEP>([&](auto IC) __attribute__((always_inline)) {
        constexpr int i = decltype(IC)::value;
        constexpr int m0 = i * NM / NSTEP, m1 = (i + 1) * NM / NSTEP;
        static_for<m1 - m0>([&](auto U) __attribute__((always_inline)) { mfma_m(std::integral_constant<int, m0 + decltype(U)::value>{}); });
        constexpr int q = i / SPQ, k = i - SPQ * q;
        if constexpr (i == NSTEP - 1) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            // lanes 0-31 hold hidden 16 s + {0-3} (quad 2 s) and {8-11} (quad 2 s + 1), lanes 32-63 {4-7} and {12-15}; the
            // operand of step s wants 16 s + 8 half + 0 .. 7 on a lane: lower lanes keep quad 2 s and take the upper lanes'
            // quad 2 s, upper lanes take the lower lanes' quad 2 s + 1 and keep their own
            const auto a0 = __builtin_amdgcn_permlane32_swap(hq[2 * s2][0], hq[2 * s2 + 1][0], false, false);
            const auto a1 = __builtin_amdgcn_permlane32_swap(hq[2 * s2][1], hq[2 * s2 + 1][1], false, false);
            const auto b0 = __builtin_amdgcn_permlane32_swap(mq[2 * s2][0], mq[2 * s2 + 1][0], false, false);
            const auto b1 = __builtin_amdgcn_permlane32_swap(mq[2 * s2][1], mq[2 * s2 + 1][1], false, false);
            hf[s2][0] = u32x4{a0[0], a1[0], a0[1], a1[1]};
            hf[s2][1] = u32x4{b0[0], b1[0], b0[1], b1[1]};
          }
        } else if constexpr (k == 0) {   // the quad's pre-activations t and erf arguments
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = Tc[4 * q + e] * (inv1 * wa[e]) + bb[e];
            ev[4 * q + e] = t;
            ek[4 * q + e] = t * kRsqrt2;
          }
        } else if constexpr (k == SPQ - 1) {   // the quad's split (and the next quad's constants)
          u32x2 s2[2];
          split4_f16(f32x4{ev[4 * q], ev[4 * q + 1], ev[4 * q + 2], ev[4 * q + 3]}, sc2, s2);
          hq[q][0] = s2[0][0];
          hq[q][1] = s2[0][1];
          mq[q][0] = s2[1][0];
          mq[q][1] = s2[1][1];
          if constexpr (q + 1 < 4) cst_rd(q + 1);
        } else {
          constexpr int pr = (k - 1) >> 2, st = (k - 1) & 3;
#pragma unroll
          for (int d = 0; d < EPS; ++d) {
            const int e = 4 * q + EPS * pr + d;
            if constexpr (st == 0) ep[e] = erf_small(ek[e]);
            if constexpr (st == 1) eq[e] = erf_large_arg(ek[e]);
            if constexpr (st == 2) eq[e] = erf_exp(eq[e]);
            if constexpr (st == 3) {
              const float er = erf_finish(ek[e], ep[e], eq[e]);
              if constexpr (BWD) {
                float u = Uc[e] * (invg * wu[e & 3]) + 0.f;
                u *= gelu_grad_finish(ev[e], er);
                ev[e] = u;
              } else {
                ev[e] = gelu_finish(ev[e], er);
              }
            }
          }
        }
        SEA_PIN();
      });
    } else {
      static_for<NM>([&](auto MC) __attribute__((always_inline)) { mfma_m(MC); });
    }
    if constexpr (S && !LAG) {           // out^T (C x 32 rows) += W[:, tile j] . hidden^T
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const uint32_t o = (uint32_t)(SECOND + n * 2048) + (rd ^ (uint32_t)(32 * ss));
          const u32x4 w0 = *(const u32x4*)(cur + o), w1 = *(const u32x4*)(cur + o + C * 64);
          acc[n] = mfma3(w0, w1, hf[ss][0], hf[ss][1], acc[n]);
        }
    }
    if constexpr (F && !PIPE) first(cur, Tc, Uc);
    if constexpr (E && LAG) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        hp[s2][0] = hf[s2][0];
        hp[s2][1] = hf[s2][1];
      }
    }
    if constexpr (STAMP) {
      SEA_PIN();
      c2 = __builtin_amdgcn_s_memtime();
      SEA_PIN();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (STAMP) {
      c3 = __builtin_amdgcn_s_memtime();
      SEA_PIN();
    }
    __syncthreads();                     // `cur` is read, `nxt` is written
    if constexpr (STAMP) {
      c4 = __builtin_amdgcn_s_memtime();
      SEA_PIN();
      tick[0] += 0;
      tick[1] += c2 - c0;                // the interleaved matrix + element-wise work (+ first products when not pipelined)
      tick[2] += c3 - c2;                // wait for the DMA
      tick[3] += c4 - c3;                // barrier
    }
    if constexpr (F && PIPE) {
      Tc = Tn;
      if constexpr (BWD) Uc = Un;
    }
  };
#undef SEA_PIN
  using Y = std::true_type;
  using No = std::false_type;
  if constexpr (PIPE) {
    body(Y{}, No{}, Y{}, 0);
    for (int j = 1; j + 1 < NTILE; ++j) body(Y{}, Y{}, Y{}, j);
    body(No{}, Y{}, Y{}, NTILE - 1);
    body(No{}, Y{}, No{}, NTILE);
  } else {
    for (int j = 0; j + 1 < NTILE; ++j) body(Y{}, Y{}, Y{}, j);
    body(No{}, Y{}, Y{}, NTILE - 1);
  }
  const uint64_t t_loop_end = STAMP ? __builtin_amdgcn_s_memtime() : 0;

  // ---- epilogue: lane = row, register e of tile n -> column 32 n + (e & 3) + 8 (e >> 2) + 4 h.  Stored as they stand those are
  // 16-byte pieces at a row stride of 4 C bytes (32 lines per wave-instruction, and as many again for the residual): 19 k
  // cycles per block at C = 96.  The wave turns its 32 x C tile through its own patch of the (now idle) stages and streams it
  // out -- and the residual in -- 1 KB per instruction.
  const float* const wi_b = (const float*)(p.Wb + (int64_t)(H / 32) * 2 * p.Npad * 64);
  {
    char* const ep = smem + wave_u * (32 * LDP);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * n + 8 * q + 4 * h;
        const f32x4 wb = *(const f32x4*)(wi_b + col);
        f32x4 bb = {0.f, 0.f, 0.f, 0.f};
        if (!BWD && p.b2) bb = *(const f32x4*)(p.b2 + col);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[n][4 * q + e] * (inv2 * wb[e]) + bb[e];
        if constexpr (BWD) {
          if (p.ln_w != nullptr) {       // keep the row's gradient w.r.t. the LayerNorm OUTPUT in the accumulators: differentiated below
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[n][4 * q + e] = v[e];
            continue;
          }
        }
        *(f32x4*)(ep + r * LDP + col * 4) = v;
      }
    if constexpr (BWD) {
      if (p.ln_w != nullptr) {
        // ---- input gradient of the LayerNorm (ln_bwd_kernel's arithmetic and summation order): float4 v = 8 n + 2 q + h of the
        // row sits in accumulator quad (n, q) of lane half h; tree levels: the bits of n, then of q (in the lane), then h
        constexpr int NTP = C <= 128 ? 4 : 8;
        const float* const xr = p.x + (int64_t)row * p.ldx;
        float t1[NTP][4], t2[NTP][4];
        f32x4 gw[NT][4], xh[NT][4];
#pragma unroll
        for (int n = 0; n < NTP; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (n < NT) {
              const int col = 32 * n + 8 * q + 4 * h;
              const f32x4 wv = *(const f32x4*)(p.ln_w + col), xv = *(const f32x4*)(xr + col);
              const int nn = n < NT ? n : 0;
              {
#pragma clang fp contract(off)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  gw[nn][q][e] = acc[nn][4 * q + e] * wv[e];
                  xh[nn][q][e] = (xv[e] - ln_mu) * ln_rs;
                }
                t1[n][q] = (gw[nn][q][0] + gw[nn][q][1]) + (gw[nn][q][2] + gw[nn][q][3]);
                t2[n][q] = (gw[nn][q][0] * xh[nn][q][0] + gw[nn][q][1] * xh[nn][q][1]) + (gw[nn][q][2] * xh[nn][q][2] + gw[nn][q][3] * xh[nn][q][3]);
              }
            } else {
              t1[n][q] = 0.f;
              t2[n][q] = 0.f;
            }
          }
#pragma unroll
        for (int o = NTP / 2; o > 0; o >>= 1)
#pragma unroll
          for (int i = 0; i < o; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              t1[i][q] += t1[i + o][q];
              t2[i][q] += t2[i + o][q];
            }
        float s1 = (t1[0][0] + t1[0][2]) + (t1[0][1] + t1[0][3]), s2 = (t2[0][0] + t2[0][2]) + (t2[0][1] + t2[0][3]);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 o4;
            {
#pragma clang fp contract(off)
              const float m1 = s1 * ln_inv_c, m2 = s2 * ln_inv_c;
#pragma unroll
              for (int e = 0; e < 4; ++e) o4[e] = ln_rs * fmaf(-xh[n][q][e], m2, gw[n][q][e] - m1);
            }
            *(f32x4*)(ep + r * LDP + (32 * n + 8 * q + 4 * h) * 4) = o4;
          }
      }
    }
    const bool has_res = !BWD && p.res != nullptr;
#pragma unroll
    for (int it = 0; it < NF4; ++it) {
      const int f = lane + 64 * it, rw = f / (C / 4), c4 = f - rw * (C / 4);
      f32x4 v = *(const f32x4*)(ep + rw * LDP + c4 * 16);
      if (m0 + rw < M) {
        if (has_res) v += *(const f32x4*)(p.res + (int64_t)(m0 + rw) * p.ldres + 4 * c4);
        *(f32x4*)(p.y + (int64_t)(m0 + rw) * p.ldy + 4 * c4) = v;
      }
    }
  }
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && p.dbg) {
      unsigned long long* d = p.dbg + ((int64_t)blockIdx.x * WAVES + wave) * 8;
      for (int k = 1; k < 4; ++k) d[k] = tick[k];
      d[0] = __builtin_amdgcn_s_memtime() - t_loop_end;   // the epilogue (stores drained)
      d[4] = t_loop_end - t_begin;                        // the loop
      d[5] = t_begin;
      d[6] = __builtin_amdgcn_s_memrealtime();
      d[7] = t_begin - t_entry;                           // the prologue
    }
  }
}

template <int C, int WAVES, bool BWD, bool PIPE, bool STAMP = false, int EPS = 4>
static int mlp_launch(const MlpArgs& p, hipStream_t st) {
  constexpr int lds = mlp_lds_bytes(C, WAVES, BWD);
  static_assert(lds * mlp_blocks_per_cu(C, WAVES) <= 160 * 1024, "LDS of a compute unit");
  auto k = mlp_fused_kernel<C, WAVES, BWD, PIPE, STAMP, EPS>;
  if (lds > 48 * 1024) {
    static bool attr_set_dev[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set_dev[dev & 63]) {
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_set_dev[dev & 63] = true;
    }
  }
  const int rows_per_block = WAVES * 32;
  hipLaunchKernelGGL(k, dim3((p.M + rows_per_block - 1) / rows_per_block), dim3(WAVES * 64), (size_t)lds, st, p);
  return (int)hipGetLastError();
}

}  // namespace sea

using namespace sea;

// out[0], out[1] (pre-zeroed): number of floats for which the branch-free GELU / GELU' of this file differ from gelu_f / gelu_grad_f
extern "C" int sea_probe_gelu_mismatches(unsigned long long* out, void* stream) {
  SEA_CHECK_ARG(out != nullptr);
  hipLaunchKernelGGL(gelu_compare_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, out);
  SEA_RETURN_LAST();
}

static const int g_mlp_eps = [] {   // (A/B knob: elements per micro-step of the element-wise work, 2 or 4)
  const char* e = getenv("SEA_MLP_EPS");
  return (e && e[0] == '2') ? 2 : 4;
}();

extern "C" int sea_probe_ln_rows(const float* x, const float* ln_w, const float* ln_b, float eps, int M, int C, float* yn, float* mean,
                                 float* rstd, void* stream) {
  SEA_CHECK_ARG(x && ln_w && ln_b && yn && mean && rstd && M > 0 && (C == 96 || C == 192));
  const dim3 grid((M + 127) / 128), block(256);
  if (C == 96)
    hipLaunchKernelGGL(ln_rows_probe_kernel<96>, grid, block, 0, (hipStream_t)stream, x, ln_w, ln_b, eps, M, yn, mean, rstd);
  else
    hipLaunchKernelGGL(ln_rows_probe_kernel<192>, grid, block, 0, (hipStream_t)stream, x, ln_w, ln_b, eps, M, yn, mean, rstd);
  SEA_RETURN_LAST();
}

extern "C" int sea_mlp_fused_supported(int C, int H) { return (H == 4 * C && (C == 96 || C == 192)) ? 1 : 0; }

static bool mlp_common_ok(const MlpArgs& p, int C) {
  return p.x && p.Wa && p.Wb && p.b1 && p.y && p.amax1 && p.M > 0 && sea_mlp_fused_supported(C, p.H) && (p.ldx % 4) == 0 &&
         (p.ldy % 4) == 0 && p.ldx >= C && p.ldy >= C &&
         ((((uintptr_t)p.x) | ((uintptr_t)p.y) | ((uintptr_t)p.Wa) | ((uintptr_t)p.Wb) | ((uintptr_t)p.b1)) & 15) == 0;
}

static int mlp_fwd_impl(const float* x, int64_t ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                        const float* res, int64_t ldres, float* y, int64_t ldy, int M, int C, int H, const uint32_t* amax_x,
                        const uint32_t* amax_h, const float* ln_w, const float* ln_b, float ln_eps, void* stream) {
  MlpArgs p = {};
  p.ln_w = ln_w; p.ln_b = ln_b; p.ln_eps = ln_eps;
  SEA_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr) && ((((uintptr_t)ln_w) | ((uintptr_t)ln_b)) & 15) == 0);
  p.x = x; p.ldx = ldx; p.Wa = (const char*)W1p; p.Wb = (const char*)W2p; p.b1 = b1; p.b2 = b2; p.res = res; p.ldres = ldres;
  p.y = y; p.ldy = ldy; p.M = M; p.H = H; p.Npad = (C + GS_BN - 1) / GS_BN * GS_BN; p.amax1 = amax_x; p.amax2 = amax_h;
  SEA_CHECK_ARG(mlp_common_ok(p, C) && amax_h != nullptr);
  SEA_CHECK_ARG((((uintptr_t)b2) & 15) == 0 && (!res || ((ldres % 4) == 0 && ldres >= C && (((uintptr_t)res) & 15) == 0)));
  const hipStream_t st = (hipStream_t)stream;
  if (g_mlp_eps == 2) return C == 96 ? mlp_launch<96, 4, false, true, false, 2>(p, st) : mlp_launch<192, 4, false, true, false, 2>(p, st);
  if (C == 96) return mlp_launch<96, 4, false, true>(p, st);
  return mlp_launch<192, 4, false, true>(p, st);
}

extern "C" int sea_mlp_fused_fwd(const float* x, int64_t ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                                 const float* res, int64_t ldres, float* y, int64_t ldy, int M, int C, int H,
                                 const uint32_t* amax_x, const uint32_t* amax_h, void* stream) {
  return mlp_fwd_impl(x, ldx, W1p, b1, W2p, b2, res, ldres, y, ldy, M, C, H, amax_x, amax_h, nullptr, nullptr, 0.f, stream);
}
// the same with a LayerNorm over the C channels of x in front (x = its INPUT; amax_x bounds its OUTPUT): convnext_orig.py:75-79
extern "C" int sea_ln_mlp_fused_fwd(const float* x, int64_t ldx, const float* ln_w, const float* ln_b, float ln_eps, const void* W1p,
                                    const float* b1, const void* W2p, const float* b2, const float* res, int64_t ldres, float* y,
                                    int64_t ldy, int M, int C, int H, const uint32_t* amax_x, const uint32_t* amax_h, void* stream) {
  SEA_CHECK_ARG(ln_w && ln_b);
  return mlp_fwd_impl(x, ldx, W1p, b1, W2p, b2, res, ldres, y, ldy, M, C, H, amax_x, amax_h, ln_w, ln_b, ln_eps, stream);
}

static int mlp_bwd_impl(const float* g, int64_t ldg, const float* x, int64_t ldx, const void* W1p, const float* b1,
                        const void* W2tp, const void* W1tp, float* dx, int64_t lddx, int M, int C, int H, const uint32_t* amax_x,
                        const float* amax_mul_dev, const float* ln_w, const float* ln_b, float ln_eps, void* stream) {
  MlpArgs p = {};
  p.ln_w = ln_w; p.ln_b = ln_b; p.ln_eps = ln_eps;
  SEA_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr) && ((((uintptr_t)ln_w) | ((uintptr_t)ln_b)) & 15) == 0);
  p.x = x; p.ldx = ldx; p.g = g; p.ldg = ldg; p.Wa = (const char*)W1p; p.Wu = (const char*)W2tp; p.Wb = (const char*)W1tp;
  p.b1 = b1; p.y = dx; p.ldy = lddx; p.M = M; p.H = H; p.Npad = (C + GS_BN - 1) / GS_BN * GS_BN; p.amax1 = amax_x;
  p.amax_mul = amax_mul_dev;
  SEA_CHECK_ARG(mlp_common_ok(p, C) && g && W2tp && amax_mul_dev && (ldg % 4) == 0 && ldg >= C &&
                ((((uintptr_t)g) | ((uintptr_t)W2tp)) & 15) == 0);
  const hipStream_t st = (hipStream_t)stream;
  if (g_mlp_eps == 2) return C == 96 ? mlp_launch<96, 4, true, false, false, 2>(p, st) : mlp_launch<192, 4, true, true, false, 2>(p, st);
  if (C == 96) return mlp_launch<96, 4, true, false>(p, st);
  return mlp_launch<192, 4, true, true>(p, st);
}

extern "C" int sea_mlp_fused_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const void* W1p, const float* b1,
                                 const void* W2tp, const void* W1tp, float* dx, int64_t lddx, int M, int C, int H,
                                 const uint32_t* amax_x, const float* amax_mul_dev, void* stream) {
  return mlp_bwd_impl(g, ldg, x, ldx, W1p, b1, W2tp, W1tp, dx, lddx, M, C, H, amax_x, amax_mul_dev, nullptr, nullptr, 0.f, stream);
}
// input gradient of sea_ln_mlp_fused_fwd w.r.t. the LayerNorm's INPUT x (frozen affine parameters)
extern "C" int sea_ln_mlp_fused_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* ln_w, const float* ln_b,
                                    float ln_eps, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* dx,
                                    int64_t lddx, int M, int C, int H, const uint32_t* amax_x, const float* amax_mul_dev,
                                    void* stream) {
  SEA_CHECK_ARG(ln_w && ln_b);
  return mlp_bwd_impl(g, ldg, x, ldx, W1p, b1, W2tp, W1tp, dx, lddx, M, C, H, amax_x, amax_mul_dev, ln_w, ln_b, ln_eps, stream);
}

// Diagnostic build of the C = 96 kernels (devtools/mlp_fused_stamps.py): the same kernel with s_memtime stamps around the
// segments of a loop iteration; dbg: 8 uint64 per wave (ceil(M / 256) * 8 waves): ticks in [MFMA streams, element-wise work, DMA
// wait, barrier], the loop's total, its start stamp and s_memrealtime at the end.  Not on the product path.
extern "C" int sea_mlp_fused_stamps(int bwd, const float* g, int64_t ldg, const float* x, int64_t ldx, const void* W1p, const float* b1,
                                    const void* W2p, const void* W1tp, const float* b2, const float* res, float* y, int M, int C,
                                    const uint32_t* amax_x, const uint32_t* amax_h, const float* amax_mul_dev,
                                    unsigned long long* dbg, void* stream) {
  SEA_CHECK_ARG(C == 96 && dbg && x && y);
  MlpArgs p = {};
  p.x = x; p.ldx = ldx; p.g = g; p.ldg = ldg; p.Wa = (const char*)W1p; p.b1 = b1; p.y = y; p.ldy = C; p.M = M; p.H = 4 * C;
  p.Npad = (C + GS_BN - 1) / GS_BN * GS_BN; p.amax1 = amax_x; p.amax2 = amax_h; p.amax_mul = amax_mul_dev; p.dbg = dbg;
  if (bwd) {
    p.Wu = (const char*)W2p; p.Wb = (const char*)W1tp;
    return mlp_launch<96, 4, true, false, true>(p, (hipStream_t)stream);
  }
  p.Wb = (const char*)W2p; p.b2 = b2; p.res = res; p.ldres = C;
  return mlp_launch<96, 4, false, true, true>(p, (hipStream_t)stream);
}

// Shared by gemm_split.hip (the three-blocks-per-CU kernel) and gemm_split_pp.hip (the ping-pong pipeline): operand-split
// helpers, the LDS swizzle, the kernel argument block and the GELU formulas of M8.
#pragma once
#include "sea_common.h"
#include <atomic>

namespace sea {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int GS_BM = 128, GS_BN = 128, GS_BK = 32;
constexpr int GS_IMG = GS_BM * GS_BK * 2;  // bytes of one term image (128 rows x 64 B)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // low half = a
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

template <int TERMS>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&out)[TERMS]) {
  float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
#pragma unroll
  for (int t = 0; t < TERMS; ++t) {
    const uint32_t p0 = pack_bf16(r0, r1), p1 = pack_bf16(r2, r3);
    out[t] = u32x2{p0, p1};
    if (t + 1 < TERMS) {  // exact: the rounded-off part of an fp32 number is itself an fp32 number
      r0 -= __uint_as_float(p0 << 16);
      r1 -= __uint_as_float(p0 & 0xffff0000u);
      r2 -= __uint_as_float(p1 << 16);
      r3 -= __uint_as_float(p1 & 0xffff0000u);
    }
  }
}

// fp16 x 2: hi = fp16(a * s), mid = fp16(a * s - hi): 2 x 11 = 22 significant bits per operand, three products
// (hi*hi', hi*mid', mid*hi'; the dropped mid*mid' is 2^-22 relative).  fp16 has 5 exponent bits, so every operand tensor
// is scaled by a power of two s that puts its largest magnitude just below 2^14 (products < 2^28, fp32 accumulate); the
// epilogue multiplies by the exact inverse.  Elements more than 2^28 below the tensor's maximum flush to zero.
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void split4_f16(const f32x4 v, float s, u32x2 (&out)[2]) {
  const float a0 = v[0] * s, a1 = v[1] * s, a2 = v[2] * s, a3 = v[3] * s;
  const uint32_t h0 = pack_f16(a0, a1), h1 = pack_f16(a2, a3);
  const f32x2 f0 = __builtin_convertvector(__builtin_bit_cast(f16x2, h0), f32x2);
  const f32x2 f1 = __builtin_convertvector(__builtin_bit_cast(f16x2, h1), f32x2);
  out[0] = u32x2{h0, h1};
  out[1] = u32x2{pack_f16(a0 - f0[0], a1 - f0[1]), pack_f16(a2 - f1[0], a3 - f1[1])};
}
// power-of-two scale for a tensor whose largest |value| has the float bits `amax_bits`: amax * scale < 2^14
__device__ __host__ __forceinline__ void pow2_scale(uint32_t amax_bits, float& scale, float& inv) {
  int E = (int)((amax_bits >> 23) & 0xffu);   // amax in [2^(E-127), 2^(E-126))
  E = E < 14 ? 14 : (E > 253 ? 253 : E);
  const uint32_t sb = (uint32_t)(267 - E) << 23, ib = (uint32_t)(E - 13) << 23;
  scale = __builtin_bit_cast(float, sb);        // 2^(140 - E)
  inv = __builtin_bit_cast(float, ib);          // 2^(E - 140)
}

// 16-byte chunk swizzle inside a 64-byte row.  ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, +32; a group must touch 16 distinct 16-byte slots of the 256-byte bank row:
//   32x32x16 fragments (lane -> row l & 31, chunk 2s + (l >> 5)):   chunk ^ ((row >> 2) & 3)
//   16x16x32 fragments (lane -> row l & 15, chunk l >> 4):          chunk ^ (-(row >> 2) & 3)
template <bool S16>
__device__ __forceinline__ int swz(int row, int chunk) {
  return S16 ? ((chunk ^ ((0 - (row >> 2)) & 3)) << 4) : ((chunk ^ ((row >> 2) & 3)) << 4);
}
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct GemmSplitArgs {
  const float* A;
  const char* W;
  float* C;
  const float* bias;
  int64_t lda, ldc, strideA, strideW, strideC;
  int M, N, K, Npad;
  int mblocks, nblocks, total, per_xcd;
  int relu;
  const uint32_t* amax_bits;   // fp16 x 2 only: float bits of max |A| (device memory): one word, or one per amax_rows rows
  int amax_rows;
  float amax_mul;              // fp16 x 2 only: the words bound max|A| / amax_mul (a producer-side bound times a constant of the
                               // consumer: ||W||_1 of the GEMM in between, max|GELU'|); 1 = the words as they are
  const float* amax_mul_dev;   // the same constant in device memory (no host round trip when the weights change every step)
  const float* w_inv;          // fp16 x 2 only: per-column inverse weight scale
  uint32_t* out_amax;          // optional: atomic max of the float bits of |C| (pre-zeroed word), for a consumer GEMM
  // fused epilogue extras (sea_gemm_split_fused): all optional
  const float* addend;         // + addend[g][m][n] before the activation (row stride ld_add, batch stride stride_add)
  int64_t ld_add, stride_add;
  float* gelu_out;             // layout of C: receives GELU(C); C keeps the pre-activation
  const float* gelu_grad_of;   // layout of C: the result is multiplied by GELU'(this)
  const float* a_gelu_grad_of; // layout of A: A is read as A * GELU'(this) (the backward of a GELU in front of the GEMM)
  int a_gelu;                  // A is read as GELU(A) (the GELU in front of the GEMM)
  int a_gate;                  // a_gelu_grad_of is a ReLU gate instead: A is read as (gate > 0 ? A : 0)
  int ko;                      // run-time knock-out bits of the timing-only builds (devtools/gemm_knockout.py); 0 in the product
};

// exact (erf) GELU and its derivative, the formulas of ATen's GeluCUDAKernelImpl / GeluBackwardCUDAKernelImpl
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = __expf(-0.5f * x * x) * 0.39894228040143267794f;   // M_2_SQRTPI * M_SQRT1_2 * 0.5
  return cdf + x * pdf;
}

// ---- epilogue of the split-GEMM kernels: a 128 x 128 block tile held as 2 x 2 waves of 64 x 64 in 32x32x16 accumulators.
// smem: at least 32 KB of block LDS that nobody reads any more (wave w uses bytes [8192 w, 8192 (w + 1))); row_inv: the
// block's 128 inverse row scales (F16); bv_c / wi_c: bias and inverse weight scale of this lane's two columns.
template <bool F16, bool EPI, int KO = 0>
__device__ __forceinline__ void gemm_split_store_tile(const GemmSplitArgs& p, const f32x16 (&acc)[2][2], int g, int m0, int n0,
                                                      char* smem, const float* row_inv, const float (&bv_c)[2],
                                                      const float (&wi_c)[2]) {
  const int M = p.M, N = p.N;
  const int64_t ldc = p.ldc, ld_add = p.ld_add;
  const int relu = p.relu;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave & 1, r = lane & 31, h = lane >> 5;
  const float* const addg = (EPI && p.addend) ? p.addend + (int64_t)g * p.stride_add : nullptr;
  float* const gelu_out = (EPI && p.gelu_out) ? p.gelu_out + (int64_t)g * p.strideC : nullptr;
  const float* const gelu_src = (EPI && p.gelu_grad_of) ? p.gelu_grad_of + (int64_t)g * p.strideC : nullptr;
  // The accumulators hold lane = column, 16 registers = rows (reg & 3) + 8 (reg >> 2) + 4 h of a 32 x 32
  // tile: stored as they stand that is 64 four-byte stores per lane, and the store tail of a tile then costs what its whole
  // K loop costs (knock-out timings, profiles/r5_gemm_knockout.md: 655 -> 502 us without the C stores; stores and loads
  // share the address path, so the tail also stalls the OTHER block's loads).  Instead each wave turns its 64 x 64 tile
  // through its own 8 KB of the (now idle) stages, 32 rows at a time, and stores 16 bytes per lane: a wave-instruction
  // writes four 256-byte row segments, 16 of them per lane instead of 64.  No block barrier: after the last in-loop
  // barrier nobody needs the stages (the fragments in flight are in registers), and a wave only reads what it wrote.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int row0_u = m0 + (wave_u >> 1) * 64;
  float* const Cg = p.C + (int64_t)g * p.strideC;
  char* const scr = smem + wave_u * 8192;
  const int lr = lane >> 4, lc = lane & 15;                   // read side: row 4 k + lr, columns 4 lc .. 4 lc + 3
  const int col4 = n0 + wn * 64 + 4 * lc;
  const bool vec = (((ldc | p.strideC | N) & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0) &&
                   (!(EPI && addg) || ((((ld_add | p.stride_add) & 3) == 0) && ((((uintptr_t)p.addend) & 15) == 0))) &&
                   (!(EPI && gelu_out) || ((((uintptr_t)p.gelu_out) & 15) == 0)) &&
                   (!(EPI && gelu_src) || ((((uintptr_t)p.gelu_grad_of) & 15) == 0));
  const int off_c = lr * (int)ldc + col4;                     // (ldc < 2^28: checked by the launcher)
  const int off_a = (EPI && addg) ? lr * (int)ld_add + col4 : 0;
  uint32_t omax = 0;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    if (mi) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the reads of the first half are done)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row_l = (e & 3) + 8 * (e >> 2) + 4 * h;
        // (exact: both scales are powers of two)
        const float v = (F16 ? acc[mi][ni][e] * (row_inv[(wave_u >> 1) * 64 + mi * 32 + row_l] * wi_c[ni]) : acc[mi][ni][e]) + bv_c[ni];
        *(float*)(scr + row_l * 256 + (ni * 32 + r) * 4) = v;
      }
      (void)col;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int row_u = row0_u + mi * 32 + 4 * k;
      f32x4 v = *(const f32x4*)(scr + k * 1024 + lr * 256 + lc * 16);
      if (row_u + lr >= M || col4 >= N) continue;
      float* const crow = Cg + (int64_t)row_u * ldc;
      if (vec) {
        if (EPI && addg) v += *(const f32x4*)((addg + (int64_t)row_u * ld_add) + off_a);
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (EPI && gelu_src) {
          const f32x4 t = *(const f32x4*)((gelu_src + (int64_t)row_u * ldc) + off_c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(t[e]);
        }
        if constexpr ((KO & 64) != 0)
          __builtin_nontemporal_store(v, (f32x4*)(crow + off_c));
        else if ((KO & 1) == 0 || __float_as_uint(v[0]) == 0x7fc12345u)
          *(f32x4*)(crow + off_c) = v;
        if (EPI && gelu_out) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = gelu_f(v[e]);
          *(f32x4*)((gelu_out + (int64_t)row_u * ldc) + off_c) = o;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t vb = __float_as_uint(v[e]) & 0x7fffffffu;
          omax = vb > omax ? vb : omax;
        }
      } else {   // unaligned or ragged output: element by element
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (col4 + e < N) {
            float x = v[e];
            if (EPI && addg) x += (addg + (int64_t)row_u * ld_add)[off_a + e];
            if (relu) x = x > 0.f ? x : 0.f;
            if (EPI && gelu_src) x *= gelu_grad_f((gelu_src + (int64_t)row_u * ldc)[off_c + e]);
            crow[off_c + e] = x;
            if (EPI && gelu_out) (gelu_out + (int64_t)row_u * ldc)[off_c + e] = gelu_f(x);
            const uint32_t vb = __float_as_uint(x) & 0x7fffffffu;
            omax = vb > omax ? vb : omax;
          }
        }
      }
    }
  }
  if (p.out_amax != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t other = (uint32_t)__shfl_xor((int)omax, o, 64);
      omax = other > omax ? other : omax;
    }
    if (lane == 0 && omax > *(volatile uint32_t*)p.out_amax) atomicMax(p.out_amax, omax);
  }
}

// launcher of the ping-pong kernel (gemm_split_pp.hip); returns false when the variant is not built for this mode
bool gemm_split_pp_launch(const GemmSplitArgs& p, int terms, int pro, bool fused, hipStream_t st);
// launcher of the one-block-per-CU kernels (gemm_split_big.hip): two terms, no fused extras; shape 0 = 256 x 256 tiles
// (N % 256 == 0, no prologue), shape 1 = 128 x 384 tiles (N % 384 == 0, any prologue)
bool gemm_split_big_launch(GemmSplitArgs p, int terms, int batch, int shape, int pro, hipStream_t st);

}  // namespace sea

// M8 (model side): fp32 GEMM on the bf16 matrix cores by operand splitting.
//
//   C[g] (M x N, fp32) = A[g] (M x K, fp32, row-major) * W[g]^T (W: N x K, frozen weights) [+ bias[n]] [ReLU]
//
// The model side of an attack step is ~1.9 TFLOP of fp32 GEMMs with FROZEN weights (ConvNeXt point-wise layers, the
// UperNet head's Winograd-domain products, reference semseg/models/uperforseg.py:200-215, 255-262 and
// backbones/convnext.py blocks), which hipBLASLt runs on v_mfma_f32_32x32x2_f32 at 100-125 TFLOP/s (peak 157).  The
// bf16 pipe is 16x faster per instruction.  An fp32 number is the exact sum of three bf16 numbers
// (hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid): 3 x 8 significant bits = the 24 of fp32, and every
// subtraction is exact), so
//      a * w = hi*hi' + hi*mid' + mid*hi' + hi*lo' + mid*mid' + lo*hi'  (+ terms below 2^-24 relative: dropped)
// is six v_mfma_f32_32x32x16_bf16 products accumulated in fp32: fp32-level accuracy at a 2.5 PFLOP/s / 6 = 416 TFLOP/s
// ceiling.  TERMS = 2 keeps hi + mid (16 significant bits, 3 products, 833 TFLOP/s ceiling) for callers that accept it.
//
// Data flow per 128 x 128 output tile and 32-deep K step (256 threads, 4 waves as 2 x 2, 64 x 64 outputs per wave):
//   A   : fp32 from global (any producer: no extra pass over the activations), split in registers (5.5 VALU ops per
//         element, hidden in the MFMA shadow), written to LDS as TERMS bf16 images
//   W   : split ONCE on the host side of the ABI (sea_gemm_split_pack) into the tile order the kernel reads:
//         [K/32][TERMS][Npad][32] bf16, so a tile is TERMS contiguous 8 KB pieces
//   LDS : [term][row][32 bf16] with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3: both the 8/16-byte
//         staging writes and the ds_read_b128 fragment reads of 32 consecutive rows are conflict-free, no padding
//         (48 KB per block at TERMS = 3 -> three blocks per CU, which is what hides the staging phases)
//   MFMA: per wave and 16-deep step 12 fragment reads feed 24 MFMAs (TERMS = 3): 0.5 LDS reads per MFMA
//   The next tile's global loads are issued before the MFMAs of the current one (register prefetch).
// Tried and dropped (profiles/r3_gemm_split_variants.md): a 128 x 256 producer / consumer-wave kernel with two 72 KB LDS
// buffers (1 block per CU: the producers' load latency, one K step ahead at most for lack of registers and LDS, set the
// pace: 124 vs 151 TFLOP/s); v_mfma_f32_16x16x32_bf16 fragments (kept as SEA_GEMM_SHAPE=16: within +-3 %).
// Fixed summation order, no atomics: bitwise reproducible.  inf / NaN inputs: a +-inf operand gives NaN (inf - inf in
// the split) where an fp32 GEMM may give inf.
#include "gemm_split.h"

namespace sea {

// EPI: the fused epilogue extras (addend / gelu_out / gelu_grad_of) are compiled in; PRO: the prologue factor GELU'(.) on A
// (PRO = 1: A * GELU'(second operand); PRO = 2: GELU(A); PRO = 3: second operand > 0 ? A : 0)
template <int TERMS, bool S16, bool F16 = false, bool EPI = false, int PRO = 0>
__global__ __launch_bounds__(256, 3) void gemm_split_kernel(const GemmSplitArgs p) {
  // (at least 32 KB: the epilogue turns the tile through 8 KB per wave)
  __shared__ __attribute__((aligned(16))) char smem[2 * TERMS * GS_IMG < 32768 ? 32768 : 2 * TERMS * GS_IMG];
  char* As = smem;
  char* Bs = smem + TERMS * GS_IMG;

  // ---- tile of this block: XCD k works on the k-th contiguous eighth of the tile list, n-blocks fastest, so the
  // blocks that share an A tile run on one XCD at the same time and meet in its L2
  const int M = p.M, N = p.N, K = p.K, Npad = p.Npad;   // (locals: the lambdas below must not take the address of p)
  const int64_t lda = p.lda, ldc = p.ldc;
  const int logical = (int)(blockIdx.x & 7) * p.per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.per_xcd || logical >= p.total) return;
  const int nb = logical % p.nblocks;
  const int t2 = logical / p.nblocks;
  const int mb = t2 % p.mblocks;
  const int g = t2 / p.mblocks;
  const int m0 = mb * GS_BM, n0 = nb * GS_BN;
  const float* const w_inv = F16 ? (const float*)((const char*)p.w_inv + (int64_t)g * p.strideW) : nullptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- staging assignment
  // A: thread -> float4 column q of rows (tid >> 3) + 32 i
  const int q = tid & 7, arow = tid >> 3;
  // addresses = wave-uniform 64-bit base (SGPRs) + one 32-bit byte offset per lane: no 64-bit VALU address arithmetic and
  // 5 address registers for the 10 loads of a K step (the launcher checks that a batch of A spans < 4 GB)
  const char* const Abase = (const char*)(p.A + (int64_t)g * p.strideA);
  float* const Cbase = p.C;
  const float* const bias = p.bias;
  const int64_t strideC = p.strideC;
  const int relu = p.relu;
  const float* const addg = (EPI && p.addend) ? p.addend + (int64_t)g * p.stride_add : nullptr;
  const int64_t ld_add = p.ld_add;
  float* const gelu_out = (EPI && p.gelu_out) ? p.gelu_out + (int64_t)g * p.strideC : nullptr;
  const float* const gelu_src = (EPI && p.gelu_grad_of) ? p.gelu_grad_of + (int64_t)g * p.strideC : nullptr;
  // fp16 x 2: one power-of-two scale per ROW of the tile, from the word that covers the row (the whole tensor, an image,
  // a Winograd tile ..): rows scaled by their own group's maximum make an image's result independent of its batch partners
  __shared__ float row_sc[F16 ? GS_BM : 1], row_inv[F16 ? GS_BM : 1];
  float a_sc[4] = {1.f, 1.f, 1.f, 1.f};
  // (called BEHIND the first tile's loads: the scale words are a second memory round trip, and a barrier)
  auto load_row_scales = [&]() {
   if constexpr (F16) {
    if (tid < GS_BM) {
      int row = m0 + tid;
      row = row < M ? row : M - 1;
      float sc, inv;
      uint32_t word = p.amax_bits[p.amax_rows > 0 ? row / p.amax_rows : 0];
      const float mul = p.amax_mul_dev ? *p.amax_mul_dev : p.amax_mul;
      if (mul != 1.f) word = __float_as_uint(__uint_as_float(word) * mul) & 0x7fffffffu;
      pow2_scale(word, sc, inv);
      row_sc[tid] = sc;
      row_inv[tid] = inv;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) a_sc[i] = row_sc[arow + 32 * i];
   }
  };
  uint32_t aoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = m0 + arow + 32 * i;
    row = row < M ? row : M - 1;  // tail rows: read something valid, never stored
    aoff[i] = (uint32_t)(((int64_t)row * lda + 4 * q) * 4);
  }
  // W: 16-byte piece pc = tid + 256 i -> (term i >> 1, row, chunk); a tile is TERMS contiguous 8 KB pieces of the packed
  // array, so every lane reads base(kb, i) + 16 tid
  const char* const Wbase = p.W + (int64_t)g * p.strideW + (int64_t)n0 * 64;
  const int64_t w_term = (int64_t)Npad * 64, w_kb = (int64_t)TERMS * w_term;
  const uint32_t woff = (uint32_t)tid * 16;

  f32x4 pa[4];
  f32x4 pt[(PRO == 1 || PRO == 3) ? 4 : 1];
  const char* const Tbase = (PRO == 1 || PRO == 3) ? (const char*)(p.a_gelu_grad_of + (int64_t)g * p.strideA) : nullptr;
  u32x4 pw[2 * TERMS];
  auto fetch = [&](int kb) {
    const char* a = Abase + (int64_t)kb * (GS_BK * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[i] = *(const f32x4*)(a + aoff[i]);
    if constexpr (PRO == 1 || PRO == 3) {
      const char* t = Tbase + (int64_t)kb * (GS_BK * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) pt[i] = *(const f32x4*)(t + aoff[i]);
    }
    const char* w = Wbase + (int64_t)kb * w_kb;
#pragma unroll
    for (int i = 0; i < 2 * TERMS; ++i) pw[i] = *(const u32x4*)(w + (int64_t)(i >> 1) * w_term + (i & 1) * 4096 + woff);
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 s[TERMS];
      if constexpr (PRO == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pa[i][e] *= gelu_grad_f(pt[i][e]);
      }
      if constexpr (PRO == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pa[i][e] = gelu_f(pa[i][e]);
      }
      if constexpr (PRO == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pa[i][e] = pt[i][e] > 0.f ? pa[i][e] : 0.f;
      }
      if constexpr (F16)
        split4_f16(pa[i], a_sc[i], s);
      else
        split4<TERMS>(pa[i], s);
      const int row = arow + 32 * i;
#pragma unroll
      for (int t = 0; t < TERMS; ++t) *(u32x2*)(As + t * GS_IMG + row * 64 + swz<S16>(row, q >> 1) + (q & 1) * 8) = s[t];
    }
#pragma unroll
    for (int i = 0; i < 2 * TERMS; ++i) {
      const int pc = tid + 256 * i, term = pc >> 9, row = (pc & 511) >> 2, chunk = pc & 3;
      *(u32x4*)(Bs + term * GS_IMG + row * 64 + swz<S16>(row, chunk)) = pw[i];
    }
  };

  const int nkb = K / GS_BK;
  float* Cg = Cbase + (int64_t)g * strideC;
  if constexpr (!S16) {
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    fetch(0);
    load_row_scales();
    // (the epilogue's per-column constants are loaded here, behind the first tile: a round trip less after the K loop)
    float bv_c[2], wi_c[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + r;
      bv_c[ni] = (bias && col < N) ? bias[col] : 0.f;
      wi_c[ni] = (F16 && col < N) ? w_inv[col] : 1.f;
    }
    for (int kb = 0; kb < nkb; ++kb) {
      stage();
      __syncthreads();
      fetch(kb + 1 < nkb ? kb + 1 : kb);  // (unconditional: the last step re-reads its own tile, nothing is staged from it)
      __builtin_amdgcn_sched_barrier(0);   // the loads go out BEFORE the MFMAs (the scheduler sinks them to the loop end otherwise)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[2][TERMS], b[2][TERMS];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row = wm * 64 + mi * 32 + r;
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            a[mi][t] = *(const bf16x8*)(As + t * GS_IMG + row * 64 + swz<false>(row, 2 * s + h));
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int row = wn * 64 + ni * 32 + r;
#pragma unroll
          for (int t = 0; t < TERMS; ++t)
            b[ni][t] = *(const bf16x8*)(Bs + t * GS_IMG + row * 64 + swz<false>(row, 2 * s + h));
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            f32x16 c = acc[mi][ni];
            // smallest products first
            if constexpr (F16) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[mi][1]), __builtin_bit_cast(f16x8, b[ni][0]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[mi][0]), __builtin_bit_cast(f16x8, b[ni][1]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[mi][0]), __builtin_bit_cast(f16x8, b[ni][0]), c, 0, 0, 0);
            } else {
              if constexpr (TERMS == 3) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], c, 0, 0, 0);
              }
              if constexpr (TERMS >= 2) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], c, 0, 0, 0);
              }
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], c, 0, 0, 0);
            }
            acc[mi][ni] = c;
          }
      }
      __syncthreads();
    }
    // ---- epilogue (gemm_split.h): through the idle stages, 16-byte stores (the K loop ends behind a barrier)
    gemm_split_store_tile<F16, EPI>(p, acc, g, m0, n0, smem, row_inv, bv_c, wi_c);
  } else {
    // ---- 16x16x32 fragments: one 32-deep MFMA step per staged tile, 4 x 4 output tiles of 16 x 16 per wave.  The chip
    // holds a higher clock on this shape in MFMA-dense loops (MI355X guide, DVFS give-back item 7).
    const int r16 = lane & 15, c16 = lane >> 4;
    f32x4v acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    load_row_scales();
    for (int kb = 0; kb < nkb; ++kb) {
      stage();
      __syncthreads();
      fetch(kb + 1 < nkb ? kb + 1 : kb);
      __builtin_amdgcn_sched_barrier(0);
      // two column halves: 2 x TERMS B fragments stay in registers while the four row tiles stream past them
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        bf16x8 b[2][TERMS];
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const int row = wn * 64 + (2 * nh + nj) * 16 + r16;
#pragma unroll
          for (int t = 0; t < TERMS; ++t) b[nj][t] = *(const bf16x8*)(Bs + t * GS_IMG + row * 64 + swz<true>(row, c16));
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          bf16x8 a[TERMS];
          const int row = wm * 64 + mi * 16 + r16;
#pragma unroll
          for (int t = 0; t < TERMS; ++t) a[t] = *(const bf16x8*)(As + t * GS_IMG + row * 64 + swz<true>(row, c16));
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) {
            f32x4v c = acc[mi][2 * nh + nj];
            if constexpr (F16) {
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[nj][0]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[nj][1]), c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[nj][0]), c, 0, 0, 0);
            } else {
              if constexpr (TERMS == 3) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[nj][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nj][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[nj][1], c, 0, 0, 0);
              }
              if constexpr (TERMS >= 2) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[nj][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nj][1], c, 0, 0, 0);
              }
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nj][0], c, 0, 0, 0);
            }
            acc[mi][2 * nh + nj] = c;
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the next row tile's fragment reads behind this tile's MFMAs
        }
      }
      __syncthreads();
    }
    // ---- epilogue: lane & 15 = column, register e = row 4 (lane >> 4) + e of the 16 x 16 tile
    uint32_t omax16 = 0;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = n0 + wn * 64 + ni * 16 + r16;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
      const float wi = F16 ? w_inv[col] : 1.f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = m0 + wm * 64 + mi * 16 + 4 * c16 + e;
          if (row < M) {
            float v = (F16 ? acc[mi][ni][e] * (row_inv[row - m0] * wi) : acc[mi][ni][e]) + bv;
            if (EPI && addg) v += addg[(int64_t)row * ld_add + col];
            if (relu) v = v > 0.f ? v : 0.f;
            if (EPI && gelu_src) v *= gelu_grad_f(gelu_src[(int64_t)row * ldc + col]);
            Cg[(int64_t)row * ldc + col] = v;
            if (EPI && gelu_out) gelu_out[(int64_t)row * ldc + col] = gelu_f(v);
            const uint32_t vb = __float_as_uint(v) & 0x7fffffffu;
            omax16 = vb > omax16 ? vb : omax16;
          }
        }
      }
    }
    if (p.out_amax != nullptr) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)omax16, o, 64);
        omax16 = other > omax16 ? other : omax16;
      }
      if (lane == 0 && omax16 > *(volatile uint32_t*)p.out_amax) atomicMax(p.out_amax, omax16);
    }
  }
}

// W (fp32) -> [K/32][TERMS][Npad][32] bf16.  trans = 0: W is (N, K) row-major with row stride ldw; trans = 1: W is
// (K, N) row-major (out[n][k] = W[k][n]).  Rows n >= N are zero.
template <int TERMS>
__global__ void gemm_split_pack_kernel(const float* __restrict__ W, int64_t ldw, int trans, int N, int K, int Npad,
                                       uint16_t* __restrict__ out) {
  const int64_t total = (int64_t)(K / 32) * Npad * 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 31);
    const int64_t t = i >> 5;
    const int n = (int)(t % Npad);
    const int kb = (int)(t / Npad);
    const int k = kb * 32 + kk;
    float rem = 0.f;
    if (n < N) rem = trans ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k];
#pragma unroll
    for (int term = 0; term < TERMS; ++term) {
      const uint32_t pk = pack_bf16(rem, 0.f);
      out[(((int64_t)kb * TERMS + term) * Npad + n) * 32 + kk] = (uint16_t)(pk & 0xffffu);
      rem -= __uint_as_float(pk << 16);
    }
  }
}

// max |x| of a (batch of) row-strided matrix as float bits (non-negative floats order like their bit patterns):
// order-independent, so the atomic makes it deterministic.  `out` must be zeroed before.
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ A, int64_t lda, int M, int K4,
                                                          int64_t strideA, int batch, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  auto take = [&](const f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t b = __float_as_uint(v[e]) & 0x7fffffffu;
      m = b > m ? b : m;
    }
  };
  if (lda == (int64_t)4 * K4 && (batch == 1 || strideA == (int64_t)M * lda)) {
    // dense: one flat stream, two 16-byte loads in flight per lane (the GEMM that follows finds the data in L2 / MALL)
    const int64_t total = (int64_t)batch * M * K4, stride = (int64_t)gridDim.x * blockDim.x;
    const f32x4* p = reinterpret_cast<const f32x4*>(A);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += 2 * stride) {
      const int64_t j = i + stride;
      const f32x4 v0 = p[i], v1 = p[j < total ? j : i];
      take(v0);
      take(v1);
    }
  } else {
    // row-strided: a block walks whole rows
    const int64_t rows = (int64_t)batch * M;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
      const int64_t g = r / M;
      const float* row = A + g * strideA + (r - g * M) * lda;
      for (int k4 = threadIdx.x; k4 < K4; k4 += blockDim.x) take(*reinterpret_cast<const f32x4*>(row + 4 * k4));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)m, o, 64);
    m = other > m ? other : m;
  }
  // ONE atomic per block: thousands of same-address atomics serialise in L2 (8192 of them cost ~100 us)
  __shared__ uint32_t wave_max[4];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t b = wave_max[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = wave_max[w] > b ? wave_max[w] : b;
    if (b) atomicMax(out, b);
  }
}

// per-row max |w| bits of W (N x K, or K x N with trans)
__global__ void rowmax_bits_kernel(const float* __restrict__ W, int64_t ldw, int trans, int N, int K, uint32_t* __restrict__ out) {
  const int n = blockIdx.x;
  uint32_t m = 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const uint32_t b = __float_as_uint(trans ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)m, o, 64);
    m = other > m ? other : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out + n, m);
}

// W -> [K/32][2][Npad][32] fp16 (hi, mid) of w * 2^s_n, + the inverse scales 2^-s_n (Npad floats) behind the images
__global__ void gemm_split_pack_f16_kernel(const float* __restrict__ W, int64_t ldw, int trans, int N, int K, int Npad,
                                           const uint32_t* __restrict__ rowmax, uint16_t* __restrict__ out,
                                           float* __restrict__ w_inv) {
  const int64_t total = (int64_t)(K / 32) * Npad * 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 31);
    const int64_t t = i >> 5;
    const int n = (int)(t % Npad);
    const int kb = (int)(t / Npad);
    const int k = kb * 32 + kk;
    float sc = 1.f, inv = 1.f;
    if (n < N) pow2_scale(rowmax[n], sc, inv);
    if (kb == 0 && kk == 0) w_inv[n] = n < N ? inv : 0.f;
    const float w = n < N ? (trans ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k]) * sc : 0.f;
    const uint32_t hi = pack_f16(w, 0.f) & 0xffffu;
    const float hf = (float)__builtin_bit_cast(f16x2, hi)[0];
    const uint32_t mid = pack_f16(w - hf, 0.f) & 0xffffu;
    out[(((int64_t)kb * 2 + 0) * Npad + n) * 32 + kk] = (uint16_t)hi;
    out[(((int64_t)kb * 2 + 1) * Npad + n) * 32 + kk] = (uint16_t)mid;
  }
}

// out[j] = float bits of max |A[g][r][:]| over all g and the rows r of group j (rows j rpw .. (j + 1) rpw - 1)
__global__ __launch_bounds__(256) void absmax_groups_kernel(const float* __restrict__ A, int64_t lda, int M, int K4,
                                                            int64_t strideA, int batch, int rpw, uint32_t* __restrict__ out) {
  const int j = blockIdx.y;
  const int r0 = j * rpw, r1 = (r0 + rpw < M) ? r0 + rpw : M;
  uint32_t m = 0;
  const int64_t n = (int64_t)(r1 - r0) * K4;
  for (int g = 0; g < batch; ++g) {
    const float* base = A + (int64_t)g * strideA + (int64_t)r0 * lda;
    if (lda == (int64_t)4 * K4) {
      const f32x4* p = reinterpret_cast<const f32x4*>(base);
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = p[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t b = __float_as_uint(v[e]) & 0x7fffffffu;
          m = b > m ? b : m;
        }
      }
    } else {
      for (int r = blockIdx.x; r < r1 - r0; r += gridDim.x)
        for (int k4 = threadIdx.x; k4 < K4; k4 += blockDim.x) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)r * lda + 4 * k4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t b = __float_as_uint(v[e]) & 0x7fffffffu;
            m = b > m ? b : m;
          }
        }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t other = (uint32_t)__shfl_xor((int)m, o, 64);
    m = other > m ? other : m;
  }
  __shared__ uint32_t wave_max[4];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t b = wave_max[0];
    for (int w = 1; w < 4; ++w) b = wave_max[w] > b ? wave_max[w] : b;
    if (b > *(volatile uint32_t*)(out + j)) atomicMax(out + j, b);
  }
}

// out[r] = float bits of max_k |A[r][k]|: ONE WORD PER ROW (the fp16 x 2 scale of a gradient operand, whose rows -- pixels --
// span many orders of magnitude).  LPR lanes share a row (LPR = 8 .. 64, a power of two), 256 / LPR rows per block; every
// word has one writer: no atomics, no zero fill, any grid.  Four independent 16-byte loads in flight per lane.
template <int LPR>
__global__ __launch_bounds__(256) void rowmax_rows_kernel(const float* __restrict__ A, int64_t lda, int M, int K4,
                                                          uint32_t* __restrict__ out) {
  constexpr int RPB = 256 / LPR;
  const int j = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < M; r0 += (int64_t)gridDim.x * RPB) {
    const int64_t row = r0 + rl;
    uint32_t m = 0;
    if (row < M) {
      const f32x4* p = reinterpret_cast<const f32x4*>(A + row * lda);
      const int last = K4 - 1;
      for (int k = j; k < K4; k += 4 * LPR) {
        const int k1 = k + LPR, k2 = k + 2 * LPR, k3 = k + 3 * LPR;   // (clamped duplicates are harmless for a maximum)
        const f32x4 v0 = p[k], v1 = p[k1 < last ? k1 : last], v2 = p[k2 < last ? k2 : last], v3 = p[k3 < last ? k3 : last];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t b0 = __float_as_uint(v0[e]) & 0x7fffffffu, b1 = __float_as_uint(v1[e]) & 0x7fffffffu;
          const uint32_t b2 = __float_as_uint(v2[e]) & 0x7fffffffu, b3 = __float_as_uint(v3[e]) & 0x7fffffffu;
          const uint32_t a = b0 > b1 ? b0 : b1, b = b2 > b3 ? b2 : b3, c = a > b ? a : b;
          m = c > m ? c : m;
        }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      const uint32_t other = (uint32_t)__shfl_xor((int)m, o, 64);
      m = other > m ? other : m;
    }
    if (j == 0 && row < M) out[row] = m;
  }
}

// zero-fill of a few words as a KERNEL: inside a captured HIP graph a hipMemsetAsync node was observed to run out of order
// with the neighbouring kernel nodes (the max|A| word was cleared after sea_absmax_bits had accumulated into it: replays
// differed from the eager loop on 1 of 6 runs with 8 such nodes per graph, on 6 of 6 with 48)
__global__ void zero_words_kernel(uint32_t* __restrict__ p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}

// C[m][n] = act(sum_s partial[s][m][n] + bias[n]) in the fixed order s = 0, 1, ..: the second pass of a split-K product
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const f32x4* __restrict__ part, int S, int64_t per_split4, int M,
                                                            int N4, const f32x4* __restrict__ bias, int relu,
                                                            float* __restrict__ C, int64_t ldc, uint32_t* __restrict__ out_amax,
                                                            const float* __restrict__ addend, int64_t ld_add) {
  const int64_t total = (int64_t)M * N4;
  uint32_t omax = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / N4;
    const int c4 = (int)(i - row * N4);
    f32x4 v = part[i];
    for (int sidx = 1; sidx < S; ++sidx) v += part[sidx * per_split4 + i];
    if (bias) v += bias[c4];
    if (addend) v += *(const f32x4*)(addend + row * ld_add + 4 * c4);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    *(f32x4*)(C + row * ldc + 4 * c4) = v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t vb = __float_as_uint(v[e]) & 0x7fffffffu;
      omax = vb > omax ? vb : omax;
    }
  }
  if (out_amax != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t other = (uint32_t)__shfl_xor((int)omax, o, 64);
      omax = other > omax ? other : omax;
    }
    if ((threadIdx.x & 63) == 0 && omax > *(volatile uint32_t*)out_amax) atomicMax(out_amax, omax);
  }
}

}  // namespace sea

using namespace sea;

static inline int gs_npad(int N) { return (N + GS_BN - 1) / GS_BN * GS_BN; }

static std::atomic<int> g_mfma_shape{[] {
  const char* e = getenv("SEA_GEMM_SHAPE");
  return (e && e[0] == '1') ? 16 : 32;
}()};

// tuning knob: MFMA fragment shape of sea_gemm_split* (32 = v_mfma_f32_32x32x16_*, 16 = v_mfma_f32_16x16x32_*); any other
// value only queries.  Returns the previous shape.  Results of the two shapes differ in the last bits (summation order).
static const int g_wide_min = [] {   // (A/B knob: smallest tile count for the 128 x 384 kernel; 0x7fffffff switches it off)
  const char* e = getenv("SEA_GEMM_WIDE_MIN");
  return e ? atoi(e) : 192;
}();

static std::atomic<int> g_pipeline{[] {
  const char* e = getenv("SEA_GEMM_PIPE");
  return (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 2;
}()};

// tuning knob: K-loop pipeline of sea_gemm_split* at one or two terms per operand: 0 = the single-stage loop (32 KB of LDS, three
// blocks per CU), 1 = ping-pong (two LDS stages, one barrier per K step, the split in the MFMA shadow, loads two K steps ahead;
// two blocks per CU), 2 = per launch (default): ping-pong for K >= 768 per block and for the K slices of a split-K product
// (row stride > K), the single-stage loop otherwise (A/B over the step's shapes: profiles/r5_gemm_pipe_ab.md).  Any other value
// only queries.  Returns the previous setting.  The two kernels give the SAME BITS (same split, same MFMA order).
extern "C" int sea_gemm_split_pipeline(int pipe) {
  const int prev = g_pipeline.load(std::memory_order_relaxed);
  if (pipe >= 0 && pipe <= 3) g_pipeline.store(pipe, std::memory_order_relaxed);
  return prev;
}

extern "C" int sea_gemm_split_mfma_shape(int shape) {
  const int prev = g_mfma_shape.load(std::memory_order_relaxed);
  if (shape == 16 || shape == 32) g_mfma_shape.store(shape, std::memory_order_relaxed);
  return prev;
}

// terms: 3 / 2 = bf16 terms per operand; 22 = fp16 x 2 (22 significant bits, per-tensor power-of-two scaling)
extern "C" int64_t sea_gemm_split_packed_bytes(int N, int K, int terms) {
  if (N <= 0 || K <= 0 || K % GS_BK || (terms != 1 && terms != 2 && terms != 3 && terms != 22)) return -1;
  if (terms == 22) return (int64_t)(K / GS_BK) * 2 * gs_npad(N) * GS_BK * 2 + (int64_t)gs_npad(N) * 8;   // + inverse scales, row maxima
  return (int64_t)(K / GS_BK) * terms * gs_npad(N) * GS_BK * 2;
}

extern "C" int sea_gemm_split_pack(const float* W, int64_t ldw, int trans, int N, int K, int terms, void* out,
                                   void* stream) {
  SEA_CHECK_ARG(W && out && N > 0 && K > 0 && (K % GS_BK) == 0 && (terms == 1 || terms == 2 || terms == 3 || terms == 22));
  SEA_CHECK_ARG(ldw >= (trans ? N : K));
  const int Npad = gs_npad(N);
  const int64_t total = (int64_t)(K / 32) * Npad * 32;
  const int grid = grid_for(total, 256);
  if (terms == 22) {
    char* base = (char*)out;
    float* w_inv = (float*)(base + (int64_t)(K / GS_BK) * 2 * Npad * GS_BK * 2);
    uint32_t* rowmax = (uint32_t*)(w_inv + Npad);
    hipLaunchKernelGGL(zero_words_kernel, dim3((Npad + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowmax, Npad);
    hipLaunchKernelGGL(rowmax_bits_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, W, ldw, trans, N, K, rowmax);
    hipLaunchKernelGGL(gemm_split_pack_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, trans, N, K, Npad,
                       rowmax, (uint16_t*)out, w_inv);
    SEA_RETURN_LAST();
  }
  if (terms == 1)
    hipLaunchKernelGGL(gemm_split_pack_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, trans, N, K, Npad,
                       (uint16_t*)out);
  else if (terms == 3)
    hipLaunchKernelGGL(gemm_split_pack_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, trans, N, K, Npad,
                       (uint16_t*)out);
  else
    hipLaunchKernelGGL(gemm_split_pack_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, trans, N, K, Npad,
                       (uint16_t*)out);
  SEA_RETURN_LAST();
}

static int gemm_split_impl(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias, int relu, int M,
                           int N, int K, int terms, int batch, int64_t strideA, int64_t strideW_bytes, int64_t strideC,
                           const uint32_t* amax_bits, int amax_rows, uint32_t* out_amax, void* stream,
                           const SeaGemmEpilogue* epi = nullptr);

extern "C" int sea_gemm_split(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias,
                              int relu, int M, int N, int K, int terms, int batch, int64_t strideA, int64_t strideW_bytes,
                              int64_t strideC, void* stream) {
  SEA_CHECK_ARG(terms == 1 || terms == 2 || terms == 3);
  return gemm_split_impl(A, lda, Wp, C, ldc, bias, relu, M, N, K, terms, batch, strideA, strideW_bytes, strideC, nullptr, 0,
                         nullptr, stream);
}

// fp16 x 2 operands (weights packed with terms = 22).  amax_bits: device word that sea_absmax_bits filled for THIS A.
// out_amax (optional, pre-zeroed word): receives the float bits of max |C| -- the scale of a consumer GEMM whose input is an
// element-wise function of C that does not grow it (GELU, ReLU), without another pass over C.
extern "C" int sea_gemm_split_f16(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias,
                                  int relu, int M, int N, int K, int batch, int64_t strideA, int64_t strideW_bytes,
                                  int64_t strideC, const uint32_t* amax_bits, int amax_rows, uint32_t* out_amax,
                                  void* stream) {
  SEA_CHECK_ARG(amax_bits != nullptr && amax_rows >= 0);
  return gemm_split_impl(A, lda, Wp, C, ldc, bias, relu, M, N, K, 22, batch, strideA, strideW_bytes, strideC, amax_bits,
                         amax_rows, out_amax, stream);
}

// act(A W^T + bias + addend) with optional second output GELU(.) / factor GELU'(.): see include/sea_hip.h
extern "C" int sea_gemm_split_fused(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias,
                                    int relu, int M, int N, int K, int terms, int batch, int64_t strideA,
                                    int64_t strideW_bytes, int64_t strideC, const uint32_t* amax_bits, int amax_rows,
                                    uint32_t* out_amax, const SeaGemmEpilogue* epi, void* stream) {
  SEA_CHECK_ARG(terms == 1 || terms == 2 || terms == 3 || (terms == 22 && amax_bits != nullptr && amax_rows >= 0));
  SEA_CHECK_ARG(terms == 22 || out_amax == nullptr);
  if (epi) {
    SEA_CHECK_ARG(!epi->addend || epi->ld_addend >= N);
    SEA_CHECK_ARG(!(epi->gelu_out && epi->gelu_grad_of) && !(relu && (epi->gelu_out || epi->gelu_grad_of)));
    SEA_CHECK_ARG(!(epi->a_gelu_grad_of || epi->a_gelu) || !(epi->addend || epi->gelu_out || epi->gelu_grad_of));
    SEA_CHECK_ARG(!(epi->a_gelu_grad_of && epi->a_gelu));
  }
  return gemm_split_impl(A, lda, Wp, C, ldc, bias, relu, M, N, K, terms, batch, strideA, strideW_bytes, strideC,
                         terms == 22 ? amax_bits : nullptr, amax_rows, out_amax, stream, epi);
}

// split-K second pass: C (M x N, row stride ldc) = act(sum over `splits` partial products (each M x N, dense) + bias)
extern "C" int sea_gemm_splitk_reduce(const float* partial, int splits, int M, int N, const float* bias, const float* addend,
                                      int64_t ld_addend, int relu, float* C, int64_t ldc, uint32_t* out_amax, void* stream) {
  SEA_CHECK_ARG(partial && C && splits >= 1 && M > 0 && N > 0 && (N % 4) == 0 && ldc >= N && (ldc % 4) == 0);
  SEA_CHECK_ARG(((((uintptr_t)partial) | ((uintptr_t)C) | ((uintptr_t)bias) | ((uintptr_t)addend)) & 15) == 0);
  SEA_CHECK_ARG(!addend || (ld_addend >= N && (ld_addend % 4) == 0));
  const int64_t total = (int64_t)M * (N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)partial,
                     splits, total, M, N / 4, (const f32x4*)bias, relu, C, ldc, out_amax, addend, ld_addend);
  SEA_RETURN_LAST();
}

extern "C" int sea_absmax_bits(const float* A, int64_t lda, int M, int K, int batch, int64_t strideA, int rows_per_word,
                               uint32_t* out_bits, void* stream) {
  SEA_CHECK_ARG(A && out_bits && M > 0 && K > 0 && (K % 4) == 0 && (lda % 4) == 0 && batch > 0 && (((uintptr_t)A) & 15) == 0);
  SEA_CHECK_ARG(rows_per_word >= 0);
  if (rows_per_word == 1 && batch == 1) {
    // one word per row: the scales of a GRADIENT operand (sea_gemm_split_f16 with amax_rows = 1)
    const int K4 = K / 4;
    const int lpr = K4 >= 64 ? 64 : (K4 >= 32 ? 32 : (K4 >= 16 ? 16 : 8));
    const int64_t blocks = ((int64_t)M * lpr + 255) / 256;
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192)), block(256);
    if (lpr == 64)
      hipLaunchKernelGGL(rowmax_rows_kernel<64>, grid, block, 0, (hipStream_t)stream, A, lda, M, K4, out_bits);
    else if (lpr == 32)
      hipLaunchKernelGGL(rowmax_rows_kernel<32>, grid, block, 0, (hipStream_t)stream, A, lda, M, K4, out_bits);
    else if (lpr == 16)
      hipLaunchKernelGGL(rowmax_rows_kernel<16>, grid, block, 0, (hipStream_t)stream, A, lda, M, K4, out_bits);
    else
      hipLaunchKernelGGL(rowmax_rows_kernel<8>, grid, block, 0, (hipStream_t)stream, A, lda, M, K4, out_bits);
    SEA_RETURN_LAST();
  }
  if (rows_per_word > 0 && rows_per_word < M) {
    // one word per group of rows_per_word consecutive rows (all batch entries): grid.y = group
    const int words = (M + rows_per_word - 1) / rows_per_word;
    SEA_CHECK_ARG(words <= 65535);
    hipLaunchKernelGGL(zero_words_kernel, dim3((words + 255) / 256), dim3(256), 0, (hipStream_t)stream, out_bits, words);
    const int64_t per = (int64_t)rows_per_word * (K / 4);
    int gx = (int)((per + 1023) / 1024);
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(absmax_groups_kernel, dim3(gx, words), dim3(256), 0, (hipStream_t)stream, A, lda, M, K / 4, strideA, batch,
                       rows_per_word, out_bits);
    SEA_RETURN_LAST();
  }
  hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_bits, 1);
  const int64_t total = (int64_t)batch * M * (K / 4);
  int grid = grid_for(total, 512);   // two float4 per lane and trip
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(absmax_bits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, lda, M, K / 4, strideA, batch, out_bits);
  SEA_RETURN_LAST();
}

static int gemm_split_impl(const float* A, int64_t lda, const void* Wp, float* C, int64_t ldc, const float* bias, int relu, int M,
                           int N, int K, int terms, int batch, int64_t strideA, int64_t strideW_bytes, int64_t strideC,
                           const uint32_t* amax_bits, int amax_rows, uint32_t* out_amax, void* stream,
                           const SeaGemmEpilogue* epi) {
  SEA_CHECK_ARG(A && Wp && C && M > 0 && N > 0 && K > 0 && (K % GS_BK) == 0 && batch > 0);
  SEA_CHECK_ARG(lda >= K && ldc >= N && (lda % 4) == 0 && (int64_t)M * lda < (1ll << 30));  // 32-bit lane offsets into A
  SEA_CHECK_ARG(((((uintptr_t)A) | ((uintptr_t)Wp)) & 15) == 0 && (((uintptr_t)C) & 3) == 0 && (strideA % 4) == 0 &&
                (strideW_bytes % 16) == 0);
  GemmSplitArgs p = {};
  p.A = A;
  p.W = (const char*)Wp;
  p.C = C;
  p.bias = bias;
  p.lda = lda;
  p.ldc = ldc;
  p.strideA = strideA;
  p.strideW = strideW_bytes;
  p.strideC = strideC;
  p.M = M;
  p.N = N;
  p.K = K;
  p.Npad = gs_npad(N);
  p.mblocks = (M + GS_BM - 1) / GS_BM;
  p.nblocks = p.Npad / GS_BN;
  const int64_t total = (int64_t)p.mblocks * p.nblocks * batch;
  SEA_CHECK_ARG(total < (1ll << 30));
  p.total = (int)total;
  p.per_xcd = (p.total + 7) / 8;
  p.relu = relu;
  p.amax_bits = amax_bits;
  p.amax_rows = amax_rows;
  p.out_amax = out_amax;
  p.addend = epi ? epi->addend : nullptr;
  p.ld_add = epi ? epi->ld_addend : 0;
  p.stride_add = epi ? epi->stride_addend : 0;
  p.gelu_out = epi ? epi->gelu_out : nullptr;
  p.gelu_grad_of = epi ? epi->gelu_grad_of : nullptr;
  p.a_gelu_grad_of = epi ? epi->a_gelu_grad_of : nullptr;
  p.a_gelu = epi ? epi->a_gelu : 0;
  p.a_gate = epi ? epi->a_gate : 0;
  p.amax_mul = (epi && epi->a_amax_mul > 0.f) ? epi->a_amax_mul : 1.f;
  p.amax_mul_dev = epi ? epi->a_amax_mul_dev : nullptr;
  p.w_inv = terms == 22 ? (const float*)((const char*)Wp + (int64_t)(K / GS_BK) * 2 * gs_npad(N) * GS_BK * 2) : nullptr;
  const dim3 grid(p.per_xcd * 8), block(256);
  // MFMA shape: 32x32x16 fragments (default) or 16x16x32 (SEA_GEMM_SHAPE=16 / sea_gemm_split_mfma_shape(16)): the chip holds a
  // higher clock on the small shape in MFMA-dense loops (MI355X guide, DVFS give-back item 7); which one wins is measured
  // IN the attack loop, where the clock is the limiter (profiles/r4_rejected_experiments.md: +3.6 % per step on the small shape)
  const bool shape16 = g_mfma_shape.load(std::memory_order_relaxed) == 16;
  const bool fused = p.addend || p.gelu_out || p.gelu_grad_of;
  SEA_CHECK_ARG(!(p.a_gelu && (fused || p.a_gelu_grad_of)) && (!p.a_gelu_grad_of || !fused));
  SEA_CHECK_ARG(!p.a_gelu_grad_of || ((terms == 1 || terms == 2 || terms == 22) && (((uintptr_t)p.a_gelu_grad_of) & 15) == 0));
  // prologue: 0 none, 1 A * GELU'(t), 2 GELU(A), 3 ReLU gate
  const int pro = p.a_gelu ? 2 : (p.a_gelu_grad_of ? (p.a_gate ? 3 : 1) : 0);
  const hipStream_t st = (hipStream_t)stream;
  const int pipe = g_pipeline.load(std::memory_order_relaxed);
  // One-block-per-CU kernels (gemm_split_big.hip) for the products that are bound by the bytes a CU pulls through its L1.
  // pipe 3 forces them wherever they apply; pipe 2 takes 256 x 256 tiles from 1024 tiles on (the Winograd-domain products) and
  // 128 x 384 tiles where they fill the chip in whole rounds (256 CUs: the 32 x 32-pixel stage's M = 8192 products).
  if (!shape16 && (terms == 22 || terms == 2) && !fused && (pipe == 2 || pipe == 3)) {
    const int64_t t256 = (N % 256) == 0 ? (int64_t)((M + 255) / 256) * (N / 256) * batch : 0;
    const int64_t t384 = (N % 384) == 0 ? (int64_t)((M + 127) / 128) * (N / 384) * batch : 0;
    if (pro == 0 && M >= 256 && t256 > 0 && (pipe == 3 ? t384 == 0 : t256 >= 1024) && gemm_split_big_launch(p, terms, batch, 0, 0, st))
      SEA_RETURN_LAST();
    // (one round of one block per CU: measured in the loop -- 39 / 48 / 45 us against 41 / 61 / 50 us for the plain / GELU' /
    // GELU launches of the M = 8192 products; on two or more rounds the 128 x 128 kernels win, profiles/r5_gemm_wide_ab.md)
    const bool wide_fits = t384 >= g_wide_min && t384 <= 256 && K >= 192;
    if (t384 > 0 && M >= 128 && (pipe == 3 || wide_fits) && gemm_split_big_launch(p, terms, batch, 1, pro, st)) SEA_RETURN_LAST();
  }
  // per launch (pipe 2): the ping-pong kernel where it wins IN the attack loop (profiles/r5_gemm_pipe_ab.md): a VALU-heavy
  // prologue (GELU / GELU' / gate on A: hidden in its MFMA shadow) on a grid that fills two blocks per CU at least as
  // well as three (768 tiles are one round of three blocks per CU but one and a half of two)
  const int64_t r2 = (total + 511) / 512 * 512, r3 = (total + 767) / 768 * 768;
  const bool pingpong = pipe == 1 || (pipe == 2 && pro != 0 && r2 <= r3);
  if (!shape16 && terms != 3 && pingpong && gemm_split_pp_launch(p, terms, pro, fused, st)) SEA_RETURN_LAST();
#define SEA_GS_LAUNCH(T, S16, F16, EPI, PRO) hipLaunchKernelGGL((gemm_split_kernel<T, S16, F16, EPI, PRO>), grid, block, 0, st, p)
#define SEA_GS_PRO(T, S16, F16)                                   \
  do {                                                            \
    if (pro == 2) SEA_GS_LAUNCH(T, S16, F16, false, 2);           \
    else if (pro == 1) SEA_GS_LAUNCH(T, S16, F16, false, 1);      \
    else if (pro == 3) SEA_GS_LAUNCH(T, S16, F16, false, 3);      \
    else if (fused) SEA_GS_LAUNCH(T, S16, F16, true, 0);          \
    else SEA_GS_LAUNCH(T, S16, F16, false, 0);                    \
  } while (0)
  if (terms == 22) {
    if (shape16) SEA_GS_PRO(2, true, true); else SEA_GS_PRO(2, false, true);
  } else if (terms == 3) {
    SEA_CHECK_ARG(pro == 0 || pro == 2);
    if (shape16) {
      if (pro == 2) SEA_GS_LAUNCH(3, true, false, false, 2); else if (fused) SEA_GS_LAUNCH(3, true, false, true, 0); else SEA_GS_LAUNCH(3, true, false, false, 0);
    } else {
      if (pro == 2) SEA_GS_LAUNCH(3, false, false, false, 2); else if (fused) SEA_GS_LAUNCH(3, false, false, true, 0); else SEA_GS_LAUNCH(3, false, false, false, 0);
    }
  } else if (terms == 1) {   // one bf16 term: the operands of a bf16-autocast GEMM, fp32 in / out (BASELINE configs[3])
    if (shape16) SEA_GS_PRO(1, true, false); else SEA_GS_PRO(1, false, false);
  } else {
    if (shape16) SEA_GS_PRO(2, true, false); else SEA_GS_PRO(2, false, false);
  }
#undef SEA_GS_PRO
#undef SEA_GS_LAUNCH
  SEA_RETURN_LAST();
}

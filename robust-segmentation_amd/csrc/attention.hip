// M7: fp32 multi-head attention on the matrix cores (Segmenter ViT-S/16 and its mask-transformer decoder:
// reference semseg/models/backbones/vit_encoder.py:106-127, explicit softmax(Q K^T * scale) V with fp32 operands).
//
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; exact fp32 products, 64 FLOP/clk/SIMD = the fp32 peak of the chip).
// Flash formulation, no N x N matrix in memory.  Three kernels of the same shape:
//   attn_fwd   wave = 32 query rows, loop over 64-key tiles:   S^T = K Q^T,  online softmax,        O^T += V^T P^T
//   attn_dq    wave = 32 query rows, loop over 64-key tiles:   S^T, dP^T = V dO^T, dS^T,             dQ^T += K^T dS^T
//   attn_dkv   wave = 32 key rows,   loop over 64-query tiles: S, dP = dO V^T, dS,   dV^T += dO^T P, dK^T += Q^T dS
// (the backward recomputes S in both of its kernels: 7 matrix products instead of 5, but no atomics and no partial
// buffers: results are bitwise reproducible and every output element has exactly one writer).
//
// Register-resident operand trick (MI355X guide, "an accumulator tile as the next MFMA's operand"): the first product
// of every kernel is computed TRANSPOSED so that the index the softmax reduces over lies on the accumulator
// registers and the row that owns the statistics lies on the lane.  Then (a) row max / row sum are in-lane loops plus
// one cross-half shuffle, (b) rescaling the output accumulators is a per-lane multiply, and (c) the probability tile
// is already laid out as the B operand of the second product: accumulator register t of lane-half h holds row
// (t&3) + 8*(t>>2) + 4*h, which is exactly the k index pair {h=0, h=1} MFMA step t consumes -- no LDS round trip.
//
// Operand maps of v_mfma_f32_32x32x2_f32: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
// C/D[i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][j = lane&31].  The head dimension (64) is walked as d = 32*h + s for
// step s of lane-half h, so that a lane's 32 operand values are contiguous in memory / LDS (8 x 16-byte reads).
#include "sea_common.h"

namespace sea {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 64;        // head dimension (ViT-S/16: 384 / 6; mask transformer: 384 / 6)
constexpr int kTile = 64;     // keys (or queries) per LDS tile
constexpr int kRow = kD + 4;  // LDS row stride in floats: 16-byte reads of 16 different rows hit 16 different bank slots
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// v_exp_f32 itself (2^x, flushes results below 2^-126 to zero: irrelevant next to a soft-max sum >= 1)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// 64 rows x 64 floats of a (row-strided) matrix -> LDS tile [64][kRow]; rows >= n_valid are clamped to the last valid
// row (their results are masked or never stored).  256 threads, 4 float4 per thread.
__device__ __forceinline__ void load_tile_regs(const float* __restrict__ base, int64_t row_stride, int row0, int n_rows,
                                               f32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (int)threadIdx.x + 256 * i;  // float4 index in the tile
    int row = row0 + (e >> 4);
    row = row < n_rows ? row : n_rows - 1;
    r[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)row * row_stride + 4 * (e & 15));
  }
}
__device__ __forceinline__ void store_tile_lds(float* __restrict__ tile, const f32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (int)threadIdx.x + 256 * i;
    *reinterpret_cast<f32x4*>(tile + (e >> 4) * kRow + 4 * (e & 15)) = r[i];
  }
}

// 32 contiguous floats of row `row`, columns [32*half, 32*half + 32), times `mul`
__device__ __forceinline__ void load_row_half(const float* __restrict__ base, int64_t row_stride, int row, int half,
                                              float mul, float (&out)[32]) {
  const float* p = base + (int64_t)row * row_stride + 32 * half;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * i + k] = v[k] * mul;
  }
}

// acc[i = tile row 32*rb + (lane&31) ... ] : X^T-style product  acc = Tile(32 rows of LDS) . Reg^T
//   acc[row r of the LDS sub-block][column = lane&31]  +=  sum_d tile[32*rb + r][d] * reg_of_lane_column[d]
__device__ __forceinline__ f32x16 tile_times_regs(const float* __restrict__ tile, int rb, int lane, const float (&reg)[32],
                                                  f32x16 acc) {
  const float* p = tile + (32 * rb + (lane & 31)) * kRow + 32 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], reg[4 * i + k], acc, 0, 0, 0);
  }
  return acc;
}

// out^T[d = 32*dt + (lane&31)][column = lane&31 of X]  +=  sum_r tile[32*rb + r][d] * X[r][column]
// X = accumulator tile whose rows r lie on the registers (register t of half h = row acc_row(t, h)).
__device__ __forceinline__ f32x16 tile_t_times_acc(const float* __restrict__ tile, int rb, int dt, int lane, const f32x16& x,
                                                   f32x16 acc) {
  const int half = lane >> 5;
  const float* p = tile + (32 * rb + 4 * half) * kRow + 32 * dt + (lane & 31);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float a = p[((t & 3) + 8 * (t >> 2)) * kRow];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[t], acc, 0, 0, 0);
  }
  return acc;
}

// store a transposed accumulator pair (d-tile 0 and 1) of 32 rows: out[row0 + (lane&31)][d], 16-byte stores
__device__ __forceinline__ void store_rows_t(float* __restrict__ base, int64_t row_stride, int row, bool ok, int lane,
                                             const f32x16& t0, const f32x16& t1, float mul) {
  if (!ok) return;
  const int half = lane >> 5;
  float* p = base + (int64_t)row * row_stride;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 a, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = t0[4 * g + k] * mul;
      b[k] = t1[4 * g + k] * mul;
    }
    *reinterpret_cast<f32x4*>(p + 8 * g + 4 * half) = a;
    *reinterpret_cast<f32x4*>(p + 32 + 8 * g + 4 * half) = b;
  }
}

struct AttnPtrs {
  const float* q;  // element (b, h, t, d) at q + b*sb + h*sh + t*st + d   (same strides for k and v)
  const float* k;
  const float* v;
  int64_t sb, sh, st;
};

// ---- forward ---------------------------------------------------------------------------------------------------------
// grid (ceil(T / 128), H, B), block 256: wave w owns query rows [128*bx + 32*w, +32).
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnPtrs p, int T, int H, float scale, float* __restrict__ out,
                                                          float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) float ks[kTile * kRow];
  __shared__ __attribute__((aligned(16))) float vs[kTile * kRow];
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q_row = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_row < T;
  const bool wave_rows = blockIdx.x * 128 + wave * 32 < T;  // wave-uniform
  const float* qb = p.q + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* kb = p.k + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* vb = p.v + (int64_t)b * p.sb + (int64_t)h * p.sh;

  float qreg[32];  // Q[q_row][32*half + s] * scale * log2(e): scores come out in the log2 domain
  load_row_half(qb, p.st, q_ok ? q_row : T - 1, half, scale * kLog2e, qreg);

  f32x16 o0 = zero16(), o1 = zero16();  // O^T: d-tiles 0 and 1, rows d on the registers, query on the lane
  float m_run = -INFINITY, l_run = 0.f;  // running max (both halves agree), running sum (this half's keys only)

  const int n_tiles = (T + kTile - 1) / kTile;
  f32x4 kr[4], vr[4];
  load_tile_regs(kb, p.st, 0, T, kr);
  load_tile_regs(vb, p.st, 0, T, vr);
  for (int j = 0; j < n_tiles; ++j) {
    __syncthreads();  // everyone is done reading the previous tile
    store_tile_lds(ks, kr);
    store_tile_lds(vs, vr);
    __syncthreads();
    if (j + 1 < n_tiles) {  // prefetch the next tile into registers; the loads fly under the MFMAs below
      load_tile_regs(kb, p.st, (j + 1) * kTile, T, kr);
      load_tile_regs(vb, p.st, (j + 1) * kTile, T, vr);
    }
    if (!wave_rows) continue;  // this wave's 32 rows lie beyond T: it only helps staging the tiles
    // S^T (keys on the registers, query on the lane), two 32-key sub-blocks
    f32x16 s0 = tile_times_regs(ks, 0, lane, qreg, zero16());
    f32x16 s1 = tile_times_regs(ks, 1, lane, qreg, zero16());
    const int key0 = j * kTile;
    if (key0 + kTile > T) {  // last tile: keys beyond T do not exist
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (key0 + acc_row(t, half) >= T) s0[t] = -INFINITY;
        if (key0 + 32 + acc_row(t, half) >= T) s1[t] = -INFINITY;
      }
    }
    float m_t = s0[0];
#pragma unroll
    for (int t = 1; t < 16; ++t) m_t = fmaxf(m_t, s0[t]);
#pragma unroll
    for (int t = 0; t < 16; ++t) m_t = fmaxf(m_t, s1[t]);
    m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
    const float m_new = fmaxf(m_run, m_t);  // finite: every tile holds at least one existing key
    const float alpha = fast_exp2(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      s0[t] = fast_exp2(s0[t] - m_new);
      s1[t] = fast_exp2(s1[t] - m_new);
      psum += s0[t] + s1[t];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      o0[t] *= alpha;
      o1[t] *= alpha;
    }
    // O^T += V^T P^T
    o0 = tile_t_times_acc(vs, 0, 0, lane, s0, o0);
    o1 = tile_t_times_acc(vs, 0, 1, lane, s0, o1);
    o0 = tile_t_times_acc(vs, 1, 0, lane, s1, o0);
    o1 = tile_t_times_acc(vs, 1, 1, lane, s1, o1);
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  store_rows_t(out + ((int64_t)b * T) * (H * kD) + h * kD, (int64_t)H * kD, q_row, q_ok, lane, o0, o1, 1.f / l_tot);
  if (q_ok && half == 0) lse[((int64_t)b * H + h) * T + q_row] = (m_run + log2f(l_tot)) * kLn2;
}

// ---- delta[b,h,t] = sum_d dO[b,t,h,d] * O[b,t,h,d] ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ go, int T, int H,
                                                         int64_t rows, float* __restrict__ delta) {
  // one 16-lane group per (b, t, h) row of 64 floats
  const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (r >= rows) return;  // whole 16-lane groups leave together
  const int sub = threadIdx.x & 15;
  const f32x4 a = *reinterpret_cast<const f32x4*>(o + r * kD + 4 * sub);
  const f32x4 g = *reinterpret_cast<const f32x4*>(go + r * kD + 4 * sub);
  float s = a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
  if (sub == 0) {
    const int64_t bt = r / H;
    const int hh = (int)(r - bt * H);
    const int64_t bb = bt / T;
    const int t = (int)(bt - bb * T);
    delta[(bb * H + hh) * T + t] = s;
  }
}

// ---- dQ ----------------------------------------------------------------------------------------------------------------
// grid (ceil(T / 128), H, B), block 256: wave w owns query rows like attn_fwd.  go: (B, T, H*64) gradient of the output;
// dq: written with the strides of q (gsb, gsh, gst) -- i.e. straight into the gradient of the packed qkv tensor.
__global__ __launch_bounds__(256, 2) void attn_dq_kernel(AttnPtrs p, int T, int H, float scale, const float* __restrict__ go,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         float* __restrict__ dq, int64_t gsb, int64_t gsh, int64_t gst) {
  __shared__ __attribute__((aligned(16))) float ks[kTile * kRow];
  __shared__ __attribute__((aligned(16))) float vs[kTile * kRow];
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q_row = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_row < T;
  const int q_ld = q_ok ? q_row : T - 1;
  const float* qb = p.q + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* kb = p.k + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* vb = p.v + (int64_t)b * p.sb + (int64_t)h * p.sh;
  float qreg[32], greg[32];
  load_row_half(qb, p.st, q_ld, half, scale * kLog2e, qreg);
  load_row_half(go + ((int64_t)b * T) * (H * kD) + h * kD, (int64_t)H * kD, q_ld, half, 1.f, greg);
  const float lse2 = lse[((int64_t)b * H + h) * T + q_ld] * kLog2e;
  const float dlt = delta[((int64_t)b * H + h) * T + q_ld];
  f32x16 dq0 = zero16(), dq1 = zero16();
  const int n_tiles = (T + kTile - 1) / kTile;
  f32x4 kr[4], vr[4];
  load_tile_regs(kb, p.st, 0, T, kr);
  load_tile_regs(vb, p.st, 0, T, vr);
  for (int j = 0; j < n_tiles; ++j) {
    __syncthreads();
    store_tile_lds(ks, kr);
    store_tile_lds(vs, vr);
    __syncthreads();
    if (j + 1 < n_tiles) {
      load_tile_regs(kb, p.st, (j + 1) * kTile, T, kr);
      load_tile_regs(vb, p.st, (j + 1) * kTile, T, vr);
    }
    if (blockIdx.x * 128 + wave * 32 >= T) continue;  // wave-uniform: rows beyond T, the wave only stages tiles
    const int key0 = j * kTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 s = tile_times_regs(ks, rb, lane, qreg, zero16());    // S^T  (log2 domain)
      f32x16 dp = tile_times_regs(vs, rb, lane, greg, zero16());   // dP^T = V dO^T
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool exists = key0 + 32 * rb + acc_row(t, half) < T;
        const float pr = exists ? fast_exp2(s[t] - lse2) : 0.f;
        s[t] = pr * (dp[t] - dlt) * scale;                         // dS^T
      }
      dq0 = tile_t_times_acc(ks, rb, 0, lane, s, dq0);             // dQ^T += K^T dS^T
      dq1 = tile_t_times_acc(ks, rb, 1, lane, s, dq1);
    }
  }
  store_rows_t(dq + (int64_t)b * gsb + (int64_t)h * gsh, gst, q_row, q_ok, lane, dq0, dq1, 1.f);
}

// ---- dK, dV ------------------------------------------------------------------------------------------------------------
// grid (ceil(T / 128), H, B), block 256: wave w owns KEY rows [128*bx + 32*w, +32); loop over 64-query tiles.
__global__ __launch_bounds__(256, 2) void attn_dkv_kernel(AttnPtrs p, int T, int H, float scale, const float* __restrict__ go,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dk, float* __restrict__ dv, int64_t gsb,
                                                          int64_t gsh, int64_t gst) {
  __shared__ __attribute__((aligned(16))) float qs[kTile * kRow];
  __shared__ __attribute__((aligned(16))) float gs[kTile * kRow];
  __shared__ float lse_s[kTile], dlt_s[kTile];
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int k_row = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool k_ok = k_row < T;
  const int k_ld = k_ok ? k_row : T - 1;
  const float* qb = p.q + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* kb = p.k + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* vb = p.v + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* gb = go + ((int64_t)b * T) * (H * kD) + h * kD;
  const int64_t gst_o = (int64_t)H * kD;
  float kreg[32], vreg[32];
  load_row_half(kb, p.st, k_ld, half, scale * kLog2e, kreg);   // S = Q K^T in the log2 domain
  load_row_half(vb, p.st, k_ld, half, 1.f, vreg);
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const int n_tiles = (T + kTile - 1) / kTile;
  f32x4 qr[4], gr[4];
  load_tile_regs(qb, p.st, 0, T, qr);
  load_tile_regs(gb, gst_o, 0, T, gr);
  for (int j = 0; j < n_tiles; ++j) {
    __syncthreads();
    store_tile_lds(qs, qr);
    store_tile_lds(gs, gr);
    if (threadIdx.x < kTile) {
      const int qq = j * kTile + (int)threadIdx.x;
      const int ql = qq < T ? qq : T - 1;
      lse_s[threadIdx.x] = lse[((int64_t)b * H + h) * T + ql] * kLog2e;
      dlt_s[threadIdx.x] = delta[((int64_t)b * H + h) * T + ql];
    }
    __syncthreads();
    if (j + 1 < n_tiles) {
      load_tile_regs(qb, p.st, (j + 1) * kTile, T, qr);
      load_tile_regs(gb, gst_o, (j + 1) * kTile, T, gr);
    }
    if (blockIdx.x * 128 + wave * 32 >= T) continue;  // wave-uniform: key rows beyond T
    const int q0 = j * kTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 s = tile_times_regs(qs, rb, lane, kreg, zero16());    // S[q on registers][key on the lane]
      f32x16 dp = tile_times_regs(gs, rb, lane, vreg, zero16());   // dP = dO V^T
      f32x16 pr;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int ql = 32 * rb + acc_row(t, half);
        const bool exists = (q0 + ql < T) && k_ok;
        pr[t] = exists ? fast_exp2(s[t] - lse_s[ql]) : 0.f;
        s[t] = pr[t] * (dp[t] - dlt_s[ql]) * scale;                 // dS
      }
      dv0 = tile_t_times_acc(gs, rb, 0, lane, pr, dv0);            // dV^T += dO^T P
      dv1 = tile_t_times_acc(gs, rb, 1, lane, pr, dv1);
      dk0 = tile_t_times_acc(qs, rb, 0, lane, s, dk0);             // dK^T += Q^T dS
      dk1 = tile_t_times_acc(qs, rb, 1, lane, s, dk1);
    }
  }
  store_rows_t(dk + (int64_t)b * gsb + (int64_t)h * gsh, gst, k_row, k_ok, lane, dk0, dk1, 1.f);
  store_rows_t(dv + (int64_t)b * gsb + (int64_t)h * gsh, gst, k_row, k_ok, lane, dv0, dv1, 1.f);
}

}  // namespace sea

using namespace sea;

// M7b (csrc/attention_bf16.hip): the same three kernels on the bf16 matrix cores by operand splitting.
int sea_attention_fwd_bf16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                           float scale, float* out, float* lse, int terms, hipStream_t stream);
int sea_attention_bwd_bf16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                           float scale, const float* grad_out, const float* lse, const float* delta, float* dq, float* dk,
                           float* dv, int64_t gsb, int64_t gsh, int64_t gst, int terms, hipStream_t stream);

int sea_attention_bwd_f16x2(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                            float scale, const float* grad_out, const float* lse, const float* delta, uint32_t* amax_ws, float* dq,
                            float* dk, float* dv, int64_t gsb, int64_t gsh, int64_t gst, hipStream_t stream);

int sea_attention_fwd_f16x2(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                            float scale, uint32_t* amax_ws, float* out, float* lse, hipStream_t stream);

// the forward with fp16 x 2 operands (22 significant bits, three MFMA products per pair); amax_ws: 4 B H words of scratch
extern "C" int sea_attention_fwd_f16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                                     int T, int D, float scale, uint32_t* amax_ws, float* out, float* lse, void* stream) {
  SEA_CHECK_ARG(q && k && v && out && lse && amax_ws && B > 0 && H > 0 && T > 0 && D == kD);
  SEA_CHECK_ARG((sb % 4) == 0 && (sh % 4) == 0 && (st % 4) == 0 &&
                ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0);
  return sea_attention_fwd_f16x2(q, k, v, sb, sh, st, B, H, T, scale, amax_ws, out, lse, (hipStream_t)stream);
}

// SEA_ATTN_TERMS (forward) / SEA_ATTN_TERMS_BWD: 3 or 2 = bf16 terms per operand on v_mfma_f32_32x32x16_bf16, 0 = the fp32
// MFMA kernels of this file.  Defaults: 3 (= the fp32 operands exactly) forward AND backward (round 4: the evaluation is
// fp32-equivalent end to end; 2 backward was round 3's default: the attack consumes only the sign of the input gradient).
static inline int attn_terms(bool backward) {   // looked up per call (two launches per layer): tests switch it in-process
  const char* e = getenv(backward ? "SEA_ATTN_TERMS_BWD" : "SEA_ATTN_TERMS");
  const int t = e ? atoi(e) : 3;
  return (t == 2 || t == 3) ? t : 0;
}

static int attention_fwd_impl(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                              int T, int D, float scale, float* out, float* lse, int terms, void* stream) {
  SEA_CHECK_ARG(q && k && v && out && lse && B > 0 && H > 0 && T > 0 && D == kD);
  SEA_CHECK_ARG((sb % 4) == 0 && (sh % 4) == 0 && (st % 4) == 0 &&
                ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0);
  if (terms)
    return sea_attention_fwd_bf16(q, k, v, sb, sh, st, B, H, T, scale, out, lse, terms, (hipStream_t)stream);
  AttnPtrs p{q, k, v, sb, sh, st};
  dim3 grid((T + 127) / 128, H, B), block(256);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, (hipStream_t)stream, p, T, H, scale, out, lse);
  SEA_RETURN_LAST();
}

// q/k/v: element (b,h,t,d) at ptr + b*sb + h*sh + t*st + d (floats), d contiguous, head dim 64, 16-byte aligned rows.
extern "C" int sea_attention_fwd(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                                 int T, int D, float scale, float* out, float* lse, void* stream) {
  return attention_fwd_impl(q, k, v, sb, sh, st, B, H, T, D, scale, out, lse, attn_terms(false), stream);
}

// same, with the number of bf16 terms of the products chosen by the caller (3, 2, or 0 = fp32 MFMA kernels)
extern "C" int sea_attention_fwd_terms(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B,
                                       int H, int T, int D, float scale, float* out, float* lse, int terms, void* stream) {
  SEA_CHECK_ARG(terms == 0 || terms == 2 || terms == 3);
  return attention_fwd_impl(q, k, v, sb, sh, st, B, H, T, D, scale, out, lse, terms, stream);
}

static int attention_bwd_impl(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                              int T, int D, float scale, const float* out, const float* grad_out, const float* lse,
                              float* delta, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh, int64_t gst,
                              int bwd_terms, void* stream, uint32_t* amax_ws = nullptr);

extern "C" int sea_attention_bwd(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                                 int T, int D, float scale, const float* out, const float* grad_out, const float* lse,
                                 float* delta, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh, int64_t gst,
                                 void* stream) {
  return attention_bwd_impl(q, k, v, sb, sh, st, B, H, T, D, scale, out, grad_out, lse, delta, dq, dk, dv, gsb, gsh, gst,
                            attn_terms(true), stream);
}

// same, with the number of bf16 terms of the backward products chosen by the caller (3, 2, or 0 = fp32 MFMA kernels):
// a caller that ALSO trains the weights through this backward wants 3; an attack that consumes sign(dx) takes 2
extern "C" int sea_attention_bwd_terms(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B,
                                       int H, int T, int D, float scale, const float* out, const float* grad_out,
                                       const float* lse, float* delta, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh,
                                       int64_t gst, int terms, void* stream) {
  SEA_CHECK_ARG(terms == 0 || terms == 2 || terms == 3);
  return attention_bwd_impl(q, k, v, sb, sh, st, B, H, T, D, scale, out, grad_out, lse, delta, dq, dk, dv, gsb, gsh, gst, terms,
                            stream);
}

// fp16 x 2 backward (22 significant bits per operand in three MFMA products per pair, csrc/attention_bf16.hip): the accuracy of
// the three-term bf16 mode at the cost of the two-term one.  amax_ws: 4 B H device words of scratch.
extern "C" int sea_attention_bwd_f16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                                     int T, int D, float scale, const float* out, const float* grad_out, const float* lse,
                                     float* delta, uint32_t* amax_ws, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh,
                                     int64_t gst, void* stream) {
  SEA_CHECK_ARG(amax_ws != nullptr);
  return attention_bwd_impl(q, k, v, sb, sh, st, B, H, T, D, scale, out, grad_out, lse, delta, dq, dk, dv, gsb, gsh, gst, 22,
                            stream, amax_ws);
}

static int attention_bwd_impl(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H,
                              int T, int D, float scale, const float* out, const float* grad_out, const float* lse,
                              float* delta, float* dq, float* dk, float* dv, int64_t gsb, int64_t gsh, int64_t gst,
                              int bwd_terms, void* stream, uint32_t* amax_ws) {
  SEA_CHECK_ARG(q && k && v && out && grad_out && lse && delta && dq && dk && dv && B > 0 && H > 0 && T > 0 && D == kD);
  SEA_CHECK_ARG((sb % 4) == 0 && (sh % 4) == 0 && (st % 4) == 0 && (gsb % 4) == 0 && (gsh % 4) == 0 && (gst % 4) == 0);
  SEA_CHECK_ARG(((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out) | ((uintptr_t)grad_out) |
                  ((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 15) == 0);
  AttnPtrs p{q, k, v, sb, sh, st};
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * T * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, out, grad_out, T, H, rows, delta);
  if (bwd_terms == 22)
    return sea_attention_bwd_f16x2(q, k, v, sb, sh, st, B, H, T, scale, grad_out, lse, delta, amax_ws, dq, dk, dv, gsb, gsh, gst, s);
  if (const int terms = bwd_terms)
    return sea_attention_bwd_bf16(q, k, v, sb, sh, st, B, H, T, scale, grad_out, lse, delta, dq, dk, dv, gsb, gsh, gst, terms, s);
  dim3 grid((T + 127) / 128, H, B), block(256);
  hipLaunchKernelGGL(attn_dq_kernel, grid, block, 0, s, p, T, H, scale, grad_out, lse, delta, dq, gsb, gsh, gst);
  hipLaunchKernelGGL(attn_dkv_kernel, grid, block, 0, s, p, T, H, scale, grad_out, lse, delta, dk, dv, gsb, gsh, gst);
  SEA_RETURN_LAST();
}

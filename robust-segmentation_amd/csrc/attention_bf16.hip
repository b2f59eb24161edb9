// M7b: the attention of M7 (csrc/attention.hip) on the bf16 matrix cores by operand splitting (see csrc/gemm_split.hip:
// an fp32 number is the exact sum of three bf16 numbers; six v_mfma_f32_32x32x16_bf16 products accumulated in fp32 give
// fp32-level accuracy, three products with two terms give 16 significant bits per operand).
//
// Same three kernels, same flash formulation, same "first product transposed" register trick and the same outputs as
// the fp32 kernels (reference semseg/models/backbones/vit_encoder.py:106-127), so the launch geometry, masking,
// online soft-max, log-sum-exp and the stores are shared in spirit; what changes is how a 64 x 64 tile is staged and
// multiplied.  The fp32 MFMA (v_mfma_f32_32x32x2_f32) needs 64 cycles per 2048 MACs; six bf16 products of the same
// 32x32 block over K = 16 need 6 x 32 cycles per 16384 MACs: 2.7 x fewer matrix-core cycles with three terms (forward),
// 5.3 x fewer with two (backward: only the sign of the input gradient is consumed by the attack, attacker.py:396).
//
// Two tile images in LDS, per bf16 term (64 rows x 128 B each):
//   RM  [row][64 d]  (d contiguous)   A operand of   acc[row][lane] += sum_d  tile[row][d] * reg_of_lane[d]
//   TR  [d][64 rows] (rows contiguous) A operand of  out[d][lane]  += sum_row tile[row][d] * X[row][lane],  X = an
//       accumulator tile (rows on the registers): its registers 8u..8u+7 ARE the B fragment of k step u, with the k order
//       k(j, half) = 16u + 8 (j >> 2) + 4 half + (j & 3) (MI355X guide, "an accumulator tile as the next MFMA's operand"),
//       so the A fragment reads the same rows: two 8-byte pieces of the TR image.
// Swizzles (conflict-free for the lane groups of ds_read_b128 / ds_read_b64): RM 16-byte chunk ^ ((row >> 1) & 7),
// TR 8-byte slot ^ ((d >> 1) & 15).  The head dimension is walked as d = 32 half + 8 s + j (a lane's operands contiguous).
#include "sea_common.h"

namespace sea {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 64, kTile = 64;
constexpr int kImg = 64 * 128;  // bytes of one term image
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // low half = a
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 8 floats -> TERMS fragments of 8 bf16 (element j of the fragment = v[j])
template <int TERMS>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[TERMS]) {
  float r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = v[i];
#pragma unroll
  for (int t = 0; t < TERMS; ++t) {
    u32x4 p;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = pack_bf16(r[2 * i], r[2 * i + 1]);
    out[t] = __builtin_bit_cast(bf16x8, p);
    if (t + 1 < TERMS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r[2 * i] -= __uint_as_float(p[i] << 16);
        r[2 * i + 1] -= __uint_as_float(p[i] & 0xffff0000u);
      }
    }
  }
}

template <int TERMS>
__device__ __forceinline__ f32x16 products(const bf16x8 (&a)[TERMS], const bf16x8 (&b)[TERMS], f32x16 c) {
  if constexpr (TERMS == 3) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
  }
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

// 64 rows x 64 floats of a (row-strided) matrix -> registers; thread e = tid + 256 i holds row e >> 4, floats 4 (e & 15)..+3
__device__ __forceinline__ void load_tile_regs(const float* __restrict__ base, int64_t row_stride, int row0, int n_rows,
                                               f32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (int)threadIdx.x + 256 * i;
    int row = row0 + (e >> 4);
    row = row < n_rows ? row : n_rows - 1;
    r[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)row * row_stride + 4 * (e & 15));
  }
}

// registers of load_tile_regs -> the bf16 term images (RM and / or TR) of the tile
template <int TERMS, bool RM, bool TR>
__device__ __forceinline__ void stage_tile(char* __restrict__ rm, char* __restrict__ tr, const f32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (int)threadIdx.x + 256 * i;
    const int row = e >> 4, d4 = e & 15;
    float x0 = r[i][0], x1 = r[i][1], x2 = r[i][2], x3 = r[i][3];
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      const uint32_t p0 = pack_bf16(x0, x1), p1 = pack_bf16(x2, x3);
      if (RM) *reinterpret_cast<u32x2*>(rm + t * kImg + row * 128 + (((d4 >> 1) ^ ((row >> 1) & 7)) << 4) + (d4 & 1) * 8) = u32x2{p0, p1};
      if (TR) {
        // element c of the float4 is head-dim index d = 4 d4 + c: row d of the TR image, 2 bytes at tile row `row`
        const int d = 4 * d4;
        char* q = tr + t * kImg + (row & 3) * 2;
        const int slot = row >> 2;
        *reinterpret_cast<uint16_t*>(q + (d + 0) * 128 + ((slot ^ (((d + 0) >> 1) & 15)) << 3)) = (uint16_t)(p0 & 0xffffu);
        *reinterpret_cast<uint16_t*>(q + (d + 1) * 128 + ((slot ^ (((d + 1) >> 1) & 15)) << 3)) = (uint16_t)(p0 >> 16);
        *reinterpret_cast<uint16_t*>(q + (d + 2) * 128 + ((slot ^ (((d + 2) >> 1) & 15)) << 3)) = (uint16_t)(p1 & 0xffffu);
        *reinterpret_cast<uint16_t*>(q + (d + 3) * 128 + ((slot ^ (((d + 3) >> 1) & 15)) << 3)) = (uint16_t)(p1 >> 16);
      }
      if (t + 1 < TERMS) {
        x0 -= __uint_as_float(p0 << 16);
        x1 -= __uint_as_float(p0 & 0xffff0000u);
        x2 -= __uint_as_float(p1 << 16);
        x3 -= __uint_as_float(p1 & 0xffff0000u);
      }
    }
  }
}

// a lane's 32 head-dim values (d = 32 half + i) times `mul`, as the B fragments of the four 16-deep k steps
template <int TERMS>
struct RowFrag {
  bf16x8 f[4][TERMS];
};
template <int TERMS>
__device__ __forceinline__ void load_row_frag(const float* __restrict__ base, int64_t row_stride, int row, int half, float mul,
                                              RowFrag<TERMS>& out) {
  const float* p = base + (int64_t)row * row_stride + 32 * half;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + 8 * s), b = *reinterpret_cast<const f32x4*>(p + 8 * s + 4);
    const float v[8] = {a[0] * mul, a[1] * mul, a[2] * mul, a[3] * mul, b[0] * mul, b[1] * mul, b[2] * mul, b[3] * mul};
    split8<TERMS>(v, out.f[s]);
  }
}

// acc[row r of sub-block rb][column = lane & 31] += sum_d tile[32 rb + r][d] * frag_of_lane_column[d]
template <int TERMS>
__device__ __forceinline__ f32x16 rm_times_frag(const char* __restrict__ rm, int rb, int lane, const RowFrag<TERMS>& q, f32x16 acc) {
  const int row = 32 * rb + (lane & 31), half = lane >> 5;
  const char* p = rm + row * 128;
  const int sw = (row >> 1) & 7;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 a[TERMS];
#pragma unroll
    for (int t = 0; t < TERMS; ++t) a[t] = *reinterpret_cast<const bf16x8*>(p + t * kImg + (((4 * half + s) ^ sw) << 4));
    acc = products<TERMS>(a, q.f[s], acc);
  }
  return acc;
}

// out_dt[d = 32 dt + (lane & 31)][column] += sum_{r in sub-block rb} tile[32 rb + r][d] * X[r][column], dt = 0, 1
template <int TERMS>
__device__ __forceinline__ void tr_times_acc(const char* __restrict__ tr, int rb, int lane, const f32x16& x, f32x16& o0, f32x16& o1) {
  const int half = lane >> 5;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[8 * u + j];
    bf16x8 b[TERMS];
    split8<TERMS>(v, b);
    const int slot = 8 * rb + 4 * u + half;  // rows 32 rb + 16 u + 4 half .. +3; the second piece is 8 rows further (slot + 2)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int d = 32 * dt + (lane & 31);
      const char* p = tr + d * 128;
      const int sw = (d >> 1) & 15;
      bf16x8 a[TERMS];
#pragma unroll
      for (int t = 0; t < TERMS; ++t) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(p + t * kImg + ((slot ^ sw) << 3));
        const u32x2 hi = *reinterpret_cast<const u32x2*>(p + t * kImg + (((slot + 2) ^ sw) << 3));
        a[t] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
      }
      if (dt == 0)
        o0 = products<TERMS>(a, b, o0);
      else
        o1 = products<TERMS>(a, b, o1);
    }
  }
}

// store a transposed accumulator pair (d-tile 0 and 1) of 32 rows: out[row][d], 16-byte stores
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

struct AttnPtrsB {
  const float* q;
  const float* k;
  const float* v;
  int64_t sb, sh, st;
};

// ---- forward: wave = 32 query rows, loop over 64-key tiles: S^T = K Q^T, online soft-max, O^T += V^T P^T ----------------
template <int TERMS>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(AttnPtrsB p, int T, int H, float scale, float* __restrict__ out,
                                                               float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) char k_rm[TERMS * kImg];
  __shared__ __attribute__((aligned(16))) char v_tr[TERMS * kImg];
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q_row = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_row < T;
  const bool wave_rows = blockIdx.x * 128 + wave * 32 < T;
  const float* qb = p.q + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* kb = p.k + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* vb = p.v + (int64_t)b * p.sb + (int64_t)h * p.sh;

  RowFrag<TERMS> qf;  // Q[q_row][.] * scale * log2(e): scores come out in the log2 domain
  load_row_frag<TERMS>(qb, p.st, q_ok ? q_row : T - 1, half, scale * kLog2e, qf);
  f32x16 o0 = zero16(), o1 = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  const int n_tiles = (T + kTile - 1) / kTile;
  f32x4 kr[4], vr[4];
  load_tile_regs(kb, p.st, 0, T, kr);
  load_tile_regs(vb, p.st, 0, T, vr);
  for (int j = 0; j < n_tiles; ++j) {
    __syncthreads();
    stage_tile<TERMS, true, false>(k_rm, nullptr, kr);
    stage_tile<TERMS, false, true>(nullptr, v_tr, vr);
    __syncthreads();
    if (j + 1 < n_tiles) {
      load_tile_regs(kb, p.st, (j + 1) * kTile, T, kr);
      load_tile_regs(vb, p.st, (j + 1) * kTile, T, vr);
    }
    if (!wave_rows) continue;
    f32x16 s0 = rm_times_frag<TERMS>(k_rm, 0, lane, qf, zero16());
    f32x16 s1 = rm_times_frag<TERMS>(k_rm, 1, lane, qf, zero16());
    const int key0 = j * kTile;
    if (key0 + kTile > T) {
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
    const float m_new = fmaxf(m_run, m_t);
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
    tr_times_acc<TERMS>(v_tr, 0, lane, s0, o0, o1);
    tr_times_acc<TERMS>(v_tr, 1, lane, s1, o0, o1);
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  store_rows_t(out + ((int64_t)b * T) * (H * kD) + h * kD, (int64_t)H * kD, q_row, q_ok, lane, o0, o1, 1.f / l_tot);
  if (q_ok && half == 0) lse[((int64_t)b * H + h) * T + q_row] = (m_run + log2f(l_tot)) * kLn2;
}

// ---- dQ: wave = 32 query rows, loop over 64-key tiles: S^T, dP^T = V dO^T, dS^T, dQ^T += K^T dS^T -------------------------
template <int TERMS>
__global__ __launch_bounds__(256, 2) void attn_dq_bf16_kernel(AttnPtrsB p, int T, int H, float scale, const float* __restrict__ go,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              float* __restrict__ dq, int64_t gsb, int64_t gsh, int64_t gst) {
  extern __shared__ __attribute__((aligned(16))) char smem_dq[];
  char* k_rm = smem_dq;
  char* k_tr = k_rm + TERMS * kImg;
  char* v_rm = k_tr + TERMS * kImg;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q_row = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_row < T;
  const int q_ld = q_ok ? q_row : T - 1;
  const float* qb = p.q + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* kb = p.k + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* vb = p.v + (int64_t)b * p.sb + (int64_t)h * p.sh;
  RowFrag<TERMS> qf, gf;
  load_row_frag<TERMS>(qb, p.st, q_ld, half, scale * kLog2e, qf);
  load_row_frag<TERMS>(go + ((int64_t)b * T) * (H * kD) + h * kD, (int64_t)H * kD, q_ld, half, 1.f, gf);
  const float lse2 = lse[((int64_t)b * H + h) * T + q_ld] * kLog2e;
  const float dlt = delta[((int64_t)b * H + h) * T + q_ld];
  f32x16 dq0 = zero16(), dq1 = zero16();
  const int n_tiles = (T + kTile - 1) / kTile;
  f32x4 kr[4], vr[4];
  load_tile_regs(kb, p.st, 0, T, kr);
  load_tile_regs(vb, p.st, 0, T, vr);
  for (int j = 0; j < n_tiles; ++j) {
    __syncthreads();
    stage_tile<TERMS, true, true>(k_rm, k_tr, kr);
    stage_tile<TERMS, true, false>(v_rm, nullptr, vr);
    __syncthreads();
    if (j + 1 < n_tiles) {
      load_tile_regs(kb, p.st, (j + 1) * kTile, T, kr);
      load_tile_regs(vb, p.st, (j + 1) * kTile, T, vr);
    }
    if (blockIdx.x * 128 + wave * 32 >= T) continue;
    const int key0 = j * kTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 s = rm_times_frag<TERMS>(k_rm, rb, lane, qf, zero16());    // S^T (log2 domain)
      f32x16 dp = rm_times_frag<TERMS>(v_rm, rb, lane, gf, zero16());   // dP^T = V dO^T
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool exists = key0 + 32 * rb + acc_row(t, half) < T;
        const float pr = exists ? fast_exp2(s[t] - lse2) : 0.f;
        s[t] = pr * (dp[t] - dlt) * scale;                              // dS^T
      }
      tr_times_acc<TERMS>(k_tr, rb, lane, s, dq0, dq1);                 // dQ^T += K^T dS^T
    }
  }
  store_rows_t(dq + (int64_t)b * gsb + (int64_t)h * gsh, gst, q_row, q_ok, lane, dq0, dq1, 1.f);
}

// ---- dK, dV: wave = 32 KEY rows, loop over 64-query tiles: S, dP = dO V^T, dS, dV^T += dO^T P, dK^T += Q^T dS -------------
template <int TERMS>
__global__ __launch_bounds__(256, TERMS == 2 ? 2 : 1) void attn_dkv_bf16_kernel(AttnPtrsB p, int T, int H, float scale, const float* __restrict__ go,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dk, float* __restrict__ dv, int64_t gsb,
                                                               int64_t gsh, int64_t gst) {
  extern __shared__ __attribute__((aligned(16))) char smem_dkv[];
  char* q_rm = smem_dkv;
  char* q_tr = q_rm + TERMS * kImg;
  char* g_rm = q_tr + TERMS * kImg;
  char* g_tr = g_rm + TERMS * kImg;
  float* lse_s = reinterpret_cast<float*>(g_tr + TERMS * kImg);
  float* dlt_s = lse_s + kTile;
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
  RowFrag<TERMS> kf, vf;
  load_row_frag<TERMS>(kb, p.st, k_ld, half, scale * kLog2e, kf);   // S = Q K^T in the log2 domain
  load_row_frag<TERMS>(vb, p.st, k_ld, half, 1.f, vf);
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const int n_tiles = (T + kTile - 1) / kTile;
  for (int j = 0; j < n_tiles; ++j) {
    // no register prefetch here: four accumulators, two row fragments and the soft-max tiles leave no room for 32
    // staging registers across the products (they would spill); the second resident block covers the load latency
    f32x4 qr[4], gr[4];
    load_tile_regs(qb, p.st, j * kTile, T, qr);
    load_tile_regs(gb, gst_o, j * kTile, T, gr);
    __syncthreads();
    stage_tile<TERMS, true, true>(q_rm, q_tr, qr);
    stage_tile<TERMS, true, true>(g_rm, g_tr, gr);
    if (threadIdx.x < kTile) {
      const int qq = j * kTile + (int)threadIdx.x;
      const int ql = qq < T ? qq : T - 1;
      lse_s[threadIdx.x] = lse[((int64_t)b * H + h) * T + ql] * kLog2e;
      dlt_s[threadIdx.x] = delta[((int64_t)b * H + h) * T + ql];
    }
    __syncthreads();
    if (blockIdx.x * 128 + wave * 32 >= T) continue;
    const int q0 = j * kTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 pr = rm_times_frag<TERMS>(q_rm, rb, lane, kf, zero16());   // S[q on registers][key on the lane]
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int ql = 32 * rb + acc_row(t, half);
        const bool exists = (q0 + ql < T) && k_ok;
        pr[t] = exists ? fast_exp2(pr[t] - lse_s[ql]) : 0.f;            // P
      }
      tr_times_acc<TERMS>(g_tr, rb, lane, pr, dv0, dv1);                // dV^T += dO^T P
      f32x16 dp = rm_times_frag<TERMS>(g_rm, rb, lane, vf, zero16());   // dP = dO V^T
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] = pr[t] * (dp[t] - dlt_s[32 * rb + acc_row(t, half)]) * scale;   // dS (in place)
      tr_times_acc<TERMS>(q_tr, rb, lane, pr, dk0, dk1);                // dK^T += Q^T dS
    }
  }
  store_rows_t(dk + (int64_t)b * gsb + (int64_t)h * gsh, gst, k_row, k_ok, lane, dk0, dk1, 1.f);
  store_rows_t(dv + (int64_t)b * gsb + (int64_t)h * gsh, gst, k_row, k_ok, lane, dv0, dv1, 1.f);
}


// =====================================================================================================================
// fp16 x 2 backward (round 5): 22 significant bits per operand in THREE products per pair, where three bf16 terms need six.
// hi = fp16(a s), mid = fp16(a s - hi) with a power-of-two scale s per operand GROUP that puts the group's largest magnitude
// below 2^14 (csrc/gemm_split.hip); fp16's 5 exponent bits then leave 17 binades under that maximum at full 22-bit
// precision and an absolute floor of 2^-38 of it below -- wide enough for scales that are BOUNDS rather than exact maxima:
//   * row fragments (the lane's own Q / K / V / dO row, the B operand of the first products): the row's exact maximum,
//     computed in the lane; the accumulator is multiplied back by the exact inverse;
//   * staged 64 x 64 tiles (K, V in dQ; Q, dO in dK / dV): ONE scale per (image, head) from a pre-pass over the
//     four tensors (attn_amax_bh_kernel): a tile's rows are the CONTRACTION index of the transposed products, so their scale
//     must be uniform over everything an accumulator sums -- all tiles of the (image, head);
//   * the accumulator operands of the transposed products: P <= 1 takes 2^14; dS = P (dP - delta) scale takes the bound
//     2 x 64 max|dO| max|V| scale (|dP| and |delta| are both 64-term dot products of dO with V rows resp. their convex
//     combination), per query row in dQ (the lane's own max|dO_i|), per (image, head) in dK / dV.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// power-of-two scale for a group whose largest |value| has the float bits `amax_bits`: amax * scale in [2^13, 2^14)
__device__ __forceinline__ void attn_pow2_scale(uint32_t amax_bits, float& scale, float& inv) {
  int E = (int)((amax_bits >> 23) & 0xffu);
  E = E < 14 ? 14 : (E > 253 ? 253 : E);
  scale = __uint_as_float((uint32_t)(267 - E) << 23);
  inv = __uint_as_float((uint32_t)(E - 13) << 23);
}
// 8 floats (already scaled) -> hi and mid fragments of 8 fp16.  (Scalars, not a 2-vector of packed words: hipcc 7.2 folds
// bit_cast<f16x2>(v[1]) of a uint2 to v[0]'s halves.)
__device__ __forceinline__ void split8_f16(const float (&v)[8], bf16x8 (&out)[2]) {
  uint32_t h[4], m[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t hh = pack_f16(v[2 * i], v[2 * i + 1]);
    const f32x2 f = __builtin_convertvector(__builtin_bit_cast(f16x2, hh), f32x2);
    h[i] = hh;
    m[i] = pack_f16(v[2 * i] - f[0], v[2 * i + 1] - f[1]);
  }
  out[0] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  out[1] = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
}
__device__ __forceinline__ f32x16 products_f16(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
  return c;
}

// stage_tile with the tile's values scaled by `sc` (a power of two) and split into fp16 hi / mid images
template <bool RM, bool TR>
__device__ __forceinline__ void stage_tile_f16(char* __restrict__ rm, char* __restrict__ tr, const f32x4 (&r)[4], float sc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (int)threadIdx.x + 256 * i;
    const int row = e >> 4, d4 = e & 15;
    const float x0 = r[i][0] * sc, x1 = r[i][1] * sc, x2 = r[i][2] * sc, x3 = r[i][3] * sc;
    const uint32_t h0 = pack_f16(x0, x1), h1 = pack_f16(x2, x3);
    const f32x2 f0 = __builtin_convertvector(__builtin_bit_cast(f16x2, h0), f32x2);
    const f32x2 f1 = __builtin_convertvector(__builtin_bit_cast(f16x2, h1), f32x2);
    const uint32_t m0 = pack_f16(x0 - f0[0], x1 - f0[1]), m1 = pack_f16(x2 - f1[0], x3 - f1[1]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t p0 = t ? m0 : h0, p1 = t ? m1 : h1;
      if (RM) *reinterpret_cast<u32x2*>(rm + t * kImg + row * 128 + (((d4 >> 1) ^ ((row >> 1) & 7)) << 4) + (d4 & 1) * 8) = u32x2{p0, p1};
      if (TR) {
        const int d = 4 * d4;
        char* q = tr + t * kImg + (row & 3) * 2;
        const int slot = row >> 2;
        *reinterpret_cast<uint16_t*>(q + (d + 0) * 128 + ((slot ^ (((d + 0) >> 1) & 15)) << 3)) = (uint16_t)(p0 & 0xffffu);
        *reinterpret_cast<uint16_t*>(q + (d + 1) * 128 + ((slot ^ (((d + 1) >> 1) & 15)) << 3)) = (uint16_t)(p0 >> 16);
        *reinterpret_cast<uint16_t*>(q + (d + 2) * 128 + ((slot ^ (((d + 2) >> 1) & 15)) << 3)) = (uint16_t)(p1 & 0xffffu);
        *reinterpret_cast<uint16_t*>(q + (d + 3) * 128 + ((slot ^ (((d + 3) >> 1) & 15)) << 3)) = (uint16_t)(p1 >> 16);
      }
    }
  }
}

// a lane's 32 head-dim values times `mul`, scaled by the ROW's own power of two (both halves of the row agree on it through
// one shuffle) and split; returns the inverse scale and the row's max |value * mul|
__device__ __forceinline__ void load_row_frag_f16(const float* __restrict__ base, int64_t row_stride, int row, int half, float mul,
                                                  RowFrag<2>& out, float& inv, float& amax) {
  const float* p = base + (int64_t)row * row_stride + 32 * half;
  f32x4 a[8];
  uint32_t mb = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    a[s] = *reinterpret_cast<const f32x4*>(p + 4 * s);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[s][e] *= mul;
      const uint32_t b = __float_as_uint(a[s][e]) & 0x7fffffffu;
      mb = b > mb ? b : mb;
    }
  }
  const uint32_t other = (uint32_t)__shfl_xor((int)mb, 32, 64);
  mb = other > mb ? other : mb;
  float sc;
  attn_pow2_scale(mb, sc, inv);
  amax = __uint_as_float(mb);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float v[8] = {a[2 * s][0] * sc, a[2 * s][1] * sc, a[2 * s][2] * sc, a[2 * s][3] * sc,
                        a[2 * s + 1][0] * sc, a[2 * s + 1][1] * sc, a[2 * s + 1][2] * sc, a[2 * s + 1][3] * sc};
    split8_f16(v, out.f[s]);
  }
}

__device__ __forceinline__ f32x16 rm_times_frag_f16(const char* __restrict__ rm, int rb, int lane, const RowFrag<2>& q, f32x16 acc) {
  const int row = 32 * rb + (lane & 31), half = lane >> 5;
  const char* p = rm + row * 128;
  const int sw = (row >> 1) & 7;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 a[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const bf16x8*>(p + t * kImg + (((4 * half + s) ^ sw) << 4));
    acc = products_f16(a, q.f[s], acc);
  }
  return acc;
}

// tr_times_acc with the accumulator operand multiplied by `bsc` (a power of two) before the split
__device__ __forceinline__ void tr_times_acc_f16(const char* __restrict__ tr, int rb, int lane, const f32x16& x, float bsc, f32x16& o0,
                                                 f32x16& o1) {
  const int half = lane >> 5;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[8 * u + j] * bsc;
    bf16x8 b[2];
    split8_f16(v, b);
    const int slot = 8 * rb + 4 * u + half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int d = 32 * dt + (lane & 31);
      const char* p = tr + d * 128;
      const int sw = (d >> 1) & 15;
      bf16x8 a[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(p + t * kImg + ((slot ^ sw) << 3));
        const u32x2 hi = *reinterpret_cast<const u32x2*>(p + t * kImg + (((slot + 2) ^ sw) << 3));
        a[t] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
      }
      if (dt == 0)
        o0 = products_f16(a, b, o0);
      else
        o1 = products_f16(a, b, o1);
    }
  }
}

// float bits of max |q|, |k|, |v|, |dO| over the T rows of every (image, head): out[(b H + h) 4 + {0, 1, 2, 3}].
// grid (H, B, Z): block z takes rows [z T / Z, (z + 1) T / Z) and merges its maxima with atomicMax (float bits of
// non-negative values order like unsigned integers), so `out` is zeroed by attn_zero_words_kernel first.  (Round 5's first
// version ran ONE block per (image, head): 48 blocks on 256 CUs, 35 us per launch, 28 launches per Segmenter step = 7 % of it.)
__global__ void attn_zero_words_kernel(uint32_t* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

template <int NT>   // 3: q, k, v (forward); 4: + dO
__global__ __launch_bounds__(256) void attn_amax_bh_kernel(AttnPtrsB p, int T, int H, const float* __restrict__ go,
                                                           uint32_t* __restrict__ out) {
  const int b = blockIdx.y, h = blockIdx.x;
  const int r0 = (int)((int64_t)T * blockIdx.z / gridDim.z), r1 = (int)((int64_t)T * (blockIdx.z + 1) / gridDim.z);
  const float* src[4] = {p.q + (int64_t)b * p.sb + (int64_t)h * p.sh, p.k + (int64_t)b * p.sb + (int64_t)h * p.sh,
                         p.v + (int64_t)b * p.sb + (int64_t)h * p.sh,
                         NT == 4 ? go + ((int64_t)b * T) * (H * kD) + h * kD : nullptr};
  const int64_t stride[4] = {p.st, p.st, p.st, (int64_t)H * kD};
  uint32_t m[4] = {0, 0, 0, 0};
  const int d4 = threadIdx.x & 15;
  for (int row = r0 + (threadIdx.x >> 4); row < r1; row += 16) {
#pragma unroll
    for (int w = 0; w < NT; ++w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src[w] + (int64_t)row * stride[w] + 4 * d4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t bits = __float_as_uint(v[e]) & 0x7fffffffu;
        m[w] = bits > m[w] ? bits : m[w];
      }
    }
  }
  __shared__ uint32_t part[4][4];
#pragma unroll
  for (int w = 0; w < NT; ++w) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t other = (uint32_t)__shfl_xor((int)m[w], o, 64);
      m[w] = other > m[w] ? other : m[w];
    }
    if ((threadIdx.x & 63) == 0) part[w][threadIdx.x >> 6] = m[w];
  }
  __syncthreads();
  if (threadIdx.x < NT) {
    const int w = threadIdx.x;
    uint32_t r = part[w][0];
    for (int i = 1; i < 4; ++i) r = part[w][i] > r ? part[w][i] : r;
    atomicMax(out + ((int64_t)b * H + h) * 4 + w, r);
  }
}

static inline void attn_amax_launch(const AttnPtrsB& p, int B, int H, int T, const float* grad_out, uint32_t* amax_ws,
                                    hipStream_t stream) {
  const int words = 4 * B * H;
  int Z = (T + 63) / 64;                 // >= 64 rows per block
  Z = Z < 1 ? 1 : (Z > 16 ? 16 : Z);
  hipLaunchKernelGGL(attn_zero_words_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, amax_ws, words);
  if (grad_out)
    hipLaunchKernelGGL(attn_amax_bh_kernel<4>, dim3(H, B, Z), dim3(256), 0, stream, p, T, H, grad_out, amax_ws);
  else
    hipLaunchKernelGGL(attn_amax_bh_kernel<3>, dim3(H, B, Z), dim3(256), 0, stream, p, T, H, grad_out, amax_ws);
}

// ---- forward, fp16 x 2: the loop of attn_fwd_bf16_kernel with 22-bit operands in three products per pair ---------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_f16_kernel(AttnPtrsB p, int T, int H, float scale, const uint32_t* __restrict__ amax_bh,
                                                              float* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) char k_rm[2 * kImg];
  __shared__ __attribute__((aligned(16))) char v_tr[2 * kImg];
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q_row = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_row < T;
  const bool wave_rows = blockIdx.x * 128 + wave * 32 < T;
  const float* qb = p.q + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* kb = p.k + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* vb = p.v + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const uint32_t* aw = amax_bh + ((int64_t)b * H + h) * 4;
  float sK, invK, sV, invV;
  attn_pow2_scale(aw[1], sK, invK);
  attn_pow2_scale(aw[2], sV, invV);
  RowFrag<2> qf;  // Q[q_row][.] * scale * log2(e): scores come out in the log2 domain
  float inv_q, amax_q;
  load_row_frag_f16(qb, p.st, q_ok ? q_row : T - 1, half, scale * kLog2e, qf, inv_q, amax_q);
  const float c_s = invK * inv_q;
  const float s_p = 16384.f;                                           // P = exp2(s - max) <= 1
  f32x16 o0 = zero16(), o1 = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  const int n_tiles = (T + kTile - 1) / kTile;
  f32x4 kr[4], vr[4];
  load_tile_regs(kb, p.st, 0, T, kr);
  load_tile_regs(vb, p.st, 0, T, vr);
  for (int j = 0; j < n_tiles; ++j) {
    __syncthreads();
    stage_tile_f16<true, false>(k_rm, nullptr, kr, sK);
    stage_tile_f16<false, true>(nullptr, v_tr, vr, sV);
    __syncthreads();
    if (j + 1 < n_tiles) {
      load_tile_regs(kb, p.st, (j + 1) * kTile, T, kr);
      load_tile_regs(vb, p.st, (j + 1) * kTile, T, vr);
    }
    if (!wave_rows) continue;
    f32x16 s0 = rm_times_frag_f16(k_rm, 0, lane, qf, zero16());
    f32x16 s1 = rm_times_frag_f16(k_rm, 1, lane, qf, zero16());
    const int key0 = j * kTile;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      s0[t] *= c_s;
      s1[t] *= c_s;
    }
    if (key0 + kTile > T) {
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
    const float m_new = fmaxf(m_run, m_t);
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
    tr_times_acc_f16(v_tr, 0, lane, s0, s_p, o0, o1);
    tr_times_acc_f16(v_tr, 1, lane, s1, s_p, o0, o1);
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  store_rows_t(out + ((int64_t)b * T) * (H * kD) + h * kD, (int64_t)H * kD, q_row, q_ok, lane, o0, o1, invV * (1.f / 16384.f) / l_tot);
  if (q_ok && half == 0) lse[((int64_t)b * H + h) * T + q_row] = (m_run + log2f(l_tot)) * kLn2;
}

// ---- dQ, fp16 x 2 ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_dq_f16_kernel(AttnPtrsB p, int T, int H, float scale, const float* __restrict__ go,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             const uint32_t* __restrict__ amax_bh, float* __restrict__ dq,
                                                             int64_t gsb, int64_t gsh, int64_t gst) {
  extern __shared__ __attribute__((aligned(16))) char smem_dq16[];
  char* k_rm = smem_dq16;
  char* k_tr = k_rm + 2 * kImg;
  char* v_rm = k_tr + 2 * kImg;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q_row = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_row < T;
  const int q_ld = q_ok ? q_row : T - 1;
  const float* qb = p.q + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* kb = p.k + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const float* vb = p.v + (int64_t)b * p.sb + (int64_t)h * p.sh;
  const uint32_t* aw = amax_bh + ((int64_t)b * H + h) * 4;
  float sK, invK, sV, invV;
  attn_pow2_scale(aw[1], sK, invK);
  attn_pow2_scale(aw[2], sV, invV);
  RowFrag<2> qf, gf;
  float inv_q, inv_g, amax_q, amax_g;
  load_row_frag_f16(qb, p.st, q_ld, half, scale * kLog2e, qf, inv_q, amax_q);
  load_row_frag_f16(go + ((int64_t)b * T) * (H * kD) + h * kD, (int64_t)H * kD, q_ld, half, 1.f, gf, inv_g, amax_g);
  // |dS| <= (|dP| + |delta|) scale <= 2 x 64 max|dO_i| max|V| scale
  float s_ds, inv_ds;
  attn_pow2_scale(__float_as_uint(128.f * amax_g * __uint_as_float(aw[2]) * scale), s_ds, inv_ds);
  const float c_s = invK * inv_q, c_dp = invV * inv_g;
  const float lse2 = lse[((int64_t)b * H + h) * T + q_ld] * kLog2e;
  const float dlt = delta[((int64_t)b * H + h) * T + q_ld];
  f32x16 dq0 = zero16(), dq1 = zero16();
  const int n_tiles = (T + kTile - 1) / kTile;
  f32x4 kr[4], vr[4];
  load_tile_regs(kb, p.st, 0, T, kr);
  load_tile_regs(vb, p.st, 0, T, vr);
  for (int j = 0; j < n_tiles; ++j) {
    __syncthreads();
    stage_tile_f16<true, true>(k_rm, k_tr, kr, sK);
    stage_tile_f16<true, false>(v_rm, nullptr, vr, sV);
    __syncthreads();
    if (j + 1 < n_tiles) {
      load_tile_regs(kb, p.st, (j + 1) * kTile, T, kr);
      load_tile_regs(vb, p.st, (j + 1) * kTile, T, vr);
    }
    if (blockIdx.x * 128 + wave * 32 >= T) continue;
    const int key0 = j * kTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 s = rm_times_frag_f16(k_rm, rb, lane, qf, zero16());    // S^T sK sq (log2 domain)
      f32x16 dp = rm_times_frag_f16(v_rm, rb, lane, gf, zero16());   // dP^T sV sg
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool exists = key0 + 32 * rb + acc_row(t, half) < T;
        const float pr = exists ? fast_exp2(s[t] * c_s - lse2) : 0.f;
        s[t] = pr * (dp[t] * c_dp - dlt) * scale;                      // dS^T
      }
      tr_times_acc_f16(k_tr, rb, lane, s, s_ds, dq0, dq1);             // dQ^T sK s_ds += K^T dS^T
    }
  }
  store_rows_t(dq + (int64_t)b * gsb + (int64_t)h * gsh, gst, q_row, q_ok, lane, dq0, dq1, invK * inv_ds);
}

// ---- dK, dV, fp16 x 2 --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_dkv_f16_kernel(AttnPtrsB p, int T, int H, float scale, const float* __restrict__ go,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              const uint32_t* __restrict__ amax_bh, float* __restrict__ dk,
                                                              float* __restrict__ dv, int64_t gsb, int64_t gsh, int64_t gst) {
  extern __shared__ __attribute__((aligned(16))) char smem_dkv16[];
  char* q_rm = smem_dkv16;
  char* q_tr = q_rm + 2 * kImg;
  char* g_rm = q_tr + 2 * kImg;
  char* g_tr = g_rm + 2 * kImg;
  float* lse_s = reinterpret_cast<float*>(g_tr + 2 * kImg);
  float* dlt_s = lse_s + kTile;
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
  const uint32_t* aw = amax_bh + ((int64_t)b * H + h) * 4;
  float sQ, invQ, sG, invG;
  attn_pow2_scale(aw[0], sQ, invQ);
  attn_pow2_scale(aw[3], sG, invG);
  RowFrag<2> kf, vf;
  float inv_k, inv_v, amax_k, amax_v;
  load_row_frag_f16(kb, p.st, k_ld, half, scale * kLog2e, kf, inv_k, amax_k);
  load_row_frag_f16(vb, p.st, k_ld, half, 1.f, vf, inv_v, amax_v);
  // |dS| <= 2 x 64 max|dO| max|V| scale over the (image, head): the delta of a query is not bounded by THIS key's V row
  float s_ds, inv_ds;
  attn_pow2_scale(__float_as_uint(128.f * __uint_as_float(aw[3]) * __uint_as_float(aw[2]) * scale), s_ds, inv_ds);
  const float s_p = 16384.f, inv_p = 1.f / 16384.f;                   // P <= 1
  const float c_s = invQ * inv_k, c_dp = invG * inv_v;
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const int n_tiles = (T + kTile - 1) / kTile;
  for (int j = 0; j < n_tiles; ++j) {
    f32x4 qr[4], gr[4];
    load_tile_regs(qb, p.st, j * kTile, T, qr);
    load_tile_regs(gb, gst_o, j * kTile, T, gr);
    __syncthreads();
    stage_tile_f16<true, true>(q_rm, q_tr, qr, sQ);
    stage_tile_f16<true, true>(g_rm, g_tr, gr, sG);
    if (threadIdx.x < kTile) {
      const int qq = j * kTile + (int)threadIdx.x;
      const int ql = qq < T ? qq : T - 1;
      lse_s[threadIdx.x] = lse[((int64_t)b * H + h) * T + ql] * kLog2e;
      dlt_s[threadIdx.x] = delta[((int64_t)b * H + h) * T + ql];
    }
    __syncthreads();
    if (blockIdx.x * 128 + wave * 32 >= T) continue;
    const int q0 = j * kTile;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 pr = rm_times_frag_f16(q_rm, rb, lane, kf, zero16());   // S sQ sk
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int ql = 32 * rb + acc_row(t, half);
        const bool exists = (q0 + ql < T) && k_ok;
        pr[t] = exists ? fast_exp2(pr[t] * c_s - lse_s[ql]) : 0.f;    // P
      }
      tr_times_acc_f16(g_tr, rb, lane, pr, s_p, dv0, dv1);            // dV^T sG s_p += dO^T P
      f32x16 dp = rm_times_frag_f16(g_rm, rb, lane, vf, zero16());   // dP sG sv
#pragma unroll
      for (int t = 0; t < 16; ++t) pr[t] = pr[t] * (dp[t] * c_dp - dlt_s[32 * rb + acc_row(t, half)]) * scale;   // dS
      tr_times_acc_f16(q_tr, rb, lane, pr, s_ds, dk0, dk1);           // dK^T sQ s_ds += Q^T dS
    }
  }
  store_rows_t(dk + (int64_t)b * gsb + (int64_t)h * gsh, gst, k_row, k_ok, lane, dk0, dk1, invQ * inv_ds);
  store_rows_t(dv + (int64_t)b * gsb + (int64_t)h * gsh, gst, k_row, k_ok, lane, dv0, dv1, invG * inv_p);
}

}  // namespace sea

using namespace sea;

// entry points used by csrc/attention.hip's sea_attention_fwd / sea_attention_bwd when the split path is selected
int sea_attention_fwd_bf16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                           float scale, float* out, float* lse, int terms, hipStream_t stream) {
  AttnPtrsB p{q, k, v, sb, sh, st};
  dim3 grid((T + 127) / 128, H, B), block(256);
  if (terms == 3)
    hipLaunchKernelGGL(attn_fwd_bf16_kernel<3>, grid, block, 0, stream, p, T, H, scale, out, lse);
  else
    hipLaunchKernelGGL(attn_fwd_bf16_kernel<2>, grid, block, 0, stream, p, T, H, scale, out, lse);
  return (int)hipGetLastError();
}

int sea_attention_bwd_bf16(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                           float scale, const float* grad_out, const float* lse, const float* delta, float* dq, float* dk,
                           float* dv, int64_t gsb, int64_t gsh, int64_t gst, int terms, hipStream_t stream) {
  AttnPtrsB p{q, k, v, sb, sh, st};
  dim3 grid((T + 127) / 128, H, B), block(256);
  const size_t lds = (size_t)4 * terms * kImg + 2 * kTile * sizeof(float);
  // 4 x 3 x 8 KB = 96 KB of dynamic LDS at three terms: opt in once per DEVICE (the attribute is per device; a process that
  // drives a second GPU would otherwise fail its first dkv launch there)
  static bool attr_set_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool& attr_set = attr_set_dev[dev & 63];
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_dkv_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * kImg + 512);
    (void)hipFuncSetAttribute((const void*)attn_dkv_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * kImg + 512);
    (void)hipFuncSetAttribute((const void*)attn_dq_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * kImg);
    (void)hipFuncSetAttribute((const void*)attn_dq_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * kImg);
    attr_set = true;
  }
  if (terms == 3) {
    hipLaunchKernelGGL(attn_dq_bf16_kernel<3>, grid, block, (size_t)3 * 3 * kImg, stream, p, T, H, scale, grad_out, lse, delta, dq, gsb, gsh, gst);
    hipLaunchKernelGGL(attn_dkv_bf16_kernel<3>, grid, block, lds, stream, p, T, H, scale, grad_out, lse, delta, dk, dv, gsb, gsh, gst);
  } else {
    hipLaunchKernelGGL(attn_dq_bf16_kernel<2>, grid, block, (size_t)3 * 2 * kImg, stream, p, T, H, scale, grad_out, lse, delta, dq, gsb, gsh, gst);
    hipLaunchKernelGGL(attn_dkv_bf16_kernel<2>, grid, block, lds, stream, p, T, H, scale, grad_out, lse, delta, dk, dv, gsb, gsh, gst);
  }
  return (int)hipGetLastError();
}

// fp16 x 2 backward: amax_ws = 4 B H device words of scratch (filled here)
int sea_attention_bwd_f16x2(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                            float scale, const float* grad_out, const float* lse, const float* delta, uint32_t* amax_ws, float* dq,
                            float* dk, float* dv, int64_t gsb, int64_t gsh, int64_t gst, hipStream_t stream) {
  AttnPtrsB p{q, k, v, sb, sh, st};
  dim3 grid((T + 127) / 128, H, B), block(256);
  const size_t lds_dkv = (size_t)4 * 2 * kImg + 2 * kTile * sizeof(float), lds_dq = (size_t)3 * 2 * kImg;
  static bool attr_set_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set_dev[dev & 63]) {
    (void)hipFuncSetAttribute((const void*)attn_dkv_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv);
    (void)hipFuncSetAttribute((const void*)attn_dq_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq);
    attr_set_dev[dev & 63] = true;
  }
  attn_amax_launch(p, B, H, T, grad_out, amax_ws, stream);
  hipLaunchKernelGGL(attn_dq_f16_kernel, grid, block, lds_dq, stream, p, T, H, scale, grad_out, lse, delta, amax_ws, dq, gsb, gsh, gst);
  hipLaunchKernelGGL(attn_dkv_f16_kernel, grid, block, lds_dkv, stream, p, T, H, scale, grad_out, lse, delta, amax_ws, dk, dv, gsb, gsh,
                     gst);
  return (int)hipGetLastError();
}

// fp16 x 2 forward: amax_ws = 4 B H device words of scratch (filled here)
int sea_attention_fwd_f16x2(const float* q, const float* k, const float* v, int64_t sb, int64_t sh, int64_t st, int B, int H, int T,
                            float scale, uint32_t* amax_ws, float* out, float* lse, hipStream_t stream) {
  AttnPtrsB p{q, k, v, sb, sh, st};
  dim3 grid((T + 127) / 128, H, B), block(256);
  attn_amax_launch(p, B, H, T, nullptr, amax_ws, stream);
  hipLaunchKernelGGL(attn_fwd_f16_kernel, grid, block, 0, stream, p, T, H, scale, amax_ws, out, lse);
  return (int)hipGetLastError();
}

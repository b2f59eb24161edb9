// M8, 256 x 256 tiles for the Winograd-domain products (fp16 x 2; N a multiple of 256; thousands of tiles).
//
// Why a third kernel.  The knock-out timings of round 5 (profiles/r5_gemm_knockout_big.log, devtools/gemm_knockout.py) say what
// bounds the largest launch (36 x 8192 x 512 x 512, 545 us isolated): not the MFMAs (-100 us without them), not HBM (1.28 GB =
// the algorithmic minimum, TCC counters), but the bytes every CU pulls through its vector L1: 4.84 GB per launch = 128 x 128
// tiles re-reading A four times and the packed weights sixty-four times out of L2 (-140 us without EITHER half of the loads).
// A 128 x 128 x 32 step needs 32 KB of loads for 768 MFMA cycles per SIMD: 42 B/clk/CU of a 64 B/clk path.  Only a larger tile
// lowers that: 256 x 256 x 32 takes 64 KB for 3072 cycles, half the bytes per flop.
//
// Shape: 512 threads = 8 waves, two per SIMD, ONE block per CU; two LDS stages of 64 KB + row scales; two configurations:
//   256 x 256  waves 4 (M) x 2 (N), wave tile 64 x 128 (eight 32x32x16 accumulators = 128 VGPRs): the Winograd-domain products
//              (N a multiple of 256, >= 1024 tiles);
//   128 x 384  waves 2 x 4, wave tile 64 x 96 (six accumulators): the 32 x 32-pixel stage of ConvNeXt (M = 8192, N = 1536 or
//              split-K slices of N = 384): 768 tiles of 128 x 128 -- one round of three blocks per CU, every block re-reading
//              its weights -- become 256 tiles = ONE block per CU, two thirds of the L1 bytes per flop; with the GELU /
//              GELU' / gate prologues of gemm_split_pp.hip.
// The K loop is the ping-pong loop of gemm_split_pp.hip: one barrier per K step, the operand split of the next tile placed by
// hand between the wave's own MFMAs, buffer loads with a zero-record descriptor past the end of K.  Same split, same MFMA
// order per accumulator as the other two kernels: the same bits.
#include "gemm_split.h"

namespace sea {

// WM x WN waves (WM WN = 8), wave tile 64 x 32 TN; PRO: 0 none, 1 A * GELU'(t), 2 GELU(A), 3 t > 0 ? A : 0
// DEPTH: register sets of staged loads (2 = a tile's loads are issued two K steps before its split)
// MI: 32-row sub-tiles of a wave's tile (2: eight waves, two per SIMD; 4: FOUR waves of 128 x 128, one per SIMD, accumulators
// in the AGPR half of the 512-register file -- a third fewer LDS fragment reads per MFMA)
// WDMA: the packed weights go straight into the LDS stage by LDS-DMA (global_load_lds_dwordx4: the chunk swizzle applied to the
// SOURCE address, one wave-instruction = 16 rows of an image = 1 KB of LDS in lane order): no staging registers, no ds_write for
// half of a K step's LDS bytes
template <bool F16, int WM, int WN, int TN, int PRO, int DEPTH, int MI = 2, bool WDMA = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_split_big_kernel(const GemmSplitArgs p) {
  static_assert(WM * WN == 8 || (WM * WN == 4 && MI == 4), "eight waves, or four with 128-row wave tiles");
  constexpr int TERMS = 2;
  constexpr int NT = 64 * WM * WN;                         // threads
  constexpr int RP = NT / 8;                               // rows of A staged per pass (eight float4 columns)
  constexpr int BM = 32 * MI * WM, WT = 32 * TN, BN = WT * WN;
  constexpr int IMG_A = BM * 64, IMG_W = BN * 64;          // bytes of one term image (rows x 64 B)
  constexpr int STAGE = TERMS * (IMG_A + IMG_W);           // A images, then W images: 64 KB in both configurations
  constexpr int NA = BM / RP;                              // float4 loads of A per thread and K step (rows arow + RP i)
  constexpr int NWT = BN * 4 / NT, NW = TERMS * NWT;       // 16-byte weight pieces per thread: per term, in all
  constexpr int SLOTS = 3 * MI * TN;                       // MFMAs of one 16-deep half: MI TN chains of three products
  constexpr int UNIT_EVERY = SLOTS / (3 * NA);             // an A micro-unit behind every UNIT_EVERY-th product of the first half
  constexpr bool HAS_T = (PRO == 1 || PRO == 3);
  static_assert(STAGE == 65536 && SLOTS % (3 * NA) == 0 && NW + 2 <= SLOTS, "configuration");
  constexpr int NBLK = TERMS * BN / 16;                    // WDMA: 1 KB blocks (16 rows of an image) of a stage's weights
  constexpr int NVM = NA * (HAS_T ? 2 : 1);                // vector-memory loads a step issues AFTER its DMAs (fetch_a)
  static_assert(!WDMA || (NT == 512 && NBLK % 8 == 0 && DEPTH == 1), "LDS-DMA weights: eight waves, one register set");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const row_sc = (float*)(smem + 2 * STAGE);
  float* const row_inv = row_sc + BM;

  const int M = p.M, N = p.N, K = p.K, Npad = p.Npad;
  const int64_t lda = p.lda, ldc = p.ldc;
  const int logical = (int)(blockIdx.x & 7) * p.per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.per_xcd || logical >= p.total) return;
  const int nb = logical % p.nblocks;
  const int t2 = logical / p.nblocks;
  const int mb = t2 % p.mblocks;
  const int g = t2 / p.mblocks;
  const int m0 = mb * BM, n0 = nb * BN;
  const float* const w_inv = F16 ? (const float*)((const char*)p.w_inv + (int64_t)g * p.strideW) : nullptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int q = tid & 7, arow = tid >> 3;              // A staging: float4 column q of rows arow + RP i

  const char* const Abase = (const char*)(p.A + (int64_t)g * p.strideA);
  const char* const Tbase = HAS_T ? (const char*)(p.a_gelu_grad_of + (int64_t)g * p.strideA) : nullptr;
  const float* const bias = p.bias;
  const int nkb = K / GS_BK;
  const char* const Wbase = p.W + (int64_t)g * p.strideW + (int64_t)n0 * 64;
  const int64_t w_term = (int64_t)Npad * 64, w_kb = (int64_t)TERMS * w_term;
  const uint32_t woff = (uint32_t)tid * 16;
  const uint32_t a_bytes = (uint32_t)((((int64_t)(M - 1)) * lda + K) * 4);
  const uint32_t w_bytes = (uint32_t)((int64_t)nkb * w_kb - (int64_t)n0 * 64);

  float a_sc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) a_sc[i] = 1.f;
  // One VGPR per address family (the K loop has no register to spare): rows arow + RP i of A share their swizzle (RP = 64 / 32
  // rows further = the same (row >> 2) & 3), so their LDS offsets differ by 64 RP i and their global offsets by a SCALAR RP i lda
  // (a row past M takes the descriptor's size as its lane offset: out of range whatever the hardware adds to it -> zeros);
  // the weight pieces of a thread (tid + 512 j of a term's image: 128 rows further each) share theirs too.
  const uint32_t aoff0 = (uint32_t)(((int64_t)(m0 + arow) * lda + 4 * q) * 4);
  const uint32_t a_wr0 = (uint32_t)(arow * 64 + swz<false>(arow, q >> 1) + (q & 1) * 8);
  const int wrow0 = tid >> 2;
  const uint32_t w_wr0 = (uint32_t)(TERMS * IMG_A + wrow0 * 64 + swz<false>(wrow0, tid & 3));
  const int a_row_step = (int)(RP * lda * 4);          // bytes between rows arow + RP i (lda < 2^22: checked by the launcher)
  const int ra = wm * 32 * MI + r, rb = wn * WT + r;
  const uint32_t a_rd = (uint32_t)(ra * 64 + swz<false>(ra, h));                 // mi: + 2048, s: ^ 32, term: + IMG_A
  const uint32_t b_rd = (uint32_t)(TERMS * IMG_A + rb * 64 + swz<false>(rb, h));   // ni: + 2048, term: + IMG_W

  struct Tile {
    f32x4 a[NA];
    f32x4 t[HAS_T ? NA : 1];
    u32x4 w[WDMA ? 1 : NW];
  };
  Tile R0 = {}, R1 = {};
  auto fetch_a = [&](Tile& R, int kb) __attribute__((always_inline)) {
    f32x4 (&Ra)[NA] = R.a;
    f32x4 (&Rt)[HAS_T ? NA : 1] = R.t;
    const int live = kb < nkb ? (int)a_bytes : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, live, 0x00020000);
#pragma unroll
    for (int i = 0; i < NA; ++i)
      Ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rs, (int)((m0 + arow + RP * i < M) ? aoff0 : a_bytes), kb * (GS_BK * 4) + i * a_row_step, 0));   // (rows past M: out of range -> zeros)
    if constexpr (HAS_T) {
      const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)Tbase, 0, live, 0x00020000);
#pragma unroll
      for (int i = 0; i < NA; ++i)
        Rt[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rt, (int)((m0 + arow + RP * i < M) ? aoff0 : a_bytes), kb * (GS_BK * 4) + i * a_row_step, 0));
    }
  };
  const int wave_u0 = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lane_off = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
  auto dma_w = [&](char* stage, int kb) __attribute__((always_inline)) {
    if constexpr (WDMA) {
      const int kbc = kb < nkb ? kb : nkb - 1;               // (past the end of K: a valid block again, never consumed)
#pragma unroll
      for (int i = 0; i < NBLK / 8; ++i) {
        const int blk = wave_u0 + 8 * i;                     // wave-uniform: block of 16 rows
        const int t = blk / (BN / 16), rb16 = blk - t * (BN / 16);
        const char* const src = Wbase + (int64_t)kbc * w_kb + (int64_t)t * w_term + rb16 * 1024 + lane_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + TERMS * IMG_A + blk * 1024), 16, 0, 0);
      }
    }
  };
  auto fetch_w = [&](Tile& R, int kb) __attribute__((always_inline)) {
    if constexpr (WDMA) return;
    u32x4 (&Rw)[WDMA ? 1 : NW] = R.w;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wbase, 0, kb < nkb ? (int)w_bytes : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < NW; ++i)   // piece j = i % NWT of term i / NWT: bytes 16 NT j + 16 tid of the term's image
      Rw[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rs, (int)woff, kb * (int)w_kb + (i / NWT) * (int)w_term + (i % NWT) * (16 * NT), 0));
  };
  auto write_w = [&](Tile& R, int i, char* stage) __attribute__((always_inline)) {
    if constexpr (!WDMA) *(u32x4*)(stage + w_wr0 + (i / NWT) * IMG_W + (i % NWT) * (16 * NT)) = R.w[i];
  };

  // staging micro-units of A row i (see gemm_split_pp.hip): (0) prologue + scale + first term, (1) remainder, (2) second term + writes
  f32x4 sv;
  uint32_t hi_a, hi_b;
  auto unit_a = [&](Tile& R, int i, int part, char* stage) __attribute__((always_inline)) {
    f32x4 (&Rt)[HAS_T ? NA : 1] = R.t;
    if (part == 0) {
      f32x4 v = R.a[i];
      if constexpr (PRO == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(Rt[i][e]);
      }
      if constexpr (PRO == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      }
      if constexpr (PRO == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = Rt[i][e] > 0.f ? v[e] : 0.f;
      }
      if constexpr (F16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= a_sc[i];
        hi_a = pack_f16(v[0], v[1]);
        hi_b = pack_f16(v[2], v[3]);
      } else {
        hi_a = pack_bf16(v[0], v[1]);
        hi_b = pack_bf16(v[2], v[3]);
      }
      sv = v;
    } else if (part == 1) {
      if constexpr (F16) {
        const f32x2 f0 = __builtin_convertvector(__builtin_bit_cast(f16x2, hi_a), f32x2);
        const f32x2 f1 = __builtin_convertvector(__builtin_bit_cast(f16x2, hi_b), f32x2);
        sv[0] -= f0[0];
        sv[1] -= f0[1];
        sv[2] -= f1[0];
        sv[3] -= f1[1];
      } else {
#pragma clang fp contract(off)
        sv[0] -= __uint_as_float(hi_a << 16);
        sv[1] -= __uint_as_float(hi_a & 0xffff0000u);
        sv[2] -= __uint_as_float(hi_b << 16);
        sv[3] -= __uint_as_float(hi_b & 0xffff0000u);
      }
    } else {
      *(u32x2*)(stage + a_wr0 + 64 * RP * i) = u32x2{hi_a, hi_b};
      const u32x2 mid = F16 ? u32x2{pack_f16(sv[0], sv[1]), pack_f16(sv[2], sv[3])}
                            : u32x2{pack_bf16(sv[0], sv[1]), pack_bf16(sv[2], sv[3])};
      *(u32x2*)(stage + IMG_A + a_wr0 + 64 * RP * i) = mid;
    }
  };

  bf16x8 fa[MI][TERMS], fb[TN][TERMS];
  auto read_a = [&](const char* stage, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int t = 0; t < TERMS; ++t) fa[mi][t] = *(const bf16x8*)(stage + ((a_rd ^ (uint32_t)(32 * s)) + mi * 2048 + t * IMG_A));
  };
  auto read_b = [&](const char* stage, int s, int ni) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < TERMS; ++t) fb[ni][t] = *(const bf16x8*)(stage + ((b_rd ^ (uint32_t)(32 * s)) + ni * 2048 + t * IMG_W));
  };
  auto mfma3 = [&](f32x16& c, int mi, int ni, int j) __attribute__((always_inline)) {
    // j-th product of the chain, smallest first: (mid, hi'), (hi, mid'), (hi, hi')
    const int ia = j == 0 ? 1 : 0, ib = j == 1 ? 1 : 0;
    if constexpr (F16)
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[mi][ia]), __builtin_bit_cast(f16x8, fb[ni][ib]), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ia], fb[ni][ib], c, 0, 0, 0);
  };

  f32x16 acc[MI][TN];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

#define SEA_PIN() __builtin_amdgcn_sched_barrier(0)
  // One K step on stage `cur`; tile kb + 1 (in Ra / Rw) is split into `nxt`, then the registers are refilled with tile `kf`.
  // 2 SLOTS MFMAs: the first 16-deep half carries the A micro-units and prefetches the second half's weight fragments column
  // by column as it releases their registers; the second half carries the weight pieces and the refill loads.
  auto step = [&](char* cur, char* nxt, Tile& Rn, int kf) __attribute__((always_inline)) {
    int unit = 0;
    dma_w(nxt, kf - DEPTH);                                          // (nobody reads `nxt` since the last barrier)
    SEA_PIN();
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          mfma3(acc[mi][ni], mi, ni, j);
          const int slot = (ni * MI + mi) * 3 + j;
          if (slot % UNIT_EVERY == 0 && unit < 3 * NA) {
            unit_a(Rn, unit / 3, unit % 3, nxt);
            ++unit;
          }
          SEA_PIN();
        }
      }
      read_b(cur, 1, ni);                                          // column ni of the first half is done: its registers take the second
      SEA_PIN();
    }
    read_a(cur, 1);
    SEA_PIN();
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          mfma3(acc[mi][ni], mi, ni, j);
          const int slot = (ni * MI + mi) * 3 + j;
          if (slot < NW) write_w(Rn, slot < NW ? slot : 0, nxt);
          if (slot == NW) fetch_a(Rn, kf);
          if (slot == NW + 1) fetch_w(Rn, kf);
          SEA_PIN();
        }
      }
    }
    if constexpr (WDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVM) : "memory");   // the DMAs are older than this step's fetch_a loads
    __syncthreads();                                               // every wave is done with `cur` and has written `nxt`
    read_a(nxt, 0);
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) read_b(nxt, 0, ni);
    SEA_PIN();
  };

  char* const st0 = smem;
  char* const st1 = smem + STAGE;
  // ---- prologue
  Tile& RA = R0;
  Tile& RB = DEPTH == 2 ? R1 : R0;
  dma_w(smem, 0);
  fetch_a(R0, 0);
  fetch_w(R0, 0);
  if constexpr (DEPTH == 2) {
    fetch_a(R1, 1);
    fetch_w(R1, 1);
  }
  if constexpr (F16) {
    if (tid < BM) {
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
    for (int i = 0; i < NA; ++i) a_sc[i] = row_sc[arow + RP * i];
  }
  SEA_PIN();
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    unit_a(R0, i, 0, st0);
    unit_a(R0, i, 1, st0);
    unit_a(R0, i, 2, st0);
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) write_w(R0, i, st0);
  fetch_a(R0, DEPTH);
  fetch_w(R0, DEPTH);
  if constexpr (WDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVM) : "memory");
  __syncthreads();
  read_a(st0, 0);
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) read_b(st0, 0, ni);
  int kb = 0;
  for (; kb + 2 <= nkb; kb += 2) {
    step(st0, st1, RB, kb + 1 + DEPTH);
    step(st1, st0, RA, kb + 2 + DEPTH);
  }
  if (kb < nkb) step(st0, st1, RB, nkb);
#undef SEA_PIN

  // ---- epilogue: each wave turns its 64 x WT tile through its share of the idle stages, 32 rows at a time, and stores 16
  // bytes per lane along the rows (see gemm_split.h for the 128 x 128 kernels' version)
  // (the per-column constants are loaded HERE, not before the K loop: the loop has no register to spare, and a spill inside it
  // is a scratch load that the in-order vmcnt makes wait for the refill loads)
  float bv_c[TN], wi_c[TN];
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + wn * WT + ni * 32 + r;
    bv_c[ni] = (bias && col < N) ? bias[col] : 0.f;
    wi_c[ni] = (F16 && col < N) ? w_inv[col] : 1.f;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int row0_u = m0 + (wave_u / WN) * 32 * MI;
  float* const Cg = p.C + (int64_t)g * p.strideC;
  constexpr int SCR = 32 * WT * 4;                      // bytes of a wave's 32-row scratch: 16 KB / 12 KB
  char* const scr = smem + wave_u * SCR;
  constexpr int F4_ROW = WT / 4;                        // float4 per scratch row: 32 / 24
  const bool vec = (((ldc | p.strideC | N) & 3) == 0) && ((((uintptr_t)p.C) & 15) == 0);
  const int relu = p.relu;
  uint32_t omax = 0;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    if (mi) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row_l = (e & 3) + 8 * (e >> 2) + 4 * h;
        const float v = (F16 ? acc[mi][ni][e] * (row_inv[(wave_u / WN) * 32 * MI + mi * 32 + row_l] * wi_c[ni]) : acc[mi][ni][e]) + bv_c[ni];
        *(float*)(scr + (row_l * WT + ni * 32 + r) * 4) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < WT / 8; ++k) {                  // 32 WT / 4 float4 of the scratch, 64 per trip, in address order
      const int f = k * 64 + lane;
      const int row_l = f / F4_ROW, c4 = f - row_l * F4_ROW;
      f32x4 v = *(const f32x4*)(scr + f * 16);
      const int row = row0_u + mi * 32 + row_l, col4 = n0 + wn * WT + 4 * c4;
      if (row >= M || col4 >= N) continue;
      float* const cp = Cg + (int64_t)row * ldc + col4;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (vec) {
        *(f32x4*)cp = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col4 + e < N) cp[e] = v[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t vb = (col4 + e < N) ? (__float_as_uint(v[e]) & 0x7fffffffu) : 0u;
        omax = vb > omax ? vb : omax;
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

template <bool F16, int WM, int WN, int TN, int PRO, int DEPTH, int MI = 2, bool WDMA = false>
static void big_launch_one(const GemmSplitArgs& p, hipStream_t st) {
  constexpr int lds = 2 * 65536 + 2 * 32 * MI * WM * (int)sizeof(float);
  auto k = gemm_split_big_kernel<F16, WM, WN, TN, PRO, DEPTH, MI, WDMA>;
  static bool attr_set_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!attr_set_dev[dev & 63]) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set_dev[dev & 63] = true;
  }
  hipLaunchKernelGGL(k, dim3(p.per_xcd * 8), dim3(64 * WM * WN), (size_t)lds, st, p);
}

// launches a one-block-per-CU kernel; the caller has checked: terms 22 or 2, no fused epilogue extras.
// shape 0: 256 x 256 tiles (N % 256 == 0, no prologue); shape 1: 128 x 384 tiles (N % 384 == 0, any prologue)
bool gemm_split_big_launch(GemmSplitArgs p, int terms, int batch, int shape, int pro, hipStream_t st) {
  const int BM = shape ? 128 : 256, BN = shape ? 384 : 256;
  if (p.ldc >= (1ll << 28) || p.lda >= (1ll << 22) || (p.N % BN) != 0 || p.Npad < p.N || (p.Npad % 128) != 0 || (shape == 0 && pro != 0))
    return false;
  if (p.Npad % BN) return false;                         // (a tile's weight rows must exist in the packed image)
  p.mblocks = (p.M + BM - 1) / BM;
  p.nblocks = p.N / BN;
  const int64_t total = (int64_t)p.mblocks * p.nblocks * batch;
  if (total >= (1ll << 30)) return false;
  p.total = (int)total;
  p.per_xcd = (p.total + 7) / 8;
  const bool f16 = terms == 22;
  if (shape == 0) {
    // A/B (env SEA_GEMM_BIG_WAVES, read per call): 4 = four waves of 128 x 128
    const char* e = getenv("SEA_GEMM_BIG_WAVES");
    if (f16 && e && e[0] == '4') {
      big_launch_one<true, 2, 2, 4, 0, 1, 4>(p, st);
      return true;
    }
    // A/B (env SEA_GEMM_WDMA, read per call): 1 = the weights by LDS-DMA
    const char* d = getenv("SEA_GEMM_WDMA");
    if (f16 && d && d[0] == '1') {
      big_launch_one<true, 4, 2, 4, 0, 1, 2, true>(p, st);
      return true;
    }
    if (f16) big_launch_one<true, 4, 2, 4, 0, 1>(p, st); else big_launch_one<false, 4, 2, 4, 0, 1>(p, st);
    return true;
  }
#define SEA_BIG_WIDE(F)                                              \
  do {                                                               \
    if (pro == 0) big_launch_one<F, 2, 4, 3, 0, 2>(p, st);           \
    else if (pro == 1) big_launch_one<F, 2, 4, 3, 1, 1>(p, st);      \
    else if (pro == 2) big_launch_one<F, 2, 4, 3, 2, 2>(p, st);      \
    else big_launch_one<F, 2, 4, 3, 3, 1>(p, st);                    \
  } while (0)
  if (f16) SEA_BIG_WIDE(true); else SEA_BIG_WIDE(false);
#undef SEA_BIG_WIDE
  return true;
}

}  // namespace sea

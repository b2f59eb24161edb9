// Depthwise 7x7 convolution (stride 1, pad 3, one filter per channel), fp32 NCHW: forward and
// backward-data of the ConvNeXt block's spatial mixing (reference convnext_orig.py:55-57).
//
// Why it is here: MIOpen/CK run this layer at 2-4 TF/s (0.5 ms for 8x96x128x128) although it is a pure
// HBM-bound stencil (read x once, write y once: 100 MB -> ~17 us at the measured copy ceiling).  It is
// ~10 % of an APGD step on UperNet-ConvNeXt-T.  backward-data of a stride-1 depthwise convolution is the
// same stencil with the filter flipped, so one kernel serves both directions.
//
// Mapping: one workgroup = one 32x32 output tile of one (image, channel) plane; the 38x38 input patch
// (3-pixel halo) is staged in LDS once (zero padded), each lane produces a 1x4 strip: per filter row it
// reads 10 consecutive floats from LDS (2 x ds_read_b128 + 1 x ds_read_b64) for 28 FMAs.  The 49
// filter taps are wave-uniform (one channel per workgroup) and live in SGPRs.
#include "sea_common.h"

namespace sea {

constexpr int DW_T = 32;          // output tile edge
constexpr int DW_K = 7, DW_P = 3;
constexpr int DW_IN = DW_T + DW_K - 1;  // 38
constexpr int DW_LD = 40;               // LDS row stride in floats (16-byte aligned rows)

template <bool FLIP, bool BIAS>
__global__ __launch_bounds__(256) void dwconv7x7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int C, int H, int W, int tiles_x) {
  __shared__ __attribute__((aligned(16))) float tile[DW_IN * DW_LD];
  const int plane = blockIdx.y;              // b*C + c
  const int c = plane % C;
  const int ty0 = (blockIdx.x / tiles_x) * DW_T, tx0 = (blockIdx.x % tiles_x) * DW_T;
  const float* xp = x + (int64_t)plane * H * W;
  // stage the input patch (zero outside the image)
  for (int i = threadIdx.x; i < DW_IN * DW_IN; i += 256) {
    const int r = i / DW_IN, q = i - r * DW_IN;
    const int gy = ty0 + r - DW_P, gx = tx0 + q - DW_P;
    float v = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = xp[(int64_t)gy * W + gx];
    tile[r * DW_LD + q] = v;
  }
  // filter taps: uniform per workgroup -> scalar loads
  float wt[DW_K * DW_K];
#pragma unroll
  for (int i = 0; i < DW_K * DW_K; ++i) wt[i] = w[c * DW_K * DW_K + (FLIP ? (DW_K * DW_K - 1 - i) : i)];
  __syncthreads();

  const int ly = threadIdx.x >> 3, lx = (threadIdx.x & 7) * 4;  // 32 rows x 8 strips of 4
  float acc[4];
  const float b0 = BIAS ? bias[c] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = b0;
#pragma unroll
  for (int ky = 0; ky < DW_K; ++ky) {
    const float* row = tile + (ly + ky) * DW_LD + lx;
    const float4 a = *reinterpret_cast<const float4*>(row);
    const float4 b = *reinterpret_cast<const float4*>(row + 4);
    const float2 d = *reinterpret_cast<const float2*>(row + 8);
    const float in[10] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y};
#pragma unroll
    for (int kx = 0; kx < DW_K; ++kx) {
      const float wv = wt[ky * DW_K + kx];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(in[j + kx], wv, acc[j]);
    }
  }
  const int gy = ty0 + ly, gx = tx0 + lx;
  if (gy < H) {
    float* yp = y + (int64_t)plane * H * W + (int64_t)gy * W + gx;
    if (gx + 3 < W && ((((uintptr_t)yp) & 15) == 0)) {
      *reinterpret_cast<float4*>(yp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (gx + j < W) yp[j] = acc[j];
    }
  }
}

// ---- channels_last variant -----------------------------------------------------------------------------
// x, y: (B, H, W, C) contiguous (what PyTorch calls channels_last for an NCHW-shaped tensor); wt: taps-major
// (49, C).  One lane owns 4 consecutive channels (16-byte accesses, lanes contiguous along C) and a strip of
// 8 output pixels of one row: per filter row it loads the 14 input pixels of the strip once and reuses them
// for the 7 horizontal taps (8x7x4 FMAs per 14+7 loads).  Vertical reuse (7 output rows read the same
// input row) is served by L1/L2.  No layout change anywhere in the ConvNeXt block: LayerNorm and the
// pointwise MLP consume the NHWC result as is.
constexpr int DWN_STRIP = 8;

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

//
// Block order: a 1-D grid whose linear id is dealt round-robin to the 8 XCDs by the hardware.  Seven output rows read
// the same input row, so neighbouring rows must meet in the SAME L2: block `id` works on unit (id % 8) * per_xcd + id / 8
// of the (image, row, strip group) linearisation, i.e. every XCD sweeps its own contiguous band of rows (one image per
// XCD at B = 8) and fetches it from the fabric once instead of all eight XCDs fetching everything.
template <bool FLIP, bool BIAS, bool ADD>
__global__ __launch_bounds__(256) void dwconv7x7_nhwc_kernel(const float4* __restrict__ x, const float4* __restrict__ wt,
                                                             const float4* __restrict__ bias,
                                                             const float4* __restrict__ addend, float4* __restrict__ y,
                                                             int CG, int H, int W, int strips_per_block, int gx, int units,
                                                             int per_xcd) {
  const int cg = threadIdx.x % CG, ps = threadIdx.x / CG;
  if (ps >= strips_per_block) return;
  const int u = per_xcd ? (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  if (u >= units) return;
  const int bx = u % gx, oy = (u / gx) % H, b = u / (gx * H);
  const int ox0 = (bx * strips_per_block + ps) * DWN_STRIP;
  if (ox0 >= W) return;
  float4 acc[DWN_STRIP];
  const float4 b0 = BIAS ? bias[cg] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < DWN_STRIP; ++j) acc[j] = b0;
  const float4* xb = x + (int64_t)b * H * W * CG;
#pragma unroll 1
  for (int ky = 0; ky < DW_K; ++ky) {
    const int iy = oy + ky - DW_P;
    if (iy < 0 || iy >= H) continue;
    const float4* row = xb + (int64_t)iy * W * CG + cg;
    float4 in[DWN_STRIP + DW_K - 1];
#pragma unroll
    for (int j = 0; j < DWN_STRIP + DW_K - 1; ++j) {
      const int ix = ox0 + j - DW_P;
      in[j] = (ix >= 0 && ix < W) ? row[(int64_t)ix * CG] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int kx = 0; kx < DW_K; ++kx) {
      const int tap = ky * DW_K + kx;
      const float4 wv = wt[(FLIP ? (DW_K * DW_K - 1 - tap) : tap) * CG + cg];
#pragma unroll
      for (int j = 0; j < DWN_STRIP; ++j) {
        acc[j].x = fmaf(in[j + kx].x, wv.x, acc[j].x);
        acc[j].y = fmaf(in[j + kx].y, wv.y, acc[j].y);
        acc[j].z = fmaf(in[j + kx].z, wv.z, acc[j].z);
        acc[j].w = fmaf(in[j + kx].w, wv.w, acc[j].w);
      }
    }
  }
  const int64_t ro = ((int64_t)b * H + oy) * W * CG + cg;
  float4* yr = y + ro;
#pragma unroll
  for (int j = 0; j < DWN_STRIP; ++j)
    if (ox0 + j < W) {
      if (ADD) acc[j] = add4(acc[j], addend[ro + (int64_t)(ox0 + j) * CG]);  // y = conv + addend (added last)
      yr[(int64_t)(ox0 + j) * CG] = acc[j];
    }
}

// Two output rows per lane: input row r feeds output row oy0 with filter row ky = r and output row oy0 + 1 with
// ky = r - 1, so every input row (14 loads) is loaded once for both: 210 loads per 16
// output float4 instead of 294 (the single-row kernel is bound by the L1 issue rate: 18 loads per output).
// Accumulation order per output is unchanged (bias, then ky ascending, kx ascending): identical bits.
template <bool FLIP, bool BIAS, bool ADD>
__global__ __launch_bounds__(256, 2) void dwconv7x7_nhwc_2row_kernel(const float4* __restrict__ x,
                                                                     const float4* __restrict__ wt,
                                                                     const float4* __restrict__ bias,
                                                                     const float4* __restrict__ addend,
                                                                     float4* __restrict__ y, int CG, int H, int W,
                                                                     int strips_per_block, int gx, int HP, int units,
                                                                     int per_xcd) {
  const int cg = threadIdx.x % CG, ps = threadIdx.x / CG;
  if (ps >= strips_per_block) return;
  const int u = per_xcd ? (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  if (u >= units) return;
  const int bx = u % gx, op = (u / gx) % HP, b = u / (gx * HP);
  const int oy0 = op * 2;
  const int ox0 = (bx * strips_per_block + ps) * DWN_STRIP;
  if (ox0 >= W) return;
  float4 acc0[DWN_STRIP], acc1[DWN_STRIP];
  const float4 b0 = BIAS ? bias[cg] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < DWN_STRIP; ++j) acc0[j] = acc1[j] = b0;
  const float4* xb = x + (int64_t)b * H * W * CG + cg;
  const float4* wb = wt + cg;

  auto load_filter_row = [&](int ky, float4* w) {
#pragma unroll
    for (int kx = 0; kx < DW_K; ++kx) {
      const int tap = ky * DW_K + kx;
      w[kx] = wb[(FLIP ? (DW_K * DW_K - 1 - tap) : tap) * CG];
    }
  };
  auto load_input_row = [&](int iy, float4* in) {
    const bool row_ok = iy >= 0 && iy < H;
    const float4* row = xb + (int64_t)(row_ok ? iy : 0) * W * CG;
#pragma unroll
    for (int j = 0; j < DWN_STRIP + DW_K - 1; ++j) {
      const int ixp = ox0 + j - DW_P;
      in[j] = (row_ok && ixp >= 0 && ixp < W) ? row[(int64_t)ixp * CG] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto accumulate = [&](const float4* in, const float4* w, float4* acc) {
#pragma unroll
    for (int kx = 0; kx < DW_K; ++kx)
#pragma unroll
      for (int j = 0; j < DWN_STRIP; ++j) {
        acc[j].x = fmaf(in[j + kx].x, w[kx].x, acc[j].x);
        acc[j].y = fmaf(in[j + kx].y, w[kx].y, acc[j].y);
        acc[j].z = fmaf(in[j + kx].z, w[kx].z, acc[j].z);
        acc[j].w = fmaf(in[j + kx].w, w[kx].w, acc[j].w);
      }
  };
  // r = 0 .. 7: input row oy0 - 3 + r; rows outside the image contribute zeros (and are not loaded).  One filter row
  // is live at a time (holding both rows of a step costs 28 more registers than the 256 budget of 2 waves/SIMD allows;
  // the second load of a filter row is an L1 hit).
#pragma unroll 1
  for (int r = 0; r < DW_K + 1; ++r) {
    float4 in[DWN_STRIP + DW_K - 1];
    float4 w[DW_K];
    load_input_row(oy0 - DW_P + r, in);
    if (r < DW_K) {
      load_filter_row(r, w);
      accumulate(in, w, acc0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (r >= 1) {
      load_filter_row(r - 1, w);
      accumulate(in, w, acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const int64_t ro = ((int64_t)b * H + oy0) * W * CG + cg;
  float4* yr = y + ro;
#pragma unroll
  for (int j = 0; j < DWN_STRIP; ++j)
    if (ox0 + j < W) {
      if (ADD) acc0[j] = add4(acc0[j], addend[ro + (int64_t)(ox0 + j) * CG]);  // y = conv + addend (added last)
      yr[(int64_t)(ox0 + j) * CG] = acc0[j];
    }
  if (oy0 + 1 < H) {
#pragma unroll
    for (int j = 0; j < DWN_STRIP; ++j)
      if (ox0 + j < W) {
        if (ADD) acc1[j] = add4(acc1[j], addend[ro + ((int64_t)W + ox0 + j) * CG]);
        yr[((int64_t)W + ox0 + j) * CG] = acc1[j];
      }
  }
}

// ---- software-pipelined variants for the ConvNeXt widths (CG = C / 4 a compile-time constant).
// The two kernels above issue a row's loads, wait for them, then run its FMAs: one wave's chain is (latency + FMAs) x rows,
// and with two waves per SIMD (the register budget) little of the latency is covered: 46.6 us for 96 channels at 128 x 128
// where the VALU floor is 15 us and the HBM floor 17 us; 16 us for 384 channels at 32 x 32 (floors 2.5 / 4.4 us).  Here the
// loop is unrolled over the filter rows with TWO input-row buffers and TWO filter-row buffers: the next input row is requested
// before the current one is consumed, the next filter row as soon as its registers fall free.  Borders cost nothing: every
// input row gets its own buffer descriptor (num_records = the row's bytes, or 0 for a row outside the image), so a pixel
// right of the row or a whole missing row reads as zeros; only the three pixels LEFT of a row's first strip need a select on
// the lane offset.  Same accumulation order per output as the kernels above (bias; ky ascending; kx ascending): same bits.
typedef float dw_f4 __attribute__((ext_vector_type(4)));

// WLDS: the block copies the 49 x CG filter into LDS first (flipped for backward-data) and takes its filter rows from there:
// they are a third of the loads (7 next to 14 per row), the same for every strip, and the vector L1 path (64 B/clk per CU)
// is the tighter bound of this kernel next to the VALU (CG <= 96: 18 / 37 / 74 KB, two blocks per CU).
template <int CG, int ROWS, bool FLIP, bool BIAS, bool ADD, bool WLDS>
__global__ __launch_bounds__(256, 2) void dwconv7x7_nhwc_pipe_kernel(const float4* __restrict__ x, const float4* __restrict__ wt,
                                                                     const float4* __restrict__ bias,
                                                                     const float4* __restrict__ addend, float4* __restrict__ y,
                                                                     int H, int W, int gx, int HP, int units, int per_xcd) {
  constexpr int SPB = 256 / CG, NIN = DWN_STRIP + DW_K - 1;
  constexpr uint32_t OOB = 0x40000000u;
  extern __shared__ __attribute__((aligned(16))) char dw_smem[];
  const int cg = threadIdx.x % CG, ps = threadIdx.x / CG;
  const int u = per_xcd ? (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  if (u >= units) return;
  if constexpr (WLDS) {
    for (int i = threadIdx.x; i < DW_K * DW_K * CG; i += 256) {
      const int tap = i / CG, c = i - tap * CG;
      *(float4*)(dw_smem + i * 16) = wt[(FLIP ? (DW_K * DW_K - 1 - tap) : tap) * CG + c];
    }
    __syncthreads();
  }
  if (ps >= SPB) return;
  const int bx = u % gx, op = (u / gx) % HP, b = u / (gx * HP);
  const int oy0 = op * ROWS;
  const int ox0 = (bx * SPB + ps) * DWN_STRIP;
  if (ox0 >= W) return;
  const int row_bytes = W * CG * 16;
  const char* const img = (const char*)(x + (int64_t)b * H * W * CG);
  const uint32_t vbase = (uint32_t)((ox0 * CG + cg) * 16);
  const float4* const wb = wt + cg;

  dw_f4 in[2][NIN], w[2][DW_K];
  auto load_input_row = [&](int iy, dw_f4 (&dst)[NIN]) __attribute__((always_inline)) {
    const bool ok = iy >= 0 && iy < H;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)(ok ? iy : 0) * row_bytes), 0, ok ? row_bytes : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      uint32_t off;
      if (j < DW_P)
        off = ox0 > 0 ? vbase - (uint32_t)((DW_P - j) * CG * 16) : OOB;
      else
        off = vbase + (uint32_t)((j - DW_P) * CG * 16);
      dst[j] = __builtin_bit_cast(dw_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    }
  };
  auto load_filter_row = [&](int ky, dw_f4 (&dst)[DW_K]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < DW_K; ++kx) {
      const int tap = ky * DW_K + kx;
      if constexpr (WLDS)
        dst[kx] = *(const dw_f4*)(dw_smem + (tap * CG + cg) * 16);
      else
        dst[kx] = *(const dw_f4*)(wb + (FLIP ? (DW_K * DW_K - 1 - tap) : tap) * CG);
    }
  };
  auto accumulate = [&](const dw_f4 (&src)[NIN], const dw_f4 (&f)[DW_K], dw_f4 (&acc)[DWN_STRIP]) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < DW_K; ++kx)
#pragma unroll
      for (int j = 0; j < DWN_STRIP; ++j) acc[j] = __builtin_elementwise_fma(src[j + kx], f[kx], acc[j]);
  };

  dw_f4 acc0[DWN_STRIP], acc1[ROWS == 2 ? DWN_STRIP : 1];
  load_input_row(oy0 - DW_P, in[0]);
  load_filter_row(0, w[0]);
  {
    const dw_f4 b0 = BIAS ? *(const dw_f4*)(bias + cg) : dw_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < DWN_STRIP; ++j) acc0[j] = b0;
    if constexpr (ROWS == 2) {
#pragma unroll
      for (int j = 0; j < DWN_STRIP; ++j) acc1[j] = b0;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (ROWS == 2) {
    // r = 0 .. 7: input row oy0 - 3 + r feeds output row oy0 with filter row r and output row oy0 + 1 with filter row r - 1
#pragma unroll
    for (int r = 0; r < DW_K + 1; ++r) {
      const int cur = r & 1;
      if (r < DW_K) load_input_row(oy0 - DW_P + r + 1, in[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (r >= 1) accumulate(in[cur], w[cur ^ 1], acc1);
      __builtin_amdgcn_sched_barrier(0);
      if (r + 1 < DW_K) load_filter_row(r + 1, w[cur ^ 1]);          // (filter row r - 1 is done with)
      __builtin_amdgcn_sched_barrier(0);
      if (r < DW_K) accumulate(in[cur], w[cur], acc0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int r = 0; r < DW_K; ++r) {
      const int cur = r & 1;
      if (r + 1 < DW_K) {
        load_input_row(oy0 - DW_P + r + 1, in[cur ^ 1]);
        load_filter_row(r + 1, w[cur ^ 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      accumulate(in[cur], w[cur], acc0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // (the launcher sends only W % 8 == 0 and, for two rows, even H here: the stores are unconditional.  Guarded stores would
  // let the compiler sink each output's whole FMA chain into its guard's block -- past every load of the loop above)
  const int64_t ro = ((int64_t)b * H + oy0) * W * CG + cg;
  dw_f4* const yr = (dw_f4*)(y + ro);
#pragma unroll
  for (int j = 0; j < DWN_STRIP; ++j) {
    if (ADD) acc0[j] += *(const dw_f4*)(addend + ro + (int64_t)(ox0 + j) * CG);  // y = conv + addend (added last)
    yr[(int64_t)(ox0 + j) * CG] = acc0[j];
  }
  if constexpr (ROWS == 2) {
#pragma unroll
    for (int j = 0; j < DWN_STRIP; ++j) {
      if (ADD) acc1[j] += *(const dw_f4*)(addend + ro + ((int64_t)W + ox0 + j) * CG);
      yr[((int64_t)W + ox0 + j) * CG] = acc1[j];
    }
  }
}

}  // namespace sea

using namespace sea;

template <int CG, int ROWS, bool FLIP, bool BIAS, bool ADD>
static void dw_pipe_launch(dim3 grid, hipStream_t s, const float4* x4, const float4* w4, const float4* b4, const float4* a4,
                           float4* y4, int H, int W, int gx, int HP, int units, int per_xcd, bool wlds) {
  if constexpr (CG <= 96) {
    if (wlds) {
      constexpr int lds = DW_K * DW_K * CG * 16;
      auto k = dwconv7x7_nhwc_pipe_kernel<CG, ROWS, FLIP, BIAS, ADD, true>;
      if constexpr (lds > 65536) {
        static bool attr_set_dev[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!attr_set_dev[dev & 63]) {
          (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          attr_set_dev[dev & 63] = true;
        }
      }
      hipLaunchKernelGGL(k, grid, dim3(256), (size_t)lds, s, x4, w4, b4, a4, y4, H, W, gx, HP, units, per_xcd);
      return;
    }
  }
  hipLaunchKernelGGL((dwconv7x7_nhwc_pipe_kernel<CG, ROWS, FLIP, BIAS, ADD, false>), grid, dim3(256), 0, s, x4, w4, b4, a4, y4, H, W,
                     gx, HP, units, per_xcd);
}

// A/B switches of the NHWC depthwise launcher (development only; unset = shipped dispatch).  Looked up per call (one
// getenv, ~0.1 us next to a 5 us launch) so that one process can compare variants.
static inline int dwconv_ab_switches() {
  const char* e = getenv("SEA_DWCONV_AB");
  return e ? atoi(e) : 0;
}

// x, y: (B,H,W,C) contiguous fp32, C % 4 == 0, C <= 1024; wt: (49, C) taps-major; bias (C) or NULL; addend (same
// shape as y) or NULL: y = conv(x) + addend, added after the taps (bitwise what a separate element-wise add gives; the
// backward of a residual block hands the skip gradient in here).
extern "C" int sea_dwconv7x7_nhwc_add(const float* x, const float* wt, const float* bias, const float* addend, float* y,
                                      int B, int C, int H, int W, int flip, void* stream) {
  SEA_CHECK_ARG(x && wt && y && B > 0 && B <= 65535 && C > 0 && (C % 4) == 0 && C <= 1024 && H > 0 && H <= 65535 && W > 0);
  SEA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)wt) | ((uintptr_t)y) | ((uintptr_t)bias) | ((uintptr_t)addend)) & 15) == 0);
  SEA_CHECK_ARG(!(addend && bias));  // forward: bias; backward-data: addend
  const int CG = C / 4;
  const int spb = 256 / CG;  // strips per block (>= 1 because CG <= 256)
  const int strips = (W + DWN_STRIP - 1) / DWN_STRIP;
  const int gx = (strips + spb - 1) / spb;
  hipStream_t s = (hipStream_t)stream;
  const float4 *x4 = (const float4*)x, *w4 = (const float4*)wt, *b4 = (const float4*)bias, *a4 = (const float4*)addend;
  float4* y4 = (float4*)y;
  // `flip` is a boolean (any non-zero value = backward-data).  A/B switches for devtools/dwconv_bench.py come from the
  // environment, read per call so that one process can compare variants: SEA_DWCONV_AB bit 1 (value 2): plain linear
  // block order; bit 2 (value 4): one output row per lane; bit 3 (value 8): force two rows per lane.
  // Two rows per lane pay off on the large maps (96 ch 128^2: 42 vs 44 us forward, 37 vs 42 us backward-data; 192 ch
  // 64^2: 21 vs 23 us) and lose on the small ones, where halving the number of blocks costs more than the loads saved
  // (768 ch 16^2: 11.2 vs 8.7 us); profiles/r2_dwconv_rows_ab.log.
  const int ab = dwconv_ab_switches();
  const int CG0 = C / 4;
  const bool pipe_ok = !(ab & 16) && (CG0 == 24 || CG0 == 48 || CG0 == 96 || CG0 == 192) && (int64_t)B * H * W * C * 4 < (1ll << 31) &&
                       (W % DWN_STRIP) == 0;
  // (the pipelined kernels keep winning with two rows down to 32 x 32: 12.3 / 10.2 us against 13.7 / 13.4)
  const bool two_rows = !(ab & 4) && H >= 2 && ((ab & 8) || (int64_t)H * W >= (pipe_ok && (H % 2) == 0 ? 1024 : 4096));
  const int HP = two_rows ? (H + 1) / 2 : H;
  const int64_t units64 = (int64_t)gx * HP * B;
  SEA_CHECK_ARG(units64 < (1ll << 30));
  const int units = (int)units64;
  const int per_xcd = (ab & 2) ? 0 : (units + 7) / 8;
  dim3 grid(per_xcd ? per_xcd * 8 : units), block(256);
#define SEA_DW_LAUNCH(F, BI, AD)                                                                                        \
  do {                                                                                                                  \
    if (two_rows)                                                                                                       \
      hipLaunchKernelGGL((dwconv7x7_nhwc_2row_kernel<F, BI, AD>), grid, block, 0, s, x4, w4, b4, a4, y4, CG, H, W, spb,  \
                         gx, HP, units, per_xcd);                                                                       \
    else                                                                                                                \
      hipLaunchKernelGGL((dwconv7x7_nhwc_kernel<F, BI, AD>), grid, block, 0, s, x4, w4, b4, a4, y4, CG, H, W, spb, gx,   \
                         units, per_xcd);                                                                               \
  } while (0)
  // software-pipelined kernels for the ConvNeXt widths (bit 4, value 16, of SEA_DWCONV_AB: the plain kernels instead)
  const bool piped = pipe_ok && (!two_rows || (H % 2) == 0);
#define SEA_DW_PIPE(CGV, F, BI, AD)                                                              \
  do {                                                                                           \
    if (two_rows)                                                                                \
      dw_pipe_launch<CGV, 2, F, BI, AD>(grid, s, x4, w4, b4, a4, y4, H, W, gx, HP, units, per_xcd, wlds); \
    else                                                                                         \
      dw_pipe_launch<CGV, 1, F, BI, AD>(grid, s, x4, w4, b4, a4, y4, H, W, gx, HP, units, per_xcd, wlds); \
  } while (0)
  // filter rows from LDS only for one row per lane (35 vs 40 us at 96 x 128^2; with two rows the copy costs what it saves:
  // profiles/r6_dwconv_pipe_ab.log); bit 5 (value 32) of SEA_DWCONV_AB: never, bit 6 (value 64): always
  const bool wlds = !(ab & 32) && (!two_rows || (ab & 64));
#define SEA_DW_PIPE_CG(F, BI, AD)                       \
  do {                                                  \
    if (CG == 24) SEA_DW_PIPE(24, F, BI, AD);           \
    else if (CG == 48) SEA_DW_PIPE(48, F, BI, AD);      \
    else if (CG == 96) SEA_DW_PIPE(96, F, BI, AD);      \
    else SEA_DW_PIPE(192, F, BI, AD);                   \
  } while (0)
  if (piped) {
    if (flip) {
      if (addend) SEA_DW_PIPE_CG(true, false, true); else SEA_DW_PIPE_CG(true, false, false);
    } else if (bias) {
      SEA_DW_PIPE_CG(false, true, false);
    } else if (addend) {
      SEA_DW_PIPE_CG(false, false, true);
    } else {
      SEA_DW_PIPE_CG(false, false, false);
    }
    SEA_RETURN_LAST();
  }
#undef SEA_DW_PIPE_CG
#undef SEA_DW_PIPE
  if (flip) {
    if (addend)
      SEA_DW_LAUNCH(true, false, true);
    else
      SEA_DW_LAUNCH(true, false, false);
  } else if (bias) {
    SEA_DW_LAUNCH(false, true, false);
  } else if (addend) {
    SEA_DW_LAUNCH(false, false, true);
  } else {
    SEA_DW_LAUNCH(false, false, false);
  }
#undef SEA_DW_LAUNCH
  SEA_RETURN_LAST();
}

// ---- weight and bias gradient of the NHWC depthwise 7x7 (PIR-AT's outer backward, train_rob_seg.py:326-363 through
// convnext_orig.py:55-57):  gw[c,ky,kx] = sum_{b,y,x} x[b, y+ky-3, x+kx-3, c] gy[b,y,x,c],  gb[c] = sum gy[b,y,x,c].
// The library's kernel for it (CK batched GEMM "bwd_weight") takes 620 us per layer: 33 layers = 20 ms = 11 % of the bf16
// outer step for 0.6 GFLOP each.  Here: a block owns (image, R rows, <= 64 channel quads); its seven waves take one tap row
// ky each, a lane 4 channels (and every PL-th pixel group of the tile when fewer than 64 quads exist).  Partial sums per block
// go to a workspace (tiles x 50 x C: 49 taps + the bias row), two more launches add the tiles in index order (32 chunks, then
// the chunks): deterministic, unlike an atomic accumulation.
namespace sea {
__global__ __launch_bounds__(448) void dwconv7x7_nhwc_wgrad_partial_kernel(const float4* __restrict__ x,
                                                                           const float4* __restrict__ gy,
                                                                           float4* __restrict__ ws, int CQ, int H, int W,
                                                                           int R, int tiles_per_image) {
  const int lane = threadIdx.x & 63, ky = threadIdx.x >> 6;          // 7 waves: tap row
  const int c0 = blockIdx.y * 64;
  const int CQL = (CQ - c0) < 64 ? (CQ - c0) : 64;                   // channel quads of this block
  const int PL = 64 / CQL;                                           // pixel lanes per wave
  const int cl = lane % CQL, pl = lane / CQL;
  const bool active = pl < PL;
  const int tile = blockIdx.x, b = tile / tiles_per_image, r0 = (tile % tiles_per_image) * R;
  const int r1 = (r0 + R) < H ? (r0 + R) : H;
  const int cq = c0 + cl;
  const float4* xb = x + (int64_t)b * H * W * CQ + cq;
  const float4* gb = gy + (int64_t)b * H * W * CQ + cq;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc[7], accb = zero;
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = zero;
  if (active) {
    // a lane walks groups of FOUR pixels along a row: 4 loads of gy and a sliding window of 10 loads of x feed 112 FMAs
    // (one pixel per step was a chain of memory round trips: 422 us at 128 x 128 x 96)
    const int GW = (W + 3) / 4, ngroups = (r1 - r0) * GW;
    for (int q = pl; q < ngroups; q += PL) {
      const int yy = r0 + q / GW, x0 = (q % GW) * 4;
      const int ys = yy + ky - 3;
      const bool row_ok = ys >= 0 && ys < H;
      const float4* grow = gb + (int64_t)yy * W * CQ;
      const float4* row = xb + (int64_t)(row_ok ? ys : yy) * W * CQ;
      float4 g[4], v[10];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = (x0 + j < W) ? grow[(int64_t)(x0 + j) * CQ] : zero;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int xs = x0 + j - 3;
        v[j] = (row_ok && xs >= 0 && xs < W) ? row[(int64_t)xs * CQ] : zero;
      }
      if (ky == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) accb.x += g[j].x, accb.y += g[j].y, accb.z += g[j].z, accb.w += g[j].w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          acc[k].x = fmaf(v[j + k].x, g[j].x, acc[k].x), acc[k].y = fmaf(v[j + k].y, g[j].y, acc[k].y);
          acc[k].z = fmaf(v[j + k].z, g[j].z, acc[k].z), acc[k].w = fmaf(v[j + k].w, g[j].w, acc[k].w);
        }
    }
  }
  // pixel lanes of a channel quad (lanes cl + s * CQL) are added into lane cl in the order s = 1, 2, ...
  for (int s2 = 1; s2 < PL; ++s2) {
    const int src = cl + s2 * CQL;
    const bool take = pl == 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float tx = __shfl(acc[k].x, src, 64), ty = __shfl(acc[k].y, src, 64), tz = __shfl(acc[k].z, src, 64),
                  tw = __shfl(acc[k].w, src, 64);
      if (take) acc[k].x += tx, acc[k].y += ty, acc[k].z += tz, acc[k].w += tw;
    }
    const float tx = __shfl(accb.x, src, 64), ty = __shfl(accb.y, src, 64), tz = __shfl(accb.z, src, 64),
                tw = __shfl(accb.w, src, 64);
    if (take) accb.x += tx, accb.y += ty, accb.z += tz, accb.w += tw;
  }
  if (lane < CQL) {
    float4* wt = ws + (int64_t)tile * 50 * CQ + cq;
#pragma unroll
    for (int k = 0; k < 7; ++k) wt[(int64_t)(ky * 7 + k) * CQ] = acc[k];
    if (ky == 0) wt[(int64_t)49 * CQ] = accb;
  }
}

// level 1: chunk z of the tiles -> ws2[z][50 C]; level 2 (chunks == 1 launch): gw (C,7,7) and gb (C).  Fixed order both times.
__global__ __launch_bounds__(256) void dwconv7x7_nhwc_wgrad_reduce_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                          float* __restrict__ gw, float* __restrict__ gb, int C,
                                                                          int n, int per_chunk) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // (t, c), c fastest
  if (idx >= 50 * C) return;
  const int i0 = blockIdx.y * per_chunk, i1 = (i0 + per_chunk) < n ? (i0 + per_chunk) : n;
  float s = 0.f;
#pragma unroll 8
  for (int i = i0; i < i1; ++i) s += src[(int64_t)i * 50 * C + idx];
  if (dst != nullptr) {
    dst[(int64_t)blockIdx.y * 50 * C + idx] = s;
    return;
  }
  const int t = idx / C, c = idx - t * C;
  if (t < 49)
    gw[c * 49 + t] = s;
  else if (gb != nullptr)
    gb[c] = s;
}
}  // namespace sea

// rows per tile: as many blocks as the chip holds at once (4 per CU), the work of a block being a chain of round trips
static inline int dw_wgrad_rows(int B, int C, int H) {
  const int chunks = (C / 4 + 63) / 64;
  int R = (int)(((int64_t)B * H * chunks) / 1024);
  return R < 1 ? 1 : (R > 8 ? 8 : R);
}
static inline int dw_wgrad_chunks(int tiles) { return tiles >= 64 ? 32 : 1; }

// floats of workspace for sea_dwconv7x7_nhwc_wgrad
extern "C" int64_t sea_dwconv7x7_nhwc_wgrad_workspace(int B, int C, int H) {
  if (B <= 0 || C <= 0 || H <= 0) return 0;
  const int R = dw_wgrad_rows(B, C, H);
  const int64_t tiles = (int64_t)B * ((H + R - 1) / R);
  return (tiles + 32) * 50 * C;
}

// x, gy (B,H,W,C) NHWC fp32 dense -> gw (C,7,7) [= the (C,1,7,7) weight gradient], gb (C) or NULL; ws: workspace floats
extern "C" int sea_dwconv7x7_nhwc_wgrad(const float* x, const float* gy, float* gw, float* gb, float* ws, int B, int C,
                                        int H, int W, void* stream) {
  SEA_CHECK_ARG(x && gy && gw && ws && B > 0 && C > 0 && (C % 4) == 0 && H > 0 && W > 0);
  SEA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)ws)) & 15) == 0);
  const int R = dw_wgrad_rows(B, C, H), tpi = (H + R - 1) / R;
  SEA_CHECK_ARG((int64_t)B * tpi < (1ll << 31));
  const int tiles = B * tpi, chunks = dw_wgrad_chunks(tiles), per = (tiles + chunks - 1) / chunks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dwconv7x7_nhwc_wgrad_partial_kernel, dim3(tiles, (C / 4 + 63) / 64), dim3(448), 0, st, (const float4*)x,
                     (const float4*)gy, (float4*)ws, C / 4, H, W, R, tpi);
  const dim3 g1((50 * C + 255) / 256, chunks), g2((50 * C + 255) / 256, 1);
  if (chunks > 1) {
    float* ws2 = ws + (int64_t)tiles * 50 * C;
    hipLaunchKernelGGL(dwconv7x7_nhwc_wgrad_reduce_kernel, g1, dim3(256), 0, st, (const float*)ws, ws2, (float*)nullptr,
                       (float*)nullptr, C, tiles, per);
    hipLaunchKernelGGL(dwconv7x7_nhwc_wgrad_reduce_kernel, g2, dim3(256), 0, st, (const float*)ws2, (float*)nullptr, gw, gb, C,
                       chunks, chunks);
  } else {
    hipLaunchKernelGGL(dwconv7x7_nhwc_wgrad_reduce_kernel, g2, dim3(256), 0, st, (const float*)ws, (float*)nullptr, gw, gb, C, tiles,
                       tiles);
  }
  SEA_RETURN_LAST();
}

extern "C" int sea_dwconv7x7_nhwc(const float* x, const float* wt, const float* bias, float* y, int B, int C, int H,
                                  int W, int flip, void* stream) {
  return sea_dwconv7x7_nhwc_add(x, wt, bias, nullptr, y, B, C, H, W, flip, stream);
}

// x, y: (planes = B*C, H, W) contiguous fp32; w: (C,1,7,7); bias: (C) or NULL.
// flip=0: forward cross-correlation (F.conv2d semantics); flip=1: backward-data (pass dy as x).
extern "C" int sea_dwconv7x7(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W,
                             int flip, void* stream) {
  SEA_CHECK_ARG(x && w && y && B > 0 && C > 0 && H > 0 && W > 0);
  const int64_t planes = (int64_t)B * C;
  SEA_CHECK_ARG(planes <= 65535 * 16);
  const int tiles_x = (W + DW_T - 1) / DW_T, tiles_y = (H + DW_T - 1) / DW_T;
  SEA_CHECK_ARG(planes <= 2147483647 / 1 && (int64_t)tiles_x * tiles_y <= 2147483647);
  // grid.y is limited to 65535: fold planes beyond that into several launches
  hipStream_t s = (hipStream_t)stream;
  const int64_t chunk = 65535;
  for (int64_t p0 = 0; p0 < planes; p0 += chunk) {
    const int np = (int)((planes - p0) < chunk ? (planes - p0) : chunk);
    // plane index p0+blockIdx.y must map to channel (p0 + y) % C: keep p0 a multiple of C
    SEA_CHECK_ARG(p0 % C == 0 || planes <= chunk);
    dim3 grid(tiles_x * tiles_y, np);
    const float* xs = x + p0 * H * W;
    float* ys = y + p0 * H * W;
    if (flip)
      hipLaunchKernelGGL((dwconv7x7_kernel<true, false>), grid, dim3(256), 0, s, xs, w, (const float*)nullptr, ys, C, H, W, tiles_x);
    else if (bias)
      hipLaunchKernelGGL((dwconv7x7_kernel<false, true>), grid, dim3(256), 0, s, xs, w, bias, ys, C, H, W, tiles_x);
    else
      hipLaunchKernelGGL((dwconv7x7_kernel<false, false>), grid, dim3(256), 0, s, xs, w, (const float*)nullptr, ys, C, H, W, tiles_x);
  }
  SEA_RETURN_LAST();
}

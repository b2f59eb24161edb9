// Shared pieces of the K2 translation units (loss_kernels.hip: register-resident and layout variants + the C ABI;
// loss_stream.hip: streaming no-gradient kernel and packed 16-bit gradient kernel).
#pragma once
#include "sea_common.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace sea {

constexpr float kLn2 = 0.69314718055994530942f;

struct __attribute__((aligned(16))) BlockPartial {
  float loss, track;
  int n_correct, pad;
};

// ---- element conversion -------------------------------------------------------------------
// logits travel as raw bits (float, or 16-bit patterns for bf16/f16) so that vector loads/stores can
// use address-space-qualified ext-vector types.
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  using raw = float;
  static __device__ __forceinline__ float to_f(raw v) { return v; }
  static __device__ __forceinline__ raw from_f(float v) { return v; }
};
template <>
struct Elem<__hip_bfloat16> {
  using raw = unsigned short;
  static __device__ __forceinline__ float to_f(raw v) { return __uint_as_float(((unsigned int)v) << 16); }
  static __device__ __forceinline__ raw from_f(float v) {
    const __hip_bfloat16 h = __float2bfloat16(v);  // round-to-nearest-even, NaN preserving
    return __builtin_bit_cast(unsigned short, h);
  }
};
template <>
struct Elem<__half> {
  using raw = unsigned short;
  static __device__ __forceinline__ float to_f(raw v) { return __half2float(__ushort_as_half(v)); }
  static __device__ __forceinline__ raw from_f(float v) { return __half_as_ushort(__float2half(v)); }
};

template <typename R, int VEC>
struct RawVec {
  typedef R type __attribute__((ext_vector_type(VEC)));
};
template <typename R>
struct RawVec<R, 1> {
  typedef R type;
};
template <typename R, int VEC>
__device__ __forceinline__ R vec_get(const typename RawVec<R, VEC>::type& p, int v) {
  if constexpr (VEC == 1)
    return p;
  else
    return p[v];
}
template <typename R, int VEC>
__device__ __forceinline__ void vec_set(typename RawVec<R, VEC>::type& p, int v, R x) {
  if constexpr (VEC == 1)
    p = x;
  else
    p[v] = x;
}

template <typename T>
using gptr = const __attribute__((address_space(1))) T*;
template <typename T>
using gptr_w = __attribute__((address_space(1))) T*;

// per-pixel loss value for a mode; ce = lse - z_y, logp = z_y - lse (<= 0)
__device__ __forceinline__ float loss_value(int mode, bool valid, bool correct, float ce, float logp, float py,
                                            float l1p, float wy) {
  switch (mode) {
    case SEA_MODE_MASK_CE: return correct ? ce : 0.f;
    case SEA_MODE_MASK_CE_BAL: return correct ? wy * ce : 0.f;
    case SEA_MODE_JS: return valid ? (kLn2 + 0.5f * (py * logp - (1.f + py) * l1p)) : 0.f;
    default: return valid ? ce : 0.f;
  }
}

// gradient coefficient K: d loss / d z_c = K * (p_c - [c == y])   
__device__ __forceinline__ float grad_coef(int mode, bool valid, bool correct, float logp, float py, float l1p,
                                           float wy) {
  if (mode == SEA_MODE_JS) return valid ? (-0.5f * (logp - l1p) * py) : 0.f;
  if (mode == SEA_MODE_CE) return valid ? 1.f : 0.f;
  return correct ? wy : 0.f;
}

// block reduction of the three per-thread sums and record write (fixed order, deterministic).
// Workspace layout: record 0 is a header {tiles per image, images, 0, 0} written by block (0,0); the
// per-block records follow, image-major.  The header lets the consumer (loss_finalize or the APGD
// bookkeeping kernel K7) find its way without the host knowing which tiling the dispatcher chose.
__device__ __forceinline__ void block_reduce_store(float ls, float ts, int nc, BlockPartial* ws) {
  __shared__ float s_l[4], s_t[4];
  __shared__ int s_n[4];
  ls = wave_sum(ls);
  ts = wave_sum(ts);
  nc = wave_sum_i(nc);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    s_l[wave] = ls;
    s_t[wave] = ts;
    s_n[wave] = nc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    BlockPartial p;
    p.loss = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
    p.track = (s_t[0] + s_t[1]) + (s_t[2] + s_t[3]);
    p.n_correct = s_n[0] + s_n[1] + s_n[2] + s_n[3];
    p.pad = 0;
    ws[1 + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = p;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
      BlockPartial hdr;
      hdr.loss = 0.f;
      hdr.track = 0.f;
      hdr.n_correct = (int)gridDim.x;  // tiles per image
      hdr.pad = (int)gridDim.y;        // images
      ws[0] = hdr;
    }
  }
}

struct LossArgs {
  const void* logits;
  const void* y;
  int y_bytes;
  const float* w;
  int mode, track_mode, B, C;
  int64_t HW;
  float gscale;
  void* dlogits;
  void* pred;
  int pred_bytes;
  float* loss_px;
  BlockPartial* partials;
  hipStream_t s;
  int force_vec;
};

static inline int tiles_for(int64_t HW, int vec) { return (int)((HW + 256 * vec - 1) / (256 * (int64_t)vec)); }
// workspace is sized for the smallest tile (VEC=1)
static inline int max_tiles(int64_t HW) { return 2 * tiles_for(HW, 1); }  // loss_split.hip: 128 pixels per block


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- 32-bit word <-> float ------------------------------------------------------------------------------------
// PPW = pixels per word: 1 (fp32) or 2 (bf16 / f16: low half = even pixel).
template <typename T>
struct Word;
template <>
struct Word<float> {
  static constexpr int PPW = 1;
  static __device__ __forceinline__ float get(uint32_t w, int) { return __uint_as_float(w); }
  static __device__ __forceinline__ uint32_t pack(float a, float) { return __float_as_uint(a); }
  static __device__ __forceinline__ uint32_t neg_inf() { return 0xff800000u; }
};
template <>
struct Word<__hip_bfloat16> {
  static constexpr int PPW = 2;
  static __device__ __forceinline__ float get(uint32_t w, int h) {
    return __uint_as_float(h ? (w & 0xffff0000u) : (w << 16));
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    return (uint32_t)Elem<__hip_bfloat16>::from_f(a) | ((uint32_t)Elem<__hip_bfloat16>::from_f(b) << 16);
  }
  static __device__ __forceinline__ uint32_t neg_inf() { return 0xff80ff80u; }
};
template <>
struct Word<__half> {
  static constexpr int PPW = 2;
  static __device__ __forceinline__ float get(uint32_t w, int h) {
    return Elem<__half>::to_f((unsigned short)(h ? (w >> 16) : (w & 0xffffu)));
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    return (uint32_t)Elem<__half>::from_f(a) | ((uint32_t)Elem<__half>::from_f(b) << 16);
  }
  static __device__ __forceinline__ uint32_t neg_inf() { return 0xfc00fc00u; }
};

// NW words per lane and plane: global load / store of 4, 2 or 1 dwords
template <int NW>
struct WordVec;
template <>
struct WordVec<4> {
  typedef uint32_t type __attribute__((ext_vector_type(4)));
};
template <>
struct WordVec<2> {
  typedef uint32_t type __attribute__((ext_vector_type(2)));
};
template <>
struct WordVec<1> {
  typedef uint32_t type;
};
template <int NW>
__device__ __forceinline__ void load_words(gptr<char> p, uint32_t (&w)[NW]) {
  using V = typename WordVec<NW>::type;
  const V v = __builtin_nontemporal_load(reinterpret_cast<gptr<V>>(p));
  if constexpr (NW == 1) {
    w[0] = v;
  } else {
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = v[k];
  }
}
template <int NW>
__device__ __forceinline__ void store_words(gptr_w<char> p, const uint32_t (&w)[NW]) {
  using V = typename WordVec<NW>::type;
  V v;
  if constexpr (NW == 1) {
    v = w[0];
  } else {
#pragma unroll
    for (int k = 0; k < NW; ++k) v[k] = w[k];
  }
  __builtin_nontemporal_store(v, reinterpret_cast<gptr_w<V>>(p));
}
__device__ __forceinline__ void fence_word(uint32_t& w) { asm volatile("" : "+v"(w)); }

// loss_stream.hip
template <typename T>
void launch_fwd(const LossArgs& a, int variant);
// loss_split.hip: ADE-sized class vectors (C = 150 / 151) split over the two halves of a wave
template <typename T>
bool dispatch_split(const LossArgs& a, int* tiles_used);

}  // namespace sea

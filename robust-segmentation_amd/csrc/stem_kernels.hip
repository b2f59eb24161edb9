// M9 (model side): the convolutional stem of the robust ConvNeXt backbones ("CVST": semseg/models/backbones/convnext_orig.py:17-38
//   Conv2d(3, 48, 3, stride 2, pad 1) -> LayerNorm(channels_first, eps 1e-6) -> GELU -> Conv2d(48, 96, 3, stride 2, pad 1)
//   -> LayerNorm(channels_first) -> GELU), forward and input gradient, for frozen parameters.
//
// The stem is 0.3 % of the model's flops and was 5 % of an attack step: a 3-channel convolution is nobody's fast path
// (CK grouped forward 128 us, an implicit-GEMM backward-data kernel 151 us for 1.4 GFLOP each), and a channels_first
// LayerNorm went NCHW -> NHWC -> LayerNorm -> NCHW -> GELU in five launches.  Here:
//   stem_conv1_ln_gelu : one lane = one output pixel; the 27 inputs and all CO = 48 output channels live in registers, so
//                        LayerNorm (statistics over the lane's own 48 values) and GELU cost no traffic: reads x (25 MB at
//                        8 x 512^2), writes the pre-LayerNorm y (kept for the backward) and a = GELU(LN(y)): 225 MB.
//   ln_gelu_cl_fwd/bwd : LayerNorm over the channels of an NHWC tensor + GELU, and its input gradient with the statistics
//                        RECOMPUTED from y (nothing saved but y); one lane = one pixel (its C floats are contiguous:
//                        16-byte accesses, statistics inside the lane); the activation side (a / da) is NHWC or NCHW.
//   stem_conv1_bwd     : one lane = a 2 x 2 patch of the input gradient x 3 channels x a quarter of dy's channels; needs dy at
//                        (i, j) ... (i+1, j+1) only (stride 2: an even row meets tap 1, an odd row taps 0 and 2).
// Layouts: the image and its gradient are NCHW (the attack's tensors); everything after the first convolution is NHWC
// ("channels_last"): that is where the library's kernels for the 48 -> 96 convolution are fast (CK forward 110-128 us,
// implicit-GEMM backward 132-151 us; its NCHW Winograd kernels take 211 / 219 us) and what the trunk's blocks read.  (The
// LayerNorm + GELU kernels can also write / read the activation side NCHW, for callers that need it.)
// Weights are read through uniform (scalar) loads: every lane of every wave uses the same 27 x CO floats.  One work item per
// thread, no grid-stride loop: inside a loop the compiler hoists all 1296 scalar loads out of it and spills them.
// All fp32 FMA in a fixed order: bitwise reproducible, and closer to an exact convolution than a Winograd kernel.
#include "sea_common.h"

namespace sea {

__device__ __forceinline__ float stem_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float stem_gelu_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = __expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

// x (B,3,H,W) NCHW, w (CO,3,3,3), y / a (B,Ho,Wo,CO) NHWC; a may be null (convolution only)
template <int CO>
__global__ __launch_bounds__(256) void stem_conv1_ln_gelu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ y,
                                                                 float* __restrict__ a, int H, int W, int Ho, int Wo,
                                                                 int64_t total, float eps) {
  const int64_t plane_in = (int64_t)H * W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) {
    const int j = (int)(idx % Wo);
    const int64_t r = idx / Wo;
    const int i = (int)(r % Ho);
    const int64_t b = r / Ho;
    const float* xb = x + b * 3 * plane_in;
    float xin[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int Y = 2 * i + ky - 1, X = 2 * j + kx - 1;
          const bool ok = Y >= 0 && Y < H && X >= 0 && X < W;
          xin[c * 9 + ky * 3 + kx] = ok ? xb[c * plane_in + (int64_t)Y * W + X] : 0.f;
        }
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      float s = bias ? bias[o] : 0.f;
#pragma unroll
      for (int t = 0; t < 27; ++t) s = fmaf(w[o * 27 + t], xin[t], s);
      acc[o] = s;
      if ((o & 1) == 1) {   // two channels' 54 scalar weights at a time (all 1296 at once spill the SGPR file into VGPR lanes)
        asm volatile("" : "+v"(acc[o - 1]), "+v"(acc[o]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float4* yp = (float4*)(y + idx * CO);
#pragma unroll
    for (int o = 0; o < CO; o += 4) yp[o / 4] = make_float4(acc[o], acc[o + 1], acc[o + 2], acc[o + 3]);
    if (a != nullptr) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < CO; ++o) s += acc[o];
      const float mu = s * (1.f / CO);
      float q = 0.f;
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        const float d = acc[o] - mu;
        q = fmaf(d, d, q);
      }
      const float rs = rsqrtf(q * (1.f / CO) + eps);
      float4* ap = (float4*)(a + idx * CO);
#pragma unroll
      for (int o = 0; o < CO; o += 4) {
        float4 v;
        v.x = stem_gelu(fmaf((acc[o] - mu) * rs, gamma[o], beta[o]));
        v.y = stem_gelu(fmaf((acc[o + 1] - mu) * rs, gamma[o + 1], beta[o + 1]));
        v.z = stem_gelu(fmaf((acc[o + 2] - mu) * rs, gamma[o + 2], beta[o + 2]));
        v.w = stem_gelu(fmaf((acc[o + 3] - mu) * rs, gamma[o + 3], beta[o + 3]));
        ap[o / 4] = v;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// dy (B,Ho,Wo,CO) NHWC -> dx (B,3,H,W) NCHW.  A block owns 64 patches (rows 2i, 2i+1 x columns 2j, 2j+1 of dx, 3 channels:
// 12 sums); wave q of its four waves takes output channels [12 q, 12 q + 12) of dy for all of them, so that the weights
// are wave-uniform (scalar loads) and a lane fetches everything it needs -- 4 neighbours x 48 bytes -- in ONE burst of 12
// loads.  (A first version looped over channel chunks with all 48 channels per lane: every iteration streamed one 16-byte
// slice of every pixel, i.e. the whole 100 MB tensor, through the caches again: 6 passes, 82 us.)  The four partial sums
// meet in LDS in a fixed order.
template <int CO>
__global__ __launch_bounds__(256) void stem_conv1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx, int H, int W, int Ho, int Wo,
                                                             int64_t total) {
  constexpr int CQ = CO / 4;   // channels per wave
  __shared__ float part[4][12][64];
  const int64_t plane_in = (int64_t)H * W;
  const int tid = threadIdx.x, ln = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t idx = (int64_t)blockIdx.x * 64 + ln;
  const bool live = idx < total;
  const int64_t ix = live ? idx : total - 1;
  const int j = (int)(ix % Wo);
  const int64_t r = ix / Wo;
  const int i = (int)(r % Ho);
  const int64_t b = r / Ho;
  const bool ok_i = i + 1 < Ho, ok_j = j + 1 < Wo, ok_ij = ok_i && ok_j;
  // (a neighbour outside the map is read at p00, a valid address, and replaced by 0)
  const float4* p00 = (const float4*)(dy + ix * CO + q * CQ);
  const float4* p01 = p00 + (ok_j ? CO / 4 : 0);
  const float4* p10 = p00 + (ok_i ? (int64_t)Wo * (CO / 4) : 0);
  const float4* p11 = p00 + (ok_ij ? (int64_t)(Wo + 1) * (CO / 4) : 0);
  float4 v00[CQ / 4], v01[CQ / 4], v10[CQ / 4], v11[CQ / 4];
#pragma unroll
  for (int k = 0; k < CQ / 4; ++k) v00[k] = p00[k], v01[k] = p01[k], v10[k] = p10[k], v11[k] = p11[k];
  float g[3][2][2];
#pragma unroll
  for (int c = 0; c < 3; ++c) g[c][0][0] = g[c][0][1] = g[c][1][0] = g[c][1][1] = 0.f;
  const float* wq = w + q * CQ * 27;
#pragma unroll
  for (int k = 0; k < CQ / 4; ++k) {
    const float d00[4] = {v00[k].x, v00[k].y, v00[k].z, v00[k].w};
    const float d01[4] = {ok_j ? v01[k].x : 0.f, ok_j ? v01[k].y : 0.f, ok_j ? v01[k].z : 0.f, ok_j ? v01[k].w : 0.f};
    const float d10[4] = {ok_i ? v10[k].x : 0.f, ok_i ? v10[k].y : 0.f, ok_i ? v10[k].z : 0.f, ok_i ? v10[k].w : 0.f};
    const float d11[4] = {ok_ij ? v11[k].x : 0.f, ok_ij ? v11[k].y : 0.f, ok_ij ? v11[k].z : 0.f, ok_ij ? v11[k].w : 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* wo = wq + (k * 4 + e) * 27;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* kk = wo + c * 9;  // kk[ky * 3 + kx]
        g[c][0][0] = fmaf(kk[4], d00[e], g[c][0][0]);
        g[c][0][1] = fmaf(kk[5], d00[e], g[c][0][1]);
        g[c][0][1] = fmaf(kk[3], d01[e], g[c][0][1]);
        g[c][1][0] = fmaf(kk[7], d00[e], g[c][1][0]);
        g[c][1][0] = fmaf(kk[1], d10[e], g[c][1][0]);
        g[c][1][1] = fmaf(kk[8], d00[e], g[c][1][1]);
        g[c][1][1] = fmaf(kk[6], d01[e], g[c][1][1]);
        g[c][1][1] = fmaf(kk[2], d10[e], g[c][1][1]);
        g[c][1][1] = fmaf(kk[0], d11[e], g[c][1][1]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) part[q][c * 4 + py * 2 + px][ln] = g[c][py][px];
  __syncthreads();
  // wave q finishes outputs 3 q ... 3 q + 2 of every patch: (channel, row, column) = (o / 4, (o / 2) & 1, o & 1)
  if (live) {
    float* xb = dx + b * 3 * plane_in;
    const int Y = 2 * i, X = 2 * j;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int o = q * 3 + t;
      const float v = ((part[0][o][ln] + part[1][o][ln]) + part[2][o][ln]) + part[3][o][ln];
      const int c = o >> 2, py = (o >> 1) & 1, px = o & 1;
      if (Y + py < H && X + px < W) xb[c * plane_in + (int64_t)(Y + py) * W + X + px] = v;
    }
  }
}

// a = GELU(LN_c(y)) for y (B,HW,C) NHWC; a is NHWC, or NCHW (B,C,HW) when A_NCHW
template <int C, bool A_NCHW>
__global__ __launch_bounds__(256) void ln_gelu_cl_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ a,
                                                             int64_t HW, int64_t total, float eps) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) {
    const float4* yp = (const float4*)(y + idx * C);
    float v[C];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      const float4 t = yp[c / 4];
      v[c] = t.x, v[c + 1] = t.y, v[c + 2] = t.z, v[c + 3] = t.w;
      s += (t.x + t.y) + (t.z + t.w);
    }
    const float mu = s * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float d = v[c] - mu;
      q = fmaf(d, d, q);
    }
    const float rs = rsqrtf(q * (1.f / C) + eps);
    const int64_t b = A_NCHW ? idx / HW : 0, p = A_NCHW ? idx - b * HW : 0;
    float* an = a + (b * C) * HW + p;            // NCHW: channel c at an[c * HW]
    float4* ac = (float4*)(a + idx * C);        // NHWC
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      float4 o;
      o.x = stem_gelu(fmaf((v[c] - mu) * rs, gamma[c], beta[c]));
      o.y = stem_gelu(fmaf((v[c + 1] - mu) * rs, gamma[c + 1], beta[c + 1]));
      o.z = stem_gelu(fmaf((v[c + 2] - mu) * rs, gamma[c + 2], beta[c + 2]));
      o.w = stem_gelu(fmaf((v[c + 3] - mu) * rs, gamma[c + 3], beta[c + 3]));
      if (A_NCHW) {
        an[(int64_t)c * HW] = o.x, an[(int64_t)(c + 1) * HW] = o.y, an[(int64_t)(c + 2) * HW] = o.z, an[(int64_t)(c + 3) * HW] = o.w;
      } else {
        ac[c / 4] = o;
      }
      __builtin_amdgcn_sched_barrier(0);   // (else the erf evaluations are overlapped C wide: 400 VGPRs)
    }
  }
}

// dy = d loss / d y given da = d loss / d a, a = GELU(u), u = xhat * gamma + beta, xhat = (y - mean) * rstd:
//   gw = da * GELU'(u) * gamma;  dy = rstd * (gw - mean_c(gw) - xhat * mean_c(gw * xhat))
// A block owns 128 pixels; waves 0-1 hold channels [0, C/2) of them, waves 2-3 channels [C/2, C): y and gw of a lane's
// channels stay in registers (C values per lane instead of 2 C), the channel half is wave-uniform (gamma / beta stay scalar
// loads) and the four sums meet through LDS in a fixed order.  y, dy NHWC; da NHWC, or NCHW when A_NCHW.
template <int C, bool A_NCHW>
__global__ __launch_bounds__(256) void ln_gelu_cl_bwd_kernel(const float* __restrict__ da, const float* __restrict__ y,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ dy,
                                                             int64_t HW, int64_t total, float eps) {
  constexpr int CL = C / 2;
  __shared__ float red[4][2][128];
  const int tid = threadIdx.x, pl = tid & 127;
  const int half = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int64_t idx = (int64_t)blockIdx.x * 128 + pl;
  const bool live = idx < total;
  const int64_t ix = live ? idx : total - 1;          // (dead lanes compute a valid pixel and do not store)
  const float4* yp = (const float4*)(y + ix * C + half * CL);
  const float* gm = gamma + half * CL;
  const float* bt = beta + half * CL;
  float xh[CL], gw[CL];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CL; c += 4) {
    const float4 t = yp[c / 4];
    xh[c] = t.x, xh[c + 1] = t.y, xh[c + 2] = t.z, xh[c + 3] = t.w;
    s += (t.x + t.y) + (t.z + t.w);
  }
  if (A_NCHW) {
    const int64_t b = ix / HW, p = ix - b * HW;
    const float* gp = da + (b * C + half * CL) * HW + p;
#pragma unroll
    for (int c = 0; c < CL; ++c) {
      gw[c] = *gp;
      gp += HW;
    }
  } else {
    const float4* gp = (const float4*)(da + ix * C + half * CL);
#pragma unroll
    for (int c = 0; c < CL; c += 4) {
      const float4 t = gp[c / 4];
      gw[c] = t.x, gw[c + 1] = t.y, gw[c + 2] = t.z, gw[c + 3] = t.w;
    }
  }
  red[0][half][pl] = s;
  __syncthreads();
  const float mu = (red[0][0][pl] + red[0][1][pl]) * (1.f / C);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CL; ++c) {
    const float d = xh[c] - mu;
    q = fmaf(d, d, q);
  }
  red[1][half][pl] = q;
  __syncthreads();
  const float rs = rsqrtf((red[1][0][pl] + red[1][1][pl]) * (1.f / C) + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < CL; ++c) {
    xh[c] = (xh[c] - mu) * rs;
    const float gmc = gm[c];
    const float u = fmaf(xh[c], gmc, bt[c]);
    gw[c] = gw[c] * stem_gelu_grad(u) * gmc;
    s1 += gw[c];
    s2 = fmaf(gw[c], xh[c], s2);
    // finish this channel before the next one starts: erff branches, and without the empty asm the compiler sinks the rest of
    // the statement below all CL evaluations and keeps three temporaries per channel alive (250 VGPRs at C = 96)
    asm volatile("" : "+v"(gw[c]), "+v"(s1), "+v"(s2));
  }
  red[2][half][pl] = s1;
  red[3][half][pl] = s2;
  __syncthreads();
  const float m1 = (red[2][0][pl] + red[2][1][pl]) * (1.f / C), m2 = (red[3][0][pl] + red[3][1][pl]) * (1.f / C);
  if (live) {
    float4* op = (float4*)(dy + ix * C + half * CL);
#pragma unroll
    for (int c = 0; c < CL; c += 4) {
      float4 o;
      o.x = rs * (gw[c] - m1 - xh[c] * m2);
      o.y = rs * (gw[c + 1] - m1 - xh[c + 1] * m2);
      o.z = rs * (gw[c + 2] - m1 - xh[c + 2] * m2);
      o.w = rs * (gw[c + 3] - m1 - xh[c + 3] * m2);
      op[c / 4] = o;
    }
  }
}

}  // namespace sea

using namespace sea;

// x (B,3,H,W) NCHW fp32 contiguous; w (CO,3,3,3); bias / gamma / beta (CO); y, a (B,Ho,Wo,CO) NHWC with Ho = (H-1)/2+1,
// Wo = (W-1)/2+1.  a == NULL: the convolution alone (gamma / beta unused).  CO = 48.
extern "C" int sea_stem_conv1_ln_gelu(const float* x, const float* w, const float* bias, const float* gamma,
                                      const float* beta, float* y, float* a, int B, int CO, int H, int W, float eps,
                                      void* stream) {
  SEA_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0 && CO == 48);
  SEA_CHECK_ARG(a == nullptr || (gamma && beta));
  SEA_CHECK_ARG(((((uintptr_t)y) | ((uintptr_t)a)) & 15) == 0);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo;
  hipLaunchKernelGGL(stem_conv1_ln_gelu_kernel<48>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, w, bias, gamma, beta, y, a, H, W, Ho, Wo, total, eps);
  SEA_RETURN_LAST();
}

// dy (B,Ho,Wo,CO) NHWC -> dx (B,3,H,W) NCHW: input gradient of the stride-2 3x3 convolution above
extern "C" int sea_stem_conv1_bwd(const float* dy, const float* w, float* dx, int B, int CO, int H, int W, void* stream) {
  SEA_CHECK_ARG(dy && w && dx && B > 0 && H > 0 && W > 0 && CO == 48);
  SEA_CHECK_ARG((((uintptr_t)dy) & 15) == 0);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo;
  hipLaunchKernelGGL(stem_conv1_bwd_kernel<48>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dy, w,
                     dx, H, W, Ho, Wo, total);
  SEA_RETURN_LAST();
}

// a = GELU(LayerNorm over C (y)) for y (B,HW,C) NHWC fp32; a NHWC, or NCHW (B,C,HW) when a_nchw; C = 48 or 96
extern "C" int sea_ln_gelu_cl_fwd(const float* y, const float* gamma, const float* beta, float* a, int a_nchw, int B, int C,
                                  int64_t HW, float eps, void* stream) {
  SEA_CHECK_ARG(y && gamma && beta && a && B > 0 && HW > 0 && (C == 48 || C == 96));
  SEA_CHECK_ARG(((((uintptr_t)y) | ((uintptr_t)a)) & 15) == 0);
  const int64_t total = (int64_t)B * HW;
  const dim3 grid((unsigned)((total + 255) / 256));
#define SEA_LN_GELU_FWD(CC, NC)                                                                                          \
  hipLaunchKernelGGL((ln_gelu_cl_fwd_kernel<CC, NC>), grid, dim3(256), 0, (hipStream_t)stream, y, gamma, beta, a, HW, total, \
                     eps)
  if (C == 48) {
    if (a_nchw) SEA_LN_GELU_FWD(48, true); else SEA_LN_GELU_FWD(48, false);
  } else {
    if (a_nchw) SEA_LN_GELU_FWD(96, true); else SEA_LN_GELU_FWD(96, false);
  }
#undef SEA_LN_GELU_FWD
  SEA_RETURN_LAST();
}

// dy (NHWC) = d loss / d y from da = d loss / d a of the op above (da NHWC, or NCHW when a_nchw; gamma, beta frozen);
// the statistics are recomputed from y
extern "C" int sea_ln_gelu_cl_bwd(const float* da, int a_nchw, const float* y, const float* gamma, const float* beta,
                                  float* dy, int B, int C, int64_t HW, float eps, void* stream) {
  SEA_CHECK_ARG(da && y && gamma && beta && dy && B > 0 && HW > 0 && (C == 48 || C == 96));
  SEA_CHECK_ARG(((((uintptr_t)y) | ((uintptr_t)dy)) & 15) == 0 && (a_nchw || (((uintptr_t)da) & 15) == 0));
  const int64_t total = (int64_t)B * HW;
  const dim3 grid((unsigned)((total + 127) / 128));
#define SEA_LN_GELU_BWD(CC, NC)                                                                                          \
  hipLaunchKernelGGL((ln_gelu_cl_bwd_kernel<CC, NC>), grid, dim3(256), 0, (hipStream_t)stream, da, y, gamma, beta, dy, HW,   \
                     total, eps)
  if (C == 48) {
    if (a_nchw) SEA_LN_GELU_BWD(48, true); else SEA_LN_GELU_BWD(48, false);
  } else {
    if (a_nchw) SEA_LN_GELU_BWD(96, true); else SEA_LN_GELU_BWD(96, false);
  }
#undef SEA_LN_GELU_BWD
  SEA_RETURN_LAST();
}

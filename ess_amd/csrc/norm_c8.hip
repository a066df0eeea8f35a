// InstanceNorm2d / train-mode BatchNorm2d on BF16_C8 tensors: bfloat16 [N][ceil(C/8)][H*W][8] -- the stored form of the
// trainable networks' activations and activation gradients in the bf16 configuration (reference layers:
// models/style_networks.py:163-164,180-182,192 and the torchvision BasicBlock BatchNorm behind :116-121).
//
// A 16-byte load is one pixel of EIGHT channel planes, so every kernel works on an 8-channel block at a time: a lane keeps 8
// running sums, a workgroup reduces 8 planes at once.  All arithmetic is fp32 (statistics: fp64 partial sums in the split
// variants, a register-resident two-pass mean / centred variance in the fused ones); only the tensors are bf16.  Channels
// past C inside the last block are zeros in memory and are written back as zeros.
//   * planes of <= 20 vectors per thread (60x80 with 256 threads, 120x160 with 1024): ONE kernel, x read once and kept in
//     registers between the reduction and the map (2 B read + 2 B written per element);
//   * larger planes and BatchNorm (few channel blocks, reduction over N x H x W): pass 1 writes one fp64 partial per
//     (group, slice, channel) -- no atomics, fixed summation order --, pass 2 totals them and applies the map.
#include <atomic>
#include "common.h"
#include <stdlib.h>

#ifndef ESS_REDUCE1_U
#define ESS_REDUCE1_U 4
#endif
namespace {

typedef unsigned int u32x4n __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void unpack8(u32x4n v, float (&f)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f[2 * q] = __builtin_bit_cast(float, v[q] << 16);
    f[2 * q + 1] = __builtin_bit_cast(float, v[q] & 0xffff0000u);
  }
}
// x (the PRE-norm tensor a convolution wrote) may be an F16_C8 tensor -- the same layout with IEEE half elements: the norm kernels
// are its only readers, and half's 11-bit significand keeps 8x more of a channel whose mean is large against its spread than
// bfloat16 does (ablation in DESIGN.md section 5: the bf16 rounding of the pre-norm tensors was the largest single contribution
// to the logit error of the bf16 configuration).  Flag: bit 8 of the kernels' `relu` argument.
__device__ __forceinline__ void unpack8h(u32x4n v, float (&f)[8]) {
  typedef _Float16 f16x2n __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned w = v[q];  // (ROCm 7.2: a bit_cast applied directly to an ext-vector element reads element 0 for every q)
    const f16x2n h = __builtin_bit_cast(f16x2n, w);
    f[2 * q] = (float)h[0];
    f[2 * q + 1] = (float)h[1];
  }
}
__device__ __forceinline__ void unpack8x(u32x4n v, float (&f)[8], int xf16) {
  if (xf16) unpack8h(v, f);  // (uniform)
  else unpack8(v, f);
}
__device__ __forceinline__ u32x4n pack8n(const float (&f)[8]) {
  bf16x8n b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)f[j];
  return __builtin_bit_cast(u32x4n, b);
}

__device__ __forceinline__ u32x4n pack8hn(const float (&f)[8]) {  // IEEE half, saturating at +-65504 (NaN kept)
  typedef _Float16 f16x8n __attribute__((ext_vector_type(8)));
  f16x8n b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float c = __builtin_amdgcn_fmed3f(f[j], -65504.f, 65504.f);
    b[j] = (_Float16)(f[j] != f[j] ? f[j] : c);
  }
  return __builtin_bit_cast(u32x4n, b);
}

// block-wide sums of 8 floats (blockDim.x a multiple of 64, <= 1024); red: 16 x 8 floats of LDS.  Fixed order.
__device__ __forceinline__ void block_sum8(float (&v)[8], float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = wave_sum(v[j]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[w * 8 + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i * 8 + j];
    v[j] = t;
  }
}

constexpr int MAXV = 20;  // 16-byte vectors a thread of the fused kernels keeps in registers

// The fused kernels keep a plane PACKED (4 VGPRs per 8 elements) between their passes.  Left alone, the compiler hoists the
// bf16 -> fp32 unpacking out of the later passes and keeps all 8 floats per vector alive (2x the registers: spills with 1024
// threads); an opaque copy per use makes every pass unpack again.
__device__ __forceinline__ u32x4n opaque(u32x4n v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ---- InstanceNorm forward, fused: one workgroup per (n, channel block)
// MV_: vectors per thread (0: the default of the thread count).  Small planes (60 x 80: 4800 vectors) run 256 x 20 by default; the
// 512 x 10 / 1024 x 5 forms put 8 / 16 waves on the CU that owns the plane (tuning switch "in_small_threads")
template <int THREADS, int MV_ = 0>
__global__ __launch_bounds__(THREADS) void in_fwd_c8_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ res,
                                                            u32x4n* __restrict__ y, float* __restrict__ stats, int CB, int C,
                                                            int hw, float eps, int relu) {
  const int xf16 = relu >> 8;  // (flags: bit 0 = ReLU, bit 8 = x is an F16_C8 tensor)
  relu &= 0xff;
  __shared__ float red[16 * 8];
  const int g = blockIdx.x, n = g / CB, cb = g - n * CB;
  const size_t base = (size_t)g * hw;
  constexpr int MV = MV_ ? MV_ : (THREADS == 1024 ? 19 : MAXV);  // (16 waves per CU: 128 registers per thread)
  u32x4n xv[MV];
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  // all loads of the plane are issued back to back from clamped (always valid) addresses: a load under a per-lane condition
  // would be waited for inside its branch, one memory round trip per vector
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    xv[k] = x[base + (i < hw ? i : hw - 1)];
  }
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    if (i < hw) {
      float f[8];
      unpack8x(xv[k], f, xf16);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
  }
  block_sum8(s, red);
  float mean[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mean[j] = s[j] / hw; q[j] = 0.f; }
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    if (i < hw) {
      float f[8];
      unpack8x(opaque(xv[k]), f, xf16);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[j] - mean[j]; q[j] += d * d; }
    }
  }
  block_sum8(q, red);
  float rstd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rstd[j] = 1.0f / sqrtf(q[j] / hw + eps);
  if (threadIdx.x < 8 && cb * 8 + (int)threadIdx.x < C) {
    float m = 0.f, r = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == (int)threadIdx.x) { m = mean[j]; r = rstd[j]; }
    stats[2 * ((size_t)n * C + cb * 8 + threadIdx.x)] = m;
    stats[2 * ((size_t)n * C + cb * 8 + threadIdx.x) + 1] = r;
  }
  constexpr int RB = (THREADS == 1024 && MV > 5) ? 1 : 5;  // residual vectors in flight per batch (register budget)
#pragma unroll
  for (int k0 = 0; k0 < MV; k0 += RB) {
    u32x4n rv[RB];
    if (res) {  // (uniform)
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int i = threadIdx.x + (k0 + u) * THREADS;
        if (k0 + u < MV) rv[u] = res[base + (i < hw ? i : hw - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int k = k0 + u, i = threadIdx.x + k * THREADS;
      if (k < MV && i < hw) {
        float f[8], rf[8];
        unpack8x(opaque(xv[k]), f, xf16);
        if (res) unpack8(rv[u], rf);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = (f[j] - mean[j]) * rstd[j];
          if (relu) t = fmaxf(t, 0.f);
          if (res) t += rf[j];
          f[j] = cb * 8 + j < C ? t : 0.f;
        }
        y[base + i] = pack8n(f);
      }
    }
  }
}

// ---- InstanceNorm forward of the "mixed" configuration (round 6; ESS_COMPUTE_F16 consumers): the fused kernel above with
//   * a second output y16 = the result as an F16_C8 tensor (what the next convolution's half matrix cores read: 11 significant bits;
//     y, the BF16_C8 form every other consumer keeps reading -- weight gradient, skip add, losses --, may be NULL);
//   * a residual that may be an F16_C8 tensor (flag bit 9);
//   * HILO: x is a [hi | lo] half pair, [N][2 CB][hw][8] (ESS_FMT_F16_C8_HILO): x = hi + lo, ~22 significant bits -- the first decoder
//     layer's pre-norm tensor, whose channel means are 6-17 standard deviations (the event latents' means): rounding IT to half was the
//     largest single term of the logit error once the operands were half (tools/hybrid_rounding_ablation.py).
template <int THREADS, int MV, bool HILO>
__global__ __launch_bounds__(THREADS) void in_fwd_c8_mix_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ res,
                                                                u32x4n* __restrict__ y, u32x4n* __restrict__ y16, float* __restrict__ stats,
                                                                int CB, int C, int hw, float eps, int flags) {
  const int relu = flags & 1, xf16 = (flags >> 8) & 1, rf16 = (flags >> 9) & 1, rpair = (flags >> 11) & 1;
  __shared__ float red[16 * 8];
  const int g = blockIdx.x, n = g / CB, cb = g - n * CB;
  const size_t base = (size_t)g * hw;
  const size_t xbase = HILO ? ((size_t)n * 2 * CB + cb) * hw : base, xlo = xbase + (size_t)CB * hw;
  const size_t rbase = rpair ? ((size_t)n * 2 * CB + cb) * hw : base;  // (bit 11: the residual is a [hi | lo] half pair, its hi parts are added)
  u32x4n xv[MV], xl[HILO ? MV : 1];
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    xv[k] = x[xbase + (i < hw ? i : hw - 1)];
    if constexpr (HILO) xl[k] = x[xlo + (i < hw ? i : hw - 1)];
  }
  auto value = [&](int k, float (&f)[8]) {
    if constexpr (HILO) {
      float l[8];
      unpack8h(opaque(xv[k]), f);
      unpack8h(opaque(xl[k]), l);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += l[j];
    } else {
      unpack8x(opaque(xv[k]), f, xf16);
    }
  };
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    if (i < hw) {
      float f[8];
      value(k, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
  }
  block_sum8(s, red);
  float mean[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mean[j] = s[j] / hw; q[j] = 0.f; }
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    if (i < hw) {
      float f[8];
      value(k, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[j] - mean[j]; q[j] += d * d; }
    }
  }
  block_sum8(q, red);
  float rstd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rstd[j] = 1.0f / sqrtf(q[j] / hw + eps);
  if (threadIdx.x < 8 && cb * 8 + (int)threadIdx.x < C) {
    float m = 0.f, r = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == (int)threadIdx.x) { m = mean[j]; r = rstd[j]; }
    stats[2 * ((size_t)n * C + cb * 8 + threadIdx.x)] = m;
    stats[2 * ((size_t)n * C + cb * 8 + threadIdx.x) + 1] = r;
  }
  constexpr int RB = 5;
#pragma unroll
  for (int k0 = 0; k0 < MV; k0 += RB) {
    u32x4n rv[RB];
    if (res) {  // (uniform)
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int i = threadIdx.x + (k0 + u) * THREADS;
        if (k0 + u < MV) rv[u] = res[rbase + (i < hw ? i : hw - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int k = k0 + u, i = threadIdx.x + k * THREADS;
      if (k < MV && i < hw) {
        float f[8], rf[8];
        value(k, f);
        if (res) unpack8x(rv[u], rf, rf16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = (f[j] - mean[j]) * rstd[j];
          if (relu) t = fmaxf(t, 0.f);
          if (res) t += rf[j];
          f[j] = cb * 8 + j < C ? t : 0.f;
        }
        if (y) y[base + i] = pack8n(f);
        y16[base + i] = pack8hn(f);
      }
    }
  }
}

// ---- InstanceNorm backward, fused (256 threads: x and dy of a 60x80 block stay in registers)
template <int THREADS = 256, int MV = MAXV>
__global__ __launch_bounds__(THREADS) void in_bwd_c8_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ dy,
                                                            const float* __restrict__ stats, u32x4n* __restrict__ dx, int CB, int C,
                                                            int hw, int relu, int xcb = 0) {
  const int xf16 = (relu >> 8) & 1;  // (flags: bit 0 = ReLU, bit 8 = x is an F16_C8 tensor)
  relu &= 0xff;
  __shared__ float red[16 * 8];
  const int g = blockIdx.x, n = g / CB, cb = g - n * CB;
  const size_t base = (size_t)g * hw;
  // xcb > 0: x keeps xcb blocks per sample of which the first CB are read (the hi parts of a [hi | lo] pre-norm tensor)
  const size_t xbase = xcb > 0 ? ((size_t)n * xcb + cb) * hw : base;
  float mean[8], rstd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j < C ? cb * 8 + j : C - 1;
    mean[j] = stats[2 * ((size_t)n * C + c)];
    rstd[j] = stats[2 * ((size_t)n * C + c) + 1];
  }
  u32x4n xv[MV], gv[MV];
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int k = 0; k < MV; ++k) {  // (unconditional loads from clamped addresses: see in_fwd_c8_kernel)
    const int i = threadIdx.x + k * THREADS;
    xv[k] = x[xbase + (i < hw ? i : hw - 1)];
    gv[k] = dy[base + (i < hw ? i : hw - 1)];
  }
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    if (i < hw) {
      float f[8], gg[8];
      unpack8x(xv[k], f, xf16);
      unpack8(gv[k], gg);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (f[j] - mean[j]) * rstd[j];
        const float gr = (relu && xh <= 0.f) ? 0.f : gg[j];
        s1[j] += gr;
        s2[j] += gr * xh;
      }
    }
  }
  block_sum8(s1, red);
  block_sum8(s2, red);
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] /= hw; s2[j] /= hw; }
#pragma unroll
  for (int k = 0; k < MV; ++k) {
    const int i = threadIdx.x + k * THREADS;
    if (i < hw) {
      float f[8], gg[8];
      unpack8x(opaque(xv[k]), f, xf16);
      unpack8(opaque(gv[k]), gg);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (f[j] - mean[j]) * rstd[j];
        const float gr = (relu && xh <= 0.f) ? 0.f : gg[j];
        f[j] = cb * 8 + j < C ? rstd[j] * (gr - s1[j] - xh * s2[j]) : 0.f;
      }
      dx[base + i] = pack8n(f);
    }
  }
}

// ---- split variants.  Group g = (n, cb) [InstanceNorm: nseg = 1] or cb [BatchNorm: nseg = N, seg_stride = CB];
// segment j of group g = hw vectors at ((j * seg_stride) + g) * hw.
// sums[((g * nsl + sl) * 8 + ch) * 2 + {0, 1}] = (sum f0, sum f1) of the slice, (f0, f1) = (x, x^2) [MODE 0] or (g, g * xhat)
// [MODE 1: backward; the ReLU mask comes from xhat (InstanceNorm: relu(IN(x))) or from the saved output y (BatchNorm)]
template <int MODE>
__global__ __launch_bounds__(256) void c8_reduce_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ y,
                                                        const u32x4n* __restrict__ dy, const float* __restrict__ stats,
                                                        double* sums, int hw, int nseg, int seg_stride, int CB, int C,
                                                        int per_sample_stats, int relu, int relu_from_y,
                                                        const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr, int xcb = 0) {
  // (flags: bit 0 = ReLU, bit 8 = x is an F16_C8 tensor; xcb > 0 (InstanceNorm, nseg = 1): x keeps xcb blocks per sample, a [hi | lo] half
  // pair -- MODE 0 sums hi + lo, MODE 1 reads the hi parts)
  const int xf16 = (relu >> 8) & 1;
  relu &= 0xff;
  __shared__ double redd[4 * 16];
  const int g = blockIdx.x, nsl = gridDim.y, sl = blockIdx.y;
  const int len = (hw + nsl - 1) / nsl;
  const int i0 = sl * len, i1 = min(hw, i0 + len);
  const int cb = g % CB;
  float mean[8], rstd[8], ma[8], mb[8];  // (ma, mb: the forward's affine map y = x ma + mb, for a ReLU mask recomputed from x)
  const bool affine_mask = MODE == 1 && relu && !relu_from_y && gamma != nullptr;
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j < C ? cb * 8 + j : C - 1;
      const size_t p = per_sample_stats ? (size_t)(g / CB) * C + c : (size_t)c;
      mean[j] = stats[2 * p];
      rstd[j] = stats[2 * p + 1];
      ma[j] = mb[j] = 0.f;
      if (affine_mask) {  // (the forward's own map, saved by bn_apply_c8_kernel behind the C (mean, rstd) pairs)
        ma[j] = stats[2 * (C + c)];
        mb[j] = stats[2 * (C + c) + 1];
      }
    }
  }
  double s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0; s1[j] = 0; }
  constexpr int U = MODE == 0 ? 4 : (ESS_REDUCE1_U);  // vectors per tensor in flight per thread
  for (int sgm = 0; sgm < nseg; ++sgm) {
    const size_t base = ((size_t)sgm * seg_stride + g) * hw;
    const size_t xbase = xcb > 0 ? ((size_t)(g / CB) * xcb + cb) * hw : base, xlo = xbase + (size_t)CB * hw;
    for (int i = i0 + threadIdx.x; i < i1; i += 256 * U) {
      u32x4n xv[U], gv[U], yv[U], xl[MODE == 0 ? U : 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {  // clamped addresses: the loads carry no per-lane branch and are all in flight together
        const int ii = i + u * 256, ic = ii < i1 ? ii : i1 - 1;
        xv[u] = x[xbase + ic];
        if (MODE == 0 && xcb > 0) xl[u] = x[xlo + ic];
        if (MODE == 1) {
          gv[u] = dy[base + ic];
          if (relu && relu_from_y) yv[u] = y[base + ic];
        }
      }
      // the U vectors of an iteration are summed in fp32 first (U <= 4 terms per channel: 2^-23 of a term), the running sums stay fp64:
      // the fp64 conversions / adds per loaded vector were 32 half-rate VALU operations -- as many issue cycles as the loads' bytes
      // take at 5 TB/s -- and the kernels streamed at 3.3-3.6 TB/s where the apply passes reach 5.5-6 (round 6)
      float p0[8], p1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { p0[j] = 0.f; p1[j] = 0.f; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i + u * 256 >= i1) continue;
        float f[8];
        unpack8x(xv[u], f, xf16);
        if (MODE == 0) {
          if (xcb > 0) {  // (uniform)
            float l[8];
            unpack8h(xl[u], l);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += l[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) { p0[j] += f[j]; p1[j] = fmaf(f[j], f[j], p1[j]); }
        } else {
          float gg[8], yy[8];
          unpack8(gv[u], gg);
          if (relu && relu_from_y) unpack8(yv[u], yy);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = (f[j] - mean[j]) * rstd[j];
            const bool off = relu && (relu_from_y ? yy[j] <= 0.f : (affine_mask ? f[j] * ma[j] + mb[j] <= 0.f : xh <= 0.f));
            const float gr = off ? 0.f : gg[j];
            p0[j] += gr;
            p1[j] = fmaf(gr, xh, p1[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0[j] += (double)p0[j]; s1[j] += (double)p1[j]; }
    }
  }
  // the 16 block sums with ONE barrier (per value: wave sum, then the waves in order -- the order block_sum_d uses; 16 calls of
  // it were 32 barriers at the end of a kernel that streams its slice in about the same time)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double a = wave_sum_d(s0[j]), b = wave_sum_d(s1[j]);
    if (lane == 0) { redd[w * 16 + 2 * j] = a; redd[w * 16 + 2 * j + 1] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0;
    for (int i = 0; i < 4; ++i) t += redd[i * 16 + threadIdx.x];
    sums[((size_t)g * nsl + sl) * 16 + threadIdx.x] = t;
  }
}

// totals of a group's partials for its 8 channels.  blockDim.x / 16 lane groups each walk every (blockDim.x / 16)-th slice (one
// 128-byte line per slice and group, all loads of a group independent), the 16 group sums are combined in group order through
// LDS: one or two memory round trips in front of the apply kernels' stream instead of nsl dependent ones (the second version:
// 16 lanes walking all slices -- 64 serial L2 round trips = ~6 us ahead of a 40 us kernel; the first let every thread total all
// 16 x nsl doubles through scalar loads).  Fixed summation order.  blockDim.x a multiple of 64, <= 1024.
__device__ __forceinline__ void group_total8(const double* sums, int g, int nsl, double (&t0)[8], double (&t1)[8]) {
  __shared__ double part[64 * 16];
  __shared__ double tot[16];
  const int j = threadIdx.x & 15, k0 = threadIdx.x >> 4, ng = blockDim.x >> 4;
  {
    const double* p = sums + (size_t)g * nsl * 16 + j;
    double t = 0;
    for (int k = k0; k < nsl; k += ng) t += p[(size_t)k * 16];
    part[k0 * 16 + j] = t;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0;
    for (int i = 0; i < ng; ++i) t += part[i * 16 + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) { t0[jj] = tot[2 * jj]; t1[jj] = tot[2 * jj + 1]; }
}

// InstanceNorm forward map (grid: (N*CB, chunks))
__global__ __launch_bounds__(256) void in_apply_c8_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ res,
                                                          u32x4n* __restrict__ y, float* __restrict__ stats, const double* sums,
                                                          int nsl, int CB, int C, int hw, float eps, int relu) {
  const int xf16 = relu >> 8;  // (flags: bit 0 = ReLU, bit 8 = x is an F16_C8 tensor)
  relu &= 0xff;
  const int g = blockIdx.x, n = g / CB, cb = g - n * CB;
  double t0[8], t1[8];
  group_total8(sums, g, nsl, t0, t1);
  float mean[8], rstd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double m = t0[j] / hw;
    double var = t1[j] / hw - m * m;
    if (var < 0) var = 0;
    mean[j] = (float)m;
    rstd[j] = (float)(1.0 / sqrt(var + (double)eps));
  }
  if (blockIdx.y == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cb * 8 + j < C) { stats[2 * ((size_t)n * C + cb * 8 + j)] = mean[j]; stats[2 * ((size_t)n * C + cb * 8 + j) + 1] = rstd[j]; }
  }
  const size_t base = (size_t)g * hw;
  constexpr int U = 4;
  const int stride = gridDim.y * 256;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += stride * U) {
    u32x4n xv[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // clamped, unconditional: all in flight together
      const int ii = i + u * stride, ic = ii < hw ? ii : hw - 1;
      xv[u] = x[base + ic];
      if (res) rv[u] = res[base + ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride;
      if (ii >= hw) continue;
      float f[8], rf[8];
      unpack8x(xv[u], f, xf16);
      if (res) unpack8(rv[u], rf);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = (f[j] - mean[j]) * rstd[j];
        if (relu) t = fmaxf(t, 0.f);
        if (res) t += rf[j];
        f[j] = cb * 8 + j < C ? t : 0.f;
      }
      y[base + ii] = pack8n(f);
    }
  }
}

// the split path's forward map with the second (F16_C8) output and an optionally half residual (see in_fwd_c8_mix_kernel; no [hi | lo] x here)
__global__ __launch_bounds__(256) void in_apply_c8_mix_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ res,
                                                              u32x4n* __restrict__ y, u32x4n* __restrict__ y16, float* __restrict__ stats,
                                                              const double* sums, int nsl, int CB, int C, int hw, float eps, int flags) {
  const int relu = flags & 1, xf16 = (flags >> 8) & 1, rf16 = (flags >> 9) & 1, hilo = (flags >> 10) & 1, rpair = (flags >> 11) & 1;
  const int g = blockIdx.x, n = g / CB, cb = g - n * CB;
  double t0[8], t1[8];
  group_total8(sums, g, nsl, t0, t1);
  float mean[8], rstd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double m = t0[j] / hw;
    double var = t1[j] / hw - m * m;
    if (var < 0) var = 0;
    mean[j] = (float)m;
    rstd[j] = (float)(1.0 / sqrt(var + (double)eps));
  }
  if (blockIdx.y == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cb * 8 + j < C) { stats[2 * ((size_t)n * C + cb * 8 + j)] = mean[j]; stats[2 * ((size_t)n * C + cb * 8 + j) + 1] = rstd[j]; }
  }
  const size_t base = (size_t)g * hw;
  const size_t xbase = hilo ? ((size_t)n * 2 * CB + cb) * hw : base, xlo = xbase + (size_t)CB * hw;  // (a [hi | lo] x: 2 CB blocks per sample)
  const size_t rbase = rpair ? ((size_t)n * 2 * CB + cb) * hw : base;
  constexpr int U = 4;
  const int stride = gridDim.y * 256;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += stride * U) {
    u32x4n xv[U], rv[U], xl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride, ic = ii < hw ? ii : hw - 1;
      xv[u] = x[xbase + ic];
      if (hilo) xl[u] = x[xlo + ic];
      if (res) rv[u] = res[rbase + ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride;
      if (ii >= hw) continue;
      float f[8], rf[8];
      unpack8x(xv[u], f, xf16);
      if (hilo) {  // (uniform)
        float l[8];
        unpack8h(xl[u], l);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += l[j];
      }
      if (res) unpack8x(rv[u], rf, rf16);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = (f[j] - mean[j]) * rstd[j];
        if (relu) t = fmaxf(t, 0.f);
        if (res) t += rf[j];
        f[j] = cb * 8 + j < C ? t : 0.f;
      }
      if (y) y[base + ii] = pack8n(f);
      y16[base + ii] = pack8hn(f);
    }
  }
}

__global__ __launch_bounds__(256) void in_bwd_apply_c8_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ dy,
                                                              const float* __restrict__ stats, const double* sums, int nsl,
                                                              u32x4n* __restrict__ dx, int CB, int C, int hw, int relu, int xcb = 0) {
  const int xf16 = (relu >> 8) & 1;  // (flags: bit 0 = ReLU, bit 8 = x is an F16_C8 tensor; xcb > 0: x keeps xcb blocks per sample, the first CB are read)
  relu &= 0xff;
  const int g = blockIdx.x, n = g / CB, cb = g - n * CB;
  double t0[8], t1[8];
  group_total8(sums, g, nsl, t0, t1);
  float mean[8], rstd[8], m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j < C ? cb * 8 + j : C - 1;
    mean[j] = stats[2 * ((size_t)n * C + c)];
    rstd[j] = stats[2 * ((size_t)n * C + c) + 1];
    m1[j] = (float)(t0[j] / hw);
    m2[j] = (float)(t1[j] / hw);
  }
  const size_t base = (size_t)g * hw;
  const size_t xbase = xcb > 0 ? ((size_t)n * xcb + cb) * hw : base;
  constexpr int U = 3;
  const int stride = gridDim.y * 256;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += stride * U) {
    u32x4n xv[U], gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride, ic = ii < hw ? ii : hw - 1;
      xv[u] = x[xbase + ic];
      gv[u] = dy[base + ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride;
      if (ii >= hw) continue;
      float f[8], gg[8];
      unpack8x(xv[u], f, xf16);
      unpack8(gv[u], gg);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (f[j] - mean[j]) * rstd[j];
        const float gr = (relu && xh <= 0.f) ? 0.f : gg[j];
        f[j] = cb * 8 + j < C ? rstd[j] * (gr - m1[j] - xh * m2[j]) : 0.f;
      }
      dx[base + ii] = pack8n(f);
    }
  }
}

// BatchNorm(train) forward map over (n, cb) = blockIdx.x; the n == 0 / chunk 0 block also writes stats and running stats
__global__ __launch_bounds__(256) void bn_apply_c8_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ res,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum, float eps,
                                                          u32x4n* __restrict__ y, float* __restrict__ stats, const double* sums,
                                                          int nsl, int N, int CB, int C, int hw, int relu) {
  const int xf16 = relu >> 8;  // (flags: bit 0 = ReLU, bit 8 = x is an F16_C8 tensor)
  relu &= 0xff;
  const int g = blockIdx.x, cb = g % CB;
  const double cnt = (double)N * hw;
  double t0[8], t1[8];
  group_total8(sums, cb, nsl, t0, t1);
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j < C ? cb * 8 + j : C - 1;
    const double m = t0[j] / cnt;
    double var = t1[j] / cnt - m * m;
    if (var < 0) var = 0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (g < CB && blockIdx.y == 0 && threadIdx.x == 0 && cb * 8 + j < C) {
      stats[2 * c] = mean; stats[2 * c + 1] = rstd;
      // the affine map y = x a + b of THIS forward, for the backward's ReLU mask and dx scale: a later in-place change of
      // gamma / beta (an optimiser writing through raw pointers bumps no version counter) cannot reach it
      stats[2 * (C + c)] = rstd * gamma[c]; stats[2 * (C + c) + 1] = beta[c] - mean * (rstd * gamma[c]);
      if (running_mean) {
        const double unb = cnt > 1 ? var * cnt / (cnt - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
    a[j] = rstd * gamma[c];
    b[j] = beta[c] - mean * a[j];
  }
  const size_t base = (size_t)g * hw;
  constexpr int U = 4;
  const int stride = gridDim.y * 256;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += stride * U) {
    u32x4n xv[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride, ic = ii < hw ? ii : hw - 1;
      xv[u] = x[base + ic];
      if (res) rv[u] = res[base + ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride;
      if (ii >= hw) continue;
      float f[8], rf[8];
      unpack8x(xv[u], f, xf16);
      if (res) unpack8(rv[u], rf);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = f[j] * a[j] + b[j];
        if (res) t += rf[j];
        if (relu) t = fmaxf(t, 0.f);
        f[j] = cb * 8 + j < C ? t : 0.f;
      }
      y[base + ii] = pack8n(f);
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_c8_kernel(const u32x4n* __restrict__ x, const u32x4n* __restrict__ y,
                                                              const u32x4n* __restrict__ dy, const float* __restrict__ gamma,
                                                              const float* __restrict__ stats, const double* sums, int nsl,
                                                              u32x4n* __restrict__ dx, u32x4n* __restrict__ dres, float* dgamma,
                                                              float* dbeta, int accumulate, int N, int CB, int C, int hw, int relu,
                                                              const float* __restrict__ beta) {
  const int xf16 = relu >> 8;  // (flags: bit 0 = ReLU, bit 8 = x is an F16_C8 tensor)
  relu &= 0xff;
  const bool mask_x = relu && beta != nullptr;  // the forward had no residual: y <= 0 <=> x a + b <= 0, y is not read
  const int g = blockIdx.x, cb = g % CB;
  const double cnt = (double)N * hw;
  double t0[8], t1[8];
  group_total8(sums, cb, nsl, t0, t1);
  float mean[8], rstd[8], m1[8], m2[8], gr[8], ma[8], mb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j < C ? cb * 8 + j : C - 1;
    mean[j] = stats[2 * c];
    rstd[j] = stats[2 * c + 1];
    if (g < CB && blockIdx.y == 0 && threadIdx.x == 0 && cb * 8 + j < C) {
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)t0[j] : (float)t0[j];
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)t1[j] : (float)t1[j];
    }
    m1[j] = (float)(t0[j] / cnt);
    m2[j] = (float)(t1[j] / cnt);
    gr[j] = stats[2 * (C + c)];  // = gamma * rstd as the forward used it (saved map, see bn_apply_c8_kernel)
    ma[j] = mb[j] = 0.f;
    if (mask_x) {
      ma[j] = stats[2 * (C + c)];
      mb[j] = stats[2 * (C + c) + 1];
    }
  }
  const size_t base = (size_t)g * hw;
  constexpr int U = 2;
  const int stride = gridDim.y * 256;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += stride * U) {
    u32x4n gv[U], yv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride, ic = ii < hw ? ii : hw - 1;
      gv[u] = dy[base + ic];
      if (relu && !mask_x) yv[u] = y[base + ic];
      if (dx || mask_x) xv[u] = x[base + ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u * stride;
      if (ii >= hw) continue;
      float gg[8], yy[8], f[8];
      unpack8(gv[u], gg);
      if (dx || mask_x) unpack8x(xv[u], f, xf16);
      if (mask_x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (f[j] * ma[j] + mb[j] <= 0.f) gg[j] = 0.f;
      } else if (relu) {
        unpack8(yv[u], yy);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (yy[j] <= 0.f) gg[j] = 0.f;
      }
      if (dres) dres[base + ii] = pack8n(gg);
      if (dx) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = cb * 8 + j < C ? gr[j] * (gg[j] - m1[j] - (f[j] - mean[j]) * rstd[j] * m2[j]) : 0.f;
        dx[base + ii] = pack8n(f);
      }
    }
  }
}

inline int split_for8(int groups, int hw) {
  // workgroups the statistics pass aims at (x slices per group) and the smallest slice (ESS_NORM_SPLIT_WGS / ESS_NORM_SPLIT_MINV: tuning)
  static const int target = [] { const char* e = getenv("ESS_NORM_SPLIT_WGS"); return e ? atoi(e) : 1024; }();
  static const int minv = [] { const char* e = getenv("ESS_NORM_SPLIT_MINV"); return e ? atoi(e) : 1024; }();
  int s = (target + groups - 1) / groups;
  const int maxs = (hw + minv - 1) / minv;  // at least `minv` pixel vectors per slice
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}
inline int chunks_for8(int groups, int hw) {
  int c = (2048 + groups - 1) / groups;
  const int maxc = (hw + 511) / 512;
  if (c > maxc) c = maxc;
  return c < 1 ? 1 : c;
}
// fused 1024-thread InstanceNorm forward: one workgroup per (n, channel block) -- worth it only with enough groups to cover the CUs
inline int in1024_min_groups() {
  static const int v = [] { const char* e = getenv("ESS_IN1024_MIN_GROUPS"); return e ? atoi(e) : 256; }();  // (B=8: 64 / 128 groups at 120x160 measured 34 / 37 us fused against 28 / 34 us split)
  return v;
}
inline int need_ws(void* ws, size_t need, size_t have, const char* what) {
  if (!ws || have < need) { ess_set_error("%s: workspace too small (%zu < %zu)", what, have, need); return ESS_EINVAL; }
  return ESS_OK;
}
inline bool al16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr, const void* e = nullptr) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) | ((uintptr_t)e)) & 15) == 0;
}

}  // namespace

// (sum, sum) pairs of doubles per group, slice and channel of the block; split_for8() keeps groups * slices <= 1024 + groups
extern "C" size_t ess_norm_workspace_c8(int32_t groups) { return (size_t)(groups > 0 ? 64 * (size_t)groups + 4096 : 0) * 8 * 16; }  // (<= 64 slices per group)

// "in_small_threads" (ESS_IN_SMALL_THREADS: 256 | 512 | 1024): threads of the fused single-plane InstanceNorm kernels on planes of
// at most 5120 vectors (the decoder's 60 x 80 level).  Every setting computes the same statistics up to the summation order.
static std::atomic<int> g_in_small{-1};
int in_small_threads() {
  int v = g_in_small.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("ESS_IN_SMALL_THREADS");
    v = e ? atoi(e) : 512;  // (B = 8, 256 channels @ 60 x 80, MI355X: forward 17.8 / 16.7 / 17.4 us, backward 22.4 / 17.5 / 19.8 us at 256 / 512 / 1024)
    g_in_small.store(v, std::memory_order_relaxed);
  }
  return v;
}
void set_in_small_threads(int v) { g_in_small.store(v == 256 || v == 1024 ? v : 512, std::memory_order_relaxed); }

extern "C" int ess_instnorm_forward_c8(const void* x, const void* residual, void* y, float* stats, int32_t N, int32_t C,
                                       int32_t hw, float eps, int32_t relu, int32_t x_f16, void* workspace, size_t workspace_bytes,
                                       ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && stats && N > 0 && C > 0 && hw > 0, "instnorm_forward_c8: bad arguments");
  ESS_CHECK_ARG(relu == 0 || relu == 1, "instnorm_forward_c8: relu must be 0 or 1");
  ESS_CHECK_ARG(al16(x, residual, y), "instnorm_forward_c8: BF16_C8 tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  relu = (relu & 1) | (x_f16 ? 0x100 : 0);  // (the kernels' flag word)
  const int CB = (C + 7) / 8, groups = N * CB;
  const u32x4n* xs = (const u32x4n*)x; const u32x4n* rs = (const u32x4n*)residual; u32x4n* ys = (u32x4n*)y;
  if (hw <= 256 * MAXV) {
    const int th = in_small_threads();
    if (th == 1024 && hw <= 1024 * 5) hipLaunchKernelGGL((in_fwd_c8_kernel<1024, 5>), dim3(groups), dim3(1024), 0, st, xs, rs, ys, stats, CB, C, hw, eps, relu);
    else if (th == 512 && hw <= 512 * 10) hipLaunchKernelGGL((in_fwd_c8_kernel<512, 10>), dim3(groups), dim3(512), 0, st, xs, rs, ys, stats, CB, C, hw, eps, relu);
    else hipLaunchKernelGGL(in_fwd_c8_kernel<256>, dim3(groups), dim3(256), 0, st, xs, rs, ys, stats, CB, C, hw, eps, relu);
    return ess_launch_status("instnorm_forward_c8");
  }
  if (hw <= 1024 * 19 && groups >= in1024_min_groups()) {
    hipLaunchKernelGGL(in_fwd_c8_kernel<1024>, dim3(groups), dim3(1024), 0, st, xs, rs, ys, stats, CB, C, hw, eps, relu);
    return ess_launch_status("instnorm_forward_c8");
  }
  int rc = need_ws(workspace, ess_norm_workspace_c8(groups), workspace_bytes, "instnorm_forward_c8");
  if (rc) return rc;
  const int nsl = split_for8(groups, hw);
  hipLaunchKernelGGL((c8_reduce_kernel<0>), dim3(groups, nsl), dim3(256), 0, st, xs, nullptr, nullptr, nullptr, (double*)workspace, hw, 1,
                     0, CB, C, 1, relu & 0x100, 0);
  hipLaunchKernelGGL(in_apply_c8_kernel, dim3(groups, chunks_for8(groups, hw)), dim3(256), 0, st, xs, rs, ys, stats,
                     (const double*)workspace, nsl, CB, C, hw, eps, relu);
  return ess_launch_status("instnorm_forward_c8(split)");
}

extern "C" int ess_instnorm_forward_c8_mixed(const void* x, const void* residual, void* y, void* y16, float* stats, int32_t N, int32_t C,
                                             int32_t hw, float eps, int32_t relu, int32_t x_fmt, int32_t res_f16, void* workspace,
                                             size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y16 && stats && N > 0 && C > 0 && hw > 0, "instnorm_forward_c8_mixed: bad arguments");
  ESS_CHECK_ARG(relu == 0 || relu == 1, "instnorm_forward_c8_mixed: relu must be 0 or 1");
  ESS_CHECK_ARG(x_fmt >= 0 && x_fmt <= 2, "instnorm_forward_c8_mixed: x_fmt is 0 (BF16_C8), 1 (F16_C8) or 2 (F16_C8 [hi | lo])");
  ESS_CHECK_ARG(res_f16 >= 0 && res_f16 <= 2, "instnorm_forward_c8_mixed: res_f16 is 0 (BF16_C8), 1 (F16_C8) or 2 (a [hi | lo] half pair: its hi parts are added)");
  ESS_CHECK_ARG(al16(x, residual, y, y16), "instnorm_forward_c8_mixed: C8 tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int flags = (relu & 1) | (x_fmt ? 0x100 : 0) | (res_f16 ? 0x200 : 0) | (res_f16 == 2 ? 0x800 : 0);
  const int CB = (C + 7) / 8, groups = N * CB;
  const u32x4n* xs = (const u32x4n*)x; const u32x4n* rs = (const u32x4n*)residual; u32x4n* ys = (u32x4n*)y; u32x4n* hs = (u32x4n*)y16;
  if (hw <= 256 * MAXV) {
    if (x_fmt == 2) {
      if (hw <= 512 * 10) hipLaunchKernelGGL((in_fwd_c8_mix_kernel<512, 10, true>), dim3(groups), dim3(512), 0, st, xs, rs, ys, hs, stats, CB, C, hw, eps, flags);
      else hipLaunchKernelGGL((in_fwd_c8_mix_kernel<256, MAXV, true>), dim3(groups), dim3(256), 0, st, xs, rs, ys, hs, stats, CB, C, hw, eps, flags);
    } else {
      if (hw <= 512 * 10) hipLaunchKernelGGL((in_fwd_c8_mix_kernel<512, 10, false>), dim3(groups), dim3(512), 0, st, xs, rs, ys, hs, stats, CB, C, hw, eps, flags);
      else hipLaunchKernelGGL((in_fwd_c8_mix_kernel<256, MAXV, false>), dim3(groups), dim3(256), 0, st, xs, rs, ys, hs, stats, CB, C, hw, eps, flags);
    }
    return ess_launch_status("instnorm_forward_c8_mixed");
  }
  int rc = need_ws(workspace, ess_norm_workspace_c8(groups), workspace_bytes, "instnorm_forward_c8_mixed");
  if (rc) return rc;
  const int nsl = split_for8(groups, hw);
  hipLaunchKernelGGL((c8_reduce_kernel<0>), dim3(groups, nsl), dim3(256), 0, st, xs, nullptr, nullptr, nullptr, (double*)workspace, hw, 1,
                     0, CB, C, 1, flags & 0x100, 0, nullptr, nullptr, x_fmt == 2 ? 2 * CB : 0);
  hipLaunchKernelGGL(in_apply_c8_mix_kernel, dim3(groups, chunks_for8(groups, hw)), dim3(256), 0, st, xs, rs, ys, hs, stats,
                     (const double*)workspace, nsl, CB, C, hw, eps, flags | (x_fmt == 2 ? 0x400 : 0));
  return ess_launch_status("instnorm_forward_c8_mixed(split)");
}

extern "C" int ess_instnorm_backward_c8(const void* x, const void* dy, const float* stats, void* dx, int32_t N, int32_t C,
                                        int32_t hw, int32_t relu, int32_t x_f16, void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(x && dy && stats && dx && N > 0 && C > 0 && hw > 0, "instnorm_backward_c8: bad arguments");
  ESS_CHECK_ARG(relu == 0 || relu == 1, "instnorm_backward_c8: relu must be 0 or 1");
  ESS_CHECK_ARG(al16(x, dy, dx), "instnorm_backward_c8: BF16_C8 tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  // x_f16 = 2: x is a [hi | lo] half pair ([N][2 CB][hw][8], the mixed configuration's first decoder layer): the hi parts are read
  ESS_CHECK_ARG(x_f16 >= 0 && x_f16 <= 2, "instnorm_backward_c8: x_f16 is 0 (BF16_C8), 1 (F16_C8) or 2 (a [hi | lo] half pair: its hi parts are read)");
  const int xcb = x_f16 == 2 ? 2 * ((C + 7) / 8) : 0;
  relu = (relu & 1) | (x_f16 ? 0x100 : 0);  // (the kernels' flag word)
  const int CB = (C + 7) / 8, groups = N * CB;
  const u32x4n* xs = (const u32x4n*)x; const u32x4n* gs = (const u32x4n*)dy; u32x4n* ds = (u32x4n*)dx;
  if (hw <= 256 * MAXV) {
    const int th = in_small_threads();
    if (th == 1024 && hw <= 1024 * 5) hipLaunchKernelGGL((in_bwd_c8_kernel<1024, 5>), dim3(groups), dim3(1024), 0, st, xs, gs, stats, ds, CB, C, hw, relu, xcb);
    else if (th == 512 && hw <= 512 * 10) hipLaunchKernelGGL((in_bwd_c8_kernel<512, 10>), dim3(groups), dim3(512), 0, st, xs, gs, stats, ds, CB, C, hw, relu, xcb);
    else hipLaunchKernelGGL((in_bwd_c8_kernel<256, MAXV>), dim3(groups), dim3(256), 0, st, xs, gs, stats, ds, CB, C, hw, relu, xcb);
    return ess_launch_status("instnorm_backward_c8");
  }
  int rc = need_ws(workspace, ess_norm_workspace_c8(groups), workspace_bytes, "instnorm_backward_c8");
  if (rc) return rc;
  const int nsl = split_for8(groups, hw);
  hipLaunchKernelGGL((c8_reduce_kernel<1>), dim3(groups, nsl), dim3(256), 0, st, xs, nullptr, gs, stats, (double*)workspace, hw, 1, 0, CB,
                     C, 1, relu, 0, nullptr, nullptr, xcb);
  hipLaunchKernelGGL(in_bwd_apply_c8_kernel, dim3(groups, chunks_for8(groups, hw)), dim3(256), 0, st, xs, gs, stats,
                     (const double*)workspace, nsl, ds, CB, C, hw, relu, xcb);
  return ess_launch_status("instnorm_backward_c8(split)");
}

extern "C" int ess_batchnorm_train_forward_c8(const void* x, const void* residual, const float* gamma, const float* beta,
                                              float* running_mean, float* running_var, float momentum, float eps, void* y,
                                              float* stats, int32_t N, int32_t C, int32_t hw, int32_t relu, int32_t x_f16, void* workspace,
                                              size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(x && gamma && beta && y && stats && N > 0 && C > 0 && hw > 0, "batchnorm_train_forward_c8: bad arguments");
  ESS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "batchnorm_train_forward_c8: running stats come in pairs");
  ESS_CHECK_ARG(al16(x, residual, y), "batchnorm_train_forward_c8: BF16_C8 tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  relu = (relu & 1) | (x_f16 ? 0x100 : 0);  // (the kernels' flag word)
  const int CB = (C + 7) / 8;
  int rc = need_ws(workspace, ess_norm_workspace_c8(CB), workspace_bytes, "batchnorm_train_forward_c8");
  if (rc) return rc;
  const int nsl = split_for8(CB, hw);
  hipLaunchKernelGGL((c8_reduce_kernel<0>), dim3(CB, nsl), dim3(256), 0, st, (const u32x4n*)x, nullptr, nullptr, nullptr,
                     (double*)workspace, hw, N, CB, CB, C, 0, relu & 0x100, 0);
  hipLaunchKernelGGL(bn_apply_c8_kernel, dim3(N * CB, chunks_for8(N * CB, hw)), dim3(256), 0, st, (const u32x4n*)x,
                     (const u32x4n*)residual, gamma, beta, running_mean, running_var, momentum, eps, (u32x4n*)y, stats,
                     (const double*)workspace, nsl, N, CB, C, hw, relu);
  return ess_launch_status("batchnorm_train_forward_c8");
}

extern "C" int ess_batchnorm_train_backward_c8(const void* x, const void* y, const void* dy, const float* gamma, const float* beta,
                                               const float* stats, void* dx, void* d_residual, float* dgamma, float* dbeta,
                                               int32_t accumulate, int32_t N, int32_t C, int32_t hw, int32_t relu, int32_t x_f16,
                                               void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  // beta != NULL (with relu): the forward was relu(bn(x)) without a residual -- the mask is recomputed from x, y is not read
  const bool mask_x = (relu & 1) && beta != nullptr;
  ESS_CHECK_ARG(x && (y || mask_x || !(relu & 1)) && dy && gamma && stats && N > 0 && C > 0 && hw > 0, "batchnorm_train_backward_c8: bad arguments");
  ESS_CHECK_ARG(!(mask_x && d_residual), "batchnorm_train_backward_c8: a residual gradient needs the mask of the saved output (beta must be NULL)");
  ESS_CHECK_ARG(al16(x, y, dy, dx, d_residual), "batchnorm_train_backward_c8: BF16_C8 tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  relu = (relu & 1) | (x_f16 ? 0x100 : 0);  // (the kernels' flag word)
  const int CB = (C + 7) / 8;
  int rc = need_ws(workspace, ess_norm_workspace_c8(CB), workspace_bytes, "batchnorm_train_backward_c8");
  if (rc) return rc;
  const int nsl = split_for8(CB, hw);
  hipLaunchKernelGGL((c8_reduce_kernel<1>), dim3(CB, nsl), dim3(256), 0, st, (const u32x4n*)x, (const u32x4n*)y, (const u32x4n*)dy,
                     stats, (double*)workspace, hw, N, CB, CB, C, 0, relu, mask_x ? 0 : 1, mask_x ? gamma : nullptr, mask_x ? beta : nullptr);
  hipLaunchKernelGGL(bn_bwd_apply_c8_kernel, dim3(N * CB, chunks_for8(N * CB, hw)), dim3(256), 0, st, (const u32x4n*)x,
                     (const u32x4n*)y, (const u32x4n*)dy, gamma, stats, (const double*)workspace, nsl, (u32x4n*)dx,
                     (u32x4n*)d_residual, dgamma, dbeta, accumulate, N, CB, C, hw, relu, mask_x ? beta : nullptr);
  return ess_launch_status("batchnorm_train_backward_c8");
}

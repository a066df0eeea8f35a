// InstanceNorm2d (decoder) and train-mode BatchNorm2d (image encoder): HBM-bound plane reductions.
// One workgroup per (n,c) plane (IN) or per channel (BN); statistics are accumulated in fp64 so a
// single E[x^2]-E[x]^2 pass is as accurate as a two-pass/Welford fp32 scheme; second pass applies.
#include "common.h"

namespace {

constexpr int NT = 512;

__global__ __launch_bounds__(NT) void instnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          float* __restrict__ y, float* __restrict__ stats, int hw,
                                                          float eps, int relu) {
  __shared__ double red[16];
  const size_t base = (size_t)blockIdx.x * hw;
  const float* xp = x + base;
  double s = 0, ss = 0;
  // small planes (<= 3 x 16 bytes per thread): x is read once and stays in registers between the reduction and the map
  constexpr int MAXV = 3;
  if ((hw & 3) == 0 && hw <= NT * 4 * MAXV && ((((uintptr_t)x) | ((uintptr_t)res) | ((uintptr_t)y)) & 15) == 0) {
    f32x4 xv[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int i = (threadIdx.x + k * NT) * 4;
      if (i < hw) {
        xv[k] = *(const f32x4*)(xp + i);
        s += (double)xv[k][0] + (double)xv[k][1] + (double)xv[k][2] + (double)xv[k][3];
        ss += (double)xv[k][0] * xv[k][0] + (double)xv[k][1] * xv[k][1] + (double)xv[k][2] * xv[k][2] + (double)xv[k][3] * xv[k][3];
      }
    }
    s = block_sum_d(s, red);
    ss = block_sum_d(ss, red);
    const double mean_d = s / hw;
    double var = ss / hw - mean_d * mean_d;
    if (var < 0) var = 0;
    const float mean = (float)mean_d;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int i = (threadIdx.x + k * NT) * 4;
      if (i < hw) {
        f32x4 v = xv[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = (v[j] - mean) * rstd;
          if (relu == 1) t = fmaxf(t, 0.f);
          v[j] = t;
        }
        if (res) { const f32x4 r = *(const f32x4*)(res + base + i); v += r; }
        if (relu == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *(f32x4*)(y + base + i) = v;
      }
    }
    return;
  }
  if ((hw & 3) == 0) {
    for (int i = threadIdx.x * 4; i < hw; i += NT * 4) {
      const f32x4 v = *(const f32x4*)(xp + i);
      s += (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3];
      ss += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += NT) { const double v = xp[i]; s += v; ss += v * v; }
  }
  s = block_sum_d(s, red);
  ss = block_sum_d(ss, red);
  const double mean_d = s / hw;
  double var = ss / hw - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
  float* yp = y + base;
  const float* rp = res ? res + base : nullptr;
  if ((hw & 3) == 0) {
    for (int i = threadIdx.x * 4; i < hw; i += NT * 4) {
      f32x4 v = *(const f32x4*)(xp + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = (v[j] - mean) * rstd;
        if (relu == 1) t = fmaxf(t, 0.f);
        v[j] = t;
      }
      if (rp) { const f32x4 r = *(const f32x4*)(rp + i); v += r; }
      if (relu == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      *(f32x4*)(yp + i) = v;
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += NT) {
      float t = (xp[i] - mean) * rstd;
      if (relu == 1) t = fmaxf(t, 0.f);
      if (rp) t += rp[i];
      if (relu == 2) t = fmaxf(t, 0.f);
      yp[i] = t;
    }
  }
}

__global__ __launch_bounds__(NT) void instnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ stats, float* __restrict__ dx, int hw,
                                                          int relu) {
  __shared__ double red[16];
  const size_t base = (size_t)blockIdx.x * hw;
  const float* xp = x + base;
  const float* gp = dy + base;
  const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
  double s1 = 0, s2 = 0;
  // small planes (<= 3 x 16 bytes per thread: the 60x80 decoder planes): one 16-byte read of x and dy, the normalised
  // values and masked gradients stay in registers between the reduction and the map
  constexpr int MAXV = 3;
  if ((hw & 3) == 0 && hw <= NT * 4 * MAXV && ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0) {
    f32x4 xh[MAXV], g[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int i = (threadIdx.x + k * NT) * 4;
      if (i < hw) {
        const f32x4 xv = *(const f32x4*)(xp + i);
        g[k] = *(const f32x4*)(gp + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[k][j] = (xv[j] - mean) * rstd;
          if (relu && xh[k][j] <= 0.f) g[k][j] = 0.f;
          s1 += g[k][j];
          s2 += (double)g[k][j] * xh[k][j];
        }
      }
    }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    const float m1 = (float)(s1 / hw), m2 = (float)(s2 / hw);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int i = (threadIdx.x + k * NT) * 4;
      if (i < hw) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (g[k][j] - m1 - xh[k][j] * m2);
        *(f32x4*)(dx + base + i) = o;
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < hw; i += NT) {
    const float xh = (xp[i] - mean) * rstd;
    float g = gp[i];
    if (relu && xh <= 0.f) g = 0.f;
    s1 += g;
    s2 += (double)g * xh;
  }
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  const float m1 = (float)(s1 / hw), m2 = (float)(s2 / hw);
  float* op = dx + base;
  for (int i = threadIdx.x; i < hw; i += NT) {
    const float xh = (xp[i] - mean) * rstd;
    float g = gp[i];
    if (relu && xh <= 0.f) g = 0.f;
    op[i] = rstd * (g - m1 - xh * m2);
  }
}

__global__ __launch_bounds__(NT) void bn_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum, float eps,
                                                          float* __restrict__ y, float* __restrict__ stats, int N, int C,
                                                          int hw, int relu) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  double s = 0, ss = 0;
  for (int n = 0; n < N; ++n) {
    const float* xp = x + ((size_t)n * C + c) * hw;
    for (int i = threadIdx.x; i < hw; i += NT) { const double v = xp[i]; s += v; ss += v * v; }
  }
  s = block_sum_d(s, red);
  ss = block_sum_d(ss, red);
  const double cnt = (double)N * hw;
  const double mean_d = s / cnt;
  double var = ss / cnt - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    stats[2 * c] = mean; stats[2 * c + 1] = rstd;
    if (running_mean) {
      const double unb = cnt > 1 ? var * cnt / (cnt - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
  const float g = gamma[c], b = beta[c];
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * hw;
    for (int i = threadIdx.x; i < hw; i += NT) {
      float t = (x[base + i] - mean) * rstd * g + b;
      if (res) t += res[base + i];
      if (relu) t = fmaxf(t, 0.f);
      y[base + i] = t;
    }
  }
}

__global__ __launch_bounds__(NT) void bn_train_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ dy, const float* __restrict__ gamma,
                                                          const float* __restrict__ stats, float* __restrict__ dx,
                                                          float* __restrict__ dres, float* dgamma, float* dbeta,
                                                          int accumulate, int N, int C, int hw, int relu) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  const float mean = stats[2 * c], rstd = stats[2 * c + 1];
  double s1 = 0, s2 = 0;
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * hw;
    for (int i = threadIdx.x; i < hw; i += NT) {
      float g = dy[base + i];
      if (relu && y[base + i] <= 0.f) g = 0.f;
      s1 += g;
      s2 += (double)g * ((x[base + i] - mean) * rstd);
    }
  }
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
  }
  const double cnt = (double)N * hw;
  const float m1 = (float)(s1 / cnt), m2 = (float)(s2 / cnt), gr = gamma[c] * rstd;
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * hw;
    for (int i = threadIdx.x; i < hw; i += NT) {
      float g = dy[base + i];
      if (relu && y[base + i] <= 0.f) g = 0.f;
      if (dres) dres[base + i] = g;
      if (dx) dx[base + i] = gr * (g - m1 - (x[base + i] - mean) * rstd * m2);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Split variants for few-but-large reduction groups (full-resolution InstanceNorm planes, BatchNorm channels):
// pass 1 accumulates per-group partial sums from many workgroups into fp64 atomics, pass 2 is a flat elementwise map.
// group g covers `nseg` segments of `hw` contiguous floats, segment j at ((j * seg_stride) + g) * hw
// (InstanceNorm: nseg = 1; BatchNorm: nseg = N, seg_stride = C).

// sums[g][slice][0..1] = (sum f0, sum f1) over the slice, with (f0, f1) = (x, x^2) [MODE 0] or (g, g*xhat) [MODE 1, backward]
template <int MODE>
__global__ __launch_bounds__(256) void group_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ stats,
                                                           double* sums, int hw, int nseg, int seg_stride, int relu,
                                                           int relu_from_y) {
  __shared__ double red[16];
  const int g = blockIdx.x, nsl = gridDim.y, sl = blockIdx.y;
  const int len = ((hw + nsl - 1) / nsl + 3) & ~3;
  const int i0 = sl * len, i1 = min(hw, i0 + len);
  float mean = 0.f, rstd = 0.f;
  if (MODE == 1) { mean = stats[2 * g]; rstd = stats[2 * g + 1]; }
  double s0 = 0, s1 = 0;
  // 16-byte loads when the planes allow it (hw % 4 == 0: every segment start is then a multiple of 4 elements); the
  // per-element arithmetic and its fp64 accumulation are unchanged
  const bool vec = (hw & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dy)) & 15) == 0;
  for (int j = 0; j < nseg; ++j) {
    const size_t base = ((size_t)j * seg_stride + g) * hw;
    if (vec) {
      for (int i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
        const f32x4 xv = *(const f32x4*)(x + base + i);
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { const double v = xv[k]; s0 += v; s1 += v * v; }
        } else {
          const f32x4 gv = *(const f32x4*)(dy + base + i);
          f32x4 yv = xv;
          if (relu && relu_from_y) yv = *(const f32x4*)(y + base + i);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mean) * rstd;
            float gr = gv[k];
            if (relu && (relu_from_y ? yv[k] <= 0.f : xh <= 0.f)) gr = 0.f;
            s0 += gr; s1 += (double)gr * xh;
          }
        }
      }
    } else {
      for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        if (MODE == 0) {
          const double v = x[base + i];
          s0 += v; s1 += v * v;
        } else {
          const float xh = (x[base + i] - mean) * rstd;
          float gr = dy[base + i];
          if (relu && (relu_from_y ? y[base + i] <= 0.f : xh <= 0.f)) gr = 0.f;
          s0 += gr; s1 += (double)gr * xh;
        }
      }
    }
  }
  s0 = block_sum_d(s0, red);
  s1 = block_sum_d(s1, red);
  // one partial per (group, slice): no atomics, no zeroing of the workspace, and a fixed summation order downstream
  if (threadIdx.x == 0) { sums[((size_t)g * nsl + sl) * 2] = s0; sums[((size_t)g * nsl + sl) * 2 + 1] = s1; }
}

// total of a group's partials (wave-uniform addresses: scalar loads), in slice order
__device__ __forceinline__ void group_total(const double* sums, int g, int nsl, double& t0, double& t1) {
  t0 = 0; t1 = 0;
  for (int k = 0; k < nsl; ++k) { t0 += sums[((size_t)g * nsl + k) * 2]; t1 += sums[((size_t)g * nsl + k) * 2 + 1]; }
}

// InstanceNorm forward map: y = act(IN(x)) (+ residual) from the group sums; (chunk 0, thread 0) writes (mean, rstd)
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             float* __restrict__ y, float* __restrict__ stats,
                                                             const double* sums, int nsl, int hw, float eps, int relu) {
  const int g = blockIdx.x;
  double t0, t1;
  group_total(sums, g, nsl, t0, t1);
  const double mean_d = t0 / hw;
  double var = t1 / hw - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (blockIdx.y == 0 && threadIdx.x == 0) { stats[2 * g] = mean; stats[2 * g + 1] = rstd; }
  const size_t base = (size_t)g * hw;
  if ((hw & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)res) | ((uintptr_t)y)) & 15) == 0) {  // 16-byte accesses
    for (int i = 4 * (blockIdx.y * 256 + threadIdx.x); i < hw; i += gridDim.y * 1024) {
      const f32x4 xv = *(const f32x4*)(x + base + i);
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (res) rv = *(const f32x4*)(res + base + i);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = (xv[k] - mean) * rstd;
        if (relu == 1) t = fmaxf(t, 0.f);
        if (res) t += rv[k];
        if (relu == 2) t = fmaxf(t, 0.f);
        o[k] = t;
      }
      *(f32x4*)(y + base + i) = o;
    }
    return;
  }
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += gridDim.y * 256) {
    float t = (x[base + i] - mean) * rstd;
    if (relu == 1) t = fmaxf(t, 0.f);
    if (res) t += res[base + i];
    if (relu == 2) t = fmaxf(t, 0.f);
    y[base + i] = t;
  }
}

__global__ __launch_bounds__(256) void instnorm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ stats, const double* sums, int nsl,
                                                                 float* __restrict__ dx, int hw, int relu) {
  const int g = blockIdx.x;
  const float mean = stats[2 * g], rstd = stats[2 * g + 1];
  double t0, t1;
  group_total(sums, g, nsl, t0, t1);
  const float m1 = (float)(t0 / hw), m2 = (float)(t1 / hw);
  const size_t base = (size_t)g * hw;
  if ((hw & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0) {  // 16-byte accesses
    for (int i = 4 * (blockIdx.y * 256 + threadIdx.x); i < hw; i += gridDim.y * 1024) {
      const f32x4 xv = *(const f32x4*)(x + base + i), gv = *(const f32x4*)(dy + base + i);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xv[k] - mean) * rstd;
        float gr = gv[k];
        if (relu && xh <= 0.f) gr = 0.f;
        o[k] = rstd * (gr - m1 - xh * m2);
      }
      *(f32x4*)(dx + base + i) = o;
    }
    return;
  }
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += gridDim.y * 256) {
    const float xh = (x[base + i] - mean) * rstd;
    float gr = dy[base + i];
    if (relu && xh <= 0.f) gr = 0.f;
    dx[base + i] = rstd * (gr - m1 - xh * m2);
  }
}

// BatchNorm(train) forward map over plane (n, c) = blockIdx.x; the n == 0 / chunk 0 block also updates running stats
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* running_mean, float* running_var, float momentum, float eps,
                                                       float* __restrict__ y, float* __restrict__ stats, const double* sums,
                                                       int nsl, int N, int C, int hw, int relu) {
  const int plane = blockIdx.x, c = plane % C;
  const double cnt = (double)N * hw;
  double t0, t1;
  group_total(sums, c, nsl, t0, t1);
  const double mean_d = t0 / cnt;
  double var = t1 / cnt - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (plane < C && blockIdx.y == 0 && threadIdx.x == 0) {
    stats[2 * c] = mean; stats[2 * c + 1] = rstd;
    if (running_mean) {
      const double unb = cnt > 1 ? var * cnt / (cnt - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
  const float gm = gamma[c], bt = beta[c];
  const size_t base = (size_t)plane * hw;
  if ((hw & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)res) | ((uintptr_t)y)) & 15) == 0) {  // 16-byte accesses
    for (int i = 4 * (blockIdx.y * 256 + threadIdx.x); i < hw; i += gridDim.y * 1024) {
      const f32x4 xv = *(const f32x4*)(x + base + i);
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (res) rv = *(const f32x4*)(res + base + i);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = (xv[k] - mean) * rstd * gm + bt;
        if (res) t += rv[k];
        if (relu) t = fmaxf(t, 0.f);
        o[k] = t;
      }
      *(f32x4*)(y + base + i) = o;
    }
    return;
  }
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += gridDim.y * 256) {
    float t = (x[base + i] - mean) * rstd * gm + bt;
    if (res) t += res[base + i];
    if (relu) t = fmaxf(t, 0.f);
    y[base + i] = t;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ gamma,
                                                           const float* __restrict__ stats, const double* sums, int nsl,
                                                           float* __restrict__ dx, float* __restrict__ dres, float* dgamma,
                                                           float* dbeta, int accumulate, int N, int C, int hw, int relu) {
  const int plane = blockIdx.x, c = plane % C;
  const float mean = stats[2 * c], rstd = stats[2 * c + 1];
  const double cnt = (double)N * hw;
  double s1, s2;
  group_total(sums, c, nsl, s1, s2);
  if (plane < C && blockIdx.y == 0 && threadIdx.x == 0) {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
  }
  const float m1 = (float)(s1 / cnt), m2 = (float)(s2 / cnt), gr = gamma[c] * rstd;
  const size_t base = (size_t)plane * hw;
  if ((hw & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)dres)) & 15) == 0) {
    for (int i = 4 * (blockIdx.y * 256 + threadIdx.x); i < hw; i += gridDim.y * 1024) {  // 16-byte accesses
      f32x4 g = *(const f32x4*)(dy + base + i);
      if (relu) {
        const f32x4 yv = *(const f32x4*)(y + base + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (yv[k] <= 0.f) g[k] = 0.f;
      }
      if (dres) *(f32x4*)(dres + base + i) = g;
      if (dx) {
        const f32x4 xv = *(const f32x4*)(x + base + i);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = gr * (g[k] - m1 - (xv[k] - mean) * rstd * m2);
        *(f32x4*)(dx + base + i) = o;
      }
    }
    return;
  }
  for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += gridDim.y * 256) {
    float g = dy[base + i];
    if (relu && y[base + i] <= 0.f) g = 0.f;
    if (dres) dres[base + i] = g;
    if (dx) dx[base + i] = gr * (g - m1 - (x[base + i] - mean) * rstd * m2);
  }
}

inline int split_for(int groups, int hw) {
  int s = (2048 + groups - 1) / groups;
  const int maxs = (hw + 2047) / 2048;  // at least 2048 elements per slice
  if (s > maxs) s = maxs;
  return s < 1 ? 1 : s;
}
inline int chunks_for(int planes, int hw) {
  int c = (4096 + planes - 1) / planes;
  const int maxc = (hw + 1023) / 1024;
  if (c > maxc) c = maxc;
  return c < 1 ? 1 : c;
}
inline int zero_ws(void* ws, size_t need, size_t have, hipStream_t, const char* what) {  // (nothing left to zero: see group_reduce_kernel)
  if (!ws || have < need) { ess_set_error("%s: workspace too small (%zu < %zu)", what, have, need); return ESS_EINVAL; }
  return ESS_OK;
}

}  // namespace

// one (sum, sum) pair of doubles per group and slice; split_for() keeps groups * slices <= 2048 + groups
extern "C" size_t ess_norm_workspace(int32_t groups) { return (size_t)(groups > 0 ? groups + 2048 : 0) * 16; }

extern "C" int ess_instnorm_forward(const float* x, const float* residual, float* y, float* stats, int32_t planes, int32_t hw,
                                    float eps, int32_t relu, void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && stats && planes > 0 && hw > 0, "instnorm_forward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (planes >= 1024 || hw < 16384) {  // enough planes to fill the chip: one fused workgroup per plane
    hipLaunchKernelGGL(instnorm_fwd_kernel, dim3(planes), dim3(NT), 0, st, x, residual, y, stats, hw, eps, relu);
    return ess_launch_status("instnorm_forward");
  }
  int rc = zero_ws(workspace, ess_norm_workspace(planes), workspace_bytes, st, "instnorm_forward");
  if (rc) return rc;
  hipLaunchKernelGGL((group_reduce_kernel<0>), dim3(planes, split_for(planes, hw)), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                     (double*)workspace, hw, 1, 0, 0, 0);
  hipLaunchKernelGGL(instnorm_apply_kernel, dim3(planes, chunks_for(planes, hw)), dim3(256), 0, st, x, residual, y, stats,
                     (const double*)workspace, split_for(planes, hw), hw, eps, relu);
  return ess_launch_status("instnorm_forward(split)");
}

extern "C" int ess_instnorm_backward(const float* x, const float* dy, const float* stats, float* dx, int32_t planes, int32_t hw,
                                     int32_t relu, void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(x && dy && stats && dx && planes > 0 && hw > 0, "instnorm_backward: bad arguments");
  ESS_CHECK_ARG(relu == 0 || relu == 1, "instnorm_backward: relu-after-residual (2) is forward only");
  hipStream_t st = (hipStream_t)stream;
  if (planes >= 1024 || hw < 16384) {
    hipLaunchKernelGGL(instnorm_bwd_kernel, dim3(planes), dim3(NT), 0, st, x, dy, stats, dx, hw, relu);
    return ess_launch_status("instnorm_backward");
  }
  int rc = zero_ws(workspace, ess_norm_workspace(planes), workspace_bytes, st, "instnorm_backward");
  if (rc) return rc;
  hipLaunchKernelGGL((group_reduce_kernel<1>), dim3(planes, split_for(planes, hw)), dim3(256), 0, st, x, nullptr, dy, stats,
                     (double*)workspace, hw, 1, 0, relu, 0);
  hipLaunchKernelGGL(instnorm_bwd_apply_kernel, dim3(planes, chunks_for(planes, hw)), dim3(256), 0, st, x, dy, stats,
                     (const double*)workspace, split_for(planes, hw), dx, hw, relu);
  return ess_launch_status("instnorm_backward(split)");
}

extern "C" int ess_batchnorm_train_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float momentum, float eps, float* y,
                                           float* stats, int32_t N, int32_t C, int32_t hw, int32_t relu, void* workspace,
                                           size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(x && gamma && beta && y && stats && N > 0 && C > 0 && hw > 0, "batchnorm_train_forward: bad arguments");
  ESS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "batchnorm_train_forward: running stats come in pairs");
  hipStream_t st = (hipStream_t)stream;
  int rc = zero_ws(workspace, ess_norm_workspace(C), workspace_bytes, st, "batchnorm_train_forward");
  if (rc) return rc;
  hipLaunchKernelGGL((group_reduce_kernel<0>), dim3(C, split_for(C, hw)), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                     (double*)workspace, hw, N, C, 0, 0);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(N * C, chunks_for(N * C, hw)), dim3(256), 0, st, x, residual, gamma, beta, running_mean,
                     running_var, momentum, eps, y, stats, (const double*)workspace, split_for(C, hw), N, C, hw, relu);
  return ess_launch_status("batchnorm_train_forward");
}

extern "C" int ess_batchnorm_train_backward(const float* x, const float* y, const float* dy, const float* gamma,
                                            const float* stats, float* dx, float* d_residual, float* dgamma, float* dbeta,
                                            int32_t accumulate, int32_t N, int32_t C, int32_t hw, int32_t relu, void* workspace,
                                            size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && dy && gamma && stats && N > 0 && C > 0 && hw > 0, "batchnorm_train_backward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int rc = zero_ws(workspace, ess_norm_workspace(C), workspace_bytes, st, "batchnorm_train_backward");
  if (rc) return rc;
  hipLaunchKernelGGL((group_reduce_kernel<1>), dim3(C, split_for(C, hw)), dim3(256), 0, st, x, y, dy, stats, (double*)workspace, hw,
                     N, C, relu, 1);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(N * C, chunks_for(N * C, hw)), dim3(256), 0, st, x, y, dy, gamma, stats,
                     (const double*)workspace, split_for(C, hw), dx, d_residual, dgamma, dbeta, accumulate, N, C, hw, relu);
  return ess_launch_status("batchnorm_train_backward");
}

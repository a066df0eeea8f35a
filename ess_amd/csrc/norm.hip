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

}  // namespace

extern "C" int ess_instnorm_forward(const float* x, const float* residual, float* y, float* stats, int32_t planes, int32_t hw,
                                    float eps, int32_t relu, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && stats && planes > 0 && hw > 0, "instnorm_forward: bad arguments");
  hipLaunchKernelGGL(instnorm_fwd_kernel, dim3(planes), dim3(NT), 0, (hipStream_t)stream, x, residual, y, stats, hw, eps, relu);
  return ess_launch_status("instnorm_forward");
}

extern "C" int ess_instnorm_backward(const float* x, const float* dy, const float* stats, float* dx, int32_t planes, int32_t hw,
                                     int32_t relu, ess_stream_t stream) {
  ESS_CHECK_ARG(x && dy && stats && dx && planes > 0 && hw > 0, "instnorm_backward: bad arguments");
  ESS_CHECK_ARG(relu == 0 || relu == 1, "instnorm_backward: relu-after-residual (2) is forward only");
  hipLaunchKernelGGL(instnorm_bwd_kernel, dim3(planes), dim3(NT), 0, (hipStream_t)stream, x, dy, stats, dx, hw, relu);
  return ess_launch_status("instnorm_backward");
}

extern "C" int ess_batchnorm_train_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float momentum, float eps, float* y,
                                           float* stats, int32_t N, int32_t C, int32_t hw, int32_t relu, ess_stream_t stream) {
  ESS_CHECK_ARG(x && gamma && beta && y && stats && N > 0 && C > 0 && hw > 0, "batchnorm_train_forward: bad arguments");
  ESS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "batchnorm_train_forward: running stats come in pairs");
  hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(C), dim3(NT), 0, (hipStream_t)stream, x, residual, gamma, beta, running_mean,
                     running_var, momentum, eps, y, stats, N, C, hw, relu);
  return ess_launch_status("batchnorm_train_forward");
}

extern "C" int ess_batchnorm_train_backward(const float* x, const float* y, const float* dy, const float* gamma,
                                            const float* stats, float* dx, float* d_residual, float* dgamma, float* dbeta,
                                            int32_t accumulate, int32_t N, int32_t C, int32_t hw, int32_t relu,
                                            ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && dy && gamma && stats && N > 0 && C > 0 && hw > 0, "batchnorm_train_backward: bad arguments");
  hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(C), dim3(NT), 0, (hipStream_t)stream, x, y, dy, gamma, stats, dx, d_residual,
                     dgamma, dbeta, accumulate, N, C, hw, relu);
  return ess_launch_status("batchnorm_train_backward");
}

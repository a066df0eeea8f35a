// Fused loss kernels (forward value + gradient w.r.t. the first argument), the flat RAdam update
// and the argmax/confusion epilogue.  All HBM-bound; each reads its inputs once or twice.
#include "common.h"

namespace {

constexpr int KMAX = 32;  // max classes held in registers
constexpr int TL_SLOTS = 2 + 3 * KMAX;   // doubles per partial-sum record of the task loss (totals first, then one record per workgroup)
constexpr int TL_MAX_BLOCKS = 2048;

// ------------------------------------------------------------------ TaskLoss = Dice + CE
// ws (double): [0] ce_sum, [1] valid_count, [2+3k] I_k = sum p_k t_k, [3+3k] sum p_k^2, [4+3k] sum t_k
// KM = compile-time class capacity: per-pixel class vectors stay in registers (no runtime indexing).
template <int KM, bool GRAD>
__global__ __launch_bounds__(256) void task_loss_kernel(const float* __restrict__ z, const int64_t* __restrict__ lab,
                                                        double* ws, float* loss, float* __restrict__ dz, float scale, int N,
                                                        int K, int hw, int ignore, int use_dice, int use_ce) {
  __shared__ double sh[2 + 3 * KMAX];
  __shared__ float coefA[KMAX], coefB[KMAX];  // dice: dL/dp_k = coefA_k * t_k + coefB_k * p_k
  __shared__ float inv_valid;
  if (!GRAD) {
    for (int i = threadIdx.x; i < 2 + 3 * K; i += blockDim.x) sh[i] = 0;
  } else {
    if (threadIdx.x < K) {
      const int k = threadIdx.x;
      const double num = 2 * ws[2 + 3 * k] + 1, den = ws[3 + 3 * k] + ws[4 + 3 * k] + 1;
      coefA[k] = (k == ignore) ? 0.f : (float)(-2.0 / den / K);
      coefB[k] = (k == ignore) ? 0.f : (float)(2.0 * num / (den * den) / K);
    }
    if (threadIdx.x == 0) inv_valid = (float)(1.0 / ws[1]);
  }
  __syncthreads();
  const size_t total = (size_t)N * hw;
  double ce = 0, cnt = 0;
  float accI[KM], accP[KM], accT[KM];
  if (!GRAD) {
#pragma unroll
    for (int k = 0; k < KM; ++k) { accI[k] = 0.f; accP[k] = 0.f; accT[k] = 0.f; }
  }
  // block-uniform trip count: every lane stays in the loop, `active` masks the tail
  for (size_t base = (size_t)blockIdx.x * blockDim.x; base < total; base += (size_t)gridDim.x * blockDim.x) {
    const size_t i = base + threadIdx.x;
    const bool active = i < total;
    const size_t ii = active ? i : total - 1;
    const size_t n = ii / hw, px = ii - n * hw;
    const float* zp = z + n * (size_t)K * hw + px;
    const int l = (int)lab[ii];
    const bool valid = active && l != ignore;
    float v[KM];
    float m = -INFINITY, zl = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (k < K) {
        v[k] = zp[(size_t)k * hw];
        m = fmaxf(m, v[k]);
        if (k == l) zl = v[k];
      } else {
        v[k] = -INFINITY;
      }
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      v[k] = (k < K) ? expf(v[k] - m) : 0.f;
      se += v[k];
    }
    const float inv = 1.f / se;
    if (!GRAD) {
      if (valid) {
        cnt += 1;
        ce += (double)(logf(se) - (zl - m));  // -log softmax at the label
#pragma unroll
        for (int k = 0; k < KM; ++k) {
          const float p = v[k] * inv;
          accP[k] += p * p;
          if (k == l) { accI[k] += p; accT[k] += 1.f; }
        }
      }
    } else if (active) {
      float* gp = dz + n * (size_t)K * hw + px;
      // a_k = dL/dp_k ; dz_j = p_j (a_j - sum_k a_k p_k) + CE term; ignored pixels get zero
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        if (k < K) {
          const float p = v[k] * inv;
          const float a = use_dice ? coefA[k] * (l == k ? 1.f : 0.f) + coefB[k] * p : 0.f;
          dot += a * p;
          v[k] = p;
        }
      }
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        if (k < K) {
          const float p = v[k];
          const float a = use_dice ? coefA[k] * (l == k ? 1.f : 0.f) + coefB[k] * p : 0.f;
          float g = p * (a - dot);
          if (use_ce) g += (p - (l == k ? 1.f : 0.f)) * inv_valid;
          gp[(size_t)k * hw] = valid ? g * scale : 0.f;
        }
      }
    }
  }
  if (!GRAD) {
    ce = wave_sum_d(ce);
    cnt = wave_sum_d(cnt);
    const bool lead = (threadIdx.x & 63) == 0;
    if (lead) { atomicAdd(&sh[0], ce); atomicAdd(&sh[1], cnt); }
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (k < K) {
        const float a = wave_sum(accI[k]), b = wave_sum(accP[k]), c = wave_sum(accT[k]);
        if (lead) { atomicAdd(&sh[2 + 3 * k], (double)a); atomicAdd(&sh[3 + 3 * k], (double)b); atomicAdd(&sh[4 + 3 * k], (double)c); }
      }
    }
    __syncthreads();
    // this workgroup's partial sums (no global atomics, no memset in front: task_loss_reduce_kernel adds them in workgroup order)
    double* part = ws + TL_SLOTS + (size_t)blockIdx.x * TL_SLOTS;
    for (int i = threadIdx.x; i < 2 + 3 * K; i += blockDim.x) part[i] = sh[i];
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    double L = 0;
    if (use_dice) {
      for (int k = 0; k < K; ++k) {
        if (k == ignore) continue;
        L += 1.0 - (2 * ws[2 + 3 * k] + 1) / (ws[3 + 3 * k] + ws[4 + 3 * k] + 1);
      }
      L /= K;
    }
    if (use_ce) L += ws[0] / ws[1];
    *loss = (float)(L * scale);
  }
}

// totals ws[0 .. 2 + 3K) = sum over workgroups (in workgroup order) of the partials the first pass left behind them
// (one workgroup per total -- 2 + 3K of them side by side: one workgroup walking the 35 totals in turn took 94 us, more than the pass it follows)
__global__ __launch_bounds__(256) void task_loss_reduce_kernel(double* ws, int nblocks, int K) {
  __shared__ double red[16];
  const int i = blockIdx.x;
  double a = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) a += ws[TL_SLOTS + (size_t)b * TL_SLOTS + i];
  a = block_sum_d(a, red);
  if (threadIdx.x == 0) ws[i] = a;
}

__global__ void task_loss_value_kernel(const double* ws, float* loss, float scale, int K, int ignore, int use_dice,
                                       int use_ce) {
  double L = 0;
  if (use_dice) {
    for (int k = 0; k < K; ++k) {
      if (k == ignore) continue;
      L += 1.0 - (2 * ws[2 + 3 * k] + 1) / (ws[3 + 3 * k] + ws[4 + 3 * k] + 1);
    }
    L /= K;
  }
  if (use_ce) L += ws[0] / ws[1];
  *loss = (float)(L * scale);
}

// ------------------------------------------------------------------ mean-type losses: partials + an ordered finalize
// ws (doubles): [1 + b] = partial sum of workgroup b ([0] unused).  Every workgroup stores its partial (plain store, no atomic, no
// memset in front); a one-workgroup finalize launch adds the partials IN WORKGROUP ORDER -- a value independent of scheduling, where
// the earlier atomicAdd on one double was not -- and writes the loss.  Two graph nodes per loss term instead of three (memset +
// kernel + finalize; a node boundary of the replayed step costs 1.55 us, profiles/r5_graph_gap_probe.txt).
// Measured and dropped (round 5): finishing inside the ONE launch -- the workgroup that draws the last ticket of an arrival counter
// adds the partials -- needs a device-scope release per workgroup, and on this part that is an L2 write-back in front of every ticket
// while the L2 is full of the gradient the kernel has just written: l1_c8_kernel 45 -> 117 us, sym_js_kernel 97 -> 183 us per
// launch (profiles/r5b_uda_bf16_eager_kernel_stats.txt before the fix), + 0.65 ms per step for 22 nodes saved.
constexpr int LOSS_MAX_BLOCKS = 2048;  // = the grid cap of the launches below (wave_uniform_grid(.., LOSS_MAX_BLOCKS)) = partial slots of ESS_LOSS_WORKSPACE_BYTES
static_assert(8 * (1 + LOSS_MAX_BLOCKS) == ESS_LOSS_WORKSPACE_BYTES, "loss workspace contract");
__device__ __forceinline__ void mean_partial(double block_acc, double* ws) {
  if (threadIdx.x == 0) ws[1 + blockIdx.x] = block_acc;
}
__global__ __launch_bounds__(256) void mean_finalize_kernel(const double* ws, float* loss, double denom, float scale, int nblocks) {
  __shared__ double red[16];
  double a = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) a += ws[1 + i];
  a = block_sum_d(a, red);
  if (threadIdx.x == 0) *loss = (float)(a / denom * scale);
}

// ------------------------------------------------------------------ symmetric JS (as two mean-KL terms)
template <int KM>
__global__ __launch_bounds__(256) void sym_js_kernel(const float* __restrict__ za, const float* __restrict__ zb, double* ws,
                                                     float* __restrict__ da, float scale, int N, int K, int hw) {
  __shared__ double red[16];
  const size_t total = (size_t)N * hw;
  const float invM = 1.f / ((float)total * K);
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hw, px = i - n * hw;
    const float* ap = za + n * (size_t)K * hw + px;
    const float* bp = zb + n * (size_t)K * hw + px;
    float va[KM], vb[KM];
    float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      va[k] = (k < K) ? ap[(size_t)k * hw] : -INFINITY;
      vb[k] = (k < K) ? bp[(size_t)k * hw] : -INFINITY;
      ma = fmaxf(ma, va[k]); mb = fmaxf(mb, vb[k]);
    }
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      va[k] = (k < K) ? expf(va[k] - ma) : 0.f;
      vb[k] = (k < K) ? expf(vb[k] - mb) : 0.f;
      sa += va[k]; sb += vb[k];
    }
    const float ia = 1.f / sa, ib = 1.f / sb;
    float dot = 0.f, l = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (k >= K) continue;
      const float pa_raw = va[k] * ia, pb_raw = vb[k] * ib;
      const float pa = fmaxf(pa_raw, 1e-10f), pb = fmaxf(pb_raw, 1e-10f);
      const float la = logf(pa), lb = logf(pb);
      l += 0.5f * (pb * (lb - la) + pa * (la - lb));
      // d/dpa of 0.5*[pb(lb - la) + pa(la - lb)] = 0.5*(-pb/pa + la - lb + 1); zero where clamped
      const float g = pa_raw > 1e-10f ? 0.5f * (-pb / pa + la - lb + 1.f) : 0.f;
      vb[k] = g;
      va[k] = pa_raw;
      dot += g * pa_raw;
    }
    acc += l;
    if (da) {
      float* gp = da + n * (size_t)K * hw + px;
#pragma unroll
      for (int k = 0; k < KM; ++k)
        if (k < K) gp[(size_t)k * hw] = va[k] * (vb[k] - dot) * invM * scale;
    }
  }
  acc = block_sum_d(acc, red);
  mean_partial(acc, ws);
}

__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, double* ws,
                                                 float* __restrict__ da, float scale, int64_t n) {
  __shared__ double red[16];
  const float gs = scale / (float)n;
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    acc += fabsf(d);
    if (da) da[i] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
  }
  acc = block_sum_d(acc, red);
  mean_partial(acc, ws);
}

// 16-byte variant (n % 4 == 0, aligned pointers): same per-element arithmetic, a quarter of the memory instructions
__global__ __launch_bounds__(256) void l1_x4_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, double* ws,
                                                    f32x4* __restrict__ da, float scale, int64_t n4, int64_t n) {
  __shared__ double red[16];
  const float gs = scale / (float)n;
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 va = a[i], vb = b[i];
    f32x4 g;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = va[j] - vb[j];
      part += fabsf(d);
      g[j] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    }
    acc += part;
    if (da) da[i] = g;
  }
  acc = block_sum_d(acc, red);
  mean_partial(acc, ws);
}

// L1 over BF16_C8 tensors (the latent / intermediate-prediction cycle losses of the bf16 configuration): 8 elements per 16-byte
// vector, fp32 differences, the gradient sign(a - b) * scale / n written as BF16_C8.  Padded tail channels are zero in both
// operands: they add nothing to the sum and get a zero gradient; `n` is the number of REAL elements.
__global__ __launch_bounds__(256) void l1_c8_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, double* ws,
                                                    uint4* __restrict__ da, float scale, int64_t nvec, int64_t n) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  __shared__ double red[16];
  const float gs = scale / (float)n;
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 va = a[i], vb = b[i];
    const unsigned ua[4] = {va.x, va.y, va.z, va.w}, ub[4] = {vb.x, vb.y, vb.z, vb.w};
    bf16x8 g;
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d0 = __builtin_bit_cast(float, ua[q] << 16) - __builtin_bit_cast(float, ub[q] << 16);
      const float d1 = __builtin_bit_cast(float, ua[q] & 0xffff0000u) - __builtin_bit_cast(float, ub[q] & 0xffff0000u);
      part += fabsf(d0) + fabsf(d1);
      g[2 * q] = (__bf16)(d0 > 0.f ? gs : (d0 < 0.f ? -gs : 0.f));
      g[2 * q + 1] = (__bf16)(d1 > 0.f ? gs : (d1 < 0.f ? -gs : 0.f));
    }
    acc += part;
    if (da) da[i] = __builtin_bit_cast(uint4, g);
  }
  acc = block_sum_d(acc, red);
  mean_partial(acc, ws);
}

// ------------------------------------------------------------------ RAdam over a flat buffer
__global__ void radam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float neg_step, float b1, float b2, float eps, int rect) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float vi = v[i] * b2 + (1.f - b2) * (gi * gi);
    const float mi = m[i] * b1 + (1.f - b1) * gi;
    v[i] = vi;
    m[i] = mi;
    p[i] = rect ? p[i] + neg_step * (mi / (sqrtf(vi) + eps)) : p[i] + neg_step * mi;
  }
}

// ------------------------------------------------------------------ argmax + confusion
__global__ __launch_bounds__(256) void argmax_conf_kernel(const float* __restrict__ z, const int64_t* __restrict__ lab,
                                                          int64_t* __restrict__ pred, unsigned long long* conf, int N, int K,
                                                          int hw, int ignore) {
  extern __shared__ unsigned int hist[];  // K*K
  for (int i = threadIdx.x; i < K * K; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const size_t total = (size_t)N * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hw, px = i - n * hw;
    const float* zp = z + n * (size_t)K * hw + px;
    float best = zp[0];
    int bi = 0;
    for (int k = 1; k < K; ++k) {
      const float t = zp[(size_t)k * hw];
      if (t > best) { best = t; bi = k; }  // first maximum wins, as torch.argmax
    }
    if (pred) pred[i] = bi;
    if (lab && conf) {
      const int64_t l = lab[i];
      if (l != ignore && l >= 0 && l < K) atomicAdd(&hist[l * K + bi], 1u);
    }
  }
  __syncthreads();
  if (conf)
    for (int i = threadIdx.x; i < K * K; i += blockDim.x)
      if (hist[i]) atomicAdd(conf + i, (unsigned long long)hist[i]);
}

// confusion histogram of GIVEN predictions (evaluation/metrics.py:4-24: bincount of label * K + prediction over the
// non-ignored pixels): the same LDS-privatised histogram as above without the argmax
__global__ __launch_bounds__(256) void label_conf_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ lab,
                                                         unsigned long long* conf, size_t total, int K, int ignore) {
  extern __shared__ unsigned int hist[];  // K*K
  for (int i = threadIdx.x; i < K * K; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int64_t l = lab[i], q = pred[i];
    if (l != ignore && l >= 0 && l < K && q >= 0 && q < K) atomicAdd(&hist[l * K + q], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += blockDim.x)
    if (hist[i]) atomicAdd(conf + i, (unsigned long long)hist[i]);
}

inline unsigned wave_uniform_grid(size_t total, int cap) {
  // blocks of 256 threads such that grid*256 divides the work into equal trip counts where possible
  size_t g = (total + 255) / 256;
  if (g > (size_t)cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" size_t ess_task_loss_workspace(int32_t K) { (void)K; return (size_t)TL_SLOTS * (1 + TL_MAX_BLOCKS) * sizeof(double); }

extern "C" int ess_task_loss(const float* logits, const int64_t* labels, float* loss, float* dlogits, float loss_scale,
                             int32_t N, int32_t K, int32_t hw, int32_t ignore_index, int32_t use_dice, int32_t use_ce,
                             void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(logits && labels && loss && workspace && N > 0 && hw > 0, "task_loss: bad arguments");
  ESS_CHECK_ARG(workspace_bytes >= ess_task_loss_workspace(K), "task_loss: workspace of %zu bytes, ess_task_loss_workspace(K) = %zu needed", workspace_bytes, ess_task_loss_workspace(K));
  ESS_CHECK_ARG(K > 0 && K <= KMAX, "task_loss: K=%d unsupported (max %d)", K, KMAX);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * hw;
  const unsigned grid = wave_uniform_grid(total, TL_MAX_BLOCKS);
#define ESS_TL(KM_, G_, DZ_)                                                                                      \
  hipLaunchKernelGGL((task_loss_kernel<KM_, G_>), dim3(grid), dim3(256), 0, st, logits, labels, (double*)workspace, loss, \
                     DZ_, loss_scale, N, K, hw, ignore_index, use_dice, use_ce)
  // first pass: per-workgroup partial sums (35 doubles at K = 11); then their ordered total.  (The partials were 2 + 3K global double
  // atomics per workgroup on the same 35 addresses and a memset in front: 102 us for 128 MB of input; see the mean losses above.)
  if (K <= 16) ESS_TL(16, false, nullptr); else ESS_TL(32, false, nullptr);
  hipLaunchKernelGGL(task_loss_reduce_kernel, dim3(2 + 3 * K), dim3(256), 0, st, (double*)workspace, (int)grid, K);
  if (dlogits) {
    if (K <= 16) ESS_TL(16, true, dlogits); else ESS_TL(32, true, dlogits);
  } else
#undef ESS_TL
    hipLaunchKernelGGL(task_loss_value_kernel, dim3(1), dim3(1), 0, st, (const double*)workspace, loss, loss_scale, K,
                       ignore_index, use_dice, use_ce);
  return ess_launch_status("task_loss");
}

extern "C" int ess_sym_js_loss(const float* a, const float* b, float* loss, float* da, float loss_scale, int32_t N, int32_t K,
                               int32_t hw, void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(a && b && loss && workspace && N > 0 && hw > 0, "sym_js_loss: bad arguments");
  ESS_CHECK_ARG(workspace_bytes >= (size_t)ESS_LOSS_WORKSPACE_BYTES, "sym_js_loss: workspace of %zu bytes, ESS_LOSS_WORKSPACE_BYTES = %zu needed", workspace_bytes, (size_t)ESS_LOSS_WORKSPACE_BYTES);
  ESS_CHECK_ARG(K > 0 && K <= KMAX, "sym_js_loss: K=%d unsupported (max %d)", K, KMAX);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * hw;
  if (K <= 16)
    hipLaunchKernelGGL((sym_js_kernel<16>), dim3(wave_uniform_grid(total, LOSS_MAX_BLOCKS)), dim3(256), 0, st, a, b, (double*)workspace, da,
                       loss_scale, N, K, hw);
  else
    hipLaunchKernelGGL((sym_js_kernel<32>), dim3(wave_uniform_grid(total, LOSS_MAX_BLOCKS)), dim3(256), 0, st, a, b, (double*)workspace, da,
                       loss_scale, N, K, hw);
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, loss, (double)total * K, loss_scale,
                     (int)wave_uniform_grid(total, LOSS_MAX_BLOCKS));
  return ess_launch_status("sym_js_loss");
}

extern "C" int ess_l1_loss(const float* a, const float* b, float* loss, float* da, float loss_scale, int64_t n, void* workspace,
                           size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(a && b && loss && workspace && n > 0, "l1_loss: bad arguments");
  ESS_CHECK_ARG(workspace_bytes >= (size_t)ESS_LOSS_WORKSPACE_BYTES, "l1_loss: workspace of %zu bytes, ESS_LOSS_WORKSPACE_BYTES = %zu needed", workspace_bytes, (size_t)ESS_LOSS_WORKSPACE_BYTES);
  hipStream_t st = (hipStream_t)stream;
  unsigned grid;
  if ((n & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)da)) & 15) == 0) {
    grid = wave_uniform_grid((size_t)n / 4, LOSS_MAX_BLOCKS);
    hipLaunchKernelGGL(l1_x4_kernel, dim3(grid), dim3(256), 0, st, (const f32x4*)a, (const f32x4*)b, (double*)workspace, (f32x4*)da, loss_scale,
                       n / 4, n);
  } else {
    grid = wave_uniform_grid((size_t)n, LOSS_MAX_BLOCKS);
    hipLaunchKernelGGL(l1_kernel, dim3(grid), dim3(256), 0, st, a, b, (double*)workspace, da, loss_scale, n);
  }
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, loss, (double)n, loss_scale, (int)grid);
  return ess_launch_status("l1_loss");
}

extern "C" int ess_l1_loss_c8(const void* a, const void* b, float* loss, void* da, float loss_scale, int64_t n_vectors, int64_t n,
                              void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(a && b && loss && workspace && n_vectors > 0 && n > 0 && n <= 8 * n_vectors, "l1_loss_c8: bad arguments");
  ESS_CHECK_ARG(workspace_bytes >= (size_t)ESS_LOSS_WORKSPACE_BYTES, "l1_loss_c8: workspace of %zu bytes, ESS_LOSS_WORKSPACE_BYTES = %zu needed", workspace_bytes, (size_t)ESS_LOSS_WORKSPACE_BYTES);
  ESS_CHECK_ARG(((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)da)) & 15) == 0, "l1_loss_c8: BF16_C8 tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = wave_uniform_grid((size_t)n_vectors, LOSS_MAX_BLOCKS);
  hipLaunchKernelGGL(l1_c8_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)a, (const uint4*)b, (double*)workspace, (uint4*)da, loss_scale,
                     n_vectors, n);
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, loss, (double)n, loss_scale, (int)grid);
  return ess_launch_status("l1_loss_c8");
}

// the same update with the step-dependent scalars read from DEVICE memory (hyper[0] = -step_size * lr, hyper[1] != 0: rectified
// Adam phase): a captured step (hipGraph) replays with the values the host wrote for THIS step, kernel arguments would be frozen
__global__ void radam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 int64_t n, const float* __restrict__ hyper, float b1, float b2, float eps) {
  const float neg_step = hyper[0];
  const bool rect = hyper[1] != 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float vi = v[i] * b2 + (1.f - b2) * (gi * gi);
    const float mi = m[i] * b1 + (1.f - b1) * gi;
    v[i] = vi;
    m[i] = mi;
    p[i] = rect ? p[i] + neg_step * (mi / (sqrtf(vi) + eps)) : p[i] + neg_step * mi;
  }
}

extern "C" int ess_radam_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2,
                                  float eps, const float* hyper, ess_stream_t stream) {
  ESS_CHECK_ARG(p && g && exp_avg && exp_avg_sq && hyper && n > 0, "radam_step_dev: bad arguments");
  hipLaunchKernelGGL(radam_dev_kernel, dim3(wave_uniform_grid((size_t)n, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg,
                     exp_avg_sq, n, hyper, beta1, beta2, eps);
  return ess_launch_status("radam_step_dev");
}

extern "C" int ess_radam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, float step_size, int32_t n_sma_ge5, ess_stream_t stream) {
  ESS_CHECK_ARG(p && g && exp_avg && exp_avg_sq && n > 0, "radam_step: bad arguments");
  const float neg_step = (float)(-(double)step_size * (double)lr);
  hipLaunchKernelGGL(radam_kernel, dim3(wave_uniform_grid((size_t)n, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg,
                     exp_avg_sq, n, neg_step, beta1, beta2, eps, n_sma_ge5);
  return ess_launch_status("radam_step");
}

extern "C" int ess_argmax_confusion(const float* logits, const int64_t* labels, int64_t* pred_lbl, int64_t* conf, int32_t N,
                                    int32_t K, int32_t hw, int32_t ignore_index, ess_stream_t stream) {
  ESS_CHECK_ARG(logits && N > 0 && K > 0 && hw > 0, "argmax_confusion: bad arguments");
  ESS_CHECK_ARG(K <= 64, "argmax_confusion: K=%d too large", K);
  ESS_CHECK_ARG((conf == nullptr) || labels, "argmax_confusion: confusion needs labels");
  hipLaunchKernelGGL(argmax_conf_kernel, dim3(wave_uniform_grid((size_t)N * hw, 1024)), dim3(256), K * K * sizeof(unsigned),
                     (hipStream_t)stream, logits, labels, pred_lbl, (unsigned long long*)conf, N, K, hw, ignore_index);
  return ess_launch_status("argmax_confusion");
}

extern "C" int ess_label_confusion(const int64_t* pred_lbl, const int64_t* labels, int64_t* conf, int64_t total, int32_t K,
                                   int32_t ignore_index, ess_stream_t stream) {
  ESS_CHECK_ARG(pred_lbl && labels && conf && total > 0 && K > 0, "label_confusion: bad arguments");
  ESS_CHECK_ARG(K <= 64, "label_confusion: K=%d too large", K);
  hipLaunchKernelGGL(label_conf_kernel, dim3(wave_uniform_grid((size_t)total, 1024)), dim3(256), K * K * sizeof(unsigned),
                     (hipStream_t)stream, pred_lbl, labels, (unsigned long long*)conf, (size_t)total, K, ignore_index);
  return ess_launch_status("label_confusion");
}

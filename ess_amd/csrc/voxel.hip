// Event stream -> voxel grid on the device (the step in front of the recurrent encoder).
//
// Reference behaviour restated here, NOT its code:
//   * trilinear flavour  -- VoxelGrid.convert, DSEC/dataset/representations.py:15-55 (one slice per call there, on a
//     DataLoader worker with put_(accumulate=True)); driven per time slice by sequence.py:144-154,202-208.
//   * temporal flavour   -- generate_voxel_grid, datasets/data_util.py:54-126 (np.add.at on the host).
//   * per-slice normalisation over the non-zero voxels -- representations.py:45-53 (unbiased std, std > 0 guard) and
//     normalize_voxel_grid, data_util.py:38-51 (population formula, no guard).
//
// Layout: all slices of a batch (B sequences x T slices) are converted by ONE launch.  Events are concatenated
// structure-of-arrays (x[], y[], pol[], t[]); slice s owns [offs[s], offs[s+1]).  grid.y = slice, grid.x strides over the
// slice's events, one thread per event: 16 B read + up to 8 (trilinear) / 2 (temporal) fp32 atomic adds that resolve
// in the L2 of the XCD owning the line -- an L2-atomic-bound kernel, no LDS tiling applies (events are unordered in
// space).  The per-event arithmetic is kept in the reference's operation order and precision (fp32 for the trilinear
// flavour, fp64 weights cast to fp32 for the temporal one), so each contribution is bit-identical; only the order of
// the floating-point additions into a voxel differs from the sequential host loop.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void voxel_trilinear_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ pol, const float* __restrict__ t,
                                                              const int64_t* __restrict__ offs, int C, int H, int W,
                                                              float* __restrict__ out) {
  const int s = blockIdx.y;
  const int64_t e0 = offs[s], e1 = offs[s + 1];
  if (e1 <= e0) return;
  const float tf = t[e0], tl = t[e1 - 1];
  float* g = out + (size_t)s * C * H * W;
  const float cm1 = (float)(C - 1), den = tl - tf;
  for (int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (int64_t)gridDim.x * blockDim.x) {
    const float xe = x[e], ye = y[e];
    const float tn = (cm1 * (t[e] - tf)) / den;  // (C-1) * (t - t0) / (t_last - t0), left to right
    if (!(tn == tn)) continue;                   // 0/0 on a one-timestamp slice: int(NaN) lands outside every bin
    const float value = 2.f * pol[e] - 1.f;
    const int x0 = (int)xe, y0 = (int)ye, t0 = (int)tn;  // truncation toward zero, like Tensor.int()
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xl = x0 + dx;
      if (xl < 0 || xl >= W) continue;
      const float wx = value * (1.f - fabsf((float)xl - xe));
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int yl = y0 + dy;
        if (yl < 0 || yl >= H) continue;
        const float wxy = wx * (1.f - fabsf((float)yl - ye));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int tb = t0 + dt;
          if (tb < 0 || tb >= C) continue;
          atomicAdd(g + ((size_t)tb * H + yl) * W + xl, wxy * (1.f - fabsf((float)tb - tn)));
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void voxel_temporal_kernel(const int32_t* __restrict__ x, const int32_t* __restrict__ y,
                                                             const double* __restrict__ t, const float* __restrict__ pol,
                                                             const int64_t* __restrict__ offs, int nb, int H, int W,
                                                             int separate, float* __restrict__ out) {
  const int s = blockIdx.y;
  const int64_t e0 = offs[s], e1 = offs[s + 1];
  if (e1 <= e0) return;
  const double first = t[e0];
  double dT = t[e1 - 1] - first;
  if (dT == 0) dT = 1.0;
  const size_t plane = (size_t)H * W;
  float* g = out + (size_t)s * (separate ? 2 : 1) * nb * plane;
  for (int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (int64_t)gridDim.x * blockDim.x) {
    const int xs = x[e], ys = y[e];
    const double ts = ((double)(nb - 1) * (t[e] - first)) / dT;
    if (!(xs < W && xs >= 0 && ys < H && ys >= 0 && ts >= 0 && ts < nb)) continue;
    float p = pol[e];
    if (p == 0.f) p = -1.f;  // polarity is +1 / -1 (data_util.py:85)
    const int ti = (int)ts;
    const double dts = ts - ti, ap = fabs((double)p);
    const float left = (float)(ap * (1.0 - dts)), right = (float)(ap * dts);
    const bool positive = p == 1.f;
    // separate_pol: channels [0,nb) positive, [nb,2nb) negative; otherwise one grid = positive - negative
    float* gp = g + (separate && !positive ? (size_t)nb * plane : 0) + (size_t)ys * W + xs;
    const float sign = (!separate && !positive) ? -1.f : 1.f;
    if (ti < nb) atomicAdd(gp + (size_t)ti * plane, sign * left);
    if (ti + 1 < nb) atomicAdd(gp + (size_t)(ti + 1) * plane, sign * right);
  }
}

// per-slice statistics of the non-zero voxels: ws[s] = {count, sum, sum of squares} (fp64)
__global__ __launch_bounds__(256) void voxel_stats_kernel(const float* __restrict__ g, int64_t per_slice, double* __restrict__ ws) {
  __shared__ double red[16];
  const int s = blockIdx.y;
  const float* p = g + (size_t)s * per_slice;
  double n = 0, sum = 0, sq = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = p[i];
    if (v != 0.f) { n += 1; sum += v; sq += (double)v * v; }
  }
  n = block_sum_d(n, red);
  sum = block_sum_d(sum, red);
  sq = block_sum_d(sq, red);
  if (threadIdx.x == 0 && n > 0) {
    atomicAdd(ws + 3 * s, n);
    atomicAdd(ws + 3 * s + 1, sum);
    atomicAdd(ws + 3 * s + 2, sq);
  }
}

// mode 0: VoxelGrid(normalize=True): mean / UNBIASED std of the non-zero voxels, (v-mean)/std if std > 0 else v-mean
// mode 1: normalize_voxel_grid: mean, sqrt(E[v^2] - mean^2), mask * (v - mean) / std (no guard)
__global__ __launch_bounds__(256) void voxel_apply_kernel(float* __restrict__ g, int64_t per_slice, const double* __restrict__ ws,
                                                          int mode) {
  const int s = blockIdx.y;
  const double n = ws[3 * s];
  if (n <= 0) return;
  const double mean = ws[3 * s + 1] / n;
  float fm, fs;
  bool divide = true;
  if (mode == 0) {
    // torch.std(): sqrt(sum((v-mean)^2) / (n-1)); one non-zero voxel -> NaN -> "std > 0" is false -> subtract only
    const double var = n > 1 ? (ws[3 * s + 2] - n * mean * mean) / (n - 1) : -1.0;
    const double sd = var > 0 ? sqrt(var) : 0.0;
    divide = sd > 0;
    fm = (float)mean;
    fs = (float)sd;
  } else {
    fm = (float)mean;
    fs = sqrtf((float)(ws[3 * s + 2] / n) - fm * fm);
  }
  float* p = g + (size_t)s * per_slice;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = p[i];
    if (v != 0.f) p[i] = divide ? (v - fm) / fs : v - fm;
  }
}

int check_slices(const char* what, const void* a, const void* b, const void* c, const void* d, const void* offs, int n_slices,
                 int C, int H, int W, const void* out, int64_t n_events) {
  ESS_CHECK_ARG(offs && out && n_slices > 0 && C > 0 && H > 0 && W > 0, "%s: bad arguments", what);
  ESS_CHECK_ARG(n_events >= 0 && (n_events == 0 || (a && b && c && d)), "%s: null event arrays", what);
  ESS_CHECK_ARG(n_slices <= 65535, "%s: at most 65535 slices per call", what);
  return ESS_OK;
}

dim3 slice_grid(int64_t n_events, int n_slices) {
  // ~4 events per thread keeps every CU busy at 100k events/slice without oversubscribing tiny slices
  int64_t bx = ceil_div64(n_events / (n_slices > 0 ? n_slices : 1) + 1, 1024);
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  return dim3((unsigned)bx, (unsigned)n_slices);
}

}  // namespace

extern "C" int ess_voxel_grid_trilinear(const float* x, const float* y, const float* pol, const float* t, const int64_t* slice_offsets,
                                        int64_t n_events, int n_slices, int channels, int height, int width, float* out,
                                        ess_stream_t stream) {
  int rc = check_slices("voxel_grid_trilinear", x, y, pol, t, slice_offsets, n_slices, channels, height, width, out, n_events);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, (size_t)n_slices * channels * height * width * sizeof(float), st);
  if (e != hipSuccess) { ess_set_error("voxel_grid_trilinear: memset failed: %s", hipGetErrorString(e)); return ESS_ELAUNCH; }
  if (n_events == 0) return ESS_OK;
  hipLaunchKernelGGL(voxel_trilinear_kernel, slice_grid(n_events, n_slices), dim3(256), 0, st, x, y, pol, t, slice_offsets, channels,
                     height, width, out);
  return ess_launch_status("voxel_grid_trilinear");
}

extern "C" int ess_voxel_grid_temporal(const int32_t* x, const int32_t* y, const double* t, const float* pol,
                                       const int64_t* slice_offsets, int64_t n_events, int n_slices, int bins, int height, int width,
                                       int separate_pol, float* out, ess_stream_t stream) {
  int rc = check_slices("voxel_grid_temporal", x, y, t, pol, slice_offsets, n_slices, bins, height, width, out, n_events);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t per = (size_t)(separate_pol ? 2 : 1) * bins * height * width;
  hipError_t e = hipMemsetAsync(out, 0, (size_t)n_slices * per * sizeof(float), st);
  if (e != hipSuccess) { ess_set_error("voxel_grid_temporal: memset failed: %s", hipGetErrorString(e)); return ESS_ELAUNCH; }
  if (n_events == 0) return ESS_OK;
  hipLaunchKernelGGL(voxel_temporal_kernel, slice_grid(n_events, n_slices), dim3(256), 0, st, x, y, t, pol, slice_offsets, bins, height,
                     width, separate_pol, out);
  return ess_launch_status("voxel_grid_temporal");
}

extern "C" size_t ess_voxel_normalize_workspace(int n_slices) { return (size_t)(n_slices > 0 ? n_slices : 0) * 3 * sizeof(double); }

extern "C" int ess_voxel_normalize(float* grid, int n_slices, int64_t elems_per_slice, int mode, void* workspace, size_t workspace_bytes,
                                   ess_stream_t stream) {
  ESS_CHECK_ARG(grid && workspace && n_slices > 0 && n_slices <= 65535 && elems_per_slice > 0, "voxel_normalize: bad arguments");
  ESS_CHECK_ARG(mode == 0 || mode == 1, "voxel_normalize: mode must be 0 (unbiased std, guarded) or 1 (population std)");
  ESS_CHECK_ARG(workspace_bytes >= ess_voxel_normalize_workspace(n_slices), "voxel_normalize: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(workspace, 0, ess_voxel_normalize_workspace(n_slices), st);
  if (e != hipSuccess) { ess_set_error("voxel_normalize: memset failed: %s", hipGetErrorString(e)); return ESS_ELAUNCH; }
  int64_t bx = ceil_div64(elems_per_slice, 256 * 8);
  if (bx > 256) bx = 256;
  const dim3 grid_dim((unsigned)bx, (unsigned)n_slices);
  hipLaunchKernelGGL(voxel_stats_kernel, grid_dim, dim3(256), 0, st, grid, elems_per_slice, (double*)workspace);
  hipLaunchKernelGGL(voxel_apply_kernel, grid_dim, dim3(256), 0, st, grid, elems_per_slice, (const double*)workspace, mode);
  return ess_launch_status("voxel_normalize");
}

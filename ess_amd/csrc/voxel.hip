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
  if ((per_slice & 3) == 0 && (((uintptr_t)g) & 15) == 0) {  // 16-byte loads; zeros add nothing to either sum
    const f32x4* p4 = (const f32x4*)p;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (per_slice >> 2); i += (int64_t)gridDim.x * blockDim.x) {
      const f32x4 v = p4[i];
      float cs = 0.f, ps = 0.f;
      double pq = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { cs += v[j] != 0.f ? 1.f : 0.f; ps += v[j]; pq += (double)v[j] * v[j]; }
      n += cs; sum += ps; sq += pq;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_slice; i += (int64_t)gridDim.x * blockDim.x) {
      const float v = p[i];
      if (v != 0.f) { n += 1; sum += v; sq += (double)v * v; }
    }
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

// ------------------------------------------------------------------------------------------------------------
// Binned variant of the trilinear flavour.  The direct kernel above issues 8 device-scope fp32 atomics per event and
// sits at the rate of that path (21 G atomics/s measured, ~3 % of the HBM roofline).  Here the events of a slice are
// first grouped by the 64x32-pixel tile of their (x0, y0) corner -- counting and placement are privatised in LDS, so the
// global atomics are one per (block, non-empty tile) -- and then ONE workgroup per (slice, tile) accumulates its events
// into a [C][33][65] tile in LDS (ds_add_f32) and stores its own 64x32 cells with plain stores (256-byte row segments); the +1 halo (cells of the right / bottom / diagonal neighbours) goes to a small table that a fifth pass
// adds into the owning tiles' first column / row, again with plain read-modify-writes: no global fp32 atomic at all.
// Measured (40 slices x 100k events -> [40,2,480,640]): 1.55 ms direct -> 0.30 ms binned; of that 0.17 ms is pass 4, and there
// the ds_add_f32 themselves (32 M lane-atomics at ~0.5 per clock per CU): the write-out is 12 us, an event-free run 19 us.
// Per-event arithmetic is the same expression sequence as the direct kernel: every contribution is bit-identical, only
// the order of the additions differs (as it already does between the direct kernel and the host loop).
constexpr int VTX = 64, VTY = 32;  // tile width / height in pixels (a tile row is 256 contiguous bytes of a plane)
constexpr int VHS = VTX + VTY + 1;  // halo entries per tile and channel: right column (VTY), bottom row (VTX), corner
constexpr int VCHUNK = 4096;       // events per workgroup in the count / placement passes (16 per thread)
constexpr int VMAXT = 2048;        // tiles per slice the LDS histograms are sized for

struct VoxBin {
  const float *x, *y, *pol, *t;
  const int64_t* offs;
  int C, H, W, tiles_x, ntl;
  int* counts;   // [S][ntl]
  int* start;    // [S][ntl]   first sorted slot of the tile (global index)
  int* cursor;   // [S][ntl]
  float4* sorted;  // [n_events] (x, y, value, t_norm)
  float* halo;     // [S][ntl][C][VHS]: right column, bottom row, corner of every tile's +1 halo
};

// tile of an event, or -1 when no corner of it can land in the grid / its time is NaN
__device__ __forceinline__ int vox_tile(const VoxBin& b, float xe, float ye, float tn) {
  const int x0 = (int)xe, y0 = (int)ye;
  if (!(tn == tn) || x0 < -1 || x0 >= b.W || y0 < -1 || y0 >= b.H) return -1;
  return ((y0 < 0 ? 0 : y0) / VTY) * b.tiles_x + (x0 < 0 ? 0 : x0) / VTX;
}

// pass 1 (PLACE = false): counts[s][tile] += events;  pass 3 (PLACE = true): sorted[...] = events grouped by tile
template <bool PLACE>
__global__ __launch_bounds__(256) void voxel_bin_kernel(const VoxBin b) {
  __shared__ int hist[VMAXT];
  __shared__ int base[PLACE ? VMAXT : 1];
  const int s = blockIdx.y;
  const int64_t e0 = b.offs[s], e1 = b.offs[s + 1];
  const int64_t c0 = e0 + (int64_t)blockIdx.x * VCHUNK;
  if (c0 >= e1) return;
  const int64_t c1 = c0 + VCHUNK < e1 ? c0 + VCHUNK : e1;
  const float tf = b.t[e0], tl = b.t[e1 - 1];
  const float cm1 = (float)(b.C - 1), den = tl - tf;
  for (int i = threadIdx.x; i < b.ntl; i += 256) hist[i] = 0;
  __syncthreads();
  constexpr int PER = VCHUNK / 256;
  int tile[PER];
  float4 ev[PLACE ? PER : 1];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t e = c0 + threadIdx.x + k * 256;
    tile[k] = -1;
    if (e < c1) {
      const float xe = b.x[e], ye = b.y[e];
      const float tn = (cm1 * (b.t[e] - tf)) / den;
      tile[k] = vox_tile(b, xe, ye, tn);
      if (PLACE) ev[k] = make_float4(xe, ye, 2.f * b.pol[e] - 1.f, tn);
      if (tile[k] >= 0) atomicAdd(&hist[tile[k]], 1);
    }
  }
  __syncthreads();
  if constexpr (!PLACE) {
    for (int i = threadIdx.x; i < b.ntl; i += 256)
      if (hist[i]) atomicAdd(&b.counts[(size_t)s * b.ntl + i], hist[i]);
  } else {
    for (int i = threadIdx.x; i < b.ntl; i += 256) {
      const int h = hist[i];
      base[i] = h ? atomicAdd(&b.cursor[(size_t)s * b.ntl + i], h) : 0;  // reserve this block's range of the tile
    }
    __syncthreads();
    for (int i = threadIdx.x; i < b.ntl; i += 256) hist[i] = 0;  // reused as the rank counter inside the range
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (tile[k] >= 0) b.sorted[base[tile[k]] + atomicAdd(&hist[tile[k]], 1)] = ev[k];
  }
}

// pass 2: start[s][i] = offs[s] + exclusive prefix of counts[s][*]; cursor = start.  One workgroup per slice.
__global__ __launch_bounds__(256) void voxel_scan_kernel(const VoxBin b) {
  __shared__ int part[256];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int per = (b.ntl + 255) / 256;
  const int lo = tid * per, hi = lo + per < b.ntl ? lo + per : b.ntl;
  const int* c = b.counts + (size_t)s * b.ntl;
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += c[i];
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int run = (int)b.offs[s];
    for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  int run = part[tid];
  for (int i = lo; i < hi; ++i) {
    b.start[(size_t)s * b.ntl + i] = run;
    b.cursor[(size_t)s * b.ntl + i] = run;
    run += c[i];
  }
}

// pass 4: one workgroup per (tile, slice)
__global__ __launch_bounds__(256) void voxel_tile_kernel(const VoxBin b, float* __restrict__ out) {
  extern __shared__ float tl_s[];  // [C][VTY+1][VTX+1]
  constexpr int PX = VTX + 1, PY = VTY + 1;
  const int s = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int by = tile / b.tiles_x, bx = tile - by * b.tiles_x;
  const int n = b.counts[(size_t)s * b.ntl + tile];
  const float4* ev = b.sorted + b.start[(size_t)s * b.ntl + tile];
  const int cells = b.C * PX * PY;
  for (int i = tid; i < cells; i += 256) tl_s[i] = 0.f;
  __syncthreads();
  const int gx0 = bx * VTX, gy0 = by * VTY;
  for (int i = tid; i < n; i += 256) {
    const float4 e = ev[i];
    const float xe = e.x, ye = e.y, value = e.z, tn = e.w;
    const int x0 = (int)xe, y0 = (int)ye, t0 = (int)tn;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xl = x0 + dx;
      if (xl < 0 || xl >= b.W) continue;
      const float wx = value * (1.f - fabsf((float)xl - xe));
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int yl = y0 + dy;
        if (yl < 0 || yl >= b.H) continue;
        const float wxy = wx * (1.f - fabsf((float)yl - ye));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int tb = t0 + dt;
          if (tb < 0 || tb >= b.C) continue;
          atomicAdd(&tl_s[(tb * PY + (yl - gy0)) * PX + (xl - gx0)], wxy * (1.f - fabsf((float)tb - tn)));
        }
      }
    }
  }
  __syncthreads();
  // own cells (local x < VTX, y < VTY): plain stores -- no other workgroup writes them in this pass, zeros included, so
  // `out` needs no memset.  64 lanes = one 256-byte row segment of a plane; the 4 waves take rows round-robin.
  float* g = out + (size_t)s * b.C * b.H * b.W;
  const int lx = tid & 63, wv = tid >> 6;
  const int gx = gx0 + lx;
  for (int r = wv; r < b.C * VTY; r += 4) {
    const int c = r / VTY, ly = r - c * VTY;
    const int gy = gy0 + ly;
    if (gx < b.W && gy < b.H) g[((size_t)c * b.H + gy) * b.W + gx] = tl_s[(c * PY + ly) * PX + lx];
  }
  // the +1 halo (cells of the right / bottom / diagonal neighbours) goes to the halo table, added by pass 5
  float* hb = b.halo + ((size_t)s * b.ntl + tile) * b.C * VHS;
  for (int i = tid; i < b.C * VHS; i += 256) {
    const int c = i / VHS, k = i - c * VHS;
    const int hy = k < VTY ? k : VTY, hx = k < VTY ? VTX : (k < VTY + VTX ? k - VTY : VTX);
    hb[i] = tl_s[(c * PY + hy) * PX + hx];
  }
}

// pass 5: every tile adds what its left / top / top-left neighbours spilled into its first column / row / corner cell.
// The receiving tile owns those cells: plain read-modify-write in a fixed order (own + left + top + corner), no atomics.
// One thread per (tile, edge cell): VTY column cells, then VTX-1 row cells (cell (0,0) is the column's).
__global__ __launch_bounds__(256) void voxel_halo_kernel(const VoxBin b, float* __restrict__ out) {
  constexpr int EC = VTY + VTX - 1;
  const int s = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= b.ntl * EC) return;
  const int tile = idx / EC, l = idx - tile * EC;
  const int by = tile / b.tiles_x, bx = tile - by * b.tiles_x;
  const bool col = l < VTY;
  const int ly = col ? l : 0, lx = col ? 0 : l - VTY + 1;
  const int gx = bx * VTX + lx, gy = by * VTY + ly;
  if (gx >= b.W || gy >= b.H) return;
  const float* hall = b.halo + (size_t)s * b.ntl * b.C * VHS;
  float* g = out + (size_t)s * b.C * b.H * b.W;
  for (int c = 0; c < b.C; ++c) {
    float add = 0.f;
    bool any = false;
    if (col && bx > 0) { add += hall[((size_t)(tile - 1) * b.C + c) * VHS + ly]; any = true; }                          // left: right column
    if ((!col || ly == 0) && by > 0) { add += hall[((size_t)(tile - b.tiles_x) * b.C + c) * VHS + VTY + lx]; any = true; }  // top: bottom row
    if (col && ly == 0 && bx > 0 && by > 0) { add += hall[((size_t)(tile - b.tiles_x - 1) * b.C + c) * VHS + VTY + VTX]; any = true; }
    if (any) g[((size_t)c * b.H + gy) * b.W + gx] += add;
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

static size_t vox_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t ess_voxel_grid_trilinear_workspace(int64_t n_events, int n_slices, int height, int width) {
  if (n_events <= 0 || n_slices <= 0 || height <= 0 || width <= 0) return 0;
  const size_t ntl = (size_t)ceil_div(width, VTX) * ceil_div(height, VTY);
  // (halo table sized for the 7 channels the 64 KiB LDS tile allows)
  return 3 * vox_align((size_t)n_slices * ntl * sizeof(int)) + vox_align((size_t)n_events * sizeof(float4)) +
         vox_align((size_t)n_slices * ntl * 7 * VHS * sizeof(float));
}

extern "C" int ess_voxel_grid_trilinear(const float* x, const float* y, const float* pol, const float* t, const int64_t* slice_offsets,
                                        int64_t n_events, int n_slices, int channels, int height, int width, float* out,
                                        void* workspace, size_t workspace_bytes, int64_t max_events_per_slice, ess_stream_t stream) {
  int rc = check_slices("voxel_grid_trilinear", x, y, pol, t, slice_offsets, n_slices, channels, height, width, out, n_events);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  const int tiles_x = ceil_div(width, VTX), ntl = tiles_x * ceil_div(height, VTY);
  const size_t lds = (size_t)channels * (VTX + 1) * (VTY + 1) * sizeof(float);
  const bool binned = workspace != nullptr && n_events > 0 && ntl <= VMAXT && channels <= 7 && n_events < ((int64_t)1 << 31) &&
                      max_events_per_slice > 0;
  if (!binned) {  // direct device-scope atomics into a zeroed grid: no workspace needed
    e = hipMemsetAsync(out, 0, (size_t)n_slices * channels * height * width * sizeof(float), st);
    if (e != hipSuccess) { ess_set_error("voxel_grid_trilinear: memset failed: %s", hipGetErrorString(e)); return ESS_ELAUNCH; }
    if (n_events == 0) return ESS_OK;
    hipLaunchKernelGGL(voxel_trilinear_kernel, slice_grid(n_events, n_slices), dim3(256), 0, st, x, y, pol, t, slice_offsets, channels,
                       height, width, out);
    return ess_launch_status("voxel_grid_trilinear");
  }
  ESS_CHECK_ARG(workspace_bytes >= ess_voxel_grid_trilinear_workspace(n_events, n_slices, height, width),
                "voxel_grid_trilinear: workspace too small");
  ESS_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "voxel_grid_trilinear: workspace must be 16-byte aligned");
  VoxBin b{};
  b.x = x; b.y = y; b.pol = pol; b.t = t; b.offs = slice_offsets;
  b.C = channels; b.H = height; b.W = width; b.tiles_x = tiles_x; b.ntl = ntl;
  const size_t tab = vox_align((size_t)n_slices * ntl * sizeof(int));
  char* w = (char*)workspace;
  b.counts = (int*)w; b.start = (int*)(w + tab); b.cursor = (int*)(w + 2 * tab); b.sorted = (float4*)(w + 3 * tab);
  b.halo = (float*)(w + 3 * tab + vox_align((size_t)n_events * sizeof(float4)));
  e = hipMemsetAsync(b.counts, 0, tab, st);
  if (e != hipSuccess) { ess_set_error("voxel_grid_trilinear: memset failed: %s", hipGetErrorString(e)); return ESS_ELAUNCH; }
  const dim3 cgrid((unsigned)ceil_div64(max_events_per_slice, VCHUNK), (unsigned)n_slices);
  hipLaunchKernelGGL(voxel_bin_kernel<false>, cgrid, dim3(256), 0, st, b);
  hipLaunchKernelGGL(voxel_scan_kernel, dim3((unsigned)n_slices), dim3(256), 0, st, b);
  hipLaunchKernelGGL(voxel_bin_kernel<true>, cgrid, dim3(256), 0, st, b);
  hipLaunchKernelGGL(voxel_tile_kernel, dim3((unsigned)ntl, (unsigned)n_slices), dim3(256), lds, st, b, out);
  hipLaunchKernelGGL(voxel_halo_kernel, dim3((unsigned)ceil_div(ntl * (VTY + VTX - 1), 256), (unsigned)n_slices), dim3(256), 0, st, b, out);
  return ess_launch_status("voxel_grid_trilinear(binned)");
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
  // few workgroups per slice for the statistics: each ends in three fp64 atomics on the slice's totals
  const dim3 sgrid((unsigned)(bx > 32 ? 32 : bx), (unsigned)n_slices);
  hipLaunchKernelGGL(voxel_stats_kernel, sgrid, dim3(256), 0, st, grid, elems_per_slice, (double*)workspace);
  hipLaunchKernelGGL(voxel_apply_kernel, grid_dim, dim3(256), 0, st, grid, elems_per_slice, (const double*)workspace, mode);
  return ess_launch_status("voxel_normalize");
}

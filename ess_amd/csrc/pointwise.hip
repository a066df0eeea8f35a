// HBM-bound glue kernels: bilinear x2 (+skip sum), 2x2 sum pool, add, event normalisation.
#include "common.h"

namespace {

// y[2H][2W] = bilinear_x2(a + b), align_corners=False: source coordinate max(0, (dst+0.5)/2 - 0.5),
// second tap clamped at the border (torch upsample_bilinear2d semantics).
__global__ void up2_bilinear_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                        int planes, int H, int W) {
  const int W2 = 2 * W, H2 = 2 * H;
  const size_t total = (size_t)planes * H2 * W2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % W2;
    const size_t t = i / W2;
    const int yy = t % H2;
    const size_t pl = t / H2;
    const float sy = fmaxf(0.f, (yy + 0.5f) * 0.5f - 0.5f), sx = fmaxf(0.f, (x + 0.5f) * 0.5f - 0.5f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* pa = a + pl * H * W;
    float v00 = pa[y0 * W + x0], v01 = pa[y0 * W + x1], v10 = pa[y1 * W + x0], v11 = pa[y1 * W + x1];
    if (b) {
      const float* pb = b + pl * H * W;
      v00 += pb[y0 * W + x0]; v01 += pb[y0 * W + x1]; v10 += pb[y1 * W + x0]; v11 += pb[y1 * W + x1];
    }
    y[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
  }
}

// Same map, W even: one thread = 4 consecutive outputs of a row (one 16-byte store).  Outputs 4j..4j+3 read the source
// columns 2j-1..2j+2 of two source rows: 8 (+8) loads for 4 outputs instead of 16 (+16), one index decode instead of
// four, 32-bit arithmetic.  Per-output expression and operation order are those of the scalar kernel (bit-identical).
__global__ __launch_bounds__(256) void up2_bilinear_add_x4_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  float* __restrict__ y, int planes, int H, int W) {
  const int W2 = 2 * W, H2 = 2 * H, Q = W2 >> 2;
  const unsigned total = (unsigned)planes * H2 * Q;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned q = i % Q, t = i / Q;
    const unsigned yy = t % H2, pl = t / H2;
    const float sy = fmaxf(0.f, (yy + 0.5f) * 0.5f - 0.5f);
    const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float ly = sy - y0, hy = 1.f - ly;
    const size_t base = (size_t)pl * H * W;
    const float* r0 = a + base + (size_t)y0 * W;
    const float* r1 = a + base + (size_t)y1 * W;
    const int c[4] = {max(2 * (int)q - 1, 0), 2 * (int)q, 2 * (int)q + 1, min(2 * (int)q + 2, W - 1)};
    float u0[4], u1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { u0[k] = r0[c[k]]; u1[k] = r1[c[k]]; }
    if (b) {
      const float* s0 = b + base + (size_t)y0 * W;
      const float* s1 = b + base + (size_t)y1 * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) { u0[k] += s0[c[k]]; u1[k] += s1[c[k]]; }
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = 4 * (int)q + k;
      const float sx = fmaxf(0.f, (x + 0.5f) * 0.5f - 0.5f);
      const int x0 = (int)sx;
      const float lx = sx - x0, hx = 1.f - lx;
      // x0 / x1 of output 4q+k inside c[]: k=0 -> (2q-1, 2q) [q = 0: (0, 1) with lx = 0], k=1,2 -> (2q, 2q+1),
      // k=3 -> (2q+1, min(2q+2, W-1))
      const int i0 = k == 0 ? 0 : (k == 3 ? 2 : 1);
      const int i1 = k == 0 ? (q == 0 ? 2 : 1) : (k == 3 ? 3 : 2);
      o[k] = hy * (hx * u0[i0] + lx * u0[i1]) + ly * (hx * u1[i0] + lx * u1[i1]);
    }
    *(f32x4*)(y + ((size_t)t * W2 + 4 * q)) = o;
  }
}

// The same map with a BF16_C8 OUTPUT ([N][C/8][2H][2W][8] bfloat16): what the 5x5 decoder convolution of the frozen encoder
// stages in the bf16 configuration (it rounds its input to bf16 either way: identical MFMA operands, half the bytes written and
// a quarter of the cache-line touches read back).  One thread = 4 consecutive output pixels of a row x 8 channels = 64 contiguous
// bytes; per output the expression and operation order of the kernels above.
__global__ __launch_bounds__(256) void up2_bilinear_add_c8_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  uint4* __restrict__ y, int N, int C, int H, int W) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const int W2 = 2 * W, H2 = 2 * H, Q = W2 >> 2, CB = (C + 7) >> 3;
  const unsigned total = (unsigned)N * CB * H2 * Q;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned q = i % Q, t = i / Q;
    const unsigned yy = t % H2, g = t / H2;  // g = n * CB + cb
    const unsigned cb = g % CB, n = g / CB;
    const float sy = fmaxf(0.f, (yy + 0.5f) * 0.5f - 0.5f);
    const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float ly = sy - y0, hy = 1.f - ly;
    const int c[4] = {max(2 * (int)q - 1, 0), 2 * (int)q, 2 * (int)q + 1, min(2 * (int)q + 2, W - 1)};
    float o[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = (int)cb * 8 + j;
      const bool ok = ch < C;
      const size_t base = ((size_t)n * C + (ok ? ch : C - 1)) * H * W;
      const float* r0 = a + base + (size_t)y0 * W;
      const float* r1 = a + base + (size_t)y1 * W;
      float u0[4], u1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { u0[k] = r0[c[k]]; u1[k] = r1[c[k]]; }
      if (b) {
        const float* s0 = b + base + (size_t)y0 * W;
        const float* s1 = b + base + (size_t)y1 * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) { u0[k] += s0[c[k]]; u1[k] += s1[c[k]]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = 4 * (int)q + k;
        const float sx = fmaxf(0.f, (x + 0.5f) * 0.5f - 0.5f);
        const int x0 = (int)sx;
        const float lx = sx - x0, hx = 1.f - lx;
        const int i0 = k == 0 ? 0 : (k == 3 ? 2 : 1);
        const int i1 = k == 0 ? (q == 0 ? 2 : 1) : (k == 3 ? 3 : 2);
        const float v = hy * (hx * u0[i0] + lx * u0[i1]) + ly * (hx * u1[i0] + lx * u1[i1]);
        o[k][j] = ok ? v : 0.f;
      }
    }
    uint4* dst = y + ((size_t)g * H2 + yy) * W2 + 4 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)o[k][j];
      dst[k] = __builtin_bit_cast(uint4, v);
    }
  }
}

// The same map again, from BF16_C8 SOURCES (the staging copies the frozen encoder's kernels leave next to -- or instead of -- their
// fp32 outputs): 16 x 16-byte loads per thread instead of 128 scalar ones, half the bytes.  Per output the expression and
// operation order of the kernels above applied to the sources' bf16 values, i.e. bit-identical to feeding those kernels the
// converted tensors.
// One thread = 4 output columns of an output ROW PAIR (2 pr - 1, 2 pr), pr = 0 .. H: both rows blend the same two source rows
// (pr - 1, pr), so the 16 loads, their unpacking, the a + b sums and the horizontal blends are done once for 8 output vectors (the
// first version did them per output row: ~110 VALU and 4 loads per output vector, and the pass ran at 2.8-3.3 TB/s -- instruction-
// bound, not memory-bound).  pr = 0 / pr = H hold one real row each (0 and 2 H - 1).  Per output the expression is unchanged.
__global__ __launch_bounds__(256) void up2_bilinear_add_c8c8_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                                    uint4* __restrict__ y, int N, int CB, int H, int W) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const int W2 = 2 * W, H2 = 2 * H, Q = W2 >> 2, HP = H + 1;
  const unsigned total = (unsigned)N * CB * HP * Q;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned q = i % Q, t = i / Q;
    const int pr = (int)(t % HP);
    const unsigned g = t / HP;  // g = n * CB + cb
    // source rows of the pair: (pr - 1, pr); row 0 alone blends (0, 1) with weight 0 on the second, row 2H - 1 alone (H - 1, H - 1)
    const int ra = pr == 0 ? 0 : pr - 1, rb = pr == 0 ? (H > 1 ? 1 : 0) : min(pr, H - 1);
    const int c[4] = {max(2 * (int)q - 1, 0), 2 * (int)q, 2 * (int)q + 1, min(2 * (int)q + 2, W - 1)};
    const uint4* r0 = a + ((size_t)g * H + ra) * W;
    const uint4* r1 = a + ((size_t)g * H + rb) * W;
    uint4 va0[4], va1[4], vb0[4], vb1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { va0[k] = r0[c[k]]; va1[k] = r1[c[k]]; }
    if (b) {
      const uint4* s0 = b + ((size_t)g * H + ra) * W;
      const uint4* s1 = b + ((size_t)g * H + rb) * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) { vb0[k] = s0[c[k]]; vb1[k] = s1[c[k]]; }
    }
    float u0[4][8], u1[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bf16x8 p0 = __builtin_bit_cast(bf16x8, va0[k]), p1 = __builtin_bit_cast(bf16x8, va1[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) { u0[k][j] = (float)p0[j]; u1[k][j] = (float)p1[j]; }
      if (b) {
        const bf16x8 q0 = __builtin_bit_cast(bf16x8, vb0[k]), q1 = __builtin_bit_cast(bf16x8, vb1[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { u0[k][j] += (float)q0[j]; u1[k][j] += (float)q1[j]; }
      }
    }
    // horizontal blends of the two source rows at the 4 output columns (shared by both output rows)
    float h0[4][8], h1[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = 4 * (int)q + k;
      const float sx = fmaxf(0.f, (x + 0.5f) * 0.5f - 0.5f);
      const int x0 = (int)sx;
      const float lx = sx - x0, hx = 1.f - lx;
      const int i0 = k == 0 ? 0 : (k == 3 ? 2 : 1);
      const int i1 = k == 0 ? (q == 0 ? 2 : 1) : (k == 3 ? 3 : 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        h0[k][j] = hx * u0[i0][j] + lx * u0[i1][j];
        h1[k][j] = hx * u1[i0][j] + lx * u1[i1][j];
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int yy = 2 * pr - 1 + rr;
      if (yy < 0 || yy >= H2) continue;
      const float sy = fmaxf(0.f, (yy + 0.5f) * 0.5f - 0.5f);
      const int y0 = (int)sy;
      const float ly = sy - y0, hy = 1.f - ly;
      uint4* dst = y + ((size_t)g * H2 + yy) * W2 + 4 * q;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)(hy * h0[k][j] + ly * h1[k][j]);
        dst[k] = __builtin_bit_cast(uint4, v);
      }
    }
  }
}

// 2x2 sum pooling (backward of nearest x2).  Wo even: one thread = 2 outputs = two 16-byte loads, one 8-byte store.
__global__ void sumpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int Ho, int Wo, int acc) {
  const size_t total = (size_t)planes * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xo = i % Wo;
    const size_t t = i / Wo;
    const int yo = t % Ho;
    const size_t pl = t / Ho;
    const float* p = x + (pl * 2 * Ho + 2 * yo) * (size_t)(2 * Wo) + 2 * xo;
    const float s = (p[0] + p[1]) + (p[2 * Wo] + p[2 * Wo + 1]);
    y[i] = acc ? y[i] + s : s;
  }
}
__global__ __launch_bounds__(256) void sumpool2_x2_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned rows, int Wo,
                                                         int acc) {
  // rows = planes * Ho output rows; input row pair of output row r starts at 2 r * (2 Wo)
  const unsigned Q = Wo >> 1, total = rows * Q;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned q = i % Q, r = i / Q;
    const float* p = x + (size_t)r * 4 * Wo + 4 * q;
    const f32x4 t0 = *(const f32x4*)p, t1 = *(const f32x4*)(p + 2 * Wo);
    float2 o;
    o.x = (t0[0] + t0[1]) + (t1[0] + t1[1]);
    o.y = (t0[2] + t0[3]) + (t1[2] + t1[3]);
    float2* dst = (float2*)(y + (size_t)r * Wo + 2 * q);
    if (acc) { const float2 old = *dst; o.x = old.x + o.x; o.y = old.y + o.y; }
    *dst = o;
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
  if ((n & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y)) & 15) == 0) {  // 16-byte accesses
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += (int64_t)gridDim.x * blockDim.x)
      ((f32x4*)y)[i] = ((const f32x4*)a)[i] + ((const f32x4*)b)[i];
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a[i] + b[i];
}

// workspace: double[3] = {count, sum, sumsq}
__global__ __launch_bounds__(256) void evnorm_reduce_kernel(const float* __restrict__ x, int64_t n, double* ws) {
  __shared__ double red[16];
  double c = 0, s = 0, ss = 0;
  if ((n & 3) == 0 && (((uintptr_t)x) & 15) == 0) {  // 16-byte loads; zeros contribute nothing to either sum
    const f32x4* x4 = (const f32x4*)x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += (int64_t)gridDim.x * blockDim.x) {
      const f32x4 v = x4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { c += v[j] != 0.f ? 1.0 : 0.0; s += v[j]; ss += (double)v[j] * v[j]; }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const float v = x[i];
      if (v != 0.f) { c += 1; s += v; ss += (double)v * v; }
    }
  }
  c = block_sum_d(c, red);
  s = block_sum_d(s, red);
  ss = block_sum_d(ss, red);
  if (threadIdx.x == 0) { atomicAdd(ws, c); atomicAdd(ws + 1, s); atomicAdd(ws + 2, ss); }
}

__global__ void evnorm_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, const double* ws) {
  const double cnt = ws[0];
  float mean = 0.f, sd = 1.f;
  const bool on = cnt > 0;
  if (on) {
    // fp32 arithmetic on the totals, as the reference does (inference_utils.py:104-105)
    const float fs = (float)ws[1], fss = (float)ws[2], fc = (float)cnt;
    mean = fs / fc;
    sd = sqrtf(fss / fc - mean * mean);
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = on ? (v != 0.f ? (v - mean) / sd : 0.f) : v;
  }
}

// All T time slices of a batch in one reduce + one map launch.  x = [B][T][chunk] (the trainers' event tensor [B, T*C, H, W]:
// slice t of sample b is a contiguous run of chunk = C*H*W floats), y = [T][B][chunk] (each normalised slice a contiguous
// [B, C, H, W] tensor); statistics per slice over the whole batch, as EventPreprocessor computes them on data_b[:, t*C:(t+1)*C].
// workspace: double[T][3].  blockIdx.y = slice.
__global__ __launch_bounds__(256) void evnorm_slices_reduce_kernel(const float* __restrict__ x, int B, int T, int64_t chunk, double* ws) {
  __shared__ double red[16];
  const int t = blockIdx.y;
  double c = 0, s = 0, ss = 0;
  if ((chunk & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
    const int64_t c4 = chunk >> 2, n4 = (int64_t)B * c4;
    const f32x4* x4 = (const f32x4*)x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t b = i / c4, off = i - b * c4;
      const f32x4 v = x4[(b * T + t) * c4 + off];
#pragma unroll
      for (int j = 0; j < 4; ++j) { c += v[j] != 0.f ? 1.0 : 0.0; s += v[j]; ss += (double)v[j] * v[j]; }
    }
  } else {
    const int64_t n = (int64_t)B * chunk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t b = i / chunk, off = i - b * chunk;
      const float v = x[(b * T + t) * chunk + off];
      if (v != 0.f) { c += 1; s += v; ss += (double)v * v; }
    }
  }
  c = block_sum_d(c, red);
  s = block_sum_d(s, red);
  ss = block_sum_d(ss, red);
  if (threadIdx.x == 0) { atomicAdd(ws + 3 * t, c); atomicAdd(ws + 3 * t + 1, s); atomicAdd(ws + 3 * t + 2, ss); }
}

__global__ void evnorm_slices_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int64_t chunk, const double* ws) {
  const int t = blockIdx.y;
  const double cnt = ws[3 * t];
  float mean = 0.f, sd = 1.f;
  const bool on = cnt > 0;
  if (on) {  // fp32 arithmetic on the totals, as the reference does (inference_utils.py:104-105)
    const float fs = (float)ws[3 * t + 1], fss = (float)ws[3 * t + 2], fc = (float)cnt;
    mean = fs / fc;
    sd = sqrtf(fss / fc - mean * mean);
  }
  if ((chunk & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
    const int64_t c4 = chunk >> 2, n4 = (int64_t)B * c4;
    const f32x4* x4 = (const f32x4*)x;
    f32x4* y4 = (f32x4*)y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t b = i / c4, off = i - b * c4;
      f32x4 v = x4[(b * T + t) * c4 + off];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = on ? (v[j] != 0.f ? (v[j] - mean) / sd : 0.f) : v[j];
      y4[(int64_t)t * n4 + i] = v;
    }
  } else {
    const int64_t n = (int64_t)B * chunk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t b = i / chunk, off = i - b * chunk;
      const float v = x[(b * T + t) * chunk + off];
      y[(int64_t)t * n + i] = on ? (v != 0.f ? (v - mean) / sd : 0.f) : v;
    }
  }
}

inline unsigned grid_for(int64_t n, int bs = 256, int cap = 256 * 16) {
  int64_t g = ceil_div64(n, bs);
  return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int ess_upsample_bilinear2x_add(const float* a, const float* b, float* y, int32_t planes, int32_t H, int32_t W,
                                           ess_stream_t stream) {
  ESS_CHECK_ARG(a && y && planes > 0 && H > 0 && W > 0, "upsample_bilinear2x_add: bad arguments");
  if ((W & 1) == 0 && (((uintptr_t)y) & 15) == 0 && (int64_t)planes * H * W < ((int64_t)1 << 31))
    hipLaunchKernelGGL(up2_bilinear_add_x4_kernel, dim3(grid_for((int64_t)planes * H * W)), dim3(256), 0, (hipStream_t)stream, a, b, y,
                       planes, H, W);
  else
    hipLaunchKernelGGL(up2_bilinear_add_kernel, dim3(grid_for((int64_t)planes * H * W * 4)), dim3(256), 0, (hipStream_t)stream, a,
                       b, y, planes, H, W);
  return ess_launch_status("upsample_bilinear2x_add");
}

extern "C" int ess_upsample_bilinear2x_add_c8(const float* a, const float* b, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                              ess_stream_t stream) {
  ESS_CHECK_ARG(a && y && N > 0 && C > 0 && H > 0 && W > 0, "upsample_bilinear2x_add_c8: bad arguments");
  ESS_CHECK_ARG((W & 1) == 0 && (((uintptr_t)y) & 15) == 0, "upsample_bilinear2x_add_c8: even source width and a 16-byte aligned output");
  const int64_t total = (int64_t)N * ((C + 7) / 8) * 2 * H * (W / 2);
  ESS_CHECK_ARG(total < ((int64_t)1 << 31), "upsample_bilinear2x_add_c8: tensor too large for 32-bit indexing");
  hipLaunchKernelGGL(up2_bilinear_add_c8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b, (uint4*)y, N, C, H, W);
  return ess_launch_status("upsample_bilinear2x_add_c8");
}

extern "C" int ess_upsample_bilinear2x_add_c8_from_c8(const void* a, const void* b, void* y, int32_t N, int32_t C, int32_t H,
                                                      int32_t W, ess_stream_t stream) {
  ESS_CHECK_ARG(a && y && N > 0 && C > 0 && H > 0 && W > 0, "upsample_bilinear2x_add_c8_from_c8: bad arguments");
  ESS_CHECK_ARG((W & 1) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y)) & 15) == 0,
                "upsample_bilinear2x_add_c8_from_c8: even source width and 16-byte aligned BF16_C8 tensors");
  const int64_t total = (int64_t)N * ((C + 7) / 8) * (H + 1) * (W / 2);  // (one thread per output row pair and 4 columns)
  ESS_CHECK_ARG((int64_t)N * ((C + 7) / 8) * 2 * H * (W / 2) < ((int64_t)1 << 31), "upsample_bilinear2x_add_c8_from_c8: tensor too large for 32-bit indexing");
  hipLaunchKernelGGL(up2_bilinear_add_c8c8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a,
                     (const uint4*)b, (uint4*)y, N, (C + 7) / 8, H, W);
  return ess_launch_status("upsample_bilinear2x_add_c8_from_c8");
}

extern "C" int ess_sumpool2x2(const float* x, float* y, int32_t planes, int32_t H_out, int32_t W_out, int32_t accumulate,
                              ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && planes > 0 && H_out > 0 && W_out > 0, "sumpool2x2: bad arguments");
  if ((W_out & 1) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && (int64_t)planes * H_out * W_out < ((int64_t)1 << 31))
    hipLaunchKernelGGL(sumpool2_x2_kernel, dim3(grid_for((int64_t)planes * H_out * W_out / 2)), dim3(256), 0, (hipStream_t)stream, x, y,
                       (unsigned)(planes * H_out), W_out, accumulate);
  else
    hipLaunchKernelGGL(sumpool2_kernel, dim3(grid_for((int64_t)planes * H_out * W_out)), dim3(256), 0, (hipStream_t)stream, x, y,
                       planes, H_out, W_out, accumulate);
  return ess_launch_status("sumpool2x2");
}

extern "C" int ess_add(const float* a, const float* b, float* y, int64_t n, ess_stream_t stream) {
  ESS_CHECK_ARG(a && b && y && n > 0, "add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
  return ess_launch_status("add");
}

// y = a + b (+ c) over bfloat16 tensors of any layout, 8 elements (one 16-byte vector) per thread and step: fp32 sum, ONE round to
// nearest even.  The gradient of an activation with several consumers in the bf16 configuration (BF16_C8 gradients).
__global__ void add_bf16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ c, uint4* __restrict__ y,
                                int64_t nv) {
  typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8v va = __builtin_bit_cast(bf16x8v, a[i]), vb = __builtin_bit_cast(bf16x8v, b[i]);
    bf16x8v vc;
    if (c) vc = __builtin_bit_cast(bf16x8v, c[i]);
    bf16x8v r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)((float)va[j] + (float)vb[j] + (c ? (float)vc[j] : 0.f));
    y[i] = __builtin_bit_cast(uint4, r);
  }
}

extern "C" int ess_add_bf16(const void* a, const void* b, const void* c, void* y, int64_t n_vectors, ess_stream_t stream) {
  ESS_CHECK_ARG(a && b && y && n_vectors > 0, "add_bf16: bad arguments");
  ESS_CHECK_ARG(((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)y)) & 15) == 0, "add_bf16: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n_vectors)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b,
                     (const uint4*)c, (uint4*)y, n_vectors);
  return ess_launch_status("add_bf16");
}

extern "C" int ess_event_normalize(const float* x, float* y, int64_t n, void* workspace, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && workspace && n > 0, "event_normalize: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  {
    hipError_t e = hipMemsetAsync(workspace, 0, 32, st);
    if (e != hipSuccess) {
      ess_set_error("event_normalize: memset failed: %s", hipGetErrorString(e));
      return ESS_ELAUNCH;
    }
  }
  // (few workgroups: each ends in three fp64 atomics on the same totals -- 1024 of them serialised for longer than the read took)
  hipLaunchKernelGGL(evnorm_reduce_kernel, dim3(grid_for(n / 4, 256, 128)), dim3(256), 0, st, x, n, (double*)workspace);
  hipLaunchKernelGGL(evnorm_apply_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n, (const double*)workspace);
  return ess_launch_status("event_normalize");
}


// sum of up to 16 device scalars (the weighted loss terms of a train step -> the step's reported total) in one launch
namespace {
struct ScalarPtrs { const float* p[16]; int n; };
__global__ void sum_scalars_kernel(const ScalarPtrs q, float* out) {
  float s = 0.f;
  for (int i = 0; i < q.n; ++i) s += *q.p[i];  // (fixed order: the order the trainer lists the terms in)
  *out = s;
}
}  // namespace

extern "C" int ess_sum_scalars(const float* const* terms, int32_t n, float* out, ess_stream_t stream) {
  ESS_CHECK_ARG(terms && out && n > 0 && n <= 16, "sum_scalars: 1..16 terms");
  ScalarPtrs q{};
  q.n = n;
  for (int i = 0; i < n; ++i) {
    ESS_CHECK_ARG(terms[i] != nullptr, "sum_scalars: null term");
    q.p[i] = terms[i];
  }
  hipLaunchKernelGGL(sum_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, q, out);
  return ess_launch_status("sum_scalars");
}

extern "C" int ess_event_normalize_slices(const float* x, float* y, int32_t B, int32_t T, int64_t chunk, void* workspace,
                                          ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && workspace && B > 0 && T > 0 && T <= 65535 && chunk > 0, "event_normalize_slices: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  {
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)T * 3 * sizeof(double), st);
    if (e != hipSuccess) {
      ess_set_error("event_normalize_slices: memset failed: %s", hipGetErrorString(e));
      return ESS_ELAUNCH;
    }
  }
  const int64_t n = (int64_t)B * chunk;
  hipLaunchKernelGGL(evnorm_slices_reduce_kernel, dim3(grid_for(n / 4, 256, 128), (unsigned)T), dim3(256), 0, st, x, B, T, chunk, (double*)workspace);
  hipLaunchKernelGGL(evnorm_slices_apply_kernel, dim3(grid_for(n / 4, 256, 1024), (unsigned)T), dim3(256), 0, st, x, y, B, T, chunk, (const double*)workspace);
  return ess_launch_status("event_normalize_slices");
}

// fp32 NCHW -> BF16_C8 ([N][ceil(C/8)][H][W][8] bfloat16, tail channels zero): one thread = one pixel vector.
namespace {
__global__ __launch_bounds__(256) void to_bf16_c8_kernel(const float* __restrict__ x, uint4* __restrict__ y, int C, int64_t hw,
                                                         int64_t total) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const int nblk = (C + 7) >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % hw, nb = i / hw;
    const int blk = (int)(nb % nblk);
    const int64_t n = nb / nblk;
    const float* p = x + ((size_t)n * C + (size_t)blk * 8) * hw + pix;
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)(blk * 8 + j < C ? p[(size_t)j * hw] : 0.f);
    y[i] = __builtin_bit_cast(uint4, b);
  }
}

// F.interpolate(mode='nearest') to an arbitrary size: src = min(floor(dst * (float)in / out), in - 1), the index
// arithmetic of torch's legacy nearest mode in fp32 (validation resizes logits to img_size_b before argmax / loss).
__global__ void resize_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int H, int W, int64_t total) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % W);
    const int64_t r = i / W;
    const int oy = (int)(r % H);
    const int64_t pl = r / H;
    const int iy = min((int)floorf(oy * sy), h - 1), ix = min((int)floorf(ox * sx), w - 1);
    y[i] = x[(pl * h + iy) * w + ix];
  }
}
}  // namespace

extern "C" int ess_resize_nearest(const float* x, float* y, int32_t planes, int32_t H_in, int32_t W_in, int32_t H_out, int32_t W_out,
                                  ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && planes > 0 && H_in > 0 && W_in > 0 && H_out > 0 && W_out > 0, "resize_nearest: bad arguments");
  const int64_t total = (int64_t)planes * H_out * W_out;
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(resize_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, H_in, W_in, H_out, W_out, total);
  return ess_launch_status("resize_nearest");
}

// split operands (ESS_COMPUTE_BF16X3): fp32 NCHW -> TWO BF16_C8 tensors, hi = bf16(v) and lo = bf16(v - hi); v = hi + lo to ~2^-17
namespace {
__global__ __launch_bounds__(256) void split_bf16_c8_kernel(const float* __restrict__ x, uint4* __restrict__ yh, uint4* __restrict__ yl,
                                                            int C, int64_t hw, int64_t total) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const int nblk = (C + 7) >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % hw, nb = i / hw;
    const int blk = (int)(nb % nblk);
    const int64_t n = nb / nblk;
    const float* p = x + ((size_t)n * C + (size_t)blk * 8) * hw + pix;
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = blk * 8 + j < C ? p[(size_t)j * hw] : 0.f;
      h[j] = (__bf16)v;
      l[j] = (__bf16)(v - (float)h[j]);
    }
    yh[i] = __builtin_bit_cast(uint4, h);
    yl[i] = __builtin_bit_cast(uint4, l);
  }
}
}  // namespace

// (library-internal: used by ess_conv2d_wgrad's split-operand path; not part of the C ABI)
int ess_split_bf16_c8_internal(const float* x, void* hi, void* lo, int N, int C, int H, int W, hipStream_t st) {
  const int64_t hw = (int64_t)H * W, total = (int64_t)N * ((C + 7) / 8) * hw;
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(split_bf16_c8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (uint4*)hi, (uint4*)lo, C, hw, total);
  return ess_launch_status("split_bf16_c8");
}

extern "C" int ess_to_bf16_c8(const float* x, void* y, int N, int C, int H, int W, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "to_bf16_c8: bad arguments");
  const int64_t hw = (int64_t)H * W, total = (int64_t)N * ((C + 7) / 8) * hw;
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(to_bf16_c8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (uint4*)y, C, hw, total);
  return ess_launch_status("to_bf16_c8");
}

// BF16_C8 -> fp32 NCHW (exact: every bf16 is an fp32).  The bridge out of the bf16 configuration's stored form, e.g. for a
// caller that wants the decoder's intermediate predictions as the reference's fp32 NCHW tensors.
namespace {
__global__ __launch_bounds__(256) void from_bf16_c8_kernel(const uint4* __restrict__ x, float* __restrict__ y, int C, int64_t hw,
                                                           int64_t total) {
  const int nblk = (C + 7) >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % hw, nb = i / hw;
    const int blk = (int)(nb % nblk);
    const int64_t n = nb / nblk;
    const uint4 v = x[i];
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
    float* p = y + ((size_t)n * C + (size_t)blk * 8) * hw + pix;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (blk * 8 + 2 * q < C) p[(size_t)(2 * q) * hw] = __builtin_bit_cast(float, u[q] << 16);
      if (blk * 8 + 2 * q + 1 < C) p[(size_t)(2 * q + 1) * hw] = __builtin_bit_cast(float, u[q] & 0xffff0000u);
    }
  }
}
}  // namespace

extern "C" int ess_from_bf16_c8(const void* x, float* y, int N, int C, int H, int W, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "from_bf16_c8: bad arguments");
  const int64_t hw = (int64_t)H * W, total = (int64_t)N * ((C + 7) / 8) * hw;
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(from_bf16_c8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, y, C, hw, total);
  return ess_launch_status("from_bf16_c8");
}

// ---- the "mixed" configuration's format bridges (round 6; ESS_COMPUTE_F16 consumers read F16_C8 tensors) ----
namespace {
__device__ __forceinline__ _Float16 half_sat(float v) {  // (saturating at +-65504, NaN kept: conv_common.h ess_f16_sat)
  const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  return (_Float16)(v != v ? v : c);
}
// fp32 NCHW -> F16_C8, or (hilo) the [hi | lo] pair [N][2 nblk][hw][8]: hi = half(v), lo = half(v - hi)
__global__ __launch_bounds__(256) void to_f16_c8_kernel(const float* __restrict__ x, uint4* __restrict__ y, int C, int64_t hw,
                                                        int64_t total, int hilo) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  const int nblk = (C + 7) >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % hw, nb = i / hw;
    const int blk = (int)(nb % nblk);
    const int64_t n = nb / nblk;
    const float* p = x + ((size_t)n * C + (size_t)blk * 8) * hw + pix;
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = blk * 8 + j < C ? p[(size_t)j * hw] : 0.f;
      h[j] = half_sat(v);
      l[j] = (_Float16)(v - (float)h[j]);
    }
    if (hilo) {
      y[((size_t)n * 2 * nblk + blk) * hw + pix] = __builtin_bit_cast(uint4, h);
      y[((size_t)n * 2 * nblk + nblk + blk) * hw + pix] = __builtin_bit_cast(uint4, l);
    } else {
      y[i] = __builtin_bit_cast(uint4, h);
    }
  }
}
// BF16_C8 -> F16_C8 (bfloat16's 8 significant bits fit a half; |v| > 65504 saturates, |v| < 6e-8 flushes)
__global__ __launch_bounds__(256) void bf16_to_f16_c8_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int64_t total) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 b = __builtin_bit_cast(bf16x8, x[i]);
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = half_sat((float)b[j]);
    y[i] = __builtin_bit_cast(uint4, h);
  }
}
// F16_C8 (src_blocks = nblk) or a [hi | lo] pair (src_blocks = 2 nblk: hi + lo) -> BF16_C8 (round to nearest even)
__global__ __launch_bounds__(256) void f16_to_bf16_c8_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int nblk, int src_blocks,
                                                             int64_t hw, int64_t total) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % hw, nb = i / hw;
    const int blk = (int)(nb % nblk);
    const int64_t n = nb / nblk;
    const f16x8 h = __builtin_bit_cast(f16x8, x[((size_t)n * src_blocks + blk) * hw + pix]);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    if (src_blocks == 2 * nblk) {
      const f16x8 l = __builtin_bit_cast(f16x8, x[((size_t)n * src_blocks + nblk + blk) * hw + pix]);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += (float)l[j];
    }
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
    y[i] = __builtin_bit_cast(uint4, b);
  }
}
inline unsigned bridge_grid(int64_t total) {
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  return (unsigned)blocks;
}
}  // namespace

extern "C" int ess_to_f16_c8(const float* x, void* y, int N, int C, int H, int W, int32_t hilo, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "to_f16_c8: bad arguments");
  const int64_t hw = (int64_t)H * W, total = (int64_t)N * ((C + 7) / 8) * hw;
  hipLaunchKernelGGL(to_f16_c8_kernel, dim3(bridge_grid(total)), dim3(256), 0, (hipStream_t)stream, x, (uint4*)y, C, hw, total, hilo ? 1 : 0);
  return ess_launch_status("to_f16_c8");
}

extern "C" int ess_bf16_c8_to_f16_c8(const void* x, void* y, int64_t n_vec, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && n_vec > 0, "bf16_c8_to_f16_c8: bad arguments");
  hipLaunchKernelGGL(bf16_to_f16_c8_kernel, dim3(bridge_grid(n_vec)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, n_vec);
  return ess_launch_status("bf16_c8_to_f16_c8");
}

extern "C" int ess_f16_c8_to_bf16_c8(const void* x, void* y, int N, int C, int H, int W, int32_t hilo, ess_stream_t stream) {
  ESS_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "f16_c8_to_bf16_c8: bad arguments");
  const int nblk = (C + 7) / 8;
  const int64_t hw = (int64_t)H * W, total = (int64_t)N * nblk * hw;
  hipLaunchKernelGGL(f16_to_bf16_c8_kernel, dim3(bridge_grid(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, nblk,
                     hilo ? 2 * nblk : nblk, hw, total);
  return ess_launch_status("f16_c8_to_bf16_c8");
}

// Weight gradient of the direct convolution on the fp32 matrix cores.
//
// GEMM view:  dW[tap][co][ci] = sum_px dY[co][px] * X[ci][px (+) tap],  reduction K = pixels.
//   A = dY (M = co), B = X (N = ci): lanes run over channels, so both LDS tiles use an ODD per-channel
//   pitch (bank-conflict free), and one A fragment is reused by all k*k taps.
// Workgroup = 4 waves = a 64(co) x 64(ci) tile, one 32x32 (co,ci) pair per wave, k*k accumulator
// blocks per wave.  The pixel range is split over blockIdx.y; partial slabs go to the workspace and a
// second kernel reduces them in a fixed order (deterministic, no float atomics).
// The 7x7 / C_in = 1 stem of the image encoder uses the "taps as N" variant instead.
#include "conv_wgrad_common.h"
#include <stdlib.h>

namespace {


// Value of the (virtually upsampled, concatenated) conv input at channel c (< C0+C1), position (gy, gx).  Branch-free in
// the per-lane quantities: the load is unconditional from a clamped address and the result selected afterwards (a load
// under a per-lane branch costs one serialized memory round trip per element, see conv_bf16.hip).
__device__ __forceinline__ float load_virtual(const WgradArgs& a, int n, int c, int gy, int gx) {
  const bool first = c < a.C0;
  const int mode = first ? a.mode0 : a.mode1;
  const int sh = mode != ESS_SRC_DIRECT ? 1 : 0;
  const int Hp = a.Hin >> sh, Wp = a.Win >> sh;
  const float* sp = first ? a.src0 : a.src1;
  const int cc = first ? c : c - a.C0, Cs = first ? a.C0 : a.C1;
  const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win && !(mode == ESS_SRC_ZERO_UP2 && ((gy | gx) & 1));
  const int cy = min(max(gy, 0), a.Hin - 1) >> sh, cx = min(max(gx, 0), a.Win - 1) >> sh;
  const float t = sp[(((size_t)n * Cs + cc) * Hp + cy) * Wp + cx];
  return ok ? t : 0.f;
}

// 8 consecutive virtual pixels gx..gx+7 (gx % 8 == 0) of one row, as floats; vector loads when the row is 16-byte
// friendly, clamped scalar loads otherwise.  `valid` masks the whole vector (channel / row out of range).
__device__ __forceinline__ void load8_virtual(const WgradArgs& a, int n, int c, int gy, int gx, bool valid, float (&f)[8]) {
  const bool first = c < a.C0;
  const int mode = first ? a.mode0 : a.mode1;
  const float* sp = first ? a.src0 : a.src1;
  const int cc = first ? c : c - a.C0, Cs = first ? a.C0 : a.C1;
  const bool rowok = valid && gy >= 0 && gy < a.Hin;
  const int cy = min(max(gy, 0), a.Hin - 1);
  if (mode == ESS_SRC_DIRECT && (a.Win & 7) == 0) {  // uniform per (tile, channel): a vector is fully in or fully out
    const bool ok = rowok && gx >= 0 && gx < a.Win;
    const float* src = sp + (((size_t)n * Cs + cc) * a.Hin + cy) * a.Win + min(max(gx, 0), a.Win - 8);
    const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = ok ? lo[j] : 0.f; f[4 + j] = ok ? hi[j] : 0.f; }
  } else if (mode == ESS_SRC_NEAREST_UP2 && (a.Win & 7) == 0) {  // 8 virtual pixels = 4 stored pixels, each used twice
    const int Wp = a.Win >> 1;
    const bool ok = rowok && gx >= 0 && gx < a.Win;
    const float* src = sp + (((size_t)n * Cs + cc) * (a.Hin >> 1) + (cy >> 1)) * Wp + (min(max(gx, 0), a.Win - 8) >> 1);
    const f32x4 v = *(const f32x4*)src;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = ok ? v[j] : 0.f; f[2 * j + 1] = f[2 * j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float t = load_virtual(a, n, c, gy, gx + j); f[j] = valid ? t : 0.f; }
  }
}

constexpr unsigned OOBF = 0x80000000u;  // beyond any buffer: bounds-checked loads return 0

template <int KS, int S>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = KS * KS;
  constexpr int PY = 65;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);  // channel-tile pair fastest, pixel split slowest
  const int pair = logical % a.npairs, split = logical / a.npairs, nsplit = a.nsplit;
  const int cot = pair / a.ci_tiles, cit = pair - cot * a.ci_tiles;
  const int cb = wave >> 1, ib = wave & 1;
  const int TWp = 1 << a.twl, THp = 64 >> a.twl;
  const int Cin = a.C0 + a.C1;
  float* dy_t = smem;            // [64][PY]
  float* x_t = smem + 64 * PY;   // [64][plx]

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const size_t HWo = (size_t)a.Hout * a.Wout;
  for (int tile = split; tile < a.ntiles; tile += nsplit) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int tr = tile - n * a.tiles_x * a.tiles_y;
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * THp, x0 = tx * TWp;
    __syncthreads();
    // ---- staging via bounds-checked buffer loads (out-of-range offset = 0): branch-free, so the unrolled rows'
    // loads are all in flight together
    {
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.dy + (size_t)n * a.Cout * HWo), 0, (unsigned)(a.Cout * HWo * 4), 0x00020000);
      const int qx = tid & (TWp - 1);
      const int rows_per_it = 256 >> a.twl;
      const int x = x0 + qx;
#pragma unroll 4
      for (int r = tid >> a.twl; r < 64 * THp; r += rows_per_it) {
        const int co = r / THp, qy = r - co * THp;  // THp is a power of two
        const int cg = cot * 64 + co, y = y0 + qy;
        const bool ok = cg < a.Cout && y < a.Hout && x < a.Wout;
        const unsigned off = ok ? (unsigned)((cg * a.Hout + y) * a.Wout + x) * 4u : OOBF;
        dy_t[co * PY + qy * TWp + qx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, (int)off, 0, 0));
      }
    }
    {
      const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
      const int Wp0 = a.Win >> sh0, Wp1 = a.Win >> sh1;
      const unsigned pl0 = (unsigned)((a.Hin >> sh0) * Wp0) * 4u, pl1 = (unsigned)((a.Hin >> sh1) * Wp1) * 4u;
      const __amdgpu_buffer_rsrc_t r0 =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.src0 + (size_t)n * a.C0 * (pl0 / 4)), 0, a.C0 * pl0, 0x00020000);
      const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.C1 ? a.src1 + (size_t)n * a.C1 * (pl1 / 4) : a.src0), 0, a.C1 * pl1, 0x00020000);
      const int iy0 = y0 * S - a.pad, ix0 = x0 * S - a.pad;
#pragma unroll 2
      for (int r = wave; r < 64 * a.IH; r += 4) {
        const int ci = r / a.IH, iy = r - ci * a.IH;
        const int cg = cit * 64 + ci;              // wave-uniform
        const bool first = cg < a.C0;
        const int mode = first ? a.mode0 : a.mode1, sh = first ? sh0 : sh1, Wp = first ? Wp0 : Wp1;
        const unsigned cbase = first ? (unsigned)cg * pl0 : (unsigned)(cg - a.C0) * pl1;
        const int gy = iy0 + iy;
        const bool rok = cg < Cin && gy >= 0 && gy < a.Hin && !(mode == ESS_SRC_ZERO_UP2 && (gy & 1));
        float* dst = x_t + ci * a.plx + iy * a.IW;
        for (int ix = lane; ix < a.IW; ix += 64) {
          const int gx = ix0 + ix;
          const bool ok = rok && gx >= 0 && gx < a.Win && !(mode == ESS_SRC_ZERO_UP2 && (gx & 1));
          const unsigned off = ok ? cbase + (unsigned)((gy >> sh) * Wp + (gx >> sh)) * 4u : OOBF;
          dst[ix] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? r0 : r1, (int)off, 0, 0));
        }
      }
    }
    __syncthreads();
    const float* ap = dy_t + (cb * 32 + p) * PY;
    const float* xp = x_t + (ib * 32 + p) * a.plx;
#pragma unroll 2
    for (int kk = 0; kk < 32; ++kk) {
      const int q = 2 * kk + half;
      const int qy = q >> a.twl, qx = q & (TWp - 1);
      const float av = ap[q];
      bsum += av;
      const float* xq = xp + qy * S * a.IW + qx * S;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
          acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xq[ky * a.IW + kx], acc[ky * KS + kx], 0, 0, 0);
    }
  }
  // ---- partial slab: ws[split][tap][co][ci]
  const int ci = cit * 64 + ib * 32 + p;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout && ci < Cin) a.ws[(((size_t)split * T + t) * a.Cout + co) * Cin + ci] = acc[t][r];
    }
  if (a.ws_b && cit == 0 && ib == 0) {
    bsum += __shfl_xor(bsum, 32, 64);  // both pixel parities
    const int co = cot * 64 + cb * 32 + p;
    if (half == 0 && co < a.Cout) a.ws_b[(size_t)split * a.Cout + co] = bsum;
  }
}

// "taps as N" variant for a single input channel (7x7 stem): dW[co][tap] = sum_px dY[co][px] X[px (+) tap]
template <int KS, int S>
__global__ __launch_bounds__(256) void wgrad_taps_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = KS * KS;
  constexpr int PY = 65;
  static_assert(T <= 64, "taps must fit two MFMA column blocks");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int cot = logical % a.npairs, split = logical / a.npairs, nsplit = a.nsplit;
  const int cb = wave >> 1, tb = wave & 1;
  const int TWp = 1 << a.twl, THp = 64 >> a.twl;
  float* dy_t = smem;
  float* x_t = smem + 64 * PY;  // [IH][IW]
  const int tap = tb * 32 + p;
  const int tky = tap / KS, tkx = tap - tky * KS;
  const bool tap_ok = tap < T;
  const int toff = tap_ok ? tky * a.IW + tkx : 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const size_t HWo = (size_t)a.Hout * a.Wout;
  for (int tile = split; tile < a.ntiles; tile += nsplit) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int tr = tile - n * a.tiles_x * a.tiles_y;
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * THp, x0 = tx * TWp;
    __syncthreads();
    {
      const int qx = tid & (TWp - 1);
      const int rows_per_it = 256 >> a.twl;
      if (a.dy_c8) {  // BF16_C8 dY: 8 blocks x 64 pixel vectors of this channel tile, two per thread
        const int nbo = (a.Cout + 7) >> 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int v = tid + i * 256;
          const int vx = v & (TWp - 1), vy = (v >> a.twl) % THp, cbk = v / 64;
          const int blk = cot * 8 + cbk, y = y0 + vy, x = x0 + vx;
          const bool ok = blk < nbo && y < a.Hout && x < a.Wout;
          const u32x4w q = ((const u32x4w*)a.dy)[((size_t)n * nbo + min(blk, nbo - 1)) * HWo + (size_t)min(y, a.Hout - 1) * a.Wout + min(x, a.Wout - 1)];
          const unsigned m = ok ? 0xffffffffu : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dy_t[(cbk * 8 + 2 * e) * PY + vy * TWp + vx] = __builtin_bit_cast(float, (q[e] << 16) & m);
            dy_t[(cbk * 8 + 2 * e + 1) * PY + vy * TWp + vx] = __builtin_bit_cast(float, q[e] & 0xffff0000u & m);
          }
        }
      } else
      for (int r = tid >> a.twl; r < 64 * THp; r += rows_per_it) {
        const int co = r / THp, qy = r - co * THp;
        const int cg = cot * 64 + co, y = y0 + qy, x = x0 + qx;
        const bool ok = cg < a.Cout && y < a.Hout && x < a.Wout;
        const float t = a.dy[((size_t)n * a.Cout + min(cg, a.Cout - 1)) * HWo + (size_t)min(y, a.Hout - 1) * a.Wout + min(x, a.Wout - 1)];
        dy_t[co * PY + qy * TWp + qx] = ok ? t : 0.f;
      }
      const int iy0 = y0 * S - a.pad, ix0 = x0 * S - a.pad;
      for (int iy = wave; iy < a.IH; iy += 4)
        for (int ix = lane; ix < a.IW; ix += 64) x_t[iy * a.IW + ix] = load_virtual(a, n, 0, iy0 + iy, ix0 + ix);
    }
    __syncthreads();
    const float* ap = dy_t + (cb * 32 + p) * PY;
#pragma unroll 4
    for (int kk = 0; kk < 32; ++kk) {
      const int q = 2 * kk + half;
      const int qy = q >> a.twl, qx = q & (TWp - 1);
      const float bv = tap_ok ? x_t[qy * S * a.IW + qx * S + toff] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[q], bv, acc, 0, 0, 0);
    }
  }
  // slab layout [split][tap][co][ci=1]
  if (tap_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout) a.ws[((size_t)split * T + tap) * a.Cout + co] = acc[r];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// bf16 variant for 3x3 / stride 1 (95 % of the weight-gradient FLOPs): v_mfma_f32_32x32x16_bf16, reduction over 16
// pixels per instruction.  A fragment = 8 consecutive pixels of dY[co] (lane half picks the second 8), B fragment =
// the same 8 pixels of X[ci] shifted by the tap.  Tiles are converted to bf16 while staged; rows of X are stored
// ONCE, aligned so that the kx = 1 fragment is a plain 16-byte vector; the kx = 0 / 2 fragments (one pixel to the
// left / right) are produced from two neighbouring vectors with v_alignbyte (4 VALU per fragment) instead of keeping
// three shifted copies in LDS.  Per-channel pitches are odd multiples of 16 B: lanes run over channels, so the
// ds_read_b128 groups hit 16 distinct slots.
// 16-byte bounds-checked buffer loads.  hipcc (ROCm 7.2) mis-lowers __builtin_amdgcn_raw_buffer_load_b64/_b128 to a
// single buffer_load_dword, so the wide form is inline asm.  A whole batch is ONE statement -- loads and their
// s_waitcnt together, early-clobber outputs (cdna_hip_programming.md 5.7 form i) -- so the compiler never sees a
// destination register that is "written" but still in flight (it may otherwise copy it before the data lands).
// `s_nop 4` covers the SGPR-write -> VMEM-read hazard on the freshly built descriptors.
__device__ __forceinline__ void ldb128x8(u32x4w (&d)[8], __amdgpu_buffer_rsrc_t r, const unsigned (&o)[8]) {
  asm volatile(
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %0, %8, %16, 0 offen\n\tbuffer_load_dwordx4 %1, %9, %16, 0 offen\n\t"
      "buffer_load_dwordx4 %2, %10, %16, 0 offen\n\tbuffer_load_dwordx4 %3, %11, %16, 0 offen\n\t"
      "buffer_load_dwordx4 %4, %12, %16, 0 offen\n\tbuffer_load_dwordx4 %5, %13, %16, 0 offen\n\t"
      "buffer_load_dwordx4 %6, %14, %16, 0 offen\n\tbuffer_load_dwordx4 %7, %15, %16, 0 offen\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
      : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]), "s"(r)
      : "memory");
}
// 6 loads against descriptor ra followed by 6 against rb
__device__ __forceinline__ void ldb128x12(u32x4w (&d)[12], __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                          const unsigned (&o)[12]) {
  asm volatile(
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %0, %12, %24, 0 offen\n\tbuffer_load_dwordx4 %1, %13, %24, 0 offen\n\t"
      "buffer_load_dwordx4 %2, %14, %24, 0 offen\n\tbuffer_load_dwordx4 %3, %15, %24, 0 offen\n\t"
      "buffer_load_dwordx4 %4, %16, %24, 0 offen\n\tbuffer_load_dwordx4 %5, %17, %24, 0 offen\n\t"
      "buffer_load_dwordx4 %6, %18, %25, 0 offen\n\tbuffer_load_dwordx4 %7, %19, %25, 0 offen\n\t"
      "buffer_load_dwordx4 %8, %20, %25, 0 offen\n\tbuffer_load_dwordx4 %9, %21, %25, 0 offen\n\t"
      "buffer_load_dwordx4 %10, %22, %25, 0 offen\n\tbuffer_load_dwordx4 %11, %23, %25, 0 offen\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]),
        "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11])
      : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]), "v"(o[8]), "v"(o[9]),
        "v"(o[10]), "v"(o[11]), "s"(ra), "s"(rb)
      : "memory");
}

__global__ __launch_bounds__(256) void wgrad_bf16_k3s1_kernel(const WgradBArgs b) {
  extern __shared__ __attribute__((aligned(16))) u32x4w smemv[];
  const WgradArgs& a = b.w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pair = logical % a.npairs, split = logical / a.npairs, nsplit = a.nsplit;
  const int cot = pair / a.ci_tiles, cit = pair - cot * a.ci_tiles;
  const int cb = wave >> 1, ib = wave & 1;
  const int TWp = 1 << a.twl, THp = 128 >> a.twl;
  const int TV = TWp >> 3;  // vectors per dY row
  const int IH = THp + 2;
  const int Cin = a.C0 + a.C1;
  u32x4w* dy_t = smemv;                 // [64][pyv]
  u32x4w* x_t = smemv + 64 * b.pyv;     // [64][pxv]

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const size_t HWo = (size_t)a.Hout * a.Wout;

  // split operands (b.split): each tile three times -- pass 0 (dY_hi, X_hi), 1 (dY_hi, X_lo), 2 (dY_lo, X_hi)
  const int npass = b.split ? 3 : 1;
  for (int tile = split; tile < a.ntiles; tile += nsplit)
  for (int pass = 0; pass < npass; ++pass) {
    const bool d_lo = pass == 2, x_lo = pass == 1;
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int tr = tile - n * a.tiles_x * a.tiles_y;
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * THp, x0 = tx * TWp;
    __syncthreads();
    // ---- staging through bounds-checked buffer loads: out-of-range offsets (OOBW) read as zero, so there is no
    // per-lane branch or select and all loads of a batch are in flight together
    {
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.dy + (size_t)n * a.Cout * HWo), 0, (unsigned)(a.Cout * HWo * 4), 0x00020000);
      constexpr int DV = 4;  // 64 * 128 / 8 / 256 vectors per thread
      float f[DV][8];
#pragma unroll
      for (int i = 0; i < DV; ++i) {
        const int v = tid + i * 256;
        const int xv = v % TV, r = v / TV;
        const int qy = r % THp, co = r / THp;
        const int cg = cot * 64 + co, y = y0 + qy, x = x0 + xv * 8;
        const bool rok = cg < a.Cout && y < a.Hout;
        const unsigned base = rok ? (unsigned)((cg * a.Hout + y) * a.Wout + x) * 4u : OOBW;
        {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            f[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, (int)((x + j < a.Wout) ? base + 4 * j : OOBW), 0, 0));
        }
      }
#pragma unroll
      for (int i = 0; i < DV; ++i) {
        const int v = tid + i * 256;
        const int xv = v % TV, r = v / TV;
        const int qy = r % THp, co = r / THp;
        dy_t[co * b.pyv + qy * TV + xv] = d_lo ? cvt8_lo(f[i]) : cvt8(f[i]);
      }
    }
    {
      const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
      const unsigned pl0 = (unsigned)((a.Hin >> sh0) * (a.Win >> sh0)) * 4u, pl1 = (unsigned)((a.Hin >> sh1) * (a.Win >> sh1)) * 4u;
      const __amdgpu_buffer_rsrc_t r0 =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.src0 + (size_t)n * a.C0 * (pl0 / 4)), 0, a.C0 * pl0, 0x00020000);
      const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.C1 ? a.src1 + (size_t)n * a.C1 * (pl1 / 4) : a.src0), 0, a.C1 * pl1, 0x00020000);
      const bool fast = (a.Win & 7) == 0 && a.mode0 != ESS_SRC_ZERO_UP2 && a.mode1 != ESS_SRC_ZERO_UP2;
      const int nxv = 64 * IH * b.rv;
      for (int v0 = 0; v0 < nxv; v0 += 256 * 3) {  // batches of 3 vectors per thread keep the register footprint small
        float f[3][8];
        u32x4w xq[12];
        unsigned xofs[12];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int v = v0 + tid + i * 256;
          const int xv = v % b.rv, r = v / b.rv;
          const int iy = r % IH, ci = r / IH;
          const int cg = cit * 64 + ci, gy = y0 - a.pad + iy, gx = x0 - 8 + xv * 8;
          const bool first = cg < a.C0;
          const int sh = first ? sh0 : sh1;
          const int Wp = a.Win >> sh;
          const unsigned pls = first ? pl0 : pl1;
          const int cc = first ? cg : cg - a.C0;
          const bool rok = v < nxv && cg < Cin && gy >= 0 && gy < a.Hin;
          if (fast) {
            const bool ok = rok && gx >= 0 && gx < a.Win;
            const unsigned o = ok ? (unsigned)cc * pls + (unsigned)((gy >> sh) * Wp + (gx >> sh)) * 4u : OOBW;
            // per-lane descriptor choice is avoided by issuing against both sources: the wrong one is out of range
            const unsigned oa = first ? o : OOBW, ob = first ? OOBW : o;
            xofs[2 * i] = oa;
            xofs[2 * i + 1] = sh0 ? OOBW : oa + 16;
            xofs[6 + 2 * i] = ob;
            xofs[6 + 2 * i + 1] = sh1 ? OOBW : ob + 16;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int x = gx + j;
              const int mode = first ? a.mode0 : a.mode1;
              const bool ok = rok && x >= 0 && x < a.Win && !(mode == ESS_SRC_ZERO_UP2 && ((gy | x) & 1));
              const unsigned o = ok ? (unsigned)cc * pls + (unsigned)((gy >> sh) * Wp + (x >> sh)) * 4u : OOBW;
              const unsigned ua = __builtin_amdgcn_raw_buffer_load_b32(r0, (int)(first ? o : OOBW), 0, 0);
              const unsigned ub = __builtin_amdgcn_raw_buffer_load_b32(r1, (int)(first ? OOBW : o), 0, 0);
              f[i][j] = __builtin_bit_cast(float, ua | ub);
            }
          }
        }
        if (fast) {
          ldb128x12(xq, r0, r1, xofs);
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              // a nearest-upsampled source holds 4 stored pixels per vector, each used twice; the source that does not
              // own this channel was read out of range and contributes zero bits
              const unsigned ua = sh0 ? xq[2 * i][j >> 1] : xq[2 * i + (j >> 2)][j & 3];
              const unsigned ub = sh1 ? xq[6 + 2 * i][j >> 1] : xq[6 + 2 * i + (j >> 2)][j & 3];
              f[i][j] = __builtin_bit_cast(float, ua | ub);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int v = v0 + tid + i * 256;
          if (v < nxv) {
            const int xv = v % b.rv, r = v / b.rv;
            const int iy = r % IH, ci = r / IH;
            x_t[ci * b.pxv + iy * b.rv + xv] = x_lo ? cvt8_lo(f[i]) : cvt8(f[i]);
          }
        }
      }
    }
    __syncthreads();
    const u32x4w* ap = dy_t + (cb * 32 + p) * b.pyv + half;
    const u32x4w* xp = x_t + (ib * 32 + p) * b.pxv + half;
    for (int qy = 0; qy < THp; ++qy) {
      for (int xs = 0; xs < TV; xs += 2) {  // one 16-pixel k-step
        const u32x4w av = ap[qy * TV + xs];
        const bf16x8w af = __builtin_bit_cast(bf16x8w, av);
        if (!x_lo) {  // (bias gradient = sum of dY: its hi part in pass 0, its lo part in pass 2)
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum += (float)af[j];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const u32x4w* row = xp + (qy + ky) * b.rv + xs;
          const u32x4w v0 = row[0], v1 = row[1], v2 = row[2];
          acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, shift_left1(v0, v1)),
                                                                    acc[ky * 3 + 0], 0, 0, 0);
          acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, v1), acc[ky * 3 + 1], 0, 0, 0);
          acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, shift_right1(v1, v2)),
                                                                    acc[ky * 3 + 2], 0, 0, 0);
        }
      }
    }
  }
  const int ci = cit * 64 + ib * 32 + p;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout && ci < Cin) a.ws[(((size_t)split * 9 + t) * a.Cout + co) * Cin + ci] = acc[t][r];
    }
  if (a.ws_b && cit == 0 && ib == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    const int co = cot * 64 + cb * 32 + p;
    if (half == 0 && co < a.Cout) a.ws_b[(size_t)split * a.Cout + co] = bsum;
  }
}


// ---- fast variant: rows that are multiples of 8 pixels, direct / nearest-upsampled sources.
// All global loads of the NEXT pixel tile (14 vectors = 28 x 16 B per thread) are issued at the top of a tile iteration
// and stay in flight across the first half of its MFMA phase; their conversion to bf16 and the LDS writes (second LDS
// stage) ride on k-steps 4-7 of that phase.  Loads are plain, unconditional dwordx4 loads from clamped (always valid)
// per-lane addresses; padding / overhang vectors are zeroed by an AND mask after conversion -- no branch, no select on
// the load, nothing for the compiler to serialize.
// Ablation (256->256 @ 60x80, B=8, 142 us with the split-K reduce): fixed part (partial sums out + reduce kernel) 50 us,
// loads + conversion alone 62 us (573 MB through L2->L1 = 9.2 TB/s: the bound), MFMA phase alone 45 us.
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_bf16_k3s1_fast_kernel(const WgradBArgs b) {
  extern __shared__ __attribute__((aligned(16))) u32x4w smemv[];
  const WgradArgs& a = b.w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pair = logical % a.npairs, split = logical / a.npairs, nsplit = a.nsplit;
  const int cot = pair / a.ci_tiles, cit = pair - cot * a.ci_tiles;
  const int cb = wave >> 1, ib = wave & 1;
  const int TWp = 1 << a.twl, THp = 128 >> a.twl;
  const int TV = TWp >> 3;
  const int IH = THp + 2;
  const int Cin = a.C0 + a.C1;
  u32x4w* dy_t = smemv;
  u32x4w* x_t = smemv + 64 * b.pyv;
  const int stage = 64 * (b.pyv + b.pxv);  // 16-byte vectors per LDS stage
  constexpr int DV = 4, XV = 10;  // vectors per thread: 64*128/8/256 of dY, ceil(64*IH*RV/256) of X (IH*RV = 36 or 40)
  const int nxv = 64 * IH * b.rv;
  const size_t HWo = (size_t)a.Hout * a.Wout;
  const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;

  // tile-independent part of the staging plan
  int d_lds[DV], d_co[DV], d_qy[DV], d_xv[DV];
#pragma unroll
  for (int i = 0; i < DV; ++i) {
    const int v = tid + i * 256;
    const int xv = v % TV, r = v / TV;
    d_qy[i] = r % THp; d_co[i] = cot * 64 + r / THp; d_xv[i] = xv;
    d_lds[i] = (r / THp) * b.pyv + (r % THp) * TV + xv;
  }
  int x_lds[XV], x_iy[XV], x_xv[XV], x_sh[XV];
  const float* x_base[XV];  // channel plane of this vector's channel in its source (sample 0)
  size_t x_ns[XV];          // sample stride of that source
  int x_w[XV];
  unsigned x_ok[XV];
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int v = tid + i * 256;
    const int xv = v % b.rv, r = v / b.rv;
    const int iy = r % IH, ci = r / IH;
    const int cg = cit * 64 + ci;
    const bool cok = v < nxv && cg < Cin;
    const int cgc = min(cg, Cin - 1);
    const bool first = cgc < a.C0;
    const int sh = first ? sh0 : sh1;
    const int Hp = a.Hin >> sh, Wp = a.Win >> sh;
    const int cc = first ? cgc : cgc - a.C0, Cs = first ? a.C0 : a.C1;
    x_base[i] = (first ? a.src0 : a.src1) + (size_t)cc * Hp * Wp;
    x_ns[i] = (size_t)Cs * Hp * Wp;
    x_w[i] = Wp; x_sh[i] = sh; x_iy[i] = iy; x_xv[i] = xv;
    x_ok[i] = cok ? 0xffffffffu : 0u;
    x_lds[i] = v < nxv ? ci * b.pxv + iy * b.rv + xv : -1;
  }

  f32x4 dreg[DV][2], xreg[XV][2];
  unsigned dmask[DV], xmask[XV];  // 0 / all-ones, ANDed onto the converted vector (cheaper than 8 multiplies, NaN-safe)
  auto issue = [&](int tile) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int tr = tile - n * a.tiles_x * a.tiles_y;
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * THp, x0 = tx * TWp;
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int y = y0 + d_qy[i], x = x0 + d_xv[i] * 8;
      dmask[i] = (d_co[i] < a.Cout && y < a.Hout && x < a.Wout) ? 0xffffffffu : 0u;
      const float* src = a.dy + ((size_t)n * a.Cout + min(d_co[i], a.Cout - 1)) * HWo + (size_t)min(y, a.Hout - 1) * a.Wout +
                         min(x, a.Wout - 8);
      dreg[i][0] = *(const f32x4*)src;
      dreg[i][1] = *(const f32x4*)(src + 4);
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int gy = y0 - a.pad + x_iy[i], gx = x0 - 8 + x_xv[i] * 8;
      xmask[i] = (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) ? x_ok[i] : 0u;
      const int cy = min(max(gy, 0), a.Hin - 1) >> x_sh[i];
      const int cx = min(max(gx, 0), a.Win - 8) >> x_sh[i];
      const float* src = x_base[i] + (size_t)n * x_ns[i] + (size_t)cy * x_w[i] + cx;
      xreg[i][0] = *(const f32x4*)src;
      // a nearest-upsampled source covers the 8 virtual pixels with 4 stored ones: the second load is not needed,
      // re-read the first address (keeps the access in range at the right image border)
      xreg[i][1] = *(const f32x4*)(src + (x_sh[i] ? 0 : 4));
    }
  };
  // convert + store ONE staged vector of the next tile into LDS stage `st` (i is a compile-time index after unrolling)
  // split operands (b.split): iteration `it` of the K loop is (tile it / npass, pass it % npass); the pass of the STAGED tile picks
  // the part: pass 0 (dY_hi, X_hi), 1 (dY_hi, X_lo), 2 (dY_lo, X_hi).  Tile-major, so the three passes re-read one tile from L2.
  const int npass = b.split ? 3 : 1;
  int cpass = 0;  // pass of the tile being committed
  auto commit_d = [&](int i, int st) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = dreg[i][0][j]; f[4 + j] = dreg[i][1][j]; }
    u32x4w v = cpass == 2 ? cvt8_lo(f) : cvt8(f);
    v[0] &= dmask[i]; v[1] &= dmask[i]; v[2] &= dmask[i]; v[3] &= dmask[i];
    dy_t[st * stage + d_lds[i]] = v;
  };
  auto commit_x = [&](int i, int st) {
    float f[8];
    if (x_sh[i]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[2 * j] = xreg[i][0][j]; f[2 * j + 1] = f[2 * j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[j] = xreg[i][0][j]; f[4 + j] = xreg[i][1][j]; }
    }
    u32x4w v = cpass == 1 ? cvt8_lo(f) : cvt8(f);
    v[0] &= xmask[i]; v[1] &= xmask[i]; v[2] &= xmask[i]; v[3] &= xmask[i];
    if (x_lds[i] >= 0) x_t[st * stage + x_lds[i]] = v;
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // K loop over this workgroup's pixel tiles.  One tile = 8 k-steps of 16 pixels x 9 taps.  Two LDS stages: while the
  // MFMAs of tile t read stage t&1, the vectors of tile t+1 (loads issued at the top of the iteration) are converted and
  // written into the other stage, two per k-step, in the shadow of the matrix pipe; the fragments of k-step k+1 are
  // read from LDS before the MFMAs of k-step k issue (register double buffer).  One barrier per tile.
  const int ksh = a.twl - 4, kmask = (1 << ksh) - 1;  // k-step -> (tile row, 16-pixel column) of the pixel tile
  struct Frag { u32x4w a; u32x4w v[3][3]; };
  if (split < a.ntiles) {
    issue(split);
#pragma unroll
    for (int i = 0; i < DV; ++i) commit_d(i, 0);
#pragma unroll
    for (int i = 0; i < XV; ++i) commit_x(i, 0);
  }
  __syncthreads();
  int st = 0;
  const int my_tiles = split < a.ntiles ? (a.ntiles - split + nsplit - 1) / nsplit : 0;
  for (int it = 0; it < my_tiles * npass; ++it, st ^= 1) {
    const int pass = it % npass;
    const bool more = it + 1 < my_tiles * npass;
    const int next_tile = split + ((it + 1) / npass) * nsplit;
    cpass = (it + 1) % npass;
    const u32x4w* ap = dy_t + st * stage + (cb * 32 + p) * b.pyv + half;
    const u32x4w* xp = x_t + st * stage + (ib * 32 + p) * b.pxv + half;
    auto read_frag = [&](int ks, Frag& f) {
      f.a = ap[2 * ks];
      const u32x4w* row = xp + (ks >> ksh) * b.rv + ((ks & kmask) << 1);
      if constexpr (TAPS == 1) {
        f.v[1][1] = row[b.rv + 1];
      } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          f.v[ky][0] = row[ky * b.rv]; f.v[ky][1] = row[ky * b.rv + 1]; f.v[ky][2] = row[ky * b.rv + 2];
        }
      }
    };
    auto mma = [&](const Frag& f) {
      const bf16x8w af = __builtin_bit_cast(bf16x8w, f.a);
      if (pass != 1) {  // (bias gradient = sum of dY: its hi part in pass 0, its lo part in pass 2)
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)af[j];
      }
      if constexpr (TAPS == 1) {  // 1x1 convolution = the centre tap of the same pixel-tile geometry
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, f.v[1][1]), acc[0], 0, 0, 0);
      } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, shift_left1(f.v[ky][0], f.v[ky][1])),
                                                                    acc[ky * 3 + 0], 0, 0, 0);
          acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, f.v[ky][1]), acc[ky * 3 + 1], 0, 0, 0);
          acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, shift_right1(f.v[ky][1], f.v[ky][2])),
                                                                    acc[ky * 3 + 2], 0, 0, 0);
        }
      }
    };
    Frag fr[2];
    read_frag(0, fr[0]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) read_frag(ks + 1, fr[(ks + 1) & 1]);
      mma(fr[ks & 1]);
      if (ks == 0 && more) issue(next_tile);  // address arithmetic + 28 loads, behind the first k-step's MFMAs
      if (more) {  // wave-uniform; the slice of the next tile's staging that rides on this k-step
        if (ks == 4) { commit_d(0, st ^ 1); commit_d(1, st ^ 1); commit_d(2, st ^ 1); commit_d(3, st ^ 1); }
        if (ks == 5) { commit_x(0, st ^ 1); commit_x(1, st ^ 1); commit_x(2, st ^ 1); }
        if (ks == 6) { commit_x(3, st ^ 1); commit_x(4, st ^ 1); commit_x(5, st ^ 1); }
        if (ks == 7) { commit_x(6, st ^ 1); commit_x(7, st ^ 1); commit_x(8, st ^ 1); commit_x(9, st ^ 1); }
      }
    }
    __syncthreads();
  }
  const int ci = cit * 64 + ib * 32 + p;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout && ci < Cin) a.ws[(((size_t)split * TAPS + t) * a.Cout + co) * Cin + ci] = acc[t][r];
    }
  if (a.ws_b && cit == 0 && ib == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    const int co = cot * 64 + cb * 32 + p;
    if (half == 0 && co < a.Cout) a.ws_b[(size_t)split * a.Cout + co] = bsum;
  }
}


// ---- 1x1 convolution with few channels (the K-class head, 32 -> K at full resolution): HBM-bound, the MFMA tile
// kernel would stage 64x64 channels to use 11x32.  Plain VALU: each workgroup walks chunks of 128 pixels, stages
// dY[Cout][128] and X[Cin][128] in LDS (coalesced rows) and each thread accumulates up to 4 (co, ci) products.
__global__ __launch_bounds__(256) void wgrad_small1x1_kernel(const WgradArgs a, int nchunks_per_img, int total_chunks) {
  constexpr int P = 128, PP = P + 1;
  __shared__ float dy_s[16 * PP];
  __shared__ float x_s[64 * PP];
  const int tid = threadIdx.x;
  const int Cin = a.C0, Cout = a.Cout;
  const int npair = Cout * Cin;
  const size_t HW = (size_t)a.Hout * a.Wout;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;
  int pco[4], pci[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int q = tid + k * 256; pco[k] = q / Cin; pci[k] = q - pco[k] * Cin; }
  for (int chunk = blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
    const int n = chunk / nchunks_per_img;
    const size_t px0 = (size_t)(chunk - n * nchunks_per_img) * P;
    __syncthreads();
    for (int i = tid; i < (Cout + Cin) * P; i += 256) {
      const int r = i / P, j = i - r * P;
      const bool isdy = r < Cout;
      const float* src = isdy ? a.dy + ((size_t)n * Cout + r) * HW : a.src0 + ((size_t)n * Cin + (r - Cout)) * HW;
      const size_t px = px0 + j;
      const float t = src[px < HW ? px : HW - 1];
      const float v = px < HW ? t : 0.f;
      if (isdy) dy_s[r * PP + j] = v; else x_s[(r - Cout) * PP + j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (tid + k * 256 < npair) {
        const float* dp = dy_s + pco[k] * PP;
        const float* xp = x_s + pci[k] * PP;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < P; ++j) s += dp[j] * xp[j];
        acc[k] += s;
      }
    }
    if (tid < Cout) {
      float s = 0.f;
      for (int j = 0; j < P; ++j) s += dy_s[tid * PP + j];
      bacc += s;
    }
  }
  // slab [split = blockIdx.x][tap = 0][co][ci]
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (tid + k * 256 < npair) a.ws[(size_t)blockIdx.x * npair + tid + k * 256] = acc[k];
  if (a.ws_b && tid < Cout) a.ws_b[(size_t)blockIdx.x * Cout + tid] = bacc;
}

// ---- the same 1x1 head on the fp32 matrix core, for C_in <= 32 (C_out <= 16 as above), H*W a multiple of 64.
// dW[co][ci] = sum_px dY[co][px] * X[ci][px] is ONE 32x32 output tile with K = all pixels: v_mfma_f32_32x32x2_f32 (exact
// fp32) contracts 2 pixels per instruction, lane (p, half) feeding channel p.  The pixel order inside a group is free (it
// is a sum), so lane (p, half) takes the 32 CONSECUTIVE pixels 32*half .. 32*half+31 of a 64-pixel group of its channel
// plane -- one whole 128-byte line per tensor per lane -- straight from global memory: no LDS, no conversion, a pure
// stream over dY and X.  Each wave walks a contiguous range of groups.  The VALU version above is LDS-bound (2
// ds_read_b32 per FMA: 393 us on the 32 -> 11 @480x640 B=8 head).
__global__ __launch_bounds__(256) void wgrad_small1x1_mfma_kernel(const WgradArgs a, int groups_per_img, int total_groups) {
  __shared__ float red[4][16 * 64];
  __shared__ float redb[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int Cin = a.C0, Cout = a.Cout;
  const size_t HW = (size_t)a.Hout * a.Wout;
  const float dmask = p < Cout ? 1.f : 0.f, xmask = p < Cin ? 1.f : 0.f;
  const int pc_d = p < Cout ? p : Cout - 1, pc_x = p < Cin ? p : Cin - 1;  // clamped channel: always a valid address
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int per = (total_groups + nw - 1) / nw;
  const int g_lo = wid * per, g_hi = g_lo + per < total_groups ? g_lo + per : total_groups;
  for (int g = g_lo; g < g_hi; ++g) {
    const int n = g / groups_per_img;
    const size_t px = (size_t)(g - n * groups_per_img) * 64 + 32 * half;
    const f32x4* dp = (const f32x4*)(a.dy + ((size_t)n * Cout + pc_d) * HW + px);
    const f32x4* xp = (const f32x4*)(a.src0 + ((size_t)n * Cin + pc_x) * HW + px);
    f32x4 dv[8], xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { dv[q] = dp[q]; xv[q] = xp[q]; }
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d1 = dv[q][j] * dmask, x1 = xv[q][j] * xmask;
        bsum += d1;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, x1, acc, 0, 0, 0);
      }
  }
  // workgroup reduction of the 4 waves, then one slab per workgroup: [split = blockIdx.x][tap 0][co][ci]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
  bsum += __shfl_xor(bsum, 32, 64);
  if (half == 0) redb[wave][p] = bsum;
  __syncthreads();
  for (int i = tid; i < 16 * 64; i += 256) {
    const int r = i >> 6, l = i & 63;
    const int co = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = l & 31;
    if (co < Cout && ci < Cin)
      a.ws[(size_t)blockIdx.x * Cout * Cin + co * Cin + ci] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
  if (a.ws_b && tid < Cout) a.ws_b[(size_t)blockIdx.x * Cout + tid] = (redb[0][tid] + redb[1][tid]) + (redb[2][tid] + redb[3][tid]);
}

// Deterministic reduction of the split-K slabs: dw[co][ci][tap] (+)= sum_split ws[split][tap][co][ci].
// One workgroup = E consecutive (co, ci) elements x NT taps x G = THREADS/E split groups: thread (g, e) sums splits g, g+G, ...
// of its element, U splits x NT taps = 36 / 8 independent coalesced loads in flight; the groups are combined in a fixed
// order through LDS.
//   <64, 9>: 3x3 layers with >= 32k weights.  The E x 9 results leave as ONE contiguous run of dw (the [co][ci][tap] order
//            makes the taps of an element adjacent); the first version wrote them with a 36-byte stride from nine
//            workgroups (read-modify-write of partial lines, 1.9 TB/s overall): 256x256x9, 16 splits: 20 -> 9 us.
//   <64, 1>: everything else (few weights, many splits): one tap per workgroup (blockIdx.y), 16 split groups of 64 lanes
//            (1024 threads) -- there the slab reads are all that matters and they want as many loads in flight as possible.
// Blocks past nblk_w (blockIdx.y == 0 only) reduce the bias slabs ws_b[split][co] the same way (one "tap").
// tm (mapped != 0; 3x3 only): slab tap t lands in tap tm.m[t] of dw, or nowhere (< 0) -- the parity phases of a stride-2 filter
struct TapMap { int mapped; signed char m[12]; };
template <int E, int NT, int THREADS>
__global__ __launch_bounds__(THREADS) void wgrad_reduce_kernel(const float* ws, const float* ws_b, float* dw, float* db, int nsplit, int T,
                                                           int CC, int Cout, int nblk_w, int accumulate, const TapMap tm) {
  constexpr int G = THREADS / E, PITCH = E * NT + 1, U = NT == 1 ? 8 : 4;
  __shared__ float part[G * PITCH];
  int blk = blockIdx.x, t0 = blockIdx.y * NT, nt = min(NT, T - t0);
  if (blk >= nblk_w) {  // bias blocks (uniform per block)
    if (blockIdx.y) return;
    blk -= nblk_w; ws = ws_b; dw = db; T = 1; CC = Cout; nt = 1;
  }
  const bool mapped = tm.mapped && T == 9;
  const int el = threadIdx.x % E, g = threadIdx.x / E;
  const int e = blk * E + el;
  float acc[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) acc[u] = 0.f;
  if (e < CC) {
    int k = g;
    for (; k + (U - 1) * G < nsplit; k += U * G) {
      float v[U][NT];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const float* src = ws + ((size_t)(k + j * G) * T + t0) * CC + e;
#pragma unroll
        for (int u = 0; u < NT; ++u) v[j][u] = u < nt ? src[(size_t)u * CC] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < U; j += 2) t += v[j][u] + v[j + 1][u];
        acc[u] += t;
      }
    }
    for (; k < nsplit; k += G) {
      const float* src = ws + ((size_t)k * T + t0) * CC + e;
#pragma unroll
      for (int u = 0; u < NT; ++u)
        if (u < nt) acc[u] += src[(size_t)u * CC];
    }
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) part[g * PITCH + el * NT + u] = acc[u];
  __syncthreads();
  for (int idx = threadIdx.x; idx < E * NT; idx += THREADS) {
    const int l = idx / NT, u = idx - l * NT;
    if (u < nt && blk * E + l < CC) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < G; ++w) t += part[w * PITCH + idx];
      const int td = mapped ? tm.m[t0 + u] : t0 + u;
      if (td < 0) continue;
      float* dst = dw + (size_t)(blk * E + l) * T + td;
      *dst = accumulate ? *dst + t : t;
    }
  }
}

struct WPlan {
  int twl, tiles_x, tiles_y, ntiles, IH, IW, plx, co_tiles, ci_tiles, nsplit, lds_bytes;
  bool taps_variant, bf16, bf16_1x1, small1x1, small1x1_mfma, c8, small1x1_c8;
  int pyv, pxv, rv;
  size_t slab_floats;
};

// 3x3 / stride 2 / pad 1 on BF16_C8 tensors (ResNet layer2/3 entry convs): four launches of the stride-1 LDS-DMA kernel, one per
// pixel-parity phase X_pq[y][x] = X[2y+p][2x+q] of the input (gathered by the DMA: WgradArgs.ps) -- tap (ky, kx) of the stride-2
// filter is tap (ky', kx') of a stride-1 3x3 correlation of dY with phase (p, q): p = 0 <-> ky = 1 = ky'; p = 1 <-> ky = 0 = ky'
// or ky = 2, ky' = 1 (same in x).  Each phase's reduce writes its 1 / 2 / 2 / 4 taps straight into dw (TapMap).
inline bool s2_phases(const EssConvDesc* d) {
  return d->fmt0 == ESS_FMT_BF16_C8 && d->fmt_out == ESS_FMT_BF16_C8 && d->ksize == 3 && d->stride == 2 && d->pad == 1 && d->C1 == 0 &&
         d->mode0 == ESS_SRC_DIRECT && d->H_in == 2 * d->H_out && d->W_in == 2 * d->W_out;
}
inline EssConvDesc s2_phase_desc(const EssConvDesc* d) {
  EssConvDesc p = *d;
  p.stride = 1; p.H_in = d->H_out; p.W_in = d->W_out;
  return p;
}

int wvalidate(const EssConvDesc* d) {
  ESS_CHECK_ARG(d != nullptr, "wgrad: null descriptor");
  ESS_CHECK_ARG(d->epilogue == ESS_EPI_LINEAR && d->out_split == 0, "wgrad: only plain convolutions have a weight gradient here");
  ESS_CHECK_ARG(d->mode0 >= ESS_SRC_DIRECT && d->mode0 <= ESS_SRC_ZERO_UP2 && d->mode1 >= ESS_SRC_DIRECT && d->mode1 <= ESS_SRC_ZERO_UP2,
                "wgrad: source modes DIRECT / NEAREST_UP2 / ZERO_UP2 only (the space-to-depth form exists for the frozen encoder's forward)");
  const int cin = d->C0 + d->C1;
  const bool taps = cin == 1 && d->ksize == 7;
  ESS_CHECK_ARG(taps || d->ksize == 1 || d->ksize == 3, "wgrad: k=%d with C_in=%d unsupported", d->ksize, cin);
  ESS_CHECK_ARG(d->stride == 1 || d->stride == 2, "wgrad: stride %d unsupported", d->stride);
  // storage formats: fmt0 (= fmt1) is X's, fmt_out is dY's
  const bool xc8 = d->fmt0 == ESS_FMT_BF16_C8, dc8 = d->fmt_out == ESS_FMT_BF16_C8;
  ESS_CHECK_ARG(d->compute == ESS_COMPUTE_FP32 || d->compute == ESS_COMPUTE_BF16 || d->compute == ESS_COMPUTE_BF16X3, "wgrad: bad compute type");
  if (xc8 || dc8) {
    ESS_CHECK_ARG(d->compute == ESS_COMPUTE_BF16, "wgrad: BF16_C8 tensors need bf16 compute");
    ESS_CHECK_ARG(d->C1 == 0 || d->fmt1 == d->fmt0, "wgrad: both sources of a concat must use the same format");
    ESS_CHECK_ARG(d->C1 == 0 || (d->C0 % 8) == 0, "wgrad: the first BF16_C8 source of a concat must have a multiple of 8 channels");
    if (xc8 && dc8)
      ESS_CHECK_ARG((d->ksize == 3 && d->stride == 1 && d->pad == 1) || (d->ksize == 1 && d->pad == 0 && d->mode0 == ESS_SRC_DIRECT && d->C1 == 0) ||
                        s2_phases(d),
                    "wgrad(BF16_C8): 3x3 / stride 1 / pad 1, 3x3 / stride 2 / pad 1 on even extents (one direct source) and 1x1 / pad 0 convolutions only");
    else if (xc8)
      ESS_CHECK_ARG(d->ksize == 1 && d->stride == 1 && d->pad == 0 && d->C1 == 0 && d->mode0 == ESS_SRC_DIRECT && d->C_out <= 32 && cin <= 32,
                    "wgrad(BF16_C8 X, fp32 dY): the 1x1 head only (C_in, C_out <= 32)");
    else
      ESS_CHECK_ARG(taps, "wgrad(fp32 X, BF16_C8 dY): the 7x7 single-channel stem only");
  }
  return ESS_OK;
}

WPlan wplan(const EssConvDesc* d) {
  WPlan w{};
  const int cin = d->C0 + d->C1, KS = d->ksize, S = d->stride;
  w.taps_variant = (cin == 1 && KS == 7);
  w.small1x1 = KS == 1 && S == 1 && d->pad == 0 && d->C1 == 0 && d->mode0 == ESS_SRC_DIRECT && d->C_out <= 16 && cin <= 64;
  w.bf16 = d->compute == ESS_COMPUTE_BF16 && KS == 3 && S == 1 && d->pad == 1;
  // 1x1 / stride 1 with at least a tile of channels: the centre tap of the same kernel (the fp32 tile kernel ran the ResNet
  // downsample gradients at 10 TFLOP/s); needs the fast variant's geometry (rows of 8 pixels, direct / upsampled sources)
  w.bf16_1x1 = d->compute == ESS_COMPUTE_BF16 && KS == 1 && S == 1 && d->pad == 0 && !w.small1x1 && (d->W_in % 8) == 0 &&
               (d->W_out % 8) == 0 && d->mode0 != ESS_SRC_ZERO_UP2 && d->mode1 != ESS_SRC_ZERO_UP2;
  w.bf16 = w.bf16 || w.bf16_1x1;
  w.c8 = d->fmt0 == ESS_FMT_BF16_C8 && d->fmt_out == ESS_FMT_BF16_C8;  // both operands BF16_C8: conv_wgrad_c8.hip
  w.small1x1_c8 = d->fmt0 == ESS_FMT_BF16_C8 && d->fmt_out != ESS_FMT_BF16_C8;
  if (w.c8) { w.bf16 = true; w.bf16_1x1 = KS == 1; w.small1x1 = false; }
  if (w.small1x1_c8) { w.bf16 = w.bf16_1x1 = false; w.small1x1 = true; }
  const int npx = w.bf16 ? 128 : 64;
  // pixel tile of 64 (fp32) / 128 (bf16): pick the width that wastes the least
  double best = 1e300;
  for (int twl = 5; twl >= (w.bf16 ? 4 : 3); --twl) {
    const int tw = 1 << twl, th = npx >> twl;
    const double c = (double)ceil_div(d->W_out, tw) * tw * ceil_div(d->H_out, th) * th + 1e-3 * (5 - twl);
    if (c < best) { best = c; w.twl = twl; }
  }
  if (w.c8 && KS == 3) w.twl = 4;  // the LDS-DMA kernel: 16 x 8 tiles (least halo, two workgroups per CU)
  const int tw = 1 << w.twl, th = npx >> w.twl;
  w.tiles_x = ceil_div(d->W_out, tw); w.tiles_y = ceil_div(d->H_out, th);
  w.ntiles = d->N * w.tiles_x * w.tiles_y;
  w.IH = (th - 1) * S + KS; w.IW = (tw - 1) * S + KS;
  w.plx = (w.IH * w.IW) | 1;
  w.co_tiles = ceil_div(d->C_out, 64);
  w.ci_tiles = w.taps_variant ? 1 : ceil_div(cin, 64);
  w.slab_floats = (size_t)KS * KS * d->C_out * cin + d->C_out;
  const int pairs = w.co_tiles * w.ci_tiles;
  // the bf16 kernel runs one workgroup per CU (512-register waves): aim at one full round of the 256 CUs
  int ns = w.bf16 ? ceil_div(256, pairs) : ceil_div(2048, pairs);
  if (ns > w.ntiles) ns = w.ntiles;
  const size_t cap = ((size_t)64 << 20) / (w.slab_floats * 4);
  if ((size_t)ns > cap) ns = (int)(cap ? cap : 1);
  if (ns < 1) ns = 1;
  w.nsplit = ns;
  if (w.small1x1) {
    const int chunks = d->N * ceil_div(d->H_out * d->W_out, 128);
    w.nsplit = chunks < 1024 ? chunks : 1024;
    w.small1x1_mfma = cin <= 32 && ((d->H_out * d->W_out) % 64) == 0;
    if (w.small1x1_c8) {
      w.small1x1_mfma = false;
      if (w.nsplit > 1024) w.nsplit = 1024;  // (33 KB of LDS per workgroup: four per CU)
    }
    if (w.small1x1_mfma) {  // one slab per workgroup of 4 waves, two workgroups per CU
      const int groups = d->N * (d->H_out * d->W_out / 64);
      w.nsplit = groups < 4 * 512 ? ceil_div(groups, 4) : 512;
    }
  }
  w.lds_bytes = (64 * 65 + (w.taps_variant ? w.IH * w.IW : 64 * w.plx)) * 4;
  if (w.bf16) {
    w.rv = tw / 8 + 2;
    w.pyv = (th * tw / 8) | 1;
    w.pxv = ((th + 2) * w.rv) | 1;
    w.lds_bytes = 64 * (w.pyv + w.pxv) * 16;
  }
  return w;
}

template <typename K>
int raise_lds(K kernel, int bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      ess_set_error("hipFuncSetAttribute(%d B LDS): %s", bytes, hipGetErrorString(e));
      return ESS_ELAUNCH;
    }
  }
  return ESS_OK;
}

}  // namespace

// ESS_COMPUTE_BF16X3: 3x3 / stride 1 / pad 1 on the fp32-staged bf16 kernels with split operands, everything else exact fp32
static EssConvDesc wresolve(const EssConvDesc* d, bool* split) {
  EssConvDesc r = *d;
  *split = false;
  if (d->compute == ESS_COMPUTE_BF16X3) {
    *split = d->ksize == 3 && d->stride == 1 && d->pad == 1;
    r.compute = *split ? ESS_COMPUTE_BF16 : ESS_COMPUTE_FP32;
  }
  return r;
}

// Split operands through the BF16_C8 kernel (conv_wgrad_c8.hip: LDS-DMA staging, twice the rate of the fp32-staged kernels): X and
// dY are split ONCE into hi / lo BF16_C8 copies (workspace), then three accumulating launches -- (dY_hi, X_hi), (dY_hi, X_lo),
// (dY_lo, X_hi).  Needs whole 8-channel blocks where the BF16_C8 kernels do; otherwise the fp32-staged kernels make three passes.
int ess_split_bf16_c8_internal(const float* x, void* hi, void* lo, int N, int C, int H, int W, hipStream_t st);
static bool split_via_c8(const EssConvDesc* d) {
  static const bool on = [] { const char* e = getenv("ESS_X3_WGRAD_C8"); return !(e && e[0] == '0'); }();
  return on && d->compute == ESS_COMPUTE_BF16X3 && d->ksize == 3 && d->stride == 1 && d->pad == 1 && (d->C1 == 0 || (d->C0 % 8) == 0) &&
         d->mode0 != ESS_SRC_ZERO_UP2 && d->mode1 != ESS_SRC_ZERO_UP2 && (d->C1 == 0 || d->mode1 == ESS_SRC_DIRECT);
}
static EssConvDesc c8_desc(const EssConvDesc* d) {
  EssConvDesc c = *d;
  c.compute = ESS_COMPUTE_BF16;
  c.fmt0 = c.fmt1 = c.fmt_out = ESS_FMT_BF16_C8;
  return c;
}
static size_t c8_copy_bytes(int N, int C, int H, int W) { return (((size_t)N * ((C + 7) / 8) * H * W * 16) + 255) & ~(size_t)255; }
struct SplitCopies { size_t x0, x1, dy, total; };
static SplitCopies split_copies(const EssConvDesc* d) {
  SplitCopies s{};
  const int s0 = d->mode0 != ESS_SRC_DIRECT ? 1 : 0, s1 = d->mode1 != ESS_SRC_DIRECT ? 1 : 0;
  s.x0 = c8_copy_bytes(d->N, d->C0, d->H_in >> s0, d->W_in >> s0);
  s.x1 = d->C1 ? c8_copy_bytes(d->N, d->C1, d->H_in >> s1, d->W_in >> s1) : 0;
  s.dy = c8_copy_bytes(d->N, d->C_out, d->H_out, d->W_out);
  s.total = 2 * (s.x0 + s.x1 + s.dy);
  return s;
}

// the lo copies of a fused split-operand call, handed to the BF16_C8 3x3 launch of the SAME thread's nested entry (no public
// signature changes; null outside that nested call)
struct WgradPasses { int npass, bias_mask; const void* dy[3]; const void* x0[3]; const void* x1[3]; mutable bool used; };  // used: the nested launch took the passes
static thread_local const WgradPasses* g_passes = nullptr;

extern "C" size_t ess_conv2d_wgrad_workspace(const EssConvDesc* d) {
  if (wvalidate(d)) return 0;
  if (split_via_c8(d)) {
    const EssConvDesc dc = c8_desc(d);
    const size_t slabs = (ess_conv2d_wgrad_workspace(&dc) + 255) & ~(size_t)255;
    return slabs ? slabs + split_copies(d).total : 0;
  }
  bool osplit;
  const EssConvDesc dres = wresolve(d, &osplit);
  d = &dres;
  const EssConvDesc dp = s2_phases(d) ? s2_phase_desc(d) : *d;
  const WPlan w = wplan(&dp);
  return (size_t)w.nsplit * w.slab_floats * 4;
}

extern "C" int ess_conv2d_wgrad(const EssConvDesc* d, const void* src0_, const void* src1_, const void* dy_, float* dw,
                                float* db, int accumulate, void* workspace, size_t workspace_bytes, ess_stream_t stream) {
  const float* src0 = (const float*)src0_; const float* src1 = (const float*)src1_; const float* dy = (const float*)dy_;
  int rc = wvalidate(d);
  if (rc) return rc;
  if (split_via_c8(d)) {
    ESS_CHECK_ARG(src0 && dy && dw && workspace, "wgrad: null pointer");
    ESS_CHECK_ARG(d->C1 == 0 || src1, "wgrad: second source missing");
    ESS_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, "wgrad: the workspace must be 256-byte aligned");
    const EssConvDesc dc = c8_desc(d);
    const size_t slabs = (ess_conv2d_wgrad_workspace(&dc) + 255) & ~(size_t)255;
    const SplitCopies sc = split_copies(d);
    ESS_CHECK_ARG(slabs && workspace_bytes >= slabs + sc.total, "wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)workspace + slabs;
    char* x0h = base; char* x0l = x0h + sc.x0;
    char* x1h = x0l + sc.x0; char* x1l = x1h + sc.x1;
    char* dyh = x1l + sc.x1; char* dyl = dyh + sc.dy;
    const int s0 = d->mode0 != ESS_SRC_DIRECT ? 1 : 0, s1 = d->mode1 != ESS_SRC_DIRECT ? 1 : 0;
    if ((rc = ess_split_bf16_c8_internal(src0, x0h, x0l, d->N, d->C0, d->H_in >> s0, d->W_in >> s0, st))) return rc;
    if (d->C1 && (rc = ess_split_bf16_c8_internal(src1, x1h, x1l, d->N, d->C1, d->H_in >> s1, d->W_in >> s1, st))) return rc;
    if ((rc = ess_split_bf16_c8_internal(dy, dyh, dyl, d->N, d->C_out, d->H_out, d->W_out, st))) return rc;
    // ONE launch of the LDS-DMA kernel walking the tile list three times (WgradBArgs::x3): one slab set, one reduce (round 5; the
    // three accumulating launches of round 4 wrote and reduced the slabs three times: ESS_X3_WGRAD_FUSED=0 brings them back)
    static const bool fused = [] { const char* e = getenv("ESS_X3_WGRAD_FUSED"); return !(e && e[0] == '0'); }();
    if (fused) {
      const void* x1h_ = d->C1 ? x1h : nullptr; const void* x1l_ = d->C1 ? x1l : nullptr;
      const WgradPasses ps{3, 0b101, {dyh, dyh, dyl}, {x0h, x0l, x0h}, {x1h_, x1l_, x1h_}, false};
      g_passes = &ps;
      rc = ess_conv2d_wgrad(&dc, x0h, x1h_, dyh, dw, db, accumulate, workspace, slabs, stream);
      g_passes = nullptr;
      if (rc || ps.used) return rc;
      // (the plan did not take the LDS-DMA kernel: the first set is done, the other two follow as accumulating launches)
      if ((rc = ess_conv2d_wgrad(&dc, x0l, x1l_, dyh, dw, nullptr, 1, workspace, slabs, stream))) return rc;
      return ess_conv2d_wgrad(&dc, x0h, x1h_, dyl, dw, db, 1, workspace, slabs, stream);
    }
    if ((rc = ess_conv2d_wgrad(&dc, x0h, d->C1 ? x1h : nullptr, dyh, dw, db, accumulate, workspace, slabs, stream))) return rc;
    if ((rc = ess_conv2d_wgrad(&dc, x0l, d->C1 ? x1l : nullptr, dyh, dw, nullptr, 1, workspace, slabs, stream))) return rc;
    return ess_conv2d_wgrad(&dc, x0h, d->C1 ? x1h : nullptr, dyl, dw, db, 1, workspace, slabs, stream);
  }
  bool osplit;
  const EssConvDesc dres = wresolve(d, &osplit);
  d = &dres;
  ESS_CHECK_ARG(src0 && dy && dw && workspace, "wgrad: null pointer");
  ESS_CHECK_ARG(d->C1 == 0 || src1, "wgrad: second source missing");
  if (s2_phases(d)) {
    const EssConvDesc dp = s2_phase_desc(d);
    const WPlan w = wplan(&dp);
    ESS_CHECK_ARG(workspace_bytes >= (size_t)w.nsplit * w.slab_floats * 4, "wgrad: workspace too small");
    ESS_CHECK_ARG((((uintptr_t)src0 | (uintptr_t)dy) & 15) == 0, "wgrad: BF16_C8 tensors must be 16-byte aligned");
    const int cin = d->C0, CC = d->C_out * cin;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(w.co_tiles * w.ci_tiles * w.nsplit));
    for (int ph = 0; ph < 4; ++ph) {
      const int p = ph >> 1, q = ph & 1;
      WgradBArgs bb{};
      WgradArgs& a = bb.w;
      a.src0 = src0; a.src1 = nullptr; a.dy = dy; a.ws = (float*)workspace;
      a.ws_b = (db && ph == 0) ? a.ws + (size_t)w.nsplit * 9 * CC : nullptr;
      a.N = d->N; a.Hin = dp.H_in; a.Win = dp.W_in; a.C0 = d->C0; a.C1 = 0; a.mode0 = ESS_SRC_DIRECT; a.mode1 = ESS_SRC_DIRECT;
      a.Cout = d->C_out; a.Hout = d->H_out; a.Wout = d->W_out; a.pad = 1;
      a.twl = w.twl; a.tiles_x = w.tiles_x; a.tiles_y = w.tiles_y; a.ntiles = w.ntiles;
      a.IH = w.IH; a.IW = w.IW; a.plx = w.plx; a.ci_tiles = w.ci_tiles; a.npairs = w.co_tiles * w.ci_tiles; a.nsplit = w.nsplit;
      a.dy_c8 = 1; a.ps = 1; a.pp = p; a.pq = q;
      bb.pyv = w.pyv; bb.pxv = w.pxv; bb.rv = w.rv;
        if ((rc = wgrad_c8_launch(bb, 9, 1, 2 * w.lds_bytes, grid, st))) return rc;
      TapMap tm{};
      tm.mapped = 1;
      for (int t = 0; t < 9; ++t) {
        const int kyp = t / 3, kxp = t % 3;
        const int ky = p == 0 ? (kyp == 1 ? 1 : -1) : (kyp == 0 ? 0 : kyp == 1 ? 2 : -1);
        const int kx = q == 0 ? (kxp == 1 ? 1 : -1) : (kxp == 0 ? 0 : kxp == 1 ? 2 : -1);
        tm.m[t] = (signed char)((ky < 0 || kx < 0) ? -1 : ky * 3 + kx);
      }
      const int nblk_w = ceil_div(CC, 64), nblk_b = a.ws_b ? ceil_div(d->C_out, 64) : 0;
      if (CC >= 32768)
        hipLaunchKernelGGL((wgrad_reduce_kernel<64, 9, 256>), dim3((unsigned)(nblk_w + nblk_b)), dim3(256), 0, st, a.ws, a.ws_b, dw, db, w.nsplit, 9,
                           CC, d->C_out, nblk_w, accumulate, tm);
      else
        hipLaunchKernelGGL((wgrad_reduce_kernel<64, 1, 1024>), dim3((unsigned)(nblk_w + nblk_b), 9u), dim3(1024), 0, st, a.ws, a.ws_b, dw, db,
                           w.nsplit, 9, CC, d->C_out, nblk_w, accumulate, tm);
    }
    return ess_launch_status("conv2d_wgrad(3x3 stride 2 by phases)");
  }
  const WPlan w = wplan(d);
  ESS_CHECK_ARG(workspace_bytes >= (size_t)w.nsplit * w.slab_floats * 4, "wgrad: workspace too small");
  ESS_CHECK_ARG(w.lds_bytes <= 160 * 1024, "wgrad: LDS tile %d B too large", w.lds_bytes);
  const int cin = d->C0 + d->C1, T = d->ksize * d->ksize;
  WgradArgs a{};
  a.src0 = src0; a.src1 = src1; a.dy = dy;
  a.ws = (float*)workspace;
  a.ws_b = db ? a.ws + (size_t)w.nsplit * T * d->C_out * cin : nullptr;
  a.N = d->N; a.Hin = d->H_in; a.Win = d->W_in; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0; a.mode1 = d->mode1;
  a.Cout = d->C_out; a.Hout = d->H_out; a.Wout = d->W_out; a.pad = d->pad;
  a.twl = w.twl; a.tiles_x = w.tiles_x; a.tiles_y = w.tiles_y; a.ntiles = w.ntiles;
  a.IH = w.IH; a.IW = w.IW; a.plx = w.plx; a.ci_tiles = w.ci_tiles; a.npairs = w.co_tiles * w.ci_tiles; a.nsplit = w.nsplit;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(w.co_tiles * w.ci_tiles * w.nsplit));
#define ESS_WG(KS_, S_)                                                                               \
  do {                                                                                                \
    if ((rc = raise_lds(wgrad_f32_kernel<KS_, S_>, w.lds_bytes))) return rc;                         \
    hipLaunchKernelGGL((wgrad_f32_kernel<KS_, S_>), grid, dim3(256), w.lds_bytes, st, a);             \
  } while (0)
  a.dy_c8 = d->fmt_out == ESS_FMT_BF16_C8;
  if (d->fmt0 == ESS_FMT_BF16_C8 || a.dy_c8)
    ESS_CHECK_ARG(((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)dy)) & 15) == 0, "wgrad: BF16_C8 tensors must be 16-byte aligned");
  if (w.small1x1_c8) {
    if ((rc = wgrad_small1x1_c8_launch(a, w.nsplit, st))) return rc;
  } else if (w.c8) {
    WgradBArgs bb{};
    bb.w = a; bb.pyv = w.pyv; bb.pxv = w.pxv; bb.rv = w.rv;
    if (w.bf16_1x1) bb.w.pad = 1;  // tile geometry of the 3x3 kernel: the X tile starts one row / column before the output tile
    if (g_passes) {
      ESS_CHECK_ARG(!w.bf16_1x1 && d->stride == 1, "wgrad: several (dY, X) sets in one launch exist for the 3x3 / stride-1 LDS-DMA kernel only");
      bb.npass = g_passes->npass; bb.bias_mask = g_passes->bias_mask;
      g_passes->used = true;
      for (int q = 0; q < 3; ++q) { bb.p_dy[q] = g_passes->dy[q]; bb.p_x0[q] = g_passes->x0[q]; bb.p_x1[q] = g_passes->x1[q]; }
    }
    if ((rc = wgrad_c8_launch(bb, w.bf16_1x1 ? 1 : 9, d->stride, 2 * w.lds_bytes, grid, st))) return rc;
  } else if (w.small1x1 && w.small1x1_mfma && ((((uintptr_t)a.src0) | ((uintptr_t)a.dy)) & 15) == 0) {
    const int per_img = d->H_out * d->W_out / 64;
    hipLaunchKernelGGL(wgrad_small1x1_mfma_kernel, dim3(w.nsplit), dim3(256), 0, st, a, per_img, d->N * per_img);
  } else if (w.small1x1) {
    const int per_img = ceil_div(d->H_out * d->W_out, 128);
    hipLaunchKernelGGL(wgrad_small1x1_kernel, dim3(w.nsplit), dim3(256), 0, st, a, per_img, d->N * per_img);
  } else if (w.bf16) {
    WgradBArgs bb{};
    bb.w = a; bb.pyv = w.pyv; bb.pxv = w.pxv; bb.rv = w.rv;
    bb.split = osplit ? 1 : 0;
    const bool fast = (d->W_in % 8) == 0 && (d->W_out % 8) == 0 && d->mode0 != ESS_SRC_ZERO_UP2 && d->mode1 != ESS_SRC_ZERO_UP2;
    if (w.bf16_1x1) {
      bb.w.pad = 1;  // tile geometry of the 3x3 kernel: the X tile starts one row / column before the output tile
      if ((rc = raise_lds(wgrad_bf16_k3s1_fast_kernel<1>, 2 * w.lds_bytes))) return rc;
      hipLaunchKernelGGL(wgrad_bf16_k3s1_fast_kernel<1>, grid, dim3(256), 2 * w.lds_bytes, st, bb);
    } else if (fast) {
      if ((rc = raise_lds(wgrad_bf16_k3s1_fast_kernel<9>, 2 * w.lds_bytes))) return rc;  // two LDS stages
      hipLaunchKernelGGL(wgrad_bf16_k3s1_fast_kernel<9>, grid, dim3(256), 2 * w.lds_bytes, st, bb);
    } else {
      if ((rc = raise_lds(wgrad_bf16_k3s1_kernel, w.lds_bytes))) return rc;
      hipLaunchKernelGGL(wgrad_bf16_k3s1_kernel, grid, dim3(256), w.lds_bytes, st, bb);
    }
  } else if (w.taps_variant) {
    ESS_CHECK_ARG(d->stride == 2, "wgrad: 7x7 stem variant is stride 2 only");
    if ((rc = raise_lds(wgrad_taps_kernel<7, 2>, w.lds_bytes))) return rc;
    hipLaunchKernelGGL((wgrad_taps_kernel<7, 2>), grid, dim3(256), w.lds_bytes, st, a);
    // the taps variant has no bias path
    a.ws_b = nullptr;
  } else if (d->ksize == 3 && d->stride == 1) ESS_WG(3, 1);
  else if (d->ksize == 3 && d->stride == 2) ESS_WG(3, 2);
  else if (d->ksize == 1 && d->stride == 1) ESS_WG(1, 1);
  else ESS_WG(1, 2);
#undef ESS_WG
  rc = ess_launch_status("conv2d_wgrad");
  if (rc) return rc;
  ESS_CHECK_ARG(!(w.taps_variant && db), "wgrad: the stem variant has no bias gradient");
  const int CC = d->C_out * cin;
  const bool wide = T == 9 && CC >= 32768;
  const int E = 64;
  const int nblk_w = ceil_div(CC, E), nblk_b = (db && a.ws_b) ? ceil_div(d->C_out, E) : 0;
  if (wide)
    hipLaunchKernelGGL((wgrad_reduce_kernel<64, 9, 256>), dim3((unsigned)(nblk_w + nblk_b)), dim3(256), 0, st, a.ws, a.ws_b, dw, db, w.nsplit, T,
                       CC, d->C_out, nblk_w, accumulate, TapMap{});
  else
    hipLaunchKernelGGL((wgrad_reduce_kernel<64, 1, 1024>), dim3((unsigned)(nblk_w + nblk_b), (unsigned)T), dim3(1024), 0, st, a.ws, a.ws_b, dw, db,
                       w.nsplit, T, CC, d->C_out, nblk_w, accumulate, TapMap{});
  return ess_launch_status("conv2d_wgrad_reduce");
}


// Two (or three) (dY, X) sets of the SAME convolution into one weight gradient: dw (+)= sum_s wgrad(X_s, dY_s).  BF16_C8 3x3 / stride 1
// / pad 1 (the LDS-DMA kernel): ONE launch walks the tile list once per set -- one prologue, one slab set, one reduce --; any other
// geometry: one accumulating call per set.  The decoder's two weight-gradient passes of a UDA step use it (functional.py, deferred
// weight gradients).
extern "C" int ess_conv2d_wgrad_sets(const EssConvDesc* d, int32_t n_sets, const void* const* src0, const void* const* src1,
                                     const void* const* dy, float* dw, float* db, int32_t accumulate, void* workspace,
                                     size_t workspace_bytes, ess_stream_t stream) {
  ESS_CHECK_ARG(d && src0 && dy && n_sets >= 1 && n_sets <= 3, "wgrad_sets: 1 to 3 sets");
  for (int q = 0; q < n_sets; ++q) ESS_CHECK_ARG(src0[q] && dy[q] && (d->C1 == 0 || (src1 && src1[q])), "wgrad_sets: null tensor in set %d", q);
  const bool one_launch = n_sets > 1 && d->compute == ESS_COMPUTE_BF16 && d->fmt0 == ESS_FMT_BF16_C8 && d->fmt_out == ESS_FMT_BF16_C8 &&
                          d->ksize == 3 && d->stride == 1 && d->pad == 1 && !g_passes;
  if (one_launch) {
    WgradPasses ps{n_sets, (1 << n_sets) - 1, {}, {}, {}, false};
    for (int q = 0; q < n_sets; ++q) { ps.dy[q] = dy[q]; ps.x0[q] = src0[q]; ps.x1[q] = d->C1 ? src1[q] : nullptr; }
    g_passes = &ps;
    int rc = ess_conv2d_wgrad(d, src0[0], d->C1 ? src1[0] : nullptr, dy[0], dw, db, accumulate, workspace, workspace_bytes, stream);
    g_passes = nullptr;
    if (rc || ps.used) return rc;
    for (int q = 1; q < n_sets; ++q)  // (the plan did not take the LDS-DMA kernel: set 0 is done, the others accumulate)
      if ((rc = ess_conv2d_wgrad(d, src0[q], d->C1 ? src1[q] : nullptr, dy[q], dw, db, 1, workspace, workspace_bytes, stream))) return rc;
    return ESS_OK;
  }
  for (int q = 0; q < n_sets; ++q) {
    const int rc = ess_conv2d_wgrad(d, src0[q], d->C1 ? src1[q] : nullptr, dy[q], dw, db, q ? 1 : accumulate, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return ESS_OK;
}

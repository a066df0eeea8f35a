// Shared declarations of the weight-gradient kernels (conv_wgrad.hip: fp32 / fp32-NCHW-staged bf16 kernels, split-K reduce,
// entry points; conv_wgrad_c8.hip: the kernels that stage BF16_C8 tensors).
#pragma once
#include "common.h"

struct WgradArgs {
  const float* src0;
  const float* src1;
  const float* dy;
  float* ws;      // [nsplit][taps][Cout][Cin]
  float* ws_b;    // [nsplit][Cout] or null
  int N, Hin, Win, C0, C1, mode0, mode1, Cout, Hout, Wout, pad;
  int twl;        // pixel tile = (64>>twl) rows x (1<<twl) cols
  int tiles_x, tiles_y, ntiles;
  int IH, IW, plx;  // X tile rows/cols, per-channel pitch (odd)
  int ci_tiles, npairs, nsplit;
  int dy_c8;      // dY is a BF16_C8 tensor (7x7 stem variant)
  // stride-2 3x3 through the stride-1 LDS-DMA kernel: ps = 1 reads the first source at pixel (2 y + pp, 2 x + pq) -- one parity
  // phase of it, gathered by the DMA itself; Hin / Win are then the phase's extents (= the output's).  ps = 0: plain.
  int ps, pp, pq;
};

constexpr unsigned OOBW = 0x80000000u;  // beyond any buffer: bounds-checked loads return 0
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));


__device__ __forceinline__ u32x4w cvt8(const float (&v)[8]) {
  bf16x8w b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
  return __builtin_bit_cast(u32x4w, b);
}
// split operands: the low part lo = bf16(v - float(bf16(v))) of eight values
__device__ __forceinline__ u32x4w cvt8_lo(const float (&v)[8]) {
  bf16x8w b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)(v[j] - (float)(__bf16)v[j]);
  return __builtin_bit_cast(u32x4w, b);
}
// bytes [off, off+16) of the 32-byte concatenation lo|hi
__device__ __forceinline__ u32x4w shift_left1(u32x4w lo, u32x4w hi) {  // window starting 14 bytes into lo (one pixel earlier than hi)
  u32x4w r;
  r[0] = __builtin_amdgcn_alignbyte(hi[0], lo[3], 2);
  r[1] = __builtin_amdgcn_alignbyte(hi[1], hi[0], 2);
  r[2] = __builtin_amdgcn_alignbyte(hi[2], hi[1], 2);
  r[3] = __builtin_amdgcn_alignbyte(hi[3], hi[2], 2);
  return r;
}
__device__ __forceinline__ u32x4w shift_right1(u32x4w lo, u32x4w hi) {  // window starting 2 bytes into lo (one pixel later)
  u32x4w r;
  r[0] = __builtin_amdgcn_alignbyte(lo[1], lo[0], 2);
  r[1] = __builtin_amdgcn_alignbyte(lo[2], lo[1], 2);
  r[2] = __builtin_amdgcn_alignbyte(lo[3], lo[2], 2);
  r[3] = __builtin_amdgcn_alignbyte(hi[0], lo[3], 2);
  return r;
}


struct WgradBArgs {
  WgradArgs w;
  int pyv, pxv, rv;  // per-channel pitches (16-B vectors) of the dY / X tiles, vectors per X row
  int split;         // split-operand bf16 (ESS_COMPUTE_BF16X3, fp32-staged kernels): every pixel tile is contracted three times --
                     // (dY_hi, X_hi), (dY_hi, X_lo), (dY_lo, X_hi) -- into the same accumulators
  // several (dY, X) sets through ONE launch of the BF16_C8 LDS-DMA kernel (round 5): npass > 1 -> the tile list is walked npass
  // times, pass p = tile / ntiles reading (p_dy[p], p_x0[p], p_x1[p]) into the same accumulators -- one prologue, one slab set and
  // one reduce instead of npass.  Users: split operands (three passes: (dY_hi, X_hi), (dY_hi, X_lo), (dY_lo, X_hi); bias_mask 0b101:
  // dY_hi is contracted twice, the bias gradient takes it once) and the two weight-gradient passes a decoder layer sees per UDA
  // step (ess_conv2d_wgrad_sets; bias_mask 0b11).  npass <= 1: w.dy / w.src0 / w.src1.
  int npass;
  int bias_mask;
  const void* p_dy[3];
  const void* p_x0[3];
  const void* p_x1[3];
};

// conv_wgrad_c8.hip: launchers (the caller has validated the geometry and sized the workspace)
int wgrad_c8_launch(const WgradBArgs& b, int taps, int sx, int lds_bytes, dim3 grid, hipStream_t st);
int wgrad_small1x1_c8_launch(const WgradArgs& a, int nsplit, hipStream_t st);

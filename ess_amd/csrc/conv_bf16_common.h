// Shared bits of the bf16-MFMA convolution kernels (conv_bf16*.hip): fragment types, staging helpers, and the launchers each
// translation unit exports to the dispatcher in conv_bf16.hip.  The three kernel families live in separate files so that the
// in-tree build compiles them in parallel.
#pragma once
#define ESS_FAST_ACT true  // (common.h: 1-ulp reciprocal in the gate activations of the bf16 kernels)
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace essconv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
  return __builtin_bit_cast(u32x4, b);
}

__device__ __forceinline__ u32x4 pack8h(const float (&v)[8]) {  // IEEE half, saturating (ESS_COMPUTE_F16)
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = ess_f16_sat(v[j]);
  return __builtin_bit_cast(u32x4, b);
}
// one K-step of the matrix cores on two 16-byte fragments: bfloat16 or, H, IEEE-half elements (same rate, same layouts)
typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
template <bool H>
__device__ __forceinline__ f32x16 ess_mfma16(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8m, a), __builtin_bit_cast(f16x8m, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// pixel positions a thread stages per 8-channel block (compile-time bound of the register prefetch), by filter geometry
constexpr int kpc(int ks, int s) { return stage_kpc(ks, s); }
constexpr unsigned OOB = 0x80000000u;  // beyond any buffer: the bounds-checked load returns 0

// conv_bf16_generic.hip
void conv_bf16_launch_generic(int key, int mb, int cb8, int epi, bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a);  // (a.f16: the H instantiations, in every launcher below)
// conv_bf16_ws.hip
void conv_bf16_launch_ws(int mb, int epi, bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a);
// conv_bf16_head.hip: 5x5 / stride 1 on a 1- or 2-channel fp32 image (the recurrent encoder's head), reads the tap-paired pack
bool conv_bf16_head_applies(const EssConvDesc* d, const EssConvPlan& pl);
void conv_bf16_launch_head(const EssConvDesc* d, const EssConvPlan& pl, hipStream_t st, const ConvKArgs& a);
// 7x7 / stride 2 on one fp32 channel (the image encoder's stem), reads the generic bf16 pack
bool conv_bf16_stem_applies(const EssConvDesc* d, const EssConvPlan& pl);
void conv_bf16_launch_stem(const EssConvDesc* d, const EssConvPlan& pl, hipStream_t st, const ConvKArgs& a);
// conv_bf16_wide.hip: 3x3 / stride 1, BF16_C8 sources and outputs, five pixel blocks per matrix wave (tile rows *th x 16 columns)
void conv_bf16_wide_tile(int mbw, int cw, int* th, int* tw);
void conv_bf16_launch_wide(int mbw, int cw, int epi, dim3 grid, hipStream_t st, const ConvKArgs& a);
// conv_bf16_poly.hip: 3x3 / stride 1 / pad 1 of ONE nearest-x2-upsampled BF16_C8 source, polyphase (2 x 2 effective filters per output parity)
void conv_bf16_poly_tile(int* th, int* tw, int* max_cin);
void conv_bf16_launch_poly(dim3 grid, hipStream_t st, const ConvKArgs& a);
// conv_bf16_pair.hip
void conv_bf16_launch_pair(int stride, int mb, bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a);

}  // namespace essconv

// Shared helpers for the ESS HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ess_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void ess_set_error(const char* fmt, ...);

#define ESS_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ess_set_error(__VA_ARGS__);           \
      return ESS_EINVAL;                    \
    }                                       \
  } while (0)

static inline int ess_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ess_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return ESS_ELAUNCH;
  }
  return ESS_OK;
}

// dynamic LDS above 64 KiB must be opted into per kernel (up to the 160 KiB of a CDNA4 CU)
// (the attribute is per kernel function and device and only ever needs raising: it is set ONCE per (kernel, device, size
// class) -- a mutex-guarded table lookup per launch instead of a runtime call; the table is a cache of what the runtime was
// told, not state a caller can observe)
void ess_allow_lds_impl(const void* kernel, size_t bytes);
template <typename K>
static inline void ess_allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) ess_allow_lds_impl((const void*)kernel, bytes);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// XCD-aware block order.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (each with a
// private 4 MiB L2).  Remap so that every XCD walks one CONTIGUOUS range of logical ids: workgroups that share an
// operand (all channel tiles of one input tile, all channel-tile pairs of one pixel split) then run on the same XCD at
// about the same time and the operand is fetched into that L2 once.  Bijective for any total; a pure speed choice.
__device__ __forceinline__ int xcd_remap(int b, int total) {
  constexpr int NX = 8;
  const int q = total / NX, r = total % NX;
  const int xcd = b % NX, i = b / NX;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// Gate activations of the fused epilogues on the hardware exp2 unit (v_exp_f32, ~1 ulp; libm's expf / tanhf cost 25-40 VALU
// instructions per call, which made the ConvLSTM epilogue as expensive as the K loop of the 8-chunk level-0 gate conv).
// FAST = false: `__frcp_rn`, the IEEE-exact division sequence (two v_div_scale, v_rcp, five FMAs, v_div_fmas,
// v_div_fixup -- 14 VALU instructions per sigmoid) for the exact-fp32 configuration, whose parity contract is 1e-3 on logits with
// exact argmax.  FAST = true (the bf16 configuration's translation units define ESS_FAST_ACT): v_rcp_f32, 1 ulp, 4 instructions --
// the ConvLSTM epilogue evaluates five activations per hidden value and cycle stamps put 12.7 k of its 19.8 k cycles in that
// arithmetic; the operands there are bf16 products to begin with.
template <bool FAST>
__device__ __forceinline__ float ess_rcp_t(float x) {
  if constexpr (FAST) return __builtin_amdgcn_rcpf(x);
  else return __frcp_rn(x);
}
template <bool FAST>
__device__ __forceinline__ float ess_sigmoid_t(float x) { return ess_rcp_t<FAST>(1.0f + __expf(-x)); }
template <bool FAST>
__device__ __forceinline__ float ess_tanh_t(float x) {
  const float t = __expf(-2.0f * fabsf(x));  // in (0, 1]: no overflow for any x
  return copysignf((1.0f - t) * ess_rcp_t<FAST>(1.0f + t), x);
}
#ifndef ESS_FAST_ACT
#define ESS_FAST_ACT false
#endif
#define ess_sigmoid(x_) ess_sigmoid_t<ESS_FAST_ACT>(x_)
#define ess_tanh(x_) ess_tanh_t<ESS_FAST_ACT>(x_)

// wave64 all-lanes sum (butterfly through DPP/shuffles)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of a double; blockDim.x must be a multiple of 64, <= 1024; red: >= 16 doubles of LDS.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

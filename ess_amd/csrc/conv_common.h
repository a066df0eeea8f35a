// Shared pieces of the direct-convolution kernels (fp32-MFMA and bf16-MFMA variants): kernel argument block,
// tile-geometry choice, descriptor validation, weight-pack row mapping.
#pragma once
#include "common.h"
#include <stdlib.h>

#ifndef ESS_EPI_AUX
#define ESS_EPI_AUX 2  // cache policy of the ConvLSTM state traffic (c_prev in, c / h' out): 2 = nt, streamed past the weights and halos in L2
#endif
#ifndef ESS_GRU_AUX
#define ESS_GRU_AUX 2  // same for the F32_C8 ConvGRU states h, u, r*h (T = 20 ConvGRU step 43.18 -> 42.78 ms)
#endif
#ifndef ESS_C8_AUX
#define ESS_C8_AUX 0  // same for the BF16_C8 outputs of the straight-line LINEAR epilogues (forward AND data-gradient forms): 0 everywhere -- nt measured +-0 in the ws / tap-paired kernels and a net loss inside the train step for the wide-tile kernel (conv_bf16_wide.hip header); a -DESS_C8_AUX=2 build changes every one of those stores
#endif
namespace essconv {

constexpr int NBW = 2;  // pixel blocks per wave

struct ConvKArgs {
  const float* src0;
  const float* src1;
  const void* wpk;
  const float* scale;
  const float* shift;
  const float* residual;
  const float* aux0;
  const float* aux1;
  float* out;
  float* out2;
  int N, Hin, Win, C0, C1, mode0, mode1;
  int Cout, Hout, Wout, pad;
  int bwl, wxl, tiles_x, n_tiles, n_cout_tiles;
  int IH, IW, row_pitch, par_off, plane;
  int ck, n_chunks;
  int act, hid, out_split;
  void* out_bf;   // optional BF16_C8 copy of `out`
  int fmt0, fmt1;  // ESS_FMT_* of the sources
  int fmt_out;     // ESS_FMT_BF16_C8: `out` / `out2` ARE BF16_C8 tensors (LINEAR epilogue), nothing is written in fp32
  int fmt_res;     // format of `residual`
  int out_f16;     // BF16_C8-output epilogue: store IEEE half instead of bfloat16 (ESS_FMT_F16_C8: a pre-norm tensor, read by the norm kernels only)
  int persist;     // 3x3 wave-specialised kernel: the grid is one resident set of workgroups, each walking several tiles
  int slab;        // wide-tile kernel: output rows of one packed weight slab (the plan's cout_tile: 32, 64 or 128)
  int split;       // split-operand bf16 (ESS_COMPUTE_BF16X3): every 16-channel chunk is contracted three times -- (w_hi, x_hi), (w_hi, x_lo), (w_lo, x_hi)
  int deep;        // ablation bits of -DESS_ABLATE builds (always 0 in the shipped library: the kernels do not test it)
  int f16;         // ESS_COMPUTE_F16: the 16-bit operands (C8 sources, packed weights, C8 residual) and every 16-bit output / copy are IEEE half; the launchers pick the kernels' H = true instantiations
  int hilo;        // (f16) the 16-bit output / copy leaves as [hi | lo]: blocks [0, CB) = half(v), blocks [CB, 2 CB) = half(v - hi)  (ESS_FMT_F16_C8_HILO / ESS_LSTM_H_HILO)
};

// 4 floats -> four 16-bit elements: bfloat16 (round to nearest even) or, H, IEEE half (saturating: ess_f16_sat below)
template <bool H> __device__ __forceinline__ uint2 ess_cvt4(float v0, float v1, float v2, float v3);
template <bool H> __device__ __forceinline__ void ess_up4(uint2 r, float (&f)[4]);


// ---- shared epilogue: accumulator block (MFMA 32x32 C/D layout: col = lane&31 = pixel, row = (r&3)+8(r>>2)+4(lane>>5)
// = output channel) -> fused affine / residual / activation / LSTM / GRU maths -> NCHW fp32 stores
// All epilogue traffic goes through bounds-checked buffer instructions: descriptor base = this sample's tensor (SGPRs),
// voffset = pixel (+ the 4-channel step of the upper half-wave) kept in ONE register per pixel block, soffset = the
// wave-uniform channel plane.  No 64-bit per-element addresses (the first version needed ~2 VGPRs per store), no bounds
// branches (an out-of-tile pixel / out-of-range channel gets an offset past the descriptor: its load returns 0 and
// its store is dropped), and every read of a block is issued before the block's first store.
#ifndef ESS_EPI_STAMP
#define ESS_EPI_STAMP(i_) do { } while (0)  // (cycle-stamp builds of conv_bf16_ws.hip define it)
#endif
typedef __amdgpu_buffer_rsrc_t ess_rsrc;
constexpr unsigned ESS_OOB = 0x80000000u;
__device__ __forceinline__ ess_rsrc ess_make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(unsigned)bytes, 0x00020000);
}
// fp32 -> IEEE half for the F16_C8 pre-norm tensors, SATURATING: a plain cast turns |v| > 65504 into +-inf, the norm's statistics
// into NaN and with them the whole channel (BF16_C8 and the reference's fp32 have ~3e38 of range); one v_med3_f32 per element
// (v_med3_f32 acts as min3 on a NaN operand: a NaN accumulator would be stored as -65504 and a diverged run would yield finite
// statistics; NaN is kept, so loss-is-NaN detection still fires: test_pre_norm_f16_saturates)
__device__ __forceinline__ _Float16 ess_f16_sat(float v) {
  const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  return (_Float16)(v != v ? v : c);
}
template <> __device__ __forceinline__ uint2 ess_cvt4<false>(float v0, float v1, float v2, float v3) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 b;
  b[0] = (__bf16)v0; b[1] = (__bf16)v1; b[2] = (__bf16)v2; b[3] = (__bf16)v3;
  return __builtin_bit_cast(uint2, b);
}
template <> __device__ __forceinline__ uint2 ess_cvt4<true>(float v0, float v1, float v2, float v3) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 h;
  h[0] = ess_f16_sat(v0); h[1] = ess_f16_sat(v1); h[2] = ess_f16_sat(v2); h[3] = ess_f16_sat(v3);
  return __builtin_bit_cast(uint2, h);
}
// the lo parts of a [hi | lo] half pair: half(v - float(half(v)))  (|v| <= 65504: exact difference, ~22 significant bits in the pair)
__device__ __forceinline__ uint2 ess_cvt4_lo(float v0, float v1, float v2, float v3) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 h;
  h[0] = (_Float16)(v0 - (float)ess_f16_sat(v0)); h[1] = (_Float16)(v1 - (float)ess_f16_sat(v1));
  h[2] = (_Float16)(v2 - (float)ess_f16_sat(v2)); h[3] = (_Float16)(v3 - (float)ess_f16_sat(v3));
  return __builtin_bit_cast(uint2, h);
}
template <> __device__ __forceinline__ void ess_up4<false>(uint2 r, float (&f)[4]) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const bf16x4 b = __builtin_bit_cast(bf16x4, r);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = (float)b[i];
}
template <> __device__ __forceinline__ void ess_up4<true>(uint2 r, float (&f)[4]) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const f16x4 b = __builtin_bit_cast(f16x4, r);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = (float)b[i];
}
__device__ __forceinline__ float ess_bload(ess_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void ess_bstore(float v, ess_rsrc r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

// linear / GRU-candidate epilogue, specialised on what exists (per-row scale, a second input) so that the common
// bias-only case keeps no dead registers
// BF16_C8 copy of an output: this lane's 4 consecutive channels (4*half .. 4*half+3 of 8-channel block `blk`) of pixel
// `pix` are 8 bytes; the two half-waves interleave to full 16-byte pixel vectors, 32 pixels = 512 contiguous bytes.
template <bool H = false>
__device__ __forceinline__ void ess_store_bf16x4(void* base, size_t sample_blk0, int blk, unsigned HW, int pix, int half,
                                                 float v0, float v1, float v2, float v3) {
  *(uint2*)((char*)base + ((sample_blk0 + blk) * HW + pix) * 16 + 8 * half) = ess_cvt4<H>(v0, v1, v2, v3);
}

// FIX pins the wave-uniform run-time options at compile time for the combinations the train step launches most (each
// one is a branch inside 16 x MB x NBW unrolled rows otherwise -- code size and registers):
//   0: nothing pinned;  1: no activation, no BF16_C8 copy, one fp32 output;  2: ReLU, one output (copy / fp32 optional);
//   3: no activation, no copy, split output (the data-gradient of a concat convolution)
template <int MB, int EPI, bool SC, bool IN, int FIX = 0, bool H = false>
__device__ __forceinline__ void conv_epilogue_rows(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half,
                                                   const unsigned (&voff)[NBW], const int (&pixi)[NBW], unsigned plane_b) {
  constexpr int COT = MB * 32;
  const unsigned HW = plane_b / 4u;
  const int c_out = EPI == ESS_EPI_LINEAR ? a.Cout : a.hid;
  const int split = (EPI == ESS_EPI_LINEAR && (FIX == 0 || FIX == 3)) ? a.out_split : 0;
  const int act = FIX == 1 || FIX == 3 ? (int)ESS_ACT_NONE : FIX == 2 ? (int)ESS_ACT_RELU : a.act;
  const bool has_bf = (FIX == 1 || FIX == 3) ? false : a.out_bf != nullptr;
  const bool has_out = FIX == 1 ? true : a.out != nullptr;
  const int c_first = split > 0 ? split : c_out;  // channels of the first output tensor
  const int nblk = (c_out + 7) >> 3;              // 8-channel blocks of the BF16_C8 copy
  const ess_rsrc r_out = ess_make_rsrc(a.out + (size_t)n * c_first * HW, (size_t)c_first * plane_b);
  const ess_rsrc r_out2 =
      ess_make_rsrc(split > 0 ? a.out2 + (size_t)n * (c_out - split) * HW : a.out, (size_t)(c_out - split) * plane_b);
  const ess_rsrc r_sc = ess_make_rsrc(SC ? a.scale : a.out, SC ? (size_t)c_out * 4 : 0);
  const ess_rsrc r_sh = ess_make_rsrc(a.shift ? a.shift : a.out, a.shift ? (size_t)c_out * 4 : 0);
  static_assert(EPI == ESS_EPI_LINEAR, "the recurrent epilogues have their own functions");
  const ess_rsrc r_in0 = ess_make_rsrc(IN ? a.residual + (size_t)n * c_out * HW : a.out, IN ? (size_t)c_out * plane_b : 0);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
    float sc[SC ? 16 : 1], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // soffset is outside the hardware range check: the channel bound is folded into voffset
      const int cu = rowbase + (r & 3) + 8 * (r >> 2);
      const unsigned vo = cu + 4 * half < c_out ? 16u * half : ESS_OOB;
      if constexpr (SC) sc[r] = ess_bload(r_sc, vo, (unsigned)cu * 4u);
      sh[r] = ess_bload(r_sh, vo, (unsigned)cu * 4u);
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      float in0v[IN ? 16 : 1];
      if constexpr (IN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = rowbase + (r & 3) + 8 * (r >> 2);
          const unsigned vo = cu + 4 * half < c_out ? voff[nb] : ESS_OOB;
          in0v[r] = ess_bload(r_in0, vo, (unsigned)cu * plane_b);
        }
      }
      float q[4];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = rowbase + (r & 3) + 8 * (r >> 2);  // + 4*half = output channel
        float v = acc[mb][nb][r];
        if constexpr (SC) v *= sc[r];
        v += sh[r];
        if constexpr (IN) v += in0v[r];
        if (act == ESS_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ESS_ACT_SIGMOID) v = ess_sigmoid(v);
        else if (act == ESS_ACT_TANH) v = ess_tanh(v);
        const int co = cu + 4 * half;
        if (has_bf) {  // wave-uniform
          q[r & 3] = co < c_out ? v : 0.f;  // tail channels of the last block are zero
          if ((r & 3) == 3 && pixi[nb] >= 0 && (rowbase >> 3) + (r >> 2) < nblk)
            ess_store_bf16x4<H>(a.out_bf, (size_t)n * nblk, (rowbase >> 3) + (r >> 2), HW, pixi[nb], half, q[0], q[1], q[2], q[3]);
        }
        if (FIX == 3 || split > 0) {  // the two halves of a wave may straddle the split: tensor and channel are chosen per lane
          const unsigned pixo = voff[nb] == ESS_OOB ? ESS_OOB : voff[nb] - 4u * half * plane_b;
          if (co < split) ess_bstore(v, r_out, pixo + (unsigned)co * plane_b, 0);
          else if (co < c_out) ess_bstore(v, r_out2, pixo + (unsigned)(co - split) * plane_b, 0);
        } else if (has_out) {  // (NULL: the caller only wants the BF16_C8 copy)
          ess_bstore(v, r_out, co < c_out ? voff[nb] : ESS_OOB, (unsigned)cu * plane_b);
        }
      }
    }
  }
}

// The commonest LINEAR epilogue -- y = acc + bias into one fp32 NCHW tensor, nothing else -- without the run-time options
// of conv_epilogue_rows (activation, residual, scale, split output, BF16_C8 copy).  Those are wave-uniform branches, but
// inside 16 x MB x NBW unrolled rows they cost registers: the general LINEAR kernel spills ~190 VGPRs in its epilogue.
template <int MB>
__device__ __forceinline__ void conv_epilogue_plain(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half,
                                                    const unsigned (&voff)[NBW], unsigned plane_b) {
  constexpr int COT = MB * 32;
  const unsigned HW = plane_b / 4u;
  const int c_out = a.Cout;
  const ess_rsrc r_out = ess_make_rsrc(a.out + (size_t)n * c_out * HW, (size_t)c_out * plane_b);
  const ess_rsrc r_sh = ess_make_rsrc(a.shift ? a.shift : a.out, a.shift ? (size_t)c_out * 4 : 0);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
    float sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = rowbase + (r & 3) + 8 * (r >> 2);
      sh[r] = a.shift ? ess_bload(r_sh, cu + 4 * half < c_out ? 16u * half : ESS_OOB, (unsigned)cu * 4u) : 0.f;  // (uniform)
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = rowbase + (r & 3) + 8 * (r >> 2);
        ess_bstore(acc[mb][nb][r] + sh[r], r_out, cu + 4 * half < c_out ? voff[nb] : ESS_OOB, (unsigned)cu * plane_b);
      }
  }
}

// ESS_ACT_SUMPOOL2: the first output (channels < out_split, or all of them) leaves as the 2x2 SUM of its pixels at half the
// resolution -- the data-gradient of a nearest-x2-upsampled source, without the full-resolution tensor in between.  A lane
// owns one pixel of a 32-pixel block (BW wide, RB = 32/BW rows): the horizontal partner is lane^1, the vertical one lane^BW
// (RB >= 2) or the same lane's other pixel block (RB == 1: blocks nb = 0/1 are rows 2k / 2k+1).  Shuffles are executed by
// every lane; which lanes store is decided afterwards.  Channels >= out_split go to out2 at full resolution.
template <int MB>
__device__ __forceinline__ void conv_epilogue_pool(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half, int x,
                                                   int y0, const int (&ly)[NBW], unsigned plane_b) {
  static_assert(NBW == 2, "the RB == 1 pairing assumes two pixel blocks per wave");
  constexpr int COT = MB * 32;
  const unsigned HW = plane_b / 4u;
  const int Wl = a.Wout >> 1, Hl = a.Hout >> 1;
  const unsigned plane_l = (unsigned)(Hl * Wl) * 4u;
  const int c_out = a.Cout, split = a.out_split;
  const int c_first = split > 0 ? split : c_out;
  const ess_rsrc r_out = ess_make_rsrc(a.out + (size_t)n * c_first * (plane_l / 4u), (size_t)c_first * plane_l);
  const ess_rsrc r_out2 =
      ess_make_rsrc(split > 0 ? a.out2 + (size_t)n * (c_out - split) * HW : a.out, (size_t)(split > 0 ? c_out - split : 0) * plane_b);
  const ess_rsrc r_sh = ess_make_rsrc(a.shift ? a.shift : a.out, a.shift ? (size_t)c_out * 4 : 0);
  const int BW = 1 << a.bwl;
  const bool rows1 = a.bwl == 5;  // RB == 1
  const bool xin = x < a.Wout, xeven = (x & 1) == 0;
  unsigned pix_h[NBW], pix_l[NBW];  // byte offsets of this lane's pixel (full res) / pooled pixel inside one channel plane
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int y = y0 + ly[nb];
    const bool inb = xin && y < a.Hout;
    pix_h[nb] = inb ? (unsigned)(y * a.Wout + x) * 4u : ESS_OOB;
    const bool st = inb && xeven && (y & 1) == 0 && (!rows1 || nb == 0);
    pix_l[nb] = st ? (unsigned)((y >> 1) * Wl + (x >> 1)) * 4u : ESS_OOB;
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = rowbase + (r & 3) + 8 * (r >> 2);
      const int co = cu + 4 * half;
      const float sh = a.shift ? ess_bload(r_sh, co < c_out ? 16u * half : ESS_OOB, (unsigned)cu * 4u) : 0.f;  // (uniform)
      float v[NBW], t[NBW];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) v[nb] = acc[mb][nb][r] + sh;
      // lane^1 through DPP (quad_perm [1,0,3,2]): a register move, not an LDS-crossbar permute
      auto xor1 = [](float f) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xf, 0xf, true));
      };
      if (rows1) {  // (choose_geom picks 32-wide blocks for pooled launches whenever the image allows)
        t[0] = v[0] + v[1];
        t[0] += xor1(t[0]);
        t[1] = 0.f;
      } else {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          t[nb] = v[nb] + __shfl_xor(v[nb], BW, 64);
          t[nb] += xor1(t[nb]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        if (co < c_first) {
          if (nb == 0 || !rows1)  // (uniform: with one-row blocks the pooled pixel lives in block 0)
            ess_bstore(t[nb], r_out, pix_l[nb] == ESS_OOB ? ESS_OOB : pix_l[nb] + (unsigned)co * plane_l, 0);
        } else if (co < c_out) {
          ess_bstore(v[nb], r_out2, pix_h[nb] == ESS_OOB ? ESS_OOB : pix_h[nb] + (unsigned)(co - split) * plane_b, 0);
        }
      }
    }
  }
}

// ---- ConvGRU epilogues (reference e2vid/model/submodules.py:255-273).  The state-like fp32 tensors -- h_prev (aux0), u (out of the
// first kernel, aux1 of the second), r*h (out2), h' (out) -- are `hid` channel planes (ESS_FMT_F32_NCHW) or channel-blocked
// (ESS_FMT_F32_C8: [N][hid/8][H][W][8] fp32, this lane's 4 channels of a pixel are ONE 16-byte access where the planes take four
// 4-byte ones into four planes).  fmt_res describes the inputs (aux0, aux1), fmt_out the fp32 outputs (out, out2).  out_bf is a
// BF16_C8 tensor: r*h in the first kernel (what the candidate convolution stages -- it would round the fp32 tensor to exactly these
// values), a copy of h' in the second (what the next time step's convolutions stage).
struct EssStateIO {
  ess_rsrc r;
  bool c8;
  unsigned HW;
  int hid, nbh;
};
__device__ __forceinline__ EssStateIO ess_state_io(const float* base, const void* dummy, int n, int hid, unsigned HW, bool c8) {
  EssStateIO s;
  s.c8 = c8; s.HW = HW; s.hid = hid; s.nbh = (hid + 7) >> 3;
  const size_t per = c8 ? (size_t)s.nbh * 8 * HW : (size_t)hid * HW;  // floats of one sample
  s.r = ess_make_rsrc(base ? (const void*)(base + (size_t)n * per) : dummy, base ? per * 4 : 0);  // absent tensor: reads as zeros
  return s;
}
// this lane's 4 channels (4*half .. +3) of 8-channel block hb at its pixel; voff = byte offset of (channel 4*half, pixel) in planes
__device__ __forceinline__ void ess_state_load4(const EssStateIO& s, int hb, int pix, int half, unsigned voff, float (&v)[4]) {
  if (s.c8) {  // (uniform)
    typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
    const unsigned o = (pix >= 0 && hb < s.nbh) ? ((unsigned)hb * s.HW + (unsigned)pix) * 32u + 16u * half : ESS_OOB;
    const u32x4c t = __builtin_amdgcn_raw_buffer_load_b128(s.r, (int)o, 0, ESS_GRU_AUX);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) v[jj] = __builtin_bit_cast(float, (unsigned)t[jj]);
  } else {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)  // soffset is outside the hardware range check: the channel bound is folded into voffset
      v[jj] = ess_bload(s.r, hb * 8 + 4 * half + jj < s.hid ? voff : ESS_OOB, (unsigned)(hb * 8 + jj) * (s.HW * 4u));
  }
}
__device__ __forceinline__ void ess_state_store4(const EssStateIO& s, int hb, int pix, int half, unsigned voff, const float (&v)[4]) {
  if (s.c8) {
    typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
    const unsigned o = (pix >= 0 && hb < s.nbh) ? ((unsigned)hb * s.HW + (unsigned)pix) * 32u + 16u * half : ESS_OOB;
    u32x4c t;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) t[jj] = hb * 8 + 4 * half + jj < s.hid ? __builtin_bit_cast(unsigned, v[jj]) : 0u;  // tail channels: zeros
    __builtin_amdgcn_raw_buffer_store_b128(t, s.r, (int)o, 0, ESS_GRU_AUX);
  } else {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      ess_bstore(v[jj], s.r, hb * 8 + 4 * half + jj < s.hid ? voff : ESS_OOB, (unsigned)(hb * 8 + jj) * (s.HW * 4u));
  }
}

// (update, reset) gates: packed row 8*q + j of a 32-row block = gate q&1 (0 update, 1 reset) of hidden hb16*16 + (q>>1)*8 + j, so
// accumulator register 8*q2 + jj is the update and 8*q2 + 4 + jj the reset gate of hidden (ct*MB+mb)*16 + q2*8 + 4*half + jj.
// u = sigmoid(.) -> out;  (sigmoid(.) * h_prev) -> out2 (fp32) and / or out_bf (BF16_C8).  h_prev NULL reads as zeros; with
// neither out2 nor out_bf the reset gate is not evaluated (first time step of a sequence: r*h = 0 whatever r is).
template <int MB, bool H = false>
__device__ __forceinline__ void conv_epilogue_gru_ur(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half,
                                                     const unsigned (&voff)[NBW], const int (&pixi)[NBW], unsigned HW, bool biased) {
  constexpr int COT = MB * 32;
  const bool in8 = a.fmt_res == ESS_FMT_F32_C8, out8 = a.fmt_out == ESS_FMT_F32_C8;
  const EssStateIO h_io = ess_state_io(a.aux0, a.wpk, n, a.hid, HW, in8);
  const EssStateIO u_io = ess_state_io(a.out, a.wpk, n, a.hid, HW, out8);
  const EssStateIO rh_io = ess_state_io(a.out2, a.wpk, n, a.hid, HW, out8);
  const ess_rsrc r_sh = ess_make_rsrc(a.shift, (size_t)(a.n_cout_tiles * COT) * 4);
  const bool need_r = a.out2 != nullptr || a.out_bf != nullptr;
  const int nbh = (a.hid + 7) >> 3;
  float hp[MB][NBW][2][4];
  if (need_r) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) ess_state_load4(h_io, (ct * MB + mb) * 2 + q2, pixi[nb], half, voff[nb], hp[mb][nb][q2]);
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
    float sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sh[r] = biased ? 0.f : ess_bload(r_sh, 16u * half, (unsigned)(rowbase + (r & 3) + 8 * (r >> 2)) * 4u);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int hb = (ct * MB + mb) * 2 + q2;
        float u[4], rh[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          u[jj] = ess_sigmoid(acc[mb][nb][8 * q2 + jj] + sh[8 * q2 + jj]);
          if (a.act == ESS_GRU_U_F16) u[jj] = (float)(_Float16)u[jj];  // (uniform) the value the F16_C8 form of u carries, in an fp32 tensor
        }
        ess_state_store4(u_io, hb, pixi[nb], half, voff[nb], u);
        if (need_r) {  // (uniform)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            rh[jj] = ess_sigmoid(acc[mb][nb][8 * q2 + 4 + jj] + sh[8 * q2 + 4 + jj]) * hp[mb][nb][q2][jj];
            if (hb * 8 + 4 * half + jj >= a.hid) rh[jj] = 0.f;
          }
          if (a.out2) ess_state_store4(rh_io, hb, pixi[nb], half, voff[nb], rh);
          if (a.out_bf && pixi[nb] >= 0 && hb < nbh)
            ess_store_bf16x4<H>(a.out_bf, (size_t)n * nbh, hb, HW, pixi[nb], half, rh[0], rh[1], rh[2], rh[3]);
        }
      }
    }
  }
}

// candidate: packed row = hidden channel, so accumulator register 4*j + jj belongs to hidden (ct*MB+mb)*32 + 8*j + 4*half + jj.
// h' = h_prev (1 - u) + tanh(.) u -> out (fp32, may be NULL when only the copy is wanted) and / or out_bf (BF16_C8 copy).
template <int MB, bool H = false>
__device__ __forceinline__ void conv_epilogue_gru_out(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half,
                                                      const unsigned (&voff)[NBW], const int (&pixi)[NBW], unsigned HW, bool biased) {
  constexpr int COT = MB * 32;
  const bool in8 = a.fmt_res == ESS_FMT_F32_C8, out8 = a.fmt_out == ESS_FMT_F32_C8;
  const EssStateIO h_io = ess_state_io(a.aux0, a.wpk, n, a.hid, HW, in8);
  const EssStateIO u_io = ess_state_io(a.aux1, a.wpk, n, a.hid, HW, in8);
  const EssStateIO o_io = ess_state_io(a.out, a.wpk, n, a.hid, HW, out8);
  const ess_rsrc r_sh = ess_make_rsrc(a.shift, (size_t)(a.n_cout_tiles * COT) * 4);
  const int nbh = (a.hid + 7) >> 3;
  const bool u16 = a.act == ESS_GRU_U_F16 && in8;
  const ess_rsrc r_u16 = ess_make_rsrc(u16 ? (const char*)a.aux1 + (size_t)n * nbh * HW * 16 : (const char*)a.wpk, u16 ? (size_t)nbh * HW * 16 : 0);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
    asm volatile("" ::: "memory");  // one 32-channel block of state reads in flight at a time (registers)
    float sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sh[r] = biased ? 0.f : ess_bload(r_sh, 16u * half, (unsigned)(rowbase + (r & 3) + 8 * (r >> 2)) * 4u);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      float hv[4][4], uv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ess_state_load4(h_io, (ct * MB + mb) * 4 + j, pixi[nb], half, voff[nb], hv[j]);
        if (u16) {  // (uniform) u is an F16_C8 tensor (ESS_GRU_U_F16 with channel-blocked inputs): this lane's 4 channels = 8 bytes
          typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));
          typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
          const int hb = (ct * MB + mb) * 4 + j;
          const unsigned ou = (pixi[nb] >= 0 && hb < nbh) ? ((unsigned)hb * HW + (unsigned)pixi[nb]) * 16u + 8u * half : ESS_OOB;
          const f16x4e uh = __builtin_bit_cast(f16x4e, (u32x2c)__builtin_amdgcn_raw_buffer_load_b64(r_u16, (int)ou, 0, ESS_GRU_AUX));
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) uv[j][jj] = (float)uh[jj];
        } else {
          ess_state_load4(u_io, (ct * MB + mb) * 4 + j, pixi[nb], half, voff[nb], uv[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hb = (ct * MB + mb) * 4 + j;
        float hn[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float o = ess_tanh(acc[mb][nb][4 * j + jj] + sh[4 * j + jj]);
          hn[jj] = hv[j][jj] * (1.f - uv[j][jj]) + o * uv[j][jj];
          if (hb * 8 + 4 * half + jj >= a.hid) hn[jj] = 0.f;
        }
        if (a.out) ess_state_store4(o_io, hb, pixi[nb], half, voff[nb], hn);
        if (a.out_bf && pixi[nb] >= 0 && hb < nbh)
          ess_store_bf16x4<H>(a.out_bf, (size_t)n * nbh, hb, HW, pixi[nb], half, hn[0], hn[1], hn[2], hn[3]);
      }
    }
  }
}

// ---- the two ConvGRU epilogues of the lean time steps as straight-line code (what conv_epilogue_lstm_c8 is to the LSTM one):
// F32_C8 states in and out, bias in the accumulators, every hidden channel of the tile real; state loads issued first, 16-byte
// stores, the BF16_C8 tensors (r*h, the copy of h') as whole pixel vectors via a half-wave swap between two hidden blocks.
template <int MB, int NB, bool H = false>
__device__ __forceinline__ void conv_epilogue_gru_ur_c8(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int n, int half,
                                                        const int (&pixi)[NB], unsigned HW) {
  typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
  const int nbh = a.hid >> 3;
  const size_t state_b = (size_t)nbh * HW * 32;
  const bool need_r = a.out_bf != nullptr;
  const bool u16 = a.act == ESS_GRU_U_F16;  // (uniform) u leaves as an F16_C8 tensor: 16 instead of 32 bytes per pixel and 8-channel block
  const ess_rsrc r_h = ess_make_rsrc((a.aux0 && need_r) ? (const char*)(a.aux0 + (size_t)n * nbh * 8 * HW) : (const char*)a.out, (a.aux0 && need_r) ? state_b : 0);
  const ess_rsrc r_u = u16 ? ess_make_rsrc((const char*)a.out + (size_t)n * nbh * HW * 16, (size_t)nbh * HW * 16)
                           : ess_make_rsrc(a.out + (size_t)n * nbh * 8 * HW, state_b);
  const ess_rsrc r_rb = ess_make_rsrc(need_r ? (const char*)a.out_bf + (size_t)n * nbh * HW * 16 : (const char*)a.out, need_r ? (size_t)nbh * HW * 16 : 0);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      unsigned vo[NB][2];  // (indexed [nb] for the code below; one pixel block's vectors in flight at a time: 128 registers)
      u32x4c hp[NB][2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int hb = (ct * MB + mb) * 2 + q2;
        vo[nb][q2] = pixi[nb] >= 0 ? ((unsigned)hb * HW + (unsigned)pixi[nb]) * 32u + 16u * half : ESS_OOB;
        hp[nb][q2] = __builtin_amdgcn_raw_buffer_load_b128(r_h, (int)vo[nb][q2], 0, ESS_GRU_AUX);  // (absent / unused: zeros)
      }
      __builtin_amdgcn_sched_barrier(0);  // (keeps hipcc from sinking the loads to their first use)
      uint2 pk[2], pu[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        u32x4c uv;
        f16x4e uh;
        float rf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float ug = ess_sigmoid(acc[mb][nb][8 * q2 + jj]);
          uv[jj] = __builtin_bit_cast(unsigned, ug);
          uh[jj] = (_Float16)ug;
          if (need_r) rf[jj] = ess_sigmoid(acc[mb][nb][8 * q2 + 4 + jj]) * __builtin_bit_cast(float, (unsigned)hp[nb][q2][jj]);
        }
        if (!u16) __builtin_amdgcn_raw_buffer_store_b128(uv, r_u, (int)vo[nb][q2], 0, ESS_GRU_AUX);
        pk[q2] = ess_cvt4<H>(rf[0], rf[1], rf[2], rf[3]);
        pu[q2] = __builtin_bit_cast(uint2, uh);
      }
      if (u16) {  // (uniform) whole pixel vectors, as r*h below: lanes 0-31 hidden block 2 (ct MB + mb), lanes 32-63 the next one
        const auto s0 = __builtin_amdgcn_permlane32_swap(pu[0].x, pu[1].x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pu[0].y, pu[1].y, false, false);
        const u32x4c vec = {s0[0], s1[0], s0[1], s1[1]};
        const unsigned o = pixi[nb] >= 0 ? ((unsigned)((ct * MB + mb) * 2 + half) * HW + (unsigned)pixi[nb]) * 16u : ESS_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(vec, r_u, (int)o, 0, ESS_GRU_AUX);
      }
      if (need_r) {  // (uniform) lanes 0-31: hidden block 2 (ct MB + mb), lanes 32-63: the next one
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        const u32x4c vec = {s0[0], s1[0], s0[1], s1[1]};
        const unsigned o = pixi[nb] >= 0 ? ((unsigned)((ct * MB + mb) * 2 + half) * HW + (unsigned)pixi[nb]) * 16u : ESS_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(vec, r_rb, (int)o, 0, ESS_GRU_AUX);
      }
    }
  }
}

template <int MB, int NB, bool H = false>
__device__ __forceinline__ void conv_epilogue_gru_out_c8(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int n, int half,
                                                         const int (&pixi)[NB], unsigned HW) {
  typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const int nbh = a.hid >> 3;
  const size_t state_b = (size_t)nbh * HW * 32;
  const ess_rsrc r_h = ess_make_rsrc(a.aux0 ? (const char*)(a.aux0 + (size_t)n * nbh * 8 * HW) : (const char*)a.aux1, a.aux0 ? state_b : 0);
  const bool u16 = a.act == ESS_GRU_U_F16;  // (uniform) u arrives as an F16_C8 tensor
  const ess_rsrc r_u = u16 ? ess_make_rsrc((const char*)a.aux1 + (size_t)n * nbh * HW * 16, (size_t)nbh * HW * 16)
                           : ess_make_rsrc(a.aux1 + (size_t)n * nbh * 8 * HW, state_b);
  const ess_rsrc r_o = ess_make_rsrc(a.out ? (const char*)(a.out + (size_t)n * nbh * 8 * HW) : (const char*)a.aux1, a.out ? state_b : 0);
  const int ncp = (H && a.hilo) ? 2 : 1;  // (uniform) [hi | lo] copy of h': 2 nbh blocks per sample (ESS_GRU_H_HILO)
  const ess_rsrc r_ob = ess_make_rsrc(a.out_bf ? (const char*)a.out_bf + (size_t)n * ncp * nbh * HW * 16 : (const char*)a.aux1, a.out_bf ? (size_t)ncp * nbh * HW * 16 : 0);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      unsigned vo[4];
      u32x4c hv[4], uv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hb = (ct * MB + mb) * 4 + j;
        vo[j] = pixi[nb] >= 0 ? ((unsigned)hb * HW + (unsigned)pixi[nb]) * 32u + 16u * half : ESS_OOB;
        hv[j] = __builtin_amdgcn_raw_buffer_load_b128(r_h, (int)vo[j], 0, ESS_GRU_AUX);
        if (u16) {
          // whole 16-byte pixel vectors, the mirror image of the (update, reset) kernel's store: lanes 0-31 fetch hidden block j, lanes
          // 32-63 block j + 1 (j even), and a half-wave swap behind the barrier hands every lane its 4 channels of both blocks.
          // (8-byte loads of the lane's own 4 channels -- two half-waves interleaved at 8-byte granularity -- measured + 10 us per
          // launch over the fp32 form they were meant to beat; a conversion right here waits for each load in turn: + 15 us.)
          if ((j & 1) == 0) {
            const unsigned ou = pixi[nb] >= 0 ? ((unsigned)(hb + half) * HW + (unsigned)pixi[nb]) * 16u : ESS_OOB;
            uv[j] = __builtin_amdgcn_raw_buffer_load_b128(r_u, (int)ou, 0, ESS_GRU_AUX);
          }
        } else {
          uv[j] = __builtin_amdgcn_raw_buffer_load_b128(r_u, (int)vo[j], 0, ESS_GRU_AUX);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (u16) {  // (uniform) swap, then 4 halfs in two registers -> 4 fp32 bit patterns
        typedef _Float16 f16x2e __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)uv[j][0], (unsigned)uv[j][2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)uv[j][1], (unsigned)uv[j][3], false, false);
          uv[j][0] = s0[0]; uv[j][1] = s1[0];          // block j:     this lane's channels 4 half .. + 3
          uv[j + 1][0] = s0[1]; uv[j + 1][1] = s1[1];  // block j + 1
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f16x2e lo = __builtin_bit_cast(f16x2e, (unsigned)uv[j][0]), hi = __builtin_bit_cast(f16x2e, (unsigned)uv[j][1]);
          uv[j][0] = __builtin_bit_cast(unsigned, (float)lo[0]); uv[j][1] = __builtin_bit_cast(unsigned, (float)lo[1]);
          uv[j][2] = __builtin_bit_cast(unsigned, (float)hi[0]); uv[j][3] = __builtin_bit_cast(unsigned, (float)hi[1]);
        }
      }
      uint2 pk[4], pl[H ? 4 : 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4c ov;
        float of[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float o = ess_tanh(acc[mb][nb][4 * j + jj]);
          const float hprev = __builtin_bit_cast(float, (unsigned)hv[j][jj]), u = __builtin_bit_cast(float, (unsigned)uv[j][jj]);
          const float hn = hprev * (1.f - u) + o * u;
          ov[jj] = __builtin_bit_cast(unsigned, hn);
          of[jj] = hn;
        }
        if (a.out) __builtin_amdgcn_raw_buffer_store_b128(ov, r_o, (int)vo[j], 0, ESS_GRU_AUX);  // (uniform)
        pk[j] = ess_cvt4<H>(of[0], of[1], of[2], of[3]);
        if constexpr (H) pl[j] = ess_cvt4_lo(of[0], of[1], of[2], of[3]);
      }
      if (a.out_bf) {  // (uniform)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[j].x, pk[j + 1].x, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[j].y, pk[j + 1].y, false, false);
          const u32x4c vec = {s0[0], s1[0], s0[1], s1[1]};
          const unsigned o = pixi[nb] >= 0 ? ((unsigned)((ct * MB + mb) * 4 + j + half) * HW + (unsigned)pixi[nb]) * 16u : ESS_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(vec, r_ob, (int)o, 0, ESS_GRU_AUX);
          if constexpr (H) {
            if (ncp == 2) {  // (uniform) the lo parts, nbh blocks further on
              const auto t0 = __builtin_amdgcn_permlane32_swap(pl[j].x, pl[j + 1].x, false, false);
              const auto t1 = __builtin_amdgcn_permlane32_swap(pl[j].y, pl[j + 1].y, false, false);
              const u32x4c vlo = {t0[0], t1[0], t0[1], t1[1]};
              __builtin_amdgcn_raw_buffer_store_b128(vlo, r_ob, (int)(o == ESS_OOB ? ESS_OOB : o + (unsigned)nbh * HW * 16u), 0, ESS_GRU_AUX);
            }
          }
        }
      }
    }
}

// Accumulators that START from the per-channel shift (bias): accumulator register r of 32-row block mb holds packed row
// (r & 3) + 8 (r >> 2) + 4 half, so four float4 loads per block fill both pixel blocks.  The loads ride in the matrix waves' wait
// for the first staged chunk; the epilogue then has no shift to fetch (in the LSTM epilogue that was one dependent round trip
// per 32-row block in front of the gate arithmetic).
template <int MB, int NB>
__device__ __forceinline__ void conv_bias_init(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int half) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *(const float4*)(a.shift + ct * (MB * 32) + mb * 32 + 8 * g + 4 * half);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) { acc[mb][nb][4 * g] = v.x; acc[mb][nb][4 * g + 1] = v.y; acc[mb][nb][4 * g + 2] = v.z; acc[mb][nb][4 * g + 3] = v.w; }
    }
}

// LINEAR epilogue with BF16_C8 OUTPUT(S) (fmt_out): the stored form of the trainable networks' activations and activation
// gradients in the bf16 configuration.  A lane owns 4 consecutive channels (4*half .. +3 of 8-channel block rowbase/8 + j) of
// its pixel.  Blocks are handled in PAIRS (j, j+1): after the arithmetic the two half-waves exchange halves with
// v_permlane32_swap so that lanes 0-31 hold the complete 16-byte pixel vector of block j and lanes 32-63 that of block j+1 --
// one 16-byte store per lane, 32 lanes = 512 contiguous bytes (the first version stored 8 bytes per lane with the two
// half-waves interleaved inside every 16-byte vector: twice the store instructions and half-filled write transactions).
// The BF16_C8 residual is read the same way (16 bytes per lane, halves exchanged back).
// Options (all wave-uniform): per-channel scale / shift, a BF16_C8 residual (the skip gradient riding on a data-gradient),
// ReLU, out_split (channels >= out_split go to out2: the data-gradient of a concat convolution; out_split % 8 == 0),
// SUMPOOL2 (the first output leaves as the 2x2 pixel sum at half resolution: gradient of a nearest-x2-upsampled source).
// Channels past C_out inside the last block are written as zeros (the BF16_C8 contract).
template <int MB, bool SC, bool SH, bool H = false>
__device__ __forceinline__ void conv_epilogue_c8_impl(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half, int x,
                                                      int y0, const int (&ly)[NBW]) {
  static_assert(NBW == 2, "the one-row pooled pairing assumes two pixel blocks per wave");
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
  constexpr int COT = MB * 32;
  // Register-lean on purpose (the matrix waves own 128 registers, 64 of them accumulators): buffer stores / loads with 32-bit
  // offsets (out-of-image lanes get an out-of-range offset instead of a predicate and a clamped 64-bit address), and the
  // scale / shift / residual vectors of ONE 32-channel block are loaded at a time (a compiler fence per block keeps hipcc from
  // hoisting all of them to the top, which spilled ~40 registers around every load: 14 k cycles per workgroup).
  const unsigned HW = (unsigned)(a.Hout * a.Wout);
  const int c_out = a.Cout, split = a.out_split;
  const int c_first = split > 0 ? split : c_out;
  const int nb_first = (c_first + 7) >> 3, nb_all = nb_first + (split > 0 ? (c_out - split + 7) >> 3 : 0);
  const bool pool = a.act == ESS_ACT_SUMPOOL2, relu = a.act == ESS_ACT_RELU;
  const int Wl = a.Wout >> 1;
  const unsigned HWl = (unsigned)((a.Hout >> 1) * Wl);
  const unsigned HW1 = pool ? HWl : HW;  // pixels per block plane of the first output
  const ess_rsrc r_o1 = ess_make_rsrc((const char*)a.out + (size_t)n * nb_first * HW1 * 16, (size_t)nb_first * HW1 * 16);
  const ess_rsrc r_o2 = ess_make_rsrc(split > 0 ? (const char*)a.out2 + (size_t)n * (nb_all - nb_first) * HW * 16 : (const char*)a.out,
                                      split > 0 ? (size_t)(nb_all - nb_first) * HW * 16 : 0);
  const ess_rsrc r_rs = ess_make_rsrc(a.residual ? (const char*)a.residual + (size_t)n * nb_all * HW * 16 : (const char*)a.out,
                                      a.residual ? (size_t)nb_all * HW * 16 : 0);
  const int BW = 1 << a.bwl;
  const bool rows1 = a.bwl == 5;
  unsigned pix16[NBW], st1_16[NBW];  // byte offset of this lane's pixel inside a block plane: full resolution / first output's
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int y = y0 + ly[nb];
    const bool inb = (y < a.Hout) & (x < a.Wout);
    const bool own = inb & !(x & 1) & !(y & 1) & (!rows1 | (nb == 0));
    pix16[nb] = inb ? (unsigned)(y * a.Wout + x) * 16u : ESS_OOB;
    st1_16[nb] = pool ? (own ? (unsigned)((y >> 1) * Wl + (x >> 1)) * 16u : ESS_OOB) : pix16[nb];
  }
  auto xor1 = [](float f) {  // lane^1 through DPP (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xf, 0xf, true));
  };
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
    asm volatile("" ::: "memory");
    // everything this 32-channel block needs from memory, issued together (one round trip per mb): per-channel scale / shift
    // of this lane's 4 channels in each of the four 8-channel blocks (the packed vectors are padded to the channel tile) and
    // the residual vectors of this lane's store blocks, one per pixel block of the wave (absent blocks / pixels read zeros)
    float4 scv[4], shv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c0 = rowbase + j * 8 + 4 * half;
      if constexpr (SC) scv[j] = *(const float4*)(a.scale + c0);
      if constexpr (SH) shv[j] = *(const float4*)(a.shift + c0);
    }
    u32x4e rva[2][NBW];
    if (a.residual) {  // (uniform)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          const int blk = (rowbase >> 3) + 2 * jh + half;
          rva[jh][nb] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, (int)((blk < nb_all && pix16[nb] != ESS_OOB) ? (unsigned)blk * HW * 16u + pix16[nb] : ESS_OOB), 0, 0);
        }
    }
#pragma unroll
    for (int jp = 0; jp < 4; jp += 2) {  // block pair (jp, jp + 1)
      const int blk0 = (rowbase >> 3) + jp;      // wave-uniform; this lane stores block blk0 + half
      const int myblk = blk0 + half;
      const u32x4e (&rv)[NBW] = rva[jp >> 1];
      uint2 pk[2][NBW];  // packed results [block of the pair][pixel block]: this lane's 4 channels
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = jp + jj, blk = blk0 + jj, c0 = blk * 8 + 4 * half;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SC) { sc[0] = scv[j].x; sc[1] = scv[j].y; sc[2] = scv[j].z; sc[3] = scv[j].w; }
        if constexpr (SH) { sh[0] = shv[j].x; sh[1] = shv[j].y; sh[2] = shv[j].z; sh[3] = shv[j].w; }
        float v[NBW][4];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[nb][i] = acc[mb][nb][4 * j + i] * sc[i] + sh[i];
          if (a.residual) {
            // lanes 0-31 loaded block blk0 (dwords 0,1 = channels 0-3; 2,3 = 4-7), lanes 32-63 block blk0 + 1:
            // swap(A = dwords 0,1 ; B = dwords 2,3) gives lanes 0-31 (A own, A partner) = channels 0-3 of (blk0, blk0+1)
            // and lanes 32-63 (B partner, B own) = channels 4-7 of (blk0, blk0+1)
            const auto s0 = __builtin_amdgcn_permlane32_swap(rv[nb][0], rv[nb][2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(rv[nb][1], rv[nb][3], false, false);
            const uint2 rr = jj == 0 ? make_uint2(s0[0], s1[0]) : make_uint2(s0[1], s1[1]);
            float rf[4];
            ess_up4<H>(rr, rf);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[nb][i] += rf[i];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (relu) v[nb][i] = fmaxf(v[nb][i], 0.f);
            if (c0 + i >= c_out) v[nb][i] = 0.f;
          }
        }
        if (pool && blk < nb_first) {  // (uniform) 2x2 pixel sums; the owner lanes (even x, even y) hold the result
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (rows1) {
              v[0][i] = v[0][i] + v[1][i];
              v[0][i] += xor1(v[0][i]);
            } else {
#pragma unroll
              for (int nb = 0; nb < NBW; ++nb) {
                v[nb][i] = v[nb][i] + __shfl_xor(v[nb][i], BW, 64);
                v[nb][i] += xor1(v[nb][i]);
              }
            }
          }
        }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          if (H || a.out_f16) {  // (uniform)
            pk[jj][nb] = ess_cvt4<true>(v[nb][0], v[nb][1], v[nb][2], v[nb][3]);
          } else {
            pk[jj][nb] = ess_cvt4<false>(v[nb][0], v[nb][1], v[nb][2], v[nb][3]);
          }
        }
      }
      // exchange halves: lanes 0-31 end up with the whole vector of block blk0, lanes 32-63 with that of blk0 + 1
      const bool to1 = myblk < nb_first;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][nb].x, pk[1][nb].x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][nb].y, pk[1][nb].y, false, false);
        const u32x4e vec = {s0[0], s1[0], s0[1], s1[1]};
        // first output (pooled or not) / second output (never pooled): a pair may straddle the split, so both stores exist,
        // each under a uniform condition, with the lanes of the other side pushed out of range
        if (blk0 < nb_first) {
          const unsigned o = (to1 && st1_16[nb] != ESS_OOB) ? (unsigned)myblk * HW1 * 16u + st1_16[nb] : ESS_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(vec, r_o1, (int)o, 0, 0);
        }
        if (split > 0 && blk0 + 1 >= nb_first) {
          const unsigned o = (!to1 && myblk < nb_all && pix16[nb] != ESS_OOB) ? (unsigned)(myblk - nb_first) * HW * 16u + pix16[nb] : ESS_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(vec, r_o2, (int)o, 0, 0);
        }
      }
    }
  }
}

// The common case of the function above -- one output, every channel of the tile real, no residual, no pooling -- as
// straight-line code: the general function spends 2.5-4 k cycles per 32-channel block in uniform branches, channel masks and
// selects (cycle stamps, round 3: 7.3-8.5 k cycles per 64 x 64 tile of a bias-only layer with nothing to load), this one is
// per block 64 optional v_max, 32 packed conversions, 8 half-wave swaps and four 16-byte stores.
template <int MB, bool SC, bool SH, bool F16, bool RELU, bool RES, int NB, bool H = false, bool HILO = false>
__device__ __forceinline__ void conv_epilogue_c8_plain(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int n, int half,
                                                       int x, int y0, const int (&ly)[NB]) {
  static_assert(!HILO || H, "a [hi | lo] output is a half tensor");
  typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
  constexpr int COT = MB * 32;
  const unsigned HW = (unsigned)(a.Hout * a.Wout);
  const int nblk = a.Cout >> 3;
  // (HILO: the output tensor holds 2 nblk blocks per sample -- hi in [0, nblk), lo in [nblk, 2 nblk))
  const ess_rsrc r_o = ess_make_rsrc((const char*)a.out + (size_t)n * (HILO ? 2 : 1) * nblk * HW * 16, (size_t)(HILO ? 2 : 1) * nblk * HW * 16);
  const ess_rsrc r_rs = ess_make_rsrc((const char*)(RES ? a.residual : a.out) + (size_t)n * nblk * HW * 16, RES ? (size_t)nblk * HW * 16 : 0);
  unsigned pix16[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int y = y0 + ly[nb];
    pix16[nb] = ((y < a.Hout) & (x < a.Wout)) ? (unsigned)(y * a.Wout + x) * 16u : ESS_OOB;
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
    float4 scv[4], shv[4];
    u32x4e rva[2][NB];  // residual vectors of this lane's two store blocks (block pair jp: block jp + half), per pixel block
    if constexpr (SC || SH || RES) {
      asm volatile("" ::: "memory");  // (one block's vectors at a time, see above)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = rowbase + j * 8 + 4 * half;
        if constexpr (SC) scv[j] = *(const float4*)(a.scale + c0);
        if constexpr (SH) shv[j] = *(const float4*)(a.shift + c0);
      }
      if constexpr (RES) {
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const unsigned pl = (unsigned)((rowbase >> 3) + 2 * jh + half) * HW * 16u;
            rva[jh][nb] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, (int)(pix16[nb] != ESS_OOB ? pl + pix16[nb] : ESS_OOB), 0, 0);
          }
      }
    }
#pragma unroll
    for (int jp = 0; jp < 4; jp += 2) {
      uint2 pk[2][NB];
      uint2 pl[HILO ? 2 : 1][HILO ? NB : 1];  // (HILO) the lo parts
      uint2 rr[2][NB];  // residual, exchanged to the accumulator layout: [block of the pair][pixel block] = this lane's 4 channels
      if constexpr (RES) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const u32x4e rv = rva[jp >> 1][nb];
          const auto s0 = __builtin_amdgcn_permlane32_swap(rv[0], rv[2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(rv[1], rv[3], false, false);
          rr[0][nb] = make_uint2(s0[0], s1[0]);
          rr[1][nb] = make_uint2(s0[1], s1[1]);
        }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = jp + jj;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mb][nb][4 * j + i];
          if constexpr (SC && SH) {
            v[0] = v[0] * scv[j].x + shv[j].x; v[1] = v[1] * scv[j].y + shv[j].y; v[2] = v[2] * scv[j].z + shv[j].z; v[3] = v[3] * scv[j].w + shv[j].w;
          } else if constexpr (SC) {
            v[0] *= scv[j].x; v[1] *= scv[j].y; v[2] *= scv[j].z; v[3] *= scv[j].w;
          } else if constexpr (SH) {
            v[0] += shv[j].x; v[1] += shv[j].y; v[2] += shv[j].z; v[3] += shv[j].w;
          }
          if constexpr (RES) {
            float rf[4];
            ess_up4<H>(rr[jj][nb], rf);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += rf[i];
          }
          if constexpr (RELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
          }
          pk[jj][nb] = ess_cvt4<F16 || H>(v[0], v[1], v[2], v[3]);
          if constexpr (HILO) pl[jj][nb] = ess_cvt4_lo(v[0], v[1], v[2], v[3]);
        }
      }
      const unsigned plane = (unsigned)((rowbase >> 3) + jp + half) * HW * 16u;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][nb].x, pk[1][nb].x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][nb].y, pk[1][nb].y, false, false);
        const u32x4e vec = {s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(vec, r_o, (int)(pix16[nb] != ESS_OOB ? plane + pix16[nb] : ESS_OOB), 0, ESS_C8_AUX);
        if constexpr (HILO) {
          const auto t0 = __builtin_amdgcn_permlane32_swap(pl[0][nb].x, pl[1][nb].x, false, false);
          const auto t1 = __builtin_amdgcn_permlane32_swap(pl[0][nb].y, pl[1][nb].y, false, false);
          const u32x4e vlo = {t0[0], t1[0], t0[1], t1[1]};
          __builtin_amdgcn_raw_buffer_store_b128(vlo, r_o, (int)(pix16[nb] != ESS_OOB ? plane + (unsigned)nblk * HW * 16u + pix16[nb] : ESS_OOB), 0, ESS_C8_AUX);
        }
      }
    }
  }
}

// The data-gradient forms as straight-line code: an optional second output (out_split, a multiple of 16 channels so that a
// block pair never straddles it) and / or a 2x2-sum-pooled first output with 32-wide pixel blocks (the two pixel blocks of a
// wave are vertical neighbours: one add in the lane, one across lane^1; only the even-x / even-y lanes of block 0 store).
template <int MB, bool POOL, int NB>
__device__ __forceinline__ void conv_epilogue_c8_dgrad(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int n, int half,
                                                       int x, int y0, const int (&ly)[NB]) {
  static_assert(!POOL || NB == 2, "the one-row pooled pairing assumes two pixel blocks per wave");
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
  constexpr int COT = MB * 32;
  const unsigned HW = (unsigned)(a.Hout * a.Wout);
  const int split = a.out_split;
  const int nb_first = (split > 0 ? split : a.Cout) >> 3, nb_second = split > 0 ? (a.Cout - split) >> 3 : 0;
  const int Wl = a.Wout >> 1;
  const unsigned HW1 = POOL ? (unsigned)((a.Hout >> 1) * Wl) : HW;
  const ess_rsrc r_o1 = ess_make_rsrc((const char*)a.out + (size_t)n * nb_first * HW1 * 16, (size_t)nb_first * HW1 * 16);
  const ess_rsrc r_o2 = ess_make_rsrc(split > 0 ? (const char*)a.out2 + (size_t)n * nb_second * HW * 16 : (const char*)a.out,
                                      split > 0 ? (size_t)nb_second * HW * 16 : 0);
  unsigned pix16[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int y = y0 + ly[nb];
    pix16[nb] = ((y < a.Hout) & (x < a.Wout)) ? (unsigned)(y * a.Wout + x) * 16u : ESS_OOB;
  }
  unsigned st_pool = ESS_OOB;  // pooled pixel of this lane (owner lanes only)
  if constexpr (POOL) {
    const int y = y0 + ly[0];
    if ((y < a.Hout) & (x < a.Wout) & !(x & 1) & !(y & 1)) st_pool = (unsigned)((y >> 1) * Wl + (x >> 1)) * 16u;
  }
  auto xor1 = [](float f) {  // lane^1 through DPP (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xf, 0xf, true));
  };
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int rowbase = ct * COT + mb * 32;
#pragma unroll
    for (int jp = 0; jp < 4; jp += 2) {
      const int blk0 = (rowbase >> 3) + jp;
      const bool first = blk0 < nb_first;  // (uniform; the whole pair: split % 16 == 0)
      if (POOL && first) {
        uint2 pk[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          bf16x4 b;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float p = acc[mb][0][4 * (jp + jj) + i] + acc[mb][1][4 * (jp + jj) + i];
            p += xor1(p);
            b[i] = (__bf16)p;
          }
          pk[jj] = __builtin_bit_cast(uint2, b);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        const u32x4e vec = {s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(vec, r_o1, (int)(st_pool != ESS_OOB ? (unsigned)(blk0 + half) * HW1 * 16u + st_pool : ESS_OOB), 0, 0);
      } else {
        uint2 pk[2][NB];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            bf16x4 b;
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = (__bf16)acc[mb][nb][4 * (jp + jj) + i];
            pk[jj][nb] = __builtin_bit_cast(uint2, b);
          }
        const unsigned plane = (unsigned)(blk0 + half - (first ? 0 : nb_first)) * HW * 16u;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][nb].x, pk[1][nb].x, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][nb].y, pk[1][nb].y, false, false);
          const u32x4e vec = {s0[0], s1[0], s0[1], s1[1]};
          const int o = (int)(pix16[nb] != ESS_OOB ? plane + pix16[nb] : ESS_OOB);
          if (first) __builtin_amdgcn_raw_buffer_store_b128(vec, r_o1, o, 0, ESS_C8_AUX);
          else __builtin_amdgcn_raw_buffer_store_b128(vec, r_o2, o, 0, ESS_C8_AUX);
        }
      }
    }
  }
}

template <int MB, bool SC, bool SH, bool H = false>
__device__ __forceinline__ void conv_epilogue_c8_sel(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half, int x,
                                                     int y0, const int (&ly)[NBW]) {
  // (all uniform) the plain form needs every channel of this workgroup's tile to exist, so the masks can go
  const bool relu = a.act == ESS_ACT_RELU, res = a.residual != nullptr;
  if constexpr (H) {  // half operands: half outputs; forward forms only (no data-gradient form); [hi | lo] only in the plain form (the entry point checks)
    const bool plain_h = a.out_split <= 0 && (a.act == ESS_ACT_NONE || relu) && (a.Cout % (MB * 32)) == 0;
    if (!plain_h) { conv_epilogue_c8_impl<MB, SC, SH, true>(a, acc, ct, n, half, x, y0, ly); return; }
    if (a.hilo) {
      if (relu) conv_epilogue_c8_plain<MB, SC, SH, true, true, false, NBW, true, true>(a, acc, ct, n, half, x, y0, ly);
      else conv_epilogue_c8_plain<MB, SC, SH, true, false, false, NBW, true, true>(a, acc, ct, n, half, x, y0, ly);
    } else if (res) {
      if (relu) conv_epilogue_c8_plain<MB, SC, SH, true, true, true, NBW, true>(a, acc, ct, n, half, x, y0, ly);
      else conv_epilogue_c8_plain<MB, SC, SH, true, false, true, NBW, true>(a, acc, ct, n, half, x, y0, ly);
    } else {
      if (relu) conv_epilogue_c8_plain<MB, SC, SH, true, true, false, NBW, true>(a, acc, ct, n, half, x, y0, ly);
      else conv_epilogue_c8_plain<MB, SC, SH, true, false, false, NBW, true>(a, acc, ct, n, half, x, y0, ly);
    }
    return;
  }
  const bool plain = a.out_split <= 0 && (a.act == ESS_ACT_NONE || relu) && (a.Cout % (MB * 32)) == 0 && !(a.out_f16 && (relu || res));
  if constexpr (!SC && !SH) {
    const bool pool = a.act == ESS_ACT_SUMPOOL2;
    const bool dgrad = (pool || a.out_split > 0) && (pool || a.act == ESS_ACT_NONE) && !res && !a.out_f16 && (a.out_split & 15) == 0 &&
                       (a.Cout % (MB * 32)) == 0 && (!pool || a.bwl == 5);
    if (dgrad) {
      if (pool) conv_epilogue_c8_dgrad<MB, true>(a, acc, ct, n, half, x, y0, ly);
      else conv_epilogue_c8_dgrad<MB, false>(a, acc, ct, n, half, x, y0, ly);
      return;
    }
  }
  if (!plain) { conv_epilogue_c8_impl<MB, SC, SH>(a, acc, ct, n, half, x, y0, ly); return; }
  if (a.out_f16) {  // (pre-norm tensors: no activation, no residual)
    conv_epilogue_c8_plain<MB, SC, SH, true, false, false>(a, acc, ct, n, half, x, y0, ly);
  } else if (res) {
    if (relu) conv_epilogue_c8_plain<MB, SC, SH, false, true, true>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_plain<MB, SC, SH, false, false, true>(a, acc, ct, n, half, x, y0, ly);
  } else {
    if (relu) conv_epilogue_c8_plain<MB, SC, SH, false, true, false>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_plain<MB, SC, SH, false, false, false>(a, acc, ct, n, half, x, y0, ly);
  }
}

// biased: the caller started its accumulators from the shift vector (conv_bias_init), nothing is left to add here
template <int MB, bool H = false>
__device__ __forceinline__ void conv_epilogue_c8(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half, int x,
                                                 int y0, const int (&ly)[NBW], bool biased = false) {
#ifndef ESS_C8_EPI_GENERAL
  if (biased) {  // (uniform)
    conv_epilogue_c8_sel<MB, false, false, H>(a, acc, ct, n, half, x, y0, ly);
  } else if (a.scale) {
    if (a.shift) conv_epilogue_c8_sel<MB, true, true, H>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_sel<MB, true, false, H>(a, acc, ct, n, half, x, y0, ly);
  } else {
    if (a.shift) conv_epilogue_c8_sel<MB, false, true, H>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_sel<MB, false, false, H>(a, acc, ct, n, half, x, y0, ly);
  }
  return;
#endif
  if (biased) {  // (uniform)
    conv_epilogue_c8_impl<MB, false, false>(a, acc, ct, n, half, x, y0, ly);
  } else if (a.scale) {
    if (a.shift) conv_epilogue_c8_impl<MB, true, true>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_impl<MB, true, false>(a, acc, ct, n, half, x, y0, ly);
  } else {
    if (a.shift) conv_epilogue_c8_impl<MB, false, true>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_impl<MB, false, false>(a, acc, ct, n, half, x, y0, ly);
  }
}

// Epilogue of the wide-tile kernel (conv_bf16_wide.hip; NB = 5 pixel blocks per wave): the straight-line forms only.  The
// dispatcher routes a launch to that kernel only when one of them applies (conv_bf16.hip, wide_pick / wide_pick_recurrent): one output with
// every channel of the tile real (scale / shift / residual / ReLU / F16 options), or the two outputs of a concat's data-gradient.
template <int MB, bool SC, bool SH, int NB, bool H = false>
__device__ __forceinline__ void conv_epilogue_c8_wide_sel(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int n, int half, int x,
                                                          int y0, const int (&ly)[NB]) {
  const bool relu = a.act == ESS_ACT_RELU, res = a.residual != nullptr;
  if constexpr (H) {  // half operands: half outputs (plain forms without a residual only: the dispatcher routes nothing else here)
    if (a.hilo) {
      if (relu) conv_epilogue_c8_plain<MB, SC, SH, true, true, false, NB, true, true>(a, acc, ct, n, half, x, y0, ly);
      else conv_epilogue_c8_plain<MB, SC, SH, true, false, false, NB, true, true>(a, acc, ct, n, half, x, y0, ly);
    } else {
      if (relu) conv_epilogue_c8_plain<MB, SC, SH, true, true, false, NB, true>(a, acc, ct, n, half, x, y0, ly);
      else conv_epilogue_c8_plain<MB, SC, SH, true, false, false, NB, true>(a, acc, ct, n, half, x, y0, ly);
    }
    return;
  }
  if constexpr (!SC && !SH) {
    if (a.out_split > 0) { conv_epilogue_c8_dgrad<MB, false>(a, acc, ct, n, half, x, y0, ly); return; }
  }
  if (a.out_f16) {
    conv_epilogue_c8_plain<MB, SC, SH, true, false, false>(a, acc, ct, n, half, x, y0, ly);
  } else if (res) {
    if (relu) conv_epilogue_c8_plain<MB, SC, SH, false, true, true>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_plain<MB, SC, SH, false, false, true>(a, acc, ct, n, half, x, y0, ly);
  } else {
    if (relu) conv_epilogue_c8_plain<MB, SC, SH, false, true, false>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_plain<MB, SC, SH, false, false, false>(a, acc, ct, n, half, x, y0, ly);
  }
}
template <int MB, int NB, bool H = false>
__device__ __forceinline__ void conv_epilogue_c8_wide(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int n, int half, int x,
                                                      int y0, const int (&ly)[NB], bool biased) {
  if (biased) {  // (uniform; the accumulators started from the shift vector)
    conv_epilogue_c8_wide_sel<MB, false, false, NB, H>(a, acc, ct, n, half, x, y0, ly);
  } else if (a.scale) {
    if (a.shift) conv_epilogue_c8_wide_sel<MB, true, true, NB, H>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_wide_sel<MB, true, false, NB, H>(a, acc, ct, n, half, x, y0, ly);
  } else {
    if (a.shift) conv_epilogue_c8_wide_sel<MB, false, true, NB, H>(a, acc, ct, n, half, x, y0, ly);
    else conv_epilogue_c8_wide_sel<MB, false, false, NB, H>(a, acc, ct, n, half, x, y0, ly);
  }
}

// ConvLSTM gate epilogue of the lean time steps as straight-line code: F32_C8 cell state in and out (or no previous state),
// accumulators started from the bias, every hidden channel of the tile real.  The general form below carries the
// NCHW / F32_C8 / bias-load choices as uniform branches inside the unrolled rows; hipcc then waits for the c_prev loads right
// where they are issued (a join of the two load forms) -- cycle stamps: 2.5 k cycles of exposed latency + 6.7 k of branchy
// arithmetic per 64 x 256 tile, 15-26 % of a matrix wave's tile.  Here the four activations that do not need c_prev (80 % of
// the transcendental work) run, in place in the accumulators, while the loads are in flight.  The BF16_C8 copy of h' leaves as
// whole 16-byte pixel vectors (half-wave swap between the two hidden blocks of a pair, as in conv_epilogue_c8).
template <int MB, int NB, bool H = false>
__device__ __forceinline__ void conv_epilogue_lstm_c8(const ConvKArgs& a, f32x16 (&acc)[MB][NB], int ct, int n, int half,
                                                      const int (&pixi)[NB], unsigned HW) {
  typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
  const int nbh = a.hid >> 3;
  const int ncp = (H && a.hilo) ? 2 : 1;  // (uniform) [hi | lo] copy of h': 2 nbh blocks per sample (ESS_LSTM_H_HILO)
  const size_t state_b = (size_t)nbh * HW * 32;
  const ess_rsrc r_prev = ess_make_rsrc(a.aux0 ? (const char*)(a.aux0 + (size_t)n * nbh * 8 * HW) : (const char*)a.out2, a.aux0 ? state_b : 0);
  const ess_rsrc r_c = ess_make_rsrc(a.out2 + (size_t)n * nbh * 8 * HW, state_b);
  const ess_rsrc r_h = ess_make_rsrc(a.out ? a.out + (size_t)n * nbh * 8 * HW : a.out2, a.out ? state_b : 0);
  const ess_rsrc r_hb = ess_make_rsrc(a.out_bf ? (const char*)a.out_bf + (size_t)n * ncp * nbh * HW * 16 : (const char*)a.out2, a.out_bf ? (size_t)ncp * nbh * HW * 16 : 0);
  unsigned vo[MB][NB];
  u32x4c cp[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      vo[mb][nb] = pixi[nb] >= 0 ? ((unsigned)(ct * MB + mb) * HW + (unsigned)pixi[nb]) * 32u + 16u * half : ESS_OOB;
      cp[mb][nb] = __builtin_amdgcn_raw_buffer_load_b128(r_prev, (int)vo[mb][nb], 0, ESS_EPI_AUX);  // (no previous state: zeros)
    }
  __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the loads to their first use to save 16 registers)
  // packed row 8*g + j of a 32-row block = gate g (in, remember, out, cell) of hidden hb*8 + j
  // (written as two phases -- the activations that do not need c_prev, then the rest; hipcc interleaves them per row to stay
  // inside 128 registers, and a scheduling barrier between the phases made it spill ~100: the first row's 12 activations are what
  // covers the load latency)
  float igc[MB][NB][4], gf[MB][NB][4], go[MB][NB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        igc[mb][nb][jj] = ess_sigmoid(acc[mb][nb][jj]) * ess_tanh(acc[mb][nb][12 + jj]);
        gf[mb][nb][jj] = ess_sigmoid(acc[mb][nb][4 + jj]);
        go[mb][nb][jj] = ess_sigmoid(acc[mb][nb][8 + jj]);
      }
  float hn[MB][NB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      u32x4c cv;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float cprev = __builtin_bit_cast(float, (unsigned)cp[mb][nb][jj]);
        const float cn = gf[mb][nb][jj] * cprev + igc[mb][nb][jj];
        cv[jj] = __builtin_bit_cast(unsigned, cn);
        hn[mb][nb][jj] = go[mb][nb][jj] * ess_tanh(cn);
      }
      __builtin_amdgcn_raw_buffer_store_b128(cv, r_c, (int)vo[mb][nb], 0, ESS_EPI_AUX);
    }
  if (a.out) {  // (uniform; NULL in the lean steps: only the BF16_C8 copy of h' is wanted)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const u32x4c hv = {__builtin_bit_cast(unsigned, hn[mb][nb][0]), __builtin_bit_cast(unsigned, hn[mb][nb][1]),
                           __builtin_bit_cast(unsigned, hn[mb][nb][2]), __builtin_bit_cast(unsigned, hn[mb][nb][3])};
        __builtin_amdgcn_raw_buffer_store_b128(hv, r_h, (int)vo[mb][nb], 0, ESS_EPI_AUX);
      }
  }
  if (a.out_bf) {  // (uniform)
    auto pack = [&](int mb, int nb) { return ess_cvt4<H>(hn[mb][nb][0], hn[mb][nb][1], hn[mb][nb][2], hn[mb][nb][3]); };
    if constexpr (MB >= 2) {
#pragma unroll
      for (int mb = 0; mb < MB; mb += 2)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const uint2 p0 = pack(mb, nb), p1 = pack(mb + 1, nb);
          const auto s0 = __builtin_amdgcn_permlane32_swap(p0.x, p1.x, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(p0.y, p1.y, false, false);
          const u32x4c vec = {s0[0], s1[0], s0[1], s1[1]};  // lanes 0-31: block mb, lanes 32-63: block mb + 1
          const unsigned o = pixi[nb] >= 0 ? ((unsigned)(ct * MB + mb + half) * HW + (unsigned)pixi[nb]) * 16u : ESS_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(vec, r_hb, (int)o, 0, ESS_EPI_AUX);
          if constexpr (H) {
            if (ncp == 2) {  // (uniform) the lo parts, nbh blocks further on
              const uint2 q0 = ess_cvt4_lo(hn[mb][nb][0], hn[mb][nb][1], hn[mb][nb][2], hn[mb][nb][3]);
              const uint2 q1 = ess_cvt4_lo(hn[mb + 1][nb][0], hn[mb + 1][nb][1], hn[mb + 1][nb][2], hn[mb + 1][nb][3]);
              const auto t0 = __builtin_amdgcn_permlane32_swap(q0.x, q1.x, false, false);
              const auto t1 = __builtin_amdgcn_permlane32_swap(q0.y, q1.y, false, false);
              const u32x4c vlo = {t0[0], t1[0], t0[1], t1[1]};
              __builtin_amdgcn_raw_buffer_store_b128(vlo, r_hb, (int)(o == ESS_OOB ? ESS_OOB : o + (unsigned)nbh * HW * 16u), 0, ESS_EPI_AUX);
            }
          }
        }
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (pixi[nb] >= 0) ess_store_bf16x4<H>(a.out_bf, (size_t)n * nbh, ct, HW, pixi[nb], half, hn[0][nb][0], hn[0][nb][1], hn[0][nb][2], hn[0][nb][3]);
    }
  }
}

// ALLOW8 = false: the caller dispatches BF16_C8 outputs to a dedicated kernel instantiation (conv_epilogue_c8 only: a fraction
// of the code and registers of this function), so the run-time branch to it is left out here.
template <int MB, int EPI, bool ALLOW8 = true, bool H = false>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half, int x,
                                              int y0, const int (&ly)[NBW], bool biased = false) {
  constexpr int COT = MB * 32;
  const unsigned HW = (unsigned)(a.Hout * a.Wout);
  const unsigned plane_b = HW * 4u;  // bytes of one channel plane
  unsigned voff[NBW];                // byte offset of (channel 4*half, this lane's pixel) inside one sample
  int pixi[NBW];                     // pixel index, -1 outside the image
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int y = y0 + ly[nb];
    const bool inb = y < a.Hout && x < a.Wout;
    pixi[nb] = inb ? y * a.Wout + x : -1;
    voff[nb] = inb ? ((unsigned)pixi[nb] + 4u * half * HW) * 4u : ESS_OOB;
  }
  if constexpr (EPI == ESS_EPI_LSTM) {
#ifndef ESS_LSTM_EPI_GENERAL
    if (biased && a.fmt_out == ESS_FMT_F32_C8 && (!a.aux0 || a.fmt_res == ESS_FMT_F32_C8) && (a.hid % (8 * MB)) == 0) {  // (uniform)
      conv_epilogue_lstm_c8<MB, NBW, H>(a, acc, ct, n, half, pixi, HW);
      return;
    }
#endif
  }
  if constexpr (EPI == ESS_EPI_GRU_OUT) {
#ifndef ESS_GRU_EPI_GENERAL
    if (biased && a.fmt_res == ESS_FMT_F32_C8 && a.aux1 && (!a.out || a.fmt_out == ESS_FMT_F32_C8) && (a.hid % (32 * MB)) == 0) {  // (uniform)
      conv_epilogue_gru_out_c8<MB, NBW, H>(a, acc, ct, n, half, pixi, HW);
      return;
    }
#endif
    conv_epilogue_gru_out<MB, H>(a, acc, ct, n, half, voff, pixi, HW, biased);
  } else if constexpr (EPI == ESS_EPI_GRU_UR) {
#ifndef ESS_GRU_EPI_GENERAL
    if (biased && a.out && !a.out2 && a.fmt_out == ESS_FMT_F32_C8 && (!a.aux0 || a.fmt_res == ESS_FMT_F32_C8) && (a.hid % (16 * MB)) == 0) {  // (uniform)
      conv_epilogue_gru_ur_c8<MB, NBW, H>(a, acc, ct, n, half, pixi, HW);
      return;
    }
#endif
    conv_epilogue_gru_ur<MB, H>(a, acc, ct, n, half, voff, pixi, HW, biased);
  } else if constexpr (EPI == ESS_EPI_LINEAR) {
    if constexpr (ALLOW8) {
      if (a.fmt_out == ESS_FMT_BF16_C8) { conv_epilogue_c8<MB, H>(a, acc, ct, n, half, x, y0, ly); return; }
    }
    const bool bare = a.act == ESS_ACT_NONE && !a.out_bf && a.out;
    if (a.act == ESS_ACT_SUMPOOL2) conv_epilogue_pool<MB>(a, acc, ct, n, half, x, y0, ly, plane_b);
    else if (bare && !a.scale && !a.residual && a.out_split == 0) conv_epilogue_plain<MB>(a, acc, ct, n, half, voff, plane_b);
    else if (bare && !a.scale && a.residual && a.out_split == 0)
      conv_epilogue_rows<MB, EPI, false, true, 1, H>(a, acc, ct, n, half, voff, pixi, plane_b);
    else if (bare && !a.scale && !a.residual && a.out_split > 0)
      conv_epilogue_rows<MB, EPI, false, false, 3, H>(a, acc, ct, n, half, voff, pixi, plane_b);
    else if (a.act == ESS_ACT_RELU && a.scale && !a.residual && a.out_split == 0)
      conv_epilogue_rows<MB, EPI, true, false, 2, H>(a, acc, ct, n, half, voff, pixi, plane_b);
    else if (!a.scale && !a.residual) conv_epilogue_rows<MB, EPI, false, false, 0, H>(a, acc, ct, n, half, voff, pixi, plane_b);
    else if (!a.scale) conv_epilogue_rows<MB, EPI, false, true, 0, H>(a, acc, ct, n, half, voff, pixi, plane_b);
    else if (!a.residual) conv_epilogue_rows<MB, EPI, true, false, 0, H>(a, acc, ct, n, half, voff, pixi, plane_b);
    else conv_epilogue_rows<MB, EPI, true, true, 0, H>(a, acc, ct, n, half, voff, pixi, plane_b);
  } else {
    const ess_rsrc r_out = ess_make_rsrc(a.out + (size_t)n * a.hid * HW, (size_t)a.hid * plane_b);
    const ess_rsrc r_out2 = ess_make_rsrc(a.out2 + (size_t)n * a.hid * HW, (size_t)a.hid * plane_b);
    const ess_rsrc r_prev =
        ess_make_rsrc(a.aux0 ? a.aux0 + (size_t)n * a.hid * HW : a.out, a.aux0 ? (size_t)a.hid * plane_b : 0);
    const ess_rsrc r_sh = ess_make_rsrc(a.shift, (size_t)(a.n_cout_tiles * COT) * 4);
    if constexpr (EPI == ESS_EPI_LSTM) {
      // packed row 8*g + j of a 32-row block = gate g (in, remember, out, cell) of hidden hb*8 + j
      float cprev[MB][NBW][4];
      // soffset is outside the hardware range check: a hidden channel past the end is folded into voffset
      auto vo = [&](int mb, int nb, int jj) { return (ct * MB + mb) * 8 + 4 * half + jj < a.hid ? voff[nb] : ESS_OOB; };
      // ESS_FMT_F32_C8 cell states (between the lean time steps): [N][hid/8][H][W][8] fp32 -- this lane's 4 channels of a pixel are
      // 16 contiguous bytes, a pixel block of the wave 1 KiB: one 16-byte access where the planes take four 4-byte ones
      const int nbh = (a.hid + 7) >> 3;
      const bool cin8 = a.fmt_res == ESS_FMT_F32_C8, cout8 = a.fmt_out == ESS_FMT_F32_C8;
      const ess_rsrc r_prev8 = ess_make_rsrc((cin8 && a.aux0) ? a.aux0 + (size_t)n * nbh * 8 * HW : a.out2, (cin8 && a.aux0) ? (size_t)nbh * HW * 32 : 0);
      const ess_rsrc r_out28 = ess_make_rsrc(a.out2 + (size_t)n * nbh * 8 * HW, cout8 ? (size_t)nbh * HW * 32 : 0);
      const ess_rsrc r_out8 = ess_make_rsrc(a.out ? a.out + (size_t)n * nbh * 8 * HW : a.out2, (cout8 && a.out) ? (size_t)nbh * HW * 32 : 0);
      auto vo8 = [&](int mb, int nb) {  // byte offset of (block ct*MB+mb, this lane's pixel, channels 4*half..+3)
        return (pixi[nb] >= 0 && ct * MB + mb < nbh) ? ((unsigned)(ct * MB + mb) * HW + (unsigned)pixi[nb]) * 32u + 16u * half : ESS_OOB;
      };
      if (cin8) {  // (uniform)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) {
            typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
            const u32x4c v = __builtin_amdgcn_raw_buffer_load_b128(r_prev8, (int)vo8(mb, nb), 0, 0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) cprev[mb][nb][jj] = __builtin_bit_cast(float, (unsigned)v[jj]);
          }
      } else {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              cprev[mb][nb][jj] = ess_bload(r_prev, vo(mb, nb, jj), (unsigned)((ct * MB + mb) * 8 + jj) * plane_b);
      }
#ifdef ESS_CV_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ESS_EPI_STAMP(43);
#endif
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int rowbase = ct * COT + mb * 32;
        float sh[16];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) sh[4 * g + jj] = biased ? 0.f : ess_bload(r_sh, 16u * half, (unsigned)(rowbase + 8 * g + jj) * 4u);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
          float hq[4], cq[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float gi = ess_sigmoid(acc[mb][nb][jj] + sh[jj]);
            const float gf = ess_sigmoid(acc[mb][nb][4 + jj] + sh[4 + jj]);
            const float go = ess_sigmoid(acc[mb][nb][8 + jj] + sh[8 + jj]);
            const float gc = ess_tanh(acc[mb][nb][12 + jj] + sh[12 + jj]);
            const float cn = gf * cprev[mb][nb][jj] + gi * gc;
            const unsigned so = (unsigned)((ct * MB + mb) * 8 + jj) * plane_b;  // hidden channel (+ 4*half in voff)
            const float hn = go * ess_tanh(cn);
            if (!cout8) {
              ess_bstore(cn, r_out2, vo(mb, nb, jj), so);
              if (a.out) ess_bstore(hn, r_out, vo(mb, nb, jj), so);  // (NULL: only the BF16_C8 copy of h' is wanted)
            }
            const bool real = (ct * MB + mb) * 8 + 4 * half + jj < a.hid;
            hq[jj] = real ? hn : 0.f;
            cq[jj] = real ? cn : 0.f;
          }
          if (cout8) {  // (uniform)
            typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
            const u32x4c cv = {__builtin_bit_cast(unsigned, cq[0]), __builtin_bit_cast(unsigned, cq[1]), __builtin_bit_cast(unsigned, cq[2]), __builtin_bit_cast(unsigned, cq[3])};
            __builtin_amdgcn_raw_buffer_store_b128(cv, r_out28, (int)vo8(mb, nb), 0, 0);
            if (a.out) {
              const u32x4c hv = {__builtin_bit_cast(unsigned, hq[0]), __builtin_bit_cast(unsigned, hq[1]), __builtin_bit_cast(unsigned, hq[2]), __builtin_bit_cast(unsigned, hq[3])};
              __builtin_amdgcn_raw_buffer_store_b128(hv, r_out8, (int)vo8(mb, nb), 0, 0);
            }
          }
          if (a.out_bf && pixi[nb] >= 0 && ct * MB + mb < ((a.hid + 7) >> 3))  // hidden block ct*MB+mb, channels 4*half..+3
            ess_store_bf16x4<H>(a.out_bf, (size_t)n * ((a.hid + 7) >> 3), ct * MB + mb, HW, pixi[nb], half, hq[0], hq[1], hq[2], hq[3]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
struct Geom {
  int bwl, wxl, TW, TH, tiles_x, tiles_y, IH, IW, row_pitch, par_off, plane;
};

inline bool is_bf16(const EssConvDesc* d) { return d->compute == ESS_COMPUTE_BF16; }
inline bool ws_enabled();
// ESS_COMPUTE_BF16X3 resolves, per convolution, to the arithmetic that runs: the bf16 3x3 / stride-1 wave-specialised kernel with
// split operands (`split`), or the exact-fp32 kernels for every other geometry.  The entry points work on the resolved copy.
// ESS_COMPUTE_F16 resolves to the bf16 kernels' H = true instantiations (`f16`): same geometry, plans and pack layouts, IEEE-half
// elements in every 16-bit tensor of the call (the resolved copy names them BF16_C8, the kernels' only 16-bit layout);
// `hilo`: the 16-bit output (LINEAR, ESS_FMT_F16_C8_HILO) / the copy of h' (LSTM, act = ESS_LSTM_H_HILO) leaves as [hi | lo].
struct ResolvedDesc { EssConvDesc d; bool split; bool f16 = false; bool hilo = false; };
// ESS_CONV_PAIR=0 switches the tap-paired 5x5 kernel off (tuning); ONE reading of the variable for resolve_compute and is_paired
inline bool pair_enabled() {
  static const bool on = [] { const char* e = getenv("ESS_CONV_PAIR"); return !(e && e[0] == '0'); }();
  return on;
}
inline ResolvedDesc resolve_compute(const EssConvDesc* d) {
  ResolvedDesc r{*d, false};
  if (d->compute == ESS_COMPUTE_F16) {
    r.f16 = true;
    r.d.compute = ESS_COMPUTE_BF16;
    if (r.d.fmt0 == ESS_FMT_F16_C8) r.d.fmt0 = ESS_FMT_BF16_C8;
    if (r.d.fmt1 == ESS_FMT_F16_C8) r.d.fmt1 = ESS_FMT_BF16_C8;
    if (r.d.fmt_res == ESS_FMT_F16_C8) r.d.fmt_res = ESS_FMT_BF16_C8;
    if (d->epilogue == ESS_EPI_LINEAR) {
      r.hilo = d->fmt_out == ESS_FMT_F16_C8_HILO;
      if (d->fmt_out == ESS_FMT_F16_C8 || d->fmt_out == ESS_FMT_F16_C8_HILO) r.d.fmt_out = ESS_FMT_BF16_C8;
    } else if (d->epilogue == ESS_EPI_LSTM) {
      r.hilo = d->act == ESS_LSTM_H_HILO;
      r.d.act = ESS_ACT_NONE;
    } else if (d->epilogue == ESS_EPI_GRU_OUT) {
      r.hilo = (d->act & ESS_GRU_H_HILO) != 0;
      r.d.act = d->act & ~ESS_GRU_H_HILO;
    }
    return r;
  }
  if (d->compute == ESS_COMPUTE_BF16X3) {
    // 3x3 / stride 1 (any epilogue): the wave-specialised kernel; 5x5 LINEAR with at least one 8-channel chunk of input (the
    // frozen E2VID's stride-2 encoder convolutions and upsample-conv decoders): the tap-paired kernel.  The 2-channel 5x5 head
    // stays exact fp32 -- three passes over a chunk that is 6/8 padding would cost twice the fp32 kernel
    const bool pair_on = pair_enabled();
    const bool ws = d->ksize == 3 && d->stride == 1 && ws_enabled();
    const bool pair = d->ksize == 5 && d->epilogue == ESS_EPI_LINEAR && d->C0 + d->C1 >= 8 && pair_on;
    // 3x3 / stride 2 (the ResNet prefix's downsampling convolutions; LINEAR): the generic tile kernel with the same three virtual
    // chunks (round 5).  1x1 convolutions stay exact fp32: they are bound by their fp32 tensors' bytes, three passes buy nothing
    static const bool gen_on = [] { const char* e = getenv("ESS_X3_GENERIC"); return !(e && e[0] == '0'); }();
    const bool gen = gen_on && d->ksize == 3 && d->stride == 2 && d->epilogue == ESS_EPI_LINEAR && d->C0 + d->C1 >= 16;
    r.d.compute = (ws || pair || gen) ? ESS_COMPUTE_BF16 : ESS_COMPUTE_FP32;
    r.split = ws || pair || gen;
  }
  return r;
}

// bf16 5x5 convolutions run on the tap-paired wave-specialised kernel (conv_bf16.hip): 8-channel chunks, two taps per MFMA
inline bool is_paired(const EssConvDesc* d) {
  return pair_enabled() && is_bf16(d) && d->ksize == 5 && d->epilogue == ESS_EPI_LINEAR;
}

inline int pick_ck(const EssConvDesc* d) {
  const int cin = d->C0 + d->C1;
  if (is_paired(d)) return 8;
  if (is_bf16(d)) return (d->ksize == 1 && d->stride == 1 && cin >= 32) ? 32 : 16;  // one v_mfma_f32_32x32x16_bf16 K-step = 16 channels
  // fp32: K-step = 2 channels; chunk sized for LDS.  Only the 2-channel 5x5 head gets a narrower chunk.
  if (d->ksize == 5 && d->stride == 1 && cin <= 2) return 2;
  return d->ksize >= 7 ? 2 : (d->ksize == 5 ? 4 : 8);
}

// diagnostic switch ESS_CONV_WS=0: the generic tile kernel instead of the wave-specialised 3x3 one
inline bool ws_enabled() {
  static const bool on = [] { const char* e = getenv("ESS_CONV_WS"); return !(e && e[0] == '0'); }();
  return on;
}

inline int packed_rows(const EssConvDesc* d);
inline int pick_mb(const EssConvDesc* d) {
  // bf16 3x3/s1 (wave-specialised kernel): 128-channel tiles where there are enough output channels -- halves the
  // activation staging and the weight re-fetch per MFMA
  // (measured: pays only for the deepest layer -- 512 -> 1024 @ 60x80: 779 -> 859 TFLOP/s; one workgroup per CU hurts the rest)
  static const int mb4_min_cin = [] { const char* e = getenv("ESS_CONV_MB4_MIN_CIN"); return e ? atoi(e) : 512; }();  // (tuning experiments)
  if (ws_enabled() && d->compute == ESS_COMPUTE_BF16 && d->ksize == 3 && d->stride == 1 && packed_rows(d) >= 256 && d->C0 + d->C1 >= mb4_min_cin &&
      d->mode0 != ESS_SRC_S2D) {  // (the space-to-depth form runs on the wide-tile kernel over 64-row slabs)
    static const bool mb4 = [] { const char* e = getenv("ESS_CONV_MB4"); return !(e && e[0] == '0'); }();
    if (mb4) return 4;
  }
  // tap-paired 5x5 / stride 2: the input tile is 4x the output tile, so a 64-row workgroup needs 95 KB of LDS and runs
  // alone on its CU with prologue, K loop and epilogue strictly in sequence; 32-row tiles fit twice
  // (measured, B=8 encoder convs: 0.191 / 0.117 / 0.105 ms -> 0.167 / 0.102 / 0.085 ms)
  if (is_paired(d) && d->stride == 2) {
    static const int m = [] { const char* e = getenv("ESS_PAIR_S2_MB"); return e ? atoi(e) : 1; }();
    if (m == 1) return 1;
  }
  static const int force = [] { const char* e = getenv("ESS_CONV_MB_FORCE"); return e ? atoi(e) : 0; }();  // (tuning experiments)
  if (force && d->compute == ESS_COMPUTE_BF16 && d->ksize == 3 && d->stride == 1 && d->epilogue == ESS_EPI_LINEAR) return force;
  return d->C_out > 32 ? 2 : 1;
}

// tile positions (per channel / per 8-channel block) a thread can stage: bound of the kernels' register prefetch
constexpr int stage_kpc(int ks, int s) {
  return s == 1 ? (ks == 1 ? 1 : ks == 3 ? 2 : ks == 5 ? 2 : 3) : (ks == 1 ? 4 : ks == 3 ? 5 : ks == 5 ? 5 : 6);
}

inline Geom choose_geom(const EssConvDesc* d) {
  Geom best{};
  double best_cost = 1e300;
  const int KS = d->ksize, S = d->stride;
  static const char* const force_geom = getenv("ESS_CONV_GEOM");  // (read once, not per candidate and launch)
  for (int bwl = 5; bwl >= 3; --bwl) {
    for (int wxl = 0; wxl <= 2; ++wxl) {
      const int BW = 1 << bwl, RB = 32 >> bwl;
      const int TW = BW << wxl, TH = (4 >> wxl) * NBW * RB;
      const int tx = ceil_div(d->W_out, TW), ty = ceil_div(d->H_out, TH);
      const int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
      if (IH * IW > stage_kpc(KS, S) * 256) continue;  // would not fit the staging registers
      if (force_geom && (force_geom[0] - '0' != bwl || force_geom[1] - '0' != wxl)) continue;  // tuning hook
      // pooled output: with 32-wide pixel blocks the vertical partner of a pixel is in the same lane (conv_epilogue_pool)
      if (d->act == ESS_ACT_SUMPOOL2 && d->W_out >= 32 && bwl != 5) continue;
      // padded MACs (dominant) + a small halo/staging term + a coalescing term: a tile row is one contiguous run of the
      // NCHW planes for both the staging loads and the epilogue stores, and the large-plane layers are bound by how
      // HBM traffic is shaped (64->64 @240x320, B=8: 16x16 tiles 131 us, 32x8 121 us, 64x4 114 us); prefer wide blocks on ties
      const double cost = (double)tx * ty * TW * TH * (1.0 + 0.02 * (double)(IH * IW) / (TH * TW * S * S) + 0.04 * 64.0 / TW) +
                          1e-3 * (5 - bwl);
      if (cost < best_cost) {
        best_cost = cost;
        Geom g{};
        g.bwl = bwl; g.wxl = wxl; g.TW = TW; g.TH = TH; g.tiles_x = tx; g.tiles_y = ty; g.IH = IH; g.IW = IW;
        if (is_bf16(d)) {
          // units: one 16-byte pixel vector (8 bf16 channels), read with ds_read_b128 (16-lane groups): the rows
          // of one pixel block must start 0 (BW=16) / 8 (BW=8) vectors apart modulo 16 to stay conflict-free
          if (S == 1) {
            int rp = IW;
            if (BW == 16) while (rp & 15) ++rp;
            if (BW == 8) while ((rp & 15) != 8) ++rp;
            g.row_pitch = rp; g.par_off = 0;
          } else {
            int pw = (IW + 1) / 2;
            if (BW == 16) while (pw & 3) ++pw;
            if (BW == 8) while ((pw & 3) != 2) ++pw;
            g.row_pitch = 2 * pw; g.par_off = pw;
          }
          g.plane = IH * g.row_pitch;
        } else {
          if (S == 1) {
            int rp = IW;
            if (BW < 32) while ((rp & 31) != BW) ++rp;  // rows of one pixel block land on disjoint banks
            g.row_pitch = rp; g.par_off = 0;
          } else {
            int pw = (IW + 1) / 2;
            if (BW < 32) while (((4 * pw) & 31) != BW) ++pw;
            g.row_pitch = 2 * pw; g.par_off = pw;
          }
          g.plane = (IH * g.row_pitch + 3) & ~3;
        }
        best = g;
      }
    }
  }
  return best;
}

inline int validate(const EssConvDesc* d) {
  ESS_CHECK_ARG(d != nullptr, "conv: null descriptor");
  if (d->compute == ESS_COMPUTE_F16) {
    // half operands: the 16-bit tensors are F16_C8 (never BF16_C8); fp32 NCHW sources only where a kernel rounds them to half itself
    // (the 5x5 head); forward forms only
    ESS_CHECK_ARG(d->fmt0 != ESS_FMT_BF16_C8 && d->fmt1 != ESS_FMT_BF16_C8 && d->fmt_out != ESS_FMT_BF16_C8 && d->fmt_res != ESS_FMT_BF16_C8,
                  "conv(f16): 16-bit tensors of a half-operand convolution are ESS_FMT_F16_C8");
    ESS_CHECK_ARG(d->act != ESS_ACT_SUMPOOL2 || d->epilogue != ESS_EPI_LINEAR, "conv(f16): no pooled (data-gradient) form");
    ESS_CHECK_ARG(d->mode0 != ESS_SRC_ZERO_UP2 && d->mode1 != ESS_SRC_ZERO_UP2, "conv(f16): no zero-inserted sources");
    if (d->epilogue == ESS_EPI_LINEAR && d->fmt_out == ESS_FMT_F16_C8_HILO)
      ESS_CHECK_ARG(d->out_split == 0 && (d->act == ESS_ACT_NONE || d->act == ESS_ACT_RELU) && (d->C_out % 64) == 0,
                    "conv(f16): a [hi | lo] output needs act in {none, relu}, no out_split, C_out %% 64 == 0");
    if (d->epilogue == ESS_EPI_LSTM) ESS_CHECK_ARG(d->act == 0 || d->act == ESS_LSTM_H_HILO, "conv(f16, LSTM): act is 0 or ESS_LSTM_H_HILO");
    if (d->epilogue == ESS_EPI_GRU_UR) ESS_CHECK_ARG(d->act == ESS_GRU_U_F32 || d->act == ESS_GRU_U_F16, "conv(f16, GRU_UR): act is ESS_GRU_U_F32 or ESS_GRU_U_F16");
    const ResolvedDesc r = resolve_compute(d);
    return validate(&r.d);
  }
  ESS_CHECK_ARG(d->fmt_out != ESS_FMT_F16_C8_HILO, "conv: ESS_FMT_F16_C8_HILO is an output format of ESS_COMPUTE_F16");
  ESS_CHECK_ARG(d->N > 0 && d->H_in > 0 && d->W_in > 0 && d->C0 > 0 && d->C1 >= 0 && d->C_out > 0, "conv: bad extents");
  ESS_CHECK_ARG(d->ksize == 1 || d->ksize == 3 || d->ksize == 5 || d->ksize == 7, "conv: ksize %d unsupported", d->ksize);
  ESS_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
  ESS_CHECK_ARG(d->H_out == (d->H_in + 2 * d->pad - d->ksize) / d->stride + 1 &&
                    d->W_out == (d->W_in + 2 * d->pad - d->ksize) / d->stride + 1,
                "conv: output extent %dx%d inconsistent with input %dx%d k%d s%d p%d", d->H_out, d->W_out, d->H_in,
                d->W_in, d->ksize, d->stride, d->pad);
  for (int s = 0; s < 2; ++s) {
    const int m = s ? d->mode1 : d->mode0;
    ESS_CHECK_ARG(m >= 0 && m <= (s ? 2 : 3), "conv: bad source mode");
    if (m == ESS_SRC_NEAREST_UP2 || m == ESS_SRC_ZERO_UP2) ESS_CHECK_ARG(!(d->H_in & 1) && !(d->W_in & 1), "conv: x2 source needs even extent");
  }
  if (d->mode0 == ESS_SRC_S2D)  // the space-to-depth view of a BF16_C8 tensor: 3x3 / stride 1 / pad 1 over 4 x the stored channels
    ESS_CHECK_ARG(d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->C1 == 0 && (d->C0 % 128) == 0 && (d->C_out % 64) == 0 &&
                      d->epilogue == ESS_EPI_LINEAR && d->out_split == 0 && d->compute == ESS_COMPUTE_BF16 &&
                      (d->fmt0 == ESS_FMT_BF16_C8 || d->fmt0 == ESS_FMT_F32_NCHW),
                  "conv: ESS_SRC_S2D needs 3x3 s1 p1, one source with C0 %% 128 == 0, C_out %% 64 == 0, LINEAR, bf16 compute");
  ESS_CHECK_ARG(d->epilogue >= 0 && d->epilogue <= 3, "conv: bad epilogue");
  ESS_CHECK_ARG(d->compute == ESS_COMPUTE_FP32 || d->compute == ESS_COMPUTE_BF16 || d->compute == ESS_COMPUTE_BF16X3, "conv: bad compute type");
  if (d->compute == ESS_COMPUTE_BF16X3)
    ESS_CHECK_ARG(d->fmt0 == ESS_FMT_F32_NCHW && d->fmt1 == ESS_FMT_F32_NCHW && d->fmt_out == ESS_FMT_F32_NCHW && d->fmt_res == ESS_FMT_F32_NCHW,
                  "conv: split-operand bf16 (ESS_COMPUTE_BF16X3) works on fp32 NCHW tensors");
  if (d->epilogue == ESS_EPI_LSTM) ESS_CHECK_ARG(d->C_out == 4 * d->hidden, "conv: LSTM needs C_out = 4*hidden");
  if (d->epilogue == ESS_EPI_GRU_UR) ESS_CHECK_ARG(d->C_out == 2 * d->hidden, "conv: GRU_UR needs C_out = 2*hidden");
  if (d->epilogue == ESS_EPI_GRU_OUT) ESS_CHECK_ARG(d->C_out == d->hidden, "conv: GRU_OUT needs C_out = hidden");
  if (d->epilogue != ESS_EPI_LINEAR)
    ESS_CHECK_ARG(d->ksize == 3 && d->stride == 1 && d->out_split == 0, "conv: recurrent epilogues are 3x3 s1");
  ESS_CHECK_ARG(d->out_split >= 0 && d->out_split < d->C_out, "conv: bad out_split");
  ESS_CHECK_ARG(d->act >= ESS_ACT_NONE && d->act <= ESS_ACT_SUMPOOL2, "conv: bad act");
  if (d->act == ESS_ACT_SUMPOOL2)
    ESS_CHECK_ARG(d->epilogue == ESS_EPI_LINEAR && d->stride == 1 && (d->H_out & 1) == 0 && (d->W_out & 1) == 0,
                  "conv: SUMPOOL2 needs the LINEAR epilogue, stride 1 and even output extents");
  ESS_CHECK_ARG((d->fmt0 == ESS_FMT_F32_NCHW || d->fmt0 == ESS_FMT_BF16_C8) && (d->fmt1 == ESS_FMT_F32_NCHW || d->fmt1 == ESS_FMT_BF16_C8),
                "conv: bad source format");
  if (d->fmt0 != ESS_FMT_F32_NCHW || d->fmt1 != ESS_FMT_F32_NCHW) {
    ESS_CHECK_ARG(d->compute == ESS_COMPUTE_BF16, "conv: BF16_C8 sources need bf16 compute");
    if (d->ksize == 5)
      ESS_CHECK_ARG(d->epilogue == ESS_EPI_LINEAR && d->mode0 == ESS_SRC_DIRECT && (d->C1 == 0 || d->mode1 == ESS_SRC_DIRECT),
                    "conv: 5x5 BF16_C8 sources must be DIRECT (LINEAR epilogue)");
    else
      ESS_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "conv: BF16_C8 sources are staged by the 1x1, 3x3 and 5x5 kernels");
    ESS_CHECK_ARG(d->C1 == 0 || d->fmt0 == d->fmt1, "conv: both sources of a concat must use the same format");
    ESS_CHECK_ARG(d->C1 == 0 || (d->C0 % 8) == 0, "conv: the first BF16_C8 source of a concat must have a multiple of 8 channels");
  }
  if (d->epilogue == ESS_EPI_LSTM) {
    // channel-blocked fp32 cell states: fmt_res describes aux0 (c), fmt_out describes out / out2 (h', c')
    ESS_CHECK_ARG((d->fmt_out == ESS_FMT_F32_NCHW || d->fmt_out == ESS_FMT_F32_C8) && (d->fmt_res == ESS_FMT_F32_NCHW || d->fmt_res == ESS_FMT_F32_C8),
                  "conv(LSTM): state tensors are ESS_FMT_F32_NCHW or ESS_FMT_F32_C8");
    ESS_CHECK_ARG((int64_t)((d->hidden + 7) / 8) * 8 * d->H_out * d->W_out * 4 < (int64_t)1 << 31, "conv(LSTM): one sample of a state must stay below 2 GiB");
    return ESS_OK;
  }
  if (d->epilogue == ESS_EPI_GRU_UR || d->epilogue == ESS_EPI_GRU_OUT) {
    ESS_CHECK_ARG(d->act == ESS_GRU_U_F32 || (d->act == ESS_GRU_U_F16 && d->compute == ESS_COMPUTE_BF16), "conv(GRU): act is ESS_GRU_U_F32 or (bf16 compute) ESS_GRU_U_F16");
    // fmt_res describes h_prev (aux0) and, in the candidate kernel, u (aux1); fmt_out the fp32 outputs (u and r*h | h')
    ESS_CHECK_ARG((d->fmt_out == ESS_FMT_F32_NCHW || d->fmt_out == ESS_FMT_F32_C8) && (d->fmt_res == ESS_FMT_F32_NCHW || d->fmt_res == ESS_FMT_F32_C8),
                  "conv(GRU): state tensors are ESS_FMT_F32_NCHW or ESS_FMT_F32_C8");
    ESS_CHECK_ARG((int64_t)((d->hidden + 7) / 8) * 8 * d->H_out * d->W_out * 4 < (int64_t)1 << 31, "conv(GRU): one sample of a state must stay below 2 GiB");
    return ESS_OK;
  }
  ESS_CHECK_ARG((d->fmt_out == ESS_FMT_F32_NCHW || d->fmt_out == ESS_FMT_BF16_C8) && (d->fmt_res == ESS_FMT_F32_NCHW || d->fmt_res == ESS_FMT_BF16_C8),
                "conv: bad output / residual format");  // (ESS_FMT_F16_C8 outputs arrive here as BF16_C8 + the kernels' out_f16 flag)
  if (d->fmt_out == ESS_FMT_BF16_C8)
    ESS_CHECK_ARG(d->epilogue == ESS_EPI_LINEAR && (d->out_split % 8) == 0 &&
                      (d->act == ESS_ACT_NONE || d->act == ESS_ACT_RELU || d->act == ESS_ACT_SUMPOOL2),
                  "conv: BF16_C8 outputs need the LINEAR epilogue, out_split %% 8 == 0 and act in {none, relu, sumpool2}");
  ESS_CHECK_ARG(d->fmt_res == d->fmt_out || d->fmt_res == ESS_FMT_F32_NCHW, "conv: a BF16_C8 residual needs a BF16_C8 output");
  ESS_CHECK_ARG((int64_t)d->C_out * d->H_out * d->W_out * 4 < (int64_t)1 << 31,
                "conv: one output sample must stay below 2 GiB (32-bit buffer offsets in the epilogue)");
  return ESS_OK;
}

// rows of the packed weight matrix that exist for this epilogue (before padding to the tile)
inline int packed_rows(const EssConvDesc* d) {
  switch (d->epilogue) {
    case ESS_EPI_LSTM: return ceil_div(d->hidden, 8) * 32;
    case ESS_EPI_GRU_UR: return ceil_div(d->hidden, 16) * 32;
    default: return d->C_out;
  }
}

inline void make_plan(const EssConvDesc* d, EssConvPlan* pl, bool split = false) {
  const int mb = pick_mb(d);
  pl->cout_tile = mb * 32;
  pl->ck = pick_ck(d);
  pl->n_chunks = ceil_div(d->C0 + d->C1, pl->ck);
  pl->n_cout_tiles = ceil_div(packed_rows(d), pl->cout_tile);
  pl->rows_padded = pl->n_cout_tiles * pl->cout_tile;
  pl->packed_elems = (int64_t)pl->rows_padded * pl->n_chunks * pl->ck * d->ksize * d->ksize;
  const Geom g = choose_geom(d);
  if (is_paired(d)) {
    const int np = (d->ksize * d->ksize + 1) / 2;  // tap pairs; 16 k-values each
    pl->packed_elems = (int64_t)pl->rows_padded * pl->n_chunks * np * 16 * (split ? 2 : 1);  // (split: a hi and a lo slab per chunk)
    pl->packed_bytes = pl->packed_elems * 2;
    pl->lds_bytes = (g.plane + np * 2 * pl->cout_tile) * 16;  // one stage; the kernel double-buffers
  } else if (is_bf16(d)) {
    if (split) pl->packed_elems *= 2;  // a hi and a lo slab per (channel tile, chunk): [tile][chunk][hi | lo][tap][c/8][cout][8]
    pl->packed_bytes = pl->packed_elems * 2;
    pl->lds_bytes = ((pl->ck / 8) * g.plane + d->ksize * d->ksize * (pl->ck / 8) * pl->cout_tile) * 16;
  } else {
    pl->packed_bytes = pl->packed_elems * 4;
    pl->lds_bytes = (pl->ck * g.plane + d->ksize * d->ksize * pl->ck * pl->cout_tile) * 4;
  }
}

// packed row -> (source tensor selector, source row) or -1
__device__ __forceinline__ int map_row(int prow, int epi, int hid, int cout, int* sel) {
  *sel = 0;
  if (epi == ESS_EPI_LSTM) {
    const int b = prow >> 5, q = (prow & 31) >> 3, j = prow & 7;
    const int hc = b * 8 + j;
    return hc < hid ? q * hid + hc : -1;
  }
  if (epi == ESS_EPI_GRU_UR) {
    const int b = prow >> 5, q = (prow & 31) >> 3, j = prow & 7;
    const int hc = b * 16 + (q >> 1) * 8 + j;
    *sel = q & 1;
    return hc < hid ? hc : -1;
  }
  return prow < cout ? prow : -1;
}

static __global__ void pack_weights_kernel(const float* w, const float* w2, float* out, int64_t total, int cot, int ck,
                                    int n_chunks, int ks, int cin, int cout, int epi, int hid, int w_kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t t = i;
  const int col = t % cot; t /= cot;
  const int cb = t % ck; t /= ck;
  const int tap = t % (ks * ks); t /= ks * ks;
  const int ch = t % n_chunks;
  const int ct = t / n_chunks;
  const int c = ch * ck + cb;
  int sel;
  const int row = map_row(ct * cot + col, epi, hid, cout, &sel);
  float v = 0.f;
  if (row >= 0 && c < cin) {
    const int ky = tap / ks, kx = tap - ky * ks;
    const float* src = sel ? w2 : w;
    if (w_kind == ESS_W_CONV) {
      v = src[(((size_t)row * cin + c) * ks + ky) * ks + kx];
    } else {
      const int rows_src = (epi == ESS_EPI_GRU_UR) ? hid : cout;
      v = src[(((size_t)c * rows_src + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
    }
  }
  out[i] = v;
}

static __global__ void pack_rows_kernel(const float* v, const float* v2, float fill, float* out, int rows_padded, int epi,
                                 int hid, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_padded) return;
  int sel;
  const int row = map_row(i, epi, hid, cout, &sel);
  out[i] = row >= 0 ? (sel ? v2[row] : v[row]) : fill;
}


// bf16-MFMA variant (conv_bf16.hip)
int conv_bf16_launch(const EssConvDesc* d, const EssConvPlan& pl, const Geom& g, const ConvKArgs& a, hipStream_t st);
int conv_bf16_pack_weights(const EssConvDesc* d, const EssConvPlan& pl, int w_kind, const float* w, const float* w2, void* packed,
                           hipStream_t st, bool split = false, bool f16 = false);
int conv_bf16_pack_weights_multi(const EssConvDesc* descs, const int32_t* kinds, const float* const* w, void* const* packed, int count,
                                 hipStream_t st);
void conv_bf16_s2d_pick(const EssConvDesc* d, int* cw, int* tiles, int* tiles_x, int* tiles_y);
bool conv_bf16_s2d_preferred(const EssConvDesc* d);

}  // namespace essconv

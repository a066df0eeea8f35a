// Direct convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Same decomposition as conv_fwd.hip (A = weights, B = activations, 4 waves x 2 pixel blocks x MB channel blocks),
// with the MFMA K-step now 16 input channels: each lane feeds 8 consecutive channels of one pixel (B) / one output
// channel (A) as a 16-byte bf16x8 fragment.  Tensors stay NCHW fp32 in HBM; the conversion to bf16 and the
// channel-interleaving ("pixel vector" = 8 channels, 16 B) happen while the tile is staged into LDS, so one
// ds_read_b128 per fragment feeds the matrix core and consecutive lanes (= consecutive pixels) read consecutive
// 16-byte slots: conflict-free without a swizzle.  Weights are packed once to bf16 [tile][chunk][tap][c/8][cout][8].
// Staging is software-pipelined: the global loads of chunk i+1 are issued before the MFMA phase of chunk i and
// written to LDS after it, so HBM/L2 latency hides under the matrix work (one LDS buffer, two barriers per chunk).
#include "conv_bf16_common.h"

namespace {

using namespace essconv;

// H (SRCBF only): IEEE-half operands and 16-bit outputs (ESS_COMPUTE_F16)
template <int KS, int S, int MB, int EPI, int CB8, bool SRCBF = false, bool H = false>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvKArgs a) {
  static_assert(!H || SRCBF, "half operands come as F16_C8 tensors");
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  constexpr int COT = MB * 32;
  constexpr int CK = CB8 * 8;
  constexpr int KPC = kpc(KS, S);
  constexpr int WSZ = KS * KS * CB8 * COT;  // weight slab of one chunk, 16-byte units
  constexpr int WV = (WSZ + 255) / 256;
  constexpr bool WPRE = WV <= 5;           // small slabs ride in registers across the MFMA phase as well
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int BW = 1 << a.bwl, WX = 1 << a.wxl, RB = 32 >> a.bwl;
  const int TW = WX << a.bwl, TH = (4 >> a.wxl) * NBW * RB;
  // logical order: channel tile fastest, then spatial tile, then sample (see xcd_remap)
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = logical % a.n_cout_tiles;
  const int sp = logical / a.n_cout_tiles;
  const int tile = sp % a.n_tiles, n = sp / a.n_tiles;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  u32x4* in_t = smem16;
  u32x4* w_t = smem16 + CB8 * a.plane;

  const int ox = p & (BW - 1), oy = p >> a.bwl;
  const int wx = wave & (WX - 1), wy = wave >> a.wxl;
  const int lx = wx * BW + ox;
  int ly[NBW], boff[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    ly[nb] = (wy * NBW + nb) * RB + oy;
    boff[nb] = half * a.plane + ly[nb] * S * a.row_pitch + lx;
  }

  f32x16 acc[MB][NBW];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  // ---- staging plan.  The input tile of a chunk is CB8 blocks of IH*IW "pixel vectors" (8 channels of one position,
  // 16 B of bf16).  Thread t owns positions t, t+256, ... of EVERY block (lanes run along x: each of the 8 per-channel
  // loads of a vector is a coalesced row segment).  Loads go through bounds-checked buffer descriptors, one per
  // source and sample: everything that must read as zero (conv padding, zero-insert holes, channels past the end of a
  // source, positions past the tile) is given an out-of-range offset, so the loads carry NO branch and NO select --
  // a load under a per-lane condition makes hipcc wait vmcnt(0) inside every branch (one serialized memory round
  // trip per element), which was the whole cost of the first version of this kernel.
  const int iy0 = y0 * S - a.pad, ix0 = x0 * S - a.pad;
  const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
  const int Wp0 = a.Win >> sh0, Wp1 = a.Win >> sh1;
  const unsigned pl0 = (unsigned)((a.Hin >> sh0) * Wp0) * 4u, pl1 = (unsigned)((a.Hin >> sh1) * Wp1) * 4u;  // plane bytes
  const __amdgpu_buffer_rsrc_t r0 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.src0 + (size_t)n * a.C0 * (pl0 / 4)), 0, a.C0 * pl0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.C1 ? a.src1 + (size_t)n * a.C1 * (pl1 / 4) : a.src0), 0, a.C1 * pl1, 0x00020000);
  unsigned v_o0[KPC], v_o1[KPC];   // fp32 sources: byte offset inside a channel plane (OOB = reads zero)
  unsigned v_k0[KPC], v_k1[KPC];   // BF16_C8 sources: keep masks (v_o* then hold the clamped pixel index)
  int v_lds[KPC];
  const int npos = a.IH * a.IW;
#pragma unroll
  for (int k = 0; k < KPC; ++k) {
    const int vi = tid + k * 256;
    const int iy = vi / a.IW, ix = vi - iy * a.IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    const bool in = vi < npos && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    const bool odd = ((gy | gx) & 1) != 0;
    const bool in0 = in && !(a.mode0 == ESS_SRC_ZERO_UP2 && odd), in1 = in && !(a.mode1 == ESS_SRC_ZERO_UP2 && odd);
    v_lds[k] = vi < npos ? iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix) : -1;
    if constexpr (SRCBF) {
      v_o0[k] = in0 ? (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0)) : 0u;
      v_o1[k] = in1 ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) : 0u;
      v_k0[k] = in0 ? 0xffffffffu : 0u;
      v_k1[k] = in1 ? 0xffffffffu : 0u;
    } else {
      v_o0[k] = in0 ? (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0)) * 4u : OOB;
      v_o1[k] = in1 ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) * 4u : OOB;
    }
  }
  // 8 consecutive channels (block cb of chunk ch) at staged position k.  fp32 NCHW sources: 8 bounds-checked dword loads,
  // converted to a bf16x8 pixel vector at the LDS write.  BF16_C8 sources: the pixel vector IS the stored form -- one
  // 16-byte load from a clamped address, zeroed by a mask where the position is padding / a zero-insert hole / past the end.
  struct Raw8 { float v[8]; };
  struct Pre { typename std::conditional<SRCBF, u32x4, Raw8>::type d; };
  const size_t hw0 = pl0 / 4, hw1 = pl1 / 4;
  const int nb0 = (a.C0 + 7) >> 3, nb1 = (a.C1 + 7) >> 3;
  const u32x4* s0 = (const u32x4*)a.src0 + (size_t)n * nb0 * hw0;
  const u32x4* s1 = a.C1 ? (const u32x4*)a.src1 + (size_t)n * nb1 * hw1 : s0;
  auto load_vec = [&](int ch, int cb, int k) -> Pre {
    const int c0 = ch * CK + cb * 8;                  // wave-uniform
    const bool first = c0 < a.C0 || a.C1 == 0;        // a block never straddles the sources (C0 % 8 == 0)
    Pre r;
    if constexpr (SRCBF) {
      const int bi = (first ? c0 : c0 - a.C0) >> 3, nbs = first ? nb0 : nb1;
      const u32x4* sp = (first ? s0 : s1) + (size_t)(bi < nbs ? bi : 0) * (first ? hw0 : hw1);
      r.d = sp[first ? v_o0[k] : v_o1[k]];
    } else {
      const unsigned pls = first ? pl0 : pl1;
      const unsigned cbase = (unsigned)(first ? c0 : c0 - a.C0) * pls;
      const unsigned off = (first ? v_o0[k] : v_o1[k]) + cbase;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        r.d.v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? r0 : r1, (int)(off + j * pls), 0, 0));
    }
    return r;
  };
  // split operands (a.split, ESS_COMPUTE_BF16X3; fp32 sources only): the K loop runs over 3 * n_chunks VIRTUAL chunks -- chunk vc / 3
  // staged as (w_hi, x_hi), (w_hi, x_lo), (w_lo, x_hi) from ONE load of its fp32 values (conv_bf16_ws.hip has the same scheme)
  auto to_lds = [&](int ch, int cb, int k, const Pre& r, bool x_lo = false) -> u32x4 {
    if constexpr (SRCBF) {
      const int c0 = ch * CK + cb * 8;
      const bool first = c0 < a.C0 || a.C1 == 0;
      const unsigned blk_ok = ((first ? c0 : c0 - a.C0) >> 3) < (first ? nb0 : nb1) ? 0xffffffffu : 0u;
      const unsigned m = (first ? v_k0[k] : v_k1[k]) & blk_ok;
      u32x4 v = r.d;
      v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
      return v;
    } else {
      if (x_lo) {
        float lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = r.d.v[j] - (float)(__bf16)r.d.v[j];
        return pack8(lo);
      }
      return pack8(r.d.v);
    }
  };

  // raw values stay in registers across the MFMA phase; the bf16 conversion / masking happens at the LDS write so that
  // nothing waits on these loads before the matrix work has been issued
  Pre pre[CB8][KPC];
  u32x4 wpre[WPRE ? WV : 1];
  const bool split = !SRCBF && a.split;
  const int nvc = split ? 3 * a.n_chunks : a.n_chunks;
  const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ * (split ? 2 : 1);
  // weight slab of virtual chunk vc: [chunk][hi | lo] with split operands (lo for the third pass)
  auto wslab = [&](int vc) { return split ? 2 * (vc / 3) + ((vc % 3) == 2 ? 1 : 0) : vc; };
#pragma unroll
  for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
    for (int k = 0; k < KPC; ++k) pre[cb][k] = load_vec(0, cb, k);
  if constexpr (WPRE) {
#pragma unroll
    for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wbase[i < WSZ ? i : 0]; }
  }

  for (int vc = 0; vc < nvc; ++vc) {
    const int ch = split ? vc / 3 : vc;
    const bool x_lo = split && (vc - 3 * ch) == 1;
    __syncthreads();  // previous chunk's fragments have been read
#pragma unroll
    for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
      for (int k = 0; k < KPC; ++k)
        if (v_lds[k] >= 0) in_t[cb * a.plane + v_lds[k]] = to_lds(ch, cb, k, pre[cb][k], x_lo);
    if constexpr (WPRE) {
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wpre[it]; }
    } else {
      const u32x4* wsrc = wbase + (size_t)wslab(vc) * WSZ;
      u32x4 wv[WV];
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wv[it] = wsrc[i < WSZ ? i : 0]; }
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wv[it]; }
    }
    __syncthreads();
    // prefetch the next chunk; the loads land while the matrix cores work on this one
    if (vc + 1 < nvc) {
      if (!split || (vc + 1) % 3 == 0) {  // (split: the fp32 values in `pre` serve the three virtual chunks of their chunk)
#pragma unroll
        for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
          for (int k = 0; k < KPC; ++k) pre[cb][k] = load_vec(split ? (vc + 1) / 3 : vc + 1, cb, k);
      }
      if constexpr (WPRE) {
        const u32x4* wsrc = wbase + (size_t)wslab(vc + 1) * WSZ;
#pragma unroll
        for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wsrc[i < WSZ ? i : 0]; }
      }
    }
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
        const int toff = ky * a.row_pitch + (S == 2 ? (kx & 1) * a.par_off + (kx >> 1) : kx);
        const u32x4* wp = w_t + (tap * CB8 + half) * COT + p;
        const u32x4* ip = in_t + toff;
#pragma unroll
        for (int kk = 0; kk < CB8; kk += 2) {
          u32x4 af[MB], bfr[NBW];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) af[mb] = wp[kk * COT + mb * 32];
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) bfr[nb] = ip[boff[nb] + kk * a.plane];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
              acc[mb][nb] = ess_mfma16<H>(af[mb], bfr[nb], acc[mb][nb]);
        }
      }
    }
  }
  conv_epilogue<MB, EPI, true, H>(a, acc, ct, n, half, x0 + lx, y0, ly);
}

template <int KS, int S, int MB, int CB8>
void launch_epi(int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (KS == 3 && S == 1) {
    switch (epi) {
      case ESS_EPI_LSTM: { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_LSTM, CB8>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_LSTM, CB8>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_UR: { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_UR, CB8>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_UR, CB8>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_OUT: { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_OUT, CB8>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_OUT, CB8>), grid, dim3(256), lds, st, a); } return;
      default: break;
    }
  }
  { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR, CB8>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR, CB8>), grid, dim3(256), lds, st, a); }
}

// BF16_C8 sources through the generic tile kernel (LINEAR epilogue): 1x1 (stride 1 / 2) and 3x3 / stride 2 -- the
// trainable networks' 1x1 head, ResNet downsample convs and their data-gradients
template <int KS, int S, int MB, int CB8>
void launch_c8(dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (a.f16) {  // ESS_COMPUTE_F16
    ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR, CB8, true, true>, lds);
    hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR, CB8, true, true>), grid, dim3(256), lds, st, a);
    return;
  }
  ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR, CB8, true>, lds);
  hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR, CB8, true>), grid, dim3(256), lds, st, a);
}
template <int KS, int S>
void launch_mb_c8(int mb, int cb8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (KS == 1 && S == 1) {
    if (cb8 == 4) {
      if (mb == 2) launch_c8<KS, S, 2, 4>(grid, lds, st, a);
      else launch_c8<KS, S, 1, 4>(grid, lds, st, a);
      return;
    }
  }
  if (mb == 2) launch_c8<KS, S, 2, 2>(grid, lds, st, a);
  else launch_c8<KS, S, 1, 2>(grid, lds, st, a);
}

template <int KS, int S>
void launch_mb(int mb, int cb8, int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (KS == 1 && S == 1) {
    if (cb8 == 4) {
      if (mb == 2) launch_epi<KS, S, 2, 4>(epi, grid, lds, st, a);
      else launch_epi<KS, S, 1, 4>(epi, grid, lds, st, a);
      return;
    }
  }
  if (mb == 2) launch_epi<KS, S, 2, 2>(epi, grid, lds, st, a);
  else launch_epi<KS, S, 1, 2>(epi, grid, lds, st, a);
}
}  // namespace

namespace essconv {

void conv_bf16_launch_generic(int key, int mb, int cb8, int epi, bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (c8) {
    switch (key) {
      case 11: launch_mb_c8<1, 1>(mb, cb8, grid, lds, st, a); break;
      case 12: launch_mb_c8<1, 2>(mb, cb8, grid, lds, st, a); break;
      case 31: launch_mb_c8<3, 1>(mb, cb8, grid, lds, st, a); break;
      default: launch_mb_c8<3, 2>(mb, cb8, grid, lds, st, a); break;
    }
    return;
  }
  switch (key) {
    case 11: launch_mb<1, 1>(mb, cb8, epi, grid, lds, st, a); break;
    case 12: launch_mb<1, 2>(mb, cb8, epi, grid, lds, st, a); break;
    case 31: launch_mb<3, 1>(mb, cb8, epi, grid, lds, st, a); break;
    case 32: launch_mb<3, 2>(mb, cb8, epi, grid, lds, st, a); break;
    case 51: launch_mb<5, 1>(mb, cb8, epi, grid, lds, st, a); break;
    case 52: launch_mb<5, 2>(mb, cb8, epi, grid, lds, st, a); break;
    case 71: launch_mb<7, 1>(mb, cb8, epi, grid, lds, st, a); break;
    default: launch_mb<7, 2>(mb, cb8, epi, grid, lds, st, a); break;
  }
}

}  // namespace essconv

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
#include "conv_common.h"
#include <stdlib.h>

namespace {

using namespace essconv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
  return __builtin_bit_cast(u32x4, b);
}

// pixel positions a thread stages per 8-channel block (compile-time bound of the register prefetch), by filter geometry
constexpr int kpc(int ks, int s) { return stage_kpc(ks, s); }
constexpr unsigned OOB = 0x80000000u;  // beyond any buffer: the bounds-checked load returns 0

template <int KS, int S, int MB, int EPI, int CB8>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvKArgs a) {
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
  unsigned v_o0[KPC], v_o1[KPC];
  int v_lds[KPC];
  const int npos = a.IH * a.IW;
#pragma unroll
  for (int k = 0; k < KPC; ++k) {
    const int vi = tid + k * 256;
    const int iy = vi / a.IW, ix = vi - iy * a.IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    const bool in = vi < npos && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    const bool odd = ((gy | gx) & 1) != 0;
    v_lds[k] = vi < npos ? iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix) : -1;
    v_o0[k] = (in && !(a.mode0 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0)) * 4u : OOB;
    v_o1[k] = (in && !(a.mode1 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) * 4u : OOB;
  }
  // 8 consecutive channels (block cb of chunk ch) at staged position k -> bf16x8
  struct Raw8 { float v[8]; };
  auto load_vec = [&](int ch, int cb, int k) -> Raw8 {
    const int c0 = ch * CK + cb * 8;                  // wave-uniform
    const bool first = c0 < a.C0 || a.C1 == 0;        // a block never straddles the sources (C0 % 8 == 0)
    const unsigned pls = first ? pl0 : pl1;
    const unsigned cbase = (unsigned)(first ? c0 : c0 - a.C0) * pls;
    const unsigned off = (first ? v_o0[k] : v_o1[k]) + cbase;
    Raw8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r.v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? r0 : r1, (int)(off + j * pls), 0, 0));
    return r;
  };

  // raw fp32 values stay in registers across the MFMA phase; the bf16 conversion happens at the LDS write so that
  // nothing waits on these loads before the matrix work has been issued
  Raw8 pre[CB8][KPC];
  u32x4 wpre[WPRE ? WV : 1];
  const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ;
#pragma unroll
  for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
    for (int k = 0; k < KPC; ++k) pre[cb][k] = load_vec(0, cb, k);
  if constexpr (WPRE) {
#pragma unroll
    for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wbase[i < WSZ ? i : 0]; }
  }

  for (int ch = 0; ch < a.n_chunks; ++ch) {
    __syncthreads();  // previous chunk's fragments have been read
#pragma unroll
    for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
      for (int k = 0; k < KPC; ++k)
        if (v_lds[k] >= 0) in_t[cb * a.plane + v_lds[k]] = pack8(pre[cb][k].v);
    if constexpr (WPRE) {
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wpre[it]; }
    } else {
      const u32x4* wsrc = wbase + (size_t)ch * WSZ;
      u32x4 wv[WV];
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wv[it] = wsrc[i < WSZ ? i : 0]; }
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wv[it]; }
    }
    __syncthreads();
    // prefetch the next chunk; the loads land while the matrix cores work on this one
    if (ch + 1 < a.n_chunks) {
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
        for (int k = 0; k < KPC; ++k) pre[cb][k] = load_vec(ch + 1, cb, k);
      if constexpr (WPRE) {
        const u32x4* wsrc = wbase + (size_t)(ch + 1) * WSZ;
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
          bf16x8 af[MB], bfr[NBW];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) af[mb] = __builtin_bit_cast(bf16x8, wp[kk * COT + mb * 32]);
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) bfr[nb] = __builtin_bit_cast(bf16x8, ip[boff[nb] + kk * a.plane]);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
        }
      }
    }
  }
  conv_epilogue<MB, EPI>(a, acc, ct, n, half, x0 + lx, y0, ly);
}

__global__ void pack_weights_bf16_kernel(const float* w, const float* w2, __bf16* out, int64_t total, int cot, int ck,
                                         int n_chunks, int ks, int cin, int cout, int epi, int hid, int w_kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t t = i;
  const int kp = t & 7; t >>= 3;
  const int col = t % cot; t /= cot;
  const int cb8 = ck >> 3;
  const int cb = t % cb8; t /= cb8;
  const int tap = t % (ks * ks); t /= ks * ks;
  const int ch = t % n_chunks;
  const int ct = t / n_chunks;
  const int c = ch * ck + cb * 8 + kp;
  int sel;
  const int row = map_row(ct * cot + col, epi, hid, cout, &sel);
  float v = 0.f;
  if (row >= 0 && c < cin) {
    const int ky = tap / ks, kx = tap - ky * ks;
    const float* src = sel ? w2 : w;
    if (w_kind == ESS_W_CONV) v = src[(((size_t)row * cin + c) * ks + ky) * ks + kx];
    else v = src[(((size_t)c * cout + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  out[i] = (__bf16)v;
}


// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for 3x3 / stride 1 (the bulk of the FLOPs).  512 threads: waves 0-3 only issue LDS reads
// and MFMAs (consumers), waves 4-7 only stage (producers: bounds-checked loads -> bf16 -> LDS).  The dispatcher places
// wave w on SIMD w % 4, so every SIMD hosts one consumer and one producer: the staging VALU/LDS-write work runs in the
// shadow of the matrix pipe instead of in front of it.  LDS is double-buffered; ONE barrier per channel chunk:
//   iteration ch: consumers read buffer ch&1 | producers convert+write chunk ch+1 into buffer (ch+1)&1 (its last
//   readers passed the previous barrier) and then issue the loads of chunk ch+2, which land during iteration ch+1.
template <int MB, int EPI, bool SRCBF>
__global__ __launch_bounds__(512, MB == 4 ? 2 : 4) void conv_bf16_ws_k3s1_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  constexpr int KS = 3, CB8 = 2, CK = 16;
  constexpr int COT = MB * 32;
  constexpr int KPC = kpc(3, 1);
  constexpr int WSZ = KS * KS * CB8 * COT;
  constexpr int WV = (WSZ + 255) / 256;
  const int role = threadIdx.x >> 8;  // 0: consumer (MFMA), 1: producer (staging) -- wave-uniform
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int BW = 1 << a.bwl, WX = 1 << a.wxl, RB = 32 >> a.bwl;
  const int TW = WX << a.bwl, TH = (4 >> a.wxl) * NBW * RB;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = logical % a.n_cout_tiles;
  const int sp = logical / a.n_cout_tiles;
  const int tile = sp % a.n_tiles, n = sp / a.n_tiles;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int bufsz = CB8 * a.plane + WSZ;  // one stage: input tile + weight slab (16-byte units)

  if (role == 1 && SRCBF) {
    // ------------------------------------------------------------------ producer, BF16_C8 sources
    // The sources are already bf16 pixel vectors ([N][C/8][H][W][8]): staging one is ONE 16-byte load and ONE
    // ds_write_b128, no conversion.  A halo row of the tile is 34 x 16 B contiguous, so an 8-channel block costs ~5
    // cache lines per row instead of 8 x 3 with fp32 NCHW planes -- the L1 line rate, not HBM, bounded the fp32 staging.
    // Padding / overhang positions read a clamped address and are zeroed by a mask; tail channels are zero in memory.
    const int iy0 = y0 - 1, ix0 = x0 - 1;
    const size_t hw = (size_t)a.Hin * a.Win;
    const int nb0 = (a.C0 + 7) >> 3, nb1 = (a.C1 + 7) >> 3;
    const u32x4* s0 = (const u32x4*)a.src0 + (size_t)n * nb0 * hw;
    const u32x4* s1 = a.C1 ? (const u32x4*)a.src1 + (size_t)n * nb1 * hw : s0;
    unsigned v_pos[KPC], v_keep[KPC];
    int v_lds[KPC];
    const int npos = a.IH * a.IW;
#pragma unroll
    for (int k = 0; k < KPC; ++k) {
      const int vi = tid + k * 256;
      const int iy = vi / a.IW, ix = vi - iy * a.IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool in = vi < npos && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
      v_lds[k] = vi < npos ? iy * a.row_pitch + ix : -1;
      v_pos[k] = in ? (unsigned)(gy * a.Win + gx) : 0u;
      v_keep[k] = in ? 0xffffffffu : 0u;
    }
    u32x4 pre[CB8][KPC];
    u32x4 wpre[WV];
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ;
    auto load_chunk = [&](int ch) {
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb) {
        const int c0 = ch * CK + cb * 8;                 // wave-uniform
        const bool first = c0 < a.C0 || a.C1 == 0;       // a block never straddles the sources (C0 % 8 == 0)
        const int bi = (first ? c0 : c0 - a.C0) >> 3, nbs = first ? nb0 : nb1;
        const u32x4* sp = (first ? s0 : s1) + (size_t)(bi < nbs ? bi : 0) * hw;  // blocks past the end: clamped, masked
#pragma unroll
        for (int k = 0; k < KPC; ++k) pre[cb][k] = sp[v_pos[k]];
      }
      const u32x4* wsrc = wbase + (size_t)ch * WSZ;
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wsrc[i < WSZ ? i : 0]; }
    };
    auto commit = [&](int ch, int buf) {
      u32x4* in_t = smem16 + buf * bufsz;
      u32x4* w_t = in_t + CB8 * a.plane;
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb) {
        const int c0 = ch * CK + cb * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const unsigned blk_ok = ((first ? c0 : c0 - a.C0) >> 3) < (first ? nb0 : nb1) ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < KPC; ++k) {
          const unsigned m = v_keep[k] & blk_ok;
          u32x4 v = pre[cb][k];
          v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
          if (v_lds[k] >= 0) in_t[cb * a.plane + v_lds[k]] = v;
        }
      }
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wpre[it]; }
    };
    load_chunk(0);
    commit(0, 0);
    if (a.n_chunks > 1) load_chunk(1);
    __syncthreads();  // stage 0 is ready
    for (int ch = 0; ch < a.n_chunks; ++ch) {
      if (ch + 1 < a.n_chunks) {
        commit(ch + 1, (ch + 1) & 1);
        if (ch + 2 < a.n_chunks) load_chunk(ch + 2);
      }
      __syncthreads();
    }
    return;
  }
  if (role == 1) {
    // ------------------------------------------------------------------------------------------- producer
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
    const int Wp0 = a.Win >> sh0, Wp1 = a.Win >> sh1;
    const unsigned pl0 = (unsigned)((a.Hin >> sh0) * Wp0) * 4u, pl1 = (unsigned)((a.Hin >> sh1) * Wp1) * 4u;
    const __amdgpu_buffer_rsrc_t r0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.src0 + (size_t)n * a.C0 * (pl0 / 4)), 0, a.C0 * pl0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.C1 ? a.src1 + (size_t)n * a.C1 * (pl1 / 4) : a.src0), 0, a.C1 * pl1, 0x00020000);
    unsigned v_o0[KPC], v_o1[KPC];
    int v_lds[KPC];
    const int npos = a.IH * a.IW;
#pragma unroll
    for (int k = 0; k < KPC; ++k) {
      const int vi = tid + k * 256;
      const int iy = vi / a.IW, ix = vi - iy * a.IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool in = vi < npos && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
      const bool odd = ((gy | gx) & 1) != 0;
      v_lds[k] = vi < npos ? iy * a.row_pitch + ix : -1;
      v_o0[k] = (in && !(a.mode0 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0)) * 4u : OOB;
      v_o1[k] = (in && !(a.mode1 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) * 4u : OOB;
    }
    struct Raw8 { float v[8]; };
    Raw8 pre[CB8][KPC];
    u32x4 wpre[WV];
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ;
    auto load_chunk = [&](int ch) {
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb) {
        const int c0 = ch * CK + cb * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const unsigned pls = first ? pl0 : pl1;
        const unsigned cbase = (unsigned)(first ? c0 : c0 - a.C0) * pls;
#pragma unroll
        for (int k = 0; k < KPC; ++k) {
          const unsigned off = (first ? v_o0[k] : v_o1[k]) + cbase;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            pre[cb][k].v[j] =
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? r0 : r1, (int)(off + j * pls), 0, 0));
        }
      }
      const u32x4* wsrc = wbase + (size_t)ch * WSZ;
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wsrc[i < WSZ ? i : 0]; }
    };
    auto commit = [&](int buf) {
      u32x4* in_t = smem16 + buf * bufsz;
      u32x4* w_t = in_t + CB8 * a.plane;
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
        for (int k = 0; k < KPC; ++k)
          if (v_lds[k] >= 0) in_t[cb * a.plane + v_lds[k]] = pack8(pre[cb][k].v);
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wpre[it]; }
    };
    load_chunk(0);
    commit(0);
    if (a.n_chunks > 1) load_chunk(1);
    __syncthreads();  // stage 0 is ready
    for (int ch = 0; ch < a.n_chunks; ++ch) {
      if (ch + 1 < a.n_chunks) {
        commit((ch + 1) & 1);
        if (ch + 2 < a.n_chunks) load_chunk(ch + 2);
      }
      __syncthreads();
    }
    return;
  }
  // --------------------------------------------------------------------------------------------- consumer
  const int ox = p & (BW - 1), oy = p >> a.bwl;
  const int wx = wave & (WX - 1), wy = wave >> a.wxl;
  const int lx = wx * BW + ox;
  int ly[NBW], boff[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    ly[nb] = (wy * NBW + nb) * RB + oy;
    boff[nb] = half * a.plane + ly[nb] * a.row_pitch + lx;
  }
  f32x16 acc[MB][NBW];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  __syncthreads();  // stage 0 is ready
  for (int ch = 0; ch < a.n_chunks; ++ch) {
    const u32x4* in_t = smem16 + (ch & 1) * bufsz;
    const u32x4* w_t = in_t + CB8 * a.plane;
    // fragments of tap t+1 are read from LDS while the MFMAs of tap t issue (register double buffer, fully unrolled)
    bf16x8 af[2][MB], bfr[2][NBW];
    auto read_tap = [&](int tap, int slot) {
      const int ky = tap / KS, kx = tap - ky * KS;
      const u32x4* wp = w_t + (tap * CB8 + half) * COT + p;
      const u32x4* ip = in_t + ky * a.row_pitch + kx;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[slot][mb] = __builtin_bit_cast(bf16x8, wp[mb * 32]);
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) bfr[slot][nb] = __builtin_bit_cast(bf16x8, ip[boff[nb]]);
    };
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      if (tap + 1 < KS * KS) read_tap(tap + 1, (tap + 1) & 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap & 1][mb], bfr[tap & 1][nb], acc[mb][nb], 0, 0, 0);
    }
    __syncthreads();
  }
  conv_epilogue<MB, EPI>(a, acc, ct, n, half, x0 + lx, y0, ly);
}


// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for 5x5 filters (stride 1 and 2): "tap pairing".  With 25 taps a 16-channel chunk needs a
// 51 KB weight slab per stage, which leaves room for neither double buffering nor a second workgroup.  Here a chunk is
// 8 channels and the K = 16 of one MFMA is TWO TAPS x 8 channels: lanes 0-31 (k 0..7) carry tap 2p, lanes 32-63
// (k 8..15) tap 2p+1 -- for the B operand that is just a different LDS offset per half-wave, for A a different slab
// row (packed [tile][chunk][pair][half][cout][8]).  13 pairs cover the 25 taps (the 26th has zero weights: 4 % waste);
// a stage is one 8-channel input tile + 26.6 KB of weights, double-buffered like the 3x3 kernel, one barrier per chunk.
template <int KS, int S, int MB, bool SRCBF>
__global__ __launch_bounds__(512, (S == 2 && MB == 2) ? 2 : 4) void conv_bf16_ws_pair_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  constexpr int NT = KS * KS, NP = (NT + 1) / 2;
  constexpr int COT = MB * 32;
  constexpr int KPC = kpc(KS, S);
  constexpr int WSZ = NP * 2 * COT;
  constexpr int WV = (WSZ + 255) / 256;
  const int role = threadIdx.x >> 8;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int BW = 1 << a.bwl, WX = 1 << a.wxl, RB = 32 >> a.bwl;
  const int TW = WX << a.bwl, TH = (4 >> a.wxl) * NBW * RB;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = logical % a.n_cout_tiles;
  const int sp = logical / a.n_cout_tiles;
  const int tile = sp % a.n_tiles, n = sp / a.n_tiles;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int bufsz = a.plane + WSZ;  // one stage: 8-channel input tile + weight slab (16-byte units)

  if (role == 1) {
    // ------------------------------------------------------------------------------------------- producer
    const int iy0 = y0 * S - a.pad, ix0 = x0 * S - a.pad;
    const int npos = a.IH * a.IW;
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ;
    constexpr bool DEEP = S == 2;
    int v_lds[KPC];
    if constexpr (SRCBF) {
      const size_t hw = (size_t)a.Hin * a.Win;
      const int nb0 = (a.C0 + 7) >> 3, nb1 = (a.C1 + 7) >> 3;
      const u32x4* s0 = (const u32x4*)a.src0 + (size_t)n * nb0 * hw;
      const u32x4* s1 = a.C1 ? (const u32x4*)a.src1 + (size_t)n * nb1 * hw : s0;
      unsigned v_pos[KPC], v_keep[KPC];
#pragma unroll
      for (int k = 0; k < KPC; ++k) {
        const int vi = tid + k * 256;
        const int iy = vi / a.IW, ix = vi - iy * a.IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool in = vi < npos && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        v_lds[k] = vi < npos ? iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix) : -1;
        v_pos[k] = in ? (unsigned)(gy * a.Win + gx) : 0u;
        v_keep[k] = in ? 0xffffffffu : 0u;
      }
      // DEEP (stride 2: one workgroup per CU, nothing else hides the memory latency): two chunks of loads in flight in
      // two register sets -- a load issued in iteration ch is committed in iteration ch+2
      struct Set { u32x4 pre[KPC]; u32x4 wpre[WV]; };
      Set sa, sb;
      auto load_chunk = [&](int ch, Set& r) {
        const int c0 = ch * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const int bi = (first ? c0 : c0 - a.C0) >> 3, nbs = first ? nb0 : nb1;
        const u32x4* sp8 = (first ? s0 : s1) + (size_t)(bi < nbs ? bi : 0) * hw;
#pragma unroll
        for (int k = 0; k < KPC; ++k) r.pre[k] = sp8[v_pos[k]];
        const u32x4* wsrc = wbase + (size_t)ch * WSZ;
#pragma unroll
        for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; r.wpre[it] = wsrc[i < WSZ ? i : 0]; }
      };
      auto commit = [&](int ch, int buf, const Set& r) {
        u32x4* in_t = smem16 + buf * bufsz;
        u32x4* w_t = in_t + a.plane;
        const int c0 = ch * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const unsigned blk_ok = ((first ? c0 : c0 - a.C0) >> 3) < (first ? nb0 : nb1) ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < KPC; ++k) {
          const unsigned m = v_keep[k] & blk_ok;
          u32x4 v = r.pre[k];
          v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
          if (v_lds[k] >= 0) in_t[v_lds[k]] = v;
        }
#pragma unroll
        for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = r.wpre[it]; }
      };
      const int nch = a.n_chunks;
      if constexpr (DEEP) {
        load_chunk(0, sa);
        if (nch > 1) load_chunk(1, sb);
        commit(0, 0, sa);
        if (nch > 2) load_chunk(2, sa);
        __syncthreads();  // stage 0 is ready
        for (int ch = 0; ch < nch; ch += 2) {
          if (ch + 1 < nch) {
            commit(ch + 1, 1, sb);
            if (ch + 3 < nch) load_chunk(ch + 3, sb);
          }
          __syncthreads();
          if (ch + 1 < nch) {
            if (ch + 2 < nch) {
              commit(ch + 2, 0, sa);
              if (ch + 4 < nch) load_chunk(ch + 4, sa);
            }
            __syncthreads();
          }
        }
      } else {
        load_chunk(0, sa);
        commit(0, 0, sa);
        if (nch > 1) load_chunk(1, sa);
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) {
          if (ch + 1 < nch) {
            commit(ch + 1, (ch + 1) & 1, sa);
            if (ch + 2 < nch) load_chunk(ch + 2, sa);
          }
          __syncthreads();
        }
      }
    } else {
      const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
      const int Wp0 = a.Win >> sh0, Wp1 = a.Win >> sh1;
      const unsigned pl0 = (unsigned)((a.Hin >> sh0) * Wp0) * 4u, pl1 = (unsigned)((a.Hin >> sh1) * Wp1) * 4u;
      const __amdgpu_buffer_rsrc_t r0 =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.src0 + (size_t)n * a.C0 * (pl0 / 4)), 0, a.C0 * pl0, 0x00020000);
      const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.C1 ? a.src1 + (size_t)n * a.C1 * (pl1 / 4) : a.src0), 0, a.C1 * pl1, 0x00020000);
      unsigned v_o0[KPC], v_o1[KPC];
#pragma unroll
      for (int k = 0; k < KPC; ++k) {
        const int vi = tid + k * 256;
        const int iy = vi / a.IW, ix = vi - iy * a.IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool in = vi < npos && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        const bool odd = ((gy | gx) & 1) != 0;
        v_lds[k] = vi < npos ? iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix) : -1;
        v_o0[k] = (in && !(a.mode0 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0)) * 4u : OOB;
        v_o1[k] = (in && !(a.mode1 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) * 4u : OOB;
      }
      struct Raw8 { float v[8]; };
      Raw8 pre[KPC];
      u32x4 wpre[WV];
      auto load_chunk = [&](int ch) {
        const int c0 = ch * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const unsigned pls = first ? pl0 : pl1;
        const unsigned cbase = (unsigned)(first ? c0 : c0 - a.C0) * pls;
#pragma unroll
        for (int k = 0; k < KPC; ++k) {
          const unsigned off = (first ? v_o0[k] : v_o1[k]) + cbase;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            pre[k].v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? r0 : r1, (int)(off + j * pls), 0, 0));
        }
        const u32x4* wsrc = wbase + (size_t)ch * WSZ;
#pragma unroll
        for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wsrc[i < WSZ ? i : 0]; }
      };
      auto commit = [&](int buf) {
        u32x4* in_t = smem16 + buf * bufsz;
        u32x4* w_t = in_t + a.plane;
#pragma unroll
        for (int k = 0; k < KPC; ++k)
          if (v_lds[k] >= 0) in_t[v_lds[k]] = pack8(pre[k].v);
#pragma unroll
        for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wpre[it]; }
      };
      load_chunk(0);
      commit(0);
      if (a.n_chunks > 1) load_chunk(1);
      __syncthreads();
      for (int ch = 0; ch < a.n_chunks; ++ch) {
        if (ch + 1 < a.n_chunks) {
          commit((ch + 1) & 1);
          if (ch + 2 < a.n_chunks) load_chunk(ch + 2);
        }
        __syncthreads();
      }
    }
    return;
  }
  // --------------------------------------------------------------------------------------------- consumer
  const int ox = p & (BW - 1), oy = p >> a.bwl;
  const int wx = wave & (WX - 1), wy = wave >> a.wxl;
  const int lx = wx * BW + ox;
  int ly[NBW], boff[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    ly[nb] = (wy * NBW + nb) * RB + oy;
    boff[nb] = ly[nb] * S * a.row_pitch + lx;
  }
  // LDS offset of tap t (wave-uniform: lives in SGPRs); each half-wave picks its tap of the pair with one select
  auto tap_off = [&](int t) {
    if (t >= NT) t = 0;  // the padding tap of the last pair: zero weights, any valid address
    const int ky = t / KS, kx = t - ky * KS;
    return ky * a.row_pitch + (S == 2 ? (kx & 1) * a.par_off + (kx >> 1) : kx);
  };
  f32x16 acc[MB][NBW];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  __syncthreads();  // stage 0 is ready
  for (int ch = 0; ch < a.n_chunks; ++ch) {
    const u32x4* in_t = smem16 + (ch & 1) * bufsz;
    const u32x4* w_t = in_t + a.plane;
    bf16x8 af[2][MB], bfr[2][NBW];
    auto read_pair = [&](int pr, int slot) {
      const u32x4* wp = w_t + (pr * 2 + half) * COT + p;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[slot][mb] = __builtin_bit_cast(bf16x8, wp[mb * 32]);
      const int toff = half ? tap_off(2 * pr + 1) : tap_off(2 * pr);
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) bfr[slot][nb] = __builtin_bit_cast(bf16x8, in_t[toff + boff[nb]]);
    };
    read_pair(0, 0);
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      if (pr + 1 < NP) read_pair(pr + 1, (pr + 1) & 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pr & 1][mb], bfr[pr & 1][nb], acc[mb][nb], 0, 0, 0);
    }
    __syncthreads();
  }
  conv_epilogue<MB, ESS_EPI_LINEAR>(a, acc, ct, n, half, x0 + lx, y0, ly);
}

// weights for the tap-paired kernel: [tile][chunk of 8 channels][pair][half][cout][8]
__global__ void pack_weights_bf16_pair_kernel(const float* w, __bf16* out, int64_t total, int cot, int n_chunks, int ks, int cin,
                                              int cout, int w_kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int nt = ks * ks, np = (nt + 1) / 2;
  int64_t t = i;
  const int kp = t & 7; t >>= 3;
  const int col = t % cot; t /= cot;
  const int hf = t & 1; t >>= 1;
  const int pr = t % np; t /= np;
  const int ch = t % n_chunks;
  const int ct = t / n_chunks;
  const int tap = 2 * pr + hf, c = ch * 8 + kp, row = ct * cot + col;
  float v = 0.f;
  if (tap < nt && row < cout && c < cin) {
    const int ky = tap / ks, kx = tap - ky * ks;
    if (w_kind == ESS_W_CONV) v = w[(((size_t)row * cin + c) * ks + ky) * ks + kx];
    else v = w[(((size_t)c * cout + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  out[i] = (__bf16)v;
}

template <int MB, bool SRCBF>
void launch_ws(int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
#define ESS_WS(E_) { ess_allow_lds(conv_bf16_ws_k3s1_kernel<MB, E_, SRCBF>, lds); hipLaunchKernelGGL((conv_bf16_ws_k3s1_kernel<MB, E_, SRCBF>), grid, dim3(512), lds, st, a); }
  switch (epi) {
    case ESS_EPI_LSTM: ESS_WS(ESS_EPI_LSTM) break;
    case ESS_EPI_GRU_UR: ESS_WS(ESS_EPI_GRU_UR) break;
    case ESS_EPI_GRU_OUT: ESS_WS(ESS_EPI_GRU_OUT) break;
    default: ESS_WS(ESS_EPI_LINEAR) break;
  }
#undef ESS_WS
}
template <bool SRCBF>
void launch_ws_mb(int mb, int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (mb == 4) launch_ws<4, SRCBF>(epi, grid, lds, st, a);
  else if (mb == 2) launch_ws<2, SRCBF>(epi, grid, lds, st, a);
  else launch_ws<1, SRCBF>(epi, grid, lds, st, a);
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

template <int S, int MB>
void launch_pair(bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (c8) {
    ess_allow_lds(conv_bf16_ws_pair_kernel<5, S, MB, true>, lds);
    hipLaunchKernelGGL((conv_bf16_ws_pair_kernel<5, S, MB, true>), grid, dim3(512), lds, st, a);
  } else {
    ess_allow_lds(conv_bf16_ws_pair_kernel<5, S, MB, false>, lds);
    hipLaunchKernelGGL((conv_bf16_ws_pair_kernel<5, S, MB, false>), grid, dim3(512), lds, st, a);
  }
}

}  // namespace

namespace essconv {

int conv_bf16_pack_weights(const EssConvDesc* d, const EssConvPlan& pl, int w_kind, const float* w, const float* w2, void* packed,
                           hipStream_t st) {
  const int64_t total = pl.packed_elems;
  if (is_paired(d)) {
    hipLaunchKernelGGL(pack_weights_bf16_pair_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, w, (__bf16*)packed, total,
                       pl.cout_tile, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, w_kind);
    return ess_launch_status("pack_weights_bf16(paired)");
  }
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, w, w2, (__bf16*)packed,
                     total, pl.cout_tile, pl.ck, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, d->epilogue, d->hidden, w_kind);
  return ess_launch_status("pack_weights_bf16");
}

// ---- many weight tensors in one launch (re-packing every trainable convolution after an optimiser step: ~65 tensors of a few
// hundred KB each, 5 us per single launch).  The job table travels in the kernel arguments; a block finds its job by a scan
// over the (wave-uniform) block ranges.  LINEAR layouts of pack_weights_bf16_kernel only.
struct PackJob {
  const float* w;
  __bf16* out;
  long long total;
  int blk_end;  // exclusive end of this job's block range
  int cot, ck, n_chunks, ks, cin, cout, w_kind;
};
constexpr int PACK_JOBS = 48;
struct PackJobs { PackJob j[PACK_JOBS]; int count; };

__global__ void pack_weights_bf16_multi_kernel(const PackJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.count && (int)blockIdx.x >= jobs.j[k].blk_end) ++k;
  const PackJob& jb = jobs.j[k];
  const int blk0 = k ? jobs.j[k - 1].blk_end : 0;
  const long long i = (long long)(blockIdx.x - blk0) * blockDim.x + threadIdx.x;
  if (i >= jb.total) return;
  long long t = i;
  const int kp = t & 7; t >>= 3;
  const int col = t % jb.cot; t /= jb.cot;
  const int cb8 = jb.ck >> 3;
  const int cb = t % cb8; t /= cb8;
  const int tap = t % (jb.ks * jb.ks); t /= jb.ks * jb.ks;
  const int ch = t % jb.n_chunks;
  const int ct = t / jb.n_chunks;
  const int c = ch * jb.ck + cb * 8 + kp;
  const int row = ct * jb.cot + col;  // LINEAR: packed row = output channel
  float v = 0.f;
  if (row < jb.cout && c < jb.cin) {
    const int ky = tap / jb.ks, kx = tap - ky * jb.ks;
    if (jb.w_kind == ESS_W_CONV) v = jb.w[(((size_t)row * jb.cin + c) * jb.ks + ky) * jb.ks + kx];
    else v = jb.w[(((size_t)c * jb.cout + row) * jb.ks + (jb.ks - 1 - ky)) * jb.ks + (jb.ks - 1 - kx)];
  }
  jb.out[i] = (__bf16)v;
}

// returns ESS_EINVAL (nothing launched) when a descriptor is not a plain bf16 LINEAR layout
int conv_bf16_pack_weights_multi(const EssConvDesc* descs, const int32_t* kinds, const float* const* w, void* const* packed, int count,
                                 hipStream_t st) {
  for (int i = 0; i < count; ++i) {
    int rc = validate(&descs[i]);
    if (rc) return rc;
    ESS_CHECK_ARG(is_bf16(&descs[i]) && descs[i].epilogue == ESS_EPI_LINEAR && !is_paired(&descs[i]) && w[i] && packed[i] &&
                      (kinds[i] == ESS_W_CONV || kinds[i] == ESS_W_TRANSPOSED),
                  "pack_weights_multi: job %d is not a plain bf16 LINEAR layout", i);
  }
  for (int i0 = 0; i0 < count; i0 += PACK_JOBS) {
    PackJobs jobs{};
    int blocks = 0;
    jobs.count = count - i0 < PACK_JOBS ? count - i0 : PACK_JOBS;
    for (int k = 0; k < jobs.count; ++k) {
      const EssConvDesc* d = &descs[i0 + k];
      EssConvPlan pl;
      make_plan(d, &pl);
      PackJob& jb = jobs.j[k];
      jb.w = w[i0 + k]; jb.out = (__bf16*)packed[i0 + k]; jb.total = pl.packed_elems;
      jb.cot = pl.cout_tile; jb.ck = pl.ck; jb.n_chunks = pl.n_chunks; jb.ks = d->ksize; jb.cin = d->C0 + d->C1; jb.cout = d->C_out;
      jb.w_kind = kinds[i0 + k];
      blocks += (int)ceil_div64(pl.packed_elems, 256);
      jb.blk_end = blocks;
    }
    hipLaunchKernelGGL(pack_weights_bf16_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, jobs);
  }
  return ess_launch_status("pack_weights_multi");
}

int conv_bf16_launch(const EssConvDesc* d, const EssConvPlan& pl, const Geom& g, const ConvKArgs& a, hipStream_t st) {
  ESS_CHECK_ARG(g.IH * g.IW <= kpc(d->ksize, d->stride) * 256, "conv(bf16): input tile of %d positions exceeds the staging capacity",
                g.IH * g.IW);
  ESS_CHECK_ARG((int64_t)(d->C0 > d->C1 ? d->C0 : d->C1) * d->H_in * d->W_in * 4 < (int64_t)1 << 31,
                "conv(bf16): one sample of a source must stay below 2 GiB (32-bit buffer offsets)");
  ESS_CHECK_ARG(d->C1 == 0 || (d->C0 % 8) == 0, "conv(bf16): the first source of a concat must have a multiple of 8 channels");
  const dim3 grid((unsigned)(g.tiles_x * g.tiles_y * pl.n_cout_tiles * d->N));
  const int mb = pl.cout_tile / 32;
  if (is_paired(d)) {  // 5x5: tap-paired wave-specialised kernel (the plan and the weight pack are specific to it)
    const size_t lds2 = 2 * (size_t)pl.lds_bytes;
    ESS_CHECK_ARG(lds2 <= 160 * 1024, "conv(bf16, 5x5): two stages of %d B exceed the 160 KiB LDS", pl.lds_bytes);
    const bool c8 = a.fmt0 == ESS_FMT_BF16_C8;
    if (c8) ESS_CHECK_ARG((((uintptr_t)a.src0 | (uintptr_t)a.src1) & 15) == 0, "conv(bf16): BF16_C8 sources must be 16-byte aligned");
    if (d->stride == 1) { if (mb == 2) launch_pair<1, 2>(c8, grid, lds2, st, a); else launch_pair<1, 1>(c8, grid, lds2, st, a); }
    else { if (mb == 2) launch_pair<2, 2>(c8, grid, lds2, st, a); else launch_pair<2, 1>(c8, grid, lds2, st, a); }
    return ess_launch_status("conv2d_forward(bf16, tap-paired)");
  }
  const bool use_ws = ws_enabled();
  if (use_ws && d->ksize == 3 && d->stride == 1 && pl.ck == 16) {
    const size_t lds2 = 2 * (size_t)pl.lds_bytes;  // double-buffered stages
    if (lds2 <= 160 * 1024) {
      if (a.fmt0 == ESS_FMT_BF16_C8) {
        ESS_CHECK_ARG((((uintptr_t)a.src0 | (uintptr_t)a.src1) & 15) == 0, "conv(bf16): BF16_C8 sources must be 16-byte aligned");
        launch_ws_mb<true>(mb, d->epilogue, grid, lds2, st, a);
      } else {
        launch_ws_mb<false>(mb, d->epilogue, grid, lds2, st, a);
      }
      return ess_launch_status("conv2d_forward(bf16, wave-specialised)");
    }
  }
  ESS_CHECK_ARG(a.fmt0 == ESS_FMT_F32_NCHW, "conv(bf16): BF16_C8 sources are only staged by the wave-specialised 3x3 kernel");
  const int key = d->ksize * 10 + d->stride;
  switch (key) {
    case 11: launch_mb<1, 1>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 12: launch_mb<1, 2>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 31: launch_mb<3, 1>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 32: launch_mb<3, 2>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 51: launch_mb<5, 1>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 52: launch_mb<5, 2>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 71: launch_mb<7, 1>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 72: launch_mb<7, 2>(mb, pl.ck / 8, d->epilogue, grid, pl.lds_bytes, st, a); break;
    default: ess_set_error("conv: no kernel for k%d s%d", d->ksize, d->stride); return ESS_ENOTSUP;
  }
  return ess_launch_status("conv2d_forward(bf16)");
}

}  // namespace essconv

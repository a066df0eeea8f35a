// Tap-paired wave-specialised 5x5 convolution on the bf16 matrix cores (see conv_bf16.hip for the dispatcher).
#include "conv_bf16_common.h"

namespace {

using namespace essconv;

template <int I, int N, class F>
__device__ __forceinline__ void pair_steps(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pair_steps<I + 1, N>(f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for 5x5 filters (stride 1 and 2): "tap pairing".  With 25 taps a 16-channel chunk needs a
// 51 KB weight slab per stage, which leaves room for neither double buffering nor a second workgroup.  Here a chunk is
// 8 channels and the K = 16 of one MFMA is TWO TAPS x 8 channels: lanes 0-31 (k 0..7) carry tap 2p, lanes 32-63
// (k 8..15) tap 2p+1 -- for the B operand that is just a different LDS offset per half-wave, for A a different slab
// row (packed [tile][chunk][pair][half][cout][8]).  13 pairs cover the 25 taps (the 26th has zero weights: 4 % waste);
// a stage is one 8-channel input tile + 26.6 KB of weights, double-buffered like the 3x3 kernel, one barrier per chunk.
// OUT8: BF16_C8 output(s) -- an instantiation that contains conv_epilogue_c8 and nothing of the fp32 epilogue variants (as in
// conv_bf16_ws.hip: the all-variants function carries ~30 k instructions of epilogues and their spills).
// H (SRCBF only): IEEE-half operands and 16-bit outputs (ESS_COMPUTE_F16)
template <int KS, int S, int MB, bool SRCBF, bool OUT8 = false, bool H = false>
__global__ __launch_bounds__(512, (S == 2 && MB == 2) ? 2 : 4) void conv_bf16_ws_pair_kernel(const ConvKArgs a) {
  static_assert(!H || SRCBF, "half operands come as F16_C8 tensors");
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
  // tile schedule: plain (one workgroup per tile) or persistent (a.persist) -- as in conv_bf16_ws.hip, one tile loop per role
  int t_start, t_count, t_first = 0, t_step = 1;
  if (a.persist) {
    const int total = a.n_cout_tiles * a.n_tiles * a.N, q = total >> 3, r = total & 7, xcd = (int)(blockIdx.x & 7);
    t_start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    t_count = q + (xcd < r ? 1 : 0);
    t_first = (int)(blockIdx.x >> 3);
    t_step = (int)(gridDim.x >> 3);
  } else {
    t_start = xcd_remap(blockIdx.x, gridDim.x);
    t_count = 1;
  }
#define ESS_TILE_LOOP for (int ti = t_first; ti < t_count; ti += t_step)
#define ESS_TILE_DECODE                                                    \
  const int logical = t_start + ti;                                        \
  const int ct = logical % a.n_cout_tiles;                                 \
  const int sp = logical / a.n_cout_tiles;                                 \
  const int tile = sp % a.n_tiles, n = sp / a.n_tiles;                     \
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;             \
  const int y0 = ty * TH, x0 = tx * TW;
  const int bufsz = a.plane + WSZ;  // one stage: 8-channel input tile + weight slab (16-byte units)

  if (role == 1) {
    ESS_TILE_LOOP {
    ESS_TILE_DECODE
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
      // split operands (a.split, ESS_COMPUTE_BF16X3): 3 * n_chunks virtual chunks, chunk vc / 3 staged as (w_hi, x_hi), (w_hi, x_lo),
      // (w_lo, x_hi) -- conv_bf16_ws.hip has the same scheme; the weight pack holds a hi and a lo slab per (tile, chunk)
      const int nvc = a.split ? 3 * a.n_chunks : a.n_chunks;
      const u32x4* wbase_s = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ * (a.split ? 2 : 1);
      auto load_chunk = [&](int vc) {
        const int ch = a.split ? vc / 3 : vc;
        const int w_lo = a.split && (vc - 3 * ch) == 2 ? 1 : 0;
        const int c0 = ch * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const unsigned pls = first ? pl0 : pl1;
        const unsigned cbase = (unsigned)(first ? c0 : c0 - a.C0) * pls;
        if (!a.split || vc == 3 * ch) {  // (split: one load per chunk serves its three virtual chunks, see conv_bf16_ws.hip)
#pragma unroll
        for (int k = 0; k < KPC; ++k) {
          const unsigned off = (first ? v_o0[k] : v_o1[k]) + cbase;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            pre[k].v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? r0 : r1, (int)(off + j * pls), 0, 0));
        }
        }
        const u32x4* wsrc = wbase_s + (size_t)(a.split ? 2 * ch + w_lo : ch) * WSZ;
#pragma unroll
        for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wsrc[i < WSZ ? i : 0]; }
      };
      auto commit = [&](int vc) {  // (the values `pre` holds belong to virtual chunk vc)
        u32x4* in_t = smem16 + (vc & 1) * bufsz;
        u32x4* w_t = in_t + a.plane;
        const bool x_lo = a.split && (vc % 3) == 1;
#pragma unroll
        for (int k = 0; k < KPC; ++k)
          if (v_lds[k] >= 0) {
            if (x_lo) {
              float lo[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) lo[j] = pre[k].v[j] - (float)(__bf16)pre[k].v[j];
              in_t[v_lds[k]] = pack8(lo);
            } else {
              in_t[v_lds[k]] = pack8(pre[k].v);
            }
          }
#pragma unroll
        for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wpre[it]; }
      };
      load_chunk(0);
      commit(0);
      if (nvc > 1) load_chunk(1);
      __syncthreads();
      for (int vc = 0; vc < nvc; ++vc) {
        if (vc + 1 < nvc) {
          commit(vc + 1);
          if (vc + 2 < nvc) load_chunk(vc + 2);
        }
        __syncthreads();
      }
    }
    }  // tile loop
    return;
  }
  // --------------------------------------------------------------------------------------------- consumer
  ESS_TILE_LOOP {
  ESS_TILE_DECODE
  int tid_t = (int)(threadIdx.x & 255);  // (opaque per tile: keeps the lane addressing from being hoisted out of the tile loop)
  asm volatile("" : "+v"(tid_t));
  const int tid = tid_t, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
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
  __builtin_amdgcn_s_setprio(1);  // (the matrix waves take issue priority over the staging waves of their SIMDs, as in conv_bf16_ws.hip)
  // ---- K loop.  The fragment reads run D = 2 (32-row tiles) or 1 tap pairs ahead of the MFMAs (D + 1 register sets, counted
  // lgkmcnt): a pair is only MB x NBW MFMAs = 64-128 cycles of matrix work per wave, about one LDS round trip.  As plain loads hipcc sinks every
  // ds_read to just above its MFMA and waits for it there with lgkmcnt(0) -- read, wait, MFMA, read, wait, MFMA: the matrix pipe
  // idles for an LDS round trip in front of every instruction (the first version of this loop; the kernel ran at 0.27 of peak).
  // So the reads and the waits are volatile asm statements, the wait taking the fragments as in/out operands (conv_bf16_ws.hip).
  const unsigned lds0 = (unsigned)(size_t)(smem16);
  const unsigned a_base = (unsigned)((a.plane + half * COT + p) * 16);
  unsigned b_base[NBW], tofs[NP];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) b_base[nb] = (unsigned)(boff[nb] * 16);
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) tofs[pr] = (unsigned)((half ? tap_off(2 * pr + 1) : tap_off(2 * pr)) * 16);
  struct Frags { u32x4 a[MB]; u32x4 b[NBW]; };
  constexpr int NR = MB + NBW;  // LDS reads per pair
  constexpr int D = MB == 1 ? 2 : 1;  // pairs the reads run ahead (64-row tiles: a pair is 128 cycles of MFMAs, and 128 registers hold two sets)
  const int nvc_c = a.split ? 3 * a.n_chunks : a.n_chunks;  // (split operands: three virtual chunks per chunk, see the staging waves)
  for (int ch = 0; ch < nvc_c; ++ch) {
    const unsigned stage_b = lds0 + (unsigned)((ch & 1) * bufsz * 16);
    const unsigned wa = stage_b + a_base;
    unsigned ba[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) ba[nb] = stage_b + b_base[nb];
    Frags f[D + 1];
    auto read_pair = [&](auto PR, Frags& F) {
      constexpr int pr = decltype(PR)::value;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const unsigned ad = wa + (unsigned)(mb * 32 * 16);  // (a local: clang refuses captured variables inside asm operands of a generic lambda)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F.a[mb]) : "v"(ad), "n"(pr * 2 * COT * 16));
      }
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const unsigned ad = ba[nb] + tofs[pr];
        asm volatile("ds_read_b128 %0, %1" : "=v"(F.b[nb]) : "v"(ad));
      }
    };
    auto wait_mma = [&](auto N, Frags& F) {
      constexpr int n_ = decltype(N)::value;
      if constexpr (MB == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(F.a[0]), "+v"(F.b[0]), "+v"(F.b[1]) : "n"(n_));
      else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(F.a[0]), "+v"(F.a[MB > 1 ? 1 : 0]), "+v"(F.b[0]), "+v"(F.b[1]) : "n"(n_));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
          acc[mb][nb] = ess_mfma16<H>(F.a[mb], F.b[nb], acc[mb][nb]);
    };
    pair_steps<0, D>([&](auto PR) { read_pair(PR, f[decltype(PR)::value]); });
    pair_steps<0, NP>([&](auto PR) {
      constexpr int pr = decltype(PR)::value;
      if constexpr (pr + D < NP) read_pair(std::integral_constant<int, pr + D>{}, f[(pr + D) % (D + 1)]);
      constexpr int ahead = (pr + D < NP) ? D : (NP - 1 - pr);  // pairs whose reads may still be in flight behind this one's
      wait_mma(std::integral_constant<int, ahead * NR>{}, f[pr % (D + 1)]);
    });
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(0);
  if constexpr (OUT8) conv_epilogue_c8<MB, H>(a, acc, ct, n, half, x0 + lx, y0, ly);
  else conv_epilogue<MB, ESS_EPI_LINEAR, false, H>(a, acc, ct, n, half, x0 + lx, y0, ly);
  }  // tile loop
#undef ESS_TILE_LOOP
#undef ESS_TILE_DECODE
}


template <int S, int MB>
void launch_pair(bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
#define ESS_PAIR(C8_, O8_) { ess_allow_lds(conv_bf16_ws_pair_kernel<5, S, MB, C8_, O8_>, lds); hipLaunchKernelGGL((conv_bf16_ws_pair_kernel<5, S, MB, C8_, O8_>), grid, dim3(512), lds, st, a); }
  const bool out8 = a.fmt_out == ESS_FMT_BF16_C8;
  if (c8 && a.f16) {  // ESS_COMPUTE_F16
#define ESS_PAIRH(O8_) { ess_allow_lds(conv_bf16_ws_pair_kernel<5, S, MB, true, O8_, true>, lds); hipLaunchKernelGGL((conv_bf16_ws_pair_kernel<5, S, MB, true, O8_, true>), grid, dim3(512), lds, st, a); }
    if (out8) ESS_PAIRH(true) else ESS_PAIRH(false)
#undef ESS_PAIRH
    return;
  }
  if (c8) { if (out8) ESS_PAIR(true, true) else ESS_PAIR(true, false) }
  else { if (out8) ESS_PAIR(false, true) else ESS_PAIR(false, false) }
#undef ESS_PAIR
}

}  // namespace

namespace essconv {

void conv_bf16_launch_pair(int stride, int mb, bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (stride == 1) { if (mb == 2) launch_pair<1, 2>(c8, grid, lds, st, a); else launch_pair<1, 1>(c8, grid, lds, st, a); }
  else { if (mb == 2) launch_pair<2, 2>(c8, grid, lds, st, a); else launch_pair<2, 1>(c8, grid, lds, st, a); }
}

}  // namespace essconv

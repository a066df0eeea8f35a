// Wave-specialised 3x3 / stride-1 convolution on the bf16 matrix cores (see conv_bf16.hip for the dispatcher).
#ifdef ESS_CV_TRACE
__device__ unsigned long long g_cv_trace[2048 * 8 * 48];
#define ESS_CT(i_) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) g_cv_trace[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 48 + (i_)] = __builtin_readcyclecounter(); } while (0)
#define ESS_EPI_STAMP(i_) ESS_CT(i_)
// wall clock (100 MHz, the same on every CU -- s_memtime is not): slot 46 at workgroup start, slot 43 at its end (non-LSTM kernels)
#define ESS_CW(i_) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) g_cv_trace[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 48 + (i_)] = wall_clock64(); } while (0)
// barrier with wait accounting: the cycles this wave spends inside the barrier are summed into `acc_`
#define ESS_SYNC_ACC(acc_) do { const unsigned long long t0_ = __builtin_readcyclecounter(); __syncthreads(); acc_ += __builtin_readcyclecounter() - t0_; } while (0)
#define ESS_CT_VAL(i_, v_) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) g_cv_trace[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 48 + (i_)] = (v_); } while (0)
#else
#define ESS_SYNC_ACC(acc_) __syncthreads()
#define ESS_CT_VAL(i_, v_) do { } while (0)
#define ESS_CW(i_) do { } while (0)
#define ESS_CT(i_) do { } while (0)
#endif
#include "conv_bf16_common.h"

namespace {

using namespace essconv;

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for 3x3 / stride 1 (the bulk of the FLOPs).  512 threads: waves 0-3 only issue LDS reads
// and MFMAs (consumers), waves 4-7 only stage (producers: bounds-checked loads -> bf16 -> LDS).  The dispatcher places
// wave w on SIMD w % 4, so every SIMD hosts one consumer and one producer: the staging VALU/LDS-write work runs in the
// shadow of the matrix pipe instead of in front of it.  LDS is double-buffered; ONE barrier per channel chunk:
//   iteration ch: consumers read buffer ch&1 | producers convert+write chunk ch+1 into buffer (ch+1)&1 (its last
//   readers passed the previous barrier) and then issue the loads of chunk ch+2, which land during iteration ch+1.
// OUT8: the output(s) are BF16_C8 tensors (LINEAR epilogue): an instantiation of its own that contains conv_epilogue_c8 and
// nothing of the fp32 epilogue variants -- the all-variants kernel is ~57k instructions with ~300 spilled registers in its
// epilogues, this one a tenth of that.
// H (SRCBF only): IEEE-half operands and 16-bit outputs (ESS_COMPUTE_F16)
template <int MB, int EPI, bool SRCBF, bool OUT8 = false, bool H = false>
__global__ __launch_bounds__(512, MB == 4 ? 2 : 4) void conv_bf16_ws_k3s1_kernel(const ConvKArgs a) {
  static_assert(!H || SRCBF, "half operands come as F16_C8 tensors");
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  constexpr int KS = 3, CB8 = 2, CK = 16;
  constexpr int COT = MB * 32;
  constexpr int KPC = kpc(3, 1);
  constexpr int WSZ = KS * KS * CB8 * COT;
  constexpr int WV = (WSZ + 255) / 256;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);  // 0: consumer (MFMA), 1: producer (staging): a scalar (wave-uniform) value
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int BW = 1 << a.bwl, WX = 1 << a.wxl, RB = 32 >> a.bwl;
  const int TW = WX << a.bwl, TH = (4 >> a.wxl) * NBW * RB;
  // Tile schedule.  Plain launch: one workgroup per tile, XCD-contiguous order (xcd_remap).  Persistent launch (a.persist: the grid
  // is one resident set of workgroups): workgroup b -- on XCD b % 8 by the dispatcher's round-robin -- walks every (grid / 8)-th
  // tile of its XCD's contiguous range, so neighbouring workgroups of an XCD still work on neighbouring tiles at the same time.
  // What the loop buys: after the last chunk's barrier the producer waves go straight on to the NEXT tile (index arithmetic, first
  // chunk's loads and LDS writes, second chunk's loads) while the matrix waves run this tile's epilogue -- the prologue of tile
  // t + 1 overlaps the epilogue of tile t inside one workgroup instead of relying on the other workgroup of the CU being out of phase.
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
  // (one tile loop PER ROLE: with a common loop hipcc hoists both roles' tile-invariant values above it and keeps the producers'
  // staging plan alive through the matrix waves' epilogue -- 59 spilled VGPRs in the 128-register instances)
#define ESS_TILE_LOOP for (int ti = t_first; ti < t_count; ti += t_step)
#define ESS_TILE_DECODE                                                    \
  const int logical = t_start + ti;                                        \
  const int ct = logical % a.n_cout_tiles;                                 \
  const int sp = logical / a.n_cout_tiles;                                 \
  const int tile = sp % a.n_tiles, n = sp / a.n_tiles;                     \
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;             \
  const int y0 = ty * TH, x0 = tx * TW;
  const int bufsz = CB8 * a.plane + WSZ;  // one stage: input tile + weight slab (16-byte units)
  ESS_CT(0);
  ESS_CW(46);

  if (role == 1 && SRCBF) {
    ESS_TILE_LOOP {
    ESS_TILE_DECODE
    // ------------------------------------------------------------------ producer, BF16_C8 sources
    // The sources are already bf16 pixel vectors ([N][C/8][H][W][8]): staging one is ONE 16-byte load and ONE
    // ds_write_b128, no conversion.  A halo row of the tile is 34 x 16 B contiguous, so an 8-channel block costs ~5
    // cache lines per row instead of 8 x 3 with fp32 NCHW planes -- the L1 line rate, not HBM, bounded the fp32 staging.
    // Padding / overhang positions read a clamped address and are zeroed by a mask; tail channels are zero in memory.
    // A source may be read nearest-x2-upsampled (the decoder's upsample + concat) or zero-inserted (data-gradient of a
    // stride-2 convolution): the position arithmetic is per source, the holes of ZERO_UP2 are masked like the padding.
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
    const int Wp0 = a.Win >> sh0, Wp1 = a.Win >> sh1;
    const size_t hw0 = (size_t)(a.Hin >> sh0) * Wp0, hw1 = (size_t)(a.Hin >> sh1) * Wp1;
    const int nb0 = (a.C0 + 7) >> 3, nb1 = (a.C1 + 7) >> 3;
    const u32x4* s0 = (const u32x4*)a.src0 + (size_t)n * nb0 * hw0;
    const u32x4* s1 = a.C1 ? (const u32x4*)a.src1 + (size_t)n * nb1 * hw1 : s0;
    unsigned v_pos0[KPC], v_pos1[KPC], v_keep0[KPC], v_keep1[KPC];
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
      v_lds[k] = vi < npos ? iy * a.row_pitch + ix : -1;
      v_pos0[k] = in0 ? (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0)) : 0u;
      v_pos1[k] = in1 ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) : 0u;
      v_keep0[k] = in0 ? 0xffffffffu : 0u;
      v_keep1[k] = in1 ? 0xffffffffu : 0u;
    }
    struct Set { u32x4 pre[CB8][KPC]; u32x4 wpre[WV]; };
    Set sa;
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ;
    auto load_chunk = [&](int ch, Set& r) {
#ifdef ESS_ABLATE
      if (a.deep & 4) return;  // (ablation build only, switch ESS_WS_ABL: no global loads)
#endif
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb) {
        const int c0 = ch * CK + cb * 8;                 // wave-uniform
        const bool first = c0 < a.C0 || a.C1 == 0;       // a block never straddles the sources (C0 % 8 == 0)
        const int bi = (first ? c0 : c0 - a.C0) >> 3, nbs = first ? nb0 : nb1;
        const u32x4* sp = (first ? s0 : s1) + (size_t)(bi < nbs ? bi : 0) * (first ? hw0 : hw1);  // blocks past the end: clamped, masked
#pragma unroll
        for (int k = 0; k < KPC; ++k) r.pre[cb][k] = sp[first ? v_pos0[k] : v_pos1[k]];
      }
      const u32x4* wsrc = wbase + (size_t)ch * WSZ;
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; r.wpre[it] = wsrc[i < WSZ ? i : 0]; }
    };
    auto commit = [&](int ch, int buf, const Set& r) {
#ifdef ESS_ABLATE
      if (a.deep & 16) return;  // (ablation build only: no LDS writes)
#endif
      u32x4* in_t = smem16 + buf * bufsz;
      u32x4* w_t = in_t + CB8 * a.plane;
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb) {
        const int c0 = ch * CK + cb * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const unsigned blk_ok = ((first ? c0 : c0 - a.C0) >> 3) < (first ? nb0 : nb1) ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < KPC; ++k) {
          const unsigned m = (first ? v_keep0[k] : v_keep1[k]) & blk_ok;
          u32x4 v = r.pre[cb][k];
          v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
          if (v_lds[k] >= 0) in_t[cb * a.plane + v_lds[k]] = v;
        }
      }
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = r.wpre[it]; }
    };
    const int nch = a.n_chunks;
    unsigned long long bar_acc = 0, wait_acc = 0, commit_acc = 0;
    (void)bar_acc; (void)wait_acc; (void)commit_acc;
    {
      ESS_CT(1);
      load_chunk(0, sa);
      commit(0, 0, sa);
      if (nch > 1) load_chunk(1, sa);
      ESS_CT(2);
      __syncthreads();  // stage 0 is ready
      for (int ch = 0; ch < nch; ++ch) {
        if (ch < 40) ESS_CT(3 + ch);
        if (ch + 1 < nch) {
#ifdef ESS_CV_TRACE
          const unsigned long long tc0 = __builtin_readcyclecounter();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned long long tc1 = __builtin_readcyclecounter();
          wait_acc += tc1 - tc0;
#endif
          commit(ch + 1, (ch + 1) & 1, sa);
#ifdef ESS_CV_TRACE
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long tc2 = __builtin_readcyclecounter();
          commit_acc += tc2 - tc1;
#endif
          if (ch + 2 < nch) load_chunk(ch + 2, sa);
        }
        if (ch == nch - 1) ESS_CT(44);
        ESS_SYNC_ACC(bar_acc);
      }
    }
    ESS_CT_VAL(42, bar_acc);
    ESS_CT_VAL(41, wait_acc);
    ESS_CT_VAL(40, commit_acc);
    ESS_CT(47);
    }  // tile loop
    return;
  }
  if (role == 1) {
    ESS_TILE_LOOP {
    ESS_TILE_DECODE
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
    // split operands (a.split, ESS_COMPUTE_BF16X3): the K loop runs over 3 * n_chunks VIRTUAL chunks -- chunk vc / 3 staged three
    // times as (w_hi, x_hi), (w_hi, x_lo), (w_lo, x_hi); hi = bf16(v), lo = bf16(v - hi).  The matrix waves do not know: same
    // fragments, same accumulators, three times the chunks.  The weight pack holds a hi and a lo slab per (tile, chunk).
    const int nvc = a.split ? 3 * a.n_chunks : a.n_chunks;
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * WSZ * (a.split ? 2 : 1);
    // (round 5: the fp32 values of a chunk are loaded ONCE and stay in `pre` for its three virtual chunks -- hi, lo and hi again are
    // three conversions of the same registers; the first version re-read them from L2 per virtual chunk, 12 bytes per element
    // through the dword-per-plane path that already bounded the fp32-source kernel)
    auto load_chunk = [&](int vc) {
      const int ch = a.split ? vc / 3 : vc;
      const int w_lo = a.split && (vc - 3 * ch) == 2 ? 1 : 0;
      if (!a.split || vc == 3 * ch) {
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
      }
      const u32x4* wsrc = wbase + (size_t)(a.split ? 2 * ch + w_lo : ch) * WSZ;
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wpre[it] = wsrc[i < WSZ ? i : 0]; }
    };
    auto commit = [&](int vc) {  // (the values `pre` holds belong to virtual chunk vc)
      u32x4* in_t = smem16 + (vc & 1) * bufsz;
      u32x4* w_t = in_t + CB8 * a.plane;
      const bool x_lo = a.split && (vc % 3) == 1;
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb)
#pragma unroll
        for (int k = 0; k < KPC; ++k)
          if (v_lds[k] >= 0) {
            if (x_lo) {
              float lo[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) lo[j] = pre[cb][k].v[j] - (float)(__bf16)pre[cb][k].v[j];
              in_t[cb * a.plane + v_lds[k]] = pack8(lo);
            } else {
              in_t[cb * a.plane + v_lds[k]] = pack8(pre[cb][k].v);
            }
          }
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ) w_t[i] = wpre[it]; }
    };
    load_chunk(0);
    commit(0);
    if (nvc > 1) load_chunk(1);
    __syncthreads();  // stage 0 is ready
    for (int vc = 0; vc < nvc; ++vc) {
      if (vc + 1 < nvc) {
        commit(vc + 1);
        if (vc + 2 < nvc) load_chunk(vc + 2);
      }
      __syncthreads();
    }
    }  // tile loop
    return;
  }
  // --------------------------------------------------------------------------------------------- consumer
  ESS_TILE_LOOP {
  ESS_TILE_DECODE
  // (the lane-dependent addressing below is tile-invariant; an opaque copy of the thread id per tile keeps hipcc from hoisting it
  // out of the tile loop, where it would stay live through the epilogue: +28 spilled VGPRs in the 128-register instances)
  int tid_t = (int)(threadIdx.x & 255);
  asm volatile("" : "+v"(tid_t));
  const int tid = tid_t, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
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
  // bias-only epilogues (the ConvLSTM / ConvGRU gates; BF16_C8 outputs without a scale): the accumulators start from the bias
  const bool biased = (EPI != ESS_EPI_LINEAR || (OUT8 && a.scale == nullptr)) && a.shift != nullptr;
  if (biased) {
    conv_bias_init<MB>(a, acc, ct, half);
  } else {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  }
  unsigned long long cbar_acc = 0;
  (void)cbar_acc;
  ESS_CT(1);
  __syncthreads();  // stage 0 is ready
  ESS_CT(2);
  // the matrix waves take issue priority over the staging waves that share their SIMDs (s_setprio is a scalar instruction that
  // ignores EXEC: `role` is a readfirstlane value, so only consumer waves reach this point)
  __builtin_amdgcn_s_setprio(1);
  // ---- K loop.  The fragment reads of tap t+1 are issued BEFORE the MFMAs of tap t (register double buffer) and waited for
  // with a counted lgkmcnt just before their own MFMAs.  hipcc undoes that order when it sees plain loads (it sinks each
  // ds_read to just above its first use and waits for it there: one exposed LDS round trip per tap), so the reads and the
  // waits are volatile asm statements -- the wait takes the fragments as in/out operands, which ties the MFMAs behind it.
  // LDS byte addresses: A (weights) = one base + immediates; B (pixels) = one base per (pixel block, filter row).
  const unsigned lds0 = (unsigned)(size_t)(smem16);  // (LDS address of the dynamic segment: its low 32 bits)
  unsigned a_base = (unsigned)((half * COT + p) * 16);
  unsigned b_base[NBW][KS];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) b_base[nb][ky] = (unsigned)((boff[nb] + ky * a.row_pitch) * 16);
  const unsigned w_off = (unsigned)(CB8 * a.plane * 16), buf_bytes = (unsigned)(bufsz * 16);
  struct Frags { u32x4 a[MB]; u32x4 b[NBW]; };
  const int nvc_c = a.split ? 3 * a.n_chunks : a.n_chunks;  // (split operands: three virtual chunks per chunk, see the staging waves)
  for (int ch = 0; ch < nvc_c; ++ch) {
    if (ch < 40) ESS_CT(3 + ch);
    const unsigned stage_b = lds0 + (ch & 1) * buf_bytes;
    const unsigned wa = stage_b + w_off + a_base;
    unsigned ba[NBW][KS];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) ba[nb][ky] = stage_b + b_base[nb][ky];
    Frags f0, f1;
#define ESS_READ_TAP(F_, TAP_)                                                                                                   \
    {                                                                                                                            \
      constexpr int ky_ = (TAP_) / KS, kx_ = (TAP_) % KS;                                                                        \
      _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                          \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.a[mb]) : "v"(wa + (unsigned)(mb * 32 * 16)), "n"((TAP_) * CB8 * COT * 16)); \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                         \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[nb]) : "v"(ba[nb][ky_]), "n"(kx_ * 16));                       \
    }
#define ESS_WAIT(F_, N_)                                                                                                         \
    {                                                                                                                            \
      if constexpr (MB == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(F_.a[0]), "+v"(F_.b[0]), "+v"(F_.b[1]) : "n"(N_));      \
      else if constexpr (MB == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(F_.a[0]), "+v"(F_.a[MB > 1 ? 1 : 0]), "+v"(F_.b[0]), "+v"(F_.b[1]) : "n"(N_)); \
      else asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(F_.a[0]), "+v"(F_.a[MB > 1 ? 1 : 0]), "+v"(F_.a[MB > 2 ? 2 : 0]), "+v"(F_.a[MB > 3 ? 3 : 0]), "+v"(F_.b[0]), "+v"(F_.b[1]) : "n"(N_)); \
    }
#define ESS_MMA(F_)                                                                                                              \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                            \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                         \
        acc[mb][nb] = ess_mfma16<H>(F_.a[mb], F_.b[nb], acc[mb][nb]);
    constexpr int NR = MB + NBW;  // LDS reads per tap
#ifdef ESS_ABLATE
    if (a.deep & 2) { __syncthreads(); continue; }  // (ablation build only: no fragment reads, no MFMAs)
#endif
    ESS_READ_TAP(f0, 0)
    ESS_READ_TAP(f1, 1) ESS_WAIT(f0, NR) ESS_MMA(f0)
    ESS_READ_TAP(f0, 2) ESS_WAIT(f1, NR) ESS_MMA(f1)
    ESS_READ_TAP(f1, 3) ESS_WAIT(f0, NR) ESS_MMA(f0)
    ESS_READ_TAP(f0, 4) ESS_WAIT(f1, NR) ESS_MMA(f1)
    ESS_READ_TAP(f1, 5) ESS_WAIT(f0, NR) ESS_MMA(f0)
    ESS_READ_TAP(f0, 6) ESS_WAIT(f1, NR) ESS_MMA(f1)
    ESS_READ_TAP(f1, 7) ESS_WAIT(f0, NR) ESS_MMA(f0)
    ESS_READ_TAP(f0, 8) ESS_WAIT(f1, NR) ESS_MMA(f1)
    ESS_WAIT(f0, 0) ESS_MMA(f0)
#undef ESS_READ_TAP
#undef ESS_WAIT
#undef ESS_MMA
    if (ch == nvc_c - 1) ESS_CT(44);
    ESS_SYNC_ACC(cbar_acc);
  }
  ESS_CT_VAL(42, cbar_acc);
  ESS_CT(45);
  __builtin_amdgcn_s_setprio(0);
#ifdef ESS_ABLATE
  if (a.deep & 8) continue;  // (ablation build only: no epilogue)
#endif
  if constexpr (OUT8) conv_epilogue_c8<MB, H>(a, acc, ct, n, half, x0 + lx, y0, ly, biased);
  else conv_epilogue<MB, EPI, false, H>(a, acc, ct, n, half, x0 + lx, y0, ly, biased);
  ESS_CT(47);
  if constexpr (EPI != ESS_EPI_LSTM) ESS_CW(43);
  }  // tile loop
#undef ESS_TILE_LOOP
#undef ESS_TILE_DECODE
}


template <int MB, bool SRCBF>
void launch_ws(int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (SRCBF) {
    if (a.f16) {  // ESS_COMPUTE_F16 (the dispatcher lets only F16_C8 sources through)
#define ESS_WSH(E_, O8_) { ess_allow_lds(conv_bf16_ws_k3s1_kernel<MB, E_, true, O8_, true>, lds); hipLaunchKernelGGL((conv_bf16_ws_k3s1_kernel<MB, E_, true, O8_, true>), grid, dim3(512), lds, st, a); }
      if (a.fmt_out == ESS_FMT_BF16_C8) ESS_WSH(ESS_EPI_LINEAR, true)
      else if (epi == ESS_EPI_LSTM) ESS_WSH(ESS_EPI_LSTM, false)
      else if (epi == ESS_EPI_GRU_UR) ESS_WSH(ESS_EPI_GRU_UR, false)
      else if (epi == ESS_EPI_GRU_OUT) ESS_WSH(ESS_EPI_GRU_OUT, false)
      else ESS_WSH(ESS_EPI_LINEAR, false)
#undef ESS_WSH
      return;
    }
  }
  if (a.fmt_out == ESS_FMT_BF16_C8) {  // (validated: LINEAR epilogue)
    ess_allow_lds(conv_bf16_ws_k3s1_kernel<MB, ESS_EPI_LINEAR, SRCBF, true>, lds);
    hipLaunchKernelGGL((conv_bf16_ws_k3s1_kernel<MB, ESS_EPI_LINEAR, SRCBF, true>), grid, dim3(512), lds, st, a);
    return;
  }
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
}  // namespace

namespace essconv {

void conv_bf16_launch_ws(int mb, int epi, bool c8, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (c8) launch_ws_mb<true>(mb, epi, grid, lds, st, a);
  else launch_ws_mb<false>(mb, epi, grid, lds, st, a);
}

}  // namespace essconv

#ifdef ESS_CV_TRACE
extern "C" int ess_debug_conv_trace(unsigned long long* host_out, size_t n) {
  (void)hipMemset((void*)nullptr, 0, 0);
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_cv_trace), n * sizeof(unsigned long long));
}
extern "C" int ess_debug_conv_trace_clear() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_cv_trace)) != hipSuccess) return -1;
  return (int)hipMemset(p, 0, sizeof(unsigned long long) * 2048 * 8 * 48);
}
#endif

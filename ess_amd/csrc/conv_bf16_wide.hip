// Wide-tile 3x3 / stride-1 convolution on the bf16 matrix cores: BF16_C8 sources, BF16_C8 output(s), LINEAR epilogue.
//
// Round 4.  The wave-specialised kernel of conv_bf16_ws.hip owns a 64-channel x 256-pixel tile per workgroup, two workgroups per
// CU.  Three rounds of cycle stamps say what that geometry costs: (i) ROUND QUANTISATION -- 256 -> 256 @ 60 x 80 at
// B = 8 is 640 tiles on 512 resident slots: 128 workgroups run a second tile alone on half the chip, 52 us where the balanced
// work is 42; no 256-pixel tile divides 8 x 60 x 80 into <= 512 pieces; (ii) STAGED BYTES -- 28.8 KB per workgroup and
// 16-channel chunk, 64 % of it the weight slab every pixel tile re-stages, through an L2 -> CU path that measures 17.7 B/clk/CU
// against the 25 B/clk the matrix pipe would need.
//
// Here a workgroup owns FIVE pixel blocks per matrix wave instead of two (a pixel block = 2 rows x 16 columns = one 32-column
// MFMA operand): 320 pixels with the four matrix waves arranged 2 (channel halves) x 2 (pixel halves), 640 pixels arranged
// 1 x 4.  One workgroup per CU (160 accumulator registers per lane at two 32-channel blocks per wave; 2 waves per SIMD = 256
// registers each), whose matrix waves issue 10 (or 5) MFMAs per filter tap against 7 (or 6) LDS fragment reads:
//
//   variant <MBW, CW>     tile (channels x pixels)   B = 8 launch of                      workgroups     staged B / MFMA clk / CU
//   <2, 2>  128 x 320     (20 x 16 px)               256 -> 256 @ 60 x 80                 240 (1 round)  17.2   (ws kernel: 25.0)
//   <2, 2>                                            (128^ + 128) -> 128 @ 120 x 160      480 (2 rounds)
//   <1, 2>   64 x 320                                 256 -> 128 @ 60 x 80                 240 (1 round)
//   <2, 1>   64 x 640     (40 x 16 px)               128 -> 64 @ 120 x 160                240 (1 round)
//   <2, 1>                                            (64^ + 64) -> 64, 64 -> 64 @ 240x320 960 (3.75 rounds)
//   <1, 1>   32 x 640                                 64^ -> 32 @ 480 x 640                1920 (7.5 rounds)
//
// The weight pack is the ws kernel's own ([64- or 32-channel tile][chunk][tap][8-channel block][channel][8]): a 128-channel
// workgroup stages the slabs of two adjacent 64-channel tiles side by side, so the plan, the packed weights and every caller are
// unchanged -- the dispatcher (conv_bf16.hip) picks this kernel per launch when its round count beats the ws kernel's.
// Structure otherwise as conv_bf16_ws.hip: 512 threads, waves 0-3 matrix (LDS fragment reads + MFMA, issue priority), waves 4-7
// staging (16-byte loads of BF16_C8 pixel vectors / packed weights -> ds_write_b128), two LDS stages, one barrier per 16-channel
// chunk, persistent tile loop (the staging waves run ahead into the next tile during the epilogue).
// (Measured and dropped: BF16_C8 outputs with the `nt` cache policy, -DESS_C8_AUX=2 for this file.  Back-to-back launches of one layer
// gain 3-4 % -- 42.3 -> 40.8, 25.2 -> 24.3, 25.9 -> 24.8, 113.4 -> 111.3 us: the output burst of a single-round launch streams past the
// L2 --, but inside the train step the next kernel reads that output and the same-box A/B goes the other way: conv time in the step
// 6.36-6.42 -> 6.45-6.49 ms, step 23.16-23.18 -> 23.19-23.20 ms.  Default policy.)
//
// S2D (round 5, second session): a 5x5 / stride-2 / pad-2 convolution is a 3x3 / stride-1 / pad-1 convolution over the SPACE-TO-DEPTH
// view of its input -- out[y][x] = sum w[ky][kx] in[2y + ky - 2][2x + kx - 2], and with ky - 2 = 2 dy + py (py = row parity of the
// input pixel) the 25 taps become, per parity class (py, px), the taps (dy, dx) in {-1, 0, (+1 only for parity 0)}^2 of the
// half-resolution phase plane in[2 . + py][2 . + px]: 9 + 6 + 6 + 4 = 25 products.  So the frozen encoder's three downsampling
// convolutions (e2vid/model/unet.py:117-181 encoders, submodules.py:176-186 RecurrentConvLayer.conv; 84 us each on the tap-paired
// kernel: 8-channel chunks of 26 MFMAs per wave between barriers, 1.5 fragment reads per MFMA, 0.30 of peak) run HERE, on 16-channel
// chunks of 90 / 60 / 60 / 40 MFMAs at 0.7 reads per MFMA: source mode ESS_SRC_S2D = the stored tensor is [N][Cin/8][2 Hin][2 Win][8],
// the descriptor's C0 = 4 Cin VIRTUAL channels, 16 per chunk, in the chunk order of s2d_class / s2d_group below (class q = py + 2 px), the
// staging waves gather a chunk's pixel vectors at (2 gy + py, 2 gx + px), and the matrix waves run a class's chunk with its own tap list
// (no products with the zero taps a dense 3x3 over 4 Cin channels would carry: 25 / 36 of the work).  The weight pack (w_kind
// ESS_W_CONV5_S2D: from the [Cout][Cin][5][5] tensor) keeps a chunk's USED taps first (class order below), so the staging waves copy
// only the used prefix of a chunk's slab.  Needs Cin % 32 == 0 (an even number of chunks per class keeps the fragment-set roles).
#include "conv_bf16_common.h"

namespace {

using namespace essconv;

// class q = py + 2 px: taps (ty * 3 + tx) a chunk of that class contracts, in the order its weight slab stores them
//   q 0 (0, 0): 0 1 2 3 4 5 6 7 8      q 1 (py 1): 0 1 2 3 4 5      q 2 (px 1): 0 1 3 4 6 7      q 3 (1, 1): 0 1 3 4
__host__ __device__ constexpr int s2d_ntaps(int q) { return q == 0 ? 9 : (q == 3 ? 4 : 6); }
// Chunk ORDER of the space-to-depth form (= order of the virtual channels, 16 per chunk; nq = chunks per class): the two column parities
// of one row parity sit next to each other -- (q 0, g), (q 2, g) for every channel group g, then (q 1, g), (q 3, g) -- because they
// read the two halves of the same 32-byte pixel pairs, i.e. the same cache lines: class-major order (all of q 0, then q 1, ...) put
// 2 nq chunks of every workgroup of the XCD between the two touches of a line, more than its 4 MB L2 holds, and the counters showed
// every input line fetched twice (profiles/r5b_conv_pmc_before_interleave.txt: 310 MB read for 157 MB of input at level 0, 5.2 TB/s).
__host__ __device__ constexpr int s2d_class(int p, int nq) { return p < 2 * nq ? ((p & 1) ? 2 : 0) : ((p & 1) ? 3 : 1); }
__host__ __device__ constexpr int s2d_group(int p, int nq) { return (p < 2 * nq ? p : p - 2 * nq) >> 1; }
constexpr int WIDE_NB = 5;    // pixel blocks per matrix wave
constexpr int WIDE_RP = 32;   // LDS row pitch in 16-byte vectors: 18 used; rows of a pixel block must start 0 mod 16 vectors apart (ds_read_b128 lane groups)
constexpr int WIDE_IW = 18;   // 16 + 2 halo columns

// EPI: ESS_EPI_LINEAR (BF16_C8 outputs), ESS_EPI_LSTM (the lean ConvLSTM step: F32_C8 cell state in / out, BF16_C8 copy of h', bias in
// the accumulators -- conv_epilogue_lstm_c8) or ESS_EPI_GRU_UR / ESS_EPI_GRU_OUT (the lean ConvGRU kernel pair: conv_epilogue_gru_*_c8)
// H: IEEE-half operands and 16-bit outputs (ESS_COMPUTE_F16) -- the same kernel with v_mfma_f32_32x32x16_f16 and half conversions in the epilogue
template <int MBW, int CW, int EPI = ESS_EPI_LINEAR, bool S2D = false, bool H = false>
__global__ __launch_bounds__(512, 2) void conv_bf16_wide_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  constexpr int KS = 3, CB8 = 2, CK = 16, NB = WIDE_NB, RP = WIDE_RP;
  constexpr int PW = 4 / CW;                 // matrix waves along pixels
  constexpr int TH = PW * NB * 2, TW = 16;   // tile: TH rows x 16 columns
  constexpr int IH = TH + 2, IW = WIDE_IW;
  constexpr int PLANE = IH * RP;             // one 8-channel block of the input tile (16-byte vectors)
  constexpr int COT = MBW * CW * 32;         // output channels per workgroup
  constexpr int WSZ = KS * KS * CB8 * COT;   // weight vectors per stage
  constexpr int WV = (WSZ + 255) / 256;
  constexpr int NPOS = IH * IW;
  constexpr int KPC = (NPOS + 255) / 256;
  constexpr int BUFSZ = CB8 * PLANE + WSZ;   // one stage (16-byte vectors)
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  // tile schedule: as conv_bf16_ws.hip (persistent: workgroup b walks every (grid / 8)-th tile of its XCD's contiguous range)
  const int SLAB = a.slab;                   // rows of one packed weight slab (the plan's cout_tile; <= COT, a power of two)
  const int NSLAB = COT / SLAB;              // slabs a workgroup stages side by side (2 for a 128-row tile over 64-row slabs)
  const int n_ct = a.n_cout_tiles / NSLAB;   // workgroup-level channel tiles
  int t_start, t_count, t_first = 0, t_step = 1;
  if (a.persist) {
    const int total = n_ct * a.n_tiles * a.N, q = total >> 3, r = total & 7, xcd = (int)(blockIdx.x & 7);
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
  const int ct = logical % n_ct;                                           \
  const int sp = logical / n_ct;                                           \
  const int tile = sp % a.n_tiles, n = sp / a.n_tiles;                     \
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;             \
  const int y0 = ty * TH, x0 = tx * TW;

  if (role == 1) {
    const int tid = threadIdx.x & 255;
    ESS_TILE_LOOP {
    ESS_TILE_DECODE
    // ------------------------------------------------------------------ staging waves (BF16_C8 sources; see conv_bf16_ws.hip)
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    // S2D: the stored source has twice the (virtual) extent and a quarter of the (virtual) channels; a.Hin / a.Win are the virtual ones
    const int sh0 = S2D ? 0 : (a.mode0 != ESS_SRC_DIRECT ? 1 : 0), sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
    const int Wp0 = S2D ? 2 * a.Win : (a.Win >> sh0), Wp1 = a.Win >> sh1;
    const size_t hw0 = S2D ? (size_t)4 * a.Hin * a.Win : (size_t)(a.Hin >> sh0) * Wp0, hw1 = (size_t)(a.Hin >> sh1) * Wp1;
    const int nb0 = S2D ? (a.C0 >> 5) : ((a.C0 + 7) >> 3), nb1 = (a.C1 + 7) >> 3;  // (S2D: Cin / 8 stored blocks)
    const int nq = S2D ? (a.n_chunks >> 2) : 1;                                     // (S2D: chunks per parity class)
    const u32x4* s0 = (const u32x4*)a.src0 + (size_t)n * nb0 * hw0;
    const u32x4* s1 = a.C1 ? (const u32x4*)a.src1 + (size_t)n * nb1 * hw1 : s0;
    unsigned v_pos0[KPC], v_pos1[KPC], v_keep0[KPC], v_keep1[KPC];
    int v_lds[KPC];
#pragma unroll
    for (int k = 0; k < KPC; ++k) {
      const int vi = tid + k * 256;
      const int iy = vi / IW, ix = vi - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool in = vi < NPOS && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
      const bool odd = ((gy | gx) & 1) != 0;
      const bool in0 = in && !(a.mode0 == ESS_SRC_ZERO_UP2 && odd), in1 = in && !(a.mode1 == ESS_SRC_ZERO_UP2 && odd);
      v_lds[k] = vi < NPOS ? iy * RP + ix : -1;
      v_pos0[k] = in0 ? (S2D ? (unsigned)(2 * gy * Wp0 + 2 * gx) : (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0))) : 0u;
      v_pos1[k] = in1 ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) : 0u;
      v_keep0[k] = in0 ? 0xffffffffu : 0u;
      v_keep1[k] = in1 ? 0xffffffffu : 0u;
    }
    // weight vectors of this thread: LDS slot i = ((tap * 2 + cb) * COT + c); global: slab c / SLAB of channel tile ct * NSLAB + ..,
    // vector ((tap * 2 + cb) * SLAB + c % SLAB) of that slab's chunk block
    unsigned w_src[WV];
#pragma unroll
    for (int it = 0; it < WV; ++it) {
      const int i = tid + it * 256;
      const int ic = i < WSZ ? i : 0;
      const int c = ic % COT, tc = ic / COT;
      w_src[it] = (unsigned)((c / SLAB) * a.n_chunks * (KS * KS * CB8 * SLAB) + tc * SLAB + (c % SLAB));  // (SLAB is wave-uniform)
    }
    struct Set { u32x4 pre[CB8][KPC]; u32x4 wpre[WV]; };
    Set sa;
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)(ct * NSLAB) * a.n_chunks * (KS * KS * CB8 * SLAB);
    auto load_chunk = [&](int ch, Set& r) {
#ifdef ESS_ABLATE
      if (a.deep & 4) return;  // (ablation build only, switch ESS_WS_ABL: no global loads)
#endif
      if constexpr (S2D) {
        const int q = s2d_class(ch, nq), cq = s2d_group(ch, nq);      // parity class, 16-channel group inside the class
        const unsigned off_q = (unsigned)((q & 1) * Wp0 + (q >> 1));  // pixel (py, px) of the 2 x 2 cell
#pragma unroll
        for (int cb = 0; cb < CB8; ++cb) {
          const u32x4* sp = s0 + (size_t)(cq * 2 + cb) * hw0;
#pragma unroll
          for (int k = 0; k < KPC; ++k) r.pre[cb][k] = sp[v_pos0[k] + off_q];
        }
        const u32x4* wsrc = wbase + (size_t)ch * (KS * KS * CB8 * SLAB);
        const int wlim = s2d_ntaps(q) * CB8 * COT;  // the used taps are the first ones of the slab (LDS index space is tap-major)
#pragma unroll
        for (int it = 0; it < WV; ++it) r.wpre[it] = wsrc[tid + it * 256 < wlim ? w_src[it] : 0];
      } else {
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb) {
        const int c0 = ch * CK + cb * 8;
        const bool first = c0 < a.C0 || a.C1 == 0;
        const int bi = (first ? c0 : c0 - a.C0) >> 3, nbs = first ? nb0 : nb1;
        const u32x4* sp = (first ? s0 : s1) + (size_t)(bi < nbs ? bi : 0) * (first ? hw0 : hw1);
#pragma unroll
        for (int k = 0; k < KPC; ++k) r.pre[cb][k] = sp[first ? v_pos0[k] : v_pos1[k]];
      }
      const u32x4* wsrc = wbase + (size_t)ch * (KS * KS * CB8 * SLAB);
#pragma unroll
      for (int it = 0; it < WV; ++it) r.wpre[it] = wsrc[w_src[it]];
      }
    };
    auto commit = [&](int ch, int buf, const Set& r) {
#ifdef ESS_ABLATE
      if (a.deep & 16) return;  // (ablation build only: no LDS writes)
#endif
      u32x4* in_t = smem16 + buf * BUFSZ;
      u32x4* w_t = in_t + CB8 * PLANE;
#pragma unroll
      for (int cb = 0; cb < CB8; ++cb) {
        const int c0 = ch * CK + cb * 8;
        const bool first = S2D || c0 < a.C0 || a.C1 == 0;
        const unsigned blk_ok = S2D ? 0xffffffffu : (((first ? c0 : c0 - a.C0) >> 3) < (first ? nb0 : nb1) ? 0xffffffffu : 0u);
#pragma unroll
        for (int k = 0; k < KPC; ++k) {
          const unsigned m = (first ? v_keep0[k] : v_keep1[k]) & blk_ok;
          u32x4 v = r.pre[cb][k];
          v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
          if (v_lds[k] >= 0) in_t[cb * PLANE + v_lds[k]] = v;
        }
      }
      const int wlim = S2D ? s2d_ntaps(s2d_class(ch, nq)) * CB8 * COT : WSZ;
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < wlim) w_t[i] = r.wpre[it]; }
    };
    const int nch = a.n_chunks;
    load_chunk(0, sa);
    commit(0, 0, sa);
    if (nch > 1) load_chunk(1, sa);
    __syncthreads();  // stage 0 is ready
    for (int ch = 0; ch < nch; ++ch) {
      if (ch + 1 < nch) {
        commit(ch + 1, (ch + 1) & 1, sa);
        if (ch + 2 < nch) load_chunk(ch + 2, sa);
      }
      __syncthreads();
    }
    }  // tile loop
    return;
  }
  // --------------------------------------------------------------------------------------------- matrix waves
  ESS_TILE_LOOP {
  ESS_TILE_DECODE
  int tid_t = (int)(threadIdx.x & 255);
  asm volatile("" : "+v"(tid_t));  // (keeps the lane addressing out of the tile loop's live range: see conv_bf16_ws.hip)
  const int tid = tid_t, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int ox = p & 15, oy = p >> 4;
  const int cw = wave % CW, pw = wave / CW;  // channel group / pixel group of this wave
  int ly[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) ly[nb] = (pw * NB + nb) * 2 + oy;
  const int ct_w = ct * CW + cw;             // this wave's channel tile in units of MBW * 32 channels
  f32x16 acc[MBW][NB];
  const bool biased = (EPI != ESS_EPI_LINEAR || a.scale == nullptr) && a.shift != nullptr;
  if (biased) {
    conv_bias_init<MBW>(a, acc, ct_w, half);
  } else {
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  }
  __syncthreads();  // stage 0 is ready
  __builtin_amdgcn_s_setprio(1);
  // LDS byte addresses: everything but two bases is an immediate (row pitch, plane and tile are compile-time here)
  const unsigned lds0 = (unsigned)(size_t)(smem16);
  const unsigned a_base = (unsigned)((half * COT + cw * MBW * 32 + p) * 16) + (unsigned)(CB8 * PLANE * 16);
  const unsigned b_base = (unsigned)((half * PLANE + (pw * NB * 2 + oy) * RP + ox) * 16);
  struct Frags { u32x4 a[MBW]; u32x4 b[NB]; };
  // ---- K loop.  Per filter tap: the fragment reads of the NEXT tap are issued before this tap's MFMAs and waited for with a
  // counted lgkmcnt (volatile asm, the wait takes the fragments as in/out operands: conv_bf16_ws.hip explains why).  The two
  // fragment sets alternate tap by tap; nine taps are an odd number, so they also swap roles from one chunk to the next, and
  // that is used: tap 0 of chunk ch + 1 is read right behind the barrier that ends chunk ch, BEFORE the MFMAs of chunk ch's last
  // tap -- ten matrix instructions cover the LDS round trip that a chunk would otherwise start with (one matrix wave per SIMD:
  // nobody else fills that gap).
#define ESS_READ_TAP(F_, TAP_, STG_) ESS_READ_TS(F_, TAP_, TAP_, STG_)
  // (TAP_: the filter tap = where the pixel fragments sit in the input tile; SLOT_: where that tap's weights sit in the staged slab)
#define ESS_READ_TS(F_, TAP_, SLOT_, STG_)                                                                                       \
    {                                                                                                                            \
      constexpr int ky_ = (TAP_) / KS, kx_ = (TAP_) % KS;                                                                        \
      const unsigned wa_ = (STG_) + a_base, ba_ = (STG_) + b_base;                                                               \
      _Pragma("unroll") for (int mb = 0; mb < MBW; ++mb)                                                                         \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.a[mb]) : "v"(wa_ + (unsigned)(mb * 32 * 16)), "n"((SLOT_) * CB8 * COT * 16)); \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[0]) : "v"(ba_), "n"((0 * 2 * RP + ky_ * RP + kx_) * 16));        \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[1]) : "v"(ba_), "n"((1 * 2 * RP + ky_ * RP + kx_) * 16));        \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[2]) : "v"(ba_), "n"((2 * 2 * RP + ky_ * RP + kx_) * 16));        \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[3]) : "v"(ba_), "n"((3 * 2 * RP + ky_ * RP + kx_) * 16));        \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[4]) : "v"(ba_), "n"((4 * 2 * RP + ky_ * RP + kx_) * 16));        \
    }
#define ESS_WAIT(F_, N_)                                                                                                         \
    {                                                                                                                            \
      if constexpr (MBW == 1)                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(F_.a[0]), "+v"(F_.b[0]), "+v"(F_.b[1]), "+v"(F_.b[2]), "+v"(F_.b[3]),        \
                     "+v"(F_.b[4]) : "n"(N_));                                                                                   \
      else                                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(F_.a[0]), "+v"(F_.a[MBW - 1]), "+v"(F_.b[0]), "+v"(F_.b[1]), "+v"(F_.b[2]),  \
                     "+v"(F_.b[3]), "+v"(F_.b[4]) : "n"(N_));                                                                    \
    }
#define ESS_TIE(F_)                                                                                                              \
    {                                                                                                                            \
      if constexpr (MBW == 1)                                                                                                    \
        asm volatile("" : "+v"(F_.a[0]), "+v"(F_.b[0]), "+v"(F_.b[1]), "+v"(F_.b[2]), "+v"(F_.b[3]), "+v"(F_.b[4]));             \
      else                                                                                                                       \
        asm volatile("" : "+v"(F_.a[0]), "+v"(F_.a[MBW - 1]), "+v"(F_.b[0]), "+v"(F_.b[1]), "+v"(F_.b[2]), "+v"(F_.b[3]), "+v"(F_.b[4])); \
    }
#define ESS_MMA(F_)                                                                                                              \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                                            \
      _Pragma("unroll") for (int mb = 0; mb < MBW; ++mb)                                                                         \
        acc[mb][nb] = ess_mfma16<H>(F_.a[mb], F_.b[nb], acc[mb][nb]);
  // one chunk: on entry tap 0 is in flight into FA_; on exit tap 0 of the next chunk (if any) is in flight into FB_
#define ESS_CHUNK(FA_, FB_, CH_)                                                                                                 \
    {                                                                                                                            \
      const unsigned stg_ = lds0 + (unsigned)(((CH_) & 1) * BUFSZ * 16);                                                         \
      ESS_READ_TAP(FB_, 1, stg_) ESS_WAIT(FA_, NR) ESS_MMA(FA_)                                                                  \
      ESS_READ_TAP(FA_, 2, stg_) ESS_WAIT(FB_, NR) ESS_MMA(FB_)                                                                  \
      ESS_READ_TAP(FB_, 3, stg_) ESS_WAIT(FA_, NR) ESS_MMA(FA_)                                                                  \
      ESS_READ_TAP(FA_, 4, stg_) ESS_WAIT(FB_, NR) ESS_MMA(FB_)                                                                  \
      ESS_READ_TAP(FB_, 5, stg_) ESS_WAIT(FA_, NR) ESS_MMA(FA_)                                                                  \
      ESS_READ_TAP(FA_, 6, stg_) ESS_WAIT(FB_, NR) ESS_MMA(FB_)                                                                  \
      ESS_READ_TAP(FB_, 7, stg_) ESS_WAIT(FA_, NR) ESS_MMA(FA_)                                                                  \
      ESS_READ_TAP(FA_, 8, stg_) ESS_WAIT(FB_, NR) ESS_MMA(FB_)                                                                  \
      ESS_WAIT(FA_, 0)                                                                                                           \
      __syncthreads(); /* every read of this stage has returned; the next stage is complete */                                  \
      /* (unconditional: behind the last chunk it reads the other stage for nothing -- a branch here lets hipcc hoist the */    \
      /* MFMAs above the reads; the empty asm ties them behind) */                                                              \
      ESS_READ_TAP(FB_, 0, lds0 + (unsigned)((((CH_) + 1) & 1) * BUFSZ * 16))                                                    \
      ESS_TIE(FA_)                                                                                                               \
      ESS_MMA(FA_)                                                                                                               \
    }
  // S2D chunks: the same pipeline over a class's own tap list (6 or 4 taps, slots 0 .. n - 1 of the slab).  An EVEN number of taps
  // leaves the fragment sets in the roles they entered with (tap 0 of the next chunk goes back into FA_).
#define ESS_STEP(FN_, FC_, TAP_, SLOT_, stg_) ESS_READ_TS(FN_, TAP_, SLOT_, stg_) ESS_WAIT(FC_, NR) ESS_MMA(FC_)
#define ESS_CHUNK_TAIL_EVEN(FA_, FB_, CH_)                                                                                       \
      ESS_WAIT(FB_, 0)                                                                                                           \
      __syncthreads();                                                                                                           \
      ESS_READ_TAP(FA_, 0, lds0 + (unsigned)((((CH_) + 1) & 1) * BUFSZ * 16))                                                    \
      ESS_TIE(FB_)                                                                                                               \
      ESS_MMA(FB_)
#define ESS_CHUNK6(FA_, FB_, CH_, T1, T2, T3, T4, T5)                                                                            \
    {                                                                                                                            \
      const unsigned stg_ = lds0 + (unsigned)(((CH_) & 1) * BUFSZ * 16);                                                         \
      ESS_STEP(FB_, FA_, T1, 1, stg_) ESS_STEP(FA_, FB_, T2, 2, stg_) ESS_STEP(FB_, FA_, T3, 3, stg_)                             \
      ESS_STEP(FA_, FB_, T4, 4, stg_) ESS_STEP(FB_, FA_, T5, 5, stg_)                                                            \
      ESS_CHUNK_TAIL_EVEN(FA_, FB_, CH_)                                                                                         \
    }
#define ESS_CHUNK4(FA_, FB_, CH_, T1, T2, T3)                                                                                    \
    {                                                                                                                            \
      const unsigned stg_ = lds0 + (unsigned)(((CH_) & 1) * BUFSZ * 16);                                                         \
      ESS_STEP(FB_, FA_, T1, 1, stg_) ESS_STEP(FA_, FB_, T2, 2, stg_) ESS_STEP(FB_, FA_, T3, 3, stg_)                             \
      ESS_CHUNK_TAIL_EVEN(FA_, FB_, CH_)                                                                                         \
    }
  constexpr int NR = MBW + NB;  // LDS reads per tap
  const int nch = a.n_chunks;
  Frags f0, f1;
#ifdef ESS_ABLATE
  if (a.deep & 2) {  // (ablation build only: no fragment reads, no MFMAs)
    for (int ch = 0; ch < nch; ++ch) __syncthreads();
  } else
#endif
  {
    ESS_READ_TAP(f0, 0, lds0)
    if constexpr (S2D) {
      const int nq = nch >> 2;  // chunks per parity class (even: validate()); chunk order: s2d_class / s2d_group above
      for (int ch = 0; ch < 2 * nq; ch += 4) {  // row parity 0: (q 0: nine taps -- the fragment sets swap roles), (q 2: six -- they keep them)
        ESS_CHUNK(f0, f1, ch)
        ESS_CHUNK6(f1, f0, ch + 1, 1, 3, 4, 6, 7)  // px = 1: columns dx in {-1, 0}
        ESS_CHUNK(f1, f0, ch + 2)
        ESS_CHUNK6(f0, f1, ch + 3, 1, 3, 4, 6, 7)
      }
      for (int ch = 2 * nq; ch < nch; ch += 2) {  // row parity 1 (rows dy in {-1, 0}): q 1, then q 3
        ESS_CHUNK6(f0, f1, ch, 1, 2, 3, 4, 5)
        ESS_CHUNK4(f0, f1, ch + 1, 1, 3, 4)
      }
    } else {
    for (int ch = 0; ch < nch; ch += 2) {
      ESS_CHUNK(f0, f1, ch)
      if (ch + 1 < nch) ESS_CHUNK(f1, f0, ch + 1)
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the read issued behind the last chunk's barrier)
  }
#undef ESS_CHUNK4
#undef ESS_CHUNK6
#undef ESS_CHUNK_TAIL_EVEN
#undef ESS_STEP
#undef ESS_READ_TS
#undef ESS_CHUNK
#undef ESS_TIE
#undef ESS_READ_TAP
#undef ESS_WAIT
#undef ESS_MMA
  __builtin_amdgcn_s_setprio(0);
#ifdef ESS_ABLATE
  if (a.deep & 8) continue;  // (ablation build only: no epilogue)
#endif
  if constexpr (EPI == ESS_EPI_LSTM) {
    int pixi[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int y = y0 + ly[nb], x = x0 + ox;
      pixi[nb] = (y < a.Hout && x < a.Wout) ? y * a.Wout + x : -1;
    }
    conv_epilogue_lstm_c8<MBW, NB, H>(a, acc, ct_w, n, half, pixi, (unsigned)(a.Hout * a.Wout));
  } else if constexpr (EPI == ESS_EPI_GRU_UR || EPI == ESS_EPI_GRU_OUT) {
    int pixi[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int y = y0 + ly[nb], x = x0 + ox;
      pixi[nb] = (y < a.Hout && x < a.Wout) ? y * a.Wout + x : -1;
    }
    if constexpr (EPI == ESS_EPI_GRU_UR) conv_epilogue_gru_ur_c8<MBW, NB, H>(a, acc, ct_w, n, half, pixi, (unsigned)(a.Hout * a.Wout));
    else conv_epilogue_gru_out_c8<MBW, NB, H>(a, acc, ct_w, n, half, pixi, (unsigned)(a.Hout * a.Wout));
  } else {
    conv_epilogue_c8_wide<MBW, NB, H>(a, acc, ct_w, n, half, x0 + ox, y0, ly, biased);
  }
  }  // tile loop
#undef ESS_TILE_LOOP
#undef ESS_TILE_DECODE
}

template <int MBW, int CW, int EPI = ESS_EPI_LINEAR, bool S2D = false>
void launch_wide_t(dim3 grid, hipStream_t st, const ConvKArgs& a) {
  constexpr int PW = 4 / CW, TH = PW * WIDE_NB * 2, PLANE = (TH + 2) * WIDE_RP, COT = MBW * CW * 32;
  constexpr size_t lds = 2 * (size_t)(2 * PLANE + 9 * 2 * COT) * 16;
  static_assert(lds <= 160 * 1024, "two stages must fit the 160 KiB LDS");
  if (a.f16) {  // ESS_COMPUTE_F16
    ess_allow_lds(conv_bf16_wide_kernel<MBW, CW, EPI, S2D, true>, lds);
    hipLaunchKernelGGL((conv_bf16_wide_kernel<MBW, CW, EPI, S2D, true>), grid, dim3(512), lds, st, a);
    return;
  }
  ess_allow_lds(conv_bf16_wide_kernel<MBW, CW, EPI, S2D>, lds);
  hipLaunchKernelGGL((conv_bf16_wide_kernel<MBW, CW, EPI, S2D>), grid, dim3(512), lds, st, a);
}

}  // namespace

namespace essconv {

void conv_bf16_wide_tile(int mbw, int cw, int* th, int* tw) {
  *tw = 16;
  *th = (4 / cw) * WIDE_NB * 2;
  (void)mbw;
}

void conv_bf16_launch_wide(int mbw, int cw, int epi, dim3 grid, hipStream_t st, const ConvKArgs& a) {
  if (a.mode0 == ESS_SRC_S2D) {  // (LINEAR, 64- or 128-channel workgroup tiles: the dispatcher offers <2, 2> and <2, 1> only)
    if (cw == 2) launch_wide_t<2, 2, ESS_EPI_LINEAR, true>(grid, st, a);
    else launch_wide_t<2, 1, ESS_EPI_LINEAR, true>(grid, st, a);
    return;
  }
  if (epi == ESS_EPI_LSTM) { launch_wide_t<2, 2, ESS_EPI_LSTM>(grid, st, a); return; }  // (recurrent epilogues: the dispatcher offers <2, 2> only)
  if (epi == ESS_EPI_GRU_UR) { launch_wide_t<2, 2, ESS_EPI_GRU_UR>(grid, st, a); return; }
  if (epi == ESS_EPI_GRU_OUT) { launch_wide_t<2, 2, ESS_EPI_GRU_OUT>(grid, st, a); return; }
  if (mbw == 2 && cw == 2) launch_wide_t<2, 2>(grid, st, a);
  else if (mbw == 2 && cw == 1) launch_wide_t<2, 1>(grid, st, a);
  else if (mbw == 1 && cw == 2) launch_wide_t<1, 2>(grid, st, a);
  else launch_wide_t<1, 1>(grid, st, a);
}

}  // namespace essconv

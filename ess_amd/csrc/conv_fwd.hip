// Direct (im2col-free) 2-D convolution on the CDNA4 matrix cores, fp32 in / fp32 accumulate.
//
// GEMM view per workgroup:  D[cout_tile][256 px] += W[cout_tile][K] * X[K][256 px],  K = (tap, cin).
//   A operand = weights (M = output channel), B operand = activations (N = output pixel), so that
//   for a fixed accumulator register the 32 lanes of a half-wave hold 32 *neighbouring pixels* of one
//   output channel -> NCHW stores are coalesced row segments, and the ConvLSTM/ConvGRU gate maths is
//   lane-local (weight rows are permuted at pack time so one lane owns all gates of a hidden channel).
//   v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, A/B = ONE float per lane (row/col = lane&31,
//   k = lane>>5), so plain NCHW planes in LDS feed it without any transposition.
// Workgroup = 4 waves; each wave owns 2 pixel blocks (32 px each) x MB channel blocks (32 each).
// A pixel block is (32/BW) rows x BW columns, BW in {8,16,32} chosen per layer to fit W_out.
// Input tile (+halo) and the weight slab of one cin-chunk are staged in LDS; sources are read
// through a mode (direct / nearest x2 / zero-insert x2) and may be the concat of two tensors, so
// torch.cat, F.interpolate(nearest) and ConvTranspose2d never materialise anything.
#include "conv_common.h"

namespace {

using namespace essconv;

// positions a thread stages per channel (compile-time bound of the register prefetch), by filter geometry
constexpr int kpc32(int ks, int s) { return stage_kpc(ks, s); }
constexpr unsigned OOB32 = 0x80000000u;  // beyond any buffer: the bounds-checked load returns 0

template <int KS, int S, int MB, int EPI, int CB>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int COT = MB * 32;
  constexpr int KPC = kpc32(KS, S);
  constexpr int WSZ = KS * KS * CB * COT;  // floats of one chunk's weight slab
  constexpr int WV = (WSZ / 4 + 255) / 256;
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
  float* in_t = smem;
  float* w_t = smem + CB * a.plane;

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

  // ---- staging plan (same scheme as conv_bf16.hip): thread t owns tile positions t, t+256, ... of every channel of
  // the chunk; loads are bounds-checked buffer loads whose offset is out of range wherever a zero is wanted (padding,
  // zero-insert holes, channels past the source), so they carry no branch; the values of chunk i+1 are fetched before
  // the MFMA phase of chunk i and written to LDS after it.
  const int iy0 = y0 * S - a.pad, ix0 = x0 * S - a.pad;
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
    v_lds[k] = vi < npos ? iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix) : -1;
    v_o0[k] = (in && !(a.mode0 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh0) * Wp0 + (gx >> sh0)) * 4u : OOB32;
    v_o1[k] = (in && !(a.mode1 == ESS_SRC_ZERO_UP2 && odd)) ? (unsigned)((gy >> sh1) * Wp1 + (gx >> sh1)) * 4u : OOB32;
  }
  auto load_one = [&](int ch, int cb, int k) -> float {
    const int c = ch * CB + cb;            // wave-uniform
    const bool first = c < a.C0 || a.C1 == 0;
    const unsigned off = first ? v_o0[k] + (unsigned)c * pl0 : v_o1[k] + (unsigned)(c - a.C0) * pl1;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(first ? r0 : r1, (int)off, 0, 0));
  };

  float pre[CB][KPC];
  const f32x4* wbase = (const f32x4*)a.wpk + (size_t)ct * a.n_chunks * (WSZ / 4);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int k = 0; k < KPC; ++k) pre[cb][k] = load_one(0, cb, k);

  for (int ch = 0; ch < a.n_chunks; ++ch) {
    __syncthreads();  // everyone finished reading the previous chunk
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int k = 0; k < KPC; ++k)
        if (v_lds[k] >= 0) in_t[cb * a.plane + v_lds[k]] = pre[cb][k];
    {
      const f32x4* wsrc = wbase + (size_t)ch * (WSZ / 4);
      f32x4 wv[WV];
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; wv[it] = wsrc[i < WSZ / 4 ? i : 0]; }
#pragma unroll
      for (int it = 0; it < WV; ++it) { const int i = tid + it * 256; if (i < WSZ / 4) ((f32x4*)w_t)[i] = wv[it]; }
    }
    __syncthreads();
    if (ch + 1 < a.n_chunks) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int k = 0; k < KPC; ++k) pre[cb][k] = load_one(ch + 1, cb, k);
    }
    // ---- MFMA over (tap, channel pair)
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
        const int toff = ky * a.row_pitch + (S == 2 ? (kx & 1) * a.par_off + (kx >> 1) : kx);
        const float* wp = w_t + (tap * CB + half) * COT + p;
        const float* ip = in_t + toff;
#pragma unroll
        for (int kk = 0; kk < CB; kk += 2) {
          float af[MB], bf[NBW];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) af[mb] = wp[kk * COT + mb * 32];
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) bf[nb] = ip[boff[nb] + kk * a.plane];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb], bf[nb], acc[mb][nb], 0, 0, 0);
        }
      }
    }
  }
  conv_epilogue<MB, EPI>(a, acc, ct, n, half, x0 + lx, y0, ly);
}

template <int KS, int S, int MB, int CB>
void launch_epi(int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (KS == 3 && S == 1) {
    switch (epi) {
      case ESS_EPI_LSTM: { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_LSTM, CB>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_LSTM, CB>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_UR: { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_UR, CB>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_UR, CB>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_OUT: { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_OUT, CB>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_OUT, CB>), grid, dim3(256), lds, st, a); } return;
      default: break;
    }
  }
  { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_LINEAR, CB>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_LINEAR, CB>), grid, dim3(256), lds, st, a); }
}

template <int KS, int S>
void launch_mb(int mb, int ck, int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  constexpr int CBD = KS >= 7 ? 2 : (KS == 5 ? 4 : 8);  // default channels per chunk for this filter size
  if constexpr (KS == 5 && S == 1) {
    if (ck == 2) {  // 2-channel head
      if (mb == 2) launch_epi<KS, S, 2, 2>(epi, grid, lds, st, a);
      else launch_epi<KS, S, 1, 2>(epi, grid, lds, st, a);
      return;
    }
  }
  if (mb == 2) launch_epi<KS, S, 2, CBD>(epi, grid, lds, st, a);
  else launch_epi<KS, S, 1, CBD>(epi, grid, lds, st, a);
}

}  // namespace

extern "C" int ess_conv2d_plan(const EssConvDesc* d, EssConvPlan* plan) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(plan != nullptr, "conv: null plan");
  const ResolvedDesc rd = resolve_compute(d);
  make_plan(&rd.d, plan, rd.split);
  return ESS_OK;
}

extern "C" int ess_conv2d_s2d_preferred(const EssConvDesc* d) {
  if (validate(d) != ESS_OK || d->mode0 != ESS_SRC_S2D) return 0;
  return conv_bf16_s2d_preferred(d) ? 1 : 0;
}

extern "C" int ess_conv2d_pack_weights(const EssConvDesc* d, int w_kind, const float* w, const float* w2, void* packed,
                                       ess_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(w && packed, "pack_weights: null pointer");
  ESS_CHECK_ARG(d->epilogue != ESS_EPI_GRU_UR || w2, "pack_weights: GRU_UR needs the reset-gate weight as w2");
  ESS_CHECK_ARG(w_kind == ESS_W_CONV || w_kind == ESS_W_TRANSPOSED || w_kind == ESS_W_CONV5_S2D, "pack_weights: bad w_kind");
  ESS_CHECK_ARG((w_kind == ESS_W_CONV5_S2D) == (d->mode0 == ESS_SRC_S2D), "pack_weights: ESS_W_CONV5_S2D goes with an ESS_SRC_S2D descriptor, and only with one");
  const ResolvedDesc rd = resolve_compute(d);
  d = &rd.d;
  EssConvPlan pl;
  make_plan(d, &pl, rd.split);
  if (is_bf16(d)) return conv_bf16_pack_weights(d, pl, w_kind, w, w2, packed, (hipStream_t)stream, rd.split, rd.f16);
  const int64_t total = pl.packed_elems;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, w2,
                     (float*)packed, total, pl.cout_tile, pl.ck, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, d->epilogue,
                     d->hidden, w_kind);
  return ess_launch_status("pack_weights");
}

extern "C" int ess_conv2d_pack_weights_multi(const EssConvDesc* descs, const int32_t* w_kinds, const float* const* w,
                                             void* const* packed, int32_t count, ess_stream_t stream) {
  ESS_CHECK_ARG(descs && w_kinds && w && packed && count > 0, "pack_weights_multi: bad arguments");
  return conv_bf16_pack_weights_multi(descs, w_kinds, w, packed, count, (hipStream_t)stream);
}

extern "C" int ess_conv2d_pack_rows(const EssConvDesc* d, const float* v, const float* v2, float fill, float* packed,
                                    ess_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(v && packed, "pack_rows: null pointer");
  ESS_CHECK_ARG(d->epilogue != ESS_EPI_GRU_UR || v2, "pack_rows: GRU_UR needs the reset-gate vector as v2");
  const ResolvedDesc rd = resolve_compute(d);
  d = &rd.d;
  EssConvPlan pl;
  make_plan(d, &pl, rd.split);
  hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(pl.rows_padded, 256)), dim3(256), 0, (hipStream_t)stream, v, v2, fill,
                     packed, pl.rows_padded, d->epilogue, d->hidden, d->C_out);
  return ess_launch_status("pack_rows");
}

extern "C" int ess_conv2d_forward(const EssConvDesc* d, const void* src0, const void* src1, const void* packed_w,
                                  const float* scale, const float* shift, const void* residual, const float* aux0,
                                  const float* aux1, void* out, void* out2, void* out_bf16, ess_stream_t stream) {
  // ESS_FMT_F16_C8 output: the BF16_C8-output path with half elements (LINEAR epilogue; no pooled output, no out_split)
  EssConvDesc dcopy;
  bool out_f16 = false;
  if (d && d->fmt_out == ESS_FMT_F16_C8 && d->compute != ESS_COMPUTE_F16) {
    ESS_CHECK_ARG(d->act != ESS_ACT_SUMPOOL2 && d->out_split == 0 && (d->fmt_res == ESS_FMT_F32_NCHW || d->fmt_res == ESS_FMT_BF16_C8),
                  "conv: an F16_C8 output takes no SUMPOOL2 / out_split; a residual comes as BF16_C8");
    dcopy = *d;
    dcopy.fmt_out = ESS_FMT_BF16_C8;
    if (residual) dcopy.fmt_res = ESS_FMT_BF16_C8;
    d = &dcopy;
    out_f16 = true;
  }
  int rc = validate(d);
  if (rc) return rc;
  const ResolvedDesc rd = resolve_compute(d);
  d = &rd.d;
  ESS_CHECK_ARG(src0 && packed_w, "conv: null pointer");
  ESS_CHECK_ARG(d->mode0 != ESS_SRC_S2D || (d->fmt0 == ESS_FMT_BF16_C8 && d->fmt_out == ESS_FMT_BF16_C8), "conv: an ESS_SRC_S2D source is a BF16_C8 tensor, and so is the output");
  ESS_CHECK_ARG(out || (out_bf16 && d->out_split == 0 && (d->epilogue == ESS_EPI_LINEAR || d->epilogue == ESS_EPI_LSTM || d->epilogue == ESS_EPI_GRU_OUT)),
                "conv: `out` may only be NULL when the BF16_C8 copy is requested (LINEAR / LSTM / GRU_OUT epilogues)");
  ESS_CHECK_ARG(d->C1 == 0 || src1, "conv: second source missing");
  if (d->epilogue != ESS_EPI_LINEAR) ESS_CHECK_ARG(shift && !scale && !residual, "conv: a recurrent epilogue takes a bias (shift) and no scale / residual");
  if (d->epilogue == ESS_EPI_LSTM) ESS_CHECK_ARG(out2, "conv: the LSTM epilogue needs the second output (c')");
  // GRU_UR: r*h goes to out2 (fp32) and / or out_bf16 (BF16_C8); with h_prev = NULL (zeros) it may be dropped altogether
  if (d->epilogue == ESS_EPI_GRU_UR) ESS_CHECK_ARG(out2 || out_bf16 || !aux0, "conv: GRU_UR needs an output for r*h (out2 or out_bf16)");
  if (d->epilogue == ESS_EPI_GRU_OUT) ESS_CHECK_ARG(aux1, "conv: GRU_OUT needs u (aux1); h_prev (aux0) may be NULL = zeros");
  if (d->out_split > 0) ESS_CHECK_ARG(out2, "conv: out_split needs out2");
  if (out_bf16)
    ESS_CHECK_ARG(d->compute == ESS_COMPUTE_BF16 && d->out_split == 0, "conv: the BF16_C8 output copy exists for bf16 compute, no out_split");
  if (d->act == ESS_ACT_SUMPOOL2)
    ESS_CHECK_ARG(out && !scale && !residual && !out_bf16, "conv: SUMPOOL2 takes no scale / residual / BF16_C8 copy");
  if (d->fmt_out == ESS_FMT_BF16_C8) {
    ESS_CHECK_ARG(out && !out_bf16, "conv: a BF16_C8 output goes to `out` (no separate copy)");
    ESS_CHECK_ARG(((((uintptr_t)out) | ((uintptr_t)out2) | ((uintptr_t)residual)) & 15) == 0, "conv: BF16_C8 tensors must be 16-byte aligned");
  }
  ESS_CHECK_ARG(!residual || d->fmt_res == d->fmt_out, "conv: the residual must come in the output's format");
  EssConvPlan pl;
  make_plan(d, &pl, rd.split);
  ESS_CHECK_ARG(pl.lds_bytes <= 160 * 1024, "conv: LDS tile %d B exceeds 160 KiB", pl.lds_bytes);
  if (d->epilogue == ESS_EPI_GRU_UR && d->act == ESS_GRU_U_F16 && d->fmt_out == ESS_FMT_F32_C8)
    // an F16_C8 update gate is WRITTEN by the straight-line (update, reset) epilogue only (conv_epilogue in conv_common.h): its conditions.
    // (The candidate launch reads it in either epilogue form.)
    ESS_CHECK_ARG(shift && d->fmt0 == ESS_FMT_BF16_C8 && (d->hidden % (pl.cout_tile / 2)) == 0 && out && !out2 && (!aux0 || d->fmt_res == ESS_FMT_F32_C8),
                  "conv(GRU_UR): an F16_C8 update gate (ESS_GRU_U_F16 with channel-blocked outputs) needs BF16_C8 sources, a bias, hidden %% %d == 0, "
                  "no fp32 r*h output and a channel-blocked h_prev", pl.cout_tile / 2);
  const Geom g = choose_geom(d);
  ConvKArgs a{};
  a.src0 = (const float*)src0; a.src1 = (const float*)src1; a.wpk = packed_w;
  a.out_bf = out_bf16; a.fmt0 = d->fmt0; a.fmt1 = d->C1 ? d->fmt1 : d->fmt0; a.scale = scale; a.shift = shift; a.residual = (const float*)residual;
  a.aux0 = aux0; a.aux1 = aux1; a.out = (float*)out; a.out2 = (float*)out2; a.fmt_out = d->fmt_out; a.fmt_res = d->fmt_res;
#ifdef ESS_ABLATE  // diagnostic builds only (-DESS_ABLATE): ESS_WS_ABL=<bits> lets the 3x3 kernel skip loads / LDS writes / MFMAs / epilogue
  { static const int abl = [] { const char* b = getenv("ESS_WS_ABL"); return b ? atoi(b) & ~1 : 0; }(); a.deep = abl; }
#endif
  a.out_f16 = out_f16 ? 1 : 0;
  a.split = rd.split ? 1 : 0;
  a.f16 = rd.f16 ? 1 : 0;
  a.hilo = rd.hilo ? 1 : 0;
  if (rd.hilo && d->epilogue == ESS_EPI_LINEAR)
    ESS_CHECK_ARG(!residual && (d->C_out % pl.cout_tile) == 0 && d->fmt_out == ESS_FMT_BF16_C8, "conv(f16): a [hi | lo] output takes no residual and whole channel tiles");
  if (rd.hilo && d->epilogue == ESS_EPI_GRU_OUT)
    ESS_CHECK_ARG(out_bf16 && shift && d->fmt_res == ESS_FMT_F32_C8 && aux1 && (!out || d->fmt_out == ESS_FMT_F32_C8) && (d->hidden % pl.cout_tile) == 0,
                  "conv(f16, GRU_OUT): a [hi | lo] copy of h' exists in the straight-line form only (channel-blocked states, whole tiles of hidden channels)");
  if (rd.hilo && d->epilogue == ESS_EPI_LSTM)
    ESS_CHECK_ARG(out_bf16 && shift && d->fmt_out == ESS_FMT_F32_C8 && (!aux0 || d->fmt_res == ESS_FMT_F32_C8) && (d->hidden % (8 * (pl.cout_tile / 32))) == 0 && pl.cout_tile >= 64,
                  "conv(f16, LSTM): a [hi | lo] copy of h' exists in the lean form only (channel-blocked cell states, whole hidden blocks per tile)");
  a.N = d->N; a.Hin = d->H_in; a.Win = d->W_in; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0; a.mode1 = d->mode1;
  a.Cout = d->C_out; a.Hout = d->H_out; a.Wout = d->W_out; a.pad = d->pad;
  a.bwl = g.bwl; a.wxl = g.wxl; a.tiles_x = g.tiles_x; a.n_tiles = g.tiles_x * g.tiles_y; a.n_cout_tiles = pl.n_cout_tiles;
  a.IH = g.IH; a.IW = g.IW; a.row_pitch = g.row_pitch; a.par_off = g.par_off; a.plane = g.plane;
  a.ck = pl.ck; a.n_chunks = pl.n_chunks; a.act = d->act; a.hid = d->hidden; a.out_split = d->out_split;
  hipStream_t st = (hipStream_t)stream;
  if (is_bf16(d)) return conv_bf16_launch(d, pl, g, a, st);
  ESS_CHECK_ARG(g.IH * g.IW <= kpc32(d->ksize, d->stride) * 256, "conv: input tile of %d positions exceeds the staging capacity",
                g.IH * g.IW);
  ESS_CHECK_ARG((int64_t)(d->C0 > d->C1 ? d->C0 : d->C1) * d->H_in * d->W_in * 4 < (int64_t)1 << 31,
                "conv: one sample of a source must stay below 2 GiB (32-bit buffer offsets)");
  const dim3 grid((unsigned)(g.tiles_x * g.tiles_y * pl.n_cout_tiles * d->N));
  const int mb = pl.cout_tile / 32;
  const int key = d->ksize * 10 + d->stride;
  switch (key) {
    case 11: launch_mb<1, 1>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 12: launch_mb<1, 2>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 31: launch_mb<3, 1>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 32: launch_mb<3, 2>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 51: launch_mb<5, 1>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 52: launch_mb<5, 2>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 71: launch_mb<7, 1>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 72: launch_mb<7, 2>(mb, pl.ck, d->epilogue, grid, pl.lds_bytes, st, a); break;
    default: ess_set_error("conv: no kernel for k%d s%d", d->ksize, d->stride); return ESS_ENOTSUP;
  }
  return ess_launch_status("conv2d_forward");
}

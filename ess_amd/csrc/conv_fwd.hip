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

template <int KS, int S, int MB, int EPI>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int COT = MB * 32;
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
  const int CB = a.ck;
  float* in_t = smem;
  float* w_t = smem + CB * a.plane;
  const int wsz = KS * KS * CB * COT;

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

  const int iy0 = y0 * S - a.pad, ix0 = x0 * S - a.pad;
  const int Ctot = a.C0 + a.C1;

  for (int ch = 0; ch < a.n_chunks; ++ch) {
    __syncthreads();  // everyone finished reading the previous chunk
    // ---- stage the input tile of this channel chunk
    for (int cb = 0; cb < CB; ++cb) {
      const int c = ch * CB + cb;
      const bool first = c < a.C0;
      const float* sp = first ? a.src0 : a.src1;
      const int cc = first ? c : c - a.C0;
      const int Cs = first ? a.C0 : a.C1;
      const int mode = first ? a.mode0 : a.mode1;
      const bool cvalid = c < Ctot;
      const int sh = mode != ESS_SRC_DIRECT ? 1 : 0;
      const int Hp = a.Hin >> sh, Wp = a.Win >> sh;
      const float* pl = sp + ((size_t)n * Cs + cc) * Hp * Wp;
      float* dst = in_t + cb * a.plane;
      if (!cvalid) {  // zero padding channels of the last chunk (block-uniform branch)
        for (int iy = wave; iy < a.IH; iy += 4)
          for (int ix = lane; ix < a.IW; ix += 64)
            dst[iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix)] = 0.f;
        continue;
      }
      for (int iy = wave; iy < a.IH; iy += 4) {
        const int gy = iy0 + iy;
        bool yok = cvalid && gy >= 0 && gy < a.Hin;
        if (mode == ESS_SRC_ZERO_UP2) yok = yok && !(gy & 1);
        const float* row = pl + (size_t)(min(max(gy, 0), a.Hin - 1) >> sh) * Wp;
        for (int ix = lane; ix < a.IW; ix += 64) {
          const int gx = ix0 + ix;
          bool ok = yok && gx >= 0 && gx < a.Win;
          if (mode == ESS_SRC_ZERO_UP2) ok = ok && !(gx & 1);
          // unconditional load from a clamped address + select: no per-element branch, loads stay batched
          const float t = row[min(max(gx, 0), a.Win - 1) >> sh];
          const float v = ok ? t : 0.f;
          const int li = iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix);
          dst[li] = v;
        }
      }
    }
    // ---- stage the weight slab (one contiguous block in the packed layout)
    {
      const float* wsrc = (const float*)a.wpk + ((size_t)ct * a.n_chunks + ch) * wsz;
      for (int i = tid * 4; i < wsz; i += 1024) *(f32x4*)(w_t + i) = *(const f32x4*)(wsrc + i);
    }
    __syncthreads();
    // ---- MFMA over (tap, channel pair)
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
        const int toff = ky * a.row_pitch + (S == 2 ? (kx & 1) * a.par_off + (kx >> 1) : kx);
        const float* wp = w_t + (tap * CB + half) * COT + p;
        const float* ip = in_t + toff;
#pragma unroll 2
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

template <int KS, int S, int MB>
void launch_epi(int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (KS == 3 && S == 1) {
    switch (epi) {
      case ESS_EPI_LSTM: { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_LSTM>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_LSTM>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_UR: { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_UR>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_UR>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_OUT: { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_OUT>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_OUT>), grid, dim3(256), lds, st, a); } return;
      default: break;
    }
  }
  { ess_allow_lds(conv_f32_kernel<KS, S, MB, ESS_EPI_LINEAR>, lds); hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_LINEAR>), grid, dim3(256), lds, st, a); }
}

template <int KS, int S>
void launch_mb(int mb, int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (mb == 2) launch_epi<KS, S, 2>(epi, grid, lds, st, a);
  else launch_epi<KS, S, 1>(epi, grid, lds, st, a);
}

}  // namespace

extern "C" int ess_conv2d_plan(const EssConvDesc* d, EssConvPlan* plan) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(plan != nullptr, "conv: null plan");
  make_plan(d, plan);
  return ESS_OK;
}

extern "C" int ess_conv2d_pack_weights(const EssConvDesc* d, int w_kind, const float* w, const float* w2, void* packed,
                                       ess_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(w && packed, "pack_weights: null pointer");
  ESS_CHECK_ARG(d->epilogue != ESS_EPI_GRU_UR || w2, "pack_weights: GRU_UR needs the reset-gate weight as w2");
  ESS_CHECK_ARG(w_kind == ESS_W_CONV || w_kind == ESS_W_TRANSPOSED, "pack_weights: bad w_kind");
  EssConvPlan pl;
  make_plan(d, &pl);
  if (is_bf16(d)) return conv_bf16_pack_weights(d, pl, w_kind, w, w2, packed, (hipStream_t)stream);
  const int64_t total = pl.packed_elems;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, w2,
                     (float*)packed, total, pl.cout_tile, pl.ck, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, d->epilogue,
                     d->hidden, w_kind);
  return ess_launch_status("pack_weights");
}

extern "C" int ess_conv2d_pack_rows(const EssConvDesc* d, const float* v, const float* v2, float fill, float* packed,
                                    ess_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(v && packed, "pack_rows: null pointer");
  ESS_CHECK_ARG(d->epilogue != ESS_EPI_GRU_UR || v2, "pack_rows: GRU_UR needs the reset-gate vector as v2");
  EssConvPlan pl;
  make_plan(d, &pl);
  hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(pl.rows_padded, 256)), dim3(256), 0, (hipStream_t)stream, v, v2, fill,
                     packed, pl.rows_padded, d->epilogue, d->hidden, d->C_out);
  return ess_launch_status("pack_rows");
}

extern "C" int ess_conv2d_forward(const EssConvDesc* d, const float* src0, const float* src1, const void* packed_w,
                                  const float* scale, const float* shift, const float* residual, const float* aux0,
                                  const float* aux1, float* out, float* out2, ess_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(src0 && packed_w && out, "conv: null pointer");
  ESS_CHECK_ARG(d->C1 == 0 || src1, "conv: second source missing");
  if (d->epilogue == ESS_EPI_LSTM || d->epilogue == ESS_EPI_GRU_UR)
    ESS_CHECK_ARG(shift && out2, "conv: recurrent epilogue needs bias and second output");
  if (d->epilogue == ESS_EPI_GRU_OUT) ESS_CHECK_ARG(aux0 && aux1, "conv: GRU_OUT needs h_prev and u");
  if (d->out_split > 0) ESS_CHECK_ARG(out2, "conv: out_split needs out2");
  EssConvPlan pl;
  make_plan(d, &pl);
  ESS_CHECK_ARG(pl.lds_bytes <= 160 * 1024, "conv: LDS tile %d B exceeds 160 KiB", pl.lds_bytes);
  const Geom g = choose_geom(d);
  ConvKArgs a{};
  a.src0 = src0; a.src1 = src1; a.wpk = packed_w; a.scale = scale; a.shift = shift; a.residual = residual;
  a.aux0 = aux0; a.aux1 = aux1; a.out = out; a.out2 = out2;
  a.N = d->N; a.Hin = d->H_in; a.Win = d->W_in; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0; a.mode1 = d->mode1;
  a.Cout = d->C_out; a.Hout = d->H_out; a.Wout = d->W_out; a.pad = d->pad;
  a.bwl = g.bwl; a.wxl = g.wxl; a.tiles_x = g.tiles_x; a.n_tiles = g.tiles_x * g.tiles_y; a.n_cout_tiles = pl.n_cout_tiles;
  a.IH = g.IH; a.IW = g.IW; a.row_pitch = g.row_pitch; a.par_off = g.par_off; a.plane = g.plane;
  a.ck = pl.ck; a.n_chunks = pl.n_chunks; a.act = d->act; a.hid = d->hidden; a.out_split = d->out_split;
  hipStream_t st = (hipStream_t)stream;
  if (is_bf16(d)) return conv_bf16_launch(d, pl, g, a, st);
  const dim3 grid((unsigned)(g.tiles_x * g.tiles_y * pl.n_cout_tiles * d->N));
  const int mb = pl.cout_tile / 32;
  const int key = d->ksize * 10 + d->stride;
  switch (key) {
    case 11: launch_mb<1, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 12: launch_mb<1, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 31: launch_mb<3, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 32: launch_mb<3, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 51: launch_mb<5, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 52: launch_mb<5, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 71: launch_mb<7, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 72: launch_mb<7, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    default: ess_set_error("conv: no kernel for k%d s%d", d->ksize, d->stride); return ESS_ENOTSUP;
  }
  return ess_launch_status("conv2d_forward");
}

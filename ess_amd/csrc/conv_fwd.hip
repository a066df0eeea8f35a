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
#include "common.h"

namespace {

constexpr int NBW = 2;  // pixel blocks per wave

struct ConvKArgs {
  const float* src0;
  const float* src1;
  const float* wpk;
  const float* scale;
  const float* shift;
  const float* residual;
  const float* aux0;
  const float* aux1;
  float* out;
  float* out2;
  int N, Hin, Win, C0, C1, mode0, mode1;
  int Cout, Hout, Wout, pad;
  int bwl, wxl, tiles_x;
  int IH, IW, row_pitch, par_off, plane;
  int ck, n_chunks;
  int act, hid, out_split;
};

template <int KS, int S, int MB, int EPI>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int COT = MB * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int BW = 1 << a.bwl, WX = 1 << a.wxl, RB = 32 >> a.bwl;
  const int TW = WX << a.bwl, TH = (4 >> a.wxl) * NBW * RB;
  const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int ct = blockIdx.y, n = blockIdx.z;
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
      for (int iy = wave; iy < a.IH; iy += 4) {
        const int gy = iy0 + iy;
        bool yok = cvalid && gy >= 0 && gy < a.Hin;
        if (mode == ESS_SRC_ZERO_UP2) yok = yok && !(gy & 1);
        const float* row = pl + (size_t)(gy >> sh) * Wp;
        for (int ix = lane; ix < a.IW; ix += 64) {
          const int gx = ix0 + ix;
          bool ok = yok && gx >= 0 && gx < a.Win;
          if (mode == ESS_SRC_ZERO_UP2) ok = ok && !(gx & 1);
          float v = 0.f;
          if (ok) v = row[gx >> sh];
          const int li = iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix);
          dst[li] = v;
        }
      }
    }
    // ---- stage the weight slab (one contiguous block in the packed layout)
    {
      const float* wsrc = a.wpk + ((size_t)ct * a.n_chunks + ch) * wsz;
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

  // ---- epilogue
  const int x = x0 + lx;
  const size_t HW = (size_t)a.Hout * a.Wout;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int y = y0 + ly[nb];
    if (y >= a.Hout || x >= a.Wout) continue;
    const size_t pix = (size_t)y * a.Wout + x;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int rowbase = ct * COT + mb * 32;
      if constexpr (EPI == ESS_EPI_LINEAR || EPI == ESS_EPI_GRU_OUT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = rowbase + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (co >= a.Cout) continue;
          float v = acc[mb][nb][r];
          if (a.scale) v *= a.scale[co];
          if (a.shift) v += a.shift[co];
          if constexpr (EPI == ESS_EPI_LINEAR) {
            const size_t idx = ((size_t)n * a.Cout + co) * HW + pix;
            if (a.residual) v += a.residual[idx];
            if (a.act == ESS_ACT_RELU) v = fmaxf(v, 0.f);
            else if (a.act == ESS_ACT_SIGMOID) v = ess_sigmoid(v);
            else if (a.act == ESS_ACT_TANH) v = tanhf(v);
            if (a.out_split > 0) {
              if (co < a.out_split) a.out[((size_t)n * a.out_split + co) * HW + pix] = v;
              else a.out2[((size_t)n * (a.Cout - a.out_split) + (co - a.out_split)) * HW + pix] = v;
            } else {
              a.out[idx] = v;
            }
          } else {  // GRU candidate: h' = h (1-u) + tanh(.) u
            const size_t idx = ((size_t)n * a.hid + co) * HW + pix;
            const float o = tanhf(v), u = a.aux1[idx], h = a.aux0[idx];
            a.out[idx] = h * (1.f - u) + o * u;
          }
        }
      } else if constexpr (EPI == ESS_EPI_LSTM) {
        // packed row 8*g + j of this 32-row block = gate g (in, remember, out, cell) of hidden hb*8 + j
        const int hb = ct * MB + mb;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int hc = hb * 8 + 4 * half + jj;
          if (hc >= a.hid) continue;
          const int pr = rowbase + 4 * half + jj;
          const float gi = ess_sigmoid(acc[mb][nb][jj] + a.shift[pr]);
          const float gf = ess_sigmoid(acc[mb][nb][4 + jj] + a.shift[pr + 8]);
          const float go = ess_sigmoid(acc[mb][nb][8 + jj] + a.shift[pr + 16]);
          const float gc = tanhf(acc[mb][nb][12 + jj] + a.shift[pr + 24]);
          const size_t idx = ((size_t)n * a.hid + hc) * HW + pix;
          const float cprev = a.aux0 ? a.aux0[idx] : 0.f;
          const float cn = gf * cprev + gi * gc;
          a.out2[idx] = cn;
          a.out[idx] = go * tanhf(cn);
        }
      } else {  // ESS_EPI_GRU_UR
        // packed row 8*q + j: gate q&1 (0 update, 1 reset) of hidden hb*16 + (q>>1)*8 + j
        const int hb = ct * MB + mb;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int hc = hb * 16 + q2 * 8 + 4 * half + jj;
            if (hc >= a.hid) continue;
            const int pr = rowbase + 16 * q2 + 4 * half + jj;
            const float u = ess_sigmoid(acc[mb][nb][8 * q2 + jj] + a.shift[pr]);
            const float rr = ess_sigmoid(acc[mb][nb][8 * q2 + 4 + jj] + a.shift[pr + 8]);
            const size_t idx = ((size_t)n * a.hid + hc) * HW + pix;
            const float h = a.aux0 ? a.aux0[idx] : 0.f;
            a.out[idx] = u;
            a.out2[idx] = rr * h;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
struct Geom {
  int bwl, wxl, TW, TH, tiles_x, tiles_y, IH, IW, row_pitch, par_off, plane;
};

int pick_ck(const EssConvDesc* d) {
  const int cin = d->C0 + d->C1;
  int ck = d->ksize >= 7 ? 2 : (d->ksize == 5 ? 4 : 8);
  while (ck > 2 && ck / 2 >= cin) ck /= 2;
  return ck;
}

int pick_mb(const EssConvDesc* d) { return d->C_out > 32 ? 2 : 1; }

Geom choose_geom(const EssConvDesc* d) {
  Geom best{};
  double best_cost = 1e300;
  const int KS = d->ksize, S = d->stride;
  for (int bwl = 5; bwl >= 3; --bwl) {
    for (int wxl = 0; wxl <= 2; ++wxl) {
      const int BW = 1 << bwl, RB = 32 >> bwl;
      const int TW = BW << wxl, TH = (4 >> wxl) * NBW * RB;
      const int tx = ceil_div(d->W_out, TW), ty = ceil_div(d->H_out, TH);
      const int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
      // padded MACs (dominant) + a small halo/staging term; prefer wide blocks on ties
      const double cost = (double)tx * ty * TW * TH * (1.0 + 0.02 * (double)(IH * IW) / (TH * TW * S * S)) +
                          1e-3 * (5 - bwl);
      if (cost < best_cost) {
        best_cost = cost;
        Geom g{};
        g.bwl = bwl; g.wxl = wxl; g.TW = TW; g.TH = TH; g.tiles_x = tx; g.tiles_y = ty; g.IH = IH; g.IW = IW;
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
        best = g;
      }
    }
  }
  return best;
}

int validate(const EssConvDesc* d) {
  ESS_CHECK_ARG(d != nullptr, "conv: null descriptor");
  ESS_CHECK_ARG(d->N > 0 && d->H_in > 0 && d->W_in > 0 && d->C0 > 0 && d->C1 >= 0 && d->C_out > 0, "conv: bad extents");
  ESS_CHECK_ARG(d->ksize == 1 || d->ksize == 3 || d->ksize == 5 || d->ksize == 7, "conv: ksize %d unsupported", d->ksize);
  ESS_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
  ESS_CHECK_ARG(d->H_out == (d->H_in + 2 * d->pad - d->ksize) / d->stride + 1 &&
                    d->W_out == (d->W_in + 2 * d->pad - d->ksize) / d->stride + 1,
                "conv: output extent %dx%d inconsistent with input %dx%d k%d s%d p%d", d->H_out, d->W_out, d->H_in,
                d->W_in, d->ksize, d->stride, d->pad);
  for (int s = 0; s < 2; ++s) {
    const int m = s ? d->mode1 : d->mode0;
    ESS_CHECK_ARG(m >= 0 && m <= 2, "conv: bad source mode");
    if (m != ESS_SRC_DIRECT) ESS_CHECK_ARG(!(d->H_in & 1) && !(d->W_in & 1), "conv: x2 source needs even extent");
  }
  ESS_CHECK_ARG(d->epilogue >= 0 && d->epilogue <= 3, "conv: bad epilogue");
  if (d->epilogue == ESS_EPI_LSTM) ESS_CHECK_ARG(d->C_out == 4 * d->hidden, "conv: LSTM needs C_out = 4*hidden");
  if (d->epilogue == ESS_EPI_GRU_UR) ESS_CHECK_ARG(d->C_out == 2 * d->hidden, "conv: GRU_UR needs C_out = 2*hidden");
  if (d->epilogue == ESS_EPI_GRU_OUT) ESS_CHECK_ARG(d->C_out == d->hidden, "conv: GRU_OUT needs C_out = hidden");
  if (d->epilogue != ESS_EPI_LINEAR)
    ESS_CHECK_ARG(d->ksize == 3 && d->stride == 1 && d->out_split == 0, "conv: recurrent epilogues are 3x3 s1");
  ESS_CHECK_ARG(d->out_split >= 0 && d->out_split < d->C_out, "conv: bad out_split");
  return ESS_OK;
}

// rows of the packed weight matrix that exist for this epilogue (before padding to the tile)
int packed_rows(const EssConvDesc* d) {
  switch (d->epilogue) {
    case ESS_EPI_LSTM: return ceil_div(d->hidden, 8) * 32;
    case ESS_EPI_GRU_UR: return ceil_div(d->hidden, 16) * 32;
    default: return d->C_out;
  }
}

void make_plan(const EssConvDesc* d, EssConvPlan* pl) {
  const int mb = pick_mb(d);
  pl->cout_tile = mb * 32;
  pl->ck = pick_ck(d);
  pl->n_chunks = ceil_div(d->C0 + d->C1, pl->ck);
  pl->n_cout_tiles = ceil_div(packed_rows(d), pl->cout_tile);
  pl->rows_padded = pl->n_cout_tiles * pl->cout_tile;
  pl->packed_elems = (int64_t)pl->rows_padded * pl->n_chunks * pl->ck * d->ksize * d->ksize;
  const Geom g = choose_geom(d);
  pl->lds_bytes = (pl->ck * g.plane + d->ksize * d->ksize * pl->ck * pl->cout_tile) * 4;
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

__global__ void pack_weights_kernel(const float* w, const float* w2, float* out, int64_t total, int cot, int ck,
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

__global__ void pack_rows_kernel(const float* v, const float* v2, float fill, float* out, int rows_padded, int epi,
                                 int hid, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_padded) return;
  int sel;
  const int row = map_row(i, epi, hid, cout, &sel);
  out[i] = row >= 0 ? (sel ? v2[row] : v[row]) : fill;
}

template <int KS, int S, int MB>
void launch_epi(int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (KS == 3 && S == 1) {
    switch (epi) {
      case ESS_EPI_LSTM: hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_LSTM>), grid, dim3(256), lds, st, a); return;
      case ESS_EPI_GRU_UR: hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_UR>), grid, dim3(256), lds, st, a); return;
      case ESS_EPI_GRU_OUT: hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_GRU_OUT>), grid, dim3(256), lds, st, a); return;
      default: break;
    }
  }
  hipLaunchKernelGGL((conv_f32_kernel<KS, S, MB, ESS_EPI_LINEAR>), grid, dim3(256), lds, st, a);
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

extern "C" int ess_conv2d_pack_weights(const EssConvDesc* d, int w_kind, const float* w, const float* w2, float* packed,
                                       ess_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  ESS_CHECK_ARG(w && packed, "pack_weights: null pointer");
  ESS_CHECK_ARG(d->epilogue != ESS_EPI_GRU_UR || w2, "pack_weights: GRU_UR needs the reset-gate weight as w2");
  ESS_CHECK_ARG(w_kind == ESS_W_CONV || w_kind == ESS_W_TRANSPOSED, "pack_weights: bad w_kind");
  EssConvPlan pl;
  make_plan(d, &pl);
  const int64_t total = pl.packed_elems;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, w2,
                     packed, total, pl.cout_tile, pl.ck, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, d->epilogue,
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

extern "C" int ess_conv2d_forward(const EssConvDesc* d, const float* src0, const float* src1, const float* packed_w,
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
  ESS_CHECK_ARG(pl.lds_bytes <= 64 * 1024, "conv: LDS tile %d B exceeds 64 KiB", pl.lds_bytes);
  const Geom g = choose_geom(d);
  ConvKArgs a{};
  a.src0 = src0; a.src1 = src1; a.wpk = packed_w; a.scale = scale; a.shift = shift; a.residual = residual;
  a.aux0 = aux0; a.aux1 = aux1; a.out = out; a.out2 = out2;
  a.N = d->N; a.Hin = d->H_in; a.Win = d->W_in; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0; a.mode1 = d->mode1;
  a.Cout = d->C_out; a.Hout = d->H_out; a.Wout = d->W_out; a.pad = d->pad;
  a.bwl = g.bwl; a.wxl = g.wxl; a.tiles_x = g.tiles_x;
  a.IH = g.IH; a.IW = g.IW; a.row_pitch = g.row_pitch; a.par_off = g.par_off; a.plane = g.plane;
  a.ck = pl.ck; a.n_chunks = pl.n_chunks; a.act = d->act; a.hid = d->hidden; a.out_split = d->out_split;
  const dim3 grid(g.tiles_x * g.tiles_y, pl.n_cout_tiles, d->N);
  const int mb = pl.cout_tile / 32;
  hipStream_t st = (hipStream_t)stream;
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

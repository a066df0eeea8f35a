/*
 * ess_hip.h -- C ABI of libess_hip.so: the MI355X (gfx950) kernels under the ESS hot path.
 *
 * The reference (uzh-rpg/ess) has no FFI: its hot path is PyTorch modules calling cuDNN through
 * torch ops.  Each entry point below names the torch op sequence of the reference it replaces
 * (file:line under the reference root) so that a maintainer can bind it from the reference's own
 * nn.Module.forward (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions:
 *   - all pointers are DEVICE pointers into caller-owned storage (PyTorch caching allocator);
 *     nothing is allocated, freed or retained by the library; workspaces are caller-provided;
 *   - tensors are dense NCHW fp32 unless stated; labels are int64;
 *   - `stream` is the caller's hipStream_t (torch.cuda.current_stream().cuda_stream); the library
 *     never synchronises, never calls hipSetDevice and keeps no mutable global state that can change a result (re-entrant).
 *     What it does keep: read-once constants (environment tuning variables, the device's compute-unit count, per-kernel LDS
 *     opt-ins) and the documented process-wide switches of ess_tuning_set, which select between kernels of equal arithmetic
 *     ("conv_wide": bit-identical results; "in_small_threads": equal up to the summation order of the norm statistics);
 *   - return value: 0 on success, negative ESS_E* otherwise; ess_last_error() gives a thread-local
 *     message.  Nothing throws across the ABI.
 */
#ifndef ESS_HIP_H
#define ESS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ess_stream_t; /* hipStream_t */

enum { ESS_OK = 0, ESS_EINVAL = -22, ESS_ENOTSUP = -95, ESS_ELAUNCH = -5 };

/* how a conv source is read (fused into the LDS tile load) */
enum { ESS_SRC_DIRECT = 0, ESS_SRC_NEAREST_UP2 = 1, ESS_SRC_ZERO_UP2 = 2,
       ESS_SRC_S2D = 3 /* space-to-depth view of a BF16_C8 tensor stored at (2 H_in, 2 W_in) with C0 / 4 channels: the virtual channels
                        * of parity class q = (row parity) + 2 (column parity) at virtual pixel (y, x) are the stored channels at
                        * (2y + (q & 1), 2x + (q >> 1)); their ORDER (which virtual channel index is which class / stored channel) is
                        * internal to the ESS_W_CONV5_S2D weight pack and the kernel that reads it.  With weights
                        * packed as ESS_W_CONV5_S2D a 3x3 / stride 1 / pad 1 descriptor over this view IS the 5x5 / stride 2 / pad 2
                        * convolution of the stored tensor (nn.Conv2d(k=5, s=2, p=2) of the frozen encoder,
                        * e2vid/model/submodules.py:176-186, e2vid/model/unet.py:40-47): mode0 only, C1 = 0, fmt0 = fmt_out =
                        * ESS_FMT_BF16_C8, LINEAR epilogue, bf16 compute, C0 % 128 == 0, C_out % 64 == 0.                       */ };
/* conv epilogues (fused into the accumulator write-out) */
enum {
  ESS_EPI_LINEAR = 0,  /* y = act(acc*scale + shift (+ residual))                                   */
  ESS_EPI_LSTM = 1,    /* ConvLSTM gates -> (h', c')            e2vid/model/submodules.py:212-230     */
  ESS_EPI_GRU_UR = 2,  /* ConvGRU update/reset -> (u, r*h)      e2vid/model/submodules.py:268-270     */
  ESS_EPI_GRU_OUT = 3  /* ConvGRU candidate -> h'               e2vid/model/submodules.py:270-271     */
};
/* output transform of the LINEAR epilogue.  SUMPOOL2: no activation, and the first output (channels < out_split, or all
 * of them) is written as the 2x2 sum of its pixels, [N][C][H_out/2][W_out/2] -- the data-gradient of a nearest-x2-upsampled
 * source (mode ESS_SRC_NEAREST_UP2 of the forward convolution) without the full-resolution tensor in between.        */
enum { ESS_ACT_NONE = 0, ESS_ACT_RELU = 1, ESS_ACT_SIGMOID = 2, ESS_ACT_TANH = 3, ESS_ACT_SUMPOOL2 = 4 };
/* arithmetic of the convolution contraction.  Tensors in HBM are fp32 either way; BF16 rounds the MFMA operands
 * (activations while staging the LDS tile, weights at pack time) to bfloat16 and accumulates in fp32.         */
enum { ESS_COMPUTE_FP32 = 0, ESS_COMPUTE_BF16 = 1,
       /* split-operand bf16 ("bf16x3"): fp32 tensors; every operand x of a 3x3 / stride-1 contraction (forward, data-gradient,
        * recurrent gates, weight gradient) enters the matrix cores as hi = bf16(x), lo = bf16(x - hi), and a product is
        * w_hi x_hi + w_hi x_lo + w_lo x_hi in fp32 accumulators: ~2^-16 relative operand error against bf16's 2^-8, at three bf16
        * MFMAs per product (the exact fp32 MFMA costs sixteen).  Every other convolution of such a descriptor runs the exact-fp32
        * kernels.  The parity-grade configuration with a matrix-core-rate step.                                          */
       ESS_COMPUTE_BF16X3 = 2,
       /* IEEE-half operands (round 6; the forward arithmetic of the "mixed" configuration): the matrix cores run
        * v_mfma_f32_32x32x16_f16 at the bf16 rate with 11 instead of 8 significant operand bits; fp32 accumulators.  Every 16-bit
        * tensor of such a call is an ESS_FMT_F16_C8 tensor (sources, residual, LINEAR output, the recurrent epilogues' copies), the
        * weights are rounded to half at pack time, conversions saturate at +-65504 (NaN kept).  Forward forms only (no SUMPOOL2, no
        * zero-inserted source, no weight gradient): gradients stay bfloat16 -- their magnitudes (1e-7 for a mean loss over 2.5 M
        * pixels) are below half's normal range.  fp32 NCHW sources only for the 2-channel 5x5 head.                         */
       ESS_COMPUTE_F16 = 3 };
/* storage format of a convolution source.  BF16_C8 = the "staging copy" a producing kernel can emit next to its
 * fp32 NCHW output (out_bf16 of ess_conv2d_forward, or ess_to_bf16_c8): bfloat16 [N][ceil(C/8)][H][W][8], i.e. the
 * 8 channels of a pixel are one 16-byte vector = one MFMA K-fragment; channels past C are zero.  A consumer conv
 * stages it with plain 16-byte copies: 4x fewer cache-line touches and half the bytes of the fp32 NCHW path.      */
/* ESS_FMT_F16_C8: the BF16_C8 layout with IEEE half elements.  ONLY as a convolution OUTPUT (fmt_out, LINEAR epilogue) that a norm
 * kernel reads (x of ess_instnorm_* / ess_batchnorm_train_*_c8 with x_f16 = 1): pre-normalisation tensors keep 11 significant bits. */
/* ESS_COMPUTE_F16 convolutions read and write F16_C8 tensors throughout.  ESS_FMT_F16_C8_HILO (fmt_out of such a LINEAR convolution; act in
 * {none, relu}, no residual, no out_split, C_out % 64 == 0): the output is [N][2 ceil(C/8)][H][W][8] halfs -- blocks [0, CB) hold
 * hi = half(v), blocks [CB, 2 CB) lo = half(v - hi), ~22 significant bits in the pair.  A consumer reads it as a plain F16_C8 source of 2 C
 * channels against a weight whose input columns are repeated ([w | w]): w (hi + lo) on the half matrix cores.  Used where an
 * activation's mean is large against its spread: the encoder convolution in front of a recurrent block, the event latents.   */
enum { ESS_FMT_F32_NCHW = 0, ESS_FMT_BF16_C8 = 1, ESS_FMT_F16_C8 = 3, ESS_FMT_F16_C8_HILO = 4,
       ESS_FMT_F32_C8 = 2 /* fp32 [N][ceil(C/8)][H][W][8]: ConvLSTM cell / ConvGRU hidden states between time steps (recurrent epilogues only) */ };
/* ConvGRU: the update gate u between the (update, reset) kernel and the candidate kernel (EssConvDesc.act of the two GRU epilogues).
 * ESS_GRU_U_F16: u is rounded to IEEE half (11 significant bits of a value in (0, 1); the gates come from bf16 operands) -- with
 * fmt_out / fmt_res = ESS_FMT_F32_C8 the tensor itself is an F16_C8 tensor ([N][hid/8][H][W][8] halfs: half the bytes of the launch
 * pair's most bandwidth-bound operand), with fp32 NCHW states the fp32 tensor carries the rounded value, so that a sequence gives the
 * same bits whichever storage its steps use.  bf16 compute only.  Reference: e2vid/model/submodules.py:255-273 (`update`).          */
enum { ESS_GRU_U_F32 = 0, ESS_GRU_U_F16 = 1 };
/* ConvLSTM with ESS_COMPUTE_F16 (EssConvDesc.act of the LSTM epilogue; 0 otherwise): the half copy of h' (out_bf16) leaves as a
 * [hi | lo] pair, [N][2 hid/8][H][W][8] (see ESS_FMT_F16_C8_HILO) -- the event latents the semantic decoder reads.  Lean form only
 * (channel-blocked cell states, hidden % 16 == 0).                                                                       */
enum { ESS_LSTM_H_HILO = 1 };
/* ... and the ConvGRU candidate launch (ESS_EPI_GRU_OUT, ESS_COMPUTE_F16): OR'ed into act next to ESS_GRU_U_*; straight-line form only
 * (channel-blocked states, hidden % 64 == 0).                                                                                     */
enum { ESS_GRU_H_HILO = 2 };
/* weight sources for ess_conv2d_pack_weights */
enum {
  ESS_W_CONV = 0,        /* nn.Conv2d weight [C_out][C_in][k][k]                                     */
  ESS_W_TRANSPOSED = 1,  /* weight [C_in][C_out][k][k] used spatially flipped: nn.ConvTranspose2d
                            forward, or the data-gradient of a Conv2d (pass its weight)              */
  ESS_W_ROWS = 2,        /* ess_conv2d_pack_weights_multi only: the job's `w` is a per-output-channel vector (a bias,
                            C_out floats) and `packed` the float buffer of ess_conv2d_pack_rows (fill 0) -- the
                            biases of the trainable convolutions ride in the weights' re-pack launch       */
  ESS_W_CONV5_S2D = 3    /* ess_conv2d_pack_weights with an ESS_SRC_S2D descriptor (and only then): w is the 5x5 / stride-2
                            convolution's weight [C_out][C0 / 4][5][5]; packed as the 3x3 taps of the four parity classes
                            (tap (ty, tx) of class (py, px) = w[..][2 ty + py][2 tx + px], absent where that index is 5)  */
};

/* One 2-D convolution over the channel-concatenation of up to two sources.
 * Replaces nn.Conv2d / torch.cat / F.interpolate(nearest) / BatchNorm2d(eval) / activation chains:
 *   ConvLayer.forward            e2vid/model/submodules.py:24-31
 *   ResidualBlock.forward        e2vid/model/submodules.py:157-172
 *   ConvLSTM.forward             e2vid/model/submodules.py:190-230
 *   ConvGRU.forward              e2vid/model/submodules.py:255-273
 *   ReLUINSConv2d / INSResBlock convs, cat+nearest-up of SemSegE2VID.forward
 *                                models/style_networks.py:69-88,158-193                             */
typedef struct EssConvDesc {
  int32_t N;
  int32_t H_in, W_in;   /* extent the filter slides over (i.e. AFTER per-source x2 upsampling)       */
  int32_t C0, C1;       /* channels of source 0 / source 1 (C1 = 0: single source)                   */
  int32_t mode0, mode1; /* ESS_SRC_*: a *_UP2 source is stored at (H_in/2, W_in/2)                   */
  int32_t C_out, H_out, W_out;
  int32_t ksize, stride, pad;
  int32_t epilogue;     /* ESS_EPI_*                                                                 */
  int32_t act;          /* ESS_ACT_* (LINEAR epilogue); GRU epilogues: ESS_GRU_U_* -- how the update gate u travels
                           from the GRU_UR launch to the GRU_OUT launch (both launches carry the same value)  */
  int32_t hidden;       /* recurrent epilogues: hidden channels (C_out = 4*hidden LSTM, 2*hidden
                           GRU_UR, hidden GRU_OUT)                                                   */
  int32_t out_split;    /* LINEAR: >0 writes channels [0,out_split) to `out` and the rest to `out2`
                           (data-gradient of a two-source conv)                                      */
  int32_t compute;      /* ESS_COMPUTE_*                                                             */
  int32_t fmt0, fmt1;   /* ESS_FMT_* of source 0 / 1.  BF16_C8 needs bf16 compute, a 1x1 / 3x3 (any source mode) or a
                           5x5 (DIRECT sources) filter, and the same format for both sources of a concat        */
  int32_t fmt_out;      /* ESS_FMT_BF16_C8 (LINEAR epilogue, bf16 compute): `out` -- and `out2` of an out_split --
                           ARE BF16_C8 tensors and no fp32 tensor is written: the stored form of the trainable
                           networks' activations and activation gradients in the bf16 configuration.  With
                           ESS_ACT_SUMPOOL2 the first output is the pooled BF16_C8 tensor.  out_split % 8 == 0.       */
  int32_t fmt_res;      /* format of `residual` (BF16_C8 only together with a BF16_C8 output).
                           LSTM epilogue: fmt_res = format of aux0 (the cell state c), fmt_out = format of out / out2
                           (h', c'): ESS_FMT_F32_NCHW or ESS_FMT_F32_C8 (a lane's 4 channels of a pixel are one 16-byte
                           access; for states that only travel to the next time step).
                           GRU epilogues: fmt_res = format of aux0 (h_prev) and aux1 (u), fmt_out = format of the fp32
                           outputs (GRU_UR: u, r*h; GRU_OUT: h'), each ESS_FMT_F32_NCHW or ESS_FMT_F32_C8               */
} EssConvDesc;

typedef struct EssConvPlan {
  int32_t cout_tile;    /* output channels per workgroup (32 or 64)                                  */
  int32_t ck;           /* input channels per LDS chunk                                              */
  int32_t n_chunks, n_cout_tiles;
  int64_t packed_elems; /* elements in the packed weight buffer (fp32 or bf16 by `compute`)          */
  int64_t packed_bytes; /* size of that buffer                                                       */
  int32_t rows_padded;  /* n_cout_tiles * cout_tile = length of packed scale/shift vectors           */
  int32_t lds_bytes;
} EssConvPlan;

const char* ess_last_error(void);
int ess_version(void);

int ess_conv2d_plan(const EssConvDesc* d, EssConvPlan* plan);

/* 1 when `d` is a valid ESS_SRC_S2D descriptor AND the space-to-depth form is the faster way to run that 5x5 / stride-2 convolution
 * on this device (its one-workgroup-per-CU tiles need a launch of at least 3/4 of the compute units' worth of tiles: B >= 4 at the
 * DSEC shape; below that the tap-paired 5x5 kernel wins), else 0.  The caller picks the descriptor (and the weight pack) by it:
 * ConvLayer.forward of the frozen encoder, e2vid/model/submodules.py:7-31,176-186.  NOTE: the two forms add the same products in
 * different orders, so a caller that follows this answer gets results that depend on N and on the device's compute-unit count in
 * the last bf16 / half bit (training batch vs a B < 4 validation or streaming call); pin one form per model where that matters.  */
int ess_conv2d_s2d_preferred(const EssConvDesc* d);

/* Re-layout a weight tensor for ess_conv2d_forward (tile-major, epilogue row permutation applied).
 * w_kind ESS_W_CONV: w is [C_out][C0+C1][k][k]; ESS_W_TRANSPOSED: w is [C0+C1][C_out][k][k].
 * For ESS_EPI_GRU_UR pass w = update_gate.weight and w2 = reset_gate.weight.                        */
int ess_conv2d_pack_weights(const EssConvDesc* d, int w_kind, const float* w, const float* w2,
                            void* packed, ess_stream_t stream);
/* ess_conv2d_pack_weights for `count` tensors in one launch (the re-pack of every trainable convolution after an optimiser
 * step).  bf16 compute, LINEAR epilogue, 1x1 / 3x3 / 7x7 layouts (and ESS_W_ROWS jobs) only; anything else is refused and nothing is launched.
 * descs / w_kinds / w / packed: host arrays of `count` entries.                                                      */
int ess_conv2d_pack_weights_multi(const EssConvDesc* descs, const int32_t* w_kinds, const float* const* w,
                                  void* const* packed, int32_t count, ess_stream_t stream);

/* Same permutation/padding for a per-output-channel vector (bias, folded BN scale/shift).
 * v2: second vector for GRU_UR (reset gate); fill: value for padded rows.                           */
int ess_conv2d_pack_rows(const EssConvDesc* d, const float* v, const float* v2, float fill,
                         float* packed, ess_stream_t stream);

/* src1 may be NULL when C1 == 0.  scale/shift: packed (ess_conv2d_pack_rows) or NULL.
 * residual: [N][C_out][H_out][W_out] or NULL (LINEAR).
 * aux0: LSTM c_prev | GRU_UR h_prev | GRU_OUT h_prev (NULL = zeros: first time step) ; aux1: GRU_OUT u.
 * out : LINEAR y | LSTM h' | GRU_UR u | GRU_OUT h' ;   out2: LSTM c' | GRU_UR r*h (fp32) | LINEAR split.
 * Recurrent epilogues take the bias as `shift`, no scale, no residual.                              */
int ess_conv2d_forward(const EssConvDesc* d, const void* src0, const void* src1, const void* packed_w,
                       const float* scale, const float* shift, const void* residual, const float* aux0,
                       const float* aux1, void* out, void* out2, void* out_bf16, ess_stream_t stream);
/* out / out2 / residual: fp32 NCHW, or BF16_C8 tensors when d->fmt_out (d->fmt_res) says so.                    */
/* src0/src1: fp32 NCHW or bf16 C8 per d->fmt0/fmt1.  out_bf16 (nullable): additionally receives `out` (LINEAR y,
 * LSTM h', GRU_OUT h') as a BF16_C8 tensor [N][ceil(C/8)][H_out][W_out][8] for the next convolution to stage from
 * (bf16 compute only; not with out_split).  With out_bf16 given, `out` may be NULL for the LINEAR, LSTM and GRU_OUT
 * epilogues: the fp32 tensor is then not written at all (a producer whose only consumer stages from the copy).
 * GRU_UR: out_bf16 receives r*h as a BF16_C8 tensor -- the second source of the candidate convolution (reference
 * e2vid/model/submodules.py:268-270) -- and out2 may then be NULL; with aux0 == NULL both may be NULL (r*h = 0).    */

/* fp32 NCHW -> BF16_C8 (round to nearest even; tail channels zero).  y: N*ceil(C/8)*H*W*8 bfloat16.        */
int ess_to_bf16_c8(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, ess_stream_t stream);
/* BF16_C8 -> fp32 NCHW (exact).                                                                                  */
int ess_from_bf16_c8(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, ess_stream_t stream);
/* Format bridges of the "mixed" configuration (ESS_COMPUTE_F16 convolutions read F16_C8 tensors):
 *   ess_to_f16_c8          fp32 NCHW -> F16_C8 [N][ceil(C/8)][H][W][8] halfs, or (hilo != 0) the [hi | lo] pair of ESS_FMT_F16_C8_HILO,
 *                          [N][2 ceil(C/8)][H][W][8] (event latents handed over as fp32 tensors: models/style_networks.py:69-88);
 *   ess_bf16_c8_to_f16_c8  BF16_C8 -> F16_C8, n_vec 16-byte vectors (exact but for |v| > 65504 / < 6e-8; the image encoder's latents);
 *   ess_f16_c8_to_bf16_c8  F16_C8, or (hilo != 0) a [hi | lo] pair summed, -> BF16_C8 (round to nearest even): the form the weight
 *                          gradient, the skip connection and the L1 terms of the trainable decoder keep reading.                    */
int ess_to_f16_c8(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t hilo, ess_stream_t stream);
int ess_bf16_c8_to_f16_c8(const void* x, void* y, int64_t n_vec, ess_stream_t stream);
int ess_f16_c8_to_bf16_c8(const void* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t hilo, ess_stream_t stream);

/* Weight gradient of the same convolution (replaces cuDNN wgrad under autograd for
 * models/style_networks.py:158-193 and the ResNet prefix :116-121).
 * dy: [N][C_out][H_out][W_out].  dw: [C_out][C0+C1][k][k] (overwritten, or accumulated when
 * accumulate != 0).  db: [C_out] or NULL.  workspace: ess_conv2d_wgrad_workspace() bytes.          */
/* Storage formats (bf16 compute): d->fmt0 (= fmt1) is the sources' format, d->fmt_out that of dy.  Both BF16_C8: 3x3 / stride 1 /
 * pad 1 (direct or nearest-upsampled, one or two sources) and 1x1 / pad 0 (stride 1 or 2) convolutions -- the tiles are
 * transposed to channel-major in registers while they are staged; X BF16_C8 + dy fp32: the 1x1 head (C_in, C_out <= 32);
 * X fp32 + dy BF16_C8: the single-channel 7x7 stem.                                                          */
size_t ess_conv2d_wgrad_workspace(const EssConvDesc* d);
int ess_conv2d_wgrad(const EssConvDesc* d, const void* src0, const void* src1, const void* dy,
                     float* dw, float* db, int accumulate, void* workspace, size_t workspace_bytes,
                     ess_stream_t stream);
/* The weight gradient over n_sets (1..3) tensor sets of the same convolution: dw (+)= sum_s wgrad(src0[s], src1[s], dy[s]).  BF16_C8
 * 3x3 / stride 1 / pad 1: one launch for all sets (one split-K slab set, one reduce); otherwise one accumulating call per set.
 * No reference counterpart (autograd accumulates per backward pass): the decoder's two weight-gradient passes of a UDA step.       */
int ess_conv2d_wgrad_sets(const EssConvDesc* d, int32_t n_sets, const void* const* src0, const void* const* src1,
                          const void* const* dy, float* dw, float* db, int32_t accumulate, void* workspace,
                          size_t workspace_bytes, ess_stream_t stream);

/* InstanceNorm2d(affine=False, eps) (+residual)(+ReLU): models/style_networks.py:163-164,180-182,192.
 * relu = 0: y = IN(x) + residual; 1: y = relu(IN(x)) + residual; 2 (forward only): y = relu(IN(x) +
 * residual) (ResidualBlock with norm='IN', e2vid/model/submodules.py:157-172).
 * stats: [N*C][2] = (mean, rstd) saved for backward.                                                */
size_t ess_norm_workspace(int32_t groups); /* groups = planes (InstanceNorm) or channels (BatchNorm) */
int ess_instnorm_forward(const float* x, const float* residual, float* y, float* stats, int32_t planes,
                         int32_t hw, float eps, int32_t relu, void* workspace, size_t workspace_bytes,
                         ess_stream_t stream);
/* dx from dy (gradient w.r.t. y; the residual branch gradient is dy itself, handled by the caller). */
int ess_instnorm_backward(const float* x, const float* dy, const float* stats, float* dx, int32_t planes,
                          int32_t hw, int32_t relu, void* workspace, size_t workspace_bytes, ess_stream_t stream);

/* ---- the same norms on BF16_C8 tensors (bfloat16 [N][ceil(C/8)][hw][8]; bf16 configuration: the stored form of the trainable
 * networks' activations and activation gradients).  fp32 arithmetic and statistics; stats: fp32 [N*C][2] (InstanceNorm) =
 * (mean, rstd); BatchNorm: fp32 [2][C][2] = C pairs (mean, rstd) followed by C pairs (a, b), the affine map y = x a + b the
 * forward applied (a = rstd gamma, b = beta - mean a): the backward takes its dx scale and -- beta != NULL -- its ReLU mask from
 * that saved map, never from the live gamma / beta.  relu: 0 / 1 as above.  workspace: ess_norm_workspace_c8(N*ceil(C/8)) (InstanceNorm) /
 * ess_norm_workspace_c8(ceil(C/8)) (BatchNorm) bytes.                                                          */
size_t ess_norm_workspace_c8(int32_t groups);
int ess_instnorm_forward_c8(const void* x, const void* residual, void* y, float* stats, int32_t N, int32_t C, int32_t hw,
                            float eps, int32_t relu, int32_t x_f16, void* workspace, size_t workspace_bytes, ess_stream_t stream);
/* InstanceNorm forward of the "mixed" configuration: as ess_instnorm_forward_c8, plus y16 = the result as an F16_C8 tensor (required;
 * what the next ESS_COMPUTE_F16 convolution reads), y (BF16_C8) optional; x_fmt: 0 BF16_C8, 1 F16_C8, 2 the [hi | lo] pair of
 * ESS_FMT_F16_C8_HILO (x = hi + lo); res_f16: 0 the residual is BF16_C8, 1 F16_C8, 2 a [hi | lo] pair (its hi parts are added).  The statistics
 * are those of the values read.  ess_instnorm_backward_c8 takes the same x with x_f16 = x_fmt (2: the hi parts are read).     */
int ess_instnorm_forward_c8_mixed(const void* x, const void* residual, void* y, void* y16, float* stats, int32_t N, int32_t C, int32_t hw,
                                  float eps, int32_t relu, int32_t x_fmt, int32_t res_f16, void* workspace, size_t workspace_bytes,
                                  ess_stream_t stream);
int ess_instnorm_backward_c8(const void* x, const void* dy, const float* stats, void* dx, int32_t N, int32_t C, int32_t hw,
                             int32_t relu, int32_t x_f16, void* workspace, size_t workspace_bytes, ess_stream_t stream);
int ess_batchnorm_train_forward_c8(const void* x, const void* residual, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, void* y, float* stats,
                                   int32_t N, int32_t C, int32_t hw, int32_t relu, int32_t x_f16, void* workspace,
                                   size_t workspace_bytes, ess_stream_t stream);
int ess_batchnorm_train_backward_c8(const void* x, const void* y, const void* dy, const float* gamma, const float* beta,
                                    const float* stats, void* dx, void* d_residual, float* dgamma, float* dbeta, int32_t accumulate,
                                    int32_t N, int32_t C, int32_t hw, int32_t relu, int32_t x_f16, void* workspace,
                                    size_t workspace_bytes, ess_stream_t stream);
/* beta: NULL -- the ReLU mask is read from the saved output y (a forward with a residual).  Non-NULL with relu = 1 -- the forward was
 * relu(bn(x)) without a residual: the mask is recomputed from x as (x a + b <= 0) with the forward's a = rstd gamma, b = beta - mean a,
 * y is not read (may be NULL), d_residual must be NULL: two tensor passes less per launch.                                          */
/* x_f16 = 1: x (the pre-normalisation tensor) is an ESS_FMT_F16_C8 tensor; every other tensor stays BF16_C8.          */

/* BatchNorm2d, training mode (ResNet prefix of StyleEncoderE2VID, models/style_networks.py:116-121):
 * batch statistics, running-stat update with momentum (unbiased var), y = act(bn(x) + residual).
 * stats: [C][2] = (mean, rstd).                                                                     */
int ess_batchnorm_train_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* y,
                                float* stats, int32_t N, int32_t C, int32_t hw, int32_t relu, void* workspace,
                                size_t workspace_bytes, ess_stream_t stream);
/* dy is the gradient w.r.t. y; `y` is needed for the ReLU mask; d_residual (nullable) receives the
 * masked gradient.  dgamma/dbeta: [C] overwritten or accumulated.                                   */
int ess_batchnorm_train_backward(const float* x, const float* y, const float* dy, const float* gamma,
                                 const float* stats, float* dx, float* d_residual, float* dgamma, float* dbeta,
                                 int32_t accumulate, int32_t N, int32_t C, int32_t hw, int32_t relu,
                                 void* workspace, size_t workspace_bytes, ess_stream_t stream);

/* y = bilinear_x2(a + b), align_corners=False (UpsampleConvLayer input: e2vid/model/submodules.py:84,
 * skip_sum e2vid/model/unet.py:12-13,176).  b may be NULL.  a: [planes][H][W] -> y: [planes][2H][2W] */
int ess_upsample_bilinear2x_add(const float* a, const float* b, float* y, int32_t planes, int32_t H,
                                int32_t W, ess_stream_t stream);
/* The same map written as a BF16_C8 tensor [N][ceil(C/8)][2H][2W][8] for a bf16 convolution to stage (the 5x5 decoder convs of
 * the frozen encoder: identical MFMA operands, half the bytes).  a, b: fp32 [N][C][H][W]; W even.                          */
int ess_upsample_bilinear2x_add_c8(const float* a, const float* b, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                   ess_stream_t stream);

/* The same map from BF16_C8 SOURCES a (and b, or NULL), [N][ceil(C/8)][H][W][8] bfloat16: what the frozen encoder's kernels leave
 * as staging copies.  Bit-identical to ess_upsample_bilinear2x_add_c8 applied to the sources' values.
 * Replaces: the same reference lines (e2vid/model/submodules.py:83-93, unet.py:12-13,176). */
int ess_upsample_bilinear2x_add_c8_from_c8(const void* a, const void* b, void* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                           ess_stream_t stream);
/* 2x2 sum pooling = backward of nearest x2 upsampling (F.interpolate, models/style_networks.py:77).
 * accumulate != 0 adds into y.                                                                      */
int ess_sumpool2x2(const float* x, float* y, int32_t planes, int32_t H_out, int32_t W_out, int32_t accumulate,
                   ess_stream_t stream);
/* y = a + b (skip_sum, e2vid/model/unet.py:12-13); in place allowed.                                */
int ess_add(const float* a, const float* b, float* y, int64_t n, ess_stream_t stream);
/* y = a + b (+ c when c != NULL) over bfloat16 tensors of n_vectors x 8 elements (16-byte aligned; any layout, e.g. BF16_C8):
 * fp32 sum, one round to nearest even; in place allowed.  Replaces autograd's gradient accumulation for an activation with
 * several consumers (models/style_networks.py:69-88: out[2] / out[4] feed the next stage and a loss).            */
/* out = sum of n (<= 16) device scalars, added in the order given: the reported total of a train step's weighted loss terms
 * (training/ess_trainer.py:116-138 adds them with one torch op per term).  `terms` is a HOST array of device pointers.  */
int ess_sum_scalars(const float* const* terms, int32_t n, float* out, ess_stream_t stream);
int ess_add_bf16(const void* a, const void* b, const void* c, void* y, int64_t n_vectors, ess_stream_t stream);

/* EventPreprocessor.__call__ normalisation (e2vid/utils/inference_utils.py:96-107) over the whole
 * tensor of n floats: non-zero mean/std, y = (x!=0)*(x-mean)/std; identity copy when all-zero.
 * workspace: >= 32 bytes, zeroed by the call.  No host synchronisation.                             */
int ess_event_normalize(const float* x, float* y, int64_t n, void* workspace, ess_stream_t stream);
/* The same normalisation for all T time slices of a batch in two launches (training/ess_trainer.py:277-280 runs it once per
 * slice data_b[:, t*C:(t+1)*C]): x = [B][T][chunk] with chunk = C*H*W (the event tensor [B, T*C, H, W] as it is: no slice
 * copies), y = [T][B][chunk] (slice t = a contiguous [B, C, H, W] tensor), statistics per slice over the whole batch.
 * workspace: T * 24 bytes.                                                                          */
int ess_event_normalize_slices(const float* x, float* y, int32_t B, int32_t T, int64_t chunk, void* workspace, ess_stream_t stream);

/* ---- events -> voxel grid on the device (the step in front of the encoder) --------------------
 * All slices of a batch in one launch: events concatenated structure-of-arrays, slice s = [slice_offsets[s],
 * slice_offsets[s+1]) (int64, device, n_slices+1 entries); `out` is zeroed by the call.
 *
 * ess_voxel_grid_trilinear replaces VoxelGrid.convert (DSEC/dataset/representations.py:15-55) as driven by
 * Sequence.events_to_voxel_grid (DSEC/dataset/sequence.py:144-154): x, y float pixel coordinates, pol in {0,1},
 * t = any increasing fp32 time; each slice is mapped to [0, channels-1] from its own first/last event.
 * out: [n_slices][channels][height][width].                                                          */
int ess_voxel_grid_trilinear(const float* x, const float* y, const float* pol, const float* t,
                             const int64_t* slice_offsets, int64_t n_events, int32_t n_slices, int32_t channels,
                             int32_t height, int32_t width, float* out, void* workspace, size_t workspace_bytes,
                             int64_t max_events_per_slice, ess_stream_t stream);
/* workspace NULL: every event issues its (up to) 8 atomic adds straight to `out`.  With a workspace of
 * ess_voxel_grid_trilinear_workspace() bytes and max_events_per_slice = the longest slice (host knowledge: it sizes the
 * launch), events are first grouped by 32x32-pixel tile and each tile is accumulated in LDS (about 8x faster).      */
size_t ess_voxel_grid_trilinear_workspace(int64_t n_events, int32_t n_slices, int32_t height, int32_t width);
/* ess_voxel_grid_temporal replaces generate_voxel_grid (datasets/data_util.py:54-126): integer pixels, fp64
 * timestamps, polarity +1 / -1 (0 counts as -1), bilinear in time only.
 * out: [n_slices][separate_pol ? 2*bins : bins][height][width] (positive bins first; else positive - negative). */
int ess_voxel_grid_temporal(const int32_t* x, const int32_t* y, const double* t, const float* pol,
                            const int64_t* slice_offsets, int64_t n_events, int32_t n_slices, int32_t bins,
                            int32_t height, int32_t width, int32_t separate_pol, float* out, ess_stream_t stream);
/* Per-slice normalisation over the non-zero voxels, in place.  mode 0: VoxelGrid(normalize=True)
 * (representations.py:45-53: unbiased std, divide only if std > 0); mode 1: normalize_voxel_grid
 * (data_util.py:38-51: sqrt(E[v^2]-mean^2), no guard).  workspace: ess_voxel_normalize_workspace() bytes.        */
size_t ess_voxel_normalize_workspace(int32_t n_slices);
int ess_voxel_normalize(float* grid, int32_t n_slices, int64_t elems_per_slice, int32_t mode, void* workspace,
                        size_t workspace_bytes, ess_stream_t stream);

/* ---- image-branch augmentation on the device: the geometric + photometric core of the albumentations
 * pipeline of datasets/cityscapes_loader.py:39-74 (HorizontalFlip, ShiftScaleRotate with rotate 0 and a constant-0 border,
 * centred PadIfNeeded, RandomCrop, GaussNoise, RandomBrightnessContrast), uint8 quantisation, ToTensor (/255), and for the label
 * map nearest sampling + the id -> trainId table of utils/labels.py:123-127 -- a whole batch in one launch.
 * img: fp32 [N][H_src][W_src] on the 0..255 scale; label (nullable): int64 [N][H_src][W_src] raw ids; id_lut (nullable):
 * int64[256]; params: fp32 [N][12] = flip, scale, dx, dy (pixels), pad_top, pad_left, crop_y, crop_x, alpha, beta (levels),
 * noise sigma (levels, 0 = off), noise seed -- drawn by the host.  out_img: fp32 [N][1][H][W] in [0,1]; out_label: int64.   */
int ess_augment_image_label(const float* img, const int64_t* label, const float* params, const int64_t* id_lut, float* out_img,
                            int64_t* out_label, int32_t N, int32_t H_src, int32_t W_src, int32_t H, int32_t W,
                            ess_stream_t stream);

/* second stage of the same pipeline (datasets/cityscapes_loader.py:45-57): A.Perspective(p=0.2) -- the homography of a jittered
 * quadrilateral onto a max_w x max_h rectangle (bilinear, constant-0 border; label nearest) followed by the resize back to
 * H x W (keep_size) --, then RandomBrightnessContrast, then A.OneOf([Sharpen, Blur(3), MotionBlur(3)], p=0.5) as one 3 x 3
 * correlation with BORDER_REFLECT_101, rounded to nearest-even like cv2.filter2D; ToTensor.  When this stage follows, stage 1 runs
 * with alpha = 1, beta = 0 and id_lut = NULL (the id -> trainId table is applied here, after the geometry, as the reference does).
 * img: fp32 [N][1][H][W] in [0,1] (stage 1's output, levels / 255); label (nullable): int64 [N][H][W] raw ids;
 * params: fp32 [N][24] = perspective fired, inverse homography (9, row-major: rectangle pixel -> source position), max_w, max_h,
 * alpha, beta (levels), stencil fired, 3 x 3 kernel (9); scratch: fp32 [N][H][W]; out_img: fp32 [N][1][H][W]; out_label int64. */
int ess_augment_perspective_filter(const float* img, const int64_t* label, const float* params, const int64_t* id_lut,
                                   float* scratch, float* out_img, int64_t* out_label, int32_t N, int32_t H, int32_t W,
                                   ess_stream_t stream);

/* TaskLoss = Dice + CrossEntropy (utils/loss_functions.py:6-24,96-135), forward AND gradient w.r.t.
 * logits in one pass pair.  logits [N][K][H][W], labels int64 [N][H][W].  loss: 1 float.
 * dlogits (nullable): d(loss*loss_scale)/dlogits.  workspace: ess_task_loss_workspace(K) bytes.     */
size_t ess_task_loss_workspace(int32_t K);
int ess_task_loss(const float* logits, const int64_t* labels, float* loss, float* dlogits, float loss_scale,
                  int32_t N, int32_t K, int32_t hw, int32_t ignore_index, int32_t use_dice, int32_t use_ce,
                  void* workspace, size_t workspace_bytes, ess_stream_t stream);
/* Workspace of the mean-type losses below (ess_sym_js_loss, ess_l1_loss, ess_l1_loss_c8): one partial sum per workgroup.  Each loss is
 * TWO launches: the kernel proper (every workgroup stores its partial: no atomics, no memset in front) and a one-workgroup finalize
 * that adds the partials in workgroup order -- the value does not depend on scheduling.  Nothing to initialise.  Do not share the
 * buffer between streams.  ABI 110 (round 6): every loss entry point takes workspace_bytes and returns ESS_EINVAL when the buffer is
 * smaller than it needs (ABI 100 callers passed 8 bytes here until round 5 made it one partial per workgroup).                  */
#define ESS_LOSS_WORKSPACE_BYTES (8 * (1 + 2048))
/* symJSDivLoss (utils/loss_functions.py:27-37): loss (1 float) and gradient w.r.t. `a` only
 * (the other argument is always computed under no_grad by the trainers).  workspace: ESS_LOSS_WORKSPACE_BYTES (see above). */
int ess_sym_js_loss(const float* a, const float* b, float* loss, float* da, float loss_scale, int32_t N,
                    int32_t K, int32_t hw, void* workspace, size_t workspace_bytes, ess_stream_t stream);
/* L1Loss mean (training/ess_trainer.py:217-229): loss and gradient w.r.t. a.  workspace: ESS_LOSS_WORKSPACE_BYTES.  */
int ess_l1_loss(const float* a, const float* b, float* loss, float* da, float loss_scale, int64_t n,
                void* workspace, size_t workspace_bytes, ess_stream_t stream);

/* L1Loss mean over BF16_C8 operands (bf16 configuration: the latents / intermediate predictions the cycle losses compare are
 * stored as BF16_C8); da (nullable): BF16_C8 gradient.  n_vectors 16-byte pixel vectors, n REAL elements (the mean's
 * denominator; padded tail channels are zero in both operands).  workspace: ESS_LOSS_WORKSPACE_BYTES.               */
int ess_l1_loss_c8(const void* a, const void* b, float* loss, void* da, float loss_scale, int64_t n_vectors, int64_t n,
                   void* workspace, size_t workspace_bytes, ess_stream_t stream);

/* RAdam.step over ONE flat parameter buffer (utils/radam.py:15-80, weight_decay = 0).
 * step_size / n_sma_ge5 are computed by the host exactly as radam.py:49-64.                         */
int ess_radam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float step_size, int32_t n_sma_ge5,
                   ess_stream_t stream);

/* The same update with the two step-dependent scalars in device memory: hyper[0] = -step_size * lr, hyper[1] != 0 in the
 * rectified phase (N_sma >= 5).  For a train step captured in a hipGraph: the host writes `hyper` before each replay.       */
int ess_radam_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2,
                       float eps, const float* hyper, ess_stream_t stream);

/* F.interpolate(x, size=(H_out, W_out), mode='nearest') on fp32 [planes][H_in][W_in] -- the resize of the validation
 * logits to img_size_b (training/ess_trainer.py:484,525; training/ess_supervised_trainer.py:284).               */
int ess_resize_nearest(const float* x, float* y, int32_t planes, int32_t H_in, int32_t W_in, int32_t H_out,
                       int32_t W_out, ess_stream_t stream);

/* argmax over K + confusion-matrix accumulation (training/ess_trainer.py:485; evaluation/metrics.py:4-24)
 * pred_lbl (nullable): int64 [N][H][W]; conf: int64 [K][K], accumulated (conf[label][pred]).         */
int ess_argmax_confusion(const float* logits, const int64_t* labels, int64_t* pred_lbl, int64_t* conf,
                         int32_t N, int32_t K, int32_t hw, int32_t ignore_index, ess_stream_t stream);

/* confusion-matrix accumulation from GIVEN predictions (evaluation/metrics.py:4-24, semseg_compute_confusion): pred_lbl,
 * labels int64 [total]; conf int64 [K][K] accumulated (conf[label][pred]) over labels != ignore_index.          */
int ess_label_confusion(const int64_t* pred_lbl, const int64_t* labels, int64_t* conf, int64_t total, int32_t K,
                        int32_t ignore_index, ess_stream_t stream);

/* ---- tuning switches: process-wide kernel choices that never change a result (every setting runs the same arithmetic in the
 * same order).  "conv_wide": 0 = the 64 x 256-pixel-tile 3x3 kernel always, 1 = the wide-tile kernel where its round count wins
 * (default; environment ESS_CONV_WIDE), 2 = the wide-tile kernel wherever it applies.  ess_tuning_get also answers the read-only
 * key "device_cus" (the compute-unit count the dispatcher's round model uses: queried from the device, 256 on MI355X).
 * "in_small_threads": 256 | 512 (default; environment ESS_IN_SMALL_THREADS) | 1024 -- threads of the fused single-plane InstanceNorm
 * kernels on planes of at most 5120 pixels; the settings differ in the summation order of the statistics only (fp32 rounding).  */
int ess_tuning_set(const char* key, int32_t value);
int ess_tuning_get(const char* key, int32_t* value);

#ifdef __cplusplus
}
#endif
#endif /* ESS_HIP_H */

/*
 * xumx_slicq_hip.h -- C ABI of the MI355X (gfx950) demix hot path of xumx-sliCQ-V2.
 *
 * The reference (sevagh/xumx-sliCQ, branch v2) is pure Python/PyTorch and has no FFI
 * layer; its boundary for this path is the nn.Module surface of
 *   xumx_slicq_v2/transforms.py  (NSGT_SL.forward :106-131, INSGT_SL.forward :154-178)
 *   xumx_slicq_v2/model.py       (Unmix.forward :69-82, _SlicedUnmixCDAE.forward :213-271)
 *   xumx_slicq_v2/phase.py       (blockwise_wiener :18-69, blockwise_phasemix_sep :96-113)
 *   xumx_slicq_v2/separator.py   (Separator.forward :133-232)
 * This library sits directly under the Python mirrors of those modules
 * (xumx_slicq_amd/ *.py bind it with ctypes; INTEGRATION.md shows the stub a maintainer
 * of the reference would add).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *  - plain C: pointers and sizes only; no torch types.
 *  - every data pointer is DEVICE memory owned by the caller (tensor.data_ptr() of a
 *    contiguous fp32 tensor); the library never frees caller memory.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *    Calls are asynchronous on that stream; nothing synchronises the device.
 *  - return value: 0 on success, negative XSQ_ERR_* otherwise; xsq_last_error() gives
 *    the message of the calling thread's last failure.  Nothing throws across the ABI.
 *  - a plan / model is immutable after creation; workspaces are per call, so one plan
 *    can serve several streams as long as each call has its own workspace.
 *
 * Coefficient arena ("ragged blocks in one allocation")
 *  The sliCQT of `BC` packed channels with `S` slices is ONE fp32 buffer holding the
 *  reference's list of per-block tensors back to back: block b (F_b bins x T_b
 *  coefficients per slice) starts at float offset 2*BC*S*cum_b, cum_b = sum_{b'<b} F_b'*T_b',
 *  and is laid out (BC, F_b, S, T_b, 2) contiguous -- exactly the tensor
 *  NSGT_SL.forward returns for that block, so Python exposes views, not copies.
 *  xsq_plan_block_table() returns (first_band, F_b, T_b, cum_b) per block.
 */
#ifndef XUMX_SLICQ_HIP_H
#define XUMX_SLICQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XSQ_ABI_VERSION 2   /* 2 (round 5): xsq_exchange_rows takes the buffer lengths; *_indirect entry points */

typedef struct xsq_plan xsq_plan;
typedef struct xsq_model xsq_model;

/* ---- library ---------------------------------------------------------------- */
int xsq_abi_version(void);
const char* xsq_last_error(void);
/* How this library was built: "arch=<offload arch>; flags=<the Makefile's EXTRA and NOPK>; date=<__DATE__ __TIME__>" -- a
 * diagnostic build (ablation / stamp macros) says so here, and bench.py copies the string into its JSON line. */
const char* xsq_build_info(void);

/* ---- sliCQT plan ---------------------------------------------------------------
 * Replaces the plan objects built by NSGT_sliced.__init__ (nsgt/slicq.py:70-151):
 * window lengths M / windows g (nsgt/nsgfwin_sl.py:8-111), centre bins and index
 * ranges (nsgt/util.py:72-100), dual windows gd (nsgt/util.py:103-116), slice window
 * (nsgt/slicing.py:7-18).  The host computes those tables (xumx_slicq_amd/plan.py);
 * this call uploads them and builds the per-band DFT matrices.
 *   Lg[nbands], c[nbands]   band lengths and centre bins (bands DC..Nyquist)
 *   g   concatenated analysis windows (sum Lg floats), peak at index 0 of each band
 *   gd  concatenated dual windows (sum Lg doubles), same order
 *   tw  slice (Tukey) window, L floats                                               */
int xsq_plan_create(xsq_plan** out, int L, int tr, int nbands, const int32_t* Lg,
                    const int32_t* c, const float* g, const double* gd, const float* tw);
int xsq_plan_destroy(xsq_plan* plan);
int xsq_plan_num_blocks(const xsq_plan* plan);
/* slice-FFT backend: 0 (default) = the hand-written LDS-resident transform when L == 18060
 * (the Bark-262 plan of both pretrained models), rocFFT otherwise; 1 = always rocFFT.     */
int xsq_plan_set_fft_backend(xsq_plan* plan, int backend);
/* per-band DFTs: 1 (default) = bands with Lg >= 24 (163 of the 263 Bark-262 bands, 99 % of the band-DFT flops; the split
 * point XSQ_D4_MIN_LG_DEFAULT of csrc/slicqt.hip) run on the radix-4 kernel (one decimation-in-
 * frequency stage fused into the operand staging, 4x fewer MFMA flops); 0 = all bands on the
 * dense grouped GEMM.                                                                      */
int xsq_plan_set_band_radix4(xsq_plan* plan, int on);
/* inverse transform, bands below the radix-4 split (Lg < 24: 100 of the 263 Bark-262 bands): 0 (default) = on the dense grouped GEMM;
 * 1 = synthesised inside the inverse slice-FFT kernel straight from the coefficients (radix-4 stage + 4..15-point
 * codelets, no Z round trip, no dense DFT-matrix GEMM) -- an experiment that measured 0.03-0.05 ms SLOWER per 240 s
 * track (the kernel is bound by its chain of memory / LDS round trips) and is kept as an A/B switch.  Ignored
 * when the plan is not eligible (rocFFT backend, other band lengths).                                       */
int xsq_plan_set_short_inline(xsq_plan* plan, int on);
/* hand-written slice FFT only: 1 = its 43 / 14 / 15-point butterflies on packed-fp32 vector instructions (v_pk_fma_f32
 * on the (re, im) register pair: half the vector instructions, bitwise the same results).  DIAGNOSTIC builds only
 * (csrc/Makefile PACKED_FFT=1): the product library does not contain those kernels and returns XSQ_ERR_ARG for on != 0 --
 * a packed-fp32 transform next to v_mfma_f32_16x16x32_bf16 waves of another stream returned wrong values on MI355X
 * (tools/probe/pk_mfma_hazard.hip), which no guard at this level can exclude, and it measured no
 * faster.                                                                                                       */
int xsq_plan_set_packed_fft(xsq_plan* plan, int on);
/* table: nblocks x 4 int64 (first_band, F_b, T_b, cum_b) */
int xsq_plan_block_table(const xsq_plan* plan, int64_t* table);
/* complex coefficients per channel-slice (sum_b F_b*T_b) */
int64_t xsq_plan_coefs_per_slice(const xsq_plan* plan);
/* slices of an n-sample signal (nsgt/slicing.py:47-72): floor((ceil(n/h)+1)/2)+1 */
int xsq_plan_num_slices(const xsq_plan* plan, int64_t n);

/* ---- forward sliCQT -----------------------------------------------------------
 * Replaces NSGT_SL.forward (transforms.py:106-131) -> NSGT_sliced.forward
 * (nsgt/slicq.py:182-196): slicing+Tukey (nsgt/slicing.py:21-72), L-point FFT and
 * per-band window * gather * IFFT (nsgt/nsgtf.py:7-84), arrange (nsgt/slicq.py:13-33).
 *   x     (BC, n) fp32 packed channels        coef  arena for BC channels, S slices  */
size_t xsq_slicqt_forward_workspace(xsq_plan* plan, int BC, int64_t n);   /* 0 on error */
int xsq_slicqt_forward(xsq_plan* plan, const float* x, int BC, int64_t n, float* coef,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same transform, with the CDAE's whitened input written beside the coefficients by the analysis kernels'
 * epilogues: xin[i] = (|coef[i]| + mean[band]) * scale[band] (model.py:238-242; Unmix.forward's abs_of_real_complex,
 * model.py:74-76), real arena layout for BC channels -- normally the head of the xsq_cdae_forward workspace, with
 * mean / scale / split from xsq_model_whitening; xsq_cdae_forward_xin(..., xin_ready = 1) then skips its magnitude
 * pass (one kernel and a second read of the 87 MB mix arena less per chunk).  xin == NULL: plain forward.     */
int xsq_slicqt_forward_xin(xsq_plan* plan, const float* x, int BC, int64_t n, float* coef, float* xin,
                           const float* mean, const float* scale, int split,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The same transform reading packed channel r at x + x_row_offsets[r] (DEVICE array of BC int64 element offsets; NULL =
 * r * n): the chunks of a (nb_samples, 2, N) track that Separator.forward stacks along the batch axis are read where
 * they lie (the reference slices views too, separator.py:153-158) instead of through a packing copy.  n = samples that
 * exist per row, n_pad >= n = the length the slice count is taken from: the zero padding of a short last chunk to
 * sllen/2 + 1 samples (separator.py:162-168) without materialising the zeros (the kernels read zeros outside [0, n)). */
int xsq_slicqt_forward_rows(xsq_plan* plan, const float* x, const int64_t* x_row_offsets, int BC, int64_t n,
                            int64_t n_pad, float* coef, float* xin, const float* mean, const float* scale, int split,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same call with the input's base pointer read from DEVICE memory when the kernel runs: x_slot != NULL -> the rows
 * start at *x_slot + x_row_offsets[r] (x is then ignored and may be NULL).  A HIP graph captured around the call follows
 * whatever tensor the caller points the slot at before each replay (an 8-byte device write) -- the graphed form of the
 * `separator(audio)` call timed at inference.py:28-31 needs no copy of the input into a static buffer.               */
int xsq_slicqt_forward_rows_indirect(xsq_plan* plan, const float* x, const float* const* x_slot, const int64_t* x_row_offsets,
                                     int BC, int64_t n, int64_t n_pad, float* coef, float* xin, const float* mean,
                                     const float* scale, int split, void* workspace, size_t workspace_bytes, void* stream);

/* ---- inverse sliCQT -----------------------------------------------------------
 * Replaces INSGT_SL.forward (transforms.py:154-178) -> NSGT_sliced.backward
 * (nsgt/slicq.py:198-230): per-band FFT, dual-window multiply and overlap-add into the
 * slice spectrum (nsgt/nsigtf.py:5-106), irfft, un-rotate + overlap-add of slices
 * (nsgt/unslicing.py:6-69), crop to `length`.  Does NOT modify `coef` (the reference
 * decoder overwrites its input).
 *   coef  arena for BC channels, S slices     y  (BC, length) fp32                    */
size_t xsq_slicqt_inverse_workspace(xsq_plan* plan, int BC, int S);       /* 0 on error */
int xsq_slicqt_inverse(xsq_plan* plan, const float* coef, int BC, int S, int64_t length,
                       float* y, void* workspace, size_t workspace_bytes, void* stream);
/* Same, but packed channel r is written at y + row_offsets[r] (DEVICE array of BC int64 element
 * offsets) instead of y + r*length: lets the caller place every chunk's stems directly in the
 * final (4, nb_samples, 2, N) tensor -- the torch.cat of separator.py:231 without the copy.   */
int xsq_slicqt_inverse_rows(xsq_plan* plan, const float* coef, int BC, int S, int64_t length,
                            float* y, const int64_t* row_offsets, void* workspace,
                            size_t workspace_bytes, void* stream);
/*   The separator's mix-phase path without the intermediate estimate arena: the coefficients of
 *   packed channel bc are masks[bc] * mix[bc % BCx], formed while the band DFT loads them
 *   (phase.py:96-113 fused into nsigtf.py:85-95).  masks: REAL arena, BC channels (what
 *   xsq_cdae_forward writes with Y = NULL); mix: complex arena, BCx channels.               */
int xsq_slicqt_inverse_masked(xsq_plan* plan, const float* masks, const float* mix, int BC, int BCx,
                              int S, int64_t length, float* y, const int64_t* row_offsets,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- CDAE model ----------------------------------------------------------------
 * Replaces Unmix.forward (model.py:69-82) -> _SlicedUnmixCDAE.forward (model.py:213-271)
 * for all blocks and the four targets, with the mix-phase estimate
 * blockwise_phasemix_sep (phase.py:96-113; == mask * X) fused into
 * the last layer.  The Wiener-EM refinement is xsq_wiener_em below.
 *
 * `params` is a HOST buffer: the fp32 tensors of the reference state_dict in its own
 * key order (per block: input_mean, input_scale; per target: 0.weight, 1.{weight,bias,
 * running_mean,running_var}, 3.weight, 4.{...}, 6.weight, 7.{...}, 9.weight, 9.bias),
 * num_batches_tracked left out.  BatchNorm (eval) is folded at creation.
 * `causal` != 0 selects _CausalConv2d for layer 1 (model.py:274-290).  The block table
 * (F_b bins, T_b coefficients per slice) is the one Unmix receives through its
 * jagged_slicq_sample_input (model.py:29-57); kf follows model.py:112-117.               */
int64_t xsq_model_num_params(int nblocks, const int32_t* F, const int32_t* T);
int xsq_model_create(xsq_model** out, int nblocks, const int32_t* F, const int32_t* T,
                     int causal, const float* params, int64_t nparams);
int xsq_model_destroy(xsq_model* model);
/* Arithmetic of the four convolution layers (the contraction only; epilogues, BatchNorm folding,
 * sigmoid and every other kernel of the path stay fp32).  Inference only.
 *   0  fp32 operands on v_mfma_f32_32x32x2_f32 (default; what the parity fixtures were pinned with)
 *   1  "bf16x3": every fp32 operand carried as hi + lo bf16, three bf16 MFMAs per product, fp32
 *      accumulation -- ~2^-17 relative per product (finer than the TF32 convolutions the
 *      reference's torch-cuda backend runs by default, model.py:130-181 under
 *      torch.backends.cudnn.allow_tf32), 3/16 of the matrix-pipe time.
 *   2  "bf16x6": every fp32 operand cut EXACTLY into three bf16 pieces, the six partial products of
 *      weight >= 2^-16 on bf16 MFMAs, fp32 accumulation; what is dropped is <= 2^-23 |ab| per
 *      product, one fp32 rounding -- measured as close to the torch-cpu reference as mode 0
 *      (1.1e-7 RMS), 6/16 of the matrix-pipe time.                                              */
int xsq_model_set_precision(xsq_model* model, int mode);
/* Fast-convolution forms of the fp32 inference layers, a bit mask (default 7 = bits 1, 2 and 4):
 *   1  layers 2 / 3 (the 4-tap time convolutions of model.py:140-170), rows of >= 127 time positions: Winograd F(2, 4) along
 *      the time taps (csrc/cdae_wino.h: five MFMA products per output pair and channel pair instead of eight; input transform
 *      with integer coefficients in registers, weights transformed on the host in fp64; ~2e-7 RMS of a layer's output against
 *      fp64 where the direct fp32 sum has ~7e-8).  Without the bit: the direct slab kernels (csrc/cdae_slab.h).
 *   2  layer 1 (model.py:130-139, non-causal): F(2, 2) along the hop (csrc/cdae_l1f.h: the strided (kf, W) convolution is a
 *      two-tap convolution in units of the hop -- three half-window products per output pair instead of four, W0 + W1 summed
 *      on the host in fp64).  Without the bit: the implicit GEMM (CdaeL1Op).
 *   4  layer 4 (model.py:171-181, non-causal): F(2, 2) along the hop the same way (csrc/cdae_l4f.h), masks only or with the
 *      estimates materialised.  Without the bit: the implicit GEMM (CdaeL4Op).
 *   8  (with bit 1) layers 2 / 3, rows of >= 253 time positions: Winograd F(4, 4) (csrc/cdae_wino4.h: seven products per
 *      output quad instead of ten).  An A/B arm, off by default: parity-green and measured 8-10 % slower than F(2, 4) on
 *      MI355X.  Its weights are built only into models created with XSQ_WINO4=1 in the environment; the bit is an error on
 *      any other model.
 * The split-bf16 modes, the causal first layer and (bits 1, 8) shorter rows always take the direct kernels.              */
int xsq_model_set_winograd(xsq_model* model, int on);
size_t xsq_cdae_workspace(const xsq_model* model, int B, int S);          /* 0 on error */
/*   X      mix coefficients, arena for 2*B channels (B, 2, ...)
 *   Y      out: mask * X, arena for 8*B channels laid out (4 targets, B, 2, ...); NULL (with masks
 *          given) writes the masks only -- the input xsq_slicqt_inverse_masked expects
 *   masks  out, optional (NULL to skip): sigmoid masks, REAL arena with the geometry of Y
 *          (one float per coefficient) -- Unmix.forward(return_masks=True)             */
int xsq_cdae_forward(xsq_model* model, const float* X, int B, int S, float* Y, float* masks,
                     void* workspace, size_t workspace_bytes, void* stream);
/* xin_ready != 0: the head of `workspace` (2 B S sum(F T) floats) already holds the whitened magnitude, written by
 * xsq_slicqt_forward_xin with this model's tables (xsq_model_whitening: device pointers to input_mean / input_scale
 * per band and the operand format of the current precision mode).                                            */
int xsq_cdae_forward_xin(xsq_model* model, const float* X, int B, int S, float* Y, float* masks,
                         void* workspace, size_t workspace_bytes, void* stream, int xin_ready);
int xsq_model_whitening(xsq_model* model, const float** mean, const float** scale, int* split);

/* ---- post-filters on the arena ---------------------------------------------------
 * Both take an explicit block table (nblocks, F[], T[]) so that they also serve the
 * reference's single-block helpers (phase.py) on shapes outside the sliCQT plan.
 *
 * xsq_phasemix replaces blockwise_phasemix_sep (phase.py:96-113, with _atan2 :72-93):
 *   Y[j,b,c] = mag[j,b,c] * X[b,c] / |X[b,c]|   (angle(0) := 0); does not modify X
 *   (the reference's _atan2 adds 1 to re(X) where X == 0).
 *   X  complex arena, 2*B channels;  mag  REAL arena, 8*B channels (4, B, 2, ...);
 *   Y  complex arena, 8*B channels.
 *
 * xsq_wiener_em replaces blockwise_wiener (phase.py:18-69) after the initial estimate:
 * norbert.wiener(v, x, iterations=1, use_softmask=False) (norbert/__init__.py:153-260)
 * per window of <= win_len frames of the flattened (slice, time) axis.
 *   X  complex arena, 2*B channels
 *   Y  in: initial estimates y0 = v * exp(i angle x) (what xsq_cdae_forward / xsq_phasemix
 *      write); out: refined estimates, in place.                                        */
int xsq_phasemix(int nblocks, const int32_t* F, const int32_t* T, const float* X,
                 const float* mag, float* Y, int B, int S, void* stream);
size_t xsq_wiener_workspace(int nblocks, const int32_t* F, const int32_t* T, int B, int S,
                            int win_len);                                   /* 0 on error */
/*   batch_group: the window maximum max(1, 0.1*max|x|) of norbert :257 spans the batch dimension
 *   (SURVEY.md quirk A13).  <= 0 or B: one group, the reference's behaviour for one call.  g < B
 *   (g divides B): every run of g consecutive batch items is its own group -- lets a caller stack
 *   independent chunks along the batch axis without changing any chunk's result.          */
int xsq_wiener_em(int nblocks, const int32_t* F, const int32_t* T, const float* X, float* Y,
                  int B, int S, int win_len, int batch_group, void* workspace,
                  size_t workspace_bytes, void* stream);

/* ---- placement of packed stems (sharded path) -------------------------------------------------
 * Replaces the hard torch.cat of separator.py:229-231 for stems that were computed on OTHER ranks and arrive packed
 * in an all-gather buffer (xumx_slicq_amd/sharding.py): row i of the table is copied as
 *   dst[table[3i+1] : + table[3i+2]] = src[table[3i] : + table[3i+2]]      (float offsets / lengths)
 * in ONE launch for all rows (item x target x sample x channel) of an exchange.  table: DEVICE int64[nrows][3];
 * max_len = the largest row length (sizes the grid).  Rows must not overlap in dst.                     */
int xsq_place_rows(const float* src, float* dst, const int64_t* table, int nrows, int64_t max_len, void* stream);

/* ---- in-place exchange of stems between the ranks of the sharded path (RCCL over xGMI) -------------------------------
 * The form of "the final waveform concat" (separator.py:229-231 across ranks) that needs no packing buffer and no
 * placement pass: every rank holds the SAME flat per-track layout, its kernels write the rows it owns in place
 * (xsq_demix_pass through out_rows), and ONE grouped ncclSend / ncclRecv per pass moves every row owner -> peers at
 * identical offsets -- point to point, one xGMI link per peer.
 *   xsq_comm_load       resolve RCCL from the library the process already uses (path of torch's librccl.so; NULL: by name)
 *   xsq_comm_unique_id  rank 0: 128 bytes to hand to every rank (torch.distributed broadcast)
 *   xsq_comm_create     ncclCommInitRank on the CURRENT device -- collective over the ranks
 *   xsq_exchange_rows   rows: HOST int64[nrows][4] = (owner rank, src float offset, dst float offset, length), the same
 *                       table on every rank; the owner sends src + src_off to every peer, every other rank receives into
 *                       dst + dst_off (src == dst and equal offsets: in place).  src_len / dst_len: floats behind src / dst --
 *                       every row's spans are checked against them before anything is queued.  Grouped ncclSend / ncclRecv
 *                       on `stream`, asynchronous; a group is closed every XSQ_EXCHANGE_GROUP_ROWS rows (default ~1024
 *                       point-to-point operations), at the same rows on every rank.  self_loop != 0 (tests, group of
 *                       one): the owner also sends to ITSELF and receives into dst -- src and dst must then not overlap. */
typedef struct xsq_comm xsq_comm;
int xsq_comm_load(const char* librccl_path);
int xsq_comm_version(void);                                                /* ncclGetVersion code, -1 on error */
int xsq_comm_unique_id(void* id128);
int xsq_comm_create(xsq_comm** out, const void* id128, int world, int rank);
int xsq_comm_destroy(xsq_comm* comm);
int xsq_comm_abort(xsq_comm* comm);                                        /* ncclCommAbort: do not wait for queued operations; frees the handle */
int xsq_exchange_rows(xsq_comm* comm, const float* src, int64_t src_len, float* dst, int64_t dst_len, const int64_t* rows,
                      int nrows, int self_loop, void* stream);

/* The same EM iteration fed by the MASKS (real arena, 8*B channels: what xsq_cdae_forward writes with Y = NULL): the
 * initial estimate y0 = mask * x (model.py:262-264 -> phase.py:96-113; == mask * X) is formed
 * while both passes load, so the CDAE's last layer stores 4 instead of 8 bytes per coefficient and the statistics pass
 * reads 48 instead of 80 bytes per time-frequency point.  Y is written only (refined estimates); same bits as
 * xsq_cdae_forward(Y) + xsq_wiener_em.  win_len and every S*T_b must be even (two frames per thread).       */
int xsq_wiener_em_masked(int nblocks, const int32_t* F, const int32_t* T, const float* X, const float* masks,
                         float* Y, int B, int S, int win_len, int batch_group, void* workspace,
                         size_t workspace_bytes, void* stream);

/* A batch too large for one pass (xsq_separator_forward splits it over the samples): the window maximum of norbert :257
 * spans the WHOLE batch, so every pass first folds the maxima of its samples into a shared table, and the EM of each pass
 * then takes max(own, table).  ext_max: DEVICE float[xsq_wiener_num_windows(...)], one entry per (block, group, window)
 * in block-major order, zeroed by the caller before the first xsq_wiener_window_max; holds max |x|^2.  Passes of one set
 * must agree on B / batch_group = number of groups.  ext_max == NULL: xsq_wiener_em_masked.                            */
int64_t xsq_wiener_num_windows(int nblocks, const int32_t* F, const int32_t* T, int B, int S, int win_len, int batch_group);
int xsq_wiener_window_max(int nblocks, const int32_t* F, const int32_t* T, const float* X, int B, int S, int win_len,
                          int batch_group, float* ext_max, void* stream);
int xsq_wiener_em_masked_ext(int nblocks, const int32_t* F, const int32_t* T, const float* X, const float* masks,
                             float* Y, int B, int S, int win_len, int batch_group, const float* ext_max,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- loss forward (validation half of training.loop, training.py:34-112 with train=False) -------
 * Replaces ComplexMSELossCriterion (loss.py:37-76) and MaskSumLossCriterion (loss.py:79-96).
 *   pred, target  complex arenas, 8*B channels (4 targets, B, 2, ...)
 *   masks         real arena, 8*B channels, or NULL
 *   out           DEVICE double[2*nblocks]: per block (complex-MSE mean over the 14 combinations,
 *                 mask-sum mean); the criteria average these over the blocks.                  */
size_t xsq_loss_workspace(int nblocks, const int32_t* F, const int32_t* T, int B, int S);
int xsq_loss_forward(int nblocks, const int32_t* F, const int32_t* T, const float* pred,
                     const float* target, const float* masks, int B, int S, double* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- dataset statistics (training.get_statistics, training.py:115-154) ---------------------------
 * Per block and frequency bin: sum and sum of squares over the S*T_b frames of the channel-mean
 * magnitude of X (C packed channels of ONE track).  out: DEVICE double[2 * sum_b F_b] (sum, sumsq per
 * bin, blocks back to back); the host merges tracks and forms mean / std like sklearn's
 * StandardScaler.partial_fit.  workspace >= 32 * sum_b F_b bytes.                              */
int xsq_magnitude_stats(int nblocks, const int32_t* F, const int32_t* T, const float* X, int C, int S,
                        double* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- training step (training.loop with train=True, training.py:34-112; BASELINE config 5) -------
 * One handle owns the parameters, their gradients and the AdamW moments as flat fp32 pools in the
 * xsq_model_create order (the reference's state_dict order), plus the GEMM-layout weights the
 * forward needs (re-gathered on the device every step).
 *   xsq_train_step: forward with BatchNorm on BATCH statistics (nn.BatchNorm2d.train(), running
 *   statistics updated with momentum 0.1 when apply_update != 0), loss = ComplexMSE + MaskSum
 *   (loss.py:37-96, SDR term off as in the published training), backward of every trainable tensor
 *   (what loss.backward() at training.py:107 produces), AdamW update (training.py:391-393).
 *     X        mix arena (2B channels, complex)      Yt   target arena (8B channels, complex)
 *     wiener   0: mix-phase (realtime model)         1: differentiable Wiener-EM (offline model)
 *     apply_update 0: gradients only (parameters and running statistics untouched)
 *     loss_out HOST double[2]: complex-MSE term, mask-sum term (the step waits for them,
 *              as loss.item() at training.py:110 does); NULL: do not wait, see xsq_train_loss
 *   xsq_train_read: what = 0 parameters, 1 gradients, 2 / 3 AdamW first / second moments -> HOST
 *   float[nparams].  xsq_train_write restores parameters (0) or a moment pool (2 / 3) from the host and
 *   xsq_train_step_count reads (set < 0) or restores the AdamW step counter: together they are the
 *   optimizer.state_dict() the reference checkpoints for resuming (training.py:419-430).        */
typedef struct xsq_train xsq_train;
int xsq_train_create(xsq_train** out, int nblocks, const int32_t* F, const int32_t* T, int causal,
                     const float* params, int64_t nparams);
int xsq_train_destroy(xsq_train* t);
size_t xsq_train_workspace(const xsq_train* t, int B, int S, int wiener);      /* 0 on error */
int xsq_train_step(xsq_train* t, const float* X, const float* Yt, int B, int S, int wiener,
                   float lr, float weight_decay, int apply_update, double* loss_out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* loss_out == NULL: the step does not wait.  The loss terms of the last 4 steps stay retrievable by ticket
 * (xsq_train_ticket: ticket of the step issued last, 0-based count of steps on this handle); xsq_train_loss blocks
 * until that step has finished and returns its two terms -- a loop can issue step k + 1 before it looks at the loss
 * of step k, which keeps the device busy across the read-back that loss.item() makes every step.            */
int64_t xsq_train_ticket(xsq_train* t);
int xsq_train_loss(xsq_train* t, int64_t ticket, double* loss_out);
int xsq_train_read(xsq_train* t, int what, float* host_out);
int xsq_train_write(xsq_train* t, int what, const float* host_in);
int64_t xsq_train_step_count(xsq_train* t, int64_t set);
/* Arithmetic of the GEMM-shaped forward / data-gradient kernels of the step: 0 fp32 MFMA (default), 2 bf16x6 (see
 * xsq_model_set_precision), 1 = "bf16": BASELINE configs[4] as written -- the reference runs forward + loss under
 * torch.autocast(dtype=bfloat16) (training.py:66-108,473-476), i.e. its convolutions contract bf16-rounded operands.  Mode 1
 * rounds both operands of every forward, data-gradient and weight-gradient contraction to bf16 (round to nearest even) and
 * accumulates one v_mfma_f32_32x32x16_bf16 product per pair in fp32; activations between the layers, BatchNorm, the
 * Wiener-EM, the loss, the master weights and AdamW stay fp32 -- fewer rounding points than autocast, which also keeps
 * BatchNorm / ReLU / sigmoid outputs in bf16 (tests/golden/training_step_bf16.npz holds the reference's own
 * fp32-vs-autocast spread; the arm is held inside it).  XSQ_TRAIN_WGRAD_FP32=1 keeps the weight gradients on fp32
 * operands (an A/B arm).                                                                                             */
int xsq_train_set_precision(xsq_train* t, int mode);

/* ---- the whole call: Separator.forward (separator.py:133-232) behind ONE entry point ----------------------------
 * The reference's chunk loop -- per chunk of <= chunk_size samples: zero-pad to sllen/2 + 1 (162-168), sliCQT (170),
 * Unmix with mix-phase or Wiener-EM (172-219), isliCQT to the chunk's length (221-227), hard concat (229-231) -- issued
 * from native code so that the host enqueues a 240 s track in ~0.1 ms instead of walking ~40 Python / ctypes calls and
 * ~300 tensor views per step (which left the eager step host-bound on slow hosts).
 *
 * xsq_demixer: per-plan cache of the row-offset tables of a call shape (device arrays, built on first use) and the
 *   fork / join events of the tail stream.  Not owned: the plan must outlive it.  The model is passed per call (its
 *   handle is rebuilt when parameters change).
 * xsq_demix_pass: ONE pass over B = items * group equal-length work items (chunks of any tracks; `group` = nb_samples
 *   of an item, the scope of the Wiener window maximum):
 *     xsq_slicqt_forward_rows (+ whitened magnitude) -> xsq_cdae_forward_xin (masks only)
 *       -> wiener == 0: xsq_slicqt_inverse_masked          (mix-phase, the estimate mask * X formed on the way in)
 *          wiener == 1: xsq_wiener_em_masked -> xsq_slicqt_inverse_rows
 *   x_rows: DEVICE int64[2B] element offsets of the input rows (NULL: contiguous (B, 2, n)); out_rows: DEVICE
 *   int64[8B] element offsets of packed channel (target, item, c) in `out`; n = samples per item, n_pad as above.
 * xsq_separator_forward: audio (nb, 2, N) -> out (4, nb, 2, N), both DEVICE fp32 contiguous.  Full chunks are stacked
 *   along the batch axis, at most max_stack (chunk, sample) pairs per pass and never more rows than one launch
 *   addresses (a batch too large for one pass is split over the samples; with Wiener-EM the passes of such a set first
 *   fold their window maxima into a shared table, xsq_wiener_window_max, so the maximum keeps its batch-wide scope,
 *   norbert/__init__.py:257); the remaining chunks run one by one, on `tail_stream` beside the
 *   stacked passes when overlap_tail != 0 and tail_stream != stream (forked / joined with events: capturable in a
 *   HIP graph).  workspace / tail_workspace: xsq_separator_workspace bytes each (the tail one may be NULL when nothing
 *   runs on the tail stream).  Same bits as the chunk-by-chunk loop.  A demixer serves ONE call at a time (its fork / join
 *   events and the workspaces are per call); concurrent callers use one demixer each.                              */
typedef struct xsq_demixer xsq_demixer;
int xsq_demixer_create(xsq_demixer** out, xsq_plan* plan);
int xsq_demixer_destroy(xsq_demixer* d);
size_t xsq_demix_pass_workspace(xsq_demixer* d, const xsq_model* model, int B, int64_t n_pad, int wiener);   /* 0 on error */
int xsq_demix_pass(xsq_demixer* d, xsq_model* model, const float* x, const int64_t* x_rows, int B, int64_t n,
                   int64_t n_pad, int group, int wiener, float* out, const int64_t* out_rows,
                   void* workspace, size_t workspace_bytes, void* stream);
int xsq_separator_workspace(xsq_demixer* d, const xsq_model* model, int nb, int64_t N, int64_t chunk_size, int max_stack,
                            int wiener, size_t* main_bytes, size_t* tail_bytes);
int xsq_separator_forward(xsq_demixer* d, xsq_model* model, const float* audio, int nb, int64_t N, int64_t chunk_size,
                          int max_stack, int wiener, int overlap_tail, float* out, void* workspace,
                          size_t workspace_bytes, void* tail_workspace, size_t tail_workspace_bytes, void* stream,
                          void* tail_stream);
/* xsq_separator_forward with the audio's base pointer in a DEVICE slot (audio_slot: device address of one `const float*`),
 * read by the first kernel of every pass when it RUNS: the call can be captured once in a HIP graph and replayed on any
 * (nb, 2, N) tensor of the captured shape after an 8-byte write to the slot (Separator.forward_graphed).              */
int xsq_separator_forward_indirect(xsq_demixer* d, xsq_model* model, const float* const* audio_slot, int nb, int64_t N,
                                   int64_t chunk_size, int max_stack, int wiener, int overlap_tail, float* out,
                                   void* workspace, size_t workspace_bytes, void* tail_workspace,
                                   size_t tail_workspace_bytes, void* stream, void* tail_stream);
/* The schedule xsq_separator_forward follows for a call shape, as pure host arithmetic (no device needed): pass i ->
 * passes[8 i ..] = (first sample of its first chunk, samples per chunk, chunks stacked, first sample index of the batch,
 * samples of the batch in this pass, 1 = may run on the tail stream, index of the first pass of its set, 1 = the set
 * shares a Wiener-EM window-maximum table).  L = slice length, coefs_per_slice = sum F_b T_b; max_item_slices <= 0: the
 * library's own cap.  Returns the number of passes (may exceed max_passes: call again), negative on error.            */
int xsq_separator_schedule(int L, int64_t coefs_per_slice, int nb_samples, int64_t N, int64_t chunk_size, int max_stack,
                           int wiener, int max_item_slices, int64_t* passes, int max_passes);
/* Test hook: cap on B * S (stacked items x slices) of one pass, normally what the 32-bit arena offsets of the band
 * kernels allow (7168); a smaller value forces the batch split on small shapes.  <= 0 restores the default.        */
int xsq_demixer_set_max_rows(xsq_demixer* d, int max_item_slices);

/* ---- per-kernel timing (bench.py roofline) ------------------------------------------
 * When enabled, every kernel launch of the library is bracketed by hipEvents recorded on
 * its own launch stream.  xsq_profile_read synchronises the outstanding events and
 * returns accumulated milliseconds / launch counts per kernel name (newline-separated
 * names, same order as ms[] / launches[]); returns the number of entries.              */
int xsq_profile_enable(int on);
int xsq_profile_reset(void);
/* Time only the kernel launched under this event name (NULL or "": all kernels).  Two event records around every
 * launch cost ~0.1 ms per 24-launch step; bench.py times every kernel in its warm-up steps and only the dominant
 * one inside the timed region.                                                                              */
int xsq_profile_filter(const char* name);
int xsq_profile_read(char* names, size_t names_bytes, double* ms, int64_t* launches,
                     int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* XUMX_SLICQ_HIP_H */

/* flowtimes.h — C ABI of libflowtimes_hip.so (MI355X / gfx950).
 *
 * The reference (ShinDongWoon/Flow-TimesNet) is pure Python/PyTorch and has no
 * FFI of its own (SURVEY.md finding 1); this ABI is therefore build-defined and
 * each entry point cites the reference code it replaces
 * (paths relative to src/timesnet_forecast/models/timesnet.py).
 *
 * Conventions
 *  - every pointer marked "dev" is a device pointer owned by the caller
 *    (PyTorch tensors); nothing is allocated or freed behind the caller's back;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*)
 *    and never synchronises, so a sequence of calls can be captured in a hipGraph;
 *  - return value: 0 = ok, <0 = bad argument (see ftn_last_error), >0 = hipError_t;
 *  - all tensors are fp32, contiguous, C (channel) fastest: x[B][L][C].
 */
#ifndef FLOWTIMES_H
#define FLOWTIMES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTN_ABI_VERSION 10
#define FTN_KMAX 16      /* max period candidates / groups per block call        */
#define FTN_MAXBR 8      /* max kernels in kernel_set                             */

/* Device-side period descriptor: what PeriodGrouper.group() returns on the host
 * in the reference (PeriodGroupResult, :275-283) plus the tiling the conv kernels
 * use.  Written by ftn_period_finalize (device) or built on the host by
 * ftn_desc_from_periods and copied up; read by every ftn_timesblock_* kernel, so
 * the host never has to synchronise to learn data-dependent shapes. */
typedef struct FtnDesc {
  int32_t n_sel;                    /* K': periods kept by the selector (:153)           */
  int32_t n_groups;                 /* G: distinct valid periods (:551)                  */
  int32_t total_px;                 /* sum_g (L + pad_g): grid pixels per batch row      */
  int32_t tiles_per_row;            /* sum_g ntx*nty: conv tiles per batch row           */
  int32_t sel_freq[FTN_KMAX];       /* rFFT bin of each kept candidate (:156)            */
  int32_t sel_period[FTN_KMAX];     /* period of each kept candidate, score order (:157) */
  int32_t sel_group[FTN_KMAX];      /* mapping[K] -> group id or -1 (:460-477)           */
  int32_t g_period[FTN_KMAX];       /* group periods, ascending (:453-458)               */
  int32_t g_pad[FTN_KMAX];          /* (-L) mod p (:531)                                 */
  int32_t g_cycles[FTN_KMAX];       /* (L+pad)/p (:533)                                  */
  int32_t g_px_off[FTN_KMAX + 1];   /* prefix sums of (L+pad_g)                          */
  int32_t g_tw[FTN_KMAX];           /* conv tile width  (phase axis)                     */
  int32_t g_th[FTN_KMAX];           /* conv tile height (cycle axis)                     */
  int32_t g_ntx[FTN_KMAX];
  int32_t g_nty[FTN_KMAX];
  int32_t g_tile_off[FTN_KMAX + 1]; /* prefix sums of ntx*nty                            */
} FtnDesc;

/* Host-side description of one TimesBlock's (folded, packed) inception weights.
 * Filled by ftn_inception_pack_weights (host); offsets are in floats
 * into the weight blob.  Replaces nn.Sequential(InceptionBlock, act,
 * InceptionBlock) (:744-762) for inference. */
typedef struct FtnPlan {
  int32_t C, CP;          /* d_model, padded to 16                                          */
  int32_t F, FP;          /* d_ff, padded to 16                                             */
  int32_t mode;           /* 0: bottleneck branches (:581-590); 1: single conv (:575-580)   */
  int32_t act;            /* 0: GELU(erf) (:643), 1: ReLU (:641)                            */
  int32_t nbr;            /* conv branches: mode 0 = len(kernel_set); mode 1 = 1 (merged)   */
  int32_t MP;             /* mode 0: mid channels per branch padded to 16                   */
  int32_t kh[FTN_MAXBR], kw[FTN_MAXBR];
  int32_t res1, res2;     /* 1 = res_proj conv present (:634-635), 0 = identity (:637)      */
  /* block 1 (d_model -> d_ff) */
  int64_t w_in1, b_in1;   /* mode 0: [nbr*MP][CP] 1x1 in->mid, bias                         */
  int64_t w_conv1[FTN_MAXBR]; /* [taps][cin/16][cout/16][16][16]                            */
  int64_t b_conv1;        /* mode 0: [nbr*MP]; mode 1: [FP] (proj folded)                   */
  int64_t w_out1, b_out1; /* mode 0: folded proj.branch[-1]: [FP][nbr*MP], [FP]             */
  int64_t w_res1, b_res1; /* [FP][CP], [FP]                                                 */
  /* block 2 (d_ff -> d_model) */
  int64_t w_in2, b_in2;   /* mode 0: [nbr*MP][FP]                                           */
  int64_t w_conv2[FTN_MAXBR];
  int64_t b_conv2;        /* mode 0: [nbr*MP]; mode 1: [CP]                                 */
  int64_t w_out2, b_out2; /* mode 0: [CP][nbr*MP], [CP]                                     */
  int64_t w_res2, b_res2; /* [CP][FP], [CP]                                                 */
  /* stage-C output projection = rows [w_in2 ; w_res2] stacked: [(nbr*MP + CP)][FP]        */
  int64_t w_c2, b_c2;
  /* the same stage-C matrices as lane-linear MFMA fragments, grouped per 32-channel
   * hidden chunk: [chunk][ w_out1: 2 tiles x KM/16 | w_res1: 2 x CP/16 | w_c2: 2 x n_ot ][lane][4]
   * (rows/cols beyond FP are zero fragments) */
  int64_t w_cfrag;
  int32_t cfrag_per_chunk;  /* fragments (of 256 floats) per hidden chunk */
  int32_t n_hchunks;        /* ceil(FP / 32)                               */
  /* bf16x3 conv engine (mode 0 only): per branch the k x k weights as three bf16 pieces in
   * K=32 MFMA fragments, [cin/16][cout/16][slab = tap pair][piece][lane][8] (offsets in
   * floats; the data are bf16).  engine: 0 = exact fp32 MFMA, 1 = bf16x3 split, 2 = plain bf16,
   * 3 = f16x2 split (below) */
  int64_t w_convbf1[FTN_MAXBR];
  int64_t w_convbf2[FTN_MAXBR];
  int32_t engine;
  /* bf16x3 stage-C fragments per 32-channel hidden chunk (3 pieces x 1 KiB each; only when both
   * res_proj convs exist): [chunk][w_out1 2 x ceil(KM/32) | w_res1 2 x ceil(CP/32) | w_c n_ot][piece] */
  int32_t cfragbf_per_chunk;
  int64_t w_cfragbf;
  /* f16x2 engine (engine 3; mode 0 only): w_convbf1/2 and w_cfragbf then hold THREE fp16 pieces per weight
   * fragment - A1 = fp16(W~), A2 = fp16(A1 2^-11), A3 = fp16(W~ - A1) of the prescaled matrix W~ = sc W with
   * sc a power of two (per conv branch, per stage-C matrix) - and activations travel as two fp16 pieces
   * (hi, (x - hi) 2^11).  The kernels start their accumulators from biases prescaled by the same sc (offsets
   * below) and multiply by 1/sc when they leave the matrix pipe. */
  int64_t b_conv1s, b_conv2s, b_out1s, b_res1s, b_c2s;
  float sc_conv1[FTN_MAXBR], sc_conv2[FTN_MAXBR];
  float sc_out1, sc_res1, sc_a2, sc_r2;   /* W_out1, W_res1, W_in2 rows of w_c2, W_res2 rows of w_c2 */
  /* split engines, stage E on the 16-bit matrix pipe (the shapes that carry w_cfragbf): w_out2 as K=32 fragments of
   * three pieces, [CP/16 row tiles][ceil(nbr*MP/32) slabs][piece][lane][8] (0 = absent); f16x2: of sc_out2 w_out2,
   * with b_out2s = sc_out2 b_out2 */
  int64_t w_out2fb, b_out2s;
  float sc_out2;
  int32_t reserved0;
  int64_t total_floats;
} FtnPlan;

int ftn_abi_version(void);
const char* ftn_last_error(void);

/* ---- weight folding + packing (host only): replaces nn.Sequential(InceptionBlock, act, InceptionBlock) ---- */
/* The reference state_dict tensors of ONE InceptionBlock (:622-637) as raw host pointers (fp32, contiguous):
 * paths.{j}.branch.{i}.{weight,bias} - bottleneck branches (:581-590): i = 0 [mid][in][1][1], 1 [mid][mid][kh][kw],
 * 2 [out][mid][1][1]; single-conv branches (ratio 1, :575-580): only i = 0, [out][in][kh][kw] - proj.{weight,bias}
 * [out][out*nk][1][1], res_proj.{weight,bias} [out][in][1][1] or NULL when in == out (nn.Identity, :637). */
typedef struct FtnInceptionBlockWeights {
  const float* branch_w[FTN_MAXBR][3];
  const float* branch_b[FTN_MAXBR][3];
  const float* proj_w;
  const float* proj_b;
  const float* res_w;
  const float* res_b;
} FtnInceptionBlockWeights;
/* floats of the packed blob for this shape (0 = bad argument); engine as FtnPlan.engine */
size_t ftn_inception_pack_floats(int d_model, int d_ff, int n_kernels, const int* kh, const int* kw,
                                 double bottleneck_ratio, int engine);
/* Folds proj into every branch (fp64), pads channels to 16, lays the matrices out as MFMA fragments and, for the
 * split engines, splits them into pieces; writes the blob (host memory, to be copied to the device 16-byte
 * aligned) and fills *plan_out.  block0 = inception[0] (d_model -> d_ff), block2 = inception[2] (d_ff -> d_model),
 * act 0 = GELU, 1 = ReLU.  flow-timesnet_amd/pack.py is a thin wrapper of this call. */
int ftn_inception_pack_weights(const FtnInceptionBlockWeights* block0, const FtnInceptionBlockWeights* block2,
                               int d_model, int d_ff, int n_kernels, const int* kh, const int* kw,
                               double bottleneck_ratio, int act, int engine, float* blob_host, size_t blob_floats,
                               FtnPlan* plan_out);

/* ---- multi-GPU exchange of the [F] partial batch sums ------------------------------
 * A batch-sharded TimesBlock needs one exchange per call: every rank's fp64 column sums of its channel-median
 * spectrum, summed in rank order on every rank, so that all ranks select the same periods (:112 is a mean over the
 * whole batch).  Instead of a collective launch each rank's k_colsum STORES its sums straight into a slot of every
 * peer's exchange buffer - device memory of the peer, mapped here once with hipIpcOpenMemHandle, one hop over xGMI -
 * followed by a sequence word; the finalize workgroup of ftn_period_finalize[_stage_a] waits (bounded) for the
 * sequence words of all ranks in its OWN buffer and sums the slots in rank order.  No host work per call beyond
 * passing this struct; no collective; deterministic.
 *   slots[r]  rank r's exchange buffer as mapped into THIS process (slots[rank] = this rank's own hipMalloc'ed
 *             buffer of ftn_exchange_bytes(world, F_cap) bytes, zeroed once before the first call)
 *   seq       this call's sequence number: identical on every rank, starts at 1, +1 per exchange (the two halves of
 *             the buffer alternate by seq & 1, so a rank one call ahead never overwrites what a peer still reads)
 * A rank that does not hear from a peer within ~2 s writes an empty descriptor (the block becomes the identity) and
 * sets the buffer's error word (ftn_exchange_error).  Not capturable in a HIP graph (seq is a launch argument). */
#define FTN_XCHG_MAXWORLD 16
typedef struct FtnExchange {
  void* slots[FTN_XCHG_MAXWORLD];
  int32_t world, rank, F_cap;
  uint64_t seq;
} FtnExchange;
size_t ftn_exchange_bytes(int world, int F_cap);
/* this rank's buffer on the current device (hipMalloc + zero) and its 64-byte hipIpcMemHandle_t, to be sent to the
 * other ranks by whatever channel the host has; the other ranks' handles are opened with ftn_exchange_open */
int ftn_exchange_alloc(int world, int F_cap, void** buf_out, void* handle64_out);
int ftn_exchange_open(const void* handle64, void** mapped_out);
int ftn_exchange_close(void* mapped);
int ftn_exchange_free(void* buf);
/* host-side read of the error word of this rank's own buffer (synchronises the stream): 0 = ok, 1 = a peer timed out */
int ftn_exchange_error(const FtnExchange* xch, void* stream);

/* ---- period selector: FFTPeriodSelector.forward (:64-159) ------------------- */
/* bytes of the DFT twiddle table for window length L */
size_t ftn_dft_table_bytes(int L);
/* fill the table (cos/sin of 2*pi*f*t/L evaluated in fp64 on the device) */
int ftn_dft_table_init(void* table_dev, int L, void* stream);
/* S1+S2 (:108-112): med[b][f] = lower-median_c |rfft_t x[b,:,c]|_f, f < L/2+1,
 * psum[f] = sum_b med[b][f] (fp64, fixed order).  med: [B][F] dev, psum: [F] dev.
 * Two kernels, bit-identical results: one workgroup per (row, 32-bin block), or - when a row's folded samples
 * and amplitude tile fit LDS (C <= 64, e.g. L = 336) and B >= 64 - one workgroup per row with x[b] resident in
 * LDS.  FTN_SEL_ROW=1 / 0 in the environment forces / forbids the second form. */
/* xch (ABI 9; may be NULL): also publish psum to slot `rank` of every rank's exchange buffer (see FtnExchange). */
/* scratch_dev (ABI 10; may be NULL): ftn_period_spectrum_scratch_bytes(B, L, C) bytes of device memory.  With it,
 * 64 < C <= 128 (d_model 128) runs the quarter-folded DFT as (row, 32-channel tile) workgroups that park their
 * amplitudes [B][F][C] there, and a second launch takes the channel medians; without it (or when the function
 * returns 0) those shapes run the (row, 32-bin block) kernel. */
size_t ftn_period_spectrum_scratch_bytes(int B, int L, int C);
int ftn_period_spectrum(const float* x_dev, int B, int L, int C, const void* table_dev,
                        float* med_dev, double* psum_dev, void* stream, const FtnExchange* xch, void* scratch_dev);
/* S3-S5 (:119-157, PeriodGrouper.group :513-557, softmax/scatter :992-1009).
 * psum: [nparts][F] partial batch sums (summed in index order; nparts>1 is the
 * multi-GPU exchange of SURVEY §8e), Btotal = global batch.  Writes the
 * descriptor, amps[B][FTN_KMAX] and group weights w[B][FTN_KMAX] (both 16-byte aligned).
 * act_dtype: dtype of the caller's activations - 0 fp32, 1 bf16, 2 fp16.  For half inputs the reference
 * rounds the batch-mean spectrum and the scores (:124, :130), the returned amplitudes (:159), the softmax
 * weights (:1000) and their scatter-added group sums (:1009) to that dtype; the kernel applies the same
 * roundings (the values are still delivered as fp32).
 * max_unique / log_base: the reference's TIMES_PERIOD_MAX_UNIQ / TIMES_PERIOD_BINNING grouping variants
 * (:350-437) with the per-depth schedule already resolved by the caller; 0 / 0.0 = unset (plain
 * duplicate-merge grouping).  log_base is a double (ABI 9): the bucket is floor(log(p) / log(base) + 1e-6) with the
 * fp32 log of the period divided by (float)log(base), base in double - torch's operand order (:352). */
int ftn_period_finalize(const double* psum_dev, int nparts, int Btotal, const float* med_dev,
                        int B, int L, int k_periods, int pmax, int min_period_threshold, int act_dtype,
                        int max_unique, double log_base, FtnDesc* desc_dev, float* amps_dev, float* weights_dev,
                        void* stream, const FtnExchange* xch);
/* (xch != NULL: psum_dev / nparts are ignored - the partial sums are the world slots of this rank's exchange buffer
 *  for sequence number xch->seq, which the kernel waits for) */
/* Host-only: PeriodGrouper.group (:513-557, env flags unset) + conv tiling for
 * periods that come from somewhere else (stub selectors in the reference tests).
 * `periods` is a host array; `desc_host` is filled on the host. */
int ftn_desc_from_periods(const int64_t* periods, int K, int L, int min_period, int max_period,
                          FtnDesc* desc_host);

/* Host-only: upper bounds for descriptors that ftn_period_finalize can write for this selector
 * configuration: returns the bound on total_px (grid pixels per batch row, > 0; < 0 on a bad argument)
 * and stores the bound on n_groups.  The selector only emits periods clamp(ceil(L/i), lo, hi) for rFFT
 * bins i (:144-145), whose pads are tiny except for bin 1, so this is far below the generic worst case. */
int ftn_selector_px_bound(int L, int k_periods, int pmax, int min_period_threshold, int* max_groups_out);

/* ---- TimesBlock conv path: _period_conv_bucketed_slicing (:955-1101) --------- */
/* Workspace / grids are sized from bounds on the device-side descriptor: max_groups >= desc->n_groups and
 * px_bound >= desc->total_px (ftn_selector_px_bound, or the exact total_px of a host-built descriptor;
 * 0 = the worst case over any max_groups distinct periods, almost 2*L*max_groups).  A descriptor that
 * exceeds them makes the call the identity y = x; nothing is ever written past the workspace.  The
 * workspace belongs to ONE call in flight: calls that may overlap (different streams, or a captured
 * graph beside eager calls) need their own.  Returns 0 for a shape it cannot run. */
size_t ftn_timesblock_workspace_bytes(const FtnPlan* plan, int B, int L, int max_groups, int px_bound);
/* y = x + sum_g w[b,g] * (inception(fold_g(x)) - fold_g(x))[:L]  (:1041-1092, :818).
 * desc/weights are device pointers.  act_dtype (0 fp32, 1 bf16, 2 fp16) = dtype of the caller's activations:
 * x_dev / y_dev are always fp32 buffers (the caller up-casts, as the reference does for its convs, :1047-1052),
 * and for a half dtype every per-group delta, each weighted term, their sum and x + sum are rounded to it
 * exactly where the reference rounds them (:1068-1069, :1092, :818), so y holds values of that dtype.
 * flags: FTN_FWD_STAGE_A_DONE = ftn_period_finalize_stage_a already ran stage A into this workspace. */
#define FTN_FWD_STAGE_A_DONE 1
/* range_flag (ABI 9; may be NULL): one int32 the kernels can write, in device memory or in pinned host memory the
 * device can reach.  Engine f16x2 carries activations as fp16 pieces (|value| < 65504); the reference computes in
 * fp32 (:1047-1056).  Wherever a value is split - x, the stage outputs a, m, a' - and at the final y (NaN / inf) the
 * kernels test it and store 1 to *range_flag when it does not fit; the caller zeroes the word, and on 1 repeats the
 * call with a plan of engine bf16x3 (full fp32 exponent range).  Other engines never touch it. */
int ftn_timesblock_forward(const float* x_dev, float* y_dev, int B, int L, const FtnPlan* plan,
                           const float* wblob_dev, const FtnDesc* desc_dev, const float* weights_dev,
                           int max_groups, int px_bound, int act_dtype, int flags, void* ws_dev, size_t ws_bytes,
                           void* stream, int* range_flag);
/* ftn_period_finalize and stage A of the block (a = W_in1 x + b per window position, which does not depend on the
 * selector) in ONE launch: workgroup 0 is the finalize kernel and then publishes the sanitised descriptor copy at
 * the head of the workspace, the other workgroups compute stage A - the selector's single-workgroup tail (~17 us)
 * no longer leaves the chip idle.  Same plan / workspace / bounds as the ftn_timesblock_forward call that
 * follows with FTN_FWD_STAGE_A_DONE.  Bottleneck-mode plans only (mode 0).
 * The two halves can also be launched apart, on the same workspace - a batch-sharded run puts stage A between
 * issuing the exchange of the partial sums and waiting for it:  psum_dev == NULL runs stage A only (med / desc /
 * amps / weights unused),  x_dev == NULL runs S3-S5 and the descriptor copy only. */
int ftn_period_finalize_stage_a(const double* psum_dev, int nparts, int Btotal, const float* med_dev, int B, int L,
                                int k_periods, int pmax, int min_period_threshold, int act_dtype, int max_unique,
                                double log_base, FtnDesc* desc_dev, float* amps_dev, float* weights_dev,
                                const float* x_dev, const FtnPlan* plan, const float* wblob_dev, int max_groups,
                                int px_bound, void* ws_dev, size_t ws_bytes, void* stream, int* range_flag,
                                const FtnExchange* xch);
/* The same call followed by the caller's per-block epilogue of TimesNet.forward (:2050-2058, eval mode):
 *   y = LayerNorm_C( x + (block(x) - x) ; gamma, beta, eps )
 * fused into the last kernel when d_model <= 64 (bottleneck mode), one extra in-place row pass otherwise. */
int ftn_timesblock_forward_norm(const float* x_dev, float* y_dev, int B, int L, const FtnPlan* plan,
                                const float* wblob_dev, const FtnDesc* desc_dev, const float* weights_dev,
                                int max_groups, int px_bound, int flags, const float* ln_gamma_dev,
                                const float* ln_beta_dev, float ln_eps, void* ws_dev, size_t ws_bytes, void* stream,
                                int* range_flag);
/* out[row][:] = LayerNorm_C( x[row][:] + (new[row][:] - x[row][:]) ) for rows x C fp32 matrices (in place
 * allowed: out == new).  Used when a block returns x unchanged (no valid period, :796-797). */
int ftn_residual_layernorm(const float* x_dev, const float* new_dev, float* out_dev, long long rows, int C,
                           const float* ln_gamma_dev, const float* ln_beta_dev, float ln_eps, void* stream);

/* ---- LowRankTemporalContext (:1340-1371) -------------------------------------- */
/* basis buffer: (L+1)*R floats = basis[l][r] (DCT-II columns r=1..R, centred over l,
 * unit L2 norm, :1344-1351) followed by the R residual column means */
size_t ftn_lrtc_basis_floats(int L, int R);
int ftn_lrtc_basis(float* basis_dev, int L, int R, void* stream);
/* out[b][l][n] = (x ? x[b][l][n] : 0) + scale * sum_r basis_c[l][r] coeff[b][n][r]
 * with the time-mean removed (:1368-1371).  scale_dev: 1 float on the device. */
int ftn_lrtc_forward(const float* coeff_dev, const float* basis_dev, const float* scale_dev,
                     const float* x_dev_or_null, float* out_dev, int B, int L, int N, int R,
                     void* stream);

/* ---- model shell around the block stack (TimesNet.forward) ------------------------ */
/* Rate / dispersion heads (:2066-2102), one pass over hidden[rows = B*S][D] (the output of
 * forecast_time_proj, time-major):
 *   pre        = hidden W_mu^T + b_mu + tail[b, min(s, hist-1), :] (+ late[b, s, :])
 *   rate       = softplus(pre) + 1e-6
 *   dispersion = softplus(hidden W_sigma^T + b_sigma) + floor + 1e-6
 * W_* are nn.Linear weights [N][D]; tail points at x[b=0, T-hist, 0] with batch stride tail_bstride
 * (elements); late (optional) is gate * late_bias laid out [.., S, N] with batch stride late_bstride
 * (0 = shared by the batch); floor is min_sigma_vector[N] or the scalar.  *bad_flag_dev gets bit 0 / 1
 * OR-ed in when a rate / dispersion is not finite and > 0 (the reference raises RuntimeError, :2095-2098);
 * the caller zeroes it.  D must be a multiple of 4, <= 128. */
int ftn_head_forward(const float* hidden_dev, long long rows, int S, int D, int N, const float* w_mu_dev,
                     const float* b_mu_dev, const float* w_sigma_dev, const float* b_sigma_dev,
                     const float* tail_dev, long long tail_bstride, int hist, const float* late_dev_or_null,
                     long long late_bstride, const float* floor_vec_dev_or_null, float floor_scalar,
                     float* rate_dev, float* disp_dev, int* bad_flag_dev, void* stream);

/* Value embedding of DataEmbedding.forward (:1283-1325) with everything row-linear folded into `add`:
 *   out[b][l][:] = x[b][l][:] W^T + add[b?][l][:]      (+ LayerNorm over D when gamma/beta are given)
 * x: the [B, L, N] input window (rows contiguous, batch stride x_bstride elements); W: nn.Linear weight
 * [D][N]; add: optional [L][D] (add_bstride 0) or [B][L][D] holding bias + positional/time-feature term
 * (+ the low-rank temporal context and constant context bias pushed through W, :1958-1996, so the
 * [B, L, N] context tensor is never written).  D must be a multiple of 4, <= 128. */
int ftn_embed_forward(const float* x_dev, long long x_bstride, int B, int L, int N, const float* w_dev, int D,
                      const float* add_dev_or_null, long long add_bstride, const float* ln_gamma_dev_or_null,
                      const float* ln_beta_dev_or_null, float ln_eps, float* out_dev, void* stream);

/* ---- measurement ---------------------------------------------------------------- */
/* hipEvent brackets around the 6 stages (A pw-in, B conv, C fused pointwise chain,
 * D conv, E pw-out, F combine) of the following ftn_timesblock_forward calls - every
 * `enable`-th call (1 = every call, 0 = off; the events themselves cost ~3 us each on the
 * stream), up to 512 recorded calls; nothing synchronises until ftn_stage_times is called.
 * The first enabling call creates the events (host time): do it outside a timed region. */
int ftn_stage_timing(int enable);
/* sums, over the calls recorded since ftn_stage_timing(1), of each stage's elapsed
 * milliseconds; synchronises on the recorded events.  nstage must be 6. */
int ftn_stage_times(float* ms_sum_host, int nstage, int* ncalls_host);

/* Diagnostic: register (or clear with NULL) a device buffer of n_u64 64-bit words; thread 0 of
 * every following k_conv / k_mlp workgroup stores s_memtime at its phase boundaries in
 * words [8*wg .. 8*wg+7].  The last registered conv/mlp launch wins; not for production. */
int ftn_debug_stamps(void* buf_dev, size_t n_u64, int which /* 1: k_conv, 2: k_mlp */);

/* ---- diagnostics --------------------------------------------------------------- */
/* writes D = A(16x8, a[i][k]=i*8+k+1) * B(8x16, b[k][j]=(k+1)*100+j) via two
 * v_mfma_f32_16x16x4_f32 to out[16][16]: verifies the lane maps the kernels assume */
int ftn_selftest_mfma(float* out_dev, void* stream);
/* Diagnostic: out[i] = GELU(in[i]) exactly as the kernels evaluate nn.GELU() (erf form, :643). */
int ftn_selftest_gelu(const float* in_dev, float* out_dev, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWTIMES_H */

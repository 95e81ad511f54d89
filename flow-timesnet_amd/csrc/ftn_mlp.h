// Shared between inception.hip and stagec_pos.hip: the split-engine stage-C argument block, the piece-product
// chain, the in-register piece split and the guarded x loads (gfx950 only).
#pragma once
#include "ftn_common.h"

template <bool XVEC>
__device__ __forceinline__ f4 load_x4(const float* __restrict__ xrow, int c, int C) {
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (xrow == nullptr) return v;
  if (XVEC) {
    if (c < C) v = *(const f4*)(xrow + c);
  } else {
    if (c + 0 < C) v.x = xrow[c + 0];
    if (c + 1 < C) v.y = xrow[c + 1];
    if (c + 2 < C) v.z = xrow[c + 2];
    if (c + 3 < C) v.w = xrow[c + 3];
  }
  return v;
}


struct MlpBfArgs {
  const float* x;
  const __bf16* m;       // P3 [N][KM/16][3][16]
  const __bf16* cfrag;   // [n_hchunks][per_chunk][3][512]
  const float* bo;
  const float* br;
  const float* bc;
  __bf16* outA;          // P3 [N][AC/16][3][16]
  float* outR;           // [N][CP]
  const FtnDesc* desc;
  int B, L, C, CP, FP, KM, AC;
  int nsKM, nsCP;        // K=32 slabs of layer 1 / of the residual (each <= 2)
  int n_oa, n_ot, n_hchunks, per_chunk;
  // f16x2 engine (NS == 2): the accumulators carry the power-of-two prescale of their weight matrix (bo / br /
  // bc then point at biases prescaled the same way): z = acc_o * inv_o;  acc_r starts as act(z) * sc_r + br~
  // and g = act(acc_r * inv_r);  a' = acc * inv_a (tiles < n_oa),  r = acc * inv_r2 - x.  All 1 otherwise.
  float inv_o, sc_r, inv_r, inv_a, inv_r2;
  int r_keeps_x;         // 1: outR = res2(g) + b (x NOT subtracted: k_out takes it out once, OutArgs.r_keeps_x)
  int* range_flag;       // f16x2 engine: set when x or a' leaves the fp16 range (ftn_common.h); may be null
  unsigned long long* dbg; size_t dbg_cap;
};

template <int NS>
__device__ __forceinline__ f4 chain_bf(const bf8 (&ap)[PxFmt<NS>::NW], const bf8 (&bp)[NS], f4 c) {
  if constexpr (NS == 3) {
    c = mfma_bf(ap[0], bp[2], c);
    c = mfma_bf(ap[2], bp[0], c);
    c = mfma_bf(ap[1], bp[1], c);
    c = mfma_bf(ap[0], bp[1], c);
    c = mfma_bf(ap[1], bp[0], c);
    return mfma_bf(ap[0], bp[0], c);
  } else if constexpr (NS == 2) {      // f16x2: A2 lo' + A3 hi + A1 hi (ftn_common.h), small terms first
    c = mfma_h(ap[1], bp[1], c);
    c = mfma_h(ap[2], bp[0], c);
    return mfma_h(ap[0], bp[0], c);
  } else {
    return mfma_bf(ap[0], bp[0], c);
  }
}

// eight fp32 values -> NS pieces of 8 (bf16: exact truncation split; fp16: hi + scaled remainder, ftn_common.h)
template <int NS>
__device__ __forceinline__ void split_pieces(const float (&v)[8], bf8 (&out)[NS]) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  if constexpr (NS == 2) {
    unsigned pc[2][4];
    split_h2<8>(v, pc);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const u4 w = {pc[p][0], pc[p][1], pc[p][2], pc[p][3]};
      out[p] = __builtin_bit_cast(bf8, w);
    }
  } else {
    unsigned pc[NS][4];
    split_trunc<NS, 8>(v, pc);
#pragma unroll
    for (int p = 0; p < NS; ++p) {
      const u4 w = {pc[p][0], pc[p][1], pc[p][2], pc[p][3]};
      out[p] = __builtin_bit_cast(bf8, w);
    }
  }
}


// ---------------------------------------------------------------- stage C, position-major (split engines, fp32 activations)
// For t < L grid pixel t of EVERY period group is window position (b, t), and two of stage C's four
// matrix products do not depend on the group at all:
//   res1(x) = W_res1 x + b           depends on (b, t) only                                   (:645-647)
//   sum_g w[b,g] res2(g_g) = W_res2 (sum_g w[b,g] g_g) + b sum_g w[b,g]      by linearity     (:1075-1092 over :651-654)
// So a wave owns 16 window positions and walks the period groups INSIDE the hidden-chunk loop: per 32-channel chunk
// res1 once, then per group  h = W_out1 m_g + b;  g_g = act(act(h) + res1);  a'_g += W_in2 g_g;  s += w_g g_g,  and
// W_res2 s once.  155 648 instead of 286 720 multiply-adds per window position at the bench shape (five groups), one
// weight refill / barrier pair per chunk for five groups' worth of work, x read once, and the per-pixel residual
// tensor R_g (written here, re-read by k_out: ~220 MB per step) shrinks to one [B*L][CP] tensor that already holds
// the weighted group sum.  Groups are processed GB at a time (registers: 28 per group at d_model 64); more than GB
// groups run as several batches that recompute res1 and add into the same R accumulators.
// The tail pixels t >= L of a grid (live zero inputs that feed the second conv's halo, never the output) have no
// window position: the blocks past n_main walk them as one-group units with x = 0 and no R.
struct MlpPosArgs {
  MlpBfArgs c;
  const float* wts;      // [B][FTN_KMAX] softmax group weights w[b,g] (finalize kernel)
  float* outRs;          // [B*L][CP]: sum_g w[b,g] (res2(g_g) + b_res2)   (x NOT subtracted: OutArgs.r_summed)
  int n_main, n_tail;    // blocks [0, n_main) own window positions, [n_main, n_main + n_tail) the tail pixels
  int abl;               // timing ablations (FTN_MLP_POS_ABL; results wrong): 1 = no weight refill / chunk barriers after chunk 0
};

// position-major stage C of the d_model-64 shape (stagec_pos.hip); act 0 GELU / 1 ReLU, nsplit = activation pieces
int ftn_launch_mlp_pos64(const MlpPosArgs& pa, int act, int nsplit, bool xvec, int tail_units_bound, hipStream_t st);
// the same for the d_model-128 shape (f16x2 only)
int ftn_launch_mlp_pos128(const MlpPosArgs& pa, int act, bool xvec, int tail_units_bound, hipStream_t st);
int ftn_mlp_pos_enabled();

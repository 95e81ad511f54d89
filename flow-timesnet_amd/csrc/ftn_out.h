// Stage E + F argument block and the pieces both of its kernels use (k_out in inception.hip, k_out_h in
// stage_out.hip); gfx950 only.
#pragma once
#include "ftn_common.h"

struct OutArgs {
  const float* x;
  float* y;
  const float* m;        // [N][KM] conv output of block 2
  const float* R;        // [N][CP]
  const float* W;        // w_out2 [CP][KM] row-major (IDENT: unused, KM == CP)
  const float* bias;
  const float* wts;      // [B][FTN_KMAX]
  const FtnDesc* desc;
  int B, L, C, CP, KM;
  const float* ln_g;     // optional fused epilogue (FAST path): y = LayerNorm_C(x + ((x + comb) - x)), the
  const float* ln_b;     // per-block residual + shared LayerNorm of TimesNet.forward (reference :2050-2058)
  float ln_eps;
  int r_keeps_x;         // R holds res2(g) + b, not res2(g) + b - x: y = x + (sum_g w_g (e_g + R_g) - (sum_g w_g) x).  Stage C then
                         // never re-reads x (93 MB of its 286 MB of fetches at the bench shape); fp32 activations only -
                         // for half inputs every per-group delta is rounded, so x must come off before the weighting
  int act_dtype;         // 1 bf16 / 2 fp16 input: the reference rounds every per-group delta, each weighted
                         // term, their sum and x + sum to the input dtype (:1068-1069, :1092, :818); 0 = fp32
  int* range_flag;       // f16x2 engine: set when an output value is not finite (ftn_common.h); may be null
  // stage E on the 16-bit matrix pipe (k_out_h, stage_out.hip): m' as activation pieces, w_out2 as K=32 fragments
  const __bf16* mh;      // [N][KM/16] piece records (H2 / P3), written by the second conv
  const __bf16* Wf;      // FtnPlan.w_out2fb
  float inv_out2;        // f16x2: 1 / sc_out2 (bias then points at b_out2s); 1 otherwise
  int r_summed;          // position-major stage C (k_mlp_pos): R is ONE [B*L][CP] tensor that already holds
                         // sum_g w_g (res2(g_g) + b) per window position: y = x + (sum_g w_g e_g + R - (sum_g w_g) x)
};

__device__ __forceinline__ f4 rnd_act4(f4 v, int act_dtype) {
  if (act_dtype == 1) { v.x = (float)(__bf16)v.x; v.y = (float)(__bf16)v.y; v.z = (float)(__bf16)v.z; v.w = (float)(__bf16)v.w; }
  else if (act_dtype == 2) { v.x = (float)(_Float16)v.x; v.y = (float)(_Float16)v.y; v.z = (float)(_Float16)v.z; v.w = (float)(_Float16)v.w; }
  return v;
}

// LayerNorm over the channel axis of an MFMA D-layout tile set: v[o][u][r] is channel 16o+4q+r of the
// pixel (u, lane&15); the four q lane-groups of a pixel are combined with two xor-shuffles.
// Two-pass (mean, then centred sum of squares), biased variance, as nn.LayerNorm.
template <int NO, int NPX>
__device__ __forceinline__ void ln_tiles(f4 (&v)[NO][NPX], int n_ot, int C, int q, const float* __restrict__ g,
                                         const float* __restrict__ b, float eps) {
  const float inv = 1.0f / (float)C;
#pragma unroll
  for (int u = 0; u < NPX; ++u) {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (o < n_ot && 16 * o + 4 * q + r < C) s += v[o][u][r];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * inv;
    float ss = 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (o < n_ot && 16 * o + 4 * q + r < C) {
          const float dv = v[o][u][r] - mean;
          ss += dv * dv;
        }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float rstd = 1.0f / sqrtf(ss * inv + eps);
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (o < n_ot && 16 * o + 4 * q + r < C) {
          const int ch = 16 * o + 4 * q + r;
          v[o][u][r] = (v[o][u][r] - mean) * rstd * g[ch] + b[ch];
        }
  }
}


// stage E + F with W_out2 m' on the 16-bit matrix pipe (stage_out.hip); act 0 GELU / 1 ReLU, nsplit 2 (f16x2) / 3 (bf16x3)
int ftn_launch_out_h(const OutArgs& oa, int act, int nsplit, bool xvec, hipStream_t st);
int ftn_out_h_enabled();

// Model-shell kernels either side of the TimesBlock stack:
//   k_head   rate / dispersion heads of TimesNet.forward (reference models/timesnet.py:2066-2102)
//   k_embed_in  value embedding GEMM with the low-rank temporal context folded in (:1958-1996, :1283-1325)
// HBM-bound: each reads its input once and writes its outputs once; the small GEMMs run on the exact
// fp32 MFMA (v_mfma_f32_16x16x4_f32) so no precision is traded.
#include "ftn_common.h"

// softplus(beta=1, threshold=20) as torch evaluates it (x > 20 -> x, else log1p(exp(x))), in the
// overflow-free form max(x,0) + log1p(exp(-|x|)).  exp2/log2 are the raw hardware ops (1 ulp);
// below 2^-6 the series replaces log(1+e), whose argument rounding would cost relative accuracy.
__device__ __forceinline__ float softplus20(float x) {
  if (x > 20.f) return x;
  const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
  const float series = e * (1.f - e * (0.5f - e * (0.33333333333f - 0.25f * e)));
  const float lg = __builtin_amdgcn_logf(1.f + e) * 0.69314718055994531f;
  return fmaxf(x, 0.f) + (e < 0.015625f ? series : lg);
}

struct HeadArgs {
  const float* hidden;   // [rows][D]
  const float* wmu;      // [N][D]
  const float* bmu;      // [N]
  const float* wsg;
  const float* bsg;
  const float* tail;     // element (b, s, n) at tail[b * tail_bs + min(s, hist-1) * N + n]
  const float* late;     // optional, element (b, s, n) at late[b * late_bs + s * N + n]
  const float* floorv;   // optional [N]
  float* rate;           // [rows][N]
  float* disp;
  int* bad;              // |= 1 if a rate is not finite-positive, |= 2 for a dispersion
  long long rows, tail_bs, late_bs;
  int S, D, N, hist;
  float floor_s;
};

// One workgroup owns a 64-wide slice of the N output series: the two weight slices live in LDS in
// MFMA A-fragment order (row i of tile o is series n0 + 16*(i>>2) + 4*o + (i&3), so a lane ends up with 16
// consecutive series of one (b, s) row = one 64-byte run, four lanes = 256 bytes).  Its four waves walk
// 16-row tiles of `hidden`; per tile 2 x 4 x D/4 MFMAs, then the fused epilogue.
template <int NS, bool VEC>
__global__ __launch_bounds__(256) void k_head(HeadArgs a) {
  extern __shared__ f4 wl[];   // [2][4][NS][64]
  const int n0 = blockIdx.x * 64;
  for (int f = threadIdx.x; f < 2 * 4 * NS * 64; f += 256) {
    const int ln = f & 63;
    int t = f >> 6;
    const int S = t % NS;
    t /= NS;
    const int o = t & 3, h = t >> 2, i = ln & 15, q = ln >> 4;
    const int n = n0 + 16 * (i >> 2) + 4 * o + (i & 3), k = 16 * S + 4 * q;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < a.N) {
      const float* __restrict__ W = (h ? a.wsg : a.wmu) + (size_t)n * a.D + k;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < a.D) v[e] = W[e];
    }
    wl[f] = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int nl0 = n0 + 16 * q;
  f4 bm[4], bs[4], fl[4];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nl0 + 4 * o + r;
      const bool in = n < a.N;
      bm[o][r] = in ? a.bmu[n] : 0.f;
      bs[o][r] = in ? a.bsg[n] : 0.f;
      fl[o][r] = (in && a.floorv) ? a.floorv[n] : a.floor_s;
    }
  int badbits = 0;
  for (long long t = (long long)blockIdx.y * 4 + wave; t * 16 < a.rows; t += (long long)gridDim.y * 4) {
    const long long row = t * 16 + j;
    const bool rok = row < a.rows;
    const long long rr = rok ? row : a.rows - 1;
    const long long b = rr / a.S;
    const int s = (int)(rr - b * a.S);
    f4 hf[NS];
#pragma unroll
    for (int S = 0; S < NS; ++S) {
      const int k = 16 * S + 4 * q;
      hf[S] = k < a.D ? *(const f4*)(a.hidden + rr * a.D + k) : f4{0.f, 0.f, 0.f, 0.f};
    }
    const float* __restrict__ tp = a.tail + b * a.tail_bs + (size_t)(s < a.hist ? s : a.hist - 1) * a.N + nl0;
    const float* __restrict__ lp = a.late ? a.late + b * a.late_bs + (size_t)s * a.N + nl0 : nullptr;
    f4 ex[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f4 tv = {0.f, 0.f, 0.f, 0.f};
      if (VEC) {
        if (nl0 + 4 * o < a.N) tv = *(const f4*)(tp + 4 * o);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nl0 + 4 * o + r < a.N) tv[r] = tp[4 * o + r];
      }
      ex[o] = tv;
    }
    f4 am[4], as[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) am[o] = as[o] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < NS; ++S) {
      f4 wm[4], ws[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        wm[o] = wl[((0 * 4 + o) * NS + S) * 64 + lane];
        ws[o] = wl[((1 * 4 + o) * NS + S) * 64 + lane];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          am[o] = mfma16(wm[o][e], hf[S][e], am[o]);
          as[o] = mfma16(ws[o][e], hf[S][e], as[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f4 pre = am[o] + bm[o] + ex[o];                       // mu_head(hidden) + history tail (:2079)
      if (lp) {
        f4 lv = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
          if (nl0 + 4 * o < a.N) lv = *(const f4*)(lp + 4 * o);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nl0 + 4 * o + r < a.N) lv[r] = lp[4 * o + r];
        }
        pre = pre + lv;                                      // + gate * late_bias (:2080-2090)
      }
      f4 rt, dp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rt[r] = softplus20(pre[r]) + 1e-6f;                  // :2091
        dp[r] = softplus20(as[o][r] + bs[o][r]) + fl[o][r] + 1e-6f;   // :2092-2094
        if (rok && nl0 + 4 * o + r < a.N) {
          if (!(rt[r] > 0.f && rt[r] <= 3.4028234664e38f)) badbits |= 1;
          if (!(dp[r] > 0.f && dp[r] <= 3.4028234664e38f)) badbits |= 2;
        }
      }
      if (!rok) continue;
      float* __restrict__ rp = a.rate + rr * a.N + nl0 + 4 * o;
      float* __restrict__ dq = a.disp + rr * a.N + nl0 + 4 * o;
      if (VEC) {
        if (nl0 + 4 * o < a.N) {
          __builtin_nontemporal_store(rt, (f4*)rp);
          __builtin_nontemporal_store(dp, (f4*)dq);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nl0 + 4 * o + r < a.N) { rp[r] = rt[r]; dq[r] = dp[r]; }
      }
    }
  }
  if (badbits) atomicOr(a.bad, badbits);
}

template <int NS>
static int launch_head(const HeadArgs& a, bool vec, hipStream_t st) {
  const int ntile = (a.N + 63) / 64;
  const long long rtiles = (a.rows + 15) / 16;
  long long gy = (rtiles + 3) / 4;
  const long long want = (4 * 256 + ntile - 1) / ntile;       // ~4 workgroups per CU in flight
  if (gy > want) gy = want;
  const size_t lds = (size_t)2 * 4 * NS * 64 * sizeof(f4);
  if (vec) hipLaunchKernelGGL((k_head<NS, true>), dim3(ntile, (unsigned)gy), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((k_head<NS, false>), dim3(ntile, (unsigned)gy), dim3(256), lds, st, a);
  FTN_CHECK_LAUNCH();
  return 0;
}

extern "C" int ftn_head_forward(const float* hidden_dev, long long rows, int S, int D, int N, const float* w_mu_dev,
                                const float* b_mu_dev, const float* w_sigma_dev, const float* b_sigma_dev,
                                const float* tail_dev, long long tail_bstride, int hist,
                                const float* late_dev_or_null, long long late_bstride,
                                const float* floor_vec_dev_or_null, float floor_scalar, float* rate_dev,
                                float* disp_dev, int* bad_flag_dev, void* stream) {
  FTN_CHECK_ARG(hidden_dev && w_mu_dev && b_mu_dev && w_sigma_dev && b_sigma_dev && tail_dev && rate_dev &&
                    disp_dev && bad_flag_dev,
                "ftn_head_forward: null pointer");
  FTN_CHECK_ARG(rows >= 1 && S >= 1 && rows % S == 0 && N >= 1 && hist >= 1 && hist <= S,
                "ftn_head_forward: rows=%lld S=%d N=%d hist=%d", rows, S, N, hist);
  FTN_CHECK_ARG(D >= 4 && D % 4 == 0 && D <= 128, "ftn_head_forward: d_model=%d must be a multiple of 4, <= 128", D);
  FTN_CHECK_ARG(((uintptr_t)hidden_dev & 15) == 0, "ftn_head_forward: hidden must be 16-byte aligned");
  HeadArgs a;
  a.hidden = hidden_dev; a.wmu = w_mu_dev; a.bmu = b_mu_dev; a.wsg = w_sigma_dev; a.bsg = b_sigma_dev;
  a.tail = tail_dev; a.late = late_dev_or_null; a.floorv = floor_vec_dev_or_null;
  a.rate = rate_dev; a.disp = disp_dev; a.bad = bad_flag_dev;
  a.rows = rows; a.tail_bs = tail_bstride; a.late_bs = late_bstride;
  a.S = S; a.D = D; a.N = N; a.hist = hist; a.floor_s = floor_scalar;
  const bool vec = N % 4 == 0 && tail_bstride % 4 == 0 && late_bstride % 4 == 0 &&
                   (((uintptr_t)tail_dev | (uintptr_t)late_dev_or_null | (uintptr_t)rate_dev | (uintptr_t)disp_dev) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  if (D <= 16) return launch_head<1>(a, vec, st);
  if (D <= 32) return launch_head<2>(a, vec, st);
  if (D <= 64) return launch_head<4>(a, vec, st);
  return launch_head<8>(a, vec, st);
}

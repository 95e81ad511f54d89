// Model-shell kernels either side of the TimesBlock stack:
//   k_head   rate / dispersion heads of TimesNet.forward (reference models/timesnet.py:2066-2102)
//   k_embed_in  value embedding GEMM with the low-rank temporal context folded in (:1958-1996, :1283-1325)
// HBM-bound: each reads its input once and writes its outputs once; the small GEMMs run on the exact
// fp32 MFMA (v_mfma_f32_16x16x4_f32) so no precision is traded.
#include <stdlib.h>
#include "ftn_common.h"
#include "ftn_mlp.h"

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
// MFMA A-fragment order (row i of tile o is series n0 + 16*o + i: the four q-lanes of a row cover one
// 64-byte run per load / store instruction).  Its four waves walk
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
    const int n = n0 + 16 * o + i, k = 16 * S + 4 * q;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    // D % 4 == 0 (checked by the entry point) and k % 4 == 0: the quad is all inside a row or all outside; one
    // 16-byte load instead of four guarded scalar ones (which serialise into four memory round trips)
    if (n < a.N && k < a.D) v = *(const f4*)((h ? a.wsg : a.wmu) + (size_t)n * a.D + k);
    wl[f] = v;
  }
  // per-series constants of the slice: [0] b_mu, [1] b_sigma, [2] dispersion floor
  float* cst = (float*)(wl + 2 * 4 * NS * 64);
  if (threadIdx.x < 192) {
    const int which = threadIdx.x >> 6, n = n0 + (threadIdx.x & 63);
    const bool in = n < a.N;
    float v = which == 2 ? a.floor_s : 0.f;
    if (in) v = which == 0 ? a.bmu[n] : which == 1 ? a.bsg[n] : (a.floorv ? a.floorv[n] : a.floor_s);
    cst[threadIdx.x] = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int nl0 = n0 + 4 * q;           // lane (j, q) of tile o holds series nl0 + 16*o .. +3 of row j
  int badbits = 0;
  for (long long t = (long long)blockIdx.y * 4 + wave; t * 16 < a.rows; t += (long long)gridDim.y * 4) {
    // the weight fragments are re-read from LDS for every row tile: hoisted into registers (128 VGPRs at
    // d_model 64) they leave one wave per SIMD and nothing to hide the hidden/tail loads behind
    asm volatile("" ::: "memory");
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
        if (nl0 + 16 * o < a.N) tv = *(const f4*)(tp + 16 * o);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nl0 + 16 * o + r < a.N) tv[r] = tp[16 * o + r];
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
      const f4 bm = *(const f4*)(cst + 16 * o + 4 * q), bs = *(const f4*)(cst + 64 + 16 * o + 4 * q);
      const f4 fl = *(const f4*)(cst + 128 + 16 * o + 4 * q);
      f4 pre = am[o] + bm + ex[o];                          // mu_head(hidden) + history tail (:2079)
      if (lp) {
        f4 lv = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
          if (nl0 + 16 * o < a.N) lv = *(const f4*)(lp + 16 * o);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nl0 + 16 * o + r < a.N) lv[r] = lp[16 * o + r];
        }
        pre = pre + lv;                                      // + gate * late_bias (:2080-2090)
      }
      f4 rt, dp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rt[r] = softplus20(pre[r]) + 1e-6f;                  // :2091
        dp[r] = softplus20(as[o][r] + bs[r]) + fl[r] + 1e-6f;         // :2092-2094
        if (rok && nl0 + 16 * o + r < a.N) {
          if (!(rt[r] > 0.f && rt[r] <= 3.4028234664e38f)) badbits |= 1;
          if (!(dp[r] > 0.f && dp[r] <= 3.4028234664e38f)) badbits |= 2;
        }
      }
      if (!rok) continue;
      float* __restrict__ rp = a.rate + rr * a.N + nl0 + 16 * o;
      float* __restrict__ dq = a.disp + rr * a.N + nl0 + 16 * o;
      if (VEC) {
        if (nl0 + 16 * o < a.N) {
          __builtin_nontemporal_store(rt, (f4*)rp);
          __builtin_nontemporal_store(dp, (f4*)dq);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nl0 + 16 * o + r < a.N) { rp[r] = rt[r]; dq[r] = dp[r]; }
      }
    }
  }
  if (badbits) atomicOr(a.bad, badbits);
}

// The same heads on the 16-bit matrix pipe (round 3; cf. k_embed_in_bf): 4*rows*N*D flops on the fp32 MFMA were the
// bound (c4 shard: 12.9 GFLOP = 82 us at that pipe's peak beside 45 us of stores).  A workgroup owns NT tiles of 16
// series; both weight slices are split into three bf16 pieces once and sit in LDS as K-32 A fragments
// ([head][tile][slab][piece][lane] x 16 B; 48 KB at d_model 128 with NT = 2 and at d_model 64 with NT = 4); a wave
// splits its 16 hidden rows once per row tile (B operand: lane (j, q) holds k = 32 S + 8 q .. + 7 of row j) and forms
// every product as six v_mfma_f32_16x16x32_bf16.  Epilogue as k_head.
template <int NT, int NS32>
__global__ __launch_bounds__(256, 3) void k_head_bf(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char hl[];     // [2][NT][NS32][3][64] bf8 | 3 x [16 NT] fp32
  bf8* __restrict__ wl = (bf8*)hl;
  constexpr int NW = 16 * NT;                                   // series per workgroup
  const int n0 = blockIdx.x * NW;
  for (int f = threadIdx.x; f < 2 * NT * NS32 * 64; f += 256) {
    const int ln = f & 63;
    int t = f >> 6;
    const int S = t % NS32;
    t /= NS32;
    const int o = t % NT, h = t / NT, i = ln & 15, qa = ln >> 4;
    const int n = n0 + 16 * o + i, k = 32 * S + 8 * qa;
    f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (n < a.N) {                                              // D % 4 == 0: a quad is all inside a row or all outside
      const float* __restrict__ wp = (h ? a.wsg : a.wmu) + (size_t)n * a.D + k;
      if (k < a.D) v0 = *(const f4*)wp;
      if (k + 4 < a.D) v1 = *(const f4*)(wp + 4);
    }
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    bf8 pc[3];
    split_pieces<3>(v, pc);
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) wl[(((h * NT + o) * NS32 + S) * 3 + pz) * 64 + ln] = pc[pz];
  }
  // per-series constants of the slice: [0] b_mu, [1] b_sigma, [2] dispersion floor
  float* cst = (float*)(hl + (size_t)2 * NT * NS32 * 3 * 1024);
  if (threadIdx.x < 3 * NW) {
    const int which = threadIdx.x / NW, n = n0 + (threadIdx.x % NW);
    const bool in = n < a.N;
    float v = which == 2 ? a.floor_s : 0.f;
    if (in) v = which == 0 ? a.bmu[n] : which == 1 ? a.bsg[n] : (a.floorv ? a.floorv[n] : a.floor_s);
    cst[threadIdx.x] = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int nl0 = n0 + 4 * q;           // lane (j, q) of tile o holds series nl0 + 16*o .. +3 of row j
  int badbits = 0;
  for (long long t = (long long)blockIdx.y * 4 + wave; t * 16 < a.rows; t += (long long)gridDim.y * 4) {
    asm volatile("" ::: "memory");      // (the weight fragments are re-read from LDS for every row tile, as in k_head)
    const long long row = t * 16 + j;
    const bool rok = row < a.rows;
    const long long rr = rok ? row : a.rows - 1;
    const long long b = rr / a.S;
    const int s = (int)(rr - b * a.S);
    bf8 hq[NS32][3];
#pragma unroll
    for (int S = 0; S < NS32; ++S) {
      const int k = 32 * S + 8 * q;
      const float* __restrict__ hp = a.hidden + rr * a.D + k;
      const f4 v0 = k < a.D ? *(const f4*)hp : f4{0.f, 0.f, 0.f, 0.f};
      const f4 v1 = k + 4 < a.D ? *(const f4*)(hp + 4) : f4{0.f, 0.f, 0.f, 0.f};
      const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      split_pieces<3>(v, hq[S]);
    }
    const float* __restrict__ tp = a.tail + b * a.tail_bs + (size_t)(s < a.hist ? s : a.hist - 1) * a.N + nl0;
    const float* __restrict__ lp = a.late ? a.late + b * a.late_bs + (size_t)s * a.N + nl0 : nullptr;
    f4 ex[NT];
#pragma unroll
    for (int o = 0; o < NT; ++o) ex[o] = nl0 + 16 * o < a.N ? *(const f4*)(tp + 16 * o) : f4{0.f, 0.f, 0.f, 0.f};
    f4 am[NT], as[NT];
#pragma unroll
    for (int o = 0; o < NT; ++o) am[o] = as[o] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < NS32; ++S)
#pragma unroll
      for (int o = 0; o < NT; ++o) {
        bf8 wm[3], ws[3];
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) {
          wm[pz] = wl[(((0 * NT + o) * NS32 + S) * 3 + pz) * 64 + lane];
          ws[pz] = wl[(((1 * NT + o) * NS32 + S) * 3 + pz) * 64 + lane];
        }
        am[o] = chain_bf<3>(wm, hq[S], am[o]);
        as[o] = chain_bf<3>(ws, hq[S], as[o]);
      }
#pragma unroll
    for (int o = 0; o < NT; ++o) {
      const f4 bm = *(const f4*)(cst + 16 * o + 4 * q), bs = *(const f4*)(cst + NW + 16 * o + 4 * q);
      const f4 fl = *(const f4*)(cst + 2 * NW + 16 * o + 4 * q);
      f4 pre = am[o] + bm + ex[o];                          // mu_head(hidden) + history tail (:2079)
      if (lp && nl0 + 16 * o < a.N) pre = pre + *(const f4*)(lp + 16 * o);      // + gate * late_bias (:2080-2090)
      f4 rt, dp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rt[r] = softplus20(pre[r]) + 1e-6f;                  // :2091
        dp[r] = softplus20(as[o][r] + bs[r]) + fl[r] + 1e-6f;         // :2092-2094
        if (rok && nl0 + 16 * o + r < a.N) {
          if (!(rt[r] > 0.f && rt[r] <= 3.4028234664e38f)) badbits |= 1;
          if (!(dp[r] > 0.f && dp[r] <= 3.4028234664e38f)) badbits |= 2;
        }
      }
      if (!rok || nl0 + 16 * o >= a.N) continue;
      __builtin_nontemporal_store(rt, (f4*)(a.rate + rr * a.N + nl0 + 16 * o));
      __builtin_nontemporal_store(dp, (f4*)(a.disp + rr * a.N + nl0 + 16 * o));
    }
  }
  if (badbits) atomicOr(a.bad, badbits);
}

template <int NT, int NS32>
static int launch_head_bf(const HeadArgs& a, hipStream_t st) {
  const int ntile = (a.N + 16 * NT - 1) / (16 * NT);
  const long long rtiles = (a.rows + 15) / 16;
  long long gy = (rtiles + 3) / 4;
  const long long want = (3 * 256 + ntile - 1) / ntile;       // three workgroups per CU in flight
  if (gy > want) gy = want;
  const size_t lds = (size_t)2 * NT * NS32 * 3 * 1024 + 3 * 16 * NT * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)k_head_bf<NT, NS32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_head_bf): %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL((k_head_bf<NT, NS32>), dim3(ntile, (unsigned)gy), dim3(256), lds, st, a);
  FTN_CHECK_LAUNCH();
  return 0;
}

static const int g_head_f32 = [] { const char* e = getenv("FTN_HEAD_F32"); return e ? atoi(e) : 0; }();   // 1: the fp32-MFMA form

template <int NS>
static int launch_head(const HeadArgs& a, bool vec, hipStream_t st) {
  const int ntile = (a.N + 63) / 64;
  const long long rtiles = (a.rows + 15) / 16;
  long long gy = (rtiles + 3) / 4;
  const long long want = (4 * 256 + ntile - 1) / ntile;       // ~4 workgroups per CU in flight
  if (gy > want) gy = want;
  const size_t lds = (size_t)2 * 4 * NS * 64 * sizeof(f4) + 192 * sizeof(float);
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
  FTN_CHECK_ARG((((uintptr_t)hidden_dev | (uintptr_t)w_mu_dev | (uintptr_t)w_sigma_dev) & 15) == 0,
                "ftn_head_forward: hidden and the head weights must be 16-byte aligned");
  HeadArgs a;
  a.hidden = hidden_dev; a.wmu = w_mu_dev; a.bmu = b_mu_dev; a.wsg = w_sigma_dev; a.bsg = b_sigma_dev;
  a.tail = tail_dev; a.late = late_dev_or_null; a.floorv = floor_vec_dev_or_null;
  a.rate = rate_dev; a.disp = disp_dev; a.bad = bad_flag_dev;
  a.rows = rows; a.tail_bs = tail_bstride; a.late_bs = late_bstride;
  a.S = S; a.D = D; a.N = N; a.hist = hist; a.floor_s = floor_scalar;
  const bool vec = N % 4 == 0 && tail_bstride % 4 == 0 && late_bstride % 4 == 0 &&
                   (((uintptr_t)tail_dev | (uintptr_t)late_dev_or_null | (uintptr_t)rate_dev | (uintptr_t)disp_dev) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  if (vec && !g_head_f32) {
    if (D <= 32) return launch_head_bf<4, 1>(a, st);
    if (D <= 64) return launch_head_bf<4, 2>(a, st);
    return launch_head_bf<2, 4>(a, st);
  }
  if (D <= 16) return launch_head<1>(a, vec, st);
  if (D <= 32) return launch_head<2>(a, vec, st);
  if (D <= 64) return launch_head<4>(a, vec, st);
  return launch_head<8>(a, vec, st);
}

// ---------------------------------------------------------------------------------------------
// Value embedding: out[b][l][:] = x[b][l][:] W^T + add[b?][l][:]   (+ optional LayerNorm over d_model)
// ---------------------------------------------------------------------------------------------
// x is the [B, L, N] input window (rows contiguous, batch stride given), W the nn.Linear weight
// [D][N].  Everything else the reference adds before / after this GEMM is linear in the row and is
// handed over pre-assembled in `add` ([L][D] shared by the batch, or [B][L][D]): the bias, the
// positional / time-feature term (gate * LayerNorm(aux) in "decoupled" mode), and the low-rank
// temporal context and constant context bias pushed through W (so the [B, L, N] context tensor is
// never materialised).  Reads x once: HBM-bound at 4*B*L*N bytes.
struct EmbedArgs {
  const float* x;
  const float* W;        // [D][N]
  const float* add;      // optional
  const float* ln_g;     // optional LayerNorm epilogue ("layer" mode)
  const float* ln_b;
  float* out;            // [B][L][D]
  long long x_bs, add_bs;
  int B, L, N, D;
  float ln_eps;
};

// + add, optional LayerNorm, store.  Lane (j = row, q) holds columns cb(o) .. cb(o)+3 of its rows
template <int RT, int NO>
__device__ __forceinline__ void embed_epilogue(const EmbedArgs& a, f4 (&acc)[RT][NO], const long long (&rr)[RT],
                                               const bool (&rok)[RT], int q) {
  const float invD = 1.0f / (float)a.D;
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const long long b = rr[t] / a.L;
    const float* __restrict__ ap = a.add ? a.add + b * a.add_bs + (rr[t] - b * a.L) * a.D : nullptr;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const int cb = 16 * o + 4 * q;
      if (ap && cb < a.D) acc[t][o] = acc[t][o] + *(const f4*)(ap + cb);      // D % 4 == 0
    }
    if (a.ln_g) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (16 * o + 4 * q < a.D) s += (acc[t][o][0] + acc[t][o][1]) + (acc[t][o][2] + acc[t][o][3]);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * invD;
      float ss = 0.f;
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (16 * o + 4 * q < a.D) {
          const f4 dv = acc[t][o] - mean;
          ss += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
        }
      ss += __shfl_xor(ss, 16);
      ss += __shfl_xor(ss, 32);
      const float rstd = 1.0f / sqrtf(ss * invD + a.ln_eps);
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const int cb = 16 * o + 4 * q;
        if (cb < a.D) acc[t][o] = (acc[t][o] - mean) * rstd * *(const f4*)(a.ln_g + cb) + *(const f4*)(a.ln_b + cb);
      }
    }
    if (!rok[t]) continue;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const int cb = 16 * o + 4 * q;
      if (cb < a.D) *(f4*)(a.out + rr[t] * a.D + cb) = acc[t][o];
    }
  }
}

// Workgroup = 4 waves x RT 16-row tiles; the K (= series) axis is walked in 64-wide chunks whose
// weight slice sits in LDS in A-fragment order, double buffered through registers.  Lane (j, q) ends
// up with columns 16*o + 4*q .. +3 of row j: 64-byte runs per store instruction, and the LayerNorm
// reduction is in-lane plus two xor-shuffles.
template <int NO, bool VEC>
__global__ __launch_bounds__(256, NO <= 4 ? 3 : 1) void k_embed_in(EmbedArgs a) {
  constexpr int RT = 2, KC = 64, NFR = NO * 4 * 64;            // f4 fragments per chunk
  constexpr int PER_T = (NFR + 255) / 256;
  __shared__ f4 wl[2][NFR];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const long long M = (long long)a.B * a.L;
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * (16 * RT);
  long long rr[RT];
  const float* xp[RT];
  bool rok[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const long long row = row0 + 16 * t + j;
    rok[t] = row < M;
    rr[t] = rok[t] ? row : M - 1;
    const long long b = rr[t] / a.L;
    xp[t] = a.x + b * a.x_bs + (rr[t] - b * a.L) * a.N;
  }
  auto load_w = [&](int kc, f4 (&st)[PER_T]) {
#pragma unroll
    for (int p = 0; p < PER_T; ++p) {
      const int f = threadIdx.x + 256 * p;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < NFR) {
        const int ln = f & 63, S = (f >> 6) & 3, o = f >> 8, i = ln & 15, qq = ln >> 4;
        const int col = 16 * o + i, k = kc + 16 * S + 4 * qq;
        if (col < a.D) {
          const float* __restrict__ wp = a.W + (size_t)col * a.N + k;
          if (VEC) {
            if (k < a.N) v = *(const f4*)wp;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k + e < a.N) v[e] = wp[e];
          }
        }
      }
      st[p] = v;
    }
  };
  auto store_w = [&](int buf, const f4 (&st)[PER_T]) {
#pragma unroll
    for (int p = 0; p < PER_T; ++p) {
      const int f = threadIdx.x + 256 * p;
      if (f < NFR) wl[buf][f] = st[p];
    }
  };
  auto load_x = [&](int kc, f4 (&xv)[RT][4]) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int S = 0; S < 4; ++S) {
        const int k = kc + 16 * S + 4 * q;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
          if (k < a.N) v = *(const f4*)(xp[t] + k);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < a.N) v[e] = xp[t][k + e];
        }
        xv[t][S] = v;
      }
  };
  f4 acc[RT][NO];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[t][o] = f4{0.f, 0.f, 0.f, 0.f};
  f4 stg[PER_T], xc[RT][4], xn[RT][4];
  const int nch = (a.N + KC - 1) / KC;
  load_w(0, stg);
  load_x(0, xc);
  store_w(0, stg);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nch;
    if (more) {
      load_w((c + 1) * KC, stg);
      load_x((c + 1) * KC, xn);
    }
#pragma unroll
    for (int S = 0; S < 4; ++S) {
      f4 wf[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) wf[o] = wl[buf][(o * 4 + S) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t][o] = mfma16(wf[o][e], xc[t][S][e], acc[t][o]);
    }
    if (more) {
      store_w(buf ^ 1, stg);
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int S = 0; S < 4; ++S) xc[t][S] = xn[t][S];
    }
    __syncthreads();
  }
  embed_epilogue<RT, NO>(a, acc, rr, rok, q);
}

// The same GEMM on the 16-bit matrix pipe (round 3): at 2*B*L*N*D flops the fp32 MFMA form above is bound by that pipe
// (d_model 128, N 4096: 48 GFLOP = 0.31 ms at its peak; the kernel took 0.61), not by the 4*B*L*N bytes it reads.  x
// and W are split in registers into three bf16 pieces each (exact truncation split, ftn_common.h) and multiplied as six
// v_mfma_f32_16x16x32_bf16 per K-32 slab - the bf16x3 engine's product, fp32-equivalent and without a
// range to guard: raw series values arrive here.  K is walked one slab at a time; the slab's W pieces sit in LDS in
// A-fragment order ([tile][piece][lane] x 16 B), staged through registers one slab ahead, x rows are requested one slab
// ahead as well.  Same accumulator layout as k_embed_in, same epilogue.
template <int NO, int RT>
__global__ __launch_bounds__(256, RT == 1 ? (NO <= 4 ? 4 : 3) : (NO <= 4 ? 3 : 2)) void k_embed_in_bf(EmbedArgs a) {
  constexpr int KC = 32, NFL = NO * 64;                        // fragment lanes per slab (each: 8 k-values x 3 pieces)
  constexpr int PER_T = (NFL + 255) / 256;
  __shared__ bf8 wl[2][NO * 3 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const long long M = (long long)a.B * a.L;
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * (16 * RT);
  long long rr[RT];
  const float* xp[RT];
  bool rok[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const long long row = row0 + 16 * t + j;
    rok[t] = row < M;
    rr[t] = rok[t] ? row : M - 1;
    const long long b = rr[t] / a.L;
    xp[t] = a.x + b * a.x_bs + (rr[t] - b * a.L) * a.N;
  }
  auto load8 = [&](const float* __restrict__ p, int k, f4 (&v)[2]) {          // N % 4 == 0
    v[0] = k < a.N ? *(const f4*)(p + k) : f4{0.f, 0.f, 0.f, 0.f};
    v[1] = k + 4 < a.N ? *(const f4*)(p + k + 4) : f4{0.f, 0.f, 0.f, 0.f};
  };
  auto load_w = [&](int kc, f4 (&st)[PER_T][2]) {
#pragma unroll
    for (int p = 0; p < PER_T; ++p) {
      const int f = threadIdx.x + 256 * p;
      st[p][0] = f4{0.f, 0.f, 0.f, 0.f}; st[p][1] = f4{0.f, 0.f, 0.f, 0.f};
      if (f < NFL) {
        const int ln = f & 63, o = f >> 6, col = 16 * o + (ln & 15);
        if (col < a.D) load8(a.W + (size_t)col * a.N, kc + 8 * (ln >> 4), st[p]);
      }
    }
  };
  auto store_w = [&](int buf, const f4 (&st)[PER_T][2]) {
#pragma unroll
    for (int p = 0; p < PER_T; ++p) {
      const int f = threadIdx.x + 256 * p;
      if (f < NFL) {
        const float v[8] = {st[p][0][0], st[p][0][1], st[p][0][2], st[p][0][3], st[p][1][0], st[p][1][1], st[p][1][2], st[p][1][3]};
        bf8 pc[3];
        split_pieces<3>(v, pc);
        const int ln = f & 63, o = f >> 6;
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) wl[buf][(o * 3 + pz) * 64 + ln] = pc[pz];
      }
    }
  };
  f4 acc[RT][NO];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[t][o] = f4{0.f, 0.f, 0.f, 0.f};
  // x rows are requested TWO slabs ahead (8 waves per CU x 8 KB in flight: one slab ahead left the HBM stream
  // latency-bound), the W slab one ahead
  f4 stg[PER_T][2], xc[RT][2], xn[RT][2], xn2[RT][2];
  const int nch = (a.N + KC - 1) / KC;
  load_w(0, stg);
#pragma unroll
  for (int t = 0; t < RT; ++t) { load8(xp[t], 8 * q, xc[t]); load8(xp[t], KC + 8 * q, xn[t]); }
  store_w(0, stg);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nch;
    if (more) load_w((c + 1) * KC, stg);
#pragma unroll
    for (int t = 0; t < RT; ++t) load8(xp[t], (c + 2) * KC + 8 * q, xn2[t]);      // (zeros past N)
    bf8 xq[RT][3];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float v[8] = {xc[t][0][0], xc[t][0][1], xc[t][0][2], xc[t][0][3], xc[t][1][0], xc[t][1][1], xc[t][1][2], xc[t][1][3]};
      split_pieces<3>(v, xq[t]);
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      bf8 wf[3];
#pragma unroll
      for (int pz = 0; pz < 3; ++pz) wf[pz] = wl[buf][(o * 3 + pz) * 64 + lane];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t][o] = chain_bf<3>(wf, xq[t], acc[t][o]);
    }
    if (more) store_w(buf ^ 1, stg);
#pragma unroll
    for (int t = 0; t < RT; ++t) { xc[t][0] = xn[t][0]; xc[t][1] = xn[t][1]; xn[t][0] = xn2[t][0]; xn[t][1] = xn2[t][1]; }
    __syncthreads();
  }
  embed_epilogue<RT, NO>(a, acc, rr, rok, q);
}

static const int g_embed_f32 = [] { const char* e = getenv("FTN_EMBED_F32"); return e ? atoi(e) : 0; }();   // 1: the fp32-MFMA form

template <int NO>
static int launch_embed(const EmbedArgs& a, bool vec, hipStream_t st) {
  const long long M = (long long)a.B * a.L;
  const unsigned nblk = (unsigned)((M + 127) / 128);
  // d_model 128: one 16-row tile per wave - at 32 rows a wave the c4 shard (46 080 rows) is 1 440 waves for 1 024 SIMDs,
  // two on some, one on others, and the launch lasts as long as the SIMDs with two (350 vs 296 us); d_model <= 64: two
  // tiles per wave share every W fragment read (51 vs 61 us at the bench shape).  FTN_EMBED_RT=1|2 forces one.
  static const int rt_env = [] { const char* e = getenv("FTN_EMBED_RT"); return e ? atoi(e) : 0; }();
  const int rt = rt_env ? rt_env : (NO > 4 ? 1 : 2);
  if (vec && !g_embed_f32 && rt == 1) hipLaunchKernelGGL((k_embed_in_bf<NO, 1>), dim3((unsigned)((M + 63) / 64)), dim3(256), 0, st, a);
  else if (vec && !g_embed_f32) hipLaunchKernelGGL((k_embed_in_bf<NO, 2>), dim3(nblk), dim3(256), 0, st, a);
  else if (vec) hipLaunchKernelGGL((k_embed_in<NO, true>), dim3(nblk), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_embed_in<NO, false>), dim3(nblk), dim3(256), 0, st, a);
  FTN_CHECK_LAUNCH();
  return 0;
}

extern "C" int ftn_embed_forward(const float* x_dev, long long x_bstride, int B, int L, int N, const float* w_dev,
                                 int D, const float* add_dev_or_null, long long add_bstride,
                                 const float* ln_gamma_dev_or_null, const float* ln_beta_dev_or_null, float ln_eps,
                                 float* out_dev, void* stream) {
  FTN_CHECK_ARG(x_dev && w_dev && out_dev, "ftn_embed_forward: null pointer");
  FTN_CHECK_ARG(B >= 1 && L >= 1 && N >= 1, "ftn_embed_forward: bad shape B=%d L=%d N=%d", B, L, N);
  FTN_CHECK_ARG(D >= 4 && D % 4 == 0 && D <= 128, "ftn_embed_forward: d_model=%d must be a multiple of 4, <= 128", D);
  FTN_CHECK_ARG((ln_gamma_dev_or_null == nullptr) == (ln_beta_dev_or_null == nullptr),
                "ftn_embed_forward: LayerNorm needs both gamma and beta");
  FTN_CHECK_ARG((((uintptr_t)out_dev | (uintptr_t)add_dev_or_null | (uintptr_t)ln_gamma_dev_or_null |
                  (uintptr_t)ln_beta_dev_or_null) & 15) == 0 && add_bstride % 4 == 0,
                "ftn_embed_forward: out / add / LayerNorm parameters must be 16-byte aligned");
  EmbedArgs a;
  a.x = x_dev; a.W = w_dev; a.add = add_dev_or_null; a.ln_g = ln_gamma_dev_or_null; a.ln_b = ln_beta_dev_or_null;
  a.out = out_dev; a.x_bs = x_bstride; a.add_bs = add_bstride; a.B = B; a.L = L; a.N = N; a.D = D; a.ln_eps = ln_eps;
  const bool vec = N % 4 == 0 && x_bstride % 4 == 0 && (((uintptr_t)x_dev | (uintptr_t)w_dev) & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  if (D <= 64) return launch_embed<4>(a, vec, st);
  return launch_embed<8>(a, vec, st);
}

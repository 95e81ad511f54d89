// Period selector on gfx950: FFTPeriodSelector.forward + PeriodGrouper.group +
// the softmax/scatter weights of the reference (models/timesnet.py:64-159,
// 513-557, 992-1009), as three launches and no host synchronisation.
//
//   k_spectrum   S1+S2  DFT-as-GEMM on v_mfma_f32_32x32x2_f32 (twiddles x window),
//                       |.|, lower median over channels          -> med[B][F]
//   k_colsum     S2     fixed-order fp64 batch sum               -> psum[F]
//   k_finalize   S3-S5  mean, DC kill, log penalty, top-k (wave arg-max),
//                       periods, grouping, tiling, softmax weights -> FtnDesc, amps, w
#include <math.h>
#include "ftn_common.h"

// ---------------------------------------------------------------- twiddle table
// cos table [L][FPAD] followed by sin table [L][FPAD]; FPAD = F rounded up to 32,
// entries with f >= F are zero.  Angles are reduced with an exact integer modulo
// and evaluated in fp64, so the fp32 table is correctly rounded.
static inline int fpad_of(int L) { return ((L / 2 + 1) + 31) & ~31; }

extern "C" size_t ftn_dft_table_bytes(int L) {
  if (L < 2) return 0;
  return (size_t)2 * L * fpad_of(L) * sizeof(float);
}

__global__ void k_dft_table(float* __restrict__ tab, int L, int F, int FPAD) {
  const int total = L * FPAD;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int t = e / FPAD, f = e - t * FPAD;
    float c = 0.f, s = 0.f;
    if (f < F) {
      const long long m = ((long long)f * t) % L;
      const double ang = 2.0 * (double)m / (double)L;  // in units of pi
      c = (float)cospi(ang);
      s = (float)sinpi(ang);
    }
    tab[e] = c;
    tab[total + e] = s;
  }
}

extern "C" int ftn_dft_table_init(void* table_dev, int L, void* stream) {
  FTN_CHECK_ARG(table_dev && L >= 2, "ftn_dft_table_init: bad table/L=%d", L);
  const int F = L / 2 + 1, FPAD = fpad_of(L);
  const int total = L * FPAD;
  hipLaunchKernelGGL(k_dft_table, dim3(ftn_cdiv(total, 256) < 1024 ? ftn_cdiv(total, 256) : 1024), dim3(256), 0,
                     (hipStream_t)stream, (float*)table_dev, L, F, FPAD);
  FTN_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- S1 + S2
// One workgroup = one batch row b and 32 frequency bins.  Wave w owns channel
// tiles w, w+NW, ... (32 channels each): rows of the MFMA are frequencies
// (A = twiddles, read [t][f] so 32 lanes read 128 contiguous bytes), columns are
// channels (B = x[b][t][c], C fastest -> 128 contiguous bytes per half-wave).
// The amplitude tile goes to LDS and the lower median over channels is taken by
// rank counting (exact ties broken by channel index, i.e. a stable sort).
// Ascending bitonic sort of 64*V values held as V registers per lane (element index =
// 64*i + lane), then the lower median sorted[(C-1)/2]; padding is +inf.  21 compare-exchange
// steps for 64 channels instead of a 64x64 rank count.
template <int V>
__device__ __forceinline__ float wave_lower_median(const float* __restrict__ row, int C, int lane) {
  float v[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < C ? row[c] : INFINITY;
  }
#pragma unroll
  for (int k = 2; k <= 64 * V; k <<= 1) {
#pragma unroll
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      if (jj >= 64) {
        const int di = jj >> 6;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          if ((i & di) == 0) {
            const bool up = ((i * 64) & k) == 0;   // lane bits never reach k >= 128
            const float lo = fminf(v[i], v[i | di]), hi = fmaxf(v[i], v[i | di]);
            v[i] = up ? lo : hi;
            v[i | di] = up ? hi : lo;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const int e = i * 64 + lane;
          const float other = __shfl_xor(v[i], jj);
          const bool up = (e & k) == 0;
          const bool lower = (lane & jj) == 0;
          v[i] = (lower == up) ? fminf(v[i], other) : fmaxf(v[i], other);
        }
      }
    }
  }
  const int t = (C - 1) >> 1;
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
    if ((t >> 6) == i) m = __shfl(v[i], t & 63);
  return m;
}

// Four independent <= 64-element sorts interleaved in one instruction stream: the 21 dependent
// compare-exchange steps of a single sort (each a cross-lane permute) leave the wave waiting on its own
// latency chain, and every workgroup of the launch reaches this phase at the same time.
__device__ __forceinline__ void wave_lower_median_x4(const float* __restrict__ r0, const float* __restrict__ r1,
                                                     const float* __restrict__ r2, const float* __restrict__ r3,
                                                     int C, int lane, float (&m)[4]) {
  float v[4];
  v[0] = lane < C ? r0[lane] : INFINITY;
  v[1] = lane < C ? r1[lane] : INFINITY;
  v[2] = lane < C ? r2[lane] : INFINITY;
  v[3] = lane < C ? r3[lane] : INFINITY;
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      const bool up = (lane & k) == 0;            // k == 64: every lane index is below 64 -> ascending
      const bool lower = (lane & jj) == 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float other = __shfl_xor(v[r], jj);
        v[r] = (lower == up) ? fminf(v[r], other) : fmaxf(v[r], other);
      }
    }
  }
  const int t = (C - 1) >> 1;
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = __shfl(v[r], t);
}

__global__ __launch_bounds__(256) void k_spectrum(const float* __restrict__ x, int L, int C,
                                                  const float* __restrict__ tab, int F, int FPAD,
                                                  float* __restrict__ med) {
  extern __shared__ __attribute__((aligned(16))) float amp[];  // [32][CS]
  const int CS = C + 1;
  const int b = blockIdx.y, f0 = blockIdx.x * 32;
  const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;
  const float* __restrict__ xb = x + (size_t)b * L * C;
  const float* __restrict__ ctab = tab + f0 + i;
  const float* __restrict__ stab = tab + (size_t)L * FPAD + f0 + i;
  const int nct = (C + 31) >> 5;
  for (int ct = wave; ct < nct; ct += nw) {
    const int c = ct * 32 + i;
    const bool cok = c < C;
    f16v re = {0}, im = {0};
    // 8 k-steps (16 time samples) per iteration: 24 independent loads are in flight before
    // the 16 MFMAs that consume them.  Addresses advance by constant strides (no per-load
    // 64-bit multiply, no bounds test) in the main loop; the ragged tail is guarded.
    const float* pc = ctab + (size_t)h * FPAD;
    const float* ps = stab + (size_t)h * FPAD;
    const float* px = xb + (size_t)h * C + (cok ? c : 0);
    const int sT = 2 * FPAD, sX = 2 * C;
    int t = 0;
    // two-deep software pipeline: the loads of block i+1 are issued before the MFMAs of
    // block i (sched_barrier keeps hipcc from sinking them back next to their uses)
    float ac[8], as[8], bv[8];
    if (L >= 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { ac[k] = pc[k * sT]; as[k] = ps[k * sT]; bv[k] = px[k * sX]; }
      pc += 8 * sT; ps += 8 * sT; px += 8 * sX;
    }
    for (; t + 16 <= L; t += 16) {
      float an[8], sn[8], bn[8];
      const bool more = t + 32 <= L;   // wave-uniform
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { an[k] = pc[k * sT]; sn[k] = ps[k * sT]; bn[k] = px[k * sX]; }
        pc += 8 * sT; ps += 8 * sT; px += 8 * sX;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xv = cok ? bv[k] : 0.f;
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[k], xv, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(as[k], xv, im, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ac[k] = an[k]; as[k] = sn[k]; bv[k] = bn[k]; }
      }
    }
    for (; t < L; t += 2) {
      const int tt = t + h;
      const bool tok = tt < L;
      const float ac = tok ? ctab[(size_t)tt * FPAD] : 0.f;
      const float as = tok ? stab[(size_t)tt * FPAD] : 0.f;
      const float bv = (tok && cok) ? xb[(size_t)tt * C + c] : 0.f;
      re = __builtin_amdgcn_mfma_f32_32x32x2f32(ac, bv, re, 0, 0, 0);
      im = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bv, im, 0, 0, 0);
    }
    if (cok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int fi = (r & 3) + 8 * (r >> 2) + 4 * h;
        amp[fi * CS + c] = hypotf(re[r], im[r]);
      }
    }
  }
  __syncthreads();
  const int target = (C - 1) >> 1;  // torch.median == sorted[(C-1)//2]
  if (C <= 64 && (32 % (4 * nw)) == 0) {
    for (int fb = wave; fb < 32; fb += 4 * nw) {          // rows fb, fb+nw, fb+2nw, fb+3nw
      float m[4];
      wave_lower_median_x4(amp + fb * CS, amp + (fb + nw) * CS, amp + (fb + 2 * nw) * CS, amp + (fb + 3 * nw) * CS, C,
                           lane, m);
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (f0 + fb + r * nw < F) med[(size_t)b * F + f0 + fb + r * nw] = m[r];
      }
    }
    return;
  }
  for (int fl = wave; fl < 32; fl += nw) {
    if (f0 + fl >= F) break;
    const float* __restrict__ row = amp + fl * CS;
    if (C <= 256) {
      float m;
      if (C <= 64) m = wave_lower_median<1>(row, C, lane);
      else if (C <= 128) m = wave_lower_median<2>(row, C, lane);
      else m = wave_lower_median<4>(row, C, lane);
      if (lane == 0) med[(size_t)b * F + f0 + fl] = m;
    } else {
      // generic: stable rank count
      for (int c = lane; c < C; c += 64) {
        const float v = row[c];
        int cnt = 0;
        for (int c2 = 0; c2 < C; ++c2) {
          const float v2 = row[c2];
          cnt += (v2 < v || (v2 == v && c2 < c)) ? 1 : 0;
        }
        if (cnt == target) med[(size_t)b * F + f0 + fl] = v;
      }
    }
  }
}

// psum[f] = sum_b med[b][f] in fp64, fixed order: 8 row-strided partial sums per
// column, combined in index order (bitwise reproducible; no atomics).
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ med, int B, int F,
                                                double* __restrict__ psum) {
  __shared__ double part[8][32];
  const int fl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + fl;
  double s = 0.0;
  if (f < F)
    for (int b = bl; b < B; b += 8) s += (double)med[(size_t)b * F + f];
  part[bl][fl] = s;
  __syncthreads();
  if (bl == 0 && f < F) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][fl];
    psum[f] = t;
  }
}

extern "C" int ftn_period_spectrum(const float* x_dev, int B, int L, int C, const void* table_dev,
                                   float* med_dev, double* psum_dev, void* stream) {
  FTN_CHECK_ARG(x_dev && table_dev && med_dev && psum_dev, "ftn_period_spectrum: null pointer");
  FTN_CHECK_ARG(B >= 1 && L >= 2 && C >= 1, "ftn_period_spectrum: bad shape B=%d L=%d C=%d", B, L, C);
  FTN_CHECK_ARG(B <= 65535, "ftn_period_spectrum: B=%d exceeds grid.y", B);
  const int F = L / 2 + 1, FPAD = fpad_of(L);
  const size_t lds = (size_t)32 * (C + 1) * sizeof(float);
  FTN_CHECK_ARG(lds <= 160 * 1024, "ftn_period_spectrum: C=%d too large for the LDS amplitude tile", C);
  int nw = (C + 31) / 32;
  if (nw > 4) nw = 4;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_spectrum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(k_spectrum, dim3(FPAD / 32, B), dim3(64 * nw), lds, (hipStream_t)stream, x_dev, L, C,
                     (const float*)table_dev, F, FPAD, med_dev);
  FTN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_colsum, dim3(ftn_cdiv(F, 32)), dim3(256), 0, (hipStream_t)stream, med_dev, B, F, psum_dev);
  FTN_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- S3 - S5
struct ArgMax { float v; int i; };

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
  // larger value wins; ties -> lower index (torch.topk's tie order is
  // implementation-defined; we fix lowest-index-first)
  if (b.i >= 0 && (a.i < 0 || b.v > a.v || (b.v == a.v && b.i < a.i))) return b;
  return a;
}

__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ psum, int nparts, int Btotal,
                                                  const float* __restrict__ med, int B, int L, int F, int kcfg,
                                                  int pmax, int min_thr, FtnDesc* __restrict__ desc,
                                                  float* __restrict__ amps, float* __restrict__ wts) {
  extern __shared__ __attribute__((aligned(16))) float score[];  // [F]
  __shared__ ArgMax wbest[4];
  __shared__ int sel_idx[FTN_KMAX];
  __shared__ FtnDesc sd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // mean over the (global) batch, DC kill, log penalty          (:112-130)
  for (int f = tid; f < F; f += 256) {
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += psum[(size_t)p * F + f];
    float m = (float)(s / (double)Btotal);
    float sc = m - 1e-8f * log1pf((float)f);
    score[f] = (f == 0) ? -INFINITY : sc;
  }
  __syncthreads();
  int k = kcfg < F - 1 ? kcfg : F - 1;                            // :122-123
  if (k > FTN_KMAX) k = FTN_KMAX;
  if (k < 0) k = 0;
  // top-k: k rounds of block arg-max; a taken bin is marked with NaN
  for (int r = 0; r < k; ++r) {
    ArgMax best = {0.f, -1};
    for (int f = tid; f < F; f += 256) {
      float v = score[f];
      if (v == v) best = better(best, ArgMax{v, f});
    }
    for (int off = 32; off > 0; off >>= 1) {
      ArgMax o;
      o.v = __shfl_xor(best.v, off);
      o.i = __shfl_xor(best.i, off);
      best = better(best, o);
    }
    if (lane == 0) wbest[wave] = best;
    __syncthreads();
    if (tid == 0) {
      ArgMax bb = wbest[0];
      for (int w = 1; w < 4; ++w) bb = better(bb, wbest[w]);
      sel_idx[r] = bb.i;
      if (bb.i >= 0) score[bb.i] = __builtin_nanf("");
    }
    __syncthreads();
  }
  if (tid == 0) {
    int periods[FTN_KMAX];
    int nsel = 0;
    const int hi = pmax < (L - 1 > 1 ? L - 1 : 1) ? pmax : (L - 1 > 1 ? L - 1 : 1);   // :138
    const int lo = min_thr;                                                            // :139
    for (int j = 0; j < FTN_KMAX; ++j) { sd.sel_freq[j] = 0; sd.sel_period[j] = 0; }
    if (hi >= lo) {
      for (int r = 0; r < k; ++r) {
        int idx = sel_idx[r];
        if (idx < 0) continue;
        if (idx < 1) idx = 1;                                     // clamp_min(1) :132
        int p = (L + idx - 1) / idx;                              // :144
        p = p < lo ? lo : (p > hi ? hi : p);                      // :145
        if ((L + p - 1) / p >= 2) {                               // :147-148
          sd.sel_freq[nsel] = idx;
          sd.sel_period[nsel] = p;
          periods[nsel] = p;
          ++nsel;
        }
      }
    }
    sd.n_sel = nsel;
    ftn_build_groups(periods, nsel, L, lo, pmax, &sd);            // grouper min/max = selector's (:972-973)
  }
  __syncthreads();
  // write the descriptor (whole struct, cooperatively)
  {
    const int* src = (const int*)&sd;
    int* dst = (int*)desc;
    for (int e = tid; e < (int)(sizeof(FtnDesc) / 4); e += 256) dst[e] = src[e];
  }
  // per-sample amplitudes and softmax-scatter weights            (:133-135, :992-1009)
  const int nsel = sd.n_sel, G = sd.n_groups;
  for (int b = tid; b < B; b += 256) {
    float a[FTN_KMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < FTN_KMAX; ++j) {
      // unconditional load from a clamped index: a guarded load compiles to load + branch + wait per candidate
      // (serialised memory round trips on this one-workgroup kernel's critical path)
      const float mv = med[(size_t)b * F + (j < nsel ? sd.sel_freq[j] : 0)];
      a[j] = (j < nsel) ? mv : 0.f;
      amps[(size_t)b * FTN_KMAX + j] = a[j];
      if (j < nsel && sd.sel_group[j] >= 0) mx = fmaxf(mx, a[j]);
    }
    float w[FTN_KMAX];
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < FTN_KMAX; ++j) {
      w[j] = 0.f;
      if (j < nsel && sd.sel_group[j] >= 0) { a[j] = expf(a[j] - mx); den += a[j]; } else a[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < FTN_KMAX; ++j) {
      if (j < nsel && sd.sel_group[j] >= 0) {
        const float s = a[j] / den;
        const int g = sd.sel_group[j];
#pragma unroll
        for (int gg = 0; gg < FTN_KMAX; ++gg) if (gg == g) w[gg] += s;
      }
    }
#pragma unroll
    for (int g = 0; g < FTN_KMAX; ++g) wts[(size_t)b * FTN_KMAX + g] = (g < G) ? w[g] : 0.f;
  }
}

extern "C" int ftn_period_finalize(const double* psum_dev, int nparts, int Btotal, const float* med_dev, int B,
                                   int L, int k_periods, int pmax, int min_period_threshold, FtnDesc* desc_dev,
                                   float* amps_dev, float* weights_dev, void* stream) {
  FTN_CHECK_ARG(psum_dev && med_dev && desc_dev && amps_dev && weights_dev, "ftn_period_finalize: null pointer");
  FTN_CHECK_ARG(B >= 1 && L >= 2 && nparts >= 1 && Btotal >= B, "ftn_period_finalize: bad shape");
  FTN_CHECK_ARG(k_periods <= FTN_KMAX, "ftn_period_finalize: k_periods=%d > FTN_KMAX=%d", k_periods, FTN_KMAX);
  // ctor clamps of FFTPeriodSelector (:59-62)
  if (k_periods < 0) k_periods = 0;
  if (pmax < 1) pmax = 1;
  if (min_period_threshold < 1) min_period_threshold = 1;
  if (min_period_threshold > pmax) min_period_threshold = pmax;
  const int F = L / 2 + 1;
  const size_t lds = (size_t)F * sizeof(float);
  FTN_CHECK_ARG(lds <= 48 * 1024, "ftn_period_finalize: L=%d too long", L);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), lds, (hipStream_t)stream, psum_dev, nparts, Btotal, med_dev, B,
                     L, F, k_periods, pmax, min_period_threshold, desc_dev, amps_dev, weights_dev);
  FTN_CHECK_LAUNCH();
  return 0;
}

// Bound on FtnDesc.total_px / n_groups for descriptors written by ftn_period_finalize: the selector can only
// produce periods clamp(ceil(L/i), lo, hi) for rFFT bins i = 1..F-1 (:144-145), at most k of them, so the
// sum of the k largest L + pad over those distinct periods bounds the grid pixels per batch row.  For
// i >= 2 the pad is < i, i.e. (almost) every group is L pixels plus a few; only bin 1 (period L-1) doubles.
extern "C" int ftn_selector_px_bound(int L, int k_periods, int pmax, int min_period_threshold, int* max_groups_out) {
  FTN_CHECK_ARG(L >= 2 && k_periods <= FTN_KMAX, "ftn_selector_px_bound: L=%d k=%d", L, k_periods);
  if (k_periods < 0) k_periods = 0;
  if (pmax < 1) pmax = 1;
  if (min_period_threshold < 1) min_period_threshold = 1;
  if (min_period_threshold > pmax) min_period_threshold = pmax;
  const int F = L / 2 + 1;
  const int k = k_periods < F - 1 ? k_periods : F - 1;
  const int hi = pmax < (L - 1 > 1 ? L - 1 : 1) ? pmax : (L - 1 > 1 ? L - 1 : 1), lo = min_period_threshold;
  int best[FTN_KMAX] = {0};
  int ndist = 0, last = -1;
  if (hi >= lo) {
    for (int i = 1; i < F; ++i) {                 // periods are non-increasing in i: distinct values are runs
      int p = (L + i - 1) / i;
      p = p < lo ? lo : (p > hi ? hi : p);
      if ((L + p - 1) / p < 2 || p == last) continue;
      last = p;
      ++ndist;
      int v = L + (p - (L % p)) % p;
      for (int s = 0; s < k; ++s)
        if (v > best[s]) { int tmp = best[s]; best[s] = v; v = tmp; }
    }
  }
  long long sum = 0;
  for (int s = 0; s < k; ++s) sum += best[s];
  if (max_groups_out) *max_groups_out = ndist < k ? (ndist > 0 ? ndist : 1) : (k > 0 ? k : 1);
  return sum > 0 ? (int)sum : L;
}

extern "C" int ftn_desc_from_periods(const int64_t* periods, int K, int L, int min_period, int max_period,
                                     FtnDesc* d) {
  FTN_CHECK_ARG(periods && d && K >= 0 && K <= FTN_KMAX && L >= 1, "ftn_desc_from_periods: bad argument (K=%d)", K);
  int p32[FTN_KMAX];
  for (int j = 0; j < FTN_KMAX; ++j) { d->sel_freq[j] = 0; d->sel_period[j] = 0; p32[j] = 0; }
  for (int j = 0; j < K; ++j) {
    long long p = periods[j];
    if (p > 0x3fffffff) p = 0x3fffffff;
    if (p < -1) p = -1;
    p32[j] = (int)p;
    d->sel_period[j] = (int)p;
  }
  d->n_sel = K;
  ftn_build_groups(p32, K, L, min_period, max_period, d);
  return 0;
}

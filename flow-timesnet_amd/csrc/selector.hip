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
#include <stdlib.h>
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
// lane ^ J partner value without the LDS crossbar where DPP can do it: J = 1, 2 are quad permutes, J = 8 a
// 16-lane row rotate, J = 4 two bank-masked row shifts (banks = groups of 4 lanes); J = 16, 32 go through
// ds_bpermute.  18 of the 21 bitonic steps of a 64-element sort then cost VALU latency instead of LDS latency.
template <int J>
__device__ __forceinline__ float lane_xor(float v) {
  const int iv = __float_as_int(v);
  if constexpr (J == 1) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false));       // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
  else if constexpr (J == 8) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x128, 0xF, 0xF, false)); // row_ror:8
  else if constexpr (J == 4) {
    int r = __builtin_amdgcn_update_dpp(iv, iv, 0x104, 0xF, 0x5, false);   // row_shl:4 -> lanes of banks 0, 2 read lane + 4
    r = __builtin_amdgcn_update_dpp(r, iv, 0x114, 0xF, 0xA, false);         // row_shr:4 -> lanes of banks 1, 3 read lane - 4
    return __int_as_float(r);
  } else return __shfl_xor(v, J);
}

template <int K, int J>
__device__ __forceinline__ void bitonic_step_x4(float (&v)[4], int lane) {
  const bool up = (lane & K) == 0;                // K == 64: every lane index is below 64 -> ascending
  const bool lower = (lane & J) == 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float other = lane_xor<J>(v[r]);
    v[r] = (lower == up) ? fminf(v[r], other) : fmaxf(v[r], other);
  }
}

// Four independent <= 64-element sorts interleaved in one instruction stream: the 21 dependent
// compare-exchange steps of a single sort leave the wave waiting on its own latency chain.
__device__ __forceinline__ void wave_lower_median_x4(const float* __restrict__ r0, const float* __restrict__ r1,
                                                     const float* __restrict__ r2, const float* __restrict__ r3,
                                                     int C, int lane, float (&m)[4]) {
  float v[4];
  v[0] = lane < C ? r0[lane] : INFINITY;
  v[1] = lane < C ? r1[lane] : INFINITY;
  v[2] = lane < C ? r2[lane] : INFINITY;
  v[3] = lane < C ? r3[lane] : INFINITY;
  bitonic_step_x4<2, 1>(v, lane);
  bitonic_step_x4<4, 2>(v, lane); bitonic_step_x4<4, 1>(v, lane);
  bitonic_step_x4<8, 4>(v, lane); bitonic_step_x4<8, 2>(v, lane); bitonic_step_x4<8, 1>(v, lane);
  bitonic_step_x4<16, 8>(v, lane); bitonic_step_x4<16, 4>(v, lane); bitonic_step_x4<16, 2>(v, lane); bitonic_step_x4<16, 1>(v, lane);
  bitonic_step_x4<32, 16>(v, lane); bitonic_step_x4<32, 8>(v, lane); bitonic_step_x4<32, 4>(v, lane);
  bitonic_step_x4<32, 2>(v, lane); bitonic_step_x4<32, 1>(v, lane);
  bitonic_step_x4<64, 32>(v, lane); bitonic_step_x4<64, 16>(v, lane); bitonic_step_x4<64, 8>(v, lane);
  bitonic_step_x4<64, 4>(v, lane); bitonic_step_x4<64, 2>(v, lane); bitonic_step_x4<64, 1>(v, lane);
  const int t = (C - 1) >> 1;
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = __shfl(v[r], t);
}

__global__ __launch_bounds__(256) void k_spectrum(const float* __restrict__ x, int B, int L, int C,
                                                  const float* __restrict__ tab, int F, int FPAD,
                                                  float* __restrict__ med, int flat) {
  extern __shared__ __attribute__((aligned(16))) float amp[];  // [32][CS]
  const int CS = C + 1;
  // Workgroup -> (batch row, 32-bin block).  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8),
  // each with its own L2: all bin blocks of one batch row are given to the SAME XCD, back to back, so x[b] is
  // fetched from HBM once and re-read from that L2 (a (bin block, b) grid spread a row's blocks over six
  // XCDs and fetched it six times).  Speed only: any mapping is correct.
  const int nfb = FPAD >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = flat ? (int)blockIdx.x / nfb : (slot / nfb) * 8 + xcd;
  if (b >= B) return;
  const int f0 = (flat ? (int)blockIdx.x % nfb : slot % nfb) * 32;
  const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;
  const float* __restrict__ xb = x + (size_t)b * L * C;
  const float* __restrict__ ctab = tab + f0 + i;
  const float* __restrict__ stab = tab + (size_t)L * FPAD + f0 + i;
  const int nct = (C + 31) >> 5;
  // Real input: X[f] = sum_t x[t] e^{-2 pi i f t / L} folds around t = L/2,
  //   Re X[f] =  sum_{tau=0}^{L/2} ce[tau] cos(2 pi f tau / L),   ce[tau] = x[tau] + x[L - tau]
  //   Im X[f] = -sum_{tau=1}^{(L-1)/2} co[tau] sin(2 pi f tau / L), co[tau] = x[tau] - x[L - tau]
  // (tau = 0 and, for even L, tau = L/2 have no partner: ce = x[tau], co = 0), which halves the fp32 MFMA work
  // of the DFT-as-GEMM - the pipe this kernel is bound by - for two extra VALU ops per sample.
  const int KT = (L >> 1) + 1;                         // folded time steps tau = 0 .. L/2
  for (int ct = wave; ct < nct; ct += nw) {
    const int c = ct * 32 + i;
    const bool cok = c < C;
    const int cc = cok ? c : 0;
    f16v re = {0}, im = {0};
    // 8 k-steps (16 folded samples) per iteration, two-deep software pipeline: the loads of block i+1 are issued
    // before the MFMAs of block i (sched_barrier keeps hipcc from sinking them back next to their uses)
    // one guarded k-step (tau = t + h): used for tau = 0, 1 and for the ragged end around L/2
    auto step_guarded = [&](int t) {
      const int tau = t + h;
      const bool tok = tau < KT;
      const int tt = tok ? tau : 0;
      const bool pair = tok && tt > 0 && 2 * tt < L;          // has a distinct partner L - tau
      const float xv = xb[(size_t)tt * C + cc];
      const float xp = xb[(size_t)(pair ? L - tt : tt) * C + cc];
      const float cv = tok ? ctab[(size_t)tt * FPAD] : 0.f;
      const float sv = tok ? stab[(size_t)tt * FPAD] : 0.f;
      const float e = (tok && cok) ? (pair ? xv + xp : xv) : 0.f;
      const float o = (pair && cok) ? xv - xp : 0.f;
      re = __builtin_amdgcn_mfma_f32_32x32x2f32(cv, e, re, 0, 0, 0);
      im = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, o, im, 0, 0, 0);
    };
    step_guarded(0);
    // interior: every tau in [2, tmid) has a distinct partner.  8 k-steps (16 folded samples) per iteration,
    // two-deep software pipeline with constant-stride pointers (no bounds tests, no per-load multiplies):
    // the loads of block i+1 are issued before the MFMAs of block i
    const int nint = (L - 1) / 2 - 1 >= 2 ? (((L - 1) / 2 + 1 - 2) / 16) : 0;     // whole 16-sample blocks in [2, (L-1)/2]
    const int sT = 2 * FPAD, sX = 2 * C;
    const float* pc = ctab + (size_t)(2 + h) * FPAD;
    const float* ps = stab + (size_t)(2 + h) * FPAD;
    const float* px = xb + (size_t)(2 + h) * C + cc;
    const float* pp = xb + (size_t)(L - 2 - h) * C + cc;
    float ac[8], as[8], be[8], bo[8];
    if (nint > 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xv = px[k * sX], xp = pp[-(k * sX)];
        ac[k] = pc[k * sT]; as[k] = ps[k * sT]; be[k] = xv + xp; bo[k] = xv - xp;
      }
      pc += 8 * sT; ps += 8 * sT; px += 8 * sX; pp -= 8 * sX;
    }
    for (int it = 0; it < nint; ++it) {
      float an[8], sn[8], en[8], on[8];
      const bool more = it + 1 < nint;   // wave-uniform
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xv = px[k * sX], xp = pp[-(k * sX)];
          an[k] = pc[k * sT]; sn[k] = ps[k * sT]; en[k] = xv + xp; on[k] = xv - xp;
        }
        pc += 8 * sT; ps += 8 * sT; px += 8 * sX; pp -= 8 * sX;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[k], cok ? be[k] : 0.f, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(as[k], cok ? bo[k] : 0.f, im, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ac[k] = an[k]; as[k] = sn[k]; be[k] = en[k]; bo[k] = on[k]; }
      }
    }
    for (int t = 2 + 16 * nint; t < KT; t += 2) step_guarded(t);
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

// psum[f] = sum_b med[b][f] in fp64, fixed order: 32 row-strided partial sums per column, combined in index
// order (bitwise reproducible; no atomics).
__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ med, int B, int F,
                                                 double* __restrict__ psum) {
  __shared__ double part[32][33];
  const int fl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + fl;
  double s = 0.0;
  if (f < F)
    for (int b = bl; b < B; b += 32) s += (double)med[(size_t)b * F + f];
  part[bl][fl] = s;
  __syncthreads();
  if (bl == 0 && f < F) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += part[k][fl];
    psum[f] = t;
  }
}

extern "C" int ftn_period_spectrum(const float* x_dev, int B, int L, int C, const void* table_dev,
                                   float* med_dev, double* psum_dev, void* stream) {
  FTN_CHECK_ARG(x_dev && table_dev && med_dev && psum_dev, "ftn_period_spectrum: null pointer");
  FTN_CHECK_ARG(B >= 1 && L >= 2 && C >= 1, "ftn_period_spectrum: bad shape B=%d L=%d C=%d", B, L, C);
  FTN_CHECK_ARG((long long)(B + 7) * (fpad_of(L) / 32) < 0x7fffffffLL, "ftn_period_spectrum: B=%d too large", B);
  const int F = L / 2 + 1, FPAD = fpad_of(L);
  const size_t lds = (size_t)32 * (C + 1) * sizeof(float);
  FTN_CHECK_ARG(lds <= 160 * 1024, "ftn_period_spectrum: C=%d too large for the LDS amplitude tile", C);
  int nw = (C + 31) / 32;
  if (nw > 4) nw = 4;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_spectrum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  const int nfb = FPAD / 32;
  hipLaunchKernelGGL(k_spectrum, dim3((unsigned)(ftn_cdiv(B, 8) * 8 * nfb)), dim3(64 * nw), lds, (hipStream_t)stream, x_dev,
                     B, L, C, (const float*)table_dev, F, FPAD, med_dev, getenv("FTN_SEL_FLAT") != nullptr ? 1 : 0);
  FTN_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_colsum, dim3(ftn_cdiv(F, 32)), dim3(1024), 0, (hipStream_t)stream, med_dev, B, F, psum_dev);
  FTN_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- S3 - S5
struct ArgMax { float v; int i; };

// Half-precision inputs (act_dtype 1 = bf16, 2 = fp16): the reference rounds the batch-mean spectrum, the
// scores, the returned amplitudes, the softmax weights and their per-group sums to the input dtype
// (:124, :130, :159, :1000, :1009 scatter_add_ in that dtype); rnd() is that rounding, the identity for fp32.
__device__ __forceinline__ float rnd_act(float v, int act_dtype) {
  if (act_dtype == 1) return (float)(__bf16)v;
  if (act_dtype == 2) return (float)(_Float16)v;
  return v;
}

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
  // larger value wins; ties -> lower index (torch.topk's tie order is
  // implementation-defined; we fix lowest-index-first)
  if (b.i >= 0 && (a.i < 0 || b.v > a.v || (b.v == a.v && b.i < a.i))) return b;
  return a;
}

__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ psum, int nparts, int Btotal,
                                                  const float* __restrict__ med, int B, int L, int F, int kcfg,
                                                  int pmax, int min_thr, FtnDesc* __restrict__ desc,
                                                  float* __restrict__ amps, float* __restrict__ wts, int act_dtype,
                                                  int max_unique, float log_base) {
  extern __shared__ __attribute__((aligned(16))) float score[];  // [F]
  __shared__ int sel_idx[FTN_KMAX];
  __shared__ FtnDesc sd;
  __shared__ float red[256][FTN_KMAX + 1];                        // block reductions of the flagged grouping
  __shared__ float colmean[FTN_KMAX], gscore[FTN_KMAX];
  __shared__ int c_assign[FTN_KMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // mean over the (global) batch, DC kill, log penalty          (:112-130)
  for (int f = tid; f < F; f += 256) {
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += psum[(size_t)p * F + f];
    float m = rnd_act((float)(s / (double)Btotal), act_dtype);
    float sc = rnd_act(m - rnd_act(1e-8f * rnd_act(log1pf((float)f), act_dtype), act_dtype), act_dtype);
    score[f] = (f == 0) ? -INFINITY : sc;
  }
  __syncthreads();
  int k = kcfg < F - 1 ? kcfg : F - 1;                            // :122-123
  if (k > FTN_KMAX) k = FTN_KMAX;
  if (k < 0) k = 0;
  // top-k by ONE wavefront, no barriers: every lane keeps the best of its strided share of the bins, a
  // shuffle butterfly reduces the 64 candidates, the winner's bin is retired (NaN) and the lane that owned it
  // rescans its share.  k <= 16 rounds of ~12 shuffles; ties resolve to the lowest bin index.
  if (wave == 0) {
    ArgMax mine = {0.f, -1};
    for (int f = lane; f < F; f += 64) {
      const float v = score[f];
      if (v == v) mine = better(mine, ArgMax{v, f});
    }
    for (int r = 0; r < k; ++r) {
      ArgMax best = mine;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        ArgMax o;
        o.v = __shfl_xor(best.v, off);
        o.i = __shfl_xor(best.i, off);
        best = better(best, o);
      }
      if (lane == 0) sel_idx[r] = best.i;
      if (best.i >= 0 && (best.i & 63) == lane) {          // owner: retire the bin, rescan its share
        score[best.i] = __builtin_nanf("");
        mine = ArgMax{0.f, -1};
        for (int f = lane; f < F; f += 64) {
          const float v = score[f];
          if (v == v) mine = better(mine, ArgMax{v, f});
        }
      }
    }
  }
  __syncthreads();
  // periods, validity, grouping and tiling by the lanes of wave 0 in parallel - lane j owns candidate j, then
  // group j (FTN_KMAX <= 64).  Integer division has no hardware instruction here (~40 VALU ops each), and the
  // ~100 divisions of this section (period = ceil(L/idx), pad, cycles, the tile-geometry search) used to run one
  // after another on a single lane: 13 us of a 24 us kernel.  Same results as ftn_build_groups (host).
  if (wave == 0) {
    const int hi = pmax < (L - 1 > 1 ? L - 1 : 1) ? pmax : (L - 1 > 1 ? L - 1 : 1);   // :138
    const int lo = min_thr;                                                            // :139
    // -- candidate j: period, kept by the selector? (:144-148)
    int idx = (lane < k) ? sel_idx[lane] : -1;
    bool kept = false;
    int p = 0;
    if (idx >= 0 && hi >= lo) {
      if (idx < 1) idx = 1;                                       // clamp_min(1) :132
      p = (L + idx - 1) / idx;                                    // :144
      p = p < lo ? lo : (p > hi ? hi : p);                        // :145
      kept = (L + p - 1) / p >= 2;                                // :147-148
    }
    const unsigned long long keptm = __ballot(kept);
    const int nsel = __popcll(keptm);
    const int slot = __popcll(keptm & ((1ull << lane) - 1ull));   // position among the kept candidates (score order)
    if (lane < FTN_KMAX) { sd.sel_freq[lane] = 0; sd.sel_period[lane] = 0; sd.sel_group[lane] = -1; }
    if (kept) { sd.sel_freq[slot] = idx; sd.sel_period[slot] = p; }
    // -- grouping (PeriodGrouper.group, flags unset, :513-557): lane s < nsel now owns kept candidate s
    // period of kept candidate `lane`: gather from its owner = the lane holding the (lane+1)-th set bit of keptm
    int owner = 0;
    {
      unsigned long long m = keptm;
      for (int t = 0; t < FTN_KMAX; ++t) {                        // t-th set bit -> lane t
        const int bit = m ? __ffsll((unsigned long long)m) - 1 : 0;
        if (t == lane) owner = bit;
        m &= m - 1ull;
      }
    }
    const int pc = __shfl(p, owner);                              // period of kept candidate `lane` (lane < nsel)
    // grouper filter: p > 0, lo <= p <= pmax, cycles >= 2 (:517-543)
    int pad = 0, cyc = 0;
    bool valid = false;
    if (lane < nsel && pc > 0 && pc >= lo && pc <= pmax) {
      pad = (pc - (L % pc)) % pc;
      cyc = (L + pad) / pc;
      valid = cyc >= 2;
    }
    // distinct valid periods, ascending: first = no earlier candidate with the same period;
    // rank = number of distinct valid periods below mine
    bool first = valid;
    int rank = 0;
    for (int t = 0; t < FTN_KMAX; ++t) {
      const int pt = __shfl(pc, t);
      const bool vt = __shfl((int)valid, t) != 0;
      if (vt && pt == pc && t < lane) first = false;
    }
    for (int t = 0; t < FTN_KMAX; ++t) {
      const int pt = __shfl(pc, t);
      const bool ft = __shfl((int)first, t) != 0;
      if (ft && pt < pc) ++rank;
    }
    const int G = __popcll(__ballot(first));
    if (valid) sd.sel_group[lane] = rank;
    int tw = 0, th = 0, ntx = 0, nty = 0;
    if (first) ftn_tile_geometry(cyc, pc, &tw, &th, &ntx, &nty);
    if (lane < FTN_KMAX) {                                        // defaults for the unused group slots
      sd.g_period[lane] = 0; sd.g_pad[lane] = 0; sd.g_cycles[lane] = 0;
      sd.g_tw[lane] = 0; sd.g_th[lane] = 0; sd.g_ntx[lane] = 0; sd.g_nty[lane] = 0;
    }
    if (first) {
      sd.g_period[rank] = pc; sd.g_pad[rank] = pad; sd.g_cycles[rank] = cyc;
      sd.g_tw[rank] = tw; sd.g_th[rank] = th; sd.g_ntx[rank] = ntx; sd.g_nty[rank] = nty;
    }
    // prefix sums over the groups in ascending order: lane g re-reads group g (same wave: LDS ops are in order)
    const int gpx = (lane < G) ? L + sd.g_pad[lane] : 0;
    const int gtl = (lane < G) ? sd.g_ntx[lane] * sd.g_nty[lane] : 0;
    int opx = 0, otl = 0;                                         // exclusive prefix of lanes < lane
    for (int t = 0; t < FTN_KMAX; ++t) {
      const int a_ = __shfl(gpx, t), b_ = __shfl(gtl, t);
      if (t < lane) { opx += a_; otl += b_; }
    }
    if (lane <= FTN_KMAX) { sd.g_px_off[lane] = opx; sd.g_tile_off[lane] = otl; }   // lanes >= G hold the totals
    if (lane == 0) {
      sd.n_sel = nsel; sd.n_groups = G;
    }
    if (lane == FTN_KMAX) { sd.total_px = opx; sd.tiles_per_row = otl; }
  }
  __syncthreads();
  // ---- TIMES_PERIOD_BINNING / TIMES_PERIOD_MAX_UNIQ (reference :350-437; resolved per block depth on the host and
  //      passed in): candidates are grouped by log bucket instead of by period, and / or only the `max_unique`
  //      groups with the largest batch-mean logsumexp survive, the others joining the kept group of nearest
  //      period.  A group's period is its member with the largest batch-mean amplitude.  Needs two reductions
  //      over the batch (column means, group scores), done in a fixed order; the small-K logic runs on thread 0.
  if (max_unique > 0 || log_base > 1.0f) {
    const int nsel = sd.n_sel;
    float part[FTN_KMAX];
#pragma unroll
    for (int j = 0; j < FTN_KMAX; ++j) part[j] = 0.f;
    for (int b = tid; b < B; b += 256)
#pragma unroll
      for (int j = 0; j < FTN_KMAX; ++j)
        if (j < nsel) part[j] += rnd_act(med[(size_t)b * F + sd.sel_freq[j]], act_dtype);
#pragma unroll
    for (int j = 0; j < FTN_KMAX; ++j) red[tid][j] = part[j];
    __syncthreads();
    if (tid < FTN_KMAX) {
      float t = 0.f;
      for (int r = 0; r < 256; ++r) t += red[r][tid];
      colmean[tid] = rnd_act(t / (float)B, act_dtype);
    }
    __syncthreads();
    // validity of every kept candidate under the grouper's filter (:517-543), initial assignment by key (:547-551)
    if (tid == 0) {
      int key[FTN_KMAX];
      for (int j = 0; j < FTN_KMAX; ++j) c_assign[j] = -1;
      for (int j = 0; j < nsel; ++j) {
        const int p = sd.sel_period[j];
        bool ok = p > 0 && p >= min_thr && p <= pmax;
        if (ok) { const int pad = (p - (L % p)) % p; ok = (L + pad) / p >= 2; }
        key[j] = !ok ? -1 : (log_base > 1.0f ? (int)floorf(logf((float)p) / logf(log_base) + 1e-6f) : p);   // :350-354
        if (!ok) continue;
        // assignment id = rank of the key among the distinct keys (sorted ascending), filled below
      }
      for (int j = 0; j < nsel; ++j) {
        if (key[j] < 0) continue;
        int rank = 0;
        for (int i = 0; i < nsel; ++i) {
          if (key[i] < 0 || key[i] >= key[j]) continue;
          bool firstocc = true;
          for (int h = 0; h < i; ++h) if (key[h] == key[i]) firstocc = false;
          if (firstocc) ++rank;
        }
        c_assign[j] = rank;
      }
    }
    __syncthreads();
    int ngroups = 0;
    for (int j = 0; j < nsel; ++j) if (c_assign[j] + 1 > ngroups) ngroups = c_assign[j] + 1;
    if (max_unique > 0 && ngroups > max_unique) {
      // group scores = batch mean of logsumexp over the members' amplitudes (:373, :386)
      float ps[FTN_KMAX];
#pragma unroll
      for (int g = 0; g < FTN_KMAX; ++g) ps[g] = 0.f;
      for (int b = tid; b < B; b += 256) {
        float a[FTN_KMAX];
#pragma unroll
        for (int j = 0; j < FTN_KMAX; ++j) a[j] = j < nsel ? rnd_act(med[(size_t)b * F + sd.sel_freq[j]], act_dtype) : 0.f;
        for (int g = 0; g < ngroups; ++g) {
          float mx = -INFINITY;
#pragma unroll
          for (int j = 0; j < FTN_KMAX; ++j) if (j < nsel && c_assign[j] == g) mx = fmaxf(mx, a[j]);
          float se = 0.f;
#pragma unroll
          for (int j = 0; j < FTN_KMAX; ++j) if (j < nsel && c_assign[j] == g) se += expf(a[j] - mx);
          ps[g] += rnd_act(logf(se) + mx, act_dtype);
        }
      }
#pragma unroll
      for (int g = 0; g < FTN_KMAX; ++g) red[tid][g] = ps[g];
      __syncthreads();
      if (tid < FTN_KMAX) {
        float t = 0.f;
        for (int r = 0; r < 256; ++r) t += red[r][tid];
        gscore[tid] = rnd_act(t / (float)B, act_dtype);
      }
      __syncthreads();
      if (tid == 0) {
        // canonical period of each group = member with the largest column mean (first on ties, :374-378)
        int gper[FTN_KMAX], keep[FTN_KMAX];
        bool kept[FTN_KMAX];
        for (int g = 0; g < ngroups; ++g) {
          int best = -1;
          for (int j = 0; j < nsel; ++j)
            if (c_assign[j] == g && (best < 0 || colmean[j] > colmean[best])) best = j;
          gper[g] = sd.sel_period[best];
          kept[g] = false;
        }
        for (int r = 0; r < max_unique; ++r) {                    // top-k by score, descending, lowest id on ties
          int best = -1;
          for (int g = 0; g < ngroups; ++g)
            if (!kept[g] && (best < 0 || gscore[g] > gscore[best])) best = g;
          keep[r] = best; kept[best] = true;
        }
        int target[FTN_KMAX];
        for (int g = 0; g < ngroups; ++g) {
          target[g] = g;
          if (kept[g]) continue;
          int bt = 0;
          float bd = fabsf((float)gper[keep[0]] - (float)gper[g]);
          for (int r = 1; r < max_unique; ++r) {                  // nearest kept period, first in keep order (:421-424)
            const float dd = fabsf((float)gper[keep[r]] - (float)gper[g]);
            if (dd < bd) { bd = dd; bt = r; }
          }
          target[g] = keep[bt];
        }
        for (int j = 0; j < nsel; ++j) if (c_assign[j] >= 0) c_assign[j] = target[c_assign[j]];
      }
      __syncthreads();
    }
    if (tid == 0) {
      // final metadata (:439-511): group period = canonical member, groups ordered by (period, canonical index)
      int gid[FTN_KMAX], gcan[FTN_KMAX], G = 0;
      for (int j = 0; j < nsel; ++j) {
        if (c_assign[j] < 0) continue;
        bool seen = false;
        for (int g = 0; g < G; ++g) if (gid[g] == c_assign[j]) seen = true;
        if (!seen) gid[G++] = c_assign[j];
      }
      for (int g = 0; g < G; ++g) {
        int best = -1;
        for (int j = 0; j < nsel; ++j)
          if (c_assign[j] == gid[g] && (best < 0 || colmean[j] > colmean[best])) best = j;
        gcan[g] = best;
      }
      for (int a_ = 1; a_ < G; ++a_) {                            // insertion sort by (period, canonical index)
        const int vg = gid[a_], vc = gcan[a_];
        int b_ = a_ - 1;
        while (b_ >= 0 && (sd.sel_period[gcan[b_]] > sd.sel_period[vc] ||
                           (sd.sel_period[gcan[b_]] == sd.sel_period[vc] && gcan[b_] > vc))) {
          gid[b_ + 1] = gid[b_]; gcan[b_ + 1] = gcan[b_]; --b_;
        }
        gid[b_ + 1] = vg; gcan[b_ + 1] = vc;
      }
      int px = 0, tiles = 0;
      for (int g = 0; g < FTN_KMAX; ++g) {
        if (g < G) {
          const int p = sd.sel_period[gcan[g]];
          const int pad = (p - (L % p)) % p;
          sd.g_period[g] = p; sd.g_pad[g] = pad; sd.g_cycles[g] = (L + pad) / p;
          sd.g_px_off[g] = px; px += L + pad;
          ftn_tile_geometry(sd.g_cycles[g], p, &sd.g_tw[g], &sd.g_th[g], &sd.g_ntx[g], &sd.g_nty[g]);
          sd.g_tile_off[g] = tiles; tiles += sd.g_ntx[g] * sd.g_nty[g];
        } else {
          sd.g_period[g] = 0; sd.g_pad[g] = 0; sd.g_cycles[g] = 0;
          sd.g_tw[g] = 0; sd.g_th[g] = 0; sd.g_ntx[g] = 0; sd.g_nty[g] = 0;
          sd.g_px_off[g] = px; sd.g_tile_off[g] = tiles;
        }
      }
      sd.g_px_off[FTN_KMAX] = px; sd.g_tile_off[FTN_KMAX] = tiles;
      for (int j = 0; j < FTN_KMAX; ++j) {
        int m = -1;
        if (j < nsel && c_assign[j] >= 0)
          for (int g = 0; g < G; ++g) if (gid[g] == c_assign[j]) m = g;
        sd.sel_group[j] = m;
      }
      sd.n_groups = G; sd.total_px = px; sd.tiles_per_row = tiles;
    }
    __syncthreads();
  }
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
      a[j] = (j < nsel) ? rnd_act(mv, act_dtype) : 0.f;
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
        const float s = rnd_act(a[j] / den, act_dtype);
        const int g = sd.sel_group[j];
#pragma unroll
        for (int gg = 0; gg < FTN_KMAX; ++gg) if (gg == g) w[gg] = rnd_act(w[gg] + s, act_dtype);
      }
    }
#pragma unroll
    for (int g = 0; g < FTN_KMAX; ++g) wts[(size_t)b * FTN_KMAX + g] = (g < G) ? w[g] : 0.f;
  }
}

extern "C" int ftn_period_finalize(const double* psum_dev, int nparts, int Btotal, const float* med_dev, int B,
                                   int L, int k_periods, int pmax, int min_period_threshold, int act_dtype,
                                   int max_unique, float log_base, FtnDesc* desc_dev, float* amps_dev,
                                   float* weights_dev, void* stream) {
  FTN_CHECK_ARG(psum_dev && med_dev && desc_dev && amps_dev && weights_dev, "ftn_period_finalize: null pointer");
  FTN_CHECK_ARG(B >= 1 && L >= 2 && nparts >= 1 && Btotal >= B, "ftn_period_finalize: bad shape");
  FTN_CHECK_ARG(k_periods <= FTN_KMAX, "ftn_period_finalize: k_periods=%d > FTN_KMAX=%d", k_periods, FTN_KMAX);
  FTN_CHECK_ARG(act_dtype >= 0 && act_dtype <= 2, "ftn_period_finalize: act_dtype=%d", act_dtype);
  // ctor clamps of FFTPeriodSelector (:59-62)
  if (k_periods < 0) k_periods = 0;
  if (pmax < 1) pmax = 1;
  if (min_period_threshold < 1) min_period_threshold = 1;
  if (min_period_threshold > pmax) min_period_threshold = pmax;
  const int F = L / 2 + 1;
  const size_t lds = (size_t)F * sizeof(float);
  FTN_CHECK_ARG(lds <= 48 * 1024, "ftn_period_finalize: L=%d too long", L);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), lds, (hipStream_t)stream, psum_dev, nparts, Btotal, med_dev, B,
                     L, F, k_periods, pmax, min_period_threshold, desc_dev, amps_dev, weights_dev, act_dtype,
                     max_unique > 0 ? max_unique : 0, log_base > 1.0f ? log_base : 0.f);
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

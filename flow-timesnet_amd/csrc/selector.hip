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
#include <string.h>
#include "ftn_common.h"
#include "ftn_finalize.h"

// ---------------------------------------------------------------- twiddle table
// cos table [L][FPAD] followed by sin table [L][FPAD]; FPAD = F rounded up to 32,
// entries with f >= F are zero.  Angles are reduced with an exact integer modulo
// and evaluated in fp64, so the fp32 table is correctly rounded.
static inline int fpad_of(int L) { return ((L / 2 + 1) + 31) & ~31; }

// Quarter-fold tables (L % 4 == 0), appended behind the two [L][FPAD] planes: for even bins f = 2m and odd bins
// f = 2m + 1 the twiddles at tau = 0 .. L/4, four planes [QP][FQ]: cos even, sin even, cos odd, sin odd
// (QP = L/4 + 1 rounded up to even, FQ = number of even bins rounded up to 32; zero outside).
static inline int qfold_qp(int L) { return ((L / 4 + 1) + 1) & ~1; }
static inline int qfold_fq(int L) { return (((L / 2 + 1) + 1) / 2 + 31) & ~31; }
static inline bool qfold_ok(int L) { return L >= 8 && (L & 3) == 0; }

extern "C" size_t ftn_dft_table_bytes(int L) {
  if (L < 2) return 0;
  size_t n = (size_t)2 * L * fpad_of(L);
  if (qfold_ok(L)) n += (size_t)4 * qfold_qp(L) * qfold_fq(L);
  return n * sizeof(float);
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

__global__ void k_dft_table_q(float* __restrict__ qt, int L, int F, int QP, int FQ) {
  const int total = QP * FQ, Q = L / 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * total; e += gridDim.x * blockDim.x) {
    const int odd = e >= total ? 1 : 0;
    const int r = e - odd * total, t = r / FQ, m = r - t * FQ;
    const int f = 2 * m + odd;
    float c = 0.f, s = 0.f;
    if (f < F && t <= Q) {
      const long long k = ((long long)f * t) % L;
      const double ang = 2.0 * (double)k / (double)L;
      c = (float)cospi(ang);
      s = (float)sinpi(ang);
    }
    qt[(size_t)(2 * odd) * total + r] = c;
    qt[(size_t)(2 * odd + 1) * total + r] = s;
  }
}

extern "C" int ftn_dft_table_init(void* table_dev, int L, void* stream) {
  FTN_CHECK_ARG(table_dev && L >= 2, "ftn_dft_table_init: bad table/L=%d", L);
  const int F = L / 2 + 1, FPAD = fpad_of(L);
  const int total = L * FPAD;
  hipLaunchKernelGGL(k_dft_table, dim3(ftn_cdiv(total, 256) < 1024 ? ftn_cdiv(total, 256) : 1024), dim3(256), 0,
                     (hipStream_t)stream, (float*)table_dev, L, F, FPAD);
  FTN_CHECK_LAUNCH();
  if (qfold_ok(L)) {
    const int QP = qfold_qp(L), FQ = qfold_fq(L);
    hipLaunchKernelGGL(k_dft_table_q, dim3(ftn_cdiv(2 * QP * FQ, 256) < 1024 ? ftn_cdiv(2 * QP * FQ, 256) : 1024), dim3(256), 0,
                       (hipStream_t)stream, (float*)table_dev + (size_t)2 * total, L, F, QP, FQ);
    FTN_CHECK_LAUNCH();
  }
  return 0;
}

// |re + i im| as sqrt(re^2 + im^2) with the raw hardware square root (v_sqrt_f32, <= 1 ulp): four instructions
// instead of the ~40 of hypotf's scaling paths, sixteen times per lane after the DFT loop (2 us of the row kernels'
// 25).  The squares of a DFT amplitude of fp32 data stay far inside the fp32 range (|X| <= L max|x|: 1e7 for
// inputs of 3e4 squares to 1e14); below 1e-19 the square underflows and the amplitude reads 0 - noise bins of a
// constant series, which the selector ranks last either way.  All three kernels use this form, so k_spectrum and
// k_spectrum_row stay bit-identical.
__device__ __forceinline__ float amp2(float re, float im) { return __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)); }

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

// The same 21-step network on NR independent rows, three issue slots per compare-exchange: the partner value comes
// from a raw v_mov_b32_dpp (hipcc wraps update_dpp in a copy, a canonicalising v_max and min + max + select: 8 slots,
// and chains its four rows through one temporary), and the exchange itself is ONE v_med3_f32 against -inf (keep the
// smaller) or +inf (keep the larger) - a per-lane constant that depends on the step only and is shared by the rows.
// Values are only permuted, so the median is the same bits as the other forms'.  NaN-free inputs (amplitudes).
template <int J>
__device__ __forceinline__ float lane_xor_raw(float v) {
  float o;
  if constexpr (J == 1) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
  else if constexpr (J == 2) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
  else if constexpr (J == 8) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
  else if constexpr (J == 4)
    asm volatile("v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(o) : "v"(v));
  else o = __shfl_xor(v, J);
  return o;
}

template <int K, int J, int NR>
__device__ __forceinline__ void bitonic_step_med3(float (&v)[NR], int lane) {
  const bool keepmin = ((lane & K) == 0) == ((lane & J) == 0);   // K == 64: ascending everywhere
  const float sel = keepmin ? -INFINITY : INFINITY;
  float o[NR];
  // a DPP read needs two wait states after the VALU write of its source, and inline asm is opaque to hipcc's hazard
  // pass: ONE s_nop 1 in front of the step's DPP group (fenced so nothing that writes v[] can slip in behind it)
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (J < 16) asm volatile("s_nop 1");
#pragma unroll
  for (int r = 0; r < NR; ++r) o[r] = lane_xor_raw<J>(v[r]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < NR; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], o[r], sel);
}

template <int NR>
__device__ __forceinline__ void wave_lower_median_rows(const float* __restrict__ base, int stride, int C, int lane,
                                                       float (&m)[NR]) {
  float v[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) v[r] = lane < C ? base[(size_t)r * stride + lane] : INFINITY;
  bitonic_step_med3<2, 1, NR>(v, lane);
  bitonic_step_med3<4, 2, NR>(v, lane); bitonic_step_med3<4, 1, NR>(v, lane);
  bitonic_step_med3<8, 4, NR>(v, lane); bitonic_step_med3<8, 2, NR>(v, lane); bitonic_step_med3<8, 1, NR>(v, lane);
  bitonic_step_med3<16, 8, NR>(v, lane); bitonic_step_med3<16, 4, NR>(v, lane); bitonic_step_med3<16, 2, NR>(v, lane);
  bitonic_step_med3<16, 1, NR>(v, lane);
  bitonic_step_med3<32, 16, NR>(v, lane); bitonic_step_med3<32, 8, NR>(v, lane); bitonic_step_med3<32, 4, NR>(v, lane);
  bitonic_step_med3<32, 2, NR>(v, lane); bitonic_step_med3<32, 1, NR>(v, lane);
  bitonic_step_med3<64, 32, NR>(v, lane); bitonic_step_med3<64, 16, NR>(v, lane); bitonic_step_med3<64, 8, NR>(v, lane);
  bitonic_step_med3<64, 4, NR>(v, lane); bitonic_step_med3<64, 2, NR>(v, lane); bitonic_step_med3<64, 1, NR>(v, lane);
  const int t = (C - 1) >> 1;
#pragma unroll
  for (int r = 0; r < NR; ++r) m[r] = __shfl(v[r], t);
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
      // (complete before the loop is entered: see k_spectrum_rowq)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(ac[k]), "+v"(as[k]), "+v"(be[k]), "+v"(bo[k]));
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
        amp[fi * CS + c] = amp2(re[r], im[r]);
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

// ---------------------------------------------------------------- S1 + S2, one batch row per workgroup
// The same DFT and median with x[b] resident in LDS.  k_spectrum gives every (row, 32-bin block) its own
// workgroup, so each of a row's bin blocks re-reads and re-folds x[b] with two dword loads per lane per k-step
// beside the two twiddle loads: four vector-memory instructions per MFMA pair, as much time on the texture path as
// on the matrix pipe.  Here ONE workgroup owns the row: it folds x[b] once into LDS (ce = x[tau] + x[L - tau],
// co = x[tau] - x[L - tau], float4 loads), wave w takes (bin block w % nfb, channel tile w / nfb) and reads its B
// operands with conflict-free ds_read_b32 - the only global loads left in the loop are the twiddles - and the
// whole [FPAD][C] amplitude tile of the row stays in LDS for the medians.  Same MFMA sequence on the same
// operands as k_spectrum: the result is bit-identical.  Needs (KT + 1) * CP * 8 + FPAD * (C + 1) * 4 bytes of LDS
// and nfb * nct <= 16 waves (L = 336, C = 64: 136 KB, 12 waves = three per SIMD); other shapes keep k_spectrum.
__global__ __launch_bounds__(1024) void k_spectrum_row(const float* __restrict__ x, int B, int L, int C,
                                                       const float* __restrict__ tab, int F, int FPAD,
                                                       float* __restrict__ med) {
  extern __shared__ __attribute__((aligned(16))) float lds_row[];
  const int KT = (L >> 1) + 1;                         // folded time steps tau = 0 .. L/2
  const int KTP = (KT + 1) & ~1;                       // rows incl. the zero row an odd KT's last k-step reads
  const int nct = (C + 31) >> 5, CP = nct * 32, CS = C + 1;
  float* __restrict__ ce = lds_row;                    // [KTP][CP]
  float* __restrict__ co = ce + (size_t)KTP * CP;      // [KTP][CP]
  float* __restrict__ amp = co + (size_t)KTP * CP;     // [FPAD][CS]
  const int b = blockIdx.x;
  const int nfb = FPAD >> 5;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63;
  const float* __restrict__ xb = x + (size_t)b * L * C;
  // ---- fold the row into LDS (same expressions as k_spectrum's step_guarded: bit-identical operands)
  if ((C & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
    // four float4 pairs in flight per thread: a plain load -> fold -> store loop pays one HBM round trip per pass
    const int c4n = CP >> 2, total = KTP * c4n;
    for (int e0 = tid; e0 < total; e0 += 4 * nthr) {
      f4 xv[4], xp[4];
      int tau[4], c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nthr;
        tau[u] = e / c4n; c[u] = (e - tau[u] * c4n) * 4;
        const bool ok = e < total && tau[u] < KT && c[u] < C;
        const bool pair = ok && tau[u] > 0 && 2 * tau[u] < L;
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        xv[u] = ok ? *(const f4*)(xb + (size_t)tau[u] * C + c[u]) : z;
        xp[u] = pair ? *(const f4*)(xb + (size_t)(L - tau[u]) * C + c[u]) : z;
        if (!pair) xp[u] = z;
        tau[u] = pair ? tau[u] : -1 - tau[u];           // sign carries `pair` to the store loop
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nthr;
        if (e >= total) continue;
        const bool pair = tau[u] >= 0;
        const int t = pair ? tau[u] : -1 - tau[u];
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        // pair: x[tau] + x[L - tau] and x[tau] - x[L - tau]; no partner (tau = 0, L/2) or padding: x[tau] (or 0) and 0
        *(f4*)(ce + (size_t)t * CP + c[u]) = pair ? xv[u] + xp[u] : xv[u];
        *(f4*)(co + (size_t)t * CP + c[u]) = pair ? xv[u] - xp[u] : z;
      }
    }
  } else {
    for (int e = tid; e < KTP * CP; e += nthr) {
      const int tau = e / CP, c = e - tau * CP;
      float ve = 0.f, vo = 0.f;
      if (tau < KT && c < C) {
        const bool pair = tau > 0 && 2 * tau < L;
        const float xv = xb[(size_t)tau * C + c];
        if (pair) { const float xp = xb[(size_t)(L - tau) * C + c]; ve = xv + xp; vo = xv - xp; }
        else ve = xv;
      }
      ce[e] = ve; co[e] = vo;
    }
  }
  __syncthreads();
  // ---- DFT: wave -> (bin block, channel tile)
  {
    const int i = lane & 31, h = lane >> 5;
    const int fbk = wave % nfb, ct = wave / nfb;
    const int f0 = fbk * 32;
    const int c = ct * 32 + i;
    const float* __restrict__ pc = tab + f0 + i + (size_t)h * FPAD;
    const float* __restrict__ ps = tab + (size_t)L * FPAD + f0 + i + (size_t)h * FPAD;
    const float* __restrict__ pe = ce + (size_t)h * CP + c;
    const float* __restrict__ po = co + (size_t)h * CP + c;
    const int sT = 2 * FPAD, sX = 2 * CP;
    const int nks = KTP >> 1;                          // k-steps (two folded samples each)
    f16v re = {0}, im = {0};
    // twiddle rows tau >= KT of an odd-KT last step meet zero operands; they are inside the table (KT < L for L >= 3)
    float ac[8], as[8];
    const int nblk = nks >> 3;
    if (nblk > 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { ac[k] = pc[k * sT]; as[k] = ps[k * sT]; }
      // the first block's twiddles are complete before the loop is entered: hipcc's wait-count pass otherwise carries
      // "ac / as may still be in flight" round the back edge and puts an s_waitcnt vmcnt(0) in front of the second MFMA
      // of EVERY iteration - i.e. behind the loads just issued for the next block, which undoes the pipelining
      // (MFMA phase at 55 % of the pipe's rate)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(ac[k]), "+v"(as[k]));
    }
    for (int it = 0; it < nblk; ++it) {
      float an[8], sn[8], be[8], bo[8];
      const bool more = it + 1 < nblk;
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { an[k] = pc[(8 * (it + 1) + k) * sT]; sn[k] = ps[(8 * (it + 1) + k) * sT]; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { be[k] = pe[(8 * it + k) * sX]; bo[k] = po[(8 * it + k) * sX]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[k], be[k], re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(as[k], bo[k], im, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ac[k] = an[k]; as[k] = sn[k]; }
      }
    }
    for (int ks = nblk * 8; ks < nks; ++ks) {
      const int tau = 2 * ks + h;
      const float cv = tau < KT ? pc[(size_t)ks * sT] : 0.f;
      const float sv = tau < KT ? ps[(size_t)ks * sT] : 0.f;
      re = __builtin_amdgcn_mfma_f32_32x32x2f32(cv, pe[ks * sX], re, 0, 0, 0);
      im = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, po[ks * sX], im, 0, 0, 0);
    }
    if (c < C) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int fi = (r & 3) + 8 * (r >> 2) + 4 * h;
        amp[(size_t)(f0 + fi) * CS + c] = amp2(re[r], im[r]);
      }
    }
  }
  __syncthreads();
  // ---- lower median over channels, four bins per wave pass
  const int nw = nthr >> 6;
  for (int fb = wave * 8; fb < F; fb += nw * 8) {       // rows fb .. fb + 7 (amp has FPAD >= fb + 8 rows)
    float m[8];
    wave_lower_median_rows<8>(amp + (size_t)fb * CS, CS, C, lane, m);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (fb + r < F) med[(size_t)b * F + fb + r] = m[r];
    }
  }
}

// ---------------------------------------------------------------- S1 + S2, row-resident, folded twice (L % 4 == 0)
// With H = L/2 and theta = 2 pi f tau / L:  cos(2 pi f (H - tau) / L) = (-1)^f cos(theta) and
// sin(2 pi f (H - tau) / L) = -(-1)^f sin(theta), so the half-folded sums of k_spectrum_row fold once more around
// tau = L/4, separately for even and odd bins:
//   even f:  Re = sum_{tau<=Q} ee[tau] cos,  ee = ce[tau] + ce[H - tau]      Im = sum eo[tau] sin,  eo = co[tau] - co[H - tau]
//   odd  f:  Re = sum_{tau< Q} oe[tau] cos,  oe = ce[tau] - ce[H - tau]      Im = sum oo[tau] sin,  oo = co[tau] + co[H - tau]
// (Q = L/4; at tau = Q: ee = ce[Q], oo = co[Q], eo = oe = 0 - the twiddles there vanish for that parity.)
// Half the fp32 MFMAs of k_spectrum_row - the pipe that kernel is bound by - for two more adds per sample.  A
// wave takes (parity, 32-bin block of that parity, channel tile).  Not bit-identical to the other two kernels
// (the four-term sums round differently, at the 1e-7 level); ranks of a sharded batch all take the same path.
// Channel-tiled form (amp_g != nullptr; d_model > 64, where four fold planes of the whole row no longer fit LDS):
// workgroup (b, blockIdx.y) folds and transforms channels [ctile * blockIdx.y, + ctile) only and writes its
// amplitudes to amp_g [B][F][Ctot]; the medians over all channels are then taken by k_median_rows.
__global__ __launch_bounds__(1024) void k_spectrum_rowq(const float* __restrict__ x, int B, int L, int Ctot,
                                                        const float* __restrict__ qtab, int F, int QP, int FQ,
                                                        int amp_rows, float* __restrict__ med, int ctile,
                                                        float* __restrict__ amp_g) {
  extern __shared__ __attribute__((aligned(16))) float lds_row[];
  const int H = L >> 1, Q = L >> 2;
  // (fused form: gridDim.x = B; tiled form: gridDim = (tiles, B), a row's tiles dispatched together so that what is in
  // flight at any time covers whole rows of x, i.e. every memory channel)
  const int tiled = amp_g != nullptr ? 1 : 0;
  const int c_base = tiled ? (int)blockIdx.x * ctile : 0;
  const int C = Ctot - c_base < ctile ? Ctot - c_base : ctile;      // channels of this workgroup
  const int nct = (C + 31) >> 5, CP = nct * 32, CS = C + 1;
  const size_t plane = (size_t)QP * CP;
  float* __restrict__ fold = lds_row;                  // [4][QP][CP]: ee, eo, oe, oo
  float* __restrict__ amp = lds_row + 4 * plane;       // [amp_rows][CS]
  const int b = tiled ? blockIdx.y : blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63;
  const float* __restrict__ xb = x + (size_t)b * L * Ctot + c_base;
  {
    const bool vec = (Ctot & 3) == 0 && (((uintptr_t)x) & 15) == 0;
    const int cw = vec ? 4 : 1, cn = CP / cw, total = QP * cn;
    for (int e0 = tid; e0 < total; e0 += 2 * nthr) {
      f4 x0[2], x1[2], x2[2], x3[2];                   // x[tau], x[L - tau], x[H - tau], x[H + tau]
      int tau[2], c[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * nthr;
        tau[u] = e / cn; c[u] = (e - tau[u] * cn) * cw;
        const bool ok = e < total && tau[u] <= Q && c[u] < C;
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        x0[u] = x1[u] = x2[u] = x3[u] = z;
        if (ok) {
          const int t = tau[u];
          if (vec) {
            x0[u] = *(const f4*)(xb + (size_t)t * Ctot + c[u]);
            if (t > 0) x1[u] = *(const f4*)(xb + (size_t)(L - t) * Ctot + c[u]);
            if (t < Q) { x2[u] = *(const f4*)(xb + (size_t)(H - t) * Ctot + c[u]); if (t > 0) x3[u] = *(const f4*)(xb + (size_t)(H + t) * Ctot + c[u]); }
          } else {
            x0[u].x = xb[(size_t)t * Ctot + c[u]];
            if (t > 0) x1[u].x = xb[(size_t)(L - t) * Ctot + c[u]];
            if (t < Q) { x2[u].x = xb[(size_t)(H - t) * Ctot + c[u]]; if (t > 0) x3[u].x = xb[(size_t)(H + t) * Ctot + c[u]]; }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * nthr;
        if (e >= total) continue;
        const int t = tau[u];
        // ce[t] = x[t] + x[L-t] (t = 0: x[0]),  co[t] = x[t] - x[L-t] (t = 0: 0);  partner H - t: x[H-t] +- x[H+t]
        // (t = 0: ce[H] = x[H], co[H] = 0; t = Q has no partner)
        const f4 ce = x0[u] + x1[u];
        const f4 co = t > 0 ? x0[u] - x1[u] : f4{0.f, 0.f, 0.f, 0.f};
        const f4 pe = x2[u] + x3[u];
        const f4 po = (t > 0 && t < Q) ? x2[u] - x3[u] : f4{0.f, 0.f, 0.f, 0.f};
        f4 ee = ce + pe, eo = co - po, oe = ce - pe, oo = co + po;
        if (t >= Q) { eo = f4{0.f, 0.f, 0.f, 0.f}; oe = f4{0.f, 0.f, 0.f, 0.f}; }
        if (t > Q) { ee = f4{0.f, 0.f, 0.f, 0.f}; oo = f4{0.f, 0.f, 0.f, 0.f}; }
        float* dst = fold + (size_t)t * CP + c[u];
        if (vec) {
          *(f4*)(dst) = ee; *(f4*)(dst + plane) = eo; *(f4*)(dst + 2 * plane) = oe; *(f4*)(dst + 3 * plane) = oo;
        } else {
          dst[0] = ee.x; dst[plane] = eo.x; dst[2 * plane] = oe.x; dst[3 * plane] = oo.x;
        }
      }
    }
  }
  __syncthreads();
  {
    const int i = lane & 31, h = lane >> 5;
    const int nfq = FQ >> 5;                           // 32-bin blocks per parity
    const int blk = wave % (2 * nfq), ct = wave / (2 * nfq);
    const int odd = blk >= nfq ? 1 : 0, m0 = (blk - odd * nfq) * 32;
    const int c = ct * 32 + i;
    const size_t tplane = (size_t)QP * FQ;
    const float* __restrict__ pc = qtab + (size_t)(2 * odd) * tplane + (size_t)h * FQ + m0 + i;
    const float* __restrict__ ps = pc + tplane;
    const float* __restrict__ pe = fold + (size_t)(2 * odd) * plane + (size_t)h * CP + c;   // ee | oe
    const float* __restrict__ po = fold + (size_t)(odd ? 3 : 1) * plane + (size_t)h * CP + c;   // eo | oo
    const int sT = 2 * FQ, sX = 2 * CP;
    const int nks = QP >> 1;
    f16v re = {0}, im = {0};
    float ac[8], as[8];
    const int nblk = nks >> 3;
    if (nblk > 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { ac[k] = pc[k * sT]; as[k] = ps[k * sT]; }
      // the first block's twiddles are complete before the loop is entered: hipcc's wait-count pass otherwise carries
      // "ac / as may still be in flight" round the back edge and puts an s_waitcnt vmcnt(0) in front of the second MFMA
      // of EVERY iteration - i.e. behind the loads just issued for the next block, which undoes the pipelining
      // (MFMA phase at 55 % of the pipe's rate)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(ac[k]), "+v"(as[k]));
    }
    for (int it = 0; it < nblk; ++it) {
      float an[8], sn[8], be[8], bo[8];
      const bool more = it + 1 < nblk;
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { an[k] = pc[(8 * (it + 1) + k) * sT]; sn[k] = ps[(8 * (it + 1) + k) * sT]; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { be[k] = pe[(8 * it + k) * sX]; bo[k] = po[(8 * it + k) * sX]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[k], be[k], re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(as[k], bo[k], im, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ac[k] = an[k]; as[k] = sn[k]; }
      }
    }
    for (int ks = nblk * 8; ks < nks; ++ks) {
      re = __builtin_amdgcn_mfma_f32_32x32x2f32(pc[(size_t)ks * sT], pe[ks * sX], re, 0, 0, 0);
      im = __builtin_amdgcn_mfma_f32_32x32x2f32(ps[(size_t)ks * sT], po[ks * sX], im, 0, 0, 0);
    }
    if (c < C && amp_g != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 2 * (m0 + (r & 3) + 8 * (r >> 2) + 4 * h) + odd;
        if (f < F) amp_g[((size_t)b * F + f) * Ctot + c_base + c] = amp2(re[r], im[r]);
      }
    } else if (c < C) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int fi = (r & 3) + 8 * (r >> 2) + 4 * h;
        amp[(size_t)(2 * (m0 + fi) + odd) * CS + c] = amp2(re[r], im[r]);
      }
    }
  }
  if (amp_g != nullptr) return;
  __syncthreads();
  const int nw = nthr >> 6;
  for (int fb = wave * 8; fb < F; fb += nw * 8) {       // rows fb .. fb + 7 < amp_rows
    float m[8];
    wave_lower_median_rows<8>(amp + (size_t)fb * CS, CS, C, lane, m);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (fb + r < F) med[(size_t)b * F + fb + r] = m[r];
    }
  }
}

// bitonic_step_med3 with the direction flipped (the upper half of a 128-element network's k = 64 stage)
template <int K, int J, int NR>
__device__ __forceinline__ void bitonic_step_med3_desc(float (&v)[NR], int lane) {
  const bool keepmin = !(((lane & K) == 0) == ((lane & J) == 0));
  const float sel = keepmin ? -INFINITY : INFINITY;
  float o[NR];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (J < 16) asm volatile("s_nop 1");
#pragma unroll
  for (int r = 0; r < NR; ++r) o[r] = lane_xor_raw<J>(v[r]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < NR; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], o[r], sel);
}

// Lower median of NR rows of 64 < C <= 128 values: the 28-step bitonic network on two registers per lane and row
// (elements lane and 64 + lane), every lane exchange a raw DPP move + one v_med3 as in wave_lower_median_rows.
template <int NR>
__device__ __forceinline__ void wave_lower_median128_rows(const float* __restrict__ base, size_t stride, int C, int lane,
                                                          float (&m)[NR]) {
  float a[NR], b[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    a[r] = base[r * stride + lane];                               // C > 64
    b[r] = 64 + lane < C ? base[r * stride + 64 + lane] : INFINITY;
  }
#define FTN_BOTH(K, J) bitonic_step_med3<K, J, NR>(a, lane); bitonic_step_med3<K, J, NR>(b, lane)
  FTN_BOTH(2, 1);
  FTN_BOTH(4, 2); FTN_BOTH(4, 1);
  FTN_BOTH(8, 4); FTN_BOTH(8, 2); FTN_BOTH(8, 1);
  FTN_BOTH(16, 8); FTN_BOTH(16, 4); FTN_BOTH(16, 2); FTN_BOTH(16, 1);
  FTN_BOTH(32, 16); FTN_BOTH(32, 8); FTN_BOTH(32, 4); FTN_BOTH(32, 2); FTN_BOTH(32, 1);
  // k = 64: elements 0..63 ascending, 64..127 descending
#define FTN_UPDOWN(J) bitonic_step_med3<64, J, NR>(a, lane); bitonic_step_med3_desc<64, J, NR>(b, lane)
  FTN_UPDOWN(32); FTN_UPDOWN(16); FTN_UPDOWN(8); FTN_UPDOWN(4); FTN_UPDOWN(2); FTN_UPDOWN(1);
#undef FTN_UPDOWN
  // k = 128, ascending everywhere: distance 64 is the register pair, then the lane distances
#pragma unroll
  for (int r = 0; r < NR; ++r) { const float lo = fminf(a[r], b[r]), hi = fmaxf(a[r], b[r]); a[r] = lo; b[r] = hi; }
  FTN_BOTH(64, 32); FTN_BOTH(64, 16); FTN_BOTH(64, 8); FTN_BOTH(64, 4); FTN_BOTH(64, 2); FTN_BOTH(64, 1);
#undef FTN_BOTH
  const int t = (C - 1) >> 1;
#pragma unroll
  for (int r = 0; r < NR; ++r) m[r] = t < 64 ? __shfl(a[r], t) : __shfl(b[r], t - 64);
}

// Lower median over the channels of amp_g [rows][C], 64 < C <= 128 (the channel-tiled k_spectrum_rowq): four rows
// per wave.
__global__ __launch_bounds__(256) void k_median_rows(const float* __restrict__ amp_g, long long rows, int C,
                                                     float* __restrict__ med) {
  const int lane = threadIdx.x & 63;
  const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (r0 >= rows) return;
  // a ragged last wave re-reads the last row (never written twice: the store below is guarded)
  const long long last = rows - 1;
  const float* __restrict__ base = amp_g + (size_t)r0 * C;
  float m[4];
  if (r0 + 3 <= last) wave_lower_median128_rows<4>(base, (size_t)C, C, lane, m);
  else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float one[1];
      const long long rr = r0 + r <= last ? r0 + r : last;
      wave_lower_median128_rows<1>(amp_g + (size_t)rr * C, (size_t)C, C, lane, one);
      m[r] = one[0];
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r0 + r <= last) med[r0 + r] = m[r];
  }
}

// psum[f] = sum_b med[b][f] in fp64, fixed order: 32 row-strided partial sums per column, combined in index
// order (bitwise reproducible; no atomics).
// With an exchange (XchArgs.world > 0) block i also stores its 32 columns into slot `rank` of every rank's exchange
// buffer (peer device memory, mapped through hipIpcOpenMemHandle) and then turns that slot's sequence word i: plain
// stores, a system-scope fence, a system-scope store of the word - the consumer is ftn_finalize.h's bounded wait.
struct XchArgs {
  char* half[FTN_XCHG_MAXWORLD];     // this call's half of every rank's buffer
  int world, rank, F_cap;
  unsigned long long seq;
};

__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ med, int B, int F,
                                                 double* __restrict__ psum, XchArgs xa) {
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
    for (int r = 0; r < xa.world; ++r) ((double*)xa.half[r])[(size_t)xa.rank * xa.F_cap + f] = t;
  }
  if (xa.world > 0 && bl == 0) {                                // wave 0 of the block holds all 32 storing lanes
    __threadfence_system();
    if (fl == 0)
      for (int r = 0; r < xa.world; ++r)
        __hip_atomic_store((unsigned long long*)(xa.half[r] + ftn_xchg_flags_off(xa.world, xa.F_cap)) +
                               (size_t)xa.rank * FTN_XCHG_NBLK + blockIdx.x,
                           xa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

extern "C" size_t ftn_exchange_bytes(int world, int F_cap) {
  if (world < 1 || world > FTN_XCHG_MAXWORLD || F_cap < 2 || F_cap > 32 * FTN_XCHG_NBLK) return 0;
  return 2 * ftn_xchg_half_bytes(world, F_cap) + 256;           // two halves + the error word's line
}

static bool xch_ok(const FtnExchange* x, int F) {
  return x->world >= 1 && x->world <= FTN_XCHG_MAXWORLD && x->rank >= 0 && x->rank < x->world && x->seq > 0 &&
         F <= x->F_cap && x->F_cap <= 32 * FTN_XCHG_NBLK && x->slots[x->rank] != nullptr;
}
static bool xch_mapped(const FtnExchange* x) {
  for (int r = 0; r < x->world; ++r)
    if (x->slots[r] == nullptr) return false;
  return true;
}
static char* xch_half(const FtnExchange* x, int r) {
  return (char*)x->slots[r] + (size_t)(x->seq & 1) * ftn_xchg_half_bytes(x->world, x->F_cap);
}
int* ftn_xch_err_word(const FtnExchange* x) {
  return (int*)((char*)x->slots[x->rank] + 2 * ftn_xchg_half_bytes(x->world, x->F_cap));
}
// the finalize side of an exchange: psum rows = the world slots of this rank's own buffer
void ftn_xch_fill(const FtnExchange* x, int F, FinalizeArgs* fa) {
  char* mine = xch_half(x, x->rank);
  fa->psum = (const double*)mine;
  fa->nparts = x->world;
  fa->psum_stride = x->F_cap;
  fa->ready = (const unsigned long long*)(mine + ftn_xchg_flags_off(x->world, x->F_cap));
  fa->ready_seq = x->seq;
  fa->ready_n = (F + 31) / 32;
  fa->xerr = ftn_xch_err_word(x);
}

// One rank's exchange buffer: allocated and zeroed here (hipMalloc: its own allocation, which is what an IPC handle
// names), exported as a 64-byte handle the other ranks open.
extern "C" int ftn_exchange_alloc(int world, int F_cap, void** buf_out, void* handle64_out) {
  const size_t n = ftn_exchange_bytes(world, F_cap);
  FTN_CHECK_ARG(n > 0 && buf_out && handle64_out, "ftn_exchange_alloc: world=%d F_cap=%d", world, F_cap);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  void* p = nullptr;
  // uncached device memory where the runtime offers it (what RCCL uses for words that GPUs exchange inside running
  // kernels): every access goes to memory, whichever GPU issues it; plain hipMalloc otherwise (the kernels use
  // system-scope loads / stores either way)
  hipError_t e = hipExtMallocWithFlags(&p, n, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; e = hipMalloc(&p, n); }
  if (e == hipSuccess) e = hipMemset(p, 0, n);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64_out, p);
  if (e != hipSuccess) {
    ftn_set_error("ftn_exchange_alloc: %s", hipGetErrorString(e));
    if (p) (void)hipFree(p);
    return (int)e;
  }
  *buf_out = p;
  return 0;
}
extern "C" int ftn_exchange_open(const void* handle64, void** mapped_out) {
  FTN_CHECK_ARG(handle64 && mapped_out, "ftn_exchange_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(mapped_out, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { ftn_set_error("hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}
extern "C" int ftn_exchange_close(void* mapped) {
  hipError_t e = mapped ? hipIpcCloseMemHandle(mapped) : hipSuccess;
  if (e != hipSuccess) { ftn_set_error("hipIpcCloseMemHandle: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}
extern "C" int ftn_exchange_free(void* buf) {
  hipError_t e = buf ? hipFree(buf) : hipSuccess;
  if (e != hipSuccess) { ftn_set_error("hipFree: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

extern "C" int ftn_exchange_error(const FtnExchange* xch, void* stream) {
  FTN_CHECK_ARG(xch && xch->world >= 1 && xch->world <= FTN_XCHG_MAXWORLD && xch->rank >= 0 && xch->rank < xch->world &&
                xch->slots[xch->rank], "ftn_exchange_error: bad exchange");
  int v = 0;
  hipError_t e = hipMemcpyAsync(&v, ftn_xch_err_word(xch), sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) { ftn_set_error("ftn_exchange_error: %s", hipGetErrorString(e)); return -(int)e - 1000; }
  return v;
}

// channel-tiled k_spectrum_rowq: 64 < C <= 128 (k_median_rows' network), the four fold planes of a 32-channel tile fit
// LDS, <= 16 waves; batch rows ride on gridDim.y
static bool qtile_fits(int L, int C) {
  if (!qfold_ok(L) || C <= 64 || C > 128) return false;
  const int QP = qfold_qp(L), FQ = qfold_fq(L);
  return 2 * (FQ / 32) <= 16 && (size_t)4 * QP * 32 * sizeof(float) <= 160 * 1024;
}

extern "C" size_t ftn_period_spectrum_scratch_bytes(int B, int L, int C) {
  if (B < 1 || B > 65535 || L < 2 || C < 1 || !qtile_fits(L, C)) return 0;
  return (size_t)B * (L / 2 + 1) * C * sizeof(float);
}

extern "C" int ftn_period_spectrum(const float* x_dev, int B, int L, int C, const void* table_dev,
                                   float* med_dev, double* psum_dev, void* stream, const FtnExchange* xch,
                                   void* scratch_dev) {
  FTN_CHECK_ARG(x_dev && table_dev && med_dev && psum_dev, "ftn_period_spectrum: null pointer");
  FTN_CHECK_ARG(xch == nullptr || xch_ok(xch, L / 2 + 1), "ftn_period_spectrum: bad exchange (world / rank / seq / F_cap)");
  FTN_CHECK_ARG(xch == nullptr || xch_mapped(xch), "ftn_period_spectrum: an exchange slot is not mapped");
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
  // row-resident form where the folded row and its amplitude tile fit LDS and there are rows enough to fill the chip
  // (FTN_SEL_ROW=1 / 0 forces / forbids it: the two kernels are bit-identical, tests compare them)
  const int KT = L / 2 + 1, KTP = (KT + 1) & ~1, nct = (C + 31) / 32;
  const size_t lds_row = (size_t)KTP * nct * 32 * 8 + (size_t)FPAD * (C + 1) * sizeof(float);
  // FTN_SEL_ROW: 0 = k_spectrum, 1 = k_spectrum_row, 2 = k_spectrum_rowq, unset = fastest form that fits
  static const int row_mode = [] { const char* e = getenv("FTN_SEL_ROW"); return e == nullptr ? -1 : atoi(e); }();
  const bool row_fits = C <= 64 && nfb * nct <= 16 && lds_row <= 160 * 1024 && L >= 3;
  const int QP = qfold_qp(L), FQ = qfold_fq(L);
  const int amp_rows = ((2 * FQ > FPAD ? 2 * FQ : FPAD) + 7) & ~7;
  const size_t lds_q = (size_t)4 * QP * nct * 32 * sizeof(float) + (size_t)amp_rows * (C + 1) * sizeof(float);
  const bool q_fits = qfold_ok(L) && C <= 64 && 2 * (FQ / 32) * nct <= 16 && lds_q <= 160 * 1024;
  if (scratch_dev != nullptr && B <= 65535 && qtile_fits(L, C) && row_mode != 0 && row_mode != 1) {
    // d_model > 64: (row, 32-channel tile) workgroups, amplitudes through the caller's scratch, medians in a second launch
    const size_t lds_t = (size_t)4 * QP * 32 * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)k_spectrum_rowq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
    if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_spectrum_rowq): %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(k_spectrum_rowq, dim3((unsigned)nct, (unsigned)B), dim3(64 * 2 * (FQ / 32)), lds_t, (hipStream_t)stream, x_dev,
                       B, L, C, (const float*)table_dev + (size_t)2 * L * FPAD, F, QP, FQ, 0, med_dev, 32, (float*)scratch_dev);
    FTN_CHECK_LAUNCH();
    const long long rows = (long long)B * F;
    hipLaunchKernelGGL(k_median_rows, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scratch_dev, rows, C, med_dev);
  } else if (q_fits && (row_mode == 2 || (row_mode < 0 && B >= 64))) {
    hipError_t e = hipFuncSetAttribute((const void*)k_spectrum_rowq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_spectrum_rowq): %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(k_spectrum_rowq, dim3((unsigned)B), dim3(64 * 2 * (FQ / 32) * nct), lds_q, (hipStream_t)stream, x_dev, B, L,
                       C, (const float*)table_dev + (size_t)2 * L * FPAD, F, QP, FQ, amp_rows, med_dev, C, (float*)nullptr);
  } else if (row_fits && (row_mode == 1 || (row_mode < 0 && B >= 64))) {
    hipError_t e = hipFuncSetAttribute((const void*)k_spectrum_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row);
    if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_spectrum_row): %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(k_spectrum_row, dim3((unsigned)B), dim3(64 * nfb * nct), lds_row, (hipStream_t)stream, x_dev, B, L, C,
                       (const float*)table_dev, F, FPAD, med_dev);
  } else {
    hipLaunchKernelGGL(k_spectrum, dim3((unsigned)(ftn_cdiv(B, 8) * 8 * nfb)), dim3(64 * nw), lds, (hipStream_t)stream, x_dev,
                       B, L, C, (const float*)table_dev, F, FPAD, med_dev, getenv("FTN_SEL_FLAT") != nullptr ? 1 : 0);
  }
  FTN_CHECK_LAUNCH();
  XchArgs xa = {};
  if (xch != nullptr) {
    for (int r = 0; r < xch->world; ++r) xa.half[r] = xch_half(xch, r);
    xa.world = xch->world; xa.rank = xch->rank; xa.F_cap = xch->F_cap; xa.seq = xch->seq;
  }
  hipLaunchKernelGGL(k_colsum, dim3(ftn_cdiv(F, 32)), dim3(1024), 0, (hipStream_t)stream, med_dev, B, F, psum_dev, xa);
  FTN_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- S3 - S5
__global__ __launch_bounds__(256) void k_finalize(FinalizeArgs fa) { finalize_body(fa); }

extern "C" int ftn_period_finalize(const double* psum_dev, int nparts, int Btotal, const float* med_dev, int B,
                                   int L, int k_periods, int pmax, int min_period_threshold, int act_dtype,
                                   int max_unique, double log_base, FtnDesc* desc_dev, float* amps_dev,
                                   float* weights_dev, void* stream, const FtnExchange* xch) {
  FTN_CHECK_ARG((psum_dev || xch) && med_dev && desc_dev && amps_dev && weights_dev, "ftn_period_finalize: null pointer");
  FTN_CHECK_ARG(xch == nullptr || xch_ok(xch, L / 2 + 1), "ftn_period_finalize: bad exchange (world / rank / seq / F_cap)");
  if (xch != nullptr) nparts = xch->world;
  FTN_CHECK_ARG((((uintptr_t)amps_dev | (uintptr_t)weights_dev) & 15) == 0, "ftn_period_finalize: amps / weights must be 16-byte aligned");
  FTN_CHECK_ARG(B >= 1 && L >= 2 && nparts >= 1 && Btotal >= B, "ftn_period_finalize: bad shape");
  FTN_CHECK_ARG(k_periods <= FTN_KMAX, "ftn_period_finalize: k_periods=%d > FTN_KMAX=%d", k_periods, FTN_KMAX);
  FTN_CHECK_ARG(act_dtype >= 0 && act_dtype <= 2, "ftn_period_finalize: act_dtype=%d", act_dtype);
  // ctor clamps of FFTPeriodSelector (:59-62)
  if (k_periods < 0) k_periods = 0;
  if (pmax < 1) pmax = 1;
  if (min_period_threshold < 1) min_period_threshold = 1;
  if (min_period_threshold > pmax) min_period_threshold = pmax;
  const int F = L / 2 + 1;
  const size_t lds = ftn_finalize_lds_bytes(F);
  FTN_CHECK_ARG(lds <= 48 * 1024, "ftn_period_finalize: L=%d too long", L);
  FinalizeArgs fa = {psum_dev, nparts, Btotal, med_dev, B, L, F, k_periods, pmax, min_period_threshold, desc_dev,
                     amps_dev, weights_dev, act_dtype, max_unique > 0 ? max_unique : 0, log_base > 1.0 ? (float)log(log_base) : 0.f};
  if (xch != nullptr) ftn_xch_fill(xch, F, &fa);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), lds, (hipStream_t)stream, fa);
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

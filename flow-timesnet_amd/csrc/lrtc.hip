// LowRankTemporalContext on gfx950 (reference models/timesnet.py:1340-1371).
//
// out[b][l][n] = (x?) + scale * sum_r basisC[l][r] * coeff[b][n][r]
//
// The time-mean subtraction of the reference (:1369) is linear, so it is folded
// into the basis (basisC = basis - mean_l basis; the reference's columns are
// already centred to ~1e-9, :1347).  The kernel is an HBM write stream
// (4*B*L*N bytes, + 4*B*L*N read when fused with x): each lane owns 4
// consecutive series n (one 16-byte store per time step, 1 KiB per wave), keeps
// its 4xR coefficients in registers and walks a slab of time steps; the basis
// row is wave-uniform (scalar loads).  R*8 FLOP per 16 stored bytes keeps the
// VALU far below the HBM roof, so no MFMA here.
#include <math.h>
#include "ftn_common.h"

// basis as the reference builds it: fp32 angle (pi/L rounded to fp32, then two
// fp32 products, :1346), cos, column mean removed, unit L2 norm (:1347-1350).
__global__ __launch_bounds__(256) void k_lrtc_basis(float* __restrict__ basis, int L, int R, float pi_over_L) {
  __shared__ double red[256];
  const int r = blockIdx.x;  // column r -> frequency r+1
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int l = tid; l < L; l += 256) {
    const float t1 = pi_over_L * ((float)l + 0.5f);
    const float ang = t1 * (float)(r + 1);
    s += (double)(float)cos((double)ang);
  }
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const float mean = (float)(red[0] / (double)L);
  __syncthreads();
  double ss = 0.0;
  for (int l = tid; l < L; l += 256) {
    const float t1 = pi_over_L * ((float)l + 0.5f);
    const float ang = t1 * (float)(r + 1);
    const float v = (float)cos((double)ang) - mean;
    ss += (double)v * (double)v;
  }
  red[tid] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  float nrm = (float)sqrt(red[0]);
  if (nrm < 1.1920929e-07f) nrm = 1.1920929e-07f;  // clamp_min(eps) :1349-1350
  __syncthreads();
  double sm = 0.0;
  for (int l = tid; l < L; l += 256) {
    const float t1 = pi_over_L * ((float)l + 0.5f);
    const float ang = t1 * (float)(r + 1);
    const float v = ((float)cos((double)ang) - mean) / nrm;
    basis[(size_t)l * R + r] = v;
    sm += (double)v;
  }
  red[tid] = sm;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  // residual column mean (~1e-9): what the reference's second centring (:1369) removes
  if (tid == 0) basis[(size_t)L * R + r] = (float)(red[0] / (double)L);
}

extern "C" int ftn_lrtc_basis(float* basis_dev, int L, int R, void* stream) {
  FTN_CHECK_ARG(basis_dev && L >= 1 && R >= 1, "ftn_lrtc_basis: bad argument L=%d R=%d", L, R);
  const float c = (float)(3.14159265358979323846 / (double)L);
  hipLaunchKernelGGL(k_lrtc_basis, dim3(R), dim3(256), 0, (hipStream_t)stream, basis_dev, L, R, c);
  FTN_CHECK_LAUNCH();
  return 0;
}

#define LRTC_LT 48  // time steps per lane-group slab (coefficients are re-read once per slab)

template <int RT, bool VEC, bool ADDX>
__global__ __launch_bounds__(256) void k_lrtc(const float* __restrict__ coeff, const float* __restrict__ basis,
                                              const float* __restrict__ scale_p, const float* __restrict__ x,
                                              float* __restrict__ out, int L, int N, int R, int nqb) {
  // nqb lanes (multiple of 64) span the series quads of this block; the remaining 256/nqb
  // lane groups take further time slabs, so narrow N still fills the workgroup.  The basis
  // rows of the block's slabs are staged once in LDS (broadcast reads), padded to RT columns.
  __shared__ float bs[(256 / 64) * LRTC_LT * RT];
  const int nsl = 256 / nqb;
  const int b = blockIdx.z;
  const int nq = threadIdx.x % nqb, sl = threadIdx.x / nqb;
  const int n0 = (blockIdx.x * nqb + nq) * 4;
  const int lblk = blockIdx.y * nsl * LRTC_LT;
  for (int i = threadIdx.x; i < nsl * LRTC_LT * RT; i += 256) {
    const int li = i / RT, r = i - li * RT;
    const int l = lblk + li;
    bs[i] = (l < L && r < R) ? basis[(size_t)l * R + r] : 0.f;
  }
  __syncthreads();
  const int l0 = lblk + sl * LRTC_LT;
  if (n0 >= N || l0 >= L) return;
  const float scale = *scale_p;
  float co[4][RT];
  if (RT % 4 == 0 && R == RT && n0 + 3 < N) {
    // the four series of this lane are 4*R contiguous floats: RT unguarded 16-byte loads in flight at once
    // (the guarded scalar form below compiles to one load + branch + wait per coefficient, i.e. 4*RT
    // serialised memory round trips at the head of every wave)
    const f4* __restrict__ cp = (const f4*)(coeff + ((size_t)b * N + n0) * R);
    f4 cv[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) cv[i] = cp[i];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < RT; ++r) co[k][r] = scale * cv[(k * RT + r) >> 2][(k * RT + r) & 3];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < RT; ++r)
        co[k][r] = (n0 + k < N && r < R) ? scale * coeff[((size_t)b * N + n0 + k) * R + r] : 0.f;
  }
  // mean over l of sum_r basis[l][r]*co[r] == sum_r cmean[r]*co[r]; cmean sits behind the basis
  const float* __restrict__ cmean = basis + (size_t)L * R;
  float mu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const float cm = r < R ? cmean[r] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) mu[k] = fmaf(cm, co[k][r], mu[k]);
  }
  const int l1 = min(l0 + LRTC_LT, L);
  const float* __restrict__ brow = bs + sl * LRTC_LT * RT;
#pragma unroll 2
  for (int l = l0; l < l1; ++l, brow += RT) {
    float acc[4] = {-mu[0], -mu[1], -mu[2], -mu[3]};
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const float bv = brow[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = fmaf(bv, co[k][r], acc[k]);
    }
    const size_t o = ((size_t)b * L + l) * N + n0;
    if (VEC) {
      f4 v = {acc[0], acc[1], acc[2], acc[3]};
      if (ADDX) v += __builtin_nontemporal_load((const f4*)(x + o));
      __builtin_nontemporal_store(v, (f4*)(out + o));                 // written once, never re-read here
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (n0 + k < N) out[o + k] = (ADDX ? x[o + k] : 0.f) + acc[k];
    }
  }
}

template <int RT>
static void launch_lrtc(dim3 grid, hipStream_t st, bool vec, bool addx, const float* coeff, const float* basis,
                        const float* scale, const float* x, float* out, int L, int N, int R, int nqb) {
  if (vec && addx) hipLaunchKernelGGL((k_lrtc<RT, true, true>), grid, dim3(256), 0, st, coeff, basis, scale, x, out, L, N, R, nqb);
  else if (vec) hipLaunchKernelGGL((k_lrtc<RT, true, false>), grid, dim3(256), 0, st, coeff, basis, scale, x, out, L, N, R, nqb);
  else if (addx) hipLaunchKernelGGL((k_lrtc<RT, false, true>), grid, dim3(256), 0, st, coeff, basis, scale, x, out, L, N, R, nqb);
  else hipLaunchKernelGGL((k_lrtc<RT, false, false>), grid, dim3(256), 0, st, coeff, basis, scale, x, out, L, N, R, nqb);
}

// basis_dev is what ftn_lrtc_basis wrote: [L][R] basis followed by R column means.
extern "C" int ftn_lrtc_forward(const float* coeff_dev, const float* basis_dev, const float* scale_dev,
                                const float* x_dev_or_null, float* out_dev, int B, int L, int N, int R,
                                void* stream) {
  FTN_CHECK_ARG(coeff_dev && basis_dev && scale_dev && out_dev, "ftn_lrtc_forward: null pointer");
  FTN_CHECK_ARG(B >= 1 && L >= 1 && N >= 1 && R >= 1, "ftn_lrtc_forward: bad shape B=%d L=%d N=%d R=%d", B, L, N, R);
  FTN_CHECK_ARG(R <= 32, "ftn_lrtc_forward: rank %d > 32 not supported", R);
  FTN_CHECK_ARG(B <= 65535 && ftn_cdiv(L, LRTC_LT) <= 65535, "ftn_lrtc_forward: grid too large");
  hipStream_t st = (hipStream_t)stream;
  const bool addx = x_dev_or_null != nullptr;
  const bool vec = (N % 4 == 0) && (((uintptr_t)out_dev & 15) == 0) && (!addx || ((uintptr_t)x_dev_or_null & 15) == 0);
  int nqb = ((ftn_cdiv(N, 4) + 63) / 64) * 64;
  if (nqb > 256) nqb = 256;
  if (nqb == 192) nqb = 256;   // 256 / nqb must be integral
  dim3 grid(ftn_cdiv(N, 4 * nqb), ftn_cdiv(L, LRTC_LT * (256 / nqb)), B);
  if (R <= 4) launch_lrtc<4>(grid, st, vec, addx, coeff_dev, basis_dev, scale_dev, x_dev_or_null, out_dev, L, N, R, nqb);
  else if (R <= 8) launch_lrtc<8>(grid, st, vec, addx, coeff_dev, basis_dev, scale_dev, x_dev_or_null, out_dev, L, N, R, nqb);
  else if (R <= 16) launch_lrtc<16>(grid, st, vec, addx, coeff_dev, basis_dev, scale_dev, x_dev_or_null, out_dev, L, N, R, nqb);
  else launch_lrtc<32>(grid, st, vec, addx, coeff_dev, basis_dev, scale_dev, x_dev_or_null, out_dev, L, N, R, nqb);
  FTN_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t ftn_lrtc_basis_floats(int L, int R) {
  if (L < 1 || R < 1) return 0;
  return (size_t)(L + 1) * R;
}

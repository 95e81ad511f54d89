// Shared host/device helpers for libflowtimes_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flowtimes.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

void ftn_set_error(const char* fmt, ...);

#define FTN_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      ftn_set_error(__VA_ARGS__);       \
      return -1;                        \
    }                                   \
  } while (0)

#define FTN_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) {                                  \
      ftn_set_error("%s:%d: %s", __FILE__, __LINE__,        \
                    hipGetErrorString(e_));                  \
      return (int)e_;                                        \
    }                                                        \
  } while (0)

static inline int ftn_pad16(int v) { return (v + 15) & ~15; }
static inline int ftn_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- conv tiling: one rule shared by the device finalize kernel and the host --
// A conv tile is th x tw grid pixels (<= FTN_TILE_PX = 22 units of 16 px);
// the kernel stages the tile plus its halo, CLIPPED to the grid, in LDS.  Tiles
// are as large as possible (a whole 336-pixel grid is one tile) subject to the
// clipped region (for a 7x7 kernel) staying <= FTN_REGION_PX pixels.
#define FTN_TILE_PX 352
#define FTN_REGION_PX 352
#define FTN_TILE_HALO 3

__host__ __device__ inline void ftn_tile_geometry(int cycles, int period, int* tw, int* th,
                                                  int* ntx, int* nty) {
  int nx = 1, w = period, h = 1, ny = cycles;
  for (;; ++nx) {
    w = (period + nx - 1) / nx;
    h = FTN_TILE_PX / w;
    if (h < 1) continue;                       // tile row wider than a tile: split columns further
    if (h > cycles) h = cycles;
    ny = (cycles + h - 1) / h;
    h = (cycles + ny - 1) / ny;
    // shrink h until the clipped staging region fits
    for (;;) {
      int rw = w + 2 * FTN_TILE_HALO; if (rw > period) rw = period;
      int rh = h + 2 * FTN_TILE_HALO; if (rh > cycles) rh = cycles;
      if (rw * rh <= FTN_REGION_PX || h == 1) break;
      --h;
    }
    ny = (cycles + h - 1) / h;
    {
      int rw = w + 2 * FTN_TILE_HALO; if (rw > period) rw = period;
      int rh = h + 2 * FTN_TILE_HALO; if (rh > cycles) rh = cycles;
      if (rw * rh <= FTN_REGION_PX || w <= 8) break;
    }
  }
  *tw = w; *th = h; *ntx = nx; *nty = ny;
}

// Groups distinct valid periods ascending, fills mapping/pad/cycles/offsets/tiles.
// `periods[K]` are the selector's kept candidates (score order, duplicates allowed).
// Mirrors PeriodGrouper.group with env flags unset (reference :513-557).
__host__ __device__ inline void ftn_build_groups(const int* periods, int K, int L, int min_period,
                                                 int max_period, FtnDesc* d) {
  int G = 0;
  for (int j = 0; j < FTN_KMAX; ++j) d->sel_group[j] = -1;
  for (int j = 0; j < K; ++j) {
    int p = periods[j];
    if (p <= 0 || p < min_period || p > max_period) continue;
    int pad = (p - (L % p)) % p;
    int cyc = (L + pad) / p;
    if (cyc < 2) continue;
    int g = 0;
    while (g < G && d->g_period[g] != p) ++g;
    if (g == G) { d->g_period[G] = p; ++G; }
  }
  // ascending insertion sort of the distinct periods
  for (int a = 1; a < G; ++a) {
    int v = d->g_period[a], b = a - 1;
    while (b >= 0 && d->g_period[b] > v) { d->g_period[b + 1] = d->g_period[b]; --b; }
    d->g_period[b + 1] = v;
  }
  int px = 0, tiles = 0;
  for (int g = 0; g < G; ++g) {
    int p = d->g_period[g];
    int pad = (p - (L % p)) % p;
    d->g_pad[g] = pad;
    d->g_cycles[g] = (L + pad) / p;
    d->g_px_off[g] = px;
    px += L + pad;
    ftn_tile_geometry(d->g_cycles[g], p, &d->g_tw[g], &d->g_th[g], &d->g_ntx[g], &d->g_nty[g]);
    d->g_tile_off[g] = tiles;
    tiles += d->g_ntx[g] * d->g_nty[g];
  }
  d->g_px_off[G] = px;
  d->g_tile_off[G] = tiles;
  for (int g = G; g < FTN_KMAX; ++g) {
    d->g_period[g] = 0; d->g_pad[g] = 0; d->g_cycles[g] = 0;
    d->g_tw[g] = 0; d->g_th[g] = 0; d->g_ntx[g] = 0; d->g_nty[g] = 0;
    if (g > G) { d->g_px_off[g] = px; d->g_tile_off[g] = tiles; }
  }
  d->g_px_off[FTN_KMAX] = px;          // every word of the descriptor is defined (it is compared / hashed bytewise)
  d->g_tile_off[FTN_KMAX] = tiles;
  for (int j = 0; j < K; ++j) {
    int p = periods[j];
    if (p <= 0 || p < min_period || p > max_period) continue;
    int pad = (p - (L % p)) % p;
    if ((L + pad) / p < 2) continue;
    for (int g = 0; g < G; ++g)
      if (d->g_period[g] == p) d->sel_group[j] = g;
  }
  d->n_groups = G;
  d->total_px = px;
  d->tiles_per_row = tiles;
}

#ifdef __HIPCC__
// v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// result register r of lane l is D[i=4*(l>>4)+r][j=l&15]  (cdna_hip_programming.md §3).
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// GELU(v) = 0.5 v (1 + erf(v/sqrt2)) = max(v,0) - a * 2^(-h(a)),  a = |v|,  h(a) = log2(2 / erfc(a/sqrt2)).
// h is smooth and nearly quadratic, so a degree-6 polynomial (weighted Chebyshev fit on [0, 8], weight =
// the term a 2^-h itself; a is clamped to 8, beyond which the term is < 1e-14) reproduces the term to
// 6e-8 and the whole evaluation needs one transcendental (v_exp_f32) instead of v_rcp_f32 + v_exp_f32:
// max abs error vs fp64 GELU over [-10, 10] 2.9e-7 in fp32 (A&S 7.1.26 erfc form: 3.3e-7; libm erff: 6.8e-7).
// VALU and MFMA work of a SIMD serialise on gfx950 (tools/ubench/mfma_valu.hip), so the instruction count
// matters: per value 3.2 full-rate-equivalent pk_fma + v_min + v_max + v_fma + v_exp = ~32 issue cycles.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 gelu_erf2(f2 v) {
  // the polynomial runs on n = -a (odd coefficients sign-flipped: the same Horner values bit for bit), so the last
  // step is a plain fma(n, e, max(v, 0)) - hipcc otherwise negates a with two v_xor per pair instead of neg modifiers
  f2 n, e, r;
  n.x = fmaxf(-fabsf(v.x), -8.0f);
  n.y = fmaxf(-fabsf(v.y), -8.0f);
  f2 p = n * -3.262207974330522e-05f + -0.0007656298694200814f;
  p = p * n + -0.008070714771747589f;
  p = p * n + -0.05339965224266052f;
  p = p * n + 0.45877760648727417f;
  p = p * n + -1.1512006521224976f;
  p = p * n + 0.9999929666519165f;
  e.x = __builtin_amdgcn_exp2f(-p.x);                         // raw v_exp_f32
  e.y = __builtin_amdgcn_exp2f(-p.y);
  r.x = fmaf(n.x, e.x, fmaxf(v.x, 0.0f));
  r.y = fmaf(n.y, e.y, fmaxf(v.y, 0.0f));
  return r;
}
__device__ __forceinline__ float gelu_erf(float v) { return gelu_erf2(f2{v, v}).x; }
// ---- bf16x3 split arithmetic -------------------------------------------------------
// An fp32 value is carried as three bf16 pieces (hi + mid + lo = 24 mantissa bits); a
// product a*b is formed on the bf16 matrix pipe as the six partial products whose weight
// is >= 2^-16 relative (hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi), accumulated in
// fp32.  Measured error is at or below that of an fp32 FMA chain, at 6
// v_mfma_f32_16x16x32_bf16 (96 cycles) instead of 8 v_mfma_f32_16x16x4_f32 (256 cycles).
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma_bf(bf8 a, bf8 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Exact three-way split by truncation: hi = top 16 bits of x, r1 = x - hi (exact, <= 16
// significant bits), mid = top 16 bits of r1, lo = r1 - mid (<= 8 significant bits, exactly a
// bf16).  hi + mid + lo == x bit for bit, and packing two values' pieces into one dword is a
// single v_perm_b32 of their upper halves - no v_cvt, no hazard nops.
__device__ __forceinline__ unsigned pack_hi16(float lo_elem, float hi_elem) {
  // result = [hi_elem.bits[31:16] : lo_elem.bits[31:16]]
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ float trunc16(float v) { return __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); }

// n (even) fp32 values -> NS piece vectors; out[p] holds n bf16 = n/2 dwords
template <int NS, int NV>
__device__ __forceinline__ void split_trunc(const float (&v)[NV], unsigned (&out)[NS][NV / 2]) {
  float r1[NV], r2[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    if (NS == 3) {
      r1[e] = v[e] - trunc16(v[e]);
      r2[e] = r1[e] - trunc16(r1[e]);
    }
  }
#pragma unroll
  for (int k = 0; k < NV / 2; ++k) {
    out[0][k] = pack_hi16(v[2 * k], v[2 * k + 1]);
    if (NS == 3) {
      out[1][k] = pack_hi16(r1[2 * k], r1[2 * k + 1]);
      out[2][k] = pack_hi16(r2[2 * k], r2[2 * k + 1]);
    }
  }
}

// "P3" activation layout: per pixel, per group of 16 channels: [hi 16][mid 16][lo 16] bf16
// (96 bytes), so one K=32 MFMA slab reads 16 contiguous bytes per piece per lane.
// Stores the 4 channels 4q..4q+3 of one group as three 8-byte pieces.
__device__ __forceinline__ void store_p3(__bf16* __restrict__ grp, int q, f4 v) {
  const float vv[4] = {v[0], v[1], v[2], v[3]};
  unsigned pc[3][2];
  split_trunc<3, 4>(vv, pc);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  *(u2*)(grp + 4 * q) = u2{pc[0][0], pc[0][1]};
  *(u2*)(grp + 16 + 4 * q) = u2{pc[1][0], pc[1][1]};
  *(u2*)(grp + 32 + 4 * q) = u2{pc[2][0], pc[2][1]};
}

// ---- f16x2 split arithmetic ("engine f16x2", NS == 2 in the kernels) -----------------------------
// An fp32 activation x is carried as TWO fp16 pieces: hi = fp16(x) (round to nearest) and
// lo' = fp16((x - hi) * 2^11); x - hi is exact in fp32 and at most half an fp16 ulp of x, so lo' has the
// magnitude of x / 2 at most (no underflow while x itself is an fp16 normal) and hi + lo' 2^-11
// reproduces x to 2^-22 relative.  Weights are split on the host into three fp16 pieces of the
// power-of-two-prescaled matrix W~ = 2^s W:  A1 = fp16(W~),  A2 = fp16(A1 2^-11),  A3 = fp16(W~ - A1),
// and a product is the three terms  A1 hi + A2 lo' + A3 hi  on v_mfma_f32_16x16x32_f16 with one fp32
// accumulator (the hi-lo, lo-hi and hi-hi partial products; lo-lo is below 2^-22).  Half the MFMAs and
// two thirds of the activation bytes of the bf16x3 scheme at the same accuracy (pack.py);
// the accumulator carries the scale 2^s (biases are prescaled on the host) and the consumer multiplies by
// 2^-s.  Range: |x| must stay below 65504, the fp16 maximum - beyond it the pieces are +-inf and the
// result is non-finite (never silently wrong); engines bf16x3 / f32 have the full fp32 range.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 mfma_h(bf8 a, bf8 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

#define FTN_H2_LOSCALE 2048.0f

// NV (even) fp32 values -> hi and lo' piece vectors, NV/2 dwords each (v_cvt_pk_f16_f32: round to nearest even).
// lo' = fp16((v - hi) * 2^11) is formed as fma(hi, -2^11, v * 2^11) by v_fma_mixlo/mixhi_f16, which read the fp16
// hi straight out of the packed dword and round the fp32 result to fp16 themselves: per PAIR of values one
// v_cvt_pk, one v_pk_mul and two v_fma_mix instead of 2 v_cvt back + 2 subtract + 2 multiply + a second v_cvt_pk
// (8 -> 4 VALU).  Same value bit for bit: v * 2^11 and hi * 2^11 are exact, their difference is the exact
// (v - hi) * 2^11 (<= 13 significant bits), and both forms round it to fp16 once, to nearest even.
template <int NV>
__device__ __forceinline__ void split_h2(const float (&v)[NV], unsigned (&out)[2][NV / 2]) {
#pragma unroll
  for (int k = 0; k < NV / 2; ++k) {
    const h2v hi = {(_Float16)v[2 * k], (_Float16)v[2 * k + 1]};
    const unsigned hw = __builtin_bit_cast(unsigned, hi);
    const f2 big = f2{v[2 * k], v[2 * k + 1]} * FTN_H2_LOSCALE;
    unsigned lw;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lw) : "v"(hw), "s"(-FTN_H2_LOSCALE), "v"(big.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw) : "v"(hw), "s"(-FTN_H2_LOSCALE), "v"(big.y));
    out[0][k] = hw;
    out[1][k] = lw;
  }
}

// "H2" activation layout: per pixel, per group of 16 channels: [hi 16][lo' 16] fp16 (64 bytes).
__device__ __forceinline__ void store_h2(__bf16* __restrict__ grp, int q, f4 v) {
  const float vv[4] = {v[0], v[1], v[2], v[3]};
  unsigned pc[2][2];
  split_h2<4>(vv, pc);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  *(u2*)(grp + 4 * q) = u2{pc[0][0], pc[0][1]};
  *(u2*)(grp + 16 + 4 * q) = u2{pc[1][0], pc[1][1]};
}

// ---- fp16 range guard of the f16x2 engine --------------------------------------------------------
// A value whose magnitude reaches the fp16 maximum (or is not a number) cannot travel as fp16 pieces: the kernels
// test every value where it is split (x, a, m, a' at their stores; y for NaN / inf at the end - a hidden value g
// that overflows makes a' non-finite, fp32 accumulation keeps that) and set the caller's flag word; the host then
// repeats the call on the bf16x3 engine (full fp32 exponent range).  The flag lives in host-visible memory or in
// device memory; all writers store the same value, so a plain store is enough.
#define FTN_H2_MAX 65504.0f
__device__ __forceinline__ bool h2_bad(float v) { return !(fabsf(v) < FTN_H2_MAX); }
__device__ __forceinline__ bool h2_bad4(f4 v) { return h2_bad(v.x) || h2_bad(v.y) || h2_bad(v.z) || h2_bad(v.w); }
__device__ __forceinline__ bool not_finite4(f4 v) {
  return !(fabsf(v.x) <= 3.402823466e38f) || !(fabsf(v.y) <= 3.402823466e38f) || !(fabsf(v.z) <= 3.402823466e38f) ||
         !(fabsf(v.w) <= 3.402823466e38f);
}
// once per wave, at the end of a kernel
__device__ __forceinline__ void raise_range_flag(int* flag, bool bad) {
  if (flag != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0)
    __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Piece-count-generic forms used by the kernels: NS = 3 bf16x3 (P3, 96 B), NS = 2 f16x2 (H2, 64 B),
// NS = 1 plain bf16 (reads only the hi piece of a P3 record).
template <int NS> struct PxFmt { static constexpr int BYTES = 96; static constexpr int ELEMS = 48; static constexpr int NW = NS; };
template <> struct PxFmt<2> { static constexpr int BYTES = 64; static constexpr int ELEMS = 32; static constexpr int NW = 3; };

template <int NS>
__device__ __forceinline__ void store_px(__bf16* __restrict__ grp, int q, f4 v) {
  if (NS == 2) store_h2(grp, q, v);
  else store_p3(grp, q, v);
}

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if (ACT == 1) return v > 0.0f ? v : 0.0f;
  return gelu_erf(v);
}
template <int ACT>
__device__ __forceinline__ f4 act4(f4 v) {
  f4 r;
  if (ACT == 1) {
    r.x = act_fn<ACT>(v.x); r.y = act_fn<ACT>(v.y); r.z = act_fn<ACT>(v.z); r.w = act_fn<ACT>(v.w);
  } else {
    const f2 lo = gelu_erf2(f2{v.x, v.y}), hi = gelu_erf2(f2{v.z, v.w});
    r.x = lo.x; r.y = lo.y; r.z = hi.x; r.w = hi.y;
  }
  return r;
}
#endif

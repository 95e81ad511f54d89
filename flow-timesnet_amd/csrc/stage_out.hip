// Stages E + F of the TimesBlock forward with stage E on the 16-bit matrix pipe (split engines, gfx950 only):
//   y = x + sum_g w[b,g] * ( act(W_out2 m'_g + b) + r_g )[:L]
// (reference models/timesnet.py :1063-1069 delta, :1075-1092 weighted sum in group order, :818 residual).
//
// k_out (inception.hip) multiplies fp32 m' rows by fp32 fragments with v_mfma_f32_16x16x4_f32: 48 issues of 32
// cycles per (16 positions, group).  Here the second conv leaves m' as the engine's activation pieces (the same
// bytes per pixel for f16x2) and W_out2 arrives as K=32 fragments of three pieces (FtnPlan.w_out2fb): 24 issues of
// 16 cycles.  The fragments (24 KB at d_model 64) sit in LDS, filled once per workgroup; a wave then strides over
// units of 16 window positions and walks the period groups in ascending order - the reference's summation order,
// no atomics - with the next group's (or the next unit's first) m' rows requested one step ahead.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "ftn_common.h"
#include "ftn_mlp.h"
#include "ftn_out.h"

template <int ACT, bool XVEC, int NS, bool RSUM, int NOT, int NSK>
__global__ __launch_bounds__(256, NOT <= 4 ? 3 : 2) void k_out_h(OutArgs a) {
  constexpr int NWP = PxFmt<NS>::NW, PXE = PxFmt<NS>::ELEMS;
  extern __shared__ __attribute__((aligned(16))) char wl[];     // [n_ot * nsKM][3 pieces][1 KiB] | bias [64] fp32
  const FtnDesc* __restrict__ d = a.desc;
  const int total = a.B * a.L, units = (total + 15) >> 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int G = d->n_groups, CP = a.CP, n_ot = CP >> 4, kmg = a.KM >> 4, nsKM = (a.KM + 31) >> 5;
  const int nvec = n_ot * nsKM * 3 * 64;                        // 16-byte vectors
  for (int i = threadIdx.x; i < nvec; i += 256) *(bf8*)(wl + (size_t)i * 16) = *(const bf8*)(a.Wf + (size_t)i * 8);
  if (threadIdx.x < 16 * NOT) ((float*)(wl + (size_t)nvec * 16))[threadIdx.x] = (int)threadIdx.x < CP ? a.bias[threadIdx.x] : 0.f;
  __syncthreads();
  const unsigned bias_off = (unsigned)nvec * 16 + q * 16;
  unsigned wl_off = lane * 16;
  const int ustep = gridDim.x * 4;
  int unit = blockIdx.x * 4 + wave;
  if (unit >= units || G <= 0) return;

  // lane (j, q) of slab s reads 16-channel group 2s + (q >> 1), channels 8 (q & 1) .. +8 of every piece
  auto load_m = [&](int bb, int tt, int g, bf8 (&dst)[NSK][NS]) {
    const int off = d->g_px_off[g], P = d->g_px_off[g + 1] - off;
    const size_t pix = (size_t)a.B * off + (size_t)bb * P + tt;
#pragma unroll
    for (int s = 0; s < NSK; ++s) {
      const int grp = 2 * s + (q >> 1);
      const __bf16* __restrict__ src = a.mh + (pix * kmg + (grp < kmg ? grp : 0)) * PXE + (q & 1) * 8;
#pragma unroll
      for (int pz = 0; pz < NS; ++pz) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[s][pz][e] = (__bf16)0.0f;
        if (grp < kmg) dst[s][pz] = *(const bf8*)(src + pz * 16);
      }
    }
  };
  auto where = [&](int u, int& bb, int& tt, bool& ok) {
    int n = u * 16 + j;
    ok = n < total;
    if (!ok) n = total - 1;
    bb = n / a.L;
    tt = n - bb * a.L;
  };

  int bb, tt;
  bool ok;
  where(unit, bb, tt, ok);
  bf8 mc[NSK][NS];
  load_m(bb, tt, 0, mc);
  bool bad = false;
  for (; unit < units; unit += ustep) {
    f4 yacc[NOT];
#pragma unroll
    for (int o = 0; o < NOT; ++o) yacc[o] = f4{0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;
    // one period group: z = W_out2 m'_g + b on the matrix pipe, yacc += w_g (act(z) + r_g)
    auto group = [&](int g) {
      f4 rr[NOT];
      if (!RSUM) {
        const int off = d->g_px_off[g], P = d->g_px_off[g + 1] - off;
        const float* __restrict__ rrow = a.R + ((size_t)a.B * off + (size_t)bb * P + tt) * CP + 4 * q;
#pragma unroll
        for (int o = 0; o < NOT; ++o) rr[o] = o < n_ot ? *(const f4*)(rrow + 16 * o) : f4{0.f, 0.f, 0.f, 0.f};
      }
      const float w = a.wts[(size_t)bb * FTN_KMAX + g];
      wsum += w;
      // (the fragments are re-read from LDS every group: laundering the offset keeps hipcc from hoisting all of
      // them - 96 registers - out of the loop)
      asm volatile("" : "+v"(wl_off));
      f4 z[NOT];
#pragma unroll
      for (int o = 0; o < NOT; ++o) {
        z[o] = *(const f4*)(wl + bias_off + o * 64);
        if (o < n_ot) {
#pragma unroll
          for (int s = 0; s < NSK; ++s) {
            if (s < nsKM) {
              bf8 fr[NWP];
#pragma unroll
              for (int pz = 0; pz < NWP; ++pz) fr[pz] = *(const bf8*)(wl + wl_off + (size_t)((o * nsKM + s) * 3 + pz) * 1024);
              z[o] = chain_bf<NS>(fr, mc[s], z[o]);
            }
          }
        }
      }
#pragma unroll
      for (int o = 0; o < NOT; ++o) {
        const f4 e = act4<ACT>(NS == 2 ? z[o] * a.inv_out2 : z[o]);
        yacc[o] += (RSUM ? e : e + rr[o]) * w;
      }
    };
    for (int g = 0; g + 1 < G; ++g) {
      bf8 mn[NSK][NS];
      load_m(bb, tt, g + 1, mn);
      group(g);
#pragma unroll
      for (int s = 0; s < NSK; ++s)
#pragma unroll
        for (int pz = 0; pz < NS; ++pz) mc[s][pz] = mn[s][pz];
    }
    // the last group, with what the epilogue needs of this position and the next unit's first rows in flight
    f4 xv[NOT], rs[NOT];
    const size_t row = (size_t)bb * a.L + tt;
#pragma unroll
    for (int o = 0; o < NOT; ++o) {
      const int ch = 16 * o + 4 * q;
      xv[o] = o < n_ot ? load_x4<XVEC>(a.x + row * a.C, ch, a.C) : f4{0.f, 0.f, 0.f, 0.f};
      if (RSUM) rs[o] = o < n_ot ? *(const f4*)(a.R + row * CP + ch) : f4{0.f, 0.f, 0.f, 0.f};
    }
    int nbb = bb, ntt = tt;
    bool nok = false;
    bf8 mn[NSK][NS];
    if (unit + ustep < units) {                                 // wave-uniform
      where(unit + ustep, nbb, ntt, nok);
      load_m(nbb, ntt, 0, mn);
    }
    group(G - 1);
    const bool ln = a.ln_g != nullptr;
    f4 yv[NOT][1];
#pragma unroll
    for (int o = 0; o < NOT; ++o) {
      f4 acc = RSUM ? yacc[o] + rs[o] : yacc[o];                // the group-summed residual of k_mlp_pos
      if (a.r_keeps_x) acc = acc - xv[o] * wsum;                // the x that stage C left inside every R_g
      const f4 nv = xv[o] + acc;
      yv[o][0] = ln ? xv[o] + (nv - xv[o]) : nv;
    }
    if (ln) ln_tiles<NOT, 1>(yv, n_ot, a.C, q, a.ln_g, a.ln_b, a.ln_eps);
#pragma unroll
    for (int o = 0; o < NOT; ++o) {
      if (o < n_ot && ok) {
        bad |= not_finite4(yv[o][0]);
        const int ch = 16 * o + 4 * q;
        float* __restrict__ yrow = a.y + row * a.C;
        if (XVEC) {
          if (ch < a.C) *(f4*)(yrow + ch) = yv[o][0];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ch + r < a.C) yrow[ch + r] = yv[o][0][r];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NSK; ++s)
#pragma unroll
      for (int pz = 0; pz < NS; ++pz) mc[s][pz] = mn[s][pz];
    bb = nbb; tt = ntt; ok = nok;
  }
  raise_range_flag(a.range_flag, bad);
}

static const int g_out_h = [] { const char* e = getenv("FTN_OUT_H"); return e ? atoi(e) : 1; }();   // 0: fp32 k_out
int ftn_out_h_enabled() { return g_out_h; }

template <int ACT, int NS, int NOT, int NSK>
static int launch_out_h_t(const OutArgs& oa, bool xvec, hipStream_t st) {
  const int n_ot = oa.CP >> 4, nsKM = (oa.KM + 31) >> 5;
  const size_t lds = (size_t)n_ot * nsKM * 3 * 1024 + 16 * NOT * sizeof(float);
  const long long units = ((long long)oa.B * oa.L + 15) / 16;
  long long nblk = (units + 3) / 4;
  const int per_cu = NOT <= 4 ? 3 : 2;                          // workgroups per CU; waves stride over the units
  if (nblk > 256 * per_cu) nblk = 256 * per_cu;
  if (nblk < 1) nblk = 1;
#define FTN_OUTH_LAUNCH(XV, RS)                                                                                    \
  {                                                                                                                \
    auto kfn = k_out_h<ACT, XV, NS, RS, NOT, NSK>;                                                                 \
    if (lds > 64 * 1024) {                                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_out_h): %s", hipGetErrorString(e)); return (int)e; } \
    }                                                                                                              \
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(256), lds, st, oa);                                         \
  }
  if (xvec && oa.r_summed) FTN_OUTH_LAUNCH(true, true)
  else if (xvec) FTN_OUTH_LAUNCH(true, false)
  else if (oa.r_summed) FTN_OUTH_LAUNCH(false, true)
  else FTN_OUTH_LAUNCH(false, false)
#undef FTN_OUTH_LAUNCH
  FTN_CHECK_LAUNCH();
  return 0;
}

template <int ACT, int NS>
static int launch_out_h_s(const OutArgs& oa, bool xvec, hipStream_t st) {
  if (oa.CP <= 64 && oa.KM <= 64) return launch_out_h_t<ACT, NS, 4, 2>(oa, xvec, st);
  return launch_out_h_t<ACT, NS, 8, 3>(oa, xvec, st);           // d_model 128, three 32-channel branches
}

int ftn_launch_out_h(const OutArgs& oa, int act, int nsplit, bool xvec, hipStream_t st) {
  if (oa.CP > 128 || oa.KM > 96 || oa.mh == nullptr || oa.Wf == nullptr || oa.act_dtype != 0 || (nsplit != 2 && nsplit != 3)) {
    ftn_set_error("ftn_launch_out_h: unsupported shape (CP=%d KM=%d nsplit=%d)", oa.CP, oa.KM, nsplit);
    return -1;
  }
  if (act == 0) return nsplit == 2 ? launch_out_h_s<0, 2>(oa, xvec, st) : launch_out_h_s<0, 3>(oa, xvec, st);
  return nsplit == 2 ? launch_out_h_s<1, 2>(oa, xvec, st) : launch_out_h_s<1, 3>(oa, xvec, st);
}

// Position-major stage C of the TimesBlock conv path (reference models/timesnet.py:645-654, 744-762, 1063-1092):
// see the comment block at MlpPosArgs in ftn_mlp.h.  Own translation unit: the kernel is instantiated for every
// engine / activation / launch shape and dominates the library's build time.
#include <stdlib.h>
#include "ftn_mlp.h"

// diagnostic cycle stamps (ftn_debug_stamps, tools/stamps_pos.py): thread 0 of a workgroup, 8 words per workgroup
__device__ __forceinline__ void stamp(unsigned long long* buf, size_t cap, size_t wg, int slot) {
  if (buf != nullptr && threadIdx.x == 0 && (wg * 8 + slot) < cap) buf[wg * 8 + slot] = __builtin_amdgcn_s_memtime();
}

// s_waitcnt vmcnt(n) for a wave-uniform n that is a multiple of STEP (vmcnt takes an immediate)
template <int STEP, int MAXN>
__device__ __forceinline__ void wait_vm_keep(int n) {
  if constexpr (MAXN <= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (n >= MAXN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXN > 63 ? 63 : MAXN) : "memory");
    else wait_vm_keep<STEP, MAXN - STEP>(n);
  }
}

// PF: every weight fragment is requested from LDS one fragment ahead of its use (two fragment buffers) and the res2
// accumulators wait in a wave-private LDS slab between chunks, which is where the second buffer's registers come from.
// Without it hipcc, at 254 registers, emits read -> wait -> 3 MFMAs per fragment: 43 exposed LDS round trips per chunk.
template <int ACT, bool XVEC, int NS, int SKM, int SCP, int NOA, int NOR, int NWV, int GB, int NBUF, int PF>
__global__ __launch_bounds__(NWV * 64, 2) void k_mlp_pos(MlpPosArgs pa) {
  const MlpBfArgs& a = pa.c;
  constexpr int NL1 = 2 * SKM + 2 * SCP, NFR = NL1 + NOA + NOR;
  constexpr int NWP = PxFmt<NS>::NW, PXE = PxFmt<NS>::ELEMS;
  constexpr int bufsz = NFR * 3 * 1024;
  extern __shared__ __attribute__((aligned(16))) char wlb[];
  const FtnDesc* __restrict__ d = a.desc;
  const int G = d->n_groups, B = a.B, L = a.L;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, qa = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const bool tail_blk = (int)blockIdx.x >= pa.n_main;
  const int UPR = (L + 15) >> 4;
  int TU = 0;                                                   // tail units per batch row
  if (tail_blk)
    for (int g = 0; g < G; ++g) TU += (d->g_pad[g] + 15) >> 4;
  const int n_tail_units = B * TU;
  const int tv0 = ((int)blockIdx.x - pa.n_main) * NWV, tvstep = pa.n_tail * NWV;
  const int iters = tail_blk ? (tv0 < n_tail_units ? (n_tail_units - tv0 + tvstep - 1) / tvstep : 0) : (G + GB - 1) / GB;
  if (tail_blk && iters == 0) return;
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 0);

  // ---- this wave's window positions (main blocks)
  const int u = __builtin_amdgcn_readfirstlane((int)blockIdx.x * NWV + wave);
  const bool active_m = !tail_blk && u < B * UPR;
  const int uc = active_m ? u : 0;
  const int bm = __builtin_amdgcn_readfirstlane(uc / UPR), t0m = (uc - bm * UPR) * 16;   // (the quotient comes out of the VALU)
  const bool ok_m = active_m && t0m + j < L;
  const int tcm = t0m + j < L ? t0m + j : L - 1;

  auto dma_chunk = [&](int hc, int buf) {
    const __bf16* __restrict__ src = a.cfrag + (size_t)hc * NFR * 3 * 512;
    char* dst = wlb + (size_t)buf * bufsz;
    for (int piece = wv; piece < NFR * 3; piece += NWV)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(dst + (size_t)piece * 1024), 16, 0, 0);
  };
  // ---- prologue, ordered for latency (it was 13 % of a workgroup's life as load -> wait -> store chains): every
  // global load is issued before anything waits - x rows, this batch row's group weights as one fixed-trip batch
  // (selects instead of a data-dependent loop), then LDS is filled.  The hidden-layer biases are not staged at all:
  // a chunk reads its 2 x 2 bias tiles straight from L2 while it waits for its weights.
  const int kmg = a.KM >> 4;
  char* __restrict__ xl = wlb + (size_t)NBUF * bufsz + (size_t)wave * (SCP * NS * 1024) + lane * 16;
  bool range_bad = false;                                       // f16x2: a value left the fp16 range (ftn_common.h)
  f4 xr[SCP][2];
  {
    const float* __restrict__ xrow = a.x + ((size_t)bm * L + tcm) * a.C;      // a real row even for idle waves (masked below)
#pragma unroll
    for (int s = 0; s < SCP; ++s) {
      xr[s][0] = load_x4<XVEC>(xrow, 32 * s + 8 * qa, a.C);
      xr[s][1] = load_x4<XVEC>(xrow, 32 * s + 8 * qa + 4, a.C);
    }
  }
  float wsum = 0.f;
  {
    float wrow[FTN_KMAX];
#pragma unroll
    for (int g = 0; g < FTN_KMAX; ++g) wrow[g] = pa.wts[(size_t)bm * FTN_KMAX + g];
#pragma unroll
    for (int g = 0; g < FTN_KMAX; ++g) wsum += (active_m && g < G) ? wrow[g] : 0.f;
  }
  // x pieces (main blocks; the tail pixels are live zeros): split once, parked in this wave's own LDS slab (lane-linear,
  // read back once per hidden chunk - 16 registers that the group loop needs more)
#pragma unroll
  for (int s = 0; s < SCP; ++s) {
    float xv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { xv[e] = active_m ? xr[s][0][e] : 0.f; xv[4 + e] = active_m ? xr[s][1][e] : 0.f; }
    if (NS == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) range_bad |= h2_bad(xv[e]);
    }
    bf8 xp[NS];
    split_pieces<NS>(xv, xp);
#pragma unroll
    for (int pz = 0; pz < NS; ++pz) *(bf8*)(xl + (size_t)(s * NS + pz) * 1024) = xp[pz];
  }
  f4 racc[NOR];
  // (PF: the accumulators live in this wave's LDS slab, [NOR][64 lanes] f4, between their once-per-chunk updates)
  char* __restrict__ rl = wlb + (size_t)NBUF * bufsz + (size_t)NWV * (SCP * NS * 1024) + (size_t)wave * (NOR * 1024) + lane * 16;
#pragma unroll
  for (int o = 0; o < NOR; ++o) {
    racc[o] = *(const f4*)(a.bc + 16 * (NOA + o) + 4 * qa) * wsum;
    if constexpr (PF == 1) *(f4*)(rl + o * 1024) = racc[o];
  }

  for (int it = 0; it < iters; ++it) {
    // ---- the groups of this pass: flat pixel index n = base[i] + tc (base wave-uniform)
    int gcnt, tc;
    bool ok;
    int base[GB];
    float wg[GB];
    if (!tail_blk) {
      const int g_lo = it * GB;
      gcnt = G - g_lo < GB ? G - g_lo : GB;
      if (!active_m) gcnt = 0;
      tc = tcm; ok = ok_m;
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const int g = g_lo + i < G ? g_lo + i : G - 1;
        const int off = d->g_px_off[g], P = d->g_px_off[g + 1] - off;      // independent scalar loads, one wait
        const float w = pa.wts[(size_t)bm * FTN_KMAX + g];
        base[i] = B * off + bm * P;
        wg[i] = i < gcnt ? w : 0.f;
      }
    } else {
      const int v = __builtin_amdgcn_readfirstlane(tv0 + it * tvstep + wave);
      const bool act = v < n_tail_units;
      const int vc = act ? v : 0;
      const int b = __builtin_amdgcn_readfirstlane(vc / TU);
      int r = vc - b * TU, g = 0;
      for (; g < G - 1; ++g) {
        const int tu = (d->g_pad[g] + 15) >> 4;
        if (r < tu) break;
        r -= tu;
      }
      const int off = d->g_px_off[g], P = d->g_px_off[g + 1] - off;
      const int t = L + 16 * r + j;
      gcnt = act ? 1 : 0;
      ok = act && t < P;
      tc = t < P ? t : P - 1;
#pragma unroll
      for (int i = 0; i < GB; ++i) { base[i] = B * off + b * P; wg[i] = 0.f; }
    }
    f4 aacc[GB][NOA];
#pragma unroll
    for (int i = 0; i < GB; ++i)
#pragma unroll
      for (int o = 0; o < NOA; ++o) aacc[i][o] = *(const f4*)(a.bc + 16 * o + 4 * qa);
    __builtin_amdgcn_sched_barrier(0);
    dma_chunk(0, 0);                                            // the previous pass's closing barrier freed the buffer
    __builtin_amdgcn_sched_barrier(0);
    bf8 mp[GB][SKM][NS];
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      if (i < gcnt) {
#pragma unroll
        for (int s = 0; s < SKM; ++s) {
          const int grp = 2 * s + (qa >> 1);
          const __bf16* __restrict__ src = a.m + ((size_t)(base[i] + tc) * kmg + (grp < kmg ? grp : 0)) * PXE + (qa & 1) * 8;
          // lanes past the last 16-channel group keep zeros: a predicated load, not a select on loaded data (which
          // would wait for the rows here instead of at their first use)
#pragma unroll
          for (int pz = 0; pz < NS; ++pz) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mp[i][s][pz][e] = (__bf16)0.0f;
            if (grp < kmg) mp[i][s][pz] = *(const bf8*)(src + pz * 16);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int hc = 0; hc < a.n_hchunks; ++hc) {
      const bool sync = !((pa.abl & 1) && hc > 0);
      const bool st = it == 0 && hc == 1;
      if (it == 0 && hc == 0) stamp(a.dbg, a.dbg_cap, blockIdx.x, 1);
      if (st) stamp(a.dbg, a.dbg_cap, blockIdx.x, 2);
      // this chunk's bias tiles (zero past d_ff): requested ahead of the weight wait, which covers their latency
      f4 bo_t[2], res1[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool tile = (hc * 2 + t) * 16 < a.FP;              // wave-uniform; FP is a multiple of 16
        res1[t] = tile ? *(const f4*)(a.br + 16 * (hc * 2 + t) + 4 * qa) : f4{0.f, 0.f, 0.f, 0.f};
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NBUF == 1 && sync && hc > 0) dma_chunk(hc, 0);
      if (sync) {
        // chunk 0's fragments were requested BEFORE this pass's m rows (vmcnt retires in order): wait for them and
        // leave the rows in flight - 80 KB per workgroup at ~11 B/clk/CU is 7.5 k cycles, which the first chunk's
        // res1 and first groups now cover
        if (hc == 0) wait_vm_keep<SKM * NS, GB * SKM * NS>(gcnt * SKM * NS);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (a raw barrier: __syncthreads() would drain vmcnt to zero again)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // chunk hc has landed; every wave has left chunk hc - 1
      }
      if (st) stamp(a.dbg, a.dbg_cap, blockIdx.x, 3);
      if (NBUF == 2 && hc + 1 < a.n_hchunks) dma_chunk(hc + 1, (hc + 1) & 1);
      unsigned wl_off = (unsigned)((NBUF == 2 ? (hc & 1) : 0) * bufsz + lane * 16);
      // every group re-reads the fragments from LDS: laundering the offset keeps hipcc from holding all 7 group-shared
      // fragments (84 registers) across the unrolled group loop, which spills
      auto ldfrag = [&](int f, bf8 (&ap)[NWP]) {
#pragma unroll
        for (int pz = 0; pz < NWP; ++pz) ap[pz] = *(const bf8*)(wlb + wl_off + (size_t)(f * 3 + pz) * 1024);
      };
#pragma unroll
      for (int t = 0; t < 2; ++t)                                 // (needed after the res1 products, which cover the load)
        bo_t[t] = (hc * 2 + t) * 16 < a.FP ? *(const f4*)(a.bo + 16 * (hc * 2 + t) + 4 * qa) : f4{0.f, 0.f, 0.f, 0.f};
      if constexpr (PF == 1) {
        if (gcnt > 0) {
          constexpr int NFG = 2 * SKM + NOA;                      // fragments one group walks: layer 1, then layer 2a
          bf8 fr[2][NWP];
          // res1(x) + b, once for all groups of the pass
          {
            bf8 xp[SCP][NS];
            ldfrag(2 * SKM, fr[0]);
#pragma unroll
            for (int s = 0; s < SCP; ++s)
#pragma unroll
              for (int pz = 0; pz < NS; ++pz) xp[s][pz] = *(const bf8*)(xl + (size_t)(s * NS + pz) * 1024);
#pragma unroll
            for (int k = 0; k < 2 * SCP; ++k) {
              if (k + 1 < 2 * SCP) ldfrag(2 * SKM + k + 1, fr[(k + 1) & 1]);
              else ldfrag(0, fr[(k + 1) & 1]);                    // the first group's first fragment
              __builtin_amdgcn_sched_barrier(0);
              res1[k / SCP] = chain_bf<NS>(fr[k & 1], xp[k % SCP], res1[k / SCP]);
            }
          }
          static_assert((2 * SCP) % 2 == 0, "group 0 starts on fragment buffer 0");
          if (NS == 2) { res1[0] = res1[0] * a.inv_r; res1[1] = res1[1] * a.inv_r; }
          f4 sacc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int i = 0; i < GB; ++i) {
            if (i < gcnt) {
              asm volatile("" : "+v"(wl_off));
              f4 h[2] = {bo_t[0], bo_t[1]};
              bf8 hp[NS];
#pragma unroll
              for (int k = 0; k < NFG; ++k) {
                constexpr int dummy = 0; (void)dummy;
                const int cur = (i * NFG + k) & 1;
                // the next fragment of this group, or - behind the last one - the first fragment of the next group (the
                // same weights for every group; after the pass's last group the read is simply not used)
                ldfrag(k + 1 < 2 * SKM ? k + 1 : (k + 1 < NFG ? NL1 + (k + 1 - 2 * SKM) : 0), fr[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (k < 2 * SKM) h[k / SKM] = chain_bf<NS>(fr[cur], mp[i][k % SKM], h[k / SKM]);
                else aacc[i][k - 2 * SKM] = chain_bf<NS>(fr[cur], hp, aacc[i][k - 2 * SKM]);
                if (k == 2 * SKM - 1) {
                  // g_g = act(act(W_out1 m_g + b) + res1(x))  (:652-654, then TimesBlock's mid activation :757)
#pragma unroll
                  for (int t = 0; t < 2; ++t) {
                    h[t] = act4<ACT>(NS == 2 ? h[t] * a.inv_o : h[t]) + res1[t];
                    h[t] = act4<ACT>(h[t]);
                    sacc[t] += h[t] * wg[i];
                  }
                  const float hv[8] = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
                  split_pieces<NS>(hv, hp);
                }
              }
            }
          }
          if (!tail_blk) {
            bf8 sp[NS], f2[2][NWP];
            ldfrag(NL1 + NOA, f2[0]);
            const float sv[8] = {sacc[0][0], sacc[0][1], sacc[0][2], sacc[0][3], sacc[1][0], sacc[1][1], sacc[1][2], sacc[1][3]};
            split_pieces<NS>(sv, sp);
            // all NOR accumulator tiles come back from the slab in ONE round trip (the groups' h / hp / fragment
            // registers are free here), not one per tile
            f4 rr[NOR];
#pragma unroll
            for (int o = 0; o < NOR; ++o) rr[o] = *(const f4*)(rl + o * 1024);
#pragma unroll
            for (int o = 0; o < NOR; ++o) {
              if (o + 1 < NOR) ldfrag(NL1 + NOA + o + 1, f2[(o + 1) & 1]);
              __builtin_amdgcn_sched_barrier(0);
              rr[o] = chain_bf<NS>(f2[o & 1], sp, rr[o]);
            }
#pragma unroll
            for (int o = 0; o < NOR; ++o) *(f4*)(rl + o * 1024) = rr[o];
          }
        }
      } else
      if (gcnt > 0) {
        // res1(x) + b, once for all groups of the pass (NS == 2: the accumulator carries sc_res1)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int s = 0; s < SCP; ++s) {
            bf8 fr[NWP], xp[NS];
            ldfrag(2 * SKM + t * SCP + s, fr);
#pragma unroll
            for (int pz = 0; pz < NS; ++pz) xp[pz] = *(const bf8*)(xl + (size_t)(s * NS + pz) * 1024);
            res1[t] = chain_bf<NS>(fr, xp, res1[t]);
          }
        if (NS == 2) { res1[0] = res1[0] * a.inv_r; res1[1] = res1[1] * a.inv_r; }
        f4 sacc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          if (i < gcnt) {
            asm volatile("" : "+v"(wl_off));
            f4 h[2] = {bo_t[0], bo_t[1]};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int s = 0; s < SKM; ++s) {
                bf8 fr[NWP];
                ldfrag(t * SKM + s, fr);
                h[t] = chain_bf<NS>(fr, mp[i][s], h[t]);
              }
            // g_g = act(act(W_out1 m_g + b) + res1(x))  (:652-654, then TimesBlock's mid activation :757)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              h[t] = act4<ACT>(NS == 2 ? h[t] * a.inv_o : h[t]) + res1[t];
              h[t] = act4<ACT>(h[t]);
              sacc[t] += h[t] * wg[i];
            }
            bf8 hp[NS];
            {
              const float hv[8] = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
              split_pieces<NS>(hv, hp);
            }
#pragma unroll
            for (int o = 0; o < NOA; ++o) {
              bf8 fr[NWP];
              ldfrag(NL1 + o, fr);
              aacc[i][o] = chain_bf<NS>(fr, hp, aacc[i][o]);
            }
          }
        }
        if (!tail_blk) {
          bf8 sp[NS];
          const float sv[8] = {sacc[0][0], sacc[0][1], sacc[0][2], sacc[0][3], sacc[1][0], sacc[1][1], sacc[1][2], sacc[1][3]};
          split_pieces<NS>(sv, sp);
#pragma unroll
          for (int o = 0; o < NOR; ++o) {
            bf8 fr[NWP];
            ldfrag(NL1 + NOA + o, fr);
            racc[o] = chain_bf<NS>(fr, sp, racc[o]);
          }
        }
      }
      if (st) stamp(a.dbg, a.dbg_cap, blockIdx.x, 4);
      if (NBUF == 1 && !((pa.abl & 1) && hc + 1 < a.n_hchunks)) __syncthreads();   // every wave is done with the single buffer
      if (st) stamp(a.dbg, a.dbg_cap, blockIdx.x, 5);
    }
    if (NBUF == 2) __syncthreads();                             // the next pass refills buffer 0
    if (it == 0) stamp(a.dbg, a.dbg_cap, blockIdx.x, 6);
    // ---- a'_g of this pass
    if (ok) {
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        if (i < gcnt) {
#pragma unroll
          for (int o = 0; o < NOA; ++o) {
            const f4 av = NS == 2 ? aacc[i][o] * a.inv_a : aacc[i][o];
            if (NS == 2) range_bad |= h2_bad4(av);
            store_px<NS == 2 ? 2 : 3>(a.outA + ((size_t)(base[i] + tc) * (a.AC >> 4) + o) * PXE, qa, av);
          }
        }
      }
    }
  }
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 7);
  if (NS == 2) raise_range_flag(a.range_flag, range_bad);
  if (!ok_m) return;
  float* __restrict__ rrow = pa.outRs + ((size_t)bm * L + tcm) * a.CP + 4 * qa;
#pragma unroll
  for (int o = 0; o < NOR; ++o) {
    if constexpr (PF == 1) racc[o] = *(const f4*)(rl + o * 1024);
    *(f4*)(rrow + 16 * o) = NS == 2 ? racc[o] * a.inv_r2 : racc[o];
  }
}

static const int g_mlp_pos = [] { const char* e = getenv("FTN_MLP_POS"); return e ? atoi(e) : 1; }();      // 0: pixel-major k_mlp_bf_u1
static const int g_mlp_pos_nwv = [] { const char* e = getenv("FTN_MLP_POS_NWV"); return e ? atoi(e) : 0; }();   // experiment: waves per workgroup
static const int g_mlp_pos_abl = [] { const char* e = getenv("FTN_MLP_POS_ABL"); return e ? atoi(e) : 0; }();   // timing ablations
static const int g_mlp_pos_pf = [] { const char* e = getenv("FTN_MLP_POS_PF"); return e ? atoi(e) : 1; }();     // 0: no fragment prefetch
static const int g_mlp_pos_gb = [] { const char* e = getenv("FTN_MLP_POS_GB"); return e ? atoi(e) : 0; }();     // experiment: groups per pass

template <int ACT, bool XVEC, int NS, int SKM, int SCP, int NOA, int NOR, int NWV, int GB, int NBUF, int PF>
static int launch_mlp_pos_t(MlpPosArgs pa, int tail_units_bound, hipStream_t st) {
  constexpr int NFR = 2 * SKM + 2 * SCP + NOA + NOR;
  const size_t lds = (size_t)NBUF * NFR * 3 * 1024 + (size_t)NWV * SCP * NS * 1024 + (PF == 1 ? (size_t)NWV * NOR * 1024 : 0);
  if (lds > 160 * 1024) { ftn_set_error("position-major stage C needs %zu B of LDS", lds); return -1; }
  const long long units = (long long)pa.c.B * ((pa.c.L + 15) / 16);
  pa.n_main = (int)((units + NWV - 1) / NWV);
  // the tail pixels are few (pad_g < period): a fixed set of workgroups strides over them
  const int tail_wg = (tail_units_bound + NWV - 1) / NWV;
  pa.n_tail = tail_wg < 512 ? tail_wg : 512;
  pa.abl = g_mlp_pos_abl;
  auto kfn = k_mlp_pos<ACT, XVEC, NS, SKM, SCP, NOA, NOR, NWV, GB, NBUF, PF>;
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_mlp_pos): %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(kfn, dim3(pa.n_main + pa.n_tail), dim3(NWV * 64), lds, st, pa);
  FTN_CHECK_LAUNCH();
  return 0;
}

// d_model-64 shape (two K slabs each, 3 + 4 output tiles).  Launch shape: two 4-wave workgroups per CU with a single
// weight buffer each (a workgroup's exposed refill is covered by its neighbour), or for small batches 2-wave
// workgroups so that the grid still covers the chip.
// d_model-64 shape (two K slabs each, 3 + 4 output tiles).  Launch shape: two 4-wave workgroups per CU with a single
// weight buffer each (a workgroup's exposed refill is covered by its neighbour); an 8-wave workgroup with two weight
// buffers (FTN_MLP_POS_NWV=8, f16x2 only) measured 5 % slower, 2-wave workgroups 55 % slower.
template <int ACT, int NS>
static int launch_mlp_pos64(const MlpPosArgs& pa, bool xvec, int tail_units_bound, hipStream_t st) {
  const int nwv = g_mlp_pos_nwv ? g_mlp_pos_nwv : 4;
  // groups per pass = what 256 registers hold: 5 with two activation pieces (f16x2) or one (bf16), 4 with three (bf16x3)
  constexpr int GBD = NS == 3 ? 4 : 5;
  const int gb = g_mlp_pos_gb ? g_mlp_pos_gb : GBD;
#define FTN_POS_CASE(W, GBV, NB, PFV)                                                                                         \
  if (nwv == W && gb == GBV && pf == PFV)                                                                                     \
    return xvec ? launch_mlp_pos_t<ACT, true, NS, 2, 2, 3, 4, W, GBV, NB, PFV>(pa, tail_units_bound, st)                     \
                : launch_mlp_pos_t<ACT, false, NS, 2, 2, 3, 4, W, GBV, NB, PFV>(pa, tail_units_bound, st);
  // fragment prefetch (template PF): the f16x2 default; FTN_MLP_POS_PF=0 the form without it
  const int pf = (NS == 2 && nwv == 4 && gb == GBD) ? (g_mlp_pos_pf ? 1 : 0) : 0;
  if constexpr (NS == 2) { FTN_POS_CASE(4, GBD, 1, 1) }
  FTN_POS_CASE(4, GBD, 1, 0)
  if constexpr (NS == 2) {
    FTN_POS_CASE(8, GBD, 2, 0)
    FTN_POS_CASE(4, 4, 1, 0)
  }
#undef FTN_POS_CASE
  ftn_set_error("position-major stage C: no build for FTN_MLP_POS_NWV=%d FTN_MLP_POS_GB=%d", nwv, gb);
  return -1;
}

// d_model-128 shape (three 32-channel branches: K = 96 = three slabs, C = 128 = four slabs, 6 + 8 output tiles; f16x2).
// 48 registers per group (m pieces 24, a' accumulators 24) + 32 for R: three groups per pass at 256 registers with 42
// spilled (two fit without spills, but then five groups are three passes: 530 us on the c4 shard against 490 with
// three and 500 for the pixel-major k_mlp_bf_u1); a chunk's 28 fragments are 84 KB, so ONE 8-wave workgroup per CU
// (84 KB + 8 x 8 KB of x pieces).  FTN_MLP_POS_GB=2 selects the spill-free form.
template <int ACT>
static int launch_mlp_pos128(const MlpPosArgs& pa, bool xvec, int tail_units_bound, hipStream_t st) {
  const int gb = g_mlp_pos_gb == 2 ? 2 : 3;
  if (gb == 3)
    return xvec ? launch_mlp_pos_t<ACT, true, 2, 3, 4, 6, 8, 8, 3, 1, 0>(pa, tail_units_bound, st)
                : launch_mlp_pos_t<ACT, false, 2, 3, 4, 6, 8, 8, 3, 1, 0>(pa, tail_units_bound, st);
  return xvec ? launch_mlp_pos_t<ACT, true, 2, 3, 4, 6, 8, 8, 2, 1, 0>(pa, tail_units_bound, st)
              : launch_mlp_pos_t<ACT, false, 2, 3, 4, 6, 8, 8, 2, 1, 0>(pa, tail_units_bound, st);
}

int ftn_launch_mlp_pos128(const MlpPosArgs& pa, int act, bool xvec, int tail_units_bound, hipStream_t st) {
#ifdef FTN_POS_DEV
  if (act == 0) return launch_mlp_pos128<0>(pa, xvec, tail_units_bound, st);
  ftn_set_error("FTN_POS_DEV build: only GELU");
  return -1;
#else
  return act == 1 ? launch_mlp_pos128<1>(pa, xvec, tail_units_bound, st) : launch_mlp_pos128<0>(pa, xvec, tail_units_bound, st);
#endif
}

int ftn_mlp_pos_enabled() { return g_mlp_pos; }

// FTN_POS_DEV=1 at compile time: only the bench shape's instantiation (GELU, f16x2), for fast kernel iteration
int ftn_launch_mlp_pos64(const MlpPosArgs& pa, int act, int nsplit, bool xvec, int tail_units_bound, hipStream_t st) {
#ifdef FTN_POS_DEV
  if (act == 0 && nsplit == 2) return launch_mlp_pos64<0, 2>(pa, xvec, tail_units_bound, st);
  ftn_set_error("FTN_POS_DEV build: only GELU / f16x2");
  return -1;
#else
  if (act == 1) {
    if (nsplit == 3) return launch_mlp_pos64<1, 3>(pa, xvec, tail_units_bound, st);
    if (nsplit == 2) return launch_mlp_pos64<1, 2>(pa, xvec, tail_units_bound, st);
    return launch_mlp_pos64<1, 1>(pa, xvec, tail_units_bound, st);
  }
  if (nsplit == 3) return launch_mlp_pos64<0, 3>(pa, xvec, tail_units_bound, st);
  if (nsplit == 2) return launch_mlp_pos64<0, 2>(pa, xvec, tail_units_bound, st);
  return launch_mlp_pos64<0, 1>(pa, xvec, tail_units_bound, st);
#endif
}

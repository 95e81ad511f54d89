// S3-S5 of the period selector as a device function (one 256-thread workgroup), shared by selector.hip's k_finalize
// and by the fused finalize + stage-A launch of inception.hip (ftn_period_finalize_stage_a).
#pragma once
#include <math.h>
#include "ftn_common.h"

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

struct FinalizeArgs {
  const double* psum; int nparts, Btotal;
  const float* med; int B, L, F, kcfg, pmax, min_thr;
  FtnDesc* desc; float* amps; float* wts;
  int act_dtype, max_unique; float log_den;   // log_den = (float)log(base) of TIMES_PERIOD_BINNING, evaluated in double on the host (0 = off)
  // multi-GPU exchange (FtnExchange): the nparts partial sums are peer-written slots; ready[p * ready_n + i] turns
  // ready_seq when block i of rank p's k_colsum has stored its columns.  psum rows are psum_stride doubles apart.
  const unsigned long long* ready; unsigned long long ready_seq; int ready_n, psum_stride; int* xerr;
  unsigned long long* dbg;   // diagnostic s_memtime stamps of the finalize workgroup's phases (ftn_debug_stamps which & 8), 8 words
};

// exchange buffer of one rank: two halves (seq parity); per half [world][F_cap] doubles, then [world][FTN_XCHG_NBLK]
// sequence words; one error word at the very end
#define FTN_XCHG_NBLK 32
__host__ __device__ inline size_t ftn_xchg_half_bytes(int world, int F_cap) {
  return ((size_t)world * ((size_t)F_cap * 8 + FTN_XCHG_NBLK * 8) + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t ftn_xchg_flags_off(int world, int F_cap) { return (size_t)world * F_cap * 8; }

// One 256-thread workgroup.  Dynamic LDS: F rounded up to a multiple of 16 floats (scores) + as many 64-bit keys
// (ftn_finalize_lds_bytes).
static inline size_t ftn_finalize_lds_bytes(int F) { return (size_t)((F + 15) & ~15) * (sizeof(float) + sizeof(unsigned long long)); }

__device__ __forceinline__ void finalize_body(const FinalizeArgs& fa) {
  const double* __restrict__ psum = fa.psum;
  const int nparts = fa.nparts, Btotal = fa.Btotal;
  const float* __restrict__ med = fa.med;
  const int B = fa.B, L = fa.L, F = fa.F, kcfg = fa.kcfg, pmax = fa.pmax, min_thr = fa.min_thr;
  FtnDesc* __restrict__ desc = fa.desc;
  float* __restrict__ amps = fa.amps;
  float* __restrict__ wts = fa.wts;
  const int act_dtype = fa.act_dtype, max_unique = fa.max_unique;
  const float log_den = fa.log_den;
  extern __shared__ __attribute__((aligned(16))) float score[];  // [F]
  __shared__ int sel_idx[FTN_KMAX];
  __shared__ FtnDesc sd;
  __shared__ float red[256][FTN_KMAX + 1];                        // block reductions of the flagged grouping; per-row scratch
  __shared__ float wred[256][FTN_KMAX + 1];
  __shared__ float colmean[FTN_KMAX], gscore[FTN_KMAX];
  __shared__ int c_assign[FTN_KMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto fstamp = [&](int slot) { if (fa.dbg != nullptr && tid == 0) fa.dbg[slot] = __builtin_amdgcn_s_memtime(); };
  fstamp(0);

  const int pstride = fa.psum_stride > 0 ? fa.psum_stride : F;
  __shared__ int peers_late;
  if (fa.ready != nullptr) {
    // every (rank, column block) sequence word must have turned ready_seq.  Bounded wait: s_memrealtime ticks at
    // 100 MHz, 2 s = 2e8 ticks - a peer that died must not hang this GPU.  System-scope loads: the words and the
    // slots are written by other GPUs (or, in a rehearsal, other processes) and must not come from a stale cache line.
    if (tid == 0) peers_late = 0;
    __syncthreads();
    const int nflag = nparts * fa.ready_n;
    for (int i = tid; i < nflag; i += 256) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(fa.ready + (size_t)(i / fa.ready_n) * FTN_XCHG_NBLK + (i % fa.ready_n), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM) != fa.ready_seq) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { peers_late = 1; break; }
        __builtin_amdgcn_s_sleep(32);
      }
    }
    __threadfence_system();
    __syncthreads();
    if (peers_late && tid == 0 && fa.xerr != nullptr) __hip_atomic_store(fa.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // mean over the (global) batch, DC kill, log penalty          (:112-130)
  for (int f = tid; f < F; f += 256) {
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) {
      const double* src = psum + (size_t)p * pstride + f;
      s += fa.ready != nullptr ? __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)src, __ATOMIC_RELAXED,
                                                                              __HIP_MEMORY_SCOPE_SYSTEM))
                               : *src;
    }
    if (fa.ready != nullptr && peers_late) s = 0.0 / 0.0;       // no valid period below: the block becomes the identity
    float m = rnd_act((float)(s / (double)Btotal), act_dtype);
    float sc = rnd_act(m - rnd_act(1e-8f * rnd_act(log1pf((float)f), act_dtype), act_dtype), act_dtype);
    score[f] = (f == 0) ? -INFINITY : sc;
  }
  __syncthreads();
  fstamp(1);
  int k = kcfg < F - 1 ? kcfg : F - 1;                            // :122-123
  if (k > FTN_KMAX) k = FTN_KMAX;
  if (k < 0) k = 0;
  // top-k by ONE wavefront, no barriers: every lane keeps the best of its strided share of the bins, a
  // shuffle butterfly reduces the 64 candidates, the winner's bin is retired (NaN) and the lane that owned it
  // rescans its share.  k <= 16 rounds of ~12 shuffles; ties resolve to the lowest bin index.
  if (F <= 256) {
    // rank counting (round 3; one bin per thread - beyond 256 bins the F^2 / 256 compares per thread overtake the
    // k rounds below): every bin counts the bins that beat it (larger score, ties to the lower index - the
    // order better() defines) with broadcast LDS reads; a bin of rank r < k is the r-th pick.  No dependent rounds:
    // the k-round shuffle arg-max below was 16 k cycles of this workgroup's 44 k at F = 169, k = 5.  NaN scores never
    // compare true, so they neither count nor are picked; missing picks stay -1.
    // One 64-bit key per bin - the score mapped to an order-preserving integer in the high word, ~bin in the low word,
    // 0 for NaN and for the padding - makes "beats" a single unsigned compare (mask-producing compare / select chains
    // are slow here: the two-compare-and-index form of this loop took 27 k cycles).  Keys sit behind score[] in LDS.
    const int FP16 = (F + 15) & ~15;
    unsigned long long* __restrict__ key = (unsigned long long*)(score + FP16);
    if (tid < FTN_KMAX) sel_idx[tid] = -1;
    for (int f = tid; f < FP16; f += 256) {
      unsigned long long kf = 0ull;
      if (f < F) {
        const float v = score[f];
        const unsigned bits = __float_as_uint(v);
        const unsigned mono = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);   // -inf .. +inf ascending, >= 0x007fffff
        if (v == v) kf = ((unsigned long long)mono << 32) | (unsigned)(~f);
      }
      key[f] = kf;
    }
    __syncthreads();
    for (int f = tid; f < F; f += 256) {
      const unsigned long long kf = key[f];
      if (kf != 0ull) {
        int rank = 0;
#pragma unroll 1
        for (int g = 0; g < FP16; g += 16) {                      // eight reads in flight per LDS round trip
          typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
          u2 a[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) a[c] = *(const u2*)(key + g + 2 * c);
#pragma unroll
          for (int c = 0; c < 8; ++c) rank += (a[c].x > kf ? 1 : 0) + (a[c].y > kf ? 1 : 0);
        }
        if (rank < k) sel_idx[rank] = f;
      }
    }
  } else if (wave == 0) {
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
  fstamp(2);
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
    // (lane t's value for a compile-time t is a v_readlane_b32, not a ds_bpermute round trip: these three loops were
    // ~100 dependent LDS-crossbar shuffles)
    const unsigned long long validm = __ballot(valid);
#pragma unroll
    for (int t = 0; t < FTN_KMAX; ++t) {
      const int pt = __builtin_amdgcn_readlane(pc, t);
      if (((validm >> t) & 1ull) && pt == pc && t < lane) first = false;
    }
    const unsigned long long firstm = __ballot(first);
#pragma unroll
    for (int t = 0; t < FTN_KMAX; ++t) {
      const int pt = __builtin_amdgcn_readlane(pc, t);
      if (((firstm >> t) & 1ull) && pt < pc) ++rank;
    }
    const int G = __popcll(firstm);
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
#pragma unroll
    for (int t = 0; t < FTN_KMAX; ++t) {
      const int a_ = __builtin_amdgcn_readlane(gpx, t), b_ = __builtin_amdgcn_readlane(gtl, t);
      if (t < lane) { opx += a_; otl += b_; }
    }
    if (lane <= FTN_KMAX) { sd.g_px_off[lane] = opx; sd.g_tile_off[lane] = otl; }   // lanes >= G hold the totals
    if (lane == 0) {
      sd.n_sel = nsel; sd.n_groups = G;
    }
    if (lane == FTN_KMAX) { sd.total_px = opx; sd.tiles_per_row = otl; }
  }
  __syncthreads();
  fstamp(3);
  // ---- TIMES_PERIOD_BINNING / TIMES_PERIOD_MAX_UNIQ (reference :350-437; resolved per block depth on the host and
  //      passed in): candidates are grouped by log bucket instead of by period, and / or only the `max_unique`
  //      groups with the largest batch-mean logsumexp survive, the others joining the kept group of nearest
  //      period.  A group's period is its member with the largest batch-mean amplitude.  Needs two reductions
  //      over the batch (column means, group scores), done in a fixed order; the small-K logic runs on thread 0.
  if (max_unique > 0 || log_den > 0.0f) {
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
        // :350-354: torch divides the fp32 log of the period by math.log(base) (a double, rounded to fp32 as the scalar operand)
        key[j] = !ok ? -1 : (log_den > 0.0f ? (int)floorf(logf((float)p) / log_den + 1e-6f) : p);
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
  // One thread per batch row, its candidate values in its own LDS row (17-float stride: conflict-free).  This is a
  // single-workgroup kernel on the critical path of every block call and it runs with a cold instruction cache:
  // what it costs is code BYTES fetched, not instructions executed - the fully unrolled, predicated 16-slot form
  // of this section (16 inlined expf, 16 divisions, a 16 x 16 select scatter) was 9 us of a 19 us launch.  So the
  // gathers are issued together (unrolled, they are one instruction each) and everything else is a rolled loop
  // over the nsel live candidates.
  const int nsel = sd.n_sel, G = sd.n_groups;
  fstamp(4);
  for (int b = tid; b < B; b += 256) {
    const float* __restrict__ row = med + (size_t)b * F;
    float* __restrict__ ar = red[tid];
    float* __restrict__ wr = wred[tid];
    float av[FTN_KMAX];
#pragma unroll
    for (int j = 0; j < FTN_KMAX; ++j) av[j] = row[j < nsel ? sd.sel_freq[j] : 0];   // clamped, unconditional
#pragma unroll
    for (int j = 0; j < FTN_KMAX; ++j) { ar[j] = av[j]; wr[j] = 0.f; }
    float mx = -INFINITY;
#pragma unroll 1
    for (int j = 0; j < FTN_KMAX; ++j) {
      const float v = j < nsel ? rnd_act(ar[j], act_dtype) : 0.f;
      ar[j] = v;
      amps[(size_t)b * FTN_KMAX + j] = v;
      if (j < nsel && sd.sel_group[j] >= 0) mx = fmaxf(mx, v);
    }
    float den = 0.f;
#pragma unroll 1
    for (int j = 0; j < nsel; ++j)
      if (sd.sel_group[j] >= 0) { const float e = expf(ar[j] - mx); ar[j] = e; den += e; }
#pragma unroll 1
    for (int j = 0; j < nsel; ++j) {
      const int g = sd.sel_group[j];
      if (g >= 0) wr[g] = rnd_act(wr[g] + rnd_act(ar[j] / den, act_dtype), act_dtype);
    }
#pragma unroll 1
    for (int g = 0; g < FTN_KMAX; ++g) wts[(size_t)b * FTN_KMAX + g] = (g < G) ? wr[g] : 0.f;
  }
  fstamp(5);
}

// TimesBlock conv path on gfx950 (reference models/timesnet.py:955-1101 with the
// inception stack of :560-654, 744-762), v1: one launch per stage, all groups of
// a block call in each launch, intermediates in a caller-provided workspace.
//
// Pixel space.  For group g (period p, pad, cycles) every batch row owns
// P_g = L + pad_g grid pixels t = cycle*p + phase; the period fold of the
// reference (:1041-1046) is exactly this re-indexing of the zero-extended
// window, so no fold pass exists here.  Pixels of all groups are laid out flat,
// group-major: n = B*px_off[g] + b*P_g + t, N = B*total_px.  Activation buffers
// are [N][channels] with channels padded to a multiple of 16.
//
// All contractions run on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains): rows =
// output channels (A = weights, row-major [out][in]), columns = 16 pixels
// (B = activations).  With the k-order "element e of lane group q is input
// channel 16s+4q+e", the A fragment is one contiguous float4 of the weight row,
// the B fragment one contiguous float4 of the pixel row, and — because result
// register r of lane (j,q) is output channel 4q+r of pixel j — an accumulator
// tile is directly the B fragment of the next 1x1 layer (no LDS, no shuffles).
//
// Stages (bottleneck mode; SURVEY finding 5 folds proj∘branch[-1] into w_out):
//   A  k_pw        a  = W_in1 x + b                  (C -> nbr*mid)
//   B  k_conv      m  = conv_k(a_k) + b              (per branch mid -> mid, zero pad)
//   C  k_mlp       g  = act(act(W_out1 m + b) + res1(x));  a' = W_in2 g + b;  r = res2(g) - x
//   D  k_conv      m' = conv_k(a'_k) + b
//   E+F k_out      y  = x + sum_g w[b,g] * (act(W_out2 m'_g + b) + r_g)[:L]
// Single-conv mode (ratio 1) replaces A by a zero-padded copy of x, B/D by one
// merged conv with proj folded in, and E by an elementwise epilogue.
#include <stdlib.h>
#include "ftn_common.h"
#include "ftn_mlp.h"
#include "ftn_out.h"

#define NPXU 4  // 16-pixel units per wave in the pointwise / conv kernels

// Diagnostic cycle stamps (ftn_debug_stamps): when a buffer is registered, thread 0 of
// every k_conv / k_mlp workgroup stores s_memtime at its phase boundaries there.  The
// buffer is read by nothing else; production runs leave the pointer null.
static unsigned long long* g_stamp_buf = nullptr;
static size_t g_stamp_cap = 0;
static int g_stamp_which = 0;  // 1: k_conv, 2: k_mlp
static const bool g_conv_generic = [] { const char* e = getenv("FTN_CONV_GENERIC"); return e != nullptr && e[0] == '1'; }();  // experiment switch
static const bool g_mlp_split = [] { const char* e = getenv("FTN_MLP_SPLIT"); return e == nullptr || e[0] != '0'; }();   // split refill (default on)
static const bool g_r_keeps_x = [] { const char* e = getenv("FTN_R_KEEPS_X"); return e == nullptr || e[0] != '0'; }();   // stage C leaves x inside R (default on)
static const bool g_mlp_pfd2 = [] { const char* e = getenv("FTN_MLP_PFD"); return e != nullptr && e[0] == '2'; }();   // experiment: fragment reads two steps ahead
static const bool g_mlp_w16 = [] { const char* e = getenv("FTN_MLP_W16"); return e != nullptr && e[0] == '1'; }();   // experiment: 16-wave double-buffered k_mlp_bf_u1
static const bool g_mlp_w4 = [] { const char* e = getenv("FTN_MLP_W4"); return e != nullptr && e[0] == '1'; }();     // experiment: 4-wave workgroups, three per CU
static const bool g_mlp_u1 = [] { const char* e = getenv("FTN_MLP_U1"); return e == nullptr || e[0] != '0'; }();    // 0: the two-unit k_mlp_bf
__device__ __forceinline__ void stamp(unsigned long long* buf, size_t cap, size_t wg, int slot) {
  if (buf != nullptr && threadIdx.x == 0 && (wg * 8 + slot) < cap) buf[wg * 8 + slot] = __builtin_amdgcn_s_memtime();
}

struct PwArgs {
  const float* x;        // [B][L][C] (when XIN)
  const float* in;       // [N][KIN]  (when !XIN)
  const float* W;        // [16*n_ot][KIN]
  const float* bias;     // [16*n_ot]
  float* out;            // [N][OUTC]
  float* R;              // [N][RC] (EPI 1: read, add, store back)
  const FtnDesc* desc;
  int B, L, C, KIN, n_ot, OUTC, RC;
  // stage A also publishes the sanitised descriptor copy every later launch reads (guard_desc below); it does
  // not need the descriptor itself, so this costs no extra launch
  const FtnDesc* guard_src; FtnDesc* guard_dst; int guard_groups, guard_px;
  int* range_flag;       // EPI 3 (f16x2 pieces): set when an output leaves the fp16 range; may be null
};

// Copies the caller's descriptor to the head of the workspace; a descriptor that exceeds the bounds the
// workspace and the grids were sized for (more groups than max_groups, more pixels than px_bound) is
// replaced by an empty one, which makes the call the identity y = x instead of a write past a buffer.
__device__ __forceinline__ void guard_desc(const FtnDesc* __restrict__ src, FtnDesc* __restrict__ dst, int max_groups,
                                           int px_bound) {
  const int* s = (const int*)src;
  int* d = (int*)dst;
  const bool bad = src->n_groups < 0 || src->n_groups > max_groups || src->total_px < 0 || src->total_px > px_bound;
  for (int e = threadIdx.x; e < (int)(sizeof(FtnDesc) / 4); e += blockDim.x) d[e] = bad ? 0 : s[e];
  if (threadIdx.x == 0) d[sizeof(FtnDesc) / 4] = bad ? 1 : 0;
}

// Workgroup barrier that waits only until at most `keep` of this wave's vector-memory operations are still in
// flight (vmcnt retires in issue order on gfx9, loads and stores alike): the conv row loop issues
//   LDS-DMA of row b+1 | compute row b | output stores of row b
// and the next barrier needs the DMA, not the stores - a plain __syncthreads() waits vmcnt(0), i.e. one HBM write
// round trip per batch row.  `keep` is wave-uniform.
__device__ __forceinline__ void barrier_keep_vm(int keep) {
  switch (keep) {
    case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
}

// ---------------------------------------------------------------- pixel decode
struct Px {
  int n;          // clamped flat pixel index
  bool ok;        // lane holds a real pixel
  const float* xrow;  // &x[b][t][0] or nullptr for t >= L (live zero pixel, :1017)
};

__device__ __forceinline__ Px decode_px(const FtnDesc* __restrict__ d, const float* __restrict__ x, int B, int L,
                                        int C, int n, int N) {
  Px p;
  p.ok = n < N;
  p.n = p.ok ? n : N - 1;
  const int G = d->n_groups;
  // the whole prefix table in one batch of scalar loads (fixed trip count), then a select chain:
  // a data-dependent loop here costs one scalar-memory round trip per group on every workgroup's
  // critical path
  int off[FTN_KMAX + 1];
#pragma unroll
  for (int i = 0; i <= FTN_KMAX; ++i) off[i] = d->g_px_off[i];
  int lo = 0, hi = off[1];
#pragma unroll
  for (int gg = 1; gg < FTN_KMAX; ++gg) {
    const bool in = gg < G && p.n >= B * off[gg];
    lo = in ? off[gg] : lo;
    hi = in ? off[gg + 1] : hi;
  }
  const int P = hi - lo;
  const int rem = p.n - B * lo;
  const int b = rem / P, t = rem - b * P;
  p.xrow = (t < L) ? x + ((size_t)b * L + t) * C : nullptr;
  return p;
}

// The same for the 16 CONSECUTIVE pixels n0 + j of one MFMA pixel unit (n0 wave-uniform).  They almost always lie
// in one period group, which a scalar walk over the (at most 16) prefix sums finds: the per-lane part is then a
// division by a uniform P.  decode_px's per-lane 16-way select chain is ~150 VALU + SALU instructions, a tenth of
// what a stage-C wave executes for its unit.  Units that straddle a group boundary take the general path.
__device__ __forceinline__ Px decode_px16(const FtnDesc* __restrict__ d, const float* __restrict__ x, int B, int L,
                                          int C, int n0, int j, int N) {
  const int G = d->n_groups;
  const int first = __builtin_amdgcn_readfirstlane(n0);
  const int last = first + 15 < N ? first + 15 : N - 1;
  int off[FTN_KMAX + 1];                                          // one batch of scalar loads, then SALU selects
#pragma unroll
  for (int i = 0; i <= FTN_KMAX; ++i) off[i] = d->g_px_off[i];
  int lo = 0, hi = off[1];
#pragma unroll
  for (int gg = 1; gg < FTN_KMAX; ++gg) {
    const bool in = gg < G && first >= B * off[gg];
    lo = in ? off[gg] : lo;
    hi = in ? off[gg + 1] : hi;
  }
  if (first < N && last < B * hi) {                              // whole unit inside one group (uniform branch)
    Px p;
    const int n = first + j;
    p.ok = n < N;
    p.n = p.ok ? n : N - 1;
    const int P = hi - lo;
    const int rem = p.n - B * lo;
    const int b = rem / P, t = rem - b * P;
    p.xrow = (t < L) ? x + ((size_t)b * L + t) * C : nullptr;
    return p;
  }
  return decode_px(d, x, B, L, C, n0 + j, N);
}

// ---------------------------------------------------------------- stage A and the generic pointwise layers
// out[n][:] = epilogue( W in[n][:] + b )  on exact fp32 MFMA, any number of output tiles and any K.
//   XIN 0: in = buffer [N][KIN] per grid pixel      XIN 1: in = x by window row (stage A, see below)
//   XIN 2: in = x by grid pixel (zero rows for the live pad pixels t >= L)
//   EPI 0: store fp32      EPI 2 / 3: store three bf16 / two fp16 pieces (input of the split conv engines)
//   EPI 4: store v - x[n]  (r = res2(g) - x)      EPI 5: store act(v)      EPI 6: out = act(v + out)  (in place)
// Stage A always runs here; EPI 4-6 with XIN 0 / 2 form the generic stage C for widths beyond the fused
// kernels' limits (more than 16 output tiles, or a hidden chunk's fragments not fitting LDS twice).
template <int ACT, int XIN, bool XVEC, int EPI>
__device__ __forceinline__ void pw_body(const PwArgs& a, const int bid) {
  // XIN 1 (stage A): a = W_in1 x + b depends on (b, t) only, not on the period group, so it is computed once per
  // window position - rows n = b*L + t of `out` - plus ONE pad row n = B*L for the live zero pixels t >= L of
  // every grid (x = 0 there, :1017, so a = bias).  The conv stage folds these rows into its period grids while
  // staging (ConvArgs.bt_L), which is the reference's reshape (:1041-1046) done by index arithmetic.
  const FtnDesc* __restrict__ d = a.desc;
  if (XIN == 1 && bid == 0 && a.guard_dst != nullptr) guard_desc(a.guard_src, a.guard_dst, a.guard_groups, a.guard_px);
  const int N = XIN == 1 ? a.B * a.L + 1 : a.B * d->total_px;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int n0 = (bid * 4 + wave) * (16 * NPXU);
  if (n0 >= N) return;
  Px px[NPXU];
#pragma unroll
  for (int u = 0; u < NPXU; ++u) {
    if (XIN == 1) {
      const int n = n0 + 16 * u + j;
      px[u].ok = n < N;
      px[u].n = px[u].ok ? n : N - 1;
      px[u].xrow = px[u].n < N - 1 ? a.x + (size_t)px[u].n * a.C : nullptr;
    } else {
      px[u] = decode_px16(d, a.x, a.B, a.L, a.C, n0 + 16 * u, j, N);
    }
  }
  const int KIN = a.KIN;
  bool range_bad = false;
  for (int og = 0; og < a.n_ot; og += 4) {
    f4 acc[4][NPXU];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f4 bv = {0.f, 0.f, 0.f, 0.f};
      if (og + o < a.n_ot) bv = *(const f4*)(a.bias + 16 * (og + o) + 4 * q);
#pragma unroll
      for (int u = 0; u < NPXU; ++u) acc[o][u] = bv;
    }
    // operands of K step s + 16 are requested before the products of step s (every load used to sit right in front of
    // its MFMAs: at d_model 128 stage A ran at a fifth of the fp32 pipe's rate)
    auto load_step = [&](int s, f4 (&bf)[NPXU], f4 (&af)[4]) {
#pragma unroll
      for (int u = 0; u < NPXU; ++u) {
        if (XIN != 0) bf[u] = load_x4<XVEC>(px[u].xrow, s + 4 * q, a.C);
        else bf[u] = *(const f4*)(a.in + (size_t)px[u].n * KIN + s + 4 * q);
      }
#pragma unroll
      for (int o = 0; o < 4; ++o)
        af[o] = og + o < a.n_ot ? *(const f4*)(a.W + (size_t)(16 * (og + o) + j) * KIN + s + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    };
    f4 bfc[NPXU], afc[4];
    load_step(0, bfc, afc);
    for (int s = 0; s < KIN; s += 16) {
      f4 bfn[NPXU], afn[4];
      const bool more = s + 16 < KIN;
      if (more) load_step(s + 16, bfn, afn);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (og + o < a.n_ot) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < NPXU; ++u) acc[o][u] = mfma16(afc[o][e], bfc[u][e], acc[o][u]);
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < NPXU; ++u) bfc[u] = bfn[u];
#pragma unroll
        for (int o = 0; o < 4; ++o) afc[o] = afn[o];
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (og + o < a.n_ot) {
#pragma unroll
        for (int u = 0; u < NPXU; ++u) {
          if (!px[u].ok) continue;
          const int ch = 16 * (og + o) + 4 * q;
          float* op = a.out + (size_t)px[u].n * a.OUTC + ch;
          if (EPI == 0) {
            *(f4*)op = acc[o][u];
          } else if (EPI == 2 || EPI == 3) {   // bf16x3 (P3) / f16x2 (H2) pieces: input of the split conv engines
            constexpr int NSP = EPI == 3 ? 2 : 3;
            if (EPI == 3) range_bad |= h2_bad4(acc[o][u]);
            store_px<NSP>((__bf16*)a.out + ((size_t)px[u].n * (a.OUTC >> 4) + (og + o)) * PxFmt<NSP>::ELEMS, q, acc[o][u]);
          } else if (EPI == 4) {
            *(f4*)op = acc[o][u] - load_x4<XVEC>(px[u].xrow, ch, a.C);
          } else if (EPI == 5) {
            *(f4*)op = act4<ACT>(acc[o][u]);
          } else {
            *(f4*)op = act4<ACT>(acc[o][u] + *(const f4*)op);
          }
        }
      }
    }
  }
  if (EPI == 3) raise_range_flag(a.range_flag, range_bad);
}

template <int ACT, int XIN, bool XVEC, int EPI>
__global__ __launch_bounds__(256) void k_pw(PwArgs a) { pw_body<ACT, XIN, XVEC, EPI>(a, (int)blockIdx.x); }

// Stage A has no use for the selector's result, and the selector ends in a one-workgroup kernel (k_finalize, ~17 us
// of serial latency with 255 CUs idle): this launch runs both - workgroup 0 is k_finalize (and then publishes the
// sanitised descriptor copy at the head of the workspace), workgroups 1.. are stage A - so stage A's ~22 us
// disappear behind the selector's tail (ftn_period_finalize_stage_a).
#include "ftn_finalize.h"
void ftn_xch_fill(const FtnExchange* x, int F, FinalizeArgs* fa);   // selector.hip
// part: 0 = both (workgroup 0 finalizes, the others run stage A), 1 = stage A only (a sharded batch runs it while
// the partial sums are exchanged), 2 = finalize + descriptor copy only (one workgroup, after that exchange)
template <int ACT, bool XVEC, int EPI>
__global__ __launch_bounds__(256) void k_finalize_pw(FinalizeArgs fa, PwArgs pa, int part) {
  if (part != 1 && blockIdx.x == 0) {
    finalize_body(fa);
    __syncthreads();
    guard_desc(fa.desc, pa.guard_dst, pa.guard_groups, pa.guard_px);
  } else if (part != 2) {
    PwArgs q = pa;
    q.guard_dst = nullptr;
    pw_body<ACT, 1, XVEC, EPI>(q, (int)blockIdx.x - (part == 0 ? 1 : 0));
  }
}

// Elementwise pieces of the generic stage C when a res_proj is the identity (d_ff == d_model):
//   mode 0: g = act(g + x)      mode 1: r = g - x      (rows = grid pixels, CH = FP = CP channels)
template <int ACT, bool XVEC>
__global__ void k_ew_ident(const float* __restrict__ x, float* __restrict__ g, float* __restrict__ r,
                           const FtnDesc* __restrict__ d, int B, int L, int C, int CH, int mode) {
  const int N = B * d->total_px;
  const int cq = CH >> 2;
  const long long total = (long long)N * cq;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(e / cq), c = (int)(e - (long long)n * cq) * 4;
    const Px px = decode_px(d, x, B, L, C, n, N);
    const f4 xv = load_x4<XVEC>(px.xrow, c, C);
    f4* gp = (f4*)(g + (size_t)n * CH + c);
    if (mode == 0) *gp = act4<ACT>(*gp + xv);
    else *(f4*)(r + (size_t)n * CH + c) = *gp - xv;
  }
}

// ---------------------------------------------------------------- stage C
// g  = act(act(W_out1 m + b) + W_res1 x + b)       hidden, d_ff channels
// a' = W_in2 g + b ;  r = W_res2 g + b - x         the stacked output projection W_c
// One wave owns NPX 16-pixel units; the hidden dimension is walked in chunks of
// 64 channels (4 MFMA row tiles).  Per chunk the workgroup stages that chunk's
// weight fragments (lane-linear, pre-packed on the host: FtnPlan.w_cfrag) in LDS
// once for its 4 waves; layer 1 and the residual read their B fragments (m, x
// rows) straight from global/L2 with a one-step-ahead prefetch; the hidden
// accumulators then ARE the B fragments of the output projection, which
// accumulates across chunks in registers.  Nothing hidden-sized touches memory.
#define MLP_HT 2   // hidden row tiles per chunk (pack.py CHUNK_TILES)

struct MlpArgs {
  const float* x;
  const float* m;        // [N][KM] conv output of block 1
  const float* cfrag;    // [n_hchunks][cfrag_per_chunk][256]
  const float* bo;       // [FP] bias of W_out1 (null when z = m)
  const float* br;       // [FP] bias of W_res1 (null when res = x)
  const float* bc;       // [16*n_ot]
  float* outA;           // [N][AC]: first n_oa output tiles (a'), may be null
  float* outG;           // [N][FP]: hidden store (single-conv mode), may be null
  float* outR;           // [N][CP]: r = res2(g) - x
  const FtnDesc* desc;
  int B, L, C, CP, FP, KM, AC;
  int nKM;               // K chunks of layer 1 (KM/16), 0: z = m (KM == FP)
  int nCP;               // K chunks of the residual (CP/16), 0: res = x (FP == CP)
  int n_oa;              // output tiles that go to outA
  int n_ot;              // total output tiles (n_oa + CP/16 when res2 is a conv)
  int res2_ident;        // 1: r = g - x  (FP == CP), taken from the hidden tiles
  int n_hchunks, cfrag_per_chunk;
  int outA_p3;           // 1: write a' as three bf16 pieces (bf16x3 conv engine), 2: two fp16 pieces (f16x2)
  unsigned long long* dbg; size_t dbg_cap;
};

// PRE: the B fragments of layer 1 (m rows, <= 3 K-chunks) and of the residual (x rows, <= 4
// K-chunks) do not depend on the hidden chunk, so they are loaded ONCE into registers and the
// chunk loop touches no global memory besides the weight DMA.
#define MLP_PRE_KM 3
#define MLP_PRE_CP 4
template <int ACT, bool XVEC, int NPX, int OTM, bool EXACT, bool PRE>
__global__ __launch_bounds__(256, (NPX >= 3 ? 2 : 1)) void k_mlp(MlpArgs a) {
  constexpr int HT = MLP_HT;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const FtnDesc* __restrict__ d = a.desc;
  const int N = a.B * d->total_px;
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 0);
  if ((int)(blockIdx.x * 4 * 16 * NPX) >= N) return;            // whole workgroup beyond the live pixels
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * (16 * NPX);
  const bool active = n0 < N;                                    // wave-uniform; idle waves still stage + sync
  // Weight fragments reach LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave instruction)
  // into two buffers: chunk hc+1 is requested right before the output-projection MFMAs of
  // chunk hc (the longest phase, no other memory traffic) and has landed by the barrier that
  // ends the chunk, so staging is off the critical path.
  const int bufsz = a.cfrag_per_chunk * 256;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto dma_chunk = [&](int hc, int buf) {
    const float* __restrict__ src = a.cfrag + (size_t)hc * bufsz;
    for (int piece = wv; piece < a.cfrag_per_chunk; piece += 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)piece * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(wl + (size_t)buf * bufsz + (size_t)piece * 256),
                                       16, 0, 0);
  };
  dma_chunk(0, 0);                                               // lands while the pixels are decoded
  Px px[NPX];
#pragma unroll
  for (int u = 0; u < NPX; ++u) px[u] = decode_px16(d, a.x, a.B, a.L, a.C, n0 + 16 * u, j, N);
  const int FP = a.FP, KM = a.KM, CP = a.CP;
  const int nht = FP >> 4;
  const int nKM = a.nKM, nCP = a.nCP, n_ot = EXACT ? OTM : a.n_ot;
  const int offWr = HT * nKM, offWc = offWr + HT * nCP;
  f4 mpre[PRE ? MLP_PRE_KM : 1][NPX], xpre[PRE ? MLP_PRE_CP : 1][NPX];
  if (PRE) {
#pragma unroll
    for (int s = 0; s < MLP_PRE_KM; ++s)
#pragma unroll
      for (int u = 0; u < NPX; ++u)
        mpre[s][u] = s < nKM ? *(const f4*)(a.m + (size_t)px[u].n * KM + 16 * s + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < MLP_PRE_CP; ++s)
#pragma unroll
      for (int u = 0; u < NPX; ++u)
        xpre[s][u] = s < nCP ? load_x4<XVEC>(px[u].xrow, 16 * s + 4 * q, a.C) : f4{0.f, 0.f, 0.f, 0.f};
  }
  f4 oacc[OTM][NPX];
#pragma unroll
  for (int o = 0; o < OTM; ++o) {
    f4 bv = {0.f, 0.f, 0.f, 0.f};
    if (o < n_ot) bv = *(const f4*)(a.bc + 16 * o + 4 * q);
#pragma unroll
    for (int u = 0; u < NPX; ++u) oacc[o][u] = bv;
  }

  __syncthreads();
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 1);
  for (int hc = 0; hc < a.n_hchunks; ++hc) {
    const float* __restrict__ wlane = wl + (size_t)(hc & 1) * bufsz + lane * 4;
    if (hc == 1) stamp(a.dbg, a.dbg_cap, blockIdx.x, 2);
    if (active) {
    f4 h[HT][NPX];
    // ---- z = W_out1 m + b   (or z = m)
    if (nKM > 0) {
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        f4 bv = {0.f, 0.f, 0.f, 0.f};
        if (hc * HT + t < nht) bv = *(const f4*)(a.bo + 16 * (hc * HT + t) + 4 * q);
#pragma unroll
        for (int u = 0; u < NPX; ++u) h[t][u] = bv;
      }
      if (PRE) {
#pragma unroll
        for (int s = 0; s < MLP_PRE_KM; ++s) {
          if (s < nKM) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const f4 af = *(const f4*)(wlane + (t * nKM + s) * 256);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < NPX; ++u) h[t][u] = mfma16(af[e], mpre[s][u][e], h[t][u]);
            }
          }
        }
      } else {
        f4 bcur[NPX];
#pragma unroll
        for (int u = 0; u < NPX; ++u) bcur[u] = *(const f4*)(a.m + (size_t)px[u].n * KM + 4 * q);
        for (int s = 0; s < nKM; ++s) {
          const int sn = s + 1 < nKM ? s + 1 : s;
          f4 bnxt[NPX];
#pragma unroll
          for (int u = 0; u < NPX; ++u) bnxt[u] = *(const f4*)(a.m + (size_t)px[u].n * KM + 16 * sn + 4 * q);
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const f4 af = *(const f4*)(wlane + (t * nKM + s) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int u = 0; u < NPX; ++u) h[t][u] = mfma16(af[e], bcur[u][e], h[t][u]);
          }
#pragma unroll
          for (int u = 0; u < NPX; ++u) bcur[u] = bnxt[u];
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (hc * HT + t < nht) v = *(const f4*)(a.m + (size_t)px[u].n * KM + 16 * (hc * HT + t) + 4 * q);
          h[t][u] = v;
        }
    }
    // ---- act, then + res1(x)                      (:652-654)
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[t][u] = act4<ACT>(h[t][u]);
    if (nCP > 0) {
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        if (hc * HT + t < nht) {
          const f4 bv = *(const f4*)(a.br + 16 * (hc * HT + t) + 4 * q);
#pragma unroll
          for (int u = 0; u < NPX; ++u) h[t][u] += bv;
        }
      }
      if (PRE) {
#pragma unroll
        for (int s = 0; s < MLP_PRE_CP; ++s) {
          if (s < nCP) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const f4 af = *(const f4*)(wlane + (offWr + t * nCP + s) * 256);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < NPX; ++u) h[t][u] = mfma16(af[e], xpre[s][u][e], h[t][u]);
            }
          }
        }
      } else {
        f4 bcur[NPX];
#pragma unroll
        for (int u = 0; u < NPX; ++u) bcur[u] = load_x4<XVEC>(px[u].xrow, 4 * q, a.C);
        for (int s = 0; s < nCP; ++s) {
          const int sn = s + 1 < nCP ? s + 1 : s;
          f4 bnxt[NPX];
#pragma unroll
          for (int u = 0; u < NPX; ++u) bnxt[u] = load_x4<XVEC>(px[u].xrow, 16 * sn + 4 * q, a.C);
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const f4 af = *(const f4*)(wlane + (offWr + t * nCP + s) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int u = 0; u < NPX; ++u) h[t][u] = mfma16(af[e], bcur[u][e], h[t][u]);
          }
#pragma unroll
          for (int u = 0; u < NPX; ++u) bcur[u] = bnxt[u];
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int u = 0; u < NPX; ++u)
          if (hc * HT + t < nht) h[t][u] += load_x4<XVEC>(px[u].xrow, 16 * (hc * HT + t) + 4 * q, a.C);
    }
    // ---- mid activation                           (:753)
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[t][u] = act4<ACT>(h[t][u]);
    // ---- optional stores taken straight from the hidden tiles
    if (a.outG != nullptr || a.res2_ident) {
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        if (hc * HT + t < nht) {
#pragma unroll
          for (int u = 0; u < NPX; ++u) {
            if (!px[u].ok) continue;
            const int ch = 16 * (hc * HT + t) + 4 * q;
            if (a.outG != nullptr) *(f4*)(a.outG + (size_t)px[u].n * FP + ch) = h[t][u];
            if (a.res2_ident)
              *(f4*)(a.outR + (size_t)px[u].n * CP + ch) = h[t][u] - load_x4<XVEC>(px[u].xrow, ch, a.C);
          }
        }
      }
    }
    // ---- request the next chunk's fragments, then the output projection: the hidden
    //      accumulators ARE the B fragments
    if (hc + 1 < a.n_hchunks) dma_chunk(hc + 1, (hc + 1) & 1);
#pragma unroll
    for (int t = 0; t < HT; ++t) {
#pragma unroll
      for (int o = 0; o < OTM; ++o) {
        if (EXACT || o < n_ot) {
          const f4 af = *(const f4*)(wlane + (offWc + t * n_ot + o) * 256);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < NPX; ++u) oacc[o][u] = mfma16(af[e], h[t][u][e], oacc[o][u]);
        }
      }
    }
    } else if (hc + 1 < a.n_hchunks) {
      dma_chunk(hc + 1, (hc + 1) & 1);
    }
    __syncthreads();   // everyone is done with this buffer and (vmcnt(0)) the next one has landed
  }
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 3);
  if (!active) return;
  // ---- epilogue: a' tiles, then r = res2(g) - x tiles
#pragma unroll
  for (int o = 0; o < OTM; ++o) {
    if (EXACT || o < n_ot) {
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        if (!px[u].ok) continue;
        if (o < a.n_oa) {
          if (a.outA_p3 == 2) store_h2((__bf16*)a.outA + ((size_t)px[u].n * (a.AC >> 4) + o) * 32, q, oacc[o][u]);
          else if (a.outA_p3) store_p3((__bf16*)a.outA + ((size_t)px[u].n * (a.AC >> 4) + o) * 48, q, oacc[o][u]);
          else *(f4*)(a.outA + (size_t)px[u].n * a.AC + 16 * o + 4 * q) = oacc[o][u];
        } else {
          const int ch = 16 * (o - a.n_oa) + 4 * q;
          *(f4*)(a.outR + (size_t)px[u].n * CP + ch) = oacc[o][u] - load_x4<XVEC>(px[u].xrow, ch, a.C);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- stage C, bf16x3 engine
// The same register-chained pointwise stage on the bf16 matrix pipe: every operand is three
// bf16 pieces and every K=32 slab is the six-product chain of k_conv_bf.  Layer-1 inputs
// (m: P3 rows written by the conv; x: split on load) are preloaded once; after the two
// GELUs the fp32 hidden accumulators of a 32-channel chunk (two row tiles) are split into
// pieces in registers and become the B operand of the output projection (the host packs
// the projection's K order to match the accumulator lane map).  512-thread workgroups
// (256 pixels) share each chunk's weight fragments, DMA-staged and double-buffered.
// NW = waves per workgroup.  NW = 8: one 256-pixel workgroup per CU, chunk weights double-buffered.
// NW = 4: 128-pixel workgroups with a single weight buffer (2 x 66 KB would not fit twice), two per CU:
// VALU and MFMA work of a SIMD serialise on gfx950 (tools/ubench/mfma_valu.hip), so what a second
// workgroup buys is cover for the first one's prologue loads, chunk barriers, DMA waits and stores.
template <int ACT, bool XVEC, int OTM, bool EXACT, int NS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_mlp_bf(MlpBfArgs a) {
  constexpr int NPX = 2;
  extern __shared__ __attribute__((aligned(16))) char wlb[];
  const FtnDesc* __restrict__ d = a.desc;
  const int N = a.B * d->total_px;
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 0);
  if ((int)(blockIdx.x * NW * 16 * NPX) >= N) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, qa = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int n0 = (blockIdx.x * NW + wave) * (16 * NPX);
  const bool active = n0 < N;
  const int bufsz = a.per_chunk * 3 * 1024;
  // fragments [f_lo, f_hi) of chunk hc -> the same slots of buffer `buf`
  auto dma_frags = [&](int hc, int buf, int f_lo, int f_hi) {
    const __bf16* __restrict__ src = a.cfrag + (size_t)hc * a.per_chunk * 3 * 512;
    for (int piece = 3 * f_lo + wv; piece < 3 * f_hi; piece += NW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(wlb + (size_t)buf * bufsz + (size_t)piece * 1024),
                                       16, 0, 0);
  };
  auto dma_chunk = [&](int hc, int buf) { dma_frags(hc, buf, 0, a.per_chunk); };
  // Prologue order matters: vmcnt retires in order, so whatever is issued before the pixel loads is
  // waited for with them.  Decode first (scalar loads only), then this wave's m / x rows, and only then
  // the chunk-0 weight DMA and the bias staging, which are not needed before the first barrier.
  Px px[NPX];
#pragma unroll
  for (int u = 0; u < NPX; ++u) px[u] = decode_px16(d, a.x, a.B, a.L, a.C, n0 + 16 * u, j, N);
  const int CP = a.CP;
  const int nsKM = a.nsKM, nsCP = a.nsCP, n_ot = EXACT ? OTM : a.n_ot;
  const int kmg = a.KM >> 4;                                  // 16-channel groups of m
  constexpr int NWP = PxFmt<NS>::NW;                          // weight pieces per fragment
  constexpr int PXE = PxFmt<NS>::ELEMS;                       // 16-bit elements per pixel and 16-channel group
  // B operands that do not depend on the hidden chunk
  bool range_bad = false;                                       // f16x2: a value left the fp16 range (ftn_common.h)
  bf8 mp[2][NPX][NS], xp[2][NPX][NS];
  f4 xraw[2][NPX][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
      const int grp = 2 * s + (qa >> 1);
      const __bf16* __restrict__ src = a.m + ((size_t)px[u].n * kmg + (grp < kmg ? grp : 0)) * PXE + (qa & 1) * 8;
#pragma unroll
      for (int pz = 0; pz < NS; ++pz) mp[s][u][pz] = *(const bf8*)(src + pz * 16);
      xraw[s][u][0] = s < nsCP ? load_x4<XVEC>(px[u].xrow, 32 * s + 8 * qa, a.C) : f4{0.f, 0.f, 0.f, 0.f};
      xraw[s][u][1] = s < nsCP ? load_x4<XVEC>(px[u].xrow, 32 * s + 8 * qa + 4, a.C) : f4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  dma_chunk(0, 0);
  // biases of both hidden layers, zero-padded to whole chunks, in LDS behind the weight buffers
  const int FPc = a.n_hchunks * 32;
  constexpr int NBUF = NW == 8 ? 2 : 1;
  float* __restrict__ bias_l = (float*)(wlb + NBUF * (size_t)bufsz);
  for (int i = threadIdx.x; i < 2 * FPc; i += NW * 64) {
    const int c = i < FPc ? i : i - FPc;
    bias_l[i] = c < a.FP ? (i < FPc ? a.bo[c] : a.br[c]) : 0.f;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
      const int grp = 2 * s + (qa >> 1);
      if (!(s < nsKM && grp < kmg)) {
#pragma unroll
        for (int pz = 0; pz < NS; ++pz)
#pragma unroll
          for (int e = 0; e < 8; ++e) mp[s][u][pz][e] = (__bf16)0.0f;
      }
      float xv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[e] = xraw[s][u][0][e]; xv[4 + e] = xraw[s][u][1][e]; }
      if (NS == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) range_bad |= h2_bad(xv[e]);
      }
      split_pieces<NS>(xv, xp[s][u]);
    }
  }
  f4 oacc[OTM][NPX];
#pragma unroll
  for (int o = 0; o < OTM; ++o) {
    f4 bv = {0.f, 0.f, 0.f, 0.f};
    if (o < n_ot) bv = *(const f4*)(a.bc + 16 * o + 4 * qa);
#pragma unroll
    for (int u = 0; u < NPX; ++u) oacc[o][u] = bv;
  }
  __syncthreads();
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 1);
  for (int hc = 0; hc < a.n_hchunks; ++hc) {
    if (hc == 1) stamp(a.dbg, a.dbg_cap, blockIdx.x, 2);
    const char* __restrict__ wl = wlb + (size_t)(NBUF == 2 ? (hc & 1) : 0) * bufsz + lane * 16;
    // request the next chunk right away: its buffer was last read before the barrier that
    // opened this chunk, and nothing below waits on vmcnt until the closing barrier.  The
    // biases come from LDS (staged once in the prologue), not from global memory.
    if (NBUF == 2 && hc + 1 < a.n_hchunks) dma_chunk(hc + 1, (hc + 1) & 1);
    f4 bo_t[2], br_t[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bo_t[t] = *(const f4*)(bias_l + 16 * (hc * 2 + t) + 4 * qa);
      br_t[t] = *(const f4*)(bias_l + FPc + 16 * (hc * 2 + t) + 4 * qa);
    }
    f4 h[2][NPX];
    bf8 hp[NPX][NS];
    bf8 fa[NWP], fb[NWP];
    auto ldfrag = [&](int f, bf8 (&ap)[NWP]) {
#pragma unroll
      for (int pz = 0; pz < NWP; ++pz) ap[pz] = *(const bf8*)(wl + (size_t)(f * 3 + pz) * 1024);
    };
    auto gelu_u = [&](int t, int u, bool addbr) {
      if (NS == 2) {                                           // undo / apply the weight prescales (see MlpBfArgs)
        h[t][u] = act4<ACT>(h[t][u] * (addbr ? a.inv_o : a.inv_r));
        if (addbr) h[t][u] = h[t][u] * a.sc_r + br_t[t];
      } else {
        h[t][u] = act4<ACT>(h[t][u]);
        if (addbr) h[t][u] += br_t[t];
      }
    };
    auto split_u = [&](int u) {
      float hv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { hv[e] = h[0][u][e]; hv[4 + e] = h[1][u][e]; }
      split_pieces<NS>(hv, hp[u]);
    };
    if (active) {
      // Statically scheduled chunk (KM <= 64, C <= 64: two K slabs each).  The 8 + 2*n_ot weight
      // fragments are walked in LDS order with a one-step-ahead register prefetch, and the
      // VALU work (GELU, piece splitting) of one row tile / pixel unit is placed in the same
      // scheduling region as MFMAs that do not depend on it, so both pipes stay busy although
      // the two waves of a SIMD run in lockstep between chunk barriers:
      //   f0 f1: L1(t0) | f2 f3: L1(t1) + G1(t0) | f4 f5: R(t0) + G1(t1) | f6 f7: R(t1) + G2(t0)
      //   G2(t1,u0) split(u0) | L2(u0) + G2(t1,u1) split(u1) | L2(u1)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NPX; ++u) h[t][u] = bo_t[t];
      ldfrag(0, fa);
      // ---- L1(t0)
      ldfrag(1, fb); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[0][u] = chain_bf<NS>(fa, mp[0][u], h[0][u]);
      ldfrag(2, fa); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[0][u] = chain_bf<NS>(fb, mp[1][u], h[0][u]);
      // ---- L1(t1) + G1(t0)
      ldfrag(3, fb); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[1][u] = chain_bf<NS>(fa, mp[0][u], h[1][u]);
      gelu_u(0, 0, true);
      ldfrag(4, fa); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[1][u] = chain_bf<NS>(fb, mp[1][u], h[1][u]);
      gelu_u(0, 1, true);
      // ---- R(t0) + G1(t1)
      ldfrag(5, fb); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[0][u] = chain_bf<NS>(fa, xp[0][u], h[0][u]);
      gelu_u(1, 0, true);
      ldfrag(6, fa); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[0][u] = chain_bf<NS>(fb, xp[1][u], h[0][u]);
      gelu_u(1, 1, true);
      // ---- R(t1) + G2(t0)
      ldfrag(7, fb); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[1][u] = chain_bf<NS>(fa, xp[0][u], h[1][u]);
      gelu_u(0, 0, false);
      ldfrag(8, fa); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NPX; ++u) h[1][u] = chain_bf<NS>(fb, xp[1][u], h[1][u]);
      gelu_u(0, 1, false);
      __builtin_amdgcn_sched_barrier(0);
      // ---- G2(t1,u0), split(u0)   (the only VALU stretch without an MFMA partner)
      gelu_u(1, 0, false);
      split_u(0);
      __builtin_amdgcn_sched_barrier(0);

      // ---- L2(u0) + G2(t1,u1), split(u1): fragments f = 8 .. 8+n_ot-1, ping-pong fa/fb
#pragma unroll
      for (int o = 0; o < OTM; ++o) {
        if (EXACT || o < n_ot) {
          if (o & 1) { ldfrag(8 + (o + 1 < n_ot ? o + 1 : 0), fa); __builtin_amdgcn_sched_barrier(0); oacc[o][0] = chain_bf<NS>(fb, hp[0], oacc[o][0]); }
          else       { ldfrag(8 + (o + 1 < n_ot ? o + 1 : 0), fb); __builtin_amdgcn_sched_barrier(0); oacc[o][0] = chain_bf<NS>(fa, hp[0], oacc[o][0]); }
          if (o == 0) gelu_u(1, 1, false);
          if (o == 1 || (n_ot == 1 && o == 0)) split_u(1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- L2(u1): the wrap-around prefetch above left fragment 8 in the next register set
#pragma unroll
      for (int o = 0; o < OTM; ++o) {
        if (EXACT || o < n_ot) {
          const bool odd = ((n_ot + o) & 1) != 0;
          if (odd) { ldfrag(8 + (o + 1 < n_ot ? o + 1 : o), fa); __builtin_amdgcn_sched_barrier(0); oacc[o][1] = chain_bf<NS>(fb, hp[1], oacc[o][1]); }
          else     { ldfrag(8 + (o + 1 < n_ot ? o + 1 : o), fb); __builtin_amdgcn_sched_barrier(0); oacc[o][1] = chain_bf<NS>(fa, hp[1], oacc[o][1]); }
        }
      }
    }
    if (hc == 1) stamp(a.dbg, a.dbg_cap, blockIdx.x, 7);
    if (hc == 1 && a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(a.dbg, a.dbg_cap, blockIdx.x, 5); }
    __syncthreads();
    if (hc == 1) stamp(a.dbg, a.dbg_cap, blockIdx.x, 4);
    if (NBUF == 1 && hc + 1 < a.n_hchunks) {
      // single buffer: refill once every wave has left the chunk.  (Refilling in halves behind a
      // mid-chunk barrier hides the DMA but costs more in barrier skew than it saves: measured +5 %.)
      dma_chunk(hc + 1, 0);
      __syncthreads();
    }
    if (hc == 1) stamp(a.dbg, a.dbg_cap, blockIdx.x, 6);
  }
  stamp(a.dbg, a.dbg_cap, blockIdx.x, 3);
  if (!active) { if (NS == 2) raise_range_flag(a.range_flag, range_bad); return; }
#pragma unroll
  for (int o = 0; o < OTM; ++o) {
    if (EXACT || o < n_ot) {
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        if (!px[u].ok) continue;
        if (o < a.n_oa) {
          const f4 av = NS == 2 ? oacc[o][u] * a.inv_a : oacc[o][u];
          if (NS == 2) range_bad |= h2_bad4(av);
          store_px<NS == 2 ? 2 : 3>(a.outA + ((size_t)px[u].n * (a.AC >> 4) + o) * PXE, qa, av);
        } else {
          const int ch = 16 * (o - a.n_oa) + 4 * qa;
          const f4 rv = NS == 2 ? oacc[o][u] * a.inv_r2 : oacc[o][u];
          *(f4*)(a.outR + (size_t)px[u].n * CP + ch) = rv - load_x4<XVEC>(px[u].xrow, ch, a.C);
        }
      }
    }
  }
  if (NS == 2) raise_range_flag(a.range_flag, range_bad);
}

// ---------------------------------------------------------------- stage C, bf16x3 engine, d_model 128
// The same chain for the reference's default pipeline shape (configs/default.yaml: d_model 128, d_ff 512,
// three kernels, ratio 4): nbr*mid = 96 = three K slabs of layer 1, d_model = 128 = four slabs of the
// residual, 6 + 8 = 14 output tiles.  With two 16-pixel units per wave that needs ~340 registers, so a wave
// owns ONE unit (oacc 56 + m pieces 36 + x pieces 48 registers) and the workgroup is 8 waves = 128 pixels;
// the 28 fragments of a chunk (84 KB) live in a single LDS buffer.  A fragment then feeds 6 MFMAs instead of
// 12 - which lands on the LDS read rate (tools/ubench/lds_patterns.hip) at about the time the SIMD needs
// for MFMA + GELU anyway.  The chunk is walked as one unrolled fragment sequence with a one-ahead prefetch.
// SKM / SCP = K=32 slabs of layer 1 / of the residual, OTM = output tiles: <3, 4, 14> is that shape; <2, 2, 7> is
// d_model 64 with three kernels of mid 16 (48 -> 64 K padding), where the smaller register footprint lets two
// 8-wave workgroups = four waves per SIMD share a CU (the two-unit k_mlp_bf above runs two).
// NWV = 8 (default): two single-buffered 128-pixel workgroups per CU.  The other launch shapes are kept as measured
// experiments: NWV = 16 (FTN_MLP_W16=1) = one 256-pixel workgroup per CU with the chunk weights
// DOUBLE-buffered - half the L2 -> LDS weight stream (1.2 GB per launch at the bench shape otherwise) and no exposed
// refill, yet slower (276 vs 248 us: sixteen waves coupled by one barrier run their MFMA and GELU phases in step);
// NWV = 4 (FTN_MLP_W4=1) = three 64-pixel workgroups per CU, twice the weight stream, the same time.
// PFD = fragment reads issued PFD steps ahead of their MFMAs (FTN_MLP_PFD=2: no gain - the waves do not wait on LDS).
template <int ACT, bool XVEC, int NS, int SKM, int SCP, int OTM, int NWV, int PFD = 1, bool SPLIT = false>
__global__ __launch_bounds__(NWV * 64, NWV == 16 ? 1 : (NWV == 4 ? 3 : 2)) void k_mlp_bf_u1(MlpBfArgs a) {
  constexpr int NFR = 2 * SKM + 2 * SCP + OTM;
  constexpr int NL1 = 2 * SKM + 2 * SCP;        // fragments of layer 1 (+ residual); the other OTM are layer 2's
  constexpr int NBUF = NWV == 16 ? 2 : 1;
  static_assert(!SPLIT || (NBUF == 1 && PFD == 1), "SPLIT is the single-buffer, one-ahead form");
  extern __shared__ __attribute__((aligned(16))) char wlb[];
  const FtnDesc* __restrict__ d = a.desc;
  const int N = a.B * d->total_px;
  if ((int)(blockIdx.x * NWV * 16) >= N) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, qa = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int n0 = (blockIdx.x * NWV + wave) * 16;
  const bool active = n0 < N;
  const int bufsz = NFR * 3 * 1024;
  auto dma_chunk = [&](int hc) {
    const __bf16* __restrict__ src = a.cfrag + (size_t)hc * NFR * 3 * 512;
    char* dst = wlb + (size_t)(NBUF == 2 ? (hc & 1) : 0) * bufsz;
    for (int piece = wv; piece < NFR * 3; piece += NWV)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(dst + (size_t)piece * 1024), 16, 0, 0);
  };
  // SPLIT: LDS = [layer-1 fragments, two buffers][layer-2 fragments, one buffer][biases].  A chunk's layer-2 fragments
  // and the NEXT chunk's layer-1 fragments are requested at the top of the chunk and land while layer 1 runs; the
  // barrier between the layers waits (counted vmcnt) for the former only.  Two barriers per chunk as before, but no
  // wave ever waits for a refill it has just issued - that wait was 17 % of the launch (ablation).
  constexpr int l1sz = NL1 * 3 * 1024, l2sz = OTM * 3 * 1024;
  auto dma_l1 = [&](int hc, int buf) {
    const __bf16* __restrict__ src = a.cfrag + (size_t)hc * NFR * 3 * 512;
    char* dst = wlb + (size_t)buf * l1sz;
    for (int piece = wv; piece < NL1 * 3; piece += NWV)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(dst + (size_t)piece * 1024), 16, 0, 0);
  };
  auto dma_l2 = [&](int hc) {
    const __bf16* __restrict__ src = a.cfrag + ((size_t)hc * NFR + NL1) * 3 * 512;
    char* dst = wlb + 2 * l1sz;
    for (int piece = wv; piece < OTM * 3; piece += NWV)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(dst + (size_t)piece * 1024), 16, 0, 0);
  };
  const int n1_mine = (NL1 * 3 - wv + NWV - 1) / NWV;          // layer-1 pieces this wave requests per chunk
  const Px px = decode_px16(d, a.x, a.B, a.L, a.C, n0, j, N);
  const int CP = a.CP;
  const int kmg = a.KM >> 4;
  constexpr int NWP = PxFmt<NS>::NW;
  constexpr int PXE = PxFmt<NS>::ELEMS;
  bool range_bad = false;                                       // f16x2: a value left the fp16 range (ftn_common.h)
  bf8 mp[SKM][NS], xp[SCP][NS];
  f4 xraw[SCP][2];
#pragma unroll
  for (int s = 0; s < SKM; ++s) {
    const int grp = 2 * s + (qa >> 1);
    const __bf16* __restrict__ src = a.m + ((size_t)px.n * kmg + (grp < kmg ? grp : 0)) * PXE + (qa & 1) * 8;
#pragma unroll
    for (int pz = 0; pz < NS; ++pz) mp[s][pz] = *(const bf8*)(src + pz * 16);
  }
#pragma unroll
  for (int s = 0; s < SCP; ++s) {
    xraw[s][0] = load_x4<XVEC>(px.xrow, 32 * s + 8 * qa, a.C);
    xraw[s][1] = load_x4<XVEC>(px.xrow, 32 * s + 8 * qa + 4, a.C);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (SPLIT) dma_l1(0, 0);
  else dma_chunk(0);
  const int FPc = a.n_hchunks * 32;
  float* __restrict__ bias_l = (float*)(wlb + (SPLIT ? (size_t)(2 * l1sz + l2sz) : (size_t)NBUF * bufsz));
  for (int i = threadIdx.x; i < 2 * FPc; i += NWV * 64) {
    const int c = i < FPc ? i : i - FPc;
    bias_l[i] = c < a.FP ? (i < FPc ? a.bo[c] : a.br[c]) : 0.f;
  }
#pragma unroll
  for (int s = 0; s < SKM; ++s) {
    if (2 * s + (qa >> 1) >= kmg) {
#pragma unroll
      for (int pz = 0; pz < NS; ++pz)
#pragma unroll
        for (int e = 0; e < 8; ++e) mp[s][pz][e] = (__bf16)0.0f;
    }
  }
#pragma unroll
  for (int s = 0; s < SCP; ++s) {
    float xv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { xv[e] = xraw[s][0][e]; xv[4 + e] = xraw[s][1][e]; }
    if (NS == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) range_bad |= h2_bad(xv[e]);
    }
    split_pieces<NS>(xv, xp[s]);
  }
  f4 oacc[OTM];
#pragma unroll
  for (int o = 0; o < OTM; ++o) oacc[o] = *(const f4*)(a.bc + 16 * o + 4 * qa);
  __syncthreads();
  if constexpr (SPLIT) {
    for (int hc = 0; hc < a.n_hchunks; ++hc) {
      const bool nxt = hc + 1 < a.n_hchunks;
      dma_l2(hc);                                    // last read in chunk hc - 1's layer 2 (every wave is past its end barrier)
      if (nxt) dma_l1(hc + 1, (hc + 1) & 1);         // that buffer was last read in chunk hc - 1's layer 1
      const char* __restrict__ wl1 = wlb + (size_t)(hc & 1) * l1sz + lane * 16;
      const char* __restrict__ wl2 = wlb + 2 * l1sz + lane * 16;
      f4 bo_t[2], br_t[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bo_t[t] = *(const f4*)(bias_l + 16 * (hc * 2 + t) + 4 * qa);
        br_t[t] = *(const f4*)(bias_l + FPc + 16 * (hc * 2 + t) + 4 * qa);
      }
      f4 h[2] = {bo_t[0], bo_t[1]};
      bf8 hp[NS];
      bf8 fr[2][NWP];
      auto ldfrag = [&](int f, bf8 (&ap)[NWP]) {
        const char* __restrict__ base = f < NL1 ? wl1 + (size_t)f * 3 * 1024 : wl2 + (size_t)(f - NL1) * 3 * 1024;
#pragma unroll
        for (int pz = 0; pz < NWP; ++pz) ap[pz] = *(const bf8*)(base + (size_t)pz * 1024);
      };
      auto step = [&](int f) {
        if (f + 1 < NFR) ldfrag(f + 1, fr[(f + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf8 (&cur)[NWP] = fr[f & 1];
        if (f < SKM) h[0] = chain_bf<NS>(cur, mp[f < SKM ? f : 0], h[0]);
        else if (f < 2 * SKM) h[1] = chain_bf<NS>(cur, mp[f < 2 * SKM ? f - SKM : 0], h[1]);
        else if (f < 2 * SKM + SCP) h[0] = chain_bf<NS>(cur, xp[f < 2 * SKM + SCP ? f - 2 * SKM : 0], h[0]);
        else if (f < NL1) h[1] = chain_bf<NS>(cur, xp[f < NL1 ? f - 2 * SKM - SCP : 0], h[1]);
        else oacc[f < NFR ? f - NL1 : 0] = chain_bf<NS>(cur, hp, oacc[f < NFR ? f - NL1 : 0]);
        if (f == SKM - 1) h[0] = NS == 2 ? act4<ACT>(h[0] * a.inv_o) * a.sc_r + br_t[0] : act4<ACT>(h[0]) + br_t[0];
        if (f == 2 * SKM - 1) h[1] = NS == 2 ? act4<ACT>(h[1] * a.inv_o) * a.sc_r + br_t[1] : act4<ACT>(h[1]) + br_t[1];
        if (f == 2 * SKM + SCP - 1) h[0] = act4<ACT>(NS == 2 ? h[0] * a.inv_r : h[0]);
        if (f == NL1 - 1) {
          h[1] = act4<ACT>(NS == 2 ? h[1] * a.inv_r : h[1]);
          float hv[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { hv[e] = h[0][e]; hv[4 + e] = h[1][e]; }
          split_pieces<NS>(hv, hp);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      if (active) {
        ldfrag(0, fr[0]);
#pragma unroll
        for (int f = 0; f < NL1 - 1; ++f) step(f);       // these steps read layer-1 fragments only (incl. the prefetch)
      }
      // layer 2's fragments have landed on every wave; the next chunk's layer-1 pieces (issued behind them) stay in flight
      barrier_keep_vm(nxt ? n1_mine : 0);
      if (active) {
#pragma unroll
        for (int f = NL1 - 1; f < NFR; ++f) step(f);     // the last layer-1 step prefetches the first layer-2 fragment
      }
      barrier_keep_vm(0);                              // every wave is done with both buffers; the next layer-1 set has landed
    }
  } else
  for (int hc = 0; hc < a.n_hchunks; ++hc) {
    const char* __restrict__ wl = wlb + (size_t)(NBUF == 2 ? (hc & 1) : 0) * bufsz + lane * 16;
    // double-buffered: the other buffer was last read in chunk hc - 1, which every wave left at the barrier
    if (NBUF == 2 && hc + 1 < a.n_hchunks) dma_chunk(hc + 1);
    f4 bo_t[2], br_t[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bo_t[t] = *(const f4*)(bias_l + 16 * (hc * 2 + t) + 4 * qa);
      br_t[t] = *(const f4*)(bias_l + FPc + 16 * (hc * 2 + t) + 4 * qa);
    }
    if (active) {
      f4 h[2] = {bo_t[0], bo_t[1]};
      bf8 hp[NS];
      // fragment reads run PFD steps ahead of their MFMAs in a ring of PFD + 1 register sets (PFD = 2 where the
      // registers allow): one step (3 MFMAs + its share of the GELUs) is shorter than an LDS read's latency when
      // sixteen waves stream fragments at once
      bf8 fr[PFD + 1][NWP];
      auto ldfrag = [&](int f, bf8 (&ap)[NWP]) {
#pragma unroll
        for (int pz = 0; pz < NWP; ++pz) ap[pz] = *(const bf8*)(wl + (size_t)(f * 3 + pz) * 1024);
      };
#pragma unroll
      for (int f = 0; f < PFD; ++f) ldfrag(f, fr[f]);
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        if (f + PFD < NFR) ldfrag(f + PFD, fr[(f + PFD) % (PFD + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        const bf8 (&cur)[NWP] = fr[f % (PFD + 1)];
        if (f < SKM) h[0] = chain_bf<NS>(cur, mp[f], h[0]);                         // layer 1, hidden tile 0
        else if (f < 2 * SKM) h[1] = chain_bf<NS>(cur, mp[f - SKM], h[1]);          // layer 1, hidden tile 1
        else if (f < 2 * SKM + SCP) h[0] = chain_bf<NS>(cur, xp[f - 2 * SKM], h[0]);               // + res1(x)
        else if (f < 2 * SKM + 2 * SCP) h[1] = chain_bf<NS>(cur, xp[f - 2 * SKM - SCP], h[1]);
        else oacc[f - 2 * SKM - 2 * SCP] = chain_bf<NS>(cur, hp, oacc[f - 2 * SKM - 2 * SCP]);    // a' | res2
        // act, then the residual adds on; NS == 2 undoes / applies the weight prescales (see MlpBfArgs)
        if (f == SKM - 1) h[0] = NS == 2 ? act4<ACT>(h[0] * a.inv_o) * a.sc_r + br_t[0] : act4<ACT>(h[0]) + br_t[0];
        if (f == 2 * SKM - 1) h[1] = NS == 2 ? act4<ACT>(h[1] * a.inv_o) * a.sc_r + br_t[1] : act4<ACT>(h[1]) + br_t[1];
        if (f == 2 * SKM + SCP - 1) h[0] = act4<ACT>(NS == 2 ? h[0] * a.inv_r : h[0]);   // TimesBlock's mid activation
        if (f == 2 * SKM + 2 * SCP - 1) {
          h[1] = act4<ACT>(NS == 2 ? h[1] * a.inv_r : h[1]);
          float hv[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { hv[e] = h[0][e]; hv[4 + e] = h[1][e]; }
          split_pieces<NS>(hv, hp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (NBUF == 1 && hc + 1 < a.n_hchunks) {
      dma_chunk(hc + 1);
      __syncthreads();
    }
  }
  if (active && px.ok && NS == 2) {
#pragma unroll
    for (int o = 0; o < OTM; ++o)
      if (o < a.n_oa) range_bad |= h2_bad4(oacc[o] * a.inv_a);
  }
  if (NS == 2) raise_range_flag(a.range_flag, range_bad);
  if (!active || !px.ok) return;
#pragma unroll
  for (int o = 0; o < OTM; ++o) {
    if (o < a.n_oa) {
      store_px<NS == 2 ? 2 : 3>(a.outA + ((size_t)px.n * (a.AC >> 4) + o) * PXE, qa, NS == 2 ? oacc[o] * a.inv_a : oacc[o]);
    } else {
      const int ch = 16 * (o - a.n_oa) + 4 * qa;
      const f4 rv = NS == 2 ? oacc[o] * a.inv_r2 : oacc[o];
      *(f4*)(a.outR + (size_t)px.n * CP + ch) = a.r_keeps_x ? rv : rv - load_x4<XVEC>(px.xrow, ch, a.C);
    }
  }
}

template <int ACT, int NS, int SKM, int SCP, int OTM, int NWV, int PFD = 1, bool SPLIT = false>
static int launch_mlp_bf_u1w(MlpBfArgs ma, bool xvec, long long Nmax, hipStream_t st) {
  ma.dbg = nullptr; ma.dbg_cap = 0;
  const size_t lds = (SPLIT ? (size_t)(2 * (2 * SKM + 2 * SCP) + OTM) * 3 * 1024
                            : (size_t)ma.per_chunk * 3 * 1024 * (NWV == 16 ? 2 : 1)) + (size_t)ma.n_hchunks * 32 * 2 * sizeof(float);
  if (lds > 160 * 1024) { ftn_set_error("stage C needs %zu B of LDS", lds); return -1; }
  const int nblk = (int)((Nmax + NWV * 16 - 1) / (NWV * 16));
  hipError_t e = xvec ? hipFuncSetAttribute((const void*)k_mlp_bf_u1<ACT, true, NS, SKM, SCP, OTM, NWV, PFD, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                      : hipFuncSetAttribute((const void*)k_mlp_bf_u1<ACT, false, NS, SKM, SCP, OTM, NWV, PFD, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_mlp_bf_u1): %s", hipGetErrorString(e)); return (int)e; }
  if (xvec) hipLaunchKernelGGL((k_mlp_bf_u1<ACT, true, NS, SKM, SCP, OTM, NWV, PFD, SPLIT>), dim3(nblk), dim3(NWV * 64), lds, st, ma);
  else hipLaunchKernelGGL((k_mlp_bf_u1<ACT, false, NS, SKM, SCP, OTM, NWV, PFD, SPLIT>), dim3(nblk), dim3(NWV * 64), lds, st, ma);
  FTN_CHECK_LAUNCH();
  return 0;
}

template <int ACT, int NS, int SKM, int SCP, int OTM>
static int launch_mlp_bf_u1(const MlpBfArgs& ma, bool xvec, long long Nmax, hipStream_t st) {
  // 16-wave double-buffered form where two chunk buffers fit LDS (d_model 64: 2 x 45 KB) unless FTN_MLP_W16=0
  if (g_mlp_w16 && (size_t)ma.per_chunk * 3 * 1024 * 2 + (size_t)ma.n_hchunks * 256 <= 160 * 1024)
    return launch_mlp_bf_u1w<ACT, NS, SKM, SCP, OTM, 16>(ma, xvec, Nmax, st);
  if (g_mlp_w4 && ((size_t)ma.per_chunk * 3 * 1024 + (size_t)ma.n_hchunks * 256) * 3 <= 160 * 1024)
    return launch_mlp_bf_u1w<ACT, NS, SKM, SCP, OTM, 4>(ma, xvec, Nmax, st);
  if constexpr (NS == 2 && OTM <= 7) {
    if (g_mlp_pfd2) return launch_mlp_bf_u1w<ACT, NS, SKM, SCP, OTM, 8, 2>(ma, xvec, Nmax, st);
  }
  // default: split refill where two workgroups' LDS still fit a CU (or, for the d_model-128 shape, one does)
  if (g_mlp_split && ma.per_chunk == 2 * SKM + 2 * SCP + OTM &&
      (size_t)(2 * (2 * SKM + 2 * SCP) + OTM) * 3 * 1024 + (size_t)ma.n_hchunks * 256 <= (OTM <= 7 ? 80 : 160) * 1024)
    return launch_mlp_bf_u1w<ACT, NS, SKM, SCP, OTM, 8, 1, true>(ma, xvec, Nmax, st);
  return launch_mlp_bf_u1w<ACT, NS, SKM, SCP, OTM, 8>(ma, xvec, Nmax, st);
}

template <int ACT, int NS>
static int launch_mlp_bf_c128(const MlpBfArgs& ma, bool xvec, long long Nmax, hipStream_t st) {
  return launch_mlp_bf_u1<ACT, NS, 3, 4, 14>(ma, xvec, Nmax, st);
}

// ---------------------------------------------------------------- stages B / D
// Grouped k x k convolution as an im2col GEMM.  One workgroup = one conv tile
// (normally a whole period grid, <= 384 pixels) x one branch x NCO output-channel
// tiles.  Per 16-input-channel chunk it stages (a) the tile plus halo, clipped to
// the grid, as [pixel][16 ch] rows of 80 B (16 consecutive pixels hit 16 distinct
// bank quads on ds_read_b128) and (b) that chunk's weight fragments for every tap,
// lane-linear (1 KiB per fragment, conflict-free).  A tap outside the grid is conv
// zero padding: the lane reads a zeroed slot instead (row/column validity bits are
// precomputed per pixel).  The tap loop is software-pipelined: the next tap's LDS
// reads are issued before the current tap's MFMAs.
struct ConvArgs {
  const float* in;       // [N][INC]
  float* out;            // [N][OUTC]
  const float* W[FTN_MAXBR];   // per branch [taps][ncc][nco][lane][4]
  const float* bias;     // [OUTC] (per branch slice at out_off)
  const FtnDesc* desc;
  int B, INC, OUTC;
  int nbr;
  int cin;               // input channels per branch (multiple of 16)
  int cout;              // output channels per branch (multiple of 16)
  int in_stride_br;      // channel offset between branches on the input  (cin or 0)
  int out_stride_br;     // channel offset between branches on the output (cout)
  int nchunk;            // output-channel chunks per branch = ceil(cout/16 / NCO)
  int region_floats;     // LDS floats reserved for the staged region (+ zero slot)
  int kh[FTN_MAXBR], kw[FTN_MAXBR];
  int order[FTN_MAXBR];  // branches sorted by descending tap count (heavy workgroups first)
  int bt_L;              // > 0: `in` holds one row per window position, [B*L + 1][INC] (row B*L = the zero-input
                         // pad pixel), shared by every period group; grid pixel t of batch row b is row
                         // b*L + t for t < L and the pad row otherwise.  0: `in` is per grid pixel, [N][INC]
  unsigned long long* dbg; size_t dbg_cap;
};

#define LDS_PX_STRIDE 20  // 16 channels + 4 pad dwords
#define CONV_NU 6         // 16-pixel units per wave: 4 waves x 6 x 16 >= FTN_TILE_PX

// One kernel row (fixed dy) of taps for NU units.  KW > 0: the dx loop is fully
// unrolled (tap offsets become ds_read immediates, no per-tap address math);
// KW == 0: runtime kw.  Row validity is folded into the column mask once per row,
// so a tap costs one bit test + one address select per unit.
template <int NCO, int NU, int KW>
__device__ __forceinline__ void conv_row(f4 (&acc)[NCO][CONV_NU], const float* __restrict__ tile,
                                         const float* __restrict__ wrow, const int (&rowaddr)[NU],
                                         const unsigned (&cmv)[NU], int kw, int zoff) {
  const int n = KW > 0 ? KW : kw;
#pragma unroll
  for (int dx = 0; dx < n; ++dx) {
    f4 bf[NU], af[NCO];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const bool v = ((cmv[u] >> dx) & 1u) != 0u;
      bf[u] = *(const f4*)(tile + (v ? rowaddr[u] + dx * LDS_PX_STRIDE : zoff));
    }
#pragma unroll
    for (int o = 0; o < NCO; ++o) af[o] = *(const f4*)(wrow + (dx * NCO + o) * 256);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 0; o < NCO; ++o)
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[o][u] = mfma16(af[o][e], bf[u][e], acc[o][u]);
  }
}

template <int NCO, int NU>
__device__ __forceinline__ void conv_taps(f4 (&acc)[NCO][CONV_NU], const float* __restrict__ tile,
                                          const float* __restrict__ wl, const int (&lbase)[CONV_NU],
                                          const unsigned (&rmask)[CONV_NU], const unsigned (&cmask)[CONV_NU],
                                          int kh, int kw, int RW, int zoff, int lane) {
  const int hy = kh >> 1, hx = kw >> 1;
  for (int dy = 0; dy < kh; ++dy) {
    int rowaddr[NU];
    unsigned cmv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      rowaddr[u] = lbase[u] + ((dy - hy) * RW - hx) * LDS_PX_STRIDE;
      cmv[u] = ((rmask[u] >> dy) & 1u) ? cmask[u] : 0u;
    }
    const float* __restrict__ wrow = wl + (size_t)dy * kw * NCO * 256 + lane * 4;
    if (kw == 7) conv_row<NCO, NU, 7>(acc, tile, wrow, rowaddr, cmv, kw, zoff);
    else if (kw == 5) conv_row<NCO, NU, 5>(acc, tile, wrow, rowaddr, cmv, kw, zoff);
    else if (kw == 3) conv_row<NCO, NU, 3>(acc, tile, wrow, rowaddr, cmv, kw, zoff);
    else if (kw == 1) conv_row<NCO, NU, 1>(acc, tile, wrow, rowaddr, cmv, kw, zoff);
    else conv_row<NCO, NU, 0>(acc, tile, wrow, rowaddr, cmv, kw, zoff);
  }
}

template <int NCO>
__global__ __launch_bounds__(256) void k_conv(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const FtnDesc* __restrict__ d = a.desc;
  const size_t wgid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  stamp(a.dbg, a.dbg_cap, wgid, 0);
  if (a.dbg != nullptr && threadIdx.x == 0 && wgid * 8 + 7 < a.dbg_cap) a.dbg[wgid * 8 + 6] = __builtin_amdgcn_s_memrealtime();
  // grid.x = max_groups (one tile per group is the common case); a workgroup walks the
  // data-dependent tile list with that stride, so no workgroup is dispatched empty unless
  // groups merged (interleaved empty workgroups skew the round-robin XCD placement)
  const int tiles_total = d->tiles_per_row;
  for (int bx = blockIdx.x; bx < tiles_total; bx += gridDim.x) {
  if (bx != (int)blockIdx.x) __syncthreads();
  const int b = blockIdx.y;
  const int zb = blockIdx.z / a.nchunk, chunk = blockIdx.z - zb * a.nchunk;
  const int br = a.order[zb];
  const int G = d->n_groups;
  int g = 0;
  for (int gg = 1; gg < G; ++gg)
    if (bx >= d->g_tile_off[gg]) g = gg;
  const int tix = bx - d->g_tile_off[g];
  const int ntx = d->g_ntx[g];
  const int ty = tix / ntx, tx = tix - ty * ntx;
  const int p = d->g_period[g], cycles = d->g_cycles[g];
  const int P = d->g_px_off[g + 1] - d->g_px_off[g];
  const int r0 = ty * d->g_th[g], c0 = tx * d->g_tw[g];
  const int th = min(d->g_th[g], cycles - r0), tw = min(d->g_tw[g], p - c0);
  const int kh = a.kh[br], kw = a.kw[br], hy = kh >> 1, hx = kw >> 1;
  // staged region = tile + halo, clipped to the grid
  const int R0 = max(0, r0 - hy), R1 = min(cycles, r0 + th + hy);
  const int C0 = max(0, c0 - hx), C1 = min(p, c0 + tw + hx);
  const int RW = C1 - C0, RH = R1 - R0;
  float* __restrict__ tile = lds;
  float* __restrict__ wl = lds + a.region_floats;
  const int zoff = a.region_floats - 16;                 // 16 zero floats at the end of the region area
  const size_t nimg = (size_t)a.B * d->g_px_off[g] + (size_t)b * P;
  const int btL = a.bt_L;
  const float* __restrict__ in = a.in + (btL > 0 ? (size_t)b * btL : nimg) * a.INC + br * a.in_stride_br;
  const float* __restrict__ in_pad = a.in + (size_t)a.B * btL * a.INC + br * a.in_stride_br;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int npx = th * tw, nunits = (npx + 15) >> 4;
  const int wrot = (wave + b) & 3;                                 // rotate so co-resident workgroups balance the SIMDs
  const int nu = nunits > wrot ? (nunits - wrot + 3) >> 2 : 0;   // units of this wave: wrot, wrot+4, ...
  const int nco_tot = a.cout >> 4, co0 = chunk * NCO;
  const int ncc = a.cin >> 4, ntaps = kh * kw;

  int lbase[CONV_NU], oidx[CONV_NU];
  unsigned rmask[CONV_NU], cmask[CONV_NU];
  bool pok[CONV_NU];
  // idx / tw by reciprocal: exact for idx < 2^20 because (idx + 0.5) / tw is never an integer
  const float inv_tw = 1.0f / (float)tw;
  const unsigned kmh = (1u << kh) - 1u, kmw = (1u << kw) - 1u;
#pragma unroll
  for (int u = 0; u < CONV_NU; ++u) {
    int idx = (wrot + 4 * u) * 16 + j;
    pok[u] = idx < npx;
    if (!pok[u]) idx = 0;
    const int r = (int)(((float)idx + 0.5f) * inv_tw), c = idx - r * tw;
    const int ri = r0 + r, ci = c0 + c;
    lbase[u] = ((ri - R0) * RW + (ci - C0)) * LDS_PX_STRIDE + 4 * q;
    oidx[u] = ri * p + ci;
    // taps dy with 0 <= ri + dy - hy < cycles are the bits [lo, hi) of the row mask (same for columns)
    const int rlo = max(0, hy - ri), rhi = min(kh, cycles + hy - ri);
    const int clo = max(0, hx - ci), chi = min(kw, p + hx - ci);
    const unsigned rm = (rhi > rlo) ? ((kmh >> (kh - rhi)) & (kmh << rlo)) & kmh : 0u;
    const unsigned cm = (chi > clo) ? ((kmw >> (kw - chi)) & (kmw << clo)) & kmw : 0u;
    rmask[u] = pok[u] ? rm : 0u;
    cmask[u] = pok[u] ? cm : 0u;
  }
  f4 acc[NCO][CONV_NU];
#pragma unroll
  for (int o = 0; o < NCO; ++o) {
    f4 bv = {0.f, 0.f, 0.f, 0.f};
    if (co0 + o < nco_tot) bv = *(const f4*)(a.bias + br * a.out_stride_br + 16 * (co0 + o) + 4 * q);
#pragma unroll
    for (int u = 0; u < CONV_NU; ++u) acc[o][u] = bv;
  }
  if (threadIdx.x < 4) *(f4*)(tile + zoff + 4 * threadIdx.x) = f4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ Wb = a.W[br];
  const int nstage = RH * RW * 4;
  const float inv_rw = 1.0f / (float)RW;
  for (int cc = 0; cc < ncc; ++cc) {
    if (cc > 0) __syncthreads();
    // weight fragments: LDS-DMA (global_load_lds_dwordx4), one 1-KiB fragment per wave
    // instruction, no VGPR round trip; all pieces of a wave are in flight together and the
    // region loads below join the same queue, so staging costs ~one L2 latency.
    {
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      const int npieces = ntaps * NCO;
      for (int piece = wv; piece < npieces; piece += 4) {
        const int tap = piece / NCO, o = piece - tap * NCO;
        if (co0 + o < nco_tot) {
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(Wb + ((size_t)(tap * ncc + cc) * nco_tot + co0 + o) * 256 + lane * 4),
              (__attribute__((address_space(3))) void*)(wl + (size_t)piece * 256), 16, 0, 0);
        } else {
          *(f4*)(wl + (size_t)piece * 256 + lane * 4) = f4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    for (int s0 = threadIdx.x; s0 < nstage; s0 += 256 * 6) {
      f4 v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int s = s0 + 256 * k;
        const int sp = s >> 2, qq = s & 3;
        const int rr = (int)(((float)sp + 0.5f) * inv_rw), cx = sp - rr * RW;
        const int tpx = (R0 + rr) * p + C0 + cx;               // grid pixel = window position t (fold, :1041-1046)
        const float* __restrict__ row = (btL > 0 && tpx >= btL) ? in_pad : in + (size_t)tpx * a.INC;
        v[k] = s < nstage ? *(const f4*)(row + 16 * cc + 4 * qq) : f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int s = s0 + 256 * k;
        if (s < nstage) *(f4*)(tile + (s >> 2) * LDS_PX_STRIDE + 4 * (s & 3)) = v[k];
      }
    }
    __syncthreads();
    if (cc == 0) stamp(a.dbg, a.dbg_cap, wgid, 1);
    switch (nu) {
      case 6: conv_taps<NCO, 6>(acc, tile, wl, lbase, rmask, cmask, kh, kw, RW, zoff, lane); break;
      case 5: conv_taps<NCO, 5>(acc, tile, wl, lbase, rmask, cmask, kh, kw, RW, zoff, lane); break;
      case 4: conv_taps<NCO, 4>(acc, tile, wl, lbase, rmask, cmask, kh, kw, RW, zoff, lane); break;
      case 3: conv_taps<NCO, 3>(acc, tile, wl, lbase, rmask, cmask, kh, kw, RW, zoff, lane); break;
      case 2: conv_taps<NCO, 2>(acc, tile, wl, lbase, rmask, cmask, kh, kw, RW, zoff, lane); break;
      case 1: conv_taps<NCO, 1>(acc, tile, wl, lbase, rmask, cmask, kh, kw, RW, zoff, lane); break;
      default: break;
    }
  }
  stamp(a.dbg, a.dbg_cap, wgid, 2);
  if (a.dbg != nullptr && threadIdx.x == 0 && wgid * 8 + 7 < a.dbg_cap) {
    a.dbg[wgid * 8 + 4] = (unsigned long long)(kh * kw);
    a.dbg[wgid * 8 + 5] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // XCC_ID
  }
  float* __restrict__ out = a.out + nimg * a.OUTC + br * a.out_stride_br;
#pragma unroll
  for (int o = 0; o < NCO; ++o) {
    if (co0 + o < nco_tot) {
#pragma unroll
      for (int u = 0; u < CONV_NU; ++u)
        if (u < nu && pok[u]) *(f4*)(out + (size_t)oidx[u] * a.OUTC + 16 * (co0 + o) + 4 * q) = acc[o][u];
    }
  }
  stamp(a.dbg, a.dbg_cap, wgid, 3);
  if (a.dbg != nullptr && threadIdx.x == 0 && wgid * 8 + 7 < a.dbg_cap) a.dbg[wgid * 8 + 7] = __builtin_amdgcn_s_memrealtime();
  }  // tile loop
}

// ---------------------------------------------------------------- stages B / D, bf16x3 engine
// Same convolution on the bf16 matrix pipe with fp32-equivalent accuracy: activations arrive
// as three bf16 pieces per value (P3 layout, written by the producing stage), weights are
// pre-split on the host, and every K=32 slab (= two taps x 16 input channels) is six
// v_mfma_f32_16x16x32_bf16 (hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi) into one fp32
// accumulator: 96 cycles instead of 256 for the same contraction in fp32 MFMA.
// One 512-thread workgroup (two waves per SIMD) owns one tile x branch and walks `bpw`
// batch rows: the weight fragments are staged once by LDS-DMA, the tile regions are
// double-buffered (row i+1 is requested before row i is computed), so neither is on the
// critical path.  NS = 1 drops the mid/lo pieces (plain bf16, BASELINE configs[2]).
struct ConvBfArgs {
  const __bf16* in;      // P3 / H2 [rows][INC/16][pieces][16]
  void* out;             // fp32 [N][OUTC] or P3 / H2 [N][OUTC/16][pieces][16]
  const __bf16* W[FTN_MAXBR];  // per branch [cc][co][slab][piece][lane][8]
  const float* bias;     // f16x2: prescaled by the branch's weight scale
  const FtnDesc* desc;
  int B, INC, OUTC, out_p3;
  int nbr, cin, cout, in_stride_br, out_stride_br, nchunk;
  int plane_bytes;       // one piece plane of a region buffer (32 B per pixel) incl. its zero pixel at the end
  int region_bytes;      // one region buffer = pieces x plane_bytes
  int wbytes;            // weight fragment bytes in LDS
  int bpw;               // batch rows per workgroup
  int sgroup;            // K=32 slabs whose weight fragments are resident at a time (>= max slabs: all of them)
  int kh[FTN_MAXBR], kw[FTN_MAXBR], order[FTN_MAXBR];
  int bt_L;              // as ConvArgs.bt_L: > 0 = input rows per window position + one pad row
  float inv[FTN_MAXBR];  // f16x2: 2^-s of the branch's prescaled weights (applied to the accumulators)
  int wg_off[2 * FTN_MAXBR + 1];   // k_conv_bf_fast: workgroups [wg_off[v], wg_off[v+1]) serve virtual branch v = branch * (cout / 16) + output tile
  int nvb;               // virtual branches of the fast path: nbr * (cout / 16)
  int* range_flag;       // f16x2 piece output (out_p3): set when an output leaves the fp16 range; may be null
  int abl;               // timing ablations of k_conv_bf_fast (FTN_CONV_ABL; results wrong): 1 no output stores, 2 no region
                         // DMA after a tile's first row, 4 no barrier after a tile's first row, 8 no MFMA work
  unsigned long long* dbg; size_t dbg_cap;
};

#define CBF_NU 3          // 16-pixel units per wave: 8 waves x 3 x 16 >= FTN_TILE_PX
// LDS image of a staged region: one PLANE per piece, 32 bytes per pixel = [channels 0-7][channels 8-15] of that
// piece.  A ds_read_b128 is served in four groups of 16 lanes - {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same + 32 (MI355X_MICROARCH.md, LDS) - and a lane (j = lane & 15, qa = lane >> 4) reads pixel j's half qa & 1:
// with a 32-byte pixel stride the eight half-0 lanes of a group fall on the even 16-byte bank quads and its eight
// half-1 lanes on the odd ones, all distinct, for every tap offset (a constant shift).  Round 1's pixel-major
// image (96 B of pieces + 16 B pad per pixel) put seven of those sixteen lanes on a shared quad.
#define CBF_PX_BYTES 32

template <int NCO, int NS>
__global__ __launch_bounds__(512) void k_conv_bf(ConvBfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  constexpr int NWP = PxFmt<NS>::NW;                          // weight pieces
  constexpr int PXE = PxFmt<NS>::ELEMS;                       // 16-bit elements per pixel and 16-channel group in memory
  const FtnDesc* __restrict__ d = a.desc;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, qa = lane >> 4;
  bool range_bad = false;                                       // f16x2: an output left the fp16 range
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int zb = blockIdx.z / a.nchunk, chunk = blockIdx.z - zb * a.nchunk;
  const int br = a.order[zb];
  const int kh = a.kh[br], kw = a.kw[br], hy = kh >> 1, hx = kw >> 1, ntaps = kh * kw;
  const int S = (ntaps + 1) >> 1;
  const int nco_tot = a.cout >> 4, co0 = chunk * NCO, ncc = a.cin >> 4;
  char* __restrict__ wl = ldsb;
  char* __restrict__ rbuf0 = ldsb + a.wbytes;
  const int plane = a.plane_bytes;
  const int zbase = plane - 256;                             // 256-byte zero block at the end of every plane (256-aligned)
  const int b_begin = blockIdx.y * a.bpw, b_end = min(a.B, b_begin + a.bpw);
  const int G = d->n_groups, tiles_total = d->tiles_per_row;
  // zero blocks of every plane of both region buffers (never overwritten by the DMA)
  if (threadIdx.x < 2 * NS * 16) {
    const int pl = threadIdx.x >> 4;                           // (buffer, piece) plane index
    *(f4*)(rbuf0 + (size_t)(pl / NS) * a.region_bytes + (size_t)(pl % NS) * plane + zbase + (threadIdx.x & 15) * 16) = f4{0.f, 0.f, 0.f, 0.f};
  }
  const size_t wgid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  stamp(a.dbg, a.dbg_cap, wgid, 0);
  if (a.dbg != nullptr && threadIdx.x == 0 && wgid * 8 + 7 < a.dbg_cap) { a.dbg[wgid * 8 + 6] = __builtin_amdgcn_s_memrealtime(); a.dbg[wgid * 8 + 4] = (unsigned long long)ntaps; a.dbg[wgid * 8 + 5] = 0; }

  for (int bx = blockIdx.x; bx < tiles_total; bx += gridDim.x) {
    int g = 0;
    for (int gg = 1; gg < G; ++gg)
      if (bx >= d->g_tile_off[gg]) g = gg;
    const int tix = bx - d->g_tile_off[g];
    const int ntx = d->g_ntx[g];
    const int ty = tix / ntx, tx = tix - ty * ntx;
    const int p = d->g_period[g], cycles = d->g_cycles[g];
    const int P = d->g_px_off[g + 1] - d->g_px_off[g];
    const int r0 = ty * d->g_th[g], c0 = tx * d->g_tw[g];
    const int th = min(d->g_th[g], cycles - r0), tw = min(d->g_tw[g], p - c0);
    const int R0 = max(0, r0 - hy), R1 = min(cycles, r0 + th + hy);
    const int C0 = max(0, c0 - hx), C1 = min(p, c0 + tw + hx);
    const int RW = C1 - C0, RH = R1 - R0;
    const int npx = th * tw, nunits = (npx + 15) >> 4;
    const int in_groups = a.INC >> 4;
    const float inv_tw = 1.0f / (float)tw, inv_rw = 1.0f / (float)RW;
    const unsigned kmh = (1u << kh) - 1u, kmw = (1u << kw) - 1u;
    const int nchunks16 = RH * RW * 2;                     // 16-byte chunks of one plane (two per pixel)
    const int ppp = (nchunks16 + 63) >> 6;                 // 1-KiB DMA instructions per plane

    // region of batch row b, channel group cc -> buffer `buf`: every lane fetches the 16 bytes (pixel, piece,
    // channel half) that belong at its linear LDS position, so the fold (:1041-1046), the clipped halo and the
    // plane split all happen in the DMA's source addresses
    const int btL = a.bt_L;
    auto dma_region = [&](int b, int cc, int buf) {
      const __bf16* __restrict__ src = a.in + (btL > 0 ? (size_t)b * btL : (size_t)a.B * d->g_px_off[g] + (size_t)b * P) * in_groups * PXE +
                                       (size_t)(br * a.in_stride_br + cc) * PXE;
      const __bf16* __restrict__ src_pad = a.in + (size_t)a.B * btL * in_groups * PXE + (size_t)(br * a.in_stride_br + cc) * PXE;
      for (int pc = wv; pc < NS * ppp; pc += 8) {
        const int pz = pc / ppp, pi = pc - pz * ppp;
        int ci = pi * 64 + lane;
        if (ci >= nchunks16) ci = nchunks16 - 1;             // tail lanes re-read the last chunk (lands in the plane's slack)
        const int sp = ci >> 1, half = ci & 1;
        const int rr = (int)(((float)sp + 0.5f) * inv_rw), cx = sp - rr * RW;
        const int tpx = (R0 + rr) * p + C0 + cx;             // grid pixel = window position t (fold, :1041-1046)
        const __bf16* __restrict__ row = (btL > 0 && tpx >= btL) ? src_pad : src + (size_t)tpx * in_groups * PXE;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(row + pz * 16 + half * 8),
            (__attribute__((address_space(3))) void*)(rbuf0 + (size_t)buf * a.region_bytes + (size_t)pz * plane + (size_t)pi * 1024), 16, 0, 0);
      }
    };
    // weight fragments of slabs [g0, g1) of channel chunk cc -> LDS slots [o][slab - g0][piece]
    const int SG = a.sgroup < S ? a.sgroup : S;               // slabs resident at a time
    auto dma_weights = [&](int cc, int g0, int g1) {
      const int nfr = (g1 - g0) * 3;                          // 1-KiB fragments per output tile
      for (int o = 0; o < NCO; ++o) {
        if (co0 + o < nco_tot) {
          const __bf16* __restrict__ src = a.W[br] + ((size_t)(cc * nco_tot + co0 + o) * S + g0) * 3 * 512;
          for (int f = wv; f < nfr; f += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)f * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void*)(wl + ((size_t)o * SG * 3 + f) * 1024), 16, 0, 0);
        } else {
          for (int f = wv; f < nfr; f += 8) *(f4*)(wl + ((size_t)o * SG * 3 + f) * 1024 + lane * 16) = f4{0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    const bool resident = ncc == 1 && SG == S;                // one staging per tile serves every batch row

    // per-lane pixel bookkeeping, once per tile (rotated by the workgroup's batch chunk so the
    // 3-unit waves spread over the SIMDs)
    const int wrot = (wave + (int)blockIdx.y) & 7;
    const int nu = nunits > wrot ? (nunits - wrot + 7) >> 3 : 0;
    int lbase[CBF_NU], oidx[CBF_NU];
    unsigned rmask[CBF_NU], cmask[CBF_NU];
    bool pok[CBF_NU];
#pragma unroll
    for (int u = 0; u < CBF_NU; ++u) {
      int idx = (wrot + 8 * u) * 16 + j;
      pok[u] = idx < npx;
      if (!pok[u]) idx = 0;
      const int r = (int)(((float)idx + 0.5f) * inv_tw), c = idx - r * tw;
      const int ri = r0 + r, ci = c0 + c;
      lbase[u] = ((ri - R0 - hy) * RW + (ci - C0 - hx)) * CBF_PX_BYTES + (qa & 1) * 16;
      oidx[u] = ri * p + ci;
      const int rlo = max(0, hy - ri), rhi = min(kh, cycles + hy - ri);
      const int clo = max(0, hx - ci), chi = min(kw, p + hx - ci);
      const unsigned rm = (rhi > rlo) ? ((kmh >> (kh - rhi)) & (kmh << rlo)) & kmh : 0u;
      const unsigned cm = (chi > clo) ? ((kmw >> (kw - chi)) & (kmw << clo)) & kmw : 0u;
      rmask[u] = pok[u] ? rm : 0u;
      cmask[u] = pok[u] ? cm : 0u;
    }
    __syncthreads();                                          // previous tile's readers are done
    if (resident) dma_weights(0, 0, S);
    if (b_begin < b_end) dma_region(b_begin, 0, 0);
    int it = 0;                                                // region buffer parity
    for (int b = b_begin; b < b_end; ++b) {
      f4 acc[NCO][CBF_NU];
#pragma unroll
      for (int o = 0; o < NCO; ++o) {
        f4 bv = {0.f, 0.f, 0.f, 0.f};
        if (co0 + o < nco_tot) bv = *(const f4*)(a.bias + br * a.out_stride_br + 16 * (co0 + o) + 4 * (lane >> 4));
#pragma unroll
        for (int u = 0; u < CBF_NU; ++u) acc[o][u] = bv;
      }
      for (int cc = 0; cc < ncc; ++cc) {
        if (!resident) { __syncthreads(); dma_weights(cc, 0, SG); }
        __syncthreads();                                      // region (b, cc) and weights have landed (vmcnt(0))
        if (b == b_begin && cc == 0) stamp(a.dbg, a.dbg_cap, wgid, 1);
        if (b == b_begin + 1 && cc == 0) stamp(a.dbg, a.dbg_cap, wgid, 2);
        // request the next region while this one is consumed
        {
          int nb = b, ncq = cc + 1;
          if (ncq == ncc) { ncq = 0; nb = b + 1; }
          if (nb < b_end) dma_region(nb, ncq, (it + 1) & 1);
        }
        const char* __restrict__ reg = rbuf0 + (size_t)(it & 1) * a.region_bytes;
        ++it;
        // slabs: lane group qa>>1 == 1 works on the odd tap of the pair.  Two-deep software
        // pipeline with ping-pong register sets: the LDS reads of slab s+1 are issued before
        // the MFMAs of slab s (sched_barrier keeps hipcc from re-serialising them).
        int tl = qa >> 1;
        int dy = tl / kw, dx = tl - dy * kw;
        int sload = 0, g0 = 0, g1 = SG;                       // resident slab group [g0, g1)
        auto load_slab = [&](bf8 (&bp)[CBF_NU][NS], bf8 (&ap)[NCO][NWP]) {
          const bool tapok = tl < ntaps;
          const int toff = (dy * RW + dx) * CBF_PX_BYTES;
#pragma unroll
          for (int u = 0; u < CBF_NU; ++u) {
            const bool v = tapok && (((rmask[u] >> dy) & (cmask[u] >> dx) & 1u) != 0u);
            const int t = lbase[u] + toff;
            const char* __restrict__ src = reg + (v ? t : ((t & 0xF0) | zbase));   // its own slot of the zero block: no bank conflict
#pragma unroll
            for (int pz = 0; pz < NS; ++pz) bp[u][pz] = *(const bf8*)(src + (size_t)pz * plane);
          }
          const int sa = (sload < g1 ? sload : g1 - 1) - g0;  // slot inside the resident group
#pragma unroll
          for (int o = 0; o < NCO; ++o)
#pragma unroll
            for (int pz = 0; pz < NWP; ++pz) ap[o][pz] = *(const bf8*)(wl + (((size_t)o * SG + sa) * 3 + pz) * 1024 + lane * 16);
          ++sload; tl += 2; dx += 2;
          if (dx >= kw) { dx -= kw; ++dy; }
          if (dx >= kw) { dx -= kw; ++dy; }
        };
        auto mma_slab = [&](const bf8 (&bp)[CBF_NU][NS], const bf8 (&ap)[NCO][NWP]) {
#pragma unroll
          for (int o = 0; o < NCO; ++o)
#pragma unroll
            for (int u = 0; u < CBF_NU; ++u) acc[o][u] = chain_bf<NS>(ap[o], bp[u], acc[o][u]);
        };
        bf8 bA[CBF_NU][NS], aA[NCO][NWP], bB[CBF_NU][NS], aB[NCO][NWP];
        for (;;) {
          const int ng = g1 - g0;
          load_slab(bA, aA);
          int sl = 0;
          for (; sl + 2 <= ng; sl += 2) {
            load_slab(bB, aB);
            __builtin_amdgcn_sched_barrier(0);
            mma_slab(bA, aA);
            __builtin_amdgcn_sched_barrier(0);
            load_slab(bA, aA);
            __builtin_amdgcn_sched_barrier(0);
            mma_slab(bB, aB);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (sl < ng) mma_slab(bA, aA);
          if (g1 >= S) break;
          // next weight group: every wave is done with the resident fragments, then restage and rewind the
          // tap walk to the group's first slab (the pipeline ran one or two slabs past it)
          g0 = g1;
          g1 = g0 + SG < S ? g0 + SG : S;
          __syncthreads();
          dma_weights(cc, g0, g1);
          __syncthreads();
          sload = g0;
          tl = 2 * g0 + (qa >> 1);
          dy = tl / kw;
          dx = tl - dy * kw;
        }
      }
      // store this batch row's tile
      const size_t nimg = (size_t)a.B * d->g_px_off[g] + (size_t)b * P;
      const float inv = a.inv[br];
#pragma unroll
      for (int o = 0; o < NCO; ++o) {
        if (co0 + o < nco_tot) {
#pragma unroll
          for (int u = 0; u < CBF_NU; ++u) {
            if (u < nu && pok[u]) {
              const int ch = br * a.out_stride_br + 16 * (co0 + o);
              const f4 v = NS == 2 ? acc[o][u] * inv : acc[o][u];
              if (NS == 2 && a.out_p3) range_bad |= h2_bad4(v);
              if (a.out_p3) store_px<NS == 2 ? 2 : 3>((__bf16*)a.out + ((nimg + oidx[u]) * (a.OUTC >> 4) + (ch >> 4)) * PXE, lane >> 4, v);
              else *(f4*)((float*)a.out + (nimg + oidx[u]) * a.OUTC + ch + 4 * (lane >> 4)) = v;
            }
          }
        }
      }
    }
  }
  stamp(a.dbg, a.dbg_cap, wgid, 3);
  if (a.dbg != nullptr && threadIdx.x == 0 && wgid * 8 + 7 < a.dbg_cap) a.dbg[wgid * 8 + 7] = __builtin_amdgcn_s_memrealtime();
  if (NS == 2) raise_range_flag(a.range_flag, range_bad);
}

// ---------------------------------------------------------------- stages B / D, split engines, fast path
// The same convolution for the common geometry - one 16-channel input group and one output tile per branch
// (mid <= 16), kernel 3x3 / 5x5 / 7x7, all slabs' weight fragments resident in LDS - with the tap walk resolved
// at compile time.  In k_conv_bf every slab costs ~45 VALU instructions per wave (per-lane tap counters, two
// shifts + a compare + a select per pixel unit, a 32-bit multiply for the row offset, address adds per piece)
// for 9 MFMAs, and VALU issue, not the matrix pipe, sets its pace (PMC: 7.3 VALU per MFMA).  Here
//   * the slab loop is fully unrolled over the kernel's taps, so each lane half's (dy, dx) is a constant and
//     its LDS offset (dy*RW + dx)*32 two scalar operations;
//   * the per-pixel tap validity (conv zero padding at the grid border) is one bit per slab in a mask built
//     once per tile: bit s of vmask[u] = tap 2s + (lane half) is inside the grid for this lane's pixel;
//   * the region planes sit CBF_FAST_PLANE bytes apart (a constant), so the second piece is a ds_read offset.
// That leaves ~10 VALU per slab: v_bfe, v_add, v_mad per pixel unit and one select for the tap offset.
// A plane = the region's pixels + a 256-byte block of zeros (one 16-byte slot per LDS bank quad) that taps outside
// the grid are redirected to.  A redirected lane reads the slot of ITS OWN would-be address ((addr & 0xF0) in the
// block), so it keeps the bank quad it would have used and a ds_read_b128 lane group stays conflict-free; with one
// shared zero pixel every mixed group paid a 2-way conflict (PMC: 17 % of this kernel's LDS cycles, and LDS time
// is level with MFMA time here).
#define CBF_FAST_ZBASE (FTN_REGION_PX * CBF_PX_BYTES)
#define CBF_FAST_PLANE (CBF_FAST_ZBASE + 256)

// fragment pieces a slab occupies in LDS: f16x2 keeps A1 and A3 only (A2 is formed by the VALU)
template <int NS> struct CbfW { static constexpr int STR = NS == 2 ? 2 : 3; };

template <int NS, int KH, int KW>
__device__ __forceinline__ void conv_fast_row(f4 (&acc)[CBF_NU], const char* __restrict__ reg, const char* __restrict__ wlane,
                                               const int (&ld)[CBF_NU], const unsigned (&vmask)[CBF_NU],
                                               int RW32, bool half1) {
  constexpr int NWP = PxFmt<NS>::NW;
  constexpr int WSTR = CbfW<NS>::STR;
  constexpr int NT = KH * KW, S = (NT + 1) / 2;
  bf8 bA[CBF_NU][NS], aA[NWP], bB[CBF_NU][NS], aB[NWP];
  auto load_slab = [&](int s, bf8 (&bp)[CBF_NU][NS], bf8 (&ap)[NWP]) {
    // taps 2s (lanes 0-31) and 2s+1 (lanes 32-63); a tap index == NT (odd tap count) is masked off by vmask
    const int t0 = 2 * s, t1 = 2 * s + 1 < NT ? 2 * s + 1 : 2 * s;
    const int c0 = (t0 / KW) * RW32 + (t0 % KW) * CBF_PX_BYTES;
    const int c1 = (t1 / KW) * RW32 + (t1 % KW) * CBF_PX_BYTES;
    const int toff = half1 ? c1 : c0;
    // weights first: the slab's first MFMA needs them, and LDS reads return in issue order
    if constexpr (NS == 2) {
      // A2 = A1 * 2^-11 exactly (an fp16 multiply by a power of two, subnormals included - the packer forms it the
      // same way), so it is not read: LDS reads and MFMA time are level in this kernel (9 : 9 per slab), the VALU
      // is not, and four v_pk_mul_f16 replace one ds_read_b128 of every slab
      ap[0] = *(const bf8*)(wlane + (s * WSTR + 0) * 1024);
      ap[2] = *(const bf8*)(wlane + (s * WSTR + 1) * 1024);
      ap[1] = __builtin_bit_cast(bf8, __builtin_bit_cast(h8, ap[0]) * (_Float16)0.00048828125f);
    } else {
#pragma unroll
      for (int pz = 0; pz < NWP; ++pz) ap[pz] = *(const bf8*)(wlane + (s * WSTR + pz) * 1024);
    }
#pragma unroll
    for (int u = 0; u < CBF_NU; ++u) {
      const int t = ld[u] + toff;
      const bool v = ((vmask[u] >> s) & 1u) != 0u;
      const int addr = v ? t : ((t & 0xF0) | CBF_FAST_ZBASE);     // valid ? pixel + tap : its slot of the zero block
#pragma unroll
      for (int pz = 0; pz < NS; ++pz) bp[u][pz] = *(const bf8*)(reg + addr + pz * CBF_FAST_PLANE);
    }
  };
  auto mma_slab = [&](const bf8 (&bp)[CBF_NU][NS], const bf8 (&ap)[NWP]) {
#pragma unroll
    for (int u = 0; u < CBF_NU; ++u) acc[u] = chain_bf<NS>(ap, bp[u], acc[u]);
  };
  // One scheduling region per slab: the LDS reads of slab s+1 interleaved one-for-one with the MFMAs of slab s
  // (sched_group_barrier), so a read batch is never waited for right after its issue: with the reads fenced off
  // behind a sched_barrier hipcc waited lgkmcnt(0) before every other MFMA group, i.e. half of the LDS latency
  // was exposed (the 4-bit lgkmcnt cannot express "all but the 18 newest").
  auto interleave = [&]() {
    static_assert(CBF_NU == 3, "interleave patterns are written for three pixel units per wave");
    if constexpr (NS == 2) {                                  // 8 reads + 4 v_pk_mul_f16 : 9 MFMAs
#pragma unroll
      for (int k = 0; k < 8; ++k) {                           // MFMA first: the wait in front of it then covers
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // only reads issued a whole slab earlier
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    } else if constexpr (NS == 3) {                           // 12 reads : 18 MFMAs
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    } else {                                                  // 4 reads : 3 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  };
  load_slab(0, bA, aA);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < S; s += 2) {
    if (s + 1 < S) load_slab(s + 1, bB, aB);
    mma_slab(bA, aA);
    if (s + 1 < S) interleave();
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < S) {
      if (s + 2 < S) load_slab(s + 2, bA, aA);
      mma_slab(bB, aB);
      if (s + 2 < S) interleave();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Launch shape: ONE workgroup per CU, each bound to one branch for its whole life; a branch gets a share of the
// workgroups proportional to its cost (host: ~3.1 k + 0.35 k cycles per K-32 slab and batch row) and a workgroup
// a contiguous range of that branch's (tile, batch row) sequence.  The branch's weight fragments (75 KB for 7x7)
// are DMA'd once per workgroup instead of once per 8 rows, and every CU finishes at about the same time; the
// (tile, 8-row chunk, branch) grid of k_conv_bf runs 480 unequal workgroups (49 / 25 / 9 taps) on 256 CUs in
// roughly 1.4 rounds - 92 us for 66 us of work (tools/stamps.py).
// NCI = 16-channel input groups per branch (mid 16: 1; mid 32: 2, late round 3).  With two, a workgroup is bound to a
// (branch, output tile) pair - a "virtual branch" - keeps both input groups' fragment sets in LDS and walks the
// (batch row, input group) sequence through the same two region buffers: the second group's products add into the
// first's accumulators, the row's outputs are stored once.
template <int NS, int NCI>
__global__ __launch_bounds__(512) void k_conv_bf_fast(ConvBfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  constexpr int PXE = PxFmt<NS>::ELEMS;
  constexpr int plane = CBF_FAST_PLANE;
  constexpr int WSTR = CbfW<NS>::STR;
  const FtnDesc* __restrict__ d = a.desc;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, qa = lane >> 4;
  bool range_bad = false;                                       // f16x2: an output left the fp16 range
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  int vb = 0;
  while (vb + 1 < a.nvb && (int)blockIdx.x >= a.wg_off[vb + 1]) ++vb;
  const int wgi = (int)blockIdx.x - a.wg_off[vb], nwg = a.wg_off[vb + 1] - a.wg_off[vb];
  const int nco = a.cout >> 4, br = vb / nco, cot = vb - br * nco;   // branch, output tile of the branch
  const int G = d->n_groups, tiles_total = d->tiles_per_row;
  const long long rows_total = (long long)tiles_total * a.B;
  const int row_lo = (int)(rows_total * wgi / nwg), row_hi = (int)(rows_total * (wgi + 1) / nwg);
  const size_t wgid = blockIdx.x;
  stamp(a.dbg, a.dbg_cap, wgid, 0);
  if (row_lo >= row_hi) return;
  const int kh = a.kh[br], kw = a.kw[br], hy = kh >> 1, hx = kw >> 1, ntaps = kh * kw;
  const int S = (ntaps + 1) >> 1;
  if (a.dbg != nullptr && threadIdx.x == 0 && wgid * 8 + 7 < a.dbg_cap) { a.dbg[wgid * 8 + 6] = __builtin_amdgcn_s_memrealtime(); a.dbg[wgid * 8 + 4] = (unsigned long long)ntaps; a.dbg[wgid * 8 + 5] = ((unsigned long long)row_lo << 32) | (unsigned)row_hi; }
  char* __restrict__ wl = ldsb;
  char* __restrict__ rbuf0 = ldsb + (size_t)NCI * S * WSTR * 1024;   // behind THIS (branch, tile)'s weight fragments
  static_assert(CBF_FAST_ZBASE % 256 == 0, "the zero block must start on a 256-byte boundary");
  if (threadIdx.x < 2 * NS * 16) {                           // zero blocks of both region buffers
    const int pl = threadIdx.x >> 4;
    *(f4*)(rbuf0 + (size_t)(pl / NS) * a.region_bytes + (size_t)(pl % NS) * plane + CBF_FAST_ZBASE + (threadIdx.x & 15) * 16) = f4{0.f, 0.f, 0.f, 0.f};
  }
  {                                                            // every slab of the branch's one output tile, once
    // f16x2: piece 1 (A2 = A1 * 2^-11) is formed by the VALU in the slab loop, so it is neither fetched nor given
    // room in LDS (packed layout: [cin group][cout tile][slab][3 pieces] x 1 KB; LDS: [cin group][slab][WSTR])
    const int npc = S * WSTR;
#pragma unroll
    for (int gi = 0; gi < NCI; ++gi) {
      const __bf16* __restrict__ src = a.W[br] + (size_t)(gi * nco + cot) * S * 3 * 512;
      for (int g = wv; g < npc; g += 8) {
        const int f = NS == 2 ? (g >> 1) * 3 + (g & 1) * 2 : g;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)f * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(wl + (size_t)(gi * npc + g) * 1024), 16, 0, 0);
      }
    }
  }
  const f4 bv = *(const f4*)(a.bias + br * a.out_stride_br + 16 * cot + 4 * qa);
  const float inv = a.inv[br];
  const int in_groups = a.INC >> 4;
  const int btL = a.bt_L;
  const int h1 = qa >> 1;                                     // this lane's tap of every pair
  const unsigned kmh = (1u << kh) - 1u, kmw = (1u << kw) - 1u;
  bool first_tile = true;

  for (int row = row_lo; row < row_hi;) {
    const int bx = row / a.B;                                  // tile (of the per-row tile list), then its batch rows
    const int b_begin = row - bx * a.B;
    const int b_end = min(a.B, b_begin + (row_hi - row));
    row += b_end - b_begin;
    int g = 0;
    for (int gg = 1; gg < G; ++gg)
      if (bx >= d->g_tile_off[gg]) g = gg;
    const int tix = bx - d->g_tile_off[g];
    const int ntx = d->g_ntx[g];
    const int ty = tix / ntx, tx = tix - ty * ntx;
    const int p = d->g_period[g], cycles = d->g_cycles[g];
    const int P = d->g_px_off[g + 1] - d->g_px_off[g];
    const int r0 = ty * d->g_th[g], c0 = tx * d->g_tw[g];
    const int th = min(d->g_th[g], cycles - r0), tw = min(d->g_tw[g], p - c0);
    const int R0 = max(0, r0 - hy), R1 = min(cycles, r0 + th + hy);
    const int C0 = max(0, c0 - hx), C1 = min(p, c0 + tw + hx);
    const int RW = C1 - C0, RH = R1 - R0;
    const int npx = th * tw, nunits = (npx + 15) >> 4;
    const float inv_tw = 1.0f / (float)tw, inv_rw = 1.0f / (float)RW;
    const int nchunks16 = RH * RW * 2;
    const int ppp = (nchunks16 + 63) >> 6;
    // descriptor values the row loop needs, read once per tile (a load inside the loop is a full round trip on
    // the critical path of every batch row, and its s_waitcnt vmcnt(0) also waits for everything else in flight)
    const size_t img0 = (size_t)a.B * d->g_px_off[g];
    // Region DMA: which 16-byte chunk a lane fetches for piece k of this wave depends on the tile only, so its
    // offset (relative to the batch row's first pixel, or to the shared pad row for the live zero pixels t >= L)
    // and its LDS slot are worked out once per tile; a batch row then costs an add, a select and the load per piece
    // (generating the addresses in the row loop was ~65 instructions per piece, most of that loop's fixed cost).
    constexpr int KPMAX = (NS * ((FTN_REGION_PX * 2 + 63) / 64) + 7) / 8;
    int poff[KPMAX], pdst[KPMAX];
    unsigned ppad = 0u;
    const int npc = NS * ppp;
#pragma unroll
    for (int k = 0; k < KPMAX; ++k) {
      const int pc = wv + 8 * k;
      const int pcc = pc < npc ? pc : 0;
      const int pz = pcc / ppp, pi = pcc - pz * ppp;
      int ci = pi * 64 + lane;
      if (ci >= nchunks16) ci = nchunks16 - 1;
      const int sp = ci >> 1, half = ci & 1;
      const int rr = (int)(((float)sp + 0.5f) * inv_rw), cx = sp - rr * RW;
      const int tpx = (R0 + rr) * p + C0 + cx;
      const bool pad = btL > 0 && tpx >= btL;
      poff[k] = (pad ? 0 : tpx * in_groups * PXE) + pz * 16 + half * 8;
      ppad |= (pad ? 1u : 0u) << k;
      pdst[k] = __builtin_amdgcn_readfirstlane(pz * plane + pi * 1024);
    }
    const __bf16* __restrict__ in_br = a.in + (size_t)(br * a.in_stride_br) * PXE;
    const __bf16* __restrict__ src_pad = in_br + (size_t)a.B * btL * in_groups * PXE;
    const size_t row_stride = (size_t)(btL > 0 ? btL : P) * in_groups * PXE;
    const __bf16* __restrict__ src0 = in_br + (btL > 0 ? (size_t)0 : img0 * in_groups * PXE);
    auto dma_region = [&](int b, int gi, int buf) {           // input group gi of batch row b
      const __bf16* __restrict__ src = src0 + (size_t)b * row_stride + gi * PXE;
      char* __restrict__ dstb = rbuf0 + (size_t)buf * a.region_bytes;
#pragma unroll
      for (int k = 0; k < KPMAX; ++k) {
        if (wv + 8 * k < npc) {
          const __bf16* __restrict__ rowp = ((ppad >> k) & 1u) ? src_pad + gi * PXE : src;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowp + poff[k]),
                                           (__attribute__((address_space(3))) void*)(dstb + pdst[k]), 16, 0, 0);
        }
      }
    };
    // the tile's first row is requested before the per-lane bookkeeping, so it lands meanwhile (the previous
    // tile's readers must be done with the buffers first)
    if (!first_tile) __syncthreads();
    first_tile = false;
    dma_region(b_begin, 0, 0);
    // per-lane pixel bookkeeping, once per tile.  Tap validity (conv zero padding at the grid border): bit s of
    // vmask[u] = tap 2s + h1 lies inside the grid for this lane's pixel, from a row mask and a column mask and a
    // division-free walk over the taps (a runtime tl / kw per slab and unit cost 20 k cycles of a 7x7 tile's
    // 34 k cycle prologue, tools/stamps.py)
    const int wrot = (wave + b_begin) & 7;
    const int nu = nunits > wrot ? (nunits - wrot + 7) >> 3 : 0;
    int ld[CBF_NU], oidx[CBF_NU];
    unsigned ooff[CBF_NU];
    unsigned vmask[CBF_NU];
    bool pok[CBF_NU];
#pragma unroll
    for (int u = 0; u < CBF_NU; ++u) {
      int idx = (wrot + 8 * u) * 16 + j;
      pok[u] = idx < npx;
      if (!pok[u]) idx = 0;
      const int r = (int)(((float)idx + 0.5f) * inv_tw), c = idx - r * tw;
      const int ri = r0 + r, ci = c0 + c;
      ld[u] = ((ri - R0 - hy) * RW + (ci - C0 - hx)) * CBF_PX_BYTES + (qa & 1) * 16;
      oidx[u] = ri * p + ci;
      {
        const int ch = br * a.out_stride_br + 16 * cot;
        ooff[u] = a.out_p3 ? (unsigned)((oidx[u] * (a.OUTC >> 4) + (ch >> 4)) * PXE)
                           : (unsigned)(oidx[u] * a.OUTC + ch + 4 * qa);
      }
      // taps dy with 0 <= ri + dy - hy < cycles are the bits [lo, hi) of the row mask (same for columns)
      const int rlo = max(0, hy - ri), rhi = min(kh, cycles + hy - ri);
      const int clo = max(0, hx - ci), chi = min(kw, p + hx - ci);
      const unsigned rm = (rhi > rlo) ? ((kmh >> (kh - rhi)) & (kmh << rlo)) & kmh : 0u;
      const unsigned cm = (chi > clo) ? ((kmw >> (kw - chi)) & (kmw << clo)) & kmw : 0u;
      // bit s = tap 2s + h1 is inside the grid.  Row dy of the kernel (kw odd: its first tap has parity dy) holds the
      // taps of this lane half at dx = par, par + 2, ... with par = (dy ^ h1) & 1, i.e. every other bit of the column
      // mask, compressed, at slab (dy kw + par - h1) / 2: ~70 instructions per pixel unit instead of a 10-instruction
      // step per slab (the 7x7 prologue spent 9 k of its 25 k cycles in that loop; tools/stamps.py)
      const unsigned ce = (cm & 1u) | ((cm >> 1) & 2u) | ((cm >> 2) & 4u) | ((cm >> 3) & 8u);
      const unsigned co = ((cm >> 1) & 1u) | ((cm >> 2) & 2u) | ((cm >> 3) & 4u) | ((cm >> 4) & 8u);
      unsigned m = 0u;
#pragma unroll
      for (int dy = 0; dy < 7; ++dy) {                         // kh <= 7 here (fast path: 3x3 / 5x5 / 7x7)
        const int par = (dy ^ h1) & 1;
        const unsigned bits = par ? co : ce;
        if (dy < kh && ((rm >> dy) & 1u)) m |= bits << ((dy * kw + par - h1) >> 1);
      }
      vmask[u] = pok[u] ? m : 0u;
    }
    int it = 0;
    int keep = 0;                                             // output stores issued behind the newest region DMA
    const int nst_row = a.dbg != nullptr ? 99 : nu * (a.out_p3 ? 2 : 1);
    const char* __restrict__ wlane = wl + lane * 16;
    for (int b = b_begin; b < b_end; ++b) {
      f4 acc[CBF_NU];
#pragma unroll
      for (int u = 0; u < CBF_NU; ++u) acc[u] = bv;
      // the (row, input group) items walk the two region buffers in turn; the item behind this one is requested now
      // (rolled: one copy of the unrolled slab loops)
#pragma unroll 1
      for (int gi = 0; gi < NCI; ++gi) {
        if (!((a.abl & 4) && (b > b_begin || gi > 0))) barrier_keep_vm(keep);   // this item (and, the first time, the weights) have landed
        keep = gi == NCI - 1 ? __builtin_amdgcn_readfirstlane(nst_row) : 0;   // stores only follow a row's last group
        if (gi == 0) {
          if (b == b_begin) stamp(a.dbg, a.dbg_cap, wgid, 1);
          if (b == b_begin + 1) stamp(a.dbg, a.dbg_cap, wgid, 2);
        }
        if (!(a.abl & 2)) {
          if (gi + 1 < NCI) dma_region(b, gi + 1, (it + 1) & 1);
          else if (b + 1 < b_end) dma_region(b + 1, 0, (it + 1) & 1);
        }
        const char* __restrict__ reg = rbuf0 + (size_t)((a.abl & 2) ? 0 : (it & 1)) * a.region_bytes;
        ++it;
        const char* __restrict__ wg_ = wlane + (size_t)gi * S * WSTR * 1024;
        if (a.abl & 8) {}
        else if (kw == 7) conv_fast_row<NS, 7, 7>(acc, reg, wg_, ld, vmask, RW * CBF_PX_BYTES, h1 != 0);
        else if (kw == 5) conv_fast_row<NS, 5, 5>(acc, reg, wg_, ld, vmask, RW * CBF_PX_BYTES, h1 != 0);
        else conv_fast_row<NS, 3, 3>(acc, reg, wg_, ld, vmask, RW * CBF_PX_BYTES, h1 != 0);
      }
      // uniform row base + per-lane offsets fixed for the tile (ooff)
      const size_t nimg = img0 + (size_t)b * P;
      if ((a.abl & 1) && acc[0][0] != 12345.678f) {}
      else if (a.out_p3) {
        __bf16* __restrict__ ob = (__bf16*)a.out + nimg * (size_t)(a.OUTC >> 4) * PXE;
#pragma unroll
        for (int u = 0; u < CBF_NU; ++u)
          if (u < nu && pok[u]) {
            const f4 v = NS == 2 ? acc[u] * inv : acc[u];
            if (NS == 2) range_bad |= h2_bad4(v);
            store_px<NS == 2 ? 2 : 3>(ob + ooff[u], qa, v);
          }
      } else {
        float* __restrict__ ob = (float*)a.out + nimg * (size_t)a.OUTC;
#pragma unroll
        for (int u = 0; u < CBF_NU; ++u)
          if (u < nu && pok[u]) *(f4*)(ob + ooff[u]) = NS == 2 ? acc[u] * inv : acc[u];
      }
    }
  }
  stamp(a.dbg, a.dbg_cap, wgid, 3);
  if (a.dbg != nullptr && threadIdx.x == 0 && wgid * 8 + 7 < a.dbg_cap) a.dbg[wgid * 8 + 7] = __builtin_amdgcn_s_memrealtime();
  if (NS == 2) raise_range_flag(a.range_flag, range_bad);
}

// ---------------------------------------------------------------- small elementwise stages
// single-conv mode, stage A: a[n][CP] = zero-extended x
__global__ void k_embed(const float* __restrict__ x, float* __restrict__ out, int B, int L, int C, int CP,
                        const FtnDesc* guard_src, FtnDesc* guard_dst, int guard_groups, int guard_px) {
  if (blockIdx.x == 0) guard_desc(guard_src, guard_dst, guard_groups, guard_px);
  // rows n = b*L + t of the window, channels zero-padded to CP, plus one all-zero pad row n = B*L
  // (the live zero pixels t >= L of every period grid, :1017); the conv folds them (ConvArgs.bt_L)
  const int cq = CP >> 2;
  const long long rows = (long long)B * L + 1;
  const long long total = rows * cq;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long n = e / cq;
    const int c = (int)(e - n * cq) * 4;
    *(f4*)(out + (size_t)n * CP + c) = load_x4<false>(n < rows - 1 ? x + (size_t)n * C : nullptr, c, C);
  }
}

// stages E + F fused:  y = x + sum_g w[b,g] * ( act(W_out2 m'_g + b) + r_g )[:L]
// (reference :1063-1069 delta, :1075-1092 weighted sum in group order, :818 residual).
// One wave owns 2 units of 16 consecutive (b,t) positions and walks the groups in
// ascending order (same summation order as the reference; deterministic, no
// atomics); the tail pixels t >= L of every grid are never touched.  HBM/L2-bound:
// reads m' and r once, x once, writes y once.
// (OutArgs, rnd_act4 and ln_tiles live in ftn_out.h, shared with stage_out.hip)

// Standalone form of the same epilogue for the shapes k_out does not fuse (d_model > 64) and for
// blocks that return x unchanged: out = LayerNorm_C(x + (nw - x)); one wave per row, in place allowed.
__global__ __launch_bounds__(256) void k_resid_ln(const float* __restrict__ x, const float* nw, float* out,
                                                  const float* __restrict__ g, const float* __restrict__ b,
                                                  float eps, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  const float* nr = nw + row * C;
  float* orow = out + row * C;
  constexpr int MAXV = 8;                      // channels cached in registers: C <= 512; beyond that re-read
  float v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    v[i] = 0.f;
    if (c < C) { const float xv = xr[c]; v[i] = xv + (nr[c] - xv); s += v[i]; }
  }
  for (int c = lane + 64 * MAXV; c < C; c += 64) { const float xv = xr[c]; s += xv + (nr[c] - xv); }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
  const float mean = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + 64 * i < C) { const float dv = v[i] - mean; ss += dv * dv; }
  for (int c = lane + 64 * MAXV; c < C; c += 64) { const float xv = xr[c]; const float dv = xv + (nr[c] - xv) - mean; ss += dv * dv; }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) ss += __shfl_xor(ss, m);
  const float rstd = 1.0f / sqrtf(ss / (float)C + eps);
  // the tail (C > 512) must be produced before the cached part overwrites an in-place row
  for (int c = lane + 64 * MAXV; c < C; c += 64) { const float xv = xr[c]; orow[c] = (xv + (nr[c] - xv) - mean) * rstd * g[c] + b[c]; }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < C) orow[c] = (v[i] - mean) * rstd * g[c] + b[c];
  }
}

// Fast path (FAST): K <= 48 and <= 4 output tiles (d_model <= 64, nbr*mid <= 48): the 12
// weight fragments are group-independent and live in registers; per group the wave only
// streams its m' and r rows, with the next group's m' rows requested one group ahead.
template <int ACT, bool XVEC, bool IDENT, bool FAST, int NPX = 2>
__global__ __launch_bounds__(256, NPX == 1 ? 3 : 2) void k_out(OutArgs a) {
  const FtnDesc* __restrict__ d = a.desc;
  const int total = a.B * a.L;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * (16 * NPX);
  if (n0 >= total) return;
  int bb[NPX], tt[NPX];
  bool ok[NPX];
#pragma unroll
  for (int u = 0; u < NPX; ++u) {
    int n = n0 + 16 * u + j;
    ok[u] = n < total;
    if (!ok[u]) n = total - 1;
    bb[u] = n / a.L;
    tt[u] = n - bb[u] * a.L;
  }
  const int G = d->n_groups, KM = a.KM, CP = a.CP, n_ot = CP >> 4;
  if (FAST) {
    const int nKM = KM >> 4;
    f4 aw[4][3], bias[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      bias[o] = o < n_ot ? *(const f4*)(a.bias + 16 * o + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s)
        aw[o][s] = (o < n_ot && s < nKM) ? *(const f4*)(a.W + (size_t)(16 * o + j) * KM + 16 * s + 4 * q)
                                         : f4{0.f, 0.f, 0.f, 0.f};
    }
    f4 yacc[4][NPX], mc[3][NPX];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int u = 0; u < NPX; ++u) yacc[o][u] = f4{0.f, 0.f, 0.f, 0.f};
    auto pix = [&](int g, int u) -> size_t {
      const int P = d->g_px_off[g + 1] - d->g_px_off[g];
      return (size_t)a.B * d->g_px_off[g] + (size_t)bb[u] * P + tt[u];
    };
    auto load_m = [&](int g, f4 (&dst)[3][NPX]) {
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        const float* __restrict__ row = a.m + pix(g, u) * KM + 4 * q;
#pragma unroll
        for (int s = 0; s < 3; ++s) dst[s][u] = s < nKM ? *(const f4*)(row + 16 * s) : f4{0.f, 0.f, 0.f, 0.f};
      }
    };
    if (G > 0) load_m(0, mc);
    float wsum[NPX];
#pragma unroll
    for (int u = 0; u < NPX; ++u) wsum[u] = 0.f;
    for (int g = 0; g < G; ++g) {
      f4 rr[4][NPX], mn[3][NPX];
      float w[NPX];
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        if (!a.r_summed) {
          const float* __restrict__ row = a.R + pix(g, u) * CP + 4 * q;
#pragma unroll
          for (int o = 0; o < 4; ++o) rr[o][u] = o < n_ot ? *(const f4*)(row + 16 * o) : f4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int o = 0; o < 4; ++o) rr[o][u] = f4{0.f, 0.f, 0.f, 0.f};
        }
        w[u] = a.wts[(size_t)bb[u] * FTN_KMAX + g];
        wsum[u] += w[u];
      }
      load_m(g + 1 < G ? g + 1 : g, mn);
      f4 z[4][NPX];
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int u = 0; u < NPX; ++u) z[o][u] = bias[o];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int u = 0; u < NPX; ++u) z[o][u] = mfma16(aw[o][s][e], mc[s][u][e], z[o][u]);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
          if (a.act_dtype == 0) yacc[o][u] += (act4<ACT>(z[o][u]) + rr[o][u]) * w[u];
          else yacc[o][u] += rnd_act4(rnd_act4(act4<ACT>(z[o][u]) + rr[o][u], a.act_dtype) * w[u], a.act_dtype);
        }
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < NPX; ++u) mc[s][u] = mn[s][u];
    }
    const bool ln = a.ln_g != nullptr;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o < n_ot) {
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
          const int ch = 16 * o + 4 * q;
          const size_t e0 = ((size_t)bb[u] * a.L + tt[u]) * a.C + ch;
          f4 xv = {0.f, 0.f, 0.f, 0.f};
          if (XVEC) {
            if (ch < a.C) xv = *(const f4*)(a.x + e0);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (ch + r < a.C) xv[r] = a.x[e0 + r];
          }
          if (a.r_summed && ch < CP)                                     // the group-summed residual of k_mlp_pos
            yacc[o][u] += *(const f4*)(a.R + ((size_t)bb[u] * a.L + tt[u]) * CP + ch);
          if (a.r_keeps_x) yacc[o][u] = yacc[o][u] - xv * wsum[u];      // the x that stage C left inside every R_g
          const f4 nv = a.act_dtype == 0 ? xv + yacc[o][u] : rnd_act4(xv + rnd_act4(yacc[o][u], a.act_dtype), a.act_dtype);
          yacc[o][u] = ln ? xv + (nv - xv) : nv;
        }
      }
    }
    if (ln) ln_tiles<4, NPX>(yacc, n_ot, a.C, q, a.ln_g, a.ln_b, a.ln_eps);
    if (a.range_flag != nullptr) {
      bool bad = false;
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int u = 0; u < NPX; ++u)
          if (o < n_ot && ok[u]) bad |= not_finite4(yacc[o][u]);
      raise_range_flag(a.range_flag, bad);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o < n_ot) {
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
          if (!ok[u]) continue;
          const int ch = 16 * o + 4 * q;
          const size_t e0 = ((size_t)bb[u] * a.L + tt[u]) * a.C + ch;
          if (XVEC) {
            if (ch < a.C) *(f4*)(a.y + e0) = yacc[o][u];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (ch + r < a.C) a.y[e0 + r] = yacc[o][u][r];
          }
        }
      }
    }
    return;
  }
  for (int og = 0; og < n_ot; og += 4) {
    f4 yacc[4][NPX];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int u = 0; u < NPX; ++u) yacc[o][u] = f4{0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < G; ++g) {
      const int P = d->g_px_off[g + 1] - d->g_px_off[g];
      size_t pn[NPX];
      float w[NPX];
#pragma unroll
      for (int u = 0; u < NPX; ++u) {
        pn[u] = (size_t)a.B * d->g_px_off[g] + (size_t)bb[u] * P + tt[u];
        w[u] = a.wts[(size_t)bb[u] * FTN_KMAX + g];
      }
      f4 z[4][NPX];
      if (IDENT) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
          for (int u = 0; u < NPX; ++u) {
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (og + o < n_ot) v = *(const f4*)(a.m + pn[u] * KM + 16 * (og + o) + 4 * q);
            z[o][u] = v;
          }
      } else {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          f4 bv = {0.f, 0.f, 0.f, 0.f};
          if (og + o < n_ot) bv = *(const f4*)(a.bias + 16 * (og + o) + 4 * q);
#pragma unroll
          for (int u = 0; u < NPX; ++u) z[o][u] = bv;
        }
        for (int s = 0; s < KM; s += 16) {
          f4 bf[NPX];
#pragma unroll
          for (int u = 0; u < NPX; ++u) bf[u] = *(const f4*)(a.m + pn[u] * KM + s + 4 * q);
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            if (og + o < n_ot) {
              const f4 af = *(const f4*)(a.W + (size_t)(16 * (og + o) + j) * KM + s + 4 * q);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < NPX; ++u) z[o][u] = mfma16(af[e], bf[u][e], z[o][u]);
            }
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (og + o < n_ot) {
#pragma unroll
          for (int u = 0; u < NPX; ++u) {
            const f4 r = *(const f4*)(a.R + pn[u] * CP + 16 * (og + o) + 4 * q);
            const f4 dl = act4<ACT>(z[o][u]) + r;
            if (a.act_dtype == 0) yacc[o][u] += dl * w[u];
            else yacc[o][u] += rnd_act4(rnd_act4(dl, a.act_dtype) * w[u], a.act_dtype);
          }
        }
      }
    }
    if (a.range_flag != nullptr) {
      bool bad = false;
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int u = 0; u < NPX; ++u)
          if (og + o < n_ot && ok[u]) bad |= not_finite4(yacc[o][u]);
      raise_range_flag(a.range_flag, bad);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (og + o < n_ot) {
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
          if (!ok[u]) continue;
          const int ch = 16 * (og + o) + 4 * q;
          const size_t e0 = ((size_t)bb[u] * a.L + tt[u]) * a.C + ch;
          if (XVEC) {
            if (ch < a.C) {
              const f4 xv = *(const f4*)(a.x + e0);
              *(f4*)(a.y + e0) = a.act_dtype == 0 ? xv + yacc[o][u] : rnd_act4(xv + rnd_act4(yacc[o][u], a.act_dtype), a.act_dtype);
            }
          } else {
            const f4 sr = rnd_act4(yacc[o][u], a.act_dtype);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (ch + r < a.C) {
                const float t = a.x[e0 + r] + sr[r];
                a.y[e0 + r] = a.act_dtype == 1 ? (float)(__bf16)t : (a.act_dtype == 2 ? (float)(_Float16)t : t);
              }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- stage timing
// Optional hipEvent brackets around the stages of ftn_timesblock_forward, kept in
// a pool so that nothing synchronises while a timed region runs; read back (with
// one synchronise) by ftn_stage_times.  Used by bench.py for the roofline figures.
#define FTN_NSTAGE 6
#define FTN_PROF_CALLS 512
static struct StageProf {
  bool on = false;
  bool created = false;
  int calls = 0;        // forwards recorded
  int every = 1;        // record every `every`-th forward
  int seen = 0;         // forwards seen since ftn_stage_timing(enable)
  hipEvent_t ev[FTN_PROF_CALLS][FTN_NSTAGE + 1];
  bool sampling() const { return on && calls < FTN_PROF_CALLS && seen % every == 0; }
} g_prof;

static void prof_mark(int stage, hipStream_t st) {
  if (g_prof.sampling()) (void)hipEventRecord(g_prof.ev[g_prof.calls][stage], st);
}

extern "C" int ftn_debug_stamps(void* buf_dev, size_t n_u64, int which) {
  g_stamp_buf = (unsigned long long*)buf_dev;
  g_stamp_cap = buf_dev ? n_u64 : 0;
  g_stamp_which = which;
  return 0;
}

extern "C" int ftn_stage_timing(int enable) {
  if (enable && !g_prof.created) {
    for (int c = 0; c < FTN_PROF_CALLS; ++c)
      for (int s = 0; s <= FTN_NSTAGE; ++s) {
        hipError_t e = hipEventCreate(&g_prof.ev[c][s]);
        if (e != hipSuccess) { ftn_set_error("hipEventCreate: %s", hipGetErrorString(e)); return (int)e; }
      }
    g_prof.created = true;
  }
  g_prof.on = enable != 0;
  g_prof.every = enable > 1 ? enable : 1;
  g_prof.calls = 0;
  g_prof.seen = 0;
  return 0;
}

extern "C" int ftn_stage_times(float* ms_sum, int nstage, int* ncalls) {
  FTN_CHECK_ARG(ms_sum && ncalls && nstage == FTN_NSTAGE, "ftn_stage_times: expects %d stages", FTN_NSTAGE);
  for (int s = 0; s < FTN_NSTAGE; ++s) ms_sum[s] = 0.f;
  *ncalls = g_prof.calls;
  for (int c = 0; c < g_prof.calls; ++c) {
    hipError_t e = hipEventSynchronize(g_prof.ev[c][FTN_NSTAGE]);
    if (e != hipSuccess) { ftn_set_error("hipEventSynchronize: %s", hipGetErrorString(e)); return (int)e; }
    for (int s = 0; s < FTN_NSTAGE; ++s) {
      float ms = 0.f;
      e = hipEventElapsedTime(&ms, g_prof.ev[c][s], g_prof.ev[c][s + 1]);
      if (e != hipSuccess) { ftn_set_error("hipEventElapsedTime: %s", hipGetErrorString(e)); return (int)e; }
      ms_sum[s] += ms;
    }
  }
  return 0;
}

// ---------------------------------------------------------------- host side
// Upper bound of FtnDesc.total_px (grid pixels per batch row): the caller's own bound when it has one
// (ftn_selector_px_bound for the native selector, the descriptor's exact total_px for a host-built one),
// otherwise the exact worst case over any `max_groups` distinct valid periods of a window of length L
// (P_g = L + (-L mod p); periods just below L pad to almost 2L).
static int worst_px_per_row(int L, int max_groups, int px_bound) {
  if (px_bound > 0) return px_bound;
  int best[FTN_KMAX] = {0};
  for (int p = 1; p < L; ++p) {
    int v = L + (p - (L % p)) % p;
    for (int s = 0; s < max_groups; ++s)
      if (v > best[s]) { int tmp = best[s]; best[s] = v; v = tmp; }
  }
  long long w = 0;
  for (int s = 0; s < max_groups; ++s) w += best[s];
  return w > 0 ? (int)w : L;
}

static void worst_tiles(int L, int max_groups, int* tiles_per_row) {
  // exact worst case of the max_groups largest tile counts over all valid periods
  int best[FTN_KMAX] = {0};
  for (int p = 1; p < L; ++p) {
    int pad = (p - (L % p)) % p, cyc = (L + pad) / p;
    if (cyc < 2) continue;
    int tw, th, ntx, nty;
    ftn_tile_geometry(cyc, p, &tw, &th, &ntx, &nty);
    int v = ntx * nty;
    for (int s = 0; s < max_groups; ++s)
      if (v > best[s]) { int tmp = best[s]; best[s] = v; v = tmp; }
  }
  int sum = 0;
  for (int s = 0; s < max_groups; ++s) sum += best[s];
  *tiles_per_row = sum > 0 ? sum : 1;
}

// Worst staged region (pixels) of a kh x kw conv over every valid period of a
// window of length L, with the tile geometry of ftn_tile_geometry.
static int conv_region_px(int L, int kh, int kw) {
  int worst = 1;
  const int hy = kh / 2, hx = kw / 2;
  for (int p = 1; p < L; ++p) {
    int pad = (p - (L % p)) % p, cyc = (L + pad) / p;
    if (cyc < 2) continue;
    int tw, th, ntx, nty;
    ftn_tile_geometry(cyc, p, &tw, &th, &ntx, &nty);
    int rw = tw + 2 * hx; if (rw > p) rw = p;
    int rh = th + 2 * hy; if (rh > cyc) rh = cyc;
    if (rw * rh > worst) worst = rw * rh;
  }
  return worst;
}

// Stage C runs as separate generic pointwise launches (k_pw) when the fused kernels cannot take the shape: more
// than 16 output tiles (nbr*mid/16 + d_model/16), or a hidden chunk's weight fragments not fitting LDS twice.
static bool stagec_generic(const FtnPlan* pl) {
  if (pl->mode != 0) return false;
  const int CA = pl->nbr * pl->MP;
  const int n_ot = CA / 16 + (pl->res2 ? pl->CP / 16 : 0);
  return n_ot > 16 || pl->cfrag_per_chunk <= 0 || (size_t)pl->cfrag_per_chunk * 1024 * 2 > 160 * 1024;
}

struct WsLayout {
  size_t offA, off0, off1, off2, off3, total;
  int c0, c1;  // channel counts of buf0 / buf1
};

#define FTN_WS_HEAD 1024   // sanitised copy of the descriptor (k_guard) at the head of the workspace

static WsLayout ws_layout(const FtnPlan* pl, int B, int L, int max_groups, int px_bound) {
  WsLayout w;
  const size_t N = (size_t)B * worst_px_per_row(L, max_groups, px_bound);
  const int CA = pl->nbr * pl->MP;
  w.c0 = pl->mode == 0 ? CA : pl->CP;                 // a / a'   (mode 1: padded x, then m')
  w.c1 = pl->mode == 0 ? CA : pl->FP;                 // m / m'   (mode 1: conv1 output)
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  // stage A output: one row per window position (+ pad row), shared by all period groups
  const int bpv = (pl->engine == 0 || pl->engine == 3 || pl->mode != 0) ? 4 : 6;       // fp32 / H2: 4 bytes per value, P3: 6
  w.offA = FTN_WS_HEAD;
  w.off0 = al(w.offA + ((size_t)B * L + 1) * (pl->mode == 0 ? CA : pl->CP) * bpv);
  w.off1 = al(w.off0 + N * w.c0 * bpv);
  w.off2 = al(w.off1 + N * w.c1 * bpv);               // R [N][CP]
  w.off3 = al(w.off2 + N * pl->CP * 4);               // G [N][FP] (mode 1)
  w.total = (pl->mode == 0 && !stagec_generic(pl)) ? w.off3 : al(w.off3 + N * pl->FP * 4);
  return w;
}

extern "C" size_t ftn_timesblock_workspace_bytes(const FtnPlan* plan, int B, int L, int max_groups, int px_bound) {
  if (!plan || B < 1 || L < 2 || max_groups < 1 || max_groups > FTN_KMAX || px_bound < 0) return 0;
  if ((long long)B * worst_px_per_row(L, max_groups, px_bound) > 0x7fffffffLL) return 0;   // flat pixel index is an int
  return ws_layout(plan, B, L, max_groups, px_bound).total;
}


template <int NCO>
static int launch_conv_t(const ConvArgs& ca, dim3 grid, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv<NCO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_conv): %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(k_conv<NCO>, grid, dim3(256), lds, st, ca);
  FTN_CHECK_LAUNCH();
  return 0;
}

static int launch_conv(ConvArgs& ca, int B, int L, int grid_x, hipStream_t st) {
  const int nco_tot = ca.cout / 16;
  int region_px = 1, max_taps = 1;
  for (int k = 0; k < ca.nbr; ++k) {
    int v = conv_region_px(L, ca.kh[k], ca.kw[k]);
    if (v > region_px) region_px = v;
    if (ca.kh[k] * ca.kw[k] > max_taps) max_taps = ca.kh[k] * ca.kw[k];
    if (ca.kh[k] > 31 || ca.kw[k] > 31) { ftn_set_error("conv kernel %dx%d too large", ca.kh[k], ca.kw[k]); return -1; }
  }
  ca.region_floats = ((region_px * LDS_PX_STRIDE + 16) + 63) & ~63;
  // output-channel tiles per workgroup: as many as keep the weight fragments (taps*NCO KiB) + region in LDS
  int NCO = nco_tot >= 4 ? 4 : (nco_tot >= 2 ? 2 : 1);
  const size_t budget = 96 * 1024;
  while (NCO > 1 && (size_t)ca.region_floats * 4 + (size_t)max_taps * NCO * 1024 > budget) NCO >>= 1;
  const size_t lds = (size_t)ca.region_floats * 4 + (size_t)max_taps * NCO * 1024;
  if (lds > 160 * 1024) { ftn_set_error("conv kernel needs %zu B of LDS (kernel too large)", lds); return -1; }
  ca.nchunk = ftn_cdiv(nco_tot, NCO);
  // heavy branches first
  for (int k = 0; k < ca.nbr; ++k) ca.order[k] = k;
  for (int i = 1; i < ca.nbr; ++i) {
    int v = ca.order[i], jj = i - 1;
    while (jj >= 0 && ca.kh[ca.order[jj]] * ca.kw[ca.order[jj]] < ca.kh[v] * ca.kw[v]) { ca.order[jj + 1] = ca.order[jj]; --jj; }
    ca.order[jj + 1] = v;
  }
  ca.dbg = (g_stamp_which & 1) ? g_stamp_buf : nullptr; ca.dbg_cap = g_stamp_cap;
  dim3 grid(grid_x, B, ca.nbr * ca.nchunk);
  if (NCO == 4) return launch_conv_t<4>(ca, grid, lds, st);
  if (NCO == 2) return launch_conv_t<2>(ca, grid, lds, st);
  return launch_conv_t<1>(ca, grid, lds, st);
}

struct ConvBfGeom { int NCO; size_t lds; int plane_bytes, region_bytes, wbytes, sgroup; bool fast; };

// LDS plan of the split conv engines for window length L (npieces = activation piece planes per region:
// 3 bf16x3, 2 f16x2, 1 plain bf16); NCO = 0 when it does not fit.
static ConvBfGeom conv_bf_geom(int L, int nbr, const int* kh, const int* kw, int cout, int npieces) {
  ConvBfGeom gm = {0, 0, 0, 0, 0, 0, false};
  int region_px = 1, smax = 1;
  // k_conv_bf_fast: mid <= 16 (one input group, one output tile per branch) or, f16x2 only, mid 17..32 (two and two),
  // kernels 3x3 / 5x5 / 7x7
  bool sq357 = cout == 16 || (cout == 32 && npieces == 2);
  for (int k = 0; k < nbr; ++k) {
    if (!(kh[k] == kw[k] && (kh[k] == 3 || kh[k] == 5 || kh[k] == 7))) sq357 = false;
    if (kh[k] > 31 || kw[k] > 31) return gm;
    int v = conv_region_px(L, kh[k], kw[k]);
    if (v > region_px) region_px = v;
    int sl = (kh[k] * kw[k] + 1) / 2;
    if (sl > smax) smax = sl;
  }
  gm.plane_bytes = ((region_px * CBF_PX_BYTES + 1023) & ~1023) + 256;    // whole DMA pieces + the zero block (see CBF_FAST_ZBASE)
  gm.region_bytes = npieces * gm.plane_bytes;
  const int nco_tot = cout / 16;
  if (sq357 && region_px <= FTN_REGION_PX && !g_conv_generic) {
    const int nci = cout / 16, wstr = npieces == 2 ? 2 : 3;       // (the fast path is only taken for cin == cout)
    const size_t need = (size_t)nci * smax * wstr * 1024 + 2 * (size_t)npieces * CBF_FAST_PLANE;
    if (need <= 160 * 1024) {
      gm.fast = true; gm.NCO = 1; gm.sgroup = smax;
      gm.plane_bytes = CBF_FAST_PLANE; gm.region_bytes = npieces * CBF_FAST_PLANE;
      gm.wbytes = nci * smax * wstr * 1024;
      gm.lds = need;
      return gm;
    }
  }
  // Output tiles per workgroup: more tiles share every pixel fragment read.  When all slabs' weights do not fit
  // beside the two region buffers, they are staged in groups of `sgroup` slabs.
  for (int nco = nco_tot >= 4 ? 4 : (nco_tot >= 2 ? 2 : 1); nco >= 1; nco >>= 1) {
    const size_t room = 160 * 1024 - 2 * (size_t)gm.region_bytes;
    int sg = (int)(room / ((size_t)nco * 3 * 1024));
    if (sg > smax) sg = smax;
    if (sg >= smax || (sg >= 8 && nco > 1) || nco == 1) {
      if (sg < 1) return gm;
      const size_t w = (size_t)nco * sg * 3 * 1024;
      gm.NCO = nco; gm.lds = w + 2 * (size_t)gm.region_bytes; gm.wbytes = (int)w; gm.sgroup = sg;
      return gm;
    }
  }
  return gm;
}

template <int NCO, int NS>
static int launch_conv_bf_t(const ConvBfArgs& ca, dim3 grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)k_conv_bf<NCO, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_conv_bf): %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL((k_conv_bf<NCO, NS>), grid, dim3(512), lds, st, ca);
  FTN_CHECK_LAUNCH();
  return 0;
}

template <int NS>
static int launch_conv_bf_n(const ConvBfArgs& ca, const ConvBfGeom& gm, dim3 grid, hipStream_t st) {
  if (gm.NCO == 4) return launch_conv_bf_t<4, NS>(ca, grid, gm.lds, st);
  if (gm.NCO == 2) return launch_conv_bf_t<2, NS>(ca, grid, gm.lds, st);
  return launch_conv_bf_t<1, NS>(ca, grid, gm.lds, st);
}

template <int NS, int NCI>
static int launch_conv_bf_fast_t(const ConvBfArgs& ca, const ConvBfGeom& gm, dim3 grid, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)k_conv_bf_fast<NS, NCI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gm.lds);
  if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_conv_bf_fast): %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL((k_conv_bf_fast<NS, NCI>), grid, dim3(512), gm.lds, st, ca);
  FTN_CHECK_LAUNCH();
  return 0;
}

static int launch_conv_bf_fast(const ConvBfArgs& ca, const ConvBfGeom& gm, dim3 grid, int nsplit, hipStream_t st) {
  if (ca.cin == 32) return launch_conv_bf_fast_t<2, 2>(ca, gm, grid, st);        // mid 32: f16x2 only (conv_bf_geom)
  if (nsplit == 3) return launch_conv_bf_fast_t<3, 1>(ca, gm, grid, st);
  if (nsplit == 2) return launch_conv_bf_fast_t<2, 1>(ca, gm, grid, st);
  return launch_conv_bf_fast_t<1, 1>(ca, gm, grid, st);
}

static const bool g_conv_quant = [] { const char* e = getenv("FTN_CONV_QUANT"); return e == nullptr || e[0] != '0'; }();
static int launch_conv_bf(ConvBfArgs& ca, const ConvBfGeom& gm, int B, int grid_x, int nsplit, hipStream_t st, int rows_est = 0) {
  const int nco_tot = ca.cout / 16;
  ca.nchunk = ftn_cdiv(nco_tot, gm.NCO);
  ca.plane_bytes = gm.plane_bytes;
  ca.region_bytes = gm.region_bytes;
  ca.wbytes = gm.wbytes;
  ca.sgroup = gm.sgroup;
  ca.dbg = ((g_stamp_which & 1) && (!(g_stamp_which & 4) || ca.bt_L > 0)) ? g_stamp_buf : nullptr; ca.dbg_cap = g_stamp_cap;   // which & 4: stage B only
  static const int conv_abl = [] { const char* e = getenv("FTN_CONV_ABL"); return e ? atoi(e) : 0; }();
  ca.abl = conv_abl;
  // batch rows per (persistent) workgroup: as many as still leave ~2 workgroups per CU in the launch - each
  // staging of a tile's weights and pixel bookkeeping is shared by the rows (8 rows: -3 % against 4 at B = 256)
  ca.bpw = 1;
  for (int bp = 8; bp > 1; bp >>= 1)
    if ((long long)grid_x * ftn_cdiv(B, bp) * ca.nbr * ftn_cdiv(nco_tot, gm.NCO) >= 448) { ca.bpw = bp; break; }
  for (int k = 0; k < ca.nbr; ++k) ca.order[k] = k;
  for (int i = 1; i < ca.nbr; ++i) {
    int v = ca.order[i], jj = i - 1;
    while (jj >= 0 && ca.kh[ca.order[jj]] * ca.kw[ca.order[jj]] < ca.kh[v] * ca.kw[v]) { ca.order[jj + 1] = ca.order[jj]; --jj; }
    ca.order[jj + 1] = v;
  }
  dim3 grid(grid_x, ftn_cdiv(B, ca.bpw), ca.nbr * ca.nchunk);
  if (gm.fast && ((ca.cin == 16 && ca.cout == 16) || (ca.cin == 32 && ca.cout == 32 && nsplit == 2))) {
    // virtual branches: (branch, 16-channel output tile); with two input groups a batch row walks two slab loops
    const int nci = ca.cin / 16, nco = ca.cout / 16, nvb = ca.nbr * nco, wstr = nsplit == 2 ? 2 : 3;
    // one workgroup per CU, shared out over the branches in proportion to their cost per batch row
    // (~3.2 k cycles + 0.28 k per K-32 slab, beside a prologue worth ~17 k whatever the kernel size: fitted to
    // tools/stamps.py after the closed-form tap masks, late round 3); every branch gets at least one
    static int ncu = 0;
    if (ncu == 0) {
      int dev = 0; hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      if (ncu < 2 * FTN_MAXBR) ncu = 256;
    }
    double cost[2 * FTN_MAXBR], tot = 0.0;
    size_t lds_fast = 0;
    for (int v = 0; v < nvb; ++v) {
      const int k = v / nco, S = (ca.kh[k] * ca.kw[k] + 1) / 2;
      cost[v] = 3.17 + 0.281 * S * nci; tot += cost[v];
      const size_t need = (size_t)nci * S * wstr * 1024 + 2 * (size_t)gm.region_bytes;
      if (need > lds_fast) lds_fast = need;
    }
    int used = 0, nwg[2 * FTN_MAXBR];
    for (int k = 0; k < nvb; ++k) { nwg[k] = (int)(ncu * cost[k] / tot); if (nwg[k] < 1) nwg[k] = 1; used += nwg[k]; }
    for (int k = 0; used < ncu; k = (k + 1) % nvb) { ++nwg[k]; ++used; }     // leftovers round-robin from the first branch
    for (int k = 0; used > ncu && k < nvb; ++k) while (nwg[k] > 1 && used > ncu) { --nwg[k]; --used; }
    // Rows are whole units: a 7x7 workgroup with 11 rows ends 9 % after one with 10 (tools/stamps.py: the launch
    // ended at 79 us with the median workgroup done at 66).  With an estimate of the row count (the descriptor is on
    // the device: groups bound x tiles of a typical grid x batch rows) pick the split that minimises
    // max_k ceil(rows / nwg_k) * cost_k; a wrong estimate only costs balance, the kernel derives the ranges itself.
    if (rows_est > 0 && g_conv_quant) {
      double bestT = 1e300;
      int best[2 * FTN_MAXBR];
      bool found = false;
      for (int kk = 0; kk < nvb; ++kk) {
        for (int r = 1; r <= rows_est; ++r) {
          const double T = r * cost[kk];                         // (the prologue is the same for every branch: it drops out)
          if (T >= bestT) break;
          int need[2 * FTN_MAXBR], sum = 0;
          bool ok = true;
          for (int k = 0; k < nvb && ok; ++k) {
            const int per = (int)(T / cost[k] + 1e-9);
            if (per < 1) { ok = false; break; }
            need[k] = (rows_est + per - 1) / per;
            sum += need[k];
          }
          if (ok && sum <= ncu) { bestT = T; for (int k = 0; k < nvb; ++k) best[k] = need[k]; found = true; break; }
        }
      }
      if (found) {
        int sum = 0;
        for (int k = 0; k < nvb; ++k) sum += best[k];
        // spare workgroups go where they shorten the longest branch next
        while (sum < ncu) {
          int arg = 0; double worst = -1.0;
          for (int k = 0; k < nvb; ++k) {
            const double t = (double)((rows_est + best[k] - 1) / best[k]) * cost[k];
            if (t > worst) { worst = t; arg = k; }
          }
          ++best[arg]; ++sum;
        }
        for (int k = 0; k < nvb; ++k) nwg[k] = best[k];
      }
    }
    ca.nvb = nvb;
    ca.wg_off[0] = 0;
    for (int k = 0; k < nvb; ++k) ca.wg_off[k + 1] = ca.wg_off[k] + nwg[k];
    const ConvBfGeom gmf = {gm.NCO, lds_fast, gm.plane_bytes, gm.region_bytes, gm.wbytes, gm.sgroup, true};
    return launch_conv_bf_fast(ca, gmf, dim3((unsigned)ca.wg_off[nvb]), nsplit, st);
  }
  if (nsplit == 3) return launch_conv_bf_n<3>(ca, gm, grid, st);
  if (nsplit == 2) return launch_conv_bf_n<2>(ca, gm, grid, st);
  return launch_conv_bf_n<1>(ca, gm, grid, st);
}

template <int ACT, int XIN, int EPI>
static int launch_pw(const PwArgs& pa, bool xvec, int nblk, hipStream_t st) {
  if (xvec) hipLaunchKernelGGL((k_pw<ACT, XIN, true, EPI>), dim3(nblk), dim3(256), 0, st, pa);
  else hipLaunchKernelGGL((k_pw<ACT, XIN, false, EPI>), dim3(nblk), dim3(256), 0, st, pa);
  FTN_CHECK_LAUNCH();
  return 0;
}

template <int ACT, bool XVEC, int NPX, int OTM, bool EXACT, bool PRE>
static int launch_mlp_t(MlpArgs ma, long long Nmax, hipStream_t st) {
  ma.dbg = (g_stamp_which & 2) ? g_stamp_buf : nullptr; ma.dbg_cap = g_stamp_cap;
  const size_t lds = (size_t)ma.cfrag_per_chunk * 1024 * 2;   // double-buffered
  if (lds > 160 * 1024) { ftn_set_error("stage C needs %zu B of LDS for two hidden chunks", lds); return -1; }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_mlp<ACT, XVEC, NPX, OTM, EXACT, PRE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_mlp): %s", hipGetErrorString(e)); return (int)e; }
  }
  const int per = 16 * NPX * 4;
  const int nblk = (int)((Nmax + per - 1) / per);
  hipLaunchKernelGGL((k_mlp<ACT, XVEC, NPX, OTM, EXACT, PRE>), dim3(nblk), dim3(256), lds, st, ma);
  FTN_CHECK_LAUNCH();
  return 0;
}

template <int ACT, bool XVEC>
static int launch_mlp_x(const MlpArgs& ma, long long Nmax, hipStream_t st) {
  const bool pre = ma.nKM <= MLP_PRE_KM && ma.nCP <= MLP_PRE_CP;
  if (ma.n_ot == 7 && pre) return launch_mlp_t<ACT, XVEC, 3, 7, true, true>(ma, Nmax, st);   // d_model 64, mid 16, 3 kernels
  if (ma.n_ot <= 8) {
    if (pre) return launch_mlp_t<ACT, XVEC, 3, 8, false, true>(ma, Nmax, st);
    return launch_mlp_t<ACT, XVEC, 3, 8, false, false>(ma, Nmax, st);
  }
  if (ma.n_ot == 14) return launch_mlp_t<ACT, XVEC, 2, 14, true, false>(ma, Nmax, st);        // d_model 128, mid 32, 3 kernels
  return launch_mlp_t<ACT, XVEC, 2, 16, false, false>(ma, Nmax, st);
}

template <int ACT>
static int launch_mlp(const MlpArgs& ma, bool xvec, long long Nmax, hipStream_t st) {
  return xvec ? launch_mlp_x<ACT, true>(ma, Nmax, st) : launch_mlp_x<ACT, false>(ma, Nmax, st);
}

template <int ACT, bool XVEC, int OTM, bool EXACT, int NS, int NW>
static int launch_mlp_bf_w(MlpBfArgs ma, long long Nmax, hipStream_t st) {
  ma.dbg = (g_stamp_which & 2) ? g_stamp_buf : nullptr; ma.dbg_cap = g_stamp_cap;
  const size_t lds = (size_t)ma.per_chunk * 3 * 1024 * (NW == 8 ? 2 : 1) + (size_t)ma.n_hchunks * 32 * 2 * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)k_mlp_bf<ACT, XVEC, OTM, EXACT, NS, NW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { ftn_set_error("hipFuncSetAttribute(k_mlp_bf): %s", hipGetErrorString(e)); return (int)e; }
  const int px_wg = NW * 32;
  const int nblk = (int)((Nmax + px_wg - 1) / px_wg);
  hipLaunchKernelGGL((k_mlp_bf<ACT, XVEC, OTM, EXACT, NS, NW>), dim3(nblk), dim3(NW * 64), lds, st, ma);
  FTN_CHECK_LAUNCH();
  return 0;
}

template <int ACT, bool XVEC, int OTM, bool EXACT, int NS>
static int launch_mlp_bf_t(const MlpBfArgs& ma, long long Nmax, hipStream_t st) {
  // 4-wave workgroups, two per CU (the 8-wave double-buffered form is kept in the kernel template:
  // 425 us against 360 us at the bench shape)
  return launch_mlp_bf_w<ACT, XVEC, OTM, EXACT, NS, 4>(ma, Nmax, st);
}

template <int ACT, int NS>
static int launch_mlp_bf(const MlpBfArgs& ma, bool xvec, long long Nmax, hipStream_t st) {
  if (ma.n_ot == 7) {
    if (xvec) return launch_mlp_bf_t<ACT, true, 7, true, NS>(ma, Nmax, st);
    return launch_mlp_bf_t<ACT, false, 7, true, NS>(ma, Nmax, st);
  }
  if (xvec) return launch_mlp_bf_t<ACT, true, 8, false, NS>(ma, Nmax, st);
  return launch_mlp_bf_t<ACT, false, 8, false, NS>(ma, Nmax, st);
}

template <int ACT>
static int forward_t(const float* x, float* y, int B, int L, const FtnPlan* pl, const float* wb, const FtnDesc* desc_in,
                     const float* wts, int max_groups, int px_bound, char* ws, hipStream_t st, const float* ln_g,
                     const float* ln_b, float ln_eps, int act_dtype, int flags, int* range_flag) {
  const WsLayout wl = ws_layout(pl, B, L, max_groups, px_bound);
  const int px_row = worst_px_per_row(L, max_groups, px_bound);
  const FtnDesc* desc = (const FtnDesc*)ws;     // sanitised copy, written by the first launch (stage A)
  float* bufA = (float*)(ws + wl.offA);
  float* buf0 = (float*)(ws + wl.off0);
  float* buf1 = (float*)(ws + wl.off1);
  float* bufR = (float*)(ws + wl.off2);
  float* bufG = (float*)(ws + wl.off3);
  const int C = pl->C, CP = pl->CP, FP = pl->FP;
  const bool xvec = (C % 4 == 0) && (((uintptr_t)x & 15) == 0);
  const long long Nmax = (long long)B * px_row;
  int tiles_row;
  worst_tiles(L, max_groups, &tiles_row);
  // (tile, batch row) work items of a conv launch, estimated: every group present, a grid of ~L pixels each
  const int rows_est = (int)((long long)B * max_groups * ((L + FTN_TILE_PX - 1) / FTN_TILE_PX) < (1 << 20) ? B * max_groups * ((L + FTN_TILE_PX - 1) / FTN_TILE_PX) : 0);
  const int nblk_pw = (int)(((long long)B * L + 1 + 16 * NPXU * 4 - 1) / (16 * NPXU * 4));   // stage A: window rows + pad row
  const int nblk_ew = 2048;
  const bool yvec = ((uintptr_t)y & 15) == 0;
  const int nblk_out = (int)(((long long)B * L + 127) / 128);
  int rc;
  prof_mark(0, st);
  if (pl->mode == 0) {
    const int CA = pl->nbr * pl->MP;
    // conv engine: exact fp32 MFMA, or the bf16 matrix pipe (3 pieces = fp32-equivalent, 1 = plain bf16)
    // activation pieces: 3 = bf16x3, 2 = f16x2, 1 = plain bf16
    const int nsplit = pl->engine == 2 ? 1 : (pl->engine == 3 ? 2 : 3);
    ConvBfGeom bfg = {0, 0, 0, 0, 0, 0};
    if (pl->engine != 0) bfg = conv_bf_geom(L, pl->nbr, pl->kh, pl->kw, pl->MP, nsplit);
    const bool use_bf = pl->engine != 0 && bfg.NCO > 0;
    const bool h2 = use_bf && pl->engine == 3;
    // (a kernel set whose weights do not fit the split engines' LDS plan runs on the exact fp32 MFMA kernels)
    // A: a = W_in1 x + b
    PwArgs pa = {};
    pa.x = x; pa.W = wb + pl->w_in1; pa.bias = wb + pl->b_in1; pa.out = bufA; pa.desc = desc;
    pa.B = B; pa.L = L; pa.C = C; pa.KIN = CP; pa.n_ot = CA / 16; pa.OUTC = CA;
    pa.guard_src = desc_in; pa.guard_dst = (FtnDesc*)ws; pa.guard_groups = max_groups; pa.guard_px = px_row;
    if (!h2) range_flag = nullptr;                              // only the f16x2 engine has a range to guard
    pa.range_flag = range_flag;
    if (flags & FTN_FWD_STAGE_A_DONE) { /* ftn_period_finalize_stage_a ran stage A and published the descriptor copy */ }
    else if (use_bf && h2) { if ((rc = launch_pw<ACT, 1, 3>(pa, xvec, nblk_pw, st))) return rc; }
    else if (use_bf) { if ((rc = launch_pw<ACT, 1, 2>(pa, xvec, nblk_pw, st))) return rc; }
    else if ((rc = launch_pw<ACT, 1, 0>(pa, xvec, nblk_pw, st))) return rc;
    prof_mark(1, st);
    ConvBfArgs cb = {};
    // stage C on the bf16 pipe too when the plan carries its fragments and the shapes fit
    const int n_ot_c = CA / 16 + (pl->res2 ? CP / 16 : 0);
    const bool mlp_bf = use_bf && pl->cfragbf_per_chunk > 0 && pl->res1 && pl->res2 && CA > 32 && CA <= 64 && CP > 32 && CP <= 64 &&
                        n_ot_c <= 8 && (size_t)pl->cfragbf_per_chunk * 3 * 1024 * 2 <= 160 * 1024;
    // the default pipeline shape (d_model 128, three kernels, mid 32): k_mlp_bf_c128
    const bool mlp_bf128 = use_bf && !mlp_bf && pl->res1 && pl->res2 && CA == 96 && CP == 128 && n_ot_c == 14 &&
                           pl->cfragbf_per_chunk == 28 &&
                           (size_t)28 * 3 * 1024 + (size_t)pl->n_hchunks * 32 * 2 * sizeof(float) <= 160 * 1024;
    // the u1 stage C of the d_model-64 shape with the FAST k_out behind it and fp32 activations: R keeps its x
    // (OutArgs.r_keeps_x); every other combination subtracts x in stage C as the reference's delta does
    const bool r_keeps_x = mlp_bf && g_mlp_u1 && g_r_keeps_x && act_dtype == 0 && (CA + 31) / 32 == 2 && (CP + 31) / 32 == 2 &&
                           n_ot_c == 7 && CA <= 48 && CP <= 64;
    // stage E on the 16-bit pipe (k_out_h): the second conv then leaves m' as activation pieces
    const bool out_h = (mlp_bf || mlp_bf128) && ftn_out_h_enabled() != 0 && pl->w_out2fb != 0 && nsplit >= 2 && act_dtype == 0 &&
                       ((CA <= 64 && CP <= 64) || (CA <= 96 && CP <= 128));
    // position-major stage C (k_mlp_pos) for the same shape: res1 / res2 once per window position, R group-summed
    const bool mlp_pos64 = mlp_bf && g_mlp_u1 && ftn_mlp_pos_enabled() != 0 && act_dtype == 0 && (CA + 31) / 32 == 2 && (CP + 31) / 32 == 2 &&
                           n_ot_c == 7 && CA == 48 && CP == 64;
    // the d_model-128 shape (f16x2): two groups per pass, one 8-wave workgroup per CU (stagec_pos.hip)
    // (its group-summed R is only understood by k_out_h at this width)
    const bool mlp_pos128 = mlp_bf128 && out_h && g_mlp_u1 && ftn_mlp_pos_enabled() != 0 && act_dtype == 0 && nsplit == 2;
    const bool mlp_pos = mlp_pos64 || mlp_pos128;
    if (use_bf) {
      cb.range_flag = range_flag;
      cb.in = (const __bf16*)bufA; cb.bt_L = L; cb.out = buf1; cb.out_p3 = (mlp_bf || mlp_bf128) ? 1 : 0; cb.bias = wb + (h2 ? pl->b_conv1s : pl->b_conv1); cb.desc = desc;
      cb.B = B; cb.INC = CA; cb.OUTC = CA; cb.nbr = pl->nbr; cb.cin = pl->MP; cb.cout = pl->MP;
      cb.in_stride_br = pl->MP / 16; cb.out_stride_br = pl->MP;
      for (int k = 0; k < pl->nbr; ++k) {
        cb.W[k] = (const __bf16*)(wb + pl->w_convbf1[k]); cb.kh[k] = pl->kh[k]; cb.kw[k] = pl->kw[k];
        cb.inv[k] = h2 ? 1.0f / pl->sc_conv1[k] : 1.0f;
      }
    }
    // B: m = conv(a)
    ConvArgs ca = {};
    ca.in = bufA; ca.bt_L = L; ca.out = buf1; ca.bias = wb + pl->b_conv1; ca.desc = desc; ca.B = B; ca.INC = CA; ca.OUTC = CA;
    ca.nbr = pl->nbr; ca.cin = pl->MP; ca.cout = pl->MP; ca.in_stride_br = pl->MP; ca.out_stride_br = pl->MP;
    for (int k = 0; k < pl->nbr; ++k) { ca.W[k] = wb + pl->w_conv1[k]; ca.kh[k] = pl->kh[k]; ca.kw[k] = pl->kw[k]; }
    if (use_bf) { if ((rc = launch_conv_bf(cb, bfg, B, max_groups, nsplit, st, rows_est))) return rc; }
    else if ((rc = launch_conv(ca, B, L, max_groups, st))) return rc;
    prof_mark(2, st);
    // C: fused pointwise chain
    MlpArgs ma = {};
    ma.x = x; ma.m = buf1; ma.cfrag = wb + pl->w_cfrag; ma.bo = wb + pl->b_out1;
    ma.br = pl->res1 ? wb + pl->b_res1 : nullptr;
    ma.bc = wb + pl->b_c2; ma.outA = buf0; ma.outG = nullptr; ma.outR = bufR; ma.desc = desc;
    ma.B = B; ma.L = L; ma.C = C; ma.CP = CP; ma.FP = FP; ma.KM = CA; ma.AC = CA;
    ma.nKM = CA / 16; ma.nCP = pl->res1 ? CP / 16 : 0;
    ma.n_hchunks = pl->n_hchunks; ma.cfrag_per_chunk = pl->cfrag_per_chunk;
    ma.n_oa = CA / 16; ma.res2_ident = pl->res2 ? 0 : 1; ma.n_ot = ma.n_oa + (pl->res2 ? CP / 16 : 0);
    ma.outA_p3 = use_bf ? (h2 ? 2 : 1) : 0;
    const bool generic_c = !(mlp_bf || mlp_bf128) && stagec_generic(pl);
    if (!generic_c && !(mlp_bf || mlp_bf128) && ma.cfrag_per_chunk != MLP_HT * (ma.nKM + ma.nCP + ma.n_ot)) { ftn_set_error("plan/cfrag layout mismatch"); return -1; }
    if (generic_c) {
      // wide blocks: the chain as pointwise launches with the hidden tensor g in the workspace
      //   g1 = act(W_out1 m + b);  g = act(g1 + res1(x));  a' = W_in2 g + b;  r = res2(g) - x
      const int nblk_px = (int)((Nmax + 16 * NPXU * 4 - 1) / (16 * NPXU * 4));
      PwArgs pg = {};
      pg.x = x; pg.desc = desc; pg.B = B; pg.L = L; pg.C = C;
      pg.in = buf1; pg.W = wb + pl->w_out1; pg.bias = wb + pl->b_out1; pg.out = bufG; pg.KIN = CA; pg.n_ot = FP / 16; pg.OUTC = FP;
      if ((rc = launch_pw<ACT, 0, 5>(pg, xvec, nblk_px, st))) return rc;
      if (pl->res1) {
        pg.in = nullptr; pg.W = wb + pl->w_res1; pg.bias = wb + pl->b_res1; pg.KIN = CP;
        if ((rc = launch_pw<ACT, 2, 6>(pg, xvec, nblk_px, st))) return rc;
      } else {
        if (xvec) hipLaunchKernelGGL((k_ew_ident<ACT, true>), dim3(nblk_ew), dim3(256), 0, st, x, bufG, bufR, desc, B, L, C, FP, 0);
        else hipLaunchKernelGGL((k_ew_ident<ACT, false>), dim3(nblk_ew), dim3(256), 0, st, x, bufG, bufR, desc, B, L, C, FP, 0);
        FTN_CHECK_LAUNCH();
      }
      pg.in = bufG; pg.W = wb + pl->w_in2; pg.bias = wb + pl->b_in2; pg.out = buf0; pg.KIN = FP; pg.n_ot = CA / 16; pg.OUTC = CA;
      if (use_bf && h2) { if ((rc = launch_pw<ACT, 0, 3>(pg, xvec, nblk_px, st))) return rc; }
      else if (use_bf) { if ((rc = launch_pw<ACT, 0, 2>(pg, xvec, nblk_px, st))) return rc; }
      else if ((rc = launch_pw<ACT, 0, 0>(pg, xvec, nblk_px, st))) return rc;
      if (pl->res2) {
        pg.W = wb + pl->w_res2; pg.bias = wb + pl->b_res2; pg.out = bufR; pg.n_ot = CP / 16; pg.OUTC = CP;
        if ((rc = launch_pw<ACT, 0, 4>(pg, xvec, nblk_px, st))) return rc;
      } else {
        if (xvec) hipLaunchKernelGGL((k_ew_ident<ACT, true>), dim3(nblk_ew), dim3(256), 0, st, x, bufG, bufR, desc, B, L, C, FP, 1);
        else hipLaunchKernelGGL((k_ew_ident<ACT, false>), dim3(nblk_ew), dim3(256), 0, st, x, bufG, bufR, desc, B, L, C, FP, 1);
        FTN_CHECK_LAUNCH();
      }
    } else
    if (mlp_bf || mlp_bf128) {
      MlpBfArgs mb = {};
      mb.x = x; mb.m = (const __bf16*)buf1; mb.cfrag = (const __bf16*)(wb + pl->w_cfragbf);
      mb.bo = wb + (h2 ? pl->b_out1s : pl->b_out1); mb.br = wb + (h2 ? pl->b_res1s : pl->b_res1);
      mb.bc = wb + (h2 ? pl->b_c2s : pl->b_c2);
      mb.inv_o = h2 ? 1.0f / pl->sc_out1 : 1.0f; mb.sc_r = h2 ? pl->sc_res1 : 1.0f; mb.inv_r = h2 ? 1.0f / pl->sc_res1 : 1.0f;
      mb.inv_a = h2 ? 1.0f / pl->sc_a2 : 1.0f; mb.inv_r2 = h2 ? 1.0f / pl->sc_r2 : 1.0f;
      mb.r_keeps_x = r_keeps_x ? 1 : 0;
      mb.range_flag = range_flag;
      mb.outA = (__bf16*)buf0; mb.outR = bufR; mb.desc = desc;
      mb.B = B; mb.L = L; mb.C = C; mb.CP = CP; mb.FP = FP; mb.KM = CA; mb.AC = CA;
      mb.nsKM = (CA + 31) / 32; mb.nsCP = (CP + 31) / 32;
      mb.n_oa = CA / 16; mb.n_ot = n_ot_c; mb.n_hchunks = pl->n_hchunks; mb.per_chunk = pl->cfragbf_per_chunk;
      if (mb.per_chunk != 2 * mb.nsKM + 2 * mb.nsCP + mb.n_ot) { ftn_set_error("plan/cfragbf layout mismatch"); return -1; }
      if (mlp_bf128 && !mlp_pos128) {
        if (nsplit == 3) { if ((rc = launch_mlp_bf_c128<ACT, 3>(mb, xvec, Nmax, st))) return rc; }
        else if (nsplit == 2) { if ((rc = launch_mlp_bf_c128<ACT, 2>(mb, xvec, Nmax, st))) return rc; }
        else if ((rc = launch_mlp_bf_c128<ACT, 1>(mb, xvec, Nmax, st))) return rc;
      } else if (mlp_pos) {
        MlpPosArgs mp = {};
        mp.c = mb; mp.c.r_keeps_x = 1; mp.wts = wts; mp.outRs = bufR;
        mp.c.dbg = (g_stamp_which & 2) ? g_stamp_buf : nullptr; mp.c.dbg_cap = g_stamp_cap;
        // units of 16 tail pixels per batch row: sum_g ceil(pad_g / 16) <= (sum_g pad_g + 15 G) / 16, sum_g pad_g <= px_row - L
        const int tail_row = px_row > L ? (px_row - L + 15 * max_groups) / 16 : 0;
        const long long tail_units = (long long)B * tail_row;
        const int tub = tail_units > (1 << 24) ? (1 << 24) : (int)tail_units;
        if (mlp_pos128) { if ((rc = ftn_launch_mlp_pos128(mp, ACT, xvec, tub, st))) return rc; }
        else if ((rc = ftn_launch_mlp_pos64(mp, ACT, nsplit, xvec, tub, st))) return rc;
      } else if (g_mlp_u1 && mb.nsKM == 2 && mb.nsCP == 2 && mb.n_ot == 7) {
        // one 16-pixel unit per wave, four waves per SIMD (see k_mlp_bf_u1)
        if (nsplit == 3) { if ((rc = launch_mlp_bf_u1<ACT, 3, 2, 2, 7>(mb, xvec, Nmax, st))) return rc; }
        else if (nsplit == 2) { if ((rc = launch_mlp_bf_u1<ACT, 2, 2, 2, 7>(mb, xvec, Nmax, st))) return rc; }
        else if ((rc = launch_mlp_bf_u1<ACT, 1, 2, 2, 7>(mb, xvec, Nmax, st))) return rc;
      } else if (nsplit == 3) { if ((rc = launch_mlp_bf<ACT, 3>(mb, xvec, Nmax, st))) return rc; }
      else if (nsplit == 2) { if ((rc = launch_mlp_bf<ACT, 2>(mb, xvec, Nmax, st))) return rc; }
      else if ((rc = launch_mlp_bf<ACT, 1>(mb, xvec, Nmax, st))) return rc;
    } else if ((rc = launch_mlp<ACT>(ma, xvec, Nmax, st))) return rc;
    prof_mark(3, st);
    // D: m' = conv(a')
    ca.in = buf0; ca.bt_L = 0; ca.out = buf1; ca.bias = wb + pl->b_conv2;
    for (int k = 0; k < pl->nbr; ++k) ca.W[k] = wb + pl->w_conv2[k];
    if (use_bf) {
      cb.in = (const __bf16*)buf0; cb.bt_L = 0; cb.bias = wb + (h2 ? pl->b_conv2s : pl->b_conv2); cb.out_p3 = out_h ? 1 : 0;
      for (int k = 0; k < pl->nbr; ++k) { cb.W[k] = (const __bf16*)(wb + pl->w_convbf2[k]); cb.inv[k] = h2 ? 1.0f / pl->sc_conv2[k] : 1.0f; }
      if ((rc = launch_conv_bf(cb, bfg, B, max_groups, nsplit, st, rows_est))) return rc;
    } else if ((rc = launch_conv(ca, B, L, max_groups, st))) return rc;
    prof_mark(4, st);
    // E+F: y = x + sum_g w (act(W_out2 m' + b) + r)
    OutArgs oa = {};
    oa.x = x; oa.y = y; oa.m = buf1; oa.R = bufR; oa.W = wb + pl->w_out2; oa.bias = wb + pl->b_out2; oa.wts = wts;
    oa.desc = desc; oa.B = B; oa.L = L; oa.C = C; oa.CP = CP; oa.KM = CA; oa.act_dtype = act_dtype;
    oa.r_keeps_x = (r_keeps_x || mlp_pos) ? 1 : 0;
    oa.r_summed = mlp_pos ? 1 : 0;
    oa.range_flag = range_flag;
    const bool fast = CA <= 48 && CP <= 64;
    if (fast || out_h) { oa.ln_g = ln_g; oa.ln_b = ln_b; oa.ln_eps = ln_eps; ln_g = nullptr; }   // fused epilogue
    // FAST path: 16 pixels per wave (3 waves/SIMD; with 32 the kernel needs > 256 registers -> 1 wave/SIMD)
    const unsigned nblk_fast = (unsigned)(((long long)B * L + 63) / 64);
    if (out_h) {
      oa.mh = (const __bf16*)buf1; oa.Wf = (const __bf16*)(wb + pl->w_out2fb);
      oa.bias = wb + (h2 ? pl->b_out2s : pl->b_out2); oa.inv_out2 = h2 ? 1.0f / pl->sc_out2 : 1.0f;
      if ((rc = ftn_launch_out_h(oa, ACT, nsplit, xvec && yvec, st))) return rc;
    } else
    if (xvec && yvec && fast) hipLaunchKernelGGL((k_out<ACT, true, false, true, 1>), dim3(nblk_fast), dim3(256), 0, st, oa);
    else if (xvec && yvec) hipLaunchKernelGGL((k_out<ACT, true, false, false>), dim3(nblk_out), dim3(256), 0, st, oa);
    else if (fast) hipLaunchKernelGGL((k_out<ACT, false, false, true, 1>), dim3(nblk_fast), dim3(256), 0, st, oa);
    else hipLaunchKernelGGL((k_out<ACT, false, false, false>), dim3(nblk_out), dim3(256), 0, st, oa);
    FTN_CHECK_LAUNCH();
    prof_mark(5, st);
  } else {
    // A: zero-extended copy of x
    hipLaunchKernelGGL(k_embed, dim3(nblk_ew), dim3(256), 0, st, x, bufA, B, L, C, CP, desc_in, (FtnDesc*)ws, max_groups, px_row);
    FTN_CHECK_LAUNCH();
    prof_mark(1, st);
    // B: m = conv_merged(x) (+ folded proj bias)
    ConvArgs ca = {};
    ca.in = bufA; ca.bt_L = L; ca.out = buf1; ca.bias = wb + pl->b_conv1; ca.desc = desc; ca.B = B; ca.INC = CP; ca.OUTC = FP;
    ca.nbr = 1; ca.cin = CP; ca.cout = FP; ca.in_stride_br = 0; ca.out_stride_br = 0;
    ca.W[0] = wb + pl->w_conv1[0]; ca.kh[0] = pl->kh[0]; ca.kw[0] = pl->kw[0];
    if ((rc = launch_conv(ca, B, L, max_groups, st))) return rc;
    prof_mark(2, st);
    // C: g = act(act(m) + res1(x)) -> G ; r = res2(g) - x
    MlpArgs ma = {};
    ma.x = x; ma.m = buf1; ma.cfrag = wb + pl->w_cfrag; ma.bo = nullptr;
    ma.br = pl->res1 ? wb + pl->b_res1 : nullptr;
    ma.bc = pl->res2 ? wb + pl->b_res2 : nullptr;
    ma.outA = nullptr; ma.outG = bufG; ma.outR = bufR; ma.desc = desc;
    ma.B = B; ma.L = L; ma.C = C; ma.CP = CP; ma.FP = FP; ma.KM = FP; ma.AC = 0;
    ma.nKM = 0; ma.nCP = pl->res1 ? CP / 16 : 0;
    ma.n_hchunks = pl->n_hchunks; ma.cfrag_per_chunk = pl->cfrag_per_chunk;
    ma.n_oa = 0; ma.res2_ident = pl->res2 ? 0 : 1; ma.n_ot = pl->res2 ? CP / 16 : 0;
    if (ma.n_ot > 16) { ftn_set_error("stage C needs %d output tiles (>16): d_model too large for v1", ma.n_ot); return -1; }
    if (ma.cfrag_per_chunk != MLP_HT * (ma.nKM + ma.nCP + ma.n_ot)) { ftn_set_error("plan/cfrag layout mismatch"); return -1; }
    if ((rc = launch_mlp<ACT>(ma, xvec, Nmax, st))) return rc;
    prof_mark(3, st);
    // D: m' = conv_merged'(g)
    ca.in = bufG; ca.bt_L = 0; ca.out = buf0; ca.bias = wb + pl->b_conv2; ca.INC = FP; ca.OUTC = CP; ca.cin = FP; ca.cout = CP;
    ca.W[0] = wb + pl->w_conv2[0];
    if ((rc = launch_conv(ca, B, L, max_groups, st))) return rc;
    prof_mark(4, st);
    // E+F: y = x + sum_g w (act(m') + r)
    OutArgs oa = {};
    oa.x = x; oa.y = y; oa.m = buf0; oa.R = bufR; oa.W = nullptr; oa.bias = nullptr; oa.wts = wts;
    oa.desc = desc; oa.B = B; oa.L = L; oa.C = C; oa.CP = CP; oa.KM = CP; oa.act_dtype = act_dtype;
    if (xvec && yvec) hipLaunchKernelGGL((k_out<ACT, true, true, false>), dim3(nblk_out), dim3(256), 0, st, oa);
    else hipLaunchKernelGGL((k_out<ACT, false, true, false>), dim3(nblk_out), dim3(256), 0, st, oa);
    FTN_CHECK_LAUNCH();
    prof_mark(5, st);
  }
  if (ln_g) {                                                   // not fused above: in-place row pass over y
    const long long rows = (long long)B * L;
    hipLaunchKernelGGL(k_resid_ln, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, y, y, ln_g, ln_b, ln_eps, rows, C);
    FTN_CHECK_LAUNCH();
  }
  prof_mark(6, st);
  if (g_prof.on) {
    if (g_prof.sampling()) ++g_prof.calls;
    ++g_prof.seen;
  }
  return 0;
}

static int forward_checked(const float* x_dev, float* y_dev, int B, int L, const FtnPlan* plan,
                           const float* wblob_dev, const FtnDesc* desc_dev, const float* weights_dev, int max_groups,
                           int px_bound, void* ws_dev, size_t ws_bytes, void* stream, const float* ln_g, const float* ln_b,
                           float ln_eps, int act_dtype, int flags, int* range_flag) {
  FTN_CHECK_ARG(x_dev && y_dev && plan && wblob_dev && desc_dev && weights_dev && ws_dev,
                "ftn_timesblock_forward: null pointer");
  FTN_CHECK_ARG(B >= 1 && B <= 65535 && L >= 2, "ftn_timesblock_forward: bad shape B=%d L=%d", B, L);
  FTN_CHECK_ARG(max_groups >= 1 && max_groups <= FTN_KMAX, "ftn_timesblock_forward: max_groups=%d", max_groups);
  FTN_CHECK_ARG(plan->CP % 16 == 0 && plan->FP % 16 == 0 && plan->CP >= plan->C && plan->FP >= plan->F,
                "ftn_timesblock_forward: plan channel padding is inconsistent");
  FTN_CHECK_ARG(plan->nbr >= 1 && plan->nbr <= FTN_MAXBR, "ftn_timesblock_forward: nbr=%d", plan->nbr);
  FTN_CHECK_ARG(plan->mode == 1 || (plan->MP % 16 == 0 && plan->MP > 0), "ftn_timesblock_forward: bad MP");
  FTN_CHECK_ARG(plan->res1 || plan->CP == plan->FP, "identity res1 needs d_model == d_ff");
  FTN_CHECK_ARG(plan->res2 || plan->CP == plan->FP, "identity res2 needs d_model == d_ff");
  FTN_CHECK_ARG(px_bound >= 0, "ftn_timesblock_forward: px_bound=%d", px_bound);
  FTN_CHECK_ARG((flags & ~FTN_FWD_STAGE_A_DONE) == 0 && !((flags & FTN_FWD_STAGE_A_DONE) && plan->mode != 0),
                "ftn_timesblock_forward: flags=%d", flags);
  FTN_CHECK_ARG(act_dtype >= 0 && act_dtype <= 2 && !(act_dtype != 0 && ln_g != nullptr),
                "ftn_timesblock_forward: act_dtype=%d (the fused LayerNorm epilogue is fp32 only)", act_dtype);
  const size_t need = ftn_timesblock_workspace_bytes(plan, B, L, max_groups, px_bound);
  FTN_CHECK_ARG(need > 0, "ftn_timesblock_forward: B*pixels per row exceeds 2^31 (B=%d L=%d)", B, L);
  FTN_CHECK_ARG(ws_bytes >= need, "ftn_timesblock_forward: workspace %zu < %zu bytes", ws_bytes, need);
  FTN_CHECK_ARG(((uintptr_t)ws_dev & 255) == 0 && ((uintptr_t)wblob_dev & 15) == 0,
                "ftn_timesblock_forward: workspace/weights must be 256/16-byte aligned");
  if (plan->act == 1)
    return forward_t<1>(x_dev, y_dev, B, L, plan, wblob_dev, desc_dev, weights_dev, max_groups, px_bound, (char*)ws_dev,
                        (hipStream_t)stream, ln_g, ln_b, ln_eps, act_dtype, flags, range_flag);
  return forward_t<0>(x_dev, y_dev, B, L, plan, wblob_dev, desc_dev, weights_dev, max_groups, px_bound, (char*)ws_dev,
                      (hipStream_t)stream, ln_g, ln_b, ln_eps, act_dtype, flags, range_flag);
}

// S3-S5 of the selector and stage A of the block in ONE launch (k_finalize_pw): see flowtimes.h
template <int ACT>
static int finalize_stage_a_t(const FinalizeArgs& fa, const PwArgs& pa, int epi, bool xvec, int nblk_pw, size_t lds,
                              int part, hipStream_t st) {
  const dim3 grid(part == 0 ? 1 + nblk_pw : (part == 1 ? nblk_pw : 1)), blk(256);
  if (xvec) {
    if (epi == 3) hipLaunchKernelGGL((k_finalize_pw<ACT, true, 3>), grid, blk, lds, st, fa, pa, part);
    else if (epi == 2) hipLaunchKernelGGL((k_finalize_pw<ACT, true, 2>), grid, blk, lds, st, fa, pa, part);
    else hipLaunchKernelGGL((k_finalize_pw<ACT, true, 0>), grid, blk, lds, st, fa, pa, part);
  } else {
    if (epi == 3) hipLaunchKernelGGL((k_finalize_pw<ACT, false, 3>), grid, blk, lds, st, fa, pa, part);
    else if (epi == 2) hipLaunchKernelGGL((k_finalize_pw<ACT, false, 2>), grid, blk, lds, st, fa, pa, part);
    else hipLaunchKernelGGL((k_finalize_pw<ACT, false, 0>), grid, blk, lds, st, fa, pa, part);
  }
  FTN_CHECK_LAUNCH();
  return 0;
}

extern "C" int ftn_period_finalize_stage_a(const double* psum_dev, int nparts, int Btotal, const float* med_dev, int B,
                                           int L, int k_periods, int pmax, int min_period_threshold, int act_dtype,
                                           int max_unique, double log_base, FtnDesc* desc_dev, float* amps_dev,
                                           float* weights_dev, const float* x_dev, const FtnPlan* plan,
                                           const float* wblob_dev, int max_groups, int px_bound, void* ws_dev,
                                           size_t ws_bytes, void* stream, int* range_flag_dev, const FtnExchange* xch) {
  // psum_dev == NULL: stage A only;  x_dev == NULL: finalize + descriptor copy only (stage A is in the workspace)
  const bool do_fin = psum_dev != nullptr || xch != nullptr, do_a = x_dev != nullptr;
  if (xch != nullptr) {
    FTN_CHECK_ARG(xch->world >= 1 && xch->world <= FTN_XCHG_MAXWORLD && xch->rank >= 0 && xch->rank < xch->world &&
                  xch->seq > 0 && L / 2 + 1 <= xch->F_cap && xch->slots[xch->rank] != nullptr,
                  "ftn_period_finalize_stage_a: bad exchange (world / rank / seq / F_cap)");
    nparts = xch->world;
  }
  FTN_CHECK_ARG(do_fin || do_a, "ftn_period_finalize_stage_a: nothing to do (psum and x both null)");
  FTN_CHECK_ARG(plan && wblob_dev && ws_dev, "ftn_period_finalize_stage_a: null pointer");
  if (do_fin) {
    FTN_CHECK_ARG(med_dev && desc_dev && amps_dev && weights_dev, "ftn_period_finalize_stage_a: null pointer");
    FTN_CHECK_ARG((((uintptr_t)amps_dev | (uintptr_t)weights_dev) & 15) == 0, "ftn_period_finalize_stage_a: amps / weights must be 16-byte aligned");
    FTN_CHECK_ARG(nparts >= 1 && Btotal >= B, "ftn_period_finalize_stage_a: bad shape");
  }
  FTN_CHECK_ARG(B >= 1 && B <= 65535 && L >= 2, "ftn_period_finalize_stage_a: bad shape");
  FTN_CHECK_ARG(k_periods <= FTN_KMAX && act_dtype >= 0 && act_dtype <= 2, "ftn_period_finalize_stage_a: k=%d act_dtype=%d", k_periods, act_dtype);
  FTN_CHECK_ARG(plan->mode == 0 && plan->MP > 0 && plan->MP % 16 == 0, "ftn_period_finalize_stage_a: bottleneck blocks only");
  FTN_CHECK_ARG(max_groups >= 1 && max_groups <= FTN_KMAX && px_bound >= 0, "ftn_period_finalize_stage_a: bounds");
  const size_t need = ftn_timesblock_workspace_bytes(plan, B, L, max_groups, px_bound);
  FTN_CHECK_ARG(need > 0 && ws_bytes >= need && ((uintptr_t)ws_dev & 255) == 0 && ((uintptr_t)wblob_dev & 15) == 0,
                "ftn_period_finalize_stage_a: workspace %zu < %zu bytes or misaligned", ws_bytes, need);
  if (k_periods < 0) k_periods = 0;
  if (pmax < 1) pmax = 1;
  if (min_period_threshold < 1) min_period_threshold = 1;
  if (min_period_threshold > pmax) min_period_threshold = pmax;
  const int F = L / 2 + 1;
  const size_t lds = ftn_finalize_lds_bytes(F);
  FTN_CHECK_ARG(lds <= 48 * 1024, "ftn_period_finalize_stage_a: L=%d too long", L);
  const WsLayout wl = ws_layout(plan, B, L, max_groups, px_bound);
  const int CA = plan->nbr * plan->MP;
  const int nsplit = plan->engine == 2 ? 1 : (plan->engine == 3 ? 2 : 3);
  ConvBfGeom bfg = {0, 0, 0, 0, 0, 0};
  if (plan->engine != 0) bfg = conv_bf_geom(L, plan->nbr, plan->kh, plan->kw, plan->MP, nsplit);
  const bool use_bf = plan->engine != 0 && bfg.NCO > 0;
  const int epi = !use_bf ? 0 : (plan->engine == 3 ? 3 : 2);
  FinalizeArgs fa = {psum_dev, nparts, Btotal, med_dev, B, L, F, k_periods, pmax, min_period_threshold, desc_dev,
                     amps_dev, weights_dev, act_dtype, max_unique > 0 ? max_unique : 0, log_base > 1.0 ? (float)log(log_base) : 0.f};
  if (xch != nullptr) ftn_xch_fill(xch, F, &fa);
  fa.dbg = ((g_stamp_which & 8) && g_stamp_cap >= 8) ? g_stamp_buf : nullptr;
  PwArgs pa = {};
  pa.x = x_dev; pa.W = wblob_dev + plan->w_in1; pa.bias = wblob_dev + plan->b_in1; pa.out = (float*)((char*)ws_dev + wl.offA);
  pa.desc = nullptr; pa.B = B; pa.L = L; pa.C = plan->C; pa.KIN = plan->CP; pa.n_ot = CA / 16; pa.OUTC = CA;
  pa.guard_src = desc_dev; pa.guard_dst = (FtnDesc*)ws_dev; pa.guard_groups = max_groups;
  pa.guard_px = worst_px_per_row(L, max_groups, px_bound);
  pa.range_flag = epi == 3 ? range_flag_dev : nullptr;
  const bool xvec = (plan->C % 4 == 0) && (((uintptr_t)x_dev & 15) == 0);
  const int nblk_pw = (int)(((long long)B * L + 1 + 16 * NPXU * 4 - 1) / (16 * NPXU * 4));
  const int part = do_fin && do_a ? 0 : (do_a ? 1 : 2);
  if (plan->act == 1) return finalize_stage_a_t<1>(fa, pa, epi, xvec, nblk_pw, lds, part, (hipStream_t)stream);
  return finalize_stage_a_t<0>(fa, pa, epi, xvec, nblk_pw, lds, part, (hipStream_t)stream);
}

extern "C" int ftn_timesblock_forward(const float* x_dev, float* y_dev, int B, int L, const FtnPlan* plan,
                                      const float* wblob_dev, const FtnDesc* desc_dev, const float* weights_dev,
                                      int max_groups, int px_bound, int act_dtype, int flags, void* ws_dev,
                                      size_t ws_bytes, void* stream, int* range_flag_dev) {
  return forward_checked(x_dev, y_dev, B, L, plan, wblob_dev, desc_dev, weights_dev, max_groups, px_bound, ws_dev, ws_bytes,
                         stream, nullptr, nullptr, 0.f, act_dtype, flags, range_flag_dev);
}

extern "C" int ftn_timesblock_forward_norm(const float* x_dev, float* y_dev, int B, int L, const FtnPlan* plan,
                                           const float* wblob_dev, const FtnDesc* desc_dev, const float* weights_dev,
                                           int max_groups, int px_bound, int flags, const float* ln_gamma_dev,
                                           const float* ln_beta_dev, float ln_eps, void* ws_dev, size_t ws_bytes,
                                           void* stream, int* range_flag_dev) {
  FTN_CHECK_ARG(ln_gamma_dev && ln_beta_dev && ln_eps >= 0.f, "ftn_timesblock_forward_norm: LayerNorm parameters");
  return forward_checked(x_dev, y_dev, B, L, plan, wblob_dev, desc_dev, weights_dev, max_groups, px_bound, ws_dev,
                         ws_bytes, stream, ln_gamma_dev, ln_beta_dev, ln_eps, 0, flags, range_flag_dev);
}

extern "C" int ftn_residual_layernorm(const float* x_dev, const float* new_dev, float* out_dev, long long rows, int C,
                                      const float* ln_gamma_dev, const float* ln_beta_dev, float ln_eps,
                                      void* stream) {
  FTN_CHECK_ARG(x_dev && new_dev && out_dev && ln_gamma_dev && ln_beta_dev, "ftn_residual_layernorm: null pointer");
  FTN_CHECK_ARG(rows >= 1 && rows <= 0x7fffffffLL * 4 && C >= 1, "ftn_residual_layernorm: rows=%lld C=%d", rows, C);
  hipLaunchKernelGGL(k_resid_ln, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x_dev, new_dev,
                     out_dev, ln_gamma_dev, ln_beta_dev, ln_eps, rows, C);
  FTN_CHECK_LAUNCH();
  return 0;
}

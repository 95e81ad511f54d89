// Host-side folding and packing of TimesBlock.inception weights for the HIP kernels (no device code).
//
// Input: the reference state_dict tensors of nn.Sequential(InceptionBlock, act, InceptionBlock) (reference
// models/timesnet.py:744-762; InceptionBranch :575-590, InceptionBlock :622-637) as raw fp32 host pointers.
// Output: the weight blob + FtnPlan every ftn_timesblock_* call takes (include/flowtimes.h).
//
// Algebra (exact, SURVEY finding 5): every InceptionBranch is linear, so proj(cat_k branch_k(u)) =
// sum_k P_k branch_k(u) + b_proj with P_k the k-th column block of proj.weight:
//  * bottleneck branch (1x1 -> kxk -> 1x1): the last 1x1 folds into P_k: W_out[:, k] = P_k W3_k,
//    b_out = b_proj + sum_k P_k b3_k (the first 1x1 cannot be folded through the kxk conv: its bias is absent in
//    the zero-padded halo);
//  * single-conv branch (ratio 1): P_k folds into the conv and the per-kernel convs merge into one conv of the
//    largest kernel size (smaller kernels centred, which preserves 'same' zero padding).
// Folding is done in fp64, results stored as fp32.  Every array of the blob starts on a 64-float boundary.
#include <math.h>
#include <string.h>
#include <vector>
#include "ftn_common.h"

namespace {

typedef std::vector<double> dvec;

struct Blob {
  float* out;          // null: only count
  size_t cap, n;
  bool ok;
  Blob(float* o, size_t c) : out(o), cap(c), n(0), ok(true) {}
  template <typename T>
  int64_t add(const T* src, size_t count) {
    const size_t off = n, pad = (64 - count % 64) % 64;
    if (out) {
      if (n + count + pad > cap) { ok = false; n += count + pad; return (int64_t)off; }
      for (size_t i = 0; i < count; ++i) out[n + i] = (float)src[i];
      for (size_t i = 0; i < pad; ++i) out[n + count + i] = 0.f;
    }
    n += count + pad;
    return (int64_t)off;
  }
  int64_t add(const dvec& v) { return add(v.data(), v.size()); }
  // 16-bit patterns packed two per float word
  int64_t add16(const std::vector<uint16_t>& v) {
    const size_t words = (v.size() + 1) / 2;
    const size_t off = n, pad = (64 - words % 64) % 64;
    if (out) {
      if (n + words + pad > cap) { ok = false; n += words + pad; return (int64_t)off; }
      uint16_t* dst = (uint16_t*)(out + n);
      for (size_t i = 0; i < v.size(); ++i) dst[i] = v[i];
      if (v.size() & 1) dst[v.size()] = 0;
      for (size_t i = 0; i < pad; ++i) out[n + words + i] = 0.f;
    }
    n += words + pad;
    return (int64_t)off;
  }
};

inline int pad16(int v) { return (v + 15) / 16 * 16; }

inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float bf16_rne(float x) {             // fp32 -> bf16 (round to nearest even), as fp32 with 16 low bits clear
  const uint64_t u = f32_bits(x);
  const uint64_t r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16;
  return bits_f32((uint32_t)r);
}
inline uint16_t bf16_pattern(float v) { return (uint16_t)(f32_bits(v) >> 16); }
inline uint16_t f16_pattern(float v) { const _Float16 h = (_Float16)v; uint16_t u; memcpy(&u, &h, 2); return u; }
inline float f16_round(float v) { return (float)(_Float16)v; }

// three bf16 pieces of an fp32 value (csrc/ftn_common.h store_p3)
inline void split3(float x, float (&p)[3]) {
  p[0] = bf16_rne(x);
  const float r1 = x - p[0];
  p[1] = bf16_rne(r1);
  p[2] = bf16_rne(r1 - p[1]);
}
// three fp16 pieces of an already prescaled weight (csrc/ftn_common.h, f16x2): A1, A2 = A1 2^-11, A3 = W~ - A1
inline void split_h2w(float ws, float (&p)[3]) {
  p[0] = f16_round(ws);
  p[1] = f16_round(p[0] * 4.8828125e-4f);
  p[2] = f16_round(ws - p[0]);
}
// power of two sc with max |sc W| in [2^10, 2^11)
inline double pow2_scale(const double* w, size_t n) {
  double m = 0.0;
  for (size_t i = 0; i < n; ++i) m = fabs(w[i]) > m ? fabs(w[i]) : m;
  if (!(m > 0.0) || !isfinite(m)) return 1.0;
  return ldexp(1.0, 10 - (int)floor(log2(m)));
}

inline int bottleneck_mid(int cin, int cout, double ratio) {   // 0: single-conv branch (:575), else :582-585
  if (fabs(ratio - 1.0) <= 1e-9) return 0;
  const int lo = cin < cout ? cin : cout;
  int mid = (int)ceil((double)lo / ratio);
  return mid < 1 ? 1 : mid;
}

// w[cout][cin][kh][kw] -> fp32 MFMA fragments [tap][cc][co][q][j][e] = W[16co + j][16cc + 4q + e][dy][dx]
dvec pack_conv(const dvec& w, int cout, int cin, int kh, int kw, int cinP, int coutP) {
  dvec out((size_t)kh * kw * cinP * coutP, 0.0);
  const int ncc = cinP / 16, nco = coutP / 16;
  for (int dy = 0; dy < kh; ++dy)
    for (int dx = 0; dx < kw; ++dx)
      for (int cc = 0; cc < ncc; ++cc)
        for (int co = 0; co < nco; ++co)
          for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 16; ++j)
              for (int e = 0; e < 4; ++e) {
                const int o = 16 * co + j, i = 16 * cc + 4 * q + e;
                const double v = (o < cout && i < cin) ? w[(((size_t)o * cin + i) * kh + dy) * kw + dx] : 0.0;
                out[((((((size_t)(dy * kw + dx) * ncc + cc) * nco + co) * 4 + q) * 16 + j) * 4) + e] = v;
              }
  return out;
}

// K=32 fragments of the split engines: [cc][co][slab][piece][qa][i][e]: lane (i, qa) holds
// W[16co + i][16cc + 8(qa & 1) + e][tap 2*slab + (qa >> 1)]; h2: fp16 pieces of sc*w, else bf16 pieces
std::vector<uint16_t> pack_conv_split(const dvec& w, int cout, int cin, int kh, int kw, int cinP, int coutP, bool h2,
                                      double sc) {
  const int nt = kh * kw, S = (nt + 1) / 2, ncc = cinP / 16, nco = coutP / 16;
  std::vector<uint16_t> out((size_t)ncc * nco * S * 3 * 512, 0);
  for (int cc = 0; cc < ncc; ++cc)
    for (int co = 0; co < nco; ++co)
      for (int s = 0; s < S; ++s)
        for (int qa = 0; qa < 4; ++qa)
          for (int i = 0; i < 16; ++i)
            for (int e = 0; e < 8; ++e) {
              const int o = 16 * co + i, ci = 16 * cc + 8 * (qa & 1) + e, tap = 2 * s + (qa >> 1);
              float v = 0.f;
              if (o < cout && ci < cin && tap < nt) v = (float)(w[((size_t)o * cin + ci) * nt + tap] * sc);
              float pc[3];
              if (h2) split_h2w(v, pc); else split3(v, pc);
              for (int pz = 0; pz < 3; ++pz) {
                const size_t idx = (((((size_t)(cc * nco + co) * S + s) * 3 + pz) * 4 + qa) * 16 + i) * 8 + e;
                out[idx] = h2 ? f16_pattern(pc[pz]) : bf16_pattern(pc[pz]);
              }
            }
  return out;
}

struct Mat {        // row-major [rows][cols] fp64
  int rows, cols;
  dvec v;
  Mat() : rows(0), cols(0) {}
  Mat(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.0) {}
  double& at(int r, int c) { return v[(size_t)r * cols + c]; }
  double get(int r, int c) const { return (r < rows && c < cols) ? v[(size_t)r * cols + c] : 0.0; }
};

// 16x16 block (rows 16R.., cols 16S..) as a lane-linear fp32 MFMA A fragment [lane = 16q + j][e] = W[16R+j][16S+4q+e]
void frag32(const Mat* W, int R, int Sx, double* dst) {
  for (int q = 0; q < 4; ++q)
    for (int j = 0; j < 16; ++j)
      for (int e = 0; e < 4; ++e) dst[(q * 16 + j) * 4 + e] = W ? W->get(16 * R + j, 16 * Sx + 4 * q + e) : 0.0;
}

// K=32 fragment of the split engines, three pieces: [piece][qa][i][e] = W[16R + i][cols[8qa + e]] (cols < 0: zero)
void frag_split(const Mat* W, int R, const int* cols, bool h2, uint16_t* dst) {
  for (int qa = 0; qa < 4; ++qa)
    for (int i = 0; i < 16; ++i)
      for (int e = 0; e < 8; ++e) {
        const int c = cols[8 * qa + e];
        const float v = (W && c >= 0) ? (float)W->get(16 * R + i, c) : 0.f;
        float pc[3];
        if (h2) split_h2w(v, pc); else split3(v, pc);
        for (int pz = 0; pz < 3; ++pz) dst[((size_t)(pz * 4 + qa) * 16 + i) * 8 + e] = h2 ? f16_pattern(pc[pz]) : bf16_pattern(pc[pz]);
      }
}

#define CHUNK_TILES 2     // hidden row tiles per fp32 stage-C chunk (MLP_HT in inception.hip)

struct BlockFold {        // one InceptionBlock, bottleneck mode
  Mat W_in, W_out;        // [CA][cinP], [coutP][CA]
  dvec b_in, b_conv, b_out;
  std::vector<dvec> conv32;                 // per branch fp32 fragments
  std::vector<std::vector<uint16_t>> convs; // per branch split-engine fragments
  std::vector<double> conv_sc;
};

bool need(const void* p, const char* what) {
  if (p == nullptr) { ftn_set_error("ftn_inception_pack_weights: %s is null", what); return false; }
  return true;
}

}  // namespace

static int pack_impl(const FtnInceptionBlockWeights* blk0, const FtnInceptionBlockWeights* blk2, int d_model, int d_ff,
                     int nk, const int* khs, const int* kws, double ratio, int act, int engine, Blob& blob,
                     FtnPlan* plan) {
  FTN_CHECK_ARG(d_model >= 1 && d_ff >= 1 && nk >= 1 && nk <= FTN_MAXBR && khs && kws && ratio > 0.0,
                "ftn_inception_pack_weights: bad shape (d_model=%d d_ff=%d kernels=%d)", d_model, d_ff, nk);
  FTN_CHECK_ARG(engine >= 0 && engine <= 3 && (act == 0 || act == 1), "ftn_inception_pack_weights: engine=%d act=%d", engine, act);
  for (int j = 0; j < nk; ++j)
    FTN_CHECK_ARG(khs[j] >= 1 && kws[j] >= 1 && (khs[j] & 1) && (kws[j] & 1),
                  "kernel sizes must be odd and positive for 'same' padding, got (%d, %d)", khs[j], kws[j]);
  const bool dry = blk0 == nullptr;          // size query: no inputs are read
  const int C = d_model, F = d_ff, CP = pad16(C), FP = pad16(F);
  memset(plan, 0, sizeof(*plan));
  plan->engine = engine; plan->C = C; plan->CP = CP; plan->F = F; plan->FP = FP; plan->act = act;
  const int mid = bottleneck_mid(C, F, ratio);
  const FtnInceptionBlockWeights* B[2] = {blk0, blk2};
  const int cins[2] = {C, F}, couts[2] = {F, C}, cinPs[2] = {CP, FP}, coutPs[2] = {FP, CP};
  for (int j = 0; j < FTN_MAXBR; ++j) plan->sc_conv1[j] = plan->sc_conv2[j] = 1.0f;
  plan->sc_out1 = plan->sc_res1 = plan->sc_a2 = plan->sc_r2 = plan->sc_out2 = 1.0f;

  auto res = [&](int bi, int32_t* flag, int64_t* w_off, int64_t* b_off, Mat* Wm, dvec* bv) -> int {
    const int cin = cins[bi], cout = couts[bi], cinP = cinPs[bi], coutP = coutPs[bi];
    const bool has = dry ? cin != cout : B[bi]->res_w != nullptr;
    if (!has) {
      FTN_CHECK_ARG(cin == cout, "res_proj of block %d missing although in_ch != out_ch", 2 * bi);
      *flag = 0; *w_off = 0; *b_off = 0;
      return 0;
    }
    Mat W(coutP, cinP);
    dvec b(coutP, 0.0);
    if (!dry) {
      if (!need(B[bi]->res_b, "res_proj.bias")) return -1;
      for (int o = 0; o < cout; ++o) {
        for (int i = 0; i < cin; ++i) W.at(o, i) = B[bi]->res_w[(size_t)o * cin + i];
        b[o] = B[bi]->res_b[o];
      }
    }
    *flag = 1; *w_off = blob.add(W.v); *b_off = blob.add(b);
    if (Wm) *Wm = W;
    if (bv) *bv = b;
    return 0;
  };

  if (mid > 0) {
    plan->mode = 0;
    const int MP = pad16(mid), CA = nk * MP;
    plan->MP = MP; plan->nbr = nk;
    for (int j = 0; j < nk; ++j) { plan->kh[j] = khs[j]; plan->kw[j] = kws[j]; }
    const bool h2 = engine == 3;
    BlockFold fb[2];
    for (int bi = 0; bi < 2; ++bi) {
      const int cin = cins[bi], cout = couts[bi], cinP = cinPs[bi], coutP = coutPs[bi];
      BlockFold& f = fb[bi];
      f.W_in = Mat(CA, cinP); f.W_out = Mat(coutP, CA);
      f.b_in.assign(CA, 0.0); f.b_conv.assign(CA, 0.0); f.b_out.assign(coutP, 0.0);
      if (!dry) {
        if (!need(B[bi]->proj_w, "proj.weight") || !need(B[bi]->proj_b, "proj.bias")) return -1;
        for (int o = 0; o < cout; ++o) f.b_out[o] = B[bi]->proj_b[o];
      }
      for (int j = 0; j < nk; ++j) {
        const int kh = khs[j], kw = kws[j];
        dvec w2((size_t)mid * mid * kh * kw, 0.0);
        if (!dry) {
          const float* w1 = B[bi]->branch_w[j][0]; const float* b1 = B[bi]->branch_b[j][0];
          const float* w2p = B[bi]->branch_w[j][1]; const float* b2 = B[bi]->branch_b[j][1];
          const float* w3 = B[bi]->branch_w[j][2]; const float* b3 = B[bi]->branch_b[j][2];
          if (!need(w1, "branch.0.weight") || !need(b1, "branch.0.bias") || !need(w2p, "branch.1.weight") ||
              !need(b2, "branch.1.bias") || !need(w3, "branch.2.weight") || !need(b3, "branch.2.bias")) return -1;
          for (int m = 0; m < mid; ++m) {
            for (int i = 0; i < cin; ++i) f.W_in.at(j * MP + m, i) = w1[(size_t)m * cin + i];
            f.b_in[j * MP + m] = b1[m];
            f.b_conv[j * MP + m] = b2[m];
          }
          for (size_t t = 0; t < w2.size(); ++t) w2[t] = w2p[t];
          const float* proj = B[bi]->proj_w;                      // [cout][nk*cout]
          for (int o = 0; o < cout; ++o) {
            const float* Pk = proj + (size_t)o * nk * cout + (size_t)j * cout;
            for (int m = 0; m < mid; ++m) {
              double s = 0.0;
              for (int t = 0; t < cout; ++t) s += (double)Pk[t] * (double)w3[(size_t)t * mid + m];
              f.W_out.at(o, j * MP + m) = s;
            }
            double sb = 0.0;
            for (int t = 0; t < cout; ++t) sb += (double)Pk[t] * (double)b3[t];
            f.b_out[o] += sb;
          }
        }
        f.conv32.push_back(pack_conv(w2, mid, mid, kh, kw, MP, MP));
        const double sc = h2 ? pow2_scale(w2.data(), w2.size()) : 1.0;
        f.conv_sc.push_back(sc);
        f.convs.push_back(pack_conv_split(w2, mid, mid, kh, kw, MP, MP, h2, sc));
      }
    }
    plan->w_in1 = blob.add(fb[0].W_in.v); plan->b_in1 = blob.add(fb[0].b_in);
    for (int j = 0; j < nk; ++j) plan->w_conv1[j] = blob.add(fb[0].conv32[j]);
    plan->b_conv1 = blob.add(fb[0].b_conv);
    plan->w_out1 = blob.add(fb[0].W_out.v); plan->b_out1 = blob.add(fb[0].b_out);
    Mat Wr1, Wr2; dvec br1, br2;
    if (res(0, &plan->res1, &plan->w_res1, &plan->b_res1, &Wr1, &br1)) return -1;
    plan->w_in2 = blob.add(fb[1].W_in.v); plan->b_in2 = blob.add(fb[1].b_in);
    for (int j = 0; j < nk; ++j) plan->w_conv2[j] = blob.add(fb[1].conv32[j]);
    plan->b_conv2 = blob.add(fb[1].b_conv);
    plan->w_out2 = blob.add(fb[1].W_out.v); plan->b_out2 = blob.add(fb[1].b_out);
    if (res(1, &plan->res2, &plan->w_res2, &plan->b_res2, &Wr2, &br2)) return -1;
    // stacked stage-C projection [W_in2 ; W_res2]
    Mat Wc(CA + (plan->res2 ? CP : 0), FP);
    dvec bc(Wc.rows, 0.0);
    for (int r = 0; r < CA; ++r) { for (int c = 0; c < FP; ++c) Wc.at(r, c) = fb[1].W_in.get(r, c); bc[r] = fb[1].b_in[r]; }
    if (plan->res2)
      for (int r = 0; r < CP; ++r) { for (int c = 0; c < FP; ++c) Wc.at(CA + r, c) = Wr2.get(r, c); bc[CA + r] = br2[r]; }
    plan->w_c2 = blob.add(Wc.v); plan->b_c2 = blob.add(bc);
    const int nKM = CA / 16, nCP = plan->res1 ? CP / 16 : 0, n_ot = Wc.rows / 16;
    const int per = CHUNK_TILES * (nKM + nCP + n_ot);
    plan->n_hchunks = (FP + 16 * CHUNK_TILES - 1) / (16 * CHUNK_TILES);
    if (n_ot > 16 || (size_t)per * 1024 * 2 > 160 * 1024) {
      plan->w_cfrag = 0; plan->cfrag_per_chunk = 0;             // generic stage C (inception.hip stagec_generic)
    } else {
      dvec cf((size_t)plan->n_hchunks * per * 256, 0.0);
      for (int hc = 0; hc < plan->n_hchunks; ++hc) {
        int k = 0;
        double* base = cf.data() + (size_t)hc * per * 256;
        for (int t = 0; t < CHUNK_TILES; ++t) for (int s = 0; s < nKM; ++s) frag32(&fb[0].W_out, hc * CHUNK_TILES + t, s, base + (size_t)(k++) * 256);
        for (int t = 0; t < CHUNK_TILES; ++t) for (int s = 0; s < nCP; ++s) frag32(&Wr1, hc * CHUNK_TILES + t, s, base + (size_t)(k++) * 256);
        for (int t = 0; t < CHUNK_TILES; ++t) for (int o = 0; o < n_ot; ++o) frag32(&Wc, o, hc * CHUNK_TILES + t, base + (size_t)(k++) * 256);
      }
      plan->w_cfrag = blob.add(cf); plan->cfrag_per_chunk = per;
    }
    for (int j = 0; j < nk; ++j) {
      plan->w_convbf1[j] = blob.add16(fb[0].convs[j]);
      plan->w_convbf2[j] = blob.add16(fb[1].convs[j]);
    }
    if (h2) {
      dvec b1s(CA), b2s(CA);
      for (int j = 0; j < nk; ++j) {
        plan->sc_conv1[j] = (float)fb[0].conv_sc[j]; plan->sc_conv2[j] = (float)fb[1].conv_sc[j];
        for (int m = 0; m < MP; ++m) { b1s[j * MP + m] = fb[0].b_conv[j * MP + m] * fb[0].conv_sc[j]; b2s[j * MP + m] = fb[1].b_conv[j * MP + m] * fb[1].conv_sc[j]; }
      }
      plan->b_conv1s = blob.add(b1s); plan->b_conv2s = blob.add(b2s);
    }
    const bool tuned = (CA > 32 && CA <= 64 && CP > 32 && CP <= 64 && n_ot <= 8) || (CA == 96 && CP == 128 && n_ot == 14);
    if (plan->res1 && plan->res2 && tuned) {      // shapes the split-engine stage-C kernels exist for
      const int nsKM = (CA + 31) / 32, nsCP = (CP + 31) / 32;
      const int perb = 2 * nsKM + 2 * nsCP + n_ot;
      Mat Wo = fb[0].W_out, Wr = Wr1, Wcs = Wc;
      if (h2) {
        const double so = pow2_scale(Wo.v.data(), Wo.v.size()), sr = pow2_scale(Wr.v.data(), Wr.v.size());
        const double sa = pow2_scale(fb[1].W_in.v.data(), fb[1].W_in.v.size()), s2 = pow2_scale(Wr2.v.data(), Wr2.v.size());
        plan->sc_out1 = (float)so; plan->sc_res1 = (float)sr; plan->sc_a2 = (float)sa; plan->sc_r2 = (float)s2;
        for (double& v : Wo.v) v *= so;
        for (double& v : Wr.v) v *= sr;
        for (int r = 0; r < Wcs.rows; ++r) for (int c = 0; c < FP; ++c) Wcs.at(r, c) *= (r < CA ? sa : s2);
        dvec bo(FP), brs(FP), bcs(Wc.rows);
        for (int i = 0; i < FP; ++i) { bo[i] = fb[0].b_out[i] * so; brs[i] = br1[i] * sr; }
        for (int r = 0; r < Wc.rows; ++r) bcs[r] = bc[r] * (r < CA ? sa : s2);
        plan->b_out1s = blob.add(bo); plan->b_res1s = blob.add(brs); plan->b_c2s = blob.add(bcs);
      }
      std::vector<uint16_t> cfb((size_t)plan->n_hchunks * perb * 3 * 512, 0);
      for (int hc = 0; hc < plan->n_hchunks; ++hc) {
        int k = 0;
        uint16_t* base = cfb.data() + (size_t)hc * perb * 3 * 512;
        int cols[32];
        for (int t = 0; t < 2; ++t) for (int s = 0; s < nsKM; ++s) { for (int c = 0; c < 32; ++c) cols[c] = 32 * s + c; frag_split(&Wo, hc * 2 + t, cols, h2, base + (size_t)(k++) * 3 * 512); }
        for (int t = 0; t < 2; ++t) for (int s = 0; s < nsCP; ++s) { for (int c = 0; c < 32; ++c) cols[c] = 32 * s + c; frag_split(&Wr, hc * 2 + t, cols, h2, base + (size_t)(k++) * 3 * 512); }
        // layer-2 columns follow the accumulator order of the hidden tile pair: k = 8qa + e ->
        // hidden channel 32hc + (e < 4 ? 4qa + e : 16 + 4qa + e - 4)
        for (int qa = 0; qa < 4; ++qa) for (int e = 0; e < 8; ++e) cols[8 * qa + e] = 32 * hc + (e < 4 ? 4 * qa + e : 16 + 4 * qa + e - 4);
        for (int o = 0; o < n_ot; ++o) frag_split(&Wcs, o, cols, h2, base + (size_t)(k++) * 3 * 512);
      }
      plan->w_cfragbf = blob.add16(cfb); plan->cfragbf_per_chunk = perb;
      // stage E (k_out_h): w_out2 [CP][CA] as K=32 fragments, row tile major
      Mat Wo2 = fb[1].W_out;
      if (h2) {
        const double s5 = pow2_scale(Wo2.v.data(), Wo2.v.size());
        plan->sc_out2 = (float)s5;
        for (double& v : Wo2.v) v *= s5;
        dvec b2s(CP);
        for (int i = 0; i < CP; ++i) b2s[i] = fb[1].b_out[i] * s5;
        plan->b_out2s = blob.add(b2s);
      }
      std::vector<uint16_t> ofb((size_t)(CP / 16) * nsKM * 3 * 512, 0);
      {
        int k = 0, cols[32];
        for (int o = 0; o < CP / 16; ++o)
          for (int s = 0; s < nsKM; ++s) { for (int c = 0; c < 32; ++c) cols[c] = 32 * s + c; frag_split(&Wo2, o, cols, h2, ofb.data() + (size_t)(k++) * 3 * 512); }
      }
      plan->w_out2fb = blob.add16(ofb);
    }
  } else {
    plan->mode = 1; plan->MP = 0; plan->nbr = 1;
    int KH = 1, KW = 1;
    for (int j = 0; j < nk; ++j) { KH = khs[j] > KH ? khs[j] : KH; KW = kws[j] > KW ? kws[j] : KW; }
    plan->kh[0] = KH; plan->kw[0] = KW;
    dvec cpk[2], bpk[2];
    for (int bi = 0; bi < 2; ++bi) {
      const int cin = cins[bi], cout = couts[bi], cinP = cinPs[bi], coutP = coutPs[bi];
      dvec Wm((size_t)cout * cin * KH * KW, 0.0), bm(coutP, 0.0);
      if (!dry) {
        if (!need(B[bi]->proj_w, "proj.weight") || !need(B[bi]->proj_b, "proj.bias")) return -1;
        for (int o = 0; o < cout; ++o) bm[o] = B[bi]->proj_b[o];
        for (int j = 0; j < nk; ++j) {
          const int kh = khs[j], kw = kws[j], oy = (KH - kh) / 2, ox = (KW - kw) / 2;
          const float* w = B[bi]->branch_w[j][0]; const float* b = B[bi]->branch_b[j][0];
          if (!need(w, "branch.0.weight") || !need(b, "branch.0.bias")) return -1;
          for (int o = 0; o < cout; ++o) {
            const float* Pk = B[bi]->proj_w + (size_t)o * nk * cout + (size_t)j * cout;
            for (int p = 0; p < cout; ++p) {
              const double pv = Pk[p];
              if (pv == 0.0) continue;
              for (int i = 0; i < cin; ++i)
                for (int dy = 0; dy < kh; ++dy)
                  for (int dx = 0; dx < kw; ++dx)
                    Wm[(((size_t)o * cin + i) * KH + oy + dy) * KW + ox + dx] += pv * (double)w[(((size_t)p * cin + i) * kh + dy) * kw + dx];
              bm[o] += pv * (double)b[p];
            }
          }
        }
      }
      cpk[bi] = pack_conv(Wm, cout, cin, KH, KW, cinP, coutP);
      bpk[bi] = bm;
    }
    plan->w_conv1[0] = blob.add(cpk[0]); plan->b_conv1 = blob.add(bpk[0]);
    Mat Wr1, Wr2;
    if (res(0, &plan->res1, &plan->w_res1, &plan->b_res1, &Wr1, nullptr)) return -1;
    plan->w_conv2[0] = blob.add(cpk[1]); plan->b_conv2 = blob.add(bpk[1]);
    if (res(1, &plan->res2, &plan->w_res2, &plan->b_res2, &Wr2, nullptr)) return -1;
    const int nCP = plan->res1 ? CP / 16 : 0, n_ot = plan->res2 ? CP / 16 : 0;
    const int per = CHUNK_TILES * (nCP + n_ot);
    plan->n_hchunks = (FP + 16 * CHUNK_TILES - 1) / (16 * CHUNK_TILES);
    dvec cf((size_t)plan->n_hchunks * (per > 0 ? per : 1) * 256, 0.0);
    for (int hc = 0; hc < plan->n_hchunks; ++hc) {
      int k = 0;
      double* base = cf.data() + (size_t)hc * (per > 0 ? per : 1) * 256;
      for (int t = 0; t < CHUNK_TILES; ++t) for (int s = 0; s < nCP; ++s) frag32(&Wr1, hc * CHUNK_TILES + t, s, base + (size_t)(k++) * 256);
      for (int t = 0; t < CHUNK_TILES; ++t) for (int o = 0; o < n_ot; ++o) frag32(&Wr2, o, hc * CHUNK_TILES + t, base + (size_t)(k++) * 256);
    }
    plan->w_cfrag = blob.add(cf); plan->cfrag_per_chunk = per;
  }
  plan->total_floats = (int64_t)blob.n;
  return 0;
}

extern "C" size_t ftn_inception_pack_floats(int d_model, int d_ff, int n_kernels, const int* kh, const int* kw,
                                            double bottleneck_ratio, int engine) {
  Blob blob(nullptr, 0);
  FtnPlan plan;
  if (pack_impl(nullptr, nullptr, d_model, d_ff, n_kernels, kh, kw, bottleneck_ratio, 0, engine, blob, &plan)) return 0;
  return blob.n;
}

extern "C" int ftn_inception_pack_weights(const FtnInceptionBlockWeights* block0, const FtnInceptionBlockWeights* block2,
                                          int d_model, int d_ff, int n_kernels, const int* kh, const int* kw,
                                          double bottleneck_ratio, int act, int engine, float* blob_host,
                                          size_t blob_floats, FtnPlan* plan_out) {
  FTN_CHECK_ARG(block0 && block2 && blob_host && plan_out, "ftn_inception_pack_weights: null pointer");
  Blob blob(blob_host, blob_floats);
  const int rc = pack_impl(block0, block2, d_model, d_ff, n_kernels, kh, kw, bottleneck_ratio, act, engine, blob, plan_out);
  if (rc) return rc;
  FTN_CHECK_ARG(blob.ok, "ftn_inception_pack_weights: blob of %zu floats is too small (%zu needed)", blob_floats, blob.n);
  return 0;
}

// tile_search.h -- kernel K1: the AV1 intra encode loop for one tile, run by one wavefront.
//   superblocks in raster order -> top-down partition RDO (rav1e encode_partition_topdown /
//   rdo_partition_decision) -> per block: 13-mode SATD pre-filter, angle-delta refinement, full RD over
//   (mode x tx type) with forward transform, quantise, static-table rate, dequantise, inverse transform,
//   SSE (rdo_mode_decision / rdo_tx_type_decision), chroma DC / same-as-luma / CfL with alpha search
//   (rdo_cfl_alpha).  Source block, prediction, residual and candidate reconstructions live in LDS; the
//   frame-sized recon / coefficient / mode-info maps live in HBM and are touched only by the owning tile.
// The decisions are bit-identical to oracle/av1o_search.c (the CPU checker), which is how parity is tested.
#pragma once
#include "dev_common.h"
#include "dev_predict.h"
#include "dev_txfm.h"
#include "dev_rate.h"

#ifndef MI_K1_WAVES_PER_SIMD
#define MI_K1_WAVES_PER_SIMD 1
#endif
#ifndef MI_COST_IN_LDS
#define MI_COST_IN_LDS 1
#endif

template <int N> struct Scratch {
  static constexpr int CS = N < 32 ? N : 32;
  uint16_t ra[EDGE_LEN(N)], rl[EDGE_LEN(N)], wa[EDGE_LEN(N)], wl[EDGE_LEN(N)], etmp[2 * N + 16];
  uint16_t srcb[3][N * N];
  uint16_t pred[N * N], dcp[N * N], rec_tmp[N * N], rec_best[3][N * N], rec_c[2][N * N];
  int32_t tbuf[N * (N + 1)], cbuf[CS * CS], dq[CS * CS], qc_tmp[CS * CS], qc_best[3][CS * CS], qc_c[2][CS * CS];
  uint8_t lev[(CS + 4) * (CS + 4) + 4];
  long long satd[13];
  int order[13];
};

struct TxRes { int eob, cul, dcc; long long sse; uint32_t rate; };

template <int MAXN> struct Ctx {
  const FrameDev *f; TileB t; LDS Scratch<MAXN> *s; uint8_t *snap;
#if MI_COST_IN_LDS
  const LDS uint16_t *cost;
#else
  const uint16_t *cost;
#endif
  const LDS uint16_t *ls;      // LDS copies of the static rate table and the scan tables
};

__device__ __forceinline__ const int *intra_mode_ctx_tab() { static __device__ const int t[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 }; return t; }
#define IS_SMOOTH_(m) ((m) == SMOOTH_PRED || (m) == SMOOTH_V_PRED || (m) == SMOOTH_H_PRED)

__device__ inline void fill_map_dev(uint8_t *m, int ms, int r, int c, int n4, int v) {
  for (int i = LANE; i < n4 * n4; i += 64) m[(r + i / n4) * ms + c + (i % n4)] = (uint8_t)v;
}
__device__ inline void set_decoded_dev(const FrameDev *f, int r, int c, int n4, int v) {
  fill_map_dev(f->m_decoded, f->mi_stride, r, c, n4, v);
  WAVE_SYNC();
}

// 4x4-Hadamard SATD of (src - pred) over an n x n block; both in LDS with pitch n
__device__ inline long long satd_dev(const LDS uint16_t *src, const LDS uint16_t *pred, int n) {
  const int nb = n >> 2, tot = nb * nb;
  long long total = 0;
  for (int b = LANE; b < tot; b += 64) {
    const int by = (b / nb) * 4, bx = (b % nb) * 4;
    int d[16], t[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) d[i * 4 + j] = (int)src[(by + i) * n + bx + j] - (int)pred[(by + i) * n + bx + j];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int a = d[i * 4] + d[i * 4 + 1], b2 = d[i * 4] - d[i * 4 + 1], c2 = d[i * 4 + 2] + d[i * 4 + 3], e = d[i * 4 + 2] - d[i * 4 + 3];
      t[i * 4] = a + c2; t[i * 4 + 1] = b2 + e; t[i * 4 + 2] = a - c2; t[i * 4 + 3] = b2 - e;
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int a = t[j] + t[4 + j], b2 = t[j] - t[4 + j], c2 = t[8 + j] + t[12 + j], e = t[8 + j] - t[12 + j];
      s += iabs_(a + c2) + iabs_(b2 + e) + iabs_(a - c2) + iabs_(b2 - e);
    }
    total += s;
  }
  return wave_sum_i64(total);
}
__device__ inline long long sse_dev(const LDS uint16_t *a, const LDS uint16_t *b, int nn) {
  long long s = 0;
  for (int i = LANE; i < nn; i += 64) { const int d = (int)a[i] - (int)b[i]; s += (long long)d * d; }
  return wave_sum_i64(s);
}

template <int MAXN, int BS>
__device__ inline long long eval_tx(Ctx<MAXN> &k, int plane, int sctx, int dctx, const LDS uint16_t *pred, int txtype, int tx_off, int tx_sym,
                                    LDS uint16_t *rec_out, LDS int32_t *qc_out, TxRes *tr) {
  constexpr int n = 4 << BS, P = n + 1, CS = n < 32 ? n : 32;
  const FrameDev *f = k.f; LDS Scratch<MAXN> *S = k.s;
  const LDS uint16_t *src = S->srcb[plane];
  for (int idx = LANE; idx < n * n; idx += 64) {
    const int i = idx / n, j = idx % n;
    S->tbuf[i * P + j] = (int)src[idx] - (int)pred[idx];
    rec_out[idx] = pred[idx];
  }
  WAVE_SYNC();
  fwd_txfm2d_dev<n>(S->tbuf, S->cbuf, txtype);
  const int eob = quantize_dev(k.ls, S->cbuf, qc_out, CS, BS, txtype, f->dc_q[plane], f->ac_q[plane]);
  tr->rate = coef_rate_dev(k.cost, k.ls, qc_out, eob, plane, BS, txtype, sctx, dctx, tx_off, tx_sym, S->lev, &tr->cul, &tr->dcc);
  if (eob > 0) {
    dequantize_dev(qc_out, S->dq, CS, BS, f->dc_q[plane], f->ac_q[plane], f->bd);
    inv_txfm2d_add_dev<n>(S->dq, S->tbuf, rec_out, txtype, f->bd);
  }
  tr->eob = eob;
  tr->sse = sse_dev(src, rec_out, n * n);
  return ((tr->sse * f->wq[plane]) >> 5) + (((long long)tr->rate * f->rdmult + 256) >> 9);
}

template <int MAXN, int BS>
__device__ inline void commit_plane(Ctx<MAXN> &k, int plane, int r, int c, const LDS uint16_t *rec, const LDS int32_t *qc, const TxRes *tr) {
  constexpr int n = 4 << BS, CS = n < 32 ? n : 32, n4 = 1 << BS;
  const FrameDev *f = k.f;
  uint16_t *gr = f->rec[plane] + (size_t)(r * 4) * f->stride + c * 4;
  int32_t *gc = f->coef[plane] + (size_t)(r * 4) * f->stride + c * 4;
  for (int idx = LANE; idx < n * n; idx += 64) gr[(idx / n) * f->stride + (idx % n)] = rec[idx];
  for (int idx = LANE; idx < CS * CS; idx += 64) gc[(idx / CS) * f->stride + (idx % CS)] = qc[idx];
  fill_map_dev(f->m_lvl[plane], f->mi_stride, r, c, n4, tr->cul);
  fill_map_dev(f->m_dc[plane], f->mi_stride, r, c, n4, tr->dcc);
  if (LANE == 0) f->m_eob[plane][r * f->mi_stride + c] = (uint16_t)tr->eob;
}

template <int MAXN, int BS>
__device__ inline void load_src_block(Ctx<MAXN> &k, int plane, int r, int c) {
  constexpr int n = 4 << BS;
  const FrameDev *f = k.f;
  const uint16_t *g = f->src[plane] + (size_t)(r * 4) * f->stride + c * 4;
  for (int idx = LANE; idx < n * n; idx += 64) k.s->srcb[plane][idx] = g[(idx / n) * f->stride + (idx % n)];
}

template <int MAXN, int BS>
__device__ long long try_block(Ctx<MAXN> &k, int r, int c) {
  constexpr int n = 4 << BS, n4 = 1 << BS, log2w = 2 + BS, nn = n * n, CS = n < 32 ? n : 32, qn = CS * CS;
  const FrameDev *f = k.f; const TileB *t = &k.t; LDS Scratch<MAXN> *S = k.s;
  const int ms = f->mi_stride, mi = r * ms + c, x = c * 4, y = r * 4;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int have_ar = availU && (c + n4 < t->mi_col_end) && f->m_decoded[(r - 1) * ms + c + n4];
  const int have_bl = availL && (r + n4 < t->mi_row_end) && f->m_decoded[(r + n4) * ms + c - 1];
  const int amode = availU ? f->m_ymode[mi - ms] : DC_PRED, lmode = availL ? f->m_ymode[mi - 1] : DC_PRED;
  const int *imc = intra_mode_ctx_tab();
  const auto *ycost = k.cost + CDF_KF_Y + (imc[amode] * 5 + imc[lmode]) * CDF_KF_Y_STRIDE;
  const int ftype_y = (availU && IS_SMOOTH_(f->m_ymode[mi - ms])) || (availL && IS_SMOOTH_(f->m_ymode[mi - 1]));
  int ftype_uv = 0;
  if (f->np > 1) ftype_uv = (availU && IS_SMOOTH_(f->m_uvmode[mi - ms])) || (availL && IS_SMOOTH_(f->m_uvmode[mi - 1]));
  LDS uint16_t *ra = S->ra + EDGE_OFF, *rl = S->rl + EDGE_OFF, *wa = S->wa + EDGE_OFF, *wl = S->wl + EDGE_OFF;

  for (int p = 0; p < f->np; p++) load_src_block<MAXN, BS>(k, p, r, c);
  int sctx_p[3] = { 0, 0, 0 }, dctx_p[3] = { 0, 0, 0 };      // all-zero / dc-sign contexts depend on the neighbours only
  for (int p = 0; p < f->np; p++) txb_ctx_dev(f, t, p, r, c, BS, BS, &sctx_p[p], &dctx_p[p]);
  if (f->dbg == 2) return 0;
  load_edges(f, 0, x, y, n, availL, availU, have_ar, have_bl, ra, rl);   // ends with WAVE_SYNC
  if (f->dbg == 3) return 0;

  // ---- luma: SATD pre-filter over the 13 modes ----
  for (int m = 0; m < 13; m++) {
    predict_block(f, x, y, log2w, availL, availU, m, 0, ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
    const long long sd = satd_dev(S->srcb[0], S->pred, n);
    if (LANE == 0) { S->satd[m] = sd; S->order[m] = m; }
  }
  WAVE_SYNC();
  if (LANE == 0) {
    for (int i = 1; i < 13; i++) { const int v = S->order[i]; int j = i; while (j > 0 && S->satd[S->order[j - 1]] > S->satd[v]) { S->order[j] = S->order[j - 1]; j--; } S->order[j] = v; }
  }
  WAVE_SYNC();
  if (f->dbg == 4) return 0;
  const int ncand = f->complex_modes ? 7 : 3;
  long long best_j = 0x7fffffffffffffffLL; int best_mode = DC_PRED, best_delta = 0, best_tx = DCT_DCT; TxRes best_tr = { 0, 0, 0, 0, 0 };
  int tx_ns = 0, tx_set = 0;
  for (int ci = 0; ci < ncand; ci++) {
    const int m = S->order[ci];
    int delta = 0;
    const int directional = m >= V_PRED && m <= D67_PRED;
    if (directional && BS >= BS_8 && f->fine_directional) {
      long long bsd = S->satd[m];
      const int dl[6] = { -1, 1, -2, 2, -3, 3 };
      for (int q = 0; q < 6; q++) {
        predict_block(f, x, y, log2w, availL, availU, m, dl[q], ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
        const long long sd = satd_dev(S->srcb[0], S->pred, n);
        if (sd < bsd) { bsd = sd; delta = dl[q]; }
      }
    }
    predict_block(f, x, y, log2w, availL, availU, m, delta, ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
    uint32_t mode_rate = ycost[m];
    if (directional && BS >= BS_8) mode_rate += k.cost[CDF_ANGLE + (m - V_PRED) * CDF_ANGLE_STRIDE + delta + 3];
    const int tx_off = intra_tx_cdf(f, BS, m, &tx_ns, &tx_set);
    const int ntx = (f->rdo_tx && tx_off >= 0) ? tx_ns : 1;
    for (int ti = 0; ti < ntx; ti++) {
      int txtype;
      if (ntx > 1) txtype = sym_to_txtype(tx_set, ti);
      else { txtype = mode_to_txtype(m); if (tx_off < 0 || txtype_to_sym(tx_set, txtype) < 0) txtype = DCT_DCT; }
      TxRes tr;
      long long j = eval_tx<MAXN, BS>(k, 0, sctx_p[0], dctx_p[0], S->pred, txtype, tx_off, tx_off >= 0 ? txtype_to_sym(tx_set, txtype) : 0, S->rec_tmp, S->qc_tmp, &tr);
      j += ((long long)mode_rate * f->rdmult + 256) >> 9;
      if (j < best_j) {
        best_j = j; best_mode = m; best_delta = delta; best_tx = txtype; best_tr = tr;
        for (int i = LANE; i < nn; i += 64) S->rec_best[0][i] = S->rec_tmp[i];
        for (int i = LANE; i < qn; i += 64) S->qc_best[0][i] = S->qc_tmp[i];
        WAVE_SYNC();
      }
    }
  }
  if (f->dbg == 5) return 0;
  commit_plane<MAXN, BS>(k, 0, r, c, S->rec_best[0], S->qc_best[0], &best_tr);
  fill_map_dev(f->m_ymode, ms, r, c, n4, best_mode);
  fill_map_dev((uint8_t *)f->m_angle_y, ms, r, c, n4, (uint8_t)(int8_t)best_delta);
  fill_map_dev(f->m_txtype, ms, r, c, n4, best_tr.eob ? best_tx : DCT_DCT);
  fill_map_dev(f->m_bsize, ms, r, c, n4, BS);
  WAVE_SYNC();
  long long total_j = best_j; int any_coef = best_tr.eob > 0;

  // ---- chroma ----
  if (f->np > 1) {
    const int cfl_allowed = BS <= BS_32;
    const auto *uvcost = cfl_allowed ? k.cost + CDF_UV_CFL + best_mode * CDF_UV_CFL_STRIDE : k.cost + CDF_UV_NOCFL + best_mode * CDF_UV_NOCFL_STRIDE;
    int cands[16], nc = 0;
    cands[nc++] = DC_PRED;
    if (best_mode != DC_PRED) cands[nc++] = best_mode;
    if (f->complex_modes) for (int m = 1; m < 13; m++) if (m != best_mode) cands[nc++] = m;
    if (cfl_allowed) cands[nc++] = UV_CFL_PRED;
    long long best_uv = 0x7fffffffffffffffLL; int buv = DC_PRED, bdelta = 0, bsign = 0, bau = 0, bav = 0; TxRes btr[3];
    btr[1] = best_tr; btr[2] = best_tr;
    const int uvset = tx_set_of(BS, f->reduced_tx_set);
    // luma average for CfL (same for every alpha)
    for (int ci = 0; ci < nc; ci++) {
      const int um = cands[ci];
      const int delta = (um == best_mode && um >= V_PRED && um <= D67_PRED && BS >= BS_8) ? best_delta : 0;
      int alpha[3] = { 0, 0, 0 }, jsign = 0;
      uint32_t mode_rate = uvcost[um];
      if (um >= V_PRED && um <= D67_PRED && BS >= BS_8) mode_rate += k.cost[CDF_ANGLE + (um - V_PRED) * CDF_ANGLE_STRIDE + delta + 3];
      int txtype = mode_to_txtype(um);
      if (txtype_to_sym(uvset, txtype) < 0) txtype = DCT_DCT;
      long long j = 0; TxRes trs[3]; int ok = 1;
      if (um == UV_CFL_PRED) {
        // rdo_cfl_alpha: per plane the alpha in -16..16 minimising prediction SSE
        const uint16_t *luma = f->rec[0] + (size_t)y * f->stride + x;
        int ls = 0;
        for (int idx = LANE; idx < nn; idx += 64) ls += luma[(idx / n) * f->stride + (idx % n)] << 3;
        ls = wave_sum_i32(ls);
        const int avg = round2_(ls, 2 * log2w), mx = (1 << f->bd) - 1;
        for (int p = 1; p < 3; p++) {
          load_edges(f, p, x, y, n, availL, availU, have_ar, have_bl, ra, rl);
          predict_block(f, x, y, log2w, availL, availU, DC_PRED, 0, ftype_uv, ra, rl, wa, wl, S->etmp, S->dcp);
          long long e0 = 0, e[32];
#pragma unroll
          for (int a = 0; a < 32; a++) e[a] = 0;
          for (int idx = LANE; idx < nn; idx += 64) {
            const int l = (luma[(idx / n) * f->stride + (idx % n)] << 3) - avg, dcv = S->dcp[idx], sv = S->srcb[p][idx];
            { const int d = sv - dcv; e0 += (long long)d * d; }
#pragma unroll
            for (int a = 0; a < 32; a++) {
              const int al = (a & 1) ? -((a >> 1) + 1) : ((a >> 1) + 1);
              const int v = al * l, sc = v >= 0 ? round2_(v, 6) : -round2_(-v, 6);
              const int d = sv - iclamp_(dcv + sc, 0, mx);
              e[a] += (long long)d * d;
            }
          }
          long long best_sse = wave_sum_i64(e0); int best_a = 0;
#pragma unroll
          for (int a = 0; a < 32; a++) {
            const long long ea = wave_sum_i64(e[a]);
            if (ea < best_sse) { best_sse = ea; best_a = (a & 1) ? -((a >> 1) + 1) : ((a >> 1) + 1); }
          }
          alpha[p] = best_a;
        }
        if (alpha[1] == 0 && alpha[2] == 0) ok = 0;
        else {
          const int su = alpha[1] == 0 ? 0 : (alpha[1] < 0 ? 1 : 2), sv = alpha[2] == 0 ? 0 : (alpha[2] < 0 ? 1 : 2);
          jsign = su * 3 + sv - 1;
          mode_rate += k.cost[CDF_CFL_SIGN + jsign];
          if (su) mode_rate += k.cost[CDF_CFL_ALPHA + ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE + iabs_(alpha[1]) - 1];
          if (sv) mode_rate += k.cost[CDF_CFL_ALPHA + ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE + iabs_(alpha[2]) - 1];
        }
      }
      if (!ok) continue;
      for (int p = 1; p < 3; p++) {
        load_edges(f, p, x, y, n, availL, availU, have_ar, have_bl, ra, rl);
        if (um == UV_CFL_PRED) {
          predict_block(f, x, y, log2w, availL, availU, DC_PRED, 0, ftype_uv, ra, rl, wa, wl, S->etmp, S->dcp);
          if (alpha[p]) predict_cfl_dev(f, x, y, log2w, alpha[p], S->dcp, S->pred);
          else { for (int i = LANE; i < nn; i += 64) S->pred[i] = S->dcp[i]; WAVE_SYNC(); }
        } else {
          predict_block(f, x, y, log2w, availL, availU, um, delta, ftype_uv, ra, rl, wa, wl, S->etmp, S->pred);
        }
        j += eval_tx<MAXN, BS>(k, p, sctx_p[p], dctx_p[p], S->pred, txtype, -1, 0, S->rec_best[p], S->qc_best[p], &trs[p]);
      }
      j += ((long long)mode_rate * f->rdmult + 256) >> 9;
      if (j < best_uv) {
        best_uv = j; buv = um; bdelta = delta; bsign = jsign; bau = alpha[1]; bav = alpha[2]; btr[1] = trs[1]; btr[2] = trs[2];
        for (int p = 1; p < 3; p++) {
          for (int i = LANE; i < nn; i += 64) S->rec_c[p - 1][i] = S->rec_best[p][i];
          for (int i = LANE; i < qn; i += 64) S->qc_c[p - 1][i] = S->qc_best[p][i];
        }
        WAVE_SYNC();
      }
    }
    for (int p = 1; p < 3; p++) { commit_plane<MAXN, BS>(k, p, r, c, S->rec_c[p - 1], S->qc_c[p - 1], &btr[p]); any_coef |= btr[p].eob > 0; }
    fill_map_dev(f->m_uvmode, ms, r, c, n4, buv);
    fill_map_dev((uint8_t *)f->m_angle_uv, ms, r, c, n4, (uint8_t)(int8_t)bdelta);
    fill_map_dev(f->m_cfl_sign, ms, r, c, n4, bsign);
    fill_map_dev(f->m_cfl_au, ms, r, c, n4, bau ? iabs_(bau) - 1 : 0);
    fill_map_dev(f->m_cfl_av, ms, r, c, n4, bav ? iabs_(bav) - 1 : 0);
    total_j += best_uv;
  }
  if (f->dbg == 6) return 0;
  // ---- skip flag ----
  const int skip = !any_coef;
  WAVE_SYNC();
  fill_map_dev(f->m_skip, ms, r, c, n4, skip);
  if (skip) for (int p = 0; p < f->np; p++) { fill_map_dev(f->m_lvl[p], ms, r, c, n4, 0); fill_map_dev(f->m_dc[p], ms, r, c, n4, 0); }
  const int sctx = (availU ? f->m_skip[mi - ms] : 0) + (availL ? f->m_skip[mi - 1] : 0);
  total_j += ((long long)k.cost[CDF_SKIP + sctx * CDF_SKIP_STRIDE + skip] * f->rdmult + 256) >> 9;
  set_decoded_dev(f, r, c, n4, 1);
  return total_j;
}

// ---- area snapshot (NONE-vs-SPLIT comparison), kept in the tile's HBM scratch ----
template <int BS> __device__ inline void area_copy_dev(const FrameDev *f, uint8_t *snap, int r, int c, int save) {
  constexpr int n = 4 << BS, n4 = 1 << BS;
  uint16_t *srec = (uint16_t *)snap;                         // [3][n*n]
  int32_t *scoef = (int32_t *)(snap + 3 * n * n * 2);        // [3][n*n]
  uint8_t *smaps = snap + 3 * n * n * 6;                     // 16 byte-maps [n4*n4]
  uint16_t *seob = (uint16_t *)(smaps + 16 * n4 * n4);       // [3][n4*n4]
  for (int p = 0; p < f->np; p++) {
    uint16_t *gr = f->rec[p] + (size_t)(r * 4) * f->stride + c * 4;
    int32_t *gc = f->coef[p] + (size_t)(r * 4) * f->stride + c * 4;
    for (int idx = LANE; idx < n * n; idx += 64) {
      const int o = (idx / n) * f->stride + (idx % n);
      if (save) { srec[p * n * n + idx] = gr[o]; scoef[p * n * n + idx] = gc[o]; }
      else { gr[o] = srec[p * n * n + idx]; gc[o] = scoef[p * n * n + idx]; }
    }
  }
  uint8_t *maps[16] = { f->m_bsize, f->m_skip, f->m_ymode, f->m_uvmode, f->m_txtype, f->m_cfl_sign, f->m_cfl_au, f->m_cfl_av,
                        (uint8_t *)f->m_angle_y, (uint8_t *)f->m_angle_uv, f->m_lvl[0], f->m_lvl[1], f->m_lvl[2], f->m_dc[0], f->m_dc[1], f->m_dc[2] };
  for (int m = 0; m < 16; m++) {
    if (f->np == 1 && (m == 11 || m == 12 || m == 14 || m == 15)) continue;
    uint8_t *g = maps[m];
    for (int i = LANE; i < n4 * n4; i += 64) {
      const int o = (r + i / n4) * f->mi_stride + c + (i % n4);
      if (save) smaps[m * n4 * n4 + i] = g[o]; else g[o] = smaps[m * n4 * n4 + i];
    }
  }
  for (int p = 0; p < f->np; p++)
    for (int i = LANE; i < n4 * n4; i += 64) {
      const int o = (r + i / n4) * f->mi_stride + c + (i % n4);
      if (save) seob[p * n4 * n4 + i] = f->m_eob[p][o]; else f->m_eob[p][o] = seob[p * n4 * n4 + i];
    }
  WAVE_SYNC();
}
#define MI_SNAP_BYTES(n) (3 * (n) * (n) * 6 + 16 * ((n) / 4) * ((n) / 4) + 3 * ((n) / 4) * ((n) / 4) * 2 + 64)

template <typename CostPtr> __device__ inline uint32_t partition_rate_dev(CostPtr cost, const FrameDev *f, const TileB *t, int r, int c, int bs, int part) {
  const int ms = f->mi_stride;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int above = availU && f->m_bsize[(r - 1) * ms + c] < bs, left = availL && f->m_bsize[r * ms + c - 1] < bs;
  return cost[CDF_PARTITION + ((bs - 1) * 4 + left * 2 + above) * CDF_PARTITION_STRIDE + part];
}

template <int MAXN, int MAXBS, int BS> struct RdPart {
  static __device__ void run(Ctx<MAXN> &k, int r, int c) {
    const FrameDev *f = k.f;
    if (r >= f->mi_rows || c >= f->mi_cols) return;
    constexpr int half = (1 << BS) >> 1, px = 4 << BS, n4 = 1 << BS;
    const int has_rows = (r + half) < f->mi_rows, has_cols = (c + half) < f->mi_cols;
    const int must_split = px > f->part_max || !has_rows || !has_cols;
    const int can_split = px > f->part_min || must_split;
    set_decoded_dev(f, r, c, n4, 0);
    if (!can_split || (f->dbg == 9 && BS == 1)) { if constexpr (BS <= MAXBS) try_block<MAXN, BS>(k, r, c); return; }
    int do_split = must_split;
    if constexpr (BS <= MAXBS) {
      if (!must_split) {
        const long long j_none = try_block<MAXN, BS>(k, r, c) + (((long long)partition_rate_dev(k.cost, f, &k.t, r, c, BS, 0) * f->rdmult + 256) >> 9);
        area_copy_dev<BS>(f, k.snap, r, c, 1);
        set_decoded_dev(f, r, c, n4, 0);
        long long j_split = ((long long)partition_rate_dev(k.cost, f, &k.t, r, c, BS, 3) * f->rdmult + 256) >> 9;
        for (int q = 0; q < 4 && j_split < j_none && f->dbg != 7; q++) {
          const int rr = r + (q >> 1) * half, cc = c + (q & 1) * half;
          if (rr >= f->mi_rows || cc >= f->mi_cols) continue;
          j_split += try_block<MAXN, BS - 1>(k, rr, cc);
          if (BS - 1 >= BS_8) j_split += ((long long)partition_rate_dev(k.cost, f, &k.t, rr, cc, BS - 1, 0) * f->rdmult + 256) >> 9;
        }
        if (j_split < j_none && f->dbg != 7 && f->dbg != 8 && !(f->dbg == 10 && BS == 1)) do_split = 1;
        else { area_copy_dev<BS>(f, k.snap, r, c, 0); set_decoded_dev(f, r, c, n4, 1); }
      }
    }
    if (do_split) {
      set_decoded_dev(f, r, c, n4, 0);
      RdPart<MAXN, MAXBS, BS - 1>::run(k, r, c); RdPart<MAXN, MAXBS, BS - 1>::run(k, r, c + half);
      RdPart<MAXN, MAXBS, BS - 1>::run(k, r + half, c); RdPart<MAXN, MAXBS, BS - 1>::run(k, r + half, c + half);
    }
  }
};
template <int MAXN, int MAXBS> struct RdPart<MAXN, MAXBS, 0> {
  static __device__ void run(Ctx<MAXN> &k, int r, int c) {
    const FrameDev *f = k.f;
    if (r >= f->mi_rows || c >= f->mi_cols) return;
    set_decoded_dev(f, r, c, 1, 0);
    try_block<MAXN, 0>(k, r, c);
  }
};

template <int MAXBS>
__global__ __launch_bounds__(64, MI_K1_WAVES_PER_SIMD) void tile_search_kernel(const FrameDev *frames, const TileJob *jobs, int njobs) {
  constexpr int MAXN = 4 << MAXBS;
  extern __shared__ __align__(16) uint8_t smem[];
  const int job = blockIdx.x;
  if (job >= njobs) return;
  const TileJob tj = jobs[job];
  const FrameDev *f = frames + tj.frame;
  Ctx<MAXN> k;
  k.f = f; k.s = (LDS Scratch<MAXN> *)smem;
  {
    LDS uint16_t *lc = (LDS uint16_t *)(smem + ((sizeof(Scratch<MAXN>) + 15) & ~(size_t)15));
    LDS uint16_t *lsc = lc + CDF_TOTAL;
#if MI_COST_IN_LDS
    for (int i = LANE; i < CDF_TOTAL; i += 64) lc[i] = f->cost[i];
    k.cost = lc;
#else
    lsc = lc; k.cost = f->cost;
#endif
    load_scans_to_lds(lsc, MAXN);
    k.ls = lsc;
    WAVE_SYNC();
  }
  k.t.mi_row_start = f->tile_row_start[tj.tile_row] * 16; k.t.mi_row_end = imin_(f->tile_row_start[tj.tile_row + 1] * 16, f->mi_rows);
  k.t.mi_col_start = f->tile_col_start[tj.tile_col] * 16; k.t.mi_col_end = imin_(f->tile_col_start[tj.tile_col + 1] * 16, f->mi_cols);
  k.snap = f->snap + (size_t)(tj.tile_row * f->tile_cols + tj.tile_col) * MI_SNAP_BYTES(MAXN);
  if (f->dbg == 1) return;
  const unsigned long long clk0 = wall_clock64();
  for (int r = k.t.mi_row_start; r < k.t.mi_row_end; r += 16)
    for (int c = k.t.mi_col_start; c < k.t.mi_col_end; c += 16)
      RdPart<MAXN, MAXBS, 4>::run(k, r, c);
  if (LANE == 0) { unsigned long long *tc = f->tile_clk + (size_t)(tj.tile_row * f->tile_cols + tj.tile_col) * 4; tc[0] = clk0; tc[1] = wall_clock64(); }
}

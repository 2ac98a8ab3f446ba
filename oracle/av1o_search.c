/* oracle/av1o_search.c -- phase 1: per-tile partition / mode / tx-type RDO with reconstruction.
 * TEST INFRASTRUCTURE (see av1o.h).
 * Structure follows rav1e (all absent from /root/reference):
 *   encode_partition_topdown + rdo_partition_decision  -> rd_partition()
 *   rdo_mode_decision / intra_frame_rdo_mode_decision  -> try_block(): SATD pre-filter, then full RD
 *   rdo_tx_type_decision                               -> loop over the intra tx set per surviving mode
 *   rdo_cfl_alpha                                      -> cfl_search()
 * Deliberate simplifications (documented in DESIGN.md): rates come from the frame's initial CDFs (no
 * in-tile adaptation during search), distortion is plain SSE, square partitions only, TX_MODE_LARGEST. */
#include "av1o_int.h"
#include "av1o_syms.h"

static const uint8_t intra_mode_ctx[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 };

typedef struct {
  Av1oFrame *f; TileB t;
  int64_t wq[3];           /* plane distortion weights, Q12 */
  int dcq[3], acq[3];      /* the quantiser steps of the block being evaluated (its segment's) */
} Search;

typedef struct {            /* saved state of one block area (for NONE-vs-SPLIT comparison) */
  int n;
  uint16_t rec[3][64 * 64];
  int32_t coef[3][64 * 64];
  uint8_t maps[12][16 * 16];
  uint8_t lvl[3][16 * 16], dc[3][16 * 16];
  uint16_t eob[3][16 * 16];
} AreaSnap;

static uint8_t *map_ptr(Av1oFrame *f, int i) {
  switch (i) {
    case 0: return f->m_bsize; case 1: return f->m_skip; case 2: return f->m_ymode; case 3: return f->m_uvmode;
    case 4: return f->m_txtype; case 5: return f->m_cfl_sign; case 6: return f->m_cfl_au; case 7: return f->m_cfl_av;
    case 8: return (uint8_t *)f->m_angle_y; case 9: return (uint8_t *)f->m_angle_uv; case 10: return f->m_txsize; default: return f->m_seg;
  }
}
static void area_copy(Av1oFrame *f, AreaSnap *s, int r, int c, int bs, int save) {
  const int n = 4 << bs, n4 = 1 << bs;
  s->n = n;
  for (int p = 0; p < f->np; p++) {
    for (int i = 0; i < n; i++) {
      uint16_t *fr = f->rec[p] + (r * 4 + i) * f->stride + c * 4;
      int32_t *fc = f->coef[p] + (r * 4 + i) * f->stride + c * 4;
      if (save) { memcpy(s->rec[p] + i * n, fr, 2 * (size_t)n); memcpy(s->coef[p] + i * n, fc, 4 * (size_t)n); }
      else { memcpy(fr, s->rec[p] + i * n, 2 * (size_t)n); memcpy(fc, s->coef[p] + i * n, 4 * (size_t)n); }
    }
    for (int i = 0; i < n4; i++) {
      int o = (r + i) * f->mi_stride + c;
      if (save) { memcpy(s->lvl[p] + i * n4, f->m_lvl[p] + o, (size_t)n4); memcpy(s->dc[p] + i * n4, f->m_dc[p] + o, (size_t)n4); memcpy(s->eob[p] + i * n4, f->m_eob[p] + o, 2 * (size_t)n4); }
      else { memcpy(f->m_lvl[p] + o, s->lvl[p] + i * n4, (size_t)n4); memcpy(f->m_dc[p] + o, s->dc[p] + i * n4, (size_t)n4); memcpy(f->m_eob[p] + o, s->eob[p] + i * n4, 2 * (size_t)n4); }
    }
  }
  for (int m = 0; m < 12; m++) {
    uint8_t *mp = map_ptr(f, m);
    for (int i = 0; i < n4; i++) {
      if (save) memcpy(s->maps[m] + i * n4, mp + (r + i) * f->mi_stride + c, (size_t)n4);
      else memcpy(mp + (r + i) * f->mi_stride + c, s->maps[m] + i * n4, (size_t)n4);
    }
  }
}
/* the two chroma planes of a 64x64 block: reconstruction, levels and the contexts its four transform blocks per plane leave behind */
typedef struct { uint16_t rec[2][64 * 64]; int32_t coef[2][64 * 64]; uint8_t lvl[2][256], dc[2][256]; uint16_t eob[2][256]; } ChromaSnap64;
static void chroma_copy64(Av1oFrame *f, ChromaSnap64 *s, int r, int c, int save) {
  for (int p = 1; p < 3; p++) {
    for (int i = 0; i < 64; i++) {
      uint16_t *fr = f->rec[p] + (r * 4 + i) * f->stride + c * 4; int32_t *fc = f->coef[p] + (r * 4 + i) * f->stride + c * 4;
      if (save) { memcpy(s->rec[p - 1] + i * 64, fr, 128); memcpy(s->coef[p - 1] + i * 64, fc, 256); }
      else { memcpy(fr, s->rec[p - 1] + i * 64, 128); memcpy(fc, s->coef[p - 1] + i * 64, 256); }
    }
    for (int i = 0; i < 16; i++) {
      const int o = (r + i) * f->mi_stride + c;
      if (save) { memcpy(s->lvl[p - 1] + i * 16, f->m_lvl[p] + o, 16); memcpy(s->dc[p - 1] + i * 16, f->m_dc[p] + o, 16); memcpy(s->eob[p - 1] + i * 16, f->m_eob[p] + o, 32); }
      else { memcpy(f->m_lvl[p] + o, s->lvl[p - 1] + i * 16, 16); memcpy(f->m_dc[p] + o, s->dc[p - 1] + i * 16, 16); memcpy(f->m_eob[p] + o, s->eob[p - 1] + i * 16, 32); }
    }
  }
}
static void set_decoded(Av1oFrame *f, int r, int c, int bs, int v) {
  const int w4 = 1 << (dim_wl(bs) - 2), h4 = 1 << (dim_hl(bs) - 2);
  for (int i = 0; i < h4; i++) memset(f->m_decoded + (r + i) * f->mi_stride + c, v, (size_t)w4);
}
static void fill_map2(uint8_t *m, int ms, int r, int c, int w4, int h4, int v) {
  for (int i = 0; i < h4; i++) memset(m + (r + i) * ms + c, v, (size_t)w4);
}
static void fill_map(uint8_t *m, int ms, int r, int c, int n4, int v) { fill_map2(m, ms, r, c, n4, n4, v); }

/* Ablation switches (environment, oracle only -- the divergence ledger of BASELINE.md section 5; tools/divergence_ledger.py): each turns ONE deliberate difference from what
 * rav1e does (as recalled) into rav1e's form -- or, for a difference that has since been closed, back into this encoder's earlier form --, so that its effect on bytes / error
 * can be measured.  0 / unset = this encoder's shipped behaviour (what HIP == oracle tests pin).  Read once per process.
 *   AV1O_ABL_SATD4=1        (round 6: closed) SATD of every block with 4x4 Hadamards, as rounds 1-5 shipped, instead of the 8x8 Hadamard for blocks of 8x8 and more (rav1e get_satd)
 *   AV1O_ABL_SEQ_TXTYPE=1   mode decision with each mode's default transform type, then the transform-type search on the winning mode only (instead of mode x type jointly)
 *   AV1O_ABL_SUB_TXTYPE=1   (round 6: closed) every sub-block of a split transform picks its own transform type, as rounds 1-5 shipped, instead of one type for the whole
 *                           block per transform size (rav1e rdo_tx_size_type / rdo_tx_type_decision) */
enum { ABL_SATD4 = 1, ABL_SEQ_TXTYPE = 2, ABL_SUB_TXTYPE = 4 };
static int abl_flags(void) {
  static int cached = -1;                                     /* (a benign race: every thread computes the same value) */
  if (cached < 0) {
    int v = 0; const char *e;
    if ((e = getenv("AV1O_ABL_SATD4")) && e[0] == '1') v |= ABL_SATD4;
    if ((e = getenv("AV1O_ABL_SEQ_TXTYPE")) && e[0] == '1') v |= ABL_SEQ_TXTYPE;
    if ((e = getenv("AV1O_ABL_SUB_TXTYPE")) && e[0] == '1') v |= ABL_SUB_TXTYPE;
    cached = v;
  }
  return cached;
}
/* SATD of blocks of 8x8 and more: the 8x8 Hadamard per 8x8 cell (rav1e get_satd), its sum brought to the scale of four 4x4 Hadamards.  Three butterfly stages per
 * direction; the sum of |H D H^T| does not depend on the order of the stages or of the two directions (cavif_rs_amd/csrc: satd_dev / satd_group<8> run them in another). */
static int64_t satd8_block(const uint16_t *src, int ss, const uint16_t *pred, int ps, int w, int h) {
  int64_t total = 0;
  for (int by = 0; by < h; by += 8) for (int bx = 0; bx < w; bx += 8) {
    int m[64];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) m[i * 8 + j] = (int)src[(by + i) * ss + bx + j] - (int)pred[(by + i) * ps + bx + j];
    for (int pass = 0; pass < 2; pass++) for (int i = 0; i < 8; i++) {                       /* rows, then columns: three butterfly stages each */
      int *v = m + (pass ? i : i * 8); const int st = pass ? 8 : 1;
      for (int len = 1; len < 8; len <<= 1) for (int k = 0; k < 8; k += 2 * len) for (int q = 0; q < len; q++) {
        const int a = v[(k + q) * st], b = v[(k + q + len) * st]; v[(k + q) * st] = a + b; v[(k + q + len) * st] = a - b;
      }
    }
    int64_t s = 0; for (int i = 0; i < 64; i++) s += iabs(m[i]);
    total += (s + 2) >> 2;                                                                   /* the 8x8 sum on the scale of four 4x4 ones */
  }
  return total;
}
/* SATD of a block: 8x8 Hadamards where both dimensions reach 8 (above), 4x4 Hadamards for the 4x4, 8x4 and 4x8 blocks */
static int64_t satd_block_wh(const uint16_t *src, int ss, const uint16_t *pred, int ps, int w, int h) {
  if (w >= 8 && h >= 8 && !(abl_flags() & ABL_SATD4)) return satd8_block(src, ss, pred, ps, w, h);
  int64_t total = 0;
  for (int by = 0; by < h; by += 4) for (int bx = 0; bx < w; bx += 4) {
    int d[16], t[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) d[i * 4 + j] = (int)src[(by + i) * ss + bx + j] - (int)pred[(by + i) * ps + bx + j];
    for (int i = 0; i < 4; i++) {
      int a = d[i * 4] + d[i * 4 + 1], b = d[i * 4] - d[i * 4 + 1], c2 = d[i * 4 + 2] + d[i * 4 + 3], e = d[i * 4 + 2] - d[i * 4 + 3];
      t[i * 4] = a + c2; t[i * 4 + 1] = b + e; t[i * 4 + 2] = a - c2; t[i * 4 + 3] = b - e;
    }
    int s = 0;
    for (int j = 0; j < 4; j++) {
      int a = t[j] + t[4 + j], b = t[j] - t[4 + j], c2 = t[8 + j] + t[12 + j], e = t[8 + j] - t[12 + j];
      s += iabs(a + c2) + iabs(b + e) + iabs(a - c2) + iabs(b - e);
    }
    total += s;
  }
  return total;
}
static int64_t satd_block(const uint16_t *src, int ss, const uint16_t *pred, int ps, int n) { return satd_block_wh(src, ss, pred, ps, n, n); }
static int64_t sse_block_wh(const uint16_t *a, int as, const uint16_t *b, int bs_, int w, int h) {
  int64_t s = 0;
  for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) { int d = (int)a[i * as + j] - (int)b[i * bs_ + j]; s += (int64_t)d * d; }
  return s;
}
static int64_t sse_block(const uint16_t *a, int as, const uint16_t *b, int bs_, int n) { return sse_block_wh(a, as, b, bs_, n, n); }

/* One transform block: residual -> fwd -> quant -> rate, dequant -> inverse -> recon; returns weighted J. */
typedef struct { int eob, cul, dcc; int64_t sse; uint32_t rate; } TxRes;
static int64_t eval_tx(Search *s, int plane, int r, int c, int txs, int bs /* block size (all-zero context) */, const uint16_t *pred /* n x n, stride n */, int txtype,
                       int tx_off, int tx_sym, int tx_ns, uint16_t *rec_out /* n x n */, int32_t *qc_out, TxRes *tr) {
  Av1oFrame *f = s->f;
  const int n = 1 << dim_wl(txs), nh = 1 << dim_hl(txs), cs = imin(n, 32), x = c * 4, y = r * 4;      /* pred / rec_out: nh rows of n samples */
  static __thread int16_t resid[64 * 64]; static __thread int32_t coef[32 * 32], dq[32 * 32];
  const uint16_t *src = f->src[plane] + y * f->stride + x;
  for (int i = 0; i < nh; i++) for (int j = 0; j < n; j++) resid[i * n + j] = (int16_t)((int)src[i * f->stride + j] - (int)pred[i * n + j]);
  av1o_fwd_txfm2d(resid, n, coef, txs, txtype, f->bd);
  int eob = av1o_quantize(coef, qc_out, txs, txtype, s->dcq[plane], s->acq[plane]);
  int sctx, dctx;
  av1o_txb_ctx(f, &s->t, plane, r, c, txs, bs, &sctx, &dctx);
  tr->rate = av1o_coef_rate_full(f, qc_out, eob, plane, txs, txtype, sctx, dctx, tx_off, tx_sym, tx_ns, &tr->cul, &tr->dcc);
  memcpy(rec_out, pred, sizeof(uint16_t) * (size_t)(n * nh));
  if (eob > 0) {
    av1o_dequantize(qc_out, dq, txs, s->dcq[plane], s->acq[plane], f->bd, eob, txtype);
    av1o_inv_txfm2d_add(dq, rec_out, n, txs, txtype, f->bd);
  }
  tr->eob = eob;
  /* distortion: luma = psychovisual cdef-dist per 8x8 cell x activity; chroma = SSE x the block's mean activity */
  if (plane == 0) tr->sse = av1o_psy_dist_luma_wh(f, rec_out, n, x, y, n, nh);
  else tr->sse = (sse_block_wh(src, f->stride, rec_out, n, n, nh) * av1o_act_mean(f, x, y, n, nh) + 8192) >> 14;
  (void)cs;
  return ((tr->sse * s->wq[plane]) >> 5) + (((int64_t)tr->rate * f->rdmult[0] + 256) >> 9);
}

static void commit_plane(Av1oFrame *f, int plane, int r, int c, int bs, const uint16_t *rec, const int32_t *qc, const TxRes *tr) {
  const int n = 1 << dim_wl(bs), nh = 1 << dim_hl(bs), cs = imin(n, 32), ch = imin(nh, 32);
  for (int i = 0; i < nh; i++) memcpy(f->rec[plane] + (r * 4 + i) * f->stride + c * 4, rec + i * n, 2 * (size_t)n);
  for (int i = 0; i < ch; i++) memcpy(f->coef[plane] + (r * 4 + i) * f->stride + c * 4, qc + i * cs, 4 * (size_t)cs);
  fill_map2(f->m_lvl[plane], f->mi_stride, r, c, n >> 2, nh >> 2, tr->cul);
  fill_map2(f->m_dc[plane], f->mi_stride, r, c, n >> 2, nh >> 2, tr->dcc);
  f->m_eob[plane][r * f->mi_stride + c] = (uint16_t)tr->eob;
}

/* rdo_cfl_alpha: per plane, the alpha (|a| in 1..16, sign) minimising prediction SSE; 0 = CFL_SIGN_ZERO */
static int cfl_best_alpha(Search *s, int plane, int r, int c, int bs, const uint16_t *dc_pred) {
  Av1oFrame *f = s->f; const int n = 1 << dim_wl(bs), nh = 1 << dim_hl(bs);
  static __thread uint16_t tmp[64 * 64];
  const uint16_t *src = f->src[plane] + r * 4 * f->stride + c * 4;
  int best = 0; int64_t best_sse = sse_block_wh(src, f->stride, dc_pred, n, n, nh);
  for (int mag = 1; mag <= 16; mag++) for (int sg = 0; sg < 2; sg++) {
    int alpha = sg ? -mag : mag;
    memcpy(tmp, dc_pred, 2 * (size_t)(n * nh));
    av1o_predict_cfl_wh(f, plane, c * 4, r * 4, dim_wl(bs), dim_hl(bs), alpha, tmp, n);
    int64_t e = sse_block_wh(src, f->stride, tmp, n, n, nh);
    if (e < best_sse) { best_sse = e; best = alpha; }
  }
  return best;
}

/* Full decision for one block; writes maps/recon/coefs for the area and returns its RD cost. */
static int64_t try_block(Search *s, int r, int c, int bs) {
  Av1oFrame *f = s->f; const TileB *t = &s->t;
  /* bs: a square (BS_4 .. BS_64) or BS_4X8 / BS_8X4 (av1o_int.h); n = width = pitch of the block-sized scratch arrays */
  const int wl = dim_wl(bs), hl = dim_hl(bs), n = 1 << wl, bh = 1 << hl, w4 = n >> 2, h4 = bh >> 2, big = dim_min_l(bs) >= 3;
  const int ms = f->mi_stride, mi = r * ms + c, txs = bs;
  const int x = c * 4, y = r * 4;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int have_ar = availU && (c + w4 < t->mi_col_end) && f->m_decoded[(r - 1) * ms + c + w4];
  const int have_bl = availL && (r + h4 < t->mi_row_end) && f->m_decoded[(r + h4) * ms + c - 1];
  const int amode = availU ? f->m_ymode[mi - ms] : DC_PRED, lmode = availL ? f->m_ymode[mi - 1] : DC_PRED;
  const uint32_t *ycost = f->cost + CDF_KF_Y + (intra_mode_ctx[amode] * 5 + intra_mode_ctx[lmode]) * CDF_KF_Y_STRIDE;
  #define IS_SMOOTH(m) ((m) == SMOOTH_PRED || (m) == SMOOTH_V_PRED || (m) == SMOOTH_H_PRED)
  const int ftype_y = (availU && IS_SMOOTH(f->m_ymode[mi - ms])) || (availL && IS_SMOOTH(f->m_ymode[mi - 1]));
  const int ftype_uv = (availU && IS_SMOOTH(f->m_uvmode[mi - ms])) || (availL && IS_SMOOTH(f->m_uvmode[mi - 1]));
  static __thread uint16_t pred[64 * 64], rec_best[3][64 * 64], rec_tmp[64 * 64];
  static __thread int32_t qc_best[3][32 * 32], qc_tmp[32 * 32];
  const uint16_t *src = f->src[0] + y * f->stride + x;
  /* the block's segment (SegmentationLevel::Simple: from its mean activity scale) selects the quantiser of all its planes */
  const int seg = av1o_block_segment(f, x, y, n, bh);
  for (int p = 0; p < 3; p++) { s->dcq[p] = f->seg_dcq[seg][p]; s->acq[p] = f->seg_acq[seg][p]; }

  /* ---- luma: SATD pre-filter over the 13 modes ---- */
  int64_t satd[13]; int order[13];
  for (int m = 0; m < 13; m++) {
    av1o_predict_intra_wh(f, t, 0, x, y, wl, hl, availL, availU, have_ar, have_bl, m, 0, ftype_y, pred, n);
    satd[m] = satd_block_wh(src, f->stride, pred, n, n, bh);
    order[m] = m;
  }
  for (int i = 1; i < 13; i++) { int v = order[i], j = i; while (j > 0 && satd[order[j - 1]] > satd[v]) { order[j] = order[j - 1]; j--; } order[j] = v; }
  const int ncand = f->cfg.complex_modes ? 7 : 3;
  /* ---- full RD over surviving (mode, angle delta) x tx type ---- */
  int64_t best_j = INT64_MAX, best_mode_j = 0; int best_mode = DC_PRED, best_delta = 0, best_tx = DCT_DCT; TxRes best_tr = { 0, 0, 0, 0, 0 };
  int tx_ns, tx_set;
  int seq_second = 0;
  for (int ci = 0; ci < ncand + ((abl_flags() & ABL_SEQ_TXTYPE) ? 1 : 0); ci++) {
    if (ci == ncand) seq_second = 1;                               /* AV1O_ABL_SEQ_TXTYPE: the extra round = the transform-type search on the winning mode */
    const int m = seq_second ? best_mode : order[ci];
    int delta = 0;
    const int directional = m >= V_PRED && m <= D67_PRED;
    if (directional && big && f->cfg.fine_directional) {
      int64_t bsd = satd[m];
      static const int dl[6] = { -1, 1, -2, 2, -3, 3 };
      for (int k = 0; k < 6; k++) {
        av1o_predict_intra_wh(f, t, 0, x, y, wl, hl, availL, availU, have_ar, have_bl, m, dl[k], ftype_y, pred, n);
        int64_t sd = satd_block_wh(src, f->stride, pred, n, n, bh);
        if (sd < bsd) { bsd = sd; delta = dl[k]; }
      }
    }
    av1o_predict_intra_wh(f, t, 0, x, y, wl, hl, availL, availU, have_ar, have_bl, m, delta, ftype_y, pred, n);
    uint32_t mode_rate = ycost[m];
    if (directional && big) mode_rate += f->cost[CDF_ANGLE + (m - V_PRED) * CDF_ANGLE_STRIDE + delta + 3];
    const int tx_off = av1o_intra_tx_cdf(f, txs, m, &tx_ns, &tx_set);
    const int seq_tx = (abl_flags() & ABL_SEQ_TXTYPE);          /* ablation: pass 0 (ci < ncand) prices the mode with its default type; pass 1 (the winner again) tries the others */
    const int ntx = (f->cfg.rdo_tx && tx_off >= 0 && !(seq_tx && !seq_second)) ? tx_ns : 1;
    for (int ti = 0; ti < ntx; ti++) {
      int txtype;
      if (ntx > 1) txtype = av1o_symbol_to_tx_type(tx_set, ti);
      else { txtype = av1o_mode_to_txtype(m); if (tx_off < 0 || !av1o_tx_type_in_set(tx_set, txtype)) txtype = DCT_DCT; }
      TxRes tr;
      int64_t j = eval_tx(s, 0, r, c, bs, bs, pred, txtype, tx_off, tx_off >= 0 ? av1o_tx_type_to_symbol(tx_set, txtype) : 0, tx_ns, rec_tmp, qc_tmp, &tr);
      j += ((int64_t)mode_rate * f->rdmult[0] + 256) >> 9;
      if (j < best_j) {
        best_mode_j = ((int64_t)mode_rate * f->rdmult[0] + 256) >> 9;
        best_j = j; best_mode = m; best_delta = delta; best_tx = txtype; best_tr = tr;
        memcpy(rec_best[0], rec_tmp, 2 * (size_t)(n * bh)); memcpy(qc_best[0], qc_tmp, 4 * (size_t)imin(n * bh, 1024));
      }
    }
  }
  /* ---- luma transform size (rav1e rdo_tx_size_type; TX_MODE_SELECT, rdo_tx_depth = 2): the largest transform against the
   * transforms one and two levels smaller (tx_depth 1 and 2: 2x2 / 4x4 transform blocks in raster order), same prediction mode,
   * each sub-block predicted from the reconstruction of the ones before it (spec transform_block), one tx type per depth for the
   * whole block (below).  A depth is abandoned as soon as its running cost reaches the best so far; the frame buffers always hold the trial in
   * progress, the best split so far waits in a snapshot of the block's area. */
  int txs_final = bs, any_coef = best_tr.eob > 0;
  if (f->tx_mode_select && bs != BS_4) {
    /* tx_depth context: the above neighbour's transform WIDTH against this block's largest transform width, the left one's HEIGHT
     * against its height (spec get_tx_size_context); category by Max_Tx_Depth: 8x8 and the 2:1 blocks of it share the first */
    const int actx = availU && dim_wl(f->m_txsize[mi - ms]) >= wl, lctx = availL && dim_hl(f->m_txsize[mi - 1]) >= hl;
    const int cat = dim_is_rect(bs) ? 0 : bs - 1;
    const uint32_t *dcost = f->cost + CDF_TX_SIZE + (cat * 3 + actx + lctx) * CDF_TX_SIZE_STRIDE;
    best_j += ((int64_t)dcost[0] * f->rdmult[0] + 256) >> 9;
    if (f->cfg.rdo_tx) {
      static AreaSnap split_snap;
      const int max_depth = (!dim_is_rect(bs) && bs >= BS_16) ? AV1O_TX_DEPTH_MAX : 1;
      for (int d = 1; d <= max_depth; d++) {
        /* sub-transforms: a square splits into 2^d x 2^d squares, a 2:1 block into its two squares (Split_Tx_Size) */
        const int stx = dim_is_rect(bs) ? dim_min_l(bs) - 2 : bs - d, hn = 4 << stx, half = 1 << stx, G = n / hn, GH = bh / hn;
        int64_t j_split = best_mode_j + (((int64_t)dcost[d] * f->rdmult[0] + 256) >> 9);
        int sub_any = 0;
        static __thread uint16_t spred[32 * 32], srec[2][32 * 32]; static __thread int32_t sqc[2][32 * 32];
        int stx_ns, stx_set;
        const int stx_off = av1o_intra_tx_cdf(f, stx, best_mode, &stx_ns, &stx_set);
        const int sntx = stx_off >= 0 ? stx_ns : 1;
        /* One transform type per block and transform size (rav1e rdo_tx_size_type: rdo_tx_type_decision runs over the whole block for each size): the depth is tried once per
         * type with every sub-block forced to it -- a chain of sub-blocks each predicted from that chain's own reconstructions --, a chain is abandoned as soon as its running
         * cost reaches the best so far, the cheapest complete chain (lowest symbol among equals) competes; its trial is repeated last so that the frame holds it. */
        const int one_tx = !(abl_flags() & ABL_SUB_TXTYPE) && sntx > 1;
        int forced = -1, forced_best = 0; int64_t forced_best_j = INT64_MAX;
        const int64_t j_split0 = j_split;
        for (int ft = 0; ft < (one_tx ? sntx + 1 : 1); ft++) {
        if (one_tx) { forced = ft < sntx ? ft : forced_best; j_split = j_split0; sub_any = 0; set_decoded(f, r, c, bs, 0); }
        for (int k = 0; k < G * GH && (j_split < best_j || (one_tx && ft == sntx)); k++) {
          const int bi = k / G, bj_ = k % G;
          const int rr = r + bi * half, cc = c + bj_ * half;
          const int sU = availU || bi, sL = availL || bj_;
          const int s_ar = sU && (cc + half < t->mi_col_end) && f->m_decoded[(rr - 1) * ms + cc + half];
          const int s_bl = sL && (rr + half < t->mi_row_end) && f->m_decoded[(rr + half) * ms + cc - 1];
          av1o_predict_intra(f, t, 0, cc * 4, rr * 4, 2 + stx, sL, sU, s_ar, s_bl, best_mode, best_delta, ftype_y, spred, hn);
          int64_t bj = INT64_MAX; int btx = DCT_DCT, cur = 0; TxRes btr = { 0, 0, 0, 0, 0 };
          for (int ti = 0; ti < sntx; ti++) {
            if (forced >= 0 && ti != forced) continue;
            int txtype;
            if (sntx > 1) txtype = av1o_symbol_to_tx_type(stx_set, ti);
            else { txtype = av1o_mode_to_txtype(best_mode); if (stx_off < 0 || !av1o_tx_type_in_set(stx_set, txtype)) txtype = DCT_DCT; }
            TxRes tr;
            const int64_t j = eval_tx(s, 0, rr, cc, stx, bs, spred, txtype, stx_off, stx_off >= 0 ? av1o_tx_type_to_symbol(stx_set, txtype) : 0, stx_ns, srec[cur], sqc[cur], &tr);
            if (j < bj) { bj = j; btx = txtype; btr = tr; cur ^= 1; }
          }
          commit_plane(f, 0, rr, cc, stx, srec[cur ^ 1], sqc[cur ^ 1], &btr);
          fill_map(f->m_txtype, ms, rr, cc, half, btr.eob ? btx : DCT_DCT);
          set_decoded(f, rr, cc, stx, 1);
          sub_any |= btr.eob > 0;
          j_split += bj;
        }
        if (one_tx && ft < sntx && j_split < forced_best_j) { forced_best_j = j_split; forced_best = ft; }
        }
        set_decoded(f, r, c, bs, 0);
        if (j_split < best_j) {
          best_j = j_split; txs_final = stx; any_coef = sub_any;
          if (d < max_depth) area_copy(f, &split_snap, r, c, bs, 1);       /* a deeper trial is about to overwrite the area */
        } else if (txs_final != bs && txs_final != stx) area_copy(f, &split_snap, r, c, bs, 0);   /* the shallower split stays */
      }
    }
  }
  if (txs_final == bs) {
    commit_plane(f, 0, r, c, bs, rec_best[0], qc_best[0], &best_tr);
    fill_map2(f->m_txtype, ms, r, c, w4, h4, best_tr.eob ? best_tx : DCT_DCT);
  }
  fill_map2(f->m_txsize, ms, r, c, w4, h4, txs_final);
  fill_map2(f->m_ymode, ms, r, c, w4, h4, best_mode);
  fill_map2((uint8_t *)f->m_angle_y, ms, r, c, w4, h4, (uint8_t)(int8_t)best_delta);
  fill_map2(f->m_bsize, ms, r, c, w4, h4, bs);
  int64_t total_j = best_j;

  /* ---- chroma ---- */
  if (f->np > 1) {
    const int cfl_allowed = dim_max_l(bs) <= 5;
    const uint32_t *uvcost = cfl_allowed ? f->cost + CDF_UV_CFL + best_mode * CDF_UV_CFL_STRIDE : f->cost + CDF_UV_NOCFL + best_mode * CDF_UV_NOCFL_STRIDE;
    int cands[16], nc = 0;
    cands[nc++] = DC_PRED;
    if (best_mode != DC_PRED) cands[nc++] = best_mode;
    if (f->cfg.complex_modes) for (int m = 1; m < 13; m++) if (m != best_mode) cands[nc++] = m;
    if (cfl_allowed) cands[nc++] = UV_CFL_PRED;
    int64_t best_uv = INT64_MAX; int buv = DC_PRED, bdelta = 0, bsign = 0, bau = 0, bav = 0; TxRes btr[3];
    /* (per-thread scratch: the search of different tiles may run on different threads) */
    static __thread uint16_t rec_c[3][64 * 64]; static __thread int32_t qc_c[3][32 * 32];
    static __thread ChromaSnap64 uv_snap64; int uv_any64 = 0, uv_in_frame = 0;
    for (int ci = 0; ci < nc; ci++) {
      const int um = cands[ci];
      int delta = (um == best_mode && um >= V_PRED && um <= D67_PRED && big) ? best_delta : 0;
      int alpha[3] = { 0, 0, 0 }, jsign = 0;
      uint32_t mode_rate = uvcost[um];
      if (um >= V_PRED && um <= D67_PRED && big) mode_rate += f->cost[CDF_ANGLE + (um - V_PRED) * CDF_ANGLE_STRIDE + delta + 3];
      int txtype = av1o_mode_to_txtype(um);
      if (!av1o_tx_type_in_set(av1o_tx_set(txs, f->cfg.reduced_tx_set), txtype)) txtype = DCT_DCT;
      int64_t j = 0; TxRes trs[3]; int ok = 1;
      for (int p = 1; p < 3 && ok; p++) {
        if (um == UV_CFL_PRED) {
          av1o_predict_intra_wh(f, t, p, x, y, wl, hl, availL, availU, have_ar, have_bl, DC_PRED, 0, ftype_uv, pred, n);
          alpha[p] = cfl_best_alpha(s, p, r, c, bs, pred);
          if (p == 2) {
            if (alpha[1] == 0 && alpha[2] == 0) { ok = 0; break; }
            int su = alpha[1] == 0 ? 0 : (alpha[1] < 0 ? 1 : 2), sv = alpha[2] == 0 ? 0 : (alpha[2] < 0 ? 1 : 2);
            jsign = su * 3 + sv - 1;
            mode_rate += f->cost[CDF_CFL_SIGN + jsign];
            if (su) mode_rate += f->cost[CDF_CFL_ALPHA + ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE + iabs(alpha[1]) - 1];
            if (sv) mode_rate += f->cost[CDF_CFL_ALPHA + ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE + iabs(alpha[2]) - 1];
          }
        }
      }
      if (!ok) continue;
      if (bs == BS_64) {
        /* a 64x64 block of a 4:4:4 frame carries four 32x32 chroma transform blocks per plane (spec get_tx_size: chroma transforms stop at
         * 32x32), each predicted from the reconstruction of the ones before it (spec transform_block) in raster order, plane after plane
         * (spec residual).  The candidate in progress lives in the frame; the best one so far waits in a snapshot of the two chroma planes. */
        int c_any = 0;
        for (int p = 1; p < 3; p++) {
          for (int k = 0; k < 4; k++) {
            const int rr = r + (k >> 1) * 8, cc = c + (k & 1) * 8;
            if (rr >= f->mi_rows || cc >= f->mi_cols) continue;           /* (a 64x64 block never straddles the frame edge: must_split) */
            const int sU = availU || (k >> 1), sL = availL || (k & 1);
            const int s_ar = sU && (cc + 8 < t->mi_col_end) && f->m_decoded[(rr - 1) * ms + cc + 8];
            const int s_bl = sL && (rr + 8 < t->mi_row_end) && f->m_decoded[(rr + 8) * ms + cc - 1];
            av1o_predict_intra(f, t, p, cc * 4, rr * 4, 5, sL, sU, s_ar, s_bl, um, delta, ftype_uv, pred, 32);
            TxRes tr;
            j += eval_tx(s, p, rr, cc, TX_32X32, bs, pred, DCT_DCT, -1, 0, 0, rec_tmp, qc_tmp, &tr);
            commit_plane(f, p, rr, cc, TX_32X32, rec_tmp, qc_tmp, &tr);
            set_decoded(f, rr, cc, BS_32, 1);
            c_any |= tr.eob > 0;
          }
          set_decoded(f, r, c, bs, 0);
        }
        j += ((int64_t)mode_rate * f->rdmult[0] + 256) >> 9;
        if (j < best_uv) {
          best_uv = j; buv = um; bdelta = delta; uv_any64 = c_any;
          if (ci + 1 < nc) chroma_copy64(f, &uv_snap64, r, c, 1);
          uv_in_frame = 1;
        } else uv_in_frame = 0;
        continue;
      }
      for (int p = 1; p < 3; p++) {
        if (um == UV_CFL_PRED) {
          av1o_predict_intra_wh(f, t, p, x, y, wl, hl, availL, availU, have_ar, have_bl, DC_PRED, 0, ftype_uv, pred, n);
          if (alpha[p]) av1o_predict_cfl_wh(f, p, x, y, wl, hl, alpha[p], pred, n);
        } else {
          av1o_predict_intra_wh(f, t, p, x, y, wl, hl, availL, availU, have_ar, have_bl, um, delta, ftype_uv, pred, n);
        }
        j += eval_tx(s, p, r, c, bs, bs, pred, txtype, -1, 0, 0, rec_best[p], qc_best[p], &trs[p]);
      }
      j += ((int64_t)mode_rate * f->rdmult[0] + 256) >> 9;
      if (j < best_uv) {
        best_uv = j; buv = um; bdelta = delta; bsign = jsign; bau = alpha[1]; bav = alpha[2]; btr[1] = trs[1]; btr[2] = trs[2];
        for (int p = 1; p < 3; p++) { memcpy(rec_c[p], rec_best[p], 2 * (size_t)(n * bh)); memcpy(qc_c[p], qc_best[p], 4 * (size_t)imin(n * bh, 1024)); }
      }
    }
    if (bs == BS_64) { if (!uv_in_frame) chroma_copy64(f, &uv_snap64, r, c, 0); any_coef |= uv_any64; }
    else for (int p = 1; p < 3; p++) { commit_plane(f, p, r, c, bs, rec_c[p], qc_c[p], &btr[p]); any_coef |= btr[p].eob > 0; }
    fill_map2(f->m_uvmode, ms, r, c, w4, h4, buv);
    fill_map2((uint8_t *)f->m_angle_uv, ms, r, c, w4, h4, (uint8_t)(int8_t)bdelta);
    fill_map2(f->m_cfl_sign, ms, r, c, w4, h4, bsign);
    fill_map2(f->m_cfl_au, ms, r, c, w4, h4, bau ? iabs(bau) - 1 : 0);
    fill_map2(f->m_cfl_av, ms, r, c, w4, h4, bav ? iabs(bav) - 1 : 0);
    total_j += best_uv;
  }
  /* ---- skip flag ---- */
  const int skip = !any_coef;
  fill_map2(f->m_skip, ms, r, c, w4, h4, skip);
  if (skip) for (int p = 0; p < f->np; p++) { fill_map2(f->m_lvl[p], ms, r, c, w4, h4, 0); fill_map2(f->m_dc[p], ms, r, c, w4, h4, 0); }
  const int sctx = (availU ? f->m_skip[mi - ms] : 0) + (availL ? f->m_skip[mi - 1] : 0);
  total_j += ((int64_t)f->cost[CDF_SKIP + sctx * CDF_SKIP_STRIDE + skip] * f->rdmult[0] + 256) >> 9;
  if (f->seg_n) {
    /* intra_segment_id after the skip flag (no pre-skip feature): a skipped block takes the predicted id, any other codes its own */
    int ctx;
    const int pred = av1o_seg_pred(availU && availL ? f->m_seg[mi - ms - 1] : -1, availU ? f->m_seg[mi - ms] : -1, availL ? f->m_seg[mi - 1] : -1, &ctx);
    fill_map2(f->m_seg, ms, r, c, w4, h4, skip ? pred : seg);
    if (!skip) total_j += ((int64_t)f->cost[CDF_SEG_ID + ctx * CDF_SEG_ID_STRIDE + av1o_seg_symbol(seg, pred, f->seg_n)] * f->rdmult[0] + 256) >> 9;
  }
  set_decoded(f, r, c, bs, 1);
  return total_j;
}

static uint32_t partition_rate(Search *s, int r, int c, int bs, int part) {
  Av1oFrame *f = s->f; const TileB *t = &s->t; const int ms = f->mi_stride;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  /* partition context: is the above block narrower / the left block lower than this node (spec 8.3.2, AboveSegPredContext-style maps) */
  const int above = availU && dim_wl(f->m_bsize[(r - 1) * ms + c]) < 2 + bs, left = availL && dim_hl(f->m_bsize[r * ms + c - 1]) < 2 + bs;
  return f->cost[CDF_PARTITION + ((bs - 1) * 4 + left * 2 + above) * CDF_PARTITION_STRIDE + part];
}

/* encode_partition_topdown.
 * `known_j` >= 0: the parent's split trial has just evaluated this block undivided, nothing it depends on has
 * changed since (every earlier sibling kept PARTITION_NONE), and the frame buffers still hold that result --
 * re-running try_block() would reproduce it bit for bit, so its cost is taken from the trial instead.  (rav1e
 * caches the trial's mode decisions for the children the same way.)  Returns 1 when the block was split. */
static int rd_partition(Search *s, int r, int c, int bs, int64_t known_j) {
  Av1oFrame *f = s->f;
  if (r >= f->mi_rows || c >= f->mi_cols) return 0;
  const int half = (1 << bs) >> 1, px = 4 << bs;
  const int has_rows = (r + half) < f->mi_rows, has_cols = (c + half) < f->mi_cols;
  const int must_split = bs > BS_4 && (px > f->cfg.part_max || !has_rows || !has_cols);
  const int can_split = bs > BS_4 && (px > f->cfg.part_min || must_split);
  if (known_j < 0 || must_split) set_decoded(f, r, c, bs, 0);
  if (!can_split) {
    if (known_j < 0) try_block(s, r, c, bs); else set_decoded(f, r, c, bs, 1);
    return 0;
  }
  int do_split = must_split;
  int64_t sub_j[4] = { -1, -1, -1, -1 };
  if (!must_split) {
    static __thread AreaSnap snap[5];
    const int64_t j_blk = known_j >= 0 ? known_j : try_block(s, r, c, bs);
    int64_t j_none = j_blk + (((int64_t)partition_rate(s, r, c, bs, PARTITION_NONE) * f->rdmult[0] + 256) >> 9);
    area_copy(f, &snap[bs], r, c, bs, 1);
    set_decoded(f, r, c, bs, 0);
    int64_t j_split = ((int64_t)partition_rate(s, r, c, bs, PARTITION_SPLIT) * f->rdmult[0] + 256) >> 9;
    for (int k = 0; k < 4 && j_split < j_none; k++) {
      int rr = r + (k >> 1) * half, cc = c + (k & 1) * half;
      if (rr >= f->mi_rows || cc >= f->mi_cols) continue;
      sub_j[k] = try_block(s, rr, cc, bs - 1);
      j_split += sub_j[k];
      if (bs - 1 >= BS_8) j_split += ((int64_t)partition_rate(s, rr, cc, bs - 1, PARTITION_NONE) * f->rdmult[0] + 256) >> 9;
    }
    /* PARTITION_HORZ / PARTITION_VERT of an 8x8 node (two 8x4 / 4x8 blocks), tried against the best of NONE / SPLIT so far */
    int rect_won = 0;
    if (bs == BS_8) {
      static AreaSnap best_snap, split_snap;
      int64_t j_best = j_none; int have_split = 0;
      if (j_split < j_none) { j_best = j_split; area_copy(f, &split_snap, r, c, bs, 1); have_split = 1; }
      for (int part = PARTITION_HORZ; part <= PARTITION_VERT; part++) {
        const int rb = part == PARTITION_HORZ ? BS_8X4 : BS_4X8;
        set_decoded(f, r, c, bs, 0);
        int64_t j = ((int64_t)partition_rate(s, r, c, bs, part) * f->rdmult[0] + 256) >> 9;
        for (int k = 0; k < 2 && j < j_best; k++) j += try_block(s, r + (part == PARTITION_HORZ ? k : 0), c + (part == PARTITION_VERT ? k : 0), rb);
        if (j < j_best) { j_best = j; rect_won = part; area_copy(f, &best_snap, r, c, bs, 1); }
      }
      if (rect_won) { area_copy(f, &best_snap, r, c, bs, 0); set_decoded(f, r, c, bs, 1); return 1; }   /* 1: the later siblings' trial results are stale */
      if (have_split) area_copy(f, &split_snap, r, c, bs, 0);      /* the 4x4 trial results are back in place for the chain below */
    }
    if (j_split < j_none) do_split = 1;
    else { area_copy(f, &snap[bs], r, c, bs, 0); set_decoded(f, r, c, bs, 1); }
  }
  if (do_split) {
    set_decoded(f, r, c, bs, 0);
    int chain = !must_split;      /* the four trial results are in place until a sibling decides to split */
    for (int k = 0; k < 4; k++) {
      const int rr = r + (k >> 1) * half, cc = c + (k & 1) * half;
      if (rd_partition(s, rr, cc, bs - 1, chain ? sub_j[k] : -1)) chain = 0;
    }
    return 1;
  }
  return 0;
}

/* encode_partition_bottomup (rav1e; SpeedTweaks.encode_bottomup, speed <= 2, ravif av1encoder.rs:575): every child is searched
 * recursively FIRST, so the undivided block competes with the children's own best partitions instead of their undivided forms.
 * Returns the RD cost of the node including its partition symbol; the winner's data is left in the frame. */
static int64_t rd_partition_bottomup(Search *s, int r, int c, int bs) {
  Av1oFrame *f = s->f;
  if (r >= f->mi_rows || c >= f->mi_cols) return 0;
  const int half = (1 << bs) >> 1, px = 4 << bs;
  const int has_rows = (r + half) < f->mi_rows, has_cols = (c + half) < f->mi_cols;
  const int must_split = bs > BS_4 && (px > f->cfg.part_max || !has_rows || !has_cols);
  const int can_split = bs > BS_4 && (px > f->cfg.part_min || must_split);
  static __thread AreaSnap snap[5];
  set_decoded(f, r, c, bs, 0);
  int64_t j_none = INT64_MAX;
  if (!must_split) {
    j_none = try_block(s, r, c, bs);
    if (bs >= BS_8) j_none += ((int64_t)partition_rate(s, r, c, bs, PARTITION_NONE) * f->rdmult[0] + 256) >> 9;
    if (!can_split) return j_none;
    area_copy(f, &snap[bs], r, c, bs, 1);
    set_decoded(f, r, c, bs, 0);
  }
  int64_t j_split = must_split ? 0 : ((int64_t)partition_rate(s, r, c, bs, PARTITION_SPLIT) * f->rdmult[0] + 256) >> 9;
  for (int k = 0; k < 4; k++) {
    if (!must_split && j_split >= j_none) break;               /* costs only grow */
    j_split += rd_partition_bottomup(s, r + (k >> 1) * half, c + (k & 1) * half, bs - 1);
  }
  if (bs == BS_8 && !must_split) {
    static AreaSnap best_snap, split_snap;
    int64_t j_best = j_none; int rect_won = 0, have_split = 0;
    if (j_split < j_none) { j_best = j_split; area_copy(f, &split_snap, r, c, bs, 1); have_split = 1; }
    for (int part = PARTITION_HORZ; part <= PARTITION_VERT; part++) {
      const int rb = part == PARTITION_HORZ ? BS_8X4 : BS_4X8;
      set_decoded(f, r, c, bs, 0);
      int64_t j = ((int64_t)partition_rate(s, r, c, bs, part) * f->rdmult[0] + 256) >> 9;
      for (int k = 0; k < 2 && j < j_best; k++) j += try_block(s, r + (part == PARTITION_HORZ ? k : 0), c + (part == PARTITION_VERT ? k : 0), rb);
      if (j < j_best) { j_best = j; rect_won = part; area_copy(f, &best_snap, r, c, bs, 1); }
    }
    if (rect_won) { area_copy(f, &best_snap, r, c, bs, 0); set_decoded(f, r, c, bs, 1); return j_best; }
    if (have_split) { area_copy(f, &split_snap, r, c, bs, 0); set_decoded(f, r, c, bs, 1); return j_split; }
  }
  if (must_split || j_split < j_none) return j_split;
  area_copy(f, &snap[bs], r, c, bs, 0);
  set_decoded(f, r, c, bs, 1);
  return j_none;
}

void av1o_search_tile(Av1oFrame *f, int tile_row, int tile_col) {
  Search s; s.f = f;
  f->cost = f->tile_cost ? f->tile_cost + (size_t)(tile_row * f->tile_cols + tile_col) * CDF_TOTAL : f->cost0;
  s.t.mi_row_start = f->tile_row_start[tile_row] * SB_MI; s.t.mi_row_end = imin(f->tile_row_start[tile_row + 1] * SB_MI, f->mi_rows);
  s.t.mi_col_start = f->tile_col_start[tile_col] * SB_MI; s.t.mi_col_end = imin(f->tile_col_start[tile_col + 1] * SB_MI, f->mi_cols);
  for (int p = 0; p < f->np; p++) {
    /* (q_y / q_p)^2 in Q12 == rav1e dist_scale */
    s.wq[p] = (((int64_t)f->ac_q[0] * f->ac_q[0]) << 12) / ((int64_t)f->ac_q[p] * f->ac_q[p]);
  }
  /* AV1O_LIVE_CDF=1: the experiment of av1o_entropy.c av1o_live_* (rates from the tile's adaptive CDFs, refreshed after every superblock) */
  void *live = getenv("AV1O_LIVE_CDF") ? av1o_live_open(f, tile_row, tile_col) : NULL;
  static __thread uint32_t live_cost[CDF_TOTAL];
  if (live) { memcpy(live_cost, f->cost0, sizeof(live_cost)); f->cost = live_cost; }
  for (int r = s.t.mi_row_start; r < s.t.mi_row_end; r += SB_MI)
    for (int c = s.t.mi_col_start; c < s.t.mi_col_end; c += SB_MI) {
      if (f->cfg.bottomup) rd_partition_bottomup(&s, r, c, BS_64); else rd_partition(&s, r, c, BS_64, -1);
      if (live) av1o_live_sb(live, r, c, live_cost);
    }
  if (live) { av1o_live_close(live); f->cost = f->cost0; }     /* the experiment's table is this call's: later stages price against the frame's own again */
}

/* oracle/av1o_entropy.c -- coefficient syntax (spec 5.11.39 coeffs(), contexts 8.3.2) shared by the
 * rate estimator and the tile bitstream writer, and phase 2: the tile writer itself (spec 5.11.1-5.11.36:
 * decode_partition / intra_frame_mode_info / residual), driven by the mode-info maps phase 1 left behind.
 * TEST INFRASTRUCTURE (see av1o.h).  rav1e equivalents: src/context/*.rs, src/ec.rs (absent). */
#include <stdio.h>
#include "av1o_int.h"
#include "av1o_syms.h"

/* ---------------- txb contexts (spec: all_zero ctx and dc_sign ctx derivation) ---------------- */
void av1o_txb_ctx(const Av1oFrame *f, const TileB *t, int plane, int r4, int c4, int txs, int bs, int *skip_ctx, int *dc_ctx) {
  const int w4 = 1 << (dim_wl(txs) - 2), h4 = 1 << (dim_hl(txs) - 2), ms = f->mi_stride;
  int top = 0, left = 0, dcs = 0, any_a = 0, any_l = 0;
  if (r4 - 1 >= t->mi_row_start) {
    for (int k = 0; k < w4; k++) if (c4 + k < f->mi_cols) {
      int l = f->m_lvl[plane][(r4 - 1) * ms + c4 + k], d = f->m_dc[plane][(r4 - 1) * ms + c4 + k];
      top = imax(top, l); any_a |= l | d; dcs += d == 1 ? -1 : (d == 2 ? 1 : 0);
    }
  }
  if (c4 - 1 >= t->mi_col_start) {
    for (int k = 0; k < h4; k++) if (r4 + k < f->mi_rows) {
      int l = f->m_lvl[plane][(r4 + k) * ms + c4 - 1], d = f->m_dc[plane][(r4 + k) * ms + c4 - 1];
      left = imax(left, l); any_l |= l | d; dcs += d == 1 ? -1 : (d == 2 ? 1 : 0);
    }
  }
  *dc_ctx = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
  if (plane == 0) {
    int ctx;
    if (bs == txs) ctx = 0;
    else if (top == 0 && left == 0) ctx = 1;
    else if (top == 0 || left == 0) ctx = 2 + (imax(top, left) > 3);
    else if (imax(top, left) <= 3) ctx = 4;
    else if (imin(top, left) <= 3) ctx = 5;
    else ctx = 6;
    *skip_ctx = ctx;
  } else {
    *skip_ctx = 7 + (any_a != 0) + (any_l != 0) + (dim_wl(bs) + dim_hl(bs) > dim_wl(txs) + dim_hl(txs) ? 3 : 0);
  }
}

/* ---------------- coeffs() ---------------- */
void av1o_code_coeffs(const Av1oFrame *f, const int32_t *qc, int eob, int plane, int txs, int txtype,
                      int skip_ctx, int dc_ctx, int tx_cdf_off, int tx_sym, int tx_nsyms,
                      const SymSink *k, int *cul_level, int *dc_cat) {
  (void)f;
  const int bwl = imin(5, dim_wl(txs)), bhl = imin(5, dim_hl(txs)), n = 1 << bwl, nh = 1 << bhl;   /* coded area n wide, nh high */
  const int txs_ctx = (dim_min_l(txs) - 2 + dim_max_l(txs) - 2 + 1) >> 1;                            /* (sqr + sqr_up + 1) >> 1 */
  const int pt = plane > 0;
  const int cls = av1o_tx_class(txtype);
  *cul_level = 0; *dc_cat = 0;
  k->sym(k->u, CDF_TXB_SKIP + (txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE, eob == 0, 2);
  if (eob == 0) return;
  if (tx_cdf_off >= 0) k->sym(k->u, tx_cdf_off, tx_sym, tx_nsyms);
  uint16_t tmp[1024];
  const uint16_t *scan = av1o_scan(txs, txtype, tmp);
  /* eob_pt */
  int eob_pt, eob_extra_bits = 0;
  if (eob < 3) eob_pt = eob; else { eob_pt = 32 - __builtin_clz((uint32_t)(eob - 1)) + 1; }
  /* eob_pt: eob=1 ->1, 2 ->2, 3..4 ->3, 5..8 ->4, ... ; base = (1 << (eob_pt-2)) + 1 */
  const int eob_multi = bwl + bhl - 4;             /* 0..6 */
  static const int pt_off[7] = { CDF_EOB_PT_16, CDF_EOB_PT_32, CDF_EOB_PT_64, CDF_EOB_PT_128, CDF_EOB_PT_256, CDF_EOB_PT_512, CDF_EOB_PT_1024 };
  static const int pt_str[7] = { CDF_EOB_PT_16_STRIDE, CDF_EOB_PT_32_STRIDE, CDF_EOB_PT_64_STRIDE, CDF_EOB_PT_128_STRIDE,
                                 CDF_EOB_PT_256_STRIDE, CDF_EOB_PT_512_STRIDE, CDF_EOB_PT_1024_STRIDE };
  k->sym(k->u, pt_off[eob_multi] + (pt * 2 + (cls == TX_CLASS_2D ? 0 : 1)) * pt_str[eob_multi], eob_pt - 1, 5 + eob_multi);
  if (eob_pt >= 3) {
    eob_extra_bits = eob_pt - 2;
    int rem = eob - ((1 << (eob_pt - 2)) + 1);
    int hi = (rem >> (eob_extra_bits - 1)) & 1;
    k->sym(k->u, CDF_EOB_EXTRA + ((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE, hi, 2);
    if (eob_extra_bits > 1) k->lit(k->u, (uint32_t)rem & ((1u << (eob_extra_bits - 1)) - 1), eob_extra_bits - 1);
  }
  /* level map with a 4-wide zero border to the right/bottom */
  const int st = n + 4;
  static __thread uint8_t lev[(32 + 4) * (32 + 4)];
  memset(lev, 0, (size_t)st * (size_t)(nh + 4));
  for (int i = 0; i < eob; i++) { int p = scan[i]; int a = iabs(qc[p]); lev[(p >> bwl) * st + (p & (n - 1))] = (uint8_t)imin(a, 127); }
  const int area = n * nh;
  for (int c = eob - 1; c >= 0; c--) {
    const int p = scan[c], row = p >> bwl, col = p & (n - 1);
    const int level = iabs(qc[p]);
    const uint8_t *L = lev + row * st + col;
    if (c == eob - 1) {
      int ctx = c == 0 ? 0 : (c <= area / 8 ? 1 : (c <= area / 4 ? 2 : 3));
      k->sym(k->u, CDF_COEFF_BASE_EOB + ((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE, imin(level, 3) - 1, 3);
    } else {
      int mag = imin(L[1], 3) + imin(L[st], 3), ctx;
      if (cls == TX_CLASS_2D) {
        mag += imin(L[st + 1], 3) + imin(L[2], 3) + imin(L[2 * st], 3);
        int m = imin((mag + 1) >> 1, 4);
        if (row == 0 && col == 0) ctx = 0;
        else if (bhl > bwl) ctx = m + (row < 2 ? 11 : (row + col < 4 ? 6 : 21));      /* spec Coeff_Base_Ctx_Offset, tall transforms */
        else if (bwl > bhl) ctx = m + (col < 2 ? 16 : (row + col < 4 ? 6 : 21));      /* wide transforms */
        else if (row + col < 2) ctx = m + 1;
        else if (row + col < 4) ctx = m + 6;
        else ctx = m + 21;
      } else if (cls == TX_CLASS_VERT) {
        mag += imin(L[2 * st], 3) + imin(L[3 * st], 3) + imin(L[4 * st], 3);
        int m = imin((mag + 1) >> 1, 4);
        ctx = m + (row == 0 ? 26 : (row == 1 ? 31 : 36));
      } else {
        mag += imin(L[2], 3) + imin(L[3], 3) + imin(L[4], 3);
        int m = imin((mag + 1) >> 1, 4);
        ctx = m + (col == 0 ? 26 : (col == 1 ? 31 : 36));
      }
      k->sym(k->u, CDF_COEFF_BASE + ((txs_ctx * 2 + pt) * 42 + ctx) * CDF_COEFF_BASE_STRIDE, imin(level, 3), 4);
    }
    if (level > 2) {
      int mag = imin(L[1], 15) + imin(L[st], 15), ctx;
      if (cls == TX_CLASS_2D) {
        mag += imin(L[st + 1], 15); mag = imin((mag + 1) >> 1, 6);
        ctx = c == 0 ? mag : ((row < 2 && col < 2) ? mag + 7 : mag + 14);
      } else if (cls == TX_CLASS_HORIZ) {
        mag += imin(L[2], 15); mag = imin((mag + 1) >> 1, 6);
        ctx = c == 0 ? mag : (col == 0 ? mag + 7 : mag + 14);
      } else {
        mag += imin(L[2 * st], 15); mag = imin((mag + 1) >> 1, 6);
        ctx = c == 0 ? mag : (row == 0 ? mag + 7 : mag + 14);
      }
      const int off = CDF_COEFF_BR + ((imin(txs_ctx, 3) * 2 + pt) * 21 + ctx) * CDF_COEFF_BR_STRIDE;
      int rem = level - 3;
      for (int idx = 0; idx < 4; idx++) {
        int s = imin(rem, 3);
        k->sym(k->u, off, s, 4);
        rem -= s;
        if (s < 3) break;
      }
    }
  }
  int cul = 0;
  for (int c = 0; c < eob; c++) {
    const int p = scan[c], v = qc[p], a = iabs(v);
    if (a) {
      if (c == 0) { k->sym(k->u, CDF_DC_SIGN + (pt * 3 + dc_ctx) * CDF_DC_SIGN_STRIDE, v < 0, 2); *dc_cat = v < 0 ? 1 : 2; }
      else k->lit(k->u, v < 0, 1);
      if (a > 14) {
        uint32_t x = (uint32_t)(a - 14);
        int len = 32 - __builtin_clz(x);
        k->lit(k->u, 0, len - 1);
        k->lit(k->u, x, len);
      }
    }
    cul += a; if (cul > 63) cul = 63;
  }
  *cul_level = cul;
}

/* ---------------- rate sink ---------------- */
typedef struct { const uint32_t *cost; uint64_t bits; } RateU;
static void rate_sym(void *u, int off, int s, int n) { (void)n; RateU *r = (RateU *)u; r->bits += r->cost[off + s]; }
static void rate_lit(void *u, uint32_t v, int nb) { (void)v; ((RateU *)u)->bits += 512u * (uint32_t)nb; }
uint32_t av1o_coef_rate_full(const Av1oFrame *f, const int32_t *qc, int eob, int plane, int txs, int txtype,
                             int skip_ctx, int dc_ctx, int tx_cdf_off, int tx_sym, int tx_nsyms, int *cul, int *dccat) {
  RateU r = { f->cost, 0 };
  SymSink k = { rate_sym, rate_lit, &r };
  av1o_code_coeffs(f, qc, eob, plane, txs, txtype, skip_ctx, dc_ctx, tx_cdf_off, tx_sym, tx_nsyms, &k, cul, dccat);
  return (uint32_t)r.bits;
}

/* ---------------- phase 2: tile writer ---------------- */
extern const uint16_t *av1o_hist_base; extern unsigned *av1o_hist;
typedef struct { RangeEnc ec; uint16_t cdf[CDF_TOTAL]; Av1oFrame *f; TileB t; uint8_t *cdef_done; uint16_t lr_cdf[4]; int lr_ref[3][2]; } TileW;
static void lr_sym(void *u, int s, int n) { TileW *w = (TileW *)u; re_symbol(&w->ec, s, w->lr_cdf, n); }
static void ec_sym(void *u, int off, int s, int n) { TileW *w = (TileW *)u; re_symbol(&w->ec, s, w->cdf + off, n); }
static void ec_lit(void *u, uint32_t v, int nb) { TileW *w = (TileW *)u; re_literal(&w->ec, v, nb); }

static const uint8_t intra_mode_ctx[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 };

int av1o_intra_tx_cdf(const Av1oFrame *f, int txs, int ymode, int *nsyms, int *set_out) {
  int set = av1o_tx_set(txs, f->cfg.reduced_tx_set);
  *set_out = set;
  if (set == 0 || f->base_q_idx == 0) { *nsyms = 0; return -1; }
  const int sq = dim_min_l(txs) - 2;                /* the CDFs are indexed by the square size of the smaller dimension */
  if (set == 1) { *nsyms = 7; return CDF_INTRA_TX1 + (sq * 13 + ymode) * CDF_INTRA_TX1_STRIDE; }
  *nsyms = 5; return CDF_INTRA_TX2 + (sq * 13 + ymode) * CDF_INTRA_TX2_STRIDE;
}

static void write_block(TileW *w, int r, int c, int bs) {
  Av1oFrame *f = w->f; const TileB *t = &w->t; const int ms = f->mi_stride, mi = r * ms + c;
  const SymSink k = { ec_sym, ec_lit, w };
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int skip = f->m_skip[mi];
  /* intra_frame_mode_info: skip, segment id, cdef, (delta q off), y mode, angle, uv mode, cfl, angle uv */
  int sctx = (availU ? f->m_skip[mi - ms] : 0) + (availL ? f->m_skip[mi - 1] : 0);
  re_symbol(&w->ec, skip, w->cdf + CDF_SKIP + sctx * CDF_SKIP_STRIDE, 2);
  if (f->seg_n && !skip) {                          /* intra_segment_id (SegIdPreSkip = 0): after the skip flag; a skipped block's id is the prediction */
    int ctx;
    const int pred = av1o_seg_pred(availU && availL ? f->m_seg[mi - ms - 1] : -1, availU ? f->m_seg[mi - ms] : -1, availL ? f->m_seg[mi - 1] : -1, &ctx);
    re_symbol(&w->ec, av1o_seg_symbol(f->m_seg[mi], pred, f->seg_n), w->cdf + CDF_SEG_ID + ctx * CDF_SEG_ID_STRIDE, 8);
  }
  if (!skip && f->enable_cdef) {
    int sbi = (r >> 4) * f->sb_cols + (c >> 4);
    if (!w->cdef_done[sbi]) {                       /* first non-skip block of this 64x64: read_cdef() */
      w->cdef_done[sbi] = 1;
      re_literal(&w->ec, (uint32_t)f->cdef_idx[sbi], f->cdef_bits);
    }
  }
  const int ymode = f->m_ymode[mi];
  const int am = intra_mode_ctx[availU ? f->m_ymode[mi - ms] : DC_PRED], lm = intra_mode_ctx[availL ? f->m_ymode[mi - 1] : DC_PRED];
  re_symbol(&w->ec, ymode, w->cdf + CDF_KF_Y + (am * 5 + lm) * CDF_KF_Y_STRIDE, 13);
  const int big = dim_min_l(bs) >= 3;                /* angle deltas: blocks of at least 8x8 */
  if (big && ymode >= V_PRED && ymode <= D67_PRED)
    re_symbol(&w->ec, f->m_angle_y[mi] + 3, w->cdf + CDF_ANGLE + (ymode - V_PRED) * CDF_ANGLE_STRIDE, 7);
  int uvmode = 0;
  if (f->np > 1) {
    uvmode = f->m_uvmode[mi];
    if (dim_max_l(bs) <= 5) re_symbol(&w->ec, uvmode, w->cdf + CDF_UV_CFL + ymode * CDF_UV_CFL_STRIDE, 14);
    else re_symbol(&w->ec, uvmode, w->cdf + CDF_UV_NOCFL + ymode * CDF_UV_NOCFL_STRIDE, 13);
    if (uvmode == UV_CFL_PRED) {
      int js = f->m_cfl_sign[mi], su = (js + 1) / 3, sv = (js + 1) % 3;
      re_symbol(&w->ec, js, w->cdf + CDF_CFL_SIGN, 8);
      if (su) re_symbol(&w->ec, f->m_cfl_au[mi], w->cdf + CDF_CFL_ALPHA + ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE, 16);
      if (sv) re_symbol(&w->ec, f->m_cfl_av[mi], w->cdf + CDF_CFL_ALPHA + ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE, 16);
    }
    if (big && uvmode >= V_PRED && uvmode <= D67_PRED)
      re_symbol(&w->ec, f->m_angle_uv[mi] + 3, w->cdf + CDF_ANGLE + (uvmode - V_PRED) * CDF_ANGLE_STRIDE, 7);
  }
  /* read_block_tx_size(): tx_depth for every intra block above 4x4 under TX_MODE_SELECT, coded even when skip */
  const int txs_y = f->m_txsize[mi];
  if (f->tx_mode_select && bs != BS_4) {
    const int actx = availU && dim_wl(f->m_txsize[mi - ms]) >= dim_wl(bs), lctx = availL && dim_hl(f->m_txsize[mi - 1]) >= dim_hl(bs);
    const int cat = dim_is_rect(bs) ? 0 : bs - 1;
    const int depth = txs_y == bs ? 0 : (dim_is_rect(bs) ? 1 : bs - txs_y);
    re_symbol(&w->ec, depth, w->cdf + CDF_TX_SIZE + (cat * 3 + actx + lctx) * CDF_TX_SIZE_STRIDE, cat == 0 ? 2 : 3);
  }
  if (skip) return;
  /* residual(): per plane the transform blocks of the block in raster order (luma may be split one level, chroma is not) */
  static __thread int32_t qc[1024];
  for (int p = 0; p < f->np; p++) {
    const int txs = p == 0 ? txs_y : (bs == BS_64 ? TX_32X32 : bs) /* chroma transforms stop at 32x32 (spec get_tx_size) */, n = imin(32, 1 << dim_wl(txs)), nh = imin(32, 1 << dim_hl(txs));
    const int stepw = 1 << (dim_wl(txs) - 2), steph = 1 << (dim_hl(txs) - 2), nbw = 1 << (dim_wl(bs) - dim_wl(txs)), nbh = 1 << (dim_hl(bs) - dim_hl(txs));
    for (int by = 0; by < nbh; by++) for (int bx = 0; bx < nbw; bx++) {
      const int rr = r + by * steph, cc = c + bx * stepw, tmi = rr * ms + cc;
      if (rr >= f->mi_rows || cc >= f->mi_cols) continue;   /* transform blocks that start outside the frame are not coded */
      const int eob = f->m_eob[p][tmi];
      const int32_t *src = f->coef[p] + (rr * 4) * f->stride + cc * 4;
      for (int i = 0; i < nh; i++) memcpy(qc + i * n, src + i * f->stride, sizeof(int32_t) * (size_t)n);
      int txtype, off = -1, sym = 0, ns = 0, set;
      if (p == 0) {
        txtype = f->m_txtype[tmi];
        off = av1o_intra_tx_cdf(f, txs, ymode, &ns, &set);
        if (off >= 0) sym = av1o_tx_type_to_symbol(set, txtype);
      } else {
        set = av1o_tx_set(txs, f->cfg.reduced_tx_set);
        txtype = av1o_mode_to_txtype(uvmode);
        if (!av1o_tx_type_in_set(set, txtype)) txtype = DCT_DCT;
      }
      int sctx2, dctx, cul, dcc;
      /* contexts must be derived from neighbours exactly as phase 1 left them */
      av1o_txb_ctx(f, t, p, rr, cc, txs, bs, &sctx2, &dctx);
      av1o_code_coeffs(f, qc, eob, p, txs, txtype, sctx2, dctx, off, sym, ns, &k, &cul, &dcc);
    }
  }
}

static void write_partition(TileW *w, int r, int c, int bs) {
  Av1oFrame *f = w->f; const TileB *t = &w->t; const int ms = f->mi_stride;
  if (r >= f->mi_rows || c >= f->mi_cols) return;
  const int half = (1 << bs) >> 1;
  const int has_rows = (r + half) < f->mi_rows, has_cols = (c + half) < f->mi_cols;
  const int actual = f->m_bsize[r * ms + c];
  int part = PARTITION_NONE;
  if (bs >= BS_8) {
    part = actual == bs ? PARTITION_NONE : (bs == BS_8 && actual == BS_8X4 ? PARTITION_HORZ : (bs == BS_8 && actual == BS_4X8 ? PARTITION_VERT : PARTITION_SPLIT));
    const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
    const int above = availU && dim_wl(f->m_bsize[(r - 1) * ms + c]) < 2 + bs, left = availL && dim_hl(f->m_bsize[r * ms + c - 1]) < 2 + bs;
    const int ctx = left * 2 + above;
    uint16_t *cdf = w->cdf + CDF_PARTITION + ((bs - 1) * 4 + ctx) * CDF_PARTITION_STRIDE;
    const int ns = bs == BS_8 ? 4 : 10;
    if (has_rows && has_cols) re_symbol(&w->ec, part, cdf, ns);
    else if (has_rows || has_cols) {
      /* split_or_horz / split_or_vert: bool from gathered partition probabilities, not adapted */
      #define PP(i) ((uint32_t)((i) > 0 ? cdf[(i) - 1] : 32768) - cdf[i])
      uint32_t psum;
      if (has_cols) psum = PP(2) + PP(3) + PP(4) + PP(6) + PP(7) + PP(9);      /* VERT SPLIT HORZ_A VERT_A VERT_B VERT_4 */
      else psum = PP(1) + PP(3) + PP(4) + PP(5) + PP(6) + PP(8);               /* HORZ SPLIT HORZ_A HORZ_B VERT_A HORZ_4 */
      #undef PP
      uint16_t bc[3] = { (uint16_t)psum, 0, 0 };   /* inverse cdf: P(symbol 0) = 32768 - psum */
      part = PARTITION_SPLIT;                        /* this encoder always splits at frame edges */
      re_symbol_noadapt(&w->ec, 1, bc, 2);
    } else part = PARTITION_SPLIT;
  }
  if (part == PARTITION_NONE) write_block(w, r, c, bs);
  else if (part == PARTITION_HORZ) { write_block(w, r, c, BS_8X4); write_block(w, r + 1, c, BS_8X4); }
  else if (part == PARTITION_VERT) { write_block(w, r, c, BS_4X8); write_block(w, r, c + 1, BS_4X8); }
  else {
    write_partition(w, r, c, bs - 1); write_partition(w, r, c + half, bs - 1);
    write_partition(w, r + half, c, bs - 1); write_partition(w, r + half, c + half, bs - 1);
  }
}

size_t av1o_code_tile(Av1oFrame *f, int tile_row, int tile_col, uint8_t **out) {
  TileW *w = (TileW *)malloc(sizeof(TileW));
  w->f = f;
  w->t.mi_row_start = f->tile_row_start[tile_row] * SB_MI; w->t.mi_row_end = imin(f->tile_row_start[tile_row + 1] * SB_MI, f->mi_rows);
  w->t.mi_col_start = f->tile_col_start[tile_col] * SB_MI; w->t.mi_col_end = imin(f->tile_col_start[tile_col + 1] * SB_MI, f->mi_cols);
  memcpy(w->cdf, f->cdf0, sizeof(w->cdf));
  w->cdef_done = (uint8_t *)calloc((size_t)f->sb_rows * f->sb_cols, 1);
  re_init(&w->ec);
  const char *hist_path = getenv("AV1O_SYM_HIST");              /* tools only, single-threaded runs (av1o_common.c) */
  if (hist_path) { av1o_hist = (unsigned *)calloc(65536, sizeof(unsigned)); av1o_hist_base = w->cdf; }
  /* tile start: switchable restoration_type CDF (libaom AOM_CDF3(9413, 22581)), RefSgrXqd = Sgrproj_Xqd_Mid */
  w->lr_cdf[0] = 32768 - 9413; w->lr_cdf[1] = 32768 - 22581; w->lr_cdf[2] = 0; w->lr_cdf[3] = 0;
  for (int p = 0; p < 3; p++) { w->lr_ref[p][0] = -32; w->lr_ref[p][1] = 31; }
  for (int r = w->t.mi_row_start; r < w->t.mi_row_end; r += SB_MI)
    for (int c = w->t.mi_col_start; c < w->t.mi_col_end; c += SB_MI) {
      av1o_write_lr_sb(f, r, c, w->lr_ref, lr_sym, ec_lit, w);      /* read_lr() precedes decode_partition() (spec 5.11.2) */
      write_partition(w, r, c, BS_64);
    }
  if (f->tile_cdf) memcpy(f->tile_cdf + (size_t)(tile_row * f->tile_cols + tile_col) * CDF_TOTAL, w->cdf, sizeof(w->cdf));   /* what a second pass prices this tile against */
  if (hist_path) { FILE *fh = fopen(hist_path, "ab"); if (fh) { fwrite(av1o_hist, sizeof(unsigned), 65536, fh); fclose(fh); } av1o_hist_base = NULL; free(av1o_hist); av1o_hist = NULL; }
  size_t n = re_finish(&w->ec, out);
  re_free(&w->ec);
  free(w->cdef_done);
  free(w);
  return n;
}

/* ---------------- experiment (R-11): pricing against the tile's LIVE CDFs inside one pass ----------------
 * rav1e prices every RDO decision against the adaptive state of the tile (receive_packet, ravif/src/av1encoder.rs:759).  This encoder prices against a static
 * table.  AV1O_LIVE_CDF=1 (environment, oracle only) measures what the difference is worth: after every finished superblock the tile's
 * symbols so far are run through the adaptation (the real writer on a scratch coder; cdef / restoration syntax carries no adaptive symbols the search prices)
 * and the search's rate table is rebuilt from the adapted CDFs.  It serialises a tile's superblocks, which is why the HIP path does not do it. */
void *av1o_live_open(Av1oFrame *f, int tile_row, int tile_col) {
  TileW *w = (TileW *)malloc(sizeof(TileW));
  w->f = f;
  w->t.mi_row_start = f->tile_row_start[tile_row] * SB_MI; w->t.mi_row_end = imin(f->tile_row_start[tile_row + 1] * SB_MI, f->mi_rows);
  w->t.mi_col_start = f->tile_col_start[tile_col] * SB_MI; w->t.mi_col_end = imin(f->tile_col_start[tile_col + 1] * SB_MI, f->mi_cols);
  memcpy(w->cdf, f->cdf0, sizeof(w->cdf));
  w->cdef_done = (uint8_t *)calloc((size_t)f->sb_rows * f->sb_cols, 1);
  re_init(&w->ec);
  return w;
}
void av1o_live_sb(void *wv, int r, int c, uint32_t *cost_out) {
  TileW *w = (TileW *)wv;
  const int save = w->f->enable_cdef;
  w->f->enable_cdef = 0;                         /* (the cdef index is a literal and not known yet) */
  write_partition(w, r, c, BS_64);
  w->f->enable_cdef = save;
  w->ec.offs = 0;                                /* the coded bytes are not wanted */
  av1o_costs_from_cdfs(w->cdf, cost_out);
}
void av1o_live_close(void *wv) { TileW *w = (TileW *)wv; re_free(&w->ec); free(w->cdef_done); free(w); }

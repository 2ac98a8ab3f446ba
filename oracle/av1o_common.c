/* oracle/av1o_common.c -- bit writer, range encoder, table helpers, rate tables, quantizer and tile setup.
 * TEST INFRASTRUCTURE (see av1o.h).  Spec references: AV1 bitstream spec sections 4.10 (descriptors),
 * 8.2 (symbol decoding; the encoder here is its inverse), 5.9.15 (tile info). */
#include "av1o_int.h"
#include <math.h>
#include <limits.h>

/* ---------------- bit writer ---------------- */
void bw_init(BitW *b) { b->cap = 256; b->buf = (uint8_t *)calloc(b->cap, 1); b->bitpos = 0; }
static void bw_grow(BitW *b, size_t need_bits) {
  size_t need = (b->bitpos + need_bits + 7) / 8 + 8;
  if (need > b->cap) {
    size_t nc = b->cap * 2 > need ? b->cap * 2 : need;
    b->buf = (uint8_t *)realloc(b->buf, nc);
    memset(b->buf + b->cap, 0, nc - b->cap);
    b->cap = nc;
  }
}
void bw_put(BitW *b, uint32_t v, int n) {
  bw_grow(b, (size_t)n);
  for (int i = n - 1; i >= 0; i--) {
    if ((v >> i) & 1) b->buf[b->bitpos >> 3] |= (uint8_t)(0x80 >> (b->bitpos & 7));
    b->bitpos++;
  }
}
void bw_su(BitW *b, int v, int n) { bw_put(b, (uint32_t)v & ((1u << n) - 1), n); }
void bw_align(BitW *b) { while (b->bitpos & 7) bw_put(b, 0, 1); }
void bw_trailing(BitW *b) { bw_put(b, 1, 1); bw_align(b); }
size_t bw_bytes(const BitW *b) { return (b->bitpos + 7) >> 3; }
size_t leb128_put(uint8_t *dst, uint64_t v) {
  size_t n = 0;
  do { uint8_t byte = v & 0x7f; v >>= 7; if (v) byte |= 0x80; dst[n++] = byte; } while (v);
  return n;
}

/* ---------------- range encoder ---------------- */
void re_init(RangeEnc *e) {
  e->pre_cap = 1024; e->pre = (uint16_t *)malloc(e->pre_cap * sizeof(uint16_t));
  e->offs = 0; e->low = 0; e->rng = 0x8000; e->cnt = -9;
}
void re_free(RangeEnc *e) { free(e->pre); e->pre = NULL; }
static inline int ilog_nz(uint32_t v) { return 32 - __builtin_clz(v); }
static void re_normalize(RangeEnc *e, uint64_t low, uint32_t rng) {
  int c = e->cnt;
  int d = 16 - ilog_nz(rng);
  int s = c + d;
  if (s >= 0) {
    if (e->offs + 2 > e->pre_cap) { e->pre_cap *= 2; e->pre = (uint16_t *)realloc(e->pre, e->pre_cap * sizeof(uint16_t)); }
    c += 16;
    uint64_t m = ((uint64_t)1 << c) - 1;
    if (s >= 8) { e->pre[e->offs++] = (uint16_t)(low >> c); low &= m; c -= 8; m >>= 8; }
    e->pre[e->offs++] = (uint16_t)(low >> c);
    s = c + d - 24;
    low &= m;
  }
  e->low = low << d; e->rng = rng << d; e->cnt = s;
}
static void re_encode_q15(RangeEnc *e, uint32_t fl, uint32_t fh, int s, int nsyms) {
  uint64_t l = e->low; uint32_t r = e->rng;
  const int N = nsyms - 1;
  if (fl < 32768) {
    uint32_t u = (((r >> 8) * (fl >> 6)) >> 1) + 4 * (uint32_t)(N - (s - 1));
    uint32_t v = (((r >> 8) * (fh >> 6)) >> 1) + 4 * (uint32_t)(N - s);
    l += r - u; r = u - v;
  } else {
    r -= (((r >> 8) * (fh >> 6)) >> 1) + 4 * (uint32_t)(N - s);
  }
  re_normalize(e, l, r);
}
void re_symbol_noadapt(RangeEnc *e, int s, const uint16_t *icdf, int nsyms) {
  re_encode_q15(e, s > 0 ? icdf[s - 1] : 32768, icdf[s], s, nsyms);
}
/* tools/k4_row_balance.py only (how the entropy kernel's adapter waves should share the CDF rows): with AV1O_SYM_HIST=<file> in the environment av1o_code_tile
 * points av1o_hist_base at the tile's CDF array and every adaptive symbol is counted under its row's offset; the tile's 65536 counters are appended to the file */
const uint16_t *av1o_hist_base = NULL; unsigned *av1o_hist = NULL;
void re_symbol(RangeEnc *e, int s, uint16_t *icdf, int nsyms) {
  re_symbol_noadapt(e, s, icdf, nsyms);
  if (av1o_hist_base && icdf >= av1o_hist_base && icdf - av1o_hist_base < 65536) av1o_hist[icdf - av1o_hist_base]++;
  /* spec 8.3.2 symbol adaptation, on inverse CDFs */
  int cnt = icdf[nsyms];
  int rate = 3 + (cnt > 15) + (cnt > 31) + imin(ilog_nz((uint32_t)nsyms) - 1, 2);
  for (int i = 0; i < nsyms - 1; i++) {
    if (i < s) icdf[i] += (uint16_t)((32768 - icdf[i]) >> rate);
    else icdf[i] -= (uint16_t)(icdf[i] >> rate);
  }
  icdf[nsyms] = (uint16_t)(cnt + (cnt < 32));
}
void re_literal(RangeEnc *e, uint32_t v, int nbits) {
  static const uint16_t half[3] = { 16384, 0, 0 };
  for (int i = nbits - 1; i >= 0; i--) re_symbol_noadapt(e, (int)((v >> i) & 1), half, 2);
}
size_t re_finish(RangeEnc *e, uint8_t **out) {
  uint64_t l = e->low; int c = e->cnt; int s = 10;
  uint64_t m = 0x3FFF;
  uint64_t x = ((l + m) & ~m) | (m + 1);
  s += c;
  if (s > 0) {
    uint64_t n = ((uint64_t)1 << (c + 16)) - 1;
    do {
      if (e->offs + 1 > e->pre_cap) { e->pre_cap *= 2; e->pre = (uint16_t *)realloc(e->pre, e->pre_cap * sizeof(uint16_t)); }
      e->pre[e->offs++] = (uint16_t)(x >> (c + 16));
      x &= n; s -= 8; c -= 8; n >>= 8;
    } while (s > 0);
  }
  size_t nb = e->offs;
  uint8_t *buf = (uint8_t *)malloc(nb ? nb : 1);
  uint32_t carry = 0;
  for (size_t i = nb; i-- > 0;) { carry = e->pre[i] + carry; buf[i] = (uint8_t)carry; carry >>= 8; }
  *out = buf;
  return nb;
}

/* ---------------- tx type helpers (spec 5.11.47 get_tx_set, 6.10.19 tables) ---------------- */
static const uint8_t kInvSet1[7] = { IDTX, DCT_DCT, V_DCT, H_DCT, ADST_ADST, ADST_DCT, DCT_ADST };
static const uint8_t kInvSet2[5] = { IDTX, DCT_DCT, ADST_ADST, ADST_DCT, DCT_ADST };
int av1o_tx_set(int txs, int reduced) {          /* spec get_tx_set, intra: the larger dimension bounds the set, the smaller picks it */
  if (dim_max_l(txs) >= 5) return 0;
  if (reduced) return 2;
  if (dim_min_l(txs) == 4) return 2;
  return 1;
}
int av1o_tx_set_count(int set) { return set == 0 ? 1 : (set == 1 ? 7 : 5); }
int av1o_symbol_to_tx_type(int set, int sym) { return set == 0 ? DCT_DCT : (set == 1 ? kInvSet1[sym] : kInvSet2[sym]); }
int av1o_tx_type_to_symbol(int set, int txtype) {
  int n = av1o_tx_set_count(set);
  for (int i = 0; i < n; i++) if (av1o_symbol_to_tx_type(set, i) == txtype) return i;
  return -1;
}
int av1o_tx_type_in_set(int set, int txtype) { return av1o_tx_type_to_symbol(set, txtype) >= 0; }
int av1o_mode_to_txtype(int mode) {
  static const uint8_t m2t[14] = { DCT_DCT, ADST_DCT, DCT_ADST, DCT_DCT, ADST_ADST, ADST_DCT, DCT_ADST,
                                   DCT_ADST, ADST_DCT, ADST_ADST, ADST_DCT, DCT_ADST, ADST_ADST, DCT_DCT };
  return m2t[mode];
}
int av1o_tx_class(int t) {
  if (t == V_DCT || t == V_ADST || t == V_FLIPADST) return TX_CLASS_VERT;
  if (t == H_DCT || t == H_ADST || t == H_FLIPADST) return TX_CLASS_HORIZ;
  return TX_CLASS_2D;
}
const uint16_t *av1o_scan(int txs, int txtype, uint16_t *tmp) {
  const int w = imin(32, 1 << dim_wl(txs)), h = imin(32, 1 << dim_hl(txs));
  int cls = av1o_tx_class(txtype);
  if (cls == TX_CLASS_2D && !dim_is_rect(txs)) {
    switch (w) { case 4: return av1_default_scan_4x4; case 8: return av1_default_scan_8x8;
                 case 16: return av1_default_scan_16x16; default: return av1_default_scan_32x32; }
  }
  if (cls == TX_CLASS_2D) {
    /* spec Default_Scan_4x8 / 8x4: plain anti-diagonals; a tall block walks each one from its top-right end, a wide block from its
     * bottom-left end */
    int k = 0;
    for (int d = 0; d < w + h - 1; d++) {
      if (h > w) { for (int r = 0; r < h; r++) { const int c = d - r; if (c >= 0 && c < w) tmp[k++] = (uint16_t)(r * w + c); } }
      else { for (int r = h - 1; r >= 0; r--) { const int c = d - r; if (c >= 0 && c < w) tmp[k++] = (uint16_t)(r * w + c); } }
    }
    return tmp;
  }
  if (cls == TX_CLASS_VERT) { for (int i = 0; i < w * h; i++) tmp[i] = (uint16_t)i; }           /* mrow scan */
  else { int k = 0; for (int c = 0; c < w; c++) for (int r = 0; r < h; r++) tmp[k++] = (uint16_t)(r * w + c); } /* mcol */
  return tmp;
}

/* ---------------- static rate tables ---------------- */
/* (15 - log2(p)) in 1/512 bit, integer only (bit-by-bit log2 by repeated squaring). */
static uint32_t neglog2_q9(uint32_t p) {
  if (p < 1) p = 1;
  int msb = 31 - __builtin_clz(p);
  uint64_t x = (uint64_t)p << (31 - msb);       /* [2^31, 2^32) */
  uint32_t frac = 0;
  for (int i = 0; i < 9; i++) {
    x = (x * x) >> 31;
    frac <<= 1;
    if (x >= ((uint64_t)1 << 32)) { frac |= 1; x >>= 1; }
  }
  return (uint32_t)(15 * 512 - (msb * 512 + (int)frac));
}
uint32_t av1o_cost_from_icdf(const uint16_t *icdf, int s, int nsyms) {
  (void)nsyms;
  uint32_t hi = s > 0 ? icdf[s - 1] : 32768, lo = icdf[s];
  return neglog2_q9(hi - lo);
}
static void cost_rows(const uint16_t *cdf, uint32_t *cost, int off, int stride, int nrows, int nsyms) {
  for (int r = 0; r < nrows; r++)
    for (int s = 0; s < nsyms; s++)
      cost[off + r * stride + s] = av1o_cost_from_icdf(cdf + off + r * stride, s, nsyms);
}
void av1o_costs_from_cdfs(const uint16_t *cdf, uint32_t *cost) {
  memset(cost, 0, sizeof(uint32_t) * CDF_TOTAL);
  cost_rows(cdf, cost, CDF_KF_Y, CDF_KF_Y_STRIDE, 25, 13);
  cost_rows(cdf, cost, CDF_ANGLE, CDF_ANGLE_STRIDE, 8, 7);
  cost_rows(cdf, cost, CDF_UV_NOCFL, CDF_UV_NOCFL_STRIDE, 13, 13);
  cost_rows(cdf, cost, CDF_UV_CFL, CDF_UV_CFL_STRIDE, 13, 14);
  cost_rows(cdf, cost, CDF_PARTITION, CDF_PARTITION_STRIDE, 4, 4);
  cost_rows(cdf, cost, CDF_PARTITION + 4 * CDF_PARTITION_STRIDE, CDF_PARTITION_STRIDE, 12, 10);
  cost_rows(cdf, cost, CDF_PARTITION + 16 * CDF_PARTITION_STRIDE, CDF_PARTITION_STRIDE, 4, 8);
  cost_rows(cdf, cost, CDF_SKIP, CDF_SKIP_STRIDE, 3, 2);
  cost_rows(cdf, cost, CDF_SEG_ID, CDF_SEG_ID_STRIDE, 3, 8);
  cost_rows(cdf, cost, CDF_INTRA_TX1, CDF_INTRA_TX1_STRIDE, 26, 7);
  cost_rows(cdf, cost, CDF_INTRA_TX2, CDF_INTRA_TX2_STRIDE, 39, 5);
  cost_rows(cdf, cost, CDF_CFL_SIGN, CDF_CFL_SIGN_STRIDE, 1, 8);
  cost_rows(cdf, cost, CDF_CFL_ALPHA, CDF_CFL_ALPHA_STRIDE, 6, 16);
  cost_rows(cdf, cost, CDF_TX_SIZE, CDF_TX_SIZE_STRIDE, 3, 2);
  cost_rows(cdf, cost, CDF_TX_SIZE + 3 * CDF_TX_SIZE_STRIDE, CDF_TX_SIZE_STRIDE, 9, 3);
  cost_rows(cdf, cost, CDF_TXB_SKIP, CDF_TXB_SKIP_STRIDE, 5 * 13, 2);
  cost_rows(cdf, cost, CDF_EOB_EXTRA, CDF_EOB_EXTRA_STRIDE, 5 * 2 * 9, 2);
  cost_rows(cdf, cost, CDF_DC_SIGN, CDF_DC_SIGN_STRIDE, 6, 2);
  cost_rows(cdf, cost, CDF_COEFF_BR, CDF_COEFF_BR_STRIDE, 5 * 2 * 21, 4);
  cost_rows(cdf, cost, CDF_COEFF_BASE, CDF_COEFF_BASE_STRIDE, 5 * 2 * 42, 4);
  cost_rows(cdf, cost, CDF_COEFF_BASE_EOB, CDF_COEFF_BASE_EOB_STRIDE, 5 * 2 * 4, 3);
  cost_rows(cdf, cost, CDF_EOB_PT_16, CDF_EOB_PT_16_STRIDE, 4, 5);
  cost_rows(cdf, cost, CDF_EOB_PT_32, CDF_EOB_PT_32_STRIDE, 4, 6);
  cost_rows(cdf, cost, CDF_EOB_PT_64, CDF_EOB_PT_64_STRIDE, 4, 7);
  cost_rows(cdf, cost, CDF_EOB_PT_128, CDF_EOB_PT_128_STRIDE, 4, 8);
  cost_rows(cdf, cost, CDF_EOB_PT_256, CDF_EOB_PT_256_STRIDE, 4, 9);
  cost_rows(cdf, cost, CDF_EOB_PT_512, CDF_EOB_PT_512_STRIDE, 4, 10);
  cost_rows(cdf, cost, CDF_EOB_PT_1024, CDF_EOB_PT_1024_STRIDE, 4, 11);
}
void av1o_build_costs(Av1oFrame *f) {
  memcpy(f->cdf0, av1_default_cdfs + f->qctx * CDF_TOTAL, sizeof(f->cdf0));
  av1o_costs_from_cdfs(f->cdf0, f->cost0);
  f->cost = f->cost0;
}

/* ---------------- quantizer selection ----------------
 * [UPSTREAM-RECALL rav1e src/rate.rs: RCState::select_qi with bitrate == 0 and
 *  QuantizerParameters::new_from_log_q]  quantizer -> log-domain midpoint of the AC entry and the
 *  nearest DC entry -> key-frame offset DQP_Q57[KF] = -(33810170/86043287) -> per-plane targets with the
 *  4:4:4 chroma offsets -> nearest table entries in the log domain.  Evaluated in double; the
 *  resolved integers are compared with the product's in tests. */
static const int16_t *qtab(int bd, int dc) { return bd == 8 ? (dc ? av1_dc_q8 : av1_ac_q8) : (dc ? av1_dc_q10 : av1_ac_q10); }
static int select_qi(long q, const int16_t *t) {
  if (q < t[0]) return 0;
  if (q >= t[255]) return 255;
  int lo = 0, hi = 255;                       /* t[lo] <= q < t[hi] */
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (t[mid] <= q) lo = mid; else hi = mid; }
  if (t[lo] == q) { while (lo > 0 && t[lo - 1] == q) lo--; return lo; }
  long th = (long)t[lo] * t[hi];
  return (q * q < th) ? lo : hi;
}
void av1o_select_quantizers(Av1oFrame *f) {
  const int bd = f->bd;
  const int16_t *ac = qtab(bd, 0), *dc = qtab(bd, 1);
  const double norm = 3.0 + (bd - 8);                       /* QSCALE + bit_depth - 8 */
  long ac_quant = ac[f->cfg.quantizer];
  int dc_qi0 = select_qi(ac_quant, dc);
  double log_ac = log2((double)ac_quant) - norm, log_dc = log2((double)dc[dc_qi0]) - norm;
  double log_base = 0.5 * (log_ac + log_dc);
  /* (AV1O_SWEEP_KFQ=<thousandths>, AV1O_SWEEP_LAMBDA=<percent>, oracle only: the recalled key-frame offset and lambda scaled, for the constants sweep of BASELINE.md
   * section 5 -- tools/constants_sweep.py; unset = 1000 / 100) */
  const char *e_kf = getenv("AV1O_SWEEP_KFQ"), *e_lm = getenv("AV1O_SWEEP_LAMBDA");
  const double kf_scale = e_kf ? atoi(e_kf) / 1000.0 : 1.0; const long lm_pct = e_lm ? atoi(e_lm) : 100;
  double log_q = log_base - kf_scale * (33810170.0 / 86043287.0);     /* key frame */
  double x = log_q > 0 ? log_q : 0;
  double y = f->np == 1 ? 0.0 : x * (1.0 / 16 + 1.0 / 32 + 1.0 / 256);
  double off[3] = { 0.0, log2(7.0 / 4.0) - y, log2(5.0 / 4.0) - y };
  long qy = 0;
  for (int p = 0; p < f->np; p++) {
    long q = lround(exp2(log_q + off[p] + norm));
    if (p == 0) qy = q;
    int aqi = select_qi(q, ac), dqi = select_qi(q, dc);
    if (p == 0) { if (aqi < 1) aqi = 1; f->base_q_idx = aqi; }
    int lo = imax(1, f->base_q_idx - 63), hi = imin(255, f->base_q_idx + 63);
    f->ac_qi[p] = p == 0 ? aqi : iclamp(aqi, lo, hi);
    f->dc_qi[p] = iclamp(dqi, lo, hi);
    f->ac_q[p] = ac[f->ac_qi[p]];
    f->dc_q[p] = dc[f->dc_qi[p]];
  }
  /* lambda = ln2/6 * (q/8)^2 SSE per bit (rav1e), here in 1/128-SSE per 1/512-bit fixed point;
     planes are weighted by (q_y/q_p)^2 == rav1e dist_scale. */
  for (int p = 0; p < f->np; p++) f->rdmult[p] = (((int64_t)qy * qy * 242273) >> 20) * lm_pct / 100;
  f->qctx = f->base_q_idx <= 20 ? 0 : (f->base_q_idx <= 60 ? 1 : (f->base_q_idx <= 120 ? 2 : 3));
}

/* ---------------- tiles (spec 5.9.15 uniform spacing; target rule ravif av1encoder.rs:665-668) ---------------- */
static int tile_log2(int blk, int target) { int k = 0; while ((blk << k) < target) k++; return k; }
void av1o_setup_tiles(Av1oFrame *f) {
  long area = (long)f->cfg.width * f->cfg.height;
  long mts = f->cfg.min_tile_size > 0 ? f->cfg.min_tile_size : 128;
  long target = area / (mts * mts);
  if (f->cfg.threads > 0 && target > f->cfg.threads) target = f->cfg.threads;
  if (f->cfg.tiles_override > 0) target = f->cfg.tiles_override;
  int sbc = f->sb_cols, sbr = f->sb_rows;
  int min_cols_log2 = tile_log2(64, sbc);
  int max_cols_log2 = tile_log2(1, imin(sbc, MAX_TILE_COLS));
  int max_rows_log2 = tile_log2(1, imin(sbr, MAX_TILE_ROWS));
  int min_log2_tiles = imax(min_cols_log2, tile_log2(2304, sbr * sbc));
  int cl = min_cols_log2, rl = imax(min_log2_tiles - cl, 0);
  for (;;) {
    int tw = (sbc + (1 << cl) - 1) >> cl, th = (sbr + (1 << rl) - 1) >> rl;
    int ncols = (sbc + tw - 1) / tw, nrows = (sbr + th - 1) / th;
    if ((long)ncols * nrows >= target) break;
    if (cl >= max_cols_log2 && rl >= max_rows_log2) break;
    if ((th >= tw && rl < max_rows_log2) || cl >= max_cols_log2) rl++; else cl++;
  }
  f->tile_cols_log2 = cl; f->tile_rows_log2 = rl;
  int tw = (sbc + (1 << cl) - 1) >> cl, th = (sbr + (1 << rl) - 1) >> rl;
  int i = 0;
  for (int s = 0; s < sbc; s += tw) f->tile_col_start[i++] = s;
  f->tile_col_start[i] = sbc; f->tile_cols = i;
  i = 0;
  for (int s = 0; s < sbr; s += th) f->tile_row_start[i++] = s;
  f->tile_row_start[i] = sbr; f->tile_rows = i;
}

/* ---------------- Tune::Psychovisual (ravif/src/av1encoder.rs:694) ----------------
 * [UPSTREAM-RECALL rav1e src/activity.rs ActivityMask, src/dist.rs cdef_dist_kernel + apply_ssim_boost, rdo.rs
 *  compute_distortion]: luma distortion is the SSE of every 8x8 cell boosted by an SSIM-like factor of the source and
 *  reconstruction variances, and every cell carries an activity scale = the same factor at equal variances.
 *  Integer restatement (Q14), bit-identical on CPU and GPU:
 *    boost(sv, dv) = 4033/16384 * (sv + dv + 16384) / sqrt(4033^2 + sv * dv)      variances = 64 x per-sample variance, 8-bit scale
 *  4x4 blocks use their own 16-sample variance scaled to the 8x8 equivalent. */
/* floor(sqrt(n)), n < 2^46: a float guess made exact by integer correction (the same two steps run on the GPU, so the result
 * does not depend on how either side rounds its floating point) */
static uint32_t isqrt46(uint64_t n) {
  uint32_t x = (uint32_t)sqrtf((float)n);
  while ((uint64_t)x * x > n) x--;
  while ((uint64_t)(x + 1) * (x + 1) <= n) x++;
  return x;
}
uint32_t av1o_psy_boost_q14(uint32_t svar, uint32_t dvar) {
  /* Q14: 16384 * (4033 / 16384) * (sv + dv + 16384) / sqrt(4033^2 + sv * dv) = 4033 * (sv + dv + 16384) / sqrt(...) */
  const uint64_t num = 4033ull * ((uint64_t)svar + dvar + 16384);
  const uint32_t den = isqrt46(16265089ull + (uint64_t)svar * dvar);
  const uint64_t t = num + den / 2;
  uint32_t q = (uint32_t)((float)t / (float)den);              /* guess, then exact */
  while ((uint64_t)q * den > t) q--;
  while ((uint64_t)(q + 1) * den <= t) q++;
  return q;
}
/* variance of a w x w cell (w = 8 or 4) as 64 x per-sample variance on the 8-bit scale */
uint32_t av1o_cell_var(int64_t sum, int64_t sum2, int w, int bd) {
  const int64_t v = w == 8 ? sum2 - ((sum * sum + 32) >> 6) : (sum2 - ((sum * sum + 8) >> 4)) << 2;
  return (uint32_t)((v < 0 ? 0 : v) >> (2 * (bd - 8)));
}
void av1o_activity(Av1oFrame *f) {
  const int cw = f->pw / 8, chh = f->ph / 8;
  f->act = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cw * chh);
  f->svar8 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cw * chh);
  f->svar4 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)f->mi_stride * f->mi_h);
  for (int cy = 0; cy < chh; cy++) for (int cx = 0; cx < cw; cx++) {
    int64_t s8 = 0, q8 = 0;
    for (int k = 0; k < 4; k++) {
      int64_t s4 = 0, q4 = 0;
      const int x0 = cx * 8 + (k & 1) * 4, y0 = cy * 8 + (k >> 1) * 4;
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { const int v = f->src[0][(size_t)(y0 + i) * f->stride + x0 + j]; s4 += v; q4 += v * v; }
      f->svar4[(y0 >> 2) * f->mi_stride + (x0 >> 2)] = av1o_cell_var(s4, q4, 4, f->bd);
      s8 += s4; q8 += q4;
    }
    const uint32_t v = av1o_cell_var(s8, q8, 8, f->bd);
    f->svar8[cy * cw + cx] = v;
    f->act[cy * cw + cx] = f->cfg.tune_psnr ? 16384u : av1o_psy_boost_q14(v, v);
  }
}
/* luma distortion of the bw x bh block at pixel (x, y): rec has pitch rs; 8x8 cells, 4x4 cells when a dimension is 4 */
int64_t av1o_psy_dist_luma_wh(const Av1oFrame *f, const uint16_t *rec, int rs, int x, int y, int bw, int bh) {
  const int cw = f->pw / 8, w = imin(bw, bh) == 4 ? 4 : 8;
  int64_t total = 0;
  for (int by = 0; by < bh; by += w) for (int bx = 0; bx < bw; bx += w) {
    int64_t sd = 0, qd = 0, sse = 0;
    for (int i = 0; i < w; i++) for (int j = 0; j < w; j++) {
      const int d = rec[(by + i) * rs + bx + j], sv = f->src[0][(size_t)(y + by + i) * f->stride + x + bx + j];
      sd += d; qd += d * d; sse += (int64_t)(sv - d) * (sv - d);
    }
    if (f->cfg.tune_psnr) { total += sse; continue; }
    const int cell = ((y + by) >> 3) * cw + ((x + bx) >> 3);
    const uint32_t svar = w == 4 ? f->svar4[((y + by) >> 2) * f->mi_stride + ((x + bx) >> 2)] : f->svar8[cell];
    const uint32_t b = av1o_psy_boost_q14(svar, av1o_cell_var(sd, qd, w, f->bd));
    int64_t d = (sse * b + 8192) >> 14;
    d = (d * f->act[cell] + 8192) >> 14;
    total += d;
  }
  return total;
}
int64_t av1o_psy_dist_luma(const Av1oFrame *f, const uint16_t *rec, int rs, int x, int y, int n) { return av1o_psy_dist_luma_wh(f, rec, rs, x, y, n, n); }
/* mean activity scale (Q14) of the cells covered by [x, x + w) x [y, y + h) */
uint32_t av1o_act_mean(const Av1oFrame *f, int x, int y, int w, int h) {
  const int cw = f->pw / 8, cx0 = x >> 3, cx1 = (x + w - 1) >> 3, cy0 = y >> 3, cy1 = (y + h - 1) >> 3;
  uint64_t s = 0; const uint64_t cnt = (uint64_t)(cx1 - cx0 + 1) * (cy1 - cy0 + 1);
  for (int cy = cy0; cy <= cy1; cy++) for (int cx = cx0; cx <= cx1; cx++) s += f->act[cy * cw + cx];
  return (uint32_t)((s + cnt / 2) / cnt);
}

void av1o_free(void *p) { free(p); }

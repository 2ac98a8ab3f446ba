/* oracle/av1o_lrf.c -- loop restoration (spec 7.17): the normative self-guided filter (sgrproj) with stripe
 * boundaries, and the encoder-side per-unit search.  TEST INFRASTRUCTURE (see av1o.h).
 * rav1e equivalent (absent, [UPSTREAM-RECALL]): src/lrf.rs -- sgrproj_solve (least-squares projection weights per
 * parameter set), rdo.rs::rdo_loop_decision (per-unit RD choice between RESTORE_NONE and sgrproj; Wiener is never
 * searched), RestorationPlaneConfig { lrf_type: RESTORE_SWITCHABLE }.  Enabled by SpeedTweaks.lrf
 * (ravif/src/av1encoder.rs:589), parameter-set list by sgr_complexity Reduced / Full (:573,:625).
 * Choices of this build (documented in DESIGN.md): restoration units are 64x64 for every plane (lr_unit_shift = 0), so a
 * unit is one 64-row stripe of one superblock column (the last unit of a row / column absorbs a remainder < 32);
 * the xqd rate is priced against the tile-start reference (static, like every other rate of phase 1). */
#include "av1o_int.h"

static const int16_t sgr_params[16][4] = {       /* spec Sgr_Params: r0, s0, r1, s1  (s = round(2^20 / (n^2 eps))) */
  { 2, 140, 1, 3236 }, { 2, 112, 1, 2158 }, { 2, 93, 1, 1618 }, { 2, 80, 1, 1438 }, { 2, 70, 1, 1295 }, { 2, 58, 1, 1177 },
  { 2, 47, 1, 1079 }, { 2, 37, 1, 996 }, { 2, 30, 1, 925 }, { 2, 25, 1, 863 }, { 0, -1, 1, 2589 }, { 0, -1, 1, 1618 },
  { 0, -1, 1, 1177 }, { 0, -1, 1, 925 }, { 2, 56, 0, -1 }, { 2, 22, 0, -1 } };
static const int xqd_min[2] = { -96, -32 }, xqd_max[2] = { 31, 95 }, xqd_mid[2] = { -32, 31 };
/* parameter sets searched: [UPSTREAM-RECALL, uncertain] rav1e SGRPROJ_REDUCED_SETS / SGRPROJ_ALL_SETS */
static const uint8_t sets_reduced[4] = { 1, 3, 6, 11 };

int av1o_lr_units(int size) { int n = (size + 32) / 64; return n < 1 ? 1 : n; }

/* get_source_sample (spec 7.17.6): CDEF output inside the stripe, deblocked frame for the two rows above / below it */
typedef struct { const uint16_t *cdef, *dbk; int stride, end_x, end_y, stripe_start, stripe_end; } LrSrc;
static inline int lr_sample(const LrSrc *s, int x, int y) {
  x = iclamp(x, 0, s->end_x); y = iclamp(y, 0, s->end_y);
  if (y < s->stripe_start) { y = imax(s->stripe_start - 2, y); return s->dbk[(size_t)y * s->stride + x]; }
  if (y > s->stripe_end) { y = imin(s->stripe_end + 2, y); return s->dbk[(size_t)y * s->stride + x]; }
  return s->cdef[(size_t)y * s->stride + x];
}

/* box filter process (spec 7.17.3) for the region x0..x0+w, y0..y0+h (inside one stripe): flt[h][w] in SGRPROJ_RST_BITS precision */
static void box_filter(const LrSrc *s, int bd, int x0, int y0, int w, int h, int r, int sparam, int pass, int32_t *flt /* pitch w */) {
  const int n = (2 * r + 1) * (2 * r + 1), one_by_n = ((1 << 12) + n / 2) / n;
  const int pw = w + 2;
  int32_t *A = (int32_t *)malloc(sizeof(int32_t) * (size_t)pw * (h + 2)), *B = (int32_t *)malloc(sizeof(int32_t) * (size_t)pw * (h + 2));
  for (int i = -1; i <= h; i++) for (int j = -1; j <= w; j++) {
    uint32_t a = 0, b = 0;
    for (int dy = -r; dy <= r; dy++) for (int dx = -r; dx <= r; dx++) { const uint32_t c = (uint32_t)lr_sample(s, x0 + j + dx, y0 + i + dy); a += c * c; b += c; }
    const uint32_t ar = (uint32_t)round2_64(a, 2 * (bd - 8)), d = (uint32_t)round2_64(b, bd - 8);
    const uint32_t p = ar * (uint32_t)n > d * d ? ar * (uint32_t)n - d * d : 0;
    const uint32_t z = (uint32_t)round2_64((int64_t)p * sparam, 20);
    const uint32_t a2 = z >= 255 ? 256 : (z == 0 ? 1 : ((z << 8) + z / 2) / (z + 1));
    const uint32_t b2 = (256 - a2) * b * (uint32_t)one_by_n;
    A[(i + 1) * pw + j + 1] = (int32_t)a2; B[(i + 1) * pw + j + 1] = (int32_t)round2_64(b2, 12);
  }
  for (int i = 0; i < h; i++) {
    const int yabs = y0 + i;
    const int shift = (pass == 0 && (yabs & 1)) ? 4 : 5;
    for (int j = 0; j < w; j++) {
      int32_t a = 0, b = 0;
      for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        int wt;
        if (pass == 0) wt = ((yabs + dy) & 1) ? (dx == 0 ? 6 : 5) : 0;
        else wt = (dx == 0 || dy == 0) ? 4 : 3;
        a += wt * A[(i + 1 + dy) * pw + j + 1 + dx]; b += wt * B[(i + 1 + dy) * pw + j + 1 + dx];
      }
      const int32_t v = a * (int32_t)s->cdef[(size_t)yabs * s->stride + x0 + j] + b;
      flt[i * w + j] = round2(v, 8 + shift - 4);
    }
  }
  free(A); free(B);
}

/* self-guided projection of one sample (spec 7.17.2) */
static inline int sgr_project(int cdef, int f0, int f1, int r0, int r1, int w0, int w1, int bd) {
  const int u = cdef << 4, w2 = 128 - w0 - w1;
  int v = w1 * u;
  v += w0 * (r0 ? f0 : u);
  v += w2 * (r1 ? f1 : u);
  return iclamp(round2(v, 11), 0, (1 << bd) - 1);
}

/* round(128 * num / det) for det > 0, saturated well outside the xqd range */
static int ratio_q7(int64_t num, int64_t det) {
  while (det >= ((int64_t)1 << 54)) { det >>= 1; num >>= 1; }
  const int neg = num < 0; if (neg) num = -num;
  if (num >= det * 4) return neg ? -512 : 512;
  const int64_t q = (num * 128 + det / 2) / det;
  return (int)(neg ? -q : q);
}
/* sgrproj_solve: least-squares weights of (flt0 - u, flt1 - u) against (src - u); integer restatement */
void av1o_sgr_solve(int64_t h00, int64_t h11, int64_t h01, int64_t c0, int64_t c1, int r0, int r1, int *xqd0, int *xqd1) {
  /* common scaling so that every product below fits 64 bits */
  int64_t m = 0;
  const int64_t v[5] = { h00, h11, h01, c0, c1 };
  for (int i = 0; i < 5; i++) { const int64_t a = v[i] < 0 ? -v[i] : v[i]; if (a > m) m = a; }
  int sh = 0; while ((m >> sh) >= ((int64_t)1 << 30)) sh++;
  h00 >>= sh; h11 >>= sh; h01 >>= sh; c0 >>= sh; c1 >>= sh;
  int xq0 = 0, xq1 = 0;
  if (r0 == 0) { if (h11 > 0) xq1 = ratio_q7(c1, h11); }
  else if (r1 == 0) { if (h00 > 0) xq0 = ratio_q7(c0, h00); }
  else {
    const int64_t det = h00 * h11 - h01 * h01;
    if (det > 0) { xq0 = ratio_q7(h11 * c0 - h01 * c1, det); xq1 = ratio_q7(h00 * c1 - h01 * c0, det); }
  }
  *xqd0 = iclamp(xq0, xqd_min[0], xqd_max[0]);
  *xqd1 = iclamp(128 - *xqd0 - xq1, xqd_min[1], xqd_max[1]);
  if (r0 == 0) *xqd0 = 0;                      /* spec read_lr_unit: not coded, 0 */
  if (r1 == 0) *xqd1 = 95;                     /* spec read_lr_unit: Clip3(-32, 95, 128 - xqd0) with xqd0 <= 31 */
}

/* bits of decode_signed_subexp_with_ref_bool(min, max + 1, k = 4, ref) for value v */
static int recenter(int r, int x) { return x > 2 * r ? x : (x >= r ? (x - r) << 1 : ((r - x) << 1) - 1); }
int av1o_subexp_code(int v, int lo, int hi_excl, int ref, uint32_t *bits /* nullable: MSB-first bit string */, int *nbits_out) {
  const int mx = hi_excl - lo, x = v - lo, r = ref - lo;
  int t = (r << 1) <= mx ? recenter(r, x) : recenter(mx - 1 - r, mx - 1 - x);
  uint32_t acc = 0; int nb = 0;
  int i = 0, mk = 0; const int k = 4;
  for (;;) {
    const int b2 = i ? k + i - 1 : k, a = 1 << b2;
    if (mx <= mk + 3 * a) {                                   /* uniform tail: ns(mx - mk) */
      const int nsy = mx - mk, val = t - mk;
      int w = 0; while ((1 << w) <= nsy) w++;                   /* w = FloorLog2(n) + 1 */
      const int m = (1 << w) - nsy;
      if (val < m) { acc = (acc << (w - 1)) | (uint32_t)val; nb += w - 1; }
      else { const int e = val + m; acc = (acc << (w - 1)) | (uint32_t)(e >> 1); nb += w - 1; acc = (acc << 1) | (uint32_t)(e & 1); nb += 1; }
      break;
    }
    if (t >= mk + a) { acc = (acc << 1) | 1; nb++; i++; mk += a; }
    else { acc = (acc << 1); nb++; acc = (acc << b2) | (uint32_t)(t - mk); nb += b2; break; }
  }
  if (bits) *bits = acc;
  *nbits_out = nb;
  return nb;
}

static uint32_t lr_rate(const Av1oFrame *f, int type, int set, int x0, int x1) {
  if (!type) return f->lr_cost[0];
  uint32_t rate = f->lr_cost[2] + 4 * 512;
  int nb;
  if (sgr_params[set][0]) { av1o_subexp_code(x0, xqd_min[0], xqd_max[0] + 1, xqd_mid[0], NULL, &nb); rate += 512u * (uint32_t)nb; }
  if (sgr_params[set][2]) { av1o_subexp_code(x1, xqd_min[1], xqd_max[1] + 1, xqd_mid[1], NULL, &nb); rate += 512u * (uint32_t)nb; }
  return rate;
}

void av1o_lr_search_and_apply(Av1oFrame *f) {
  const int W = f->w, H = f->h, bd = f->bd;
  f->lr_unit_cols = av1o_lr_units(W); f->lr_unit_rows = av1o_lr_units(H);
  const int nu = f->lr_unit_cols * f->lr_unit_rows;
  for (int p = 0; p < f->np; p++) { f->lr_type[p] = (uint8_t *)calloc((size_t)nu, 1); f->lr_set[p] = (uint8_t *)calloc((size_t)nu, 1); f->lr_xqd[p] = (int8_t *)calloc((size_t)nu * 2, 1); }
  if (!f->enable_restoration) return;
  /* switchable restoration_type: default CDF (libaom default_switchable_restore_cdf = AOM_CDF3(9413, 22581)) */
  static const uint16_t sw_icdf[3] = { 32768 - 9413, 32768 - 22581, 0 };
  for (int s = 0; s < 3; s++) f->lr_cost[s] = av1o_cost_from_icdf(sw_icdf, s, 3);
  int64_t wq[3];
  for (int p = 0; p < f->np; p++) wq[p] = (((int64_t)f->ac_q[0] * f->ac_q[0]) << 12) / ((int64_t)f->ac_q[p] * f->ac_q[p]);
  const int nsets = f->cfg.sgr_full ? 16 : 4;
  int32_t *flt[2]; flt[0] = (int32_t *)malloc(sizeof(int32_t) * 128 * 128); flt[1] = (int32_t *)malloc(sizeof(int32_t) * 128 * 128);
  for (int p = 0; p < f->np; p++) {
    uint16_t *out = (uint16_t *)malloc((size_t)f->pw * f->ph * 2);
    memcpy(out, f->rec[p], (size_t)f->pw * f->ph * 2);
    LrSrc s; s.cdef = f->rec[p]; s.dbk = f->dbk[p]; s.stride = f->stride; s.end_x = W - 1; s.end_y = H - 1;
    for (int ur = 0; ur < f->lr_unit_rows; ur++) for (int uc = 0; uc < f->lr_unit_cols; uc++) {
      const int x0 = uc * 64, x1 = uc == f->lr_unit_cols - 1 ? W : x0 + 64;
      const int y0 = imax(0, ur * 64 - 8), y1 = ur == f->lr_unit_rows - 1 ? H : ur * 64 + 56;
      const int uw = x1 - x0, ui = ur * f->lr_unit_cols + uc;
      int64_t sse_none = 0;
      for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) { const int d = (int)f->rec[p][(size_t)y * f->stride + x] - (int)f->src[p][(size_t)y * f->stride + x]; sse_none += d * d; }
      /* distortions of a unit are scaled by its mean activity (rav1e prices LRF decisions with the same per-8x8 bias;
         the cdef-dist kernel itself is not used here) */
      const int64_t act = av1o_act_mean(f, x0, y0, x1 - x0, y1 - y0);
      sse_none = (sse_none * act + 8192) >> 14;
      int64_t best_cost = ((sse_none * wq[p]) >> 5) + (((int64_t)lr_rate(f, 0, 0, 0, 0) * f->rdmult[0] + 256) >> 9);
      int best_type = 0, best_set = 0, best_x0 = 0, best_x1 = 0;
      for (int si = 0; si < nsets; si++) {
        const int set = f->cfg.sgr_full ? si : sets_reduced[si];
        const int r0 = sgr_params[set][0], r1 = sgr_params[set][2];
        int64_t h00 = 0, h11 = 0, h01 = 0, c0 = 0, c1 = 0;
        /* the unit stripe by stripe (a unit that absorbed a remainder spans two stripes) */
        for (int ys = y0; ys < y1;) {
          const int stripe = (ys + 8) / 64, ye = imin(y1, stripe * 64 + 56);
          s.stripe_start = stripe * 64 - 8; s.stripe_end = s.stripe_start + 63;
          if (r0) box_filter(&s, bd, x0, ys, uw, ye - ys, r0, sgr_params[set][1], 0, flt[0] + (ys - y0) * uw);
          if (r1) box_filter(&s, bd, x0, ys, uw, ye - ys, r1, sgr_params[set][3], 1, flt[1] + (ys - y0) * uw);
          ys = ye;
        }
        for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
          const int k = (y - y0) * uw + (x - x0);
          const int u = (int)f->rec[p][(size_t)y * f->stride + x] << 4, sv = ((int)f->src[p][(size_t)y * f->stride + x] << 4) - u;
          const int64_t f0 = r0 ? flt[0][k] - u : 0, f1 = r1 ? flt[1][k] - u : 0;
          h00 += f0 * f0; h11 += f1 * f1; h01 += f0 * f1; c0 += f0 * sv; c1 += f1 * sv;
        }
        int xq0, xq1;
        av1o_sgr_solve(h00, h11, h01, c0, c1, r0, r1, &xq0, &xq1);
        int64_t sse = 0;
        for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
          const int k = (y - y0) * uw + (x - x0);
          const int v = sgr_project(f->rec[p][(size_t)y * f->stride + x], flt[0][k], flt[1][k], r0, r1, xq0, xq1, bd);
          const int d = v - (int)f->src[p][(size_t)y * f->stride + x]; sse += d * d;
        }
        sse = (sse * act + 8192) >> 14;
        const int64_t cost = ((sse * wq[p]) >> 5) + (((int64_t)lr_rate(f, 1, set, xq0, xq1) * f->rdmult[0] + 256) >> 9);
        if (cost < best_cost) { best_cost = cost; best_type = 1; best_set = set; best_x0 = xq0; best_x1 = xq1; }
      }
      f->lr_type[p][ui] = (uint8_t)best_type; f->lr_set[p][ui] = (uint8_t)best_set; f->lr_xqd[p][2 * ui] = (int8_t)best_x0; f->lr_xqd[p][2 * ui + 1] = (int8_t)best_x1;
      if (best_type) {
        const int r0 = sgr_params[best_set][0], r1 = sgr_params[best_set][2];
        for (int ys = y0; ys < y1;) {
          const int stripe = (ys + 8) / 64, ye = imin(y1, stripe * 64 + 56);
          s.stripe_start = stripe * 64 - 8; s.stripe_end = s.stripe_start + 63;
          if (r0) box_filter(&s, bd, x0, ys, uw, ye - ys, r0, sgr_params[best_set][1], 0, flt[0] + (ys - y0) * uw);
          if (r1) box_filter(&s, bd, x0, ys, uw, ye - ys, r1, sgr_params[best_set][3], 1, flt[1] + (ys - y0) * uw);
          ys = ye;
        }
        for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
          const int k = (y - y0) * uw + (x - x0);
          out[(size_t)y * f->stride + x] = (uint16_t)sgr_project(f->rec[p][(size_t)y * f->stride + x], flt[0][k], flt[1][k], r0, r1, best_x0, best_x1, bd);
        }
      }
    }
    memcpy(f->rec[p], out, (size_t)f->pw * f->ph * 2);
    free(out);
  }
  free(flt[0]); free(flt[1]);
}

/* read_lr() for the superblock at (r, c) of a tile (spec 5.11.57/58), written through the sink of the tile writer */
void av1o_write_lr_sb(Av1oFrame *f, int r, int c, int ref_xqd[3][2], void (*sym)(void *, int, int), void (*lit)(void *, uint32_t, int), void *u) {
  if (!f->enable_restoration) return;
  for (int p = 0; p < f->np; p++) {
    const int urs = (r * 4 + 63) / 64, ure = imin(((r + 16) * 4 + 63) / 64, f->lr_unit_rows);
    const int ucs = (c * 4 + 63) / 64, uce = imin(((c + 16) * 4 + 63) / 64, f->lr_unit_cols);
    for (int ur = urs; ur < ure; ur++) for (int uc = ucs; uc < uce; uc++) {
      const int ui = ur * f->lr_unit_cols + uc;
      const int type = f->lr_type[p][ui];
      sym(u, type ? 2 : 0, 3);                                  /* restoration_type: NONE 0, WIENER 1, SGRPROJ 2 */
      if (!type) continue;
      const int set = f->lr_set[p][ui];
      lit(u, (uint32_t)set, 4);
      for (int i = 0; i < 2; i++) {
        const int v = f->lr_xqd[p][2 * ui + i];
        if (sgr_params[set][2 * i]) {
          uint32_t bits; int nb;
          av1o_subexp_code(v, xqd_min[i], xqd_max[i] + 1, ref_xqd[p][i], &bits, &nb);
          lit(u, bits, nb);
        }
        ref_xqd[p][i] = v;
      }
    }
  }
}

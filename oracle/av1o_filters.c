/* oracle/av1o_filters.c -- deblocking (spec 7.14) and CDEF (spec 7.15): normative filters plus the
 * encoder-side parameter choice.  TEST INFRASTRUCTURE (see av1o.h).
 * rav1e equivalents (absent): src/deblock.rs (deblock_filter_optimize, fast path = libaom's q-based formula),
 * src/cdef.rs + rdo.rs::rdo_loop_decision (per-64x64 strength index search, fixed 8-entry strength list). */
#include "av1o_int.h"
#include <stdio.h>

/* ------------------------------------------------------------------ deblock */
/* transform size at a 4x4 position: luma from the block's tx size, chroma (4:4:4) always the block's largest transform */
/* transform extent across the edge direction: width for vertical edges (pass 0), height for horizontal ones; chroma transforms = the block */
static int tx_px(const Av1oFrame *f, int plane, int pass, int r, int c) {
  int code = plane == 0 ? f->m_txsize[r * f->mi_stride + c] : f->m_bsize[r * f->mi_stride + c];
  if (plane && code == BS_64) code = TX_32X32;                 /* the chroma transforms of a 64x64 block are 32x32 (spec get_tx_size) */
  return 1 << (pass == 0 ? dim_wl(code) : dim_hl(code));
}

static void filter_edge_sample(uint16_t *px, int step /* distance between p/q samples */, int filter_size, int plane, int lvl, int sharp, int bd) {
  /* px points at q0; p_i = px[-(i+1)*step], q_i = px[i*step] */
  const int sh = sharp > 4 ? 2 : (sharp > 0 ? 1 : 0);
  const int limit = sharp > 0 ? iclamp(lvl >> sh, 1, 9 - sharp) : imax(1, lvl >> sh);
  const int blimit = 2 * (lvl + 2) + limit, thresh = lvl >> 4;
  const int s8 = bd - 8;
  const int limit_bd = limit << s8, blimit_bd = blimit << s8, thresh_bd = thresh << s8, flat_bd = 1 << s8;
  #define P(i) ((int)px[-((i) + 1) * step])
  #define Q(i) ((int)px[(i) * step])
  const int p0 = P(0), p1 = P(1), q0 = Q(0), q1 = Q(1);
  const int hev = iabs(p1 - p0) > thresh_bd || iabs(q1 - q0) > thresh_bd;
  const int flen = filter_size == 4 ? 4 : (plane != 0 ? 6 : (filter_size == 8 ? 8 : 16));
  int mask = iabs(p1 - p0) <= limit_bd && iabs(q1 - q0) <= limit_bd && (iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2) <= blimit_bd;
  if (flen >= 6) mask = mask && iabs(P(2) - p1) <= limit_bd && iabs(Q(2) - q1) <= limit_bd;
  if (flen >= 8) mask = mask && iabs(P(3) - P(2)) <= limit_bd && iabs(Q(3) - Q(2)) <= limit_bd;
  if (!mask) return;
  int flat = 0, flat2 = 0;
  if (filter_size >= 8) {
    flat = iabs(p1 - p0) <= flat_bd && iabs(q1 - q0) <= flat_bd && iabs(P(2) - p0) <= flat_bd && iabs(Q(2) - q0) <= flat_bd;
    if (flen >= 8) flat = flat && iabs(P(3) - p0) <= flat_bd && iabs(Q(3) - q0) <= flat_bd;
  }
  if (filter_size >= 16) {
    flat2 = iabs(P(6) - p0) <= flat_bd && iabs(Q(6) - q0) <= flat_bd && iabs(P(5) - p0) <= flat_bd && iabs(Q(5) - q0) <= flat_bd &&
            iabs(P(4) - p0) <= flat_bd && iabs(Q(4) - q0) <= flat_bd;
  }
  if (filter_size == 4 || !flat) {
    const int lo = -(1 << (bd - 1)), hi = (1 << (bd - 1)) - 1, off = 0x80 << s8;
    int ps1 = p1 - off, ps0 = p0 - off, qs0 = q0 - off, qs1 = q1 - off;
    int filt = hev ? iclamp(ps1 - qs1, lo, hi) : 0;
    filt = iclamp(filt + 3 * (qs0 - ps0), lo, hi);
    int f1 = iclamp(filt + 4, lo, hi) >> 3, f2 = iclamp(filt + 3, lo, hi) >> 3;
    px[0] = (uint16_t)(iclamp(qs0 - f1, lo, hi) + off);
    px[-step] = (uint16_t)(iclamp(ps0 + f2, lo, hi) + off);
    if (!hev) {
      int f = round2(f1, 1);
      px[step] = (uint16_t)(iclamp(qs1 - f, lo, hi) + off);
      px[-2 * step] = (uint16_t)(iclamp(ps1 + f, lo, hi) + off);
    }
    return;
  }
  const int log2size = (filter_size == 8 || !flat2) ? 3 : 4;
  const int n = log2size == 4 ? 6 : (plane == 0 ? 3 : 2);
  const int n2 = (log2size == 3 && plane == 0) ? 0 : 1;
  int F[16], out[16];                               /* index i+8 for i in -8..7 */
  for (int i = -(n + 1); i <= n; i++) F[i + 8] = i < 0 ? P(-i - 1) : Q(i);
  for (int i = -n; i < n; i++) {
    int t = 0;
    for (int j = -n; j <= n; j++) { int p = iclamp(i + j, -(n + 1), n); t += F[p + 8] * (iabs(j) <= n2 ? 2 : 1); }
    out[i + 8] = round2(t, log2size);
  }
  for (int i = -n; i < n; i++) { if (i < 0) px[i * step] = (uint16_t)out[i + 8]; else px[i * step] = (uint16_t)out[i + 8]; }
  #undef P
  #undef Q
}

/* Geometry of one candidate edge position: returns the filter size (0 = no edge here). */
static int edge_fsz(const Av1oFrame *f, int plane, int pass, int r, int c) {
  const int x = c * 4, y = r * 4;
  if (x >= f->w || y >= f->h) return 0;
  if (pass == 0 && c == 0) return 0;
  if (pass == 1 && r == 0) return 0;
  const int cur = imin(64, tx_px(f, plane, pass, r, c));
  if (pass == 0 ? (x % cur) != 0 : (y % cur) != 0) return 0;   /* tx (== block) edge? origins are aligned to their size */
  const int prev = pass == 0 ? imin(64, tx_px(f, plane, pass, r, c - 1)) : imin(64, tx_px(f, plane, pass, r - 1, c));
  const int base = imin(cur, prev);
  return plane == 0 ? imin(16, base) : imin(8, base);
}

/* ---- level search (rav1e deblock.rs deblock_filter_optimize with fast_deblock == false, [UPSTREAM-RECALL]):
 * every edge line is judged on the UNFILTERED reconstruction, independently of the other edges.  With sharpness 0 the
 * filter of one line is a piecewise-constant function of the level L: off below the smallest level that satisfies the
 * masks (limit = L, blimit = 3L + 4), and within [Lmin, 63] it can only change where thresh = L >> 4 changes (hev).  So one
 * line contributes at most four (range of levels, SSE delta) pairs, accumulated as a difference array per (plane, pass);
 * prefix sums give the frame SSE change for every level, the minimum picks the level (lowest level on ties).
 * Luma: vertical and horizontal edges get their own level; chroma: one level per plane over both passes. */
static void deblock_tally_line(const Av1oFrame *f, int plane, const uint16_t *rec, const uint16_t *src, int step, int fsz, int64_t *diff /* [65] */) {
  const int s8 = f->bd - 8, half = fsz == 4 ? 2 : (fsz == 8 ? 4 : 8), one = 1 << s8;
  int R[16];
  for (int i = -half; i < half; i++) R[i + 8] = rec[i * step];
  #define RP(i) R[7 - (i)]
  #define RQ(i) R[8 + (i)]
  const int flen = fsz == 4 ? 4 : (plane != 0 ? 6 : (fsz == 8 ? 8 : 16));
  int dmax = imax(iabs(RP(1) - RP(0)), iabs(RQ(1) - RQ(0)));
  if (flen >= 6) dmax = imax(dmax, imax(iabs(RP(2) - RP(1)), iabs(RQ(2) - RQ(1))));
  if (flen >= 8) dmax = imax(dmax, imax(iabs(RP(3) - RP(2)), iabs(RQ(3) - RQ(2))));
  const int b = iabs(RP(0) - RQ(0)) * 2 + iabs(RP(1) - RQ(1)) / 2;
  #undef RP
  #undef RQ
  const int B = (b + one - 1) >> s8;
  const int lmin = imax(1, imax((dmax + one - 1) >> s8, B > 4 ? (B - 4 + 2) / 3 : 0));
  if (lmin > 63) return;
  int64_t sse0 = 0;
  for (int i = -half; i < half; i++) { const int d = R[i + 8] - (int)src[i * step]; sse0 += d * d; }
  for (int a = lmin; a < 64; a = (a | 15) + 1) {
    const int bnd = imin(64, (a | 15) + 1);
    uint16_t t[16];
    for (int i = 0; i < 16; i++) t[i] = (uint16_t)((i >= 8 - half && i < 8 + half) ? R[i] : 0);
    filter_edge_sample(t + 8, 1, fsz, plane, a, 0, f->bd);
    int64_t sse = 0;
    for (int i = -half; i < half; i++) { const int d = (int)t[i + 8] - (int)src[i * step]; sse += d * d; }
    diff[a] += sse - sse0; diff[bnd] -= sse - sse0;
  }
}
void av1o_deblock_search(Av1oFrame *f, int64_t tally[3][2][64]) {
  static __thread int64_t diff[3][2][65];
  memset(diff, 0, sizeof(diff));
  for (int plane = 0; plane < f->np; plane++) for (int pass = 0; pass < 2; pass++)
    for (int r = 0; r < f->mi_rows; r++) for (int c = 0; c < f->mi_cols; c++) {
      const int fsz = edge_fsz(f, plane, pass, r, c);
      if (!fsz) continue;
      for (int i = 0; i < 4; i++) {
        const size_t o = pass == 0 ? (size_t)(r * 4 + i) * f->stride + c * 4 : (size_t)(r * 4) * f->stride + c * 4 + i;
        deblock_tally_line(f, plane, f->rec[plane] + o, f->src[plane] + o, pass == 0 ? 1 : f->stride, fsz, diff[plane][pass]);
      }
    }
  for (int plane = 0; plane < 3; plane++) for (int pass = 0; pass < 2; pass++) {
    int64_t acc = 0;
    for (int l = 0; l < 64; l++) { acc += diff[plane][pass][l]; tally[plane][pass][l] = acc; }
  }
}
/* AV1O_SELFCHECK=1: the tallies against a brute-force evaluation (the real filter on every line at every level) */
static void deblock_selfcheck(const Av1oFrame *f, int64_t tally[3][2][64]) {
  for (int plane = 0; plane < f->np; plane++) for (int pass = 0; pass < 2; pass++) for (int l = 0; l < 64; l++) {
    int64_t acc = 0;
    for (int r = 0; r < f->mi_rows; r++) for (int c = 0; c < f->mi_cols; c++) {
      const int fsz = edge_fsz(f, plane, pass, r, c);
      if (!fsz || !l) continue;
      for (int i = 0; i < 4; i++) {
        const size_t o = pass == 0 ? (size_t)(r * 4 + i) * f->stride + c * 4 : (size_t)(r * 4) * f->stride + c * 4 + i;
        const int step = pass == 0 ? 1 : f->stride;
        uint16_t t[16];
        for (int k = -8; k < 8; k++) t[k + 8] = (k >= -(fsz / 2 < 2 ? 2 : (fsz == 8 ? 4 : (fsz == 4 ? 2 : 8))) && k < (fsz == 4 ? 2 : (fsz == 8 ? 4 : 8))) ? f->rec[plane][o + k * step] : 0;
        uint16_t u[16]; memcpy(u, t, sizeof(t));
        filter_edge_sample(u + 8, 1, fsz, plane, l, 0, f->bd);
        for (int k = -8; k < 8; k++) if (u[k + 8] != t[k + 8]) {
          const int s = f->src[plane][o + k * step], d0 = (int)t[k + 8] - s, d1 = (int)u[k + 8] - s;
          acc += (int64_t)d1 * d1 - (int64_t)d0 * d0;
        }
      }
    }
    if (acc != tally[plane][pass][l]) { fprintf(stderr, "av1o deblock selfcheck: plane %d pass %d level %d: %lld vs %lld\n", plane, pass, l, (long long)acc, (long long)tally[plane][pass][l]); abort(); }
  }
}
static void deblock_pick_levels(Av1oFrame *f) {
  int64_t tally[3][2][64];
  av1o_deblock_search(f, tally);
  if (getenv("AV1O_SELFCHECK")) deblock_selfcheck(f, tally);
  for (int pass = 0; pass < 2; pass++) {
    int best = 0;
    for (int l = 1; l < 64; l++) if (tally[0][pass][l] < tally[0][pass][best]) best = l;
    f->lf_level[pass] = best;
  }
  for (int plane = 1; plane < 3; plane++) {
    int best = 0;
    if (plane < f->np) for (int l = 1; l < 64; l++) if (tally[plane][0][l] + tally[plane][1][l] < tally[plane][0][best] + tally[plane][1][best]) best = l;
    f->lf_level[plane + 1] = best;
  }
  if (!f->lf_level[0] && !f->lf_level[1]) f->lf_level[2] = f->lf_level[3] = 0;   /* spec 5.9.11: chroma levels are not coded then */
}

void av1o_deblock_frame(Av1oFrame *f) {
  f->lf_sharp = 0;
  if (f->cfg.fast_deblock) {
    /* libaom/rav1e q-based guess (rav1e deblock_filter_optimize, fast_deblock path, key frame) */
    const int q = f->ac_q[0];
    int lvl;
    if (f->bd == 8) lvl = (q * 17563 - 421574 + (1 << 17)) >> 18;
    else lvl = ((q * 20723 + 4060632 + (1 << 19)) >> 20) - 4;
    lvl = iclamp(lvl, 0, 63);
    f->lf_level[0] = f->lf_level[1] = lvl; f->lf_level[2] = f->lf_level[3] = lvl;
  } else deblock_pick_levels(f);
  if (!f->lf_level[0] && !f->lf_level[1]) return;
  for (int plane = 0; plane < f->np; plane++) {
    for (int pass = 0; pass < 2; pass++) {
      const int L = plane == 0 ? f->lf_level[pass] : f->lf_level[plane + 1];
      if (!L) continue;
      for (int r = 0; r < f->mi_rows; r++) for (int c = 0; c < f->mi_cols; c++) {
        const int fsz = edge_fsz(f, plane, pass, r, c);
        if (!fsz) continue;
        const int x = c * 4, y = r * 4;
        for (int i = 0; i < 4; i++) {
          uint16_t *px = pass == 0 ? f->rec[plane] + (y + i) * f->stride + x : f->rec[plane] + y * f->stride + x + i;
          filter_edge_sample(px, pass == 0 ? 1 : f->stride, fsz, plane, L, f->lf_sharp, f->bd);
        }
      }
    }
  }
}

/* ------------------------------------------------------------------ CDEF */
static const int8_t cdef_dirs[8][2][2] = { { { -1, 1 }, { -2, 2 } }, { { 0, 1 }, { -1, 2 } }, { { 0, 1 }, { 0, 2 } }, { { 0, 1 }, { 1, 2 } },
                                           { { 1, 1 }, { 2, 2 } }, { { 1, 0 }, { 2, 1 } }, { { 1, 0 }, { 2, 0 } }, { { 1, 0 }, { 2, -1 } } };
static int floor_log2(unsigned v) { return 31 - __builtin_clz(v); }
static int cdef_direction(const uint16_t *img, int stride, int bd, int *var) {
  static const int div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };
  int cost[8] = { 0 }, partial[8][15]; memset(partial, 0, sizeof(partial));
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) {
    int x = (img[i * stride + j] >> (bd - 8)) - 128;
    partial[0][i + j] += x; partial[1][i + j / 2] += x; partial[2][i] += x; partial[3][3 + i - j / 2] += x;
    partial[4][7 + i - j] += x; partial[5][3 - i / 2 + j] += x; partial[6][j] += x; partial[7][i / 2 + j] += x;
  }
  for (int i = 0; i < 8; i++) { cost[2] += partial[2][i] * partial[2][i]; cost[6] += partial[6][i] * partial[6][i]; }
  cost[2] *= div_table[8]; cost[6] *= div_table[8];
  for (int i = 0; i < 7; i++) {
    cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
    cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
  }
  cost[0] += partial[0][7] * partial[0][7] * div_table[8]; cost[4] += partial[4][7] * partial[4][7] * div_table[8];
  for (int i = 1; i < 8; i += 2) {
    for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
    cost[i] *= div_table[8];
    for (int j = 0; j < 3; j++) cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
  }
  int best = 0, dir = 0;
  for (int d = 0; d < 8; d++) if (cost[d] > best) { best = cost[d]; dir = d; }
  *var = (best - cost[(dir + 4) & 7]) >> 10;
  return dir;
}
static inline int constrain(int diff, int thr, int damping) {
  if (!thr) return 0;
  int adj = imax(0, damping - floor_log2((unsigned)thr)), mag = iabs(diff);
  int v = iclamp(thr - (mag >> adj), 0, mag);
  return diff < 0 ? -v : v;
}
/* filter one 8x8 block of `plane` from deblocked frame `in` into `out` (both frame-sized, same stride) */
static void cdef_filter8(const Av1oFrame *f, const uint16_t *in, uint16_t *out, int x0, int y0, int pri, int sec, int damping, int dir) {
  static const int pri_taps[2][2] = { { 4, 2 }, { 3, 3 } }, sec_taps[2][2] = { { 2, 1 }, { 2, 1 } };
  const int cs = f->bd - 8, st = f->stride, fw = f->mi_cols * 4, fh = f->mi_rows * 4;
  for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) {
    const int x = in[(y0 + i) * st + x0 + j];
    int sum = 0, mx = x, mn = x;
    for (int k = 0; k < 2; k++) for (int sgn = -1; sgn <= 1; sgn += 2) {
      int yy = y0 + i + sgn * cdef_dirs[dir][k][0], xx = x0 + j + sgn * cdef_dirs[dir][k][1];
      if (yy >= 0 && yy < fh && xx >= 0 && xx < fw) {
        int p = in[yy * st + xx];
        sum += pri_taps[(pri >> cs) & 1][k] * constrain(p - x, pri, damping);
        mx = imax(mx, p); mn = imin(mn, p);
      }
      for (int doff = -2; doff <= 2; doff += 4) {
        const int d2 = (dir + doff) & 7;
        yy = y0 + i + sgn * cdef_dirs[d2][k][0]; xx = x0 + j + sgn * cdef_dirs[d2][k][1];
        if (yy >= 0 && yy < fh && xx >= 0 && xx < fw) {
          int s = in[yy * st + xx];
          sum += sec_taps[(pri >> cs) & 1][k] * constrain(s - x, sec, damping);
          mx = imax(mx, s); mn = imin(mn, s);
        }
      }
    }
    out[(y0 + i) * st + x0 + j] = (uint16_t)iclamp(x + ((8 + sum - (sum < 0)) >> 4), mn, mx);
  }
}

void av1o_cdef_search_and_apply(Av1oFrame *f) {
  /* rav1e FrameInvariants: cdef_damping 3, cdef_bits 3, fixed strength list (pri*4 + sec code) */
  static const int strengths[8] = { 0, 1 * 4 + 0, 2 * 4 + 1, 3 * 4 + 1, 5 * 4 + 2, 7 * 4 + 3, 10 * 4 + 3, 13 * 4 + 3 };
  f->cdef_damping = 3; f->cdef_bits = 3;
  for (int i = 0; i < 8; i++) { f->cdef_y[i] = strengths[i]; f->cdef_uv[i] = strengths[i]; }
  for (int i = 0; i < f->sb_rows * f->sb_cols; i++) f->cdef_idx[i] = -1;
  const size_t npx = (size_t)f->pw * f->ph;
  uint16_t *in[3], *tmp[3];
  /* the deblocked frame is kept: loop restoration reads it across stripe boundaries (spec 7.17.6) */
  for (int p = 0; p < f->np; p++) { in[p] = (uint16_t *)malloc(npx * 2); memcpy(in[p], f->rec[p], npx * 2); f->dbk[p] = in[p]; }
  if (!f->enable_cdef) return;
  for (int p = 0; p < f->np; p++) tmp[p] = (uint16_t *)malloc(npx * 2);
  const int cs = f->bd - 8, ms = f->mi_stride;
  int64_t wq[3];
  for (int p = 0; p < f->np; p++) wq[p] = (((int64_t)f->ac_q[0] * f->ac_q[0]) << 12) / ((int64_t)f->ac_q[p] * f->ac_q[p]);
  for (int sr = 0; sr < f->sb_rows; sr++) for (int sc = 0; sc < f->sb_cols; sc++) {
    /* list the 8x8 blocks that are filtered at all */
    int nb = 0, bl[64][2], dirs[64], vars[64];
    for (int r = sr * 16; r < imin(sr * 16 + 16, f->mi_rows); r += 2) for (int c = sc * 16; c < imin(sc * 16 + 16, f->mi_cols); c += 2) {
      int sk = f->m_skip[r * ms + c] && f->m_skip[(r + 1) * ms + c] && f->m_skip[r * ms + c + 1] && f->m_skip[(r + 1) * ms + c + 1];
      if (sk) continue;
      bl[nb][0] = r; bl[nb][1] = c;
      dirs[nb] = cdef_direction(in[0] + (r * 4) * f->stride + c * 4, f->stride, f->bd, &vars[nb]);
      nb++;
    }
    if (!nb) continue;                       /* cdef_idx stays -1: nothing coded, nothing filtered */
    int best = 0; int64_t best_cost = INT64_MAX;
    for (int idx = 0; idx < 8; idx++) {
      int64_t cost = 0;
      for (int b = 0; b < nb; b++) {
        const int x0 = bl[b][1] * 4, y0 = bl[b][0] * 4;
        for (int p = 0; p < f->np; p++) {
          const int st = p == 0 ? f->cdef_y[idx] : f->cdef_uv[idx];
          int pri = (st >> 2) << cs, sec = (st & 3); if (sec == 3) sec = 4; sec <<= cs;
          int dir = pri == 0 ? 0 : dirs[b], damping = f->cdef_damping + cs - (p > 0);
          if (p == 0) { int var = vars[b]; int vs = (var >> 6) ? imin(floor_log2((unsigned)(var >> 6)), 12) : 0; pri = var ? (pri * (4 + vs) + 8) >> 4 : 0; }
          if (pri == 0 && sec == 0) { for (int i = 0; i < 8; i++) memcpy(tmp[p] + (y0 + i) * f->stride + x0, in[p] + (y0 + i) * f->stride + x0, 16); }
          else cdef_filter8(f, in[p], tmp[p], x0, y0, pri, sec, damping, dir);
          int64_t sse = 0;
          if (p == 0) sse = av1o_psy_dist_luma(f, tmp[p] + (size_t)y0 * f->stride + x0, f->stride, x0, y0, 8);
          else {
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { int d = (int)tmp[p][(y0 + i) * f->stride + x0 + j] - (int)f->src[p][(y0 + i) * f->stride + x0 + j]; sse += d * d; }
            sse = (sse * f->act[(y0 >> 3) * (f->pw / 8) + (x0 >> 3)] + 8192) >> 14;
          }
          cost += (sse * wq[p]) >> 5;
        }
      }
      if (cost < best_cost) { best_cost = cost; best = idx; }
    }
    f->cdef_idx[sr * f->sb_cols + sc] = (int8_t)best;
    for (int b = 0; b < nb; b++) {
      const int x0 = bl[b][1] * 4, y0 = bl[b][0] * 4;
      for (int p = 0; p < f->np; p++) {
        const int st = p == 0 ? f->cdef_y[best] : f->cdef_uv[best];
        int pri = (st >> 2) << cs, sec = (st & 3); if (sec == 3) sec = 4; sec <<= cs;
        int dir = pri == 0 ? 0 : dirs[b], damping = f->cdef_damping + cs - (p > 0);
        if (p == 0) { int var = vars[b]; int vs = (var >> 6) ? imin(floor_log2((unsigned)(var >> 6)), 12) : 0; pri = var ? (pri * (4 + vs) + 8) >> 4 : 0; }
        if (pri == 0 && sec == 0) continue;
        cdef_filter8(f, in[p], f->rec[p], x0, y0, pri, sec, damping, dir);
      }
    }
  }
  for (int p = 0; p < f->np; p++) free(tmp[p]);
}

/* oracle/av1o_txfm.c -- 2-D transforms, quantize, dequantize.  TEST INFRASTRUCTURE (see av1o.h).
 * Inverse path = AV1 spec 7.12.3 (dequant) + 7.13.3 (2-D inverse), normative.
 * Forward path + quantizer rounding = encoder-side: rav1e src/transform/forward.rs and
 * src/quantize/mod.rs are absent from /root/reference; the forward network is the transposed inverse
 * (tools/gen_txfm.py) with libaom/rav1e's stage shifts, the quantizer uses rav1e's dead-zone offsets as recalled. */
#include "av1o_int.h"
#include "txfm_gen.h"

typedef void (*tx1d_fn)(int32_t *);
static void ident4(int32_t *x) { for (int i = 0; i < 4; i++) x[i] = (int32_t)(((int64_t)x[i] * 5793 + 2048) >> 12); }
static void ident8(int32_t *x) { for (int i = 0; i < 8; i++) x[i] *= 2; }
static void ident16(int32_t *x) { for (int i = 0; i < 16; i++) x[i] = (int32_t)(((int64_t)x[i] * 11586 + 2048) >> 12); }
static void ident32(int32_t *x) { for (int i = 0; i < 32; i++) x[i] *= 4; }

/* kind: 0 dct, 1 adst, 2 identity */
static tx1d_fn pick(int fwd, int kind, int n) {
  if (kind == 2) return n == 4 ? ident4 : n == 8 ? ident8 : n == 16 ? ident16 : ident32;
  if (kind == 1) {
    if (fwd) return n == 4 ? av1_fadst4 : n == 8 ? av1_fadst8 : av1_fadst16;
    return n == 4 ? av1_iadst4 : n == 8 ? av1_iadst8 : av1_iadst16;
  }
  if (fwd) return n == 4 ? av1_fdct4 : n == 8 ? av1_fdct8 : n == 16 ? av1_fdct16 : n == 32 ? av1_fdct32 : av1_fdct64;
  return n == 4 ? av1_idct4 : n == 8 ? av1_idct8 : n == 16 ? av1_idct16 : n == 32 ? av1_idct32 : av1_idct64;
}
/* vertical (column) kind, horizontal (row) kind of a tx type */
static void tx_kinds(int t, int *col, int *row) {
  switch (t) {
    case DCT_DCT: *col = 0; *row = 0; break;
    case ADST_DCT: *col = 1; *row = 0; break;
    case DCT_ADST: *col = 0; *row = 1; break;
    case ADST_ADST: *col = 1; *row = 1; break;
    case IDTX: *col = 2; *row = 2; break;
    case V_DCT: *col = 0; *row = 2; break;
    case H_DCT: *col = 2; *row = 0; break;
    case V_ADST: *col = 1; *row = 2; break;
    case H_ADST: *col = 2; *row = 1; break;
    default: *col = 0; *row = 0; break;
  }
}
static inline int32_t rshift_round(int32_t v, int s) { return s <= 0 ? (int32_t)((uint32_t)v << -s) : (v + (1 << (s - 1))) >> s; }

/* w x h transform (txs code, av1o_int.h dim_wl / dim_hl); coefficients row-major in the coded area cw x ch = min(32, w) x min(32, h).
 * Rectangular 2:1 sizes (4x8, 8x4): libaom's stage shifts {2, -1, 0} and the sqrt(2) scale after the row pass forward, the spec's
 * 1/sqrt(2) scale (Round2(x * 2896, 12)) before the row pass and rowShift 0 inverse (7.13.3). */
void av1o_fwd_txfm2d(const int16_t *resid, int rstride, int32_t *coef, int txs, int txtype, int bd) {
  (void)bd;
  const int w = 1 << dim_wl(txs), h = 1 << dim_hl(txs), cw = imin(w, 32), ch = imin(h, 32);
  static const int8_t sh_sq[5][3] = { { 2, 0, 0 }, { 2, -1, 0 }, { 2, -2, 0 }, { 2, -4, 0 }, { 0, -2, -2 } };
  static const int8_t sh_rect[3] = { 2, -1, 0 };
  const int8_t *sh = dim_is_rect(txs) ? sh_rect : sh_sq[txs];
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  tx1d_fn colf = pick(1, ck, h), rowf = pick(1, rk, w);
  static __thread int32_t buf[64 * 64];
  int32_t t[64];
  for (int c = 0; c < w; c++) {
    for (int r = 0; r < h; r++) t[r] = rshift_round(resid[r * rstride + c], -sh[0]);
    colf(t);
    for (int r = 0; r < h; r++) buf[r * w + c] = rshift_round(t[r], -sh[1]);
  }
  for (int r = 0; r < ch; r++) {          /* rows >= 32 of a 64-point block are discarded */
    for (int c = 0; c < w; c++) t[c] = buf[r * w + c];
    rowf(t);
    for (int c = 0; c < cw; c++) {
      int32_t v = rshift_round(t[c], -sh[2]);
      if (dim_is_rect(txs)) v = (int32_t)(((int64_t)v * 5793 + 2048) >> 12);
      coef[r * cw + c] = v;
    }
  }
}

void av1o_inv_txfm2d_add(const int32_t *dq, uint16_t *dst, int dstride, int txs, int txtype, int bd) {
  const int w = 1 << dim_wl(txs), h = 1 << dim_hl(txs), cw = imin(w, 32), ch = imin(h, 32);
  static const int8_t row_shift_sq[5] = { 0, 1, 2, 2, 2 };
  const int row_shift = dim_is_rect(txs) ? 0 : row_shift_sq[txs];
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  tx1d_fn colf = pick(0, ck, h), rowf = pick(0, rk, w);
  static __thread int32_t res[64 * 64];
  int32_t t[64];
  const int rmax = (1 << (bd + 7)) - 1, rmin = -(1 << (bd + 7));
  const int cbits = imax(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
  for (int i = 0; i < h; i++) {
    if (i < ch) {
      for (int j = 0; j < w; j++) {
        int v = j < cw ? iclamp(dq[i * cw + j], rmin, rmax) : 0;
        if (dim_is_rect(txs)) v = round2(v * 2896, 12);
        t[j] = v;
      }
      rowf(t);
      for (int j = 0; j < w; j++) res[i * w + j] = iclamp(round2(t[j], row_shift), cmin, cmax);
    } else {
      for (int j = 0; j < w; j++) res[i * w + j] = 0;
    }
  }
  const int mx = (1 << bd) - 1;
  for (int j = 0; j < w; j++) {
    for (int i = 0; i < h; i++) t[i] = res[i * w + j];
    colf(t);
    for (int i = 0; i < h; i++) {
      int v = dst[i * dstride + j] + round2(t[i], 4);
      dst[i * dstride + j] = (uint16_t)iclamp(v, 0, mx);
    }
  }
}

/* rav1e QuantizationContext::quantize (recalled): dead-zone offsets 109/256 (DC, AC level>=1), 98/256 (AC level 0),
 * eob threshold (1 - 88/256) q; no trellis. */
int av1o_quantize(const int32_t *coef, int32_t *qc, int txs, int txtype, int dcq, int acq) {
  const int nc = imin(32, 1 << dim_wl(txs)) * imin(32, 1 << dim_hl(txs));
  const int ls = txs == TX_32X32 ? 1 : (txs == TX_64X64 ? 2 : 0);
  uint16_t tmp[1024];
  const uint16_t *scan = av1o_scan(txs, txtype, tmp);
  /* (AV1O_SWEEP_DZ=<percent>, oracle only: the recalled dead-zone offsets scaled, for the constants sweep of BASELINE.md section 5 -- tools/constants_sweep.py; unset = 100) */
  static int dz_pct = -1;
  if (dz_pct < 0) { const char *e = getenv("AV1O_SWEEP_DZ"); dz_pct = e ? atoi(e) : 100; }
  const int dc_off = dcq * 109 * dz_pct / 25600, off0 = acq * 98 * dz_pct / 25600, off1 = acq * 109 * dz_pct / 25600, off_eob = acq * 88 * dz_pct / 25600;
  memset(qc, 0, sizeof(int32_t) * (size_t)nc);
  int64_t a0 = (int64_t)iabs(coef[0]) << ls;
  int l0 = (int)((a0 + dc_off) / dcq);
  int eob = l0 ? 1 : 0;
  const int64_t thr = acq - off_eob;
  for (int i = nc - 1; i >= 1; i--) {
    if (((int64_t)iabs(coef[scan[i]]) << ls) >= thr) { eob = i + 1; break; }
  }
  if (eob == 0) return 0;
  qc[0] = coef[0] < 0 ? -l0 : l0;
  for (int i = 1; i < eob; i++) {
    int p = scan[i];
    int64_t a = (int64_t)iabs(coef[p]) << ls;
    int lv0 = (int)(a / acq);
    int off = lv0 > 0 ? off1 : off0;
    int lv = lv0 + ((a + off) >= (int64_t)(lv0 + 1) * acq);
    qc[p] = coef[p] < 0 ? -lv : lv;
  }
  return eob;
}

/* spec 7.12.3 */
void av1o_dequantize(const int32_t *qc, int32_t *dq, int txs, int dcq, int acq, int bd, int eob, int txtype) {
  (void)eob; (void)txtype;
  const int nc = imin(32, 1 << dim_wl(txs)) * imin(32, 1 << dim_hl(txs));
  const int sh = txs == TX_32X32 ? 1 : (txs == TX_64X64 ? 2 : 0);
  const int mx = (1 << (7 + bd)) - 1, mn = -(1 << (7 + bd));
  for (int i = 0; i < nc; i++) {
    int q = i == 0 ? dcq : acq;
    int64_t v = (int64_t)iabs(qc[i]) * q;
    v &= 0xFFFFFF;
    v >>= sh;
    if (qc[i] < 0) v = -v;
    dq[i] = (int32_t)(v < mn ? mn : (v > mx ? mx : v));
  }
}

// dev_txfm.h -- 2-D forward/inverse integer transforms, quantise and dequantise for one transform
// block, executed by one wavefront: lane c owns column c (then lane r owns row r) of the block held
// in LDS with a padded row pitch; each lane runs the straight-line 1-D butterfly network in registers.
// No MFMA: these are fixed integer butterflies (spec 7.13.2 for the inverse; the forward network is the
// transposed inverse, see tools/gen_txfm.py).  Quantiser rounding follows rav1e's dead-zone rule (recalled).
#pragma once
#include "dev_common.h"
#include "txfm_gen.hip.h"

template <int N> __device__ __forceinline__ void ident_fwd_inv(int32_t *x) {
  if (N == 4) { for (int i = 0; i < 4; i++) x[i] = (int32_t)(((long long)x[i] * 5793 + 2048) >> 12); }
  else if (N == 8) { for (int i = 0; i < 8; i++) x[i] *= 2; }
  else if (N == 16) { for (int i = 0; i < 16; i++) x[i] = (int32_t)(((long long)x[i] * 11586 + 2048) >> 12); }
  else { for (int i = 0; i < N; i++) x[i] *= 4; }
}
// kind: 0 dct, 1 adst, 2 identity
template <int N> __device__ __forceinline__ void tx1d(int32_t *x, int kind, bool fwd) {
  if (kind == 2) { ident_fwd_inv<N>(x); return; }
  if (N == 4) { if (kind == 1) { if (fwd) av1_fadst4(x); else av1_iadst4(x); } else { if (fwd) av1_fdct4(x); else av1_idct4(x); } }
  else if (N == 8) { if (kind == 1) { if (fwd) av1_fadst8(x); else av1_iadst8(x); } else { if (fwd) av1_fdct8(x); else av1_idct8(x); } }
  else if (N == 16) { if (kind == 1) { if (fwd) av1_fadst16(x); else av1_iadst16(x); } else { if (fwd) av1_fdct16(x); else av1_idct16(x); } }
  else if (N == 32) { if (fwd) av1_fdct32(x); else av1_idct32(x); }
  else { if (fwd) av1_fdct64(x); else av1_idct64(x); }
}
__device__ __forceinline__ void tx_kinds(int t, int *col, int *row) {
  switch (t) {
    case DCT_DCT: *col = 0; *row = 0; break;   case ADST_DCT: *col = 1; *row = 0; break;
    case DCT_ADST: *col = 0; *row = 1; break;  case ADST_ADST: *col = 1; *row = 1; break;
    case IDTX: *col = 2; *row = 2; break;      case V_DCT: *col = 0; *row = 2; break;
    case H_DCT: *col = 2; *row = 0; break;     case V_ADST: *col = 1; *row = 2; break;
    case H_ADST: *col = 2; *row = 1; break;    default: *col = 0; *row = 0; break;
  }
}
__device__ __forceinline__ int32_t rshift_round_(int32_t v, int s) { return s <= 0 ? (int32_t)((uint32_t)v << -s) : (v + (1 << (s - 1))) >> s; }

// tbuf: int32 [N][N+1] holding the residual on entry; coef out: [CS][CS], CS = min(N,32)
// (the code shape is pinned like the block searches', tile_search.h MI_K1_TRY_ATTR: the 2-D transforms of 16 points and more are functions of their own -- several
// call sites share them and the instruction cache holds one copy --, the small ones are inlined)
template <int N> __device__ __forceinline__ void fwd_txfm2d_body(LDS int32_t *tbuf, LDS int32_t *coef, int txtype) {
  constexpr int P = N + 1, CS = N < 32 ? N : 32;
  constexpr int TXS = N == 4 ? 0 : N == 8 ? 1 : N == 16 ? 2 : N == 32 ? 3 : 4;
  const int8_t sh[5][3] = { { 2, 0, 0 }, { 2, -1, 0 }, { 2, -2, 0 }, { 2, -4, 0 }, { 0, -2, -2 } };
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  for (int c = LANE; c < N; c += 64) {
    int32_t x[N];
#pragma unroll
    for (int r = 0; r < N; r++) x[r] = rshift_round_(tbuf[r * P + c], -sh[TXS][0]);
    tx1d<N>(x, ck, true);
#pragma unroll
    for (int r = 0; r < N; r++) tbuf[r * P + c] = rshift_round_(x[r], -sh[TXS][1]);
  }
  WAVE_SYNC();
  for (int r = LANE; r < CS; r += 64) {
    int32_t x[N];
#pragma unroll
    for (int c = 0; c < N; c++) x[c] = tbuf[r * P + c];
    tx1d<N>(x, rk, true);
#pragma unroll
    for (int c = 0; c < CS; c++) coef[r * CS + c] = rshift_round_(x[c], -sh[TXS][2]);
  }
  WAVE_SYNC();
}

template <int N> __device__ __attribute__((noinline)) void fwd_txfm2d_big(LDS int32_t *tbuf, LDS int32_t *coef, int txtype) { fwd_txfm2d_body<N>(tbuf, coef, txtype); }
template <int N> __device__ __forceinline__ void fwd_txfm2d_dev(LDS int32_t *tbuf, LDS int32_t *coef, int txtype) {
  if constexpr (N >= 16) fwd_txfm2d_big<N>(tbuf, coef, txtype); else fwd_txfm2d_body<N>(tbuf, coef, txtype);
}

// dq in: [CS][CS] dequantised; adds the residual to rec[N*N] (u16, pitch N) in place.
template <int N> __device__ __forceinline__ void inv_txfm2d_add_body(const LDS int32_t *dq, LDS int32_t *tbuf, LDS uint16_t *rec, int txtype, int bd) {
  constexpr int P = N + 1, CS = N < 32 ? N : 32;
  constexpr int ROWSH = N == 4 ? 0 : N == 8 ? 1 : 2;
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  const int rmax = (1 << (bd + 7)) - 1, rmin = -(1 << (bd + 7));
  const int cbits = imax_(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
  for (int i = LANE; i < N; i += 64) {
    int32_t x[N];
    if (i < CS) {
#pragma unroll
      for (int j = 0; j < N; j++) x[j] = j < CS ? dq[i * CS + j] : 0;      // already clamped to [rmin, rmax] by the dequantiser (same bounds)
      tx1d<N>(x, rk, false);
#pragma unroll
      for (int j = 0; j < N; j++) tbuf[i * P + j] = iclamp_(round2_(x[j], ROWSH), cmin, cmax);
    } else {
#pragma unroll
      for (int j = 0; j < N; j++) tbuf[i * P + j] = 0;
    }
  }
  WAVE_SYNC();
  const int mx = (1 << bd) - 1;
  for (int j = LANE; j < N; j += 64) {
    int32_t x[N];
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = tbuf[i * P + j];
    tx1d<N>(x, ck, false);
#pragma unroll
    for (int i = 0; i < N; i++) rec[i * N + j] = (uint16_t)iclamp_((int)rec[i * N + j] + round2_(x[i], 4), 0, mx);
  }
  WAVE_SYNC();
}

template <int N> __device__ __attribute__((noinline)) void inv_txfm2d_add_big(const LDS int32_t *dq, LDS int32_t *tbuf, LDS uint16_t *rec, int txtype, int bd) { inv_txfm2d_add_body<N>(dq, tbuf, rec, txtype, bd); }
template <int N> __device__ __forceinline__ void inv_txfm2d_add_dev(const LDS int32_t *dq, LDS int32_t *tbuf, LDS uint16_t *rec, int txtype, int bd) {
  if constexpr (N >= 16) inv_txfm2d_add_big<N>(dq, tbuf, rec, txtype, bd); else inv_txfm2d_add_body<N>(dq, tbuf, rec, txtype, bd);
}

// returns eob (wave-uniform); qc [CS*CS].  All magnitudes fit 32 bits (|coef| < 2^22, q < 2^13).
__device__ inline int quantize_dev(const LDS uint16_t *ls, const LDS int32_t *coef, LDS int32_t *qc, int n /*coded size*/, int txs, int txtype, int dcq, int acq,
                                   uint32_t dc_recip, uint32_t recip /* floor((2^32-1)/q) for the dc and ac step */) {
  const int nc = n * n, cls = tx_class_of(txtype);
  const int lsh = txs == 3 ? 1 : (txs == 4 ? 2 : 0);
  const uint32_t dc_off = (uint32_t)(dcq * 109 / 256), off0 = (uint32_t)(acq * 98 / 256), off1 = (uint32_t)(acq * 109 / 256), off_eob = (uint32_t)(acq * 88 / 256);
  const uint32_t thr = (uint32_t)acq - off_eob, uq = (uint32_t)acq;
  int last = 0;
  for (int i = LANE; i < nc; i += 64) {
    if (i >= 1) { const int p = scan_pos(ls, n, cls, i); if (((uint32_t)iabs_(coef[p]) << lsh) >= thr) last = i + 1; }
  }
  last = wave_max_i32(last);
  const uint32_t a0 = (uint32_t)iabs_(coef[0]) << lsh;
  const uint32_t x0 = a0 + dc_off;                      // x/q by multiply-high with floor((2^32-1)/q) and one fix-up: x/q - 1 < hi <= x/q for x < 2^31
  uint32_t l0u = __umulhi(x0, dc_recip);
  if (x0 - l0u * (uint32_t)dcq >= (uint32_t)dcq) l0u++;
  const int l0 = (int)l0u;
  int eob = last;
  if (eob == 0) eob = l0 ? 1 : 0;
  for (int i = LANE; i < nc; i += 64) {
    const int p = scan_pos(ls, n, cls, i);
    int v = 0;
    if (i < eob) {
      if (i == 0) v = coef[0] < 0 ? -l0 : l0;
      else {
        const uint32_t a = (uint32_t)iabs_(coef[p]) << lsh;
        uint32_t lv0 = __umulhi(a, recip);            // a/q - 1 < lv0 <= a/q for a < 2^31
        if (a - lv0 * uq >= uq) lv0++;
        const uint32_t off = lv0 > 0 ? off1 : off0;
        const int lv = (int)lv0 + ((a + off) >= (lv0 + 1) * uq);
        v = coef[p] < 0 ? -lv : lv;
      }
    }
    qc[p] = v;
  }
  WAVE_SYNC();
  return eob;
}

__device__ inline void dequantize_dev(const LDS int32_t *qc, LDS int32_t *dq, int n, int txs, int dcq, int acq, int bd) {
  const int nc = n * n, sh = txs == 3 ? 1 : (txs == 4 ? 2 : 0);
  const int mx = (1 << (7 + bd)) - 1, mn = -(1 << (7 + bd));
  for (int i = LANE; i < nc; i += 64) {
    const uint32_t q = (uint32_t)(i == 0 ? dcq : acq);
    uint32_t m = (uint32_t)iabs_(qc[i]) * q;        // < 2^32 for every level this encoder can emit
    m &= 0xFFFFFF; m >>= sh;
    const int v = qc[i] < 0 ? -(int)m : (int)m;
    dq[i] = v < mn ? mn : (v > mx ? mx : v);
  }
  WAVE_SYNC();
}

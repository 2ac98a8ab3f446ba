// dev_rate.h -- coefficient-coding contexts (spec 5.11.39 coeffs() + its CDF selection) evaluated in
// parallel: every lane prices the symbols of its own scan positions against the frame's static rate
// table and the wave sums the result.  rav1e prices candidates with a counting writer over adaptive CDFs
// (src/context/*.rs, absent from /root/reference); see DESIGN.md for the static-table divergence.
#pragma once
#include "dev_common.h"

// all_zero and dc_sign contexts from the level/dc maps left by the neighbours (inside the tile only)
template <typename FP, typename TP> __device__ inline void txb_ctx_dev(FP f, TP t, int plane, int r4, int c4, int txs, int bs, int *skip_ctx, int *dc_ctx) {
  const int w4 = 1 << txs, ms = f->mi_stride;
  int top = 0, left = 0, dcs = 0, any_a = 0, any_l = 0;
  const int k = LANE;
  if (k < w4) {
    // both neighbours with unconditional loads (clamped to the block's own cell where there is none): one round trip, not two
    const bool ha = r4 - 1 >= t->mi_row_start && c4 + k < f->mi_cols, hl = c4 - 1 >= t->mi_col_start && r4 + k < f->mi_rows;
    const int ia = ha ? (r4 - 1) * ms + c4 + k : r4 * ms + c4, il = hl ? (r4 + k) * ms + c4 - 1 : r4 * ms + c4;
    const int la = f->m_lvl[plane][ia], da = f->m_dc[plane][ia], ll = f->m_lvl[plane][il], dl = f->m_dc[plane][il];
    if (ha) { top = la; any_a = la | da; dcs += da == 1 ? -1 : (da == 2 ? 1 : 0); }
    if (hl) { left = ll; any_l = ll | dl; dcs += dl == 1 ? -1 : (dl == 2 ? 1 : 0); }
  }
  top = wave_max_i32(top); left = wave_max_i32(left); dcs = wave_sum_i32(dcs);
  any_a = wave_or_i32(any_a); any_l = wave_or_i32(any_l);
  *dc_ctx = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
  if (plane == 0) {
    int ctx;
    if (bs == txs) ctx = 0;
    else if (top == 0 && left == 0) ctx = 1;
    else if (top == 0 || left == 0) ctx = 2 + (imax_(top, left) > 3);
    else if (imax_(top, left) <= 3) ctx = 4;
    else if (imin_(top, left) <= 3) ctx = 5;
    else ctx = 6;
    *skip_ctx = ctx;
  } else {
    *skip_ctx = 7 + (any_a != 0) + (any_l != 0) + (bs > txs ? 3 : 0);
  }
}

__device__ __forceinline__ int eob_to_pt(int eob) { return eob < 3 ? eob : (32 - __clz(eob - 1) + 1); }
__device__ __forceinline__ int eob_pt_cdf(int eob_multi, int pt, int cls) {
  const int off[7] = { CDF_EOB_PT_16, CDF_EOB_PT_32, CDF_EOB_PT_64, CDF_EOB_PT_128, CDF_EOB_PT_256, CDF_EOB_PT_512, CDF_EOB_PT_1024 };
  const int str[7] = { CDF_EOB_PT_16_STRIDE, CDF_EOB_PT_32_STRIDE, CDF_EOB_PT_64_STRIDE, CDF_EOB_PT_128_STRIDE,
                       CDF_EOB_PT_256_STRIDE, CDF_EOB_PT_512_STRIDE, CDF_EOB_PT_1024_STRIDE };
  return off[eob_multi] + (pt * 2 + (cls == TXC_2D ? 0 : 1)) * str[eob_multi];
}
// context of coeff_base (not the eob position) from the padded level map; L points at this coefficient
__device__ __forceinline__ int base_ctx(const LDS uint8_t *L, int st, int cls, int row, int col) {
  int mag = imin_(L[1], 3) + imin_(L[st], 3);
  if (cls == TXC_2D) {
    mag += imin_(L[st + 1], 3) + imin_(L[2], 3) + imin_(L[2 * st], 3);
    const int m = imin_((mag + 1) >> 1, 4);
    if (row == 0 && col == 0) return 0;
    if (row + col < 2) return m + 1;
    if (row + col < 4) return m + 6;
    return m + 21;
  } else if (cls == TXC_VERT) {
    mag += imin_(L[2 * st], 3) + imin_(L[3 * st], 3) + imin_(L[4 * st], 3);
    const int m = imin_((mag + 1) >> 1, 4);
    return m + (row == 0 ? 26 : (row == 1 ? 31 : 36));
  }
  mag += imin_(L[2], 3) + imin_(L[3], 3) + imin_(L[4], 3);
  const int m = imin_((mag + 1) >> 1, 4);
  return m + (col == 0 ? 26 : (col == 1 ? 31 : 36));
}
__device__ __forceinline__ int br_ctx(const LDS uint8_t *L, int st, int cls, int row, int col, int c) {
  int mag = imin_(L[1], 15) + imin_(L[st], 15);
  if (cls == TXC_2D) { mag += imin_(L[st + 1], 15); mag = imin_((mag + 1) >> 1, 6); return c == 0 ? mag : ((row < 2 && col < 2) ? mag + 7 : mag + 14); }
  if (cls == TXC_HORIZ) { mag += imin_(L[2], 15); mag = imin_((mag + 1) >> 1, 6); return c == 0 ? mag : (col == 0 ? mag + 7 : mag + 14); }
  mag += imin_(L[2 * st], 15); mag = imin_((mag + 1) >> 1, 6); return c == 0 ? mag : (row == 0 ? mag + 7 : mag + 14);
}

// Fills the padded level map (LDS, (n+4)^2 bytes) from qc; all lanes.
__device__ inline void build_level_map(const LDS int32_t *qc, LDS uint8_t *lev, int n) {
  const int st = n + 4, words = (st * st + 3) >> 2, bwl = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5;
  LDS uint32_t *lw = (LDS uint32_t *)lev;                       // lev is 4-byte aligned and padded to a whole word
  for (int i = LANE; i < words; i += 64) lw[i] = 0;
  WAVE_SYNC();
  for (int i = LANE; i < n * n; i += 64) lev[(i >> bwl) * st + (i & (n - 1))] = (uint8_t)imin_(iabs_(qc[i]), 127);
  WAVE_SYNC();
}

// Rate (1/512 bit) of coeffs() for one transform block. tx_off >= 0: luma tx-type symbol is priced too.
// LDS-resident slices of the static rate table that coefficient coding touches (tx sizes 0..max only)
struct CoefCost { const LDS uint16_t *txb, *eobx, *dcs, *br, *base, *beob, *eobpt[4]; };
__device__ __forceinline__ int coef_cost_entries(int maxtxs) {
  const int t = maxtxs + 1, tb = imin_(maxtxs, 3) + 1;
  return t * 13 * CDF_TXB_SKIP_STRIDE + t * 18 * CDF_EOB_EXTRA_STRIDE + 6 * CDF_DC_SIGN_STRIDE + tb * 42 * CDF_COEFF_BR_STRIDE +
         t * 84 * CDF_COEFF_BASE_STRIDE + t * 8 * CDF_COEFF_BASE_EOB_STRIDE + 4 * (CDF_EOB_PT_16_STRIDE + CDF_EOB_PT_64_STRIDE + CDF_EOB_PT_256_STRIDE + CDF_EOB_PT_1024_STRIDE);
}
#define COEF_COST_MAX_ENTRIES(maxtxs) (((maxtxs) + 1) * (13 * 3 + 18 * 3 + 84 * 5 + 8 * 4) + 18 + ((maxtxs) < 3 ? (maxtxs) + 1 : 4) * 42 * 5 + 4 * (6 + 8 + 10 + 12))
// all threads of the workgroup copy; returns the table pointers
// Where each slice of the rate table sits behind `dst` (a pure function of dst and maxtxs: callers that know both at compile
// time get constant LDS addresses, nothing is kept in registers or on the stack), and where it comes from in the flat table.
__device__ __forceinline__ CoefCost coef_cost_layout(const LDS uint16_t *dst, int maxtxs, int *src_off = nullptr, int *count = nullptr) {
  const int t = maxtxs + 1, tb = imin_(maxtxs, 3) + 1;
  const int src[10] = { CDF_TXB_SKIP, CDF_EOB_EXTRA, CDF_DC_SIGN, CDF_COEFF_BR, CDF_COEFF_BASE, CDF_COEFF_BASE_EOB, CDF_EOB_PT_16, CDF_EOB_PT_64, CDF_EOB_PT_256, CDF_EOB_PT_1024 };
  const int cnt[10] = { t * 13 * CDF_TXB_SKIP_STRIDE, t * 18 * CDF_EOB_EXTRA_STRIDE, 6 * CDF_DC_SIGN_STRIDE, tb * 42 * CDF_COEFF_BR_STRIDE, t * 84 * CDF_COEFF_BASE_STRIDE,
                        t * 8 * CDF_COEFF_BASE_EOB_STRIDE, 4 * CDF_EOB_PT_16_STRIDE, 4 * CDF_EOB_PT_64_STRIDE, 4 * CDF_EOB_PT_256_STRIDE, 4 * CDF_EOB_PT_1024_STRIDE };
  int o[10]; int acc = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) { o[i] = acc; acc += cnt[i]; if (src_off) src_off[i] = src[i]; if (count) count[i] = cnt[i]; }
  CoefCost cc;
  cc.txb = dst + o[0]; cc.eobx = dst + o[1]; cc.dcs = dst + o[2]; cc.br = dst + o[3]; cc.base = dst + o[4]; cc.beob = dst + o[5];
  cc.eobpt[0] = dst + o[6]; cc.eobpt[1] = dst + o[7]; cc.eobpt[2] = dst + o[8]; cc.eobpt[3] = dst + o[9];
  return cc;
}
// all threads of the workgroup copy the slices
__device__ inline void load_coef_cost(LDS uint16_t *dst, const uint16_t *cost, int maxtxs, int tid, int nthreads) {
  int src[10], cnt[10];
  coef_cost_layout(dst, maxtxs, src, cnt);
  int o = 0;
  for (int k = 0; k < 10; k++) { for (int i = tid; i < cnt[k]; i += nthreads) dst[o + i] = cost[src[k] + i]; o += cnt[k]; }
}

template <typename CostPtr> __device__ inline uint32_t coef_rate_dev(const CoefCost &cc, CostPtr cost, const LDS uint16_t *ls, const LDS int32_t *qc, int eob, int plane, int txs, int txtype,
                                         int skip_ctx, int dc_ctx, int tx_off, int tx_sym, LDS uint8_t *lev, int *cul_out, int *dc_cat) {
  const int n = imin_(32, 4 << txs), bwl = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5;
  const int pt = plane > 0, cls = tx_class_of(txtype), txs_ctx = txs;
  *cul_out = 0; *dc_cat = 0;
  uint32_t head = cc.txb[(txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE + (eob == 0)];
  if (eob == 0) return head;
  if (tx_off >= 0) head += cost[tx_off + tx_sym];
  const int eob_pt = eob_to_pt(eob), eob_multi = 2 * bwl - 4;
  {
    static_assert(CDF_EOB_PT_16_STRIDE == 6 && CDF_EOB_PT_64_STRIDE == 8 && CDF_EOB_PT_256_STRIDE == 10 && CDF_EOB_PT_1024_STRIDE == 12, "stride = 6 + 2 sq");
    const int sq = eob_multi >> 1;                        // square transforms: 16 / 64 / 256 / 1024 coefficients
    head += cc.eobpt[sq][(pt * 2 + (cls == TXC_2D ? 0 : 1)) * (6 + 2 * sq) + eob_pt - 1];
  }
  if (eob_pt >= 3) {
    const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
    head += cc.eobx[((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE + hi];
    head += 512u * (uint32_t)(nb - 1);
  }
  build_level_map(qc, lev, n);
  const int st = n + 4, area = n * n;
  int bits = 0; int cul = 0, dcc = 0;
  for (int c = LANE; c < eob; c += 64) {
    const int p = scan_pos(ls, n, cls, c), row = p >> bwl, col = p & (n - 1);
    const int v = qc[p], level = iabs_(v);
    const LDS uint8_t *L = lev + row * st + col;
    if (c == eob - 1) {
      const int ctx = c == 0 ? 0 : (c <= area / 8 ? 1 : (c <= area / 4 ? 2 : 3));
      bits += cc.beob[((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE + imin_(level, 3) - 1];
    } else {
      const int ctx = base_ctx(L, st, cls, row, col);
      bits += cc.base[((txs_ctx * 2 + pt) * 42 + ctx) * CDF_COEFF_BASE_STRIDE + imin_(level, 3)];
    }
    if (level > 2) {
      const int ctx = br_ctx(L, st, cls, row, col, c);
      const int off = ((imin_(txs_ctx, 3) * 2 + pt) * 21 + ctx) * CDF_COEFF_BR_STRIDE;
      int rem = level - 3;
      for (int idx = 0; idx < 4; idx++) { const int s = imin_(rem, 3); bits += cc.br[off + s]; rem -= s; if (s < 3) break; }
    }
    if (level) {
      if (c == 0) { bits += cc.dcs[(pt * 3 + dc_ctx) * CDF_DC_SIGN_STRIDE + (v < 0)]; dcc = v < 0 ? 1 : 2; }
      else bits += 512;
      if (level > 14) { const int len = 32 - __clz(level - 14); bits += 512 * (2 * len - 1); }
    }
    cul += level;
  }
  bits = wave_sum_i32(bits);
  cul = wave_sum_i32(imin_(cul, 1 << 20));
  dcc = __builtin_amdgcn_readfirstlane(dcc);             // only scan position 0 (lane 0, first iteration) sets it
  *cul_out = imin_(cul, 63); *dc_cat = dcc;
  return head + (uint32_t)bits;
}

// Padded level maps, one region per coded size so that the zero padding written once at kernel start stays zero
// (the interior is fully rewritten by every evaluation of that size).
#define LEV_OFF(cs) ((cs) == 4 ? 0 : (cs) == 8 ? 64 : (cs) == 16 ? 64 + 144 : 64 + 144 + 400)
#define LEV_BYTES(maxcs) ((maxcs) == 4 ? 64 : (maxcs) == 8 ? 64 + 144 : (maxcs) == 16 ? 64 + 144 + 400 : 64 + 144 + 400 + 1296)

// Quantise + level map + dequantise + rate of one transform block in one sweep: each lane keeps the scan positions,
// magnitudes and levels of its coefficients in registers between the steps (quantize_dev + build_level_map +
// dequantize_dev + coef_rate_dev did four sweeps with LDS round trips in between; results are identical).
//   cbuf: in = forward coefficients [CS*CS], out = dequantised coefficients (only meaningful when eob > 0)
//   qc:   out = quantised levels, raster [CS*CS];  levbase: this wave's level-map regions (LEV_OFF)
// Returns eob (wave-uniform); *rate_out in 1/512 bit.
template <int CS, typename CostPtr>
__device__ inline int quant_rate_dev(const CoefCost &cc, CostPtr cost, const LDS uint16_t *ls, LDS int32_t *cbuf, LDS int32_t *qc, LDS uint8_t *levbase,
                                     int plane, int txs, int txtype, int dcq, int acq, uint32_t dc_recip, uint32_t ac_recip, int bd,
                                     int skip_ctx, int dc_ctx, int tx_off, int tx_sym, uint32_t *rate_out, int *cul_out, int *dc_cat) {
  constexpr int n = CS, nc = CS * CS, IT = nc < 64 ? 1 : nc / 64, bwl = CS == 4 ? 2 : CS == 8 ? 3 : CS == 16 ? 4 : 5, st = CS + 4;
  LDS uint8_t *lev = levbase + LEV_OFF(CS);
  const uint32_t tx_cost = cost[tx_off >= 0 ? tx_off + tx_sym : 0];     // the only global-memory operand: issued first, consumed after the quantiser
  const int cls = tx_class_of(txtype), pt = plane > 0, txs_ctx = txs;
  const int lsh = txs == 3 ? 1 : (txs == 4 ? 2 : 0);
  const uint32_t dc_off = (uint32_t)(dcq * 109 / 256), off0 = (uint32_t)(acq * 98 / 256), off1 = (uint32_t)(acq * 109 / 256), off_eob = (uint32_t)(acq * 88 / 256);
  const uint32_t thr = (uint32_t)acq - off_eob, uq = (uint32_t)acq;
  int pos[IT]; uint32_t mag[IT]; int neg[IT];
  int last = 0;
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int i = LANE + 64 * k;
    const bool valid = nc >= 64 || i < nc;
    pos[k] = valid ? scan_pos(ls, n, cls, i) : 0;
    const int c = valid ? cbuf[pos[k]] : 0;
    mag[k] = (uint32_t)iabs_(c) << lsh; neg[k] = c < 0;
    if (valid && i >= 1 && mag[k] >= thr) last = i + 1;
  }
  last = wave_max_i32(last);
  // dc level by every lane (uniform LDS read), as quantize_dev
  const uint32_t x0 = ((uint32_t)iabs_(cbuf[0]) << lsh) + dc_off;
  uint32_t l0u = __umulhi(x0, dc_recip);
  if (x0 - l0u * (uint32_t)dcq >= (uint32_t)dcq) l0u++;
  const int l0 = (int)l0u;
  int eob = last;
  if (eob == 0) eob = l0 ? 1 : 0;
  const int sh = lsh, dmx = (1 << (7 + bd)) - 1, dmn = -(1 << (7 + bd));
  int lvl[IT];
  WAVE_SYNC();                                   // every lane has read its coefficients before cbuf is overwritten
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int i = LANE + 64 * k;
    const bool valid = nc >= 64 || i < nc;
    int lv = 0;
    if (valid && i < eob) {
      if (i == 0) lv = l0;
      else {
        const uint32_t a = mag[k];
        uint32_t lv0 = __umulhi(a, ac_recip);          // a/q - 1 < lv0 <= a/q for a < 2^31
        if (a - lv0 * uq >= uq) lv0++;
        const uint32_t off = lv0 > 0 ? off1 : off0;
        lv = (int)lv0 + ((a + off) >= (lv0 + 1) * uq);
      }
    }
    lvl[k] = lv;
    if (valid) {
      const int pp = pos[k];
      qc[pp] = neg[k] ? -lv : lv;
      lev[(pp >> bwl) * st + (pp & (n - 1))] = (uint8_t)imin_(lv, 127);
      uint32_t m = (uint32_t)lv * (uint32_t)(pp == 0 ? dcq : acq);          // dequantize_dev
      m &= 0xFFFFFF; m >>= sh;
      const int v = neg[k] ? -(int)m : (int)m;
      cbuf[pp] = v < dmn ? dmn : (v > dmx ? dmx : v);
    }
  }
  WAVE_SYNC();
  *cul_out = 0; *dc_cat = 0;
  uint32_t head = cc.txb[(txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE + (eob == 0)];
  if (eob == 0) { *rate_out = head; return 0; }
  if (tx_off >= 0) head += tx_cost;
  const int eob_pt = eob_to_pt(eob), eob_multi = 2 * bwl - 4;
  {
    static_assert(CDF_EOB_PT_16_STRIDE == 6 && CDF_EOB_PT_64_STRIDE == 8 && CDF_EOB_PT_256_STRIDE == 10 && CDF_EOB_PT_1024_STRIDE == 12, "stride = 6 + 2 sq");
    const int sq = eob_multi >> 1;
    head += cc.eobpt[sq][(pt * 2 + (cls == TXC_2D ? 0 : 1)) * (6 + 2 * sq) + eob_pt - 1];
  }
  if (eob_pt >= 3) {
    const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
    head += cc.eobx[((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE + hi];
    head += 512u * (uint32_t)(nb - 1);
  }
  constexpr int area = nc;
  int bits = 0, cul = 0, dcc = 0;
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int c = LANE + 64 * k;
    if (c < eob) {
      const int pp = pos[k], row = pp >> bwl, col = pp & (n - 1), level = lvl[k];
      const LDS uint8_t *L = lev + row * st + col;
      if (c == eob - 1) {
        const int ctx = c == 0 ? 0 : (c <= area / 8 ? 1 : (c <= area / 4 ? 2 : 3));
        bits += cc.beob[((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE + imin_(level, 3) - 1];
      } else {
        const int ctx = base_ctx(L, st, cls, row, col);
        bits += cc.base[((txs_ctx * 2 + pt) * 42 + ctx) * CDF_COEFF_BASE_STRIDE + imin_(level, 3)];
      }
      if (level > 2) {
        const int ctx = br_ctx(L, st, cls, row, col, c);
        const int off = ((imin_(txs_ctx, 3) * 2 + pt) * 21 + ctx) * CDF_COEFF_BR_STRIDE;
        int rem = level - 3;
        for (int idx = 0; idx < 4; idx++) { const int s2 = imin_(rem, 3); bits += cc.br[off + s2]; rem -= s2; if (s2 < 3) break; }
      }
      if (level) {
        if (c == 0) { bits += cc.dcs[(pt * 3 + dc_ctx) * CDF_DC_SIGN_STRIDE + neg[k]]; dcc = neg[k] ? 1 : 2; }
        else bits += 512;
        if (level > 14) { const int len = 32 - __clz(level - 14); bits += 512 * (2 * len - 1); }
      }
      cul += level;
    }
  }
  bits = wave_sum_i32(bits);
  cul = wave_sum_i32(imin_(cul, 1 << 20));
  dcc = __builtin_amdgcn_readfirstlane(dcc);
  *cul_out = imin_(cul, 63); *dc_cat = dcc;
  *rate_out = head + (uint32_t)bits;
  return eob;
}

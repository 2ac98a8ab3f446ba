// dev_rect.h -- rectangular blocks (PARTITION_HORZ / PARTITION_VERT of an 8x8 node: 8x4 and 4x8 blocks with their 2:1 transforms) for the
// tile search K1; mirrors oracle/av1o_search.c try_block for the codes BS_4X8 = 5 / BS_8X4 = 6 (oracle/av1o_int.h).  The (mode x tx type) trials, the
// two halves of the tx-size trial and the chroma candidates run four per wavefront (eval_group_wh, one candidate per 16-lane row) like the square
// small blocks; the SATD stage and the full mode set of speed <= 1 keep one candidate per wavefront.
#pragma once
enum { BS_4X8 = 5, BS_8X4 = 6 };
__device__ __forceinline__ int dim_wl(int code) { return code <= 4 ? 2 + code : (code == 5 ? 2 : 3); }   // log2 width / height in samples
__device__ __forceinline__ int dim_hl(int code) { return code <= 4 ? 2 + code : (code == 5 ? 3 : 2); }

// spec Default_Scan_4x8 (tall: each anti-diagonal from its top-right end) / Default_Scan_8x4 (wide: from its bottom-left end)
static __device__ const uint8_t rect_scan_4x8[32] = { 0, 1, 4, 2, 5, 8, 3, 6, 9, 12, 7, 10, 13, 16, 11, 14, 17, 20, 15, 18, 21, 24, 19, 22, 25, 28, 23, 26, 29, 27, 30, 31 };
static __device__ const uint8_t rect_scan_8x4[32] = { 0, 8, 1, 16, 9, 2, 24, 17, 10, 3, 25, 18, 11, 4, 26, 19, 12, 5, 27, 20, 13, 6, 28, 21, 14, 7, 29, 22, 15, 30, 23, 31 };
__device__ __forceinline__ int rect_scan_pos(int wl, int hl, int cls, int i) {
  const int w = 1 << wl, h = 1 << hl;
  if (cls == TXC_2D) return hl > wl ? rect_scan_4x8[i] : rect_scan_8x4[i];
  if (cls == TXC_VERT) return i;                              // mrow
  const int c = i / h, r = i - c * h; return r * w + c;       // mcol
}

// raw edges of a w x h block: above[-1 .. w+h-1], left[-1 .. w+h-1] (oracle av1o_predict_intra_wh edge preparation)
__device__ inline void load_edges_wh(const LDS FrameDev *f, int plane, int x, int y, int w, int h, int have_left, int have_above, int have_ar, int have_bl,
                                     LDS uint16_t *above, LDS uint16_t *left) {
  const int bd = f->bd, rs = f->stride, tot = w + h;
  const uint16_t *rec = f->rec[plane];
  const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1;
  const int lim_a = imin_(max_x, x + (have_ar ? 2 * w : w) - 1), lim_l = imin_(max_y, y + (have_bl ? 2 * h : h) - 1);
  for (int i = LANE; i <= tot; i += 64) {
    const bool corner = i == tot;
    int ia, il;
    if (have_above) ia = (y - 1) * rs + (corner ? (have_left ? x - 1 : x) : imin_(lim_a, x + i));
    else ia = y * rs + (have_left ? x - 1 : x);
    if (have_left) il = imin_(lim_l, y + i) * rs + x - 1;
    else il = (have_above ? y - 1 : y) * rs + x;
    uint16_t a = rec[ia], l = rec[il];
    if (!have_above && !have_left) { a = (uint16_t)(corner ? (1 << (bd - 1)) : (1 << (bd - 1)) - 1); l = (uint16_t)((1 << (bd - 1)) + 1); }
    if (corner) { above[-1] = a; left[-1] = a; } else { above[i] = a; left[i] = l; }
  }
  WAVE_SYNC();
}

// w x h prediction (dev_predict.h predict_block with both dimensions)
__device__ inline void predict_block_wh(const LDS FrameDev *f, int x, int y, int wl, int hl, int have_left, int have_above, int mode, int angle_delta, int ftype,
                                        const LDS uint16_t *ra, const LDS uint16_t *rl, LDS uint16_t *wa, LDS uint16_t *wl_, LDS uint16_t *tmp, LDS uint16_t *pred) {
  const int w = 1 << wl, h = 1 << hl, bd = f->bd, nn = w * h;
  const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1;
  if (mode == PAETH_PRED) {
    const int tl = ra[-1];
    for (int idx = LANE; idx < nn; idx += 64) {
      const int i = idx >> wl, j = idx & (w - 1);
      const int base = ra[j] + rl[i] - tl;
      const int pl = iabs_(base - rl[i]), pt = iabs_(base - ra[j]), ptl = iabs_(base - tl);
      pred[idx] = (pl <= pt && pl <= ptl) ? rl[i] : (pt <= ptl ? ra[j] : (uint16_t)tl);
    }
  } else if (mode == DC_PRED) {
    int v;
    if (have_left || have_above) {
      int s = 0;
      for (int k = LANE; k < imax_(w, h); k += 64) s += ((have_above && k < w) ? ra[k] : 0) + ((have_left && k < h) ? rl[k] : 0);
      s = wave_sum_i32(s);
      if (have_left && have_above) v = (s + ((w + h) >> 1)) / (w + h);
      else if (have_left) v = (s + (h >> 1)) >> hl;
      else v = (s + (w >> 1)) >> wl;
    } else v = 1 << (bd - 1);
    for (int idx = LANE; idx < nn; idx += 64) pred[idx] = (uint16_t)v;
  } else if (mode == SMOOTH_PRED || mode == SMOOTH_V_PRED || mode == SMOOTH_H_PRED) {
    const uint8_t *sw = sm_weights_dev(wl), *sh = sm_weights_dev(hl);
    const int bl = rl[h - 1], tr = ra[w - 1];
    for (int idx = LANE; idx < nn; idx += 64) {
      const int i = idx >> wl, j = idx & (w - 1);
      int p;
      if (mode == SMOOTH_PRED) p = round2_(sh[i] * ra[j] + (256 - sh[i]) * bl + sw[j] * rl[i] + (256 - sw[j]) * tr, 9);
      else if (mode == SMOOTH_V_PRED) p = round2_(sh[i] * ra[j] + (256 - sh[i]) * bl, 8);
      else p = round2_(sw[j] * rl[i] + (256 - sw[j]) * tr, 8);
      pred[idx] = (uint16_t)p;
    }
  } else {
    const int pa = mode_angle_of(mode) + angle_delta * 3;
    for (int i = LANE; i < w + h + 1; i += 64) { wa[i - 1] = ra[i - 1]; wl_[i - 1] = rl[i - 1]; }
    WAVE_SYNC();
    int up_a = 0, up_l = 0;
    if (pa != 90 && pa != 180) {
      if (pa > 90 && pa < 180 && w + h >= 24) {
        if (LANE == 0) { const int v = round2_(wl_[0] * 5 + wa[-1] * 6 + wa[0] * 5, 4); wa[-1] = (uint16_t)v; wl_[-1] = (uint16_t)v; }
        WAVE_SYNC();
      }
      if (have_above) edge_filter_dev(wa, imin_(w, max_x - x + 1) + (pa < 90 ? h : 0) + 1, edge_strength_dev(w, h, ftype, pa - 90), tmp);
      if (have_left) edge_filter_dev(wl_, imin_(h, max_y - y + 1) + (pa > 180 ? w : 0) + 1, edge_strength_dev(w, h, ftype, pa - 180), tmp);
    }
    up_a = edge_upsample_sel_dev(w, h, ftype, pa - 90);
    if (up_a) edge_upsample_dev(wa, w + (pa < 90 ? h : 0), bd, tmp);
    up_l = edge_upsample_sel_dev(w, h, ftype, pa - 180);
    if (up_l) edge_upsample_dev(wl_, h + (pa > 180 ? w : 0), bd, tmp);
    int dx = 0, dy = 0;
    if (pa < 90) dx = dr_deriv_dev(pa); else if (pa > 90 && pa < 180) dx = dr_deriv_dev(180 - pa);
    if (pa > 90 && pa < 180) dy = dr_deriv_dev(pa - 90); else if (pa > 180) dy = dr_deriv_dev(270 - pa);
    for (int idx = LANE; idx < nn; idx += 64) {
      const int i = idx >> wl, j = idx & (w - 1);
      int v;
      if (pa < 90) {
        const int max_base = (w + h - 1) << up_a;
        const int id = (i + 1) * dx, base = (id >> (6 - up_a)) + (j << up_a), sh = ((id << up_a) >> 1) & 0x1F;
        v = base < max_base ? round2_(wa[base] * (32 - sh) + wa[base + 1] * sh, 5) : wa[max_base];
      } else if (pa > 90 && pa < 180) {
        int id = (j << 6) - (i + 1) * dx, base = id >> (6 - up_a);
        if (base >= -(1 << up_a)) { const int sh = ((id << up_a) >> 1) & 0x1F; v = round2_(wa[base] * (32 - sh) + wa[base + 1] * sh, 5); }
        else { id = (i << 6) - (j + 1) * dy; base = id >> (6 - up_l); const int sh = ((id << up_l) >> 1) & 0x1F; v = round2_(wl_[base] * (32 - sh) + wl_[base + 1] * sh, 5); }
      } else if (pa > 180) {
        const int id = (j + 1) * dy, base = (id >> (6 - up_l)) + (i << up_l), sh = ((id << up_l) >> 1) & 0x1F;
        v = round2_(wl_[base] * (32 - sh) + wl_[base + 1] * sh, 5);
      } else if (pa == 90) v = wa[j];
      else v = wl_[i];
      pred[idx] = (uint16_t)v;
    }
  }
  WAVE_SYNC();
}

// 4x4-Hadamard SATD of a w x h block (satd_dev with both dimensions; pitch w)
__device__ inline long long satd_wh(const LDS uint16_t *src, const LDS uint16_t *pred, int w, int h) {
  const int units = w * (h >> 2);
  int total = 0;
  for (int u = LANE; u < units; u += 64) {
    const int x = u % w, o = (u / w) * 4 * w + x;
    const int d0 = (int)src[o] - (int)pred[o], d1 = (int)src[o + w] - (int)pred[o + w];
    const int d2 = (int)src[o + 2 * w] - (int)pred[o + 2 * w], d3 = (int)src[o + 3 * w] - (int)pred[o + 3 * w];
    const int a = d0 + d1, b = d0 - d1, c = d2 + d3, e = d2 - d3;
    int t[4] = { a + c, b + e, a - c, b - e };
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { int v = satd_quad_step<0xB1>(t[i], 1); v = satd_quad_step<0x4E>(v, 2); s += iabs_(v); }
    total += s;
  }
  return (long long)wave_sum_i32(total);
}

// 2:1 transforms, W x H in {4x8, 8x4}: libaom's forward stage shifts {2, -1, 0} with the sqrt(2) scale after the row pass, the spec's inverse
// (7.13.3: Round2(x * 2896, 12) before the row pass, rowShift 0).  tbuf: int32 [H][W + 1]; coef / dq: [H][W].
template <int W, int H> __device__ inline void fwd_txfm_rect(LDS int32_t *tbuf, LDS int32_t *coef, int txtype) {
  constexpr int P = W + 1;
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  for (int c = LANE; c < W; c += 64) {
    int32_t x[H];
#pragma unroll
    for (int r = 0; r < H; r++) x[r] = tbuf[r * P + c] << 2;
    tx1d<H>(x, ck, true);
#pragma unroll
    for (int r = 0; r < H; r++) tbuf[r * P + c] = rshift_round_(x[r], 1);
  }
  WAVE_SYNC();
  for (int r = LANE; r < H; r += 64) {
    int32_t x[W];
#pragma unroll
    for (int c = 0; c < W; c++) x[c] = tbuf[r * P + c];
    tx1d<W>(x, rk, true);
#pragma unroll
    for (int c = 0; c < W; c++) coef[r * W + c] = (int32_t)(((long long)x[c] * 5793 + 2048) >> 12);
  }
  WAVE_SYNC();
}
template <int W, int H> __device__ inline void inv_txfm_rect_add(const LDS int32_t *dq, LDS int32_t *tbuf, LDS uint16_t *rec, int txtype, int bd) {
  constexpr int P = W + 1;
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  const int cbits = imax_(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
  for (int i = LANE; i < H; i += 64) {
    int32_t x[W];
#pragma unroll
    for (int j = 0; j < W; j++) x[j] = round2_(dq[i * W + j] * 2896, 12);      // dq already clamped to 8 + bd bits by the dequantiser
    tx1d<W>(x, rk, false);
#pragma unroll
    for (int j = 0; j < W; j++) tbuf[i * P + j] = iclamp_(x[j], cmin, cmax);
  }
  WAVE_SYNC();
  const int mx = (1 << bd) - 1;
  for (int j = LANE; j < W; j += 64) {
    int32_t x[H];
#pragma unroll
    for (int i = 0; i < H; i++) x[i] = tbuf[i * P + j];
    tx1d<H>(x, ck, false);
#pragma unroll
    for (int i = 0; i < H; i++) rec[i * W + j] = (uint16_t)iclamp_((int)rec[i * W + j] + round2_(x[i], 4), 0, mx);
  }
  WAVE_SYNC();
}

// tx set / CDF row of a 2:1 transform (oracle av1o_tx_set / av1o_intra_tx_cdf): the larger dimension (8) bounds the set, the smaller (4) indexes the CDFs
template <int TS, typename FP> __device__ __forceinline__ int rect_tx_cdf(FP f, int ymode, int *nsyms, int *set_out) {
  const int set = Tools<TS>::reduced_tx_set(f) ? 2 : 1;
  *set_out = set;
  if (f->base_q_idx == 0) { *nsyms = 0; return -1; }
  if (set == 1) { *nsyms = 7; return CDF_INTRA_TX1 + ymode * CDF_INTRA_TX1_STRIDE; }
  *nsyms = 5; return CDF_INTRA_TX2 + ymode * CDF_INTRA_TX2_STRIDE;
}

// Quantise (rav1e dead-zone rule as quantize_dev), level map, rate (oracle av1o_code_coeffs priced against the static table, one scan position per
// lane), dequantise in place.  cbuf: coefficients in, dequantised out; qc: levels out; lev: (w + 4) x (h + 4) bytes.  Returns eob.
template <typename CostPtr>
__device__ inline int rect_quant_rate(CostPtr cost, LDS int32_t *cbuf, LDS int32_t *qc, LDS uint8_t *lev, int plane, int wl, int hl, int txtype, int dcq, int acq,
                                      uint32_t dc_recip, uint32_t ac_recip, int bd, int skip_ctx, int dc_ctx, int tx_off, int tx_sym, uint32_t *rate_out, int *cul_out, int *dc_cat) {
  const int w = 1 << wl, h = 1 << hl, nc = w * h, st = w + 4;
  const int cls = tx_class_of(txtype), pt = plane > 0, txs_ctx = 1;          // (sqr + sqr_up + 1) >> 1 of 4x8 / 8x4
  const uint32_t dc_off = (uint32_t)(dcq * 109 / 256), off0 = (uint32_t)(acq * 98 / 256), off1 = (uint32_t)(acq * 109 / 256), off_eob = (uint32_t)(acq * 88 / 256);
  const uint32_t thr = (uint32_t)acq - off_eob, uq = (uint32_t)acq;
  const int i = LANE, valid = i < nc;
  const int pos = valid ? rect_scan_pos(wl, hl, cls, i) : 0;
  const int cf = valid ? cbuf[pos] : 0;
  const uint32_t mag = (uint32_t)iabs_(cf); const int neg = cf < 0;
  int last = (valid && i >= 1 && mag >= thr) ? i + 1 : 0;
  last = wave_max_i32(last);
  const uint32_t x0 = (uint32_t)iabs_(cbuf[0]) + dc_off;
  uint32_t l0u = __umulhi(x0, dc_recip);
  if (x0 - l0u * (uint32_t)dcq >= (uint32_t)dcq) l0u++;
  const int l0 = (int)l0u;
  int eob = last;
  if (eob == 0) eob = l0 ? 1 : 0;
  WAVE_SYNC();
  int lv = 0;
  if (valid && i < eob) {
    if (i == 0) lv = l0;
    else {
      uint32_t lv0 = __umulhi(mag, ac_recip);
      if (mag - lv0 * uq >= uq) lv0++;
      const uint32_t off = lv0 > 0 ? off1 : off0;
      lv = (int)lv0 + ((mag + off) >= (lv0 + 1) * uq);
    }
  }
  // level map with a 4-wide zero border
  for (int q = LANE; q < st * (h + 4); q += 64) lev[q] = 0;
  WAVE_SYNC();
  const int dmx = (1 << (7 + bd)) - 1, dmn = -(1 << (7 + bd));
  if (valid) {
    qc[pos] = neg ? -lv : lv;
    lev[(pos >> wl) * st + (pos & (w - 1))] = (uint8_t)imin_(lv, 127);
    uint32_t m = (uint32_t)lv * (uint32_t)(pos == 0 ? dcq : acq);
    m &= 0xFFFFFF;
    const int v = neg ? -(int)m : (int)m;
    cbuf[pos] = v < dmn ? dmn : (v > dmx ? dmx : v);
  }
  WAVE_SYNC();
  *cul_out = 0; *dc_cat = 0;
  uint32_t head = cost[CDF_TXB_SKIP + (txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE + (eob == 0)];
  if (eob == 0) { *rate_out = head; for (int q = LANE; q < st * (h + 4); q += 64) lev[q] = 0; WAVE_SYNC(); return 0; }
  if (tx_off >= 0) head += cost[tx_off + tx_sym];
  const int eob_pt = eob_to_pt(eob);
  head += cost[eob_pt_cdf(1, pt, cls) + eob_pt - 1];                          // eob_multi = log2(32) - 4 = 1: the 32-coefficient table
  if (eob_pt >= 3) {
    const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
    head += cost[CDF_EOB_EXTRA + ((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE + hi] + 512u * (uint32_t)(nb - 1);
  }
  uint32_t mine = 0;
  if (valid && i < eob) {
    const int row = pos >> wl, col = pos & (w - 1), level = lv;
    const LDS uint8_t *L = lev + row * st + col;
    if (i == eob - 1) {
      const int ctx = i == 0 ? 0 : (i <= nc / 8 ? 1 : (i <= nc / 4 ? 2 : 3));
      mine += cost[CDF_COEFF_BASE_EOB + ((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE + imin_(level, 3) - 1];
    } else {
      int ctx = base_ctx(L, st, cls, row, col);
      if (cls == TXC_2D && !(row == 0 && col == 0)) {                          // spec Coeff_Base_Ctx_Offset of the 2:1 sizes
        int mg = imin_(L[1], 3) + imin_(L[st], 3) + imin_(L[st + 1], 3) + imin_(L[2], 3) + imin_(L[2 * st], 3);
        const int m = imin_((mg + 1) >> 1, 4);
        if (hl > wl) ctx = m + (row < 2 ? 11 : (row + col < 4 ? 6 : 21));
        else ctx = m + (col < 2 ? 16 : (row + col < 4 ? 6 : 21));
      }
      mine += cost[CDF_COEFF_BASE + ((txs_ctx * 2 + pt) * 42 + ctx) * CDF_COEFF_BASE_STRIDE + imin_(level, 3)];
    }
    if (level > 2) {
      const int off = CDF_COEFF_BR + ((txs_ctx * 2 + pt) * 21 + br_ctx(L, st, cls, row, col, i)) * CDF_COEFF_BR_STRIDE;
      int rem = level - 3;
      for (int idx = 0; idx < 4; idx++) { const int s = imin_(rem, 3); mine += cost[off + s]; rem -= s; if (s < 3) break; }
    }
    if (level) {
      if (i == 0) mine += cost[CDF_DC_SIGN + (pt * 3 + dc_ctx) * CDF_DC_SIGN_STRIDE + neg];
      else mine += 512u;
      if (level > 14) { const uint32_t xg = (uint32_t)(level - 14); const int len = 32 - __clz(xg); mine += 512u * (uint32_t)(2 * len - 1); }
    }
  }
  const int cul = imin_(wave_sum_i32((valid && i < eob) ? lv : 0), 63);
  const int dcneg = __builtin_amdgcn_readlane(neg, 0), dclv = __builtin_amdgcn_readlane(lv, 0);
  *cul_out = cul; *dc_cat = dclv ? (dcneg ? 1 : 2) : 0;
  *rate_out = head + (uint32_t)wave_sum_i32((int)mine);
  // the square sizes keep their level maps' borders zero for the whole tile (dev_rate.h LEV_OFF: 4x4 at 0, 8x8 at 64): leave this region as found
  for (int q = LANE; q < st * (h + 4); q += 64) lev[q] = 0;
  WAVE_SYNC();
  return eob;
}

// One 2:1 transform block by one wave (eval_tx for the rectangular sizes).  psv2 / act: the two 4x4 cells' source variances and the activity of the
// 8x8 cell the block lies in (luma, Tune::Psychovisual); cact: the same activity for chroma.
template <int MAXN, int WL, int HL, int NW, int TS>
__device__ inline long long eval_rect(const Ctx<MAXN, NW, TS> k, int plane, int sctx, int dctx, const LDS uint16_t *src, const LDS uint16_t *pred, int txtype, int tx_off, int tx_sym,
                                      LDS uint16_t *rec_out, LDS int32_t *qc_out, TxRes *tr, int psv_a, int psv_b, int act) {
  constexpr int W = 1 << WL, H = 1 << HL, P = W + 1, NN = W * H;
  const LDS FrameDev *f = k.f(); LDS WaveScratch<MAXN> *S = k.s();
  for (int idx = LANE; idx < NN; idx += 64) { S->tbuf[(idx >> WL) * P + (idx & (W - 1))] = (int)src[idx] - (int)pred[idx]; rec_out[idx] = pred[idx]; }
  WAVE_SYNC();
  fwd_txfm_rect<W, H>(S->tbuf, S->cbuf, txtype);
  const int eob = rect_quant_rate(k.cost(), S->cbuf, qc_out, S->lev, plane, WL, HL, txtype, f->dc_q[plane], f->ac_q[plane], f->dc_recip[plane], f->ac_recip[plane], f->bd,
                                  sctx, dctx, tx_off, tx_sym, &tr->rate, &tr->cul, &tr->dcc);
  if (eob > 0) inv_txfm_rect_add<W, H>(S->cbuf, S->tbuf, rec_out, txtype, f->bd);
  tr->eob = eob;
  // distortion: luma = the two 4x4 cells priced like psy_dist_wave (boost x activity), chroma = SSE x activity
  int sd[2] = { 0, 0 }, qd[2] = { 0, 0 }, se[2] = { 0, 0 };
  if (LANE < NN) {
    const int i = LANE >> WL, j = LANE & (W - 1), cell = W == 8 ? (j >> 2) : (i >> 2);
    const int d = rec_out[LANE], e = (int)src[LANE] - d;
    sd[cell] = d; qd[cell] = d * d; se[cell] = e * e;
  }
  long long dist;
  if (plane == 0 && !Tools<TS>::tune_psnr(f)) {
    dist = 0;
#pragma unroll
    for (int cell = 0; cell < 2; cell++) {
      const uint32_t s1 = (uint32_t)wave_sum_i32(sd[cell]), s2 = (uint32_t)wave_sum_i32(qd[cell]), s3 = (uint32_t)wave_sum_i32(se[cell]);
      dist += psy_cell_dist(s3, s1, s2, (uint32_t)(cell ? psv_b : psv_a), (uint32_t)act, 4, f->bd);
    }
  } else {
    const long long e = (long long)wave_sum_i32(se[0] + se[1]);
    dist = plane == 0 ? e : (e * act + 8192) >> 14;
  }
  tr->sse = dist;
  return ((dist * f->wq[plane]) >> 5) + (((long long)tr->rate * f->rdmult + 256) >> 9);
}

// eval_group() (dev_group.h) for the 2:1 sizes: four candidates per wavefront, one per 16-lane row, two coefficients per lane.  Same arithmetic and
// rounding points as eval_rect() / rect_quant_rate() above.  The rate of the eob class comes from the 32-coefficient table, which is not among the
// LDS-resident slices: the six entries of its row are fetched up front (lane k of the row holds entry k) and picked after the quantiser.
template <int WL, int HL, typename CostPtr>
__device__ inline void eval_group_wh(const CoefCost &cc, CostPtr cost, const LDS FrameDev *f, LDS GroupBuf8 *gb, const LDS uint16_t *src, const LDS uint16_t *pred, int plane, int txtype,
                                     int skip_ctx, int dc_ctx, int tx_off, int tx_sym, int psv_a, int psv_b, int psy_act, GroupRes *res) {
  constexpr int W = 1 << WL, H = 1 << HL, P = W + 1, nc = 32, IT = 2, st = W + 4, txs_ctx = 1;
  const int gl = GROUP_LANE, bd = f->bd;
  const int cls = tx_class_of(txtype), pt = plane > 0;
  const uint32_t tx_cost = cost[tx_off >= 0 ? tx_off + tx_sym : 0];
  const uint32_t ep_cost = cost[eob_pt_cdf(1, pt, cls) + imin_(gl, 5)];
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int idx = gl + 16 * k, i = idx >> WL, j = idx & (W - 1);
    const int pv = pred[idx];
    gb->tbuf[i * P + j] = (int)src[idx] - pv;
    gb->rec[idx] = (uint16_t)pv;
  }
  for (int i = gl; i < (st * (H + 4) + 3) / 4; i += 16) ((LDS uint32_t *)gb->lev)[i] = 0;
  WAVE_SYNC();
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  if (gl < W) {                                              // columns: H-point
    int32_t x[H];
#pragma unroll
    for (int r = 0; r < H; r++) x[r] = (int32_t)((uint32_t)gb->tbuf[r * P + gl] << 2);
    tx1d<H>(x, ck, true);
#pragma unroll
    for (int r = 0; r < H; r++) gb->tbuf[r * P + gl] = rshift_round_(x[r], 1);
  }
  WAVE_SYNC();
  if (gl < H) {                                              // rows: W-point, then the sqrt(2) scale of the 2:1 sizes
    int32_t x[W];
#pragma unroll
    for (int c = 0; c < W; c++) x[c] = gb->tbuf[gl * P + c];
    tx1d<W>(x, rk, true);
#pragma unroll
    for (int c = 0; c < W; c++) gb->cbuf[gl * W + c] = (int32_t)(((long long)x[c] * 5793 + 2048) >> 12);
  }
  WAVE_SYNC();
  const int dcq = f->dc_q[plane], acq = f->ac_q[plane];
  const uint32_t dc_recip = f->dc_recip[plane], ac_recip = f->ac_recip[plane];
  const uint32_t dc_off = (uint32_t)(dcq * 109 / 256), off0 = (uint32_t)(acq * 98 / 256), off1 = (uint32_t)(acq * 109 / 256), off_eob = (uint32_t)(acq * 88 / 256);
  const uint32_t thr = (uint32_t)acq - off_eob, uq = (uint32_t)acq;
  int pos[IT]; uint32_t mag[IT]; int neg[IT];
  int last = 0;
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int i = gl + 16 * k;
    pos[k] = rect_scan_pos(WL, HL, cls, i);
    const int c = gb->cbuf[pos[k]];
    mag[k] = (uint32_t)iabs_(c); neg[k] = c < 0;
    if (i >= 1 && mag[k] >= thr) last = i + 1;
  }
  last = row_max_i32(last);
  const uint32_t x0 = (uint32_t)iabs_(gb->cbuf[0]) + dc_off;
  uint32_t l0u = __umulhi(x0, dc_recip);
  if (x0 - l0u * (uint32_t)dcq >= (uint32_t)dcq) l0u++;
  const int l0 = (int)l0u;
  int eob = last;
  if (eob == 0) eob = l0 ? 1 : 0;
  const int dmx = (1 << (7 + bd)) - 1, dmn = -(1 << (7 + bd));
  int lvl[IT];
  WAVE_SYNC();
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int i = gl + 16 * k;
    int lv = 0;
    if (i < eob) {
      if (i == 0) lv = l0;
      else {
        const uint32_t a = mag[k];
        uint32_t lv0 = __umulhi(a, ac_recip);
        if (a - lv0 * uq >= uq) lv0++;
        const uint32_t off = lv0 > 0 ? off1 : off0;
        lv = (int)lv0 + ((a + off) >= (lv0 + 1) * uq);
      }
    }
    lvl[k] = lv;
    const int pp = pos[k];
    gb->qc[pp] = neg[k] ? -lv : lv;
    gb->lev[(pp >> WL) * st + (pp & (W - 1))] = (uint8_t)imin_(lv, 127);
    uint32_t m = (uint32_t)lv * (uint32_t)(pp == 0 ? dcq : acq);
    m &= 0xFFFFFF;
    const int v = neg[k] ? -(int)m : (int)m;
    gb->cbuf[pp] = v < dmn ? dmn : (v > dmx ? dmx : v);
  }
  WAVE_SYNC();
  uint32_t head = cc.txb[(txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE + (eob == 0)];
  int bits = 0, cul = 0, dcc = 0;
  const int eob_pt = eob_to_pt(imax_(eob, 1));
  const uint32_t ep = (uint32_t)__shfl((int)ep_cost, (LANE & 48) + eob_pt - 1);       // every row picks its own class's entry
  if (eob > 0) {
    if (tx_off >= 0) head += tx_cost;
    head += ep;
    if (eob_pt >= 3) {
      const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
      head += cc.eobx[((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE + hi];
      head += 512u * (uint32_t)(nb - 1);
    }
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const int c = gl + 16 * k;
      if (c < eob) {
        const int pp = pos[k], row = pp >> WL, col = pp & (W - 1), level = lvl[k];
        const LDS uint8_t *L = gb->lev + row * st + col;
        if (c == eob - 1) {
          const int ctx = c == 0 ? 0 : (c <= nc / 8 ? 1 : (c <= nc / 4 ? 2 : 3));
          bits += cc.beob[((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE + imin_(level, 3) - 1];
        } else {
          int ctx = base_ctx(L, st, cls, row, col);
          if (cls == TXC_2D && !(row == 0 && col == 0)) {                          // spec Coeff_Base_Ctx_Offset of the 2:1 sizes
            const int mg = imin_(L[1], 3) + imin_(L[st], 3) + imin_(L[st + 1], 3) + imin_(L[2], 3) + imin_(L[2 * st], 3), m = imin_((mg + 1) >> 1, 4);
            ctx = HL > WL ? m + (row < 2 ? 11 : (row + col < 4 ? 6 : 21)) : m + (col < 2 ? 16 : (row + col < 4 ? 6 : 21));
          }
          bits += cc.base[((txs_ctx * 2 + pt) * 42 + ctx) * CDF_COEFF_BASE_STRIDE + imin_(level, 3)];
        }
        if (level > 2) {
          const int off = ((txs_ctx * 2 + pt) * 21 + br_ctx(L, st, cls, row, col, c)) * CDF_COEFF_BR_STRIDE;
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
  }
  bits = row_sum_i32(bits);
  cul = row_sum_i32(imin_(cul, 1 << 20));
  dcc = row_max_i32(dcc);
  res->eob = eob; res->cul = imin_(cul, 63); res->dcc = dcc; res->rate = head + (uint32_t)bits;
  {
    const int cbits = imax_(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
    const bool act_r = gl < H && eob > 0, act_c = gl < W && eob > 0;
    if (act_r) {
      int32_t x[W];
#pragma unroll
      for (int j = 0; j < W; j++) x[j] = round2_(gb->cbuf[gl * W + j] * 2896, 12);
      tx1d<W>(x, rk, false);
#pragma unroll
      for (int j = 0; j < W; j++) gb->tbuf[gl * P + j] = iclamp_(x[j], cmin, cmax);
    }
    WAVE_SYNC();
    const int mx = (1 << bd) - 1;
    if (act_c) {
      int32_t x[H];
#pragma unroll
      for (int i = 0; i < H; i++) x[i] = gb->tbuf[i * P + gl];
      tx1d<H>(x, ck, false);
#pragma unroll
      for (int i = 0; i < H; i++) gb->rec[i * W + gl] = (uint16_t)iclamp_((int)gb->rec[i * W + gl] + round2_(x[i], 4), 0, mx);
    }
    WAVE_SYNC();
  }
  // distortion: luma = the two 4x4 cells priced like psy_dist_wave (boost x activity), chroma (psv_a < 0) = SSE x activity
  int se[2] = { 0, 0 }, sd[2] = { 0, 0 }, qd[2] = { 0, 0 };
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int idx = gl + 16 * k, i = idx >> WL, j = idx & (W - 1), cell = W == 8 ? (j >> 2) : (i >> 2);
    const int rv = gb->rec[idx], d = (int)src[idx] - rv;
    se[cell] += __mul24(d, d); sd[cell] += rv; qd[cell] += __mul24(rv, rv);
  }
  if (psv_a >= 0) {
    int dist = 0;
#pragma unroll
    for (int cell = 0; cell < 2; cell++)
      dist += psy_cell_dist((uint32_t)row_sum_i32(se[cell]), (uint32_t)row_sum_i32(sd[cell]), (uint32_t)row_sum_i32(qd[cell]), (uint32_t)(cell ? psv_b : psv_a), (uint32_t)psy_act, 4, bd);
    res->sse = dist;
  } else res->sse = (int)(((unsigned long long)(uint32_t)row_sum_i32(se[0] + se[1]) * (uint32_t)psy_act + 8192) >> 14);
}

// ---- the block search for an 8x4 / 4x8 block (oracle try_block with bs = BS_8X4 / BS_4X8): same candidates, same order, same tie-breaks ----
template <int WL, int HL> __device__ inline void commit_rect(const LDS FrameDev *f, int plane, int r, int c, const LDS uint16_t *rec, const LDS int32_t *qc, int eob, int cul, int dcc) {
  constexpr int W = 1 << WL, H = 1 << HL;
  uint16_t *gr = f->rec[plane] + (size_t)(r * 4) * f->stride + c * 4;
  int32_t *gc = f->coef[plane] + (size_t)(r * 4) * f->stride + c * 4;
  for (int idx = LANE; idx < W * H; idx += 64) { gr[(idx >> WL) * f->stride + (idx & (W - 1))] = rec[idx]; gc[(idx >> WL) * f->stride + (idx & (W - 1))] = qc[idx]; }
  if (LANE < 2) { const int rr = r + (H == 8 ? LANE : 0), cc = c + (W == 8 ? LANE : 0); f->m_lvl[plane][rr * f->mi_stride + cc] = (uint8_t)cul; f->m_dc[plane][rr * f->mi_stride + cc] = (uint8_t)dcc; }
  if (LANE == 0) f->m_eob[plane][r * f->mi_stride + c] = (uint16_t)eob;
}
// the block's two cells of a byte map
template <int WL, int HL> __device__ __forceinline__ void fill_rect(uint8_t *m, int ms, int r, int c, int v) {
  if (LANE < 2) m[(r + (HL == 3 ? LANE : 0)) * ms + c + (WL == 3 ? LANE : 0)] = (uint8_t)v;
}
// all_zero / dc_sign contexts of a transform block of w4 x h4 cells inside a block (txb_ctx_dev with both dimensions); `whole`: the transform is the block
template <typename FP, typename TP> __device__ inline void txb_ctx_wh(FP f, TP t, int plane, int r4, int c4, int w4, int h4, int whole, int *skip_ctx, int *dc_ctx) {
  const int ms = f->mi_stride;
  int top = 0, left = 0, dcs = 0, any_a = 0, any_l = 0;
  const int k = LANE;
  if (k < imax_(w4, h4)) {
    const bool ha = k < w4 && r4 - 1 >= t->mi_row_start && c4 + k < f->mi_cols, hl = k < h4 && c4 - 1 >= t->mi_col_start && r4 + k < f->mi_rows;
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
    if (whole) ctx = 0;
    else if (top == 0 && left == 0) ctx = 1;
    else if (top == 0 || left == 0) ctx = 2 + (imax_(top, left) > 3);
    else if (imax_(top, left) <= 3) ctx = 4;
    else if (imin_(top, left) <= 3) ctx = 5;
    else ctx = 6;
    *skip_ctx = ctx;
  } else *skip_ctx = 7 + (any_a != 0) + (any_l != 0) + (whole ? 0 : 3);
}

template <int MAXN, int BSR, int NW, int TS>
__device__ MI_K1_TRY_ATTR long long try_block_rect(const Ctx<MAXN, NW, TS> k, int r, int c, long long budget = J_INF) {
  static_assert(NW == 4 && MAXN <= 32, "the 2:1 block search deals its candidates to four wavefronts (its chroma stages hard-code the wave roles; no one-candidate-per-wave fallback is left)");
  constexpr int WL = BSR == BS_4X8 ? 2 : 3, HL = BSR == BS_4X8 ? 3 : 2, W_ = 1 << WL, H_ = 1 << HL, NN = W_ * H_, w4 = W_ >> 2, h4 = H_ >> 2;
  const LDS FrameDev *f = k.f(); const LDS TileB *t = k.t(); LDS WaveScratch<MAXN> *S = k.s(); LDS SharedScratch<MAXN> *SH = k.sh();
  const int W = NW > 1 ? WAVE_ID : 0;
  const int ms = f->mi_stride, mi = r * ms + c, x = c * 4, y = r * 4;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int can_ar = availU && (c + w4 < t->mi_col_end), can_bl = availL && (r + h4 < t->mi_row_end);
  const int iU = availU ? mi - ms : mi, iL = availL ? mi - 1 : mi;
  constexpr int SHAPE = BSR == BS_8X4 ? 1 : 2;                 // (decoded_before: the halves of the node in coding order)
  const int have_ar = can_ar && decoded_before(r, c, SHAPE, r - 1, c + w4), have_bl = can_bl && decoded_before(r, c, SHAPE, r + h4, c - 1);
  const int amode = availU ? uni32(f->m_ymode[iU]) : DC_PRED, lmode = availL ? uni32(f->m_ymode[iL]) : DC_PRED;
  const int uvU = f->np > 1 ? uni32(f->m_uvmode[iU]) : 0, uvL = f->np > 1 ? uni32(f->m_uvmode[iL]) : 0;
  const int v_skU = f->m_skip[iU], v_skL = f->m_skip[iL], v_skUL = f->m_skip[availU && availL ? mi - ms - 1 : mi];        // bit 0 = skip, the rest = segment id
  const int nb_skip = (availU ? uni32(v_skU) & 1 : 0) + (availL ? uni32(v_skL) & 1 : 0);
  const int seg_nb = (availU && availL ? (uni32(v_skUL) >> 1) + 1 : 0) | ((availU ? (uni32(v_skU) >> 1) + 1 : 0) << 4) | ((availL ? (uni32(v_skL) >> 1) + 1 : 0) << 8);   // as in try_block
  const int nb_txU = availU ? uni32(f->m_txsize[iU]) : -1, nb_txL = availL ? uni32(f->m_txsize[iL]) : -1;
  const uint16_t *ycost = k.cost() + CDF_KF_Y + (intra_mode_ctx(amode) * 5 + intra_mode_ctx(lmode)) * CDF_KF_Y_STRIDE;
  const int ftype_y = IS_SMOOTH_(amode) || IS_SMOOTH_(lmode);
  const int ftype_uv = f->np > 1 && ((availU && IS_SMOOTH_(uvU)) || (availL && IS_SMOOTH_(uvL)));
  LDS uint16_t *wa = S->wa + EDGE_OFF, *wl = S->wl + EDGE_OFF;
  PH_BEGIN();

  // ---- stage the source block, the raw edges and the transform contexts of every plane (plane p by wave p % NW), the psychovisual references
  for (int p = 0; p < f->np; p++) if (p % NW == W) {
    int sc_, dc_;
    txb_ctx_wh(f, t, p, r, c, w4, h4, 1, &sc_, &dc_);
    if (LANE == 0) { SH->sctx[p] = sc_; SH->dctx[p] = dc_; }
    const uint16_t *g = f->src[p] + (size_t)y * f->stride + x;
    for (int idx = LANE; idx < NN; idx += 64) SH->srcb[p][idx] = g[(idx >> WL) * f->stride + (idx & (W_ - 1))];
    load_edges_wh(f, p, x, y, W_, H_, availL, availU, have_ar, have_bl, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF);
  }
  if (W == NW - 1) {
    const int a = (int)f->act[(y >> 3) * (f->pw >> 3) + (x >> 3)];
    if (LANE < 2) SH->psv4[LANE] = (int)f->svar4[(r + (HL == 3 ? LANE : 0)) * ms + c + (WL == 3 ? LANE : 0)];
    if (LANE == 0) { SH->pact[0] = a; SH->cact = a; SH->seg_nb = seg_nb; }
    seg_select(f, SH, a);
  }
  PH(1);
  WG_SYNC();
  PH(2);
  const int sctx_y = SH->sctx[0], dctx_y = SH->dctx[0];
  const LDS uint16_t *ra = SH->ra[0] + EDGE_OFF, *rl = SH->rl[0] + EDGE_OFF;
  const int psv_a = Tools<TS>::tune_psnr(f) ? 0 : SH->psv4[0], psv_b = Tools<TS>::tune_psnr(f) ? 0 : SH->psv4[1], act = SH->pact[0];

  // ---- luma: SATD over the 13 modes (mode m by wave m % NW), stable sort ----
  for (int m = W; m < 13; m += NW) {
    predict_block_wh(f, x, y, WL, HL, availL, availU, m, 0, ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
    const long long sd = satd_wh(SH->srcb[0], S->pred, W_, H_);
    if (LANE == 0) SH->satd[m] = sd;
  }
  PH(3);
  WG_SYNC();
  if (LANE < 13) {
    const long long mine = SH->satd[LANE];
    int rank = 0;
    for (int j = 0; j < 13; j++) { const long long o = SH->satd[j]; rank += (o < mine) || (o == mine && j < LANE); }
    SH->order[rank] = LANE;
  }
  WG_SYNC();
  // ---- full RD over the surviving modes x tx types (no angle deltas below 8x8): evaluation e = ci * ntx + ti by wave e % NW ----
  const int ncand = Tools<TS>::FULL ? 7 : 3;
  int tx_ns = 0, tx_set = 0;
  const int tx_off0 = rect_tx_cdf<TS>(f, 0, &tx_ns, &tx_set);
  const int ntx = (Tools<TS>::rdo_tx(f) && tx_off0 >= 0) ? tx_ns : 1;
  // the surviving modes are predicted once (candidate ci by wave ci % NW) into the prediction cache and shared by their tx-type trials
  LDS uint16_t *pcache = MAXN <= 16 ? (LDS uint16_t *)SH->lpred : (LDS uint16_t *)SH->split_rec;      // [7][NN]; both are free until the tx-size trial
  for (int ci = W; ci < ncand; ci += NW) predict_block_wh(f, x, y, WL, HL, availL, availU, SH->order[ci], 0, ftype_y, ra, rl, wa, wl, S->etmp, pcache + ci * NN);
  PH(12);
  WG_SYNC();
  long long my_j = J_INF; int my_e = 1 << 30, my_mode = DC_PRED, my_tx = DCT_DCT, cur = 0; TxRes my_tr = { 0, 0, 0, 0, 0 }; uint32_t my_mrate = 0;
  // up to sixteen (mode x tx type) trials in ONE round, four per wave (one per 16-lane row, eval_group_wh): the 3 x 5 trials of speed 4
  constexpr bool CAN_GROUP = NW == 4 && MAXN <= 32;
  bool grouped = false, parked = false; int my_g = 0;
  if constexpr (CAN_GROUP) grouped = true;
  if constexpr (CAN_GROUP) if (grouped) {
    const int g = GROUP_ID, total = ncand * ntx, rounds = (total + 15) >> 4;
    LDS uint16_t *park_rec = (LDS uint16_t *)S->dcp; LDS int32_t *park_qc = (LDS int32_t *)(S->dcp + 64);     // between rounds a wave parks its best candidate here (dcp is idle until chroma)
    for (int rd = 0; rd < rounds; rd++) {
      int e;
      if (ncand == 3 && ntx == 5) e = g < 3 ? g * 5 + W : (W < 3 ? W * 5 + 4 : -1);   // a wave's rows 0..2 share the tx type (no divergence in the 1-D networks)
      else e = rd * 16 + W * 4 + g;
      const bool live = e >= 0 && e < total;
      const int ee = live ? e : 0, ci = ee / ntx, ti = ee - ci * ntx, m = SH->order[ci];
      const uint32_t mode_rate = ycost[m];
      int ns2, set2;
      const int tx_off = rect_tx_cdf<TS>(f, m, &ns2, &set2);
      int txtype;
      if (ntx > 1) txtype = sym_to_txtype(tx_set, ti);
      else { txtype = mode_to_txtype(m); if (tx_off < 0 || txtype_to_sym(tx_set, txtype) < 0) txtype = DCT_DCT; }
      GroupRes gr;
      eval_group_wh<WL, HL>(k.cc(), k.cost(), f, &S->grp[g], SH->srcb[0], pcache + ci * NN, 0, txtype, sctx_y, dctx_y, tx_off, tx_off >= 0 ? txtype_to_sym(tx_set, txtype) : 0,
                            Tools<TS>::tune_psnr(f) ? -1 : psv_a, psv_b, act, &gr);
      long long j = rd_dist32(f, 0, gr.sse) + rd_rate32(f, gr.rate) + rd_rate32(f, mode_rate);
      if (!live) j = J_INF;
      bool improved = false;
#pragma unroll
      for (int gg = 0; gg < 4; gg++) {
        const long long jg = ((long long)__builtin_amdgcn_readlane((int)(j >> 32), gg * 16) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)j, gg * 16);
        const int eg = __builtin_amdgcn_readlane(e, gg * 16);
        if (jg < my_j || (jg == my_j && eg < my_e)) {
          my_j = jg; my_e = eg; my_g = gg; improved = true;
          my_mode = __builtin_amdgcn_readlane(m, gg * 16); my_tx = __builtin_amdgcn_readlane(txtype, gg * 16);
          my_tr.eob = __builtin_amdgcn_readlane(gr.eob, gg * 16); my_tr.cul = __builtin_amdgcn_readlane(gr.cul, gg * 16); my_tr.dcc = __builtin_amdgcn_readlane(gr.dcc, gg * 16);
          my_mrate = (uint32_t)__builtin_amdgcn_readlane((int)mode_rate, gg * 16);
        }
      }
      if (rounds > 1 && improved) {                        // wave-uniform
        for (int i = LANE; i < NN; i += 64) { park_rec[i] = S->grp[my_g].rec[i]; park_qc[i] = S->grp[my_g].qc[i]; }
        parked = true;
        WAVE_SYNC();
      }
    }
  }
  if (!grouped)
  for (int e = W; e < ncand * ntx; e += NW) {
    const int ci = e / ntx, ti = e - ci * ntx, m = SH->order[ci];
    const LDS uint16_t *cpred = pcache + ci * NN;
    const uint32_t mode_rate = ycost[m];
    int ns2, set2;
    const int tx_off = rect_tx_cdf<TS>(f, m, &ns2, &set2);
    int txtype;
    if (ntx > 1) txtype = sym_to_txtype(tx_set, ti);
    else { txtype = mode_to_txtype(m); if (tx_off < 0 || txtype_to_sym(tx_set, txtype) < 0) txtype = DCT_DCT; }
    TxRes tr;
    long long j = eval_rect<MAXN, WL, HL, NW>(k, 0, sctx_y, dctx_y, SH->srcb[0], cpred, txtype, tx_off, tx_off >= 0 ? txtype_to_sym(tx_set, txtype) : 0, S->rec[cur], S->qc[cur], &tr, psv_a, psv_b, act);
    j += ((long long)mode_rate * f->rdmult + 256) >> 9;
    if (j < my_j) { my_j = j; my_e = e; my_mode = m; my_tx = txtype; my_tr = tr; my_mrate = mode_rate; cur ^= 1; }
  }
  if (LANE == 0) { SH->wbest_j[W] = my_j; SH->wbest_e[W] = my_e; }
  PH(6);
  WG_SYNC();
  int win = 0;
  for (int w2 = 1; w2 < NW; w2++) if (SH->wbest_j[w2] < SH->wbest_j[win] || (SH->wbest_j[w2] == SH->wbest_j[win] && SH->wbest_e[w2] < SH->wbest_e[win])) win = w2;
  const long long best_j = SH->wbest_j[win];
  if (W == win) {
    const int b = cur ^ 1;
    const LDS uint16_t *best_rec = S->rec[b]; const LDS int32_t *best_qc = S->qc[b];
    if constexpr (CAN_GROUP) if (grouped) {
      if (parked) { best_rec = (const LDS uint16_t *)S->dcp; best_qc = (const LDS int32_t *)(S->dcp + 64); }
      else { best_rec = S->grp[my_g].rec; best_qc = S->grp[my_g].qc; }
    }
    commit_rect<WL, HL>(f, 0, r, c, best_rec, best_qc, my_tr.eob, my_tr.cul, my_tr.dcc);
    fill_rect<WL, HL>(f->m_ymode, ms, r, c, my_mode);
    fill_rect<WL, HL>((uint8_t *)f->m_angle_y, ms, r, c, 0);
    fill_rect<WL, HL>(f->m_txtype, ms, r, c, my_tr.eob ? my_tx : DCT_DCT);
    fill_rect<WL, HL>(f->m_bsize, ms, r, c, BSR);
    fill_rect<WL, HL>(f->m_txsize, ms, r, c, BSR);
    if (f->np > 1) for (int i = LANE; i < NN; i += 64) SH->luma_rec[i] = best_rec[i];
    if (LANE == 0) { SH->lm_mode = my_mode; SH->lm_eob = my_tr.eob; SH->lm_mode_j = ((long long)my_mrate * f->rdmult + 256) >> 9; SH->lm_tx = my_tr.cul; SH->lm_delta = my_tr.dcc; }
  }
  WG_SYNC();
  const int best_mode = SH->lm_mode;
  long long luma_j = best_j; int any_coef = SH->lm_eob > 0;
  // ---- luma transform size: the 2:1 transform against its two 4x4 halves (Split_Tx_Size), tx_depth priced with the 8x8 category ----
  if (Tools<TS>::tx_mode_select(f)) {
    const int actx = nb_txU >= 0 && dim_wl(nb_txU) >= WL, lctx = nb_txL >= 0 && dim_hl(nb_txL) >= HL;
    const uint16_t *dcost = k.cost() + CDF_TX_SIZE + (actx + lctx) * CDF_TX_SIZE_STRIDE;
    luma_j += ((long long)dcost[0] * f->rdmult + 256) >> 9;
    if (Tools<TS>::rdo_tx(f)) {
      long long j_split = SH->lm_mode_j + (((long long)dcost[1] * f->rdmult + 256) >> 9);
      int stx_ns = 0, stx_set = 0;
      const int stx_off = intra_tx_cdf_r(f, Tools<TS>::reduced_tx_set(f), 0, best_mode, &stx_ns, &stx_set);
      const int sntx = stx_off >= 0 ? stx_ns : 1;
      LDS uint16_t *split_rec = (LDS uint16_t *)SH->ssrc + 64;             // [NN] (ssrc[0..31] = the two sub-sources)
      LDS int32_t *split_qc = MAXN <= 16 ? (LDS int32_t *)SH->lpred : (LDS int32_t *)SH->split_qc;   // [2][16]
      LDS int *sub = (LDS int *)SH->dsd;                                     // [2][4]: tx, eob, cul, dcc of the halves; [8..11]: the halves' outer neighbour contexts
      int sub_any = 0;
      if (W == 0) for (int idx = LANE; idx < NN; idx += 64) { const int q = W_ == 8 ? ((idx & 7) >> 2) : (idx >> 4), i = (idx >> WL) & 3, j = idx & 3; SH->ssrc[q * 16 + i * 4 + j] = SH->srcb[0][idx]; }
      if (W == 1 && LANE < 4) {
        // outer neighbour contexts of the two halves (txb_ctx_wh's loads): lane = 2 * half + (0: above, 1: left); level | dc << 8 | available << 16.  The second half's
        // inner neighbour (the first half) comes from the chain.
        const int q = LANE >> 1, side = LANE & 1, rr = r + (H_ == 8 ? q : 0), cc = c + (W_ == 8 ? q : 0);
        const bool inner = q == 1 && (side == 0 ? H_ == 8 : W_ == 8);
        const bool have = !inner && (side == 0 ? (rr - 1 >= t->mi_row_start && cc < f->mi_cols) : (cc - 1 >= t->mi_col_start && rr < f->mi_rows));
        const int ia = have ? (side == 0 ? (rr - 1) * ms + cc : rr * ms + cc - 1) : rr * ms + cc;
        const int l = f->m_lvl[0][ia], d = f->m_dc[0][ia];
        sub[8 + LANE] = have ? (l | (d << 8) | (1 << 16)) : 0;
      }
      WG_SYNC();
      // One chain per transform type (rav1e rdo_tx_type_decision: both halves with the same type), each on a 16-lane row like the square blocks' trial (tile_search.h):
      // five types: the four DCT / ADST combinations on wave 0's rows, IDTX alone on wave 1; otherwise four per wave on waves 0 and 1.  Chain state in waves 2 / 3's scratch.
      {
        constexpr int CH_BYTES = NN * 2 + NN * 4 + 2 * 4;
        const int g = GROUP_ID, gl = GROUP_LANE;
        const int e = sntx == 5 ? (W == 0 ? g + 1 : ((W == 1 && g == 0) ? 0 : 64)) : W * 4 + g;
        const bool has_chain = e < sntx, wave_has = sntx == 5 ? W < 2 : W * 4 < sntx;
        LDS long long *const cres_j = (LDS long long *)SH->cj; LDS int *const cres_any = (LDS int *)(cres_j + 8);
        const long long thr = luma_j < budget ? luma_j : budget;
        if (wave_has) {
          LDS uint8_t *cst = (LDS uint8_t *)k.wave(2 + W) + g * CH_BYTES;
          LDS uint16_t *canvas = (LDS uint16_t *)cst; LDS int32_t *cqc = (LDS int32_t *)(cst + NN * 2); LDS uint32_t *cmeta = (LDS uint32_t *)(cst + NN * 2 + NN * 4);
          LDS uint16_t *A = S->pred + g * 64 + EDGE_OFF, *Lf = A + 32, *ppred = S->dcp + g * 64;
          int txtype;
          if (sntx > 1) txtype = sym_to_txtype(stx_set, has_chain ? e : 0);
          else { txtype = mode_to_txtype(best_mode); if (stx_off < 0 || txtype_to_sym(stx_set, txtype) < 0) txtype = DCT_DCT; }
          const int tx_sym = stx_off >= 0 ? txtype_to_sym(stx_set, txtype) : 0;
          const bool dirm = best_mode >= V_PRED && best_mode <= D67_PRED;
          const int pa = dirm ? mode_angle_of(best_mode) : 0;
          const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1, rs = f->stride, bd = f->bd;
          const uint16_t *grec = f->rec[0];
          long long jc = has_chain ? j_split : J_INF;
          int any = 0;
#pragma unroll 1
          for (int q = 0; q < 2; q++) {
            if (MI_BALLOT64(jc < thr) == 0ull) break;
            const int bi = H_ == 8 ? q : 0, bj = W_ == 8 ? q : 0, rr = r + bi, cc = c + bj, sx = x + bj * 4, sy = y + bi * 4;
            const int sU = availU || bi, sL = availL || bj;
            {
              // availability of the halves' above-right / below-left runs (oracle: the decoded flags of the cells they start in)
              const bool c_ar = sU && cc + 1 < t->mi_col_end, c_bl = sL && rr + 1 < t->mi_row_end;
              const int s_ar = c_ar && decoded_before(r, c, SHAPE, rr - 1, cc + 1), s_bl = c_bl && decoded_before(r, c, SHAPE, rr + 1, cc - 1);   // (both cells lie outside the block)
              const int lim_a = imin_(max_x, sx + (s_ar ? 8 : 4) - 1), lim_l = imin_(max_y, sy + (s_bl ? 8 : 4) - 1);
              auto px = [&](int ax, int ay) -> int {
                const int xr = ax - x, yr = ay - y;
                if (xr >= 0 && xr < W_ && yr >= 0 && yr < H_) return (int)canvas[yr * W_ + xr];
                if (yr == -1 && xr >= -1 && xr < W_ + H_) return (int)ra[xr];
                if (xr == -1 && yr >= 0 && yr < W_ + H_) return (int)rl[yr];
                return (int)grec[(size_t)ay * rs + ax];
              };
              for (int i = gl; i <= 8; i += 16) {
                const bool corner = i == 8;
                int a, l;
                if (sU) a = px(corner ? (sL ? sx - 1 : sx) : imin_(lim_a, sx + i), sy - 1); else a = px(sL ? sx - 1 : sx, sy);
                if (sL) l = px(sx - 1, imin_(lim_l, sy + i)); else l = px(sx, sU ? sy - 1 : sy);
                if (!sU && !sL) { a = corner ? (1 << (bd - 1)) : (1 << (bd - 1)) - 1; l = (1 << (bd - 1)) + 1; }
                if (corner) { A[-1] = (uint16_t)a; Lf[-1] = (uint16_t)a; } else { A[i] = (uint16_t)a; Lf[i] = (uint16_t)l; }
              }
            }
            WAVE_SYNC();
            if (dirm) predict_dir_group<4>(f, sx, sy, sL, sU, pa, ftype_y, A, Lf, &S->gpred[g], ppred);
            else predict_nondir_group<4>(best_mode, sL, sU, bd, A, Lf, ppred);
            int ssc, sdc;
            {
              const uint32_t m0 = q ? cmeta[0] : 0u;
              const int inner = (int)(((m0 >> 16) & 0xFF) | ((m0 >> 24) << 8) | (1u << 16));      // the first half as a neighbour: level | dc << 8 | available
              const int nt = (q == 1 && H_ == 8) ? inner : sub[8 + 2 * q], nl = (q == 1 && W_ == 8) ? inner : sub[8 + 2 * q + 1];
              const int top = (nt >> 16) ? (nt & 0xFF) : 0, left = (nl >> 16) ? (nl & 0xFF) : 0, dt = (nt >> 16) ? ((nt >> 8) & 0xFF) : 0, dl = (nl >> 16) ? ((nl >> 8) & 0xFF) : 0;
              const int dcs = (dt == 1 ? -1 : (dt == 2 ? 1 : 0)) + (dl == 1 ? -1 : (dl == 2 ? 1 : 0));
              sdc = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
              if (top == 0 && left == 0) ssc = 1;
              else if (top == 0 || left == 0) ssc = 2 + (imax_(top, left) > 3);
              else if (imax_(top, left) <= 3) ssc = 4;
              else if (imin_(top, left) <= 3) ssc = 5;
              else ssc = 6;
            }
            GroupRes gr;
            eval_group<4>(k.cc(), k.cost(), k.ls(), f, &S->grp[g], SH->ssrc + q * 16, ppred, 0, 0, txtype, ssc, sdc, stx_off, tx_sym, Tools<TS>::tune_psnr(f) ? -1 : SH->psv4[q], SH->pact[0], &gr);
            if (has_chain) jc += rd_dist32(f, 0, gr.sse) + rd_rate32(f, gr.rate);
            {
              const LDS uint16_t *srec = S->grp[g].rec; const LDS int32_t *sqc = S->grp[g].qc;
              canvas[(bi * 4 + (gl >> 2)) * W_ + bj * 4 + (gl & 3)] = srec[gl]; cqc[q * 16 + gl] = sqc[gl];
              cmeta[q] = (uint32_t)gr.eob | ((uint32_t)gr.cul << 16) | ((uint32_t)gr.dcc << 24);
            }
            any |= gr.eob > 0;
            WAVE_SYNC();
          }
          // (every lane stores, rows without a chain into a slot nobody reads: see the square blocks' trial in tile_search.h for why no divergent region ends here)
          const int slot = sntx == 5 ? (W == 0 ? g + 1 : (g == 0 ? 0 : 4 + g)) : W * 4 + g;
          cres_j[slot] = (has_chain && jc < thr) ? jc : J_INF; cres_any[slot] = any;
        }
        WG_SYNC();
        int be = -1; long long bjc = J_INF;
        for (int e2 = 0; e2 < sntx; e2++) { const long long v = cres_j[e2]; if (v < bjc) { bjc = v; be = e2; } }
        j_split = bjc;
        if (be >= 0) {
          const int ow = sntx == 5 ? (be == 0 ? 1 : 0) : be >> 2, og = sntx == 5 ? (be == 0 ? 0 : be - 1) : be & 3;
          if (W == ow) {
            const LDS uint8_t *cst = (const LDS uint8_t *)k.wave(2 + W) + og * CH_BYTES;
            const LDS uint16_t *canvas = (const LDS uint16_t *)cst; const LDS int32_t *cqc = (const LDS int32_t *)(cst + NN * 2); const LDS uint32_t *cmeta = (const LDS uint32_t *)(cst + NN * 2 + NN * 4);
            if (LANE < NN) { split_rec[LANE] = canvas[LANE]; split_qc[LANE] = cqc[LANE]; }
            int btx;
            if (sntx > 1) btx = sym_to_txtype(stx_set, be);
            else { btx = mode_to_txtype(best_mode); if (stx_off < 0 || txtype_to_sym(stx_set, btx) < 0) btx = DCT_DCT; }
            if (LANE < 2) { const uint32_t m = cmeta[LANE]; const int eob = (int)(m & 0xFFFF); sub[LANE * 4 + 0] = eob ? btx : DCT_DCT; sub[LANE * 4 + 1] = eob; sub[LANE * 4 + 2] = (int)((m >> 16) & 0xFF); sub[LANE * 4 + 3] = (int)(m >> 24); }
          }
          sub_any = cres_any[be];
          WG_SYNC();
        }
      }
      if (j_split < luma_j) {
        luma_j = j_split; any_coef = sub_any;
        uint16_t *gr_ = f->rec[0] + (size_t)y * f->stride + x;
        int32_t *gc_ = f->coef[0] + (size_t)y * f->stride + x;
        for (int i = threadIdx.x; i < NN; i += 64 * NW) {
          const uint16_t v = split_rec[i]; gr_[(i >> WL) * f->stride + (i & (W_ - 1))] = v; if (f->np > 1) SH->luma_rec[i] = v;
          const int q = W_ == 8 ? ((i & 7) >> 2) : (i >> 4), ii = (i >> WL) & 3, jj = i & 3;
          gc_[(i >> WL) * f->stride + (i & (W_ - 1))] = split_qc[q * 16 + ii * 4 + jj];
        }
        if (W == 0 && LANE < 2) {
          const int rr = r + (H_ == 8 ? LANE : 0), cc = c + (W_ == 8 ? LANE : 0), o = rr * ms + cc;
          f->m_txsize[o] = 0; f->m_txtype[o] = (uint8_t)sub[LANE * 4 + 0]; f->m_eob[0][o] = (uint16_t)sub[LANE * 4 + 1];
          f->m_lvl[0][o] = (uint8_t)sub[LANE * 4 + 2]; f->m_dc[0][o] = (uint8_t)sub[LANE * 4 + 3];
        }
      } else if (W == 0 && LANE < 2) {                                        // the undivided transform stays: its contexts back over the trial's
        const int o = (r + (H_ == 8 ? LANE : 0)) * ms + c + (W_ == 8 ? LANE : 0);
        f->m_lvl[0][o] = (uint8_t)SH->lm_tx; f->m_dc[0][o] = (uint8_t)SH->lm_delta;
      }
      WG_SYNC();
    }
  }
  if (luma_j >= budget) return luma_j;
  long long total_j = luma_j;
  PH(13);
  // ---- chroma with the simple candidate set (DC, luma's mode, CfL): the CfL alpha scan on all four waves (plane x half of the range), then every
  // candidate of a plane in one grouped evaluation (waves 0 and 2, one candidate per 16-lane row) -- the square path's scheme (tile_search.h) ----
  bool cgrouped = false;
  if constexpr (CAN_GROUP) cgrouped = f->np > 1 && !Tools<TS>::FULL;
  if constexpr (CAN_GROUP) if (cgrouped) {
    const uint16_t *uvcost = k.cost() + CDF_UV_CFL + best_mode * CDF_UV_CFL_STRIDE;
    const int nplain = best_mode != DC_PRED ? 2 : 1, nc = nplain + 1, uvset = Tools<TS>::reduced_tx_set(f) ? 2 : 1;
    const int mx = (1 << f->bd) - 1;
    {
      const int p = (W >> 1) + 1, half = W & 1;
      const LDS uint16_t *luma = SH->luma_rec;
      int lsum = LANE < NN ? (int)luma[LANE] << 3 : 0;
      lsum = wave_sum_i32(lsum);
      const int avg = round2_(lsum, WL + HL);
      predict_block_wh(f, x, y, WL, HL, availL, availU, DC_PRED, 0, ftype_uv, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF, wa, wl, S->etmp, S->dcp);
      long long best_sse = J_INF; int best_idx = 1 << 20;
      const int l = LANE < NN ? ((int)luma[LANE] << 3) - avg : 0, dcv = LANE < NN ? (int)S->dcp[LANE] : 0, sv = LANE < NN ? (int)SH->srcb[p][LANE] : 0;
      if (half == 0) { const int d = sv - dcv; best_sse = (long long)wave_sum_i32(LANE < NN ? d * d : 0); best_idx = -1; }
      const int la = iabs_(l), ng = l < 0;
#pragma unroll
      for (int kq = 0; kq < 8; kq++) {                                       // scan position 2k is alpha +(k + 1), 2k + 1 is -(k + 1)
        const int mag = half * 8 + kq + 1;
        const int rr = round2_(__mul24(mag, la), 6), sc = ng ? -rr : rr;
        const int dp = sv - iclamp_(dcv + sc, 0, mx), dm = sv - iclamp_(dcv - sc, 0, mx);
        const long long ep = (long long)wave_sum_i32(LANE < NN ? __mul24(dp, dp) : 0), em = (long long)wave_sum_i32(LANE < NN ? __mul24(dm, dm) : 0);
        if (ep < best_sse) { best_sse = ep; best_idx = half * 16 + 2 * kq; }
        if (em < best_sse) { best_sse = em; best_idx = half * 16 + 2 * kq + 1; }
      }
      if (LANE == 0) { SH->ca_sse[p - 1][half] = best_sse; SH->ca_idx[p - 1][half] = best_idx; }
    }
    WG_SYNC();
    int alpha_u = 0, alpha_v = 0;
#pragma unroll
    for (int pp = 0; pp < 2; pp++) {
      const int idx = SH->ca_sse[pp][1] < SH->ca_sse[pp][0] ? SH->ca_idx[pp][1] : SH->ca_idx[pp][0];
      const int al = idx < 0 ? 0 : ((idx & 1) ? -((idx >> 1) + 1) : ((idx >> 1) + 1));
      if (pp == 0) alpha_u = al; else alpha_v = al;
    }
    const int cfl_ok = alpha_u != 0 || alpha_v != 0;
    GroupRes gr = { 0, 0, 0, 0, 0 };
    const int cw = (W & 1) == 0;                              // waves 0 (plane U) and 2 (plane V): their S->dcp already holds the plane's DC prediction
    // (the luma mode's prediction of plane U / V by the otherwise idle wave 1 / 3, into wave 0's / 2's S->pred: tile_search.h try_block)
    if (!cw && nplain == 2) {
      const int p = (W >> 1) + 1;
      predict_block_wh(f, x, y, WL, HL, availL, availU, best_mode, 0, ftype_uv, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF, wa, wl, S->etmp, k.wave(W - 1)->pred + NN);
    }
    if (cw) {
      const int p = (W >> 1) + 1;
      {
        const int al = p == 1 ? alpha_u : alpha_v;
        LDS uint16_t *cp = S->pred + (nc - 1) * NN;
        int lsum = LANE < NN ? (int)SH->luma_rec[LANE] << 3 : 0;
        lsum = wave_sum_i32(lsum);
        const int avg = round2_(lsum, WL + HL);
        if (LANE < NN) { const int l = ((int)SH->luma_rec[LANE] << 3) - avg, v = al * l, sc = v >= 0 ? round2_(v, 6) : -round2_(-v, 6); cp[LANE] = (uint16_t)iclamp_((int)S->dcp[LANE] + sc, 0, mx); }
      }
      WAVE_SYNC();
    }
    WG_SYNC();
    if (cw) {
      const int p = (W >> 1) + 1;
      const int g = GROUP_ID, cand = imin_(g, nc - 1);
      const int um = cand == nc - 1 ? UV_CFL_PRED : (cand == 0 ? DC_PRED : best_mode);
      int txtype = mode_to_txtype(um);
      if (txtype_to_sym(uvset, txtype) < 0) txtype = DCT_DCT;
      eval_group_wh<WL, HL>(k.cc(), k.cost(), f, &S->grp[g], SH->srcb[p], cand == 0 ? (const LDS uint16_t *)S->dcp : (const LDS uint16_t *)(S->pred + cand * NN), p, txtype,
                            SH->sctx[p], SH->dctx[p], -1, 0, -1, 0, SH->cact, &gr);
      const long long jp = rd_dist32(f, p, gr.sse) + rd_rate32(f, gr.rate);
      if (GROUP_LANE == 0 && g < nc) SH->cj[g][p - 1] = jp;
    }
    WG_SYNC();
    long long best_uv = J_INF; int bc = 0, b_sign = 0;
#pragma unroll
    for (int cnd = 0; cnd < 3; cnd++) {
      if (cnd < nc) {
        const int is_cfl = cnd == nc - 1, um = is_cfl ? UV_CFL_PRED : (cnd == 0 ? DC_PRED : best_mode);
        int jsign = 0;
        const uint32_t mode_rate = uv_mode_rate(k.cost(), uvcost, um, false, 0, is_cfl && cfl_ok, alpha_u, alpha_v, &jsign);
        if (!is_cfl || cfl_ok) {
          const long long j = SH->cj[cnd][0] + SH->cj[cnd][1] + rd_rate32(f, mode_rate);
          if (j < best_uv) { best_uv = j; bc = cnd; b_sign = jsign; }
        }
      }
    }
    if (cw) {
      const int p = (W >> 1) + 1, chose_cfl = bc == nc - 1;
      const int beob = __builtin_amdgcn_readlane(gr.eob, bc * 16), bcul = __builtin_amdgcn_readlane(gr.cul, bc * 16), bdcc = __builtin_amdgcn_readlane(gr.dcc, bc * 16);
      commit_rect<WL, HL>(f, p, r, c, S->grp[bc].rec, S->grp[bc].qc, beob, bcul, bdcc);
      if (LANE == 0) SH->ceob[p - 1] = beob;
      if (p == 1) {
        fill_rect<WL, HL>(f->m_uvmode, ms, r, c, chose_cfl ? UV_CFL_PRED : (bc == 0 ? DC_PRED : best_mode));
        fill_rect<WL, HL>((uint8_t *)f->m_angle_uv, ms, r, c, 0);
        fill_rect<WL, HL>(f->m_cfl_sign, ms, r, c, chose_cfl ? b_sign : 0);
        fill_rect<WL, HL>(f->m_cfl_au, ms, r, c, (chose_cfl && alpha_u) ? iabs_(alpha_u) - 1 : 0);
        fill_rect<WL, HL>(f->m_cfl_av, ms, r, c, (chose_cfl && alpha_v) ? iabs_(alpha_v) - 1 : 0);
      }
    }
    WG_SYNC();
    any_coef |= (SH->ceob[0] > 0) | (SH->ceob[1] > 0);
    total_j += best_uv;
  }
  // ---- chroma with the FULL candidate set of speed <= 1 (oracle order: DC, the luma mode, the other eleven modes, CfL), in the kernels instantiated for that set
  // (Tools<TS>::FULL): the CfL alpha scan on all four waves (plane x half of the range), then the candidates four per wavefront (eval_group_wh, one candidate per 16-lane
  // row): wave W takes plane W / 2 + 1 and the candidates of parity W % 2, eight candidates of a plane per round, two rounds -- the later candidates first, so that
  // the likely winners (DC, the luma mode) are still in the wave's buffers at the end; the first round's best waits in LDS (rec + levels: 192 B per plane).  Until
  // round 4 this set ran one candidate per ROUND on two waves (14 rounds of predict + evaluate + three barriers): half of config 5's critical path. ----
  if constexpr (Tools<TS>::FULL) if (f->np > 1 && !cgrouped) {
    const uint16_t *uvcost = k.cost() + CDF_UV_CFL + best_mode * CDF_UV_CFL_STRIDE;
    unsigned long long cand_pack = 0; int nc = 0;
    auto push = [&](int m) { cand_pack |= (unsigned long long)m << (4 * nc); nc++; };
    push(DC_PRED);
    if (best_mode != DC_PRED) push(best_mode);
    if (Tools<TS>::FULL) for (int m = 1; m < 13; m++) if (m != best_mode) push(m);
    push(UV_CFL_PRED);
    const int uvset = Tools<TS>::reduced_tx_set(f) ? 2 : 1;
    const int mx = (1 << f->bd) - 1;
    const int p = (W >> 1) + 1, half = W & 1;
    const LDS uint16_t *pra = SH->ra[p] + EDGE_OFF, *prl = SH->rl[p] + EDGE_OFF;
    int lavg;
    {
      // rdo_cfl_alpha (oracle cfl_best_alpha: alpha 0, then +1, -1, +2, -2, ... +16, -16; the first strictly smaller SSE wins): this wave scans one half of the range
      const LDS uint16_t *luma = SH->luma_rec;
      int lsum = LANE < NN ? (int)luma[LANE] << 3 : 0;
      lsum = wave_sum_i32(lsum);
      lavg = round2_(lsum, WL + HL);
      predict_block_wh(f, x, y, WL, HL, availL, availU, DC_PRED, 0, ftype_uv, pra, prl, wa, wl, S->etmp, S->dcp);
      long long best_sse = J_INF; int best_idx = 1 << 20;
      const int l = LANE < NN ? ((int)luma[LANE] << 3) - lavg : 0, dcv = LANE < NN ? (int)S->dcp[LANE] : 0, sv = LANE < NN ? (int)SH->srcb[p][LANE] : 0;
      if (half == 0) { const int d = sv - dcv; best_sse = (long long)wave_sum_i32(LANE < NN ? d * d : 0); best_idx = -1; }
      const int la = iabs_(l), ng = l < 0;
#pragma unroll
      for (int kq = 0; kq < 8; kq++) {                                       // scan position 2k is alpha +(k + 1), 2k + 1 is -(k + 1)
        const int mag = half * 8 + kq + 1;
        const int rr = round2_(__mul24(mag, la), 6), sc = ng ? -rr : rr;
        const int dp = sv - iclamp_(dcv + sc, 0, mx), dm = sv - iclamp_(dcv - sc, 0, mx);
        const long long ep = (long long)wave_sum_i32(LANE < NN ? __mul24(dp, dp) : 0), em = (long long)wave_sum_i32(LANE < NN ? __mul24(dm, dm) : 0);
        if (ep < best_sse) { best_sse = ep; best_idx = half * 16 + 2 * kq; }
        if (em < best_sse) { best_sse = em; best_idx = half * 16 + 2 * kq + 1; }
      }
      if (LANE == 0) { SH->ca_sse[p - 1][half] = best_sse; SH->ca_idx[p - 1][half] = best_idx; }
    }
    WG_SYNC();
    int alpha_u = 0, alpha_v = 0;
#pragma unroll
    for (int pp = 0; pp < 2; pp++) {
      const int idx = SH->ca_sse[pp][1] < SH->ca_sse[pp][0] ? SH->ca_idx[pp][1] : SH->ca_idx[pp][0];
      const int al = idx < 0 ? 0 : ((idx & 1) ? -((idx >> 1) + 1) : ((idx >> 1) + 1));
      if (pp == 0) alpha_u = al; else alpha_v = al;
    }
    const int cfl_ok = alpha_u != 0 || alpha_v != 0;
    // what waits in LDS across the rounds: the first round's best candidate of each plane (reconstruction, levels, eob / cul / dcc)
    LDS uint16_t *park_rec = (LDS uint16_t *)SH->lpred + (p - 1) * 96; LDS int32_t *park_qc = (LDS int32_t *)(park_rec + 32); LDS int *park_meta = (LDS int *)SH->order + (p - 1) * 3;
    long long best_uv = J_INF; int b_ci = 1 << 30, b_sign = 0, b_round = 0;
    const int ns = nc <= 4 ? 1 : 2, nrounds = nc > 4 * ns ? 2 : 1, dealt = half < ns;
    GroupRes gr = { 0, 0, 0, 0, 0 };
    const int g = GROUP_ID;
#pragma unroll 1
    for (int rd = 0; rd < nrounds; rd++) {
      const int base = nrounds == 2 && rd == 0 ? 4 * ns : 0;                  // candidates base .. base + 4 ns - 1 of the list
      // this wave's (at most four) predictions of the round, side by side in S->pred (the DC candidate reads S->dcp)
#pragma unroll 1
      for (int g2 = 0; g2 < 4; g2++) {
        const int ci = base + half + ns * g2;
        if (dealt && ci < nc) {
          const int um = lut4(cand_pack, ci);
          LDS uint16_t *cp = S->pred + g2 * NN;
          if (um == UV_CFL_PRED) {
            const int al = p == 1 ? alpha_u : alpha_v;
            if (LANE < NN) { const int l = ((int)SH->luma_rec[LANE] << 3) - lavg, v = al * l, sc = v >= 0 ? round2_(v, 6) : -round2_(-v, 6); cp[LANE] = (uint16_t)iclamp_((int)S->dcp[LANE] + sc, 0, mx); }
            WAVE_SYNC();
          } else if (um != DC_PRED) predict_block_wh(f, x, y, WL, HL, availL, availU, um, 0, ftype_uv, pra, prl, wa, wl, S->etmp, cp);
        }
      }
      const int ci = base + half + ns * g, live = dealt && ci < nc;
      const int um = lut4(cand_pack, live ? ci : 0);
      int txtype = mode_to_txtype(um);
      if (txtype_to_sym(uvset, txtype) < 0) txtype = DCT_DCT;
      if (dealt) eval_group_wh<WL, HL>(k.cc(), k.cost(), f, &S->grp[g], SH->srcb[p], um == DC_PRED ? (const LDS uint16_t *)S->dcp : (const LDS uint16_t *)(S->pred + g * NN), p, txtype,
                            SH->sctx[p], SH->dctx[p], -1, 0, -1, 0, SH->cact, &gr);
      const long long jp = rd_dist32(f, p, gr.sse) + rd_rate32(f, gr.rate);
      if (GROUP_LANE == 0 && live) SH->cj[ci][p - 1] = jp;
      WG_SYNC();
      // every wave: the best of the round's candidates, in list order (strictly smaller wins: the oracle's loop)
      long long r_best = J_INF; int r_ci = 1 << 30, r_sign = 0;
      for (int cc = base; cc < imin_(nc, base + 4 * ns); cc++) {
        const int um2 = lut4(cand_pack, cc), is_cfl = um2 == UV_CFL_PRED;
        if (is_cfl && !cfl_ok) continue;
        int jsign = 0;
        const uint32_t mode_rate = uv_mode_rate(k.cost(), uvcost, um2, false, 0, is_cfl, alpha_u, alpha_v, &jsign);
        const long long j = SH->cj[cc][0] + SH->cj[cc][1] + rd_rate32(f, mode_rate);
        if (j < r_best) { r_best = j; r_ci = cc; r_sign = jsign; }
      }
      if (r_best < best_uv || (r_best == best_uv && r_ci < b_ci)) { best_uv = r_best; b_ci = r_ci; b_sign = r_sign; b_round = rd; }
      if (nrounds == 2 && rd == 0 && r_ci < nc && (r_ci - base) % ns == half) {       // this wave holds the first round's best candidate of its plane: park it
        const int gg = (r_ci - base) / ns;
        if (LANE < NN) { park_rec[LANE] = S->grp[gg].rec[LANE]; park_qc[LANE] = S->grp[gg].qc[LANE]; }
        const int e_ = __builtin_amdgcn_readlane(gr.eob, gg * 16), c_ = __builtin_amdgcn_readlane(gr.cul, gg * 16), d_ = __builtin_amdgcn_readlane(gr.dcc, gg * 16);
        if (LANE == 0) { park_meta[0] = e_; park_meta[1] = c_; park_meta[2] = d_; }
      }
      // (no barrier here: the next round writes other cj entries, and the parked data is read after that round's barrier)
    }
    const int b_um = lut4(cand_pack, b_ci), chose_cfl = b_um == UV_CFL_PRED;
    const int from_park = nrounds == 2 && b_round == 0;
    if (from_park ? half == 0 : b_ci % ns == half) {                        // the wave that commits plane p
      int beob, bcul, bdcc;
      if (from_park) {
        beob = park_meta[0]; bcul = park_meta[1]; bdcc = park_meta[2];
        commit_rect<WL, HL>(f, p, r, c, (const LDS uint16_t *)park_rec, (const LDS int32_t *)park_qc, beob, bcul, bdcc);
      } else {
        const int gg = b_ci / ns;
        beob = __builtin_amdgcn_readlane(gr.eob, gg * 16); bcul = __builtin_amdgcn_readlane(gr.cul, gg * 16); bdcc = __builtin_amdgcn_readlane(gr.dcc, gg * 16);
        commit_rect<WL, HL>(f, p, r, c, S->grp[gg].rec, S->grp[gg].qc, beob, bcul, bdcc);
      }
      if (LANE == 0) SH->ceob[p - 1] = beob;
      if (p == 1) {
        fill_rect<WL, HL>(f->m_uvmode, ms, r, c, b_um);
        fill_rect<WL, HL>((uint8_t *)f->m_angle_uv, ms, r, c, 0);
        fill_rect<WL, HL>(f->m_cfl_sign, ms, r, c, chose_cfl ? b_sign : 0);
        fill_rect<WL, HL>(f->m_cfl_au, ms, r, c, (chose_cfl && alpha_u) ? iabs_(alpha_u) - 1 : 0);
        fill_rect<WL, HL>(f->m_cfl_av, ms, r, c, (chose_cfl && alpha_v) ? iabs_(alpha_v) - 1 : 0);
      }
    }
    WG_SYNC();
    any_coef |= (SH->ceob[0] > 0) | (SH->ceob[1] > 0);
    total_j += best_uv;
  }
  // ---- skip flag ----
  PH(9);
  const int skip = !any_coef;
  int seg_ctx = 0;                                                       // intra_segment_id, as in try_block
  const int seg_nb2 = SH->seg_nb, seg_ul = (seg_nb2 & 15) - 1, seg_u = ((seg_nb2 >> 4) & 15) - 1, seg_l = (seg_nb2 >> 8) - 1;
  const int seg_p = seg_pred(seg_ul, seg_u, seg_l, &seg_ctx), seg_own = f->seg_n ? SH->seg : 0, seg_fin = f->seg_n ? (skip ? seg_p : seg_own) : 0;
  if (f->seg_n && !skip) total_j += ((long long)k.cost()[CDF_SEG_ID + seg_ctx * CDF_SEG_ID_STRIDE + seg_symbol(seg_own, seg_p, f->seg_n)] * f->rdmult + 256) >> 9;
  if (W == 0) {
    fill_rect<WL, HL>(f->m_skip, ms, r, c, skip | (seg_fin << 1));
    if (skip) for (int p = 0; p < f->np; p++) { fill_rect<WL, HL>(f->m_lvl[p], ms, r, c, 0); fill_rect<WL, HL>(f->m_dc[p], ms, r, c, 0); }
  }
  total_j += ((long long)k.cost()[CDF_SKIP + nb_skip * CDF_SKIP_STRIDE + skip] * f->rdmult + 256) >> 9;
  WG_SYNC();
  return total_j;
}

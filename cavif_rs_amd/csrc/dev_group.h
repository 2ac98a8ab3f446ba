// dev_group.h -- four transform-block evaluations per wavefront, for 4x4 and 8x8 blocks.
// A 4x4 / 8x8 evaluation keeps at most 16 lanes busy (8 columns, 64 coefficients), so the one-candidate-per-wave
// scheme of tile_search.h ran four such evaluations one after the other on a wave that was three quarters idle.
// Here each 16-lane DPP row ("group") owns one candidate: same arithmetic, same rounding points, same results as
// eval_tx()/quant_rate_dev() -- only the lane mapping differs: residual, quantiser, rate and SSE use 16 lanes per
// candidate, the 1-D networks use lane l < N of the group as column / row l, reductions stay inside the row (DPP
// row_ror all-reduce).  Everything that is wave-uniform in the single-candidate code is row-uniform here.
#pragma once
#include "dev_predict.h"
#include "dev_txfm.h"
#include "dev_rate.h"

struct GroupBuf8 {                    // one candidate's working set (N <= 8)
  int32_t tbuf[8 * 9];                // residual / transposition buffer, padded pitch
  int32_t cbuf[64];                   // forward coefficients, then the dequantised block
  int32_t qc[64];                     // quantised levels (raster)
  uint16_t rec[64];                   // reconstruction
  uint8_t lev[144];                   // padded level map (re-zeroed for every evaluation: the memory is shared with the one-candidate path)
};
#define GROUP_LANE (LANE & 15)
#define GROUP_ID (LANE >> 4)

template <int CTRL> __device__ __forceinline__ int dpp_row_(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int row_sum_i32(int v) { v += dpp_row_<0x128>(v); v += dpp_row_<0x124>(v); v += dpp_row_<0x122>(v); v += dpp_row_<0x121>(v); return v; }   // row_ror 8, 4, 2, 1
__device__ __forceinline__ int row_max_i32(int v) {
  v = imax_(v, dpp_row_<0x128>(v)); v = imax_(v, dpp_row_<0x124>(v)); v = imax_(v, dpp_row_<0x122>(v)); v = imax_(v, dpp_row_<0x121>(v)); return v;
}

// ---- 4x4 blocks: both 1-D passes in registers.  Every 4-point network of txfm_gen.hip.h rounds each output once, so it is a
// 4x4 integer matrix: out[k] = R12(A[k] . x) + sign * R12(B[k] . x)  (B is the odd half of the inverse DCT, whose butterfly adds two
// separately rounded terms; zero everywhere else).  A group's 16 lanes hold the block one sample per lane, a quad = one column (then
// one row); a pass is four quad broadcasts (DPP quad_perm) and four 24-bit multiply-adds, the transposition between the passes one
// ds_bpermute.  Table: [inverse][kind][k] -> A[0..3], B[0..3].
static __device__ const int tx4_tab[2][3][4][8] = {
  { { { 2896, 2896, 2896, 2896, 0, 0, 0, 0 }, { 3784, 1567, -1567, -3784, 0, 0, 0, 0 }, { 2896, -2896, -2896, 2896, 0, 0, 0, 0 }, { 1567, -3784, 3784, -1567, 0, 0, 0, 0 } },
    { { 1321, 2482, 3344, 3803, 0, 0, 0, 0 }, { 3344, 3344, 0, -3344, 0, 0, 0, 0 }, { 3803, -1321, -3344, 2482, 0, 0, 0, 0 }, { 2482, -3803, 3344, -1321, 0, 0, 0, 0 } },
    { { 5793, 0, 0, 0, 0, 0, 0, 0 }, { 0, 5793, 0, 0, 0, 0, 0, 0 }, { 0, 0, 5793, 0, 0, 0, 0, 0 }, { 0, 0, 0, 5793, 0, 0, 0, 0 } } },
  { { { 2896, 0, 2896, 0, 0, 3784, 0, 1567 }, { 2896, 0, -2896, 0, 0, 1567, 0, -3784 }, { 2896, 0, -2896, 0, 0, 1567, 0, -3784 }, { 2896, 0, 2896, 0, 0, 3784, 0, 1567 } },
    { { 1321, 3344, 3803, 2482, 0, 0, 0, 0 }, { 2482, 3344, -1321, -3803, 0, 0, 0, 0 }, { 3344, 0, -3344, 3344, 0, 0, 0, 0 }, { 3803, -3344, 2482, -1321, 0, 0, 0, 0 } },
    { { 5793, 0, 0, 0, 0, 0, 0, 0 }, { 0, 5793, 0, 0, 0, 0, 0, 0 }, { 0, 0, 5793, 0, 0, 0, 0, 0 }, { 0, 0, 0, 5793, 0, 0, 0, 0 } } } };
__device__ __forceinline__ int tx4_dot(int v, const int *m) {      // R12(m . quad), the quad's four values broadcast to each of its lanes
  int acc = __mul24(m[0], dpp_row_<0x00>(v)) + 2048;
  acc += __mul24(m[1], dpp_row_<0x55>(v));
  acc += __mul24(m[2], dpp_row_<0xAA>(v));
  acc += __mul24(m[3], dpp_row_<0xFF>(v));
  return acc >> 12;
}
__device__ __forceinline__ int tx4_fwd(int v, int kind) { return tx4_dot(v, tx4_tab[0][kind][LANE & 3]); }
__device__ __forceinline__ int tx4_inv(int v, int kind) {
  const int *m = tx4_tab[1][kind][LANE & 3];
  const int e = tx4_dot(v, m), o = tx4_dot(v, m + 4);              // o = R12(0) = 0 outside the DCT
  return (LANE & 2) ? e - o : e + o;
}
__device__ __forceinline__ int group_transpose4(int v) { return __shfl(v, (LANE & 48) | ((LANE & 3) << 2) | ((LANE >> 2) & 3)); }

struct GroupRes { int eob, cul, dcc, sse; uint32_t rate; };

// A 4x4 candidate from its forward coefficients on, WITHOUT leaving the registers: the group's 16 lanes hold the block in raster order (lane gl = position gl), so
// the quantiser, the level-map contexts and the rate work on the lane's own coefficient and on its neighbours' levels fetched with DPP row shifts (right: +1 .. +3,
// below: +4, +8, +12, diagonal: +5; a shift that leaves the 16-lane row reads zero like the level map's padding did) -- where the generic form wrote coefficients,
// levels and a padded level map to LDS and read them back through the scan order (three round trips and a map reset per candidate).  The scan index a rate context
// needs is the inverse scan of the lane's position (a packed constant for the default scan, arithmetic for the row / column scans).  Sums over the block do not depend
// on the lane order, so every result is the generic path's bit for bit.  LDS is touched for the cost tables and for the two things the caller reads: levels and pixels.
template <int CTRL> __device__ __forceinline__ int dpp_row_z(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }   // source outside the row: 0
template <typename FP>
__device__ __forceinline__ void eval_group4_tail(const CoefCost &cc, FP f, LDS GroupBuf8 *gb, int coef, int t_src, int t_rec, int tidx, int plane, int txs, int txtype, int ck, int rk,
                                                 int skip_ctx, int dc_ctx, int tx_off, uint32_t tx_cost, int psy_sv, int psy_act, GroupRes *res) {
  constexpr int N = 4, nc = 16;
  const int gl = GROUP_LANE, row = gl >> 2, col = gl & 3, bd = f->bd;
  const int dcq = f->dc_q[plane], acq = f->ac_q[plane];
  const uint32_t dc_recip = f->dc_recip[plane], ac_recip = f->ac_recip[plane];
  const int cls = tx_class_of(txtype), pt = plane > 0, txs_ctx = txs;
  const uint32_t dc_off = (uint32_t)(dcq * 109 / 256), off0 = (uint32_t)(acq * 98 / 256), off1 = (uint32_t)(acq * 109 / 256), off_eob = (uint32_t)(acq * 88 / 256);
  const uint32_t thr = (uint32_t)acq - off_eob, uq = (uint32_t)acq;
  // scan index of raster position gl: inverse of av1_default_scan_4x4 (2-D), the position itself (row scan), the transposed position (column scan)
  const int c = cls == TXC_2D ? lut4(0xFEA9DB83C7426510ULL, gl) : (cls == TXC_VERT ? gl : col * 4 + row);
  const uint32_t mag = (uint32_t)iabs_(coef); const int neg = coef < 0;
  const int last = row_max_i32((c >= 1 && mag >= thr) ? c + 1 : 0);
  int l0 = 0;
  if (gl == 0) {                                                   // the DC level (scan position 0 = raster position 0 in every scan)
    const uint32_t x0 = mag + dc_off;
    uint32_t l0u = __umulhi(x0, dc_recip);
    if (x0 - l0u * (uint32_t)dcq >= (uint32_t)dcq) l0u++;
    l0 = (int)l0u;
  }
  l0 = row_max_i32(l0);
  int eob = last;
  if (eob == 0) eob = l0 ? 1 : 0;
  int lv = 0;
  if (c < eob) {
    if (c == 0) lv = l0;
    else {
      uint32_t lv0 = __umulhi(mag, ac_recip);
      if (mag - lv0 * uq >= uq) lv0++;
      const uint32_t off = lv0 > 0 ? off1 : off0;
      lv = (int)lv0 + ((mag + off) >= (lv0 + 1) * uq);
    }
  }
  gb->qc[gl] = neg ? -lv : lv;
  int dq;
  {
    const int dmx = (1 << (7 + bd)) - 1, dmn = -(1 << (7 + bd));
    uint32_t m = (uint32_t)lv * (uint32_t)(gl == 0 ? dcq : acq);
    m &= 0xFFFFFF;
    const int v = neg ? -(int)m : (int)m;
    dq = v < dmn ? dmn : (v > dmx ? dmx : v);
  }
  // neighbours' levels (clipped at 15: the contexts clip at 3 or 15), zero beyond the block
  const int l15 = imin_(lv, 15);
  // (every lane executes every shift -- a lane that skipped one would not lend its value either -- and masks what wrapped into the next row of the block afterwards)
  const int s1 = dpp_row_z<0x101>(l15), s2 = dpp_row_z<0x102>(l15), s3 = dpp_row_z<0x103>(l15), s5 = dpp_row_z<0x105>(l15);
  const int b1 = dpp_row_z<0x104>(l15), b2 = dpp_row_z<0x108>(l15), b3 = dpp_row_z<0x10C>(l15);
  const int r1 = col < 3 ? s1 : 0, r2 = col < 2 ? s2 : 0, r3 = col < 1 ? s3 : 0, d1 = col < 3 ? s5 : 0;
  uint32_t head = cc.txb[(txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE + (eob == 0)];
  int bits = 0, cul = 0, dcc = 0;
  if (eob > 0) {
    if (tx_off >= 0) head += tx_cost;
    const int eob_pt = eob_to_pt(eob);
    head += cc.eobpt[0][(pt * 2 + (cls == TXC_2D ? 0 : 1)) * CDF_EOB_PT_16_STRIDE + eob_pt - 1];
    if (eob_pt >= 3) {
      const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
      head += cc.eobx[((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE + hi];
      head += 512u * (uint32_t)(nb - 1);
    }
    if (c < eob) {
      const int level = lv;
      if (c == eob - 1) {
        const int ctx = c == 0 ? 0 : (c <= nc / 8 ? 1 : (c <= nc / 4 ? 2 : 3));
        bits += cc.beob[((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE + imin_(level, 3) - 1];
      } else {
        // base_ctx (dev_rate.h) on the fetched neighbours
        int mag2 = imin_(r1, 3) + imin_(b1, 3), ctx;
        if (cls == TXC_2D) {
          mag2 += imin_(d1, 3) + imin_(r2, 3) + imin_(b2, 3);
          const int m = imin_((mag2 + 1) >> 1, 4);
          ctx = (row == 0 && col == 0) ? 0 : (row + col < 2 ? m + 1 : (row + col < 4 ? m + 6 : m + 21));
        } else if (cls == TXC_VERT) {
          mag2 += imin_(b2, 3) + imin_(b3, 3);                    // (the fifth tap lies four rows below: outside a 4x4 block)
          const int m = imin_((mag2 + 1) >> 1, 4);
          ctx = m + (row == 0 ? 26 : (row == 1 ? 31 : 36));
        } else {
          mag2 += imin_(r2, 3) + imin_(r3, 3);
          const int m = imin_((mag2 + 1) >> 1, 4);
          ctx = m + (col == 0 ? 26 : (col == 1 ? 31 : 36));
        }
        bits += cc.base[((txs_ctx * 2 + pt) * 42 + ctx) * CDF_COEFF_BASE_STRIDE + imin_(level, 3)];
      }
      if (level > 2) {
        // br_ctx (dev_rate.h)
        int mg = r1 + b1, ctx;
        if (cls == TXC_2D) { mg = imin_((mg + d1 + 1) >> 1, 6); ctx = c == 0 ? mg : ((row < 2 && col < 2) ? mg + 7 : mg + 14); }
        else if (cls == TXC_HORIZ) { mg = imin_((mg + r2 + 1) >> 1, 6); ctx = c == 0 ? mg : (col == 0 ? mg + 7 : mg + 14); }
        else { mg = imin_((mg + b2 + 1) >> 1, 6); ctx = c == 0 ? mg : (row == 0 ? mg + 7 : mg + 14); }
        const int off = ((imin_(txs_ctx, 3) * 2 + pt) * 21 + ctx) * CDF_COEFF_BR_STRIDE;
        int rem = level - 3;
        for (int idx = 0; idx < 4; idx++) { const int s2 = imin_(rem, 3); bits += cc.br[off + s2]; rem -= s2; if (s2 < 3) break; }
      }
      if (level) {
        if (c == 0) { bits += cc.dcs[(pt * 3 + dc_ctx) * CDF_DC_SIGN_STRIDE + neg]; dcc = neg ? 1 : 2; }
        else bits += 512;
        if (level > 14) { const int len = 32 - __clz(level - 14); bits += 512 * (2 * len - 1); }
      }
      cul += level;
    }
  }
  bits = row_sum_i32(bits);
  // the level sum (the context keeps min(sum, 63): clamping each lane's share to 63 first gives the same value) and the dc sign class (set by the one lane that
  // holds scan position 0) share one row reduction
  const int cd = row_sum_i32(imin_(cul, 63) | (dcc << 16));
  res->eob = eob; res->cul = imin_(cd & 0xFFFF, 63); res->dcc = cd >> 16; res->rate = head + (uint32_t)bits;
  // inverse transform + reconstruction (this lane's sample sits at the transposed index, see tx4_tab)
  if (eob > 0) {
    const int cbits = imax_(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
    int v = iclamp_(tx4_inv(dq, rk), cmin, cmax);                  // row pass (no intermediate shift at this size)
    v = group_transpose4(v);
    t_rec = iclamp_(t_rec + round2_(tx4_inv(v, ck), 4), 0, (1 << bd) - 1);
  }
  gb->rec[tidx] = (uint16_t)t_rec;
  WAVE_SYNC();
  const int d = t_src - t_rec;
  const int s = row_sum_i32(__mul24(d, d));
  if (psy_sv >= 0) { const int sd = row_sum_i32(t_rec), qd = row_sum_i32(__mul24(t_rec, t_rec)); res->sse = psy_cell_dist((uint32_t)s, (uint32_t)sd, (uint32_t)qd, (uint32_t)psy_sv, (uint32_t)psy_act, 4, bd); }
  else res->sse = (int)(((unsigned long long)(uint32_t)s * (uint32_t)psy_act + 8192) >> 14);
}

// All arguments may differ between the four groups of the wave (they are uniform inside a group).  `live` = false groups
// run the same code on their own buffers with a harmless input (keeps the wave converged); their result is ignored.
template <int N, typename CostPtr>
__device__ inline void eval_group(const CoefCost &cc, CostPtr cost, const LDS uint16_t *ls, const LDS FrameDev *f, LDS GroupBuf8 *gb,
                                  const LDS uint16_t *src, const LDS uint16_t *pred, int plane, int txs, int txtype,
                                  int skip_ctx, int dc_ctx, int tx_off, int tx_sym, int psy_sv, int psy_act, GroupRes *res) {
  constexpr int P = N + 1, nc = N * N, IT = nc / 16, bwl = N == 4 ? 2 : 3, st = N + 4;
  const int gl = GROUP_LANE;
  const int bd = f->bd;
  const uint32_t tx_cost = cost[tx_off >= 0 ? tx_off + tx_sym : 0];     // the one global-memory operand of the evaluation: issued first, consumed after the transforms
  int ck, rk; tx_kinds(txtype, &ck, &rk);
  [[maybe_unused]] int t_src = 0, t_rec = 0;                        // N == 4: this lane's sample (transposed index, see tx4_tab) lives in registers
  [[maybe_unused]] const int tidx = ((gl & 3) << 2) | (gl >> 2);
  if constexpr (N == 4) {
    t_src = src[tidx]; t_rec = pred[tidx];
    int v = tx4_fwd((int)((uint32_t)(t_src - t_rec) << 2), ck);    // column pass: the quad holds one column
    v = group_transpose4(v);
    const int coef = tx4_fwd(v, rk);                                // row pass: the quad holds one row; the lane's coefficient is raster position gl
    eval_group4_tail(cc, f, gb, coef, t_src, t_rec, tidx, plane, txs, txtype, ck, rk, skip_ctx, dc_ctx, tx_off, tx_cost, psy_sv, psy_act, res);
    return;
  } else {
  // ---- residual, reconstruction seed, level-map reset
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int idx = gl + 16 * k, i = idx / N, j = idx % N;
    const int pv = pred[idx];
    gb->tbuf[i * P + j] = (int)src[idx] - pv;
    gb->rec[idx] = (uint16_t)pv;
  }
  for (int i = gl; i < 36; i += 16) ((LDS uint32_t *)gb->lev)[i] = 0;
  WAVE_SYNC();
  // ---- forward 2-D transform (fwd_txfm2d_dev: shifts {2, 0, 0} for 4x4, {2, -1, 0} for 8x8)
  if (gl < N) {
    int32_t x[N];
#pragma unroll
    for (int r = 0; r < N; r++) x[r] = (int32_t)((uint32_t)gb->tbuf[r * P + gl] << 2);
    tx1d<N>(x, ck, true);
#pragma unroll
    for (int r = 0; r < N; r++) gb->tbuf[r * P + gl] = N == 8 ? rshift_round_(x[r], 1) : x[r];
  }
  WAVE_SYNC();
  if (gl < N) {
    int32_t x[N];
#pragma unroll
    for (int c = 0; c < N; c++) x[c] = gb->tbuf[gl * P + c];
    tx1d<N>(x, rk, true);
#pragma unroll
    for (int c = 0; c < N; c++) gb->cbuf[gl * N + c] = x[c];
  }
  WAVE_SYNC();
  }
  // ---- quantise + level map + dequantise + rate (quant_rate_dev, 16 lanes per candidate)
  const int dcq = f->dc_q[plane], acq = f->ac_q[plane];
  const uint32_t dc_recip = f->dc_recip[plane], ac_recip = f->ac_recip[plane];
  const int cls = tx_class_of(txtype), pt = plane > 0, txs_ctx = txs;
  const uint32_t dc_off = (uint32_t)(dcq * 109 / 256), off0 = (uint32_t)(acq * 98 / 256), off1 = (uint32_t)(acq * 109 / 256), off_eob = (uint32_t)(acq * 88 / 256);
  const uint32_t thr = (uint32_t)acq - off_eob, uq = (uint32_t)acq;
  int pos[IT]; uint32_t mag[IT]; int neg[IT];
  int last = 0;
#pragma unroll
  for (int k = 0; k < IT; k++) {
    const int i = gl + 16 * k;
    pos[k] = scan_pos(ls, N, cls, i);
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
    gb->lev[(pp >> bwl) * st + (pp & (N - 1))] = (uint8_t)imin_(lv, 127);
    uint32_t m = (uint32_t)lv * (uint32_t)(pp == 0 ? dcq : acq);
    m &= 0xFFFFFF;
    const int v = neg[k] ? -(int)m : (int)m;
    gb->cbuf[pp] = v < dmn ? dmn : (v > dmx ? dmx : v);
  }
  WAVE_SYNC();
  uint32_t head = cc.txb[(txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE + (eob == 0)];
  int bits = 0, cul = 0, dcc = 0;
  if (eob > 0) {
    if (tx_off >= 0) head += tx_cost;
    const int eob_pt = eob_to_pt(eob);
    head += cc.eobpt[N == 4 ? 0 : 1][(pt * 2 + (cls == TXC_2D ? 0 : 1)) * (N == 4 ? CDF_EOB_PT_16_STRIDE : CDF_EOB_PT_64_STRIDE) + eob_pt - 1];
    if (eob_pt >= 3) {
      const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
      head += cc.eobx[((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE + hi];
      head += 512u * (uint32_t)(nb - 1);
    }
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const int c = gl + 16 * k;
      if (c < eob) {
        const int pp = pos[k], row = pp >> bwl, col = pp & (N - 1), level = lvl[k];
        const LDS uint8_t *L = gb->lev + row * st + col;
        if (c == eob - 1) {
          const int ctx = c == 0 ? 0 : (c <= nc / 8 ? 1 : (c <= nc / 4 ? 2 : 3));
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
  }
  bits = row_sum_i32(bits);
  const int cd = row_sum_i32(imin_(cul, 63) | (dcc << 16));   // one reduction for both (see eval_group4_tail); only the group's lane 0 (scan position 0) sets dcc
  res->eob = eob; res->cul = imin_(cd & 0xFFFF, 63); res->dcc = cd >> 16; res->rate = head + (uint32_t)bits;
  // ---- inverse 2-D transform + reconstruction (inv_txfm2d_add_dev), only where the candidate has coefficients
  {
    constexpr int ROWSH = N == 4 ? 0 : 1;
    const int cbits = imax_(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
    if constexpr (N == 4) {
      if (eob > 0) {
        int v = iclamp_(tx4_inv(gb->cbuf[gl], rk), cmin, cmax);    // row pass (ROWSH = 0)
        v = group_transpose4(v);
        t_rec = iclamp_(t_rec + round2_(tx4_inv(v, ck), 4), 0, (1 << bd) - 1);
        gb->rec[tidx] = (uint16_t)t_rec;
      }
      WAVE_SYNC();
    } else {
      const bool act = gl < N && eob > 0;
      if (act) {
        int32_t x[N];
  #pragma unroll
        for (int j = 0; j < N; j++) x[j] = gb->cbuf[gl * N + j];
        tx1d<N>(x, rk, false);
  #pragma unroll
        for (int j = 0; j < N; j++) gb->tbuf[gl * P + j] = iclamp_(round2_(x[j], ROWSH), cmin, cmax);
      }
      WAVE_SYNC();
      const int mx = (1 << bd) - 1;
      if (act) {
        int32_t x[N];
  #pragma unroll
        for (int i = 0; i < N; i++) x[i] = gb->tbuf[i * P + gl];
        tx1d<N>(x, ck, false);
  #pragma unroll
        for (int i = 0; i < N; i++) gb->rec[i * N + gl] = (uint16_t)iclamp_((int)gb->rec[i * N + gl] + round2_(x[i], 4), 0, mx);
      }
      WAVE_SYNC();
    }
  }
  // distortion (Tune::Psychovisual): psy_sv >= 0 = luma, the block is one cdef-dist cell (its SSE boosted by the SSIM-like
  // factor of source / reconstruction variance) x activity; psy_sv < 0 = plain SSE x the activity scale psy_act (Q14)
  int s = 0, sd = 0, qd = 0;
  if constexpr (N == 4) { const int d = t_src - t_rec; s = __mul24(d, d); sd = t_rec; qd = __mul24(t_rec, t_rec); }
  else
#pragma unroll
  for (int k = 0; k < IT; k++) { const int idx = gl + 16 * k; const int rv = gb->rec[idx], d = (int)src[idx] - rv; s += __mul24(d, d); sd += rv; qd += __mul24(rv, rv); }
  s = row_sum_i32(s);                            // <= 64 * 1023^2 < 2^27
  if (psy_sv >= 0) { sd = row_sum_i32(sd); qd = row_sum_i32(qd); res->sse = psy_cell_dist((uint32_t)s, (uint32_t)sd, (uint32_t)qd, (uint32_t)psy_sv, (uint32_t)psy_act, N == 4 ? 4 : 8, bd); }
  else res->sse = (int)(((unsigned long long)(uint32_t)s * (uint32_t)psy_act + 8192) >> 14);
}

// ---------------------------------------------------------------------------------------------------------------
// Directional intra prediction + SATD, four predictions per wavefront (4x4 / 8x8 blocks): each 16-lane row runs one
// prediction angle.  Same steps as predict_block()'s directional branch in dev_predict.h (spec 7.11.2.4: edge copies,
// edge filter, edge upsampling, two-tap interpolation); the corner filter of that branch only exists for blocks of
// 12 samples and more and is therefore absent here.
struct GroupPredBuf { uint16_t wa[48], wl[48], tmp[32], pred[64]; };      // EDGE_OFF + 2*8 + 16 entries per edge

__device__ __forceinline__ void edge_filter_group(LDS uint16_t *buf, int sz, int strength, LDS uint16_t *tmp, int gl) {
  // row-uniform arguments; the syncs are executed by every row (strength 0 rows just pass through)
  if (strength) for (int i = gl; i < sz; i += 16) tmp[i] = buf[i - 1];
  WAVE_SYNC();
  if (strength) for (int i = 1 + gl; i < sz; i += 16) {
    int s = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int k = iclamp_(i - 2 + j, 0, sz - 1);
      const int kw = strength == 1 ? (j == 0 || j == 4 ? 0 : (j == 2 ? 8 : 4)) : (strength == 2 ? (j == 0 || j == 4 ? 0 : (j == 2 ? 6 : 5)) : (j == 0 || j == 4 ? 2 : 4));
      s += kw * tmp[k];
    }
    buf[i - 1] = (uint16_t)((s + 8) >> 4);
  }
  WAVE_SYNC();
}
__device__ __forceinline__ void edge_upsample_group(LDS uint16_t *buf, int num_px, int on, int bd, LDS uint16_t *tmp, int gl) {
  if (on) for (int i = gl; i < num_px + 3; i += 16) {
    uint16_t v;
    if (i == 0) v = buf[-1]; else if (i == num_px + 2) v = buf[num_px - 1]; else v = buf[i - 2];
    tmp[i] = v;
  }
  WAVE_SYNC();
  const int mx = (1 << bd) - 1;
  if (on) {
    if (gl == 0) buf[-2] = tmp[0];
    for (int i = gl; i < num_px; i += 16) {
      int s = -(int)tmp[i] + 9 * (int)tmp[i + 1] + 9 * (int)tmp[i + 2] - (int)tmp[i + 3];
      s = iclamp_(round2_(s, 4), 0, mx);
      buf[2 * i - 1] = (uint16_t)s;
      buf[2 * i] = tmp[i + 2];
    }
  }
  WAVE_SYNC();
}

// pa: prediction angle of this row (mode angle + 3 * delta), row-uniform.  Rows with live == false still walk the code
// (pa = 90 keeps them cheap).  Output: gp->pred[N*N], or `out` (row-uniform) when given.
template <int N>
__device__ inline void predict_dir_group(const LDS FrameDev *f, int x, int y, int have_left, int have_above, int pa, int ftype,
                                         const LDS uint16_t *ra, const LDS uint16_t *rl, LDS GroupPredBuf *gp, LDS uint16_t *out = nullptr) {
  constexpr int log2w = N == 4 ? 2 : 3, nn = N * N;
  const int gl = GROUP_LANE, bd = f->bd;
  const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1;
  LDS uint16_t *wa = gp->wa + EDGE_OFF, *wl = gp->wl + EDGE_OFF;
  for (int i = gl; i < 2 * N + 1; i += 16) { wa[i - 1] = ra[i - 1]; wl[i - 1] = rl[i - 1]; }
  WAVE_SYNC();
  const int diag = pa != 90 && pa != 180;
  {
    const int st_a = (diag && have_above) ? edge_strength_dev(N, N, ftype, pa - 90) : 0;
    const int num_a = imin_(N, max_x - x + 1) + (pa < 90 ? N : 0) + 1;
    edge_filter_group(wa, num_a, st_a, gp->tmp, gl);
    const int st_l = (diag && have_left) ? edge_strength_dev(N, N, ftype, pa - 180) : 0;
    const int num_l = imin_(N, max_y - y + 1) + (pa > 180 ? N : 0) + 1;
    edge_filter_group(wl, num_l, st_l, gp->tmp, gl);
  }
  const int up_a = edge_upsample_sel_dev(N, N, ftype, pa - 90);
  edge_upsample_group(wa, N + (pa < 90 ? N : 0), up_a, bd, gp->tmp, gl);
  const int up_l = edge_upsample_sel_dev(N, N, ftype, pa - 180);
  edge_upsample_group(wl, N + (pa > 180 ? N : 0), up_l, bd, gp->tmp, gl);
  int dx = 0, dy = 0;
  if (pa < 90) dx = dr_deriv_dev(pa); else if (pa > 90 && pa < 180) dx = dr_deriv_dev(180 - pa);
  if (pa > 90 && pa < 180) dy = dr_deriv_dev(pa - 90); else if (pa > 180) dy = dr_deriv_dev(270 - pa);
  for (int idx = gl; idx < nn; idx += 16) {
    const int i = idx >> log2w, j = idx & (N - 1);
    int v;
    if (pa < 90) {
      const int max_base = (2 * N - 1) << up_a;
      const int id = (i + 1) * dx, base = (id >> (6 - up_a)) + (j << up_a), sh = ((id << up_a) >> 1) & 0x1F;
      v = base < max_base ? round2_(wa[base] * (32 - sh) + wa[base + 1] * sh, 5) : wa[max_base];
    } else if (pa > 90 && pa < 180) {
      int id = (j << 6) - (i + 1) * dx, base = id >> (6 - up_a);
      if (base >= -(1 << up_a)) { const int sh = ((id << up_a) >> 1) & 0x1F; v = round2_(wa[base] * (32 - sh) + wa[base + 1] * sh, 5); }
      else { id = (i << 6) - (j + 1) * dy; base = id >> (6 - up_l); const int sh = ((id << up_l) >> 1) & 0x1F; v = round2_(wl[base] * (32 - sh) + wl[base + 1] * sh, 5); }
    } else if (pa > 180) {
      const int id = (j + 1) * dy, base = (id >> (6 - up_l)) + (i << up_l), sh = ((id << up_l) >> 1) & 0x1F;
      v = round2_(wl[base] * (32 - sh) + wl[base + 1] * sh, 5);
    } else if (pa == 90) v = wa[j];
    else v = wl[i];
    (out ? out : (LDS uint16_t *)gp->pred)[idx] = (uint16_t)v;
  }
  WAVE_SYNC();
}

// The five non-directional predictions (DC, PAETH, SMOOTH, SMOOTH_V, SMOOTH_H) of a 4x4 / 8x8 block, one per 16-lane row like the directional ones above: the branches of
// predict_block() for these modes with the same arithmetic, on the block's raw edges.  `mode` is row-uniform; a row with mode < 0 idles.  The smooth weights
// (spec Sm_Weights_Tx_4x4 / 8x8) are packed immediates instead of a table in memory.  Until round 5 these five ran one after the other through the generic
// one-block-per-wave predictor on the two waves that had no directional rows: the longest chain of the SATD stage.
template <int N>
__device__ inline void predict_nondir_group(int mode, int have_left, int have_above, int bd, const LDS uint16_t *ra, const LDS uint16_t *rl, LDS uint16_t *pred) {
  constexpr int log2w = N == 4 ? 2 : 3, IT = N * N / 16;
  constexpr unsigned long long wt = N == 4 ? 0x405595FFULL : 0x202532496992C5FFULL;     // { 255, 149, 85, 64 } / { 255, 197, 146, 105, 73, 50, 37, 32 }
  const int gl = GROUP_LANE;
  if (mode == DC_PRED) {
    int v = 1 << (bd - 1);
    if (have_left || have_above) {
      int s = gl < N ? (have_above ? (int)ra[gl] : 0) + (have_left ? (int)rl[gl] : 0) : 0;
      s = row_sum_i32(s);
      v = (have_left && have_above) ? (s + N) >> (log2w + 1) : (s + (N >> 1)) >> log2w;
    }
#pragma unroll
    for (int k = 0; k < IT; k++) pred[gl + 16 * k] = (uint16_t)v;
  } else if (mode == PAETH_PRED) {
    const int tl = ra[-1];
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const int idx = gl + 16 * k, i = idx >> log2w, j = idx & (N - 1);
      const int a = ra[j], l = rl[i], base = a + l - tl;
      const int pl = iabs_(base - l), pt = iabs_(base - a), ptl = iabs_(base - tl);
      pred[idx] = (uint16_t)((pl <= pt && pl <= ptl) ? l : (pt <= ptl ? a : tl));
    }
  } else if (mode >= 0) {
    const int bl = rl[N - 1], tr = ra[N - 1];
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const int idx = gl + 16 * k, i = idx >> log2w, j = idx & (N - 1);
      const int wi = lut8(wt, i), wj = lut8(wt, j);
      const int vt = wi * (int)ra[j] + (256 - wi) * bl, ht = wj * (int)rl[i] + (256 - wj) * tr;
      pred[idx] = (uint16_t)(mode == SMOOTH_PRED ? round2_(vt + ht, 9) : round2_(mode == SMOOTH_V_PRED ? vt : ht, 8));
    }
  }
  WAVE_SYNC();
}

// SATD of one row's prediction (satd_dev with 16 lanes): N == 4: one 4x4 Hadamard (one lane per column); N == 8: the 8x8 Hadamard of the block, four samples per lane
// (satd8_lane, tile_search.h), on the scale of four 4x4 ones
__device__ __forceinline__ int satd8_lane(int d0, int d1, int d2, int d3);
template <int N> __device__ inline int satd_group(const LDS uint16_t *src, const LDS uint16_t *pred) {
  constexpr int units = N * (N / 4);              // 16 for 8x8, 4 for 4x4: whole quads
  const int u = GROUP_LANE;
  if constexpr (N == 8) {
    const int xx = (u & 3) + ((u >> 1) & 4), o = (u & 4) * N + xx;
    const int d0 = (int)src[o] - (int)pred[o], d1 = (int)src[o + N] - (int)pred[o + N];
    const int d2 = (int)src[o + 2 * N] - (int)pred[o + 2 * N], d3 = (int)src[o + 3 * N] - (int)pred[o + 3 * N];
    return (row_sum_i32(satd8_lane(d0, d1, d2, d3)) + 2) >> 2;
  }
  int s = 0;
  if (u < units) {
    const int xx = u % N, o = (u / N) * 4 * N + xx;
    const int d0 = (int)src[o] - (int)pred[o], d1 = (int)src[o + N] - (int)pred[o + N];
    const int d2 = (int)src[o + 2 * N] - (int)pred[o + 2 * N], d3 = (int)src[o + 3 * N] - (int)pred[o + 3 * N];
    const int a = d0 + d1, b = d0 - d1, c = d2 + d3, e = d2 - d3;
    int t[4] = { a + c, b + e, a - c, b - e };
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int v = __builtin_amdgcn_update_dpp(0, t[i], 0xB1, 0xF, 0xF, false);
      v = (LANE & 1) ? v - t[i] : t[i] + v;
      int w2 = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
      v = (LANE & 2) ? w2 - v : v + w2;
      s += iabs_(v);
    }
  }
  return row_sum_i32(s);
}

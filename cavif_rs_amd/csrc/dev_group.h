// dev_group.h -- four transform-block evaluations per wavefront, for 4x4 and 8x8 blocks.
// A 4x4 / 8x8 evaluation keeps at most 16 lanes busy (8 columns, 64 coefficients), so the one-candidate-per-wave
// scheme of tile_search.h ran four such evaluations one after the other on a wave that was three quarters idle.
// Here each 16-lane DPP row ("group") owns one candidate: same arithmetic, same rounding points, same results as
// eval_tx()/quant_rate_dev() -- only the lane mapping differs: residual, quantiser, rate and SSE use 16 lanes per
// candidate, the 1-D networks use lane l < N of the group as column / row l, reductions stay inside the row (DPP
// row_ror all-reduce).  Everything that is wave-uniform in the single-candidate code is row-uniform here.
#pragma once
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

struct GroupRes { int eob, cul, dcc, sse; uint32_t rate; };

// All arguments may differ between the four groups of the wave (they are uniform inside a group).  `live` = false groups
// run the same code on their own buffers with a harmless input (keeps the wave converged); their result is ignored.
template <int N, typename CostPtr>
__device__ inline void eval_group(const CoefCost &cc, CostPtr cost, const LDS uint16_t *ls, const LDS FrameDev *f, LDS GroupBuf8 *gb,
                                  const LDS uint16_t *src, const LDS uint16_t *pred, int plane, int txs, int txtype,
                                  int skip_ctx, int dc_ctx, int tx_off, int tx_sym, GroupRes *res) {
  constexpr int P = N + 1, nc = N * N, IT = nc / 16, bwl = N == 4 ? 2 : 3, st = N + 4;
  const int gl = GROUP_LANE;
  const int bd = f->bd;
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
  int ck, rk; tx_kinds(txtype, &ck, &rk);
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
    if (tx_off >= 0) head += cost[tx_off + tx_sym];
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
  cul = row_sum_i32(imin_(cul, 1 << 20));
  dcc = row_max_i32(dcc);                        // only the group's lane 0 (scan position 0) sets it
  res->eob = eob; res->cul = imin_(cul, 63); res->dcc = dcc; res->rate = head + (uint32_t)bits;
  // ---- inverse 2-D transform + reconstruction (inv_txfm2d_add_dev), only where the candidate has coefficients
  {
    constexpr int ROWSH = N == 4 ? 0 : 1;
    const int cbits = imax_(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
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
  int s = 0;
#pragma unroll
  for (int k = 0; k < IT; k++) { const int idx = gl + 16 * k; const int d = (int)src[idx] - (int)gb->rec[idx]; s += __mul24(d, d); }
  res->sse = row_sum_i32(s);                     // <= 64 * 1023^2 < 2^27
}

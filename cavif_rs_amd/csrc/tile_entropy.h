// tile_entropy.h -- kernel K4: the tile's final bitstream.  One wavefront per tile walks the mode-info
// maps that K1 left in HBM (partition tree, modes, tx types, quantised levels) in AV1 coding order and
// range-codes them with adaptive CDFs held in LDS (spec 5.11 syntax, 8.2 symbol coder, 8.3 CDF selection).
// The walk is inherently serial per tile; the wave's lanes cooperate on staging each transform block's
// levels + context map into LDS and on the CDF adaptation, lane 0 drives the range coder.
// rav1e equivalents (absent from /root/reference): src/ec.rs (WriterBase), src/context/*.rs.
#pragma once
#include "dev_common.h"
#include "dev_rate.h"
#include "restoration.h"

// Three pipelined forms of this kernel (walker wave + range-coder wave; walker | four CDF-adapter waves | coder through an LDS ring; the same
// three stages as three kernels with the record stream in HBM) were measured on the MI355X in round 3 and all lost against the single wave
// (43.5 ms per 1024 tiles: 44.0 / 46.0 / 75.9 ms, profiles/r03_variants_ab.txt); they are not in the tree.
#define MI_K4_THREADS 64
struct RangeEncDev {
  uint16_t *pre; uint32_t cap, offs;
  uint32_t low; uint32_t rng; int cnt;   // low stays below 2^31: 16 + cnt + 9 + d bits, flushed whenever cnt + d >= 0
};

__device__ __forceinline__ void re_init_dev(RangeEncDev *e, uint16_t *pre, uint32_t cap) {
  e->pre = pre; e->cap = cap; e->offs = 0; e->low = 0; e->rng = 0x8000; e->cnt = -9;
}
__device__ __forceinline__ void re_put16(RangeEncDev *e, uint16_t v) {
  if (e->offs < e->cap && LANE == 0) e->pre[e->offs] = v;   // offs counts every unit, stored or not: overflow <=> offs > cap at the end (re_finish_dev)
  e->offs++;
}
__device__ __forceinline__ void re_normalize_dev(RangeEncDev *e, uint32_t low, uint32_t rng) {
  int c = e->cnt;
  const int d = 16 - (32 - __clz(rng));
  int s = c + d;
  if (s >= 0) {
    c += 16;
    uint32_t m = (1u << c) - 1;
    if (s >= 8) { re_put16(e, (uint16_t)(low >> c)); low &= m; c -= 8; m >>= 8; }
    re_put16(e, (uint16_t)(low >> c));
    s = c + d - 24;
    low &= m;
  }
  e->low = low << d; e->rng = rng << d; e->cnt = s;
}
__device__ __forceinline__ void re_encode_q15_dev(RangeEncDev *e, uint32_t fl, uint32_t fh, int s, int nsyms) {
  uint32_t l = e->low; uint32_t r = e->rng;
  const int N = nsyms - 1;
  if (fl < 32768) {
    const uint32_t u = (((r >> 8) * (fl >> 6)) >> 1) + 4 * (uint32_t)(N - (s - 1));
    const uint32_t v = (((r >> 8) * (fh >> 6)) >> 1) + 4 * (uint32_t)(N - s);
    l += r - u; r = u - v;
  } else {
    r -= (((r >> 8) * (fh >> 6)) >> 1) + 4 * (uint32_t)(N - s);
  }
  re_normalize_dev(e, l, r);
}
// encode + adapt.  The coder state (low, rng, cnt, offs) is wave-uniform and every operand that reaches it goes through
// v_readfirstlane / v_readlane, so the compiler keeps it in SGPRs and the range arithmetic runs on the scalar unit.  The CDF row
// is read ONCE, lane i holding entry i (entry nsyms = the adaptation counter): the symbol's two bounds and the counter come
// out of that register by v_readlane, the same register feeds lane i's adaptation (spec 8.3.2 update rule, one step instead
// of a loop) -- one LDS round trip per symbol instead of three dependent ones.
__device__ __forceinline__ void re_symbol_dev(RangeEncDev *e, int s_in, LDS uint16_t *icdf, int nsyms_in) {
  const int s = uni32(s_in), nsyms = uni32(nsyms_in);
  const int i = LANE;
  const int v = icdf[imin_(i, nsyms)];
  const uint32_t fl0 = (uint32_t)__builtin_amdgcn_readlane(v, imax_(s - 1, 0)), fh = (uint32_t)__builtin_amdgcn_readlane(v, s);
  const int cnt = __builtin_amdgcn_readlane(v, nsyms);
  re_encode_q15_dev(e, s > 0 ? fl0 : 32768u, fh, s, nsyms);
  const int rate = 3 + (cnt > 15) + (cnt > 31) + imin_((32 - __clz(nsyms)) - 1, 2);
  if (i < nsyms - 1) icdf[i] = (uint16_t)(i < s ? v + ((32768 - v) >> rate) : v - (v >> rate));
  else if (i == nsyms) icdf[nsyms] = (uint16_t)(cnt + (cnt < 32));
  WAVE_SYNC();
}
__device__ __forceinline__ void re_bool_dev(RangeEncDev *e, int bit_in, uint32_t icdf0) {
  const int bit = uni32(bit_in);
  re_encode_q15_dev(e, bit ? icdf0 : 32768, bit ? 0 : icdf0, bit, 2);
}
__device__ __forceinline__ void re_literal_dev(RangeEncDev *e, uint32_t v_in, int nbits_in) {
  const uint32_t v = (uint32_t)uni32((int)v_in); const int nbits = uni32(nbits_in);
  for (int i = nbits - 1; i >= 0; i--) re_bool_dev(e, (int)((v >> i) & 1), 16384);
}
// returns number of bytes; out must hold them.  (lane 0)
__device__ __forceinline__ uint32_t re_finish_dev(RangeEncDev *e, uint8_t *out, uint32_t out_cap) {
  unsigned long long l = e->low; int c = e->cnt; int s = 10;
  const unsigned long long m = 0x3FFF;
  unsigned long long x = ((l + m) & ~m) | (m + 1);
  s += c;
  if (s > 0) {
    unsigned long long n = (1ULL << (c + 16)) - 1;
    do {
      const uint16_t u16 = (uint16_t)(x >> (c + 16));
      re_put16(e, u16);
      x &= n; s -= 8; c -= 8; n >>= 8;
    } while (s > 0);
  }
  const uint32_t nb = e->offs;
  if (nb > e->cap || nb > out_cap) return 0xFFFFFFFFu;
  uint32_t carry = 0;
  for (uint32_t i = nb; i-- > 0;) { carry = e->pre[i] + carry; out[i] = (uint8_t)carry; carry >>= 8; }
  return nb;
}

struct TileWriter {
  const FrameDev *f; TileB t; RangeEncDev ec; LDS uint16_t *cdf;   // cdf: LDS [CDF_TOTAL]
  LDS int32_t *qc; LDS uint8_t *lev; const LDS uint16_t *ls;       // LDS staging + LDS copy of the scan tables
  LDS uint16_t *rec_off, *rec_br; LDS uint32_t *rec_lv;            // per-coefficient records of the current transform block
  LDS uint16_t *lr_cdf; LDS int *lr_ref;                           // switchable restoration_type CDF (3 symbols + counter), RefSgrXqd[plane][2]
  int cdef_pending;                                                 // the 64x64 superblock being walked has not signalled its cdef_idx yet
  int sb_cols_tile;
  // Frame scalars that steer the walk, pinned to SGPRs once per tile (MI_K4_UNIFORM): a value that reaches a branch through a vector
  // load is "divergent" to the compiler, the coder state behind such a branch becomes a per-lane value and the range arithmetic moves
  // to the vector unit under exec masks.  Every control value the walk loads (skip, modes, transform sizes, eob, block sizes,
  // restoration types) goes through v_readfirstlane for the same reason.
  int np, mi_rows, mi_cols, ms, tx_mode_select, enable_cdef, cdef_bits, enable_restoration, sb_cols, fw, fh;
  struct TxCfg { int reduced_tx_set, base_q_idx; } txc;
#if MI_PROFILE == 2
  unsigned long long prof[16], pt;
#endif
};
// K4 phase timers (probe builds, -DMI_PROFILE=2): cycles per phase and event counts, flushed into wave 3's slots of the tile's K1 record
#ifndef MI_PROFILE
#define MI_PROFILE 0
#endif
#if MI_PROFILE == 2
#define K4PH(i) do { const unsigned long long n_ = clock64(); w->prof[i] += n_ - w->pt; w->pt = n_; } while (0)
#define K4CNT(i, n) do { w->prof[i] += (unsigned long long)(n); } while (0)
#else
#define K4PH(i) do {} while (0)
#define K4CNT(i, n) do {} while (0)
#endif
#ifndef MI_K4_UNIFORM
#define MI_K4_UNIFORM 1
#endif
#if MI_K4_UNIFORM
#define U_(v) uni32((int)(v))
#else
#define U_(v) ((int)(v))
#endif

// Code one transform block's coefficients (levels + padded level map already staged in LDS).
// Two phases: (P) every lane derives the CDF rows (contexts) of its own scan positions -- they depend only on the
// level map, not on the coder state -- and leaves (cdf offset, level, sign) records in LDS; (S) the wave walks
// the records in coding order and drives the adaptive range coder, wave-uniform.
__device__ __forceinline__ void code_coeffs_lane0(TileWriter *w, int eob_in, int plane, int txs, int txtype, int skip_ctx, int dc_ctx,
                                                  int tx_off, int tx_sym, int tx_ns) {
  const int eob = uni32(eob_in);
  RangeEncDev *e = &w->ec; LDS uint16_t *cdf = w->cdf; const LDS int32_t *qc = w->qc; const LDS uint8_t *lev = w->lev;
#if MI_RECT_PART
  const bool rect = txs > 4;                              // 5 = 4x8, 6 = 8x4 (dev_rect.h)
  const int bwl = rect ? (txs == 5 ? 2 : 3) : imin_(5, 2 + txs), bhl = rect ? (txs == 5 ? 3 : 2) : bwl, n = 1 << bwl, nh = 1 << bhl;
  const int pt = plane > 0, cls = tx_class_of(txtype), txs_ctx = rect ? 1 : txs;
#else
  const int n = imin_(32, 4 << txs), bwl = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5, nh = n, bhl = bwl;
  const int pt = plane > 0, cls = tx_class_of(txtype), txs_ctx = txs;
#endif
  re_symbol_dev(e, eob == 0, cdf + CDF_TXB_SKIP + (txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE, 2);
  K4CNT(9, 1); K4CNT(10, eob == 0);
  if (eob == 0) { K4PH(3); return; }
  // ---- (P) contexts, lane-parallel
  const int st = n + 4, area = n * nh;
  for (int c = LANE; c < eob; c += 64) {
#if MI_RECT_PART
    const int p = rect ? rect_scan_pos(bwl, bhl, cls, c) : scan_pos(w->ls, n, cls, c), row = p >> bwl, col = p & (n - 1);
#else
    const int p = scan_pos(w->ls, n, cls, c), row = p >> bwl, col = p & (n - 1);
#endif
    const int v = qc[p], level = iabs_(v);
    const LDS uint8_t *L = lev + row * st + col;
    int off;                                                 // last position: its base_eob CDF row; the others: the base context (0..41)
    if (c == eob - 1) {
      const int ctx = c == 0 ? 0 : (c <= area / 8 ? 1 : (c <= area / 4 ? 2 : 3));
      off = CDF_COEFF_BASE_EOB + ((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE;
    } else {
      int bctx = base_ctx(L, st, cls, row, col);
#if MI_RECT_PART
      if (rect && cls == TXC_2D && !(row == 0 && col == 0)) {          // spec Coeff_Base_Ctx_Offset of the 2:1 sizes
        const int mg = imin_(L[1], 3) + imin_(L[st], 3) + imin_(L[st + 1], 3) + imin_(L[2], 3) + imin_(L[2 * st], 3), m = imin_((mg + 1) >> 1, 4);
        bctx = bhl > bwl ? m + (row < 2 ? 11 : (row + col < 4 ? 6 : 21)) : m + (col < 2 ? 16 : (row + col < 4 ? 6 : 21));
      }
#endif
      off = CDF_COEFF_BASE + ((txs_ctx * 2 + pt) * 42 + bctx) * CDF_COEFF_BASE_STRIDE;
    }
    int boff = 0;
    if (level > 2) boff = CDF_COEFF_BR + ((imin_(txs_ctx, 3) * 2 + pt) * 21 + br_ctx(L, st, cls, row, col, c)) * CDF_COEFF_BR_STRIDE;
    w->rec_off[c] = (uint16_t)off; w->rec_br[c] = (uint16_t)boff; w->rec_lv[c] = ((uint32_t)level << 1) | (uint32_t)(v < 0);
  }
  WAVE_SYNC();
  K4PH(3);
  // ---- (S) serial coding
  if (tx_off >= 0) re_symbol_dev(e, tx_sym, cdf + tx_off, tx_ns);
  const int eob_pt = eob_to_pt(eob), eob_multi = bwl + bhl - 4;
  re_symbol_dev(e, eob_pt - 1, cdf + eob_pt_cdf(eob_multi, pt, cls), 5 + eob_multi);
  if (eob_pt >= 3) {
    const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
    re_symbol_dev(e, hi, cdf + CDF_EOB_EXTRA + ((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE, 2);
    if (nb > 1) re_literal_dev(e, (uint32_t)rem & ((1u << (nb - 1)) - 1), nb - 1);
  }
  // the records of 64 scan positions at a time sit in a register (lane j = position cb + j) and are picked by v_readlane
  for (int cb = (eob - 1) & ~63; cb >= 0; cb -= 64) {
    const int li = imin_(cb + LANE, eob - 1);
    const uint32_t r_lv = w->rec_lv[li]; const int r_off = w->rec_off[li], r_br = w->rec_br[li];
    for (int c = imin_(eob - 1, cb + 63); c >= cb; c--) {
      const int j = c - cb;
      const int level = (int)((uint32_t)__builtin_amdgcn_readlane((int)r_lv, j) >> 1);
      const int boff = __builtin_amdgcn_readlane(r_off, j);
      if (c == eob - 1) re_symbol_dev(e, imin_(level, 3) - 1, cdf + boff, 3);
      else re_symbol_dev(e, imin_(level, 3), cdf + boff, 4);
      if (level > 2) {
        const int bl = __builtin_amdgcn_readlane(r_br, j);
        int rem = level - 3;
        for (int idx = 0; idx < 4; idx++) { const int s = imin_(rem, 3); re_symbol_dev(e, s, cdf + bl, 4); rem -= s; if (s < 3) break; }
      }
    }
  }
  K4PH(4); K4CNT(11, eob);
  for (int cb = 0; cb < eob; cb += 64) {
    const uint32_t r_lv = w->rec_lv[imin_(cb + LANE, eob - 1)];
    for (int c = cb; c < imin_(eob, cb + 64); c++) {
      const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)r_lv, c - cb); const int a = (int)(m >> 1), neg = (int)(m & 1);
      if (a) {
        if (c == 0) re_symbol_dev(e, neg, cdf + CDF_DC_SIGN + (pt * 3 + dc_ctx) * CDF_DC_SIGN_STRIDE, 2);
        else re_bool_dev(e, neg, 16384);
        if (a > 14) { const uint32_t xg = (uint32_t)(a - 14); const int len = 32 - __clz(xg); re_literal_dev(e, 0, len - 1); re_literal_dev(e, xg, len); }
      }
    }
  }
  K4PH(5);
}

template <int BS> __device__ __forceinline__ void write_block_dev(TileWriter *w, int r, int c) {
  const FrameDev *f = w->f; const TileB *t = &w->t; const int ms = w->ms, mi = r * ms + c;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  // The block's mode info and its neighbours' in ONE batch of unconditional loads (a neighbour outside the tile reads the block's
  // own cell and is replaced by its default afterwards): as `avail ? map[..] : dflt` at the point of use each of them was a
  // separate round trip to L2 on the single wave that codes the tile.
  const int iU = availU ? mi - ms : mi, iL = availL ? mi - 1 : mi;
  const int l_skip = f->m_skip[mi], l_ymode = f->m_ymode[mi], l_txs_y = f->m_txsize[mi];
  const int l_skU = f->m_skip[iU], l_skL = f->m_skip[iL], l_ymU = f->m_ymode[iU], l_ymL = f->m_ymode[iL], l_txU = f->m_txsize[iU], l_txL = f->m_txsize[iL];
  const int l_ay = f->m_angle_y[mi], l_cdef = f->cdef_idx[(r >> 4) * w->sb_cols + (c >> 4)];
  int l_uvmode = 0, l_auv = 0, l_js = 0, l_au = 0, l_av = 0;
  if (w->np > 1) { l_uvmode = f->m_uvmode[mi]; l_auv = f->m_angle_uv[mi]; l_js = f->m_cfl_sign[mi]; l_au = f->m_cfl_au[mi]; l_av = f->m_cfl_av[mi]; }
  const int skip = U_(l_skip), ymode = U_(l_ymode), txs_y = U_(l_txs_y), v_skU = U_(l_skU), v_skL = U_(l_skL), v_ymU = U_(l_ymU), v_ymL = U_(l_ymL);
  const int v_txU = U_(l_txU), v_txL = U_(l_txL), v_ay = U_(l_ay), v_cdef = U_(l_cdef);
  const int uvmode = U_(l_uvmode), v_auv = U_(l_auv), v_js = U_(l_js), v_au = U_(l_au), v_av = U_(l_av);
  {
    RangeEncDev *e = &w->ec; LDS uint16_t *cdf = w->cdf;
    const int sctx = (availU ? v_skU : 0) + (availL ? v_skL : 0);
    re_symbol_dev(e, skip, cdf + CDF_SKIP + sctx * CDF_SKIP_STRIDE, 2);
    if (!skip && w->enable_cdef) {
      if (w->cdef_pending) { w->cdef_pending = 0; re_literal_dev(e, (uint32_t)v_cdef, w->cdef_bits); }   // first non-skip block of the superblock (spec 5.11.56)
    }
    const int am = intra_mode_ctx(availU ? v_ymU : DC_PRED), lm = intra_mode_ctx(availL ? v_ymL : DC_PRED);
    re_symbol_dev(e, ymode, cdf + CDF_KF_Y + (am * 5 + lm) * CDF_KF_Y_STRIDE, 13);
    if (BS >= BS_8 && ymode >= V_PRED && ymode <= D67_PRED)
      re_symbol_dev(e, v_ay + 3, cdf + CDF_ANGLE + (ymode - V_PRED) * CDF_ANGLE_STRIDE, 7);
    if (w->np > 1) {
      const int um = uvmode;
      if (BS <= BS_32) re_symbol_dev(e, um, cdf + CDF_UV_CFL + ymode * CDF_UV_CFL_STRIDE, 14);
      else re_symbol_dev(e, um, cdf + CDF_UV_NOCFL + ymode * CDF_UV_NOCFL_STRIDE, 13);
      if (um == UV_CFL_PRED) {
        const int js = v_js, su = (js + 1) / 3, sv = (js + 1) % 3;
        re_symbol_dev(e, js, cdf + CDF_CFL_SIGN, 8);
        if (su) re_symbol_dev(e, v_au, cdf + CDF_CFL_ALPHA + ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE, 16);
        if (sv) re_symbol_dev(e, v_av, cdf + CDF_CFL_ALPHA + ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE, 16);
      }
      if (BS >= BS_8 && um >= V_PRED && um <= D67_PRED)
        re_symbol_dev(e, v_auv + 3, cdf + CDF_ANGLE + (um - V_PRED) * CDF_ANGLE_STRIDE, 7);
    }
  }
  // read_block_tx_size(): tx_depth of every intra block above 4x4 under TX_MODE_SELECT, coded even when skip
  if (BS > 0 && w->tx_mode_select) {
    const int maxw = 4 << BS;
#if MI_RECT_PART
    const int actx = availU && (1 << dim_wl(v_txU)) >= maxw, lctx = availL && (1 << dim_hl(v_txL)) >= maxw;
#else
    const int actx = availU && (4 << v_txU) >= maxw, lctx = availL && (4 << v_txL) >= maxw;
#endif
    re_symbol_dev(&w->ec, BS - txs_y, w->cdf + CDF_TX_SIZE + ((BS - 1) * 3 + actx + lctx) * CDF_TX_SIZE_STRIDE, BS == 1 ? 2 : 3);
  }
  K4PH(1); K4CNT(8, 1);
  if (skip) return;                                    // wave-uniform
  // residual(): per plane the transform blocks of the block in raster order (luma may be split one level, chroma is not)
  for (int p = 0; p < w->np; p++) {
    const int txs = p == 0 ? txs_y : BS, l2n = imin_(5, 2 + txs), n = 1 << l2n, step = 1 << txs, nblk = 1 << (BS - txs);
    for (int bi = 0; bi < nblk * nblk; bi++) {
      const int rr = r + (bi / nblk) * step, cc = c + (bi % nblk) * step, tmi = rr * ms + cc;
      if (rr >= w->mi_rows || cc >= w->mi_cols) continue;
      const int l_eob = f->m_eob[p][tmi], l_txt = f->m_txtype[tmi];      // issued with the coefficient loads below
      const int32_t *src = f->coef[p] + (size_t)(rr * 4) * f->stride + cc * 4;
      WAVE_SYNC();
      for (int idx = LANE; idx < n * n; idx += 64) w->qc[idx] = src[(idx >> l2n) * f->stride + (idx & (n - 1))];
      WAVE_SYNC();
      build_level_map(w->qc, w->lev, n);
      const int eob = U_(l_eob), v_txt = U_(l_txt);
      int txtype, off = -1, sym = 0, ns = 0, set;
      if (p == 0) {
        txtype = v_txt;
        off = intra_tx_cdf(&w->txc, txs, ymode, &ns, &set);
        if (off >= 0) sym = txtype_to_sym(set, txtype);
      } else {
        set = tx_set_of(txs, w->txc.reduced_tx_set);
        txtype = mode_to_txtype(uvmode);
        if (txtype_to_sym(set, txtype) < 0) txtype = DCT_DCT;
      }
      int sctx2, dctx;
      txb_ctx_dev(f, t, p, rr, cc, txs, BS, &sctx2, &dctx);
      K4PH(2);
      code_coeffs_lane0(w, eob, p, txs, txtype, sctx2, dctx, off, sym, ns);
    }
  }
  WAVE_SYNC();
}

#if MI_RECT_PART
// An 8x4 / 4x8 block (oracle write_block with a 2:1 size): no angle deltas, CfL allowed, tx_depth in the 8x8 category, one 2:1 transform or its two
// 4x4 halves per luma block, one 2:1 transform per chroma plane.
template <int BSR> __device__ __forceinline__ void write_block_rect(TileWriter *w, int r, int c) {
  constexpr int WL = BSR == BS_4X8 ? 2 : 3, HL = BSR == BS_4X8 ? 3 : 2, W_ = 1 << WL, H_ = 1 << HL;
  const FrameDev *f = w->f; const TileB *t = &w->t; const int ms = w->ms, mi = r * ms + c;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int iU = availU ? mi - ms : mi, iL = availL ? mi - 1 : mi;
  const int l_skip = f->m_skip[mi], l_ymode = f->m_ymode[mi], l_txs_y = f->m_txsize[mi];
  const int l_skU = f->m_skip[iU], l_skL = f->m_skip[iL], l_ymU = f->m_ymode[iU], l_ymL = f->m_ymode[iL], l_txU = f->m_txsize[iU], l_txL = f->m_txsize[iL];
  const int l_cdef = f->cdef_idx[(r >> 4) * w->sb_cols + (c >> 4)];
  int l_uvmode = 0, l_js = 0, l_au = 0, l_av = 0;
  if (w->np > 1) { l_uvmode = f->m_uvmode[mi]; l_js = f->m_cfl_sign[mi]; l_au = f->m_cfl_au[mi]; l_av = f->m_cfl_av[mi]; }
  const int skip = U_(l_skip), ymode = U_(l_ymode), txs_y = U_(l_txs_y), v_skU = U_(l_skU), v_skL = U_(l_skL), v_ymU = U_(l_ymU), v_ymL = U_(l_ymL);
  const int v_txU = U_(l_txU), v_txL = U_(l_txL), v_cdef = U_(l_cdef), uvmode = U_(l_uvmode), v_js = U_(l_js), v_au = U_(l_au), v_av = U_(l_av);
  RangeEncDev *e = &w->ec; LDS uint16_t *cdf = w->cdf;
  re_symbol_dev(e, skip, cdf + CDF_SKIP + ((availU ? v_skU : 0) + (availL ? v_skL : 0)) * CDF_SKIP_STRIDE, 2);
  if (!skip && w->enable_cdef && w->cdef_pending) { w->cdef_pending = 0; re_literal_dev(e, (uint32_t)v_cdef, w->cdef_bits); }
  const int am = intra_mode_ctx(availU ? v_ymU : DC_PRED), lm = intra_mode_ctx(availL ? v_ymL : DC_PRED);
  re_symbol_dev(e, ymode, cdf + CDF_KF_Y + (am * 5 + lm) * CDF_KF_Y_STRIDE, 13);
  if (w->np > 1) {
    re_symbol_dev(e, uvmode, cdf + CDF_UV_CFL + ymode * CDF_UV_CFL_STRIDE, 14);
    if (uvmode == UV_CFL_PRED) {
      const int js = v_js, su = (js + 1) / 3, sv = (js + 1) % 3;
      re_symbol_dev(e, js, cdf + CDF_CFL_SIGN, 8);
      if (su) re_symbol_dev(e, v_au, cdf + CDF_CFL_ALPHA + ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE, 16);
      if (sv) re_symbol_dev(e, v_av, cdf + CDF_CFL_ALPHA + ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE, 16);
    }
  }
  if (w->tx_mode_select) {
    const int actx = availU && dim_wl(v_txU) >= WL, lctx = availL && dim_hl(v_txL) >= HL;
    re_symbol_dev(e, txs_y == BSR ? 0 : 1, cdf + CDF_TX_SIZE + (actx + lctx) * CDF_TX_SIZE_STRIDE, 2);
  }
  if (skip) return;
  for (int p = 0; p < w->np; p++) {
    const int txs = p == 0 ? txs_y : BSR, split = txs != BSR;                       // luma may be two 4x4 transforms
    for (int bi = 0; bi < (split ? 2 : 1); bi++) {
      const int rr = r + ((split && H_ == 8) ? bi : 0), cc = c + ((split && W_ == 8) ? bi : 0), tmi = rr * ms + cc;
      if (rr >= w->mi_rows || cc >= w->mi_cols) continue;
      const int l_eob = f->m_eob[p][tmi], l_txt = f->m_txtype[tmi];
      const int tw = split ? 4 : W_, th = split ? 4 : H_, twl = split ? 2 : WL;
      const int32_t *src = f->coef[p] + (size_t)(rr * 4) * f->stride + cc * 4;
      WAVE_SYNC();
      for (int idx = LANE; idx < tw * th; idx += 64) w->qc[idx] = src[(idx >> twl) * f->stride + (idx & (tw - 1))];
      WAVE_SYNC();
      {                                                                            // level map with its 4-wide zero border (build_level_map for tw x th)
        const int st = tw + 4;
        for (int i = LANE; i < st * (th + 4); i += 64) w->lev[i] = 0;
        WAVE_SYNC();
        for (int i = LANE; i < tw * th; i += 64) w->lev[(i >> twl) * st + (i & (tw - 1))] = (uint8_t)imin_(iabs_(w->qc[i]), 127);
        WAVE_SYNC();
      }
      const int eob = U_(l_eob), v_txt = U_(l_txt);
      int txtype, off = -1, sym = 0, ns = 0, set;
      if (p == 0) {
        txtype = v_txt;
        off = split ? intra_tx_cdf(&w->txc, 0, ymode, &ns, &set) : rect_tx_cdf(&w->txc, ymode, &ns, &set);
        if (off >= 0) sym = txtype_to_sym(set, txtype);
      } else {
        set = w->txc.reduced_tx_set ? 2 : 1;
        txtype = mode_to_txtype(uvmode);
        if (txtype_to_sym(set, txtype) < 0) txtype = DCT_DCT;
      }
      int sctx2, dctx;
      txb_ctx_wh(f, t, p, rr, cc, tw >> 2, th >> 2, !split, &sctx2, &dctx);
      code_coeffs_lane0(w, eob, p, split ? 0 : BSR, txtype, sctx2, dctx, off, sym, ns);
    }
  }
  WAVE_SYNC();
}
#endif

// Partition symbol of the node (r, c, bs >= 1); returns 0 (NONE) or 3 (SPLIT).  spec 5.11.4
__device__ __forceinline__ int write_partition_symbol(TileWriter *w, int r, int c, int bs) {
  const FrameDev *f = w->f; const TileB *t = &w->t; const int ms = w->ms;
  const int half = (1 << bs) >> 1;
  const int has_rows = (r + half) < w->mi_rows, has_cols = (c + half) < w->mi_cols;
  const int actual = U_(f->m_bsize[r * ms + c]);
#if MI_RECT_PART
  int part = actual == bs ? 0 : ((bs == BS_8 && actual == BS_8X4) ? 1 : ((bs == BS_8 && actual == BS_4X8) ? 2 : 3));
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int above = availU && dim_wl(U_(f->m_bsize[(r - 1) * ms + c])) < 2 + bs, left = availL && dim_hl(U_(f->m_bsize[r * ms + c - 1])) < 2 + bs;
#else
  int part = actual == bs ? 0 : 3;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int above = availU && f->m_bsize[(r - 1) * ms + c] < bs, left = availL && f->m_bsize[r * ms + c - 1] < bs;
#endif
  LDS uint16_t *cdf = w->cdf + CDF_PARTITION + ((bs - 1) * 4 + left * 2 + above) * CDF_PARTITION_STRIDE;
  const int ns = bs == BS_8 ? 4 : 10;
  if (has_rows && has_cols) re_symbol_dev(&w->ec, part, cdf, ns);
  else if (has_rows || has_cols) {
#define PP_(i) ((uint32_t)((i) > 0 ? cdf[(i) - 1] : 32768) - cdf[i])
    uint32_t psum;
    if (has_cols) psum = PP_(2) + PP_(3) + PP_(4) + PP_(6) + PP_(7) + PP_(9);
    else psum = PP_(1) + PP_(3) + PP_(4) + PP_(5) + PP_(6) + PP_(8);
#undef PP_
    re_bool_dev(&w->ec, 1, (uint32_t)U_(psum));
  }
  if (!(has_rows && has_cols)) part = 3;
  K4PH(0);
  return part;
}

// read_lr() of the superblock at (r, c) (spec 5.11.57 / 5.11.58): with 64x64 units at most one unit per plane
__device__ __forceinline__ void write_lr_sb(TileWriter *w, int r, int c) {
  const FrameDev *f = w->f;
  if (!w->enable_restoration) return;
  const int ucols = lr_units_of(w->fw), urows = lr_units_of(w->fh), n = ucols * urows;
  const int urs = (r * 4 + 63) / 64, ure = imin_(((r + 16) * 4 + 63) / 64, urows);
  const int ucs = (c * 4 + 63) / 64, uce = imin_(((c + 16) * 4 + 63) / 64, ucols);
  for (int p = 0; p < w->np; p++) for (int ur = urs; ur < ure; ur++) for (int uc = ucs; uc < uce; uc++) {
    const int ui = p * n + ur * ucols + uc;
    const int type = U_(f->lr_type[ui]);
    re_symbol_dev(&w->ec, type ? 2 : 0, w->lr_cdf, 3);
    if (!type) continue;
    const int set = U_(f->lr_set[ui]);
    re_literal_dev(&w->ec, (uint32_t)set, 4);
    int r0, s0, r1, s1; sgr_param(set, &r0, &s0, &r1, &s1);
    for (int i = 0; i < 2; i++) {
      const int v = U_(f->lr_xqd[ui * 2 + i]);
      if (i == 0 ? r0 : r1) {
        uint32_t bits; const int nb = lr_subexp_code(v, i == 0 ? -96 : -32, i == 0 ? 32 : 96, w->lr_ref[p * 2 + i], &bits);
        re_literal_dev(&w->ec, bits, nb);
      }
      WAVE_SYNC();
      if (LANE == 0) w->lr_ref[p * 2 + i] = v;
      WAVE_SYNC();
    }
  }
}

// Iterative Z-order walk of one superblock (explicit stack, depth <= 5) so that every block-size instance of
// write_block_dev is inlined exactly once and the range-coder state stays in registers.
template <int MAXBS> __device__ __forceinline__ void write_superblock(TileWriter *w, int r0, int c0) {
  const FrameDev *f = w->f;
  w->cdef_pending = 1;
  K4PH(7);
  write_lr_sb(w, r0, c0);
  K4PH(6);
  int sr[5], sc[5], sk[5];
  int sp = 0; sr[0] = r0; sc[0] = c0; sk[0] = 0;
  while (sp >= 0) {
    const int bs = 4 - sp;
    int r = 0, c = 0, kk = 0;
#pragma unroll
    for (int q = 0; q < 5; q++) if (q == sp) { r = sr[q]; c = sc[q]; kk = sk[q]; }
    if (kk == 0) {
      if (r >= w->mi_rows || c >= w->mi_cols) { sp--; continue; }
      const int part = bs == 0 ? 0 : write_partition_symbol(w, r, c, bs);
      if (part == 0) {
        switch (bs) {
          case 0: write_block_dev<0>(w, r, c); break;
          case 1: write_block_dev<1>(w, r, c); break;
          case 2: write_block_dev<2>(w, r, c); break;
          case 3: if constexpr (MAXBS >= 3) write_block_dev<3>(w, r, c); break;
          default: if constexpr (MAXBS >= 4) write_block_dev<4>(w, r, c); break;
        }
        sp--; continue;
      }
#if MI_RECT_PART
      if (part == 1) { write_block_rect<BS_8X4>(w, r, c); write_block_rect<BS_8X4>(w, r + 1, c); sp--; continue; }
      if (part == 2) { write_block_rect<BS_4X8>(w, r, c); write_block_rect<BS_4X8>(w, r, c + 1); sp--; continue; }
#endif
    }
    if (kk == 4) { sp--; continue; }
    const int half = (1 << bs) >> 1;
#pragma unroll
    for (int q = 0; q < 5; q++) if (q == sp) sk[q] = kk + 1;
    const int nr = r + (kk >> 1) * half, ncol = c + (kk & 1) * half;
#pragma unroll
    for (int q = 0; q < 5; q++) if (q == sp + 1) { sr[q] = nr; sc[q] = ncol; sk[q] = 0; }
    sp++;
  }
}

// <= 128 registers per lane, so that an entropy wave fits the slot one finished search workgroup frees on a SIMD and
// batch A's entropy coding can run next to batch B's search instead of waiting for its tail
// (dynamic LDS: with a compile-time LDS size that caps the occupancy the compiler pads the VGPR allocation to 176)
// LDS sized by the largest block the launch can meet (MAXBS 2: 16x16 -> 15.5 KB, so two entropy workgroups fit the LDS one
// finished search workgroup frees; MAXBS 4: 32x32 coded coefficients -> 28 KB)
template <int CS> struct EntropyLds {
  uint16_t cdf[CDF_TOTAL];
  int32_t qc[CS * CS];
  uint8_t lev[(CS + 4) * (CS + 4) + 4];
  uint16_t scans[SCAN_LDS_ENTRIES(CS)];
  uint16_t rec_off[CS * CS], rec_br[CS * CS];
  uint32_t rec_lv[CS * CS];
  uint16_t lr_cdf[4]; int lr_ref[6];
};

template <int MAXBS>
__global__ __launch_bounds__(MI_K4_THREADS) void tile_entropy_kernel(const FrameDev *__restrict__ frames, const TileJob *__restrict__ jobs, int njobs, uint16_t *precarry, uint32_t pre_cap) {
  constexpr int CS = MAXBS <= 2 ? 16 : 32;
  extern __shared__ __align__(16) uint8_t k4_smem[];            // sizeof(EntropyLds<CS>), passed at launch
  EntropyLds<CS> &L = *(EntropyLds<CS> *)k4_smem;
  const int job = blockIdx.x;
  if (job >= njobs) return;
  const TileJob tj = jobs[job];
  const FrameDev *f = frames + tj.frame;
  if (frame_idle(f)) { if (LANE == 0) f->tile_len[tj.tile_row * f->tile_cols + tj.tile_col] = 0; return; }
  TileWriter w;
  w.f = f;
  w.t.mi_row_start = f->tile_row_start[tj.tile_row] * 16; w.t.mi_row_end = imin_(f->tile_row_start[tj.tile_row + 1] * 16, f->mi_rows);
  w.t.mi_col_start = f->tile_col_start[tj.tile_col] * 16; w.t.mi_col_end = imin_(f->tile_col_start[tj.tile_col + 1] * 16, f->mi_cols);
  w.cdf = (LDS uint16_t *)L.cdf; w.qc = (LDS int32_t *)L.qc; w.lev = (LDS uint8_t *)L.lev; w.cdef_pending = 1; w.ls = (LDS uint16_t *)L.scans;
  w.rec_off = (LDS uint16_t *)L.rec_off; w.rec_br = (LDS uint16_t *)L.rec_br; w.rec_lv = (LDS uint32_t *)L.rec_lv;
  w.lr_cdf = (LDS uint16_t *)L.lr_cdf; w.lr_ref = (LDS int *)L.lr_ref;
  w.np = U_(f->np); w.mi_rows = U_(f->mi_rows); w.mi_cols = U_(f->mi_cols); w.ms = U_(f->mi_stride); w.tx_mode_select = U_(f->tx_mode_select);
  w.enable_cdef = U_(f->enable_cdef); w.cdef_bits = U_(f->cdef_bits); w.enable_restoration = U_(f->enable_restoration); w.sb_cols = U_(f->sb_cols);
  w.fw = U_(f->w); w.fh = U_(f->h); w.txc.reduced_tx_set = U_(f->reduced_tx_set); w.txc.base_q_idx = U_(f->base_q_idx);
  if (LANE < 4) L.lr_cdf[LANE] = (uint16_t)(LANE == 0 ? 32768 - 9413 : (LANE == 1 ? 32768 - 22581 : 0));   // libaom default_switchable_restore_cdf
  if (LANE < 6) L.lr_ref[LANE] = (LANE & 1) ? 31 : -32;                                                      // Sgrproj_Xqd_Mid
  load_scans_to_lds((LDS uint16_t *)L.scans, CS);
  w.sb_cols_tile = (w.t.mi_col_end - w.t.mi_col_start + 15) >> 4;
  for (int i = LANE; i < CDF_TOTAL; i += 64) L.cdf[i] = f->cdf0[i];
  re_init_dev(&w.ec, precarry + (size_t)job * pre_cap, pre_cap);
#if MI_PROFILE == 2
  for (int i = 0; i < 16; i++) w.prof[i] = 0;
  w.pt = clock64();
#endif
  const unsigned long long clk0 = wall_clock64();
  WAVE_SYNC();
  for (int r = w.t.mi_row_start; r < w.t.mi_row_end; r += 16)
    for (int c = w.t.mi_col_start; c < w.t.mi_col_end; c += 16)
      write_superblock<MAXBS>(&w, r, c);
  WAVE_SYNC();
  if (LANE == 0) {
    const int ti = f->tile_base + tj.tile_row * f->tile_cols + tj.tile_col;
    (void)ti;
    uint8_t *out = f->tile_out + (size_t)(tj.tile_row * f->tile_cols + tj.tile_col) * f->tile_out_cap;
    f->tile_len[tj.tile_row * f->tile_cols + tj.tile_col] = re_finish_dev(&w.ec, out, f->tile_out_cap);
#if MI_PROFILE == 2
    { TileWriter *w_ = &w; TileWriter *w = w_; K4PH(7); if (f->prof_out) for (int i = 0; i < 16; i++) f->prof_out[(size_t)job * 128 + 96 + i] = w->prof[i]; }
#endif
    unsigned long long *tc = f->tile_clk + (size_t)(tj.tile_row * f->tile_cols + tj.tile_col) * 4; tc[2] = clk0; tc[3] = wall_clock64();
  }
}


// dev_blk64.h -- the 64x64 level of the tile search's 32x32 class (K1, tile_search.h); mirrors oracle/av1o_search.c try_block for bs == BS_64.
// SpeedTweaks asks for partition_range (4, 64) at speed <= 1 below the high-quality threshold (ravif/src/av1encoder.rs:556-566).
//
// A 64x64 block is the whole superblock.  Its working set does not fit the class's per-wave scratch four times over in the generic form (a 64x64
// residual + two reconstructions + levels per wave), so this level runs on a leaner layout laid over the same LDS (WaveScratch::b64):
//   * luma, undivided: one candidate per wavefront as everywhere else.  The 64-point column pass reads source and prediction directly and keeps only
//     the 32 output rows the row pass uses (the rest of a 64x64 transform is discarded by the format: 32x32 coded coefficients); the block is
//     reconstructed in place over its prediction; a wave's best candidate waits in the workgroup's HBM scratch (Blk64Hbm), not in LDS.
//   * luma transform-size trial (depth 1: four 32x32, depth 2: sixteen 16x16 transform blocks) and chroma (a 64x64 block of a 4:4:4 frame carries
//     four 32x32 transform blocks per plane, spec get_tx_size; each is predicted from the reconstruction of the ones before it, spec transform_block):
//     every transform block is an ordinary evaluation of the 32x32 class (eval_tx).  Only the bottom row and the right column of a finished transform
//     block stay in LDS (the next ones' edges); reconstructions and levels go to the HBM scratch and reach the frame when their candidate has won.
// Chroma candidates are dealt like the generic path's: two wave pairs, one candidate per pair, one plane per wave; no CfL at this size.
#pragma once

// HBM scratch of one persistent workgroup, behind its area snapshots
struct Blk64Hbm {
  uint16_t lrec[4][64 * 64]; int32_t lqc[4][32 * 32];                 // luma: each wave's best undivided candidate
  uint16_t trec[64 * 64]; int32_t tqc[64 * 64];                       // luma transform-size trial: the best chain so far; sub-block q's levels at tqc + q * (coded samples per sub-block)
  uint16_t crec4[4][64 * 64]; int32_t cqc4[4][64 * 64];               // luma transform-size trial: each wave's chain in progress (one transform type for the whole block)
  uint16_t crec[2][2][2][64 * 64]; int32_t cqc[2][2][2][4 * 32 * 32]; // chroma [wave pair][buffer][plane - 1]: the pair's candidate in progress and its best so far
};
typedef unsigned int v2u_t __attribute__((vector_size(8)));    // 8- / 16-byte LDS -> HBM copies (built-in vectors: assignable across address spaces)
typedef unsigned int v4u_t __attribute__((vector_size(16)));
#define MI_BLK64_HBM_BYTES ((sizeof(Blk64Hbm) + 255) & ~(size_t)255)

// forward 64x64 DCT of (src - pred), both 64 x 64 with pitch 64: coef[32][32]; tbuf: int32 [32][65]  (dev_txfm.h fwd_txfm2d_dev<64> without the rows it discards)
__device__ inline void fwd_txfm64_dev(const LDS uint16_t *src, const LDS uint16_t *pred, LDS int32_t *tbuf, LDS int32_t *coef) {
  constexpr int P = 65;
  {
    const int c = LANE;
    int32_t x[64];
#pragma unroll
    for (int r = 0; r < 64; r++) x[r] = (int)src[r * 64 + c] - (int)pred[r * 64 + c];
    av1_fdct64(x);
#pragma unroll
    for (int r = 0; r < 32; r++) tbuf[r * P + c] = rshift_round_(x[r], 2);
  }
  WAVE_SYNC();
  if (LANE < 32) {
    const int r = LANE;
    int32_t x[64];
#pragma unroll
    for (int c = 0; c < 64; c++) x[c] = tbuf[r * P + c];
    av1_fdct64(x);
#pragma unroll
    for (int c = 0; c < 32; c++) coef[r * 32 + c] = rshift_round_(x[c], 2);
  }
  WAVE_SYNC();
}
// inverse: dq[32][32] -> residual added to rec[64 * 64] in place (inv_txfm2d_add_dev<64>; the rows the row pass leaves at zero are not stored)
__device__ inline void inv_txfm64_add_dev(const LDS int32_t *dq, LDS int32_t *tbuf, LDS uint16_t *rec, int bd) {
  constexpr int P = 65;
  const int cbits = imax_(bd + 6, 16), cmax = (1 << (cbits - 1)) - 1, cmin = -(1 << (cbits - 1));
  if (LANE < 32) {
    const int i = LANE;
    int32_t x[64];
#pragma unroll
    for (int j = 0; j < 64; j++) x[j] = j < 32 ? dq[i * 32 + j] : 0;
    av1_idct64(x);
#pragma unroll
    for (int j = 0; j < 64; j++) tbuf[i * P + j] = iclamp_(round2_(x[j], 2), cmin, cmax);
  }
  WAVE_SYNC();
  const int mx = (1 << bd) - 1;
  {
    const int j = LANE;
    int32_t x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) x[i] = i < 32 ? tbuf[i * P + j] : 0;
    av1_idct64(x);
#pragma unroll
    for (int i = 0; i < 64; i++) rec[i * 64 + j] = (uint16_t)iclamp_((int)rec[i * 64 + j] + round2_(x[i], 4), 0, mx);
  }
  WAVE_SYNC();
}

template <int NW, int TS>
__device__ MI_K1_TRY_ATTR long long try_block64(const Ctx<32, NW, TS> k, int r, int c, long long budget = J_INF) {
  static_assert(NW == 4, "the 64x64 level deals its candidates to four wavefronts");
  constexpr int MAXN = 32, BS = 4, n = 64, n4 = 16, log2w = 6, nn = n * n;
  const LDS FrameDev *f = k.f(); const LDS TileB *t = k.t(); LDS WaveScratch<MAXN> *S = k.s(); LDS SharedScratch<MAXN> *SH = k.sh();
  LDS Blk64Wave *B = &S->b64;
  Blk64Hbm *H = (Blk64Hbm *)(k.snap() + MI_SNAP_BYTES_ALL(64));
  const int W = WAVE_ID;
  const int ms = f->mi_stride, mi = r * ms + c, x = c * 4, y = r * 4;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int can_ar = availU && (c + n4 < t->mi_col_end), can_bl = availL && (r + n4 < t->mi_row_end);
  const int iU = availU ? mi - ms : mi, iL = availL ? mi - 1 : mi;
  const int v_ymU = f->m_ymode[iU], v_ymL = f->m_ymode[iL];
  const int v_uvU = f->np > 1 ? f->m_uvmode[iU] : 0, v_uvL = f->np > 1 ? f->m_uvmode[iL] : 0;
  const int v_skU = f->m_skip[iU], v_skL = f->m_skip[iL], v_txU = f->m_txsize[iU], v_txL = f->m_txsize[iL];
  const int v_skUL = f->m_skip[availU && availL ? mi - ms - 1 : mi];
  const int have_ar = can_ar, have_bl = 0;                     // (decoded_before: the superblock above-right is final, the one below-left never is)
  const int amode = availU ? uni32(v_ymU) : DC_PRED, lmode = availL ? uni32(v_ymL) : DC_PRED;
  const int nb_skip = (availU ? uni32(v_skU) & 1 : 0) + (availL ? uni32(v_skL) & 1 : 0);
  const int seg_nb = (availU && availL ? (uni32(v_skUL) >> 1) + 1 : 0) | ((availU ? (uni32(v_skU) >> 1) + 1 : 0) << 4) | ((availL ? (uni32(v_skL) >> 1) + 1 : 0) << 8);
  const int nb_txU = availU ? uni32(v_txU) : -1, nb_txL = availL ? uni32(v_txL) : -1;
  const uint16_t *ycost = k.cost() + CDF_KF_Y + (intra_mode_ctx(amode) * 5 + intra_mode_ctx(lmode)) * CDF_KF_Y_STRIDE;
  const int ftype_y = IS_SMOOTH_(amode) || IS_SMOOTH_(lmode);
  const int ftype_uv = f->np > 1 && ((availU && IS_SMOOTH_(uni32(v_uvU))) || (availL && IS_SMOOTH_(uni32(v_uvL))));
  LDS uint16_t *wa = S->wa + EDGE_OFF, *wl = S->wl + EDGE_OFF;
  PH_BEGIN();

  // ---- stage: luma source (all waves), raw edges + transform contexts of every plane (plane p by wave p + 1), psychovisual references (wave 0) ----
  {
    const uint16_t *g = f->src[0] + (size_t)y * f->stride + x;
    for (int idx = threadIdx.x; idx < nn; idx += 64 * NW) SH->x64.src64[idx] = g[(idx >> 6) * f->stride + (idx & 63)];
  }
  for (int p = 0; p < f->np; p++) if (p + 1 == W) {
    if (p == 0) {
      int sc_, dc_;
      txb_ctx_dev(f, t, 0, r, c, BS, BS, &sc_, &dc_);
      if (LANE == 0) { SH->sctx[0] = sc_; SH->dctx[0] = dc_; }
    }
    load_edges(f, p, x, y, n, availL, availU, have_ar, have_bl, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF);
    if (LANE < n4) {                                           // the (level, dc) contexts the block's outer neighbours left behind, per 4x4 cell
      const int k2 = LANE;
      const bool ha = availU && c + k2 < f->mi_cols, hl = availL && r + k2 < f->mi_rows;
      const int ia = ha ? (r - 1) * ms + c + k2 : mi, il = hl ? (r + k2) * ms + c - 1 : mi;
      const int la = f->m_lvl[p][ia], da = f->m_dc[p][ia], ll = f->m_lvl[p][il], dl2 = f->m_dc[p][il];
      LDS uint8_t (*nt)[2] = p == 0 ? SH->nb_top : SH->x64.cnb_top[p - 1]; LDS uint8_t (*nl)[2] = p == 0 ? SH->nb_left : SH->x64.cnb_left[p - 1];
      nt[k2][0] = (uint8_t)(ha ? la : 0); nt[k2][1] = (uint8_t)(ha ? da : 0);
      nl[k2][0] = (uint8_t)(hl ? ll : 0); nl[k2][1] = (uint8_t)(hl ? dl2 : 0);
    }
  }
  if (W == 0) {
    const int cw = f->pw >> 3;
    const int cell = ((y >> 3) + LANE / 8) * cw + (x >> 3) + LANE % 8;
    const int a = (int)f->act[cell];
    SH->pact[LANE] = a; SH->psv[LANE] = (int)f->svar8[cell];
    const int tot = wave_sum_i32(a);
    const int cact = (tot + 32) / 64;
    if (LANE == 0) { SH->cact = cact; SH->seg_nb = seg_nb; }
    seg_select(f, SH, cact);
  }
  PH(1);
  WG_SYNC();
  PH(2);
  const int sctx_y = SH->sctx[0], dctx_y = SH->dctx[0];
  const LDS uint16_t *ra = SH->ra[0] + EDGE_OFF, *rl = SH->rl[0] + EDGE_OFF;

  // ---- luma: SATD pre-filter over the 13 modes, dealt by cost like the generic path ----
  {
    const int deal = W == 0 ? 0x0F043 : W == 1 ? 0x0F165 : W == 2 ? 0x2C97 : 0x0FBA8;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const int m = (deal >> (4 * i)) & 15;
      if (m < 13) {
        predict_block(f, x, y, log2w, availL, availU, m, 0, ftype_y, ra, rl, wa, wl, S->etmp, B->pred);
        const long long sd = satd_dev(SH->x64.src64, B->pred, n);
        if (LANE == 0) SH->satd[m] = sd;
      }
    }
  }
  PH(3);
  WG_SYNC();
  PH(2);
  if (LANE < 13) {
    const long long mine = SH->satd[LANE];
    int rank = 0;
    for (int j = 0; j < 13; j++) { const long long o = SH->satd[j]; rank += (o < mine) || (o == mine && j < LANE); }
    SH->order[rank] = LANE;
  }
  WAVE_SYNC();
  const int ncand = Tools<TS>::FULL ? 7 : 3;
  auto dl_of = [](int q) { const int a = (q >> 1) + 1; return (q & 1) ? a : -a; };
  const int refine = Tools<TS>::fine_directional(f);
  if (refine) {
#pragma unroll 1
    for (int u = W; u < ncand * 6; u += NW) {
      const int ci = u / 6, q = u - ci * 6, m = SH->order[ci];
      if (m >= V_PRED && m <= D67_PRED) {
        predict_block(f, x, y, log2w, availL, availU, m, dl_of(q), ftype_y, ra, rl, wa, wl, S->etmp, B->pred);
        const long long sd = satd_dev(SH->x64.src64, B->pred, n);
        if (LANE == 0) SH->dsd[ci][q] = sd;
      }
    }
    PH(5);
    WG_SYNC();
    PH(2);
  }
  // ---- full RD over the surviving (mode, delta): candidate e by wave e % NW; a 64x64 transform is always DCT_DCT and its type is not coded ----
  const bool tx_trial = Tools<TS>::tx_mode_select(f) && Tools<TS>::rdo_tx(f);
  long long my_j = J_INF; int my_e = 1 << 30, my_mode = DC_PRED, my_delta = 0; TxRes my_tr = { 0, 0, 0, 0, 0 };
  uint32_t my_mrate = 0;
#pragma unroll 1
  for (int e = W; e < ncand; e += NW) {
    const int m = SH->order[e];
    const int directional = m >= V_PRED && m <= D67_PRED;
    int delta = 0;
    if (directional && refine) {
      long long bsd = SH->satd[m];
      for (int q = 0; q < 6; q++) { const long long sd = SH->dsd[e][q]; if (sd < bsd) { bsd = sd; delta = dl_of(q); } }
    }
    predict_block(f, x, y, log2w, availL, availU, m, delta, ftype_y, ra, rl, wa, wl, S->etmp, B->pred);
    const uint32_t mode_rate = y_mode_rate(k.cost(), ycost, m, directional, delta);
    TxRes tr;
    fwd_txfm64_dev(SH->x64.src64, B->pred, B->tbuf, B->cbuf);
    const int eob = quant_rate_dev<32>(k.cc(), k.cost(), k.ls(), B->cbuf, B->qc, S->lev, 0, BS, DCT_DCT, f->dc_q[0], f->ac_q[0], f->dc_recip[0], f->ac_recip[0],
                                       f->bd, sctx_y, dctx_y, -1, 0, &tr.rate, &tr.cul, &tr.dcc);
    if (eob > 0) inv_txfm64_add_dev(B->cbuf, B->tbuf, B->pred, f->bd);
    tr.eob = eob;
    if (!Tools<TS>::tune_psnr(f)) tr.sse = psy_dist_wave<64>(SH->x64.src64, B->pred, (const LDS int *)SH->psv, (const LDS int *)SH->pact, f->bd);
    else tr.sse = sse_dev(SH->x64.src64, B->pred, nn);
    long long j = ((tr.sse * f->wq[0]) >> 5) + (((long long)tr.rate * f->rdmult + 256) >> 9);
    j += ((long long)mode_rate * f->rdmult + 256) >> 9;
    if (j < my_j) {
      my_j = j; my_e = e; my_mode = m; my_delta = delta; my_tr = tr; my_mrate = mode_rate;
      for (int i = LANE; i < nn / 4; i += 64) ((v2u_t *)H->lrec[W])[i] = ((const LDS v2u_t *)B->pred)[i];
      for (int i = LANE; i < 256; i += 64) ((v4u_t *)H->lqc[W])[i] = ((const LDS v4u_t *)B->qc)[i];
      WAVE_SYNC();                                          // every lane has copied its share before the next candidate's prediction overwrites the block
    }
  }
  if (LANE == 0) { SH->wbest_j[W] = my_j; SH->wbest_e[W] = my_e; }
  PH(6);
  WG_SYNC();
  PH(2);
  int win = 0;
  for (int w2 = 1; w2 < NW; w2++) if (SH->wbest_j[w2] < SH->wbest_j[win] || (SH->wbest_j[w2] == SH->wbest_j[win] && SH->wbest_e[w2] < SH->wbest_e[win])) win = w2;
  const long long best_j = SH->wbest_j[win];
  {
    // the winner's reconstruction and levels reach the frame (every wave copies a share out of the winner's HBM slot)
    uint16_t *gr = f->rec[0] + (size_t)y * f->stride + x;
    int32_t *gc = f->coef[0] + (size_t)y * f->stride + x;
    const uint2 *sr = (const uint2 *)H->lrec[win]; const uint4 *sq = (const uint4 *)H->lqc[win];
    for (int u = threadIdx.x; u < nn / 4; u += 64 * NW) *(uint2 *)(gr + (u >> 4) * f->stride + 4 * (u & 15)) = sr[u];
    for (int u = threadIdx.x; u < 256; u += 64 * NW) *(uint4 *)(gc + (u >> 3) * f->stride + 4 * (u & 7)) = sq[u];
  }
  if (W == win) {
    fill_map_dev(f->m_lvl[0], ms, r, c, n4, my_tr.cul);
    fill_map_dev(f->m_dc[0], ms, r, c, n4, my_tr.dcc);
    if (LANE == 0) f->m_eob[0][mi] = (uint16_t)my_tr.eob;
    fill_map_dev(f->m_ymode, ms, r, c, n4, my_mode);
    fill_map_dev((uint8_t *)f->m_angle_y, ms, r, c, n4, (uint8_t)(int8_t)my_delta);
    fill_map_dev(f->m_txtype, ms, r, c, n4, DCT_DCT);
    fill_map_dev(f->m_bsize, ms, r, c, n4, BS);
    fill_map_dev(f->m_txsize, ms, r, c, n4, BS);
    if (LANE == 0) {
      SH->lm_mode = my_mode; SH->lm_delta = my_delta; SH->lm_tx = DCT_DCT; SH->lm_eob = my_tr.eob;
      SH->lm_mode_j = ((long long)my_mrate * f->rdmult + 256) >> 9;
    }
  }
  WG_SYNC();
  const int best_mode = SH->lm_mode, best_delta = SH->lm_delta;
  long long luma_j = best_j; int any_coef = SH->lm_eob > 0;

  // ---- luma transform size: depth 1 = four 32x32, depth 2 = sixteen 16x16 transform blocks, raster order, each predicted from the ones before it ----
  if (Tools<TS>::tx_mode_select(f)) {
    const int actx = nb_txU >= 0 && (1 << dim_wl(nb_txU)) >= n, lctx = nb_txL >= 0 && (1 << dim_hl(nb_txL)) >= n;
    const uint16_t *dcost = k.cost() + CDF_TX_SIZE + ((BS - 1) * 3 + actx + lctx) * CDF_TX_SIZE_STRIDE;
    luma_j += ((long long)dcost[0] * f->rdmult + 256) >> 9;
    if (tx_trial) {
      LDS int *const sub_tx = (LDS int *)SH->dsd, *const sub_eob = sub_tx + 16, *const sub_cul = sub_tx + 32, *const sub_dcc = sub_tx + 48;
      auto trial = [&](auto depth_c) -> bool {
        constexpr int D = decltype(depth_c)::value;
        constexpr int SBS = BS - D, G = 1 << D, hn = n >> D, half = n4 >> D, hnn = hn * hn, sqn = hnn, scp = hn / 8, pcp = 8;
        static_assert(2 * G * G * hn <= 512, "a chain's boundaries fit a quarter of x64.bnd");
        LDS uint16_t *brow = SH->x64.bnd + W * 512, *rcol = brow + G * G * hn;    // this wave's chain: bottom row / right column of sub-block q at + q * hn
        long long j_split = SH->lm_mode_j + (((long long)dcost[D] * f->rdmult + 256) >> 9);
        int stx_ns = 0, stx_set = 0;
        const int stx_off = intra_tx_cdf_r(f, Tools<TS>::reduced_tx_set(f), SBS, best_mode, &stx_ns, &stx_set);
        const int sntx = stx_off >= 0 ? stx_ns : 1;
        int sub_any = 0;
        // One chain per wavefront and round (transform type rd * NW + W for the whole block: rav1e rdo_tx_type_decision), every wave on its own: sub-source, raw edges from its
        // chain's boundaries, prediction and evaluation in its private scratch (sub-source = rec[1], prediction = dcp), reconstruction and levels in its HBM canvas; a
        // barrier only between rounds: the best complete chain so far moves to trec / tqc, the next round's chains must beat it (a later type never wins a tie).
        LDS uint16_t *A = S->pred + EDGE_OFF, *Lf = S->pred + (MAXN * MAXN / 2) + EDGE_OFF, *psrc = S->rec[1];
        LDS int *ppsv = (LDS int *)S->etmp, *ppact = ppsv + 16;                   // (etmp is predict_block's temporary: free between two predictions)
        LDS uint32_t *cmeta = (LDS uint32_t *)SH->cj + W * 16;
        uint16_t *crec = H->crec4[W]; int32_t *cqc = H->cqc4[W];
        const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1, rs = f->stride, bd = f->bd;
        const uint16_t *grec = f->rec[0];
        const long long j0 = j_split;
        long long best_chain = J_INF;
#pragma unroll 1
        for (int rd = 0; rd * NW < sntx; rd++) {
          const int e = rd * NW + W;
          const bool has_chain = e < sntx;
          long long thr = luma_j < budget ? luma_j : budget;
          if (best_chain < thr) thr = best_chain;
          int txtype;
          if (sntx > 1) txtype = sym_to_txtype(stx_set, has_chain ? e : 0);
          else { txtype = mode_to_txtype(best_mode); if (stx_off < 0 || txtype_to_sym(stx_set, txtype) < 0) txtype = DCT_DCT; }
          const int tx_sym = stx_off >= 0 ? txtype_to_sym(stx_set, txtype) : 0;
          long long jc = has_chain ? j0 : J_INF;
          int any = 0;
          if (has_chain) {
#pragma unroll 1
            for (int q = 0; q < G * G; q++) {
              if (!(jc < thr)) break;                              // wave-uniform
              const int bi = q / G, bj = q % G;
              const int sx = x + bj * hn, sy = y + bi * hn;
              const int sU = availU || bi, sL = availL || bj;
              for (int idx = LANE; idx < hnn; idx += 64) psrc[idx] = SH->x64.src64[(bi * hn + idx / hn) * n + bj * hn + idx % hn];
              {
                // above-right / below-left availability (spec BlockDecoded): what lies right of or below a 64x64 block -- the next superblocks -- is never decoded yet
                const int s_ar = bi == 0 ? (bj < G - 1 ? availU : have_ar) : (bj < G - 1 ? 1 : 0);
                const int s_bl = bj == 0 ? (bi < G - 1 ? availL : have_bl) : 0;
                const int lim_a = imin_(max_x, sx + (s_ar ? 2 * hn : hn) - 1), lim_l = imin_(max_y, sy + (s_bl ? 2 * hn : hn) - 1);
                auto px = [&](int ax, int ay) -> int {                  // absolute sample position -> value
                  const int xr = ax - x, yr = ay - y;
                  if (xr >= 0 && xr < n && yr >= 0 && yr < n) {
                    const int qq = (yr / hn) * G + xr / hn;
                    return (yr % hn) == hn - 1 ? (int)brow[qq * hn + xr % hn] : (int)rcol[qq * hn + yr % hn];   // only rows / columns next to a later sub-block are asked for
                  }
                  if (yr == -1 && xr >= -1 && xr < 2 * n) return (int)ra[xr];
                  if (xr == -1 && yr >= 0 && yr < 2 * n) return (int)rl[yr];
                  return (int)grec[(size_t)ay * rs + ax];
                };
                for (int i = LANE; i <= 2 * hn; i += 64) {
                  const bool corner = i == 2 * hn;
                  int a, l;
                  if (sU) a = px(corner ? (sL ? sx - 1 : sx) : imin_(lim_a, sx + i), sy - 1); else a = px(sL ? sx - 1 : sx, sy);
                  if (sL) l = px(sx - 1, imin_(lim_l, sy + i)); else l = px(sx, sU ? sy - 1 : sy);
                  if (!sU && !sL) { a = corner ? (1 << (bd - 1)) : (1 << (bd - 1)) - 1; l = (1 << (bd - 1)) + 1; }
                  if (corner) { A[-1] = (uint16_t)a; Lf[-1] = (uint16_t)a; } else { A[i] = (uint16_t)a; Lf[i] = (uint16_t)l; }
                }
                WAVE_SYNC();
                predict_block(f, sx, sy, log2w - D, sL, sU, best_mode, best_delta, ftype_y, A, Lf, wa, wl, S->etmp, S->dcp);
              }
              if (LANE < scp * scp) { const int pc = (bi * scp + LANE / scp) * pcp + bj * scp + LANE % scp; ppsv[LANE] = SH->psv[pc]; ppact[LANE] = SH->pact[pc]; }
              // contexts of the sub-block: neighbours outside the block from the staged maps, inside from this chain's sub-blocks
              int ssc, sdc;
              {
                // (the oracle's av1o_txb_ctx leaves out neighbour cells beyond the frame: the staged outer contexts are zero there, an inner neighbour counts its cells inside)
                int top = 0, left = 0, dcs = 0;
                const int nti = iclamp_(f->mi_cols - (c + bj * half), 0, half), nli = iclamp_(f->mi_rows - (r + bi * half), 0, half);
                const uint32_t mt = bi ? cmeta[q - G] : 0u, ml = bj ? cmeta[q - 1] : 0u;
                if (bi == 0) { for (int k2 = 0; k2 < half; k2++) { const int l = SH->nb_top[bj * half + k2][0], d = SH->nb_top[bj * half + k2][1]; top = imax_(top, l); dcs += d == 1 ? -1 : (d == 2 ? 1 : 0); } }
                else if (nti > 0) { const int d = (int)(mt >> 24); top = (int)((mt >> 16) & 0xFF); dcs += nti * (d == 1 ? -1 : (d == 2 ? 1 : 0)); }
                if (bj == 0) { for (int k2 = 0; k2 < half; k2++) { const int l = SH->nb_left[bi * half + k2][0], d = SH->nb_left[bi * half + k2][1]; left = imax_(left, l); dcs += d == 1 ? -1 : (d == 2 ? 1 : 0); } }
                else if (nli > 0) { const int d = (int)(ml >> 24); left = (int)((ml >> 16) & 0xFF); dcs += nli * (d == 1 ? -1 : (d == 2 ? 1 : 0)); }
                sdc = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
                if (top == 0 && left == 0) ssc = 1;
                else if (top == 0 || left == 0) ssc = 2 + (imax_(top, left) > 3);
                else if (imax_(top, left) <= 3) ssc = 4;
                else if (imin_(top, left) <= 3) ssc = 5;
                else ssc = 6;
              }
              WAVE_SYNC();
              TxRes tr;
              jc += eval_tx<MAXN, SBS, NW>(k, 0, ssc, sdc, S->dcp, txtype, stx_off, tx_sym, S->rec[0], S->qc[0], &tr, psrc, (const LDS int *)ppsv, (const LDS int *)ppact);
              {
                // the chain keeps the sub-block: boundaries in LDS (the next sub-blocks' edges), reconstruction and levels in its HBM canvas
                const LDS uint16_t *srec = S->rec[0]; const LDS int32_t *sqc = S->qc[0];
                for (int i = LANE; i < hn; i += 64) { brow[q * hn + i] = srec[(hn - 1) * hn + i]; rcol[q * hn + i] = srec[i * hn + hn - 1]; }
                const int ro = bi * hn * n + bj * hn;
                for (int i = LANE; i < hnn; i += 64) crec[ro + (i / hn) * n + (i % hn)] = srec[i];
                for (int i = LANE; i < sqn; i += 64) cqc[q * sqn + i] = sqc[i];
                if (LANE == 0) cmeta[q] = (uint32_t)tr.eob | ((uint32_t)tr.cul << 16) | ((uint32_t)tr.dcc << 24);
              }
              any |= tr.eob > 0;
              WAVE_SYNC();
            }
          }
          if (LANE == 0) { SH->wbest_j[W] = jc < thr ? jc : J_INF; SH->wbest_e[W] = any; }
          WG_SYNC();
          int sw = 0;
          for (int w2 = 1; w2 < NW; w2++) if (SH->wbest_j[w2] < SH->wbest_j[sw]) sw = w2;            // the lowest wave = the lowest symbol among equals
          const long long rj = SH->wbest_j[sw];
          if (rj < best_chain) {                                 // (thr made it strictly smaller than every earlier round's)
            best_chain = rj; sub_any = SH->wbest_e[sw];
            if (W == sw) {
              for (int i = LANE; i < nn / 4; i += 64) ((v2u_t *)H->trec)[i] = ((const v2u_t *)crec)[i];
              for (int i = LANE; i < G * G * sqn / 4; i += 64) ((v4u_t *)H->tqc)[i] = ((const v4u_t *)cqc)[i];
              if (LANE < G * G) { const uint32_t m = cmeta[LANE]; const int eob = (int)(m & 0xFFFF); sub_eob[LANE] = eob; sub_cul[LANE] = (int)((m >> 16) & 0xFF); sub_dcc[LANE] = (int)(m >> 24); sub_tx[LANE] = eob ? txtype : DCT_DCT; }
            }
          }
          WG_SYNC();
        }
        j_split = best_chain;
        if (j_split < luma_j) {
          // this depth wins: its reconstruction, levels and contexts replace the best so far in the frame
          luma_j = j_split; any_coef = sub_any;
          const int tid = threadIdx.x, T = 64 * NW;
          uint16_t *gr_ = f->rec[0] + (size_t)y * f->stride + x;
          int32_t *gc_ = f->coef[0] + (size_t)y * f->stride + x;
          for (int i = tid; i < nn; i += T) gr_[(i >> 6) * f->stride + (i & 63)] = H->trec[i];
          for (int i = tid; i < G * G * sqn; i += T) { const int q = i / sqn, j2 = i - q * sqn; gc_[((q / G) * hn + j2 / hn) * f->stride + (q % G) * hn + j2 % hn] = H->tqc[i]; }
          if (W == 0) {
            fill_map_dev(f->m_txsize, ms, r, c, n4, SBS);
#pragma unroll 1
            for (int q = 0; q < G * G; q++) {
              const int rr = r + (q / G) * half, cc = c + (q % G) * half;
              fill_map_dev(f->m_lvl[0], ms, rr, cc, half, sub_cul[q]);
              fill_map_dev(f->m_dc[0], ms, rr, cc, half, sub_dcc[q]);
              fill_map_dev(f->m_txtype, ms, rr, cc, half, sub_tx[q]);
              if (LANE == 0) f->m_eob[0][rr * ms + cc] = (uint16_t)sub_eob[q];
            }
          }
        }
        WG_SYNC();
        return false;
      };
      if (trial(std::integral_constant<int, 1>{})) return luma_j;
      if constexpr (MI_TX_DEPTH_MAX >= 2) if (trial(std::integral_constant<int, 2>{})) return luma_j;
    }
  }
  PH(13);
  if (luma_j >= budget) return luma_j;
  long long total_j = luma_j;

  // ---- chroma: candidate ci2 by wave pair, plane (W & 1) + 1 within the pair; four 32x32 transform blocks per plane, one after the other ----
  if (f->np > 1) {
    const uint16_t *uvcost = k.cost() + CDF_UV_NOCFL + best_mode * CDF_UV_NOCFL_STRIDE;       // no CfL above 32x32
    unsigned long long cand_pack = 0; int nc = 0;
    auto push = [&](int m) { cand_pack |= (unsigned long long)m << (4 * nc); nc++; };
    push(DC_PRED);
    if (best_mode != DC_PRED) push(best_mode);
    if (Tools<TS>::FULL) for (int m = 1; m < 13; m++) if (m != best_mode) push(m);
    const int pair = ((W >> 1) & 1) ^ 1, p = (W & 1) + 1;
    long long pb_j = J_INF; int pb_c = 1 << 30, pb_delta = 0, pb_buf = 0, pb_any = 0, pb_eob = 0, pb_cul = 0, pb_dcc = 0;   // per transform block: eob 16 bits (two words), cul 8 bits, dcc 2 bits
    int pb_eob_hi = 0, cbuf_i = 0;
    const int nmine = pair == 0 ? (nc + 1) / 2 : nc / 2, nother = nc - nmine;        // pair 0: candidates 0, 2, 4, ...; pair 1: the odd ones (the winner rule only looks at (cost, index))
    const int rounds = imax_(nmine, nother);
    LDS uint16_t *brow = SH->x64.bnd + W * 256, *rcol = brow + 128;                      // this wave's sub-block boundaries: 4 x 32 samples each way
    const LDS uint8_t (*nt)[2] = SH->x64.cnb_top[p - 1]; const LDS uint8_t (*nl)[2] = SH->x64.cnb_left[p - 1];
    const LDS uint16_t *pra = SH->ra[p] + EDGE_OFF, *prl = SH->rl[p] + EDGE_OFF;
#pragma unroll 1
    for (int rd = 0; rd < rounds; rd++) {
      const int valid = rd < nmine;
      const int ci2 = valid ? 2 * rd + pair : 0;
      const int um = lut4(cand_pack, ci2);
      const int directional = um >= V_PRED && um <= D67_PRED;
      const int delta = (um == best_mode && directional) ? best_delta : 0;
      int jsign = 0;
      const uint32_t mode_rate = uv_mode_rate(k.cost(), uvcost, um, directional, delta, false, 0, 0, &jsign);
      long long jp = 0; int c_eob = 0, c_eob_hi = 0, c_cul = 0, c_dcc = 0;
#pragma unroll 1
      for (int q = 0; q < 4; q++) {
        const int bi = q >> 1, bj = q & 1;
        const int sx = x + bj * 32, sy = y + bi * 32;
        // all waves: the sub-sources of both chroma planes, and the sub-block's mean activity (chroma distortion = SSE x that)
        for (int pp = 1; pp < 3; pp++) {
          const uint16_t *g = f->src[pp] + (size_t)sy * f->stride + sx;
          for (int idx = threadIdx.x; idx < 1024; idx += 64 * NW) SH->srcb[pp][idx] = g[(idx >> 5) * f->stride + (idx & 31)];
        }
        if (W == 0) {
          int a = 0;
          if (LANE < 16) a = SH->pact[(bi * 4 + LANE / 4) * 8 + bj * 4 + LANE % 4];
          const int tot = wave_sum_i32(a);
          if (LANE == 0) SH->cact = (tot + 8) / 16;
        }
        TxRes tr = { 0, 0, 0, 0, 0 };
        int ssc = 0, sdc = 0;
        if (valid) {
          const int sU = availU || bi, sL = availL || bj;
          const int s_ar = bi == 0 ? (bj == 0 ? availU : have_ar) : (bj == 0 ? 1 : 0);
          const int s_bl = bj == 0 ? (bi == 0 ? availL : have_bl) : 0;
          LDS uint16_t *A = S->dcp + EDGE_OFF, *Lf = S->dcp + 512 + EDGE_OFF;
          const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1, rs = f->stride, bd = f->bd;
          const uint16_t *grec = f->rec[p];
          const int lim_a = imin_(max_x, sx + (s_ar ? 64 : 32) - 1), lim_l = imin_(max_y, sy + (s_bl ? 64 : 32) - 1);
          auto px = [&](int ax, int ay) -> int {
            const int xr = ax - x, yr = ay - y;
            if (xr >= 0 && xr < n && yr >= 0 && yr < n) {
              const int qq = (yr >> 5) * 2 + (xr >> 5);
              return (yr & 31) == 31 ? (int)brow[qq * 32 + (xr & 31)] : (int)rcol[qq * 32 + (yr & 31)];
            }
            if (yr == -1 && xr >= -1 && xr < 2 * n) return (int)pra[xr];
            if (xr == -1 && yr >= 0 && yr < 2 * n) return (int)prl[yr];
            return (int)grec[(size_t)ay * rs + ax];
          };
          for (int i = LANE; i <= 64; i += 64) {
            const bool corner = i == 64;
            int a, l;
            if (sU) a = px(corner ? (sL ? sx - 1 : sx) : imin_(lim_a, sx + i), sy - 1); else a = px(sL ? sx - 1 : sx, sy);
            if (sL) l = px(sx - 1, imin_(lim_l, sy + i)); else l = px(sx, sU ? sy - 1 : sy);
            if (!sU && !sL) { a = corner ? (1 << (bd - 1)) : (1 << (bd - 1)) - 1; l = (1 << (bd - 1)) + 1; }
            if (corner) { A[-1] = (uint16_t)a; Lf[-1] = (uint16_t)a; } else { A[i] = (uint16_t)a; Lf[i] = (uint16_t)l; }
          }
          WAVE_SYNC();
          predict_block(f, sx, sy, 5, sL, sU, um, delta, ftype_uv, A, Lf, wa, wl, S->etmp, S->pred);
          // all_zero / dc_sign contexts: 8 cells above and left -- outside the block from the staged maps, inside from this candidate's earlier transform blocks
          {
            int any_a = 0, any_l = 0, dcs = 0;
            const int nti = iclamp_(f->mi_cols - (c + bj * 8), 0, 8), nli = iclamp_(f->mi_rows - (r + bi * 8), 0, 8);
            if (bi == 0) { for (int k2 = 0; k2 < 8; k2++) { const int l = nt[bj * 8 + k2][0], d = nt[bj * 8 + k2][1]; any_a |= l | d; dcs += d == 1 ? -1 : (d == 2 ? 1 : 0); } }
            else if (nti > 0) { const int l = (c_cul >> (8 * (q - 2))) & 255, d = (c_dcc >> (2 * (q - 2))) & 3; any_a = l | d; dcs += nti * (d == 1 ? -1 : (d == 2 ? 1 : 0)); }
            if (bj == 0) { for (int k2 = 0; k2 < 8; k2++) { const int l = nl[bi * 8 + k2][0], d = nl[bi * 8 + k2][1]; any_l |= l | d; dcs += d == 1 ? -1 : (d == 2 ? 1 : 0); } }
            else if (nli > 0) { const int l = (c_cul >> (8 * (q - 1))) & 255, d = (c_dcc >> (2 * (q - 1))) & 3; any_l = l | d; dcs += nli * (d == 1 ? -1 : (d == 2 ? 1 : 0)); }
            sdc = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
            ssc = 7 + (any_a != 0) + (any_l != 0) + 3;
          }
        }
        WG_SYNC();                                               // sub-sources and activity staged
        if (valid) {
          jp += eval_tx<MAXN, BS_32, NW>(k, p, ssc, sdc, S->pred, DCT_DCT, -1, 0, S->rec[0], S->qc[0], &tr, SH->srcb[p]);
          for (int i = LANE; i < 32; i += 64) { brow[q * 32 + i] = S->rec[0][31 * 32 + i]; rcol[q * 32 + i] = S->rec[0][i * 32 + 31]; }
          uint16_t *hr = H->crec[pair][cbuf_i][p - 1] + bi * 32 * 64 + bj * 32; int32_t *hq = H->cqc[pair][cbuf_i][p - 1] + q * 1024;
          for (int i = LANE; i < 256; i += 64) *(v2u_t *)(hr + (i >> 3) * 64 + 4 * (i & 7)) = ((const LDS v2u_t *)S->rec[0])[i];
          for (int i = LANE; i < 256; i += 64) ((v4u_t *)hq)[i] = ((const LDS v4u_t *)S->qc[0])[i];
          if (q < 2) c_eob |= tr.eob << (16 * q); else c_eob_hi |= tr.eob << (16 * (q - 2));
          c_cul |= tr.cul << (8 * q); c_dcc |= tr.dcc << (2 * q);
          WAVE_SYNC();
        }
        WG_SYNC();                                               // everyone is done with the sub-sources before the next ones are staged
      }
      if (valid && LANE == 0) SH->cj[ci2][p - 1] = jp;
      WG_SYNC();
      if (valid) {
        const long long j = SH->cj[ci2][0] + SH->cj[ci2][1] + (((long long)mode_rate * f->rdmult + 256) >> 9);
        if (j < pb_j) {
          pb_j = j; pb_c = ci2; pb_delta = delta; pb_buf = cbuf_i; pb_eob = c_eob; pb_eob_hi = c_eob_hi; pb_cul = c_cul; pb_dcc = c_dcc; pb_any = (c_eob | c_eob_hi) != 0;
          cbuf_i ^= 1;
        }
      }
    }
    if (LANE == 0 && (W & 1) == 0) { SH->pbest_j[pair] = pb_j; SH->pbest_c[pair] = pb_c; }
    WG_SYNC();
    int wp = 0;
    if ((SH->pbest_j[1] < SH->pbest_j[0] || (SH->pbest_j[1] == SH->pbest_j[0] && SH->pbest_c[1] < SH->pbest_c[0]))) wp = 1;
    const long long best_uv = SH->pbest_j[wp];
    if (pair == wp) {
      // the winning pair's best candidate reaches the frame: plane p by this wave
      uint16_t *gr = f->rec[p] + (size_t)y * f->stride + x;
      int32_t *gc = f->coef[p] + (size_t)y * f->stride + x;
      const uint2 *sr = (const uint2 *)H->crec[pair][pb_buf][p - 1]; const uint4 *sq = (const uint4 *)H->cqc[pair][pb_buf][p - 1];
      for (int u = LANE; u < nn / 4; u += 64) *(uint2 *)(gr + (u >> 4) * f->stride + 4 * (u & 15)) = sr[u];
      for (int u = LANE; u < 1024; u += 64) { const int q = u >> 8, v = u & 255; *(uint4 *)(gc + ((q >> 1) * 32 + (v >> 3)) * f->stride + (q & 1) * 32 + 4 * (v & 7)) = sq[u]; }
#pragma unroll 1
      for (int q = 0; q < 4; q++) {
        const int rr = r + (q >> 1) * 8, cc = c + (q & 1) * 8;
        fill_map_dev(f->m_lvl[p], ms, rr, cc, 8, (pb_cul >> (8 * q)) & 255);
        fill_map_dev(f->m_dc[p], ms, rr, cc, 8, (pb_dcc >> (2 * q)) & 3);
        if (LANE == 0) f->m_eob[p][rr * ms + cc] = (uint16_t)((q < 2 ? pb_eob >> (16 * q) : pb_eob_hi >> (16 * (q - 2))) & 0xffff);
      }
      if (LANE == 0) SH->ceob[p - 1] = pb_any;
      if (p == 1) {
        fill_map_dev(f->m_uvmode, ms, r, c, n4, lut4(cand_pack, pb_c));
        fill_map_dev((uint8_t *)f->m_angle_uv, ms, r, c, n4, (uint8_t)(int8_t)pb_delta);
        fill_map_dev(f->m_cfl_sign, ms, r, c, n4, 0);
        fill_map_dev(f->m_cfl_au, ms, r, c, n4, 0);
        fill_map_dev(f->m_cfl_av, ms, r, c, n4, 0);
      }
    }
    WG_SYNC();
    any_coef |= (SH->ceob[0] > 0) | (SH->ceob[1] > 0);
    total_j += best_uv;
  }
  PH(9);
  // ---- skip flag, segment id ----
  const int skip = !any_coef;
  int seg_ctx = 0;
  const int seg_nb2 = SH->seg_nb, seg_ul = (seg_nb2 & 15) - 1, seg_u = ((seg_nb2 >> 4) & 15) - 1, seg_l = (seg_nb2 >> 8) - 1;
  const int seg_p = seg_pred(seg_ul, seg_u, seg_l, &seg_ctx), seg_own = f->seg_n ? SH->seg : 0, seg_fin = f->seg_n ? (skip ? seg_p : seg_own) : 0;
  if (W == 0) {
    fill_map_dev(f->m_skip, ms, r, c, n4, skip | (seg_fin << 1));
    if (skip) for (int pp = 0; pp < f->np; pp++) { fill_map_dev(f->m_lvl[pp], ms, r, c, n4, 0); fill_map_dev(f->m_dc[pp], ms, r, c, n4, 0); }
  }
  total_j += ((long long)k.cost()[CDF_SKIP + nb_skip * CDF_SKIP_STRIDE + skip] * f->rdmult + 256) >> 9;
  if (f->seg_n && !skip) total_j += ((long long)k.cost()[CDF_SEG_ID + seg_ctx * CDF_SEG_ID_STRIDE + seg_symbol(seg_own, seg_p, f->seg_n)] * f->rdmult + 256) >> 9;
  WG_SYNC();
  return total_j;
}

/* oracle/av1o_headers.c -- OBU syntax (spec 5.3-5.11: temporal delimiter, sequence header with
 * reduced_still_picture_header, frame OBU) and the top-level av1o_encode().
 * TEST INFRASTRUCTURE (see av1o.h).  Mirrors what rav1e's header.rs emits for EncoderConfig{still_picture:true}
 * (ravif/src/av1encoder.rs:670-701) as far as the syntax is concerned. */
#include "av1o_int.h"
#include <stdio.h>

static int tile_log2(int blk, int target) { int k = 0; while ((blk << k) < target) k++; return k; }

static void write_sequence_header(const Av1oFrame *f, BitW *b) {
  const Av1oConfig *c = &f->cfg;
  bw_put(b, c->mono ? 0 : 1, 3);       /* seq_profile: 0 = 4:0:0/4:2:0, 1 = 4:4:4 */
  bw_put(b, 1, 1);                     /* still_picture */
  bw_put(b, 1, 1);                     /* reduced_still_picture_header */
  bw_put(b, 31, 5);                    /* seq_level_idx[0]: "maximum parameters" */
  int wb = 32 - __builtin_clz((unsigned)imax(c->width - 1, 1)), hb = 32 - __builtin_clz((unsigned)imax(c->height - 1, 1));
  bw_put(b, (uint32_t)(wb - 1), 4); bw_put(b, (uint32_t)(hb - 1), 4);
  bw_put(b, (uint32_t)(c->width - 1), wb); bw_put(b, (uint32_t)(c->height - 1), hb);
  bw_put(b, 0, 1);                     /* use_128x128_superblock */
  bw_put(b, 0, 1);                     /* enable_filter_intra */
  bw_put(b, 1, 1);                     /* enable_intra_edge_filter */
  bw_put(b, 0, 1);                     /* enable_superres */
  bw_put(b, (uint32_t)f->enable_cdef, 1);
  bw_put(b, (uint32_t)f->enable_restoration, 1);
  /* color_config() */
  bw_put(b, c->bit_depth > 8, 1);      /* high_bitdepth */
  if (c->mono) bw_put(b, 1, 1);        /* mono_chrome (profile 0 only) */
  bw_put(b, (uint32_t)c->has_color_desc, 1);
  int cp = 2, tc = 2, mc = 2;
  if (c->has_color_desc) { cp = c->color_primaries; tc = c->transfer; mc = c->matrix; bw_put(b, (uint32_t)cp, 8); bw_put(b, (uint32_t)tc, 8); bw_put(b, (uint32_t)mc, 8); }
  if (c->mono) {
    bw_put(b, (uint32_t)c->full_range, 1);
  } else if (cp == 1 && tc == 13 && mc == 0) {
    /* sRGB/identity: color_range = 1, 4:4:4 implied, no bits */
  } else {
    bw_put(b, (uint32_t)c->full_range, 1);
    /* seq_profile 1: subsampling 0,0 implied */
  }
  if (!c->mono) bw_put(b, 1, 1);       /* separate_uv_delta_q */
  bw_put(b, 0, 1);                     /* film_grain_params_present */
  bw_trailing(b);
}

static void write_delta_q(BitW *b, int d) { if (d) { bw_put(b, 1, 1); bw_su(b, d, 7); } else bw_put(b, 0, 1); }

static void write_frame_header(const Av1oFrame *f, BitW *b, int tile_size_bytes) {
  /* uncompressed_header() with reduced_still_picture_header = 1: key frame, shown */
  bw_put(b, 0, 1);                     /* disable_cdf_update */
  bw_put(b, 0, 1);                     /* allow_screen_content_tools */
  bw_put(b, 0, 1);                     /* render_and_frame_size_different */
  /* tile_info() */
  const int sbc = f->sb_cols, sbr = f->sb_rows;
  const int min_cols_log2 = tile_log2(64, sbc), max_cols_log2 = tile_log2(1, imin(sbc, MAX_TILE_COLS));
  const int max_rows_log2 = tile_log2(1, imin(sbr, MAX_TILE_ROWS));
  const int min_log2_tiles = imax(min_cols_log2, tile_log2(2304, sbr * sbc));
  bw_put(b, 1, 1);                     /* uniform_tile_spacing_flag */
  for (int k = min_cols_log2; k < max_cols_log2; k++) { if (k < f->tile_cols_log2) bw_put(b, 1, 1); else { bw_put(b, 0, 1); break; } }
  const int min_rows_log2 = imax(min_log2_tiles - f->tile_cols_log2, 0);
  for (int k = min_rows_log2; k < max_rows_log2; k++) { if (k < f->tile_rows_log2) bw_put(b, 1, 1); else { bw_put(b, 0, 1); break; } }
  if (f->tile_cols_log2 > 0 || f->tile_rows_log2 > 0) {
    bw_put(b, 0, f->tile_cols_log2 + f->tile_rows_log2);   /* context_update_tile_id */
    bw_put(b, (uint32_t)(tile_size_bytes - 1), 2);
  }
  /* quantization_params() */
  bw_put(b, (uint32_t)f->base_q_idx, 8);
  write_delta_q(b, f->dc_qi[0] - f->base_q_idx);
  if (f->np > 1) {
    const int diff = (f->dc_qi[1] != f->dc_qi[2]) || (f->ac_qi[1] != f->ac_qi[2]);
    bw_put(b, (uint32_t)diff, 1);      /* diff_uv_delta (separate_uv_delta_q = 1) */
    write_delta_q(b, f->dc_qi[1] - f->base_q_idx); write_delta_q(b, f->ac_qi[1] - f->base_q_idx);
    if (diff) { write_delta_q(b, f->dc_qi[2] - f->base_q_idx); write_delta_q(b, f->ac_qi[2] - f->base_q_idx); }
  }
  bw_put(b, 0, 1);                     /* using_qmatrix */
  /* segmentation_params(): primary_ref_frame = NONE -> update_map = 1, temporal_update = 0, update_data = 1 without bits; the one feature is ALT_Q */
  bw_put(b, f->seg_n > 0, 1);
  if (f->seg_n > 0)
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) {
      const int on = j == 0 && i < f->seg_n;
      bw_put(b, (uint32_t)on, 1);
      if (on) bw_su(b, f->seg_qidx[i] - f->base_q_idx, 9);      /* feature_value su(1 + 8) */
    }
  if (f->base_q_idx > 0) bw_put(b, 0, 1); /* delta_q_present */
  /* loop_filter_params(): CodedLossless == 0 */
  bw_put(b, (uint32_t)f->lf_level[0], 6); bw_put(b, (uint32_t)f->lf_level[1], 6);
  if (f->np > 1 && (f->lf_level[0] || f->lf_level[1])) { bw_put(b, (uint32_t)f->lf_level[2], 6); bw_put(b, (uint32_t)f->lf_level[3], 6); }
  bw_put(b, (uint32_t)f->lf_sharp, 3);
  bw_put(b, 0, 1);                     /* loop_filter_delta_enabled */
  /* cdef_params() */
  if (f->enable_cdef) {
    bw_put(b, (uint32_t)(f->cdef_damping - 3), 2);
    bw_put(b, (uint32_t)f->cdef_bits, 2);
    for (int i = 0; i < (1 << f->cdef_bits); i++) {
      bw_put(b, (uint32_t)(f->cdef_y[i] >> 2), 4); bw_put(b, (uint32_t)(f->cdef_y[i] & 3), 2);
      if (f->np > 1) { bw_put(b, (uint32_t)(f->cdef_uv[i] >> 2), 4); bw_put(b, (uint32_t)(f->cdef_uv[i] & 3), 2); }
    }
  }
  /* lr_params() */
  if (f->enable_restoration) {
    for (int p = 0; p < f->np; p++) bw_put(b, 1, 2);   /* lr_type 1 -> RESTORE_SWITCHABLE (Remap_Lr_Type) */
    bw_put(b, 0, 1);                                   /* lr_unit_shift = 0: 64x64 units; 4:4:4 / 4:0:0 -> no lr_uv_shift */
  }
  bw_put(b, (uint32_t)f->tx_mode_select, 1);   /* tx_mode_select: TX_MODE_SELECT / TX_MODE_LARGEST */
  bw_put(b, (uint32_t)f->cfg.reduced_tx_set, 1);
}

size_t av1o_write_obus(Av1oFrame *f, uint8_t **tile_data, size_t *tile_len, uint8_t **out) {
  const int ntiles = f->tile_cols * f->tile_rows;
  size_t max_tile = 0, total = 0;
  for (int i = 0; i < ntiles; i++) { if (i + 1 < ntiles && tile_len[i] > max_tile) max_tile = tile_len[i]; total += tile_len[i]; }
  int tsb = 1; while (tsb < 4 && (max_tile ? max_tile - 1 : 0) >= ((size_t)1 << (8 * tsb))) tsb++;
  BitW sh; bw_init(&sh); write_sequence_header(f, &sh);
  BitW fh; bw_init(&fh); write_frame_header(f, &fh, tsb);
  /* frame_obu: header, byte_alignment, tile_group_obu */
  bw_align(&fh);
  if (ntiles > 1) { bw_put(&fh, 0, 1); bw_align(&fh); }   /* tile_start_and_end_present_flag */
  size_t fh_bytes = bw_bytes(&fh);
  size_t frame_payload = fh_bytes + total + (size_t)(ntiles - 1) * (size_t)tsb;
  uint8_t *o = (uint8_t *)malloc(frame_payload + bw_bytes(&sh) + 64), *p = o;
  *p++ = 0x12; *p++ = 0x00;                                  /* temporal delimiter */
  *p++ = 0x0A; p += leb128_put(p, bw_bytes(&sh)); memcpy(p, sh.buf, bw_bytes(&sh)); p += bw_bytes(&sh);
  *p++ = 0x32; p += leb128_put(p, frame_payload); memcpy(p, fh.buf, fh_bytes); p += fh_bytes;
  for (int i = 0; i < ntiles; i++) {
    if (i + 1 < ntiles) { size_t v = tile_len[i] - 1; for (int k = 0; k < tsb; k++) *p++ = (uint8_t)(v >> (8 * k)); }
    memcpy(p, tile_data[i], tile_len[i]); p += tile_len[i];
  }
  free(sh.buf); free(fh.buf);
  *out = o;
  return (size_t)(p - o);
}

/* ------------------------------------------------------------------ top level */
static void *zalloc(size_t n) { return calloc(n ? n : 1, 1); }

int av1o_encode(const Av1oConfig *cfg, const uint16_t *const planes[3], const int strides[3], Av1oResult *out) {
  if (!cfg || cfg->width < 1 || cfg->height < 1 || (cfg->bit_depth != 8 && cfg->bit_depth != 10)) return 4;
  Av1oFrame *f = (Av1oFrame *)zalloc(sizeof(Av1oFrame));
  f->cfg = *cfg;
  if (f->cfg.part_min > f->cfg.part_max) f->cfg.part_min = f->cfg.part_max;
  f->w = cfg->width; f->h = cfg->height; f->bd = cfg->bit_depth; f->np = cfg->mono ? 1 : 3;
  f->mi_cols = 2 * ((f->w + 7) >> 3); f->mi_rows = 2 * ((f->h + 7) >> 3);
  f->sb_cols = (f->mi_cols + 15) >> 4; f->sb_rows = (f->mi_rows + 15) >> 4;
  f->pw = f->sb_cols * 64; f->ph = f->sb_rows * 64; f->stride = f->pw;
  f->mi_stride = f->pw / 4; f->mi_h = f->ph / 4;
  const size_t npx = (size_t)f->pw * f->ph, nmi = (size_t)f->mi_stride * f->mi_h;
  for (int p = 0; p < f->np; p++) {
    f->src[p] = (uint16_t *)zalloc(npx * 2); f->rec[p] = (uint16_t *)zalloc(npx * 2); f->coef[p] = (int32_t *)zalloc(npx * 4);
    f->m_lvl[p] = (uint8_t *)zalloc(nmi); f->m_dc[p] = (uint8_t *)zalloc(nmi); f->m_eob[p] = (uint16_t *)zalloc(nmi * 2);
    /* source with edge replication into the padding (rav1e pads its frames the same way) */
    for (int y = 0; y < f->ph; y++) {
      const uint16_t *srow = planes[p] + (size_t)imin(y, f->h - 1) * strides[p];
      uint16_t *d = f->src[p] + (size_t)y * f->stride;
      for (int x = 0; x < f->pw; x++) d[x] = srow[imin(x, f->w - 1)];
    }
  }
  f->m_bsize = (uint8_t *)zalloc(nmi); f->m_skip = (uint8_t *)zalloc(nmi); f->m_ymode = (uint8_t *)zalloc(nmi);
  f->m_uvmode = (uint8_t *)zalloc(nmi); f->m_txtype = (uint8_t *)zalloc(nmi); f->m_cfl_sign = (uint8_t *)zalloc(nmi);
  f->m_cfl_au = (uint8_t *)zalloc(nmi); f->m_cfl_av = (uint8_t *)zalloc(nmi); f->m_txsize = (uint8_t *)zalloc(nmi); f->m_seg = (uint8_t *)zalloc(nmi);
  f->tx_mode_select = cfg->rdo_tx || cfg->inter_tx_split;
  f->m_angle_y = (int8_t *)zalloc(nmi); f->m_angle_uv = (int8_t *)zalloc(nmi); f->m_decoded = (uint8_t *)zalloc(nmi);
  f->cdef_idx = (int8_t *)zalloc((size_t)f->sb_cols * f->sb_rows);
  av1o_select_quantizers(f);
  av1o_build_costs(f);
  av1o_setup_tiles(f);
  av1o_activity(f);
  av1o_segmentation(f);
  f->enable_cdef = cfg->cdef; f->enable_restoration = cfg->lrf;
  const int ntiles = f->tile_cols * f->tile_rows;
  uint8_t **td = (uint8_t **)zalloc(sizeof(uint8_t *) * (size_t)ntiles); size_t *tl = (size_t *)zalloc(sizeof(size_t) * (size_t)ntiles);
  /* rdo_passes = 2: the first pass is a complete encode whose only product is the CDFs every tile ends with; the second pass prices each tile's
     search against the table of those (everything else -- decisions made in coding order, filters, syntax -- as in a single pass) */
  const int passes = cfg->rdo_passes >= 2 ? 2 : 1;
  if (passes == 2) f->tile_cdf = (uint16_t *)zalloc((size_t)ntiles * CDF_TOTAL * sizeof(uint16_t));
  for (int pass = 0; pass < passes; pass++) {
    if (pass == 1) {
      f->tile_cost = (uint32_t *)zalloc((size_t)ntiles * CDF_TOTAL * sizeof(uint32_t));
      for (int i = 0; i < ntiles; i++) { av1o_costs_from_cdfs(f->tile_cdf + (size_t)i * CDF_TOTAL, f->tile_cost + (size_t)i * CDF_TOTAL); free(td[i]); td[i] = NULL; }
      free(f->tile_cdf); f->tile_cdf = NULL;
      memset(f->m_decoded, 0, nmi);
      for (int p = 0; p < f->np; p++) { free(f->dbk[p]); free(f->lr_type[p]); free(f->lr_set[p]); free(f->lr_xqd[p]); f->dbk[p] = NULL; f->lr_type[p] = NULL; f->lr_set[p] = NULL; f->lr_xqd[p] = NULL; }
    }
    /* phase 1 */
    for (int tr = 0; tr < f->tile_rows; tr++) for (int tc = 0; tc < f->tile_cols; tc++) av1o_search_tile(f, tr, tc);
    f->cost = f->cost0;
    /* loop filter decisions + final reconstruction */
    av1o_deblock_frame(f);
    av1o_cdef_search_and_apply(f);
    av1o_lr_search_and_apply(f);
    /* phase 2 */
    for (int tr = 0; tr < f->tile_rows; tr++) for (int tc = 0; tc < f->tile_cols; tc++)
      tl[tr * f->tile_cols + tc] = av1o_code_tile(f, tr, tc, &td[tr * f->tile_cols + tc]);
  }
  free(f->tile_cost); f->tile_cost = NULL;
  memset(out, 0, sizeof(*out));
  out->obu_len = av1o_write_obus(f, td, tl, &out->obu);
  for (int i = 0; i < ntiles; i++) free(td[i]);
  free(td); free(tl);
  /* results */
  out->recon_stride = f->w;
  for (int p = 0; p < f->np; p++) {
    out->recon[p] = (uint16_t *)malloc((size_t)f->w * f->h * 2);
    int64_t sse = 0;
    for (int y = 0; y < f->h; y++) {
      memcpy(out->recon[p] + (size_t)y * f->w, f->rec[p] + (size_t)y * f->stride, (size_t)f->w * 2);
      for (int x = 0; x < f->w; x++) { int d = (int)f->rec[p][(size_t)y * f->stride + x] - (int)f->src[p][(size_t)y * f->stride + x]; sse += (int64_t)d * d; }
    }
    out->total_sse[p] = sse;
  }
  out->mi_cols = f->mi_cols; out->mi_rows = f->mi_rows; out->mi_stride = f->mi_stride;
  out->m_bsize = f->m_bsize; out->m_ymode = f->m_ymode; out->m_uvmode = f->m_uvmode; out->m_skip = f->m_skip; out->m_txtype = f->m_txtype;
  for (int i = 0; i < 4; i++) out->lf_level[i] = f->lf_level[i];
  out->seg_n = f->seg_n; for (int i = 0; i < 8; i++) out->seg_qidx[i] = f->seg_qidx[i];
  if (f->seg_n) for (size_t i = 0; i < nmi; i++) f->m_skip[i] = (uint8_t)(f->m_skip[i] | (f->m_seg[i] << 1));      /* the dump uses the product's packing */
  out->base_q_idx = f->base_q_idx; out->tile_cols = f->tile_cols; out->tile_rows = f->tile_rows;
  for (int p = 0; p < f->np; p++) { free(f->src[p]); free(f->rec[p]); free(f->coef[p]); free(f->m_lvl[p]); free(f->m_dc[p]); free(f->m_eob[p]); }
  for (int p = 0; p < f->np; p++) { free(f->dbk[p]); free(f->lr_type[p]); free(f->lr_set[p]); free(f->lr_xqd[p]); }
  free(f->m_txsize); free(f->m_seg); free(f->act); free(f->svar8); free(f->svar4);
  free(f->m_cfl_sign); free(f->m_cfl_au); free(f->m_cfl_av); free(f->m_angle_y); free(f->m_angle_uv); free(f->m_decoded); free(f->cdef_idx);
  free(f);
  return 0;
}

void av1o_free_result(Av1oResult *r) {
  if (!r) return;
  free(r->obu); for (int p = 0; p < 3; p++) free(r->recon[p]);
  free(r->m_bsize); free(r->m_ymode); free(r->m_uvmode); free(r->m_skip); free(r->m_txtype);
  memset(r, 0, sizeof(*r));
}

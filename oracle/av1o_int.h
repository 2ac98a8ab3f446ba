/* oracle/av1o_int.h -- internal types of the CPU oracle (test infrastructure). */
#ifndef ORACLE_AV1O_INT_H
#define ORACLE_AV1O_INT_H
#include "av1o.h"
#include "av1_tables.h"
#include <stdlib.h>
#include <string.h>

/* rav1e rdo_tx_size_type searches tx_depth 0..2 (rdo_tx_depth = 2) when rdo_tx_decision is on; 1 = one level only (round-2 behaviour) */
#ifndef AV1O_TX_DEPTH_MAX
#define AV1O_TX_DEPTH_MAX 2
#endif

/* Rectangular partitions (PARTITION_HORZ / PARTITION_VERT) of 8x8 nodes: rav1e tries them up to its
 * non_square_partition_max_threshold (8x8 from speed 2 on) [UPSTREAM-RECALL].  Always searched since round 3 (oracle and HIP path). */
/* Block-size and transform-size codes: 0..4 = the squares 4 << code; 5 = 4 wide x 8 tall, 6 = 8 wide x 4 tall (same numbering for both). */
enum { BS_4X8 = 5, BS_8X4 = 6, TX_4X8 = 5, TX_8X4 = 6 };
static inline int dim_wl(int code) { return code <= 4 ? 2 + code : (code == 5 ? 2 : 3); }     /* log2 of the width in samples */
static inline int dim_hl(int code) { return code <= 4 ? 2 + code : (code == 5 ? 3 : 2); }     /* log2 of the height */
static inline int dim_is_rect(int code) { return code > 4; }
static inline int dim_min_l(int code) { const int a = dim_wl(code), b = dim_hl(code); return a < b ? a : b; }
static inline int dim_max_l(int code) { const int a = dim_wl(code), b = dim_hl(code); return a > b ? a : b; }
static inline int dim_code(int wl, int hl) { return wl == hl ? wl - 2 : (wl == 2 && hl == 3 ? 5 : 6); }

#define MI 4
#define SB 64
#define SB_MI 16
#define MAX_TILE_COLS 64
#define MAX_TILE_ROWS 64

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs(int a) { return a < 0 ? -a : a; }
static inline int round2(int x, int n) { return n == 0 ? x : (x + (1 << (n - 1))) >> n; }
static inline int64_t round2_64(int64_t x, int n) { return n == 0 ? x : (x + ((int64_t)1 << (n - 1))) >> n; }

/* ---------------- bit writer (OBU headers, spec f(n)) ---------------- */
typedef struct { uint8_t *buf; size_t cap, bitpos; } BitW;
void bw_init(BitW *b);
void bw_put(BitW *b, uint32_t v, int n);
void bw_su(BitW *b, int v, int n);           /* su(n) */
void bw_align(BitW *b);                       /* byte_alignment(): zero bits */
void bw_trailing(BitW *b);                    /* trailing_bits(): 1 then zeros */
size_t bw_bytes(const BitW *b);
size_t leb128_put(uint8_t *dst, uint64_t v);  /* returns length */

/* ---------------- range encoder (spec 8.2 inverse; daala od_ec_enc) ---------------- */
typedef struct {
  uint16_t *pre; size_t pre_cap, offs;
  uint64_t low; uint32_t rng; int cnt;
} RangeEnc;
void re_init(RangeEnc *e);
void re_free(RangeEnc *e);
void re_symbol(RangeEnc *e, int s, uint16_t *icdf, int nsyms);    /* encodes + adapts (spec 8.2.6 / 8.3.2) */
void re_symbol_noadapt(RangeEnc *e, int s, const uint16_t *icdf, int nsyms);
void re_literal(RangeEnc *e, uint32_t v, int nbits);               /* L(n): equiprobable bits, MSB first */
size_t re_finish(RangeEnc *e, uint8_t **out);                      /* returns malloc'd bytes */

/* ---------------- frame state ---------------- */
typedef struct Av1oFrame {
  Av1oConfig cfg;
  int w, h, bd, np;
  int mi_cols, mi_rows, sb_cols, sb_rows;
  int pw, ph, stride;           /* padded plane geometry (multiples of 64) */
  int mi_stride, mi_h;          /* maps geometry: pw/4, ph/4 */
  uint16_t *src[3], *rec[3];
  int32_t *coef[3];             /* signed quantized levels at pixel positions */
  /* mode-info maps, one entry per 4x4 (mi) */
  uint8_t *m_bsize, *m_skip, *m_ymode, *m_uvmode, *m_txtype, *m_cfl_sign, *m_cfl_au, *m_cfl_av;
  uint8_t *m_txsize;            /* luma transform size of the block (TX_4X4 .. TX_64X64), uniform over the block */
  int tx_mode_select;           /* TX_MODE_SELECT: rdo_tx_decision || inter_tx_split (rav1e FrameInvariants, recall) */
  int8_t *m_angle_y, *m_angle_uv;
  uint8_t *m_lvl[3], *m_dc[3];  /* coefficient contexts left behind by each tx block */
  uint16_t *m_eob[3];           /* eob of the tx block whose top-left mi this is */
  uint8_t *m_decoded;           /* "BlockDecoded" flag per mi (valid inside the current SB only) */
  int8_t *cdef_idx;             /* per 64x64 (sb_rows * sb_cols), -1 = none coded */
  /* quantizer */
  int base_q_idx, qctx;
  int dc_qi[3], ac_qi[3];       /* absolute q indices per plane */
  int dc_q[3], ac_q[3];         /* quantizer step values per plane */
  int64_t rdmult[3];            /* lambda in (1/128 SSE per 1/512 bit) units per plane */
  /* tiles (in SB units) */
  int tile_cols_log2, tile_rows_log2, tile_cols, tile_rows;
  int tile_col_start[MAX_TILE_COLS + 1], tile_row_start[MAX_TILE_ROWS + 1];
  /* static rate table: cost[CDF offset + symbol] in 1/512 bit */
  uint16_t cdf0[CDF_TOTAL];
  uint32_t cost0[CDF_TOTAL];   /* ... of the frame's initial CDFs */
  const uint32_t *cost;        /* the table the search prices against: cost0, or the tile's own in a second pass (rdo_passes = 2) */
  uint32_t *tile_cost;         /* [tiles][CDF_TOTAL], second pass only */
  uint16_t *tile_cdf;          /* [tiles][CDF_TOTAL]: the CDFs every tile ended its first pass with (rdo_passes = 2) */
  /* loop filter / cdef frame params */
  int lf_level[4], lf_sharp;
  int cdef_damping, cdef_bits, cdef_y[8], cdef_uv[8];
  int enable_cdef, enable_restoration;
  /* loop restoration: deblocked (pre-CDEF) frame, per-unit decisions (64x64 units, raster over lr_unit_rows x lr_unit_cols) */
  uint16_t *dbk[3];
  int lr_unit_rows, lr_unit_cols;
  uint8_t *lr_type[3], *lr_set[3]; int8_t *lr_xqd[3];
  uint32_t lr_cost[3];          /* static cost of the switchable restoration_type symbols */
  /* Tune::Psychovisual: per 8x8 cell activity scale (Q14) and source variance, per 4x4 source variance (8x8-equivalent) */
  uint32_t *act, *svar8, *svar4;
  /* segmentation (av1o_segment.c): seg_n = 0 -> off; segment i: absolute luma-AC index and the per-plane steps; thresholds in scale buckets */
  uint8_t *m_seg;               /* segment id per mi (the predicted id where the block is skipped, spec read_segment_id) */
  int seg_n, seg_mean, seg_thr[7], seg_qidx[8], seg_dcq[8][3], seg_acq[8][3];
  int64_t sse[3];
} Av1oFrame;

typedef struct { int mi_row_start, mi_row_end, mi_col_start, mi_col_end; } TileB;

/* common */
int  av1o_tx_set(int txs, int reduced);                 /* 0 dct-only, 1 = 7 types, 2 = 5 types (spec get_tx_set, intra) */
int  av1o_tx_set_count(int set);
int  av1o_tx_type_in_set(int set, int txtype);
int  av1o_tx_type_to_symbol(int set, int txtype);       /* inverse of Tx_Type_Intra_Inv_SetN */
int  av1o_symbol_to_tx_type(int set, int sym);
int  av1o_mode_to_txtype(int mode);                      /* Mode_To_Txfm */
int  av1o_tx_class(int txtype);
const uint16_t *av1o_scan(int txs, int txtype, uint16_t *tmp); /* positions within min(32,N) square */
uint32_t av1o_cost_from_icdf(const uint16_t *icdf, int s, int nsyms);
void av1o_build_costs(Av1oFrame *f);
void av1o_costs_from_cdfs(const uint16_t *cdf, uint32_t *cost);   /* the static rate table of one set of CDFs */
void av1o_select_quantizers(Av1oFrame *f);               /* rav1e rate.rs constant-Q key frame rule (recall) */
void av1o_setup_tiles(Av1oFrame *f);
uint32_t av1o_psy_boost_q14(uint32_t svar, uint32_t dvar);
uint32_t av1o_cell_var(int64_t sum, int64_t sum2, int w, int bd);
void av1o_activity(Av1oFrame *f);
int64_t av1o_psy_dist_luma(const Av1oFrame *f, const uint16_t *rec, int rs, int x, int y, int n);
int64_t av1o_psy_dist_luma_wh(const Av1oFrame *f, const uint16_t *rec, int rs, int x, int y, int bw, int bh);
uint32_t av1o_act_mean(const Av1oFrame *f, int x, int y, int w, int h);
/* segmentation */
#define AV1O_SEG_BINS 4096
int  av1o_ilog2_q11(uint32_t x);
int  av1o_seg_bucket(uint32_t scale_q14);
void av1o_segmentation(Av1oFrame *f);
int  av1o_block_segment(const Av1oFrame *f, int x, int y, int w, int h);
int  av1o_seg_pred(int prev_ul, int prev_u, int prev_l, int *ctx);
int  av1o_seg_symbol(int seg, int pred, int max);

/* prediction (spec 7.11.2) */
typedef struct { uint16_t above[2 * 64 + 16 + 32], left[2 * 64 + 16 + 32]; } EdgeBuf;  /* index +16 = position 0 */
void av1o_predict_intra(const Av1oFrame *f, const TileB *t, int plane, int x, int y, int log2w,
                        int have_left, int have_above, int have_above_rt, int have_below_lft,
                        int mode, int angle_delta, int filter_type, uint16_t *dst, int dst_stride);
void av1o_predict_cfl(const Av1oFrame *f, int plane, int x, int y, int log2w, int alpha, uint16_t *dst, int dst_stride);
void av1o_predict_intra_wh(const Av1oFrame *f, const TileB *t, int plane, int x, int y, int log2w, int log2h,
                           int have_left, int have_above, int have_above_rt, int have_below_lft,
                           int mode, int angle_delta, int filter_type, uint16_t *dst, int dst_stride);
void av1o_predict_cfl_wh(const Av1oFrame *f, int plane, int x, int y, int log2w, int log2h, int alpha, uint16_t *dst, int dst_stride);

/* transforms + quant (spec 7.13.3 inverse; forward = encoder side) */
void av1o_fwd_txfm2d(const int16_t *resid, int rstride, int32_t *coef, int txs, int txtype, int bd);
void av1o_inv_txfm2d_add(const int32_t *dq, uint16_t *dst, int dstride, int txs, int txtype, int bd);
int  av1o_quantize(const int32_t *coef, int32_t *qc, int txs, int txtype, int dcq, int acq);  /* returns eob */
void av1o_dequantize(const int32_t *qc, int32_t *dq, int txs, int dcq, int acq, int bd, int eob, int txtype);

/* phase 1: decisions + reconstruction (all tiles) ; phase 2: entropy coding of one tile */
void   av1o_search_tile(Av1oFrame *f, int tile_row, int tile_col);
size_t av1o_code_tile(Av1oFrame *f, int tile_row, int tile_col, uint8_t **out);
uint32_t av1o_coef_rate(const Av1oFrame *f, const int32_t *qc, int eob, int plane, int txs, int txtype,
                        int txb_skip_ctx, int dc_sign_ctx, int *cul_level, int *dc_cat);
void av1o_txb_ctx(const Av1oFrame *f, const TileB *t, int plane, int r4, int c4, int txs, int bs, int *skip_ctx, int *dc_ctx);

void *av1o_live_open(Av1oFrame *f, int tile_row, int tile_col); void av1o_live_sb(void *w, int r, int c, uint32_t *cost_out); void av1o_live_close(void *w);   /* AV1O_LIVE_CDF experiment */
/* loop filters (spec 7.14, 7.15) */
void av1o_deblock_frame(Av1oFrame *f);
void av1o_cdef_search_and_apply(Av1oFrame *f);
/* loop restoration (spec 7.17) */
void av1o_lr_search_and_apply(Av1oFrame *f);
int  av1o_lr_units(int size);
void av1o_write_lr_sb(Av1oFrame *f, int r, int c, int ref_xqd[3][2], void (*sym)(void *, int, int), void (*lit)(void *, uint32_t, int), void *u);

/* headers */
size_t av1o_write_obus(Av1oFrame *f, uint8_t **tile_data, size_t *tile_len, uint8_t **out);
#endif

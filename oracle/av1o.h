/* oracle/av1o.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar C restatement of the AV1 still-picture (intra-only) encode path that cavif-rs reaches
 * through rav1e:  ravif/src/av1encoder.rs:749-771 (encode_to_av1) -> rav1e ^0.8.1 (absent from
 * /root/reference, unpinned: Cargo.lock is git-ignored).  PARITY UNPINNED: no golden bitstream
 * exists in the reference and rav1e cannot be built here, so this oracle is pinned instead by
 *   (1) dav1d 1.5.3 decoding every emitted stream to exactly the oracle's reconstruction,
 *   (2) the reference's own size-window tests (ravif/src/lib.rs:43-147),
 *   (3) the front-end known answers of SURVEY.md section 8a.
 * Everything normative (prediction, inverse transforms, dequant, CDFs, range coder, OBU syntax,
 * deblock, CDEF) follows the AV1 bitstream specification; encoder-side choices follow rav1e's
 * structure as recalled and are documented in DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this directory.
 * NOT RE-ENTRANT: the search keeps its block scratch in function-local statics (av1o_search.c try_block / eval_tx, av1o_entropy.c);
 * one encode at a time per process (the tests and bench.py fan out over processes, never threads).
 */
#ifndef ORACLE_AV1O_H
#define ORACLE_AV1O_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (AV1 spec section 6.10.x / symbols) ---- */
enum { BS_4 = 0, BS_8 = 1, BS_16 = 2, BS_32 = 3, BS_64 = 4 };          /* square block sizes only: px = 4 << bs */
enum { TX_4X4 = 0, TX_8X8, TX_16X16, TX_32X32, TX_64X64 };
enum { DC_PRED = 0, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED,
       SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED, UV_CFL_PRED, N_INTRA_MODES = 13 };
enum { DCT_DCT = 0, ADST_DCT, DCT_ADST, ADST_ADST, FLIPADST_DCT, DCT_FLIPADST, FLIPADST_FLIPADST,
       ADST_FLIPADST, FLIPADST_ADST, IDTX, V_DCT, H_DCT, V_ADST, H_ADST, V_FLIPADST, H_FLIPADST };
enum { PARTITION_NONE = 0, PARTITION_HORZ, PARTITION_VERT, PARTITION_SPLIT };
enum { TX_CLASS_2D = 0, TX_CLASS_HORIZ, TX_CLASS_VERT };

/* mirrors Av1EncodeConfig + SpeedTweaks (ravif/src/av1encoder.rs:533-552, 649-660) */
typedef struct Av1oConfig {
  int width, height;
  int bit_depth;          /* 8 | 10 */
  int mono;               /* 1 = ChromaSampling::Cs400 (alpha plane), 0 = Cs444 */
  int quantizer;          /* rav1e quantizer 0..255 (quality_to_quantizer) */
  int full_range;         /* PixelRange::Full */
  int has_color_desc, color_primaries, transfer, matrix;
  int threads;            /* <=0: unspecified */
  /* SpeedTweaks */
  int part_min, part_max; /* px */
  int complex_modes, fine_directional, rdo_tx, reduced_tx_set, fast_deblock, cdef, lrf, sgr_full,
      bottomup, tx_domain_rate, inter_tx_split;
  int min_tile_size;
  int tiles_override;     /* >0: force this tile target (tests) */
  int tune_psnr;          /* 0 = Tune::Psychovisual (what ravif always sets, av1encoder.rs:694), 1 = Tune::Psnr (plain SSE; ablation) */
  int rdo_passes;         /* <= 1: the search prices against the table of the frame's initial CDFs; 2: the whole encode runs twice and the second search
                             prices every tile against the CDFs that tile ended its first pass with (an extension towards rav1e's adaptive pricing; not in ravif) */
} Av1oConfig;

typedef struct Av1oResult {
  uint8_t *obu; size_t obu_len;           /* TD + sequence header + frame OBU */
  uint16_t *recon[3]; int recon_stride;   /* final (post-filter) reconstruction, visible w x h */
  /* decision dump for GPU parity debugging */
  int mi_cols, mi_rows, mi_stride;
  uint8_t *m_bsize, *m_ymode, *m_uvmode, *m_skip, *m_txtype;
  int base_q_idx;
  int tile_cols, tile_rows;
  int64_t total_sse[3];
  int lf_level[4];                         /* deblock levels: luma vertical / horizontal edges, U, V */
  int seg_n, seg_qidx[8];                  /* segmentation: segments in use (0 = off) and their luma-AC quantiser indices; m_skip carries (segment id << 1) | skip */
} Av1oResult;

int  av1o_tweaks_from_preset(int speed, int quantizer, Av1oConfig *c);   /* av1encoder.rs:554-606 */
int  av1o_encode(const Av1oConfig *cfg, const uint16_t *const planes[3], const int strides[3], Av1oResult *out);
void av1o_free_result(Av1oResult *r);

/* ---- ravif-level front end (ravif/src/av1encoder.rs) ---- */
int      av1o_quality_to_quantizer(float quality);                          /* :526-530 */
void     av1o_rgb_to_ycbcr(const uint8_t rgb[3], int depth, uint16_t out[3]); /* :504-524 BT.601 */
uint16_t av1o_to_ten(uint8_t x);                                             /* :485-487 */

typedef struct RavifOracleEncoder {       /* ravif::Encoder (:67-86) */
  float quality, alpha_quality; int speed; int color_model /*0 YCbCr,1 RGB*/; int depth /*8,10,0=auto*/;
  int alpha_mode /*0 dirty,1 clean,2 premultiplied*/; int threads; int tiles_override; int rdo_passes;
} RavifOracleEncoder;
typedef struct RavifOracleImage { uint8_t *avif; size_t avif_len, color_byte_size, alpha_byte_size; } RavifOracleImage;
int  ravif_oracle_encode_rgba(const RavifOracleEncoder *e, const uint8_t *rgba, int w, int h, int stride_px, RavifOracleImage *out);
int  ravif_oracle_encode_rgb(const RavifOracleEncoder *e, const uint8_t *rgb, int w, int h, int stride_px, RavifOracleImage *out);
size_t av1o_avif_container(const uint8_t *color, size_t color_len, const uint8_t *alpha, size_t alpha_len,
                           int w, int h, int depth, int mono_color, int cp, int tc, int mc, int full_range,
                           int premultiplied, uint8_t **out);
void av1o_free(void *p);
/* ravif/src/dirtyalpha.rs */
void av1o_premultiplied_minmax(uint8_t px, uint8_t alpha, uint8_t *lo, uint8_t *hi);
int  av1o_blurred_dirty_alpha(const uint8_t *rgba, int w, int h, int stride_px, uint8_t *out /* w*h*4 */);

#ifdef __cplusplus
}
#endif
#endif

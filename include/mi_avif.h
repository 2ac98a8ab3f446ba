/* mi_avif.h -- C ABI of the MI355X-native AV1 still-picture encode path (libmi_avif.so).
 *
 * Drop-in boundary for the one hot path cavif-rs delegates to rav1e.  The reference has no FFI here: the
 * path sits behind rav1e's Rust API, used in ravif/src/av1encoder.rs:749-771 (encode_to_av1) and configured
 * at :662-708 (rav1e_config).  Each entry point below names the reference interface it stands in for; the
 * Rust binding a ravif maintainer would add is shown in INTEGRATION.md.
 *
 * Threading: every function is re-entrant and thread-safe (no global mutable state beyond per-device
 * read-only tables built on first use); each call/batch owns its HIP stream.  All pointers are plain host
 * pointers unless a name says `dev`.  Buffers returned through `uint8_t**` / mi_encoded_image are owned by
 * the caller and released with mi_free().
 *
 * Status codes mirror ravif::Error (ravif/src/error.rs:7-25) plus the builder asserts (:117,146,159,188).
 */
#ifndef MI_AVIF_H
#define MI_AVIF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MI_OK = 0, MI_TOO_FEW_PIXELS = 1, MI_UNSUPPORTED = 2, MI_ENCODING_ERROR = 3, MI_INVALID_ARGUMENT = 4, MI_NO_DEVICE = 5 };

/* ---- level 1: one AV1 frame; mirrors Av1EncodeConfig (:649-660) + SpeedTweaks (:533-552) ---- */
typedef struct mi_av1_config {
  uint32_t width, height;
  uint8_t bit_depth;        /* 8 | 10 */
  uint8_t quantizer;        /* rav1e quantizer 0..255 */
  uint8_t speed;            /* 1..10 (informational once the tweaks below are filled) */
  uint8_t chroma;           /* 0 = Cs444, 1 = Cs400 */
  uint8_t pixel_range;      /* 0 = Limited, 1 = Full */
  int32_t threads;          /* bounds the tile target min(threads, w*h / min_tile_size^2) (:665-668).  <= 0 = unspecified: the
                               reference then takes rayon::current_num_threads() (:666), i.e. the host's core count; a GPU has no
                               such number, so the target is left uncapped (1080p speed 4 -> 31, i.e. 32 tiles).  Pass the
                               host's core count to get the reference's tile split: the Rust binding of INTEGRATION.md and the
                               cavif_mi command line do (no -j = this host's logical cores). */
  int8_t has_color_desc; uint8_t matrix, transfer, primaries;
  /* resolved SpeedTweaks (mi_av1_tweaks_from_preset fills them; callers may override) */
  uint8_t part_min, part_max, complex_pred_modes, sgr_full, encode_bottomup, rdo_tx_decision,
          reduced_tx_set, fine_directional_intra, fast_deblock, lrf, cdef, inter_tx_split, tx_domain_rate;
  int8_t tx_domain_distortion;
  uint16_t min_tile_size;
  int32_t tiles_override;   /* >0 forces the tile target (tests) */
  int32_t device;           /* HIP ordinal */
  uint8_t tune_psnr;        /* 0 = Tune::Psychovisual, what ravif always sets (:694); 1 = Tune::Psnr (plain SSE; ablation only) */
  uint8_t rdo_passes;       /* exactly 2 selects two-pass pricing, ANY other value means one pass (the struct grew by this field: a caller that fills it field by field
                               without zeroing must not switch modes by accident -- zero-initialise the struct, or start from mi_av1_tweaks_from_preset).
                               One pass: the tile search prices against the table of the frame's initial CDFs.  2 (an extension, not in ravif): the
                               whole encode runs twice and the second search prices every tile against the CDFs that tile ended the first
                               pass with -- a step towards rav1e's adaptive pricing that keeps tiles and superblock rows independent */
} mi_av1_config;

/* SpeedTweaks::from_my_preset (ravif/src/av1encoder.rs:554-606) */
int mi_av1_tweaks_from_preset(uint8_t speed, uint8_t quantizer, mi_av1_config *cfg);
/* quality_to_quantizer (ravif/src/av1encoder.rs:526-530) */
int mi_quality_to_quantizer(float quality);
/* rgb_to_ycbcr + casts (ravif/src/av1encoder.rs:504-524), host evaluation of the same f32 FMA chain (known-answer tests) */
void mi_rgb_to_ycbcr(const uint8_t rgb[3], int depth, uint16_t out[3]);

/* encode_to_av1 (ravif/src/av1encoder.rs:749-771): planes are host pointers to 8-bit (uint8) or 10-bit (uint16)
 * samples, Y/U/V or a single plane for Cs400.  Returns the concatenated KEY-frame OBUs (TD + sequence header +
 * frame).  recon (optional, may be NULL) receives malloc'd uint16 planes of the final reconstruction. */
int mi_av1_encode_planes(const mi_av1_config *cfg, const void *const planes[3], const size_t stride_bytes[3],
                         uint8_t **out_obu, size_t *out_len, uint16_t *recon[3]);

/* ---- level 2: ravif::Encoder (:67-86) and EncodedImage (:54-61) ---- */
typedef struct mi_ravif_encoder {
  float quality, alpha_quality;   /* with_quality :116, with_alpha_quality :145 */
  uint8_t speed;                  /* with_speed :158 */
  uint8_t color_model;            /* 0 YCbCr, 1 RGB (with_internal_color_model :174) */
  uint8_t depth;                  /* 8, 10, 0 = Auto (== 10, :266,:339) */
  uint8_t alpha_mode;             /* 0 UnassociatedDirty, 1 UnassociatedClean, 2 Premultiplied (:197) */
  int32_t threads;                /* with_num_threads :187; <=0 = None */
  const uint8_t *exif; size_t exif_len;   /* with_exif :208; copied by mi_batch_create / every encode call, need not outlive it */
  int32_t device;
  int32_t tiles_override;
  int32_t rdo_passes;             /* extension: see mi_av1_config.rdo_passes (exactly 2 = two passes; anything else = one pass, ravif's behaviour).  The struct must be
                                     zero-initialised or come from mi_ravif_encoder_default: both structs have grown at the tail and may grow again */
} mi_ravif_encoder;

typedef struct mi_encoded_image { uint8_t *avif_file; size_t avif_len, color_byte_size, alpha_byte_size; } mi_encoded_image;

void mi_ravif_encoder_default(mi_ravif_encoder *e);                      /* Encoder::new (:88-102) */
int  mi_ravif_encode_rgba(const mi_ravif_encoder *e, const uint8_t *rgba, uint32_t w, uint32_t h, size_t stride_px, mi_encoded_image *out);  /* :243 */
int  mi_ravif_encode_rgb (const mi_ravif_encoder *e, const uint8_t *rgb,  uint32_t w, uint32_t h, size_t stride_px, mi_encoded_image *out);  /* :318 */
/* encode_raw_planes_8_bit / _10_bit (:366,:390): interleaved [Y,U,V] triples + optional alpha plane */
int  mi_ravif_encode_raw_planes_8 (const mi_ravif_encoder *e, uint32_t w, uint32_t h, const uint8_t  *yuv, const uint8_t  *alpha, uint8_t range, uint8_t matrix, mi_encoded_image *out);
int  mi_ravif_encode_raw_planes_10(const mi_ravif_encoder *e, uint32_t w, uint32_t h, const uint16_t *yuv, const uint16_t *alpha, uint8_t range, uint8_t matrix, mi_encoded_image *out);

/* ---- many images, all GPUs of the node: the reference's files.into_par_iter() (src/main.rs:223) ----
 * One host thread per device pulls runs of equally-shaped images and pushes each run through a resident batch; images are
 * independent (no collective).  devices == NULL / ndev <= 0: every visible device.  status (nullable) gets one code per image;
 * the return value is the first failure.  out[i].avif_file is malloc'd (mi_free). */
typedef struct mi_image_desc { const uint8_t *pixels; uint32_t width, height; size_t stride_px /* 0 = width */; int channels /* 3 RGB8 | 4 RGBA8 */; } mi_image_desc;
int  mi_ravif_encode_batch(const mi_ravif_encoder *e, size_t n, const mi_image_desc *in, mi_encoded_image *out, int *status, const int *devices, int ndev);
/* streaming form: image i is pulled through `fetch` right before it is staged (the callback may block until a loader has produced
 * the pixels), so file loading overlaps the GPU work -- what rayon's work stealing gives the reference when load and encode sit in
 * one par_iter body (src/main.rs:179-223).  fetch returns MI_OK or the image's status.  `release` (nullable) is called once per
 * successfully fetched image as soon as its pixels have been copied into the pinned staging: the caller may free them and let its
 * loaders run further ahead (bounded host memory for any number of files).  Without it the pixels must stay valid until the call
 * returns.  Indices are fetched in increasing order per device thread; both callbacks may be called from several threads. */
typedef int (*mi_fetch_fn)(void *user, size_t index, mi_image_desc *desc);
typedef void (*mi_release_fn)(void *user, size_t index);
int  mi_ravif_encode_stream(const mi_ravif_encoder *e, size_t n, mi_fetch_fn fetch, mi_release_fn release, void *user, mi_encoded_image *out, int *status, const int *devices, int ndev);

/* The one-call entry points above (mi_ravif_encode_rgba / _rgb / _batch / _stream) keep their device arenas and pinned staging in
 * a process-wide pool keyed by (device, shape, settings), so a loop of calls with the same settings pays the allocation once
 * (what a long-lived rav1e thread pool is to the reference).  At most 12 objects / 96 GB are retained; this frees them now. */
void mi_release_cached(void);

/* PNG -> RGBA8 as cavif's load_rgba does (src/main.rs:265-283: RGB gets alpha 255, 16-bit samples keep their high byte, gray is
 * replicated); all colour types, bit depths, tRNS and Adam7.  Host code over zlib.  *rgba is malloc'd (mi_free), w*h*4 bytes. */
int  mi_png_decode_rgba(const uint8_t *data, size_t len, uint8_t **rgba, uint32_t *w, uint32_t *h);

/* ---- batch: the data-parallel path (src/main.rs:223 files.into_par_iter()); images resident in HBM ---- */
typedef struct mi_batch mi_batch;
/* n images of w x h, channels 3 (RGB8) or 4 (RGBA8) on HIP device `e->device` */
mi_batch *mi_batch_create(const mi_ravif_encoder *e, int n_images, uint32_t w, uint32_t h, int channels);
int  mi_batch_upload(mi_batch *b, int index, const uint8_t *pixels, size_t stride_px);   /* copy into the pinned staging + H2D into the batch's HBM input slot (blocking) */
/* zero-copy form: fill the batch's PINNED host staging of image `index` (w*h*channels bytes, rows packed) in place, then enqueue
 * the H2D of a range of images on the batch's stream (returns at once; ordered before the next mi_batch_encode[_async]) */
uint8_t *mi_batch_input(mi_batch *b, int index);
int  mi_batch_upload_async(mi_batch *b, int first, int count);
int  mi_batch_set_count(mi_batch *b, int n_images);                                       /* images of the next run (<= the count the batch was created for) */
int  mi_batch_encode(mi_batch *b);                                                        /* the hot path over all resident images */
/* split form: enqueue the GPU work and return; wait = sync + one packed D2H + OBU/container assembly.  Two batches
 * driven alternately overlap one batch's entropy coding / loop filters with the next batch's tile search. */
int  mi_batch_encode_async(mi_batch *b);
int  mi_batch_wait(mi_batch *b);
int  mi_batch_get(mi_batch *b, int index, mi_encoded_image *out);                         /* copies; caller frees avif_file */
int  mi_batch_get_recon(mi_batch *b, int index, int alpha, uint16_t *planes[3]);          /* malloc'd w*h uint16 planes (tests) */
/* per-kernel HIP-event time (ms) of the last mi_batch_encode: 0 front-end, 1 tile search, 2 deblock, 3 cdef, 4 entropy, 5 pack+D2H, 6 host assembly */
double mi_batch_stage_ms(const mi_batch *b, int stage);
int  mi_batch_num_tiles(const mi_batch *b);
/* profiling aid: per tile [K1 start, K1 end, K4 start, K4 end] in wall_clock64 ticks (100 MHz) of the last encode */
int  mi_batch_tile_clocks(mi_batch *b, unsigned long long *out);
void mi_batch_destroy(mi_batch *b);

/* AVIF container (avif-serialize Aviffy::to_vec, call site ravif/src/av1encoder.rs:457-473) */
size_t mi_avif_serialize(const uint8_t *color, size_t color_len, const uint8_t *alpha, size_t alpha_len,
                         uint32_t w, uint32_t h, uint8_t depth, uint8_t matrix, int premultiplied,
                         const uint8_t *exif, size_t exif_len, uint8_t **out);
int  mi_device_count(void);
void mi_free(void *p);
const char *mi_version(void);

#ifdef __cplusplus
}
#endif
#endif

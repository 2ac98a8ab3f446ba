/* oracle/ravif_front.c -- restatement of the in-tree ravif layer (TEST INFRASTRUCTURE, see av1o.h):
 *   quality_to_quantizer          ravif/src/av1encoder.rs:526-530
 *   SpeedTweaks::from_my_preset   ravif/src/av1encoder.rs:554-606
 *   rgb_to_ycbcr / to_ten / gbr   ravif/src/av1encoder.rs:485-524
 *   encode_rgba / encode_rgb / encode_raw_planes_internal   :243-275, :318-350, :400-480
 *   AVIF container (avif-serialize ^0.8.6, absent)  call site :457-473; layout per ISO/IEC 23008-12 + AV1-ISOBMFF
 * Compile with -ffp-contract=off: the only fused operations are the explicit fmaf() calls that mirror mul_add. */
#include "av1o_int.h"
#include <math.h>

int av1o_quality_to_quantizer(float quality) {
  float q = quality / 100.f;
  float x = q >= 0.82f ? (1.f - q) * 2.6f : (q > 0.25f ? fmaf(q, -0.5f, 1.f - 0.125f) : 1.f - q);
  float r = roundf(x * 255.f);
  return r <= 0.f ? 0 : (r >= 255.f ? 255 : (int)r);
}

int av1o_tweaks_from_preset(int speed, int quantizer, Av1oConfig *c) {
  if (speed < 1 || speed > 10) return 4;
  const int low_quality = quantizer < av1o_quality_to_quantizer(55.f);
  const int high_quality = quantizer > av1o_quality_to_quantizer(80.f);
  const int max_block = high_quality ? 16 : 64;
  int mn, mx;
  if (speed == 0) { mn = 4; mx = imin(64, max_block); }
  else if (speed == 1 && low_quality) { mn = 4; mx = imin(64, max_block); }
  else if (speed == 2 && low_quality) { mn = 4; mx = imin(32, max_block); }
  else if (speed >= 1 && speed <= 4) { mn = 4; mx = 16; }
  else if (speed >= 5 && speed <= 8) { mn = 8; mx = 16; }
  else { mn = 16; mx = 16; }
  c->part_min = mn; c->part_max = mx;
  c->complex_modes = speed <= 1; c->sgr_full = speed <= 2; c->bottomup = speed <= 2;
  c->rdo_tx = speed <= 4 && !high_quality; c->reduced_tx_set = speed == 4 || speed >= 9;
  c->fine_directional = speed <= 6; c->fast_deblock = speed >= 7 && !high_quality;
  c->lrf = low_quality && speed <= 8; c->cdef = low_quality && speed <= 9;
  c->inter_tx_split = speed >= 9; c->tx_domain_rate = speed >= 10;
  static const int mts[5] = { 4096, 2048, 1024, 512, 256 };
  c->min_tile_size = (speed <= 4 ? mts[speed] : 128) * (high_quality ? 2 : 1);
  return 0;
}

uint16_t av1o_to_ten(uint8_t x) { return (uint16_t)(((uint16_t)x << 2) | ((uint16_t)x >> 6)); }

void av1o_rgb_to_ycbcr(const uint8_t rgb[3], int depth, uint16_t out[3]) {
  static const float m[3] = { 0.2990f, 0.5870f, 0.1140f };       /* BT601 */
  const float max_value = (float)((1 << depth) - 1);
  const float scale = max_value / 255.f;
  const float shift = roundf(max_value * 0.5f);
  const float r = (float)rgb[0], g = (float)rgb[1], b = (float)rgb[2];
  const float y = fmaf(scale * m[2], b, fmaf(scale * m[0], r, scale * m[1] * g));
  const float cb = fmaf(fmaf(b, scale, -y), 0.5f / (1.f - m[2]), shift);
  const float cr = fmaf(fmaf(r, scale, -y), 0.5f / (1.f - m[0]), shift);
  const float v[3] = { roundf(y), roundf(cb), roundf(cr) };
  const float sat = depth == 8 ? 255.f : 65535.f;                 /* `as u8` / `as u16` saturate at the type only */
  for (int i = 0; i < 3; i++) out[i] = (uint16_t)(v[i] < 0.f ? 0.f : (v[i] > sat ? sat : v[i]));
}

/* ------------------------------------------------------------------ container */
typedef struct { uint8_t *b; size_t n, cap; } Buf;
static void bput(Buf *o, const void *d, size_t n) {
  if (o->n + n > o->cap) { o->cap = (o->n + n) * 2 + 64; o->b = (uint8_t *)realloc(o->b, o->cap); }
  memcpy(o->b + o->n, d, n); o->n += n;
}
static void b8(Buf *o, unsigned v) { uint8_t x = (uint8_t)v; bput(o, &x, 1); }
static void b16(Buf *o, unsigned v) { uint8_t x[2] = { (uint8_t)(v >> 8), (uint8_t)v }; bput(o, x, 2); }
static void b32(Buf *o, uint32_t v) { uint8_t x[4] = { (uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v }; bput(o, x, 4); }
static size_t box_begin(Buf *o, const char *type) { size_t at = o->n; b32(o, 0); bput(o, type, 4); return at; }
static size_t fullbox_begin(Buf *o, const char *type, int version, uint32_t flags) { size_t at = box_begin(o, type); b32(o, ((uint32_t)version << 24) | flags); return at; }
static void box_end(Buf *o, size_t at) { uint32_t sz = (uint32_t)(o->n - at); o->b[at] = (uint8_t)(sz >> 24); o->b[at + 1] = (uint8_t)(sz >> 16); o->b[at + 2] = (uint8_t)(sz >> 8); o->b[at + 3] = (uint8_t)sz; }
static void av1c(Buf *o, int profile, int depth, int mono) {
  size_t a = box_begin(o, "av1C");
  b8(o, 0x81); b8(o, (unsigned)(profile << 5) | 31);
  b8(o, (unsigned)((depth > 8) << 6) | (unsigned)(mono << 4) | (unsigned)(mono ? 0x0C : 0));  /* tier 0, hbd, !12bit, mono, ssx, ssy, csp 0 */
  b8(o, 0);
  box_end(o, a);
}

size_t av1o_avif_container(const uint8_t *color, size_t color_len, const uint8_t *alpha, size_t alpha_len,
                           int w, int h, int depth, int mono_color, int cp, int tc, int mc, int full_range,
                           int premultiplied, uint8_t **out) {
  Buf o = { 0, 0, 0 };
  const int has_alpha = alpha && alpha_len;
  size_t a = box_begin(&o, "ftyp"); bput(&o, "avif", 4); b32(&o, 0); bput(&o, "avif", 4); bput(&o, "mif1", 4); bput(&o, "miaf", 4); box_end(&o, a);
  size_t meta = fullbox_begin(&o, "meta", 0, 0);
  a = fullbox_begin(&o, "hdlr", 0, 0); b32(&o, 0); bput(&o, "pict", 4); b32(&o, 0); b32(&o, 0); b32(&o, 0); b8(&o, 0); box_end(&o, a);
  a = fullbox_begin(&o, "pitm", 0, 0); b16(&o, 1); box_end(&o, a);
  a = fullbox_begin(&o, "iloc", 0, 0); b8(&o, 0x44); b8(&o, 0x00); b16(&o, has_alpha ? 2 : 1);
  size_t off_pos[2] = { 0, 0 };
  for (int i = 0; i < 1 + has_alpha; i++) { b16(&o, (unsigned)(i + 1)); b16(&o, 0); b16(&o, 1); off_pos[i] = o.n; b32(&o, 0); b32(&o, (uint32_t)(i ? alpha_len : color_len)); }
  box_end(&o, a);
  a = fullbox_begin(&o, "iinf", 0, 0); b16(&o, has_alpha ? 2 : 1);
  for (int i = 0; i < 1 + has_alpha; i++) { size_t e = fullbox_begin(&o, "infe", 2, 0); b16(&o, (unsigned)(i + 1)); b16(&o, 0); bput(&o, "av01", 4); b8(&o, 0); box_end(&o, e); }
  box_end(&o, a);
  if (has_alpha) {
    a = fullbox_begin(&o, "iref", 0, 0);
    size_t e = box_begin(&o, "auxl"); b16(&o, 2); b16(&o, 1); b16(&o, 1); box_end(&o, e);
    if (premultiplied) { e = box_begin(&o, "prem"); b16(&o, 1); b16(&o, 1); b16(&o, 2); box_end(&o, e); }
    box_end(&o, a);
  }
  a = box_begin(&o, "iprp");
  size_t ipco = box_begin(&o, "ipco");
  size_t e = fullbox_begin(&o, "ispe", 0, 0); b32(&o, (uint32_t)w); b32(&o, (uint32_t)h); box_end(&o, e);                 /* 1 */
  e = fullbox_begin(&o, "pixi", 0, 0); if (mono_color) { b8(&o, 1); b8(&o, (unsigned)depth); } else { b8(&o, 3); b8(&o, (unsigned)depth); b8(&o, (unsigned)depth); b8(&o, (unsigned)depth); } box_end(&o, e); /* 2 */
  av1c(&o, mono_color ? 0 : 1, depth, mono_color);                                                                      /* 3 */
  e = box_begin(&o, "colr"); bput(&o, "nclx", 4); b16(&o, (unsigned)cp); b16(&o, (unsigned)tc); b16(&o, (unsigned)mc); b8(&o, full_range ? 0x80 : 0); box_end(&o, e); /* 4 */
  if (has_alpha) {
    av1c(&o, 0, depth, 1);                                                                                               /* 5 */
    e = fullbox_begin(&o, "auxC", 0, 0); { const char urn[] = "urn:mpeg:mpegB:cicp:systems:auxiliary:alpha"; bput(&o, urn, sizeof(urn)); } box_end(&o, e); /* 6 */
    e = fullbox_begin(&o, "pixi", 0, 0); b8(&o, 1); b8(&o, (unsigned)depth); box_end(&o, e);                              /* 7 */
  }
  box_end(&o, ipco);
  e = fullbox_begin(&o, "ipma", 0, 0); b32(&o, has_alpha ? 2 : 1);
  b16(&o, 1); b8(&o, 4); b8(&o, 1); b8(&o, 2); b8(&o, 0x80 | 3); b8(&o, 4);
  if (has_alpha) { b16(&o, 2); b8(&o, 4); b8(&o, 1); b8(&o, 7); b8(&o, 0x80 | 5); b8(&o, 6); }
  box_end(&o, e);
  box_end(&o, a);
  box_end(&o, meta);
  a = box_begin(&o, "mdat");
  uint32_t off = (uint32_t)o.n;
  o.b[off_pos[0]] = (uint8_t)(off >> 24); o.b[off_pos[0] + 1] = (uint8_t)(off >> 16); o.b[off_pos[0] + 2] = (uint8_t)(off >> 8); o.b[off_pos[0] + 3] = (uint8_t)off;
  bput(&o, color, color_len);
  if (has_alpha) {
    off = (uint32_t)o.n;
    o.b[off_pos[1]] = (uint8_t)(off >> 24); o.b[off_pos[1] + 1] = (uint8_t)(off >> 16); o.b[off_pos[1] + 2] = (uint8_t)(off >> 8); o.b[off_pos[1] + 3] = (uint8_t)off;
    bput(&o, alpha, alpha_len);
  }
  box_end(&o, a);
  *out = o.b;
  return o.n;
}

/* ------------------------------------------------------------------ Encoder::encode_rgba / encode_rgb */
static int encode_planes(const RavifOracleEncoder *e, int w, int h, uint16_t *pl[3], const uint16_t *alpha_plane, int depth, int matrix, RavifOracleImage *out) {
  const int quantizer = av1o_quality_to_quantizer(e->quality), aquant = av1o_quality_to_quantizer(e->alpha_quality);
  Av1oConfig c; memset(&c, 0, sizeof(c));
  c.width = w; c.height = h; c.bit_depth = depth; c.mono = 0; c.quantizer = quantizer; c.full_range = 1;
  c.has_color_desc = 1; c.color_primaries = 1; c.transfer = 13; c.matrix = matrix; c.threads = e->threads; c.tiles_override = e->tiles_override; c.rdo_passes = e->rdo_passes;
  if (av1o_tweaks_from_preset(e->speed, quantizer, &c)) return 4;
  Av1oResult rc, ra; memset(&ra, 0, sizeof(ra));
  const uint16_t *planes[3] = { pl[0], pl[1], pl[2] }; int strides[3] = { w, w, w };
  int st = av1o_encode(&c, planes, strides, &rc);
  if (st) return st;
  if (alpha_plane) {
    Av1oConfig a; memset(&a, 0, sizeof(a));
    a.width = w; a.height = h; a.bit_depth = depth; a.mono = 1; a.quantizer = aquant; a.full_range = 1; a.has_color_desc = 0;
    a.threads = e->threads; a.tiles_override = e->tiles_override; a.rdo_passes = e->rdo_passes;
    av1o_tweaks_from_preset(e->speed, aquant, &a);
    const uint16_t *ap[3] = { alpha_plane, 0, 0 };
    st = av1o_encode(&a, ap, strides, &ra);
    if (st) { av1o_free_result(&rc); return st; }
  }
  out->avif_len = av1o_avif_container(rc.obu, rc.obu_len, ra.obu, ra.obu_len, w, h, depth, 0, 1, 13, matrix, 1, e->alpha_mode == 2, &out->avif);
  out->color_byte_size = rc.obu_len; out->alpha_byte_size = ra.obu_len;
  av1o_free_result(&rc); av1o_free_result(&ra);
  return 0;
}

static int encode_common(const RavifOracleEncoder *e, const uint8_t *px, int bpp, int w, int h, int stride_px, RavifOracleImage *out) {
  if (!e || !px || w < 1 || h < 1) return 4;
  uint8_t *cleaned = NULL;
  if (bpp == 4 && e->alpha_mode == 2) {                       /* convert_alpha_8bit: Premultiplied (:282-296), as written: a == 0 or
                                                                 a == 255 -> RGBA8::default(); else (c * 255 / a) as u8 (wraps) */
    cleaned = (uint8_t *)malloc((size_t)w * h * 4);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
      const uint8_t *p = px + ((size_t)y * stride_px + x) * 4; uint8_t *o = cleaned + ((size_t)y * w + x) * 4;
      if (p[3] == 0 || p[3] == 255) { o[0] = o[1] = o[2] = o[3] = 0; }
      else { for (int k = 0; k < 3; k++) o[k] = (uint8_t)((unsigned)p[k] * 255u / p[3]); o[3] = p[3]; }
    }
    px = cleaned; stride_px = w;
  }
  if (bpp == 4 && e->alpha_mode == 1) {                       /* convert_alpha_8bit: UnassociatedClean -> blurred_dirty_alpha (:277-281) */
    cleaned = (uint8_t *)malloc((size_t)w * h * 4);
    if (av1o_blurred_dirty_alpha(px, w, h, stride_px, cleaned)) { px = cleaned; stride_px = w; } else { free(cleaned); cleaned = NULL; }
  }
  const int depth = e->depth == 8 ? 8 : 10;                   /* BitDepth::Auto == Ten (:266,:339) */
  const size_t n = (size_t)w * h;
  uint16_t *pl[3]; for (int i = 0; i < 3; i++) pl[i] = (uint16_t *)malloc(n * 2);
  uint16_t *al = NULL; int use_alpha = 0;
  if (bpp == 4) for (int y = 0; y < h && !use_alpha; y++) for (int x = 0; x < w; x++) if (px[((size_t)y * stride_px + x) * 4 + 3] != 255) { use_alpha = 1; break; }
  if (use_alpha) al = (uint16_t *)malloc(n * 2);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const uint8_t *p = px + ((size_t)y * stride_px + x) * (size_t)bpp; size_t i = (size_t)y * w + x;
    if (e->color_model == 1) {                                 /* ColorModel::RGB -> planes G,B,R, Identity matrix */
      if (depth == 8) { pl[0][i] = p[1]; pl[1][i] = p[2]; pl[2][i] = p[0]; }
      else { pl[0][i] = av1o_to_ten(p[1]); pl[1][i] = av1o_to_ten(p[2]); pl[2][i] = av1o_to_ten(p[0]); }
    } else {
      uint16_t o3[3]; av1o_rgb_to_ycbcr(p, depth, o3); pl[0][i] = o3[0]; pl[1][i] = o3[1]; pl[2][i] = o3[2];
    }
    if (al) al[i] = depth == 8 ? p[3] : av1o_to_ten(p[3]);
  }
  int st = encode_planes(e, w, h, pl, al, depth, e->color_model == 1 ? 0 : 6, out);
  for (int i = 0; i < 3; i++) free(pl[i]);
  free(al);
  free(cleaned);
  return st;
}
int ravif_oracle_encode_rgba(const RavifOracleEncoder *e, const uint8_t *rgba, int w, int h, int stride_px, RavifOracleImage *out) { return encode_common(e, rgba, 4, w, h, stride_px, out); }
int ravif_oracle_encode_rgb(const RavifOracleEncoder *e, const uint8_t *rgb, int w, int h, int stride_px, RavifOracleImage *out) { return encode_common(e, rgb, 3, w, h, stride_px, out); }

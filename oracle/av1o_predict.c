/* oracle/av1o_predict.c -- AV1 intra prediction, spec section 7.11.2 (normative), square blocks, 4:4:4 / 4:0:0.
 * TEST INFRASTRUCTURE (see av1o.h).  rav1e equivalent: src/predict.rs (absent from /root/reference). */
#include "av1o_int.h"

static const uint8_t sm_w4[4] = { 255, 149, 85, 64 };
static const uint8_t sm_w8[8] = { 255, 197, 146, 105, 73, 50, 37, 32 };
static const uint8_t sm_w16[16] = { 255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16 };
static const uint8_t sm_w32[32] = { 255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74,
                                    66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8 };
static const uint8_t sm_w64[64] = { 255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150,
                                    144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
                                    65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20,
                                    18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4 };
static const uint8_t *sm_weights(int log2w) {
  switch (log2w) { case 2: return sm_w4; case 3: return sm_w8; case 4: return sm_w16; case 5: return sm_w32; default: return sm_w64; }
}
/* Dr_Intra_Derivative indexed by angle (only the listed angles are ever used) */
static int dr_deriv(int a) {
  switch (a) {
    case 3: return 1023; case 6: return 547; case 9: return 372; case 14: return 273; case 17: return 215;
    case 20: return 178; case 23: return 151; case 26: return 132; case 29: return 116; case 32: return 102;
    case 36: return 90; case 39: return 80; case 42: return 71; case 45: return 64; case 48: return 57;
    case 51: return 51; case 54: return 45; case 58: return 40; case 61: return 35; case 64: return 31;
    case 67: return 27; case 70: return 23; case 73: return 19; case 76: return 15; case 81: return 11;
    case 84: return 7; case 87: return 3; default: return 0;
  }
}
static const int mode_to_angle[9] = { 0, 90, 180, 45, 135, 113, 157, 203, 67 };

static int edge_filter_strength(int w, int h, int filter_type, int delta) {   /* 7.11.2.9 */
  int d = iabs(delta), blk = w + h, s = 0;
  if (filter_type == 0) {
    if (blk <= 8) { if (d >= 56) s = 1; }
    else if (blk <= 12) { if (d >= 40) s = 1; }
    else if (blk <= 16) { if (d >= 40) s = 1; }
    else if (blk <= 24) { if (d >= 8) s = 1; if (d >= 16) s = 2; if (d >= 32) s = 3; }
    else if (blk <= 32) { if (d >= 1) s = 1; if (d >= 4) s = 2; if (d >= 32) s = 3; }
    else { if (d >= 1) s = 3; }
  } else {
    if (blk <= 8) { if (d >= 40) s = 1; if (d >= 64) s = 2; }
    else if (blk <= 16) { if (d >= 20) s = 1; if (d >= 48) s = 2; }
    else if (blk <= 24) { if (d >= 4) s = 3; }
    else { if (d >= 1) s = 3; }
  }
  return s;
}
static int edge_upsample(int w, int h, int filter_type, int delta) {            /* 7.11.2.10 */
  int d = iabs(delta), blk = w + h;
  if (d <= 0 || d >= 40) return 0;
  return filter_type == 0 ? (blk <= 16) : (blk <= 8);
}
static void edge_filter(uint16_t *buf /* buf[-1] valid */, int sz, int strength) { /* 7.11.2.12 */
  static const int kern[3][5] = { { 0, 4, 8, 4, 0 }, { 0, 5, 6, 5, 0 }, { 2, 4, 4, 4, 2 } };
  if (!strength) return;
  uint16_t edge[2 * 64 + 2];
  for (int i = 0; i < sz; i++) edge[i] = buf[i - 1];
  for (int i = 1; i < sz; i++) {
    int s = 0;
    for (int j = 0; j < 5; j++) { int k = iclamp(i - 2 + j, 0, sz - 1); s += kern[strength - 1][j] * edge[k]; }
    buf[i - 1] = (uint16_t)((s + 8) >> 4);
  }
}
static void edge_upsample_do(uint16_t *buf, int num_px, int bd) {                 /* 7.11.2.11 */
  int dup[64 + 3];
  dup[0] = buf[-1];
  for (int i = -1; i < num_px; i++) dup[i + 2] = buf[i];
  dup[num_px + 2] = buf[num_px - 1];
  buf[-2] = (uint16_t)dup[0];
  int mx = (1 << bd) - 1;
  for (int i = 0; i < num_px; i++) {
    int s = -dup[i] + 9 * dup[i + 1] + 9 * dup[i + 2] - dup[i + 3];
    s = iclamp(round2(s, 4), 0, mx);
    buf[2 * i - 1] = (uint16_t)s;
    buf[2 * i] = (uint16_t)dup[i + 2];
  }
}

void av1o_predict_intra(const Av1oFrame *f, const TileB *t, int plane, int x, int y, int log2w,
                        int have_left, int have_above, int have_above_rt, int have_below_lft,
                        int mode, int angle_delta, int filter_type, uint16_t *dst, int ds) {
  av1o_predict_intra_wh(f, t, plane, x, y, log2w, log2w, have_left, have_above, have_above_rt, have_below_lft, mode, angle_delta, filter_type, dst, ds);
}
/* w x h block (a transform block of the block being predicted): the above-right run is as long as the block is WIDE and the
 * below-left run as long as it is HIGH (libaom / dav1d edge preparation: n_topright_px <= txw, n_bottomleft_px <= txh), the rest
 * of the w + h samples an angle may reach repeats the last one. */
void av1o_predict_intra_wh(const Av1oFrame *f, const TileB *t, int plane, int x, int y, int log2w, int log2h,
                           int have_left, int have_above, int have_above_rt, int have_below_lft,
                           int mode, int angle_delta, int filter_type, uint16_t *dst, int ds) {
  (void)t;
  const int w = 1 << log2w, h = 1 << log2h, bd = f->bd;
  const int max_x = f->mi_cols * MI - 1, max_y = f->mi_rows * MI - 1;
  const uint16_t *rec = f->rec[plane]; const int rs = f->stride;
  uint16_t above_buf[16 + 2 * 64 + 64 + 16], left_buf[16 + 2 * 64 + 64 + 16];
  uint16_t *above = above_buf + 16, *left = left_buf + 16;
  const int n = w + h;
  /* 7.11.2 edge preparation */
  if (!have_above && have_left) { for (int i = 0; i < n; i++) above[i] = rec[y * rs + x - 1]; }
  else if (!have_above && !have_left) { for (int i = 0; i < n; i++) above[i] = (uint16_t)((1 << (bd - 1)) - 1); }
  else {
    int lim = imin(max_x, x + (have_above_rt ? 2 * w : w) - 1);
    for (int i = 0; i < n; i++) above[i] = rec[(y - 1) * rs + imin(lim, x + i)];
  }
  if (!have_left && have_above) { for (int i = 0; i < n; i++) left[i] = rec[(y - 1) * rs + x]; }
  else if (!have_left && !have_above) { for (int i = 0; i < n; i++) left[i] = (uint16_t)((1 << (bd - 1)) + 1); }
  else {
    int lim = imin(max_y, y + (have_below_lft ? 2 * h : h) - 1);
    for (int i = 0; i < n; i++) left[i] = rec[imin(lim, y + i) * rs + x - 1];
  }
  if (have_above && have_left) above[-1] = rec[(y - 1) * rs + x - 1];
  else if (have_above) above[-1] = rec[(y - 1) * rs + x];
  else if (have_left) above[-1] = rec[y * rs + x - 1];
  else above[-1] = (uint16_t)(1 << (bd - 1));
  left[-1] = above[-1];

  if (mode == PAETH_PRED) {
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
      int base = above[j] + left[i] - above[-1];
      int pl = iabs(base - left[i]), pt = iabs(base - above[j]), ptl = iabs(base - above[-1]);
      dst[i * ds + j] = (pl <= pt && pl <= ptl) ? left[i] : (pt <= ptl ? above[j] : above[-1]);
    }
  } else if (mode == DC_PRED) {
    int v;
    if (have_left && have_above) { int s = 0; for (int k = 0; k < w; k++) s += above[k]; for (int k = 0; k < h; k++) s += left[k]; v = (s + ((w + h) >> 1)) / (w + h); }
    else if (have_left) { int s = 0; for (int k = 0; k < h; k++) s += left[k]; v = (s + (h >> 1)) >> log2h; }
    else if (have_above) { int s = 0; for (int k = 0; k < w; k++) s += above[k]; v = (s + (w >> 1)) >> log2w; }
    else v = 1 << (bd - 1);
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) dst[i * ds + j] = (uint16_t)v;
  } else if (mode == SMOOTH_PRED) {
    const uint8_t *sw = sm_weights(log2w), *sh = sm_weights(log2h);
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
      int p = sh[i] * above[j] + (256 - sh[i]) * left[h - 1] + sw[j] * left[i] + (256 - sw[j]) * above[w - 1];
      dst[i * ds + j] = (uint16_t)round2(p, 9);
    }
  } else if (mode == SMOOTH_V_PRED) {
    const uint8_t *sw = sm_weights(log2h);
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++)
      dst[i * ds + j] = (uint16_t)round2(sw[i] * above[j] + (256 - sw[i]) * left[h - 1], 8);
  } else if (mode == SMOOTH_H_PRED) {
    const uint8_t *sw = sm_weights(log2w);
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++)
      dst[i * ds + j] = (uint16_t)round2(sw[j] * left[i] + (256 - sw[j]) * above[w - 1], 8);
  } else {
    /* directional, 7.11.2.4 */
    int p_angle = mode_to_angle[mode] + angle_delta * 3;
    int up_above = 0, up_left = 0;
    /* enable_intra_edge_filter == 1 always in this encoder */
    if (p_angle != 90 && p_angle != 180) {
      if (p_angle > 90 && p_angle < 180 && (w + h) >= 24) {
        int v = round2(left[0] * 5 + above[-1] * 6 + above[0] * 5, 4);
        above[-1] = left[-1] = (uint16_t)v;
      }
      if (have_above) {
        int st = edge_filter_strength(w, h, filter_type, p_angle - 90);
        int num = imin(w, max_x - x + 1) + (p_angle < 90 ? h : 0) + 1;
        edge_filter(above, num, st);
      }
      if (have_left) {
        int st = edge_filter_strength(w, h, filter_type, p_angle - 180);
        int num = imin(h, max_y - y + 1) + (p_angle > 180 ? w : 0) + 1;
        edge_filter(left, num, st);
      }
    }
    up_above = edge_upsample(w, h, filter_type, p_angle - 90);
    if (up_above) edge_upsample_do(above, w + (p_angle < 90 ? h : 0), bd);
    up_left = edge_upsample(w, h, filter_type, p_angle - 180);
    if (up_left) edge_upsample_do(left, h + (p_angle > 180 ? w : 0), bd);
    int dx = 0, dy = 0;
    if (p_angle < 90) dx = dr_deriv(p_angle); else if (p_angle > 90 && p_angle < 180) dx = dr_deriv(180 - p_angle);
    if (p_angle > 90 && p_angle < 180) dy = dr_deriv(p_angle - 90); else if (p_angle > 180) dy = dr_deriv(270 - p_angle);
    if (p_angle < 90) {
      int max_base = (w + h - 1) << up_above;
      for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
        int idx = (i + 1) * dx, base = (idx >> (6 - up_above)) + (j << up_above);
        int sh = ((idx << up_above) >> 1) & 0x1F;
        dst[i * ds + j] = base < max_base ? (uint16_t)round2(above[base] * (32 - sh) + above[base + 1] * sh, 5) : above[max_base];
      }
    } else if (p_angle > 90 && p_angle < 180) {
      for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
        int idx = (j << 6) - (i + 1) * dx, base = idx >> (6 - up_above);
        if (base >= -(1 << up_above)) {
          int sh = ((idx << up_above) >> 1) & 0x1F;
          dst[i * ds + j] = (uint16_t)round2(above[base] * (32 - sh) + above[base + 1] * sh, 5);
        } else {
          idx = (i << 6) - (j + 1) * dy; base = idx >> (6 - up_left);
          int sh = ((idx << up_left) >> 1) & 0x1F;
          dst[i * ds + j] = (uint16_t)round2(left[base] * (32 - sh) + left[base + 1] * sh, 5);
        }
      }
    } else if (p_angle > 180) {
      for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
        int idx = (j + 1) * dy, base = (idx >> (6 - up_left)) + (i << up_left);
        int sh = ((idx << up_left) >> 1) & 0x1F;
        dst[i * ds + j] = (uint16_t)round2(left[base] * (32 - sh) + left[base + 1] * sh, 5);
      }
    } else if (p_angle == 90) {
      for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) dst[i * ds + j] = above[j];
    } else {
      for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) dst[i * ds + j] = left[i];
    }
  }
}

/* 7.11.5 chroma-from-luma for 4:4:4: dst holds the DC prediction on entry. */
void av1o_predict_cfl(const Av1oFrame *f, int plane, int x, int y, int log2w, int alpha, uint16_t *dst, int ds) {
  av1o_predict_cfl_wh(f, plane, x, y, log2w, log2w, alpha, dst, ds);
}
void av1o_predict_cfl_wh(const Av1oFrame *f, int plane, int x, int y, int log2w, int log2h, int alpha, uint16_t *dst, int ds) {
  (void)plane;
  const int w = 1 << log2w, h = 1 << log2h, mx = (1 << f->bd) - 1;
  const uint16_t *luma = f->rec[0]; const int rs = f->stride;
  int sum = 0;
  for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) sum += luma[(y + i) * rs + x + j] << 3;
  int avg = round2(sum, log2w + log2h);
  for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
    int l = (luma[(y + i) * rs + x + j] << 3) - avg;
    int v = alpha * l;
    int scaled = v >= 0 ? round2(v, 6) : -round2(-v, 6);
    dst[i * ds + j] = (uint16_t)iclamp(dst[i * ds + j] + scaled, 0, mx);
  }
}

/* oracle/dirtyalpha.c -- restatement of ravif/src/dirtyalpha.rs:5-124 (blurred_dirty_alpha, bleed_opaque_color,
 * blur_transparent_pixels, premultiplied_minmax).  TEST INFRASTRUCTURE (see av1o.h).
 * Neighbourhoods follow loop9's convention: the 3x3 window is clamped to the image (edge pixels replicated). */
#include "av1o_int.h"

void av1o_premultiplied_minmax(uint8_t px, uint8_t alpha, uint8_t *lo, uint8_t *hi) {   /* dirtyalpha.rs:115-124 */
  const uint16_t a = alpha;
  const uint16_t rounded = (uint16_t)((uint16_t)((uint16_t)px * a) / 255 * 255);
  const uint8_t low = (uint8_t)((uint16_t)(rounded + 16) / a);      /* `as u8` wraps */
  const uint8_t high = (uint8_t)((uint16_t)(rounded + 239) / a);
  *lo = low < px ? low : px; *hi = high > px ? high : px;
}
static inline const uint8_t *px_at(const uint8_t *img, int w, int h, int stride_px, int x, int y) {
  x = x < 0 ? 0 : (x >= w ? w - 1 : x); y = y < 0 ? 0 : (y >= h ? h - 1 : y);
  return img + ((size_t)y * stride_px + x) * 4;
}
static inline uint8_t clamp8(uint8_t v, uint8_t lo, uint8_t hi) { v = v > lo ? v : lo; return v < hi ? v : hi; }

/* returns 1 and fills out (w*h*4, tightly packed) when the image was changed, 0 when blurred_dirty_alpha returns None */
int av1o_blurred_dirty_alpha(const uint8_t *rgba, int w, int h, int stride_px, uint8_t *out) {
  uint64_t sum[3] = { 0, 0, 0 }, weights = 0;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const uint8_t *m = px_at(rgba, w, h, stride_px, x, y);
    if (m[3] == 255 || m[3] == 0) continue;
    int any0 = 0;
    for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) any0 |= px_at(rgba, w, h, stride_px, x + dx, y + dy)[3] == 0;
    if (any0) { const uint32_t wt = 256u - m[3]; weights += wt; for (int c = 0; c < 3; c++) sum[c] += (uint64_t)m[c] * wt; }
  }
  if (weights == 0) return 0;
  const uint8_t bg[4] = { (uint8_t)(sum[0] / weights), (uint8_t)(sum[1] / weights), (uint8_t)(sum[2] / weights), 0 };
  uint8_t *tmp = (uint8_t *)malloc((size_t)w * h * 4);
  /* bleed_opaque_color */
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const uint8_t *m = px_at(rgba, w, h, stride_px, x, y); uint8_t *o = tmp + ((size_t)y * w + x) * 4;
    if (m[3] == 255) { memcpy(o, m, 4); continue; }
    uint32_t wsum = 0, s[3] = { 0, 0, 0 };
    for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
      const uint8_t *c = px_at(rgba, w, h, stride_px, x + dx, y + dy);
      if (c[3] == 0) continue;
      const uint32_t wt = 256u - c[3]; wsum += wt; for (int k = 0; k < 3; k++) s[k] += (uint32_t)c[k] * wt;
    }
    if (wsum == 0) { memcpy(o, bg, 4); continue; }
    uint8_t avg[3]; for (int k = 0; k < 3; k++) avg[k] = (uint8_t)(s[k] / wsum);
    if (m[3] == 0) { o[0] = avg[0]; o[1] = avg[1]; o[2] = avg[2]; o[3] = 0; }
    else { for (int k = 0; k < 3; k++) { uint8_t lo, hi; av1o_premultiplied_minmax(m[k], m[3], &lo, &hi); o[k] = clamp8(avg[k], lo, hi); } o[3] = m[3]; }
  }
  /* blur_transparent_pixels */
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const uint8_t *m = px_at(tmp, w, h, w, x, y); uint8_t *o = out + ((size_t)y * w + x) * 4;
    if (m[3] == 255) { memcpy(o, m, 4); continue; }
    uint16_t s[3] = { 0, 0, 0 };
    for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) { const uint8_t *c = px_at(tmp, w, h, w, x + dx, y + dy); for (int k = 0; k < 3; k++) s[k] = (uint16_t)(s[k] + c[k]); }
    uint8_t avg[3]; for (int k = 0; k < 3; k++) avg[k] = (uint8_t)(s[k] / 9);
    if (m[3] == 0) { o[0] = avg[0]; o[1] = avg[1]; o[2] = avg[2]; o[3] = 0; }
    else { for (int k = 0; k < 3; k++) { uint8_t lo, hi; av1o_premultiplied_minmax(m[k], m[3], &lo, &hi); o[k] = clamp8(avg[k], lo, hi); } o[3] = m[3]; }
  }
  free(tmp);
  return 1;
}

/* oracle/av1o_segment.c -- segmentation: per-segment quantiser offsets fitted to the frame's activity scales, and the segment of a block.
 * TEST INFRASTRUCTURE (see av1o.h).
 * Follows rav1e segmentation.rs (absent from /root/reference) as recalled [UPSTREAM-RECALL]:
 *   segmentation_optimize_inner: k-means (k = 8 .. 3) over the sorted log2 of the per-8x8 scales, the k whose centroids are most evenly
 *     spaced wins (last minimum = fewest segments), each centroid's scale s gets the quantiser index whose step is nearest base / sqrt(s)
 *     ("scale * Q'^2 = Q^2"), never below index 1 (no lossless segment); thresholds half way between centroids;
 *   SegmentationLevel::Simple (every speed preset but 0): a block's segment is looked up from its mean scale, no RD search over segments;
 *   segment 0 = the largest scale (finest quantiser).
 * Integer arithmetic throughout (the HIP path runs the same steps on the device): logs are floor-type Q11 values from repeated squaring,
 * bucketed to 1/256 octave over [2^6, 2^22) -- a 4096-bin histogram carries everything k-means needs. */
#include "av1o_int.h"
#include <stdio.h>

int av1o_ilog2_q11(uint32_t x) {
  if (x == 0) x = 1;
  int msb = 31; while (!(x >> msb)) msb--;
  uint64_t m = (uint64_t)x << (31 - msb);              /* mantissa in [2^31, 2^32) */
  int frac = 0;
  for (int i = 0; i < 11; i++) { m = (m * m) >> 31; frac <<= 1; if (m >> 32) { frac |= 1; m >>= 1; } }
  return (msb << 11) | frac;
}
int av1o_seg_bucket(uint32_t scale_q14) { return iclamp((av1o_ilog2_q11(scale_q14) - (6 << 11)) >> 3, 0, AV1O_SEG_BINS - 1); }

/* 1-D k-means over the histogram: cnt[b] / wsum[b] = number / sum of the values below b (prefix arrays, AV1O_SEG_BINS + 1 entries) */
static int value_at(const uint32_t *cnt, uint32_t idx) {       /* the idx-th smallest value */
  int lo = 0, hi = AV1O_SEG_BINS - 1;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (cnt[mid + 1] > idx) hi = mid; else lo = mid + 1; }
  return lo;
}
static void kmeans(const uint32_t *cnt, const uint64_t *wsum, int k, int *c) {
  const uint32_t n = cnt[AV1O_SEG_BINS];
  for (int j = 0; j < k; j++) c[j] = value_at(cnt, (uint32_t)(((uint64_t)j * (n - 1)) / (uint64_t)(k - 1)));
  int limit = 0; while ((n >> limit) != 0) limit++;
  limit *= 2;
  for (int it = 0; it < limit; it++) {
    int changed = 0, lo = 0, t[8];
    for (int j = 0; j + 1 < k; j++) t[j] = (c[j] + c[j + 1] + 1) >> 1;       /* boundaries from this round's starting centroids */
    for (int j = 0; j < k; j++) {
      const int hi = j == k - 1 ? AV1O_SEG_BINS : imax(lo, t[j]);              /* cluster j = the values in [lo, hi) */
      const uint32_t m = cnt[hi] - cnt[lo];
      if (m) { const int nc = (int)((wsum[hi] - wsum[lo] + m / 2) / m); if (nc != c[j]) changed = 1; c[j] = nc; }
      lo = hi;
    }
    if (!changed) break;
  }
}

/* The quantiser index of a segment whose scale lies `dev` (1/256 octave) above the frame's mean scale: the index (>= 1) whose AC step is nearest, in the log
 * domain and with the lower index on ties, to base_step / sqrt(scale / mean).  Exported for tests/test_independent_checks.py (checked against a float restatement). */
int av1o_seg_qidx_for_dev(int base_q_idx, int bd, int dev) {
  const int16_t *ac = bd == 8 ? av1_ac_q8 : av1_ac_q10;
  const int target = av1o_ilog2_q11((uint32_t)ac[base_q_idx]) - dev * 4;    /* Q11: minus half the scale's log */
  int qi = 1, best = 1 << 30;
  for (int q = 1; q < 256; q++) { const int d = iabs(av1o_ilog2_q11((uint32_t)ac[q]) - target); if (d < best) { best = d; qi = q; } }
  return qi;
}
int av1o_ac_step(int bd, int qidx) { return (bd == 8 ? av1_ac_q8 : av1_ac_q10)[iclamp(qidx, 0, 255)]; }

void av1o_segmentation(Av1oFrame *f) {
  f->seg_n = 0;
  for (int i = 0; i < 8; i++) { f->seg_qidx[i] = f->base_q_idx; for (int p = 0; p < 3; p++) { f->seg_dcq[i][p] = f->dc_q[p]; f->seg_acq[i][p] = f->ac_q[p]; } }
  const int cw = f->pw / 8, vw = (f->w + 7) >> 3, vh = (f->h + 7) >> 3;
  uint32_t *cnt = (uint32_t *)calloc(AV1O_SEG_BINS + 1, sizeof(uint32_t));
  uint64_t *wsum = (uint64_t *)calloc(AV1O_SEG_BINS + 1, sizeof(uint64_t));
  for (int cy = 0; cy < vh; cy++) for (int cx = 0; cx < vw; cx++) cnt[av1o_seg_bucket(f->act[cy * cw + cx]) + 1]++;
  int bmin = -1, bmax = -1;
  for (int b = 0; b < AV1O_SEG_BINS; b++) if (cnt[b + 1]) { if (bmin < 0) bmin = b; bmax = b; }
  for (int b = 0; b < AV1O_SEG_BINS; b++) { wsum[b + 1] = wsum[b] + (uint64_t)b * cnt[b + 1]; cnt[b + 1] += cnt[b]; }
  const uint32_t n = cnt[AV1O_SEG_BINS];
  if (bmin == bmax || n < 2 || getenv("AV1O_NO_SEGMENTATION")) { free(cnt); free(wsum); return; }            /* one scale everywhere (flat image, Tune::Psnr): no segmentation; the environment switch: ablation runs of the tests only */
  const int mean = (int)((wsum[AV1O_SEG_BINS] + n / 2) / n);
  int best_k = 0, best_c[8]; uint64_t best_var = 0;
  for (int k = 8; k >= 3; k--) {
    int c[9]; kmeans(cnt, wsum, k, c);
    int64_t sum = 0;
    for (int j = 0; j + 1 < k; j++) sum += c[j + 1] - c[j];
    const int64_t mu = sum / (k - 1);
    uint64_t var = 0;
    for (int j = 0; j + 1 < k; j++) { const int64_t d = (c[j + 1] - c[j]) - mu; var += (uint64_t)(d * d); }
    if (!best_k || var <= best_var) { best_k = k; best_var = var; memcpy(best_c, c, sizeof(int) * (size_t)k); }
  }
  /* quantiser index per segment: nearest step (log domain, lower index on ties) to base / sqrt(scale / mean scale) */
  const int16_t *ac = f->bd == 8 ? av1_ac_q8 : av1_ac_q10, *dc = f->bd == 8 ? av1_dc_q8 : av1_dc_q10;
  f->seg_n = best_k; f->seg_mean = mean;
  for (int i = 0; i < best_k; i++) {
    const int qi = av1o_seg_qidx_for_dev(f->base_q_idx, f->bd, best_c[best_k - 1 - i] - mean);   /* deviation from the mean scale in 1/256 octave */
    f->seg_qidx[i] = qi;
    for (int p = 0; p < f->np; p++) {
      f->seg_dcq[i][p] = dc[iclamp(qi + f->dc_qi[p] - f->base_q_idx, 0, 255)];
      f->seg_acq[i][p] = ac[p == 0 ? qi : iclamp(qi + f->ac_qi[p] - f->base_q_idx, 0, 255)];
    }
  }
  for (int j = 0; j + 1 < best_k; j++) f->seg_thr[j] = (best_c[j] + best_c[j + 1] + 1) >> 1;
  if (getenv("AV1O_SEG_DEBUG")) { fprintf(stderr, "seg: n=%d mean=%d base=%d q:", f->seg_n, mean, f->base_q_idx); for (int i = 0; i < best_k; i++) fprintf(stderr, " %d", f->seg_qidx[i]); fprintf(stderr, " thr:"); for (int j = 0; j + 1 < best_k; j++) fprintf(stderr, " %d", f->seg_thr[j]); fprintf(stderr, "\n"); }
  free(cnt); free(wsum);
}

/* SegmentationLevel::Simple: the segment of the block at pixel (x, y), w x h, from its mean activity scale */
int av1o_block_segment(const Av1oFrame *f, int x, int y, int w, int h) {
  if (!f->seg_n) return 0;
  const int b = av1o_seg_bucket(av1o_act_mean(f, x, y, w, h));
  int a = 0;
  for (int j = 0; j + 1 < f->seg_n; j++) a += b >= f->seg_thr[j];
  return f->seg_n - 1 - a;
}

/* spec 5.11.9 read_segment_id: prediction and CDF context from the above-left / above / left neighbours (-1 = not available) */
int av1o_seg_pred(int prev_ul, int prev_u, int prev_l, int *ctx) {
  if (prev_ul < 0) *ctx = 0;
  else if (prev_ul == prev_u && prev_ul == prev_l) *ctx = 2;
  else if (prev_ul == prev_u || prev_ul == prev_l || prev_u == prev_l) *ctx = 1;
  else *ctx = 0;
  if (prev_u == -1) return prev_l == -1 ? 0 : prev_l;
  if (prev_l == -1) return prev_u;
  return prev_ul == prev_u ? prev_u : prev_l;
}
static int neg_deinterleave(int diff, int ref, int max) {
  if (!ref) return diff;
  if (ref >= max - 1) return max - diff - 1;
  if (2 * ref < max) {
    if (diff <= 2 * ref) return (diff & 1) ? ref + ((diff + 1) >> 1) : ref - (diff >> 1);
    return diff;
  }
  if (diff <= 2 * (max - ref - 1)) return (diff & 1) ? ref + ((diff + 1) >> 1) : ref - (diff >> 1);
  return max - (diff + 1);
}
/* the symbol that decodes to `seg` (neg_deinterleave is a bijection of 0 .. max - 1) */
int av1o_seg_symbol(int seg, int pred, int max) {
  for (int d = 0; d < max; d++) if (neg_deinterleave(d, pred, max) == seg) return d;
  return 0;
}

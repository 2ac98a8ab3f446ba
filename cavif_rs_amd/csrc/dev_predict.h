// dev_predict.h -- AV1 intra prediction (spec 7.11.2), one wavefront per block, edges and the
// predicted block staged in that wave's LDS region.  Square blocks, 4:4:4 / 4:0:0.
// Mirrors rav1e src/predict.rs in function (that file is not in /root/reference; the arithmetic is
// the normative AV1 prediction process).
#pragma once
#include "dev_common.h"

#define EDGE_OFF 16
#define EDGE_LEN(N) (EDGE_OFF + 2 * (N) + 16)

__device__ __forceinline__ const uint8_t *sm_weights_dev(int log2w) {
  static __device__ const uint8_t w4[4] = { 255, 149, 85, 64 };
  static __device__ const uint8_t w8[8] = { 255, 197, 146, 105, 73, 50, 37, 32 };
  static __device__ const uint8_t w16[16] = { 255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16 };
  static __device__ const uint8_t w32[32] = { 255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74,
                                              66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8 };
  static __device__ const uint8_t w64[64] = { 255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150,
                                              144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
                                              65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20,
                                              18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4 };
  switch (log2w) { case 2: return w4; case 3: return w8; case 4: return w16; case 5: return w32; default: return w64; }
}
__device__ __forceinline__ int dr_deriv_dev(int a) {
  switch (a) {
    case 3: return 1023; case 6: return 547; case 9: return 372; case 14: return 273; case 17: return 215;
    case 20: return 178; case 23: return 151; case 26: return 132; case 29: return 116; case 32: return 102;
    case 36: return 90; case 39: return 80; case 42: return 71; case 45: return 64; case 48: return 57;
    case 51: return 51; case 54: return 45; case 58: return 40; case 61: return 35; case 64: return 31;
    case 67: return 27; case 70: return 23; case 73: return 19; case 76: return 15; case 81: return 11;
    case 84: return 7; case 87: return 3; default: return 0;
  }
}
__device__ __forceinline__ int edge_strength_dev(int w, int h, int ft, int delta) {
  const int d = iabs_(delta), blk = w + h; int s = 0;
  if (ft == 0) {
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
__device__ __forceinline__ int edge_upsample_sel_dev(int w, int h, int ft, int delta) {
  const int d = iabs_(delta), blk = w + h;
  if (d <= 0 || d >= 40) return 0;
  return ft == 0 ? (blk <= 16) : (blk <= 8);
}

// Raw (unfiltered) edges of the block at pixel (x,y) of `plane`: above[-1..2n-1], left[-1..2n-1]. (spec 7.11.2 steps 1-4)
__device__ inline void load_edges(const LDS FrameDev *f, int plane, int x, int y, int n, int have_left, int have_above,
                                  int have_ar, int have_bl, LDS uint16_t *above /* +EDGE_OFF */, LDS uint16_t *left) {
  const int bd = f->bd, rs = f->stride;
  const uint16_t *rec = f->rec[plane];
  const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1;
  const int tot = 2 * n;
  // index tot is the corner; every sample comes from one unconditional load per edge (addresses selected, not the loads)
  const int na = x + (have_ar ? 2 * n : n) - 1, nl = y + (have_bl ? 2 * n : n) - 1;
  const int lim_a = imin_(max_x, na), lim_l = imin_(max_y, nl);
  for (int i = LANE; i <= tot; i += 64) {
    const bool corner = i == tot;
    int ia, il;
    if (have_above) ia = (y - 1) * rs + (corner ? (have_left ? x - 1 : x) : imin_(lim_a, x + i));
    else ia = y * rs + (have_left ? x - 1 : x);                  // no row above: the left neighbour's sample (or, with no neighbour at all, any valid address)
    if (have_left) il = imin_(lim_l, y + i) * rs + x - 1;
    else il = (have_above ? y - 1 : y) * rs + x;
    uint16_t a = rec[ia], l = rec[il];
    if (!have_above && !have_left) { a = (uint16_t)(corner ? (1 << (bd - 1)) : (1 << (bd - 1)) - 1); l = (uint16_t)((1 << (bd - 1)) + 1); }
    if (corner) { above[-1] = a; left[-1] = a; }
    else { above[i] = a; left[i] = l; }
  }
  WAVE_SYNC();
}

__device__ inline void edge_filter_dev(LDS uint16_t *buf, int sz, int strength, LDS uint16_t *tmp) {
  if (!strength) return;
  for (int i = LANE; i < sz; i += 64) tmp[i] = buf[i - 1];
  WAVE_SYNC();
  for (int i = 1 + LANE; i < sz; i += 64) {
    int s = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int k = iclamp_(i - 2 + j, 0, sz - 1);
      const int kw = strength == 1 ? (j == 0 || j == 4 ? 0 : (j == 2 ? 8 : 4)) : (strength == 2 ? (j == 0 || j == 4 ? 0 : (j == 2 ? 6 : 5)) : (j == 0 || j == 4 ? 2 : 4));
      s += kw * tmp[k];
    }
    buf[i - 1] = (uint16_t)((s + 8) >> 4);
  }
  WAVE_SYNC();
}
__device__ inline void edge_upsample_dev(LDS uint16_t *buf, int num_px, int bd, LDS uint16_t *tmp) {
  // dup[0] = buf[-1]; dup[i+2] = buf[i] (i=-1..num_px-1); dup[num_px+2] = buf[num_px-1]
  for (int i = LANE; i < num_px + 3; i += 64) {
    uint16_t v;
    if (i == 0) v = buf[-1]; else if (i == num_px + 2) v = buf[num_px - 1]; else v = buf[i - 2];
    tmp[i] = v;
  }
  WAVE_SYNC();
  const int mx = (1 << bd) - 1;
  if (LANE == 0) buf[-2] = tmp[0];
  for (int i = LANE; i < num_px; i += 64) {
    int s = -(int)tmp[i] + 9 * (int)tmp[i + 1] + 9 * (int)tmp[i + 2] - (int)tmp[i + 3];
    s = iclamp_(round2_(s, 4), 0, mx);
    buf[2 * i - 1] = (uint16_t)s;
    buf[2 * i] = tmp[i + 2];
  }
  WAVE_SYNC();
}

// Predict an n x n block into pred[n*n] from raw edges. wa/wl are working copies (modified by filters).
__device__ inline void predict_block(const LDS FrameDev *f, int x, int y, int log2w, int have_left, int have_above,
                                     int mode, int angle_delta, int ftype, const LDS uint16_t *ra, const LDS uint16_t *rl,
                                     LDS uint16_t *wa, LDS uint16_t *wl, LDS uint16_t *tmp, LDS uint16_t *pred) {
  const int n = 1 << log2w, bd = f->bd, nn = n * n;
  const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1;
  if (mode == PAETH_PRED) {
    const int tl = ra[-1];
    for (int idx = LANE; idx < nn; idx += 64) {
      const int i = idx >> log2w, j = idx & (n - 1);
      const int base = ra[j] + rl[i] - tl;
      const int pl = iabs_(base - rl[i]), pt = iabs_(base - ra[j]), ptl = iabs_(base - tl);
      pred[idx] = (pl <= pt && pl <= ptl) ? rl[i] : (pt <= ptl ? ra[j] : (uint16_t)tl);
    }
  } else if (mode == DC_PRED) {
    int v;
    if (have_left || have_above) {
      int s = 0;
      for (int k = LANE; k < n; k += 64) s += (have_above ? ra[k] : 0) + (have_left ? rl[k] : 0);
      s = wave_sum_i32(s);
      if (have_left && have_above) v = (s + n) / (2 * n);
      else v = (s + (n >> 1)) >> log2w;
    } else v = 1 << (bd - 1);
    for (int idx = LANE; idx < nn; idx += 64) pred[idx] = (uint16_t)v;
  } else if (mode == SMOOTH_PRED || mode == SMOOTH_V_PRED || mode == SMOOTH_H_PRED) {
    const uint8_t *sw = sm_weights_dev(log2w);
    const int bl = rl[n - 1], tr = ra[n - 1];
    for (int idx = LANE; idx < nn; idx += 64) {
      const int i = idx >> log2w, j = idx & (n - 1);
      int p;
      if (mode == SMOOTH_PRED) p = round2_(sw[i] * ra[j] + (256 - sw[i]) * bl + sw[j] * rl[i] + (256 - sw[j]) * tr, 9);
      else if (mode == SMOOTH_V_PRED) p = round2_(sw[i] * ra[j] + (256 - sw[i]) * bl, 8);
      else p = round2_(sw[j] * rl[i] + (256 - sw[j]) * tr, 8);
      pred[idx] = (uint16_t)p;
    }
  } else {
    const int pa = mode_angle_of(mode) + angle_delta * 3;
    // working copies of the edges
    for (int i = LANE; i < 2 * n + 1; i += 64) { wa[i - 1] = ra[i - 1]; wl[i - 1] = rl[i - 1]; }
    WAVE_SYNC();
    int up_a = 0, up_l = 0;
    if (pa != 90 && pa != 180) {
      if (pa > 90 && pa < 180 && 2 * n >= 24) {
        if (LANE == 0) { const int v = round2_(wl[0] * 5 + wa[-1] * 6 + wa[0] * 5, 4); wa[-1] = (uint16_t)v; wl[-1] = (uint16_t)v; }
        WAVE_SYNC();
      }
      if (have_above) {
        const int st = edge_strength_dev(n, n, ftype, pa - 90);
        const int num = imin_(n, max_x - x + 1) + (pa < 90 ? n : 0) + 1;
        edge_filter_dev(wa, num, st, tmp);
      }
      if (have_left) {
        const int st = edge_strength_dev(n, n, ftype, pa - 180);
        const int num = imin_(n, max_y - y + 1) + (pa > 180 ? n : 0) + 1;
        edge_filter_dev(wl, num, st, tmp);
      }
    }
    up_a = edge_upsample_sel_dev(n, n, ftype, pa - 90);
    if (up_a) edge_upsample_dev(wa, n + (pa < 90 ? n : 0), bd, tmp);
    up_l = edge_upsample_sel_dev(n, n, ftype, pa - 180);
    if (up_l) edge_upsample_dev(wl, n + (pa > 180 ? n : 0), bd, tmp);
    int dx = 0, dy = 0;
    if (pa < 90) dx = dr_deriv_dev(pa); else if (pa > 90 && pa < 180) dx = dr_deriv_dev(180 - pa);
    if (pa > 90 && pa < 180) dy = dr_deriv_dev(pa - 90); else if (pa > 180) dy = dr_deriv_dev(270 - pa);
    for (int idx = LANE; idx < nn; idx += 64) {
      const int i = idx >> log2w, j = idx & (n - 1);
      int v;
      if (pa < 90) {
        const int max_base = (2 * n - 1) << up_a;
        const int id = (i + 1) * dx, base = (id >> (6 - up_a)) + (j << up_a), sh = ((id << up_a) >> 1) & 0x1F;
        v = base < max_base ? round2_(wa[base] * (32 - sh) + wa[base + 1] * sh, 5) : wa[max_base];
      } else if (pa > 90 && pa < 180) {
        int id = (j << 6) - (i + 1) * dx, base = id >> (6 - up_a);
        if (base >= -(1 << up_a)) { const int sh = ((id << up_a) >> 1) & 0x1F; v = round2_(wa[base] * (32 - sh) + wa[base + 1] * sh, 5); }
        else { id = (i << 6) - (j + 1) * dy; base = id >> (6 - up_l); const int sh = ((id << up_l) >> 1) & 0x1F; v = round2_(wl[base] * (32 - sh) + wl[base + 1] * sh, 5); }
      } else if (pa > 180) {
        const int id = (j + 1) * dy, base = (id >> (6 - up_l)) + (i << up_l), sh = ((id << up_l) >> 1) & 0x1F;
        v = round2_(wl[base] * (32 - sh) + wl[base + 1] * sh, 5);
      } else if (pa == 90) v = wa[j];
      else v = wl[i];
      pred[idx] = (uint16_t)v;
    }
  }
  WAVE_SYNC();
}

// spec 7.11.5 (4:4:4): pred holds the DC prediction on entry; luma reconstruction read from f->rec[0].
__device__ inline void predict_cfl_dev(const LDS FrameDev *f, int x, int y, int log2w, int alpha, const LDS uint16_t *dcp, LDS uint16_t *out) {
  const int n = 1 << log2w, nn = n * n, mx = (1 << f->bd) - 1, rs = f->stride;
  const uint16_t *luma = f->rec[0] + y * rs + x;
  int s = 0;
  for (int idx = LANE; idx < nn; idx += 64) s += luma[(idx >> log2w) * rs + (idx & (n - 1))] << 3;
  s = wave_sum_i32(s);
  const int avg = round2_(s, 2 * log2w);
  for (int idx = LANE; idx < nn; idx += 64) {
    const int l = (luma[(idx >> log2w) * rs + (idx & (n - 1))] << 3) - avg;
    const int v = alpha * l;
    const int sc = v >= 0 ? round2_(v, 6) : -round2_(-v, 6);
    out[idx] = (uint16_t)iclamp_(dcp[idx] + sc, 0, mx);
  }
  WAVE_SYNC();
}

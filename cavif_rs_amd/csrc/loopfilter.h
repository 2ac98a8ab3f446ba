// loopfilter.h -- kernels K2 (deblocking, spec 7.14) and K3 (CDEF direction + strength search + filter,
// spec 7.15).  Frame-level, fully data-parallel: K2 runs one thread per 4-sample edge line (all edges of a
// pass are independent because a filter never reaches past half of the smaller adjoining transform);
// K3 runs one workgroup per 64x64 filter block, one wavefront per 8x8 block (64 lanes = 64 pixels),
// wave-reduced SSE per strength candidate, LDS reduction across the workgroup.
// rav1e equivalents (absent): src/deblock.rs, src/cdef.rs, rdo.rs::rdo_loop_decision.
#pragma once
#include "dev_common.h"
#include "dev_pk16.h"

__device__ inline void filter_edge_sample_dev(uint16_t *px, int step, int filter_size, int plane, int lvl, int sharp, int bd) {
  const int sh = sharp > 4 ? 2 : (sharp > 0 ? 1 : 0);
  const int limit = sharp > 0 ? iclamp_(lvl >> sh, 1, 9 - sharp) : imax_(1, lvl >> sh);
  const int blimit = 2 * (lvl + 2) + limit, thresh = lvl >> 4, s8 = bd - 8;
  const int limit_bd = limit << s8, blimit_bd = blimit << s8, thresh_bd = thresh << s8, flat_bd = 1 << s8;
#define PX_P(i) ((int)px[-((i) + 1) * step])
#define PX_Q(i) ((int)px[(i) * step])
  const int p0 = PX_P(0), p1 = PX_P(1), q0 = PX_Q(0), q1 = PX_Q(1);
  const int hev = iabs_(p1 - p0) > thresh_bd || iabs_(q1 - q0) > thresh_bd;
  const int flen = filter_size == 4 ? 4 : (plane != 0 ? 6 : (filter_size == 8 ? 8 : 16));
  int mask = iabs_(p1 - p0) <= limit_bd && iabs_(q1 - q0) <= limit_bd && (iabs_(p0 - q0) * 2 + iabs_(p1 - q1) / 2) <= blimit_bd;
  if (flen >= 6) mask = mask && iabs_(PX_P(2) - p1) <= limit_bd && iabs_(PX_Q(2) - q1) <= limit_bd;
  if (flen >= 8) mask = mask && iabs_(PX_P(3) - PX_P(2)) <= limit_bd && iabs_(PX_Q(3) - PX_Q(2)) <= limit_bd;
  if (!mask) return;
  int flat = 0, flat2 = 0;
  if (filter_size >= 8) {
    flat = iabs_(p1 - p0) <= flat_bd && iabs_(q1 - q0) <= flat_bd && iabs_(PX_P(2) - p0) <= flat_bd && iabs_(PX_Q(2) - q0) <= flat_bd;
    if (flen >= 8) flat = flat && iabs_(PX_P(3) - p0) <= flat_bd && iabs_(PX_Q(3) - q0) <= flat_bd;
  }
  if (filter_size >= 16)
    flat2 = iabs_(PX_P(6) - p0) <= flat_bd && iabs_(PX_Q(6) - q0) <= flat_bd && iabs_(PX_P(5) - p0) <= flat_bd && iabs_(PX_Q(5) - q0) <= flat_bd &&
            iabs_(PX_P(4) - p0) <= flat_bd && iabs_(PX_Q(4) - q0) <= flat_bd;
  if (filter_size == 4 || !flat) {
    const int lo = -(1 << (bd - 1)), hi = (1 << (bd - 1)) - 1, off = 0x80 << s8;
    const int ps1 = p1 - off, ps0 = p0 - off, qs0 = q0 - off, qs1 = q1 - off;
    int filt = hev ? iclamp_(ps1 - qs1, lo, hi) : 0;
    filt = iclamp_(filt + 3 * (qs0 - ps0), lo, hi);
    const int f1 = iclamp_(filt + 4, lo, hi) >> 3, f2 = iclamp_(filt + 3, lo, hi) >> 3;
    px[0] = (uint16_t)(iclamp_(qs0 - f1, lo, hi) + off);
    px[-step] = (uint16_t)(iclamp_(ps0 + f2, lo, hi) + off);
    if (!hev) {
      const int fo = round2_(f1, 1);
      px[step] = (uint16_t)(iclamp_(qs1 - fo, lo, hi) + off);
      px[-2 * step] = (uint16_t)(iclamp_(ps1 + fo, lo, hi) + off);
    }
    return;
  }
  const int log2size = (filter_size == 8 || !flat2) ? 3 : 4;
  const int n = log2size == 4 ? 6 : (plane == 0 ? 3 : 2);
  const int n2 = (log2size == 3 && plane == 0) ? 0 : 1;
  int F[16], out[16];
  for (int i = -(n + 1); i <= n; i++) F[i + 8] = i < 0 ? PX_P(-i - 1) : PX_Q(i);
  for (int i = -n; i < n; i++) {
    int t = 0;
    for (int j = -n; j <= n; j++) { const int p = iclamp_(i + j, -(n + 1), n); t += F[p + 8] * (iabs_(j) <= n2 ? 2 : 1); }
    out[i + 8] = round2_(t, log2size);
  }
  for (int i = -n; i < n; i++) px[i * step] = (uint16_t)out[i + 8];
#undef PX_P
#undef PX_Q
}

// Edge geometry shared by the level search and the filter: thread id -> (r, c, line i); returns the filter size or 0.
__device__ __forceinline__ int deblock_edge(const FrameDev *f, int plane, int pass, uint32_t tid, int *r_, int *c_, int *i_) {
  int r, c, i;                   // (tid < 4 x the frame's 4x4 cells: 32-bit divisions)
  if (pass == 0) {       // consecutive threads walk mi columns of one pixel row
    const uint32_t mc = (uint32_t)f->mi_cols; const int line = (int)(tid / mc); c = (int)(tid - (uint32_t)line * mc); r = line >> 2; i = line & 3;
  } else {               // consecutive threads walk pixel columns of one mi row
    const uint32_t mc4 = (uint32_t)f->mi_cols * 4u; r = (int)(tid / mc4); const int col_px = (int)(tid - (uint32_t)r * mc4); c = col_px >> 2; i = col_px & 3;
  }
  *r_ = r; *c_ = c; *i_ = i;
  if (r >= f->mi_rows || c >= f->mi_cols) return 0;
  const int x = c * 4, y = r * 4;
  if (x >= f->w || y >= f->h) return 0;
  if (pass == 0 && c == 0) return 0;
  if (pass == 1 && r == 0) return 0;
  const int ms = f->mi_stride;
  const uint8_t *txm = plane == 0 ? f->m_txsize : f->m_bsize;     // luma: the block's transform size; chroma (4:4:4): the block's largest transform
  // 2:1 transform codes (5 = 4x8, 6 = 8x4): the extent across the edge direction -- width for vertical edges, height for horizontal ones
  auto ext = [&](int code) { return code <= 4 ? ((plane && code == 4) ? 32 : 4 << code) : (((code == 5) == (pass == 0)) ? 4 : 8); };   // (the chroma transforms of a 64x64 block are 32x32)
  const int cur = imin_(64, ext(txm[r * ms + c]));
  if (pass == 0 ? (x % cur) != 0 : (y % cur) != 0) return 0;
  const int prev = pass == 0 ? imin_(64, ext(txm[r * ms + c - 1])) : imin_(64, ext(txm[(r - 1) * ms + c]));
  const int base = imin_(cur, prev);
  return plane == 0 ? imin_(16, base) : imin_(8, base);
}

// K2a: deblock level search (rav1e deblock_filter_optimize with fast_deblock == false; oracle/av1o_filters.c
// deblock_tally_line).  One thread per edge line, judged on the unfiltered reconstruction: with sharpness 0 the filter of a
// line is off below the smallest level Lmin that passes the masks and can only change where L >> 4 changes, so a line adds at
// most two (level range, SSE delta) pairs to the (plane, pass) difference array -- LDS first, then one global atomic per
// non-zero entry and workgroup.  grid = (line chunks, plane * 2 + pass, frame).
// Edge lines are sparse among the candidate positions (one 4x4 column in two .. four carries a transform edge: a quarter of the lanes had work, SQ_THREAD_CYCLES_VALU
// at 23 % in round 4), so a workgroup first walks MI_DBK_CHUNK candidate positions with the cheap geometry test and queues the edge lines it finds in LDS (packed
// column | line | size), then judges the queued lines with every lane busy.  The tallies are exact 64-bit sums: the order of the lines does not matter.
#define MI_DBK_CHUNK 2048
__device__ __forceinline__ int deblock_queue_lines(const FrameDev *f, int plane, int pass, uint32_t base, LDS uint32_t *list, LDS int *cnt) {
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
  for (uint32_t it = 0; it < MI_DBK_CHUNK; it += 256) {
    int r, c, i;
    const int fsz = deblock_edge(f, plane, pass, base + it + threadIdx.x, &r, &c, &i);
    if (fsz) list[__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = (uint32_t)c | ((uint32_t)(r * 4 + i) << 14) | ((uint32_t)(fsz >> 3) << 30);   // fsz 4 / 8 / 16 -> 0 / 1 / 2; 14 + 16 + 2 bits: 4x4 columns < 16384 and lines < 65536 = the 65536 x 65536 the entry points accept
  }
  __syncthreads();
  return *cnt;
}
__global__ __launch_bounds__(256) void deblock_tally_kernel(const FrameDev *__restrict__ frames, int nframes) {
  const FrameDev *f = frames + blockIdx.z;
  const int plane = blockIdx.y >> 1, pass = blockIdx.y & 1;
  if (plane >= f->np || f->fast_deblock || frame_idle(f)) return;
  if ((uint32_t)blockIdx.x * MI_DBK_CHUNK >= (uint32_t)f->mi_cols * (uint32_t)f->mi_rows * 4u) return;
  __shared__ long long ldiff[65];
  __shared__ uint32_t list_s[MI_DBK_CHUNK];
  __shared__ int cnt_s;
  if (threadIdx.x < 65) ldiff[threadIdx.x] = 0;
  const int nlines = deblock_queue_lines(f, plane, pass, (uint32_t)blockIdx.x * MI_DBK_CHUNK, (LDS uint32_t *)list_s, (LDS int *)&cnt_s);
  // A lane works on one line at a time and on one FILTER RUN of it per pass of the outer loop.  Above its lowest level lmin the level reaches the filter only through
  // the high-edge-variance threshold (L >> 4), and hev only falls as L grows: one run at lmin, and when hev held there a second one at the first multiple of 16 where
  // it no longer does -- each run's SSE change is booked from its level to the next run's (or to 64).  Lines whose lmin is above 63 never get this far: a lane
  // without a run refills from the queue first, so the filter runs execute on fuller wavefronts (2.54 -> 2.32 ms; lane utilisation of the kernel was 34 %).
  const int s8 = f->bd - 8, one = 1 << s8, step = pass == 0 ? 1 : f->stride;
  int j = threadIdx.x, a = 0, a_next = 64, sse0 = 0, fsz = 4;
  int R[16], S[16];
  bool have = false;
  for (;;) {
    while (!have && j < nlines) {
      const uint32_t e = list_s[j];
      j += 256;
      const int c = (int)(e & 0x3FFF), line = (int)((e >> 14) & 0xFFFF), r = line >> 2, i = line & 3;
      fsz = 4 << (e >> 30);
      const int half = fsz == 4 ? 2 : (fsz == 8 ? 4 : 8);
      const size_t o = pass == 0 ? (size_t)(r * 4 + i) * f->stride + c * 4 : (size_t)(r * 4) * f->stride + c * 4 + i;
      const uint16_t *rec = f->rec[plane] + o, *src = f->src[plane] + o;
#pragma unroll
      for (int k = 0; k < 16; k++) { const int q = k - 8; const bool in = q >= -half && q < half; R[k] = in ? (int)rec[(long long)q * step] : 0; S[k] = in ? (int)src[(long long)q * step] : 0; }
      const int flen = fsz == 4 ? 4 : (plane != 0 ? 6 : (fsz == 8 ? 8 : 16));
      int dmax = imax_(iabs_(R[6] - R[7]), iabs_(R[9] - R[8]));
      if (flen >= 6) dmax = imax_(dmax, imax_(iabs_(R[5] - R[6]), iabs_(R[10] - R[9])));
      if (flen >= 8) dmax = imax_(dmax, imax_(iabs_(R[4] - R[5]), iabs_(R[11] - R[10])));
      const int b = iabs_(R[7] - R[8]) * 2 + iabs_(R[6] - R[9]) / 2;
      const int B = (b + one - 1) >> s8;
      const int lmin = imax_(1, imax_((dmax + one - 1) >> s8, B > 4 ? (B - 2) / 3 : 0));
      if (lmin <= 63) {
        have = true; a = lmin;
        const int hm = imax_(iabs_(R[6] - R[7]), iabs_(R[9] - R[8]));
        const int k16 = ((hm + one - 1) >> s8) << 4;          // the first multiple of 16 whose threshold is not below hm
        a_next = (hm > ((lmin >> 4) << s8) && k16 < 64) ? k16 : 64;
        sse0 = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { const int d = R[k] - S[k]; sse0 += d * d; }
      }
    }
    if (!have) break;
    uint16_t t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = (uint16_t)R[k];
    filter_edge_sample_dev(t + 8, 1, fsz, plane, a, 0, f->bd);
    int sse = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int d = (int)t[k] - S[k]; sse += d * d; }
    const long long dl = (long long)(sse - sse0);
    if (dl) { atomicAdd((unsigned long long *)&ldiff[a], (unsigned long long)dl); atomicAdd((unsigned long long *)&ldiff[a_next], (unsigned long long)(-dl)); }
    if (a_next < 64) { a = a_next; a_next = 64; } else have = false;
  }
  __syncthreads();
  if (threadIdx.x < 65 && ldiff[threadIdx.x]) atomicAdd((unsigned long long *)&f->lf_tally[(plane * 2 + pass) * 65 + threadIdx.x], (unsigned long long)ldiff[threadIdx.x]);
}
// one thread per frame: prefix sums -> level per (luma vertical, luma horizontal, U, V); lowest level wins ties
__global__ void deblock_pick_kernel(FrameDev *frames, int nframes) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nframes) return;
  FrameDev *f = frames + k;
  if (frame_idle(f)) return;
  if (!f->fast_deblock) {
    int lv[4] = { 0, 0, 0, 0 };
    for (int pass = 0; pass < 2; pass++) {
      long long acc = 0, best = 0; int bl = 0;
      for (int l = 0; l < 64; l++) { acc += f->lf_tally[pass * 65 + l]; if (acc < best) { best = acc; bl = l; } }
      lv[pass] = bl;
    }
    for (int plane = 1; plane < f->np; plane++) {
      long long a0 = 0, a1 = 0, best = 0; int bl = 0;
      for (int l = 0; l < 64; l++) { a0 += f->lf_tally[(plane * 2) * 65 + l]; a1 += f->lf_tally[(plane * 2 + 1) * 65 + l]; if (a0 + a1 < best) { best = a0 + a1; bl = l; } }
      lv[plane + 1] = bl;
    }
    if (!lv[0] && !lv[1]) lv[2] = lv[3] = 0;
    for (int i = 0; i < 4; i++) f->lf_level[i] = lv[i];
  }
  for (int i = 0; i < 4; i++) f->lf_out[i] = f->lf_level[i];
}

// pass 0: vertical edges (filter along x), pass 1: horizontal edges.  A workgroup queues the edge lines among MI_DBK_CHUNK candidate positions (deblock_queue_lines),
// then filters them one per lane: the lines of a pass are independent (a filter never reaches past half of the smaller adjoining transform).
__global__ __launch_bounds__(256) void deblock_kernel(const FrameDev *__restrict__ frames, int nframes, int pass) {
  const FrameDev *f = frames + blockIdx.z;
  const int plane = blockIdx.y;
  if (plane >= f->np || frame_idle(f)) return;
  const int L = plane == 0 ? f->lf_level[pass] : f->lf_level[plane + 1];
  if (!L) return;
  if ((uint32_t)blockIdx.x * MI_DBK_CHUNK >= (uint32_t)f->mi_cols * (uint32_t)f->mi_rows * 4u) return;
  __shared__ uint32_t list_s[MI_DBK_CHUNK];
  __shared__ int cnt_s;
  const int nlines = deblock_queue_lines(f, plane, pass, (uint32_t)blockIdx.x * MI_DBK_CHUNK, (LDS uint32_t *)list_s, (LDS int *)&cnt_s);
  for (int j = threadIdx.x; j < nlines; j += 256) {
    const uint32_t e = list_s[j];
    const int c = (int)(e & 0x3FFF), line = (int)((e >> 14) & 0xFFFF), r = line >> 2, i = line & 3, fsz = 4 << (e >> 30);
    const int x = c * 4, y = r * 4;
    uint16_t *px = pass == 0 ? f->rec[plane] + (size_t)(y + i) * f->stride + x : f->rec[plane] + (size_t)y * f->stride + x + i;
    filter_edge_sample_dev(px, pass == 0 ? 1 : f->stride, fsz, plane, L, f->lf_sharp, f->bd);
  }
}

// ---------------------------------------------------------------- CDEF
// sums over lanes 0..31 and over lanes 32..63 on the DPP network: afterwards lanes 16..31 hold the first half's total, lanes 48..63 the second half's
__device__ __forceinline__ int half_sum_i32(int v) {
  v += DPP_(0, v, 0xB1, 0xF); v += DPP_(0, v, 0x4E, 0xF); v += DPP_(0, v, 0x141, 0xF); v += DPP_(0, v, 0x140, 0xF);
  v += DPP_(0, v, 0x142, 0xA);
  return v;
}
__device__ __forceinline__ int cdef_dir_off(int dir, int k, int comp) {
  const int8_t d[8][2][2] = { { { -1, 1 }, { -2, 2 } }, { { 0, 1 }, { -1, 2 } }, { { 0, 1 }, { 0, 2 } }, { { 0, 1 }, { 1, 2 } },
                              { { 1, 1 }, { 2, 2 } }, { { 1, 0 }, { 2, 1 } }, { { 1, 0 }, { 2, 0 } }, { { 1, 0 }, { 2, -1 } } };
  return d[dir][k][comp];
}
__device__ __forceinline__ int constrain_dev(int diff, int thr, int damping) {
  if (!thr) return 0;
  const int adj = imax_(0, damping - (31 - __clz(thr))), mag = iabs_(diff);
  const int v = iclamp_(thr - (mag >> adj), 0, mag);
  return diff < 0 ? -v : v;
}
// The 12 tap samples of one pixel for direction `dir` (spec 7.15.3 / cdef_get_at): order k=0,1 x sign -,+ x
// { primary(dir), secondary(dir-2), secondary(dir+2) }.  A sample outside the frame (CdefAvailable = 0) is skipped by the spec; here it takes the centre pixel's
// value `x`, which is the same thing -- its difference is 0, so constrain() adds nothing, and it cannot move the clamp bounds that start at x -- and spares the
// filter arithmetic a validity test per tap and use (a quarter of the strength search's instructions in round 4).
// `interior` (wave-uniform): the whole 8x8 block lies at least two samples inside the frame on every side, so no tap needs a test -- all but the frame's border blocks.
__device__ __forceinline__ void cdef_load_taps(const FrameDev *f, const uint16_t *in, int py, int px_, int dir, int x, int *tap, bool interior) {
  const int st = f->stride, fw = f->mi_cols * 4, fh = f->mi_rows * 4;
  int n = 0;
  if (interior) {
    const uint16_t *c = in + py * st + px_;
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
      for (int sg = -1; sg <= 1; sg += 2) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const int d2 = q == 0 ? dir : ((dir + (q == 1 ? -2 : 2)) & 7);
          tap[n++] = c[sg * (cdef_dir_off(d2, k, 0) * st + cdef_dir_off(d2, k, 1))];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
#pragma unroll
    for (int sg = -1; sg <= 1; sg += 2) {
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int d2 = q == 0 ? dir : ((dir + (q == 1 ? -2 : 2)) & 7);
        const int yy = py + sg * cdef_dir_off(d2, k, 0), xx = px_ + sg * cdef_dir_off(d2, k, 1);
        const bool in_frame = yy >= 0 && yy < fh && xx >= 0 && xx < fw;
        const int val = in[in_frame ? yy * st + xx : py * st + px_];          // one unconditional load per tap (the centre's address where the tap is outside)
        tap[n++] = in_frame ? val : x;
      }
    }
  }
}
// filter value of one pixel from its preloaded taps
__device__ __forceinline__ int cdef_apply_taps(int x, const int *tap, int pri, int sec, int damping, int cs) {
  int sum = 0, mx = x, mn = x;
  const int pt0 = ((pri >> cs) & 1) ? 3 : 4, pt1 = ((pri >> cs) & 1) ? 3 : 2;
#pragma unroll
  for (int n = 0; n < 12; n++) {
    const int k = n / 6, q = n % 3, t = tap[n];
    const int wgt = q == 0 ? (k == 0 ? pt0 : pt1) : (k == 0 ? 2 : 1);
    sum += wgt * constrain_dev(t - x, q == 0 ? pri : sec, damping);
    mx = imax_(mx, t); mn = imin_(mn, t);
  }
  return iclamp_(x + ((8 + sum - (sum < 0)) >> 4), mn, mx);
}
// The same filter in pieces, for the strength search: the clamp bounds and the secondary taps' sum do not depend on the primary strength, and the
// fixed strength list repeats its secondary strengths (0, 0, 1, 1, 2, 3, 3, 3), so the search computes bounds once per sample and a secondary sum
// once per distinct strength.  cdef_apply_taps(x, ...) == cdef_finish(x, cdef_pri_sum(...) + cdef_sec_sum(...), mn, mx).
__device__ __forceinline__ void cdef_bounds(int x, const int *tap, int *mn_, int *mx_) {
  int mx = x, mn = x;
#pragma unroll
  for (int n = 0; n < 12; n++) { mx = imax_(mx, tap[n]); mn = imin_(mn, tap[n]); }
  *mn_ = mn; *mx_ = mx;
}
__device__ __forceinline__ int cdef_pri_sum(int x, const int *tap, int pri, int damping, int cs) {
  const int pt0 = ((pri >> cs) & 1) ? 3 : 4, pt1 = ((pri >> cs) & 1) ? 3 : 2;
  int sum = 0;
#pragma unroll
  for (int n = 0; n < 12; n += 3) sum += (n < 6 ? pt0 : pt1) * constrain_dev(tap[n] - x, pri, damping);
  return sum;
}
__device__ __forceinline__ int cdef_sec_sum(int x, const int *tap, int sec, int damping) {
  int sum = 0;
#pragma unroll
  for (int n = 0; n < 12; n++) if (n % 3 != 0) sum += (n < 6 ? 2 : 1) * constrain_dev(tap[n] - x, sec, damping);
  return sum;
}
__device__ __forceinline__ int cdef_finish(int x, int sum, int mn, int mx) { return iclamp_(x + ((8 + sum - (sum < 0)) >> 4), mn, mx); }
// direction search for one 8x8 luma block by its 64 lanes (one pixel each): the 8 x 15 partial sums are
// accumulated with LDS atomics, lanes 0..7 turn them into the 8 costs (spec 7.15.2)
#define LDS_ADD(ptr, v) __hip_atomic_fetch_add((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
__device__ inline int cdef_direction_dev(const uint16_t *img, int stride, int bd, LDS int *part /* [8][16] */, int *var_out) {
  const int div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };
  const int lane = LANE, i = lane >> 3, j = lane & 7;
  for (int q = lane; q < 128; q += 64) part[q] = 0;
  WAVE_SYNC();
  const int x = (img[(size_t)i * stride + j] >> (bd - 8)) - 128;
  LDS_ADD(&part[0 * 16 + i + j], x); LDS_ADD(&part[1 * 16 + i + j / 2], x); LDS_ADD(&part[2 * 16 + i], x); LDS_ADD(&part[3 * 16 + 3 + i - j / 2], x);
  LDS_ADD(&part[4 * 16 + 7 + i - j], x); LDS_ADD(&part[5 * 16 + 3 - i / 2 + j], x); LDS_ADD(&part[6 * 16 + j], x); LDS_ADD(&part[7 * 16 + i / 2 + j], x);
  WAVE_SYNC();
  const int d = lane & 7;
  int partial[15];
#pragma unroll
  for (int q = 0; q < 15; q++) partial[q] = part[d * 16 + q];
  int cost = 0;
  if (d == 2 || d == 6) { for (int q = 0; q < 8; q++) cost += partial[q] * partial[q]; cost *= div_table[8]; }
  else if (d == 0 || d == 4) {
    for (int q = 0; q < 7; q++) cost += (partial[q] * partial[q] + partial[14 - q] * partial[14 - q]) * div_table[q + 1];
    cost += partial[7] * partial[7] * div_table[8];
  } else {
    for (int q = 0; q < 5; q++) cost += partial[3 + q] * partial[3 + q];
    cost *= div_table[8];
    for (int q = 0; q < 3; q++) cost += (partial[q] * partial[q] + partial[10 - q] * partial[10 - q]) * div_table[2 * q + 2];
  }
  int best = 0, dir = 0, costs[8];
#pragma unroll
  for (int q = 0; q < 8; q++) costs[q] = __shfl(cost, q, 64);
#pragma unroll
  for (int q = 0; q < 8; q++) if (costs[q] > best) { best = costs[q]; dir = q; }
  int opp = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) if (q == ((dir + 4) & 7)) opp = costs[q];
  *var_out = (best - opp) >> 10;
  return dir;
}
__device__ __forceinline__ void cdef_strengths(const FrameDev *f, int plane, int idx, int var, int *pri, int *sec, int *damping) {
  const int cs = f->bd - 8;
  const int st = plane == 0 ? f->cdef_y[idx] : f->cdef_uv[idx];
  int p = (st >> 2) << cs, s = st & 3; if (s == 3) s = 4; s <<= cs;
  *damping = f->cdef_damping + cs - (plane > 0);
  if (plane == 0) { const int vs = (var >> 6) ? imin_(31 - __clz(var >> 6), 12) : 0; p = var ? (p * (4 + vs) + 8) >> 4 : 0; }
  *pri = p; *sec = s;
}

// grid.x = sb index, grid.y = frame; 256 threads = 4 waves, wave w handles 8x8 blocks w, w+4, ...
// Every strength index >= 1 of the fixed list has a non-zero primary strength, so the filter direction of a block is
// its luma direction for all candidates and the 12 tap samples per pixel and plane are loaded once.
// (three waves per SIMD: at four the packed strength search spills 40 VGPRs -- 0.13 ms faster, and 5.7 GB of scratch traffic per launch on top of the 2.7 GB the planes cost)
__global__ __launch_bounds__(256, 3) void cdef_kernel(const FrameDev *__restrict__ frames, int write_final) {
  const FrameDev *f = frames + blockIdx.y;
  const int sbi = blockIdx.x;
  if (sbi >= f->sb_rows * f->sb_cols || frame_idle(f)) return;
  __shared__ unsigned long long costs[8];
  __shared__ int any_blocks, best_idx;
  __shared__ int part_s[4][128];
  __shared__ int dirvar[64][2];
  const int wave = threadIdx.x >> 6, lane = LANE;
  LDS int *part = (LDS int *)part_s[wave];
  const int sr = sbi / f->sb_cols, sc = sbi % f->sb_cols, ms = f->mi_stride, cs = f->bd - 8;
  if (threadIdx.x < 8) costs[threadIdx.x] = 0;
  if (threadIdx.x == 0) { any_blocks = 0; best_idx = 0; }
  __syncthreads();
  const int py_l = lane >> 3, px_l = lane & 7;
  if (f->enable_cdef) {
    // Strength search, two 8x8 blocks per wavefront pass: lanes 0..31 take block bA, lanes 32..63 block bB = bA + 4, every lane two horizontally adjacent samples
    // in packed 16-bit arithmetic (dev_pk16.h) -- 10-bit samples, their differences, the constrained differences (|.| <= strength) and a pixel's weighted tap sums
    // (< 1024 + 256) all fit 16 bits.  Same filter as cdef_apply_taps / cdef_pri_sum / cdef_sec_sum, sample for sample; a block's sums land on its own half's lanes.
    const int hs = lane >> 5, l32 = lane & 31, py2 = l32 >> 2, px2 = (l32 & 3) * 2;
    const int st = f->stride, fw = f->mi_cols * 4, fh = f->mi_rows * 4;
    for (int it = 0; it < 8; it++) {
      const int bA = wave + 8 * it, bB = bA + 4;
      const int rA = sr * 16 + (bA >> 3) * 2, cA = sc * 16 + (bA & 7) * 2, rB = sr * 16 + (bB >> 3) * 2, cB = sc * 16 + (bB & 7) * 2;
      bool actA = rA < f->mi_rows && cA < f->mi_cols, actB = rB < f->mi_rows && cB < f->mi_cols;
      if (actA) actA = !(f->m_skip[rA * ms + cA] & f->m_skip[(rA + 1) * ms + cA] & f->m_skip[rA * ms + cA + 1] & f->m_skip[(rA + 1) * ms + cA + 1] & 1);   // bit 0 of the map (the rest is the segment id)
      if (actB) actB = !(f->m_skip[rB * ms + cB] & f->m_skip[(rB + 1) * ms + cB] & f->m_skip[rB * ms + cB + 1] & f->m_skip[(rB + 1) * ms + cB + 1] & 1);
      actA = uni32(actA) != 0; actB = uni32(actB) != 0;
      if (!actA && !actB) continue;
      int dirA = 0, varA = 0, dirB = 0, varB = 0;
      if (actA) { dirA = cdef_direction_dev(f->rec[0] + (size_t)(rA * 4) * st + cA * 4, st, f->bd, part, &varA); if (lane == 0) { dirvar[bA][0] = dirA; dirvar[bA][1] = varA; } }
      if (actB) { dirB = cdef_direction_dev(f->rec[0] + (size_t)(rB * 4) * st + cB * 4, st, f->bd, part, &varB); if (lane == 0) { dirvar[bB][0] = dirB; dirvar[bB][1] = varB; } }
      // an idle half walks the other half's block (valid addresses, results unused)
      const bool useB = actB && (hs || !actA);
      const bool act = hs ? actB : actA;
      const int r = useB ? rB : rA, c = useB ? cB : cA;
      const int y = r * 4 + py2, x = c * 4 + px2;
      const bool intA = rA * 4 >= 2 && cA * 4 >= 2 && rA * 4 + 10 <= fh && cA * 4 + 10 <= fw, intB = rB * 4 >= 2 && cB * 4 >= 2 && rB * 4 + 10 <= fh && cB * 4 + 10 <= fw;
      const bool interior = (!actA || intA) && (!actB || intB);                       // wave-uniform: no tap of either block needs a frame test
      const int cell = (r >> 1) * (f->pw >> 3) + (c >> 1);
      const uint32_t cell_sv = f->svar8[cell], cell_act = f->act[cell];
      long long cst = 0;                                                              // candidate idx of block A on lane idx, of block B on lane 32 + idx
      for (int p = 0; p < f->np; p++) {
        const uint16_t *in = f->rec[p];
        const pk16 un = pk_load2(in + y * st + x), sv = pk_load2(f->src[p] + y * st + x);
        pk16 tap[12]; int n = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
#pragma unroll
          for (int sg = -1; sg <= 1; sg += 2) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
              const int dA = q == 0 ? dirA : ((dirA + (q == 1 ? -2 : 2)) & 7), dB = q == 0 ? dirB : ((dirB + (q == 1 ? -2 : 2)) & 7);
              const int dyA = sg * cdef_dir_off(dA, k, 0), dxA = sg * cdef_dir_off(dA, k, 1), dyB = sg * cdef_dir_off(dB, k, 0), dxB = sg * cdef_dir_off(dB, k, 1);
              const int dy = useB ? dyB : dyA, dx = useB ? dxB : dxA;
              if (interior) tap[n] = pk_load2(in + (y + dy) * st + x + dx);
              else {
                // a sample outside the frame takes its centre sample's value (cdef_load_taps); the pair is fetched where its row exists
                const int yy = y + dy, xx = x + dx;
                const bool rowok = yy >= 0 && yy < fh;
                const pk16 t = pk_load2(in + (rowok ? yy : y) * st + (rowok ? xx : x));
                const uint32_t tv = pk_to_u32(t), cv = pk_to_u32(un);
                const bool ok0 = rowok && xx >= 0 && xx < fw, ok1 = rowok && xx + 1 >= 0 && xx + 1 < fw;
                tap[n] = pk_from_u32(((ok0 ? tv : cv) & 0xFFFFu) | ((ok1 ? tv : cv) & 0xFFFF0000u));
              }
              n++;
            }
          }
        }
        pk16 mn = un, mx = un;
#pragma unroll
        for (int q = 0; q < 12; q++) { mx = pk_max(mx, tap[q]); mn = pk_min(mn, tap[q]); }
        const bool psy = p == 0 && !f->tune_psnr;
        uint32_t my_sse = 0, my_s = 0, my_q = 0;
        int ssec = -1; pk16 ssum = pk_splat(0);
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
          int priA, priB, sec, damping, sec2, damping2;
          cdef_strengths(f, p, idx, varA, &priA, &sec, &damping); cdef_strengths(f, p, idx, varB, &priB, &sec2, &damping2);     // the variance only moves the luma primary strength
          const int pri = useB ? priB : priA;
          if (sec != ssec) {                                                          // wave-uniform: the list repeats its secondary strengths
            ssec = sec; ssum = pk_splat(0);
            if (sec) {
              const pk16 thr = pk_splat(sec), adj = pk_splat(imax_(0, damping - (31 - __clz(sec))));
#pragma unroll
              for (int q = 0; q < 12; q++) if (q % 3 != 0) {
                const pk16 diff = tap[q] - un, mag = pk_abs(diff), sgn = diff >> pk_splat(15);
                const pk16 v = pk_min(pk_max(thr - (mag >> adj), pk_splat(0)), mag);
                ssum += ((v ^ sgn) - sgn) * pk_splat(q < 6 ? 2 : 1);
              }
            }
          }
          pk16 sum = ssum;
          {
            const int odd = (pri >> cs) & 1;
            const pk16 thr = pk_splat(pri), adj = pk_splat(pri ? imax_(0, damping - (31 - __clz(pri))) : 0), w0 = pk_splat(odd ? 3 : 4), w1 = pk_splat(odd ? 3 : 2);
#pragma unroll
            for (int q = 0; q < 12; q += 3) {
              const pk16 diff = tap[q] - un, mag = pk_abs(diff), sgn = diff >> pk_splat(15);
              const pk16 v = pk_min(pk_max(thr - (mag >> adj), pk_splat(0)), mag);       // 0 when the strength is 0
              sum += ((v ^ sgn) - sgn) * (q < 6 ? w0 : w1);
            }
          }
          // cdef_finish: x + ((8 + sum - (sum < 0)) >> 4), clamped to the taps' range (no strength at all leaves x: sum = 0, mn <= x <= mx)
          const pk16 v = pk_min(pk_max(un + ((pk_splat(8) + sum + (sum >> pk_splat(15))) >> pk_splat(4)), mn), mx);
          const pk16 d = v - sv;
          int e = pk_dot2(d, d, 0);                                                     // 32 lanes x 2 samples x 1023^2 < 2^26 per block
          e = half_sum_i32(e);
          const uint32_t eA = (uint32_t)__builtin_amdgcn_readlane(e, 31), eB = (uint32_t)__builtin_amdgcn_readlane(e, 63);
          if (l32 == idx) my_sse = hs ? eB : eA;
          if (psy) {
            int s1 = half_sum_i32(pk_dot2(v, pk_splat(1), 0)), s2 = half_sum_i32(pk_dot2(v, v, 0));
            const uint32_t s1A = (uint32_t)__builtin_amdgcn_readlane(s1, 31), s1B = (uint32_t)__builtin_amdgcn_readlane(s1, 63);
            const uint32_t s2A = (uint32_t)__builtin_amdgcn_readlane(s2, 31), s2B = (uint32_t)__builtin_amdgcn_readlane(s2, 63);
            if (l32 == idx) { my_s = hs ? s1B : s1A; my_q = hs ? s2B : s2A; }
          }
        }
        // Tune::Psychovisual (rav1e rdo_loop_plane_error): luma through the cdef-dist kernel of the 8x8 block, chroma SSE x activity
        long long e;
        if (psy) e = psy_cell_dist(my_sse, my_s, my_q, cell_sv, cell_act, 8, f->bd);
        else e = (long long)(((unsigned long long)my_sse * cell_act + 8192) >> 14);
        cst += (e * f->wq[p]) >> 5;
      }
      if (act && l32 < 8) atomicAdd(&costs[l32], (unsigned long long)cst);
      if (lane == 0) any_blocks = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int best = -1;
    if (any_blocks) { best = 0; unsigned long long bc = costs[0]; for (int idx = 1; idx < 8; idx++) if (costs[idx] < bc) { bc = costs[idx]; best = idx; } }
    best_idx = best;
    f->cdef_idx[sbi] = (int8_t)best;
  }
  __syncthreads();
  if (!write_final) return;
  const int best = best_idx;
  for (int b = wave; b < 64; b += 4) {
    const int r = sr * 16 + (b >> 3) * 2, c = sc * 16 + (b & 7) * 2;
    if (r >= f->mi_rows || c >= f->mi_cols) continue;
    const int sk = f->m_skip[r * ms + c] & f->m_skip[(r + 1) * ms + c] & f->m_skip[r * ms + c + 1] & f->m_skip[(r + 1) * ms + c + 1] & 1;      // bit 0 of the map (the rest is the segment id)
    const int filt = best > 0 && !sk;                    // index 0 of the list is (0, 0): nothing to filter
    const int ydir = filt ? dirvar[b][0] : 0, var = filt ? dirvar[b][1] : 0;
    for (int p = 0; p < f->np; p++) {
      const int y = r * 4 + py_l, x = c * 4 + px_l;
      int v = f->rec[p][(size_t)y * f->stride + x];
      if (filt) {
        int pri, sec, damping; cdef_strengths(f, p, best, var, &pri, &sec, &damping);
        if (pri || sec) { int tap[12]; cdef_load_taps(f, f->rec[p], y, x, ydir, v, tap, r * 4 >= 2 && c * 4 >= 2 && r * 4 + 10 <= f->mi_rows * 4 && c * 4 + 10 <= f->mi_cols * 4); v = cdef_apply_taps(v, tap, pri, sec, damping, cs); }
      }
      f->fin[p][(size_t)y * f->stride + x] = (uint16_t)v;
    }
  }
}

// ---------------------------------------------------------------- activity mask (Tune::Psychovisual)
// One thread per 8x8 luma cell of the padded source: the four 4x4 variances (8x8-equivalent), the 8x8 variance and the cell's
// activity scale = boost(var, var) (rav1e activity.rs ActivityMask::fill_scales; oracle av1o_activity).  grid = (cells / 256, frames)
__global__ __launch_bounds__(256) void activity_kernel(const FrameDev *__restrict__ frames) {
  const FrameDev *f = frames + blockIdx.y;
  const int cw = f->pw >> 3, chh = f->ph >> 3, cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= cw * chh || frame_idle(f)) return;
  // the state the later stages expect zeroed (decoded flags, deblock tallies, the tile search's progress counters): cleared here, ahead of the
  // tile search on the same stream, instead of one memset per frame queued between the kernels
  for (int i = cell; i < f->zero_words; i += cw * chh) ((uint32_t *)f->m_decoded)[i] = 0u;
  const int cy = cell / cw, cx = cell - cy * cw;
  uint32_t s8 = 0, q8 = 0;
  for (int k = 0; k < 4; k++) {
    const int x0 = cx * 8 + (k & 1) * 4, y0 = cy * 8 + (k >> 1) * 4;
    uint32_t s4 = 0, q4 = 0;
    for (int i = 0; i < 4; i++) {
      const uint16_t *row = f->src[0] + (size_t)(y0 + i) * f->stride + x0;
      for (int j = 0; j < 4; j++) { const uint32_t v = row[j]; s4 += v; q4 += v * v; }
    }
    ((uint32_t *)f->svar4)[(y0 >> 2) * f->mi_stride + (x0 >> 2)] = psy_cell_var(s4, q4, 4, f->bd);
    s8 += s4; q8 += q4;
  }
  const uint32_t v = psy_cell_var(s8, q8, 8, f->bd);
  ((uint32_t *)f->svar8)[cell] = v;
  ((uint32_t *)f->act)[cell] = f->tune_psnr ? 16384u : psy_boost_q14(v, v);
}

// Segmentation (oracle/av1o_segment.c, rav1e segmentation.rs as recalled): one workgroup per frame, after activity_kernel.  The visible cells'
// scale buckets go into an LDS histogram; prefix sums turn every k-means step into a handful of lookups, so the six fits (k = 8 .. 3) run on
// six lanes; lane 0 picks the most evenly spaced one, derives each segment's quantiser index (nearest step to base / sqrt(scale / mean)) and
// fills the table the tile search and the entropy coder read.  lf_out[4 .. 12] = segment count + indices for the host's frame header.
__device__ inline int seg_value_at(const uint32_t *cnt, uint32_t idx) {
  int lo = 0, hi = MI_SEG_BINS - 1;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (cnt[mid + 1] > idx) hi = mid; else lo = mid + 1; }
  return lo;
}
__global__ __launch_bounds__(256) void segment_kernel(FrameDev *frames) {
  FrameDev *f = frames + blockIdx.x;
  if (frame_idle(f)) return;
  __shared__ uint32_t cnt[MI_SEG_BINS + 1];
  __shared__ unsigned long long wsum[MI_SEG_BINS + 1];
  __shared__ int lq[256], cent[6][8];
  __shared__ unsigned long long var[6];
  const int tid = threadIdx.x;
  SegTab *st = (SegTab *)f->seg;
  const int16_t *ac = f->bd == 8 ? av1_ac_q8_dev : av1_ac_q10_dev, *dc = f->bd == 8 ? av1_dc_q8_dev : av1_dc_q10_dev;
  for (int i = tid; i <= MI_SEG_BINS; i += 256) cnt[i] = 0;
  lq[tid] = seg_ilog2_q11((uint32_t)ac[tid]);
  __syncthreads();
  const int cw = f->pw >> 3, vw = (f->w + 7) >> 3, vh = (f->h + 7) >> 3;
  for (int i = tid; i < vw * vh; i += 256) { const int cy = i / vw, cx = i - cy * vw; atomicAdd(&cnt[seg_bucket(f->act[cy * cw + cx]) + 1], 1u); }
  __syncthreads();
  // prefix sums over the bins (count and bucket-weighted count): 16 bins per thread, the 256 partial sums scanned by thread 0
  {
    __shared__ uint32_t pc[256]; __shared__ unsigned long long pw[256]; __shared__ int pmin[256], pmax[256];
    uint32_t c = 0; unsigned long long ws = 0; int bmin = -1, bmax = -1;
    for (int b = tid * 16; b < tid * 16 + 16; b++) { const uint32_t m = cnt[b + 1]; if (m) { if (bmin < 0) bmin = b; bmax = b; } c += m; ws += (unsigned long long)b * m; }
    pc[tid] = c; pw[tid] = ws; pmin[tid] = bmin; pmax[tid] = bmax;
    __syncthreads();
    if (tid == 0) {
      uint32_t cc = 0; unsigned long long ww = 0; int lo = -1, hi = -1;
      for (int i = 0; i < 256; i++) { const uint32_t m = pc[i]; const unsigned long long w = pw[i]; pc[i] = cc; pw[i] = ww; cc += m; ww += w; if (pmin[i] >= 0) { if (lo < 0) lo = pmin[i]; hi = pmax[i]; } }
      wsum[0] = 0;
      var[0] = (lo == hi || cc < 2) ? 1 : 0;       // flag: one scale everywhere -> no segmentation
    }
    __syncthreads();
    c = pc[tid]; ws = pw[tid];
    for (int b = tid * 16; b < tid * 16 + 16; b++) { const uint32_t m = cnt[b + 1]; ws += (unsigned long long)b * m; c += m; wsum[b + 1] = ws; cnt[b + 1] = c; }
  }
  __syncthreads();
  const bool off = var[0] != 0;
  __syncthreads();
  const uint32_t n = cnt[MI_SEG_BINS];
  if (!off && tid < 6) {
    const int k = 8 - tid;
    int c[8];
    for (int j = 0; j < k; j++) c[j] = seg_value_at(cnt, (uint32_t)(((unsigned long long)j * (n - 1)) / (unsigned long long)(k - 1)));
    int limit = 0; while ((n >> limit) != 0) limit++;
    limit *= 2;
    for (int it = 0; it < limit; it++) {
      int changed = 0, lo = 0, t[8];
      for (int j = 0; j + 1 < k; j++) t[j] = (c[j] + c[j + 1] + 1) >> 1;
      for (int j = 0; j < k; j++) {
        const int hi = j == k - 1 ? MI_SEG_BINS : imax_(lo, t[j]);
        const uint32_t m = cnt[hi] - cnt[lo];
        if (m) { const int nc = (int)((wsum[hi] - wsum[lo] + m / 2) / m); if (nc != c[j]) changed = 1; c[j] = nc; }
        lo = hi;
      }
      if (!changed) break;
    }
    long long sum = 0;
    for (int j = 0; j + 1 < k; j++) sum += c[j + 1] - c[j];
    const long long mu = sum / (k - 1);
    unsigned long long v = 0;
    for (int j = 0; j + 1 < k; j++) { const long long d = (c[j + 1] - c[j]) - mu; v += (unsigned long long)(d * d); }
    var[tid] = v;
    for (int j = 0; j < 8; j++) cent[tid][j] = j < k ? c[j] : 0;
  }
  __syncthreads();
  if (tid == 0) {
    int *out = f->lf_out + 4;
    if (off) {
      st->n = 0; f->seg_n = 0; out[0] = 0;
    } else {
      int bi = 0;
      for (int i = 1; i < 6; i++) if (var[i] <= var[bi]) bi = i;          // k = 8 .. 3 in lane order: the last minimum = the fewest segments
      const int k = 8 - bi, mean = (int)((wsum[MI_SEG_BINS] + n / 2) / n), lbase = lq[f->base_q_idx];
      st->n = k; st->mean = mean; f->seg_n = k; out[0] = k;
      // plane deltas of the frame's quantiser (the header's DeltaQ*): dc index per plane and chroma ac index relative to base_q_idx
      for (int i = 0; i < k; i++) {
        const int target = lbase - (cent[bi][k - 1 - i] - mean) * 4;
        int qi = 1, bdiff = 1 << 30;
        for (int q = 1; q < 256; q++) { const int d = iabs_(lq[q] - target); if (d < bdiff) { bdiff = d; qi = q; } }
        st->qidx[i] = qi; out[1 + i] = qi;
        for (int p = 0; p < 3; p++) {
          const int dq = dc[iclamp_(qi + f->seg_ddc[p], 0, 255)], aq = ac[p == 0 ? qi : iclamp_(qi + f->seg_dac[p], 0, 255)];
          st->q[i].dc_q[p] = dq; st->q[i].ac_q[p] = aq;
          st->q[i].dc_recip[p] = 0xFFFFFFFFu / (uint32_t)imax_(1, dq); st->q[i].ac_recip[p] = 0xFFFFFFFFu / (uint32_t)imax_(1, aq);
        }
      }
      for (int j = 0; j + 1 < k; j++) st->thr[j] = (cent[bi][j] + cent[bi][j + 1] + 1) >> 1;
    }
  }
}

// ---------------------------------------------------------------- K0: front end
// RGBA8/RGB8 (HBM) -> planar Y,Cb,Cr (BT.601 full range, f32 FMA exactly as ravif rgb_to_ycbcr,
// av1encoder.rs:504-524) or G,B,R, 8->10 bit expansion, optional alpha plane, edge replication into the
// 64-aligned padding.  One thread per output pixel of the padded plane; 4 B read, 3*2 (+2) B written.
struct FrontParams { float sy_r, sy_g, sy_b, scale, kcb, kcr, shift; int depth, color_model, bpp; };
__global__ __launch_bounds__(256) void frontend_kernel(const uint8_t *pix, int w, int h, int stride_px, FrontParams fp,
                                                       uint16_t *p0, uint16_t *p1, uint16_t *p2, uint16_t *pa, int pw, int ph, int *alpha_flag) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= pw || y >= ph) return;
  const int sx = imin_(x, w - 1), sy = imin_(y, h - 1);
  const uint8_t *p = pix + ((size_t)sy * stride_px + sx) * fp.bpp;
  const int R = p[0], G = p[1], B = p[2], A = fp.bpp == 4 ? p[3] : 255;
  uint16_t o0, o1, o2;
  if (fp.color_model == 1) {
    if (fp.depth == 8) { o0 = (uint16_t)G; o1 = (uint16_t)B; o2 = (uint16_t)R; }
    else { o0 = (uint16_t)((G << 2) | (G >> 6)); o1 = (uint16_t)((B << 2) | (B >> 6)); o2 = (uint16_t)((R << 2) | (R >> 6)); }
  } else {
    const float r = (float)R, g = (float)G, b = (float)B;
    const float yv = fmaf(fp.sy_b, b, fmaf(fp.sy_r, r, fp.sy_g * g));
    const float cb = fmaf(fmaf(b, fp.scale, -yv), fp.kcb, fp.shift);
    const float cr = fmaf(fmaf(r, fp.scale, -yv), fp.kcr, fp.shift);
    const float sat = fp.depth == 8 ? 255.f : 65535.f;
    const float v0 = roundf(yv), v1 = roundf(cb), v2 = roundf(cr);
    o0 = (uint16_t)(v0 < 0.f ? 0.f : (v0 > sat ? sat : v0));
    o1 = (uint16_t)(v1 < 0.f ? 0.f : (v1 > sat ? sat : v1));
    o2 = (uint16_t)(v2 < 0.f ? 0.f : (v2 > sat ? sat : v2));
  }
  const size_t o = (size_t)y * pw + x;
  p0[o] = o0; p1[o] = o1; p2[o] = o2;
  if (pa) pa[o] = fp.depth == 8 ? (uint16_t)A : (uint16_t)((A << 2) | (A >> 6));
  if (A != 255 && x < w && y < h && alpha_flag) *alpha_flag = 1;
}

// ---------------------------------------------------------------- AlphaColorMode::Premultiplied
// convert_alpha_8bit, Premultiplied branch (ravif/src/av1encoder.rs:282-296), as written: a == 0 or a == 255 -> RGBA8::default(),
// otherwise every colour channel becomes (c * 255 / a) as u8 (the cast wraps).  One thread per pixel.
__global__ __launch_bounds__(256) void premultiply_kernel(const uint8_t *src, uint8_t *dst, size_t npx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx) return;
  const uchar4 p = ((const uchar4 *)src)[i];
  uchar4 o = make_uchar4(0, 0, 0, 0);
  if (p.w != 0 && p.w != 255) { o.x = (uint8_t)((unsigned)p.x * 255u / p.w); o.y = (uint8_t)((unsigned)p.y * 255u / p.w); o.z = (uint8_t)((unsigned)p.z * 255u / p.w); o.w = p.w; }
  ((uchar4 *)dst)[i] = o;
}

// ---------------------------------------------------------------- dirty-alpha cleaner (ravif/src/dirtyalpha.rs:17-124)
// Three 3x3 stencil passes over RGBA8 in HBM, window clamped to the image (loop9 convention):
//   scan  : (256-a)-weighted mean colour of semi-transparent pixels that touch a fully transparent one  (:24-33)
//   bleed : bleed_opaque_color (:45-76)      blur : blur_transparent_pixels (:79-100)
// acc = { weights, sum_r, sum_g, sum_b } (u64).  weights == 0 means blurred_dirty_alpha returns None: both
// rewrite passes then degenerate to copies, so no host round trip is needed to decide.
__device__ __forceinline__ const uint8_t *rgba_at(const uint8_t *img, int w, int h, int x, int y) {
  x = x < 0 ? 0 : (x >= w ? w - 1 : x); y = y < 0 ? 0 : (y >= h ? h - 1 : y);
  return img + ((size_t)y * w + x) * 4;
}
__device__ __forceinline__ void premultiplied_minmax_dev(int px, int alpha, int *lo, int *hi) {   // :115-124, `as u8` wraps
  const int rounded = (px * alpha) / 255 * 255;
  const int low = ((rounded + 16) / alpha) & 255, high = ((rounded + 239) / alpha) & 255;
  *lo = imin_(low, px); *hi = imax_(high, px);
}
__global__ __launch_bounds__(256) void alpha_scan_kernel(const uint8_t *img, int w, int h, unsigned long long *acc) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  unsigned long long wt = 0, sr = 0, sg = 0, sb = 0;
  if (x < w && y < h) {
    const uint8_t *m = rgba_at(img, w, h, x, y);
    if (m[3] != 255 && m[3] != 0) {
      int any0 = 0;
      for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) any0 |= rgba_at(img, w, h, x + dx, y + dy)[3] == 0;
      if (any0) { wt = 256u - m[3]; sr = (unsigned long long)m[0] * wt; sg = (unsigned long long)m[1] * wt; sb = (unsigned long long)m[2] * wt; }
    }
  }
  // per-lane values are < 2^16: reduce in 32 bits per wave, then one 64-bit atomic per wave and component
  const int rw = wave_sum_i32((int)wt), rr = wave_sum_i32((int)sr), rg = wave_sum_i32((int)sg), rb = wave_sum_i32((int)sb);
  if (LANE == 0 && rw) { atomicAdd(acc + 0, (unsigned long long)rw); atomicAdd(acc + 1, (unsigned long long)rr); atomicAdd(acc + 2, (unsigned long long)rg); atomicAdd(acc + 3, (unsigned long long)rb); }
}
__global__ __launch_bounds__(256) void alpha_rewrite_kernel(const uint8_t *src, uint8_t *dst, int w, int h, const unsigned long long *acc, int pass) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const uint8_t *m = rgba_at(src, w, h, x, y);
  uint8_t *o = dst + ((size_t)y * w + x) * 4;
  const unsigned long long weights = acc[0];
  int r = m[0], g = m[1], b = m[2], a = m[3];
  if (weights != 0 && a != 255) {
    int avg[3]; bool use_bg = false;
    if (pass == 0) {                               // bleed_opaque_color
      unsigned int wsum = 0, s[3] = { 0, 0, 0 };
      for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        const uint8_t *c = rgba_at(src, w, h, x + dx, y + dy);
        if (c[3] == 0) continue;
        const unsigned int wt = 256u - c[3]; wsum += wt; s[0] += c[0] * wt; s[1] += c[1] * wt; s[2] += c[2] * wt;
      }
      if (wsum == 0) { use_bg = true; r = (int)(acc[1] / weights) & 255; g = (int)(acc[2] / weights) & 255; b = (int)(acc[3] / weights) & 255; a = 0; }
      else { avg[0] = (int)(s[0] / wsum) & 255; avg[1] = (int)(s[1] / wsum) & 255; avg[2] = (int)(s[2] / wsum) & 255; }
    } else {                                       // blur_transparent_pixels
      int s[3] = { 0, 0, 0 };
      for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) { const uint8_t *c = rgba_at(src, w, h, x + dx, y + dy); s[0] += c[0]; s[1] += c[1]; s[2] += c[2]; }
      avg[0] = s[0] / 9; avg[1] = s[1] / 9; avg[2] = s[2] / 9;
    }
    if (!use_bg) {
      if (a == 0) { r = avg[0]; g = avg[1]; b = avg[2]; }
      else {
        int lo, hi;
        premultiplied_minmax_dev(m[0], a, &lo, &hi); r = imin_(imax_(avg[0], lo), hi);
        premultiplied_minmax_dev(m[1], a, &lo, &hi); g = imin_(imax_(avg[1], lo), hi);
        premultiplied_minmax_dev(m[2], a, &lo, &hi); b = imin_(imax_(avg[2], lo), hi);
      }
    }
  }
  o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b; o[3] = (uint8_t)a;
}

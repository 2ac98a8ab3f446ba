// restoration.h -- kernel K5: loop restoration (spec 7.17), self-guided filter only (rav1e never searches Wiener).
// One workgroup (4 wavefronts) per (restoration unit, plane, frame).  Units are 64x64 for every plane (lr_unit_shift 0),
// i.e. one 64-row stripe of one superblock column; the last unit of a row / column absorbs a remainder < 32 samples, so a
// unit is walked as 1, 2 or 4 chunks of at most 64x64 samples that each lie inside one stripe.  Per chunk the source window
// (CDEF output inside the stripe, deblocked frame for the two rows above / below it, spec get_source_sample) is staged in
// LDS once; per parameter set the box sums -> (A, B) maps live in LDS, the two filtered values of a thread's 16 samples in
// registers.  Search = rav1e lrf.rs sgrproj_solve per set (least-squares projection weights, integer restatement shared
// with oracle/av1o_lrf.c) + RD choice against RESTORE_NONE (rdo.rs::rdo_loop_decision); then the winner is applied.
#pragma once
#include "dev_common.h"

#define LR_WP 70                                   // window pitch: 64 + 2 * 3
#define LR_AP 66                                   // (A, B) pitch: 64 + 2
struct LrLds {
  uint16_t win[LR_WP * LR_WP];
  uint16_t A[LR_AP * LR_AP];
  uint32_t B[LR_AP * LR_AP];
  uint16_t a2tab[256];
  long long red[4][6];
  long long tot[6];
  int xq[2];
};

__device__ __forceinline__ void sgr_param(int set, int *r0, int *s0, int *r1, int *s1) {
  const short t[16][4] = { { 2, 140, 1, 3236 }, { 2, 112, 1, 2158 }, { 2, 93, 1, 1618 }, { 2, 80, 1, 1438 }, { 2, 70, 1, 1295 }, { 2, 58, 1, 1177 },
                           { 2, 47, 1, 1079 }, { 2, 37, 1, 996 }, { 2, 30, 1, 925 }, { 2, 25, 1, 863 }, { 0, -1, 1, 2589 }, { 0, -1, 1, 1618 },
                           { 0, -1, 1, 1177 }, { 0, -1, 1, 925 }, { 2, 56, 0, -1 }, { 2, 22, 0, -1 } };
  *r0 = t[set][0]; *s0 = t[set][1]; *r1 = t[set][2]; *s1 = t[set][3];
}
__device__ __forceinline__ int lr_units_of(int size) { const int n = (size + 32) / 64; return n < 1 ? 1 : n; }

// bits of decode_signed_subexp_with_ref_bool(lo, hi_excl, k = 4, ref) for v; *bits = the MSB-first bit string
__device__ __forceinline__ int lr_recenter(int r, int x) { return x > 2 * r ? x : (x >= r ? (x - r) << 1 : ((r - x) << 1) - 1); }
__device__ inline int lr_subexp_code(int v, int lo, int hi_excl, int ref, uint32_t *bits) {
  const int mx = hi_excl - lo, x = v - lo, r = ref - lo;
  const int t = (r << 1) <= mx ? lr_recenter(r, x) : lr_recenter(mx - 1 - r, mx - 1 - x);
  uint32_t acc = 0; int nb = 0, i = 0, mk = 0;
  for (;;) {
    const int b2 = i ? 4 + i - 1 : 4, a = 1 << b2;
    if (mx <= mk + 3 * a) {
      const int nsy = mx - mk, val = t - mk;
      int w = 0; while ((1 << w) <= nsy) w++;
      const int m = (1 << w) - nsy;
      if (val < m) { acc = (acc << (w - 1)) | (uint32_t)val; nb += w - 1; }
      else { const int e = val + m; acc = (acc << (w - 1)) | (uint32_t)(e >> 1); acc = (acc << 1) | (uint32_t)(e & 1); nb += w; }
      break;
    }
    if (t >= mk + a) { acc = (acc << 1) | 1; nb++; i++; mk += a; }
    else { acc <<= 1; nb++; acc = (acc << b2) | (uint32_t)(t - mk); nb += b2; break; }
  }
  *bits = acc;
  return nb;
}

__device__ inline int lr_ratio_q7(long long num, long long det) {
  while (det >= (1LL << 54)) { det >>= 1; num >>= 1; }
  const int neg = num < 0; if (neg) num = -num;
  if (num >= det * 4) return neg ? -512 : 512;
  const long long q = (num * 128 + det / 2) / det;
  return (int)(neg ? -q : q);
}
__device__ inline void lr_sgr_solve(long long h00, long long h11, long long h01, long long c0, long long c1, int r0, int r1, int *xqd0, int *xqd1) {
  long long m = 0;
  const long long v[5] = { h00, h11, h01, c0, c1 };
  for (int i = 0; i < 5; i++) { const long long a = v[i] < 0 ? -v[i] : v[i]; if (a > m) m = a; }
  int sh = 0; while ((m >> sh) >= (1LL << 30)) sh++;
  h00 >>= sh; h11 >>= sh; h01 >>= sh; c0 >>= sh; c1 >>= sh;
  int xq0 = 0, xq1 = 0;
  if (r0 == 0) { if (h11 > 0) xq1 = lr_ratio_q7(c1, h11); }
  else if (r1 == 0) { if (h00 > 0) xq0 = lr_ratio_q7(c0, h00); }
  else {
    const long long det = h00 * h11 - h01 * h01;
    if (det > 0) { xq0 = lr_ratio_q7(h11 * c0 - h01 * c1, det); xq1 = lr_ratio_q7(h00 * c1 - h01 * c0, det); }
  }
  int x0 = iclamp_(xq0, -96, 31);
  int x1 = iclamp_(128 - x0 - xq1, -32, 95);
  if (r0 == 0) x0 = 0;
  if (r1 == 0) x1 = 95;
  *xqd0 = x0; *xqd1 = x1;
}
__device__ __forceinline__ int lr_project(int cdef, int f0, int f1, int r0, int r1, int w0, int w1, int mx) {
  const int u = cdef << 4, w2 = 128 - w0 - w1;
  const int v = w1 * u + w0 * (r0 ? f0 : u) + w2 * (r1 ? f1 : u);
  return iclamp_(round2_(v, 11), 0, mx);
}

struct LrChunk { int x0, y0, w, h, stripe_start, stripe_end; };

// stage the chunk's source window (origin x0 - 3, y0 - 3) in LDS
__device__ inline void lr_load_window(LrLds &L, const FrameDev *f, int plane, const LrChunk &c) {
  const int ex = f->w - 1, ey = f->h - 1, st = f->stride;
  const uint16_t *cdef = f->fin[plane], *dbk = f->rec[plane];
  const int ww = c.w + 6, wh = c.h + 6;
  for (int i = threadIdx.x; i < ww * wh; i += 256) {
    const int wy = ww == 70 ? i / 70 : i / ww, wx = i - wy * ww;
    const int x = iclamp_(c.x0 - 3 + wx, 0, ex); int y = iclamp_(c.y0 - 3 + wy, 0, ey);
    int v;
    if (y < c.stripe_start) { y = imax_(c.stripe_start - 2, y); v = dbk[(size_t)y * st + x]; }
    else if (y > c.stripe_end) { y = imin_(c.stripe_end + 2, y); v = dbk[(size_t)y * st + x]; }
    else v = cdef[(size_t)y * st + x];
    L.win[wy * LR_WP + wx] = (uint16_t)v;
  }
  __syncthreads();
}
// Box filter process of a chunk (spec 7.17.3) in two steps:
//   AB : the (sum of squares, sum) box sums of the (w + 2) x (h + 2) grid -> (A, B) of one parameter set (its s), into LDS.  Radius 2 (pass 0) weights rows of
//        odd parity only, so only those rows are computed.
//   F  : the 3x3 weighting of (A, B) -> filtered value of the thread's 16 samples.
// Both steps walk ROW SEGMENTS (round 4): a thread owns a run of neighbouring positions of one row and slides its window along it -- the box sums as running sums
// of column sums (5 + 5 window samples fetched per new position at radius 2 instead of 25, 3 instead of 9 at radius 1), the 3x3 weighting as running sums of the
// (A, B) columns (3 + 3 fetched per sample instead of 9 + 9).  The kernel was bound by the LDS pipeline (SQ_LDS_IDX_ACTIVE = every cycle of its 12.7 ms,
// profiles/r04_pmc_summary.json); the arithmetic -- integer sums -- and so every result is unchanged.
// A thread's 16 samples: row tid / 4 of the chunk, columns 16 * (tid % 4) .. + 15 (a chunk is at most 64 x 64).
// (Keeping the box sums in registers across the parameter sets of one unit was tried: 258 VGPRs, one wave per SIMD, no gain.)
#define LR_SEG 11                                  /* positions per box-sum task: 66 = 6 x 11 */
template <int R> __device__ inline void lr_box_AB(LrLds &L, const LrChunk &c, int sparam, int bd) {
  constexpr int n = (2 * R + 1) * (2 * R + 1), one_by_n = ((1 << 12) + n / 2) / n, NC = LR_SEG + 2 * R;
  const int s2 = 2 * (bd - 8), s1 = bd - 8;
  const int aw = c.w + 2, ah = c.h + 2;
  const int nrows = R == 2 ? (ah - (c.y0 & 1) + 1) >> 1 : ah, nseg = (aw + LR_SEG - 1) / LR_SEG;
  for (int task = threadIdx.x; task < nrows * nseg; task += 256) {
    const int row = task / nseg, seg = task - row * nseg;
    const int pi = R == 2 ? (c.y0 & 1) + 2 * row : row, pj0 = seg * LR_SEG;       // pi = i + 1, pj = j + 1
    const uint16_t *wp = L.win + (pi + 2 - R) * LR_WP + pj0 + 2 - R;          // top-left sample of the first position's box
    const int ncol = imin_(NC, aw - pj0 + 2 * R);                                  // window columns this segment touches (all inside the staged window)
    uint32_t cs[NC], cq[NC];
#pragma unroll
    for (int x = 0; x < NC; x++) {
      uint32_t a = 0, b = 0;
      if (x < ncol) {
#pragma unroll
        for (int dy = 0; dy <= 2 * R; dy++) { const uint32_t v = wp[dy * LR_WP + x]; a += v * v; b += v; }
      }
      cq[x] = a; cs[x] = b;
    }
#pragma unroll
    for (int j = 0; j < LR_SEG; j++) {
      if (pj0 + j < aw) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int dx = 0; dx <= 2 * R; dx++) { a += cq[j + dx]; b += cs[j + dx]; }
        const uint32_t ar = s2 ? (a + (1u << (s2 - 1))) >> s2 : a, d = s1 ? (b + (1u << (s1 - 1))) >> s1 : b;
        const uint32_t p = ar * (uint32_t)n > d * d ? ar * (uint32_t)n - d * d : 0;
        const uint32_t z = (uint32_t)(((unsigned long long)p * (unsigned)sparam + (1u << 19)) >> 20);
        const uint32_t a2 = z >= 255 ? 256 : L.a2tab[z];
        const uint32_t b2 = (256 - a2) * b * (uint32_t)one_by_n;
        L.A[pi * LR_AP + pj0 + j] = (uint16_t)a2; L.B[pi * LR_AP + pj0 + j] = (b2 + (1u << 11)) >> 12;
      }
    }
  }
  __syncthreads();
}
// flt[k]: the two filters' outputs of sample k, packed (pass 0 in the low, pass 1 in the high 16 bits: a filtered sample in Q4 is a convex combination of
// samples << 4, at most 16 368 + rounding for 10-bit input) -- sixteen registers per thread instead of thirty-two across the least-squares solve
__device__ __forceinline__ int lr_flt0(int v) { return (int)(short)(v & 0xFFFF); }
__device__ __forceinline__ int lr_flt1(int v) { return v >> 16; }
__device__ __forceinline__ int lr_row_of_thread() { return (int)(threadIdx.x >> 2); }
__device__ __forceinline__ int lr_col_of_thread() { return (int)(threadIdx.x & 3) * 16; }
__device__ inline void lr_box_F(LrLds &L, const LrChunk &c, int pass, int flt[16]) {
  const int py = lr_row_of_thread(), px0 = lr_col_of_thread(), yabs = c.y0 + py;
  const bool rowok = py < c.h;
  // column terms of the 3x3 weighting at (A, B) column x (x = px + 1): `m` = the sample's own row, `u` = the rows above + below
  const int o0 = (py + 1) * LR_AP + px0;                                           // column px0 - 1 + 1 of the sample's row
  const bool odd = (yabs & 1) != 0;
  int am[3] = { 0, 0, 0 }, au[3] = { 0, 0, 0 }, bm[3] = { 0, 0, 0 }, bu[3] = { 0, 0, 0 };
  auto fetch = [&](int x, int slot) {                                              // x: offset from o0
    if (pass == 0) {
      if (odd) { am[slot] = L.A[o0 + x]; bm[slot] = (int)L.B[o0 + x]; }
      else { au[slot] = L.A[o0 + x - LR_AP] + L.A[o0 + x + LR_AP]; bu[slot] = (int)(L.B[o0 + x - LR_AP] + L.B[o0 + x + LR_AP]); }
    } else {
      am[slot] = L.A[o0 + x]; bm[slot] = (int)L.B[o0 + x];
      au[slot] = L.A[o0 + x - LR_AP] + L.A[o0 + x + LR_AP]; bu[slot] = (int)(L.B[o0 + x - LR_AP] + L.B[o0 + x + LR_AP]);
    }
  };
  if (rowok && px0 < c.w) { fetch(0, 0); fetch(1, 1); }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    int out = 0;
    const int px = px0 + k;
    if (rowok && px < c.w) {
      const int l = k % 3, m = (k + 1) % 3, r = (k + 2) % 3;                       // slots of columns px - 1, px, px + 1
      fetch(k + 2, r);
      int a, b, shift;
      if (pass == 0) {
        if (odd) { a = 6 * am[m] + 5 * (am[l] + am[r]); b = 6 * bm[m] + 5 * (bm[l] + bm[r]); shift = 4; }
        else { a = 6 * au[m] + 5 * (au[l] + au[r]); b = 6 * bu[m] + 5 * (bu[l] + bu[r]); shift = 5; }
      } else {
        a = 4 * (am[m] + am[l] + am[r] + au[m]) + 3 * (au[l] + au[r]);
        b = 4 * (bm[m] + bm[l] + bm[r] + bu[m]) + 3 * (bu[l] + bu[r]);
        shift = 5;
      }
      const int cd = L.win[(py + 3) * LR_WP + px + 3];
      out = round2_(a * cd + b, 4 + shift);
    }
    flt[k] = pass == 0 ? (out & 0xFFFF) : ((flt[k] & 0xFFFF) | (int)((uint32_t)out << 16));
  }
  __syncthreads();
}

// block-wide sum of up to 6 per-thread 64-bit values (|v| < 2^40) -> L.tot[], visible to all threads after return
__device__ inline void lr_block_sum(LrLds &L, const long long *v, int n) {
  const int wave = threadIdx.x >> 6;
  for (int i = 0; i < n; i++) { const long long s = wave_sum_i64(v[i]); if (LANE == 0) L.red[wave][i] = s; }
  __syncthreads();
  if (threadIdx.x < n) L.tot[threadIdx.x] = L.red[0][threadIdx.x] + L.red[1][threadIdx.x] + L.red[2][threadIdx.x] + L.red[3][threadIdx.x];
  __syncthreads();
}

__device__ __forceinline__ int lr_chunks(const FrameDev *f, int ur, int uc, int ucols, int urows, LrChunk *ch) {
  const int W = f->w, H = f->h;
  const int x0 = uc * 64, x1 = uc == ucols - 1 ? W : x0 + 64;
  const int y0 = imax_(0, ur * 64 - 8), y1 = ur == urows - 1 ? H : ur * 64 + 56;
  int n = 0;
  for (int ys = y0; ys < y1;) {
    const int stripe = (ys + 8) / 64, ye = imin_(y1, stripe * 64 + 56);
    for (int xs = x0; xs < x1; xs += 64) {
      LrChunk c; c.x0 = xs; c.w = imin_(64, x1 - xs); c.y0 = ys; c.h = ye - ys; c.stripe_start = stripe * 64 - 8; c.stripe_end = c.stripe_start + 63;
      ch[n++] = c;
    }
    ys = ye;
  }
  return n;
}

// The search: (unit, plane) x every parameter set of the frame's list -> least-squares weights, activity-scaled SSE, RD cost per set.
// grid = (units, planes, frames).  Results go to f->lr_cand[(plane * units + unit) * 16 + set index].  One workgroup walks the sets one after the other on ONE staged
// window (until round 5 every set had its own workgroup, which staged the same window and read the same source samples again: 4 x the planes' bytes per launch).
struct LrCand { long long cost; int xq0, xq1; };
__global__ __launch_bounds__(256, 4) void lr_search_kernel(const FrameDev *__restrict__ frames) {
  const FrameDev *f = frames + blockIdx.z;
  const int nsets = f->sgr_full ? 16 : 4;
  const int plane = blockIdx.y;
  if (plane >= f->np || !f->enable_restoration || frame_idle(f)) return;
  const int ucols = lr_units_of(f->w), urows = lr_units_of(f->h);
  const int ui = blockIdx.x;
  if (ui >= ucols * urows) return;
  const int ur = ui / ucols, uc = ui - ur * ucols;
  __shared__ LrLds L;
  L.a2tab[threadIdx.x] = (uint16_t)(threadIdx.x == 0 ? 1 : ((threadIdx.x << 8) + threadIdx.x / 2) / (threadIdx.x + 1));
  // the unit's chunks live in LDS: a per-thread array indexed at run time is a scratch (HBM-backed) array -- every lane wrote and re-read its own copy of these
  // wave-uniform 96 bytes, which was most of the kernel's memory traffic in round 4 (13.9 GB per launch for 0.4 GB of planes)
  __shared__ LrChunk ch[4]; __shared__ int nch_s;
  if (threadIdx.x == 0) nch_s = lr_chunks(f, ur, uc, ucols, urows, ch);
  const int st = f->stride, bd = f->bd, mx = (1 << bd) - 1;
  const uint16_t *src = f->src[plane];
  const int reduced[4] = { 1, 3, 6, 11 };
  __syncthreads();
  const int nch = nch_s;
#pragma unroll 1
  for (int si = 0; si < nsets; si++) {
  const int set = f->sgr_full ? si : reduced[si & 3];
  int r0, s0, r1, s1; sgr_param(set, &r0, &s0, &r1, &s1);
  int flt[16];
  for (int k = 0; k < 16; k++) flt[k] = 0;
  long long acc[6];
  int xq0 = 0, xq1 = 0;
  // sweep 0: normal equations; sweep 1: SSE with the solved weights
  for (int sweep = 0; sweep < 2; sweep++) {
    for (int i = 0; i < 6; i++) acc[i] = 0;
    for (int q = 0; q < nch; q++) {
      const LrChunk c = ch[q];
      if (sweep == 0 || nch > 1) {
        if (si == 0 || nch > 1) lr_load_window(L, f, plane, c);       // a one-chunk unit's window stays staged for all its sets
        if (r0) { lr_box_AB<2>(L, c, s0, bd); lr_box_F(L, c, 0, flt); }
        if (r1) { lr_box_AB<1>(L, c, s1, bd); lr_box_F(L, c, 1, flt); }
      }
      const int py = lr_row_of_thread(), px0 = lr_col_of_thread();
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int px = px0 + k;
        if (py < c.h && px < c.w) {
          const size_t o = (size_t)(c.y0 + py) * st + c.x0 + px;
          const int cd = L.win[(py + 3) * LR_WP + px + 3], sv = src[o];
          if (sweep == 0) {
            const int u = cd << 4, e = (sv << 4) - u;
            const int f0 = r0 ? lr_flt0(flt[k]) - u : 0, f1 = r1 ? lr_flt1(flt[k]) - u : 0;
            acc[0] += (long long)(f0 * f0); acc[1] += (long long)(f1 * f1); acc[2] += (long long)(f0 * f1); acc[3] += (long long)(f0 * e); acc[4] += (long long)(f1 * e);
          } else {
            const int d = lr_project(cd, lr_flt0(flt[k]), lr_flt1(flt[k]), r0, r1, xq0, xq1, mx) - sv;
            acc[5] += (long long)(d * d);
          }
        }
      }
      if (nch > 1) __syncthreads();                            // the window is re-staged for the next chunk
    }
    if (sweep == 0) {
      lr_block_sum(L, acc, 5);
      if (threadIdx.x == 0) { int a, b; lr_sgr_solve(L.tot[0], L.tot[1], L.tot[2], L.tot[3], L.tot[4], r0, r1, &a, &b); L.xq[0] = a; L.xq[1] = b; }
      __syncthreads();
      xq0 = L.xq[0]; xq1 = L.xq[1];
    } else {
      // the unit's mean activity scale (Q14) over the 8x8 cells it covers scales every distortion of the unit
      const int ux0 = uc * 64, ux1 = uc == ucols - 1 ? f->w : ux0 + 64, uy0 = imax_(0, ur * 64 - 8), uy1 = ur == urows - 1 ? f->h : ur * 64 + 56;
      const int cx0 = ux0 >> 3, cx1 = (ux1 - 1) >> 3, cy0 = uy0 >> 3, cy1 = (uy1 - 1) >> 3, ncx = cx1 - cx0 + 1, cnt = ncx * (cy1 - cy0 + 1);
      long long a = 0;
      for (int i = threadIdx.x; i < cnt; i += 256) a += f->act[(cy0 + i / ncx) * (f->pw >> 3) + cx0 + i % ncx];
      acc[4] = a;
      lr_block_sum(L, acc + 4, 2);                             // tot[0] = activity sum, tot[1] = SSE
      if (threadIdx.x == 0) {
        const long long unit_act = (L.tot[0] + cnt / 2) / cnt;
        uint32_t rate = f->lr_cost[2] + 4 * 512, bits;
        if (r0) rate += 512u * (uint32_t)lr_subexp_code(xq0, -96, 32, -32, &bits);
        if (r1) rate += 512u * (uint32_t)lr_subexp_code(xq1, -32, 96, 31, &bits);
        LrCand cnd;
        cnd.cost = ((((L.tot[1] * unit_act + 8192) >> 14) * f->wq[plane]) >> 5) + (((long long)rate * f->rdmult + 256) >> 9);
        cnd.xq0 = xq0; cnd.xq1 = xq1;
        ((LrCand *)f->lr_cand)[(size_t)(plane * ucols * urows + ui) * 16 + si] = cnd;
      }
    }
  }
  __syncthreads();                                             // the next set reuses the (A, B) maps and the reduction slots
  }
}

// RD choice between RESTORE_NONE and the searched sets (first minimum in set order), then the winner is applied.
// grid = (units, planes, frames)
__global__ __launch_bounds__(256) void lr_kernel(const FrameDev *__restrict__ frames) {
  const FrameDev *f = frames + blockIdx.z;
  const int plane = blockIdx.y;
  if (plane >= f->np || !f->enable_restoration || frame_idle(f)) return;
  const int ucols = lr_units_of(f->w), urows = lr_units_of(f->h);
  const int ui = blockIdx.x;
  if (ui >= ucols * urows) return;
  const int ur = ui / ucols, uc = ui - ur * ucols;
  __shared__ LrLds L;
  L.a2tab[threadIdx.x] = (uint16_t)(threadIdx.x == 0 ? 1 : ((threadIdx.x << 8) + threadIdx.x / 2) / (threadIdx.x + 1));
  __shared__ LrChunk ch[4]; __shared__ int nch_s;                    // (LDS, not a per-thread array: see lr_search_kernel)
  if (threadIdx.x == 0) nch_s = lr_chunks(f, ur, uc, ucols, urows, ch);
  const int st = f->stride, bd = f->bd, mx = (1 << bd) - 1;
  const uint16_t *cdef = f->fin[plane], *src = f->src[plane];
  uint16_t *out = f->lrp[plane];
  __syncthreads();
  const int nch = nch_s;
  // RESTORE_NONE
  long long acc[2];
  {
    int s = 0;
    for (int q = 0; q < nch; q++) {
      const LrChunk c = ch[q];
      for (int idx = threadIdx.x; idx < c.w * c.h; idx += 256) {
        const int py = c.w == 64 ? idx >> 6 : idx / c.w, px = idx - py * c.w; const size_t o = (size_t)(c.y0 + py) * st + c.x0 + px;
        const int d = (int)cdef[o] - (int)src[o]; s += d * d;
      }
    }
    acc[0] = s;
    const int ux0 = uc * 64, ux1 = uc == ucols - 1 ? f->w : ux0 + 64, uy0 = imax_(0, ur * 64 - 8), uy1 = ur == urows - 1 ? f->h : ur * 64 + 56;
    const int cx0 = ux0 >> 3, cx1 = (ux1 - 1) >> 3, cy0 = uy0 >> 3, cy1 = (uy1 - 1) >> 3, ncx = cx1 - cx0 + 1, cnt = ncx * (cy1 - cy0 + 1);
    long long a = 0;
    for (int i = threadIdx.x; i < cnt; i += 256) a += f->act[(cy0 + i / ncx) * (f->pw >> 3) + cx0 + i % ncx];
    acc[1] = a;
    lr_block_sum(L, acc, 2);
    if (threadIdx.x == 0) {
      const long long unit_act = (L.tot[1] + cnt / 2) / cnt;
      long long best_cost = ((((L.tot[0] * unit_act + 8192) >> 14) * f->wq[plane]) >> 5) + (((long long)f->lr_cost[0] * f->rdmult + 256) >> 9);
      int best_type = 0, best_set = 0, bx0 = 0, bx1 = 0;
      const int nsets = f->sgr_full ? 16 : 4;
      const int reduced[4] = { 1, 3, 6, 11 };
      const LrCand *cn = (const LrCand *)f->lr_cand + (size_t)(plane * ucols * urows + ui) * 16;
      for (int si = 0; si < nsets; si++) if (cn[si].cost < best_cost) { best_cost = cn[si].cost; best_type = 1; best_set = f->sgr_full ? si : reduced[si]; bx0 = cn[si].xq0; bx1 = cn[si].xq1; }
      const int n = ucols * urows;
      f->lr_type[plane * n + ui] = (uint8_t)best_type; f->lr_set[plane * n + ui] = (uint8_t)best_set;
      f->lr_xqd[(plane * n + ui) * 2] = (int8_t)bx0; f->lr_xqd[(plane * n + ui) * 2 + 1] = (int8_t)bx1;
      L.xq[0] = bx0; L.xq[1] = bx1; L.red[0][0] = best_type; L.red[0][1] = best_set;
    }
    __syncthreads();
  }
  const int best_type = (int)L.red[0][0], best_set = (int)L.red[0][1], xq0 = L.xq[0], xq1 = L.xq[1];
  __syncthreads();
  if (!best_type) {
    for (int q = 0; q < nch; q++) {
      const LrChunk c = ch[q];
      for (int idx = threadIdx.x; idx < c.w * c.h; idx += 256) { const int py = c.w == 64 ? idx >> 6 : idx / c.w, px = idx - py * c.w; const size_t o = (size_t)(c.y0 + py) * st + c.x0 + px; out[o] = cdef[o]; }
    }
    return;
  }
  int r0, s0, r1, s1; sgr_param(best_set, &r0, &s0, &r1, &s1);
  int flt[16];
  for (int k = 0; k < 16; k++) flt[k] = 0;
  for (int q = 0; q < nch; q++) {
    const LrChunk c = ch[q];
    lr_load_window(L, f, plane, c);
    if (r0) { lr_box_AB<2>(L, c, s0, bd); lr_box_F(L, c, 0, flt); }
    if (r1) { lr_box_AB<1>(L, c, s1, bd); lr_box_F(L, c, 1, flt); }
    const int py = lr_row_of_thread(), px0 = lr_col_of_thread();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int px = px0 + k;
      if (py < c.h && px < c.w) {
        const size_t o = (size_t)(c.y0 + py) * st + c.x0 + px;
        out[o] = (uint16_t)lr_project(L.win[(py + 3) * LR_WP + px + 3], lr_flt0(flt[k]), lr_flt1(flt[k]), r0, r1, xq0, xq1, mx);
      }
    }
    if (nch > 1) __syncthreads();
  }
}

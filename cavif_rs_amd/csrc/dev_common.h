// dev_common.h -- device-side frame descriptor and wave helpers for the MI355X AV1 intra path.
// One wavefront (64 lanes) owns one tile during the search and entropy-coding kernels; all
// cross-lane traffic is wave shuffles or that wave's own LDS region.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "av1_tables.h"

// rav1e rdo_tx_size_type searches tx_depth 0..2 (rdo_tx_depth = 2) when rdo_tx_decision is on; 1 = one level only (oracle AV1O_TX_DEPTH_MAX)
#define MI_TX_DEPTH_MAX 2
// Block / transform size codes 5 (4x8) and 6 (8x4) appear in m_bsize / m_txsize: PARTITION_HORZ / PARTITION_VERT of 8x8 nodes (dev_rect.h).
#define MI_MAX_TILE_COLS 64
#define MI_MAX_TILE_ROWS 64

enum { BS_4 = 0, BS_8 = 1, BS_16 = 2, BS_32 = 3, BS_64 = 4 };
enum { DC_PRED = 0, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED,
       SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED, UV_CFL_PRED };
enum { DCT_DCT = 0, ADST_DCT, DCT_ADST, ADST_ADST, FLIPADST_DCT, DCT_FLIPADST, FLIPADST_FLIPADST,
       ADST_FLIPADST, FLIPADST_ADST, IDTX, V_DCT, H_DCT, V_ADST, H_ADST, V_FLIPADST, H_FLIPADST };
enum { TXC_2D = 0, TXC_HORIZ = 1, TXC_VERT = 2 };

// Segmentation (segment_kernel in loopfilter.h; oracle/av1o_segment.c): what the frame's activity scales were fitted to.  n = 0: off.
// Segment i: q[i] = its quantiser steps and reciprocals per plane, laid out like FrameDev::dc_q .. ac_recip (the tile search copies
// a block's entry over those fields of its LDS copy).  thr[]: scale-bucket thresholds between the n clusters (ascending).
#define MI_SEG_BINS 4096
struct SegTab {
  int n, mean, thr[7], qidx[8], pad_[3];
  struct Q { int dc_q[3], ac_q[3]; uint32_t dc_recip[3], ac_recip[3]; } q[8];
};
// Everything a kernel needs to know about one plane-set being encoded (one AV1 frame: the colour
// image or the alpha plane of one input image).  Lives in device memory; pointers are device pointers.
struct FrameDev {
  // ---- head: what the tile search (K1) reads; this part is staged in LDS per tile (FRAMEDEV_K1_BYTES) ----
  int w, h, bd, np;
  int mi_cols, mi_rows, sb_cols, sb_rows;
  int pw, ph, stride, mi_stride, mi_h;
  uint16_t *src[3], *rec[3];               // source, in-loop reconstruction (deblocked in place)
  int32_t *coef[3];
  uint8_t *m_bsize, *m_skip, *m_ymode, *m_uvmode, *m_txtype, *m_cfl_sign, *m_cfl_au, *m_cfl_av, *m_decoded;   // m_skip: bit 0 = skip, bits 1.. = segment id
  uint8_t *m_txsize;                       // luma transform size of the block (0 = 4x4 .. 4 = 64x64), uniform over the block
  int8_t *m_angle_y, *m_angle_uv;
  uint8_t *m_lvl[3], *m_dc[3];
  uint16_t *m_eob[3];
  // quantizer / lambda
  int base_q_idx, qctx, dc_q[3], ac_q[3];
  uint32_t dc_recip[3], ac_recip[3];          // floor((2^32 - 1) / q): quantisation divides by multiply-high + one fix-up
  long long rdmult, wq[3];
  // tools
  int part_min, part_max, complex_modes, fine_directional, rdo_tx, reduced_tx_set, tx_mode_select, bottomup;
  int tile_cols;
  // Tune::Psychovisual: per 8x8 cell activity scale (Q14) and source variance, per 4x4 source variance (8x8-equivalent)
  const uint32_t *act, *svar8, *svar4; int tune_psnr;
  int seg_n;                 // segments in use (0 = segmentation off); written by segment_kernel, like *seg
  const SegTab *seg;
  int *sb_prog;              // K1 work queue: superblocks finished per (frame SB row, tile column), then one word of finished-root bits per superblock, then the frame's
                             // sticky error word (search_error_word): a wait that gave up; all zeroed before every encode
  int dbg;                   // debug bisect level (0 = off; probe builds only)
  const uint16_t *cost;      // static rate table [CDF_TOTAL] (cost per symbol in 1/512 bit, same flat layout as the CDF context)
  // ---- tail: frame-level stages and the entropy coder ----
  unsigned long long *prof_out;  // profiling builds: per launch-wide tile job, 4 waves x 16 phase cycle counters
  unsigned long long *tile_clk;  // per tile: [start, end] of K1 and of K4 in wall_clock64 ticks (100 MHz), 4 values
  uint16_t *fin[3];                        // post-CDEF output
  uint16_t *lrp[3];                        // post-loop-restoration output (the final picture when enable_restoration)
  uint8_t *lr_type, *lr_set; int8_t *lr_xqd;   // per (plane, restoration unit): 0 none / 1 sgrproj, parameter set, xqd[2]
  void *lr_cand;                           // search scratch: per (plane, unit, set) { cost, xqd } (restoration.h LrCand, 16 per unit)
  uint32_t lr_cost[3];                     // static cost of the switchable restoration_type symbols (1/512 bit)
  int enable_restoration, sgr_full, enable_cdef, fast_deblock;
  int8_t *cdef_idx;
  // tiles (SB units)
  int tile_rows, tile_cols_log2, tile_rows_log2;
  int tile_col_start[MI_MAX_TILE_COLS + 1], tile_row_start[MI_MAX_TILE_ROWS + 1];
  const uint16_t *cdf0;      // initial CDFs [CDF_TOTAL]
  // two-pass pricing (mi_av1_config.rdo_passes = 2): pass 1's entropy coder leaves every tile's final CDFs in cdf_out, cdf_cost_kernel turns them
  // into per-tile rate tables, and pass 2's tile search prices tile t against tile_cost + t * CDF_TOTAL.  Both null in a one-pass encode.
  const uint16_t *tile_cost; uint16_t *cdf_out; uint16_t *tile_cost_buf;
  // loop filter / cdef
  int lf_level[4], lf_sharp, cdef_damping, cdef_bits, cdef_y[8], cdef_uv[8];
  int seg_ddc[3], seg_dac[3];   // the frame header's plane deltas (dc / ac index of plane p minus base_q_idx): a segment's steps are looked up at its index + these
  long long *lf_tally;       // deblock level search: [3 planes][2 passes][65] SSE-delta difference arrays (zeroed per encode)
  int zero_words;            // 32-bit words of the block starting at m_decoded (decoded flags, lf_tally, sb_prog) that every encode starts from zero: the activity kernel clears it
  int *lf_out;               // the frame's 4 chosen levels, read back by the host for the frame header
  // per-tile outputs
  uint8_t *tile_out;         // per tile: tile_out_cap bytes
  uint32_t *tile_len;        // per tile
  uint32_t tile_out_cap;
  int tile_base;             // index of this frame's first tile in the launch-wide tile list
  // Alpha frames of an RGBA batch are planned before the front end has reported whether the image uses its alpha channel
  // (ravif/src/av1encoder.rs:246): `active` points at the image's flag, and every stage of an unused alpha frame returns at
  // once (no host round trip between the front end and the tile search).  nullptr: always active.
  const int *active;
};
template <typename FP> __device__ __forceinline__ int *search_error_word(FP f) { return f->sb_prog + f->sb_rows * f->tile_cols + f->sb_rows * f->sb_cols; }
__device__ __forceinline__ bool frame_idle(const FrameDev *f) { return f->active != nullptr && *f->active == 0; }
#define FRAMEDEV_K1_BYTES ((int)(offsetof(FrameDev, fin) + 15) & ~15)

struct TileJob { int frame; int tile_row, tile_col; };

#define LANE ((int)(threadIdx.x & 63))
// Explicit LDS address space: pointers carrying it compile to ds_read/ds_write instead of flat_* accesses.
#ifndef LDS                                      /* (the CPU test harness tests/emu/ predefines it empty) */
#define LDS __attribute__((address_space(3)))
#endif

__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int iclamp_(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs_(int a) { return a < 0 ? -a : a; }
__device__ __forceinline__ int round2_(int x, int n) { return n == 0 ? x : (x + (1 << (n - 1))) >> n; }

// run-time indexed small tables without a memory lookup: 4-bit / 8-bit entries packed into a 64-bit immediate
__device__ __forceinline__ int lut4(unsigned long long tab, int i) { return (int)((tab >> (4 * i)) & 15); }
__device__ __forceinline__ int lut8(unsigned long long tab, int i) { return (int)((tab >> (8 * i)) & 255); }
// nominal prediction angle of the directional modes V, H, D45, D135, D113, D157, D203, D67 (mode 1..8; 0 for DC)
__device__ __forceinline__ int mode_angle_of(int m) { return m == 8 ? 67 : lut8(0xcb9d71872db45a00ULL, m); }
// above / left intra mode -> KF y-mode context (spec Intra_Mode_Context)
__device__ __forceinline__ int intra_mode_ctx(int m) { return lut4(0x210344443210ULL, m); }

// Wave-uniform values the compiler cannot prove uniform (results of calls, values loaded from memory): pin them to SGPRs.  What
// stays live across a call to the (no-callee-saved-registers) block search is then kept by v_writelane instead of scratch.
__device__ __forceinline__ int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
  return (long long)(((unsigned long long)(uint32_t)uni32((int)((unsigned long long)v >> 32)) << 32) | (uint32_t)uni32((int)v));
}

// Wave-wide reductions on the DPP network (no LDS round trips): quad_perm, row_half_mirror, row_mirror,
// row_bcast:15, row_bcast:31; the total lands in lane 63 and is broadcast through an SGPR.
#define DPP_(old, v, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rmask), 0xF, false)
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += DPP_(0, v, 0xB1, 0xF); v += DPP_(0, v, 0x4E, 0xF); v += DPP_(0, v, 0x141, 0xF); v += DPP_(0, v, 0x140, 0xF);
  v += DPP_(0, v, 0x142, 0xA); v += DPP_(0, v, 0x143, 0xC);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {
  v = imax_(v, DPP_(v, v, 0xB1, 0xF)); v = imax_(v, DPP_(v, v, 0x4E, 0xF)); v = imax_(v, DPP_(v, v, 0x141, 0xF)); v = imax_(v, DPP_(v, v, 0x140, 0xF));
  v = imax_(v, DPP_(v, v, 0x142, 0xA)); v = imax_(v, DPP_(v, v, 0x143, 0xC));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_or_i32(int v) {
  v |= DPP_(0, v, 0xB1, 0xF); v |= DPP_(0, v, 0x4E, 0xF); v |= DPP_(0, v, 0x141, 0xF); v |= DPP_(0, v, 0x140, 0xF);
  v |= DPP_(0, v, 0x142, 0xA); v |= DPP_(0, v, 0x143, 0xC);
  return __builtin_amdgcn_readlane(v, 63);
}
// 64-bit sum as two carry-free 32-bit sums of the 24-bit-split halves (each lane's partial must be < 2^40)
__device__ __forceinline__ long long wave_sum_i64(long long v) {
  const int lo = wave_sum_i32((int)(v & 0xFFFFFF)), hi = wave_sum_i32((int)(v >> 24));
  return ((long long)hi << 24) + (long long)lo;
}
// WAVE_SYNC: orders one wavefront's own LDS/global traffic between phases (no cross-wave rendezvous);
// WG_SYNC: workgroup barrier, used where the waves of a tile exchange results.
// Lanes of ONE wavefront exchanging data through LDS: the LDS unit executes a wave's instructions in issue order, so all that is needed is
// that the compiler keeps the order (wavefront-scope fence + scheduling barrier).  A workgroup-scope fence here would drain every
// outstanding global load and store of the wave (s_waitcnt vmcnt(0)) at each of the thousands of exchanges per block.
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define WG_SYNC() __syncthreads()

struct TileB { int mi_row_start, mi_row_end, mi_col_start, mi_col_end; };

// ---- segmentation helpers (oracle/av1o_segment.c: av1o_ilog2_q11, av1o_seg_bucket, av1o_seg_pred, av1o_seg_symbol; spec 5.11.9)
__device__ __forceinline__ int seg_ilog2_q11(uint32_t x) {
  if (x == 0) x = 1;
  const int msb = 31 - __clz((int)x);
  unsigned long long m = (unsigned long long)x << (31 - msb);
  int frac = 0;
#pragma unroll
  for (int i = 0; i < 11; i++) { m = (m * m) >> 31; frac <<= 1; if (m >> 32) { frac |= 1; m >>= 1; } }
  return (msb << 11) | frac;
}
__device__ __forceinline__ int seg_bucket(uint32_t scale_q14) { return iclamp_((seg_ilog2_q11(scale_q14) - (6 << 11)) >> 3, 0, MI_SEG_BINS - 1); }
// neighbours' ids (-1 = outside the tile) -> predicted id, CDF context
__device__ __forceinline__ int seg_pred(int ul, int u, int l, int *ctx) {
  *ctx = ul < 0 ? 0 : ((ul == u && ul == l) ? 2 : ((ul == u || ul == l || u == l) ? 1 : 0));
  if (u == -1) return l == -1 ? 0 : l;
  if (l == -1) return u;
  return ul == u ? u : l;
}
__device__ __forceinline__ int seg_neg_deinterleave(int diff, int ref, int max) {
  if (!ref) return diff;
  if (ref >= max - 1) return max - diff - 1;
  if (2 * ref < max) { if (diff <= 2 * ref) return (diff & 1) ? ref + ((diff + 1) >> 1) : ref - (diff >> 1); return diff; }
  if (diff <= 2 * (max - ref - 1)) return (diff & 1) ? ref + ((diff + 1) >> 1) : ref - (diff >> 1);
  return max - (diff + 1);
}
__device__ __forceinline__ int seg_symbol(int seg, int pred, int max) {     // the symbol that decodes to `seg`
  int d = 0;
  for (int k = 0; k < 8; k++) if (k < max && seg_neg_deinterleave(k, pred, max) == seg) d = k;
  return d;
}

// ---- Tune::Psychovisual helpers (oracle/av1o_common.c av1o_psy_boost_q14 / av1o_cell_var; rav1e dist.rs cdef_dist_kernel,
// activity.rs, recalled): integer, bit-identical to the CPU restatement.
// Both roots and quotients are float GUESSES made exact by integer correction, so the guesses may be cheap: the operands go to float directly (an fma instead of a 64-bit
// integer -> float conversion), the square root and the reciprocal are the raw hardware approximations (1 ulp; an IEEE-correct sqrtf / division is a ten-instruction
// sequence each) -- whatever they return, the while loops leave floor(sqrt(n)) and floor(t / den).
__device__ __forceinline__ uint32_t psy_isqrt46(unsigned long long n, float nf) {       // floor(sqrt(n)), n < 2^46; nf ~ n
  uint32_t x = (uint32_t)__builtin_amdgcn_sqrtf(nf);
  while ((unsigned long long)x * x > n) x--;
  while ((unsigned long long)(x + 1) * (x + 1) <= n) x++;
  return x;
}
__device__ inline uint32_t psy_boost_q14(uint32_t sv, uint32_t dv) {
  const unsigned long long num = 4033ull * ((unsigned long long)sv + dv + 16384);
  const uint32_t den = psy_isqrt46(16265089ull + (unsigned long long)sv * dv, __builtin_fmaf((float)sv, (float)dv, 16265089.0f));
  const unsigned long long t = num + den / 2;
  const float tf = __builtin_fmaf(4033.0f, (float)sv + (float)dv + 16384.0f, (float)(den >> 1));
  uint32_t q = (uint32_t)(tf * __builtin_amdgcn_rcpf((float)den));                    // guess, then exact
  while ((unsigned long long)q * den > t) q--;
  while ((unsigned long long)(q + 1) * den <= t) q++;
  return q;
}
// 64 x variance of a w x w cell (w = 8 or 4, 4x4 scaled to the 8x8 equivalent) on the 8-bit scale
__device__ __forceinline__ uint32_t psy_cell_var(uint32_t sum, uint32_t sum2, int w, int bd) {
  int v = w == 8 ? (int)sum2 - (int)((sum * sum + 32u) >> 6) : ((int)sum2 - (int)((sum * sum + 8u) >> 4)) << 2;
  if (v < 0) v = 0;
  return (uint32_t)v >> (2 * (bd - 8));
}
// distortion of one cell: SSE boosted by the SSIM-like factor, then by the cell's activity scale
__device__ __forceinline__ int psy_cell_dist(uint32_t sse, uint32_t sd, uint32_t qd, uint32_t svar, uint32_t act, int w, int bd) {
  const uint32_t b = psy_boost_q14(svar, psy_cell_var(sd, qd, w, bd));
  unsigned long long d = ((unsigned long long)sse * b + 8192) >> 14;
  d = (d * act + 8192) >> 14;
  return (int)d;
}

// RD cost terms of a candidate whose distortion fits 32 bits (grouped evaluations: blocks up to 8x8): u32 x 64-bit products (v_mad_u64_u32 + one 32-bit multiply)
// where `(long long)int * long long` is a full 64 x 64 multiply behind a sign extension.  Same values: both factors are non-negative and the products stay below 2^63.
template <typename FP> __device__ __forceinline__ long long rd_dist32(FP f, int plane, int sse) { return (long long)(((unsigned long long)(uint32_t)sse * (unsigned long long)f->wq[plane]) >> 5); }
template <typename FP> __device__ __forceinline__ long long rd_rate32(FP f, uint32_t rate) { return (long long)(((unsigned long long)rate * (unsigned long long)f->rdmult + 256) >> 9); }
__device__ __forceinline__ int tx_set_of(int txs, int reduced) { return txs >= 3 ? 0 : (reduced ? 2 : (txs == 2 ? 2 : 1)); }
__device__ __forceinline__ int tx_set_count(int set) { return set == 0 ? 1 : (set == 1 ? 7 : 5); }
__device__ __forceinline__ int sym_to_txtype(int set, int s) {
  if (set == 0) return DCT_DCT;
  // (tables as packed 4-bit constants: a const array indexed at run time becomes a global-memory lookup, ~1 us on the block search's critical path)
  if (set == 1) return lut4(0x213ba09ULL, s);            // IDTX, DCT_DCT, V_DCT, H_DCT, ADST_ADST, ADST_DCT, DCT_ADST
  return lut4(0x21309ULL, s);                             // IDTX, DCT_DCT, ADST_ADST, ADST_DCT, DCT_ADST
}
__device__ __forceinline__ int txtype_to_sym(int set, int t) {
  const int n = tx_set_count(set);
  for (int i = 0; i < n; i++) if (sym_to_txtype(set, i) == t) return i;
  return -1;
}
__device__ __forceinline__ int mode_to_txtype(int m) {
  // DCT_DCT, ADST_DCT, DCT_ADST, DCT_DCT, ADST_ADST, ADST_DCT, DCT_ADST, DCT_ADST, ADST_DCT, ADST_ADST, ADST_DCT, DCT_ADST, ADST_ADST, DCT_DCT
  return lut4(0x3213122130210ULL, m);
}
__device__ __forceinline__ int tx_class_of(int t) {
  if (t == V_DCT || t == V_ADST || t == V_FLIPADST) return TXC_VERT;
  if (t == H_DCT || t == H_ADST || t == H_FLIPADST) return TXC_HORIZ;
  return TXC_2D;
}
// LDS copy of the default scans: [4x4 | 8x8 | 16x16 | 32x32] = 16 + 64 + 256 + 1024 entries
#define SCAN_LDS_ENTRIES(maxn) ((maxn) >= 32 ? 1360 : 336)
__device__ inline void load_scans_to_lds(LDS uint16_t *ls, int maxn) {
  for (int i = LANE; i < 16; i += 64) ls[i] = av1_default_scan_4x4[i];
  for (int i = LANE; i < 64; i += 64) ls[16 + i] = av1_default_scan_8x8[i];
  for (int i = LANE; i < 256; i += 64) ls[80 + i] = av1_default_scan_16x16[i];
  if (maxn >= 32) for (int i = LANE; i < 1024; i += 64) ls[336 + i] = av1_default_scan_32x32[i];
}
// scan position i -> raster position within the n x n coded area (n = min(32, tx size))
__device__ __forceinline__ int scan_pos(const LDS uint16_t *ls, int n, int cls, int i) {
  if (cls == TXC_2D) return ls[(n == 4 ? 0 : n == 8 ? 16 : n == 16 ? 80 : 336) + i];
  if (cls == TXC_VERT) return i;                       // mrow scan
  const int c = i / n, r = i - c * n; return r * n + c; // mcol scan
}
// intra tx-type CDF row for luma; returns -1 when the type is not signalled.  `reduced`: the frame's reduced_tx_set switch (a constant in the tile-search kernels
// instantiated for a fixed tool set, tile_search.h Tools)
template <typename FP> __device__ __forceinline__ int intra_tx_cdf_r(FP f, bool reduced, int txs, int ymode, int *nsyms, int *set_out) {
  const int set = tx_set_of(txs, reduced);
  *set_out = set;
  if (set == 0 || f->base_q_idx == 0) { *nsyms = 0; return -1; }
  if (set == 1) { *nsyms = 7; return CDF_INTRA_TX1 + (txs * 13 + ymode) * CDF_INTRA_TX1_STRIDE; }
  *nsyms = 5; return CDF_INTRA_TX2 + (txs * 13 + ymode) * CDF_INTRA_TX2_STRIDE;
}
template <typename FP> __device__ __forceinline__ int intra_tx_cdf(FP f, int txs, int ymode, int *nsyms, int *set_out) { return intra_tx_cdf_r(f, f->reduced_tx_set != 0, txs, ymode, nsyms, set_out); }

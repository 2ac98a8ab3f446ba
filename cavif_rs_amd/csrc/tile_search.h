// tile_search.h -- kernel K1: the AV1 intra encode loop for one tile.
//   superblocks in raster order -> top-down partition RDO (rav1e encode_partition_topdown /
//   rdo_partition_decision) -> per block: 13-mode SATD pre-filter, angle-delta refinement, full RD over
//   (mode x tx type) with forward transform, quantise, static-table rate, dequantise, inverse transform,
//   SSE (rdo_mode_decision / rdo_tx_type_decision), chroma DC / same-as-luma / CfL with alpha search
//   (rdo_cfl_alpha).
// Mapping: one WORKGROUP of NW wavefronts per tile.  The serial spine (block after block) is walked by all
// waves in lock step; inside a block the RDO candidates are dealt round-robin to the waves -- one wavefront
// per candidate -- and the winner is picked through LDS with the oracle's tie-break (lowest candidate
// index among equal costs), so the decisions are bit-identical to oracle/av1o_search.c.
// Source block, edges, prediction, residual and candidate reconstructions live in LDS; the frame-sized
// recon / coefficient / mode-info maps live in HBM and are touched only by the owning tile.
#pragma once
#include "dev_common.h"
#include "dev_predict.h"
#include "dev_txfm.h"
#include "dev_rate.h"
#include "dev_group.h"

// The shape of the search's code, fixed explicitly (left to the inliner's cost model it flips with unrelated edits: with the block searches inlined into
// separate partition functions K1 <2,4> went from 116 to 142 ms): the block searches are functions of their own -- entered with every argument in registers
// and, not being tail-called, given LLVM's no-callee-saved-registers treatment --, the partition walkers are inlined into the kernel.
#ifndef MI_BALLOT64                                      /* (the CPU test harness tests/emu/ predefines it) */
#define MI_BALLOT64(p) __builtin_amdgcn_ballot_w64(p)
#endif
#define MI_K1_TRY_ATTR __attribute__((noinline, not_tail_called))
#define MI_K1_WALK_INLINE __forceinline__
// The loop over the four children of a split node: unrolled, the recursion's inlining multiplies the node's code by four per level (64 copies of the 8x8 node's body in the
// 16x16 class: ~600 KB of code, 1127 SGPR spills in the kernel body); rolled (-DMI_K1_WALK_ROLLED=1), one copy per level.
#ifndef MI_K1_PAIRED_CHAINS                             /* 0: one row per chain (16 / 4 serial steps per depth), tools/build_variant.sh A/B */
#define MI_K1_PAIRED_CHAINS 1
#endif
#ifndef MI_K1_PAIRED_MIN_N                               /* the smallest block whose trial runs paired chains (8: every trial of the 16x16 class) */
#define MI_K1_PAIRED_MIN_N 8
#endif
#ifndef MI_K1_PAIRED_MAX_HN                              /* the largest sub-block whose chains run paired: 8 = every grouped chain of the 16x16 class.  (With the build flags of the
                                                           round's first half pairing the 8x8 sub-block chains bought nothing and spilled more -- profiles/r06v_ab_pair_only_4x4.txt --;
                                                           with -sink-insts-to-avoid-spills / -disable-machine-licm it is worth 0.6 ms: profiles/r06_ab_switches_under_final_flags.txt) */
#define MI_K1_PAIRED_MAX_HN 8
#endif
#ifndef MI_K1_CHAIN_PRIO_WAVES
#define MI_K1_CHAIN_PRIO_WAVES 0
#endif
#ifndef MI_K1_WALK_ROLLED
#define MI_K1_WALK_ROLLED 1
#endif
#if MI_K1_WALK_ROLLED
#define MI_K1_WALK_SPLIT_LOOP _Pragma("unroll 1")
#else
#define MI_K1_WALK_SPLIT_LOOP _Pragma("unroll")
#endif
#define MI_K1_WG_PER_CU 4                            /* the 16x16 class: 40.9 KB of LDS and 128 VGPRs per workgroup -> exactly four per CU */
#define WAVE_ID ((int)(threadIdx.x >> 6))
#ifndef MI_PROFILE
#define MI_PROFILE 0
#endif
#ifndef MI_PROF_MAXN
#define MI_PROF_MAXN 16                             /* profiling builds time the 16x16-class launch (-DMI_PROF_MAXN=32: the 64x64 class, tools/k1_phases.py 5) */
#endif
// bisect hooks (MI_DEBUG_LEVEL): probe builds only (-DMI_DEBUG_HOOKS=1); release builds carry no debug branches in K1
#ifndef MI_DEBUG_HOOKS
#define MI_DEBUG_HOOKS 0
#endif
#if MI_DEBUG_HOOKS
#define DBG_IS(f, v) ((f)->dbg == (v))
#else
#define DBG_IS(f, v) false
#endif
// phase timers (profiling builds only): per wave, accumulated in LDS, flushed to the tile's clock record
#if MI_PROFILE
#define PH_BEGIN() unsigned long long ph_t_ = clock64()
#define PH(i) do { const unsigned long long n_ = clock64(); if (LANE == 0 && !(MI_PROFILE == 3 && (i) >= 16 && (i) <= 21)) SH->prof[W][i] += n_ - ph_t_; ph_t_ = n_; } while (0)
// -DMI_PROFILE=3: slots 16..21 hold the time per block size (4x4, 8x8, 16x16, 32x32 / 64x64, 8x4, 4x8) instead of the evaluation's sub-phases
#define PH_SIZE(slot, call) [&]() { const unsigned long long t0_ = clock64(); const long long r_ = (call); if (MI_PROFILE == 3 && LANE == 0) k.sh()->prof[WAVE_ID][16 + (slot)] += clock64() - t0_; return r_; }()
#else
#define PH_SIZE(slot, call) (call)
#define PH_BEGIN() do {} while (0)
#define PH(i) do {} while (0)
#endif

// area snapshots of the partition search, one area per resident workgroup in HBM
#define MI_SNAP_BYTES(n) (3 * (n) * (n) * 6 + 18 * ((n) / 4) * ((n) / 4) + 3 * ((n) / 4) * ((n) / 4) * 2 + 64)
#define MI_SNAP_BYTES_SQ(n) (2 * MI_SNAP_BYTES(n))            /* sum over the levels < 4/3 of the largest */
#define MI_SNAP_BYTES_ALL(n) (MI_SNAP_BYTES_SQ(n) + 2 * MI_SNAP_BYTES(8))   /* + the 8x8 node's best rectangular / split candidates */

// The 32x32 class also evaluates 64x64 blocks (dev_blk64.h): their candidates run one per wavefront too, on a leaner working set laid over the
// wave's scratch -- the 64x64 prediction (reconstructed in place), the 32 rows of the column pass that the row pass keeps, the 32x32 coded area.
struct Blk64Wave { uint16_t pred[64 * 64]; int32_t tbuf[32 * 65]; int32_t cbuf[32 * 32]; int32_t qc[32 * 32]; };
struct Blk64None { uint8_t none_; };
struct Blk64Shared { uint16_t src64[64 * 64], bnd[2048];     /* bnd: four waves' chain boundaries (dev_blk64.h) */ uint8_t cnb_top[2][16][2], cnb_left[2][16][2]; };
template <int N> struct WaveScratch {            // private to one wavefront
  static constexpr int CS = N < 32 ? N : 32, NBUF = 2, DCP_LEN = N * N, E = N == 32 ? 64 : N;   // E: the largest block whose edges the wave prepares
  uint16_t wa[EDGE_LEN(E)], wl[EDGE_LEN(E)], etmp[2 * E + 16];
  union {
    struct {
      uint16_t pred[N * N], dcp[N * N];
      union {                                          // one candidate at a time (all sizes) or four at a time (4x4 / 8x8, dev_group.h)
        struct {
          uint16_t rec[2][N * N];
          int32_t tbuf[N * (N + 1)], cbuf[CS * CS], qc[2][CS * CS];  // cbuf doubles as the dequantised block
        };
        GroupBuf8 grp[4];
        GroupPredBuf gpred[4];                         // four directional predictions at a time (SATD stages of 4x4 / 8x8 blocks)
      };
    };
    typename std::conditional<N == 32, Blk64Wave, Blk64None>::type b64;
  };
  uint8_t lev[LEV_BYTES(CS)];                     // one padded level map per coded size (dev_rate.h LEV_OFF)
};
template <int N> struct SharedScratch {          // shared by the waves of the tile
  static constexpr int E = N == 32 ? 64 : N, NC = E >= 16 ? (E / 8) * (E / 8) : 1, SNC = E >= 32 ? (E / 16) * (E / 16) : 1;   // E: the class's largest block; its 8x8 cells; a sub-block's
  uint16_t ra[3][EDGE_LEN(E)], rl[3][EDGE_LEN(E)];
  uint16_t srcb[3][N * N];
  uint16_t luma_rec[N * N];
  long long satd[13], dsd[7][6];
  long long wbest_j[4], cj[16][2], pbest_j[2];
  int order[13], wbest_e[4], pbest_c[2], calpha[2][2], ldelta[7];
  uint16_t part_cost[40];                          // the partition symbols the walker prices (part_cost_idx): the slice of the rate table it needs, per frame / tile like the coefficient slices
  uint16_t lpred[N <= 32 ? 768 : 4];               // final luma predictions of the surviving modes, n*n samples each (three at 16x16, up to seven at 8x8 / 4x4)
  long long ca_sse[2][2]; int ca_idx[2][2];                  // CfL alpha search, [plane][half of the alpha range]
  int lm_mode, lm_delta, lm_tx, lm_eob, ceob[2], sctx[3], dctx[3];
  // luma transform-size trial, LDS-resident: the undivided winner is already committed to the frame; the four sub-blocks are
  // reconstructed into split_rec / split_qc (for MAXN <= 16 these alias lpred, which is dead after the mode decision) and only
  // reach the frame if the split wins.  ssrc = the four sub-sources, spred = the current sub-block's prediction, nb_* = the
  // (level, dc) contexts the block's outer neighbours left behind.
  long long lm_mode_j;   // (the trial's per-sub-block results alias dsd / satd, dead by then)
  // Tune::Psychovisual references of the block being evaluated: source variance + activity scale per 8x8 cell (a 4x4 block:
  // its own variance), the four 4x4 variances of an 8x8 block, and the block's mean activity for chroma
  int psv[NC], pact[NC], psv4[4], spsv[SNC], spact[SNC], cact, seg, seg_nb;
  uint16_t ssrc[N * N], spred[(E / 2) * (E / 2)];
  uint8_t nb_top[16][2], nb_left[16][2];
  int32_t split_qc[N <= 16 ? 1 : (N >= 64 ? 4096 : N * N)];
  uint16_t split_rec[N <= 16 ? 1 : N * N];
  // the 64x64 level of the 32x32 class (dev_blk64.h): the luma source, the chroma planes' outer neighbour contexts, and the bottom rows / right columns
  // of the sub-blocks done so far (the transform blocks of a 64x64 block are predicted one from the other; their reconstructions wait in HBM)
  typename std::conditional<N == 32, Blk64Shared, Blk64None>::type x64;
  TileB tile; uint8_t *snap;                       // the tile's bounds (mi units) and its snapshot area (per-tile constants of Ctx)
  int q_item;                                      // the work item the workgroup has just claimed
  int fine;                                        // the launch synchronises per root block instead of per superblock (root_wait / root_publish)
#if MI_PROFILE
  unsigned long long prof[4][32];
#endif
};

struct TxRes { int eob, cul, dcc; long long sse; uint32_t rate; };
#if MI_PROFILE
extern __shared__ __align__(16) uint8_t mi_prof_smem_[];             // the workgroup's LDS block starts with its SharedScratch (any MAXN: prof is the last member)
template <int N> __device__ __forceinline__ LDS unsigned long long &mi_prof_slot_n(int w, int i) { return ((LDS SharedScratch<N> *)mi_prof_smem_)->prof[w][i]; }
#endif

// What the block search needs to find its working set.  Everything lives in the workgroup's LDS block at offsets fixed by
// (MAXN, NW) -- shared scratch, per-wave scratch, scan tables, coefficient cost slices, the frame descriptor head -- so the
// context is two 32-bit LDS pointers passed BY VALUE (registers): the accessors fold into ds_read offsets.  (It used to be a
// struct on the kernel's stack passed by reference: every use inside the non-inlined block search was a flat load from scratch,
// ~40 per call, each holding both wait counters.)
// TS, the tool set a kernel is instantiated for (the host launches the matching instantiation, mi_avif.hip search_mode):
//   bit 0 (FULL)  the full candidate set of speed <= 1 (complex_pred_modes): FULL == f->complex_modes, the kernels read the template parameter
//   bit 1 (FIXED) the switches of ravif's speed 4 below the high-quality threshold (av1encoder.rs:576-586: rdo_tx_decision, reduced_tx_set, fine_directional_intra,
//                 hence tx_mode_select; Tune::Psychovisual) are constants of the kernel instead of loads from the frame descriptor: the headline configuration's
//                 block searches lose the code of the other settings (-3 % K1).  Every other combination of switches runs the instantiations without the bit.
template <int TS> struct Tools {
  static constexpr bool FULL = (TS & 1) != 0, FIXED = (TS & 2) != 0;
  template <typename F> static __device__ __forceinline__ bool rdo_tx(F f) { return FIXED ? true : f->rdo_tx != 0; }
  template <typename F> static __device__ __forceinline__ bool reduced_tx_set(F f) { return FIXED ? true : f->reduced_tx_set != 0; }
  template <typename F> static __device__ __forceinline__ bool fine_directional(F f) { return FIXED ? true : f->fine_directional != 0; }
  template <typename F> static __device__ __forceinline__ bool tx_mode_select(F f) { return FIXED ? true : f->tx_mode_select != 0; }
  template <typename F> static __device__ __forceinline__ bool tune_psnr(F f) { return FIXED ? false : f->tune_psnr != 0; }
};
template <int MAXN, int NW, int TS_ = 0> struct Ctx {
  static constexpr int TS = TS_;
  static constexpr int MAXBS = MAXN == 16 ? 2 : 4;     // the largest transform whose rate slices the class needs: the 32x32 class evaluates 64x64 blocks too (dev_blk64.h)
  static constexpr size_t SH_BYTES = (sizeof(SharedScratch<MAXN>) + 15) & ~(size_t)15, WS_BYTES = (sizeof(WaveScratch<MAXN>) + 15) & ~(size_t)15;
  static constexpr size_t SC_BYTES = SCAN_LDS_ENTRIES(MAXN) * 2, CC_BYTES = (COEF_COST_MAX_ENTRIES(MAXBS) * 2 + 15) & ~(size_t)15;
  static constexpr size_t LS_OFF = SH_BYTES + NW * WS_BYTES, CC_OFF = LS_OFF + SC_BYTES, F_OFF = CC_OFF + CC_BYTES;
  LDS uint8_t *base;                  // the workgroup's LDS block (wave-uniform)
  LDS WaveScratch<MAXN> *ws;          // this wave's scratch inside it
  __device__ __forceinline__ LDS SharedScratch<MAXN> *sh() const { return (LDS SharedScratch<MAXN> *)base; }
  __device__ __forceinline__ LDS WaveScratch<MAXN> *s() const { return ws; }
  __device__ __forceinline__ LDS WaveScratch<MAXN> *wave(int w) const { return (LDS WaveScratch<MAXN> *)(base + SH_BYTES + (size_t)w * WS_BYTES); }
  __device__ __forceinline__ const LDS uint16_t *ls() const { return (const LDS uint16_t *)(base + LS_OFF); }
  __device__ __forceinline__ LDS uint16_t *cc_base() const { return (LDS uint16_t *)(base + CC_OFF); }
  __device__ __forceinline__ CoefCost cc() const { return coef_cost_layout(cc_base(), MAXBS); }
  __device__ __forceinline__ const LDS FrameDev *f() const { return (const LDS FrameDev *)(base + F_OFF); }
  __device__ __forceinline__ const LDS TileB *t() const { return &sh()->tile; }
  __device__ __forceinline__ uint8_t *snap() const { return sh()->snap; }
  __device__ __forceinline__ const uint16_t *cost() const { return f()->cost; }
};

// The block's segment (oracle/av1o_segment.c av1o_block_segment: looked up from its mean activity scale, no RD search over segments).  Called by
// every lane of one wave with a wave-uniform scale while the block is being staged: the segment's quantiser steps replace the frame's in the
// workgroup's LDS copy of the descriptor, which is where every evaluation of the block reads them.
static_assert(offsetof(FrameDev, ac_recip) - offsetof(FrameDev, dc_q) == 36 && sizeof(SegTab::Q) == 48, "a SegTab entry is copied over FrameDev::dc_q .. ac_recip");
template <typename SHT> __device__ __forceinline__ void seg_select(const LDS FrameDev *f, LDS SHT *SH, int scale) {
  if (!f->seg_n) return;
  const SegTab *st = f->seg;
  const int b = seg_bucket((uint32_t)uni32(scale));
  int a = 0;
  for (int j = 0; j + 1 < f->seg_n; j++) a += b >= st->thr[j];
  const int seg = f->seg_n - 1 - a;
  if (LANE < 12) ((LDS uint32_t *)((LDS FrameDev *)f)->dc_q)[LANE] = ((const uint32_t *)&st->q[seg])[LANE];
  if (LANE == 0) SH->seg = seg;
}

#define IS_SMOOTH_(m) ((m) == SMOOTH_PRED || (m) == SMOOTH_V_PRED || (m) == SMOOTH_H_PRED)
#define J_INF 0x7fffffffffffffffLL

// one wave fills an n4 x n4 area of a byte map
__device__ inline void fill_map_dev(uint8_t *m, int ms, int r, int c, int n4, int v) {
  for (int i = LANE; i < n4 * n4; i += 64) m[(r + i / n4) * ms + c + (i % n4)] = (uint8_t)v;
}
// "Was the 4x4 cell (pr, pc) decoded before the block at (cur_r, cur_c)?" -- the spec's BlockDecoded flag (7.11.2: above-right / below-left availability), which the
// oracle keeps as a map (m_decoded).  In this search it is a function of the two positions: tiles are walked superblock by superblock, a superblock in quadtree order,
// every trial of a node (NONE, SPLIT, HORZ, VERT) evaluates its blocks in coding order with the node's area considered undecoded, and everything a work item reads of
// other superblocks is final before it starts (the queue's dependencies).  So a cell of an earlier superblock row, or of a superblock further left in the same row, is
// decoded; a later one is not; inside the superblock the quadtree (Morton) order of the 4x4 cells decides -- except inside the current block's own 8x8 node when that
// node is being tried as two 8x4 (rows in order) or two 4x8 blocks (columns in order: the left half's lower cell precedes the right half although its Morton index
// is larger).  Callers check the tile bounds (cells outside the tile are unavailable whatever their order).  shape: 0 = square block, 1 = 8x4, 2 = 4x8.
__device__ __forceinline__ int morton4_(int v) { v = (v | (v << 2)) & 0x33; return (v | (v << 1)) & 0x55; }
__device__ __forceinline__ int decoded_before(int cur_r, int cur_c, int shape, int pr, int pc) {
  const int dsr = (pr >> 4) - (cur_r >> 4), dsc = (pc >> 4) - (cur_c >> 4);
  if (dsr != 0) return dsr < 0;
  if (dsc != 0) return dsc < 0;
  if (shape != 0 && (pr >> 1) == (cur_r >> 1) && (pc >> 1) == (cur_c >> 1)) return shape == 1 ? pr < cur_r : pc < cur_c;
  return ((morton4_(pr & 15) << 1) | morton4_(pc & 15)) < ((morton4_(cur_r & 15) << 1) | morton4_(cur_c & 15));
}

// SATD of (src - pred) over an n x n block; both in LDS with pitch n (oracle satd_block_wh): 4x4 Hadamards for a 4x4 block, one 8x8 Hadamard per 8x8 cell for everything
// larger (rav1e get_satd), each cell's sum brought to the scale of four 4x4 ones ((s + 2) >> 2).
// 4x4: one lane per (column, group of 4 rows): the vertical butterflies run in registers, the horizontal ones across the four lanes of a quad with DPP quad_perm (the
// sum of |H D H^T| does not depend on the order of the passes, so the value equals the oracle's row-then-column form exactly).
template <int CTRL> __device__ __forceinline__ int satd_quad_step(int v, int odd_mask) {
  const int p = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
  return (LANE & odd_mask) ? p - v : v + p;
}
// 8x8: an 8x8 cell occupies one 16-lane DPP row, four samples of one column per lane -- lane bits 0, 1 = column & 3, bit 2 = which four rows, bit 3 = column >> 2.  Two of
// the three vertical stages run in registers; the third (rows r and r + 4: lane ^ 4) and the three horizontal ones (lane ^ 1, ^ 2, ^ 8) run across lanes.  DPP has no
// "lane ^ 4", so that stage pairs lane i with lane 7 - i (row_half_mirror) and runs FIRST: the upper half then holds its differences in reverse order, the two quad
// stages transform a reversed vector -- a Walsh-Hadamard transform of an index-complemented input is the same transform up to signs --, and which of (A + B, A - B) lands
// where does not matter to a sum of magnitudes.  Returns the lane's share of the cell's sum of |H8 D H8^T|.
__device__ __forceinline__ int satd8_lane(int d0, int d1, int d2, int d3) {
  const int a = d0 + d1, b = d0 - d1, c = d2 + d3, e = d2 - d3;
  const int t[4] = { a + c, b + e, a - c, b - e };
  const int m4 = (LANE & 4) ? -1 : 0, m1 = (LANE & 1) ? -1 : 0, m2 = (LANE & 2) ? -1 : 0, m8 = (LANE & 8) ? -1 : 0;     // (v ^ m) - m = -v where the lane's bit is set
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int v = t[i];
    v = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false) + ((v ^ m4) - m4);     // row_half_mirror
    v = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false) + ((v ^ m1) - m1);      // quad_perm [1,0,3,2]
    v = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false) + ((v ^ m2) - m2);      // quad_perm [2,3,0,1]
    v = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false) + ((v ^ m8) - m8);     // row_ror 8
    s += iabs_(v);
  }
  return s;                                       // <= 4 * 64 * 1023 per lane; a cell's sum <= 64 * 64 * 1023 < 2^23
}
__device__ __forceinline__ int satd_row_sum_(int v) {    // all-reduce over the 16-lane row (row_ror 8, 4, 2, 1)
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false); v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false); v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);
  return v;
}
__device__ inline long long satd_dev(const LDS uint16_t *src, const LDS uint16_t *pred, int n) {
  int total = 0;
  if (n >= 8) {
    const int cells = n >> 3, units = n * (n >> 2);         // sixteen lanes per 8x8 cell: whole DPP rows are active or idle
    for (int u = LANE; u < units; u += 64) {
      const int cell = u >> 4, x = (cell % cells) * 8 + (u & 3) + ((u >> 1) & 4), y = (cell / cells) * 8 + (u & 4), o = y * n + x;
      const int d0 = (int)src[o] - (int)pred[o], d1 = (int)src[o + n] - (int)pred[o + n];
      const int d2 = (int)src[o + 2 * n] - (int)pred[o + 2 * n], d3 = (int)src[o + 3 * n] - (int)pred[o + 3 * n];
      const int cs = satd_row_sum_(satd8_lane(d0, d1, d2, d3));
      if ((u & 15) == 0) total += (cs + 2) >> 2;            // <= 64 cells (64x64) * 2^21 fits int
    }
    return (long long)wave_sum_i32(total);
  }
  const int units = n * (n >> 2);                 // multiple of 4: quads are either fully active or fully idle
  for (int u = LANE; u < units; u += 64) {
    const int x = u % n, o = (u / n) * 4 * n + x;
    const int d0 = (int)src[o] - (int)pred[o], d1 = (int)src[o + n] - (int)pred[o + n];
    const int d2 = (int)src[o + 2 * n] - (int)pred[o + 2 * n], d3 = (int)src[o + 3 * n] - (int)pred[o + 3 * n];
    const int a = d0 + d1, b = d0 - d1, c = d2 + d3, e = d2 - d3;
    int t[4] = { a + c, b + e, a - c, b - e };
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int v = satd_quad_step<0xB1>(t[i], 1);      // quad_perm [1,0,3,2]
      v = satd_quad_step<0x4E>(v, 2);             // quad_perm [2,3,0,1]
      s += iabs_(v);
    }
    total += s;
  }
  return (long long)wave_sum_i32(total);
}
__device__ inline long long sse_dev(const LDS uint16_t *a, const LDS uint16_t *b, int nn) {
  int s = 0;                                     // per lane <= 64 samples * 1023^2 < 2^27
  for (int i = LANE; i < nn; i += 64) { const int d = (int)a[i] - (int)b[i]; s += __mul24(d, d); }
  if (nn <= 1024) return (long long)wave_sum_i32(s);   // <= 1024 * 1023^2 < 2^31: one 32-bit reduction
  return wave_sum_i64((long long)s);
}

// Psychovisual luma distortion of an n x n block by one wave (oracle av1o_psy_dist_luma): the 64 lanes are dealt to the 8x8
// cells (64 / cells lanes per cell; a 4x4 block is one cell of 16 samples), sums reduce inside a cell's lane group, every
// group prices its own cell (boost + activity) in parallel, the cells add up.
template <int n> __device__ inline long long psy_dist_wave(const LDS uint16_t *src, const LDS uint16_t *rec, const LDS int *sv, const LDS int *act, int bd) {
  constexpr int w = n == 4 ? 4 : 8, cp = n / w, ncell = cp * cp;
  constexpr int LPC = n == 4 ? 16 : 64 / ncell, PPL = n == 4 ? 1 : 64 / LPC;
  const int cell = LANE / LPC, li = LANE % LPC;
  uint32_t sd = 0, qd = 0, se = 0;
  if (cell < ncell) {
    const int cy = cell / cp, cx = cell % cp;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
      const int q = li + LPC * k, o = (cy * w + q / w) * n + cx * w + q % w;
      const int d = rec[o], e = (int)src[o] - d;
      sd += (uint32_t)d; qd += (uint32_t)__mul24(d, d); se += (uint32_t)__mul24(e, e);
    }
  }
  if constexpr (LPC == 64) { sd = (uint32_t)wave_sum_i32((int)sd); qd = (uint32_t)wave_sum_i32((int)qd); se = (uint32_t)wave_sum_i32((int)se); }
  else if constexpr (LPC == 16) { sd = (uint32_t)row_sum_i32((int)sd); qd = (uint32_t)row_sum_i32((int)qd); se = (uint32_t)row_sum_i32((int)se); }
  else if constexpr (LPC == 4) {
    sd += (uint32_t)DPP_(0, sd, 0xB1, 0xF); sd += (uint32_t)DPP_(0, sd, 0x4E, 0xF);
    qd += (uint32_t)DPP_(0, qd, 0xB1, 0xF); qd += (uint32_t)DPP_(0, qd, 0x4E, 0xF);
    se += (uint32_t)DPP_(0, se, 0xB1, 0xF); se += (uint32_t)DPP_(0, se, 0x4E, 0xF);
  }
  int dist = 0;
  if (cell < ncell && li == 0) dist = psy_cell_dist(se, sd, qd, (uint32_t)sv[cell], (uint32_t)act[cell], w, bd);
  if constexpr (n >= 32) return wave_sum_i64((long long)dist);
  return (long long)wave_sum_i32(dist);
}

// Rate of a candidate's mode symbols from the static table.  Every entry is fetched with an UNCONDITIONAL load (entry 0 of its table
// where the symbol is not coded), so the loads leave as one batch; written as `if (coded) rate += cost[...]` each of them was a
// separate round trip to L2 inside its own branch.
__device__ __forceinline__ uint32_t y_mode_rate(const uint16_t *cost, const uint16_t *ycost, int m, bool angle_coded, int delta) {
  const uint32_t r_mode = ycost[m], r_ang = cost[CDF_ANGLE + (angle_coded ? (m - V_PRED) * CDF_ANGLE_STRIDE + delta + 3 : 0)];
  return r_mode + (angle_coded ? r_ang : 0u);
}
__device__ __forceinline__ uint32_t uv_mode_rate(const uint16_t *cost, const uint16_t *uvcost, int um, bool angle_coded, int delta, bool cfl, int alpha_u, int alpha_v, int *jsign) {
  const int su = alpha_u == 0 ? 0 : (alpha_u < 0 ? 1 : 2), sv = alpha_v == 0 ? 0 : (alpha_v < 0 ? 1 : 2);
  const int js = cfl ? su * 3 + sv - 1 : 0;
  const uint32_t r_mode = uvcost[um];
  const uint32_t r_ang = cost[CDF_ANGLE + (angle_coded ? (um - V_PRED) * CDF_ANGLE_STRIDE + delta + 3 : 0)];
  const uint32_t r_sign = cost[CDF_CFL_SIGN + js];
  const uint32_t r_au = cost[CDF_CFL_ALPHA + ((cfl && su) ? ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE + iabs_(alpha_u) - 1 : 0)];
  const uint32_t r_av = cost[CDF_CFL_ALPHA + ((cfl && sv) ? ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE + iabs_(alpha_v) - 1 : 0)];
  *jsign = js;
  return r_mode + (angle_coded ? r_ang : 0u) + (cfl ? r_sign + (su ? r_au : 0u) + (sv ? r_av : 0u) : 0u);
}

// One transform block by one wave: residual -> fwd -> quant -> rate, dequant -> inverse -> recon; returns weighted J.
template <int MAXN, int BS, int NW, int TS>
__device__ inline long long eval_tx(const Ctx<MAXN, NW, TS> k, int plane, int sctx, int dctx, const LDS uint16_t *pred, int txtype, int tx_off, int tx_sym,
                                    LDS uint16_t *rec_out, LDS int32_t *qc_out, TxRes *tr, const LDS uint16_t *src_override = nullptr,
                                    const LDS int *psv = nullptr, const LDS int *pact = nullptr) {
  constexpr int n = 4 << BS, P = n + 1, CS = n < 32 ? n : 32;
  const LDS FrameDev *f = k.f(); LDS WaveScratch<MAXN> *S = k.s();
  const LDS uint16_t *src = src_override ? src_override : k.sh()->srcb[plane];
#if MI_PROFILE
  LDS SharedScratch<MAXN> *SH = k.sh(); const int W = WAVE_ID;
#endif
  PH_BEGIN();
  for (int idx = LANE; idx < n * n; idx += 64) {
    const int i = idx / n, j = idx % n;
    S->tbuf[i * P + j] = (int)src[idx] - (int)pred[idx];
    rec_out[idx] = pred[idx];
  }
  WAVE_SYNC();
  PH(16);
  fwd_txfm2d_dev<n>(S->tbuf, S->cbuf, txtype);
  PH(17);
  const int eob = quant_rate_dev<CS>(k.cc(), k.cost(), k.ls(), S->cbuf, qc_out, S->lev, plane, BS, txtype, f->dc_q[plane], f->ac_q[plane], f->dc_recip[plane], f->ac_recip[plane],
                                     f->bd, sctx, dctx, tx_off, tx_sym, &tr->rate, &tr->cul, &tr->dcc);
  PH(19);
  if (eob > 0) inv_txfm2d_add_dev<n>(S->cbuf, S->tbuf, rec_out, txtype, f->bd);
  tr->eob = eob;
  PH(20);
  // distortion: luma = psychovisual cdef-dist per 8x8 cell x activity; chroma = SSE x the block's mean activity
  if (plane == 0 && !Tools<TS>::tune_psnr(f)) tr->sse = psy_dist_wave<n>(src, rec_out, psv ? psv : (const LDS int *)k.sh()->psv, pact ? pact : (const LDS int *)k.sh()->pact, f->bd);
  else { const long long e = sse_dev(src, rec_out, n * n); tr->sse = plane == 0 ? e : (e * k.sh()->cact + 8192) >> 14; }
  PH(21);
  return ((tr->sse * f->wq[plane]) >> 5) + (((long long)tr->rate * f->rdmult + 256) >> 9);
}

// one wave writes a plane's result back to the frame buffers
template <int BS, typename RecPtr, typename QcPtr>
__device__ inline void commit_plane(const LDS FrameDev *f, int plane, int r, int c, RecPtr rec, QcPtr qc, int eob, int cul, int dcc) {
  constexpr int n = 4 << BS, CS = n < 32 ? n : 32, n4 = 1 << BS;
  uint16_t *gr = f->rec[plane] + (size_t)(r * 4) * f->stride + c * 4;
  int32_t *gc = f->coef[plane] + (size_t)(r * 4) * f->stride + c * 4;
  for (int idx = LANE; idx < n * n; idx += 64) gr[(idx / n) * f->stride + (idx % n)] = rec[idx];
  for (int idx = LANE; idx < CS * CS; idx += 64) gc[(idx / CS) * f->stride + (idx % CS)] = qc[idx];
  fill_map_dev(f->m_lvl[plane], f->mi_stride, r, c, n4, cul);
  fill_map_dev(f->m_dc[plane], f->mi_stride, r, c, n4, dcc);
  if (LANE == 0) f->m_eob[plane][r * f->mi_stride + c] = (uint16_t)eob;
}

#include "dev_rect.h"

// `budget`: the caller only needs to know whether the block's cost stays below it (split trials: cost of the
// undivided block minus what the earlier sub-blocks already cost).  Costs only grow, so once the luma part alone
// reaches the budget the rest of the evaluation cannot change the caller's decision and is skipped.
template <int MAXN, int BS, int NW, int TS>
// not_tail_called: with every argument in registers the calls would be marked `tail`, and LLVM's interprocedural register
// allocation then refuses its no-callee-saved-registers treatment for this function (TargetFrameLowering::isSafeForNoCSROpt):
// the prologue / epilogue would spill and reload 46 VGPRs + 34 SGPRs per call.
__device__ MI_K1_TRY_ATTR long long try_block(const Ctx<MAXN, NW, TS> k, int r, int c, long long budget = J_INF) {
  constexpr int n = 4 << BS, n4 = 1 << BS, log2w = 2 + BS, nn = n * n, CS = n < 32 ? n : 32, qn = CS * CS;
  const LDS FrameDev *f = k.f(); const LDS TileB *t = k.t(); LDS WaveScratch<MAXN> *S = k.s(); LDS SharedScratch<MAXN> *SH = k.sh();
  const int W = NW > 1 ? WAVE_ID : 0;
  const int ms = f->mi_stride, mi = r * ms + c, x = c * 4, y = r * 4;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  // Neighbour context of the block, fetched as ONE batch of unconditional loads: where a neighbour lies outside the tile the load
  // goes to the block's own cell (always addressable) and the value is replaced afterwards.  Written as `cond ? map[i] : dflt` each
  // load sat in its own branch with its own s_waitcnt -- a dozen serial round trips to L2 at the head of every block evaluation.
  const int can_ar = availU && (c + n4 < t->mi_col_end), can_bl = availL && (r + n4 < t->mi_row_end);
  const int iU = availU ? mi - ms : mi, iL = availL ? mi - 1 : mi;
  const int v_ymU = f->m_ymode[iU], v_ymL = f->m_ymode[iL];
  const int v_uvU = f->np > 1 ? f->m_uvmode[iU] : 0, v_uvL = f->np > 1 ? f->m_uvmode[iL] : 0;
  const int v_skU = f->m_skip[iU], v_skL = f->m_skip[iL], v_txU = f->m_txsize[iU], v_txL = f->m_txsize[iL];
  const int v_skUL = f->m_skip[availU && availL ? mi - ms - 1 : mi];                                  // (segment id of the above-left neighbour)
  const int have_ar = can_ar && decoded_before(r, c, 0, r - 1, c + n4), have_bl = can_bl && decoded_before(r, c, 0, r + n4, c - 1);
  const int amode = availU ? uni32(v_ymU) : DC_PRED, lmode = availL ? uni32(v_ymL) : DC_PRED;
  const int nb_skip = (availU ? uni32(v_skU) & 1 : 0) + (availL ? uni32(v_skL) & 1 : 0);              // skip context (bit 0 of the map)
  // neighbours' segment ids + 1 (0 = outside the tile), packed: needed again when the block's skip flag is known -- parked in LDS, not in registers
  const int seg_nb = (availU && availL ? (uni32(v_skUL) >> 1) + 1 : 0) | ((availU ? (uni32(v_skU) >> 1) + 1 : 0) << 4) | ((availL ? (uni32(v_skL) >> 1) + 1 : 0) << 8);
  const int nb_txU = availU ? uni32(v_txU) : -1, nb_txL = availL ? uni32(v_txL) : -1;                   // tx-size context (-1: no neighbour)
  const uint16_t *ycost = k.cost() + CDF_KF_Y + (intra_mode_ctx(amode) * 5 + intra_mode_ctx(lmode)) * CDF_KF_Y_STRIDE;
  const int ftype_y = IS_SMOOTH_(amode) || IS_SMOOTH_(lmode);                                           // DC_PRED (no neighbour) is not smooth
  const int ftype_uv = f->np > 1 && ((availU && IS_SMOOTH_(uni32(v_uvU))) || (availL && IS_SMOOTH_(uni32(v_uvL))));
  LDS uint16_t *wa = S->wa + EDGE_OFF, *wl = S->wl + EDGE_OFF;
  PH_BEGIN();

  // ---- stage the source block and the raw edges of every plane (plane p by wave p % NW) ----
  for (int p = 0; p < f->np; p++) if ((p + 1) % NW == W) {      // waves 1..3: wave 0 carries the longest candidate chains below
    int sc_, dc_;                                            // all-zero / dc-sign contexts depend on the neighbours only
    txb_ctx_dev(f, t, p, r, c, BS, BS, &sc_, &dc_);
    if (LANE == 0) { SH->sctx[p] = sc_; SH->dctx[p] = dc_; }
    const uint16_t *g = f->src[p] + (size_t)y * f->stride + x;
    for (int idx = LANE; idx < nn; idx += 64) SH->srcb[p][idx] = g[(idx / n) * f->stride + (idx % n)];
    load_edges(f, p, x, y, n, availL, availU, have_ar, have_bl, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF);
  }
  if (W == 0) {                                               // wave 0 stages no plane: the block's psychovisual references
    constexpr int cp = n >= 8 ? n / 8 : 1, ncell = cp * cp;
    const int cw = f->pw >> 3;
    int a = 0;
    if (LANE < ncell) {
      const int cell = ((y >> 3) + LANE / cp) * cw + (x >> 3) + LANE % cp;
      a = (int)f->act[cell]; SH->pact[LANE] = a; SH->psv[LANE] = (int)(n == 4 ? f->svar4[mi] : f->svar8[cell]);
    }
    if (BS == 1 && LANE < 4) SH->psv4[LANE] = (int)f->svar4[(r + (LANE >> 1)) * ms + c + (LANE & 1)];
    const int tot = wave_sum_i32(a);
    const int cact = (tot + ncell / 2) / ncell;
    if (LANE == 0) { SH->cact = cact; SH->seg_nb = seg_nb; }
    seg_select(f, SH, cact);
  }
  PH(1);
  WG_SYNC();
  PH(2);
  if (DBG_IS(f, 3)) return 0;
  const int sctx_y = SH->sctx[0], dctx_y = SH->dctx[0];        // (chroma reads SH->sctx[p] in place: a private array indexed by p lives in scratch)
  const LDS uint16_t *ra = SH->ra[0] + EDGE_OFF, *rl = SH->rl[0] + EDGE_OFF;

  // ---- luma: SATD pre-filter over the 13 modes (mode m by wave m % NW) ----
  constexpr bool SMALL_GROUPED = BS <= BS_8 && NW == 4 && MAXN <= 32;      // (the 64x64 class runs one wavefront per workgroup)
  if constexpr (SMALL_GROUPED) {
    // 4x4 / 8x8: the eight directional modes run four per wave (one prediction angle per 16-lane row, dev_group.h) on
    // waves 0 and 1 -- rows sorted so that a wave's rows mostly share the interpolation branch --, the five others
    // on rows of waves 2 and 3 (predict_nondir_group)
    if (W < 2) {
      const int g = GROUP_ID;
      // wave 0: the predictions that read ONE edge (V, D203: left only; D45, D67: above only); wave 1: the three between 90 and 180 degrees that interpolate on both, and H
      const int m = W == 0 ? (g == 0 ? V_PRED : g == 1 ? D203_PRED : g == 2 ? D45_PRED : D67_PRED) : (g < 3 ? D135_PRED + g : H_PRED);
      LDS GroupPredBuf *gp = &S->gpred[g];
      predict_dir_group<n>(f, x, y, availL, availU, mode_angle_of(m), ftype_y, ra, rl, gp);
      const int sd = satd_group<n>(SH->srcb[0], gp->pred);
      if (GROUP_LANE == 0) SH->satd[m] = (long long)sd;
    } else {
      // the five others the same way, one per row: DC, PAETH, SMOOTH on wave 2, SMOOTH_V, SMOOTH_H on wave 3
      const int g = GROUP_ID;
      const int m = W == 2 ? (g == 0 ? DC_PRED : g == 1 ? PAETH_PRED : g == 2 ? SMOOTH_PRED : -1) : (g == 0 ? SMOOTH_V_PRED : g == 1 ? SMOOTH_H_PRED : -1);
      LDS GroupPredBuf *gp = &S->gpred[g];
      predict_nondir_group<n>(m, availL, availU, f->bd, ra, rl, gp->pred);
      const int sd = satd_group<n>(SH->srcb[0], gp->pred);
      if (m >= 0 && GROUP_LANE == 0) SH->satd[m] = (long long)sd;
    }
  } else if constexpr (NW == 4) {
  // dealt by cost rather than round-robin: the six diagonal modes (edge filter + interpolation) weigh about three
  // cheap ones, so each wave gets 2 + 1, 2 + 1, 1 + 3 and 1 + 2 of them
    const int deal = W == 0 ? 0x0F043 : W == 1 ? 0x0F165 : W == 2 ? 0x2C97 : 0x0FBA8;     // four mode nibbles per wave, 0xF = none
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int m = (deal >> (4 * i)) & 15;
      if (m < 13) {
        predict_block(f, x, y, log2w, availL, availU, m, 0, ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
        const long long sd = satd_dev(SH->srcb[0], S->pred, n);
        if (LANE == 0) SH->satd[m] = sd;
      }
    }
  } else {
    for (int m = W; m < 13; m += NW) {
      predict_block(f, x, y, log2w, availL, availU, m, 0, ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
      const long long sd = satd_dev(SH->srcb[0], S->pred, n);
      if (LANE == 0) SH->satd[m] = sd;
    }
  }
  PH(3);
  WG_SYNC();
  PH(2);
  // stable sort by SATD as a rank computation (mode m = lane): every wave writes the same 13 values, so no
  // workgroup barrier is needed before they are read back
  if (LANE < 13) {
    const long long mine = SH->satd[LANE];
    int rank = 0;
    for (int j = 0; j < 13; j++) { const long long o = SH->satd[j]; rank += (o < mine) || (o == mine && j < LANE); }
    SH->order[rank] = LANE;
  }
  WAVE_SYNC();
  PH(4);
  const int ncand = Tools<TS>::FULL ? 7 : 3;
  // angle-delta refinement by SATD: unit (ci, q) by wave (ci*6+q) % NW
  auto dl_of = [](int q) { const int a = (q >> 1) + 1; return (q & 1) ? a : -a; };        // -1, 1, -2, 2, -3, 3
  const int refine = BS >= BS_8 && Tools<TS>::fine_directional(f);
  bool refine_grouped = false;
  if constexpr (SMALL_GROUPED) refine_grouped = refine && ncand == 3;
  if constexpr (SMALL_GROUPED) if (refine_grouped) {
    // Six probes per DIRECTIONAL candidate among the three, dealt densely to the 16 rows of the workgroup (probe p: pass p / 16, wave (p / 4) % 4, row p % 4): one
    // pass when at most two candidates are directional, and only wave 0 takes a second one (two rows) when all three are.  (Until round 5 wave 3 always owned
    // the +-3 probes in two passes of its own: the phase's longest wave at 5.9 % of the kernel against 3.5 % for the others.)
    const int g = GROUP_ID;
    const int o0 = SH->order[0], o1 = SH->order[1], o2 = SH->order[2];
    const bool d0 = o0 >= V_PRED && o0 <= D67_PRED, d1 = o1 >= V_PRED && o1 <= D67_PRED, d2 = o2 >= V_PRED && o2 <= D67_PRED;
    const int nprobe = 6 * ((int)d0 + (int)d1 + (int)d2);
    for (int pass = 0; pass * 16 + W * 4 < nprobe; pass++) {   // wave-uniform
      const int pr = pass * 16 + W * 4 + g;
      const bool live = pr < nprobe;
      const int kd = live ? pr / 6 : 0, q = live ? pr - kd * 6 : 0;
      const int ci = d0 ? (kd == 0 ? 0 : (d1 ? (kd == 1 ? 1 : 2) : 2)) : (d1 ? (kd == 0 ? 1 : 2) : 2);     // the kd-th directional candidate
      const int m = SH->order[ci];
      LDS GroupPredBuf *gp = &S->gpred[g];
      predict_dir_group<n>(f, x, y, availL, availU, live ? mode_angle_of(m) + 3 * dl_of(q) : 90, ftype_y, ra, rl, gp);
      const int sd = satd_group<n>(SH->srcb[0], gp->pred);
      if (live && GROUP_LANE == 0) SH->dsd[ci][q] = (long long)sd;
    }
    PH(5);
    WG_SYNC();
    PH(2);
  }
  if (refine && !refine_grouped) {
    for (int u = W; u < ncand * 6; u += NW) {
      const int ci = u / 6, q = u - ci * 6, m = SH->order[ci];
      if (m >= V_PRED && m <= D67_PRED) {
        predict_block(f, x, y, log2w, availL, availU, m, dl_of(q), ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
        const long long sd = satd_dev(SH->srcb[0], S->pred, n);
        if (LANE == 0) SH->dsd[ci][q] = sd;
      }
    }
    PH(5);
    WG_SYNC();
    PH(2);
  }
  // ---- full RD over surviving (mode, delta) x tx type: eval e = ci*ntx + ti by wave e % NW ----
  int tx_ns = 0, tx_set = 0;
  const int tx_off0 = intra_tx_cdf_r(f, Tools<TS>::reduced_tx_set(f), BS, 0, &tx_ns, &tx_set);
  const int ntx = (Tools<TS>::rdo_tx(f) && tx_off0 >= 0) ? tx_ns : 1;
  const bool tx_trial = BS > 0 && Tools<TS>::tx_mode_select(f) && Tools<TS>::rdo_tx(f);        // one-level-smaller luma transforms are tried after the mode decision
  LDS int32_t *split_qc = MAXN <= 16 ? (LDS int32_t *)SH->lpred : (LDS int32_t *)SH->split_qc;
  LDS uint16_t *split_rec = MAXN <= 16 ? SH->lpred + 512 : (LDS uint16_t *)SH->split_rec;
  LDS uint16_t *spred = SH->spred;
  // the surviving (mode, delta) predictions are built once (candidate ci by wave ci) and shared by its tx-type trials
  const bool pred_cached = MAXN <= 32 && NW > 1 && ncand * nn <= 768;
  if (pred_cached) {
    for (int ci = NW - 1 - W; ci < ncand; ci += NW) {          // waves 3, 2, 1
      const int m = SH->order[ci];
      int delta = 0;
      if (m >= V_PRED && m <= D67_PRED && refine) {
        long long bsd = SH->satd[m];
        for (int q = 0; q < 6; q++) { const long long sd = SH->dsd[ci][q]; if (sd < bsd) { bsd = sd; delta = dl_of(q); } }
      }
      predict_block(f, x, y, log2w, availL, availU, m, delta, ftype_y, ra, rl, wa, wl, S->etmp, SH->lpred + ci * nn);
      if (LANE == 0) SH->ldelta[ci] = delta;
    }
    PH(12);
    WG_SYNC();
    PH(2);
  }
  long long my_j = J_INF; int my_e = 1 << 30, my_mode = DC_PRED, my_delta = 0, my_tx = DCT_DCT, cur = 0; TxRes my_tr = { 0, 0, 0, 0, 0 };
  uint32_t my_mrate = 0;                                      // mode (+ angle) rate of this wave's best candidate
  // 4x4 / 8x8: the (mode x tx type) trials of the block sixteen at a time, four per wave (one per 16-lane row, dev_group.h):
  // ONE round for the 3 x 5 trials of speed 4, two for 3 x 7, four for 7 x 7.  Between rounds a wave parks its best
  // candidate's reconstruction and levels in S->dcp (idle during the luma trials).
  bool grouped = false, parked = false; int my_g = 0;
  if constexpr (SMALL_GROUPED) {
    grouped = pred_cached;
    if (grouped) {
      const int g = GROUP_ID, total = ncand * ntx, rounds = (total + 15) >> 4;
      LDS uint16_t *park_rec = (LDS uint16_t *)S->dcp; LDS int32_t *park_qc = (LDS int32_t *)(S->dcp + 64);
      for (int rd = 0; rd < rounds; rd++) {
        int e;
        // 3 x 5 trials: a wave's rows 0..2 hold one of the four DCT / ADST combinations for the three modes, so each 1-D pass of the wave walks ONE butterfly network;
        // the three identity trials (a scaling, no network) ride in row 3 of waves 0..2.  (Until round 5 wave 0 held the identity trials and row 3 the fifth combination:
        // three of the four waves then walked both networks in one of their passes -- 10 + 10 network walks per block instead of 8 + 8.)
        if (ncand == 3 && ntx == 5) e = g < 3 ? g * 5 + W + 1 : (W < 3 ? W * 5 : -1);
        else e = rd * 16 + W * 4 + g;
        const bool live = e >= 0 && e < total;
        const int ee = live ? e : 0, ci = ee / ntx, ti = ee - ci * ntx, m = SH->order[ci];
        const int delta = SH->ldelta[ci];
        const uint32_t mode_rate = y_mode_rate(k.cost(), ycost, m, m >= V_PRED && m <= D67_PRED && BS >= BS_8, delta);
        int ns2, set2;
        const int tx_off = intra_tx_cdf_r(f, Tools<TS>::reduced_tx_set(f), BS, m, &ns2, &set2);
        int txtype;
        if (ntx > 1) txtype = sym_to_txtype(tx_set, ti);
        else { txtype = mode_to_txtype(m); if (tx_off < 0 || txtype_to_sym(tx_set, txtype) < 0) txtype = DCT_DCT; }
        GroupRes gr;
        eval_group<n>(k.cc(), k.cost(), k.ls(), f, &S->grp[g], SH->srcb[0], SH->lpred + ci * nn, 0, BS, txtype, sctx_y, dctx_y, tx_off,
                      tx_off >= 0 ? txtype_to_sym(tx_set, txtype) : 0, Tools<TS>::tune_psnr(f) ? -1 : SH->psv[0], SH->pact[0], &gr);
        long long j = rd_dist32(f, 0, gr.sse) + rd_rate32(f, gr.rate) + rd_rate32(f, mode_rate);
        if (!live) j = J_INF;
        bool improved = false;
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
          const long long jg = ((long long)__builtin_amdgcn_readlane((int)(j >> 32), gg * 16) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)j, gg * 16);
          const int eg = __builtin_amdgcn_readlane(e, gg * 16);
          if (jg < my_j || (jg == my_j && eg < my_e)) {
            my_j = jg; my_e = eg; my_g = gg; improved = true;
            my_mode = __builtin_amdgcn_readlane(m, gg * 16); my_delta = __builtin_amdgcn_readlane(delta, gg * 16); my_tx = __builtin_amdgcn_readlane(txtype, gg * 16);
            my_tr.eob = __builtin_amdgcn_readlane(gr.eob, gg * 16); my_tr.cul = __builtin_amdgcn_readlane(gr.cul, gg * 16); my_tr.dcc = __builtin_amdgcn_readlane(gr.dcc, gg * 16);
            my_mrate = (uint32_t)__builtin_amdgcn_readlane((int)mode_rate, gg * 16);
          }
        }
        if (rounds > 1 && improved) {                      // wave-uniform
          for (int i = LANE; i < nn; i += 64) { park_rec[i] = S->grp[my_g].rec[i]; park_qc[i] = S->grp[my_g].qc[i]; }
          parked = true;
          WAVE_SYNC();
        }
      }
    }
  }
  if (!grouped)
  for (int e = W; e < ncand * ntx; e += NW) {
    const int ci = e / ntx, ti = e - ci * ntx, m = SH->order[ci];
    const int directional = m >= V_PRED && m <= D67_PRED;
    int delta = 0;
    const LDS uint16_t *lpred = S->pred;
    if (pred_cached) { delta = SH->ldelta[ci]; lpred = SH->lpred + ci * nn; }
    else {
      if (directional && refine) {
        long long bsd = SH->satd[m];
        for (int q = 0; q < 6; q++) { const long long sd = SH->dsd[ci][q]; if (sd < bsd) { bsd = sd; delta = dl_of(q); } }
      }
      predict_block(f, x, y, log2w, availL, availU, m, delta, ftype_y, ra, rl, wa, wl, S->etmp, S->pred);
    }
    const uint32_t mode_rate = y_mode_rate(k.cost(), ycost, m, directional && BS >= BS_8, delta);
    int ns2, set2;
    const int tx_off = intra_tx_cdf_r(f, Tools<TS>::reduced_tx_set(f), BS, m, &ns2, &set2);
    int txtype;
    if (ntx > 1) txtype = sym_to_txtype(tx_set, ti);
    else { txtype = mode_to_txtype(m); if (tx_off < 0 || txtype_to_sym(tx_set, txtype) < 0) txtype = DCT_DCT; }
    TxRes tr;
    long long j = eval_tx<MAXN, BS, NW>(k, 0, sctx_y, dctx_y, lpred, txtype, tx_off, tx_off >= 0 ? txtype_to_sym(tx_set, txtype) : 0, S->rec[cur], S->qc[cur], &tr);
    j += ((long long)mode_rate * f->rdmult + 256) >> 9;
    if (j < my_j) {
      my_j = j; my_e = e; my_mode = m; my_delta = delta; my_tx = txtype; my_tr = tr; my_mrate = mode_rate;
      cur ^= 1;
    }
  }
  if (LANE == 0) { SH->wbest_j[W] = my_j; SH->wbest_e[W] = my_e; }
  PH(6);
  WG_SYNC();
  PH(2);
  int win = 0;
  for (int w2 = 1; w2 < NW; w2++) if (SH->wbest_j[w2] < SH->wbest_j[win] || (SH->wbest_j[w2] == SH->wbest_j[win] && SH->wbest_e[w2] < SH->wbest_e[win])) win = w2;
  const long long best_j = SH->wbest_j[win];
  if (W == win) {
    const int b = cur ^ 1;                                  // buffer holding this wave's best
    const LDS uint16_t *best_rec = S->rec[b]; const LDS int32_t *best_qc = S->qc[b];
    if constexpr (SMALL_GROUPED) if (grouped) {
      if (parked) { best_rec = (const LDS uint16_t *)S->dcp; best_qc = (const LDS int32_t *)(S->dcp + 64); }
      else { best_rec = S->grp[my_g].rec; best_qc = S->grp[my_g].qc; }
    }
    commit_plane<BS>(f, 0, r, c, best_rec, best_qc, my_tr.eob, my_tr.cul, my_tr.dcc);
    fill_map_dev(f->m_ymode, ms, r, c, n4, my_mode);
    fill_map_dev((uint8_t *)f->m_angle_y, ms, r, c, n4, (uint8_t)(int8_t)my_delta);
    fill_map_dev(f->m_txtype, ms, r, c, n4, my_tr.eob ? my_tx : DCT_DCT);
    fill_map_dev(f->m_bsize, ms, r, c, n4, BS);
    fill_map_dev(f->m_txsize, ms, r, c, n4, BS);
    if (f->np > 1) for (int i = LANE; i < nn; i += 64) SH->luma_rec[i] = best_rec[i];
    if (LANE == 0) {
      SH->lm_mode = my_mode; SH->lm_delta = my_delta; SH->lm_tx = my_tx; SH->lm_eob = my_tr.eob;
      SH->lm_mode_j = ((long long)my_mrate * f->rdmult + 256) >> 9;
    }
  }
  PH(7);
  WG_SYNC();
  PH(2);
  if (DBG_IS(f, 5)) return 0;
  const int best_mode = SH->lm_mode, best_delta = SH->lm_delta;
  long long luma_j = best_j; int any_coef = SH->lm_eob > 0;
  // ---- luma transform size (rav1e rdo_tx_size_type under TX_MODE_SELECT; oracle/av1o_search.c try_block): the largest
  // transform against four transforms one level smaller with the winning mode.  Each sub-block is predicted from the
  // reconstruction of the ones before it (spec transform_block), so the four steps are serial; inside a step the tx types
  // of the sub-block are dealt to the waves (16-lane rows for 4x4 / 8x8 transforms) like the candidates of a block.
  if constexpr (BS > 0) if (Tools<TS>::tx_mode_select(f)) {
    const int maxw = 4 << BS;
    const int actx = nb_txU >= 0 && (1 << dim_wl(nb_txU)) >= maxw, lctx = nb_txL >= 0 && (1 << dim_hl(nb_txL)) >= maxw;   // neighbours may carry 2:1 transform codes
    const uint16_t *dcost = k.cost() + CDF_TX_SIZE + ((BS - 1) * 3 + actx + lctx) * CDF_TX_SIZE_STRIDE;
    luma_j += ((long long)dcost[0] * f->rdmult + 256) >> 9;
    if (tx_trial) {
      // The trial of one depth D (1: 2x2 transform blocks one level smaller, 2: 4x4 blocks two levels smaller, raster order), LDS-resident:
      // returns true when the caller's budget is exhausted (try_block returns at once).  The frame always holds the best transform
      // size found so far (depth 0 was committed above; a winning depth replaces it), so a deeper trial needs no snapshot.
      LDS int *const sub_tx = (LDS int *)SH->dsd, *const sub_eob = sub_tx + 16, *const sub_cul = sub_tx + 32, *const sub_dcc = sub_tx + 48, *const psv16 = sub_tx + 64;   // dsd / satd are dead after the mode decision
      LDS int *const sfl_r = (LDS int *)SH->satd, *const sfl_b = sfl_r + 4;
      LDS uint16_t *ssrc = (LDS uint16_t *)SH->ssrc;         // the sub-sources, sub-block after sub-block
      auto trial = [&](auto depth_c) -> bool {
        constexpr int D = decltype(depth_c)::value;
        constexpr int SBS = BS - D, G = 1 << D, hn = n >> D, half = n4 >> D, hnn = hn * hn, SCS = hn < 32 ? hn : 32, sqn = SCS * SCS;
        long long j_split = SH->lm_mode_j + (((long long)dcost[D] * f->rdmult + 256) >> 9);
        int stx_ns = 0, stx_set = 0;
        const int stx_off = intra_tx_cdf_r(f, Tools<TS>::reduced_tx_set(f), SBS, best_mode, &stx_ns, &stx_set);
        const int sntx = stx_off >= 0 ? stx_ns : 1;
        int sub_any = 0;
#if MI_PROFILE
        const unsigned long long tr_t0_ = clock64();
        auto tr_done_ = [&]() { if (LANE == 0) SH->prof[W][BS == 1 ? 31 : (D == 1 ? 29 : 30)] += clock64() - tr_t0_; };
#else
        auto tr_done_ = []() {};
#endif
        PH(13);
        // ---- stage 0, all waves: the sub-sources; wave 0: outer neighbour contexts, the 4x4 source variances (depth 2 of a 16x16 block)
        // and the decoded flags of the cells right of / below the block that a sub-block's above-right / below-left edge may reach
        for (int q = W; q < G * G; q += NW) {
          const int so = (q / G) * hn * n + (q % G) * hn;
          for (int idx = LANE; idx < hnn; idx += 64) ssrc[q * hnn + idx] = SH->srcb[0][so + (idx / hn) * n + (idx % hn)];
        }
        if (W == 0) {
          if (LANE < n4) {
            const int k2 = LANE;
            const bool ha = availU && c + k2 < f->mi_cols, hl = availL && r + k2 < f->mi_rows;         // unconditional loads, one batch
            const int ia = ha ? (r - 1) * ms + c + k2 : mi, il = hl ? (r + k2) * ms + c - 1 : mi;
            const int la = f->m_lvl[0][ia], da = f->m_dc[0][ia], ll = f->m_lvl[0][il], dl2 = f->m_dc[0][il];
            SH->nb_top[k2][0] = (uint8_t)(ha ? la : 0); SH->nb_top[k2][1] = (uint8_t)(ha ? da : 0);
            SH->nb_left[k2][0] = (uint8_t)(hl ? ll : 0); SH->nb_left[k2][1] = (uint8_t)(hl ? dl2 : 0);
          }
          if (hn == 4 && n == 16 && LANE < 16) psv16[LANE] = (int)f->svar4[(r + (LANE >> 2)) * ms + c + (LANE & 3)];
          if (LANE >= 1 && LANE < G) {
            const bool ca = c + n4 < t->mi_col_end, cb = r + n4 < t->mi_row_end;
            sfl_r[LANE] = ca && decoded_before(r, c, 0, r + LANE * half - 1, c + n4); sfl_b[LANE] = cb && decoded_before(r, c, 0, r + n4, c + LANE * half - 1);
          }
        }
        WG_SYNC();
        PH(22);
        // (the block's wave-uniform geometry under second names: the chain loop below re-declares the first ones as scalars)
        [[maybe_unused]] const int x_ = x, y_ = y, r_ = r, c_ = c, availU_ = availU, availL_ = availL, have_ar_ = have_ar, have_bl_ = have_bl, best_mode_ = best_mode, best_delta_ = best_delta, ftype_y_ = ftype_y;
        if constexpr (SBS <= BS_8 && NW == 4 && MAXN <= 32) {
          // 4x4 / 8x8 sub-blocks: one CHAIN per transform type (rav1e rdo_tx_type_decision: the whole block with one type), each on a 16-lane row: the row prepares its
          // sub-block's raw edges from its own chain's reconstructions, predicts it, evaluates it with its type, keeps reconstruction / levels / contexts in the chain's
          // state and moves on -- no workgroup barrier until every chain is through.  Five types (the reduced set): the four DCT / ADST combinations on wave 0's rows,
          // IDTX alone on wave 1 (a wave whose rows mix identity, DCT and ADST walks all three 1-D networks); up to eight types: four per wave on waves 0 and 1.  The
          // chains' state lies in the scratch of waves 2 / 3, which have nothing to do here.  A chain whose running cost reaches the best so far (or the caller's
          // budget) is dead: its row runs on harmlessly while another row of the wave is alive.  (Until round 6 every sub-block picked its own type: slots of
          // predict -> barrier -> evaluate -> barrier -> pick + copy -> barrier, profiles/r05i_k1_phase_profile.txt: 13 % of the kernel in those barriers.)
          constexpr int CH_BYTES = nn * 2 + G * G * sqn * 4 + G * G * 4;     // a chain: the block's reconstruction, the sub-blocks' levels, eob | cul << 16 | dcc << 24 per sub-block
          static_assert(4 * CH_BYTES <= (int)sizeof(WaveScratch<MAXN>) && hnn <= 64 && 4 * 64 <= MAXN * MAXN, "four chains fit a wavefront's scratch; edges and predictions of four rows fit pred / dcp");
          const int g = GROUP_ID, gl = GROUP_LANE;
          // PAIRED chains (the 16x16 class with the reduced set's five types): a chain runs on TWO rows of one wave -- the skewed wavefront over the sub-block grid that the
          // per-sub-block search of rounds 3-5 ran on two wave pairs (a sub-block needs its left neighbour and the row above up to its above-right neighbour: slot = bj + 2 bi,
          // 10 slots for the 16 sub-blocks of a 4 x 4 grid; (0), (1, 2), (3) for a 2 x 2 grid whose prediction reads no above-right samples), now inside a chain: the
          // pair's rows share the chain's state, a wave barrier orders the slots.  Types 1, 2 on wave 0, types 3, 4 on wave 1, IDTX on wave 2 (two rows, two idle); the
          // chains' state: types 1..4 in wave 3's scratch, IDTX directly in the block's split buffers.  16 serial steps -> 10 slots, 4 -> 3.
          const bool paired = MI_K1_PAIRED_CHAINS && MAXN == 16 && sntx == 5 && n >= MI_K1_PAIRED_MIN_N && hn <= MI_K1_PAIRED_MAX_HN;
          const int mem = paired ? (g & 1) : 0;                                                          // which sub-block of a slot this row takes
          const int e = paired ? (W == 0 ? 1 + (g >> 1) : (W == 1 ? 3 + (g >> 1) : ((W == 2 && g < 2) ? 0 : 64)))
                               : (sntx == 5 ? (W == 0 ? g + 1 : ((W == 1 && g == 0) ? 0 : 64)) : W * 4 + g);   // this row's transform type (symbol)
          const bool has_chain = e < sntx, wave_has = paired ? W < 3 : (sntx == 5 ? W < 2 : W * 4 < sntx);  // row-uniform, wave-uniform
          LDS long long *const cres_j = (LDS long long *)SH->cj;                                          // [8] complete chains' costs (the chroma costs' place: dead during luma)
          LDS int *const cres_any = (LDS int *)(cres_j + 8);                                              // [8]
          LDS uint32_t *const meta_idtx = (LDS uint32_t *)SH->cj + 32;                                    // [16] the paired IDTX chain's contexts
          const long long thr = uni64(luma_j < budget ? luma_j : budget);                                // a chain at or above it can neither win nor keep the block below the budget
          // where chain `ce` of this trial keeps its state
          auto chain_state = [&](int ce, int cw, int cg, LDS uint16_t **cv, LDS int32_t **cq, LDS uint32_t **cm) {
            if (paired && ce == 0) { *cv = split_rec; *cq = split_qc; *cm = meta_idtx; return; }
            LDS uint8_t *cst = paired ? (LDS uint8_t *)k.wave(3) + (ce >= 1 && ce <= 4 ? ce - 1 : 0) * CH_BYTES : (LDS uint8_t *)k.wave(2 + cw) + cg * CH_BYTES;
            *cv = (LDS uint16_t *)cst; *cq = (LDS int32_t *)(cst + nn * 2); *cm = (LDS uint32_t *)(cst + nn * 2 + G * G * sqn * 4);
          };
          if (wave_has) {
            // wave-uniform inputs of the chain loop, made scalar: the loop's many divergent regions (row-local edge loops, per-row contexts) otherwise carry them in VGPRs
            const int x = uni32(x_), y = uni32(y_), r = uni32(r_), c = uni32(c_), availU = uni32(availU_), availL = uni32(availL_), have_ar = uni32(have_ar_), have_bl = uni32(have_bl_);
            const int best_mode = uni32(best_mode_), best_delta = uni32(best_delta_), ftype_y = uni32(ftype_y_);
            LDS uint16_t *canvas; LDS int32_t *cqc; LDS uint32_t *cmeta;
            chain_state(e, W, g, &canvas, &cqc, &cmeta);
            LDS uint16_t *A = S->pred + g * 64 + EDGE_OFF, *Lf = A + 32, *ppred = S->dcp + g * 64;
            int txtype;
            if (sntx > 1) txtype = sym_to_txtype(stx_set, has_chain ? e : 0);
            else { txtype = mode_to_txtype(best_mode); if (stx_off < 0 || txtype_to_sym(stx_set, txtype) < 0) txtype = DCT_DCT; }
            const int tx_sym = stx_off >= 0 ? txtype_to_sym(stx_set, txtype) : 0;
            const bool dirm = best_mode >= V_PRED && best_mode <= D67_PRED;
            const int pa = dirm ? mode_angle_of(best_mode) + 3 * best_delta : 0;
            const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1, rs = f->stride, bd = f->bd;
            const uint16_t *grec = f->rec[0];
            long long jc = has_chain ? j_split : J_INF;
            int any = 0;
            if (W < MI_K1_CHAIN_PRIO_WAVES) __builtin_amdgcn_s_setprio(3);   // off (0 waves): with chains, issue priority for the chain-carrying waves COSTS 1.4 % (profiles/r06p_ab_chain_priority.txt; it gained 1 % for round 5's slots)
            const bool uses_ar = dirm && pa < 90;
            const int nslot = paired ? (G == 4 ? 10 : (uses_ar ? 4 : 3)) : G * G;
#pragma unroll 1
            for (int slot = 0; slot < nslot; slot++) {
              if (MI_BALLOT64(jc < thr) == 0ull) break;          // every chain of this wave is dead
              // the slot's sub-blocks: q0 for the pair's first row, q1 for its second (-1: none; that row then repeats q0: same chain, same inputs, same stores)
              int q0 = slot, q1 = -1;
              if (paired) {
                if (G == 4) { const int lo = imax_(0, (slot - 2) >> 1), hi = imin_(3, slot >> 1); q0 = lo * 4 + slot - 2 * lo; q1 = lo + 1 <= hi ? (lo + 1) * 4 + slot - 2 * (lo + 1) : -1; }
                else if (!uses_ar) { q0 = slot == 0 ? 0 : (slot == 1 ? 1 : 3); q1 = slot == 1 ? 2 : -1; }
              }
              const int q = (mem == 1 && q1 >= 0) ? q1 : q0;
              const int bi = q / G, bj = q % G;
              const int sx = x + bj * hn, sy = y + bi * hn;
              const int sU = availU || bi, sL = availL || bj;
              {
                // raw edges of the sub-block (spec 7.11.2; load_edges with the samples taken from the block's raw edges, this chain's sub-blocks so far, or -- beyond the
                // block's right / bottom edge -- the frame), 16 lanes per chain
                const int s_ar = bi == 0 ? (bj < G - 1 ? availU : have_ar) : (bj < G - 1 ? 1 : sfl_r[bi]);
                const int s_bl = bj == 0 ? (bi < G - 1 ? availL : have_bl) : (bi < G - 1 ? 0 : sfl_b[bj]);
                const int lim_a = imin_(max_x, sx + (s_ar ? 2 * hn : hn) - 1), lim_l = imin_(max_y, sy + (s_bl ? 2 * hn : hn) - 1);
                // (straight-line: one LDS load from a selected pointer per sample -- this chain's reconstructions, the block's raw above edge or its raw left edge --,
                // the frame only for the rare samples right of / below the block)
                auto px = [&](int ax, int ay) -> int {
                  const int xr = ax - x, yr = ay - y;
                  const bool in_blk = xr >= 0 && xr < n && yr >= 0 && yr < n, on_a = yr == -1 && xr >= -1 && xr < 2 * n, on_l = xr == -1 && yr >= 0 && yr < 2 * n;
                  const LDS uint16_t *p = in_blk ? (const LDS uint16_t *)canvas + (yr * n + xr) : (on_a ? ra + xr : rl + (on_l ? yr : 0));
                  int v = (int)*p;
                  if (!(in_blk || on_a || on_l)) v = (int)grec[(size_t)ay * rs + ax];
                  return v;
                };
#pragma unroll
                for (int it = 0; it < (2 * hn + 16) / 16; it++) {
                  const int i = gl + 16 * it;
                  const bool act = i <= 2 * hn, corner = i == 2 * hn;
                  const int ii = act ? i : 0;                     // (idle lanes compute a harmless sample)
                  int a, l;
                  if (sU) a = px(corner ? (sL ? sx - 1 : sx) : imin_(lim_a, sx + ii), sy - 1); else a = px(sL ? sx - 1 : sx, sy);
                  if (sL) l = px(sx - 1, imin_(lim_l, sy + ii)); else l = px(sx, sU ? sy - 1 : sy);
                  if (!sU && !sL) { a = corner ? (1 << (bd - 1)) : (1 << (bd - 1)) - 1; l = (1 << (bd - 1)) + 1; }
                  if (act && corner) { A[-1] = (uint16_t)a; Lf[-1] = (uint16_t)a; }
                  if (act && !corner) { A[ii] = (uint16_t)a; Lf[ii] = (uint16_t)l; }
                }
              }
              WAVE_SYNC();
              if (dirm) predict_dir_group<hn>(f, sx, sy, sL, sU, pa, ftype_y, A, Lf, &S->gpred[g], ppred);
              else predict_nondir_group<hn>(best_mode, sL, sU, bd, A, Lf, ppred);
              PH(23);
              // all_zero / dc_sign contexts of the sub-block (txb_ctx_dev with bs != txs): outer neighbours from the staged contexts, inner ones from this chain
              int ssc, sdc;
              {
                int top = 0, left = 0, dcs = 0;
                const uint32_t mt = bi ? cmeta[q - G] : 0u, ml = bj ? cmeta[q - 1] : 0u;
#pragma unroll
                for (int k2 = 0; k2 < half; k2++) {
                  int l, d;
                  if (bi == 0) { l = SH->nb_top[bj * half + k2][0]; d = SH->nb_top[bj * half + k2][1]; } else { l = (int)((mt >> 16) & 0xFF); d = (int)(mt >> 24); if (n >= 32 && c + bj * half + k2 >= f->mi_cols) l = d = 0; }
                  top = imax_(top, l); dcs += d == 1 ? -1 : (d == 2 ? 1 : 0);
                  if (bj == 0) { l = SH->nb_left[bi * half + k2][0]; d = SH->nb_left[bi * half + k2][1]; } else { l = (int)((ml >> 16) & 0xFF); d = (int)(ml >> 24); if (n >= 32 && r + bi * half + k2 >= f->mi_rows) l = d = 0; }
                  left = imax_(left, l); dcs += d == 1 ? -1 : (d == 2 ? 1 : 0);
                }
                sdc = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
                if (top == 0 && left == 0) ssc = 1;
                else if (top == 0 || left == 0) ssc = 2 + (imax_(top, left) > 3);
                else if (imax_(top, left) <= 3) ssc = 4;
                else if (imin_(top, left) <= 3) ssc = 5;
                else ssc = 6;
              }
              constexpr int pcp = n >= 8 ? n / 8 : 1;
              const int psv_q = hn == 4 ? (n == 8 ? SH->psv4[q] : psv16[q]) : SH->psv[bi * pcp + bj];
              const int pact_q = hn == 4 ? SH->pact[(bi >> 1) * pcp + (bj >> 1)] : SH->pact[bi * pcp + bj];
              GroupRes gr;
              eval_group<hn>(k.cc(), k.cost(), k.ls(), f, &S->grp[g], ssrc + q * hnn, ppred, 0, SBS, txtype, ssc, sdc, stx_off, tx_sym, Tools<TS>::tune_psnr(f) ? -1 : psv_q, pact_q, &gr);
              {
                const long long j = rd_dist32(f, 0, gr.sse) + rd_rate32(f, gr.rate);
                long long jslot = j;
                if (paired) {                                    // both rows of a pair add both sub-blocks' costs: the chain's cost is the same number in either
                  const long long jo = ((long long)__shfl((int)(j >> 32), LANE ^ 16) << 32) | (unsigned int)__shfl((int)j, LANE ^ 16);
                  jslot = q1 >= 0 ? j + jo : (mem == 0 ? j : jo);
                }
                if (has_chain) jc += jslot;
              }
              // the chain keeps the sub-block: reconstruction (the next sub-blocks' edges), levels, contexts
              if (has_chain) {
                const LDS uint16_t *srec = S->grp[g].rec; const LDS int32_t *sqc = S->grp[g].qc;
                const int ro = bi * hn * n + bj * hn;
#pragma unroll
                for (int i = gl; i < hnn; i += 16) canvas[ro + (i / hn) * n + (i % hn)] = srec[i];
#pragma unroll
                for (int i = gl; i < sqn; i += 16) cqc[q * sqn + i] = sqc[i];
                cmeta[q] = (uint32_t)gr.eob | ((uint32_t)gr.cul << 16) | ((uint32_t)gr.dcc << 24);      // (row-uniform: every lane of the row stores the same word)
              }
              any |= gr.eob > 0;
              WAVE_SYNC();
              PH(25);
            }
            // (a loop that ended early left every chain of the wave at or above thr; one that ran through left complete costs)
            // Every lane stores, rows without a chain into a slot nobody reads: NO divergent region may end together with this wave-level one.  With
            // `if (gl == 0 && has_chain) { ... }` as the region's last statement hipcc (ROCm 7.2) narrows EXEC for the inner `if` without saving it (the outer
            // region's restore follows anyway) and then places rematerialised VALU instructions and spill reloads of LATER code between the two -- computed in
            // lane 0 of each row only (seen on gfx950: the row index of the next trial's edge buffers; the CPU emulator cannot see it).
            if (paired) any |= __shfl(any, LANE ^ 16);
            const int rslot = paired ? (has_chain ? e : 5 + (g & 1)) : (sntx == 5 ? (W == 0 ? g + 1 : (g == 0 ? 0 : 4 + g)) : W * 4 + g);
            cres_j[rslot] = (has_chain && jc < thr) ? jc : J_INF; cres_any[rslot] = any;
          }
          __builtin_amdgcn_s_setprio(0);
          WG_SYNC();
          PH(26);
          int be = -1; long long bjc = J_INF;
          for (int e2 = 0; e2 < sntx; e2++) { const long long v = cres_j[e2]; if (v < bjc) { bjc = v; be = e2; } }     // the cheapest complete chain, the lowest symbol among equals
          j_split = bjc;
          if (be >= 0) {                                          // (uniform over the workgroup: LDS values) the winner's state -> the block's split buffers, for the commit below
            const int ow = paired ? 0 : (sntx == 5 ? (be == 0 ? 1 : 0) : be >> 2), og = sntx == 5 ? (be == 0 ? 0 : be - 1) : be & 3;
            if (W == ow) {
              LDS uint16_t *canvas; LDS int32_t *cqc; LDS uint32_t *cmeta;
              chain_state(be, ow, og, &canvas, &cqc, &cmeta);
              if (canvas != split_rec) {                         // (the paired IDTX chain already lives in the split buffers)
                for (int i = LANE; i < nn; i += 64) split_rec[i] = canvas[i];
                for (int i = LANE; i < G * G * sqn; i += 64) split_qc[i] = cqc[i];
              }
              int btx;
              if (sntx > 1) btx = sym_to_txtype(stx_set, be);
              else { btx = mode_to_txtype(best_mode); if (stx_off < 0 || txtype_to_sym(stx_set, btx) < 0) btx = DCT_DCT; }
              if (LANE < G * G) { const uint32_t m = cmeta[LANE]; const int eob = (int)(m & 0xFFFF); sub_eob[LANE] = eob; sub_cul[LANE] = (int)((m >> 16) & 0xFF); sub_dcc[LANE] = (int)(m >> 24); sub_tx[LANE] = eob ? btx : DCT_DCT; }
            }
            sub_any = cres_any[be];
            WG_SYNC();
          }
          PH(27);
        } else {
          // larger sub-blocks: one chain per wavefront and round (transform type rd * NW + W), every wave on its own -- raw edges from its chain's reconstructions,
          // prediction, evaluation, all in its private scratch (canvas = rec[1], levels = qc[1], prediction = dcp) --, a barrier only between rounds: the best
          // complete chain so far moves into the block's split buffers, the next round's chains must beat it (a later type never wins a tie).
          static_assert(nn <= (int)(sizeof(S->rec[1]) / 2) && G * G * sqn <= (int)(sizeof(S->qc[1]) / 4) && G * G <= 16, "a chain's state fits the wave's second candidate buffers");
          LDS uint16_t *canvas = S->rec[1]; LDS int32_t *cqc = S->qc[1]; LDS uint32_t *cmeta = (LDS uint32_t *)SH->cj + W * 16;
          LDS uint16_t *A = S->pred + EDGE_OFF, *Lf = S->pred + (MAXN * MAXN / 2) + EDGE_OFF;
          LDS int *ppsv = (LDS int *)S->etmp, *ppact = ppsv + 16;           // (etmp is predict_block's temporary: free between two predictions)
          static_assert(sizeof(S->etmp) >= 32 * sizeof(int), "the sub-block's psychovisual references fit etmp");
          const int max_x = f->mi_cols * 4 - 1, max_y = f->mi_rows * 4 - 1, rs = f->stride, bd = f->bd;
          const uint16_t *grec = f->rec[0];
          const long long j0 = j_split;
          long long best_chain = J_INF;
          sub_any = 0;
#pragma unroll 1
          for (int rd = 0; rd * NW < sntx; rd++) {
            const int e = rd * NW + W;
            const bool has_chain = e < sntx;
            long long thr = luma_j < budget ? luma_j : budget;
            if (best_chain < thr) thr = best_chain;
            int txtype;
            if (sntx > 1) txtype = sym_to_txtype(stx_set, has_chain ? e : 0);
            else { txtype = mode_to_txtype(best_mode); if (stx_off < 0 || txtype_to_sym(stx_set, txtype) < 0) txtype = DCT_DCT; }
            const int tx_sym = stx_off >= 0 ? txtype_to_sym(stx_set, txtype) : 0;
            long long jc = has_chain ? j0 : J_INF;
            int any = 0;
            if (has_chain) {
#pragma unroll 1
              for (int q = 0; q < G * G; q++) {
                if (!(jc < thr)) break;                          // wave-uniform
                const int bi = q / G, bj = q % G;
                const int sx = x + bj * hn, sy = y + bi * hn;
                const int sU = availU || bi, sL = availL || bj;
                {
                  const int s_ar = bi == 0 ? (bj < G - 1 ? availU : have_ar) : (bj < G - 1 ? 1 : sfl_r[bi]);
                  const int s_bl = bj == 0 ? (bi < G - 1 ? availL : have_bl) : (bi < G - 1 ? 0 : sfl_b[bj]);
                  const int lim_a = imin_(max_x, sx + (s_ar ? 2 * hn : hn) - 1), lim_l = imin_(max_y, sy + (s_bl ? 2 * hn : hn) - 1);
                  auto px = [&](int ax, int ay) -> int {                  // absolute sample position -> value
                    const int xr = ax - x, yr = ay - y;
                    if (xr >= 0 && xr < n && yr >= 0 && yr < n) return (int)canvas[yr * n + xr];
                    if (yr == -1 && xr >= -1 && xr < 2 * n) return (int)ra[xr];
                    if (xr == -1 && yr >= 0 && yr < 2 * n) return (int)rl[yr];
                    return (int)grec[(size_t)ay * rs + ax];
                  };
                  for (int i = LANE; i <= 2 * hn; i += 64) {
                    const bool corner = i == 2 * hn;
                    int a, l;
                    if (sU) a = px(corner ? (sL ? sx - 1 : sx) : imin_(lim_a, sx + i), sy - 1); else a = px(sL ? sx - 1 : sx, sy);
                    if (sL) l = px(sx - 1, imin_(lim_l, sy + i)); else l = px(sx, sU ? sy - 1 : sy);
                    if (!sU && !sL) { a = corner ? (1 << (bd - 1)) : (1 << (bd - 1)) - 1; l = (1 << (bd - 1)) + 1; }
                    if (corner) { A[-1] = (uint16_t)a; Lf[-1] = (uint16_t)a; } else { A[i] = (uint16_t)a; Lf[i] = (uint16_t)l; }
                  }
                  WAVE_SYNC();
                  predict_block(f, sx, sy, log2w - D, sL, sU, best_mode, best_delta, ftype_y, A, Lf, wa, wl, S->etmp, S->dcp);
                }
                PH(23);
                // all_zero / dc_sign contexts of the sub-block (txb_ctx_dev with bs != txs): outer neighbours from the staged contexts, inner ones from this chain
                int ssc, sdc;
                {
                  int top = 0, left = 0, dcs = 0;
                  const uint32_t mt = bi ? cmeta[q - G] : 0u, ml = bj ? cmeta[q - 1] : 0u;
#pragma unroll
                  for (int k2 = 0; k2 < half; k2++) {
                    int l, d;
                    // (a 32x32 block may reach past the frame's last 8x8 column / row: the oracle's av1o_txb_ctx leaves out neighbour cells beyond the frame; smaller blocks never do)
                    if (bi == 0) { l = SH->nb_top[bj * half + k2][0]; d = SH->nb_top[bj * half + k2][1]; } else { l = (int)((mt >> 16) & 0xFF); d = (int)(mt >> 24); if (n >= 32 && c + bj * half + k2 >= f->mi_cols) l = d = 0; }
                    top = imax_(top, l); dcs += d == 1 ? -1 : (d == 2 ? 1 : 0);
                    if (bj == 0) { l = SH->nb_left[bi * half + k2][0]; d = SH->nb_left[bi * half + k2][1]; } else { l = (int)((ml >> 16) & 0xFF); d = (int)(ml >> 24); if (n >= 32 && r + bi * half + k2 >= f->mi_rows) l = d = 0; }
                    left = imax_(left, l); dcs += d == 1 ? -1 : (d == 2 ? 1 : 0);
                  }
                  sdc = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
                  if (top == 0 && left == 0) ssc = 1;
                  else if (top == 0 || left == 0) ssc = 2 + (imax_(top, left) > 3);
                  else if (imax_(top, left) <= 3) ssc = 4;
                  else if (imin_(top, left) <= 3) ssc = 5;
                  else ssc = 6;
                }
                // psychovisual references of the sub-block: its own variance (4x4) or its 8x8 cells', and the activity of the 8x8 cells it lies in
                constexpr int pcp = n >= 8 ? n / 8 : 1, scp = hn >= 8 ? hn / 8 : 1;
                if (LANE < scp * scp) {
                  if constexpr (hn == 4) { ppsv[0] = n == 8 ? SH->psv4[q] : psv16[q]; ppact[0] = SH->pact[(bi >> 1) * pcp + (bj >> 1)]; }
                  else { const int pc = (bi * scp + LANE / scp) * pcp + bj * scp + LANE % scp; ppsv[LANE] = SH->psv[pc]; ppact[LANE] = SH->pact[pc]; }
                }
                WAVE_SYNC();
                TxRes tr;
                jc += eval_tx<MAXN, SBS, NW>(k, 0, ssc, sdc, S->dcp, txtype, stx_off, tx_sym, S->rec[0], S->qc[0], &tr, ssrc + q * hnn, (const LDS int *)ppsv, (const LDS int *)ppact);
                {
                  const int ro = bi * hn * n + bj * hn;
                  for (int i = LANE; i < hnn; i += 64) canvas[ro + (i / hn) * n + (i % hn)] = S->rec[0][i];
                  for (int i = LANE; i < sqn; i += 64) cqc[q * sqn + i] = S->qc[0][i];
                  if (LANE == 0) cmeta[q] = (uint32_t)tr.eob | ((uint32_t)tr.cul << 16) | ((uint32_t)tr.dcc << 24);
                }
                any |= tr.eob > 0;
                WAVE_SYNC();
                PH(25);
              }
            }
            if (LANE == 0) { SH->wbest_j[W] = jc < thr ? jc : J_INF; SH->wbest_e[W] = any; }
            WG_SYNC();
            PH(26);
            int sw = 0;
            for (int w2 = 1; w2 < NW; w2++) if (SH->wbest_j[w2] < SH->wbest_j[sw]) sw = w2;          // the lowest wave = the lowest symbol among equals
            const long long rj = SH->wbest_j[sw];
            if (rj < best_chain) {                               // (thr made it strictly smaller than every earlier round's)
              best_chain = rj; sub_any = SH->wbest_e[sw];
              if (W == sw) {
                for (int i = LANE; i < nn; i += 64) split_rec[i] = canvas[i];
                for (int i = LANE; i < G * G * sqn; i += 64) split_qc[i] = cqc[i];
                if (LANE < G * G) { const uint32_t m = cmeta[LANE]; const int eob = (int)(m & 0xFFFF); sub_eob[LANE] = eob; sub_cul[LANE] = (int)((m >> 16) & 0xFF); sub_dcc[LANE] = (int)(m >> 24); sub_tx[LANE] = eob ? txtype : DCT_DCT; }
              }
            }
            WG_SYNC();
            PH(27);
          }
          j_split = best_chain;
        }
        if (j_split < luma_j) {
          // this depth wins: its reconstruction, levels and contexts replace the best so far in the frame
          luma_j = j_split; any_coef = sub_any;
          const int tid = threadIdx.x, T = 64 * NW;
          uint16_t *gr_ = f->rec[0] + (size_t)y * f->stride + x;
          int32_t *gc_ = f->coef[0] + (size_t)y * f->stride + x;
          for (int i = tid; i < nn; i += T) { const uint16_t v = split_rec[i]; gr_[(i / n) * f->stride + (i % n)] = v; if (f->np > 1) SH->luma_rec[i] = v; }
          for (int i = tid; i < G * G * sqn; i += T) { const int q = i / sqn, j2 = i - q * sqn; gc_[((q / G) * hn + j2 / SCS) * f->stride + (q % G) * hn + j2 % SCS] = split_qc[i]; }
          if (W == 0) {
            fill_map_dev(f->m_txsize, ms, r, c, n4, SBS);
#pragma unroll 1
            for (int q = 0; q < G * G; q++) {
              const int rr = r + (q / G) * half, cc = c + (q % G) * half;
              fill_map_dev(f->m_lvl[0], ms, rr, cc, half, sub_cul[q]);
              fill_map_dev(f->m_dc[0], ms, rr, cc, half, sub_dcc[q]);
              fill_map_dev(f->m_txtype, ms, rr, cc, half, sub_tx[q]);
              if (LANE == 0) f->m_eob[0][rr * ms + cc] = (uint16_t)sub_eob[q];
            }
          }
        }
        WG_SYNC();
        PH(28);
        tr_done_();
        return false;
      };
      if (trial(std::integral_constant<int, 1>{})) return luma_j;
      if constexpr (BS >= BS_16 && MI_TX_DEPTH_MAX >= 2) if (trial(std::integral_constant<int, 2>{})) return luma_j;
    }
  }
  PH(13);
  if (luma_j >= budget) return luma_j;                      // wave-uniform: every wave reads the same LDS values
  long long total_j = luma_j;

  // ---- chroma, 4x4 / 8x8 blocks with the simple candidate set (DC, luma's mode, CfL): the CfL alpha scan on all four
  // waves (plane x half of the range), then every candidate of a plane in one grouped evaluation (dev_group.h) ----
  bool cgrouped = false;
  if constexpr (SMALL_GROUPED) cgrouped = f->np > 1 && !Tools<TS>::FULL;
  if constexpr (SMALL_GROUPED) if (cgrouped) {
    const uint16_t *uvcost = k.cost() + CDF_UV_CFL + best_mode * CDF_UV_CFL_STRIDE;
    const int nplain = best_mode != DC_PRED ? 2 : 1, nc = nplain + 1, uvset = tx_set_of(BS, Tools<TS>::reduced_tx_set(f));
    const int bdelta = (best_mode >= V_PRED && best_mode <= D67_PRED && BS >= BS_8) ? best_delta : 0;
    const int mx = (1 << f->bd) - 1;
    {
      // rdo_cfl_alpha (see the one-candidate path below for the scan order and the tie rule)
      const int p = (W >> 1) + 1, half = W & 1;
      const LDS uint16_t *luma = SH->luma_rec;
      int lsum = 0;
      for (int idx = LANE; idx < nn; idx += 64) lsum += luma[idx] << 3;
      lsum = wave_sum_i32(lsum);
      const int avg = round2_(lsum, 2 * log2w);
      predict_block(f, x, y, log2w, availL, availU, DC_PRED, 0, ftype_uv, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF, wa, wl, S->etmp, S->dcp);
      long long best_sse = J_INF; int best_idx = 1 << 20;
      if (half == 0) {
        int e0 = 0;
        for (int idx = LANE; idx < nn; idx += 64) { const int d = (int)SH->srcb[p][idx] - (int)S->dcp[idx]; e0 += __mul24(d, d); }
        best_sse = (long long)wave_sum_i32(e0); best_idx = -1;
      }
      int e[16];
#pragma unroll
      for (int a = 0; a < 16; a++) e[a] = 0;
      for (int idx = LANE; idx < nn; idx += 64) {
        const int l = ((int)luma[idx] << 3) - avg, dcv = S->dcp[idx], sv = SH->srcb[p][idx];
        const int la = iabs_(l), neg = l < 0;
#pragma unroll
        for (int kq = 0; kq < 8; kq++) {
          const int mag = half * 8 + kq + 1;
          const int rr = round2_(__mul24(mag, la), 6), sc = neg ? -rr : rr;
          const int dp = sv - iclamp_(dcv + sc, 0, mx), dm = sv - iclamp_(dcv - sc, 0, mx);
          e[2 * kq] += __mul24(dp, dp); e[2 * kq + 1] += __mul24(dm, dm);
        }
      }
#pragma unroll
      for (int a = 0; a < 16; a++) {
        const long long ea = (long long)wave_sum_i32(e[a]);
        if (ea < best_sse) { best_sse = ea; best_idx = half * 16 + a; }
      }
      if (LANE == 0) { SH->ca_sse[p - 1][half] = best_sse; SH->ca_idx[p - 1][half] = best_idx; }
    }
    PH(8);
    WG_SYNC();
    PH(2);
    int alpha_u = 0, alpha_v = 0;
#pragma unroll
    for (int pp = 0; pp < 2; pp++) {
      const int idx = SH->ca_sse[pp][1] < SH->ca_sse[pp][0] ? SH->ca_idx[pp][1] : SH->ca_idx[pp][0];
      const int al = idx < 0 ? 0 : ((idx & 1) ? -((idx >> 1) + 1) : ((idx >> 1) + 1));
      if (pp == 0) alpha_u = al; else alpha_v = al;
    }
    const int cfl_ok = alpha_u != 0 || alpha_v != 0;
    GroupRes gr = { 0, 0, 0, 0, 0 };
    const int cw = (W & 1) == 0;                              // waves 0 (plane U) and 2 (plane V): their S->dcp already holds the plane's DC prediction
    // The candidates' predictions go side by side into the evaluating wave's S->pred (4 x 64 samples; candidate 0 = DC stays in S->dcp), then row g of that wave
    // evaluates candidate g.  The luma mode's prediction of plane U / V is made by wave 1 / 3 (idle otherwise) straight into wave 0's / 2's S->pred while that wave
    // builds the CfL prediction: the stage's longest chain loses one block prediction for one barrier.
    if (!cw && nplain == 2) {
      const int p = (W >> 1) + 1;
      predict_block(f, x, y, log2w, availL, availU, best_mode, bdelta, ftype_uv, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF, wa, wl, S->etmp, k.wave(W - 1)->pred + nn);
    }
    if (cw) {
      const int p = (W >> 1) + 1;
      {
        const int al = p == 1 ? alpha_u : alpha_v;
        LDS uint16_t *cp = S->pred + (nc - 1) * nn;
        int lsum = 0;
        for (int idx = LANE; idx < nn; idx += 64) lsum += SH->luma_rec[idx] << 3;
        lsum = wave_sum_i32(lsum);
        const int avg = round2_(lsum, 2 * log2w);
        for (int idx = LANE; idx < nn; idx += 64) {
          const int l = ((int)SH->luma_rec[idx] << 3) - avg, v = al * l, sc = v >= 0 ? round2_(v, 6) : -round2_(-v, 6);
          cp[idx] = (uint16_t)iclamp_((int)S->dcp[idx] + sc, 0, mx);
        }
      }
      WAVE_SYNC();
    }
    WG_SYNC();                                                // the luma mode's predictions (waves 1, 3) are in place
    if (cw) {
      const int p = (W >> 1) + 1;
      const int g = GROUP_ID, cand = imin_(g, nc - 1);
      const int um = cand == nc - 1 ? UV_CFL_PRED : (cand == 0 ? DC_PRED : best_mode);
      int txtype = mode_to_txtype(um);
      if (txtype_to_sym(uvset, txtype) < 0) txtype = DCT_DCT;
      eval_group<n>(k.cc(), k.cost(), k.ls(), f, &S->grp[g], SH->srcb[p], cand == 0 ? (const LDS uint16_t *)S->dcp : (const LDS uint16_t *)(S->pred + cand * nn), p, BS, txtype, SH->sctx[p], SH->dctx[p], -1, 0, -1, SH->cact, &gr);
      const long long jp = rd_dist32(f, p, gr.sse) + rd_rate32(f, gr.rate);
      if (GROUP_LANE == 0 && g < nc) SH->cj[g][p - 1] = jp;
    }
    PH(9);
    WG_SYNC();
    PH(2);
    long long best_uv = J_INF; int bc = 0, b_sign = 0;
#pragma unroll
    for (int cnd = 0; cnd < 3; cnd++) {
      if (cnd < nc) {
        const int is_cfl = cnd == nc - 1, um = is_cfl ? UV_CFL_PRED : (cnd == 0 ? DC_PRED : best_mode);
        int jsign = 0;
        const uint32_t mode_rate = uv_mode_rate(k.cost(), uvcost, um, !is_cfl && cnd == 1 && um >= V_PRED && um <= D67_PRED && BS >= BS_8, bdelta,
                                                is_cfl && cfl_ok, alpha_u, alpha_v, &jsign);
        if (!is_cfl || cfl_ok) {
          const long long j = SH->cj[cnd][0] + SH->cj[cnd][1] + rd_rate32(f, mode_rate);
          if (j < best_uv) { best_uv = j; bc = cnd; b_sign = jsign; }
        }
      }
    }
    if (cw) {
      const int p = (W >> 1) + 1, chose_cfl = bc == nc - 1;
      const int eob = __builtin_amdgcn_readlane(gr.eob, 0), eob1 = __builtin_amdgcn_readlane(gr.eob, 16), eob2 = __builtin_amdgcn_readlane(gr.eob, 32);
      const int cul = __builtin_amdgcn_readlane(gr.cul, 0), cul1 = __builtin_amdgcn_readlane(gr.cul, 16), cul2 = __builtin_amdgcn_readlane(gr.cul, 32);
      const int dcc = __builtin_amdgcn_readlane(gr.dcc, 0), dcc1 = __builtin_amdgcn_readlane(gr.dcc, 16), dcc2 = __builtin_amdgcn_readlane(gr.dcc, 32);
      const int beob = bc == 0 ? eob : (bc == 1 ? eob1 : eob2), bcul = bc == 0 ? cul : (bc == 1 ? cul1 : cul2), bdcc = bc == 0 ? dcc : (bc == 1 ? dcc1 : dcc2);
      commit_plane<BS>(f, p, r, c, S->grp[bc].rec, S->grp[bc].qc, beob, bcul, bdcc);
      if (LANE == 0) SH->ceob[p - 1] = beob;
      if (p == 1) {
        const int buv = chose_cfl ? UV_CFL_PRED : (bc == 0 ? DC_PRED : best_mode);
        fill_map_dev(f->m_uvmode, ms, r, c, n4, buv);
        fill_map_dev((uint8_t *)f->m_angle_uv, ms, r, c, n4, (uint8_t)(int8_t)((!chose_cfl && bc == 1) ? bdelta : 0));
        fill_map_dev(f->m_cfl_sign, ms, r, c, n4, chose_cfl ? b_sign : 0);
        fill_map_dev(f->m_cfl_au, ms, r, c, n4, (chose_cfl && alpha_u) ? iabs_(alpha_u) - 1 : 0);
        fill_map_dev(f->m_cfl_av, ms, r, c, n4, (chose_cfl && alpha_v) ? iabs_(alpha_v) - 1 : 0);
      }
    }
    PH(10);
    WG_SYNC();
    PH(2);
    any_coef |= (SH->ceob[0] > 0) | (SH->ceob[1] > 0);
    total_j += best_uv;
  }
  // ---- chroma, 4x4 / 8x8 blocks with the FULL candidate set of speed <= 1 (oracle order: DC, the luma mode, the other eleven modes, CfL), in the kernels instantiated
  // for that set (Ctx::FULL: inside the speed-4 kernels this code cost 2 % of K1 by its presence alone, as a called function 9 %): the CfL alpha scan on all four
  // waves (plane x half of the alpha range), then the candidates four per wavefront (dev_group.h, one candidate per 16-lane row): wave W takes plane W / 2 + 1 and the
  // candidates of parity W % 2, eight candidates of a plane per round, two rounds for the 13 or 14 candidates -- the later candidates first, so the likely winners
  // are still in the wave's buffers at the end; the first round's best waits in LDS (dev_rect.h has the same scheme for the 2:1 blocks).  Until round 4 this set ran
  // one candidate per wave pair and round (seven rounds of predict + evaluate + two barriers); measured on config 5: tile search 5.54 -> 3.94 s with both. ----
  bool cfull = false;
  if constexpr (SMALL_GROUPED && Tools<TS>::FULL) cfull = f->np > 1 && Tools<TS>::FULL;
  if constexpr (SMALL_GROUPED && Tools<TS>::FULL) if (cfull) {
    static_assert(NW == 4 && MAXN <= 32, "the full candidate set's chroma stage deals (plane, parity) to four wavefronts");
    const uint16_t *uvcost = k.cost() + CDF_UV_CFL + best_mode * CDF_UV_CFL_STRIDE;
    unsigned long long cand_pack = 0; int nc = 0;
    auto push = [&](int m) { cand_pack |= (unsigned long long)m << (4 * nc); nc++; };
    push(DC_PRED);
    if (best_mode != DC_PRED) push(best_mode);
    if (Tools<TS>::FULL) for (int m = 1; m < 13; m++) if (m != best_mode) push(m);
    push(UV_CFL_PRED);
    const int uvset = tx_set_of(BS, Tools<TS>::reduced_tx_set(f));
    const int mx = (1 << f->bd) - 1;
    const int p = (W >> 1) + 1, half = W & 1;
    const LDS uint16_t *pra = SH->ra[p] + EDGE_OFF, *prl = SH->rl[p] + EDGE_OFF;
    int lavg;
    {
      // rdo_cfl_alpha (oracle cfl_best_alpha: alpha 0, then +1, -1, ... +16, -16; the first strictly smaller SSE wins): this wave scans one half of the range
      const LDS uint16_t *luma = SH->luma_rec;
      int lsum = 0;
      for (int idx = LANE; idx < nn; idx += 64) lsum += luma[idx] << 3;
      lsum = wave_sum_i32(lsum);
      lavg = round2_(lsum, 2 * log2w);
      predict_block(f, x, y, log2w, availL, availU, DC_PRED, 0, ftype_uv, pra, prl, wa, wl, S->etmp, S->dcp);
      long long best_sse = J_INF; int best_idx = 1 << 20;
      if (half == 0) {
        int e0 = 0;
        for (int idx = LANE; idx < nn; idx += 64) { const int d = (int)SH->srcb[p][idx] - (int)S->dcp[idx]; e0 += __mul24(d, d); }
        best_sse = (long long)wave_sum_i32(e0); best_idx = -1;
      }
      int e[16];
#pragma unroll
      for (int a2 = 0; a2 < 16; a2++) e[a2] = 0;
      for (int idx = LANE; idx < nn; idx += 64) {
        const int l = ((int)luma[idx] << 3) - lavg, dcv = S->dcp[idx], sv = SH->srcb[p][idx];
        const int la = iabs_(l), neg = l < 0;
#pragma unroll
        for (int kq = 0; kq < 8; kq++) {
          const int mag = half * 8 + kq + 1;
          const int rr = round2_(__mul24(mag, la), 6), sc = neg ? -rr : rr;
          const int dp = sv - iclamp_(dcv + sc, 0, mx), dm = sv - iclamp_(dcv - sc, 0, mx);
          e[2 * kq] += __mul24(dp, dp); e[2 * kq + 1] += __mul24(dm, dm);
        }
      }
#pragma unroll
      for (int a2 = 0; a2 < 16; a2++) {
        const long long ea = (long long)wave_sum_i32(e[a2]);
        if (ea < best_sse) { best_sse = ea; best_idx = half * 16 + a2; }
      }
      if (LANE == 0) { SH->ca_sse[p - 1][half] = best_sse; SH->ca_idx[p - 1][half] = best_idx; }
    }
    PH(8);
    WG_SYNC();
    PH(2);
    int alpha_u = 0, alpha_v = 0;
#pragma unroll
    for (int pp = 0; pp < 2; pp++) {
      const int idx = SH->ca_sse[pp][1] < SH->ca_sse[pp][0] ? SH->ca_idx[pp][1] : SH->ca_idx[pp][0];
      const int al = idx < 0 ? 0 : ((idx & 1) ? -((idx >> 1) + 1) : ((idx >> 1) + 1));
      if (pp == 0) alpha_u = al; else alpha_v = al;
    }
    const int cfl_ok = alpha_u != 0 || alpha_v != 0;
    // what waits in LDS across the rounds: the first round's best candidate of each plane (reconstruction, levels, eob / cul / dcc); lpred is dead after the luma search
    LDS uint16_t *park_rec = (LDS uint16_t *)SH->lpred + (p - 1) * 3 * nn; LDS int32_t *park_qc = (LDS int32_t *)(park_rec + nn); LDS int *park_meta = (LDS int *)SH->order + (p - 1) * 3;
    static_assert(2 * 3 * nn <= 768, "the parked candidates fit lpred");
    long long best_uv = J_INF; int b_ci = 1 << 30, b_sign = 0, b_round = 0, b_delta = 0;
    const int ns = nc <= 4 ? 1 : 2, nrounds = nc > 4 * ns ? 2 : 1, dealt = half < ns;
    GroupRes gr = { 0, 0, 0, 0, 0 };
    const int g = GROUP_ID;
    auto delta_of = [&](int um) { return (um == best_mode && um >= V_PRED && um <= D67_PRED && BS >= BS_8) ? best_delta : 0; };
#pragma unroll 1
    for (int rd = 0; rd < nrounds; rd++) {
      const int base = nrounds == 2 && rd == 0 ? 4 * ns : 0;                  // candidates base .. base + 4 ns - 1 of the list
      // this wave's (at most four) predictions of the round, side by side in S->pred (the DC candidate reads S->dcp)
#pragma unroll 1
      for (int g2 = 0; g2 < 4; g2++) {
        const int ci = base + half + ns * g2;
        if (dealt && ci < nc) {
          const int um = lut4(cand_pack, ci);
          LDS uint16_t *cp = S->pred + g2 * nn;
          if (um == UV_CFL_PRED) {
            const int al = p == 1 ? alpha_u : alpha_v;
            for (int idx = LANE; idx < nn; idx += 64) {
              const int l = ((int)SH->luma_rec[idx] << 3) - lavg, v = al * l, sc = v >= 0 ? round2_(v, 6) : -round2_(-v, 6);
              cp[idx] = (uint16_t)iclamp_((int)S->dcp[idx] + sc, 0, mx);
            }
            WAVE_SYNC();
          } else if (um != DC_PRED) predict_block(f, x, y, log2w, availL, availU, um, delta_of(um), ftype_uv, pra, prl, wa, wl, S->etmp, cp);
        }
      }
      const int ci = base + half + ns * g, live = dealt && ci < nc;
      const int um = lut4(cand_pack, live ? ci : 0);
      int txtype = mode_to_txtype(um);
      if (txtype_to_sym(uvset, txtype) < 0) txtype = DCT_DCT;
      if (dealt) eval_group<n>(k.cc(), k.cost(), k.ls(), f, &S->grp[g], SH->srcb[p], um == DC_PRED ? (const LDS uint16_t *)S->dcp : (const LDS uint16_t *)(S->pred + g * nn), p, BS, txtype,
                    SH->sctx[p], SH->dctx[p], -1, 0, -1, SH->cact, &gr);
      const long long jp = rd_dist32(f, p, gr.sse) + rd_rate32(f, gr.rate);
      if (GROUP_LANE == 0 && live) SH->cj[ci][p - 1] = jp;
      PH(9);
      WG_SYNC();
      PH(2);
      // every wave: the best of the round's candidates, in list order (strictly smaller wins: the oracle's loop)
      long long r_best = J_INF; int r_ci = 1 << 30, r_sign = 0, r_delta = 0;
      for (int cc = base; cc < imin_(nc, base + 4 * ns); cc++) {
        const int um2 = lut4(cand_pack, cc), is_cfl = um2 == UV_CFL_PRED;
        if (is_cfl && !cfl_ok) continue;
        int jsign = 0;
        const int d2 = delta_of(um2);
        const uint32_t mode_rate = uv_mode_rate(k.cost(), uvcost, um2, um2 >= V_PRED && um2 <= D67_PRED && BS >= BS_8, d2, is_cfl, alpha_u, alpha_v, &jsign);
        const long long j = SH->cj[cc][0] + SH->cj[cc][1] + rd_rate32(f, mode_rate);
        if (j < r_best) { r_best = j; r_ci = cc; r_sign = jsign; r_delta = d2; }
      }
      if (r_best < best_uv || (r_best == best_uv && r_ci < b_ci)) { best_uv = r_best; b_ci = r_ci; b_sign = r_sign; b_round = rd; b_delta = r_delta; }
      if (nrounds == 2 && rd == 0 && r_ci < nc && (r_ci - base) % ns == half) {       // this wave holds the first round's best candidate of its plane: park it
        const int gg = (r_ci - base) / ns;
        for (int i = LANE; i < nn; i += 64) { park_rec[i] = S->grp[gg].rec[i]; park_qc[i] = S->grp[gg].qc[i]; }
        const int e_ = __builtin_amdgcn_readlane(gr.eob, gg * 16), c_ = __builtin_amdgcn_readlane(gr.cul, gg * 16), d_ = __builtin_amdgcn_readlane(gr.dcc, gg * 16);
        if (LANE == 0) { park_meta[0] = e_; park_meta[1] = c_; park_meta[2] = d_; }
      }
      // (no barrier here: the next round writes other cj entries, and the parked data is read after that round's barrier)
    }
    const int b_um = lut4(cand_pack, b_ci), chose_cfl = b_um == UV_CFL_PRED;
    const int from_park = nrounds == 2 && b_round == 0;
    if (from_park ? half == 0 : b_ci % ns == half) {                        // the wave that commits plane p
      int beob, bcul, bdcc;
      if (from_park) {
        beob = park_meta[0]; bcul = park_meta[1]; bdcc = park_meta[2];
        commit_plane<BS>(f, p, r, c, (const LDS uint16_t *)park_rec, (const LDS int32_t *)park_qc, beob, bcul, bdcc);
      } else {
        const int gg = b_ci / ns;
        beob = __builtin_amdgcn_readlane(gr.eob, gg * 16); bcul = __builtin_amdgcn_readlane(gr.cul, gg * 16); bdcc = __builtin_amdgcn_readlane(gr.dcc, gg * 16);
        commit_plane<BS>(f, p, r, c, S->grp[gg].rec, S->grp[gg].qc, beob, bcul, bdcc);
      }
      if (LANE == 0) SH->ceob[p - 1] = beob;
      if (p == 1) {
        fill_map_dev(f->m_uvmode, ms, r, c, n4, b_um);
        fill_map_dev((uint8_t *)f->m_angle_uv, ms, r, c, n4, (uint8_t)(int8_t)b_delta);
        fill_map_dev(f->m_cfl_sign, ms, r, c, n4, chose_cfl ? b_sign : 0);
        fill_map_dev(f->m_cfl_au, ms, r, c, n4, (chose_cfl && alpha_u) ? iabs_(alpha_u) - 1 : 0);
        fill_map_dev(f->m_cfl_av, ms, r, c, n4, (chose_cfl && alpha_v) ? iabs_(alpha_v) - 1 : 0);
      }
    }
    PH(10);
    WG_SYNC();
    PH(2);
    any_coef |= (SH->ceob[0] > 0) | (SH->ceob[1] > 0);
    total_j += best_uv;
  }
  // ---- chroma: candidate ci2 by wave pair (ci2 & 1), plane (W & 1) + 1 within the pair ----
  if constexpr (NW >= 2) if (f->np > 1 && !cgrouped && !cfull) {
    const int cfl_allowed = BS <= BS_32;
    const uint16_t *uvcost = cfl_allowed ? k.cost() + CDF_UV_CFL + best_mode * CDF_UV_CFL_STRIDE : k.cost() + CDF_UV_NOCFL + best_mode * CDF_UV_NOCFL_STRIDE;
    // candidate list: up to 14 modes of 4 bits packed into one 64-bit value (a private array indexed at run time lives in scratch)
    unsigned long long cand_pack = 0; int nc = 0;
    auto push = [&](int m) { cand_pack |= (unsigned long long)m << (4 * nc); nc++; };
    push(DC_PRED);
    if (best_mode != DC_PRED) push(best_mode);
    if (Tools<TS>::FULL) for (int m = 1; m < 13; m++) if (m != best_mode) push(m);
    if (cfl_allowed) push(UV_CFL_PRED);
    const int uvset = tx_set_of(BS, Tools<TS>::reduced_tx_set(f));
    constexpr int NPAIR = 2;
    const int pair = ((W >> 1) & 1) ^ 1, active = W < 4;      // pair 0 (two plain candidates) = waves 2, 3: they have the lighter luma share
    long long pb_j = J_INF; int pb_c = 1 << 30, pb_delta = 0, pb_sign = 0, pb_au = 0, pb_av = 0, ccur = 0; TxRes pb_tr = { 0, 0, 0, 0, 0 };
    // pair 0: candidates 0 and the odd ones; pair 1: the even ones from 2 on and CfL (always last) -- the winner rule
    // below only looks at (cost, candidate index), so the dealing order does not change the decision.
    int mine[16], nmine = 0, nother = 0;
    for (int i = 0; i < nc; i++) {
      const int pr = (lut4(cand_pack, i) == UV_CFL_PRED) ? 1 : (i == 0 || (i & 1)) ? 0 : 1;
      if (pr == pair) mine[nmine++] = i; else nother++;
    }
    const int rounds = imax_(nmine, nother);
    for (int rd = 0; rd < rounds; rd++) {
      const int valid = active && rd < nmine;
      int ci2 = 0;
      for (int q = 0; q < 16; q++) if (q == rd && valid) ci2 = mine[q];
      const int um = valid ? lut4(cand_pack, ci2) : DC_PRED;
      const int delta = (um == best_mode && um >= V_PRED && um <= D67_PRED && BS >= BS_8) ? best_delta : 0;
      int txtype = mode_to_txtype(um);
      if (txtype_to_sym(uvset, txtype) < 0) txtype = DCT_DCT;
      {
        const int p = (W & 1) + 1;
        // rdo_cfl_alpha: the alpha in -16..16 minimising this plane's prediction SSE.  The 1 + 32 trial values are
        // scanned in two halves: by the wave that owns the CfL candidate (pair 1) and by its idle peer of pair 0;
        // each reports (sse, position in the scan order), the earliest position wins ties exactly like one scan.
        const int cfl_main = valid && um == UV_CFL_PRED;
        const int cfl_help = cfl_allowed && active && pair == 0 && rd == nother - 1;
        if (cfl_main || cfl_help) {
          const int half = cfl_help ? 1 : 0;
          const LDS uint16_t *luma = SH->luma_rec;
          int lsum = 0;
          for (int idx = LANE; idx < nn; idx += 64) lsum += luma[idx] << 3;
          lsum = wave_sum_i32(lsum);
          const int avg = round2_(lsum, 2 * log2w), mx = (1 << f->bd) - 1;
          predict_block(f, x, y, log2w, availL, availU, DC_PRED, 0, ftype_uv, SH->ra[p] + EDGE_OFF, SH->rl[p] + EDGE_OFF, wa, wl, S->etmp, S->dcp);
          long long best_sse = J_INF; int best_idx = 1 << 20;
          if (half == 0) {
            int e0 = 0;
            for (int idx = LANE; idx < nn; idx += 64) { const int d = (int)SH->srcb[p][idx] - (int)S->dcp[idx]; e0 += __mul24(d, d); }
            best_sse = wave_sum_i64((long long)e0); best_idx = -1;
          }
          // scan position aa = 2k is alpha +(k+1), aa = 2k+1 is -(k+1); the scaled luma term of -a is minus that of +a,
          // so each magnitude is scaled and rounded once.  A CfL block has <= 1024 samples: its SSE stays below 2^31.
          int e[16];
#pragma unroll
          for (int a = 0; a < 16; a++) e[a] = 0;
          for (int idx = LANE; idx < nn; idx += 64) {
            const int l = ((int)luma[idx] << 3) - avg, dcv = S->dcp[idx], sv = SH->srcb[p][idx];
            const int la = iabs_(l), neg = l < 0;
#pragma unroll
            for (int kq = 0; kq < 8; kq++) {
              const int mag = half * 8 + kq + 1;
              const int r = round2_(__mul24(mag, la), 6), sc = neg ? -r : r;       // alpha = +mag
              const int dp = sv - iclamp_(dcv + sc, 0, mx), dm = sv - iclamp_(dcv - sc, 0, mx);
              e[2 * kq] += __mul24(dp, dp); e[2 * kq + 1] += __mul24(dm, dm);
            }
          }
#pragma unroll
          for (int a = 0; a < 16; a++) {
            const long long ea = (long long)wave_sum_i32(e[a]);
            if (ea < best_sse) { best_sse = ea; best_idx = half * 16 + a; }
          }
          if (LANE == 0) { SH->ca_sse[p - 1][half] = best_sse; SH->ca_idx[p - 1][half] = best_idx; }
        }
      }
      PH(8);
      WG_SYNC();                                            // (A) alphas of both planes visible
      PH(2);
      int alpha_u = 0, alpha_v = 0, jsign = 0, ok = valid;
      if (valid && um == UV_CFL_PRED) {
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
          const int idx = SH->ca_sse[pp][1] < SH->ca_sse[pp][0] ? SH->ca_idx[pp][1] : SH->ca_idx[pp][0];
          const int al = idx < 0 ? 0 : ((idx & 1) ? -((idx >> 1) + 1) : ((idx >> 1) + 1));
          if (pp == 0) alpha_u = al; else alpha_v = al;
        }
        if (alpha_u == 0 && alpha_v == 0) ok = 0;
      }
      const uint32_t mode_rate = uv_mode_rate(k.cost(), uvcost, um, um >= V_PRED && um <= D67_PRED && BS >= BS_8, delta,
                                              valid && um == UV_CFL_PRED && ok, alpha_u, alpha_v, &jsign);
      TxRes trp = { 0, 0, 0, 0, 0 };
      if (ok) {
        {
          const int p = (W & 1) + 1;
          const LDS uint16_t *pra = SH->ra[p] + EDGE_OFF, *prl = SH->rl[p] + EDGE_OFF;
          if (um == UV_CFL_PRED) {
            const int al = p == 1 ? alpha_u : alpha_v;
            predict_block(f, x, y, log2w, availL, availU, DC_PRED, 0, ftype_uv, pra, prl, wa, wl, S->etmp, S->dcp);
            if (al) {
              // predict_cfl against the LDS copy of the luma reconstruction
              int lsum = 0;
              for (int idx = LANE; idx < nn; idx += 64) lsum += SH->luma_rec[idx] << 3;
              lsum = wave_sum_i32(lsum);
              const int avg = round2_(lsum, 2 * log2w), mx = (1 << f->bd) - 1;
              for (int idx = LANE; idx < nn; idx += 64) {
                const int l = ((int)SH->luma_rec[idx] << 3) - avg, v = al * l, sc = v >= 0 ? round2_(v, 6) : -round2_(-v, 6);
                S->pred[idx] = (uint16_t)iclamp_((int)S->dcp[idx] + sc, 0, mx);
              }
            } else { for (int i = LANE; i < nn; i += 64) S->pred[i] = S->dcp[i]; }
            WAVE_SYNC();
          } else {
            predict_block(f, x, y, log2w, availL, availU, um, delta, ftype_uv, pra, prl, wa, wl, S->etmp, S->pred);
          }
          const long long jp = eval_tx<MAXN, BS, NW>(k, p, SH->sctx[p], SH->dctx[p], S->pred, txtype, -1, 0, S->rec[ccur], S->qc[ccur], &trp);
          if (LANE == 0) SH->cj[ci2][p - 1] = jp;
        }
      }
      PH(9);
      WG_SYNC();                                            // (B) both planes' costs visible
      PH(2);
      if (valid && ok) {
        const long long j = SH->cj[ci2][0] + SH->cj[ci2][1] + rd_rate32(f, mode_rate);
        if (j < pb_j) {
          pb_j = j; pb_c = ci2; pb_delta = delta; pb_sign = jsign; pb_au = alpha_u; pb_av = alpha_v; pb_tr = trp;
          ccur ^= 1;
        }
      }
    }
    if (LANE == 0 && (W & 1) == 0 && W < 4) { SH->pbest_j[pair] = pb_j; SH->pbest_c[pair] = pb_c; }
    WG_SYNC();
    PH(2);
    int wp = 0;
    if ((SH->pbest_j[1] < SH->pbest_j[0] || (SH->pbest_j[1] == SH->pbest_j[0] && SH->pbest_c[1] < SH->pbest_c[0]))) wp = 1;
    const long long best_uv = SH->pbest_j[wp];
    if (active && pair == wp) {
      const int p = (W & 1) + 1, b = ccur ^ 1;
      commit_plane<BS>(f, p, r, c, (const LDS uint16_t *)S->rec[b], (const LDS int32_t *)S->qc[b], pb_tr.eob, pb_tr.cul, pb_tr.dcc);
      if (LANE == 0) SH->ceob[p - 1] = pb_tr.eob;
      if (p == 1) {
        const int buv = lut4(cand_pack, pb_c);
        fill_map_dev(f->m_uvmode, ms, r, c, n4, buv);
        fill_map_dev((uint8_t *)f->m_angle_uv, ms, r, c, n4, (uint8_t)(int8_t)pb_delta);
        fill_map_dev(f->m_cfl_sign, ms, r, c, n4, pb_sign);
        fill_map_dev(f->m_cfl_au, ms, r, c, n4, pb_au ? iabs_(pb_au) - 1 : 0);
        fill_map_dev(f->m_cfl_av, ms, r, c, n4, pb_av ? iabs_(pb_av) - 1 : 0);
      }
    }
    PH(10);
    WG_SYNC();
    PH(2);
    any_coef |= (SH->ceob[0] > 0) | (SH->ceob[1] > 0);
    total_j += best_uv;
    (void)qn;
  }
  if (DBG_IS(f, 6)) return 0;
  // ---- skip flag ----
  const int skip = !any_coef;
  // intra_segment_id follows the skip flag (no pre-skip feature): a skipped block takes the predicted id, any other codes its own
  int seg_ctx = 0;
  const int seg_nb2 = SH->seg_nb, seg_ul = (seg_nb2 & 15) - 1, seg_u = ((seg_nb2 >> 4) & 15) - 1, seg_l = (seg_nb2 >> 8) - 1;
  const int seg_p = seg_pred(seg_ul, seg_u, seg_l, &seg_ctx), seg_own = f->seg_n ? SH->seg : 0, seg_fin = f->seg_n ? (skip ? seg_p : seg_own) : 0;
  if (W == 0) {
    fill_map_dev(f->m_skip, ms, r, c, n4, skip | (seg_fin << 1));
    if (skip) for (int p = 0; p < f->np; p++) { fill_map_dev(f->m_lvl[p], ms, r, c, n4, 0); fill_map_dev(f->m_dc[p], ms, r, c, n4, 0); }
  }
  const int sctx = nb_skip;
  total_j += ((long long)k.cost()[CDF_SKIP + sctx * CDF_SKIP_STRIDE + skip] * f->rdmult + 256) >> 9;
  if (f->seg_n && !skip) total_j += ((long long)k.cost()[CDF_SEG_ID + seg_ctx * CDF_SEG_ID_STRIDE + seg_symbol(seg_own, seg_p, f->seg_n)] * f->rdmult + 256) >> 9;
  PH(11);
  WG_SYNC();
  PH(2);
  return total_j;
}

#include "dev_blk64.h"
// the evaluation of one block: the generic search, or -- for the 64x64 level of the 32x32 class -- its cooperative form
template <int MAXN, int BS, int NW, int TS> __device__ __forceinline__ long long blk_eval(const Ctx<MAXN, NW, TS> k, int r, int c, long long budget = J_INF) {
  if constexpr (MAXN == 32 && BS == 4) return PH_SIZE(3, try_block64<NW>(k, r, c, budget)); else return PH_SIZE(BS < 3 ? BS : 3, (try_block<MAXN, BS, NW>(k, r, c, budget)));
}

// ---- area snapshot (NONE-vs-SPLIT comparison), kept in the workgroup's HBM scratch; whole workgroup ----
// Wave p copies plane p (reconstruction + levels, four samples per lane: 8- and 16-byte accesses), the last wave the 17 mode-info byte maps (one
// row of the area per lane, the map's pointer picked out of the LDS frame descriptor) and the eob maps.  m_decoded is not part of a snapshot.
template <int BYTES> __device__ __forceinline__ void copy_row_(void *dst, const void *src) {
  if constexpr (BYTES == 1) *(uint8_t *)dst = *(const uint8_t *)src;
  else if constexpr (BYTES == 2) *(uint16_t *)dst = *(const uint16_t *)src;
  else if constexpr (BYTES == 4) *(uint32_t *)dst = *(const uint32_t *)src;
  else if constexpr (BYTES == 8) *(uint2 *)dst = *(const uint2 *)src;
  else { for (int i = 0; i < BYTES / 16; i++) ((uint4 *)dst)[i] = ((const uint4 *)src)[i]; }
}
static_assert(offsetof(FrameDev, m_eob) - offsetof(FrameDev, m_bsize) == 18 * sizeof(void *), "area_copy_dev indexes m_bsize .. m_dc[2] as one array of 18 map pointers");
template <int BS, int NW> __device__ inline void area_copy_dev(const LDS FrameDev *f, uint8_t *snap, int r, int c, int save) {
  constexpr int n = 4 << BS, n4 = 1 << BS, Q = n / 4;
#if MI_PROFILE
  const unsigned long long ac_t0_ = clock64();
#define mi_prof_slot(w, i) mi_prof_slot_n<MI_PROF_MAXN>(w, i)
#endif
  uint8_t *smaps = snap + 3 * n * n * 6;                     // 17 byte-maps [n4*n4]
  uint8_t *seob = smaps + 18 * n4 * n4;                      // [3][n4*n4] uint16
  const int W = NW > 1 ? WAVE_ID : 0, np = f->np, rs = f->stride, ms = f->mi_stride;
  for (int p = W; p < np; p += NW) {
    uint16_t *gr = f->rec[p] + (size_t)(r * 4) * rs + c * 4;
    int32_t *gc = f->coef[p] + (size_t)(r * 4) * rs + c * 4;
    uint2 *srec = (uint2 *)(snap + (size_t)p * n * n * 2);
    uint4 *scoef = (uint4 *)(snap + 3 * n * n * 2 + (size_t)p * n * n * 4);
    for (int u = LANE; u < n * Q; u += 64) {
      const int o = (u / Q) * rs + 4 * (u % Q);
      if (save) { srec[u] = *(const uint2 *)(gr + o); scoef[u] = *(const uint4 *)(gc + o); }
      else { *(uint2 *)(gr + o) = srec[u]; *(uint4 *)(gc + o) = scoef[u]; }
    }
  }
  if (W == NW - 1) {
    const LDS unsigned long long *mp = (const LDS unsigned long long *)&f->m_bsize;     // 18 pointers: ... m_cfl_av, m_decoded (skipped), m_txsize, angles, m_lvl[3], m_dc[3]
    for (int u = LANE; u < 17 * n4; u += 64) {
      const int m = u / n4, row = u - m * n4, mi = m + (m >= 8);
      if (np == 1 && (mi == 13 || mi == 14 || mi == 16 || mi == 17)) continue;             // 4:0:0: no chroma context maps
      uint8_t *g = (uint8_t *)mp[mi] + (r + row) * ms + c, *sp = smaps + m * n4 * n4 + row * n4;
      if (save) copy_row_<n4>(sp, g); else copy_row_<n4>(g, sp);
    }
    const LDS unsigned long long *ep = (const LDS unsigned long long *)&f->m_eob[0];
    for (int u = LANE; u < np * n4; u += 64) {
      const int p = u / n4, row = u - p * n4;
      uint8_t *g = (uint8_t *)((uint16_t *)ep[p] + (r + row) * ms + c), *sp = seob + (p * n4 * n4 + row * n4) * 2;
      if (save) copy_row_<2 * n4>(sp, g); else copy_row_<2 * n4>(g, sp);
    }
  }
  WG_SYNC();
#if MI_PROFILE
  if (LANE == 0) mi_prof_slot(WAVE_ID, 14) += clock64() - ac_t0_;    // (profiling builds) the walker's area copies
#endif
}

// ---- partition symbol rates for the walkers ----
// The walker used to price every partition symbol with two dependent round trips to L2 (the neighbours' block sizes, then the rate table) executed by all four waves
// with nothing else to do: ~20 of them per 16x16 root, a tenth of the kernel.  Now the slice of the rate table the walker can ask for sits in LDS (8x8 nodes: NONE, HORZ,
// VERT, SPLIT per context; larger nodes: NONE and SPLIT), and a node fetches the block sizes of its outer neighbours -- above / left of its two halves -- in one batch
// when it is entered; inside a split trial the siblings are undivided blocks of the child size, which never count as narrower / lower than the child.
__device__ __forceinline__ int part_cost_idx(int bs, int ctx, int part) { return bs == 1 ? ctx * 4 + part : 16 + (bs - 2) * 8 + ctx * 2 + (part == 3); }
template <typename SHT> __device__ inline void load_part_cost(LDS SHT *SH, const uint16_t *cost, int tid) {
  if (tid < 40) {
    const int bs = tid < 16 ? 1 : 2 + (tid - 16) / 8, e = tid < 16 ? tid : (tid - 16) % 8, ctx = tid < 16 ? e >> 2 : e >> 1, part = tid < 16 ? (e & 3) : ((e & 1) ? 3 : 0);
    SH->part_cost[tid] = cost[CDF_PARTITION + ((bs - 1) * 4 + ctx) * CDF_PARTITION_STRIDE + part];
  }
}
// partition context of a node of size code `bs` from the block-size codes above / left of it (spec 8.3.2: is the above block narrower, the left block lower than the node)
__device__ __forceinline__ int part_ctx_of(int bs, int availU, int availL, int code_above, int code_left) {
  return ((availL && dim_hl(code_left) < 2 + bs) ? 2 : 0) + ((availU && dim_wl(code_above) < 2 + bs) ? 1 : 0);
}
struct PartNb { int a0, a1, l0, l1, availU, availL; };          // block-size codes above the node's left / right half and left of its upper / lower half
template <int BS> __device__ __forceinline__ PartNb part_neighbours(const LDS FrameDev *f, const LDS TileB *t, int r, int c, bool halves) {
  constexpr int half = (1 << BS) >> 1;
  const int ms = f->mi_stride, mi = r * ms + c;
  PartNb n; n.availU = r > t->mi_row_start; n.availL = c > t->mi_col_start;
  const int v0 = f->m_bsize[n.availU ? mi - ms : mi], v1 = f->m_bsize[(n.availU && halves) ? mi - ms + half : mi];
  const int v2 = f->m_bsize[n.availL ? mi - 1 : mi], v3 = f->m_bsize[(n.availL && halves) ? mi + half * ms - 1 : mi];
  n.a0 = uni32(v0); n.a1 = uni32(v1); n.l0 = uni32(v2); n.l1 = uni32(v3);
  return n;
}
template <typename SHT> __device__ __forceinline__ long long part_j(const LDS SHT *SH, const LDS FrameDev *f, int bs, int ctx, int part) {
  return ((long long)SH->part_cost[part_cost_idx(bs, ctx, part)] * f->rdmult + 256) >> 9;
}

// `known_j` >= 0: the parent's split trial has just evaluated this block undivided, every earlier sibling kept
// PARTITION_NONE, and the frame buffers still hold that result -- try_block() would reproduce it bit for bit, so
// its cost is taken from the trial (oracle/av1o_search.c rd_partition does the same).  Returns 1 when split.
// ---- fine-grained launches: dependencies per root block ----
// A launch with fewer runnable superblocks than resident workgroups (single images: the wavefront over a tile's superblocks is short) synchronises
// at the granularity of the search's root blocks (the largest block size of the class: 16x16 or 32x32) instead of whole superblocks.  A workgroup still
// claims a superblock and walks its roots in coding order; before a root it waits for exactly the roots that root reads -- left, above, above-left,
// and above-right / below-left where those precede it in coding order (the only case in which the decoder, and so the search, treats them as
// available) -- and after it publishes the root's bit in the superblock's word of f->sb_prog's second array.  Every root a workgroup waits for
// belongs to its own superblock or to one earlier in the work list, so the no-deadlock argument of the list order holds unchanged; the decoded
// flags stay exact because a later neighbour of a block always waits for it.  Steady state for 16x16 roots: a superblock starts 0.75 of a
// superblock time after its left neighbour and 1.125 after the one above, against 1 and 2 (profiles/r03m_*).
// Since round 5 every launch of the 16x16 class runs this way (mi_avif.hip search_enqueue: list order 3 * row + 2 * column), and only the roots that other superblocks
// read take part: a root waits only if it lies in its superblock's first row or column (the others read nothing outside it), publishes only if it lies in the last row or
// column (nothing else is ever polled), and a waiting root polls foreign superblocks' bits only -- 7 waits and 7 publishes per 16 roots (MI_K1_ROOT_SKIP=0: all 16).
#ifndef MI_K1_ROOT_SKIP
#define MI_K1_ROOT_SKIP 1
#endif
// The acquire after a dependency wait: ONE wavefront invalidates (an agent-scope acquire is `buffer_inv sc1`: the compute unit's vector cache and this XCD's L2 lose their
// lines, for everybody), the workgroup barrier hands the ordering on to the other three -- four invalidations per wait cost ~1 % of the launch (profiles/r05zn_ab_k1_acquire.txt;
// the same file: with every L2 writeback / invalidation of the launch removed, which breaks the results across XCDs, the launch is no faster than this).
// This leans on the target, not on the HIP / LLVM memory model: on gfx942 / gfx950 in the default (non-tgsplit) mode the four waves of a workgroup share one vector L1,
// so wave 0's invalidation serves all of them.  Any other target fences on every wave.
#if defined(__gfx942__) || defined(__gfx950__)
#define MI_K1_ACQUIRE() do { if (WAVE_ID == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); WG_SYNC(); } while (0)
#else
#define MI_K1_ACQUIRE() do { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); WG_SYNC(); } while (0)
#endif
// Bound of the dependency polls (log2): 2^25 polls x s_sleep 16 are tens of seconds -- long enough for any legitimate wait of a product launch.  Experiment builds
// (tools/build_variant.sh) use 2^20, so that a variant with a broken protocol costs seconds of the GPU lease, not the lease.
#ifndef MI_K1_POLL_LOG2
#define MI_K1_POLL_LOG2 25
#endif
#define MI_K1_POLL_MAX (1u << MI_K1_POLL_LOG2)
__device__ __forceinline__ int root_z(int bi, int bj) { return ((bi & 1) << 1) | (bj & 1) | ((bi & 2) << 2) | ((bj & 2) << 1); }   // Morton index in the superblock
template <int MAXBS, int MAXN, int NW, int TS> __device__ inline void root_wait(const Ctx<MAXN, NW, TS> k, int r, int c) {
  constexpr int G = 1 << (4 - MAXBS);
  const LDS FrameDev *f = k.f(); const LDS TileB *t = k.t();
  // a root below the superblock's first row and right of its first column reads its own superblock only (left, above, above-left, above-right are inside it, the
  // below-left one is inside it or later in coding order): nothing to wait for, nothing to invalidate -- 9 of the 16 roots of a 64x64 superblock
  if (MI_K1_ROOT_SKIP && ((r >> MAXBS) & (G - 1)) != 0 && ((c >> MAXBS) & (G - 1)) != 0) return;
  if (threadIdx.x == 0) {
    const int *mask = f->sb_prog + f->sb_rows * f->tile_cols;
    const int gr = r >> MAXBS, gc = c >> MAXBS, zc = root_z(gr & (G - 1), gc & (G - 1)), sr = r >> 4, sc = c >> 4;
    auto need = [&](int dr, int dc, bool only_if_earlier) {
      const int rr = (gr + dr) << MAXBS, cc = (gc + dc) << MAXBS;
      if (rr < t->mi_row_start || rr >= t->mi_row_end || cc < t->mi_col_start || cc >= t->mi_col_end) return;
      const int sr2 = rr >> 4, sc2 = cc >> 4, z2 = root_z((gr + dr) & (G - 1), (gc + dc) & (G - 1));
      if (MI_K1_ROOT_SKIP && sr2 == sr && sc2 == sc) return;               // a root of this superblock: this workgroup did it (program order), and it may never be published
      if (only_if_earlier && !(sr2 < sr || (sr2 == sr && (sc2 < sc || (sc2 == sc && z2 < zc))))) return;
      const int *w = mask + sr2 * f->sb_cols + sc2;
      // Bounded (~2^25 polls are tens of seconds: a protocol error, a preempted or shared device must not hang the GPU), and a wait that gives up marks the frame:
      // the entropy stage then reports every tile of it as failed (tile_len = 0xFFFFFFFF) and the host returns MI_ENCODING_ERROR instead of a stream whose
      // reconstruction the search did not see.
      // (once any wait of the frame has given up, the others leave at their next check instead of each running to its own bound: a broken launch ends in about one bound)
      unsigned spin = 0;
      while (!((__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> z2) & 1) && spin < MI_K1_POLL_MAX) {
        __builtin_amdgcn_s_sleep(16); spin++;
        if ((spin & 4095u) == 0u && __hip_atomic_load(search_error_word(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) spin = MI_K1_POLL_MAX;
      }
      if (spin >= MI_K1_POLL_MAX) __hip_atomic_store(search_error_word(f), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    need(0, -1, false); need(-1, 0, false); need(-1, -1, false); need(-1, 1, true); need(1, -1, true);
  }
  WG_SYNC();
  MI_K1_ACQUIRE();
}
template <int MAXBS, int MAXN, int NW, int TS> __device__ inline void root_publish(const Ctx<MAXN, NW, TS> k, int r, int c) {
  constexpr int G = 1 << (4 - MAXBS);
  const LDS FrameDev *f = k.f();
  // only the last row and the last column of a superblock's roots are read from outside it (by the superblocks right, below, below-left of it): the other 9 of 16
  // are never polled, and their data is covered by the release of the next root that is
  if (MI_K1_ROOT_SKIP && ((r >> MAXBS) & (G - 1)) != G - 1 && ((c >> MAXBS) & (G - 1)) != G - 1) return;
  WG_SYNC();                                                               // every wave's stores of this root are issued
  if (threadIdx.x == 0) {
    int *w = f->sb_prog + f->sb_rows * f->tile_cols + (r >> 4) * f->sb_cols + (c >> 4);
    __hip_atomic_fetch_or(w, 1 << root_z((r >> MAXBS) & (G - 1), (c >> MAXBS) & (G - 1)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MAXN, int MAXBS, int BS, int NW> struct RdPart {
  template <int TS> static __device__ MI_K1_WALK_INLINE int run(const Ctx<MAXN, NW, TS> k, int r, int c, long long known_j) {
    const LDS FrameDev *f = k.f();
    if (r >= f->mi_rows || c >= f->mi_cols) return 0;
    constexpr int half = (1 << BS) >> 1, px = 4 << BS, n4 = 1 << BS;
    const int has_rows = (r + half) < f->mi_rows, has_cols = (c + half) < f->mi_cols;
    const int must_split = px > f->part_max || !has_rows || !has_cols;
    const int can_split = px > f->part_min || must_split;
    if (!can_split || (DBG_IS(f, 9) && BS == 1)) {
      if constexpr (BS <= MAXBS) { if (known_j < 0) blk_eval<MAXN, BS, NW>(k, r, c); }
      return 0;
    }
    int do_split = must_split;
    long long sub_j[4] = { -1, -1, -1, -1 };
    if constexpr (BS <= MAXBS) {
      if (!must_split) {
        const PartNb nb = part_neighbours<BS>(f, k.t(), r, c, BS - 1 >= BS_8);
        const int pctx = part_ctx_of(BS, nb.availU, nb.availL, nb.a0, nb.l0);
        const long long j_blk = known_j >= 0 ? known_j : uni64(blk_eval<MAXN, BS, NW>(k, r, c));
        const long long j_none = uni64(j_blk + part_j(k.sh(), f, BS, pctx, 0));
        area_copy_dev<BS, NW>(f, k.snap(), r, c, 1);
        long long j_split = uni64(part_j(k.sh(), f, BS, pctx, 3));
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (!(j_split < j_none) || DBG_IS(f, 7)) break;
          const int rr = r + (q >> 1) * half, cc = c + (q & 1) * half;
          if (rr >= f->mi_rows || cc >= f->mi_cols) continue;
          sub_j[q] = uni64(blk_eval<MAXN, BS - 1, NW>(k, rr, cc, j_none - j_split));
          j_split += sub_j[q];
          if (BS - 1 >= BS_8) {
            // the child's own partition symbol (NONE): above / left of it lie the node's outer neighbours or a sibling of its own size (never narrower / lower)
            const int cctx = part_ctx_of(BS - 1, (q >> 1) ? 1 : nb.availU, (q & 1) ? 1 : nb.availL, (q >> 1) ? BS - 1 : ((q & 1) ? nb.a1 : nb.a0), (q & 1) ? BS - 1 : ((q >> 1) ? nb.l1 : nb.l0));
            j_split += uni64(part_j(k.sh(), f, BS - 1, cctx, 0));
          }
        }
        if constexpr (BS == 1) {
          // PARTITION_HORZ / PARTITION_VERT (two 8x4 / 4x8 blocks) against the best of NONE / SPLIT so far (oracle rd_partition)
          uint8_t *best_snap = k.snap() + MI_SNAP_BYTES_SQ(4 << MAXBS), *split_snap = best_snap + MI_SNAP_BYTES(8);
          long long j_best = j_none; int have_split = 0, rect_won = 0;
          if (j_split < j_none) { j_best = j_split; area_copy_dev<BS, NW>(f, split_snap, r, c, 1); have_split = 1; }
          {
            long long j = uni64(part_j(k.sh(), f, BS, pctx, 1));
            for (int k2 = 0; k2 < 2 && j < j_best; k2++) j += uni64(PH_SIZE(4, (try_block_rect<MAXN, BS_8X4, NW>(k, r + k2, c, j_best - j))));
            if (j < j_best) { j_best = j; rect_won = 1; area_copy_dev<BS, NW>(f, best_snap, r, c, 1); }
          }
          {
            long long j = uni64(part_j(k.sh(), f, BS, pctx, 2));
            for (int k2 = 0; k2 < 2 && j < j_best; k2++) j += uni64(PH_SIZE(5, (try_block_rect<MAXN, BS_4X8, NW>(k, r, c + k2, j_best - j))));
            if (j < j_best) { j_best = j; rect_won = 2; area_copy_dev<BS, NW>(f, best_snap, r, c, 1); }
          }
          if (rect_won) { area_copy_dev<BS, NW>(f, best_snap, r, c, 0); return 1; }   // 1: the later siblings' trial results are stale
          if (have_split) area_copy_dev<BS, NW>(f, split_snap, r, c, 0);
        }
        if (j_split < j_none && !DBG_IS(f, 7) && !DBG_IS(f, 8) && !(DBG_IS(f, 10) && BS == 1)) do_split = 1;
        else { area_copy_dev<BS, NW>(f, k.snap(), r, c, 0); }
      }
    }
    if (do_split) {
      int chain = !must_split && !DBG_IS(f, 11);      // the four trial results are in place until a sibling decides to split
      MI_K1_WALK_SPLIT_LOOP
      for (int q = 0; q < 4; q++)
      {
        const int rr = r + (q >> 1) * half, cc = c + (q & 1) * half;
        const bool sync_root = BS - 1 == MAXBS && MAXBS < 4 && k.sh()->fine && rr < f->mi_rows && cc < f->mi_cols;
        if constexpr (BS - 1 == MAXBS && MAXBS < 4) if (sync_root) root_wait<MAXBS>(k, rr, cc);
        if (RdPart<MAXN, MAXBS, BS - 1, NW>::run(k, rr, cc, chain ? sub_j[q] : -1)) chain = 0;
        if constexpr (BS - 1 == MAXBS && MAXBS < 4) if (sync_root) root_publish<MAXBS>(k, rr, cc);
      }
      return 1;
    }
    return 0;
  }
};
template <int MAXN, int MAXBS, int NW> struct RdPart<MAXN, MAXBS, 0, NW> {
  template <int TS> static __device__ MI_K1_WALK_INLINE int run(const Ctx<MAXN, NW, TS> k, int r, int c, long long known_j) {
    const LDS FrameDev *f = k.f();
    if (r >= f->mi_rows || c >= f->mi_cols) return 0;
    if (known_j >= 0) return 0;
    blk_eval<MAXN, 0, NW>(k, r, c);
    return 0;
  }
};

// encode_partition_bottomup (SpeedTweaks.encode_bottomup, speed <= 2, ravif av1encoder.rs:575; oracle rd_partition_bottomup): every
// child is searched recursively first, the undivided block competes with the children's own best partitions.  One snapshot
// per level (the levels are live at the same time); returns the node's RD cost including its partition symbol.
template <int MAXN> __device__ __forceinline__ constexpr size_t snap_level_off(int bs, int maxbs) {
  size_t o = 0;
  for (int b = maxbs; b > bs; b--) o += MI_SNAP_BYTES(4 << b);
  return o;
}
template <int MAXN, int MAXBS, int BS, int NW> struct RdPartBU {
  template <int TS> static __device__ MI_K1_WALK_INLINE long long run(const Ctx<MAXN, NW, TS> k, int r, int c) {
    const LDS FrameDev *f = k.f();
    if (r >= f->mi_rows || c >= f->mi_cols) return 0;
    constexpr int half = (1 << BS) >> 1, px = 4 << BS, n4 = 1 << BS;
    const int has_rows = (r + half) < f->mi_rows, has_cols = (c + half) < f->mi_cols;
    const int must_split = px > f->part_max || !has_rows || !has_cols;
    const int can_split = px > f->part_min || must_split;
    long long j_none = J_INF;
    int pctx = 0;
    if (!must_split) { const PartNb nb = part_neighbours<BS>(f, k.t(), r, c, false); pctx = part_ctx_of(BS, nb.availU, nb.availL, nb.a0, nb.l0); }
    if constexpr (BS <= MAXBS) {
      if (!must_split) {
        j_none = uni64(blk_eval<MAXN, BS, NW>(k, r, c));
        j_none += uni64(part_j(k.sh(), f, BS, pctx, 0));
        if (!can_split) return j_none;
        area_copy_dev<BS, NW>(f, k.snap() + snap_level_off<MAXN>(BS, MAXBS), r, c, 1);
      }
    }
    long long j_split = must_split ? 0 : uni64(part_j(k.sh(), f, BS, pctx, 3));
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
      if (!must_split && j_split >= j_none) break;
      const int rr = r + (q >> 1) * half, cc = c + (q & 1) * half;
      const bool sync_root = BS - 1 == MAXBS && MAXBS < 4 && k.sh()->fine && rr < f->mi_rows && cc < f->mi_cols;
      if constexpr (BS - 1 == MAXBS && MAXBS < 4) if (sync_root) root_wait<MAXBS>(k, rr, cc);
      j_split += RdPartBU<MAXN, MAXBS, BS - 1, NW>::run(k, rr, cc);
      if constexpr (BS - 1 == MAXBS && MAXBS < 4) if (sync_root) root_publish<MAXBS>(k, rr, cc);
    }
    if constexpr (BS == 1) if (!must_split) {
      uint8_t *best_snap = k.snap() + MI_SNAP_BYTES_SQ(4 << MAXBS), *split_snap = best_snap + MI_SNAP_BYTES(8);
      long long j_best = j_none; int have_split = 0, rect_won = 0;
      if (j_split < j_none) { j_best = j_split; area_copy_dev<BS, NW>(f, split_snap, r, c, 1); have_split = 1; }
      {
        long long j = uni64(part_j(k.sh(), f, BS, pctx, 1));
        for (int k2 = 0; k2 < 2 && j < j_best; k2++) j += uni64(PH_SIZE(4, (try_block_rect<MAXN, BS_8X4, NW>(k, r + k2, c, j_best - j))));
        if (j < j_best) { j_best = j; rect_won = 1; area_copy_dev<BS, NW>(f, best_snap, r, c, 1); }
      }
      {
        long long j = uni64(part_j(k.sh(), f, BS, pctx, 2));
        for (int k2 = 0; k2 < 2 && j < j_best; k2++) j += uni64(PH_SIZE(5, (try_block_rect<MAXN, BS_4X8, NW>(k, r, c + k2, j_best - j))));
        if (j < j_best) { j_best = j; rect_won = 2; area_copy_dev<BS, NW>(f, best_snap, r, c, 1); }
      }
      if (rect_won) { area_copy_dev<BS, NW>(f, best_snap, r, c, 0); return j_best; }
      if (have_split) { area_copy_dev<BS, NW>(f, split_snap, r, c, 0); return j_split; }
    }
    if (must_split || j_split < j_none) return j_split;
    if constexpr (BS <= MAXBS) { area_copy_dev<BS, NW>(f, k.snap() + snap_level_off<MAXN>(BS, MAXBS), r, c, 0); }
    return j_none;
  }
};
template <int MAXN, int MAXBS, int NW> struct RdPartBU<MAXN, MAXBS, 0, NW> {
  template <int TS> static __device__ MI_K1_WALK_INLINE long long run(const Ctx<MAXN, NW, TS> k, int r, int c) {
    const LDS FrameDev *f = k.f();
    if (r >= f->mi_rows || c >= f->mi_cols) return 0;
    return uni64(blk_eval<MAXN, 0, NW>(k, r, c));
  }
};

// The two block-size classes of the search: MAXBS = 2 -- blocks up to 16x16 (MAXN = 16: every speed from 3 on, and the high-quality end of speeds 1 and 2) --
// and MAXBS = 4 -- blocks up to 64x64: the generic one-candidate-per-wavefront search up to 32x32 (MAXN = 32) plus the 64x64 level of dev_blk64.h.
constexpr int k1_maxn(int maxbs) { return maxbs <= 2 ? 16 : 32; }
#define MI_K1_POOL_BYTES(maxbs) ((maxbs) <= 2 ? (size_t)MI_SNAP_BYTES_ALL(16) : (size_t)MI_SNAP_BYTES_ALL(64) + MI_BLK64_HBM_BYTES)   /* HBM scratch per persistent workgroup */
template <int MAXBS, int NW> constexpr size_t k1_lds_bytes() {
  constexpr int MAXN = k1_maxn(MAXBS);
  return ((sizeof(SharedScratch<MAXN>) + 15) & ~(size_t)15) + NW * ((sizeof(WaveScratch<MAXN>) + 15) & ~(size_t)15) + SCAN_LDS_ENTRIES(MAXN) * 2 +
         ((COEF_COST_MAX_ENTRIES(MAXBS) * 2 + 15) & ~(size_t)15) + (size_t)FRAMEDEV_K1_BYTES;
}

static_assert(MI_PROFILE || k1_lds_bytes<2, 4>() <= 40960, "K1 <2,4> must fit four workgroups per CU (160 KB LDS)");
static_assert(MI_PROFILE || k1_lds_bytes<4, 4>() <= 163840, "K1 <4,4> must fit the CU's 160 KB of LDS");
static_assert(sizeof(Blk64Wave) <= offsetof(WaveScratch<32>, lev) - offsetof(WaveScratch<32>, pred), "the 64x64 candidate's working set lies over the wave's block scratch, not over its level maps");

// ---- the launch: a work queue of superblocks ----
// Persistent workgroups (as many as fit the GPU, or one per item when there are fewer) claim superblocks from one list per launch.  An item is
// (tile job, superblock row, superblock column inside the tile); the list is sorted by 2 * row + column (then by job), so everything an item waits
// for -- its left neighbour and the superblock two to the right in the row above (its above-right neighbour must be final) -- sits earlier in the
// list and has been claimed by a workgroup that is running or done: no deadlock, whatever the number of resident workgroups.  Legal because the
// search prices against static rate tables: a superblock depends on its left / above / above-right neighbours' reconstruction, mode info and decoded
// flags only; the block order inside a superblock and every decision are those of a serial walk of the tile.  A row's counter in f->sb_prog counts
// its finished superblocks (they finish in column order because each waits for its left neighbour); release / acquire at device scope.
// Measured on the MI355X (profiles/r03_variants_ab.txt): 1024 tiles of a 32 x 1080p batch, one workgroup per tile 159.5 ms, this queue 139.2 ms
// (no resident slot idles while its tile's neighbours are still at work); it also replaces the row workers single images used to get.
struct SbItem { uint32_t job; uint16_t sbr, sbc; };
// BU: the bottom-up partition walker (speed <= 2) is a separate instantiation so that the top-down kernels do not carry its code
template <int MAXBS, int NW, bool BU, int TS>
__global__ __launch_bounds__(64 * NW, (NW == 4 && MAXBS == 2 ? MI_K1_WG_PER_CU : 1)) void tile_search_kernel(const FrameDev *__restrict__ frames, const TileJob *__restrict__ jobs,
                                                                                                          const SbItem *__restrict__ items, int nitems, int *next_item, uint8_t *snap_pool, int items_per_wg) {
  constexpr int MAXN = k1_maxn(MAXBS);
  extern __shared__ __align__(16) uint8_t smem[];
  using K = Ctx<MAXN, NW, TS>;
  K k;
  k.base = (LDS uint8_t *)smem;
  k.ws = (LDS WaveScratch<MAXN> *)(smem + K::SH_BYTES + (size_t)(NW > 1 ? WAVE_ID : 0) * K::WS_BYTES);
  LDS uint16_t *lsc = (LDS uint16_t *)(smem + K::LS_OFF);
  LDS FrameDev *lf = (LDS FrameDev *)(smem + K::F_OFF);
  // the scan tables once per workgroup; the frame descriptor head and the coefficient slices of the rate table whenever the frame changes
  if (WAVE_ID == 0) load_scans_to_lds(lsc, MAXN);
  for (int i = LANE; i < (int)sizeof(k.s()->lev); i += 64) k.s()->lev[i] = 0;        // level-map padding stays zero for the workgroup's life
  if (threadIdx.x == 0) k.sh()->snap = snap_pool + (size_t)blockIdx.x * MI_K1_POOL_BYTES(MAXBS);
#if MI_PROFILE
  if (threadIdx.x < 128) ((LDS unsigned long long *)k.sh()->prof)[threadIdx.x] = 0;
  const FrameDev *prof_f = nullptr;
#endif
  int cur_frame = -1, cur_job = -1;
#if MI_PROFILE
  const unsigned long long k_t0_ = clock64(); unsigned long long w_acc_ = 0;
#endif
  // items_per_wg > 0: the workgroup leaves after that many work items and the launch has one workgroup per items_per_wg items -- workgroup slots come free all
  // through the launch, so the kernels of the other batch slots (entropy coder, loop filters) are dispatched beside this one instead of behind it.  The
  // no-deadlock argument is unchanged: workgroups are dispatched in index order and claim items in list order, so whatever an item waits for was claimed by a
  // workgroup that started earlier.  0: persistent workgroups (one per resident slot) that loop until the list is empty.
  for (int done_items = 0; items_per_wg <= 0 || done_items < items_per_wg; done_items++) {
#if MI_PROFILE
    const unsigned long long w_t0_ = clock64();
#endif
    WG_SYNC();                                                             // everyone is done with the previous item (and has read q_item)
    if (threadIdx.x == 0) k.sh()->q_item = atomicAdd(next_item, 1);
    WG_SYNC();
    const int it = k.sh()->q_item;
    if (it >= nitems) break;
    const SbItem item = items[it];
    const int job = (int)(item.job & 0x7fffffffu), sbr = item.sbr, sbc = item.sbc;
    const bool fine = (item.job >> 31) != 0;                               // per-root synchronisation (the whole launch has the flag or not)
    const TileJob tj = jobs[job];
    const FrameDev *gf = frames + tj.frame;
    if (frame_idle(gf)) continue;
    if (tj.frame != cur_frame) {
      for (int i = threadIdx.x; i < FRAMEDEV_K1_BYTES / 4; i += 64 * NW) ((LDS uint32_t *)lf)[i] = ((const uint32_t *)gf)[i];   // only the head: K1 never reads the tail through `lf`
      load_coef_cost(k.cc_base(), gf->cost, MAXBS, threadIdx.x, 64 * NW);
      load_part_cost(k.sh(), gf->cost, (int)threadIdx.x);
      cur_frame = tj.frame;
    }
    if (job != cur_job) {
      if (threadIdx.x == 0) {
        LDS TileB *t = &k.sh()->tile;
        t->mi_row_start = gf->tile_row_start[tj.tile_row] * 16; t->mi_row_end = imin_(gf->tile_row_start[tj.tile_row + 1] * 16, gf->mi_rows);
        t->mi_col_start = gf->tile_col_start[tj.tile_col] * 16; t->mi_col_end = imin_(gf->tile_col_start[tj.tile_col + 1] * 16, gf->mi_cols);
      }
    }
    const bool new_tile = job != cur_job;
    cur_job = job;
    if (threadIdx.x == 0) k.sh()->fine = fine;
    WG_SYNC();
    if (new_tile && gf->tile_cost != nullptr) {                            // second pass of a two-pass encode: the tile's own rate table
      const uint16_t *tc = gf->tile_cost + (size_t)(tj.tile_row * gf->tile_cols + tj.tile_col) * CDF_TOTAL;
      if (threadIdx.x == 0) lf->cost = tc;
      load_coef_cost(k.cc_base(), tc, MAXBS, threadIdx.x, 64 * NW);
      load_part_cost(k.sh(), tc, (int)threadIdx.x);
      WG_SYNC();
    }
    const int row0 = k.t()->mi_row_start, row1 = k.t()->mi_row_end, col0 = k.t()->mi_col_start, col1 = k.t()->mi_col_end;
    const int ncols = (col1 - col0 + 15) >> 4, nrows = (row1 - row0 + 15) >> 4, r = row0 + 16 * sbr, c = col0 + 16 * sbc;
    int *const prog = gf->sb_prog + (r >> 4) * gf->tile_cols + tj.tile_col;        // this row's counter; the row above: prog - tile_cols
    if (!fine && (sbc > 0 || sbr > 0)) {
      // Polling with relaxed loads (they bypass the non-coherent cache levels by themselves) and ONE acquire once both conditions hold: an acquire
      // per poll invalidates this XCD's L2 every few hundred cycles for as long as any workgroup waits, under the feet of the ones at work.
      // Bounded like root_wait (2^25 polls are tens of seconds): a wait that gives up marks the frame through its sticky error word, the entropy stage then fails
      // every tile of it and the host returns MI_ENCODING_ERROR -- a protocol error or a preempted device must not hang the GPU in the default (coarse) mode either.
      if (threadIdx.x == 0) {
        unsigned spin = 0;
        auto gave_up = [&]() { return (spin & 4095u) == 0u && __hip_atomic_load(search_error_word(gf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; };   // another wait of the frame did
        if (sbc > 0) while (__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sbc && spin < MI_K1_POLL_MAX) { __builtin_amdgcn_s_sleep(16); spin++; if (gave_up()) spin = MI_K1_POLL_MAX; }
        if (sbr > 0) { const int need = imin_(sbc + 2, ncols); while (__hip_atomic_load(prog - gf->tile_cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && spin < MI_K1_POLL_MAX) { __builtin_amdgcn_s_sleep(16); spin++; if (gave_up()) spin = MI_K1_POLL_MAX; } }
        if (spin >= MI_K1_POLL_MAX) __hip_atomic_store(search_error_word(gf), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      WG_SYNC();
      MI_K1_ACQUIRE();
    }
    unsigned long long *tc = gf->tile_clk + (size_t)(tj.tile_row * gf->tile_cols + tj.tile_col) * 4;
    if (threadIdx.x == 0 && sbr == 0 && sbc == 0) tc[0] = wall_clock64();
#if MI_PROFILE
    w_acc_ += clock64() - w_t0_;
#endif
    if constexpr (BU) RdPartBU<MAXN, MAXBS, 4, NW>::run(k, r, c); else RdPart<MAXN, MAXBS, 4, NW>::run(k, r, c, -1);
    WG_SYNC();                                                             // every wave's stores of this superblock are issued
    if (threadIdx.x == 0) {
      if (!fine) __hip_atomic_store(prog, sbc + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (sbr == nrows - 1 && sbc == ncols - 1) tc[1] = wall_clock64();
    }
#if MI_PROFILE
    prof_f = gf;
#endif
  }
  // the last workgroup to leave zeroes the launch's counter pair (claims, departures) for the next launch: no memset per encode
  if (threadIdx.x == 0 && atomicAdd(next_item + 1, 1) == (int)gridDim.x - 1) { next_item[1] = 0; next_item[0] = 0; }
#if MI_PROFILE
  if (LANE == 0) { k.sh()->prof[WAVE_ID][0] = w_acc_; k.sh()->prof[WAVE_ID][15] = clock64() - k_t0_; }   // claim + dependency waits; the workgroup's life
  WG_SYNC();
  if (prof_f && prof_f->prof_out && threadIdx.x < 128) prof_f->prof_out[(size_t)blockIdx.x * 128 + threadIdx.x] = ((LDS unsigned long long *)k.sh()->prof)[threadIdx.x];
#endif
}

// mi_avif.hip -- the engine behind include/mi_avif.h: plans AV1 frames, owns the HBM arena of a batch,
// launches the kernels (K0 front end, K1 tile search, K2 deblock, K3 CDEF, K4 tile entropy coding, pack),
// and assembles OBUs + AVIF containers on the host.  One HIP stream per batch, no hidden device syncs
// other than the two points where the host needs device results (alpha flags, tile lengths).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <memory>
#include <thread>
#include <atomic>
#include <future>
#include <chrono>
#include <exception>
#include <algorithm>
#include "host_av1.h"
#include "png_reader.h"
#include "tile_search.h"
#include "tile_entropy.h"
#include "loopfilter.h"
#include "restoration.h"

// The product library reads no environment variables; probe builds (-DMI_TUNING_KNOBS: tools/) get MI_AVIF_TIMING=1 (host-side timeline on stderr),
// MI_K1_GRID_PER_CU=n (fewer persistent search workgroups per CU) and, with -DMI_DEBUG_HOOKS=1, MI_DEBUG_LEVEL (bisect levels of the tile search).
// the streaming form's rotation: resident batch objects per image shape, and the first run's share of a full run
#ifndef MI_STREAM_SLOTS_DEFAULT
#define MI_STREAM_SLOTS_DEFAULT 2          /* 256 x 1080p files end to end: 1.32 s with two, 1.48 with three, 1.51 with four (profiles/r05zk_e2e_knobs.txt, matrix 5) */
#endif
#define MI_STREAM_FIRST_RUN_NUM 1         /* a worker's first run as a share of a full one: a whole run (a half run started the GPU ~0.03 s earlier and cost two runs of 16 images, */
#define MI_STREAM_FIRST_RUN_DEN 1         /* 0.09 s of tile search each against 0.10 for 32: worker phase 1.24 -> 1.20 s on 256 files, profiles/r05zk_e2e_knobs.txt matrix 6) */
static inline bool mi_timing_enabled() {
#ifdef MI_TUNING_KNOBS
  static const bool on = getenv("MI_AVIF_TIMING") != nullptr; return on;
#else
  return false;
#endif
}
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "mi_avif: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return MI_ENCODING_ERROR; } } while (0)

namespace mi {

__global__ void pack_tiles_kernel(const FrameDev *frames, const TileJob *jobs, int njobs, const uint32_t *offsets, uint8_t *packed) {
  const int job = blockIdx.x;
  if (job >= njobs) return;
  const TileJob tj = jobs[job];
  const FrameDev *f = frames + tj.frame;
  const int ti = tj.tile_row * f->tile_cols + tj.tile_col;
  const uint32_t len = f->tile_len[ti];
  const uint8_t *src = f->tile_out + (size_t)ti * f->tile_out_cap;
  uint8_t *dst = packed + offsets[job];
  for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) dst[i] = src[i];
}

// ---- two-pass pricing (mi_av1_config.rdo_passes = 2): the rate table of the CDFs a tile ended its first pass with ----
__device__ inline uint32_t neg_log2_q9_dev(uint32_t p) {      // host_av1.h neg_log2_q9, integer only: (15 - log2 p) * 512
  if (p < 1u) p = 1u;
  const int msb = 31 - __clz(p);
  unsigned long long x = (unsigned long long)p << (31 - msb);
  uint32_t frac = 0;
  for (int i = 0; i < 9; i++) { x = (x * x) >> 31; frac <<= 1; if (x >> 32) { frac |= 1; x >>= 1; } }
  return (uint32_t)(15 * 512 - (msb * 512 + (int)frac));
}
// grid (tiles, frames): tile t of the frame; also switches the frame over to its second pass (tile_cost set, cdf_out cleared) -- by the
// block of tile 0, after a grid-wide ... no: by a separate tiny launch (pass_flip_kernel), the frames are read by every block here
__global__ __launch_bounds__(256) void cdf_cost_kernel(const FrameDev *frames) {
  const FrameDev *f = frames + blockIdx.y;
  const int tile = blockIdx.x;
  if (tile >= f->tile_cols * f->tile_rows || f->cdf_out == nullptr || frame_idle(f)) return;
  const uint16_t *cdf = f->cdf_out + (size_t)tile * CDF_TOTAL;
  uint16_t *cost = f->tile_cost_buf + (size_t)tile * CDF_TOTAL;
  for (int i = threadIdx.x; i < CDF_TOTAL; i += 256) cost[i] = 0;
  __syncthreads();
#define MI_ROW_(o, st, n, k) for (int i = threadIdx.x; i < (n) * (k); i += 256) { const int r = i / (k), s = i - r * (k); const uint16_t *row = cdf + (o) + r * (st); \
    cost[(o) + r * (st) + s] = (uint16_t)neg_log2_q9_dev((s > 0 ? (uint32_t)row[s - 1] : 32768u) - (uint32_t)row[s]); }
  MI_COST_ROWS(MI_ROW_)
#undef MI_ROW_
}
__global__ void pass_flip_kernel(FrameDev *frames, int nframes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nframes && frames[i].cdf_out != nullptr) { frames[i].tile_cost = frames[i].tile_cost_buf; frames[i].cdf_out = nullptr; }
}

// ---- per-device read-only tables ----
struct DeviceTables { uint16_t *cost[4] = { 0, 0, 0, 0 }; uint16_t *cdf0[4] = { 0, 0, 0, 0 }; bool ready = false; };
static std::mutex g_tab_mu;
#define MI_MAX_DEVICES 64
static DeviceTables g_tabs[MI_MAX_DEVICES];
// First launch of anything from this library makes the runtime load the gfx950 code object (~2 MB, 0.1-0.2 s): ensure_tables
// does it once per device with an empty kernel, so callers can overlap it with their allocations (mi_ravif_encode_stream does).
__global__ void module_warm_kernel() {}
static int ensure_tables(int dev) {
  std::lock_guard<std::mutex> lk(g_tab_mu);
  if (dev < 0 || dev >= MI_MAX_DEVICES) { fprintf(stderr, "mi_avif: HIP ordinal %d outside the supported 0..%d\n", dev, MI_MAX_DEVICES - 1); return MI_INVALID_ARGUMENT; }
  DeviceTables &t = g_tabs[dev];
  if (t.ready) return MI_OK;
  for (int q = 0; q < 4; q++) {
    const std::vector<uint16_t> cost = build_cost_table(q);
    HIP_OK(hipMalloc(&t.cost[q], CDF_TOTAL * 2)); HIP_OK(hipMalloc(&t.cdf0[q], CDF_TOTAL * 2));
    HIP_OK(hipMemcpy(t.cost[q], cost.data(), CDF_TOTAL * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(t.cdf0[q], av1_default_cdfs + (size_t)q * CDF_TOTAL, CDF_TOTAL * 2, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(module_warm_kernel, dim3(1), dim3(64), 0, 0);
  HIP_OK(hipDeviceSynchronize());
  t.ready = true;
  return MI_OK;
}

// ---- one AV1 frame (colour image or alpha plane) inside a batch ----
struct FramePlan {
  mi_av1_config cfg{}; int np = 3, image = 0; bool is_alpha = false;
  int mi_cols = 0, mi_rows = 0, sb_cols = 0, sb_rows = 0, pw = 0, ph = 0, mi_stride = 0, mi_h = 0, ntiles = 0, maxbs = 2;
  QuantSel q{}; Tiling tiles; FrameHeaderInfo hdr{};
  FrameDev dev{};
  size_t arena_bytes = 0; uint8_t *arena = nullptr;
  std::vector<uint8_t> obu;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct FramePlan;
static size_t zeroed_bytes(const FramePlan &p);

static int lr_units_host(uint32_t size) { const int n = ((int)size + 32) / 64; return n < 1 ? 1 : n; }
static size_t zeroed_bytes(const FramePlan &p) { return align_up((size_t)p.mi_stride * p.mi_h, 256) + align_up(6 * 65 * sizeof(long long), 256) + ((size_t)p.sb_rows * p.tiles.cols + (size_t)p.sb_rows * p.sb_cols + 1) * sizeof(int); }   // decoded flags, deblock tallies, K1's per-row counters and per-superblock root masks
static void plan_geometry(FramePlan &p) {
  const mi_av1_config &c = p.cfg;
  p.np = c.chroma == 1 ? 1 : 3;
  p.mi_cols = 2 * ((c.width + 7) >> 3); p.mi_rows = 2 * ((c.height + 7) >> 3);
  p.sb_cols = (p.mi_cols + 15) >> 4; p.sb_rows = (p.mi_rows + 15) >> 4;
  p.pw = p.sb_cols * 64; p.ph = p.sb_rows * 64; p.mi_stride = p.pw / 4; p.mi_h = p.ph / 4;
  int part_max = c.part_max, part_min = c.part_min;
  if (part_min > part_max) part_min = part_max;
  p.cfg.part_max = (uint8_t)part_max; p.cfg.part_min = (uint8_t)part_min;
  p.maxbs = part_max <= 16 ? 2 : 4;                   // the search's two block-size classes (tile_search.h k1_maxn): up to 16x16, up to 64x64
  p.q = select_quantizers(c.quantizer, c.bit_depth, p.np);
  p.tiles = plan_tiles((int)c.width, (int)c.height, p.sb_cols, p.sb_rows, c.min_tile_size, c.threads, c.tiles_override);
  p.ntiles = p.tiles.cols * p.tiles.rows;
}

// carve the frame's arena; returns bytes needed (dry run when base == nullptr)
static size_t carve(FramePlan &p, uint8_t *base, uint32_t tile_cap) {
  size_t off = 0;
  auto take = [&](size_t bytes) { uint8_t *ptr = base ? base + off : nullptr; off = align_up(off + bytes, 256); return ptr; };
  const size_t npx = (size_t)p.pw * p.ph, nmi = (size_t)p.mi_stride * p.mi_h;
  FrameDev &d = p.dev;
  for (int i = 0; i < p.np; i++) {
    d.src[i] = (uint16_t *)take(npx * 2); d.rec[i] = (uint16_t *)take(npx * 2); d.fin[i] = (uint16_t *)take(npx * 2);
    d.coef[i] = (int32_t *)take(npx * 4);
    d.m_lvl[i] = take(nmi); d.m_dc[i] = take(nmi); d.m_eob[i] = (uint16_t *)take(nmi * 2);
  }
  d.m_bsize = take(nmi); d.m_skip = take(nmi); d.m_ymode = take(nmi); d.m_uvmode = take(nmi); d.m_txtype = take(nmi);
  d.m_cfl_sign = take(nmi); d.m_cfl_au = take(nmi); d.m_cfl_av = take(nmi); d.m_txsize = take(nmi);
  // state the kernels expect zeroed before every encode, in one block (one memset): decoded flags + deblock tallies
  d.m_decoded = take(zeroed_bytes(p)); d.lf_tally = (long long *)(d.m_decoded + align_up(nmi, 256));
  d.sb_prog = (int *)(d.m_decoded + align_up(nmi, 256) + align_up(6 * 65 * sizeof(long long), 256));
  d.zero_words = (int)((zeroed_bytes(p) + 3) / 4);
  d.lf_out = (int *)take(64);                              // 4 deblock levels, segment count, 8 segment indices
  d.seg = (const SegTab *)take(sizeof(SegTab));
  d.m_angle_y = (int8_t *)take(nmi); d.m_angle_uv = (int8_t *)take(nmi);
  d.cdef_idx = (int8_t *)take((size_t)p.sb_cols * p.sb_rows);
  { const size_t ncell = (size_t)(p.pw / 8) * (p.ph / 8); d.act = (const uint32_t *)take(ncell * 4); d.svar8 = (const uint32_t *)take(ncell * 4); d.svar4 = (const uint32_t *)take(nmi * 4); }
  {
    const size_t nlr = (size_t)lr_units_host(p.cfg.width) * lr_units_host(p.cfg.height) * p.np;
    for (int i = 0; i < 3; i++) d.lrp[i] = (i < p.np && p.cfg.lrf) ? (uint16_t *)take(npx * 2) : nullptr;
    d.lr_type = take(nlr); d.lr_set = take(nlr); d.lr_xqd = (int8_t *)take(nlr * 2);
    d.lr_cand = p.cfg.lrf ? take(nlr * 16 * sizeof(LrCand)) : nullptr;
  }
  d.tile_out = take((size_t)p.ntiles * tile_cap);
  d.tile_len = (uint32_t *)take((size_t)p.ntiles * 4);
  d.tile_clk = (unsigned long long *)take((size_t)p.ntiles * 32);
  d.tile_cost = nullptr; d.cdf_out = nullptr; d.tile_cost_buf = nullptr;
  if (p.cfg.rdo_passes == 2) { d.cdf_out = (uint16_t *)take((size_t)p.ntiles * CDF_TOTAL * 2); d.tile_cost_buf = (uint16_t *)take((size_t)p.ntiles * CDF_TOTAL * 2); }
  d.prof_out = nullptr;
  d.tile_out_cap = tile_cap;
  return off;
}

static uint32_t tile_capacity(const FramePlan &p) {
  int tw = 0, th = 0;
  for (int i = 0; i < p.tiles.cols; i++) tw = std::max(tw, p.tiles.col_start[i + 1] - p.tiles.col_start[i]);
  for (int i = 0; i < p.tiles.rows; i++) th = std::max(th, p.tiles.row_start[i + 1] - p.tiles.row_start[i]);
  const size_t px = (size_t)tw * th * 4096;
  return (uint32_t)align_up(px * p.np * 2 + 4096, 256);
}

static void fill_dev(FramePlan &p, const DeviceTables &tab) {
  FrameDev &d = p.dev; const mi_av1_config &c = p.cfg;
  d.w = c.width; d.h = c.height; d.bd = c.bit_depth; d.np = p.np;
  d.mi_cols = p.mi_cols; d.mi_rows = p.mi_rows; d.sb_cols = p.sb_cols; d.sb_rows = p.sb_rows;
  d.pw = p.pw; d.ph = p.ph; d.stride = p.pw; d.mi_stride = p.mi_stride; d.mi_h = p.mi_h;
  d.base_q_idx = p.q.base_q_idx; d.qctx = p.q.qctx; d.rdmult = p.q.rdmult;
  d.seg_n = 0;
  for (int i = 0; i < 3; i++) { d.seg_ddc[i] = i < p.np ? p.q.dc_qi[i] - p.q.base_q_idx : 0; d.seg_dac[i] = i < p.np ? p.q.ac_qi[i] - p.q.base_q_idx : 0; }
  for (int i = 0; i < 3; i++) { d.dc_q[i] = p.q.dc_q[i]; d.ac_q[i] = p.q.ac_q[i]; d.wq[i] = p.q.wq[i]; d.dc_recip[i] = 0xFFFFFFFFu / (uint32_t)std::max(1, p.q.dc_q[i]); d.ac_recip[i] = 0xFFFFFFFFu / (uint32_t)std::max(1, p.q.ac_q[i]); }
  d.part_min = c.part_min; d.part_max = c.part_max; d.complex_modes = c.complex_pred_modes; d.fine_directional = c.fine_directional_intra;
  d.bottomup = c.encode_bottomup;
  d.tx_mode_select = c.rdo_tx_decision || c.inter_tx_split;    // rav1e FrameInvariants.tx_mode_select (recall)
  d.rdo_tx = c.rdo_tx_decision; d.reduced_tx_set = c.reduced_tx_set; d.enable_cdef = c.cdef; d.fast_deblock = c.fast_deblock;
  d.enable_restoration = c.lrf; d.sgr_full = c.sgr_full; d.tune_psnr = c.tune_psnr;
  { const uint32_t cdf[3] = { 9413, 22581, 32768 }; uint32_t lo = 0; for (int i = 0; i < 3; i++) { d.lr_cost[i] = neg_log2_q9(cdf[i] - lo); lo = cdf[i]; } }   // libaom default_switchable_restore_cdf
  d.tile_cols = p.tiles.cols; d.tile_rows = p.tiles.rows; d.tile_cols_log2 = p.tiles.cols_log2; d.tile_rows_log2 = p.tiles.rows_log2;
  for (int i = 0; i <= p.tiles.cols; i++) d.tile_col_start[i] = p.tiles.col_start[i];
  for (int i = 0; i <= p.tiles.rows; i++) d.tile_row_start[i] = p.tiles.row_start[i];
  d.cost = tab.cost[p.q.qctx]; d.cdf0 = tab.cdf0[p.q.qctx];
#if MI_DEBUG_HOOKS
  d.dbg = getenv("MI_DEBUG_LEVEL") ? atoi(getenv("MI_DEBUG_LEVEL")) : 0;
#else
  d.dbg = 0;
#endif
  // fast_deblock: the q formula; otherwise K2a searches the levels on the device and the host reads them back for the header
  const int lvl = c.fast_deblock ? deblock_level_from_q(p.q.ac_q[0], c.bit_depth) : 0;
  d.lf_level[0] = d.lf_level[1] = d.lf_level[2] = d.lf_level[3] = lvl; d.lf_sharp = 0;
  static const int strengths[8] = { 0, 1 * 4 + 0, 2 * 4 + 1, 3 * 4 + 1, 5 * 4 + 2, 7 * 4 + 3, 10 * 4 + 3, 13 * 4 + 3 };   // rav1e's fixed list
  d.cdef_damping = 3; d.cdef_bits = 3;
  for (int i = 0; i < 8; i++) { d.cdef_y[i] = strengths[i]; d.cdef_uv[i] = strengths[i]; }
  FrameHeaderInfo &h = p.hdr;
  h.cfg = c; h.np = p.np; h.sb_cols = p.sb_cols; h.sb_rows = p.sb_rows; h.q = p.q; h.tiles = p.tiles;
  for (int i = 0; i < 4; i++) h.lf_level[i] = d.lf_level[i];
  h.seg_n = 0; for (int i = 0; i < 8; i++) h.seg_qidx[i] = p.q.base_q_idx;
  h.lf_sharp = 0; h.enable_cdef = c.cdef; h.cdef_damping = 3; h.cdef_bits = 3; h.enable_restoration = c.lrf; h.tx_mode_select = d.tx_mode_select;
  for (int i = 0; i < 8; i++) { h.cdef_y[i] = strengths[i]; h.cdef_uv[i] = strengths[i]; }
}

// ---- K1 launch: the tile search as a work queue of superblocks (tile_search.h) ----
// Persistent workgroups: as many as the device holds at once for this instantiation (asked from the runtime, not assumed), capped by the number of items.
#ifndef MI_K1_ITEMS_PER_WG_DEFAULT
#define MI_K1_ITEMS_PER_WG_DEFAULT 0
#endif
template <int MAXBS, int NW, bool BU, int TS> static hipError_t launch_search_t(const FrameDev *d_frames, const TileJob *d_jobs, const SbItem *d_items, int nitems, int *d_next, uint8_t *d_snap_pool, int *grid_out, int device, hipStream_t s) {
  const size_t lds = k1_lds_bytes<MAXBS, NW>();
  static int resident[MI_MAX_DEVICES];                    // per instantiation and device; 0 = not asked yet
  if (resident[device] == 0) {
    hipError_t e = hipFuncSetAttribute((const void *)tile_search_kernel<MAXBS, NW, BU, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int per_cu = 0, cus = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)tile_search_kernel<MAXBS, NW, BU, TS>, 64 * NW, lds);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return e;
    resident[device] = std::max(1, per_cu) * std::max(1, cus);
#ifdef MI_TUNING_KNOBS                                  // probe builds only: MI_K1_GRID_PER_CU=n asks for fewer persistent workgroups per CU than fit
    if (const char *v = getenv("MI_K1_GRID_PER_CU")) { const int n = atoi(v); if (n > 0 && n < per_cu) resident[device] = n * std::max(1, cus); }
#endif
  }
  // MI_K1_ITEMS_PER_WG=n (n > 0): workgroups that leave after n items instead of persistent ones (tile_search.h); only when the list is longer than the device holds
#ifdef MI_TUNING_KNOBS                                  // probe builds only: the product library reads no environment variable on the encode path
  static const int ipw_env = getenv("MI_K1_ITEMS_PER_WG") ? atoi(getenv("MI_K1_ITEMS_PER_WG")) : MI_K1_ITEMS_PER_WG_DEFAULT;
#else
  static const int ipw_env = MI_K1_ITEMS_PER_WG_DEFAULT;
#endif
  const int ipw = (ipw_env > 0 && nitems > resident[device]) ? ipw_env : 0;
  const int grid = ipw ? (nitems + ipw - 1) / ipw : std::min(nitems, resident[device]);
  if (grid_out) { *grid_out = grid; return hipSuccess; }   // dry run: the caller sizes the snapshot pool
  hipLaunchKernelGGL((tile_search_kernel<MAXBS, NW, BU, TS>), dim3(grid), dim3(64 * NW), lds, s, d_frames, d_jobs, d_items, nitems, d_next, d_snap_pool, ipw);
  return hipGetLastError();
}
static size_t k1_snap_bytes(int maxbs) { return MI_K1_POOL_BYTES(maxbs); }
// every frame of a launch comes from one encoder configuration, so the partition order (top-down / bottom-up) is per launch; the
// jobs must all belong to frames of the same block-size class (one instantiation per class).  grid_out != nullptr: only report the grid.
// `tools`: the tool set the kernels are instantiated for (tile_search.h Tools): bit 0 = the full candidate set of speed <= 1 (complex_pred_modes), bit 1 = the switches of
// ravif's speed 4 as constants -- instantiated where speed 4 runs (blocks up to 16x16, top-down); any other combination of switches runs the general kernels
static hipError_t launch_search(int maxbs, bool bottomup, int tools, const FrameDev *d_frames, const TileJob *d_jobs, const SbItem *d_items, int nitems, int *d_next, uint8_t *d_snap_pool, int *grid_out, int device, hipStream_t s) {
  if (nitems <= 0) { if (grid_out) *grid_out = 0; return hipSuccess; }
#define MI_LAUNCH_(MB, BU_, TS_) launch_search_t<MB, 4, BU_, TS_>(d_frames, d_jobs, d_items, nitems, d_next, d_snap_pool, grid_out, device, s)
#ifdef MI_FAST_BUILD                                     // experiment builds only (tools/build_variant.sh): the headline configuration's instantiation and nothing else
  return MI_LAUNCH_(2, false, 2);
#else
  const bool full = (tools & 1) != 0;
  if (maxbs <= 2 && !bottomup && tools == 2) return MI_LAUNCH_(2, false, 2);
  if (maxbs <= 2) return bottomup ? (full ? MI_LAUNCH_(2, true, 1) : MI_LAUNCH_(2, true, 0)) : (full ? MI_LAUNCH_(2, false, 1) : MI_LAUNCH_(2, false, 0));
  return bottomup ? (full ? MI_LAUNCH_(4, true, 1) : MI_LAUNCH_(4, true, 0)) : (full ? MI_LAUNCH_(4, false, 1) : MI_LAUNCH_(4, false, 0));
#endif
#undef MI_LAUNCH_
}
// jobs must all belong to frames of the same block-size class
// K4, one instantiation per block-size class like K1 (jobs + first_job .. first_job + njobs of the grouped job list)
static hipError_t launch_entropy(int maxbs, const FrameDev *d_frames, const TileJob *d_jobs, int njobs, uint16_t *d_precarry, uint32_t pre_cap, uint32_t *d_recbuf, uint32_t rec_cap, hipStream_t s) {
  if (njobs <= 0) return hipSuccess;
  // a launch that leaves wave slots free (fewer than 512 tiles: 6 waves each still fit the device in one round) runs four adapter waves per tile
  const bool sparse = njobs < 512;
#ifdef MI_FAST_BUILD
  hipLaunchKernelGGL((tile_entropy_kernel<2, MI_K4_ADAPTERS>), dim3(njobs), dim3(MI_K4_THREADS_OF(MI_K4_ADAPTERS)), sizeof(EntropyLds<16>), s, d_frames, d_jobs, njobs, d_precarry, pre_cap, d_recbuf, rec_cap);
  return hipGetLastError();
#endif
  if (maxbs <= 2) {
    if (sparse) hipLaunchKernelGGL((tile_entropy_kernel<2, MI_K4_ADAPTERS_SPARSE>), dim3(njobs), dim3(MI_K4_THREADS_OF(MI_K4_ADAPTERS_SPARSE)), sizeof(EntropyLds<16>), s, d_frames, d_jobs, njobs, d_precarry, pre_cap, d_recbuf, rec_cap);
    else hipLaunchKernelGGL((tile_entropy_kernel<2, MI_K4_ADAPTERS>), dim3(njobs), dim3(MI_K4_THREADS_OF(MI_K4_ADAPTERS)), sizeof(EntropyLds<16>), s, d_frames, d_jobs, njobs, d_precarry, pre_cap, d_recbuf, rec_cap);
  } else {
    if (sparse) hipLaunchKernelGGL((tile_entropy_kernel<4, MI_K4_ADAPTERS_SPARSE>), dim3(njobs), dim3(MI_K4_THREADS_OF(MI_K4_ADAPTERS_SPARSE)), sizeof(EntropyLds<32>), s, d_frames, d_jobs, njobs, d_precarry, pre_cap, d_recbuf, rec_cap);
    else hipLaunchKernelGGL((tile_entropy_kernel<4, MI_K4_ADAPTERS>), dim3(njobs), dim3(MI_K4_THREADS_OF(MI_K4_ADAPTERS)), sizeof(EntropyLds<32>), s, d_frames, d_jobs, njobs, d_precarry, pre_cap, d_recbuf, rec_cap);
  }
  return hipGetLastError();
}

// The work list of a set of tile jobs (grouped by block-size class, class_begin[2..5]) and the device objects a queue launch needs: the items, the claim
// counters (self-resetting: the last workgroup to leave a launch zeroes its pair), one snapshot area per persistent workgroup.
struct FramePlan;
struct SearchQueue {
  std::vector<SbItem> items; int q_begin[6] = { 0, 0, 0, 0, 0, 0 };
  SbItem *d_items = nullptr; SbItem *h_items = nullptr; size_t items_cap = 0; int *d_next = nullptr; uint8_t *d_snap = nullptr; size_t snap_bytes = 0;
  void free_device() {
    if (d_items) (void)hipFree(d_items); if (h_items) (void)hipHostFree(h_items); if (d_next) (void)hipFree(d_next); if (d_snap) (void)hipFree(d_snap);
    d_items = nullptr; h_items = nullptr; d_next = nullptr; d_snap = nullptr; items_cap = 0; snap_bytes = 0;
  }
};
}  // namespace mi
namespace mi {
// The frame-level stages between the tile search and the entropy coder, shared by the batch and the single-frame entry
// points: K2a deblock level search -> level pick -> K2 deblock (vertical, horizontal edges) -> K3 CDEF.
static hipError_t launch_loop_filters(FrameDev *d_frames, int nframes, int max_mi_cells, int max_sb, int max_lr_units, int max_lr_sets, hipStream_t s, hipEvent_t ev_cdef) {
  hipLaunchKernelGGL(deblock_tally_kernel, dim3((max_mi_cells + MI_DBK_CHUNK - 1) / MI_DBK_CHUNK, 6, nframes), dim3(256), 0, s, d_frames, nframes);
  hipLaunchKernelGGL(deblock_pick_kernel, dim3((nframes + 63) / 64), dim3(64), 0, s, d_frames, nframes);
  for (int pass = 0; pass < 2; pass++)
    hipLaunchKernelGGL(deblock_kernel, dim3((max_mi_cells + MI_DBK_CHUNK - 1) / MI_DBK_CHUNK, 3, nframes), dim3(256), 0, s, d_frames, nframes, pass);
  if (ev_cdef) { hipError_t e = hipEventRecord(ev_cdef, s); if (e != hipSuccess) return e; }
  hipLaunchKernelGGL(cdef_kernel, dim3(max_sb, nframes), dim3(256), 0, s, d_frames, 1);
  if (max_lr_units > 0) {
    hipLaunchKernelGGL(lr_search_kernel, dim3(max_lr_units, 3, nframes), dim3(256), 0, s, d_frames);
    hipLaunchKernelGGL(lr_kernel, dim3(max_lr_units, 3, nframes), dim3(256), 0, s, d_frames);
  }
  return hipGetLastError();
}

}  // namespace mi

namespace mi {
// Builds the launch's work list -- per block-size class, the superblocks of the class's tiles in (2 * row + column, job) order; jobs are indexed inside
// their class segment of d_jobs -- and enqueues one queue launch per class on `s`.
// the launches over a work list that is already on the device (the second pass of a two-pass encode reuses the first one's)
// every frame of a launch comes from one encoder configuration: which walker and which candidate set the kernels are instantiated for
// bit 0: bottom-up walker; bits 1..: the kernels' tool set (tile_search.h Tools).  The walker and the candidate set follow from the speed alone, the same for every frame of a
// launch; rdo_tx_decision also depends on the frame's quantiser (av1encoder.rs:576: `speed <= 4 && !high_quality`), and a launch holds the colour and the alpha frames of
// its pictures, each with its own quality: the kernels with the speed-4 switches as constants run only when EVERY frame of the launch has them.
static int search_mode(const std::vector<FramePlan> &frames) {
  if (frames.empty()) return 0;
  const mi_av1_config &c0 = frames[0].cfg;
  bool speed4_switches = true;
  for (const FramePlan &p : frames) {
    const mi_av1_config &c = p.cfg;
    speed4_switches = speed4_switches && !c.complex_pred_modes && c.rdo_tx_decision && c.reduced_tx_set && c.fine_directional_intra && !c.tune_psnr;   // (tx_mode_select follows from rdo_tx_decision)
  }
  return (c0.encode_bottomup != 0 ? 1 : 0) | (c0.complex_pred_modes != 0 ? 2 : 0) | (speed4_switches ? 4 : 0);
}
// The walker and the candidate set are template parameters of the launch (taken from frames[0]): every frame must agree on them.  They follow from the speed alone, which a
// batch shares, but mi_av1_config lets a caller override the resolved tweaks per call -- a mixed launch would run the wrong candidate set for some frames, silently.
static bool search_mode_consistent(const std::vector<FramePlan> &frames) {
  for (const FramePlan &p : frames)
    if ((p.cfg.encode_bottomup != 0) != (frames[0].cfg.encode_bottomup != 0) || (p.cfg.complex_pred_modes != 0) != (frames[0].cfg.complex_pred_modes != 0)) return false;
  return true;
}
static int search_launch(SearchQueue &q, int mode /* search_mode() */, const int class_begin[6], const FrameDev *d_frames, const TileJob *d_jobs, int device, hipStream_t s) {
  for (int cls = 2; cls <= 4; cls++)
    HIP_OK(launch_search(cls, (mode & 1) != 0, mode >> 1, d_frames, d_jobs + class_begin[cls], q.d_items + q.q_begin[cls], q.q_begin[cls + 1] - q.q_begin[cls], q.d_next + 2 * cls, q.d_snap, nullptr, device, s));
  return MI_OK;
}
// the queue's device objects hold `nitems` work items and `snap_need` bytes of area snapshots (grown, never shrunk).  hipFree / hipMalloc wait for the device:
// a batch object reserves its worst case when it is made (search_reserve), so that an encode whose image count grows does not stall behind the other slots' kernels
static int queue_fit(SearchQueue &q, size_t nitems, size_t snap_need, hipStream_t s) {
  if (nitems > q.items_cap) {
    if (q.d_items) (void)hipFree(q.d_items); if (q.h_items) (void)hipHostFree(q.h_items);
    q.d_items = nullptr; q.h_items = nullptr; q.items_cap = nitems + nitems / 8;
    HIP_OK(hipMalloc(&q.d_items, q.items_cap * sizeof(SbItem))); HIP_OK(hipHostMalloc(&q.h_items, q.items_cap * sizeof(SbItem)));
  }
  if (!q.d_next) { HIP_OK(hipMalloc(&q.d_next, 16 * sizeof(int))); HIP_OK(hipMemsetAsync(q.d_next, 0, 16 * sizeof(int), s)); }   // once: every launch leaves its pair zeroed
  if (snap_need > q.snap_bytes) { if (q.d_snap) (void)hipFree(q.d_snap); q.d_snap = nullptr; q.snap_bytes = snap_need; HIP_OK(hipMalloc(&q.d_snap, snap_need)); }
  return MI_OK;
}
static int search_reserve(SearchQueue &q, const std::vector<FramePlan> &frames, int device, hipStream_t s) {
  size_t per_class[5] = { 0, 0, 0, 0, 0 }, items = 0, snap_need = 0;
  for (const FramePlan &p : frames) { per_class[std::max(p.maxbs, 2)] += (size_t)p.sb_rows * p.sb_cols; items += (size_t)p.sb_rows * p.sb_cols; }
  const int mode = search_mode(frames);
  for (int cls = 2; cls <= 4; cls++) for (int tools : { mode >> 1, (mode >> 1) & ~2 }) {        // a run with fewer frames may run the other instantiation (search_mode)
    int grid = 0;
    HIP_OK(launch_search(cls, (mode & 1) != 0, tools, nullptr, nullptr, nullptr, (int)per_class[cls], nullptr, nullptr, &grid, device, s));
    snap_need = std::max(snap_need, (size_t)grid * k1_snap_bytes(cls));
  }
  return queue_fit(q, items, snap_need, s);
}
static int search_enqueue(SearchQueue &q, const std::vector<FramePlan> &frames, const std::vector<TileJob> &jobs, const int class_begin[6], const FrameDev *d_frames, const TileJob *d_jobs, int device, hipStream_t s) {
  q.items.clear();
  size_t snap_need = 0;
  if (!search_mode_consistent(frames)) return MI_INVALID_ARGUMENT;
  const int mode = search_mode(frames);
  for (int cls = 2; cls <= 4; cls++) {
    q.q_begin[cls] = (int)q.items.size();
    // Synchronisation grain and list order.  Blocks up to 16x16 (classes below 4): per root block (tile_search.h root_wait / root_publish), where a superblock can start
    // ~0.75 of a superblock time after its left neighbour and ~1.125 after the one above; the list is ordered by 3 * row + 2 * column -- any a * row + b * column
    // with a > b > 0 lists a superblock after its left and its above-right neighbour, and 1.5 columns per row is the closest of the small ratios to what the roots
    // allow.  Until round 5 a full batch used whole-superblock flags and 2 * row + column: K1 106.6 -> 103.0 ms on 32 x 1080p, 121.1 -> 110.6 ms with 16 tiles per
    // image, where the longest tile's chain and not the device's throughput bounds the launch (profiles/r05zr_k1_sync_grain_and_order.txt).  64x64 superblocks
    // are their own roots: whole-superblock flags, two columns per row.
    bool fine = cls < 4;
    int key_a = fine ? 3 : 2, key_b = fine ? 2 : 1;
#ifdef MI_TUNING_KNOBS
    if (const char *v = getenv("MI_K1_FINE")) fine = cls < 4 && atoi(v) != 0;
    if (const char *v = getenv("MI_K1_KEY")) { int a = 0, b = 0; if (sscanf(v, "%d,%d", &a, &b) == 2 && a > b && b > 0 && a < 64) { key_a = a; key_b = b; } }
#endif
    std::vector<std::vector<SbItem>> by_key;
    for (int j = class_begin[cls]; j < class_begin[cls + 1]; j++) {
      const TileJob &tj = jobs[j]; const FramePlan &p = frames[tj.frame];
      const int rows = std::min(p.tiles.row_start[tj.tile_row + 1], p.sb_rows) - p.tiles.row_start[tj.tile_row];
      const int cols = std::min(p.tiles.col_start[tj.tile_col + 1], p.sb_cols) - p.tiles.col_start[tj.tile_col];
      if ((int)by_key.size() < key_a * rows + key_b * cols) by_key.resize(key_a * rows + key_b * cols);
      for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) by_key[key_a * r + key_b * c].push_back(SbItem{ (uint32_t)(j - class_begin[cls]), (uint16_t)r, (uint16_t)c });
    }
    for (auto &v : by_key) q.items.insert(q.items.end(), v.begin(), v.end());
    const int nitems = (int)q.items.size() - q.q_begin[cls];
    int grid = 0;
    HIP_OK(launch_search(cls, (mode & 1) != 0, mode >> 1, nullptr, nullptr, nullptr, nitems, nullptr, nullptr, &grid, device, s));
    if (fine) for (int i = q.q_begin[cls]; i < (int)q.items.size(); i++) q.items[i].job |= 0x80000000u;
    snap_need = std::max(snap_need, (size_t)grid * k1_snap_bytes(cls));
  }
  q.q_begin[5] = (int)q.items.size();
  if (int st = queue_fit(q, q.items.size(), snap_need, s)) return st;
  memcpy(q.h_items, q.items.data(), q.items.size() * sizeof(SbItem));
  HIP_OK(hipMemcpyAsync(q.d_items, q.h_items, q.items.size() * sizeof(SbItem), hipMemcpyHostToDevice, s));
  return search_launch(q, mode, class_begin, d_frames, d_jobs, device, s);
}
}  // namespace mi

using namespace mi;

// ================================================================ batch object
struct mi_batch {
  mi_ravif_encoder enc{}; int n = 0; uint32_t w = 0, h = 0; int channels = 3, device = 0, depth = 10;
  std::vector<uint8_t> exif;                                       // the batch's own copy of enc.exif (the caller's buffer need not outlive mi_batch_create)
  hipStream_t stream = nullptr;
  hipStream_t stream_hi = nullptr; hipEvent_t ev_hi = nullptr;   // MI_POSTK1_PRIORITY=1 (probe): the stages after the tile search on a high-priority stream of their own
  int cap = 0;                                                     // images the batch was created for (n = images of the current run <= cap)
  uint8_t *d_pixels = nullptr; size_t pixel_bytes = 0;            // cap * w*h*channels
  uint8_t *h_pixels = nullptr;                                     // pinned staging of the same size: the H2D source (async, no pageable copies)
  int *d_alpha_flags = nullptr; std::vector<int> alpha_flags;
  uint8_t *d_clean = nullptr, *d_clean_tmp = nullptr; unsigned long long *d_alpha_acc = nullptr;   // dirty-alpha cleaner (RGBA, UnassociatedClean)
  std::vector<FramePlan> frames;                                  // colour frames [0..n), alpha frames after
  uint8_t *d_arena = nullptr; size_t arena_bytes = 0, aux_bytes = 0;   // aux: pre-carry units + symbol records
  FrameDev *d_frames = nullptr; TileJob *d_jobs = nullptr; uint16_t *d_precarry = nullptr; uint32_t pre_cap = 0;
  uint32_t *d_recbuf = nullptr; uint32_t rec_cap = 0;             // K4's symbol records: three rotating superblock buffers per tile
  uint32_t *d_offsets = nullptr; uint8_t *d_packed = nullptr; size_t packed_cap = 0, packed_max = 0; unsigned long long *d_prof = nullptr;
  int *h_alpha = nullptr; FrameDev *h_frames = nullptr; TileJob *h_jobs = nullptr;   // pinned: alpha flags (D2H), frame descriptors and tile jobs (H2D sources)
  uint8_t *h_packed = nullptr; uint32_t *h_lens = nullptr; int *h_lf = nullptr;   // pinned: packed tiles, tile lengths, deblock levels + segment indices (13 per frame)
  std::vector<TileJob> jobs;
  SearchQueue queue;                                               // the tile search's work list and its device objects (allocated on first use)
  std::vector<std::vector<uint8_t>> files; std::vector<size_t> color_sz, alpha_sz;
  hipEvent_t ev[8]{}; double stage_ms[8]{};
  bool planned = false, in_flight = false;
};

static int batch_plan(mi_batch *b, bool with_alpha_frames) {
  // (re)build frame plans: colour for every image [0, n), then (RGBA input) one alpha frame per image [n, 2n) -- whether an alpha
  // frame is used is decided on the device (FrameDev::active)
  b->frames.clear(); b->jobs.clear();
  const int quantizer = quality_to_quantizer(b->enc.quality), aquant = quality_to_quantizer(b->enc.alpha_quality);
  auto make = [&](int image, bool alpha) {
    FramePlan p; p.image = image; p.is_alpha = alpha;
    mi_av1_config &c = p.cfg;
    c.width = b->w; c.height = b->h; c.bit_depth = (uint8_t)b->depth; c.quantizer = (uint8_t)(alpha ? aquant : quantizer);
    c.chroma = alpha ? 1 : 0; c.pixel_range = 1; c.threads = b->enc.threads; c.device = b->device; c.tiles_override = b->enc.tiles_override; c.rdo_passes = (uint8_t)(b->enc.rdo_passes == 2 ? 2 : 1);
    c.has_color_desc = alpha ? 0 : 1; c.primaries = 1; c.transfer = 13; c.matrix = b->enc.color_model == 1 ? 0 : 6;
    tweaks_from_preset(b->enc.speed, c.quantizer, &c);
    plan_geometry(p);
    return p;
  };
  for (int i = 0; i < b->n; i++) b->frames.push_back(make(i, false));
  if (with_alpha_frames) for (int i = 0; i < b->n; i++) b->frames.push_back(make(i, true));
  return MI_OK;
}

static void batch_free_device(mi_batch *b) {
  if (b->d_arena) (void)hipFree(b->d_arena); b->d_arena = nullptr;
  if (b->d_frames) (void)hipFree(b->d_frames); b->d_frames = nullptr;
  if (b->d_jobs) (void)hipFree(b->d_jobs); b->d_jobs = nullptr;
  if (b->d_precarry) (void)hipFree(b->d_precarry); b->d_precarry = nullptr;
  if (b->d_recbuf) (void)hipFree(b->d_recbuf); b->d_recbuf = nullptr;
  if (b->d_offsets) (void)hipFree(b->d_offsets); b->d_offsets = nullptr;
  if (b->d_prof) (void)hipFree(b->d_prof); b->d_prof = nullptr;
  b->queue.free_device();
  if (b->d_packed) (void)hipFree(b->d_packed); b->d_packed = nullptr;
  if (b->h_packed) (void)hipHostFree(b->h_packed); b->h_packed = nullptr;
  if (b->h_lens) (void)hipHostFree(b->h_lens); b->h_lens = nullptr;
  if (b->h_lf) (void)hipHostFree(b->h_lf); b->h_lf = nullptr;
  if (b->h_alpha) (void)hipHostFree(b->h_alpha); b->h_alpha = nullptr;
  if (b->h_frames) (void)hipHostFree(b->h_frames); b->h_frames = nullptr;
  if (b->h_jobs) (void)hipHostFree(b->h_jobs); b->h_jobs = nullptr;
}

// allocate arena for the worst case: every image has an alpha frame when channels == 4
static int batch_alloc(mi_batch *b) {
  const DeviceTables &tab = g_tabs[b->device];
  batch_plan(b, b->channels == 4);
  std::vector<FramePlan> worst = b->frames;
  size_t total = 0, max_tiles = 0; uint32_t max_cap = 0; size_t packed = 0;
  for (auto &p : worst) {
    const uint32_t cap = tile_capacity(p);
    p.arena_bytes = carve(p, nullptr, cap);
    total += align_up(p.arena_bytes, 4096); max_tiles += (size_t)p.ntiles; max_cap = std::max(max_cap, cap);
    packed += (size_t)p.ntiles * cap;
  }
  (void)tab;
  HIP_OK(hipMalloc(&b->d_arena, total)); b->arena_bytes = total;
  HIP_OK(hipMalloc(&b->d_frames, sizeof(FrameDev) * worst.size()));
  HIP_OK(hipMalloc(&b->d_jobs, sizeof(TileJob) * max_tiles));
  b->pre_cap = max_cap;
  HIP_OK(hipMalloc(&b->d_precarry, (size_t)max_tiles * (size_t)max_cap * 2));
  { int max_np = 1; for (auto &p : worst) max_np = std::max(max_np, p.np); b->rec_cap = MI_K4_SB_RECORDS(max_np); }
  HIP_OK(hipMalloc(&b->d_recbuf, (size_t)max_tiles * 3 * (size_t)b->rec_cap * 4));
  b->aux_bytes = (size_t)max_tiles * (size_t)max_cap * 2 + (size_t)max_tiles * 3 * (size_t)b->rec_cap * 4;
  HIP_OK(hipMalloc(&b->d_offsets, max_tiles * 4));
  HIP_OK(hipMalloc(&b->d_prof, std::max<size_t>(max_tiles, 2048) * 128 * 8)); HIP_OK(hipMemset(b->d_prof, 0, std::max<size_t>(max_tiles, 2048) * 128 * 8));   // profiling builds: per tile job (K4) / per persistent workgroup (K1)
  // Packed payloads: the worst case is the sum of the tile capacities (raw size, hundreds of MB of pinned memory per batch), the
  // usual case a few per cent of it: start at 1/16 and let mi_batch_wait grow the pair when a run needs more.
  b->packed_max = std::min<size_t>(packed, (size_t)1 << 31);
  b->packed_cap = std::min(b->packed_max, align_up(std::max<size_t>(packed / 16, (size_t)1 << 20), 4096));
  HIP_OK(hipMalloc(&b->d_packed, b->packed_cap));
  HIP_OK(hipHostMalloc(&b->h_packed, b->packed_cap));
  HIP_OK(hipHostMalloc(&b->h_lens, max_tiles * 4));
  HIP_OK(hipHostMalloc(&b->h_lf, worst.size() * 13 * sizeof(int)));
  HIP_OK(hipHostMalloc(&b->h_alpha, sizeof(int) * b->cap));
  HIP_OK(hipHostMalloc(&b->h_frames, sizeof(FrameDev) * worst.size()));
  HIP_OK(hipHostMalloc(&b->h_jobs, sizeof(TileJob) * max_tiles));
  return search_reserve(b->queue, worst, b->device, b->stream);   // the tile search's work list and snapshot pool for a full batch: no (device-synchronising) reallocation inside an encode
}

extern "C" {

const char *mi_version(void) { return "mi_avif 0.1 (gfx950)"; }
int mi_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
void mi_free(void *p) { free(p); }
int mi_quality_to_quantizer(float q) { return quality_to_quantizer(q); }
int mi_av1_tweaks_from_preset(uint8_t speed, uint8_t quantizer, mi_av1_config *cfg) { if (!cfg) return MI_INVALID_ARGUMENT; return tweaks_from_preset(speed, quantizer, cfg); }
void mi_rgb_to_ycbcr(const uint8_t rgb[3], int depth, uint16_t out[3]) { rgb_to_ycbcr_host(rgb, depth, out); }

void mi_ravif_encoder_default(mi_ravif_encoder *e) {     // Encoder::new, ravif/src/av1encoder.rs:88-102
  memset(e, 0, sizeof(*e));
  e->quality = 80.f; e->alpha_quality = 80.f; e->speed = 5; e->color_model = 0; e->depth = 0; e->alpha_mode = 1; e->threads = 0; e->device = 0;
}

size_t mi_avif_serialize(const uint8_t *color, size_t color_len, const uint8_t *alpha, size_t alpha_len, uint32_t w, uint32_t h,
                         uint8_t depth, uint8_t matrix, int premultiplied, const uint8_t *exif, size_t exif_len, uint8_t **out) {
  std::vector<uint8_t> v = avif_container(color, color_len, alpha, alpha_len, w, h, depth, matrix, premultiplied != 0, exif, exif_len);
  *out = (uint8_t *)malloc(v.size()); memcpy(*out, v.data(), v.size());
  return v.size();
}

// The filters address samples inside a plane with 32-bit offsets (row * stride + column): a padded plane must stay below 2^31 samples.  That is 60 x the largest picture
// any AV1 level allows (level 6.x: 35.6 MPix); beyond it the entry points answer MI_INVALID_ARGUMENT instead of filtering the wrong samples.
static bool mi_plane_too_large(uint32_t w, uint32_t h) {
  const uint64_t pw = ((uint64_t)w + 63) & ~63ull, ph = ((uint64_t)h + 63) & ~63ull;
  return (pw + 64) * (ph + 64) >= (1ull << 31);
}
mi_batch *mi_batch_create(const mi_ravif_encoder *e, int n_images, uint32_t w, uint32_t h, int channels) {
  if (!e || n_images < 1 || w < 1 || h < 1 || w > 65536 || h > 65536 || (channels != 3 && channels != 4) || e->alpha_mode > 2) return nullptr;
  if (mi_plane_too_large(w, h)) return nullptr;
  if (e->speed < 1 || e->speed > 10 || !(e->quality >= 1.f && e->quality <= 100.f) || !(e->alpha_quality >= 1.f && e->alpha_quality <= 100.f)) return nullptr;
  if (mi_device_count() <= e->device) { fprintf(stderr, "mi_avif: no HIP device %d (the HIP path is mandatory; there is no CPU fallback)\n", e->device); return nullptr; }
  if (hipSetDevice(e->device) != hipSuccess) return nullptr;
  const bool timing = mi_timing_enabled();
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3; };
  mi_batch *b = new mi_batch();
  b->enc = *e; b->n = b->cap = n_images; b->w = w; b->h = h; b->channels = channels; b->device = e->device; b->depth = e->depth == 8 ? 8 : 10;
  if (e->exif && e->exif_len) b->exif.assign(e->exif, e->exif + e->exif_len);
  b->enc.exif = b->exif.empty() ? nullptr : b->exif.data(); b->enc.exif_len = b->exif.size();
  b->alpha_flags.assign(n_images, 0);
  b->pixel_bytes = (size_t)n_images * w * h * channels;
#ifdef MI_TUNING_KNOBS                                  // probe builds only (profiles/r05r_prio_probe.txt: no gain)
  if (getenv("MI_POSTK1_PRIORITY") && atoi(getenv("MI_POSTK1_PRIORITY")) > 0) {
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&b->stream_hi, hipStreamNonBlocking, hi) != hipSuccess || hipEventCreateWithFlags(&b->ev_hi, hipEventDisableTiming) != hipSuccess) b->stream_hi = nullptr;
  }
#endif
  bool ok = hipStreamCreate(&b->stream) == hipSuccess && hipMalloc(&b->d_pixels, b->pixel_bytes) == hipSuccess && hipHostMalloc(&b->h_pixels, b->pixel_bytes) == hipSuccess &&
            hipMalloc(&b->d_alpha_flags, sizeof(int) * n_images) == hipSuccess;
  if (ok && channels == 4 && e->alpha_mode == 1)
    ok = hipMalloc(&b->d_clean, b->pixel_bytes) == hipSuccess && hipMalloc(&b->d_clean_tmp, (size_t)w * h * 4) == hipSuccess &&
         hipMalloc(&b->d_alpha_acc, sizeof(unsigned long long) * 4 * n_images) == hipSuccess;
  if (ok && channels == 4 && e->alpha_mode == 2) ok = hipMalloc(&b->d_clean, b->pixel_bytes) == hipSuccess;   // premultiplied pixels
  for (int i = 0; i < 8 && ok; i++) ok = hipEventCreate(&b->ev[i]) == hipSuccess;
  const double t_px = since();
  if (ok) ok = batch_alloc(b) == MI_OK;
  const double t_arena = since();
  if (ok) ok = ensure_tables(e->device) == MI_OK;              // last: a caller may be warming the device up on another thread meanwhile
  if (timing) fprintf(stderr, "[mi_avif] batch_create %d x %ux%u: pixels + pinned staging %.1f ms, arena (%.2f GB) %.1f ms, tables %.1f ms\n", n_images, w, h, t_px, b->arena_bytes / 1e9, t_arena - t_px, since() - t_arena);
  if (!ok) { mi_batch_destroy(b); return nullptr; }
  b->files.resize(n_images); b->color_sz.assign(n_images, 0); b->alpha_sz.assign(n_images, 0);
  return b;
}

// The batch owns a pinned host staging area laid out like its HBM input slot; H2D always starts from there.
uint8_t *mi_batch_input(mi_batch *b, int index) {
  if (!b || index < 0 || index >= b->cap) return nullptr;
  return b->h_pixels + (size_t)index * b->w * b->h * b->channels;
}
int mi_batch_set_count(mi_batch *b, int n_images) {
  if (!b || b->in_flight || n_images < 1 || n_images > b->cap) return MI_INVALID_ARGUMENT;
  b->n = n_images; b->alpha_flags.assign(n_images, 0);
  return MI_OK;
}
// enqueue the H2D of images [first, first + count) from the pinned staging on the batch's stream; returns at once
int mi_batch_upload_async(mi_batch *b, int first, int count) {
  if (!b || first < 0 || count < 1 || first + count > b->cap) return MI_INVALID_ARGUMENT;
  (void)hipSetDevice(b->device);
  const size_t img = (size_t)b->w * b->h * b->channels;
  HIP_OK(hipMemcpyAsync(b->d_pixels + first * img, b->h_pixels + first * img, count * img, hipMemcpyHostToDevice, b->stream));
  return MI_OK;
}
int mi_batch_upload(mi_batch *b, int index, const uint8_t *pixels, size_t stride_px) {
  if (!b || index < 0 || index >= b->cap || !pixels) return MI_INVALID_ARGUMENT;
  const size_t row = (size_t)b->w * b->channels;
  uint8_t *dst = mi_batch_input(b, index);
  for (uint32_t y = 0; y < b->h; y++) memcpy(dst + y * row, pixels + (size_t)y * stride_px * b->channels, row);
  if (int st = mi_batch_upload_async(b, index, 1)) return st;
  HIP_OK(hipStreamSynchronize(b->stream));
  return MI_OK;
}

// debug/profiling: per-tile [K1 start, K1 end, K4 start, K4 end] ticks of the last encode; out must hold 4*num_tiles values
int mi_batch_tile_clocks(mi_batch *b, unsigned long long *out) {
  if (!b || !out) return MI_INVALID_ARGUMENT;
  (void)hipSetDevice(b->device);
  size_t o = 0;
  for (auto &p : b->frames) { HIP_OK(hipMemcpy(out + (size_t)p.dev.tile_base * 4, p.dev.tile_clk, (size_t)p.ntiles * 32, hipMemcpyDeviceToHost)); o += (size_t)p.ntiles * 4; }
  return MI_OK;
}
// profiling builds (MI_PROFILE=1): K1 phase cycle counters, 64 values per tile job of the last encode
int mi_batch_phase_profile(mi_batch *b, unsigned long long *out) {
  if (!b || !out || !b->d_prof) return MI_INVALID_ARGUMENT;
  (void)hipSetDevice(b->device);
  HIP_OK(hipMemcpy(out, b->d_prof, std::max<size_t>(b->jobs.size(), 2048) * 128 * 8, hipMemcpyDeviceToHost));     // rows: tile jobs (K4) or persistent workgroups (K1: up to the resident grid); unused rows are zero
  return MI_OK;
}
int mi_batch_num_tiles(const mi_batch *b) { return b ? (int)b->jobs.size() : 0; }
double mi_batch_stage_ms(const mi_batch *b, int stage) { return (b && stage >= 0 && stage < 8) ? b->stage_ms[stage] : 0.0; }

// Enqueues the GPU part of the hot path (K0..K4 + tile-length readback) on the batch's stream and returns.
int mi_batch_encode_async(mi_batch *b) {
  if (!b) return MI_INVALID_ARGUMENT;
  if (b->in_flight) return MI_INVALID_ARGUMENT;
  (void)hipSetDevice(b->device);
  const DeviceTables &tab = g_tabs[b->device];
  hipStream_t s = b->stream;
  // ---- plan colour frames and, for RGBA input, an alpha frame per image (idle on the device unless the front end flags the image)
  batch_plan(b, b->channels == 4);
  size_t off = 0;
  auto place = [&](FramePlan &p) { const uint32_t cap = tile_capacity(p); p.arena = b->d_arena + off; p.arena_bytes = carve(p, p.arena, cap); off += align_up(p.arena_bytes, 4096); fill_dev(p, tab); };
  for (auto &p : b->frames) place(p);
  // ---- K0 front end: RGBA8 -> planes (+ alpha plane into a staging slot at the end of the colour frame's fin[] planes)
  HIP_OK(hipEventRecord(b->ev[0], s));
  HIP_OK(hipMemsetAsync(b->d_alpha_flags, 0, sizeof(int) * b->n, s));
  const FrontConsts fc = front_consts(b->depth);
  FrontParams fp{ fc.sy_r, fc.sy_g, fc.sy_b, fc.scale, fc.kcb, fc.kcr, fc.shift, b->depth, b->enc.color_model, b->channels };
  const uint8_t *front_src = b->d_pixels;
  if (b->d_clean && b->enc.alpha_mode == 2) {                  // convert_alpha_8bit: Premultiplied (av1encoder.rs:282-296)
    const size_t npx = (size_t)b->n * b->w * b->h;
    hipLaunchKernelGGL(premultiply_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, b->d_pixels, b->d_clean, npx);
    HIP_OK(hipGetLastError());
    front_src = b->d_clean;
  } else if (b->d_clean) {                                     // convert_alpha_8bit: UnassociatedClean (av1encoder.rs:277-281)
    HIP_OK(hipMemsetAsync(b->d_alpha_acc, 0, sizeof(unsigned long long) * 4 * b->n, s));
    const dim3 g((b->w + 255) / 256, b->h), blk(256);
    for (int i = 0; i < b->n; i++) {
      const uint8_t *in = b->d_pixels + (size_t)i * b->w * b->h * 4; uint8_t *outp = b->d_clean + (size_t)i * b->w * b->h * 4;
      hipLaunchKernelGGL(alpha_scan_kernel, g, blk, 0, s, in, (int)b->w, (int)b->h, b->d_alpha_acc + 4 * i);
      hipLaunchKernelGGL(alpha_rewrite_kernel, g, blk, 0, s, in, b->d_clean_tmp, (int)b->w, (int)b->h, b->d_alpha_acc + 4 * i, 0);
      hipLaunchKernelGGL(alpha_rewrite_kernel, g, blk, 0, s, (const uint8_t *)b->d_clean_tmp, outp, (int)b->w, (int)b->h, b->d_alpha_acc + 4 * i, 1);
    }
    HIP_OK(hipGetLastError());
    front_src = b->d_clean;
  }
  for (int i = 0; i < b->n; i++) {
    FramePlan &p = b->frames[i];
    uint16_t *alpha_stage = b->channels == 4 ? p.dev.fin[0] : nullptr;      // fin[0] is free until CDEF runs
    hipLaunchKernelGGL(frontend_kernel, dim3((p.pw + 255) / 256, p.ph), dim3(256), 0, s,
                       front_src + (size_t)i * b->w * b->h * b->channels, (int)b->w, (int)b->h, (int)b->w, fp,
                       p.dev.src[0], p.dev.src[1], p.dev.src[2], alpha_stage, p.pw, p.ph, b->d_alpha_flags + i);
  }
  HIP_OK(hipGetLastError());
  if (b->channels == 4) {
    HIP_OK(hipMemcpyAsync(b->h_alpha, b->d_alpha_flags, sizeof(int) * b->n, hipMemcpyDeviceToHost, s));    // read in mi_batch_wait
    for (size_t k = b->n; k < b->frames.size(); k++) {
      FramePlan &a = b->frames[k], &col = b->frames[a.image];
      a.dev.active = b->d_alpha_flags + a.image;
      HIP_OK(hipMemcpyAsync(a.dev.src[0], col.dev.fin[0], (size_t)a.pw * a.ph * 2, hipMemcpyDeviceToDevice, s));
    }
  }
  // ---- tile job list, frame descriptors
  b->jobs.clear();
  int max_mi_cells = 0, max_sb = 0, max_lr = 0, max_lr_sets = 4, class_begin[6] = { 0, 0, 0, 0, 0, 0 };
  // tile jobs grouped by block-size class (one K1 instantiation per class); tile_base indexes the grouped list
  for (int cls = 2; cls <= 4; cls++) {
    class_begin[cls] = (int)b->jobs.size();
    for (size_t k = 0; k < b->frames.size(); k++) {
      FramePlan &p = b->frames[k];
      if (std::max(p.maxbs, 2) != cls) continue;
      p.dev.tile_base = (int)b->jobs.size();
      p.dev.prof_out = b->d_prof;
      for (int tr = 0; tr < p.tiles.rows; tr++) for (int tc = 0; tc < p.tiles.cols; tc++) b->jobs.push_back(TileJob{ (int)k, tr, tc });
    }
  }
  class_begin[5] = (int)b->jobs.size();
  for (size_t k = 0; k < b->frames.size(); k++) {
    FramePlan &p = b->frames[k];
    max_mi_cells = std::max(max_mi_cells, p.mi_cols * p.mi_rows * 4); max_sb = std::max(max_sb, p.sb_cols * p.sb_rows);
    if (p.cfg.lrf) { max_lr = std::max(max_lr, lr_units_host(p.cfg.width) * lr_units_host(p.cfg.height)); if (p.cfg.sgr_full) max_lr_sets = 16; }
  }
  for (size_t k = 0; k < b->frames.size(); k++) b->h_frames[k] = b->frames[k].dev;          // pinned: the copies below never block the host
  memcpy(b->h_jobs, b->jobs.data(), sizeof(TileJob) * b->jobs.size());
  HIP_OK(hipMemcpyAsync(b->d_frames, b->h_frames, sizeof(FrameDev) * b->frames.size(), hipMemcpyHostToDevice, s));
  HIP_OK(hipMemcpyAsync(b->d_jobs, b->h_jobs, sizeof(TileJob) * b->jobs.size(), hipMemcpyHostToDevice, s));
  const int njobs = (int)b->jobs.size(), nframes = (int)b->frames.size();
  // ---- activity mask (Tune::Psychovisual) -> K1 tile search -> K2a/K2 deblock (level search + filter), K3 CDEF, K5 restoration -> K4 entropy coding.
  // A two-pass encode (rdo_passes = 2) runs the chain twice: between the passes every tile's final CDFs become its rate table, the frames switch
  // over to them, and the activity kernel clears the per-encode state again; the events time the last pass.
  int max_cells = 0; for (auto &p : b->frames) max_cells = std::max(max_cells, (p.pw / 8) * (p.ph / 8));
  const int passes = b->frames[0].cfg.rdo_passes == 2 ? 2 : 1;
  const bool bottomup = b->frames[0].cfg.encode_bottomup != 0;
  for (int pass = 0; pass < passes; pass++) {
    if (pass == 1) {
      int max_tiles = 1; for (auto &p : b->frames) max_tiles = std::max(max_tiles, p.ntiles);
      hipLaunchKernelGGL(cdf_cost_kernel, dim3(max_tiles, nframes), dim3(256), 0, s, b->d_frames);
      hipLaunchKernelGGL(pass_flip_kernel, dim3((nframes + 63) / 64), dim3(64), 0, s, b->d_frames, nframes);
    }
    hipLaunchKernelGGL(activity_kernel, dim3((max_cells + 255) / 256, nframes), dim3(256), 0, s, b->d_frames);
    hipLaunchKernelGGL(segment_kernel, dim3(nframes), dim3(256), 0, s, b->d_frames);
    HIP_OK(hipEventRecord(b->ev[1], s));
    if (pass == 0) { if (int st = search_enqueue(b->queue, b->frames, b->jobs, class_begin, b->d_frames, b->d_jobs, b->device, s)) return st; }
    else if (int st = search_launch(b->queue, search_mode(b->frames), class_begin, b->d_frames, b->d_jobs, b->device, s)) return st;
    HIP_OK(hipEventRecord(b->ev[2], s));
    hipStream_t sp = s;
    if (b->stream_hi) { sp = b->stream_hi; HIP_OK(hipStreamWaitEvent(sp, b->ev[2], 0)); }
    HIP_OK(launch_loop_filters(b->d_frames, nframes, max_mi_cells, max_sb, max_lr, max_lr_sets, sp, b->ev[3]));
    HIP_OK(hipEventRecord(b->ev[4], sp));
    for (int cls = 2; cls <= 4; cls++)
      HIP_OK(launch_entropy(cls, b->d_frames, b->d_jobs + class_begin[cls], class_begin[cls + 1] - class_begin[cls], b->d_precarry + (size_t)class_begin[cls] * (size_t)b->pre_cap, b->pre_cap,
                            b->d_recbuf + (size_t)class_begin[cls] * 3 * (size_t)b->rec_cap, b->rec_cap, sp));
    if (b->stream_hi) { HIP_OK(hipEventRecord(b->ev_hi, sp)); HIP_OK(hipStreamWaitEvent(s, b->ev_hi, 0)); }
  }
  HIP_OK(hipGetLastError());
  // ---- tile lengths -> offsets -> pack -> one D2H
  HIP_OK(hipEventRecord(b->ev[5], s));
  for (size_t k = 0; k < b->frames.size(); k++) { FramePlan &p = b->frames[k]; HIP_OK(hipMemcpyAsync(b->h_lens + p.dev.tile_base, p.dev.tile_len, (size_t)p.ntiles * 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(b->h_lf + 13 * k, p.dev.lf_out, 13 * sizeof(int), hipMemcpyDeviceToHost, s)); }
  b->in_flight = true;
  return MI_OK;
}

// Waits for the enqueued work, compacts + downloads the tile payloads (one D2H) and assembles OBUs and containers.
int mi_batch_wait(mi_batch *b) {
  if (!b || !b->in_flight) return MI_INVALID_ARGUMENT;
  (void)hipSetDevice(b->device);
  hipStream_t s = b->stream;
  b->in_flight = false;
  const int njobs = (int)b->jobs.size();
  std::vector<uint32_t> offsets(njobs);
  HIP_OK(hipStreamSynchronize(s));
  if (b->channels == 4) for (int i = 0; i < b->n; i++) b->alpha_flags[i] = b->h_alpha[i];
  auto idle = [&](const FramePlan &p) { return p.is_alpha && !b->alpha_flags[p.image]; };
  size_t total = 0;
  for (int j = 0; j < njobs; j++) {
    if (b->h_lens[j] == 0xFFFFFFFFu) { fprintf(stderr, "mi_avif: tile %d overflowed its output buffer (or its frame's tile search gave up waiting for a neighbour)\n", j); return MI_ENCODING_ERROR; }
    offsets[j] = (uint32_t)total; total += b->h_lens[j];
  }
  if (total > b->packed_max) return MI_ENCODING_ERROR;
  if (total > b->packed_cap) {                                 // rare (near-lossless settings): grow the packed pair, keep it
    (void)hipFree(b->d_packed); (void)hipHostFree(b->h_packed); b->d_packed = nullptr; b->h_packed = nullptr;
    b->packed_cap = std::min(b->packed_max, align_up(total + total / 2, 4096));
    HIP_OK(hipMalloc(&b->d_packed, b->packed_cap));
    HIP_OK(hipHostMalloc(&b->h_packed, b->packed_cap));
  }
  HIP_OK(hipMemcpyAsync(b->d_offsets, offsets.data(), (size_t)njobs * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(pack_tiles_kernel, dim3(njobs), dim3(256), 0, s, b->d_frames, b->d_jobs, njobs, b->d_offsets, b->d_packed);
  HIP_OK(hipMemcpyAsync(b->h_packed, b->d_packed, total, hipMemcpyDeviceToHost, s));
  HIP_OK(hipEventRecord(b->ev[6], s));
  HIP_OK(hipStreamSynchronize(s));
  // ---- host assembly
  for (size_t k = 0; k < b->frames.size(); k++) {
    FramePlan &p = b->frames[k];
    if (idle(p)) { p.obu.clear(); continue; }
    std::vector<std::pair<const uint8_t *, size_t>> tl;
    for (int t = 0; t < p.ntiles; t++) { const int j = p.dev.tile_base + t; tl.push_back({ b->h_packed + offsets[j], (size_t)b->h_lens[j] }); }
    static_assert(offsetof(FrameHeaderInfo, seg_n) == offsetof(FrameHeaderInfo, lf_level) + 4 * sizeof(int) && offsetof(FrameHeaderInfo, seg_qidx) == offsetof(FrameHeaderInfo, seg_n) + sizeof(int),
                  "FrameDev::lf_out reports 13 ints: lf_level[4], seg_n, seg_qidx[8]");
    memcpy((char *)&p.hdr + offsetof(FrameHeaderInfo, lf_level), b->h_lf + 13 * k, 13 * sizeof(int));
    p.obu = assemble_obus(p.hdr, tl);
  }
  for (int i = 0; i < b->n; i++) {
    const FramePlan *alpha = nullptr;
    if (b->channels == 4 && b->alpha_flags[i]) alpha = &b->frames[b->n + i];
    const FramePlan &col = b->frames[i];
    b->files[i] = avif_container(col.obu.data(), col.obu.size(), alpha ? alpha->obu.data() : nullptr, alpha ? alpha->obu.size() : 0,
                                 b->w, b->h, b->depth, col.cfg.matrix, b->enc.alpha_mode == 2, b->enc.exif, b->enc.exif_len);
    b->color_sz[i] = col.obu.size(); b->alpha_sz[i] = alpha ? alpha->obu.size() : 0;
  }
  HIP_OK(hipEventRecord(b->ev[7], s));
  HIP_OK(hipEventSynchronize(b->ev[7]));
  for (int i = 0; i < 7; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]); b->stage_ms[i] = ms; }
  return MI_OK;
}

int mi_batch_encode(mi_batch *b) {
  const int st = mi_batch_encode_async(b);
  return st ? st : mi_batch_wait(b);
}

int mi_batch_get(mi_batch *b, int index, mi_encoded_image *out) {
  if (!b || !out || index < 0 || index >= b->n || b->files[index].empty()) return MI_INVALID_ARGUMENT;
  const std::vector<uint8_t> &f = b->files[index];
  out->avif_file = (uint8_t *)malloc(f.size()); memcpy(out->avif_file, f.data(), f.size());
  out->avif_len = f.size(); out->color_byte_size = b->color_sz[index]; out->alpha_byte_size = b->alpha_sz[index];
  return MI_OK;
}

int mi_batch_get_recon(mi_batch *b, int index, int alpha, uint16_t *planes[3]) {
  if (!b || index < 0 || index >= b->n) return MI_INVALID_ARGUMENT;
  (void)hipSetDevice(b->device);
  const FramePlan *p = nullptr;
  if (!alpha) p = &b->frames[index]; else if (b->channels == 4 && b->alpha_flags[index]) p = &b->frames[b->n + index];
  if (!p) return MI_INVALID_ARGUMENT;
  for (int i = 0; i < 3; i++) planes[i] = nullptr;
  for (int i = 0; i < p->np; i++) {
    planes[i] = (uint16_t *)malloc((size_t)b->w * b->h * 2);
    HIP_OK(hipMemcpy2D(planes[i], (size_t)b->w * 2, p->cfg.lrf ? p->dev.lrp[i] : p->dev.fin[i], (size_t)p->pw * 2, (size_t)b->w * 2, b->h, hipMemcpyDeviceToHost));
  }
  return MI_OK;
}

void mi_batch_destroy(mi_batch *b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  batch_free_device(b);
  if (b->d_pixels) (void)hipFree(b->d_pixels);
  if (b->h_pixels) (void)hipHostFree(b->h_pixels);
  if (b->d_alpha_flags) (void)hipFree(b->d_alpha_flags);
  if (b->d_clean) (void)hipFree(b->d_clean);
  if (b->d_clean_tmp) (void)hipFree(b->d_clean_tmp);
  if (b->d_alpha_acc) (void)hipFree(b->d_alpha_acc);
  for (int i = 0; i < 8; i++) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
  if (b->ev_hi) (void)hipEventDestroy(b->ev_hi);
  if (b->stream_hi) (void)hipStreamDestroy(b->stream_hi);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

// ---- batch objects behind the one-call entry points are pooled ----
// Creating a batch (hipMalloc of the arena, pinned staging: ~0.2 s for 32 x 1080p) and destroying it cost more than encoding a
// small image, and a caller of ravif::Encoder::encode_rgba calls it in a loop with the same settings.  mi_ravif_encode_rgba/_rgb,
// _batch and _stream therefore take their batch objects from a process-wide pool keyed by (device, capacity, shape, settings) and
// hand them back afterwards; mi_release_cached() (or process exit) frees them.  Explicit mi_batch_create objects are not pooled.
struct PoolKey {
  int device, cap, channels; uint32_t w, h; float quality, alpha_quality; uint8_t speed, color_model, depth, alpha_mode; int32_t threads, tiles_override, rdo_passes;
  bool operator==(const PoolKey &o) const {
    return device == o.device && cap == o.cap && channels == o.channels && w == o.w && h == o.h && quality == o.quality && alpha_quality == o.alpha_quality &&
           speed == o.speed && color_model == o.color_model && depth == o.depth && alpha_mode == o.alpha_mode && threads == o.threads && tiles_override == o.tiles_override && rdo_passes == o.rdo_passes;
  }
};
static PoolKey pool_key(const mi_ravif_encoder *e, int cap, uint32_t w, uint32_t h, int channels) {
  return PoolKey{ e->device, cap, channels, w, h, e->quality, e->alpha_quality, e->speed, e->color_model, e->depth, e->alpha_mode, e->threads, e->tiles_override, e->rdo_passes == 2 ? 2 : 1 };
}
static std::mutex g_pool_mu;
static std::vector<std::pair<PoolKey, mi_batch *>> g_pool;          // oldest first; never destroyed at process exit (the runtime may be gone by then)
static size_t batch_footprint(const mi_batch *b) { return b->arena_bytes + b->aux_bytes + 3 * b->pixel_bytes + b->packed_cap; }
static constexpr size_t MI_POOL_MAX_ITEMS = 8, MI_POOL_MAX_BYTES = (size_t)32 << 30;     // what the one-call entry points may keep between calls (mi_release_cached() frees it)

static mi_batch *pool_acquire(const mi_ravif_encoder *e, int cap, uint32_t w, uint32_t h, int channels) {
  const PoolKey key = pool_key(e, cap, w, h, channels);
  mi_batch *b = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++) if (g_pool[i].first == key) { b = g_pool[i].second; g_pool.erase(g_pool.begin() + i); break; }
  }
  if (!b) return mi_batch_create(e, cap, w, h, channels);
  b->exif.clear();
  if (e->exif && e->exif_len) b->exif.assign(e->exif, e->exif + e->exif_len);
  b->enc.exif = b->exif.empty() ? nullptr : b->exif.data(); b->enc.exif_len = b->exif.size();
  b->n = b->cap; b->alpha_flags.assign(b->cap, 0);
  return b;
}
static void pool_release(mi_batch *b) {
  if (!b) return;
  if (b->in_flight) { mi_batch_destroy(b); return; }
  const PoolKey key = pool_key(&b->enc, b->cap, b->w, b->h, b->channels);
  std::vector<mi_batch *> evict;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back({ key, b });
    size_t bytes = 0; for (auto &x : g_pool) bytes += batch_footprint(x.second);
    while (g_pool.size() > MI_POOL_MAX_ITEMS || (bytes > MI_POOL_MAX_BYTES && g_pool.size() > 1)) { bytes -= batch_footprint(g_pool.front().second); evict.push_back(g_pool.front().second); g_pool.erase(g_pool.begin()); }
  }
  for (mi_batch *x : evict) mi_batch_destroy(x);
}
void mi_release_cached(void) {
  std::vector<std::pair<PoolKey, mi_batch *>> all;
  { std::lock_guard<std::mutex> lk(g_pool_mu); all.swap(g_pool); }
  for (auto &x : all) mi_batch_destroy(x.second);
}

static int encode_one(const mi_ravif_encoder *e, const uint8_t *px, int channels, uint32_t w, uint32_t h, size_t stride_px, mi_encoded_image *out) {
  if (!e || !px || !out || w < 1 || h < 1) return MI_INVALID_ARGUMENT;
  mi_batch *b = pool_acquire(e, 1, w, h, channels);
  if (!b) return mi_device_count() > e->device ? MI_INVALID_ARGUMENT : MI_NO_DEVICE;
  int st = mi_batch_upload(b, 0, px, stride_px);
  if (st == MI_OK) st = mi_batch_encode(b);
  if (st == MI_OK) st = mi_batch_get(b, 0, out);
  pool_release(b);
  return st;
}
int mi_ravif_encode_rgba(const mi_ravif_encoder *e, const uint8_t *rgba, uint32_t w, uint32_t h, size_t stride_px, mi_encoded_image *out) { return encode_one(e, rgba, 4, w, h, stride_px, out); }

// PNG -> RGBA8 (cavif's load_rgba, src/main.rs:265-283); host code, no GPU involved
int mi_png_decode_rgba(const uint8_t *data, size_t len, uint8_t **rgba, uint32_t *w, uint32_t *h) {
  if (!data || !rgba || !w || !h) return MI_INVALID_ARGUMENT;
  try {                                                       // nothing may unwind through the C ABI
    std::vector<uint8_t> px;
    const int st = png_decode_rgba(data, len, px, *w, *h);
    if (st) return st;
    *rgba = (uint8_t *)malloc(px.size());
    if (!*rgba) return MI_ENCODING_ERROR;
    memcpy(*rgba, px.data(), px.size());
    return MI_OK;
  } catch (const std::exception &) { return MI_ENCODING_ERROR; }
}

// The reference's files.into_par_iter() (src/main.rs:223) over the GPUs of one node: images are independent, so a host
// thread per device pulls runs of equally-shaped images from a shared cursor and pushes each run through one resident
// batch (no collective, no cross-device traffic).  status[i] receives the per-image result; returns the first failure.
// Streaming form of the fan-out: image i is obtained through `fetch(user, i, &desc)` when a worker is about to stage it (the
// call may block until the pixels exist -- e.g. until a loader thread has decoded the file), so loading, upload, encoding and
// assembly of consecutive runs overlap.  fetch returns MI_OK or a status that becomes the image's status.
int mi_ravif_encode_stream(const mi_ravif_encoder *e, size_t n, mi_fetch_fn fetch, mi_release_fn release, void *user, mi_encoded_image *out, int *status, const int *devices, int ndev) {
  if (!e || !fetch || (n && !out)) return MI_INVALID_ARGUMENT;
  const int have = mi_device_count();
  std::vector<int> devs;
  if (devices && ndev > 0) devs.assign(devices, devices + ndev); else for (int d = 0; d < have; d++) devs.push_back(d);
  for (int d : devs) if (d < 0 || d >= have) { fprintf(stderr, "mi_avif: no HIP device %d (no CPU fallback)\n", d); return MI_NO_DEVICE; }
  if (devs.empty()) { fprintf(stderr, "mi_avif: no HIP device (no CPU fallback)\n"); return MI_NO_DEVICE; }
  std::vector<int> st(n, MI_OK);
  for (size_t i = 0; i < n; i++) { out[i].avif_file = nullptr; out[i].avif_len = out[i].color_byte_size = out[i].alpha_byte_size = 0; }
  std::atomic<size_t> cursor{ 0 };
  // A run holds at most 32 images, and no more than an even share of the job when it is small (64 files on 8 GPUs: 8 each, not 32 + 32 + nothing).
  const size_t max_run = std::min<size_t>(32, std::max<size_t>(1, (n + devs.size() - 1) / devs.size()));
  const bool timing = mi_timing_enabled();
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3; };
  // Per device: two resident batch objects (from the pool) per image shape (MI_STREAM_SLOTS_DEFAULT).  While one batch encodes, the host fills the next one's
  // pinned staging and enqueues its H2D + encode, so uploads, tile search, entropy coding and the host-side assembly of consecutive runs overlap
  // (the same rotation bench.py drives).  Memory is bounded: a batch object holds as many images of its shape as fit MI_SLOT_BYTES (one 12 MP RGBA
  // image needs ~1.5 GB: such a shape gets runs of a few images, not 32), and before a new object is made the worker gives back the objects of the
  // shapes it has not used for the longest time until the total stays under its budget (a share of the device's free memory at the start).
  constexpr size_t MI_SLOT_BYTES = (size_t)8 << 30;
  auto est_bytes = [](uint32_t w, uint32_t h, int ch, size_t images) { return images * (size_t)w * h * (ch == 4 ? 110 : 82) + ((size_t)32 << 20); };   // arena + records + staging per pixel (measured on the planner)
  auto worker = [&](int dev) {
    mi_ravif_encoder enc = *e; enc.device = dev;
    std::future<int> warm = std::async(std::launch::async, [dev]() { return hipSetDevice(dev) == hipSuccess ? ensure_tables(dev) : (int)MI_ENCODING_ERROR; });
    size_t budget = (size_t)48 << 30;
    { size_t fr = 0, tot = 0; if (hipSetDevice(dev) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess && fr) budget = fr / 10 * 6; }
    { int sharing = 0; for (int d2 : devs) sharing += d2 == dev; budget /= (size_t)std::max(1, sharing); }      // workers on the same ordinal (devices = [0, 0]) split what is free
    static constexpr int NSLOT_MAX = 4;
    int NSLOT = MI_STREAM_SLOTS_DEFAULT;
#ifdef MI_TUNING_KNOBS
    if (const char *v = getenv("MI_STREAM_SLOTS")) NSLOT = std::min(NSLOT_MAX, std::max(1, atoi(v)));
#endif
    struct Slot { mi_batch *b = nullptr; std::future<mi_batch *> making; std::vector<size_t> idx; bool busy = false; size_t bytes = 0; };
    struct Shape { uint32_t w, h; int ch; size_t cap; Slot slot[NSLOT_MAX]; int next = 0; size_t runs = 0, last_use = 0; };
    std::vector<std::unique_ptr<Shape>> shapes;
    size_t live_bytes = 0, tick = 0;
    auto collect = [&](Slot &sl) {
      if (!sl.busy) return;
      const int rc = mi_batch_wait(sl.b);
      if (timing) fprintf(stderr, "[mi_avif %8.1f ms] dev %d: run of %zu done (on the device: front end %.1f, tile search %.1f, deblock %.1f, cdef + restoration %.1f, entropy %.1f, pack + D2H %.1f ms)\n", since(), dev, sl.idx.size(),
                          mi_batch_stage_ms(sl.b, 0), mi_batch_stage_ms(sl.b, 1), mi_batch_stage_ms(sl.b, 2), mi_batch_stage_ms(sl.b, 3), mi_batch_stage_ms(sl.b, 4), mi_batch_stage_ms(sl.b, 5));
      for (size_t k = 0; k < sl.idx.size(); k++) st[sl.idx[k]] = rc == MI_OK ? mi_batch_get(sl.b, (int)k, &out[sl.idx[k]]) : rc;
      sl.busy = false;
    };
    auto drop = [&](Slot &sl) {                                 // finish the slot's run and give its object back to the device
      collect(sl);
      if (sl.making.valid()) sl.b = sl.making.get();
      if (sl.b) mi_batch_destroy(sl.b);
      sl.b = nullptr; live_bytes -= std::min(live_bytes, sl.bytes); sl.bytes = 0;
    };
    auto make_room = [&](Shape *keep, size_t need) {
      while (live_bytes + need > budget) {
        Shape *victim = nullptr;
        for (auto &c : shapes) if (c.get() != keep && (!victim || c->last_use < victim->last_use)) { bool any = false; for (Slot &sl : c->slot) any |= sl.b || sl.making.valid(); if (any) victim = c.get(); }
        if (!victim) break;
        for (Slot &sl : victim->slot) drop(sl);
      }
    };
    auto ensure_slot = [&](Shape *sh, int j) {
      Slot &sl = sh->slot[j];
      if (sl.b || sl.making.valid()) return;
      const size_t need = est_bytes(sh->w, sh->h, sh->ch, sh->cap);
      make_room(sh, need);
      sl.bytes = need; live_bytes += need;
      const mi_ravif_encoder ec = enc; const uint32_t w = sh->w, h = sh->h; const int ch = sh->ch; const int c = (int)sh->cap;
      sl.making = std::async(std::launch::async, [ec, c, w, h, ch]() { return pool_acquire(&ec, c, w, h, ch); });
    };
    auto shape_for = [&](const mi_image_desc &x) {
      for (auto &c : shapes) if (c->w == x.width && c->h == x.height && c->ch == x.channels) return c.get();
      const size_t per = est_bytes(x.width, x.height, x.channels, 1);
      const size_t cap = std::max<size_t>(1, std::min(max_run, MI_SLOT_BYTES / per));
      shapes.emplace_back(new Shape{ x.width, x.height, x.channels, cap, {}, 0, 0, 0 });
      return shapes.back().get();
    };
    // one run (<= the shape's capacity) through the shape's next slot
    auto submit = [&](Shape *sh, const std::vector<mi_image_desc> &d, const std::vector<size_t> &run, size_t i0, bool more) {
      const mi_image_desc &d0 = d[run[0]];
      sh->last_use = ++tick; sh->runs++;
      const int j = sh->next; sh->next = (j + 1) % NSLOT;
      Slot &sl = sh->slot[j];
      collect(sl);                                           // the slot's previous run, if any
      ensure_slot(sh, j);
      if (sl.making.valid()) sl.b = sl.making.get();
      if (!sl.b) { live_bytes -= std::min(live_bytes, sl.bytes); sl.bytes = 0; }      // the object could not be made: nothing of it is resident
      int rc = sl.b ? mi_batch_set_count(sl.b, (int)run.size()) : MI_ENCODING_ERROR;
      if (timing) fprintf(stderr, "[mi_avif %8.1f ms] dev %d: slot %d ready\n", since(), dev, j);
      if (rc == MI_OK) {
        const size_t row = (size_t)d0.width * d0.channels;
        for (size_t k = 0; k < run.size(); k++) {
          const mi_image_desc &x = d[run[k]];
          uint8_t *dst = mi_batch_input(sl.b, (int)k);
          const size_t sp = x.stride_px ? x.stride_px : x.width;
          if (sp == x.width) memcpy(dst, x.pixels, row * d0.height);
          else for (uint32_t y = 0; y < d0.height; y++) memcpy(dst + y * row, x.pixels + (size_t)y * sp * d0.channels, row);
        }
        if (timing) fprintf(stderr, "[mi_avif %8.1f ms] dev %d: slot %d staged\n", since(), dev, j);
        rc = mi_batch_upload_async(sl.b, 0, (int)run.size());
        if (timing) fprintf(stderr, "[mi_avif %8.1f ms] dev %d: slot %d upload enqueued\n", since(), dev, j);
      }
      if (release) for (size_t k : run) release(user, i0 + k);   // staged (or failed): the caller's pixels are no longer read
      if (rc == MI_OK) rc = mi_batch_encode_async(sl.b);
      if (timing) fprintf(stderr, "[mi_avif %8.1f ms] dev %d: run of %zu enqueued on slot %d\n", since(), dev, run.size(), j);
      if (rc != MI_OK) { for (size_t k : run) st[i0 + k] = rc; return; }
      sl.idx.clear(); for (size_t k : run) sl.idx.push_back(i0 + k);
      sl.busy = true;
      // a shape that keeps coming gets its next slot made while the GPU works on this run (not earlier: hipMalloc / hipHostMalloc on another
      // thread hold runtime locks that stall this thread's copies and launches; not for a shape seen once: a directory of differently sized files)
      if (more && sh->runs >= 2) ensure_slot(sh, sh->next);
    };
    size_t claims = 0;
    for (;;) {
      // every claim is a full run, the first one included (MI_STREAM_FIRST_RUN_NUM / _DEN)
      size_t first_run = std::max<size_t>(1, max_run * MI_STREAM_FIRST_RUN_NUM / MI_STREAM_FIRST_RUN_DEN);
#ifdef MI_TUNING_KNOBS
      if (const char *v = getenv("MI_STREAM_FIRST_RUN")) first_run = std::min(max_run, (size_t)std::max(1, atoi(v)));
#endif
      const size_t want = claims++ == 0 ? first_run : max_run;
      const size_t i0 = cursor.fetch_add(want);                // claim the index range [i0, i1)
      if (i0 >= n) break;
      const size_t i1 = std::min(n, i0 + want);
      std::vector<mi_image_desc> d(i1 - i0);
      // images are staged in arrival order: a run is handed over as soon as the next image has another shape or the run is full
      std::vector<size_t> run; Shape *run_shape = nullptr;
      auto flush = [&](bool more) { if (!run.empty()) { submit(run_shape, d, run, i0, more); run.clear(); } };
      for (size_t i = i0; i < i1; i++) {
        const int rc = fetch(user, i, &d[i - i0]);
        const mi_image_desc &x = d[i - i0];
        if (rc != MI_OK) { st[i] = rc; continue; }
        if (!x.pixels || !x.width || !x.height || (x.channels != 3 && x.channels != 4)) { st[i] = MI_INVALID_ARGUMENT; if (release) release(user, i); continue; }
        Shape *sh = shape_for(x);
        if (sh != run_shape || run.size() >= sh->cap) flush(true);
        run_shape = sh; run.push_back(i - i0);
        if (run.size() == 1) ensure_slot(sh, sh->next);        // made in the background while the rest of the run arrives
      }
      if (timing) fprintf(stderr, "[mi_avif %8.1f ms] dev %d: images %zu..%zu fetched\n", since(), dev, i0, i1);
      flush(cursor.load() < n);
    }
    for (auto &c : shapes) for (Slot &sl : c->slot) {
      collect(sl);
      if (sl.making.valid()) sl.b = sl.making.get();
      pool_release(sl.b);                                      // back to the pool: the next call (or nobody, at process exit) gets them
      sl.b = nullptr;
    }
    warm.get();
    if (timing) fprintf(stderr, "[mi_avif %8.1f ms] dev %d: worker done\n", since(), dev);
  };
  std::vector<std::thread> th;
  for (int d : devs) th.emplace_back(worker, d);
  for (auto &t : th) t.join();
  int first = MI_OK;
  for (size_t i = 0; i < n; i++) { if (status) status[i] = st[i]; if (first == MI_OK && st[i] != MI_OK) first = st[i]; }
  return first;
}
// The reference's files.into_par_iter() (src/main.rs:223) over the GPUs of one node with every image already in host memory.
static int fetch_from_array(void *user, size_t i, mi_image_desc *d) { *d = ((const mi_image_desc *)user)[i]; return MI_OK; }
int mi_ravif_encode_batch(const mi_ravif_encoder *e, size_t n, const mi_image_desc *in, mi_encoded_image *out, int *status, const int *devices, int ndev) {
  if (!e || (n && (!in || !out))) return MI_INVALID_ARGUMENT;
  return mi_ravif_encode_stream(e, n, fetch_from_array, nullptr, (void *)in, out, status, devices, ndev);
}
int mi_ravif_encode_rgb(const mi_ravif_encoder *e, const uint8_t *rgb, uint32_t w, uint32_t h, size_t stride_px, mi_encoded_image *out) { return encode_one(e, rgb, 3, w, h, stride_px, out); }

// level 1: caller-supplied planes (encode_to_av1). Planes go straight into the frame's src[] (edge-replicated on the host).
int mi_av1_encode_planes(const mi_av1_config *cfg, const void *const planes[3], const size_t stride_bytes[3], uint8_t **out_obu, size_t *out_len, uint16_t *recon[3]) {
  if (!cfg || !planes || !planes[0] || !stride_bytes || !out_obu || !out_len || (cfg->bit_depth != 8 && cfg->bit_depth != 10)) return MI_INVALID_ARGUMENT;
  // rav1e rejects these with InvalidWidth / InvalidHeight (the sequence header carries at most 16 bits per dimension)
  if (cfg->width < 1 || cfg->height < 1 || cfg->width > 65536 || cfg->height > 65536 || mi_plane_too_large(cfg->width, cfg->height)) return MI_INVALID_ARGUMENT;
  auto pow2_4_64 = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
  if (!pow2_4_64(cfg->part_min) || !pow2_4_64(cfg->part_max) || cfg->part_min > cfg->part_max || cfg->chroma > 1) return MI_INVALID_ARGUMENT;
  if (mi_device_count() <= cfg->device) { fprintf(stderr, "mi_avif: no HIP device %d (no CPU fallback)\n", cfg->device); return MI_NO_DEVICE; }
  HIP_OK(hipSetDevice(cfg->device));
  if (int st = ensure_tables(cfg->device)) return st;
  const DeviceTables &tab = g_tabs[cfg->device];
  FramePlan p; p.cfg = *cfg; plan_geometry(p);
  for (int i = 0; i < p.np; i++) if (!planes[i]) return MI_TOO_FEW_PIXELS;
  const uint32_t cap = tile_capacity(p);
  p.arena_bytes = carve(p, nullptr, cap);
  // every device allocation and the stream are owned by this guard: all return paths release them
  struct Guard {
    hipStream_t s = nullptr; uint8_t *arena = nullptr; FrameDev *d_frame = nullptr; TileJob *d_jobs = nullptr; uint16_t *d_pre = nullptr; uint32_t *d_rec = nullptr; SearchQueue queue;
    ~Guard() { queue.free_device(); if (d_frame) (void)hipFree(d_frame); if (d_jobs) (void)hipFree(d_jobs); if (d_pre) (void)hipFree(d_pre); if (d_rec) (void)hipFree(d_rec); if (arena) (void)hipFree(arena); if (s) (void)hipStreamDestroy(s); }
  } g;
  HIP_OK(hipStreamCreate(&g.s));
  hipStream_t s = g.s;
  HIP_OK(hipMalloc(&g.arena, p.arena_bytes));
  carve(p, g.arena, cap); fill_dev(p, tab); p.dev.tile_base = 0;
  const size_t npx = (size_t)p.pw * p.ph;
  std::vector<uint16_t> host(npx);
  for (int i = 0; i < p.np; i++) {
    for (int y = 0; y < p.ph; y++) {
      const uint8_t *row = (const uint8_t *)planes[i] + (size_t)std::min<int>(y, cfg->height - 1) * stride_bytes[i];
      for (int x = 0; x < p.pw; x++) { const int sx = std::min<int>(x, cfg->width - 1); host[(size_t)y * p.pw + x] = cfg->bit_depth == 8 ? row[sx] : ((const uint16_t *)row)[sx]; }
    }
    HIP_OK(hipMemcpy(p.dev.src[i], host.data(), npx * 2, hipMemcpyHostToDevice));
  }
  std::vector<TileJob> jobs;
  for (int tr = 0; tr < p.tiles.rows; tr++) for (int tc = 0; tc < p.tiles.cols; tc++) jobs.push_back(TileJob{ 0, tr, tc });
  HIP_OK(hipMalloc(&g.d_frame, sizeof(FrameDev))); HIP_OK(hipMalloc(&g.d_jobs, sizeof(TileJob) * jobs.size())); HIP_OK(hipMalloc(&g.d_pre, (size_t)jobs.size() * (size_t)cap * 2));
  const uint32_t rec_cap = MI_K4_SB_RECORDS(p.np);
  HIP_OK(hipMalloc(&g.d_rec, (size_t)jobs.size() * 3 * (size_t)rec_cap * 4));
  FrameDev *d_frame = g.d_frame; TileJob *d_jobs = g.d_jobs; uint16_t *d_pre = g.d_pre;
  HIP_OK(hipMemcpyAsync(d_frame, &p.dev, sizeof(FrameDev), hipMemcpyHostToDevice, s));
  HIP_OK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(TileJob) * jobs.size(), hipMemcpyHostToDevice, s));
  const int njobs = (int)jobs.size();
  {
    int class_begin[6] = { 0, 0, 0, 0, 0, 0 };
    for (int cls = std::max(p.maxbs, 2) + 1; cls <= 5; cls++) class_begin[cls] = njobs;
    std::vector<FramePlan> one(1, p);
    for (int pass = 0; pass < (p.cfg.rdo_passes == 2 ? 2 : 1); pass++) {
      if (pass == 1) {                                          // two-pass pricing: the tiles' final CDFs become their rate tables
        hipLaunchKernelGGL(cdf_cost_kernel, dim3(njobs, 1), dim3(256), 0, s, d_frame);
        hipLaunchKernelGGL(pass_flip_kernel, dim3(1), dim3(64), 0, s, d_frame, 1);
      }
      hipLaunchKernelGGL(activity_kernel, dim3(((p.pw / 8) * (p.ph / 8) + 255) / 256, 1), dim3(256), 0, s, d_frame);
      hipLaunchKernelGGL(segment_kernel, dim3(1), dim3(256), 0, s, d_frame);
      if (pass == 0) { if (int st = search_enqueue(g.queue, one, jobs, class_begin, d_frame, d_jobs, cfg->device, s)) return st; }
      else if (int st = search_launch(g.queue, search_mode(one), class_begin, d_frame, d_jobs, cfg->device, s)) return st;
      HIP_OK(launch_loop_filters(d_frame, 1, p.mi_cols * p.mi_rows * 4, p.sb_cols * p.sb_rows, p.cfg.lrf ? lr_units_host(p.cfg.width) * lr_units_host(p.cfg.height) : 0, p.cfg.sgr_full ? 16 : 4, s, nullptr));
      HIP_OK(launch_entropy(p.maxbs, d_frame, d_jobs, njobs, d_pre, cap, g.d_rec, rec_cap, s));
    }
  }
  HIP_OK(hipGetLastError());
  std::vector<uint32_t> lens(njobs);
  HIP_OK(hipMemcpyAsync(lens.data(), p.dev.tile_len, (size_t)njobs * 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(p.hdr.lf_level, p.dev.lf_out, 13 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  std::vector<std::vector<uint8_t>> td(njobs); std::vector<std::pair<const uint8_t *, size_t>> tl;
  for (int j = 0; j < njobs; j++) {
    if (lens[j] == 0xFFFFFFFFu) { fprintf(stderr, "mi_avif: tile %d overflowed its output buffer (or its frame's tile search gave up waiting for a neighbour)\n", j); return MI_ENCODING_ERROR; }
    td[j].resize(lens[j]);
    HIP_OK(hipMemcpy(td[j].data(), p.dev.tile_out + (size_t)j * cap, lens[j], hipMemcpyDeviceToHost));
    tl.push_back({ td[j].data(), td[j].size() });
  }
  std::vector<uint8_t> obu = assemble_obus(p.hdr, tl);
  uint16_t *rec_out[3] = { nullptr, nullptr, nullptr };
  if (recon) for (int i = 0; i < p.np; i++) {
    rec_out[i] = (uint16_t *)malloc((size_t)cfg->width * cfg->height * 2);
    const hipError_t e = rec_out[i] ? hipMemcpy2D(rec_out[i], (size_t)cfg->width * 2, p.cfg.lrf ? p.dev.lrp[i] : p.dev.fin[i], (size_t)p.pw * 2, (size_t)cfg->width * 2, cfg->height, hipMemcpyDeviceToHost) : hipErrorOutOfMemory;
    if (e != hipSuccess) { for (int k = 0; k <= i; k++) free(rec_out[k]); return MI_ENCODING_ERROR; }
  }
  *out_obu = (uint8_t *)malloc(obu.size());
  if (!*out_obu) { for (int k = 0; k < 3; k++) free(rec_out[k]); return MI_ENCODING_ERROR; }
  memcpy(*out_obu, obu.data(), obu.size()); *out_len = obu.size();
  if (recon) for (int i = 0; i < 3; i++) recon[i] = rec_out[i];
  return MI_OK;
}

static int raw_planes(const mi_ravif_encoder *e, uint32_t w, uint32_t h, const void *yuv, const void *alpha, int depth, uint8_t range, uint8_t matrix, mi_encoded_image *out) {
  if (!e || !yuv || !out || w < 1 || h < 1) return MI_INVALID_ARGUMENT;
  const size_t n = (size_t)w * h, bps = depth == 8 ? 1 : 2;
  std::vector<uint8_t> pl[3]; for (auto &v : pl) v.resize(n * bps);
  for (size_t i = 0; i < n; i++) for (int c = 0; c < 3; c++) {
    if (depth == 8) pl[c][i] = ((const uint8_t *)yuv)[i * 3 + c]; else ((uint16_t *)pl[c].data())[i] = ((const uint16_t *)yuv)[i * 3 + c];
  }
  mi_av1_config c{}; c.width = w; c.height = h; c.bit_depth = (uint8_t)depth; c.quantizer = (uint8_t)quality_to_quantizer(e->quality);
  c.chroma = 0; c.pixel_range = range; c.threads = e->threads; c.has_color_desc = 1; c.primaries = 1; c.transfer = 13; c.matrix = matrix; c.device = e->device; c.tiles_override = e->tiles_override; c.rdo_passes = (uint8_t)(e->rdo_passes == 2 ? 2 : 1);
  if (int st = tweaks_from_preset(e->speed, c.quantizer, &c)) return st;
  const void *pp[3] = { pl[0].data(), pl[1].data(), pl[2].data() }; const size_t sb[3] = { w * bps, w * bps, w * bps };
  uint8_t *cobu = nullptr, *aobu = nullptr; size_t clen = 0, alen = 0;
  if (int st = mi_av1_encode_planes(&c, pp, sb, &cobu, &clen, nullptr)) return st;
  if (alpha) {
    mi_av1_config a = c; a.quantizer = (uint8_t)quality_to_quantizer(e->alpha_quality); a.chroma = 1; a.pixel_range = 1; a.has_color_desc = 0;
    tweaks_from_preset(e->speed, a.quantizer, &a);
    const void *ap[3] = { alpha, nullptr, nullptr };
    if (int st = mi_av1_encode_planes(&a, ap, sb, &aobu, &alen, nullptr)) { free(cobu); return st; }
  }
  out->avif_len = mi_avif_serialize(cobu, clen, aobu, alen, w, h, (uint8_t)depth, matrix, e->alpha_mode == 2, e->exif, e->exif_len, &out->avif_file);
  out->color_byte_size = clen; out->alpha_byte_size = alen;
  free(cobu); free(aobu);
  return MI_OK;
}
int mi_ravif_encode_raw_planes_8(const mi_ravif_encoder *e, uint32_t w, uint32_t h, const uint8_t *yuv, const uint8_t *alpha, uint8_t range, uint8_t matrix, mi_encoded_image *out) { return raw_planes(e, w, h, yuv, alpha, 8, range, matrix, out); }
int mi_ravif_encode_raw_planes_10(const mi_ravif_encoder *e, uint32_t w, uint32_t h, const uint16_t *yuv, const uint16_t *alpha, uint8_t range, uint8_t matrix, mi_encoded_image *out) { return raw_planes(e, w, h, yuv, alpha, 10, range, matrix, out); }

}  // extern "C"

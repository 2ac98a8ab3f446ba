// host_av1.h -- host-side (C++) frame planning and bitstream assembly for the MI355X AV1 intra path:
// ravif's in-tree configuration logic (quality -> quantizer, SpeedTweaks, tile target), rav1e's constant-Q
// key-frame quantiser rule (recalled), static rate tables, OBU headers, AVIF container.
// Everything here is cheap scalar work per image; the pixel work is in the HIP kernels.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>
#include "../../include/mi_avif.h"
#include "dev_common.h"

namespace mi {

// ---------------------------------------------------------------- ravif-level configuration
inline int quality_to_quantizer(float quality) {            // ravif/src/av1encoder.rs:526-530
  const float q = quality / 100.f;
  const float x = q >= 0.82f ? (1.f - q) * 2.6f : (q > 0.25f ? std::fmaf(q, -0.5f, 1.f - 0.125f) : 1.f - q);
  const float r = std::round(x * 255.f);
  return r <= 0.f ? 0 : (r >= 255.f ? 255 : (int)r);
}

inline int tweaks_from_preset(int speed, int quantizer, mi_av1_config *c) {   // :554-606
  if (speed < 1 || speed > 10) return MI_INVALID_ARGUMENT;
  const bool low_quality = quantizer < quality_to_quantizer(55.f);
  const bool high_quality = quantizer > quality_to_quantizer(80.f);
  const int max_block = high_quality ? 16 : 64;
  struct R { int lo, hi; } range;
  if (speed == 1 && low_quality) range = { 4, std::min(64, max_block) };
  else if (speed == 2 && low_quality) range = { 4, std::min(32, max_block) };
  else if (speed <= 4) range = { 4, 16 };
  else if (speed <= 8) range = { 8, 16 };
  else range = { 16, 16 };
  c->speed = (uint8_t)speed;
  c->part_min = (uint8_t)range.lo; c->part_max = (uint8_t)range.hi;
  c->complex_pred_modes = speed <= 1; c->sgr_full = speed <= 2; c->encode_bottomup = speed <= 2;
  c->rdo_tx_decision = speed <= 4 && !high_quality; c->reduced_tx_set = speed == 4 || speed >= 9;
  c->fine_directional_intra = speed <= 6; c->fast_deblock = speed >= 7 && !high_quality;
  c->lrf = low_quality && speed <= 8; c->cdef = low_quality && speed <= 9;
  c->inter_tx_split = speed >= 9; c->tx_domain_rate = speed >= 10; c->tx_domain_distortion = -1;
  const int base = speed == 1 ? 2048 : speed == 2 ? 1024 : speed == 3 ? 512 : speed == 4 ? 256 : 128;
  c->min_tile_size = (uint16_t)(base * (high_quality ? 2 : 1));
  return MI_OK;
}

struct FrontConsts { float sy_r, sy_g, sy_b, scale, kcb, kcr, shift; };
inline FrontConsts front_consts(int depth) {                 // constants of rgb_to_ycbcr, :504-511
  const float m[3] = { 0.2990f, 0.5870f, 0.1140f };
  const float max_value = (float)((1 << depth) - 1);
  FrontConsts k;
  k.scale = max_value / 255.f; k.shift = std::round(max_value * 0.5f);
  k.sy_r = k.scale * m[0]; k.sy_g = k.scale * m[1]; k.sy_b = k.scale * m[2];
  k.kcb = 0.5f / (1.f - m[2]); k.kcr = 0.5f / (1.f - m[0]);
  return k;
}
inline void rgb_to_ycbcr_host(const uint8_t rgb[3], int depth, uint16_t out[3]) {
  const FrontConsts k = front_consts(depth);
  const float r = rgb[0], g = rgb[1], b = rgb[2];
  const float y = std::fmaf(k.sy_b, b, std::fmaf(k.sy_r, r, k.sy_g * g));
  const float cb = std::fmaf(std::fmaf(b, k.scale, -y), k.kcb, k.shift);
  const float cr = std::fmaf(std::fmaf(r, k.scale, -y), k.kcr, k.shift);
  const float v[3] = { std::round(y), std::round(cb), std::round(cr) };
  const float sat = depth == 8 ? 255.f : 65535.f;
  for (int i = 0; i < 3; i++) out[i] = (uint16_t)(v[i] < 0.f ? 0.f : (v[i] > sat ? sat : v[i]));
}

// ---------------------------------------------------------------- quantiser / lambda
struct QuantSel { int base_q_idx, qctx, dc_qi[3], ac_qi[3], dc_q[3], ac_q[3]; long long rdmult, wq[3]; };

inline const int16_t *qlookup(int bd, bool dc) { return bd == 8 ? (dc ? av1_dc_q8 : av1_ac_q8) : (dc ? av1_dc_q10 : av1_ac_q10); }
inline int nearest_qi(long q, const int16_t *t) {            // nearest entry in the log domain (rav1e select_qi)
  if (q < t[0]) return 0;
  if (q >= t[255]) return 255;
  const int16_t *hi = std::upper_bound(t, t + 256, (int16_t)q);   // first entry > q
  const int lo_i = (int)(hi - t) - 1;
  if (t[lo_i] == q) return (int)(std::lower_bound(t, t + 256, (int16_t)q) - t);
  return (q * q < (long)t[lo_i] * t[lo_i + 1]) ? lo_i : lo_i + 1;
}
// [UPSTREAM-RECALL] rav1e rate.rs: constant-Q key frame (bitrate 0) -> QuantizerParameters::new_from_log_q
inline QuantSel select_quantizers(int quantizer, int bd, int np) {
  QuantSel s{};
  const int16_t *ac = qlookup(bd, false), *dc = qlookup(bd, true);
  const double norm = 3.0 + (bd - 8);
  const long ac_quant = ac[quantizer];
  const int dc_qi0 = nearest_qi(ac_quant, dc);
  const double log_base = 0.5 * ((std::log2((double)ac_quant) - norm) + (std::log2((double)dc[dc_qi0]) - norm));
  const double log_q = log_base - (33810170.0 / 86043287.0);
  const double x = log_q > 0 ? log_q : 0;
  const double y = np == 1 ? 0.0 : x * (1.0 / 16 + 1.0 / 32 + 1.0 / 256);
  const double off[3] = { 0.0, std::log2(7.0 / 4.0) - y, std::log2(5.0 / 4.0) - y };
  long qy = 0;
  for (int p = 0; p < np; p++) {
    const long q = std::lround(std::exp2(log_q + off[p] + norm));
    if (p == 0) qy = q;
    int aqi = nearest_qi(q, ac); const int dqi = nearest_qi(q, dc);
    if (p == 0) { aqi = std::max(aqi, 1); s.base_q_idx = aqi; }
    const int lo = std::max(1, s.base_q_idx - 63), hi = std::min(255, s.base_q_idx + 63);
    s.ac_qi[p] = p == 0 ? aqi : std::clamp(aqi, lo, hi);
    s.dc_qi[p] = std::clamp(dqi, lo, hi);
    s.ac_q[p] = ac[s.ac_qi[p]]; s.dc_q[p] = dc[s.dc_qi[p]];
  }
  s.rdmult = ((long long)qy * qy * 242273) >> 20;               // ln2/6 * (q/8)^2, 1/128 SSE per 1/512 bit
  for (int p = 0; p < np; p++) s.wq[p] = (((long long)s.ac_q[0] * s.ac_q[0]) << 12) / ((long long)s.ac_q[p] * s.ac_q[p]);
  s.qctx = s.base_q_idx <= 20 ? 0 : (s.base_q_idx <= 60 ? 1 : (s.base_q_idx <= 120 ? 2 : 3));
  return s;
}

// ---------------------------------------------------------------- tiles (spec 5.9.15; target rule :665-668)
struct Tiling { int cols_log2, rows_log2, cols, rows; std::vector<int> col_start, row_start; };
inline int tile_log2(int blk, int target) { int k = 0; while ((blk << k) < target) k++; return k; }
inline Tiling plan_tiles(int width, int height, int sb_cols, int sb_rows, int min_tile_size, int threads, int override_target) {
  long target = ((long)width * height) / ((long)std::max(min_tile_size, 1) * std::max(min_tile_size, 1));
  if (threads > 0) target = std::min<long>(target, threads);
  if (override_target > 0) target = override_target;
  const int min_cols = tile_log2(64, sb_cols), max_cols = tile_log2(1, std::min(sb_cols, 64)), max_rows = tile_log2(1, std::min(sb_rows, 64));
  const int min_tiles = std::max(min_cols, tile_log2(2304, sb_rows * sb_cols));
  int cl = min_cols, rl = std::max(min_tiles - cl, 0);
  auto dims = [&](int &tw, int &th) { tw = (sb_cols + (1 << cl) - 1) >> cl; th = (sb_rows + (1 << rl) - 1) >> rl; };
  for (;;) {
    int tw, th; dims(tw, th);
    const long count = (long)((sb_cols + tw - 1) / tw) * ((sb_rows + th - 1) / th);
    if (count >= target || (cl >= max_cols && rl >= max_rows)) break;
    if ((th >= tw && rl < max_rows) || cl >= max_cols) rl++; else cl++;
  }
  Tiling t; t.cols_log2 = cl; t.rows_log2 = rl;
  int tw, th; dims(tw, th);
  for (int s = 0; s < sb_cols; s += tw) t.col_start.push_back(s);
  t.cols = (int)t.col_start.size(); t.col_start.push_back(sb_cols);
  for (int s = 0; s < sb_rows; s += th) t.row_start.push_back(s);
  t.rows = (int)t.row_start.size(); t.row_start.push_back(sb_rows);
  return t;
}

// ---------------------------------------------------------------- static rate table
inline uint32_t neg_log2_q9(uint32_t p) {                      // (15 - log2 p) * 512, integer only
  p = std::max(p, 1u);
  const int msb = 31 - __builtin_clz(p);
  uint64_t x = (uint64_t)p << (31 - msb);
  uint32_t frac = 0;
  for (int i = 0; i < 9; i++) { x = (x * x) >> 31; frac <<= 1; if (x >> 32) { frac |= 1; x >>= 1; } }
  return (uint32_t)(15 * 512 - (msb * 512 + (int)frac));
}
// the rows of the CDF tables that carry symbol probabilities: X(offset, stride, rows, alphabet size).  One list for the host table (initial CDFs)
// and the device kernel that prices a tile's final CDFs (cdf_cost_kernel, two-pass pricing).
#define MI_COST_ROWS(X) \
  X(CDF_KF_Y, CDF_KF_Y_STRIDE, 25, 13) X(CDF_ANGLE, CDF_ANGLE_STRIDE, 8, 7) X(CDF_UV_NOCFL, CDF_UV_NOCFL_STRIDE, 13, 13) X(CDF_UV_CFL, CDF_UV_CFL_STRIDE, 13, 14) \
  X(CDF_PARTITION, CDF_PARTITION_STRIDE, 4, 4) X(CDF_PARTITION + 4 * CDF_PARTITION_STRIDE, CDF_PARTITION_STRIDE, 12, 10) X(CDF_PARTITION + 16 * CDF_PARTITION_STRIDE, CDF_PARTITION_STRIDE, 4, 8) \
  X(CDF_SKIP, CDF_SKIP_STRIDE, 3, 2) X(CDF_SEG_ID, CDF_SEG_ID_STRIDE, 3, 8) X(CDF_INTRA_TX1, CDF_INTRA_TX1_STRIDE, 26, 7) X(CDF_INTRA_TX2, CDF_INTRA_TX2_STRIDE, 39, 5) \
  X(CDF_CFL_SIGN, CDF_CFL_SIGN_STRIDE, 1, 8) X(CDF_CFL_ALPHA, CDF_CFL_ALPHA_STRIDE, 6, 16) X(CDF_TX_SIZE, CDF_TX_SIZE_STRIDE, 3, 2) X(CDF_TX_SIZE + 3 * CDF_TX_SIZE_STRIDE, CDF_TX_SIZE_STRIDE, 9, 3) \
  X(CDF_TXB_SKIP, CDF_TXB_SKIP_STRIDE, 65, 2) X(CDF_EOB_EXTRA, CDF_EOB_EXTRA_STRIDE, 90, 2) X(CDF_DC_SIGN, CDF_DC_SIGN_STRIDE, 6, 2) \
  X(CDF_COEFF_BR, CDF_COEFF_BR_STRIDE, 210, 4) X(CDF_COEFF_BASE, CDF_COEFF_BASE_STRIDE, 420, 4) X(CDF_COEFF_BASE_EOB, CDF_COEFF_BASE_EOB_STRIDE, 40, 3) \
  X(CDF_EOB_PT_16, CDF_EOB_PT_16_STRIDE, 4, 5) X(CDF_EOB_PT_32, CDF_EOB_PT_32_STRIDE, 4, 6) X(CDF_EOB_PT_64, CDF_EOB_PT_64_STRIDE, 4, 7) \
  X(CDF_EOB_PT_128, CDF_EOB_PT_128_STRIDE, 4, 8) X(CDF_EOB_PT_256, CDF_EOB_PT_256_STRIDE, 4, 9) X(CDF_EOB_PT_512, CDF_EOB_PT_512_STRIDE, 4, 10) X(CDF_EOB_PT_1024, CDF_EOB_PT_1024_STRIDE, 4, 11)
inline std::vector<uint16_t> build_cost_table(int qctx) {
  const uint16_t *cdf = av1_default_cdfs + (size_t)qctx * CDF_TOTAL;
  std::vector<uint16_t> cost(CDF_TOTAL, 0);
  auto rows = [&](int off, int stride, int nrows, int nsyms) {
    for (int r = 0; r < nrows; r++) for (int s = 0; s < nsyms; s++) {
      const uint16_t *row = cdf + off + r * stride;
      const uint32_t hi = s > 0 ? row[s - 1] : 32768u, lo = row[s];
      cost[off + r * stride + s] = (uint16_t)neg_log2_q9(hi - lo);
    }
  };
#define MI_ROW_(o, st, n, k) rows(o, st, n, k);
  MI_COST_ROWS(MI_ROW_)
#undef MI_ROW_
  return cost;
}

// ---------------------------------------------------------------- bit-level writers
class BitWriter {
 public:
  void put(uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) bit((v >> i) & 1); }
  void su(int v, int n) { put((uint32_t)v & ((1u << n) - 1), n); }
  void align() { while (nbits_ & 7) bit(0); }
  void trailing() { bit(1); align(); }
  const std::vector<uint8_t> &bytes() const { return buf_; }
 private:
  void bit(uint32_t b) { if ((nbits_ & 7) == 0) buf_.push_back(0); if (b) buf_.back() |= (uint8_t)(0x80 >> (nbits_ & 7)); nbits_++; }
  std::vector<uint8_t> buf_; size_t nbits_ = 0;
};
inline void put_leb128(std::vector<uint8_t> &o, uint64_t v) { do { uint8_t b = v & 0x7f; v >>= 7; if (v) b |= 0x80; o.push_back(b); } while (v); }

struct FrameHeaderInfo {                                       // what the OBU writer needs about one frame
  mi_av1_config cfg; int np; int sb_cols, sb_rows; QuantSel q; Tiling tiles;
  int lf_level[4], seg_n, seg_qidx[8];                         // one block: what the device reports back per frame (FrameDev::lf_out, 13 ints)
  int lf_sharp; int enable_cdef, cdef_damping, cdef_bits, cdef_y[8], cdef_uv[8];
  int enable_restoration, tx_mode_select;
};

inline std::vector<uint8_t> sequence_header(const FrameHeaderInfo &h) {   // spec 5.5, reduced_still_picture_header
  const mi_av1_config &c = h.cfg; const bool mono = c.chroma == 1;
  BitWriter b;
  b.put(mono ? 0 : 1, 3); b.put(1, 1); b.put(1, 1); b.put(31, 5);
  const int wb = 32 - __builtin_clz(std::max<uint32_t>(c.width - 1, 1)), hb = 32 - __builtin_clz(std::max<uint32_t>(c.height - 1, 1));
  b.put(wb - 1, 4); b.put(hb - 1, 4); b.put(c.width - 1, wb); b.put(c.height - 1, hb);
  b.put(0, 1); b.put(0, 1); b.put(1, 1);          // use_128x128_superblock, enable_filter_intra, enable_intra_edge_filter
  b.put(0, 1); b.put(h.enable_cdef, 1); b.put(h.enable_restoration, 1);   // enable_superres, enable_cdef, enable_restoration
  b.put(c.bit_depth > 8, 1);
  if (mono) b.put(1, 1);
  b.put(c.has_color_desc ? 1 : 0, 1);
  int cp = 2, tc = 2, mc = 2;
  if (c.has_color_desc) { cp = c.primaries; tc = c.transfer; mc = c.matrix; b.put(cp, 8); b.put(tc, 8); b.put(mc, 8); }
  if (mono) b.put(c.pixel_range, 1);
  else if (!(cp == 1 && tc == 13 && mc == 0)) b.put(c.pixel_range, 1);
  if (!mono) b.put(1, 1);                         // separate_uv_delta_q
  b.put(0, 1);                                    // film_grain_params_present
  b.trailing();
  return b.bytes();
}

inline std::vector<uint8_t> frame_obu_header(const FrameHeaderInfo &h, int tile_size_bytes, int ntiles) {   // spec 5.9 + tile group prefix
  BitWriter b;
  auto delta_q = [&](int d) { if (d) { b.put(1, 1); b.su(d, 7); } else b.put(0, 1); };
  b.put(0, 1); b.put(0, 1); b.put(0, 1);          // disable_cdf_update, allow_screen_content_tools, render_and_frame_size_different
  const int min_cols = tile_log2(64, h.sb_cols), max_cols = tile_log2(1, std::min(h.sb_cols, 64)), max_rows = tile_log2(1, std::min(h.sb_rows, 64));
  const int min_tiles = std::max(min_cols, tile_log2(2304, h.sb_rows * h.sb_cols));
  b.put(1, 1);                                    // uniform_tile_spacing_flag
  for (int k = min_cols; k < max_cols; k++) { if (k < h.tiles.cols_log2) b.put(1, 1); else { b.put(0, 1); break; } }
  for (int k = std::max(min_tiles - h.tiles.cols_log2, 0); k < max_rows; k++) { if (k < h.tiles.rows_log2) b.put(1, 1); else { b.put(0, 1); break; } }
  if (h.tiles.cols_log2 > 0 || h.tiles.rows_log2 > 0) { b.put(0, h.tiles.cols_log2 + h.tiles.rows_log2); b.put(tile_size_bytes - 1, 2); }
  b.put(h.q.base_q_idx, 8);
  delta_q(h.q.dc_qi[0] - h.q.base_q_idx);
  if (h.np > 1) {
    const bool diff = h.q.dc_qi[1] != h.q.dc_qi[2] || h.q.ac_qi[1] != h.q.ac_qi[2];
    b.put(diff, 1);
    delta_q(h.q.dc_qi[1] - h.q.base_q_idx); delta_q(h.q.ac_qi[1] - h.q.base_q_idx);
    if (diff) { delta_q(h.q.dc_qi[2] - h.q.base_q_idx); delta_q(h.q.ac_qi[2] - h.q.base_q_idx); }
  }
  b.put(0, 1);                                    // using_qmatrix
  // segmentation_params(): primary_ref_frame = NONE -> update_map = 1, temporal_update = 0, update_data = 1 without bits; one feature, ALT_Q
  b.put(h.seg_n > 0, 1);
  if (h.seg_n > 0)
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) {
      const bool on = j == 0 && i < h.seg_n;
      b.put(on, 1);
      if (on) b.put((uint32_t)(h.seg_qidx[i] - h.q.base_q_idx) & 0x1FFu, 9);    // feature_value su(1 + 8)
    }
  if (h.q.base_q_idx > 0) b.put(0, 1);            // delta_q_present
  b.put(h.lf_level[0], 6); b.put(h.lf_level[1], 6);
  if (h.np > 1 && (h.lf_level[0] || h.lf_level[1])) { b.put(h.lf_level[2], 6); b.put(h.lf_level[3], 6); }
  b.put(h.lf_sharp, 3); b.put(0, 1);
  if (h.enable_cdef) {
    b.put(h.cdef_damping - 3, 2); b.put(h.cdef_bits, 2);
    for (int i = 0; i < (1 << h.cdef_bits); i++) {
      b.put(h.cdef_y[i] >> 2, 4); b.put(h.cdef_y[i] & 3, 2);
      if (h.np > 1) { b.put(h.cdef_uv[i] >> 2, 4); b.put(h.cdef_uv[i] & 3, 2); }
    }
  }
  if (h.enable_restoration) {                     // lr_params(): RESTORE_SWITCHABLE on every plane, 64x64 units (lr_unit_shift 0)
    for (int p = 0; p < h.np; p++) b.put(1, 2);
    b.put(0, 1);
  }
  b.put(h.tx_mode_select, 1);                     // tx_mode_select: TX_MODE_SELECT / TX_MODE_LARGEST
  b.put(h.cfg.reduced_tx_set, 1);
  b.align();
  if (ntiles > 1) { b.put(0, 1); b.align(); }     // tile_start_and_end_present_flag
  return b.bytes();
}

// TD + sequence header + frame OBU from the per-tile payloads
inline std::vector<uint8_t> assemble_obus(const FrameHeaderInfo &h, const std::vector<std::pair<const uint8_t *, size_t>> &tiles) {
  size_t max_tile = 0, total = 0;
  for (size_t i = 0; i < tiles.size(); i++) { if (i + 1 < tiles.size()) max_tile = std::max(max_tile, tiles[i].second); total += tiles[i].second; }
  int tsb = 1; while (tsb < 4 && (max_tile ? max_tile - 1 : 0) >= ((size_t)1 << (8 * tsb))) tsb++;
  const std::vector<uint8_t> sh = sequence_header(h), fh = frame_obu_header(h, tsb, (int)tiles.size());
  const size_t payload = fh.size() + total + (tiles.size() - 1) * (size_t)tsb;
  std::vector<uint8_t> o; o.reserve(payload + sh.size() + 32);
  o.push_back(0x12); o.push_back(0x00);
  o.push_back(0x0A); put_leb128(o, sh.size()); o.insert(o.end(), sh.begin(), sh.end());
  o.push_back(0x32); put_leb128(o, payload); o.insert(o.end(), fh.begin(), fh.end());
  for (size_t i = 0; i < tiles.size(); i++) {
    if (i + 1 < tiles.size()) { const size_t v = tiles[i].second - 1; for (int k = 0; k < tsb; k++) o.push_back((uint8_t)(v >> (8 * k))); }
    o.insert(o.end(), tiles[i].first, tiles[i].first + tiles[i].second);
  }
  return o;
}

// ---------------------------------------------------------------- AVIF container (avif-serialize equivalent)
class BoxWriter {
 public:
  std::vector<uint8_t> out;
  void u8(unsigned v) { out.push_back((uint8_t)v); }
  void u16(unsigned v) { u8(v >> 8); u8(v); }
  void u32(uint32_t v) { u16(v >> 16); u16(v & 0xffff); }
  void str4(const char *s) { out.insert(out.end(), s, s + 4); }
  size_t begin(const char *type) { const size_t at = out.size(); u32(0); str4(type); return at; }
  size_t begin_full(const char *type, int version = 0, uint32_t flags = 0) { const size_t at = begin(type); u32(((uint32_t)version << 24) | flags); return at; }
  void end(size_t at) { const uint32_t n = (uint32_t)(out.size() - at); out[at] = n >> 24; out[at + 1] = n >> 16; out[at + 2] = n >> 8; out[at + 3] = n; }
  void patch32(size_t at, uint32_t v) { out[at] = v >> 24; out[at + 1] = v >> 16; out[at + 2] = v >> 8; out[at + 3] = v; }
};
inline std::vector<uint8_t> avif_container(const uint8_t *color, size_t color_len, const uint8_t *alpha, size_t alpha_len,
                                           uint32_t w, uint32_t h, int depth, int matrix, bool premultiplied,
                                           const uint8_t *exif, size_t exif_len) {
  // Exif (ravif with_exif :208-218 -> avif-serialize set_exif :470-472): an 'Exif' item after the image items, linked to the
  // colour item by a 'cdsc' reference; its payload is the HEIF ExifDataBlock = u32 exif_tiff_header_offset (0) + the bytes given.
  const bool has_alpha = alpha && alpha_len;
  const bool has_exif = exif && exif_len;
  const int n_img = 1 + (int)has_alpha, n_items = n_img + (int)has_exif, exif_id = n_img + 1;
  BoxWriter b;
  auto av1c = [&](int profile, bool mono) {
    const size_t a = b.begin("av1C");
    b.u8(0x81); b.u8((profile << 5) | 31); b.u8(((depth > 8) << 6) | (mono << 4) | (mono ? 0x0C : 0)); b.u8(0);
    b.end(a);
  };
  size_t a = b.begin("ftyp"); b.str4("avif"); b.u32(0); b.str4("avif"); b.str4("mif1"); b.str4("miaf"); b.end(a);
  const size_t meta = b.begin_full("meta");
  a = b.begin_full("hdlr"); b.u32(0); b.str4("pict"); b.u32(0); b.u32(0); b.u32(0); b.u8(0); b.end(a);
  a = b.begin_full("pitm"); b.u16(1); b.end(a);
  a = b.begin_full("iloc"); b.u8(0x44); b.u8(0); b.u16(n_items);
  size_t off_pos[3] = { 0, 0, 0 };
  for (int i = 0; i < n_img; i++) { b.u16(i + 1); b.u16(0); b.u16(1); off_pos[i] = b.out.size(); b.u32(0); b.u32((uint32_t)(i ? alpha_len : color_len)); }
  if (has_exif) { b.u16(exif_id); b.u16(0); b.u16(1); off_pos[2] = b.out.size(); b.u32(0); b.u32((uint32_t)(exif_len + 4)); }
  b.end(a);
  a = b.begin_full("iinf"); b.u16(n_items);
  for (int i = 0; i < n_img; i++) { const size_t e = b.begin_full("infe", 2); b.u16(i + 1); b.u16(0); b.str4("av01"); b.u8(0); b.end(e); }
  if (has_exif) { const size_t e = b.begin_full("infe", 2); b.u16(exif_id); b.u16(0); b.str4("Exif"); b.u8(0); b.end(e); }
  b.end(a);
  if (has_alpha || has_exif) {
    a = b.begin_full("iref");
    size_t e;
    if (has_alpha) { e = b.begin("auxl"); b.u16(2); b.u16(1); b.u16(1); b.end(e); }
    if (has_alpha && premultiplied) { e = b.begin("prem"); b.u16(1); b.u16(1); b.u16(2); b.end(e); }
    if (has_exif) { e = b.begin("cdsc"); b.u16(exif_id); b.u16(1); b.u16(1); b.end(e); }
    b.end(a);
  }
  a = b.begin("iprp");
  const size_t ipco = b.begin("ipco");
  size_t e = b.begin_full("ispe"); b.u32(w); b.u32(h); b.end(e);
  e = b.begin_full("pixi"); b.u8(3); b.u8(depth); b.u8(depth); b.u8(depth); b.end(e);
  av1c(1, false);
  e = b.begin("colr"); b.str4("nclx"); b.u16(1); b.u16(13); b.u16(matrix); b.u8(0x80); b.end(e);
  if (has_alpha) {
    av1c(0, true);
    e = b.begin_full("auxC"); { const char urn[] = "urn:mpeg:mpegB:cicp:systems:auxiliary:alpha"; b.out.insert(b.out.end(), urn, urn + sizeof(urn)); } b.end(e);
    e = b.begin_full("pixi"); b.u8(1); b.u8(depth); b.end(e);
  }
  b.end(ipco);
  e = b.begin_full("ipma"); b.u32(n_img);
  b.u16(1); b.u8(4); b.u8(1); b.u8(2); b.u8(0x80 | 3); b.u8(4);
  if (has_alpha) { b.u16(2); b.u8(4); b.u8(1); b.u8(7); b.u8(0x80 | 5); b.u8(6); }
  b.end(e);
  b.end(a);
  b.end(meta);
  a = b.begin("mdat");
  b.patch32(off_pos[0], (uint32_t)b.out.size()); b.out.insert(b.out.end(), color, color + color_len);
  if (has_alpha) { b.patch32(off_pos[1], (uint32_t)b.out.size()); b.out.insert(b.out.end(), alpha, alpha + alpha_len); }
  if (has_exif) { b.patch32(off_pos[2], (uint32_t)b.out.size()); b.u32(0); b.out.insert(b.out.end(), exif, exif + exif_len); }
  b.end(a);
  return std::move(b.out);
}

// deblock level: libaom/rav1e q-based estimate (fast_deblock path)
inline int deblock_level_from_q(int ac_q, int bd) {
  int lvl = bd == 8 ? (ac_q * 17563 - 421574 + (1 << 17)) >> 18 : ((ac_q * 20723 + 4060632 + (1 << 19)) >> 20) - 4;
  return std::clamp(lvl, 0, 63);
}

}  // namespace mi

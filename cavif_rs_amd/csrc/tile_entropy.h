// tile_entropy.h -- kernel K4: the tile's final bitstream (spec 5.11 syntax, 8.2 symbol coder, 8.3 CDF selection).
// One workgroup per tile, a pipeline of wavefronts over the tile's superblocks:
//   producer (wave 0)      walks the mode-info maps K1 left in HBM in AV1 coding order and lays every symbol down as a 32-bit record --
//                          (CDF row, symbol, alphabet) or a literal -- in the superblock's record buffer; a transform block's coefficient
//                          symbols are written by all lanes at once (contexts depend on the level map only, exclusive scans place them);
//   adapters (MI_K4_ADAPTERS waves)  own disjoint sets of CDF rows (a row's state depends only on the symbols coded through that row, never on
//                          the coder state): each turns ITS records into bounds records in place (read the row, pick fl / fh, adapt);
//   coder (last wave)      runs the range recurrence of the finished records on the scalar unit and leaves `low` -- a sum -- to its 64 lanes: prefix sum of the
//                          shifts, every symbol's contribution added into a ring of byte accumulators in LDS (RangeEncDev below).
// While the coder works on superblock k the adapters convert k + 1 and the producer writes k + 2; one workgroup barrier per superblock.
// A tile's serial chain used to be ~100 instructions per symbol on one wave (a lone wave issues a dependent instruction every ~7 cycles:
// 46 ms per 1080p tile); it is now the busiest stage's share.
// rav1e equivalents (absent from /root/reference): src/ec.rs (WriterBase), src/context/*.rs.
#pragma once
#include "dev_common.h"
#include "dev_rate.h"
#include "restoration.h"

// The coder's state.  An AV1 range coder keeps two things: the 16-bit range, whose next value depends on the symbol before it -- a serial chain --, and `low`,
// which is nothing but a sum: symbol i adds delta_i = rng_i - u_i at the bit position the window stands at, T_i = the normalisation shifts d_k of all symbols before
// it.  The tile's bytes are the big number  L = sum_i delta_i * 2^-T_i  written from bit 14 downwards (libaom od_ec_enc: cnt starts at -9, the first byte leaves when
// 17 window bits are there).  So the wave's scalar unit runs the range recurrence only (k4_rng_step: 19 instructions per symbol, nothing about `low`, no flush branch)
// and drops (delta, d) into lane j of two vector registers; the 64 lanes then place the chunk's deltas at once: prefix sum of d -> bit offset -> three byte slices
// added into a ring of byte accumulators in LDS (ds_add).  Accumulators below the window are final and leave as the pre-carry units re_finish_dev resolves
// (a unit = its own low byte + what the two accumulators after it hold above their low bytes, so that a unit's carry part stays below 2^8).
#define MI_K4_ACC 256                      /* ring entries: a chunk of 64 symbols moves the window by at most 64 * 15 bits = 120 bytes */
struct RangeEncDev {
  uint16_t *pre; uint32_t cap;           // pre-carry units in HBM (one per output byte: byte + carry part), stored while below cap
  uint32_t rng;                          // the range, 2^15 <= rng < 2^16 (wave-uniform: an SGPR)
  uint32_t tbits;                        // T: bits the window has moved so far
  uint32_t flushed;                      // units handed to `pre` so far (stored or not: overflow <=> the total exceeds cap)
  LDS uint32_t *acc;                     // byte k of the stream accumulates in acc[k & (MI_K4_ACC - 1)]
};
__device__ __forceinline__ void re_init_dev(RangeEncDev *e, uint16_t *pre, uint32_t cap, LDS uint32_t *acc) {
  e->pre = pre; e->cap = cap; e->rng = 0x8000; e->tbits = 0; e->flushed = 0; e->acc = acc;
}
// units [e->flushed, end): whole wave.  Unit k needs acc[k], acc[k + 1], acc[k + 2] final.
__device__ __forceinline__ void re_flush_units(RangeEncDev *e, uint32_t end) {
  for (uint32_t base = e->flushed; base < end; base += 64) {
    const uint32_t k = base + (uint32_t)LANE;
    const uint32_t a0 = e->acc[k & (MI_K4_ACC - 1)], a1 = e->acc[(k + 1) & (MI_K4_ACC - 1)], a2 = e->acc[(k + 2) & (MI_K4_ACC - 1)];
    if (k < end && k < e->cap) e->pre[k] = (uint16_t)((a0 & 255u) + ((a1 >> 8) & 255u) + (a2 >> 16));
    WAVE_SYNC();
    if (k < end) e->acc[k & (MI_K4_ACC - 1)] = 0;
    WAVE_SYNC();
  }
  if (end > e->flushed) e->flushed = end;
}
// ---- symbol records ----
//   adaptive symbol  bits 31..30 = 00: bits 0..15 the CDF row's offset, 16..19 the symbol, 20..23 alphabet size - 1;
//                    bit 29: instead of a symbol, the "is it split" bool of a partition node at the frame edge, priced from the row as it stands
//                    (bit 16 = has_cols), resolved by the row's adapter in coding order
//   bounds           bits 31..30 = 01: bits 0..9 fl >> 6 (512 = the top of the range), 10..19 fh >> 6, 20..23 alphabet size - 1 - symbol
//   literal          bit 31: bits 0..19 the value, 20..24 the bit count (equiprobable bools, most significant first)
#define K4_REC(off, s, ns) ((uint32_t)(off) | ((uint32_t)(s) << 16) | ((uint32_t)((ns) - 1) << 20))
#define K4_PEDGE(off, has_cols) ((uint32_t)(off) | ((uint32_t)(has_cols) << 16) | 0x20000000u)
#define K4_LIT(v, nb) (0x80000000u | (uint32_t)(v) | ((uint32_t)(nb) << 20))
#define K4_BOUNDS(fl6, fh6, nms) (0x40000000u | (uint32_t)(fl6) | ((uint32_t)(fh6) << 10) | ((uint32_t)(nms) << 20))
// an equiprobable bool as bounds: (fl >> 6, fh >> 6, N - 1 - s) = (256, 0, 0) for a one, (512 = the top, 256, 1) for a zero -- sign bits, the most frequent
// literals, reach the coder as ordinary records
#define K4_BIT(b) ((b) ? K4_BOUNDS(256u, 0u, 0u) : K4_BOUNDS(512u, 256u, 1u))
// adapter waves per tile: two when the launch fills the device (1024 tiles x 4 waves = every SIMD's four wave slots at K4's register count; four
// adapters were measured slower there: a second round of workgroups), four when it does not (single images: the adapters are the busiest stage)
#ifndef MI_K4_ADAPTERS
#define MI_K4_ADAPTERS 2
#endif
#ifndef MI_K4_LB_EXTRA
#define MI_K4_LB_EXTRA                             /* probe builds: a second __launch_bounds__ argument (workgroups per CU) */
#endif
#define MI_K4_ADAPTERS_SPARSE 4
#define MI_K4_THREADS_OF(NA) (64 * (2 + (NA)))
// records of a typical worst superblock: per coefficient the base level, four base-range symbols, the sign and two more; per transform
// block five header symbols; per block sixteen; the partition nodes and the restoration units ...
#define MI_K4_SB_RECORDS(np) ((uint32_t)(4096 * 8 * (np) + 256 * 5 * (np) + 256 * 16 + 512) > 2u * MI_K4_TXB_RECORDS ? (uint32_t)(4096 * 8 * (np) + 256 * 5 * (np) + 256 * 16 + 512) : 2u * MI_K4_TXB_RECORDS)
// ... and what the records between two capacity checks (a block's header + one transform block of at most 1024 coefficients: per coefficient the base
// level, four base-range symbols, the sign and a Golomb tail of up to 2 * 15 - 1 bit records) can need at most: the producer hands a buffer on early
// when less than this is left (k4_room), so a buffer holds any superblock's records in as many pieces as it takes
#define MI_K4_TXB_RECORDS ((uint32_t)(1024 * 36 + 1024))
// which adapter owns a CDF row: the low bits of its offset (the hot tables have strides 5 and 3: neighbouring contexts and the same context of
// neighbouring transform sizes land on different waves)
// MI_K4_SPREAD (measured, not kept): with two adapter waves the producer and the coder take rows as well -- owners 2 and 3: the producer turns its own rows' records
// into bounds before it hands a buffer on, the coder right before it codes one (a row's state only depends on the symbols coded through it, in order, and each of
// them meets the buffers in order).  After the coder lost its `low` arithmetic (RangeEncDev) the second adapter was the busiest stage of a long tile by a third
// (12.8 ms against 7.5 / 9.5 / 8.7, profiles/r06k_k4_stage_profiles.txt); the table below -- owner by (context index + table block) & 7, fitted on the oracle's symbol
// streams of two 1080p pictures -- evens the four stages out (all 8 ... 11 ms, profiles/r06m_k4s2_stage.txt), the bytes are the same, and the kernel is SLOWER:
// 15.5 against 14.6 ms (profiles/r06l_*, r06m_*).  The four waves of a SIMD -- one stage of each of four tiles -- share its instruction issue: what the two extra
// passes over every buffer add is paid by all of them, and a tile does not end sooner for being balanced.
#ifndef MI_K4_SPREAD
#define MI_K4_SPREAD 0
#endif
#ifndef MI_K4_SPREAD_LUT
#define MI_K4_SPREAD_LUT 0x8704u                         /* 2 bits per bucket, bucket 0 first: 0 1 0 0 3 1 0 2 */
#endif
template <int NA> __device__ __forceinline__ int k4_row_owner(uint32_t row) {
  static_assert(NA == 2 || NA == 3 || NA == 4, "two, three or four adapters");
  // the parity (or the low two bits) of (context index + table block) of the stride-5 tables -- the coefficient base-level rows are 85 % of all
  // adaptive symbols and four of them (contexts 21 / 22 of the 16x16 luma and chroma blocks) carry 70 %; plain offset parity put 68 % of a 1080p
  // tile's symbols on one of two waves, this puts 53 ... 58 % there, and 35 ... 43 % on the busiest of four (measured on the oracle's symbol stream)
  if constexpr (NA == 3) return (int)((row / 5u + row / 210u) % 3u);
  if constexpr (NA == 2 && MI_K4_SPREAD) return (int)((MI_K4_SPREAD_LUT >> (2u * ((row / 5u + row / 210u) & 7u))) & 3u);
  return (int)((row / 5u + row / 210u) & (uint32_t)(NA - 1));
}
// exclusive prefix sum over the 64 lanes (lane order), *total = the wave's sum
__device__ __forceinline__ int wave_excl_scan_i32(int v, int *total) {
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl(x, imax_(LANE - d, 0)); if (LANE >= d) x += t; }
  *total = __builtin_amdgcn_readlane(x, 63);
  return x - v;
}
#ifndef MI_BALLOT64                                      /* (the CPU test harness tests/emu/ predefines both, like LDS in dev_common.h) */
#define MI_BALLOT64(p) __builtin_amdgcn_ballot_w64(p)
#endif
__device__ __forceinline__ unsigned long long k4_ballot(bool p) { return MI_BALLOT64(p); }
// Adapter `a` over one superblock's records: only its own rows' records are visited (ballot of the chunk, lowest set bit first).  The row sits in a
// register (lane i = entry i, lane nsyms = the adaptation counter) and stays there while consecutive records name it.
#ifndef MI_BITSET0_64                                    /* clears bit `j` of a 64-bit scalar mask: one instruction instead of the four of m &= m - 1 (K4 -0.45 ms, profiles/r06R_*) */
#define MI_BITSET0_64(m, j) asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j))
#endif
template <int NA> __device__ __forceinline__ void k4_adapt_sb(LDS uint16_t *cdf, uint32_t *buf, int n_in, int a) {
  const int n = uni32(n_in), i = LANE;
  int row = -1, v = 0;
  uint32_t nx = n > 0 ? buf[imin_(i, n - 1)] : 0u;
  for (int cb = 0; cb < n; cb += 64) {
    const uint32_t rv = cb + i < n ? nx : 0x80000000u;
    if (cb + 64 < n) nx = buf[imin_(cb + 64 + i, n - 1)];                    // the next chunk's load flies behind this chunk's chain (its records of this adapter's rows are nobody else's to change)
    const bool mine = (rv >> 30) == 0u && k4_row_owner<NA>(rv & 0xFFFFu) == a;
    unsigned long long todo = k4_ballot(mine);
    uint32_t outv = rv;
    while (todo) {
      const int j = __builtin_ctzll(todo); MI_BITSET0_64(todo, j);
      const uint32_t rec = (uint32_t)__builtin_amdgcn_readlane((int)rv, j);
      const int r = (int)(rec & 0xFFFFu);
      uint32_t out;
      if (__builtin_expect((rec & 0x20000000u) != 0u, 0)) {  // partition node at the frame edge (rare: kept off the straight path): P(the partitions that split this way), not adapted
        const int cv = cdf[r + imin_(i, 10)];
        uint32_t psum = 0;
        const uint32_t set = (rec & 0x10000u) ? 0x2DCu : 0x17Au;              // has_cols: partitions 2, 3, 4, 6, 7, 9; else 1, 3, 4, 5, 6, 8
#pragma unroll
        for (int q = 1; q < 10; q++) if ((set >> q) & 1u) psum += (uint32_t)__builtin_amdgcn_readlane(cv, q - 1) - (uint32_t)__builtin_amdgcn_readlane(cv, q);
        out = K4_BOUNDS(psum >> 6, 0, 0);
      } else {
        const int s = (int)((rec >> 16) & 15u), nsyms = (int)((rec >> 20) & 15u) + 1;
        if (r != row) { v = cdf[r + imin_(i, nsyms)]; row = r; }
        const uint32_t fl0 = (uint32_t)__builtin_amdgcn_readlane(v, s - 1), fh = (uint32_t)__builtin_amdgcn_readlane(v, s);   // (s = 0: v_readlane takes the lane modulo 64, the value is not used)
        const int cnt = __builtin_amdgcn_readlane(v, nsyms);
        out = K4_BOUNDS(s > 0 ? fl0 >> 6 : 512u, fh >> 6, nsyms - 1 - s);
        // spec 8.3.2, one step for every entry: rate = 3 + (cnt > 15) + (cnt > 31) + min(floor(log2(nsyms)), 2); cnt <= 32.  Written without a branch:
        // entries below the symbol move up by (32768 - v) >> rate, the others down by v >> rate; entry nsyms - 1 stays, entry nsyms counts.
        const int rate = 4 + (nsyms > 3) + (cnt >> 4);
        const bool below = i < s;
        const int t = below ? 32768 - v : v, sh = t >> rate;
        const int moved = below ? v + sh : v - sh;
        const int va = i == nsyms ? imin_(cnt + 1, 32) : (i < nsyms - 1 ? moved : v);
        if (i <= nsyms) cdf[r + i] = (uint16_t)va;            // (entry nsyms - 1 is rewritten unchanged; storing every symbol's row is CHEAPER than writing a row back when
                                                              // another one comes -- K4 +0.2 ms with that, profiles/r06R_ab_k4_adapter_diet.txt)
        v = va;
      }
      outv = i == j ? out : outv;
    }
    if (mine) buf[cb + i] = outv;
  }
}
// The coder over one superblock's finished records.
#ifndef MI_SMUL32
#define MI_SMUL32(r, a, b) asm("s_mul_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b))
#endif
#ifndef MI_SSEL_GE                                       /* r = a >= b ? x : y on the scalar unit, both sides computed (as a C conditional the compiler makes it a branch around the multiply) */
#define MI_SSEL_GE(r, a, b, x, y) asm("s_cmp_ge_u32 %1, %2\n\ts_cselect_b32 %0, %3, %4" : "=s"(r) : "s"(a), "s"(b), "s"(x), "s"(y) : "scc")
#endif
#ifndef MI_WRITELANE                                     /* v_writelane_b32 has no clang builtin; a constant lane, a scalar value */
#define MI_WRITELANE(v, s, lane) asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(lane))
#endif
__device__ __forceinline__ uint32_t k4_smul(uint32_t a, uint32_t b) { uint32_t r; MI_SMUL32(r, a, b); return r; }   // (the compiler would route a provably-24-bit uniform product through the vector unit)
// One symbol of the range recurrence, all of it on the scalar unit: record J's bounds (fl >> 6, fh >> 6, 4 * (N - 1 - s): lane J of three registers) against the
// range; what the symbol adds to `low` and how far it moves the window go to lane J of vdelta / vd.  libaom od_ec_encode_q15 + od_ec_enc_normalize without `low`.
template <int J> __device__ __forceinline__ void k4_rng_step(uint32_t &rng, int vfl, int vfh, int vnm, int &vdelta, int &vd) {
  const uint32_t fl6 = (uint32_t)__builtin_amdgcn_readlane(vfl, J), fh6 = (uint32_t)__builtin_amdgcn_readlane(vfh, J), nm = (uint32_t)__builtin_amdgcn_readlane(vnm, J);
  const uint32_t r = rng, r8 = r >> 8, top = r - nm;
  const uint32_t v = k4_smul(r8, fh6) >> 1;                                     // (the 4 * (N - 1 - s) of both bounds: inside `top`)
  uint32_t u; MI_SSEL_GE(u, fl6, 512u, top, (k4_smul(r8, fl6) >> 1) + 4u);      // the first symbol of an alphabet: u = rng
  const uint32_t nw = u - v, delta = top - u;
  const int d = __builtin_clz(nw) - 16;                                          // (never 0: every symbol keeps a range of 4 at least)
  rng = nw << d;
  MI_WRITELANE(vdelta, delta, J); MI_WRITELANE(vd, d, J);
}
template <int J0> __device__ __forceinline__ void k4_rng_steps16(uint32_t &rng, int vfl, int vfh, int vnm, int &vdelta, int &vd) {
  k4_rng_step<J0 + 0>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 1>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 2>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 3>(rng, vfl, vfh, vnm, vdelta, vd);
  k4_rng_step<J0 + 4>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 5>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 6>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 7>(rng, vfl, vfh, vnm, vdelta, vd);
  k4_rng_step<J0 + 8>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 9>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 10>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 11>(rng, vfl, vfh, vnm, vdelta, vd);
  k4_rng_step<J0 + 12>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 13>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 14>(rng, vfl, vfh, vnm, vdelta, vd); k4_rng_step<J0 + 15>(rng, vfl, vfh, vnm, vdelta, vd);
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl(x, imax_(LANE - d, 0)); if (LANE >= d) x += t; }
  return x;
}
__device__ __forceinline__ void k4_code_sb(RangeEncDev *e, const uint32_t *buf, int n_in) {
  const int n = uni32(n_in);
  uint32_t rv = n > 0 ? buf[imin_(LANE, n - 1)] : 0u;
  uint32_t rng = e->rng;
  for (int cb = 0; cb < n; cb += 64) {
    const int m = imin_(64, n - cb);
    const uint32_t cur = LANE < m ? rv : K4_BOUNDS(512u, 0u, 0u);              // past the end: a symbol that leaves range and window where they are
    if (cb + 64 < n) rv = buf[imin_(cb + 64 + LANE, n - 1)];                 // the next chunk's load flies behind this chunk's arithmetic
    // every record carries its bounds (the adapters turned the symbols into them, the producer wrote literal bits as such)
    const int vfl = (int)(cur & 1023u), vfh = (int)((cur >> 10) & 1023u), vnm = (int)((cur >> 18) & 60u);
    int vdelta = 0, vd = 0;
    k4_rng_steps16<0>(rng, vfl, vfh, vnm, vdelta, vd);
    if (m > 16) k4_rng_steps16<16>(rng, vfl, vfh, vnm, vdelta, vd);
    if (m > 32) k4_rng_steps16<32>(rng, vfl, vfh, vnm, vdelta, vd);
    if (m > 48) k4_rng_steps16<48>(rng, vfl, vfh, vnm, vdelta, vd);
    // the chunk's 64 additions to `low`, lane-parallel: symbol i's delta has its bit 0 at stream bit 14 + T_i (bit 0 = the top bit of byte 0)
    const int incl = wave_incl_scan_i32(vd);
    const uint32_t q = 14u + e->tbits + (uint32_t)(incl - vd), kk = q >> 3;
    const uint32_t val = (uint32_t)vdelta << (7u - (q & 7u));                  // < 2^23: byte kk and the two above it
    if (val & 255u) (void)__hip_atomic_fetch_add(e->acc + (kk & (MI_K4_ACC - 1)), val & 255u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((val >> 8) & 255u) (void)__hip_atomic_fetch_add(e->acc + ((kk - 1u) & (MI_K4_ACC - 1)), (val >> 8) & 255u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (val >> 16) (void)__hip_atomic_fetch_add(e->acc + ((kk - 2u) & (MI_K4_ACC - 1)), val >> 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    e->tbits += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
    WAVE_SYNC();
    // later symbols reach byte ((14 + T) >> 3) - 2 at the highest: the units four and more below the window are final
    const uint32_t kmin = (14u + e->tbits) >> 3;
    if (kmin >= 4u && kmin - 4u > e->flushed) re_flush_units(e, kmin - 4u);
  }
  e->rng = rng;
}

// Ends the stream (libaom od_ec_enc_done: low rounded up to a multiple of 2^14, bit 14 set, as many bytes as reach that bit) and resolves the carries of the
// pre-carry units into `out`; returns the number of bytes (0xFFFFFFFF: a buffer overflowed).
// Whole wave: the units form one big base-256 number (bits 8.. of a unit belong to the byte above it), normalised 64 units at a time from
// the least significant end with a carry-lookahead scan over the lanes (generate: the byte sum reaches 256, propagate: it is 255) -- exactly the
// bytes of the unit-by-unit loop `carry += pre[i]; out[i] = carry & 255; carry >>= 8` without its thousands of dependent global round trips.
__device__ __forceinline__ uint32_t re_finish_dev(RangeEncDev *e, uint8_t *out, uint32_t out_cap) {
  // the window's bit 0 stands at stream bit q = 14 + T, i.e. at bit sh of byte kend; V = what the stream holds from byte kend - 3 down (every accumulator there is
  // still in the ring: re_flush_units stays five bytes behind the window).  Rounding up only adds: the difference goes into the ring like a symbol's delta.
  const uint32_t q = 14u + e->tbits, kend = q >> 3, sh = 7u - (q & 7u);
  {
    unsigned long long V = 0;
    for (uint32_t j = 0; j < 4; j++) if (kend >= j) V += (unsigned long long)e->acc[(kend - j) & (MI_K4_ACC - 1)] << (8 * j);
    const unsigned long long A = 0x3FFFull << sh, B = 0x4000ull << sh;
    const uint32_t add = (uint32_t)((((V + A) & ~(B - 1)) | B) - V);             // < 2^23
    WAVE_SYNC();
    if (LANE == 0) {
      e->acc[kend & (MI_K4_ACC - 1)] += add & 255u;
      e->acc[(kend - 1u) & (MI_K4_ACC - 1)] += (add >> 8) & 255u;
      if (kend >= 2u) e->acc[(kend - 2u) & (MI_K4_ACC - 1)] += add >> 16;
    }
    WAVE_SYNC();
  }
  const uint32_t nb = (e->tbits >> 3) + 1u, nu = kend + 1u;    // bytes of the tile; units down to the window's last byte (nb + 1 or nb + 2: the bytes past nb are zero once the carries are in)
  re_flush_units(e, nu);
  if (nu > e->cap || nb > out_cap) return 0xFFFFFFFFu;
  __threadfence();                                         // the unit stores, read back below in another lane order
  WAVE_SYNC();
  int cin = 0, ein = 0;                                    // carry into the chunk's last byte; the high bits of the unit right after the chunk
  for (int top = (int)nu; top > 0; top -= 64) {            // lane i <-> unit top - 64 + i (lane 63 the least significant)
    const int idx = top - 64 + LANE;
    const int p = idx >= 0 ? (int)e->pre[idx] : 0;
    const int hi = p >> 8;
    int en = __shfl(hi, imin_(LANE + 1, 63));
    if (LANE == 63) en = ein;
    const int t = (p & 0xFF) + en;
    int G = t >= 256, P = t == 255;                        // (generate, propagate) of the lanes LANE .. 63, by doubling
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int g2 = __shfl(G, imin_(LANE + d, 63)), p2 = __shfl(P, imin_(LANE + d, 63));
      if (LANE + d < 64) { G |= P & g2; P &= p2; }
    }
    const int gx = __shfl(G, imin_(LANE + 1, 63)), px = __shfl(P, imin_(LANE + 1, 63));
    const int ci = LANE == 63 ? cin : (gx | (px & cin));
    if (idx >= 0 && idx < (int)nb) out[idx] = (uint8_t)(t + ci);
    cin = __builtin_amdgcn_readlane(G | (P & cin), 0); ein = __builtin_amdgcn_readlane(hi, 0);
  }
  return nb;
}

struct TileWriter {                                                 // the producer's state
  const FrameDev *f; TileB t;
  uint32_t *out; uint32_t n, cap;                                   // the current record buffer (HBM), records written so far, its capacity
  uint32_t *bufs; LDS uint32_t *nrec; LDS int *total; int chunk;    // the ring of record buffers, their published counts, the number of buffers of the tile (-1 while unknown), the current one's index
  LDS int32_t *qc; LDS uint8_t *lev; const LDS uint16_t *ls;       // LDS staging + LDS copy of the scan tables
  LDS uint16_t *rec_off, *rec_br; LDS uint32_t *rec_lv;            // per-coefficient context rows / levels of the current transform block
  LDS int *lr_ref;                                                  // RefSgrXqd[plane][2]
  LDS uint16_t *cdf; int own_rows;                                  // the tables; own_rows: the producer adapts the rows k4_row_owner gives it (owner 2)
  int cdef_pending;                                                 // the 64x64 superblock being walked has not signalled its cdef_idx yet
  int sb_cols_tile;
  // Frame scalars that steer the walk, pinned to SGPRs once per tile: every control value the walk loads (skip, modes, transform sizes, eob,
  // block sizes, restoration types) goes through v_readfirstlane as well, so that the record counter and the walk's branches stay scalar.
  int np, mi_rows, mi_cols, ms, tx_mode_select, enable_cdef, cdef_bits, enable_restoration, sb_cols, fw, fh, seg_n;
  struct TxCfg { int reduced_tx_set, base_q_idx; } txc;
#if MI_PROFILE == 2
  unsigned long long prof[16], pt;
#endif
};
// the producer's emitters (wave-uniform arguments; lane 0 stores)
__device__ __forceinline__ void k4_put(TileWriter *w, uint32_t rec) { if (LANE == 0 && w->n < w->cap) w->out[w->n] = rec; w->n++; }
__device__ __forceinline__ void k4_sym(TileWriter *w, int s, int off, int ns) { k4_put(w, K4_REC(uni32(off), uni32(s), uni32(ns))); }
// (one copy of the adapter loop for the producer's hand-off sites; the arguments by value: the writer's state stays in registers)
__device__ __attribute__((noinline)) void k4_adapt_own_rows(LDS uint16_t *cdf, uint32_t *buf, int n, int owner) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the records were stored by other lanes of this wave
  WAVE_SYNC();
  k4_adapt_sb<2>(cdf, buf, n, owner);
}
// The producer hands its buffer to the adapters: publishes the count, meets the other stages at the workgroup barrier (they sit in the kernel's
// stage loop; a barrier is a barrier wherever the wave executes it) and moves on to the next of the three buffers.  `last`: the tile ends here.
// (Measured and not kept, profiles/r06n_*: a ring of six half-size buffers with release / acquire counters in LDS instead of the barrier -- stages running ahead of
// each other -- is byte-identical and exactly as fast, 14.65 ms: what a tile loses against its busiest stage is not the lockstep.)
__device__ __forceinline__ void k4_handoff(TileWriter *w, bool last) {
  if (w->own_rows) k4_adapt_own_rows(w->cdf, w->out, (int)(w->n < w->cap ? w->n : w->cap), 2);
  WAVE_SYNC();
  if (LANE == 0) { w->nrec[w->chunk % 3] = w->n; if (last) *w->total = w->chunk + 1; }
  __syncthreads();
  w->chunk++;
  w->out = w->bufs + (size_t)(w->chunk % 3) * w->cap; w->n = 0;
}
// room for a block header + one transform block, or the buffer goes on early (wave-uniform: the record counter is scalar)
__device__ __forceinline__ void k4_room(TileWriter *w) { if (w->n + MI_K4_TXB_RECORDS > w->cap) k4_handoff(w, false); }
__device__ __forceinline__ void k4_lit(TileWriter *w, uint32_t v, int nbits) {           // equiprobable bools, most significant first: a record per bit
  const int nb = uni32(nbits); const uint32_t uv = (uint32_t)uni32((int)v);
  for (int q = nb - 1; q >= 0; q--) k4_put(w, K4_BIT((uv >> q) & 1u));
}
#define CDF_LR_SWITCHABLE CDF_TOTAL                                 /* the switchable restoration_type row (3 symbols + counter) sits behind the tables in LDS */
// K4 phase timers (probe builds, -DMI_PROFILE=2): cycles per phase and event counts, flushed into wave 3's slots of the tile's K1 record
#ifndef MI_PROFILE
#define MI_PROFILE 0
#endif
#if MI_PROFILE == 2
#define K4PH(i) do { const unsigned long long n_ = clock64(); w->prof[i] += n_ - w->pt; w->pt = n_; } while (0)
#define K4CNT(i, n) do { w->prof[i] += (unsigned long long)(n); } while (0)
#else
#define K4PH(i) do {} while (0)
#define K4CNT(i, n) do {} while (0)
#endif
#define U_(v) uni32((int)(v))                              /* wave-uniform values loaded from memory: pinned to SGPRs (the coder runs on the scalar unit) */

// One transform block's symbols (levels + padded level map already staged in LDS).  (P) every lane derives the CDF rows (contexts) of its own
// scan positions -- they depend only on the level map -- into LDS; (R) the records in coding order: the header symbols (lane 0), then per scan
// position from eob - 1 down to 0 the base level and its base-range symbols, then per position from 0 up the sign and the Golomb tail;
// exclusive scans place each lane's records and the lanes write them side by side.
__device__ __forceinline__ void code_coeffs_lane0(TileWriter *w, int eob_in, int plane, int txs, int txtype, int skip_ctx, int dc_ctx,
                                                  int tx_off, int tx_sym, int tx_ns) {
  const int eob = uni32(eob_in);
  const LDS int32_t *qc = w->qc; const LDS uint8_t *lev = w->lev;
  const bool rect = txs > 4;                              // 5 = 4x8, 6 = 8x4 (dev_rect.h)
  const int bwl = rect ? (txs == 5 ? 2 : 3) : imin_(5, 2 + txs), bhl = rect ? (txs == 5 ? 3 : 2) : bwl, n = 1 << bwl, nh = 1 << bhl;
  const int pt = plane > 0, cls = tx_class_of(txtype), txs_ctx = rect ? 1 : txs;
  k4_room(w);
  k4_sym(w, eob == 0, CDF_TXB_SKIP + (txs_ctx * 13 + skip_ctx) * CDF_TXB_SKIP_STRIDE, 2);
  K4CNT(9, 1); K4CNT(10, eob == 0);
  if (eob == 0) { K4PH(3); return; }
  // ---- (P) contexts, lane-parallel
  const int st = n + 4, area = n * nh;
  for (int c = LANE; c < eob; c += 64) {
    const int p = rect ? rect_scan_pos(bwl, bhl, cls, c) : scan_pos(w->ls, n, cls, c), row = p >> bwl, col = p & (n - 1);
    const int v = qc[p], level = iabs_(v);
    const LDS uint8_t *L = lev + row * st + col;
    int off;                                                 // last position: its base_eob CDF row; the others: the base context (0..41)
    if (c == eob - 1) {
      const int ctx = c == 0 ? 0 : (c <= area / 8 ? 1 : (c <= area / 4 ? 2 : 3));
      off = CDF_COEFF_BASE_EOB + ((txs_ctx * 2 + pt) * 4 + ctx) * CDF_COEFF_BASE_EOB_STRIDE;
    } else {
      int bctx = base_ctx(L, st, cls, row, col);
      if (rect && cls == TXC_2D && !(row == 0 && col == 0)) {          // spec Coeff_Base_Ctx_Offset of the 2:1 sizes
        const int mg = imin_(L[1], 3) + imin_(L[st], 3) + imin_(L[st + 1], 3) + imin_(L[2], 3) + imin_(L[2 * st], 3), m = imin_((mg + 1) >> 1, 4);
        bctx = bhl > bwl ? m + (row < 2 ? 11 : (row + col < 4 ? 6 : 21)) : m + (col < 2 ? 16 : (row + col < 4 ? 6 : 21));
      }
      off = CDF_COEFF_BASE + ((txs_ctx * 2 + pt) * 42 + bctx) * CDF_COEFF_BASE_STRIDE;
    }
    int boff = 0;
    if (level > 2) boff = CDF_COEFF_BR + ((imin_(txs_ctx, 3) * 2 + pt) * 21 + br_ctx(L, st, cls, row, col, c)) * CDF_COEFF_BR_STRIDE;
    w->rec_off[c] = (uint16_t)off; w->rec_br[c] = (uint16_t)boff; w->rec_lv[c] = ((uint32_t)level << 1) | (uint32_t)(v < 0);
  }
  WAVE_SYNC();
  K4PH(3);
  // ---- (R) records
  if (tx_off >= 0) k4_sym(w, tx_sym, tx_off, tx_ns);
  {
    const int eob_pt = eob_to_pt(eob), eob_multi = bwl + bhl - 4;
    k4_sym(w, eob_pt - 1, eob_pt_cdf(eob_multi, pt, cls), 5 + eob_multi);
    if (eob_pt >= 3) {
      const int nb = eob_pt - 2, rem = eob - ((1 << (eob_pt - 2)) + 1), hi = (rem >> (nb - 1)) & 1;
      k4_sym(w, hi, CDF_EOB_EXTRA + ((txs_ctx * 2 + pt) * 9 + (eob_pt - 3)) * CDF_EOB_EXTRA_STRIDE, 2);
      if (nb > 1) k4_lit(w, (uint32_t)rem & ((1u << (nb - 1)) - 1), nb - 1);
    }
  }
  uint32_t *recs = w->out; const uint32_t cap = w->cap;
  uint32_t nrec = w->n;
  for (int top = eob - 1; top >= 0; top -= 64) {             // lane L takes position top - L: lane order = coding order
    const int c = top - LANE;
    int cnt = 0, level = 0, off = 0, boff = 0;
    if (c >= 0) { level = (int)(w->rec_lv[c] >> 1); off = w->rec_off[c]; boff = w->rec_br[c]; cnt = 1 + (level > 2 ? imin_(4, (level - 3) / 3 + 1) : 0); }
    int tot; const uint32_t at = nrec + (uint32_t)wave_excl_scan_i32(cnt, &tot);
    if (c >= 0 && at + (uint32_t)cnt <= cap) {
      recs[at] = c == eob - 1 ? K4_REC(off, imin_(level, 3) - 1, 3) : K4_REC(off, imin_(level, 3), 4);
      int rem = level - 3;
      for (int q = 1; q < cnt; q++) { const int s2 = imin_(rem, 3); recs[at + q] = K4_REC(boff, s2, 4); rem -= s2; }
    }
    nrec += (uint32_t)tot;
  }
  for (int cb = 0; cb < eob; cb += 64) {
    const int c = cb + LANE;
    int cnt = 0, a = 0, neg = 0, len = 0;
    if (c < eob) { const uint32_t m = w->rec_lv[c]; a = (int)(m >> 1); neg = (int)(m & 1); if (a > 14) len = 32 - __clz(a - 14); cnt = (a ? 1 : 0) + (a > 14 ? 2 * len - 1 : 0); }
    int tot; uint32_t at = nrec + (uint32_t)wave_excl_scan_i32(cnt, &tot);
    if (a && at + (uint32_t)cnt <= cap) {
      recs[at++] = c == 0 ? K4_REC(CDF_DC_SIGN + (pt * 3 + dc_ctx) * CDF_DC_SIGN_STRIDE, neg, 2) : K4_BIT(neg);
      if (a > 14) {                                          // Golomb tail: len - 1 zeros, then the len bits of a - 14, a record per bit
        for (int q = 0; q < len - 1; q++) recs[at++] = K4_BIT(0);
        for (int q = len - 1; q >= 0; q--) recs[at++] = K4_BIT(((uint32_t)(a - 14) >> q) & 1u);
      }
    }
    nrec += (uint32_t)tot;
  }
  w->n = nrec;
  K4PH(4); K4CNT(11, eob);
}

// intra_segment_id (spec 5.11.8 / 5.11.9, no pre-skip feature): after the skip flag; a skipped block's id is the prediction and nothing is coded.
// Arguments: the m_skip entries (bit 0 = skip, the rest = segment id) of the block and of its above / left / above-left neighbours.
__device__ __forceinline__ void k4_segment_id(TileWriter *w, int own, int up, int left, int upleft, int availU, int availL) {
  if (!w->seg_n || (own & 1)) return;
  int ctx;
  const int pred = seg_pred(availU && availL ? upleft >> 1 : -1, availU ? up >> 1 : -1, availL ? left >> 1 : -1, &ctx);
  k4_sym(w, seg_symbol(own >> 1, pred, w->seg_n), CDF_SEG_ID + ctx * CDF_SEG_ID_STRIDE, 8);
}

// One transform block's inputs, loaded ahead: while the wave turns block k into records, the loads of block k + 1 -- eob, type, coefficients, and the neighbours'
// level / dc cells its all_zero and dc_sign contexts come from -- are in flight (everything K4 reads was left by K1: nothing here depends on the walk).  A lone wave
// pays every dependent round trip to L2 / HBM in full: staged one after the other they were 39 % of the producer's time (profiles/r04_k4_phase_profile.txt).
template <int BS> struct K4TxPre {
  static constexpr int NC = BS <= 1 ? 1 : (BS == 2 ? 4 : 16);               // coefficients per lane: up to 8x8 / 16x16 / 32x32
  int l_eob, l_txt, la, da, ll, dl; int32_t cv[NC];
  int p, bi, rr, cc, txs;                                                   // which block (wave-uniform)
};
template <int BS> __device__ __forceinline__ void k4_tx_issue(const TileWriter *w, K4TxPre<BS> &o) {
  const FrameDev *f = w->f; const TileB *t = &w->t; const int ms = w->ms, p = o.p, rr = o.rr, cc = o.cc, tmi = rr * ms + cc, k = LANE;
  o.l_eob = f->m_eob[p][tmi]; o.l_txt = f->m_txtype[tmi];
  {                                                                         // txb_ctx_dev's loads (dev_rate.h): a neighbour outside the tile / frame reads the block's own cell
    const bool ha = rr - 1 >= t->mi_row_start && cc + k < w->mi_cols, hl = cc - 1 >= t->mi_col_start && rr + k < w->mi_rows;
    const int ia = (k < 16 && ha) ? (rr - 1) * ms + cc + k : tmi, il = (k < 16 && hl) ? (rr + k) * ms + cc - 1 : tmi;   // (at most 16 cells: the 64x64 transform)
    o.la = f->m_lvl[p][ia]; o.da = f->m_dc[p][ia]; o.ll = f->m_lvl[p][il]; o.dl = f->m_dc[p][il];
  }
  const int l2n = imin_(5, 2 + o.txs), n = 1 << l2n;
  const int32_t *src = f->coef[p] + (size_t)(rr * 4) * f->stride + cc * 4;
#pragma unroll
  for (int q = 0; q < K4TxPre<BS>::NC; q++) { const int idx = LANE + 64 * q; o.cv[q] = idx < n * n ? src[(idx >> l2n) * f->stride + (idx & (n - 1))] : 0; }
}
// all_zero and dc_sign contexts (txb_ctx_dev) from the cells loaded ahead: two sums over the lanes that hold cells (8; 16 for the 64x64 transform of a 64x64
// block) -- the counts of cells above / left that are non-zero, above 3, coded at all, each in its own 5-bit field, and the dc signs
template <int BS> __device__ __forceinline__ void k4_txb_ctx(const TileWriter *w, const K4TxPre<BS> &o, int bs, int *skip_ctx, int *dc_ctx) {
  const TileB *t = &w->t; const int k = LANE, w4 = 1 << o.txs;
  const bool ha = k < w4 && o.rr - 1 >= t->mi_row_start && o.cc + k < w->mi_cols, hl = k < w4 && o.cc - 1 >= t->mi_col_start && o.rr + k < w->mi_rows;
  int pk = 0, dcs = 0;
  if (ha) { pk += (o.la != 0) + ((o.la > 3) << 5) + (((o.la | o.da) != 0) << 20); dcs += o.da == 1 ? -1 : (o.da == 2 ? 1 : 0); }
  if (hl) { pk += ((o.ll != 0) << 10) + ((o.ll > 3) << 15) + (((o.ll | o.dl) != 0) << 25); dcs += o.dl == 1 ? -1 : (o.dl == 2 ? 1 : 0); }
  pk += DPP_(0, pk, 0xB1, 0xF); pk += DPP_(0, pk, 0x4E, 0xF); pk += DPP_(0, pk, 0x141, 0xF);         // quad_perm, quad_perm, row_half_mirror: lanes 0 .. 7
  dcs += DPP_(0, dcs, 0xB1, 0xF); dcs += DPP_(0, dcs, 0x4E, 0xF); dcs += DPP_(0, dcs, 0x141, 0xF);
  if constexpr (BS >= 4) { pk += DPP_(0, pk, 0x140, 0xF); dcs += DPP_(0, dcs, 0x140, 0xF); }           // row_mirror: lanes 0 .. 15
  pk = uni32(pk); dcs = uni32(dcs);
  const int top0 = (pk & 31) == 0, left0 = ((pk >> 10) & 31) == 0, top3 = ((pk >> 5) & 31) != 0, left3 = ((pk >> 15) & 31) != 0;
  *dc_ctx = dcs < 0 ? 1 : (dcs > 0 ? 2 : 0);
  if (o.p == 0) {
    int ctx;
    if (bs == o.txs) ctx = 0;
    else if (top0 && left0) ctx = 1;
    else if (top0 || left0) ctx = 2 + (top3 || left3);
    else if (!(top3 || left3)) ctx = 4;
    else if (!(top3 && left3)) ctx = 5;
    else ctx = 6;
    *skip_ctx = ctx;
  } else {
    *skip_ctx = 7 + (((pk >> 20) & 31) != 0) + (((pk >> 25) & 31) != 0) + (bs > o.txs ? 3 : 0);
  }
}
// the block's next transform block in coding order (per plane the transform blocks in raster order; those outside the frame are skipped); false: none left
template <int BS> __device__ __forceinline__ bool k4_tx_next(const TileWriter *w, int r, int c, int txs_y, int p, int bi, K4TxPre<BS> &o) {
  for (; p < w->np; p++, bi = 0) {
    const int txs = p == 0 ? txs_y : (BS == BS_64 ? 3 : BS) /* chroma transforms stop at 32x32 (spec get_tx_size) */, step = 1 << txs, nblk = 1 << (BS - txs);
    for (; bi < nblk * nblk; bi++) {
      const int rr = r + (bi / nblk) * step, cc = c + (bi % nblk) * step;
      if (rr >= w->mi_rows || cc >= w->mi_cols) continue;
      o.p = p; o.bi = bi; o.rr = rr; o.cc = cc; o.txs = txs;
      return true;
    }
  }
  return false;
}

template <int BS> __device__ __forceinline__ void write_block_dev(TileWriter *w, int r, int c) {
  k4_room(w);
  const FrameDev *f = w->f; const TileB *t = &w->t; const int ms = w->ms, mi = r * ms + c;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  // The block's mode info and its neighbours' in ONE batch of unconditional loads (a neighbour outside the tile reads the block's
  // own cell and is replaced by its default afterwards): as `avail ? map[..] : dflt` at the point of use each of them was a
  // separate round trip to L2 on the single wave that codes the tile.
  const int iU = availU ? mi - ms : mi, iL = availL ? mi - 1 : mi;
  const int l_skip = f->m_skip[mi], l_ymode = f->m_ymode[mi], l_txs_y = f->m_txsize[mi];
  const int l_skU = f->m_skip[iU], l_skL = f->m_skip[iL], l_ymU = f->m_ymode[iU], l_ymL = f->m_ymode[iL], l_txU = f->m_txsize[iU], l_txL = f->m_txsize[iL];
  const int l_skUL = f->m_skip[availU && availL ? mi - ms - 1 : mi];
  const int l_ay = f->m_angle_y[mi], l_cdef = f->cdef_idx[(r >> 4) * w->sb_cols + (c >> 4)];
  int l_uvmode = 0, l_auv = 0, l_js = 0, l_au = 0, l_av = 0;
  if (w->np > 1) { l_uvmode = f->m_uvmode[mi]; l_auv = f->m_angle_uv[mi]; l_js = f->m_cfl_sign[mi]; l_au = f->m_cfl_au[mi]; l_av = f->m_cfl_av[mi]; }
  const int sk_seg = U_(l_skip), skip = sk_seg & 1, ymode = U_(l_ymode), txs_y = U_(l_txs_y), v_skU = U_(l_skU), v_skL = U_(l_skL), v_ymU = U_(l_ymU), v_ymL = U_(l_ymL);   // m_skip: bit 0 = skip, the rest = segment id
  const int v_txU = U_(l_txU), v_txL = U_(l_txL), v_ay = U_(l_ay), v_cdef = U_(l_cdef);
  const int uvmode = U_(l_uvmode), v_auv = U_(l_auv), v_js = U_(l_js), v_au = U_(l_au), v_av = U_(l_av);
  {
    const int cdf = 0;                                   // CDF rows are named by their offset in the tables
    const int sctx = (availU ? v_skU & 1 : 0) + (availL ? v_skL & 1 : 0);
    k4_sym(w, skip, cdf + CDF_SKIP + sctx * CDF_SKIP_STRIDE, 2);
    k4_segment_id(w, sk_seg, v_skU, v_skL, U_(l_skUL), availU, availL);
    if (!skip && w->enable_cdef) {
      if (w->cdef_pending) { w->cdef_pending = 0; k4_lit(w, (uint32_t)v_cdef, w->cdef_bits); }   // first non-skip block of the superblock (spec 5.11.56)
    }
    const int am = intra_mode_ctx(availU ? v_ymU : DC_PRED), lm = intra_mode_ctx(availL ? v_ymL : DC_PRED);
    k4_sym(w, ymode, cdf + CDF_KF_Y + (am * 5 + lm) * CDF_KF_Y_STRIDE, 13);
    if (BS >= BS_8 && ymode >= V_PRED && ymode <= D67_PRED)
      k4_sym(w, v_ay + 3, cdf + CDF_ANGLE + (ymode - V_PRED) * CDF_ANGLE_STRIDE, 7);
    if (w->np > 1) {
      const int um = uvmode;
      if (BS <= BS_32) k4_sym(w, um, cdf + CDF_UV_CFL + ymode * CDF_UV_CFL_STRIDE, 14);
      else k4_sym(w, um, cdf + CDF_UV_NOCFL + ymode * CDF_UV_NOCFL_STRIDE, 13);
      if (um == UV_CFL_PRED) {
        const int js = v_js, su = (js + 1) / 3, sv = (js + 1) % 3;
        k4_sym(w, js, cdf + CDF_CFL_SIGN, 8);
        if (su) k4_sym(w, v_au, cdf + CDF_CFL_ALPHA + ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE, 16);
        if (sv) k4_sym(w, v_av, cdf + CDF_CFL_ALPHA + ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE, 16);
      }
      if (BS >= BS_8 && um >= V_PRED && um <= D67_PRED)
        k4_sym(w, v_auv + 3, cdf + CDF_ANGLE + (um - V_PRED) * CDF_ANGLE_STRIDE, 7);
    }
  }
  // read_block_tx_size(): tx_depth of every intra block above 4x4 under TX_MODE_SELECT, coded even when skip
  if (BS > 0 && w->tx_mode_select) {
    const int maxw = 4 << BS;
    const int actx = availU && (1 << dim_wl(v_txU)) >= maxw, lctx = availL && (1 << dim_hl(v_txL)) >= maxw;
    k4_sym(w, BS - txs_y, CDF_TX_SIZE + ((BS - 1) * 3 + actx + lctx) * CDF_TX_SIZE_STRIDE, BS == 1 ? 2 : 3);
  }
  K4PH(1); K4CNT(8, 1);
  if (skip) return;                                    // wave-uniform
  // residual(): per plane the transform blocks of the block in raster order (luma may be split one level, chroma is not), each one's loads a block ahead
  K4TxPre<BS> cur;
  if (!k4_tx_next<BS>(w, r, c, txs_y, 0, 0, cur)) return;
  k4_tx_issue<BS>(w, cur);
  for (;;) {
    K4TxPre<BS> nxt;
    const bool more = k4_tx_next<BS>(w, r, c, txs_y, cur.p, cur.bi + 1, nxt);
    if (more) k4_tx_issue<BS>(w, nxt);
    {
      const int p = cur.p, txs = cur.txs, l2n = imin_(5, 2 + txs), n = 1 << l2n;
      WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < K4TxPre<BS>::NC; q++) { const int idx = LANE + 64 * q; if (idx < n * n) w->qc[idx] = cur.cv[q]; }
      WAVE_SYNC();
      build_level_map(w->qc, w->lev, n);
      const int eob = U_(cur.l_eob), v_txt = U_(cur.l_txt);
      int txtype, off = -1, sym = 0, ns = 0, set;
      if (p == 0) {
        txtype = v_txt;
        off = intra_tx_cdf(&w->txc, txs, ymode, &ns, &set);
        if (off >= 0) sym = txtype_to_sym(set, txtype);
      } else {
        set = tx_set_of(txs, w->txc.reduced_tx_set);
        txtype = mode_to_txtype(uvmode);
        if (txtype_to_sym(set, txtype) < 0) txtype = DCT_DCT;
      }
      int sctx2, dctx;
      k4_txb_ctx<BS>(w, cur, BS, &sctx2, &dctx);
      K4PH(2);
      code_coeffs_lane0(w, eob, p, txs, txtype, sctx2, dctx, off, sym, ns);
    }
    if (!more) break;
    cur = nxt;
  }
  WAVE_SYNC();
}

// An 8x4 / 4x8 block (oracle write_block with a 2:1 size): no angle deltas, CfL allowed, tx_depth in the 8x8 category, one 2:1 transform or its two
// 4x4 halves per luma block, one 2:1 transform per chroma plane.
template <int BSR> __device__ __forceinline__ void write_block_rect(TileWriter *w, int r, int c) {
  constexpr int WL = BSR == BS_4X8 ? 2 : 3, HL = BSR == BS_4X8 ? 3 : 2, W_ = 1 << WL, H_ = 1 << HL;
  k4_room(w);
  const FrameDev *f = w->f; const TileB *t = &w->t; const int ms = w->ms, mi = r * ms + c;
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int iU = availU ? mi - ms : mi, iL = availL ? mi - 1 : mi;
  const int l_skip = f->m_skip[mi], l_ymode = f->m_ymode[mi], l_txs_y = f->m_txsize[mi];
  const int l_skU = f->m_skip[iU], l_skL = f->m_skip[iL], l_ymU = f->m_ymode[iU], l_ymL = f->m_ymode[iL], l_txU = f->m_txsize[iU], l_txL = f->m_txsize[iL];
  const int l_cdef = f->cdef_idx[(r >> 4) * w->sb_cols + (c >> 4)], l_skUL = f->m_skip[availU && availL ? mi - ms - 1 : mi];
  int l_uvmode = 0, l_js = 0, l_au = 0, l_av = 0;
  if (w->np > 1) { l_uvmode = f->m_uvmode[mi]; l_js = f->m_cfl_sign[mi]; l_au = f->m_cfl_au[mi]; l_av = f->m_cfl_av[mi]; }
  const int sk_seg = U_(l_skip), skip = sk_seg & 1, ymode = U_(l_ymode), txs_y = U_(l_txs_y), v_skU = U_(l_skU), v_skL = U_(l_skL), v_ymU = U_(l_ymU), v_ymL = U_(l_ymL);
  const int v_txU = U_(l_txU), v_txL = U_(l_txL), v_cdef = U_(l_cdef), uvmode = U_(l_uvmode), v_js = U_(l_js), v_au = U_(l_au), v_av = U_(l_av);
  const int cdf = 0;
  k4_sym(w, skip, cdf + CDF_SKIP + ((availU ? v_skU & 1 : 0) + (availL ? v_skL & 1 : 0)) * CDF_SKIP_STRIDE, 2);
  k4_segment_id(w, sk_seg, v_skU, v_skL, U_(l_skUL), availU, availL);
  if (!skip && w->enable_cdef && w->cdef_pending) { w->cdef_pending = 0; k4_lit(w, (uint32_t)v_cdef, w->cdef_bits); }
  const int am = intra_mode_ctx(availU ? v_ymU : DC_PRED), lm = intra_mode_ctx(availL ? v_ymL : DC_PRED);
  k4_sym(w, ymode, cdf + CDF_KF_Y + (am * 5 + lm) * CDF_KF_Y_STRIDE, 13);
  if (w->np > 1) {
    k4_sym(w, uvmode, cdf + CDF_UV_CFL + ymode * CDF_UV_CFL_STRIDE, 14);
    if (uvmode == UV_CFL_PRED) {
      const int js = v_js, su = (js + 1) / 3, sv = (js + 1) % 3;
      k4_sym(w, js, cdf + CDF_CFL_SIGN, 8);
      if (su) k4_sym(w, v_au, cdf + CDF_CFL_ALPHA + ((su - 1) * 3 + sv) * CDF_CFL_ALPHA_STRIDE, 16);
      if (sv) k4_sym(w, v_av, cdf + CDF_CFL_ALPHA + ((sv - 1) * 3 + su) * CDF_CFL_ALPHA_STRIDE, 16);
    }
  }
  if (w->tx_mode_select) {
    const int actx = availU && dim_wl(v_txU) >= WL, lctx = availL && dim_hl(v_txL) >= HL;
    k4_sym(w, txs_y == BSR ? 0 : 1, cdf + CDF_TX_SIZE + (actx + lctx) * CDF_TX_SIZE_STRIDE, 2);
  }
  if (skip) return;
  for (int p = 0; p < w->np; p++) {
    const int txs = p == 0 ? txs_y : BSR, split = txs != BSR;                       // luma may be two 4x4 transforms
    for (int bi = 0; bi < (split ? 2 : 1); bi++) {
      const int rr = r + ((split && H_ == 8) ? bi : 0), cc = c + ((split && W_ == 8) ? bi : 0), tmi = rr * ms + cc;
      if (rr >= w->mi_rows || cc >= w->mi_cols) continue;
      const int l_eob = f->m_eob[p][tmi], l_txt = f->m_txtype[tmi];
      const int tw = split ? 4 : W_, th = split ? 4 : H_, twl = split ? 2 : WL;
      const int32_t *src = f->coef[p] + (size_t)(rr * 4) * f->stride + cc * 4;
      WAVE_SYNC();
      for (int idx = LANE; idx < tw * th; idx += 64) w->qc[idx] = src[(idx >> twl) * f->stride + (idx & (tw - 1))];
      WAVE_SYNC();
      {                                                                            // level map with its 4-wide zero border (build_level_map for tw x th)
        const int st = tw + 4;
        for (int i = LANE; i < st * (th + 4); i += 64) w->lev[i] = 0;
        WAVE_SYNC();
        for (int i = LANE; i < tw * th; i += 64) w->lev[(i >> twl) * st + (i & (tw - 1))] = (uint8_t)imin_(iabs_(w->qc[i]), 127);
        WAVE_SYNC();
      }
      const int eob = U_(l_eob), v_txt = U_(l_txt);
      int txtype, off = -1, sym = 0, ns = 0, set;
      if (p == 0) {
        txtype = v_txt;
        off = split ? intra_tx_cdf(&w->txc, 0, ymode, &ns, &set) : rect_tx_cdf<0>(&w->txc, ymode, &ns, &set);
        if (off >= 0) sym = txtype_to_sym(set, txtype);
      } else {
        set = w->txc.reduced_tx_set ? 2 : 1;
        txtype = mode_to_txtype(uvmode);
        if (txtype_to_sym(set, txtype) < 0) txtype = DCT_DCT;
      }
      int sctx2, dctx;
      txb_ctx_wh(f, t, p, rr, cc, tw >> 2, th >> 2, !split, &sctx2, &dctx);
      code_coeffs_lane0(w, eob, p, split ? 0 : BSR, txtype, sctx2, dctx, off, sym, ns);
    }
  }
  WAVE_SYNC();
}

// Partition symbol of the node (r, c, bs >= 1); returns 0 (NONE) or 3 (SPLIT).  spec 5.11.4
__device__ __forceinline__ int write_partition_symbol(TileWriter *w, int r, int c, int bs) {
  const FrameDev *f = w->f; const TileB *t = &w->t; const int ms = w->ms;
  const int half = (1 << bs) >> 1;
  const int has_rows = (r + half) < w->mi_rows, has_cols = (c + half) < w->mi_cols;
  const int actual = U_(f->m_bsize[r * ms + c]);
  int part = actual == bs ? 0 : ((bs == BS_8 && actual == BS_8X4) ? 1 : ((bs == BS_8 && actual == BS_4X8) ? 2 : 3));
  const int availU = r > t->mi_row_start, availL = c > t->mi_col_start;
  const int above = availU && dim_wl(U_(f->m_bsize[(r - 1) * ms + c])) < 2 + bs, left = availL && dim_hl(U_(f->m_bsize[r * ms + c - 1])) < 2 + bs;
  const int cdf = CDF_PARTITION + ((bs - 1) * 4 + left * 2 + above) * CDF_PARTITION_STRIDE;
  const int ns = bs == BS_8 ? 4 : 10;
  if (has_rows && has_cols) k4_sym(w, part, cdf, ns);
  else if (has_rows || has_cols) k4_put(w, K4_PEDGE(uni32(cdf), has_cols));   // the bool's probability comes from the row's state at this point of the stream: its adapter's job
  if (!(has_rows && has_cols)) part = 3;
  K4PH(0);
  return part;
}

// read_lr() of the superblock at (r, c) (spec 5.11.57 / 5.11.58): with 64x64 units at most one unit per plane
__device__ __forceinline__ void write_lr_sb(TileWriter *w, int r, int c) {
  const FrameDev *f = w->f;
  if (!w->enable_restoration) return;
  const int ucols = lr_units_of(w->fw), urows = lr_units_of(w->fh), n = ucols * urows;
  const int urs = (r * 4 + 63) / 64, ure = imin_(((r + 16) * 4 + 63) / 64, urows);
  const int ucs = (c * 4 + 63) / 64, uce = imin_(((c + 16) * 4 + 63) / 64, ucols);
  for (int p = 0; p < w->np; p++) for (int ur = urs; ur < ure; ur++) for (int uc = ucs; uc < uce; uc++) {
    const int ui = p * n + ur * ucols + uc;
    const int type = U_(f->lr_type[ui]);
    k4_sym(w, type ? 2 : 0, CDF_LR_SWITCHABLE, 3);
    if (!type) continue;
    const int set = U_(f->lr_set[ui]);
    k4_lit(w, (uint32_t)set, 4);
    int r0, s0, r1, s1; sgr_param(set, &r0, &s0, &r1, &s1);
    for (int i = 0; i < 2; i++) {
      const int v = U_(f->lr_xqd[ui * 2 + i]);
      if (i == 0 ? r0 : r1) {
        uint32_t bits; const int nb = lr_subexp_code(v, i == 0 ? -96 : -32, i == 0 ? 32 : 96, w->lr_ref[p * 2 + i], &bits);
        k4_lit(w, bits, nb);
      }
      WAVE_SYNC();
      if (LANE == 0) w->lr_ref[p * 2 + i] = v;
      WAVE_SYNC();
    }
  }
}

// Iterative Z-order walk of one superblock (explicit stack, depth <= 5) so that every block-size instance of
// write_block_dev is inlined exactly once and the range-coder state stays in registers.
template <int MAXBS> __device__ __forceinline__ void write_superblock(TileWriter *w, int r0, int c0) {
  const FrameDev *f = w->f;
  w->cdef_pending = 1;
  K4PH(7);
  write_lr_sb(w, r0, c0);
  K4PH(6);
  int sr[5], sc[5], sk[5];
  int sp = 0; sr[0] = r0; sc[0] = c0; sk[0] = 0;
  while (sp >= 0) {
    const int bs = 4 - sp;
    int r = 0, c = 0, kk = 0;
#pragma unroll
    for (int q = 0; q < 5; q++) if (q == sp) { r = sr[q]; c = sc[q]; kk = sk[q]; }
    if (kk == 0) {
      if (r >= w->mi_rows || c >= w->mi_cols) { sp--; continue; }
      const int part = bs == 0 ? 0 : write_partition_symbol(w, r, c, bs);
      if (part == 0) {
        switch (bs) {
          case 0: write_block_dev<0>(w, r, c); break;
          case 1: write_block_dev<1>(w, r, c); break;
          case 2: write_block_dev<2>(w, r, c); break;
          case 3: if constexpr (MAXBS >= 3) write_block_dev<3>(w, r, c); break;
          default: if constexpr (MAXBS >= 4) write_block_dev<4>(w, r, c); break;
        }
        sp--; continue;
      }
      if (part == 1) { write_block_rect<BS_8X4>(w, r, c); write_block_rect<BS_8X4>(w, r + 1, c); sp--; continue; }
      if (part == 2) { write_block_rect<BS_4X8>(w, r, c); write_block_rect<BS_4X8>(w, r, c + 1); sp--; continue; }
    }
    if (kk == 4) { sp--; continue; }
    const int half = (1 << bs) >> 1;
#pragma unroll
    for (int q = 0; q < 5; q++) if (q == sp) sk[q] = kk + 1;
    const int nr = r + (kk >> 1) * half, ncol = c + (kk & 1) * half;
#pragma unroll
    for (int q = 0; q < 5; q++) if (q == sp + 1) { sr[q] = nr; sc[q] = ncol; sk[q] = 0; }
    sp++;
  }
}

// LDS sized by the largest block the launch can meet (MAXBS 2: 16x16 -> 15.6 KB; MAXBS 4: 32x32 coded coefficients -> 28 KB); dynamic, so
// that the compiler does not pad the register allocation to the occupancy a compile-time LDS size implies
template <int CS> struct EntropyLds {
  uint16_t cdf[CDF_TOTAL + 8];            // the tables, then the switchable restoration_type row (CDF_LR_SWITCHABLE)
  int32_t qc[CS * CS];
  uint8_t lev[(CS + 4) * (CS + 4) + 4];
  uint16_t scans[SCAN_LDS_ENTRIES(CS)];
  uint16_t rec_off[CS * CS], rec_br[CS * CS];
  uint32_t rec_lv[CS * CS];
  int lr_ref[6];
  uint32_t acc[MI_K4_ACC];                  // the coder's ring of byte accumulators (RangeEncDev)
  uint32_t nrec[3];                        // records in the three rotating buffers
  int total;                               // buffers of the tile: -1 until the producer has handed on its last one
};

// recbuf: per tile job three record buffers of rec_cap entries (producer -> adapters -> coder, rotating per superblock)
template <int MAXBS, int NA>
__global__ __launch_bounds__(MI_K4_THREADS_OF(NA) MI_K4_LB_EXTRA) void tile_entropy_kernel(const FrameDev *__restrict__ frames, const TileJob *__restrict__ jobs, int njobs, uint16_t *precarry, uint32_t pre_cap,
                                                                   uint32_t *recbuf, uint32_t rec_cap) {
  constexpr int CS = MAXBS <= 2 ? 16 : 32, MI_K4_THREADS = MI_K4_THREADS_OF(NA);
  extern __shared__ __align__(16) uint8_t k4_smem[];            // sizeof(EntropyLds<CS>), passed at launch
  EntropyLds<CS> &L = *(EntropyLds<CS> *)k4_smem;
  const int job = blockIdx.x;
  if (job >= njobs) return;
  const TileJob tj = jobs[job];
  const FrameDev *f = frames + tj.frame;
  const int tile = tj.tile_row * f->tile_cols + tj.tile_col;
  if (frame_idle(f)) { if (threadIdx.x == 0) f->tile_len[tile] = 0; return; }
  const int wave = uni32((int)(threadIdx.x >> 6));             // 0 producer, 1 .. NA adapters, NA + 1 coder
  // Issue priorities by stage: a tile lasts as long as its busiest stage, and every SIMD hosts one stage of each of four tiles.  The last adapter (the busiest stage:
  // 24.1 M busy cycles per tile against the producer's 15.9 M, profiles/r04_k4_phase_profile.txt) goes first, then the other adapters, the coder, the producer:
  // 20.3 -> 17.3 ms on the 1024-tile batch (profiles/r05zj_ab_k4prio2.txt; adapters first and equal 18.0, coder first 19.5).
  if (wave == NA) __builtin_amdgcn_s_setprio(3); else if (wave >= 1 && wave < NA) __builtin_amdgcn_s_setprio(2); else if (wave == NA + 1) __builtin_amdgcn_s_setprio(1);
  uint32_t *const bufs = recbuf + (size_t)job * 3 * rec_cap;
  const int row0 = f->tile_row_start[tj.tile_row] * 16, row1 = imin_(f->tile_row_start[tj.tile_row + 1] * 16, f->mi_rows);
  const int col0 = f->tile_col_start[tj.tile_col] * 16, col1 = imin_(f->tile_col_start[tj.tile_col + 1] * 16, f->mi_cols);
  const int sbc = uni32((col1 - col0 + 15) >> 4), nsb = uni32(((row1 - row0 + 15) >> 4) * sbc);
  // the tables: every wave a share
  for (int i = threadIdx.x; i < CDF_TOTAL; i += MI_K4_THREADS) L.cdf[i] = f->cdf0[i];
  if (threadIdx.x < 4) L.cdf[CDF_LR_SWITCHABLE + threadIdx.x] = (uint16_t)(threadIdx.x == 0 ? 32768 - 9413 : (threadIdx.x == 1 ? 32768 - 22581 : 0));   // libaom default_switchable_restore_cdf
  if (threadIdx.x < 6) L.lr_ref[threadIdx.x] = (threadIdx.x & 1) ? 31 : -32;                                                                              // Sgrproj_Xqd_Mid
  TileWriter w;
  RangeEncDev ec;
  const unsigned long long clk0 = wall_clock64();
#if MI_PROFILE == 2
  unsigned long long busy = 0;
#endif
  if (wave == 0) {
    w.f = f;
    w.t.mi_row_start = row0; w.t.mi_row_end = row1; w.t.mi_col_start = col0; w.t.mi_col_end = col1;
    w.qc = (LDS int32_t *)L.qc; w.lev = (LDS uint8_t *)L.lev; w.cdef_pending = 1; w.ls = (LDS uint16_t *)L.scans;
    w.rec_off = (LDS uint16_t *)L.rec_off; w.rec_br = (LDS uint16_t *)L.rec_br; w.rec_lv = (LDS uint32_t *)L.rec_lv;
    w.lr_ref = (LDS int *)L.lr_ref; w.cap = rec_cap; w.out = bufs; w.n = 0; w.cdf = (LDS uint16_t *)L.cdf; w.own_rows = NA == 2 && MI_K4_SPREAD;
    w.np = U_(f->np); w.mi_rows = U_(f->mi_rows); w.mi_cols = U_(f->mi_cols); w.ms = U_(f->mi_stride); w.tx_mode_select = U_(f->tx_mode_select);
    w.seg_n = U_(f->seg_n); w.enable_cdef = U_(f->enable_cdef); w.cdef_bits = U_(f->cdef_bits); w.enable_restoration = U_(f->enable_restoration); w.sb_cols = U_(f->sb_cols);
    w.fw = U_(f->w); w.fh = U_(f->h); w.txc.reduced_tx_set = U_(f->reduced_tx_set); w.txc.base_q_idx = U_(f->base_q_idx);
    load_scans_to_lds((LDS uint16_t *)L.scans, CS);
    w.sb_cols_tile = sbc;
#if MI_PROFILE == 2
    for (int i = 0; i < 16; i++) w.prof[i] = 0;
    w.pt = clock64();
#endif
  } else if (wave == NA + 1) {
    for (int i = LANE; i < MI_K4_ACC; i += 64) L.acc[i] = 0;
    re_init_dev(&ec, precarry + (size_t)job * pre_cap, pre_cap, (LDS uint32_t *)L.acc);
  }
  if (threadIdx.x == 0) L.total = -1;
  __syncthreads();
  int overflow = 0;
  // Stage loop.  Step t: the producer fills buffer t, the adapters work on buffer t - 1, the coder on buffer t - 2; a barrier per step hands them on.
  // A buffer is one superblock's records, or a piece of it when the superblock needs more than a buffer holds (k4_room): the producer meets the
  // barrier from inside its walk then, the other stages from here -- they run until two steps after the tile's last buffer.
  if (wave == 0) {
    w.bufs = bufs; w.nrec = (LDS uint32_t *)L.nrec; w.total = (LDS int *)&L.total; w.chunk = 0;
    for (int sb = 0; sb < nsb; sb++) {
#if MI_PROFILE == 2
      const unsigned long long t0_ = clock64(); w.pt = t0_;
#endif
      write_superblock<MAXBS>(&w, row0 + 16 * (sb / sbc), col0 + 16 * (sb % sbc));
#if MI_PROFILE == 2
      busy += clock64() - t0_;
#endif
      k4_handoff(&w, sb == nsb - 1);
    }
    __syncthreads(); __syncthreads();                         // the last buffer's two further steps
  } else {
    for (int t = 0;; t++) {
#if MI_PROFILE == 2
      const unsigned long long t0_ = clock64();
#endif
      const int total = L.total;                              // (any value read while the producer is still writing it gives the same decisions below)
      if (wave <= NA) {
        if (t >= 1 && (total < 0 || t - 1 < total)) k4_adapt_sb<NA>((LDS uint16_t *)L.cdf, bufs + (size_t)((t - 1) % 3) * rec_cap, (int)imin_((int)L.nrec[(t - 1) % 3], (int)rec_cap), wave - 1);
      } else {
        if (t >= 2 && (total < 0 || t - 2 < total)) {
          const uint32_t n = L.nrec[(t - 2) % 3]; if (n > rec_cap) overflow = 1;   // (k4_room keeps n below the capacity: a guard, not a path)
          uint32_t *const buf = bufs + (size_t)((t - 2) % 3) * rec_cap; const int nc = (int)imin_((int)n, (int)rec_cap);
          if constexpr (NA == 2 && MI_K4_SPREAD) {                // the coder's own rows (owner 3): symbols -> bounds right before they are coded
            k4_adapt_sb<NA>((LDS uint16_t *)L.cdf, buf, nc, 3);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); WAVE_SYNC();
          }
          k4_code_sb(&ec, buf, nc);
        }
      }
#if MI_PROFILE == 2
      busy += clock64() - t0_;
#endif
      __syncthreads();
      const int total2 = L.total;
      if (total2 >= 0 && t >= total2 + 1) break;
    }
  }
  if (wave == 0 && f->cdf_out != nullptr)                       // two-pass pricing: what this tile's CDFs have adapted to (every adapter is past the last barrier)
    for (int i = LANE; i < CDF_TOTAL; i += 64) f->cdf_out[(size_t)tile * CDF_TOTAL + i] = L.cdf[i];
  if (wave == NA + 1) {
    uint8_t *out = f->tile_out + (size_t)tile * f->tile_out_cap;
    uint32_t out_len = re_finish_dev(&ec, out, f->tile_out_cap);       // whole wave
    if (overflow || *search_error_word(f) != 0) out_len = 0xFFFFFFFFu;     // (a tile search that gave up waiting: tile_search.h root_wait)
    if (LANE == 0) {
      f->tile_len[tile] = out_len;
      unsigned long long *tc = f->tile_clk + (size_t)tile * 4; tc[2] = clk0; tc[3] = wall_clock64();
    }
  }
#if MI_PROFILE == 2
  if (LANE == 0 && f->prof_out) {
    if (wave == 0) { TileWriter *w_ = &w; TileWriter *w = w_; for (int i = 0; i < 16; i++) f->prof_out[(size_t)job * 128 + 96 + i] = w->prof[i]; }
    f->prof_out[(size_t)job * 128 + 64 + wave] = busy;          // busy cycles of every stage wave (barrier waits excluded)
  }
#endif
}

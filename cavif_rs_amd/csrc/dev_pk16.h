// dev_pk16.h -- two 16-bit samples per lane: the packed-math forms (v_pk_sub_i16, v_pk_max_i16, v_pk_min_i16, v_pk_ashrrev_i16, v_pk_mul_lo_u16,
// v_dot2_i32_i16) of gfx950 for arithmetic on 10-bit samples and their differences, which fit 16 bits with room to spare.  A packed instruction costs one issue slot
// like most 32-bit integer VALU instructions (tools/probe/valu_rates.hip) and does two samples' work.  Used by the CDEF strength search (loopfilter.h).
#pragma once
#include <stdint.h>
#ifndef MI_PK16_DEFINED                              /* (the CPU test harness tests/emu/ supplies a struct with the same operators and these functions) */
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_splat(int v) { return pk16{ (short)v, (short)v }; }
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ pk16 pk_min(pk16 a, pk16 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ pk16 pk_from_u32(uint32_t v) { return __builtin_bit_cast(pk16, v); }       // low half = the sample at the lower address
__device__ __forceinline__ uint32_t pk_to_u32(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ int pk_dot2(pk16 a, pk16 b, int acc) { return __builtin_amdgcn_sdot2(a, b, acc, false); }   // acc + a.x * b.x + a.y * b.y
#endif
__device__ __forceinline__ pk16 pk_abs(pk16 a) { return pk_max(a, -a); }
// two adjacent 16-bit samples at any 2-byte-aligned address (global memory takes unaligned dword loads)
__device__ __forceinline__ pk16 pk_load2(const uint16_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return pk_from_u32(v); }

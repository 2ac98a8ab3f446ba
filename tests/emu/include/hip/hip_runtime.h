// hip_runtime.h (SIMT emulator shim) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Lets g++ compile cavif_rs_amd/csrc/mi_avif.hip -- the UNCHANGED product sources, host engine and HIP kernels -- into
// tests/emu/_build/libmi_avif_emu.so, in which every GPU thread is a fiber on the CPU: a workgroup's threads run one after
// another between the points where the GPU makes them meet (cross-lane operations: v_readlane / DPP / v_readfirstlane /
// __shfl; the wavefront-scope WAVE_SYNC; __syncthreads), LDS is a per-workgroup array, HBM is host memory and the HIP runtime
// calls are synchronous stand-ins.  Purpose: `-m "not gpu"` tests can run the very kernels that ship against the CPU oracle on
// small images when no MI355X is attached, so that a kernel edit is checked for byte parity before it costs GPU time.
// It is 10^3..10^4 times slower than one host core running the oracle; it is never linked into cavif_rs_amd/libmi_avif.so,
// which still fails with MI_NO_DEVICE without a GPU.  Cross-lane semantics follow the CDNA3/4 ISA (DPP controls quad_perm,
// row_shl/shr/ror, row_mirror, row_half_mirror, row_bcast15/31 with row / bank masks and bound_ctrl).
// Lanes that skip a divergent region wait at the next cross-lane operation they reach; an operation is resolved for the lanes
// standing at the same source-level site as soon as every lane they read from stands there too (the kernels keep whole quads /
// rows active around partial-wave DPP), so a region's lanes catch up with the ones waiting behind it.
// What it cannot show: timing, register pressure, and bugs that depend on true lockstep execution inside a wavefront.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <functional>

#define MI_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local          /* one workgroup at a time per OS thread: its fibers share the thread's statics */
#define __align__(n) alignas(n)
#define LDS                              /* dev_common.h: address_space(3) on the GPU */

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
namespace emu {
struct Idx3 { unsigned x, y, z; };
enum XKind { X_READLANE = 1, X_READFIRST, X_DPP, X_SHFL };
int xlane(int kind, int val, int old, int p0, int rm, int bm, int bc, int site);
void wave_barrier();
void wg_barrier();
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);
unsigned long long ticks();
void spin_yield();
}
extern thread_local emu::Idx3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
// dynamic LDS of the two kernels that use it (tile_search.h `smem`, tile_entropy.h `k4_smem`)
extern thread_local alignas(16) uint8_t smem[];
extern thread_local alignas(16) uint8_t k4_smem[];

// ---- device intrinsics ----
#define __builtin_amdgcn_readlane(v, lane) emu::xlane(emu::X_READLANE, (int)(v), 0, (int)(lane), 0, 0, 0, __COUNTER__)
#define __builtin_amdgcn_readfirstlane(v) emu::xlane(emu::X_READFIRST, (int)(v), 0, 0, 0, 0, 0, __COUNTER__)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu::xlane(emu::X_DPP, (int)(src), (int)(old), (int)(ctrl), (int)(rm), (int)(bm), (int)(bc), __COUNTER__)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __threadfence() {}
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_s_sleep(n) emu::spin_yield()      /* a workgroup waiting for another one: let its OS thread run */
inline int __shfl(int v, int src, int width = 64) { (void)width; return emu::xlane(emu::X_SHFL, v, 0, src, 0, 0, 0, -1); }
inline void __syncthreads() { emu::wg_barrier(); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline int __clz(unsigned v) { return v == 0 ? 32 : __builtin_clz(v); }
inline int __mul24(int a, int b) { return (int)((uint32_t)((a << 8) >> 8) * (uint32_t)((b << 8) >> 8)); }
inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }             /* dev_common.h: raw v_sqrt_f32 / v_rcp_f32 guesses */
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline unsigned long long clock64() { return emu::ticks(); }
inline unsigned long long wall_clock64() { return emu::ticks(); }
// global / LDS atomics: workgroups of one launch may run on several OS threads
template <typename T> inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_load(ptr, order, scope) __atomic_load_n((ptr), (order))
#define __hip_atomic_store(ptr, v, order, scope) __atomic_store_n((ptr), (v), (order))
#define __hip_atomic_fetch_add(ptr, v, order, scope) __atomic_fetch_add((ptr), (v), (order))
// MI_EMU_DROP_PUBLISH=1 (tests/test_emu_kernels.py): the tile search's root bits are never published, so every dependency wait runs into its poll bound -- the
// failure path (sticky error word -> every tile of the frame fails -> MI_ENCODING_ERROR) without touching the product code
inline bool emu_drop_publish() { static const bool on = getenv("MI_EMU_DROP_PUBLISH") != nullptr; return on; }
template <typename T> inline T emu_fetch_or(T *p, T v, int order) { return emu_drop_publish() ? __atomic_load_n(p, order == __ATOMIC_RELEASE ? __ATOMIC_ACQUIRE : order) : __atomic_fetch_or(p, v, order); }
#define __hip_atomic_fetch_or(ptr, v, order, scope) emu_fetch_or((ptr), (__typeof__(*(ptr)))(v), (order))
// wave ballot from the cross-lane shuffle the emulator has (butterfly OR of the lanes' own bits)
inline unsigned long long emu_ballot64(bool p) {
  const int l = (int)(threadIdx.x & 63);
  int lo = (p && l < 32) ? (int)(1u << l) : 0, hi = (p && l >= 32) ? (int)(1u << (l - 32)) : 0;
  for (int d = 1; d < 64; d <<= 1) { const int a = __shfl(lo, l ^ d), b = __shfl(hi, l ^ d); lo |= a; hi |= b; }
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
#define MI_BALLOT64(p) emu_ballot64(p)
#define MI_SMUL32(r, a, b) ((r) = (a) * (b))
#define MI_BITSET0_64(m, j) ((m) &= ~(1ull << (j)))
#define MI_SSEL_GE(r, a, b, x, y) ((r) = (a) >= (b) ? (x) : (y))
#define MI_WRITELANE(v, s, lane) ((v) = ((int)(threadIdx.x & 63) == (lane)) ? (int)(s) : (v))     /* tile_entropy.h: v_writelane_b32 */
#define MI_MAD24(r, x, c, acc) ((r) = __mul24((x), (c)) + (acc))     /* txfm_gen.hip.h: v_mad_i32_i24 */
// dev_pk16.h: two 16-bit lanes in a struct with the operators the device code uses on its clang vector type
#define MI_PK16_DEFINED
struct pk16 { short x, y; };
inline pk16 operator-(pk16 a) { return pk16{ (short)-a.x, (short)-a.y }; }
inline pk16 operator-(pk16 a, pk16 b) { return pk16{ (short)(a.x - b.x), (short)(a.y - b.y) }; }
inline pk16 operator+(pk16 a, pk16 b) { return pk16{ (short)(a.x + b.x), (short)(a.y + b.y) }; }
inline pk16 operator*(pk16 a, pk16 b) { return pk16{ (short)(a.x * b.x), (short)(a.y * b.y) }; }
inline pk16 operator^(pk16 a, pk16 b) { return pk16{ (short)(a.x ^ b.x), (short)(a.y ^ b.y) }; }
inline pk16 operator>>(pk16 a, pk16 b) { return pk16{ (short)(a.x >> b.x), (short)(a.y >> b.y) }; }
inline pk16 &operator+=(pk16 &a, pk16 b) { a = a + b; return a; }
inline pk16 pk_splat(int v) { return pk16{ (short)v, (short)v }; }
inline pk16 pk_max(pk16 a, pk16 b) { return pk16{ a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y }; }
inline pk16 pk_min(pk16 a, pk16 b) { return pk16{ a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y }; }
inline pk16 pk_from_u32(uint32_t v) { return pk16{ (short)(v & 0xFFFF), (short)(v >> 16) }; }
inline uint32_t pk_to_u32(pk16 v) { return (uint32_t)(uint16_t)v.x | ((uint32_t)(uint16_t)v.y << 16); }
inline int pk_dot2(pk16 a, pk16 b, int acc) { return acc + a.x * b.x + a.y * b.y; }
struct uchar4 { unsigned char x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{ x, y, z, w }; }

// ---- HIP runtime stand-ins: one "device", synchronous streams ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct emu_stream_ *hipStream_t;
typedef struct emu_event_ { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// MI_EMU_DEVICES=n (default 1): n emulated devices.  hipSetDevice is per host thread, every hipMalloc belongs to the thread's current device, and a copy, a memset or a
// kernel argument that names another device's memory aborts the process (emu::check_dev_ptr): per-device tables, arenas and budgets of the multi-device fan-out are
// exercised for real.  The devices differ: device d reports 3 - (d & 1) compute units and 192 - 64 * (d & 1) MB of free memory.
namespace emu { int device_count(); int &cur_dev(); void check_dev_ptr(const void *p, const char *what); void note_launch(); }
inline hipError_t hipGetDeviceCount(int *n) { *n = emu::device_count(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= emu::device_count()) return hipErrorInvalidValue; emu::cur_dev() = d; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = (size_t)(192 - 64 * (emu::cur_dev() & 1)) << 20; *tot = (size_t)4 << 30; return hipSuccess; }   // small devices: the stream's eviction path gets exercised
void *emu_alloc(size_t n);
void *emu_alloc_dev(size_t n);
void emu_free(void *p);
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)emu_alloc_dev(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { *p = (T *)emu_alloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { emu_free(p); return hipSuccess; }
inline hipError_t hipHostFree(void *p) { emu_free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { emu::check_dev_ptr(d, "hipMemcpy dst"); emu::check_dev_ptr(s, "hipMemcpy src"); memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
  emu::check_dev_ptr(d, "hipMemcpy2D dst"); emu::check_dev_ptr(s, "hipMemcpy2D src");
  for (size_t y = 0; y < h; y++) memmove((uint8_t *)d + y * dp, (const uint8_t *)s + y * sp, w);
  return hipSuccess;
}
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
inline hipError_t hipMemset(void *d, int v, size_t n) { emu::check_dev_ptr(d, "hipMemset"); memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, struct emu_event_ *, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
double emu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event_{ 0.0 }; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event_{ 0.0 }; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = emu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
// a small "device": 3 compute units x 2 resident workgroups, so that persistent-workgroup launches see fewer workgroups than work items
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int dev) { *v = 3 - (dev & 1); return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 2; return hipSuccess; }
namespace emu {
template <typename T> inline void check_arg(const T &) {}
template <typename T> inline void check_arg(T *const &p) { check_dev_ptr((const void *)p, "kernel argument"); }
template <typename... A> inline void check_args(const A &...a) { (check_arg(a), ...); note_launch(); }
}
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) (emu::check_args(__VA_ARGS__), emu::launch((grid), (block), (size_t)(lds), [=]() { kern(__VA_ARGS__); }))

// tools/probe/exit_cost.hip PINNED_MB DEVICE_MB CHUNKS [STREAMS HEAP_MB EVENTS]: what a process pays outside its own work -- HIP runtime start, hipHostMalloc / hipMalloc of what the
// command line's three batch slots hold, and the time between _exit() and the parent seeing the process gone (measured by the calling shell).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main(int argc, char **argv) {
  const double t0 = now();
  const size_t pinned = (size_t)atol(argv[1]) << 20, dev = (size_t)atol(argv[2]) << 20; const int chunks = argc > 3 ? atoi(argv[3]) : 3;
  int n = 0; (void)hipGetDeviceCount(&n);
  const double t1 = now();
  void *h[16] = {}, *d[16] = {};
  for (int i = 0; i < chunks && pinned; i++) if (hipHostMalloc(&h[i], pinned / chunks) != hipSuccess) return 2;
  const double t2 = now();
  for (int i = 0; i < chunks && dev; i++) { if (hipMalloc(&d[i], dev / chunks) != hipSuccess) return 3; touch<<<(unsigned)((dev / chunks / 4096 + 255) / 256), 256>>>((char *)d[i], dev / chunks); }
  const int nstreams = argc > 4 ? atoi(argv[4]) : 0; const size_t heap = argc > 5 ? (size_t)atol(argv[5]) << 20 : 0; const int nev = argc > 6 ? atoi(argv[6]) : 0;
  hipStream_t st[64];
  for (int i = 0; i < nstreams && i < 64; i++) { (void)hipStreamCreate(&st[i]); if (d[0]) touch<<<1, 64, 0, st[i]>>>((char *)d[0], 4096); }
  for (int i = 0; i < nev; i++) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, nstreams ? st[i % nstreams] : 0); }
  if (heap) { char *hp = (char *)malloc(heap); for (size_t i = 0; i < heap; i += 4096) hp[i] = 1; }
  (void)hipDeviceSynchronize();
  const double t3 = now();
  fprintf(stderr, "runtime up %.3f s, pinned %zu MB in %.3f s, device %zu MB in %.3f s; main at %.6f leaving at %.6f\n", t1 - t0, pinned >> 20, t2 - t1, dev >> 20, t3 - t2, t0, now());
  _exit(0);
}

// Host-to-device rates for the ingestion leg (256 circuits x 7.2 MB of float32 traces): pageable hipMemcpyAsync as the library did it in round 3,
// a pinned source, and pageable memory staged through a pinned ring by T copy threads.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d tools/micro/h2d.hip -lpthread && /tmp/h2d
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t piece = 2000ull * 900 * 4, n = 256, total = piece * n;
  char* src = (char*)malloc(total);
  for (size_t i = 0; i < total; i += 4096) src[i] = (char)i;            // touch every page
  char* dst; CK(hipMalloc(&dst, total));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (size_t c = 0; c < n; ++c) CK(hipMemcpyAsync(dst + c * piece, src + c * piece, piece, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    double t = now() - t0;
    printf("pageable hipMemcpyAsync x256: %.1f ms  %.1f GB/s\n", t * 1e3, total / t / 1e9);
  }
  {
    char* pin; double t0 = now();
    CK(hipHostMalloc(&pin, total, hipHostMallocDefault));
    printf("hipHostMalloc of %.2f GB: %.1f ms\n", total / 1e9, (now() - t0) * 1e3);
    t0 = now(); memcpy(pin, src, total); printf("memcpy into it, one thread: %.1f ms\n", (now() - t0) * 1e3);
    for (int rep = 0; rep < 2; ++rep) {
      t0 = now();
      for (size_t c = 0; c < n; ++c) CK(hipMemcpyAsync(dst + c * piece, pin + c * piece, piece, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      double t = now() - t0;
      printf("pinned hipMemcpyAsync x256: %.1f ms  %.1f GB/s\n", t * 1e3, total / t / 1e9);
    }
    CK(hipHostFree(pin));
    t0 = now(); CK(hipHostRegister(src, total, hipHostRegisterDefault)); double tr = now() - t0;
    t0 = now();
    for (size_t c = 0; c < n; ++c) CK(hipMemcpyAsync(dst + c * piece, src + c * piece, piece, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    double t = now() - t0;
    printf("hipHostRegister: %.1f ms, then copy %.1f ms (%.1f GB/s)\n", tr * 1e3, t * 1e3, total / t / 1e9);
    t0 = now(); CK(hipHostUnregister(src)); printf("unregister %.1f ms\n", (now() - t0) * 1e3);
  }
  for (int T : {1, 2, 4, 8, 12}) {
    const int slots = 2 * T + 2;
    std::vector<char*> ring(slots);
    std::vector<hipEvent_t> ev(slots);
    for (int s = 0; s < slots; ++s) { CK(hipHostMalloc(&ring[s], piece, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[s], hipEventDisableTiming)); }
    for (int rep = 0; rep < 2; ++rep) {
      // every thread owns slots t, t + T, ... and the pieces c = t, t + T, ...: it waits for its slot's last copy, fills it, enqueues the DMA
      std::vector<hipStream_t> sts(T);
      for (auto& s : sts) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        CK(hipSetDevice(0));
        int k = 0;
        for (size_t c = t; c < n; c += T, ++k) {
          const int s = t + T * (k & 1);
          if (k >= 2) CK(hipEventSynchronize(ev[s]));
          memcpy(ring[s], src + c * piece, piece);
          CK(hipMemcpyAsync(dst + c * piece, ring[s], piece, hipMemcpyHostToDevice, sts[t]));
          CK(hipEventRecord(ev[s], sts[t]));
        }
        CK(hipStreamSynchronize(sts[t]));
      });
      for (auto& x : th) x.join();
      double t = now() - t0;
      printf("staged through a pinned ring, %2d threads: %.1f ms  %.1f GB/s\n", T, t * 1e3, total / t / 1e9);
      for (auto& s : sts) CK(hipStreamDestroy(s));
    }
    for (int s = 0; s < slots; ++s) { CK(hipHostFree(ring[s])); CK(hipEventDestroy(ev[s])); }
  }
  return 0;
}

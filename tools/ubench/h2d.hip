// H2D staging options for the host-env actor path (frames arrive in pageable numpy memory): pageable hipMemcpyAsync vs
// CPU memcpy into a pinned staging buffer + DMA.  build: hipcc --offload-arch=gfx950 -O2 h2d.hip -o h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (size_t n : {(size_t)20 * 28224, (size_t)120 * 28224}) {
    char* pageable = (char*)malloc(n); memset(pageable, 1, n);
    char* pinned; hipHostMalloc((void**)&pinned, n, hipHostMallocDefault); memset(pinned, 2, n);
    char* dev; hipMalloc((void**)&dev, n);
    const int it = 200;
    for (int mode = 0; mode < 3; ++mode) {
      double t0 = 0;
      for (int i = -20; i < it; ++i) {
        if (i == 0) t0 = now();
        if (mode == 0) hipMemcpyAsync(dev, pageable, n, hipMemcpyHostToDevice, st);
        if (mode == 1) { memcpy(pinned, pageable, n); hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, st); }
        if (mode == 2) hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
      }
      printf("%8zu bytes  %-28s %7.1f us\n", n, mode == 0 ? "pageable async+sync" : mode == 1 ? "memcpy->pinned, async+sync" : "pinned async+sync", (now() - t0) / it * 1e6);
    }
    free(pageable); hipHostFree(pinned); hipFree(dev);
  }
  return 0;
}

// GPU probe (not a test): how fast can [C][span][D] staging blocks reach a caller's fresh pageable buffer?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_probe tests/gpu_probes/d2h_probe.hip -lpthread && /tmp/d2h_probe
// (a) hipMemcpy2DAsync straight into pageable memory (what DrawSink does), fresh and pre-touched;
// (b) D2H into a pinned ring + N host threads scattering rows into the pageable buffer (fresh: the threads also take
//     the first-touch page faults).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define OK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t C = 65536, D = 1024, rows = 32, span = 4;  // 16 GiB in out, staging block 2 GiB
  const int threads = argc > 1 ? std::atoi(argv[1]) : 16;
  const size_t block_doubles = C * span * D;
  double* dev;
  OK(hipMalloc(&dev, block_doubles * sizeof(double)));
  OK(hipMemset(dev, 1, block_doubles * sizeof(double)));
  hipStream_t s;
  OK(hipStreamCreate(&s));
  auto run_direct = [&](bool touch) {
    double* out = static_cast<double*>(std::malloc(C * rows * D * sizeof(double)));
    if (touch) std::memset(out, 0, C * rows * D * sizeof(double));
    const double t0 = now();
    for (size_t first = 0; first < rows; first += span)
      OK(hipMemcpy2DAsync(out + first * D, rows * D * sizeof(double), dev, span * D * sizeof(double),
                          span * D * sizeof(double), C, hipMemcpyDeviceToHost, s));
    OK(hipStreamSynchronize(s));
    const double dt = now() - t0;
    std::printf("direct 2D copy into %s pageable memory: %.2f s, %.1f GB/s\n", touch ? "pre-touched" : "fresh", dt,
                C * rows * D * 8 / dt / 1e9);
    std::free(out);
  };
  run_direct(false);
  run_direct(true);
  // pinned ring: chunks of `cc` chains of the staging block
  const size_t cc = 2048, chunk_doubles = cc * span * D;  // 64 MiB
  const int ring = 4;
  double* pin[ring];
  hipEvent_t ev[ring];
  for (int i = 0; i < ring; ++i) {
    OK(hipHostMalloc(&pin[i], chunk_doubles * sizeof(double), hipHostMallocDefault));
    OK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  }
  for (int pass = 0; pass < 6; ++pass) {
    const int nt = pass % 3 == 0 ? 4 : pass % 3 == 1 ? 8 : threads;
    const bool touched = pass >= 3;  // (b') the same into memory that has its pages already
    double* out = static_cast<double*>(std::malloc(C * rows * D * sizeof(double)));
    if (touched) std::memset(out, 0, C * rows * D * sizeof(double));
    const double t0 = now();
    size_t job = 0;
    std::vector<std::thread> scatter[ring];
    for (size_t first = 0; first < rows; first += span) {
      for (size_t c0 = 0; c0 < C; c0 += cc, ++job) {
        const int slot = job % ring;
        for (auto& t : scatter[slot]) t.join();  // the slot's previous scatter is done
        scatter[slot].clear();
        OK(hipMemcpyAsync(pin[slot], dev + c0 * span * D, chunk_doubles * sizeof(double), hipMemcpyDeviceToHost, s));
        OK(hipEventRecord(ev[slot], s));
        for (int t = 0; t < nt; ++t)
          scatter[slot].emplace_back([=] {
            if (t == 0) (void)hipEventSynchronize(ev[slot]);
            else (void)hipEventSynchronize(ev[slot]);
            for (size_t c = c0 + t; c < c0 + cc; c += nt)
              std::memcpy(out + c * rows * D + first * D, pin[slot] + (c - c0) * span * D, span * D * sizeof(double));
          });
      }
    }
    for (auto& v : scatter)
      for (auto& t : v) t.join();
    const double dt = now() - t0;
    std::printf("pinned ring of %d x %zu MiB + %d scatter threads into %s pageable memory: %.2f s, %.1f GB/s\n", ring,
                chunk_doubles * 8 >> 20, nt, touched ? "pre-touched" : "fresh", dt, C * rows * D * 8 / dt / 1e9);
    std::free(out);
  }
  return 0;
}

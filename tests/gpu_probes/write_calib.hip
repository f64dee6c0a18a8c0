// GPU probe: calibrates rocprofv3's WRITE_SIZE / FETCH_SIZE on THIS engine's access pattern -- a wavefront streams one
// 8 KB row (64 lanes x 16 bytes x 8) per "chain", non-temporal or plain, a known byte count.
//   hipcc --offload-arch=gfx950 -O3 -o write_calib tests/gpu_probes/write_calib.hip
//   rocprofv3 --pmc WRITE_SIZE -d out -- ./write_calib        (and --pmc FETCH_SIZE in a run of its own)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ void store_rows(double* out, long long rows) {
  const int lane = threadIdx.x;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    v2f64* p = reinterpret_cast<v2f64*>(out + r * 1024) + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v2f64 v;
      v[0] = static_cast<double>(r);
      v[1] = static_cast<double>(k);
      if (NT) __builtin_nontemporal_store(v, p + k * 64);
      else p[k * 64] = v;
    }
  }
}
template <bool NT>
__global__ void load_rows(const double* in, long long rows, double* sink) {
  const int lane = threadIdx.x;
  double acc = 0;
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const v2f64* p = reinterpret_cast<const v2f64*>(in + r * 1024) + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const v2f64 v = NT ? __builtin_nontemporal_load(p + k * 64) : p[k * 64];
      acc += v[0] + v[1];
    }
  }
  if (acc == 12345.678) sink[0] = acc;
}
int main() {
  const long long rows = 9LL * 65536;  // 9 planes of 65 536 x 1 024 doubles = 4.83 GB: the headline launch's compulsory writes
  double* buf;
  double* sink;
  if (hipMalloc(&buf, rows * 1024 * sizeof(double)) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(store_rows<true>, dim3(1024), dim3(64), 0, 0, buf, rows);
    hipLaunchKernelGGL(store_rows<false>, dim3(1024), dim3(64), 0, 0, buf, rows);
    hipLaunchKernelGGL(load_rows<true>, dim3(1024), dim3(64), 0, 0, buf, rows, sink);
    hipLaunchKernelGGL(load_rows<false>, dim3(1024), dim3(64), 0, 0, buf, rows, sink);
  }
  hipDeviceSynchronize();
  std::printf("known bytes per kernel: %.3f GB (store_rows<nt>, store_rows<plain>, load_rows<nt>, load_rows<plain>)\n",
              rows * 1024 * 8 / 1e9);
  return 0;
}

// GPU probe: the DPP / permlane-swap butterfly of wn_traj.h against the plain __shfl_xor butterfly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../walnuts_amd/csrc/wn_traj.h"

__global__ void probe(const double* in, double* out_fast, double* out_ref, double* out_lane) {
  double v = in[blockIdx.x * 64 + threadIdx.x];
  out_fast[blockIdx.x * 64 + threadIdx.x] = wn::wave_sum(v);
  double r = v;
  for (int off = 1; off < 64; off <<= 1) r = r + __shfl_xor(r, off, 64);
  out_ref[blockIdx.x * 64 + threadIdx.x] = r;
  out_lane[blockIdx.x * 64 + threadIdx.x] = wn::lane_value(v, (blockIdx.x * 7) & 63);
}

int main() {
  const int B = 4096, N = B * 64;
  double *h = (double*)malloc(N * 8), *f = (double*)malloc(N * 8), *r = (double*)malloc(N * 8), *l = (double*)malloc(N * 8);
  srand(1);
  for (int i = 0; i < N; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * ((i % 7 == 0) ? 1e6 : 1.0);
  double *d, *df, *dr, *dl;
  hipMalloc(&d, N * 8); hipMalloc(&df, N * 8); hipMalloc(&dr, N * 8); hipMalloc(&dl, N * 8);
  hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(B), dim3(64), 0, 0, d, df, dr, dl);
  hipMemcpy(f, df, N * 8, hipMemcpyDeviceToHost); hipMemcpy(r, dr, N * 8, hipMemcpyDeviceToHost);
  hipMemcpy(l, dl, N * 8, hipMemcpyDeviceToHost);
  long bad = 0, badlane = 0, nonuni = 0;
  for (int i = 0; i < N; ++i) {
    if (memcmp(&f[i], &r[i], 8)) { if (bad < 5) printf("mismatch blk %d lane %d fast %.17g ref %.17g\n", i / 64, i % 64, f[i], r[i]); ++bad; }
    if (memcmp(&f[i], &f[(i / 64) * 64], 8)) ++nonuni;
    if (l[i] != h[(i / 64) * 64 + (((i / 64) * 7) & 63)]) ++badlane;
  }
  printf("wave_sum mismatches %ld of %d, non-uniform %ld, lane_value mismatches %ld\n", bad, N, nonuni, badlane);
  return bad || badlane;
}

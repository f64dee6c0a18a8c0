// GPU probe: per-lane (divergent) evaluation of the counter stream + dlog/dexp against the host build of
// the same header, and the lane_value hand-out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../walnuts_amd/csrc/wn_traj.h"

__global__ void probe(double* u_out, double* lu_out, double* e_out, double* pick) {
  const int lane = threadIdx.x & 63;
  const uint32_t chain = blockIdx.x;
  double u = wnd::stream_uniform(77, chain, 3, wnd::kStreamTree, lane);
  double lu = wnd::dlog(u);
  u_out[blockIdx.x * 64 + lane] = u;
  lu_out[blockIdx.x * 64 + lane] = lu;
  e_out[blockIdx.x * 64 + lane] = wnd::dexp(lu * 3.0);
  // hand-out like Traj::next_draw_slot
  int n = 0;
  double acc = 0;
  for (int k = 0; k < 40; ++k) {
    int j = wn::uni(n); ++n;
    acc += wn::lane_value(lu, j) * (k + 1) + wn::lane_value(u, j);
  }
  pick[blockIdx.x * 64 + lane] = acc;
}

int main() {
  const int B = 2048, N = B * 64;
  std::vector<double> u(N), lu(N), e(N), pk(N);
  double *du, *dl, *de, *dp;
  hipMalloc(&du, N * 8); hipMalloc(&dl, N * 8); hipMalloc(&de, N * 8); hipMalloc(&dp, N * 8);
  hipLaunchKernelGGL(probe, dim3(B), dim3(64), 0, 0, du, dl, de, dp);
  hipMemcpy(u.data(), du, N * 8, hipMemcpyDeviceToHost); hipMemcpy(lu.data(), dl, N * 8, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), de, N * 8, hipMemcpyDeviceToHost); hipMemcpy(pk.data(), dp, N * 8, hipMemcpyDeviceToHost);
  long bu = 0, bl = 0, be = 0, bp = 0;
  for (int b = 0; b < B; ++b) {
    double hl[64], hu[64];
    for (int l = 0; l < 64; ++l) {
      hu[l] = wnd::stream_uniform(77, b, 3, wnd::kStreamTree, l);
      hl[l] = wnd::dlog(hu[l]);
      double he = wnd::dexp(hl[l] * 3.0);
      if (memcmp(&hu[l], &u[b * 64 + l], 8)) ++bu;
      if (memcmp(&hl[l], &lu[b * 64 + l], 8)) { if (bl < 5) printf("log mismatch u=%.17g dev %.17g host %.17g\n", hu[l], lu[b*64+l], hl[l]); ++bl; }
      if (memcmp(&he, &e[b * 64 + l], 8)) ++be;
    }
    double acc = 0;
    for (int k = 0; k < 40; ++k) acc += hl[k] * (k + 1) + hu[k];
    for (int l = 0; l < 64; ++l) if (memcmp(&acc, &pk[b * 64 + l], 8)) ++bp;
  }
  printf("uniform mismatches %ld, log %ld, exp %ld, hand-out %ld of %d\n", bu, bl, be, bp, N);
  return 0;
}

// GPU probe: issue cost and dependent latency of the fp64 vector instructions the transition kernel is made of
// (one wavefront on one SIMD, s_memtime around unrolled chains).   hipcc --offload-arch=gfx950 -O3 -o fp64_latency fp64_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, int OP>
__global__ void probe(double* out, unsigned long long* cyc, double seed) {
  double v[CHAINS];
  for (int c = 0; c < CHAINS; ++c) v[c] = seed + c + threadIdx.x;
  const double k = seed * 0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
  for (int i = 0; i < 512 / CHAINS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(k));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(k));
      if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[c]) : "v"(k));
      if (OP == 3) asm volatile("v_mov_b32 %0, %0" : "+v"(*(int*)&v[c]));
      if (OP == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(v[c]) : "v"(threadIdx.x) : "vcc");
      if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(*(int*)&v[c]) : "v"(threadIdx.x));
      if (OP == 6) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(*(int*)&v[c]) : : "a0");
      if (OP == 7) asm volatile("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %0, s20, 5" : "+v"(*(int*)&v[c]) : : "s20");
      // the cross-lane steps of the reductions, each as a dependent chain (result feeds the next step)
      if (OP == 8) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(*(int*)&v[c]));
      if (OP == 9) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(*(int*)&v[c]), "+v"(((int*)&v[c])[1]));
      if (OP == 10) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(*(int*)&v[c]), "+v"(((int*)&v[c])[1]));
      if (OP == 11) asm volatile("v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20" : "+v"(*(int*)&v[c]) : : "s20");
      if (OP == 12) asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0" : "+v"(*(int*)&v[c]) : : "s20");

    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += v[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS, int OP>
void run(const char* name) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<CHAINS, OP>), 1, 64, 0, 0, out, cyc, 1.0000001);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-14s chains=%d  %6.2f cycles per instruction (512 instructions)\n", name, CHAINS, h / 512.0);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>("v_add_f64"); run<2, 0>("v_add_f64"); run<4, 0>("v_add_f64"); run<8, 0>("v_add_f64");
  run<1, 1>("v_mul_f64"); run<2, 1>("v_mul_f64"); run<4, 1>("v_mul_f64"); run<8, 1>("v_mul_f64");
  run<1, 2>("v_fma_f64"); run<2, 2>("v_fma_f64"); run<4, 2>("v_fma_f64"); run<8, 2>("v_fma_f64");
  run<1, 3>("v_mov_b32"); run<8, 3>("v_mov_b32");
  run<1, 4>("v_mad_u64_u32"); run<2, 4>("v_mad_u64_u32"); run<4, 4>("v_mad_u64_u32"); run<8, 4>("v_mad_u64_u32");
  run<1, 5>("v_xor_b32"); run<8, 5>("v_xor_b32");
  run<1, 6>("acc wr+rd pair"); run<8, 6>("acc wr+rd pair");
  run<1, 7>("readl+writel"); run<8, 7>("readl+writel");
  run<1, 8>("mov_dpp"); run<4, 8>("mov_dpp");
  run<1, 9>("permlane32_swap"); run<4, 9>("permlane32_swap");
  run<1, 10>("permlane16_swap"); run<4, 10>("permlane16_swap");
  run<1, 11>("readfirstl+mov"); run<4, 11>("readfirstl+mov");
  run<1, 12>("readl+add_u32(s)"); run<4, 12>("readl+add_u32(s)");
  return 0;
}

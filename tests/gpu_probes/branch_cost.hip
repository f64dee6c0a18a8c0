// GPU probe: what a scalar branch costs ONE wavefront on its SIMD (gfx950): taken / not taken, short forward hops,
// loop back-edges, with and without vector work between.  s_memtime around 256 repetitions, one wavefront.
//   hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ void probe(double* out, unsigned long long* cyc, double seed, int never) {
  double v = seed + threadIdx.x, w = seed * 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (KIND == 0) {  // 4 independent fma, no branch
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1" : "+v"(v), "+v"(w));)
  }
  if (KIND == 1) {  // ... + a NOT-taken conditional branch
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n s_cmp_eq_u32 %2, 1\n s_cbranch_scc1 1f\n1:" : "+v"(v), "+v"(w) : "s"(never) : "scc");)
  }
  if (KIND == 2) {  // ... + a TAKEN forward branch over one instruction
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n s_cmp_eq_u32 %2, 0\n s_cbranch_scc1 1f\n v_mov_b32 v100, 0\n1:" : "+v"(v), "+v"(w) : "s"(never) : "scc", "v100");)
  }
  if (KIND == 3) {  // ... + a TAKEN forward branch over 40 instructions (beyond a 64-byte fetch line or two)
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n s_cmp_eq_u32 %2, 0\n s_cbranch_scc1 1f\n .rept 40\n v_mov_b32 v100, 0\n .endr\n1:" : "+v"(v), "+v"(w) : "s"(never) : "scc", "v100");)
  }
  if (KIND == 4) {  // unconditional s_branch over one instruction
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n s_branch 1f\n v_mov_b32 v100, 0\n1:" : "+v"(v), "+v"(w) : : "v100");)
  }
  if (KIND == 5) {  // a loop of 64 iterations: 4 fma + counter + taken BACKWARD branch
    int n = 64 + never;
    asm volatile("1:\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n s_sub_u32 %2, %2, 1\n s_cmp_lg_u32 %2, 0\n s_cbranch_scc1 1b" : "+v"(v), "+v"(w), "+s"(n) : : "scc");
  }
  if (KIND == 6) {  // vcc branch as the compiler writes it: v_cmp -> s_and vcc, exec -> s_cbranch_vccnz (not taken)
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n v_cmp_gt_f64 vcc, 0, %0\n s_and_b64 vcc, exec, vcc\n s_cbranch_vccnz 1f\n1:" : "+v"(v), "+v"(w) : : "vcc");)
  }
  if (KIND == 7) {  // 16 fma between taken forward branches
    REP16(asm volatile(".rept 8\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n .endr\n s_cmp_eq_u32 %2, 0\n s_cbranch_scc1 1f\n v_mov_b32 v100, 0\n1:" : "+v"(v), "+v"(w) : "s"(never) : "scc", "v100");)
  }
  if (KIND == 8) {  // 16 fma, no branch
    REP16(asm volatile(".rept 8\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %1, %1\n .endr" : "+v"(v), "+v"(w));)
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  out[threadIdx.x] = v + w;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, int reps) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<KIND>), 1, 64, 0, 0, out, cyc, 1.0000001, 0);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-64s %7.1f ticks per repetition (%d repetitions)\n", name, (double)h / reps, reps);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("4 fma (2 chains)", 64);
  run<1>("4 fma + s_cmp + s_cbranch NOT taken", 64);
  run<2>("4 fma + s_cmp + s_cbranch TAKEN over 1 instruction", 64);
  run<3>("4 fma + s_cmp + s_cbranch TAKEN over 40 instructions", 64);
  run<4>("4 fma + s_branch over 1 instruction", 64);
  run<5>("loop: 4 fma + s_sub + s_cmp + s_cbranch TAKEN backward", 64);
  run<6>("4 fma + v_cmp + s_and + s_cbranch_vccnz NOT taken", 64);
  run<7>("16 fma + s_cmp + s_cbranch TAKEN over 1 instruction", 16);
  run<8>("16 fma", 16);
  return 0;
}

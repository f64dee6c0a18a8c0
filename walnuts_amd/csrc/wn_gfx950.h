// wn_gfx950.h -- the gfx950 (MI355X, CDNA4) platform layer of the kernels: the HIP runtime, LDS address-space
// pointers, and the wavefront primitives everything else is written on -- wave-uniform broadcast, the packed
// two-sum butterfly (DPP / v_permlane*_swap), lane reads, laundered kernel arguments, streaming loads and stores.
// wn_hip.h includes this file for the product build; the CPU test tier substitutes tests/cpusim/wn_cpusim.h, which
// provides the same names over a lock-step host emulation.  No other source file of walnuts_amd/csrc knows which.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define WN_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) double name[]
// LDS is addressed through address-space-3 pointers only, so every pool / scratch access is a
// ds_* instruction (no flat aperture tests)
#define WN_LDS __attribute__((address_space(3)))
typedef double v2f64 __attribute__((ext_vector_type(2)));
// N doubles per lane addressed by a wave-uniform run-time index (VGPR-relative addressing)
#define WN_VEC_OF(N) __attribute__((ext_vector_type(N)))

namespace wn {

// launch sizes of the element-wise / summary kernels
constexpr int kSummaryBlock = 256;
constexpr int kSummaryLagSlabChains = 8192;
constexpr int kSummaryCandidateCap = 2048;

__device__ __forceinline__ uint64_t bits_of(double d) { return static_cast<uint64_t>(__double_as_longlong(d)); }
__device__ __forceinline__ double double_of(uint64_t u) { return __longlong_as_double(static_cast<long long>(u)); }

// ---- wave-uniform helpers ----------------------------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
  const uint64_t u = bits_of(v);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u >> 32));
  return double_of((static_cast<uint64_t>(hi) << 32) | lo);
}

// xor-butterfly sum over the 64 lanes, offsets 32,1,2,4,8,16: every lane ends with the same bits
// (a+b == b+a), and the CPU oracle replays exactly this association order.
// gfx950: offsets 1,2 are quad permutes, 4 and 8 are row_half_mirror / row_mirror (the groups are already
// uniform there, so the mirrored lane holds the xor partner's value), 16 and 32 are v_permlane{16,32}_swap.
// All VALU: no LDS crossbar traffic (ds_bpermute) on the reduction path.
template <int CTRL>
__device__ __forceinline__ double dpp_partner(double v) {
  const uint64_t u = bits_of(v);
  const int lo = static_cast<int>(u), hi = static_cast<int>(u >> 32);
  // mov_dpp (no tied `old` operand): one v_mov_b32_dpp per dword, no preparatory copies; every lane has a
  // valid source in these patterns
  const int plo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  const int phi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return double_of((static_cast<uint64_t>(static_cast<uint32_t>(phi)) << 32) | static_cast<uint32_t>(plo));
}
// Two sums at once.  Offset 32 goes first and packs the pair: after one v_permlane32_swap per dword, lanes 0-31
// hold a[l] + a[l+32] and lanes 32-63 hold b[l-32] + b[l]; offsets 1..16 never leave a 32-lane half, so ONE
// butterfly finishes both (18 VALU instead of 36).  Returns the packed register: a's sum in lanes 0-31, b's in
// lanes 32-63.
__device__ __forceinline__ double wave_sum_packed(double a, double b) {
  double v;
  {
    const uint64_t ua = bits_of(a), ub = bits_of(b);
    const auto lo = __builtin_amdgcn_permlane32_swap(static_cast<uint32_t>(ua), static_cast<uint32_t>(ub), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(static_cast<uint32_t>(ua >> 32), static_cast<uint32_t>(ub >> 32),
                                                     false, false);
    v = double_of((static_cast<uint64_t>(hi[0]) << 32) | lo[0]) + double_of((static_cast<uint64_t>(hi[1]) << 32) | lo[1]);
  }
  v = v + dpp_partner<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
  v = v + dpp_partner<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
  v = v + dpp_partner<0x141>(v);  // row_half_mirror      : partner quad  (lane ^ 4)
  v = v + dpp_partner<0x140>(v);  // row_mirror           : partner octet (lane ^ 8)
  {
    const uint64_t u = bits_of(v);
    const uint32_t lo = static_cast<uint32_t>(u), hi = static_cast<uint32_t>(u >> 32);
    const auto p = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto q = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = double_of((static_cast<uint64_t>(q[0]) << 32) | p[0]) + double_of((static_cast<uint64_t>(q[1]) << 32) | p[1]);
  }
  return v;
}
// A per-lane condition on a value the packed butterfly left uniform within each half of the wavefront, as a SCALAR
// branch condition: does it hold in lane 0 or in lane 32 (= in any lane)?  One v_cmp into a lane mask and a scalar
// test of the mask: no v_readlane round trip in front of the compare.
__device__ __forceinline__ bool either_half(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
// value held by lane `src_lane` (wave-uniform index) as a scalar
__device__ __forceinline__ double lane_value(double v, int src_lane) {
  const uint64_t u = bits_of(v);
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(u), src_lane);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(u >> 32), src_lane);
  return double_of((static_cast<uint64_t>(hi) << 32) | lo);
}

__device__ __forceinline__ int lane_value(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
// The value of the lane below / above (lane 0 / lane 63 keep their own): one DPP move per dword (wave_shr:1 /
// wave_shl:1, a full-wavefront shift gfx9 still has) where __shfl_up / __shfl_down are two ds_bpermute round trips
// through the LDS crossbar each.  For models whose gradient reads the neighbouring coordinates (rw1).
__device__ __forceinline__ double lane_below(double x) {
  const unsigned long long b = bits_of(x);
  int lo = static_cast<int>(static_cast<unsigned>(b)), hi = static_cast<int>(static_cast<unsigned>(b >> 32));
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return double_of((static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) | static_cast<unsigned>(lo));
}
__device__ __forceinline__ double lane_above(double x) {
  const unsigned long long b = bits_of(x);
  int lo = static_cast<int>(static_cast<unsigned>(b)), hi = static_cast<int>(static_cast<unsigned>(b >> 32));
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return double_of((static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) | static_cast<unsigned>(lo));
}
// reg[dst_lane] = v for a wave-uniform value and index: a select on the lane id (v_writelane_b32 would do it in one
// instruction per dword, but this compiler has no builtin for it and hands an inline-asm "s" operand a VECTOR
// register whenever it knows the value uniform without having it in a scalar one)
__device__ __forceinline__ int opaque_lane_id();
__device__ __forceinline__ void set_lane(int& reg, int v, int dst_lane) { reg = opaque_lane_id() == dst_lane ? v : reg; }
__device__ __forceinline__ void set_lane(double& reg, double v, int dst_lane) { reg = opaque_lane_id() == dst_lane ? v : reg; }

// A double parked in the accumulator half of the register file (AGPRs).  Vector arithmetic cannot read AGPRs, so the
// compiler uses them only as spill space and shuffles whole vectors in and out at region boundaries as it sees
// fit; a value parked explicitly stays put, and a vector that is read once per doubling costs exactly one
// v_accvgpr_read per register per doubling.
struct ParkedDouble {
  uint32_t lo, hi;
};
__device__ __forceinline__ void park(ParkedDouble& a, double v) {
  const uint64_t u = bits_of(v);
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.lo) : "v"(static_cast<uint32_t>(u)));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.hi) : "v"(static_cast<uint32_t>(u >> 32)));
}
__device__ __forceinline__ double fetch(const ParkedDouble& a) {
  uint32_t lo, hi;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a.lo));
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(a.hi));
  return double_of((static_cast<uint64_t>(hi) << 32) | lo);
}

// The same interface over an ordinary register: kernels built for three or more wavefronts per SIMD name no accumulator
// registers at all -- as soon as a kernel does, the compiler splits its register budget evenly between the two files
// (84 + 84 of 168 at three wavefronts per SIMD), which is not what a kernel whose arithmetic state alone is 112 needs.
struct PlainDouble {
  double v;
};
__device__ __forceinline__ void park(PlainDouble& a, double v) { a.v = v; }
__device__ __forceinline__ double fetch(const PlainDouble& a) { return a.v; }

// The lane's index within its wavefront, COMPUTED where it is asked for (v_mbcnt_lo/hi on an all-ones mask: two VALU
// instructions, independent of EXEC): whatever is derived from it -- addresses, padding masks, counter words -- is rebuilt
// at the use instead of being hoisted to the kernel entry and held for the whole kernel.  Deliberately NOT threadIdx.x:
// that value arrives in v0 and nothing can recompute it, so a kernel that reads it late keeps a VGPR alive from the
// entry to the last use -- across every divergent region of the transition.  With one wavefront per SIMD the allocator
// parks such a register in the accumulator file and reloads it where it sees fit, and a reload it places INSIDE a
// region that runs under a partial EXEC mask restores the active lanes only (seen with rocgdb in the funnel (4,4)
// warmup kernel: v0 = {0, garbage x 63} after a reload inside an `if (lane == 0)` block, then a fault on the first
// address derived from it).  The wavefront's index within the workgroup is wave-uniform and lives in an SGPR, whose
// spills (v_writelane / v_readlane) do not depend on EXEC.
__device__ __forceinline__ int opaque_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// a + b on the scalar unit, opaque to the optimiser: the sum is formed where it is written instead of being folded
// into (and re-ordering) the address arithmetic that follows.  Used once, for the workgroup's first chain index --
// the plain `begin + blockIdx.x` cost the headline kernel 2 % through a different register allocation
// (profiles/r04/ab_first_chain.txt).
__device__ __forceinline__ int opaque_scalar_add(int a, int b) {
  asm volatile("s_add_u32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");
  return a;
}
// a (wave-uniform) double the optimiser cannot see through: what is computed from it stays where it is written (the
// step size of a leaf: h * inv_mass, hoisted out of the leaf loop, is sixteen products kept in -- and fetched back
// from -- the accumulator file for every leaf instead of sixteen multiplications)
__device__ __forceinline__ double opaque_uniform(double v) {
  unsigned long long b = bits_of(v);
  asm volatile("" : "+v"(b));  // (a vector register: an "s" operand fails to compile where the value sits in one)
  return double_of(b);
}
// a wave-uniform pointer into global memory, pinned in a scalar register pair the optimiser cannot see through: address
// arithmetic on it stays "scalar base + vector offset" instead of being re-associated into per-lane 64-bit addresses
// (the address space is kept: the accesses stay global_load / global_store)
using GlobalBytes = __attribute__((address_space(1))) char*;
using GlobalV2 = __attribute__((address_space(1))) v2f64*;
__device__ __forceinline__ GlobalBytes opaque_scalar_pointer(const void* p) {
  // (readfirstlane: nothing when the value already sits in scalar registers, which is where a uniform pointer usually
  // is; where the allocator kept it in vector registers an "s" operand alone does not compile)
  const unsigned long long bits = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits >> 32));
  GlobalBytes g = (GlobalBytes)((static_cast<unsigned long long>(hi) << 32) | lo);
  asm volatile("" : "+s"(g));
  return g;
}
__device__ __forceinline__ v2f64 load_pair_at(GlobalBytes base, unsigned byte_offset) {
  return *reinterpret_cast<GlobalV2>(base + byte_offset);
}
__device__ __forceinline__ void store_pair_at(GlobalBytes base, unsigned byte_offset, v2f64 v) {
  *reinterpret_cast<GlobalV2>(base + byte_offset) = v;
}
// "this register may have changed" as far as the optimiser can tell (no instruction): what is computed from it cannot
// be hoisted across this point
__device__ __forceinline__ void launder(double& v) { asm volatile("" : "+v"(v)); }
// the wavefront's index within its workgroup, read once at the kernel entry (wave-uniform: a scalar register)
__device__ __forceinline__ int wave_in_workgroup() { return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6); }

// The kernel's single by-value argument struct, read back from the kernel-argument segment behind an optimisation
// barrier: fields used once or twice per transition are then fetched where they are used (s_load) instead of being
// loaded at the kernel entry and held -- or spilled to VGPR lanes -- for the whole kernel.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T& kernel_argument(const T&) {
  typedef const __attribute__((address_space(4))) T CT;
  CT* p = (CT*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *p;
}

// Wait here for every load this wavefront has in flight (s_waitcnt vmcnt(0) lgkmcnt(0)).  The compiler places its
// waits where a loaded register is first read and merges "may be in flight" over all control-flow paths: a load
// into long-lived state registers at the end of a RARE path makes it guard every later use of those registers on
// the COMMON paths with counted waits -- which also wait for stores (one counter for both on this hardware).
__device__ __forceinline__ void drain_loads() { __builtin_amdgcn_s_waitcnt(0x0070); }

// read-once / write-once traffic streamed past the L2 (nt)
__device__ __forceinline__ v2f64 stream_load(const v2f64* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(v2f64 v, v2f64* p) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stream_store(double v, double* p) { __builtin_nontemporal_store(v, p); }

#if defined(WN_TIMELINE)
__device__ __forceinline__ unsigned long long shader_clock() { return __builtin_amdgcn_s_memtime(); }
#endif

}  // namespace wn

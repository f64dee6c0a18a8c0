// wn_cpusim.h -- TEST INFRASTRUCTURE: a lock-step host emulation of one HIP workgroup plus the sliver
// of the HIP runtime API that walnuts_amd/csrc uses.  tests/cpusim/build.py compiles the product's
// .hip sources with g++ -DWN_CPU_SIM against this header into tests/cpusim/libwalnuts_sim.so so that
// the `-m "not gpu"` suite can drive the real host logic and the kernels' control flow (span pool,
// random-number order, adaptation) against the oracle.  Every lane is a cooperative fiber of the
// launching thread (its own stack, switched at every cross-lane operation and barrier: ~20 ns a
// switch; round 3's one OS thread per lane paid a futex round trip there and made the CPU tier take
// 17 minutes).  It is never linked into libwalnuts_hip.so and proves nothing about the GPU build:
// device parity is established by the `-m gpu` tests only.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <deque>
#include <functional>
#include <sys/mman.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
// static LDS arrays: blocks run one after another in the emulation, so function-local statics are
// exactly "storage shared by the threads of the running block"
#define __shared__ static
#define WN_LDS
typedef double v2f64 __attribute__((vector_size(16)));
#define WN_VEC_OF(N) __attribute__((vector_size(8 * (N))))

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

// ---- fibers: callee-saved registers on the fiber's own stack, the stack pointer is the whole context -------------
#if defined(__x86_64__)
extern "C" void wnsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.weak wnsim_switch
.hidden wnsim_switch
.type wnsim_switch,@function
wnsim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size wnsim_switch,.-wnsim_switch
)");
#else
#error "tests/cpusim: the fiber switch is written for x86-64 (the container the CPU tier runs in)"
#endif

namespace wnsim {
struct Barrier {  // arrivals park here until the last one comes
  unsigned expected = 0, arrived = 0;
  std::vector<unsigned> parked;
};
struct Block {
  unsigned nthreads = 0;
  Barrier bar;
  std::vector<Barrier> wave_bar;  // cross-lane operations are wavefront-scoped
  std::vector<uint64_t> xchg;
  double* smem = nullptr;
  // the scheduler: lanes that can run, in the order they became runnable
  std::deque<unsigned> ready;
  std::vector<void*> sp;  // a suspended lane's stack pointer
  void* main_sp = nullptr;
  unsigned running = 0, finished = 0;
  std::function<void()> body;
};
inline thread_local Block* blk = nullptr;
inline thread_local dim3 tidx, bidx, gdim, bdim;

constexpr size_t kFiberStackBytes = size_t{1} << 20;
// one region of lane stacks per launching thread, grown on demand and kept (pages are touched lazily)
inline char* fiber_stacks(unsigned lanes) {
  thread_local char* base = nullptr;
  thread_local unsigned have = 0;
  if (lanes > have) {
    if (base) munmap(base, static_cast<size_t>(have) * kFiberStackBytes);
    void* p = mmap(nullptr, static_cast<size_t>(lanes) * kFiberStackBytes, PROT_READ | PROT_WRITE,
                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) {
      std::fprintf(stderr, "cpusim: cannot map %u lane stacks\n", lanes);
      std::abort();
    }
    base = static_cast<char*>(p);
    have = lanes;
  }
  return base;
}
// hand the processor to the next runnable lane (or back to launch() when every lane has finished)
inline void run_next(void** save_sp) {
  Block& B = *blk;
  if (B.ready.empty()) {
    if (B.finished != B.nthreads) {
      std::fprintf(stderr, "cpusim: deadlock -- %u of %u lanes wait at barriers nobody else will reach\n",
                   B.nthreads - B.finished, B.nthreads);
      std::abort();
    }
    wnsim_switch(save_sp, B.main_sp);
    return;
  }
  const unsigned next = B.ready.front();
  B.ready.pop_front();
  B.running = next;
  tidx = dim3(next);
  wnsim_switch(save_sp, B.sp[next]);
  // (back in this lane: whoever resumed it has set `running` and tidx to it)
}
inline void arrive(Barrier& b) {
  Block& B = *blk;
  if (++b.arrived == b.expected) {  // the last one: everybody else becomes runnable, this lane simply goes on
    b.arrived = 0;
    for (unsigned t : b.parked) B.ready.push_back(t);
    b.parked.clear();
    return;
  }
  const unsigned me = B.running;
  b.parked.push_back(me);
  run_next(&B.sp[me]);
}
inline void sync() { arrive(blk->bar); }
inline void wave_sync() { arrive(blk->wave_bar[tidx.x >> 6]); }
inline void fiber_entry() {
  Block& B = *blk;
  B.body();
  ++B.finished;
  void* dead = nullptr;
  run_next(&dead);  // never comes back
  std::abort();
}
// a cross-lane move inside one wavefront (only that wavefront's lanes need to arrive: one wavefront of a
// workgroup may run scalar work the others skip)
template <class T>
T exchange(T v, unsigned src_tid) {
  static_assert(sizeof(T) <= 8, "exchange moves at most 8 bytes");
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  blk->xchg[tidx.x] = raw;
  wave_sync();
  const uint64_t r = blk->xchg[src_tid];
  wave_sync();
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
// value of the wavefront's first lane (all lanes are active everywhere in these kernels)
template <class T>
T readfirstlane(T v) {
  return exchange(v, tidx.x & ~63u);
}

// (launches from several host threads -- the shards of walnutpie_sample_device_multi -- run one after another: static
// __shared__ arrays are function-local statics here)
inline std::mutex& launch_mutex() {
  static std::mutex m;
  return m;
}
template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t smem_bytes, A... args) {
  std::lock_guard<std::mutex> one_at_a_time(launch_mutex());
  for (unsigned b = 0; b < grid.x; ++b) {
    Block B;
    B.nthreads = block.x;
    B.bar.expected = block.x;
    B.xchg.assign(block.x, 0);
    B.wave_bar.resize((block.x + 63) / 64);
    for (unsigned w = 0; w * 64 < block.x; ++w) B.wave_bar[w].expected = std::min(64u, block.x - w * 64);
    const size_t bytes = ((smem_bytes + 63) / 64 + 1) * 64;
    B.smem = static_cast<double*>(std::aligned_alloc(64, bytes));
    std::memset(B.smem, 0xCD, bytes);
    B.body = [&] { kernel(args...); };
    char* stacks = fiber_stacks(block.x);
    B.sp.resize(block.x);
    for (unsigned t = 0; t < block.x; ++t) {
      // a fresh lane: six callee-saved registers, then fiber_entry as the address the first switch returns to
      // (the slot holding it is 16-byte aligned, so the entry sees the stack a call would have left)
      void** top = reinterpret_cast<void**>(stacks + static_cast<size_t>(t + 1) * kFiberStackBytes);
      top[-1] = nullptr;
      top[-2] = reinterpret_cast<void*>(&fiber_entry);
      for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
      B.sp[t] = top - 8;
      B.ready.push_back(t);
    }
    Block* outer = blk;
    blk = &B;
    bidx = dim3(b);
    gdim = grid;
    bdim = block;
    run_next(&B.main_sp);  // returns when the last lane has finished
    blk = outer;
    std::free(B.smem);
  }
}
}  // namespace wnsim

#define threadIdx wnsim::tidx
#define blockIdx wnsim::bidx
#define gridDim wnsim::gdim
#define blockDim wnsim::bdim
#define WN_DYN_SMEM(name) double* name = wnsim::blk->smem
#define __builtin_amdgcn_readfirstlane(x) wnsim::readfirstlane(x)

template <class T>
inline T __shfl_xor(T v, int off, int = 64) {
  const unsigned base = wnsim::tidx.x & ~63u;
  return wnsim::exchange(v, base + ((wnsim::tidx.x & 63u) ^ static_cast<unsigned>(off)));
}
template <class T>
inline T __shfl(T v, int src, int = 64) {
  return wnsim::exchange(v, (wnsim::tidx.x & ~63u) + static_cast<unsigned>(src));
}
template <class T>
inline T __shfl_up(T v, unsigned delta, int = 64) {  // lanes below `delta` keep their own value, as on the device
  const unsigned l = wnsim::tidx.x & 63u;
  return wnsim::exchange(v, (wnsim::tidx.x & ~63u) + (l >= delta ? l - delta : l));
}
template <class T>
inline T __shfl_down(T v, unsigned delta, int = 64) {
  const unsigned l = wnsim::tidx.x & 63u, top = std::min(63u, wnsim::bdim.x - 1 - (wnsim::tidx.x & ~63u));
  return wnsim::exchange(v, (wnsim::tidx.x & ~63u) + (l + delta <= top ? l + delta : l));
}
inline void __syncthreads() { wnsim::sync(); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
using std::fabs;
using std::fmax;

// ---- the platform layer's names (walnuts_amd/csrc/wn_gfx950.h) over the emulation --------------
namespace wn {
// the emulation pays one OS thread per lane: keep its launches small
constexpr int kSummaryBlock = 64;
constexpr int kSummaryLagSlabChains = 256;
constexpr int kSummaryCandidateCap = 8;
inline int uni(int v) { return wnsim::readfirstlane(v); }
inline double uni(double v) { return wnsim::readfirstlane(v); }
inline void drain_loads() {}
struct ParkedDouble {
  double v;
};
inline void park(ParkedDouble& a, double v) { a.v = v; }
inline double fetch(const ParkedDouble& a) { return a.v; }
struct PlainDouble {
  double v;
};
inline void park(PlainDouble& a, double v) { a.v = v; }
inline double fetch(const PlainDouble& a) { return a.v; }
inline double lane_value(double v, int src_lane) { return __shfl(v, src_lane, 64); }
inline int lane_value(int v, int src_lane) { return __shfl(v, src_lane, 64); }
inline void set_lane(int& reg, int v, int dst_lane) { if (static_cast<int>(wnsim::tidx.x & 63u) == dst_lane) reg = v; }
inline void set_lane(double& reg, double v, int dst_lane) { if (static_cast<int>(wnsim::tidx.x & 63u) == dst_lane) reg = v; }
inline bool either_half(bool c) { return __shfl(static_cast<int>(c), 0, 64) != 0 || __shfl(static_cast<int>(c), 32, 64) != 0; }
inline double wave_sum(double v) {  // xor butterfly, offsets 32,1,2,4,8,16: the device's association order
  v = v + __shfl_xor(v, 32, 64);
  for (int off = 1; off < 32; off <<= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
inline double wave_sum_packed(double a, double b) {  // a's total in lanes 0-31, b's in lanes 32-63
  const double sa = wave_sum(a), sb = wave_sum(b);
  return (wnsim::tidx.x & 63u) < 32u ? sa : sb;
}
inline int opaque_scalar_add(int a, int b) { return a + b; }
inline double lane_below(double x) { return __shfl_up(x, 1u, 64); }
inline double lane_above(double x) { return __shfl_down(x, 1u, 64); }
inline double opaque_uniform(double v) { return v; }
inline void launder(double&) {}
using GlobalBytes = char*;
inline GlobalBytes opaque_scalar_pointer(const void* p) { return const_cast<char*>(static_cast<const char*>(p)); }
inline v2f64 load_pair_at(GlobalBytes base, unsigned byte_offset) { return *reinterpret_cast<const v2f64*>(base + byte_offset); }
inline void store_pair_at(GlobalBytes base, unsigned byte_offset, v2f64 v) { *reinterpret_cast<v2f64*>(base + byte_offset) = v; }
inline int opaque_lane_id() { return static_cast<int>(wnsim::tidx.x & 63u); }
inline int wave_in_workgroup() { return static_cast<int>(wnsim::tidx.x >> 6); }
template <class T>
inline const T& kernel_argument(const T& by_value) { return by_value; }
inline v2f64 stream_load(const v2f64* p) { return *p; }
inline void stream_store(v2f64 v, v2f64* p) { *p = v; }
inline void stream_store(double v, double* p) { *p = v; }
}  // namespace wn

// ---- the sliver of the HIP runtime the host code uses ------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline const char* hipGetErrorString(hipError_t) { return "cpusim error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
inline hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
struct hipDeviceProp_t {
  int multiProcessorCount;
};
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  p->multiProcessorCount = 2;
  return hipSuccess;
}
typedef void* hipStream_t;
constexpr unsigned hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = reinterpret_cast<void*>(0x1);
  return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
struct hipEventSim {
  std::chrono::steady_clock::time_point t;
};
typedef hipEventSim* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new hipEventSim();
  return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  *free_b = size_t{1} << 30;
  *total_b = size_t{1} << 31;
  return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) {
  *p = std::aligned_alloc(64, ((n + 63) / 64) * 64);
  std::memset(*p, 0xCD, n);
  return *p ? hipSuccess : 1;
}
inline hipError_t hipFree(void* p) {
  std::free(p);
  return hipSuccess;
}
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memcpy(d, s, n);
  return hipSuccess;
}
constexpr hipError_t hipErrorPeerAccessAlreadyEnabled = 704;
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) {
  *can = 1;
  return hipSuccess;
}
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) {
  std::memcpy(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height,
                                   hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r)
    std::memcpy(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  std::memset(d, v, n);
  return hipSuccess;
}
constexpr unsigned hipHostMallocDefault = 0;
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
  *p = std::aligned_alloc(64, ((n + 63) / 64) * 64);
  return *p ? hipSuccess : 1;
}
inline hipError_t hipHostFree(void* p) {
  std::free(p);
  return hipSuccess;
}
constexpr unsigned hipHostRegisterDefault = 0;
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
  wnsim::launch(kernel, grid, block, static_cast<size_t>(smem), __VA_ARGS__)

// wn_sample.hip -- walnutpie_sample_device(): the device-model sibling of the reference's ctypes
// entry point walnutpie_sample_cfunc (python/src/walnutpie/walnutpy.cpp:134-222 -> run_sampler
// :20-84 -> walnutpie::walnuts, api.hpp:35-69), built on the batched engine.
//
// What is kept: argument list after the model, validation and error types, output layout
// out[C][max_sampling + max_warmup*save_warmup][num_params] written chain-major, final_lengths
// [C warmup rows | C sampling rows], stepsize_out[C], inv_metric_out[C][num_params], progress lines.
// Initial positions and the step-size search draw from the engine's counter-based streams on the device; in
// walnutpie_sample_device_reference_streams they consume the same libstdc++ streams as the reference
// (seed_seq{seed,1} and seed_seq{seed,2}; walnutpy.cpp:187-189,75-76), generated on the host, so that there
// those inputs are bit-identical to the reference's.
// What differs (documented in INTEGRATION.md): chains advance in lock step, so the controllers' stopping
// rules (adapt.hpp:172-229, sampler.hpp:117-158) are evaluated on whole iterations and every chain gets the same
// length (the reference's thread-per-chain workers stop wherever they happen to be), and the
// per-chain trajectory randomness comes from the counter-based generator keyed by seed+id+num_chains
// (walnutpy.cpp:82) instead of mt19937_64.
#include "wn_hip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <csignal>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iomanip>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include "../../include/walnuts_hip.h"
#include "wn_refstream.h"

extern "C" int wn_engine_adapt_step_with_normals(wn_engine* e, const double* normals, WalnutpyError** err);
extern "C" void* wn_internal_make_error(const char* msg, int type);

namespace {

struct EngineGuard {
  wn_engine* e = nullptr;
  ~EngineGuard() {
    if (e) wn_engine_destroy(e);
  }
};

[[noreturn]] void rethrow(WalnutpyError* err) {
  std::string msg = walnutpie_get_error_message(err);
  const WalnutpyErrorType t = walnutpie_get_error_type(err);
  walnutpie_destroy_error(err);
  if (t == config) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
#define WN_CALL(expr)                   \
  do {                                  \
    WalnutpyError* call_err_ = nullptr; \
    if ((expr) != 0) rethrow(call_err_); \
  } while (0)

void finite_positive(double v, const char* name) {  // validate.hpp: validate_finite_positive
  if (!(std::isfinite(v) && v > 0)) throw std::invalid_argument(std::string(name) + " must be finite and > 0");
}
void probability(double v, const char* name) {  // validate.hpp: validate_probability
  if (!(v > 0 && v < 1)) throw std::invalid_argument(std::string(name) + " must be in (0, 1)");
}

// python/src/walnutpie/interrupts.hpp:34-102: while a sampling call runs, SIGINT sets a flag instead of killing
// the process (SA_RESETHAND: a second Ctrl-C gets the previous disposition back); the previous handler is restored
// on the way out.  The reference's controllers poll the flag (adapt.hpp:227, sampler.hpp:154); here it is polled
// between the launches of two iterations, and a raised flag ends the call with error type `interrupt`
// (errors.hpp:42-47: an empty message).
struct InterruptException {};
volatile std::sig_atomic_t wn_interrupted = 0;
class InterruptGuard {
 public:
  InterruptGuard() {
    wn_interrupted = 0;
    std::memset(&custom_, 0, sizeof(custom_));
    sigemptyset(&custom_.sa_mask);
    sigaddset(&custom_.sa_mask, SIGINT);
    custom_.sa_flags = SA_RESETHAND;
    custom_.sa_handler = &InterruptGuard::on_signal;
    sigaction(SIGINT, &custom_, &before_);
  }
  ~InterruptGuard() { sigaction(SIGINT, &before_, nullptr); }
  InterruptGuard(const InterruptGuard&) = delete;
  InterruptGuard& operator=(const InterruptGuard&) = delete;
  void throw_if_interrupted() const {
    if (wn_interrupted) throw InterruptException{};
  }

 private:
  static void on_signal(int) { wn_interrupted = 1; }
  struct sigaction before_, custom_;
};

// ---- small RAII holders: a half-built sink releases what it had (no leak when a later allocation fails) ----
struct DevBlock {
  double* p = nullptr;
  DevBlock() = default;
  DevBlock(const DevBlock&) = delete;
  DevBlock& operator=(const DevBlock&) = delete;
  ~DevBlock() { reset(); }
  bool alloc(size_t doubles) {
    reset();
    return hipMalloc(reinterpret_cast<void**>(&p), doubles * sizeof(double)) == hipSuccess;
  }
  void reset() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  double* release() {
    double* q = p;
    p = nullptr;
    return q;
  }
};
struct Event {
  hipEvent_t e = nullptr;
  Event() = default;
  Event(const Event&) = delete;
  Event& operator=(const Event&) = delete;
  ~Event() {
    if (e) (void)hipEventDestroy(e);
  }
  void create() {
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) throw std::runtime_error("cannot create a HIP event");
  }
};
struct Stream {
  hipStream_t s = nullptr;
  Stream() = default;
  Stream(const Stream&) = delete;
  Stream& operator=(const Stream&) = delete;
  ~Stream() {
    if (s) (void)hipStreamDestroy(s);
  }
  void create() {
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("cannot create a HIP stream");
  }
};
// The caller's output buffer, page-locked for the duration of the call (opt-in: WALNUTS_AMD_PIN_OUTPUT=1).  A
// device-to-host copy into pageable memory goes through the runtime's bounce buffers and blocks the host thread; into
// registered memory it is a DMA that overlaps the launches.  Measured at 65 536 x 1 024 x 32 draws = 16 GiB
// (profiles/r03/sample_device_e2e.txt): pageable 1.0 s for the sampling phase (16 GB/s, the host blocked), registered
// 0.37 s (46 GB/s, overlapped) -- after 0.7 s spent registering 16 GiB of fresh pages.  The registration runs on the
// call's preparation thread (sample_device_impl), but page-locking 16 GiB contends with the host-side initial streams
// for the process's address space (they take 1.1 s beside it instead of 0.4 s): 1.88 s for the whole call against
// 1.77 s pageable.  A loss for a buffer used once, a gain for a caller that reuses a registered buffer: hence opt-in.
// Best effort (RLIMIT_MEMLOCK, memory the caller registered already): on failure the pageable path is used.
struct PinnedRange {
  void* p = nullptr;
  PinnedRange(void* ptr, size_t bytes) {
    const char* env = std::getenv("WALNUTS_AMD_PIN_OUTPUT");
    if (ptr == nullptr || bytes == 0 || env == nullptr || env[0] != '1') return;
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess) p = ptr;
    else (void)hipGetLastError();
  }
  PinnedRange(const PinnedRange&) = delete;
  PinnedRange& operator=(const PinnedRange&) = delete;
  ~PinnedRange() {
    if (p) (void)hipHostUnregister(p);
  }
};

// A caller's fresh buffer (numpy.empty: 16 GiB at the headline size) has no pages yet: the device-to-host copies then
// take every first-touch fault on the runtime's one staging thread -- 11.5 GB/s into fresh memory against 21.6 GB/s
// into touched memory on the GPU box (tests/gpu_probes/d2h_probe.hip).  A few helper threads populate the range with
// madvise(MADV_POPULATE_WRITE) -- which faults pages in WITHOUT changing their content, so it can run beside the
// copies and beside whatever the caller keeps in rows this call never writes -- while the engine is created and the
// warmup runs.  Best effort: an older kernel (EINVAL) or an odd mapping simply leaves the faults to the copies.
// Measured in fresh processes, 65 536 x 1 024, 20 + 32 iterations, 16 GiB of draws (profiles/r04/prefault_ab.txt):
// the call takes 0.70-0.76 s with the helpers (8 threads, 256 MiB per madvise; started after the device allocations)
// against 1.07-2.05 s without; smaller slices lose the gain (2 MiB: 1.06 s).
class Prefault {
 public:
  // `share`: the fraction of the process's helper budget this call may use (a shard of a multi-device call: its
  // share of the chains)
  Prefault(void* ptr, size_t bytes, double share = 1.0) {
#if defined(MADV_POPULATE_WRITE) || defined(__linux__)
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
    const size_t page = static_cast<size_t>(sysconf(_SC_PAGESIZE));
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(ptr) + page - 1) / page * page;
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(ptr) + bytes) / page * page;
    if (ptr == nullptr || hi <= lo || hi - lo < (size_t{64} << 20)) return;
    if (const char* env = std::getenv("WALNUTS_AMD_NO_PREFAULT"))
      if (env[0] == '1') return;
    cpu_set_t set;
    int cores = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 1;
    int n = std::max(1, static_cast<int>(std::min(8, cores / 2) * share + 0.5));
    if (const char* env = std::getenv("WALNUTS_AMD_PREFAULT_THREADS")) n = std::max(1, std::atoi(env));
    const size_t pages = (hi - lo) / page;
    for (int t = 0; t < n; ++t) {
      const uintptr_t a = lo + pages * static_cast<size_t>(t) / static_cast<size_t>(n) * page;
      const uintptr_t b = lo + pages * static_cast<size_t>(t + 1) / static_cast<size_t>(n) * page;
      try {
      workers_.emplace_back([a, b, this] {
        // in slices, so that a call that ends early (an error, Ctrl-C) does not wait for gigabytes of faults
        uintptr_t slice = uintptr_t{256} << 20;
        if (const char* env = std::getenv("WALNUTS_AMD_PREFAULT_SLICE_KB")) {
          // (at least one page: a zero or non-numeric value would leave the loop below spinning in place)
          const long long kb = std::atoll(env);
          slice = std::max<uintptr_t>(static_cast<uintptr_t>(sysconf(_SC_PAGESIZE)),
                                      static_cast<uintptr_t>(kb > 0 ? kb : 0) << 10);
        }
        for (uintptr_t p = a; p < b && !stop_.load(std::memory_order_relaxed); p += slice)
          if (madvise(reinterpret_cast<void*>(p), static_cast<size_t>(std::min(slice, b - p)), MADV_POPULATE_WRITE) != 0) return;
      });
      } catch (...) {  // no thread to be had: the pages are then populated by whoever writes them first
        break;
      }
    }
#else
    (void)ptr, (void)bytes;
#endif
  }
  Prefault(const Prefault&) = delete;
  Prefault& operator=(const Prefault&) = delete;
  ~Prefault() {  // the helpers never outlive the call: the caller may free the buffer right after it
    stop_.store(true, std::memory_order_relaxed);
    for (auto& w : workers_) w.join();
  }

 private:
  std::vector<std::thread> workers_;
  std::atomic<bool> stop_{false};
};

// Device-to-host at the PCIe rate into a PAGEABLE destination.  A copy straight into pageable memory is staged by the
// runtime on one thread: 21-34 GB/s into populated pages on the GPU boxes.  Here the device writes contiguous chain
// ranges of a staging block into a small ring of pinned chunks (plain DMA: 53 GB/s) and a few host threads scatter each
// chunk's rows into the caller's out[C][rows][D] while the next chunk is on its way (tests/gpu_probes/d2h_probe.hip:
// 16 GiB in 0.32 s against 0.50 s into populated pages).  One dispatcher thread queues the chunk copies -- it is the one
// that waits for a free chunk, so the sampling loop never does --, the workers wait for a chunk's event and copy its
// rows, taking the first-touch faults of a fresh buffer in parallel.  The whole call, 65 536 x 1 024, 20 + 32
// iterations, 16 GiB of draws into a fresh numpy buffer, fresh processes (profiles/r04/prefault_ab.txt): 0.50-0.57 s,
// against 0.70-0.76 s for one strided copy into a buffer populated by helper threads and 1.07-2.05 s for round 3's
// plain copy; two workers 0.73-0.86 s, eight no better than four.
class BounceRing {
 public:
  struct Job {
    const double* block;  // [C][span][D] on the device
    size_t span, fill, first;  // rows per chain in the block, valid rows, first destination row
    hipEvent_t drained;        // recorded behind the job's last chunk copy
    uint64_t ticket;
  };
  BounceRing(int device, size_t chains, size_t rows, size_t dim, double* out, size_t span, hipStream_t copy)
      : device_(device), C_(chains), rows_(rows), D_(dim), out_(out), copy_(copy) {
    const size_t per_chain = span * D_ * sizeof(double);
    cc_ = std::max<size_t>(1, std::min(C_, kChunkBytes / std::max<size_t>(1, per_chain)));
    // one chain's rows alone are larger than a chunk (the ring would pin kRing times that): direct copies
    if (cc_ * per_chain > kChunkBytes) return;
    for (int k = 0; k < kRing; ++k) {
      void* p = nullptr;
      if (hipHostMalloc(&p, cc_ * per_chain, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        release();
        return;
      }
      pin_[k] = static_cast<double*>(p);
      if (hipEventCreateWithFlags(&ev_[k], hipEventDisableTiming) != hipSuccess) {
        release();
        return;
      }
    }
    int nworkers = kWorkers;
    if (const char* env = std::getenv("WALNUTS_AMD_BOUNCE_WORKERS")) nworkers = std::max(1, std::atoi(env));
    try {
      dispatcher_ = std::thread([this] { dispatch(); });
      for (int w = 0; w < nworkers; ++w) workers_.emplace_back([this] { work(); });
      usable_ = true;
    } catch (...) {
      // a thread could not be started: stop and join the ones that were (a joinable std::thread must not be destroyed)
      // and leave the ring unusable -- the sink then copies directly, as it does for outputs too small for the ring
      shut_down(true);
      release();
    }
  }
  BounceRing(const BounceRing&) = delete;
  BounceRing& operator=(const BounceRing&) = delete;
  ~BounceRing() {
    shut_down(true);
    (void)hipStreamSynchronize(copy_);  // (no chunk is a DMA target any more when it is freed)
    release();
  }
  bool usable() const { return usable_; }
  // sampling thread: the copy stream has been ordered behind the launches that filled the block
  uint64_t submit(Job j) {
    std::unique_lock<std::mutex> lk(mu_);
    rethrow_locked();
    j.ticket = ++submitted_;
    jobs_.push_back(j);
    cv_.notify_all();
    return j.ticket;
  }
  // sampling thread: the job's `drained` event has been recorded (it can be waited for on a stream now)
  void wait_recorded(uint64_t ticket) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return recorded_ >= ticket || error_; });
    rethrow_locked();
  }
  // every submitted row is in the caller's buffer when this returns
  void finish() {
    shut_down(false);
    std::unique_lock<std::mutex> lk(mu_);
    rethrow_locked();
  }

 private:
  struct Task {
    int slot;
    size_t c0, c1, first, fill;
  };
  static constexpr size_t kChunkBytes = size_t{64} << 20;
  static constexpr int kRing = 8, kWorkers = 4;
  void rethrow_locked() {
    if (error_) std::rethrow_exception(error_);
  }
  void fail(std::exception_ptr e) {
    std::lock_guard<std::mutex> lk(mu_);
    if (!error_) error_ = e;
    abort_ = true;
    cv_.notify_all();
  }
  void dispatch() {
    try {
      if (hipSetDevice(device_) != hipSuccess) throw std::runtime_error("cannot select the device");
      for (;;) {
        Job j;
        {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return !jobs_.empty() || closing_ || abort_; });
          if (abort_ || jobs_.empty()) return;
          j = jobs_.front();
          jobs_.pop_front();
        }
        for (size_t c0 = 0; c0 < C_; c0 += cc_) {
          const size_t c1 = std::min(C_, c0 + cc_);
          int slot = -1;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] {
              for (int k = 0; k < kRing; ++k)
                if (!slot_busy_[k]) return true;
              return abort_;
            });
            if (abort_) return;
            for (int k = 0; k < kRing; ++k)
              if (!slot_busy_[k]) slot = k;
            slot_busy_[slot] = true;
          }
          // the chunk arrives compact: [c1 - c0][fill][D]
          if (hipMemcpy2DAsync(pin_[slot], j.fill * D_ * sizeof(double), j.block + c0 * j.span * D_,
                               j.span * D_ * sizeof(double), j.fill * D_ * sizeof(double), c1 - c0, hipMemcpyDeviceToHost,
                               copy_) != hipSuccess ||
              hipEventRecord(ev_[slot], copy_) != hipSuccess)
            throw std::runtime_error("copying draws to the host failed");
          std::lock_guard<std::mutex> lk(mu_);
          tasks_.push_back(Task{slot, c0, c1, j.first, j.fill});
          cv_.notify_all();
        }
        if (hipEventRecord(j.drained, copy_) != hipSuccess) throw std::runtime_error("copying draws to the host failed");
        std::lock_guard<std::mutex> lk(mu_);
        recorded_ = j.ticket;
        cv_.notify_all();
      }
    } catch (...) {
      fail(std::current_exception());
    }
  }
  void work() {
    try {
      if (hipSetDevice(device_) != hipSuccess) throw std::runtime_error("cannot select the device");
      for (;;) {
        Task t;
        {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return !tasks_.empty() || workers_closing_ || abort_; });
          if (abort_ || tasks_.empty()) return;
          t = tasks_.front();
          tasks_.pop_front();
        }
        if (hipEventSynchronize(ev_[t.slot]) != hipSuccess) throw std::runtime_error("copying draws to the host failed");
        const double* src = pin_[t.slot];
        for (size_t c = t.c0; c < t.c1; ++c)
          std::memcpy(out_ + c * rows_ * D_ + t.first * D_, src + (c - t.c0) * t.fill * D_, t.fill * D_ * sizeof(double));
        std::lock_guard<std::mutex> lk(mu_);
        slot_busy_[t.slot] = false;
        cv_.notify_all();
      }
    } catch (...) {
      fail(std::current_exception());
    }
  }
  // orderly (every queued job and task is carried out) or, from the destructor of a call that is being unwound, at once
  void shut_down(bool abort_now) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (abort_now && (!jobs_.empty() || !tasks_.empty() || submitted_ != recorded_)) abort_ = true;
      closing_ = true;
      cv_.notify_all();
    }
    if (dispatcher_.joinable()) dispatcher_.join();
    {
      std::lock_guard<std::mutex> lk(mu_);
      workers_closing_ = true;
      cv_.notify_all();
    }
    for (auto& w : workers_)
      if (w.joinable()) w.join();
    workers_.clear();
  }
  void release() {
    for (int k = 0; k < kRing; ++k) {
      if (pin_[k]) (void)hipHostFree(pin_[k]);
      if (ev_[k]) (void)hipEventDestroy(ev_[k]);
      pin_[k] = nullptr;
      ev_[k] = nullptr;
    }
  }
  int device_;
  size_t C_, rows_, D_;
  double* out_;
  hipStream_t copy_;
  size_t cc_ = 1;
  bool usable_ = false;
  double* pin_[kRing] = {};
  hipEvent_t ev_[kRing] = {};
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Job> jobs_;
  std::deque<Task> tasks_;
  bool slot_busy_[kRing] = {};
  uint64_t submitted_ = 0, recorded_ = 0;
  bool closing_ = false, workers_closing_ = false, abort_ = false;
  std::exception_ptr error_;
  std::thread dispatcher_;
  std::vector<std::thread> workers_;
};

// The draw sink (handlers.hpp:63-116 writes every draw straight into the caller's buffer, whatever its size).
// The device writes the draws of up to `span` consecutive iterations of all chains into one of two staging blocks
// [C][span][D]; a full block goes to the caller's out[C][rows][D] on a second stream while the next iterations fill the
// other block, so the device never holds more than two blocks of draws and [C][T][D] may exceed HBM (65 536 chains x
// 1 000 draws x 1 024 = 537 GB).  Large outputs leave through the pinned ring above, small ones (and anything the ring
// cannot take) as ONE strided copy.
class DrawSink {
 public:
  // `rows`: rows per chain of the caller's buffer; `capacity`: how many of them this sink will be asked to write
  // `engine`: whose transition launches fill the blocks.  The copies are ordered against ALL of its launches through
  // the engine (wn_engine_release_stream / _wait_event), not against one stream of it: with chain groups a launch is
  // several kernels on several streams.
  DrawSink(size_t chains, size_t rows, size_t capacity, size_t dim, double* out, wn_engine* engine, int device)
      : C_(chains), rows_(rows), D_(dim), out_(out), engine_(engine) {
    if (C_ * capacity * D_ == 0) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = size_t{1} << 30;
    // Both blocks together: a quarter of what the engine left, and no more than 8 GiB -- a block that holds the whole
    // run is copied only when the run is over (nothing overlaps), and 52 GiB of fresh device memory for two such
    // blocks took the first call of a process ~1 s to map (profiles/r04/prefault_ab.txt).  At the headline size a block
    // is then 8 iterations: one launch of 8 transitions.
    size_t budget = std::min(free_b / 4, size_t{8} << 30);
    if (const char* env = std::getenv("WALNUTS_AMD_DRAW_STAGING_BYTES")) budget = std::strtoull(env, nullptr, 10);
    const size_t per_iter = C_ * D_ * sizeof(double);
    span_ = std::max<size_t>(1, std::min(capacity, budget / 2 / per_iter));
    // other processes or ranks may share the device: on failure halve the span until the two blocks fit
    for (;;) {
      if (block_[0].alloc(C_ * span_ * D_) && block_[1].alloc(C_ * span_ * D_)) break;
      (void)hipGetLastError();
      block_[0].reset();
      block_[1].reset();
      if (span_ == 1) throw std::runtime_error("cannot allocate the device draw staging buffer");
      span_ = (span_ + 1) / 2;
    }
    copy_.create();
    for (int b = 0; b < 2; ++b) drained_[b].create();
    // WALNUTS_AMD_BOUNCE: 1 = the pinned ring whatever the size (tests), 0 = never; default: from 256 MiB of output
    bool ring = C_ * capacity * D_ * sizeof(double) >= (size_t{256} << 20);
    if (const char* env = std::getenv("WALNUTS_AMD_BOUNCE")) ring = env[0] == '1';
    if (ring) {
      ring_ = std::make_unique<BounceRing>(device, C_, rows_, D_, out_, span_, copy_.s);
      if (!ring_->usable()) ring_.reset();
    }
  }
  size_t stride() const { return span_ * D_; }  // doubles between two chains' rows in a staging block
  bool uses_ring() const { return ring_ != nullptr; }
  // where the next iteration's draws go (device pointer of chain 0's row)
  double* next_row() {
    if (fill_ == 0 && busy_[cur_]) {  // the block still feeds a copy: the kernels must not overwrite it yet
      if (ring_) ring_->wait_recorded(ticket_[cur_]);
      WN_CALL(wn_engine_wait_event(engine_, drained_[cur_].e, &call_err_));
      busy_[cur_] = false;
    }
    return block_[cur_].p + fill_ * D_;
  }
  // how many consecutive iterations the current block still takes (a multi-transition launch writes that many rows)
  size_t room() const { return span_ - fill_; }
  // the `n` iterations launched into next_row() (rows D doubles apart) are queued on the compute stream
  void rows_done(size_t n) {
    fill_ += n;
    written_ += n;
    if (fill_ == span_) flush();
  }
  size_t written() const { return written_; }
  // everything written so far is in the caller's buffer when this returns
  void finish() {
    if (fill_ > 0) flush();
    if (ring_) ring_->finish();
    if (copy_.s && hipStreamSynchronize(copy_.s) != hipSuccess) throw std::runtime_error("copying draws to the host failed");
  }

 private:
  void flush() {
    const size_t first = written_ - fill_;
    WN_CALL(wn_engine_release_stream(engine_, copy_.s, &call_err_));  // the copy runs behind every launch made so far
    if (ring_) {
      ticket_[cur_] = ring_->submit(BounceRing::Job{block_[cur_].p, span_, fill_, first, drained_[cur_].e, 0});
    } else if (hipMemcpy2DAsync(out_ + first * D_, rows_ * D_ * sizeof(double), block_[cur_].p, span_ * D_ * sizeof(double),
                                fill_ * D_ * sizeof(double), C_, hipMemcpyDeviceToHost, copy_.s) != hipSuccess ||
               hipEventRecord(drained_[cur_].e, copy_.s) != hipSuccess) {
      throw std::runtime_error("copying draws to the host failed");
    }
    busy_[cur_] = true;
    cur_ ^= 1;
    fill_ = 0;
  }
  size_t C_, rows_, D_;
  double* out_;
  wn_engine* engine_;
  Stream copy_;
  DevBlock block_[2];
  Event drained_[2];
  uint64_t ticket_[2] = {0, 0};
  bool busy_[2] = {false, false};
  size_t span_ = 1, fill_ = 0, written_ = 0;
  int cur_ = 0;
  std::unique_ptr<BounceRing> ring_;  // (declared last: its threads stop before the blocks and the stream go away)
};

// Resident mode (walnutpie_sample_device_resident): the sampling draws stay in one [C][S][D] block in HBM; every
// `thin`-th of them is also copied to the caller's buffer, row by row, on a second stream.
class ResidentDraws {
 public:
  ResidentDraws(size_t chains, size_t max_sampling, size_t dim, int thin, double* out, size_t out_rows,
                size_t out_first_row, wn_engine* engine)
      : C_(chains), S_(max_sampling), D_(dim), thin_(thin), out_(out), out_rows_(out_rows), out_first_(out_first_row),
        engine_(engine) {
    size_t free_b = 0, total_b = 0;
    const size_t bytes = C_ * S_ * D_ * sizeof(double);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b - free_b / 16) {
      std::stringstream ss;
      ss << "the sampling draws (" << C_ << " chains x " << S_ << " draws x " << D_ << " doubles = " << (bytes >> 20)
         << " MiB) do not fit the device's free memory (" << (free_b >> 20) << " MiB): lower max_sampling_iter or use "
         << "walnutpie_sample_device, which streams them to the host";
      throw std::runtime_error(ss.str());
    }
    if (bytes > 0 && !block_.alloc(C_ * S_ * D_)) throw std::runtime_error("cannot allocate the device-resident draw block");
    copy_.create();
  }
  int64_t stride() const { return static_cast<int64_t>(S_ * D_); }
  // the thinned rows follow the warmup rows ACTUALLY written (handlers.hpp:73-89: a chain writes sequentially), which
  // is known once the warmup loop has ended -- before that the controller may still stop it early
  void set_first_row(size_t row) { out_first_ = row; }
  double* next_row() { return block_.p + written_ * D_; }
  void row_done() {
    if (thin_ > 0 && written_ % static_cast<size_t>(thin_) == 0) {
      const size_t k = written_ / static_cast<size_t>(thin_);
      WN_CALL(wn_engine_release_stream(engine_, copy_.s, &call_err_));  // (every chain group's launch, not one stream's)
      if (hipMemcpy2DAsync(out_ + (out_first_ + k) * D_, out_rows_ * D_ * sizeof(double), block_.p + written_ * D_,
                           S_ * D_ * sizeof(double), D_ * sizeof(double), C_, hipMemcpyDeviceToHost,
                           copy_.s) != hipSuccess)
        throw std::runtime_error("copying thinned draws to the host failed");
    }
    ++written_;
  }
  size_t written() const { return written_; }
  void finish() {
    if (hipStreamSynchronize(copy_.s) != hipSuccess) throw std::runtime_error("copying thinned draws to the host failed");
  }
  double* release() { return block_.release(); }

 private:
  size_t C_, S_, D_;
  int thin_;
  double* out_;
  size_t out_rows_, out_first_;
  wn_engine* engine_;
  DevBlock block_;
  Stream copy_;
  size_t written_ = 0;
};

// Bounded run-ahead of the host: the steps are asynchronous, so without this the host queues every launch of the run
// ahead of the GPU -- a Ctrl-C would only be noticed after the whole queue has drained, and progress lines would
// report enqueued, not completed, iterations.  One event per iteration in a small ring; before enqueuing iteration n
// the host waits for iteration n - kDepth.
class RunAhead {
 public:
  static constexpr int kDepth = 4;
  explicit RunAhead(hipStream_t s) : stream_(s) {
    for (auto& e : ring_) e.create();
  }
  void before_enqueue() {
    if (count_ >= kDepth && hipEventSynchronize(ring_[count_ % kDepth].e) != hipSuccess)
      throw std::runtime_error("waiting for an earlier iteration failed");
  }
  void after_enqueue() {
    if (hipEventRecord(ring_[count_ % kDepth].e, stream_) != hipSuccess) throw std::runtime_error("event record failed");
    ++count_;
  }

 private:
  hipStream_t stream_;
  Event ring_[kDepth];
  size_t count_ = 0;
};

struct Printer {  // python/src/walnutpie/handlers.hpp:17-59
  PRINT_CALLBACK print;
  size_t refresh;
  size_t iter = 0;
  bool in_warmup = true;
  void progress(size_t num_chains) {
    ++iter;
    if (refresh == 0 || print == nullptr || iter % refresh != 0) return;
    if (num_chains > 16) {  // lock-step chains: one line for all of them instead of 65 536 callbacks per refresh
      std::stringstream ss;
      ss << "Chains [1-" << num_chains << "]: Iteration " << iter << "\t" << (in_warmup ? "(Warmup)" : "(Sampling)")
         << std::endl;
      const std::string s = ss.str();
      print(s.c_str(), s.length(), false);
      return;
    }
    for (size_t c = 0; c < num_chains; ++c) {
      std::stringstream ss;
      ss << "Chain [" << (c + 1) << "]: Iteration " << iter << "\t" << (in_warmup ? "(Warmup)" : "(Sampling)")
         << std::endl;
      const std::string s = ss.str();
      print(s.c_str(), s.length(), false);
    }
  }
};

}  // namespace

// WALNUTS_AMD_TIMING=1: wall-clock time of the call's phases on stderr (where a drop-in call spends its time)
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  PhaseTimer() : on(std::getenv("WALNUTS_AMD_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[walnuts_amd] %-34s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

constexpr int kMaxFusedTransitions = 8;  // transitions per launch between two looks of a controller

// ---- several devices behind one call (walnutpie_sample_device_multi) ------------------------------------------------
// The chains shard embarrassingly (adapt.hpp:257-258: nothing is pooled per transition): shard s is a contiguous block
// of GLOBAL chain ids on devices[s], driven by its own host thread, engine and stream, and writes its own slice of the
// caller's out[C][T][D] -- no exchange on the data path.  What the shards share is what the reference's controller
// threads look at: the warmup spread (adapt.hpp:193-221) and R-hat of the log density (sampler.hpp:139-145), each
// reduced in the two stages the engine exposes (wn_engine_warmup_sums / _warmup_max_rel, wn_engine_lp_sums / _lp_sq_dev)
// with a rendezvous of the shard threads in between, so that every shard takes the same stopping decision at the same
// iteration.
struct ShardAborted {};  // another shard failed: leave quietly, the wrapper reports that shard's error
class Coordinator {
 public:
  Coordinator(int shards, size_t dims) : n_(shards), D_(dims), slots_(static_cast<size_t>(shards) * (dims + 4), 0.0) {}
  // a failing shard stops taking part; the others notice at their next rendezvous
  void abandon() {
    std::lock_guard<std::mutex> lk(mu_);
    failed_ = true;
    --n_;
    if (arrived_ >= n_ && n_ > 0) release();
    cv_.notify_all();
  }
  bool failed() const { return failed_; }
  // adapt.hpp:193-221 over ALL shards' chains
  bool warmup_converged(int shard, wn_engine* e, size_t total_chains, double step_tol, double mass_tol) {
    double* mine = slot(shard);
    WN_CALL(wn_engine_warmup_sums(e, mine, mine + 1, &call_err_));
    rendezvous();
    std::vector<double> total(D_ + 1, 0.0);
    for (int s = 0; s < shards(); ++s)
      for (size_t i = 0; i <= D_; ++i) total[i] += slot(s)[i];  // (every shard adds in the same order: same total)
    rendezvous();  // everyone has read the sums before the slots are reused
    double rel_step = 0, rel_mass = 0;
    WN_CALL(wn_engine_warmup_max_rel(e, total[0], total.data() + 1, total_chains, &rel_step, &rel_mass, &call_err_));
    mine[0] = rel_step;
    mine[1] = rel_mass;
    rendezvous();
    double ms = 0, mm = 0;
    for (int s = 0; s < shards(); ++s) {
      ms = std::max(ms, slot(s)[0]);
      mm = std::max(mm, slot(s)[1]);
    }
    rendezvous();
    return mm <= mass_tol && ms <= step_tol;
  }
  // sampler.hpp:139-145 over ALL shards' chains
  double rhat(int shard, wn_engine* e) {
    double* mine = slot(shard);
    WN_CALL(wn_engine_lp_sums(e, mine, &call_err_));  // sum of means, sum of sample variances, chains
    rendezvous();
    double s0 = 0, s1 = 0, s2 = 0;
    for (int s = 0; s < shards(); ++s) {
      s0 += slot(s)[0];
      s1 += slot(s)[1];
      s2 += slot(s)[2];
    }
    rendezvous();
    double q = 0;
    WN_CALL(wn_engine_lp_sq_dev(e, s0 / s2, &q, &call_err_));
    mine[0] = q;
    rendezvous();
    double qq = 0;
    for (int s = 0; s < shards(); ++s) qq += slot(s)[0];
    rendezvous();
    const double variance_of_means = qq / (s2 - 1);  // util.hpp:401-404
    const double mean_of_variances = s1 / s2;
    return std::sqrt(1 + variance_of_means / mean_of_variances);  // sampler.hpp:145
  }

 private:
  int shards() const { return total_shards_; }
  double* slot(int s) { return slots_.data() + static_cast<size_t>(s) * (D_ + 4); }
  void release() {
    arrived_ = 0;
    ++generation_;
  }
  void rendezvous() {
    std::unique_lock<std::mutex> lk(mu_);
    if (failed_) throw ShardAborted{};
    const unsigned long gen = generation_;
    if (++arrived_ >= n_) {
      release();
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return generation_ != gen || failed_; });
    }
    if (failed_) throw ShardAborted{};
  }
  int n_;
  const int total_shards_ = n_;
  size_t D_;
  std::vector<double> slots_;
  std::mutex mu_;
  std::condition_variable cv_;
  int arrived_ = 0;
  unsigned long generation_ = 0;
  std::atomic<bool> failed_{false};
};
struct ShardCtx {
  int shard = 0, device = 0;
  size_t chain_begin = 0, total_chains = 0;
  Coordinator* coord = nullptr;
  const InterruptGuard* interrupt = nullptr;  // the call's one SIGINT guard
  int* lengths_warmup = nullptr;              // final_lengths slices of this shard
  int* lengths_sampling = nullptr;
};

struct ResidentRequest {  // walnutpie_sample_device_resident
  int thin;
  wn_chains** chains_out;
  bool all_gather = false;  // multi-device: chains_out is an array of num_devices handles, every device gets the whole block
};

static int sample_device_impl(
    bool reference_streams, const ResidentRequest* resident, const ShardCtx* shard,
    int model, const double* model_params, int num_params, const double* inits, size_t num_chains,
    unsigned int seed, unsigned int id, double init_radius, const double* init_inv_metric, int min_warmup_iter,
    int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,
    int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,
    double mass_converge_tol, double rhat_converge_tol, double mass_init_count, double mass_additive_smoothing,
    double max_macro_steps_target, double step_size_init, double step_accept_rate_target,
    double step_learning_rate, double step_gradient_decay, double step_sq_gradient_decay,
    double step_stabilization, double step_learn_rate_decay, bool save_warmup, double* out, size_t out_size,
    int* final_lengths, double* stepsize_out, double* inv_metric_out, int refresh, PRINT_CALLBACK print,
    WalnutpyError** err) {
  try {
    // walnutpy.cpp:151-160
    if (refresh < 0) {
      std::stringstream msg;
      msg << "refresh must be non-negative, was " << refresh;
      throw std::invalid_argument(msg.str());
    }
    if (num_params < 1) throw std::invalid_argument("num_params must be in {1, 2, ... }");
    if (max_sampling_iter < 0 || max_warmup_iter < 0) throw std::invalid_argument("iteration counts must be >= 0");
    const size_t warm_rows = save_warmup ? static_cast<size_t>(max_warmup_iter) : 0;
    size_t samp_rows = static_cast<size_t>(max_sampling_iter);  // sampling rows per chain in the caller's buffer
    if (resident != nullptr) {
      if (resident->thin < 0) throw std::invalid_argument("thin must be non-negative");
      if (resident->chains_out == nullptr) throw std::invalid_argument("chains_out must not be null");
      if (max_sampling_iter < 1) throw std::invalid_argument("resident draws need max_sampling_iter >= 1");
      *resident->chains_out = nullptr;
      const size_t t = static_cast<size_t>(resident->thin);
      samp_rows = t == 0 ? 0 : (static_cast<size_t>(max_sampling_iter) + t - 1) / t;
    }
    const size_t rows = samp_rows + warm_rows;
    if (rows > 0 && out == nullptr) throw std::invalid_argument("out must not be null");
    const size_t draws_offset = static_cast<size_t>(num_params) * rows;
    // (a shard of walnutpie_sample_device_multi sees its own slice of a buffer the wrapper has checked as a whole)
    const size_t total_chains = shard != nullptr ? shard->total_chains : num_chains;
    const size_t chain_begin = shard != nullptr ? shard->chain_begin : 0;
    if (shard == nullptr && out_size < num_chains * draws_offset) {
      std::stringstream ss;
      ss << "Output buffer too small. Expected at least " << num_chains << " chains of " << draws_offset
         << " doubles, got " << out_size;
      throw std::runtime_error(ss.str());
    }
    // WarmupConfigBuilder / SamplingConfigBuilder validation (config.hpp:656-850, 978-1059)
    if (static_cast<size_t>(min_warmup_iter) > static_cast<size_t>(max_warmup_iter))
      throw std::invalid_argument("min_iter cannot be greater than than max_iter");
    finite_positive(step_size_converge_tol, "step_size_converge_tol");
    finite_positive(mass_converge_tol, "mass_converge_tol");
    finite_positive(mass_init_count, "mass_init_count");
    finite_positive(mass_additive_smoothing, "mass_additive_smoothing");
    finite_positive(max_macro_steps_target, "max_macro_steps_target");
    probability(step_accept_rate_target, "step_accept_rate_target");
    finite_positive(step_learning_rate, "step_learning_rate");
    probability(step_gradient_decay, "step_gradient_decay");
    probability(step_sq_gradient_decay, "step_sq_gradient_decay");
    finite_positive(step_stabilization, "step_stabilization");
    probability(step_learn_rate_decay, "step_learn_rate_decay");
    if (static_cast<size_t>(min_sampling_iter) > static_cast<size_t>(max_sampling_iter))
      throw std::invalid_argument("min_iter must be <= max_iter");
    if (!(std::isfinite(rhat_converge_tol) && rhat_converge_tol > 1))
      throw std::invalid_argument("rhat_convergence_tol must be finite and > 1");
    finite_positive(max_hamiltonian_error, "max_hamiltonian_error");
    if (min_micro_steps < 1) throw std::invalid_argument("min_micro_steps must be in {1, 2, ... }");
    finite_positive(step_size_init, "step size");  // config.hpp:222

    wn_config cfg;
    wn_default_config(&cfg);
    cfg.max_trajectory_doublings = max_trajectory_doublings;
    cfg.max_step_halvings = max_step_halvings;
    cfg.min_micro_steps = min_micro_steps;
    cfg.max_hamiltonian_error = max_hamiltonian_error;
    cfg.mass_init_count = mass_init_count;
    cfg.max_macro_steps_target = max_macro_steps_target;
    cfg.step_accept_rate_target = step_accept_rate_target;
    cfg.step_learning_rate = step_learning_rate;
    cfg.step_gradient_decay = step_gradient_decay;
    cfg.step_sq_gradient_decay = step_sq_gradient_decay;
    cfg.step_stabilization = step_stabilization;
    cfg.step_learn_rate_decay = step_learn_rate_decay;
    // the entry point that reproduces the reference's streams also keeps the reference's element-wise arithmetic
    // (x86-64 -O3: every product rounded), independent of the process environment
    if (reference_streams) cfg.fused_multiply_add = 0;
    if (shard != nullptr) cfg.device = shard->device;

    PhaseTimer timer;
    // (declared first of the call's resources: destroyed last, after the copies into the buffer have been waited for;
    // started once the device allocations are done -- see below)
    std::unique_ptr<Prefault> populate;
    EngineGuard guard;
    WN_CALL(wn_engine_create(&guard.e, model, num_params, model_params, num_chains, &cfg, &call_err_));
    wn_engine* e = guard.e;
    const size_t D = static_cast<size_t>(num_params);
    timer.mark("engine created");
    // Preparation thread: what the iterations need that does not depend on the host streams -- the draw staging blocks
    // (or the resident draw block) allocated: 0.6-1.0 s for 16-32 GiB of fresh device memory at the headline size; the
    // caller's buffer registered if asked for -- beside the ~0.4 s the streams take on this thread.
    const hipStream_t compute = reinterpret_cast<hipStream_t>(wn_engine_stream(e));
    std::unique_ptr<PinnedRange> pinned;
    std::unique_ptr<DrawSink> sink_holder;
    std::unique_ptr<ResidentDraws> kept;
    std::exception_ptr prep_error;
    std::thread prep([&] {
      try {
        if (hipSetDevice(cfg.device) != hipSuccess) throw std::runtime_error("cannot select the device");
        pinned = std::make_unique<PinnedRange>(out, num_chains * draws_offset * sizeof(double));
        sink_holder = std::make_unique<DrawSink>(num_chains, rows, resident != nullptr ? warm_rows : rows, D, out, e, cfg.device);
        if (resident != nullptr)
          kept = std::make_unique<ResidentDraws>(num_chains, static_cast<size_t>(max_sampling_iter), D, resident->thin,
                                                 out, rows, warm_rows, e);
      } catch (...) {
        prep_error = std::current_exception();
      }
    });
    struct PrepJoiner {
      std::thread& t;
      ~PrepJoiner() {
        if (t.joinable()) t.join();
      }
    } join_prep{prep};

    // The reference's two host streams (wn_refstream.h): the step-size search's normals -- mt19937_64(seed_seq{seed, 2}),
    // the engine shared by the chains in order, a fresh normal distribution per chain (walnutpy.cpp:75-80, util.hpp:288)
    // -- are produced by a second thread while this one produces the initial positions; both hand the non-sequential
    // half of the work to the same worker pool.
    // -- ONLY in walnutpie_sample_device_reference_streams.  walnutpie_sample_device / _resident draw both from the
    // counter-based generator on the device (wn_init.h: streams kStreamInitPos / kStreamInitStep keyed by `seed` and the
    // chain id), as they draw the trajectories' variates: one sequential mt19937_64 for 67 M polar-method normals was
    // 0.4-0.5 s of a 1.0 s call at 65 536 x 1 024 (profiles/r03/sample_device_e2e.txt) to reproduce the first two of
    // the reference's streams bit for bit in a mode whose third stream differs anyway.
    std::unique_ptr<wnref::Workers> pool;
    std::vector<double> z;
    std::thread step_stream;
    struct Joiner {
      std::thread& t;
      ~Joiner() {
        if (t.joinable()) t.join();
      }
    } join_step_stream{step_stream};
    if (reference_streams) {
      pool = std::make_unique<wnref::Workers>(wnref::usable_threads());
      z.resize(num_chains * D);
      step_stream = std::thread([&] {
        std::seed_seq ss{seed, 2u};
        std::mt19937_64 rng(ss);
        wnref::polar_stream_fill(rng, *pool, 1.0, num_chains, D, /*fresh_per_chain=*/true, z.data());
      });
    }
    // initial positions (walnutpy.cpp:176-190)
    if (inits != nullptr) {
      for (size_t i = 0; i < num_chains * D; ++i)
        if (!std::isfinite(inits[i])) throw std::invalid_argument("positions must be finite");
      WN_CALL(wn_engine_set_positions(e, inits, &call_err_));
    } else if (reference_streams) {
      finite_positive(init_radius, "init_scale");
      std::vector<double> pos(num_chains * D);
      std::seed_seq ss{seed, 1u};
      std::mt19937_64 rng(ss);
      // one detail::Random -- one normal distribution -- for all chains (config.hpp:261-266); x *= init_radius
      wnref::polar_stream_fill(rng, *pool, init_radius, num_chains, D, /*fresh_per_chain=*/false, pos.data());
      WN_CALL(wn_engine_set_positions(e, pos.data(), &call_err_));
    } else {
      finite_positive(init_radius, "init_scale");
      // config.hpp:258-268, all chains at once; the stream is keyed by the GLOBAL chain id
      WN_CALL(wn_engine_init_positions(e, seed, static_cast<uint32_t>(chain_begin), init_radius, &call_err_));
    }
    timer.mark(reference_streams ? "initial positions (host stream)" : "initial positions");
    // masses (walnutpy.cpp:64-73).  NB the reference hands init_inv_metric to the builder's
    // masses(): reproduced as is.
    if (init_inv_metric != nullptr) {
      WN_CALL(wn_engine_set_masses(e, init_inv_metric, &call_err_));
    } else {
      WN_CALL(wn_engine_init_masses_from_grad(e, mass_additive_smoothing, &call_err_));
    }
    {
      std::vector<double> steps(num_chains, step_size_init);
      WN_CALL(wn_engine_set_step_sizes(e, steps.data(), &call_err_));
    }
    timer.mark("initial masses");
    // adapt_step_build (walnutpy.cpp:75-80): on the normals the second thread produced, or with the device's own
    if (reference_streams) {
      if (step_stream.joinable()) step_stream.join();
      pool->wait_idle();
      WN_CALL(wn_engine_adapt_step_with_normals(e, z.data(), &call_err_));
      std::vector<double>().swap(z);
    } else {
      WN_CALL(wn_engine_adapt_step(e, seed, static_cast<uint32_t>(chain_begin), &call_err_));
    }
    timer.mark(reference_streams ? "step-size search (host stream)" : "step-size search");
    // walnutpy.cpp:82: walnuts<mt19937_64>(seed + id + num_chains, ...)
    if (reference_streams) {
      WN_CALL(wn_engine_seed_reference_streams(e, static_cast<uint64_t>(seed) + id + num_chains, &call_err_));
    } else {
      WN_CALL(wn_engine_seed(e, static_cast<uint64_t>(seed) + id + total_chains, static_cast<uint32_t>(chain_begin),
                             &call_err_));
    }

    // walnutpy.cpp: interrupt::walnutpy_interrupt_handler on the stack of the call (one guard for all shards)
    std::unique_ptr<InterruptGuard> own_guard;
    if (shard == nullptr || shard->interrupt == nullptr) own_guard = std::make_unique<InterruptGuard>();
    const InterruptGuard& interrupt = own_guard ? *own_guard : *shard->interrupt;
    timer.mark("chains seeded");
    prep.join();
    if (prep_error) std::rethrow_exception(prep_error);
    DrawSink& sink = *sink_holder;
    RunAhead pace(compute);
    timer.mark("preparation thread joined (output registered, draw blocks allocated)");
    // The helpers start HERE, not at the call's entry: page population and hipMalloc both go through the process's
    // address-space lock, and with the helpers running the engine's and the staging blocks' allocations took 0.9 s
    // instead of 0.03 s (profiles/r04/prefault_ab.txt).  From here on the call only launches kernels and copies.
    // (Only for the sink's direct copies: the pinned ring's scatter threads take their first-touch faults in parallel
    // by themselves, and the helpers beside them only contend -- 0.80-0.83 s with both against 0.51 s with the ring
    // alone, same file.)
    if (!sink.uses_ring())
      populate = std::make_unique<Prefault>(out, num_chains * draws_offset * sizeof(double),
                                            static_cast<double>(num_chains) / static_cast<double>(std::max<size_t>(1, total_chains)));
    // (progress lines: shard 0 speaks for all chains)
    Printer printer{shard != nullptr && shard->shard != 0 ? nullptr : print, static_cast<size_t>(refresh)};
    // Consecutive iterations between two looks of a controller go out as ONE launch (wn_engine_*_steps: the workgroup
    // that fetched a chain runs them back to back -- the chains are independent, adapt.hpp:116-127 / sampler.hpp:82-93
    // are per-chain loops): the launch and its tail, the last chains finishing while the chip drains, are paid once
    // per launch.  Host-fed reference streams cover one transition per launch.
    const int fuse_limit = reference_streams ? 1 : kMaxFusedTransitions;
    constexpr int publish_stride = 5;  // adapt.hpp: snapshots every 5 iterations
    for (int it = 0; it < max_warmup_iter;) {  // AdaptWorker loop, adapt.hpp:116-127
      interrupt.throw_if_interrupted();
      pace.before_enqueue();
      int n = std::min({fuse_limit, max_warmup_iter - it, publish_stride - it % publish_stride});
      if (save_warmup) n = static_cast<int>(std::min<size_t>(static_cast<size_t>(n), sink.room()));
      double* dst = save_warmup ? sink.next_row() : nullptr;
      WN_CALL(wn_engine_warmup_steps(e, n, dst, static_cast<int64_t>(sink.stride()), static_cast<int64_t>(D), &call_err_));
      if (save_warmup) sink.rows_done(static_cast<size_t>(n));
      pace.after_enqueue();
      for (int k = 0; k < n; ++k) {  // (a Ctrl-C raised from inside a progress callback ends the call at once)
        printer.progress(total_chains);
        interrupt.throw_if_interrupted();
      }
      it += n;
      // controller_loop (adapt.hpp:172-229) on the snapshots published every publish_stride = 5 iterations
      if (it >= min_warmup_iter && it < max_warmup_iter && it % publish_stride == 0) {
        if (shard != nullptr) {
          if (shard->coord->warmup_converged(shard->shard, e, total_chains, step_size_converge_tol, mass_converge_tol)) break;
        } else {
          double rel_step = 0, rel_mass = 0;
          WN_CALL(wn_engine_warmup_spread(e, &rel_step, &rel_mass, &call_err_));
          if (rel_mass <= mass_converge_tol && rel_step <= step_size_converge_tol) break;
        }
      }
    }
    const size_t written_warmup = sink.written();
    if (kept) kept->set_first_row(written_warmup);
    if (timer.on) WN_CALL(wn_engine_synchronize(e, &call_err_));
    timer.mark("warmup iterations");
    WN_CALL(wn_engine_freeze(e, &call_err_));  // on_warmup_complete, handlers.hpp:91-101
    printer.in_warmup = false;
    if (stepsize_out != nullptr) WN_CALL(wn_engine_get_step_sizes(e, stepsize_out, &call_err_));
    if (inv_metric_out != nullptr) WN_CALL(wn_engine_get_inv_mass(e, inv_metric_out, &call_err_));
    size_t sampled = 0;
    // controller_loop (sampler.hpp:117-158): R-hat of the log density once every chain has min_iter draws.  The
    // reference's controller looks on a 1 ms timer, not after every draw: here every `rhat_stride` iterations
    // (each look is a handful of small launches and a blocking read-back that would otherwise serialise every
    // transition with the host).
    constexpr int rhat_stride = 5;
    const auto controller_looks_after = [&](int it) {
      return it >= min_sampling_iter && it >= 2 && it < max_sampling_iter && total_chains > 1 &&
             (it - min_sampling_iter) % rhat_stride == 0;
    };
    for (int it = 0; it < max_sampling_iter;) {  // ChainWorker loop, sampler.hpp:82-93
      interrupt.throw_if_interrupted();
      pace.before_enqueue();
      int n = 1;  // up to the controller's next look
      while (n < fuse_limit && it + n < max_sampling_iter && !controller_looks_after(it + n)) ++n;
      if (kept) {
        WN_CALL(wn_engine_sample_steps(e, n, kept->next_row(), kept->stride(), static_cast<int64_t>(D), &call_err_));
        for (int k = 0; k < n; ++k) kept->row_done();
      } else {
        n = static_cast<int>(std::min<size_t>(static_cast<size_t>(n), sink.room()));
        WN_CALL(wn_engine_sample_steps(e, n, sink.next_row(), static_cast<int64_t>(sink.stride()), static_cast<int64_t>(D),
                                       &call_err_));
        sink.rows_done(static_cast<size_t>(n));
      }
      sampled += static_cast<size_t>(n);
      pace.after_enqueue();
      for (int k = 0; k < n; ++k) {  // (a Ctrl-C raised from inside a progress callback ends the call at once)
        printer.progress(total_chains);
        interrupt.throw_if_interrupted();
      }
      it += n;
      if (controller_looks_after(it)) {
        double rhat = 0;
        if (shard != nullptr) {
          rhat = shard->coord->rhat(shard->shard, e);
        } else {
          WN_CALL(wn_engine_rhat(e, &rhat, &call_err_));
        }
        if (printer.print != nullptr && refresh != 0) {
          std::stringstream ss;
          ss << "Controller: R-hat at " << std::setprecision(10) << rhat << std::endl;  // handlers.hpp:160-176
          const std::string msg = ss.str();
          print(msg.c_str(), msg.length(), false);
        }
        if (rhat <= rhat_converge_tol) break;
      }
    }
    WN_CALL(wn_engine_check(e, &call_err_));
    timer.mark("sampling iterations");
    sink.finish();
    if (kept) kept->finish();
    timer.mark("draws in the caller's buffer");
    interrupt.throw_if_interrupted();
    {  // walnutpy.cpp:215-218
      int* lw = shard != nullptr ? shard->lengths_warmup : final_lengths;
      int* ls = shard != nullptr ? shard->lengths_sampling : final_lengths + num_chains;
      for (size_t c = 0; c < num_chains; ++c) {
        lw[c] = static_cast<int>(written_warmup);
        ls[c] = static_cast<int>(sampled);
      }
    }
    if (kept) {
      // the block changes hands: a wn_chains of `sampled` draws per chain that frees it when it is destroyed
      WN_CALL(wn_engine_synchronize(e, &call_err_));
      std::vector<int64_t> lengths(num_chains, static_cast<int64_t>(sampled));
      double* block = kept->release();
      WalnutpyError* adopt_err = nullptr;
      if (wn_chains_adopt(resident->chains_out, block, num_chains, static_cast<size_t>(max_sampling_iter), D,
                          static_cast<int64_t>(max_sampling_iter) * static_cast<int64_t>(D), lengths.data(), cfg.device,
                          &adopt_err) != 0) {
        (void)hipFree(block);
        rethrow(adopt_err);
      }
    }
    timer.mark("lengths written, draw block handed over");
    sink_holder.reset();
    kept.reset();
    wn_engine_destroy(guard.e);
    guard.e = nullptr;
    timer.mark("engine and staging memory released");
    return 0;
  } catch (const ShardAborted&) {
    if (err) *err = nullptr;  // (another shard holds the error this call reports)
  } catch (const InterruptException&) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error("", interrupt));
  } catch (const std::invalid_argument& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), config));
  } catch (const std::exception& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), generic));
  } catch (...) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error("Unknown error", generic));
  }
  return -1;
}

#define WN_SAMPLE_ARGS                                                                                            \
  model, model_params, num_params, inits, num_chains, seed, id, init_radius, init_inv_metric, min_warmup_iter,   \
      max_warmup_iter, min_sampling_iter, max_sampling_iter, max_trajectory_doublings, max_step_halvings,        \
      min_micro_steps, max_hamiltonian_error, step_size_converge_tol, mass_converge_tol, rhat_converge_tol,      \
      mass_init_count, mass_additive_smoothing, max_macro_steps_target, step_size_init, step_accept_rate_target, \
      step_learning_rate, step_gradient_decay, step_sq_gradient_decay, step_stabilization, step_learn_rate_decay, \
      save_warmup, out, out_size, final_lengths, stepsize_out, inv_metric_out, refresh, print, err
#define WN_SAMPLE_PARAMS                                                                                          \
  int model, const double *model_params, int num_params, const double *inits, size_t num_chains,                 \
      unsigned int seed, unsigned int id, double init_radius, const double *init_inv_metric, int min_warmup_iter, \
      int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,           \
      int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,   \
      double mass_converge_tol, double rhat_converge_tol, double mass_init_count,                                \
      double mass_additive_smoothing, double max_macro_steps_target, double step_size_init,                      \
      double step_accept_rate_target, double step_learning_rate, double step_gradient_decay,                     \
      double step_sq_gradient_decay, double step_stabilization, double step_learn_rate_decay, bool save_warmup,  \
      double *out, size_t out_size, int *final_lengths, double *stepsize_out, double *inv_metric_out,            \
      int refresh, PRINT_CALLBACK print, WalnutpyError **err

// (internal, for the tests) the host streams above as a function: `count_per_chain` normals for each of `num_chains`
// chains from mt19937_64(seed_seq{seed, stream}), one distribution for all chains or a fresh one per chain
extern "C" void wn_internal_reference_normals(unsigned int seed, unsigned int stream, size_t num_chains,
                                              size_t count_per_chain, int fresh_per_chain, double scale, double* out) {
  wnref::Workers pool(wnref::usable_threads());
  std::seed_seq ss{seed, stream};
  std::mt19937_64 rng(ss);
  wnref::polar_stream_fill(rng, pool, scale, num_chains, count_per_chain, fresh_per_chain != 0, out);
}

// ---- the reference's host-model entry points (walnutpy.cpp:134-149, 227-245): exported so that the library loads
// where libwalnutpy is expected, and refusing to sample -- a host callback cannot run inside a GPU-resident
// trajectory, and there is no CPU path in this library to fall back to
namespace {
int refuse_host_model(const char* symbol, WalnutpyError** err) {
  const std::string msg = std::string(symbol) +
                          ": this library samples device models only (a host log-density callback cannot be called "
                          "from a GPU-resident trajectory and there is no CPU path here); use walnutpie_sample_device "
                          "with a model compiled through walnuts_amd/csrc/wn_model_api.h, or the reference library for "
                          "host models";
  if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(msg.c_str(), config));
  return -1;
}
}  // namespace
extern "C" int walnutpie_sample_cfunc(WN_LOGP_CFUNC, void*, int, const double*, WN_REFERENCE_SAMPLING_PARAMS) {
  (void)num_chains, (void)seed, (void)id, (void)init_radius, (void)init_inv_metric, (void)min_warmup_iter;
  (void)max_warmup_iter, (void)min_sampling_iter, (void)max_sampling_iter, (void)max_trajectory_doublings;
  (void)max_step_halvings, (void)min_micro_steps, (void)max_hamiltonian_error, (void)step_size_converge_tol;
  (void)mass_converge_tol, (void)rhat_converge_tol, (void)mass_init_count, (void)mass_additive_smoothing;
  (void)max_macro_steps_target, (void)step_size_init, (void)step_accept_rate_target, (void)step_learning_rate;
  (void)step_gradient_decay, (void)step_sq_gradient_decay, (void)step_stabilization, (void)step_learn_rate_decay;
  (void)save_warmup, (void)out, (void)out_size, (void)final_lengths, (void)stepsize_out, (void)inv_metric_out;
  (void)refresh, (void)print;
  return refuse_host_model("walnutpie_sample_cfunc", err);
}
extern "C" int walnutpie_sample_bridgestan(const char*, const char*, STREAM_CALLBACK, unsigned int, const char*,
                                           WN_REFERENCE_SAMPLING_PARAMS) {
  (void)num_chains, (void)seed, (void)id, (void)init_radius, (void)init_inv_metric, (void)min_warmup_iter;
  (void)max_warmup_iter, (void)min_sampling_iter, (void)max_sampling_iter, (void)max_trajectory_doublings;
  (void)max_step_halvings, (void)min_micro_steps, (void)max_hamiltonian_error, (void)step_size_converge_tol;
  (void)mass_converge_tol, (void)rhat_converge_tol, (void)mass_init_count, (void)mass_additive_smoothing;
  (void)max_macro_steps_target, (void)step_size_init, (void)step_accept_rate_target, (void)step_learning_rate;
  (void)step_gradient_decay, (void)step_sq_gradient_decay, (void)step_stabilization, (void)step_learn_rate_decay;
  (void)save_warmup, (void)out, (void)out_size, (void)final_lengths, (void)stepsize_out, (void)inv_metric_out;
  (void)refresh, (void)print;
  return refuse_host_model("walnutpie_sample_bridgestan", err);
}
extern "C" char walnutpie_separator_char(void) { return '\x1C'; }  // walnutpy.cpp:224-225 (ASCII file separator)

extern "C" int walnutpie_sample_device(WN_SAMPLE_PARAMS) {
  return sample_device_impl(false, nullptr, nullptr, WN_SAMPLE_ARGS);
}
extern "C" int walnutpie_sample_device_reference_streams(WN_SAMPLE_PARAMS) {
  return sample_device_impl(true, nullptr, nullptr, WN_SAMPLE_ARGS);
}
#undef WN_SAMPLE_PARAMS
#define WN_SAMPLE_PARAMS_NOERR                                                                                    \
  int model, const double *model_params, int num_params, const double *inits, size_t num_chains,                 \
      unsigned int seed, unsigned int id, double init_radius, const double *init_inv_metric, int min_warmup_iter, \
      int max_warmup_iter, int min_sampling_iter, int max_sampling_iter, int max_trajectory_doublings,           \
      int max_step_halvings, int min_micro_steps, double max_hamiltonian_error, double step_size_converge_tol,   \
      double mass_converge_tol, double rhat_converge_tol, double mass_init_count,                                \
      double mass_additive_smoothing, double max_macro_steps_target, double step_size_init,                      \
      double step_accept_rate_target, double step_learning_rate, double step_gradient_decay,                     \
      double step_sq_gradient_decay, double step_stabilization, double step_learn_rate_decay, bool save_warmup,  \
      double *out, size_t out_size, int *final_lengths, double *stepsize_out, double *inv_metric_out,            \
      int refresh, PRINT_CALLBACK print
extern "C" int walnutpie_sample_device_resident(WN_SAMPLE_PARAMS_NOERR, int thin, wn_chains** chains_out,
                                                WalnutpyError** err) {
  const ResidentRequest req{thin, chains_out};
  return sample_device_impl(false, &req, nullptr, WN_SAMPLE_ARGS);
}

// walnutpie_sample_device over several devices of the node: one host thread, engine and stream per entry of `devices`
// (an ordinal may repeat: two shards on one device overlap each other's launch tails), contiguous shards of the global
// chain ids, every shard writing its own slice of the caller's buffers.  Results do not depend on the sharding: the
// random streams are keyed by global chain id, the controllers look at all chains (Coordinator).
// `resident` (walnutpie_sample_device_multi_resident): every shard keeps its sampling draws in a block on ITS device;
// when all shards are done the blocks -- contiguous slabs of the chain-major [C][S][D] layout -- are gathered into one
// block on devices[0], one peer-to-peer copy per shard (hipMemcpyPeerAsync: over the shard's own xGMI link, all
// inbound copies at once), and handed over as one wn_chains for the wn_summary_* functions.
static int sample_multi_impl(const ResidentRequest* resident, WN_SAMPLE_PARAMS_NOERR, const int* devices,
                             int num_devices, WalnutpyError** err) {
  std::vector<wn_chains*> shard_chains(static_cast<size_t>(std::max(num_devices, 0)), nullptr);
  struct ChainsGuard {
    std::vector<wn_chains*>& v;
    ~ChainsGuard() {
      for (auto* c : v)
        if (c != nullptr) wn_chains_destroy(c);
    }
  } chains_guard{shard_chains};
  try {
    if (devices == nullptr || num_devices < 1) throw std::invalid_argument("devices must name at least one device");
    if (resident != nullptr) {
      if (resident->chains_out == nullptr) throw std::invalid_argument("chains_out must not be null");
      if (resident->thin < 0) throw std::invalid_argument("thin must be >= 0");
      if (max_sampling_iter < 1) throw std::invalid_argument("resident draws need max_sampling_iter >= 1");
      for (int d = 0; d < (resident->all_gather ? std::max(num_devices, 1) : 1); ++d) resident->chains_out[d] = nullptr;
    }
    if (num_chains < static_cast<size_t>(num_devices)) throw std::invalid_argument("fewer chains than devices");
    if (num_params < 1) throw std::invalid_argument("num_params must be in {1, 2, ... }");
    if (max_sampling_iter < 0 || max_warmup_iter < 0) throw std::invalid_argument("iteration counts must be >= 0");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) throw std::runtime_error("cannot count the devices");
    for (int s = 0; s < num_devices; ++s)
      if (devices[s] < 0 || devices[s] >= visible) throw std::invalid_argument("device ordinal out of range");
    const size_t D = static_cast<size_t>(num_params);
    // (the caller's rows per chain: every sampling draw, or -- resident -- only every thin-th of them)
    const size_t thin = resident != nullptr ? static_cast<size_t>(resident->thin) : 1;
    const size_t samp_rows = thin == 0 ? 0 : (static_cast<size_t>(max_sampling_iter) + thin - 1) / thin;
    const size_t rows = samp_rows + (save_warmup ? static_cast<size_t>(max_warmup_iter) : 0);
    if (rows > 0 && out == nullptr) throw std::invalid_argument("out must not be null");
    if (out_size < num_chains * rows * D) {  // walnutpy.cpp:153-160
      std::stringstream ss;
      ss << "Output buffer too small. Expected at least " << num_chains << " chains of " << rows * D << " doubles, got "
         << out_size;
      throw std::runtime_error(ss.str());
    }
    InterruptGuard guard;
    Coordinator coord(num_devices, D);
    std::vector<WalnutpyError*> errors(static_cast<size_t>(num_devices), nullptr);
    std::vector<int> rcs(static_cast<size_t>(num_devices), 0);
    std::vector<std::thread> threads;
    const size_t base = num_chains / static_cast<size_t>(num_devices), extra = num_chains % static_cast<size_t>(num_devices);
    size_t begin = 0;
    for (int s = 0; s < num_devices; ++s) {
      const size_t count = base + (static_cast<size_t>(s) < extra ? 1 : 0);
      ShardCtx ctx;
      ctx.shard = s;
      ctx.device = devices[s];
      ctx.chain_begin = begin;
      ctx.total_chains = num_chains;
      ctx.coord = &coord;
      ctx.interrupt = &guard;
      ctx.lengths_warmup = final_lengths + begin;
      ctx.lengths_sampling = final_lengths + num_chains + begin;
      threads.emplace_back([&, ctx, count, s] {
        const double* inits_s = inits == nullptr ? nullptr : inits + ctx.chain_begin * D;
        const double* metric_s = init_inv_metric == nullptr ? nullptr : init_inv_metric + ctx.chain_begin * D;
        double* out_s = out == nullptr ? nullptr : out + ctx.chain_begin * rows * D;
        double* step_s = stepsize_out == nullptr ? nullptr : stepsize_out + ctx.chain_begin;
        double* metric_out_s = inv_metric_out == nullptr ? nullptr : inv_metric_out + ctx.chain_begin * D;
        const ResidentRequest shard_req{resident != nullptr ? resident->thin : 0, &shard_chains[static_cast<size_t>(s)]};
        rcs[s] = sample_device_impl(
            false, resident != nullptr ? &shard_req : nullptr, &ctx, model, model_params, num_params, inits_s, count, seed, id, init_radius, metric_s,
            min_warmup_iter, max_warmup_iter, min_sampling_iter, max_sampling_iter, max_trajectory_doublings,
            max_step_halvings, min_micro_steps, max_hamiltonian_error, step_size_converge_tol, mass_converge_tol,
            rhat_converge_tol, mass_init_count, mass_additive_smoothing, max_macro_steps_target, step_size_init,
            step_accept_rate_target, step_learning_rate, step_gradient_decay, step_sq_gradient_decay,
            step_stabilization, step_learn_rate_decay, save_warmup, out_s, count * rows * D, nullptr, step_s,
            metric_out_s, refresh, print, &errors[s]);
        if (rcs[s] != 0) coord.abandon();
      });
      begin += count;
    }
    for (auto& t : threads) t.join();
    // the first shard that failed with an error of its own speaks for the call
    int rc = 0;
    for (int s = 0; s < num_devices; ++s) {
      if (rcs[s] != 0 && errors[s] != nullptr && rc == 0) {
        rc = -1;
        if (err) *err = errors[s];
        else walnutpie_destroy_error(errors[s]);
        errors[s] = nullptr;
      }
    }
    for (auto* e : errors)
      if (e != nullptr) walnutpie_destroy_error(e);
    if (rc == 0)
      for (int s = 0; s < num_devices; ++s)
        if (rcs[s] != 0) throw std::runtime_error("a shard ended without reporting its error");
    if (rc == 0 && resident != nullptr) {
      // gather: shard s's [count_s][S][D] block is rows [begin_s, begin_s + count_s) of the whole [C][S][D] block.
      // One destination: devices[0] (gather), or every listed device (all_gather: the north star's exchange -- every
      // device ends with every shard's draws).  Every (destination, source) pair is its own hipMemcpyPeerAsync on its
      // own stream of the destination, so the inbound copies of a device run side by side, each over the xGMI link of
      // its source -- not one after the other on one in-order stream.  Peer access is switched on per pair first: without
      // it the runtime stages a peer copy through host memory.  (Unmeasured on more than one physical device: this pool
      // has one GPU per box; the one-device tests list a device several times, where a "peer" copy is a local copy.)
      const size_t S = static_cast<size_t>(max_sampling_iter);
      struct RestoreDevice {  // the gather selects the destinations: the calling thread gets its current device back
        int before = -1;
        RestoreDevice() { if (hipGetDevice(&before) != hipSuccess) before = -1; }
        ~RestoreDevice() { if (before >= 0) (void)hipSetDevice(before); }
      } restore_device;
      const int destinations = resident->all_gather ? num_devices : 1;
      std::vector<size_t> first(static_cast<size_t>(num_devices) + 1, 0);
      for (int s = 0; s < num_devices; ++s)
        first[static_cast<size_t>(s) + 1] = first[static_cast<size_t>(s)] + wn_chains_num_chains(shard_chains[static_cast<size_t>(s)]);
      std::vector<DevBlock> whole(static_cast<size_t>(destinations));
      std::vector<std::unique_ptr<Stream>> copies;
      for (int d = 0; d < destinations; ++d) {
        const int dst = devices[d];
        if (hipSetDevice(dst) != hipSuccess) throw std::runtime_error("cannot select the device");
        if (!whole[static_cast<size_t>(d)].alloc(num_chains * S * D)) throw std::runtime_error("cannot allocate the gathered draw block");
        for (int s = 0; s < num_devices; ++s) {
          const int src = devices[s];
          if (src != dst) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dst, src) != hipSuccess || can == 0) {
              std::stringstream ss;
              ss << "device " << dst << " has no peer-to-peer path to device " << src
                 << ": the shards' draws cannot be gathered on it (list devices of one xGMI hive)";
              throw std::runtime_error(ss.str());
            }
            const hipError_t en = hipDeviceEnablePeerAccess(src, 0);
            if (en != hipSuccess && en != hipErrorPeerAccessAlreadyEnabled) throw std::runtime_error("cannot enable peer access");
            (void)hipGetLastError();  // (an "already enabled" is not an error to carry along)
          }
          copies.push_back(std::make_unique<Stream>());
          copies.back()->create();
          const size_t count = first[static_cast<size_t>(s) + 1] - first[static_cast<size_t>(s)];
          if (hipMemcpyPeerAsync(whole[static_cast<size_t>(d)].p + first[static_cast<size_t>(s)] * S * D, dst,
                                 wn_chains_device_draws(shard_chains[static_cast<size_t>(s)]), src,
                                 count * S * D * sizeof(double), copies.back()->s) != hipSuccess)
            throw std::runtime_error("gathering the shards' draws failed");
        }
      }
      for (auto& c : copies)
        if (hipStreamSynchronize(c->s) != hipSuccess) throw std::runtime_error("gathering the shards' draws failed");
      std::vector<int64_t> lengths(num_chains);
      for (size_t c = 0; c < num_chains; ++c) lengths[c] = final_lengths[num_chains + c];
      for (int d = 0; d < destinations; ++d) resident->chains_out[d] = nullptr;
      for (int d = 0; d < destinations; ++d) {
        double* block = whole[static_cast<size_t>(d)].release();
        WalnutpyError* adopt_err = nullptr;
        if (wn_chains_adopt(&resident->chains_out[d], block, num_chains, S, D, static_cast<int64_t>(S * D), lengths.data(),
                            devices[d], &adopt_err) != 0) {
          (void)hipFree(block);
          for (int k = 0; k < d; ++k) {  // (nothing half-built is handed back)
            wn_chains_destroy(resident->chains_out[k]);
            resident->chains_out[k] = nullptr;
          }
          rethrow(adopt_err);
        }
      }
    }
    return rc;
  } catch (const std::invalid_argument& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), config));
  } catch (const std::exception& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), generic));
  } catch (...) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error("Unknown error", generic));
  }
  return -1;
}
#define WN_SAMPLE_ARGS_NOERR                                                                                       \
  model, model_params, num_params, inits, num_chains, seed, id, init_radius, init_inv_metric, min_warmup_iter,     \
      max_warmup_iter, min_sampling_iter, max_sampling_iter, max_trajectory_doublings, max_step_halvings,          \
      min_micro_steps, max_hamiltonian_error, step_size_converge_tol, mass_converge_tol, rhat_converge_tol,        \
      mass_init_count, mass_additive_smoothing, max_macro_steps_target, step_size_init, step_accept_rate_target,  \
      step_learning_rate, step_gradient_decay, step_sq_gradient_decay, step_stabilization, step_learn_rate_decay, \
      save_warmup, out, out_size, final_lengths, stepsize_out, inv_metric_out, refresh, print
extern "C" int walnutpie_sample_device_multi(WN_SAMPLE_PARAMS_NOERR, const int* devices, int num_devices,
                                             WalnutpyError** err) {
  return sample_multi_impl(nullptr, WN_SAMPLE_ARGS_NOERR, devices, num_devices, err);
}
extern "C" int walnutpie_sample_device_multi_resident(WN_SAMPLE_PARAMS_NOERR, const int* devices, int num_devices, int thin,
                                                      wn_chains** chains_out, WalnutpyError** err) {
  const ResidentRequest req{thin, chains_out, false};
  return sample_multi_impl(&req, WN_SAMPLE_ARGS_NOERR, devices, num_devices, err);
}
extern "C" int walnutpie_sample_device_multi_allgather(WN_SAMPLE_PARAMS_NOERR, const int* devices, int num_devices, int thin,
                                                       wn_chains** chains_out, WalnutpyError** err) {
  const ResidentRequest req{thin, chains_out, true};
  return sample_multi_impl(&req, WN_SAMPLE_ARGS_NOERR, devices, num_devices, err);
}

// ---- walnutpie_ess / walnutpie_r_hat / walnutpie_mcse (walnutpy.cpp:333-369) ----------------------------------
// The reference's ctypes layer binds these three with (draws, num_draws, num_params, lengths, num_chains, out, err);
// `draws` is what Eigen::Map<const MatrixXd>(draws, num_draws, num_params) reads (walnutpy.cpp:89): a COLUMN-major
// num_draws x num_params matrix of the stacked chains.  Same symbols and arguments here: the draws are uploaded in
// the device layout, summarised by wn_summary_* (wn_summary.hip) and the num_params results copied back.
namespace {
template <class F>
int summary_shim(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains, double* out,
                 WalnutpyError** err, F summarise) {
  try {
    if (draws == nullptr || lengths == nullptr || out == nullptr) throw std::invalid_argument("null argument");
    if (num_draws < 0 || num_params < 1 || num_chains < 1) throw std::invalid_argument("sizes must be positive");
    std::vector<int64_t> sizes(static_cast<size_t>(num_chains));
    int64_t total = 0;
    for (int m = 0; m < num_chains; ++m) {
      sizes[m] = lengths[m];
      total += lengths[m];
    }
    if (total != num_draws)  // MarkovChainsUnified, summary.hpp:266-280
      throw std::invalid_argument("The number of rows in draws and sum of chain_sizes must be equal.");
    const size_t N = static_cast<size_t>(num_draws), D = static_cast<size_t>(num_params);
    std::vector<double> rows(N * D);  // [draw][param]
    for (size_t d = 0; d < D; ++d)
      for (size_t n = 0; n < N; ++n) rows[n * D + d] = draws[d * N + n];
    wn_chains* ch = nullptr;
    WN_CALL(wn_chains_upload(&ch, rows.data(), D, sizes.data(), static_cast<size_t>(num_chains), 0, &call_err_));
    struct Guard {
      wn_chains* c;
      ~Guard() { wn_chains_destroy(c); }
    } guard{ch};
    WN_CALL(summarise(ch, out, &call_err_));
    return 0;
  } catch (const std::invalid_argument& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), config));
  } catch (const std::exception& ex) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error(ex.what(), generic));
  } catch (...) {
    if (err) *err = static_cast<WalnutpyError*>(wn_internal_make_error("Unknown error", generic));
  }
  return -1;
}
}  // namespace

extern "C" int walnutpie_ess(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains,
                             double* out, WalnutpyError** err) {
  return summary_shim(draws, num_draws, num_params, lengths, num_chains, out, err, wn_summary_effective_sample_size);
}
extern "C" int walnutpie_r_hat(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains,
                               double* out, WalnutpyError** err) {
  return summary_shim(draws, num_draws, num_params, lengths, num_chains, out, err, wn_summary_r_hat);
}
extern "C" int walnutpie_mcse(const double* draws, int num_draws, int num_params, const int* lengths, int num_chains,
                              double* out, WalnutpyError** err) {
  return summary_shim(draws, num_draws, num_params, lengths, num_chains, out, err,
                      wn_summary_monte_carlo_standard_error);
}

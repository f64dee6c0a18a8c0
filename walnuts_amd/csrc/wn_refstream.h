// wn_refstream.h -- the reference's host-side initial streams, bit for bit, at many-chain scale.
//
// walnutpie_sample_cfunc draws every chain's initial position from ONE mt19937_64(seed_seq{seed, 1}) through ONE
// detail::Random (walnutpy.cpp:187-189, config.hpp:259-268) and feeds the step-size search from ONE
// mt19937_64(seed_seq{seed, 2}) with a fresh normal distribution per chain (walnutpy.cpp:75-80, util.hpp:288).  The
// streams are sequential by construction -- libstdc++'s normal_distribution is Marsaglia's polar method, it REJECTS,
// so where chain c's numbers start depends on every rejection before it -- and at 65 536 chains x 1 024 parameters a
// plain loop over std::normal_distribution is 2.6 s per stream: 5.3 of the 5.7 s of a whole 52-iteration call
// (profiles/r03/sample_device_e2e.txt).
//
// Same numbers, with ONLY the engine sequential.  Every attempt of the polar method consumes exactly two engine outputs
// whether it is accepted or not, so attempt k reads outputs 2k and 2k+1 whatever happened before it.  The calling
// thread does nothing but run the engine into blocks of raw 64-bit outputs (~1.5 ns each).  Workers take a block each:
// phase A evaluates the rejection test of its attempts and counts the accepted ones; the caller, scanning the blocks in
// order, turns the counts into the index of each block's first accepted attempt in the whole stream; phase B evaluates
// sqrt(-2 log(r2) / r2) and the two products of every accepted attempt and writes them where the j-th accepted attempt of
// the stream belongs -- a closed form of j, for one distribution shared by all chains as for a fresh one per chain.
// (Round 3's first version ran the rejection test on the calling thread too: ~10 ns per normal, 0.8 s per stream at the
// headline size; this one ~2.5 ns.)
//
// libstdc++ (bits/random.tcc, normal_distribution::operator()):
//     do { x = 2 * canonical() - 1; y = 2 * canonical() - 1; r2 = x * x + y * y; } while (r2 > 1 || r2 == 0);
//     mult = sqrt(-2 * log(r2) / r2);  saved = x * mult;  return y * mult;      // next call returns `saved`
// with every result passed through `ret * stddev + mean` (= ret * 1.0 + 0.0), and, for a 64-bit engine,
// generate_canonical<double, 53>() = double(engine()) / 2^64, replaced by nextafter(1, 0) when that rounds to 1.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include <sched.h>

namespace wnref {

inline double canonical(uint64_t engine_output) {
  const double u = static_cast<double>(engine_output) * 0x1p-64;  // (an exact scaling: what sum / 2^64 computes)
  return u >= 1.0 ? std::nextafter(1.0, 0.0) : u;
}

// cores this process may use (affinity mask; a container on a big host is often limited to a few)
inline unsigned usable_threads() {
  cpu_set_t set;
  unsigned n = 0;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = static_cast<unsigned>(CPU_COUNT(&set));
  if (n == 0) n = std::thread::hardware_concurrency();
  return std::max(1u, std::min(n, 32u));
}

// a handful of worker threads for pass 2, alive for the duration of one sampling call
class Workers {
 public:
  explicit Workers(unsigned n) {
    for (unsigned i = 0; i < n; ++i) threads_.emplace_back([this] { loop(); });
  }
  ~Workers() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  void submit(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      jobs_.push_back(std::move(job));
      ++pending_;
    }
    cv_.notify_one();
  }
  void wait_idle() {
    std::unique_lock<std::mutex> lk(mu_);
    idle_.wait(lk, [this] { return pending_ == 0; });
  }

 private:
  void loop() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
        if (jobs_.empty()) return;
        job = std::move(jobs_.front());
        jobs_.pop_front();
      }
      job();
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) idle_.notify_all();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::deque<std::function<void()>> jobs_;
  std::mutex mu_;
  std::condition_variable cv_, idle_;
  size_t pending_ = 0;
  bool stop_ = false;
};

// `chains` x `count` values of the reference's stream over `eng`, times `scale` (the caller's `x *= scale`, exact for
// scale 1), to out[chains][count].  fresh_per_chain = false: ONE std::normal_distribution<double> serves all chains (a
// saved second variate crosses chain boundaries, config.hpp:259-268); true: a new distribution per chain (util.hpp:288: a
// saved variate is dropped at the boundary).  The engine is run AHEAD in blocks: it is left in a later state than the
// plain loops would leave it (the callers' engines live for one stream).  Returns when every value is written.
inline void polar_stream_fill(std::mt19937_64& eng, Workers& pool, double scale, size_t chains, size_t count,
                              bool fresh_per_chain, double* out) {
  const size_t total = chains * count;
  if (total == 0) return;
  const size_t per_chain = (count + 1) / 2;  // accepted attempts a chain with its own distribution consumes
  const size_t needed = fresh_per_chain ? chains * per_chain : (total + 1) / 2;
  constexpr size_t kAttempts = size_t{1} << 15;  // per block: 64 Ki engine outputs, 512 KB
  struct Block {
    std::vector<uint64_t> raw;
    std::vector<uint32_t> accepted;  // attempts of this block that pass the rejection test, in order
    std::atomic<bool> tested{false};
  };
  const auto xy = [](const Block& b, uint32_t k, double& x, double& y, double& r2) {
    x = 2.0 * canonical(b.raw[2 * k]) - 1.0;
    y = 2.0 * canonical(b.raw[2 * k + 1]) - 1.0;
    r2 = x * x + y * y;
  };
  const size_t depth = 2 * static_cast<size_t>(usable_threads()) + 2;  // blocks between the engine and the scan
  std::deque<std::shared_ptr<Block>> in_flight;
  size_t first = 0;  // accepted attempts of the stream before the block at the front
  while (first < needed) {
    while (in_flight.size() < depth) {
      auto b = std::make_shared<Block>();
      b->raw.resize(2 * kAttempts);
      for (auto& r : b->raw) r = eng();  // the sequential part, and nothing else
      pool.submit([b, xy] {  // phase A
        b->accepted.reserve(kAttempts);
        for (uint32_t k = 0; k < kAttempts; ++k) {
          double x, y, r2;
          xy(*b, k, x, y, r2);
          if (!(r2 > 1.0 || r2 == 0.0)) b->accepted.push_back(k);
        }
        b->tested.store(true, std::memory_order_release);
      });
      in_flight.push_back(std::move(b));
    }
    std::shared_ptr<Block> b = in_flight.front();
    in_flight.pop_front();
    while (!b->tested.load(std::memory_order_acquire)) std::this_thread::yield();
    const size_t take = std::min(b->accepted.size(), needed - first);
    const size_t base = first;
    pool.submit([=] {  // phase B: accepted attempt j = base + t of the stream
      for (size_t t = 0; t < take; ++t) {
        double x, y, r2;
        xy(*b, b->accepted[t], x, y, r2);
        const double mult = std::sqrt(-2 * std::log(r2) / r2);
        const size_t j = base + t;
        double* o;
        bool second;
        if (fresh_per_chain) {
          const size_t c = j / per_chain, k = j % per_chain;
          o = out + c * count + 2 * k;
          second = 2 * k + 1 < count;
        } else {
          o = out + 2 * j;
          second = 2 * j + 1 < total;
        }
        o[0] = ((y * mult) * 1.0 + 0.0) * scale;
        if (second) o[1] = ((x * mult) * 1.0 + 0.0) * scale;
      }
    });
    first += b->accepted.size();
  }
  pool.wait_idle();  // (blocks tested beyond the end of the stream are dropped with their last reference)
}

}  // namespace wnref

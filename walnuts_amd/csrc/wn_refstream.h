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
// Same numbers, split in two passes.  Pass 1 is the sequential part and nothing else: the engine, the two canonical
// uniforms per attempt, the rejection test -- it records (x, y, r2) of every ACCEPTED attempt (~5 ns per normal).
// Pass 2, sqrt(-2 log(r2) / r2) and the two products, is a pure function of one record and runs on every core the
// process may use, chunk by chunk, while pass 1 is already producing the next chunk.
//
// libstdc++ (bits/random.tcc, normal_distribution::operator()):
//     do { x = 2 * canonical() - 1; y = 2 * canonical() - 1; r2 = x * x + y * y; } while (r2 > 1 || r2 == 0);
//     mult = sqrt(-2 * log(r2) / r2);  saved = x * mult;  return y * mult;      // next call returns `saved`
// with every result passed through `ret * stddev + mean` (= ret * 1.0 + 0.0), and, for a 64-bit engine,
// generate_canonical<double, 53>() = double(engine()) / 2^64, replaced by nextafter(1, 0) when that rounds to 1.
#pragma once

#include <algorithm>
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

inline double canonical(std::mt19937_64& eng) {
  const double u = static_cast<double>(eng()) * 0x1p-64;  // (an exact scaling: what sum / 2^64 computes)
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

// One accepted attempt of the polar method and where its (up to) two normals go.
struct Accepted {
  double x, y, r2;
  double* first;   // receives y * mult (the value the distribution returns first); never null
  double* second;  // receives x * mult (the saved value), or null when the distribution is discarded before using it
};

inline void finish_records(const Accepted* rec, size_t n, double scale) {
  for (size_t i = 0; i < n; ++i) {
    const double mult = std::sqrt(-2 * std::log(rec[i].r2) / rec[i].r2);
    *rec[i].first = ((rec[i].y * mult) * 1.0 + 0.0) * scale;
    if (rec[i].second != nullptr) *rec[i].second = ((rec[i].x * mult) * 1.0 + 0.0) * scale;
  }
}

// `count` standard normals of ONE std::normal_distribution<double> over `eng`, written to out[0 .. count) times
// `scale` (the caller's `x *= scale`, exact for scale 1).  `carry`: a saved second variate of an earlier call on the
// same distribution object goes to out[0] first (its record was submitted by that call).  Returns, through `carry`,
// whether this call leaves a saved variate for the next one, and in that case where it must be delivered.
// Records are handed to `pool` in chunks; the caller waits for the pool before reading `out`.
class PolarStream {
 public:
  PolarStream(std::mt19937_64& eng, Workers& pool, double scale) : eng_(eng), pool_(pool), scale_(scale) { fresh_chunk(); }
  ~PolarStream() { flush(); }
  // the next `count` values of the distribution go to out[0 .. count)
  void fill(double* out, size_t count) {
    size_t i = 0;
    if (saved_ != nullptr && count > 0) {  // the distribution holds a saved variate: it is the next value
      saved_->second = out;
      saved_ = nullptr;
      i = 1;
    }
    for (; i < count; i += 2) {
      double x, y, r2;
      do {
        x = 2.0 * canonical(eng_) - 1.0;
        y = 2.0 * canonical(eng_) - 1.0;
        r2 = x * x + y * y;
      } while (r2 > 1.0 || r2 == 0.0);
      chunk_->push_back(Accepted{x, y, r2, out + i, i + 1 < count ? out + i + 1 : nullptr});
      if (i + 1 >= count) saved_ = &chunk_->back();  // one value left over in the distribution
      else if (chunk_->size() >= kChunk) flush();
    }
  }
  // a new distribution object over the same engine (util.hpp:288): a saved variate is dropped
  void reset_distribution() { saved_ = nullptr; }
  void flush() {
    if (chunk_->empty()) return;
    if (saved_ != nullptr) return;  // its destination is not known yet: the chunk leaves with the next flush
    std::shared_ptr<std::vector<Accepted>> c = chunk_;
    const double scale = scale_;
    pool_.submit([c, scale] { finish_records(c->data(), c->size(), scale); });
    fresh_chunk();
  }
  // the stream ends here: a saved variate that nobody will ask for is dropped, everything recorded is handed over
  void finish() {
    saved_ = nullptr;
    flush();
  }

 private:
  static constexpr size_t kChunk = size_t{1} << 16;
  void fresh_chunk() {
    chunk_ = std::make_shared<std::vector<Accepted>>();
    chunk_->reserve(kChunk + 1);
  }
  std::mt19937_64& eng_;
  Workers& pool_;
  double scale_;
  std::shared_ptr<std::vector<Accepted>> chunk_;
  Accepted* saved_ = nullptr;  // record whose x * mult is the distribution's saved variate (lives in *chunk_)
};

}  // namespace wnref

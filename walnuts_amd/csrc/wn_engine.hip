// wn_engine.hip -- host side of the C ABI in include/walnuts_hip.h: owns the chain-major
// HBM planes, picks the launch geometry and drives the persistent transition kernel.
#include "wn_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/walnuts_hip.h"
#include "wn_elementwise.h"
#include "wn_init.h"
#include "wn_launch.h"
#include "wn_traj.h"

#include "wn_host.h"

// Host-side reproduction of the reference's per-chain random streams (api.hpp:46-51 + detail::Random,
// util.hpp:78-162): engine m = mt19937_64(seed_seq{seed, m+1}); per transition D normals (libstdc++'s polar
// method with its cached second variate), then one engine output per bernoulli / uniform.  The variates are
// generated here and fed to the kernel (kRngBuffer); after the launch each engine is advanced by the number of
// scalar draws its chain actually consumed.  Parity mode for small runs: one host round trip per transition.
struct ReferenceStreams {
  std::vector<std::mt19937_64> eng;
  std::vector<std::normal_distribution<double>> normal;
  std::vector<std::mt19937_64> after_normals;
  std::vector<double> z, u;
  std::vector<int32_t> used;
  int pool = 0;
};

// (internal, for the tests) wnd::sqrt_normal on the device for arguments handed in from the host
static __global__ void sqrt_probe_kernel(const double* x, double* y, long long n, int checked) {
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x)
    y[i] = checked ? wnd::sqrt_normal<true>(x[i]) : wnd::sqrt_normal<false>(x[i]);
}

struct wn_engine {
  int model = 0, D = 0, Dp = 0;
  size_t C = 0;
  wn_config cfg{};
  wn::Geometry geo{};
  int device = 0;
  int num_cus = 256;
  int grid = 0;
  int pool_lds = 0, pool_total = 0;
  bool im_in_lds = false;  // streaming kernels: the chain's inverse mass parked in LDS (wn_traj.h: TrajMem::im_lds)
  bool no_far_end_sums = false;  // experiment switch (WALNUTS_AMD_NO_FAR_END_SUMS=1)
  bool hold_moving_end = false;  // streaming kernels: the moving end's (theta, rho) stay in registers (TrajMem, HOLD)
  int64_t arena_stride = 0;  // doubles per persistent workgroup: HBM part of the span pool (+ streaming scratch)
  size_t smem = 0;
  hipStream_t stream = nullptr;

  DevBuf<double> theta, mass, inv_mass, chol_mass, draw_mean, draw_ssd, score_mean, score_ssd;
  DevBuf<double> step_init, step_size, adam, est_weight, mm_state, logp, model_params, arena, z_buf, u_buf;
  DevBuf<double> lp_stats, mon_partial, mon_out, mon_colsum, mon_rel_mass, mon_rel_step;
  DevBuf<int32_t> min_micro, depth, rng_draws, failed_ext;
  DevBuf<int64_t> grad_evals;
  DevBuf<uint32_t> counter, error_flags;
  DevBuf<unsigned long long> scratch64;

  uint64_t seed = 0;
  uint32_t chain_offset = 0;
  uint32_t transition = 0;
  int64_t warmup_iter = 0;
  int64_t iteration = 0;
  bool adapters_ready = false;
  bool frozen = false;
  bool variates_pending = false;
  int u_stride = 0;
  std::unique_ptr<ReferenceStreams> ref_streams;

  // HIP event pairs around the transition launches: a fixed ring (the last kEventRing launches since the last
  // timing reset can be read back), created once
  static constexpr size_t kEventRing = 1024;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;  // launches since the last timing reset
  bool timing = false;     // record events around the launches (wn_engine_timing_reset switches it on)
  hipEvent_t region_begin = nullptr, region_end = nullptr;  // wn_engine_region_begin / _region_ms
  size_t region_launches = 0;
  bool own_stream = true;
  // Chain groups (round 4): with few work items per resident workgroup (config #2: 4, config #3: 5) a launch's tail --
  // the last chains finishing while the chip drains -- is 25-45 % of it (profiles/r03/item_balance.txt).  The chains are
  // then split into `groups` contiguous blocks, each with its own stream, chain counter and arena slice, launched
  // independently: nothing orders group 1's launch n + 1 behind group 0's launch n, so one group's tail is filled by the
  // other's workgroups (two engines on two streams measured +12 % / +23 % on configs #2 / #3 and +2 % on the headline,
  // profiles/r03/two_groups.txt; in the engine: +13 % / +26 % / +2 %, profiles/r04/ab_chain_groups.txt -- as long as
  // nothing re-aligns the groups: a join of the streams at every step gives the lock-step numbers back).  Everything
  // else the engine does runs on `stream` and first waits for the groups (join_groups(), reached through use_device()).
  static constexpr int kMaxGroups = 4;
  int groups = 1;
  size_t group_begin[kMaxGroups + 1] = {};
  int group_grid[kMaxGroups] = {};
  hipStream_t gstream[kMaxGroups] = {};  // [0] is `stream`
  hipEvent_t gdone[kMaxGroups] = {}, main_point = nullptr;
  hipEvent_t ext_point = nullptr, rel_point = nullptr;  // wn_engine_wait_stream / _release_stream
  uint32_t work_base[kMaxGroups] = {};  // value of each group's device-side chain counter at its next launch
  // register kernels, warmup: the mass estimator's observation of a launch's last transition is applied by the next
  // launch's first prologue (wn_chip.h kDeferObservation).  Until then it is PENDING: the planes and weights hold the
  // state before it, the position plane what it will observe.  Everything but a warmup launch applies it first
  // (flush_pending_observation(), reached through use_device()), so nothing outside the kernels ever sees the difference.
  bool est_pending = false;
  bool in_flush = false;
  bool groups_ahead = false;  // a group stream holds launches `stream` has not waited for
  bool main_moved = true;     // `stream` has done something since the groups last waited for it
  bool in_step = false;

  ~wn_engine() {
    for (auto& ev : events) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    if (region_begin) (void)hipEventDestroy(region_begin);
    if (region_end) (void)hipEventDestroy(region_end);
    for (int g = 1; g < kMaxGroups; ++g) {
      if (gstream[g]) (void)hipStreamDestroy(gstream[g]);
      if (gdone[g]) (void)hipEventDestroy(gdone[g]);
    }
    if (main_point) (void)hipEventDestroy(main_point);
    if (ext_point) (void)hipEventDestroy(ext_point);
    if (rel_point) (void)hipEventDestroy(rel_point);
    if (stream && own_stream) (void)hipStreamDestroy(stream);
  }
  // `stream` waits for what the group streams hold
  void join_groups() {
    for (int g = 1; g < groups; ++g) HIP_OK(hipStreamWaitEvent(stream, gdone[g], 0));
    groups_ahead = false;
  }

  std::pair<hipEvent_t, hipEvent_t>& next_events() {
    const size_t slot = events_used++ % kEventRing;
    if (slot == events.size()) {
      hipEvent_t a, b;
      HIP_OK(hipEventCreate(&a));
      HIP_OK(hipEventCreate(&b));
      events.emplace_back(a, b);
    }
    return events[slot];
  }

  void use_device() {
    HIP_OK(hipSetDevice(device));
    if (est_pending && !in_step && !in_flush) flush_pending_observation();
    if (groups > 1 && !in_step) {  // anything but a transition launch: ordered after every group, and the groups after it
      if (groups_ahead) join_groups();
      main_moved = true;
    }
  }
  void flush_pending_observation();

  void upload_rows(DevBuf<double>& dst, const double* host, double pad_value) {
    // host [C][D] -> device [C][Dp]; padding columns keep their fill value
    use_device();
    if (Dp != D) {
      std::vector<double> padded(C * static_cast<size_t>(Dp), pad_value);
      for (size_t c = 0; c < C; ++c) std::memcpy(&padded[c * Dp], host + c * D, sizeof(double) * D);
      HIP_OK(hipMemcpyAsync(dst.p, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_OK(hipStreamSynchronize(stream));
    } else {
      HIP_OK(hipMemcpyAsync(dst.p, host, C * static_cast<size_t>(D) * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_OK(hipStreamSynchronize(stream));
    }
  }
  void download_rows(const DevBuf<double>& src, double* host) {
    use_device();
    HIP_OK(hipMemcpy2DAsync(host, sizeof(double) * D, src.p, sizeof(double) * Dp, sizeof(double) * D, C,
                            hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
  }
  template <class T>
  void download(const DevBuf<T>& src, T* host, size_t count) {
    use_device();
    HIP_OK(hipMemcpyAsync(host, src.p, count * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
  }
  void fill(DevBuf<double>& b, double v) {
    const int blocks = static_cast<int>(std::min<size_t>((b.n + 255) / 256, 4096));
    hipLaunchKernelGGL(wn::fill_kernel, dim3(blocks), dim3(256), 0, stream, b.p, static_cast<long long>(b.n), v);
    HIP_OK(hipGetLastError());
  }
  // throws if any transition of any chain SINCE THE PREVIOUS CHECK reported a device-side error: the kernels OR their
  // error bits into one word, which is read and cleared here (a caller that supplied too few variates, or hit a pool
  // limit, can correct that and carry on; the draws of the failed transitions are not valid)
  void check_transitions() {
    use_device();
    uint32_t flags = 0;
    HIP_OK(hipMemcpyAsync(&flags, error_flags.p, sizeof(flags), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemsetAsync(error_flags.p, 0, sizeof(uint32_t), stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (flags & wn::kErrPoolExhausted)
      throw std::runtime_error("a chain exhausted the span pool: its draws are not valid (lower max_trajectory_doublings)");
    if (flags & wn::kErrVariatesExhausted)
      throw std::runtime_error("a transition consumed more host-fed uniforms than wn_engine_set_variates supplied");
  }

  void ensure_adapters() {
    if (adapters_ready) return;
    use_device();
    const int blocks = static_cast<int>(std::min<size_t>((C * Dp + 255) / 256, 4096));
    hipLaunchKernelGGL(wn::begin_warmup_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<int>(C), Dp,
                       cfg.mass_init_count, mass.p, draw_mean.p, draw_ssd.p, score_mean.p, score_ssd.p,
                       est_weight.p, step_init.p, adam.p, mm_state.p);
    HIP_OK(hipGetLastError());
    adapters_ready = true;
    warmup_iter = 0;
    est_pending = false;
  }

  wn::Params make_params(bool warm, double* draws_dev, int64_t draws_stride, int fused = 1, int64_t draws_tstride = 0) {
    wn::Params P{};
    P.num_chains = static_cast<int32_t>(C);
    P.dim = D;
    P.dim_padded = Dp;
    P.warmup = warm ? 1 : 0;
    P.theta = theta.p;
    P.inv_mass = inv_mass.p;
    P.chol_mass = chol_mass.p;
    P.est_draw_mean = draw_mean.p;
    P.est_draw_ssd = draw_ssd.p;
    P.est_score_mean = score_mean.p;
    P.est_score_ssd = score_ssd.p;
    P.step_size = step_size.p;
    P.min_micro = min_micro.p;
    P.adam = adam.p;
    P.est_weight = est_weight.p;
    P.mm_state = mm_state.p;
    P.logp_out = logp.p;
    P.depth_out = depth.p;
    P.grad_evals = grad_evals.p;
    P.rng_draws = rng_draws.p;
    P.failed_ext = failed_ext.p;
    P.lp_stats = lp_stats.p;
    P.draws_out = draws_dev;
    P.draws_stride = draws_stride;
    P.draws_tstride = draws_tstride;
    P.fused = fused;
    P.model_params = model_params.p;
    P.max_depth = cfg.max_trajectory_doublings;
    P.max_halvings = cfg.max_step_halvings;
    P.cfg_min_micro = cfg.min_micro_steps;
    P.fma = cfg.fused_multiply_add ? 1 : 0;
    P.max_error = cfg.max_hamiltonian_error;
    P.mass_init_count = cfg.mass_init_count;
    P.macro_target = cfg.max_macro_steps_target;
    P.adam_target = cfg.step_accept_rate_target;
    P.adam_lr = cfg.step_learning_rate;
    P.adam_b1 = cfg.step_gradient_decay;
    P.adam_b2 = cfg.step_sq_gradient_decay;
    P.adam_eps = cfg.step_stabilization;
    P.adam_decay = cfg.step_learn_rate_decay;
    P.seed = seed;
    P.chain_offset = chain_offset;
    P.transition = transition;
    P.rng_mode = variates_pending ? wn::kRngBuffer : wn::kRngPhilox;
    P.u_stride = u_stride;
    P.z_buf = z_buf.p;
    P.u_buf = u_buf.p;
    P.warmup_iter = warmup_iter;
    P.arena = arena.p;
    P.arena_stride = arena_stride;
    P.pool_lds = pool_lds;
    P.im_in_lds = (im_in_lds ? 1u : 0u) | (no_far_end_sums ? 2u : 0u) | (hold_moving_end ? 4u : 0u);
    P.pool_total = pool_total;
    P.est_mode = (warm && est_pending) ? 1 : 0;
    P.work_counter = counter.p;
    P.error_flags = error_flags.p;
    return P;
  }

  void feed_reference_streams();
  void advance_reference_streams();

  // One launch = `fused` transitions of every chain, back to back on the workgroup that fetched the chain (the chain's
  // k-th draw row at draws_dev + chain * draws_stride + k * draws_tstride).  Host-fed variates cover one transition.
  void step(bool warm, double* draws_dev, int64_t draws_stride, int fused = 1, int64_t draws_tstride = 0,
            bool flush_only = false) {
    if (fused < 1) throw std::invalid_argument("transitions per launch must be at least 1");
    if (fused > 1 && (ref_streams || variates_pending))
      throw std::invalid_argument("host-fed variates cover one transition: transitions per launch must be 1");
    // (a sampling launch reads the frozen planes; freeze has applied a pending observation -- a caller that samples
    // without freezing gets it applied here)
    if (!flush_only && !warm && est_pending) flush_pending_observation();
    in_step = !ref_streams;  // (a transition launch does not make `stream` wait for the groups -- unless variates are
                             // fed from the host first, which writes buffers the groups' previous launches read)
    struct Leave {
      bool& flag;
      ~Leave() { flag = false; }
    } leave{in_step};
    use_device();
    if (ref_streams && !flush_only) feed_reference_streams();
    in_step = true;
    wn::Params P = make_params(warm, draws_dev, draws_stride, fused, draws_tstride);
    if (flush_only) P.est_mode = 2;
    if (groups > 1 && main_moved) {  // the group streams catch up with what `stream` did since their last launches
      HIP_OK(hipEventRecord(main_point, stream));
      for (int g = 1; g < groups; ++g) HIP_OK(hipStreamWaitEvent(gstream[g], main_point, 0));
      main_moved = false;
    }
    std::pair<hipEvent_t, hipEvent_t>* timed = (timing && !flush_only) ? &next_events() : nullptr;
    for (int g = 0; g < groups; ++g) {
      // The chain counter is never reset: every launch performs exactly as many fetches as it has chains (one per
      // processed chain), so a group's launch n starts at n * its chain count (mod 2^32) -- one memset per transition
      // less between two kernels.  (The first chain of the group is folded into the base: fetched = begin + ...)
      const uint32_t count = static_cast<uint32_t>(group_begin[g + 1] - group_begin[g]);
      P.chain_begin = static_cast<int32_t>(group_begin[g]);
      P.num_chains = static_cast<int32_t>(group_begin[g + 1]);
      P.work_counter = counter.p + g;
      P.work_base = work_base[g] - static_cast<uint32_t>(group_begin[g]);
      P.arena = arena.p + static_cast<size_t>(g) * static_cast<size_t>(grid) * static_cast<size_t>(arena_stride);
      hipStream_t s = g == 0 ? stream : gstream[g];
      try {
        // (per-launch HIP events: only between wn_engine_timing_reset and the read-back)
        if (timed != nullptr && g == 0) HIP_OK(hipEventRecord(timed->first, s));
        wn::launch_transition(model, geo, group_grid[g], smem, s, P);
        HIP_OK(hipGetLastError());
      } catch (...) {
        // a launch that did not happen fetched nothing: counter and base start over together (a kernel that did start
        // and then failed leaves the device in an error state anyway; the memset then fails too and is ignored)
        (void)hipMemsetAsync(counter.p + g, 0, sizeof(uint32_t), s);
        work_base[g] = 0;
        if (timed != nullptr) --events_used;  // (the pair taken for this launch has no end event: hand it back)
        throw;
      }
      work_base[g] += count;  // (only once the launch is known to be queued)
      if (g > 0) {
        HIP_OK(hipEventRecord(gdone[g], s));
        groups_ahead = true;
      }
    }
    if (timed != nullptr) {
      // the launch has ended when its LAST kernel has: the end event waits for every group (per-launch timing is a
      // diagnostic mode -- it joins the groups' streams at every launch, which the plain mode never does)
      for (int g = 1; g < groups; ++g) HIP_OK(hipStreamWaitEvent(stream, gdone[g], 0));
      HIP_OK(hipEventRecord(timed->second, stream));
    }
    if (flush_only) return;  // (not a transition: the stream keys and the iteration counts stay)
    ++region_launches;
    variates_pending = false;
    transition += static_cast<uint32_t>(fused);
    iteration += fused;
    if (warm) {
      warmup_iter += fused;
      est_pending = !geo.mem;  // (register kernels: the launch's last observation waits for the next prologue)
    }
    if (ref_streams) advance_reference_streams();
  }
};

// the pending observation by itself: one launch of the warmup kernel in its observe-only mode, over every chain group
void wn_engine::flush_pending_observation() {
  if (!est_pending || in_flush) return;
  in_flush = true;
  struct Leave {
    wn_engine& e;
    ~Leave() {
      e.in_flush = false;
      e.in_step = false;
    }
  } leave{*this};
  step(true, nullptr, 0, 1, 0, /*flush_only=*/true);
  est_pending = false;
}

void wn_engine::feed_reference_streams() {
  ReferenceStreams& r = *ref_streams;
  const size_t Dn = static_cast<size_t>(D);
  for (size_t m = 0; m < C; ++m) {
    for (size_t i = 0; i < Dn; ++i) r.z[m * Dn + i] = r.normal[m](r.eng[m]);  // util.hpp:124-127, index order
    r.after_normals[m] = r.eng[m];
    std::mt19937_64 look = r.eng[m];
    // uniform_real_distribution(0,1) and bernoulli_distribution(0.5) both draw one generate_canonical value
    for (int j = 0; j < r.pool; ++j) r.u[m * r.pool + j] = std::generate_canonical<double, 53>(look);
  }
  if (z_buf.n == 0) z_buf.alloc(C * static_cast<size_t>(Dp));
  if (u_buf.n < C * static_cast<size_t>(r.pool)) u_buf.alloc(C * static_cast<size_t>(r.pool));
  HIP_OK(hipMemsetAsync(z_buf.p, 0, z_buf.n * sizeof(double), stream));
  upload_rows(z_buf, r.z.data(), 0.0);
  HIP_OK(hipMemcpyAsync(u_buf.p, r.u.data(), C * static_cast<size_t>(r.pool) * sizeof(double), hipMemcpyHostToDevice,
                        stream));
  HIP_OK(hipStreamSynchronize(stream));
  u_stride = r.pool;
  variates_pending = true;
}

void wn_engine::advance_reference_streams() {
  ReferenceStreams& r = *ref_streams;
  download(rng_draws, r.used.data(), C);
  for (size_t m = 0; m < C; ++m) {
    if (r.used[m] > r.pool) throw std::runtime_error("reference-stream pool exhausted");
    r.eng[m] = r.after_normals[m];
    r.eng[m].discard(static_cast<unsigned long long>(r.used[m]));
  }
}

namespace {

int required_pool(const wn_config& c) {
  // other end of the accumulated span 3 + its selection 1, one entry (<= 3 vectors) per stack level
  // 1..max_depth-2, the span under construction 3, the parked state of a reversibility check 3, slack
  const int levels = std::max(1, c.max_trajectory_doublings - 1);
  return 4 + 3 * levels + 3 + 3 + 2;
}

void build_engine(wn_engine& e, int model, int num_params, const double* model_params, size_t num_chains,
                  const wn_config& cfg) {
  if (num_params < 1) throw std::invalid_argument("num_params must be positive");
  if (num_chains < 1) throw std::invalid_argument("num_chains must be positive");
  if (!wn::registry_error().empty()) throw std::invalid_argument(wn::registry_error());
  const wn::ModelOps& ops = wn::model_ops(model);  // throws for an id no model registered
  if (cfg.max_trajectory_doublings < 1) throw std::invalid_argument("max_nuts_depth must be positive");
  if (cfg.max_trajectory_doublings > wn::kMaxLevels + 1)
    throw std::invalid_argument("max_trajectory_doublings exceeds the device span stack");
  if (cfg.max_step_halvings < 1) throw std::invalid_argument("max_step_halvings must be positive");
  if (cfg.min_micro_steps < 1) throw std::invalid_argument("min_micro_steps must be positive");
  if (!(cfg.max_hamiltonian_error > 0) || !std::isfinite(cfg.max_hamiltonian_error))
    throw std::invalid_argument("max_hamiltonian_error must be positive and finite");
  if (ops.uses_params && model_params == nullptr)
    throw std::invalid_argument(std::string(ops.name) + " model needs a parameter vector of num_params doubles");
  ops.validate(num_params);

  e.model = model;
  e.D = num_params;
  e.C = num_chains;
  e.cfg = cfg;
  e.device = cfg.device;
  e.geo = wn::choose_geometry(num_params, cfg.waves_per_chain, cfg.elems_per_lane, ops.uses_params, ops.preferred_epl(num_params),
                              ops.hold_tiles(wn::kHeldWaves), ops.register_dim_limit);
  e.Dp = wn::padded_dim(e.geo, num_params);
  e.use_device();
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, e.device));
  e.num_cus = prop.multiProcessorCount;
  HIP_OK(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking));

  // residency: how many chains (workgroups) share a CU, and how much of the span pool sits in LDS
  const size_t lds_per_cu = 160 * 1024;
  e.pool_total = required_pool(cfg) + (e.geo.mem ? wn::kMemRoleVectors : 0);
  if (e.pool_total > wn::kMaxPool)
    throw std::invalid_argument("max_trajectory_doublings needs more span-pool vectors than the device free mask holds");
  const int wps = wn::waves_per_simd(model, e.geo);
  const int hold_tiles = e.geo.mem ? ops.hold_tiles(e.geo.nw) : 0;
  const bool hold_fits = hold_tiles > 0 && num_params <= 2 * 64 * e.geo.nw * hold_tiles;
  const size_t vec_bytes = sizeof(double) * e.Dp;
  int wg_per_cu = 0;
  // residency for `want` workgroups per CU (0: the default for this geometry) -> whether the moving end is held
  auto residency = [&](int want) {
    wg_per_cu = want > 0 ? want : wn::default_workgroups_per_cu(e.geo, wps);
    wg_per_cu = std::max(1, std::min(wg_per_cu, 32 / e.geo.nw));
    if (!e.geo.mem) wg_per_cu = std::min(wg_per_cu, std::max(1, 4 * wps / e.geo.nw));
    const size_t fixed = wn::transition_smem_bytes(e.geo.nw, 0, e.Dp);
    const size_t budget = lds_per_cu / wg_per_cu;
    if (fixed > budget) throw std::invalid_argument("workgroups_per_cu too high for the LDS-resident state");
    int lds_vecs = budget > fixed + 256 ? static_cast<int>((budget - fixed - 256) / vec_bytes) : 0;
    if (cfg.lds_vectors >= 0 && cfg.lds_vectors < lds_vecs) lds_vecs = cfg.lds_vectors;
    if (e.geo.mem) lds_vecs = 0;  // streaming backend: vectors are far larger than LDS
    e.pool_lds = std::min(lds_vecs, e.pool_total);
    e.smem = wn::transition_smem_bytes(e.geo.nw, e.pool_lds, e.Dp);
    e.im_in_lds = false;
    e.hold_moving_end = false;
    if (e.geo.mem) {
      // one more vector per workgroup, if the CU's LDS holds it for every resident workgroup: the inverse mass
      const char* off = std::getenv("WALNUTS_AMD_NO_LDS_MASS");
      const char* nf = std::getenv("WALNUTS_AMD_NO_FAR_END_SUMS");
      e.no_far_end_sums = nf != nullptr && nf[0] == '1';
      if (e.smem + vec_bytes <= budget && !(off != nullptr && off[0] == '1')) {
        e.im_in_lds = true;
        e.smem += vec_bytes;
        // ... and, if the chain's vectors fit the registers the kernels set aside for it, the moving end (TrajMem, HOLD)
        const char* nh = std::getenv("WALNUTS_AMD_NO_HELD_STATE");
        // (such a kernel keeps the exp / log tables in LDS too, and a halo model's wavefront-edge elements)
        const size_t tables = sizeof(double) * (wn::kLdsTableDoubles + 2 * 2 * wn::kMemHoldTiles * e.geo.nw);  // (two copies of the edges)
        e.hold_moving_end = hold_fits && e.smem + tables <= budget && !(nh != nullptr && nh[0] == '1');
        if (e.hold_moving_end) e.smem += tables;
      }
    }
    return e.hold_moving_end;
  };
  if (cfg.workgroups_per_cu > 0) {
    residency(cfg.workgroups_per_cu);
  } else if (hold_fits) {
    // a streaming kernel that can hold the moving end in registers wants the CU -- its LDS for the inverse mass, a
    // wavefront's full register budget -- for ONE chain; if the hold is then refused (no room for the inverse mass
    // and the tables, or switched off), the kernel that streams both ends gets its usual residency back
    if (!residency(1)) residency(0);
  } else {
    residency(0);
  }
  const int usable_cus = std::max(1, e.num_cus - std::max(0, cfg.reserved_cus));
  e.grid = static_cast<int>(std::min<size_t>(num_chains, static_cast<size_t>(usable_cus) * wg_per_cu));

  const size_t plane = num_chains * static_cast<size_t>(e.Dp);
  for (DevBuf<double>* b : {&e.theta, &e.mass, &e.inv_mass, &e.chol_mass, &e.draw_mean, &e.draw_ssd, &e.score_mean, &e.score_ssd})
    b->alloc(plane);
  e.step_init.alloc(num_chains);
  e.step_size.alloc(num_chains);
  e.adam.alloc(6 * num_chains);
  e.est_weight.alloc(2 * num_chains);
  e.mm_state.alloc(2 * num_chains);
  e.logp.alloc(num_chains);
  e.min_micro.alloc(num_chains);
  e.depth.alloc(num_chains);
  e.rng_draws.alloc(num_chains);
  e.failed_ext.alloc(num_chains);
  e.grad_evals.alloc(num_chains);
  // chain groups: as configured, or two when there are more chains than resident workgroups (with at most one chain
  // per workgroup there is no tail to fill: 1024 and 256 chains measured the same with 1-4 groups); host-fed variates
  // and an adopted stream (wn_engine_set_stream) go back to one
  {
    int want = cfg.chain_groups;
    if (const char* v = std::getenv("WALNUTS_AMD_CHAIN_GROUPS")) want = std::atoi(v);
    // (... and one when a CU holds a single workgroup of this kernel -- the streaming kernels with the inverse mass in
    // LDS --: the second group's workgroups then start only as the first group's retire, i.e. two tails instead of one;
    // config #4 measured 15.3 ms per step with one group against 15.9 ms with two)
    if (want <= 0) want = (num_chains > static_cast<size_t>(e.grid) && !(e.geo.mem && wg_per_cu == 1)) ? 2 : 1;
    e.groups = std::max(1, std::min({want, wn_engine::kMaxGroups, static_cast<int>(num_chains)}));
  }
  for (int g = 0; g <= e.groups; ++g) e.group_begin[g] = num_chains * static_cast<size_t>(g) / static_cast<size_t>(e.groups);
  e.gstream[0] = e.stream;
  for (int g = 0; g < e.groups; ++g) {
    e.group_grid[g] = static_cast<int>(std::min<size_t>(e.group_begin[g + 1] - e.group_begin[g], static_cast<size_t>(e.grid)));
    if (g > 0) {
      HIP_OK(hipStreamCreateWithFlags(&e.gstream[g], hipStreamNonBlocking));
      HIP_OK(hipEventCreateWithFlags(&e.gdone[g], hipEventDisableTiming));
    }
  }
  if (e.groups > 1) HIP_OK(hipEventCreateWithFlags(&e.main_point, hipEventDisableTiming));
  e.counter.alloc(wn_engine::kMaxGroups);
  HIP_OK(hipMemsetAsync(e.counter.p, 0, wn_engine::kMaxGroups * sizeof(uint32_t), e.stream));
  e.error_flags.alloc(1);
  HIP_OK(hipMemsetAsync(e.error_flags.p, 0, sizeof(uint32_t), e.stream));
  e.lp_stats.alloc(3 * num_chains);
  e.mon_partial.alloc(2 * static_cast<size_t>(wn::monitor_runs(static_cast<int>(num_chains))));
  e.mon_out.alloc(4);
  e.mon_colsum.alloc(e.Dp);
  e.mon_rel_mass.alloc(num_chains);
  e.mon_rel_step.alloc(num_chains);
  e.scratch64.alloc(1);
  // what LDS does not hold (deep trees only) overflows to a per-workgroup HBM arena
  const size_t arena_vecs = static_cast<size_t>(std::max(0, e.pool_total - e.pool_lds)) +
                            (e.geo.mem ? wn::kMemScratchVectors : 0);
  e.arena_stride = static_cast<int64_t>(arena_vecs) * e.Dp;
  e.arena.alloc(std::max<size_t>(1, static_cast<size_t>(e.groups) * static_cast<size_t>(e.grid) * arena_vecs * e.Dp));
  e.model_params.alloc(e.Dp);

  // InitConfigBuilder defaults (config.hpp:197-207): step 0.1, positions 0, masses 1
  HIP_OK(hipMemsetAsync(e.theta.p, 0, plane * sizeof(double), e.stream));
  e.fill(e.mass, 1.0);
  e.fill(e.inv_mass, 1.0);
  e.fill(e.step_init, 0.1);
  HIP_OK(hipMemsetAsync(e.grad_evals.p, 0, num_chains * sizeof(int64_t), e.stream));
  HIP_OK(hipMemsetAsync(e.depth.p, 0, num_chains * sizeof(int32_t), e.stream));
  HIP_OK(hipMemsetAsync(e.rng_draws.p, 0, num_chains * sizeof(int32_t), e.stream));
  HIP_OK(hipMemsetAsync(e.failed_ext.p, 0, num_chains * sizeof(int32_t), e.stream));
  HIP_OK(hipMemsetAsync(e.logp.p, 0, num_chains * sizeof(double), e.stream));
  HIP_OK(hipMemsetAsync(e.lp_stats.p, 0, 3 * num_chains * sizeof(double), e.stream));
  {
    std::vector<double> mp(e.Dp, 1.0);
    if (model_params) std::copy(model_params, model_params + num_params, mp.begin());
    ops.host_params(mp.data(), num_params);  // the model's own validation / transformation (wn_models.h)
    HIP_OK(hipMemcpyAsync(e.model_params.p, mp.data(), mp.size() * sizeof(double), hipMemcpyHostToDevice, e.stream));
    HIP_OK(hipStreamSynchronize(e.stream));
  }
  wn::prepare_kernels(model, e.geo, e.smem);
}

void run_init(wn_engine& e, bool pos, bool masses, bool step, double scale, double smoothing, uint64_t pos_seed,
              uint32_t pos_off, uint64_t step_seed, uint32_t step_off, const double* z_dev = nullptr) {
  e.use_device();
  wn::InitParams Q{};
  Q.num_chains = static_cast<int32_t>(e.C);
  Q.dim = e.D;
  Q.dim_padded = e.Dp;
  Q.do_positions = pos;
  Q.do_masses = masses;
  Q.do_step = step;
  Q.theta = e.theta.p;
  Q.mass = e.mass.p;
  Q.step_init = e.step_init.p;
  Q.grad_evals = e.grad_evals.p;
  Q.model_params = e.model_params.p;
  Q.z_buf = z_dev;
  Q.scratch = e.arena.p;
  Q.scratch_stride = e.arena_stride;
  Q.scale = scale;
  Q.smoothing = smoothing;
  Q.pos_seed = pos_seed;
  Q.step_seed = step_seed;
  Q.pos_chain_offset = pos_off;
  Q.step_chain_offset = step_off;
  const int grid = e.geo.mem ? e.grid : static_cast<int>(std::min<size_t>(e.C, static_cast<size_t>(e.num_cus) * 8));
  wn::launch_init(e.model, e.geo, grid, wn::transition_smem_bytes(e.geo.nw, 0, e.Dp), e.stream, Q);
  HIP_OK(hipGetLastError());
  e.adapters_ready = false;
}

}  // namespace

extern "C" {

const char* walnutpie_get_error_message(const WalnutpyError* err) {
  if (err == nullptr) return "Something went wrong: No error found";
  return err->msg.c_str();
}
WalnutpyErrorType walnutpie_get_error_type(const WalnutpyError* err) { return err == nullptr ? generic : err->type; }
void walnutpie_destroy_error(WalnutpyError* err) { delete err; }

// which counter-based stream definition this build draws from (wn_devmath.h kStreamVersion: the map from
// (seed, chain, transition, index) to variates; results at a fixed seed are comparable only within one version)
int wn_stream_version(void) { return wnd::kStreamVersion; }
// the code-generation flags this library was compiled with (Makefile CODEGEN_FLAGS): run-time models use the same
#ifndef WN_CODEGEN_FLAGS
#define WN_CODEGEN_FLAGS ""
#endif
const char* wn_build_flags(void) { return WN_CODEGEN_FLAGS; }
#ifndef WN_COMPILER_VERSION
#define WN_COMPILER_VERSION ""
#endif
const char* wn_build_compiler(void) { return WN_COMPILER_VERSION; }

int wn_internal_sqrt_probe(const double* x, double* y, size_t n, int checked) {
  DevBuf<double> dx, dy;
  try {
    dx.alloc(n);
    dy.alloc(n);
    HIP_OK(hipMemcpyAsync(dx.p, x, n * sizeof(double), hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(sqrt_probe_kernel, dim3(1024), dim3(256), 0, nullptr, dx.p, dy.p, static_cast<long long>(n), checked);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(y, dy.p, n * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    HIP_OK(hipStreamSynchronize(nullptr));
  } catch (...) {
    return -1;
  }
  return 0;
}

int wn_model_id(const char* name) {
  if (name == nullptr) return -1;
  for (int i = 0; i < wn::kMaxModels; ++i) {
    const wn::ModelOps* ops = wn::model_table()[i];
    if (ops != nullptr && std::strcmp(ops->name, name) == 0) return i;
  }
  return -1;
}

// ---- device models compiled at run time (walnuts_amd/models.py; INTEGRATION.md "Adding a device model") --------------
// the registration of a model's own shared object, called from its static initialiser when it is loaded
int wn_plugin_register_model(const void* ops, const void* abi) {
  const auto* theirs = static_cast<const wn::ModelAbi*>(abi);
  const wn::ModelAbi ours = wn::model_abi();
  const auto* m = static_cast<const wn::ModelOps*>(ops);
  if (theirs == nullptr || m == nullptr || theirs->version != ours.version || theirs->sizeof_ops != ours.sizeof_ops ||
      theirs->sizeof_params != ours.sizeof_params || theirs->sizeof_geometry != ours.sizeof_geometry) {
    wn::registry_error() = "a device model was compiled against other headers than this library (wn_launch.h "
                           "kModelAbiVersion / struct sizes differ): rebuild it with walnuts_amd.build_device_model";
    return -1;
  }
  return wn::register_model_here(m) ? 0 : -1;
}
// what went wrong in the last registration ("" if nothing has); the message stays until the next failure
const char* wn_model_error(void) { return wn::registry_error().c_str(); }
// forget a failed registration (a run-time model whose id was taken is reported once, not by every later engine)
void wn_model_clear_error(void) { wn::registry_error().clear(); }
// Launch geometries.  The engine's choice depends on the MODEL as well as on num_params and the wn_config's requests:
// a model with held streaming kernels (ModelOps::hold_tiles) leaves the register kernels at its register_dim_limit.
// wn_geometry_for_model: the ONE geometry build_engine picks for a REGISTERED model (the same choose_geometry call).
int wn_geometry_for_model(int model, int num_params, int waves_per_chain, int elems_per_lane, int* nw, int* epl,
                          int* streaming, WalnutpyError** err) {
  return guarded(err, [&] {
    if (num_params < 1) throw std::invalid_argument("num_params must be positive");
    const wn::ModelOps& ops = wn::model_ops(model);
    const wn::Geometry g = wn::choose_geometry(num_params, waves_per_chain, elems_per_lane, ops.uses_params, ops.preferred_epl(num_params),
                                               ops.hold_tiles(wn::kHeldWaves), ops.register_dim_limit);
    if (nw != nullptr) *nw = g.nw;
    if (epl != nullptr) *epl = g.epl;
    if (streaming != nullptr) *streaming = g.mem ? 1 : 0;
  });
}
// wn_geometry_candidates: EVERY geometry build_engine may pick for these requests, over all traits a model can have
// (no held streaming kernels; held kernels with the register kernels up to 4 096 or up to 8 192 parameters) -- what a
// model that is compiled at run time, and therefore not registered yet, has to instantiate.  out: triples
// (waves per chain, elements per lane, streaming), at most `max` of them; *count = how many there are.
int wn_geometry_candidates(int num_params, int waves_per_chain, int elems_per_lane, int preferred_elems_per_lane,
                           int* out, int max, int* count, WalnutpyError** err) {
  return guarded(err, [&] {
    if (num_params < 1) throw std::invalid_argument("num_params must be positive");
    if (count == nullptr) throw std::invalid_argument("null argument");
    const int traits[3][2] = {{0, wn::kMaxRegisterDim}, {wn::kMemHoldTiles, 4096}, {wn::kMemHoldTiles, 8192}};
    int n = 0;
    wn::Geometry seen[3];
    for (const auto& t : traits) {
      const wn::Geometry g = wn::choose_geometry(num_params, waves_per_chain, elems_per_lane, false,
                                                 preferred_elems_per_lane, t[0], t[1]);
      bool dup = false;
      for (int i = 0; i < n; ++i) dup = dup || (seen[i].nw == g.nw && seen[i].epl == g.epl && seen[i].mem == g.mem);
      if (dup) continue;
      seen[n] = g;
      if (out != nullptr && n < max) {
        out[3 * n] = g.nw;
        out[3 * n + 1] = g.epl;
        out[3 * n + 2] = g.mem ? 1 : 0;
      }
      ++n;
    }
    *count = n;
  });
}
// (kept: the choice for a model WITHOUT held streaming kernels and with the default register limit)
int wn_geometry_for(int num_params, int waves_per_chain, int elems_per_lane, int preferred_elems_per_lane, int* nw,
                    int* epl, int* streaming, WalnutpyError** err) {
  return guarded(err, [&] {
    if (num_params < 1) throw std::invalid_argument("num_params must be positive");
    const wn::Geometry g = wn::choose_geometry(num_params, waves_per_chain, elems_per_lane, false, preferred_elems_per_lane);
    if (nw != nullptr) *nw = g.nw;
    if (epl != nullptr) *epl = g.epl;
    if (streaming != nullptr) *streaming = g.mem ? 1 : 0;
  });
}

// WALNUTS_AMD_FMA=0/1 overrides the library default (fused) for callers that do not build a wn_config themselves
// (walnutpie_sample_device keeps the reference's argument list)
static int default_fma() {
  const char* v = std::getenv("WALNUTS_AMD_FMA");
  return (v != nullptr && v[0] == '0') ? 0 : 1;
}

void wn_default_config(wn_config* c) {
  c->max_trajectory_doublings = 5;
  c->max_step_halvings = 5;
  c->min_micro_steps = 1;
  c->device = 0;
  c->max_hamiltonian_error = 0.5;
  c->mass_init_count = 4.0;
  c->max_macro_steps_target = 15.0;
  c->step_accept_rate_target = 0.8;
  c->step_learning_rate = 0.05;
  c->step_gradient_decay = 0.8;
  c->step_sq_gradient_decay = 0.9;
  c->step_stabilization = 1e-4;
  c->step_learn_rate_decay = 0.5;
  c->waves_per_chain = 0;
  c->elems_per_lane = 0;
  c->workgroups_per_cu = 0;
  c->lds_vectors = -1;
  c->fused_multiply_add = default_fma();
  c->reserved_cus = 0;
  c->chain_groups = 0;
}

int wn_engine_create(wn_engine** out, int model, int num_params, const double* model_params, size_t num_chains,
                     const wn_config* cfg, WalnutpyError** err) {
  return guarded(err, [&] {
    if (out == nullptr || cfg == nullptr) throw std::invalid_argument("null argument");
    auto e = std::make_unique<wn_engine>();
    build_engine(*e, model, num_params, model_params, num_chains, *cfg);
    *out = e.release();
  });
}
void wn_engine_destroy(wn_engine* e) { delete e; }

int wn_engine_set_positions(wn_engine* e, const double* positions, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || positions == nullptr) throw std::invalid_argument("null argument"); e->upload_rows(e->theta, positions, 0.0); });
}
int wn_engine_set_masses(wn_engine* e, const double* masses, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || masses == nullptr) throw std::invalid_argument("null argument");
    for (size_t i = 0; i < e->C * static_cast<size_t>(e->D); ++i)
      if (!(masses[i] > 0) || !std::isfinite(masses[i])) throw std::invalid_argument("masses must be positive and finite");
    e->fill(e->mass, 1.0);
    e->upload_rows(e->mass, masses, 1.0);
    e->adapters_ready = false;
  });
}
int wn_engine_set_step_sizes(wn_engine* e, const double* steps, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || steps == nullptr) throw std::invalid_argument("null argument");
    for (size_t i = 0; i < e->C; ++i)
      if (!(steps[i] > 0) || !std::isfinite(steps[i])) throw std::invalid_argument("step size must be positive and finite");
    e->use_device();
    HIP_OK(hipMemcpyAsync(e->step_init.p, steps, e->C * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    e->adapters_ready = false;
  });
}
int wn_engine_init_positions(wn_engine* e, uint64_t seed, uint32_t chain_offset, double scale, WalnutpyError** err) {
  return guarded(err, [&] {
    if (!(scale > 0) || !std::isfinite(scale)) throw std::invalid_argument("init_scale must be positive and finite");
    run_init(*e, true, false, false, scale, 0.0, seed, chain_offset, 0, 0);
  });
}
int wn_engine_init_masses_from_grad(wn_engine* e, double smoothing, WalnutpyError** err) {
  return guarded(err, [&] {
    if (!(smoothing > 0 && smoothing < 1)) throw std::invalid_argument("mass_smoothing must be in (0, 1)");
    run_init(*e, false, true, false, 1.0, smoothing, 0, 0, 0, 0);
  });
}
int wn_engine_average_masses(wn_engine* e, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    const int C = static_cast<int>(e->C);
    hipLaunchKernelGGL(wn::mass_log_colsum_kernel, dim3((e->D + 255) / 256), dim3(256), 0, e->stream, C, e->D, e->Dp,
                       e->mass.p, e->mon_colsum.p);
    const int blocks = static_cast<int>(std::min<size_t>((e->C * e->Dp + 255) / 256, 4096));
    hipLaunchKernelGGL(wn::mass_broadcast_kernel, dim3(blocks), dim3(256), 0, e->stream, C, e->D, e->Dp,
                       e->mon_colsum.p, e->mass.p);
    HIP_OK(hipGetLastError());
    e->adapters_ready = false;
  });
}
int wn_engine_get_masses(wn_engine* e, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument"); e->download_rows(e->mass, out); });
}
int wn_engine_adapt_step(wn_engine* e, uint64_t seed, uint32_t chain_offset, WalnutpyError** err) {
  return guarded(err, [&] { run_init(*e, false, false, true, 1.0, 0.0, 0, 0, seed, chain_offset); });
}
int wn_engine_adapt_step_with_normals(wn_engine* e, const double* normals, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    if (e->z_buf.n == 0) e->z_buf.alloc(e->C * static_cast<size_t>(e->Dp));
    HIP_OK(hipMemsetAsync(e->z_buf.p, 0, e->z_buf.n * sizeof(double), e->stream));
    e->upload_rows(e->z_buf, normals, 0.0);
    run_init(*e, false, false, true, 1.0, 0.0, 0, 0, 0, 0, e->z_buf.p);
  });
}
void* wn_internal_make_error(const char* msg, int type) {
  return new WalnutpyError{msg, static_cast<WalnutpyErrorType>(type)};
}
int wn_engine_seed(wn_engine* e, uint64_t seed, uint32_t chain_offset, WalnutpyError** err) {
  return guarded(err, [&] {
    e->seed = seed;
    e->chain_offset = chain_offset;
    e->transition = 0;
  });
}
int wn_engine_seed_reference_streams(wn_engine* e, uint64_t seed, WalnutpyError** err) {
  return guarded(err, [&] {
    auto r = std::make_unique<ReferenceStreams>();
    r->eng.reserve(e->C);
    for (size_t m = 0; m < e->C; ++m) {
      std::seed_seq ss{static_cast<size_t>(seed), m + 1u};  // api.hpp:48-49
      r->eng.emplace_back(ss);
    }
    r->normal.assign(e->C, std::normal_distribution<double>(0.0, 1.0));
    r->after_normals = r->eng;
    r->z.assign(e->C * static_cast<size_t>(e->D), 0.0);
    // scalar draws per transition: one bernoulli + one Metropolis uniform per doubling, one Barker uniform
    // per inner merge: at most 2^max_depth - 1 + max_depth
    const int md = e->cfg.max_trajectory_doublings;
    if (md > 16) throw std::invalid_argument("reference streams support max_trajectory_doublings <= 16");
    r->pool = (1 << md) - 1 + md;
    r->u.assign(e->C * static_cast<size_t>(r->pool), 0.0);
    r->used.assign(e->C, 0);
    e->ref_streams = std::move(r);
    e->seed = seed;
    e->transition = 0;
  });
}
int wn_engine_set_variates(wn_engine* e, const double* normals, const double* uniforms, int u_per_chain,
                           WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || normals == nullptr || uniforms == nullptr) throw std::invalid_argument("null argument");
    if (u_per_chain < 1) throw std::invalid_argument("u_per_chain must be positive");
    e->use_device();
    if (e->z_buf.n == 0) e->z_buf.alloc(e->C * static_cast<size_t>(e->Dp));
    if (e->u_buf.n < e->C * static_cast<size_t>(u_per_chain)) e->u_buf.alloc(e->C * static_cast<size_t>(u_per_chain));
    HIP_OK(hipMemsetAsync(e->z_buf.p, 0, e->z_buf.n * sizeof(double), e->stream));
    e->upload_rows(e->z_buf, normals, 0.0);
    HIP_OK(hipMemcpyAsync(e->u_buf.p, uniforms, e->C * static_cast<size_t>(u_per_chain) * sizeof(double),
                          hipMemcpyHostToDevice, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    e->u_stride = u_per_chain;
    e->variates_pending = true;
  });
}

int wn_engine_warmup_step(wn_engine* e, double* draws_dev, int64_t draws_stride, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e->frozen) throw std::runtime_error("warmup_step after freeze");
    e->ensure_adapters();
    e->step(true, draws_dev, draws_stride);
  });
}
int wn_engine_warmup_steps(wn_engine* e, int transitions, double* draws_dev, int64_t draws_stride,
                           int64_t draws_transition_stride, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr) throw std::invalid_argument("null argument");
    if (e->frozen) throw std::runtime_error("warmup_step after freeze");
    e->ensure_adapters();
    e->step(true, draws_dev, draws_stride, transitions, draws_transition_stride);
  });
}
int wn_engine_freeze(wn_engine* e, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr) throw std::invalid_argument("null argument");
    e->ensure_adapters();
    e->use_device();
    const int blocks = static_cast<int>(std::min<size_t>((e->C * e->Dp + 255) / 256, 4096));
    hipLaunchKernelGGL(wn::freeze_kernel, dim3(blocks), dim3(256), 0, e->stream, static_cast<int>(e->C), e->Dp,
                       e->draw_ssd.p, e->score_ssd.p, e->est_weight.p, e->adam.p, e->mm_state.p,
                       e->cfg.max_macro_steps_target, e->cfg.min_micro_steps, e->inv_mass.p, e->chol_mass.p,
                       e->step_size.p, e->min_micro.p);
    HIP_OK(hipGetLastError());
    e->frozen = true;
    if (e->ref_streams)  // WalnutsSampler builds a new detail::Random over the same engine (walnuts.hpp:642)
      e->ref_streams->normal.assign(e->C, std::normal_distribution<double>(0.0, 1.0));
  });
}
int wn_engine_sample_step(wn_engine* e, double* draws_dev, int64_t draws_stride, WalnutpyError** err) {
  return guarded(err, [&] {
    if (!e->frozen) throw std::runtime_error("sample_step before freeze");
    e->step(false, draws_dev, draws_stride);
  });
}
int wn_engine_sample_steps(wn_engine* e, int transitions, double* draws_dev, int64_t draws_stride,
                           int64_t draws_transition_stride, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr) throw std::invalid_argument("null argument");
    if (!e->frozen) throw std::runtime_error("sample_step before freeze");
    e->step(false, draws_dev, draws_stride, transitions, draws_transition_stride);
  });
}
int wn_engine_synchronize(wn_engine* e, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    HIP_OK(hipStreamSynchronize(e->stream));
  });
}
int wn_engine_check(wn_engine* e, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr) throw std::invalid_argument("null argument"); e->check_transitions(); });
}

int wn_engine_get_positions(wn_engine* e, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument"); e->download_rows(e->theta, out); });
}
int wn_engine_get_inv_mass(wn_engine* e, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument");
    if (!e->frozen) {
      e->ensure_adapters();
      e->use_device();
      const int blocks = static_cast<int>(std::min<size_t>((e->C * e->Dp + 255) / 256, 4096));
      hipLaunchKernelGGL(wn::inv_mass_estimate_kernel, dim3(blocks), dim3(256), 0, e->stream, static_cast<int>(e->C),
                         e->Dp, e->draw_ssd.p, e->score_ssd.p, e->est_weight.p, e->inv_mass.p);
      HIP_OK(hipGetLastError());
    }
    e->download_rows(e->inv_mass, out);
  });
}
int wn_engine_get_step_sizes(wn_engine* e, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument");
    if (e->frozen) {
      e->download(e->step_size, out, e->C);
    } else if (e->adapters_ready) {
      std::vector<double> a(6 * e->C);
      e->download(e->adam, a.data(), a.size());
      for (size_t c = 0; c < e->C; ++c) out[c] = wnd::dexp(a[6 * c]);
    } else {
      e->download(e->step_init, out, e->C);
    }
  });
}
int wn_engine_get_logp(wn_engine* e, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument"); e->download(e->logp, out, e->C); });
}
int wn_engine_get_min_micro(wn_engine* e, int32_t* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument");
    if (e->frozen) {
      e->download(e->min_micro, out, e->C);
    } else {
      std::vector<double> mm(2 * e->C);
      e->download(e->mm_state, mm.data(), mm.size());
      for (size_t c = 0; c < e->C; ++c) {
        const long long est = std::llround(mm[2 * c] / mm[2 * c + 1] / e->cfg.max_macro_steps_target);
        out[c] = static_cast<int32_t>(std::max<long long>(est, e->cfg.min_micro_steps));
      }
    }
  });
}
int wn_engine_get_depths(wn_engine* e, int32_t* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument"); e->download(e->depth, out, e->C); });
}
int wn_engine_get_grad_evals(wn_engine* e, int64_t* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument"); e->download(e->grad_evals, out, e->C); });
}
int wn_engine_get_failed_extensions(wn_engine* e, int32_t* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument");
    e->download(e->failed_ext, out, e->C);
  });
}
int wn_engine_get_rng_draws(wn_engine* e, int32_t* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument"); e->download(e->rng_draws, out, e->C); });
}
int wn_engine_get_adam(wn_engine* e, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e == nullptr || out == nullptr) throw std::invalid_argument("null argument"); e->download(e->adam, out, 6 * e->C); });
}
int wn_engine_get_estimator(wn_engine* e, double* dm, double* ds, double* sm, double* ss, double* w,
                            WalnutpyError** err) {
  return guarded(err, [&] {
    e->download_rows(e->draw_mean, dm);
    e->download_rows(e->draw_ssd, ds);
    e->download_rows(e->score_mean, sm);
    e->download_rows(e->score_ssd, ss);
    e->download(e->est_weight, w, 2 * e->C);
  });
}
int wn_engine_total_grad_evals(wn_engine* e, int64_t* out, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    HIP_OK(hipMemsetAsync(e->scratch64.p, 0, sizeof(unsigned long long), e->stream));
    hipLaunchKernelGGL(wn::sum_i64_kernel, dim3(std::min<size_t>(256, (e->C + 255) / 256)), dim3(256), 0, e->stream, e->grad_evals.p,
                       static_cast<int>(e->C), e->scratch64.p);
    HIP_OK(hipGetLastError());
    unsigned long long v = 0;
    HIP_OK(hipMemcpyAsync(&v, e->scratch64.p, sizeof(v), hipMemcpyDeviceToHost, e->stream));
    HIP_OK(hipStreamSynchronize(e->stream));
    *out = static_cast<int64_t>(v);
  });
}

// ---- cross-chain monitors (adapt.hpp:172-229, sampler.hpp:117-158) ---------------------------------
int wn_engine_lp_sums(wn_engine* e, double* out /*[3]: sum of means, sum of sample variances, chains*/,
                      WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    const int C = static_cast<int>(e->C);
    const int runs = wn::monitor_runs(C);
    hipLaunchKernelGGL(wn::lp_sums_kernel, dim3((runs + 63) / 64), dim3(64), 0, e->stream, C, e->lp_stats.p,
                       e->mon_partial.p);
    hipLaunchKernelGGL(wn::finish_sums_kernel<2>, dim3(1), dim3(64), 0, e->stream, e->mon_partial.p, runs, e->mon_out.p);
    HIP_OK(hipGetLastError());
    e->download(e->mon_out, out, 2);
    out[2] = static_cast<double>(C);
  });
}
int wn_engine_lp_sq_dev(wn_engine* e, double mean_of_means, double* out, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    const int runs = wn::monitor_runs(static_cast<int>(e->C));
    hipLaunchKernelGGL(wn::lp_sqdev_kernel, dim3((runs + 63) / 64), dim3(64), 0, e->stream,
                       static_cast<int>(e->C), e->lp_stats.p, mean_of_means, e->mon_partial.p);
    hipLaunchKernelGGL(wn::finish_sums_kernel<1>, dim3(1), dim3(64), 0, e->stream, e->mon_partial.p, runs, e->mon_out.p);
    HIP_OK(hipGetLastError());
    e->download(e->mon_out, out, 1);
  });
}
int wn_engine_rhat(wn_engine* e, double* rhat, WalnutpyError** err) {
  return guarded(err, [&] {
    double s[3], q;
    WalnutpyError* inner = nullptr;
    if (wn_engine_lp_sums(e, s, &inner) != 0 || wn_engine_lp_sq_dev(e, s[0] / s[2], &q, &inner) != 0) {
      std::string msg = inner ? inner->msg : "monitor failed";
      delete inner;
      throw std::runtime_error(msg);
    }
    const double variance_of_means = q / (s[2] - 1);  // util.hpp:401-404
    const double mean_of_variances = s[1] / s[2];
    *rhat = std::sqrt(1 + variance_of_means / mean_of_variances);  // sampler.hpp:145
  });
}
// The warmup controller's statistic (adapt.hpp:193-221) in the two stages a multi-GPU driver needs: (1) this
// engine's sums over chains of log step and of log mass per dimension -- D+1 doubles to all-reduce (SUM) --,
// (2) given the sums over ALL chains, this engine's largest relative distances -- 2 doubles to all-reduce (MAX).
int wn_engine_warmup_sums(wn_engine* e, double* sum_log_step, double* colsum_log_mass, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e->frozen) throw std::runtime_error("warmup monitor after freeze");
    e->ensure_adapters();
    e->use_device();
    const int C = static_cast<int>(e->C);
    const int runs = wn::monitor_runs(C);
    hipLaunchKernelGGL(wn::log_step_sum_kernel, dim3((runs + 63) / 64), dim3(64), 0, e->stream, C, e->adam.p,
                       e->mon_partial.p);
    hipLaunchKernelGGL(wn::finish_sums_kernel<1>, dim3(1), dim3(64), 0, e->stream, e->mon_partial.p, runs, e->mon_out.p);
    hipLaunchKernelGGL(wn::log_mass_colsum_kernel, dim3((e->D + 255) / 256), dim3(256), 0, e->stream, C, e->D, e->Dp,
                       e->draw_ssd.p, e->score_ssd.p, e->est_weight.p, e->mon_colsum.p);
    HIP_OK(hipGetLastError());
    e->download(e->mon_out, sum_log_step, 1);
    e->download(e->mon_colsum, colsum_log_mass, static_cast<size_t>(e->D));
  });
}
int wn_engine_warmup_max_rel(wn_engine* e, double sum_log_step, const double* colsum_log_mass, size_t total_chains,
                             double* max_rel_diff_step, double* max_rel_diff_mass, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e->frozen) throw std::runtime_error("warmup monitor after freeze");
    if (total_chains < e->C) throw std::invalid_argument("total_chains is smaller than this engine's chain count");
    e->ensure_adapters();
    e->use_device();
    const int C = static_cast<int>(e->C);
    HIP_OK(hipMemcpyAsync(e->mon_colsum.p, colsum_log_mass, static_cast<size_t>(e->D) * sizeof(double),
                          hipMemcpyHostToDevice, e->stream));
    const double n = static_cast<double>(total_chains);
    const double mean_log_step = sum_log_step / n;  // adapt.hpp:201-202
    hipLaunchKernelGGL(wn::warmup_spread_kernel, dim3(C), dim3(256), 0, e->stream, C, e->D, e->Dp, e->draw_ssd.p,
                       e->score_ssd.p, e->est_weight.p, e->adam.p, e->mon_colsum.p, n, mean_log_step,
                       e->mon_rel_mass.p, e->mon_rel_step.p);
    hipLaunchKernelGGL(wn::max2_kernel, dim3(1), dim3(256), 0, e->stream, C, e->mon_rel_mass.p, e->mon_rel_step.p,
                       e->mon_out.p);
    HIP_OK(hipGetLastError());
    double m[2];
    e->download(e->mon_out, m, 2);
    *max_rel_diff_mass = m[0];
    *max_rel_diff_step = m[1];
  });
}
int wn_engine_warmup_spread(wn_engine* e, double* max_rel_diff_step, double* max_rel_diff_mass, WalnutpyError** err) {
  return guarded(err, [&] {
    WalnutpyError* inner = nullptr;
    double sum_log_step = 0;
    std::vector<double> colsum(static_cast<size_t>(e->D));
    if (wn_engine_warmup_sums(e, &sum_log_step, colsum.data(), &inner) != 0 ||
        wn_engine_warmup_max_rel(e, sum_log_step, colsum.data(), e->C, max_rel_diff_step, max_rel_diff_mass, &inner) != 0) {
      const std::string msg = inner ? inner->msg : "monitor failed";
      delete inner;
      throw std::runtime_error(msg);
    }
  });
}

int wn_engine_lanes(const wn_engine* e) { return 64 * e->geo.nw; }
int wn_engine_is_streaming(const wn_engine* e) { return e->geo.mem ? 1 : 0; }
int wn_engine_dim_padded(const wn_engine* e) { return e->Dp; }
int wn_engine_workgroups(const wn_engine* e) { return e->grid; }
int wn_engine_chain_groups(const wn_engine* e) { return e->groups; }
int wn_engine_held_tiles(const wn_engine* e) {
  return (e->geo.mem && e->hold_moving_end) ? wn::model_ops(e->model).hold_tiles(e->geo.nw) : 0;
}
int wn_engine_lds_vectors(const wn_engine* e) { return e->pool_lds; }
int64_t wn_engine_iteration(const wn_engine* e) { return e->iteration; }
void* wn_engine_stream(const wn_engine* e) { return reinterpret_cast<void*>(e->stream); }
double* wn_engine_positions_device(const wn_engine* e) { return e->theta.p; }
int wn_engine_last_kernel_ms(wn_engine* e, float* ms, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e->events_used == 0) throw std::runtime_error("no transition has been timed: call wn_engine_timing_reset first");
    e->use_device();
    auto& ev = e->events[e->events_used - 1];
    HIP_OK(hipEventSynchronize(ev.second));
    HIP_OK(hipEventElapsedTime(ms, ev.first, ev.second));
  });
}
int wn_engine_timing_reset(wn_engine* e, WalnutpyError** err) {
  return guarded(err, [&] {
    e->events_used = 0;
    e->timing = true;
  });
}
int wn_engine_kernel_times(wn_engine* e, float* ms_out, int max_launches, int* num_launches, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    // the last min(launches, ring) launches since the reset, oldest first
    const size_t have = std::min(e->events_used, wn_engine::kEventRing);
    const int n = static_cast<int>(have);
    if (num_launches) *num_launches = n;
    for (int i = 0; i < n && i < max_launches; ++i) {
      const size_t slot = (e->events_used - have + static_cast<size_t>(i)) % wn_engine::kEventRing;
      HIP_OK(hipEventSynchronize(e->events[slot].second));
      HIP_OK(hipEventElapsedTime(&ms_out[i], e->events[slot].first, e->events[slot].second));
    }
  });
}
int wn_engine_region_begin(wn_engine* e, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    if (e->region_begin == nullptr) {
      HIP_OK(hipEventCreate(&e->region_begin));
      HIP_OK(hipEventCreate(&e->region_end));
    }
    e->timing = false;  // one pair of events for the whole region instead of one pair per launch
    e->region_launches = 0;
    HIP_OK(hipEventRecord(e->region_begin, e->stream));
  });
}
int wn_engine_region_ms(wn_engine* e, float* total_ms, int* launches, WalnutpyError** err) {
  return guarded(err, [&] {
    if (e->region_begin == nullptr) throw std::runtime_error("wn_engine_region_begin has not been called");
    e->use_device();
    HIP_OK(hipEventRecord(e->region_end, e->stream));
    HIP_OK(hipEventSynchronize(e->region_end));
    HIP_OK(hipEventElapsedTime(total_ms, e->region_begin, e->region_end));
    if (launches) *launches = static_cast<int>(e->region_launches);
  });
}
int wn_engine_set_stream(wn_engine* e, void* stream, WalnutpyError** err) {
  return guarded(err, [&] {
    e->use_device();
    HIP_OK(hipStreamSynchronize(e->stream));
    if (e->own_stream && e->stream) HIP_OK(hipStreamDestroy(e->stream));
    e->stream = reinterpret_cast<hipStream_t>(stream);
    e->own_stream = false;
    // the caller orders its own work (collectives on the draws) behind the launches on THIS stream: one chain group
    e->groups = 1;
    e->gstream[0] = e->stream;
    e->group_begin[1] = e->C;
    e->group_grid[0] = static_cast<int>(std::min<size_t>(e->C, static_cast<size_t>(e->grid)));
    e->groups_ahead = false;
  });
}
int wn_engine_wait_stream(wn_engine* e, void* stream, WalnutpyError** err) {
  return guarded(err, [&] {
    HIP_OK(hipSetDevice(e->device));  // (not use_device(): the groups are not joined, they only get one more wait each)
    if (e->ext_point == nullptr) HIP_OK(hipEventCreateWithFlags(&e->ext_point, hipEventDisableTiming));
    HIP_OK(hipEventRecord(e->ext_point, reinterpret_cast<hipStream_t>(stream)));
    for (int g = 0; g < e->groups; ++g) HIP_OK(hipStreamWaitEvent(e->gstream[g], e->ext_point, 0));
  });
}
int wn_engine_wait_event(wn_engine* e, void* event, WalnutpyError** err) {
  return guarded(err, [&] {
    HIP_OK(hipSetDevice(e->device));
    for (int g = 0; g < e->groups; ++g) HIP_OK(hipStreamWaitEvent(e->gstream[g], reinterpret_cast<hipEvent_t>(event), 0));
  });
}
int wn_engine_release_stream(wn_engine* e, void* stream, WalnutpyError** err) {
  return guarded(err, [&] {
    HIP_OK(hipSetDevice(e->device));
    if (e->rel_point == nullptr) HIP_OK(hipEventCreateWithFlags(&e->rel_point, hipEventDisableTiming));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_OK(hipEventRecord(e->rel_point, e->stream));
    HIP_OK(hipStreamWaitEvent(s, e->rel_point, 0));
    // (gdone[g] was recorded behind group g's last launch; a group that has not launched yet has nothing to wait for)
    if (e->groups_ahead)
      for (int g = 1; g < e->groups; ++g) HIP_OK(hipStreamWaitEvent(s, e->gdone[g], 0));
  });
}
int wn_lanes_for_model_dim(int model, int num_params, int waves_per_chain, int elems_per_lane) {
  try {
    return 64 * wn::choose_geometry(num_params, waves_per_chain, elems_per_lane, wn::model_ops(model).uses_params,
                                    wn::model_ops(model).preferred_epl(num_params), wn::model_ops(model).hold_tiles(wn::kHeldWaves),
                                    wn::model_ops(model).register_dim_limit).nw;
  } catch (...) {
    return -1;
  }
}
int wn_lanes_for_dim(int num_params, int waves_per_chain, int elems_per_lane) {
  return wn_lanes_for_model_dim(WN_MODEL_STD_NORMAL, num_params, waves_per_chain, elems_per_lane);
}

}  // extern "C"

"""CPU tier: the host logic of walnutpie_sample_device that round 2 added -- the chunked draw sink, the SIGINT
guard (python/src/walnutpie/interrupts.hpp:34-102 -> error type `interrupt`) and the reference's three summary
symbols (walnutpy.cpp:333-369) -- under the test-only workgroup emulation."""
import os
import signal
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpusim"))
import build as simbuild  # noqa: E402
import summary_parity as sp  # noqa: E402
import walnuts_amd as wa  # noqa: E402


@pytest.fixture(scope="module")
def sim():
    return simbuild.build()


def _run(sim, **kw):
    args = dict(num_params=5, num_chains=3, seed=11, min_warmup_iter=4, max_warmup_iter=4, min_sampling_iter=7,
                max_sampling_iter=7, save_warmup=True, save_inv_metric=True, lib_path=sim, refresh=0)
    args.update(kw)
    return wa.walnuts_device(wa.MODEL_STD_NORMAL, **args)


@pytest.mark.timeout(600)
def test_chunked_draw_sink_equals_one_block(sim, monkeypatch):
    whole = _run(sim)
    # staging blocks of 2 iterations (3 chains x 5 params x 8 bytes x 2 iterations x 2 blocks), then of 1
    for budget in (3 * 5 * 8 * 2 * 2, 1):
        monkeypatch.setenv("WALNUTS_AMD_DRAW_STAGING_BYTES", str(budget))
        parts = _run(sim)
        for a, b in zip(whole, parts):
            assert np.array_equal(np.asarray(a), np.asarray(b))
            assert np.array_equal(a.warmup.warmup_draws, b.warmup.warmup_draws)
            assert a.warmup.stepsize == b.warmup.stepsize
    assert np.asarray(whole[0]).shape == (7, 5) and whole[0].warmup.warmup_draws.shape == (4, 5)


@pytest.mark.timeout(600)
def test_sigint_ends_the_call_with_keyboard_interrupt(sim):
    before = signal.getsignal(signal.SIGINT)
    seen = []

    def on_print(text):
        seen.append(text)
        if len(seen) == 2:
            os.kill(os.getpid(), signal.SIGINT)   # what Ctrl-C does while the library call is running

    with pytest.raises(KeyboardInterrupt):
        _run(sim, max_warmup_iter=50, min_warmup_iter=50, refresh=1, print_callback=on_print)
    assert 2 <= len(seen) <= 3 * 2 + 3, "the call must stop at the next iteration boundary"
    assert signal.getsignal(signal.SIGINT) is before, "the previous SIGINT disposition must be restored"
    _run(sim)  # and the library is usable afterwards


@pytest.mark.timeout(600)
def test_reference_summary_symbols(sim, oracle):
    rng = np.random.default_rng(3)
    chains = sp.ar_chains(rng, 3, 4, [23, 17, 29], np.array([0.9, 0.1, 0.5, 0.0]))
    s = wa.Summarizer(chains, lib_path=sim)
    assert np.array_equal(s.ess(), sp.wnso.effective_sample_size(chains))
    assert np.array_equal(s.r_hat(), sp.wnso.r_hat(chains))
    assert np.array_equal(s.mcse(), sp.wnso.monte_carlo_standard_error(chains))
    # MarkovChainsUnified's own validation (summary.hpp:277-281): config error
    import ctypes as C
    lib = wa.load_library(sim)
    draws = np.asfortranarray(np.concatenate(chains))
    lengths = np.array([23, 17, 30], dtype=np.intc)
    out, err = np.zeros(4), C.c_void_p()
    rc = lib.walnutpie_ess(draws.ctypes.data_as(C.POINTER(C.c_double)), draws.shape[0], 4,
                           lengths.ctypes.data_as(C.POINTER(C.c_int)), 3, out.ctypes.data_as(C.POINTER(C.c_double)),
                           C.byref(err))
    assert rc == -1 and lib.walnutpie_get_error_type(err) == 1
    assert b"sum of chain_sizes" in lib.walnutpie_get_error_message(err)
    lib.walnutpie_destroy_error(err)


@pytest.mark.timeout(600)
def test_resident_draws_and_thinned_rows_equal_the_streamed_run(sim, oracle):
    """walnutpie_sample_device_resident: the draws that stay on the device are the streamed run's, the host gets rows
    0, thin, 2 thin, ..., and the summaries over the resident block equal the oracle's over the streamed draws."""
    whole = _run(sim)
    for thin in (3, 0):
        res, chains = _run(sim, keep_on_device=True, thin=thin)
        assert chains.num_chains() == 3 and chains.dims() == 5 and chains.num_draws() == 3 * 7
        for a, b in zip(whole, res):
            want = np.asarray(a)[::thin] if thin else np.zeros((0, 5))
            assert np.array_equal(np.asarray(b), want), thin
            assert np.array_equal(a.warmup.warmup_draws, b.warmup.warmup_draws)   # warmup rows still come to the host
        draws = [np.asarray(a) for a in whole]
        assert np.array_equal(chains.mean(), sp.wnso.mean(draws))
        assert np.array_equal(chains.r_hat(), sp.wnso.r_hat(draws))
        chains.close()
    with pytest.raises(ValueError, match="thin"):
        _run(sim, keep_on_device=True, thin=-1)


@pytest.mark.timeout(600)
def test_resident_thinned_rows_follow_the_warmup_rows_actually_written(sim):
    """The warmup controller may stop before max_warmup_iter (adapt.hpp:172-229): with save_warmup the thinned sampling
    rows then follow the warmup rows that were WRITTEN (handlers.hpp:73-89), exactly where walnutpie_sample_device puts
    its sampling rows and where the wrapper slices them -- not at row max_warmup_iter."""
    kw = dict(min_warmup_iter=5, max_warmup_iter=15, step_size_converge_tol=1e6, mass_converge_tol=1e6)
    whole = _run(sim, **kw)
    n_warm = len(whole[0].warmup.warmup_draws)
    assert n_warm == 5, n_warm                                    # the loose tolerances stop warmup at its first look
    res, chains = _run(sim, keep_on_device=True, thin=2, **kw)
    for a, b in zip(whole, res):
        assert np.array_equal(a.warmup.warmup_draws, b.warmup.warmup_draws)
        assert len(b) == 4 and np.array_equal(np.asarray(b), np.asarray(a)[::2])
        assert np.any(np.asarray(b) != 0.0)
    chains.close()


@pytest.mark.timeout(600)
def test_resident_mode_argument_errors_come_back_as_config_errors(sim):
    import ctypes as C
    lib = wa.load_library(sim)
    with pytest.raises(ValueError, match="min_iter must be"):     # the reference's own validation still runs first
        _run(sim, keep_on_device=True, min_sampling_iter=9, max_sampling_iter=7)


@pytest.mark.timeout(900)
def test_multi_device_call_equals_the_single_engine_call(sim):
    """walnutpie_sample_device_multi with devices = {0, 0}: two shards (3 + 2 chains) on two host threads, engines and
    streams write the same draws, warmup draws, step sizes and inverse metrics as the one-engine call -- random streams
    are keyed by global chain id, initial positions included."""
    kw = dict(num_chains=5, save_inv_metric=True)
    whole = _run(sim, **kw)
    split = _run(sim, devices=[0, 0], **kw)
    three = _run(sim, devices=[0, 0, 0], **kw)
    for other in (split, three):
        assert len(other) == 5
        for a, b in zip(whole, other):
            assert np.array_equal(np.asarray(a), np.asarray(b))
            assert np.array_equal(a.warmup.warmup_draws, b.warmup.warmup_draws)
            assert a.warmup.stepsize == b.warmup.stepsize and np.array_equal(a.warmup.inv_metric, b.warmup.inv_metric)
    with pytest.raises(ValueError, match="fewer chains than devices"):
        _run(sim, devices=[0] * 6, **kw)
    with pytest.raises(ValueError, match="out of range"):
        _run(sim, devices=[0, 7], **kw)
    with pytest.raises(ValueError, match="min_iter must be"):      # a config error of the shards comes back as one
        _run(sim, devices=[0, 0], min_sampling_iter=9, max_sampling_iter=7, **kw)


@pytest.mark.timeout(900)
def test_multi_device_controllers_look_at_all_chains(sim):
    """Early stopping over shards: the shards' controller statistics are reduced over ALL chains, so every shard stops
    at the same iteration -- the one the single-engine call stops at (loose tolerances: first look)."""
    kw = dict(num_chains=6, min_warmup_iter=5, max_warmup_iter=20, step_size_converge_tol=1e6, mass_converge_tol=1e6,
              min_sampling_iter=4, max_sampling_iter=30, rhat_converge_tol=1e6)
    whole = _run(sim, **kw)
    split = _run(sim, devices=[0, 0], **kw)
    assert [len(a) for a in whole] == [len(b) for b in split] == [4] * 6
    assert [len(a.warmup.warmup_draws) for a in split] == [5] * 6
    for a, b in zip(whole, split):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    tight = _run(sim, devices=[0, 0], **{**kw, "rhat_converge_tol": 1.0 + 1e-12, "step_size_converge_tol": 1e-12,
                                         "mass_converge_tol": 1e-12, "max_warmup_iter": 10, "max_sampling_iter": 9})
    assert [len(a) for a in tight] == [9] * 6 and [len(a.warmup.warmup_draws) for a in tight] == [10] * 6


@pytest.mark.timeout(900)
def test_results_carry_the_stream_version(sim):
    """every result says which definition of the counter-based streams produced it (wn_stream_version)."""
    out = _run(sim, num_chains=2)
    import walnuts_amd as wa

    assert wa.stream_version(sim) == 2 and all(a.warmup.stream_version == 2 for a in out)


@pytest.mark.timeout(900)
def test_multi_device_resident_gathers_the_shards_draws(sim, oracle):
    """walnutpie_sample_device_multi_resident with devices = {0, 0} / {0, 0, 0}: every shard keeps its sampling draws
    on its device, the blocks are gathered into ONE wn_chains (peer copies) -- the thinned rows, the warmup rows and the
    on-device summaries over the gathered block equal the one-engine resident call's, early stop included."""
    kw = dict(num_chains=5, save_inv_metric=True, keep_on_device=True, thin=2, min_sampling_iter=4, max_sampling_iter=9,
              rhat_converge_tol=1e6)
    one, chains_one = _run(sim, **kw)
    assert [len(a) for a in one] == [2] * 5 and chains_one.num_draws() == 5 * 4
    for devices in ([0, 0], [0, 0, 0]):
        many, chains = _run(sim, devices=devices, **kw)
        assert chains.num_chains() == 5 and chains.num_draws() == 5 * 4
        for a, b in zip(one, many):
            assert np.array_equal(np.asarray(a), np.asarray(b))
            assert np.array_equal(a.warmup.warmup_draws, b.warmup.warmup_draws)
            assert a.warmup.stepsize == b.warmup.stepsize
        assert np.array_equal(chains.mean(), chains_one.mean())
        assert np.array_equal(chains.r_hat(), chains_one.r_hat())
        assert np.array_equal(chains.effective_sample_size(), chains_one.effective_sample_size())
        chains.close()
    # the all-gather (walnutpie_sample_device_multi_allgather): one handle per listed device, every one the whole block
    many, every = _run(sim, devices=[0, 0, 0], all_gather=True, **kw)
    assert len(every) == 3
    for a, b in zip(one, many):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    for chains in every:
        assert chains.num_chains() == 5 and chains.num_draws() == 5 * 4
        assert np.array_equal(chains.mean(), chains_one.mean())
        assert np.array_equal(chains.sample_variance(), chains_one.sample_variance())
        assert np.array_equal(chains.r_hat(), chains_one.r_hat())
        chains.close()
    with pytest.raises(ValueError, match="all_gather"):
        _run(sim, num_chains=5, all_gather=True)
    streamed = _run(sim, num_chains=5, min_sampling_iter=4, max_sampling_iter=9, rhat_converge_tol=1e6)
    assert np.array_equal(chains_one.mean(), sp.wnso.mean([np.asarray(s) for s in streamed]))
    with pytest.raises(ValueError, match="max_sampling_iter"):
        _run(sim, devices=[0, 0], num_chains=5, keep_on_device=True, thin=1, min_sampling_iter=0, max_sampling_iter=0)


@pytest.mark.timeout(900)
def test_pinned_ring_path_of_the_draw_sink(sim, monkeypatch):
    """WALNUTS_AMD_BOUNCE=1 forces the large-output path of the draw sink at test sizes: the staging blocks leave through
    a ring of pinned chunks, a dispatcher thread and scatter workers instead of one strided copy.  Same rows in the
    caller's buffer -- whole blocks, partial last blocks, one-iteration blocks, early stops, several shards -- and a
    call that ends by Ctrl-C shuts the threads down and leaves the library usable."""
    whole = _run(sim, num_chains=5)
    early_kw = dict(num_chains=5, min_warmup_iter=5, max_warmup_iter=15, step_size_converge_tol=1e6, mass_converge_tol=1e6,
                    min_sampling_iter=4, max_sampling_iter=30, rhat_converge_tol=1e6)
    early = _run(sim, **early_kw)
    monkeypatch.setenv("WALNUTS_AMD_BOUNCE", "1")
    for budget in (None, 5 * 5 * 8 * 3 * 2, 1):
        if budget is not None:
            monkeypatch.setenv("WALNUTS_AMD_DRAW_STAGING_BYTES", str(budget))
        for kw, want in ((dict(num_chains=5), whole), (early_kw, early), (dict(num_chains=5, devices=[0, 0]), whole)):
            got = _run(sim, **kw)
            for a, b in zip(want, got):
                assert np.array_equal(np.asarray(a), np.asarray(b)), (budget, kw)
                assert np.array_equal(a.warmup.warmup_draws, b.warmup.warmup_draws), (budget, kw)
    seen = []

    def on_print(text):
        seen.append(text)
        if len(seen) == 2:
            os.kill(os.getpid(), signal.SIGINT)

    with pytest.raises(KeyboardInterrupt):
        _run(sim, max_warmup_iter=50, min_warmup_iter=50, refresh=1, print_callback=on_print)
    again = _run(sim, num_chains=5)
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(whole, again))

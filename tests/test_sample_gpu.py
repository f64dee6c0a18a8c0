"""GPU tier: the boundary pieces of walnutpie_sample_device on the device -- draw sink through a small staging
buffer at the headline chain count, SIGINT -> error type `interrupt`, the reference's summary symbols."""
import os
import signal

import numpy as np
import pytest

import summary_parity as sp
import walnuts_amd as wa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu):
    return gpu


def test_sigint_ends_the_device_call_with_keyboard_interrupt():
    before = signal.getsignal(signal.SIGINT)
    seen = []

    def on_print(text):
        seen.append(text)
        if len(seen) == 8:
            os.kill(os.getpid(), signal.SIGINT)

    with pytest.raises(KeyboardInterrupt):
        wa.walnuts_device(wa.MODEL_STD_NORMAL, num_params=64, num_chains=4, seed=3, min_warmup_iter=400,
                          max_warmup_iter=400, min_sampling_iter=10, max_sampling_iter=10, refresh=1,
                          print_callback=on_print)
    assert len(seen) < 4 * 4, "the call must stop at the next iteration boundary"
    assert signal.getsignal(signal.SIGINT) is before
    out = wa.walnuts_device(wa.MODEL_STD_NORMAL, num_params=64, num_chains=4, seed=3, min_warmup_iter=5,
                            max_warmup_iter=5, min_sampling_iter=5, max_sampling_iter=5)
    assert np.all(np.isfinite(np.asarray(out[0])))


def test_sigint_while_the_pinned_ring_is_copying():
    """Ctrl-C in the middle of a run whose draws leave through the sink's pinned ring (4 GiB of output: dispatcher and
    scatter threads busy): the call ends with KeyboardInterrupt at the next launch boundary (the ring's destructor joins
    its threads before the call returns), and the same call runs to the end afterwards with the rows of a run that takes
    the direct path."""
    C, D, S = 8192, 1024, 64
    kw = dict(num_params=D, num_chains=C, seed=13, min_warmup_iter=4, max_warmup_iter=4, min_sampling_iter=S,
              max_sampling_iter=S)
    seen = []

    def on_print(text):
        seen.append(text)
        if len(seen) == 20:
            os.kill(os.getpid(), signal.SIGINT)

    with pytest.raises(KeyboardInterrupt):
        wa.walnuts_device(wa.MODEL_STD_NORMAL, refresh=1, print_callback=on_print, **kw)
    a = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    os.environ["WALNUTS_AMD_BOUNCE"] = "0"
    try:
        b = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    finally:
        del os.environ["WALNUTS_AMD_BOUNCE"]
    for c in (0, 1, 4095, 4096, C - 1):
        assert np.array_equal(np.asarray(a[c]), np.asarray(b[c])), c


def test_draw_sink_streams_the_headline_chain_count_through_a_small_staging_buffer(monkeypatch):
    """65 536 chains x 1 024 params x 16 draws = 8.6 GB of draws through two staging blocks of two iterations each
    (2.1 GB together): what the whole [C][T][D] block of a default 1 000-draw run could not do in 288 GB of HBM.
    Every row must arrive, in the caller's [C][T][D] layout, and equal what a small batch of the same chains gets."""
    C, D, T = 65536, 1024, 16
    monkeypatch.setenv("WALNUTS_AMD_DRAW_STAGING_BYTES", str(2 * 2 * C * D * 8))
    kw = dict(num_params=D, seed=21, min_warmup_iter=3, max_warmup_iter=3, min_sampling_iter=T, max_sampling_iter=T,
              init_inv_metric=np.ones(D), step_size_init=0.5)
    rng = np.random.default_rng(4)
    inits = rng.normal(size=(C, D))
    big = wa.walnuts_device(wa.MODEL_STD_NORMAL, num_chains=C, inits=inits, **kw)
    assert len(big) == C and all(b.shape == (T, D) for b in big[:8])
    for c in (0, 1, 777, C - 1):
        x = np.asarray(big[c])
        assert np.all(np.isfinite(x)) and np.any(x[1:] != x[:-1]), c   # every row arrived, and the chain moved
    # chain-id keyed streams: the first 64 chains of the big run are a 64-chain run with the same job seed ...
    # (walnutpy.cpp:82 keys the streams by seed + id + num_chains, so the small run compensates with its id)
    small = wa.walnuts_device(wa.MODEL_STD_NORMAL, num_chains=64, inits=inits[:64], id=1 + C - 64, **kw)
    for c in range(64):
        assert np.array_equal(np.asarray(big[c]), np.asarray(small[c])), c


def test_draw_copies_are_ordered_behind_every_chain_group(monkeypatch):
    """The staging blocks' copies (and the resident mode's thinned rows) leave on a second stream: they must wait for
    the launches of EVERY chain group, and a block must not be refilled by any group while it is still being copied.
    8 192 chains (two groups on 1 024 workgroups) through staging blocks of one iteration each -- a flush per iteration
    -- against the same call with one chain group; then the resident mode's thinned rows against the streamed rows."""
    C, D, T = 8192, 1024, 12
    kw = dict(num_params=D, num_chains=C, seed=5, min_warmup_iter=4, max_warmup_iter=4, min_sampling_iter=T,
              max_sampling_iter=T, save_warmup=True)
    monkeypatch.setenv("WALNUTS_AMD_DRAW_STAGING_BYTES", str(2 * C * D * 8))
    two = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    thinned, chains = wa.walnuts_device(wa.MODEL_STD_NORMAL, keep_on_device=True, thin=1, **kw)
    monkeypatch.setenv("WALNUTS_AMD_CHAIN_GROUPS", "1")
    one = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    for c in list(range(0, C, 257)) + [C // 2 - 1, C // 2, C - 1]:
        assert np.array_equal(np.asarray(two[c]), np.asarray(one[c])), c
        assert np.array_equal(two[c].warmup.warmup_draws, one[c].warmup.warmup_draws), c
        assert np.array_equal(np.asarray(thinned[c]), np.asarray(one[c])), c


def test_reference_summary_symbols_on_the_device(oracle):
    rng = np.random.default_rng(3)
    chains = sp.ar_chains(rng, 6, 130, [230, 170, 290, 201, 199, 333], rng.uniform(0, 0.95, size=130))
    s = wa.Summarizer(chains)
    assert np.array_equal(s.ess(), sp.wnso.effective_sample_size(chains))
    assert np.array_equal(s.r_hat(), sp.wnso.r_hat(chains))
    assert np.array_equal(s.mcse(), sp.wnso.monte_carlo_standard_error(chains))
    assert np.allclose(s.mean(), np.mean(np.concatenate(chains), axis=0))


def test_resident_draws_thinned_mode_on_the_device(oracle):
    """walnutpie_sample_device_resident at a size where the draw block matters (4 096 chains x 256 params x 24 draws):
    the host receives rows 0, 4, 8, ... of exactly the draws the streaming call returns, the device keeps all of them,
    and the on-device summaries over the resident block equal the oracle's over the streamed draws."""
    C, D, S = 4096, 256, 24
    kw = dict(num_params=D, num_chains=C, seed=33, min_warmup_iter=6, max_warmup_iter=6, min_sampling_iter=S,
              max_sampling_iter=S)
    whole = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    res, chains = wa.walnuts_device(wa.MODEL_STD_NORMAL, keep_on_device=True, thin=4, **kw)
    assert chains.num_chains() == C and chains.num_draws() == C * S
    for c in (0, 1, 2047, C - 1):
        assert np.array_equal(np.asarray(res[c]), np.asarray(whole[c])[::4]), c
        assert res[c].warmup.stepsize == whole[c].warmup.stepsize
    sub = [np.asarray(w) for w in whole]
    assert np.array_equal(chains.mean(), sp.wnso.mean(sub))
    assert np.array_equal(chains.r_hat(), sp.wnso.r_hat(sub))
    chains.close()
    res0, chains0 = wa.walnuts_device(wa.MODEL_STD_NORMAL, keep_on_device=True, thin=0, **kw)   # nothing to the host
    assert all(np.asarray(r).shape == (0, D) for r in res0[:4]) and chains0.num_draws() == C * S
    assert np.array_equal(chains0.mean(), sp.wnso.mean(sub))


def test_multi_device_resident_on_one_gpu_gathers_the_shards_draws():
    """walnutpie_sample_device_multi_resident with devices = {0, 0, 0} on the one GPU there is: three shards keep their
    draws in their own blocks, the blocks are gathered by hipMemcpyPeerAsync into one wn_chains -- thinned rows and
    on-device summaries equal the one-engine resident call's bit for bit."""
    C, D, S = 3000, 256, 16
    kw = dict(num_params=D, num_chains=C, seed=9, min_warmup_iter=6, max_warmup_iter=6, min_sampling_iter=S,
              max_sampling_iter=S, keep_on_device=True, thin=4)
    one, chains_one = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    many, chains = wa.walnuts_device(wa.MODEL_STD_NORMAL, devices=[0, 0, 0], **kw)
    assert chains.num_chains() == C and chains.num_draws() == C * S
    for c in (0, 999, 1000, 1999, 2000, C - 1):
        assert np.array_equal(np.asarray(many[c]), np.asarray(one[c])), c
        assert many[c].warmup.stepsize == one[c].warmup.stepsize
    assert np.array_equal(chains.mean(), chains_one.mean())
    assert np.array_equal(chains.r_hat(), chains_one.r_hat())
    assert np.array_equal(chains.quantiles([0.1, 0.5]), chains_one.quantiles([0.1, 0.5]))
    chains.close()
    # walnutpie_sample_device_multi_allgather: every listed device ends with the whole block (all-pairs copies on their
    # own streams) -- three handles, each equal to the one-engine call's
    many, every = wa.walnuts_device(wa.MODEL_STD_NORMAL, devices=[0, 0, 0], all_gather=True, **kw)
    assert len(every) == 3
    for c in (0, 999, 1000, 1999, 2000, C - 1):
        assert np.array_equal(np.asarray(many[c]), np.asarray(one[c])), c
    for ch in every:
        assert ch.num_chains() == C and ch.num_draws() == C * S
        assert np.array_equal(ch.mean(), chains_one.mean())
        assert np.array_equal(ch.quantiles([0.1, 0.5]), chains_one.quantiles([0.1, 0.5]))
        ch.close()
    chains_one.close()


def test_multi_device_call_on_one_gpu_equals_the_single_engine_call():
    """walnutpie_sample_device_multi with devices = {0, 0} (and {0, 0, 0}: uneven shards): two / three engines on their
    own host threads and streams of the one GPU there is, each writing its slice of the caller's buffers -- the same
    draws, warmup draws, step sizes and inverse metrics as the one-engine call, bit for bit; and the controllers
    (reduced over all shards) stop every shard where the one-engine call stops."""
    C, D = 1000, 200
    kw = dict(num_params=D, num_chains=C, seed=21, min_warmup_iter=10, max_warmup_iter=10, min_sampling_iter=12,
              max_sampling_iter=12, save_warmup=True, save_inv_metric=True)
    whole = wa.walnuts_device(wa.MODEL_STD_NORMAL, **kw)
    for devices in ([0, 0], [0, 0, 0]):
        split = wa.walnuts_device(wa.MODEL_STD_NORMAL, devices=devices, **kw)
        assert len(split) == C
        for c in (0, 1, 332, 333, 334, 499, 500, 666, 667, C - 1):
            assert np.array_equal(np.asarray(split[c]), np.asarray(whole[c])), (devices, c)
            assert np.array_equal(split[c].warmup.warmup_draws, whole[c].warmup.warmup_draws)
            assert split[c].warmup.stepsize == whole[c].warmup.stepsize
            assert np.array_equal(split[c].warmup.inv_metric, whole[c].warmup.inv_metric)
    early = dict(kw, min_warmup_iter=5, max_warmup_iter=40, step_size_converge_tol=1e6, mass_converge_tol=1e6,
                 min_sampling_iter=4, max_sampling_iter=30, rhat_converge_tol=1e6)
    a = wa.walnuts_device(wa.MODEL_STD_NORMAL, **early)
    b = wa.walnuts_device(wa.MODEL_STD_NORMAL, devices=[0, 0], **early)
    assert {len(x) for x in a} == {len(x) for x in b} == {4}
    assert {len(x.warmup.warmup_draws) for x in b} == {5}
    assert all(np.array_equal(np.asarray(a[c]), np.asarray(b[c])) for c in (0, 499, 500, C - 1))

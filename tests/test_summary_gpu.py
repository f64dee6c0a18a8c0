"""GPU tier: posterior summaries on the device (walnuts_amd/csrc/wn_summary.hip through the C ABI) against the
oracle, the reference's known answers, and -- at sizes the oracle cannot reach -- size-independent properties."""
import time

import numpy as np
import pytest

import summary_parity as sp
import walnuts_amd as wa
from walnuts_amd import summary as ws

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu):
    return gpu


def test_reference_known_answers_on_device():
    sp.check_reference_golden()


@pytest.mark.parametrize("C,D,lens", [
    (5, 1, [3, 4, 5, 6, 7]),                      # shortest legal chains, one column
    (7, 130, [90, 64, 77, 120, 33, 101, 64]),     # ragged, three column tiles, several lag blocks
    (64, 300, None),                              # equal lengths
    (700, 66, "ragged"),                          # more chains than one run of the two-stage sums over chains
    (300, 130, "short"),                          # <= 32 draws, even D: the one-pass two-columns-per-lane moments
])
def test_summaries_match_oracle_bitwise(C, D, lens):
    rng = np.random.default_rng(C * 1000 + D)
    if lens == "ragged":
        lens = [int(n) for n in rng.integers(20, 40, size=C)]
    if lens == "short":
        lens = [int(n) for n in rng.integers(3, 33, size=C)]
    lens = lens or [48] * C
    phi = np.where(np.arange(D) % 4 == 0, 0.9, np.where(np.arange(D) % 4 == 1, -0.5, 0.0))
    sp.check_all(sp.ar_chains(rng, C, D, lens, phi))


def test_many_chains_slabbed_lag_table_matches_oracle_bitwise():
    """More chains than one slab of the ESS's lag table (8 192): slabs, runs of 256 chains inside them and a partial
    last run give the oracle's bits; ragged lengths."""
    rng = np.random.default_rng(77)
    C, D = 16384 + 300, 3
    lens = [int(n) for n in rng.integers(6, 40, size=C)]
    phi = np.array([0.9, 0.0, -0.4])
    sp.check_all(sp.ar_chains(rng, C, D, lens, phi), probs=(0.1, 0.5), full_acov=False)


def test_many_quantiles_and_ties():
    rng = np.random.default_rng(9)
    x = rng.integers(-5, 6, size=(400, 9)).astype(float)
    sp.check_all([x[:150], x[150:]], probs=np.linspace(0, 1, 41), full_acov=False)
    z = rng.integers(0, 2, size=(30000, 3)).astype(float)      # two values, 15 000 ties each: every radix pass runs
    z[:, 2] += rng.normal(size=30000) * 1e-9                   # ... next to a column that takes the gathered finish
    sp.check_all([z[:9000], z[9000:]], probs=[0.0, 0.3, 0.5, 0.77, 1.0], full_acov=False)
    y = rng.normal(size=(999, 5)) * 10.0 ** rng.integers(-300, 300, size=5)   # every binade
    sp.check_all([y[:500], y[500:]], probs=[0.0, 1e-9, 0.5, 1 - 1e-9, 1.0], full_acov=False)


def test_view_of_the_samplers_draw_buffer():
    """Summaries straight from the [C][T][D] buffer the engine writes (no host round trip)."""
    import torch

    C, D, T = 96, 200, 40
    e = wa.DeviceEngine(wa.MODEL_DIAG_NORMAL, D, C, params=np.array([(1.0 + d % 5) ** 2 for d in range(D)]))
    e.init_positions(3, 0, 2.0)
    e.init_masses_from_grad(1e-5)
    e.set_step_sizes(1.0)
    e.adapt_step(4, 0)
    e.seed_chains(5, 0)
    for _ in range(60):
        e.warmup_step()
    e.freeze()
    draws = torch.empty((C, T + 3, D), dtype=torch.float64, device="cuda")   # stride larger than T * D
    for t in range(T):
        e.sample_step(draws.data_ptr() + t * D * 8, (T + 3) * D)
    e.synchronize()
    lengths = [T - (c % 4) for c in range(C)]                                # ragged view of the same buffer
    dev = wa.MarkovChains.from_device(draws.data_ptr(), C, T, D, chain_stride=(T + 3) * D, lengths=lengths,
                                      stream=e.stream)
    host = draws.cpu().numpy()
    chains = [host[c, :lengths[c]] for c in range(C)]
    import wnso
    assert np.array_equal(ws.mean(dev), wnso.mean(chains))
    assert np.array_equal(ws.sample_variance(dev), wnso.sample_variance(chains))
    assert np.array_equal(ws.r_hat(dev), wnso.r_hat(chains))
    assert np.array_equal(ws.effective_sample_size(dev), wnso.effective_sample_size(chains))
    assert np.array_equal(ws.quantiles(dev, [0.05, 0.5, 0.95]), wnso.quantiles(chains, [0.05, 0.5, 0.95]))
    sd = ws.sample_standard_deviation(dev)
    assert np.allclose(sd, [1.0 + d % 5 for d in range(D)], rtol=0.15)       # the chains sample the target


def test_full_size_properties():
    """16 384 chains x 1 024 dims x 48 draws (6.4 GB) of iid normals with known per-dimension location/scale."""
    import torch

    C, D, T = 16384, 1024, 48
    g = torch.Generator(device="cuda").manual_seed(11)
    scale = 1.0 + (torch.arange(D, device="cuda", dtype=torch.float64) % 5)
    loc = 0.25 * torch.arange(D, device="cuda", dtype=torch.float64)
    draws = torch.randn((C, T, D), generator=g, device="cuda", dtype=torch.float64) * scale + loc
    torch.cuda.synchronize()
    dev = wa.MarkovChains.from_device(draws.data_ptr(), C, T, D)
    N = C * T
    timings = {}

    def timed(name, f):
        t0 = time.perf_counter()
        r = f()
        timings[name] = time.perf_counter() - t0
        return r

    m = timed("mean", lambda: ws.mean(dev))
    v = timed("sample_variance", lambda: ws.sample_variance(dev))
    flat = draws.reshape(N, D)
    assert np.allclose(m, flat.mean(dim=0).cpu().numpy(), rtol=1e-12, atol=1e-12)
    assert np.allclose(v, flat.var(dim=0, unbiased=True).cpu().numpy(), rtol=1e-10)
    probs = [0.0, 0.05, 0.5, 0.95, 1.0]
    q = timed("quantiles", lambda: ws.quantiles(dev, probs))
    assert np.array_equal(q[0], flat.min(dim=0).values.cpu().numpy())
    assert np.array_equal(q[-1], flat.max(dim=0).values.cpu().numpy())
    # an order statistic has exactly `rank` draws below it: check by counting, every 64th column
    for k, p in enumerate(probs[1:-1], start=1):
        h = p * (N - 1)
        lo = int(np.floor(h))
        for d in range(0, D, 64):
            col = flat[:, d]
            below = int((col < float(q[k, d])).sum().item())
            at_or_below = int((col <= float(q[k, d])).sum().item())
            assert below <= lo + 1 and at_or_below >= lo, (p, d, below, at_or_below, lo)
    assert np.all(np.diff(q, axis=0) >= 0)
    r = timed("r_hat", lambda: ws.r_hat(dev))
    # iid chains: var(chain means) = sigma^2 / T, so the reference's statistic (summary.hpp:616-618) is sqrt(1 + 1/T)
    assert np.allclose(r, np.sqrt(1.0 + 1.0 / T), rtol=2e-3)
    ess = timed("effective_sample_size", lambda: ws.effective_sample_size(dev))
    assert np.all(ess > 0.5 * N) and np.all(ess < 2.0 * N)          # iid draws: ESS ~ N
    mcse = timed("monte_carlo_standard_error", lambda: ws.monte_carlo_standard_error(dev))
    assert np.allclose(mcse, np.sqrt(v) / np.sqrt(ess), rtol=1e-12)
    gb = C * T * D * 8 / 1e9
    print("\nsummary timings on %.1f GB of draws: " % gb + ", ".join(f"{k} {1e3 * t:.1f} ms" for k, t in timings.items()))

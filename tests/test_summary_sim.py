"""CPU tier: the device summaries (walnuts_amd/csrc/wn_summary.hip) under the workgroup emulation against the
oracle and the reference's known answers.  Small on purpose."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpusim"))
import build as simbuild  # noqa: E402
import summary_parity as sp  # noqa: E402


@pytest.fixture(scope="module")
def sim():
    return simbuild.build()


@pytest.mark.timeout(600)
def test_emulated_summaries_reference_known_answers(sim, oracle):
    sp.check_reference_golden(lib_path=sim)


@pytest.mark.timeout(600)
def test_emulated_summaries_match_oracle_bitwise(sim, oracle):
    rng = np.random.default_rng(5)
    # ragged chains, two column tiles (D > 64), lags beyond one lag block, strong and weak autocorrelation
    D = 70
    phi = np.where(np.arange(D) % 3 == 0, 0.95, 0.1)
    sp.check_all(sp.ar_chains(rng, 3, D, [41, 37, 52], phi), lib_path=sim, probs=(0.0, 0.6))
    # short chains, even D: the one-pass moments with two columns per lane (a second, nearly empty column tile)
    D = 130
    phi = np.where(np.arange(D) % 3 == 0, 0.9, -0.3)
    sp.check_all(sp.ar_chains(rng, 3, D, [32, 3, 17], phi), lib_path=sim, probs=(0.25,), full_acov=False)
    # more order statistics than one radix-select sweep holds; duplicates, negative values, zeros of both signs
    x = rng.integers(-3, 4, size=(30, 3)).astype(float)
    x[x == 0] *= np.where(rng.uniform(size=(x == 0).sum()) < 0.5, -1.0, 1.0)
    sp.check_all([x[:11], x[11:]], lib_path=sim, probs=np.linspace(0, 1, 11))


@pytest.mark.timeout(600)
def test_emulated_lag_table_in_slabs(sim, oracle):
    # more chains than one slab of the ESS's lag table (256 under the emulation): two slabs, a partial last run
    rng = np.random.default_rng(8)
    C = 310
    chains = sp.ar_chains(rng, C, 1, [int(n) for n in rng.integers(5, 9, size=C)], np.array([0.5]))
    dev = sp.wa.MarkovChains.from_host(chains, lib_path=sim)
    assert np.array_equal(sp.ws.effective_sample_size(dev), sp.wnso.effective_sample_size(chains))
    assert np.array_equal(sp.ws.r_hat(dev), sp.wnso.r_hat(chains))

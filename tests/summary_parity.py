"""Shared body of the device-vs-oracle summary tests (CPU tier under the emulation, GPU tier on the device)."""
import json
import os

import numpy as np
import pytest

import walnuts_amd as wa
import wnso
from walnuts_amd import summary as ws

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "summary_reference.json")))


def gold_chains(key):
    return [np.array(c["values"]).reshape(c["rows"], c["cols"]) for c in GOLD[key]["chains"]]


def ar_chains(rng, C, D, lens, phi):
    """ragged AR(1) chains with per-dimension autocorrelation phi[d] and per-dimension scale/offset"""
    out = []
    scale = 1.0 + np.arange(D) % 7
    for n in lens:
        x = np.zeros((n, D))
        x[0] = rng.normal(size=D)
        for t in range(1, n):
            x[t] = phi * x[t - 1] + np.sqrt(1 - phi ** 2) * rng.normal(size=D)
        out.append(x * scale + 0.3 * np.arange(D))
    return out


def check_all(chains, lib_path=None, probs=(0.0, 0.05, 0.25, 0.5, 0.6, 0.75, 0.95, 1.0), exact=True, full_acov=True):
    """every summary of the device against the oracle on the same chains; bit-exact where `exact`"""
    dev = wa.MarkovChains.from_host(chains, lib_path=lib_path)
    assert dev.num_chains() == len(chains) and dev.dims() == chains[0].shape[1]
    assert dev.num_draws() == sum(c.shape[0] for c in chains) and dev.min_chain_size() == min(c.shape[0] for c in chains)
    pairs = [("mean", ws.mean(dev), wnso.mean(chains)),
             ("sample_variance", ws.sample_variance(dev), wnso.sample_variance(chains)),
             ("sample_standard_deviation", ws.sample_standard_deviation(dev), wnso.sample_standard_deviation(chains)),
             ("quantiles", ws.quantiles(dev, probs), wnso.quantiles(chains, probs))]
    if full_acov:
        pairs.append(("autocovariance", ws.autocovariance(dev), wnso.autocovariance(chains)))
    if len(chains) >= 2 and dev.min_chain_size() >= 3:
        pairs.append(("r_hat", ws.r_hat(dev), wnso.r_hat(chains)))
    if dev.min_chain_size() >= 3:
        pairs.append(("effective_sample_size", ws.effective_sample_size(dev), wnso.effective_sample_size(chains)))
        pairs.append(("monte_carlo_standard_error", ws.monte_carlo_standard_error(dev),
                      wnso.monte_carlo_standard_error(chains)))
    for name, got, want in pairs:
        assert got.shape == want.shape, name
        if exact:
            assert np.array_equal(got, want, equal_nan=True), \
                f"{name}: max rel diff {np.nanmax(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)):.3e}"
        else:
            assert np.allclose(got, want, rtol=1e-10, atol=0, equal_nan=True), name
    dev.close()
    return dict((n, g) for n, g, _ in pairs)


def check_reference_golden(lib_path=None):
    """the device against the reference's own known answers (tests/summary_test.cpp), reference tolerances"""
    ex = wa.MarkovChains.from_host(gold_chains("example_chains"), lib_path=lib_path)
    assert np.allclose(ws.mean(ex), [8.0, 9.0], atol=1e-10) and np.allclose(ws.sample_variance(ex), [24.0, 24.0], atol=1e-10)
    for key in ("quantiles_quartiles", "quantiles_interior"):
        g = GOLD[key]
        assert np.allclose(ws.quantiles(ex, g["probs"]), np.array(g["values"]).reshape(g["shape"]), rtol=0, atol=1e-10)
    assert ws.quantiles(ex, []).shape == (0, 2)
    for bad in (-0.1, 1.1, float("nan")):
        with pytest.raises(ValueError, match=r"probs must be in \[0, 1\]"):
            ws.quantiles(ex, [0.5, bad])
    g = GOLD["quantiles_doc_example"]
    doc = wa.MarkovChains.from_host([np.array(g["column"]).reshape(-1, 1)], lib_path=lib_path)
    assert ws.quantiles(doc, g["probs"])[0, 0] == g["expected"]
    g = GOLD["autocovariance_full"]
    ac = wa.MarkovChains.from_host(gold_chains("acov_chains"), lib_path=lib_path)
    assert np.allclose(ws.autocovariance(ac), np.array(g["values"]).reshape(g["shape"]), rtol=0, atol=1e-10)
    for key in ("rhat_converged", "rhat_sqrt_ten", "rhat_ragged"):
        rc = wa.MarkovChains.from_host(gold_chains(key), lib_path=lib_path)
        assert np.allclose(ws.r_hat(rc), GOLD[key]["expected"], rtol=4 * np.finfo(float).eps, atol=0)
    ar1 = wa.MarkovChains.from_host(gold_chains("ar1_chains"), lib_path=lib_path)
    assert np.allclose(ws.effective_sample_size(ar1), GOLD["ess_three_chain"]["expected"], rtol=0, atol=1e-5)
    assert np.allclose(ws.monte_carlo_standard_error(ar1), GOLD["mcse_three_chain"]["expected"], rtol=0, atol=1e-7)
    fl = wa.MarkovChains.from_host(gold_chains("ess_floor"), lib_path=lib_path)
    assert 0 < ws.effective_sample_size(fl)[0] <= 6.0 * np.log10(6.0) + 1e-10
    # preconditions (summary.hpp:595-603,665-667)
    one = wa.MarkovChains.from_host([np.arange(6.0).reshape(3, 2)], lib_path=lib_path)
    with pytest.raises(ValueError, match="at least two chains"):
        ws.r_hat(one)
    short = wa.MarkovChains.from_host([np.arange(6.0).reshape(3, 2), np.arange(4.0).reshape(2, 2)], lib_path=lib_path)
    with pytest.raises(ValueError, match="at least 3 draws"):
        ws.r_hat(short)
    with pytest.raises(ValueError, match="at least 3 draws"):
        ws.effective_sample_size(wa.MarkovChains.from_host([np.array([[1.0], [2.0]])], lib_path=lib_path))
    with pytest.raises(ValueError, match="same number of columns"):
        wa.MarkovChains.from_host([np.zeros((2, 2)), np.zeros((2, 3))], lib_path=lib_path)
    with pytest.raises(ValueError, match="sum of chain sizes"):
        wa.MarkovChains.from_host(np.zeros((5, 2)), sizes=[2, 2], lib_path=lib_path)
    single = wa.MarkovChains.from_host([np.array([[3.0, 7.0]])], lib_path=lib_path)
    assert np.array_equal(ws.mean(single), [3.0, 7.0]) and np.all(np.isnan(ws.sample_variance(single)))
    assert np.allclose(ws.autocovariance(single), 0.0, atol=1e-10)

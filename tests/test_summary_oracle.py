"""The posterior-summary oracle (oracle/wn_summary_oracle.cpp) against the reference's own known answers
(tests/summary_test.cpp of the reference, extracted as data into tests/golden/summary_reference.json by
tests/golden/make_summary_golden.py).  Tolerances are the reference's (test_util.hpp expect_near: 1e-10 unless
the test states another)."""
import json
import os

import numpy as np
import pytest

import wnso

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "summary_reference.json")))


def chains(key):
    return [np.array(c["values"]).reshape(c["rows"], c["cols"]) for c in GOLD[key]["chains"]]


def test_mean_and_variance_hand_values():
    ex = chains("example_chains")  # columns 1,3,...,15 and 2,4,...,16 (tests/summary_test.cpp:248-345)
    assert np.allclose(wnso.mean(ex), [8.0, 9.0], rtol=0, atol=1e-10)
    assert np.allclose(wnso.sample_variance(ex), [24.0, 24.0], rtol=0, atol=1e-10)
    assert np.allclose(wnso.sample_standard_deviation(ex), np.sqrt([24.0, 24.0]), rtol=0, atol=1e-10)
    stacked = np.concatenate(ex)
    assert np.array_equal(wnso.mean(stacked, sizes=[2, 3, 3]), wnso.mean(ex))       # unified == split (:303-309)
    one = [np.array([[3.0, 7.0]])]
    assert np.array_equal(wnso.mean(one), [3.0, 7.0])                               # :266-275
    assert np.all(np.isnan(wnso.sample_variance(one)))                              # :341-353
    two = [np.array([[1.0, 5.0], [3.0, 9.0]])]
    assert np.allclose(wnso.sample_variance(two), [2.0, 8.0], atol=1e-10)           # :330-339


@pytest.mark.parametrize("key", ["quantiles_quartiles", "quantiles_interior"])
def test_quantiles_match_numpy_values_of_the_reference(key):
    g = GOLD[key]
    got = wnso.quantiles(chains("example_chains"), g["probs"])
    assert got.shape == tuple(g["shape"])
    assert np.allclose(got, np.array(g["values"]).reshape(g["shape"]), rtol=0, atol=1e-10)


def test_quantiles_doc_example_and_errors():
    g = GOLD["quantiles_doc_example"]
    got = wnso.quantiles([np.array(g["column"]).reshape(-1, 1)], g["probs"])
    assert got[0, 0] == g["expected"]                                               # EXPECT_DOUBLE_EQ, :558-568
    ex = chains("example_chains")
    for bad in (-0.1, 1.1, float("nan")):                                           # :431-487
        with pytest.raises(ValueError, match=r"probs must be in \[0, 1\]"):
            wnso.quantiles(ex, [0.5, bad])
    assert wnso.quantiles(ex, []).shape == (0, 2)                                   # :491-498
    assert np.array_equal(wnso.quantiles(ex, [0.0, 1.0]), [[1.0, 2.0], [15.0, 16.0]])


def test_autocovariance_matches_reference_table():
    g = GOLD["autocovariance_full"]
    got = wnso.autocovariance(chains("acov_chains"))
    assert got.shape == tuple(g["shape"])
    assert np.allclose(got, np.array(g["values"]).reshape(g["shape"]), rtol=0, atol=1e-10)
    assert np.allclose(wnso.autocovariance([np.array([[3.0, 7.0]])]), 0.0, atol=1e-10)   # :719-729
    assert wnso.autocovariance(chains("autocovariance_len7")).shape == (7, 2)             # :695-704
    # the reference's FFT padding rule keeps its known answers (summary.hpp:39-52)
    assert [wnso.fft_next_good_size(n) for n in (0, 1, 2, 3, 7, 11, 13, 17, 31, 97, 1000, 1001)] == \
        [2, 2, 2, 3, 8, 12, 15, 18, 32, 100, 1000, 1024]


@pytest.mark.parametrize("key", ["rhat_converged", "rhat_sqrt_ten", "rhat_ragged"])
def test_r_hat_exact_values(key):
    got = wnso.r_hat(chains(key))
    exp = np.array(GOLD[key]["expected"])
    assert np.allclose(got, exp, rtol=4 * np.finfo(float).eps, atol=0)              # EXPECT_DOUBLE_EQ = 4 ulp


def test_r_hat_errors():
    with pytest.raises(ValueError, match="at least two chains"):                    # :752-766
        wnso.r_hat([np.arange(6.0).reshape(3, 2)])
    with pytest.raises(ValueError, match="at least 3 draws"):                       # :769-808
        wnso.r_hat([np.arange(6.0).reshape(3, 2), np.arange(4.0).reshape(2, 2)])


def test_effective_sample_size_and_mcse_python_reference():
    ar1 = chains("ar1_chains")
    ess = wnso.effective_sample_size(ar1)
    assert np.allclose(ess, GOLD["ess_three_chain"]["expected"], rtol=0, atol=GOLD["ess_three_chain"]["abs_tol"])
    assert ess[0] > 5.0 * ess[1]                                                    # :1087-1097
    mcse = wnso.monte_carlo_standard_error(ar1)
    assert np.allclose(mcse, GOLD["mcse_three_chain"]["expected"], rtol=0, atol=GOLD["mcse_three_chain"]["abs_tol"])
    sd = wnso.sample_standard_deviation(ar1)
    assert np.allclose(sd, GOLD["sd_three_chain"]["expected"], rtol=0, atol=1e-6)   # data printed to 6 decimals
    assert np.allclose(mcse, sd / np.sqrt(ess), rtol=1e-15)                         # :1168-1178
    single = wnso.effective_sample_size(ar1[:1])                                    # :1060-1069
    assert single[0] > single[1] > 0
    stacked = np.concatenate(ar1)
    assert np.array_equal(wnso.effective_sample_size(stacked, sizes=[20, 20, 20]), ess)


def test_effective_sample_size_floor_and_errors():
    fl = chains("ess_floor")                                                        # :1117-1132
    ess = wnso.effective_sample_size(fl)
    assert 0 < ess[0] <= 6.0 * np.log10(6.0) + 1e-10
    with pytest.raises(ValueError, match="at least 3 draws"):                       # :1019-1035
        wnso.effective_sample_size([np.array([[1.0], [2.0]])])

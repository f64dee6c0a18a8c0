#!/usr/bin/env python3
"""Extract the known-answer DATA (inputs + expected outputs) of the reference's posterior-summary tests into
tests/golden/summary_reference.json.  Run in the build container only (it reads /root/reference):

    python tests/golden/make_summary_golden.py

Source: /root/reference/tests/summary_test.cpp.  Every `name << a, b, ...;` comma initialiser inside the named
test/helper is evaluated as arithmetic (entries such as `-64.0 / 27.0`), and scalar expectations are copied with
their tolerances.  No reference code is stored -- only numbers."""
import json
import math
import os
import re

SRC = "/root/reference/tests/summary_test.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "summary_reference.json")
text = open(SRC).read()


def body(header_regex):
    m = re.search(header_regex + r"[^{]*\{", text)
    assert m, header_regex
    i = m.end()
    depth = 1
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1], text[:m.start()].count("\n") + 1


def comma_inits(src):
    """-> {name: [values]} for every `name << v0, v1, ...;` (in order of appearance; repeated names get suffixes)"""
    out = {}
    for m in re.finditer(r"(\w+)\s*<<\s*([^;]+);", src):
        name, vals = m.group(1), m.group(2)
        if name in ("std", "out"):
            continue
        nums = [float(eval(" ".join(tok.split()), {"__builtins__": {}}, {})) for tok in re.sub(r"//[^\n]*", "", vals).split(",")]
        key = name
        k = 1
        while key in out:
            k += 1
            key = f"{name}#{k}"
        out[key] = nums
    return out


def shapes(src):
    return {m.group(1): (int(m.group(2)), int(m.group(3)))
            for m in re.finditer(r"Eigen::MatrixXd\s+(\w+)\((\d+),\s*(\d+)\)", src)}


def chains_of(fn_regex, names):
    b, line = body(fn_regex)
    vals, shp = comma_inits(b), shapes(b)
    return {"line": line, "chains": [{"rows": shp[n][0], "cols": shp[n][1], "values": vals[n]} for n in names]}


gold = {"source": "tests/summary_test.cpp of flatironinstitute/walnuts (reference @ /root/reference)"}

# ---- inputs
gold["example_chains"] = chains_of(r"make_example_chains\(\)", ["c0", "c1", "c2"])          # :16-29
gold["acov_chains"] = chains_of(r"make_acov_chains\(\)", ["ca", "cb", "cc"])                # :610-622
gold["ar1_chains"] = {"chains": []}
for i in range(3):                                                                         # :973-1013
    c = chains_of(rf"make_ar1_chain_{i}\(\)", ["c"])
    gold["ar1_chains"]["chains"].append(c["chains"][0])

# ---- expected outputs, by reference test
def expected_matrix(test_regex, name="expected"):
    b, line = body(test_regex)
    return {"line": line, "shape": list(shapes(b)[name]), "values": comma_inits(b)[name], "probs": comma_inits(b).get("probs")}


gold["quantiles_quartiles"] = expected_matrix(r"TEST\(Quantiles, QuartilesMatchNumpy\)")
gold["quantiles_interior"] = expected_matrix(r"TEST\(Quantiles, InteriorProbsMatchNumpy\)")
b, line = body(r"TEST\(Quantiles, DocExampleMatchesPseudocode\)")
gold["quantiles_doc_example"] = {"line": line, "column": comma_inits(b)["c"], "probs": comma_inits(b)["probs"],
                                 "expected": float(re.search(r"EXPECT_DOUBLE_EQ\(result\(0, 0\), ([\d.]+)\)", b).group(1))}
gold["autocovariance_full"] = expected_matrix(r"TEST\(Autocovariance, FullResultMatchesReference\)")

for key, test in (("rhat_converged", "ConvergedChainsGiveRHatOfOne"), ("rhat_sqrt_ten", "EqualWithinChainVarianceGivesSqrtTen"),
                  ("rhat_ragged", "RaggedChainsMatchExactFractionalResult")):
    b, line = body(rf"TEST\(RHat, {test}\)")
    vals, shp = comma_inits(b), shapes(b)
    names = [n for n in ("c0", "c1", "c2") if n in vals]
    exp = [float(eval(e, {"__builtins__": {}}, {"std": type("s", (), {"sqrt": staticmethod(math.sqrt)})}))
           for e in re.findall(r"EXPECT_DOUBLE_EQ\(rhat\(\d\), ([^;]+)\);", b.replace("std::sqrt", "std.sqrt"))]
    gold[key] = {"line": line, "chains": [{"rows": shp[n][0], "cols": shp[n][1], "values": vals[n]} for n in names],
                 "expected": exp}

b, line = body(r"TEST\(EffectiveSampleSize, ThreeChainMatchesPythonReference\)")
gold["ess_three_chain"] = {"line": line, "expected": [float(x) for x in re.findall(r"EXPECT_NEAR\(ess\(\d\), ([\d.]+), 1e-5\)", b)],
                           "abs_tol": 1e-5}
b, line = body(r"TEST\(MonteCarloStandardError, ThreeChainMatchesPythonReference\)")
gold["mcse_three_chain"] = {"line": line, "expected": [float(x) for x in re.findall(r"EXPECT_NEAR\(mcse\(\d\), ([\d.]+), 1e-7\)", b)],
                            "abs_tol": 1e-7}
b, line = body(r"TEST\(EffectiveSampleSize, FloorPreventsTauHatFromGoingTooSmall\)")
vals, shp = comma_inits(b), shapes(b)
gold["ess_floor"] = {"line": line, "chains": [{"rows": shp[n][0], "cols": shp[n][1], "values": vals[n]} for n in ("c0", "c1")]}
b, line = body(r"TEST\(Autocovariance, LoopFftNextGoodSize\)")
gold["autocovariance_len7"] = {"line": line, "chains": [{"rows": 7, "cols": 2, "values": comma_inits(b)["c"]}]}
# the comment block above the MCSE tests (:1137-1140) also quotes the standard deviations
m = re.search(r"SD\s*=\s*\[([\d.]+),\s*([\d.]+)\]", text)
gold["sd_three_chain"] = {"expected": [float(m.group(1)), float(m.group(2))]}

json.dump(gold, open(OUT, "w"), indent=1)
print("wrote", OUT, "with", len(gold), "entries")

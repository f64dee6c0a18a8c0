#!/usr/bin/env python3
"""Files a PMC run collected on the GPU box (profiles/pmc.sh <tag> ... -> gpurun_out/pmc_<tag>/) under profiles/:
copies the summary to profiles/$PROFILE_ROUND/pmc_<tag>.txt and enters its HBM traffic, keyed by workload AND by the hash of
walnuts_amd/csrc it was measured on, into profiles/pmc_traffic.json (bench.py refuses entries of another hash).
usage: python profiles/record_pmc.py <tag> [<tag> ...]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    entries = json.load(open(path)) if os.path.exists(path) else []
    for tag in sys.argv[1:]:
        src = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
        os.makedirs(os.path.join(ROOT, "profiles", os.environ.get("PROFILE_ROUND", "r06")), exist_ok=True)
        shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, "profiles", os.environ.get("PROFILE_ROUND", "r06"), f"pmc_{tag}.txt"))
        e = json.load(open(os.path.join(src, "traffic.json")))
        key = lambda x: (x["model"], x["chains"], x["dim"], x["phase"], x.get("transitions_per_launch", 1))
        entries = [x for x in entries if key(x) != key(e)] + [e]
        print(tag, key(e), e["csrc_sha"], f"{e['bytes_per_launch'] / 1e9:.3f} GB")
    json.dump(entries, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()

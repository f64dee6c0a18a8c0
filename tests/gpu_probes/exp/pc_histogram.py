#!/usr/bin/env python3
"""Reduce rocprofv3's PC-sampling CSV (hundreds of MB) to what travels back: the header, a few rows, and histograms by
instruction / code-object offset / stall reason.   pc_histogram.py <raw dir> <out dir>"""
import collections
import csv
import glob
import os
import sys

raw, out = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
files = [f for f in glob.glob(os.path.join(raw, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
with open(os.path.join(out, "header.txt"), "w") as h:
    for f in glob.glob(os.path.join(raw, "**", "*.csv"), recursive=True):
        h.write("== %s (%d bytes)\n" % (f, os.path.getsize(f)))
        with open(f, newline="") as fh:
            for i, line in enumerate(fh):
                if i > 6:
                    break
                h.write(line[:1500])
if not files:
    sys.exit("no pc_sampling csv under " + raw)

for f in files:
    tag = os.path.basename(f).replace(".csv", "")
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        lower = {c.lower(): c for c in cols}
        def pick(*names):
            for n in names:
                for lc, c in lower.items():
                    if n in lc:
                        return c
            return None
        c_inst = pick("instruction")
        c_comment = pick("instruction_comment", "comment")
        c_off = pick("code_object_offset", "offset", "pc")
        c_disp = pick("dispatch_id", "dispatch")
        c_stall = pick("stall_reason", "stall")
        c_issued = pick("wave_issued", "issued")
        c_type = pick("inst_type", "instruction_type")
        c_nostall = pick("no_issue", "not_issued", "arb")
        by_inst = collections.Counter()
        by_stall = collections.Counter()
        by_inst_stall = collections.defaultdict(collections.Counter)
        other = {c: collections.Counter() for c in cols if c not in (c_inst, c_comment, c_off) and "timestamp" not in c.lower()
                 and "correlation" not in c.lower() and "exec" not in c.lower()}
        n = 0
        for row in rd:
            n += 1
            key = (row.get(c_off, "") if c_off else "", row.get(c_inst, "") if c_inst else "",
                   (row.get(c_comment, "") if c_comment and c_comment != c_inst else "")[:80])
            by_inst[key] += 1
            if c_stall:
                s = (row.get(c_issued, "") if c_issued else "", row.get(c_stall, ""), row.get(c_nostall, "") if c_nostall else "")
                by_stall[s] += 1
                by_inst_stall[key][s] += 1
            for c, cnt in other.items():
                if len(cnt) < 4096:
                    cnt[row.get(c, "")] += 1
    with open(os.path.join(out, "histogram_%s.txt" % tag), "w") as o:
        o.write("# %s: %d samples; columns: %s\n" % (f, n, cols))
        for c, cnt in other.items():
            if 0 < len(cnt) <= 64:
                o.write("# %s: %s\n" % (c, dict(cnt.most_common(64))))
        if by_stall:
            o.write("\n## (issued, stall reason, arbiter) over all samples\n")
            for s, k in by_stall.most_common():
                o.write("%8d %6.2f%%  %s\n" % (k, 100.0 * k / n, s))
        o.write("\n## top instructions\n")
        for key, k in by_inst.most_common(600):
            line = "%8d %6.2f%%  %s | %s | %s" % (k, 100.0 * k / n, key[0], key[1], key[2])
            if by_inst_stall:
                line += "   " + "; ".join("%s=%d" % ("/".join(x for x in s if x), v) for s, v in by_inst_stall[key].most_common(3))
            o.write(line + "\n")
        o.write("\n## by offset (all)\n")
        def offkey(k):
            try:
                return int(k[0], 0)
            except ValueError:
                return 0
        for key in sorted(by_inst, key=offkey):
            o.write("%8d  %s | %s\n" % (by_inst[key], key[0], key[1]))
print("pc histogram written")

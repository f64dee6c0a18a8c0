import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import parity
for model, D, C, w, s in (("std_normal", 100, 4, 40, 20), ("diag_normal", 257, 6, 25, 10), ("funnel", 16, 8, 25, 10), ("std_normal", 1024, 4, 15, 10)):
    worst, growth = parity.check_reference_stream_run(model, D, C, seed=48, warmup=w, sampling=s, rtol=1e9)
    print(model, D, "growth:", " ".join("%.0e" % g for g in growth))

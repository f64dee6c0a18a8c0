"""Turns rocprofv3 rocpd databases (gpurun_out/<dir>/*_results.db) into the small text summaries committed
under profiles/.  Usage: python profiles/summarize.py <round-tag> <trace_db> [<fetch_db> <write_db>]"""
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    lines = ["name | calls | total_us | avg_us | pct"]
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{name} | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f}")
    rows = cur.execute("select name,(end-start)/1e3,grid_x,workgroup_x,lds_size,vgpr_count,sgpr_count from kernels "
                       "where name like '%transition_kernel%' order by start").fetchall()
    lines.append("")
    lines.append("transition_kernel dispatches (us, grid threads, block, LDS bytes, VGPRs, SGPRs):")
    for r in rows:
        lines.append(f"  {r[1]:.1f} us  grid={r[2]} block={r[3]} lds={r[4]} vgpr={r[5]} sgpr={r[6]}")
    if rows:
        # the timed phase's kernel = the last dispatch's; with chain groups (wn_config::chain_groups) a launch is two of
        # them on two streams.  Right after a join (freeze, a read-back) the two start aligned -- the first runs alone
        # on the whole chip, the second waits for its slots -- and drift apart over the next launches; from then on each
        # kernel lasts one launch period while sharing the chip, which is bench.py's avg_launch_ms.
        last = [r[1] for r in rows if r[0] == rows[-1][0]]
        lines.append("")
        lines.append(f"timed phase's kernel ({rows[-1][0][:60]}...): {len(last)} dispatches, mean {sum(last) / len(last):.1f} us")
        if len(last) > 8:
            steady = last[4:]
            lines.append(f"  steady state (without the first four dispatches after the join): {len(steady)} dispatches, "
                         f"mean {sum(steady) / len(steady):.1f} us, min {min(steady):.1f}, max {max(steady):.1f}")
    return lines, rows


def per_kernel(db):
    """The driver's command runs several workloads (bench.py's legs), each followed by its parity gate on a small
    engine: one block per (transition kernel, grid), in the order of first dispatch.  A leg's launches have the
    workload's grid (resident workgroups x block), its gate's a grid of 64 chains.  Within a leg the WARM instantiation
    runs the adaptation launches (and the warmup leg's timed ones), the other one the timed sampling launches; with two
    chain groups a launch is two overlapping dispatches of the same kernel, each lasting about the launch period --
    their mean over the later dispatches is what bench.py reports as that leg's avg_launch_ms."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name,(end-start)/1e3,grid_x,workgroup_x from kernels where name like '%transition_kernel%' "
                       "order by start").fetchall()
    order, by = [], {}
    for name, us, grid, block in rows:
        key = (name, grid, block)
        if key not in by:
            order.append(key)
            by[key] = []
        by[key].append(us)
    lines = ["", "per (transition kernel, grid) in order of first dispatch -- durations in us:"]
    for key in order:
        d = by[key]
        name, grid, block = key
        tail = d[len(d) // 3:] if len(d) >= 6 else d   # (past the ramp: the first dispatches after a join run alone)
        lines.append(f"  {name[:130]}  grid={grid} block={block}")
        lines.append(f"    {len(d)} dispatches; last {len(tail)}: mean {sum(tail) / len(tail):.1f}, min {min(tail):.1f}, "
                     f"max {max(tail):.1f}")
    return lines


def counter(db, kernel_like="%transition_kernel%"):
    cur = sqlite3.connect(db).cursor()
    return cur.execute("select counter_name,value,duration from counters_collection where kernel_name like ? order by start",
                       (kernel_like,)).fetchall()


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    out = [f"# rocprofv3 summary {tag}", "", "## --kernel-trace --stats", ""]
    lines, rows = kernel_stats(trace)
    out += lines
    if "--per-kernel" in sys.argv:
        sys.argv.remove("--per-kernel")
        out += per_kernel(trace)
    if len(sys.argv) >= 5:
        f, w = counter(sys.argv[3]), counter(sys.argv[4])
        out += ["", "## PMC (separate passes; KB per dispatch of transition_kernel; kernels run serialised and slower under PMC)", ""]
        for name, val, dur in f + w:
            out.append(f"  {name} = {val:.1f} KB   (dispatch {dur/1e3:.1f} us)")
        fs = [v for _, v, _ in f][-2:]
        ws = [v for _, v, _ in w][-2:]
        if fs and ws:
            fetch, write = sum(fs) / len(fs), sum(ws) / len(ws)
            out += ["",
                    f"steady-state sampling launches: FETCH_SIZE {fetch:.0f} KB, WRITE_SIZE {write:.0f} KB",
                    "gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts wide coalesced reads at 1/2 ->",
                    f"HBM traffic per launch ~= (2*FETCH_SIZE + WRITE_SIZE) * 1024 = {(2 * fetch + write) * 1024 / 1e9:.2f} GB"]
    print("\n".join(out))


if __name__ == "__main__":
    main()

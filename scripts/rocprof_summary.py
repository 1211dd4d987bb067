"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table.

    python scripts/rocprof_summary.py gpurun_out/prof_X/trace_results.db [steps] > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), "
                       f"max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {db}")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches"
          + (f"; {steps + 1} bench steps incl. 1 warm-up" if steps else ""))
    print(f"{'kernel':<64} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'share':>6}")
    for n, c, s, a, mn, mx in rows:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{short[:64]:<64} {c:>7} {s / 1e6:>10.3f} {a / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {s / total:>6.1%}")
    if "grid_x" in cols:
        # the GEMM family again, split by launch geometry (= problem shape): workgroups = grid_x / workgroup_x
        print("\n# GEMM kernel (gemm_smf16_kernel) by launch geometry (workgroups; two tiles per workgroup except in the last round)")
        print(f"{'kernel':<64} {'tiles':>7} {'calls':>7} {'avg_us':>10} {'min_us':>9} {'total_ms':>10}")
        q = (f"select {name_col}, grid_x / workgroup_x, count(*), avg(end - start), min(end - start), sum(end - start) "
             f"from kernels where {name_col} like '%gemm_smf16_kernel%' group by {name_col}, grid_x / workgroup_x order by 6 desc")
        for n, g, c, a, mn, sm in cur.execute(q).fetchall():
            short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            print(f"{short[:64]:<64} {g:>7} {c:>7} {a / 1e3:>10.2f} {mn / 1e3:>9.2f} {sm / 1e6:>10.3f}")
    gaps(cur, cols, name_col)


def gaps(cur, cols, name_col):
    """idle time between consecutive kernels of the stream the encoder runs on (the one with the most GEMM time)"""
    key = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
    if key is None:
        print(f"\n# no stream / queue column in {cols}")
        return
    best = cur.execute(f"select {key}, sum(end - start) from kernels where {name_col} like '%gemm_smf16_kernel%' group by {key} "
                       f"order by 2 desc limit 1").fetchone()
    if not best:
        return
    rows = cur.execute(f"select start, end, {name_col} from kernels where {key} = ? order by start", (best[0],)).fetchall()
    g = []
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        d = s1 - e0
        if 0 <= d < 50_000:                 # longer pauses are step boundaries / host waits, not launch gaps
            g.append(d)
    if not g:
        return
    g.sort()
    busy = sum(e - s for s, e, _ in rows)
    print(f"\n# encoder stream ({key} {best[0]}): {len(rows)} kernels, {busy / 1e6:.1f} ms busy; idle between consecutive kernels "
          f"(gaps < 50 us): n={len(g)} mean {sum(g) / len(g) / 1e3:.2f} us, median {g[len(g) // 2] / 1e3:.2f}, p90 {g[len(g) * 9 // 10] / 1e3:.2f}, "
          f"sum {sum(g) / 1e6:.2f} ms ({sum(g) / (busy + sum(g)):.1%} of the stream's span)")


if __name__ == "__main__":
    main()

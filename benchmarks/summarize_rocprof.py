"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text summary
(top kernels by total time) that can be committed under profiles/."""
import csv
import glob
import sys
from collections import defaultdict


def main(prof_dir: str, out_path: str, top: int = 40) -> None:
    stats = glob.glob(f"{prof_dir}/**/*kernel_stats.csv", recursive=True)
    lines = []
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        lines.append(f"# source: {stats[0]}")
        lines.append(f"{'calls':>8} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  name")
        for r in rows[:top]:
            lines.append(f"{int(r['Calls']):>8} {float(r['TotalDurationNs'])/1e6:>10.3f} "
                         f"{float(r['AverageNs'])/1e3:>10.2f} {float(r['MinNs'])/1e3:>9.2f} {float(r['MaxNs'])/1e3:>9.2f} "
                         f"{float(r['Percentage']):>6.2f}  {r['Name'][:150]}")
    else:
        traces = glob.glob(f"{prof_dir}/**/*kernel_trace.csv", recursive=True)
        agg = defaultdict(lambda: [0, 0.0])
        for t in traces:
            for r in csv.DictReader(open(t)):
                d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a = agg[r["Kernel_Name"]]
                a[0] += 1
                a[1] += d
        tot = sum(v[1] for v in agg.values()) or 1.0
        lines.append(f"# aggregated from {len(traces)} kernel_trace files")
        lines.append(f"{'calls':>8} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  name")
        for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            lines.append(f"{c:>8} {d/1e6:>10.3f} {d/c/1e3:>10.2f} {100*d/tot:>6.2f}  {name[:150]}")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:top + 2]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)

"""Decode-step timeline from a rocprofv3 --kernel-trace CSV: where a serving step's wall time goes on the GPU.

    python benchmarks/step_timeline.py <prof_dir> <out.txt> [anchor-kernel-substring] [steps]

Steps are delimited by consecutive launches of the anchor kernel (default: the gate_up weight stream of layer 0 is not
identifiable, so the per-step arg-max `argmax_split_kernel`); the last `steps` complete steps are averaged.  Reported per step: wall
time between anchors, GPU-busy time (union of kernel intervals), idle time, and per-kernel call counts / summed durations.
"""
import csv
import glob
import sys
from collections import defaultdict


def main(prof_dir, out_path, anchor="argmax_split_kernel", steps=100):
    traces = glob.glob(f"{prof_dir}/**/*kernel_trace.csv", recursive=True)
    rows = []
    for t in traces:
        for r in csv.DictReader(open(t)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(marks) < 3:
        open(out_path, "w").write(f"anchor {anchor!r} seen {len(marks)} times in {len(rows)} dispatches\n")
        return
    # steady state: anchors whose spacing is within 2x the median spacing, taken from the end
    gaps = [rows[marks[i + 1]][0] - rows[marks[i]][0] for i in range(len(marks) - 1)]
    med = sorted(gaps)[len(gaps) // 2]
    good = [i for i, g in enumerate(gaps) if g < 2 * med]
    good = good[-steps:]
    agg = defaultdict(lambda: [0, 0])
    wall = busy = 0
    for i in good:
        a, b = marks[i], marks[i + 1]
        seg = rows[a:b]
        wall += rows[b][0] - rows[a][0]
        cur_s, cur_e = seg[0][0], seg[0][1]
        for s, e, n in seg:
            agg[n][0] += 1
            agg[n][1] += e - s
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += min(cur_e, rows[b][0]) - cur_s
    n = len(good)
    lines = [f"# {n} steady decode steps between launches of {anchor} ({len(rows)} dispatches in the trace)",
             f"wall per step   {wall / n / 1e3:9.1f} us", f"GPU busy        {busy / n / 1e3:9.1f} us",
             f"GPU idle        {(wall - busy) / n / 1e3:9.1f} us", "",
             f"{'calls/step':>10} {'us/step':>10} {'avg_us':>9}  kernel"]
    for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{c / n:>10.2f} {d / n / 1e3:>10.1f} {d / c / 1e3:>9.2f}  {name[:140]}")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:45]))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2], a[3] if len(a) > 3 else "argmax_split_kernel", int(a[4]) if len(a) > 4 else 100)

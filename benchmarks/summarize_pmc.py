"""Condense a rocprofv3 --pmc output directory into per-kernel averages of each counter."""
import csv
import glob
import sys
from collections import defaultdict


def main(prof_dir: str, out_path: str, match: str = "") -> None:
    files = glob.glob(f"{prof_dir}/**/*counter_collection.csv", recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if match and match not in name:
                continue
            a = agg[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    lines = [f"# source: {files}", "# per-kernel AVERAGE counter value per dispatch"]
    for name, ctrs in agg.items():
        for c, (n, tot) in sorted(ctrs.items()):
            lines.append(f"{n:>6} dispatches  {c:>16} avg {tot / n:>16.1f}   {name[:110]}")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")

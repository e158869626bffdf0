"""Per-position timeline of a decode layer inside the step, from a rocprofv3 --kernel-trace CSV.

    python benchmarks/r06_step_sequence.py <prof_dir> <out.txt> [layer-start-kernel-substring]

Decode steps are delimited by the per-step arg-max launch; inside a step every launch of the layer-start kernel (default: the qkv
weight stream, `wstream_gemm_kernel<4, 5, 1`) opens a layer.  For every position of the layer's launch sequence: the kernel, its
average duration, and the average GAP between the previous kernel's end and its start (a dependent launch boundary as the
hardware timestamps see it).  Sum of durations + gaps = the layer's wall time.
"""
import csv
import glob
import sys
from collections import defaultdict


def main(prof_dir, out_path, start_sub="wstream_gemm_kernel<4, 5, 1", anchor="argmax_split_kernel"):
    rows = []
    for t in glob.glob(f"{prof_dir}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(t)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    gaps = [rows[marks[i + 1]][0] - rows[marks[i]][0] for i in range(len(marks) - 1)]
    if not gaps:
        open(out_path, "w").write("no steps found\n")
        return
    med = sorted(gaps)[len(gaps) // 2]
    good = [i for i, g in enumerate(gaps) if 0.7 * med < g < 1.3 * med][-60:]
    pos_dur, pos_gap, pos_name = defaultdict(list), defaultdict(list), {}
    layer_wall, step_wall, head_tail = [], [], []
    for i in good:
        seg = rows[marks[i]:marks[i + 1] + 1]
        starts = [j for j, r in enumerate(seg) if start_sub in r[2]]
        if len(starts) < 2:
            continue
        step_wall.append(seg[-1][0] - seg[0][0])
        head_tail.append((seg[starts[0]][0] - seg[0][0], seg[-1][0] - seg[starts[-1]][0]))
        for a, b in zip(starts[:-1], starts[1:]):
            layer_wall.append(seg[b][0] - seg[a][0])
            for k in range(a, b):
                p = k - a
                pos_name[p] = seg[k][2]
                pos_dur[p].append(seg[k][1] - seg[k][0])
                pos_gap[p].append(seg[k][0] - seg[k - 1][1])
    with open(out_path, "w") as f:
        f.write(f"# {len(good)} steady decode steps, {len(layer_wall)} layers; ns timestamps of rocprofv3 --kernel-trace\n")
        f.write(f"step wall (arg-max to arg-max) {sum(step_wall) / len(step_wall) / 1e3:9.1f} us\n")
        f.write(f"layer wall (qkv start to next qkv start) {sum(layer_wall) / len(layer_wall) / 1e3:7.2f} us\n")
        f.write(f"before the first layer {sum(h for h, _ in head_tail) / len(head_tail) / 1e3:7.1f} us, last layer start to next arg-max "
                f"{sum(t for _, t in head_tail) / len(head_tail) / 1e3:7.1f} us\n\n")
        f.write("pos   gap_us   dur_us   kernel\n")
        tg = td = 0.0
        for p in sorted(pos_name):
            g = sum(pos_gap[p]) / len(pos_gap[p]) / 1e3
            d = sum(pos_dur[p]) / len(pos_dur[p]) / 1e3
            tg += g
            td += d
            f.write(f"{p:3d}  {g:7.2f}  {d:7.2f}   {pos_name[p][:140]}\n")
        f.write(f"sum  {tg:7.2f}  {td:7.2f}\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])

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
        # what a step runs OUTSIDE its layers, in launch order (one representative step): the tail of the last layer's launches
        # (final norm / lm_head / arg-max) and everything between the arg-max and the first layer of the next step
        i = good[len(good) // 2]
        seg = rows[marks[i]:marks[i + 1] + 1]
        starts = [j for j, r in enumerate(seg) if start_sub in r[2]]
        f.write("\noutside the layers (one step, launch order): gap_us  dur_us  kernel\n")
        per = starts[1] - starts[0] if len(starts) > 1 else 9
        for k in list(range(0, starts[0])) + list(range(starts[-1] + per, len(seg))):
            g = (seg[k][0] - seg[k - 1][1]) / 1e3 if k else 0.0
            f.write(f"  {g:7.2f} {(seg[k][1] - seg[k][0]) / 1e3:7.2f}   {seg[k][2][:120]}\n")
    print(open(out_path).read())


if __name__ == "__main__" and not (len(sys.argv) > 4 and sys.argv[4] == "prefill"):
    main(*sys.argv[1:4])


def prefill_region(prof_dir, out_path):
    """Busy vs wall of the LAST step's prefill (first extend-attention launch of the step .. first fused decode launch): kernels by
    total time, idle gaps above 5 us with the kernels around them."""
    rows = []
    for t in glob.glob(f"{prof_dir}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(t)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ext = [i for i, r in enumerate(rows) if "extend_attention" in r[2]]
    dec = [i for i, r in enumerate(rows) if "wstream_combine_rope_kernel" in r[2]]
    # the last step's prefill (run bench.py with --no-kernel-roofline --no-parity --no-reference-scheduler: nothing after the job
    # launches the extend kernel): its 2 x L extend launches, from the arg-max that closed the previous step's decode
    nl = sum(1 for i in dec if i > ext[-1]) and len([i for i in ext if i > max([d for d in dec if d < ext[-1]], default=-1)])
    first_ext = ext[-nl]
    am = [i for i, r in enumerate(rows) if "argmax" in r[2] and i < first_ext]
    start = (am[-1] + 1) if am else first_ext
    end = next(i for i in dec if i > ext[-1])
    seg = rows[start:end]
    wall = seg[-1][1] - seg[0][0]
    busy, cur_e = 0, seg[0][0]
    gaps = []
    agg = defaultdict(lambda: [0, 0])
    for k, (s, e, n) in enumerate(seg):
        if s > cur_e:
            if s - cur_e > 5000:
                gaps.append((s - cur_e, seg[k - 1][2][:60], n[:60]))
            busy += e - s
        elif e > cur_e:
            busy += e - cur_e
        cur_e = max(cur_e, e)
        agg[n][0] += 1
        agg[n][1] += e - s
    with open(out_path, "w") as f:
        f.write(f"# prefill region of the last step: {len(seg)} launches\nwall {wall / 1e3:.1f} us, GPU busy {busy / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us ({100 * (wall - busy) / wall:.1f} %)\n\n")
        f.write("calls   total_us   avg_us  kernel\n")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            f.write(f"{c:5d}  {t / 1e3:9.1f}  {t / c / 1e3:7.1f}  {n[:130]}\n")
        f.write("\nidle gaps > 5 us (us, kernel before, kernel after), largest first\n")
        for g, a, b in sorted(gaps, reverse=True)[:25]:
            f.write(f"{g / 1e3:8.1f}  {a}  ->  {b}\n")
        f.write(f"gaps > 5 us: {len(gaps)}, total {sum(g for g, _, _ in gaps) / 1e3:.1f} us\n")
    print(open(out_path).read())


if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[4] == "prefill":
    prefill_region(sys.argv[1], sys.argv[2])

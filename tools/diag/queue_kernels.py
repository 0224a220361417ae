"""Diagnostic: kernel time per HIP queue and kernel name over the optimizer steps of a rocprofv3 kernel trace (which kernels make up the
main stream's critical path?).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --other-math-steps 0 --amp-steps 0
    python tools/diag/queue_kernels.py /tmp/tr [top=30]
"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("k_adam")]
    if len(adam) < 4:
        print("too few steps")
        return
    a, b = adam[1], adam[-1]
    steps = len(adam) - 2
    seg = rows[a + 1:b + 1]
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for s, e, n, q in seg:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)[:70]
        v = per[q][n]
        v[0] += 1
        v[1] += (e - s) / 1e6
    for q, ks in sorted(per.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        tot = sum(v[1] for v in ks.values())
        print(f"--- queue {q}: {tot / steps:.2f} ms of kernels per step, {sum(v[0] for v in ks.values()) / steps:.0f} launches per step")
        for n, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:top]:
            print(f"   {t / steps:7.3f} ms  {c / steps:6.1f} x {t / c * 1e3:7.1f} us  {n}")


if __name__ == "__main__":
    main()

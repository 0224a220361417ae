"""Diagnostic: per-queue busy intervals of one training step from a rocprofv3 kernel trace -- when does each stream start / finish
inside the step, how long is the tail of the weight-gradient stream after the main stream's last backward kernel?

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --other-math-steps 0 --amp-steps 0
    python tools/diag/stream_timeline.py /tmp/tr
"""
import collections
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("k_adam")]
    if len(adam) < 5:
        print("too few steps")
        return
    for a, b in list(zip(adam[2:-1], adam[3:]))[:3]:
        seg = rows[a + 1:b + 1]
        t0, t1 = rows[a][1], rows[b][1]
        print(f"--- step of {(t1 - t0) / 1e6:.2f} ms, {len(seg)} kernels")
        per = collections.defaultdict(list)
        for s, e, n, q in seg:
            per[q].append((s, e, n))
        for q, ks in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
            busy = sum(e - s for s, e, _ in ks)
            first, last = ks[0], ks[-1]
            names = collections.Counter()
            for s, e, n in ks:
                names[n.split("(")[0][:40]] += e - s
            top = ", ".join(f"{k} {v / 1e6:.2f}" for k, v in names.most_common(4))
            print(f" queue {q}: {len(ks):4d} kernels, busy {busy / 1e6:6.2f} ms, first at +{(first[0] - t0) / 1e6:6.2f} ms, last ends at +{(last[1] - t0) / 1e6:6.2f} ms "
                  f"({last[2].split('(')[0][:36]}); top: {top}")
        # 1-ms buckets: which queues are busy (fraction of the bucket)
        nb = int((t1 - t0) / 1e6) + 1
        for q, ks in sorted(per.items()):
            frac = [0.0] * nb
            for s, e, _ in ks:
                for bk in range(max(0, int((s - t0) / 1e6)), min(nb - 1, int((e - t0) / 1e6)) + 1):
                    lo, hi = t0 + bk * 1e6, t0 + (bk + 1) * 1e6
                    frac[bk] += max(0.0, min(e, hi) - max(s, lo)) / 1e6
            print(f" queue {q} busy per ms: " + " ".join(f"{min(x, 9.99):4.2f}" for x in frac))


if __name__ == "__main__":
    main()

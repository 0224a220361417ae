"""Diagnostic: from a rocprofv3 kernel trace (CSV) of bench.py, how busy is the GPU?  Union of kernel intervals, per-stream busy time
and idle gaps over the last N steps (steps are delimited by the k_adam launches).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --other-math-steps 0
    python tools/diag/trace_gaps.py /tmp/tr
"""
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
    print(f"{len(rows)} kernels, {len(adam)} optimizer steps")
    if len(adam) < 5:
        return
    # steps 3 .. n-3: between consecutive k_adam ends (skips warm-up and the isolated / other-math tail)
    for a, b in list(zip(adam[2:-1], adam[3:]))[:4]:
        seg = rows[a + 1:b + 1]
        t0, t1 = rows[a][1], rows[b][1]
        busy, cur_s, cur_e = 0, None, None
        for s, e, _, _ in seg:
            s = max(s, t0)
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += (cur_e - cur_s) if cur_e is not None else 0
        ksum = sum(e - s for s, e, _, _ in seg)
        gaps = []
        last = t0
        for s, e, n, _ in seg:
            if s > last:
                gaps.append((s - last, n))
            last = max(last, e)
        big = sum(g for g, _ in gaps if g > 20000)
        per_q = {}
        for s, e, _, q in seg:
            per_q[q] = per_q.get(q, 0) + (e - s)
        print(f"step {(t1 - t0) / 1e6:7.2f} ms: GPU busy (union) {busy / 1e6:6.2f} ms, idle {(t1 - t0 - busy) / 1e6:5.2f} ms "
              f"({len(gaps)} gaps, {big / 1e6:.2f} ms in gaps > 20 us), kernel-sum {ksum / 1e6:6.2f} ms, {len(seg)} kernels; per queue "
              + ", ".join(f"{q}: {v / 1e6:.1f}" for q, v in sorted(per_q.items(), key=lambda kv: -kv[1])[:5]))
        if os.environ.get("RD_TRACE_TOP"):
            import collections
            for q in sorted(per_q, key=per_q.get, reverse=True)[:3]:
                agg = collections.defaultdict(lambda: [0, 0])
                for s_, e_, n_, q_ in seg:
                    if q_ == q:
                        a_ = agg[n_.split("(")[0][:60]]
                        a_[0] += 1
                        a_[1] += e_ - s_
                print(f"     queue {q}:", [(k, c, round(t / 1e6, 2)) for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]])
        worst = sorted(gaps, reverse=True)[:6]
        print("     largest gaps (us, next kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in worst])


if __name__ == "__main__":
    main()

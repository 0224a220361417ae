"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/*_pmc_hbm_traffic.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-math-steps 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-math-steps 0
    python tools/diag/pmc_to_json.py /tmp/pmc_f /tmp/pmc_w out.json

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are collected in separate passes, both are reported in KB, and on
gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads (doubled here); WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        f_kb = sf / nf if nf else 0.0
        w_kb = sw / nw if nw else 0.0
        kernels[k] = {"dispatches": int(max(nf, nw)), "FETCH_SIZE_avg_KB_raw": round(f_kb, 2), "WRITE_SIZE_avg_KB": round(w_kb, 2),
                      "hbm_bytes_per_launch_corrected": int(2 * f_kb * 1024 + w_kb * 1024)}
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 3 --warmup 1 "
            "--no-cpu-baseline --other-math-steps 0`, averaged per dispatch. gfx950 correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE "
            "reports 1/2 of wide coalesced reads -> doubled; WRITE_SIZE exact. Infinity-Cache hits are included in FETCH_SIZE (fabric-side counter).")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from radardistill_amd import native
    json.dump({"note": note, "csrc_sha": native.csrc_sha(), "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:12]:
        print(v["dispatches"], v["hbm_bytes_per_launch_corrected"], k[:100])


if __name__ == "__main__":
    main()

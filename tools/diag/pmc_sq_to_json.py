"""Turn one rocprofv3 SQ counter pass over bench.py into profiles/*_pmc_mfma_busy_*.json (MFMA-busy evidence per kernel).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \\
              SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/sq -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-math-steps 0
    python tools/diag/pmc_sq_to_json.py /tmp/sq out.json

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the SIMDs (32 per
v_mfma_f32_32x32x16_bf16); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave and are disjoint buckets.
mfma_busy_frac = MFMA busy cycles / (kernel duration x 2.4 GHz x 1024 SIMDs): the share of the chip's matrix-pipe time at the spec
clock that the kernel used (dispatches are serialised by the counter collection, so this is the isolated figure).
"""
import collections
import csv
import glob
import json
import os
import sys

CLK_GHZ, SIMDS = 2.4, 1024


def main():
    d, out = sys.argv[1:3]
    cf = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    if tf:
        for r in csv.DictReader(open(tf[0])):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc = collections.defaultdict(lambda: {"n": set(), "c": collections.defaultdict(float), "ns": 0.0})
    seen = set()
    for r in csv.DictReader(open(cf)):
        k = r["Kernel_Name"]
        a = acc[k]
        a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r["Dispatch_Id"]
        if (k, did) not in seen:
            seen.add((k, did))
            a["n"].add(did)
            a["ns"] += dur.get(did, 0)
    kernels = {}
    for k, a in sorted(acc.items(), key=lambda kv: -kv[1]["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        n = len(a["n"])
        c = {x: v / n for x, v in a["c"].items()}
        mf = c.get("SQ_INSTS_MFMA", 0.0)
        if mf <= 0:
            continue
        ns = a["ns"] / n if a["ns"] else None
        wave = c.get("SQ_WAVE_CYCLES", 0.0)
        kernels[k] = {
            "dispatches": n, "avg_duration_us": None if ns is None else round(ns / 1e3, 2),
            "mfma_insts": round(mf), "mfma_busy_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)),
            "mfma_busy_frac_at_2p4GHz": None if not ns else round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (ns * CLK_GHZ * SIMDS), 4),
            "valu_insts_per_mfma": round((c.get("SQ_INSTS_VALU", 0.0) - mf) / mf, 2),
            "wave_time_share": None if wave <= 0 else {"parked_waitcnt_or_barrier": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 3),
                                                        "issue_stall": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
                                                        "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3)},
        }
    json.dump({"note": __doc__.split("Units", 1)[1].strip().replace("\n", " "), "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:12]:
        print(k[:70], v)


if __name__ == "__main__":
    main()

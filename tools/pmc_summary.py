#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, as MI355X_MICROARCH.md prescribes).
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <launches to skip (warm-up step)> > json
Units: FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads
(TCC_EA0_RDREQ x 64 B with 128-B requests tallied at 64 B), so the read side is doubled; WRITE_SIZE is exact."""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].split("(")[0].replace("cjs::", "").replace("void ", "")
        a = acc.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
out = {"steps_in_profile": steps, "note": "per step; bytes = KiB*1024; fetch doubled per the gfx950 FETCH_SIZE correction", "kernels": {}}
tot = 0.0
for k in sorted(set(f) | set(w)):
    fb = 2.0 * f.get(k, [0, 0.0])[1] * 1024 / steps
    wb = w.get(k, [0, 0.0])[1] * 1024 / steps
    calls = max(f.get(k, [0, 0])[0], w.get(k, [0, 0])[0]) / steps
    out["kernels"][k] = {"launches_per_step": calls, "fetch_bytes_per_step": round(fb), "write_bytes_per_step": round(wb),
                         "hbm_bytes_per_launch": round((fb + wb) / calls) if calls else 0}
    tot += fb + wb                      # every kernel of the step (hs_*, scan_u32_single, dev_fill_kernel, shard_meta_kernel ... included)
out["pipeline_hbm_bytes_per_step"] = round(tot)
out["input_bytes"] = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000_000
sc = [v for k, v in out["kernels"].items() if k.startswith("rs_scatter<unsigned long")]     # all template variants of the 64-bit scatter
if sc:
    calls = sum(v["launches_per_step"] for v in sc)
    out["rs_scatter_hbm_bytes_per_launch"] = round(sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in sc) / calls)
    out["rs_scatter_launches_per_step"] = calls
# full-size launches only (one pass over all suffixes of the step; the large-group path of round 2 makes a few tiny ones): per
# launch the counter files hold one row each, so take the rows above half of the largest value
def full_rows(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "rs_scatter<unsigned long" in r["Kernel_Name"]]
    big = [v for v in vals if vals and v > 0.5 * max(vals)]
    return (sum(big) / len(big) if big else 0.0), len(big)
ff, nf = full_rows(sys.argv[1], "FETCH_SIZE")
fw, nw = full_rows(sys.argv[2], "WRITE_SIZE")
if nf and nw:
    out["rs_scatter_full_hbm_bytes_per_launch"] = round((2.0 * ff + fw) * 1024)
    out["rs_scatter_full_launches_per_step"] = nf / steps
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""profiles/calib_r02.json from the counter-calibration run (scripts/ubench/gather_calib under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, a one-off run script of round 2, in the git history): known bytes of five access patterns
against what the counters report (KiB units), and the correction factors derived from them."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = ROOT / "gpurun_out"
req = {}
for line in open(src / "calib_plain.jsonl"):
    if line.startswith("{"):
        d = json.loads(line)
        req[d["kernel"]] = d
ctr = defaultdict(dict)
for sub, name in (("calib_fetch/fetch", "FETCH_SIZE"), ("calib_write/write", "WRITE_SIZE")):
    acc = defaultdict(list)
    for r in csv.DictReader(open(src / f"{sub}_counter_collection.csv")):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        ctr[k][name + "_KiB"] = sum(v) / len(v)
out = {"_how": "scripts/ubench/gather_calib.hip on one MI355X; counters in KiB per launch",
       "kernels": {}}
for k, d in req.items():
    e = dict(requested_bytes=d["requested_bytes"], ms=d["ms"], requested_GBs=d["GBs"], note=d["note"])
    e.update(ctr.get(k, {}))
    if "FETCH_SIZE_KiB" in e:
        e["fetch_bytes_raw"] = e["FETCH_SIZE_KiB"] * 1024
    if "WRITE_SIZE_KiB" in e:
        e["write_bytes_raw"] = e["WRITE_SIZE_KiB"] * 1024
    out["kernels"][k] = e
K = out["kernels"]
n_gather = 16 * 1024 * 1024
idx_bytes = n_gather * 4
rec_fetch = K["k_gather48_random"]["fetch_bytes_raw"] - idx_bytes / 2   # index stream: coalesced, tallied at 1/2
out["findings"] = {
    "coalesced_16B_read_factor": K["k_stream_read16"]["requested_bytes"] / K["k_stream_read16"]["fetch_bytes_raw"],
    "coalesced_16B_write_factor": K["k_stream_write16"]["requested_bytes"] / K["k_stream_write16"]["write_bytes_raw"],
    "gather48": {
        "requested_bytes_per_gather": 48,
        "line_bytes_per_gather_model": 96,     # a 16-B-aligned 48-B record spans 1.5 64-B lines on average
        "counter_bytes_per_gather": rec_fetch / n_gather,
        "model": "FETCH_SIZE tallies one 64-B unit per request; a request is 64 B, or 128 B when both halves "
                 "of a 128-B-aligned line are needed: (4 x 64 + 2 x 64[=one 128-B request] + 2 x 128) / 8 = 80 B "
                 "per gather counted, 96 B moved",
        "counter_to_moved_factor": 96.0 / (rec_fetch / n_gather),
        "moved_to_requested": 2.0,
    },
    "atomic_records": {
        "write_bytes_per_record_update": K["k_atomic_records"]["write_bytes_raw"] / (14 * 1024 * 1024),
        "fetch_bytes_per_record_update": (K["k_atomic_records"]["fetch_bytes_raw"] - 14 * 1024 * 1024 * 4 / 2)
        / (14 * 1024 * 1024),
        "note": "an L2-missing float atomic on a 64-B record is tallied as ONE 64-B write, hardly any fetch: "
                "the atomic is forwarded to the memory side",
    },
}
(ROOT / "profiles" / "calib_r02.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
print(json.dumps(out["findings"], indent=1))

#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs that scripts/profile.sh leaves under gpurun_out/prof_<tag>/ into the
small, tracked summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table (gs::* + sort kernels)
  profiles/<tag>_pmc.json           per-kernel, per-launch means of every PMC counter collected
  profiles/<tag>_summary.md         the table DESIGN.md / bench.py's roofline object refer to

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]": FETCH_SIZE and
WRITE_SIZE come from separate --pmc passes; rocprofv3 reports them in KiB; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B for wide coalesced reads, so it is doubled ("corrected");
WRITE_SIZE is reported as is (uncalibrated per the guide).
"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
GATHER_FACTOR = 1.18            # profiles/calib_r02.json: counter_to_moved_factor of the 48-B gather
GATHER_KERNELS = ("k_rasterize_forward", "k_rasterize_backward")


def read_factor(kernel_name: str) -> float:
    return GATHER_FACTOR if any(g in kernel_name for g in GATHER_KERNELS) else 2.0


def short(name: str) -> str:
    name = name.strip('"')
    m = re.search(r"gs::(k_\w+)(<[^>(]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    if "radix_sort_onesweep" in name:
        if "onesweep_histograms" in name or "histogram" in name:
            return "rocprim::radix_onesweep_histograms"
        return "rocprim::radix_onesweep_" + str(abs(hash(name)) % 1000)
    m = re.search(r"rocprim::\w+::detail::(\w+)", name)
    if m:
        return "rocprim::" + m.group(1)
    return name[:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)

    # 1. kernel stats
    stats = []
    with open(src / "trace" / "trace_kernel_stats.csv") as f:
        for row in csv.DictReader(f):
            stats.append(row)
    with open(dst / f"{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in stats:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])

    # 2. PMC passes
    pmc = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [values]
    for sub in ("fetch", "write", "sq", "sq2", "sq3", "tcc"):
        p = src / sub / f"{sub}_counter_collection.csv"
        if not p.exists():
            continue
        with open(p) as f:
            for row in csv.DictReader(f):
                pmc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, ctrs in pmc.items():
        if "gs::" not in k and "rocprim" not in k:
            continue
        d = {c: sum(v) / len(v) for c, v in ctrs.items()}
        d["launches_sampled"] = max(len(v) for v in ctrs.values())
        if "FETCH_SIZE" in d:
            d["hbm_read_factor"] = read_factor(k)
            d["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * read_factor(k)
            d["hbm_read_bytes_x2_upper_bound"] = d["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
        out[k[:200]] = d
    with open(dst / f"{tag}_pmc.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)

    # 3. markdown
    avg = {r["Name"]: float(r["AverageNs"]) for r in stats}
    calls = {r["Name"]: int(r["Calls"]) for r in stats}
    lines = [f"# rocprofv3 summary, tag {tag}", "",
             "Source: `scripts/profile.sh` on one MI355X (gpurun), summarised by "
             "`scripts/summarize_profile.py`.", "",
             "| kernel | calls | avg µs | % | HBM read MB (FETCH_SIZE × 2, × 1.18 for the gather kernels) | HBM write MB | VALU insts/wave-avg | "
             "VALU busy (ACTIVE_INST_VALU/BUSY_CYCLES) | LDS insts | wait_any/wave_cycles |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    pct = {r["Name"]: r["Percentage"] for r in stats}
    for name in sorted(avg, key=lambda n: -avg[n] * calls[n]):
        if "gs::" not in name and "rocprim" not in name:
            continue
        d = out.get(name[:200], {})
        rd = d.get("hbm_read_bytes_corrected")
        wr = d.get("hbm_write_bytes")
        valu = d.get("SQ_INSTS_VALU")
        busy = None
        if d.get("SQ_BUSY_CYCLES"):
            busy = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_BUSY_CYCLES"]
        wait = None
        if d.get("SQ_WAVE_CYCLES"):
            wait = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
        fmt = lambda v, s="%.1f": "–" if v is None else s % v
        lines.append("| `%s` | %d | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (
            short(name), calls[name], avg[name] / 1e3, pct[name],
            fmt(None if rd is None else rd / 1e6), fmt(None if wr is None else wr / 1e6),
            fmt(valu, "%.3g"), fmt(busy, "%.3f"), fmt(d.get("SQ_INSTS_LDS"), "%.3g"),
            fmt(wait, "%.3f")))
    (dst / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
    # per-launch HBM traffic (bytes) of the gs:: kernels, for bench.py's roofline.traffic
    traffic = {}
    for name, d in out.items():
        m = re.search(r"gs::(k_\w+)", name)
        if m and "hbm_read_bytes_corrected" in d and "hbm_write_bytes" in d:
            traffic[m.group(1)] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    traffic["_source"] = (f"profiles/{tag}_pmc.json (FETCH_SIZE KiB x 2 for streaming kernels, x {GATHER_FACTOR} for "
                          "k_rasterize_* per profiles/calib_r02.json, + WRITE_SIZE)")
    # the row-f2 kernels (scripts/profile_f2.sh, tags f2*) keep their own file
    # ... and so do profiles of the non-default configurations (tags like r01i_c3)
    tname = "traffic_f2.json" if tag.startswith("f2") else \
        ("traffic_%s.json" % tag.split("_", 1)[1] if "_" in tag else "traffic.json")
    if len(traffic) > 1:   # (a kernel-stats-only profile has no counters: keep the last measured traffic file)
        (dst / tname).write_text(json.dumps(traffic, indent=1, sort_keys=True) + "\n")
    # every gs:: kernel of the step in one small file that bench.py attaches to its JSON line
    # (`kernels_profiled`, `roofline_valu`): rocprofv3 average duration, measured HBM bytes, the VALU
    # issue RATE (SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES; the key still says "of_8", but there is no fixed
    # ceiling: a saturated plain-fp32 loop reads 13.2, DPP / compare / SGPR-operand / fp64 loops 7.5 —
    # profiles/valu_calib_r06.json; scripts/issue_roofline.py turns it into a fraction of the kernel's own
    # saturated mix) and — from the instrumented build's work counters
    # (scripts/work_stats.py -> profiles/work_stats_<tag>_<cfg>.json) — the fraction of lanes that do
    # needed work in a compositing step.
    kern = {}
    for name in avg:
        m = re.search(r"gs::(k_\w+)(<[^>(]*>)?", name)
        if not m:
            continue
        key = m.group(1) + (m.group(2) or "")
        d = out.get(name[:200], {})
        e = {"calls": calls[name], "avg_us": avg[name] / 1e3}
        if "hbm_read_bytes_corrected" in d:
            e["hbm_read_bytes"] = d["hbm_read_bytes_corrected"]
        if "hbm_write_bytes" in d:
            e["hbm_write_bytes"] = d["hbm_write_bytes"]
        if d.get("SQ_BUSY_CYCLES"):
            e["valu_busy_of_8"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_BUSY_CYCLES"]
            e["valu_insts_per_launch"] = d.get("SQ_INSTS_VALU")
            if d.get("SQ_INSTS_VALU"):
                # the CU's one scalar unit serves four SIMDs: a kernel near one scalar instruction per
                # vector instruction is co-bound by it (the forward's walk)
                e["salu_per_valu"] = d.get("SQ_INSTS_SALU", 0) / d["SQ_INSTS_VALU"]
        if d.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
        kern[key] = e
    cfg = tag.split("_", 1)[1] if "_" in tag else "c2"
    ws = dst / ("work_stats_%s_%s.json" % (tag.split("_")[0], cfg))
    if ws.exists():
        w = json.loads(ws.read_text())
        for k, side in (("k_rasterize_forward", "forward"), ("k_rasterize_backward", "backward")):
            for key in kern:
                if key.startswith(k):
                    kern[key]["live_lane_frac"] = w[side]["live_lanes_per_needing_step"] / 64.0
                    kern[key]["steps_per_list_entry"] = w[side]["steps_per_list_entry"]
                    kern[key]["block_entries_per_list_entry"] = w[side]["block_entries"] / max(w["M"], 1)
    kname = "kernels.json" if "_" not in tag else "kernels_%s.json" % cfg
    (dst / kname).write_text(json.dumps({"tag": tag, "kernels": kern}, indent=1, sort_keys=True) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

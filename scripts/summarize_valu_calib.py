#!/usr/bin/env python3
"""gpurun_out/valu_calib_<tag>/ (scripts/gpu_valu_calib.sh) -> profiles/valu_calib_<tag>.json:
per instruction class of scripts/ubench/valu_rate.hip, at 4 / 5 / 8 waves per SIMD: cycles per wave-instruction per
SIMD at the nominal 2.4 GHz (the ubench's own HIP-event timing), and — from the rocprofv3 PMC pass of the same
binary — what the counters of scripts/profile.sh read for that SATURATED loop:
    valu_busy_of = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES     (the ratio bench.py called "of 8" until round 5)
    active_per_inst = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU   (counter units charged per issued wave-instruction)
"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
KINDS = {0: "v_mul_f32", 1: "v_pk_mul_f32", 9: "v_pk_fma_f32", 2: "v_fma_f64", 3: "v_mul_f64", 10: "v_add_f64",
         7: "v_cvt_f64_f32", 8: "v_cvt_f32_f64", 4: "v_exp_f32", 5: "v_rcp_f32", 6: "v_cndmask_b32",
         11: "v_cmp_le_f32", 12: "v_med3_f32", 13: "v_permlane32_swap", 14: "v_add_f32_dpp",
         15: "v_cndmask_e64 sgpr", 16: "v_cndmask indep", 17: "v_mov_b32", 18: "v_add_u32", 19: "v_and_b32",
         20: "v_add_f32", 21: "v_fma_f32", 25: "cmp+cndmask vcc (2)", 26: "cmp+cndmask sgpr(2)",
         27: "cmp,nop3,cndmask vcc", 22: "v_cmp_e64 ->sgpr", 23: "v_max_f32", 24: "v_mul_f32 sgpr src",
         100: "mix forward", 101: "mix backward"}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    src = ROOT / "gpurun_out" / f"valu_calib_{tag}"
    out = {"source": "scripts/gpu_valu_calib.sh on one MI355X", "classes": {}}
    rate = defaultdict(dict)
    for line in (src / "valu_rate.txt").read_text().splitlines():
        m = re.match(r"(.+?)\s+(\d+) waves/SIMD:\s+([\d.]+) us\s+->\s+([\d.]+) cycles", line)
        if m:
            rate[m.group(1).strip()]["w%s" % m.group(2)] = {"us": float(m.group(3)), "cycles_per_inst": float(m.group(4))}
    for name, r in rate.items():
        out["classes"][name] = {"cycles_per_wave_instruction_at_2p4GHz": r}
    for w in (4, 5):
        files = list((src / f"pmc_w{w}").rglob("*counter_collection.csv"))
        if not files:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(files[0])):
            m = re.search(r"k<(\d+)>", row["Kernel_Name"])
            if m:
                acc[int(m.group(1))][row["Counter_Name"]].append(float(row["Counter_Value"]))
            m = re.search(r"kmix<(\d+)>", row["Kernel_Name"])
            if m:
                acc[100 + int(m.group(1))][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for kind, ctr in acc.items():
            name = KINDS.get(kind, "kind %d" % kind)
            # (two launches per class: the warm-up and the timed one — the means)
            c = {k: sum(v) / len(v) for k, v in ctr.items()}
            e = out["classes"].setdefault(name, {})
            d = {"counters": c}
            if c.get("SQ_BUSY_CYCLES"):
                d["valu_busy_of"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_BUSY_CYCLES"]
            if c.get("SQ_INSTS_VALU"):
                d["active_per_inst"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_INSTS_VALU"]
            if c.get("SQ_WAVE_CYCLES"):
                d["active_valu_over_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
            e["pmc_w%d" % w] = d
    dst = ROOT / "gpurun_out" / f"valu_calib_{tag}.json"
    dst.write_text(json.dumps(out, indent=1, sort_keys=True))
    for name, e in sorted(out["classes"].items()):
        r = e.get("cycles_per_wave_instruction_at_2p4GHz", {})
        p4, p5 = e.get("pmc_w4", {}), e.get("pmc_w5", {})
        print("%-22s cyc w4 %5s w5 %5s w8 %5s | busy_of w4 %6.2f w5 %6.2f | act/inst %5.2f" % (
            name, r.get("w4", {}).get("cycles_per_inst", "-"), r.get("w5", {}).get("cycles_per_inst", "-"),
            r.get("w8", {}).get("cycles_per_inst", "-"), p4.get("valu_busy_of", float("nan")),
            p5.get("valu_busy_of", float("nan")), p4.get("active_per_inst", float("nan"))))


if __name__ == "__main__":
    main()

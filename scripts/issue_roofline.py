#!/usr/bin/env python3
"""an ISSUE roofline for the two compositing kernels that can be checked.

    issue_roofline = sum over instruction classes c of (dynamic wave-instructions of class c) x (measured cycles per
                     wave-instruction of class c per SIMD) / (1024 SIMDs x 2.4 GHz) / (kernel time)

  * cycles per class: scripts/ubench/valu_rate.hip at 4 waves per SIMD (profiles/valu_calib_r06.json), in cycles of
    the nominal 2.4 GHz — i.e. they are TIMES, whatever the clock really was;
  * dynamic instructions: the kernel's SQ_INSTS_VALU per launch (rocprofv3 PMC pass, profiles/kernels*.json) split by
    the STATIC class mix of the kernel's loop bodies — every instruction between a label and a later backward branch
    to it, from `hipcc -S` of gs_raster.hip (prologue / epilogue code, executed once per wave, is left out of the mix);
  * kernel time: rocprofv3 --kernel-trace average of the same profile.
Also, from the same calibration: what SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES would READ if a loop of this mix were
issue-saturated (the harmonic combination of the per-class saturated readings: 13.2 plain, 7.5 DPP / compare / SGPR
operand / fp64 / packed, 7.8 transcendental) — the ceiling of the ratio bench.py printed as "of 8" until round 5.

    python scripts/issue_roofline.py [kernels.json ...]  ->  profiles/issue_roofline_r06.json
"""
import json
import os
import re
import subprocess
import sys
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "opensplat_amd" / "csrc"
KERNELS = {   # short name in profiles/kernels*.json -> (mangled-name fragment, the ubench's mixed loop of its class mix)
    "k_rasterize_forward<true, 1, false>": ("k_rasterize_forwardILb1ELi1ELb0EE", "mix forward"),
    "k_rasterize_backward_q<true, false>": ("k_rasterize_backward_qILb1ELb0EE", "mix backward"),
}
# class -> the valu_rate.hip loop that prices it
PRICE = {"plain": "v_fma_f32", "dpp": "v_add_f32_dpp", "cmp": "v_cmp_le_f32", "cndmask": "v_cndmask_e64 sgpr",
         "sgpr_src": "v_mul_f32 sgpr src", "fp64": "v_fma_f64", "packed": "v_pk_fma_f32", "trans": "v_exp_f32",
         "lane": "v_add_f32_dpp"}
SIMDS, CLOCK = 1024, 2.4e9


def classify(op, operands):
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if "_f64" in op:
        return "fp64"
    if op.startswith("v_pk_"):
        return "packed"
    if "dpp" in op or "row_" in operands or "quad_perm" in operands or op.startswith("v_permlane"):
        return "dpp"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "cndmask"
    # a scalar register or a literal among the sources (VOP3 / literal encodings issue at the slower rate)
    srcs = operands.split(",")[1:]
    if any(re.match(r"\s*(s\d+|s\[|vcc|exec|0x[0-9a-f]{3,}|-?\d{3,})", s) for s in srcs):
        return "sgpr_src"
    return "plain"


def loop_mix(asm, fragment):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2gs") and fragment in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    in_loop = [False] * len(body)
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            for k in range(label_at[m.group(1)], i + 1):
                in_loop[k] = True
    mix, total_valu, loop_valu = Counter(), 0, 0
    for i, l in enumerate(body):
        m = re.match(r"\s+(v_\w+)\s*(.*?)(;.*)?$", l)
        if not m:
            continue
        total_valu += 1
        if in_loop[i]:
            loop_valu += 1
            mix[classify(m.group(1), m.group(2))] += 1
    return mix, loop_valu, total_valu


def main():
    calib = json.load(open(ROOT / "profiles" / "valu_calib_r06.json"))["classes"]
    price = {c: calib[k]["cycles_per_wave_instruction_at_2p4GHz"]["w4"]["cycles_per_inst"] for c, k in PRICE.items()}
    sat = {c: calib[k]["pmc_w4"]["valu_busy_of"] / calib[k]["pmc_w4"].get("active_per_inst", 1.0)
           for c, k in PRICE.items()}     # saturated reading per issued instruction of the class
    active = {c: calib[k]["pmc_w4"].get("active_per_inst", 1.0) for c, k in PRICE.items()}
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    asm_path = "/tmp/gs_raster_issue.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-fno-slp-vectorize", "-Wno-unused-function", "-S", "--cuda-device-only",
                           str(CSRC / "gs_raster.hip"), "-o", asm_path], stderr=subprocess.DEVNULL)
    asm = open(asm_path).read()
    files = sys.argv[1:] or ["kernels.json", "kernels_c3.json"]
    out = {"price_cycles_at_2p4GHz_w4": price, "saturated_reading_per_class": {c: sat[c] * active[c] for c in sat},
           "source": "scripts/issue_roofline.py: class mix of the loop bodies from hipcc -S, SQ_INSTS_VALU and kernel "
                     "time from the committed rocprofv3 passes, class prices from profiles/valu_calib_r06.json",
           "profiles": {}}
    for f in files:
        prof = json.load(open(ROOT / "profiles" / f))["kernels"]
        rows = {}
        for short, (frag, mixname) in KERNELS.items():
            e = prof.get(short)
            if e is None:
                continue
            mix, loop_valu, total_valu = loop_mix(asm, frag)
            n = sum(mix.values())
            frac = {c: mix[c] / n for c in mix}
            insts = e["valu_insts_per_launch"]
            # additive model: the classes' own prices weighted by the mix (over-estimates an interleaved stream: the
            # slower classes' extra cycles are partly filled by other waves' instructions) ...
            additive = sum(frac[c] * price[c] for c in frac)
            act = sum(frac[c] * active[c] for c in frac)
            additive_reading = act / sum(frac[c] * active[c] / (sat[c] * active[c]) for c in frac)
            # ... so the price used is the MEASURED one of a saturated loop interleaving the classes in this mix
            mixed = calib[mixname]
            cyc_per_inst = mixed["cycles_per_wave_instruction_at_2p4GHz"]["w4"]["cycles_per_inst"]
            sat_reading = mixed["pmc_w4"]["valu_busy_of"]
            issue_s = insts * cyc_per_inst / (SIMDS * CLOCK)
            rows[short] = {
                "additive_model": {"mean_cycles_per_instruction": additive, "counter_reading_if_saturated": additive_reading,
                                   "issue_roofline": insts * additive / (SIMDS * CLOCK) * 1e6 / e["avg_us"]},
                "mixed_loop": mixname,
                "kernel_us": e["avg_us"], "valu_wave_instructions_per_launch": insts,
                "loop_class_mix": {c: round(v, 4) for c, v in sorted(frac.items())},
                "static_valu_in_loops": loop_valu, "static_valu_total": total_valu,
                "mean_cycles_per_instruction_priced": cyc_per_inst,
                "issue_time_us": issue_s * 1e6,
                "issue_roofline": issue_s * 1e6 / e["avg_us"],
                "counter_reading": e.get("valu_busy_of_8"),
                "counter_reading_if_saturated": sat_reading,
                "counter_frac_of_saturated": (e.get("valu_busy_of_8") or 0.0) / sat_reading,
            }
        out["profiles"][f] = rows
    dst = ROOT / "profiles" / "issue_roofline_r06.json"
    dst.write_text(json.dumps(out, indent=1, sort_keys=True))
    for f, rows in out["profiles"].items():
        for k, r in rows.items():
            print(f, k, "issue_roofline %.3f (%.1f of %.1f us), priced %.2f cyc/inst; counter %.2f of %.2f saturated = %.2f"
                  % (r["issue_roofline"], r["issue_time_us"], r["kernel_us"], r["mean_cycles_per_instruction_priced"],
                     r["counter_reading"] or 0, r["counter_reading_if_saturated"], r["counter_frac_of_saturated"]))
            print("    mix", r["loop_class_mix"])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh per kernel:
    python tools/pmc_summary.py gpurun_out/pmc_cfg2 profiles/r03_pmc_cfg2.json

Per kernel (mean over its dispatches; dispatches of the warm-up included -- they run the same code on the same shapes):
  duration_us          End - Start timestamps of the dispatch (of the counter pass: ~5-10 % slower than un-profiled runs)
  counters             raw means.  Units (MI355X_MICROARCH.md, constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
                       count quad-cycles summed over the waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles; GRBM_GUI_ACTIVE is summed
                       over the 8 XCDs; FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE tallies 128-B requests at 64 B on gfx950)
  derived
    (no clock: GRBM_GUI_ACTIVE / duration reads 2.5 - 7 "GHz" for dispatches well under 0.3 ms -- the guide's caveat -- and is
     not stored.  The measured clocks are in profiles/r04_clock.txt: 2.39 - 2.40 GHz sustained for any instruction mix
     (csrc/probes/clock_probe.hip), 2.15 - 2.28 GHz inside the 10 - 50 us kernels of the step (s_memtime against s_memrealtime,
     tools/kbench.py with the stamps build).)
    waves_per_simd     mean resident waves = SQ_WAVE_CYCLES * 4 / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
    wave_active_valu   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: share of a wave's lifetime spent issuing vector (incl. matrix) work
    wave_wait_any      SQ_WAIT_ANY / SQ_WAVE_CYCLES: parked on s_waitcnt / barriers
    wave_wait_inst     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: issue stalled (matrix-pipe / dependency)
    mfma_pipe_util     SQ_INSTS_MFMA * 64 cycles (v_mfma_f64_16x16x4: measured, csrc/probes) / (duration * 2.4 GHz * 1024 SIMDs):
                       share of the matrix-pipe time at the nominal clock (the kernels run at 2.15 - 2.28 GHz: profiles/r04_clock.txt)
    valu_issue_util    (SQ_INSTS_VALU - SQ_INSTS_MFMA) * 4 cycles / (duration * 2.4 GHz * 1024 SIMDs)   (4 = best-case issue cost;
                       fp64 matrix and vector instructions share the datapath: the two shares add up against ONE budget)
    mfma_tflops        SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 flops / duration (MOPS unit: 512 flops) -- executed, padding included
    hbm_bytes          2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per launch
"""
import csv
import json
import os
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [values]
    dur = defaultdict(dict)                           # kernel -> dispatch -> ns
    if not os.path.exists(path):
        return per, dur
    acc = defaultdict(float)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            acc[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for (k, _, c), v in acc.items():
        per[k][c].append(v)
    return per, dur


def main():
    src, out = sys.argv[1], sys.argv[2]
    merged, durs = defaultdict(dict), {}
    for name in ("time", "insts", "fetch", "write"):
        per, dur = load(os.path.join(src, name + ".csv"))
        for k, cs in per.items():
            for c, vs in cs.items():
                merged[k][c] = sum(vs) / len(vs)
            if name == "time" or k not in durs:
                d = list(dur[k].values())
                durs[k] = (sum(d) / len(d), len(d))
    res = {}
    for k, c in merged.items():
        if not k.startswith(("void lgn::", "lgn::")):
            continue
        ns, n = durs[k]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        d = {"dispatches": n, "duration_us": ns / 1e3, "counters": {x: round(v, 1) for x, v in sorted(c.items())}, "derived": {}}
        dv = d["derived"]
        if gui > 0 and "SQ_WAVE_CYCLES" in c:
            dv["waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / (gui * 1024)
        if "SQ_INSTS_MFMA" in c:
            nominal = ns * 2.4 * 1024                      # SIMD cycles at the nominal 2.4 GHz
            dv["mfma_pipe_util"] = c["SQ_INSTS_MFMA"] * 64 / nominal
            dv["valu_issue_util"] = (c.get("SQ_INSTS_VALU", 0) - c["SQ_INSTS_MFMA"]) * 4 / nominal
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc > 0:
            dv["wave_active_valu"] = c.get("SQ_ACTIVE_INST_VALU", 0) / wc
            dv["wave_wait_any"] = c.get("SQ_WAIT_ANY", 0) / wc
            dv["wave_wait_inst"] = c.get("SQ_WAIT_INST_ANY", 0) / wc
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
            dv["mfma_tflops"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / (ns * 1e-9) / 1e12
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            dv["hbm_bytes"] = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
            dv["hbm_TBps"] = dv["hbm_bytes"] / (ns * 1e-9) / 1e12
        d["derived"] = {x: round(v, 4) for x, v in dv.items()}
        res[k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]] = d
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["duration_us"] * kv[1]["dispatches"]))
    with open(out, "w") as fh:
        json.dump({"_doc": __doc__.strip().splitlines()[0] + " -- see tools/pmc_summary.py for units and formulas", "kernels": res}, fh, indent=1)
    print(f"{'kernel':58s} {'us':>7s} {'w/simd':>6s} {'valu':>5s} {'wait':>5s} {'stall':>5s} {'mfma%':>6s} {'valu%':>6s} {'MB':>7s}")
    for k, d in list(res.items())[:24]:
        v = d["derived"]
        print(f"{k[:58]:58s} {d['duration_us']:7.1f} {v.get('waves_per_simd', 0):6.2f} {v.get('wave_active_valu', 0):5.2f} "
              f"{v.get('wave_wait_any', 0):5.2f} {v.get('wave_wait_inst', 0):5.2f} {100 * v.get('mfma_pipe_util', 0):6.1f} {100 * v.get('valu_issue_util', 0):6.1f} "
              f"{v.get('hbm_bytes', 0) / 1e6:7.1f}")


if __name__ == "__main__":
    main()

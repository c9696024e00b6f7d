#!/usr/bin/env python3
"""Copy what tools/evidence.sh wrote under gpurun_out/ into profiles/ under this round's names:
    python tools/evidence_collect.py gpurun_out/ev r03"""
import os
import shutil
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench.jsonl"), os.path.join(prof, f"{tag}_bench.jsonl"))
for name in ("cfg2", "cfg2_module_api", "cfg4", "cfg5", "b64", "cfg2_mlp_v1", "cfg2_split_tail"):
    f = os.path.join(src, f"{name}_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(prof, f"{tag}_{name}_kernel_stats.csv"))
    else:
        print("missing", f)
for cfg in ("cfg2", "cfg5"):
    d = os.path.join(src, f"pmc_{cfg}")
    if os.path.isdir(d):
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), d, os.path.join(prof, f"{tag}_pmc_{cfg}.json")])
    else:
        print("missing", d)
for name in ("cfg5_ab.txt", "kernel_sequence_cfg5.txt", "kernel_sequence_cfg2.txt", "kernel_sequence_b64.txt", "local_stamps.txt", "mlp_roles_ab.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(prof, f"{tag}_{name}"))
    else:
        print("missing", f)
f = os.path.join(src, "step_tail.txt")
if os.path.exists(f):
    shutil.copy(f, os.path.join(prof, f"{tag}_step_tail.txt"))

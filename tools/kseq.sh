#!/bin/bash
# kernel SEQUENCE of one step (rocprofv3 --kernel-trace): names and durations of the launches between two step-closing kernels
# (l1_adam, or step_tail of the single-process step).
#   bash tools/kseq.sh NAME -- python3 bench.py --config cfg5 --no-cpu-baseline --no-extras --steps 3 --warmup 1
NAME=$1; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
ARGS=(); for x in "$@"; do if [ -f "$ROOT/$x" ]; then ARGS+=("$ROOT/$x"); else ARGS+=("$x"); fi; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/ks_$NAME" -- "${ARGS[@]}" > "$OUT/ks_$NAME.log" 2>&1 || { tail -20 "$OUT/ks_$NAME.log"; exit 1; }
cd "$ROOT"
python3 - "$OUT/ks_$NAME" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "l1_adam" in r["Kernel_Name"] or "step_tail" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:5.1f}  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:110]}")
    prev_end = e
print(f"step: {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches")
PY
rm -rf "$OUT/ks_$NAME"

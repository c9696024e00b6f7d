#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command, top kernels printed:   bash tools/kprof.sh NAME [TOP] -- python3 tools/kbench.py mlp_bwd 50
# (the summary CSV stays in gpurun_out/kp_NAME_kernel_stats.csv)
NAME=$1; shift; TOP=12
if [ "$1" != "--" ]; then TOP=$1; shift; fi
shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
ARGS=(); for x in "$@"; do if [ -f "$ROOT/$x" ]; then ARGS+=("$ROOT/$x"); else ARGS+=("$x"); fi; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kp_$NAME" -- "${ARGS[@]}" > "$OUT/kp_$NAME.log" 2>&1 || { tail -20 "$OUT/kp_$NAME.log"; exit 1; }
cd "$ROOT"
python3 tools/kstats.py "$OUT/kp_$NAME" $TOP
find "$OUT/kp_$NAME" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kp_${NAME}_kernel_stats.csv"
rm -rf "$OUT/kp_$NAME"

#!/bin/bash
# one counter pass (LDS instructions / bank-conflict cycles per kernel) over a short cfg2 bench run:   bash tools/pmc_lds.sh OUTDIR
OUT=${1:-gpurun_out/pmc_lds}; ROOT=$(pwd); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/run" -- python3 "$ROOT/bench.py" --config cfg2 --no-cpu-baseline --no-extras --steps 6 --warmup 2 > "$OUT/run.log" 2>&1
find "$OUT/run" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/lds.csv"; rm -rf "$OUT/run"
python3 - "$OUT/lds.csv" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]].add(r["Dispatch_Id"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0))[:14]:
    d = len(n[k])
    print(f"{k[:74]:74s} x{d:3d}  insts {c.get('SQ_INSTS_LDS',0)/d/1e6:6.2f}M  active {c.get('SQ_ACTIVE_INST_LDS',0)/d/1e6:6.2f}M  idx_active {c.get('SQ_LDS_IDX_ACTIVE',0)/d/1e6:6.2f}M  conflict {c.get('SQ_LDS_BANK_CONFLICT',0)/d/1e6:6.2f}M")
PY

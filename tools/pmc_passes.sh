#!/bin/bash
# SQ / TCC / GRBM counter passes over bench.py (one rocprofv3 run per counter set: 8 SQ slots, FETCH_SIZE and WRITE_SIZE apart,
# no trace domains together with --pmc).  Usage on the GPU box, from the repo root:
#     bash tools/pmc_passes.sh cfg2 gpurun_out/pmc_cfg2        (then: python tools/pmc_summary.py gpurun_out/pmc_cfg2 ...)
set -e
CFG=${1:-cfg2}; OUT=${2:-gpurun_out/pmc_$CFG}; ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --config "$CFG" --no-cpu-baseline --no-extras --steps 6 --warmup 2 > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/$name.csv"
  rm -rf "$OUT/$name"
}
run time   SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run insts  SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run fetch  FETCH_SIZE
run write  WRITE_SIZE
ls -la "$OUT"

#!/bin/bash
# In-kernel timing evidence of a round, on the GPU box from the repo root (needs `make stamps` and the probe binaries built here):
#     bash tools/clock_evidence.sh gpurun_out/clk        (then, here: cp gpurun_out/clk/clock.txt profiles/rNN_clock.txt, ...)
# clock.txt        sustained shader clock under four instruction mixes (csrc/probes/clock_probe.hip), then for every hot kernel at the
#                  cfg2 shape: lifetime of the first / last workgroup of the grid on the 100 MHz counter and the shader-clock cycles
#                  between its stamps (tools/kbench.py with the stamps build of the library)
# probes.txt       what one wave per SIMD keeps up beside a stream of fp64 matrix instructions (csrc/probes/mfma_issue_probe.hip) and
#                  the issue rates of csrc/probes/mfma_rate_probe.hip / mfma_chain_probe.hip; what a kernel node of a replayed graph
#                  costs beside its work (csrc/probes/launch_overhead_probe.hip)
# placement.txt    which workgroups share a CU (csrc/probes/placement_probe.hip)
set -e
OUT=${1:-gpurun_out/clk}
mkdir -p "$OUT"
P=lgn-autoencoder_amd/csrc/probes
echo "[clock] clock_probe"
./$P/clock_probe.probe > "$OUT/clock.txt"
kb() {  # reader, op, extra env...
  local reader=$1 op=$2; shift 2
  echo "[clock] kbench $op ($reader) $*"
  env "$@" KB_STAMPS=$reader python3 tools/kbench.py $op 20 2>&1 | grep -v amdgpu.ids
}
for spec in "lgn_debug_stamps_bwd3 level_bwd_enc" "lgn_debug_stamps_fwd2 level_fwd_enc" "lgn_debug_stamps_fwd2 level_fwd_dec" \
            "lgn_debug_stamps_bwd3 level_bwd_dec" "lgn_debug_stamps_mlp_chain mlp_fwd" "lgn_debug_stamps_mlp_chain mlp_bwd"; do
  kb $spec | grep -v "^ *[0-9]* *t=" >> "$OUT/clock.txt"
done
kb lgn_debug_stamps_bwd2 level_bwd_enc KB_BATCH=256 KB_N=150 >> "$OUT/clock.txt"
echo "[clock] mfma probes"
for pr in mfma_rate_probe mfma_chain_probe mfma_issue_probe launch_overhead_probe; do
  echo "== $pr" >> "$OUT/probes.txt"
  timeout -k 5 120 ./$P/$pr.probe >> "$OUT/probes.txt" 2>&1
done
echo "[clock] placement_probe"
./$P/placement_probe.probe > "$OUT/placement.txt"
ls -la "$OUT"

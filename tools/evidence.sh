#!/bin/bash
# One pass over everything profiles/ holds for a round, on the GPU box from the repo root:
#     bash tools/evidence.sh gpurun_out/ev            (then, here: python tools/evidence_collect.py gpurun_out/ev r03)
# bench lines (all configs + the 64-jet step), rocprofv3 --kernel-trace --stats of the same commands, and the --pmc passes of
# cfg2 / cfg5 (tools/pmc_passes.sh).  Every step prints a line, so the call never looks hung.
set -e
OUT=${1:-gpurun_out/ev}; ROOT=$(pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
: > "$OUT/bench.jsonl"
for spec in "cfg2" "cfg1" "cfg4" "cfg5" "cfg2 --batch 64"; do
  echo "[evidence] bench --config $spec"
  python3 bench.py --config $spec 2> "$OUT/bench_err.log" | grep '^{' >> "$OUT/bench.jsonl"
done
cd /tmp; export TMPDIR=/tmp
stats() {  # name, bench args...
  local name=$1; shift
  echo "[evidence] rocprofv3 --kernel-trace --stats: bench.py $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks_$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/ks_$name.log" 2>&1
  find "$OUT/ks_$name" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${name}_kernel_stats.csv"
  rm -rf "$OUT/ks_$name"
}
stats cfg2 --config cfg2 --no-extras
stats cfg2_module_api --config cfg2 --harness module --no-extras
stats cfg4 --config cfg4 --no-extras
stats cfg5 --config cfg5 --no-extras
stats b64 --config cfg2 --batch 64 --no-extras
# the 12-wave CGMLP kernels (csrc/mlp_mfma.hip) in place of the chain kernels, for the comparison in DESIGN.md
LGN_AMD_MLP_V1=1 stats cfg2_mlp_v1 --config cfg2 --no-extras
# the step's tail as three launches (reduce_segments, rad_finalize_batch, l1_adam) instead of step_tail_kernel
LGN_AMD_SPLIT_TAIL=1 stats cfg2_split_tail --config cfg2 --no-extras
cd "$ROOT"
# fused against split tail, and the data-parallel branch on one rank (tools/ab.sh, tools/dp_bench.py)
echo "[evidence] A/B of the step tail"
bash tools/ab.sh "$OUT/ab_tail" LGN_AMD_SPLIT_TAIL=1 - -- --config cfg2 --no-extras > /dev/null 2>&1
bash tools/ab.sh "$OUT/ab_tail64" LGN_AMD_SPLIT_TAIL=1 - -- --config cfg2 --batch 64 --no-extras > /dev/null 2>&1
{ echo "== 512 jets (bench.py --config cfg2): three launches (LGN_AMD_SPLIT_TAIL=1) | one launch (-)"; cat "$OUT/ab_tail/ab.txt";
  echo "== 64 jets"; cat "$OUT/ab_tail64/ab.txt";
  echo "== data-parallel branch on one rank (tools/dp_bench.py 64 300): one-launch reductions, then LGN_AMD_SPLIT_TAIL=1";
  python3 tools/dp_bench.py 64 300 2>&1 | grep jets; LGN_AMD_SPLIT_TAIL=1 python3 tools/dp_bench.py 64 300 2>&1 | grep jets; } > "$OUT/step_tail.txt"

# cfg5, round 6: the decoder's moments as a tensor between two kernels per level (LGN_AMD_DEC_UNFUSED=1: round 5) | the encoder's
# backward sweeps as two kernels (LGN_AMD_MOMENTS_SPLIT=1) | separate tail launches (LGN_AMD_SPLIT_TAIL=1) | default
echo "[evidence] A/B of the cfg5 step"
bash tools/ab.sh "$OUT/ab_cfg5" LGN_AMD_DEC_UNFUSED=1 LGN_AMD_MOMENTS_SPLIT=1 LGN_AMD_SPLIT_TAIL=1 - -- --config cfg5 --no-extras > /dev/null 2>&1
cp "$OUT/ab_cfg5/ab.txt" "$OUT/cfg5_ab.txt"
# CGMLP chain kernels, round 6: the 12-wave kernels (LGN_AMD_MLP_V1=1) | one role per wave / one chain wave per 16 rows
# (LGN_AMD_MLP_BWD1=1: round 5's chain kernels at 512 jets) | default (roles; three chain waves per 16 rows), at 512, 64 and 32 jets
echo "[evidence] A/B of the CGMLP kernels"
for b in 512 64 32; do
  bash tools/ab.sh "$OUT/ab_mlp$b" LGN_AMD_MLP_V1=1 LGN_AMD_MLP_BWD1=1 - -- --config cfg2 --batch $b --no-extras > /dev/null 2>&1
  { echo "== $b jets"; cat "$OUT/ab_mlp$b/ab.txt"; } >> "$OUT/mlp_roles_ab.txt"
done
echo "[evidence] kernel sequences"
bash tools/kseq.sh cfg5 -- python3 bench.py --config cfg5 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > "$OUT/kernel_sequence_cfg5.txt" 2>&1
bash tools/kseq.sh cfg2 -- python3 bench.py --config cfg2 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > "$OUT/kernel_sequence_cfg2.txt" 2>&1
bash tools/kseq.sh b64 -- python3 bench.py --config cfg2 --batch 64 --no-cpu-baseline --no-extras --steps 3 --warmup 1 > "$OUT/kernel_sequence_b64.txt" 2>&1
rm -f "$ROOT"/gpurun_out/ks_cfg5.log "$ROOT"/gpurun_out/ks_cfg2.log "$ROOT"/gpurun_out/ks_b64.log
echo "[evidence] in-kernel stamps of the fused decoder backward and of the static encoder backward"
{ python3 tools/sep_stamps.py 2>&1 | tail -2; python3 tools/local_bench.py stamps 2>&1 | grep -v amdgpu.ids; } > "$OUT/local_stamps.txt"
for cfg in cfg2 cfg5; do
  echo "[evidence] pmc passes $cfg"
  bash tools/pmc_passes.sh $cfg "$OUT/pmc_$cfg" > "$OUT/pmc_$cfg.log" 2>&1
done
ls -la "$OUT"

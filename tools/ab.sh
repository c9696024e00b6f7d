#!/bin/bash
# A/B of library variants inside ONE gpurun call (boxes differ by 2-5 %):  bash tools/ab.sh OUT lib1.so lib2.so ... [-- bench args]
# Each variant -- a library under lgn-autoencoder_amd/lgn/_lib/ (selected with LGN_AMD_LIB), or NAME=VALUE: a switch of the default
# library ("-" = the default library, no switch) -- runs bench.py twice, alternating.
OUT=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p "$OUT"; : > "$OUT/ab.txt"
for rep in 1 2; do
  for lib in "${LIBS[@]}"; do
    case "$lib" in
      -) VAR=LGN_AB_NONE=1 ;;
      *=*) VAR="$lib" ;;
      *) VAR="LGN_AMD_LIB=$(pwd)/lgn-autoencoder_amd/lgn/_lib/$lib" ;;
    esac
    env "$VAR" python3 bench.py --no-cpu-baseline "$@" 2>> "$OUT/ab_err.log" | grep '^{' > "$OUT/ab_line.json"
    python3 - "$OUT/ab_line.json" "$lib" "$rep" >> "$OUT/ab.txt" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
ks = " ".join(f"{k['us_per_launch']:.1f}" for k in r.get("kernels", []))
print(f"{sys.argv[2]:28s} rep {sys.argv[3]}: {d['value']:9.0f} {d['unit']}  {d['ms_per_step']:.4f} ms  dominant {r.get('us_per_launch', 0):.2f} us  kernels: {ks}")
P
  done
done
cat "$OUT/ab.txt"

#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in liblgn_amd.so, read from the gfx950 code object's metadata notes.
usage: kregs.py [PATTERN ...] [--lib PATH]     (PATTERN: substrings of the demangled kernel name; default: everything with spills)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
here = os.path.dirname(os.path.abspath(__file__))
lib = os.path.join(here, "..", "lgn-autoencoder_amd", "lgn", "_lib", "liblgn_amd.so")
args = sys.argv[1:]
if "--lib" in args:
    i = args.index("--lib")
    lib = args[i + 1]
    del args[i:i + 2]
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "x")], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
    notes = ""
    for k in range(len(starts) - 1):          # one bundle per translation unit
        part, co = os.path.join(tmp, f"b{k}"), os.path.join(tmp, f"co{k}")
        open(part, "wb").write(blob[starts[k]:starts[k + 1]])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
        notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, stdout=subprocess.PIPE).stdout.decode()
rows = []
for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
    name = get("name").strip("'\"")
    rows.append((name, get("vgpr_count"), (re.match(r"\s*(\d+)", blk) or [None, "?"])[1], get("sgpr_count"), get("vgpr_spill_count"),
                 get("private_segment_fixed_size"), get("group_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows).encode(), stdout=subprocess.PIPE).stdout.decode().split("\n")
print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spill':>6} {'scratch':>8} {'lds':>7}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"^void lgn::\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    if args and not any(a in n for a in args):
        continue
    if not args and r[4] in ("0", "?"):
        continue
    print(f"{r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>6} {r[5]:>8} {r[6]:>7}  {n}")

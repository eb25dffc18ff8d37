#!/bin/bash
# What would the traversal kernels cost in registers if the row length were a compile-time constant?  Static experiment, no GPU and
# no change to the product: a scratch copy of csrc/ gets `__builtin_assume(dim == D)` as the first statement of graph_search_kernel and
# graph_insert_search_kernel, the traversal units are compiled with and without -DEXP_ASSUME_DIM=D, and the compiler's resource remarks
# (hipcc -Rpass-analysis=kernel-resource-usage) are printed side by side.  `dim` is the row length in 32-bit words: 128 for C2 / C4,
# 32 for C5's 128-byte int8 records.  Round 5's result (profiles/r5_fixed_dim_static.txt) is the lead DESIGN.md 9 puts first.
#   usage: tools/fixed_dim_static.sh > profiles/r5_fixed_dim_static.txt
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
mkdir -p $T/pkg/x && cp -r $R/hnswindex.net_amd/csrc $T/pkg/x/csrc && cp -r $R/include $T/pkg/include   # csrc includes ../../include/...
cd $T/pkg/x/csrc
python3 - <<'E'
import re
for f, kern in (("dk_search_kernels.h", "graph_search_kernel(const float *__restrict__ rows"), ("dk_insert_kernels.h", "graph_insert_search_kernel(const float *__restrict__ rows")):
    s = open(f).read()
    i = s.index(kern)
    j = s.index("{\n", i)
    s = s[:j] + "{\n#ifdef EXP_ASSUME_DIM\n    __builtin_assume(dim == EXP_ASSUME_DIM);\n#endif\n" + s[j + 2:]
    open(f, "w").write(s)
E
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I. --cuda-device-only"
compile() { # unit, dim (0 = as shipped), tag
  local d=""; [ "$2" != 0 ] && d="-DEXP_ASSUME_DIM=$2"
  hipcc $FLAGS $d -c $1 -o $T/$3.o -Rpass-analysis=kernel-resource-usage 2> $T/$3.txt
}
report() { # unit, dim, tag
  python3 - $T/$3.txt $T/$3.o "$1" "$2" <<'E'
import re, subprocess, sys, os
txt, obj, unit, dim = sys.argv[1:5]
rows, cur = [], None
for line in open(txt):
    m = re.search(r"remark:\s*(.*?)\s*\[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"k": re.sub(r"\(.*", "", name).replace("void hnsw::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
print(f"== {unit}   dim {'runtime (as shipped)' if dim == '0' else 'assumed ' + dim}   device object {os.path.getsize(obj)} bytes")
for r in rows:
    print(f"   {r['k']:<52} VGPRs {r.get('VGPRs'):>3}  waves/SIMD the registers allow {r.get('Occupancy [waves/SIMD]')}  spilled SGPRs {r.get('SGPRs Spill'):>3}  "
          f"spilled VGPRs {r.get('VGPRs Spill')}  scratch {r.get('ScratchSize [bytes/lane]')}")
E
}
echo "# tools/fixed_dim_static.sh: the traversal kernels with the row length known at compile time (static; scratch copy + __builtin_assume; the product is unchanged)"
echo "# graph_*_kernel<METRIC, NS, HASHED, FORM>: form 1 = latency variant, 2 = lean; 'waves/SIMD' is what the register count allows, the shipped kernels pin theirs with amdgpu_waves_per_eu"
SPECS="traverse_sq_search_lean.hip:128 traverse_i8_search_lean.hip:32 traverse_sq_search_lat.hip:128 traverse_sq_insert_lat.hip:128"
for spec in $SPECS; do
  set -- ${spec%%:*} ${spec##*:}
  compile $1 0 a_${1%.hip} &
  compile $1 $2 b_${1%.hip} &
done
wait
for spec in $SPECS; do
  u=${spec%%:*}; d=${spec##*:}
  report $u 0 a_${u%.hip}
  report $u $d b_${u%.hip}
done

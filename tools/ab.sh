#!/bin/bash
# GPU box, repo root: tools/ab.sh <tag> <pytest -k expr | -> name1 name2 ...
#   names: "main" = cleanba_amd/libcleanba_mi.so, anything else = cleanba_amd/abl_<name>.so (tools/variants.sh, or a saved copy of an older build).
# Per library: tools/microbench.py (isolated per-kernel times of one 3840-frame minibatch) and tools/pipeline_probe.py (rollout alone / update alone /
# pipelined step), two interleaved rounds so that a box-level drift shows up as a difference between rounds rather than between libraries.
# The parity tests named by the -k expression run first, on the shipped library.
tag=$1; kexpr=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
if [ "$kexpr" != "-" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -15 > $out/pytest.log; tail -4 $out/pytest.log
fi
for round in 1 2; do
  for name in "$@"; do
    so=$PWD/cleanba_amd/abl_$name.so; [ $name = main ] && so=$PWD/cleanba_amd/libcleanba_mi.so
    echo "== $name round $round" | tee -a $out/micro.txt $out/pipe.txt > /dev/null
    CBM_SO=$so timeout 300 python tools/microbench.py 8 >> $out/micro.txt 2>> $out/err.txt
    CBM_SO=$so timeout 300 python tools/pipeline_probe.py >> $out/pipe.txt 2>> $out/err.txt
  done
done
python - "$out" <<'PY'
import re, sys, collections
out = sys.argv[1]
cur = None; rows = collections.OrderedDict(); names = []
for l in open(out + "/micro.txt"):
    m = re.match(r"== (\S+) round (\d)", l)
    if m: cur = m.group(1) + "#" + m.group(2); names.append(cur); continue
    m = re.match(r"\s+(\w+)\s+([\d.]+) us", l)
    if m: rows.setdefault(m.group(1), {})[cur] = float(m.group(2))
    m = re.match(r"minibatch fwd\+loss\+bwd: ([\d.]+) ms", l)
    if m: rows.setdefault("minibatch_ms", {})[cur] = float(m.group(1)) * 1000
    m = re.match(r"\s+sum of GEMM kernels ([\d.]+)", l)
    if m: rows.setdefault("sum_gemm", {})[cur] = float(m.group(1))
print("%-14s" % "isolated us" + "".join("%12s" % n[:11] for n in names))
for k, v in rows.items(): print("%-14s" % k + "".join("%12.1f" % v.get(n, float("nan")) for n in names))
print(open(out + "/pipe.txt").read())
PY

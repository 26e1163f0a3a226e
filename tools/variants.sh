#!/bin/bash
# Build ablation variants of libcleanba_mi.so (extra -D flags for the kernel translation units) into cleanba_amd/abl_<name>.so.
# usage: tools/variants.sh name1 "-DFOO=1 -DBAR=2" name2 "-DBAZ" ...      then on the GPU box: tools/variants_run.sh name1 name2 ...
set -e
cd "$(dirname "$0")/../cleanba_amd/csrc"
F="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-math-errno -Wno-unused-function -Wno-unused-result -Wno-unused-value -Wno-pass-failed"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift; shift
  d=/tmp/abl_$name; mkdir -p $d
  for f in gemm_layers conv1 wgrad_frames conv_regw dense_wgrad pointwise; do /opt/rocm/bin/hipcc $F $defs -c $f.hip -o $d/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o comm.o $d/pointwise.o env.o $d/dense_wgrad.o $d/conv_regw.o $d/gemm_layers.o $d/conv1.o $d/wgrad_frames.o -ldl -o ../abl_$name.so
  echo built abl_$name.so "($defs)"
done

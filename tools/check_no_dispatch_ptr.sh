#!/bin/bash
# No kernel of the library may ask for the dispatch packet pointer: the compiler does when it has moved a private array to LDS
# (amdgpu-promote-alloca needs the block shape), and reading that packet at kernel entry is a host-memory round trip — 9-16 us per launch
# measured on the actor tail (profiles/NOTES_r03_r04.md round 4).  Compiles every translation unit to ISA and lists offenders; exit 1 if any.
cd "$(dirname "$0")/../cleanba_amd/csrc"
F="-O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off -fno-math-errno -S --cuda-device-only"
bad=0
for f in api comm gemm_layers conv1 wgrad_frames dense_wgrad conv_regw pointwise env; do
  /opt/rocm/bin/hipcc $F -o /tmp/chk_$f.s $f.hip 2>/dev/null &
done
wait
for f in api comm gemm_layers conv1 wgrad_frames dense_wgrad conv_regw pointwise env; do
  out=$(awk -v F=$f '/\.amdhsa_kernel /{name=$2} /amdhsa_user_sgpr_dispatch_ptr 1/{print "  " F ".hip: " name}' /tmp/chk_$f.s)
  if [ -n "$out" ]; then echo "$out"; bad=1; fi
done
[ $bad = 0 ] && echo "no kernel reads the dispatch packet" || { echo "kernels above read the dispatch packet (a private array was moved to LDS)"; exit 1; }

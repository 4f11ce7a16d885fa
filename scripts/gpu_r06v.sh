#!/bin/bash
# round 6: SQ counters of conv_wino4 on HandSegNet conv4_2 at the bench shape (32 x 40x40, 512 -> 512) -- the tree against the kernel as it was before
# the round's two late changes (libhp3d_w4r5form.so: fourteen-operation transform, one v_add_u32 in front of each window load)  -> profiles/r06_sq_counters.md
for L in libhp3d libhp3d_w4r5form; do
  HP3D_LIB=$(pwd)/hand3d_amd/$L.so bash scripts/gpu_w2pmc.sh r06v/$L 32 40 40 512 512 0 3 wino4 > gpurun_out/r06v_$L.txt 2>&1
  echo "== $L"; tail -4 gpurun_out/r06v_$L.txt
done

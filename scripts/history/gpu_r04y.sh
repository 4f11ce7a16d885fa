#!/bin/bash
# round 4: conv_wino4w.hip after the first A/B (gpu_r04x.sh): its GPU tests (per-op incl. tail pieces, whole path at B = 32), then the default bench line
# with the option off / on (mode 1 = the layers with Cin >= 256) and on with a weight ring of 9 half planes (libhp3d_ring9.so), same box
OUT=gpurun_out/${1:-r04y}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide_items" -p no:cacheprovider -s > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "wide items|passed|failed" $OUT/pytest.log | tail -12
run() { ( if [ -n "$2" ]; then export HP3D_LIB=$2; fi; timeout 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --option wino4_wide=$3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err ); python -c "
import json; d=json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run base "" 0
run wide "" 1
run wide_ring9 hand3d_amd/libhp3d_ring9.so 1
run base2 "" 0

#!/bin/bash
# round 4: after conv_h16's 7x7 / 1x1 forms: f16 parity tests + reference-code C5 fixture, config 5's bench line and per-layer table, then the
# half-precision counter passes (scripts/gpu_pmc_h16.sh; summarise HERE afterwards: python scripts/h16_counters.py gpurun_out/<tag>/pmc r04)
OUT=gpurun_out/${1:-r04v}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5_fixture.py -m gpu -q -k "f16 or c5" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log
timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 300 $OUT/bench_c5.json
timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 3 --warmup 1 --cpu-seconds 0 --no-host-path --layers --option streams=1 > $OUT/layers_c5.txt 2>&1
bash scripts/gpu_pmc_h16.sh ${1:-r04v}/pmc r04

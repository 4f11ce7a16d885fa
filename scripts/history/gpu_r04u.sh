#!/bin/bash
# round 4: conv_h16's 7x7 / 1x1 forms (option f16_k7k1): the f16 parity tests (incl. the reference-code C5 fixture), then config 5's per-layer
# table and bench line with the forms off / on
OUT=gpurun_out/${1:-r04u}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5_fixture.py -m gpu -q -x -k "f16 or c5" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log
for v in 0 1; do
  timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --option f16_k7k1=$v > $OUT/bench_k$v.json 2> $OUT/bench_k$v.err; tail -c 600 $OUT/bench_k$v.json
  timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 3 --warmup 1 --cpu-seconds 0 --no-host-path --layers --option streams=1 --option f16_k7k1=$v > $OUT/layers_k$v.txt 2>&1; grep -E "7x7|1x1|Mconv|conv6|total" $OUT/layers_k$v.txt | head -40
done

#!/bin/bash
# round 4, call L: the whole GPU suite on the current tree (epilogue fast path, row-wise up-sampling, one-stream default), per-layer tables
# of the half-precision mode (config 5 shape and the bench shape), mask flips against the float64 oracle (cache made on the CPU box)
OUT=gpurun_out/${1:-r04l}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rx > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 0 --no-host-path > $OUT/bench_f32.json 2> $OUT/layers_f32.txt; python -c "import json;d=json.load(open('$OUT/bench_f32.json'));print('f32 B32 320', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --layers --cpu-seconds 0 --no-host-path --dtype f16 --batch 128 --height 480 --width 640 > $OUT/bench_c5.json 2> $OUT/layers_c5.txt; python -c "import json;d=json.load(open('$OUT/bench_c5.json'));print('C5 f16 B128 480x640', d['value'], d['ms_per_step'], d['roofline'])"
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --layers --cpu-seconds 0 --no-host-path --dtype f16 > $OUT/bench_f16_b32.json 2> $OUT/layers_f16_b32.txt; python -c "import json;d=json.load(open('$OUT/bench_f16_b32.json'));print('f16 B32 320', d['value'], d['ms_per_step'])"
timeout 600 python tests/helpers/mask_flip_vs_oracle.py 256 $OUT/mask_flip_vs_oracle.md > $OUT/mask_flip.log 2>&1; echo "flip exit $?"; tail -6 $OUT/mask_flip.log

OUT=gpurun_out/f16; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "f16" -s 2>&1 | tail -12
for D in f32 f16; do
python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-host-path --layers --dtype $D > $OUT/b_$D.json 2> $OUT/layers_$D.txt
python -c "
import json;d=json.loads(open('$OUT/b_$D.json').read().strip().splitlines()[-1]);print('$D', d['value'],d['ms_per_step'],d['roofline']['achieved'], d['roofline']['frac'])"
done

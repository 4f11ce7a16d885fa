for S in 1 2 4; do python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-images 0 --streams $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $S', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done

#!/bin/bash
# round 6, final visit of the committed tree: the whole GPU suite + rocprofv3 stats + PMC passes + bench with layers (scripts/gpu_round.sh),
# smoke(), the config-5 line + its PMC traffic passes, and the driver's exact command (-> profiles/r06_* via scripts/summarize_prof.py,
# scripts/h16_counters.py and scripts/make_configs_md.py)
TAG=${1:-r06}
# (before the call, here: `git rev-parse --short HEAD > .tree_id` -- the snapshot carries no .git, the traffic files' stamps name the tree from it)
bash scripts/gpu_round.sh $TAG pmc
OUT=gpurun_out/$TAG
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
timeout 300 python bench.py --dtype f16 --batch 128 --height 480 --width 640 --steps 10 --warmup 2 --cpu-seconds 0 --no-host-path --no-other-configs --layers > $OUT/bench_c5_f16.json 2> $OUT/bench_layers_c5_f16.txt; echo "c5 exit $?"
bash scripts/gpu_pmc_h16.sh $TAG/pmch16 $TAG > $OUT/pmc_h16.log 2>&1; echo "pmc h16 exit $?"; tail -3 $OUT/pmc_h16.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2> $OUT/bench_driver_time.txt; echo "driver cmd exit $?"; tail -3 $OUT/bench_driver_time.txt
python -c "
import json
d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1])
print('driver line', d['value'], d['value_min'], d['value_max'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('value_host_u8'), 'other_configs wall', d.get('other_configs_wall_s'))
for c in d.get('other_configs', []): print('  ', c.get('config'), c.get('images_per_s'), c.get('ms_per_step'), c.get('executed_frac_of_dense_peak'), c.get('error'))
c=json.loads(open('$OUT/bench_c5_f16.json').read().strip().splitlines()[-1]); print('c5', c['value'], c['ms_per_step'], c['roofline']['frac'], c['roofline'].get('traffic'))
"

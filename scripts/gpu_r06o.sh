#!/bin/bash
# round 6: bench.py under the driver's launcher at world size 1 on real hardware: RCCL through the C ABI (comm init under the deadline, weight
# broadcast, per-step keypoint all-gather), the JSON line's comm / rccl_ranks fields; then --gpus 2 on a one-GPU box (must refuse in one line)
OUT=gpurun_out/r06o; mkdir -p $OUT
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-other-configs > $OUT/launched.json 2> $OUT/launched.err ) 2>&1 | grep real
echo "launched exit $?"; python - <<'P'
import json
c=json.loads(open('gpurun_out/r06o/launched.json').read().strip().splitlines()[-1])
print('launched:', c['value'], c['ms_per_step'], c['n_gpus'], c['config']['comm'], c['config']['rccl_ranks'], c['config'].get('cpu_affinity'))
P
tail -3 $OUT/launched.err
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/two.json 2> $OUT/two.err; echo "gpus2 exit $?"; tail -2 $OUT/two.err
HP3D_RCCL_TIMEOUT=5 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 0 --no-other-configs --no-host-path 2>/dev/null | python -c "
import sys,json; c=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deadline 5 s:', c['value'], c['config']['comm'], c['config']['rccl_ranks'])"

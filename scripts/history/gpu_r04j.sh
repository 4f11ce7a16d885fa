OUT=gpurun_out/r04j; mkdir -p $OUT
python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --steps 50 --warmup 10 --workload posenet --batch 1 --layers > $OUT/c2.json 2> $OUT/c2_layers.txt
python bench.py --gpus 1 --cpu-seconds 0 --no-host-path --steps 50 --warmup 10 --batch 1 --height 240 --width 320 --layers > $OUT/c1.json 2> $OUT/c1_layers.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /root/repo/$OUT/trace -o t -- python /root/repo/bench.py --gpus 1 --cpu-seconds 0 --no-host-path --steps 20 --warmup 5 --workload posenet --batch 1 > /dev/null 2>&1
ls /root/repo/$OUT/trace

#!/bin/bash
# Round-end check of the tree as the driver does it: the whole GPU suite, smoke(), the default bench line.
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -rx > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; head -c 400 $OUT/bench.json

#!/bin/bash
# round 5: the float32 first block (conv1_1 -> conv1_2 + pool) in image chunks through a chunk-sized activation (option first_chunk),
# with plain and non-temporal conv1_1 stores, against the whole-batch launches; same box.
# (REJECTED: the options first_chunk / first_chunk_nt existed only for this visit and were removed from the tree afterwards -- profiles/r05_tuning_notes.md section 11)
OUT=gpurun_out/${1:-r05m}
mkdir -p $OUT
B="python bench.py --cpu-seconds 0 --no-host-path --no-other-configs --steps 20 --warmup 5 --layers"
run() { # tag, options...
  T=$1; shift
  $B "$@" > $OUT/$T.json 2> $OUT/$T.txt
  echo "== $T: $(python -c "import json; d=json.load(open('$OUT/$T.json')); print(d['ms_per_step'], d['value'])")"
  python - <<PY
import re
rows = {}
for ln in open('$OUT/$T.txt'):
    m = re.match(r'(\S+/conv1_[12])\s+\S+\s+([0-9.]+)', ln)
    if m: rows[m.group(1)] = rows.get(m.group(1), 0.0) + float(m.group(2))
print('   ' + '  '.join('%s %.3f' % kv for kv in rows.items()) + '   first blocks total %.3f' % sum(rows.values()))
PY
}
run whole
run chunk8 --option first_chunk=8
run chunk8nt --option first_chunk=8 --option first_chunk_nt=1
run chunk4 --option first_chunk=4
run chunk16 --option first_chunk=16
run chunk16nt --option first_chunk=16 --option first_chunk_nt=1
run whole2
run chunk8b --option first_chunk=8

#!/bin/bash
# secondary benches under environment settings: bash tools/ab_sapg_env.sh "A=1" "A=0"
for setting in "$@"; do
  for cfg in 3 4 5; do
    (export $setting; python3 tools/bench_sapg.py --config $cfg --iters 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$setting', 'config $cfg', round(d['value'],1), d['unit'])")
  done
done

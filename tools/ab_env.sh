#!/bin/bash
# bench A/B over environment settings: bash tools/ab_env.sh "A=1 B=0" "A=0" ...   (each argument = one setting; run on the GPU box)
set -eo pipefail
for rep in 1 2; do
for setting in "$@"; do
  (export $setting; python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-batched --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$setting', round(d['value'],1), round(d['ms_per_step'],5), d['final_psnr_db'])")
done; done

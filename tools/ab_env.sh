#!/bin/bash
# bench A/B over environment settings: bash tools/ab_env.sh [-n REPS] "A=1 B=0" "A=0" ...   (each argument = one setting;
# settings are alternated REPS times; run on the GPU box)
set -eo pipefail
REPS=2
if [ "$1" = "-n" ]; then REPS=$2; shift 2; fi
for rep in $(seq $REPS); do
for setting in "$@"; do
  (export $setting; python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-batched 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$setting', round(d['value'],1), round(d['ms_per_step'],5), d['final_psnr_db'], '512^2:', round(d['extra_512']['value']))")
done; done

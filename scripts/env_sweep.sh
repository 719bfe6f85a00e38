#!/bin/bash
# Step time of the graph-replayed bf16 train step under a list of environment settings (one bench run each):
#   bash scripts/env_sweep.sh "A=1" "B=2 C=3" ...        ("X=1" = the defaults)
for cfg in "$@"; do
  printf "%-44s " "$cfg"
  env $cfg timeout -k 10 150 python bench.py --no-f32 --no-cpu-baseline --profile-steps 0 --steps 200 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'])"
done

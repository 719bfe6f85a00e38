#!/bin/bash
# repeat the graph-replayed bench under an environment setting and count NaN losses (race hunting)
n=${N:-8}
bad=0
for i in $(seq 1 $n); do
  out=$(timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-steps 0 --no-f32 2>/dev/null | tail -1 | python3 -c "import sys,json,math; j=json.loads(sys.stdin.read()); l=j['last_step_losses']; print('NAN' if any(isinstance(v,float) and math.isnan(v) for v in l.values()) else 'ok')")
  [ "$out" = "NAN" ] && bad=$((bad+1))
done
echo "$* : $bad / $n NaN"

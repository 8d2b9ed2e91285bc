# developer A/B: where the fused policy's Threefry sits in the step kernel
for rep in 1 2 3; do
  for v in "" "--flags=-DRBG_PROBE_POLICY_LATE"; do
    timeout -k 10 200 python tools/probe_envgen.py --steps 2000 --warm 200 --names online_parallel_rw,offline_parallel_rw --configs 20:1:1 $v 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('${v:-early}', d['generator'], d['us_per_step'])
"
  done
done

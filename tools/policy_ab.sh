# developer A/B on one box: the env step with the policy inside the launch, this tree against
# prebuilt libraries of earlier commits (tools/ab/librbg_<tag>.so) and variant builds
#   bash tools/policy_ab.sh [tag ...]      e.g. r4b r4c
for rep in 1 2 3; do
  for v in tree "$@"; do
    case "$v" in
      tree) arg="" ;;
      -D*) arg="--flags=$v" ;;
      *) arg="--lib=tools/ab/librbg_$v.so" ;;
    esac
    timeout -k 10 200 python tools/probe_envgen.py --steps 2000 --warm 200 --names online_parallel_rw,offline_parallel_rw --configs 20:1:1 $arg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['generator'], d['us_per_step'])
"
  done
done

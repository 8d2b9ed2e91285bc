# developer A/B on one box: the two rolling generators' env step, this tree against prebuilt
# libraries of earlier commits (tools/ab/librbg_<tag>.so) or variant builds (-D...)
for rep in 1 2; do
  for v in tree "$@"; do
    case "$v" in
      tree) arg="" ;;
      -D*) arg="--flags=$v" ;;
      *) arg="--lib=tools/ab/librbg_$v.so" ;;
    esac
    for spec in "online_seq_rw 32:0:8:3:1 300" "online_random_seed 64:0:10:3:2 400"; do
      set -- $spec "$@"
      timeout -k 10 250 python tools/probe_envgen.py --steps 1500 --warm $3 --names $1 --configs $2 $arg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['generator'], d['us_per_step'])
"
      shift 3
    done
  done
done

# Developer probe: the rolling-refill defaults and variant builds of the library on ONE box
# (tools/probe_envgen.py rows, one JSON line per measurement, into gpurun_out/).
set -u
mkdir -p gpurun_out
OUT=gpurun_out/${OUT:-r4_envgen_variants.jsonl}
run() { # label, probe_envgen arguments...
  label="$1"; shift
  for spec in "online_seq_rw ${SEQ:-32:0:8:3:1} 300" "online_random_seed ${SEED:-64:0:10:3:2} 400"; do
    name=$(echo $spec | cut -d' ' -f1); cfg=$(echo $spec | cut -d' ' -f2); warm=$(echo $spec | cut -d' ' -f3)
    timeout -k 10 250 python tools/probe_envgen.py --steps 1500 --warm $warm --names $name --configs $cfg "$@" 2>gpurun_out/r4_envgen_err.log | while read -r line; do
      echo "{\"variant\": \"$label\", \"row\": $line}" | tee -a $OUT
    done
  done
}
for v in "$@"; do
  case "$v" in
    default) run default ;;
    no_miss_snapshot) run no_miss_snapshot --no-miss-snapshot ;;
    sweep) SEQ=32:0:8:3:1,32:0:8:3:2,24:0:8:3:1,40:0:8:3:1,48:0:10:3:1 SEED=64:0:10:3:2,64:0:10:3:3,48:0:10:3:2,80:0:10:3:2,56:0:10:3:3,64:0:12:3:2 run sweep ;;
    *) run "$v" --flags=-DRBG_PROBE_$v ;;
  esac
done

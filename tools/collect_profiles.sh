# rocprofv3 passes behind profiles/rN_* (run on the GPU box through gpurun, in two calls:
#   bash tools/collect_profiles.sh a [tag]   default run (+ SQ issue counters) and the step population
#   bash tools/collect_profiles.sh s [tag]   the step population alone
#   bash tools/collect_profiles.sh b [tag]   stream (65 536 and 262 144 envs), record, --all-configs
# then `python tools/summarize_profile.py <tag> "" _step _stream _stream262k _record` here;
# tag defaults to r4).
# Counters are collected in their own passes (--pmc with --kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="rocprofv3 --output-format csv"
T=${2:-r4}
run() { # tag-suffix, args...
  sfx=$1; shift
  timeout -k 10 250 $P --kernel-trace --stats -d gpurun_out/prof_${T}${sfx}_trace -o x -- python3 bench.py "$@" > gpurun_out/prof_${T}${sfx}_bench.json 2> gpurun_out/prof_${T}${sfx}_trace.err
  echo "trace$sfx done"
  timeout -k 10 250 $P --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_${T}${sfx}_fetch -o x -- python3 bench.py "$@" > /dev/null 2> gpurun_out/prof_${T}${sfx}_fetch.err
  timeout -k 10 250 $P --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_${T}${sfx}_write -o x -- python3 bench.py "$@" > /dev/null 2> gpurun_out/prof_${T}${sfx}_write.err
  echo "pmc$sfx done"
}
if [ "$1" = "a" ]; then
  run "" --cpu-sample 0
  timeout -k 10 250 $P --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d gpurun_out/prof_${T}_issue -o x -- python3 bench.py --mode generator --cpu-sample 0 > /dev/null 2> gpurun_out/prof_${T}_issue.err
  echo "issue done"
  run _step --mode step-replay
elif [ "$1" = "s" ]; then  # the step population alone
  run _step --mode step-replay
else
  run _stream --mode stream
  run _stream262k --mode stream --stream-batch 262144
  run _record --mode record
  timeout -k 10 250 $P --kernel-trace --stats -d gpurun_out/prof_${T}_configs_trace -o x -- python3 bench.py --all-configs > gpurun_out/prof_${T}_configs.json 2> gpurun_out/prof_${T}_configs.err
fi
echo all done

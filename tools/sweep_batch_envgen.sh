# developer sweep: the rolling defaults of SeedExtension / SequentialRandomWalk at other batch sizes
for B in 256 1024 16384; do
  timeout -k 10 300 python tools/probe_envgen.py --names online_random_seed --batch $B --steps 1500 --warm 300 --configs 64:0:10:3:2,48:1:2,24:1:2 || exit 1
  timeout -k 10 300 python tools/probe_envgen.py --names online_seq_rw --batch $B --steps 1500 --warm 300 --configs 32:0:8:3:1,48:1:2,24:1:2 || exit 1
  timeout -k 10 300 python tools/probe_envgen.py --names online_parallel_rw --batch $B --steps 1500 --warm 300 --configs 20:1:1,0:1:1 || exit 1
done

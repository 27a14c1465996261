# usage: tools/lanes_sweep.sh <out dir> "<lanes> <slots>" ...   (run from the repo root on the GPU box)
out=${1:-gpurun_out/lanes}; shift
mkdir -p $out
for cfg in "$@"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --lanes $1 --slots $2 --no-cpu-baseline --no-batch64 --no-configs1 --no-reuse-sensitivity > $out/lanes_$1_$2.json 2> $out/lanes_$1_$2.err
  python tools/show_line.py $out/lanes_$1_$2.json "lanes $1 slots $2" || tail -5 $out/lanes_$1_$2.err
done

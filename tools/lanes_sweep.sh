mkdir -p gpurun_out/r3i
for cfg in "2 256" "3 256" "2 512" "4 128"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --lanes $1 --slots $2 --no-cpu-baseline --no-batch64 --no-configs1 > gpurun_out/r3i/lanes_$1_$2.json 2> gpurun_out/r3i/lanes_$1_$2.err
  python tools/show_line.py gpurun_out/r3i/lanes_$1_$2.json "lanes $1 slots $2" || tail -5 gpurun_out/r3i/lanes_$1_$2.err
done

# the train step with the split Winograd kernel against the fp32 one: alternating processes on ONE box (bench.py --steps 30 --no-also)
mkdir -p gpurun_out
{
echo "# alternating processes on one box: python bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline; value = volume-pairs/s (ms per step median)"
for i in 1 2 3 4; do for x in 1 0; do
  v=$(TMF_WINO_X=$x timeout 300 python bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step_median'])")
  echo "TMF_WINO_X=$x  $v"
done; done
} > gpurun_out/r06_step_ab.txt 2>&1
cat gpurun_out/r06_step_ab.txt

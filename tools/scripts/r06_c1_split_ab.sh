#!/bin/bash
# A/B of the first block's split form (TMF_C1_SPLIT 1 | 0): alternating bench processes on one box + the per-kernel times
O=gpurun_out; mkdir -p $O
{
echo "# alternating processes on one box: python bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline; value = volume-pairs/s (ms per step)"
for i in 1 2 3; do
  for v in 1 0; do
    TMF_C1_SPLIT=$v python3 bench.py --steps 30 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('TMF_C1_SPLIT=$v ', d['value'], d['ms_per_step'])"
  done
done
} > $O/r06_c1_split_ab.txt 2>&1
cat $O/r06_c1_split_ab.txt

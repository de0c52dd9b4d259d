#!/bin/bash
# fp16x3 C2 step, three runs; kernel averages of the x3 chains from a kernel trace
for i in 1 2 3; do python bench.py --precision fp16x3 --steps 60 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M')"; done
bash scripts/stats_step.sh fp16x3 2>&1 | grep "x3\|ms_per_step"

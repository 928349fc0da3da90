#!/bin/bash
# kernel time of the headline workload under different per-phase wave priorities (see phase_prio)
for m in 0 33210 33221 32210 33110 33211 33310 32100 33200 33321 23210; do echo -n "MATE_STAGGER=$m "; MATE_STAGGER=$m python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['roofline']['kernel_avg_us'],2))"; done

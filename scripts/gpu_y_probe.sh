#!/bin/bash
# the half-product pair pass (k_pairs_y_probe) next to k_schur_pairs2 under rocprofv3, config 2 (general + spherical) and the configs[4] size
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for mode in general spherical; do
SSFM_PAIRS_Y_PROBE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/yprobe_$mode -o ba -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-probe --mode $mode > $OUT/yprobe_$mode.log 2>&1
python3 -c "
import csv
for r in csv.DictReader(open('$OUT/yprobe_$mode/ba_kernel_stats.csv')):
    if 'pairs' in r['Name']: print('$mode', r['Name'].split('(')[0], 'avg us', float(r['AverageNs'])/1e3)"
done
SSFM_PAIRS_Y_PROBE=1 CHECK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/yprobe_scale -o ba -- python3 scripts/dev/scale.py > $OUT/yprobe_scale.log 2>&1
python3 -c "
import csv
for r in csv.DictReader(open('$OUT/yprobe_scale/ba_kernel_stats.csv')):
    if 'pairs' in r['Name'] and 'lists' not in r['Name']: print('configs4-size', r['Name'].split('(')[0], 'avg us', float(r['AverageNs'])/1e3)"
SSFM_PAIRS_Y_PROBE=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/yprobe_pmc -o ba -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/yprobe_pmc.log 2>&1
python3 -c "
import csv, collections
a=collections.defaultdict(list)
for r in csv.DictReader(open('$OUT/yprobe_pmc/ba_counter_collection.csv')):
    if 'pairs' in r['Kernel_Name'] and 'lists' not in r['Kernel_Name']: a[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
for k,v in a.items(): print(k, 'FETCH_SIZE x2 MB per launch', 2*sum(v)/len(v)/1024)"

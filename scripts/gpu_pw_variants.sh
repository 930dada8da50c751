cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for opts in "" "final_least_squares=0" "final_least_squares=0,lo_starting_iterations=1000000" "min_num_iterations=200,final_least_squares=0,lo_starting_iterations=1000000"; do
SSFM_PW_OPTS=$opts rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pwv -o pw -- python3 scripts/bench_pairwise.py 100000 100000 1 > gpurun_out/pwv.log 2>&1
python3 -c "
import csv
r=[x for x in csv.DictReader(open('gpurun_out/pwv/pw_kernel_stats.csv')) if 'lomsac' in x['Name']][0]
print('opts=[$opts]', 'calls', r['Calls'], 'total ms', float(r['TotalDurationNs'])/1e6)"
done

#!/bin/bash
# kernel trace of the default bench command with the side streams live (final tree)
OUT=gpurun_out/r4tr; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof -o vitb16 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $R/$OUT/bench.json 2> $R/$OUT/err.log)
cat $OUT/bench.json | cut -c1-200
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4tr/kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'unfold' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
w=csv.DictWriter(open('gpurun_out/r4tr/kernel_trace_last_step.csv','w',newline=''), fieldnames=list(rows[0].keys()))
w.writeheader()
for r in rows[a:b+1]: w.writerow(r)
PY
rm -rf $OUT/prof $OUT/kernel_trace.csv; ls -la $OUT

#!/bin/bash
# kernel timeline of the two-stream step (who runs when): 16-frame encoder passes, with and without the workgroup caps
# (under rocprofv3 the small decode kernels slow down more than the big ones: tools/stream_phases.py gives the undisturbed picture)
R=$GRAFT_REPO_ROOT; S=$R/gpurun_out/r5c; mkdir -p $S
cd /tmp && export TMPDIR=/tmp
for v in base caps; do
  if [ $v = caps ]; then CAPS=256,256,224,224; else CAPS=off; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -o x -- python3 $R/bench.py --steps 2 --warmup 1 --sam-chunk 16 --sam-caps $CAPS --no-cpu-baseline --no-parity --no-b1 > $S/tl_$v.txt 2>&1 || exit 1
  python3 - "$(find /tmp/tl_$v -name '*kernel_trace.csv' | head -1)" $S/tl_$v.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# keep the columns needed, shorten names
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f)
    w.writerow(['queue', 'start', 'end', 'grid', 'name'])
    t0 = min(int(r['Start_Timestamp']) for r in rows)
    for r in rows:
        n = r['Kernel_Name']
        n = n.replace('(anonymous namespace)::', '').replace('void ', '')[:60]
        w.writerow([r['Queue_Id'], int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), n])
P
done

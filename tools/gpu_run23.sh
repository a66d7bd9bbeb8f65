#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
export MARAY_CACHE_DIR=/tmp/mc
cd /tmp && export TMPDIR=/tmp
for rows in 8 512 4096; do
  rm -rf /tmp/fl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl -- python3 $GRAFT_REPO_ROOT/tools/exp_floor.py $rows 60 > /dev/null 2>&1
  python3 - $rows <<'PY'
import csv, glob, sys
f=glob.glob('/tmp/fl/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'maray_jit_rows' in r['Kernel_Name'] or 'maray_jit_pixels' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-80:]
dur={'rows':[], 'pixels':[]}; gap={'r->p':[], 'p->r':[]}
prev=None
for r in rows:
    k='rows' if 'rows' in r['Kernel_Name'] else 'pixels'
    dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    if prev is not None:
        g=int(r['Start_Timestamp'])-int(prev['End_Timestamp'])
        gap['r->p' if k=='pixels' else 'p->r'].append(g)
    prev=r
med=lambda v: sorted(v)[len(v)//2] if v else None
print(sys.argv[1],'rows: kernel ns median rows',med(dur['rows']),'pixels',med(dur['pixels']),'gap r->p',med(gap['r->p']),'gap p->r',med(gap['p->r']))
PY
done

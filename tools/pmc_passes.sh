#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01c/pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --reads 100000 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -- $CMD > $O/a.json 2> $O/a.err
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/b -- $CMD > $O/b.json 2> $O/b.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c -- $CMD > $O/c.json 2> $O/c.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/d -- $CMD > $O/d.json 2> $O/d.err
cd $R
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r01c/pmc'
out=open(O+'/summary.csv','w')
out.write('pass,kernel,counter,dispatches,sum\n')
for p in 'abcd':
    for f in glob.glob(O+'/%s/**/*counter_collection.csv'%p, recursive=True):
        agg=collections.defaultdict(float); nd=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'bt2g' not in k: continue
            agg[(k,r['Counter_Name'])]+=float(r['Counter_Value']); nd[(k,r['Counter_Name'])].add(r['Dispatch_Id'])
        for (k,c),v in sorted(agg.items()):
            out.write('%s,%s,%s,%d,%.0f\n'%(p,k,c,len(nd[(k,c)]),v))
out.close()
print(open(O+'/summary.csv').read())
PY
find $O -name "*.csv" -size +2M -delete
tail -2 $O/d.err

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/b1_prof -o lay -- python3 $R/tools/layer_profile.py run --plan $O/b1_plan.json --batch 1 --iters 5 > $O/b1_prof.log 2>&1
python3 $R/tools/layer_profile.py report --plan $O/b1_plan.json --trace $O/b1_prof/lay_kernel_trace.csv > $O/b1_layers.md
python3 - <<'P'
import csv,os,collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
rows=list(csv.DictReader(open(O+'/b1_prof/lay_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last forward: take the last N kernels between two stem kernels
idx=[i for i,r in enumerate(rows) if 'stem_conv1' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
seg=rows[a:b]
t0=int(seg[0]['Start_Timestamp']); t1=int(rows[b]['Start_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
print('kernels',len(seg),'span us',(t1-t0)/1e3,'busy us',busy/1e3)
agg=collections.defaultdict(lambda:[0,0])
for r in seg:
    n=r['Kernel_Name'].split('(')[0][:70]; agg[n][0]+=1; agg[n][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for n,(c,t) in sorted(agg.items(),key=lambda x:-x[1][1])[:40]: print(f'{t/1e3:9.1f} us {c:4d}  {n}')
gaps=[(int(seg[i+1]['Start_Timestamp'])-int(seg[i]['End_Timestamp']))/1e3 for i in range(len(seg)-1)]
print('sum gaps us',sum(g for g in gaps if g>0),'neg overlap',sum(g for g in gaps if g<0))
P

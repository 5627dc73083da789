cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/gap -o lay -- python3 $R/tools/layer_profile.py run --plan $O/gap_plan.json --compute-dtype 2 --height 1024 --width 1024 --batch 8 --iters 3 > $O/gap.log 2>&1
python3 - <<'P'
import csv,os,re
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
rows=list(csv.DictReader(open(O+'/gap/lay_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'stem_conv1' in r['Kernel_Name']]
seg=rows[idx[-2]:idx[-1]]
def short(n): return re.sub(r'\(.*$','',n.replace('void quber::','').replace('(anonymous namespace)::',''))[:50]
for i in range(1,len(seg)):
    gap=(int(seg[i]['Start_Timestamp'])-int(seg[i-1]['End_Timestamp']))/1e3
    if gap>8: print(f"{gap:7.1f} us before {short(seg[i]['Kernel_Name'])} (after {short(seg[i-1]['Kernel_Name'])}) scratch {seg[i]['Scratch_Size']}")
print('total gaps us', sum(max(0,(int(seg[i]['Start_Timestamp'])-int(seg[i-1]['End_Timestamp']))/1e3) for i in range(1,len(seg))), 'kernels', len(seg), 'span ms', (int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e6)
P
rm -rf $O/gap

import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=len(rows); steps=int(sys.argv[2])
tot=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
t0=min(int(r['Start_Timestamp']) for r in rows); t1=max(int(r['End_Timestamp']) for r in rows)
print("launches",n,"per step ~",n/steps,"sum dur ms/step",tot/1e6/steps,"span ms/step",(t1-t0)/1e6/steps)
from collections import defaultdict
d=defaultdict(lambda:[0,0])
for r in rows:
    k=r['Kernel_Name'][:70]; d[k][0]+=1; d[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,(c,t) in sorted(d.items(),key=lambda x:-x[1][1])[:25]:
    print("%-70s %7.1f/step %8.3f ms/step %6.1f us"%(k,c/steps,t/1e6/steps,t/1e3/c))

# the GEMM launches by grid: (kernel, Grid_Size_X/Y/Z) -> count and mean duration
if "--grids" in sys.argv:
    gd = defaultdict(lambda: [0, 0])
    for r in rows:
        if "sgemm_kernel" not in r['Kernel_Name'] and "--all" not in sys.argv:
            continue
        k = (r['Kernel_Name'][35:95] if 'sgemm_kernel' in r['Kernel_Name'] else r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
        gd[k][0] += 1; gd[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for k, (c, t) in sorted(gd.items(), key=lambda x: -x[1][1])[:40]:
        print("%-60s grid %6s %4s %3s  %6.2f/step %7.1f us  %7.3f ms/step" % (k[0], k[1], k[2], k[3], c / steps, t / 1e3 / c, t / 1e6 / steps))

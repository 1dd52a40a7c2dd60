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

"""One optimizer step of a rocprofv3 kernel trace, launch by launch (steps are delimited by adam_multi_kernel).
usage: python tools/exp/trace_step.py <trace dir> [step index]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_multi' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = int(rows[a + 1]['Start_Timestamp'])
busy = 0
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    s = int(r['Start_Timestamp']) - t0
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    busy += d
    g = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    print("%8.1f us %7.1f us  wg%6d  %s" % (s / 1e3, d / 1e3, g, n[:110]))
print("launches %d  span %.1f us  busy %.1f us" % (b - a, (int(rows[b]['End_Timestamp']) - t0) / 1e3, busy / 1e3))

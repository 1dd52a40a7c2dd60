"""Timeline of the LAST step of a rocprofv3 kernel trace: run-length groups of consecutive launches of one kernel with
their start offset, span and summed duration.   python tools/exp/trace_timeline.py <dir> <launches_per_step>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2])
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
groups = []
for r in rows:
    name = r['Kernel_Name']
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')[:int(sys.argv[3]) if len(sys.argv) > 3 else 64]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if groups and groups[-1][0] == name:
        g = groups[-1]; g[2] = e; g[3] += e - s; g[4] += 1
    else:
        groups.append([name, s, e, e - s, 1])
for name, s, e, d, c in groups:
    print("%9.3f ms  span %8.3f  busy %8.3f  x%-4d %s" % (s / 1e6, (e - s) / 1e6, d / 1e6, c, name))
print("step span %.3f ms" % ((int(rows[-1]['End_Timestamp']) - t0) / 1e6))

"""Per-step summary of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> <steps incl. warm-up>"""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    short = r['Name'].split('(')[0][:64]
    print(f"{short:66s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:9.1f} us  per-step {float(r['TotalDurationNs'])/steps/1e3:9.1f} us")

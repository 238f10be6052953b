import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_pull_periodic_vec"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if name in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1_000_000]
t0 = int(big[5]["Start_Timestamp"])
for r in rows:
    s = int(r["Start_Timestamp"]) - t0
    e = int(r["End_Timestamp"]) - t0
    if -100_000 <= s < 6_700_000:
        print("%10.1f %10.1f us  dur %8.1f  q%s s%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r.get("Stream_Id"), r["Kernel_Name"][:50]))

"""Prints the tail of a rocprofv3 --memory-copy-trace CSV next to the kernels that ran beside the copies:
which engine moved the bytes, when, and how long each copy took.  (A diagnostic for the host pipeline; not part of the round check.)"""
import csv, glob, sys
d = sys.argv[1]
cp = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
kn = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?"), r.get("Bytes", r.get("Size", ""))) for r in cp]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], "") for r in kn]
ev.sort()
t0 = ev[0][0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for s, e, what, b in ev[-last:]:
    if (e - s) < 200000 and not what.startswith("copy"): continue
    print(f"{(s - t0) / 1e6:10.2f} ms  +{(e - s) / 1e6:7.2f} ms  {what} {b}")

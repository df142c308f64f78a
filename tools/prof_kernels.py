"""usage: prof_kernels.py <dir with stats/*kernel_stats.csv> <comma list of name fragments>
per-step launch count and average duration of the matching kernels of a rocprofv3 --kernel-trace --stats --output-format csv run of
bench.py (steps are counted by the adamw_kernel launches)."""
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/stats/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
n=[int(r["Calls"]) for r in rows if "adamw" in r["Name"]][0]
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2].split(",")):
        print(f'{float(r["AverageNs"])/1e3:8.1f} us x{int(r["Calls"])/n:4.1f}  {r["Name"][:100]}')
print("kernel ms/step", round(sum(float(r["TotalDurationNs"]) for r in rows)/n/1e6,3))

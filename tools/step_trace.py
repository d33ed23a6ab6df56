"""Developer view: the kernel sequence of ONE replayed denoise step from a rocprofv3 --kernel-trace CSV.
usage: python tools/step_trace.py <..._kernel_trace.csv>   (the step = from one conv_thin_kernel launch to the next)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "conv_thin_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]                       # a late step (graph replay)
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n)
    return n[:90]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = short(r["Kernel_Name"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.2f}  dur {(e - s) / 1e3:7.2f}  {name}")
    prev_end = e
    k = tot.setdefault(name, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
print("---- totals")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{us:9.1f} us  x{c:4d}  {n}")
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} kernels")

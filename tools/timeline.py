"""one training step from a rocprofv3 kernel trace: per launch start offset, duration, queue; union times per class"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import os
MARK = os.environ.get("TIMELINE_FIRST_KERNEL", "void conv_first_fwd_kernel")      # the first launch of a step
firsts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(MARK)]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a, b = firsts[-k], firsts[-k + 1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
def cls(n):
    if "conv_band_kernel" in n or "conv_stream" in n or "conv_gemm_kernel" in n: return "M"
    if "conv_wgrad_kernel" in n or "conv_wgrad_col_kernel" in n or "conv_wgrad1_wide_kernel" in n: return "W"
    return "h"
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
qs = {}
iv = {"M": [], "W": [], "h": []}
for r in step:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = qs.setdefault(r["Queue_Id"], len(qs))
    c = cls(r["Kernel_Name"])
    iv[c].append((s, e))
    if len(sys.argv) > 2: print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} q{q} {c} {n} grid {r['Grid_Size_X']}")
end = max(int(r["End_Timestamp"]) for r in step) - t0
print(f"step span {end/1e3:.1f} us; union MFMA-main {union(iv['M'])/1e3:.1f}, wgrad {union(iv['W'])/1e3:.1f}, "
      f"any MFMA {union(iv['M']+iv['W'])/1e3:.1f}, other {union(iv['h'])/1e3:.1f}, anything {union(iv['M']+iv['W']+iv['h'])/1e3:.1f}")

"""Gaps between the decode-step launches inside one full(): reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv) and prints, for the
longest run of consecutive k_decode_mega launches, the sum of their durations, the span, and what sits in the gaps."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs, cur = [], []
for r in rows:
    if "k_decode_mega" in r["Kernel_Name"]:
        cur.append(r)
    elif cur and len(cur) > 50 and ("k_gemm" in r["Kernel_Name"] or "k_mel" in r["Kernel_Name"]):
        runs.append(cur); cur = []
if cur: runs.append(cur)
runs = [x for x in runs if len(x) > 150]
for run in runs[-2:]:
    st = [int(r["Start_Timestamp"]) for r in run]; en = [int(r["End_Timestamp"]) for r in run]
    dur = sum(e - s for s, e in zip(st, en)); span = en[-1] - st[0]
    gaps = [st[i + 1] - en[i] for i in range(len(run) - 1)]
    gaps_s = sorted(gaps)
    print("%d launches: kernels %.2f ms, span %.2f ms, gaps %.2f ms (median %.1f us, p90 %.1f us, max %.1f us); mean kernel %.1f us" % (
        len(run), dur / 1e6, span / 1e6, (span - dur) / 1e6, gaps_s[len(gaps) // 2] / 1e3, gaps_s[int(len(gaps) * 0.9)] / 1e3, gaps_s[-1] / 1e3, dur / len(run) / 1e3))

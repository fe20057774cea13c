#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel statistics of the headline bench, and the HBM traffic counters of the
# decode step (separate --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Summaries land in gpurun_out/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-concurrent --no-second-path --steps 3 --json-out $OUT/bench_line.json > $OUT/stats.out 2> $OUT/stats.err || true
find $OUT/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/tools/decode_probe.py small 20 64 0 > $OUT/pmc_$c.log 2>&1 || true
done
python3 - <<PY
import csv, glob, json, os
out = "$OUT"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_decode_mega" in row.get("Kernel_Name", "") and row.get("Counter_Name") == c:
                vals.append(float(row["Counter_Value"]))
    res[c] = {"launches": len(vals), "mean": sum(vals) / len(vals) if vals else None, "min": min(vals) if vals else None, "max": max(vals) if vals else None}
# the form bench.py reads (copy to profiles/r02_decode_step_pmc.json)
summary = {"kernel": "k_decode_mega (ggml-small shape, 1 token, n_past = 64)",
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/decode_probe.py small 20 64 0 (tools/profile_gpu.sh; one counter per pass)",
           "FETCH_SIZE_KB_per_launch": res["FETCH_SIZE"]["mean"], "WRITE_SIZE_KB_per_launch": res["WRITE_SIZE"]["mean"], "launches": res["FETCH_SIZE"]["launches"],
           "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE x 1024 for wide streams"}
json.dump(summary, open(os.path.join(out, "decode_step_pmc.json"), "w"), indent=1)
print(json.dumps(res))
PY
head -12 $OUT/kernel_stats.csv | cut -c1-200
tail -1 $OUT/bench_line.json | cut -c1-400

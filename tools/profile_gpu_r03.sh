#!/bin/bash
# Round-3 profiles (runs on the GPU box under gpurun): rocprofv3 kernel statistics and HBM traffic counters (one counter per pass, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; the counter passes carry --kernel-trace only).  Summaries land in gpurun_out/prof3/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$ROOT/tools'); import wsynth; wsynth.model_path('small'); wsynth.quant_model_path('small', 'q5_0')"
echo "== kernel stats: the headline bench"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-concurrent --no-second-path --no-configs --steps 3 --json-out $OUT/bench_line.json > $OUT/stats.out 2> $OUT/stats.err || true
find $OUT/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
echo "== kernel stats: the several-rows step"; date
for cfg in "8 chunks" "5 beams"; do set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rows_$1 -- python3 $ROOT/tools/rows_probe.py small $1 $2 30 110 > $OUT/rows_$1.log 2>&1 || true
  find $OUT/rows_$1 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/rows_$1_kernel_stats.csv
done
echo "== kernel stats: quantised model (encoder + one-row step + 5-row step)"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quant -- python3 $ROOT/tools/decode_probe.py small:q5_0 30 64 0 > $OUT/quant.log 2>&1 || true
find $OUT/quant -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/quant_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quant_rows -- python3 $ROOT/tools/rows_probe.py small:q5_0 5 beams 30 64 > $OUT/quant_rows.log 2>&1 || true
find $OUT/quant_rows -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/quant_rows_kernel_stats.csv
echo "== HBM traffic counters"; date
for rep in 1 2 3; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_mega_${c}_$rep -- python3 $ROOT/tools/decode_probe.py small 20 64 0 > $OUT/pmc_mega_${c}_$rep.log 2>&1 || true
done; done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_rows8_$c -- python3 $ROOT/tools/rows_probe.py small 8 chunks 10 110 > $OUT/pmc_rows8_$c.log 2>&1 || true
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_rows5_$c -- python3 $ROOT/tools/rows_probe.py small 5 beams 10 110 > $OUT/pmc_rows5_$c.log 2>&1 || true
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_quant_$c -- python3 $ROOT/tools/decode_probe.py small:q5_0 20 64 0 > $OUT/pmc_quant_$c.log 2>&1 || true
done
python3 - <<PY
import csv, glob, json, os, statistics
out = "$OUT"
def collect(dirpat, kernel, counter):
    vals = []
    for f in glob.glob(os.path.join(out, dirpat, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    return vals
res = {}
reps = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    reps[c] = []
    for rep in (1, 2, 3):
        v = collect("pmc_mega_%s_%d" % (c, rep), "k_decode_mega", c)
        if v: reps[c].append(sum(v) / len(v))
mega = {"kernel": "k_decode_mega (ggml-small shape, 1 token, n_past = 64)",
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/decode_probe.py small 20 64 0 (tools/profile_gpu_r03.sh; one counter per pass, three collections)",
        "FETCH_SIZE_KB_per_launch": statistics.median(reps["FETCH_SIZE"]) if reps["FETCH_SIZE"] else None,
        "WRITE_SIZE_KB_per_launch": statistics.median(reps["WRITE_SIZE"]) if reps["WRITE_SIZE"] else None,
        "FETCH_SIZE_KB_collections": reps["FETCH_SIZE"], "WRITE_SIZE_KB_collections": reps["WRITE_SIZE"],
        "traffic_MB_min_median_max": None,
        "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE x 1024 for wide streams"}
if len(reps["FETCH_SIZE"]) == 3 and len(reps["WRITE_SIZE"]) == 3:
    t = sorted((2 * f + w) * 1024 / 1e6 for f, w in zip(sorted(reps["FETCH_SIZE"]), sorted(reps["WRITE_SIZE"])))
    mega["traffic_MB_min_median_max"] = [round(t[0], 1), round(t[1], 1), round(t[2], 1)]
json.dump(mega, open(os.path.join(out, "decode_step_pmc.json"), "w"), indent=1)
rows = {}
for tag, kern in (("rows8", "k_decode_rows_np12"), ("rows5", "k_decode_rows_np12"), ("quant", "k_decode_mega_q")):
    f = collect("pmc_%s_FETCH_SIZE" % tag, kern, "FETCH_SIZE"); w = collect("pmc_%s_WRITE_SIZE" % tag, kern, "WRITE_SIZE")
    rows[tag] = {"kernel": kern, "launches": len(f), "FETCH_SIZE_KB_per_launch": sum(f) / len(f) if f else None, "WRITE_SIZE_KB_per_launch": sum(w) / len(w) if w else None,
                 "traffic_MB_per_launch": round((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6, 1) if f and w else None}
json.dump(rows, open(os.path.join(out, "rows_and_quant_pmc.json"), "w"), indent=1)
print(json.dumps(mega)[:600]); print(json.dumps(rows))
PY
for f in bench_kernel_stats rows_8_kernel_stats rows_5_kernel_stats quant_kernel_stats quant_rows_kernel_stats; do echo "-- $f"; head -8 $OUT/$f.csv | cut -c1-180; done
tail -1 $OUT/bench_line.json | cut -c1-300
date

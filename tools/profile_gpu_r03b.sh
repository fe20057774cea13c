#!/bin/bash
# Round-3 profiles at the final code state (runs on the GPU box under gpurun).  Kernel statistics, the HBM counters of the single-token step (three collections), the HBM counters of the several-rows kernel (one counter per pass, --kernel-trace
# only), the kernel timeline of a lock-step group and the in-kernel timelines.  Summaries land in gpurun_out/prof3b/ and are copied to profiles/ by hand.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof3b
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$ROOT/tools'); import wsynth; wsynth.model_path('small'); wsynth.quant_model_path('small', 'q5_0')"
echo "== kernel stats: the headline bench"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-concurrent --no-second-path --no-configs --steps 3 --json-out $OUT/bench_line.json > $OUT/stats.out 2> $OUT/stats.err || true
find $OUT/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
echo "== kernel stats: the several-rows step, F16 and quantised"; date
for cfg in "8 chunks 110" "5 beams 64"; do set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rows_$1 -- python3 $ROOT/tools/rows_probe.py small $1 $2 30 $3 > $OUT/rows_$1.log 2>&1 || true
  find $OUT/rows_$1 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/rows_$1_kernel_stats.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quant_rows -- python3 $ROOT/tools/rows_probe.py small:q5_0 5 beams 30 64 > $OUT/quant_rows.log 2>&1 || true
find $OUT/quant_rows -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/quant_rows_kernel_stats.csv
echo "== HBM traffic counters of the single-token step (three collections)"; date
for rep in 1 2 3; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_mega_${c}_$rep -- python3 $ROOT/tools/decode_probe.py small 20 64 0 > $OUT/pmc_mega_${c}_$rep.log 2>&1 || true
done; done
echo "== HBM traffic counters of the several-rows kernel"; date
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_rows8_$c -- python3 $ROOT/tools/rows_probe.py small 8 chunks 10 110 > $OUT/pmc_rows8_$c.log 2>&1 || true
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_rows5_$c -- python3 $ROOT/tools/rows_probe.py small 5 beams 10 64 > $OUT/pmc_rows5_$c.log 2>&1 || true
done
python3 - <<PY
import csv, glob, json, os
out = "$OUT"
def collect(dirpat, kernel, counter):
    vals = []
    for f in glob.glob(os.path.join(out, dirpat, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    return vals
import statistics
reps = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    reps[c] = []
    for rep in (1, 2, 3):
        v = collect("pmc_mega_%s_%d" % (c, rep), "k_decode_mega", c)
        if v: reps[c].append(sum(v) / len(v))
mega = {"kernel": "k_decode_mega (ggml-small shape, 1 token, n_past = 64)",
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/decode_probe.py small 20 64 0 (tools/profile_gpu_r03b.sh; one counter per pass, three collections)",
        "FETCH_SIZE_KB_per_launch": statistics.median(reps["FETCH_SIZE"]) if reps["FETCH_SIZE"] else None,
        "WRITE_SIZE_KB_per_launch": statistics.median(reps["WRITE_SIZE"]) if reps["WRITE_SIZE"] else None,
        "FETCH_SIZE_KB_collections": reps["FETCH_SIZE"], "WRITE_SIZE_KB_collections": reps["WRITE_SIZE"],
        "traffic_MB_min_median_max": None,
        "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE x 1024 for wide streams"}
if len(reps["FETCH_SIZE"]) == 3 and len(reps["WRITE_SIZE"]) == 3:
    t = sorted((2 * f + w) * 1024 / 1e6 for f, w in zip(sorted(reps["FETCH_SIZE"]), sorted(reps["WRITE_SIZE"])))
    mega["traffic_MB_min_median_max"] = [round(t[0], 1), round(t[1], 1), round(t[2], 1)]
json.dump(mega, open(os.path.join(out, "decode_step_pmc.json"), "w"), indent=1)
print(json.dumps(mega)[:500])
rows = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/rows_probe.py small 8 chunks 10 110 | small 5 beams 10 64 (tools/profile_gpu_r03b.sh)",
        "note": "gfx950: read bytes = 2 x FETCH_SIZE x 1024 for wide streams (MI355X_MICROARCH.md, HBM); traffic = that + WRITE_SIZE x 1024, per launch"}
for tag in ("rows8", "rows5"):
    f = collect("pmc_%s_FETCH_SIZE" % tag, "k_decode_rows_np12", "FETCH_SIZE"); w = collect("pmc_%s_WRITE_SIZE" % tag, "k_decode_rows_np12", "WRITE_SIZE")
    rows[tag] = {"kernel": "k_decode_rows_np12", "launches": len(f), "FETCH_SIZE_KB_per_launch": sum(f) / len(f) if f else None, "WRITE_SIZE_KB_per_launch": sum(w) / len(w) if w else None,
                 "traffic_MB_per_launch": round((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6, 1) if f and w else None}
json.dump(rows, open(os.path.join(out, "rows_pmc.json"), "w"), indent=1)
print(json.dumps(rows))
PY
echo "== kernel timeline of a lock-step group of 8 chunks"; date
rocprofv3 --kernel-trace --output-format csv -d $OUT/c8 -- python3 $ROOT/tools/chunks8_probe.py small 8 3 > $OUT/c8.log 2>&1 || true
( echo "# rocprofv3 --kernel-trace -- python3 tools/chunks8_probe.py small 8 3 ; python3 tools/chunks8_probe.py - - - <trace dir>   (tools/profile_gpu_r03b.sh)"; grep "^rep" $OUT/c8.log; python3 $ROOT/tools/chunks8_probe.py - - - $OUT/c8 ) > $OUT/chunks8_timeline.txt 2>&1 || true
rm -rf $OUT/c8 $OUT/stats $OUT/rows_8 $OUT/rows_5 $OUT/quant_rows $OUT/pmc_rows8_FETCH_SIZE $OUT/pmc_rows8_WRITE_SIZE $OUT/pmc_rows5_FETCH_SIZE $OUT/pmc_rows5_WRITE_SIZE $OUT/pmc_mega_*_[123]
echo "== in-kernel timelines"; date
cd $ROOT
( echo "# WHISPER_AMD_ROWS_TRACE=0 python3 tools/rows_trace.py small 8 110"; WHISPER_AMD_ROWS_TRACE=0 python3 tools/rows_trace.py small 8 110; echo; echo "# WHISPER_AMD_ROWS_TRACE=0 python3 tools/rows_trace.py small 5 64"; WHISPER_AMD_ROWS_TRACE=0 python3 tools/rows_trace.py small 5 64 ) > $OUT/rows_trace.txt 2>&1 || true
( echo "# python3 tools/mega_trace.py small 64"; python3 tools/mega_trace.py small 64 ) > $OUT/mega_trace.txt 2>&1 || true
for f in bench_kernel_stats rows_8_kernel_stats rows_5_kernel_stats quant_rows_kernel_stats; do echo "-- $f"; head -5 $OUT/$f.csv | cut -c1-160; done
head -12 $OUT/chunks8_timeline.txt; tail -5 $OUT/rows_trace.txt; tail -5 $OUT/mega_trace.txt
tail -1 $OUT/bench_line.json | cut -c1-300
date

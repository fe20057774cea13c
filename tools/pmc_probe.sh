#!/bin/bash
# usage: tools/pmc_probe.sh "COUNTER1 COUNTER2 ..."   (one rocprofv3 --pmc pass over the decode-step probe; prints per-launch means for k_decode_mega)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_probe; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/decode_probe.py small 20 64 0 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_decode_mega" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items(): print("%-28s mean %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY

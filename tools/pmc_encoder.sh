#!/bin/bash
# usage: tools/pmc_encoder.sh "COUNTER1 COUNTER2 ..."   (one rocprofv3 --pmc pass over the encoder probe; per-launch means per kernel)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_enc_${2:-a}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/encode_probe.py small ${3:-1} 2 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()): print("%-42s %-22s mean %.5g  (n=%d)" % (k[0], k[1], sum(v) / len(v), len(v)))
PY

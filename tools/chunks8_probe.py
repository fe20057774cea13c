"""Eight 30 s chunks transcribed together on one device (whisper_amd_full_batch: the lock-step group of bench.py's `concurrent_chunks`), alone -
for a kernel trace of that job:   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/chunks8_probe.py [model=small] [chunks=8] [reps=2]
With a trace directory as 4th argument instead (no GPU needed):  python3 tools/chunks8_probe.py - - - <dir>  prints the timeline of the rows kernel
in the LAST repetition: launches, mean duration, mean gap between consecutive launches, and what ran before the first of them."""
import csv, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))


def timeline(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    pat = os.environ.get("PROBE_KERNEL", "k_decode_rows")
    is_rows = [i for i, r in enumerate(rows) if pat in r[2]]
    if not is_rows:
        print("no", pat, "launch in", d); return
    # the last repetition = the last run of rows launches with gaps < 20 ms between them
    last = [is_rows[-1]]
    for i in reversed(is_rows[:-1]):
        if rows[last[0]][0] - rows[i][1] > 20e6: break
        last.insert(0, i)
    dur = [(rows[i][1] - rows[i][0]) / 1e3 for i in last]
    gap = [(rows[b][0] - rows[a][1]) / 1e3 for a, b in zip(last, last[1:])]
    print("rows-kernel launches in the last repetition: %d; duration mean %.1f us (min %.1f, max %.1f); gap between consecutive launches mean %.1f us (median %.1f, max %.1f)"
          % (len(last), sum(dur) / len(dur), min(dur), max(dur), sum(gap) / max(1, len(gap)), sorted(gap)[len(gap) // 2] if gap else 0, max(gap) if gap else 0))
    big = sorted(range(len(gap)), key=lambda i: gap[i])[-8:]
    print("largest gaps (us) and the pass they follow:", ["%.0f after pass %d" % (gap[i], i) for i in sorted(big)])
    # everything the device ran across three consecutive passes in the middle of the repetition
    mid = last[len(last) // 2]
    t_ref = rows[mid][0]
    print("around three passes in the middle (us relative to the first one's start):")
    for s_, e_, n_ in rows[mid:]:
        if s_ > rows[last[min(len(last) - 1, len(last) // 2 + 3)]][0]: break
        print("  %9.1f .. %9.1f  (%7.1f)  %s" % ((s_ - t_ref) / 1e3, (e_ - t_ref) / 1e3, (e_ - s_) / 1e3, n_.split("(")[0][:70]))
    # what ran in the 80 ms before the first launch of the repetition
    t0 = rows[last[0]][0]
    by = {}
    first = None
    for s, e, n in rows:
        if t0 - 80e6 <= s < t0:
            first = s if first is None else first
            k = n.split("(")[0][:60]
            by.setdefault(k, [0, 0.0]); by[k][0] += 1; by[k][1] += (e - s) / 1e3
    print("before the first pass (80 ms window; first kernel %.1f ms ahead of it):" % ((t0 - (first or t0)) / 1e6))
    for k, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-62s %5d launches %9.1f us" % (k, n, us))
    print("  total kernel time in that window: %.1f ms" % (sum(v[1] for v in by.values()) / 1e3))


if len(sys.argv) > 4:
    timeline(sys.argv[4]); sys.exit(0)

import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)
fp = W.FullParams(lib, best_of=1, temperature_inc=0.0, language="en", no_context=True)
tst = [ctx.create_state() for _ in range(NB)]
pcm = [wsynth.synth_audio(480000, 100 + i) for i in range(NB)]
for r in range(reps):
    t1 = time.perf_counter()
    W.full_batch(ctx, tst, fp, pcm)
    dt = time.perf_counter() - t1
    print("rep %d: %d chunks in %.1f ms = %.1fx real time" % (r, NB, 1e3 * dt, 30.0 * NB / dt), flush=True)

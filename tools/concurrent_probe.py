"""Aggregate real-time factor of n chunks transcribed concurrently on one GPU (whisper_amd_full_batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
sys.path.insert(0, ROOT)
from bench import Hip
hip = Hip(); hip.set_device(0)
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path("small"), W.WhisperContextParameters(lib, flash_attn=True), lib=lib)
fp = W.FullParams(lib, 0, best_of=1, temperature_inc=0.0)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 12, 16]:
    sts = [ctx.create_state() for _ in range(n)]
    pcm = [wsynth.synth_audio(480000, 100 + i) for i in range(n)]
    for kind, args in (("host PCM", pcm), ("device PCM", [(hip.to_device(p_), 480000) for p_ in pcm])):
        W.full_batch(ctx, sts, fp, args)
        hip.sync(); t = time.perf_counter(); W.full_batch(ctx, sts, fp, args); hip.sync(); dt = time.perf_counter() - t
        print("%2d chunks, %-10s: %.1f ms, aggregate RTF %.1f" % (n, kind, 1e3 * dt, 30.0 * n / dt))
    for s in sts: s.free()

"""Chunks transcribed together on one GPU (whisper_amd_full_batch): aggregate real-time factor by lock-step group size.
usage: concurrent_probe.py [n_chunks] ; WHISPER_AMD_BATCH_GROUP / WHISPER_AMD_NO_BATCHER select the mode (read per call)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path("small"), W.WhisperContextParameters(lib), lib=lib)
fp = W.FullParams(lib, best_of=1, temperature_inc=0.0, language="en", no_context=True)
sts = [ctx.create_state() for _ in range(n)]
pcm = [wsynth.synth_audio(480000, 100 + i) for i in range(n)]
W.full_batch(ctx, sts, fp, pcm)
t = time.perf_counter(); W.full_batch(ctx, sts, fp, pcm); dt = time.perf_counter() - t
st_, rw_ = C.c_long(), C.c_long()
lib.whisper_amd_batch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
lib.whisper_amd_batch_stats(ctx.ptr, st_, rw_)
print("%d chunks, group %s%s: %.1f ms -> %.0fx real-time aggregate; %d passes, %.2f rows per pass" % (
    n, os.environ.get("WHISPER_AMD_BATCH_GROUP", "4 (default)"), " (no batcher)" if os.environ.get("WHISPER_AMD_NO_BATCHER") else "", 1e3 * dt, 30.0 * n / dt,
    st_.value, rw_.value / max(1, st_.value)))

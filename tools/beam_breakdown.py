"""Where a beam-search chunk's wall time goes (per-state stage timers): usage: python tools/beam_breakdown.py [model=small] [beam=5]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib), lib=lib)
st = ctx.create_state()
pcm = wsynth.synth_audio(480000, 5)
fp = W.FullParams(lib, 1, beam_size=beam, temperature_inc=0.0)
st.full(fp, pcm)
tm = (C.c_int64 * 12)()
lib.whisper_amd_get_timings_us.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
lib.whisper_amd_reset_timings.argtypes = [C.c_void_p]
lib.whisper_amd_reset_timings(st.ptr)
t0 = time.perf_counter(); st.full(fp, pcm); dt = time.perf_counter() - t0
lib.whisper_amd_get_timings_us(st.ptr, tm)
names = ["t_sample", "t_encode", "t_decode", "t_batchd", "t_prompt", "t_mel", "n_sample", "n_encode", "n_decode", "n_batchd", "n_prompt", "n_fail"]
print("%s beam %d: wall %.1f ms; " % (name, beam, 1e3 * dt) + "  ".join("%s %d" % (n, v) for n, v in zip(names, tm)))
print("rows stats", st.rows_stats())

"""Where one full() call on a 30 s chunk spends its wall time (per-state counters of the library)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd")); sys.path.insert(0, ROOT)
import wsynth, whisper_rs as W
from bench import Hip
hip = Hip(); hip.set_device(0)
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path("small"), W.WhisperContextParameters(lib, flash_attn=True), lib=lib)
st = ctx.create_state()
fp = W.FullParams(lib, best_of=1, temperature_inc=0.0, language="en", no_context=True)
pcm = hip.to_device(wsynth.synth_audio(480000, 0))
lib.whisper_amd_get_timings_us.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
lib.whisper_amd_reset_timings.argtypes = [C.c_void_p]
for _ in range(2): st.full(fp, (pcm, 480000))
lib.whisper_amd_reset_timings(st.ptr)
hip.sync(); t = time.perf_counter(); st.full(fp, (pcm, 480000)); hip.sync(); wall = 1e3 * (time.perf_counter() - t)
tm = (C.c_int64 * 12)(); lib.whisper_amd_get_timings_us(st.ptr, tm)
names = ["sample", "encode", "decode", "batchd", "prompt", "mel"]
print("wall %.2f ms; " % wall + ", ".join("%s %.2f ms (n=%d)" % (names[i], tm[i] / 1e3, tm[6 + i] if i < 5 else 1) for i in range(6)))
print("unaccounted: %.2f ms" % (wall - sum(tm[i] for i in (1, 2, 3, 4, 5)) / 1e3))

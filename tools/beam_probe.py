"""Beam-search transcription timing (config 4's decode mode): full() with beam_size = 5 on one 30 s chunk."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib, flash_attn=False), lib=lib)
st = ctx.create_state()
pcm = wsynth.synth_audio(480000, 0)
fp = W.FullParams(lib, 1, beam_size=5, temperature_inc=0.0)
st.full(fp, pcm)
t = time.perf_counter(); st.full(fp, pcm); dt = time.perf_counter() - t
ntok = sum(len(s["ids"]) for s in st.segments())
print("%s beam 5: %.1f ms per 30 s chunk (RTF %.1f), %d tokens in the result" % (name, 1e3 * dt, 30.0 / dt, ntok))

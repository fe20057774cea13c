"""Multi-row decode step microbenchmark (beam / several decoders): n tokens per call through whisper_decode_with_state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib, flash_attn=True), lib=lib)
st = ctx.create_state(); st.pcm_to_mel(wsynth.synth_audio(480000, 0)); st.encode(0)
tok = [int(lib.whisper_token_sot(ctx.ptr))] * 64
st.decode(tok, 0)                      # 64 cells of context
st.decode(tok[:n], 64)
t = time.perf_counter()
for _ in range(reps): st.decode(tok[:n], 64)
dt = time.perf_counter() - t
print("%s: %d-row decode step %.3f ms (%.4f ms / token)" % (name, n, 1e3 * dt / reps, 1e3 * dt / reps / n))

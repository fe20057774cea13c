"""Encoder microbenchmark: ms per 30 s chunk (conv + encoder + cross K/V), both summation orders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
pcm = wsynth.synth_audio(480000, 0)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for flash in ([True, False] if len(sys.argv) < 3 else [bool(int(sys.argv[2]))]):
    mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)      # "small:q5_0" = the quantised file
    ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib, flash_attn=flash), lib=lib)
    st = ctx.create_state(); st.pcm_to_mel(pcm); st.encode(0)
    t = time.perf_counter()
    for _ in range(iters): st.encode(0)
    print("%s flash_attn=%d: encode %.3f ms / 30 s chunk" % (name, flash, (time.perf_counter() - t) * 1000 / iters))
    st.free(); ctx.free()

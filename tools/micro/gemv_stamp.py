"""Temporary: per-phase cycle stamps of the fused LN+GEMV decode kernel (needs a -DWA_STAMP build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np, wsynth, whisper_rs as W
lib = W.load_library()
lib.whisper_amd_stamp_probe.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong)]
ctx = W.WhisperContext.new_with_params(wsynth.model_path("small"), W.WhisperContextParameters(lib), lib=lib)
st = ctx.create_state(); st.pcm_to_mel(wsynth.synth_audio(480000, 0)); st.encode(0)
st.decode([50258, 50259, 50359], 0)
out = (C.c_ulonglong * 32)()
for rep in range(3):
    lib.whisper_amd_stamp_probe(ctx.ptr, st.ptr, out)
    for k in range(2):
        v = list(out[k * 16:k * 16 + 9]); n = int(v[8]); d = [v[i + 1] - v[i] for i in range(n - 1)]
        print("kernel %d: phases (cycles @100MHz ticks?)" % k, d, "total", v[n - 1] - v[0])

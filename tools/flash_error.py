"""Error of the MFMA path (flash_attn = true) against the reference-order path of the same library: encoder output and greedy ids."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: None)
lib.whisper_amd_get_embd_enc.restype = C.c_int64
lib.whisper_amd_get_embd_enc.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64]
for name in sys.argv[1:] or ["s128", "small"]:
    pcm = wsynth.synth_audio(480000, 0)
    out = {}
    for flash in (False, True):
        ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib, flash_attn=flash), lib=lib)
        d = ctx.model_n_audio_state()
        st = ctx.create_state(); st.pcm_to_mel(pcm); st.encode(0)
        buf = np.zeros(1500 * d, np.float32)
        lib.whisper_amd_get_embd_enc(st.ptr, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
        st.free(); st = ctx.create_state()
        st.full(W.FullParams(lib, 0, best_of=1, temperature_inc=0.0), pcm)
        ids = [t for s in st.segments() for t in s["ids"]]
        out[flash] = (buf, ids)
        st.free(); ctx.free()
    e = np.abs(out[True][0] - out[False][0])
    a, b = out[False][1], out[True][1]
    first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
    print("%s: embd_enc max |err| %.3e rms err %.3e (rms value %.3e); ids %d vs %d, first mismatch %s" %
          (name, e.max(), np.sqrt((e ** 2).mean()), np.sqrt((out[False][0] ** 2).mean()), len(a), len(b), first))

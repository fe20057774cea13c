"""Tolerance path with a reduced audio context (streaming windows): encoder output vs the reference-order path."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: None)
lib.whisper_amd_get_embd_enc.restype = C.c_int64
lib.whisper_amd_get_embd_enc.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64]
for name in sys.argv[1:] or ["s128", "small"]:
    for actx in (50, 64, 100, 257):
        pcm = wsynth.synth_audio(16000 * 2, 0)
        out = {}
        for flash in (False, True):
            ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib, flash_attn=flash), lib=lib)
            d = ctx.model_n_audio_state()
            st = ctx.create_state()
            st.full(W.FullParams(lib, 0, best_of=1, temperature_inc=0.0, audio_ctx=actx, single_segment=True), pcm)
            buf = np.zeros(1500 * d, np.float32)
            n = lib.whisper_amd_get_embd_enc(st.ptr, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
            ids = [t for s in st.segments() for t in s["ids"]]
            out[flash] = (buf[:actx * d].copy(), ids)
            st.free(); ctx.free()
        e = np.abs(out[True][0] - out[False][0])
        print("%s audio_ctx %d: max |err| %.3e rms %.3e finite %s; ids equal %s" % (name, actx, e.max(), np.sqrt((e ** 2).mean()), np.isfinite(out[True][0]).all(), out[True][1] == out[False][1]))

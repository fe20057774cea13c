"""Decode-step microbenchmark for rocprofv3: encode once, then replay the 1-token decoder pass."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n_past = int(sys.argv[3]) if len(sys.argv) > 3 else 64
flash = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)      # "small:q5_0" = the quantised file
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib, flash_attn=flash), lib=lib)
st = ctx.create_state()
import time
st.pcm_to_mel(wsynth.synth_audio(480000, 0)); st.encode(0); t0 = time.perf_counter(); st.encode(0); print("encode: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
lib.whisper_amd_decode_step_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
ms = C.c_float()
for _ in range(3):
    rc = lib.whisper_amd_decode_step_probe(ctx.ptr, st.ptr, n_past, iters, C.byref(ms))
    print("decode step: rc=%d %.4f ms/token (n_past=%d)" % (rc, ms.value, n_past))

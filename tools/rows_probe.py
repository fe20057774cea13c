"""Several-rows decode-step microbenchmark for rocprofv3: encode B chunks, then replay the B-row one-launch step (wa_rows.hip).
usage: python tools/rows_probe.py [model=small] [B=8] [mode=chunks|beams] [iters=30] [n_past=110]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "chunks"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
n_past = int(sys.argv[5]) if len(sys.argv) > 5 else 110
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)
n_st = B if mode == "chunks" else 1
sts = [ctx.create_state() for _ in range(n_st)]
for i, st in enumerate(sts):
    st.pcm_to_mel(wsynth.synth_audio(480000, i)); st.encode(0)
VP = C.c_void_p
lib.whisper_amd_rows_step_probe.argtypes = [VP, C.POINTER(VP), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
arr = (VP * B)(*[VP(sts[i % n_st].ptr) for i in range(B)])
ms = C.c_float(0)
for _ in range(3):
    rc = lib.whisper_amd_rows_step_probe(ctx.ptr, arr, B, n_past, iters, C.byref(ms))
    print("rows step: rc=%d %.4f ms/step (B=%d %s, n_past=%d)" % (rc, ms.value, B, mode, n_past))

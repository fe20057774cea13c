import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W
os.environ["WHISPER_AMD_MEGA_DBG"] = "1"
name = "s128"
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib), lib=lib)
meg = ctx.create_state()
pcm = wsynth.synth_audio(480000, 0)
meg.pcm_to_mel(pcm); meg.encode(0)
sot = ctx.token_sot(); prompt = [sot, sot + 1, sot + 102]
L, d, nv, H = ctx.model_n_text_layer(), ctx.model_n_text_state(), ctx.n_vocab(), ctx.model_n_text_head()
lib.whisper_amd_mega_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.whisper_amd_seq_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
os.environ["WHISPER_AMD_NO_MEGA"] = "1"
meg.decode(prompt, 0)
n_past = 3
for step, tok in enumerate([1000, 2000, 3000]):
    sv = np.zeros(L * 8 * 2 * d, dtype=np.uint32); sl = np.zeros(nv, dtype=np.float32)
    lib.whisper_amd_seq_debug(ctx.ptr, meg.ptr, tok, n_past, sv.ctypes.data, sl.ctypes.data)
    g = np.zeros(L * 8 * 2 * d, dtype=np.uint64); lg = np.zeros(nv, dtype=np.float32)
    lib.whisper_amd_mega_debug(ctx.ptr, meg.ptr, tok, n_past, g.ctypes.data, lg.ctypes.data)
    a = np.fromfile("gpurun_out/seq_dbg.bin", dtype=np.float32).reshape(L, H, 1500)
    bb = np.fromfile("gpurun_out/mega_dbg.bin", dtype=np.float32).reshape(L, H, 5120)
    b = bb[:, :, :3072].reshape(L, H, 2, 1536)
    pm = b[:, :, 1, :1500]
    pa = np.fromfile("gpurun_out/seq_part.bin", dtype=np.float32).reshape(L, H, 32, 64); pb = bb[:, :, 3072:].reshape(L, H, 32, 64)
    dp = np.nonzero(pa.view(np.uint32) != pb.view(np.uint32))
    print("step", step, "partial-chain diffs:", len(dp[0]), [(int(w), int(x), int(y), int(z)) for w, x, y, z in zip(*dp)][:12])
    for (w, x, y, z) in list(zip(*dp))[:4]: print("    l%d h%d chain %d dh %d: seq %r mega %r" % (w, x, y, z, pa[w, x, y, z], pb[w, x, y, z]))
    diff = np.nonzero(a.view(np.uint32) != pm.view(np.uint32))
    print("step", step, "prob diffs:", len(diff[0]), [(int(x), int(y), int(z)) for x, y, z in zip(*diff)][:10])
    for (x, y, z) in list(zip(*diff))[:5]:
        print("   l%d h%d key %d: seq p=%r mega p=%r  score=%r max=%r" % (x, y, z, a[x, y, z], pm[x, y, z], b[x, y, 0, z], b[x, y, 0, :1500].max()))
    n_past += 1

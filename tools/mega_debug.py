"""Determinism / first-difference probe of the one-launch decode step: runs the same step several times and reports
the first hand-off edge whose granule values differ between runs, and whether the logits match the launch sequence."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W

name = sys.argv[1] if len(sys.argv) > 1 else "s128"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib), lib=lib)
os.environ["WHISPER_AMD_NO_MEGA"] = "1"; ref = ctx.create_state()
os.environ["WHISPER_AMD_NO_MEGA"] = "0"; meg = ctx.create_state()
pcm = wsynth.synth_audio(480000, 0)
for st in (ref, meg):
    st.pcm_to_mel(pcm); st.encode(0)
sot = ctx.token_sot(); prompt = [sot, sot + 1, sot + 102]
L, d, nv = ctx.model_n_text_layer(), ctx.model_n_text_state(), ctx.n_vocab()
names = ["QKV", "AO", "X1", "QC", "AO2", "X2", "HF", "X3"]
sizes = [3 * d // 2, d // 2, d, d // 2, d // 2, d, 2 * d, d]
lib.whisper_amd_mega_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.whisper_amd_seq_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
# a few sequence tokens through the launch-sequence path on BOTH states so that their KV caches agree
toks = [1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000]
for st in (ref, meg):
    os.environ["WHISPER_AMD_NO_MEGA"] = "1"
    st.decode(prompt, 0)
n_past = len(prompt)
bad_total = 0
for step, tok in enumerate(toks):
    ref.decode([tok], n_past)
    want = ref.get_logits_last(1).copy()
    sv = np.zeros(L * 8 * 2 * d, dtype=np.uint32); sl = np.zeros(nv, dtype=np.float32)
    lib.whisper_amd_seq_debug(ctx.ptr, meg.ptr, tok, n_past, sv.ctypes.data, sl.ctypes.data)
    sv = sv.reshape(L, 8, 2 * d)
    print("step %d: seq_debug logits %s ref" % (step, "==" if np.array_equal(sl.view(np.uint32), want.view(np.uint32)) else "!="))
    grs, lgs = [], []
    for r in range(runs):
        g = np.zeros(L * 8 * 2 * d, dtype=np.uint64); lg = np.zeros(nv, dtype=np.float32)
        rc = lib.whisper_amd_mega_debug(ctx.ptr, meg.ptr, tok, n_past, g.ctypes.data, lg.ctypes.data)
        if rc != 0: print("mega_debug rc", rc); sys.exit(2)
        grs.append(g.reshape(L, 8, 2 * d)); lgs.append(lg)
    for r in range(runs):
        okl = np.array_equal(lgs[r].view(np.uint32), want.view(np.uint32))
        first = None
        for l in range(L):
            for e in range(8):
                a = grs[0][l, e, :sizes[e]] & 0xffffffff; b = grs[r][l, e, :sizes[e]] & 0xffffffff
                if not np.array_equal(a, b) and first is None: first = (l, names[e], int((a != b).sum()), np.nonzero(a != b)[0][:8].tolist())
        if r == 0:
            fs = None
            for l in range(L):
                for e in range(8):
                    a = sv[l, e, :sizes[e]].astype(np.uint64); b = grs[0][l, e, :sizes[e]] & 0xffffffff
                    tg = grs[0][l, e, :sizes[e]] >> 32
                    if (not np.array_equal(a, b) or len(set(tg.tolist())) != 1) and fs is None:
                        idx = np.nonzero(a != b)[0]
                        fs = (l, names[e], int(len(idx)), idx[:10].tolist(), [hex(int(x)) for x in a[idx[:4]]], [hex(int(x)) for x in b[idx[:4]]], sorted(set(tg.tolist()))[:4])
            print("   mega vs launch sequence, first differing edge:", fs)
        if not okl or first: bad_total += 1
        print("step %d (n_past %d) run %d: logits %s ref; vs run0 first diff: %s" % (step, n_past, r, "==" if okl else "!=", first))
    n_past += 1
print("mega_debug: %d bad runs" % bad_total)

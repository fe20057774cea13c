"""Bring-up probe (GPU box): product (HIP) vs the live reference engine (oracle/_ref) stage by stage."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W

name = sys.argv[1] if len(sys.argv) > 1 else "s128"
ref = W.load_library(os.path.join(ROOT, "oracle/_ref/libwhisper_ref.so"))
amd = W.load_library()
W.set_log_callback(ref, None)
W.set_log_callback(amd, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.model_path(name)
pcm = wsynth.synth_audio(480000, 0)

rc = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False), lib=ref); rs = rc.create_state()
t = time.time(); ac = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(amd), lib=amd); as_ = ac.create_state(); print("amd load+state %.2fs" % (time.time() - t))
d = rc.model_n_audio_state(); nv = rc.n_vocab()

def stat(tag, a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    diff = np.abs(a - b)
    print("%-22s max|d|=%.3e mean|d|=%.3e ref[min,max]=[%.3f,%.3f] nan=%d" % (tag, diff.max(), diff.mean(), b.min(), b.max(), int(np.isnan(a).sum())))

# ---- mel
rs.pcm_to_mel(pcm, 8); as_.pcm_to_mel(pcm)
ref.ref_shim_mel_data.restype = C.POINTER(C.c_float); ref.ref_shim_mel_data.argtypes = [C.c_void_p]
ref.ref_shim_mel_n_len.argtypes = [C.c_void_p]
n_len = ref.ref_shim_mel_n_len(rs.ptr)
mel_ref = np.ctypeslib.as_array(ref.ref_shim_mel_data(rs.ptr), shape=(80 * n_len,)).reshape(80, n_len).copy()
amd.whisper_amd_get_mel.restype = C.c_int64
amd.whisper_amd_get_mel.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]
nl, nm = C.c_int(), C.c_int()
n = amd.whisper_amd_get_mel(as_.ptr, None, 0, nl, nm)
mel_amd = np.empty(n, np.float32); amd.whisper_amd_get_mel(as_.ptr, mel_amd.ctypes.data_as(C.POINTER(C.c_float)), n, nl, nm)
mel_amd = mel_amd.reshape(nm.value, nl.value)
print("mel shape", mel_amd.shape, mel_ref.shape, "n_len_org", as_.n_len(), rs.n_len())
stat("mel", mel_amd, mel_ref)

# ---- encoder on the SAME mel (inject the reference mel into both)
as_.set_mel(mel_ref); rs.set_mel(mel_ref)
t = time.time(); rs.encode(0, 8); tr = time.time() - t
t = time.time(); as_.encode(0); ta = time.time() - t
t = time.time(); as_.encode(0); ta2 = time.time() - t
print("encode: ref %.1f ms, amd %.1f ms (2nd %.1f ms)" % (tr * 1e3, ta * 1e3, ta2 * 1e3))
for nm_, fr, fa in (("embd_conv", "ref_shim_get_embd_conv", "whisper_amd_get_embd_conv"), ("embd_enc", "ref_shim_get_embd_enc", "whisper_amd_get_embd_enc")):
    f = getattr(ref, fr); f.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    x = np.empty(1500 * d, np.float32); r = f(rs.ptr, x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
    g = getattr(amd, fa); g.restype = C.c_int64; g.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64]
    y = np.empty(1500 * d, np.float32); r2 = g(as_.ptr, y.ctypes.data_as(C.POINTER(C.c_float)), y.size)
    if nm_ == "embd_conv": x = x.reshape(d, 1500).T.copy().ravel()     # reference keeps conv output time-fastest
    stat(nm_, y, x)

# ---- teacher-forced logits
sot = rc.token_sot()
seqs = [([sot, sot + 1, rc.token_transcribe()], 0), ([rc.token_beg() + 5], 3), ([1234], 4), ([4321, 777, 31000, 15, 50], 5), ([220], 10)]
for toks, n_past in seqs:
    rs.decode(toks, n_past, 8); as_.decode(toks, n_past)
    lr = rs.get_logits_last(len(toks)); la = as_.get_logits_last(len(toks))
    stat("logits n=%d past=%d" % (len(toks), n_past), la, lr)
    print("   argmax", int(la.argmax()), int(lr.argmax()), "top5 overlap", len(set(np.argsort(-la)[:5]) & set(np.argsort(-lr)[:5])))

# ---- full greedy
for tinc in (0.0, 0.2):
    out = []
    for lib, st in ((ref, rs), (amd, as_)):
        fp = W.FullParams(lib, best_of=1, temperature_inc=tinc, n_threads=8)
        t = time.time(); st.full(fp, pcm); dt = time.time() - t
        segs = st.segments(); out.append(segs)
        print("full tinc=%.1f %s: %.3fs nseg=%d ntok=%d" % (tinc, "ref" if lib is ref else "amd", dt, len(segs), sum(len(s["ids"]) for s in segs)))
    same = len(out[0]) == len(out[1]) and all(a["ids"] == b["ids"] and a["t0"] == b["t0"] and a["t1"] == b["t1"] and a["text"] == b["text"] for a, b in zip(*out))
    print("   identical segments:", same)
    if not same:
        for a, b in zip(*out):
            print("   ref", a["t0"], a["t1"], a["ids"][:16]); print("   amd", b["t0"], b["t1"], b["ids"][:16])

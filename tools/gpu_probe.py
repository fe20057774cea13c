"""Bring-up probe (GPU box): product (HIP) vs the live reference engine (oracle/_ref) stage by stage.
usage: gpu_probe.py <shape> [flash_attn 0|1] [full 0|1]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W

name = sys.argv[1] if len(sys.argv) > 1 else "s128"
flash = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
do_full = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
ref = W.load_library(os.path.join(ROOT, "oracle/_ref/libwhisper_ref.so"))
amd = W.load_library()
W.set_log_callback(ref, None)
W.set_log_callback(amd, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.model_path(name)
pcm = wsynth.synth_audio(480000, 0)
print("=== %s flash_attn=%s" % (name, flash))
rc = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False), lib=ref); rs = rc.create_state()
t = time.time(); ac = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(amd, flash_attn=flash), lib=amd); as_ = ac.create_state(); print("amd load+state %.2fs" % (time.time() - t))
d = rc.model_n_audio_state(); nv = rc.n_vocab()

def stat(tag, a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    ndiff = int((a.view(np.uint32) != b.view(np.uint32)).sum())
    diff = np.abs(a.astype(np.float64) - b)
    print("%-22s bit-exact=%-5s ndiff=%d/%d max|d|=%.3e mean|d|=%.3e nan=%d" % (tag, ndiff == 0, ndiff, a.size, diff.max(), diff.mean(), int(np.isnan(a).sum())))

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
stat("mel", mel_amd.reshape(nm.value, nl.value), mel_ref)

t = time.time(); rs.encode(0, 16); tr = time.time() - t
t = time.time(); as_.encode(0); ta = time.time() - t
t = time.time(); as_.encode(0); ta2 = time.time() - t
print("encode: ref %.1f ms, amd %.1f ms (2nd %.1f ms)" % (tr * 1e3, ta * 1e3, ta2 * 1e3))
for nm_, fr, fa in (("embd_conv", "ref_shim_get_embd_conv", "whisper_amd_get_embd_conv"), ("embd_enc", "ref_shim_get_embd_enc", "whisper_amd_get_embd_enc")):
    f = getattr(ref, fr); f.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    x = np.empty(1500 * d, np.float32); f(rs.ptr, x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
    g = getattr(amd, fa); g.restype = C.c_int64; g.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64]
    y = np.empty(1500 * d, np.float32); g(as_.ptr, y.ctypes.data_as(C.POINTER(C.c_float)), y.size)
    if nm_ == "embd_conv": x = x.reshape(d, 1500).T.copy().ravel()
    stat(nm_, y, x)

sot = rc.token_sot()
seqs = [([sot, sot + 1, rc.token_transcribe()], 0), ([rc.token_beg() + 5], 3), ([1234], 4), ([4321, 777, 31000, 15, 50], 5), ([220], 10),
        (list(range(1000, 1040)), 11), ([7], 51)]
for toks, n_past in seqs:
    rs.decode(toks, n_past, 16); as_.decode(toks, n_past)
    lr = rs.get_logits_last(len(toks)); la = as_.get_logits_last(len(toks))
    stat("logits n=%d past=%d" % (len(toks), n_past), la, lr)

if do_full:
    cases = [("greedy tinc=0", dict(best_of=1, temperature_inc=0.0)), ("greedy ladder", dict(best_of=2, temperature_inc=0.2)),
             ("beam5", dict(strategy=1, beam_size=3, best_of=2, temperature_inc=0.0))]
    for tag, kw in cases:
        out = []
        for lib, cx in ((ref, rc), (amd, ac)):
            st = cx.create_state()     # fresh state: the reference's no_speech_prob reads stale logits of earlier calls
            strategy = kw.get("strategy", 0)
            kk = {k: v for k, v in kw.items() if k != "strategy"}
            fp = W.FullParams(lib, strategy, n_threads=16, **kk)
            t = time.time(); st.full(fp, pcm); dt = time.time() - t
            segs = st.segments(); out.append(segs)
            print("full %-14s %s: %.3fs nseg=%d ntok=%d" % (tag, "ref" if lib is ref else "amd", dt, len(segs), sum(len(s["ids"]) for s in segs)))
            st.free()
        same = len(out[0]) == len(out[1]) and all(a["ids"] == b["ids"] and a["t0"] == b["t0"] and a["t1"] == b["t1"] and a["text"] == b["text"] for a, b in zip(*out))
        pexact = same and all(np.array_equal(np.float32(a["p"]), np.float32(b["p"])) and np.array_equal(np.float32(a["plog"]), np.float32(b["plog"])) for a, b in zip(*out))
        print("   identical segments: %s   token p/plog bit-identical: %s" % (same, pexact))
        if not same:
            for a, b in list(zip(*out))[:3]:
                print("   ref", a["t0"], a["t1"], a["ids"][:12]); print("   amd", b["t0"], b["t1"], b["ids"][:12])

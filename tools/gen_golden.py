#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE ENGINE itself (oracle/_ref/libwhisper_ref.so, compiled
from /root/reference by oracle/Makefile) on seeded synthetic models and audio.  Run in the build container
(the reference cannot travel); the fixtures are data only: inputs are regenerated from seeds by
tools/wsynth.py, outputs are stored as SHA-256 digests of the exact bytes plus small samples.

  python tools/gen_golden.py            # writes tests/golden/*.json / *.npz
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth  # noqa: E402
import whisper_rs as W  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ref = W.load_library(os.path.join(ROOT, "oracle", "_ref", "libwhisper_ref.so"))
W.set_log_callback(ref, None)
ref.ref_shim_mel_data.restype = C.POINTER(C.c_float); ref.ref_shim_mel_data.argtypes = [C.c_void_p]
ref.ref_shim_mel_n_len.argtypes = [C.c_void_p]; ref.ref_shim_mel_n_len_org.argtypes = [C.c_void_p]
for f in ("ref_shim_get_embd_conv", "ref_shim_get_embd_enc"):
    getattr(ref, f).argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def scripted_sequences(ctx):
    sot = ctx.token_sot()
    return [([sot, sot + 1, ctx.token_transcribe()], 0), ([ctx.token_beg() + 5], 3), ([1234], 4), ([4321, 777, 31000, 15, 50], 5),
            ([220], 10), (list(range(1000, 1040)), 11), ([7], 51)]


def segs_of(st):
    return [dict(t0=s["t0"], t1=s["t1"], text=s["text"].decode("latin1"), ids=s["ids"], tids=s["tids"],
                 p=[float(np.float32(x)) for x in s["p"]], plog=[float(np.float32(x)) for x in s["plog"]]) for s in st.segments()]


from gen_golden_cases import FULL_CASES, DTW_CASES  # noqa: E402


def main():
    os.makedirs(OUT, exist_ok=True)
    for shape, model_seed in (("s128", 0), ("s192", 0)):
        mp = wsynth.model_path(shape, model_seed)
        ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False), lib=ref)
        d = ctx.model_n_audio_state()
        gold = dict(shape=shape, model_seed=model_seed, model_bytes=os.path.getsize(mp), model_sha256=hashlib.sha256(open(mp, "rb").read()).hexdigest())
        samples = {}
        # ---- mel for several lengths (30 s, 1 s, 0.35 s, 16001 samples)
        gold["mel"] = {}
        for tag, n, seed in (("30s_seed0", 480000, 0), ("30s_seed1", 480000, 1), ("1s", 16000, 2), ("0.35s", 5600, 3), ("16001", 16001, 4), ("11s", 176000, 5)):
            st = ctx.create_state()
            pcm = wsynth.synth_audio(n, seed)
            st.pcm_to_mel(pcm, 4)
            n_len = ref.ref_shim_mel_n_len(st.ptr)
            mel = np.ctypeslib.as_array(ref.ref_shim_mel_data(st.ptr), shape=(80 * n_len,)).copy()
            gold["mel"][tag] = dict(n_samples=n, audio_seed=seed, n_len=n_len, n_len_org=ref.ref_shim_mel_n_len_org(st.ptr), sha256=digest(mel),
                                    sum=float(mel.astype(np.float64).sum()))
            samples["mel_" + tag] = mel.reshape(80, n_len)[::8, :: max(1, n_len // 64)].copy()
            st.free()
        # ---- encoder + teacher-forced logits (audio seed 0)
        st = ctx.create_state()
        pcm = wsynth.synth_audio(480000, 0)
        st.pcm_to_mel(pcm, 4)
        st.encode(0, 8)
        for nm, fn in (("embd_conv", "ref_shim_get_embd_conv"), ("embd_enc", "ref_shim_get_embd_enc")):
            x = np.empty(1500 * d, np.float32)
            getattr(ref, fn)(st.ptr, x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
            if nm == "embd_conv":
                x = x.reshape(d, 1500).T.copy().ravel()
            gold[nm] = dict(sha256=digest(x), sum=float(x.astype(np.float64).sum()), absmax=float(np.abs(x).max()))
            samples[nm] = x.reshape(1500, d)[::25].copy()
        gold["logits"] = []
        for toks, n_past in scripted_sequences(ctx):
            st.decode(toks, n_past, 8)
            lg = st.get_logits_last(len(toks))
            top = np.argsort(-lg, kind="stable")[:32]
            gold["logits"].append(dict(tokens=toks, n_past=n_past, sha256=digest(lg), top_ids=top.tolist(), absmax=float(np.abs(lg).max())))
            samples["logits_%d_%d" % (len(toks), n_past)] = lg.copy()
        # encoder at a non-zero mel offset and the language detector
        st.encode(1000, 8)
        x = np.empty(1500 * d, np.float32)
        ref.ref_shim_get_embd_enc(st.ptr, x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
        gold["embd_enc_offset1000"] = dict(sha256=digest(x))
        lang_id, probs = st.lang_detect(0, 8)
        gold["lang_detect"] = dict(lang_id=lang_id, probs_sha256=digest(probs))
        st.free()
        # ---- full transcriptions on fresh states
        gold["full"] = {}
        for tag, kw in FULL_CASES.items():
            for aseed in (0, 1):
                st = ctx.create_state()
                kk = {k: v for k, v in kw.items() if k != "strategy"}
                fp = W.FullParams(ref, kw.get("strategy", 0), n_threads=8, **kk)
                st.full(fp, wsynth.synth_audio(480000, aseed))
                gold["full"]["%s_seed%d" % (tag, aseed)] = segs_of(st)
                st.free()
        # two consecutive full() calls on ONE state (carries prompt_past / stale logits state across calls)
        st = ctx.create_state()
        fp = W.FullParams(ref, 0, n_threads=8, best_of=1, temperature_inc=0.0, no_context=False)
        st.full(fp, wsynth.synth_audio(480000, 0)); a = segs_of(st)
        st.full(fp, wsynth.synth_audio(480000, 1)); b = segs_of(st)
        gold["full"]["two_calls_same_state"] = [a, b]
        st.free()
        # short inputs
        for tag, n in (("1s", 16000), ("0.05s", 800)):
            st = ctx.create_state()
            fp = W.FullParams(ref, 0, n_threads=8, best_of=1, temperature_inc=0.0)
            st.full(fp, wsynth.synth_audio(n, 7))
            gold["full"]["short_" + tag] = segs_of(st)
            st.free()
        # DTW token timestamps (config 4): separate contexts with alignment heads (N top-most layers / a custom list)
        gold["dtw"] = {}
        for tag, preset, kw in DTW_CASES:
            dctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False, dtw_preset=preset, **kw), lib=ref)
            for aseed in (0, 1):
                st = dctx.create_state()
                st.full(W.FullParams(ref, 0, n_threads=8, best_of=1, temperature_inc=0.0), wsynth.synth_audio(480000, aseed))
                gold["dtw"]["%s_seed%d" % (tag, aseed)] = [dict(t0=s_["t0"], t1=s_["t1"], ids=s_["ids"], t_dtw=s_["t_dtw"]) for s_ in st.segments()]
                st.free()
            dctx.free()
        json.dump(gold, open(os.path.join(OUT, shape + ".json"), "w"), indent=1)
        np.savez_compressed(os.path.join(OUT, shape + "_samples.npz"), **samples)
        print(shape, "done:", {k: len(v) for k, v in gold["full"].items()})
        ctx.free()


if __name__ == "__main__":
    main()

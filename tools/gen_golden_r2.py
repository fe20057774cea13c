#!/usr/bin/env python3
"""Round-2 golden vectors: the decode modes BASELINE configs 3/4/5 name that round 1 left uncovered, produced by the
REFERENCE ENGINE itself (oracle/_ref/libwhisper_ref.so) on the seeded synthetic models.  Data only (ids, timestamps, p / plog).

  * several decoders at their default widths: beam_size 5 / 8, greedy best_of 5 with the 0.2 temperature ladder
    (the 5- and 8-row products: whisper.cpp:7100-7106, 7239-7291, 7442-7476), also language "zh" and "auto";
  * beam search TOGETHER with DTW token timestamps (config 4: whisper.cpp:8772-8933 behind 7239-7291);
  * whisper_tokenize and params.initial_prompt (whisper.cpp:3288-3336, 6911-6921);
  * whisper_full_parallel with 2 processors (whisper.cpp:7736-7864);
  * four different chunks for the whisper_amd_full_batch == per-chunk test (config 3's unit of work);
  * the same multi-decoder modes on the Q5_0 model.
Run in the build container: python tools/gen_golden_r2.py -> tests/golden/r2_cases.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth  # noqa: E402
import whisper_rs as W  # noqa: E402

MULTI_CASES = {
    "beam5": dict(strategy=1, beam_size=5, temperature_inc=0.0),
    "beam5_zh": dict(strategy=1, beam_size=5, temperature_inc=0.0, language="zh"),
    "beam8": dict(strategy=1, beam_size=8, temperature_inc=0.0),
    "greedy_best5_ladder": dict(strategy=0, best_of=5, temperature_inc=0.2),
    "beam_defaults": dict(strategy=1),                    # whisper_full_default_params(BEAM_SEARCH) untouched: beam 5, ladder 0.2
    "greedy_lang_auto": dict(strategy=0, best_of=1, temperature_inc=0.0, language="auto"),
    "greedy_initial_prompt": dict(strategy=0, best_of=1, temperature_inc=0.0, no_context=False, initial_prompt=" abc def zz hello q"),
}
# (tag, preset, context kwargs, full kwargs)
DTW_BEAM_CASES = [
    ("ntop2_beam5_zh", 1, dict(dtw_n_top=2), dict(strategy=1, beam_size=5, temperature_inc=0.0, language="zh")),
    ("custom_beam5", 2, dict(dtw_heads=[(1, 0), (2, 1), (2, 0)]), dict(strategy=1, beam_size=5, temperature_inc=0.0)),
]
TOKENIZE_TEXTS = [" abc def zz hello q", "hello world", " a", "", "zzzzzz  yx", " the quick brown fox jumps over the lazy dog", "A,b.c!d?"]
QUANT_CASES = {
    "beam5": dict(strategy=1, beam_size=5, temperature_inc=0.0),
    "greedy_best5_ladder": dict(strategy=0, best_of=5, temperature_inc=0.2),
    "beam8": dict(strategy=1, beam_size=8, temperature_inc=0.0),
}
BATCH_SEEDS = (0, 1, 2, 3)


def segs(st, dtw=False):
    out = []
    for s in st.segments():
        e = dict(t0=s["t0"], t1=s["t1"], text=s["text"].decode("latin1"), ids=s["ids"], tids=s["tids"],
                 p=[float(np.float32(x)) for x in s["p"]], plog=[float(np.float32(x)) for x in s["plog"]])
        if dtw:
            e["t_dtw"] = s["t_dtw"]
        out.append(e)
    return out


def params(Wm, lib, kw, **extra):
    kk = {k: v for k, v in kw.items() if k != "strategy"}
    kk.update(extra)
    return Wm.FullParams(lib, kw.get("strategy", 0), **kk)


def full_parallel(lib, Wm, mp, n_proc, pcm):
    """whisper_full_parallel on the context's own state -> [(t0, t1, ids)]"""
    cp = Wm.WhisperContextParameters(lib, use_gpu=lib is not None and hasattr(lib, "whisper_amd_full_batch"))
    lib.whisper_init_from_file_with_params.restype = C.c_void_p
    lib.whisper_init_from_file_with_params.argtypes = [C.c_char_p, type(cp.c)]
    fp = Wm.FullParams(lib, 0, best_of=1, temperature_inc=0.0, n_threads=4)
    lib.whisper_full_parallel.restype = C.c_int
    lib.whisper_full_parallel.argtypes = [C.c_void_p, type(fp.c), C.POINTER(C.c_float), C.c_int, C.c_int]
    for f, rt in (("whisper_full_n_segments", C.c_int), ("whisper_full_get_segment_t0", C.c_int64), ("whisper_full_get_segment_t1", C.c_int64)):
        getattr(lib, f).restype = rt
    lib.whisper_full_n_segments.argtypes = [C.c_void_p]
    lib.whisper_full_get_segment_t0.argtypes = [C.c_void_p, C.c_int]; lib.whisper_full_get_segment_t1.argtypes = [C.c_void_p, C.c_int]
    lib.whisper_full_n_tokens.restype = C.c_int; lib.whisper_full_n_tokens.argtypes = [C.c_void_p, C.c_int]
    lib.whisper_full_get_token_id.restype = C.c_int; lib.whisper_full_get_token_id.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.whisper_free.argtypes = [C.c_void_p]
    ctxp = lib.whisper_init_from_file_with_params(mp.encode(), cp.c)
    assert ctxp
    rc = lib.whisper_full_parallel(ctxp, fp.c, pcm.ctypes.data_as(C.POINTER(C.c_float)), len(pcm), n_proc)
    assert rc == 0, rc
    got = [[int(lib.whisper_full_get_segment_t0(ctxp, i)), int(lib.whisper_full_get_segment_t1(ctxp, i)),
            [lib.whisper_full_get_token_id(ctxp, i, j) for j in range(lib.whisper_full_n_tokens(ctxp, i))]] for i in range(lib.whisper_full_n_segments(ctxp))]
    lib.whisper_free(ctxp)
    return got


if __name__ == "__main__":
    ref = W.load_library(os.path.join(ROOT, "oracle", "_ref", "libwhisper_ref.so"))
    W.set_log_callback(ref, None)
    gold = {}
    for shape in ("s128", "s192"):
        mp = wsynth.model_path(shape)
        ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False), lib=ref)
        g = dict(multi={}, dtw_beam={}, batch={})
        for tag, kw in MULTI_CASES.items():
            for aseed in (0, 1):
                st = ctx.create_state()
                st.full(params(W, ref, kw, n_threads=8), wsynth.synth_audio(480000, aseed))
                g["multi"]["%s_seed%d" % (tag, aseed)] = dict(segs=segs(st), lang_id=st.full_lang_id())
                st.free()
            print(shape, tag, sum(len(s["ids"]) for s in g["multi"][tag + "_seed0"]["segs"]), flush=True)
        for tag, preset, ckw, fkw in DTW_BEAM_CASES:
            dctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False, dtw_preset=preset, **ckw), lib=ref)
            for aseed in (0, 1):
                st = dctx.create_state()
                st.full(params(W, ref, fkw, n_threads=8), wsynth.synth_audio(480000, aseed))
                g["dtw_beam"]["%s_seed%d" % (tag, aseed)] = segs(st, dtw=True)
                st.free()
            dctx.free()
        for aseed in BATCH_SEEDS:
            st = ctx.create_state()
            st.full(W.FullParams(ref, 0, best_of=1, temperature_inc=0.0, n_threads=8), wsynth.synth_audio(480000, aseed))
            g["batch"]["seed%d" % aseed] = segs(st)
            st.free()
        g["tokenize"] = [dict(text=t, ids=ctx.tokenize(t)) for t in TOKENIZE_TEXTS]
        ctx.free()
        g["full_parallel_2"] = full_parallel(ref, W, mp, 2, wsynth.synth_audio(960000, 4))
        g["full_parallel_3"] = full_parallel(ref, W, mp, 3, wsynth.synth_audio(16000 * 75, 6))
        gold[shape] = g
    qg = {}
    mp = wsynth.quant_model_path("s128", "q5_0")
    ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False), lib=ref)
    for tag, kw in QUANT_CASES.items():
        for aseed in (0, 1):
            st = ctx.create_state()
            st.full(params(W, ref, kw, n_threads=8), wsynth.synth_audio(480000, aseed))
            qg["%s_seed%d" % (tag, aseed)] = segs(st)
            st.free()
    ctx.free()
    gold["s128_q5_0"] = qg
    json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "r2_cases.json"), "w"), indent=1)
    print("done")

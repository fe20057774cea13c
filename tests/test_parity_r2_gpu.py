"""Parity tests, part 2 (MI355X): the decode modes and shapes of BASELINE configs 1 / 3 / 4 / 5 that part 1 leaves out.

  * goldens from the reference engine (tests/golden/r2_cases.json, tools/gen_golden_r2.py): 5 and 8 decoders (beam_size 5 / 8,
    best_of 5 with the temperature ladder, the untouched beam defaults), language "zh" and "auto", initial_prompt + whisper_tokenize,
    beam search WITH DTW token timestamps, whisper_full_parallel, four chunks through whisper_amd_full_batch, the same
    multi-decoder modes on a Q5_0 model;
  * the reference engine itself (oracle/_ref/libwhisper_ref.so, a prebuilt binary on the GPU box) run LIVE beside the product
    where the shape is too big for committed vectors: ggml-tiny shape greedy full() + a named alignment-head preset (config 1's
    substitute), ggml-small shape encoder + teacher-forced logits (config 2's size), a large-v3-wide Q5_0 model with the
    streaming call pattern (config 5), a medium-wide model with beam_size 5 + DTW + "zh" (config 4).
Everything here is bit-exact (flash_attn = false): SHA-256 of float bytes, identical ids / timestamps / p / plog.
"""
import ctypes as C
import hashlib
import json
import os
import threading

import numpy as np
import pytest

import wsynth
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _segs(st, dtw=False):
    out = []
    for s in st.segments():
        e = dict(t0=s["t0"], t1=s["t1"], text=s["text"].decode("latin1"), ids=s["ids"], tids=s["tids"],
                 p=[float(np.float32(x)) for x in s["p"]], plog=[float(np.float32(x)) for x in s["plog"]])
        if dtw:
            e["t_dtw"] = s["t_dtw"]
        out.append(e)
    return out


def _params(wrs, lib, kw, **extra):
    kk = {k: v for k, v in kw.items() if k != "strategy"}
    kk.update(extra)
    return wrs.FullParams(lib, kw.get("strategy", 0), **kk)


def _get(lib, fn, st, n):
    f = getattr(lib, fn)
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64]
    out = np.empty(n, np.float32)
    r = f(st.ptr, out.ctypes.data_as(C.POINTER(C.c_float)), n)
    assert r == n, (fn, r, n)
    return out


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLDEN, "r2_cases.json")))


@pytest.fixture(scope="module")
def gen():
    import gen_golden_r2
    return gen_golden_r2


# ------------------------------------------------------------------------------------------------------------
# goldens of the reference engine
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["s128", "s192"])
def test_five_and_eight_decoders_languages_initial_prompt(wrs, amd_lib, gold, gen, shape):
    """beam_size 5 / 8, best_of 5 + ladder, default beam params, language zh / auto, initial_prompt: identical segments, ids, p, plog
    and detected language id (the 5- and 8-row products, whisper.cpp:7100-7106, 7239-7291; 4037-4110; 6911-6921)."""
    ctx = wrs.WhisperContext.new_with_params(wsynth.model_path(shape), wrs.WhisperContextParameters(amd_lib), lib=amd_lib)
    for tag, kw in gen.MULTI_CASES.items():
        for aseed in (0, 1):
            st = ctx.create_state()
            st.full(_params(wrs, amd_lib, kw), wsynth.synth_audio(480000, aseed))
            want = gold[shape]["multi"]["%s_seed%d" % (tag, aseed)]
            assert _segs(st) == want["segs"], (shape, tag, aseed)
            assert st.full_lang_id() == want["lang_id"], (shape, tag, aseed)
            # the steps with several decoders went through the one-launch form (wa_rows.hip), none was sent back for a time-out
            served, back = st.rows_stats()
            if kw.get("strategy", 0) == 1 and amd_lib.whisper_amd_rows_enabled(st.ptr): assert served > 0      # (s192: d % 128 != 0 keeps the launch sequence), (shape, tag, aseed, served, back)
            assert back <= 2, (shape, tag, aseed, served, back)       # (an uncertifiable soft-max sum, ~1e-9 each, may send a pass back)
            st.free()
    ctx.free()


@pytest.mark.parametrize("shape", ["s128", "s192"])
def test_beam_search_with_dtw_timestamps(wrs, amd_lib, gold, gen, shape):
    """config 4's combination: beam_size 5 (and "zh") with DTW token timestamps (whisper.cpp:8772-8933 behind 7239-7291)."""
    for tag, preset, ckw, fkw in gen.DTW_BEAM_CASES:
        ctx = wrs.WhisperContext.new_with_params(wsynth.model_path(shape), wrs.WhisperContextParameters(amd_lib, dtw_preset=preset, **ckw), lib=amd_lib)
        for aseed in (0, 1):
            st = ctx.create_state()
            st.full(_params(wrs, amd_lib, fkw), wsynth.synth_audio(480000, aseed))
            assert _segs(st, dtw=True) == gold[shape]["dtw_beam"]["%s_seed%d" % (tag, aseed)], (shape, tag, aseed)
            st.free()
        ctx.free()


@pytest.mark.parametrize("shape", ["s128", "s192"])
def test_tokenize_matches_reference(wrs, amd_lib, gold, shape):
    """whisper_tokenize (whisper.cpp:3288-3336, 3973-3986): identical ids; a too-small buffer returns -needed."""
    ctx = wrs.WhisperContext.new_with_params(wsynth.model_path(shape), wrs.WhisperContextParameters(amd_lib), lib=amd_lib)
    for e in gold[shape]["tokenize"]:
        assert ctx.tokenize(e["text"]) == e["ids"], e["text"]
    long = next(e for e in gold[shape]["tokenize"] if len(e["ids"]) > 3)
    arr = (C.c_int32 * 2)()
    assert amd_lib.whisper_tokenize(ctx.ptr, long["text"].encode(), arr, 2) == -len(long["ids"])
    ctx.free()


@pytest.mark.parametrize("shape", ["s128", "s192"])
def test_full_parallel_matches_reference(wrs, amd_lib, gold, gen, shape):
    """whisper_full_parallel with 2 and 3 processors (whisper.cpp:7736-7864): the stitched segment list of the reference engine."""
    mp = wsynth.model_path(shape)
    assert gen.full_parallel(amd_lib, wrs, mp, 2, wsynth.synth_audio(960000, 4)) == gold[shape]["full_parallel_2"]
    assert gen.full_parallel(amd_lib, wrs, mp, 3, wsynth.synth_audio(16000 * 75, 6)) == gold[shape]["full_parallel_3"]


@pytest.mark.parametrize("shape", ["s128", "s192"])
def test_full_batch_equals_per_chunk_reference(wrs, amd_lib, gold, gen, shape):
    """config 3's unit of work: four different chunks transcribed together by whisper_amd_full_batch == the reference engine's
    result for each chunk alone."""
    ctx = wrs.WhisperContext.new_with_params(wsynth.model_path(shape), wrs.WhisperContextParameters(amd_lib), lib=amd_lib)
    states = [ctx.create_state() for _ in gen.BATCH_SEEDS]
    pcms = [wsynth.synth_audio(480000, s) for s in gen.BATCH_SEEDS]
    wrs.full_batch(ctx, states, wrs.FullParams(amd_lib, 0, best_of=1, temperature_inc=0.0), pcms)
    for st, s in zip(states, gen.BATCH_SEEDS):
        assert _segs(st) == gold[shape]["batch"]["seed%d" % s], (shape, s)
        st.free()
    # ... and the chunks' single-token steps really were decoded in lock step (one pass for several chunks' rows)
    steps, rows = C.c_long(), C.c_long()
    amd_lib.whisper_amd_batch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    amd_lib.whisper_amd_batch_stats(ctx.ptr, steps, rows)
    assert steps.value > 50 and rows.value > 2 * steps.value, (steps.value, rows.value)
    # ... each pass as ONE launch (wa_rows.hip)
    if shape == "s128": assert amd_lib.whisper_amd_batch_one_launch(ctx.ptr) >= steps.value - 2, (amd_lib.whisper_amd_batch_one_launch(ctx.ptr), steps.value)
    ctx.free()


def test_quantised_five_and_eight_decoders(wrs, amd_lib, gold, gen):
    """The multi-row quantised products (M = 5, 8) on the Q5_0 model: beam 5 / 8 and best_of 5 with the ladder."""
    ctx = wrs.WhisperContext.new_with_params(wsynth.quant_model_path("s128", "q5_0"), wrs.WhisperContextParameters(amd_lib), lib=amd_lib)
    for tag, kw in gen.QUANT_CASES.items():
        for aseed in (0, 1):
            st = ctx.create_state()
            st.full(_params(wrs, amd_lib, kw), wsynth.synth_audio(480000, aseed))
            assert _segs(st) == gold["s128_q5_0"]["%s_seed%d" % (tag, aseed)], (tag, aseed)
            served, back = st.rows_stats()          # the several-decoder steps as ONE launch (the quantised form of wa_rows.hip)
            if kw.get("strategy", 0) == 1: assert served > 0, (tag, aseed, served, back)
            assert back <= 2, (tag, aseed, served, back)
            st.free()
    ctx.free()


def test_two_states_on_two_threads_greedy(wrs, amd_lib, gold):
    """Two states of one context driven from two host threads at once (greedy, i.e. both want the device's one-launch slot):
    each result equals the sequential golden; no deadlock (the slot is handed over, never held across a failure)."""
    ctx = wrs.WhisperContext.new_with_params(wsynth.model_path("s128"), wrs.WhisperContextParameters(amd_lib), lib=amd_lib)
    out, err = {}, []

    def work(seed):
        try:
            for _ in range(2):
                st = ctx.create_state()
                st.full(wrs.FullParams(amd_lib, 0, best_of=1, temperature_inc=0.0), wsynth.synth_audio(480000, seed))
                out[seed] = _segs(st)
                st.free()
        except Exception as e:       # noqa: BLE001
            err.append(e)

    ths = [threading.Thread(target=work, args=(s,)) for s in (0, 1)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
        assert not t.is_alive(), "a greedy full() is stuck waiting for the one-launch slot"
    assert not err, err
    for s in (0, 1):
        assert out[s] == gold["s128"]["batch"]["seed%d" % s]
    ctx.free()


# ------------------------------------------------------------------------------------------------------------
# the reference engine run live beside the product
# ------------------------------------------------------------------------------------------------------------
def _both(wrs, amd_lib, ref_lib, mp, amd_kw=None, **ctx_kw):
    a = wrs.WhisperContext.new_with_params(mp, wrs.WhisperContextParameters(amd_lib, **ctx_kw), lib=amd_lib)
    r = wrs.WhisperContext.new_with_params(mp, wrs.WhisperContextParameters(ref_lib, use_gpu=False, **ctx_kw), lib=ref_lib)
    return a, r


def test_tiny_shape_full_greedy_and_named_dtw_preset_live(wrs, amd_lib, ref_lib):
    """BASELINE config 1's substitute (no cargo / jfk.wav / ggml-tiny.bin offline): a ggml-tiny-shaped model, greedy full() with the
    library's default parameters (best_of 5, ladder 0.2) and with DTW through the named preset WHISPER_AHEADS_TINY
    (whisper.cpp:424-438), against the reference engine on the same inputs."""
    mp = wsynth.model_path("tiny")
    for ctx_kw, fkw, dtw in ((dict(), dict(strategy=0), False),
                             (dict(dtw_preset=4), dict(strategy=0, best_of=1, temperature_inc=0.0), True)):      # 4 = WHISPER_AHEADS_TINY
        a, r = _both(wrs, amd_lib, ref_lib, mp, **ctx_kw)
        pcm = wsynth.synth_audio(480000, 11)
        sa, sr = a.create_state(), r.create_state()
        sa.full(_params(wrs, amd_lib, fkw), pcm)
        sr.full(_params(wrs, ref_lib, fkw, n_threads=8), pcm)
        got, want = _segs(sa, dtw), _segs(sr, dtw)
        assert sum(len(s["ids"]) for s in want) > 0
        assert got == want
        for x in (sa, sr):
            x.free()
        a.free(); r.free()


def test_small_shape_encoder_and_logits_live(wrs, amd_lib, ref_lib):
    """BASELINE.json's size: ggml-small shape, encoder output and five teacher-forced decoder passes (prompt, single tokens, a 5-token
    batch), digests equal to the reference engine's on the same inputs."""
    mp = wsynth.model_path("small")
    a, r = _both(wrs, amd_lib, ref_lib, mp)
    ref_lib.ref_shim_get_embd_enc.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    pcm = wsynth.synth_audio(480000, 0)
    sa, sr = a.create_state(), r.create_state()
    sa.pcm_to_mel(pcm); sa.encode(0)
    sr.pcm_to_mel(pcm, 8); sr.encode(0, 16)
    x = np.empty(1500 * 768, np.float32)
    ref_lib.ref_shim_get_embd_enc(sr.ptr, x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
    assert digest(_get(amd_lib, "whisper_amd_get_embd_enc", sa, 1500 * 768)) == digest(x)
    sot = a.token_sot()
    for toks, n_past in (([sot, sot + 1, a.token_transcribe()], 0), ([a.token_beg() + 3], 3), ([1234], 4), ([4321, 777, 31000, 15, 50], 5), ([220], 10)):
        sa.decode(toks, n_past); sr.decode(toks, n_past, 16)
        assert digest(sa.get_logits_last(len(toks))) == digest(sr.get_logits_last(len(toks))), (toks, n_past)
    for x_ in (sa, sr):
        x_.free()
    a.free(); r.free()


def test_large_v3_wide_q5_0_streaming_live(wrs, amd_lib, ref_lib):
    """BASELINE config 5's shapes: n_mels 128, n_vocab 51866, d 1280 (few layers), Q5_0 weights written by the reference's own
    quantizer: encoder output, teacher-forced logits and the streaming call pattern against the reference engine."""
    import gen_golden_quant as gq
    mp = wsynth.quant_model_path("w1280", "q5_0")
    a, r = _both(wrs, amd_lib, ref_lib, mp)
    ref_lib.ref_shim_get_embd_enc.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    pcm = wsynth.synth_audio(480000, 0)
    sa, sr = a.create_state(), r.create_state()
    sa.pcm_to_mel(pcm); sa.encode(0)
    sr.pcm_to_mel(pcm, 8); sr.encode(0, 16)
    x = np.empty(1500 * 1280, np.float32)
    ref_lib.ref_shim_get_embd_enc(sr.ptr, x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
    assert digest(_get(amd_lib, "whisper_amd_get_embd_enc", sa, 1500 * 1280)) == digest(x)
    sot = a.token_sot()
    for toks, n_past in (([sot, sot + 1, a.token_transcribe()], 0), ([a.token_beg() + 3], 3), ([1234], 4), ([4321, 777, 31000, 15, 50], 5)):
        sa.decode(toks, n_past); sr.decode(toks, n_past, 16)
        assert digest(sa.get_logits_last(len(toks))) == digest(sr.get_logits_last(len(toks))), (toks, n_past)
    for x_ in (sa, sr):
        x_.free()
    got, want = gq.stream_run(wrs, amd_lib, a), gq.stream_run(wrs, ref_lib, r, 16)
    assert got == want and sum(len(s["ids"]) for w in want for s in w) > 0
    a.free(); r.free()


def test_medium_wide_beam5_zh_dtw_live(wrs, amd_lib, ref_lib):
    """BASELINE config 4's shapes and mode: d 1024 / 16 heads (few layers) against the reference engine: (a) teacher-forced 5- and
    8-token batches of distinct tokens - the 5- and 8-row products at K = 1024 / 4096 -, logits digests equal; (b) beam_size 5,
    language "zh", DTW token timestamps on (N top-most layers): identical ids / p / plog / t_dtw.  (This random model repeats one
    token, and a 220-token window makes the REFERENCE's DTW pass run out of its dtw_mem_size: the window is capped at 48 tokens.)"""
    mp = wsynth.model_path("m1024")
    a, r = _both(wrs, amd_lib, ref_lib, mp, dtw_preset=1, dtw_n_top=2)
    pcm = wsynth.synth_audio(480000, 5)
    sa, sr = a.create_state(), r.create_state()
    sa.pcm_to_mel(pcm); sa.encode(0)
    sr.pcm_to_mel(pcm, 8); sr.encode(0, 16)
    sot = a.token_sot()
    for toks, n_past in (([sot, sot + 2, a.token_transcribe()], 0), ([4321, 777, 31000, 15, 50], 3), ([9, 99, 999, 9999, 19999, 29999, 39999, 49999], 8),
                         ([1, 2, 3, 4, 5], 16), ([700], 21)):
        sa.decode(toks, n_past); sr.decode(toks, n_past, 16)
        assert digest(sa.get_logits_last(len(toks))) == digest(sr.get_logits_last(len(toks))), (toks, n_past)
    for x_ in (sa, sr):
        x_.free()
    fkw = dict(strategy=1, beam_size=5, temperature_inc=0.0, language="zh", max_tokens=48)
    sa, sr = a.create_state(), r.create_state()
    sa.full(_params(wrs, amd_lib, fkw), pcm)
    sr.full(_params(wrs, ref_lib, fkw, n_threads=16), pcm)
    got, want = _segs(sa, True), _segs(sr, True)
    assert sum(len(s["ids"]) for s in want) > 0
    assert got == want
    for x_ in (sa, sr):
        x_.free()
    a.free(); r.free()


def _full_depth(wrs, amd_lib, ref_lib, mp, d, n_mels_note):
    """encoder output + three teacher-forced decoder passes (a prompt, a single token through the one-launch step, a 5-token batch through the
    several-rows step) of a FULL-DEPTH model against the reference engine on the same inputs; the one-launch forms must be the ones that ran."""
    a, r = _both(wrs, amd_lib, ref_lib, mp)
    ref_lib.ref_shim_get_embd_enc.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    amd_lib.whisper_amd_mega_enabled.argtypes = [C.c_void_p]
    pcm = wsynth.synth_audio(480000, 0)
    sa, sr = a.create_state(), r.create_state()
    sa.pcm_to_mel(pcm); sa.encode(0)
    sr.pcm_to_mel(pcm, 8); sr.encode(0, 16)
    x = np.empty(1500 * d, np.float32)
    ref_lib.ref_shim_get_embd_enc(sr.ptr, x.ctypes.data_as(C.POINTER(C.c_float)), x.size)
    assert digest(_get(amd_lib, "whisper_amd_get_embd_enc", sa, 1500 * d)) == digest(x), n_mels_note
    sot = a.token_sot()
    for toks, n_past in (([sot, sot + 1, a.token_transcribe()], 0), ([a.token_beg() + 3], 3), ([4321, 777, 31000, 15, 50], 4)):
        sa.decode(toks, n_past); sr.decode(toks, n_past, 16)
        assert digest(sa.get_logits_last(len(toks))) == digest(sr.get_logits_last(len(toks))), (toks, n_past)
    assert amd_lib.whisper_amd_mega_enabled(sa.ptr) == 1
    served, back = sa.rows_stats()
    # the prompt and the 5-token batch - and, for a wide quantised model, the single token too (its step is the several-rows kernel with one row: wa_internal.h)
    assert served == (3 if ("q5_0" in n_mels_note and d > 768) else 2) and back == 0, (served, back)
    for x_ in (sa, sr):
        x_.free()
    a.free(); r.free()


def test_medium_full_depth_live(wrs, amd_lib, ref_lib):
    """BASELINE config 4's model at FULL depth: ggml-medium shape (d 1024, 16 heads, 24 + 24 layers)."""
    _full_depth(wrs, amd_lib, ref_lib, wsynth.model_path("medium"), 1024, "medium")


def test_large_v3_q5_0_full_depth_live(wrs, amd_lib, ref_lib):
    """BASELINE config 5's model at FULL depth: ggml-large-v3 shape (d 1280, 20 heads, 32 + 32 layers, 128 mels, 51866 tokens), Q5_0 weights written
    by the reference's own quantizer."""
    _full_depth(wrs, amd_lib, ref_lib, wsynth.quant_model_path("large-v3", "q5_0"), 1280, "large-v3 q5_0")


def test_small_shape_full_greedy_live(wrs, amd_lib, ref_lib):
    """BASELINE.json's headline workload as a test: ggml-small shape, greedy full() of a 30 s chunk - segments, ids, p, plog identical to the
    reference engine's (bench.py checks the same inside its cpu_baseline leg)."""
    mp = wsynth.model_path("small")
    a, r = _both(wrs, amd_lib, ref_lib, mp)
    pcm = wsynth.synth_audio(480000, 0)
    fkw = dict(strategy=0, best_of=1, temperature_inc=0.0, language="en", no_context=True)
    sa, sr = a.create_state(), r.create_state()
    sa.full(_params(wrs, amd_lib, fkw), pcm)
    sr.full(_params(wrs, ref_lib, fkw, n_threads=16), pcm)
    # the decode loop's own record of decoder 0 (the random model writes 220 tokens without a closing timestamp, i.e. possibly no segment):
    # raw token ids and the loop's decisions, then the segment lists
    info_r, info_g = (C.c_double * 8)(), (C.c_double * 8)()
    ids_r, ids_g = (C.c_int32 * 512)(), (C.c_int32 * 512)()
    ref_lib.ref_shim_decoder_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int]
    amd_lib.whisper_amd_decoder_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int]
    n_r = ref_lib.ref_shim_decoder_info(sr.ptr, 0, info_r, ids_r, 512)
    n_g = amd_lib.whisper_amd_decoder_info(sa.ptr, 0, info_g, ids_g, 512)
    assert n_r > 100 and n_r == n_g and list(ids_r[:n_r]) == list(ids_g[:n_g])
    assert list(info_r) == list(info_g)             # failed / completed / has_ts / seek_delta / result_len / avg_logprobs / entropy / no_speech_prob
    assert _segs(sa) == _segs(sr)
    for x_ in (sa, sr):
        x_.free()
    a.free(); r.free()


# ------------------------------------------------------------------------------------------------------------
# compiled callers at the boundary (tests/native/, built by oracle/Makefile in the container)
# ------------------------------------------------------------------------------------------------------------
def test_compiled_replay_of_the_whisper_rs_call_order(tmp_path):
    """One C++ source (tests/native/replay_whisper_rs.cpp: whisper-rs's call order, 48- / 296-byte structs by value, sret returns, getters
    re-entered inside new_segment) compiled against the product (header + libwhisper.so) and against the reference (header + engine):
    the two programs print the same bytes, greedy and beam search."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    amd, ref = os.path.join(root, "tests", "native", "replay_amd"), os.path.join(root, "oracle", "_ref", "replay_ref")
    assert os.path.exists(amd), "tests/native/replay_amd missing (python -c 'import __graft_entry__ as g; g.build()')"
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/replay_ref not built")
    pcm = str(tmp_path / "pcm.f32")
    wsynth.synth_audio(480000, 0).tofile(pcm)
    mp = wsynth.model_path("s128")
    for mode in ([], ["beam"]):
        a = subprocess.run([amd, mp, pcm, "1"] + mode, capture_output=True, timeout=600)
        r = subprocess.run([ref, mp, pcm, "0"] + mode, capture_output=True, timeout=600)
        assert a.returncode == 0 and r.returncode == 0, (a.returncode, r.returncode, a.stderr[-400:])
        assert a.stdout == r.stdout and b"seg 0 " in a.stdout, mode
    gold = json.load(open(os.path.join(GOLDEN, "s128.json")))["full"]["greedy_tinc0_seed0"]
    a = subprocess.run([amd, mp, pcm, "1"], capture_output=True, timeout=600).stdout.decode()
    ids = [int(l.split()[3]) for l in a.splitlines() if l.startswith(" tok ")]
    assert ids == [i for s in gold for i in s["ids"]]


def test_reference_bench_program_runs_on_the_product(tmp_path):
    """The reference's own examples/bench/bench.cpp, compiled unmodified against include/whisper.h (-> whisper_amd.h) and linked with
    libwhisper.so (oracle/Makefile: whisper-bench-amd): its whole protocol (set_mel, encode, 256 single-token decodes, 64 batches of 5,
    16 prompts of 256, print_timings) runs through the context-level API of the product."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "whisper-bench-amd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/whisper-bench-amd not built (needs the reference's bench.cpp: container only)")
    r = subprocess.run([exe, "-m", wsynth.model_path("s128"), "-t", "4"], capture_output=True, timeout=600)
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, err[-800:]
    assert "encode time" in err and "decode time" in err and "batchd time" in err and "prompt time" in err, err[-800:]


def test_full_batch_with_chunks_that_leave_early(wrs, amd_lib, gold):
    """Lock-step groups under membership churn: five chunks in one call (a group of four and a single), among them a 1 s input that
    finishes after a few tokens and a 0.05 s input that is refused at once - the others must not wait for them, and every chunk's
    result equals the reference engine's for that input alone."""
    g1 = json.load(open(os.path.join(GOLDEN, "s128.json")))["full"]
    ctx = wrs.WhisperContext.new_with_params(wsynth.model_path("s128"), wrs.WhisperContextParameters(amd_lib), lib=amd_lib)
    pcms = [wsynth.synth_audio(480000, 0), wsynth.synth_audio(16000, 7), wsynth.synth_audio(480000, 1), wsynth.synth_audio(800, 7), wsynth.synth_audio(480000, 2)]
    want = [gold["s128"]["batch"]["seed0"], g1["short_1s"], gold["s128"]["batch"]["seed1"], g1["short_0.05s"], gold["s128"]["batch"]["seed2"]]
    for _ in range(2):                # twice: the second call reuses the states
        states = [ctx.create_state() for _ in pcms]
        wrs.full_batch(ctx, states, wrs.FullParams(amd_lib, 0, best_of=1, temperature_inc=0.0), pcms)
        for i, st in enumerate(states):
            got = _segs(st)
            assert [(s["t0"], s["t1"], s["ids"], s["p"], s["plog"]) for s in got] == [(s["t0"], s["t1"], s["ids"], s["p"], s["plog"]) for s in want[i]], i
            st.free()
    ctx.free()


@pytest.mark.parametrize("shape", ["s128", "s128:q5_0"])
def test_environment_switches_do_not_change_a_bit(shape):
    """"results are bit-identical under every combination" of the backend's switches.  They are read once per process, so one
    process per combination (tools/switch_check.py: four chunks in a lock-step group, greedy, beam 5, best_of 3 with the ladder -> one digest)."""
    import subprocess, sys
    combos = [{}, {"WHISPER_AMD_ROWS_HOST_OUT": "0"}, {"WHISPER_AMD_NO_RUN_AHEAD": "1"}, {"WHISPER_AMD_NO_ROWS": "1"}, {"WHISPER_AMD_NO_BATCHER": "1"},
              {"WHISPER_AMD_NO_MEGA": "1", "WHISPER_AMD_NO_ROWS": "1"}, {"WHISPER_AMD_NO_OVERLAP": "1"}, {"WHISPER_AMD_SINGLE_ROWS": "1"},
              # ... and the test build whose one-launch kernels stall waves and workgroups at random (lock-step passes running ahead, windows, beams under stalls)
              {"WA_LIB": os.path.join(ROOT, "whisper-rust_amd", "libwhisper_chaos.so")},
              # ... and every cross soft-max total of the several-rows kernel through its in-order path (otherwise taken ~3e-5 of the time), also under stalls
              {"WHISPER_AMD_ROWS_FORCE_INORDER": "1"},
              {"WHISPER_AMD_ROWS_FORCE_INORDER": "1", "WA_LIB": os.path.join(ROOT, "whisper-rust_amd", "libwhisper_chaos.so")}]
    lines = []
    for env_extra in combos:
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "switch_check.py"), shape], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (env_extra, r.stdout[-400:], r.stderr[-800:])
        line = [l for l in r.stdout.splitlines() if l.startswith("digest")]
        assert line, (env_extra, r.stdout[-400:])
        lines.append(line[-1])
    assert len(set(lines)) == 1, list(zip([str(c) for c in combos], lines))


@pytest.mark.parametrize("shape", ["s128", "s128:q5_0"])
def test_lockstep_group_with_chunks_of_several_windows(wrs, amd_lib, shape):
    """Members of a lock-step group whose audio spans several 30 s windows (45 s, 70 s, 30 s, 100 s): they encode again in the middle of the group's life,
    leave their run-ahead windows and open new ones at different times, finish at different times.  Each chunk's segments must equal its solo run's."""
    mp = wsynth.quant_model_path(*shape.split(":")) if ":" in shape else wsynth.model_path(shape)
    ctx = wrs.WhisperContext.new_with_params(mp, wrs.WhisperContextParameters(amd_lib), lib=amd_lib)
    lens = [45, 70, 30, 100]
    pcms = [wsynth.synth_audio(16000 * n, 40 + i) for i, n in enumerate(lens)]
    fp = wrs.FullParams(amd_lib, 0, best_of=1, temperature_inc=0.0)
    solo = []
    for p in pcms:
        st = ctx.create_state(); st.full(fp, p); solo.append(_segs(st)); st.free()
    states = [ctx.create_state() for _ in pcms]
    wrs.full_batch(ctx, states, fp, pcms)
    for i, st in enumerate(states):
        assert _segs(st) == solo[i], (shape, lens[i])
        st.free()
    steps, rows = C.c_long(), C.c_long()
    amd_lib.whisper_amd_batch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    amd_lib.whisper_amd_batch_stats(ctx.ptr, steps, rows)
    assert steps.value > 50 and rows.value > steps.value, (steps.value, rows.value)      # passes really were shared
    ctx.free()

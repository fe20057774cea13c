"""BASELINE configs 4 and 5 on the driver's clock (called by bench.py on rank 0, N = 1): each returns a dict for the JSON line.

config 4: ggml-medium shape (d 1024, 24 + 24 layers), beam_size 5, language "zh", DTW token timestamps (N top-most layers).
config 5: ggml-large-v3 shape (d 1280, 32 + 32 layers, 128 mels, 51866 tokens) with Q5_0 weights written by the reference's own
          quantizer: the reference's whisper-bench protocol (examples/bench/bench.cpp; printed per model in scripts/bench-all-gg.txt:256-284)
          and the streaming call pattern (examples/stream/stream.cpp:311-335: 6 s windows every 3 s, reduced audio context).
Models are synthetic (no ggml-*.bin offline); the CPU reference (oracle/_ref/libwhisper_ref.so = the reference engine compiled from
its own sources) runs the same inputs beside the product, on a bounded sample.  Test / bench infrastructure only.
"""
import ctypes as C
import os
import re
import subprocess
import time

import numpy as np

import wsynth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0


def _dec_weight_bytes(shape, bytes_per_weight):
    d, Ld, nv = shape["d"], shape["dec"], shape["n_vocab"]
    return bytes_per_weight * (14 * Ld * d * d + nv * d)


def _kv_bytes(shape, n_past):
    d, Ld = shape["d"], shape["dec"]
    return 4 * Ld * 1500 * d, 4 * Ld * d * n_past          # cross K/V of one chunk, self K/V of one row


def _segs(st):
    return [(s["t0"], s["t1"], s["ids"]) for s in st.segments()]


def _probe_rows(lib, ctx, states, B, n_past, iters=30):
    VP = C.c_void_p
    lib.whisper_amd_rows_step_probe.argtypes = [VP, C.POINTER(VP), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    arr = (VP * B)(*[VP(s.ptr) for s in states[:B]])
    ms = C.c_float(0)
    rc = lib.whisper_amd_rows_step_probe(ctx.ptr, arr, B, n_past, iters, C.byref(ms))
    return (ms.value if rc == 0 and ms.value > 0 else None), rc


def rows_roofline(lib, ctx, states, shape, B, n_past, same_chunk, bytes_per_weight=2.0):
    """Device time of one B-row decoder pass (HIP events around back-to-back launches) against its algorithmic bytes: every weight row once
    (W), the encoder K/V of every DIFFERENT chunk among the rows, the self K/V of every row (W + B (KVx + KVs))."""
    ms, rc = _probe_rows(lib, ctx, states if not same_chunk else [states[0]] * B, B, n_past)
    if ms is None:
        return {"error": "rows_step_probe rc %d" % rc}
    kvx, kvs = _kv_bytes(shape, n_past)
    n_chunks = 1 if same_chunk else len({s.ptr for s in states[:B]})
    nbytes = int(_dec_weight_bytes(shape, bytes_per_weight) + n_chunks * kvx + B * kvs)
    gbs = nbytes / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters where this very shape was collected (tools/profile_gpu_r03b.sh -> profiles/r03_rows_pmc.json, the kernel's final
    # state: ggml-small, 8 rows of 8 chunks at n_past 110 / 5 rows of one chunk at n_past 64; 2 x FETCH_SIZE + WRITE_SIZE)
    # - off-line, not in this run
    traffic = src = None
    for fn in ("r03_rows_pmc.json",):
        try:
            import json
            pj = json.load(open(os.path.join(ROOT, "profiles", fn)))
            key = "rows8" if (B == 8 and not same_chunk) else "rows5" if (B == 5 and same_chunk) else None
            if key and shape["d"] == 768 and bytes_per_weight == 2.0:
                traffic = int(pj[key]["traffic_MB_per_launch"] * 1e6); src = "profiles/%s (collected off-line)" % fn
                break
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return {"bound": "hbm", "traffic_source": src, "kernel": "k_decode_rows: %d token rows in ONE persistent launch (wa_rows.hip), n_past=%d, %s" %
                                      (B, n_past, "rows of one chunk (beams)" if same_chunk else "%d different chunks" % n_chunks),
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "bytes_per_step": nbytes, "ms_per_step_device": round(ms, 4)}


# ------------------------------------------------------------------------------------------------------------
def config4(W, lib, ref, hip, nthr, with_cpu=True):
    shape = wsynth.SHAPES["medium"]
    t_gen = time.perf_counter()
    mp = wsynth.model_path("medium")
    t_gen = time.perf_counter() - t_gen
    out = {"workload": "ggml-medium-shaped synthetic F16 model (d 1024, 24 + 24 layers), beam_size 5, language zh, DTW token timestamps "
                       "(N top-most 2 layers), one 30 s chunk; max_tokens 48 per window in the run compared with the CPU reference: a random "
                       "model writes one 220-token segment, on which the REFERENCE's DTW pass runs out of its dtw_mem_size",
           "model_write_s": round(t_gen, 1)}
    pcm = wsynth.synth_audio(480000, 5)
    a = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib, dtw_preset=1, dtw_n_top=2), lib=lib)
    kw = dict(beam_size=5, temperature_inc=0.0, language="zh")
    sa = a.create_state()
    for tag, extra in (("capped48", dict(max_tokens=48)), ("full_window", dict())):
        fp = W.FullParams(lib, 1, **dict(kw, **extra))
        sa.full(fp, pcm); hip.sync()
        r0 = sa.rows_stats()
        t1 = time.perf_counter()
        n_rep = 2
        for _ in range(n_rep):
            sa.full(fp, pcm)
        hip.sync()
        dt = (time.perf_counter() - t1) / n_rep
        r1 = sa.rows_stats()
        ntok = sum(len(s["ids"]) for s in sa.segments())
        out[tag] = {"value": round(30.0 / dt, 2), "unit": "x real-time", "ms_per_chunk": round(1e3 * dt, 1), "tokens": ntok,
                    "tokens_per_s": round(ntok / dt, 1), "decoder_passes_one_launch": (r1[0] - r0[0]) // n_rep,
                    "decoder_passes_sent_back": (r1[1] - r0[1]) // n_rep}
        if tag == "capped48":
            got = _segs(sa)
    out["roofline"] = rows_roofline(lib, a, [sa], shape, 5, 64, True)
    sa.free()
    if with_cpu and ref is not None:
        r = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False, dtw_preset=1, dtw_n_top=2), lib=ref)
        sr = r.create_state()
        t1 = time.perf_counter()
        sr.full(W.FullParams(ref, 1, n_threads=nthr, max_tokens=48, **kw), pcm)
        rdt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(30.0 / rdt, 3), "unit": "x real-time", "cores": nthr, "kind": "reference",
                               "sample": "the capped48 run: one 30 s chunk, same file / audio / parameters, %.1f s" % rdt,
                               "segments_identical_to_gpu": _segs(sr) == got}
        sr.free(); r.free()
    a.free()
    return out


# ------------------------------------------------------------------------------------------------------------
def _bench_program(exe, mp, nthr, timeout):
    r = subprocess.run([exe, "-m", mp, "-t", str(nthr)], capture_output=True, timeout=timeout)
    err = r.stderr.decode(errors="replace") + r.stdout.decode(errors="replace")
    if r.returncode != 0:
        return {"error": "rc %d: %s" % (r.returncode, err[-300:])}
    res = {}
    for key, lab in (("encode", "Enc_ms"), ("decode", "Dec_ms_per_token"), ("batchd", "Bch5_ms_per_token"), ("prompt", "PP_ms_per_token")):
        m = re.search(r"%s time =\s*([0-9.]+) ms /\s*(\d+) runs \(\s*([0-9.]+) ms per run\)" % key, err)
        if m:
            res[lab] = float(m.group(3))
    return res


def _stream(W, lib, ctx, n_threads=None):
    """examples/stream/stream.cpp:311-335 as tools/gen_golden_quant.stream_run: 4 windows of 6 s, one every 3 s (15 s of audio)."""
    pcm = wsynth.synth_audio(16000 * 15, 9)
    st = ctx.create_state()
    out, prompt = [], []
    t1 = time.perf_counter()
    for it in range(4):
        win = np.ascontiguousarray(pcm[it * 48000: it * 48000 + 96000])
        kw = dict(best_of=1, temperature_inc=0.0, single_segment=True, max_tokens=32, audio_ctx=768, no_context=True)
        if n_threads:
            kw["n_threads"] = n_threads
        if prompt:
            kw["prompt_tokens"] = prompt
        st.full(W.FullParams(lib, 0, **kw), win)
        sg = _segs(st)
        out.append(sg)
        prompt = [i for s_ in sg for i in s_[2]][-16:]
    dt = time.perf_counter() - t1
    st.free()
    return out, dt


def config5(W, lib, ref, hip, nthr, with_cpu=True):
    shape = wsynth.SHAPES["large-v3"]
    t_gen = time.perf_counter()
    mp = wsynth.quant_model_path("large-v3", "q5_0")
    t_gen = time.perf_counter() - t_gen
    out = {"workload": "ggml-large-v3-shaped synthetic model (d 1280, 32 + 32 layers, 128 mels) quantised to Q5_0 by the reference's quantizer "
                       "(%d MB): whisper-bench protocol + streaming pattern (6 s windows every 3 s, audio_ctx 768, 32 tokens, carried prompt)" % (os.path.getsize(mp) >> 20),
           "model_write_s": round(t_gen, 1)}
    exe = os.path.join(ROOT, "oracle", "_ref", "whisper-bench-amd")
    if os.path.exists(exe):
        out["whisper_bench"] = _bench_program(exe, mp, 4, 300)
        out["whisper_bench"]["program"] = "the reference's examples/bench/bench.cpp, unmodified, linked with libwhisper.so (256 x 1 token, 64 x 5 tokens, 16 x 256 tokens)"
    a = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)
    _stream(W, lib, a)          # warm
    got, dt = _stream(W, lib, a)
    ntok = sum(len(s[2]) for w_ in got for s in w_)
    out["streaming"] = {"value": round(15.0 / dt, 2), "unit": "x real-time", "ms_per_window": round(1e3 * dt / 4, 1), "tokens": ntok, "tokens_per_s": round(ntok / dt, 1)}
    # decode-step roofline: Q5_0 weights at 22 bytes per 32 (550.3 + 245.8 MB for large-v3)
    st = a.create_state()
    st.pcm_to_mel(wsynth.synth_audio(480000, 0)); st.encode(0)
    # (a wide quantised model's single-token step IS the several-rows kernel with one row - wa_internal.h: single_via_rows -, k_decode_mega_q is timed beside it)
    ms = C.c_float()
    lib.whisper_amd_decode_step_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    mega_ms = ms.value if lib.whisper_amd_decode_step_probe(a.ptr, st.ptr, 64, 50, C.byref(ms)) == 0 and ms.value > 0 else None
    rows_ms, _rc = _probe_rows(lib, a, [st], 1, 64, 50)
    step_ms = min(x for x in (mega_ms, rows_ms) if x) if (mega_ms or rows_ms) else None
    if step_ms:
        kvx, kvs = _kv_bytes(shape, 64)
        nbytes = int(_dec_weight_bytes(shape, 22.0 / 32.0) + kvx + kvs)
        gbs = nbytes / (step_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": ("k_decode_rows_q_np20 with one token row" if rows_ms and step_ms == rows_ms else "k_decode_mega_q") +
                                                     ": the single-token decoder pass of a quantised model as one launch, n_past=64",
                           "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                           "bytes_per_step": nbytes, "ms_per_step_device": round(step_ms, 4),
                           "ms_k_decode_mega_q": round(mega_ms, 4) if mega_ms else None, "ms_k_decode_rows_q_one_row": round(rows_ms, 4) if rows_ms else None}
    st.free()
    if with_cpu and ref is not None:
        r = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False), lib=ref)
        want, rdt = _stream(W, ref, r, nthr)
        cpu = {"streaming": {"value": round(15.0 / rdt, 3), "unit": "x real-time", "ms_per_window": round(1e3 * rdt / 4, 1)}, "cores": nthr, "kind": "reference",
               "segments_identical_to_gpu": want == got}
        # the bench protocol on a bounded sample: one encode, 8 single tokens, 2 batches of 5, 1 prompt of 256
        rs = r.create_state()
        rs.pcm_to_mel(wsynth.synth_audio(480000, 0), nthr)
        t1 = time.perf_counter(); rs.encode(0, nthr); enc = time.perf_counter() - t1
        tok = [int(r.token_sot())] * 256
        t1 = time.perf_counter()
        for i in range(8):
            rs.decode(tok[:1], i, nthr)
        dec = (time.perf_counter() - t1) / 8
        t1 = time.perf_counter()
        for i in range(2):
            rs.decode(tok[:5], 0, nthr)
        b5 = (time.perf_counter() - t1) / 10
        t1 = time.perf_counter(); rs.decode(tok, 0, nthr); pp = (time.perf_counter() - t1) / 256
        cpu["whisper_bench"] = {"Enc_ms": round(1e3 * enc, 1), "Dec_ms_per_token": round(1e3 * dec, 2), "Bch5_ms_per_token": round(1e3 * b5, 2),
                                "PP_ms_per_token": round(1e3 * pp, 3), "sample": "1 encode, 8 x 1 token, 2 x 5 tokens, 1 x 256 tokens"}
        rs.free(); r.free()
        out["cpu_baseline"] = cpu
    a.free()
    return out

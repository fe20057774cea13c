#!/usr/bin/env python3
"""Golden vectors for the heuristic token-level timestamps and segment wrapping (params.token_timestamps / max_len /
split_on_word, whisper.cpp:8326-8616, 6047-6100): the REFERENCE ENGINE itself (oracle/_ref/libwhisper_ref.so) on the seeded
synthetic s128 model and audio.  Run in the build container; writes tests/golden/s128_token_ts.json (data only)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth  # noqa: E402
import whisper_rs as W  # noqa: E402

TTS_CASES = {
    "token_ts": dict(best_of=1, temperature_inc=0.0, token_timestamps=True),
    "token_ts_maxlen": dict(best_of=1, temperature_inc=0.0, token_timestamps=True, max_len=24),
    "token_ts_maxlen_word": dict(best_of=1, temperature_inc=0.0, token_timestamps=True, max_len=12, split_on_word=True, thold_pt=0.05),
}


def segs(st):
    return [dict(t0=s["t0"], t1=s["t1"], text=s["text"].decode("latin1"), ids=s["ids"], tok_t0=s["tok_t0"], tok_t1=s["tok_t1"],
                 vlen=[float(v) for v in s["vlen"]]) for s in st.segments()]


if __name__ == "__main__":
    ref = W.load_library(os.path.join(ROOT, "oracle", "_ref", "libwhisper_ref.so"))
    W.set_log_callback(ref, None)
    ctx = W.WhisperContext.new_with_params(wsynth.model_path("s128"), W.WhisperContextParameters(ref, use_gpu=False), lib=ref)
    gold = {}
    for tag, kw in TTS_CASES.items():
        for aseed in (0, 1):
            st = ctx.create_state()
            st.full(W.FullParams(ref, 0, n_threads=8, **kw), wsynth.synth_audio(480000, aseed))
            gold["%s_seed%d" % (tag, aseed)] = segs(st)
            st.free()
    json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "s128_token_ts.json"), "w"), indent=1)
    print({k: (len(v), sum(len(s["ids"]) for s in v)) for k, v in gold.items()})

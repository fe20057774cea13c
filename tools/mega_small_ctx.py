"""One-launch decode step vs launch sequence through full() on reduced audio contexts: identical segments / token ids / probabilities."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: None)
bad = 0
for name in sys.argv[1:] or ["s128", "small"]:
    ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib), lib=lib)
    for actx in (1, 7, 31, 32, 33, 50, 64, 100, 257, 1000, 1499):
        pcm = wsynth.synth_audio(16000 * 3, 5)
        res = {}
        for nomega in ("1", "0"):
            os.environ["WHISPER_AMD_NO_MEGA"] = nomega
            st = ctx.create_state()
            rc = 0
            try:
                st.full(W.FullParams(lib, 0, best_of=1, temperature_inc=0.0, audio_ctx=actx, single_segment=True), pcm)
            except W.WhisperError as e:
                rc = e.code
            res[nomega] = (rc, [(s["t0"], s["t1"], s["ids"], s["p"]) for s in st.segments()] if rc == 0 else None)
            st.free()
        ok = res["1"] == res["0"]
        bad += 0 if ok else 1
        print("%s audio_ctx %4d: rc %d, %d tokens, one-launch == launch sequence: %s" % (name, actx, res["0"][0], sum(len(s[2]) for s in (res["0"][1] or [])), ok))
    ctx.free()
sys.exit(1 if bad else 0)

"""One digest over everything a set of decode modes returns (ids, p, plog, times, text), for comparing the backend's environment switches: the
switches are read once per process, so tests/test_parity_r2_gpu.py runs this script once per combination and compares the lines.
usage: [WHISPER_AMD_...=...] python tools/switch_check.py [model=s128]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W

name = sys.argv[1] if len(sys.argv) > 1 else "s128"
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)


def segs(st):
    return [dict(t0=s["t0"], t1=s["t1"], text=s["text"].decode("latin1"), ids=s["ids"], tids=s["tids"],
                 p=[np.float32(x).tobytes().hex() for x in s["p"]], plog=[np.float32(x).tobytes().hex() for x in s["plog"]]) for s in st.segments()]


out = {}
# four chunks together (the lock-step group; quantised and F16 alike)
sts = [ctx.create_state() for _ in range(4)]
W.full_batch(ctx, sts, W.FullParams(lib, 0, best_of=1, temperature_inc=0.0), [wsynth.synth_audio(480000, 20 + i) for i in range(4)])
out["batch4"] = [segs(s) for s in sts]
# one chunk alone: greedy (the overlap window), beam 5, best_of 3 on the temperature ladder
st = sts[0]
for tag, fp in (("greedy", W.FullParams(lib, 0, best_of=1, temperature_inc=0.0)),
                ("beam5", W.FullParams(lib, 1, beam_size=5, temperature_inc=0.0)),
                ("best_of3", W.FullParams(lib, 0, best_of=3, temperature=0.4, temperature_inc=0.2))):
    st.full(fp, wsynth.synth_audio(480000, 31))
    out[tag] = segs(st)
print("digest", hashlib.sha256(json.dumps(out, sort_keys=True).encode()).hexdigest(), "tokens", sum(len(s["ids"]) for k in out for ss in (out[k] if k == "batch4" else [out[k]]) for s in ss))

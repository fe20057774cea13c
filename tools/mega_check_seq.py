"""As mega_check.py, but the two states run one after the other (no other kernel in flight beside the one-launch step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"; n_tok = int(sys.argv[2]) if len(sys.argv) > 2 else 24
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
ctx = W.WhisperContext.new_with_params(wsynth.model_path(name), W.WhisperContextParameters(lib), lib=lib)
pcm = wsynth.synth_audio(480000, 0); sot = ctx.token_sot(); prompt = [sot, sot + 1, sot + 102]
outs = []
for mode in ("1", "0"):
    os.environ["WHISPER_AMD_NO_MEGA"] = mode
    st = ctx.create_state(); st.pcm_to_mel(pcm); st.encode(0); st.decode(prompt, 0)
    toks = [1000 + 37 * i for i in range(n_tok)]; rows = []
    for i, t in enumerate(toks):
        st.decode([t], len(prompt) + i); rows.append(st.get_logits_last(1).copy())
    outs.append(rows); st.free()
bad = sum(not np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(*outs))
print("mega_check_seq %s: %d tokens, %d mismatching" % (name, n_tok, bad))

"""One-launch decode step (wa_mega.hip) against the launch-sequence path: bit-exact logits, token by token.

usage: python tools/mega_check.py [model=small] [n_tokens=24]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W

name = sys.argv[1] if len(sys.argv) > 1 else "small"
n_tok = int(sys.argv[2]) if len(sys.argv) > 2 else 24
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)      # "small:q5_0" = the quantised file
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)
os.environ["WHISPER_AMD_NO_MEGA"] = "1"
ref = ctx.create_state()
os.environ["WHISPER_AMD_NO_MEGA"] = "0"
meg = ctx.create_state()
pcm = wsynth.synth_audio(480000, 0)
for st in (ref, meg):
    st.pcm_to_mel(pcm); st.encode(0)
sot = ctx.token_sot()
prompt = [sot, sot + 1, sot + 102]
bad = 0
for st in (ref, meg):
    st.decode(prompt, 0)
tok = int(np.argmax(ref.get_logits_last(len(prompt))[:50000]))
for i in range(n_tok):
    ref.decode([tok], len(prompt) + i); meg.decode([tok], len(prompt) + i)
    a = ref.get_logits_last(1); b = meg.get_logits_last(1)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if not same:
        bad += 1
        d = np.abs(a - b); print("token %d: MISMATCH max|d|=%g at %d  (%d differing)  nan=%d" % (i, np.nanmax(d), int(np.nanargmax(d)), int((a != b).sum()), int(np.isnan(b).sum())))
    tok = int(np.argmax(a[:50000]))
print("mega_check %s: %d tokens, %d mismatching" % (name, n_tok, bad))
sys.exit(1 if bad else 0)

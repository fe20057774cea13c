"""Several-rows one-launch decode step (wa_rows.hip) against the one-row step and the launch sequence.

1. B identical rows: every row's hand-off granules must equal the one-row step's (whisper_amd_mega_debug), edge by edge, and every
   logits row the launch sequence's - localises a difference to a phase.
2. Small batches through whisper_decode (n tokens of one sequence, causal mask, logits of the last row): bit-equal to the launch sequence,
   and served by the one-launch form (whisper_amd_rows_stats).
3. Timing of the step (rows of one chunk / rows of different chunks).

usage: python tools/rows_check.py [model=small] [steps=4]
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W

name = sys.argv[1] if len(sys.argv) > 1 else "small"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)
os.environ["WHISPER_AMD_NO_MEGA"] = "1"; ref = ctx.create_state()
os.environ["WHISPER_AMD_NO_MEGA"] = "0"; meg = ctx.create_state()
others = [ctx.create_state() for _ in range(3)]
pcm = wsynth.synth_audio(480000, 0)
for i, st in enumerate([ref, meg] + others):
    st.pcm_to_mel(pcm if i < 2 else wsynth.synth_audio(480000, i)); st.encode(0)
sot = ctx.token_sot(); prompt = [sot, sot + 1, sot + 102]
L, d, nv = ctx.model_n_text_layer(), ctx.model_n_text_state(), ctx.n_vocab()
names = ["QKV", "AO", "X1", "QC", "AO2", "X2", "HF", "X3"]
sizes = [3 * d // 2, d // 2, d, d // 2, d // 2, d, 2 * d, d]
VP = C.c_void_p
lib.whisper_amd_mega_debug.argtypes = [VP, VP, C.c_int, C.c_int, VP, VP]
lib.whisper_amd_rows_debug.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, VP, VP]
lib.whisper_amd_rows_stats.argtypes = [VP, C.POINTER(C.c_long)]
lib.whisper_amd_rows_step_probe.argtypes = [VP, C.POINTER(VP), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
print("rows enabled:", lib.whisper_amd_rows_enabled(VP(meg.ptr)))
bad = 0

# ---- 1. identical rows against the one-row step ----
os.environ["WHISPER_AMD_NO_MEGA"] = "1"
for st in (ref, meg): st.decode(prompt, 0)
n_past = len(prompt)
toks = [1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000][:steps]
for step, tok in enumerate(toks):
    ref.decode([tok], n_past)
    want = ref.get_logits_last(1).copy()
    g1 = np.zeros(L * 8 * 2 * d, dtype=np.uint64); l1 = np.zeros(nv, dtype=np.float32)
    rc = lib.whisper_amd_mega_debug(ctx.ptr, meg.ptr, tok, n_past, g1.ctypes.data, l1.ctypes.data)
    g1 = g1.reshape(L, 8, 2 * d)
    for B in (1, 2, 5, 8):
        gB = np.zeros(L * 8 * B * 2 * d, dtype=np.uint64); lB = np.zeros(B * nv, dtype=np.float32)
        rcB = lib.whisper_amd_rows_debug(ctx.ptr, meg.ptr, B, tok, n_past, gB.ctypes.data, lB.ctypes.data)
        gB = gB.reshape(L, 8, B, 2 * d); lB = lB.reshape(B, nv)
        first = None
        for l in (range(L) if ":" not in name else []):      # (quantised models: other granule formats than the one-row step's - logits only)
            for e in range(8):
                for b in range(B):
                    a = g1[l, e, :sizes[e]] & 0xffffffff; v = gB[l, e, b, :sizes[e]] & 0xffffffff
                    if not np.array_equal(a, v) and first is None:
                        idx = np.nonzero(a != v)[0]
                        first = (l, names[e], b, len(idx), idx[:6].tolist(), [hex(int(x)) for x in a[idx[:3]]], [hex(int(x)) for x in v[idx[:3]]])
        okl = all(np.array_equal(lB[b].view(np.uint32), want.view(np.uint32)) for b in range(B))
        if rc != 0 or rcB != 0 or first or not okl: bad += 1
        print("step %d n_past %d B %d: rc %d/%d  logits %s launch sequence; first differing edge vs one-row step: %s" % (step, n_past, B, rc, rcB, "==" if okl else "!=", first))
    meg.decode([tok], n_past)      # (launch sequence: NO_MEGA is still set for this state? no - meg was created with the one-launch step; its cell is the same either way)
    n_past += 1

# ---- 2. small batches through whisper_decode ----
os.environ["WHISPER_AMD_NO_MEGA"] = "0"
st0 = (C.c_long * 2)(); lib.whisper_amd_rows_stats(VP(meg.ptr), st0)
for n in (2, 3, 5, 8):
    batch = [3000 + 17 * i for i in range(n)]
    ref.decode(batch, n_past); meg.decode(batch, n_past)
    a = ref.get_logits_last(n); b = meg.get_logits_last(n)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if not same: bad += 1
    print("batch of %d at n_past %d: logits %s  (max |d| %g)" % (n, n_past, "==" if same else "!=", float(np.nanmax(np.abs(a - b)))))
    n_past += n
st1 = (C.c_long * 2)(); lib.whisper_amd_rows_stats(VP(meg.ptr), st1)
print("passes served by the several-rows step: %d, sent back to the launch sequence: %d" % (st1[0] - st0[0], st1[1] - st0[1]))
if st1[0] - st0[0] != 4: bad += 1

# ---- 3. timing ----
for B in (2, 4, 5, 8):
    for mode, sts in (("one chunk", [meg] * B), ("chunks", ([meg] + others) * 2)):
        arr = (VP * B)(*[VP(s.ptr) for s in sts[:B]])
        ms = C.c_float(0)
        rc = lib.whisper_amd_rows_step_probe(ctx.ptr, arr, B, 64, 50, C.byref(ms))
        print("B %d rows of %-9s: rc %d  %.3f ms / step" % (B, mode, rc, ms.value))
        if rc != 0: bad += 1
print("rows_check %s: %d problems" % (name, bad))
sys.exit(1 if bad else 0)

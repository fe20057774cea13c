"""Timeline of one workgroup of the several-rows one-launch step (wa_rows.hip: MB_T stamps, 100 MHz wall clock).
usage: WHISPER_AMD_ROWS_TRACE=<workgroup> python tools/rows_trace.py [model=small] [B=5] [n_past=64]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W
name = sys.argv[1] if len(sys.argv) > 1 else "small"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_past = int(sys.argv[3]) if len(sys.argv) > 3 else 64
os.environ.setdefault("WHISPER_AMD_ROWS_TRACE", "0")
lib = W.load_library(os.environ.get("WA_LIB")); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)
st = ctx.create_state()
st.pcm_to_mel(wsynth.synth_audio(480000, 0)); st.encode(0)
L, d, nv = ctx.model_n_text_layer(), ctx.model_n_text_state(), ctx.n_vocab()
VP = C.c_void_p
lib.whisper_amd_rows_debug.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, VP, VP]
rc = lib.whisper_amd_rows_debug(ctx.ptr, st.ptr, B, 1000, n_past, None, None)
print("rows_debug rc", rc, "enabled", lib.whisper_amd_rows_enabled(VP(st.ptr)), flush=True)
t = np.fromfile(os.path.join(ROOT, "gpurun_out", "rows_trace.bin"), dtype=np.uint32).astype(np.int64)
names = ["LN1", "qkv", "self", "gather AO", "out", "LN2", "cq", "cross", "gather AO2", "co", "LN3", "fc1", "gather HF", "fc2"]
print("rc", rc, "workgroup", os.environ["WHISPER_AMD_ROWS_TRACE"], "B", B, "n_past", n_past)
tot = np.zeros(14)
for l in range(L):
    s = t[l * 32:l * 32 + 14]; nxt = t[(l + 1) * 32]
    dt = np.diff(np.append(s, nxt)) / 100.0
    tot += dt
    if l in (0, 1, L // 2, L - 1): print("layer %2d: " % l + "  ".join("%s %.2f" % (n, x) for n, x in zip(names, dt)) + "   | %.1f us" % dt.sum())
cn = ["issue loads", "q arrived", "barrier", "scores", "max exch", "exp", "sum exch", "p16", "PV", "gather parts", "finish"]
ct = np.zeros(10)
for l in range(L): ct += np.diff(t[l * 32 + 16:l * 32 + 27]) / 100.0
print("cross unit (mean): " + "  ".join("%s>%s %.2f" % (cn[i], cn[i + 1], x) for i, x in enumerate(ct / L)))
bw = np.zeros(4)
for l in range(L):
    bw += np.array([t[l*32+14]-t[l*32+4], t[l*32+27]-t[l*32+6], t[l*32+15]-t[l*32+11], t[l*32+28]-t[l*32+13]]) / 100.0
print("wait at the barrier in front of the products (mean): out %.2f  cq %.2f  fc1 %.2f  fc2 %.2f" % tuple(bw / L))
ln = np.zeros(3)
for l in range(1, L): ln += np.array([t[l*32+29]-t[l*32], t[l*32+30]-t[l*32+29], t[l*32+1]-t[l*32+30]]) / 100.0
print("LN1 (mean over layers >= 1): wait for the row %.2f  sums + mean certificate %.2f  squares, normalise, store + barrier %.2f" % tuple(ln / max(1, L - 1)))
print("LN1 rows of the traced wave whose variance went through the in-order sum: %d of %d layers" % (sum(1 for l in range(1, L) if t[l*32+31] != 0), L - 1))
sn = ["entry", "q|k|v in", "barrier", "scores + max", "barrier", "soft-max", "p16 + leftover V, barrier", "P V", "finish", "barrier"]
su = np.zeros(9)
for l in range(L): su += np.diff(t[4096 + l * 16:4096 + l * 16 + 10]) / 100.0
print("self unit (mean): " + "  ".join("%s>%s %.2f" % (sn[i], sn[i + 1], x) for i, x in enumerate(su / L)))
print("mean   : " + "  ".join("%s %.2f" % (n, x) for n, x in zip(names, tot / L)) + "   | %.1f us per layer" % (tot.sum() / L))
print("final LayerNorm + logits: %.1f us;  whole step %.1f us" % ((t[L * 32 + 1] - t[L * 32]) / 100.0, (t[L * 32 + 1] - t[0]) / 100.0))

# every workgroup at one layer: when each of them passed the stamps (us after the first workgroup entered the layer)
g = t[8192:8192 + 256 * 16].reshape(256, 16)[:, :14]
g = g[g[:, 0] > 0]
t0 = g[:, 0].min()
print("all %d workgroups at the traced layer: stamp = time the phase STARTED on the workgroup (min / median / max, us)" % len(g))
for k, n in enumerate(names):
    c = (g[:, k] - t0) / 100.0
    print("  %-10s %6.2f %6.2f %6.2f" % (n, c.min(), np.median(c), c.max()))

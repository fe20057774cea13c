"""Timeline of the one-launch decode step: per phase, when one workgroup of each role saw its input / published."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import numpy as np
import wsynth, whisper_rs as W
os.environ["WHISPER_AMD_MEGA_DBG"] = "1"
name = sys.argv[1] if len(sys.argv) > 1 else "small"
n_past = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: sys.stderr.write(t) if l >= 3 else None)
mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)      # "small:q5_0" = the quantised file
ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib)
st = ctx.create_state()
st.pcm_to_mel(wsynth.synth_audio(480000, 0)); st.encode(0)
L, d, nv, H = ctx.model_n_text_layer(), ctx.model_n_text_state(), ctx.n_vocab(), ctx.model_n_text_head()
lib.whisper_amd_mega_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
g = np.zeros(L * 8 * 2 * d, dtype=np.uint64); lg = np.zeros(nv, dtype=np.float32)
for it in range(3):
    rc = lib.whisper_amd_mega_debug(ctx.ptr, st.ptr, 1000, n_past, g.ctypes.data, lg.ctypes.data)
tr = np.fromfile("gpurun_out/mega_dbg.bin", dtype=np.uint32)[L * H * 5120:].reshape(-1, 8)
names = ["P1 ln+qkv", "P3 out", "P4 ln+cq", "P6 cout", "P7 ln+fc1", "P8 fc2", "S  self", "C  cross"]
t0 = int(tr[0, 2])            # layer 0 P1: LayerNorm of the embeddings done
us = lambda t: (int(t) - t0) / 100.0
print("rc", rc, " times in us since layer-0 LN ready; [input seen (polls)] [lds ready] [published]")
order = [0, 6, 1, 2, 7, 3, 4, 5]
for l in range(L):
    for p in order:
        r = tr[l * 8 + p]
        extra = "  scores %.2f softmax %.2f pv %.2f gathered %.2f" % (us(r[4]), us(r[5]), us(r[6]), us(r[7])) if p >= 6 else ("  ln: sums %.2f squares %.2f" % (us(r[4]), us(r[5])) if p in (0, 2, 4) else ("  gathered+barrier %.2f dot %.2f" % (us(r[4]), us(r[5])) if p == 5 else ""))
        print("L%02d %-10s in %8.2f (%4d polls)  ready %8.2f  pub %8.2f%s" % (l, names[p], us(r[0]), r[1], us(r[2]) if r[2] else 0.0, us(r[3]), extra))
r = tr[L * 8]
print("final      in %8.2f (%4d polls)  ready %8.2f  done %8.2f" % (us(r[0]), r[1], us(r[2]), us(r[3])))
print("final done %.2f us after layer-0 LayerNorm ready" % us(r[3]))
print("entry %8.2f  token picked %8.2f  (kernel start to layer-0 LayerNorm ready: %.2f us)" % (us(r[6]), us(r[7]), -us(r[6])))

# per-workgroup stamps of one layer's P3 -> P4 (MG_WGTRACE_LAYER): how far apart the workgroups are
flat = np.fromfile("gpurun_out/mega_dbg.bin", dtype=np.uint32)[L * H * 5120:]
nG = 256 - 5 * H
w = flat[1024:1024 + 8 * nG].reshape(nG, 8).astype(np.int64)
w0 = tr[4 * 8 + 2]         # workgroup 0 writes the ordinary slots
w[0, 0], w[0, 1], w[0, 2], w[0, 4], w[0, 5] = w0[0], w0[1], w0[2], w0[4], w0[5]
def stat(name, col, rows=slice(None)):
    v = np.array([us(x) for x in w[rows, col]])
    v = v[np.isfinite(v)]
    print("%-28s min %8.2f  median %8.2f  p90 %8.2f  max %8.2f  (argmax wg %d)" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max(), int(np.argmax(v))))
print("layer 4, all %d GEMV workgroups:" % nG)
stat("P3 out-projection published", 3); stat("P4 input complete (sweep)", 0); stat("P4 sums", 4); stat("P4 squares", 5); stat("P4 LayerNorm ready", 2); stat("P4 cross query published", 6)
pub = np.array([us(x) for x in w[:, 3]]); inn = np.array([us(x) for x in w[:, 0]])
print("last P3 publish -> first / median / last 'input complete': %.2f / %.2f / %.2f us" % (inn.min() - pub.max(), np.median(inn) - pub.max(), inn.max() - pub.max()))
print("polls per workgroup: median %d max %d" % (np.median(w[:, 1]), w[:, 1].max()))
f = flat[3000:3003]
print("cross-attention finish (layer 4, head 0): start %.2f  tree done %.2f  leftovers done %.2f" % (us(f[0]), us(f[1]), us(f[2])))
f = flat[3004:3007]
print("self-attention finish  (layer 4, head 0): start %.2f  tree done %.2f  leftovers done %.2f" % (us(f[0]), us(f[1]), us(f[2])))

w = flat[4096:4096 + 8 * nG].reshape(nG, 8).astype(np.int64)
w0 = tr[4 * 8 + 4]
w[0, 0], w[0, 1], w[0, 2], w[0, 4], w[0, 5], w[0, 6] = w0[0], w0[1], w0[2], w0[4], w0[5], w0[3]
w[0, 3] = tr[4 * 8 + 3][3]
print("layer 4, P6 -> P7:")
stat("P6 out-projection published", 3); stat("P7 input complete (sweep)", 0); stat("P7 sums", 4); stat("P7 LayerNorm ready", 2); stat("P7 FC1 published (wave 1)", 6)
hp = np.array([us(x) for x in flat[3100:3100 + H]]); hi = np.array([us(x) for x in flat[3200:3200 + H]])
print("cross-attention per head: query seen %s" % " ".join("%.2f" % x for x in hi))
print("cross-attention per head: published  %s" % " ".join("%.2f" % x for x in hp))
f = flat[3010:3017]
print("cross-attention (layer 4, head 0, workgroup 0): query in LDS %.2f | scores + wave max done %.2f | barrier %.2f | maxima exchanged %.2f | exp + sums done %.2f | total exchanged %.2f | p16 written + barrier %.2f" % tuple(us(x) for x in f))
f = flat[3020:3024].astype(np.int64)
cyc = (int(f[2]) - int(f[0])) & 0xffffffff; wall = (int(f[3]) - int(f[1])) & 0xffffffff
print("shader clock during the launch: %d cycles in %.2f us = %.0f MHz" % (cyc, wall / 100.0, cyc / (wall / 100.0)))

import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))
import wsynth, whisper_rs as W
lib = W.load_library(); W.set_log_callback(lib, lambda l, t: None)
for name in sys.argv[1:]:
    mp = wsynth.quant_model_path(*name.split(":")) if ":" in name else wsynth.model_path(name)
    open(mp, "rb").read()      # page cache
    t = time.perf_counter(); ctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib), lib=lib); dt = time.perf_counter() - t
    print("%s: %.0f MB, load %.2f s" % (name, os.path.getsize(mp) / 1e6, dt)); ctx.free()

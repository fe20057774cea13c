"""bench.py's N > 1 control flow end to end on CPU: two ranks under torch.distributed.run with the gloo backend and the stub engine
(--stub: no HIP library) - rank / chunk partition, the weight-image broadcast into the buffer the loader would parse, barriers, the MAX
over ranks of the timed region, the token SUM and the config-3 object (8 chunks per GPU) are bench.py's real code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    port = 29600 + (os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--stub"]
    r = subprocess.run(cmd, capture_output=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-1500:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_gloo_stub():
    out = _run(2)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["tokens_decoded_per_step"] == 14           # 7 "tokens" per chunk, one chunk per rank, summed over ranks
    assert out["config"]["parallelism"] == "chunk-dp2"
    # whole-job value: audio of ALL ranks / MAX over ranks of the wall time (rank 1's stub step is the slower one: 4 ms per chunk)
    assert 0 < out["value"] <= 30.0 * 2 * 2 / (2 * 0.004)
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 60.0) < 0.5
    c3 = out["config3"]
    assert c3["chunks"] == 16 and c3["tokens"] == 7 * 16 and c3["value"] > 0
    assert out["stub"]["chunk_ids_rank0"] == [0] and out["stub"]["image_checksum"] > 0


def test_single_rank_stub():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "1", "--warmup", "0"], capture_output=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-800:]
    j = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config3"]["chunks"] == 8

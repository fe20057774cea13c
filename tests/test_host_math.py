"""Host-side arithmetic of the sampling path that no longer calls libm per vocabulary entry (whisper-rust_amd/csrc/wa_expf8.h):
the 8-wide restatement of glibc's expf and the ordered F32 sums / probability rows built on it must equal the reference's plain
libm loops bit for bit (whisper.cpp:6109-6143, 6309-6333).  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vector_expf_and_ordered_sums_equal_libm(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path / "sampling_math")
    # built like the host sources of the library: AVX2 on the command line, FMA only inside the functions that carry the target attribute
    subprocess.check_call(["g++", "-O2", "-mavx2", "-pthread", os.path.join(ROOT, "tests", "native", "sampling_math.cpp"),
                           "-I", os.path.join(ROOT, "whisper-rust_amd", "csrc"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "expf8: 0 mismatches" in out.stdout or "not usable" in out.stdout, out.stdout

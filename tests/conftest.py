"""pytest configuration: markers, import paths, shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI surface (no compute call needs a GPU).
`-m gpu`       : parity tests proper - every one calls the product through the C ABI
                 (whisper-rust_amd/libwhisper.so) and compares with golden vectors / the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "whisper-rust_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libwhisper_ref.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
PRODUCT_LIB = os.path.join(ROOT, "whisper-rust_amd", "libwhisper.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def wrs():
    import whisper_rs
    return whisper_rs


@pytest.fixture(scope="session")
def amd_lib(wrs):
    """The product. No fallback: a missing library is a hard failure, not a skip."""
    lib = wrs.load_library(PRODUCT_LIB)
    wrs.set_log_callback(lib, lambda lvl, txt: sys.stderr.write(txt) if lvl >= 3 else None)
    return lib


@pytest.fixture(scope="session")
def ref_lib(wrs):
    """The reference engine compiled from /root/reference by oracle/Makefile (travels as a prebuilt .so)."""
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libwhisper_ref.so not built")
    lib = wrs.load_library(REF_LIB)
    wrs.set_log_callback(lib, None)
    return lib

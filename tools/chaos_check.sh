#!/bin/bash
# On the GPU box: the one-launch decode step with random ~25 us stalls in front of its products (libwhisper_chaos.so = wa_mega.hip built with
# -DMG_CHAOS, `make -C whisper-rust_amd libwhisper_chaos.so`) against the launch sequence, bit for bit, on several shapes.  The kernel of
# commit f39fc2a (LayerNorm outputs and gathered inputs in one LDS area) fails this on every token; the test suite runs the same check
# (tests/test_parity_gpu.py::test_one_launch_step_under_stalls).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for m in small m1024 w1280 small:q5_0; do WA_LIB=$ROOT/whisper-rust_amd/libwhisper_chaos.so python3 $ROOT/tools/mega_check.py $m 16 2>&1 | tail -1; done

#!/bin/bash
# usage: tools/make_rust_sys_tree.sh DEST
# Lays out DEST/whisper.cpp: a drop-in for the vendored `sys/whisper.cpp` directory of whisper-rs (see whisper-rust_amd/rust_sys/CMakeLists.txt
# and INTEGRATION.md): CMake project + the backend's sources + the two headers bindgen reads through sys/wrapper.h.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DEST=${1:?usage: make_rust_sys_tree.sh DEST}/whisper.cpp
rm -rf "$DEST"; mkdir -p "$DEST/src" "$DEST/include" "$DEST/ggml/include"
cp "$ROOT/whisper-rust_amd/rust_sys/CMakeLists.txt" "$DEST/CMakeLists.txt"
cp "$ROOT"/whisper-rust_amd/csrc/*.cpp "$ROOT"/whisper-rust_amd/csrc/*.hip "$ROOT"/whisper-rust_amd/csrc/*.h "$DEST/src/"
cp "$ROOT/include/whisper_amd.h" "$ROOT/include/whisper.h" "$DEST/include/"
# the sources include "../include/whisper_amd.h" relative to csrc/: keep that relation
sed -i 's|"\.\./\.\./include/whisper_amd.h"|"../include/whisper_amd.h"|' "$DEST"/src/*.h "$DEST"/src/*.cpp 2>/dev/null || true
cat > "$DEST/ggml/include/ggml.h" <<'H'
/* ggml.h - stands where sys/wrapper.h of whisper-rs looks for it.  The three ggml items whisper-rs binds (ggml_log_level, the log /
 * abort callback types, ggml_log_set, ggml_cpu_has_*) are declared by the backend's own header. */
#include "../../include/whisper_amd.h"
H
echo "$DEST"

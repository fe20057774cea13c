"""C-ABI surface of the product library (no GPU needed: nothing here launches a kernel).

Checks that libwhisper.so loads, exports every symbol include/whisper_amd.h declares, and that the
by-value structs have the layout measured on the reference header (48 / 296 / 56 bytes,
`vad` at offset 260).  The same checks are run against the reference engine when it is available, so the
two libraries are interchangeable behind one binding.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, PRODUCT_LIB, REF_LIB


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "whisper_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*WHISPER_API[^;(]*?\b(\w+)\s*\(", txt, flags=re.M)
    return sorted(set(names) - {"__attribute__"})


def test_header_declares_the_whisper_rs_surface():
    names = set(_declared_symbols())
    # every function whisper-rs actually calls
    needed = """whisper_init_from_file_with_params_no_state whisper_init_from_buffer_with_params_no_state whisper_init_state
    whisper_free whisper_free_state whisper_full_with_state whisper_full_default_params whisper_pcm_to_mel_with_state
    whisper_set_mel_with_state whisper_encode_with_state whisper_decode_with_state whisper_lang_auto_detect_with_state
    whisper_get_logits_from_state whisper_n_len_from_state whisper_full_n_segments_from_state whisper_full_lang_id_from_state
    whisper_full_get_segment_t0_from_state whisper_full_get_segment_t1_from_state whisper_full_get_segment_text_from_state
    whisper_full_get_segment_speaker_turn_next_from_state whisper_full_n_tokens_from_state whisper_full_get_token_text_from_state
    whisper_full_get_token_id_from_state whisper_full_get_token_data_from_state whisper_full_get_token_p_from_state
    whisper_tokenize whisper_n_vocab whisper_n_text_ctx whisper_n_audio_ctx whisper_is_multilingual whisper_model_n_vocab
    whisper_model_n_audio_ctx whisper_model_n_audio_state whisper_model_n_audio_head whisper_model_n_audio_layer
    whisper_model_n_text_ctx whisper_model_n_text_state whisper_model_n_text_head whisper_model_n_text_layer whisper_model_n_mels
    whisper_model_ftype whisper_model_type whisper_token_to_str whisper_model_type_readable whisper_token_eot whisper_token_sot
    whisper_token_solm whisper_token_prev whisper_token_nosp whisper_token_not whisper_token_beg whisper_token_lang
    whisper_token_translate whisper_token_transcribe whisper_print_timings whisper_reset_timings whisper_lang_id
    whisper_lang_max_id whisper_lang_str whisper_lang_str_full whisper_log_set whisper_print_system_info ggml_cpu_has_avx
    ggml_cpu_has_avx2 ggml_cpu_has_fma ggml_cpu_has_f16c ggml_log_set""".split()
    missing = [n for n in needed if n not in names]
    assert not missing, missing


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(PRODUCT_LIB)
    missing = [n for n in _declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layout_matches_reference_header(wrs, amd_lib):
    out = (C.c_size_t * 6)()
    amd_lib.whisper_amd_abi_sizes.argtypes = [C.POINTER(C.c_size_t)]
    amd_lib.whisper_amd_abi_sizes(out)
    assert list(out[:4]) == [48, 296, 56, 260]
    assert C.sizeof(wrs.whisper_context_params) == out[0]
    assert C.sizeof(wrs.whisper_full_params) == out[1]
    assert C.sizeof(wrs.whisper_token_data) == out[2]
    assert wrs.whisper_full_params.vad.offset == out[3]
    assert wrs.whisper_full_params.greedy.offset == out[4]
    assert wrs.whisper_full_params.language.offset == out[5]


def _defaults(wrs, lib, strategy):
    p = lib.whisper_full_default_params(strategy)
    return dict(strategy=p.strategy, n_max_text_ctx=p.n_max_text_ctx, no_context=p.no_context, language=p.language,
                suppress_blank=p.suppress_blank, suppress_nst=p.suppress_nst, temperature=p.temperature,
                max_initial_ts=p.max_initial_ts, length_penalty=p.length_penalty, temperature_inc=p.temperature_inc,
                entropy_thold=p.entropy_thold, logprob_thold=p.logprob_thold, no_speech_thold=p.no_speech_thold,
                best_of=p.greedy.best_of, beam_size=p.beam_search.beam_size, patience=p.beam_search.patience,
                print_progress=p.print_progress, print_timestamps=p.print_timestamps, thold_pt=p.thold_pt,
                thold_ptsum=p.thold_ptsum, grammar_penalty=p.grammar_penalty, vad=p.vad, vad_thr=p.vad_params.threshold,
                vad_pad=p.vad_params.speech_pad_ms, n_threads=p.n_threads, audio_ctx=p.audio_ctx, max_tokens=p.max_tokens)


@pytest.mark.parametrize("strategy", [0, 1])
def test_default_params(wrs, amd_lib, strategy):
    d = _defaults(wrs, amd_lib, strategy)
    # whisper.cpp:5914-6019
    assert d["language"] == b"en" and d["no_context"] and d["suppress_blank"] and not d["suppress_nst"]
    assert d["n_max_text_ctx"] == 16384 and d["temperature"] == 0.0 and d["max_initial_ts"] == 1.0
    assert abs(d["temperature_inc"] - 0.2) < 1e-7 and abs(d["entropy_thold"] - 2.4) < 1e-6
    assert d["logprob_thold"] == -1.0 and abs(d["no_speech_thold"] - 0.6) < 1e-6 and d["length_penalty"] == -1.0
    assert (d["best_of"], d["beam_size"]) == ((5, -1) if strategy == 0 else (-1, 5))
    if os.path.exists(REF_LIB):
        ref = wrs.load_library(REF_LIB)
        assert _defaults(wrs, ref, strategy) == d
        a, b = amd_lib.whisper_context_default_params(), ref.whisper_context_default_params()
        for f, _ in wrs.whisper_context_params._fields_:
            if f != "dtw_aheads":
                assert getattr(a, f) == getattr(b, f), f


def test_language_table(wrs, amd_lib):
    assert amd_lib.whisper_lang_max_id() == 99
    assert amd_lib.whisper_lang_id(b"de") == 2 and amd_lib.whisper_lang_id(b"german") == 2     # include/whisper.h:358-362
    assert amd_lib.whisper_lang_str(2) == b"de" and amd_lib.whisper_lang_str_full(2) == b"german"
    assert amd_lib.whisper_lang_id(b"klingon") == -1 and amd_lib.whisper_lang_str(1000) is None
    if os.path.exists(REF_LIB):
        ref = wrs.load_library(REF_LIB)
        wrs.set_log_callback(ref, None)
        for i in range(100):
            assert amd_lib.whisper_lang_str(i) == ref.whisper_lang_str(i)
            assert amd_lib.whisper_lang_str_full(i) == ref.whisper_lang_str_full(i)
            assert amd_lib.whisper_lang_id(ref.whisper_lang_str(i)) == i


def test_constructor_failures_return_null(wrs, amd_lib, tmp_path):
    """Mirrors the model-free Rust unit tests (src/whisper_ctx_wrapper.rs:490-513, 598-614)."""
    with pytest.raises(wrs.WhisperError):
        wrs.WhisperContext.new_with_params(str(tmp_path / "does-not-exist.bin"), lib=amd_lib)
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a model at all" * 10)
    with pytest.raises(wrs.WhisperError):
        wrs.WhisperContext.new_with_params(str(bad), lib=amd_lib)
    with pytest.raises(wrs.WhisperError):
        wrs.WhisperContext.new_from_buffer_with_params(b"\x00" * 64, lib=amd_lib)
    # use_gpu = false must be refused: the product has no CPU path
    import wsynth
    mp = wsynth.model_path("s128")
    with pytest.raises(wrs.WhisperError):
        wrs.WhisperContext.new_with_params(mp, wrs.WhisperContextParameters(amd_lib, use_gpu=False), lib=amd_lib)


def test_gelu_table_matches_reference_table(amd_lib):
    """The device GELU table is built on the host exactly like ggml's (vec.h:552-585, ggml-cpu.c:3509-3517)."""
    if not os.path.exists(REF_LIB):
        pytest.skip("reference library not built")
    ref = C.CDLL(REF_LIB)
    a = np.zeros(65536, np.uint16)
    b = np.zeros(65536, np.uint16)
    amd_lib.whisper_amd_gelu_table_f16.argtypes = [C.c_void_p]
    ref.ref_shim_gelu_table_f16.argtypes = [C.c_void_p]
    amd_lib.whisper_amd_gelu_table_f16(a.ctypes.data)
    ref.ref_shim_gelu_table_f16(b.ctypes.data)
    finite = ~np.isnan(a.view(np.float16)) & ~np.isnan(b.view(np.float16))
    assert np.array_equal(a[finite], b[finite])
    assert np.array_equal(np.isnan(a.view(np.float16)), np.isnan(b.view(np.float16)))


def test_one_launch_step_role_map(amd_lib):
    """Workgroup -> role map of the one-launch decode step (wa_mega.h: mg_role_of), a pure function: every role index exactly once
    for every head count, and the four cross-attention workgroups of a head 8 workgroups apart (one XCD under round-robin dispatch)."""
    f = amd_lib.whisper_amd_mega_role_of
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    f.restype = None
    for n_wg in (256, 250):                    # 250: not a multiple of 8 -> the plain layout
        for H in (1, 2, 3, 6, 8, 12, 16, 20, 32):
            seen = {}
            for b in range(n_wg):
                role, idx = C.c_int(-1), C.c_int(-1)
                f(n_wg, H, b, C.byref(role), C.byref(idx))
                assert (role.value, idx.value) not in seen
                seen[(role.value, idx.value)] = b
            n_g = n_wg - 5 * H
            assert sorted(i for r, i in seen if r == 0) == list(range(n_g))
            assert sorted(i for r, i in seen if r == 1) == list(range(H))
            assert sorted(i for r, i in seen if r == 2) == list(range(4 * H))
            if n_wg % 8 == 0:
                for h in range(H):
                    assert len({seen[(2, 4 * h + w)] % 8 for w in range(4)}) == 1, (H, h)


def test_by_value_struct_layout_field_by_field(tmp_path):
    """Every field of the structs that cross the ABI by value (whisper_context_params 48 B, whisper_full_params 296 B, whisper_token_data
    56 B, ...): sizeof / alignof / offsetof as a C++ compiler sees them through include/whisper_amd.h == the listing the reference's
    whisper.h gave (tests/golden/abi_layout.txt); where /root/reference is present the reference header is compiled again and
    compared too (tests/native/abi_layout.cpp)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "abi_layout.cpp")
    want = open(os.path.join(ROOT, "tests", "golden", "abi_layout.txt")).read()
    exe = str(tmp_path / "abi_amd")
    subprocess.check_call(["g++", "-std=c++17", "-DABI_HEADER=\"whisper_amd.h\"", "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
    assert subprocess.check_output([exe]).decode() == want
    exe2 = str(tmp_path / "abi_whisper_h")       # through the drop-in name include/whisper.h as well
    subprocess.check_call(["g++", "-std=c++17", "-DABI_HEADER=\"whisper.h\"", "-I" + os.path.join(ROOT, "include"), src, "-o", exe2])
    assert subprocess.check_output([exe2]).decode() == want
    ref_inc = "/root/reference/sys/whisper.cpp"
    if os.path.exists(os.path.join(ref_inc, "include", "whisper.h")):
        exe3 = str(tmp_path / "abi_ref")
        subprocess.check_call(["g++", "-std=c++17", "-DABI_HEADER=\"whisper.h\"", "-I" + ref_inc + "/include", "-I" + ref_inc + "/ggml/include", src, "-o", exe3])
        assert subprocess.check_output([exe3]).decode() == want, "the fixture no longer matches the reference header"


def test_rust_sys_tree_builds_and_links_like_build_rs(tmp_path):
    """f3, the Rust link path: tools/make_rust_sys_tree.sh lays out a drop-in for whisper-rs' vendored `sys/whisper.cpp`; configured and
    built the way sys/build.rs drives CMake (hipcc as compiler = the `hipblas` feature, static libraries, install), it yields the archives
    of build.rs' link line (sys/build.rs:275-301: whisper, ggml, ggml-base, ggml-cpu, + ggml-hip) and the two headers sys/wrapper.h
    includes.  A C program that includes them as wrapper.h does links statically against exactly that line and runs (no GPU call)."""
    import shutil
    import subprocess
    if not shutil.which("cmake") or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("cmake / hipcc not available")
    tree = subprocess.check_output([os.path.join(ROOT, "tools", "make_rust_sys_tree.sh"), str(tmp_path)]).decode().strip()
    assert open(os.path.join(tree, "CMakeLists.txt")).read().count('project("whisper.cpp" VERSION ') == 1        # build.rs greps this line
    build, out = str(tmp_path / "build"), str(tmp_path / "out")
    subprocess.check_call(["cmake", "-S", tree, "-B", build, "-DCMAKE_CXX_COMPILER=/opt/rocm/bin/hipcc", "-DCMAKE_C_COMPILER=/opt/rocm/bin/hipcc",
                           "-DCMAKE_BUILD_TYPE=Release", "-DBUILD_SHARED_LIBS=OFF", "-DGGML_HIP=ON", "-DWHISPER_BUILD_TESTS=OFF",
                           "-DCMAKE_INSTALL_PREFIX=" + out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", build, "--target", "install", "-j", "8"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for a in ("whisper", "ggml", "ggml-base", "ggml-cpu", "ggml-hip"):
        assert os.path.exists(os.path.join(out, "lib", "lib%s.a" % a)), a
    src = tmp_path / "wrapper_user.c"
    src.write_text('#include <include/whisper.h>\n#include <ggml/include/ggml.h>\n#include <stdio.h>\n'
                   'int main(void) { struct whisper_full_params p = whisper_full_default_params(WHISPER_SAMPLING_BEAM_SEARCH);\n'
                   '  struct whisper_context_params c = whisper_context_default_params();\n'
                   '  printf("%d %d %d %zu %zu %s\\n", p.beam_search.beam_size, (int) c.use_gpu, whisper_lang_id("zh"), sizeof p, sizeof c, whisper_print_system_info());\n'
                   '  return 0; }\n')
    exe = str(tmp_path / "wrapper_user")
    subprocess.check_call(["gcc", str(src), "-I" + tree, "-o", exe, "-L" + os.path.join(out, "lib"), "-lwhisper", "-lggml", "-lggml-base", "-lggml-cpu",
                           "-lggml-hip", "-L/opt/rocm/lib", "-lamdhip64", "-lstdc++", "-lpthread", "-lm", "-Wl,-rpath,/opt/rocm/lib"])
    got = subprocess.check_output([exe]).decode().split()
    assert got[:5] == ["5", "1", "1", "296", "48"], got


def test_one_launch_kernels_use_no_scratch(tmp_path):
    """The persistent decode kernels must not touch scratch memory: a spilled register is reloaded behind `s_waitcnt vmcnt(0)`, i.e. behind the wave's
    outstanding weight loads (addresses hoisted out of the layer loop into scratch cost a tenth of every pass until round 3).  Read from the
    code objects the build just made (AMDGPU metadata: private_segment_fixed_size, vgpr_spill_count).  k_decode_mega_q is exempt: its d = 1280 instantiation
    is known to spill ("what is left")."""
    tools = "/opt/rocm/lib/llvm/bin"
    objs = [os.path.join(ROOT, "whisper-rust_amd", "build", n) for n in ("wa_rows_12.o", "wa_rows_16.o", "wa_rows_20.o", "wa_mega.o")]
    if not all(os.path.exists(o) for o in objs) or not os.path.exists(os.path.join(tools, "clang-offload-bundler")):
        pytest.skip("no build tree / LLVM tools here")
    seen = {}
    for o in objs:
        fat, co = str(tmp_path / "fat"), str(tmp_path / "co")
        subprocess.check_call([os.path.join(tools, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, o])
        subprocess.check_call([os.path.join(tools, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        notes = subprocess.check_output([os.path.join(tools, "llvm-readelf"), "--notes", co], text=True)
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split(":", 1)[1].strip()
            elif name and (line.startswith(".private_segment_fixed_size:") or line.startswith(".vgpr_spill_count:")):
                seen.setdefault(name, {})[line.split(":")[0]] = int(line.split(":")[1])
    kernels = [k for k in seen if "k_decode_rows" in k or k.startswith("_Z13k_decode_mega")]
    assert len(kernels) == 7, sorted(seen)
    for k in kernels:
        assert seen[k] == {".private_segment_fixed_size": 0, ".vgpr_spill_count": 0}, (k, seen[k])

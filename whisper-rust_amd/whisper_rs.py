"""Host-side mirror of the whisper-rs safe API over the C ABI (`include/whisper_amd.h`).

The reference's host layer is Rust (`src/whisper_ctx_wrapper.rs`, `src/whisper_state.rs`,
`src/whisper_params.rs`); there is no Rust toolchain in this image, so tests and the bench drive the
C ABI through this thin ctypes mirror instead.  Names, argument meaning and error behaviour follow
whisper-rs:

  WhisperContext.new_with_params(path, params)   src/whisper_ctx_wrapper.rs:35-47
  WhisperContext.create_state()                  src/whisper_ctx_wrapper.rs:438-454
  WhisperState.full(params, pcm)                 src/whisper_state.rs:289-321
  WhisperState.full_n_segments / full_get_segment_text / _t0 / _t1 / token getters
                                                 src/whisper_state.rs:329-606
  WhisperState.pcm_to_mel / set_mel / encode / decode / get_logits / lang_detect
                                                 src/whisper_state.rs:50-260
  FullParams(strategy) + setters                 src/whisper_params.rs:36-803

The same binding can load EITHER library because both export the same ABI:
  * the product:  whisper-rust_amd/libwhisper.so      (HIP; fails loudly when absent)
  * the checker:  oracle/_ref/libwhisper_ref.so       (reference CPU engine; tests/bench only)
Nothing in this file computes anything: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "libwhisper.so")

WHISPER_SAMPLING_GREEDY = 0
WHISPER_SAMPLING_BEAM_SEARCH = 1

# enum whisper_alignment_heads_preset (include/whisper_amd.h)
AHEADS_NONE, AHEADS_N_TOP_MOST, AHEADS_CUSTOM = 0, 1, 2


class WhisperError(RuntimeError):
    """Mirror of whisper-rs `WhisperError` (src/error.rs): carries the C return code."""

    def __init__(self, msg: str, code: int = 0):
        super().__init__(msg)
        self.code = code


# ---------------------------------------------------------------------------------------------------
# C structs (layout checked against sizeof/offsetof exported by the library in tests/test_abi.py)
# ---------------------------------------------------------------------------------------------------
class whisper_ahead(C.Structure):
    _fields_ = [("n_text_layer", C.c_int), ("n_head", C.c_int)]


class whisper_aheads(C.Structure):
    _fields_ = [("n_heads", C.c_size_t), ("heads", C.POINTER(whisper_ahead))]


class whisper_context_params(C.Structure):
    _fields_ = [
        ("use_gpu", C.c_bool),
        ("flash_attn", C.c_bool),
        ("gpu_device", C.c_int),
        ("dtw_token_timestamps", C.c_bool),
        ("dtw_aheads_preset", C.c_int),
        ("dtw_n_top", C.c_int),
        ("dtw_aheads", whisper_aheads),
        ("dtw_mem_size", C.c_size_t),
    ]


class whisper_token_data(C.Structure):
    _fields_ = [
        ("id", C.c_int32), ("tid", C.c_int32),
        ("p", C.c_float), ("plog", C.c_float), ("pt", C.c_float), ("ptsum", C.c_float),
        ("t0", C.c_int64), ("t1", C.c_int64), ("t_dtw", C.c_int64),
        ("vlen", C.c_float),
    ]


class whisper_vad_params(C.Structure):
    _fields_ = [
        ("threshold", C.c_float), ("min_speech_duration_ms", C.c_int), ("min_silence_duration_ms", C.c_int),
        ("max_speech_duration_s", C.c_float), ("speech_pad_ms", C.c_int), ("samples_overlap", C.c_float),
    ]


NEW_SEGMENT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
ENCODER_BEGIN_CB = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_void_p)
ABORT_CB = C.CFUNCTYPE(C.c_bool, C.c_void_p)
LOGITS_FILTER_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.POINTER(whisper_token_data), C.c_int,
                               C.POINTER(C.c_float), C.c_void_p)
LOG_CB = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)


class _greedy(C.Structure):
    _fields_ = [("best_of", C.c_int)]


class _beam(C.Structure):
    _fields_ = [("beam_size", C.c_int), ("patience", C.c_float)]


class whisper_full_params(C.Structure):
    _fields_ = [
        ("strategy", C.c_int),
        ("n_threads", C.c_int), ("n_max_text_ctx", C.c_int), ("offset_ms", C.c_int), ("duration_ms", C.c_int),
        ("translate", C.c_bool), ("no_context", C.c_bool), ("no_timestamps", C.c_bool), ("single_segment", C.c_bool),
        ("print_special", C.c_bool), ("print_progress", C.c_bool), ("print_realtime", C.c_bool),
        ("print_timestamps", C.c_bool),
        ("token_timestamps", C.c_bool), ("thold_pt", C.c_float), ("thold_ptsum", C.c_float), ("max_len", C.c_int),
        ("split_on_word", C.c_bool), ("max_tokens", C.c_int),
        ("debug_mode", C.c_bool), ("audio_ctx", C.c_int),
        ("tdrz_enable", C.c_bool),
        ("suppress_regex", C.c_char_p),
        ("initial_prompt", C.c_char_p), ("prompt_tokens", C.POINTER(C.c_int32)), ("prompt_n_tokens", C.c_int),
        ("language", C.c_char_p), ("detect_language", C.c_bool),
        ("suppress_blank", C.c_bool), ("suppress_nst", C.c_bool),
        ("temperature", C.c_float), ("max_initial_ts", C.c_float), ("length_penalty", C.c_float),
        ("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float),
        ("no_speech_thold", C.c_float),
        ("greedy", _greedy),
        ("beam_search", _beam),
        ("new_segment_callback", NEW_SEGMENT_CB), ("new_segment_callback_user_data", C.c_void_p),
        ("progress_callback", PROGRESS_CB), ("progress_callback_user_data", C.c_void_p),
        ("encoder_begin_callback", ENCODER_BEGIN_CB), ("encoder_begin_callback_user_data", C.c_void_p),
        ("abort_callback", ABORT_CB), ("abort_callback_user_data", C.c_void_p),
        ("logits_filter_callback", LOGITS_FILTER_CB), ("logits_filter_callback_user_data", C.c_void_p),
        ("grammar_rules", C.c_void_p), ("n_grammar_rules", C.c_size_t), ("i_start_rule", C.c_size_t),
        ("grammar_penalty", C.c_float),
        ("vad", C.c_bool), ("vad_model_path", C.c_char_p),
        ("vad_params", whisper_vad_params),
    ]


# ---------------------------------------------------------------------------------------------------
# library loading
# ---------------------------------------------------------------------------------------------------
_LIBS: dict[str, C.CDLL] = {}


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen a whisper C-ABI library and declare the prototypes used by this mirror.

    Fails loudly (OSError) when the library is missing: there is no fallback implementation.
    """
    path = os.path.abspath(path or PRODUCT_LIB)
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise OSError(f"{path} not found - build it first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    P, I, F = C.c_void_p, C.c_int, C.c_float

    def proto(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    proto("whisper_context_default_params", whisper_context_params)
    proto("whisper_full_default_params", whisper_full_params, I)
    proto("whisper_init_from_file_with_params_no_state", P, C.c_char_p, whisper_context_params)
    proto("whisper_init_from_buffer_with_params_no_state", P, P, C.c_size_t, whisper_context_params)
    proto("whisper_init_state", P, P)
    proto("whisper_free", None, P)
    proto("whisper_free_state", None, P)
    proto("whisper_pcm_to_mel_with_state", I, P, P, C.POINTER(F), I, I)
    proto("whisper_set_mel_with_state", I, P, P, C.POINTER(F), I, I)
    proto("whisper_encode_with_state", I, P, P, I, I)
    proto("whisper_decode_with_state", I, P, P, C.POINTER(C.c_int32), I, I, I)
    proto("whisper_lang_auto_detect_with_state", I, P, P, I, I, C.POINTER(F))
    proto("whisper_get_logits_from_state", C.POINTER(F), P)
    proto("whisper_n_len_from_state", I, P)
    proto("whisper_full_with_state", I, P, P, whisper_full_params, P, I)
    proto("whisper_full_n_segments_from_state", I, P)
    proto("whisper_full_lang_id_from_state", I, P)
    proto("whisper_full_get_segment_t0_from_state", C.c_int64, P, I)
    proto("whisper_full_get_segment_t1_from_state", C.c_int64, P, I)
    proto("whisper_full_get_segment_text_from_state", C.c_char_p, P, I)
    proto("whisper_full_get_segment_speaker_turn_next_from_state", C.c_bool, P, I)
    proto("whisper_full_get_segment_no_speech_prob_from_state", F, P, I)
    proto("whisper_full_n_tokens_from_state", I, P, I)
    proto("whisper_full_get_token_text_from_state", C.c_char_p, P, P, I, I)
    proto("whisper_full_get_token_id_from_state", C.c_int32, P, I, I)
    proto("whisper_full_get_token_data_from_state", whisper_token_data, P, I, I)
    proto("whisper_full_get_token_p_from_state", F, P, I, I)
    proto("whisper_tokenize", I, P, C.c_char_p, C.POINTER(C.c_int32), I)
    for n in ("whisper_n_vocab", "whisper_n_text_ctx", "whisper_n_audio_ctx", "whisper_is_multilingual",
              "whisper_model_n_vocab", "whisper_model_n_audio_ctx", "whisper_model_n_audio_state",
              "whisper_model_n_audio_head", "whisper_model_n_audio_layer", "whisper_model_n_text_ctx",
              "whisper_model_n_text_state", "whisper_model_n_text_head", "whisper_model_n_text_layer",
              "whisper_model_n_mels", "whisper_model_ftype", "whisper_model_type",
              "whisper_token_eot", "whisper_token_sot", "whisper_token_solm", "whisper_token_prev",
              "whisper_token_nosp", "whisper_token_not", "whisper_token_beg", "whisper_token_translate",
              "whisper_token_transcribe"):
        proto(n, I, P)
    proto("whisper_token_lang", I, P, I)
    proto("whisper_token_to_str", C.c_char_p, P, I)
    proto("whisper_model_type_readable", C.c_char_p, P)
    proto("whisper_lang_max_id", I)
    proto("whisper_lang_id", I, C.c_char_p)
    proto("whisper_lang_str", C.c_char_p, I)
    proto("whisper_lang_str_full", C.c_char_p, I)
    proto("whisper_print_system_info", C.c_char_p)
    proto("whisper_print_timings", None, P)
    proto("whisper_reset_timings", None, P)
    proto("whisper_log_set", None, LOG_CB, P)
    if hasattr(lib, "whisper_amd_full_batch"):
        proto("whisper_amd_full_batch", I, P, C.POINTER(C.c_void_p), I, whisper_full_params, C.POINTER(C.c_void_p), C.POINTER(C.c_int))
    if hasattr(lib, "whisper_amd_rows_stats"):
        proto("whisper_amd_rows_stats", None, P, C.POINTER(C.c_long))
        proto("whisper_amd_rows_enabled", I, P)
        proto("whisper_amd_batch_one_launch", C.c_long, P)
    _LIBS[path] = lib
    return lib


_KEEP_LOG_CB = {}


def set_log_callback(lib: C.CDLL, fn=None):
    """Mirror of whisper-rs `install_logging_hooks` (src/lib.rs:67-70): route library logs to `fn(level, text)`.
    With fn=None logs are silenced."""
    cb = LOG_CB((lambda lvl, txt, ud: None) if fn is None else (lambda lvl, txt, ud: fn(lvl, (txt or b"").decode(errors="replace"))))
    _KEEP_LOG_CB[id(lib)] = cb
    lib.whisper_log_set(cb, None)


# ---------------------------------------------------------------------------------------------------
# whisper-rs mirror classes
# ---------------------------------------------------------------------------------------------------
class WhisperContextParameters:
    """src/whisper_ctx.rs:472-657 (`WhisperContextParameters`)."""

    def __init__(self, lib: C.CDLL, use_gpu: bool = True, flash_attn: bool = False, gpu_device: int = 0,
                 dtw_preset: int = AHEADS_NONE, dtw_n_top: int = -1, dtw_heads: Sequence[tuple] = ()):
        p = lib.whisper_context_default_params()
        p.use_gpu = use_gpu
        p.flash_attn = flash_attn
        p.gpu_device = gpu_device
        if dtw_preset != AHEADS_NONE:
            p.dtw_token_timestamps = True
            p.dtw_aheads_preset = dtw_preset
            p.dtw_n_top = dtw_n_top
            if dtw_preset == AHEADS_CUSTOM:
                self._heads = (whisper_ahead * len(dtw_heads))(*[whisper_ahead(a, b) for a, b in dtw_heads])
                p.dtw_aheads.n_heads = len(dtw_heads)
                p.dtw_aheads.heads = self._heads
        self.c = p


class FullParams:
    """src/whisper_params.rs:36-803 (`FullParams`): defaults from `whisper_full_default_params`, then setters."""

    def __init__(self, lib: C.CDLL, strategy: int = WHISPER_SAMPLING_GREEDY, **kw):
        self.c = lib.whisper_full_default_params(strategy)
        self._keep = {}
        # whisper-rs turns the C library's stdout printing off by default in its examples
        self.c.print_progress = False
        for k, v in kw.items():
            self.set(k, v)

    def set(self, name: str, value):
        if name == "best_of":
            self.c.greedy.best_of = value
        elif name == "beam_size":
            self.c.beam_search.beam_size = value
        elif name in ("language", "initial_prompt", "suppress_regex"):
            b = None if value is None else value.encode()
            self._keep[name] = b  # whisper-rs leaks the CString; we keep a reference
            setattr(self.c, name, b)
        elif name == "prompt_tokens":
            arr = (C.c_int32 * len(value))(*value)
            self._keep[name] = arr
            self.c.prompt_tokens = arr
            self.c.prompt_n_tokens = len(value)
        elif name.endswith("_callback"):
            ftype = dict(new_segment_callback=NEW_SEGMENT_CB, progress_callback=PROGRESS_CB,
                         encoder_begin_callback=ENCODER_BEGIN_CB, abort_callback=ABORT_CB,
                         logits_filter_callback=LOGITS_FILTER_CB)[name]
            cb = ftype(value)
            self._keep[name] = cb
            setattr(self.c, name, cb)
        else:
            setattr(self.c, name, value)
        return self


class WhisperContext:
    def __init__(self, lib: C.CDLL, ptr: int):
        self.lib, self.ptr = lib, ptr

    @classmethod
    def new_with_params(cls, path: str, params: Optional[WhisperContextParameters] = None,
                        lib: Optional[C.CDLL] = None) -> "WhisperContext":
        lib = lib or load_library()
        params = params or WhisperContextParameters(lib)
        ptr = lib.whisper_init_from_file_with_params_no_state(path.encode(), params.c)
        if not ptr:
            raise WhisperError("InitError")  # src/whisper_ctx.rs:38-42
        ctx = cls(lib, ptr)
        ctx._params = params
        return ctx

    @classmethod
    def new_from_buffer_with_params(cls, buf: bytes, params: Optional[WhisperContextParameters] = None,
                                    lib: Optional[C.CDLL] = None) -> "WhisperContext":
        lib = lib or load_library()
        params = params or WhisperContextParameters(lib)
        if isinstance(buf, np.ndarray):      # parsed in place (the library copies nothing beyond the call, whisper.cpp:3684-3719)
            arr = np.ascontiguousarray(buf, dtype=np.uint8)
            ptr = lib.whisper_init_from_buffer_with_params_no_state(C.c_void_p(arr.ctypes.data), arr.size, params.c)
        else:
            cbuf = C.create_string_buffer(buf, len(buf))
            ptr = lib.whisper_init_from_buffer_with_params_no_state(C.cast(cbuf, C.c_void_p), len(buf), params.c)
        if not ptr:
            raise WhisperError("InitError")
        ctx = cls(lib, ptr)
        ctx._params = params
        return ctx

    def create_state(self) -> "WhisperState":
        ptr = self.lib.whisper_init_state(self.ptr)
        if not ptr:
            raise WhisperError("InitError")
        return WhisperState(self, ptr)

    def __getattr__(self, name):
        # n_vocab(), model_n_text_layer(), token_eot(), ... -> whisper_<name>(ctx)
        if name.startswith("_"):
            raise AttributeError(name)
        fn = getattr(self.lib, "whisper_" + name)
        return lambda *a: fn(self.ptr, *a)

    def token_to_bytes(self, tok: int) -> bytes:
        return self.lib.whisper_token_to_str(self.ptr, tok)

    def tokenize(self, text: str, max_tokens: int = 1024) -> list[int]:
        arr = (C.c_int32 * max_tokens)()
        n = self.lib.whisper_tokenize(self.ptr, text.encode(), arr, max_tokens)
        if n < 0:
            raise WhisperError("InvalidText", n)
        return list(arr[:n])

    def free(self):
        if self.ptr:
            self.lib.whisper_free(self.ptr)
            self.ptr = None


def full_batch(ctx: "WhisperContext", states: Sequence["WhisperState"], params: "FullParams", pcms: Sequence) -> int:
    """Extension: transcribe several independent chunks concurrently on one device (whisper_amd_full_batch).
    `pcms[i]` is a numpy f32 array or a (device_ptr, n) tuple."""
    n = len(states)
    keep, ptrs, lens = [], (C.c_void_p * n)(), (C.c_int * n)()
    for i, p in enumerate(pcms):
        if isinstance(p, tuple):
            ptrs[i], lens[i] = p[0], p[1]
        else:
            a = np.ascontiguousarray(p, dtype=np.float32); keep.append(a)
            ptrs[i], lens[i] = a.ctypes.data, len(a)
    sp = (C.c_void_p * n)(*[s.ptr for s in states])
    r = ctx.lib.whisper_amd_full_batch(ctx.ptr, sp, n, params.c, ptrs, lens)
    if r != 0:
        raise WhisperError("GenericError(%d)" % r, r)
    return r


class WhisperState:
    def __init__(self, ctx: WhisperContext, ptr: int):
        self.ctx, self.lib, self.ptr = ctx, ctx.lib, ptr

    # -- low level stage API (src/whisper_state.rs:50-260) ---------------------------------------
    def pcm_to_mel(self, pcm: np.ndarray, n_threads: int = 1):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        r = self.lib.whisper_pcm_to_mel_with_state(self.ctx.ptr, self.ptr, pcm.ctypes.data_as(C.POINTER(C.c_float)),
                                                   len(pcm), n_threads)
        if r != 0:
            raise WhisperError("UnableToCalculateSpectrogram", r)

    def set_mel(self, mel: np.ndarray):
        """mel: [n_mel][n_len] f32 (layout of whisper.cpp:3904-3923)."""
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        n_mel, n_len = mel.shape
        r = self.lib.whisper_set_mel_with_state(self.ctx.ptr, self.ptr, mel.ctypes.data_as(C.POINTER(C.c_float)),
                                                n_len, n_mel)
        if r != 0:
            raise WhisperError("InvalidMelBands", r)

    def encode(self, offset: int = 0, n_threads: int = 1):
        r = self.lib.whisper_encode_with_state(self.ctx.ptr, self.ptr, offset, n_threads)
        if r != 0:
            raise WhisperError("UnableToCalculateEvaluation", r)

    def decode(self, tokens: Sequence[int], n_past: int, n_threads: int = 1):
        arr = (C.c_int32 * len(tokens))(*tokens)
        r = self.lib.whisper_decode_with_state(self.ctx.ptr, self.ptr, arr, len(tokens), n_past, n_threads)
        if r != 0:
            raise WhisperError("UnableToCalculateEvaluation", r)

    def get_logits_last(self, n_tokens: int) -> np.ndarray:
        """Row n_tokens-1 of the logits of the last decode() (the only valid row, whisper.cpp:2965-2971)."""
        nv = self.ctx.n_vocab()
        p = self.lib.whisper_get_logits_from_state(self.ptr)
        return np.ctypeslib.as_array(p, shape=(n_tokens * nv,))[(n_tokens - 1) * nv:].copy()

    def lang_detect(self, offset_ms: int = 0, n_threads: int = 1):
        probs = (C.c_float * (self.lib.whisper_lang_max_id() + 1))()
        r = self.lib.whisper_lang_auto_detect_with_state(self.ctx.ptr, self.ptr, offset_ms, n_threads, probs)
        if r < 0:
            raise WhisperError("GenericError", r)
        return r, np.array(probs[:], dtype=np.float32)

    def rows_stats(self) -> tuple:
        """Extension: (decoder passes of several token rows served by the one-launch form, passes sent back to the launch sequence)."""
        out = (C.c_long * 2)()
        self.lib.whisper_amd_rows_stats(self.ptr, out)
        return int(out[0]), int(out[1])

    def n_len(self) -> int:
        return self.lib.whisper_n_len_from_state(self.ptr)

    # -- full pipeline (src/whisper_state.rs:289-321) ---------------------------------------------
    def full(self, params: FullParams, pcm) -> int:
        """`pcm`: numpy f32 array on the host, or an int device pointer + length tuple (ptr, n)."""
        if isinstance(pcm, tuple):
            ptr, n = pcm
        else:
            pcm = np.ascontiguousarray(pcm, dtype=np.float32)
            if len(pcm) == 0:
                raise WhisperError("NoSamples")  # src/whisper_state.rs:294-298
            self._pcm_keep = pcm
            ptr, n = pcm.ctypes.data, len(pcm)
        r = self.lib.whisper_full_with_state(self.ctx.ptr, self.ptr, params.c, C.c_void_p(ptr), n)
        if r != 0:
            raise WhisperError("GenericError(%d)" % r, r)  # src/whisper_state.rs:310-320
        return r

    def full_n_segments(self) -> int:
        return self.lib.whisper_full_n_segments_from_state(self.ptr)

    def full_lang_id(self) -> int:
        return self.lib.whisper_full_lang_id_from_state(self.ptr)

    def full_get_segment_t0(self, i: int) -> int:
        return self.lib.whisper_full_get_segment_t0_from_state(self.ptr, i)

    def full_get_segment_t1(self, i: int) -> int:
        return self.lib.whisper_full_get_segment_t1_from_state(self.ptr, i)

    def full_get_segment_bytes(self, i: int) -> bytes:
        return self.lib.whisper_full_get_segment_text_from_state(self.ptr, i)

    def full_get_segment_text(self, i: int) -> str:
        return self.full_get_segment_bytes(i).decode(errors="replace")

    def full_get_segment_no_speech_prob(self, i: int) -> float:
        return self.lib.whisper_full_get_segment_no_speech_prob_from_state(self.ptr, i)

    def full_n_tokens(self, i: int) -> int:
        return self.lib.whisper_full_n_tokens_from_state(self.ptr, i)

    def full_get_token_id(self, i: int, j: int) -> int:
        return self.lib.whisper_full_get_token_id_from_state(self.ptr, i, j)

    def full_get_token_data(self, i: int, j: int) -> whisper_token_data:
        return self.lib.whisper_full_get_token_data_from_state(self.ptr, i, j)

    def full_get_token_prob(self, i: int, j: int) -> float:
        return self.lib.whisper_full_get_token_p_from_state(self.ptr, i, j)

    def full_get_token_bytes(self, i: int, j: int) -> bytes:
        return self.lib.whisper_full_get_token_text_from_state(self.ctx.ptr, self.ptr, i, j)

    def segments(self) -> list[dict]:
        """Convenience: everything the examples print (examples/basic_use.rs:330-341) + token ids."""
        out = []
        for i in range(self.full_n_segments()):
            nt = self.full_n_tokens(i)
            toks = [self.full_get_token_data(i, j) for j in range(nt)]
            out.append(dict(t0=self.full_get_segment_t0(i), t1=self.full_get_segment_t1(i),
                            text=self.full_get_segment_bytes(i),
                            ids=[t.id for t in toks], tids=[t.tid for t in toks],
                            p=[t.p for t in toks], plog=[t.plog for t in toks],
                            pt=[t.pt for t in toks], ptsum=[t.ptsum for t in toks],
                            t_dtw=[t.t_dtw for t in toks],
                            tok_t0=[t.t0 for t in toks], tok_t1=[t.t1 for t in toks], vlen=[t.vlen for t in toks],
                            no_speech_prob=self.full_get_segment_no_speech_prob(i)))
        return out

    def free(self):
        if self.ptr:
            self.lib.whisper_free_state(self.ptr)
            self.ptr = None

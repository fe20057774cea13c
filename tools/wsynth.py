"""Synthetic inputs for the Whisper hot path: legacy-ggml model files and 16 kHz mono f32 audio.

No real `ggml-*.bin` or `jfk.wav` is obtainable offline, so every parity test and the
bench run on seeded synthetic models/audio written here.  The file format restated below is the one the
reference loader parses (`sys/whisper.cpp/src/whisper.cpp:1503-1974`) and the reference converter writes
(`sys/whisper.cpp/models/convert-pt-to-ggml.py:266-340`): magic, 11 hparams, mel filters, vocab, then
(n_dims, name_len, ttype, ne[], name, data) tensor records.  Tensor names: `whisper-arch.h:42-107`.

This module is shared test/bench infrastructure; it is never imported by the product library.
"""
from __future__ import annotations

import os
import struct
import numpy as np

GGML_MAGIC = 0x67676D6C

# (n_audio_state, n_audio_head, n_audio_layer, n_text_layer, n_mels, n_vocab)
SHAPES = {
    # tiny synthetic shapes for fast CPU-oracle tests (not real Whisper sizes)
    "s64":    dict(d=64,   heads=1,  enc=2,  dec=3,  n_mels=80,  n_vocab=51865),
    "s128":   dict(d=128,  heads=2,  enc=3,  dec=4,  n_mels=80,  n_vocab=51865),
    "s192":   dict(d=192,  heads=3,  enc=2,  dec=2 + 1, n_mels=80, n_vocab=51865),   # d not a multiple of 128: tile guards
    "w1280":  dict(d=1280, heads=20, enc=1,  dec=3,  n_mels=128, n_vocab=51866),      # large-v3's width with few layers: the wide-model paths
    "m1024":  dict(d=1024, heads=16, enc=1,  dec=3,  n_mels=80,  n_vocab=51865),      # medium's width and head count with few layers (config 4's products)
    # real Whisper shapes (SURVEY.md §8 header)
    "tiny":   dict(d=384,  heads=6,  enc=4,  dec=4,  n_mels=80,  n_vocab=51865),
    "base":   dict(d=512,  heads=8,  enc=6,  dec=6,  n_mels=80,  n_vocab=51865),
    "small":  dict(d=768,  heads=12, enc=12, dec=12, n_mels=80,  n_vocab=51865),
    "medium": dict(d=1024, heads=16, enc=24, dec=24, n_mels=80,  n_vocab=51865),
    "large-v3": dict(d=1280, heads=20, enc=32, dec=32, n_mels=128, n_vocab=51866),
}

N_AUDIO_CTX = 1500
N_TEXT_CTX = 448
N_FFT_BINS = 201
N_BASE_VOCAB = 50257


# --------------------------------------------------------------------------------------------------
# Slaney mel filterbank (the formula behind OpenAI's mel_filters.npz; librosa.filters.mel, norm="slaney")
# --------------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(n_mels: int = 80, n_fft: int = 400, sr: int = 16000) -> np.ndarray:
    """[n_mels, 1 + n_fft/2] float32 triangular Slaney-normalised filters."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, sr / 2, n_bins)
    mel_pts = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_pts[2:n_mels + 2] - mel_pts[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


# --------------------------------------------------------------------------------------------------
# vocabulary: 50257 unique byte strings; id 220 == " " (as in GPT-2), id 50256 == "" (multilingual)
# --------------------------------------------------------------------------------------------------
def synth_vocab() -> list[bytes]:
    words = []
    for i in range(N_BASE_VOCAB):
        if i == 220:
            words.append(b" ")
            continue
        if i == N_BASE_VOCAB - 1:
            words.append(b"")
            continue
        n, s = i, ""
        while True:
            s = chr(ord("a") + n % 26) + s
            n = n // 26 - 1
            if n < 0:
                break
        if i % 3 == 0:
            s = " " + s
        words.append(s.encode())
    assert len(set(words)) == len(words)
    return words


# --------------------------------------------------------------------------------------------------
# tensors
# --------------------------------------------------------------------------------------------------
def _tensor_list(shape: dict, rng: np.random.Generator):
    """Yield (name, ndarray) in numpy (row-major) shapes; 2-D+ matrices F16, 1-D F32."""
    d, Le, Ld, n_mels, n_vocab = shape["d"], shape["enc"], shape["dec"], shape["n_mels"], shape["n_vocab"]

    def mat(o, i, scale=1.0):
        return (rng.standard_normal((o, i), dtype=np.float32) * (scale / np.sqrt(i))).astype(np.float16)

    def vec(n, mean=0.0, sigma=0.1):
        return (mean + sigma * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)

    yield "encoder.positional_embedding", rng.standard_normal((N_AUDIO_CTX, d), dtype=np.float32)
    yield "encoder.conv1.weight", (rng.standard_normal((d, n_mels, 3), dtype=np.float32) / np.sqrt(3 * n_mels)).astype(np.float16)
    yield "encoder.conv1.bias", vec(d).reshape(d, 1)
    yield "encoder.conv2.weight", (rng.standard_normal((d, d, 3), dtype=np.float32) / np.sqrt(3 * d)).astype(np.float16)
    yield "encoder.conv2.bias", vec(d).reshape(d, 1)
    yield "encoder.ln_post.weight", vec(d, 1.0)
    yield "encoder.ln_post.bias", vec(d)
    for i in range(Le):
        p = f"encoder.blocks.{i}."
        yield p + "attn_ln.weight", vec(d, 1.0)
        yield p + "attn_ln.bias", vec(d)
        yield p + "attn.query.weight", mat(d, d)
        yield p + "attn.query.bias", vec(d)
        yield p + "attn.key.weight", mat(d, d)
        yield p + "attn.value.weight", mat(d, d)
        yield p + "attn.value.bias", vec(d)
        yield p + "attn.out.weight", mat(d, d)
        yield p + "attn.out.bias", vec(d)
        yield p + "mlp_ln.weight", vec(d, 1.0)
        yield p + "mlp_ln.bias", vec(d)
        yield p + "mlp.0.weight", mat(4 * d, d)
        yield p + "mlp.0.bias", vec(4 * d)
        yield p + "mlp.2.weight", mat(d, 4 * d)
        yield p + "mlp.2.bias", vec(d)
    # decoder.positional_embedding with sigma~3 keeps decoding non-degenerate
    yield "decoder.positional_embedding", 3.0 * rng.standard_normal((N_TEXT_CTX, d), dtype=np.float32)
    yield "decoder.token_embedding.weight", rng.standard_normal((n_vocab, d), dtype=np.float32).astype(np.float16)
    yield "decoder.ln.weight", vec(d, 1.0)
    yield "decoder.ln.bias", vec(d)
    for i in range(Ld):
        p = f"decoder.blocks.{i}."
        for a in ("attn", "cross_attn"):
            yield p + a + "_ln.weight", vec(d, 1.0)
            yield p + a + "_ln.bias", vec(d)
            yield p + a + ".query.weight", mat(d, d)
            yield p + a + ".query.bias", vec(d)
            yield p + a + ".key.weight", mat(d, d)
            yield p + a + ".value.weight", mat(d, d)
            yield p + a + ".value.bias", vec(d)
            yield p + a + ".out.weight", mat(d, d)
            yield p + a + ".out.bias", vec(d)
        yield p + "mlp_ln.weight", vec(d, 1.0)
        yield p + "mlp_ln.bias", vec(d)
        yield p + "mlp.0.weight", mat(4 * d, d)
        yield p + "mlp.0.bias", vec(4 * d)
        yield p + "mlp.2.weight", mat(d, 4 * d)
        yield p + "mlp.2.bias", vec(d)


def write_model(path: str, shape_name: str = "s64", seed: int = 0, with_tensors: bool = True) -> str:
    """Write a synthetic legacy-ggml Whisper model (ftype=1: F16 matrices, F32 vectors)."""
    shape = SHAPES[shape_name]
    rng = np.random.default_rng(seed)
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(struct.pack("<I", GGML_MAGIC))
        f.write(struct.pack("<11i", shape["n_vocab"], N_AUDIO_CTX, shape["d"], shape["heads"], shape["enc"],
                            N_TEXT_CTX, shape["d"], shape["heads"], shape["dec"], shape["n_mels"], 1))
        filt = mel_filterbank(shape["n_mels"])
        f.write(struct.pack("<2i", shape["n_mels"], N_FFT_BINS))
        f.write(filt.tobytes())
        vocab = synth_vocab()
        f.write(struct.pack("<i", len(vocab)))
        for w in vocab:
            f.write(struct.pack("<I", len(w)))
            f.write(w)
        if with_tensors:
            for name, arr in _tensor_list(shape, rng):
                ttype = 1 if arr.dtype == np.float16 else 0
                nb = name.encode()
                f.write(struct.pack("<3i", arr.ndim, len(nb), ttype))
                for dim in reversed(arr.shape):  # ggml ne[] is fastest-varying first
                    f.write(struct.pack("<i", dim))
                f.write(nb)
                f.write(np.ascontiguousarray(arr).tobytes())
    os.replace(tmp, path)
    return path


def model_path(shape_name: str, seed: int = 0, cache_dir: str | None = None) -> str:
    """Path of a cached synthetic model, written on first use (models are too big to commit)."""
    cache_dir = cache_dir or os.environ.get("WHISPER_AMD_CACHE", "/tmp/whisper_amd_cache")
    os.makedirs(cache_dir, exist_ok=True)
    p = os.path.join(cache_dir, f"synth-{shape_name}-seed{seed}.bin")
    if not os.path.exists(p):
        write_model(p, shape_name, seed)
    return p


def quant_model_path(shape_name: str, qtype: str, seed: int = 0, cache_dir: str | None = None) -> str:
    """Path of the cached Q5_0 / Q8_0 version of a synthetic model, produced on first use by the REFERENCE's own quantizer
    (examples/quantize, compiled by oracle/Makefile into oracle/_ref/quantize-ref; test infrastructure only)."""
    import subprocess
    src = model_path(shape_name, seed, cache_dir)
    dst = src[:-4] + "-" + qtype + ".bin"
    if not os.path.exists(dst):
        tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "quantize-ref")
        if not os.path.exists(tool):
            raise FileNotFoundError("oracle/_ref/quantize-ref missing (make -C oracle ref)")
        tmp = dst + ".tmp%d" % os.getpid()
        subprocess.run([tool, src, tmp, qtype], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.replace(tmp, dst)
    return dst


# --------------------------------------------------------------------------------------------------
# audio
# --------------------------------------------------------------------------------------------------
def synth_audio(n_samples: int = 480000, seed: int = 0) -> np.ndarray:
    """Seeded 16 kHz mono f32 test signal: two tones/chirp with a slow envelope + gaussian noise."""
    rng = np.random.default_rng(1000 + seed)
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    f0 = 180.0 + 40.0 * (seed % 7)
    x = 0.3 * np.sin(2 * np.pi * f0 * t) + 0.2 * np.sin(2 * np.pi * (500.0 + 100.0 * t) * t)
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * 0.7 * t + seed)
    x += 0.05 * rng.standard_normal(n_samples)
    return x.astype(np.float32)


if __name__ == "__main__":
    import sys
    name = sys.argv[1] if len(sys.argv) > 1 else "s64"
    out = sys.argv[2] if len(sys.argv) > 2 else f"/tmp/synth-{name}.bin"
    print(write_model(out, name), os.path.getsize(out))

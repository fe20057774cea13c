#!/usr/bin/env python3
"""bench.py - headline benchmark of the Whisper hot path on MI355X (contract: see the task brief).

Metric (BASELINE.json): real-time factor = audio-seconds / wall-seconds of `WhisperState::full`
(`whisper_full_with_state`) on 30 s, 16 kHz mono f32 chunks, ggml-small-shaped model, greedy decoding.
A "step" = one full() over one batch of synthetic chunks (default one 30 s chunk per GPU), PCM already
resident in HBM when the timed region starts.  N > 1: one process per GPU, independent chunks per rank
(no collective in the data path; weak scaling), barrier + max over ranks around the timed region.

The headline `value` is measured with flash_attn = false: the reference-order path whose results are bit-identical to the reference
engine (the parity tests' bar).  The other path (flash_attn = true: F16 MFMA encoder, tested to |d logit| <= 1e-3 max|logit|) is
timed in the same run and reported beside it as `flash_attn_path`, never as `value`.

Extra objects on the JSON line:
  roofline     - the decode step (dominant by time): HBM-bound; achieved = algorithmic bytes per decode
                 step (weights + cross K/V + self K/V) / device time per step measured with
                 HIP events on the state's stream (whisper_amd_decode_step_probe).
  encoder      - encoder ms per 30 s chunk + fraction of the dense F16 MFMA peak (second half of the metric).
  cpu_baseline - the reference engine itself (oracle/_ref, kind "reference") timed on this box's host
                 cores on the same chunk, same parameters (bounded: one chunk, token count capped).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "whisper-rust_amd"))

import numpy as np  # noqa: E402
import wsynth  # noqa: E402
import whisper_rs as W  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3   # f32-input MFMA = the F32 vector rate (MI355X_MICROARCH.md, Matrix cores)


def encoder_flops(shape):
    d, Le, Ld, n_mels, T = shape["d"], shape["enc"], shape["dec"], shape["n_mels"], 1500
    return 2 * d * n_mels * 3 * 3000 + 2 * d * d * 3 * 1500 + Le * (8 * T * d * d + 4 * T * T * d + 16 * T * d * d) + Ld * 4 * T * d * d


def decode_bytes(shape, n_past):
    d, Ld, nv, T = shape["d"], shape["dec"], shape["n_vocab"], 1500
    return 2 * (14 * Ld * d * d + nv * d) + 4 * Ld * T * d + 4 * Ld * d * n_past


class Hip:
    """The few HIP runtime calls the bench needs (device buffers for resident PCM, sync)."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]

    def check(self, r, what):
        if r != 0:
            raise RuntimeError("%s failed with hipError %d" % (what, r))

    def set_device(self, i):
        self.check(self.lib.hipSetDevice(i), "hipSetDevice")

    def to_device(self, arr):
        p = C.c_void_p()
        self.check(self.lib.hipMalloc(C.byref(p), arr.nbytes), "hipMalloc")
        self.check(self.lib.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1), "hipMemcpy")
        return p.value

    def sync(self):
        self.check(self.lib.hipDeviceSynchronize(), "hipDeviceSynchronize")


class StubEngine:
    """Stands in for the HIP library in the CPU rehearsal of the N > 1 control flow (tests/test_bench_dist.py, --stub): the model image is
    only check-summed, a step sleeps a few ms per chunk and "decodes" 7 tokens per chunk.  Everything around it - rank / chunk partition,
    the weight broadcast, barriers, the MAX / SUM reductions, the config-3 object - is bench.py's real code."""

    def __init__(self, image, rank):
        self.sum = int(np.asarray(image, dtype=np.uint8).astype(np.uint64).sum())
        self.rank = rank

    def step(self, chunk_ids):
        time.sleep(0.002 * len(chunk_ids) * (1 + self.rank))
        return 7 * len(chunk_ids)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="small", choices=list(wsynth.SHAPES))
    ap.add_argument("--chunks-per-gpu", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the 8-concurrent-chunks extra (threads upset rocprofv3)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 4 (medium, beam 5 + DTW) and 5 (large-v3 Q5_0) - they write 1.5 + 3.1 GB of synthetic models")
    ap.add_argument("--flash-attn", type=int, default=0, help="0 = reference-order path, bit-identical to whisper.cpp CPU (default: what whisper-rs gets, "
                    "src/whisper_ctx.rs:490, and the path that meets north_star's parity bar); 1 = F16-MFMA tolerance path as the headline")
    ap.add_argument("--no-second-path", action="store_true", help="skip timing the other flash_attn setting in the same run")
    ap.add_argument("--json-out", default=None, help="also write the JSON line to this file (profiler runs mix their log into stdout)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path on a 1-GPU box / on CPU)")
    ap.add_argument("--stub", action="store_true", help="CPU rehearsal of the N > 1 control flow: no HIP library, a stub step (tests/test_bench_dist.py)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        n_dev = 0 if args.stub else torch.cuda.device_count()
        local_dev = local_rank % max(1, n_dev)          # rehearsal: several ranks may share one card
        if world > max(1, n_dev) and not args.stub:
            # ranks sharing a card: the one-launch decode steps want every CU for themselves (their workgroups wait for each other, two of
            # them interleaved could starve) - the rehearsal runs the launch sequence; one rank per GPU (the real run) is unaffected
            os.environ["WHISPER_AMD_NO_MEGA"] = "1"
        if not args.stub:
            torch.cuda.set_device(local_dev)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        local_dev = local_rank
    cdev = "cuda" if (dist is not None and args.dist_backend == "nccl") else "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        return int(t.item())

    hip = lib = None
    if not args.stub:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # one hardware queue per concurrent chunk stream
        hip = Hip()
        hip.set_device(local_dev)
        lib = W.load_library()          # fails loudly if the HIP library is missing
        W.set_log_callback(lib, lambda lvl, txt: sys.stderr.write(txt) if lvl >= 3 else None)
    shape = wsynth.SHAPES[args.model]

    # ---- weights: rank 0 owns the file; the other ranks receive its image over RCCL / xGMI straight into the buffer the loader parses
    #      (no per-rank disk read, no intermediate copies: the library copies nothing beyond the init call, whisper.cpp:3684-3719)
    mp = None
    image = None
    if rank == 0:
        mp = wsynth.model_path("s64" if args.stub else args.model)
    if world > 1:
        n = torch.tensor([os.path.getsize(mp) if rank == 0 else 0], dtype=torch.int64, device=cdev)
        dist.broadcast(n, 0)
        image = np.fromfile(mp, dtype=np.uint8) if rank == 0 else np.empty(int(n.item()), dtype=np.uint8)
        host_t = torch.from_numpy(image)            # shares the numpy buffer
        if cdev == "cuda":
            dev_t = host_t.to("cuda") if rank == 0 else torch.empty(int(n.item()), dtype=torch.uint8, device="cuda")
            dist.broadcast(dev_t, 0)
            if rank != 0:
                host_t.copy_(dev_t)                 # one D2H copy into the buffer init_from_buffer reads
            del dev_t
        else:
            dist.broadcast(host_t, 0)               # in place
    n_chunks = args.chunks_per_gpu
    # chunk ids of this rank: the block partition of [0, world * n_chunks) (chunk_dp.shard_chunks) - the id seeds the chunk's audio
    import chunk_dp
    my_ids = list(chunk_dp.shard_chunks(world * n_chunks, rank, world))
    if args.stub:
        eng = StubEngine(image if image is not None else np.fromfile(mp, dtype=np.uint8), rank)
        ctx = states = pcm_host = pcm_dev = fp = None

        def step():
            return eng.step(my_ids)
    else:
        cparams = W.WhisperContextParameters(lib, gpu_device=local_dev, flash_attn=bool(args.flash_attn))
        if world > 1 and rank != 0:
            ctx = W.WhisperContext.new_from_buffer_with_params(image, cparams, lib=lib)
        else:
            ctx = W.WhisperContext.new_with_params(mp, cparams, lib=lib)
        image = None
        states = [ctx.create_state() for _ in my_ids]
        pcm_host = [wsynth.synth_audio(480000, i) for i in my_ids]
        pcm_dev = [hip.to_device(p) for p in pcm_host]
        fp = W.FullParams(lib, best_of=1, temperature_inc=0.0, language="en", no_context=True)

        def step():
            if n_chunks == 1:
                states[0].full(fp, (pcm_dev[0], 480000))
            else:       # independent chunks of this rank: one host thread each, their single-token steps decoded in lock step
                W.full_batch(ctx, states, fp, [(dp, 480000) for dp in pcm_dev])
            return sum(st.full_n_tokens(i) for st in states for i in range(st.full_n_segments()))

    def sync():
        if hip is not None:
            hip.sync()

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    t0 = time.perf_counter()
    ntok = 0
    for _ in range(args.steps):
        ntok = step()
    sync()
    barrier()
    dt = reduce_max(time.perf_counter() - t0)
    ntok_all = reduce_sum(ntok)

    audio_s = 30.0 * n_chunks * world * args.steps
    rtf = audio_s / dt
    out = {
        "metric": "real-time factor (audio-sec/wall-sec), greedy, 30 s synthetic chunks",
        "value": round(rtf, 2), "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16 operands / f32 accumulate", "data": "synthetic",
        "config": {"workload": "ggml-%s-shaped synthetic F16 model, greedy best_of=1 temperature_inc=0, %d x 30 s 16 kHz f32 chunk per GPU, PCM resident in HBM"
                               % (args.model, n_chunks),
                   "tokens_decoded_per_step": ntok_all, "flash_attn": bool(args.flash_attn), "parallelism": "chunk-dp%d" % world},
        "tokens_per_s": round(ntok_all * args.steps / dt, 1),
    }

    # ---- BASELINE config 3's shape on every N: 8 chunks per GPU (64 over 8 GPUs), each rank's chunks through its lock-step batcher;
    #      same protocol (barrier, MAX over ranks); reported beside the headline, which stays one chunk per GPU unless --chunks-per-gpu says so
    if (world > 1 or args.stub) and n_chunks != 8:
        c3_ids = list(chunk_dp.shard_chunks(world * 8, rank, world))
        if args.stub:
            def c3_step():
                return eng.step(c3_ids)
        else:
            c3_states = [ctx.create_state() for _ in c3_ids]
            c3_pcm = [hip.to_device(wsynth.synth_audio(480000, 1000 + i)) for i in c3_ids]

            def c3_step():
                W.full_batch(ctx, c3_states, fp, [(dp, 480000) for dp in c3_pcm])
                return sum(st.full_n_tokens(i) for st in c3_states for i in range(st.full_n_segments()))
        c3_step(); sync(); barrier()
        t1 = time.perf_counter()
        c3_tok = c3_step()
        sync(); barrier()
        c3_dt = reduce_max(time.perf_counter() - t1)
        c3_tok = reduce_sum(c3_tok)
        out["config3"] = {"workload": "8 x 30 s chunks per GPU in one whisper_amd_full_batch call per rank (BASELINE config 3: 64 chunks over 8 GPUs)",
                          "chunks": 8 * world, "value": round(30.0 * 8 * world / c3_dt, 1), "unit": "x real-time (aggregate over all GPUs)",
                          "ms": round(1e3 * c3_dt, 1), "tokens": c3_tok, "tokens_per_s": round(c3_tok / c3_dt, 1), "scaling": "weak"}
        if not args.stub:
            for st_ in c3_states:
                st_.free()

    if args.stub:
        if rank == 0:
            out["stub"] = {"image_checksum": eng.sum, "chunk_ids_rank0": my_ids}
            print(json.dumps(out), flush=True)
        if dist is not None:
            ok = reduce_sum(1 if eng.sum == reduce_max(float(eng.sum)) else 0)      # every rank parsed the image rank 0 sent
            if rank == 0 and ok != world:
                sys.exit("weight broadcast: %d of %d ranks hold rank 0's image" % (ok, world))
            dist.barrier()
            dist.destroy_process_group()
        return

    if rank == 0 and world == 1:
        # ---- the same step with the PCM handed over in HOST memory (what whisper-rs does): + 1.92 MB H2D per chunk
        t1 = time.perf_counter()
        for _ in range(args.steps):
            states[0].full(fp, pcm_host[0])
        hip.sync()
        out["value_incl_h2d"] = round(30.0 * args.steps / (time.perf_counter() - t1), 2)
        # ---- the other summation-order path in the same run (never the headline unless asked for with --flash-attn)
        if not args.no_second_path:
            octx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(lib, gpu_device=local_dev, flash_attn=not bool(args.flash_attn)), lib=lib)
            ost = octx.create_state()
            ost.full(fp, (pcm_dev[0], 480000)); hip.sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ost.full(fp, (pcm_dev[0], 480000))
            hip.sync()
            odt = time.perf_counter() - t1
            out["flash_attn_path" if not args.flash_attn else "reference_order_path"] = {
                "flash_attn": not bool(args.flash_attn), "value": round(30.0 * args.steps / odt, 2), "unit": "x real-time", "ms_per_step": round(1e3 * odt / args.steps, 3),
                "parity": ("tested tolerance: |d logit| <= 1e-3 max|logit| (~5e-2 absolute on these models), greedy ids equal up to a near tie "
                           "(tests/test_parity_gpu.py::test_flash_path_within_tolerance) - NOT north_star's bit-identical bar") if not args.flash_attn else
                          "bit-identical to the reference engine (tests/test_parity_gpu.py)"}
            ost.free(); octx.free()

    if rank == 0:
        # ---- stage timings of the last step (per-state counters) + roofline probes
        tm = (C.c_int64 * 12)()
        lib.whisper_amd_get_timings_us.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.whisper_amd_reset_timings.argtypes = [C.c_void_p]
        st = states[0]
        lib.whisper_amd_reset_timings(st.ptr)
        st.full(fp, (pcm_dev[0], 480000))
        lib.whisper_amd_get_timings_us(st.ptr, tm)
        t_sample, t_encode, t_decode, t_batchd, t_prompt, t_mel, n_sample, n_encode, n_decode = [int(x) for x in tm[:9]]
        enc_ms = 1e-3 * t_encode / max(1, n_encode)
        dec_ms_wall = 1e-3 * t_decode / max(1, n_decode)
        ms = C.c_float()
        lib.whisper_amd_decode_step_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        n_past = 64
        rc = lib.whisper_amd_decode_step_probe(ctx.ptr, st.ptr, n_past, 200, C.byref(ms))
        dbytes = decode_bytes(shape, n_past)
        lib.whisper_amd_mega_enabled.argtypes = [C.c_void_p]
        if rc == 0 and ms.value > 0:
            gbs = dbytes / (ms.value * 1e-3) / 1e9
            # HBM bytes per launch from the PMC counters (collected off-line with tools/profile_gpu.sh, one counter per pass, summary
            # committed under profiles/): 2 x FETCH_SIZE (gfx950 tallies 128-byte reads at 64 bytes) + WRITE_SIZE, in KiB
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "r03_decode_step_pmc.json")
            if args.model == "small" and os.path.exists(pmc) and lib.whisper_amd_mega_enabled(st.ptr):
                try:
                    pj = json.load(open(pmc))
                    traffic = int((2 * pj["FETCH_SIZE_KB_per_launch"] + pj["WRITE_SIZE_KB_per_launch"]) * 1024)
                    traffic_src = "profiles/r03_decode_step_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x 2) - collected off-line, not in this run"
                except (KeyError, ValueError):
                    traffic, traffic_src = None, None
            kname = ("k_decode_mega: the whole single-token decoder pass as ONE persistent launch (256 workgroups, granule hand-offs), n_past=64"
                     if lib.whisper_amd_mega_enabled(st.ptr) else
                     "decode step = hipGraph of 122 launches (k_gemv_exact weight streaming + k_attn_exact), 1 token, n_past=64")
            out["roofline"] = {"bound": "hbm", "kernel": kname,
                               "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "traffic_source": traffic_src, "bytes_per_step": dbytes, "ms_per_step_device": round(ms.value, 4),
                               "ms_per_token_wall_in_full": round(dec_ms_wall, 4)}
        # ---- throughput mode on the same GPU: 8 independent chunks transcribed together (one host thread each, single-token steps in lock step)
        if world == 1 and not args.no_concurrent:
            try:
                NB = 8      # BASELINE config 3: 64 chunks over 8 GPUs = 8 per GPU
                tst = [ctx.create_state() for _ in range(NB)]
                tp = [hip.to_device(wsynth.synth_audio(480000, 100 + i)) for i in range(NB)]
                W.full_batch(ctx, tst, fp, [(p_, 480000) for p_ in tp])
                hip.sync(); t1 = time.perf_counter()
                W.full_batch(ctx, tst, fp, [(p_, 480000) for p_ in tp])
                hip.sync(); tdt = time.perf_counter() - t1
                bs_, br_ = C.c_long(), C.c_long()
                lib.whisper_amd_batch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
                lib.whisper_amd_batch_stats(ctx.ptr, bs_, br_)
                ntk = sum(s_.full_n_tokens(i) for s_ in tst for i in range(s_.full_n_segments()))
                rows_pp = br_.value / max(1, bs_.value)
                out["concurrent_chunks"] = {"chunks": NB, "value": round(30.0 * NB / tdt, 1), "unit": "x real-time (aggregate)", "ms": round(1e3 * tdt, 1),
                                            "tokens": ntk, "lockstep_passes": bs_.value, "rows_per_pass": round(rows_pp, 2),
                                            "vs_single_chunk": round(30.0 * NB / tdt / rtf, 2),
                                            "bytes_per_pass": int(2 * (14 * shape["dec"] * shape["d"] ** 2 + shape["n_vocab"] * shape["d"]) + rows_pp * (4 * shape["dec"] * 1500 * shape["d"] + 4 * shape["dec"] * shape["d"] * 110)),
                                            "note": "lock-step batched decode: one decoder pass reads every weight row once for all chunks' tokens (W + B (KVx + KVs) bytes per pass)"}
                out["concurrent_chunks"]["tokens_per_s"] = round(ntk / tdt, 1)
                out["concurrent_chunks"]["passes_one_launch"] = int(lib.whisper_amd_batch_one_launch(ctx.ptr))
                import bench_configs
                # the 8-row pass itself (HIP events around back-to-back launches, rows of 8 different chunks) against W + 8 (KVx + KVs)
                out["concurrent_chunks"]["roofline"] = bench_configs.rows_roofline(lib, ctx, tst, shape, NB, 110, False)
                # ... and the 5-row pass of a beam-search / best_of step (rows of ONE chunk)
                out["beam5_step"] = bench_configs.rows_roofline(lib, ctx, tst, shape, 5, 64, True)
                for s_ in tst:
                    s_.free()
            except Exception as ex:  # extension only; never fail the headline
                out["concurrent_chunks"] = {"error": str(ex)}
        eflops = encoder_flops(shape)
        # the pipe the encoder's products run on: reference order = F32 MFMA (exact fmaf chains, 157.3 TFLOP/s = the F32 vector peak);
        # flash_attn = F16 MFMA (2.5 PFLOP/s dense)
        epeak = MFMA_F16_PEAK_TFLOPS if args.flash_attn else MFMA_F32_PEAK_TFLOPS
        out["encoder"] = {"ms_per_30s_chunk": round(enc_ms, 3), "gflop": round(eflops / 1e9, 1),
                          "achieved_tflops": round(eflops / (enc_ms * 1e-3) / 1e12, 1) if enc_ms > 0 else None,
                          "pipe": "f16 mfma (v_mfma_f32_16x16x32_f16)" if args.flash_attn else "f32 mfma (v_mfma_f32_16x16x4_f32 / 16x16x1_4b: bitwise fmaf chains)",
                          "peak_tflops": epeak,
                          "frac": round(eflops / (enc_ms * 1e-3) / 1e12 / epeak, 4) if enc_ms > 0 else None,
                          "frac_of_f16_mfma_peak": round(eflops / (enc_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4) if enc_ms > 0 else None,
                          "mel_ms": round(1e-3 * t_mel, 3)}
        # ---- the reference's whisper-bench protocol (examples/bench/bench.cpp: 256 single tokens, 64 batches of 5, 16 prompts of 256;
        #      SURVEY.md 8d) through the public C API, host work included - figures that do not depend on how many tokens a model decodes
        try:
            tok = [int(ctx.lib.whisper_token_sot(ctx.ptr))] * 256
            def _run(n_tok, n_past, reps):
                st.decode(tok[:n_tok], n_past)
                hip.sync(); t_ = time.perf_counter()
                for _ in range(reps):
                    st.decode(tok[:n_tok], n_past)
                hip.sync()
                return 1e3 * (time.perf_counter() - t_) / (reps * n_tok)
            out["whisper_bench"] = {"decode_ms_per_token": round(_run(1, 64, 64), 4), "batch5_ms_per_token": round(_run(5, 64, 32), 4),
                                    "prompt256_ms_per_token": round(_run(256, 0, 8), 4)}
        except Exception as ex:  # extension only
            out["whisper_bench"] = {"error": str(ex)}
        out["host_sampling_ms_per_token"] = round(1e-3 * t_sample / max(1, n_decode), 4)
        ov = (C.c_int * 2)()
        lib.whisper_amd_overlap_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.whisper_amd_overlap_stats(st.ptr, ov)
        out["host_overlap"] = {"predictions_confirmed": int(ov[0]), "predictions_wrong": int(ov[1])}

        # ---- CPU baseline: the reference engine on this box's host cores (rank 0, N == 1 only)
        ref_path = os.path.join(ROOT, "oracle", "_ref", "libwhisper_ref.so")
        if world == 1 and not args.no_cpu_baseline and os.path.exists(ref_path):
            ref = W.load_library(ref_path)
            W.set_log_callback(ref, None)
            cores = os.cpu_count() or 1
            rctx = W.WhisperContext.new_with_params(mp, W.WhisperContextParameters(ref, use_gpu=False), lib=ref)
            rst = rctx.create_state()
            # thread sweep on a bounded sample (encode once + 16 single-token decode steps; SURVEY.md 8d: "sweep 8/16/32/all and report
            # best"): single-token GEMVs stop scaling long before all cores are busy, so more threads is not faster
            sweep = {}
            sot = int(rctx.token_sot())
            for nt in sorted({n for n in (4, 8, 16, 32, 64) if n <= cores} | {min(cores, 64)}):
                rst.pcm_to_mel(pcm_host[0], nt)
                t1 = time.perf_counter(); rst.encode(0, nt); te = time.perf_counter() - t1
                rst.decode([sot, sot + 1, sot + 102], 0, nt)
                t1 = time.perf_counter()
                for i_ in range(16):
                    rst.decode([1000 + i_], 3 + i_, nt)
                td = (time.perf_counter() - t1) / 16
                sweep[nt] = {"encode_s": round(te, 3), "decode_ms_per_token": round(1e3 * td, 3), "est_chunk_s": round(te + 220 * td, 2)}
            nthr = min(sweep, key=lambda n: sweep[n]["est_chunk_s"])
            rst.free(); rst = rctx.create_state()
            rfp = W.FullParams(ref, best_of=1, temperature_inc=0.0, language="en", no_context=True, n_threads=nthr)
            t1 = time.perf_counter()
            rst.full(rfp, pcm_host[0])
            rdt = time.perf_counter() - t1
            # identity check on FRESH states on both sides (the reference's no-speech probe reads logits an earlier call
            # left behind, so a re-used state can legitimately answer differently): raw decoder-0 token sequence + segments
            gst = ctx.create_state()
            gst.full(fp, (pcm_dev[0], 480000))
            info = (C.c_double * 8)(); ids_r = (C.c_int32 * 512)(); ids_g = (C.c_int32 * 512)()
            ref.ref_shim_decoder_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int]
            lib.whisper_amd_decoder_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int]
            n_r = ref.ref_shim_decoder_info(rst.ptr, 0, info, ids_r, 512)
            n_g = lib.whisper_amd_decoder_info(gst.ptr, 0, info, ids_g, 512)
            rtok = n_r
            same = n_r == n_g and list(ids_r[:n_r]) == list(ids_g[:n_g]) and \
                [(s["t0"], s["t1"], s["ids"]) for s in rst.segments()] == [(s["t0"], s["t1"], s["ids"]) for s in gst.segments()]
            out["cpu_baseline"] = {"value": round(30.0 / rdt, 3), "unit": "x real-time", "cores": nthr, "kind": "reference",
                                   "sample": "one 30 s chunk (seed 0), same model file and FullParams, plain ggml-cpu AVX2 build (OpenBLAS absent), %d tokens decoded, %.2f s, "
                                             "at the best thread count of the sweep" % (rtok, rdt),
                                   "host_cores": cores, "thread_sweep": {str(k): v for k, v in sweep.items()},
                                   "token_ids_identical_to_gpu": bool(same)}
        # ---- BASELINE configs 4 and 5 (tools/bench_configs.py), each with the CPU reference beside it at the thread count of the sweep above
        if world == 1 and not args.no_configs and args.model == "small":
            import bench_configs
            ref_lib = None
            if not args.no_cpu_baseline and os.path.exists(ref_path):
                ref_lib = W.load_library(ref_path)
            nthr_c = out.get("cpu_baseline", {}).get("cores", 16)
            for name, fn in (("config4", bench_configs.config4), ("config5", bench_configs.config5)):
                t1 = time.perf_counter()
                try:
                    out[name] = fn(W, lib, ref_lib, hip, nthr_c, with_cpu=ref_lib is not None)
                except Exception as ex:  # extension only; never fail the headline
                    out[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                out[name]["wall_s"] = round(time.perf_counter() - t1, 1)
        print(json.dumps(out), flush=True)
        if args.json_out:
            with open(args.json_out, "w") as f:
                f.write(json.dumps(out) + "\n")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

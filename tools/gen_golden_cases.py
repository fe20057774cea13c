"""The full() parameter sets the golden vectors were generated with (shared by tools/gen_golden.py and tests/)."""
FULL_CASES = {
    "greedy_tinc0": dict(strategy=0, best_of=1, temperature_inc=0.0),
    "greedy_ladder": dict(strategy=0, best_of=2, temperature_inc=0.2),
    "beam3": dict(strategy=1, beam_size=3, best_of=2, temperature_inc=0.0),
    "greedy_no_timestamps": dict(strategy=0, best_of=1, temperature_inc=0.0, no_timestamps=True),
    "greedy_single_segment_maxtok": dict(strategy=0, best_of=1, temperature_inc=0.0, single_segment=True, max_tokens=24),
    "greedy_audio_ctx": dict(strategy=0, best_of=1, temperature_inc=0.0, audio_ctx=512),
    "greedy_offset_duration": dict(strategy=0, best_of=1, temperature_inc=0.0, offset_ms=5000, duration_ms=12000),
    "greedy_prompt_context": dict(strategy=0, best_of=1, temperature_inc=0.0, no_context=False, prompt_tokens=[500, 600, 700, 800]),
    "greedy_suppress_nst_translate": dict(strategy=0, best_of=1, temperature_inc=0.0, suppress_nst=True, translate=True, language="de"),
}

# (tag, whisper_alignment_heads_preset, WhisperContextParameters kwargs): 1 = N_TOP_MOST, 2 = CUSTOM
DTW_CASES = [
    ("ntop2", 1, dict(dtw_n_top=2)),
    ("custom", 2, dict(dtw_heads=[(1, 0), (2, 1), (2, 0)])),
]

// abi_layout.cpp - prints sizeof / offsetof of every field of the structs that cross the whisper C ABI by value.
// Compiled twice by tests/test_abi.py: once against the reference header (-DABI_HEADER='"whisper.h"' with the reference's include
// paths, where /root/reference is present) and once against include/whisper_amd.h; the two outputs must be identical, and both must
// equal the committed listing tests/golden/abi_layout.txt (what the reference header gave when the fixture was made).
// ref: sys/whisper.cpp/include/whisper.h:106-151, 192-199, 485-588; callers src/whisper_params.rs:50, src/whisper_state.rs:289-321.
#include ABI_HEADER
#include <cstddef>
#include <cstdio>

#define SZ(T)      printf("sizeof %s = %zu align %zu\n", #T, sizeof(T), alignof(T))
#define OFF(T, f)  printf("  %s.%s @ %zu size %zu\n", #T, #f, offsetof(T, f), sizeof(((T *) 0)->f))

int main() {
    SZ(whisper_ahead); OFF(whisper_ahead, n_text_layer); OFF(whisper_ahead, n_head);
    SZ(whisper_aheads); OFF(whisper_aheads, n_heads); OFF(whisper_aheads, heads);
    SZ(whisper_context_params);
    OFF(whisper_context_params, use_gpu); OFF(whisper_context_params, flash_attn); OFF(whisper_context_params, gpu_device);
    OFF(whisper_context_params, dtw_token_timestamps); OFF(whisper_context_params, dtw_aheads_preset); OFF(whisper_context_params, dtw_n_top);
    OFF(whisper_context_params, dtw_aheads); OFF(whisper_context_params, dtw_mem_size);
    SZ(whisper_token_data);
    OFF(whisper_token_data, id); OFF(whisper_token_data, tid); OFF(whisper_token_data, p); OFF(whisper_token_data, plog); OFF(whisper_token_data, pt);
    OFF(whisper_token_data, ptsum); OFF(whisper_token_data, t0); OFF(whisper_token_data, t1); OFF(whisper_token_data, t_dtw); OFF(whisper_token_data, vlen);
    SZ(whisper_model_loader); OFF(whisper_model_loader, context); OFF(whisper_model_loader, read); OFF(whisper_model_loader, eof); OFF(whisper_model_loader, close);
    SZ(whisper_grammar_element); OFF(whisper_grammar_element, type); OFF(whisper_grammar_element, value);
    SZ(whisper_vad_params);
    OFF(whisper_vad_params, threshold); OFF(whisper_vad_params, min_speech_duration_ms); OFF(whisper_vad_params, min_silence_duration_ms);
    OFF(whisper_vad_params, max_speech_duration_s); OFF(whisper_vad_params, speech_pad_ms); OFF(whisper_vad_params, samples_overlap);
    SZ(whisper_full_params);
    OFF(whisper_full_params, strategy); OFF(whisper_full_params, n_threads); OFF(whisper_full_params, n_max_text_ctx); OFF(whisper_full_params, offset_ms);
    OFF(whisper_full_params, duration_ms); OFF(whisper_full_params, translate); OFF(whisper_full_params, no_context); OFF(whisper_full_params, no_timestamps);
    OFF(whisper_full_params, single_segment); OFF(whisper_full_params, print_special); OFF(whisper_full_params, print_progress);
    OFF(whisper_full_params, print_realtime); OFF(whisper_full_params, print_timestamps); OFF(whisper_full_params, token_timestamps);
    OFF(whisper_full_params, thold_pt); OFF(whisper_full_params, thold_ptsum); OFF(whisper_full_params, max_len); OFF(whisper_full_params, split_on_word);
    OFF(whisper_full_params, max_tokens); OFF(whisper_full_params, debug_mode); OFF(whisper_full_params, audio_ctx); OFF(whisper_full_params, tdrz_enable);
    OFF(whisper_full_params, suppress_regex); OFF(whisper_full_params, initial_prompt); OFF(whisper_full_params, prompt_tokens);
    OFF(whisper_full_params, prompt_n_tokens); OFF(whisper_full_params, language); OFF(whisper_full_params, detect_language);
    OFF(whisper_full_params, suppress_blank); OFF(whisper_full_params, suppress_nst); OFF(whisper_full_params, temperature);
    OFF(whisper_full_params, max_initial_ts); OFF(whisper_full_params, length_penalty); OFF(whisper_full_params, temperature_inc);
    OFF(whisper_full_params, entropy_thold); OFF(whisper_full_params, logprob_thold); OFF(whisper_full_params, no_speech_thold);
    OFF(whisper_full_params, greedy); OFF(whisper_full_params, greedy.best_of);
    OFF(whisper_full_params, beam_search); OFF(whisper_full_params, beam_search.beam_size); OFF(whisper_full_params, beam_search.patience);
    OFF(whisper_full_params, new_segment_callback); OFF(whisper_full_params, new_segment_callback_user_data);
    OFF(whisper_full_params, progress_callback); OFF(whisper_full_params, progress_callback_user_data);
    OFF(whisper_full_params, encoder_begin_callback); OFF(whisper_full_params, encoder_begin_callback_user_data);
    OFF(whisper_full_params, abort_callback); OFF(whisper_full_params, abort_callback_user_data);
    OFF(whisper_full_params, logits_filter_callback); OFF(whisper_full_params, logits_filter_callback_user_data);
    OFF(whisper_full_params, grammar_rules); OFF(whisper_full_params, n_grammar_rules); OFF(whisper_full_params, i_start_rule);
    OFF(whisper_full_params, grammar_penalty); OFF(whisper_full_params, vad); OFF(whisper_full_params, vad_model_path); OFF(whisper_full_params, vad_params);
    // enum values that are ABI (bindgen emits them as constants)
    printf("enum WHISPER_SAMPLING_GREEDY=%d WHISPER_SAMPLING_BEAM_SEARCH=%d\n", (int) WHISPER_SAMPLING_GREEDY, (int) WHISPER_SAMPLING_BEAM_SEARCH);
    printf("enum WHISPER_AHEADS_NONE=%d N_TOP_MOST=%d CUSTOM=%d TINY_EN=%d TINY=%d BASE_EN=%d BASE=%d SMALL_EN=%d SMALL=%d MEDIUM_EN=%d MEDIUM=%d LARGE_V1=%d LARGE_V2=%d LARGE_V3=%d LARGE_V3_TURBO=%d\n",
           (int) WHISPER_AHEADS_NONE, (int) WHISPER_AHEADS_N_TOP_MOST, (int) WHISPER_AHEADS_CUSTOM, (int) WHISPER_AHEADS_TINY_EN, (int) WHISPER_AHEADS_TINY,
           (int) WHISPER_AHEADS_BASE_EN, (int) WHISPER_AHEADS_BASE, (int) WHISPER_AHEADS_SMALL_EN, (int) WHISPER_AHEADS_SMALL, (int) WHISPER_AHEADS_MEDIUM_EN,
           (int) WHISPER_AHEADS_MEDIUM, (int) WHISPER_AHEADS_LARGE_V1, (int) WHISPER_AHEADS_LARGE_V2, (int) WHISPER_AHEADS_LARGE_V3, (int) WHISPER_AHEADS_LARGE_V3_TURBO);
    printf("enum GGML_LOG_LEVEL_NONE=%d DEBUG=%d INFO=%d WARN=%d ERROR=%d CONT=%d\n", (int) GGML_LOG_LEVEL_NONE, (int) GGML_LOG_LEVEL_DEBUG, (int) GGML_LOG_LEVEL_INFO,
           (int) GGML_LOG_LEVEL_WARN, (int) GGML_LOG_LEVEL_ERROR, (int) GGML_LOG_LEVEL_CONT);
    printf("enum WHISPER_GRETYPE_END=%d ALT=%d RULE_REF=%d CHAR=%d CHAR_NOT=%d CHAR_RNG_UPPER=%d CHAR_ALT=%d\n", (int) WHISPER_GRETYPE_END, (int) WHISPER_GRETYPE_ALT,
           (int) WHISPER_GRETYPE_RULE_REF, (int) WHISPER_GRETYPE_CHAR, (int) WHISPER_GRETYPE_CHAR_NOT, (int) WHISPER_GRETYPE_CHAR_RNG_UPPER, (int) WHISPER_GRETYPE_CHAR_ALT);
    return 0;
}

// replay_whisper_rs.cpp - a compiled caller at the drop-in boundary: replays, in whisper-rs's order and with whisper-rs's
// argument passing (the 48- and 296-byte parameter structs BY VALUE, whisper_full_params and whisper_token_data returned by value
// through the hidden sret pointer), what `WhisperContext::new_with_params`, `create_state`, `WhisperState::full` and the
// `full_get_*` getters call:
//   src/whisper_ctx.rs:33 (init_from_file_with_params_no_state), src/whisper_ctx_wrapper.rs:446 (init_state),
//   src/whisper_params.rs:50 (full_default_params), src/whisper_state.rs:301 (full_with_state), :330-606 (getters),
//   src/whisper_params.rs:415-446 (the new_segment trampoline re-enters the getters inside the call).
// The SAME source is compiled against include/whisper_amd.h + libwhisper.so (product) and against the reference's whisper.h +
// oracle/_ref/libwhisper_ref.so; tests/test_parity_r2_gpu.py requires the two programs to print the same bytes.
//   usage: replay <model> <pcm.f32> <use_gpu 0|1> [beam]
#include ABI_HEADER
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

static int g_new_segments = 0, g_tokens_seen_in_callback = 0;
static void on_new_segment(struct whisper_context *, struct whisper_state * st, int n_new, void *) {
    g_new_segments += n_new;
    const int n = whisper_full_n_segments_from_state(st);          // getters are valid inside the callback
    for (int i = n - n_new; i < n; ++i) g_tokens_seen_in_callback += whisper_full_n_tokens_from_state(st, i);
}
static void quiet(enum ggml_log_level, const char *, void *) {}

int main(int argc, char ** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model pcm.f32 use_gpu [beam]\n", argv[0]); return 2; }
    whisper_log_set(quiet, nullptr);
    std::vector<float> pcm;
    { FILE * f = fopen(argv[2], "rb"); if (!f) return 3; fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
      pcm.resize((size_t) n / 4); if (fread(pcm.data(), 4, pcm.size(), f) != pcm.size()) return 3; fclose(f); }

    struct whisper_context_params cp = whisper_context_default_params();           // 48 bytes, returned by value
    cp.use_gpu = atoi(argv[3]) != 0;
    struct whisper_context * ctx = whisper_init_from_file_with_params_no_state(argv[1], cp);      // passed by value
    if (!ctx) { printf("InitError\n"); return 1; }
    struct whisper_state * st = whisper_init_state(ctx);
    if (!st) { printf("InitError(state)\n"); return 1; }

    const bool beam = argc > 4 && strcmp(argv[4], "beam") == 0;
    struct whisper_full_params fp = whisper_full_default_params(beam ? WHISPER_SAMPLING_BEAM_SEARCH : WHISPER_SAMPLING_GREEDY);    // 296 bytes via sret
    fp.print_progress = false; fp.print_realtime = false; fp.print_timestamps = false; fp.print_special = false;
    fp.n_threads = 8;
    fp.language = "en";
    if (beam) { fp.beam_search.beam_size = 3; fp.temperature_inc = 0.0f; }
    else      { fp.greedy.best_of = 1; fp.temperature_inc = 0.0f; }
    fp.new_segment_callback = on_new_segment;
    const int rc = whisper_full_with_state(ctx, st, fp, pcm.data(), (int) pcm.size());            // 296 bytes by value (caller's stack copy)
    printf("rc %d\n", rc);
    const int n_seg = whisper_full_n_segments_from_state(st);
    printf("segments %d lang %d callback_segments %d\n", n_seg, whisper_full_lang_id_from_state(st), g_new_segments);
    int n_tok_total = 0;
    for (int i = 0; i < n_seg; ++i) {
        const char * text = whisper_full_get_segment_text_from_state(st, i);
        printf("seg %d t0 %lld t1 %lld turn %d text ", i, (long long) whisper_full_get_segment_t0_from_state(st, i),
               (long long) whisper_full_get_segment_t1_from_state(st, i), (int) whisper_full_get_segment_speaker_turn_next_from_state(st, i));
        for (const unsigned char * c = (const unsigned char *) text; *c; ++c) printf("%02x", *c);
        printf("\n");
        const int nt = whisper_full_n_tokens_from_state(st, i);
        n_tok_total += nt;
        for (int j = 0; j < nt; ++j) {
            const whisper_token_data td = whisper_full_get_token_data_from_state(st, i, j);       // 56 bytes via sret
            uint32_t pb, lb; memcpy(&pb, &td.p, 4); memcpy(&lb, &td.plog, 4);
            const float p2 = whisper_full_get_token_p_from_state(st, i, j);
            printf(" tok %d id %d tid %d p %08x plog %08x t_dtw %lld same_p %d same_id %d text_len %zu\n", j, td.id, td.tid, pb, lb, (long long) td.t_dtw,
                   (int) (memcmp(&p2, &td.p, 4) == 0), (int) (whisper_full_get_token_id_from_state(st, i, j) == td.id),
                   strlen(whisper_full_get_token_text_from_state(ctx, st, i, j)));
        }
    }
    printf("tokens %d seen_in_callback %d\n", n_tok_total, g_tokens_seen_in_callback);
    printf("n_vocab %d n_text_ctx %d n_audio_ctx %d multilingual %d eot %d sot %d beg %d lang_en %d type %s\n", whisper_n_vocab(ctx), whisper_n_text_ctx(ctx),
           whisper_n_audio_ctx(ctx), whisper_is_multilingual(ctx), whisper_token_eot(ctx), whisper_token_sot(ctx), whisper_token_beg(ctx), whisper_lang_id("en"),
           whisper_model_type_readable(ctx));
    whisper_free_state(st);
    whisper_free(ctx);
    return 0;
}

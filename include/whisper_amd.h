/*
 * whisper_amd.h - C ABI of the MI355X-native Whisper backend (libwhisper.so).
 *
 * This is the DROP-IN BOUNDARY: every entry point below has the same symbol name,
 * argument list, struct layout and return-code convention as the one whisper-rs binds through
 * bindgen over the reference's `sys/whisper.cpp/include/whisper.h` (cited per group as `ref:`),
 * so `whisper-rs-sys` can link this library instead of whisper.cpp + ggml without touching the
 * Rust sources.  Plain pointers and sizes only; no C++ / torch types.
 *
 * Layout facts verified by tests/test_abi.py against sizes measured on the reference header
 * (x86-64 SysV): whisper_context_params 48 B, whisper_full_params 296 B (`vad` at offset 260),
 * whisper_token_data 56 B.  Both params structs are passed BY VALUE, whisper_token_data and
 * whisper_full_params are also RETURNED by value (hidden sret pointer).
 *
 * Every function may be called with a null ctx/state only where the reference tolerates it.
 * No C++ exception crosses this boundary.
 */
#ifndef WHISPER_AMD_H
#define WHISPER_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#  define WHISPER_API __attribute__((visibility("default")))
#else
#  define WHISPER_API
#endif

/* ref: include/whisper.h:33-36 */
#define WHISPER_SAMPLE_RATE 16000
#define WHISPER_N_FFT       400
#define WHISPER_HOP_LENGTH  160
#define WHISPER_CHUNK_SIZE  30

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * The three ggml items that leak into whisper.h and that whisper-rs binds
 * ref: ggml/include/ggml.h:550-557 (log levels), :614 (abort cb), :2105-2109 (log cb, ggml_log_set)
 *      ggml/include/ggml-cpu.h:80-85 (ggml_cpu_has_*), src/standalone.rs:150-170
 * ---------------------------------------------------------------------------------------------- */
enum ggml_log_level {
    GGML_LOG_LEVEL_NONE  = 0,
    GGML_LOG_LEVEL_DEBUG = 1,
    GGML_LOG_LEVEL_INFO  = 2,
    GGML_LOG_LEVEL_WARN  = 3,
    GGML_LOG_LEVEL_ERROR = 4,
    GGML_LOG_LEVEL_CONT  = 5,
};
typedef void (*ggml_log_callback)(enum ggml_log_level level, const char * text, void * user_data);
typedef bool (*ggml_abort_callback)(void * data);

WHISPER_API void ggml_log_set(ggml_log_callback log_callback, void * user_data);
WHISPER_API int  ggml_cpu_has_avx (void);
WHISPER_API int  ggml_cpu_has_avx2(void);
WHISPER_API int  ggml_cpu_has_fma (void);
WHISPER_API int  ggml_cpu_has_f16c(void);
/* ref: ggml/include/ggml-backend.h (ggml_backend_load_all): the reference's own examples call it before whisper_init_* (examples/bench/
 * bench.cpp:159, examples/cli); here there is one built-in backend and nothing to load - a no-op kept so that they link unmodified. */
WHISPER_API void ggml_backend_load_all(void);

/* ------------------------------------------------------------------------------------------------
 * Opaque handles and scalar typedefs            ref: include/whisper.h:80-86
 * ---------------------------------------------------------------------------------------------- */
struct whisper_context;      /* read-only model: weights resident in HBM, vocab, filters          */
struct whisper_state;        /* per-stream mutable state: mel, activations, KV caches, results    */
struct whisper_full_params;
struct whisper_vad_context;  /* VAD is out of scope: constructors return NULL */
struct whisper_vad_segments;

typedef int32_t whisper_pos;
typedef int32_t whisper_token;
typedef int32_t whisper_seq_id;

/* ref: include/whisper.h:88-104 - numeric order is ABI */
enum whisper_alignment_heads_preset {
    WHISPER_AHEADS_NONE,
    WHISPER_AHEADS_N_TOP_MOST,
    WHISPER_AHEADS_CUSTOM,
    WHISPER_AHEADS_TINY_EN,
    WHISPER_AHEADS_TINY,
    WHISPER_AHEADS_BASE_EN,
    WHISPER_AHEADS_BASE,
    WHISPER_AHEADS_SMALL_EN,
    WHISPER_AHEADS_SMALL,
    WHISPER_AHEADS_MEDIUM_EN,
    WHISPER_AHEADS_MEDIUM,
    WHISPER_AHEADS_LARGE_V1,
    WHISPER_AHEADS_LARGE_V2,
    WHISPER_AHEADS_LARGE_V3,
    WHISPER_AHEADS_LARGE_V3_TURBO,
};

/* ref: include/whisper.h:106-114 */
typedef struct whisper_ahead  { int n_text_layer; int n_head; } whisper_ahead;
typedef struct whisper_aheads { size_t n_heads; const whisper_ahead * heads; } whisper_aheads;

/* ref: include/whisper.h:116-129 (48 bytes).  `use_gpu=false` is rejected by this backend: there
 * is no CPU path in the product (init returns NULL with an error log). `gpu_device` = HIP device. */
struct whisper_context_params {
    bool  use_gpu;
    bool  flash_attn;            /* false (default): reference summation order, bit-identical to whisper.cpp CPU; true: F16-MFMA encoder / prompt
                                    products (tolerance-level parity) and, as in the reference, DTW off (whisper.cpp:3724-3727) */
    int   gpu_device;
    bool  dtw_token_timestamps;
    enum whisper_alignment_heads_preset dtw_aheads_preset;
    int   dtw_n_top;
    struct whisper_aheads dtw_aheads;
    size_t dtw_mem_size;
};

/* ref: include/whisper.h:131-151 (56 bytes) */
typedef struct whisper_token_data {
    whisper_token id;
    whisper_token tid;
    float   p;
    float   plog;
    float   pt;
    float   ptsum;
    int64_t t0;
    int64_t t1;
    int64_t t_dtw;
    float   vlen;
} whisper_token_data;

/* ref: include/whisper.h:153-159 */
typedef struct whisper_model_loader {
    void * context;
    size_t (*read)(void * ctx, void * output, size_t read_size);
    bool   (*eof)(void * ctx);
    void   (*close)(void * ctx);
} whisper_model_loader;

/* ref: include/whisper.h:162-190 (grammar types are part of whisper_full_params' layout; grammar
 * sampling itself is out of scope, SURVEY.md §2 row 17: n_grammar_rules > 0 is ignored with a warning) */
enum whisper_gretype {
    WHISPER_GRETYPE_END            = 0,
    WHISPER_GRETYPE_ALT            = 1,
    WHISPER_GRETYPE_RULE_REF       = 2,
    WHISPER_GRETYPE_CHAR           = 3,
    WHISPER_GRETYPE_CHAR_NOT       = 4,
    WHISPER_GRETYPE_CHAR_RNG_UPPER = 5,
    WHISPER_GRETYPE_CHAR_ALT       = 6,
};
typedef struct whisper_grammar_element { enum whisper_gretype type; uint32_t value; } whisper_grammar_element;

/* ref: include/whisper.h:192-199 */
typedef struct whisper_vad_params {
    float threshold;
    int   min_speech_duration_ms;
    int   min_silence_duration_ms;
    float max_speech_duration_s;
    int   speech_pad_ms;
    float samples_overlap;
} whisper_vad_params;

/* ref: include/whisper.h:453-456 */
enum whisper_sampling_strategy { WHISPER_SAMPLING_GREEDY, WHISPER_SAMPLING_BEAM_SEARCH };

/* ref: include/whisper.h:458-480 - callbacks run on the thread that called whisper_full* */
typedef void (*whisper_new_segment_callback)(struct whisper_context * ctx, struct whisper_state * state, int n_new, void * user_data);
typedef void (*whisper_progress_callback)(struct whisper_context * ctx, struct whisper_state * state, int progress, void * user_data);
typedef bool (*whisper_encoder_begin_callback)(struct whisper_context * ctx, struct whisper_state * state, void * user_data);
typedef void (*whisper_logits_filter_callback)(struct whisper_context * ctx, struct whisper_state * state,
                                               const whisper_token_data * tokens, int n_tokens, float * logits, void * user_data);

/* ref: include/whisper.h:485-588 (296 bytes; field order is ABI) */
struct whisper_full_params {
    enum whisper_sampling_strategy strategy;

    int n_threads;               /* advisory on this backend */
    int n_max_text_ctx;
    int offset_ms;
    int duration_ms;

    bool translate;
    bool no_context;
    bool no_timestamps;
    bool single_segment;
    bool print_special;
    bool print_progress;
    bool print_realtime;
    bool print_timestamps;

    bool  token_timestamps;      /* heuristic per-token t0 / t1 / vlen (whisper.cpp:8326-8616), with max_len / split_on_word wrapping */
    float thold_pt;
    float thold_ptsum;
    int   max_len;
    bool  split_on_word;
    int   max_tokens;

    bool debug_mode;
    int  audio_ctx;

    bool tdrz_enable;

    const char * suppress_regex;

    const char * initial_prompt;
    const whisper_token * prompt_tokens;
    int prompt_n_tokens;

    const char * language;
    bool detect_language;

    bool suppress_blank;
    bool suppress_nst;

    float temperature;
    float max_initial_ts;
    float length_penalty;

    float temperature_inc;
    float entropy_thold;
    float logprob_thold;
    float no_speech_thold;

    struct { int best_of; } greedy;
    struct { int beam_size; float patience; } beam_search;

    whisper_new_segment_callback new_segment_callback;
    void * new_segment_callback_user_data;
    whisper_progress_callback progress_callback;
    void * progress_callback_user_data;
    whisper_encoder_begin_callback encoder_begin_callback;
    void * encoder_begin_callback_user_data;
    ggml_abort_callback abort_callback;
    void * abort_callback_user_data;
    whisper_logits_filter_callback logits_filter_callback;
    void * logits_filter_callback_user_data;

    const whisper_grammar_element ** grammar_rules;
    size_t n_grammar_rules;
    size_t i_start_rule;
    float  grammar_penalty;

    bool         vad;
    const char * vad_model_path;
    whisper_vad_params vad_params;
};

/* ref: include/whisper.h:436-442 */
struct whisper_timings { float sample_ms, encode_ms, decode_ms, batchd_ms, prompt_ms; };

/* ref: include/whisper.h:679-683 */
struct whisper_vad_context_params { int n_threads; bool use_gpu; int gpu_device; };

/* ------------------------------------------------------------------------------------------------
 * Model load / state / free
 * ref: include/whisper.h:204-239, 266-269; engine whisper.cpp:3640-3889.  NULL on failure.
 * whisper-rs calls only the *_no_state constructors + whisper_init_state
 * (src/whisper_ctx.rs:33,60; src/whisper_ctx_wrapper.rs:446).
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API struct whisper_context * whisper_init_from_file_with_params  (const char * path_model, struct whisper_context_params params);
WHISPER_API struct whisper_context * whisper_init_from_buffer_with_params(void * buffer, size_t buffer_size, struct whisper_context_params params);
WHISPER_API struct whisper_context * whisper_init_with_params            (struct whisper_model_loader * loader, struct whisper_context_params params);
WHISPER_API struct whisper_context * whisper_init_from_file_with_params_no_state  (const char * path_model, struct whisper_context_params params);
WHISPER_API struct whisper_context * whisper_init_from_buffer_with_params_no_state(void * buffer, size_t buffer_size, struct whisper_context_params params);
WHISPER_API struct whisper_context * whisper_init_with_params_no_state            (struct whisper_model_loader * loader, struct whisper_context_params params);
WHISPER_API struct whisper_context * whisper_init_from_file           (const char * path_model);
WHISPER_API struct whisper_context * whisper_init_from_buffer         (void * buffer, size_t buffer_size);
WHISPER_API struct whisper_context * whisper_init                     (struct whisper_model_loader * loader);
WHISPER_API struct whisper_context * whisper_init_from_file_no_state  (const char * path_model);
WHISPER_API struct whisper_context * whisper_init_from_buffer_no_state(void * buffer, size_t buffer_size);
WHISPER_API struct whisper_context * whisper_init_no_state            (struct whisper_model_loader * loader);

WHISPER_API struct whisper_state * whisper_init_state(struct whisper_context * ctx);

WHISPER_API void whisper_free               (struct whisper_context * ctx);
WHISPER_API void whisper_free_state         (struct whisper_state * state);
WHISPER_API void whisper_free_params        (struct whisper_full_params * params);
WHISPER_API void whisper_free_context_params(struct whisper_context_params * params);

/* ref: include/whisper.h:252-263 - OpenVINO is another vendor's runtime: always returns 1 ("not enabled") */
WHISPER_API int whisper_ctx_init_openvino_encoder_with_state(struct whisper_context * ctx, struct whisper_state * state,
                                                             const char * model_path, const char * device, const char * cache_dir);
WHISPER_API int whisper_ctx_init_openvino_encoder(struct whisper_context * ctx, const char * model_path,
                                                  const char * device, const char * cache_dir);

/* ------------------------------------------------------------------------------------------------
 * Stage API: PCM -> log-mel -> encoder (+cross K/V) -> decoder logits
 * ref: include/whisper.h:274-338, 413-414; engine whisper.cpp:3891-3971.
 * `samples` may be a host pointer or a HIP device pointer (detected with hipPointerGetAttributes).
 * Returns: pcm_to_mel/encode 0 | -1; set_mel 0 | -1 (n_mel mismatch); decode 0 | 1.
 * whisper_get_logits*: row n_tokens-1 of the last decode call is valid (whisper.cpp:2965-2971).
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API int whisper_pcm_to_mel           (struct whisper_context * ctx, const float * samples, int n_samples, int n_threads);
WHISPER_API int whisper_pcm_to_mel_with_state(struct whisper_context * ctx, struct whisper_state * state, const float * samples, int n_samples, int n_threads);
WHISPER_API int whisper_set_mel              (struct whisper_context * ctx, const float * data, int n_len, int n_mel);
WHISPER_API int whisper_set_mel_with_state   (struct whisper_context * ctx, struct whisper_state * state, const float * data, int n_len, int n_mel);
WHISPER_API int whisper_encode               (struct whisper_context * ctx, int offset, int n_threads);
WHISPER_API int whisper_encode_with_state    (struct whisper_context * ctx, struct whisper_state * state, int offset, int n_threads);
WHISPER_API int whisper_decode               (struct whisper_context * ctx, const whisper_token * tokens, int n_tokens, int n_past, int n_threads);
WHISPER_API int whisper_decode_with_state    (struct whisper_context * ctx, struct whisper_state * state, const whisper_token * tokens, int n_tokens, int n_past, int n_threads);
WHISPER_API float * whisper_get_logits           (struct whisper_context * ctx);
WHISPER_API float * whisper_get_logits_from_state(struct whisper_state * state);

/* ------------------------------------------------------------------------------------------------
 * Tokenizer / languages                        ref: include/whisper.h:345-387; whisper.cpp:3973-4110
 * tokenize: count, or -(needed) when n_max_tokens is too small.
 * lang_auto_detect: id >= 0, or -1 (offset<0), -2 (offset past end), -6 (encode), -7 (decode).
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API int whisper_tokenize   (struct whisper_context * ctx, const char * text, whisper_token * tokens, int n_max_tokens);
WHISPER_API int whisper_token_count(struct whisper_context * ctx, const char * text);
WHISPER_API int          whisper_lang_max_id  (void);
WHISPER_API int          whisper_lang_id      (const char * lang);
WHISPER_API const char * whisper_lang_str     (int id);
WHISPER_API const char * whisper_lang_str_full(int id);
WHISPER_API int whisper_lang_auto_detect           (struct whisper_context * ctx, int offset_ms, int n_threads, float * lang_probs);
WHISPER_API int whisper_lang_auto_detect_with_state(struct whisper_context * ctx, struct whisper_state * state, int offset_ms, int n_threads, float * lang_probs);

/* ------------------------------------------------------------------------------------------------
 * Model / vocabulary introspection              ref: include/whisper.h:389-433; whisper.cpp:4120-4259
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API int whisper_n_len           (struct whisper_context * ctx);
WHISPER_API int whisper_n_len_from_state(struct whisper_state * state);   /* = n_len_org (whisper.cpp:4185-4187) */
WHISPER_API int whisper_n_vocab         (struct whisper_context * ctx);
WHISPER_API int whisper_n_text_ctx      (struct whisper_context * ctx);
WHISPER_API int whisper_n_audio_ctx     (struct whisper_context * ctx);
WHISPER_API int whisper_is_multilingual (struct whisper_context * ctx);

WHISPER_API int whisper_model_n_vocab      (struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_ctx  (struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_state(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_head (struct whisper_context * ctx);
WHISPER_API int whisper_model_n_audio_layer(struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_ctx   (struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_state (struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_head  (struct whisper_context * ctx);
WHISPER_API int whisper_model_n_text_layer (struct whisper_context * ctx);
WHISPER_API int whisper_model_n_mels       (struct whisper_context * ctx);
WHISPER_API int whisper_model_ftype        (struct whisper_context * ctx);
WHISPER_API int whisper_model_type         (struct whisper_context * ctx);
WHISPER_API const char * whisper_model_type_readable(struct whisper_context * ctx);

WHISPER_API const char * whisper_token_to_str(struct whisper_context * ctx, whisper_token token);
WHISPER_API whisper_token whisper_token_eot (struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_sot (struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_solm(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_prev(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_nosp(struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_not (struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_beg (struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_lang(struct whisper_context * ctx, int lang_id);
WHISPER_API whisper_token whisper_token_translate (struct whisper_context * ctx);
WHISPER_API whisper_token whisper_token_transcribe(struct whisper_context * ctx);

/* ------------------------------------------------------------------------------------------------
 * Timings / system info / logging       ref: include/whisper.h:443-448, 729; whisper.cpp:4261-4355
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API struct whisper_timings * whisper_get_timings(struct whisper_context * ctx);
WHISPER_API void whisper_print_timings(struct whisper_context * ctx);
WHISPER_API void whisper_reset_timings(struct whisper_context * ctx);
WHISPER_API const char * whisper_print_system_info(void);
WHISPER_API void whisper_log_set(ggml_log_callback log_callback, void * user_data);

/* ------------------------------------------------------------------------------------------------
 * Full pipeline                                ref: include/whisper.h:591-623; whisper.cpp:5898-6019,
 *                                                   6795-7864
 * whisper_full_with_state: 0 ok; -2 mel; -3 language detect; -4 too many decoders; -5 audio_ctx;
 * -6 encode; -7 KV realloc; -8 / -9 decode   (same codes as whisper.cpp:6810-7433).
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API struct whisper_context_params * whisper_context_default_params_by_ref(void);
WHISPER_API struct whisper_context_params   whisper_context_default_params       (void);
WHISPER_API struct whisper_full_params * whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy);
WHISPER_API struct whisper_full_params   whisper_full_default_params       (enum whisper_sampling_strategy strategy);

WHISPER_API int whisper_full           (struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples);
WHISPER_API int whisper_full_with_state(struct whisper_context * ctx, struct whisper_state * state, struct whisper_full_params params, const float * samples, int n_samples);
WHISPER_API int whisper_full_parallel  (struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples, int n_processors);

/* ------------------------------------------------------------------------------------------------
 * Result getters (valid until the next whisper_full* on the same state; readable from inside
 * new_segment_callback)                         ref: include/whisper.h:627-669, 732-733; whisper.cpp:7866-8033
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API int     whisper_full_n_segments           (struct whisper_context * ctx);
WHISPER_API int     whisper_full_n_segments_from_state(struct whisper_state * state);
WHISPER_API int     whisper_full_lang_id              (struct whisper_context * ctx);
WHISPER_API int     whisper_full_lang_id_from_state   (struct whisper_state * state);
WHISPER_API int64_t whisper_full_get_segment_t0           (struct whisper_context * ctx, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t0_from_state(struct whisper_state * state, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t1           (struct whisper_context * ctx, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t1_from_state(struct whisper_state * state, int i_segment);
WHISPER_API bool    whisper_full_get_segment_speaker_turn_next           (struct whisper_context * ctx, int i_segment);
WHISPER_API bool    whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state * state, int i_segment);
WHISPER_API const char * whisper_full_get_segment_text           (struct whisper_context * ctx, int i_segment);
WHISPER_API const char * whisper_full_get_segment_text_from_state(struct whisper_state * state, int i_segment);
WHISPER_API int     whisper_full_n_tokens           (struct whisper_context * ctx, int i_segment);
WHISPER_API int     whisper_full_n_tokens_from_state(struct whisper_state * state, int i_segment);
WHISPER_API const char * whisper_full_get_token_text           (struct whisper_context * ctx, int i_segment, int i_token);
WHISPER_API const char * whisper_full_get_token_text_from_state(struct whisper_context * ctx, struct whisper_state * state, int i_segment, int i_token);
WHISPER_API whisper_token whisper_full_get_token_id           (struct whisper_context * ctx, int i_segment, int i_token);
WHISPER_API whisper_token whisper_full_get_token_id_from_state(struct whisper_state * state, int i_segment, int i_token);
WHISPER_API whisper_token_data whisper_full_get_token_data           (struct whisper_context * ctx, int i_segment, int i_token);
WHISPER_API whisper_token_data whisper_full_get_token_data_from_state(struct whisper_state * state, int i_segment, int i_token);
WHISPER_API float   whisper_full_get_token_p           (struct whisper_context * ctx, int i_segment, int i_token);
WHISPER_API float   whisper_full_get_token_p_from_state(struct whisper_state * state, int i_segment, int i_token);
WHISPER_API float   whisper_full_get_segment_no_speech_prob           (struct whisper_context * ctx, int i_segment);
WHISPER_API float   whisper_full_get_segment_no_speech_prob_from_state(struct whisper_state * state, int i_segment);

/* ------------------------------------------------------------------------------------------------
 * VAD + bench helpers: exported for link completeness only ("stubs returning
 * unsupported are acceptable").             ref: include/whisper.h:677-725
 * ---------------------------------------------------------------------------------------------- */
WHISPER_API struct whisper_vad_params         whisper_vad_default_params(void);
WHISPER_API struct whisper_vad_context_params whisper_vad_default_context_params(void);
WHISPER_API struct whisper_vad_context * whisper_vad_init_from_file_with_params(const char * path_model, struct whisper_vad_context_params params);
WHISPER_API struct whisper_vad_context * whisper_vad_init_with_params          (struct whisper_model_loader * loader, struct whisper_vad_context_params params);
WHISPER_API bool    whisper_vad_detect_speech(struct whisper_vad_context * vctx, const float * samples, int n_samples);
WHISPER_API int     whisper_vad_n_probs(struct whisper_vad_context * vctx);
WHISPER_API float * whisper_vad_probs  (struct whisper_vad_context * vctx);
WHISPER_API struct whisper_vad_segments * whisper_vad_segments_from_probs  (struct whisper_vad_context * vctx, struct whisper_vad_params params);
WHISPER_API struct whisper_vad_segments * whisper_vad_segments_from_samples(struct whisper_vad_context * vctx, struct whisper_vad_params params, const float * samples, int n_samples);
WHISPER_API int   whisper_vad_segments_n_segments(struct whisper_vad_segments * segments);
WHISPER_API float whisper_vad_segments_get_segment_t0(struct whisper_vad_segments * segments, int i_segment);
WHISPER_API float whisper_vad_segments_get_segment_t1(struct whisper_vad_segments * segments, int i_segment);
WHISPER_API void  whisper_vad_free_segments(struct whisper_vad_segments * segments);
WHISPER_API void  whisper_vad_free         (struct whisper_vad_context  * ctx);

WHISPER_API int          whisper_bench_memcpy          (int n_threads);
WHISPER_API const char * whisper_bench_memcpy_str      (int n_threads);
WHISPER_API int          whisper_bench_ggml_mul_mat    (int n_threads);
WHISPER_API const char * whisper_bench_ggml_mul_mat_str(int n_threads);

/* ================================================================================================
 * Extensions of this backend (not in the reference header; prefix whisper_amd_).
 * They exist for tests, the bench and multi-GPU chunk sharding; whisper-rs never needs them.
 * ============================================================================================== */

/* sizeof/offsetof self-description so a binding can verify its struct layout at run time:
 * out[0]=sizeof(context_params) out[1]=sizeof(full_params) out[2]=sizeof(token_data)
 * out[3]=offsetof(full_params, vad) out[4]=offsetof(full_params, greedy) out[5]=offsetof(full_params, language) */
WHISPER_API void whisper_amd_abi_sizes(size_t out[6]);

/* Device-resident intermediates for parity tests.  Each copies up to `cap` elements into `dst`
 * (host) and returns the full element count (call with cap=0 to size).  -1 when not available.
 *   mel:      [n_mel][n_len] f32 as produced by whisper_pcm_to_mel*  (whisper.cpp:3186-3276)
 *   embd_enc: [n_audio_ctx][n_state] f32 encoder output              (whisper.cpp:2056-2287)
 *   embd_conv:[n_audio_ctx][n_state] f32 conv-stem output (+GELU), BEFORE the positional add */
WHISPER_API int64_t whisper_amd_get_mel      (struct whisper_state * state, float * dst, int64_t cap, int * n_len, int * n_mel);
WHISPER_API int64_t whisper_amd_get_embd_enc (struct whisper_state * state, float * dst, int64_t cap);
WHISPER_API int64_t whisper_amd_get_embd_conv(struct whisper_state * state, float * dst, int64_t cap);

/* Per-state stage timers in microseconds + call counts (the reference keeps them per state but
 * only prints ctx->state's, whisper.cpp:868-881,4274-4296):
 * out = { t_sample, t_encode, t_decode, t_batchd, t_prompt, t_mel, n_sample, n_encode, n_decode,
 *         n_batchd, n_prompt, n_fail_p + n_fail_h } */
WHISPER_API void whisper_amd_get_timings_us(struct whisper_state * state, int64_t out[12]);
WHISPER_API void whisper_amd_reset_timings (struct whisper_state * state);

/* The F16 GELU table the device kernels use (65536 entries; vec.h:571-585 semantics). */
WHISPER_API void whisper_amd_gelu_table_f16(uint16_t * dst);

/* HIP stream (hipStream_t as void*) every kernel of this state is launched on - for external
 * event timing (bench.py) and stream-ordered hand-off of device PCM buffers. */
WHISPER_API void * whisper_amd_state_stream(struct whisper_state * state);

/* Decode-loop bookkeeping of decoder `j` after whisper_full* (for tests: compares the loop's internal decisions with
 * the reference even when no segment was emitted): out = { failed, completed, has_ts, seek_delta, result_len,
 * avg_logprobs, entropy, no_speech_prob }; ids receives up to max_ids token ids; returns the token count. */
WHISPER_API int whisper_amd_decoder_info(struct whisper_state * state, int j, double out[8], int32_t * ids, int max_ids);

/* Measurement helper (bench.py): replays the single-token decoder pass `n_iters` times back to back on the
 * state's stream between two HIP events and returns the average DEVICE time of one decode step in ms.
 * Needs a prior whisper_encode*; touches KV cell `n_past` of the state.  0 on success. */
WHISPER_API int whisper_amd_decode_step_probe(struct whisper_context * ctx, struct whisper_state * state, int n_past, int n_iters, float * ms_per_step);

/* 1 when the state runs the single-token decoder pass as ONE persistent launch (wa_mega.hip), 0 when it replays the
 * captured launch sequence (WHISPER_AMD_NO_MEGA=1, unsupported shape, or after a hand-off time-out). */
WHISPER_API int whisper_amd_mega_enabled(struct whisper_state * state);
/* Role of workgroup `wg` of the n_wg workgroups of the one-launch step for a model with n_head text heads (no device needed): role 0 =
 * weight streaming (index = its rank), 1 = self-attention of head `index`, 2 = cross-attention (index = 4 head + quarter). */
WHISPER_API void whisper_amd_mega_role_of(int n_wg, int n_head, int wg, int * role, int * index);

/* Host-overlapped greedy decoding (the device decodes its own prediction of the next token while the host applies the
 * reference's sampling rules to the previous logits): out = { predictions confirmed, predictions wrong (step redone) }
 * since the last whisper_amd_reset_timings. */
WHISPER_API void whisper_amd_overlap_stats(struct whisper_state * state, int out[2]);

/* Debugging aid (tools/mega_check.py): runs the one-launch step for (token, position n_past) on KV cell n_past and copies
 * out the hand-off granules [n_text_layer][8][2 * n_text_state] (tag << 32 | value bits) and the logits [n_vocab].
 * Returns the step's status word (0 = ok), < 0 when the one-launch step is not available. */
WHISPER_API int whisper_amd_mega_debug(struct whisper_context * ctx, struct whisper_state * state, int token, int n_past,
                                       unsigned long long * granules_out, float * logits_out);
/* The same step through the launch sequence, stage by stage: values only, in the granule layout above (32-bit words). */
WHISPER_API int whisper_amd_seq_debug(struct whisper_context * ctx, struct whisper_state * state, int token, int n_past, unsigned * values_out,
                                      float * logits_out);

/* The decode step for 2..8 token rows as ONE launch (wa_rows.hip: beam / best_of steps, small batches, lock-step chunks):
 * out = { passes served by it, passes sent to the launch sequence after a status word } since the state was created. */
WHISPER_API void whisper_amd_rows_stats(struct whisper_state * state, long out[2]);
WHISPER_API int  whisper_amd_rows_enabled(struct whisper_state * state);
/* Debugging aid (tools/rows_check.py): B identical rows (token, position n_past) of this state through the several-rows step; copies out the
 * hand-off granules [n_text_layer][8][B][2 * n_text_state] and the logits [B][n_vocab].  Returns the status word (0 = ok), < 0: not available. */
WHISPER_API int whisper_amd_rows_debug(struct whisper_context * ctx, struct whisper_state * state, int B, int token, int n_past,
                                       unsigned long long * granules_out, float * logits_out);
/* Measurement helper (bench.py): average DEVICE time of one B-row step, row i on states[i]'s cells and encoder K / V (null = states[0]). */
WHISPER_API int whisper_amd_rows_step_probe(struct whisper_context * ctx, struct whisper_state ** states, int B, int n_past, int n_iters, float * ms_per_step);

/* Chunk-parallel transcription on ONE device (the reference's model: one state + thread per chunk, whisper.cpp:7771-7806):
 * runs `n_chunks` independent whisper_full_with_state jobs, each on its own state / HIP stream / host thread; mel, encoder and
 * prompts overlap on the device, and the single-token decode steps of the chunks are served in LOCK STEP - one decoder pass reads
 * every weight row once for all chunks' tokens (F16 models; up to 8 rows per pass).  Results are those of each chunk alone.
 * samples[i] may be host or device pointers.  Returns 0 or the first non-zero per-chunk code. */
WHISPER_API int whisper_amd_full_batch(struct whisper_context * ctx, struct whisper_state ** states, int n_chunks,
                                       struct whisper_full_params params, const float * const * samples, const int * n_samples);
/* how the last whisper_amd_full_batch call on this context decoded: lock-step passes and the token rows they served */
WHISPER_API void whisper_amd_batch_stats(struct whisper_context * ctx, long * steps, long * rows);
/* ... and how many of those passes were ONE launch (wa_rows.hip) rather than the launch sequence */
WHISPER_API long whisper_amd_batch_one_launch(struct whisper_context * ctx);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_AMD_H */

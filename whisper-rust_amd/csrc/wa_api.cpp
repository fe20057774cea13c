// wa_api.cpp - the C ABI (include/whisper_amd.h): constructors, stage API, getters, language table,
// tokenizer, logging, default parameters.  Everything here is host-side glue; the compute lives in
// wa_encode.cpp / wa_decode.cpp / wa_kernels.hip and the decode loop in wa_full.cpp.
#include "wa_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <regex>
#include <thread>

// -------------------------------------------------------------------------------------------------
// logging / time / HIP errors
// -------------------------------------------------------------------------------------------------
static void wa_log_default(ggml_log_level, const char * text, void *) { fputs(text, stderr); fflush(stderr); }
static ggml_log_callback g_log_cb = wa_log_default;
static void * g_log_ud = nullptr;

void wa_log(ggml_log_level level, const char * fmt, ...) {
    if (!g_log_cb) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (n < (int) sizeof(buf)) { g_log_cb(level, buf, g_log_ud); return; }
    std::vector<char> big(n + 1);
    va_start(ap, fmt);
    vsnprintf(big.data(), big.size(), fmt, ap);
    va_end(ap);
    g_log_cb(level, big.data(), g_log_ud);
}

int64_t wa_time_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool wa_hip_ok(hipError_t e, const char * what, const char * file, int line) {
    if (e == hipSuccess) return true;
    WA_ERROR("HIP error %d (%s) at %s:%d: %s\n", (int) e, hipGetErrorString(e), file, line, what);
    return false;
}

extern "C" {

void whisper_log_set(ggml_log_callback cb, void * ud) { g_log_cb = cb ? cb : wa_log_default; g_log_ud = ud; }   // whisper.cpp:8935-8939
void ggml_log_set(ggml_log_callback cb, void * ud)    { whisper_log_set(cb, ud); }

// This library never runs the reference's CPU kernels; the four probes whisper-rs binds
// (src/standalone.rs:150-170) report the build host's ISA like ggml-cpu does.
int ggml_cpu_has_avx (void) { return __builtin_cpu_supports("avx")  ? 1 : 0; }
int ggml_cpu_has_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }
int ggml_cpu_has_fma (void) { return __builtin_cpu_supports("fma")  ? 1 : 0; }
int ggml_cpu_has_f16c(void) { return __builtin_cpu_supports("f16c") ? 1 : 0; }
void ggml_backend_load_all(void) {}

void whisper_amd_abi_sizes(size_t out[6]) {
    out[0] = sizeof(whisper_context_params); out[1] = sizeof(whisper_full_params); out[2] = sizeof(whisper_token_data);
    out[3] = offsetof(whisper_full_params, vad); out[4] = offsetof(whisper_full_params, greedy); out[5] = offsetof(whisper_full_params, language);
}

// how the last whisper_amd_full_batch call on this context decoded: lock-step passes and the token rows they served
void whisper_amd_batch_stats(struct whisper_context * ctx, long * steps, long * rows) { if (ctx) { *steps = ctx->batch_steps; *rows = ctx->batch_rows; } }
long whisper_amd_batch_one_launch(struct whisper_context * ctx) { return ctx ? ctx->batch_one_launch : 0; }

} // extern "C"

// -------------------------------------------------------------------------------------------------
// languages (ids are ABI: token = sot + 1 + id).  ref: whisper.cpp:313-414
// -------------------------------------------------------------------------------------------------
static const char * const k_lang[100][2] = {
    {"en","english"},{"zh","chinese"},{"de","german"},{"es","spanish"},{"ru","russian"},{"ko","korean"},{"fr","french"},
    {"ja","japanese"},{"pt","portuguese"},{"tr","turkish"},{"pl","polish"},{"ca","catalan"},{"nl","dutch"},{"ar","arabic"},
    {"sv","swedish"},{"it","italian"},{"id","indonesian"},{"hi","hindi"},{"fi","finnish"},{"vi","vietnamese"},{"he","hebrew"},
    {"uk","ukrainian"},{"el","greek"},{"ms","malay"},{"cs","czech"},{"ro","romanian"},{"da","danish"},{"hu","hungarian"},
    {"ta","tamil"},{"no","norwegian"},{"th","thai"},{"ur","urdu"},{"hr","croatian"},{"bg","bulgarian"},{"lt","lithuanian"},
    {"la","latin"},{"mi","maori"},{"ml","malayalam"},{"cy","welsh"},{"sk","slovak"},{"te","telugu"},{"fa","persian"},
    {"lv","latvian"},{"bn","bengali"},{"sr","serbian"},{"az","azerbaijani"},{"sl","slovenian"},{"kn","kannada"},
    {"et","estonian"},{"mk","macedonian"},{"br","breton"},{"eu","basque"},{"is","icelandic"},{"hy","armenian"},{"ne","nepali"},
    {"mn","mongolian"},{"bs","bosnian"},{"kk","kazakh"},{"sq","albanian"},{"sw","swahili"},{"gl","galician"},{"mr","marathi"},
    {"pa","punjabi"},{"si","sinhala"},{"km","khmer"},{"sn","shona"},{"yo","yoruba"},{"so","somali"},{"af","afrikaans"},
    {"oc","occitan"},{"ka","georgian"},{"be","belarusian"},{"tg","tajik"},{"sd","sindhi"},{"gu","gujarati"},{"am","amharic"},
    {"yi","yiddish"},{"lo","lao"},{"uz","uzbek"},{"fo","faroese"},{"ht","haitian creole"},{"ps","pashto"},{"tk","turkmen"},
    {"nn","nynorsk"},{"mt","maltese"},{"sa","sanskrit"},{"lb","luxembourgish"},{"my","myanmar"},{"bo","tibetan"},
    {"tl","tagalog"},{"mg","malagasy"},{"as","assamese"},{"tt","tatar"},{"haw","hawaiian"},{"ln","lingala"},{"ha","hausa"},
    {"ba","bashkir"},{"jw","javanese"},{"su","sundanese"},{"yue","cantonese"},
};
#define WA_N_LANG 100

// -------------------------------------------------------------------------------------------------
// tokenizer (whisper_tokenize, params.initial_prompt).  Contract: whisper.cpp:3288-3336 - GPT-2's pre-tokenisation pattern, then
// every piece is covered left to right by the LONGEST vocabulary entry that starts at the current byte; a byte no entry starts at is
// skipped with an error log.  Pinned by the reference engine's goldens (tests/golden/r2_cases.json: "tokenize", initial_prompt).
// -------------------------------------------------------------------------------------------------
std::vector<int> wa_tokenize(const wa_vocab & vocab, const std::string & text) {
    static const std::regex piece_re(R"('s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+)");
    size_t longest = 1;                                    // no entry is longer: bounds the candidate lengths tried per position
    for (const auto & kv : vocab.token_to_id) longest = std::max(longest, kv.first.size());
    std::vector<int> ids;
    std::string cand;
    for (std::sregex_iterator it(text.begin(), text.end(), piece_re), end; it != end; ++it) {
        const std::string piece = it->str();
        for (size_t at = 0; at < piece.size();) {
            size_t took = 0;
            for (size_t len = std::min(longest, piece.size() - at); len > 0 && took == 0; --len) {
                cand.assign(piece, at, len);
                const auto hit = vocab.token_to_id.find(cand);
                if (hit != vocab.token_to_id.end()) { ids.push_back(hit->second); took = len; }
            }
            if (took == 0) { WA_ERROR("unknown token\n"); took = 1; }
            at += took;
        }
    }
    return ids;
}

// -------------------------------------------------------------------------------------------------
// model loaders (file / buffer) - ref: whisper.cpp:3640-3749
// -------------------------------------------------------------------------------------------------
namespace {
struct file_src { FILE * f; bool eof; };
struct buf_src  { const uint8_t * p; size_t size, off; };

whisper_context * init_common(whisper_model_loader * loader, whisper_context_params params) {
    if (params.flash_attn && params.dtw_token_timestamps) {
        WA_WARN("%s: dtw_token_timestamps is not supported with flash_attn - disabling\n", __func__);
        params.dtw_token_timestamps = false;
    }
    WA_INFO("%s: use gpu    = %d\n", __func__, params.use_gpu);
    WA_INFO("%s: flash attn = %d\n", __func__, params.flash_attn);
    WA_INFO("%s: gpu_device = %d\n", __func__, params.gpu_device);
    WA_INFO("%s: dtw        = %d\n", __func__, params.dtw_token_timestamps);

    if (!params.use_gpu) {
        // no CPU fallback exists in this library by design
        WA_ERROR("%s: use_gpu=false requested, but this backend has no CPU path (MI355X/HIP only)\n", __func__);
        loader->close(loader->context);
        return nullptr;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        WA_ERROR("%s: no HIP device available\n", __func__);
        loader->close(loader->context);
        return nullptr;
    }
    if (params.gpu_device < 0 || params.gpu_device >= n_dev) {
        WA_ERROR("%s: gpu_device %d out of range (%d devices)\n", __func__, params.gpu_device, n_dev);
        loader->close(loader->context);
        return nullptr;
    }
    // spin-wait synchronisation: the decode loop synchronises once per token, an interrupt wake-up costs tens of microseconds
    (void) hipSetDeviceFlags(hipDeviceScheduleSpin);
    (void) hipGetLastError();
    whisper_context * ctx = new whisper_context;
    ctx->params = params;
    ctx->device = params.gpu_device;
    ctx->exact = !params.flash_attn;   // flash_attn trades the reference's summation order for MFMA speed, as it does in the reference
    bool ok = false;
    try { ok = wa_model_load(loader, *ctx); } catch (const std::exception & e) { WA_ERROR("%s: exception: %s\n", __func__, e.what()); ok = false; }
    loader->close(loader->context);
    if (!ok) {
        WA_ERROR("%s: failed to load model\n", __func__);
        wa_model_free(*ctx);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

whisper_context * with_state(whisper_context * ctx) {
    if (!ctx) return nullptr;
    ctx->state = whisper_init_state(ctx);
    if (!ctx->state) { whisper_free(ctx); return nullptr; }
    return ctx;
}
} // namespace

extern "C" {

struct whisper_context_params whisper_context_default_params(void) {     // whisper.cpp:3622-3638
    whisper_context_params r;
    memset(&r, 0, sizeof(r));
    r.use_gpu = true; r.flash_attn = false; r.gpu_device = 0;
    r.dtw_token_timestamps = false; r.dtw_aheads_preset = WHISPER_AHEADS_NONE; r.dtw_n_top = -1;
    r.dtw_aheads.n_heads = 0; r.dtw_aheads.heads = nullptr;
    r.dtw_mem_size = 1024 * 1024 * 128;
    return r;
}
struct whisper_context_params * whisper_context_default_params_by_ref(void) {
    auto * p = new whisper_context_params(); *p = whisper_context_default_params(); return p;
}

struct whisper_context * whisper_init_with_params_no_state(struct whisper_model_loader * loader, struct whisper_context_params params) {
    if (!loader) return nullptr;
    return init_common(loader, params);
}

struct whisper_context * whisper_init_from_file_with_params_no_state(const char * path_model, struct whisper_context_params params) {
    WA_INFO("%s: loading model from '%s'\n", __func__, path_model ? path_model : "(null)");
    file_src src{ path_model ? fopen(path_model, "rb") : nullptr, false };
    if (!src.f) { WA_ERROR("%s: failed to open '%s'\n", __func__, path_model ? path_model : "(null)"); return nullptr; }
    setvbuf(src.f, nullptr, _IOFBF, 1 << 22);
    whisper_model_loader loader;
    loader.context = &src;
    loader.read  = [](void * c, void * out, size_t n) { auto * s = (file_src *) c; size_t r = fread(out, 1, n, s->f); if (r < n) s->eof = true; return r; };
    loader.eof   = [](void * c) { return ((file_src *) c)->eof; };
    loader.close = [](void * c) { auto * s = (file_src *) c; if (s->f) { fclose(s->f); s->f = nullptr; } };
    whisper_context * ctx = init_common(&loader, params);
    if (ctx) ctx->path_model = path_model;
    return ctx;
}

struct whisper_context * whisper_init_from_buffer_with_params_no_state(void * buffer, size_t buffer_size, struct whisper_context_params params) {
    WA_INFO("%s: loading model from buffer\n", __func__);
    if (!buffer) return nullptr;
    buf_src src{ (const uint8_t *) buffer, buffer_size, 0 };
    whisper_model_loader loader;
    loader.context = &src;
    loader.read = [](void * c, void * out, size_t n) {
        auto * s = (buf_src *) c;
        const size_t k = s->off + n < s->size ? n : s->size - s->off;     // whisper.cpp:3702
        memcpy(out, s->p + s->off, k);
        s->off += k;
        return k;
    };
    loader.eof   = [](void * c) { auto * s = (buf_src *) c; return s->off >= s->size; };
    loader.close = [](void *) {};
    return init_common(&loader, params);
}

struct whisper_context * whisper_init_from_file_with_params(const char * p, struct whisper_context_params params) { return with_state(whisper_init_from_file_with_params_no_state(p, params)); }
struct whisper_context * whisper_init_from_buffer_with_params(void * b, size_t n, struct whisper_context_params params) { return with_state(whisper_init_from_buffer_with_params_no_state(b, n, params)); }
struct whisper_context * whisper_init_with_params(struct whisper_model_loader * l, struct whisper_context_params params) { return with_state(whisper_init_with_params_no_state(l, params)); }
struct whisper_context * whisper_init_from_file(const char * p) { return whisper_init_from_file_with_params(p, whisper_context_default_params()); }
struct whisper_context * whisper_init_from_buffer(void * b, size_t n) { return whisper_init_from_buffer_with_params(b, n, whisper_context_default_params()); }
struct whisper_context * whisper_init(struct whisper_model_loader * l) { return whisper_init_with_params(l, whisper_context_default_params()); }
struct whisper_context * whisper_init_from_file_no_state(const char * p) { return whisper_init_from_file_with_params_no_state(p, whisper_context_default_params()); }
struct whisper_context * whisper_init_from_buffer_no_state(void * b, size_t n) { return whisper_init_from_buffer_with_params_no_state(b, n, whisper_context_default_params()); }
struct whisper_context * whisper_init_no_state(struct whisper_model_loader * l) { return whisper_init_with_params_no_state(l, whisper_context_default_params()); }

// alignment heads per text layer (ref: whisper.cpp:1190-1250, get_alignment_heads_by_layer)
static const whisper_ahead k_ah_tiny_en[]   = { {1,0},{2,0},{2,5},{3,0},{3,1},{3,2},{3,3},{3,4} };
static const whisper_ahead k_ah_tiny[]      = { {2,2},{3,0},{3,2},{3,3},{3,4},{3,5} };
static const whisper_ahead k_ah_base_en[]   = { {3,3},{4,7},{5,1},{5,5},{5,7} };
static const whisper_ahead k_ah_base[]      = { {3,1},{4,2},{4,3},{4,7},{5,1},{5,2},{5,4},{5,6} };
static const whisper_ahead k_ah_small_en[]  = { {6,6},{7,0},{7,3},{7,8},{8,2},{8,5},{8,7},{9,0},{9,4},{9,8},{9,10},{10,0},{10,1},{10,2},{10,3},{10,6},{10,11},{11,2},{11,4} };
static const whisper_ahead k_ah_small[]     = { {5,3},{5,9},{8,0},{8,4},{8,7},{8,8},{9,0},{9,7},{9,9},{10,5} };
static const whisper_ahead k_ah_medium_en[] = { {11,4},{14,1},{14,12},{14,14},{15,4},{16,0},{16,4},{16,9},{17,12},{17,14},{18,7},{18,10},{18,15},{20,0},{20,3},{20,9},{20,14},{21,12} };
static const whisper_ahead k_ah_medium[]    = { {13,15},{15,4},{15,15},{16,1},{20,0},{23,4} };
static const whisper_ahead k_ah_large_v1[]  = { {9,19},{11,2},{11,4},{11,17},{22,7},{22,11},{22,17},{23,2},{23,15} };
static const whisper_ahead k_ah_large_v2[]  = { {10,12},{13,17},{16,11},{16,12},{16,13},{17,15},{17,16},{18,4},{18,11},{18,19},{19,11},{21,2},{21,3},{22,3},{22,9},{22,12},{23,5},{23,7},{23,13},{25,5},{26,1},{26,12},{27,15} };
static const whisper_ahead k_ah_large_v3[]  = { {7,0},{10,17},{12,18},{13,12},{16,1},{17,14},{19,11},{21,4},{24,1},{25,6} };
static const whisper_ahead k_ah_large_v3t[] = { {2,4},{2,11},{3,3},{3,6},{3,11},{3,14} };

static bool aheads_init(const whisper_context & ctx, whisper_state & st) {
    const auto & p = ctx.params;
    const auto & hp = ctx.model.hp;
    st.aheads.assign(hp.n_text_layer, {});
    st.aheads_n = 0;
    if (!p.dtw_token_timestamps) return true;
    whisper_aheads set{ 0, nullptr };
#define WA_AH(x) set = whisper_aheads{ sizeof(x) / sizeof(x[0]), x }
    switch (p.dtw_aheads_preset) {
        case WHISPER_AHEADS_NONE: WA_ERROR("%s: dtw_token_timestamps requires an alignment heads preset\n", __func__); return false;
        case WHISPER_AHEADS_N_TOP_MOST:
            if (p.dtw_n_top > hp.n_text_layer || p.dtw_n_top <= 0) { WA_ERROR("%s: dtw_n_top must be between %d and %d for this model\n", __func__, 1, hp.n_text_layer); return false; }
            break;
        case WHISPER_AHEADS_CUSTOM:
            if (p.dtw_aheads.n_heads == 0) { WA_ERROR("%s: dtw_aheads.n_heads should be > 0\n", __func__); return false; }
            if (p.dtw_aheads.heads == nullptr) { WA_ERROR("%s: dtw_aheads.heads unset\n", __func__); return false; }
            set = p.dtw_aheads;
            break;
        case WHISPER_AHEADS_TINY_EN:   WA_AH(k_ah_tiny_en); break;
        case WHISPER_AHEADS_TINY:      WA_AH(k_ah_tiny); break;
        case WHISPER_AHEADS_BASE_EN:   WA_AH(k_ah_base_en); break;
        case WHISPER_AHEADS_BASE:      WA_AH(k_ah_base); break;
        case WHISPER_AHEADS_SMALL_EN:  WA_AH(k_ah_small_en); break;
        case WHISPER_AHEADS_SMALL:     WA_AH(k_ah_small); break;
        case WHISPER_AHEADS_MEDIUM_EN: WA_AH(k_ah_medium_en); break;
        case WHISPER_AHEADS_MEDIUM:    WA_AH(k_ah_medium); break;
        case WHISPER_AHEADS_LARGE_V1:  WA_AH(k_ah_large_v1); break;
        case WHISPER_AHEADS_LARGE_V2:  WA_AH(k_ah_large_v2); break;
        case WHISPER_AHEADS_LARGE_V3:  WA_AH(k_ah_large_v3); break;
        case WHISPER_AHEADS_LARGE_V3_TURBO: WA_AH(k_ah_large_v3t); break;
    }
#undef WA_AH
    for (int il = 0; il < hp.n_text_layer; ++il) {
        if (p.dtw_aheads_preset == WHISPER_AHEADS_N_TOP_MOST) {
            if (il >= hp.n_text_layer - p.dtw_n_top) for (int h = 0; h < hp.n_text_head; ++h) st.aheads[il].push_back(h);
        } else {
            for (size_t i = 0; i < set.n_heads; ++i) {
                if (set.heads[i].n_text_layer == il) {
                    if (set.heads[i].n_head >= hp.n_text_head) {
                        WA_ERROR("%s: selected alignment heads are not compatible with this model\n", __func__);
                        return false;
                    }
                    st.aheads[il].push_back(set.heads[i].n_head);
                }
            }
        }
        st.aheads_n += (int) st.aheads[il].size();
    }
    if (p.dtw_aheads_preset != WHISPER_AHEADS_N_TOP_MOST)
        for (size_t i = 0; i < set.n_heads; ++i)
            if (set.heads[i].n_text_layer >= hp.n_text_layer || set.heads[i].n_text_layer < 0) {
                WA_ERROR("%s: selected alignment heads are not compatible with this model\n", __func__);
                return false;
            }
    return true;
}

struct whisper_state * whisper_init_state(struct whisper_context * ctx) {
    if (!ctx) return nullptr;
    whisper_state * st = new whisper_state;
    bool ok = false;
    try { ok = wa_state_alloc(*ctx, *st) && aheads_init(*ctx, *st); } catch (...) { ok = false; }
    if (!ok) { WA_ERROR("%s: failed to allocate state\n", __func__); whisper_free_state(st); return nullptr; }
    return st;
}

void whisper_free_state(struct whisper_state * st) { if (st) { wa_state_release(*st); delete st; } }
void whisper_free(struct whisper_context * ctx) {
    if (!ctx) return;
    wa_batcher_free_all(*ctx);
    whisper_free_state(ctx->state);
    wa_model_free(*ctx);
    delete ctx;
}
void whisper_free_context_params(struct whisper_context_params * p) { delete p; }
void whisper_free_params(struct whisper_full_params * p) { delete p; }

int whisper_ctx_init_openvino_encoder_with_state(struct whisper_context *, struct whisper_state *, const char *, const char *, const char *) { return 1; }
int whisper_ctx_init_openvino_encoder(struct whisper_context *, const char *, const char *, const char *) { return 1; }

// -------------------------------------------------------------------------------------------------
// stage API (ref: whisper.cpp:3891-3971)
// -------------------------------------------------------------------------------------------------
int whisper_pcm_to_mel_with_state(struct whisper_context * ctx, struct whisper_state * st, const float * samples, int n_samples, int) {
    if (!ctx || !st || !wa_mel_compute(*ctx, *st, samples, n_samples)) { WA_ERROR("%s: failed to compute mel spectrogram\n", __func__); return -1; }
    return 0;
}
int whisper_pcm_to_mel(struct whisper_context * ctx, const float * samples, int n_samples, int n_threads) {
    return whisper_pcm_to_mel_with_state(ctx, ctx ? ctx->state : nullptr, samples, n_samples, n_threads);
}
int whisper_set_mel_with_state(struct whisper_context * ctx, struct whisper_state * st, const float * data, int n_len, int n_mel) {
    if (!ctx || !st || !wa_mel_set(*ctx, *st, data, n_len, n_mel)) return -1;
    return 0;
}
int whisper_set_mel(struct whisper_context * ctx, const float * data, int n_len, int n_mel) {
    return whisper_set_mel_with_state(ctx, ctx ? ctx->state : nullptr, data, n_len, n_mel);
}
int whisper_encode_with_state(struct whisper_context * ctx, struct whisper_state * st, int offset, int) {
    if (!ctx || !st || !wa_encode(*ctx, *st, offset, nullptr, nullptr)) { WA_ERROR("%s: failed to eval\n", __func__); return -1; }
    return 0;
}
int whisper_encode(struct whisper_context * ctx, int offset, int n_threads) { return whisper_encode_with_state(ctx, ctx ? ctx->state : nullptr, offset, n_threads); }

int whisper_decode_with_state(struct whisper_context * ctx, struct whisper_state * st, const whisper_token * tokens, int n_tokens, int n_past, int) {
    if (!ctx || !st || n_tokens <= 0) return 1;
    auto & b = st->batch;                                   // whisper_batch_prep_legacy, whisper.cpp:544-556
    b.n_tokens = n_tokens;
    b.token.assign(tokens, tokens + n_tokens);
    b.pos.resize(n_tokens); b.seq_id.assign(n_tokens, 0); b.logits.assign(n_tokens, 0);
    for (int i = 0; i < n_tokens; ++i) b.pos[i] = n_past + i;
    b.logits[n_tokens - 1] = 1;
    wa_kv_seq_rm(st->kv_self, 0, n_past, -1);
    if (!wa_decode(*ctx, *st, b, false, nullptr, nullptr)) { WA_ERROR("%s: failed to eval\n", __func__); return 1; }
    return 0;
}
int whisper_decode(struct whisper_context * ctx, const whisper_token * tokens, int n_tokens, int n_past, int n_threads) {
    if (!ctx || !ctx->state) { WA_ERROR("%s: ERROR state was not loaded.\n", __func__); return -1; }
    return whisper_decode_with_state(ctx, ctx->state, tokens, n_tokens, n_past, n_threads);
}
float * whisper_get_logits_from_state(struct whisper_state * st) { return st->logits.data(); }
float * whisper_get_logits(struct whisper_context * ctx) { return ctx->state->logits.data(); }

// -------------------------------------------------------------------------------------------------
// tokenizer / languages (ref: whisper.cpp:3973-4110)
// -------------------------------------------------------------------------------------------------
int whisper_tokenize(struct whisper_context * ctx, const char * text, whisper_token * tokens, int n_max_tokens) {
    std::vector<int> res;
    try { res = wa_tokenize(ctx->vocab, text); } catch (...) { return -1; }
    if (n_max_tokens < (int) res.size()) {
        WA_ERROR("%s: too many resulting tokens: %d (max %d)\n", __func__, (int) res.size(), n_max_tokens);
        return -(int) res.size();
    }
    for (size_t i = 0; i < res.size(); ++i) tokens[i] = res[i];
    return (int) res.size();
}
int whisper_token_count(struct whisper_context * ctx, const char * text) { return -whisper_tokenize(ctx, text, nullptr, 0); }

int whisper_lang_max_id(void) { return WA_N_LANG - 1; }
int whisper_lang_id(const char * lang) {
    if (lang) {
        for (int i = 0; i < WA_N_LANG; ++i) if (strcmp(k_lang[i][0], lang) == 0) return i;
        for (int i = 0; i < WA_N_LANG; ++i) if (strcmp(k_lang[i][1], lang) == 0) return i;
    }
    WA_ERROR("%s: unknown language '%s'\n", __func__, lang ? lang : "(null)");
    return -1;
}
const char * whisper_lang_str(int id) {
    if (id >= 0 && id < WA_N_LANG) return k_lang[id][0];
    WA_ERROR("%s: unknown language id %d\n", __func__, id);
    return nullptr;
}
const char * whisper_lang_str_full(int id) {
    if (id >= 0 && id < WA_N_LANG) return k_lang[id][1];
    WA_ERROR("%s: unknown language id %d\n", __func__, id);
    return nullptr;
}

int whisper_lang_auto_detect_with_state(struct whisper_context * ctx, struct whisper_state * st, int offset_ms, int n_threads, float * lang_probs) {
    const int seek = offset_ms / 10;
    if (seek < 0) { WA_ERROR("%s: offset %dms is before the start of the audio\n", __func__, offset_ms); return -1; }
    if (seek >= st->mel_n_len_org) { WA_ERROR("%s: offset %dms is past the end of the audio (%dms)\n", __func__, offset_ms, st->mel_n_len_org * 10); return -2; }
    if (whisper_encode_with_state(ctx, st, seek, n_threads) != 0) { WA_ERROR("%s: failed to encode\n", __func__); return -6; }
    const whisper_token prompt[1] = { whisper_token_sot(ctx) };
    if (whisper_decode_with_state(ctx, st, prompt, 1, 0, n_threads) != 0) { WA_ERROR("%s: failed to decode\n", __func__); return -7; }

    // softmax over the language-token logits, in double, sorted descending (whisper.cpp:4068-4109)
    std::vector<std::pair<double, int>> lp;
    for (int i = 0; i < WA_N_LANG; ++i) {
        const int tok = whisper_token_lang(ctx, i);
        if (tok >= ctx->vocab.n_vocab) continue;
        lp.emplace_back((double) st->logits[tok], i);
    }
    if (lp.empty()) return -7;
    std::stable_sort(lp.begin(), lp.end(), [](const std::pair<double, int> & a, const std::pair<double, int> & b) { return a.first > b.first; });
    const double mx = lp[0].first;
    double sum = 0.0;
    for (auto & kv : lp) { kv.first = exp(kv.first - mx); sum += kv.first; }
    for (auto & kv : lp) kv.first /= sum;
    if (lang_probs) for (auto & kv : lp) lang_probs[kv.second] = (float) kv.first;
    return lp[0].second;
}
int whisper_lang_auto_detect(struct whisper_context * ctx, int offset_ms, int n_threads, float * lang_probs) {
    return whisper_lang_auto_detect_with_state(ctx, ctx->state, offset_ms, n_threads, lang_probs);
}

// -------------------------------------------------------------------------------------------------
// introspection (ref: whisper.cpp:4120-4259)
// -------------------------------------------------------------------------------------------------
int whisper_n_len_from_state(struct whisper_state * st) { return st->mel_n_len_org; }
int whisper_n_len(struct whisper_context * ctx) { return ctx->state->mel_n_len_org; }
int whisper_n_vocab(struct whisper_context * ctx) { return ctx->vocab.n_vocab; }
int whisper_n_text_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_text_ctx; }
int whisper_n_audio_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_audio_ctx; }
int whisper_is_multilingual(struct whisper_context * ctx) { return ctx->vocab.is_multilingual() ? 1 : 0; }
int whisper_model_n_vocab(struct whisper_context * ctx) { return ctx->model.hp.n_vocab; }
int whisper_model_n_audio_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_audio_ctx; }
int whisper_model_n_audio_state(struct whisper_context * ctx) { return ctx->model.hp.n_audio_state; }
int whisper_model_n_audio_head(struct whisper_context * ctx) { return ctx->model.hp.n_audio_head; }
int whisper_model_n_audio_layer(struct whisper_context * ctx) { return ctx->model.hp.n_audio_layer; }
int whisper_model_n_text_ctx(struct whisper_context * ctx) { return ctx->model.hp.n_text_ctx; }
int whisper_model_n_text_state(struct whisper_context * ctx) { return ctx->model.hp.n_text_state; }
int whisper_model_n_text_head(struct whisper_context * ctx) { return ctx->model.hp.n_text_head; }
int whisper_model_n_text_layer(struct whisper_context * ctx) { return ctx->model.hp.n_text_layer; }
int whisper_model_n_mels(struct whisper_context * ctx) { return ctx->model.hp.n_mels; }
int whisper_model_ftype(struct whisper_context * ctx) { return ctx->model.hp.ftype; }
int whisper_model_type(struct whisper_context * ctx) { return ctx->model.type; }
const char * whisper_model_type_readable(struct whisper_context * ctx) {
    switch (ctx->model.type) { case 1: return "tiny"; case 2: return "base"; case 3: return "small"; case 4: return "medium"; case 5: return "large"; default: return "unknown"; }
}
const char * whisper_token_to_str(struct whisper_context * ctx, whisper_token token) {
    // the reference throws std::out_of_range across the C boundary here (map::at); we return "" instead
    if (token < 0 || token >= (int) ctx->vocab.id_to_token.size()) return "";
    return ctx->vocab.id_to_token[token].c_str();
}
whisper_token whisper_token_eot (struct whisper_context * ctx) { return ctx->vocab.token_eot; }
whisper_token whisper_token_sot (struct whisper_context * ctx) { return ctx->vocab.token_sot; }
whisper_token whisper_token_solm(struct whisper_context * ctx) { return ctx->vocab.token_solm; }
whisper_token whisper_token_prev(struct whisper_context * ctx) { return ctx->vocab.token_prev; }
whisper_token whisper_token_nosp(struct whisper_context * ctx) { return ctx->vocab.token_nosp; }
whisper_token whisper_token_not (struct whisper_context * ctx) { return ctx->vocab.token_not; }
whisper_token whisper_token_beg (struct whisper_context * ctx) { return ctx->vocab.token_beg; }
whisper_token whisper_token_lang(struct whisper_context * ctx, int lang_id) { return ctx->vocab.token_sot + 1 + lang_id; }
whisper_token whisper_token_translate (struct whisper_context * ctx) { return ctx->vocab.token_translate; }
whisper_token whisper_token_transcribe(struct whisper_context * ctx) { return ctx->vocab.token_transcribe; }

// -------------------------------------------------------------------------------------------------
// timings / system info (ref: whisper.cpp:4261-4355)
// -------------------------------------------------------------------------------------------------
struct whisper_timings * whisper_get_timings(struct whisper_context * ctx) {
    if (!ctx->state) return nullptr;
    auto * s = ctx->state;
    auto * t = new whisper_timings;
    t->sample_ms = 1e-3f * s->t_sample_us / std::max(1, s->n_sample);
    t->encode_ms = 1e-3f * s->t_encode_us / std::max(1, s->n_encode);
    t->decode_ms = 1e-3f * s->t_decode_us / std::max(1, s->n_decode);
    t->batchd_ms = 1e-3f * s->t_batchd_us / std::max(1, s->n_batchd);
    t->prompt_ms = 1e-3f * s->t_prompt_us / std::max(1, s->n_prompt);
    return t;
}
void whisper_print_timings(struct whisper_context * ctx) {
    const int64_t t_end = wa_time_us();
    WA_INFO("\n");
    WA_INFO("%s:     load time = %8.2f ms\n", __func__, ctx->t_load_us / 1000.0f);
    if (auto * s = ctx->state) {
        const int n_sample = std::max(1, s->n_sample), n_encode = std::max(1, s->n_encode), n_decode = std::max(1, s->n_decode),
                  n_batchd = std::max(1, s->n_batchd), n_prompt = std::max(1, s->n_prompt);
        WA_INFO("%s:     fallbacks = %3d p / %3d h\n", __func__, s->n_fail_p, s->n_fail_h);
        WA_INFO("%s:      mel time = %8.2f ms\n", __func__, s->t_mel_us / 1000.0f);
        WA_INFO("%s:   sample time = %8.2f ms / %5d runs ( %8.2f ms per run)\n", __func__, 1e-3f * s->t_sample_us, n_sample, 1e-3f * s->t_sample_us / n_sample);
        WA_INFO("%s:   encode time = %8.2f ms / %5d runs ( %8.2f ms per run)\n", __func__, 1e-3f * s->t_encode_us, n_encode, 1e-3f * s->t_encode_us / n_encode);
        WA_INFO("%s:   decode time = %8.2f ms / %5d runs ( %8.2f ms per run)\n", __func__, 1e-3f * s->t_decode_us, n_decode, 1e-3f * s->t_decode_us / n_decode);
        WA_INFO("%s:   batchd time = %8.2f ms / %5d runs ( %8.2f ms per run)\n", __func__, 1e-3f * s->t_batchd_us, n_batchd, 1e-3f * s->t_batchd_us / n_batchd);
        WA_INFO("%s:   prompt time = %8.2f ms / %5d runs ( %8.2f ms per run)\n", __func__, 1e-3f * s->t_prompt_us, n_prompt, 1e-3f * s->t_prompt_us / n_prompt);
    }
    WA_INFO("%s:    total time = %8.2f ms\n", __func__, (t_end - ctx->t_start_us) / 1000.0f);
}
void whisper_amd_reset_timings(struct whisper_state * s) {
    s->t_mel_us = s->t_sample_us = s->t_encode_us = s->t_decode_us = s->t_batchd_us = s->t_prompt_us = 0;
    s->n_sample = s->n_encode = s->n_decode = s->n_batchd = s->n_prompt = 0;
    s->n_spec_ok = s->n_spec_miss = 0;
}
void whisper_amd_overlap_stats(struct whisper_state * s, int out[2]) { out[0] = s->n_spec_ok; out[1] = s->n_spec_miss; }
void whisper_reset_timings(struct whisper_context * ctx) {
    ctx->t_start_us = wa_time_us();
    if (ctx->state) whisper_amd_reset_timings(ctx->state);
}
void whisper_amd_get_timings_us(struct whisper_state * s, int64_t out[12]) {
    out[0] = s->t_sample_us; out[1] = s->t_encode_us; out[2] = s->t_decode_us; out[3] = s->t_batchd_us; out[4] = s->t_prompt_us; out[5] = s->t_mel_us;
    out[6] = s->n_sample; out[7] = s->n_encode; out[8] = s->n_decode; out[9] = s->n_batchd; out[10] = s->n_prompt; out[11] = s->n_fail_p + s->n_fail_h;
}
const char * whisper_print_system_info(void) {
    static std::string s;
    s = "WHISPER : COREML = 0 | OPENVINO = 0 | HIP : ARCH = gfx950 | MFMA = 1 | WAVE = 64 | ";
    return s.c_str();
}

// -------------------------------------------------------------------------------------------------
// default full params (ref: whisper.cpp:5914-6019)
// -------------------------------------------------------------------------------------------------
struct whisper_vad_params whisper_vad_default_params(void) {    // whisper.cpp:4380-4390
    whisper_vad_params r;
    r.threshold = 0.5f; r.min_speech_duration_ms = 250; r.min_silence_duration_ms = 100;
    r.max_speech_duration_s = 3.4028234663852886e+38f; r.speech_pad_ms = 30; r.samples_overlap = 0.1f;
    return r;
}

struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy) {
    whisper_full_params r;
    memset(&r, 0, sizeof(r));
    r.strategy = strategy;
    r.n_threads = std::min(4, (int) std::thread::hardware_concurrency());
    r.n_max_text_ctx = 16384;
    r.no_context = true;
    r.print_progress = true;
    r.print_timestamps = true;
    r.thold_pt = 0.01f; r.thold_ptsum = 0.01f;
    r.language = "en";
    r.suppress_blank = true;
    r.temperature = 0.0f; r.max_initial_ts = 1.0f; r.length_penalty = -1.0f;
    r.temperature_inc = 0.2f; r.entropy_thold = 2.4f; r.logprob_thold = -1.0f; r.no_speech_thold = 0.6f;
    r.greedy.best_of = -1;
    r.beam_search.beam_size = -1; r.beam_search.patience = -1.0f;
    r.grammar_penalty = 100.0f;
    r.vad_params = whisper_vad_default_params();
    if (strategy == WHISPER_SAMPLING_GREEDY) r.greedy.best_of = 5;
    if (strategy == WHISPER_SAMPLING_BEAM_SEARCH) { r.beam_search.beam_size = 5; r.beam_search.patience = -1.0f; }
    return r;
}
struct whisper_full_params * whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy) {
    auto * p = new whisper_full_params(); *p = whisper_full_default_params(strategy); return p;
}

// -------------------------------------------------------------------------------------------------
// full pipeline entry points
// -------------------------------------------------------------------------------------------------
int whisper_full_with_state(struct whisper_context * ctx, struct whisper_state * st, struct whisper_full_params params, const float * samples, int n_samples) {
    if (!ctx || !st) return -1;
    try { return wa_full(ctx, st, params, samples, n_samples); }
    catch (const std::exception & e) { WA_ERROR("%s: exception: %s\n", __func__, e.what()); return -1; }
    catch (...) { WA_ERROR("%s: unknown exception\n", __func__); return -1; }
}
int whisper_full(struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples) {
    if (!ctx || !ctx->state) return -1;
    if (params.vad) { WA_ERROR("%s: VAD is not supported by this backend\n", __func__); return -1; }
    return whisper_full_with_state(ctx, ctx->state, params, samples, n_samples);
}
// Lock-step decode groups for chunks transcribed together on one device (whisper_amd_full_batch, whisper_full_parallel): groups of
// WHISPER_AMD_BATCH_GROUP chunks share a decoder pass (wa_decode.cpp: wa_batcher).  Where the pass is ONE launch (wa_rows.hip; it owns the
// device while it runs) the default is one group of up to 8 - every weight row read once for eight tokens.  Where only the launch
// sequence is available (a pass is then a chain of short latency-bound launches) groups of 4 run side by side on their own streams:
// two 4-row passes finish sooner than one 8-row pass after the other (round 2: 8 chunks 488x real time as one group, 589x as two).
struct wa_batch_groups {
    whisper_context * ctx;
    std::vector<whisper_state *> members;
    std::vector<wa_batcher *> bats;
    wa_batch_groups(whisper_context * c, const std::vector<whisper_state *> & m) : ctx(c), members(m) {
        static const bool off = getenv("WHISPER_AMD_NO_BATCHER") != nullptr;
        int group = !members.empty() && members[0]->rows_enabled ? WA_MAX_DECODERS : 4;
        if (const char * g = getenv("WHISPER_AMD_BATCH_GROUP")) group = std::max(2, std::min(WA_MAX_DECODERS, atoi(g)));
        for (size_t i0 = 0; i0 < members.size() && !off; i0 += group) {
            const size_t n = std::min((size_t) group, members.size() - i0);
            wa_batcher * b = wa_batcher_create(*ctx, (int) n);       // null for a group of one / a quantised model: those chunks decode on their own
            bats.push_back(b);
            for (size_t i = i0; i < i0 + n; ++i) members[i]->batcher = b;
        }
    }
    ~wa_batch_groups() {
        for (auto * st : members) st->batcher = nullptr;
        ctx->batch_steps = ctx->batch_rows = ctx->batch_one_launch = 0;
        for (auto * b : bats) if (b) { long st_ = 0, rw_ = 0, ol_ = 0; wa_batcher_stats(b, &st_, &rw_, &ol_); ctx->batch_steps += st_; ctx->batch_rows += rw_; ctx->batch_one_launch += ol_; wa_batcher_destroy(b); }
    }
};

int whisper_full_parallel(struct whisper_context * ctx, struct whisper_full_params params, const float * samples, int n_samples, int n_processors) {
    // ref: whisper.cpp:7736-7864: the audio is cut into n_processors equal parts, each transcribed on a state of its own (here:
    // concurrently on this device, one HIP stream each), the segment lists appended with their time offsets.  As in the reference
    // the transcription may be degraded near the cuts.
    if (!ctx || !ctx->state) return -1;
    if (n_processors <= 1) return whisper_full(ctx, params, samples, n_samples);
    if (params.vad) { WA_ERROR("%s: VAD is not supported by this backend\n", __func__); return -1; }
    const int offset_samples = (WHISPER_SAMPLE_RATE * params.offset_ms) / 1000;
    const int n_per = (n_samples - offset_samples) / n_processors;
    std::vector<whisper_state *> states;
    std::vector<std::thread> workers;
    std::vector<int> rcs(n_processors, 0);
    for (int i = 0; i < n_processors - 1; ++i) {
        whisper_state * st = whisper_init_state(ctx);
        if (!st) { for (auto * s2 : states) whisper_free_state(s2); return -1; }
        states.push_back(st);
    }
    std::vector<whisper_state *> members = { ctx->state };
    members.insert(members.end(), states.begin(), states.end());
    {
    wa_batch_groups groups(ctx, members);           // the parts' single-token steps in lock step (as whisper_amd_full_batch); ends with the joins
    for (int i = 0; i < n_processors - 1; ++i) {
        whisper_state * st = states[i];
        const int start = offset_samples + (i + 1) * n_per;
        const int n_cur = (i == n_processors - 2) ? n_samples - start : n_per;
        whisper_full_params pc = params;
        pc.offset_ms = 0; pc.print_progress = false; pc.print_realtime = false;
        pc.new_segment_callback = nullptr; pc.new_segment_callback_user_data = nullptr;
        pc.progress_callback = nullptr; pc.progress_callback_user_data = nullptr;
        workers.emplace_back([=, &rcs]() { rcs[i + 1] = whisper_full_with_state(ctx, st, pc, samples + start, n_cur); wa_batcher_leave(st->batcher); });
    }
    {
        whisper_full_params pc = params;
        pc.print_realtime = false;
        rcs[0] = whisper_full_with_state(ctx, ctx->state, pc, samples, offset_samples + n_per);
        wa_batcher_leave(ctx->state->batcher);
    }
    for (auto & w : workers) w.join();
    }
    const int64_t offset_t = (int64_t) params.offset_ms / 10.0;
    for (int i = 0; i < n_processors - 1; ++i) {
        for (auto & r : states[i]->result_all) {
            r.t0 += 100 * ((i + 1) * n_per) / WHISPER_SAMPLE_RATE + offset_t;
            r.t1 += 100 * ((i + 1) * n_per) / WHISPER_SAMPLE_RATE + offset_t;
            if (!ctx->state->result_all.empty()) r.t0 = std::max(r.t0, ctx->state->result_all.back().t1);    // no overlapping segments
            ctx->state->result_all.push_back(std::move(r));
            if (params.new_segment_callback) params.new_segment_callback(ctx, ctx->state, 1, params.new_segment_callback_user_data);
        }
        ctx->state->t_mel_us += states[i]->t_mel_us; ctx->state->t_sample_us += states[i]->t_sample_us; ctx->state->t_encode_us += states[i]->t_encode_us;
        ctx->state->t_decode_us += states[i]->t_decode_us; ctx->state->t_batchd_us += states[i]->t_batchd_us; ctx->state->t_prompt_us += states[i]->t_prompt_us;
        ctx->state->n_sample += states[i]->n_sample; ctx->state->n_encode += states[i]->n_encode; ctx->state->n_decode += states[i]->n_decode;
        ctx->state->n_batchd += states[i]->n_batchd; ctx->state->n_prompt += states[i]->n_prompt;
        whisper_free_state(states[i]);
    }
    ctx->state->t_mel_us /= n_processors; ctx->state->t_sample_us /= n_processors; ctx->state->t_encode_us /= n_processors; ctx->state->t_decode_us /= n_processors;
    WA_WARN("%s: the audio has been split into %d chunks; the transcription quality may be degraded near the boundaries\n", __func__, n_processors);
    return rcs[0];
}

// -------------------------------------------------------------------------------------------------
// result getters (ref: whisper.cpp:7866-8033); VAD time remapping does not apply (no VAD here)
// -------------------------------------------------------------------------------------------------
int whisper_full_n_segments_from_state(struct whisper_state * st) { return (int) st->result_all.size(); }
int whisper_full_n_segments(struct whisper_context * ctx) { return (int) ctx->state->result_all.size(); }
int whisper_full_lang_id_from_state(struct whisper_state * st) { return st->lang_id; }
int whisper_full_lang_id(struct whisper_context * ctx) { return ctx->state->lang_id; }
int64_t whisper_full_get_segment_t0_from_state(struct whisper_state * st, int i) { return st->result_all[i].t0; }
int64_t whisper_full_get_segment_t0(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].t0; }
int64_t whisper_full_get_segment_t1_from_state(struct whisper_state * st, int i) { return st->result_all[i].t1; }
int64_t whisper_full_get_segment_t1(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].t1; }
bool whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state * st, int i) { return st->result_all[i].speaker_turn_next; }
bool whisper_full_get_segment_speaker_turn_next(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].speaker_turn_next; }
const char * whisper_full_get_segment_text_from_state(struct whisper_state * st, int i) { return st->result_all[i].text.c_str(); }
const char * whisper_full_get_segment_text(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].text.c_str(); }
int whisper_full_n_tokens_from_state(struct whisper_state * st, int i) { return (int) st->result_all[i].tokens.size(); }
int whisper_full_n_tokens(struct whisper_context * ctx, int i) { return (int) ctx->state->result_all[i].tokens.size(); }
const char * whisper_full_get_token_text_from_state(struct whisper_context * ctx, struct whisper_state * st, int i, int j) {
    return whisper_token_to_str(ctx, st->result_all[i].tokens[j].id);
}
const char * whisper_full_get_token_text(struct whisper_context * ctx, int i, int j) { return whisper_full_get_token_text_from_state(ctx, ctx->state, i, j); }
whisper_token whisper_full_get_token_id_from_state(struct whisper_state * st, int i, int j) { return st->result_all[i].tokens[j].id; }
whisper_token whisper_full_get_token_id(struct whisper_context * ctx, int i, int j) { return ctx->state->result_all[i].tokens[j].id; }
whisper_token_data whisper_full_get_token_data_from_state(struct whisper_state * st, int i, int j) { return st->result_all[i].tokens[j]; }
whisper_token_data whisper_full_get_token_data(struct whisper_context * ctx, int i, int j) { return ctx->state->result_all[i].tokens[j]; }
float whisper_full_get_token_p_from_state(struct whisper_state * st, int i, int j) { return st->result_all[i].tokens[j].p; }
float whisper_full_get_token_p(struct whisper_context * ctx, int i, int j) { return ctx->state->result_all[i].tokens[j].p; }
float whisper_full_get_segment_no_speech_prob_from_state(struct whisper_state * st, int i) { return st->result_all[i].no_speech_prob; }
float whisper_full_get_segment_no_speech_prob(struct whisper_context * ctx, int i) { return ctx->state->result_all[i].no_speech_prob; }

// -------------------------------------------------------------------------------------------------
// VAD / bench: link-completeness stubs (out of scope)
// -------------------------------------------------------------------------------------------------
struct whisper_vad_context_params whisper_vad_default_context_params(void) { whisper_vad_context_params r; r.n_threads = 4; r.use_gpu = false; r.gpu_device = 0; return r; }
struct whisper_vad_context * whisper_vad_init_from_file_with_params(const char *, struct whisper_vad_context_params) { WA_ERROR("VAD is not supported by this backend\n"); return nullptr; }
struct whisper_vad_context * whisper_vad_init_with_params(struct whisper_model_loader *, struct whisper_vad_context_params) { WA_ERROR("VAD is not supported by this backend\n"); return nullptr; }
bool    whisper_vad_detect_speech(struct whisper_vad_context *, const float *, int) { return false; }
int     whisper_vad_n_probs(struct whisper_vad_context *) { return 0; }
float * whisper_vad_probs  (struct whisper_vad_context *) { return nullptr; }
struct whisper_vad_segments * whisper_vad_segments_from_probs(struct whisper_vad_context *, struct whisper_vad_params) { return nullptr; }
struct whisper_vad_segments * whisper_vad_segments_from_samples(struct whisper_vad_context *, struct whisper_vad_params, const float *, int) { return nullptr; }
int   whisper_vad_segments_n_segments(struct whisper_vad_segments *) { return 0; }
float whisper_vad_segments_get_segment_t0(struct whisper_vad_segments *, int) { return 0.0f; }
float whisper_vad_segments_get_segment_t1(struct whisper_vad_segments *, int) { return 0.0f; }
void  whisper_vad_free_segments(struct whisper_vad_segments *) {}
void  whisper_vad_free(struct whisper_vad_context *) {}

int          whisper_bench_memcpy(int) { return 0; }
const char * whisper_bench_memcpy_str(int) { return "whisper_bench_memcpy: not applicable to the HIP backend (use bench.py)\n"; }
int          whisper_bench_ggml_mul_mat(int) { return 0; }
const char * whisper_bench_ggml_mul_mat_str(int) { return "whisper_bench_ggml_mul_mat: not applicable to the HIP backend (use bench.py)\n"; }

// -------------------------------------------------------------------------------------------------
// extensions
// -------------------------------------------------------------------------------------------------
static int64_t copy_out(whisper_state * st, const float * d_src, int64_t n, float * dst, int64_t cap) {
    if (!d_src) return -1;
    if (dst && cap > 0) {
        (void) hipSetDevice(st->ctx->device);
        (void) hipStreamSynchronize(st->stream);
        if (hipMemcpy(dst, d_src, (size_t) std::min(n, cap) * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    return n;
}
int64_t whisper_amd_get_mel(struct whisper_state * st, float * dst, int64_t cap, int * n_len, int * n_mel) {
    if (n_len) *n_len = st->mel_n_len;
    if (n_mel) *n_mel = st->mel_n_mel;
    return copy_out(st, st->d_mel, (int64_t) st->mel_n_len * st->mel_n_mel, dst, cap);
}
int64_t whisper_amd_get_embd_enc(struct whisper_state * st, float * dst, int64_t cap) {
    if (!st->have_enc) return -1;
    return copy_out(st, st->d_embd_enc, (int64_t) st->enc_n_ctx * st->ctx->model.hp.n_audio_state, dst, cap);
}
int64_t whisper_amd_get_embd_conv(struct whisper_state * st, float * dst, int64_t cap) {
    if (!st->have_enc) return -1;
    return copy_out(st, st->d_embd_conv, (int64_t) st->enc_n_ctx * st->ctx->model.hp.n_audio_state, dst, cap);
}
void whisper_amd_gelu_table_f16(uint16_t * dst) {
    // recomputed on the host exactly as the loader does; device copy is identical by construction
    for (int i = 0; i < 65536; ++i) {
        _Float16 h; uint16_t u = (uint16_t) i; memcpy(&h, &u, 2);
        const float x = (float) h;
        const float g = 0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x)));
        _Float16 o = (_Float16) g; memcpy(&dst[i], &o, 2);
    }
}
void * whisper_amd_state_stream(struct whisper_state * st) { return (void *) st->stream; }

int whisper_amd_decoder_info(struct whisper_state * st, int j, double out[8], int32_t * ids, int max_ids) {
    if (!st || j < 0 || j >= WA_MAX_DECODERS) return -1;
    const auto & d = st->decoders[j];
    out[0] = d.failed; out[1] = d.completed; out[2] = d.has_ts; out[3] = d.seek_delta; out[4] = d.sequence.result_len;
    out[5] = d.sequence.avg_logprobs; out[6] = d.sequence.entropy; out[7] = st->no_speech_prob;
    const int n = (int) d.sequence.tokens.size();
    for (int i = 0; i < n && i < max_ids; ++i) ids[i] = d.sequence.tokens[i].id;
    return n;
}

int whisper_amd_full_batch(struct whisper_context * ctx, struct whisper_state ** states, int n_chunks, struct whisper_full_params params,
                           const float * const * samples, const int * n_samples) {
    // Chunks are independent: each runs the ordinary whisper_full_with_state loop on its own state and HIP
    // stream from its own host thread.  A single decode step is latency-bound and occupies a fraction of the 256 CUs, so the
    // streams overlap on the device; the weights are shared and read-only.  Callbacks, if any, fire concurrently.
    if (!ctx || !states || n_chunks <= 0) return -1;
    std::vector<int> rc(n_chunks, 0);
    std::vector<std::thread> th;
    th.reserve(n_chunks);
    // The one-launch steps own the device while they run (one slot per device, wa_decode.cpp).  Members of a lock-step group do not open
    // host-overlap windows (wa_full.cpp); a chunk that decodes alone - the last one of its group - takes the slot step by step.
    // ... and where the chunks' loops ask for a plain single-token step at the same time, ONE decoder pass serves a group of them
    {
        wa_batch_groups groups(ctx, std::vector<whisper_state *>(states, states + n_chunks));
        for (int i = 0; i < n_chunks; ++i)
            th.emplace_back([&, i]() {
                rc[i] = whisper_full_with_state(ctx, states[i], params, samples[i], n_samples[i]);
                wa_batcher_leave(states[i]->batcher);
            });
        for (auto & t : th) t.join();
    }
    for (int r : rc) if (r != 0) return r;
    return 0;
}

} // extern "C"

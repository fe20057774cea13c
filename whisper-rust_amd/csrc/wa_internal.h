// wa_internal.h - internal types of the MI355X Whisper backend (not part of the C ABI).
//
// Architecture: a static per-model execution plan, no graph interpreter.
//   whisper_context : parsed legacy-ggml model; every weight resident in ONE HBM arena in
//                     kernel-friendly layouts (fused QKV, fused cross-KV, K-padded conv1).
//   whisper_state   : one HIP stream + a fixed activation arena + KV caches + decode bookkeeping.
//   kernels         : wa_kernels.hip (hand-written gfx950 HIP), launched in a fixed order by
//                     wa_encode.cpp / wa_decode.cpp.
#pragma once

#include "../../include/whisper_amd.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <random>
#include <set>
#include <string>
#include <atomic>
#include <vector>

#define WA_MAX_DECODERS 8          // ref: whisper.cpp:148 (WHISPER_MAX_DECODERS)
#define WA_TPAD         128        // activation row padding (GEMM M tile)

// ---------------------------------------------------------------------------------------------
// logging (ref: whisper.cpp:111-138, 8935-8969: global callback, default prints to stderr)
// ---------------------------------------------------------------------------------------------
void wa_log(ggml_log_level level, const char * fmt, ...) __attribute__((format(printf, 2, 3)));
#define WA_INFO(...)  wa_log(GGML_LOG_LEVEL_INFO,  __VA_ARGS__)
#define WA_WARN(...)  wa_log(GGML_LOG_LEVEL_WARN,  __VA_ARGS__)
#define WA_ERROR(...) wa_log(GGML_LOG_LEVEL_ERROR, __VA_ARGS__)
#define WA_DEBUG(...) do { } while (0)

int64_t wa_time_us();

#define WA_HIP_OK(expr) wa_hip_ok((expr), #expr, __FILE__, __LINE__)
bool wa_hip_ok(hipError_t e, const char * what, const char * file, int line);

static inline int wa_pad(int x, int n) { return ((x + n - 1) / n) * n; }

// ---------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------
struct wa_hparams {       // ref: whisper.cpp:623-636
    int32_t n_vocab = 51864, n_audio_ctx = 1500, n_audio_state = 384, n_audio_head = 6, n_audio_layer = 4;
    int32_t n_text_ctx = 448, n_text_state = 384, n_text_head = 6, n_text_layer = 4, n_mels = 80, ftype = 1;
    float   eps = 1e-5f;
};

struct wa_vocab {         // ref: whisper.cpp:462-491
    int n_vocab = 51864;
    std::map<std::string, int> token_to_id;
    std::vector<std::string>   id_to_token;   // dense, size n_vocab after load
    int token_eot = 50256, token_sot = 50257, token_translate = 50357, token_transcribe = 50358;
    int token_solm = 50359, token_prev = 50360, token_nosp = 50361, token_not = 50362, token_beg = 50363;
    bool is_multilingual() const { return n_vocab >= 51865; }
    int  num_languages()  const { return n_vocab - 51765 - (is_multilingual() ? 1 : 0); }
};

typedef uint16_t wa_f16;  // raw IEEE half bits on the host side

struct wa_ln  { const float * w = nullptr; const float * b = nullptr; };
// A linear layer.  F16 weights: w [n_out][n_in].  Quantised weights (ggml block formats, ggml-common.h:187-214; wtype = ggml type
// 6 Q5_0 / 8 Q8_0): qs = the quants as signed bytes [n_out][8][n_in/32][4] (element 4l + e of block b at [row][l][b][e]; Q5_0's
// 5-bit values expanded at load), qd [n_out][n_in/32] the block scales widened from F16 to F32 (exact).
struct wa_lin {
    const wa_f16 * w = nullptr; const float * b = nullptr; const float * s = nullptr; int n_out = 0, n_in = 0;
    int wtype = 1; const int8_t * qs = nullptr; const float * qd = nullptr;
};

struct wa_enc_layer {
    wa_ln  attn_ln, mlp_ln;
    wa_lin qkv;      // fused [3d][d]: rows 0..d-1 query, d..2d-1 key (bias 0), 2d..3d-1 value
    wa_lin out, fc1, fc2;
};

struct wa_dec_layer {
    wa_ln  attn_ln, cross_ln, mlp_ln;
    wa_lin qkv;      // fused [3d][d] with per-column scale s[]: q,k columns d_h^-1/4, v columns 1
    wa_lin out;
    wa_lin cross_q, cross_out;
    wa_lin fc1, fc2;
};

struct wa_model {
    int type = 0;    // e_model: 1 tiny, 2 base, 3 small, 4 medium, 5 large (ref: whisper.cpp:96-103)
    wa_hparams hp;
    int n_mel_filt = 0, n_fft_filt = 0;
    std::vector<float> filters;     // host copy [n_mel][n_fft]
    int n_loaded = 0;               // 0 => header/vocab-only test model (ref: whisper.cpp:1959-1960)
    int wtype = 1;                  // ggml type of the 2-D weight matrices: 1 F16, 6 Q5_0, 8 Q8_0 (whisper.cpp:1567-1573)

    // ---- device (all inside `arena`) ----
    void * arena = nullptr; size_t arena_size = 0;
    const float  * d_filters = nullptr;   // [n_mel][n_fft]
    const float  * d_hann    = nullptr;   // [400]
    const float  * d_sincos  = nullptr;   // [2][400]  sin then cos (whisper.cpp:3031-3037)
    const wa_f16 * d_gelu    = nullptr;   // [65536]   F16 GELU table (vec.h:571-585)
    const float  * e_pe = nullptr;        // [n_audio_ctx][d]
    wa_lin conv1;                         // w [d][3*n_mels -> padded to mult of 32], k-major (k*n_mels+ic)
    wa_lin conv2;                         // w [d][3*d], k-major (k*d+ic)
    int    conv1_kpad = 0;
    const wa_f16 * conv1_g = nullptr;     // w [d][3*n_mels] in the file's (ggml im2col) order ic*3+k: reference-order path
    const wa_f16 * conv2_g = nullptr;     // w [d][3*d]      likewise
    wa_ln  e_ln;
    std::vector<wa_enc_layer> enc;
    const float  * d_pe = nullptr;        // [n_text_ctx][d]
    const wa_f16 * d_te = nullptr;        // [n_vocab][d]  (F16 models)
    wa_lin te_q;                          // the same matrix of a quantised model
    wa_ln  d_ln;
    std::vector<wa_dec_layer> dec;
    wa_lin cross_kv;                      // fused over ALL decoder layers: [L*2d][d]; per column bias+scale
    void * d_mega_layers = nullptr;       // device table of wa_mega_layer (wa_mega.h): per-layer pointers for the one-launch decode step
    int    n_cu = 0;                      // compute units of the device (= workgroups of the one-launch decode step)
};

struct whisper_context {
    int64_t t_load_us = 0, t_start_us = 0;
    whisper_context_params params;
    int device = 0;
    bool exact = true;                    // !params.flash_attn: encoder/prompt in the reference's summation order (bit-exact)
    wa_model model;
    wa_vocab vocab;
    whisper_state * state = nullptr;      // default state (only for the non-_no_state constructors)
    std::string path_model;
    std::vector<void *> batcher_cache;    // idle lock-step batchers (wa_decode.cpp), kept between whisper_amd_full_batch / whisper_full_parallel calls
    long batch_steps = 0, batch_rows = 0, batch_one_launch = 0; // the last whisper_amd_full_batch call: lock-step passes, the token rows they served, passes that were one launch
};

// ---------------------------------------------------------------------------------------------
// KV cell bookkeeping (ref: whisper.cpp:725-750, 1049-1167). Metadata only; data lives in HBM.
// ---------------------------------------------------------------------------------------------
// the sequence ids a cell belongs to (whisper.cpp:1049-1167 keeps a std::set per cell; the ids here are decoder indices and their scratch
// twins, < 2 WA_MAX_DECODERS): one word - beam search relabels every cell of every live decoder four times per step
struct wa_seq_set {
    uint64_t bits = 0;
    void insert(int32_t s) { if (s >= 0 && s < 64) bits |= 1ull << s; }
    void erase(int32_t s)  { if (s >= 0 && s < 64) bits &= ~(1ull << s); }
    void clear() { bits = 0; }
    bool empty() const { return bits == 0; }
    size_t count(int32_t s) const { return s >= 0 && s < 64 && ((bits >> s) & 1ull) ? 1 : 0; }
};
struct wa_kv_cell { int32_t pos = -1; wa_seq_set seq_id; bool has(int32_t s) const { return seq_id.count(s) != 0; } };

struct wa_kv_cache {
    uint32_t head = 0, size = 0, n = 0;
    std::vector<wa_kv_cell> cells;
    wa_f16 * k = nullptr;     // self: [n_layer][size][d]
    wa_f16 * v = nullptr;     // self: [n_layer][size][d]
};

struct wa_batch {             // ref: whisper.cpp:505-513 (one seq id per token is all whisper uses)
    int n_tokens = 0;
    std::vector<int32_t> token, pos, seq_id;
    std::vector<int8_t>  logits;
};

bool    wa_kv_find_slot(wa_kv_cache & c, const wa_batch & b);
int32_t wa_kv_cell_max (const wa_kv_cache & c);
void    wa_kv_clear    (wa_kv_cache & c);
void    wa_kv_seq_rm   (wa_kv_cache & c, int32_t seq, int32_t p0, int32_t p1);
void    wa_kv_seq_cp   (wa_kv_cache & c, int32_t src, int32_t dst, int32_t p0, int32_t p1);

// ---------------------------------------------------------------------------------------------
// decode bookkeeping (ref: whisper.cpp:493-503, 816-853)
// ---------------------------------------------------------------------------------------------
struct wa_segment {
    int64_t t0 = 0, t1 = 0;
    std::string text;
    float no_speech_prob = 0.0f;
    std::vector<whisper_token_data> tokens;
    bool speaker_turn_next = false;
};

struct wa_sequence {
    std::vector<whisper_token_data> tokens;
    int result_len = 0;
    double sum_logprobs_all = 0, sum_logprobs = 0, avg_logprobs = 0, entropy = 0, score = 0;
};

struct wa_decoder {
    wa_sequence sequence;
    int  i_batch = 0, seek_delta = 0;
    bool failed = false, completed = false, has_ts = false;
    std::vector<float> probs, logits, logprobs;
    std::vector<std::pair<double, int>> logits_id;
    mutable std::mt19937 rng;
};

struct whisper_state {
    whisper_context * ctx = nullptr;
    hipStream_t stream = nullptr;

    // timers (ref: whisper.cpp:868-881)
    int64_t t_sample_us = 0, t_encode_us = 0, t_decode_us = 0, t_batchd_us = 0, t_prompt_us = 0, t_mel_us = 0;
    int32_t n_sample = 0, n_encode = 0, n_decode = 0, n_batchd = 0, n_prompt = 0, n_fail_p = 0, n_fail_h = 0;

    // ---- log-mel ----
    int mel_n_len = 0, mel_n_len_org = 0, mel_n_mel = 0;
    float * d_mel = nullptr; size_t d_mel_cap = 0;      // [n_mel][n_len] f32 (device is the source of truth)
    float * d_pcm = nullptr; size_t d_pcm_cap = 0;      // staging for host PCM
    unsigned int * d_mel_max = nullptr;                 // ordered-uint encoding of the running max

    // ---- encoder activations (rows padded to WA_TPAD) ----
    int enc_n_ctx = 0;            // n_audio_ctx actually encoded (audio_ctx override)
    int enc_tpad  = 0;
    wa_f16 * d_melT   = nullptr;  // [2*n_ctx + 2 + pad][n_mels] f16 time-major window (+1 zero row in front)
    wa_f16 * d_h1     = nullptr;  // [1 + 2*n_ctx + pad][d] f16 conv1 output (+1 zero row in front)
    float  * d_x      = nullptr;  // [tpad][d] f32 residual stream
    wa_f16 * d_xn     = nullptr;  // [tpad][d] f16 LN output / attention output (GEMM A operand)
    wa_f16 * d_qk     = nullptr;  // [tpad][2d] f16 Q | K
    wa_f16 * d_vt     = nullptr;  // [d][tpad] f16 V transposed
    wa_f16 * d_attn_p = nullptr;  // [n_head][tpad][tpad] f16 soft-max probabilities of a layer (reference-order MFMA attention: wa_launch_attn_exact_mfma)
    wa_f16 * d_attn_pl = nullptr; // [n_head][tpad][32] f16 probabilities of the n_kv % 32 leftover cells
    wa_f16 * d_ao     = nullptr;  // [tpad][d] f16 attention output
    wa_f16 * d_ff     = nullptr;  // [tpad][4d] f16 GELU(fc1)
    float  * d_embd_enc  = nullptr; // [tpad][d] f32 encoder output
    float  * d_embd_conv = nullptr; // [tpad][d] f32 conv output before pos-emb (tests; may be null)
    bool     have_enc = false;

    // ---- cross KV (written by the encoder): [n_layer][n_head][ctx_pad][d_h] f16 each ----
    wa_f16 * d_cross_k = nullptr, * d_cross_v = nullptr;
    int cross_tpad = 0;

    // ---- self KV ----
    wa_kv_cache kv_self;
    int kv_self_n_dec = 0;

    // ---- decoder activations (rows = tokens in the batch, padded) ----
    int dec_mpad = 0;             // capacity in rows
    int32_t * d_tok = nullptr, * d_pos = nullptr, * d_cell = nullptr; // [mpad]
    int8_t  * d_mask = nullptr; size_t d_mask_cap = 0;                // [n_tokens][n_kv] 1 = masked
    float  * d_dx  = nullptr;     // [mpad][d] f32 residual
    wa_f16 * d_dxn = nullptr;     // [mpad][d] f16
    wa_f16 * d_dqkv = nullptr;    // [mpad][3d] f16 (q scaled | k scaled | v)
    wa_f16 * d_dao = nullptr;     // [mpad][d] f16
    wa_f16 * d_dff = nullptr;     // [mpad][4d] f16
    wa_f16 * d_dq  = nullptr;     // [mpad][d] f16 cross query
    float  * d_att_partial = nullptr; // [512][32][64] f32: P V partial-sum chains of (token, head) pairs (decode)
    wa_f16 * d_att_pleft = nullptr;   // [512][32] f16: probabilities of the n_kv % 32 leftover cells
    wa_f16 * d_im2col = nullptr;      // reference-order conv: [2T][3*n_mels] / [T][3d] f16
    float  * d_logits = nullptr;  // [mpad][n_vocab] f32
    int32_t * d_rows = nullptr;   // [mpad] row indices that need logits
    float  * d_aheads_qk = nullptr; // DTW capture
    // quantised models (wa_quant.hip): F32 operands of the quantised products and their Q8_0 form (shared by encoder and decoder)
    float  * d_q32a = nullptr;      // [tpad][d]   LayerNorm / attention output
    float  * d_q32b = nullptr;      // [tpad][4d]  GELU(fc1)
    int8_t * d_q8   = nullptr;      // [tpad][4d]  Q8_0 operand rows (q8_rows = tpad)
    int      q8_rows = 0;
    float  * d_q8d  = nullptr;      // [tpad][4d / 32]

    // hipGraph of the single-token decoder pass (launch-bound inner loop); parameters that change per token live in d_dyn
    int32_t * d_dyn = nullptr;            // {n_kv, kv_head}
    hipGraphExec_t dec_graph = nullptr;
    int dec_graph_T = 0; uint32_t dec_graph_kv_size = 0; const void * dec_graph_kv_k = nullptr;
    bool graphs_enabled = true;

    // one-launch decode step (wa_mega.hip): hand-off granules [layer][8][2d], status word, launch sequence number
    unsigned long long * d_mega_gr = nullptr;
    unsigned long long * d_mega_cgr = nullptr;   // [layer][head][2048]: exchange area of the cross-attention workgroups
    float * d_mega_out = nullptr;         // [n_vocab] logits + status word
    unsigned * d_mega_status = nullptr;   // = d_mega_out + n_vocab
    unsigned mega_seq = 0;
    bool mega_enabled = false;
    // a hand-off time-out (e.g. a co-tenant kernel held CUs: the step's workgroups were not all resident) pauses the one-launch forms for
    // `*_pause` decoder passes, which take the launch sequence, and then tries again (doubling the pause each time; after 8 time-outs: off)
    int mega_pause = 0, mega_timeouts = 0, rows_pause = 0, rows_timeouts = 0;
    unsigned spec_seq[2] = { 0, 0 };               // launch numbers of the two launches in flight (their echo is checked on arrival)
    // host overlap (wa_decode.cpp wa_spec_*): the device predicts the next token and decodes it while the host still
    // applies the reference's sampling rules to the previous logits; two output / record / state buffers alternate
    float * d_mega_out2 = nullptr;                 // second logits + status + token buffer
    unsigned * d_mega_rec[2] = { nullptr, nullptr };   // candidate records [n_workgroups][8]
    int * d_mega_ps[2] = { nullptr, nullptr };     // sampling state after a launch's token
    unsigned * d_mega_smask = nullptr;             // [n_vocab / 32 + 1] per-call suppression bits
    float * h_spec[2] = { nullptr, nullptr };      // pinned [n_vocab + 16]
    hipStream_t copy_stream = nullptr;
    // the decode step for 2..8 token rows as one launch (wa_rows.hip): hand-off granules [layer][8][8][2d], the cross-attention quarters' exchange
    // area [layer][8][head][2048], {status, sequence echo}; allocated on first use (wa_rows_prepare)
    unsigned long long * d_rows_gr = nullptr, * d_rows_cgr = nullptr;
    unsigned * d_rows_status = nullptr;
    bool rows_enabled = false;
    long n_rows_steps = 0, n_rows_fallback = 0;      // passes served by the one-launch form / sent to the launch sequence after a status
    struct wa_batcher * batcher = nullptr;   // set while this state is a member of a whisper_amd_full_batch call (wa_decode.cpp)
    bool solo_step = false;                  // the next plain step of a lock-step member is decoded by the member alone (its row of a pass asked to be redone by the launch sequence)
    // several logits rows of one pass (beam search, best_of): with `defer_rows` set by the caller wa_decode leaves them in the pinned staging rows and every
    // decoder's own host thread moves its row into `logits` (wa_full.cpp: process_logits) - five 207 KB copies side by side instead of one after the other
    // wide quantised models (d > 768): the single-token step runs on the several-rows kernel with ONE row - it streams its weights through LDS-DMA and has no
    // spilled prefetch registers (large-v3-q5_0: 2.04 ms against k_decode_mega_q's 2.31); WHISPER_AMD_SINGLE_ROWS=0 / 1 overrides
    bool single_via_rows = false;
    bool defer_rows = false;
    int  staged_n = 0;
    std::vector<int32_t> staged_of;      // staged_of[r] = batch index of staging row r
    bool spec_owner = false;                 // this state holds its device's one-launch slot (wa_spec_begin .. wa_spec_end)
    hipEvent_t ev_k[2] = { nullptr, nullptr }, ev_c[2] = { nullptr, nullptr };
    int n_spec_ok = 0, n_spec_miss = 0;

    // pinned host staging
    int32_t * h_stage_i32 = nullptr; int8_t * h_stage_mask = nullptr; size_t h_mask_cap = 0;
    float * h_logits_pinned = nullptr; size_t h_logits_cap = 0;

    wa_batch batch;
    wa_decoder decoders[WA_MAX_DECODERS];

    std::vector<float> logits;            // [n_tokens][n_vocab] host copy (only flagged rows valid)
    std::vector<wa_segment> result_all;
    std::vector<whisper_token> prompt_past;
    int   lang_id = 0;
    float no_speech_prob = 0.0f;
    int32_t exp_n_audio_ctx = 0;

    // heuristic token timestamps (ref: whisper.cpp:937-942): carried from segment to segment within one call
    int64_t t_beg = 0, t_last = 0;
    whisper_token tid_last = 0;
    std::vector<float> energy;             // PCM signal energy

    // DTW (ref: whisper.cpp:856-860, 946-948)
    std::vector<std::vector<int>> aheads;  // per text layer: list of heads
    std::vector<int> aheads_slot;          // per text layer: slot in d_aheads_qk during the DTW pass (-1: none)
    std::vector<float> aheads_cross_QKs_data;
    int aheads_n = 0;
};

// ---------------------------------------------------------------------------------------------
// internal entry points
// ---------------------------------------------------------------------------------------------
bool wa_model_load(whisper_model_loader * loader, whisper_context & ctx);           // wa_loader.cpp
void wa_model_free(whisper_context & ctx);

bool wa_mel_compute(whisper_context & ctx, whisper_state & st, const float * samples, int n_samples); // wa_encode.cpp
bool wa_mel_set    (whisper_context & ctx, whisper_state & st, const float * data, int n_len, int n_mel);
bool wa_encode     (whisper_context & ctx, whisper_state & st, int mel_offset, ggml_abort_callback cb, void * cb_data);
bool wa_decode     (whisper_context & ctx, whisper_state & st, const wa_batch & batch, bool save_aheads,
                    ggml_abort_callback cb, void * cb_data);                         // wa_decode.cpp
// host-overlapped greedy decoding on the one-launch step (wa_decode.cpp); launch k decodes position pos0 + k
struct wa_spec_state { int last, penult, seek_delta, has_ts; };
bool wa_spec_begin (whisper_context & ctx, whisper_state & st, const std::vector<uint32_t> & suppress_bits);
bool wa_spec_launch(whisper_context & ctx, whisper_state & st, int k, int pos, int token /* < 0: the device's own pick */, const wa_spec_state & after);
int  wa_spec_wait  (whisper_context & ctx, whisper_state & st, int k, int * token_used);   // 0 ok (logits in st.logits row 0), 1 redo, -1 gave up
void wa_spec_drain (whisper_context & ctx, whisper_state & st);
void wa_spec_end   (whisper_context & ctx, whisper_state & st);
// lock-step batched decode of the chunks of one whisper_amd_full_batch call (wa_decode.cpp)
struct wa_batcher;
wa_batcher * wa_batcher_create(whisper_context & ctx, int n_members);      // null: not applicable (quantised model, one chunk)
void wa_batcher_leave(wa_batcher * b);
void wa_batcher_destroy(wa_batcher * b);          // hands the batcher back to its context's cache
void wa_batcher_release(wa_batcher * b);          // frees it
void wa_batcher_free_all(whisper_context & ctx);  // whisper_free
void wa_batcher_stats(const wa_batcher * b, long * steps, long * rows, long * one_launch = nullptr);
bool wa_state_alloc(whisper_context & ctx, whisper_state & st);
void wa_state_release(whisper_state & st);
bool wa_kv_self_realloc(whisper_context & ctx, whisper_state & st, int n_cells);
// Encoder passes in flight on a device (wa_encode.cpp).  A one-launch decode step needs every CU's LDS to itself: beside several streams of
// encoder launches its workgroups can wait many milliseconds for a free CU (seen: a hand-off time-out in a chunk's prompt pass while seven
// other chunks of its group were still encoding).  Steps outside a window take the launch sequence while this is non-zero.
std::atomic<int> & wa_encoders_in_flight(int device);
bool wa_rows_prepare(whisper_context & ctx, whisper_state & st);      // buffers of the several-rows one-launch step (wa_encode.cpp); false: not available

std::vector<int> wa_tokenize(const wa_vocab & vocab, const std::string & text);       // wa_api.cpp
int  wa_full(whisper_context * ctx, whisper_state * st, whisper_full_params params, const float * samples, int n_samples); // wa_full.cpp

// wa_full.cpp - host side of whisper_full_with_state: seek loop over 30 s windows, temperature
// fallback ladder, prompt construction, logits filtering, greedy / beam sampling, segment assembly.
//
// Behavioural contract: sys/whisper.cpp/src/whisper.cpp:6795-7711 (loop), 6109-6417 (logit rules),
// 6432-6613 (sampling and scoring).  The structure below is our own (one runner object per call,
// explicit phases); every rule cites the reference lines it reproduces.  All device work goes through
// wa_encode / wa_decode.
#include "wa_internal.h"
#include "wa_expf8.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <regex>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <memory>
#include <immintrin.h>

void wa_dtw_timestamps(whisper_context * ctx, whisper_state * st, const whisper_full_params & params, int i_segment, size_t n_segments,
                       int seek, int n_frames, int medfilt_width);   // wa_dtw.cpp

namespace {

const char * const k_non_speech[] = {   // whisper.cpp:6102-6107
    "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^",
    "_", "`", "{", "|", "}", "~", "「", "」", "『", "』", "<<", ">>", "<<<", ">>>", "--",
    "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "♪♪",
    "♪♪♪", "♩", "♪", "♫", "♬", "♭", "♮", "♯",
};

// log-softmax / softmax over the filtered logits (whisper.cpp:6109-6143).  The reference forms
//     lse = logf( sum_i expf(logits[i] - max) ) + max
// as a sequential F32 sum in index order with libm expf.  The same value is produced here without evaluating
// expf for terms that provably cannot change the running sum: in round-to-nearest S + t == S whenever
// t < ulp(S)/2, and ulp(S)/2 >= 2^-25 S, so every term with  logits[i] - max < ln(S) - 17.5  (2^-25 = e^-17.33;
// the 0.17 margin covers the <= 1 ulp errors of expf/logf) is skipped - for a peaked distribution that is almost
// all of the vocabulary.  The terms that matter go through the same libm calls in the same order.
// `n_max` elements are scanned for the maximum: the reference takes it over the WHOLE vector it is handed, which
// for the no-speech probe is every row of the prompt's logits buffer, not just the n it normalises (whisper.cpp:6113).
// 8-wide helpers for the 51865-wide passes (host file is built with -mavx2, no FMA contraction)
inline float max_f32(const float * x, size_t n) {
    size_t i = 0;
    float mx = -INFINITY;
    if (n >= 8) {
        __m256 m = _mm256_loadu_ps(x);
        for (i = 8; i + 8 <= n; i += 8) m = _mm256_max_ps(m, _mm256_loadu_ps(x + i));
        float t[8]; _mm256_storeu_ps(t, m);
        for (int k = 0; k < 8; ++k) if (t[k] > mx) mx = t[k];
    }
    for (; i < n; ++i) if (x[i] > mx) mx = x[i];
    return mx;
}

float logsumexp_ref_order(const float * logits, int n, size_t n_max = 0) {
    const float mx = max_f32(logits, n_max > (size_t) n ? n_max : (size_t) n);
    if (wa_expf8_usable()) return logf(wa_sum_expf8(logits, n, mx)) + mx;      // the same sum, eight expf at a time (wa_expf8.h)
    float S = 0.0f;
    float thr = -INFINITY;                  // terms with (logits[i] - mx) < thr leave S unchanged
    auto add = [&](int i) {                 // index order is preserved: candidates of a block are visited in order
        const float d = logits[i] - mx;
        if (!(d >= thr) || !(logits[i] > -INFINITY)) return;
        const float S0 = S;
        S += expf(d);
        if (S != S0) thr = logf(S) - 17.5f;
    };
    int i = 0;
    const __m256 vmx = _mm256_set1_ps(mx), vninf = _mm256_set1_ps(-INFINITY);
    for (; i + 8 <= n; i += 8) {
        const __m256 x = _mm256_loadu_ps(logits + i);
        const __m256 d = _mm256_sub_ps(x, vmx);
        // thr only grows, so a lane that fails the test now would also fail it later in this block
        const __m256 ok = _mm256_and_ps(_mm256_cmp_ps(d, _mm256_set1_ps(thr), _CMP_GE_OQ), _mm256_cmp_ps(x, vninf, _CMP_GT_OQ));
        int mask = _mm256_movemask_ps(ok);
        while (mask) { const int k = __builtin_ctz(mask); mask &= mask - 1; add(i + k); }
    }
    for (; i < n; ++i) add(i);
    return logf(S) + mx;
}
void compute_logprobs(const float * logits, int n, float * logprobs, size_t n_max = 0) {
    const float lse = logsumexp_ref_order(logits, n, n_max);
    int i = 0;
    const __m256 vl = _mm256_set1_ps(lse), vninf = _mm256_set1_ps(-INFINITY);
    for (; i + 8 <= n; i += 8) {
        const __m256 x = _mm256_loadu_ps(logits + i);
        const __m256 fin = _mm256_cmp_ps(x, vninf, _CMP_GT_OQ);
        _mm256_storeu_ps(logprobs + i, _mm256_blendv_ps(vninf, _mm256_sub_ps(x, vl), fin));
    }
    for (; i < n; ++i) logprobs[i] = logits[i] > -INFINITY ? logits[i] - lse : -INFINITY;
}
void compute_probs(const float * logits, int n, const float * logprobs, float * probs) {
    if (wa_expf8_usable()) { wa_probs_expf8(logits, n, logprobs, probs); return; }
    for (int i = 0; i < n; ++i) probs[i] = logits[i] == -INFINITY ? 0.0f : expf(logprobs[i]);
}

// -------------------------------------------------------------------------------------------------
// Heuristic token-level timestamps and segment wrapping (params.token_timestamps / max_len / split_on_word): host-side
// post-processing of a finished segment.  Behavioural contract: whisper.cpp:8326-8616 and 6047-6100 (pinned by the reference
// engine's goldens, tests/golden/s128_token_ts.json).  Organised here as a time line over flat arrays that is built in four passes:
//   anchors  - tokens whose own timestamp guess is trusted pin a boundary of the time line;
//   spread   - every run of boundaries still open between two pinned ones is divided in proportion to the tokens' spoken weights;
//   repair   - boundaries that ended up out of order are pushed forward;
//   snap     - each text token's ends slide to where the windowed |PCM| energy crosses half of its local mean.
// Integer / float / double types of every quantity are the contract's (int64 boundaries, float weights and energies, double
// proportions), which is what makes the results identical.
// -------------------------------------------------------------------------------------------------
float spoken_weight(const char * text) {
    static const std::array<float, 256> weight_of = [] {
        std::array<float, 256> w;
        w.fill(1.00f);
        w[(unsigned char) ' '] = 0.01f;
        w[(unsigned char) ','] = 2.00f;
        for (unsigned char c : { '.', '!', '?' }) w[c] = 3.00f;
        for (unsigned char c = '0'; c <= '9'; ++c) w[c] = 3.00f;
        return w;
    }();
    float total = 0.0f;
    for (const unsigned char * c = (const unsigned char *) text; *c; ++c) total += weight_of[*c];
    return total;
}

// mean |x| over the window [i - half, i + half] clipped to the signal, always divided by the full window length
std::vector<float> signal_energy(const float * signal, int n, int half) {
    std::vector<float> e((size_t) std::max(n, 0));
    const float full = (float) (2 * half + 1);
    for (int i = 0; i < n; ++i) {
        const int lo = std::max(i - half, 0), hi = std::min(i + half, n - 1);
        float acc = 0;
        for (int k = lo; k <= hi; ++k) acc += fabsf(signal[k]);
        e[i] = acc / full;
    }
    return e;
}

struct token_timeline {
    const int n;
    const int64_t seg_lo, seg_hi;                       // the segment's own span, 10 ms units
    std::vector<int64_t> lo, hi;                        // per token: start / end boundary
    std::vector<float> weight;
    const std::vector<float> & energy;

    token_timeline(const std::vector<whisper_token_data> & toks, int64_t t0, int64_t t1, const std::vector<float> & en)
        : n((int) toks.size()), seg_lo(t0), seg_hi(t1), lo(toks.size()), hi(toks.size()), weight(toks.size()), energy(en) {
        for (int j = 0; j < n; ++j) { lo[j] = toks[j].t0; hi[j] = toks[j].t1; }
    }
    float energy_at(int k) const { return energy[(size_t) std::min(std::max(k, 0), (int) energy.size() - 1)]; }
    int to_sample(int64_t t) const {
        const int smp = (int) (((t - seg_lo) * WHISPER_SAMPLE_RATE) / 100);
        return std::min(std::max(smp, 0), (int) energy.size() - 1);
    }
    int64_t to_time(int smp) const { return (100ll * smp) / WHISPER_SAMPLE_RATE + seg_lo; }

    // pass 1.  `origin`, `resume`, `high_tid` live in the state: a window's segments continue the previous segment's time line.
    void anchors(whisper_context * ctx, const std::vector<whisper_token_data> & toks, float min_pt, float min_ptsum, int64_t & origin, int64_t & resume,
                 whisper_token & high_tid) {
        const whisper_token beg = ctx->vocab.token_beg;
        if (toks[0].id == beg) { lo[0] = hi[0] = lo[1] = seg_lo; origin = resume = seg_lo; high_tid = beg; }
        else lo[0] = resume;
        for (int j = 0; j < n; ++j) {
            weight[j] = spoken_weight(whisper_token_to_str(ctx, toks[j].id));
            const int64_t guess = origin + 2 * (toks[j].tid - beg);
            const bool trusted = toks[j].pt > min_pt && toks[j].ptsum > min_ptsum && toks[j].tid > high_tid && guess <= seg_hi;
            if (!trusted) continue;
            if (j > 0) hi[j - 1] = guess;
            lo[j] = guess;
            high_tid = toks[j].tid;
        }
        hi[n - 2] = seg_hi;
        lo[n - 1] = hi[n - 1] = seg_hi;
        resume = seg_hi;
    }
    // pass 2: runs [first, last] whose inner boundaries are open; `last` is the next token with a pinned end (or the final token)
    void spread() {
        for (int first = 0; first < n;) {
            int last = first;
            while (last < n && hi[last] < 0) ++last;          // open = below zero (token data starts out with t0 = t1 = -1); pinned ends are times >= 0
            if (last >= n) last = n - 1;
            if (last > first) {
                double total = 0.0;
                for (int j = first; j <= last; ++j) total += weight[j];
                const double span = (double) (hi[last] - lo[first]);
                for (int j = first; j < last; ++j) {
                    const int64_t cut = (int64_t) ((double) lo[j] + span * weight[j] / total);
                    hi[j] = cut;
                    lo[j + 1] = cut;
                }
            }
            first = last + 1;
        }
    }
    // pass 3
    void repair() {
        for (int j = 0; j + 1 < n; ++j) {
            if (hi[j] < 0) lo[j + 1] = hi[j];
            if (j > 0 && hi[j - 1] > lo[j]) { lo[j] = hi[j - 1]; hi[j] = std::max(lo[j], hi[j]); }
        }
    }
    // pass 4, one text token: both ends move to the nearest crossing of half the mean energy of the token's neighbourhood
    void snap(int j) {
        const int reach = WHISPER_SAMPLE_RATE / 8, n_smp = (int) energy.size();
        int s_lo = to_sample(lo[j]), s_hi = to_sample(hi[j]);
        const int w_lo = std::max(s_lo - reach, 0), w_hi = std::min(s_hi + reach, n_smp);
        float total = 0.0f;
        for (int k = w_lo; k < w_hi; ++k) total += energy_at(k);
        const float level = 0.5 * total / (w_hi - w_lo);
        const auto loud = [&](int k) { return energy_at(k) > level; };
        const auto quiet = [&](int k) { return energy_at(k) < level; };
        int k = s_lo;
        if (loud(k) && j > 0) {                              // speech already under way: the start moves back, but not into the previous token
            while (k > 0 && loud(k)) --k;
            lo[j] = to_time(k);
            if (lo[j] < hi[j - 1]) lo[j] = hi[j - 1]; else s_lo = k;
        } else {                                             // leading silence: the start moves forward
            while (quiet(k) && k < s_hi) ++k;
            s_lo = k;
            lo[j] = to_time(k);
        }
        k = s_hi;
        if (loud(k)) {                                       // still speaking at the end: it moves forward, but not into the next token
            while (k < n_smp - 1 && loud(k)) ++k;
            hi[j] = to_time(k);
            if (j < n - 1 && hi[j] > lo[j + 1]) hi[j] = lo[j + 1]; else s_hi = k;
        } else {                                             // trailing silence: the end moves back
            while (quiet(k) && k > s_lo) --k;
            s_hi = k;
            hi[j] = to_time(k);
        }
    }
};

void token_level_timestamps(whisper_context * ctx, whisper_state * st, int i_segment, float thold_pt, float thold_ptsum) {
    auto & seg = st->result_all[i_segment];
    auto & toks = seg.tokens;
    if (st->energy.empty()) { WA_ERROR("%s: no signal data available\n", __func__); return; }
    if (toks.empty()) return;
    if (toks.size() == 1) { toks[0].t0 = seg.t0; toks[0].t1 = seg.t1; return; }
    token_timeline line(toks, seg.t0, seg.t1, st->energy);
    line.anchors(ctx, toks, thold_pt, thold_ptsum, st->t_beg, st->t_last, st->tid_last);
    line.spread();
    line.repair();
    for (int j = 0; j < line.n; ++j) if (toks[j].id < ctx->vocab.token_eot) line.snap(j);
    for (int j = 0; j < line.n; ++j) { toks[j].t0 = line.lo[j]; toks[j].t1 = line.hi[j]; toks[j].vlen = line.weight[j]; }
}

// Re-cut the most recent segment so that no piece exceeds max_len characters of token text (params.max_len, split_on_word); returns
// the number of segments it became.  Two steps: find the token positions where a new piece starts, then rebuild the pieces.
int wrap_segment(whisper_context * ctx, whisper_state * st, int max_len, bool split_on_word) {
    const wa_segment whole = st->result_all.back();
    const auto & toks = whole.tokens;
    const int n = (int) toks.size();
    const auto printable = [&](int i) { return toks[i].id < ctx->vocab.token_eot; };
    std::vector<int> starts = { 0 };                    // first token of every piece
    for (int i = 0, chars = 0; i < n; ++i) {
        if (!printable(i)) continue;
        const char * txt = whisper_token_to_str(ctx, toks[i].id);
        const int len = (int) strlen(txt);
        const bool may_cut = i > starts.back() && (!split_on_word || txt[0] == ' ');
        if (chars + len > max_len && may_cut) { starts.push_back(i); chars = 0; }
        chars += len;
    }
    st->result_all.pop_back();
    for (size_t k = 0; k < starts.size(); ++k) {
        const int first = starts[k], last = k + 1 < starts.size() ? starts[k + 1] : n;
        const bool final_piece = k + 1 == starts.size();
        wa_segment piece;
        piece.t0 = k == 0 ? whole.t0 : toks[first].t0;
        piece.t1 = final_piece ? whole.t1 : toks[last].t0;
        if (k == 0) piece.no_speech_prob = whole.no_speech_prob;       // (later pieces keep the default, as in the contract)
        piece.speaker_turn_next = final_piece ? whole.speaker_turn_next : false;
        piece.tokens.assign(toks.begin() + first, toks.begin() + last);
        for (int i = first; i < last; ++i) if (printable(i)) piece.text += whisper_token_to_str(ctx, toks[i].id);
        st->result_all.push_back(std::move(piece));
    }
    return (int) starts.size();
}

// a beam candidate = the sequence of decoder `decoder_idx` as it stands + one drawn token (the reference copies the whole sequence into every
// candidate, whisper.cpp:7218-7224; here only the decoders that survive get one built)
struct beam_candidate { int decoder_idx; int seek_delta; bool has_ts; whisper_token_data tok; double sum_logprobs_all; };

bool same_tokens(const wa_sequence & a, const wa_sequence & b) {   // whisper.cpp:6419-6430
    if (a.tokens.size() != b.tokens.size()) return false;
    for (int i = (int) a.tokens.size() - 1; i >= 0; --i) if (a.tokens[i].id != b.tokens[i].id) return false;
    return true;
}

// Host workers for the per-decoder passes of a multi-decoder step (beam search, best_of > 1): the reference spreads them over
// n_threads (whisper.cpp:7442-7476).  Each decoder's logits rules, its log-soft-max over the vocabulary and its draws touch only that
// decoder's buffers and its own RNG, so the results do not depend on the schedule.  Lives for one whisper_full call.
struct wa_pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    const std::function<void(int)> * fn = nullptr;
    int n_items = 0, next = 0, done = 0;
    unsigned long gen = 0;
    bool stop = false;

    explicit wa_pool(int n_workers) {
        for (int i = 0; i < n_workers; ++i) th.emplace_back([this] { worker(); });
    }
    ~wa_pool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (auto & t : th) t.join();
    }
    void drain(std::unique_lock<std::mutex> & lk) {        // take items until none is left (lock held on entry and exit)
        while (next < n_items) {
            const int i = next++;
            lk.unlock();
            (*fn)(i);
            lk.lock();
            if (++done == n_items) cv_done.notify_all();
        }
    }
    void worker() {
        std::unique_lock<std::mutex> lk(m);
        unsigned long seen = 0;
        for (;;) {
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            drain(lk);
        }
    }
    void run(int n, const std::function<void(int)> & f) {
        if (n <= 1 || th.empty()) { for (int i = 0; i < n; ++i) f(i); return; }
        std::unique_lock<std::mutex> lk(m);
        fn = &f; n_items = n; next = 0; done = 0; ++gen;
        cv.notify_all();
        drain(lk);                                         // the calling thread works too
        cv_done.wait(lk, [&] { return done == n_items; });
    }
};

struct runner {
    whisper_context * ctx;
    whisper_state   * st;
    whisper_full_params p;
    const wa_vocab & vocab;
    const int n_vocab;
    std::vector<int> suppress_ids;      // tokens killed by suppress_regex / suppress_nst, resolved once per call
    int blank_id = -1;
    std::unique_ptr<wa_pool> pool;      // created with the first multi-decoder step
    // (WHISPER_AMD_SAMPLE_TRACE) where decoder 0's host time goes: logit rules + log-soft-max, probabilities, top-k draws, beam bookkeeping, state machine
    int64_t tr_rules = 0, tr_probs = 0, tr_topk = 0, tr_beam = 0, tr_state = 0, tr_steps = 0;
    const bool tr_on = getenv("WHISPER_AMD_SAMPLE_TRACE") != nullptr;
    void par_for(int n, const std::function<void(int)> & f) {
        // one thread per decoder (up to 8): n_threads is sized for a CPU engine's matrix products, which this backend does not run
        if (n > 1 && !pool) pool.reset(new wa_pool(std::min(n, 8) - 1));
        if (pool) pool->run(n, f); else for (int i = 0; i < n; ++i) f(i);
    }

    runner(whisper_context * c, whisper_state * s, const whisper_full_params & params)
        : ctx(c), st(s), p(params), vocab(c->vocab), n_vocab(c->vocab.n_vocab) {}

    void resolve_suppress_lists() {
        auto sp = vocab.token_to_id.find(" ");
        blank_id = sp == vocab.token_to_id.end() ? -1 : sp->second;   // the reference throws here (whisper.cpp:6191); we skip
        if (p.suppress_regex) {                                        // whisper.cpp:6232-6239
            std::regex re(p.suppress_regex);
            for (const auto & kv : vocab.token_to_id) if (std::regex_match(kv.first, re)) suppress_ids.push_back(kv.second);
        }
        if (p.suppress_nst) {                                          // whisper.cpp:6243-6260
            for (const char * t : k_non_speech) {
                const std::string a = t, b = " " + a;
                for (const auto & s : { a, b }) { auto it = vocab.token_to_id.find(s); if (it != vocab.token_to_id.end()) suppress_ids.push_back(it->second); }
            }
            for (const char * t : { " -", " '" }) { auto it = vocab.token_to_id.find(t); if (it != vocab.token_to_id.end()) suppress_ids.push_back(it->second); }
        }
    }

    // the call-constant part of the logit rules below as a bit set, for the device's next-token prediction (wa_spec_*)
    std::vector<uint32_t> suppress_bits() const {
        std::vector<uint32_t> b((size_t) n_vocab / 32 + 1, 0u);
        auto set = [&](int i) { if (i >= 0 && i < n_vocab) b[i >> 5] |= 1u << (i & 31); };
        set(vocab.token_not); set(vocab.token_sot); set(vocab.token_nosp);
        if (!p.tdrz_enable) set(vocab.token_solm);
        set(vocab.token_translate); set(vocab.token_transcribe); set(vocab.token_prev);
        for (int i = 0; i < 100; ++i) set(vocab.token_sot + 1 + i);
        if (p.no_timestamps) for (int i = vocab.token_beg; i < n_vocab; ++i) set(i);
        for (int id : suppress_ids) set(id);
        return b;
    }

    // ---- whisper_process_logits (whisper.cpp:6149-6417) ----
    // `full_probs`: the samplers that draw from the distribution need every probs[i]; the greedy arg-max needs only
    // the timestamp range and the candidates around the maximum (sample_token_best), so the 51865 expf calls are skipped.
    void process_logits(wa_decoder & dec, float temperature, bool full_probs) {
        const int64_t tr0 = tr_on ? wa_time_us() : 0;
        const auto & cur = dec.sequence.tokens;
        const bool is_initial = cur.empty();
        const int n = n_vocab;
        dec.logits.resize(n); dec.probs.resize(n); dec.logprobs.resize(n);
        float * logits = dec.logits.data();
        if (st->staged_n > 0)       // this decoder's row is still in the staging rows of the pass (wa_decode: defer_rows): it goes into state->logits here, on this thread
            for (int r = 0; r < st->staged_n; ++r)
                if (st->staged_of[r] == dec.i_batch) memcpy(st->logits.data() + (size_t) dec.i_batch * n, st->h_logits_pinned + (size_t) r * n, n * sizeof(float));
        memcpy(logits, st->logits.data() + (size_t) dec.i_batch * n, n * sizeof(float));
        if (temperature > 0.0f) for (int i = 0; i < n; ++i) logits[i] /= temperature;

        if (p.suppress_blank && is_initial) { logits[vocab.token_eot] = -INFINITY; if (blank_id >= 0) logits[blank_id] = -INFINITY; }
        logits[vocab.token_not] = -INFINITY;
        if (p.no_timestamps) for (int i = vocab.token_beg; i < n; ++i) logits[i] = -INFINITY;
        logits[vocab.token_sot] = -INFINITY;
        logits[vocab.token_nosp] = -INFINITY;
        if (!p.tdrz_enable) logits[vocab.token_solm] = -INFINITY;
        logits[vocab.token_translate] = -INFINITY;
        logits[vocab.token_transcribe] = -INFINITY;
        logits[vocab.token_prev] = -INFINITY;
        for (int i = 0; i < 100; ++i) { const int t = vocab.token_sot + 1 + i; if (t < n) logits[t] = -INFINITY; }   // all g_lang entries
        if (p.logits_filter_callback) p.logits_filter_callback(ctx, st, cur.data(), (int) cur.size(), logits, p.logits_filter_callback_user_data);
        for (int id : suppress_ids) logits[id] = -INFINITY;

        {   // timestamps come in pairs, except directly before EOT (whisper.cpp:6264-6281)
            const bool last_ts = !cur.empty() && cur.back().id >= vocab.token_beg;
            const bool penult_ts = cur.size() < 2 || cur[cur.size() - 2].id >= vocab.token_beg;
            if (last_ts) {
                if (penult_ts) for (int i = vocab.token_beg; i < n; ++i) logits[i] = -INFINITY;
                else           for (int i = 0; i < vocab.token_eot; ++i) logits[i] = -INFINITY;
            }
        }
        if (is_initial && p.max_initial_ts > 0.0f) {                   // whisper.cpp:6285-6292
            const float precision = float(WHISPER_CHUNK_SIZE) / ctx->model.hp.n_audio_ctx;
            const int tid0 = (int) std::round(p.max_initial_ts / precision);
            for (int i = vocab.token_beg + tid0 + 1; i < n; ++i) logits[i] = -INFINITY;
        }
        if (dec.has_ts) {                                              // monotonic timestamps (whisper.cpp:6296-6302)
            const int tid0 = dec.seek_delta / 2;
            for (int i = vocab.token_beg; i < vocab.token_beg + tid0 && i < n; ++i) logits[i] = -INFINITY;
        }
        float * logprobs = dec.logprobs.data();
        compute_logprobs(logits, n, logprobs);
        {   // if the timestamp mass beats every text token, force a timestamp (whisper.cpp:6309-6333)
            float ts_logprob = -INFINITY;
            {
                const float mx = max_f32(logprobs + vocab.token_beg, n - vocab.token_beg);
                float lse = 0.0f;
                if (wa_expf8_usable()) lse = wa_sum_expf8(logprobs + vocab.token_beg, n - vocab.token_beg, mx);
                else for (int i = vocab.token_beg; i < n; ++i) if (logprobs[i] > -INFINITY) lse += expf(logprobs[i] - mx);
                if (lse > 0.0f) ts_logprob = logf(lse) + mx;
            }
            const float max_text = max_f32(logprobs, vocab.token_beg);
            if (ts_logprob > max_text) for (int i = 0; i < vocab.token_beg; ++i) { logits[i] = -INFINITY; logprobs[i] = -INFINITY; }
        }
        const int64_t tr1 = tr_on ? wa_time_us() : 0;
        if (full_probs) compute_probs(logits, n, logprobs, dec.probs.data());
        else compute_probs(logits + vocab.token_beg, n - vocab.token_beg, logprobs + vocab.token_beg, dec.probs.data() + vocab.token_beg);
        if (tr_on && &dec == &st->decoders[0]) { const int64_t tr2 = wa_time_us(); tr_rules += tr1 - tr0; tr_probs += tr2 - tr1; }
    }

    // timestamp statistics shared by both samplers (whisper.cpp:6447-6465 / 6529-6547)
    void ts_stats(const wa_decoder & dec, int & tid, float & pt, float & ptsum, int tid_init) const {
        double sum_ts = 0.0, max_ts = 0.0;
        tid = tid_init;
        for (int i = vocab.token_beg; i < n_vocab; ++i) {
            const float pr = dec.probs[i];
            if (pr == -INFINITY) continue;
            sum_ts += pr;
            if (max_ts < pr) { max_ts = pr; tid = i; }
        }
        pt = (float) (max_ts / (sum_ts + 1e-10));
        ptsum = (float) sum_ts;
    }

    whisper_token_data sample_token(const wa_decoder & dec, bool best) const {        // whisper.cpp:6432-6489
        whisper_token_data r = { 0, 0, 0.0f, 0.0f, 0.0f, 0.0f, -1, -1, -1, 0.0f };
        ts_stats(dec, r.tid, r.pt, r.ptsum, 0);
        if (best) {
            // reference: first i maximising probs[i] = expf(logprobs[i]) (strict <, whisper.cpp:6468-6474).  expf is
            // monotonic, so the maximum is expf(max logprob); only entries within 1e-5 of it can round to the same
            // probability, and only those are exponentiated.
            const float lmax = max_f32(dec.logprobs.data(), n_vocab);
            if (lmax > -INFINITY) {
                const float pmax = expf(lmax);
                const __m256 vth = _mm256_set1_ps(lmax - 1e-5f);
                int i = 0;
                bool found = false;
                for (; i + 8 <= n_vocab && !found; i += 8) {
                    int mask = _mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(dec.logprobs.data() + i), vth, _CMP_GE_OQ));
                    while (mask) {
                        const int k = __builtin_ctz(mask); mask &= mask - 1;
                        if (expf(dec.logprobs[i + k]) == pmax) { r.id = i + k; r.p = pmax; r.plog = dec.logprobs[i + k]; found = true; break; }
                    }
                }
                for (; i < n_vocab && !found; ++i)
                    if (dec.logprobs[i] >= lmax - 1e-5f && expf(dec.logprobs[i]) == pmax) { r.id = i; r.p = pmax; r.plog = dec.logprobs[i]; found = true; }
                if (!(r.p > 0.0f)) { r.id = 0; r.p = 0.0f; r.plog = 0.0f; }     // all-zero probabilities: the reference keeps its initial {0, 0, 0}
            }
        } else {
            std::discrete_distribution<> dist(dec.probs.begin(), dec.probs.end());
            r.id = dist(dec.rng);
            r.p = dec.probs[r.id];
            r.plog = dec.logprobs[r.id];
        }
        if (r.id >= vocab.token_beg) { r.tid = r.id; r.pt = r.p; }
        return r;
    }

    std::vector<whisper_token_data> sample_token_topk(wa_decoder & dec, int k) const { // whisper.cpp:6491-6564
        // (the reference also partial-sorts logits_id here; the result is unused, so is the sort)
        int tid; float pt, ptsum;
        ts_stats(dec, tid, pt, ptsum, vocab.token_beg);
        std::discrete_distribution<> dist(dec.probs.begin(), dec.probs.end());
        std::vector<whisper_token_data> out;
        out.reserve(k);
        for (int i = 0; i < k; ++i) {
            const int id = dist(dec.rng);
            whisper_token_data t = { id, tid, dec.probs[id], dec.logprobs[id], pt, ptsum, -1, -1, -1, 0.0f };
            if (t.id >= vocab.token_beg) { t.tid = t.id; t.pt = t.p; }
            out.push_back(t);
        }
        return out;
    }

    void sequence_score(wa_sequence & seq) const {                                      // whisper.cpp:6567-6613
        if (seq.result_len == 0) return;
        double result = 0.0;
        for (int i = 0; i < seq.result_len; ++i) result += seq.tokens[i].plog;
        seq.sum_logprobs = result;
        seq.avg_logprobs = result / seq.result_len;
        double penalty = seq.result_len;
        if (p.length_penalty > 0.0f) penalty = pow((5.0 + penalty) / 6.0, p.length_penalty);
        seq.score = result / penalty;
        std::map<whisper_token, int> counts;
        int cnt = 0;
        for (int i = std::max(0, seq.result_len - 32); i < seq.result_len; ++i) { counts[seq.tokens[i].id]++; cnt++; }
        double entropy = 0.0;
        for (const auto & kv : counts) { const double q = kv.second / (double) cnt; entropy -= q * log(q); }
        seq.entropy = entropy;
    }

    void prep_batch(const whisper_token * tokens, int n, int n_past, int seq) {         // whisper.cpp:544-556
        auto & b = st->batch;
        b.n_tokens = n;
        b.token.assign(tokens, tokens + n);
        b.pos.resize(n); b.seq_id.assign(n, seq); b.logits.assign(n, 0);
        for (int i = 0; i < n; ++i) b.pos[i] = n_past + i;
        b.logits[n - 1] = 1;
    }

    int run(const float * samples, int n_samples);
};

int runner::run(const float * samples, int n_samples) {
    auto & result_all = st->result_all;
    result_all.clear();

    if (n_samples > 0) {
        if (!wa_mel_compute(*ctx, *st, samples, n_samples)) { WA_ERROR("%s: failed to compute log mel spectrogram\n", __func__); return -2; }
    }
    // language auto-detect (whisper.cpp:6815-6830)
    if (p.language == nullptr || strlen(p.language) == 0 || strcmp(p.language, "auto") == 0 || p.detect_language) {
        std::vector<float> probs(whisper_lang_max_id() + 1, 0.0f);
        const int lang_id = whisper_lang_auto_detect_with_state(ctx, st, 0, p.n_threads, probs.data());
        if (lang_id < 0) { WA_ERROR("%s: failed to auto-detect language\n", __func__); return -3; }
        st->lang_id = lang_id;
        p.language = whisper_lang_str(lang_id);
        WA_INFO("%s: auto-detected language: %s (p = %f)\n", __func__, p.language, probs[lang_id]);
        if (p.detect_language) return 0;
    }
    if (p.token_timestamps) {                       // whisper.cpp:6832-6839
        st->t_beg = 0; st->t_last = 0; st->tid_last = 0;
        if (n_samples > 0) {
            std::vector<float> host;
            const float * pcm = samples;
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, samples) == hipSuccess && attr.type == hipMemoryTypeDevice) {    // PCM handed over in HBM
                host.resize(n_samples);
                if (!WA_HIP_OK(hipMemcpy(host.data(), samples, (size_t) n_samples * sizeof(float), hipMemcpyDeviceToHost))) return -2;
                pcm = host.data();
            } else (void) hipGetLastError();
            st->energy = signal_energy(pcm, n_samples, 32);
        }
    }
    if (p.n_grammar_rules > 0) WA_WARN("%s: grammar sampling is not implemented by this backend; rules ignored\n", __func__);

    const int seek_start = p.offset_ms / 10;
    const int seek_end = p.duration_ms == 0 ? st->mel_n_len_org : seek_start + p.duration_ms / 10;
    const int delta_min = 10;                       // < 100 ms of audio is not processed (whisper.cpp:6847-6852)
    if (seek_end < seek_start + delta_min) {
        WA_WARN("%s: input is too short - %d ms < 100 ms. consider padding the input audio with silence\n", __func__, (seek_end - seek_start) * 10);
        return 0;
    }

    std::vector<float> temperatures;                // whisper.cpp:6856-6863
    if (p.temperature_inc > 0.0f) for (float t = p.temperature; t < 1.0f + 1e-6f; t += p.temperature_inc) temperatures.push_back(t);
    else temperatures.push_back(p.temperature);

    int n_decoders = 1;
    if (p.strategy == WHISPER_SAMPLING_GREEDY) n_decoders = p.greedy.best_of;
    else if (p.strategy == WHISPER_SAMPLING_BEAM_SEARCH) n_decoders = std::max(p.greedy.best_of, p.beam_search.beam_size);
    n_decoders = std::max(1, n_decoders);
    if (n_decoders > WA_MAX_DECODERS) { WA_ERROR("%s: too many decoders requested (%d), max = %d\n", __func__, n_decoders, WA_MAX_DECODERS); return -4; }
    for (int j = 1; j < n_decoders; ++j) st->decoders[j].rng = std::mt19937(j);     // whisper.cpp:6897

    auto & prompt_past = st->prompt_past;
    if (p.no_context) prompt_past.clear();
    std::vector<whisper_token> prompt_tokens;
    if (!p.prompt_tokens && p.initial_prompt) {     // whisper.cpp:6911-6921
        auto toks = wa_tokenize(vocab, p.initial_prompt);
        prompt_tokens.assign(toks.begin(), toks.end());
        p.prompt_tokens = prompt_tokens.data();
        p.prompt_n_tokens = (int) prompt_tokens.size();
    }
    if (p.prompt_tokens && p.prompt_n_tokens > 0) { // prepended to prompt_past (whisper.cpp:6924-6930)
        for (int i = 0; i < p.prompt_n_tokens; ++i) prompt_past.push_back(p.prompt_tokens[i]);
        std::rotate(prompt_past.begin(), prompt_past.end() - p.prompt_n_tokens, prompt_past.end());
    }
    if (p.audio_ctx > ctx->model.hp.n_audio_ctx) {
        WA_ERROR("%s: audio_ctx is larger than the maximum allowed (%d > %d)\n", __func__, p.audio_ctx, ctx->model.hp.n_audio_ctx);
        return -5;
    }
    st->exp_n_audio_ctx = p.audio_ctx;

    std::vector<whisper_token> prompt_init = { vocab.token_sot };     // whisper.cpp:6941-6965
    if (vocab.is_multilingual()) {
        const int lang_id = whisper_lang_id(p.language);
        st->lang_id = lang_id;
        prompt_init.push_back(vocab.token_sot + 1 + lang_id);
        prompt_init.push_back(p.translate ? vocab.token_translate : vocab.token_transcribe);
    }
    {
        const bool is_distil = ctx->model.hp.n_text_layer == 2 && ctx->model.hp.n_vocab != 51866;
        if (is_distil && !p.no_timestamps) { WA_WARN("%s: using first release distilled models - forcing no_timestamps\n", __func__); p.no_timestamps = true; }
    }
    if (p.no_timestamps) prompt_init.push_back(vocab.token_not);

    resolve_suppress_lists();

    const int n_text_ctx = ctx->model.hp.n_text_ctx;
    int seek = seek_start;
    std::vector<whisper_token> prompt;
    prompt.reserve(n_text_ctx);
    std::vector<std::vector<beam_candidate>> bc_per_dec(n_decoders);
    std::vector<beam_candidate> beam_candidates;

    while (true) {
        if (p.progress_callback) p.progress_callback(ctx, st, (100 * (seek - seek_start)) / (seek_end - seek_start), p.progress_callback_user_data);
        if (seek + delta_min >= seek_end) break;
        if (p.encoder_begin_callback && !p.encoder_begin_callback(ctx, st, p.encoder_begin_callback_user_data)) {
            WA_ERROR("%s: encoder_begin_callback returned false - aborting\n", __func__);
            break;
        }
        if (!wa_encode(*ctx, *st, seek, p.abort_callback, p.abort_callback_user_data)) { WA_ERROR("%s: failed to encode\n", __func__); return -6; }

        // a very short tail: drop the text context, it tends to make the decoder hallucinate (whisper.cpp:7012-7016)
        if (seek > seek_start && seek + 500 >= seek_end) prompt_past.clear();

        int best_decoder_id = 0;
        for (int it = 0; it < (int) temperatures.size(); ++it) {
            const float t_cur = temperatures[it];
            int n_dec = 1;                         // whisper.cpp:7023-7042
            if (p.strategy == WHISPER_SAMPLING_GREEDY) { if (t_cur > 0.0f) n_dec = p.greedy.best_of; }
            else if (p.strategy == WHISPER_SAMPLING_BEAM_SEARCH) n_dec = t_cur > 0.0f ? p.greedy.best_of : p.beam_search.beam_size;
            n_dec = std::max(1, n_dec);

            // greedy arg-max (t == 0) never draws from the distribution: it does not need all 51865 probabilities
            const bool need_full_probs = p.strategy == WHISPER_SAMPLING_BEAM_SEARCH || !(t_cur < 1e-6f);
            for (int j = 0; j < n_dec; ++j) {      // whisper.cpp:7047-7069
                auto & dec = st->decoders[j];
                dec.sequence.tokens.clear();
                dec.sequence.result_len = 0;
                dec.sequence.sum_logprobs_all = 0.0;
                dec.sequence.sum_logprobs = -INFINITY;
                dec.sequence.avg_logprobs = -INFINITY;
                dec.sequence.entropy = 0.0;
                dec.sequence.score = -INFINITY;
                dec.seek_delta = 100 * WHISPER_CHUNK_SIZE;
                dec.failed = dec.completed = dec.has_ts = false;
            }

            {   // prompt = [prev + last n_take past tokens] + sot/lang/task (whisper.cpp:7073-7085)
                prompt.clear();
                if (!prompt_past.empty() && t_cur < 0.5f && p.n_max_text_ctx > 0) {
                    const int n_take = std::min(std::min(p.n_max_text_ctx, n_text_ctx / 2), int(prompt_past.size()));
                    prompt = { vocab.token_prev };
                    prompt.insert(prompt.begin() + 1, prompt_past.end() - n_take, prompt_past.end());
                }
                prompt.insert(prompt.end(), prompt_init.begin(), prompt_init.end());

                if (st->kv_self_n_dec < n_dec) {   // grow the self KV for several decoders (whisper.cpp:7095-7113)
                    const int factor = n_dec > 1 ? n_dec + 2 : 1;
                    if (!wa_kv_self_realloc(*ctx, *st, wa_pad(n_text_ctx, 256) * factor)) { WA_ERROR("%s: KV cache reallocation failed\n", __func__); return -7; }
                    st->kv_self_n_dec = n_dec;
                }
                wa_kv_clear(st->kv_self);
                prep_batch(prompt.data(), (int) prompt.size(), 0, 0);
                if (!wa_decode(*ctx, *st, st->batch, false, p.abort_callback, p.abort_callback_user_data)) { WA_ERROR("%s: failed to decode\n", __func__); return -8; }

                {   // no-speech probability "from the first logits" (whisper.cpp:7124-7134).  Mirrored quirks of the
                    // reference: it normalises ROW 0 of state->logits - which only the flagged LAST row of a prompt decode
                    // writes, so row 0 keeps whatever an earlier call left there (zeros on a fresh state) - and takes the
                    // soft-max shift from the maximum over ALL rows, which can underflow every term and yield +inf.
                    std::vector<float> lp(n_vocab), pr(n_vocab);
                    compute_logprobs(st->logits.data(), n_vocab, lp.data(), st->logits.size());
                    compute_probs(st->logits.data(), n_vocab, lp.data(), pr.data());
                    st->no_speech_prob = pr[vocab.token_nosp];
                }
                const int64_t ts = wa_time_us();
                st->decoders[0].i_batch = (int) prompt.size() - 1;
                process_logits(st->decoders[0], t_cur, need_full_probs);
                for (int j = 1; j < n_dec; ++j) {
                    auto & dec = st->decoders[j];
                    wa_kv_seq_cp(st->kv_self, 0, j, -1, -1);
                    dec.probs = st->decoders[0].probs; dec.logits = st->decoders[0].logits; dec.logprobs = st->decoders[0].logprobs;
                }
                st->t_sample_us += wa_time_us() - ts;
            }

            // Greedy arg-max decoding with one decoder on the one-launch step: overlap the host's logit rules with the NEXT
            // decode step, which runs on the device's own prediction of the token (wa_decode.cpp wa_spec_*); every token is
            // still derived here from the logits with the reference's rules, a wrong prediction only costs a redone step.
            struct overlap_guard {
                whisper_context * c; whisper_state * s; bool on = false;
                ~overlap_guard() { if (on) wa_spec_end(*c, *s); }
            } ov { ctx, st };
            int ov_launched = 0;            // launches issued so far: L_0 .. L_{ov_launched-1}; L_k decodes the k-th sampled token
            {
                const char * e = getenv("WHISPER_AMD_NO_OVERLAP");
                const int P = (int) prompt.size();
                const auto & kvc = st->kv_self;
                const bool seq_cells = wa_kv_cell_max(kvc) == P && P < (int) kvc.size && kvc.cells[P].pos < 0;
                if (!(e && e[0] == '1') && p.strategy == WHISPER_SAMPLING_GREEDY && n_dec == 1 && t_cur < 1e-6f && !p.logits_filter_callback &&
                    ((st->mega_enabled && st->mega_pause == 0) || st->batcher) && ctx->model.n_loaded > 0 && seq_cells)      // (a lock-step member: the window runs on its group's passes)
                    ov.on = wa_spec_begin(*ctx, *st, suppress_bits());
            }

            for (int i = 0, n_max = n_text_ctx / 2 - 4; i < n_max; ++i) {
                const int64_t ts0 = wa_time_us();
                if (p.strategy == WHISPER_SAMPLING_BEAM_SEARCH) for (auto & bc : bc_per_dec) bc.clear();

                // ---- sample one token per live decoder (whisper.cpp:7169-7227) ----
                par_for(n_dec, [&](int j) {         // (each decoder: its own probabilities, RNG, sequence and candidate list)
                    auto & dec = st->decoders[j];
                    if (dec.completed || dec.failed) return;
                    if (p.strategy == WHISPER_SAMPLING_GREEDY) {
                        dec.sequence.tokens.push_back(sample_token(dec, t_cur < 1e-6f));
                        dec.sequence.sum_logprobs_all += dec.sequence.tokens.back().plog;
                    } else {
                        const int64_t trk = tr_on ? wa_time_us() : 0;
                        const auto drawn = sample_token_topk(dec, p.beam_search.beam_size);
                        if (tr_on && j == 0) tr_topk += wa_time_us() - trk;
                        for (const auto & tok : drawn) {
                            double sum_all = dec.sequence.sum_logprobs_all;
                            sum_all += tok.plog;
                            bc_per_dec[j].push_back({ j, dec.seek_delta, dec.has_ts, tok, sum_all });
                        }
                    }
                });
                const int64_t trb0 = tr_on ? wa_time_us() : 0;
                beam_candidates.clear();
                for (auto & bc : bc_per_dec) {
                    beam_candidates.insert(beam_candidates.end(), bc.begin(), bc.end());
                    if (!bc.empty()) st->n_sample += 1;
                }

                // ---- beam search: keep the best candidates, re-label KV cells (whisper.cpp:7239-7291) ----
                if (p.strategy == WHISPER_SAMPLING_BEAM_SEARCH) {
                    std::sort(beam_candidates.begin(), beam_candidates.end(), [](const beam_candidate & a, const beam_candidate & b) {
                        if (a.sum_logprobs_all != b.sum_logprobs_all) return a.sum_logprobs_all > b.sum_logprobs_all;
                        return a.decoder_idx < b.decoder_idx;
                    });
                    // two candidates spell the same sequence: the same new token behind equal token ids so far (whisper.cpp:6419-6430)
                    auto same_candidate = [&](const beam_candidate & a, const beam_candidate & b) {
                        return a.tok.id == b.tok.id && (a.decoder_idx == b.decoder_idx || same_tokens(st->decoders[a.decoder_idx].sequence, st->decoders[b.decoder_idx].sequence));
                    };
                    uint32_t cur_c = 0;
                    std::vector<wa_sequence> next(n_dec);        // built from the decoders' sequences as they stand; swapped in once all are chosen
                    for (int j = 0; j < n_dec; ++j) {
                        auto & dec = st->decoders[j];
                        if (dec.completed || dec.failed) continue;
                        if (cur_c >= beam_candidates.size()) cur_c = 0;
                        const auto & cur = beam_candidates[cur_c++];
                        while (beam_candidates.size() > cur_c && same_candidate(beam_candidates[cur_c], cur) && i > 0) ++cur_c;
                        dec.seek_delta = cur.seek_delta;
                        dec.has_ts = cur.has_ts;
                        next[j] = st->decoders[cur.decoder_idx].sequence;
                        next[j].tokens.push_back(cur.tok);
                        next[j].sum_logprobs_all = cur.sum_logprobs_all;
                        wa_kv_seq_cp(st->kv_self, cur.decoder_idx, WA_MAX_DECODERS + j, -1, -1);
                    }
                    for (int j = 0; j < n_dec; ++j) {
                        auto & dec = st->decoders[j];
                        if (dec.completed || dec.failed) continue;
                        std::swap(dec.sequence, next[j]);
                    }
                    for (int j = 0; j < n_dec; ++j) {
                        auto & dec = st->decoders[j];
                        if (dec.completed || dec.failed) continue;
                        wa_kv_seq_rm(st->kv_self, j, -1, -1);
                        wa_kv_seq_cp(st->kv_self, WA_MAX_DECODERS + j, j, -1, -1);
                        wa_kv_seq_rm(st->kv_self, WA_MAX_DECODERS + j, -1, -1);
                    }
                }

                const int64_t trb1 = tr_on ? wa_time_us() : 0;
                // ---- per-decoder state machine (whisper.cpp:7297-7379) ----
                for (int j = 0; j < n_dec; ++j) {
                    auto & dec = st->decoders[j];
                    if (dec.completed || dec.failed) continue;
                    auto & result_len = dec.sequence.result_len;
                    const auto & token = dec.sequence.tokens.back();
                    if (token.id > vocab.token_beg) {           // timestamp token: slide the window
                        const int seek_delta_new = 2 * (token.id - vocab.token_beg);
                        if (dec.has_ts && dec.seek_delta > seek_delta_new && result_len < i) { dec.failed = true; continue; }   // no going back in time
                        dec.seek_delta = seek_delta_new;
                        result_len = i + 1;
                        dec.has_ts = true;
                    }
                    if (token.id == vocab.token_eot || (p.max_tokens > 0 && i >= p.max_tokens) ||
                        (dec.has_ts && seek + dec.seek_delta + delta_min >= seek_end)) {
                        if (result_len == 0 && !p.no_timestamps) {
                            if (seek + dec.seek_delta + delta_min >= seek_end) result_len = i + 1;
                            else { dec.failed = true; continue; }
                        }
                        if (p.single_segment || p.no_timestamps) { result_len = i + 1; dec.seek_delta = 100 * WHISPER_CHUNK_SIZE; }
                        dec.completed = true;
                        continue;
                    }
                    if (ctx->model.n_loaded == 0) { dec.seek_delta = 100 * WHISPER_CHUNK_SIZE; dec.completed = true; continue; }   // test models
                    // repetition loop guard (whisper.cpp:7374-7378)
                    if (i == n_max - 1 && (result_len == 0 || dec.seek_delta < 100 * WHISPER_CHUNK_SIZE / 2)) { dec.failed = true; continue; }
                }
                {
                    bool all_done = true;
                    for (int j = 0; j < n_dec; ++j) if (!st->decoders[j].completed && !st->decoders[j].failed) all_done = false;
                    if (tr_on) { tr_beam += trb1 - trb0; tr_state += wa_time_us() - trb1; tr_steps += 1; }
                    if (all_done) break;
                }
                st->t_sample_us += wa_time_us() - ts0;

                // ---- next-token logits for every live decoder in ONE decoder pass (whisper.cpp:7403-7434) ----
                {
                    auto & b = st->batch;
                    b.n_tokens = 0; b.token.clear(); b.pos.clear(); b.seq_id.clear(); b.logits.clear();
                    const int n_past = (int) prompt.size() + i;
                    for (int j = 0; j < n_dec; ++j) {
                        auto & dec = st->decoders[j];
                        if (dec.failed || dec.completed) continue;
                        dec.i_batch = b.n_tokens;
                        b.token.push_back(dec.sequence.tokens.back().id);
                        b.pos.push_back(n_past);
                        b.seq_id.push_back(j);
                        b.logits.push_back(1);
                        b.n_tokens++;
                    }
                    bool have_logits = false;
                    if (ov.on) {
                        auto & dec = st->decoders[0];
                        const auto & toks = dec.sequence.tokens;
                        auto state_after = [&](int k) {     // sampling state after the k-th token, as process_logits will see it
                            wa_spec_state s0; s0.last = toks[k].id; s0.penult = k > 0 ? toks[k - 1].id : -1; s0.seek_delta = dec.seek_delta; s0.has_ts = dec.has_ts ? 1 : 0;
                            return s0;
                        };
                        const wa_spec_state none = { 0, -1, 0, 0 };
                        auto can_launch = [&](int k) { return k < n_max && n_past - i + k < n_text_ctx; };
                        bool ok = true;
                        if (ov_launched <= i) { ok = wa_spec_launch(*ctx, *st, i, n_past, toks[i].id, state_after(i)); ov_launched = i + 1; }
                        if (ok && ov_launched == i + 1 && can_launch(i + 1)) { ok = wa_spec_launch(*ctx, *st, i + 1, n_past + 1, -1, none); ov_launched = i + 2; }
                        int used = -1, rc = ok ? wa_spec_wait(*ctx, *st, i, &used) : -1;
                        if (rc == 0 && used != toks[i].id) {            // the device decoded another token than the rules give: redo this step
                            st->n_spec_miss++;
                            wa_spec_drain(*ctx, *st);
                            ok = wa_spec_launch(*ctx, *st, i, n_past, toks[i].id, state_after(i)); ov_launched = i + 1;
                            if (ok && can_launch(i + 1)) { ok = wa_spec_launch(*ctx, *st, i + 1, n_past + 1, -1, none); ov_launched = i + 2; }
                            rc = ok ? wa_spec_wait(*ctx, *st, i, &used) : -1;
                        } else if (rc == 0 && i > 0) st->n_spec_ok++;
                        if (rc == 0) {
                            if (!wa_kv_find_slot(st->kv_self, b) || (int) st->kv_self.head != n_past) { WA_ERROR("%s: KV cells out of step with the overlapped decode\n", __func__); return -9; }
                            st->kv_self.n = std::min(st->kv_self.size, (uint32_t) std::max(1, wa_kv_cell_max(st->kv_self)));
                            if (p.abort_callback && p.abort_callback(p.abort_callback_user_data)) { WA_ERROR("%s: failed to decode\n", __func__); return -9; }
                            have_logits = true;
                        } else if (rc == 1 && st->batcher) {
                            // this member's row of the pass wants the launch sequence (an uncertifiable soft-max sum): the step is decoded below the plain
                            // way, the window goes on behind it - the next iteration asks for step i + 1 with its token, then runs ahead again
                            wa_spec_drain(*ctx, *st);
                            ov_launched = i + 1;
                            st->solo_step = true;       // (the same inputs would send the group's pass back again)
                        } else {        // the one-launch step is not available for this token (or any more): finish the window the plain way
                            static const bool trace_w = getenv("WHISPER_AMD_BATCH_TRACE") != nullptr;
                            if (trace_w) fprintf(stderr, "[full] overlap window ends at step %d: ok %d rc %d used %d launched %d\n", i, (int) ok, rc, used, ov_launched);
                            wa_spec_drain(*ctx, *st);
                            wa_spec_end(*ctx, *st);
                            ov.on = false;
                        }
                    }
                    st->defer_rows = b.n_tokens > 1;        // (every row of this batch belongs to a live decoder, whose process_logits below fetches it)
                    const bool dec_ok = have_logits || wa_decode(*ctx, *st, b, false, p.abort_callback, p.abort_callback_user_data);
                    st->defer_rows = false;
                    if (!dec_ok) { st->staged_n = 0; WA_ERROR("%s: failed to decode\n", __func__); return -9; }
                    const int64_t ts1 = wa_time_us();
                    par_for(n_dec, [&](int j) {
                        auto & dec = st->decoders[j];
                        if (dec.failed || dec.completed) return;
                        process_logits(dec, t_cur, need_full_probs);
                    });
                    st->staged_n = 0;
                    st->t_sample_us += wa_time_us() - ts1;
                }
            }

            if (tr_on && tr_steps > 0)
                fprintf(stderr, "[sample] %ld steps, decoder 0 (us per step): rules + log-soft-max %.1f  probabilities %.1f  top-k draws %.1f  beam bookkeeping %.1f  state machine %.1f\n", (long) tr_steps,
                        (double) tr_rules / tr_steps, (double) tr_probs / tr_steps, (double) tr_topk / tr_steps, (double) tr_beam / tr_steps, (double) tr_state / tr_steps);
            {   // rank the sequences (whisper.cpp:7484-7517)
                double best_score = -INFINITY;
                for (int j = 0; j < n_dec; ++j) {
                    auto & dec = st->decoders[j];
                    if (dec.failed) continue;
                    dec.sequence.tokens.resize(dec.sequence.result_len);
                    sequence_score(dec.sequence);
                    if (dec.sequence.result_len > 32 && dec.sequence.entropy < p.entropy_thold) { dec.failed = true; st->n_fail_h++; continue; }
                    if (best_score < dec.sequence.score) { best_score = dec.sequence.score; best_decoder_id = j; }
                }
            }
            bool success = true;                    // temperature fallback (whisper.cpp:7519-7544)
            if (it != (int) temperatures.size() - 1) {
                const auto & dec = st->decoders[best_decoder_id];
                if (dec.failed || (dec.sequence.avg_logprobs < p.logprob_thold && st->no_speech_prob < p.no_speech_thold)) { success = false; st->n_fail_p++; }
            }
            if (success) break;
        }

        // ---- turn the winning token sequence into segments (whisper.cpp:7547-7707) ----
        {
            const auto & best = st->decoders[best_decoder_id];
            int seek_delta = best.seek_delta;
            const int result_len = best.sequence.result_len;
            const auto & tokens_cur = best.sequence.tokens;
            const size_t n_segments_before = result_all.size();
            const bool is_no_speech = st->no_speech_prob > p.no_speech_thold && best.sequence.avg_logprobs < p.logprob_thold;

            prompt_past.clear();
            if (prompt.front() == vocab.token_prev) prompt_past.insert(prompt_past.end(), prompt.begin() + 1, prompt.end() - prompt_init.size());
            for (int i = 0; i < result_len && !is_no_speech; ++i) prompt_past.push_back(tokens_cur[i].id);

            auto push_segment = [&](int64_t t0, int64_t t1, const std::string & text, int i0, int i1, bool speaker_turn_next) {
                if (p.print_realtime) {
                    if (p.print_timestamps) printf("[%lld --> %lld]  %s\n", (long long) t0, (long long) t1, text.c_str());
                    else { printf("%s", text.c_str()); fflush(stdout); }
                }
                wa_segment seg;
                seg.t0 = t0; seg.t1 = t1; seg.text = text; seg.no_speech_prob = st->no_speech_prob; seg.speaker_turn_next = speaker_turn_next;
                for (int j = i0; j < i1; ++j) seg.tokens.push_back(tokens_cur[j]);
                result_all.push_back(std::move(seg));
                int n_new = 1;
                if (p.token_timestamps) {                   // whisper.cpp:7619-7626
                    token_level_timestamps(ctx, st, (int) result_all.size() - 1, p.thold_pt, p.thold_ptsum);
                    if (p.max_len > 0) n_new = wrap_segment(ctx, st, p.max_len, p.split_on_word);
                }
                if (p.new_segment_callback && !ctx->params.dtw_token_timestamps) p.new_segment_callback(ctx, st, n_new, p.new_segment_callback_user_data);
            };

            if (!tokens_cur.empty() && ctx->model.n_loaded > 0 && !is_no_speech) {
                int i0 = 0;
                int64_t t0 = seek + 2 * (tokens_cur.front().tid - vocab.token_beg);
                std::string text;
                bool speaker_turn_next = false;
                for (int i = 0; i < (int) tokens_cur.size(); ++i) {
                    if (p.print_special || tokens_cur[i].id < vocab.token_eot) text += whisper_token_to_str(ctx, tokens_cur[i].id);
                    if (p.tdrz_enable && tokens_cur[i].id == vocab.token_solm) speaker_turn_next = true;
                    if (tokens_cur[i].id > vocab.token_beg && !p.single_segment) {
                        const int64_t t1 = seek + 2 * (tokens_cur[i].tid - vocab.token_beg);
                        if (!text.empty()) push_segment(t0, t1, text, i0, i + 1, speaker_turn_next);
                        text = "";
                        while (i < (int) tokens_cur.size() && tokens_cur[i].id > vocab.token_beg) i++;
                        i--;
                        t0 = t1;
                        i0 = i + 1;
                        speaker_turn_next = false;
                    }
                }
                if (!text.empty()) push_segment(t0, seek + seek_delta, text, i0, (int) tokens_cur.size(), speaker_turn_next);
            }

            {   // DTW token timestamps (whisper.cpp:7680-7692)
                const int n_segments = (int) (result_all.size() - n_segments_before);
                if (ctx->params.dtw_token_timestamps && n_segments) {
                    const int n_frames = std::min(std::min(WHISPER_CHUNK_SIZE * 100, seek_delta), seek_end - seek);
                    wa_dtw_timestamps(ctx, st, p, (int) result_all.size() - n_segments, n_segments, seek, n_frames, 7);
                    if (p.new_segment_callback)
                        for (int seg = (int) result_all.size() - n_segments; seg < n_segments; seg++)
                            p.new_segment_callback(ctx, st, seg, p.new_segment_callback_user_data);
                }
            }
            // a lone closing timestamp means "nothing more in this window" (whisper.cpp:7695-7701)
            const bool single_timestamp_ending = tokens_cur.size() > 1 && tokens_cur[tokens_cur.size() - 2].id < vocab.token_beg &&
                                                 tokens_cur[tokens_cur.size() - 1].id > vocab.token_beg;
            if (single_timestamp_ending) seek_delta = std::min(seek_end - seek, WHISPER_CHUNK_SIZE * 100);
            seek += seek_delta;
        }
    }
    return 0;
}

} // namespace

int wa_full(whisper_context * ctx, whisper_state * st, whisper_full_params params, const float * samples, int n_samples) {
    runner r(ctx, st, params);
    return r.run(samples, n_samples);
}

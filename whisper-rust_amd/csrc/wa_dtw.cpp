// wa_dtw.cpp - DTW token-level timestamps (BASELINE config 4).
//
// ref: whisper_exp_compute_token_level_timestamps_dtw whisper.cpp:8772-8933, dtw_and_backtrace 8647-8731,
// median_filter 8737-8770, alignment-head selection 1190-1303, QK capture in the decoder graph 2737-2752 / 2838-2845.
//
// Device part: one extra decoder pass over [sot, lang, not, text..., eot] whose cross-attention kernels also write
// their F32 soft-max probabilities (k_attn_exact's qk_out).  Only the alignment heads are copied back.  The rest
// (normalise over tokens, 7-wide median over audio, mean over heads, DTW + backtrace) is host post-processing on a
// [tokens x frames/2 x heads] cube, as in the reference, in the same arithmetic order (F64 sums of ggml_norm/ggml_mean).
#include "wa_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

void wa_dtw_timestamps(whisper_context * ctx, whisper_state * st, const whisper_full_params & params, int i_segment, size_t n_segments,
                       int seek, int n_frames, int medfilt_width) {
    const auto & hp = ctx->model.hp;
    const auto & vocab = ctx->vocab;
    const int n_audio_ctx = st->exp_n_audio_ctx > 0 ? st->exp_n_audio_ctx : hp.n_audio_ctx;
    const int H = hp.n_text_head;
    if (st->aheads_n <= 0 || n_frames > 2 * n_audio_ctx || medfilt_width % 2 == 0) return;
    if (!WA_HIP_OK(hipSetDevice(ctx->device))) return;

    // sot + [lang] + not + text tokens of the new segments + eot   (whisper.cpp:8800-8817)
    std::vector<whisper_token> tokens = { vocab.token_sot };
    if (vocab.is_multilingual()) {
        const int lang_id = whisper_lang_id(params.language);
        st->lang_id = lang_id;
        tokens.push_back(vocab.token_sot + 1 + lang_id);
    }
    const size_t sot_len = tokens.size();
    tokens.push_back(vocab.token_not);
    for (size_t i = i_segment; i < i_segment + n_segments; ++i)
        for (const auto & t : st->result_all[i].tokens) if (t.id < vocab.token_eot) tokens.push_back(t.id);
    tokens.push_back(vocab.token_eot);
    const int n_tokens = (int) tokens.size();
    if (n_tokens > st->dec_mpad) { WA_WARN("%s: too many tokens for DTW (%d)\n", __func__, n_tokens); return; }

    // capture buffer: [layer slot][token][head][n_audio_ctx] F32 for the layers that own alignment heads
    std::vector<int> slot(hp.n_text_layer, -1);
    int n_slots = 0;
    for (int il = 0; il < hp.n_text_layer; ++il) if (!st->aheads[il].empty()) slot[il] = n_slots++;
    const size_t per_layer = (size_t) n_tokens * H * n_audio_ctx;
    if (st->d_aheads_qk) { (void) hipFree(st->d_aheads_qk); st->d_aheads_qk = nullptr; }
    if (!WA_HIP_OK(hipMalloc((void **) &st->d_aheads_qk, per_layer * n_slots * sizeof(float)))) return;

    // the decoder pass (whisper.cpp:8823-8829)
    wa_kv_clear(st->kv_self);
    auto & b = st->batch;
    b.n_tokens = n_tokens;
    b.token.assign(tokens.begin(), tokens.end());
    b.pos.resize(n_tokens); b.seq_id.assign(n_tokens, 0); b.logits.assign(n_tokens, 0);
    for (int i = 0; i < n_tokens; ++i) b.pos[i] = i;
    b.logits[n_tokens - 1] = 1;
    wa_kv_seq_rm(st->kv_self, 0, 0, -1);
    // wa_decode indexes the capture buffer by layer; give it a slot table through aheads_slot
    st->aheads_slot = slot;
    const bool ok = wa_decode(*ctx, *st, b, true, nullptr, nullptr);
    if (!ok) { WA_ERROR("%s: decoder pass failed\n", __func__); (void) hipFree(st->d_aheads_qk); st->d_aheads_qk = nullptr; return; }

    // copy back the alignment heads only: data[t + n_tokens*(j + n_audio_ctx*k)] (layout of aheads_cross_QKs, whisper.cpp:8844-8856)
    const int n_heads = st->aheads_n;
    const int n_audio_tokens = n_frames / 2;
    if (n_audio_tokens <= 0) { (void) hipFree(st->d_aheads_qk); st->d_aheads_qk = nullptr; return; }
    std::vector<float> rows((size_t) n_tokens * n_audio_ctx);
    std::vector<float> w((size_t) n_tokens * n_audio_tokens * n_heads);     // w[t + n_tokens*(j + n_audio_tokens*k)]
    int k = 0;
    for (int il = 0; il < hp.n_text_layer; ++il) {
        for (int h : st->aheads[il]) {
            const float * src = st->d_aheads_qk + (size_t) slot[il] * per_layer + (size_t) h * n_audio_ctx;
            if (!WA_HIP_OK(hipMemcpy2D(rows.data(), (size_t) n_audio_ctx * sizeof(float), src, (size_t) H * n_audio_ctx * sizeof(float),
                                       (size_t) n_audio_ctx * sizeof(float), n_tokens, hipMemcpyDeviceToHost))) return;
            for (int j = 0; j < n_audio_tokens; ++j)
                for (int t = 0; t < n_tokens; ++t) w[t + (size_t) n_tokens * (j + (size_t) n_audio_tokens * k)] = rows[(size_t) t * n_audio_ctx + j];
            ++k;
        }
    }
    (void) hipFree(st->d_aheads_qk); st->d_aheads_qk = nullptr;

    // Per alignment head (independent: spread over a few host threads - a medium model's 32 heads x 223 tokens x 750 frames took ~0.5 s
    // in one thread with a sort per median):
    //   ggml_norm over the token axis, eps 1e-9 (whisper.cpp:8864, ops.cpp:3225-3242), in place in w[.][j][t];
    //   7-wide median over the audio axis with reflect padding, per token (whisper.cpp:8737-8770) -> med[kk][t][j].  The median of 7 is a
    //   13-exchange selection network on a transposed copy of the head (rows of frames: it vectorises); other widths sort.
    if (medfilt_width >= n_audio_tokens) { WA_WARN("%s: too few audio frames for the median filter\n", __func__); return; }
    std::vector<float> med((size_t) n_heads * n_tokens * n_audio_tokens);  // med[kk][t][j]
    auto one_head = [&](int kk) {
        for (int j = 0; j < n_audio_tokens; ++j) {
            float * x = &w[(size_t) n_tokens * (j + (size_t) n_audio_tokens * kk)];
            double sum = 0.0;
            for (int t = 0; t < n_tokens; ++t) sum += (double) x[t];
            const float mean = sum / n_tokens;
            double sum2 = 0.0;
            for (int t = 0; t < n_tokens; ++t) { const float v = x[t] - mean; x[t] = v; sum2 += (double) (v * v); }
            const float variance = sum2 / n_tokens;
            const float scale = 1.0f / sqrtf(variance + 1e-9f);
            for (int t = 0; t < n_tokens; ++t) x[t] = x[t] * scale;
        }
        const int hw = medfilt_width / 2, M_ = n_audio_tokens;
        std::vector<float> row((size_t) M_ + 2 * hw), filt((size_t) medfilt_width);
        for (int t = 0; t < n_tokens; ++t) {
            for (int j = 0; j < M_; ++j) row[hw + j] = w[t + (size_t) n_tokens * (j + (size_t) M_ * kk)];
            for (int o = 1; o <= hw; ++o) { row[hw - o] = row[hw + o]; row[hw + M_ - 1 + o] = row[hw + M_ - 1 - o]; }      // reflect (idx -> -idx, 2 (M - 1) - idx)
            float * out = &med[((size_t) kk * n_tokens + t) * M_];
            if (medfilt_width == 7) {
                const float * r = row.data();
                for (int j = 0; j < M_; ++j) {
                    float p0 = r[j], p1 = r[j + 1], p2 = r[j + 2], p3 = r[j + 3], p4 = r[j + 4], p5 = r[j + 5], p6 = r[j + 6];
#define WA_CX(a, b) do { const float lo_ = std::min(a, b), hi_ = std::max(a, b); a = lo_; b = hi_; } while (0)
                    WA_CX(p0, p5); WA_CX(p0, p3); WA_CX(p1, p6); WA_CX(p2, p4); WA_CX(p0, p1); WA_CX(p3, p5); WA_CX(p2, p6);
                    WA_CX(p2, p3); WA_CX(p3, p6); WA_CX(p4, p5); WA_CX(p1, p4); WA_CX(p1, p3); WA_CX(p3, p4);
#undef WA_CX
                    out[j] = p3;
                }
            } else {
                for (int j = 0; j < M_; ++j) {
                    std::copy(row.begin() + j, row.begin() + j + medfilt_width, filt.begin());
                    std::sort(filt.begin(), filt.end());
                    out[j] = filt[filt.size() / 2];
                }
            }
        }
    };
    {
        const int n_thr = std::max(1, std::min(n_heads, 8));
        std::vector<std::thread> th;
        for (int i = 1; i < n_thr; ++i) th.emplace_back([&, i] { for (int kk = i; kk < n_heads; kk += n_thr) one_head(kk); });
        for (int kk = 0; kk < n_heads; kk += n_thr) one_head(kk);
        for (auto & t_ : th) t_.join();
    }
    // mean over heads (F64 sum in head order, ops.cpp:2033-2041 / vec.h:908-914), times -1
    std::vector<float> cost_in((size_t) n_tokens * n_audio_tokens);          // x[t][j]
    for (int t = 0; t < n_tokens; ++t)
        for (int j = 0; j < n_audio_tokens; ++j) {
            double sum = 0.0;
            for (int kk = 0; kk < n_heads; ++kk) sum += (double) med[((size_t) kk * n_tokens + t) * n_audio_tokens + j];
            float v = (float) sum;
            v /= (float) n_heads;
            cost_in[(size_t) t * n_audio_tokens + j] = v * -1.0f;
        }

    // drop the sot sequence and eot (whisper.cpp:8880-8882): rows [sot_len, n_tokens - 1)
    const int N = n_tokens - (int) sot_len - 1, M = n_audio_tokens;
    if (N <= 0) return;
    auto X = [&](int i, int j) { return cost_in[(size_t) (i + sot_len) * n_audio_tokens + j]; };

    // DTW + backtrace (whisper.cpp:8647-8731)
    std::vector<float> cost((size_t) (N + 1) * (M + 1), INFINITY);
    std::vector<int32_t> trace((size_t) (N + 1) * (M + 1), -1);
    auto C = [&](int i, int j) -> float & { return cost[(size_t) j * (N + 1) + i]; };
    auto TR = [&](int i, int j) -> int32_t & { return trace[(size_t) j * (N + 1) + i]; };
    C(0, 0) = 0.0f;
    for (int j = 1; j < M + 1; ++j)
        for (int i = 1; i < N + 1; ++i) {
            const float c0 = C(i - 1, j - 1), c1 = C(i - 1, j), c2 = C(i, j - 1);
            float c; int32_t t;
            if (c0 < c1 && c0 < c2) { c = c0; t = 0; } else if (c1 < c0 && c1 < c2) { c = c1; t = 1; } else { c = c2; t = 2; }
            C(i, j) = X(i - 1, j - 1) + c;
            TR(i, j) = t;
        }
    for (int j = 0; j < M + 1; ++j) TR(0, j) = 2;
    for (int i = 0; i < N + 1; ++i) TR(i, 0) = 1;
    std::vector<std::pair<int32_t, int32_t>> path;       // (token index, time index), built backwards
    {
        int i = N, j = M;
        while (i > 0 || j > 0) {
            path.emplace_back(i - 1, j - 1);
            const int32_t t = TR(i, j);
            if (t == 0) { --i; --j; } else if (t == 1) { --i; } else if (t == 2) { --j; } else break;
        }
        std::reverse(path.begin(), path.end());
    }

    // place the timestamps on the text tokens of the new segments (whisper.cpp:8894-8920)
    int32_t last_v = 0;
    size_t seg_i = i_segment;
    const size_t seg_end = i_segment + n_segments;
    size_t tok_i = 0;
    auto advance_to_text = [&]() -> bool {
        while (seg_i < seg_end) {
            auto & toks = st->result_all[seg_i].tokens;
            while (tok_i < toks.size() && !(toks[tok_i].id < vocab.token_eot)) ++tok_i;
            if (tok_i < toks.size()) return true;
            ++seg_i; tok_i = 0;
        }
        return false;
    };
    for (const auto & pr : path) {
        const int32_t v = pr.first;
        if (v != last_v) {
            const int64_t timestamp = (int64_t) pr.second * 2 + seek;      // one DTW index = 20 ms
            last_v = v;
            if (!advance_to_text()) break;
            st->result_all[seg_i].tokens[tok_i].t_dtw = timestamp;
            ++tok_i;
        }
    }
}

// wa_decode.cpp - KV-cell bookkeeping and one decoder pass over a batch of tokens.
//
// ref: whisper_kv_cache_* whisper.cpp:1049-1167 (cell metadata: pos + set of sequence ids; beams share
// cells by id, no data copies), whisper_build_graph_decoder whisper.cpp:2474-2852,
// whisper_decode_internal whisper.cpp:2864-2994.
#include "wa_internal.h"
#include "wa_kernels.h"
#include "wa_mega.h"
#include "wa_rows.h"

#include <cmath>
#include <mutex>
#include <deque>
#include <condition_variable>
#include <vector>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <limits>

// -------------------------------------------------------------------------------------------------
// KV cells ("paged" cache: page = one cell)
// -------------------------------------------------------------------------------------------------
bool wa_kv_find_slot(wa_kv_cache & c, const wa_batch & b) {
    const uint32_t n_ctx = c.size, n_tokens = b.n_tokens;
    if (n_tokens > n_ctx) { WA_ERROR("%s: n_tokens=%d > n_ctx=%d\n", __func__, n_tokens, n_ctx); return false; }
    uint32_t n_tested = 0;
    while (true) {
        if (c.head + n_tokens > n_ctx) { n_tested += n_ctx - c.head; c.head = 0; continue; }
        bool found = true;
        for (uint32_t i = 0; i < n_tokens; ++i) {
            if (c.cells[c.head + i].pos >= 0) { found = false; c.head += i + 1; n_tested += i + 1; break; }
        }
        if (found) break;
        if (n_tested >= n_ctx) return false;
    }
    for (uint32_t i = 0; i < n_tokens; ++i) {
        c.cells[c.head + i].pos = b.pos[i];
        c.cells[c.head + i].seq_id.insert(b.seq_id[i]);
    }
    return true;
}

int32_t wa_kv_cell_max(const wa_kv_cache & c) {
    for (uint32_t i = c.size - 1; i > 0; --i)
        if (c.cells[i].pos >= 0 && !c.cells[i].seq_id.empty()) return i + 1;
    return 1;
}

void wa_kv_clear(wa_kv_cache & c) {
    for (auto & cell : c.cells) { cell.pos = -1; cell.seq_id.clear(); }
    c.head = 0;
    // the reference also zeroes the buffer (whisper.cpp:1118); masked cells contribute exactly 0 to the
    // softmax and our V rows are always finite, so the data need not be touched here.
}

void wa_kv_seq_rm(wa_kv_cache & c, int32_t seq, int32_t p0, int32_t p1) {
    uint32_t new_head = c.size;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = std::numeric_limits<int32_t>::max();
    for (uint32_t i = 0; i < c.size; ++i) {
        auto & cell = c.cells[i];
        if (cell.pos >= p0 && cell.pos < p1) {
            if (seq < 0) cell.seq_id.clear();
            else if (cell.has(seq)) cell.seq_id.erase(seq);
            else continue;
            if (cell.seq_id.empty()) { cell.pos = -1; if (new_head == c.size) new_head = i; }
        }
    }
    if (new_head != c.size) c.head = new_head;
}

void wa_kv_seq_cp(wa_kv_cache & c, int32_t src, int32_t dst, int32_t p0, int32_t p1) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = std::numeric_limits<int32_t>::max();
    c.head = 0;
    for (auto & cell : c.cells)
        if (cell.has(src) && cell.pos >= p0 && cell.pos < p1) cell.seq_id.insert(dst);
}

// -------------------------------------------------------------------------------------------------
// decoder pass
// -------------------------------------------------------------------------------------------------
// small-M products stream the weights once (GEMV, HBM-bound, reference summation order);
// larger M (prompt): reference-order VALU GEMM, or the MFMA GEMM when flash_attn is on
// LayerNorm followed by a projection: one fused launch for M <= 8, LayerNorm kernel + GEMM otherwise
static void ln_linear(hipStream_t s, bool exact, wa_epi_mode mode, const float * x, int d, const wa_ln & ln, float eps, wa_f16 * xn, const wa_lin & L,
                      int M, const int32_t * rows, const wa_epi & e) {
    if (M <= 8) { wa_launch_ln_gemv_exact(s, mode, x, d, rows, ln.w, ln.b, eps, L.w, L.n_in, M, L.n_out, L.n_in, e); return; }
    wa_launch_layernorm_exact(s, x, d, M, d, ln.w, ln.b, eps, xn, d, nullptr, 0);
    if (exact) wa_launch_gemm_exact(s, mode, xn, d, L.w, L.n_in, M, L.n_out, L.n_in, e);
    else       wa_launch_gemm(s, mode, xn, d, L.w, L.n_in, M, L.n_out, L.n_in, e);
}

static void linear(hipStream_t s, bool exact, wa_epi_mode mode, const wa_f16 * A, int lda, const wa_lin & L, int M, const wa_epi & e) {
    if (M <= 8)     wa_launch_gemv_exact(s, mode, A, lda, nullptr, L.w, L.n_in, M, L.n_out, L.n_in, e);   // reference order is free here
    else if (exact) wa_launch_gemm_exact(s, mode, A, lda, L.w, L.n_in, M, L.n_out, L.n_in, e);
    else            wa_launch_gemm(s, mode, A, lda, L.w, L.n_in, M, L.n_out, L.n_in, e);
}

// Pure launch sequence of one decoder pass (no host synchronisation, no KV metadata): tokens/positions/
// rows/mask are already in d_tok/d_pos/d_rows/d_mask.  `mask` may be null (every cell < n_kv visible).
// `dyn`: device {n_kv, kv_head}; when non-null the kernels read both from there (the captured graph is replayed with
// different values every token) and the scalar arguments are ignored.
// `rowp` (device, n_tokens entries): the rows are single tokens of DIFFERENT states decoded in lock step (wa_batcher below) - row m's
// key / value go to its own state's cell and its attention reads its own state's cells and encoder K / V; activations, stream and
// scratch are `st`'s (the batcher's private state); `kv_size` = cells per layer of every member state.
static void decode_launch(whisper_context & ctx, whisper_state & st, int n_tokens, int n_kv, int kv_head, const int8_t * mask, int n_rows,
                          bool save_aheads, const int * dyn = nullptr, const wa_rowptr * rowp = nullptr, uint32_t kv_size = 0) {
    const auto & m  = ctx.model;
    const auto & hp = m.hp;
    auto & kv = st.kv_self;
    const int n_vocab = hp.n_vocab;
    const int d = hp.n_text_state, H = hp.n_text_head;
    const int T = st.enc_n_ctx > 0 ? st.enc_n_ctx : (st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx);
    hipStream_t s = st.stream;

    wa_launch_dec_embed(s, st.d_tok, st.d_pos, n_tokens, d, m.d_te, m.d_pe, st.d_dx);

    const float KQscale = pow(float(64), -0.25);       // whisper.cpp:2522
    const size_t kv_layer = (size_t) (rowp ? kv_size : kv.size) * d;
    const size_t cross_layer = (size_t) H * st.cross_tpad * 64;

    for (int il = 0; il < hp.n_text_layer; ++il) {
        const auto & L = m.dec[il];
        // ---- masked self-attention ----
        {   // LayerNorm + fused q|k|v: q scaled -> d_dq ; k scaled, v -> straight into their KV cells [kv_head, kv_head + n_tokens)
            wa_epi e; e.bias = L.qkv.b; e.scale = L.qkv.s; e.out = st.d_dq; e.ldo = d;
            e.out2 = kv.k + il * kv_layer; e.ldo2 = d; e.out3 = kv.v + il * kv_layer; e.ldo3 = d;
            e.split0 = d; e.split1 = 2 * d; e.row_off = kv_head; e.dyn = dyn;
            e.rowp = rowp; e.rowp_off = (long long) (il * kv_layer);
            ln_linear(s, ctx.exact, WA_EPI_DEC_QKV, st.d_dx, d, L.attn_ln, hp.eps, st.d_dxn, L.qkv, n_tokens, nullptr, e);
        }
        wa_launch_attn_exact(s, st.d_dq, d, kv.k + il * kv_layer, 64, d, kv.v + il * kv_layer, 64, d, H, n_tokens, n_kv, mask, 1.0f,
                             st.d_att_partial, st.d_att_pleft, st.d_dao, d, nullptr, dyn, nullptr, nullptr, nullptr, rowp, 0, (long long) (il * kv_layer));
        {
            wa_epi e; e.bias = L.out.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d;
            linear(s, ctx.exact, WA_EPI_RESID, st.d_dao, d, L.out, n_tokens, e);
        }
        // ---- cross-attention over the encoder K/V ----
        {
            wa_epi e; e.bias = L.cross_q.b; e.out = st.d_dq; e.ldo = d;
            ln_linear(s, ctx.exact, WA_EPI_F16, st.d_dx, d, L.cross_ln, hp.eps, st.d_dxn, L.cross_q, n_tokens, nullptr, e);
        }
        float * qk_out = nullptr;
        if (save_aheads && st.d_aheads_qk && il < (int) st.aheads_slot.size() && st.aheads_slot[il] >= 0)
            qk_out = st.d_aheads_qk + (size_t) st.aheads_slot[il] * n_tokens * H * T;
        wa_launch_attn_exact(s, st.d_dq, d, st.d_cross_k + il * cross_layer, (size_t) st.cross_tpad * 64, 64, st.d_cross_v + il * cross_layer,
                             (size_t) st.cross_tpad * 64, 64, H, n_tokens, T, nullptr, KQscale, st.d_att_partial, st.d_att_pleft, st.d_dao, d, qk_out,
                             nullptr, nullptr, nullptr, nullptr, rowp, 1, (long long) (il * cross_layer));
        {
            wa_epi e; e.bias = L.cross_out.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d;
            linear(s, ctx.exact, WA_EPI_RESID, st.d_dao, d, L.cross_out, n_tokens, e);
        }
        // ---- feed-forward ----
        {
            wa_epi e; e.bias = L.fc1.b; e.gelu = m.d_gelu; e.out = st.d_dff; e.ldo = 4 * d;
            ln_linear(s, ctx.exact, WA_EPI_GELU_F16, st.d_dx, d, L.mlp_ln, hp.eps, st.d_dxn, L.fc1, n_tokens, nullptr, e);
        }
        {
            wa_epi e; e.bias = L.fc2.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d;
            linear(s, ctx.exact, WA_EPI_RESID, st.d_dff, 4 * d, L.fc2, n_tokens, e);
        }
    }
    // final LayerNorm + logits = token_embedding . x, for the flagged rows only (the reference computes all rows and
    // copies out the flagged ones, whisper.cpp:2835, 2965-2971)
    if (n_rows) {
        wa_epi e; e.out = st.d_logits; e.ldo = n_vocab;
        wa_launch_ln_gemv_exact(s, WA_EPI_F32, st.d_dx, d, st.d_rows, m.d_ln.w, m.d_ln.b, hp.eps, m.d_te, d, n_rows, n_vocab, d, e);
    }
}

// The same pass for a quantised model (Q5_0 / Q8_0 weights, wa_quant.hip): every product quantises its F32 operand row to Q8_0
// first (fused into the LayerNorm launch where the operand is a LayerNorm output), so attention and GELU hand over F32.
static void decode_launch_quant(whisper_context & ctx, whisper_state & st, int n_tokens, int n_kv, int kv_head, const int8_t * mask, int n_rows,
                                const int32_t * h_rows, bool save_aheads, const int * dyn = nullptr) {
    const auto & m  = ctx.model;
    const auto & hp = m.hp;
    auto & kv = st.kv_self;
    const int n_vocab = hp.n_vocab, d = hp.n_text_state, H = hp.n_text_head;
    const int T = st.enc_n_ctx > 0 ? st.enc_n_ctx : (st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx);
    hipStream_t s = st.stream;
    auto qmul = [&](wa_epi_mode mode, const wa_lin & L, int M, const wa_epi & e) {      // operand already in d_q8 / d_q8d
        wa_launch_qgemm_exact(s, mode, st.d_q8, st.d_q8d, M, L.qs, L.qd, L.n_out, L.n_in, e);
    };
    auto qlin = [&](wa_epi_mode mode, const float * A, int lda, const wa_lin & L, int M, const wa_epi & e) {
        wa_launch_quantize_q8_0(s, A, lda, M, L.n_in, st.d_q8, st.d_q8d);
        qmul(mode, L, M, e);
    };
    auto ln_q = [&](const float * x, int rows, const wa_ln & ln) {
        if (rows == 1 && d <= 2048) wa_launch_ln_q8_row(s, x, d, ln.w, ln.b, hp.eps, st.d_q8, st.d_q8d);
        else wa_launch_layernorm_exact(s, x, d, rows, d, ln.w, ln.b, hp.eps, nullptr, 0, nullptr, 0, st.d_q8, st.d_q8d);
    };
    wa_launch_dec_embed_q(s, st.d_tok, st.d_pos, n_tokens, d, m.te_q.qs, m.te_q.qd, m.d_pe, st.d_dx);
    const float KQscale = pow(float(64), -0.25);
    const size_t kv_layer = (size_t) kv.size * d, cross_layer = (size_t) H * st.cross_tpad * 64;
    // second operand buffer (upper half of d_q8 / d_q8d) for the one product whose output is quantised by its own launch
    int8_t * q8b = st.d_q8 + (size_t) st.q8_rows * 2 * d; float * q8bd = st.d_q8d + (size_t) st.q8_rows * 2 * d / 32;
    for (int il = 0; il < hp.n_text_layer; ++il) {
        const auto & L = m.dec[il];
        ln_q(st.d_dx, n_tokens, L.attn_ln);
        {
            wa_epi e; e.bias = L.qkv.b; e.scale = L.qkv.s; e.out = st.d_dq; e.ldo = d;
            e.out2 = kv.k + il * kv_layer; e.ldo2 = d; e.out3 = kv.v + il * kv_layer; e.ldo3 = d;
            e.split0 = d; e.split1 = 2 * d; e.row_off = kv_head; e.dyn = dyn;
            qmul(WA_EPI_DEC_QKV, L.qkv, n_tokens, e);
        }
        wa_launch_attn_exact(s, st.d_dq, d, kv.k + il * kv_layer, 64, d, kv.v + il * kv_layer, 64, d, H, n_tokens, n_kv, mask, 1.0f,
                             st.d_att_partial, st.d_att_pleft, st.d_dao, d, nullptr, dyn, nullptr, st.d_q8, st.d_q8d);
        { wa_epi e; e.bias = L.out.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d; qmul(WA_EPI_RESID, L.out, n_tokens, e); }
        ln_q(st.d_dx, n_tokens, L.cross_ln);
        { wa_epi e; e.bias = L.cross_q.b; e.out = st.d_dq; e.ldo = d; qmul(WA_EPI_F16, L.cross_q, n_tokens, e); }
        float * qk_out = nullptr;
        if (save_aheads && st.d_aheads_qk && il < (int) st.aheads_slot.size() && st.aheads_slot[il] >= 0)
            qk_out = st.d_aheads_qk + (size_t) st.aheads_slot[il] * n_tokens * H * T;
        wa_launch_attn_exact(s, st.d_dq, d, st.d_cross_k + il * cross_layer, (size_t) st.cross_tpad * 64, 64, st.d_cross_v + il * cross_layer,
                             (size_t) st.cross_tpad * 64, 64, H, n_tokens, T, nullptr, KQscale, st.d_att_partial, st.d_att_pleft, st.d_dao, d, qk_out,
                             nullptr, nullptr, st.d_q8, st.d_q8d);
        { wa_epi e; e.bias = L.cross_out.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d; qmul(WA_EPI_RESID, L.cross_out, n_tokens, e); }
        ln_q(st.d_dx, n_tokens, L.mlp_ln);
        wa_epi e2; e2.bias = L.fc2.b; e2.out = st.d_dx; e2.ldo = d; e2.resid = st.d_dx; e2.ldr = d;
        if (n_tokens == 1) {
            wa_launch_qgemv_gelu_q8(s, st.d_q8, st.d_q8d, L.fc1.qs, L.fc1.qd, 4 * d, d, L.fc1.b, m.d_gelu, q8b, q8bd);
            wa_launch_qgemm_exact(s, WA_EPI_RESID, q8b, q8bd, 1, L.fc2.qs, L.fc2.qd, d, 4 * d, e2);
        } else {
            { wa_epi e; e.bias = L.fc1.b; e.gelu = m.d_gelu; e.out = st.d_q32b; e.ldo = 4 * d; qmul(WA_EPI_GELU_F32, L.fc1, n_tokens, e); }
            qlin(WA_EPI_RESID, st.d_q32b, 4 * d, L.fc2, n_tokens, e2);
        }
    }
    if (n_rows) {       // final LayerNorm of the flagged rows (packed), then the logits product over the quantised token embedding
        const int nb = d >> 5;
        for (int i = 0; i < n_rows; ++i)
            wa_launch_layernorm_exact(s, st.d_dx + (size_t) h_rows[i] * d, d, 1, d, m.d_ln.w, m.d_ln.b, hp.eps, nullptr, 0, nullptr, 0,
                                      st.d_q8 + (size_t) i * d, st.d_q8d + (size_t) i * nb);
        wa_epi e; e.out = st.d_logits; e.ldo = n_vocab;
        qmul(WA_EPI_F32, m.te_q, n_rows, e);
    }
}

// -------------------------------------------------------------------------------------------------
// the single-token pass as ONE launch (wa_mega.hip).  Only one such launch may be in flight per device: its
// workgroups wait for each other, so two of them interleaved by the dispatcher could each hold CUs the other needs.
// Returns 1 = done (logits in h_logits_pinned), 0 = not applicable / gave up (caller runs the launch sequence).
// -------------------------------------------------------------------------------------------------
// One slot per DEVICE (two devices never wait for each other).  Ownership rules: mega_step and the probes hold it for one launch +
// synchronise; the host-overlap window (wa_spec_begin .. wa_spec_end) holds it while its launches are in flight and records that in
// st.spec_owner, so that wa_spec_end releases exactly what wa_spec_begin took - also on every failure path (a begin that fails after
// taking the slot gives it back itself).  While a window owns the slot, other states' single-token steps on that device wait; steps
// that do not need the slot (several tokens, masks, beams: the launch sequence) run concurrently on their own streams.
static std::mutex & mega_slot(int device) { static std::mutex m[64]; return m[(unsigned) device & 63u]; }
// (tests) every cross soft-max total of the several-rows kernel through its in-order path
static int rows_force_inorder() { static const int v = getenv("WHISPER_AMD_ROWS_FORCE_INORDER") != nullptr ? 1 : 0; return v; }

// a one-launch form gave up at a hand-off: pause it (the launch sequence serves meanwhile), try again later, give up for good after 8 time-outs
static void one_launch_timeout(int & pause, int & timeouts, bool & enabled, const char * what, unsigned status) {
    timeouts += 1;
    if (timeouts > 8) { enabled = false; WA_WARN("%s: gave up at hand-off %u for the 9th time - using the launch sequence from now on\n", what, status); return; }
    pause = 32 << std::min(timeouts, 6);
    WA_WARN("%s: gave up at hand-off %u - the launch sequence serves the next %d decoder passes, then it is tried again\n", what, status, pause);
}

static bool mega_args(whisper_context & ctx, whisper_state & st, wa_mega_args & a, int token, int pos, int n_kv, int kv_head) {
    const auto & m = ctx.model;
    const auto & hp = m.hp;
    const int T = st.enc_n_ctx > 0 ? st.enc_n_ctx : (st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx);
    if (!st.mega_enabled || st.mega_pause > 0 || n_kv > WA_MEGA_KV_ROOM || n_kv < 1 || kv_head < 0 || kv_head >= n_kv || T < 1 || (T >> 5) > 47 || T > st.cross_tpad) return false;
    a.layers = (const wa_mega_layer *) m.d_mega_layers;
    a.n_layer = hp.n_text_layer; a.d = hp.n_text_state; a.n_head = hp.n_text_head; a.n_vocab = hp.n_vocab; a.eps = hp.eps; a.rn_d = 1.0 / (double) hp.n_text_state;
    a.te = m.d_te; a.pe = m.d_pe; a.lnf_w = m.d_ln.w; a.lnf_b = m.d_ln.b; a.gelu = m.d_gelu;
    a.quant = m.wtype != 1 ? 1 : 0; a.te_d = nullptr;
    if (a.quant) { a.te = (const wa_f16 *) m.te_q.qs; a.te_d = m.te_q.qd; }
    a.kv_k = st.kv_self.k; a.kv_v = st.kv_self.v; a.kv_layer_stride = (unsigned long long) st.kv_self.size * hp.n_text_state;
    a.cross_k = st.d_cross_k; a.cross_v = st.d_cross_v;
    a.cross_layer_stride = (unsigned long long) hp.n_text_head * st.cross_tpad * 64; a.cross_tpad = st.cross_tpad; a.T = T;
    a.granules = st.d_mega_gr; a.edge_stride = 4 * hp.n_text_state; a.cross_gr = st.d_mega_cgr;      // (4 d: the quantised form hands the MLP's hidden row over in F32)
    a.logits = st.d_mega_out; a.status = st.d_mega_status; a.dbg = nullptr;
    a.token = token; a.pos = pos; a.n_kv = n_kv; a.kv_head = kv_head;
    a.spec = 0; a.rec_in = st.d_mega_rec[1]; a.rec_out = st.d_mega_rec[0]; a.n_rec = std::min(m.n_cu, 256);
    a.ps_in = st.d_mega_ps[1]; a.ps_out = st.d_mega_ps[0]; a.smask = st.d_mega_smask;
    a.token_beg = ctx.vocab.token_beg; a.token_eot = ctx.vocab.token_eot;
    a.s_last = token; a.s_penult = -1; a.s_seek_delta = 0; a.s_has_ts = 0;
    a.kq_scale = pow(float(64), -0.25);       // whisper.cpp:2522
    st.mega_seq += 1; if (st.mega_seq == 0) st.mega_seq = 1;
    a.seq = st.mega_seq;
    return true;
}

static int mega_step(whisper_context & ctx, whisper_state & st, int token, int pos, int n_kv, int kv_head) {
    wa_mega_args a;
    if (!mega_args(ctx, st, a, token, pos, n_kv, kv_head)) return 0;
    const int n_vocab = ctx.model.hp.n_vocab;
    hipStream_t s = st.stream;
    unsigned status = 0, echo = 0;
    {
        // (a member of a lock-step group never WAITS for the slot: its group may hold it for the others' run-ahead windows, and they wait for this member)
        std::unique_lock<std::mutex> lk(mega_slot(ctx.device), std::defer_lock);
        if (st.batcher) { if (!lk.try_lock()) return 0; } else lk.lock();
        if (!wa_launch_decode_mega(s, a, std::min(ctx.model.n_cu, 256))) { st.mega_enabled = false; return 0; }
        (void) hipMemcpyAsync(st.h_logits_pinned, st.d_mega_out, ((size_t) n_vocab + 3) * sizeof(float), hipMemcpyDeviceToHost, s);
        if (!WA_HIP_OK(hipStreamSynchronize(s))) { st.mega_enabled = false; return 0; }
        status = ((const unsigned *) st.h_logits_pinned)[n_vocab];
        echo = ((const unsigned *) st.h_logits_pinned)[n_vocab + 2];
    }
    if (status == 0 && echo == a.seq) return 1;
    (void) hipMemsetAsync(st.d_mega_status, 0, sizeof(unsigned), s);
    if (status == WA_MEGA_REDO) return 0;       // an uncertifiable soft-max sum (~1e-9 per soft-max): this token goes through the launch sequence
    // a hand-off timed out (workgroups not co-resident?), or the launch never ran (no echo of its number): the logits are not this step's
    one_launch_timeout(st.mega_pause, st.mega_timeouts, st.mega_enabled, "one-launch decode step", status);
    return 0;
}

// -------------------------------------------------------------------------------------------------
// 2..8 token rows as ONE launch (wa_rows.hip): the rows of a beam / best_of step (one state, cells shared by sequence id: per-row masks)
// or single tokens of different states decoded in lock step (wa_batcher: per-row K / V).  `bst` owns granules, logits buffer and stream.
// Returns 1 = done (logits rows in bst.h_logits_pinned), 0 = not applicable / gave up for this pass (caller runs the launch sequence).
// The slot is taken with try_lock by default: a step that finds another one-launch pass on the device just takes the launch sequence.
// -------------------------------------------------------------------------------------------------
static int rows_step(whisper_context & ctx, whisper_state & bst, int B, const wa_rows_row * rows, int n_out, const int32_t * out_row, int T, int cross_tpad,
                     uint32_t kv_size, bool wait_slot) {
    const auto & m = ctx.model;
    const auto & hp = m.hp;
    if (bst.rows_pause > 0) { bst.rows_pause -= 1; return 0; }
    if (B < 1 || B > WA_ROWS_MAX || n_out < 1 || n_out > B || !bst.rows_enabled || T < 1 || (T >> 5) > 47 || T > cross_tpad) return 0;
    for (int i = 0; i < B; ++i)
        if (rows[i].n_kv < 1 || rows[i].n_kv > WA_ROWS_MAXKV || rows[i].kv_head < 0 || rows[i].kv_head >= rows[i].n_kv) return 0;
    const int n_wg = std::min(m.n_cu, 256);
    wa_rows_args a;
    memset(&a, 0, sizeof(a)); a.force_inorder = rows_force_inorder();
    const int quant = m.wtype != 1 ? 1 : 0;
    if (wa_rows_lds_bytes(hp.n_text_state, B, n_wg, quant, &a.slot_bytes) == 0) return 0;
    if (!wa_rows_prepare(ctx, bst)) return 0;
    a.layers = (const wa_mega_layer *) m.d_mega_layers;
    a.n_layer = hp.n_text_layer; a.d = hp.n_text_state; a.n_head = hp.n_text_head; a.n_vocab = hp.n_vocab; a.eps = hp.eps; a.rn_d = 1.0 / (double) hp.n_text_state;
    a.te = m.d_te; a.pe = m.d_pe; a.lnf_w = m.d_ln.w; a.lnf_b = m.d_ln.b; a.gelu = m.d_gelu; a.te_d = nullptr; a.quant = quant;
    if (quant) { a.te = (const wa_f16 *) m.te_q.qs; a.te_d = m.te_q.qd; }
    a.kv_layer_stride = (unsigned long long) kv_size * hp.n_text_state;
    a.cross_layer_stride = (unsigned long long) hp.n_text_head * cross_tpad * 64; a.cross_tpad = cross_tpad; a.T = T;
    a.granules = bst.d_rows_gr; a.row_gr = 2 * hp.n_text_state; a.cross_gr = bst.d_rows_cgr;
    a.logits = bst.d_logits; a.status = bst.d_rows_status; a.row_status = bst.d_rows_status + 4; a.dbg = nullptr;
    a.kq_scale = pow(float(64), -0.25);
    a.B = B;
    for (int i = 0; i < B; ++i) a.rows[i] = rows[i];
    a.n_out = n_out;
    for (int i = 0; i < n_out; ++i) a.out_row[i] = out_row ? out_row[i] : i;
    bst.mega_seq += 1; if (bst.mega_seq == 0) bst.mega_seq = 1;
    a.seq = bst.mega_seq;
    hipStream_t s = bst.stream;
    unsigned * h_status = (unsigned *) (bst.h_logits_pinned + (size_t) WA_MAX_DECODERS * hp.n_vocab);      // (the staging buffer has 64 spare words)
    std::unique_lock<std::mutex> lk(mega_slot(ctx.device), std::defer_lock);
    if (wait_slot) lk.lock(); else if (!lk.try_lock()) return 0;
    // the logits rows go straight into the pinned staging rows (whole-line stores, wa_rows.hip: mb_logits_out); WHISPER_AMD_ROWS_HOST_OUT=0: device rows + a copy
    static const bool host_out_env = getenv("WHISPER_AMD_ROWS_HOST_OUT") == nullptr || atoi(getenv("WHISPER_AMD_ROWS_HOST_OUT")) != 0;
    float * h_dev = nullptr;
    const bool host_out = host_out_env && hipHostGetDevicePointer((void **) &h_dev, bst.h_logits_pinned, 0) == hipSuccess && h_dev;
    if (host_out) a.logits = h_dev;
    (void) hipMemsetAsync(bst.d_rows_status + 4, 0, WA_ROWS_MAX * sizeof(unsigned), s);
    if (!wa_launch_decode_rows(s, a, n_wg)) { bst.rows_enabled = false; return 0; }
    if (!host_out) (void) hipMemcpyAsync(bst.h_logits_pinned, bst.d_logits, (size_t) n_out * hp.n_vocab * sizeof(float), hipMemcpyDeviceToHost, s);
    (void) hipMemcpyAsync(h_status, bst.d_rows_status, 12 * sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (!WA_HIP_OK(hipStreamSynchronize(s))) { bst.rows_enabled = false; return 0; }
    lk.unlock();
    unsigned status = h_status[0]; const unsigned echo = h_status[1];
    for (int i = 0; i < B && status == 0; ++i) if (h_status[4 + i] != 0) status = WA_MEGA_REDO;      // (one caller for all rows here: any row to be redone sends the pass back)
    if (status == 0 && echo == a.seq) { bst.n_rows_steps += 1; return 1; }
    bst.n_rows_fallback += 1;
    (void) hipMemsetAsync(bst.d_rows_status, 0, sizeof(unsigned), s);
    if (status == WA_MEGA_REDO) return 0;          // an uncertifiable soft-max sum (~1e-9 per soft-max): this pass goes through the launch sequence
    one_launch_timeout(bst.rows_pause, bst.rows_timeouts, bst.rows_enabled, "several-rows one-launch step", status);
    return 0;
}

// -------------------------------------------------------------------------------------------------
// host overlap for greedy decoding (wa_full.cpp).  The host's per-token work - the reference's logit rules, an ORDERED F32
// log-sum-exp over 51 865 logits, libm exp for the timestamp range (whisper.cpp:6149-6489) - costs about a third of a
// decode step and cannot start before the step's logits exist.  So the device predicts the token itself (wa_mega.hip:
// candidate records under the same rules, merged by the next launch) and decodes it right away; the host derives the
// token the exact way from the logits meanwhile and compares.  A wrong prediction (rare) costs one step: the launches
// in flight are discarded and the step is redone with the right token.  Launch k writes output / record / state buffer
// k & 1 and reads buffer (k - 1) & 1; at most launches k and k + 1 are in flight while the host works on logits k - 1...k.
// -------------------------------------------------------------------------------------------------
static bool bspec_begin(wa_batcher & b, whisper_state & st, const std::vector<uint32_t> & bits);
static bool bspec_launch(wa_batcher & b, whisper_state & st, int k, int pos, int token, const wa_spec_state & after);
static int  bspec_wait(wa_batcher & b, whisper_state & st, int * token_used);
static void bspec_drain(wa_batcher & b, whisper_state & st);
static void bspec_end(wa_batcher & b, whisper_state & st);

bool wa_spec_begin(whisper_context & ctx, whisper_state & st, const std::vector<uint32_t> & bits) {
    if (st.batcher) return bspec_begin(*st.batcher, st, bits);        // a member of a lock-step group: its window runs on the group's passes
    if (!st.mega_enabled || st.mega_pause > 0 || bits.size() > (size_t) ctx.model.hp.n_vocab / 32 + 2) return false;
    if (!WA_HIP_OK(hipSetDevice(ctx.device))) return false;
    if (!st.copy_stream) {      // first use on this state
        for (int b = 0; b < 2; ++b) {
            if (!WA_HIP_OK(hipHostMalloc((void **) &st.h_spec[b], ((size_t) ctx.model.hp.n_vocab + 16) * sizeof(float)))) return false;
            if (!WA_HIP_OK(hipEventCreateWithFlags(&st.ev_k[b], hipEventDisableTiming)) || !WA_HIP_OK(hipEventCreateWithFlags(&st.ev_c[b], hipEventDisableTiming))) return false;
        }
        if (!WA_HIP_OK(hipStreamCreateWithFlags(&st.copy_stream, hipStreamNonBlocking))) { st.copy_stream = nullptr; return false; }
    }
    std::vector<uint32_t> sab;
    if (const char * e = getenv("WHISPER_AMD_OVERLAP_SABOTAGE")) {      // tests: make the device's prediction wrong on purpose (every
        if (e[0] == '1') {                                              // 3rd token id is hidden from it) - results must not change
            sab = bits;
            for (size_t i = 0; i < sab.size(); ++i) sab[i] |= 0x49249249u << (i % 3);
        }
    }
    const std::vector<uint32_t> & up = sab.empty() ? bits : sab;
    (void) hipMemcpyAsync(st.d_mega_smask, up.data(), up.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st.stream);
    if (!WA_HIP_OK(hipStreamSynchronize(st.stream))) return false;
    mega_slot(ctx.device).lock();       // last: nothing below can fail, so a false return never leaves the slot taken
    st.spec_owner = true;
    return true;
}

void wa_spec_end(whisper_context & ctx, whisper_state & st) {
    if (st.batcher) { bspec_end(*st.batcher, st); return; }
    (void) hipStreamSynchronize(st.stream);
    if (st.copy_stream) (void) hipStreamSynchronize(st.copy_stream);
    (void) hipMemsetAsync(st.d_mega_smask, 0, ((size_t) st.ctx->model.hp.n_vocab / 32 + 2) * sizeof(uint32_t), st.stream);   // plain steps use no mask
    (void) hipStreamSynchronize(st.stream);
    if (st.spec_owner) { st.spec_owner = false; mega_slot(ctx.device).unlock(); }      // idempotent: only the owner gives the slot back
}

// the window's launch for a state whose single-token step is the several-rows kernel with one row (single_via_rows): same records, sampling state and
// suppression bits as the k_decode_mega form; behind the logits: [n_vocab] status, [+1] the launch's sequence number, [+2] the token decoded, [+4] the row's status
static bool spec_launch_rows(whisper_context & ctx, whisper_state & st, int k, int pos, int token, const wa_spec_state & after) {
    const auto & m = ctx.model; const auto & hp = m.hp;
    const int T = st.enc_n_ctx > 0 ? st.enc_n_ctx : (st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx);
    const int n_wg = std::min(m.n_cu, 256), quant = m.wtype != 1 ? 1 : 0, n_kv = pos + 1;
    if (!st.rows_enabled || st.rows_pause > 0 || n_kv > WA_ROWS_MAXKV || T < 1 || (T >> 5) > 47 || T > st.cross_tpad) return false;
    wa_rows_args a;
    memset(&a, 0, sizeof(a)); a.force_inorder = rows_force_inorder();
    if (wa_rows_lds_bytes(hp.n_text_state, 1, n_wg, quant, &a.slot_bytes) == 0 || !wa_rows_prepare(ctx, st)) return false;
    const int b = k & 1;
    a.layers = (const wa_mega_layer *) m.d_mega_layers;
    a.n_layer = hp.n_text_layer; a.d = hp.n_text_state; a.n_head = hp.n_text_head; a.n_vocab = hp.n_vocab; a.eps = hp.eps; a.rn_d = 1.0 / (double) hp.n_text_state;
    a.te = m.d_te; a.pe = m.d_pe; a.lnf_w = m.d_ln.w; a.lnf_b = m.d_ln.b; a.gelu = m.d_gelu; a.quant = quant;
    if (quant) { a.te = (const wa_f16 *) m.te_q.qs; a.te_d = m.te_q.qd; }
    a.kv_layer_stride = (unsigned long long) st.kv_self.size * hp.n_text_state;
    a.cross_layer_stride = (unsigned long long) hp.n_text_head * st.cross_tpad * 64; a.cross_tpad = st.cross_tpad; a.T = T;
    a.granules = st.d_rows_gr; a.row_gr = 2 * hp.n_text_state; a.cross_gr = st.d_rows_cgr;
    a.logits = b ? st.d_mega_out2 : st.d_mega_out;
    a.status = (unsigned *) (a.logits + hp.n_vocab); a.tok_out = (int *) (a.logits + hp.n_vocab + 2); a.row_status = (unsigned *) (a.logits + hp.n_vocab + 4);
    a.kq_scale = pow(float(64), -0.25); a.B = 1; a.n_out = 1; a.out_row[0] = 0;
    a.token_beg = ctx.vocab.token_beg; a.token_eot = ctx.vocab.token_eot;
    wa_rows_row & r = a.rows[0];
    r.kv_k = st.kv_self.k; r.kv_v = st.kv_self.v; r.cross_k = st.d_cross_k; r.cross_v = st.d_cross_v; r.mask = nullptr;
    r.n_kv = n_kv; r.kv_head = pos; r.token = token < 0 ? 0 : token; r.pos = pos;       // greedy steady state: cell == position
    r.spec = token < 0 ? 1 : 0;
    r.rec_in = st.d_mega_rec[b ^ 1]; r.rec_out = st.d_mega_rec[b]; r.ps_in = (const int *) st.d_mega_ps[b ^ 1]; r.ps_out = (int *) st.d_mega_ps[b]; r.smask = st.d_mega_smask;
    r.s_last = after.last; r.s_penult = after.penult; r.s_seek_delta = after.seek_delta; r.s_has_ts = after.has_ts;
    st.mega_seq += 1; if (st.mega_seq == 0) st.mega_seq = 1;
    a.seq = st.mega_seq;
    (void) hipMemsetAsync(a.row_status, 0, sizeof(unsigned), st.stream);
    if (!wa_launch_decode_rows(st.stream, a, n_wg)) { st.rows_enabled = false; st.single_via_rows = false; return false; }
    st.spec_seq[b] = a.seq;
    // (on the launch's own stream: beside the next launch of THIS kernel a copy on another stream gets no CU until that launch has finished - wa_batcher)
    (void) hipMemcpyAsync(st.h_spec[b], a.logits, ((size_t) hp.n_vocab + 8) * sizeof(float), hipMemcpyDeviceToHost, st.stream);
    return WA_HIP_OK(hipEventRecord(st.ev_c[b], st.stream));
}

bool wa_spec_launch(whisper_context & ctx, whisper_state & st, int k, int pos, int token, const wa_spec_state & after) {
    if (st.batcher) return bspec_launch(*st.batcher, st, k, pos, token, after);
    if (st.single_via_rows) return spec_launch_rows(ctx, st, k, pos, token, after);
    wa_mega_args a;
    if (!mega_args(ctx, st, a, token < 0 ? 0 : token, pos, pos + 1, pos)) return false;       // greedy steady state: cell == position
    const int b = k & 1;
    a.logits = b ? st.d_mega_out2 : st.d_mega_out;
    a.status = (unsigned *) (a.logits + ctx.model.hp.n_vocab);
    a.spec = token < 0 ? 1 : 0;
    a.rec_in = st.d_mega_rec[b ^ 1]; a.rec_out = st.d_mega_rec[b];
    a.ps_in = st.d_mega_ps[b ^ 1];   a.ps_out = st.d_mega_ps[b];
    a.s_last = after.last; a.s_penult = after.penult; a.s_seek_delta = after.seek_delta; a.s_has_ts = after.has_ts;
    if (!wa_launch_decode_mega(st.stream, a, a.n_rec)) return false;
    st.spec_seq[b] = a.seq;
    (void) hipEventRecord(st.ev_k[b], st.stream);
    (void) hipStreamWaitEvent(st.copy_stream, st.ev_k[b], 0);
    (void) hipMemcpyAsync(st.h_spec[b], a.logits, ((size_t) ctx.model.hp.n_vocab + 3) * sizeof(float), hipMemcpyDeviceToHost, st.copy_stream);
    return WA_HIP_OK(hipEventRecord(st.ev_c[b], st.copy_stream));
}

int wa_spec_wait(whisper_context & ctx, whisper_state & st, int k, int * token_used) {
    const int64_t t0 = wa_time_us();
    if (st.batcher) {
        const int rc = bspec_wait(*st.batcher, st, token_used);
        st.t_decode_us += wa_time_us() - t0; st.n_decode++;
        return rc;
    }
    const int b = k & 1, n_vocab = ctx.model.hp.n_vocab;
    if (!WA_HIP_OK(hipEventSynchronize(st.ev_c[b]))) { st.mega_enabled = false; return -1; }
    unsigned status = ((const unsigned *) st.h_spec[b])[n_vocab];
    const int i_echo = st.single_via_rows ? 1 : 2, i_tok = st.single_via_rows ? 2 : 1;          // (spec_launch_rows: the words behind the logits)
    if (status == 0 && ((const unsigned *) st.h_spec[b])[n_vocab + i_echo] != st.spec_seq[b]) status = 9999u;      // the launch never ran: not this step's logits
    if (status == 0 && st.single_via_rows && ((const unsigned *) st.h_spec[b])[n_vocab + 4] != 0) status = WA_MEGA_REDO;
    if (token_used) *token_used = ((const int *) st.h_spec[b])[n_vocab + i_tok];
    st.t_decode_us += wa_time_us() - t0; st.n_decode++;
    if (status != 0) {
        wa_spec_drain(ctx, st);
        (void) hipMemsetAsync(st.d_mega_out + n_vocab, 0, sizeof(unsigned), st.stream);
        (void) hipMemsetAsync(st.d_mega_out2 + n_vocab, 0, sizeof(unsigned), st.stream);
        (void) hipStreamSynchronize(st.stream);
        if (status == WA_MEGA_REDO) return 1;
        one_launch_timeout(st.mega_pause, st.mega_timeouts, st.mega_enabled, "one-launch decode step", status);
        return -1;
    }
    st.logits.resize(n_vocab);
    memcpy(st.logits.data(), st.h_spec[b], (size_t) n_vocab * sizeof(float));
    return 0;
}

void wa_spec_drain(whisper_context &, whisper_state & st) {
    if (st.batcher) { bspec_drain(*st.batcher, st); return; }
    (void) hipStreamSynchronize(st.stream);
    (void) hipStreamSynchronize(st.copy_stream);
}

// -------------------------------------------------------------------------------------------------
// Lock-step batched decode of several independent chunks on one device (whisper_amd_full_batch; the reference's model is one state
// + thread per chunk, whisper.cpp:7771-7806, its bench a 5-token batched decode, examples/bench/bench.cpp:110-123).  Every chunk
// keeps running the ordinary whisper_full loop on its own host thread; where that loop asks for a plain single-token step, the
// request goes to the batcher instead.  When every member still decoding has a request in, ONE decoder pass serves them all: each
// weight row is read once for B tokens (bytes per step = W + B (KVx + KVs)) - the rows are the members' tokens, each
// attending over its own state's cells and encoder K / V (wa_rowptr) - and every member gets its logits row back.  Nothing else
// changes: sampling, KV bookkeeping and results are each member's own, bit-identical to a solo run (same kernels, same order).
// -------------------------------------------------------------------------------------------------
// A member's request for one decoder step.  Plain: (token, pos, cells) given, the member waits for its logits at once.  Run-ahead (greedy members,
// wa_spec_* below): step k of the member's window at position pos = cell pos, its token either given or - token < 0 - picked by the device from the
// candidate records the member's previous pass left (wa_rows.hip: mb_pick); the member asks for step k + 1 before it waits for step k, so pass k + 1
// runs while the members apply the reference's sampling rules to the logits of pass k.
struct wa_breq {
    int k = 0, token = 0, pos = 0, n_kv = 0, kv_head = 0;
    bool ahead = false;                             // a run-ahead request (leaves records; wa_breq::token < 0: picks its token)
    wa_spec_state after = { 0, -1, 0, 0 };
    long pass = -1; int row = -1;                   // once launched: the pass and its row
    int result = 0;                                 // 0 pending, 1 launched / served, -1 not served (the member decodes alone)
};
struct wa_bslot {                                   // a member of the group
    whisper_state * st = nullptr;
    std::deque<wa_breq> q;                          // requests not yet collected, in step order (<= 2)
    unsigned * d_rec[2] = { nullptr, nullptr }; int * d_ps[2] = { nullptr, nullptr }; unsigned * d_smask = nullptr;      // run-ahead: by the parity of the member's step
    bool ahead = false;                             // inside a run-ahead window
};
struct wa_bpass { unsigned seq = 0; int B = 0, parity = 0; bool launched = false, done = false, ok = false; unsigned status = 0; };

struct wa_batcher {
    whisper_context * ctx = nullptr;
    whisper_state * bst = nullptr;                  // private state: activations, scratch and stream of the batched pass
    wa_rowptr * h_rowp = nullptr, * d_rowp = nullptr;
    std::mutex m;
    std::condition_variable cv;
    int n_members = 0;                              // threads that may still submit
    wa_bslot slots[WA_MAX_DECODERS];
    long n_steps = 0, n_rows = 0, n_one_launch = 0;       // passes, the token rows they served, passes that were ONE launch (wa_rows.hip)
    int64_t t_pass_us = 0, t_gap_us = 0, t_last_end = 0, t_created = 0, t_first = 0;   // (WHISPER_AMD_BATCH_TRACE) time inside the synchronous passes / between them
    // asynchronous passes (the one-launch form): two sets of output buffers, alternating; pass n may overwrite set n & 1 because every member
    // has collected pass n - 2 before it asks for the step that pass n serves
    float * d_out[2] = { nullptr, nullptr }, * h_out[2] = { nullptr, nullptr };       // [8][n_vocab] logits
    unsigned * d_stat[2] = { nullptr, nullptr }, * h_stat[2] = { nullptr, nullptr };  // {status, sequence echo}, then the tokens the rows decoded [8]
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_k[2] = { nullptr, nullptr }, ev_c[2] = { nullptr, nullptr };
    long n_launched = 0, n_ahead = 0, n_picks = 0;       // passes launched; run-ahead rows, rows whose token the device picked
    wa_bpass passes[4];
    bool slot_held = false;                         // this group holds the device's one-launch slot (while passes are in flight / members run ahead)
    bool async_ok = true;
    // the pass for B rows as a hipGraph (122 launches for ggml-small): captured on first use, replayed while T and the cell count stay
    hipGraphExec_t graph[WA_MAX_DECODERS + 1] = {};
    int graph_T[WA_MAX_DECODERS + 1] = {}; uint32_t graph_kv[WA_MAX_DECODERS + 1] = {};
    bool graphs_ok = true;
};

// Batchers (a private state - activations, granules, stream - and the captured graphs) are kept on the context between calls: building one
// allocates a whole whisper_state, and its hipMalloc / hipFree calls synchronise the device under the other chunks' encoders.  wa_batcher_create
// takes a free one (or makes one), wa_batcher_destroy hands it back; whisper_free releases them (wa_batcher_free_all).
static std::mutex & batcher_cache_mutex() { static std::mutex m; return m; }
static void batcher_reset(wa_batcher & b, int n_members) {
    b.n_members = n_members; b.n_steps = b.n_rows = b.n_one_launch = 0; b.t_pass_us = b.t_gap_us = b.t_last_end = 0; b.t_created = wa_time_us(); b.t_first = 0; b.n_ahead = b.n_picks = 0;
    for (auto & sl : b.slots) { sl.st = nullptr; sl.q.clear(); sl.ahead = false; }
}
wa_batcher * wa_batcher_create(whisper_context & ctx, int n_members) {
    if (ctx.model.n_loaded == 0 || n_members < 2) return nullptr;
    wa_batcher * b = nullptr;
    {
        std::lock_guard<std::mutex> lk(batcher_cache_mutex());
        if (!ctx.batcher_cache.empty()) { b = (wa_batcher *) ctx.batcher_cache.back(); ctx.batcher_cache.pop_back(); }
    }
    if (b) { batcher_reset(*b, n_members); return b; }
    b = new wa_batcher();
    b->ctx = &ctx;
    batcher_reset(*b, n_members);
    b->bst = whisper_init_state(&ctx);
    if (!b->bst || !WA_HIP_OK(hipHostMalloc((void **) &b->h_rowp, WA_MAX_DECODERS * sizeof(wa_rowptr))) ||
        !WA_HIP_OK(hipMalloc((void **) &b->d_rowp, WA_MAX_DECODERS * sizeof(wa_rowptr)))) { wa_batcher_release(b); return nullptr; }
    return b;
}
void wa_batcher_release(wa_batcher * b) {
    if (!b) return;
    for (auto & g : b->graph) if (g) (void) hipGraphExecDestroy(g);
    for (int i = 0; i < 2; ++i) {
        if (b->d_out[i]) (void) hipFree(b->d_out[i]);
        if (b->h_out[i]) (void) hipHostFree(b->h_out[i]);
        if (b->d_stat[i]) (void) hipFree(b->d_stat[i]);
        if (b->h_stat[i]) (void) hipHostFree(b->h_stat[i]);
        if (b->ev_k[i]) (void) hipEventDestroy(b->ev_k[i]);
        if (b->ev_c[i]) (void) hipEventDestroy(b->ev_c[i]);
    }
    for (auto & sl : b->slots) {
        for (int i = 0; i < 2; ++i) { if (sl.d_rec[i]) (void) hipFree(sl.d_rec[i]); if (sl.d_ps[i]) (void) hipFree(sl.d_ps[i]); }
        if (sl.d_smask) (void) hipFree(sl.d_smask);
    }
    if (b->copy_stream) (void) hipStreamDestroy(b->copy_stream);
    if (b->bst) whisper_free_state(b->bst);
    if (b->h_rowp) (void) hipHostFree(b->h_rowp);
    if (b->d_rowp) (void) hipFree(b->d_rowp);
    delete b;
}
void wa_batcher_destroy(wa_batcher * b) {           // back to its context's cache
    if (!b) return;
    std::lock_guard<std::mutex> lk(batcher_cache_mutex());
    b->ctx->batcher_cache.push_back(b);
}
void wa_batcher_free_all(whisper_context & ctx) {
    std::vector<void *> all;
    { std::lock_guard<std::mutex> lk(batcher_cache_mutex()); all.swap(ctx.batcher_cache); }
    for (void * b : all) wa_batcher_release((wa_batcher *) b);
}
void wa_batcher_stats(const wa_batcher * b, long * steps, long * rows, long * one_launch) { if (b) { *steps = b->n_steps; *rows = b->n_rows; if (one_launch) *one_launch = b->n_one_launch; } }

// the request of a member that the next pass serves: its oldest one that no pass serves yet (null: none)
static wa_breq * batcher_next(wa_bslot & sl) {
    for (auto & r : sl.q) if (r.pass == -1 && r.result == 0) return &r;
    return nullptr;
}
static wa_bslot * batcher_slot(wa_batcher & b, whisper_state & st, bool make) {
    for (auto & sl : b.slots) if (sl.st == &st) return &sl;
    if (make) for (auto & sl : b.slots) if (!sl.st) { sl.st = &st; sl.q.clear(); sl.ahead = false; return &sl; }
    return nullptr;
}
// buffers of the asynchronous passes, on first use
static bool batcher_async_prepare(wa_batcher & b) {
    if (b.d_out[0]) return true;
    if (!b.async_ok) return false;
    const size_t n_vocab = (size_t) b.ctx->model.hp.n_vocab;
    bool ok = WA_HIP_OK(hipStreamCreateWithFlags(&b.copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2 && ok; ++i) {
        ok = WA_HIP_OK(hipMalloc((void **) &b.d_out[i], WA_MAX_DECODERS * n_vocab * sizeof(float))) && WA_HIP_OK(hipHostMalloc((void **) &b.h_out[i], WA_MAX_DECODERS * n_vocab * sizeof(float))) &&
             WA_HIP_OK(hipMalloc((void **) &b.d_stat[i], 128)) && WA_HIP_OK(hipMemset(b.d_stat[i], 0, 128)) && WA_HIP_OK(hipHostMalloc((void **) &b.h_stat[i], 128)) &&
             WA_HIP_OK(hipEventCreateWithFlags(&b.ev_k[i], hipEventDisableTiming)) && WA_HIP_OK(hipEventCreateWithFlags(&b.ev_c[i], hipEventDisableTiming));
    }
    if (!ok) b.async_ok = false;
    return ok;
}
static void batcher_release_device(wa_batcher & b) {       // (b.m held) give the device's one-launch slot back once nothing of this group is in flight
    if (!b.slot_held) return;
    for (const auto & sl : b.slots) if (sl.st && (sl.ahead || !sl.q.empty())) return;
    b.slot_held = false;
    mega_slot(b.ctx->device).unlock();
}

// The requests of a group as ONE launch (wa_rows.hip), asynchronously: launch, copy the rows out behind it, record an event - no host wait here.
// (b.m held)  false: not launched (the caller falls back)
static bool batcher_launch_async(wa_batcher & b, wa_bslot ** run, int B, int T, uint32_t kv_size) {
    whisper_context & ctx = *b.ctx;
    whisper_state & bs = *b.bst;
    const auto & m = ctx.model; const auto & hp = m.hp;
    if (!bs.rows_enabled || bs.rows_pause > 0 || !batcher_async_prepare(b) || T < 1 || (T >> 5) > 47 || T > bs.cross_tpad) return false;
    const int n_wg = std::min(m.n_cu, 256), quant = m.wtype != 1 ? 1 : 0;
    wa_rows_args a;
    memset(&a, 0, sizeof(a)); a.force_inorder = rows_force_inorder();
    if (wa_rows_lds_bytes(hp.n_text_state, B, n_wg, quant, &a.slot_bytes) == 0 || !wa_rows_prepare(ctx, bs)) return false;
    for (int i = 0; i < B; ++i) { const wa_breq & r = *batcher_next(*run[i]); if (r.n_kv < 1 || r.n_kv > WA_ROWS_MAXKV || r.kv_head < 0 || r.kv_head >= r.n_kv) return false; }
    const int par = (int) (b.n_launched & 1);
    a.layers = (const wa_mega_layer *) m.d_mega_layers;
    a.n_layer = hp.n_text_layer; a.d = hp.n_text_state; a.n_head = hp.n_text_head; a.n_vocab = hp.n_vocab; a.eps = hp.eps; a.rn_d = 1.0 / (double) hp.n_text_state;
    a.te = m.d_te; a.pe = m.d_pe; a.lnf_w = m.d_ln.w; a.lnf_b = m.d_ln.b; a.gelu = m.d_gelu; a.quant = quant;
    if (quant) { a.te = (const wa_f16 *) m.te_q.qs; a.te_d = m.te_q.qd; }
    a.kv_layer_stride = (unsigned long long) kv_size * hp.n_text_state;
    a.cross_layer_stride = (unsigned long long) hp.n_text_head * bs.cross_tpad * 64; a.cross_tpad = bs.cross_tpad; a.T = T;
    a.granules = bs.d_rows_gr; a.row_gr = 2 * hp.n_text_state; a.cross_gr = bs.d_rows_cgr;
    // the logits rows go straight into the pinned host rows (whole 256-byte stores, wa_rows.hip: mb_logits_out): kernel + 15 us instead of kernel - 30 us + a
    // 63 us copy per pass (WHISPER_AMD_ROWS_HOST_OUT=0: through device memory and a copy)
    static const bool host_out_env = getenv("WHISPER_AMD_ROWS_HOST_OUT") == nullptr || atoi(getenv("WHISPER_AMD_ROWS_HOST_OUT")) != 0;
    float * h_dev = nullptr;
    const bool host_out = host_out_env && hipHostGetDevicePointer((void **) &h_dev, b.h_out[par], 0) == hipSuccess && h_dev;
    a.logits = host_out ? h_dev : b.d_out[par]; a.status = b.d_stat[par]; a.tok_out = (int *) (b.d_stat[par] + 4); a.row_status = b.d_stat[par] + 12;
    a.kq_scale = pow(float(64), -0.25); a.B = B; a.n_out = B;
    a.token_beg = ctx.vocab.token_beg; a.token_eot = ctx.vocab.token_eot;
    for (int i = 0; i < B; ++i) {
        wa_bslot & sl = *run[i];
        wa_breq & r = *batcher_next(sl);
        whisper_state & ms = *sl.st;
        wa_rows_row & rr = a.rows[i];
        rr.kv_k = ms.kv_self.k; rr.kv_v = ms.kv_self.v; rr.cross_k = ms.d_cross_k; rr.cross_v = ms.d_cross_v; rr.mask = nullptr;
        rr.n_kv = r.n_kv; rr.kv_head = r.kv_head; rr.token = r.token < 0 ? 0 : r.token; rr.pos = r.pos;
        a.out_row[i] = i;
        if (r.ahead) {
            b.n_ahead += 1; if (r.token < 0) b.n_picks += 1;
            const int p = r.k & 1;
            rr.spec = r.token < 0 ? 1 : 0;
            rr.rec_in = sl.d_rec[p ^ 1]; rr.rec_out = sl.d_rec[p]; rr.ps_in = sl.d_ps[p ^ 1]; rr.ps_out = sl.d_ps[p]; rr.smask = sl.d_smask;
            rr.s_last = r.after.last; rr.s_penult = r.after.penult; rr.s_seek_delta = r.after.seek_delta; rr.s_has_ts = r.after.has_ts;
        }
    }
    bs.mega_seq += 1; if (bs.mega_seq == 0) bs.mega_seq = 1;
    a.seq = bs.mega_seq;
    if (!b.slot_held) { mega_slot(ctx.device).lock(); b.slot_held = true; }
    hipStream_t s = bs.stream;
    (void) hipMemsetAsync(b.d_stat[par] + 12, 0, WA_ROWS_MAX * sizeof(unsigned), s);
    if (!wa_launch_decode_rows(s, a, n_wg)) { bs.rows_enabled = false; batcher_release_device(b); return false; }
    // The rows go out on the SAME stream, in front of the next pass.  (On a stream of their own they were a copy kernel running beside the next
    // pass's persistent workgroups, which starved it: 1.66 MB took 0.95 ms - the members got pass k's logits when pass k + 1 ended, and every
    // second pass started 0.4 ms late; passes alternating between two streams, so that a copy could overlap the next kernel, gave the same starved copy
    // kernel.  On the pass's stream the copy is a DMA of 63 us; by default there is none: host_out above.)
    if (!host_out) (void) hipMemcpyAsync(b.h_out[par], b.d_out[par], (size_t) B * hp.n_vocab * sizeof(float), hipMemcpyDeviceToHost, s);
    (void) hipMemcpyAsync(b.h_stat[par], b.d_stat[par], 128, hipMemcpyDeviceToHost, s);
    (void) hipEventRecord(b.ev_c[par], s);
    wa_bpass & ps = b.passes[b.n_launched & 3];
    ps = wa_bpass(); ps.seq = a.seq; ps.B = B; ps.parity = par; ps.launched = true;
    for (int i = 0; i < B; ++i) { wa_breq & r = *batcher_next(*run[i]); r.pass = b.n_launched; r.row = i; r.result = 1; }
    b.n_launched += 1; b.n_steps += 1; b.n_rows += B; b.n_one_launch += 1;
    return true;
}

// The same requests through the launch sequence, synchronously (no one-launch form: odd shapes, a paused form, WHISPER_AMD_NO_MEGA).  Run-ahead
// requests cannot be served this way: their members are told to go on alone.  (b.m held)
static void batcher_run_sync(wa_batcher & b, wa_bslot ** run, int B, int T, uint32_t kv_size) {
    whisper_context & ctx = *b.ctx;
    whisper_state & bs = *b.bst;
    const int n_vocab = ctx.model.hp.n_vocab;
    const int64_t t_begin = wa_time_us();
    if (b.t_last_end) b.t_gap_us += t_begin - b.t_last_end;
    bool ok = ctx.model.wtype == 1;     // (the launch sequence of a quantised model has no per-row K / V form: the members decode alone)
    for (int i = 0; i < B; ++i) if (batcher_next(*run[i])->ahead) ok = false;
    if (ok) {
        hipStream_t s = bs.stream;
        int32_t * h_tok = bs.h_stage_i32, * h_pos = h_tok + bs.dec_mpad, * h_rows = h_pos + bs.dec_mpad;
        for (int i = 0; i < B; ++i) {
            whisper_state & ms = *run[i]->st; const wa_breq & r = *batcher_next(*run[i]);
            h_tok[i] = r.token; h_pos[i] = r.pos; h_rows[i] = i;
            b.h_rowp[i] = { ms.kv_self.k, ms.kv_self.v, ms.d_cross_k, ms.d_cross_v, r.n_kv, r.kv_head };
        }
        bs.enc_n_ctx = T;
        (void) hipMemcpyAsync(bs.d_tok, h_tok, B * sizeof(int32_t), hipMemcpyHostToDevice, s);
        (void) hipMemcpyAsync(bs.d_pos, h_pos, B * sizeof(int32_t), hipMemcpyHostToDevice, s);
        (void) hipMemcpyAsync(bs.d_rows, h_rows, B * sizeof(int32_t), hipMemcpyHostToDevice, s);
        (void) hipMemcpyAsync(b.d_rowp, b.h_rowp, B * sizeof(wa_rowptr), hipMemcpyHostToDevice, s);
        // (every row's own n_kv comes from its wa_rowptr; the launch-uniform n_kv only picks the kernel variant: <= 512 cells, one block per pair)
        static const bool no_graph = getenv("WHISPER_AMD_NO_GRAPH") != nullptr;
        if (b.graphs_ok && !no_graph && (!b.graph[B] || b.graph_T[B] != T || b.graph_kv[B] != kv_size)) {
            if (b.graph[B]) { (void) hipGraphExecDestroy(b.graph[B]); b.graph[B] = nullptr; }
            hipGraph_t g = nullptr;
            if (WA_HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) {
                decode_launch(ctx, bs, B, 512, 0, nullptr, B, false, nullptr, b.d_rowp, kv_size);
                if (WA_HIP_OK(hipStreamEndCapture(s, &g)) && g) {
                    if (!WA_HIP_OK(hipGraphInstantiate(&b.graph[B], g, nullptr, nullptr, 0))) b.graph[B] = nullptr;
                    (void) hipGraphDestroy(g);
                }
            }
            if (b.graph[B]) { b.graph_T[B] = T; b.graph_kv[B] = kv_size; } else b.graphs_ok = false;
        }
        if (b.graph[B] && !no_graph) ok = WA_HIP_OK(hipGraphLaunch(b.graph[B], s)) && ok;
        else decode_launch(ctx, bs, B, 512, 0, nullptr, B, false, nullptr, b.d_rowp, kv_size);
        ok = hipGetLastError() == hipSuccess && ok;
        (void) hipMemcpyAsync(bs.h_logits_pinned, bs.d_logits, (size_t) B * n_vocab * sizeof(float), hipMemcpyDeviceToHost, s);
        ok = WA_HIP_OK(hipStreamSynchronize(s)) && ok;
        if (!ok) b.graphs_ok = false;
    }
    // every member copies ITS row out on its own thread (the staging buffer is not written again before all of them are back with their next requests)
    for (int i = 0; i < B; ++i) { wa_breq & r = *batcher_next(*run[i]); r.pass = -2; r.row = i; r.result = ok ? 1 : -1; }
    b.n_steps += 1; b.n_rows += B;
    b.t_last_end = wa_time_us();
    b.t_pass_us += b.t_last_end - t_begin;
}

// (b.m held) launch a pass once every member still decoding has a request in that no pass serves yet
static void batcher_try_launch(wa_batcher & b) {
    wa_bslot * run[WA_MAX_DECODERS];
    int n = 0;
    for (auto & sl : b.slots) if (sl.st && batcher_next(sl)) run[n++] = &sl;
    if (n == 0 || n < b.n_members) return;
    whisper_context & ctx = *b.ctx;
    const auto & hp = ctx.model.hp;
    // members must agree on what the kernels take as launch-uniform: cells per layer and the audio context
    const whisper_state & s0 = *run[0]->st;
    const int T = s0.enc_n_ctx > 0 ? s0.enc_n_ctx : hp.n_audio_ctx;
    wa_bslot * ok_run[WA_MAX_DECODERS]; int B = 0;
    for (int i = 0; i < n; ++i) {
        const whisper_state & s = *run[i]->st;
        const int Ts = s.enc_n_ctx > 0 ? s.enc_n_ctx : hp.n_audio_ctx;
        if (s.kv_self.size == s0.kv_self.size && Ts == T && s.cross_tpad == b.bst->cross_tpad) ok_run[B++] = run[i];
        else batcher_next(*run[i])->result = -1;           // decoded by its own thread the ordinary way
    }
    // (tests: a pass whose launch failed must not hand out the stale contents of the staging buffer - every member then decodes alone)
    static const bool test_fail = getenv("WHISPER_AMD_TEST_FAIL_BATCH_LAUNCH") != nullptr;
    if (test_fail || !WA_HIP_OK(hipSetDevice(ctx.device))) { for (int i = 0; i < B; ++i) batcher_next(*ok_run[i])->result = -1; b.cv.notify_all(); return; }
    if (B > 0 && !batcher_launch_async(b, ok_run, B, T, s0.kv_self.size)) {
        bool any_ahead = false;
        for (int i = 0; i < B; ++i) any_ahead = any_ahead || batcher_next(*ok_run[i])->ahead;
        if (any_ahead) { for (int i = 0; i < B; ++i) for (auto & r : ok_run[i]->q) if (r.pass == -1 && r.result == 0) r.result = -1; }      // (no one-launch form: every member on its own from here)
        else batcher_run_sync(b, ok_run, B, T, s0.kv_self.size);
    }
    static const bool trace = getenv("WHISPER_AMD_BATCH_TRACE") != nullptr;
    if (trace && b.n_steps == 1) b.t_first = wa_time_us();
    if (trace && (b.n_steps % 100) == 0) fprintf(stderr, "[batcher] %ld passes (%ld as one launch; %ld run-ahead rows, %ld picked by the device): %.3f ms in a synchronous pass, %.3f ms between them (mean); first pass %.1f ms after the group formed, now %.1f ms\n", b.n_steps, b.n_one_launch, b.n_ahead, b.n_picks,
                                                 1e-3 * b.t_pass_us / std::max(1L, b.n_steps), 1e-3 * b.t_gap_us / std::max(1L, b.n_steps), 1e-3 * (b.t_first - b.t_created), 1e-3 * (wa_time_us() - b.t_created));
    b.cv.notify_all();
}

// Collect the request `k` of a member: wait until a pass serves it and has finished, copy the member's logits row into st.logits.
// 0 = done, 1 = the pass wants to be redone by the launch sequence (an uncertifiable soft-max sum), -1 = not served / the form gave up.
static int batcher_collect(wa_batcher & b, wa_bslot & sl, int * token_used) {
    std::unique_lock<std::mutex> lk(b.m);
    if (sl.q.empty()) return -1;
    b.cv.wait(lk, [&] { return sl.q.front().result != 0; });
    wa_breq r = sl.q.front();
    const size_t n_vocab = (size_t) b.ctx->model.hp.n_vocab;
    if (r.result != 1) { sl.q.pop_front(); batcher_release_device(b); return -1; }
    whisper_state & st = *sl.st;
    if (r.pass == -2) {             // a synchronous pass: the row is in the private state's staging buffer
        const float * row = b.bst->h_logits_pinned + (size_t) r.row * n_vocab;
        sl.q.pop_front();
        lk.unlock();
        st.logits.resize(n_vocab);
        memcpy(st.logits.data(), row, n_vocab * sizeof(float));
        return 0;
    }
    wa_bpass & ps = b.passes[r.pass & 3];
    const int par = ps.parity;
    lk.unlock();
    const bool synced = WA_HIP_OK(hipEventSynchronize(b.ev_c[par]));
    lk.lock();
    if (!ps.done) {                 // the first member back reads the pass's status word and the echo of its sequence number
        ps.done = true;
        ps.status = synced ? b.h_stat[par][0] : 9998u;
        ps.ok = synced && ps.status == 0 && b.h_stat[par][1] == ps.seq;
        if (!ps.ok) {
            whisper_state & bs = *b.bst;
            bs.n_rows_fallback += 1;
            static const bool trace_f = getenv("WHISPER_AMD_BATCH_TRACE") != nullptr;
            if (trace_f) { const int * tk = (const int *) (b.h_stat[par] + 4); fprintf(stderr, "[batcher] pass %ld failed: status %u echo %u seq %u B %d tokens %d %d %d %d %d %d %d %d  logits[0][0..3] %g %g %g %g\n", r.pass, ps.status, b.h_stat[par][1], ps.seq, ps.B,
                                    tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], tk[7], b.h_out[par][0], b.h_out[par][1], b.h_out[par][2], b.h_out[par][3]); }
            (void) hipMemsetAsync(b.d_stat[par], 0, sizeof(unsigned), bs.stream);
            if (ps.status != WA_MEGA_REDO) one_launch_timeout(bs.rows_pause, bs.rows_timeouts, bs.rows_enabled, "several-rows one-launch step (lock-step group)", ps.status);
        } else b.bst->n_rows_steps += 1;
    }
    const bool ok = ps.ok && b.h_stat[par][12 + r.row] == 0;
    const unsigned status = ps.ok ? b.h_stat[par][12 + r.row] : ps.status;      // (WA_MEGA_REDO for THIS row: its member has the step redone, the others go on)
    const float * row = b.h_out[par] + (size_t) r.row * n_vocab;
    if (token_used) *token_used = ((const int *) (b.h_stat[par] + 4))[r.row];
    sl.q.pop_front();
    batcher_release_device(b);
    lk.unlock();
    if (!ok) return status == WA_MEGA_REDO ? 1 : -1;
    st.logits.resize(n_vocab);
    memcpy(st.logits.data(), row, n_vocab * sizeof(float));
    return 0;
}

// a member's plain single-token step: 1 = logits delivered into st.logits, 0 = not served (the caller decodes it itself)
static int batcher_step(wa_batcher & b, whisper_state & st, int token, int pos, int n_kv, int kv_head) {
    wa_bslot * sl;
    {
        std::unique_lock<std::mutex> lk(b.m);
        if (b.n_members < 2) return 0;                  // the last chunk still decoding: nothing to share a pass with
        sl = batcher_slot(b, st, true);
        if (!sl || !sl->q.empty()) return 0;
        wa_breq r; r.token = token; r.pos = pos; r.n_kv = n_kv; r.kv_head = kv_head;
        sl->q.push_back(r);
        batcher_try_launch(b);
    }
    return batcher_collect(b, *sl, nullptr) == 0 ? 1 : 0;
}
// a member is done (or failed): the others no longer wait for it
void wa_batcher_leave(wa_batcher * b) {
    if (!b) return;
    std::unique_lock<std::mutex> lk(b->m);
    b->n_members -= 1;
    if (b->n_members < 2) {         // nobody to share a pass with: whoever still waits decodes alone
        bool any = false;
        for (auto & sl : b->slots) for (auto & r : sl.q) if (r.pass == -1 && r.result == 0) { r.result = -1; any = true; }
        if (any) b->cv.notify_all();
    } else batcher_try_launch(*b);
}

// ---- run-ahead windows of lock-step members (the wa_spec_* protocol of wa_full.cpp, served by the group's passes) ----
static bool bspec_begin(wa_batcher & b, whisper_state & st, const std::vector<uint32_t> & bits) {
    whisper_context & ctx = *b.ctx;
    static const bool trace_b = getenv("WHISPER_AMD_BATCH_TRACE") != nullptr;
    if (trace_b) fprintf(stderr, "[batcher] run-ahead window asked for: members %d rows_enabled %d pause %d\n", b.n_members, (int) b.bst->rows_enabled, b.bst->rows_pause);
    const size_t n_mask = (size_t) ctx.model.hp.n_vocab / 32 + 2;
    std::unique_lock<std::mutex> lk(b.m);
    if (b.n_members < 2 || !b.bst->rows_enabled || b.bst->rows_pause > 0 || bits.size() > n_mask) return false;
    static const bool off = getenv("WHISPER_AMD_NO_RUN_AHEAD") != nullptr;
    if (off || !batcher_async_prepare(b)) return false;
    wa_bslot * sl = batcher_slot(b, st, true);
    if (!sl || !sl->q.empty()) return false;
    if (!sl->d_smask) {
        bool ok = WA_HIP_OK(hipMalloc((void **) &sl->d_smask, n_mask * 4));
        for (int i = 0; i < 2 && ok; ++i) ok = WA_HIP_OK(hipMalloc((void **) &sl->d_rec[i], 512 * 8 * 4)) && WA_HIP_OK(hipMemset(sl->d_rec[i], 0, 512 * 8 * 4)) &&
                                               WA_HIP_OK(hipMalloc((void **) &sl->d_ps[i], 32)) && WA_HIP_OK(hipMemset(sl->d_ps[i], 0, 32));
        if (!ok) return false;
    }
    std::vector<uint32_t> up(n_mask, 0u);
    std::copy(bits.begin(), bits.end(), up.begin());
    if (const char * e = getenv("WHISPER_AMD_OVERLAP_SABOTAGE")) if (e[0] == '1') for (size_t i = 0; i < up.size(); ++i) up[i] |= 0x49249249u << (i % 3);      // (tests: wrong predictions on purpose)
    if (!WA_HIP_OK(hipMemcpy(sl->d_smask, up.data(), n_mask * 4, hipMemcpyHostToDevice))) return false;
    sl->ahead = true;
    return true;
}
static bool bspec_launch(wa_batcher & b, whisper_state & st, int k, int pos, int token, const wa_spec_state & after) {
    std::unique_lock<std::mutex> lk(b.m);
    wa_bslot * sl = batcher_slot(b, st, false);
    if (!sl || !sl->ahead || sl->q.size() >= 2 || b.n_members < 2) return false;
    wa_breq r; r.k = k; r.token = token; r.pos = pos; r.n_kv = pos + 1; r.kv_head = pos; r.ahead = true; r.after = after;      // greedy steady state: cell == position
    sl->q.push_back(r);
    batcher_try_launch(b);
    return true;
}
static int bspec_wait(wa_batcher & b, whisper_state & st, int * token_used) {
    wa_bslot * sl;
    { std::unique_lock<std::mutex> lk(b.m); sl = batcher_slot(b, st, false); }
    return sl ? batcher_collect(b, *sl, token_used) : -1;
}
static void bspec_drain(wa_batcher & b, whisper_state & st) {          // everything of this member that is in flight: waited for and dropped
    wa_bslot * sl;
    { std::unique_lock<std::mutex> lk(b.m); sl = batcher_slot(b, st, false); if (!sl) return;
      for (auto & r : sl->q) if (r.pass == -1 && r.result == 0) r.result = -1; }
    for (;;) {
        { std::unique_lock<std::mutex> lk(b.m); if (sl->q.empty()) break; }
        (void) batcher_collect(b, *sl, nullptr);
    }
}
static void bspec_end(wa_batcher & b, whisper_state & st) {
    bspec_drain(b, st);
    std::unique_lock<std::mutex> lk(b.m);
    if (wa_bslot * sl = batcher_slot(b, st, false)) sl->ahead = false;
    batcher_release_device(b);
}

bool wa_decode(whisper_context & ctx, whisper_state & st, const wa_batch & batch, bool save_aheads, ggml_abort_callback abort_cb,
               void * abort_data) {
    const int64_t t0 = wa_time_us();
    const auto & m  = ctx.model;
    const auto & hp = m.hp;
    const int n_vocab = hp.n_vocab, n_tokens = batch.n_tokens;
    if (n_tokens <= 0 || n_tokens > st.dec_mpad) { WA_ERROR("%s: bad batch size %d\n", __func__, n_tokens); return false; }
    if (!WA_HIP_OK(hipSetDevice(ctx.device))) return false;

    // validate the batch BEFORE it takes KV cells: an error return must not leave cells allocated
    for (int i = 0; i < n_tokens; ++i)
        if (batch.token[i] < 0 || batch.token[i] >= n_vocab || batch.pos[i] < 0 || batch.pos[i] >= hp.n_text_ctx) {
            WA_ERROR("%s: token %d / position %d out of range\n", __func__, batch.token[i], batch.pos[i]);
            return false;
        }
    {
        int n_rows_req = 0;
        for (int i = 0; i < n_tokens; ++i) n_rows_req += batch.logits[i] ? 1 : 0;
        if (n_rows_req > WA_MAX_DECODERS) { WA_ERROR("%s: too many logits rows requested (%d)\n", __func__, n_rows_req); return false; }
    }
    auto & kv = st.kv_self;
    if (!wa_kv_find_slot(kv, batch)) return false;
    kv.n = std::min(kv.size, (uint32_t) std::max(1, wa_kv_cell_max(kv)));     // padding = 1 (whisper.cpp:2892-2893)
    const int n_kv = kv.n, kv_head = kv.head;

    st.logits.resize((size_t) n_tokens * n_vocab);
    if (m.n_loaded == 0) {          // header-only test model
        std::fill(st.logits.begin(), st.logits.end(), 0.0f);
        return !(abort_cb && abort_cb(abort_data));
    }

    hipStream_t s = st.stream;

    // ---- inputs: tokens, positions, rows that need logits, KQ mask (whisper.cpp:2912-2956) ----
    int32_t * h_tok = st.h_stage_i32, * h_pos = h_tok + st.dec_mpad, * h_rows = h_pos + st.dec_mpad;
    int n_rows = 0;
    for (int i = 0; i < n_tokens; ++i) {
        h_tok[i] = batch.token[i];
        h_pos[i] = batch.pos[i];
        if (batch.logits[i]) h_rows[n_rows++] = i;
    }
    if ((size_t) n_tokens * n_kv > st.h_mask_cap) { WA_ERROR("%s: mask overflow\n", __func__); return false; }
    int8_t * h_mask = st.h_stage_mask;
    for (int j = 0; j < n_tokens; ++j) {
        const int32_t pos = batch.pos[j], seq = batch.seq_id[j];
        for (int i = 0; i < n_kv; ++i) h_mask[(size_t) j * n_kv + i] = (!kv.cells[i].has(seq) || kv.cells[i].pos > pos) ? 1 : 0;
    }
    // single token, every cell below n_kv visible (the greedy steady state): one persistent launch (wa_mega.hip), or - when that
    // is not available - a replay of the captured hipGraph of the launch sequence
    bool need_mask = false;
    for (size_t i = 0; i < (size_t) n_tokens * n_kv && !need_mask; ++i) need_mask = h_mask[i] != 0;
    const bool steady = n_tokens == 1 && n_rows == 1 && !need_mask && !save_aheads;
    bool done = false, from_batcher = false;
    const bool solo = st.solo_step; st.solo_step = false;
    const bool device_free = wa_encoders_in_flight(ctx.device).load(std::memory_order_relaxed) == 0;      // (wa_internal.h: no one-launch step beside encoder passes)
    if (steady && st.batcher && !solo) done = from_batcher = batcher_step(*st.batcher, st, h_tok[0], h_pos[0], n_kv, kv_head) == 1;
    if (!done && steady && st.single_via_rows && device_free) {        // a wide quantised model: the several-rows kernel with ONE row (wa_internal.h: single_via_rows)
        const wa_rows_row r1 = { kv.k, kv.v, st.d_cross_k, st.d_cross_v, nullptr, n_kv, kv_head, h_tok[0], h_pos[0] };
        const int T1 = st.enc_n_ctx > 0 ? st.enc_n_ctx : (st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx);
        done = rows_step(ctx, st, 1, &r1, 1, nullptr, T1, st.cross_tpad, kv.size, st.batcher == nullptr) == 1;
    }
    if (done) { }
    else if (!done && steady && st.mega_enabled && st.mega_pause > 0) st.mega_pause -= 1;      // (paused after a time-out: this pass takes the launch sequence)
    else if (!done && steady && st.mega_enabled && device_free) done = mega_step(ctx, st, h_tok[0], h_pos[0], n_kv, kv_head) == 1;
    if (!done && !steady && n_tokens <= WA_ROWS_MAX && n_rows >= 1 && !save_aheads && st.rows_enabled && n_kv <= WA_ROWS_MAXKV && device_free) {
        // one token per live decoder (beam search, best_of, the bench's small batches): all rows in ONE launch (wa_rows.hip)
        if (need_mask) (void) hipMemcpyAsync(st.d_mask, h_mask, (size_t) n_tokens * n_kv, hipMemcpyHostToDevice, s);
        wa_rows_row rr[WA_MAX_DECODERS];
        for (int i = 0; i < n_tokens; ++i)
            rr[i] = { kv.k, kv.v, st.d_cross_k, st.d_cross_v, need_mask ? st.d_mask + (size_t) i * n_kv : nullptr, n_kv, kv_head + i, h_tok[i], h_pos[i] };
        const int T = st.enc_n_ctx > 0 ? st.enc_n_ctx : (st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx);
        static const bool wait_slot = getenv("WHISPER_AMD_ROWS_WAIT") != nullptr;
        done = rows_step(ctx, st, n_tokens, rr, n_rows, h_rows, T, st.cross_tpad, kv.size, wait_slot) == 1;
    }
    if (!done) {
    (void) hipMemcpyAsync(st.d_tok,  h_tok,  n_tokens * sizeof(int32_t), hipMemcpyHostToDevice, s);
    (void) hipMemcpyAsync(st.d_pos,  h_pos,  n_tokens * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (n_rows) (void) hipMemcpyAsync(st.d_rows, h_rows, n_rows * sizeof(int32_t), hipMemcpyHostToDevice, s);
    (void) hipMemcpyAsync(st.d_mask, h_mask, (size_t) n_tokens * n_kv, hipMemcpyHostToDevice, s);

    const int32_t row0 = 0;
    if (st.graphs_enabled && steady) {
        const int T = st.enc_n_ctx > 0 ? st.enc_n_ctx : hp.n_audio_ctx;
        if (st.dec_graph && (st.dec_graph_T != T || st.dec_graph_kv_size != kv.size || st.dec_graph_kv_k != kv.k)) {
            (void) hipGraphExecDestroy(st.dec_graph); st.dec_graph = nullptr;
        }
        int32_t * h_dyn = st.h_stage_i32 + 3 * st.dec_mpad;
        h_dyn[0] = n_kv; h_dyn[1] = kv_head;
        (void) hipMemcpyAsync(st.d_dyn, h_dyn, 2 * sizeof(int32_t), hipMemcpyHostToDevice, s);
        if (!st.dec_graph) {
            hipGraph_t g = nullptr;
            if (WA_HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) {
                if (m.wtype != 1) decode_launch_quant(ctx, st, 1, n_kv, kv_head, nullptr, 1, &row0, false, st.d_dyn);
                else              decode_launch(ctx, st, 1, n_kv, kv_head, nullptr, 1, false, st.d_dyn);
                if (WA_HIP_OK(hipStreamEndCapture(s, &g)) && g) {
                    if (!WA_HIP_OK(hipGraphInstantiate(&st.dec_graph, g, nullptr, nullptr, 0))) st.dec_graph = nullptr;
                    (void) hipGraphDestroy(g);
                }
            }
            if (st.dec_graph) { st.dec_graph_T = T; st.dec_graph_kv_size = kv.size; st.dec_graph_kv_k = kv.k; }
            else { st.graphs_enabled = false; WA_WARN("%s: hipGraph capture failed, falling back to eager launches\n", __func__); }
        }
        if (st.dec_graph) { if (!WA_HIP_OK(hipGraphLaunch(st.dec_graph, s))) return false; }
        else if (m.wtype != 1) decode_launch_quant(ctx, st, n_tokens, n_kv, kv_head, nullptr, n_rows, h_rows, save_aheads);
        else decode_launch(ctx, st, n_tokens, n_kv, kv_head, nullptr, n_rows, save_aheads);
    } else if (m.wtype != 1) {
        decode_launch_quant(ctx, st, n_tokens, n_kv, kv_head, need_mask ? st.d_mask : nullptr, n_rows, h_rows, save_aheads);
    } else {
        decode_launch(ctx, st, n_tokens, n_kv, kv_head, need_mask ? st.d_mask : nullptr, n_rows, save_aheads);
    }
    if (n_rows) (void) hipMemcpyAsync(st.h_logits_pinned, st.d_logits, (size_t) n_rows * n_vocab * sizeof(float), hipMemcpyDeviceToHost, s);
    if (!WA_HIP_OK(hipStreamSynchronize(s))) return false;
    }
    st.staged_n = 0;
    if (!from_batcher) {     // (the batcher delivered its row straight into st.logits)
        if (st.defer_rows && n_rows > 1) { st.staged_n = n_rows; st.staged_of.assign(h_rows, h_rows + n_rows); }      // the caller's per-decoder threads fetch them
        else for (int r = 0; r < n_rows; ++r)
            memcpy(st.logits.data() + (size_t) h_rows[r] * n_vocab, st.h_logits_pinned + (size_t) r * n_vocab, n_vocab * sizeof(float));
    }

    const int64_t dt = wa_time_us() - t0;
    if (n_tokens == 1)       { st.t_decode_us += dt; st.n_decode++; }
    else if (n_tokens < 16)  { st.t_batchd_us += dt; st.n_batchd += n_tokens; }
    else                     { st.t_prompt_us += dt; st.n_prompt += n_tokens; }
    return !(abort_cb && abort_cb(abort_data));
}

// -------------------------------------------------------------------------------------------------
// measurement helper: replay the single-token decoder pass `n_iters` times back to back on the state's
// stream between two HIP events (no host work in between) -> average device time of ONE decode step.
// Used by bench.py for the HBM roofline of the decode step.  Uses KV cells [0, n_past]
// of sequence 0 and leaves the cell metadata untouched.
// -------------------------------------------------------------------------------------------------
extern "C" int whisper_amd_decode_step_probe(struct whisper_context * ctx, struct whisper_state * st, int n_past, int n_iters, float * ms_per_step) {
    if (!ctx || !st || !ms_per_step || n_iters <= 0 || ctx->model.n_loaded == 0) return -1;
    if (n_past < 0 || n_past + 1 > (int) st->kv_self.size || n_past >= ctx->model.hp.n_text_ctx) return -1;
    if (!WA_HIP_OK(hipSetDevice(ctx->device))) return -1;
    hipStream_t s = st->stream;
    int32_t * h = st->h_stage_i32;
    h[0] = ctx->vocab.token_eot > 100 ? 100 : 0;   // any valid token
    h[1] = n_past;
    h[2] = 0;
    (void) hipMemcpyAsync(st->d_tok,  h,     sizeof(int32_t), hipMemcpyHostToDevice, s);
    (void) hipMemcpyAsync(st->d_pos,  h + 1, sizeof(int32_t), hipMemcpyHostToDevice, s);
    (void) hipMemcpyAsync(st->d_rows, h + 2, sizeof(int32_t), hipMemcpyHostToDevice, s);
    hipEvent_t e0, e1;
    if (!WA_HIP_OK(hipEventCreate(&e0)) || !WA_HIP_OK(hipEventCreate(&e1))) return -1;
    if (st->mega_enabled) {     // the one-launch step: n_iters launches back to back, each with its own sequence number
        wa_mega_args a;
        if (mega_args(*ctx, *st, a, h[0], n_past, n_past + 1, n_past)) {
            std::lock_guard<std::mutex> lk(mega_slot(ctx->device));
            const int n_wg = std::min(ctx->model.n_cu, 256);
            (void) wa_launch_decode_mega(s, a, n_wg);      // warm-up
            (void) hipEventRecord(e0, s);
            for (int i = 0; i < n_iters; ++i) { mega_args(*ctx, *st, a, h[0], n_past, n_past + 1, n_past); (void) wa_launch_decode_mega(s, a, n_wg); }
            (void) hipEventRecord(e1, s);
            if (!WA_HIP_OK(hipEventSynchronize(e1))) return -1;
            float ms = 0.f;
            (void) hipEventElapsedTime(&ms, e0, e1);
            (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
            unsigned status = 0;
            (void) hipMemcpy(&status, st->d_mega_status, sizeof(status), hipMemcpyDeviceToHost);
            if (status != 0) { (void) hipMemset(st->d_mega_status, 0, sizeof(unsigned)); st->mega_enabled = false; return -2; }
            *ms_per_step = ms / n_iters;
            return 0;
        }
    }
    h[3] = n_past + 1; h[4] = n_past;
    (void) hipMemcpyAsync(st->d_dyn, h + 3, 2 * sizeof(int32_t), hipMemcpyHostToDevice, s);
    hipGraphExec_t ge = nullptr;
    const int32_t row0 = 0;
    if (st->graphs_enabled) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            if (ctx->model.wtype != 1) decode_launch_quant(*ctx, *st, 1, n_past + 1, n_past, nullptr, 1, &row0, false, st->d_dyn);
            else                       decode_launch(*ctx, *st, 1, n_past + 1, n_past, nullptr, 1, false, st->d_dyn);
            if (hipStreamEndCapture(s, &g) == hipSuccess && g) { if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) ge = nullptr; (void) hipGraphDestroy(g); }
        }
    }
    auto one = [&]() {
        if (ge) (void) hipGraphLaunch(ge, s);
        else if (ctx->model.wtype != 1) decode_launch_quant(*ctx, *st, 1, n_past + 1, n_past, nullptr, 1, &row0, false);
        else decode_launch(*ctx, *st, 1, n_past + 1, n_past, nullptr, 1, false);
    };
    one();      // warm-up
    (void) hipEventRecord(e0, s);
    for (int i = 0; i < n_iters; ++i) one();
    (void) hipEventRecord(e1, s);
    if (!WA_HIP_OK(hipEventSynchronize(e1))) return -1;
    float ms = 0.f;
    (void) hipEventElapsedTime(&ms, e0, e1);
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    if (ge) (void) hipGraphExecDestroy(ge);
    *ms_per_step = ms / n_iters;
    return 0;
}

// the workgroup -> role map of the one-launch step (wa_mega.h), for the CPU tests
extern "C" void whisper_amd_mega_role_of(int n_wg, int n_head, int wg, int * role, int * index) { mg_role_of(n_wg, n_head, wg, *role, *index); }

// -------------------------------------------------------------------------------------------------
// debugging aid (tools/mega_check.py): run the one-launch step for (token, pos) on KV cell `n_past` and copy out the
// hand-off granules [layer][8][2d] (tag << 32 | value) and the logits.  KV metadata is not touched.
// -------------------------------------------------------------------------------------------------
extern "C" int whisper_amd_mega_debug(struct whisper_context * ctx, struct whisper_state * st, int token, int n_past, unsigned long long * granules_out,
                                      float * logits_out) {
    static float * d_dbg = nullptr;
    if (!ctx || !st || !st->mega_enabled) return -1;
    if (!WA_HIP_OK(hipSetDevice(ctx->device))) return -1;
    wa_mega_args a;
    if (!mega_args(*ctx, *st, a, token, n_past, n_past + 1, n_past)) return -2;
    const auto & hp = ctx->model.hp;
    const size_t n_dbg = (size_t) hp.n_text_layer * hp.n_text_head * 5120 + 8192;
    if (getenv("WHISPER_AMD_MEGA_DBG")) { if (!d_dbg) (void) hipMalloc((void **) &d_dbg, n_dbg * 4); a.dbg = d_dbg; }
    std::lock_guard<std::mutex> lk(mega_slot(ctx->device));
    int n_wg = std::min(ctx->model.n_cu, 256);
    if (const char * e = getenv("WHISPER_AMD_MEGA_WG")) n_wg = std::max(2 * hp.n_text_head + 1, std::min(n_wg, atoi(e)));
    (void) wa_launch_decode_mega(st->stream, a, n_wg);
    if (!WA_HIP_OK(hipStreamSynchronize(st->stream))) return -3;
    if (a.dbg) { std::vector<float> h(n_dbg); (void) hipMemcpy(h.data(), d_dbg, n_dbg * 4, hipMemcpyDeviceToHost); FILE * f = fopen("gpurun_out/mega_dbg.bin", "wb"); if (f) { fwrite(h.data(), 4, n_dbg, f); fclose(f); } }
    if (granules_out)      // [layer][8][2d]: the first 2d granules of every edge (the device keeps 4d per edge)
        (void) hipMemcpy2D(granules_out, (size_t) 2 * hp.n_text_state * 8, st->d_mega_gr, (size_t) 4 * hp.n_text_state * 8, (size_t) 2 * hp.n_text_state * 8,
                           (size_t) hp.n_text_layer * WA_MEGA_EDGES, hipMemcpyDeviceToHost);
    if (logits_out) (void) hipMemcpy(logits_out, st->d_mega_out, (size_t) hp.n_vocab * 4, hipMemcpyDeviceToHost);
    unsigned status = 0;
    (void) hipMemcpy(&status, st->d_mega_status, 4, hipMemcpyDeviceToHost);
    return (int) status;
}

// -------------------------------------------------------------------------------------------------
// the several-rows one-launch step (wa_rows.hip): statistics, a debugging entry and a timing probe
// -------------------------------------------------------------------------------------------------
extern "C" void whisper_amd_rows_stats(struct whisper_state * st, long out[2]) {
    out[0] = st ? st->n_rows_steps : 0; out[1] = st ? st->n_rows_fallback : 0;
}
extern "C" int whisper_amd_rows_enabled(struct whisper_state * st) { return st && st->rows_enabled ? 1 : 0; }

static bool rows_probe_args(whisper_context & ctx, whisper_state & own, whisper_state ** sts, int B, const int * tokens, int n_past, wa_rows_args & a) {
    const auto & m = ctx.model; const auto & hp = m.hp;
    const int T = own.enc_n_ctx > 0 ? own.enc_n_ctx : hp.n_audio_ctx;
    memset(&a, 0, sizeof(a)); a.force_inorder = rows_force_inorder();
    if (B < 1 || B > WA_ROWS_MAX || !own.rows_enabled || n_past < 0 || n_past + 1 > (int) own.kv_self.size || n_past + 1 > WA_ROWS_MAXKV || (T >> 5) > 47) return false;
    const int quant = m.wtype != 1 ? 1 : 0;
    if (wa_rows_lds_bytes(hp.n_text_state, B, std::min(m.n_cu, 256), quant, &a.slot_bytes) == 0 || !wa_rows_prepare(ctx, own)) return false;
    a.layers = (const wa_mega_layer *) m.d_mega_layers;
    a.n_layer = hp.n_text_layer; a.d = hp.n_text_state; a.n_head = hp.n_text_head; a.n_vocab = hp.n_vocab; a.eps = hp.eps; a.rn_d = 1.0 / (double) hp.n_text_state;
    a.te = m.d_te; a.pe = m.d_pe; a.lnf_w = m.d_ln.w; a.lnf_b = m.d_ln.b; a.gelu = m.d_gelu; a.quant = quant;
    if (quant) { a.te = (const wa_f16 *) m.te_q.qs; a.te_d = m.te_q.qd; }
    a.kv_layer_stride = (unsigned long long) own.kv_self.size * hp.n_text_state;
    a.cross_layer_stride = (unsigned long long) hp.n_text_head * own.cross_tpad * 64; a.cross_tpad = own.cross_tpad; a.T = T;
    a.granules = own.d_rows_gr; a.row_gr = 2 * hp.n_text_state; a.cross_gr = own.d_rows_cgr;
    a.logits = own.d_logits; a.status = own.d_rows_status; a.row_status = own.d_rows_status + 4; a.kq_scale = pow(float(64), -0.25); a.B = B;
    a.n_out = B; for (int i = 0; i < B; ++i) a.out_row[i] = i;
    for (int i = 0; i < B; ++i) {
        whisper_state & ms = sts && sts[i] ? *sts[i] : own;
        if (ms.kv_self.size != own.kv_self.size || ms.cross_tpad != own.cross_tpad) return false;
        a.rows[i] = { ms.kv_self.k, ms.kv_self.v, ms.d_cross_k, ms.d_cross_v, nullptr, n_past + 1, n_past, tokens ? tokens[i] : 100, n_past };
    }
    return true;
}
static void rows_next_seq(whisper_state & st, wa_rows_args & a) { st.mega_seq += 1; if (st.mega_seq == 0) st.mega_seq = 1; a.seq = st.mega_seq; }

// B rows = the SAME (token, position n_past) of this state (cell n_past; identical rows write it identically): every row's granules must then
// equal the one-row step's (whisper_amd_mega_debug) and every logits row the launch sequence's.  granules_out [layer][8][B][2 d], logits_out [B][n_vocab].
extern "C" int whisper_amd_rows_debug(struct whisper_context * ctx, struct whisper_state * st, int B, int token, int n_past,
                                      unsigned long long * granules_out, float * logits_out) {
    if (!ctx || !st) return -1;
    if (!WA_HIP_OK(hipSetDevice(ctx->device))) return -1;
    int toks[WA_ROWS_MAX]; for (int i = 0; i < WA_ROWS_MAX; ++i) toks[i] = token;
    wa_rows_args a;
    if (!rows_probe_args(*ctx, *st, nullptr, B, toks, n_past, a)) return -2;
    rows_next_seq(*st, a);
    static float * d_dbg = nullptr;
    if (const char * e = getenv("WHISPER_AMD_ROWS_TRACE")) {       // stamps of workgroup `e` (tools/rows_trace.py)
        if (!d_dbg) (void) hipMalloc((void **) &d_dbg, 16384 * 4);
        (void) hipMemset(d_dbg, 0, 16384 * 4);
        const int twg = atoi(e);
        (void) hipMemcpy((int *) d_dbg + 4095, &twg, 4, hipMemcpyHostToDevice);
        a.dbg = d_dbg;
    }
    std::lock_guard<std::mutex> lk(mega_slot(ctx->device));
    if (!wa_launch_decode_rows(st->stream, a, std::min(ctx->model.n_cu, 256))) return -4;
    if (a.dbg) { rows_next_seq(*st, a); (void) wa_launch_decode_rows(st->stream, a, std::min(ctx->model.n_cu, 256)); }      // (the stamps of a warm run)
    if (!WA_HIP_OK(hipStreamSynchronize(st->stream))) return -3;
    if (a.dbg) { std::vector<unsigned> h(16384); (void) hipMemcpy(h.data(), d_dbg, 16384 * 4, hipMemcpyDeviceToHost); FILE * f = fopen("gpurun_out/rows_trace.bin", "wb"); if (f) { fwrite(h.data(), 4, 16384, f); fclose(f); } else WA_WARN("%s: cannot write gpurun_out/rows_trace.bin\n", __func__); }
    const auto & hp = ctx->model.hp;
    if (granules_out) (void) hipMemcpy(granules_out, st->d_rows_gr, (size_t) hp.n_text_layer * WA_MEGA_EDGES * B * a.row_gr * 8, hipMemcpyDeviceToHost);
    if (logits_out) (void) hipMemcpy(logits_out, st->d_logits, (size_t) B * hp.n_vocab * 4, hipMemcpyDeviceToHost);
    unsigned status[2] = { 0, 0 };
    (void) hipMemcpy(status, st->d_rows_status, 8, hipMemcpyDeviceToHost);
    if (status[0] != 0) (void) hipMemset(st->d_rows_status, 0, 4);
    if (status[0] == 0 && status[1] != a.seq) return -5;
    return (int) status[0];
}

// Measurement helper (bench.py): the B-row step `n_iters` times back to back between two HIP events; row i attends over states[i]'s cells
// [0, n_past] and encoder K / V (states[i] null = the first state: rows of one chunk, as a beam step).  Cell n_past of every state is overwritten.
extern "C" int whisper_amd_rows_step_probe(struct whisper_context * ctx, struct whisper_state ** states, int B, int n_past, int n_iters, float * ms_per_step) {
    if (!ctx || !states || !states[0] || !ms_per_step || n_iters <= 0) return -1;
    if (!WA_HIP_OK(hipSetDevice(ctx->device))) return -1;
    whisper_state & own = *states[0];
    wa_rows_args a;
    if (!rows_probe_args(*ctx, own, states, B, nullptr, n_past, a)) return -2;
    const int n_wg = std::min(ctx->model.n_cu, 256);
    hipStream_t s = own.stream;
    hipEvent_t e0, e1;
    if (!WA_HIP_OK(hipEventCreate(&e0)) || !WA_HIP_OK(hipEventCreate(&e1))) return -1;
    std::lock_guard<std::mutex> lk(mega_slot(ctx->device));
    rows_next_seq(own, a);
    if (!wa_launch_decode_rows(s, a, n_wg)) return -4;       // warm-up
    (void) hipEventRecord(e0, s);
    for (int i = 0; i < n_iters; ++i) { rows_next_seq(own, a); (void) wa_launch_decode_rows(s, a, n_wg); }
    (void) hipEventRecord(e1, s);
    if (!WA_HIP_OK(hipEventSynchronize(e1))) return -3;
    float ms = 0.f;
    (void) hipEventElapsedTime(&ms, e0, e1);
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    unsigned status[2] = { 0, 0 };
    (void) hipMemcpy(status, own.d_rows_status, 8, hipMemcpyDeviceToHost);
    if (status[0] != 0) { (void) hipMemset(own.d_rows_status, 0, 4); return (int) status[0]; }
    if (status[1] != a.seq) return -5;
    *ms_per_step = ms / n_iters;
    return 0;
}

extern "C" int whisper_amd_mega_enabled(struct whisper_state * st) { return st && st->mega_enabled ? 1 : 0; }

// The same step through the launch sequence, stage by stage, with every stage's output copied out in the granule
// layout of whisper_amd_mega_debug (values only: F32 bits, or two F16 per word) - to localise a difference.
extern "C" int whisper_amd_seq_debug(struct whisper_context * ctxp, struct whisper_state * stp, int token, int n_past, unsigned * values_out,
                                     float * logits_out) {
    if (!ctxp || !stp || ctxp->model.wtype != 1) return -1;
    whisper_context & ctx = *ctxp; whisper_state & st = *stp;
    if (!WA_HIP_OK(hipSetDevice(ctx.device))) return -1;
    const auto & m = ctx.model; const auto & hp = m.hp;
    const int d = hp.n_text_state, H = hp.n_text_head, n_kv = n_past + 1, kv_head = n_past;
    const int T = st.enc_n_ctx > 0 ? st.enc_n_ctx : hp.n_audio_ctx;
    hipStream_t s = st.stream;
    auto & kv = st.kv_self;
    const int32_t h3[3] = { token, n_past, 0 };
    (void) hipMemcpy(st.d_tok, &h3[0], 4, hipMemcpyHostToDevice);
    (void) hipMemcpy(st.d_pos, &h3[1], 4, hipMemcpyHostToDevice);
    (void) hipMemcpy(st.d_rows, &h3[2], 4, hipMemcpyHostToDevice);
    const size_t es = (size_t) 2 * d;
    auto out = [&](int l, int e) { return values_out + ((size_t) l * 8 + e) * es; };
    float * d_qk = nullptr; std::vector<float> h_qk, h_part;
    if (getenv("WHISPER_AMD_MEGA_DBG")) { (void) hipMalloc((void **) &d_qk, (size_t) H * T * 4); h_qk.resize((size_t) hp.n_text_layer * H * T); h_part.resize((size_t) hp.n_text_layer * H * 2048); }
    auto grab = [&](unsigned * dst, const void * src, size_t bytes) { (void) hipStreamSynchronize(s); (void) hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); };
    wa_launch_dec_embed(s, st.d_tok, st.d_pos, 1, d, m.d_te, m.d_pe, st.d_dx);
    const float KQscale = pow(float(64), -0.25);
    const size_t kv_layer = (size_t) kv.size * d, cross_layer = (size_t) H * st.cross_tpad * 64;
    for (int il = 0; il < hp.n_text_layer; ++il) {
        const auto & L = m.dec[il];
        {
            wa_epi e; e.bias = L.qkv.b; e.scale = L.qkv.s; e.out = st.d_dq; e.ldo = d;
            e.out2 = kv.k + il * kv_layer; e.ldo2 = d; e.out3 = kv.v + il * kv_layer; e.ldo3 = d;
            e.split0 = d; e.split1 = 2 * d; e.row_off = kv_head;
            ln_linear(s, true, WA_EPI_DEC_QKV, st.d_dx, d, L.attn_ln, hp.eps, st.d_dxn, L.qkv, 1, nullptr, e);
        }
        grab(out(il, 0), st.d_dq, (size_t) d * 2);
        grab(out(il, 0) + d / 2, kv.k + il * kv_layer + (size_t) kv_head * d, (size_t) d * 2);
        grab(out(il, 0) + d, kv.v + il * kv_layer + (size_t) kv_head * d, (size_t) d * 2);
        wa_launch_attn_exact(s, st.d_dq, d, kv.k + il * kv_layer, 64, d, kv.v + il * kv_layer, 64, d, H, 1, n_kv, nullptr, 1.0f,
                             st.d_att_partial, st.d_att_pleft, st.d_dao, d, nullptr, nullptr);
        grab(out(il, 1), st.d_dao, (size_t) d * 2);
        { wa_epi e; e.bias = L.out.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d; linear(s, true, WA_EPI_RESID, st.d_dao, d, L.out, 1, e); }
        grab(out(il, 2), st.d_dx, (size_t) d * 4);
        { wa_epi e; e.bias = L.cross_q.b; e.out = st.d_dq; e.ldo = d; ln_linear(s, true, WA_EPI_F16, st.d_dx, d, L.cross_ln, hp.eps, st.d_dxn, L.cross_q, 1, nullptr, e); }
        grab(out(il, 3), st.d_dq, (size_t) d * 2);
        wa_launch_attn_exact(s, st.d_dq, d, st.d_cross_k + il * cross_layer, (size_t) st.cross_tpad * 64, 64, st.d_cross_v + il * cross_layer,
                             (size_t) st.cross_tpad * 64, 64, H, 1, T, nullptr, KQscale, st.d_att_partial, st.d_att_pleft, st.d_dao, d, d_qk);
        grab(out(il, 4), st.d_dao, (size_t) d * 2);
        if (d_qk) (void) hipMemcpy(h_qk.data() + (size_t) il * H * T, d_qk, (size_t) H * T * 4, hipMemcpyDeviceToHost);
        if (d_qk) (void) hipMemcpy(h_part.data() + (size_t) il * H * 2048, st.d_att_partial, (size_t) H * 2048 * 4, hipMemcpyDeviceToHost);
        { wa_epi e; e.bias = L.cross_out.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d; linear(s, true, WA_EPI_RESID, st.d_dao, d, L.cross_out, 1, e); }
        grab(out(il, 5), st.d_dx, (size_t) d * 4);
        { wa_epi e; e.bias = L.fc1.b; e.gelu = m.d_gelu; e.out = st.d_dff; e.ldo = 4 * d; ln_linear(s, true, WA_EPI_GELU_F16, st.d_dx, d, L.mlp_ln, hp.eps, st.d_dxn, L.fc1, 1, nullptr, e); }
        grab(out(il, 6), st.d_dff, (size_t) 4 * d * 2);
        { wa_epi e; e.bias = L.fc2.b; e.out = st.d_dx; e.ldo = d; e.resid = st.d_dx; e.ldr = d; linear(s, true, WA_EPI_RESID, st.d_dff, 4 * d, L.fc2, 1, e); }
        grab(out(il, 7), st.d_dx, (size_t) d * 4);
    }
    {
        wa_epi e; e.out = st.d_logits; e.ldo = hp.n_vocab;
        wa_launch_ln_gemv_exact(s, WA_EPI_F32, st.d_dx, d, st.d_rows, m.d_ln.w, m.d_ln.b, hp.eps, m.d_te, d, 1, hp.n_vocab, d, e);
    }
    if (logits_out) grab((unsigned *) logits_out, st.d_logits, (size_t) hp.n_vocab * 4);
    if (d_qk) { FILE * f = fopen("gpurun_out/seq_dbg.bin", "wb"); if (f) { fwrite(h_qk.data(), 4, h_qk.size(), f); fclose(f); } (void) hipFree(d_qk); }
    if (d_qk) { FILE * f = fopen("gpurun_out/seq_part.bin", "wb"); if (f) { fwrite(h_part.data(), 4, h_part.size(), f); fclose(f); } }
    return 0;
}

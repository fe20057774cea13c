// wa_encode.cpp - state allocation, log-mel and the encoder pass (conv stem + L x (MHSA, FFN) +
// cross K/V precompute), as a fixed launch sequence on the state's HIP stream.
//
// ref: whisper_init_state whisper.cpp:3390-3561 (allocate once per state), log_mel_spectrogram
// whisper.cpp:3186-3276, whisper_encode_internal whisper.cpp:2376-2472 and the three graph builders
// whisper.cpp:1994-2364.  There is no graph IR here: the "graph" is this function.
#include "wa_internal.h"
#include "wa_kernels.h"
#include "wa_mega.h"
#include "wa_rows.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

template <typename T> static bool dev_alloc(T *& p, size_t n_elem, bool zero = true) {
    p = nullptr;
    const size_t bytes = n_elem * sizeof(T);
    if (!WA_HIP_OK(hipMalloc((void **) &p, bytes ? bytes : 256))) return false;
    if (zero && bytes) return WA_HIP_OK(hipMemset(p, 0, bytes));
    return true;
}
template <typename T> static void dev_free(T *& p) { if (p) { (void) hipFree(p); p = nullptr; } }

bool wa_kv_self_realloc(whisper_context & ctx, whisper_state & st, int n_cells) {
    const auto & hp = ctx.model.hp;
    // refused once, where the cache is made (not per decode call): the attention kernels keep a launch's scores / probabilities in LDS
    if (n_cells > WA_ATT_MAXKV) { WA_ERROR("%s: %d KV cells exceed the attention kernels' limit (%d)\n", __func__, n_cells, WA_ATT_MAXKV); return false; }
    if (st.dec_graph) { (void) hipGraphExecDestroy(st.dec_graph); st.dec_graph = nullptr; }
    dev_free(st.kv_self.k);
    dev_free(st.kv_self.v);
    st.kv_self.size = n_cells;
    st.kv_self.head = 0;
    st.kv_self.n = 0;
    st.kv_self.cells.assign(n_cells, wa_kv_cell());
    const size_t n = (size_t) hp.n_text_layer * n_cells * hp.n_text_state;
    if (!dev_alloc(st.kv_self.k, n) || !dev_alloc(st.kv_self.v, n)) return false;
    dev_free(st.d_mask);
    st.d_mask_cap = (size_t) st.dec_mpad * n_cells;
    if (!dev_alloc(st.d_mask, st.d_mask_cap)) return false;
    if (st.h_stage_mask) { (void) hipHostFree(st.h_stage_mask); st.h_stage_mask = nullptr; }
    st.h_mask_cap = st.d_mask_cap;
    if (!WA_HIP_OK(hipHostMalloc((void **) &st.h_stage_mask, st.h_mask_cap))) return false;
    return true;
}

// Buffers of the several-rows one-launch decode step (wa_rows.hip), on first use: only states that decode several rows pay for them.
bool wa_rows_prepare(whisper_context & ctx, whisper_state & st) {
    if (!st.rows_enabled) return false;
    if (st.d_rows_gr) return true;
    const auto & hp = ctx.model.hp;
    const int dt = hp.n_text_state, Ht = hp.n_text_head;
    const size_t row_gr = (size_t) 2 * dt;
    if (!dev_alloc(st.d_rows_gr, (size_t) hp.n_text_layer * WA_MEGA_EDGES * WA_ROWS_MAX * row_gr) ||
        !dev_alloc(st.d_rows_cgr, (size_t) hp.n_text_layer * WA_ROWS_MAX * Ht * WA_ROWS_CGR) || !dev_alloc(st.d_rows_status, 32)) {
        dev_free(st.d_rows_gr); dev_free(st.d_rows_cgr); dev_free(st.d_rows_status);
        st.rows_enabled = false;
        return false;
    }
    return true;
}

bool wa_state_alloc(whisper_context & ctx, whisper_state & st) {
    const auto & hp = ctx.model.hp;
    const int d = hp.n_audio_state;
    st.ctx = &ctx;
    if (!WA_HIP_OK(hipSetDevice(ctx.device))) return false;
    if (!WA_HIP_OK(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking))) return false;

    // ---- encoder ----
    const int T = hp.n_audio_ctx, tpad = wa_pad(T, WA_TPAD);
    st.enc_tpad = tpad;
    st.cross_tpad = tpad;
    if (!dev_alloc(st.d_mel_max, 1)) return false;
    // melT: 1 zero row + 2T frames + zero rows so that the strided conv view (3 rows + K padding) stays in bounds
    if (!dev_alloc(st.d_melT, (size_t) (2 * T + 8) * hp.n_mels + 512)) return false;
    if (!dev_alloc(st.d_h1,   (size_t) (2 * T + 8) * d)) return false;
    if (!dev_alloc(st.d_x,    (size_t) tpad * d)) return false;
    if (!dev_alloc(st.d_xn,   (size_t) tpad * d)) return false;
    if (!dev_alloc(st.d_qk,   (size_t) tpad * 2 * d)) return false;
    if (!dev_alloc(st.d_vt,   (size_t) d * tpad)) return false;
    if ((ctx.exact || ctx.model.wtype != 1) && hp.n_audio_state == hp.n_audio_head * 64)      // probability buffers of the MFMA reference-order attention
        if (!dev_alloc(st.d_attn_p, (size_t) hp.n_audio_head * tpad * tpad) || !dev_alloc(st.d_attn_pl, (size_t) hp.n_audio_head * tpad * 32)) return false;
    if (!dev_alloc(st.d_ao,   (size_t) tpad * d)) return false;
    if (!dev_alloc(st.d_ff,   (size_t) tpad * 4 * d)) return false;
    if (!dev_alloc(st.d_embd_enc,  (size_t) tpad * d)) return false;
    if (!dev_alloc(st.d_embd_conv, (size_t) tpad * d)) return false;
    const size_t n_cross = (size_t) hp.n_text_layer * hp.n_text_head * tpad * 64;
    if (!dev_alloc(st.d_cross_k, n_cross) || !dev_alloc(st.d_cross_v, n_cross)) return false;

    // ---- decoder ----
    const int mpad = wa_pad(hp.n_text_ctx, 64);
    st.dec_mpad = mpad;
    if (!dev_alloc(st.d_tok, mpad) || !dev_alloc(st.d_pos, mpad) || !dev_alloc(st.d_cell, mpad) || !dev_alloc(st.d_rows, mpad)) return false;
    if (!dev_alloc(st.d_dx,   (size_t) mpad * d)) return false;
    if (!dev_alloc(st.d_dxn,  (size_t) mpad * d)) return false;
    if (!dev_alloc(st.d_dqkv, (size_t) mpad * 3 * d)) return false;
    if (!dev_alloc(st.d_dao,  (size_t) mpad * d)) return false;
    if (!dev_alloc(st.d_dff,  (size_t) mpad * 4 * d)) return false;
    if (!dev_alloc(st.d_dq,   (size_t) mpad * d)) return false;
    if (!dev_alloc(st.d_logits, (size_t) WA_MAX_DECODERS * hp.n_vocab + 64, false)) return false;
    if (!dev_alloc(st.d_dyn, 8)) return false;
    { const char * g = getenv("WHISPER_AMD_NO_GRAPH"); st.graphs_enabled = !(g && g[0] == '1'); }
    {   // one-launch decode step: needs one workgroup per CU with all of them resident, d_head 64, d <= 1280
        const char * g = getenv("WHISPER_AMD_NO_MEGA");
        const int dt = hp.n_text_state, Ht = hp.n_text_head;
        st.mega_enabled = !(g && g[0] == '1') && ctx.model.d_mega_layers && ctx.model.n_loaded > 0 && dt == Ht * 64 && dt % 128 == 0 &&
                          dt <= WA_MEGA_MAX_D && hp.n_audio_ctx <= 1500 && ctx.model.n_cu >= 5 * Ht + 16 && tpad <= WA_MEGA_MAX_T;
        if (st.mega_enabled) {
            // logits of the step and, right behind them, the status word: one device-to-host copy per token
            if (!dev_alloc(st.d_mega_cgr, (size_t) hp.n_text_layer * Ht * WA_MEGA_CGR)) return false;
            if (!dev_alloc(st.d_mega_gr, (size_t) hp.n_text_layer * WA_MEGA_EDGES * 4 * dt) || !dev_alloc(st.d_mega_out, (size_t) hp.n_vocab + 64)) return false;
            st.d_mega_status = (unsigned *) (st.d_mega_out + hp.n_vocab);
            if (!dev_alloc(st.d_mega_out2, (size_t) hp.n_vocab + 64) || !dev_alloc(st.d_mega_smask, (size_t) hp.n_vocab / 32 + 2)) return false;
            for (int b = 0; b < 2; ++b)
                if (!dev_alloc(st.d_mega_rec[b], (size_t) 512 * 8) || !dev_alloc(st.d_mega_ps[b], 8)) return false;
            {   // the several-rows form shares the one-launch step's preconditions (its buffers come with the first such pass: wa_rows_prepare)
                const char * r = getenv("WHISPER_AMD_NO_ROWS");
                int slot = 0;
                st.rows_enabled = !(r && r[0] == '1') && wa_rows_lds_bytes(dt, 2, std::min(ctx.model.n_cu, 256), ctx.model.wtype != 1 ? 1 : 0, &slot) != 0;
                const char * sr = getenv("WHISPER_AMD_SINGLE_ROWS");
                st.single_via_rows = st.rows_enabled && wa_rows_lds_bytes(dt, 1, std::min(ctx.model.n_cu, 256), ctx.model.wtype != 1 ? 1 : 0, &slot) != 0 &&
                                     (sr ? sr[0] == '1' : (ctx.model.wtype != 1 && dt > 768));
            }
            // (the copy stream, events and pinned buffers of the host overlap are created on first use, wa_spec_begin: a state that
            //  only ever runs inside whisper_amd_full_batch keeps ONE stream - extra streams cost the concurrent chunks their overlap)
        }
    }
    if (!dev_alloc(st.d_att_partial, (size_t) 512 * 32 * 64) || !dev_alloc(st.d_att_pleft, (size_t) 512 * 32)) return false;
    if (ctx.model.wtype != 1) {
        if (!dev_alloc(st.d_q32a, (size_t) tpad * d) || !dev_alloc(st.d_q32b, (size_t) tpad * 4 * d) || !dev_alloc(st.d_q8, (size_t) tpad * 4 * d) ||
            !dev_alloc(st.d_q8d, (size_t) tpad * 4 * d / 32)) return false;
        st.q8_rows = tpad;
    }
    if (!dev_alloc(st.d_im2col, std::max((size_t) 2 * T * 3 * hp.n_mels, (size_t) T * 3 * d) + 64)) return false;
    if (!WA_HIP_OK(hipHostMalloc((void **) &st.h_stage_i32, (size_t) 4 * mpad * sizeof(int32_t)))) return false;
    st.h_logits_cap = (size_t) WA_MAX_DECODERS * hp.n_vocab + 64;
    if (!WA_HIP_OK(hipHostMalloc((void **) &st.h_logits_pinned, st.h_logits_cap * sizeof(float)))) return false;

    // self-attention KV: n_text_ctx padded to 256 cells (whisper.cpp:3403-3406)
    st.kv_self_n_dec = 1;
    if (!wa_kv_self_realloc(ctx, st, wa_pad(hp.n_text_ctx, 256))) return false;

    st.decoders[0].rng = std::mt19937(0);   // whisper.cpp:3486
    WA_INFO("%s: kv self size  = %7.2f MB\n", __func__, 2.0 * hp.n_text_layer * st.kv_self.size * d * 2 / 1e6);
    WA_INFO("%s: kv cross size = %7.2f MB\n", __func__, 2.0 * n_cross * 2 / 1e6);
    return true;
}

void wa_state_release(whisper_state & st) {
    if (st.ctx) (void) hipSetDevice(st.ctx->device);
    if (st.stream) (void) hipStreamSynchronize(st.stream);
    dev_free(st.d_mel); dev_free(st.d_pcm); dev_free(st.d_mel_max);
    dev_free(st.d_melT); dev_free(st.d_h1); dev_free(st.d_x); dev_free(st.d_xn); dev_free(st.d_qk); dev_free(st.d_vt); dev_free(st.d_attn_p); dev_free(st.d_attn_pl);
    dev_free(st.d_ao); dev_free(st.d_ff); dev_free(st.d_embd_enc); dev_free(st.d_embd_conv);
    dev_free(st.d_cross_k); dev_free(st.d_cross_v);
    dev_free(st.kv_self.k); dev_free(st.kv_self.v);
    dev_free(st.d_tok); dev_free(st.d_pos); dev_free(st.d_cell); dev_free(st.d_rows); dev_free(st.d_mask);
    dev_free(st.d_dx); dev_free(st.d_dxn); dev_free(st.d_dqkv); dev_free(st.d_dao); dev_free(st.d_dff); dev_free(st.d_dq);
    if (st.dec_graph) { (void) hipGraphExecDestroy(st.dec_graph); st.dec_graph = nullptr; }
    dev_free(st.d_q32a); dev_free(st.d_q32b); dev_free(st.d_q8); dev_free(st.d_q8d);
    dev_free(st.d_mega_gr); dev_free(st.d_mega_cgr); dev_free(st.d_mega_out); st.d_mega_status = nullptr;
    dev_free(st.d_mega_out2); dev_free(st.d_mega_smask);
    dev_free(st.d_rows_gr); dev_free(st.d_rows_cgr); dev_free(st.d_rows_status);
    for (int b = 0; b < 2; ++b) {
        dev_free(st.d_mega_rec[b]); dev_free(st.d_mega_ps[b]);
        if (st.h_spec[b]) { (void) hipHostFree(st.h_spec[b]); st.h_spec[b] = nullptr; }
        if (st.ev_k[b]) { (void) hipEventDestroy(st.ev_k[b]); st.ev_k[b] = nullptr; }
        if (st.ev_c[b]) { (void) hipEventDestroy(st.ev_c[b]); st.ev_c[b] = nullptr; }
    }
    if (st.copy_stream) { (void) hipStreamDestroy(st.copy_stream); st.copy_stream = nullptr; }
    dev_free(st.d_dyn); dev_free(st.d_att_partial); dev_free(st.d_att_pleft); dev_free(st.d_im2col); dev_free(st.d_logits); dev_free(st.d_aheads_qk);
    if (st.h_stage_i32)     { (void) hipHostFree(st.h_stage_i32);     st.h_stage_i32 = nullptr; }
    if (st.h_stage_mask)    { (void) hipHostFree(st.h_stage_mask);    st.h_stage_mask = nullptr; }
    if (st.h_logits_pinned) { (void) hipHostFree(st.h_logits_pinned); st.h_logits_pinned = nullptr; }
    if (st.stream) { (void) hipStreamDestroy(st.stream); st.stream = nullptr; }
}

// -------------------------------------------------------------------------------------------------
// log-mel
// -------------------------------------------------------------------------------------------------
static bool mel_reserve(whisper_state & st, size_t n) {
    if (n <= st.d_mel_cap) return true;
    dev_free(st.d_mel);
    st.d_mel_cap = 0;
    if (!dev_alloc(st.d_mel, n, false)) return false;
    st.d_mel_cap = n;
    return true;
}

bool wa_mel_compute(whisper_context & ctx, whisper_state & st, const float * samples, int n_samples) {
    const int64_t t0 = wa_time_us();
    const auto & m = ctx.model;
    if (n_samples <= 0 || !samples) return false;
    if (!WA_HIP_OK(hipSetDevice(ctx.device))) return false;

    // frame counts (whisper.cpp:3205-3225): 30 s of zero padding + 200 reflected samples each side
    const int64_t padded = (int64_t) n_samples + 480000 + 400;
    st.mel_n_mel     = m.n_mel_filt;
    st.mel_n_len     = (int) ((padded - 400) / 160);
    st.mel_n_len_org = 1 + (n_samples + 200 - 400) / 160;
    if (!mel_reserve(st, (size_t) st.mel_n_mel * st.mel_n_len)) return false;

    // the samples may already live in HBM (bench / chunk sharding) - then no copy at all
    const float * d_pcm = nullptr;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, samples) == hipSuccess && attr.type == hipMemoryTypeDevice) {
        d_pcm = samples;
    } else {
        (void) hipGetLastError();   // clear the "invalid value" left by a plain host pointer
        if ((size_t) n_samples > st.d_pcm_cap) {
            dev_free(st.d_pcm);
            st.d_pcm_cap = 0;
            if (!dev_alloc(st.d_pcm, (size_t) n_samples, false)) return false;
            st.d_pcm_cap = n_samples;
        }
        if (!WA_HIP_OK(hipMemcpyAsync(st.d_pcm, samples, (size_t) n_samples * sizeof(float), hipMemcpyHostToDevice, st.stream))) return false;
        d_pcm = st.d_pcm;
    }
    wa_launch_mel(st.stream, d_pcm, n_samples, m.d_hann, m.d_sincos, m.d_filters, st.mel_n_mel, m.n_fft_filt, st.d_mel, st.mel_n_len,
                  st.d_mel_max);
    if (!WA_HIP_OK(hipStreamSynchronize(st.stream))) return false;   // host PCM buffer may be released by the caller
    st.t_mel_us += wa_time_us() - t0;
    return true;
}

bool wa_mel_set(whisper_context & ctx, whisper_state & st, const float * data, int n_len, int n_mel) {
    if (n_mel != ctx.model.n_mel_filt) {
        WA_ERROR("%s: invalid number of mel bands: %d (expected %d)\n", __func__, n_mel, ctx.model.n_mel_filt);
        return false;
    }
    if (!WA_HIP_OK(hipSetDevice(ctx.device))) return false;
    st.mel_n_len = n_len; st.mel_n_len_org = n_len; st.mel_n_mel = n_mel;
    if (!mel_reserve(st, std::max<size_t>((size_t) n_len * n_mel, 1))) return false;     // n_len == 0 is legal: an all-zero window
    if ((size_t) n_len * n_mel)                                                            // (whisper.cpp:2399-2421; examples/bench/bench.cpp:82)
        if (!WA_HIP_OK(hipMemcpy(st.d_mel, data, (size_t) n_len * n_mel * sizeof(float), hipMemcpyDefault))) return false;
    return true;
}

// -------------------------------------------------------------------------------------------------
// encoder
// -------------------------------------------------------------------------------------------------
std::atomic<int> & wa_encoders_in_flight(int device) {
    static std::atomic<int> n[64];
    return n[device >= 0 && device < 64 ? device : 0];
}

bool wa_encode(whisper_context & ctx, whisper_state & st, int mel_offset, ggml_abort_callback abort_cb, void * abort_data) {
    const int64_t t0 = wa_time_us();
    struct in_flight { std::atomic<int> & n; in_flight(std::atomic<int> & a) : n(a) { n.fetch_add(1); } ~in_flight() { n.fetch_sub(1); } } busy(wa_encoders_in_flight(ctx.device));
    const auto & m  = ctx.model;
    const auto & hp = m.hp;
    if (!WA_HIP_OK(hipSetDevice(ctx.device))) return false;
    if (m.n_loaded == 0) {           // header-only test model: nothing to compute (whisper.cpp:1959-1960)
        st.t_encode_us += wa_time_us() - t0; st.n_encode++;
        return !(abort_cb && abort_cb(abort_data));
    }
    if (!st.d_mel || st.mel_n_mel != hp.n_mels) { WA_ERROR("%s: no mel spectrogram\n", __func__); return false; }

    const int d = hp.n_audio_state, H = hp.n_audio_head;
    const int T = st.exp_n_audio_ctx > 0 ? st.exp_n_audio_ctx : hp.n_audio_ctx;     // whisper.cpp:2000
    const int tpad = st.enc_tpad;
    hipStream_t s = st.stream;
    st.enc_n_ctx = T;

    // mel window [offset, offset + 2T) -> time-major F16 (zero rows around it = conv padding)
    const int rows_total = 2 * T + 8;
    wa_launch_mel_window(s, st.d_mel, hp.n_mels, st.mel_n_len, mel_offset, 2 * T, st.d_melT, rows_total);

    // Two interchangeable implementations of every dense product:
    //   exact (flash_attn == false): reference summation order on the F32 matrix cores -> bit-identical to whisper.cpp CPU
    //   fast  (flash_attn == true) : MFMA (v_mfma_f32_16x16x32_f16), same rounding points, different F32 order
    const bool exact = ctx.exact || m.wtype != 1;       // (quantised models: the conv stem stays F16 and runs in the reference order)
    auto gemm = [&](wa_epi_mode mode, const wa_f16 * A, int lda, const wa_f16 * W, int ldw, int M, int N, int K, const wa_epi & e) {
        if (exact) wa_launch_gemm_exact(s, mode, A, lda, W, ldw, M, N, K, e);
        else       wa_launch_gemm(s, mode, A, lda, W, ldw, M, N, K, e);
    };

    // conv1 (k3 s1 p1) + bias + GELU -> h1 (F16, time-major, one zero row in front = conv2's left padding)
    {
        wa_epi e; e.bias = m.conv1.b; e.gelu = m.d_gelu; e.out = st.d_h1 + d; e.ldo = d;
        if (exact) {   // im2col in ggml's column order ic*3+k (ops.cpp:5925-5937), K = 3*n_mels with F64 leftovers
            wa_launch_im2col3(s, st.d_melT, hp.n_mels, 0, 1, hp.n_mels, 2 * T, st.d_im2col, 3 * hp.n_mels);
            wa_launch_gemm_exact(s, WA_EPI_GELU_F16, st.d_im2col, 3 * hp.n_mels, m.conv1_g, 3 * hp.n_mels, 2 * T, d, 3 * hp.n_mels, e);
        } else {       // out row t reads melT rows t..t+2 == one contiguous run of the k-major weights: no im2col buffer
            wa_launch_gemm(s, WA_EPI_GELU_F16, st.d_melT, hp.n_mels, m.conv1.w, m.conv1_kpad, 2 * T, d, m.conv1_kpad, e);
        }
    }
    // conv2 (k3 s2 p1) + bias + GELU, then + positional embedding -> residual stream x (F32)
    {
        wa_epi e; e.bias = m.conv2.b; e.gelu = m.d_gelu; e.out = st.d_x; e.ldo = d; e.resid = m.e_pe; e.ldr = d; e.dbg = st.d_embd_conv;
        if (exact) {
            wa_launch_im2col3(s, st.d_h1, d, 0, 2, d, T, st.d_im2col, 3 * d);
            wa_launch_gemm_exact(s, WA_EPI_CONV2, st.d_im2col, 3 * d, m.conv2_g, 3 * d, T, d, 3 * d, e);
        } else {       // out row t reads h1 rows 2t..2t+2 (with the +1 row shift) as a strided view
            wa_launch_gemm(s, WA_EPI_CONV2, st.d_h1, 2 * d, m.conv2.w, 3 * d, T, d, 3 * d, e);
        }
    }

    const float KQscale = 1.0f / sqrtf(float(64));    // whisper.cpp:2087
    if (m.wtype != 1) {
        // Quantised model: every 2-D weight is Q5_0 / Q8_0 and the reference multiplies it with the Q8_0 form of the F32
        // activation row (wa_quant.hip).  Same sequence as below; the operands of the products stay F32 until they are quantised.
        auto qmul = [&](wa_epi_mode mode, const wa_lin & L, int M, const wa_epi & e) {       // operand rows already in d_q8 / d_q8d
            wa_launch_qgemm_exact(s, mode, st.d_q8, st.d_q8d, M, L.qs, L.qd, L.n_out, L.n_in, e);
        };
        auto qlin = [&](wa_epi_mode mode, const float * A, int lda, const wa_lin & L, int M, const wa_epi & e) {
            wa_launch_quantize_q8_0(s, A, lda, M, L.n_in, st.d_q8, st.d_q8d);
            qmul(mode, L, M, e);
        };
        for (int il = 0; il < hp.n_audio_layer; ++il) {
            const auto & L = m.enc[il];
            // (LayerNorm and attention quantise their F32 result rows themselves: wa_q8_store)
            wa_launch_layernorm_exact(s, st.d_x, d, T, d, L.attn_ln.w, L.attn_ln.b, hp.eps, nullptr, 0, nullptr, 0, st.d_q8, st.d_q8d);
            static const bool no_mfma_attn = getenv("WHISPER_AMD_NO_EXACT_MFMA") != nullptr;
            if (st.d_attn_p && T >= 128 && !no_mfma_attn) {
                // q, k, v are F16 here whatever the weight type (whisper.cpp:2181-2202): the reference-order attention on the matrix cores as for an
                // F16 model (Q | K row-major, V transposed); its F32 result is quantised for the out-projection by one more launch
                { wa_epi e; e.bias = L.qkv.b; e.out = st.d_qk; e.ldo = 2 * d; e.out2 = st.d_vt; e.ldo2 = tpad; e.split0 = 2 * d; qmul(WA_EPI_ENC_QKV, L.qkv, T, e); }
                wa_launch_attn_exact_mfma(s, st.d_qk, 2 * d, st.d_vt, tpad, T, d, H, KQscale, st.d_attn_p, st.d_attn_pl, (T + 127) & ~127, st.d_ao, d, st.d_q32a);
                wa_launch_quantize_q8_0(s, st.d_q32a, d, T, d, st.d_q8, st.d_q8d);
            } else {
            { wa_epi e; e.bias = L.qkv.b; e.out = st.d_ff; e.ldo = 3 * d; qmul(WA_EPI_F16, L.qkv, T, e); }
            wa_launch_attn_exact(s, st.d_ff, 3 * d, st.d_ff + d, 64, 3 * d, st.d_ff + 2 * d, 64, 3 * d, H, T, T, nullptr, KQscale,
                                 st.d_att_partial, st.d_att_pleft, st.d_ao, d, nullptr, nullptr, nullptr, st.d_q8, st.d_q8d);
            }
            { wa_epi e; e.bias = L.out.b; e.out = st.d_x; e.ldo = d; e.resid = st.d_x; e.ldr = d; qmul(WA_EPI_RESID, L.out, T, e); }
            wa_launch_layernorm_exact(s, st.d_x, d, T, d, L.mlp_ln.w, L.mlp_ln.b, hp.eps, nullptr, 0, nullptr, 0, st.d_q8, st.d_q8d);
            { wa_epi e; e.bias = L.fc1.b; e.gelu = m.d_gelu; e.out = st.d_q32b; e.ldo = 4 * d; qmul(WA_EPI_GELU_F32, L.fc1, T, e); }
            { wa_epi e; e.bias = L.fc2.b; e.out = st.d_x; e.ldo = d; e.resid = st.d_x; e.ldr = d; qlin(WA_EPI_RESID, st.d_q32b, 4 * d, L.fc2, T, e); }
        }
        wa_launch_layernorm_exact(s, st.d_x, d, T, d, m.e_ln.w, m.e_ln.b, hp.eps, nullptr, 0, st.d_embd_enc, d);
        {
            wa_epi e; e.bias = m.cross_kv.b; e.scale = m.cross_kv.s; e.out = st.d_cross_k; e.out2 = st.d_cross_v; e.aux0 = st.cross_tpad; e.aux1 = d;
            qlin(WA_EPI_CROSS_KV, st.d_embd_enc, d, m.cross_kv, T, e);
        }
    } else {
    for (int il = 0; il < hp.n_audio_layer; ++il) {
        const auto & L = m.enc[il];
        if (exact) wa_launch_layernorm_exact(s, st.d_x, d, T, d, L.attn_ln.w, L.attn_ln.b, hp.eps, st.d_xn, d, nullptr, 0);
        else       wa_launch_layernorm(s, st.d_x, d, T, d, L.attn_ln.w, L.attn_ln.b, hp.eps, st.d_xn, d, nullptr, 0);
        static const bool no_mfma_attn = getenv("WHISPER_AMD_NO_EXACT_MFMA") != nullptr;
        if (exact && st.d_attn_p && T >= 128 && !no_mfma_attn) {      // reference order on the matrix cores: Q | K row-major, V transposed, P through HBM
            wa_epi e; e.bias = L.qkv.b; e.out = st.d_qk; e.ldo = 2 * d; e.out2 = st.d_vt; e.ldo2 = tpad; e.split0 = 2 * d;
            wa_launch_gemm_exact(s, WA_EPI_ENC_QKV, st.d_xn, d, L.qkv.w, d, T, 3 * d, d, e);
            wa_launch_attn_exact_mfma(s, st.d_qk, 2 * d, st.d_vt, tpad, T, d, H, KQscale, st.d_attn_p, st.d_attn_pl, (T + 127) & ~127, st.d_ao, d);
        } else if (exact) {   // Q | K | V row-major in one [T][3d] buffer (the FFN buffer is free here)
            wa_epi e; e.bias = L.qkv.b; e.out = st.d_ff; e.ldo = 3 * d;
            wa_launch_gemm_exact(s, WA_EPI_F16, st.d_xn, d, L.qkv.w, d, T, 3 * d, d, e);
            wa_launch_attn_exact(s, st.d_ff, 3 * d, st.d_ff + d, 64, 3 * d, st.d_ff + 2 * d, 64, 3 * d, H, T, T, nullptr, KQscale,
                                 st.d_att_partial, st.d_att_pleft, st.d_ao, d, nullptr);
        } else {       // V lands transposed for the MFMA P V product
            wa_epi e; e.bias = L.qkv.b; e.out = st.d_qk; e.ldo = 2 * d; e.out2 = st.d_vt; e.ldo2 = tpad; e.split0 = 2 * d;
            wa_launch_gemm(s, WA_EPI_ENC_QKV, st.d_xn, d, L.qkv.w, d, T, 3 * d, d, e);
            wa_launch_enc_attn(s, st.d_qk, 2 * d, st.d_vt, tpad, T, d, H, KQscale, st.d_ao, d);
        }
        {   // out projection + bias + residual
            wa_epi e; e.bias = L.out.b; e.out = st.d_x; e.ldo = d; e.resid = st.d_x; e.ldr = d;
            gemm(WA_EPI_RESID, st.d_ao, d, L.out.w, d, T, d, d, e);
        }
        if (exact) wa_launch_layernorm_exact(s, st.d_x, d, T, d, L.mlp_ln.w, L.mlp_ln.b, hp.eps, st.d_xn, d, nullptr, 0);
        else       wa_launch_layernorm(s, st.d_x, d, T, d, L.mlp_ln.w, L.mlp_ln.b, hp.eps, st.d_xn, d, nullptr, 0);
        {
            wa_epi e; e.bias = L.fc1.b; e.gelu = m.d_gelu; e.out = st.d_ff; e.ldo = 4 * d;
            gemm(WA_EPI_GELU_F16, st.d_xn, d, L.fc1.w, d, T, 4 * d, d, e);
        }
        {
            wa_epi e; e.bias = L.fc2.b; e.out = st.d_x; e.ldo = d; e.resid = st.d_x; e.ldr = d;
            gemm(WA_EPI_RESID, st.d_ff, 4 * d, L.fc2.w, 4 * d, T, d, 4 * d, e);
        }
    }
    // ln_post -> F32 encoder output (API / tests) + F16 copy (operand of the cross K/V GEMM)
    if (exact) wa_launch_layernorm_exact(s, st.d_x, d, T, d, m.e_ln.w, m.e_ln.b, hp.eps, st.d_xn, d, st.d_embd_enc, d);
    else       wa_launch_layernorm(s, st.d_x, d, T, d, m.e_ln.w, m.e_ln.b, hp.eps, st.d_xn, d, st.d_embd_enc, d);

    // cross-attention K/V of ALL decoder layers in one GEMM (whisper.cpp:2290-2364):
    // K = (Wk enc) * d_h^-1/4, V = Wv enc + b, both F16, laid out [layer][head][t][64]
    {
        wa_epi e; e.bias = m.cross_kv.b; e.scale = m.cross_kv.s; e.out = st.d_cross_k; e.out2 = st.d_cross_v; e.aux0 = st.cross_tpad; e.aux1 = d;
        gemm(WA_EPI_CROSS_KV, st.d_xn, d, m.cross_kv.w, d, T, hp.n_text_layer * 2 * d, d, e);
    }
    }
    if (!WA_HIP_OK(hipStreamSynchronize(s))) return false;
    st.have_enc = true;
    st.t_encode_us += wa_time_us() - t0;
    st.n_encode++;
    return !(abort_cb && abort_cb(abort_data));
}

// wa_loader.cpp - legacy-ggml model file parser -> HBM-resident weights.
//
// Format (ref: whisper.cpp:1503-1974 loader, models/convert-pt-to-ggml.py:266-340 writer):
//   u32 magic 0x67676d6c | 11 x i32 hparams | i32 n_mel, i32 n_fft, f32 filters[n_mel*n_fft] |
//   i32 n_vocab_file, (u32 len, bytes)* | tensor records (i32 n_dims, i32 name_len, i32 ttype,
//   i32 ne[n_dims] fastest-first, name, raw data) until EOF.
// Unlike the reference (one ggml tensor per record, generic layouts) the weights land in ONE device
// arena in the layouts the kernels want:
//   * encoder/decoder q,k,v fused into [3d][d] (+ bias / per-column scale vectors),
//   * all decoder layers' cross-attention k,v fused into [L*2d][d] so the cross K/V of a chunk is
//     ONE GEMM over the encoder output,
//   * conv weights re-ordered k-major ([oc][k][ic]) so conv1/conv2 read the activations as a plain
//     strided view (no im2col buffer), conv1's K padded to a multiple of 32 with zeros.
#include "wa_internal.h"
#include "wa_mega.h"

#include <cmath>
#include <cstring>

namespace {

template <typename T> bool rd(whisper_model_loader * l, T & v) { return l->read(l->context, &v, sizeof(T)) == sizeof(T); }

inline float  h2f_host(wa_f16 h) { _Float16 v; memcpy(&v, &h, 2); return (float) v; }
inline wa_f16 f2h_host(float f)  { _Float16 v = (_Float16) f; wa_f16 h; memcpy(&h, &v, 2); return h; }

// ggml_gelu_f32 (vec.h:552-554) -> F16 table (ggml-cpu.c:3509-3517)
inline float gelu_f32(float x) {
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}

struct slot {             // where a named tensor goes
    int    kind;          // 0 raw copy, 1 conv weight re-order ([oc][ic][3] -> [oc][3][ic], row stride ld),
                          // 2 quantised blocks -> kernel layout (off = signed quant bytes [row][8][block][4], off3 = scales [row][block])
    size_t off;           // byte offset in the arena
    int    type;          // expected ggml type: 0 f32, 1 f16
    int64_t ne[3];        // expected ne[] (fastest first)
    int    ld;            // kind 1: destination row stride in elements
    size_t off2 = 0;      // kind 1: byte offset of the second, unpermuted copy (ggml im2col order)
    size_t off3 = 0;
    bool   seen = false;
};

struct arena_builder {
    size_t size = 0;
    size_t take(size_t bytes) { size_t o = size; size += (bytes + 255) & ~size_t(255); return o; }
};

} // namespace

void wa_model_free(whisper_context & ctx) {
    if (ctx.model.d_mega_layers) { (void) hipFree(ctx.model.d_mega_layers); ctx.model.d_mega_layers = nullptr; }
    if (ctx.model.arena) { (void) hipFree(ctx.model.arena); ctx.model.arena = nullptr; }
}

bool wa_model_load(whisper_model_loader * loader, whisper_context & wctx) {
    WA_INFO("%s: loading model\n", __func__);
    const int64_t t_start = wa_time_us();
    wctx.t_start_us = t_start;

    auto & model = wctx.model;
    auto & vocab = wctx.vocab;
    auto & hp    = model.hp;

    {   // magic
        uint32_t magic = 0;
        rd(loader, magic);
        if (magic != 0x67676d6c) { WA_ERROR("%s: invalid model data (bad magic)\n", __func__); return false; }
    }
    {   // hparams, in file order (whisper.cpp:1527-1537)
        int32_t * f[11] = { &hp.n_vocab, &hp.n_audio_ctx, &hp.n_audio_state, &hp.n_audio_head, &hp.n_audio_layer, &hp.n_text_ctx,
                            &hp.n_text_state, &hp.n_text_head, &hp.n_text_layer, &hp.n_mels, &hp.ftype };
        for (auto p : f) if (!rd(loader, *p)) { WA_ERROR("%s: truncated header\n", __func__); return false; }

        model.type = 0;
        switch (hp.n_audio_layer) { case 4: model.type = 1; break; case 6: model.type = 2; break; case 12: model.type = 3; break;
                                    case 24: model.type = 4; break; case 32: model.type = 5; break; }
        const int qntvr = hp.ftype / 1000;   // GGML_QNT_VERSION_FACTOR
        hp.ftype %= 1000;
        WA_INFO("%s: n_vocab       = %d\n", __func__, hp.n_vocab);
        WA_INFO("%s: n_audio_ctx   = %d\n", __func__, hp.n_audio_ctx);
        WA_INFO("%s: n_audio_state = %d\n", __func__, hp.n_audio_state);
        WA_INFO("%s: n_audio_head  = %d\n", __func__, hp.n_audio_head);
        WA_INFO("%s: n_audio_layer = %d\n", __func__, hp.n_audio_layer);
        WA_INFO("%s: n_text_ctx    = %d\n", __func__, hp.n_text_ctx);
        WA_INFO("%s: n_text_state  = %d\n", __func__, hp.n_text_state);
        WA_INFO("%s: n_text_head   = %d\n", __func__, hp.n_text_head);
        WA_INFO("%s: n_text_layer  = %d\n", __func__, hp.n_text_layer);
        WA_INFO("%s: n_mels        = %d\n", __func__, hp.n_mels);
        WA_INFO("%s: ftype         = %d\n", __func__, hp.ftype);
        WA_INFO("%s: qntvr         = %d\n", __func__, qntvr);

        if (hp.n_vocab <= 0 || hp.n_audio_ctx <= 0 || hp.n_audio_state <= 0 || hp.n_audio_head <= 0 || hp.n_audio_layer < 0 ||
            hp.n_text_ctx <= 0 || hp.n_text_state <= 0 || hp.n_text_head <= 0 || hp.n_text_layer < 0 || hp.n_mels <= 0) {
            WA_ERROR("%s: invalid model (bad hparams)\n", __func__);
            return false;
        }
        // upper bounds the kernels' fixed-size LDS and index arithmetic rely on (every real Whisper model: 1500 / 448 / <= 51866 / <= 1280)
        if (hp.n_audio_ctx > 2048 || hp.n_text_ctx > 512 || hp.n_vocab > (1 << 20) || hp.n_audio_state > 8192 || hp.n_text_state > 8192 ||
            hp.n_audio_layer > 256 || hp.n_text_layer > 256 || hp.n_mels > 1024) {
            WA_ERROR("%s: model dimensions beyond what this backend supports (n_audio_ctx %d, n_text_ctx %d, n_vocab %d)\n", __func__, hp.n_audio_ctx,
                     hp.n_text_ctx, hp.n_vocab);
            return false;
        }
        // ftype % 1000 names the type of the 2-D weights (whisper.cpp:1567-1573; ggml_ftype: 1 F16, 7 Q8_0, 8 Q5_0).  ftype 0 (all-F32)
        // aborts in the reference's own conv path; the other quantised formats are not built.
        const int ft = hp.ftype % 1000;
        model.wtype = ft == 1 ? 1 : ft == 8 ? 6 : ft == 7 ? 8 : -1;
        if (model.wtype < 0) {
            WA_ERROR("%s: unsupported ftype %d (this backend loads F16, Q5_0 and Q8_0 models)\n", __func__, hp.ftype);
            return false;
        }
    }
    {   // mel filters
        int32_t n_mel = 0, n_fft = 0;
        rd(loader, n_mel); rd(loader, n_fft);
        if (n_mel <= 0 || n_fft <= 0 || n_mel > 1024 || n_fft > 4096) { WA_ERROR("%s: invalid mel filter header\n", __func__); return false; }
        model.n_mel_filt = n_mel; model.n_fft_filt = n_fft;
        model.filters.resize((size_t) n_mel * n_fft);
        loader->read(loader->context, model.filters.data(), model.filters.size() * sizeof(float));
    }
    {   // vocab (whisper.cpp:1607-1693)
        int32_t n_vocab = 0;
        rd(loader, n_vocab);
        if (n_vocab < 0 || n_vocab > (1 << 24)) { WA_ERROR("%s: invalid vocab size\n", __func__); return false; }
        vocab.id_to_token.assign(std::max(n_vocab, hp.n_vocab), std::string());
        std::vector<char> tmp;
        for (int i = 0; i < n_vocab; ++i) {
            uint32_t len = 0;
            rd(loader, len);
            std::string word;
            if (len > 0) {
                if (len > (1u << 20)) { WA_ERROR("%s: invalid vocab entry\n", __func__); return false; }
                tmp.resize(len);
                loader->read(loader->context, tmp.data(), len);
                word.assign(tmp.data(), len);
            }
            vocab.token_to_id[word] = i;
            vocab.id_to_token[i] = word;
        }
        vocab.n_vocab = hp.n_vocab;
        if (vocab.is_multilingual()) {
            vocab.token_eot++; vocab.token_sot++;
            const int dt = vocab.num_languages() - 98;
            vocab.token_translate += dt; vocab.token_transcribe += dt; vocab.token_solm += dt; vocab.token_prev += dt;
            vocab.token_nosp += dt; vocab.token_not += dt; vocab.token_beg += dt;
        }
        if (n_vocab < hp.n_vocab) {
            WA_INFO("%s: adding %d extra tokens\n", __func__, hp.n_vocab - n_vocab);
            for (int i = n_vocab; i < hp.n_vocab; ++i) {
                std::string word;
                if      (i >  vocab.token_beg)        word = "[_TT_" + std::to_string(i - vocab.token_beg) + "]";
                else if (i == vocab.token_eot)        word = "[_EOT_]";
                else if (i == vocab.token_sot)        word = "[_SOT_]";
                else if (i == vocab.token_translate)  word = "[_TRANSLATE_]";
                else if (i == vocab.token_transcribe) word = "[_TRANSCRIBE_]";
                else if (i == vocab.token_solm)       word = "[_SOLM_]";
                else if (i == vocab.token_prev)       word = "[_PREV_]";
                else if (i == vocab.token_nosp)       word = "[_NOSP_]";
                else if (i == vocab.token_not)        word = "[_NOT_]";
                else if (i == vocab.token_beg)        word = "[_BEG_]";
                else if (i > vocab.token_sot && i <= vocab.token_sot + vocab.num_languages()) {
                    const char * ls = whisper_lang_str(i - vocab.token_sot - 1);
                    word = "[_LANG_" + std::string(ls ? ls : "?") + "]";
                } else word = "[_extra_token_" + std::to_string(i) + "]";
                vocab.token_to_id[word] = i;
                vocab.id_to_token[i] = word;
            }
        }
        WA_INFO("%s: n_langs       = %d\n", __func__, vocab.num_languages());
    }

    const int d = hp.n_audio_state, Le = hp.n_audio_layer, Ld = hp.n_text_layer;
    if (hp.n_text_state != d) { WA_ERROR("%s: n_text_state != n_audio_state is not supported\n", __func__); return false; }
    if (d % 64 != 0 || d / hp.n_audio_head != 64 || d / hp.n_text_head != 64) {
        WA_ERROR("%s: unsupported head size (kernels are built for d_head = 64, as in every Whisper model)\n", __func__);
        return false;
    }
    if (model.n_mel_filt != hp.n_mels) { WA_ERROR("%s: mel filter count %d != n_mels %d\n", __func__, model.n_mel_filt, hp.n_mels); return false; }

    // ---------------------------------------------------------------------------------------------
    // arena plan
    // ---------------------------------------------------------------------------------------------
    arena_builder ab;
    std::map<std::string, slot> slots;
    auto add = [&](const std::string & name, int kind, size_t off, int type, int64_t n0, int64_t n1, int64_t n2, int ld = 0) {
        slot s; s.kind = kind; s.off = off; s.type = type; s.ne[0] = n0; s.ne[1] = n1; s.ne[2] = n2; s.ld = ld;
        slots[name] = s;
    };
    const size_t F = sizeof(float), H = sizeof(wa_f16);

    const size_t o_filters = ab.take((size_t) model.n_mel_filt * model.n_fft_filt * F);
    const size_t o_hann    = ab.take(400 * F);
    const size_t o_sincos  = ab.take(800 * F);
    const size_t o_gelu    = ab.take(65536 * H);

    struct lin_off { size_t w, b, s; size_t qs = 0, qh = 0, qd = 0; };
    const int QT = model.wtype;                                  // 1, 6 or 8
    // Kernel layout of a quantised [n_out][n_in] matrix: the quants as signed bytes (Q5_0's 5-bit values are expanded once, here),
    // ordered [row][lane l = 0..7][block][4] - lane l of a row's 8-lane group owns elements 4l..4l+3 of EVERY block, so its bytes
    // are contiguous over the blocks (16-byte loads cover four blocks) - and the block scales [row][block] as F32.
    auto take_q = [&](lin_off & o, size_t n_out, size_t n_in) {
        o.qs = ab.take(n_out * n_in); o.qh = 0; o.qd = ab.take(n_out * (n_in / 32) * 4);
    };
    auto add_q = [&](const std::string & name, const lin_off & o, size_t row0, int64_t n_in, int64_t n_rows) {
        slot s; s.kind = 2; s.off = o.qs + row0 * (size_t) n_in; s.off2 = 0; s.off3 = o.qd + row0 * (size_t) (n_in / 32) * 4; s.type = QT;
        s.ne[0] = n_in; s.ne[1] = n_rows; s.ne[2] = 1; s.ld = 0;
        slots[name] = s;
    };
    struct ln_off  { size_t w, b; };
    auto take_ln = [&](const std::string & base) {
        ln_off o{ ab.take(d * F), ab.take(d * F) };
        add(base + ".weight", 0, o.w, 0, d, 1, 1);
        add(base + ".bias",   0, o.b, 0, d, 1, 1);
        return o;
    };
    auto take_lin = [&](const std::string & base, int n_out, int n_in) {
        lin_off o{ QT == 1 ? ab.take((size_t) n_out * n_in * H) : 0, ab.take(n_out * F), 0 };
        if (QT == 1) add(base + ".weight", 0, o.w, 1, n_in, n_out, 1);
        else { take_q(o, n_out, n_in); add_q(base + ".weight", o, 0, n_in, n_out); }
        add(base + ".bias",   0, o.b, 0, n_out, 1, 1);
        return o;
    };
    // fused q|k|v block: [3d][d] weights + [3d] bias (+ [3d] scale, filled by us)
    auto take_qkv = [&](const std::string & base, bool with_scale) {
        lin_off o{ QT == 1 ? ab.take((size_t) 3 * d * d * H) : 0, ab.take(3 * d * F), with_scale ? ab.take(3 * d * F) : 0 };
        if (QT == 1) {
            add(base + ".query.weight", 0, o.w,                         1, d, d, 1);
            add(base + ".key.weight",   0, o.w + (size_t) d * d * H,     1, d, d, 1);
            add(base + ".value.weight", 0, o.w + (size_t) 2 * d * d * H, 1, d, d, 1);
        } else {
            take_q(o, 3 * d, d);
            add_q(base + ".query.weight", o, 0, d, d); add_q(base + ".key.weight", o, d, d, d); add_q(base + ".value.weight", o, 2 * d, d, d);
        }
        add(base + ".query.bias",   0, o.b,             0, d, 1, 1);
        add(base + ".value.bias",   0, o.b + 2 * d * F, 0, d, 1, 1);
        return o;
    };

    const size_t o_epe = ab.take((size_t) hp.n_audio_ctx * d * F);
    add("encoder.positional_embedding", 0, o_epe, 0, d, hp.n_audio_ctx, 1);
    model.conv1_kpad = wa_pad(3 * hp.n_mels, 32);
    const size_t o_c1w = ab.take((size_t) d * model.conv1_kpad * H), o_c1b = ab.take(d * F);
    const size_t o_c2w = ab.take((size_t) d * 3 * d * H),            o_c2b = ab.take(d * F);
    const size_t o_c1g = ab.take((size_t) d * 3 * hp.n_mels * H), o_c2g = ab.take((size_t) d * 3 * d * H);
    add("encoder.conv1.weight", 1, o_c1w, 1, 3, hp.n_mels, d, model.conv1_kpad);
    add("encoder.conv1.bias",   0, o_c1b, 0, 1, d, 1);
    add("encoder.conv2.weight", 1, o_c2w, 1, 3, d, d, 3 * d);
    add("encoder.conv2.bias",   0, o_c2b, 0, 1, d, 1);
    slots["encoder.conv1.weight"].off2 = o_c1g;
    slots["encoder.conv2.weight"].off2 = o_c2g;
    const ln_off o_eln = take_ln("encoder.ln_post");

    struct enc_off { ln_off attn_ln, mlp_ln; lin_off qkv, out, fc1, fc2; };
    std::vector<enc_off> eo(Le);
    for (int i = 0; i < Le; ++i) {
        const std::string p = "encoder.blocks." + std::to_string(i) + ".";
        eo[i].attn_ln = take_ln(p + "attn_ln");
        eo[i].qkv     = take_qkv(p + "attn", false);
        eo[i].out     = take_lin(p + "attn.out", d, d);
        eo[i].mlp_ln  = take_ln(p + "mlp_ln");
        eo[i].fc1     = take_lin(p + "mlp.0", 4 * d, d);
        eo[i].fc2     = take_lin(p + "mlp.2", d, 4 * d);
    }

    const size_t o_dpe = ab.take((size_t) hp.n_text_ctx * d * F);
    const size_t o_dte = QT == 1 ? ab.take((size_t) hp.n_vocab * d * H) : 0;
    lin_off o_teq{ 0, 0, 0 };
    add("decoder.positional_embedding",   0, o_dpe, 0, d, hp.n_text_ctx, 1);
    if (QT == 1) add("decoder.token_embedding.weight", 0, o_dte, 1, d, hp.n_vocab, 1);
    else { take_q(o_teq, hp.n_vocab, d); add_q("decoder.token_embedding.weight", o_teq, 0, d, hp.n_vocab); }
    const ln_off o_dln = take_ln("decoder.ln");

    // cross k|v of all layers fused: rows [il*2d, il*2d+d) key, [il*2d+d, (il+1)*2d) value
    const size_t o_ckv_w = QT == 1 ? ab.take((size_t) Ld * 2 * d * d * H) : 0, o_ckv_b = ab.take((size_t) Ld * 2 * d * F),
                 o_ckv_s = ab.take((size_t) Ld * 2 * d * F);
    lin_off o_ckvq{ 0, 0, 0 };
    if (QT != 1) take_q(o_ckvq, (size_t) Ld * 2 * d, d);

    struct dec_off { ln_off attn_ln, cross_ln, mlp_ln; lin_off qkv, out, cq, cout, fc1, fc2; };
    std::vector<dec_off> dof(Ld);
    for (int i = 0; i < Ld; ++i) {
        const std::string p = "decoder.blocks." + std::to_string(i) + ".";
        dof[i].attn_ln  = take_ln(p + "attn_ln");
        dof[i].qkv      = take_qkv(p + "attn", true);
        dof[i].out      = take_lin(p + "attn.out", d, d);
        dof[i].cross_ln = take_ln(p + "cross_attn_ln");
        dof[i].cq       = take_lin(p + "cross_attn.query", d, d);
        dof[i].cout     = take_lin(p + "cross_attn.out", d, d);
        if (QT == 1) {
            add(p + "cross_attn.key.weight",   0, o_ckv_w + ((size_t) i * 2 * d) * d * H,     1, d, d, 1);
            add(p + "cross_attn.value.weight", 0, o_ckv_w + ((size_t) i * 2 * d + d) * d * H, 1, d, d, 1);
        } else {
            add_q(p + "cross_attn.key.weight", o_ckvq, (size_t) i * 2 * d, d, d); add_q(p + "cross_attn.value.weight", o_ckvq, (size_t) i * 2 * d + d, d, d);
        }
        add(p + "cross_attn.value.bias",   0, o_ckv_b + ((size_t) i * 2 * d + d) * F,     0, d, 1, 1);
        dof[i].mlp_ln   = take_ln(p + "mlp_ln");
        dof[i].fc1      = take_lin(p + "mlp.0", 4 * d, d);
        dof[i].fc2      = take_lin(p + "mlp.2", d, 4 * d);
    }

    // ---------------------------------------------------------------------------------------------
    // host staging image of the arena
    // ---------------------------------------------------------------------------------------------
    std::vector<uint8_t> img;
    try { img.assign(ab.size, 0); } catch (...) { WA_ERROR("%s: out of host memory for %zu bytes\n", __func__, ab.size); return false; }

    memcpy(img.data() + o_filters, model.filters.data(), model.filters.size() * F);
    {   // Hann window and sin/cos tables exactly as whisper.cpp:3031-3047 (float cosf/sinf of a double argument)
        float * hann = (float *) (img.data() + o_hann);
        float * sc   = (float *) (img.data() + o_sincos);
        for (int i = 0; i < 400; ++i) {
            hann[i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / 400));
            const double theta = (2 * M_PI * i) / 400;
            sc[i]       = sinf(theta);
            sc[400 + i] = cosf(theta);
        }
        wa_f16 * g = (wa_f16 *) (img.data() + o_gelu);
        for (int i = 0; i < 65536; ++i) g[i] = f2h_host(gelu_f32(h2f_host((wa_f16) i)));
    }
    const float KQscale = pow(float(64), -0.25);   // whisper.cpp:2316, 2522
    for (int i = 0; i < Ld; ++i) {
        float * s = (float *) (img.data() + dof[i].qkv.s);
        for (int c = 0; c < 3 * d; ++c) s[c] = c < 2 * d ? KQscale : 1.0f;
        float * cs = (float *) (img.data() + o_ckv_s) + (size_t) i * 2 * d;
        for (int c = 0; c < 2 * d; ++c) cs[c] = c < d ? KQscale : 1.0f;
    }

    // ---------------------------------------------------------------------------------------------
    // tensor records
    // ---------------------------------------------------------------------------------------------
    size_t total_size = 0;
    model.n_loaded = 0;
    std::vector<uint8_t> tmp;
    while (true) {
        int32_t n_dims = 0, length = 0, ttype = 0;
        rd(loader, n_dims); rd(loader, length); rd(loader, ttype);
        if (loader->eof(loader->context)) break;
        if (n_dims < 1 || n_dims > 3 || length <= 0 || length > 256) { WA_ERROR("%s: corrupt tensor header\n", __func__); return false; }
        int64_t ne[3] = { 1, 1, 1 };
        int64_t nelements = 1;
        for (int i = 0; i < n_dims; ++i) { int32_t v = 0; rd(loader, v); ne[i] = v; nelements *= v; }
        std::string name(length, '\0');
        loader->read(loader->context, &name[0], length);

        auto it = slots.find(name);
        if (it == slots.end()) { WA_ERROR("%s: unknown tensor '%s' in model file\n", __func__, name.c_str()); return false; }
        slot & s = it->second;
        if (nelements != s.ne[0] * s.ne[1] * s.ne[2]) { WA_ERROR("%s: tensor '%s' has wrong size in model file\n", __func__, name.c_str()); return false; }
        if (ne[0] != s.ne[0] || ne[1] != s.ne[1] || ne[2] != s.ne[2]) {
            WA_ERROR("%s: tensor '%s' has wrong shape in model file: got [%d, %d, %d], expected [%d, %d, %d]\n", __func__, name.c_str(),
                     (int) ne[0], (int) ne[1], (int) ne[2], (int) s.ne[0], (int) s.ne[1], (int) s.ne[2]);
            return false;
        }
        if (ttype != s.type) { WA_ERROR("%s: tensor '%s' has type %d in model file, expected %d\n", __func__, name.c_str(), ttype, s.type); return false; }
        const size_t nbytes = s.type == 6 ? (size_t) nelements / 32 * 22 : s.type == 8 ? (size_t) nelements / 32 * 34 : (size_t) nelements * (s.type == 0 ? F : H);
        if (s.kind == 2) {        // block_q5_0 { f16 d; u32 qh; u8 qs[16] } / block_q8_0 { f16 d; i8 qs[32] } (ggml-common.h:187-214) -> arrays
            tmp.resize(nbytes);
            if (loader->read(loader->context, tmp.data(), nbytes) != nbytes) { WA_ERROR("%s: truncated tensor '%s'\n", __func__, name.c_str()); return false; }
            const size_t bsz = s.type == 6 ? 22 : 34, nbr = (size_t) s.ne[0] / 32, rows = (size_t) s.ne[1];
            int8_t * qs = (int8_t *) (img.data() + s.off); float * qd = (float *) (img.data() + s.off3);
            for (size_t r = 0; r < rows; ++r)
                for (size_t b = 0; b < nbr; ++b) {
                    const uint8_t * blk = tmp.data() + (r * nbr + b) * bsz;
                    wa_f16 dh; memcpy(&dh, blk, 2);
                    qd[r * nbr + b] = h2f_host(dh);
                    int8_t v[32];
                    if (s.type == 6) {      // element j < 16: low nibble of qs[j], j + 16: high nibble; bit e of qh: fifth bit; value - 16
                        uint32_t qh; memcpy(&qh, blk + 2, 4);
                        for (int j = 0; j < 16; ++j) {
                            v[j]      = (int8_t) ((int) ((blk[6 + j] & 0x0f) | (((qh >> j) & 1u) << 4)) - 16);
                            v[j + 16] = (int8_t) ((int) ((blk[6 + j] >> 4)   | (((qh >> (j + 16)) & 1u) << 4)) - 16);
                        }
                    } else memcpy(v, blk + 2, 32);
                    for (int l = 0; l < 8; ++l) memcpy(qs + ((r * 8 + l) * nbr + b) * 4, v + 4 * l, 4);
                }
        } else if (s.kind == 0) {
            if (loader->read(loader->context, img.data() + s.off, nbytes) != nbytes) { WA_ERROR("%s: truncated tensor '%s'\n", __func__, name.c_str()); return false; }
        } else {
            tmp.resize(nbytes);
            if (loader->read(loader->context, tmp.data(), nbytes) != nbytes) { WA_ERROR("%s: truncated tensor '%s'\n", __func__, name.c_str()); return false; }
            // file: [oc][ic][k] (k fastest) -> arena: [oc][k*IC + ic], row stride ld
            const int IC = (int) s.ne[1], OC = (int) s.ne[2];
            const wa_f16 * src = (const wa_f16 *) tmp.data();
            wa_f16 * dst = (wa_f16 *) (img.data() + s.off);
            memcpy(img.data() + s.off2, tmp.data(), nbytes);
            for (int oc = 0; oc < OC; ++oc)
                for (int ic = 0; ic < IC; ++ic)
                    for (int k = 0; k < 3; ++k) dst[(size_t) oc * s.ld + k * IC + ic] = src[((size_t) oc * IC + ic) * 3 + k];
        }
        s.seen = true;
        total_size += nbytes;
        model.n_loaded++;
    }
    WA_INFO("%s: model size    = %7.2f MB\n", __func__, total_size / 1e6);
    if (model.n_loaded == 0) {
        WA_WARN("%s: WARN no tensors loaded from model file - assuming empty model for testing\n", __func__);
    } else if (model.n_loaded != (int) slots.size()) {
        WA_ERROR("%s: ERROR not all tensors loaded from model file - expected %zu, got %d\n", __func__, slots.size(), model.n_loaded);
        return false;
    }

    // ---------------------------------------------------------------------------------------------
    // upload
    // ---------------------------------------------------------------------------------------------
    if (!WA_HIP_OK(hipSetDevice(wctx.device))) return false;
    if (!WA_HIP_OK(hipMalloc(&model.arena, ab.size))) return false;
    model.arena_size = ab.size;
    if (!WA_HIP_OK(hipMemcpy(model.arena, img.data(), ab.size, hipMemcpyHostToDevice))) return false;
    WA_INFO("%s: %12s total size = %8.2f MB\n", __func__, "HIP0", ab.size / 1e6);

    uint8_t * base = (uint8_t *) model.arena;
    auto PF = [&](size_t o) { return (const float *) (base + o); };
    auto PH = [&](size_t o) { return (const wa_f16 *) (base + o); };
    auto LN = [&](ln_off o) { wa_ln r; r.w = PF(o.w); r.b = PF(o.b); return r; };
    auto LIN = [&](lin_off o, int n_out, int n_in) {
        wa_lin r; r.w = PH(o.w); r.b = PF(o.b); r.s = o.s ? PF(o.s) : nullptr; r.n_out = n_out; r.n_in = n_in;
        if (o.qs) { r.w = nullptr; r.wtype = model.wtype; r.qs = (const int8_t *) (base + o.qs); r.qd = PF(o.qd); }
        return r;
    };

    model.d_filters = PF(o_filters); model.d_hann = PF(o_hann); model.d_sincos = PF(o_sincos); model.d_gelu = PH(o_gelu);
    model.e_pe = PF(o_epe);
    model.conv1 = LIN(lin_off{ o_c1w, o_c1b, 0 }, d, model.conv1_kpad);
    model.conv2 = LIN(lin_off{ o_c2w, o_c2b, 0 }, d, 3 * d);
    model.conv1_g = PH(o_c1g); model.conv2_g = PH(o_c2g);
    model.e_ln = LN(o_eln);
    model.enc.resize(Le);
    for (int i = 0; i < Le; ++i) {
        model.enc[i].attn_ln = LN(eo[i].attn_ln); model.enc[i].mlp_ln = LN(eo[i].mlp_ln);
        model.enc[i].qkv = LIN(eo[i].qkv, 3 * d, d);  model.enc[i].out = LIN(eo[i].out, d, d);
        model.enc[i].fc1 = LIN(eo[i].fc1, 4 * d, d);  model.enc[i].fc2 = LIN(eo[i].fc2, d, 4 * d);
    }
    model.d_pe = PF(o_dpe); model.d_te = QT == 1 ? PH(o_dte) : nullptr; model.d_ln = LN(o_dln);
    if (QT != 1) { lin_off t = o_teq; t.b = 0; model.te_q = LIN(t, hp.n_vocab, d); model.te_q.b = nullptr; }
    model.dec.resize(Ld);
    for (int i = 0; i < Ld; ++i) {
        auto & L = model.dec[i];
        L.attn_ln = LN(dof[i].attn_ln); L.cross_ln = LN(dof[i].cross_ln); L.mlp_ln = LN(dof[i].mlp_ln);
        L.qkv = LIN(dof[i].qkv, 3 * d, d); L.out = LIN(dof[i].out, d, d);
        L.cross_q = LIN(dof[i].cq, d, d);  L.cross_out = LIN(dof[i].cout, d, d);
        L.fc1 = LIN(dof[i].fc1, 4 * d, d); L.fc2 = LIN(dof[i].fc2, d, 4 * d);
    }
    { lin_off t = o_ckvq; t.w = o_ckv_w; t.b = o_ckv_b; t.s = o_ckv_s; model.cross_kv = LIN(t, Ld * 2 * d, d); }

    {   // per-layer pointer table of the one-launch decode step (wa_mega.hip)
        std::vector<wa_mega_layer> tab(Ld);
        for (int i = 0; i < Ld; ++i) {
            const auto & L = model.dec[i];
            wa_mega_layer & t = tab[i];
            // F16 model: the matrices; quantised model: their signed-byte quants (kernel layout) in the same fields + the block scales
            auto W = [&](const wa_lin & l) { return QT == 1 ? l.w : (const wa_f16 *) l.qs; };
            auto D = [&](const wa_lin & l) { return QT == 1 ? (const float *) nullptr : l.qd; };
            t.ln1_w = L.attn_ln.w;  t.ln1_b = L.attn_ln.b;  t.qkv_w = W(L.qkv); t.qkv_b = L.qkv.b; t.qkv_s = L.qkv.s; t.qkv_d = D(L.qkv);
            t.out_w = W(L.out);     t.out_b = L.out.b;      t.out_d = D(L.out);
            t.ln2_w = L.cross_ln.w; t.ln2_b = L.cross_ln.b; t.cq_w = W(L.cross_q); t.cq_b = L.cross_q.b; t.cq_d = D(L.cross_q);
            t.co_w  = W(L.cross_out); t.co_b = L.cross_out.b; t.co_d = D(L.cross_out);
            t.ln3_w = L.mlp_ln.w;   t.ln3_b = L.mlp_ln.b;   t.fc1_w = W(L.fc1); t.fc1_b = L.fc1.b; t.fc1_d = D(L.fc1);
            t.fc2_w = W(L.fc2);     t.fc2_b = L.fc2.b;      t.fc2_d = D(L.fc2);
        }
        if (Ld > 0) {
            if (!WA_HIP_OK(hipMalloc(&model.d_mega_layers, tab.size() * sizeof(wa_mega_layer)))) return false;
            if (!WA_HIP_OK(hipMemcpy(model.d_mega_layers, tab.data(), tab.size() * sizeof(wa_mega_layer), hipMemcpyHostToDevice))) return false;
        }
        hipDeviceProp_t prop;
        if (WA_HIP_OK(hipGetDeviceProperties(&prop, wctx.device))) model.n_cu = prop.multiProcessorCount;
    }

    wctx.t_load_us = wa_time_us() - t_start;
    return true;
}

// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// This translation unit *includes the reference engine in place* (the file stays under
// /root/reference; nothing is copied) and adds a few extern "C" accessors for internals that the
// public whisper.h API does not expose (SURVEY.md §8c "Reaching internals"): the log-mel buffer,
// the encoder output, the cross/self KV caches.  It is linked instead of whisper.o into
// oracle/_ref/libwhisper_ref.so and is used only by tests/, tools/gen_golden.py and bench.py's
// cpu_baseline leg.
#include REF_WHISPER_CPP

extern "C" {

// mel.data layout [n_mel][n_len] (whisper.cpp:3186-3276)
int ref_shim_mel_n_len(struct whisper_state * st)      { return st->mel.n_len; }
int ref_shim_mel_n_len_org(struct whisper_state * st)  { return st->mel.n_len_org; }
int ref_shim_mel_n_mel(struct whisper_state * st)      { return st->mel.n_mel; }
const float * ref_shim_mel_data(struct whisper_state * st) { return st->mel.data.data(); }

// encoder output embd_enc [n_audio_ctx][n_state] f32 (whisper.cpp:2056-2287)
int ref_shim_get_embd_enc(struct whisper_state * st, float * dst, int n_floats) {
    if (!st->embd_enc) return -1;
    const size_t nb = ggml_nbytes(st->embd_enc);
    if ((size_t) n_floats * sizeof(float) < nb) return -(int)(nb / sizeof(float));
    ggml_backend_tensor_get(st->embd_enc, dst, 0, nb);
    return (int)(nb / sizeof(float));
}

// conv stem output embd_conv [n_state][n_ctx] (time fastest) f32 (whisper.cpp:1994-2054)
int ref_shim_get_embd_conv(struct whisper_state * st, float * dst, int n_floats) {
    if (!st->embd_conv) return -1;
    const size_t nb = ggml_nbytes(st->embd_conv);
    if ((size_t) n_floats * sizeof(float) < nb) return -(int)(nb / sizeof(float));
    ggml_backend_tensor_get(st->embd_conv, dst, 0, nb);
    return (int)(nb / sizeof(float));
}

// raw F16 KV caches (whisper.cpp:3403-3438)
size_t ref_shim_kv_cross_nbytes(struct whisper_state * st, int which) {
    return ggml_nbytes(which == 0 ? st->kv_cross.k : st->kv_cross.v);
}
int ref_shim_get_kv_cross(struct whisper_state * st, int which, void * dst, size_t nbytes) {
    ggml_tensor * t = which == 0 ? st->kv_cross.k : st->kv_cross.v;
    if (nbytes < ggml_nbytes(t)) return -1;
    ggml_backend_tensor_get(t, dst, 0, ggml_nbytes(t));
    return 0;
}
size_t ref_shim_kv_self_nbytes(struct whisper_state * st, int which) {
    return ggml_nbytes(which == 0 ? st->kv_self.k : st->kv_self.v);
}
int ref_shim_get_kv_self(struct whisper_state * st, int which, void * dst, size_t nbytes) {
    ggml_tensor * t = which == 0 ? st->kv_self.k : st->kv_self.v;
    if (nbytes < ggml_nbytes(t)) return -1;
    ggml_backend_tensor_get(t, dst, 0, ggml_nbytes(t));
    return 0;
}
int ref_shim_kv_self_size(struct whisper_state * st) { return (int) st->kv_self.size; }

// per-state timing counters (whisper.cpp:868-881); ctx->state is null on the whisper-rs path, so
// whisper_print_timings cannot see them.
void ref_shim_get_timings_us(struct whisper_state * st, int64_t * out /*[12]*/) {
    out[0] = st->t_sample_us; out[1] = st->t_encode_us; out[2] = st->t_decode_us;
    out[3] = st->t_batchd_us; out[4] = st->t_prompt_us; out[5] = st->t_mel_us;
    out[6] = st->n_sample;    out[7] = st->n_encode;    out[8] = st->n_decode;
    out[9] = st->n_batchd;    out[10] = st->n_prompt;   out[11] = st->n_fail_p + st->n_fail_h;
}
void ref_shim_reset_timings(struct whisper_state * st) {
    st->t_sample_us = st->t_encode_us = st->t_decode_us = st->t_batchd_us = st->t_prompt_us = st->t_mel_us = 0;
    st->n_sample = st->n_encode = st->n_decode = st->n_batchd = st->n_prompt = 0;
}

// the F16 GELU table the CPU backend actually uses (vec.h:571-585, ggml-cpu.c:3509-3517): lets the
// tests check our device-side table bit for bit.
void ref_shim_gelu_table_f16(uint16_t * dst /*[65536]*/) {
    extern ggml_fp16_t ggml_table_gelu_f16[1 << 16]; // ggml-cpu/vec.cpp:6
    ggml_cpu_init();
    for (int i = 0; i < 65536; ++i) dst[i] = ggml_table_gelu_f16[i];
}

} // extern "C"

// decoder bookkeeping after whisper_full* (whisper.cpp:816-853): lets the tests compare the decode loop's
// internal decisions (failed / completed / result_len / seek_delta / scores) and the raw token sequence
// even when no segment was emitted.
extern "C" int ref_shim_decoder_info(struct whisper_state * st, int j, double * out /*[8]*/, int32_t * ids, int max_ids) {
    const auto & d = st->decoders[j];
    out[0] = d.failed; out[1] = d.completed; out[2] = d.has_ts; out[3] = d.seek_delta; out[4] = d.sequence.result_len;
    out[5] = d.sequence.avg_logprobs; out[6] = d.sequence.entropy; out[7] = st->no_speech_prob;
    const int n = (int) d.sequence.tokens.size();
    for (int i = 0; i < n && i < max_ids; ++i) ids[i] = d.sequence.tokens[i].id;
    return n;
}

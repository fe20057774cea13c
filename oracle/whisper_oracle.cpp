// oracle/whisper_oracle.cpp - TEST INFRASTRUCTURE ONLY.  Never linked, loaded or called by the product
// (whisper-rust_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// A plain scalar C++ restatement of the NUMERICS of the reference's Whisper hot path as executed by its
// ggml-cpu backend on an AVX2/FMA/F16C host (the build recipe in oracle/Makefile), written so that
// every floating-point operation happens in the same order and with the same rounding as there:
//
//   log-mel            whisper.cpp:3015-3276   (wo_mel)
//   conv stem          whisper.cpp:1994-2054, ggml.c:3922-3950, ops.cpp:5866-5945  (im2col F16 + mul_mat)
//   encoder blocks     whisper.cpp:2056-2287   (wo_encode)
//   cross K/V          whisper.cpp:2290-2364
//   decoder + logits   whisper.cpp:2474-2852, 2864-2994 (wo_decode; one sequence, contiguous KV cells)
//   mul_mat            ggml-cpu.c:1269-1470 -> ggml_vec_dot_f16 vec.cpp:191-231 (4x8 F32 partial sums, fixed tree,
//                      F64 leftovers), activations rounded to F16 first (ggml-cpu.c:1331-1366)
//   norm               ops.cpp:3199-3248 (F64 sums), soft_max ops.cpp:4731-4827 + vec.cpp:257-308 (8-lane
//                      ggml_v_expf vec.h:774-811, F64 running sum, libm expf for the n%8 tail)
//   gelu               vec.h:552-585 through the F16 table of ggml-cpu.c:3509-3517
//
// PINNED: tests/test_oracle.py requires BIT-EXACT equality of mel, encoder output and logits with the
// reference engine itself (oracle/_ref/libwhisper_ref.so, live when present) and with the golden vectors
// that engine generated (tests/golden/, made by tools/gen_golden.py).
//
// Build: g++ -O2 -ffp-contract=off -mavx2 -mfma -mf16c -fopenmp (fmaf -> one hardware FMA, nothing else fused).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <immintrin.h>

typedef uint16_t f16;
static inline float h2f(f16 h) { return _cvtsh_ss(h); }
static inline f16   f2h(float f) { return _cvtss_sh(f, 0); }   // round to nearest even, like GGML_FP32_TO_FP16 with F16C

// ---------------------------------------------------------------------------------------------------
// ggml_vec_dot_f16, AVX2 flavour (vec.cpp:191-231, simd-mappings.h:367-385, 400-443)
// ---------------------------------------------------------------------------------------------------
static float dot_f16(int n, const f16 * x, const f16 * y) {
    const int np = n & ~31;
    float sum[4][8];
    for (int j = 0; j < 4; ++j) for (int l = 0; l < 8; ++l) sum[j][l] = 0.0f;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; ++j)
            for (int l = 0; l < 8; ++l) sum[j][l] = fmaf(h2f(x[i + j * 8 + l]), h2f(y[i + j * 8 + l]), sum[j][l]);
    for (int l = 0; l < 8; ++l) { sum[0][l] = sum[0][l] + sum[2][l]; sum[1][l] = sum[1][l] + sum[3][l]; }
    for (int l = 0; l < 8; ++l) sum[0][l] = sum[0][l] + sum[1][l];
    float t0[4];
    for (int l = 0; l < 4; ++l) t0[l] = sum[0][l] + sum[0][l + 4];
    const float t1a = t0[0] + t0[1], t1b = t0[2] + t0[3];
    double sumf = (double) (t1a + t1b);
    for (int i = np; i < n; ++i) sumf += (double) (h2f(x[i]) * h2f(y[i]));
    return (float) sumf;
}

// ggml_v_expf, AVX2+FMA flavour (vec.h:774-811), one lane
static float v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    uint32_t zb; memcpy(&zb, &z, 4);
    const uint32_t e = zb << 23;
    uint32_t one; { const float o = 1.0f; memcpy(&one, &o, 4); }
    float k; { const uint32_t kb = e + one; memcpy(&k, &kb, 4); }
    const float an = fabsf(n);
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, 0x1.ffffecp-1f * b);
    if (!(an > 126.0f)) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    float s1, s2;
    { const uint32_t v = g + 0x7f000000u; memcpy(&s1, &v, 4); }
    { const uint32_t v = e - g; memcpy(&s2, &v, 4); }
    if (an > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}

// soft_max over one row: y = softmax(x*scale + mask) (ops.cpp:4792-4818, vec.cpp:257-308)
static void softmax_row(int n, const float * x, float scale, const float * mask, float * y, std::vector<float> & wp) {
    wp.resize(n);
    for (int i = 0; i < n; ++i) wp[i] = x[i] * scale;
    if (mask) for (int i = 0; i < n; ++i) wp[i] += 1.0f * mask[i];
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = wp[i] > mx ? wp[i] : mx;
    double sum = 0.0;
    int i = 0;
    for (; i + 7 < n; i += 8) {
        float v[8];
        for (int l = 0; l < 8; ++l) { v[l] = v_expf(wp[i + l] - mx); y[i + l] = v[l]; }
        float a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
        float b0 = a0 + a2, b1 = a1 + a3;
        sum += (double) (b0 + b1);
    }
    for (; i < n; ++i) { const float v = expf(wp[i] - mx); sum += (double) v; y[i] = v; }
    const float inv = (float) (1.0 / sum);
    for (int k = 0; k < n; ++k) y[k] = y[k] * inv;
}

// ggml_norm + mul + add (ops.cpp:3225-3242, whisper.cpp:2121-2126)
static void layernorm_row(int d, const float * x, const float * w, const float * b, float eps, float * y) {
    double sum = 0.0;
    for (int i = 0; i < d; ++i) sum += (double) x[i];
    const float mean = sum / d;
    double sum2 = 0.0;
    for (int i = 0; i < d; ++i) { const float v = x[i] - mean; y[i] = v; sum2 += (double) (v * v); }
    const float variance = sum2 / d;
    const float scale = 1.0f / sqrtf(variance + eps);
    for (int i = 0; i < d; ++i) { float t = y[i] * scale; t = t * w[i]; y[i] = t + b[i]; }
}

// ---------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------
struct tensor { int type = 0; std::vector<int> ne; std::vector<uint8_t> data;
                const float * f() const { return (const float *) data.data(); }
                const f16 * h() const { return (const f16 *) data.data(); } };

struct wo_model {
    int n_vocab, n_audio_ctx, d, n_head, n_enc, n_text_ctx, n_dec, n_mels, ftype;
    int n_mel_f, n_fft_f;
    std::vector<float> filters;
    std::map<std::string, tensor> t;
    f16 gelu[65536];
    float hann[400], sinv[400], cosv[400];
    // state
    std::vector<float> mel; int mel_n_len = 0, mel_n_len_org = 0;
    std::vector<float> embd_conv, embd_enc;               // [T][d]
    std::vector<f16> cross_k, cross_v;                    // [layer][T][d] , [layer][d][T]
    std::vector<f16> self_k, self_v;                      // [layer][n_ctx][d], [layer][d][n_ctx]
    int kv_n_ctx = 0;
    std::vector<float> logits;                            // last row of the last decode
    const tensor & get(const std::string & n) const { return t.at(n); }
};

static float gelu_f32(float x) { return 0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x))); }
static float gelu(const wo_model & m, float x) {          // ggml_vec_gelu_f32 with GGML_GELU_FP16 (vec.h:571-585)
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    return h2f(m.gelu[f2h(x)]);
}

// dst[r1][r0] = dot(W[r0], f16(X[r1]))   (ggml_mul_mat(W, X)); W f16 [n_out][k], X f32 [rows][k]
static void mul_mat(const f16 * W, int n_out, int k, const float * X, int rows, float * dst) {
    std::vector<f16> x16((size_t) rows * k);
    for (size_t i = 0; i < x16.size(); ++i) x16[i] = f2h(X[i]);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int o = 0; o < n_out; ++o) dst[(size_t) r * n_out + o] = dot_f16(k, W + (size_t) o * k, x16.data() + (size_t) r * k);
}

extern "C" {

void * wo_load(const char * path) {
    FILE * f = fopen(path, "rb");
    if (!f) return nullptr;
    wo_model * m = new wo_model;
    auto rd32 = [&]() { int32_t v = 0; if (fread(&v, 4, 1, f) != 1) v = 0; return v; };
    if ((uint32_t) rd32() != 0x67676d6c) { fclose(f); delete m; return nullptr; }
    m->n_vocab = rd32(); m->n_audio_ctx = rd32(); m->d = rd32(); m->n_head = rd32(); m->n_enc = rd32();
    m->n_text_ctx = rd32(); rd32(); rd32(); m->n_dec = rd32(); m->n_mels = rd32(); m->ftype = rd32();
    m->n_mel_f = rd32(); m->n_fft_f = rd32();
    m->filters.resize((size_t) m->n_mel_f * m->n_fft_f);
    if (fread(m->filters.data(), 4, m->filters.size(), f) != m->filters.size()) { fclose(f); delete m; return nullptr; }
    const int nv = rd32();
    for (int i = 0; i < nv; ++i) { const uint32_t len = (uint32_t) rd32(); if (len) fseek(f, len, SEEK_CUR); }
    while (true) {
        int32_t hdr[3];
        if (fread(hdr, 4, 3, f) != 3) break;
        tensor t; t.type = hdr[2];
        size_t n = 1;
        for (int i = 0; i < hdr[0]; ++i) { const int v = rd32(); t.ne.push_back(v); n *= v; }
        std::string name(hdr[1], '\0');
        if (fread(&name[0], 1, hdr[1], f) != (size_t) hdr[1]) break;
        t.data.resize(n * (t.type == 0 ? 4 : 2));
        if (fread(t.data.data(), 1, t.data.size(), f) != t.data.size()) break;
        m->t[name] = std::move(t);
    }
    fclose(f);
    for (int i = 0; i < 65536; ++i) m->gelu[i] = f2h(gelu_f32(h2f((f16) i)));                 // ggml-cpu.c:3509-3517
    for (int i = 0; i < 400; ++i) {                                                          // whisper.cpp:3031-3047
        const double theta = (2 * M_PI * i) / 400;
        m->sinv[i] = sinf(theta); m->cosv[i] = cosf(theta);
        m->hann[i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / 400));
    }
    m->kv_n_ctx = ((m->n_text_ctx + 255) / 256) * 256;
    m->self_k.assign((size_t) m->n_dec * m->kv_n_ctx * m->d, 0);
    m->self_v.assign((size_t) m->n_dec * m->kv_n_ctx * m->d, 0);
    return m;
}
void wo_free(void * p) { delete (wo_model *) p; }
void wo_gelu_table(void * p, uint16_t * dst) { memcpy(dst, ((wo_model *) p)->gelu, sizeof(f16) * 65536); }

// ---------------------------------------------------------------------------------------------------
// log-mel (whisper.cpp:3054-3276).  FMA placement = what gcc -O3 -mfma emits for the reference source
// (oracle/Makefile recipe), read off its disassembly: two chained FMAs per butterfly output,
// fma(re,re,im*im) for the power, ((p1 f1 + p0 f0) + p2 f2) + p3 f3 for the 4-term filter partial sums.
// ---------------------------------------------------------------------------------------------------
static void dft(const wo_model & m, const float * in, int N, float * out) {
    const int step = 400 / N;
    for (int k = 0; k < N; ++k) {
        float re = 0, im = 0;
        for (int n = 0; n < N; ++n) {
            const int idx = (k * n * step) % 400;
            re = fmaf(in[n], m.cosv[idx], re);
            im = fmaf(-in[n], m.sinv[idx], im);
        }
        out[k * 2 + 0] = re; out[k * 2 + 1] = im;
    }
}
static void fft(const wo_model & m, float * in, int N, float * out) {
    if (N == 1) { out[0] = in[0]; out[1] = 0; return; }
    const int half = N / 2;
    if (N - half * 2 == 1) { dft(m, in, N, out); return; }
    float * even = in + N;
    for (int i = 0; i < half; ++i) even[i] = in[2 * i];
    float * even_fft = out + 2 * N;
    fft(m, even, half, even_fft);
    float * odd = even;
    for (int i = 0; i < half; ++i) odd[i] = in[2 * i + 1];
    float * odd_fft = even_fft + N;
    fft(m, odd, half, odd_fft);
    const int step = 400 / N;
    for (int k = 0; k < half; ++k) {
        const int idx = k * step;
        const float re = m.cosv[idx], im = -m.sinv[idx];
        const float ro = odd_fft[2 * k], io = odd_fft[2 * k + 1];
        out[2 * k + 0]          = fmaf(-im, io, fmaf(re, ro, even_fft[2 * k + 0]));
        out[2 * k + 1]          = fmaf(im, ro, fmaf(re, io, even_fft[2 * k + 1]));
        out[2 * (k + half) + 0] = fmaf(im, io, fmaf(-re, ro, even_fft[2 * k + 0]));
        out[2 * (k + half) + 1] = fmaf(-im, ro, fmaf(-re, io, even_fft[2 * k + 1]));
    }
}

int wo_mel(void * p, const float * samples, int n_samples) {
    wo_model & m = *(wo_model *) p;
    const int pad1 = 16000 * 30, pad2 = 200;
    std::vector<float> padded((size_t) n_samples + pad1 + 2 * pad2, 0.0f);
    memcpy(padded.data() + pad2, samples, (size_t) n_samples * 4);
    for (int i = 0; i < pad2; ++i) padded[i] = samples[pad2 - i];           // reverse_copy(samples+1, samples+1+200)
    const int n_len = (int) ((padded.size() - 400) / 160);
    m.mel_n_len = n_len;
    m.mel_n_len_org = 1 + (n_samples + pad2 - 400) / 160;
    const int n_mel = m.n_mel_f, n_fft = m.n_fft_f;
    m.mel.assign((size_t) n_mel * n_len, 0.0f);
    const int n_w = n_samples + pad2;
    const int n_active = std::min(n_w / 160 + 1, n_len);
#pragma omp parallel
    {
        std::vector<float> fft_in(800, 0.0f), fft_out(3200);
#pragma omp for schedule(static)
        for (int i = 0; i < n_len; ++i) {
            if (i >= n_active) { for (int j = 0; j < n_mel; ++j) m.mel[(size_t) j * n_len + i] = log10(1e-10); continue; }
            const int off = i * 160;
            const int lim = std::min(400, n_w - off);
            for (int j = 0; j < lim; ++j) fft_in[j] = m.hann[j] * padded[off + j];
            for (int j = std::max(lim, 0); j < 800; ++j) fft_in[j] = 0.0f;
            fft(m, fft_in.data(), 400, fft_out.data());
            for (int j = 0; j < n_fft; ++j) fft_out[j] = fmaf(fft_out[2 * j], fft_out[2 * j], fft_out[2 * j + 1] * fft_out[2 * j + 1]);
            for (int j = 0; j < n_mel; ++j) {
                const float * flt = m.filters.data() + (size_t) j * n_fft;
                double sum = 0.0;
                int k = 0;
                for (; k < n_fft - 3; k += 4) {
                    float s4 = fft_out[k + 1] * flt[k + 1];
                    s4 = fmaf(fft_out[k], flt[k], s4);
                    s4 = fmaf(fft_out[k + 2], flt[k + 2], s4);
                    s4 = fmaf(fft_out[k + 3], flt[k + 3], s4);
                    sum += (double) s4;
                }
                for (; k < n_fft; ++k) sum += (double) (fft_out[k] * flt[k]);
                sum = log10(sum > 1e-10 ? sum : 1e-10);
                m.mel[(size_t) j * n_len + i] = sum;
            }
        }
    }
    double mmax = -1e20;
    for (float v : m.mel) if (v > mmax) mmax = v;
    mmax -= 8.0;
    for (float & v : m.mel) { if (v < mmax) v = mmax; v = (v + 4.0) / 4.0; }
    return 0;
}
int wo_mel_n_len(void * p) { return ((wo_model *) p)->mel_n_len; }
int wo_mel_n_len_org(void * p) { return ((wo_model *) p)->mel_n_len_org; }
const float * wo_mel_data(void * p) { return ((wo_model *) p)->mel.data(); }
void wo_set_mel(void * p, const float * data, int n_len) {
    wo_model & m = *(wo_model *) p;
    m.mel.assign(data, data + (size_t) m.n_mel_f * n_len); m.mel_n_len = n_len; m.mel_n_len_org = n_len;
}

// ---------------------------------------------------------------------------------------------------
// encoder (whisper.cpp:1994-2364)
// ---------------------------------------------------------------------------------------------------
// ggml_conv_1d: im2col to F16 [OL][IC*3] (ops.cpp:5925-5937), then dot with kernel rows [OC][IC*3] (ggml.c:3929-3937)
static void conv1d(const f16 * W, int OC, int IC, const float * in /*[IC][IL]*/, int IL, int stride, int OL, float * out /*[OC][OL]*/) {
    std::vector<f16> col((size_t) OL * IC * 3);
    for (int ol = 0; ol < OL; ++ol)
        for (int ic = 0; ic < IC; ++ic)
            for (int k = 0; k < 3; ++k) {
                const int ii = ol * stride + k - 1;
                col[((size_t) ol * IC + ic) * 3 + k] = (ii < 0 || ii >= IL) ? (f16) 0 : f2h(in[(size_t) ic * IL + ii]);
            }
#pragma omp parallel for schedule(static)
    for (int oc = 0; oc < OC; ++oc)
        for (int ol = 0; ol < OL; ++ol) out[(size_t) oc * OL + ol] = dot_f16(IC * 3, col.data() + (size_t) ol * IC * 3, W + (size_t) oc * IC * 3);
}

int wo_encode(void * p, int mel_offset) {
    wo_model & m = *(wo_model *) p;
    const int T = m.n_audio_ctx, d = m.d, H = m.n_head, dh = d / H, n_mels = m.n_mels;
    // mel window (whisper.cpp:2399-2418)
    std::vector<float> win((size_t) n_mels * 2 * T, 0.0f);
    const int i0 = std::min(mel_offset, m.mel_n_len), i1 = std::min(mel_offset + 2 * T, m.mel_n_len);
    for (int j = 0; j < n_mels; ++j) for (int i = i0; i < i1; ++i) win[(size_t) j * 2 * T + (i - i0)] = m.mel[(size_t) j * m.mel_n_len + i];

    std::vector<float> c1((size_t) d * 2 * T), c2((size_t) d * T);
    conv1d(m.get("encoder.conv1.weight").h(), d, n_mels, win.data(), 2 * T, 1, 2 * T, c1.data());
    { const float * b = m.get("encoder.conv1.bias").f(); for (int oc = 0; oc < d; ++oc) for (int t = 0; t < 2 * T; ++t) { float & v = c1[(size_t) oc * 2 * T + t]; v = gelu(m, v + b[oc]); } }
    conv1d(m.get("encoder.conv2.weight").h(), d, d, c1.data(), 2 * T, 2, T, c2.data());
    { const float * b = m.get("encoder.conv2.bias").f(); for (int oc = 0; oc < d; ++oc) for (int t = 0; t < T; ++t) { float & v = c2[(size_t) oc * T + t]; v = gelu(m, v + b[oc]); } }
    m.embd_conv.resize((size_t) T * d);
    for (int t = 0; t < T; ++t) for (int c = 0; c < d; ++c) m.embd_conv[(size_t) t * d + c] = c2[(size_t) c * T + t];

    std::vector<float> x((size_t) T * d), cur((size_t) T * d), q((size_t) T * d), k((size_t) T * d), v((size_t) T * d), att((size_t) T * d), ff((size_t) T * 4 * d);
    { const float * pe = m.get("encoder.positional_embedding").f(); for (size_t i = 0; i < x.size(); ++i) x[i] = pe[i] + m.embd_conv[i]; }
    const float KQscale = 1.0f / sqrtf(float(dh));
    for (int il = 0; il < m.n_enc; ++il) {
        const std::string pf = "encoder.blocks." + std::to_string(il) + ".";
        const float * lw = m.get(pf + "attn_ln.weight").f(), * lb = m.get(pf + "attn_ln.bias").f();
#pragma omp parallel for
        for (int t = 0; t < T; ++t) layernorm_row(d, &x[(size_t) t * d], lw, lb, 1e-5f, &cur[(size_t) t * d]);
        mul_mat(m.get(pf + "attn.query.weight").h(), d, d, cur.data(), T, q.data());
        mul_mat(m.get(pf + "attn.key.weight").h(),   d, d, cur.data(), T, k.data());
        mul_mat(m.get(pf + "attn.value.weight").h(), d, d, cur.data(), T, v.data());
        { const float * qb = m.get(pf + "attn.query.bias").f(), * vb = m.get(pf + "attn.value.bias").f();
          for (int t = 0; t < T; ++t) for (int c = 0; c < d; ++c) { q[(size_t) t * d + c] += qb[c]; v[(size_t) t * d + c] += vb[c]; } }
        // K, V cast to F16 (whisper.cpp:2181-2200); Q rounded to F16 as mul_mat's src1; P rounded to F16 likewise
        std::vector<f16> q16((size_t) T * d), k16((size_t) T * d), vt16((size_t) d * T);
        for (size_t i = 0; i < q16.size(); ++i) { q16[i] = f2h(q[i]); k16[i] = f2h(k[i]); }
        for (int t = 0; t < T; ++t) for (int c = 0; c < d; ++c) vt16[(size_t) c * T + t] = f2h(v[(size_t) t * d + c]);
#pragma omp parallel
        {
            std::vector<float> s(T), pr(T), wp;
            std::vector<f16> p16(T);
#pragma omp for collapse(2) schedule(static)
            for (int h = 0; h < H; ++h)
                for (int t = 0; t < T; ++t) {
                    for (int u = 0; u < T; ++u) s[u] = dot_f16(dh, &k16[(size_t) u * d + h * dh], &q16[(size_t) t * d + h * dh]);
                    softmax_row(T, s.data(), KQscale, nullptr, pr.data(), wp);
                    for (int u = 0; u < T; ++u) p16[u] = f2h(pr[u]);
                    for (int c = 0; c < dh; ++c) att[(size_t) t * d + h * dh + c] = dot_f16(T, &vt16[(size_t) (h * dh + c) * T], p16.data());
                }
        }
        mul_mat(m.get(pf + "attn.out.weight").h(), d, d, att.data(), T, cur.data());
        { const float * b = m.get(pf + "attn.out.bias").f(); for (int t = 0; t < T; ++t) for (int c = 0; c < d; ++c) { size_t i = (size_t) t * d + c; x[i] = (cur[i] + b[c]) + x[i]; } }
        lw = m.get(pf + "mlp_ln.weight").f(); lb = m.get(pf + "mlp_ln.bias").f();
#pragma omp parallel for
        for (int t = 0; t < T; ++t) layernorm_row(d, &x[(size_t) t * d], lw, lb, 1e-5f, &cur[(size_t) t * d]);
        mul_mat(m.get(pf + "mlp.0.weight").h(), 4 * d, d, cur.data(), T, ff.data());
        { const float * b = m.get(pf + "mlp.0.bias").f(); for (int t = 0; t < T; ++t) for (int c = 0; c < 4 * d; ++c) { float & z = ff[(size_t) t * 4 * d + c]; z = gelu(m, z + b[c]); } }
        mul_mat(m.get(pf + "mlp.2.weight").h(), d, 4 * d, ff.data(), T, cur.data());
        { const float * b = m.get(pf + "mlp.2.bias").f(); for (int t = 0; t < T; ++t) for (int c = 0; c < d; ++c) { size_t i = (size_t) t * d + c; x[i] = (cur[i] + b[c]) + x[i]; } }
    }
    m.embd_enc.resize((size_t) T * d);
    { const float * lw = m.get("encoder.ln_post.weight").f(), * lb = m.get("encoder.ln_post.bias").f();
#pragma omp parallel for
      for (int t = 0; t < T; ++t) layernorm_row(d, &x[(size_t) t * d], lw, lb, 1e-5f, &m.embd_enc[(size_t) t * d]); }

    // cross K/V (whisper.cpp:2316-2356): K scaled by d_h^-1/4 then F16; V + bias then F16, stored transposed
    const float Kscale = pow(float(dh), -0.25);
    m.cross_k.assign((size_t) m.n_dec * T * d, 0); m.cross_v.assign((size_t) m.n_dec * T * d, 0);
    for (int il = 0; il < m.n_dec; ++il) {
        const std::string pf = "decoder.blocks." + std::to_string(il) + ".cross_attn.";
        mul_mat(m.get(pf + "key.weight").h(), d, d, m.embd_enc.data(), T, k.data());
        mul_mat(m.get(pf + "value.weight").h(), d, d, m.embd_enc.data(), T, v.data());
        const float * vb = m.get(pf + "value.bias").f();
        for (int t = 0; t < T; ++t) for (int c = 0; c < d; ++c) {
            m.cross_k[((size_t) il * T + t) * d + c] = f2h(k[(size_t) t * d + c] * Kscale);
            m.cross_v[((size_t) il * d + c) * T + t] = f2h(v[(size_t) t * d + c] + vb[c]);
        }
    }
    return 0;
}
const float * wo_embd_conv(void * p) { return ((wo_model *) p)->embd_conv.data(); }
const float * wo_embd_enc(void * p) { return ((wo_model *) p)->embd_enc.data(); }

// ---------------------------------------------------------------------------------------------------
// decoder (whisper.cpp:2474-2852): one sequence, cells == positions (what whisper_decode_with_state does)
// ---------------------------------------------------------------------------------------------------
int wo_decode(void * p, const int32_t * tokens, int n_tokens, int n_past) {
    wo_model & m = *(wo_model *) p;
    const int T = m.n_audio_ctx, d = m.d, H = m.n_head, dh = d / H, n_ctx = m.kv_n_ctx, N = n_tokens;
    const int n_kv = n_past + n_tokens;
    const float KQscale = pow(float(dh), -0.25);
    std::vector<float> x((size_t) N * d), cur((size_t) N * d), q((size_t) N * d), k((size_t) N * d), v((size_t) N * d), att((size_t) N * d), ff((size_t) N * 4 * d);
    { const f16 * te = m.get("decoder.token_embedding.weight").h(); const float * pe = m.get("decoder.positional_embedding").f();
      for (int j = 0; j < N; ++j) for (int c = 0; c < d; ++c) x[(size_t) j * d + c] = h2f(te[(size_t) tokens[j] * d + c]) + pe[(size_t) (n_past + j) * d + c]; }
    for (int il = 0; il < m.n_dec; ++il) {
        const std::string pf = "decoder.blocks." + std::to_string(il) + ".";
        auto LN = [&](const std::string & name, const std::vector<float> & in, std::vector<float> & out) {
            const float * lw = m.get(pf + name + ".weight").f(), * lb = m.get(pf + name + ".bias").f();
            for (int j = 0; j < N; ++j) layernorm_row(d, &in[(size_t) j * d], lw, lb, 1e-5f, &out[(size_t) j * d]);
        };
        LN("attn_ln", x, cur);
        mul_mat(m.get(pf + "attn.query.weight").h(), d, d, cur.data(), N, q.data());
        mul_mat(m.get(pf + "attn.key.weight").h(),   d, d, cur.data(), N, k.data());
        mul_mat(m.get(pf + "attn.value.weight").h(), d, d, cur.data(), N, v.data());
        { const float * qb = m.get(pf + "attn.query.bias").f(), * vb = m.get(pf + "attn.value.bias").f();
          for (int j = 0; j < N; ++j) for (int c = 0; c < d; ++c) {
              size_t i = (size_t) j * d + c;
              q[i] = (q[i] + qb[c]) * KQscale; k[i] = k[i] * KQscale; v[i] = v[i] + vb[c];
              m.self_k[((size_t) il * n_ctx + n_past + j) * d + c] = f2h(k[i]);
              m.self_v[((size_t) il * d + c) * n_ctx + n_past + j] = f2h(v[i]);
          } }
        std::vector<f16> q16((size_t) N * d);
        for (size_t i = 0; i < q16.size(); ++i) q16[i] = f2h(q[i]);
        {
            std::vector<float> s(n_kv), pr(n_kv), mask(n_kv), wp;
            std::vector<f16> p16(n_kv);
            for (int h = 0; h < H; ++h) for (int j = 0; j < N; ++j) {
                for (int u = 0; u < n_kv; ++u) { s[u] = dot_f16(dh, &m.self_k[((size_t) il * n_ctx + u) * d + h * dh], &q16[(size_t) j * d + h * dh]); mask[u] = u > n_past + j ? -INFINITY : 0.0f; }
                softmax_row(n_kv, s.data(), 1.0f, mask.data(), pr.data(), wp);
                for (int u = 0; u < n_kv; ++u) p16[u] = f2h(pr[u]);
                for (int c = 0; c < dh; ++c) att[(size_t) j * d + h * dh + c] = dot_f16(n_kv, &m.self_v[((size_t) il * d + h * dh + c) * n_ctx], p16.data());
            }
        }
        mul_mat(m.get(pf + "attn.out.weight").h(), d, d, att.data(), N, cur.data());
        { const float * b = m.get(pf + "attn.out.bias").f(); for (int j = 0; j < N; ++j) for (int c = 0; c < d; ++c) { size_t i = (size_t) j * d + c; x[i] = (cur[i] + b[c]) + x[i]; } }
        LN("cross_attn_ln", x, cur);
        mul_mat(m.get(pf + "cross_attn.query.weight").h(), d, d, cur.data(), N, q.data());
        { const float * qb = m.get(pf + "cross_attn.query.bias").f(); for (int j = 0; j < N; ++j) for (int c = 0; c < d; ++c) q[(size_t) j * d + c] += qb[c]; }
        for (size_t i = 0; i < q16.size(); ++i) q16[i] = f2h(q[i]);
#pragma omp parallel
        {
            std::vector<float> s(T), pr(T), wp;
            std::vector<f16> p16(T);
#pragma omp for collapse(2) schedule(static)
            for (int h = 0; h < H; ++h) for (int j = 0; j < N; ++j) {
                for (int u = 0; u < T; ++u) s[u] = dot_f16(dh, &m.cross_k[((size_t) il * T + u) * d + h * dh], &q16[(size_t) j * d + h * dh]);
                softmax_row(T, s.data(), KQscale, nullptr, pr.data(), wp);
                for (int u = 0; u < T; ++u) p16[u] = f2h(pr[u]);
                for (int c = 0; c < dh; ++c) att[(size_t) j * d + h * dh + c] = dot_f16(T, &m.cross_v[((size_t) il * d + h * dh + c) * T], p16.data());
            }
        }
        mul_mat(m.get(pf + "cross_attn.out.weight").h(), d, d, att.data(), N, cur.data());
        { const float * b = m.get(pf + "cross_attn.out.bias").f(); for (int j = 0; j < N; ++j) for (int c = 0; c < d; ++c) { size_t i = (size_t) j * d + c; x[i] = (cur[i] + b[c]) + x[i]; } }
        LN("mlp_ln", x, cur);
        mul_mat(m.get(pf + "mlp.0.weight").h(), 4 * d, d, cur.data(), N, ff.data());
        { const float * b = m.get(pf + "mlp.0.bias").f(); for (int j = 0; j < N; ++j) for (int c = 0; c < 4 * d; ++c) { float & z = ff[(size_t) j * 4 * d + c]; z = gelu(m, z + b[c]); } }
        mul_mat(m.get(pf + "mlp.2.weight").h(), d, 4 * d, ff.data(), N, cur.data());
        { const float * b = m.get(pf + "mlp.2.bias").f(); for (int j = 0; j < N; ++j) for (int c = 0; c < d; ++c) { size_t i = (size_t) j * d + c; x[i] = (cur[i] + b[c]) + x[i]; } }
    }
    { const float * lw = m.get("decoder.ln.weight").f(), * lb = m.get("decoder.ln.bias").f();
      for (int j = 0; j < N; ++j) layernorm_row(d, &x[(size_t) j * d], lw, lb, 1e-5f, &cur[(size_t) j * d]); }
    m.logits.resize(m.n_vocab);
    mul_mat(m.get("decoder.token_embedding.weight").h(), m.n_vocab, d, &cur[(size_t) (N - 1) * d], 1, m.logits.data());
    return 0;
}
const float * wo_logits(void * p) { return ((wo_model *) p)->logits.data(); }
int wo_n_vocab(void * p) { return ((wo_model *) p)->n_vocab; }
int wo_n_state(void * p) { return ((wo_model *) p)->d; }
int wo_n_audio_ctx(void * p) { return ((wo_model *) p)->n_audio_ctx; }

} // extern "C"

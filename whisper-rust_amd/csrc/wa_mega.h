// wa_mega.h - the single-token decode step as ONE persistent launch (wa_mega.hip).
//
// ref: whisper_build_graph_decoder whisper.cpp:2474-2852 for n_tokens == 1 (the greedy steady state).
// The launch sequence of wa_decode.cpp (122 dependent launches, each a few microseconds of fill / drain around
// < 1 us of streaming) becomes one grid of one workgroup per CU whose workgroups hand activations to each other
// through 8-byte {tag, value} granules (write-through stores, polled loads) and prefetch everything that does
// not depend on the token (weights, self / cross K and V) ahead of the hand-off that needs it.
#pragma once
#include "wa_kernels.h"

struct wa_mega_layer {          // device-resident table, one entry per decoder layer (pointers into the model arena)
    const float * ln1_w, * ln1_b; const wa_f16 * qkv_w; const float * qkv_b, * qkv_s;    // [3d][d], bias (0 for k), column scale
    const wa_f16 * out_w;  const float * out_b;
    const float * ln2_w, * ln2_b; const wa_f16 * cq_w;  const float * cq_b;
    const wa_f16 * co_w;   const float * co_b;
    const float * ln3_w, * ln3_b; const wa_f16 * fc1_w; const float * fc1_b;
    const wa_f16 * fc2_w;  const float * fc2_b;
    // quantised model (Q5_0 / Q8_0 files): the *_w pointers then hold the signed-byte quants in the kernel layout of wa_quant.hip
    // ([row][lane 0..7][block][4]) and these the F32 block scales [row][block]; null for an F16 model
    const float * qkv_d, * out_d, * cq_d, * co_d, * fc1_d, * fc2_d;
};

struct wa_mega_args {
    // model
    const wa_mega_layer * layers; int n_layer, d, n_head, n_vocab; float eps;
    double rn_d;                                                                // 1.0 / (double) d (LayerNorm: keeps an F64 division off every phase)
    const wa_f16 * te; const float * pe; const float * lnf_w, * lnf_b; const wa_f16 * gelu;
    const float * te_d; int quant;                                              // quantised model: te = quants, te_d = block scales of the token embedding
    // state
    wa_f16 * kv_k, * kv_v; unsigned long long kv_layer_stride;                  // self K/V [layer][cell][d]
    const wa_f16 * cross_k, * cross_v; unsigned long long cross_layer_stride;   // cross K/V [layer][head][tpad][64]
    int cross_tpad, T;
    unsigned long long * granules; int edge_stride;                            // [layer][8][edge_stride] hand-off granules
    unsigned long long * cross_gr;                                              // [layer][head][2048] exchange area of a head's four cross-attention workgroups
    float * logits;                                                             // [n_vocab], then: status word, the token decoded (spec), the launch's sequence number once it has run
    unsigned * status;                                                          // 0 = ok; else code of the hand-off that timed out
    float * dbg;                                                                // optional [layer][head][2][1536]: cross-attention scores, probabilities
    // launch
    int token, pos, n_kv, kv_head; unsigned seq; float kq_scale;
    // next-token prediction (host overlap, wa_decode.cpp: wa_spec_*).  Every launch leaves per-workgroup candidate records
    // of its logits under the reference's logit rules (whisper.cpp:6149-6333, approximated: the host verifies every token);
    // with spec != 0 the launch takes its input token from the records and state of the previous launch instead of `token`.
    int spec;
    const unsigned * rec_in; unsigned * rec_out; int n_rec;          // [n_workgroups][8]: {max text logit, id, max timestamp logit, id, sum exp(ts - max ts)}
    const int * ps_in; int * ps_out;                                 // {last token, token before it (-1: none), seek_delta, has_ts}
    const unsigned * smask;                                          // bit i set: token i is suppressed for the whole call
    int token_beg, token_eot;
    int s_last, s_penult, s_seek_delta, s_has_ts;                    // spec == 0: the state after `token`, from the host
};

#define WA_MEGA_EDGES 8
#define WA_MEGA_CGR 2048
#define WA_MEGA_REDO 9000u      // status: a soft-max sum could not be certified order-independent - recompute this token with the launch sequence
#define WA_MEGA_MAX_D 1280
#define WA_MEGA_MAX_KV 512
#define WA_MEGA_KV_ROOM 448          // most cells a one-launch step attends over (n_text_ctx of every released model): the LDS rows behind them serve mg_attn_finish
#define WA_MEGA_MAX_T 1536

#if defined(__HIPCC__)
#define WA_HD __host__ __device__
#else
#define WA_HD
#endif
// Role of workgroup b of n (role 0: weight streaming, index = its rank; 1: self-attention of head index; 2: cross-attention, index =
// 4 head + quarter).  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one - observed, not promised: the
// cross-attention role checks it at run time), so with n % 8 == 0 the four quarters of a head are placed 8 apart: their three
// exchanges per layer then stay inside one L2.  Heads sit in groups of 8 at the top of the grid; the slots of a group that
// has fewer than 8 heads go to self-attention first, then to weight streaming.
WA_HD inline void mg_role_of(int n, int H, int b, int & role, int & idx) {
    const int G8 = (H + 7) >> 3, top = n - 32 * G8;
    const int R = H - 8 * (G8 - 1);                     // heads in the last group (1..8)
    const int F = 4 * (8 - R);                          // free slots up there
    const int s_low = H > F ? H - F : 0;                // self-attention heads placed below `top`
    if ((n & 7) != 0 || top - s_low < 1) {              // plain layout
        const int nG = n - 5 * H;
        if (b < nG) { role = 0; idx = b; } else if (b < nG + H) { role = 1; idx = b - nG; } else { role = 2; idx = b - nG - H; }
        return;
    }
    if (b >= top) {
        const int t = b - top, r = t & 7, k = t >> 3, q = k >> 2, w = k & 3, h = r + 8 * q;
        if (h < H) { role = 2; idx = 4 * h + w; return; }
        const int f = (r - R) + (8 - R) * w;            // free slot number
        if (f < H) { role = 1; idx = f; } else { role = 0; idx = (top - s_low) + (f - H); }
        return;
    }
    if (b >= top - s_low) { role = 1; idx = F + (b - (top - s_low)); } else { role = 0; idx = b; }
}

// n_wg workgroups of 512 threads, every one of them resident at once (n_wg <= number of CUs; 1 workgroup per CU)
bool   wa_launch_decode_mega(hipStream_t s, const wa_mega_args & a, int n_wg);      // a.quant selects the quantised-weights form of the kernel; false: the launch failed
size_t wa_mega_lds_bytes();

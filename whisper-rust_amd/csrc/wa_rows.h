// wa_rows.h - the decode step for 2..8 token ROWS as ONE persistent launch (wa_rows.hip).
//
// ref: whisper_build_graph_decoder whisper.cpp:2474-2852 with n_tokens = n_decoders_cur rows: beam search and best_of > 1 batch one
// token per live decoder (whisper.cpp:7404-7431), the reference's bench a 5-token batch (examples/bench/bench.cpp:110-123), and
// lock-step chunks (wa_decode.cpp: wa_batcher) one token per chunk.  wa_mega.hip is the one-row form of the same idea, tuned around
// a single token's dependency chain; with several rows the work per phase is larger and rows differ in where their K / V live, so
// this kernel is organised differently:
//   * every workgroup does everything (no roles): its slice of the rows of every matrix for ALL token rows, and - in the attention
//     phases - the (row, head) / (row, head, quarter) units dealt to it, so the encoder K / V of different chunks stream over all CUs;
//   * a workgroup's weight rows arrive by LDS-DMA, issued by a dedicated wave one phase ahead into one of two LDS slots: the streaming
//     is off the vector-memory queues of the waves that poll hand-offs, and the products read weights AND activations from LDS;
//   * LayerNorm is one wave per token row (no cross-wave reductions);
//   * hand-offs are the 8-byte {tag, value} granules of wa_mega.hip.
// Arithmetic is the reference order throughout (same chains, trees, certified F64 sums as wa_exact.hip): logits are bit-identical to
// the launch sequence, which stays as the fallback.
#pragma once
#include "wa_mega.h"

#define WA_ROWS_MAX 8
#define WA_ROWS_MAXKV 2048           // most self-attention cells a row may attend over in this kernel (scores / probabilities in LDS)
#define WA_ROWS_CGR 2048             // granules per (layer, row, head) of the cross-attention exchange area

struct wa_rows_row {                 // where ONE token row's state lives (layer 0; the kernel adds the layer offsets)
    wa_f16 * kv_k, * kv_v;           // self K / V cells [layer][cell][d]
    const wa_f16 * cross_k, * cross_v;
    const int8_t * mask;             // [n_kv] 1 = cell hidden from this row (beams share cells by sequence id); null: every cell < n_kv visible
    int n_kv, kv_head;               // cells attended over; the cell this row's new key / value go to
    int token, pos;
    // next-token prediction for lock-step members that decode greedily (wa_decode.cpp: the run-ahead batcher; as wa_mega_args): with `smask` set
    // the launch leaves per-workgroup candidate records of the row's logits under the reference's logit rules (approximated: the host verifies
    // every token); with spec != 0 the row's token is picked from the records and state its previous pass left, not taken from `token`.
    int spec;
    const unsigned * rec_in; unsigned * rec_out;      // [n_workgroups][8]: {max text logit, id, max timestamp logit, id, sum exp(ts - max ts)}
    const int * ps_in; int * ps_out;                  // {last token, token before it (-1: none), seek_delta, has_ts}; ps_out[4] = the token this launch decoded
    const unsigned * smask;                           // bit i set: token i is suppressed for the row's whole call; null: no records
    int s_last, s_penult, s_seek_delta, s_has_ts;     // spec == 0: the state after `token`, from the host
};

struct wa_rows_args {
    // model (as wa_mega_args)
    const wa_mega_layer * layers; int n_layer, d, n_head, n_vocab; float eps;
    double rn_d;
    const wa_f16 * te; const float * pe; const float * lnf_w, * lnf_b; const wa_f16 * gelu;
    const float * te_d; int quant;
    unsigned long long kv_layer_stride, cross_layer_stride;
    int cross_tpad, T;
    unsigned long long * granules; int row_gr;          // [layer][8][B][row_gr] hand-off granules (row_gr = 2 d)
    unsigned long long * cross_gr;                      // [layer][B][head][WA_ROWS_CGR]
    float * logits;                                     // [B][n_vocab]
    unsigned * status;                                  // [0] 0 = ok, else the code of the hand-off that timed out; [1] = seq once the launch has run
    unsigned * row_status;                              // [B] (cleared by the caller) WA_MEGA_REDO: a soft-max sum of this row could not be certified order-independent -
                                                        // the row's logits may be an ulp off, the row is to be redone by the launch sequence; the other rows stand
    float * dbg;
    int * tok_out;                                      // [B] the token every row decoded (given, or picked from its records), null: not wanted
    float kq_scale; unsigned seq;
    int B, slot_bytes;
    int force_inorder;                                  // tests (WHISPER_AMD_ROWS_FORCE_INORDER): every cross soft-max total takes the in-order path of mb_unit_cross
    int token_beg, token_eot;                           // (records) first timestamp token, end-of-text token
    int n_out, out_row[WA_ROWS_MAX];                    // token rows whose logits are wanted (the reference flags batch.logits rows): logits row m = token row out_row[m]
    wa_rows_row rows[WA_ROWS_MAX];
};

// LDS the kernel needs for B rows of a d-wide model with `n_wg` workgroups; 0 when it does not fit (the caller keeps the launch sequence).
// slot_bytes = size of one of the two weight slots.
size_t wa_rows_lds_bytes(int d, int B, int n_wg, int quant, int * slot_bytes);
// n_wg workgroups of 512 threads, all resident at once (1 per CU); false: shape not supported
bool wa_launch_decode_rows(hipStream_t s, const wa_rows_args & a, int n_wg);

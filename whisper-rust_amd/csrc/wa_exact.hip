// wa_exact.hip - the REFERENCE-ORDER kernels: every floating-point sum is formed in exactly the order
// the reference's ggml-cpu AVX2 path forms it, so results are bit-identical to whisper.cpp CPU
// (oracle/whisper_oracle.cpp is the scalar statement of the same order, pinned bit-exact to the
// reference engine).  Used for (a) every decode-step kernel - the decode step is HBM-bound, the order costs
// nothing - and (b) the encoder / prompt GEMMs and attention when flash_attn == false.
//
//   dot products   ggml_vec_dot_f16 (vec.cpp:191-231): 32 F32 partial sums s[i mod 32], each an FMA
//                  chain in k order; fixed reduction tree (wa_tree32); K % 32 leftovers added in F64.
//   soft_max       ops.cpp:4792-4818 + vec.cpp:257-308: expf polynomial on groups of 8 with the 8-lane
//                  tree, F64 running sum in group order, libm expf for the n % 8 tail.
//   norm           ops.cpp:3225-3242: F64 sums; evaluated in parallel when an exponent-range certificate
//                  proves every partial sum exact (then any order gives the reference's value), else by
//                  one lane in index order.
// Compiled with -ffp-contract=off: fmaf is the only fused operation.
#include "wa_device.h"
#include <cstdlib>

// =================================================================================================
// GEMM (any M): C = A W^T in ggml_vec_dot_f16 order.  VALU, not MFMA: the MFMA's internal summation
// order is not the reference's.  Block = 4 waves in 2x2; wave = 8x8 lanes, each lane owns a 2x2 output
// patch with all 32 partial sums of each output in registers (128 accumulators), so the final tree is
// lane-local.  LDS rows are 40 halfs: the b128 reads of a 16-lane group hit disjoint banks.
// Roofline: F32 VALU-bound (157 TFLOP/s peak), 128 FMAs per 16 LDS b128 reads.
// =================================================================================================
#define EX_LD 40

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_exact(const wa_f16 * __restrict__ A, int lda, const wa_f16 * __restrict__ W, int ldw,
                                                       int M, int N, int K, wa_epi e) {
    __shared__ __attribute__((aligned(16))) wa_f16 As[2][32 * EX_LD];
    __shared__ __attribute__((aligned(16))) wa_f16 Ws[2][32 * EX_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane >> 3, lj = lane & 7;
    const int tiles_n = (N + 31) / 32;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;

    float acc[2][2][32];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[p][q][i] = 0.0f;

    const int np = K & ~31, nsteps = np >> 5;
    // global -> LDS staging: threads 0..127 carry A chunks, 128..255 carry W chunks (16 B each)
    const int lrow = (tid & 127) >> 2, lkc = tid & 3;
    const bool isA = tid < 128;
    const wa_f16 * gsrc;
    {
        if (isA) { int gm = m0 + lrow; gm = gm < M ? gm : M - 1; gsrc = A + (size_t) gm * lda + lkc * 8; }
        else     { int gn = n0 + lrow; gn = gn < N ? gn : N - 1; gsrc = W + (size_t) gn * ldw + lkc * 8; }
    }
    uint4 stage = make_uint4(0, 0, 0, 0);
    if (nsteps > 0) stage = *(const uint4 *) gsrc;
    if (nsteps > 0) {
        *(uint4 *) (isA ? &As[0][lrow * EX_LD + lkc * 8] : &Ws[0][lrow * EX_LD + lkc * 8]) = stage;
    }
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) stage = *(const uint4 *) (gsrc + (s + 1) * 32);
        half8 a[2][4], w[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[p][c] = *(const half8 *) (&As[buf][(wm * 16 + li * 2 + p) * EX_LD + c * 8]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) w[q][c] = *(const half8 *) (&Ws[buf][(wn * 16 + lj * 2 + q) * EX_LD + c * 8]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[p][q][c * 8 + i] = fmaf((float) w[q][c][i], (float) a[p][c][i], acc[p][q][c * 8 + i]);
        if (s + 1 < nsteps) *(uint4 *) (isA ? &As[buf ^ 1][lrow * EX_LD + lkc * 8] : &Ws[buf ^ 1][lrow * EX_LD + lkc * 8]) = stage;
        __syncthreads();
    }

#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = m0 + wm * 16 + li * 2 + p, n = n0 + wn * 16 + lj * 2 + q;
            if (m < M && n < N) {
                float res = wa_tree32(acc[p][q]);
                if (np < K) {      // leftovers in F64, index order (vec.cpp:221-223)
                    double sumf = (double) res;
                    const wa_f16 * ar = A + (size_t) m * lda, * wr = W + (size_t) n * ldw;
                    for (int i = np; i < K; ++i) sumf += (double) (h2f(wr[i]) * h2f(ar[i]));
                    res = (float) sumf;
                }
                epi_store<EPI>(e, m, n, res);
            }
        }
}

// -------------------------------------------------------------------------------------------------
// The same product (same lanes, same 32 partial sums per output, same order) with the operand tiles brought in by LDS-DMA through a
// ring of 4 stages of 64 k each: the register-staged form above waits out a memory round trip per 32-k step (~1 us against 0.24 us
// of FMAs).  K % 64 == 0.  XOR-swizzled 16-byte slots as in k_gemm_f16_dma (a thread reads whole rows: the 8 distinct rows of a
// wave's li / lj land on 8 distinct slots); every wave issues its own two 1 KB pieces per stage (a fifth, loading wave would not fit
// beside two 240-VGPR workgroups per CU).
// -------------------------------------------------------------------------------------------------
#define EXD_NST 4
template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_exact_dma(const wa_f16 * __restrict__ A, int lda, const wa_f16 * __restrict__ W, int ldw,
                                                           int M, int N, int K, wa_epi e) {
    __shared__ __attribute__((aligned(1024))) wa_f16 S[EXD_NST][2][32 * 64];       // 4 x (4 KB + 4 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane >> 3, lj = lane & 7;
    const int tiles_n = (N + 31) / 32;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;

    // this wave's piece (8 rows x 128 B) of either operand: rows 8 wave + lane / 8, slot lane % 8 holds k-chunk slot ^ ((row >> 1) & 7)
    const int prow = 8 * wave + (lane >> 3), pchunk = (lane & 7) ^ ((prow >> 1) & 7);
    const wa_f16 * ga = A + (size_t) min(m0 + prow, M - 1) * lda + pchunk * 8;
    const wa_f16 * gw = W + (size_t) min(n0 + prow, N - 1) * ldw + pchunk * 8;
    const int nk = K >> 6;
#define EXD_ISSUE(kt_) do { const int k0_ = min((kt_), nk - 1) * 64; const int st_ = (kt_) % EXD_NST; \
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *) (ga + k0_), (__attribute__((address_space(3))) void *) (&S[st_][0][wave * 8 * 64]), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *) (gw + k0_), (__attribute__((address_space(3))) void *) (&S[st_][1][wave * 8 * 64]), 16, 0, 0); } while (0)

    float acc[2][2][32];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[p][q][i] = 0.0f;
#pragma unroll
    for (int p = 0; p < EXD_NST - 1; ++p) EXD_ISSUE(p);

    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed when at most the (EXD_NST - 2) x 2 loads issued after it are outstanding (every stage issues 2, also past the end)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((EXD_NST - 2) * 2) : "memory");
        EXD_ISSUE(kt + EXD_NST - 1);          // into the stage read in the previous iteration: every wave is past it (barrier)
        const wa_f16 * As = S[kt % EXD_NST][0], * Ws = S[kt % EXD_NST][1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {      // two 32-element steps of the reference's loop per stage
            half8 a[2][4], w[2][4];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c) a[p][c] = *(const half8 *) (&As[(wm * 16 + li * 2 + p) * 64 + (((ks * 4 + c) ^ li) * 8)]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) w[q][c] = *(const half8 *) (&Ws[(wn * 16 + lj * 2 + q) * 64 + (((ks * 4 + c) ^ lj) * 8)]);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            acc[p][q][c * 8 + i] = fmaf((float) w[q][c][i], (float) a[p][c][i], acc[p][q][c * 8 + i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus loads of the last iterations still target this block's LDS
#undef EXD_ISSUE
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = m0 + wm * 16 + li * 2 + p, n = n0 + wn * 16 + lj * 2 + q;
            if (m < M && n < N) epi_store<EPI>(e, m, n, wa_tree32(acc[p][q]));
        }
}

// -------------------------------------------------------------------------------------------------
// The same product ON THE MATRIX CORES, still bit-identical.  v_mfma_f32_16x16x4_f32 computes, per output element,
// fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c)))) - an ascending F32 fmaf chain (tools/micro/mfma_f32_exact.hip: 0 mismatches
// in 10^5 random outputs, F16-valued and arbitrary operands, chained 24 deep).  One of the reference's 32 partial sums of an output,
// s[r] (elements k = 32 s + r, s ascending: vec.cpp:191-231), is exactly such a chain, so partial sum r of a 16 x 16 output tile is
// one accumulator tile fed with the k-elements r, 32 + r, 64 + r, 96 + r of every 128-k stage: 32 accumulator tiles (128 VGPRs) per
// wave, the final tree (wa_tree32) lane-local as before.  Rate = the F32 MFMA's (155 TFLOP/s, the same peak as the VALU form - but
// the VALU form spent its issue slots on conversions and operand traffic and reached 20 % of it).
//   * workgroup = 4 waves as 2 x 2, tile 32 x 32, wave tile 16 x 16; MX_WGS workgroups per CU (<= 168 VGPRs), each with its own
//     barrier: while one waits for a stage or issues its loads, the others' MFMAs keep the SIMD's matrix pipe busy (one 8-wave
//     workgroup per CU, every wave at the same barrier every stage, reached 43 % of the MFMA rate);
//   * operands by LDS-DMA into a ring of NST stages of 128 k (A 8 KB + W 8 KB), 4 pieces per wave and stage, one LDS-only
//     barrier per stage;
//   * a lane's fragment read is ONE 16-byte piece = its (row, k-step) element of 8 consecutive partial sums; the slot of chunk c
//     of row r is c ^ g(r) with g = (r & 15) ^ (4 if 4 <= r & 15 <= 11): conflict-free over ds_read_b128's lane groups
//     ({0-3, 12-15, 20-27}, ... - MI355X guide, LDS), which mix two k-steps.
// K % 128 == 0 (a partial sum's chain then is whole MFMAs); other shapes take the VALU kernels above.
// -------------------------------------------------------------------------------------------------
#define MX_NST 3                                             // 48 KB of LDS per workgroup
#define MX_BM 32
#define MX_BN 32
#define MX_STAGE_HALFS ((MX_BM + MX_BN) * 128)
#define MX_PCS 4                                             // 1 KB pieces per wave and stage
__device__ __forceinline__ int mx_g(int row) { const int r = row & 15; return r ^ ((r >= 4 && r <= 11) ? 4 : 0); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_exact_mfma(const wa_f16 * __restrict__ A, int lda, const wa_f16 * __restrict__ W, int ldw,
                                                            int M, int N, int K, wa_epi e) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char mx_smem[];
    wa_f16 * S = (wa_f16 *) mx_smem;                         // [MX_NST][A 32 x 128 | W 32 x 128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (as k_gemm_f16_dma): XCD x takes a contiguous eighth of the tiles, cut along the longer of M and N
    const int tiles_m = (M + MX_BM - 1) / MX_BM, tiles_n = (N + MX_BN - 1) / MX_BN, per_xcd = (tiles_m * tiles_n + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= tiles_m * tiles_n) return;
    const int bm = N > M ? t % tiles_m : t / tiles_n, bn = N > M ? t / tiles_m : t % tiles_n;
    const int m0 = bm * MX_BM, n0 = bn * MX_BN;
    A += (size_t) blockIdx.y * e.bs_a; W += (size_t) blockIdx.y * e.bs_w;       // batched launch: one problem per grid.y

    // this wave's four 1 KB pieces of a stage: A rows [8 wave, 8 wave + 8) and W rows [8 wave, 8 wave + 8); a piece = 4 rows x 16 slots
    const int prow = lane >> 4, pslot = lane & 15;
    unsigned goff[MX_PCS];                                   // byte offset of this lane's 16 bytes from A / W (uniform base + 32-bit lane offset)
    int ldst[MX_PCS];
#pragma unroll
    for (int p = 0; p < MX_PCS; ++p) {
        const int row = 8 * wave + 4 * (p & 1) + prow;
        const int chunk = pslot ^ mx_g(row);
        goff[p] = p < 2 ? ((unsigned) min(m0 + row, M - 1) * (unsigned) lda + chunk * 8) * 2u : ((unsigned) min(n0 + row, N - 1) * (unsigned) ldw + chunk * 8) * 2u;
        ldst[p] = (p < 2 ? 0 : MX_BM * 128) + (8 * wave + 4 * (p & 1)) * 128;      // piece base, wave-uniform (the DMA adds lane * 16 B itself)
    }
    const int nk = K >> 7;
    const __attribute__((address_space(1))) char * Ab = (const __attribute__((address_space(1))) char *) A, * Wb = (const __attribute__((address_space(1))) char *) W;
#define MX_ISSUE(kt_) do { const int k0_ = min((kt_), nk - 1) * 256; wa_f16 * st_ = S + ((kt_) % MX_NST) * MX_STAGE_HALFS; \
        _Pragma("unroll") for (int p = 0; p < MX_PCS; ++p) \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) ((p < 2 ? Ab : Wb) + k0_ + goff[p]), (__attribute__((address_space(3))) void *) (st_ + ldst[p]), 16, 0, 0); \
        } while (0)

    f32x4 acc[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < MX_NST - 1; ++p) MX_ISSUE(p);

    // epilogue operands now (clamped, unconditional): D row = 4 (lane / 16) + j, column = lane % 16
    const int fr = lane & 15, fk = lane >> 4;
    const int on = n0 + wn * 16 + fr, om = m0 + wm * 16 + 4 * fk;
    constexpr bool RES = EPI == WA_EPI_RESID || EPI == WA_EPI_CONV2;
    const wa_epi_pre pcol = epi_preload<RES ? WA_EPI_F32 : EPI>(e, 0, min(on, N - 1));
    float pres[4] = { 0.f, 0.f, 0.f, 0.f };
    if (RES) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pres[j] = e.resid[(size_t) min(om + j, M - 1) * e.ldr + min(on, N - 1)];
    }
    // fragment addresses: row (wm 16 + fr | wn 16 + fr), k-step fk of the stage, chunk 4 fk + c -> slot (4 fk + c) ^ g(row)
    const int ga = mx_g(fr);
    const wa_f16 * fa = S + (wm * 16 + fr) * 128, * fw = S + MX_BM * 128 + (wn * 16 + fr) * 128;

    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed when at most the (MX_NST - 2) x MX_PCS loads this wave issued after it are outstanding
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((MX_NST - 2) * MX_PCS) : "memory");
        const int so = (kt % MX_NST) * MX_STAGE_HALFS;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                     // four fragment reads in flight before the first conversion of a half
            half8 af[2], wf[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                af[c] = *(const half8 *) (fa + so + (((4 * fk + 2 * hf + c) ^ ga) * 8));
                wf[c] = *(const half8 *) (fw + so + (((4 * fk + 2 * hf + c) ^ ga) * 8));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (hf == 0) {
                MX_ISSUE(kt + MX_NST - 1);    // into the stage read in the previous iteration: every wave is past it (barrier)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[(2 * hf + c) * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32((float) af[c][i], (float) wf[c][i], acc[(2 * hf + c) * 8 + i], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus loads of the last iterations still target this block's LDS
#undef MX_ISSUE
    float res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s32[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) s32[r] = acc[r][j];
        res[j] = wa_tree32(s32);
    }
    if (EPI == WA_EPI_GELU_F16 || EPI == WA_EPI_CONV2) {       // the four table look-ups in flight together
#pragma unroll
        for (int j = 0; j < 4; ++j) { float v = res[j]; if (e.bias) v = v + pcol.bias; res[j] = wa_gelu_nb(v, e.gelu); }
    }
    if (on >= N) return;
    if (EPI == WA_EPI_ATTN_PV) {       // + the n_kv % 32 leftover cells in F64, in index order (vec.cpp:221-223), then F16
        const int np = e.aux0, nl = e.aux1;
        const half8 * vrow = (const half8 *) (W + (size_t) on * ldw + np);      // np % 32 == 0, ldw % 8 == 0: 16-byte aligned; 32 cells in bounds (ldw >= np + 32)
        const wa_f16 * pl = (const wa_f16 *) e.out2 + (size_t) blockIdx.y * e.bs_o2;
        half8 vl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) vl[c] = vrow[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(om + j, M - 1);
            const half8 * prow = (const half8 *) (pl + (size_t) m * 32);
            half8 pv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) pv[c] = prow[c];
            double sumf = (double) res[j];
            float prod[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) prod[c] = (float) vl[c >> 3][c & 7] * (float) pv[c >> 3][c & 7];
#pragma unroll
            for (int c = 0; c < 32; ++c) if (c < nl) sumf += (double) prod[c];
            if (om + j < M) {
                if (e.out3) ((float *) e.out3)[(size_t) (om + j) * e.ldo3 + blockIdx.y * 64 + on] = (float) sumf;      // quantised models: F32, quantised by the next launch
                else ((wa_f16 *) e.out)[(size_t) (om + j) * e.ldo + blockIdx.y * 64 + on] = f2h((float) sumf);
            }
        }
        return;
    }
    if (EPI == WA_EPI_ENC_QKV && on >= e.split0 && om + 3 < M) {      // V transposed: this lane's four rows are consecutive m - one 8-byte store
        unsigned short h4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { float v = res[j]; if (e.bias) v = v + pcol.bias; h4[j] = f2h(v); }
        typedef unsigned mx_u2 __attribute__((ext_vector_type(2)));
        mx_u2 pk; pk.x = (unsigned) h4[0] | ((unsigned) h4[1] << 16); pk.y = (unsigned) h4[2] | ((unsigned) h4[3] << 16);
        *(mx_u2 *) ((wa_f16 *) e.out2 + (size_t) (on - e.split0) * e.ldo2 + om) = pk;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = om + j;
        if (m >= M) continue;
        if (EPI == WA_EPI_GELU_F16) ((wa_f16 *) e.out)[(size_t) m * e.ldo + on] = f2h(res[j]);
        else if (EPI == WA_EPI_CONV2) {
            if (e.dbg) e.dbg[(size_t) m * e.ldo + on] = res[j];
            ((float *) e.out)[(size_t) m * e.ldo + on] = pres[j] + res[j];
        } else {
            wa_epi_pre pre = pcol;
            if (RES) pre.resid = pres[j];
            epi_apply<EPI>(e, m, on, res[j], pre);
        }
    }
}

template <int EPI>
static void gemm_exact_mfma_launch(hipStream_t s, const wa_f16 * A, int lda, const wa_f16 * W, int ldw, int M, int N, int K, const wa_epi & e, int batch = 1) {
    static bool attr_done = false;
    constexpr int lds = MX_NST * MX_STAGE_HALFS * 2;
    if (!attr_done) { (void) hipFuncSetAttribute((const void *) k_gemm_exact_mfma<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
    const int grid = ((((M + MX_BM - 1) / MX_BM) * ((N + MX_BN - 1) / MX_BN) + 7) / 8) * 8;
    hipLaunchKernelGGL((k_gemm_exact_mfma<EPI>), dim3(grid, batch), dim3(256), lds, s, A, lda, W, ldw, M, N, K, e);
}

void wa_launch_gemm_exact(hipStream_t s, wa_epi_mode mode, const wa_f16 * A, int lda, const wa_f16 * W, int ldw, int M, int N, int K,
                          const wa_epi & e) {
    static const bool no_mfma = getenv("WHISPER_AMD_NO_EXACT_MFMA") != nullptr;
    if (!no_mfma && K % 128 == 0 && lda % 8 == 0 && ldw % 8 == 0) {
#define WA_CASE(E) case E: gemm_exact_mfma_launch<E>(s, A, lda, W, ldw, M, N, K, e); return;
        switch (mode) {
            WA_CASE(WA_EPI_F16) WA_CASE(WA_EPI_ENC_QKV) WA_CASE(WA_EPI_GELU_F16) WA_CASE(WA_EPI_RESID)
            WA_CASE(WA_EPI_CONV2) WA_CASE(WA_EPI_F32) WA_CASE(WA_EPI_CROSS_KV) WA_CASE(WA_EPI_DEC_QKV)
            default: break;
        }
#undef WA_CASE
    }
    const int grid = ((M + 31) / 32) * ((N + 31) / 32);
    static const bool no_dma = getenv("WHISPER_AMD_NO_GEMM_DMA") != nullptr;
    const bool dma = !no_dma && K % 64 == 0 && K >= 256 && lda % 8 == 0 && ldw % 8 == 0;
#define WA_CASE(E) case E: if (dma) hipLaunchKernelGGL((k_gemm_exact_dma<E>), dim3(grid), dim3(256), 0, s, A, lda, W, ldw, M, N, K, e); \
                           else     hipLaunchKernelGGL((k_gemm_exact<E>), dim3(grid), dim3(256), 0, s, A, lda, W, ldw, M, N, K, e); break;
    switch (mode) {
        WA_CASE(WA_EPI_F16) WA_CASE(WA_EPI_ENC_QKV) WA_CASE(WA_EPI_GELU_F16) WA_CASE(WA_EPI_RESID)
        WA_CASE(WA_EPI_CONV2) WA_CASE(WA_EPI_F32) WA_CASE(WA_EPI_CROSS_KV) WA_CASE(WA_EPI_DEC_QKV)
    }
#undef WA_CASE
}

// =================================================================================================
// LayerNorm in reference order (ops.cpp:3225-3242): one wave per row.
// =================================================================================================
// LayerNorm statistics of one row by one wave, reference-order semantics.  The row is read ONCE into registers
// (NPL values per lane, d <= 64 * NPL) and mirrored into `lrow` (LDS, d floats) for the fallback; returns mean and scale.
template <int LN_NPL>
__device__ __forceinline__ void wa_ln_stats(const float * __restrict__ xr, int d, float eps, int lane, float * lrow, float (&xv)[LN_NPL],
                                            float & mean, float & scale) {
    double s = 0.0, a = 0.0;
#pragma unroll
    for (int k = 0; k < LN_NPL; ++k) { const int i = lane + 64 * k; xv[k] = xr[i < d ? i : d - 1]; }      // unconditional loads: all in flight together
#pragma unroll
    for (int k = 0; k < LN_NPL; ++k) { const int i = lane + 64 * k; if (i < d) lrow[i] = xv[k]; else xv[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < LN_NPL; ++k) { s += (double) xv[k]; a += (double) fabsf(xv[k]); }    // padding adds exact zeros
    s = wave_sum_d(s); a = wave_sum_d(a);
    float mean_hi;
    if (!wa_sum_bounds(s, a, d, mean, mean_hi, 1.0 / (double) d)) {      // a mean near zero: second-level certificate (wa_device.h), else in order
        bool same = true;
#pragma unroll
        for (int k = 0; k < LN_NPL; ++k) if (lane + 64 * k < d) same &= wa_mean_indifferent(xv[k], mean, mean_hi);
        if (!__all(same)) {
            if (lane == 0) s = wa_seq_sum_lds(lrow, d, false, 0.0f);
            s = __shfl(s, 0, WAVE);
            mean = (float) (s / (double) d);
        }
    }
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < LN_NPL; ++k) if (lane + 64 * k < d) { const float v = xv[k] - mean; s2 += (double) (v * v); }
    s2 = wave_sum_d(s2);
    float variance;
    if (!wa_sum_certain(s2, s2, d, variance)) {
        if (lane == 0) s2 = wa_seq_sum_lds(lrow, d, true, mean);
        s2 = __shfl(s2, 0, WAVE);
        variance = (float) (s2 / (double) d);
    }
    scale = 1.0f / sqrtf(variance + eps);
}

template <int LN_NPL>
__global__ __launch_bounds__(256) void k_layernorm_exact(const float * __restrict__ x, int ldx, int rows, int d, const float * __restrict__ w,
                                                         const float * __restrict__ b, float eps, wa_f16 * __restrict__ out16, int ld16,
                                                         float * __restrict__ out32, int ld32, int8_t * __restrict__ qs, float * __restrict__ qd) {
    __shared__ __attribute__((aligned(16))) float lrows[4][64 * LN_NPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float * xr = x + (size_t) row * ldx;
    float xv[LN_NPL], gw[LN_NPL], gb[LN_NPL], mean, scale;
#pragma unroll
    for (int k = 0; k < LN_NPL; ++k) { const int i = lane + 64 * k, ic = i < d ? i : d - 1; gw[k] = w[ic]; gb[k] = b[ic]; }
    wa_ln_stats<LN_NPL>(xr, d, eps, lane, lrows[wave], xv, mean, scale);
#pragma unroll
    for (int k = 0; k < LN_NPL; ++k) {
        const int i = lane + 64 * k;
        if (i < d) {
            float y = xv[k] - mean;
            y = y * scale;
            y = y * gw[k];
            y = y + gb[k];
            if (out16) out16[(size_t) row * ld16 + i] = f2h(y);
            if (out32) out32[(size_t) row * ld32 + i] = y;
            if (qs) wa_q8_store(y, row, i >> 5, i & 31, d >> 5, qs, qd);      // d % 32 == 0: a half-wave holds one block
        }
    }
}
void wa_launch_layernorm_exact(hipStream_t stream, const float * x, int ldx, int rows, int d, const float * w, const float * b, float eps,
                               wa_f16 * out16, int ld16, float * out32, int ld32, int8_t * qs, float * qd) {
    // elements per lane sized to the row (d <= 1280): 6 for d <= 384, 12 for <= 768, 16 for <= 1024
#define WA_LN_CASE(NPL) hipLaunchKernelGGL((k_layernorm_exact<NPL>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, rows, d, w, b, eps, out16, ld16, out32, ld32, qs, qd)
    if (d <= 384) WA_LN_CASE(6); else if (d <= 768) WA_LN_CASE(12); else if (d <= 1024) WA_LN_CASE(16); else WA_LN_CASE(20);
#undef WA_LN_CASE
}

// =================================================================================================
// GEMV (M <= 8 tokens, K % 32 == 0): weight streaming, HBM-bound.  8 lanes per output row: lane u owns
// elements 4u..4u+3 of every 32-element block, i.e. partial sums (j = u/2, l = 4(u%2)+e); the xor-4 / xor-2 /
// xor-1 exchanges then reproduce the reference's (s0+s2)+(s1+s3), l<->l+4 and final pairings exactly.
// Activations staged once per block in LDS; weights go straight from HBM to VGPRs (8 B per lane per step).
// =================================================================================================
// Optional fused prologue (ln.x != null): the activations are LayerNorm(x) of F32 rows, normalised and rounded to
// F16 by every block for itself (M <= 8 rows: cheaper than a separate launch on the latency-bound decode step).
struct wa_ln_in { const float * x = nullptr; int ldx = 0; const float * w = nullptr; const float * b = nullptr; float eps = 0.f; };

#define GEMV_BATCH 24      // weight loads kept in flight per lane (8 B each)
#define GEMV_THREADS 256   // 4 waves = 32 weight rows per block iteration
#define GEMV_LN_NPL 8      // LayerNorm elements per thread (K <= 2048)
#define GEMV_WLN_NPL 20    // LayerNorm elements per lane of the wave-per-row form (K <= 1280)

typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// Block-wide LayerNorm of one F32 row (reference-order semantics, certified F64 sums), written as F16 into `dst`.
// All 256 threads share the row, so each touches K/256 elements: the prologue is issue-bound, not bandwidth-bound.
__device__ __forceinline__ void wa_block_layernorm(const float * __restrict__ xr, int K, const wa_ln_in & ln, wa_f16 * __restrict__ dst,
                                                   double * __restrict__ red /*[16] shared*/, float * __restrict__ lrow /*[K] shared*/, int tid,
                                                   int8_t * __restrict__ qs = nullptr, float * __restrict__ qd = nullptr /* when set: the row as Q8_0 (wa_q8_store, K % 32 == 0) */) {
    const int lane = tid & 63, wave = tid >> 6;
    float xv[GEMV_LN_NPL], gw[GEMV_LN_NPL], gb[GEMV_LN_NPL];
#pragma unroll
    for (int k = 0; k < GEMV_LN_NPL; ++k) {
        const int i = tid + GEMV_THREADS * k;
        const bool ok = i < K;
        const int ic = ok ? i : K - 1;                       // unconditional loads: all in flight together
        xv[k] = xr[ic]; gw[k] = ln.w[ic]; gb[k] = ln.b[ic];
        if (ok) lrow[i] = xv[k]; else { xv[k] = 0.0f; gw[k] = 0.0f; gb[k] = 0.0f; }
    }
    double s = 0.0, a = 0.0;
#pragma unroll
    for (int k = 0; k < GEMV_LN_NPL; ++k) { s += (double) xv[k]; a += (double) fabsf(xv[k]); }
    s = wave_sum_d(s); a = wave_sum_d(a);
    if (lane == 0) { red[wave] = s; red[4 + wave] = a; }
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    a = (red[4] + red[5]) + (red[6] + red[7]);
    float mean, mean_hi;
    if (!wa_sum_bounds(s, a, K, mean, mean_hi, 1.0 / (double) K)) {          // block-uniform decision; second-level certificate, else in order
        bool same = true;
#pragma unroll
        for (int k = 0; k < GEMV_LN_NPL; ++k) if (tid + GEMV_THREADS * k < K) same &= wa_mean_indifferent(xv[k], mean, mean_hi);
        if (__syncthreads_and(same ? 1 : 0) == 0) {
            if (tid == 0) red[8] = wa_seq_sum_lds(lrow, K, false, 0.0f);
            __syncthreads();
            mean = (float) (red[8] / (double) K);
        }
    }
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < GEMV_LN_NPL; ++k) if (tid + GEMV_THREADS * k < K) { const float v = xv[k] - mean; s2 += (double) (v * v); }
    s2 = wave_sum_d(s2);
    if (lane == 0) red[12 + wave] = s2;
    __syncthreads();
    s2 = (red[12] + red[13]) + (red[14] + red[15]);
    float variance;
    if (!wa_sum_certain(s2, s2, K, variance)) {
        if (tid == 0) red[9] = wa_seq_sum_lds(lrow, K, true, mean);
        __syncthreads();
        variance = (float) (red[9] / (double) K);
    }
    const float scale = 1.0f / sqrtf(variance + ln.eps);
#pragma unroll
    for (int k = 0; k < GEMV_LN_NPL; ++k) {
        const int i = tid + GEMV_THREADS * k;
        if (i < K) {
            float y = xv[k] - mean;
            y = y * scale;
            y = y * gw[k];
            y = y + gb[k];
            if (dst) dst[i] = f2h(y);
            if (qs) wa_q8_store(y, 0, i >> 5, i & 31, K >> 5, qs, qd);
        }
    }
    __syncthreads();       // `red` is reused by the next row
}

// one F32 row -> LayerNorm -> Q8_0, by a whole block (the decode step of a quantised model: one wave per row, k_layernorm_exact, leaves
// a single wave with 20 elements per lane; here every thread has at most 8)
__global__ __launch_bounds__(GEMV_THREADS) void k_ln_q8_row(const float * __restrict__ x, int K, wa_ln_in ln, int8_t * __restrict__ qs, float * __restrict__ qd) {
    __shared__ double red[16];
    __shared__ __attribute__((aligned(16))) float lrow[GEMV_THREADS * GEMV_LN_NPL];
    wa_block_layernorm(x, K, ln, nullptr, red, lrow, threadIdx.x, qs, qd);
}
void wa_launch_ln_q8_row(hipStream_t stream, const float * x, int K, const float * w, const float * b, float eps, int8_t * qs, float * qd) {
    wa_ln_in ln; ln.x = x; ln.ldx = K; ln.w = w; ln.b = b; ln.eps = eps;
    hipLaunchKernelGGL(k_ln_q8_row, dim3(1), dim3(GEMV_THREADS), 0, stream, x, K, ln, qs, qd);
}

template <int MT, int EPI>
__global__ __launch_bounds__(GEMV_THREADS) void k_gemv_exact(const wa_f16 * __restrict__ A, int lda, const int32_t * __restrict__ rows, wa_ln_in ln,
                                                             const wa_f16 * __restrict__ W, int ldw, int M, int N, int K, wa_epi e) {
    extern __shared__ __attribute__((aligned(16))) wa_f16 xs[];   // [MT][K]
    __shared__ double red[16];
    __shared__ __attribute__((aligned(16))) float lrow[4 * 64 * GEMV_WLN_NPL];      // (>= GEMV_THREADS * GEMV_LN_NPL)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane & 7, slot = lane >> 3;
    const int nsteps = K >> 5;
    const int nthr = blockDim.x, rpb = 8 * (nthr >> 6);      // 256 threads = 32 rows per block iteration; 64 = 8 (narrow products: more blocks)

    // The weights do not depend on the previous kernel's output: issue the first batch of loads BEFORE staging the
    // activations, so the HBM round trip overlaps the LayerNorm / copy prologue instead of following it.
    half4v wv[GEMV_BATCH];
    auto load_batch = [&](const wa_f16 * wrow, int s0) {
#pragma unroll
        for (int b = 0; b < GEMV_BATCH; ++b) {
            const int s = s0 + b < nsteps ? s0 + b : nsteps - 1;
            wv[b] = *(const half4v *) (wrow + s * 32);
        }
    };
    {
        const int n = blockIdx.x * rpb + wave * 8 + slot;
        const int nn = n < N ? n : N - 1;
        load_batch(W + (size_t) nn * ldw + 4 * u, 0);
    }

    if (ln.x && MT > 1 && M > 1 && K <= 64 * GEMV_WLN_NPL) {
        // several rows (beam / multi-decoder steps): one wave per row, rows in parallel (the block-wide form below spends two block
        // reductions per row, one row after the other: 8-10 us of a 17 us launch at 5 rows)
        for (int m = wave; m < M; m += GEMV_THREADS / 64) {
            const int src = rows ? rows[m] : m;
            float xv[GEMV_WLN_NPL], gw[GEMV_WLN_NPL], gb[GEMV_WLN_NPL], mean, scale;
#pragma unroll
            for (int k = 0; k < GEMV_WLN_NPL; ++k) { const int i = lane + 64 * k, ic = i < K ? i : K - 1; gw[k] = ln.w[ic]; gb[k] = ln.b[ic]; }
            wa_ln_stats<GEMV_WLN_NPL>(ln.x + (size_t) src * ln.ldx, K, ln.eps, lane, lrow + (size_t) wave * 64 * GEMV_WLN_NPL, xv, mean, scale);
#pragma unroll
            for (int k = 0; k < GEMV_WLN_NPL; ++k) {
                const int i = lane + 64 * k;
                if (i < K) {
                    float y = xv[k] - mean;
                    y = y * scale;
                    y = y * gw[k];
                    y = y + gb[k];
                    xs[(size_t) m * K + i] = f2h(y);
                }
            }
        }
        for (int m = M; m < MT; ++m)
            for (int i = tid; i < K; i += GEMV_THREADS) xs[(size_t) m * K + i] = 0;
        __syncthreads();
    } else if (ln.x) {
        for (int m = 0; m < MT; ++m) {
            if (m < M) {
                const int src = rows ? rows[m] : m;
                wa_block_layernorm(ln.x + (size_t) src * ln.ldx, K, ln, xs + (size_t) m * K, red, lrow, tid);
            } else {
                for (int i = tid; i < K; i += GEMV_THREADS) xs[(size_t) m * K + i] = 0;
            }
        }
    } else {
        // 8 chunks of 16 bytes per thread and round, all loads of a round in flight together (unconditional, clamped: a predicated load
        // makes the compiler wait for every outstanding one - the K = 4d product then spent 7 serial round trips here)
        typedef unsigned stg_u4 __attribute__((ext_vector_type(4)));
        const int kc = K >> 3, total = MT * kc;
        for (int c0 = 0; c0 < total; c0 += 8 * nthr) {
            stg_u4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = min(c0 + tid + j * nthr, total - 1), m = c / kc, cc = c - m * kc;
                const int src = rows ? rows[min(m, M - 1)] : min(m, M - 1);
                v[j] = *(const stg_u4 *) (A + (size_t) src * lda + cc * 8);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + tid + j * nthr;
                if (c < total) { const int m = c / kc; *(stg_u4 *) (&xs[(size_t) c * 8]) = m < M ? v[j] : (stg_u4) (0u); }
            }
        }
    }
    __syncthreads();
    bool first = true;
    for (int nb = blockIdx.x * rpb; nb < N; nb += gridDim.x * rpb) {
        const int n = nb + wave * 8 + slot;
        const int nn = n < N ? n : N - 1;
        const wa_f16 * wrow = W + (size_t) nn * ldw + 4 * u;
        float acc[MT][4];
        wa_epi_pre pre[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][i] = 0.0f;
            if (u == 0 && m < M) pre[m] = epi_preload<EPI>(e, m, nn);    // bias / scale / residual: fetched under the K loop
        }
        auto fma_step = [&](int b, int sidx) {     // (float) of a packed half feeds v_fma_mix_f32 directly: 4 VALU ops per step and token
            const half4v w4 = wv[b];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const half4v x4 = *(const half4v *) (&xs[(size_t) m * K + sidx * 32 + 4 * u]);
                acc[m][0] = fmaf((float) w4[0], (float) x4[0], acc[m][0]);
                acc[m][1] = fmaf((float) w4[1], (float) x4[1], acc[m][1]);
                acc[m][2] = fmaf((float) w4[2], (float) x4[2], acc[m][2]);
                acc[m][3] = fmaf((float) w4[3], (float) x4[3], acc[m][3]);
            }
        };
        for (int s0 = 0; s0 < nsteps; s0 += GEMV_BATCH) {
            if (!(first && s0 == 0)) load_batch(wrow, s0);
            if (s0 + GEMV_BATCH <= nsteps) {          // full batch: straight-line code, the LDS reads pipeline freely
#pragma unroll
                for (int b = 0; b < GEMV_BATCH; ++b) fma_step(b, s0 + b);
            } else {
#pragma unroll
                for (int b = 0; b < GEMV_BATCH; ++b) if (s0 + b < nsteps) fma_step(b, s0 + b);
            }
        }
        first = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {         // directed: the result is complete in lane u == 0 of each 8-lane group
                float v = acc[m][i];
                v = v + dpp_f32<0x104>(v);          // row_shl:4  s[j] + s[j+2]
                v = v + dpp_f32<0x102>(v);          // row_shl:2  (s0+s2) + (s1+s3)
                t[i] = v + dpp_f32<0x101>(v);       // row_shl:1  a[l] + a[l+4]
            }
            const float res = (t[0] + t[1]) + (t[2] + t[3]);
            if (u == 0 && n < N && m < M) epi_apply<EPI>(e, m, n, res, pre[m]);
        }
    }
}

template <int MT>
static void gemv_exact_dispatch(hipStream_t s, wa_epi_mode mode, const wa_f16 * A, int lda, const int32_t * rows, const wa_ln_in & ln,
                                const wa_f16 * W, int ldw, int M, int N, int K, const wa_epi & e) {
    // single-row products with few output rows and no LayerNorm prologue (the three into the residual stream: N = d) run one wave per
    // block: 4 x the blocks (N / 8 instead of N / 32 - 24 blocks for d = 768 left 232 CUs idle).  Not for several rows: staging MT x K
    // activations per 8 weight rows with 64 threads costs more than the extra blocks give (5-row step 1.23 -> 1.34 ms).
    const int nthr = (MT == 1 && !ln.x && N <= 2048) ? 64 : GEMV_THREADS, rpb = 8 * (nthr / 64);
    int grid = (N + rpb - 1) / rpb;
    if (grid > 4096) grid = 4096;
    const size_t lds = (size_t) MT * K * sizeof(wa_f16);
#define WA_CASE(E) case E: { \
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_gemv_exact<MT, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
        hipLaunchKernelGGL((k_gemv_exact<MT, E>), dim3(grid), dim3(nthr), lds, s, A, lda, rows, ln, W, ldw, M, N, K, e); } break;
    switch (mode) {
        WA_CASE(WA_EPI_F16) WA_CASE(WA_EPI_GELU_F16) WA_CASE(WA_EPI_RESID) WA_CASE(WA_EPI_F32) WA_CASE(WA_EPI_DEC_QKV)
        default: break;
    }
#undef WA_CASE
}

static void gemv_exact_any(hipStream_t stream, wa_epi_mode mode, const wa_f16 * A, int lda, const int32_t * rows, const wa_ln_in & ln,
                           const wa_f16 * W, int ldw, int M, int N, int K, const wa_epi & e) {
    if (M <= 1)      gemv_exact_dispatch<1>(stream, mode, A, lda, rows, ln, W, ldw, M, N, K, e);
    else if (M <= 2) gemv_exact_dispatch<2>(stream, mode, A, lda, rows, ln, W, ldw, M, N, K, e);
    else if (M <= 4) gemv_exact_dispatch<4>(stream, mode, A, lda, rows, ln, W, ldw, M, N, K, e);
    else if (M <= 5) gemv_exact_dispatch<5>(stream, mode, A, lda, rows, ln, W, ldw, M, N, K, e);      // beam_size / best_of default to 5
    else             gemv_exact_dispatch<8>(stream, mode, A, lda, rows, ln, W, ldw, M, N, K, e);
}

void wa_launch_gemv_exact(hipStream_t stream, wa_epi_mode mode, const wa_f16 * A, int lda, const int32_t * rows, const wa_f16 * W, int ldw,
                          int M, int N, int K, const wa_epi & e) {
    gemv_exact_any(stream, mode, A, lda, rows, wa_ln_in(), W, ldw, M, N, K, e);
}

void wa_launch_ln_gemv_exact(hipStream_t stream, wa_epi_mode mode, const float * x, int ldx, const int32_t * rows, const float * ln_w,
                             const float * ln_b, float eps, const wa_f16 * W, int ldw, int M, int N, int K, const wa_epi & e) {
    wa_ln_in ln; ln.x = x; ln.ldx = ldx; ln.w = ln_w; ln.b = ln_b; ln.eps = eps;
    gemv_exact_any(stream, mode, nullptr, 0, rows, ln, W, ldw, M, N, K, e);
}

// =================================================================================================
// im2col in ggml's column order (ops.cpp:5925-5937): dst[t][ic*3 + k] = src[(t*stride + k + row0)][ic]
// src is time-major F16 with the conv's zero padding already materialised as zero rows.
// =================================================================================================
__global__ void k_im2col3(const wa_f16 * __restrict__ src, int src_ld, int row0, int stride, int IC, int OL, wa_f16 * __restrict__ dst, int dst_ld) {
    const size_t total = (size_t) OL * IC * 3;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x) {
        const int t = (int) (i / (IC * 3)), r = (int) (i - (size_t) t * IC * 3);
        const int ic = r / 3, k = r - ic * 3;
        dst[(size_t) t * dst_ld + r] = src[(size_t) (t * stride + k + row0) * src_ld + ic];
    }
}
void wa_launch_im2col3(hipStream_t s, const wa_f16 * src, int src_ld, int row0, int stride, int IC, int OL, wa_f16 * dst, int dst_ld) {
    const size_t total = (size_t) OL * IC * 3;
    int grid = (int) ((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_im2col3, dim3(grid), dim3(256), 0, s, src, src_ld, row0, stride, IC, OL, dst, dst_ld);
}

// =================================================================================================
// Attention in reference order, d_head = 64, one (query, head) per block (x RS residue splits):
//   scores  s[c] = vec_dot_f16(64; K[c], q) -> * scale (+ mask)                 (whisper.cpp:2636 / 2189 / 2732)
//   softmax exactly as ops.cpp:4792-4818                                        (F32 probabilities)
//   out[dh] = vec_dot_f16(n_kv; V[.][dh], f16(P))                               (whisper.cpp:2647 / 2202 / 2754)
// The 32 partial-sum chains of the P V product (cells c = r mod 32) are independent: RS == 1 keeps all of
// them in the block and finishes in LDS (encoder: thousands of blocks); RS == 4 spreads them over 4 blocks
// per (query, head) to put more CUs on a single decode token and leaves the tree to k_attn_combine.
// =================================================================================================
#define ATT_MAXKV WA_ATT_MAXKV      // static LDS of k_attn_exact (scores, probabilities): every caller checks n_kv against it
#define ATT_THREADS 512
#define ATT_KB 6            // keys per lane group whose K loads are issued together

template <int RS>
__global__ __launch_bounds__(ATT_THREADS) void k_attn_exact(const wa_f16 * __restrict__ q, int ldq, const wa_f16 * kbase, size_t k_head_stride,
                                                            int k_row_stride, const wa_f16 * vbase, size_t v_head_stride, int v_row_stride,
                                                            int n_kv_arg, const int8_t * __restrict__ mask, float scale, float * __restrict__ partial,
                                                            wa_f16 * __restrict__ p_left, wa_f16 * __restrict__ out, int ldo, float * __restrict__ qk_out,
                                                            const int * __restrict__ dyn, float * __restrict__ out32, int8_t * __restrict__ q8,
                                                            float * __restrict__ q8d, const wa_rowptr * __restrict__ rowp, int rowp_cross, long long rowp_off) {
    constexpr int NW = ATT_THREADS / 64;
    if (rowp) {        // query row j belongs to its own state: that state's cells (self) or encoder K / V (cross)
        const wa_rowptr r = rowp[blockIdx.y];
        kbase = (rowp_cross ? r.cross_k : r.kv_k) + rowp_off; vbase = (rowp_cross ? r.cross_v : r.kv_v) + rowp_off;
        if (!rowp_cross) n_kv_arg = r.n_kv;
    }
    const int n_kv = dyn ? dyn[0] : n_kv_arg;
    __shared__ float sc[ATT_MAXKV];
    __shared__ wa_f16 p16[ATT_MAXKV];
    __shared__ float gs[ATT_MAXKV / 8];
    __shared__ float red[NW];
    __shared__ double redd[NW];
    __shared__ float s_inv;
    __shared__ __attribute__((aligned(16))) wa_f16 qs[64];
    __shared__ float part[RS == 1 ? 32 * 64 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_head = gridDim.x / RS;
    const int h = blockIdx.x / RS, rs = blockIdx.x % RS, j = blockIdx.y;
    const wa_f16 * kp = kbase + (size_t) h * k_head_stride;
    const wa_f16 * vp = vbase + (size_t) h * v_head_stride;
    const int8_t * mrow = mask ? mask + (size_t) j * n_kv : nullptr;

    if (tid < 64) qs[tid] = q[(size_t) j * ldq + h * 64 + tid];
    __syncthreads();

    // ---- scores: 4 lanes per key; lane a owns partial sums j = a (elements 8a..8a+7 and 32+8a..32+8a+7) ----
    float lmax = -INFINITY;
    {
        const int a = tid & 3, kslot = tid >> 2;                 // 128 keys per pass
        float qa[8], qb[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) { qa[l] = h2f(qs[8 * a + l]); qb[l] = h2f(qs[32 + 8 * a + l]); }
        for (int c0 = 0; c0 < n_kv; c0 += (ATT_THREADS / 4) * ATT_KB) {
            uint4 ka[ATT_KB], kb[ATT_KB];
#pragma unroll
            for (int b = 0; b < ATT_KB; ++b) {
                int c = c0 + b * (ATT_THREADS / 4) + kslot; c = c < n_kv ? c : n_kv - 1;
                const wa_f16 * kr = kp + (size_t) c * k_row_stride;
                ka[b] = *(const uint4 *) (kr + 8 * a);
                kb[b] = *(const uint4 *) (kr + 32 + 8 * a);
            }
#pragma unroll
            for (int b = 0; b < ATT_KB; ++b) {
                const int c = c0 + b * (ATT_THREADS / 4) + kslot;
                const wa_f16 * k8a = (const wa_f16 *) &ka[b], * k8b = (const wa_f16 *) &kb[b];
                float v[8];
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    float t = fmaf(h2f(k8a[l]), qa[l], 0.0f);
                    t = fmaf(h2f(k8b[l]), qb[l], t);
                    t = t + dpp_f32<0x4e>(t);                // quad_perm [2,3,0,1]: s[j] + s[j+2]
                    v[l] = t + dpp_f32<0xb1>(t);             // quad_perm [1,0,3,2]: (s0+s2) + (s1+s3)
                }
                const float t0 = v[0] + v[4], t1 = v[1] + v[5], t2 = v[2] + v[6], t3 = v[3] + v[7];
                float r = ((t0 + t1) + (t2 + t3)) * scale;
                if (c < n_kv) {
                    if (mrow && mrow[c]) r = -INFINITY;
                    if (a == 0) sc[c] = r;
                    lmax = fmaxf(lmax, r);
                }
            }
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    float mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);

    // ---- exp: polynomial for the multiple-of-8 body, libm expf for the tail ----
    const int n8 = n_kv & ~7;
    for (int c = tid; c < n_kv; c += ATT_THREADS) sc[c] = c < n8 ? wa_expf(sc[c] - mx) : wa_expf_libm(sc[c] - mx);
    __syncthreads();
    for (int g = tid; g < (n8 >> 3); g += ATT_THREADS) {
        const float * v = &sc[g * 8];
        gs[g] = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
    }
    __syncthreads();
    {   // F64 running sum in group order (vec.cpp:278-305): summed in parallel, accepted when the order cannot matter
        // (all terms >= 0: two orders differ by <= 2 n u S; 1/S -> F32 is monotonic), else redone in order by one lane.
        double ps = 0.0;
        const int ng = n8 >> 3;
        for (int g = tid; g < ng; g += ATT_THREADS) ps += (double) gs[g];
        for (int c = n8 + tid; c < n_kv; c += ATT_THREADS) ps += (double) sc[c];
        ps = wave_sum_d(ps);
        if (lane == 0) redd[wave] = ps;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += redd[w];
            const double delta = 2.0 * (double) (ng + 8) * 0x1p-53 * sum * 1.000001;
            const float ilo = (float) (1.0 / (sum + delta)), ihi = (float) (1.0 / (sum - delta));
            if (ilo != ihi) {
                sum = 0.0;
                for (int g = 0; g < ng; ++g) sum += (double) gs[g];
                for (int c = n8; c < n_kv; ++c) sum += (double) sc[c];
                s_inv = (float) (1.0 / sum);
            } else s_inv = ilo;
        }
        __syncthreads();
    }
    const float inv = s_inv;
    for (int c = tid; c < n_kv; c += ATT_THREADS) {
        const float p = sc[c] * inv;
        if (qk_out && rs == 0) qk_out[((size_t) j * n_head + h) * n_kv + c] = p;
        p16[c] = f2h(p);
    }
    __syncthreads();

    // ---- P V: chains r = c mod 32, lane = dh ----
    const int np = n_kv & ~31, nsteps = np >> 5;
    constexpr int RPW = 32 / RS / NW;     // residues per wave
    float acc[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) acc[i] = 0.0f;
    const int r0 = rs * (32 / RS) + wave * RPW;
    constexpr int CB = 16 / RPW > 1 ? 16 / RPW : 1;       // steps whose V loads are issued together
    for (int s0 = 0; s0 < nsteps; s0 += CB) {
        wa_f16 vv[CB][RPW];
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int sidx = s0 + b < nsteps ? s0 + b : nsteps - 1;
                vv[b][i] = vp[(size_t) (sidx * 32 + r0 + i) * v_row_stride + lane];
            }
#pragma unroll
        for (int b = 0; b < CB; ++b)
            if (s0 + b < nsteps) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) acc[i] = fmaf(h2f(vv[b][i]), h2f(p16[(s0 + b) * 32 + r0 + i]), acc[i]);
            }
    }
    if (RS == 1) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) part[(r0 + i) * 64 + lane] = acc[i];
        __syncthreads();
        if (tid < 64) {
            float s32[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) s32[r] = part[r * 64 + tid];
            double sumf = (double) wa_tree32(s32);
            const int nl = n_kv - np;
            float prod[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) {     // leftover cells: loads first (independent), ordered F64 accumulation after
                const int cc = c < nl ? c : 0;
                prod[c] = h2f(vp[(size_t) (np + cc) * v_row_stride + tid]) * h2f(p16[np + cc]);
            }
#pragma unroll
            for (int c = 0; c < 32; ++c) if (c < nl) sumf += (double) prod[c];
            if (q8) wa_q8_store((float) sumf, j, 2 * h + (tid >> 5), tid & 31, ldo >> 5, q8, q8d);       // F32 hand-over, quantised (wa_quant.hip)
            else if (out32) out32[(size_t) j * ldo + h * 64 + tid] = (float) sumf;
            else out[(size_t) j * ldo + h * 64 + tid] = f2h((float) sumf);
        }
    } else {
        const size_t pb = ((size_t) j * n_head + h) * 32;
#pragma unroll
        for (int i = 0; i < RPW; ++i) partial[(pb + r0 + i) * 64 + lane] = acc[i];
        if (rs == 0 && tid < 32) p_left[pb + tid] = (np + tid < n_kv) ? p16[np + tid] : (wa_f16) 0;
    }
}

// -------------------------------------------------------------------------------------------------
// Encoder self-attention ON THE MATRIX CORES, still in reference order (T queries x T keys per head, no mask).
//   scores   a score is vec_dot_f16 over 64 elements = 32 partial sums of TWO fmaf each, then the tree.  v_mfma_f32_16x16x1_4b_f32 does a
//            rank-1 update (one fmaf per element: tools/micro/mfma_f32_exact.hip) of FOUR independent 16 x 16 blocks: lane group b = lane / 16
//            feeds block b, so with each lane holding the elements 8 b .. 8 b + 7 (and 32 + ...) of its query / key row, instruction i of
//            a step updates the partial sums {i, 8 + i, 16 + i, 24 + i} of a 16-query x 16-key tile: 16 instructions per tile, no padding
//            (the K = 4 form would waste half its slots on a 2-long chain), and wa_tree32 is lane-local over the 8 x 16 accumulators;
//   soft-max exactly k_attn_exact_mq's (ops.cpp:4792-4818): 32 lanes per query row, certified F64 total;
//   P        F16 probabilities to HBM, [head][T][kvp] with zeros from np = T & ~31 on, the leftover cells' apart ([head][T][32]);
//   P V      = a reference-order GEMM with K = np (k_gemm_exact_mfma, WA_EPI_ATTN_PV adds the leftovers in F64) against V^T.
// 110 MB of P per layer travel through HBM (< 20 us); one query per block (the VALU form above) re-read K and V from L2 instead.
// -------------------------------------------------------------------------------------------------
#define AS_Q 16
#ifndef AS_PAD
#define AS_PAD 4                                             // (was 8: rows 4 apart then start 32 banks = 0 banks apart - a 2-way conflict on every score store)
#endif
#define AS_THREADS 512                                       // 8 waves: two per SIMD (the score tiles of one hide the other's loads and LDS trips)
typedef float as_f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(AS_THREADS) void k_attn_scores_mfma(const wa_f16 * __restrict__ qk, int ldqk, int d, int T, float scale, wa_f16 * __restrict__ P,
                                                                 wa_f16 * __restrict__ p_left, int kvp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char as_smem[];
    float * sc = (float *) as_smem;                                     // [16][kvs] scores, then exponentials
    const int kvs = kvp + AS_PAD;                                       // row stride: the rows 4 fb + j of the four lane groups of a store land 16 banks apart (kvp % 32 == 0)
    __shared__ float red[AS_THREADS / 64][AS_Q];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, j0 = blockIdx.y * AS_Q;
    const int fr = lane & 15, fb = lane >> 4;

    // ---- scores: wave w takes the key tiles w, w + 4, ... ----
    float qf[2][8];
    {
        const wa_f16 * qrow = qk + (size_t) min(j0 + fr, T - 1) * ldqk + h * 64 + 8 * fb;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const half8 v = *(const half8 *) (qrow + 32 * st);
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[st][i] = (float) v[i];
        }
    }
    const wa_f16 * kcol = qk + d + h * 64 + 8 * fb;
    const int ntiles = (T + 15) >> 4;
    float lmax[4] = { -INFINITY, -INFINITY, -INFINITY, -INFINITY };
    half8 kn[2];
    {
        const wa_f16 * kr = kcol + (size_t) min(wave * 16 + fr, T - 1) * ldqk;
        kn[0] = *(const half8 *) kr; kn[1] = *(const half8 *) (kr + 32);
    }
    for (int kt = wave; kt < ntiles; kt += AS_THREADS / 64) {
        const half8 k0 = kn[0], k1 = kn[1];
        {   // the next tile's key rows fly behind this tile's arithmetic
            const wa_f16 * kr = kcol + (size_t) min((kt + AS_THREADS / 64) * 16 + fr, T - 1) * ldqk;
            kn[0] = *(const half8 *) kr; kn[1] = *(const half8 *) (kr + 32);
        }
        as_f16v acc[8];
        const as_f16v zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(qf[0][i], (float) k0[i], zero, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(qf[1][i], (float) k1[i], acc[i], 0, 0, 0);
        const int key = kt * 16 + fr;
        // partial sum r = 8 b + i sits in acc[i][4 b + j]: wa_tree32, lane-local, on 4-vectors (j = 0..3: packed F32 adds)
        f32x4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 b0 = __builtin_shufflevector(acc[i], acc[i], 0, 1, 2, 3), b1 = __builtin_shufflevector(acc[i], acc[i], 4, 5, 6, 7);
            const f32x4 b2 = __builtin_shufflevector(acc[i], acc[i], 8, 9, 10, 11), b3 = __builtin_shufflevector(acc[i], acc[i], 12, 13, 14, 15);
            a[i] = (b0 + b2) + (b1 + b3);
        }
        const f32x4 t0 = a[0] + a[4], t1 = a[1] + a[5], t2 = a[2] + a[6], t3 = a[3] + a[7];
        const f32x4 r4 = ((t0 + t1) + (t2 + t3)) * scale;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (key < T) { sc[(size_t) (4 * fb + j) * kvs + key] = r4[j]; lmax[j] = fmaxf(lmax[j], r4[j]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // maximum over the 16 lanes (keys) of a lane group
        float m = lmax[j];
        m = fmaxf(m, dpp_f32<0x128>(m)); m = fmaxf(m, dpp_f32<0x124>(m)); m = fmaxf(m, dpp_f32<0x122>(m)); m = fmaxf(m, dpp_f32<0x121>(m));
        if (fr == 0) red[wave][4 * fb + j] = m;
    }
    __syncthreads();

    // ---- soft-max: 32 lanes (two DPP rows) per query row ----
    constexpr int LPR = AS_THREADS / AS_Q;
    const int row = tid / LPR, u = tid % LPR, jq = j0 + row;
    float * srow = sc + (size_t) row * kvs;
    float mx = red[0][row];
#pragma unroll
    for (int w = 1; w < AS_THREADS / 64; ++w) mx = fmaxf(mx, red[w][row]);
    const int n8 = T & ~7, ng = n8 >> 3, np = T & ~31;
    double ps = 0.0;
    for (int g = u; g < ng; g += LPR) {
        float4 lo = *(const float4 *) (srow + 8 * g), hi = *(const float4 *) (srow + 8 * g + 4);
        lo.x = wa_expf(lo.x - mx); lo.y = wa_expf(lo.y - mx); lo.z = wa_expf(lo.z - mx); lo.w = wa_expf(lo.w - mx);
        hi.x = wa_expf(hi.x - mx); hi.y = wa_expf(hi.y - mx); hi.z = wa_expf(hi.z - mx); hi.w = wa_expf(hi.w - mx);
        *(float4 *) (srow + 8 * g) = lo; *(float4 *) (srow + 8 * g + 4) = hi;
        ps += (double) (((lo.x + hi.x) + (lo.z + hi.z)) + ((lo.y + hi.y) + (lo.w + hi.w)));
    }
    for (int c = n8 + u; c < T; c += LPR) { const float ev = wa_expf_libm(srow[c] - mx); srow[c] = ev; ps += (double) ev; }
    ps += dpp_f64<0x111>(ps); ps += dpp_f64<0x112>(ps); ps += dpp_f64<0x114>(ps); ps += dpp_f64<0x118>(ps);      // lane 15 of a DPP row: its total
    ps += dpp_f64<0x142, 0xa>(ps);                                                                                // row_bcast:15 into rows 1 and 3: lanes 31 / 63 hold a half-wave
    double tot = __shfl(ps, lane | 31, WAVE);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this row's exponentials are in LDS (same wave: in order)
    float inv;
    {
        const double delta = 2.0 * (double) (ng + 8) * 0x1p-53 * tot * 1.000001;
        const float ilo = (float) (1.0 / (tot + delta)), ihi = (float) (1.0 / (tot - delta));
        inv = ilo;
        if (ilo != ihi) {                    // the order could matter: group sums and tail in index order (vec.cpp:278-305), by one lane
            double sum = 0.0;
            if (u == 0) {
                for (int g = 0; g < ng; ++g) { const float * v = srow + 8 * g; sum += (double) (((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]))); }
                for (int c = n8; c < T; ++c) sum += (double) srow[c];
            }
            sum = __shfl(sum, lane & ~31, WAVE);
            inv = (float) (1.0 / sum);
        }
    }
    if (jq >= T) return;
    wa_f16 * prow = P + ((size_t) h * T + jq) * kvp;
    wa_f16 * plrow = p_left + ((size_t) h * T + jq) * 32;
    typedef unsigned as_u4 __attribute__((ext_vector_type(4)));
    for (int g = u; g < (kvp >> 3); g += LPR) {
        as_u4 pk = { 0u, 0u, 0u, 0u };
        const int c0 = 8 * g;
        if (c0 < np) {
            const float4 lo = *(const float4 *) (srow + c0), hi = *(const float4 *) (srow + c0 + 4);
            pk.x = (unsigned) f2h(lo.x * inv) | ((unsigned) f2h(lo.y * inv) << 16); pk.y = (unsigned) f2h(lo.z * inv) | ((unsigned) f2h(lo.w * inv) << 16);
            pk.z = (unsigned) f2h(hi.x * inv) | ((unsigned) f2h(hi.y * inv) << 16); pk.w = (unsigned) f2h(hi.z * inv) | ((unsigned) f2h(hi.w * inv) << 16);
        } else {
            for (int c = c0; c < c0 + 8 && c < T; ++c) plrow[c - np] = f2h(srow[c] * inv);      // a leftover cell: apart, P stays zero there
        }
        *(as_u4 *) (prow + c0) = pk;
    }
}

void wa_launch_attn_exact_mfma(hipStream_t s, const wa_f16 * qk, int ldqk, const wa_f16 * vt, int ldvt, int T, int d, int n_head, float scale,
                               wa_f16 * p, wa_f16 * p_left, int kvp, wa_f16 * out, int ldo, float * out32) {
    static bool attr_done = false;
    const int lds = AS_Q * (kvp + AS_PAD) * (int) sizeof(float);
    if (!attr_done) { (void) hipFuncSetAttribute((const void *) k_attn_scores_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); attr_done = true; }
    hipLaunchKernelGGL(k_attn_scores_mfma, dim3(n_head, (T + AS_Q - 1) / AS_Q), dim3(AS_THREADS), lds, s, qk, ldqk, d, T, scale, p, p_left, kvp);
    const int np = T & ~31;
    wa_epi e; e.out = out; e.ldo = ldo; e.out2 = p_left; e.bs_o2 = (long long) T * 32; e.aux0 = np; e.aux1 = T - np;
    e.out3 = out32; e.ldo3 = ldo;
    e.bs_a = (long long) T * kvp; e.bs_w = (long long) 64 * ldvt;
    gemm_exact_mfma_launch<WA_EPI_ATTN_PV>(s, p, kvp, vt, ldvt, T, 64, (np + 127) & ~127, e, n_head);
}

// -------------------------------------------------------------------------------------------------
// The same attention for NQ queries of one head per block (RS == 1 only): K and V rows are loaded ONCE per block and used for all NQ
// queries.  One query per block re-reads the head's whole K and V per query - at T = 1500 that is 6.9 GB of L2 traffic per encoder
// layer (590 us, half of the reference-order encoder); the arithmetic per query is exactly k_attn_exact's.
// Dynamic LDS: sc [NQ][kvp] f32 | gs [NQ][kvp / 8] f32 | part [NQ][32 * 64] f32 | p16 [NQ][kvp] f16, kvp = n_kv rounded up to 32.
// -------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(ATT_THREADS) void k_attn_exact_mq(const wa_f16 * __restrict__ q, int ldq, const wa_f16 * __restrict__ kbase, size_t k_head_stride,
                                                               int k_row_stride, const wa_f16 * __restrict__ vbase, size_t v_head_stride, int v_row_stride,
                                                               int n_tokens, int n_kv, int kvp, const int8_t * __restrict__ mask, float scale,
                                                               wa_f16 * __restrict__ out, int ldo, float * __restrict__ qk_out,
                                                               float * __restrict__ out32, int8_t * __restrict__ q8, float * __restrict__ q8d) {
    constexpr int NW = ATT_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char mq_smem[];
    float  * sc   = (float *) mq_smem;                                  // [NQ][kvp]
    float  * gs   = sc + (size_t) NQ * kvp;                             // [NQ][kvp / 8]
    float  * part = gs + (size_t) NQ * (kvp >> 3);                      // [NQ][32 * 64]
    wa_f16 * p16  = (wa_f16 *) (part + (size_t) NQ * 32 * 64);         // [NQ][kvp]
    __shared__ float red[NQ][NW];
    __shared__ double redd[NQ][NW];
    __shared__ float s_inv[NQ];
    __shared__ __attribute__((aligned(16))) wa_f16 qs[NQ][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_head = gridDim.x;
    const int h = blockIdx.x, j0 = blockIdx.y * NQ;
    const wa_f16 * kp = kbase + (size_t) h * k_head_stride;
    const wa_f16 * vp = vbase + (size_t) h * v_head_stride;

    if (tid < 64 * NQ) { const int qi = tid >> 6, jq = min(j0 + qi, n_tokens - 1); qs[qi][tid & 63] = q[(size_t) jq * ldq + h * 64 + (tid & 63)]; }
    __syncthreads();

    // ---- scores: 4 lanes per key; lane a owns partial sums j = a (elements 8a..8a+7 and 32+8a..32+8a+7) ----
    float lmax[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lmax[qi] = -INFINITY;
    {
        const int a = tid & 3, kslot = tid >> 2;                 // 128 keys per pass
        float qa[NQ][8], qb[NQ][8];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
            for (int l = 0; l < 8; ++l) { qa[qi][l] = h2f(qs[qi][8 * a + l]); qb[qi][l] = h2f(qs[qi][32 + 8 * a + l]); }
        constexpr int KB = 4;                                    // keys per lane group whose K loads are issued together
        typedef unsigned mq_u4 __attribute__((ext_vector_type(4)));
        for (int c0 = 0; c0 < n_kv; c0 += (ATT_THREADS / 4) * KB) {
            mq_u4 ka[KB], kb[KB];
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                int c = c0 + b * (ATT_THREADS / 4) + kslot; c = c < n_kv ? c : n_kv - 1;
                const wa_f16 * kr = kp + (size_t) c * k_row_stride;
                ka[b] = *(const mq_u4 *) (kr + 8 * a);
                kb[b] = *(const mq_u4 *) (kr + 32 + 8 * a);
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int c = c0 + b * (ATT_THREADS / 4) + kslot;
                float k8a[8], k8b[8];
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    k8a[2 * l] = h2f((wa_f16) (ka[b][l] & 0xffffu)); k8a[2 * l + 1] = h2f((wa_f16) (ka[b][l] >> 16));
                    k8b[2 * l] = h2f((wa_f16) (kb[b][l] & 0xffffu)); k8b[2 * l + 1] = h2f((wa_f16) (kb[b][l] >> 16));
                }
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    float v[8];
#pragma unroll
                    for (int l = 0; l < 8; ++l) {
                        float t = fmaf(k8a[l], qa[qi][l], 0.0f);
                        t = fmaf(k8b[l], qb[qi][l], t);
                        t = t + dpp_f32<0x4e>(t);                // quad_perm [2,3,0,1]: s[j] + s[j+2]
                        v[l] = t + dpp_f32<0xb1>(t);             // quad_perm [1,0,3,2]: (s0+s2) + (s1+s3)
                    }
                    const float t0 = v[0] + v[4], t1 = v[1] + v[5], t2 = v[2] + v[6], t3 = v[3] + v[7];
                    float r = ((t0 + t1) + (t2 + t3)) * scale;
                    if (c < n_kv) {
                        if (mask && mask[(size_t) min(j0 + qi, n_tokens - 1) * n_kv + c]) r = -INFINITY;
                        if (a == 0) sc[(size_t) qi * kvp + c] = r;
                        lmax[qi] = fmaxf(lmax[qi], r);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) { const float m = wave_max(lmax[qi]); if (lane == 0) red[qi][wave] = m; }
    __syncthreads();
    float mx[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        mx[qi] = red[qi][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mx[qi] = fmaxf(mx[qi], red[qi][w]);
    }

    // ---- exp: polynomial for the multiple-of-8 body, libm expf for the tail ----
    const int n8 = n_kv & ~7, ng = n8 >> 3;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
        for (int c = tid; c < n_kv; c += ATT_THREADS) { float * p = &sc[(size_t) qi * kvp + c]; *p = c < n8 ? wa_expf(*p - mx[qi]) : wa_expf_libm(*p - mx[qi]); }
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
        for (int g = tid; g < ng; g += ATT_THREADS) {
            const float * v = &sc[(size_t) qi * kvp + g * 8];
            gs[(size_t) qi * (kvp >> 3) + g] = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
        }
    __syncthreads();
    {   // F64 sums in group order (vec.cpp:278-305): summed in parallel, accepted when the order cannot matter, else redone in order
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            double ps = 0.0;
            for (int g = tid; g < ng; g += ATT_THREADS) ps += (double) gs[(size_t) qi * (kvp >> 3) + g];
            for (int c = n8 + tid; c < n_kv; c += ATT_THREADS) ps += (double) sc[(size_t) qi * kvp + c];
            ps = wave_sum_d(ps);
            if (lane == 0) redd[qi][wave] = ps;
        }
        __syncthreads();
        if (tid < NQ) {
            const int qi = tid;
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += redd[qi][w];
            const double delta = 2.0 * (double) (ng + 8) * 0x1p-53 * sum * 1.000001;
            const float ilo = (float) (1.0 / (sum + delta)), ihi = (float) (1.0 / (sum - delta));
            if (ilo != ihi) {
                sum = 0.0;
                for (int g = 0; g < ng; ++g) sum += (double) gs[(size_t) qi * (kvp >> 3) + g];
                for (int c = n8; c < n_kv; ++c) sum += (double) sc[(size_t) qi * kvp + c];
                s_inv[qi] = (float) (1.0 / sum);
            } else s_inv[qi] = ilo;
        }
        __syncthreads();
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const float inv = s_inv[qi];
        for (int c = tid; c < n_kv; c += ATT_THREADS) {
            const float p = sc[(size_t) qi * kvp + c] * inv;
            if (qk_out && j0 + qi < n_tokens) qk_out[((size_t) (j0 + qi) * n_head + h) * n_kv + c] = p;
            p16[(size_t) qi * kvp + c] = f2h(p);
        }
    }
    __syncthreads();

    // ---- P V: chains r = c mod 32 (4 per wave), lane = dh; the V rows are loaded once for all NQ queries ----
    const int np = n_kv & ~31, nsteps = np >> 5;
    constexpr int RPW = 32 / NW;          // residues per wave
    float acc[NQ][RPW];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int i = 0; i < RPW; ++i) acc[qi][i] = 0.0f;
    const int r0 = wave * RPW;
    constexpr int CB = 4;                 // steps whose V loads are issued together
    for (int s0 = 0; s0 < nsteps; s0 += CB) {
        wa_f16 vv[CB][RPW];
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int sidx = s0 + b < nsteps ? s0 + b : nsteps - 1;
                vv[b][i] = vp[(size_t) (sidx * 32 + r0 + i) * v_row_stride + lane];
            }
#pragma unroll
        for (int b = 0; b < CB; ++b)
            if (s0 + b < nsteps) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const float vf = h2f(vv[b][i]);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) acc[qi][i] = fmaf(vf, h2f(p16[(size_t) qi * kvp + (s0 + b) * 32 + r0 + i]), acc[qi][i]);
                }
            }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int i = 0; i < RPW; ++i) part[(size_t) qi * 2048 + (r0 + i) * 64 + lane] = acc[qi][i];
    __syncthreads();
    if (wave < NQ && j0 + wave < n_tokens) {      // wave qi finishes query qi: the tree over the 32 chains, then the leftover cells in order
        const int qi = wave, j = j0 + qi;
        float s32[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) s32[r] = part[(size_t) qi * 2048 + r * 64 + lane];
        double sumf = (double) wa_tree32(s32);
        const int nl = n_kv - np;
        float prod[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int cc = c < nl ? c : 0;
            prod[c] = h2f(vp[(size_t) (np + cc) * v_row_stride + lane]) * h2f(p16[(size_t) qi * kvp + np + cc]);
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) if (c < nl) sumf += (double) prod[c];
        if (q8) wa_q8_store((float) sumf, j, 2 * h + (lane >> 5), lane & 31, ldo >> 5, q8, q8d);
        else if (out32) out32[(size_t) j * ldo + h * 64 + lane] = (float) sumf;
        else out[(size_t) j * ldo + h * 64 + lane] = f2h((float) sumf);
    }
}

__global__ __launch_bounds__(64) void k_attn_combine(const float * __restrict__ partial, const wa_f16 * __restrict__ p_left,
                                                     const wa_f16 * vbase, size_t v_head_stride, int v_row_stride, int n_kv_arg,
                                                     wa_f16 * __restrict__ out, int ldo, const int * __restrict__ dyn, float * __restrict__ out32,
                                                     int8_t * __restrict__ q8, float * __restrict__ q8d, const wa_rowptr * __restrict__ rowp, int rowp_cross,
                                                     long long rowp_off) {
    const int j = blockIdx.x, h = blockIdx.y, n_head = gridDim.y, dh = threadIdx.x;
    if (rowp) { const wa_rowptr r = rowp[j]; vbase = (rowp_cross ? r.cross_v : r.kv_v) + rowp_off; if (!rowp_cross) n_kv_arg = r.n_kv; }
    const int n_kv = dyn ? dyn[0] : n_kv_arg;
    const size_t pb = ((size_t) j * n_head + h) * 32;
    float s32[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) s32[r] = partial[(pb + r) * 64 + dh];
    double sumf = (double) wa_tree32(s32);
    const int np = n_kv & ~31, nl = n_kv - np;
    const wa_f16 * vp = vbase + (size_t) h * v_head_stride;
    float prod[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {     // loads first (independent), ordered F64 accumulation after
        const int cc = c < nl ? c : 0;
        prod[c] = h2f(vp[(size_t) (np + cc) * v_row_stride + dh]) * h2f(p_left[pb + cc]);
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) if (c < nl) sumf += (double) prod[c];
    if (q8) wa_q8_store((float) sumf, j, 2 * h + (dh >> 5), dh & 31, ldo >> 5, q8, q8d);
    else if (out32) out32[(size_t) j * ldo + h * 64 + dh] = (float) sumf;
    else out[(size_t) j * ldo + h * 64 + dh] = f2h((float) sumf);
}

void wa_launch_attn_exact(hipStream_t s, const wa_f16 * q, int ldq, const wa_f16 * kbase, size_t k_head_stride, int k_row_stride,
                          const wa_f16 * vbase, size_t v_head_stride, int v_row_stride, int n_head, int n_tokens, int n_kv, const int8_t * mask,
                          float scale, float * partial, wa_f16 * p_left, wa_f16 * out, int ldo, float * qk_out, const int * dyn, float * out32,
                          int8_t * q8, float * q8d, const wa_rowptr * rowp, int rowp_cross, long long rowp_off) {
    // few (token, head) pairs and a long key range (decode cross-attention): spread the 32 partial-sum chains over 4 blocks
    // per pair and finish in k_attn_combine; otherwise one block per pair finishes in LDS (encoder, prompt, self-attention)
    const bool split = (long) n_tokens * n_head < 512 && n_kv > 512;
    if (!split && !dyn && !rowp && n_tokens >= 4 && n_kv <= 2048 && (long) ((n_tokens + 3) / 4) * n_head >= 256) {
        // enough (query, head) pairs to fill the chip four queries at a time: share the K / V loads between them
        constexpr int NQ = 4;
        const int kvp = (n_kv + 31) & ~31;
        const size_t lds = (size_t) NQ * kvp * 4 + (size_t) NQ * (kvp >> 3) * 4 + (size_t) NQ * 2048 * 4 + (size_t) NQ * kvp * 2;
        static bool attr_done = false;
        if (!attr_done) { (void) hipFuncSetAttribute((const void *) k_attn_exact_mq<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); attr_done = true; }
        hipLaunchKernelGGL((k_attn_exact_mq<NQ>), dim3(n_head, (n_tokens + NQ - 1) / NQ), dim3(ATT_THREADS), lds, s, q, ldq, kbase, k_head_stride, k_row_stride,
                           vbase, v_head_stride, v_row_stride, n_tokens, n_kv, kvp, mask, scale, out, ldo, qk_out, out32, q8, q8d);
        return;
    }
    if (!split) {
        hipLaunchKernelGGL((k_attn_exact<1>), dim3(n_head, n_tokens), dim3(ATT_THREADS), 0, s, q, ldq, kbase, k_head_stride, k_row_stride, vbase,
                           v_head_stride, v_row_stride, n_kv, mask, scale, partial, p_left, out, ldo, qk_out, dyn, out32, q8, q8d, rowp, rowp_cross, rowp_off);
    } else {
        hipLaunchKernelGGL((k_attn_exact<4>), dim3(n_head * 4, n_tokens), dim3(ATT_THREADS), 0, s, q, ldq, kbase, k_head_stride, k_row_stride, vbase,
                           v_head_stride, v_row_stride, n_kv, mask, scale, partial, p_left, out, ldo, qk_out, dyn, out32, q8, q8d, rowp, rowp_cross, rowp_off);
        hipLaunchKernelGGL(k_attn_combine, dim3(n_tokens, n_head), dim3(64), 0, s, partial, p_left, vbase, v_head_stride, v_row_stride, n_kv,
                           out, ldo, dyn, out32, q8, q8d, rowp, rowp_cross, rowp_off);
    }
}

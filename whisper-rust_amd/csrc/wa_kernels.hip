// wa_kernels.hip - hand-written gfx950 (CDNA4, wave64) kernels of the Whisper hot path.
//
// Numerics contract (DESIGN.md "Rounding points"): the parity target is the reference's ggml-cpu
// pipeline, which rounds every GEMM activation operand to F16, accumulates in F32, keeps K/V and the
// softmax probabilities in F16 and evaluates GELU through a 64K-entry F16 table
// (ggml-cpu.c:1331-1366, ggml.c:3929, whisper.cpp:2181-2202, vec.h:571-585).  Every kernel below
// rounds at exactly those points; only the F32 summation ORDER differs (MFMA / wave reductions).
// The file is compiled with -ffp-contract=off: an fma appears only where it is written.
#include "wa_device.h"
#include <cstdlib>

// =================================================================================================
// MFMA GEMM: C = A * W^T, both operands K-contiguous F16, F32 accumulate (v_mfma_f32_16x16x32_f16).
// Block = 256 threads = 4 waves in a 2x2 grid; block tile BM x BN x 32, wave tile (BM/2) x (BN/2).
// LDS rows are padded to 40 halfs (80 B) so the 16-lane ds_read_b128 groups hit disjoint banks.
// Roofline: MFMA-bound for the encoder shapes (AI ~ 1900 FLOP/B).
// =================================================================================================
#define GEMM_BK  32
#define GEMM_LDS 40

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void k_gemm_f16(const wa_f16 * __restrict__ A, int lda, const wa_f16 * __restrict__ W, int ldw,
                                                  int M, int N, int K, wa_epi e) {
    constexpr int TM = BM / 32, TN = BN / 32;          // 16x16 tiles per wave in m / n
    constexpr int CA = BM * 4 / 256, CB = BN * 4 / 256; // 16-byte chunks per thread per operand
    __shared__ __attribute__((aligned(16))) wa_f16 As[2][BM * GEMM_LDS];
    __shared__ __attribute__((aligned(16))) wa_f16 Bs[2][BN * GEMM_LDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
    const int m0 = bm * BM, n0 = bn * BN;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[CA], rb[CB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            int gm = m0 + row; gm = gm < M ? gm : M - 1;
            ra[i] = *(const uint4 *) (A + (size_t) gm * lda + k0 + kc * 8);
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            int gn = n0 + row; gn = gn < N ? gn : N - 1;
            rb[i] = *(const uint4 *) (W + (size_t) gn * ldw + k0 + kc * 8);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            *(uint4 *) (&As[buf][row * GEMM_LDS + kc * 8]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = tid + 256 * i, row = c >> 2, kc = c & 3;
            *(uint4 *) (&Bs[buf][row * GEMM_LDS + kc * 8]) = rb[i];
        }
    };

    const int nk = K / GEMM_BK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * GEMM_BK);

        half8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[i] = *(const half8 *) (&As[buf][(wm * (BM / 2) + i * 16 + fr) * GEMM_LDS + fg * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bf[j] = *(const half8 *) (&Bs[buf][(wn * (BN / 2) + j * 16 + fr) * GEMM_LDS + fg * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);

        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // C layout of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 16 + fg * 4 + r;
                if (m < M && n < N) epi_store<EPI>(e, m, n, acc[i][j][r]);
            }
        }
}

// -------------------------------------------------------------------------------------------------
// The same product with the operand tiles brought in by LDS-DMA (global_load_lds_dwordx4: HBM / L2 -> LDS without staging registers)
// through a ring of NST stages, so NST - 1 k-steps of loads are in flight behind the MFMAs of the current one; with register
// staging (above) every k-step waits out a full memory round trip, which is what bounds the encoder's GEMMs (K = 768 .. 3072).
// K % 64 == 0.  A wave-instruction writes 64 lanes x 16 B = 8 tile rows of 128 B contiguously; the 16-byte slot of row r holding
// k-chunk c is c ^ ((r >> 1) & 7), which makes the MFMA fragment reads (16 rows, one chunk) hit 16 distinct slots of the 256-byte
// bank space.  One LDS-only barrier per k-step.
// Tiles: BT x BT x 64 per workgroup, 2 x 2 waves.  BT = 64 (wave tile 32 x 32: 4 fragment reads feed 4 MFMAs) where only that fills
// 256 CUs (N = 768 products at M = 1500); BT = 128 (wave tile 64 x 64: 8 reads feed 16 MFMAs, a quarter of the barriers per flop)
// for the wide products (q|k|v, first MLP product, cross K/V).
// -------------------------------------------------------------------------------------------------
#define G2_BK  64
typedef __attribute__((address_space(1))) const void g2_gptr;
typedef __attribute__((address_space(3))) void g2_lptr;

template <int EPI, int BT, int NST, int NL>
__global__ __launch_bounds__(256 + 64 * NL) void k_gemm_f16_dma(const wa_f16 * __restrict__ A, int lda, const wa_f16 * __restrict__ W, int ldw,
                                                      int M, int N, int K, wa_epi e) {
    constexpr int TW = BT / 32;             // 16x16 tiles per MFMA wave and dimension
    constexpr int LPO = BT / 8 / NL;        // DMA wave-instructions (8 rows each) per loader wave, operand and stage (NL loader waves)
    constexpr bool RES = EPI == WA_EPI_RESID || EPI == WA_EPI_CONV2;       // epilogues with a per-element operand
    static_assert(BT == 64 || !RES, "the 128-tile form keeps only per-column epilogue operands in registers");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    wa_f16 (*S)[2][BT * G2_BK] = (wa_f16 (*)[2][BT * G2_BK]) smem_raw;      // [NST][2][BT x 64]

    // 4 + NL waves: 0-3 multiply (2 x 2 grid of wave tiles), the other NL only issue the LDS-DMA loads.  Issuing a 1 KiB piece holds a wave for
    // 100-185 cycles (measured here: 2048 cycles per k-step of a 128-tile when the MFMA waves issued their own 8 pieces, 512 of them
    // MFMA), so the loads get waves of their own and the two kinds of issue overlap on each SIMD.
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const bool loader = wave8 >= 4;
    const int wave = loader ? wave8 - 4 : wave8;        // index among the multiplying / the loading waves
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    // XCD-aware tile order.  Workgroups go round-robin to the 8 XCDs (workgroup b -> XCD b % 8), each with its own 4 MB L2: XCD x takes a
    // contiguous eighth of the tiles, cut along the longer of M and N, so it streams the smaller operand's panel once and only its own
    // eighth of the other (TCC hit rate 91 % on the encoder's products, profiles/).
    const int tiles_m = (M + BT - 1) / BT, tiles_n = (N + BT - 1) / BT, per_xcd = (tiles_m * tiles_n + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= tiles_m * tiles_n) return;
    const int bm = N > M ? t % tiles_m : t / tiles_n, bn = N > M ? t / tiles_m : t % tiles_n;
    const int m0 = bm * BT, n0 = bn * BT;

    // this lane's LPO row groups (8 rows each) of either operand: group g = LPO * wave + i, row = 8 g + lane / 8, slot = lane % 8
    const wa_f16 * ga[LPO], * gb[LPO];
#pragma unroll
    for (int i = 0; i < LPO; ++i) {
        const int row = (LPO * wave + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        ga[i] = A + (size_t) min(m0 + row, M - 1) * lda + chunk * 8;
        gb[i] = W + (size_t) min(n0 + row, N - 1) * ldw + chunk * 8;
    }
    const int nk = K / G2_BK;
#define G2_ISSUE(kt_) do { const int k0_ = min((kt_), nk - 1) * G2_BK; const int st_ = (kt_) % NST; \
        _Pragma("unroll") for (int i = 0; i < LPO; ++i) { \
            __builtin_amdgcn_global_load_lds((g2_gptr *) (ga[i] + k0_), (g2_lptr *) (&S[st_][0][(LPO * wave + i) * 8 * G2_BK]), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((g2_gptr *) (gb[i] + k0_), (g2_lptr *) (&S[st_][1][(LPO * wave + i) * 8 * G2_BK]), 16, 0, 0); \
        } } while (0)

    f32x4 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (loader) {
#pragma unroll
        for (int p = 0; p < NST - 1; ++p) G2_ISSUE(p);
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed when at most the (NST - 2) x 2 LPO loads issued after it are outstanding (every stage issues 2 LPO, also past the end)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NST - 2) * 2 * LPO) : "memory");
            G2_ISSUE(kt + NST - 1);         // into the stage read in the previous iteration: every MFMA wave is past it (barrier)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus loads of the last iterations still target this block's LDS
        return;
    }

    const int fr = lane & 15, fg = lane >> 4, sw = (fr >> 1) & 7;
    // The V third of the encoder's fused q|k|v product is stored transposed ([d][T], the P V product's B operand): its blocks compute
    // the transposed tile (operand roles swapped), so that the 16 lanes of an accumulator row hold consecutive t - 32-byte runs
    // instead of 2-byte stores a row apart.
    const bool tr = EPI == WA_EPI_ENC_QKV && n0 >= e.split0;
    const int oa = tr ? 1 : 0, ra = (tr ? wn : wm) * (BT / 2), rb = (tr ? wm : wn) * (BT / 2);
    // epilogue operands now (clamped indices, unconditional: a predicated load after the loop would cost a serial round trip each).
    // Per-column operands (bias, scale): one per n this lane owns; per-element ones (residual) only in the 64-tile form.
    wa_epi_pre pcol[TW][4];        // non-transposed: [j][0] by column n(j); transposed: [i][r] by row n(i, r)
    float pres[RES ? TW : 1][RES ? TW : 1][4];
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (tr || r == 0) pcol[a][r] = epi_preload<EPI == WA_EPI_RESID || EPI == WA_EPI_CONV2 ? WA_EPI_F32 : EPI>(e, 0, min(tr ? n0 + wn * (BT / 2) + a * 16 + fg * 4 + r : n0 + wn * (BT / 2) + a * 16 + fr, N - 1));
    if (RES) {
#pragma unroll
        for (int i = 0; i < (RES ? TW : 1); ++i)
#pragma unroll
            for (int j = 0; j < (RES ? TW : 1); ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pres[i][j][r] = e.resid[(size_t) min(m0 + wm * (BT / 2) + i * 16 + fg * 4 + r, M - 1) * e.ldr + min(n0 + wn * (BT / 2) + j * 16 + fr, N - 1)];
    }
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_barrier" ::: "memory");          // the loader waves have seen stage kt land
        const wa_f16 * As = S[kt % NST][oa], * Bs = S[kt % NST][oa ^ 1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 af[TW], bf[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) af[i] = *(const half8 *) (&As[(ra + i * 16 + fr) * G2_BK + (((ks * 4 + fg) ^ sw) * 8)]);
#pragma unroll
            for (int j = 0; j < TW; ++j) bf[j] = *(const half8 *) (&Bs[(rb + j * 16 + fr) * G2_BK + (((ks * 4 + fg) ^ sw) * 8)]);
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
#undef G2_ISSUE
    if (EPI == WA_EPI_GELU_F16 || EPI == WA_EPI_CONV2) {       // GELU table look-ups of the whole tile first, all in flight together
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r];
                    if (e.bias) v = v + pcol[j][0].bias;
                    acc[i][j][r] = wa_gelu_nb(v, e.gelu);
                }
    }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = tr ? n0 + wn * (BT / 2) + i * 16 + fg * 4 + r : n0 + wn * (BT / 2) + j * 16 + fr;
                const int m = tr ? m0 + wm * (BT / 2) + j * 16 + fr : m0 + wm * (BT / 2) + i * 16 + fg * 4 + r;
                if (m >= M || n >= N) continue;
                if (EPI == WA_EPI_GELU_F16) {
                    ((wa_f16 *) e.out)[(size_t) m * e.ldo + n] = f2h(acc[i][j][r]);
                } else if (EPI == WA_EPI_CONV2) {
                    if (e.dbg) e.dbg[(size_t) m * e.ldo + n] = acc[i][j][r];
                    ((float *) e.out)[(size_t) m * e.ldo + n] = pres[RES ? i : 0][RES ? j : 0][r] + acc[i][j][r];
                } else {
                    wa_epi_pre pre = tr ? pcol[i][r] : pcol[j][0];
                    if (RES) pre.resid = pres[RES ? i : 0][RES ? j : 0][r];
                    epi_apply<EPI>(e, m, n, acc[i][j][r], pre);
                }
            }
        }
}

template <int BT, int NST, int NL>
static void gemm_dma_dispatch(hipStream_t s, wa_epi_mode mode, const wa_f16 * A, int lda, const wa_f16 * W, int ldw, int M, int N, int K, const wa_epi & e) {
    const int grid = ((((M + BT - 1) / BT) * ((N + BT - 1) / BT) + 7) / 8) * 8;        // a multiple of the 8 XCDs (see the tile order in the kernel)
    constexpr int lds = NST * 2 * BT * G2_BK * 2;
#define WA_GEMM_CASE(E) case E: { \
        static bool attr_done = false; \
        if (!attr_done && lds > 64 * 1024) { (void) hipFuncSetAttribute((const void *) k_gemm_f16_dma<E, BT, NST, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; } \
        hipLaunchKernelGGL((k_gemm_f16_dma<E, BT, NST, NL>), dim3(grid), dim3(256 + 64 * NL), lds, s, A, lda, W, ldw, M, N, K, e); } break;
    switch (mode) {
        WA_GEMM_CASE(WA_EPI_F16)
        WA_GEMM_CASE(WA_EPI_ENC_QKV)
        WA_GEMM_CASE(WA_EPI_GELU_F16)
        WA_GEMM_CASE(WA_EPI_F32)
        WA_GEMM_CASE(WA_EPI_CROSS_KV)
        WA_GEMM_CASE(WA_EPI_DEC_QKV)
        default:
            if constexpr (BT == 64) {
                switch (mode) { WA_GEMM_CASE(WA_EPI_RESID) WA_GEMM_CASE(WA_EPI_CONV2) default: break; }
            }
            break;
    }
#undef WA_GEMM_CASE
}

template <int BM, int BN>
static void gemm_dispatch(hipStream_t s, wa_epi_mode mode, const wa_f16 * A, int lda, const wa_f16 * W, int ldw, int M, int N, int K,
                          const wa_epi & e) {
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
#define WA_GEMM_CASE(E) case E: hipLaunchKernelGGL((k_gemm_f16<BM, BN, E>), dim3(grid), dim3(256), 0, s, A, lda, W, ldw, M, N, K, e); break;
    switch (mode) {
        WA_GEMM_CASE(WA_EPI_F16)
        WA_GEMM_CASE(WA_EPI_ENC_QKV)
        WA_GEMM_CASE(WA_EPI_GELU_F16)
        WA_GEMM_CASE(WA_EPI_RESID)
        WA_GEMM_CASE(WA_EPI_CONV2)
        WA_GEMM_CASE(WA_EPI_F32)
        WA_GEMM_CASE(WA_EPI_CROSS_KV)
        WA_GEMM_CASE(WA_EPI_DEC_QKV)
    }
#undef WA_GEMM_CASE
}

void wa_launch_gemm(hipStream_t stream, wa_epi_mode mode, const wa_f16 * A, int lda, const wa_f16 * W, int ldw, int M, int N, int K,
                    const wa_epi & e) {
    const long big = (long) ((M + 127) / 128) * ((N + 127) / 128);      // number of 128x128 tiles
    static const bool no_dma = getenv("WHISPER_AMD_NO_GEMM_DMA") != nullptr;
    if (!no_dma && K % G2_BK == 0 && K / G2_BK >= 4 && lda % 8 == 0 && ldw % 8 == 0) {
        // 64-tiles (4 stages = 64 KB, two workgroups per CU) everywhere.  128-tiles (3 stages = 96 KB, one workgroup per CU) halve the
        // bytes a CU pulls in per flop, but measured slower on ggml-small's wide products at M = 1500 (first MLP product 30 vs 24 us,
        // cross K/V 159 vs 119 us; q|k|v 17 vs 19): one workgroup per CU leaves the loader waves' issue rate and the epilogue exposed.
        // WHISPER_AMD_GEMM_128=1 selects them for products without a per-element epilogue operand.
        static const bool big_tiles = getenv("WHISPER_AMD_GEMM_128") != nullptr;
        const bool res = mode == WA_EPI_RESID || mode == WA_EPI_CONV2;
        // (4 loader waves: 8 of them - 2 pieces each - measured the same 1.50 ms per encoder pass: the CU's LDS-DMA intake is the limit,
        //  not the issue rate of a wave)
        if (big_tiles && !res && big >= 200) gemm_dma_dispatch<128, 3, 4>(stream, mode, A, lda, W, ldw, M, N, K, e);
        else                                 gemm_dma_dispatch<64, 4, 4>(stream, mode, A, lda, W, ldw, M, N, K, e);
        return;
    }
    gemm_dispatch<64, 64>(stream, mode, A, lda, W, ldw, M, N, K, e);        // K % 64 != 0 or K < 256: the register-staged form
}

// =================================================================================================
// log-mel spectrogram (whisper.cpp:3076-3276 restated for one workgroup per frame)
//   frame -> Hann -> radix-2 decimation 400->200->100->50->25 with naive 25-point DFT leaves (same
//   butterfly structure and table-driven twiddles as the reference, F32) -> power -> 80x201 filterbank
//   accumulated in F64 from F32 4-term partial sums -> log10 -> F32.  Global max by ordered atomicMax,
//   then clamp/scale in a second kernel (the reference's barrier between its two elementwise phases).
// =================================================================================================
__device__ __forceinline__ float pcm_padded(const float * __restrict__ pcm, int n, int p) {
    // reflect 200 samples in front (whisper.cpp:3217), zeros behind (whisper.cpp:3214)
    const int i = p < 200 ? 200 - p : p - 200;
    return i < n ? pcm[i] : 0.0f;
}

__device__ __forceinline__ unsigned int f32_ordered(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void k_mel_frames(const float * __restrict__ pcm, int n_samples, const float * __restrict__ hann,
                                                    const float * __restrict__ sincos, const float * __restrict__ filters, int n_mel,
                                                    int n_bins, float * __restrict__ mel, int n_len, int n_active,
                                                    unsigned int * __restrict__ mel_max) {
    __shared__ float xin[400];
    __shared__ float2 bufA[400];
    __shared__ float2 bufB[400];
    __shared__ float pw[208];
    __shared__ float s_sin[400], s_cos[400];
    __shared__ float s_max[4];

    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const float neg10 = (float) -10.0;   // (float) log10(1e-10)  (whisper.cpp:3177)

    if (frame >= n_active) {
        for (int j = tid; j < n_mel; j += 256) mel[(size_t) j * n_len + frame] = neg10;
        if (tid == 0) atomicMax(mel_max, f32_ordered(neg10));
        return;
    }

    const int n_w = n_samples + 200;           // samples visible to the worker (whisper.cpp:3232)
    const int offset = frame * 160;
    for (int j = tid; j < 400; j += 256) {
        s_sin[j] = sincos[j];
        s_cos[j] = sincos[400 + j];
        xin[j] = (offset + j < n_w) ? hann[j] * pcm_padded(pcm, n_samples, offset + j) : 0.0f;
    }
    __syncthreads();

    // leaves: 16 DFTs of 25 points over x[o + 16 n]  (whisper.cpp:3054-3070)
    for (int t = tid; t < 400; t += 256) {
        const int o = t / 25, k = t - o * 25;
        float re = 0.f, im = 0.f;
        for (int n = 0; n < 25; ++n) {
            const int idx = (k * n * 16) % 400;
            const float v = xin[o + 16 * n];
            re = fmaf(v, s_cos[idx], re);
            im = fmaf(-v, s_sin[idx], im);
        }
        bufA[o * 25 + k] = make_float2(re, im);
    }
    __syncthreads();

    // four radix-2 combine levels (whisper.cpp:3103-3117): node (o, s) <- even (o, 2s), odd (o + s, 2s)
    float2 * src = bufA;
    float2 * dst = bufB;
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl) {
        const int s = 8 >> lvl;              // stride of the node being produced: 8, 4, 2, 1
        const int half_n = 25 << lvl;        // size of each child: 25, 50, 100, 200
        if (tid < 200) {
            const int o = tid / half_n, k = tid - o * half_n;
            const float2 E = src[o * half_n + k];
            const float2 O = src[(o + s) * half_n + k];
            const int idx = k * s;
            const float re = s_cos[idx], im = -s_sin[idx];
            float2 lo, hi;
            lo.x = fmaf(-im, O.y, fmaf(re, O.x, E.x));
            lo.y = fmaf(im, O.x, fmaf(re, O.y, E.y));
            hi.x = fmaf(im, O.y, fmaf(-re, O.x, E.x));
            hi.y = fmaf(-im, O.x, fmaf(-re, O.y, E.y));
            dst[o * 2 * half_n + k] = lo;
            dst[o * 2 * half_n + k + half_n] = hi;
        }
        __syncthreads();
        float2 * t2 = src; src = dst; dst = t2;
    }
    // after 4 swaps the spectrum is in `src`
    if (tid < n_bins) {
        const float2 v = src[tid];
        pw[tid] = fmaf(v.x, v.x, v.y * v.y);   // gcc's contraction in the reference build: fma(re, re, im*im)
    }
    __syncthreads();

    float lmax = -3.0e38f;
    if (tid < n_mel) {
        const float * f = filters + (size_t) tid * n_bins;
        double sum = 0.0;
        int k = 0;
        for (; k < n_bins - 3; k += 4) {
            float s4 = pw[k + 1] * f[k + 1];          // order of the reference build's contraction:
            s4 = fmaf(pw[k], f[k], s4);               // ((p1 f1 + p0 f0) + p2 f2) + p3 f3
            s4 = fmaf(pw[k + 2], f[k + 2], s4);
            s4 = fmaf(pw[k + 3], f[k + 3], s4);
            sum += (double) s4;
        }
        for (; k < n_bins; ++k) sum += (double) (pw[k] * f[k]);
        sum = log10(sum > 1e-10 ? sum : 1e-10);
        const float out = (float) sum;
        mel[(size_t) tid * n_len + frame] = out;
        lmax = out;
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) s_max[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        atomicMax(mel_max, f32_ordered(m));
    }
}

__global__ void k_mel_norm(float * __restrict__ mel, size_t n, const unsigned int * __restrict__ mel_max) {
    const double mmax = (double) f32_unordered(*mel_max) - 8.0;         // whisper.cpp:3245-3252
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        float v = mel[i];
        if ((double) v < mmax) v = (float) mmax;
        mel[i] = (float) (((double) v + 4.0) / 4.0);
    }
}

void wa_launch_mel(hipStream_t stream, const float * pcm, int n_samples, const float * hann, const float * sincos,
                   const float * filters, int n_mel, int n_fft_bins, float * mel, int n_len, unsigned int * mel_max) {
    (void) hipMemsetAsync(mel_max, 0, sizeof(unsigned int), stream);
    int n_active = (n_samples + 200) / 160 + 1;
    if (n_active > n_len) n_active = n_len;
    hipLaunchKernelGGL(k_mel_frames, dim3(n_len), dim3(256), 0, stream, pcm, n_samples, hann, sincos, filters, n_mel, n_fft_bins, mel,
                       n_len, n_active, mel_max);
    const size_t n = (size_t) n_mel * n_len;
    int grid = (int) ((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_mel_norm, dim3(grid), dim3(256), 0, stream, mel, n, mel_max);
}

// mel[n_mel][n_len] f32 -> melT[1 + t][ic] f16 for t in [0, n_frames); zero elsewhere (whisper.cpp:2399-2418)
__global__ void k_mel_window(const float * __restrict__ mel, int n_mel, int n_len, int seek, int n_frames, wa_f16 * __restrict__ melT,
                             int rows_total) {
    const size_t total = (size_t) rows_total * n_mel;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x) {
        const int row = (int) (i / n_mel), ic = (int) (i - (size_t) row * n_mel);
        const int t = row - 1;
        float v = 0.0f;
        if (t >= 0 && t < n_frames && seek + t < n_len) v = mel[(size_t) ic * n_len + seek + t];
        melT[i] = f2h(v);
    }
}

void wa_launch_mel_window(hipStream_t stream, const float * mel, int n_mel, int n_len, int seek, int n_frames, wa_f16 * melT,
                          int rows_total) {
    const size_t total = (size_t) rows_total * n_mel;
    int grid = (int) ((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_mel_window, dim3(grid), dim3(256), 0, stream, mel, n_mel, n_len, seek, n_frames, melT, rows_total);
}

// =================================================================================================
// LayerNorm of the tolerance path (flash_attn = true): one wave per row, 16-byte loads, F32 sums (two-pass variance as in
// ops.cpp:3225-3242, but in F32 and in wave order - within the path's 1e-3 contract; the reference-order kernel with its
// certified F64 sums is k_layernorm_exact).  d % 4 == 0, d <= 256 NV.
// =================================================================================================
template <int NV>
__global__ __launch_bounds__(256) void k_layernorm(const float * __restrict__ x, int ldx, int rows, int d, const float * __restrict__ w,
                                                   const float * __restrict__ b, float eps, wa_f16 * __restrict__ out16, int ld16,
                                                   float * __restrict__ out32, int ld32) {
    typedef float ln_f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 ln_h4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float * xr = x + (size_t) row * ldx;
    const int nv = d >> 2;
    ln_f4 xv[NV], gw[NV], gb[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {      // unconditional loads (clamped index): all in flight together
        const int i = lane + 64 * k, ic = i < nv ? i : nv - 1;
        xv[k] = *(const ln_f4 *) (xr + 4 * ic); gw[k] = *(const ln_f4 *) (w + 4 * ic); gb[k] = *(const ln_f4 *) (b + 4 * ic);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) if (lane + 64 * k < nv) s += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
    const float mean = wave_sum(s) / (float) d;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) if (lane + 64 * k < nv) { const ln_f4 v = xv[k] - mean; s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
    const float scale = 1.0f / sqrtf(wave_sum(s2) / (float) d + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            const ln_f4 y = (xv[k] - mean) * scale * gw[k] + gb[k];
            if (out16) { ln_h4 h; h.x = (_Float16) y.x; h.y = (_Float16) y.y; h.z = (_Float16) y.z; h.w = (_Float16) y.w; *(ln_h4 *) (out16 + (size_t) row * ld16 + 4 * i) = h; }
            if (out32) *(ln_f4 *) (out32 + (size_t) row * ld32 + 4 * i) = y;
        }
    }
}
void wa_launch_layernorm(hipStream_t stream, const float * x, int ldx, int rows, int d, const float * w, const float * b, float eps,
                         wa_f16 * out16, int ld16, float * out32, int ld32) {
#define WA_LN_CASE(NV) hipLaunchKernelGGL((k_layernorm<NV>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, rows, d, w, b, eps, out16, ld16, out32, ld32)
    if (d <= 512) WA_LN_CASE(2); else if (d <= 768) WA_LN_CASE(3); else if (d <= 1024) WA_LN_CASE(4); else WA_LN_CASE(5);
#undef WA_LN_CASE
}

// =================================================================================================
// Encoder self-attention, d_head = 64 (whisper.cpp:2181-2206 semantics, tolerance path):
//   S = Q K^T (F16 operands, F32 acc) ; P = softmax(scale * S) in F32 over all keys ; P -> F16 ;
//   O = P V (F16 operands, F32 acc) -> F16 (it is the next GEMM's A operand).
// One sweep over the keys with a running row maximum (base-2 exponentials, scale * log2(e) folded into one multiply):
// P~ = 2^(s - m) goes to F16 unnormalised, O is rescaled when the maximum moves (skipped when it did not, which is the
// rule after the first few tiles) and divided by the row sum at the end.  The kernel is VALU-bound (one exponential and
// ~8 other operations per score against 1/8 MFMA per score), so everything per score that can go has gone: one FMA + one
// v_exp_f32 + one conversion per score, the row maximum over the 16 lanes of a row is four DPP rotations, and the row sums
// come from the matrix pipe (P times a column of ones).
// Block = 4 waves x 16 query rows; K and V^T tiles of 64 keys by LDS-DMA into a ring of 4 stages (three tiles of loads in
// flight, XOR-swizzled slots); one LDS-only barrier per tile; P crosses LDS once per tile (per-wave scratch, 72-half rows,
// no barrier: a wave's LDS operations complete in order).
// =================================================================================================
#define ATT_LD 72
#define ATT_NST 4

__global__ __launch_bounds__(512) void k_enc_attn(const wa_f16 * __restrict__ qk, int ldqk, const wa_f16 * __restrict__ vt, int ldvt, int T,
                                                  int d, int n_head, float scale, wa_f16 * __restrict__ out, int ldo) {
    // K and V^T tiles (64 rows x 128 B each) arrive by LDS-DMA in a ring of ATT_NST stages, three tiles ahead of the arithmetic; same
    // XOR-swizzled 16-byte slots as k_gemm_f16_dma (LDS-DMA cannot pad rows)
    __shared__ __attribute__((aligned(1024))) wa_f16 KV[ATT_NST][2][64 * 64];
    __shared__ __attribute__((aligned(16))) wa_f16 Ps[4][16 * ATT_LD];

    // 8 waves: 0-3 own 16 query rows each, 4-7 only issue the LDS-DMA pieces (as in k_gemm_f16_dma: issuing one holds a wave 100+ cycles)
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const int fr = lane & 15, fg = lane >> 4, sw = (fr >> 1) & 7;
    // XCD-aware order (workgroup b -> XCD b % 8): an XCD takes a contiguous eighth of the (head, query tile) pairs, head-major, so a
    // head's K / V (384 KB at T = 1500) is streamed into one or two L2s instead of all eight
    const int n_tiles = (T + 63) / 64;
    const int per_xcd = (n_head * n_tiles + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= n_head * n_tiles) return;
    const int h = t / n_tiles;
    const int q0 = (t % n_tiles) * 64 + wave * 16;

    // Q fragments: A operand, row = fr, k = dh
    half8 qf[2];
    {
        int q = q0 + fr; q = q < T ? q : T - 1;
        const wa_f16 * qp = qk + (size_t) q * ldqk + h * 64;
        qf[0] = *(const half8 *) (qp + fg * 8);
        qf[1] = *(const half8 *) (qp + 32 + fg * 8);
    }
    const float c2 = scale * 1.44269504088896340736f;

    // this lane's two row groups (8 rows each) of either tile: group g = 2 * wave + i, row = 8 g + lane / 8, 16-byte slot = lane % 8
    const wa_f16 * kbase = qk + d + h * 64;
    const wa_f16 * vbase = vt + (size_t) h * 64 * ldvt;
    int krow[2], kch[2];
    const wa_f16 * gv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * wave + i) * 8 + (lane >> 3);
        kch[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        krow[i] = row;
        gv[i] = vbase + (size_t) row * ldvt + kch[i];
    }
#define ATT_ISSUE(kt_) do { const int t0_ = min((kt_), n_tiles - 1) * 64; const int st_ = (kt_) % ATT_NST; \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
            __builtin_amdgcn_global_load_lds((g2_gptr *) (kbase + (size_t) min(t0_ + krow[i], T - 1) * ldqk + kch[i]), \
                                             (g2_lptr *) (&KV[st_][0][(2 * wave + i) * 8 * 64]), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((g2_gptr *) (gv[i] + t0_), (g2_lptr *) (&KV[st_][1][(2 * wave + i) * 8 * 64]), 16, 0, 0); \
        } } while (0)

    float m_run[4];
    f32x4 o_acc[4], l_acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) m_run[r] = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) o_acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (_Float16) 1.0f;

    if (loader) {
#pragma unroll
        for (int p = 0; p < ATT_NST - 1; ++p) ATT_ISSUE(p);
        for (int kt = 0; kt < n_tiles; ++kt) {
            // tile kt has landed when at most the (ATT_NST - 2) x 4 loads issued after it are outstanding (every stage issues 4, also past the end)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((ATT_NST - 2) * 4) : "memory");
            ATT_ISSUE(kt + ATT_NST - 1);    // into the stage read in the previous iteration: every computing wave is past it (barrier)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus loads of the last iterations still target this block's LDS
        return;
    }
    for (int kt = 0; kt < n_tiles; ++kt) {
        asm volatile("s_barrier" ::: "memory");          // the loader waves have seen tile kt land
        const wa_f16 * Ks = KV[kt % ATT_NST][0], * Vs = KV[kt % ATT_NST][1];
        f32x4 s[4];         // raw dot products; the soft-max runs in base 2 on s * c2
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 b = *(const half8 *) (&Ks[(nt * 16 + fr) * 64 + (((ks * 4 + fg) ^ sw) * 8)]);
                s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[ks], b, s[nt], 0, 0, 0);
            }
        }
        if (kt == n_tiles - 1) {        // only the last tile has keys past the end
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (kt * 64 + nt * 16 + fr >= T) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[nt][r] = -INFINITY;
                }
        }
        float corr[4], mc[4];
        bool moved = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = fmaxf(fmaxf(s[0][r], s[1][r]), fmaxf(s[2][r], s[3][r]));
            // the maximum over the row's 16 lanes, in every one of them: four rotations, one VALU operation each (s_nop 1: a DPP read of
            // a VGPR needs two wait states after the VALU write)
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(t));
            const float m_new = fmaxf(m_run[r], t);        // finite from the first tile on: keys 0..63 exist for every row
            moved = moved || (m_new != m_run[r]);
            corr[r] = __builtin_amdgcn_exp2f((m_run[r] - m_new) * c2);
            m_run[r] = m_new;
            mc[r] = -m_new * c2;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[nt][r], c2, mc[r]));
                Ps[wave][(fg * 4 + r) * ATT_LD + nt * 16 + fr] = __builtin_bit_cast(wa_f16, (_Float16) p);
            }
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {      // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                l_acc[r] = l_acc[r] * corr[r];
#pragma unroll
                for (int j = 0; j < 4; ++j) o_acc[j][r] = o_acc[j][r] * corr[r];
            }
        }
        asm volatile("" ::: "memory");       // Ps[wave] is private to the wave and LDS operations of a wave complete in order: no barrier
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8 a = *(const half8 *) (&Ps[wave][fr * ATT_LD + ks * 32 + fg * 8]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const half8 b = *(const half8 *) (&Vs[(j * 16 + fr) * 64 + (((ks * 4 + fg) ^ sw) * 8)]);
                o_acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, o_acc[j], 0, 0, 0);
            }
            l_acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, ones, l_acc, 0, 0, 0);      // row sums of the F16 probabilities, on the matrix pipe
        }
    }
#undef ATT_ISSUE
    float inv_l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) inv_l[r] = (float) (1.0 / (double) l_acc[r]);   // ops.cpp:4815-4818 (every column of l_acc holds the row sum)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + fg * 4 + r;
            if (q < T) out[(size_t) q * ldo + h * 64 + j * 16 + fr] = f2h(o_acc[j][r] * inv_l[r]);
        }
}

void wa_launch_enc_attn(hipStream_t stream, const wa_f16 * qk, int ldqk, const wa_f16 * vt, int ldvt, int T, int d, int n_head, float scale,
                        wa_f16 * out, int ldo) {
    const int grid = ((((T + 63) / 64) * n_head + 7) / 8) * 8;
    hipLaunchKernelGGL(k_enc_attn, dim3(grid), dim3(512), 0, stream, qk, ldqk, vt, ldvt, T, d, n_head, scale, out, ldo);
}

// =================================================================================================
// decoder
// =================================================================================================
// x[j] = F32(token_embedding[tok]) + positional_embedding[pos]      (whisper.cpp:2531-2534)
__global__ void k_dec_embed(const int32_t * __restrict__ tok, const int32_t * __restrict__ pos, int n_tokens, int d,
                            const wa_f16 * __restrict__ te, const float * __restrict__ pe, float * __restrict__ x) {
    const int j = blockIdx.x;
    const int t = tok[j], p = pos[j];
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(size_t) j * d + i] = h2f(te[(size_t) t * d + i]) + pe[(size_t) p * d + i];
}

void wa_launch_dec_embed(hipStream_t stream, const int32_t * tok, const int32_t * pos, int n_tokens, int d, const wa_f16 * te,
                         const float * pe, float * x) {
    hipLaunchKernelGGL(k_dec_embed, dim3(n_tokens), dim3(256), 0, stream, tok, pos, n_tokens, d, te, pe, x);
}

// k_gemm_bf16.hip — bf16 x bf16 -> f32 MFMA GEMM with fused epilogue for gfx950.
//
//   out[M][N] = epilogue( A[M][K] . W[N][K]^T )        A, W row-major bf16 (K contiguous)
//
// This ONE kernel family carries every dense contraction of the encoder (SURVEY.md §8a rows S2-S4, L2, L3, L6
// pw1/pw2, D1): 96.8 % of the path's FLOPs.  There is a single kernel, gemm_smf16_kernel, instantiated for four
// tile heights (BM = 256 / 192 / 128 / 64 rows x 256 columns) and five epilogues (bf16, f32, f32 + residual, f32 +
// LayerNorm(residual) from per-row statistics, GLU).  Every instantiation computes an
// output element with the SAME arithmetic — v_mfma_f32_16x16x32_bf16 over ascending 32-deep k-steps into one f32
// accumulator, then bias / activation / alpha / residual in a fixed order — so a row's result does not depend on
// M, on the tile height the launcher picks, or on where the row sits in a tile: the encoder is batch-invariant
// (an utterance alone == the same utterance inside a batch of 256; tests/test_gpu_fullsize.py).  The families this
// file used to hold (32x32x16 tiles, a four-granule ring, a two-K-tile ring, a persistent grid) are in the history
// up to round 2 together with their measurements (profiles/r01*, r02*).
//
// Structure of gemm_smf16_kernel:
//   * 8 waves = 2 wave groups (wm) x 4 column slices (wn); a wave owns (BM/2) x 64 outputs.
//   * K tiles are 64 deep (one 128-byte cache line per row), fetched HBM/L2 -> LDS by global_load_lds_dwordx4 (no
//     VGPR round trip), 8 rows x 128 B per wave instruction, issued from inline asm; 16-byte chunks of an LDS row
//     are XOR-swizzled by (row >> 1) & 7 on the per-lane SOURCE address and again on the ds_read_b128 fragment
//     reads (conflict-free).
//   * LDS, tiles of 256 / 192 rows = five 32-KiB operand-part slots ("split ring"): a K tile is two parts (A rows,
//     weight rows), part p lives in slot p mod 5; during K tile t a wave issues its pieces of A(t+2), which have a whole
//     extra K tile to land, and its weight pieces of the NEXT K tile — wave group 0 in its phase-0 memory half, wave
//     group 1 (one barrier behind: it would have two phases for them to land instead of three) one phase earlier,
//     between the MFMAs of phase 1 of the K tile before.  Tiles of 128 / 64 rows = three whole K-tile stages: both
//     operands are issued two K tiles ahead.  Waits are COUNTED s_waitcnt vmcnt(n) across raw s_barriers (guide §5
//     T3+T4), never a drain.
//   * two consecutive tiles per workgroup (singles in the last round), the ring carried from the first into the second:
//     the last K tiles of tile 0 already issue tile 1's first K tiles, which land during tile 0's epilogue.
//   * the two wave groups run half a phase apart (ping-pong): one issues its MFMAs while the other does its
//     fragment reads and DMA issues.
//   * XCD-aware, grouped blockIdx -> tile order (each XCD, private L2, walks a contiguous run of tiles).
//   * epilogue through a per-wave 4-KiB LDS scratch (aliasing a dead slot / stage) so every global access is a whole 128 / 256-
//     byte row segment: +bias, ReLU / SiLU / GLU, *alpha, +residual (f32, prefetched), per-utterance row mask.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "rs_common.h"

namespace {

struct GemmParams {
    const uint16_t* A; const uint16_t* W; void* out;
    const float* bias; const float* residual; const int32_t* mask_lens;
    int lda, ldw, ldc, M, N, K, flags;
    float alpha;
    int mask_rows_per_step, mask_steps;
    int mask_row0;     // global row index of this launch's row 0 (launches chunked over M: see rs_launch_gemm)
    int tiles_m, tiles_n;
    int group_m;       // row panels per XCD tile group
    const float* res_ln_stats;   // OUT_RESLN: per row (mean, rstd) of the LayerNorm the residual operand still has to go through
    const float* res_ln_g;       //            its weight / bias [N]
    const float* res_ln_b;
    uint16_t* out2;    // OUT_RES2: bf16 copy of the result, row pitch ld2 elements
    int ld2;
    int pairs;         // > 0: a workgroup runs two consecutive tiles of its XCD's run (xcd_split); 2: the LDS ring carries over; 0: one tile
    long long* trace;  // debug: per-tile timestamps (scripts/gemm_trace.py); nullptr in production
    // CONVA instantiations (rs_gemm_args.conv_C > 0): byte strides of the patch rows in the channels-last input and the K-tile walk
    int cv_T2, cv_F2;                       // m = (b * cv_T2 + t2) * cv_F2 + f2
    unsigned cv_b_bytes, cv_t_bytes, cv_f_bytes;   // T1 * F1 * C * 2, 2 * F1 * C * 2, 2 * C * 2
    unsigned cv_seg_bytes;                  // F1 * C * 2: from a kernel row of the patch to the next
    int cv_tps, cv_inv;                     // K tiles per kernel row (3 * C / 64); ceil(65536 / cv_tps): k / cv_tps == (k * cv_inv) >> 16 (launcher checks)
    const uint16_t* Wfm;                    // BREG instantiations: the weight once more, fragment-major [N / 16][K / 32][64 lanes][8] (wfm_shuffle_kernel)
};

// OUT_BF16S / OUT_F32S: the plain bf16 / f32 epilogues with icefall's Swoosh activations compiled in (the Zipformer family).
// They are instantiations of their own: with the branches in the shared epilogue the FastConformer's ffn_up launches (256-row
// tiles, SiLU) ran 7 % slower (317.6 vs 295.5 us, profiles/r05c_bench.json against BENCH_r04.json).
// OUT_RES2: OUT_RES that also stores the result as bf16 (GemmParams.out2): the Zipformer layers' residual branches.
enum { OUT_BF16 = 0, OUT_F32 = 1, OUT_RES = 2, OUT_GLU = 3, OUT_RESLN = 4, OUT_BF16S = 5, OUT_F32S = 6, OUT_RES2 = 7 };

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

// The DMAs are issued from inline asm (global_load_lds_dwordx4, scalar base + 32-bit lane offset): hipcc then counts
// only the epilogue's own loads / stores, which are all younger than the DMAs in flight, so its counted waits stay
// correct and it never drains the ring (guide §5 "three .s-level traps" (b)).  M0 (the DMA's LDS base) is written and
// consumed inside one asm statement; nothing else in this kernel uses M0 (gfx950 DS instructions do not).
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// BREG instantiations: a weight fragment goes global -> VGPR (one contiguous 1-KiB run per wave instruction in the fragment-major
// copy), never through LDS.  Issued from inline asm like the DMAs so that the counted vmcnt waits below see ONE in-order queue;
// the compiler believes the register is written at once, so every consumer is fenced by breg_wait, which names the registers as
// read-write operands of the s_waitcnt.
__device__ __forceinline__ void gload16(u32x4_t& dst, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// ONE fence per use, after whichever counted wait ran: an asm per branch would give the registers one definition per branch and
// the merge makes the compiler copy them — at a point where the loads may still be in flight (seen in the ISA of the first
// version: v_mov_b64 of the set at the loop head, stale fragments in 192-row launches).
__device__ __forceinline__ void breg_fence(u32x4_t (&r)[4]) {
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// W bf16 [N][ldw] -> fragment-major [N / 16][K / 32][64][8]: fragment (nb, kb), lane l = W[16 nb + (l & 15)][32 kb + 8 (l >> 4) .. + 8]
// — what v_mfma_f32_16x16x32_bf16 takes as its weight operand.  One thread per 16-byte piece; N % 16 == 0, K % 32 == 0.
__global__ __launch_bounds__(256) void wfm_shuffle_kernel(const uint16_t* __restrict__ W, int ldw, int N, int K, uint16_t* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)N * K / 8;
    if (idx >= total) return;
    const int l = (int)(idx & 63);
    const size_t f = idx >> 6;
    const int kbs = K / 32;
    const int kb = (int)(f % kbs), nb = (int)(f / kbs);
    reinterpret_cast<uint4*>(out)[idx] = *reinterpret_cast<const uint4*>(W + (size_t)(nb * 16 + (l & 15)) * ldw + kb * 32 + (l >> 4) * 8);
}

__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }

// Work split of one XCD's run of `xcount` tiles over its workgroups.  A tile costs ~5 us on top of its own time when it is
// a workgroup of its own (launch of a 160-KiB-LDS, 512-thread group + the imbalance of a lockstep round:
// profiles/r03a_gemm_tile_sched_ab.txt), so a workgroup runs TWO consecutive tiles back to back — except in the last
// round, which is dealt as single tiles so that all 32 CUs of the XCD stay busy to the end:
//   xcount = 2 * pairs + singles;  singles = xcount mod 64 when that is at most 32 (one more, half-length round), else the
//   remainder is paired up too.  Slots [0, pairs) run tiles 2s, 2s+1; slots [pairs, pairs + singles) one tile each.
__host__ __device__ inline void xcd_split(int xcount, int pair_mode, int& n_pairs, int& n_singles) {
    if (!pair_mode) { n_pairs = 0; n_singles = xcount; return; }
    const int rem = xcount % 64;
    n_singles = rem <= 32 ? rem : (xcount & 1);
    n_pairs = (xcount - n_singles) / 2;
}

// ---- epilogue (no workgroup barrier: the scratch is per wave, LDS operations of one wave execute in order).
// bf16 output: chunks of 32 rows x 64 columns; f32: 16 rows x 64 columns; both 4 KiB, 16-byte pieces XOR-swizzled by
// row.  All global accesses are raw buffer operations (masked rows / columns get an out-of-range offset), so the
// number of VMEM operations a wave issues here is a compile-time constant.
// Lane layout of the accumulators: the weight fragment is the MFMA A operand, so D = (A.W^T)^T and block (i, j) holds
// m = .. + i*16 + (lane & 15), n = .. + j*16 + 4*(lane >> 4) + 0..3.
template <int MI, int NI, int OUT_KIND, bool MASK, int PF>
__device__ __forceinline__ void smf16_epilogue(const GemmParams& p, f32x4_t (&acc)[MI][NI], char* scr, int cm0, int cn0,
                                               int wm, int wn, int lane) {
    constexpr bool SWOOSH = OUT_KIND == OUT_BF16S || OUT_KIND == OUT_F32S, RES2 = OUT_KIND == OUT_RES2;
    constexpr int OUT = OUT_KIND == OUT_BF16S ? OUT_BF16 : (OUT_KIND == OUT_F32S ? OUT_F32 : (RES2 ? OUT_RES : OUT_KIND));
    // OUT_RESLN: the residual operand is y, the previous layer's output BEFORE its output LayerNorm; that norm is applied
    // here from per-row statistics (layernorm2_kernel writes them instead of the normalised f32 rows: one 145-MB write
    // per layer boundary less), with the arithmetic of the norm kernel: fma((y - mean) * rstd, g, b).
    constexpr bool RESLN = OUT == OUT_RESLN, RES = OUT == OUT_RES || RESLN, out_f32 = OUT == OUT_F32 || RES, GLU = OUT == OUT_GLU, rowmask = MASK;
    constexpr int TM = MI * 16, TN = NI * 16;
    constexpr unsigned OOB = 0xfffffff0u;                         // beyond every buffer: loads return 0, stores are dropped
    const int frow = lane & 15, fch = lane >> 4;
    const int flags = p.flags;
    const bool has_bias = flags & RS_GEMM_BIAS, relu = flags & RS_GEMM_RELU, silu = flags & RS_GEMM_SILU;
    const bool swl = SWOOSH && (flags & RS_GEMM_SWOOSHL), swr = SWOOSH && (flags & RS_GEMM_SWOOSHR);
    const float alpha = p.alpha;
    // GLU: the output has N / 2 columns (ldc is the caller's row pitch of that narrower matrix).  The launcher keeps
    // M * ldc * element size below 2^31 (it cuts a taller problem into row chunks).
    const size_t out_bytes = (size_t)p.M * p.ldc * (out_f32 ? 4 : 2);
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    const auto res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.residual : (const float*)p.out), 0,
                                                            (int)out_bytes, 0x00020000);
    const auto bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_bias ? p.bias : (const float*)p.out), 0,
                                                             has_bias ? p.N * 4 : 0, 0x00020000);
    const int wrow0 = cm0 + wm * TM, wcol0 = cn0 + wn * TN;
    float4 bias_r[NI];
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int n = wcol0 + jj * 16 + 4 * fch;
        const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(bias_rsrc, (unsigned)n * 4u, 0, 0);   // no bias / n >= N: zeros
        bias_r[jj] = __builtin_bit_cast(float4, b);
    }
    auto finish = [&](int i, int jj) -> float4 {
        float4 v = make_float4(acc[i][jj][0] + bias_r[jj].x, acc[i][jj][1] + bias_r[jj].y, acc[i][jj][2] + bias_r[jj].z,
                               acc[i][jj][3] + bias_r[jj].w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        if constexpr (SWOOSH) {
            if (swl) { v.x = swoosh_l_f(v.x); v.y = swoosh_l_f(v.y); v.z = swoosh_l_f(v.z); v.w = swoosh_l_f(v.w); }
            if (swr) { v.x = swoosh_r_f(v.x); v.y = swoosh_r_f(v.y); v.z = swoosh_r_f(v.z); v.w = swoosh_r_f(v.w); }
        }
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        return v;
    };
    auto row_keep = [&](int m) -> bool {
        if (!rowmask || m >= p.M) return true;      // rowmask is a template constant
        const int step = (m + p.mask_row0) / p.mask_rows_per_step;
        const int b = step / p.mask_steps;
        return step - b * p.mask_steps < p.mask_lens[b];
    };
    if constexpr (GLU) {
        // GLU pairs inside the wave (weight rows interleaved in blocks of 32 by the loader: columns [0, 32) of a
        // wave's 64 are the values, [32, 64) their gates, i.e. accumulator blocks jj and jj + NI / 2 of the SAME lane):
        // out[m][n / 2 ...] = bf16( (a + bias_a) * sigmoid(g + bias_g) ).  Chunks of 32 rows x 32 output columns
        // (2 KiB, 64-byte rows, 16-byte pieces XOR-swizzled by row); two 16-row x 64-byte stores per chunk.
        static_assert(NI == 4, "GLU epilogue: 64-column wave tiles");
        const int rr4 = lane >> 2, cc = lane & 3;
        const int ocol0 = wcol0 >> 1;
#pragma unroll
        for (int c = 0; c < MI / 2; ++c) {
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float4 a = finish(2 * c + il, jj), g = finish(2 * c + il, jj + 2);
                    const int row = il * 16 + frow;
                    const int piece = (jj * 2 + (fch >> 1)) ^ ((row >> 1) & 3);
                    *reinterpret_cast<u16x4_t*>(scr + row * 64 + piece * 16 + (fch & 1) * 8) =
                        pack_bf16x4(a.x * sigmoid_f(g.x), a.y * sigmoid_f(g.y), a.z * sigmoid_f(g.z), a.w * sigmoid_f(g.w));
                }
#pragma unroll
            for (int sgm = 0; sgm < 2; ++sgm) {
                const int row = sgm * 16 + rr4;
                u32x4_t d = *reinterpret_cast<const u32x4_t*>(scr + row * 64 + ((cc ^ ((row >> 1) & 3)) * 16));
                const int m = wrow0 + c * 32 + row, n = ocol0 + cc * 8;
                if (!row_keep(m)) d = (u32x4_t){0u, 0u, 0u, 0u};
                const unsigned off = (m < p.M && 2 * n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(d, out_rsrc, off, 0, 0);
            }
        }
    } else if constexpr (!out_f32) {
        // bf16: chunks of 32 rows x 64 columns (4 KiB, 128-byte rows, 16-byte pieces XOR-swizzled by row)
        const int rr8 = lane >> 3, cc = lane & 7;
#pragma unroll
        for (int c = 0; c < MI / 2; ++c) {
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int jj = 0; jj < NI; ++jj) {
                    const float4 v = finish(2 * c + il, jj);
                    const int row = il * 16 + frow;
                    const int piece = (jj * 2 + (fch >> 1)) ^ (row & 7);
                    *reinterpret_cast<u16x4_t*>(scr + row * 128 + piece * 16 + (fch & 1) * 8) = pack_bf16x4(v.x, v.y, v.z, v.w);
                }
#pragma unroll
            for (int sgm = 0; sgm < 4; ++sgm) {
                const int row = sgm * 8 + rr8;
                u32x4_t d = *reinterpret_cast<const u32x4_t*>(scr + row * 128 + ((cc ^ (row & 7)) * 16));
                const int m = wrow0 + c * 32 + row, n = wcol0 + cc * 8;
                if (!row_keep(m)) d = (u32x4_t){0u, 0u, 0u, 0u};
                const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(d, out_rsrc, off, 0, 0);
            }
        }
    } else {
        // f32: chunks of 16 rows x 64 columns (4 KiB, 256-byte rows, 16-byte pieces XOR-swizzled by row).
        // The residual rows (row-major, whole 256-byte segments) are requested PF chunks ahead: with one chunk of
        // lookahead every chunk paid a full HBM round trip (a dependent chain of MI latencies per wave); 3 ahead is the
        // whole-path optimum (profiles/r02u_bench_ab.txt: all of them ahead is a 24-load burst per wave).
        const int rr4 = lane >> 4, cc = lane & 15;
        constexpr int PFE = (RESLN && MI >= 8 && PF > 1) ? PF - 1 : PF;    // 256-row tiles: one chunk less ahead instead of spilling
        constexpr int NPF = RES ? (PFE < 1 ? 1 : (PFE > MI ? MI : PFE)) : 1;
        u32x4_t rvq[NPF][4];
        float2 stq[NPF][4];                                           // RESLN: (mean, rstd) of the rows of a chunk
        float4 ln_g = make_float4(0.f, 0.f, 0.f, 0.f), ln_b = ln_g;
        const auto st_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RESLN ? p.res_ln_stats : (const float*)p.out), 0,
                                                               RESLN ? p.M * 8 : 0, 0x00020000);
        if constexpr (RESLN) {
            const int n = wcol0 + cc * 4;
            if (n < p.N) {
                ln_g = *reinterpret_cast<const float4*>(p.res_ln_g + n);
                ln_b = *reinterpret_cast<const float4*>(p.res_ln_b + n);
            }
        }
        auto load_res = [&](int i, u32x4_t (&rv)[4], float2 (&st)[4]) {
#pragma unroll
            for (int sgm = 0; sgm < 4; ++sgm) {
                const int m = wrow0 + i * 16 + sgm * 4 + rr4, n = wcol0 + cc * 4;
                const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u : OOB;
                rv[sgm] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, off, 0, 0);
                if constexpr (RESLN)
                    st[sgm] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(st_rsrc, m < p.M ? (unsigned)m * 8u : OOB, 0, 0));
            }
        };
        if constexpr (RES) {
#pragma unroll
            for (int d = 0; d < NPF; ++d) load_res(d, rvq[d], stq[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            u32x4_t rv[4];
            float2 st[4];
            if constexpr (RES) {
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) { rv[sgm] = rvq[i % NPF][sgm]; st[sgm] = stq[i % NPF][sgm]; }
            }
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) {
                const float4 v = finish(i, jj);
                const int piece = (jj * 4 + fch) ^ frow;
                *reinterpret_cast<float4*>(scr + frow * 256 + piece * 16) = v;
            }
#pragma unroll
            for (int sgm = 0; sgm < 4; ++sgm) {
                const int row = sgm * 4 + rr4;
                float4 v = *reinterpret_cast<const float4*>(scr + row * 256 + ((cc ^ row) * 16));
                if constexpr (RES) {
                    float4 r = __builtin_bit_cast(float4, rv[sgm]);
                    if constexpr (RESLN) {
                        const float mean = st[sgm].x, rstd = st[sgm].y;
                        r.x = fmaf((r.x - mean) * rstd, ln_g.x, ln_b.x); r.y = fmaf((r.y - mean) * rstd, ln_g.y, ln_b.y);
                        r.z = fmaf((r.z - mean) * rstd, ln_g.z, ln_b.z); r.w = fmaf((r.w - mean) * rstd, ln_g.w, ln_b.w);
                    }
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                const int m = wrow0 + i * 16 + row, n = wcol0 + cc * 4;
                if (!row_keep(m)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), out_rsrc, off, 0, 0);
                if constexpr (RES2) {
                    if (m < p.M && n < p.N) *reinterpret_cast<u16x4_t*>(p.out2 + (size_t)m * p.ld2 + n) = pack_bf16x4(v.x, v.y, v.z, v.w);
                }
            }
            if constexpr (RES) {
                if (i + NPF < MI) load_res(i + NPF, rvq[i % NPF], stq[i % NPF]);   // refill the slot this chunk just consumed
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// =====================================================================================================
// The kernel.  Main-loop schedule: ping-pong wave groups, two phases per K tile, each [12 fragment reads | 32 MFMAs]
// (BM = 256) per 32-deep k-step:
//                             group 0:      mem(p) | B | mfma(p) | B | mem(p+1) | B | ...
//                             group 1:  B | mem(p) | B | mfma(p) | B | ...                       (one barrier late)
// RAW: K tile t+1 is waited for (each wave: counted vmcnt for its own pieces) in the LAST phase of K tile t — group 0 at
// the end of its MFMA half, group 1 in its memory half, i.e. in the same barrier interval — and first read one interval
// later, after a barrier that every wave's wait precedes.  WAR: the slots written during K tile t held K tile t-1; the
// first DMA of K tile t is issued in group 0's mem(t, 0), when group 1 sits in the MFMA half of K tile t-1's last phase:
// its reads of that phase retired (lgkmcnt(0)) before the barrier that opened the interval.
// Two other schedules were built and measured this round and removed again (profiles/r03a_gemm_tile_sched_ab.txt,
// r03a_bench_sched_ab.txt; code in the history at 613b95c): no ping-pong (one barrier per K tile) and four phases per K
// tile (16 MFMAs each, the guide's 8-phase shape).  Per shape they are within +-4 % of this one either way — the loop is
// paced by the LDS traffic of the 128 x 64 wave tile and the clock, not by its barrier structure — and on the whole
// path both lose ~1 % (58.8 vs 59.5 ms / step).
template <int BM, int OUT, bool MASK, int EPF, bool TRACE, bool CONVA = false, bool BREG = false>
__global__ __launch_bounds__(512, 2) void gemm_smf16_kernel(GemmParams p) {
    constexpr int BN = 256, WN = 4, NWAVES = 8;
    constexpr int TM = BM / 2, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int SLOT = 32768, NSLOT = 5;
    constexpr int LA = BM / 8 / NWAVES, LB = BN / 8 / NWAVES;     // DMA pieces (8 rows x 128 B) per wave and K tile
    constexpr int NPH = 2;                                        // phases per K tile
    static_assert((BM == 256 || BM == 192 || BM == 128 || BM == 64) && LB == 4 && (MI % 2) == 0 && BM * 128 <= SLOT, "tile heights");
    // R3 (tiles of <= 128 rows): the LDS is THREE whole K-tile stages (A BM x 128 B | B 32 KiB; 48 KiB at 128 rows) instead
    // of the five part slots: during K tile t a wave issues ALL of K tile t+2, so both operands have a whole K tile more
    // to land (the split ring gives that to A only; its B(t+1) must arrive within the K tile it was issued in, and the
    // short K tiles of a low tile sit on exactly that round trip: 128-row tiles ran 4 - 17 % slower on the split ring,
    // profiles/r03j_gemm_ring3_ab.txt).  Waits leave K tile t+2 (LA + LB pieces) in flight.  Taller tiles do not fit three
    // stages in 160 KiB.
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + SLOT;
    constexpr bool R3 = 3 * STAGE <= NSLOT * SLOT;
    static_assert(!BREG || (!R3 && !TRACE && NI == 4), "BREG: split-ring tiles (256 / 192 rows) only");
    long long tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_w0 = 0, tr_stall = 0;
    if constexpr (TRACE) { tr_t0 = __builtin_readcyclecounter(); tr_w0 = (long long)wall_clock64(); }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fch = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // XCD-aware, bijective tile order: workgroups b, b+8, b+16 .. run on one XCD (private L2); that XCD owns a
    // contiguous run of tiles.  Inside the run tiles are grouped: group_m row panels form a group that is walked
    // m-fastest, then across n, so the ~32 tiles an XCD runs at the same time are group_m A panels x 32/group_m
    // weight tiles (fewer distinct bytes per round).
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, xslot = bid >> 3;
    const int q8 = nwg >> 3, rr = nwg & 7;
    const int xbase = xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8;
    const int xcount = xcd < rr ? q8 + 1 : q8;
    int n_pairs, n_singles;
    xcd_split(xcount, p.pairs, n_pairs, n_singles);
    if (xslot >= n_pairs + n_singles) return;
    const int first_tile = xslot < n_pairs ? 2 * xslot : 2 * n_pairs + (xslot - n_pairs);
    const int n_my = xslot < n_pairs ? 2 : 1;
    const int nk = p.K / 64;                                      // K tiles, >= 1
    // The LDS ring keeps running from the first tile of a pair into the second when a tile has at least two K tiles
    // (p.pairs == 2): the last K tiles of tile 0 already issue the DMAs of tile 1's first K tiles, which land during
    // tile 0's epilogue, and tile 1 starts without a prologue.  Otherwise every tile restarts the ring (p.pairs == 1).
    const bool carry = p.pairs == 2 && nk >= 2;
    const int dr = lane >> 3, dpc = lane & 7;                     // lane = (row l >> 3 of a DMA piece, physical chunk l & 7)
    auto tile_origin = [&](int j, int& m0, int& n0) {
        const int wg = xbase + first_tile + j;
        const int per_group = p.group_m * p.tiles_n;
        const int g = wg / per_group, r = wg - g * per_group;
        const int left = p.tiles_m - g * p.group_m;
        const int gm = left < p.group_m ? left : p.group_m;
        const int tile_n = r / gm;
        m0 = (g * p.group_m + (r - tile_n * gm)) * BM;
        n0 = tile_n * BN;
    };
    // per-lane byte offsets of this wave's DMA pieces; rows past the matrix are clamped to its last row (their
    // products land in masked outputs)
    auto lane_offsets = [&](int m0, int n0, unsigned (&o)[LA + LB]) {
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int row = (wave + NWAVES * j) * 8 + dr;
            int gr = m0 + row;
            gr = gr < p.M ? gr : p.M - 1;
            if constexpr (CONVA) {                                // patch row of pixel (b, t2, f2) in the channels-last input
                const int bt = gr / p.cv_F2, f2 = gr - bt * p.cv_F2;
                const int b = bt / p.cv_T2, t2 = bt - b * p.cv_T2;
                o[j] = (unsigned)b * p.cv_b_bytes + (unsigned)t2 * p.cv_t_bytes + (unsigned)f2 * p.cv_f_bytes + (unsigned)((dpc ^ swz64(row)) * 16);
            } else {
                o[j] = (unsigned)gr * (unsigned)(p.lda * 2) + (unsigned)((dpc ^ swz64(row)) * 16);
            }
        }
        if constexpr (BREG) return;                              // the weights do not go through LDS
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int row = (wave * LB + j) * 8 + dr;
            int gr = n0 + row;
            gr = gr < p.N ? gr : p.N - 1;
            o[LA + j] = (unsigned)gr * (unsigned)(p.ldw * 2) + (unsigned)((dpc ^ swz64(row)) * 16);
        }
    };
    // BREG: two register sets of NI fragments, one per phase of a K tile.  Set ks is reloaded with the fragments of K tile t + 1
    // right after the MFMAs of phase (t, ks) were issued (their operand reads are long done when the data comes back), and waited
    // for in the memory half of phase (t + 1, ks): 1.5 phases of flight.  Queue order per K tile t and wave:
    //   A(t+2) x LA (between the MFMAs of phase 0) | B(t+1, 0) x NI | B(t+1, 1) x NI
    // so "B(t, 0) landed" = vmcnt(NI) (B(t, 1) younger), "B(t, 1) landed" = vmcnt(LA if A(t+2) was issued + NI if B(t+1, 0) was),
    // and A(t+1), issued during K tile t-1 BEFORE B(t, *), has landed whenever B(t, 1) has: the wait_next of the LDS form is implied.
    u32x4_t breg[2][4];
    const unsigned bvoff = (unsigned)lane * 16u;
    auto bfrag_base = [&](int n0_, int kb) -> const char* {       // fragment (column block of this wave, k-step kb), jj = 0
        const int nb = (n0_ >> 4) + wn * NI;
        const size_t off = ((size_t)nb * (size_t)(p.K >> 5) + (size_t)kb) << 10;
        return reinterpret_cast<const char*>(p.Wfm) + off;
    };
    auto bload = [&](int set, int n0_, int kb) {
        const char* b0 = bfrag_base(n0_, kb);
        const size_t jstride = (size_t)(p.K >> 5) << 10;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) gload16(breg[set][jj], bvoff, b0 + jj * jstride);
    };
    auto frag = [&](const char* part, int row, int chunk) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(part + row * 128 + ((chunk ^ swz64(row)) << 4));
    };
    auto wrap = [](int v) { return v >= NSLOT ? v - NSLOT : v; };

    int m0, n0;
    unsigned off[LA + LB], offn[LA + LB];
    tile_origin(0, m0, n0);
    lane_offsets(m0, n0, off);
    int sa = 0;                                                   // slot of A(t); B(t) = sa + 1, B(t+1) = sa + 3, A(t+2) = sa + 4 (mod 5)
  for (int it = 0; it < n_my; ++it) {
    const bool has_next = carry && it + 1 < n_my;                 // the ring runs on into another tile
    int m1 = m0, n1 = n0;
    if (has_next) { tile_origin(it + 1, m1, n1); lane_offsets(m1, n1, offn); }
    // piece j of K tile k (k >= nk: K tile k - nk of the NEXT tile) into slot sl
    auto dma_a = [&](int j, int k, int dst) {                     // dst: byte offset of the part in LDS
        const bool nx = k >= nk;
        const int kt = nx ? k - nk : k;
        size_t kbytes = (size_t)kt * 128;
        if constexpr (CONVA) {                                    // K tile kt = kernel row sg of the patch, tile kt - sg * tps inside it
            const int sg = (kt * p.cv_inv) >> 16;
            kbytes = (size_t)sg * p.cv_seg_bytes + (size_t)(kt - sg * p.cv_tps) * 128;
        }
        glds16(nx ? offn[j] : off[j], reinterpret_cast<const char*>(p.A) + kbytes, lds0 + dst + (wave + NWAVES * j) * 1024);
    };
    auto dma_b = [&](int j, int k, int dst) {
        if constexpr (BREG) return;
        const bool nx = k >= nk;
        glds16(nx ? offn[LA + j] : off[LA + j], reinterpret_cast<const char*>(p.W) + (size_t)(nx ? k - nk : k) * 128,
               lds0 + dst + (wave * LB + j) * 1024);
    };
    if constexpr (BREG) {
        if (it == 0 || !carry) {
            // prologue: A(0) -> slot 0, A(1) -> slot 2, then the weight fragments of K tile 0 into the two register sets;
            // A(0) has landed when at most A(1) and the fragments are still in flight
            sa = 0;
#pragma unroll
            for (int j = 0; j < LA; ++j) dma_a(j, 0, 0);
            if (nk > 1) {
#pragma unroll
                for (int j = 0; j < LA; ++j) dma_a(j, 1, 2 * SLOT);
            }
            bload(0, n0, 0);
            bload(1, n0, 1);
            if (nk > 1) wait_vmcnt_imm<LA + 2 * NI>();
            else wait_vmcnt_imm<2 * NI>();
            __builtin_amdgcn_s_barrier();
        } else {
            // second tile of a pair with the A ring carried over: A(0), A(1) were issued by the previous tile's last K tiles; the
            // weight fragments are NOT held across the epilogue (registers), they are asked for here
            bload(0, n0, 0);
            bload(1, n0, 1);
        }
    } else
    if (it == 0 || !carry) {
        // prologue: A(0) -> slot 0, B(0) -> slot 1, A(1) -> slot 2; K tile 0 is complete when all but the last LA landed
        // (R3: K tiles 0 and 1 whole -> stages 0 and 1; all but the last LA + LB)
        sa = 0;
#pragma unroll
        for (int j = 0; j < LA; ++j) dma_a(j, 0, 0);
#pragma unroll
        for (int j = 0; j < LB; ++j) dma_b(j, 0, R3 ? A_BYTES : SLOT);
        if (nk > 1) {
#pragma unroll
            for (int j = 0; j < LA; ++j) dma_a(j, 1, R3 ? STAGE : 2 * SLOT);
            if constexpr (R3) {
#pragma unroll
                for (int j = 0; j < LB; ++j) dma_b(j, 1, STAGE + A_BYTES);
                wait_vmcnt<LA + LB>();
            } else if (wm == 1) {                                 // group 1 issues its B pieces one phase early (see the K loop)
#pragma unroll
                for (int j = 0; j < LB; ++j) dma_b(j, 1, 3 * SLOT);
                wait_vmcnt<LA + LB>();
            } else {
                wait_vmcnt<LA>();
            }
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one barrier behind from here on
    if constexpr (TRACE) tr_t1 = __builtin_readcyclecounter();

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.0f;

    for (int t = 0; t < nk; ++t) {
        // split ring: sa = slot of A(t); B(t) = sa + 1; this K tile issues B(t+1) -> sa + 3 and A(t+2) -> sa + 4 (mod 5).
        // R3: sa = stage of K tile t (mod 3); this K tile issues B(t+2), A(t+2) -> stage sa + 2.
        auto wrap3 = [](int v) { return v >= 3 ? v - 3 : v; };
        const int a_cur = R3 ? sa * STAGE : sa * SLOT;
        const int b_cur = R3 ? a_cur + A_BYTES : wrap(sa + 1) * SLOT;
        const int bdst = R3 ? wrap3(sa + 2) * STAGE + A_BYTES : wrap(sa + 3) * SLOT;
        const int adst = R3 ? wrap3(sa + 2) * STAGE : wrap(sa + 4) * SLOT;
        constexpr int KB = R3 ? 2 : 1;                            // how many K tiles ahead the B pieces are issued
        const char* at = smem + a_cur;
        const char* bt = smem + b_cur;
        const bool has_1 = t + 1 < nk || has_next;                // a K tile follows (in this tile or the next)
        const bool has_b = t + KB < nk || has_next, has_a = t + 2 < nk || has_next;
        auto wait_next = [&]() {                                  // K tile t + 1 landed; what was issued for t + 2 may stay in flight
            if constexpr (R3) {
                if (has_a) wait_vmcnt<LA + LB>();
                else wait_vmcnt<0>();
            } else {
                if (has_a) wait_vmcnt<LA>();
                else wait_vmcnt<0>();
            }
        };
        auto timed_wait_next = [&]() {
            if constexpr (TRACE) { const long long a = __builtin_readcyclecounter(); wait_next(); tr_stall += __builtin_readcyclecounter() - a; }
            else wait_next();
        };
#pragma unroll
        for (int ks = 0; ks < NPH; ++ks) {                        // phase = one 32-deep k-step
            const bool last = ks == NPH - 1;
            bf16x8_t bfr[NI], af[MI];
            const bool has_b1 = t + 1 < nk;                       // BREG: K tile t + 1 of THIS tile exists (its fragments are reloaded below)
            if constexpr (!BREG) {
#pragma unroll
                for (int jj = 0; jj < NI; ++jj) bfr[jj] = frag(bt, wn * TN + jj * 16 + frow, ks * 4 + fch);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = frag(at, wm * TM + i * 16 + frow, ks * 4 + fch);
            if constexpr (BREG) {
                // the fragments of this phase (see the queue order at `breg`)
                if (ks == 0) wait_vmcnt_imm<NI>();
                else if (has_a && has_b1) wait_vmcnt_imm<LA + NI>();
                else if (has_a) wait_vmcnt_imm<LA>();
                else if (has_b1) wait_vmcnt_imm<NI>();
                else wait_vmcnt_imm<0>();
                breg_fence(breg[ks]);
#pragma unroll
                for (int jj = 0; jj < NI; ++jj) bfr[jj] = __builtin_bit_cast(bf16x8_t, breg[ks][jj]);
            }
            // Split ring: wave group 0 issues its B pieces of K tile t+1 here, beside its fragment reads — the slot (the one
            // A(t-1) lived in) is free from the barrier that opened this phase.  Group 1 runs one barrier behind and must
            // have its pieces landed by the end of ITS phase-1 memory half (group 0 reads K tile t+1 right after that
            // barrier): issued here they would have two phases to land (measured: 15 % of a 192-row K = 4096 main loop spent
            // in that wait, profiles/r03l_gemm_trace.txt), so group 1 issues its B pieces of K tile t+2 a phase EARLIER,
            // between the MFMAs of phase 1 of K tile t, into the slot of A(t), which nobody reads any more by then
            // (profiles/r03m_gemm_early_b_ab.txt, r03m_gemm_trace.txt: that wait 11 -> 2 us per ffn_down tile, 192-row
            // launches -1 .. -8 %, 256-row launches unchanged; the shader clock drops with the stall gone — power cap).
            if (ks == 0 && has_b && (R3 || wm == 0)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < LB; ++j) dma_b(j, t + KB, bdst);
            }
            if constexpr (!BREG)
            if (last && has_1 && wm == 1) timed_wait_next();      // group 1: one barrier behind, waits in its memory half
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int jj = 0; jj < NI; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj], af[i], acc[i][jj], 0, 0, 0);
                if (ks == 0 && i < LA && has_a) {                 // the A pieces of K tile t+2 between the MFMAs of phase 0
                    __builtin_amdgcn_sched_barrier(0);
                    dma_a(i, t + 2, adst);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!R3) {
                    if (ks == 1 && i < LB && wm == 1 && has_a) { // group 1: its B pieces of K tile t+2 (see above)
                        __builtin_amdgcn_sched_barrier(0);
                        dma_b(i, t + 2, a_cur);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (BREG) {
                if (has_b1) {                                     // this set's fragments of K tile t + 1 (the MFMAs above have read it)
                    __builtin_amdgcn_sched_barrier(0);
                    bload(ks, n0, 2 * (t + 1) + ks);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
            if (last && has_1 && wm == 0) timed_wait_next();      // group 0: before the barrier its reads follow
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        sa = R3 ? wrap3(sa + 1) : wrap(sa + 2);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                   // pairs with group 1's extra barrier: nobody reads this tile's parts any more
    {
        int em0 = __builtin_amdgcn_readfirstlane(m0), en0 = __builtin_amdgcn_readfirstlane(n0);
        asm volatile("" : "+s"(em0), "+s"(en0));                  // keep the addresses out of the main loop's live ranges
        if constexpr (TRACE) tr_t2 = __builtin_readcyclecounter();
        // Epilogue scratch: 4 KiB per wave in the slot that held this tile's LAST B part.  With the ring carried on,
        // slots sa .. sa+2 hold the next tile's A(0), B(0), A(1) and sa+3 (= last A) already receives group 1's pieces
        // of the next tile's B(1); sa+4 (= last B) is dead until the next tile's first K tile issues A(2) into it, which
        // happens after the barrier below.
        // (R3: stages sa, sa+1 hold the next tile's K tiles 0 and 1; sa+2 — the last K tile — is dead.)
        char* scr = smem + (R3 ? (sa + 2 >= 3 ? sa - 1 : sa + 2) * STAGE : wrap(sa + 4) * SLOT) + wave * 4096;
        smf16_epilogue<MI, NI, OUT, MASK, EPF>(p, acc, scr, em0, en0, wm, wn, lane);
        if constexpr (TRACE) {
            // wave 0 (group 0) and wave 4 (group 1) each write a record: [0] prologue, [1] main loop, [2] epilogue
            // issue, [3] store drain, [4] cycles stalled in the K-tile waits, [5] block, [6] / [7] wall clock
            const long long t3 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t4 = __builtin_readcyclecounter();
            if (lane == 0 && wn == 0) {
                long long* tr = p.trace + ((size_t)(xbase + first_tile + it) * 2 + wm) * 8;
                tr[0] = tr_t1 - tr_t0; tr[1] = tr_t2 - tr_t1; tr[2] = t3 - tr_t2; tr[3] = t4 - t3; tr[4] = tr_stall;
                tr[5] = bid; tr[6] = tr_w0; tr[7] = (long long)wall_clock64();   // 100 MHz, chip-wide
            }
            tr_t0 = __builtin_readcyclecounter(); tr_w0 = (long long)wall_clock64(); tr_stall = 0;
        }
    }
    if (it + 1 < n_my) {
        // second tile of the pair: every wave is done with this tile's parts and with its epilogue scratch before
        // anybody's next DMA lands there.  The epilogue's loads / stores stay in flight: they are younger than the carried
        // DMAs and older than the ones issued next, vmcnt retires in order, so the counted waits of the next tile cover them.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (has_next) {
            m0 = m1; n0 = n1;
#pragma unroll
            for (int j = 0; j < LA + LB; ++j) off[j] = offn[j];
        } else {
            tile_origin(it + 1, m0, n0);
            lane_offsets(m0, n0, off);
        }
    }
  }
}

// Process-wide A/B knobs (debug / tuning only; the defaults are the measured winners and nothing in the product path
// writes them).  Atomics initialised once from the environment, so concurrent first launches from the encoder thread
// and the decode workers are safe; they select code paths, not state, which is why they are not per context.
std::atomic<long long*> g_trace{nullptr};
std::atomic<int> g_tile{0};        // forced tile height (RS_GEMM_TILE / rs_debug_set_gemm_tile); 0 = by shape
std::atomic<int> g_group_m{0};     // row panels per XCD tile group; 0 = by shape
std::atomic<int> g_pairs{2};       // RS_GEMM_PAIRS: 2 = two tiles per workgroup with the LDS ring carried from the first into the second,
                                   // 1 = two tiles, the ring restarts, 0 = one tile per workgroup
std::atomic<int> g_breg{0};        // RS_GEMM_BREG: 1 = 256- / 192-row launches keep the weight operand out of LDS (fragment-major copy, global -> VGPR)
void gemm_knobs_from_env() {
    static std::once_flag once;
    std::call_once(once, [] {
        auto env = [](const char* name, std::atomic<int>& v) { if (const char* e = getenv(name)) v = atoi(e); };
        env("RS_GEMM_TILE", g_tile);
        env("RS_GEMM_GROUP_M", g_group_m);
        env("RS_GEMM_PAIRS", g_pairs);
        env("RS_GEMM_BREG", g_breg);
    });
}

// The fragment-major copy of a weight operand.  A registered tensor (rs_set_tensor: immutable for the life of the context) is
// shuffled once and cached; anything else (tests, micro-benchmarks) is shuffled into a scratch on every launch.
int wfm_operand(rs_ctx* ctx, const uint16_t* W, int ldw, int N, int K, hipStream_t s, const uint16_t** out) {
    const size_t bytes = (size_t)N * K * 2;
    auto it = ctx->wfm.find(W);
    if (it != ctx->wfm.end()) { *out = reinterpret_cast<const uint16_t*>(it->second); return RS_OK; }
    bool registered = false;
    for (const auto& kv : ctx->tensors)
        if (kv.second.first == (const void*)W && kv.second.second >= (size_t)N * ldw * 2) { registered = true; break; }
    void* dst = nullptr;
    if (registered) {
        if (hipMalloc(&dst, bytes) != hipSuccess) return rs_fail(ctx, RS_EHIP, "gemm: no memory for the fragment-major copy of a weight (%zu bytes)", bytes);
        ctx->wfm[W] = dst;
    } else {
        if (ctx->wfm_scratch_bytes < bytes) {
            if (ctx->wfm_scratch) { (void)hipStreamSynchronize(s); (void)hipFree(ctx->wfm_scratch); ctx->wfm_scratch = nullptr; ctx->wfm_scratch_bytes = 0; }
            if (hipMalloc(&ctx->wfm_scratch, bytes) != hipSuccess) return rs_fail(ctx, RS_EHIP, "gemm: no memory for the fragment-major scratch (%zu bytes)", bytes);
            ctx->wfm_scratch_bytes = bytes;
        }
        dst = ctx->wfm_scratch;
    }
    const size_t pieces = (size_t)N * K / 8;
    hipLaunchKernelGGL(wfm_shuffle_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, W, ldw, N, K, reinterpret_cast<uint16_t*>(dst));
    *out = reinterpret_cast<const uint16_t*>(dst);
    return RS_OK;
}

template <int BM>
int launch_smf16(rs_ctx* ctx, GemmParams& p, hipStream_t s) {
    constexpr int LDS = 5 * 32768;
    constexpr int EPF = 3;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + 255) / 256;
    const int ntiles = p.tiles_m * p.tiles_n;
    p.pairs = p.trace ? 0 : g_pairs.load();
    int nwg;                                                      // 8 x the workgroups of the fullest XCD run (the others exit at once)
    {
        int np, ns;
        xcd_split((ntiles + 7) / 8, p.pairs, np, ns);
        nwg = 8 * (np + ns);
        if (ntiles % 8 && ntiles > 8) {                           // runs of q and q + 1 tiles: size the grid for the larger need
            int np2, ns2;
            xcd_split(ntiles / 8, p.pairs, np2, ns2);
            if (8 * (np2 + ns2) > nwg) nwg = 8 * (np2 + ns2);
        }
    }
    // row panels per XCD tile group (profiles/r02r_gemm_group_m_sweep.txt): N = 1024 (4 weight tiles) likes 2 panels at
    // K = 4096 and 6 below; one-tile-wide problems (the subsampling GEMMs) 16; everything else is flat from 6 up
    p.group_m = g_group_m.load() > 0 ? g_group_m.load()
              : (p.tiles_n == 4 ? (p.K >= 4096 ? 2 : 6) : (p.K >= 4096 ? 4 : (p.tiles_n <= 8 && p.K <= 2560 ? 16 : 8)));
    const bool swoosh = p.flags & (RS_GEMM_SWOOSHL | RS_GEMM_SWOOSHR);
    const int out = (p.flags & RS_GEMM_RESIDUAL) ? (p.res_ln_stats ? OUT_RESLN : (p.out2 ? OUT_RES2 : OUT_RES))
                  : ((p.flags & RS_GEMM_OUT_F32) ? (swoosh ? OUT_F32S : OUT_F32) : ((p.flags & RS_GEMM_GLU) ? OUT_GLU : (swoosh ? OUT_BF16S : OUT_BF16)));
    const bool mask = p.flags & RS_GEMM_ROWMASK;
    if (p.cv_tps > 0) {                                           // patches read in place (launcher: plain bf16 output + row mask)
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_smf16_kernel<BM, OUT_BF16, true, EPF, false, true>, LDS); rc != RS_OK) return rc;
        hipLaunchKernelGGL((gemm_smf16_kernel<BM, OUT_BF16, true, EPF, false, true>), dim3(nwg), dim3(512), LDS, s, p);
        return RS_OK;
    }
    // register-resident weights ($RS_GEMM_BREG): the tall split-ring tiles of the FastConformer's epilogues, whole 64-column wave tiles
    if constexpr (BM >= 192) {
        if (g_breg.load() > 0 && !p.trace && !mask && p.N % 64 == 0 && p.N >= 256 &&
            (out == OUT_BF16 || out == OUT_RES || out == OUT_RESLN || out == OUT_F32 || out == OUT_GLU)) {
            if (int rc = wfm_operand(ctx, p.W, p.ldw, p.N, p.K, s, &p.Wfm); rc != RS_OK) return rc;
#define RS_SMF_B(O)                                                                                                \
            do {                                                                                                   \
                if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_smf16_kernel<BM, O, false, EPF, false, false, true>, LDS); rc != RS_OK) return rc; \
                hipLaunchKernelGGL((gemm_smf16_kernel<BM, O, false, EPF, false, false, true>), dim3(nwg), dim3(512), LDS, s, p); \
            } while (0)
            if (out == OUT_BF16) RS_SMF_B(OUT_BF16);
            else if (out == OUT_RES) RS_SMF_B(OUT_RES);
            else if (out == OUT_RESLN) RS_SMF_B(OUT_RESLN);
            else if (out == OUT_F32) RS_SMF_B(OUT_F32);
            else RS_SMF_B(OUT_GLU);
#undef RS_SMF_B
            return RS_OK;
        }
    }
#define RS_SMF(O, MK, TR)                                                                                          \
    do {                                                                                                           \
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_smf16_kernel<BM, O, MK, EPF, TR>, LDS); rc != RS_OK) return rc; \
        hipLaunchKernelGGL((gemm_smf16_kernel<BM, O, MK, EPF, TR>), dim3(nwg), dim3(512), LDS, s, p);  \
    } while (0)
    if (p.trace) {
        if constexpr (BM >= 192) {
            if (out == OUT_RES && !mask) RS_SMF(OUT_RES, false, true);
            else if (out == OUT_BF16 && !mask) RS_SMF(OUT_BF16, false, true);
            else return rs_fail(ctx, RS_EINVAL, "gemm trace: plain bf16 or residual output only");
            return RS_OK;
        } else {
            return rs_fail(ctx, RS_EINVAL, "gemm trace: 256- / 192-row tiles only");
        }
    }
    if (out == OUT_RES && !mask) RS_SMF(OUT_RES, false, false);
    else if (out == OUT_RESLN && !mask) RS_SMF(OUT_RESLN, false, false);
    else if (out == OUT_F32 && !mask) RS_SMF(OUT_F32, false, false);
    else if (out == OUT_BF16 && !mask) RS_SMF(OUT_BF16, false, false);
    else if (out == OUT_BF16 && mask) RS_SMF(OUT_BF16, true, false);
    else if (out == OUT_GLU && !mask) RS_SMF(OUT_GLU, false, false);
    else if (out == OUT_RES2 && !mask) RS_SMF(OUT_RES2, false, false);
    else if (out == OUT_BF16S && !mask) RS_SMF(OUT_BF16S, false, false);
    else if (out == OUT_F32S && !mask) RS_SMF(OUT_F32S, false, false);
    else return rs_fail(ctx, RS_EINVAL, "gemm: the row mask combines with plain bf16 output only");
#undef RS_SMF
    return RS_OK;
}

// Tile height for a problem.  All tiles of a launch cost the same and the CUs run them in lockstep rounds (one 160-KiB-
// LDS workgroup per CU), so the height minimises  rounds x (time of one tile),  with the tile time measured on MI355X
// (profiles/r03a_gemm_tile_sched_ab.txt, r03a_gemm_b32_tiles.txt; launch time / rounds):
//     T(BM) = nk * kt[BM] + fix[BM] (+ the f32 epilogue: 13 us with a residual read-modify-write, half without)
// kt = K-tile time at the power-capped clock of a busy chip — 256 rows are MFMA-paced, shorter tiles sit on the DMA
// round trip of a K tile —, fix = workgroup launch + prologue + epilogue + round imbalance.  The f32 residual epilogue
// moves 2 x BM x 256 x 4 bytes per tile from all CUs at once: it runs at the HBM rate (13 us per round of 256 tiles).
// At the benchmark batch this gives 256 rows to ffn_up / qkv, 192 rows to pw1 and the N = 1024 residual family; at
// B = 8 / 32 / 64 / 128 it reproduces the measured optimum of every encoder shape (profiles/r03k_gemm_tiles_b*.txt, taken
// with the 128- and 64-row tiles on the three-stage ring).  The choice never changes a result (see the header).
int pick_tile_height(int M, int N, int K, int n_cus, int flags) {
    static const int bms[4] = {256, 192, 128, 64};
    static const double kt[4] = {1.246, 1.146, 1.03, 0.96}, fix[4] = {11.5, 7.0, 5.0, 4.0};
    const long tn = (N + 255) / 256;
    const int nk = K / 64;
    const double epi = (flags & RS_GEMM_RESIDUAL) ? 13.0 : ((flags & RS_GEMM_OUT_F32) ? 6.5 : 0.0);
    int best = 256;
    double best_cost = 1e30;
    for (int i = 0; i < 4; ++i) {
        const long tiles = (long)((M + bms[i] - 1) / bms[i]) * tn;
        const long rounds = (tiles + n_cus - 1) / n_cus;
        const double cost = (double)rounds * (nk * kt[i] + fix[i] + epi * bms[i] / 192.0);
        if (cost < best_cost) { best_cost = cost; best = bms[i]; }
    }
    return best;
}

}  // namespace

// tuning hooks for A/B runs (scripts/gemm_bench.py, tests); not part of the public header
extern "C" void rs_debug_set_gemm_tile(int bm) { gemm_knobs_from_env(); g_tile = bm; }
extern "C" void rs_debug_set_gemm_trace(long long* buf) { g_trace = buf; }
extern "C" void rs_debug_set_gemm_group_m(int v) { gemm_knobs_from_env(); g_group_m = v; }
extern "C" void rs_debug_set_gemm_pairs(int v) { gemm_knobs_from_env(); g_pairs = v; }
extern "C" void rs_debug_set_gemm_breg(int v) { gemm_knobs_from_env(); g_breg = v; }
extern "C" int rs_debug_gemm_tile_height(int M, int N, int K, int n_cus, int flags) { return pick_tile_height(M, N, K, n_cus > 0 ? n_cus : 256, flags); }

static int launch_rows(rs_ctx* ctx, GemmParams& p, int bm, hipStream_t s) {
    switch (bm) {
        case 256: return launch_smf16<256>(ctx, p, s);
        case 192: return launch_smf16<192>(ctx, p, s);
        case 128: return launch_smf16<128>(ctx, p, s);
        case 64: return launch_smf16<64>(ctx, p, s);
        default: return rs_fail(ctx, RS_EINVAL, "gemm: tile height %d (256, 192, 128 or 64)", bm);
    }
}

int rs_launch_gemm(rs_ctx* ctx, const rs_gemm_args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return rs_fail(ctx, RS_EINVAL, "gemm: empty shape %d %d %d", a.M, a.N, a.K);
    if (a.K % 64) return rs_fail(ctx, RS_EINVAL, "gemm: K=%d must be a multiple of 64", a.K);
    const bool f32 = a.flags & (RS_GEMM_OUT_F32 | RS_GEMM_RESIDUAL);
    if (a.N % 4 || a.ldc % 4) return rs_fail(ctx, RS_EINVAL, "gemm: N=%d and ldc=%d must be multiples of 4", a.N, a.ldc);
    if (!f32 && ((a.ldc % 8) || (a.N % 8))) return rs_fail(ctx, RS_EINVAL, "gemm: bf16 output needs N %% 8 == 0 and ldc %% 8 == 0 (got %d, %d)", a.N, a.ldc);
    if ((a.lda % 8) || (a.ldw % 8) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) || ((uintptr_t)a.out & 15))
        return rs_fail(ctx, RS_EINVAL, "gemm: operands must be 16-byte aligned (lda %d ldw %d)", a.lda, a.ldw);
    if ((a.flags & RS_GEMM_ROWMASK) && (!a.mask_lens || a.mask_rows_per_step <= 0 || a.mask_steps <= 0))
        return rs_fail(ctx, RS_EINVAL, "gemm: row mask requested without lens");
    if ((a.flags & RS_GEMM_ROWMASK) && f32) return rs_fail(ctx, RS_EINVAL, "gemm: the row mask combines with bf16 output only");
    if ((a.flags & RS_GEMM_BIAS) && (!a.bias || ((uintptr_t)a.bias & 15)))
        return rs_fail(ctx, RS_EINVAL, "gemm: bias flag without a 16-byte aligned pointer");
    if ((a.flags & RS_GEMM_RESIDUAL) && (!a.residual || ((uintptr_t)a.residual & 15)))
        return rs_fail(ctx, RS_EINVAL, "gemm: residual flag without a 16-byte aligned pointer");
    if (a.res_ln_stats && (!(a.flags & RS_GEMM_RESIDUAL) || !a.res_ln_g || !a.res_ln_b || ((uintptr_t)a.res_ln_g & 15) ||
                           ((uintptr_t)a.res_ln_b & 15) || ((uintptr_t)a.res_ln_stats & 7)))
        return rs_fail(ctx, RS_EINVAL, "gemm: a normalised residual needs the residual flag, row statistics and aligned weight / bias");
    if (a.flags & RS_GEMM_GLU) {
        if (a.flags & (RS_GEMM_RELU | RS_GEMM_SILU | RS_GEMM_RESIDUAL | RS_GEMM_OUT_F32 | RS_GEMM_ROWMASK | RS_GEMM_SWOOSHL | RS_GEMM_SWOOSHR))
            return rs_fail(ctx, RS_EINVAL, "gemm: GLU combines with a bias only");
        if ((a.N % 64) || a.alpha != 1.0f) return rs_fail(ctx, RS_EINVAL, "gemm: GLU needs N %% 64 == 0 and alpha == 1 (N=%d)", a.N);
    }
    if (a.out_bf16 && (!(a.flags & RS_GEMM_RESIDUAL) || a.res_ln_stats || (a.ld_bf16 % 4) || ((uintptr_t)a.out_bf16 & 7)))
        return rs_fail(ctx, RS_EINVAL, "gemm: the bf16 copy combines with a plain residual output only (8-byte aligned, pitch %% 4)");
    if ((a.flags & (RS_GEMM_SWOOSHL | RS_GEMM_SWOOSHR)) && (a.flags & RS_GEMM_RESIDUAL))
        return rs_fail(ctx, RS_EINVAL, "gemm: the Swoosh activations combine with plain bf16 / f32 output only");
    if ((size_t)a.N * a.ldw * 2 >= (1ull << 32)) return rs_fail(ctx, RS_EINVAL, "gemm: weight matrix beyond 4 GiB");
    int cv_tps = 0, cv_inv = 0;
    if (a.conv_C > 0) {
        const int C = a.conv_C;
        if (C % 64 || a.K != 9 * C || a.conv_T2 <= 0 || a.conv_F2 <= 0 || a.M % (a.conv_T2 * a.conv_F2) || 2 * (a.conv_T2 - 1) + 3 > a.conv_T1 ||
            2 * (a.conv_F2 - 1) + 3 > a.conv_F1)
            return rs_fail(ctx, RS_EINVAL, "gemm: convolution patches need C %% 64 == 0, K == 9 C and an input that covers the output (C %d K %d)", C, a.K);
        if ((a.flags & ~(RS_GEMM_BIAS | RS_GEMM_RELU | RS_GEMM_SILU)) != RS_GEMM_ROWMASK)
            return rs_fail(ctx, RS_EINVAL, "gemm: convolution patches combine with plain bf16 output and the row mask only");
        if ((size_t)(a.M / (a.conv_T2 * a.conv_F2)) * a.conv_T1 * a.conv_F1 * C * 2 >= (1ull << 32) - 65536)
            return rs_fail(ctx, RS_EINVAL, "gemm: convolution input beyond 4 GiB (run it in chunks of utterances)");
        cv_tps = 3 * C / 64;
        cv_inv = (65536 + cv_tps - 1) / cv_tps;
        for (int k = 0; k < 2 * (a.K / 64); ++k)
            if (((k * cv_inv) >> 16) != k / cv_tps) return rs_fail(ctx, RS_EINVAL, "gemm: K-tile reciprocal is not exact for C = %d", C);
    }
    gemm_knobs_from_env();
    if (ctx->n_cus <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
        ctx->n_cus = n;
    }
    const int bm = g_tile.load() > 0 ? g_tile.load() : pick_tile_height(a.M, a.N, a.K, ctx->n_cus, a.flags);
    if (bm != 256 && bm != 192 && bm != 128 && bm != 64) return rs_fail(ctx, RS_EINVAL, "gemm: RS_GEMM_TILE=%d (256, 192, 128 or 64)", bm);
    // the kernel addresses A and the output with 32-bit byte offsets: a taller problem runs as row chunks
    const size_t out_row = (size_t)a.ldc * (f32 ? 4 : 2), a_row = a.conv_C > 0 ? 2 : (size_t)a.lda * 2;   // (patches: the input size was checked above)
    size_t max_rows = ((1ull << 31) - 65536) / out_row;
    if (((1ull << 32) - 65536) / a_row < max_rows) max_rows = ((1ull << 32) - 65536) / a_row;
    if (a.conv_C > 0 && (size_t)a.M > max_rows / 768 * 768) return rs_fail(ctx, RS_EINVAL, "gemm: convolution output beyond 2 GiB (run it in chunks of utterances)");
    if (max_rows < 768) return rs_fail(ctx, RS_EINVAL, "gemm: row pitch too large (lda %d, ldc %d)", a.lda, a.ldc);
    max_rows = max_rows / 768 * 768;                              // whole tiles of every height
    const double flops = 2.0 * a.M * (double)a.N * a.K;
    const double bytes = 2.0 * ((double)a.M * a.K + (double)a.N * a.K) +
                         (double)a.M * a.N * ((a.flags & RS_GEMM_OUT_F32) ? 4 : ((a.flags & RS_GEMM_GLU) ? 1 : 2)) +
                         ((a.flags & RS_GEMM_RESIDUAL) ? (double)a.M * a.N * 4 : 0.0);   // residual is read once
    ctx->prof_tag[0] = a.M; ctx->prof_tag[1] = a.N; ctx->prof_tag[2] = a.K; ctx->prof_tag[3] = a.flags;
    rs_prof_begin(ctx, RS_PROF_GEMM, s, flops, bytes);
    int rc = RS_OK;
    for (size_t r0 = 0; r0 < (size_t)a.M && rc == RS_OK; r0 += max_rows) {
        GemmParams p;
        const size_t rows = (size_t)a.M - r0 < max_rows ? (size_t)a.M - r0 : max_rows;
        p.A = a.A + r0 * a.lda; p.W = a.W;
        p.out = reinterpret_cast<char*>(a.out) + r0 * out_row;
        p.bias = a.bias; p.residual = a.residual ? a.residual + r0 * a.ldc : nullptr; p.mask_lens = a.mask_lens;
        p.res_ln_stats = a.res_ln_stats ? a.res_ln_stats + r0 * 2 : nullptr; p.res_ln_g = a.res_ln_g; p.res_ln_b = a.res_ln_b;
        p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.M = (int)rows; p.N = a.N; p.K = a.K; p.flags = a.flags;
        p.alpha = a.alpha; p.mask_rows_per_step = a.mask_rows_per_step; p.mask_steps = a.mask_steps; p.mask_row0 = (int)r0;
        p.out2 = a.out_bf16 ? a.out_bf16 + r0 * a.ld_bf16 : nullptr; p.ld2 = a.ld_bf16;
        p.tiles_m = p.tiles_n = 0; p.group_m = 1;
        p.trace = g_trace.load();
        p.cv_tps = cv_tps; p.cv_inv = cv_inv; p.cv_T2 = a.conv_T2; p.cv_F2 = a.conv_F2;
        p.Wfm = nullptr;
        if (cv_tps > 0) {
            p.trace = nullptr;
            p.cv_b_bytes = (unsigned)((size_t)a.conv_T1 * a.conv_F1 * a.conv_C * 2);
            p.cv_t_bytes = (unsigned)((size_t)2 * a.conv_F1 * a.conv_C * 2);
            p.cv_f_bytes = (unsigned)(2 * a.conv_C * 2);
            p.cv_seg_bytes = (unsigned)((size_t)a.conv_F1 * a.conv_C * 2);
        }
        rc = launch_rows(ctx, p, bm, s);
    }
    rs_prof_end(ctx, RS_PROF_GEMM, s);      // paired with rs_prof_begin on every path
    if (rc != RS_OK) return rc;
    RS_CHECK_LAUNCH(ctx, "gemm_smf16");
    return RS_OK;
}

// k_gemm_bf16.hip — bf16 x bf16 -> f32 MFMA GEMM with fused epilogue for gfx950.
//
//   out[M][N] = epilogue( A[M][K] . W[N][K]^T )        A, W row-major bf16 (K contiguous)
//
// These kernels carry every dense contraction of the encoder (SURVEY.md §8a rows S2-S4, L2, L3, L6
// pw1/pw2, D1): 96.8 % of the path's FLOPs.  Families, newest first (all share the staging, tile-order and epilogue
// ideas; the older ones stay selectable for A/B runs, DESIGN.md §4 "A/B knobs"):
//
//   gemm_smf16_kernel (default for the big shapes)  v_mfma_f32_16x16x32_bf16, 256x256 / 192x256 tiles, 64-deep K
//       tiles fetched as whole cache lines, LDS = five 32-KiB operand-part slots ("split ring": B(t+1) issued first,
//       A(t+2) one K tile further ahead), ping-pong wave groups, LDS-staged epilogue incl. the conv module's GLU.
//   gemm_lmf16_kernel  its predecessor: ring of two whole K tiles, optionally persistent (RS_GEMM_RING=0).
//   gemm_mf16_kernel   16x16x32 MFMA over a ring of four 32-deep granules (round 1; RS_GEMM_PERSISTENT=0).
//   gemm_bf16_kernel   v_mfma_f32_32x32x16_bf16; template parameters BM x BN tile, WM x WN waves, BK-deep K steps
//       through an NST-stage LDS ring; serves the small / narrow problems (128x128 tiles) and stays selectable
//       for the big ones (RS_GEMM_BIG=1).
//
// Common structure:
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip); loads of later stages stay
//     in flight across the (raw) s_barrier: the wait is a COUNTED s_waitcnt vmcnt(n), never a drain
//     (guide §5 T3+T4).
//   * 16-byte chunks of an LDS row are XOR-swizzled by row so the ds_read_b128 fragment reads are bank-conflict
//     free.  global_load_lds writes LDS linearly, so the swizzle is applied to the per-lane SOURCE address and
//     again on the read.
//   * blockIdx -> tile mapping is XCD-aware (each of the 8 XCDs, private L2, walks a contiguous run of tiles)
//     and grouped (group_m A row panels x a few weight tiles run together on an XCD).
//   * epilogue: the weight fragment is the MFMA A operand, so each lane ends up with consecutive output columns
//     of one row: +bias, ReLU / SiLU / GLU, *alpha, +residual (f32, prefetched), per-utterance row mask.  The
//     whole-line kernels pass the result through a per-wave LDS scratch so that every global store is a full
//     128 / 256-byte row segment; the older families store from registers.
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
    int tiles_m, tiles_n;
    int group_m;       // row panels per tile group (see tile_origin)
    int skew_cycles;   // one-off start delay of every second dispatch round (see launch_variant)
    long long* trace;  // debug: per-tile timestamps (scripts/gemm_trace.py); nullptr in production
};

template <int BK>
struct Swz {
    // 16-byte chunks per row and the row-dependent XOR that spreads a 16-lane read group over all
    // 64 banks (BK=64: rows alternate bank halves -> use (row>>1)&7; BK=32: 4 rows per 256 B bank
    // row -> use (row>>2)&3).
    static constexpr int CHUNKS = BK / 8;
    static constexpr int ROW_BYTES = BK * 2;
    __device__ static __forceinline__ int x(int row) {
        return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
    }
};

// issues the wave's global_load_lds number J0 .. J1-1 (of PER_WAVE) for one operand tile
template <int BK, int ROWS, int NWAVES, int J0 = 0, int J1 = -1>
__device__ __forceinline__ void stage_rows(const uint16_t* __restrict__ base, int ld, int row0, int max_row, int k0,
                                           char* lds_tile, int wave, int lane) {
    using S = Swz<BK>;
    constexpr int ROWS_PER_INST = 1024 / S::ROW_BYTES;       // 8 (BK=64) or 16 (BK=32)
    constexpr int INSTS = ROWS / ROWS_PER_INST;
    constexpr int PER_WAVE = INSTS / NWAVES;
    static_assert(INSTS % NWAVES == 0, "tile rows must split evenly over the waves");
    constexpr int JE = J1 < 0 ? PER_WAVE : (J1 < PER_WAVE ? J1 : PER_WAVE);
    const int r = lane / S::CHUNKS, pc = lane % S::CHUNKS;
#pragma unroll
    for (int j = J0; j < JE; ++j) {
        const int rbase = (wave * PER_WAVE + j) * ROWS_PER_INST;
        const int row = rbase + r;
        const int c = pc ^ S::x(row);
        int grow = row0 + row;
        grow = grow < max_row ? grow : max_row - 1;
        const uint16_t* src = base + (size_t)grow * ld + k0 + c * 8;
        char* dst = lds_tile + rbase * S::ROW_BYTES;  // wave-uniform; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

template <int BK>
__device__ __forceinline__ bf16x8_t read_frag(const char* lds_tile, int row, int chunk) {
    using S = Swz<BK>;
    const int off = row * S::ROW_BYTES + ((chunk ^ S::x(row)) << 4);
    return *reinterpret_cast<const bf16x8_t*>(lds_tile + off);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

template <int BM, int BN, int BK, int NST, int WM, int WN, bool PERSIST, bool RES, bool TRACE = false, bool PP = false>
__global__ __launch_bounds__(64 * WM * WN, (BM / WM) * (BN / WN) >= 128 * 128 ? 1 : 2) void gemm_bf16_kernel(GemmParams p) {
    constexpr int NWAVES = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int LOADS_PER_STAGE = STAGE_BYTES / 1024 / NWAVES;   // global_load_lds per wave per stage
    constexpr int KS = BK / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31, fhalf = lane >> 5;

    // XCD-aware, bijective tile order: workgroups b, b+8, b+16 .. run on one XCD (private L2); that
    // XCD owns a contiguous run of tiles, n-fastest, so it reads each A row panel once.
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, rr = nwg & 7;
    const int xbase = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    const int xcount = xcd < rr ? q + 1 : q;
    // PERSIST: the grid is one workgroup per CU and each walks its XCD's run with stride = the number
    // of workgroups on that XCD; otherwise one tile per workgroup.
    const int nslots = PERSIST ? (((int)gridDim.x - xcd + 7) >> 3) : xcount;
    if (slot >= xcount) return;

    if (PERSIST && p.skew_cycles > 0) {
        // All tiles cost the same, so without this every CU alternates in lockstep between "all MFMA"
        // and "all stores" and the output bursts are paid at full HBM-write latency.  Four start
        // phases a quarter tile apart spread the store traffic of the chip over time.
        const long long wait = (long long)p.skew_cycles * (slot & 3);
        const long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }

    const int nk = p.K / BK;
    const int flags = p.flags;
    const bool has_bias = flags & RS_GEMM_BIAS, relu = flags & RS_GEMM_RELU, silu = flags & RS_GEMM_SILU;
    constexpr bool has_res = RES;   // residual epilogue is its own instantiation: no dead residual registers elsewhere
    const bool out_f32 = RES || (flags & RS_GEMM_OUT_F32);
    const bool rowmask = flags & RS_GEMM_ROWMASK;
    const float alpha = p.alpha;

    auto issue = [&](int m0, int n0, int t) {
        char* st = smem + (t % NST) * STAGE_BYTES;
        stage_rows<BK, BM, NWAVES>(p.A, p.lda, m0, p.M, t * BK, st, wave, lane);
        stage_rows<BK, BN, NWAVES>(p.W, p.ldw, n0, p.N, t * BK, st + A_BYTES, wave, lane);
    };
    // Linear tile index -> (tile_m, tile_n), grouped: group_m row panels form a group that is walked
    // m-fastest, then across n.  The ~32 tiles an XCD runs at the same time are then group_m A panels x
    // 32/group_m weight tiles instead of 2 x 16: fewer distinct bytes per round, and the weight matrix
    // is re-streamed through that XCD's L2 once per group_m panels instead of once per two.
    auto tile_origin = [&](int j, int& m0, int& n0) {
        const int wg = xbase + j;
        const int per_group = p.group_m * p.tiles_n;
        const int g = wg / per_group, r = wg - g * per_group;
        const int left = p.tiles_m - g * p.group_m;
        const int gm = left < p.group_m ? left : p.group_m;
        const int tile_n = r / gm;
        m0 = (g * p.group_m + (r - tile_n * gm)) * BM;
        n0 = tile_n * BN;
    };

    int m0, n0;
    tile_origin(slot, m0, n0);
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(m0, n0, t);

    for (int j = slot; j < xcount; j += nslots) {
        long long ts0 = 0, ts1 = 0, ts2 = 0, wall0 = 0;
        if constexpr (TRACE) { ts0 = __builtin_readcyclecounter(); wall0 = (long long)wall_clock64(); }
        f32x16_t acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jj = 0; jj < NI; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.0f;

        if constexpr (PP) {
            // ---- ping-pong main loop (BK = 32 granules in a 4-deep ring).  The two wave rows (wm = 0 / 1:
            // one wave of each per SIMD) run half a phase apart: while one group issues its 8 MFMAs of a
            // 16-deep k slice, the other fetches its next fragments from LDS and issues its share of the
            // DMA for the granule three ahead; two barriers per phase keep the alternation exact.
            //   group 0:      mem(p) | B | mfma(p) | B | mem(p+1) | B | ...
            //   group 1:  B | mem(p) | B | mfma(p) | B | ...                       (one barrier late)
            // RAW: granule g+1 is waited for (counted vmcnt) in the memory part of g's LAST phase and first
            // read one phase later, i.e. after a barrier that both groups' waits precede.  WAR: the DMA
            // into granule g-1's buffer starts in g's first phase, a full phase after both groups drained
            // (lgkmcnt(0) before the barrier) their last reads of it.
            static_assert(!PP || (BK == 32 && NST == 4 && WM == 2 && !PERSIST), "ping-pong loop: 32-deep granules, ring of 4");
            if (nk >= 3) wait_vmcnt<2 * LOADS_PER_STAGE>();
            else if (nk == 2) wait_vmcnt<LOADS_PER_STAGE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (wm == 1) __builtin_amdgcn_s_barrier();
            for (int g = 0; g < nk; ++g) {
                const char* at = smem + (g & 3) * STAGE_BYTES;
                const char* bt = at + A_BYTES;
                char* nst = smem + ((g + 3) & 3) * STAGE_BYTES;
                const bool more = g + 3 < nk;
                const int nk0 = (g + 3) * BK;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t af[MI], bfr[NI];
#pragma unroll
                    for (int jj = 0; jj < NI; ++jj) bfr[jj] = read_frag<BK>(bt, wn * TN + jj * 32 + frow, ks * 2 + fhalf);
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = read_frag<BK>(at, wm * TM + i * 32 + frow, ks * 2 + fhalf);
                    if (ks == 1 && g + 1 < nk) {
                        if (g + 3 < nk) wait_vmcnt<LOADS_PER_STAGE + LOADS_PER_STAGE / 2>();
                        else if (g + 2 < nk) wait_vmcnt<LOADS_PER_STAGE>();
                        else wait_vmcnt<0>();
                    }
                    if (more) {
                        if (ks == 0) stage_rows<BK, BM, NWAVES>(p.A, p.lda, m0, p.M, nk0, nst, wave, lane);
                        else stage_rows<BK, BN, NWAVES>(p.W, p.ldw, n0, p.N, nk0, nst + A_BYTES, wave, lane);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int jj = 0; jj < NI; ++jj)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[jj], af[i], acc[i][jj], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (wm == 0) __builtin_amdgcn_s_barrier();
        } else
        for (int t = 0; t < nk; ++t) {
            // stage t must have landed: at most NST-2 younger stages may still be in flight.
            // In the persistent loop the previous tile's stores share the counter and loads/stores may
            // retire out of order with respect to each other, so the first wait of a tile is a drain.
            if (t + NST - 2 < nk && !(PERSIST && t == 0)) wait_vmcnt<LOADS_PER_STAGE * (NST - 2)>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // everyone's part of stage t is in LDS; stage t-1's buffer is free
            if constexpr (TRACE) { if (t == 0) ts1 = __builtin_readcyclecounter(); }
            const char* at = smem + (t % NST) * STAGE_BYTES;
            const char* bt = at + A_BYTES;
            // software pipeline over the k sub-steps: the fragments of ks+1 are requested from LDS before
            // the MFMAs of ks issue, and the next stage's DMA is issued under the first fragment reads
            bf16x8_t af[2][MI], bfr[2][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[0][i] = read_frag<BK>(at, wm * TM + i * 32 + frow, fhalf);
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) bfr[0][jj] = read_frag<BK>(bt, wn * TN + jj * 32 + frow, fhalf);
            const bool more = t + NST - 1 < nk;
            char* nst = smem + ((t + NST - 1) % NST) * STAGE_BYTES;
            const int nk0 = (t + NST - 1) * BK;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                // the next stage's DMA is spread over the k sub-steps (a global_load_lds costs ~100 issue
                // cycles: eight of them in one burst after the barrier left the MFMA pipe idle for a
                // third of every K step)
                if (more) {
                    constexpr int PW_A = (BM / (1024 / (BK * 2))) / NWAVES, PW_B = (BN / (1024 / (BK * 2))) / NWAVES;
                    constexpr int HALF = KS / 2 > 0 ? KS / 2 : 1;
                    if (KS >= 2) {
                        if (ks < HALF) {
                            if (ks == 0) stage_rows<BK, BM, NWAVES, 0, (PW_A + HALF - 1) / HALF>(p.A, p.lda, m0, p.M, nk0, nst, wave, lane);
                            if (ks == 1 && HALF > 1) stage_rows<BK, BM, NWAVES, (PW_A + HALF - 1) / HALF, PW_A>(p.A, p.lda, m0, p.M, nk0, nst, wave, lane);
                        } else {
                            if (ks == HALF) stage_rows<BK, BN, NWAVES, 0, (PW_B + HALF - 1) / HALF>(p.W, p.ldw, n0, p.N, nk0, nst + A_BYTES, wave, lane);
                            if (ks == HALF + 1 && HALF > 1) stage_rows<BK, BN, NWAVES, (PW_B + HALF - 1) / HALF, PW_B>(p.W, p.ldw, n0, p.N, nk0, nst + A_BYTES, wave, lane);
                        }
                    }
                }
                if (ks + 1 < KS) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        af[nxt][i] = read_frag<BK>(at, wm * TM + i * 32 + frow, (ks + 1) * 2 + fhalf);
#pragma unroll
                    for (int jj = 0; jj < NI; ++jj)
                        bfr[nxt][jj] = read_frag<BK>(bt, wn * TN + jj * 32 + frow, (ks + 1) * 2 + fhalf);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jj = 0; jj < NI; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][jj], af[cur][i], acc[i][jj], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }

        // ---- next tile's first stages go in flight before this tile's epilogue, so their HBM/L2
        // latency (and the workgroup launch a non-persistent grid would pay) hides under the stores
        if constexpr (TRACE) ts2 = __builtin_readcyclecounter();
        int cm0 = m0, cn0 = n0;
        // make the epilogue's addresses un-hoistable: computed at kernel entry they would be kept alive
        // across the main loop and spilled (~90 VGPRs of scratch traffic around every tile)
        asm volatile("" : "+s"(cm0), "+s"(cn0));
        if (PERSIST && j + nslots < xcount) {
            __builtin_amdgcn_s_barrier();   // all fragment reads of the last stage are done
            tile_origin(j + nslots, m0, n0);
#pragma unroll
            for (int t = 0; t < NST - 1; ++t)
                if (t < nk) issue(m0, n0, t);
        }

        // ---- epilogue, straight from registers.  The MFMAs ran with the weight fragment as the A
        // operand, so D = (A.W^T)^T: lane = (m = lane&31, half h), register r <-> n = (r&3) + 8*(r>>2) + 4*h.
        // Each lane owns 4 consecutive n per register quad: one 16-byte f32 (8-byte bf16) access; the
        // two half-waves cover a contiguous 32-byte (16-byte) run of one output row.
        // Loads first, math later.  The unit of work is half a 32x32 block (two register quads = the
        // pair that the bf16 store widens with v_permlane32_swap).  Without a residual the bias of the
        // whole wave tile is fetched up front; with one, the (bias, residual) of unit k+1 is requested
        // before unit k is processed, so no store waits on a load issued right before it (per-block
        // exposed L2/HBM latency was 27 % of a K=1024 tile — profiles/r01k_gemm_tile_timeline.txt)
        // and only 32 VGPRs are in flight (the 128x128 kernel keeps 3 workgroups per CU).
        constexpr int UNITS = MI * NI * 2;
        constexpr bool BIAS_PRELOAD = !has_res && NI <= 2;   // wide wave tiles fetch the bias per unit (registers)
        float4 bias_all[BIAS_PRELOAD ? NI : 1][BIAS_PRELOAD ? 4 : 1];
        if constexpr (BIAS_PRELOAD) {
            if (has_bias) {
#pragma unroll
                for (int jj = 0; jj < NI; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = cn0 + wn * TN + jj * 32 + 4 * fhalf + 8 * g;
                        bias_all[jj][g] = n < p.N ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
            }
        }
        auto load_unit = [&](int u, float4 (&rv)[2], float4 (&bv)[2]) {   // RES kernels only
            const int blk = u >> 1, gp = u & 1;
            const int i = blk / NI, jj = blk % NI;
            const int m = cm0 + wm * TM + i * 32 + frow;
            const int nb = cn0 + wn * TN + jj * 32 + 4 * fhalf;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int n = nb + 8 * (2 * gp + q);
                const bool ok = m < p.M && n < p.N;
                rv[q] = ok ? *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.ldc + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[q] = (has_bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        float4 rv_next[2], bv_next[2];
        if constexpr (has_res) load_unit(0, rv_next, bv_next);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int blk = u >> 1, gp = u & 1;
            const int i = blk / NI, jj = blk % NI;
            const int m = cm0 + wm * TM + i * 32 + frow;
            const bool m_ok = m < p.M;
            bool keep = true;
            if (rowmask && m_ok) {
                const int step = m / p.mask_rows_per_step;
                const int b = step / p.mask_steps;
                keep = step - b * p.mask_steps < p.mask_lens[b];
            }
            const size_t rowoff = (size_t)m * p.ldc;
            const int nb = cn0 + wn * TN + jj * 32 + 4 * fhalf;
            float4 rv[2], bv[2];
            if constexpr (has_res) {
#pragma unroll
                for (int q = 0; q < 2; ++q) { rv[q] = rv_next[q]; bv[q] = bv_next[q]; }
                if (u + 1 < UNITS) load_unit(u + 1, rv_next, bv_next);
            } else if constexpr (BIAS_PRELOAD) {
#pragma unroll
                for (int q = 0; q < 2; ++q) bv[q] = has_bias ? bias_all[jj][2 * gp + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int n = nb + 8 * (2 * gp + q);
                    bv[q] = (has_bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            float4 v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int g = 2 * gp + q;
                v[q] = make_float4(acc[i][jj][4 * g], acc[i][jj][4 * g + 1], acc[i][jj][4 * g + 2], acc[i][jj][4 * g + 3]);
                v[q].x += bv[q].x; v[q].y += bv[q].y; v[q].z += bv[q].z; v[q].w += bv[q].w;
                if (relu) { v[q].x = fmaxf(v[q].x, 0.f); v[q].y = fmaxf(v[q].y, 0.f); v[q].z = fmaxf(v[q].z, 0.f); v[q].w = fmaxf(v[q].w, 0.f); }
                if (silu) { v[q].x = silu_f(v[q].x); v[q].y = silu_f(v[q].y); v[q].z = silu_f(v[q].z); v[q].w = silu_f(v[q].w); }
                v[q].x *= alpha; v[q].y *= alpha; v[q].z *= alpha; v[q].w *= alpha;
                if constexpr (has_res) { v[q].x += rv[q].x; v[q].y += rv[q].y; v[q].z += rv[q].z; v[q].w += rv[q].w; }
                if (!keep) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (out_f32) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int n = nb + 8 * (2 * gp + q);
                    if (m_ok && n < p.N) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + rowoff + n) = v[q];
                }
            } else {
                // bf16: pair the two register quads across the half-waves with v_permlane32_swap so
                // every lane stores 8 consecutive columns (16 bytes) and the two half-waves together a
                // contiguous 32-byte run of the row (guide T21)
                const u16x4_t pa = pack_bf16x4(v[0].x, v[0].y, v[0].z, v[0].w);
                const u16x4_t pb = pack_bf16x4(v[1].x, v[1].y, v[1].z, v[1].w);
                unsigned a0 = __builtin_bit_cast(uint2, pa).x, a1 = __builtin_bit_cast(uint2, pa).y;
                unsigned b0 = __builtin_bit_cast(uint2, pb).x, b1 = __builtin_bit_cast(uint2, pb).y;
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                // h = 0: [own g | partner's g] -> columns 8g .. 8g+7 ; h = 1: [partner's g+1 | own g+1]
                const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                const int n = cn0 + wn * TN + jj * 32 + 8 * (2 * gp + fhalf);
                if (m_ok && n < p.N) {
                    uint16_t* dst = reinterpret_cast<uint16_t*>(p.out) + rowoff + n;
                    if (n + 8 <= p.N) *reinterpret_cast<uint4*>(dst) = o;
                    else *reinterpret_cast<uint2*>(dst) = make_uint2(o.x, o.y);   // N % 8 == 4 tail
                }
            }
            // one unit at a time (the next unit's loads are already in flight): without this fence the
            // scheduler hoists every unit's loads above the first store and spills
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (TRACE) {
            const long long ts3 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long ts4 = __builtin_readcyclecounter();
            if (tid == 0) {
                long long* tr = p.trace + (size_t)(xbase + j) * 8;
                tr[0] = 0; tr[1] = ts1 - ts0; tr[2] = ts2 - ts0; tr[3] = ts3 - ts0; tr[4] = ts4 - ts0; tr[5] = bid;
                tr[6] = wall0; tr[7] = (long long)wall_clock64();   // 100 MHz, chip-wide
            }
        }
    }
}


// =====================================================================================================
// 16x16x32 variant.  Same tile (256 x 256, 8 waves of 128 x 64), ping-pong wave groups over a ring of four
// 32-deep granules, but the products run on v_mfma_f32_16x16x32_bf16.  Reason: the chip is package-power
// limited in the sustained regime and, registers only, the 16x16x32 shape sustains 2.45 PF/s
// where 32x32x16 is throttled to 1.9-2.06 PF/s (scripts/mfma_power.hip, profiles/r01w_mfma_shape_power.txt):
// fewer joules per FLOP is the lever that is left.
//   fragments: lane = (row = lane & 15, k chunk = lane >> 4), 8 bf16 (16 B) per lane, one ds_read_b128 per
//   16 x 32 block.  LDS rows are 64 B (4 chunks); chunk c of row r sits at c ^ f[(r >> 2) & 3] with
//   f = {0, 2, 3, 1}: the four 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) then hit 16 distinct
//   16-byte slots.
//   D^T trick as above: weights are the A operand, so a lane ends up with m = lane & 15 and the 4 consecutive
//   n = 4 * (lane >> 4) + reg of every 16 x 16 block: 16-byte f32 / 8-byte bf16 accesses, 64 / 32 contiguous
//   bytes per row and instruction.
__device__ __forceinline__ int swz16(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_rows16(const uint16_t* __restrict__ base, int ld, int row0, int max_row, int k0,
                                             char* lds_tile, int wave, int lane) {
    constexpr int INSTS = ROWS / 16, PER_WAVE = INSTS / NWAVES;
    static_assert(INSTS % NWAVES == 0, "tile rows must split evenly over the waves");
    const int r = lane >> 2, pc = lane & 3;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int rbase = (wave * PER_WAVE + j) * 16;
        const int row = rbase + r;
        const int c = pc ^ swz16(row);
        int grow = row0 + row;
        grow = grow < max_row ? grow : max_row - 1;
        const uint16_t* src = base + (size_t)grow * ld + k0 + c * 8;
        char* dst = lds_tile + rbase * 64;   // wave-uniform; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8_t read_frag16(const char* lds_tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8_t*>(lds_tile + row * 64 + ((chunk ^ swz16(row)) << 4));
}

// runtime (wave-uniform) counted wait for the literals this kernel needs
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 8: wait_vmcnt<8>(); break;
        default: wait_vmcnt<0>(); break;
    }
}

template <int BM, int BN, bool RES>
__global__ __launch_bounds__(512, 2) void gemm_mf16_kernel(GemmParams p) {
    constexpr int BK = 32, NST = 4, WM = 2, WN = 4, NWAVES = 8;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16, MH = MI / 2;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INSTS = BM / 16, LB = BN / 16 / NWAVES;
    static_assert(BN == 256 && (A_INSTS == 16 || A_INSTS == 12) && (MI % 2) == 0, "tile shapes: 256 x 256 or 192 x 256");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fch = lane >> 4;
    // A-tile DMA instructions (16 rows each) of this wave per granule: ids wave, wave + 8 (< A_INSTS).
    // 192-row tiles have 12: the waves of group 0 issue two, those of group 1 one.
    const int LAw = (A_INSTS - wave + NWAVES - 1) / NWAVES;

    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, rr = nwg & 7;
    const int xbase = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    const int xcount = xcd < rr ? q + 1 : q;
    if (slot >= xcount) return;
    int m0, n0;
    {
        const int wg = xbase + slot;
        const int per_group = p.group_m * p.tiles_n;
        const int g = wg / per_group, r = wg - g * per_group;
        const int left = p.tiles_m - g * p.group_m;
        const int gm = left < p.group_m ? left : p.group_m;
        const int tile_n = r / gm;
        m0 = (g * p.group_m + (r - tile_n * gm)) * BM;
        n0 = tile_n * BN;
    }
    const int nk = p.K / BK;
    const int flags = p.flags;
    const bool has_bias = flags & RS_GEMM_BIAS, relu = flags & RS_GEMM_RELU, silu = flags & RS_GEMM_SILU;
    const bool out_f32 = RES || (flags & RS_GEMM_OUT_F32);
    const bool rowmask = flags & RS_GEMM_ROWMASK;
    const float alpha = p.alpha;

    auto issue_a = [&](int t) {
        char* tile = smem + (t & 3) * STAGE_BYTES;
        const int r = lane >> 2, pc = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = wave + NWAVES * j;
            if (id < A_INSTS) {                                   // wave-uniform
                const int row = id * 16 + r;
                int grow = m0 + row;
                grow = grow < p.M ? grow : p.M - 1;
                const uint16_t* src = p.A + (size_t)grow * p.lda + t * BK + (pc ^ swz16(row)) * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(tile + id * 16 * 64), 16, 0, 0);
            }
        }
    };
    auto issue_b = [&](int t) { stage_rows16<BN, NWAVES>(p.W, p.ldw, n0, p.N, t * BK, smem + (t & 3) * STAGE_BYTES + A_BYTES, wave, lane); };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) { issue_a(t); issue_b(t); }

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    // ping-pong main loop: see the PP branch of gemm_bf16_kernel for the barrier / RAW / WAR argument
    wait_vmcnt_rt(nk >= 3 ? 2 * (LAw + LB) : (nk == 2 ? LAw + LB : 0));
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int g = 0; g < nk; ++g) {
        const char* at = smem + (g & 3) * STAGE_BYTES;
        const char* bt = at + A_BYTES;
        const bool more = g + 3 < nk;
        bf16x8_t bfr[NI];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {       // ks = which half of the wave's 128 activation rows
            bf16x8_t af[MH];
            if (ks == 0) {
#pragma unroll
                for (int j = 0; j < NI; ++j) bfr[j] = read_frag16(bt, wn * TN + j * 16 + frow, fch);
            }
#pragma unroll
            for (int i = 0; i < MH; ++i) af[i] = read_frag16(at, wm * TM + (ks * MH + i) * 16 + frow, fch);
            if (ks == 1 && g + 1 < nk)     // younger than granule g+1: all of g+2, the A part of g+3
                wait_vmcnt_rt(g + 3 < nk ? 2 * LAw + LB : (g + 2 < nk ? LAw + LB : 0));
            if (more) {
                if (ks == 0) issue_a(g + 3);
                else issue_b(g + 3);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MH; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[ks * MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[ks * MH + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue from registers: block (i, j) -> m = .. + i*16 + (lane & 15), n = .. + j*16 + 4*(lane >> 4) + 0..3
    int cm0 = m0, cn0 = n0;
    asm volatile("" : "+s"(cm0), "+s"(cn0));   // keep the addresses out of the main loop's live ranges
    float4 bias_r[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = cn0 + wn * TN + j * 16 + 4 * fch;
        bias_r[j] = (has_bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto load_res = [&](int i, float4 (&rv)[NI]) {
        const int m = cm0 + wm * TM + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = cn0 + wn * TN + j * 16 + 4 * fch;
            rv[j] = (m < p.M && n < p.N) ? *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.ldc + n)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 rv_next[NI];
    if constexpr (RES) load_res(0, rv_next);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = cm0 + wm * TM + i * 16 + frow;
        const bool m_ok = m < p.M;
        bool keep = true;
        if (rowmask && m_ok) {
            const int step = m / p.mask_rows_per_step;
            const int b = step / p.mask_steps;
            keep = step - b * p.mask_steps < p.mask_lens[b];
        }
        float4 rv[NI];
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < NI; ++j) rv[j] = rv_next[j];
            if (i + 1 < MI) load_res(i + 1, rv_next);
        }
        const size_t rowoff = (size_t)m * p.ldc;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = cn0 + wn * TN + j * 16 + 4 * fch;
            float4 v = make_float4(acc[i][j][0] + bias_r[j].x, acc[i][j][1] + bias_r[j].y, acc[i][j][2] + bias_r[j].z,
                                   acc[i][j][3] + bias_r[j].w);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
            if constexpr (RES) { v.x += rv[j].x; v.y += rv[j].y; v.z += rv[j].z; v.w += rv[j].w; }
            if (!keep) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m_ok && n < p.N) {
                if (out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + rowoff + n) = v;
                else *reinterpret_cast<u16x4_t*>(reinterpret_cast<uint16_t*>(p.out) + rowoff + n) = pack_bf16x4(v.x, v.y, v.z, v.w);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// =====================================================================================================
// Cross-tile 16x16x32 kernels — shared pieces (DMA helper, LDS-staged epilogue).  The kernel itself is
// gemm_lmf16_kernel below; its first version (gemm_pmf16_kernel: 32-deep granules, six DMA schedules, ablation
// builds) was removed once the whole-line kernel superseded it — the measurements that led there are kept in
// profiles/r02b .. r02e_*.txt and the code in the git history of this file.
//
// Why a workgroup walks several tiles / keeps its LDS ring running across tile boundaries: with one big-LDS
// workgroup per CU, a tile's epilogue and the next tile's prologue (first HBM / L2 round trip) overlap with
// nothing (profiles/r01k_gemm_tile_timeline.txt: 6-11 us of a 41 us K = 1024 tile).  So
//   * the last K tile of an output tile already DMAs the first K tile of the NEXT one: the epilogue runs with those
//     loads in flight and the next main loop starts on data that has landed (grid = CUs - reserved workgroups in
//     persistent mode; with one tile per workgroup the same code simply has no next tile);
//   * the epilogue goes through a 4 KiB per-wave LDS scratch (the 32 KiB the ring leaves free) that turns the MFMA
//     accumulator layout (16 rows x 8 bytes per instruction) into whole 128 / 256-byte row segments: half the store
//     instructions, every one a full cache line;
//   * stores are fire-and-forget (vmcnt retires in order and counts stores: the next tile's waits cover them).
// The DMAs are issued from inline asm (global_load_lds_dwordx4, scalar base + 32-bit lane offset): hipcc then counts
// only the epilogue's own loads / stores, which are all younger than the DMAs in flight, so its counted waits stay
// correct and it never drains the ring (guide §5 "three .s-level traps" (b)).  Epilogue memory operations are raw
// buffer loads / stores with an out-of-range offset for masked rows: they always issue, whatever the row mask.
// M0 (the DMA's LDS base) is written and consumed inside one asm statement.  Nothing else in these kernels
// uses M0 (gfx950 DS instructions do not), so it is not saved / restored around the statement.
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// ---- epilogue shared by the cross-tile kernels (no workgroup barrier: the scratch is per wave, LDS operations of
// one wave execute in order).  bf16 output: chunks of 32 rows x 64 columns; f32: 16 rows x 64 columns; both 4 KiB,
// 16-byte pieces XOR-swizzled by row.  All global accesses are raw buffer operations (masked rows / columns get an
// out-of-range offset), so the number of VMEM operations a wave issues here is a compile-time constant.
template <int MI, int NI, int OUT, bool MASK, int PF = 1>
__device__ __forceinline__ void pmf16_epilogue(const GemmParams& p, f32x4_t (&acc)[MI][NI], char* scr, int cm0, int cn0,
                                               int wm, int wn, int lane) {
    constexpr bool RES = OUT == 2, out_f32 = OUT == 1 || OUT == 2, GLU = OUT == 3, rowmask = MASK;
    constexpr int TM = MI * 16, TN = NI * 16;
    constexpr unsigned OOB = 0xfffffff0u;                         // beyond every buffer: loads return 0, stores are dropped
    const int frow = lane & 15, fch = lane >> 4;
    const int flags = p.flags;
    const bool has_bias = flags & RS_GEMM_BIAS, relu = flags & RS_GEMM_RELU, silu = flags & RS_GEMM_SILU;
    const float alpha = p.alpha;
    // GLU: the output has N / 2 columns (ldc is the caller's row pitch of that narrower matrix)
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
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        return v;
    };
    auto row_keep = [&](int m) -> bool {
        if (!rowmask || m >= p.M) return true;      // rowmask is a template constant
        const int step = m / p.mask_rows_per_step;
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
        // lookahead every chunk paid a full HBM round trip (a dependent chain of MI latencies per wave, ~14 us of a
        // 36 us K = 1024 tile); with PF = MI every residual load of the tile is in flight before the first chunk
        // is touched and the epilogue costs one latency.
        const int rr4 = lane >> 4, cc = lane & 15;
        constexpr int NPF = RES ? (PF < 1 ? 1 : (PF > MI ? MI : PF)) : 1;
        u32x4_t rvq[NPF][4];
        auto load_res = [&](int i, u32x4_t (&rv)[4]) {
#pragma unroll
            for (int sgm = 0; sgm < 4; ++sgm) {
                const int m = wrow0 + i * 16 + sgm * 4 + rr4, n = wcol0 + cc * 4;
                const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u : OOB;
                rv[sgm] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, off, 0, 0);
            }
        };
        if constexpr (RES) {
#pragma unroll
            for (int d = 0; d < NPF; ++d) load_res(d, rvq[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            u32x4_t rv[4];
            if constexpr (RES) {
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) rv[sgm] = rvq[i % NPF][sgm];
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
                    const float4 r = __builtin_bit_cast(float4, rv[sgm]);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                const int m = wrow0 + i * 16 + row, n = wcol0 + cc * 4;
                if (!row_keep(m)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                const unsigned off = (m < p.M && n < p.N) ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), out_rsrc, off, 0, 0);
            }
            if constexpr (RES) {
                if (i + NPF < MI) load_res(i + NPF, rvq[i % NPF]);       // refill the slot this chunk just consumed
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// =====================================================================================================
// Cross-tile kernel with WHOLE-LINE operand fetches (the default for the big shapes).
//
// gemm_pmf16_kernel stages 32-deep granules: every DMA instruction fetches 16 rows x 64 B, i.e. HALF of each 128-byte
// cache line; the other half is fetched again one granule later, after 32 KiB of other lines went through the 32 KiB
// L1, so the L2 -> L1 path moves every operand byte twice.  Measured (profiles/r02e_*: ablation 349): fetching 8 rows
// x 128 B per instruction instead is worth +17 %.  This kernel therefore keeps 64-deep K tiles in LDS:
//   * LDS: ring of two K tiles, each [BM + 256 rows][128 B] (64 KiB at BM = 256), 16-byte chunks XOR-swizzled by
//     (row >> 1) & 7 (conflict-free for the 16 x 4-chunk fragment reads of v_mfma_f32_16x16x32_bf16), plus the
//     4 KiB-per-wave epilogue scratch;
//   * a K tile is consumed in two phases (k-steps of 32), each [12 fragment reads | 32 MFMAs], ping-pong wave
//     groups as before;
//   * K tile t+1 (8 DMA instructions per wave) is issued during phase (t, 0) — NM0 of them with the fragment reads,
//     the rest between the MFMAs — into the buffer whose last reads finished in phase (t-1, 1);
//   * it is waited for with vmcnt(0) one interval before its first read: group 0 at the END of its MFMA half of phase
//     (t, 1), group 1 (one barrier behind) in its memory half of (t, 1).  Nothing younger than those DMAs exists at
//     that point (the previous epilogue's stores are older, vmcnt retires in order), so no counting is needed;
//   * the ring keeps running across the tiles of a persistent workgroup exactly as in gemm_pmf16_kernel.
__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }

// ABL (profiling builds): 1 = no MFMAs, 2 = no DMAs in the main loop, 3 = no fragment reads (wrong results);
// 4 = correct results plus a per-tile timeline in p.trace (scripts/gemm_trace.py).
// EPF: residual prefetch depth of the f32 epilogue in 16-row chunks (pmf16_epilogue).
template <int BM, int OUT, bool MASK, int NM0, int ABL = 0, int EPF = 1>
__global__ __launch_bounds__(512, 2) void gemm_lmf16_kernel(GemmParams p) {
    constexpr bool TRACE = ABL == 4;
    long long tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_w0 = 0, tr_stall = 0;
    if constexpr (TRACE) { tr_t0 = __builtin_readcyclecounter(); tr_w0 = (long long)wall_clock64(); }
    constexpr int BN = 256, WN = 4, NWAVES = 8;
    constexpr int TM = BM / 2, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int LA = BM / 8 / NWAVES, LB = BN / 8 / NWAVES, NP = LA + LB;   // DMA pieces (8 rows x 128 B) per wave and K tile
    constexpr int RING_BYTES = 2 * STAGE_BYTES, SCR_BYTES = 4096;
    static_assert((BM == 256 || BM == 192) && LB == 4 && (MI % 2) == 0, "tile shapes: 256 x 256 or 192 x 256");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fch = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    char* scr = smem + RING_BYTES + wave * SCR_BYTES;

    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, rr = nwg & 7;
    const int xbase = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    const int xcount = xcd < rr ? q + 1 : q;
    const int nslots = ((int)gridDim.x - xcd + 7) >> 3;
    if (slot >= xcount) return;
    auto tile_origin = [&](int j, int& m0, int& n0) {
        const int wg = xbase + j;
        const int per_group = p.group_m * p.tiles_n;
        const int g = wg / per_group, r = wg - g * per_group;
        const int left = p.tiles_m - g * p.group_m;
        const int gm = left < p.group_m ? left : p.group_m;
        const int tile_n = r / gm;
        m0 = (g * p.group_m + (r - tile_n * gm)) * BM;
        n0 = tile_n * BN;
    };
    // per-lane byte offsets of this wave's DMA pieces: lane = (row l >> 3 of the piece, physical chunk l & 7)
    const int dr = lane >> 3, dpc = lane & 7;
    unsigned off[NP];
    auto lane_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int row = (wave + NWAVES * j) * 8 + dr;
            int gr = m0 + row;
            gr = gr < p.M ? gr : p.M - 1;
            off[j] = (unsigned)gr * (unsigned)(p.lda * 2) + (unsigned)((dpc ^ swz64(row)) * 16);
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int row = (wave * LB + j) * 8 + dr;
            int gr = n0 + row;
            gr = gr < p.N ? gr : p.N - 1;
            off[LA + j] = (unsigned)gr * (unsigned)(p.ldw * 2) + (unsigned)((dpc ^ swz64(row)) * 16);
        }
    };
    // piece q of K tile t into ring buffer `buf`
    auto dma = [&](int q, int t, int buf) {
        const char* base = (q < LA ? reinterpret_cast<const char*>(p.A) : reinterpret_cast<const char*>(p.W)) + (size_t)t * 128;
        const unsigned dst = lds0 + buf * STAGE_BYTES + (q < LA ? (wave + NWAVES * q) * 1024 : A_BYTES + (wave * LB + (q - LA)) * 1024);
        glds16(off[q], base, dst);
    };
    auto frag = [&](const char* tile, int row, int chunk) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ swz64(row)) << 4));
    };

    const int nk = p.K / 64;                                      // K tiles per output tile, >= 2 (launcher)
    int m0, n0;
    tile_origin(slot, m0, n0);
    lane_offsets(m0, n0);
    auto wait_ktile = [&]() { wait_vmcnt<0>(); };
    int gc = 0;                                                   // K tiles consumed so far (ring position)
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) dma(qq, 0, 0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one barrier behind from here on
    if constexpr (TRACE) tr_t1 = __builtin_readcyclecounter();

    for (int j = slot; j < xcount; j += nslots) {
        const bool has_next = j + nslots < xcount;
        const int cm0 = m0, cn0 = n0;
        f32x4_t acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jj = 0; jj < NI; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.0f;

        for (int t = 0; t < nk; ++t) {
            const char* at = smem + ((gc + t) & 1) * STAGE_BYTES;
            const char* bt = at + A_BYTES;
            const bool own = t + 1 < nk;                          // K tile t+1 belongs to this output tile
            const bool more = own || has_next;
            const int tn = own ? t + 1 : 0;
            const int nbuf = (gc + t + 1) & 1;
            if (!own && has_next) {                               // all DMAs of this tile are issued: switch to the next tile
                tile_origin(j + nslots, m0, n0);
                lane_offsets(m0, n0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t bfr[NI], af[MI];
                if (ABL != 3 || (t == 0 && ks == 0)) {
#pragma unroll
                    for (int jj = 0; jj < NI; ++jj) bfr[jj] = frag(bt, wn * TN + jj * 16 + frow, ks * 4 + fch);
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = frag(at, wm * TM + i * 16 + frow, ks * 4 + fch);
                }
                if (ks == 0 && more && ABL != 2) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int qq = 0; qq < (NM0 < NP ? NM0 : NP); ++qq) dma(qq, tn, nbuf);
                }
                if (ks == 1 && more && wm == 1) {                           // group 1: K tile t+1 landed (see header)
                    if constexpr (TRACE) { const long long a = __builtin_readcyclecounter(); wait_ktile(); tr_stall += __builtin_readcyclecounter() - a; }
                    else wait_ktile();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                constexpr int REST = NP - (NM0 < NP ? NM0 : NP);            // pieces issued between the MFMAs of phase (t, 0)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if constexpr (ABL == 1) asm volatile("" :: "v"(af[i]), "v"(bfr[i & 3]));
                    else {
#pragma unroll
                    for (int jj = 0; jj < NI; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj], af[i], acc[i][jj], 0, 0, 0);
                    }
                    if (REST > 0 && ks == 0 && i < REST && more && ABL != 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma(NP - REST + i, tn, nbuf);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (ks == 1 && more && wm == 0) {                           // group 0: before the barrier its reads follow
                    if constexpr (TRACE) { const long long a = __builtin_readcyclecounter(); wait_ktile(); tr_stall += __builtin_readcyclecounter() - a; }
                    else wait_ktile();
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        gc += nk;
        {
            int em0 = __builtin_amdgcn_readfirstlane(cm0), en0 = __builtin_amdgcn_readfirstlane(cn0);
            asm volatile("" : "+s"(em0), "+s"(en0));
            // Group 1 runs one barrier behind: its last in-loop barrier pairs with one more barrier of group 0.  Group 0
            // passes it BEFORE its epilogue (after the last tile): placed after the epilogue, as it was, group 1 sat at
            // that barrier until group 0 had issued its whole epilogue and the two epilogues ran back to back
            // (profiles/r02u_gemm_tile_timeline.txt: group 1's "main loop" 2.2 - 4.2 us longer than group 0's).  At
            // this point group 1 is past its last fragment reads, so nothing reads the ring any more.
            if (!has_next && wm == 0) __builtin_amdgcn_s_barrier();
            if constexpr (TRACE) tr_t2 = __builtin_readcyclecounter();
            pmf16_epilogue<MI, NI, OUT, MASK, EPF>(p, acc, scr, em0, en0, wm, wn, lane);
            if constexpr (TRACE) {
                // wave 0 (group 0) and wave 4 (group 1) each write a record: [0] prologue, [1] main loop, [2] epilogue
                // issue, [3] store drain, [4] cycles stalled in the K-tile waits, [5] block, [6] / [7] wall clock
                const long long t3 = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const long long t4 = __builtin_readcyclecounter();
                if (lane == 0 && wn == 0) {
                    long long* tr = p.trace + ((size_t)(xbase + j) * 2 + wm) * 8;
                    tr[0] = tr_t1 - tr_t0; tr[1] = tr_t2 - tr_t1; tr[2] = t3 - tr_t2; tr[3] = t4 - t3; tr[4] = tr_stall;
                    tr[5] = bid; tr[6] = tr_w0; tr[7] = (long long)wall_clock64();
                }
            }
        }
    }
}

// =====================================================================================================
// Split-ring variant of the whole-line kernel (one tile per workgroup).
//
// What bounds gemm_lmf16_kernel's main loop is not the MFMA rate but the round trip of a K tile's DMAs
// (profiles/r02u_gemm_tile_timeline.txt): its ring holds two K tiles, so all 56 - 64 KiB of K tile t + 1 are issued
// during phase (t, 0) and must have landed by the end of phase (t, 1) — one K tile of lookahead, and the pieces a wave
// issues last queue behind everything issued before them (the CU's L1 -> LDS path moves 64 B/clk: 64 KiB is ~0.5 us
// of transfer on top of the latency).  A K tile takes ~1.4 us whatever the tile height (1.08 us of MFMA at 1.9 GHz
// for 256 rows, 0.81 for 192).  LDS has no room for a third K tile, but it has room for HALF of one: here the
// 160 KiB are five 32-KiB slots, a K tile is two parts (A rows, weight rows), part p lives in slot p mod 5, and
// during K tile t a wave issues first its pieces of B(t+1) — needed at the end of this K tile, now with nothing
// queued ahead of them — and then its pieces of A(t+2), which have a whole extra K tile to land.  The wait before
// K tile t + 1 is a counted vmcnt(LA): the A(t+2) pieces just issued may stay in flight.
// The epilogue scratch aliases slot 0 (the ring is dead by then: both wave groups are past their last fragment
// reads when group 0 passes its extra barrier).
//
// PP = false (EXPERIMENTAL, never validated on hardware: written after the round's GPU minutes were spent; opt-in with
// RS_GEMM_RING=5, tests behind RS_TEST_EXPERIMENTAL=1): no ping-pong — all eight waves run the same schedule and meet at
// ONE barrier per K tile (RAW for K tile t+1, WAR for the slots of K tile t) instead of four; the overlap of one
// wave's fragment reads with the other's MFMAs is left to the two waves that share a SIMD.  The timeline shows ~165
// cycles of barrier / wait overhead per 512-cycle MFMA interval; this is the cheapest way to find out how much of it
// the explicit ping-pong buys back.
template <int BM, int OUT, bool MASK, int NM0, int EPF = 1, bool TRACE = false, bool PP = true>
__global__ __launch_bounds__(512, 2) void gemm_smf16_kernel(GemmParams p) {
    constexpr int BN = 256, WN = 4, NWAVES = 8;
    constexpr int TM = BM / 2, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int SLOT = 32768, NSLOT = 5;
    constexpr int LA = BM / 8 / NWAVES, LB = BN / 8 / NWAVES, NP = LA + LB;   // DMA pieces (8 rows x 128 B) per wave and K tile
    static_assert((BM == 256 || BM == 192) && LB == 4 && (MI % 2) == 0 && BM * 128 <= SLOT, "tile shapes: 256 x 256 or 192 x 256");
    long long tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_w0 = 0, tr_stall = 0;
    if constexpr (TRACE) { tr_t0 = __builtin_readcyclecounter(); tr_w0 = (long long)wall_clock64(); }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fch = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    char* scr = smem + wave * 4096;

    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, xslot = bid >> 3;
    const int q8 = nwg >> 3, rr = nwg & 7;
    const int xbase = xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8;
    const int xcount = xcd < rr ? q8 + 1 : q8;
    if (xslot >= xcount) return;
    int m0, n0;
    {
        const int wg = xbase + xslot;
        const int per_group = p.group_m * p.tiles_n;
        const int g = wg / per_group, r = wg - g * per_group;
        const int left = p.tiles_m - g * p.group_m;
        const int gm = left < p.group_m ? left : p.group_m;
        const int tile_n = r / gm;
        m0 = (g * p.group_m + (r - tile_n * gm)) * BM;
        n0 = tile_n * BN;
    }
    // per-lane byte offsets of this wave's DMA pieces: lane = (row l >> 3 of the piece, physical chunk l & 7)
    const int dr = lane >> 3, dpc = lane & 7;
    unsigned off[NP];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int row = (wave + NWAVES * j) * 8 + dr;
        int gr = m0 + row;
        gr = gr < p.M ? gr : p.M - 1;
        off[j] = (unsigned)gr * (unsigned)(p.lda * 2) + (unsigned)((dpc ^ swz64(row)) * 16);
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int row = (wave * LB + j) * 8 + dr;
        int gr = n0 + row;
        gr = gr < p.N ? gr : p.N - 1;
        off[LA + j] = (unsigned)gr * (unsigned)(p.ldw * 2) + (unsigned)((dpc ^ swz64(row)) * 16);
    }
    auto dma_a = [&](int j, int t, int sl) {
        glds16(off[j], reinterpret_cast<const char*>(p.A) + (size_t)t * 128, lds0 + sl * SLOT + (wave + NWAVES * j) * 1024);
    };
    auto dma_b = [&](int j, int t, int sl) {
        glds16(off[LA + j], reinterpret_cast<const char*>(p.W) + (size_t)t * 128, lds0 + sl * SLOT + (wave * LB + j) * 1024);
    };
    auto frag = [&](const char* part, int row, int chunk) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(part + row * 128 + ((chunk ^ swz64(row)) << 4));
    };

    const int nk = p.K / 64;                                      // K tiles, >= 2 (launcher)
    // prologue: A(0) -> slot 0, B(0) -> slot 1, A(1) -> slot 2; K tile 0 is complete when all but the last LA landed
#pragma unroll
    for (int j = 0; j < LA; ++j) dma_a(j, 0, 0);
#pragma unroll
    for (int j = 0; j < LB; ++j) dma_b(j, 0, 1);
#pragma unroll
    for (int j = 0; j < LA; ++j) dma_a(j, 1, 2);
    wait_vmcnt<LA>();
    __builtin_amdgcn_s_barrier();
    if (PP && wm == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one barrier behind from here on
    if constexpr (TRACE) tr_t1 = __builtin_readcyclecounter();

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.0f;

    int sa = 0;                                                   // slot of A(t); B(t) = sa + 1, B(t+1) = sa + 3, A(t+2) = sa + 4 (mod 5)
    for (int t = 0; t < nk; ++t) {
        auto wrap = [](int v) { return v >= NSLOT ? v - NSLOT : v; };
        const int sb = wrap(sa + 1), sbn = wrap(sa + 3), san = wrap(sa + 4);
        const char* at = smem + sa * SLOT;
        const char* bt = smem + sb * SLOT;
        const bool has_b = t + 1 < nk, has_a = t + 2 < nk;
        // piece q of this K tile's issue order: the LB pieces of B(t+1) first, then the LA pieces of A(t+2)
        auto issue = [&](int q) {
            if (q < LB) { if (has_b) dma_b(q, t + 1, sbn); }
            else if (has_a) dma_a(q - LB, t + 2, san);
        };
        auto wait_next = [&]() {                                  // K tile t + 1 landed; A(t+2) may stay in flight
            if (has_a) wait_vmcnt<LA>();
            else wait_vmcnt<0>();
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t bfr[NI], af[MI];
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) bfr[jj] = frag(bt, wn * TN + jj * 16 + frow, ks * 4 + fch);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = frag(at, wm * TM + i * 16 + frow, ks * 4 + fch);
            if (ks == 0 && has_b) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qq = 0; qq < (NM0 < NP ? NM0 : NP); ++qq) issue(qq);
            }
            if (PP && ks == 1 && has_b && wm == 1) {              // group 1: one barrier behind, waits in its memory half
                if constexpr (TRACE) { const long long a = __builtin_readcyclecounter(); wait_next(); tr_stall += __builtin_readcyclecounter() - a; }
                else wait_next();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PP) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            constexpr int REST = NP - (NM0 < NP ? NM0 : NP);      // pieces issued between the MFMAs of phase (t, 0)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int jj = 0; jj < NI; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj], af[i], acc[i][jj], 0, 0, 0);
                if (REST > 0 && ks == 0 && i < REST && has_b) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(NP - REST + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (ks == 1 && has_b && (wm == 0 || !PP)) {           // group 0 (without ping-pong: every wave): before the barrier its reads follow
                if constexpr (TRACE) { const long long a = __builtin_readcyclecounter(); wait_next(); tr_stall += __builtin_readcyclecounter() - a; }
                else wait_next();
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (PP || ks == 1) __builtin_amdgcn_s_barrier();      // without ping-pong: the one barrier of the K tile
            __builtin_amdgcn_sched_barrier(0);
        }
        sa = wrap(sa + 2);
    }
    if (PP && wm == 0) __builtin_amdgcn_s_barrier();             // pairs with group 1's extra barrier: nobody reads the ring any more
    {
        int em0 = __builtin_amdgcn_readfirstlane(m0), en0 = __builtin_amdgcn_readfirstlane(n0);
        asm volatile("" : "+s"(em0), "+s"(en0));
        if constexpr (TRACE) tr_t2 = __builtin_readcyclecounter();
        pmf16_epilogue<MI, NI, OUT, MASK, EPF>(p, acc, scr, em0, en0, wm, wn, lane);
        if constexpr (TRACE) {
            const long long t3 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t4 = __builtin_readcyclecounter();
            if (lane == 0 && wn == 0) {
                long long* tr = p.trace + ((size_t)(xbase + xslot) * 2 + wm) * 8;
                tr[0] = tr_t1 - tr_t0; tr[1] = tr_t2 - tr_t1; tr[2] = t3 - tr_t2; tr[3] = t4 - t3; tr[4] = tr_stall;
                tr[5] = bid; tr[6] = tr_w0; tr[7] = (long long)wall_clock64();
            }
        }
    }
}

template <int BM, int NM0, int EPF = 1, bool PP = true>
int launch_smf16(rs_ctx* ctx, GemmParams& p, hipStream_t s) {
    constexpr int LDS = 5 * 32768;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + 255) / 256;
    const int nwg = p.tiles_m * p.tiles_n;
    extern std::atomic<int> g_group_m;
    p.group_m = g_group_m.load() > 0 ? g_group_m.load()
              : (p.tiles_n == 4 ? (p.K >= 4096 ? 2 : 6) : (p.K >= 4096 ? 4 : (p.tiles_n <= 8 && p.K <= 2560 ? 16 : 8)));
    p.skew_cycles = 0;
    const int out = (p.flags & RS_GEMM_RESIDUAL) ? 2 : ((p.flags & RS_GEMM_OUT_F32) ? 1 : ((p.flags & RS_GEMM_GLU) ? 3 : 0));
    const bool mask = p.flags & RS_GEMM_ROWMASK;
#define RS_SMF(O, MK, TR)                                                                                         \
    do {                                                                                                          \
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_smf16_kernel<BM, O, MK, NM0, EPF, TR, PP>, LDS); rc != RS_OK) return rc; \
        hipLaunchKernelGGL((gemm_smf16_kernel<BM, O, MK, NM0, EPF, TR, PP>), dim3(nwg), dim3(512), LDS, s, p);   \
    } while (0)
    if (p.trace) {
        if (out == 2 && !mask) RS_SMF(2, false, true);
        else if (out == 0 && !mask) RS_SMF(0, false, true);
        else return rs_fail(ctx, RS_EINVAL, "gemm trace: plain bf16 or residual output only");
        return RS_OK;
    }
    if (out == 2 && !mask) RS_SMF(2, false, false);
    else if (out == 1 && !mask) RS_SMF(1, false, false);
    else if (out == 0 && !mask) RS_SMF(0, false, false);
    else if (out == 0 && mask) RS_SMF(0, true, false);
    else if (out == 3 && !mask) RS_SMF(3, false, false);
    else return rs_fail(ctx, RS_EINVAL, "gemm: row mask with f32 output has no big-tile kernel");
#undef RS_SMF
    return RS_OK;
}

template <int BM, int NM0, int ABL = 0, int EPF = 1>
int launch_lmf16(rs_ctx* ctx, GemmParams& p, hipStream_t s, int grid_cap) {
    constexpr int BN = 256;
    constexpr int LDS = 2 * (BM + BN) * 128 + 8 * 4096;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int nwg = p.tiles_m * p.tiles_n;
    extern std::atomic<int> g_group_m;
    // row panels per XCD tile group (profiles/r02r_gemm_group_m_sweep.txt): N = 1024 (4 weight tiles) likes 2 panels at
    // K = 4096 and 6 below; one-tile-wide problems (the subsampling GEMMs) 16; everything else is flat from 6 up
    p.group_m = g_group_m.load() > 0 ? g_group_m.load()
              : (p.tiles_n == 4 ? (p.K >= 4096 ? 2 : 6) : (p.K >= 4096 ? 4 : (p.tiles_n <= 8 && p.K <= 2560 ? 16 : 8)));
    p.skew_cycles = 0;
    const int grid = nwg < grid_cap ? nwg : grid_cap;
    const int out = (p.flags & RS_GEMM_RESIDUAL) ? 2 : ((p.flags & RS_GEMM_OUT_F32) ? 1 : ((p.flags & RS_GEMM_GLU) ? 3 : 0));
    const bool mask = p.flags & RS_GEMM_ROWMASK;
    if constexpr (ABL != 0) {
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_lmf16_kernel<BM, 0, false, NM0, ABL>, LDS); rc != RS_OK) return rc;
        hipLaunchKernelGGL((gemm_lmf16_kernel<BM, 0, false, NM0, ABL>), dim3(grid), dim3(512), LDS, s, p);
        return RS_OK;
    }
    if (p.trace) {      // debug build of the same kernel that records a per-tile timeline (one tile per workgroup only)
        if (mask || out == 1 || out == 3 || grid != nwg) return rs_fail(ctx, RS_EINVAL, "gemm trace: plain bf16 or residual output, one tile per workgroup");
        if (out == 2) {
            if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_lmf16_kernel<BM, 2, false, NM0, 4, EPF>, LDS); rc != RS_OK) return rc;
            hipLaunchKernelGGL((gemm_lmf16_kernel<BM, 2, false, NM0, 4, EPF>), dim3(grid), dim3(512), LDS, s, p);
        } else {
            if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_lmf16_kernel<BM, 0, false, NM0, 4, EPF>, LDS); rc != RS_OK) return rc;
            hipLaunchKernelGGL((gemm_lmf16_kernel<BM, 0, false, NM0, 4, EPF>), dim3(grid), dim3(512), LDS, s, p);
        }
        return RS_OK;
    }
#define RS_LMF(O, MK)                                                                                         \
    do {                                                                                                      \
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_lmf16_kernel<BM, O, MK, NM0, 0, EPF>, LDS); rc != RS_OK) return rc; \
        hipLaunchKernelGGL((gemm_lmf16_kernel<BM, O, MK, NM0, 0, EPF>), dim3(grid), dim3(512), LDS, s, p);      \
    } while (0)
    if (out == 2 && !mask) RS_LMF(2, false);
    else if (out == 1 && !mask) RS_LMF(1, false);
    else if (out == 0 && !mask) RS_LMF(0, false);
    else if (out == 0 && mask) RS_LMF(0, true);
    else if (out == 3 && !mask) RS_LMF(3, false);
    else return rs_fail(ctx, RS_EINVAL, "gemm: row mask with f32 output has no persistent kernel");
#undef RS_LMF
    return RS_OK;
}

// Process-wide A/B knobs (debug / tuning only; the defaults are the measured winners and nothing in the product
// path writes them).  Atomics initialised once from the environment, so concurrent first launches from the encoder
// thread and the decode worker are safe; they are deliberately not per-context: they select code paths, not state.
extern std::atomic<int> g_skew, g_persistent, g_group_m, g_variant, g_big, g_reserve, g_res_prefetch, g_ring, g_192_pct;
extern std::atomic<long long*> g_trace;
void gemm_knobs_from_env();

template <int BM, int BN, int BK, int NST, int WM, int WN, bool PERSIST, bool RES, bool PP = false>
int launch_variant2(rs_ctx* ctx, GemmParams& p, hipStream_t s) {
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    constexpr int LDS = NST * STAGE_BYTES;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int nwg = p.tiles_m * p.tiles_n;
    // measured (profiles/r01o_gemm_group_m.txt): with operands hot in the 256 MiB infinity cache, 8
    // panels x 4 weight tiles per XCD round beats the plain n-fastest order by 15-19 % (ffn_up 392 ->
    // 318 us); inside the encoder, where A comes from HBM, the gain is 1.7 % (776 vs 784 TF/s).
    // Narrow, short-K problems like 16 panels; K = 4096 (2 MiB per A panel) likes 4.
    p.group_m = g_group_m.load() > 0 ? g_group_m.load() : (p.K >= 4096 ? 4 : (p.tiles_n <= 8 && p.K <= 2560 ? 16 : 8));
    constexpr int CUS = 256;
    const bool persist = PERSIST && nwg > CUS;
    // quarter of one tile's main-loop time at ~1 PF/s in shader cycles (2.4 GHz)
    const double skew_frac = g_skew < 0 ? 0.0 : g_skew / 100.0;   // measured: any start skew loses (profiles/r01_gemm_persistent.txt)
    p.skew_cycles = persist ? (int)(skew_frac * (2.0 * BM * BN * (double)p.K / 1.0e15 * 256.0) * 2.4e9) : 0;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, PERSIST, RES>, LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, false, RES>, LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, false, RES, true>, LDS); rc != RS_OK) return rc;
    if (p.trace)     // debug build of the same kernel that records per-tile timestamps
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, false, RES, true>), dim3(nwg), dim3(64 * WM * WN), LDS, s, p);
    else if (persist)
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, PERSIST, RES>), dim3(CUS), dim3(64 * WM * WN), LDS, s, p);
    else if (PP) {
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, false, RES, false, PP>, LDS); rc != RS_OK) return rc;
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, false, RES, false, PP>), dim3(nwg), dim3(64 * WM * WN), LDS, s, p);
    } else
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, NST, WM, WN, false, RES>), dim3(nwg), dim3(64 * WM * WN), LDS, s, p);
    return RS_OK;
}

template <int BM, int BN, int BK, int NST, int WM, int WN, bool PERSIST = false, bool PP = false>
int launch_variant(rs_ctx* ctx, GemmParams& p, hipStream_t s) {
    if (p.flags & RS_GEMM_RESIDUAL) return launch_variant2<BM, BN, BK, NST, WM, WN, PERSIST, true, PP>(ctx, p, s);
    return launch_variant2<BM, BN, BK, NST, WM, WN, PERSIST, false, PP>(ctx, p, s);
}

template <int BM, int BN>
int launch_mf16(rs_ctx* ctx, GemmParams& p, hipStream_t s) {
    constexpr int LDS = 4 * (BM + BN) * 32 * 2;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int nwg = p.tiles_m * p.tiles_n;
    p.group_m = g_group_m.load() > 0 ? g_group_m.load() : (p.K >= 4096 ? 4 : (p.tiles_n <= 8 && p.K <= 2560 ? 16 : 8));
    p.skew_cycles = 0;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_mf16_kernel<BM, BN, false>, LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)gemm_mf16_kernel<BM, BN, true>, LDS); rc != RS_OK) return rc;
    if (p.flags & RS_GEMM_RESIDUAL) hipLaunchKernelGGL((gemm_mf16_kernel<BM, BN, true>), dim3(nwg), dim3(512), LDS, s, p);
    else hipLaunchKernelGGL((gemm_mf16_kernel<BM, BN, false>), dim3(nwg), dim3(512), LDS, s, p);
    return RS_OK;
}

std::atomic<long long*> g_trace{nullptr};
std::atomic<int> g_variant{0}, g_skew{-1}, g_persistent{2}, g_group_m{0} /* 0 = by shape */, g_big{0};
std::atomic<int> g_reserve{0};   // CUs the persistent kernel leaves free when a context does not say (rs_set_option)
// residual chunks requested ahead by the f32 epilogue: 3 (whole path 64.2 vs 64.5 ms/step with 1 and 65.0 with all 6:
// profiles/r02u_bench_ab.txt; in isolation 6 is the fastest, in the pipeline its 24-load burst per wave is not)
std::atomic<int> g_res_prefetch{3};
std::atomic<int> g_192_pct{88};   // the 192-row tile is chosen when its rounds x height is below this percentage of the 256-row tile's
// split-ring kernel for the big shapes: 2 (default) = B pieces beside the fragment reads, A pieces between the MFMAs;
// 1 = five pieces beside the reads; 0 = the two-K-tile ring (gemm_lmf16_kernel).  Same box, whole path:
// 64.0 -> 61.7 ms/step, ffn_down 314 -> 276 us (profiles/r02x_*)
std::atomic<int> g_ring{2};
void gemm_knobs_from_env() {
    static std::once_flag once;
    std::call_once(once, [] {
        auto env = [](const char* name, std::atomic<int>& v) { if (const char* e = getenv(name)) v = atoi(e); };
        env("RS_GEMM_VARIANT", g_variant);        // force one kernel variant (microbenchmarks); 0 = by shape
        env("RS_GEMM_GROUP_M", g_group_m);        // row panels per XCD tile group; 0 = by shape
        env("RS_GEMM_PERSISTENT", g_persistent);  // 2 (default) = whole-line kernel, one tile per workgroup; 1 = persistent grid; 0 = round-1 kernels
        env("RS_GEMM_RESERVE_CUS", g_reserve);    // CUs the persistent grid leaves to other streams (contexts may override)
        env("RS_GEMM_BIG", g_big);                // big-tile kernel family (DESIGN.md A/B knob table)
        env("RS_GEMM_RES_PREFETCH", g_res_prefetch);   // 3 (default) / 6 / 1 residual chunks in flight in the f32 epilogue
        env("RS_GEMM_192_PCT", g_192_pct);        // tile-height rule (88)
        env("RS_GEMM_RING", g_ring);              // 2 (default) / 1: split-ring kernel (gemm_smf16_kernel); 0: gemm_lmf16_kernel; 5: experimental (no ping-pong)
    });
}

}  // namespace

// tuning hook for A/B runs (scripts/gemm_bench.py); not part of the public header
extern "C" void rs_debug_set_gemm_variant(int v) { gemm_knobs_from_env(); g_variant = v; }
extern "C" void rs_debug_set_gemm_skew(int v) { g_skew = v; }
extern "C" void rs_debug_set_gemm_trace(long long* buf) { g_trace = buf; }
extern "C" void rs_debug_set_gemm_persistent(int v) { gemm_knobs_from_env(); g_persistent = v; }
extern "C" void rs_debug_set_gemm_group_m(int v) { gemm_knobs_from_env(); g_group_m = v; }

// kernel variant for a problem (the numbers are the cases of rs_launch_gemm's switch; RS_GEMM_VARIANT forces one)
static int gemm_pick_variant(const rs_gemm_args& a) {
    gemm_knobs_from_env();
    int v = g_variant;
    if (v == 0) {
        // measured on MI355X (profiles/r01_gemm_variants.txt, r01n_gemm_tile_height.txt): big tiles win
        // whenever they fill the chip at least twice; below that the 128x128 kernels (2-3 workgroups
        // per CU) hide their epilogue better.  All tiles of a launch cost the same and the 256 CUs run
        // them in lockstep rounds, so the tile HEIGHT is picked to minimise rounds x height: at
        // M=35328, N=1024 a 256-row tile needs 3 rounds with the last one 16 % full, a 192-row tile
        // fills 2.9 rounds (ffn_down 413 -> 351 us).
        constexpr long CUS = 256;
        const long tn = (a.N + 255) / 256;
        const long t256 = (long)((a.M + 255) / 256) * tn, t192 = (long)((a.M + 191) / 192) * tn;
        if (a.M < 1024 || a.N < 256) v = 1;
        else if (t256 < 2 * CUS) v = a.K <= 1024 ? 7 : 1;
        else if (g_persistent.load() == 9) v = 9;          // the 32x32x16 persistent experiment of round 1
        else {
            // the 192-row tile is a little less efficient per flop; inside the two-stream pipeline (decode
            // workgroups borrow CUs, so rounds are not exact) it only pays where the round count drops by
            // a quarter (N = 1024: 3 rounds of 256 rows -> 3 of 192), not for 7 -> 6.75 or 5 -> 4.5
            // (profiles/r01q_kernel_stats.txt: qkv 245 -> 273 us, pw1 177 -> 187 us with it)
            const long c192 = ((t192 + CUS - 1) / CUS) * 192, c256 = ((t256 + CUS - 1) / CUS) * 256;
            // (threshold re-tunable: RS_GEMM_192_PCT; with the split ring the 192-row tile lost most of its per-row
            // handicap — 1.2 vs 1.4 us per K tile — so the rule of round 1 may be too strict: DESIGN.md §8)
            v = c192 * 100 < c256 * g_192_pct.load() ? 10 : 2;
        }
        const int big = g_big;    // A/B knob for whole-pipeline runs: remap the 256x256 choice
        // big == 0: the 16x16x32 kernels (whole path 71.7 -> 69.5 ms/step with them: profiles/r01w_*);
        // 1 = the 32x32x16 kernels they replaced; 20 = 32x32x16 ping-pong; 31 = 16x16x32 with 256-row tiles only
        if (big == 0) { if (v == 2) v = 30; else if (v == 10) v = 32; }
        else if (big == 33) { if (v == 10) v = 32; }          // hybrid for A/B: 32x32x16 plain epilogue, 16x16x32 residual
        else if (big == 31) { if (v == 2 || v == 10) v = 30; }
        else if (v == 2 && big > 1) v = big;
        // (variant 9, the persistent tile loop, is ~20 % faster in isolation — profiles/r01_gemm_persistent.txt —
        // but one 128 KiB-LDS workgroup per CU for the whole launch starves the decode stream of the
        // two-stage pipeline; it is selected with rs_debug_set_gemm_persistent(1) / RS_GEMM_PERSISTENT=1
        // for single-stream use)
    }
    // the big-tile choice (30 / 32) is served by the whole-line cross-tile kernel (60 / 62): RS_GEMM_PERSISTENT = 2
    // (default) one tile per workgroup, 1 = persistent grid of (CUs - reserved) workgroups, 0 = the round-1 kernels
    if ((v == 30 || v == 32) && g_variant == 0 && g_persistent.load() != 0 && a.K >= 128 && (a.N % 8) == 0 &&
        (size_t)a.M * a.ldc * ((a.flags & (RS_GEMM_OUT_F32 | RS_GEMM_RESIDUAL)) ? 4 : 2) < (1ull << 31) &&
        (size_t)a.M * a.lda * 2 < (1ull << 32) && (size_t)a.N * a.ldw * 2 < (1ull << 32) &&
        !((a.flags & RS_GEMM_ROWMASK) && (a.flags & (RS_GEMM_OUT_F32 | RS_GEMM_RESIDUAL))))
        v += 30 + (g_persistent.load() == 2 ? 1000 : 0);
    // residual / f32 epilogue of the 192-row tile: deep residual prefetch (RS_GEMM_RES_PREFETCH=1 restores one chunk ahead)
    if (g_variant == 0 && v % 1000 == 62 && g_res_prefetch.load() != 1) v += g_res_prefetch.load() == 3 ? 30 : 20;
    // split-ring kernel (RS_GEMM_RING: 1 = NM0 5, 2 = NM0 4)
    if (g_variant == 0 && g_ring.load() > 0 && v >= 1000) {
        const int k = v % 1000;
        const int r = g_ring.load();       // 5: the experimental no-ping-pong build
        if (k == 60) v = r == 1 ? 1200 : (r == 5 ? 1220 : 1210);
        else if (k == 62 || k == 82 || k == 92) v = r == 1 ? 1202 : (r == 5 ? 1222 : 1212);
    }
    return v;
}

// the GLU epilogue (RS_GEMM_GLU) exists in the whole-line kernel only: 256- / 192-row tiles of 64-column wave tiles
static bool gemm_variant_has_glu(int v) { const int k = v % 1000; return k == 50 || k == 52 || k == 60 || k == 62 || k == 70 || k == 72 || k == 82 || k == 92 || k == 200 || k == 202 || k == 210 || k == 212 || k == 220 || k == 222; }

bool rs_gemm_has_glu(int M, int N, int K) {
    rs_gemm_args a{};
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N / 2; a.flags = RS_GEMM_BIAS | RS_GEMM_GLU;
    return gemm_variant_has_glu(gemm_pick_variant(a));
}

int rs_launch_gemm(rs_ctx* ctx, const rs_gemm_args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return rs_fail(ctx, RS_EINVAL, "gemm: empty shape %d %d %d", a.M, a.N, a.K);
    if (a.K % 64) return rs_fail(ctx, RS_EINVAL, "gemm: K=%d must be a multiple of 64", a.K);
    if (a.N % 4 || a.ldc % 4) return rs_fail(ctx, RS_EINVAL, "gemm: N=%d and ldc=%d must be multiples of 4", a.N, a.ldc);
    if (!(a.flags & RS_GEMM_OUT_F32) && (a.ldc % 8)) return rs_fail(ctx, RS_EINVAL, "gemm: bf16 output needs ldc %% 8 == 0 (got %d)", a.ldc);
    if ((a.lda % 8) || (a.ldw % 8) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) || ((uintptr_t)a.out & 15))
        return rs_fail(ctx, RS_EINVAL, "gemm: operands must be 16-byte aligned (lda %d ldw %d)", a.lda, a.ldw);
    if ((a.flags & RS_GEMM_ROWMASK) && (!a.mask_lens || a.mask_rows_per_step <= 0 || a.mask_steps <= 0))
        return rs_fail(ctx, RS_EINVAL, "gemm: row mask requested without lens");
    if ((a.flags & RS_GEMM_BIAS) && (!a.bias || ((uintptr_t)a.bias & 15)))
        return rs_fail(ctx, RS_EINVAL, "gemm: bias flag without a 16-byte aligned pointer");
    if ((a.flags & RS_GEMM_RESIDUAL) && (!a.residual || ((uintptr_t)a.residual & 15)))
        return rs_fail(ctx, RS_EINVAL, "gemm: residual flag without a 16-byte aligned pointer");
    if (a.flags & RS_GEMM_GLU) {
        if (a.flags & (RS_GEMM_RELU | RS_GEMM_SILU | RS_GEMM_RESIDUAL | RS_GEMM_OUT_F32 | RS_GEMM_ROWMASK))
            return rs_fail(ctx, RS_EINVAL, "gemm: GLU combines with a bias only");
        if ((a.N % 64) || a.alpha != 1.0f) return rs_fail(ctx, RS_EINVAL, "gemm: GLU needs N %% 64 == 0 and alpha == 1 (N=%d)", a.N);
    }
    GemmParams p;
    p.A = a.A; p.W = a.W; p.out = a.out; p.bias = a.bias; p.residual = a.residual; p.mask_lens = a.mask_lens;
    p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.M = a.M; p.N = a.N; p.K = a.K; p.flags = a.flags;
    p.alpha = a.alpha; p.mask_rows_per_step = a.mask_rows_per_step; p.mask_steps = a.mask_steps;
    p.tiles_m = p.tiles_n = 0;
    p.trace = g_trace.load();
    gemm_knobs_from_env();
    const double flops = 2.0 * a.M * (double)a.N * a.K;
    const double bytes = 2.0 * ((double)a.M * a.K + (double)a.N * a.K) +
                         (double)a.M * a.N * ((a.flags & RS_GEMM_OUT_F32) ? 4 : ((a.flags & RS_GEMM_GLU) ? 1 : 2)) +
                         ((a.flags & RS_GEMM_RESIDUAL) ? (double)a.M * a.N * 4 : 0.0);   // residual is read once
    rs_prof_begin(ctx, RS_PROF_GEMM, s, flops, bytes);
    int rc;
    int v = gemm_pick_variant(a);
    // the GLU epilogue lives in the whole-line kernel: problems the heuristics give to the small-tile kernels run
    // on its 256-row tile instead (correct for every M; the encoder only asks for it where it is the natural choice)
    if ((a.flags & RS_GEMM_GLU) && !gemm_variant_has_glu(v)) {
        if (a.K < 128) return rs_fail(ctx, RS_EINVAL, "gemm: the GLU epilogue needs K >= 128 (K=%d)", a.K);
        v = 1060;
    }
    if (ctx->n_cus <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
        ctx->n_cus = n;
    }
    const int reserve = ctx->gemm_reserved_cus >= 0 ? ctx->gemm_reserved_cus : g_reserve.load();
    const int grid_cap = ctx->n_cus - reserve > 8 ? ctx->n_cus - reserve : 8;
    // 50-72: the whole-line cross-tile kernel, persistent grid; +1000: the same kernel, one tile per workgroup
    const int pgrid = v >= 1000 ? (1 << 30) : grid_cap;
    switch (v % 1000) {
        // 5x / 6x / 7x: the whole-line kernel with 8 / 5 / 3 of a K tile's DMAs issued beside the fragment reads
        case 50: rc = launch_lmf16<256, 8>(ctx, p, s, pgrid); break;
        case 52: rc = launch_lmf16<192, 8>(ctx, p, s, pgrid); break;
        case 60: rc = launch_lmf16<256, 5>(ctx, p, s, pgrid); break;
        case 61: rc = launch_lmf16<256, 5, 1>(ctx, p, s, pgrid); break;          // ablations of 60 (wrong results)
        case 63: rc = launch_lmf16<256, 5, 2>(ctx, p, s, pgrid); break;
        case 65: rc = launch_lmf16<256, 5, 3>(ctx, p, s, pgrid); break;
        case 62: rc = launch_lmf16<192, 5>(ctx, p, s, pgrid); break;
        // 82 / 92: 62 with every / three of the six residual chunks of the f32 epilogue requested up front
        case 82: rc = launch_lmf16<192, 5, 0, 6>(ctx, p, s, pgrid); break;
        case 92: rc = launch_lmf16<192, 5, 0, 3>(ctx, p, s, pgrid); break;
        // 20x: the split-ring kernel (B(t+1) first, A(t+2) a K tile further ahead), one tile per workgroup
        case 200: rc = launch_smf16<256, 5>(ctx, p, s); break;
        case 202: rc = launch_smf16<192, 5, 3>(ctx, p, s); break;
        case 210: rc = launch_smf16<256, 4>(ctx, p, s); break;      // only the B pieces beside the fragment reads
        case 212: rc = launch_smf16<192, 4, 3>(ctx, p, s); break;
        case 220: rc = launch_smf16<256, 4, 1, false>(ctx, p, s); break;   // EXPERIMENTAL: no ping-pong, one barrier per K tile
        case 222: rc = launch_smf16<192, 4, 3, false>(ctx, p, s); break;
        case 70: rc = launch_lmf16<256, 3>(ctx, p, s, pgrid); break;
        case 72: rc = launch_lmf16<192, 3>(ctx, p, s, pgrid); break;
        case 1: rc = launch_variant<128, 128, 64, 2, 2, 2>(ctx, p, s); break;   // small problems
        case 2: rc = launch_variant<256, 256, 64, 2, 2, 4>(ctx, p, s); break;   // big tile, drain per K step
        case 3: rc = launch_variant<256, 256, 32, 4, 2, 4>(ctx, p, s); break;   // big tile, 4-stage ring, counted vmcnt
        case 7: rc = launch_variant<128, 128, 32, 3, 2, 2>(ctx, p, s); break;   // 3 WGs per CU
        case 9: rc = launch_variant<256, 256, 64, 2, 2, 4, true>(ctx, p, s); break;    // persistent, one WG per CU
        case 10: rc = launch_variant<192, 256, 64, 2, 2, 4>(ctx, p, s); break;  // 3/4 tile: fewer idle CUs in the last round
        // ping-pong wave groups: +14-17 % over case 2 in ~10 ms bursts, identical in the sustained regime (the
        // package sits at its ~1.4 kW cap and the clock drops from 2.04 to 1.94 GHz instead:
        // profiles/r01t_gemm_sustained_power.txt); opt-in with RS_GEMM_BIG=20
        case 20: rc = launch_variant<256, 256, 32, 4, 2, 4, false, true>(ctx, p, s); break;
        case 30: rc = launch_mf16<256, 256>(ctx, p, s); break;   // 16x16x32 MFMA, ping-pong wave groups
        case 32: rc = launch_mf16<192, 256>(ctx, p, s); break;   // the same with the 3/4-height tile
        default: rc = rs_fail(ctx, RS_EINVAL, "gemm: unknown RS_GEMM_VARIANT %d", v);
    }
    rs_prof_end(ctx, RS_PROF_GEMM, s);
    if (rc != RS_OK) return rc;
    RS_CHECK_LAUNCH(ctx, "gemm_bf16");
    return RS_OK;
}

// k_gemm_bf16.hip — bf16 x bf16 -> f32 MFMA GEMM with fused epilogue for gfx950.
//
//   out[M][N] = epilogue( A[M][K] . W[N][K]^T )        A, W row-major bf16 (K contiguous)
//
// This one kernel carries every dense contraction of the encoder (SURVEY.md §8a rows S2-S4,
// L2, L3, L6 pw1/pw2, D1): 96.8 % of the path's FLOPs.  Structure:
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
//     2x2 blocks of v_mfma_f32_32x32x16_bf16), BK = 64.
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double
//     buffered: the DMA of K-tile t+1 is in flight while tile t is multiplied.
//   * LDS rows are 128 B (64 bf16); 16-B chunks are XOR-swizzled with ((row>>1)&7) so the
//     ds_read_b128 fragment reads (32 rows x one chunk per half-wave) hit 16 distinct
//     4-bank slots per 16-lane group: conflict free.  global_load_lds writes LDS linearly,
//     so the swizzle is applied to the per-lane SOURCE address and again on the read.
//   * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs (private L2) walks a
//     contiguous run of tiles, n-fastest, so an XCD reads each A row-panel once.
//   * epilogue fused in registers: +bias, ReLU/SiLU, *alpha, +residual (f32), per-utterance
//     row mask, store bf16 or f32.
#include "rs_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmParams {
    const uint16_t* A; const uint16_t* W; void* out;
    const float* bias; const float* residual; const int32_t* mask_lens;
    int lda, ldw, ldc, M, N, K, flags;
    float alpha;
    int mask_rows_per_step, mask_steps;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ base, int ld, int row0, int max_row,
                                           int k0, char* lds_tile, int wave, int lane) {
    // 16 wave-instructions cover 128 rows x 128 B; wave w issues 4 of them (8 rows each).
    const int r = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + r;
        const int c = pc ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < max_row ? grow : max_row - 1;
        const uint16_t* src = base + (size_t)grow * ld + k0 + c * 8;
        char* dst = lds_tile + (wave * 32 + j * 8) * 128;  // wave-uniform; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8_t read_frag(const char* lds_tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8_t*>(lds_tile + off);
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                   // [2][TILE_BYTES]
    char* Bs = smem + 2 * TILE_BYTES;  // [2][TILE_BYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective remap (blocks b, b+8, b+16.. share an XCD)
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / BK;
    stage_tile(p.A, p.lda, m0, p.M, 0, As, wave, lane);
    stage_tile(p.W, p.ldw, n0, p.N, 0, Bs, wave, lane);

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int t = 0; t < nk; ++t) {
        __syncthreads();  // tile t landed (vmcnt(0) inside) and buffer (t+1)&1 is free
        const int cur = t & 1;
        if (t + 1 < nk) {
            stage_tile(p.A, p.lda, m0, p.M, (t + 1) * BK, As + (cur ^ 1) * TILE_BYTES, wave, lane);
            stage_tile(p.W, p.ldw, n0, p.N, (t + 1) * BK, Bs + (cur ^ 1) * TILE_BYTES, wave, lane);
        }
        const char* at = As + cur * TILE_BYTES;
        const char* bt = Bs + cur * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = read_frag(at, wm * 64 + i * 32 + frow, ks * 2 + fhalf);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = read_frag(bt, wn * 64 + j * 32 + frow, ks * 2 + fhalf);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int flags = p.flags;
    const bool has_bias = flags & RS_GEMM_BIAS, relu = flags & RS_GEMM_RELU, silu = flags & RS_GEMM_SILU;
    const bool has_res = flags & RS_GEMM_RESIDUAL, out_f32 = flags & RS_GEMM_OUT_F32;
    const bool rowmask = flags & RS_GEMM_ROWMASK;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        const bool n_ok = n < p.N;
        const float bv = (has_bias && n_ok) ? p.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (!n_ok || m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (relu) v = fmaxf(v, 0.0f);
                if (silu) v = silu_f(v);
                v *= p.alpha;
                const size_t o = (size_t)m * p.ldc + n;
                if (has_res) v += p.residual[o];
                if (rowmask) {
                    const int step = m / p.mask_rows_per_step;
                    const int b = step / p.mask_steps;
                    if (step - b * p.mask_steps >= p.mask_lens[b]) v = 0.0f;
                }
                if (out_f32) reinterpret_cast<float*>(p.out)[o] = v;
                else reinterpret_cast<uint16_t*>(p.out)[o] = f32_to_bf16(v);
            }
        }
    }
}

}  // namespace

int rs_launch_gemm(rs_ctx* ctx, const rs_gemm_args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return rs_fail(ctx, RS_EINVAL, "gemm: empty shape %d %d %d", a.M, a.N, a.K);
    if (a.K % BK) return rs_fail(ctx, RS_EINVAL, "gemm: K=%d must be a multiple of %d", a.K, BK);
    if ((a.lda % 8) || (a.ldw % 8) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15))
        return rs_fail(ctx, RS_EINVAL, "gemm: operands must be 16-byte aligned (lda %d ldw %d)", a.lda, a.ldw);
    if ((a.flags & RS_GEMM_ROWMASK) && (!a.mask_lens || a.mask_rows_per_step <= 0 || a.mask_steps <= 0))
        return rs_fail(ctx, RS_EINVAL, "gemm: row mask requested without lens");
    if ((a.flags & RS_GEMM_BIAS) && !a.bias) return rs_fail(ctx, RS_EINVAL, "gemm: bias flag without pointer");
    if ((a.flags & RS_GEMM_RESIDUAL) && !a.residual) return rs_fail(ctx, RS_EINVAL, "gemm: residual flag without pointer");
    GemmParams p;
    p.A = a.A; p.W = a.W; p.out = a.out; p.bias = a.bias; p.residual = a.residual; p.mask_lens = a.mask_lens;
    p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.M = a.M; p.N = a.N; p.K = a.K; p.flags = a.flags;
    p.alpha = a.alpha; p.mask_rows_per_step = a.mask_rows_per_step; p.mask_steps = a.mask_steps;
    p.tiles_m = (a.M + BM - 1) / BM; p.tiles_n = (a.N + BN - 1) / BN;
    const int nwg = p.tiles_m * p.tiles_n;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
        attr_set = true;
    }
    const double flops = 2.0 * a.M * (double)a.N * a.K;
    const double bytes = 2.0 * ((double)a.M * a.K + (double)a.N * a.K) +
                         (double)a.M * a.N * ((a.flags & RS_GEMM_OUT_F32) ? 4 : 2);
    rs_prof_begin(ctx, RS_PROF_GEMM, s, flops, bytes);
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3(nwg), dim3(NTHREADS), 4 * TILE_BYTES, s, p);
    rs_prof_end(ctx, RS_PROF_GEMM, s);
    RS_CHECK_LAUNCH(ctx, "gemm_bf16");
    return RS_OK;
}

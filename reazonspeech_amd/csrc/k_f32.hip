// k_f32.hip — the float32 PARITY MODE of the encoder (rs_set_option "precision_f32").
//
// The reference runs this model in float32 with no autocast (pkg/nemo-asr/src/transcribe.py:26-28, :48-53).  The
// throughput mode of this library feeds bf16 operands to the matrix cores; this file is the same encoder with float32
// weights, float32 activations and float32 arithmetic end to end, so that "greedy token ids identical to the float32
// reference" is a statement about the whole path and not only about the decode loop:
//
//   gemm_f32_kernel        every dense contraction on v_mfma_f32_16x16x4_f32 (an exact f32 fma chain; 157 TF/s peak), one
//                          fixed summation order per output element (ascending 16-blocks, inside a block k = e + 4 kk),
//                          independent of M and of the tile position: batch-invariant like the bf16 family
//   attention_f32_kernel   rel-pos attention, one wave per query row, float32 scores / softmax / PV
//   glu_dwconv_silu_f32    conv-module middle with IEEE exp / divide
//   (LayerNorm: k_layernorm.hip's kernels already compute and can store float32; subsampling: the float32 instantiations
//    of k_subsample.hip)
//
// Speed is not the object (about 1 s per batch of 256 x 10 s against 55 ms in the throughput mode); the kernels are kept
// simple enough to be read against oracle/model.py line by line.
#include <stdlib.h>

#include "rs_common.h"

int rs_launch_sub_conv0_dw1_f32(rs_ctx* ctx, const float* feats, const int32_t* lens_stage, int B, int t_max, int T2, int F2,
                                float* out, hipStream_t s);
int rs_launch_sub_dw_f32(rs_ctx* ctx, const float* in, const float* w, const float* b, const int32_t* lens_out, int B, int t_in,
                         int f_in, int t_out, int f_out, float* out, hipStream_t s);

namespace {

// IEEE forms (the throughput mode uses v_exp / v_rcp approximations: rs_common.h sigmoid_f)
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }     // torch's exact GELU

struct GemmF32 {
    const float* A; const float* W; float* out;
    const float* bias; const float* residual; const int32_t* mask_lens;
    int lda, ldw, ldc, M, N, K, flags;
    float alpha;
    int mask_rows_per_step, mask_steps;
    // CONV instantiations: A is a channels-last map [n][H][W][C] and row m = (n, oh, ow) of the product is its 3 x 3 patch with padding 1
    // and stride cv_stride in (kh, kw, c) order (K = 9 C, C % 32 == 0), gathered by the loader instead of being written out first
    int cv_H, cv_W, cv_C, cv_OH, cv_OW, cv_stride;
    // per-column affine + PReLU epilogue (a folded inference BatchNorm): v = fma(v, col_scale[n], col_shift[n]) (+ residual), then
    // v < 0 ? prelu[n] * v : v when prelu is given; replaces the bias / activation / alpha steps when col_scale != nullptr
    const float* col_scale; const float* col_shift; const float* prelu;
};

constexpr int GT = 128;   // tile rows / columns
constexpr int GK = 32;    // K depth of a stage
constexpr int GP = 36;    // LDS row pitch in floats (144 B: 16-byte aligned, rows 8 apart share a bank group)

// out[M][N] = epilogue(A[M][K] . W[N][K]^T), all float32.  256 threads = 2 x 2 waves, a wave owns 64 x 64 outputs
// (4 x 4 MFMA blocks).  The weight fragment is the MFMA A operand, so a lane's four accumulator registers are four
// CONSECUTIVE columns of one output row: bias / residual / store are float4 accesses.
// N64: tiles of 128 rows x 64 columns, the four waves stacked over the rows (32 x 64 each) — for products with 64 output columns
// (the first ResNet stage of the AV-HuBERT video trunk) whose 128-column tiles would multiply a clamped copy of the weights half the
// time.  CONV: see GemmF32.  Neither changes the order in which the K products of an output element are added.
// X3: every float32 product as THREE bf16 matrix-core terms — x = hi + lo with hi = bf16(x), lo = bf16(x - hi), 16 mantissa bits per
// operand, a . b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b (the dropped lo_a lo_b is 2^-16 of the product), accumulated in float32 on
// v_mfma_f32_16x16x32_bf16.  gfx950 has no tf32 / xf32 matrix instruction, and the exact v_mfma_f32_16x16x4_f32 peaks at 157 TF/s;
// three bf16 terms run at a third of 2.5 PF/s.  The operands are split once per K stage by the loader (on their way from the prefetch
// registers into LDS, which holds a hi and a lo plane per tile); loader, gather, tiling and epilogue are the exact kernel's.  NOT
// bit-faithful to an IEEE float32 chain (relative error of a product <= 2^-16, of a K-term sum far less): the AV-HuBERT family's option
// products="x3" and the "fp32x3" precision of the transducer families — a mode of its own next to the exact "fp32", measured against
// the same goldens (every id of the three 256-row goldens identical: profiles/r06_19_*), never silently substituted for it.
__device__ __forceinline__ void split_bf16x3(const float4& v, u16x4_t& hi, u16x4_t& lo) {
    hi = pack_bf16x4(v.x, v.y, v.z, v.w);
    lo = pack_bf16x4(v.x - bf16_to_f32(hi[0]), v.y - bf16_to_f32(hi[1]), v.z - bf16_to_f32(hi[2]), v.w - bf16_to_f32(hi[3]));
}
constexpr int XP = 40;    // X3: LDS row pitch of a plane in bf16 (80 B: 16-byte aligned, the 16 rows of a fragment read hit 16 different slots)

template <bool CONV, bool N64, bool X3 = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmF32 p) {
    // Two LDS stages (round 6): the next K stage is fetched into registers while this one is multiplied and written into the OTHER
    // buffer afterwards, so a stage costs one workgroup barrier instead of two; two workgroups per CU (launch bound) leave the
    // prefetch registers in VGPRs — at three the compiler parked them in scratch on their way to LDS (profiles/r06_05_*: 144 bytes
    // of scratch per lane).  The arithmetic per output element is unchanged: ascending 16-blocks, inside a block k = e + 4 kk.
    constexpr int TN = N64 ? 64 : GT, MI = N64 ? 2 : 4;
    // (X3: a buffer is a hi plane followed by a lo plane of [rows][XP] bf16 = rows * XP floats)
    __shared__ __attribute__((aligned(16))) float As[2][X3 ? GT * XP : GT * GP];
    __shared__ __attribute__((aligned(16))) float Ws[2][X3 ? TN * XP : TN * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = N64 ? wave : wave >> 1, wn = N64 ? 0 : wave & 1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * TN;
    const int lr = tid >> 3, lc = tid & 7;            // staging: row lr + 32 i, 16-byte chunk lc of the 128-byte K slice
    // (macros, not lambdas: with the prefetch registers captured by reference the arrays stayed stack objects — 128 bytes of scratch
    // traffic per lane and K stage)
    float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    const float* a_ptr[4];
    const float* w_ptr[4];
    int cv_ih[4], cv_iw[4];                           // CONV: top-left input pixel of the row's patch (may be -1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + lr + 32 * i, n = n0 + lr + 32 * i;
        m = m < p.M ? m : p.M - 1;                    // rows past the matrix: clamped, their products are never stored
        n = n < p.N ? n : p.N - 1;
        if constexpr (CONV) {
            const int ow = m % p.cv_OW, r = m / p.cv_OW, oh = r % p.cv_OH, img = r / p.cv_OH;
            cv_ih[i] = p.cv_stride * oh - 1;
            cv_iw[i] = p.cv_stride * ow - 1;
            a_ptr[i] = p.A + (size_t)img * p.cv_H * p.cv_W * p.cv_C + 4 * lc;
        } else {
            cv_ih[i] = cv_iw[i] = 0;
            a_ptr[i] = p.A + (size_t)m * p.lda + 4 * lc;
        }
        w_ptr[i] = p.W + (size_t)n * p.ldw + 4 * lc;
    }
    // CONV: a K stage of 32 lies inside one tap (C % 32 == 0); a pixel outside the map reads a clamped address and is replaced by zeros
    // (an unconditional load + select: a predicated definition of the prefetch registers is what the compiler demotes to scratch)
#define RS_F32_ALOAD(R, I, k0)                                                                                  \
    do {                                                                                                       \
        if constexpr (CONV) {                                                                                  \
            const int tap = (k0) / p.cv_C, c0 = (k0) - tap * p.cv_C, kh = tap / 3, kw = tap - 3 * kh;         \
            const int ih = cv_ih[I] + kh, iw = cv_iw[I] + kw;                                                  \
            const bool ok = (unsigned)ih < (unsigned)p.cv_H && (unsigned)iw < (unsigned)p.cv_W;                \
            const float4 t = *reinterpret_cast<const float4*>(a_ptr[I] + ((size_t)((ok ? ih : 0) * p.cv_W + (ok ? iw : 0)) * p.cv_C + c0)); \
            R = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);                                                      \
        } else {                                                                                               \
            R = *reinterpret_cast<const float4*>(a_ptr[I] + (k0));                                             \
        }                                                                                                      \
    } while (0)
#define RS_F32_GLOAD(k0)                                                                                        \
    do {                                                                                                       \
        RS_F32_ALOAD(ra0, 0, k0); rw0 = *reinterpret_cast<const float4*>(w_ptr[0] + (k0));                     \
        RS_F32_ALOAD(ra1, 1, k0); rw1 = *reinterpret_cast<const float4*>(w_ptr[1] + (k0));                     \
        RS_F32_ALOAD(ra2, 2, k0); if constexpr (!N64) rw2 = *reinterpret_cast<const float4*>(w_ptr[2] + (k0)); \
        RS_F32_ALOAD(ra3, 3, k0); if constexpr (!N64) rw3 = *reinterpret_cast<const float4*>(w_ptr[3] + (k0)); \
    } while (0)
#define RS_F32_STASH(buf)                                                                                       \
    do {                                                                                                       \
        if constexpr (X3) {                                                                                    \
            uint16_t* ah = reinterpret_cast<uint16_t*>(As[buf]) + lr * XP + 4 * lc;                            \
            uint16_t* wh = reinterpret_cast<uint16_t*>(Ws[buf]) + lr * XP + 4 * lc;                            \
            u16x4_t h, l;                                                                                      \
            split_bf16x3(ra0, h, l); *reinterpret_cast<u16x4_t*>(ah) = h; *reinterpret_cast<u16x4_t*>(ah + GT * XP) = l; \
            split_bf16x3(ra1, h, l); *reinterpret_cast<u16x4_t*>(ah + 32 * XP) = h; *reinterpret_cast<u16x4_t*>(ah + 32 * XP + GT * XP) = l; \
            split_bf16x3(ra2, h, l); *reinterpret_cast<u16x4_t*>(ah + 64 * XP) = h; *reinterpret_cast<u16x4_t*>(ah + 64 * XP + GT * XP) = l; \
            split_bf16x3(ra3, h, l); *reinterpret_cast<u16x4_t*>(ah + 96 * XP) = h; *reinterpret_cast<u16x4_t*>(ah + 96 * XP + GT * XP) = l; \
            split_bf16x3(rw0, h, l); *reinterpret_cast<u16x4_t*>(wh) = h; *reinterpret_cast<u16x4_t*>(wh + TN * XP) = l; \
            split_bf16x3(rw1, h, l); *reinterpret_cast<u16x4_t*>(wh + 32 * XP) = h; *reinterpret_cast<u16x4_t*>(wh + 32 * XP + TN * XP) = l; \
            if constexpr (!N64) {                                                                              \
                split_bf16x3(rw2, h, l); *reinterpret_cast<u16x4_t*>(wh + 64 * XP) = h; *reinterpret_cast<u16x4_t*>(wh + 64 * XP + TN * XP) = l; \
                split_bf16x3(rw3, h, l); *reinterpret_cast<u16x4_t*>(wh + 96 * XP) = h; *reinterpret_cast<u16x4_t*>(wh + 96 * XP + TN * XP) = l; \
            }                                                                                                  \
        } else {                                                                                               \
        float* ad = As[buf] + lr * GP + 4 * lc;                                                               \
        float* wd = Ws[buf] + lr * GP + 4 * lc;                                                               \
        *reinterpret_cast<float4*>(ad) = ra0; *reinterpret_cast<float4*>(ad + 32 * GP) = ra1;                 \
        *reinterpret_cast<float4*>(ad + 64 * GP) = ra2; *reinterpret_cast<float4*>(ad + 96 * GP) = ra3;       \
        *reinterpret_cast<float4*>(wd) = rw0; *reinterpret_cast<float4*>(wd + 32 * GP) = rw1;                 \
        if constexpr (!N64) { *reinterpret_cast<float4*>(wd + 64 * GP) = rw2; *reinterpret_cast<float4*>(wd + 96 * GP) = rw3; } \
        }                                                                                                      \
    } while (0)
    f32x4_t acc[4][MI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    RS_F32_GLOAD(0);
    RS_F32_STASH(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < p.K; k0 += GK) {
        const bool more = k0 + GK < p.K;
        if (more) RS_F32_GLOAD(k0 + GK);
        if constexpr (X3) {
            // one 32-deep step: three bf16 products per output block, the two small terms first
            const uint16_t* Ah = reinterpret_cast<const uint16_t*>(As[cur]);
            const uint16_t* Wh = reinterpret_cast<const uint16_t*>(Ws[cur]);
            bf16x8_t ah[MI], al[MI], wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const uint16_t* q = Ah + (wm * 16 * MI + i * 16 + fr) * XP + 8 * fq;
                ah[i] = *reinterpret_cast<const bf16x8_t*>(q);
                al[i] = *reinterpret_cast<const bf16x8_t*>(q + GT * XP);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint16_t* q = Wh + (wn * 64 + i * 16 + fr) * XP + 8 * fq;
                wh[i] = *reinterpret_cast<const bf16x8_t*>(q);
                wl[i] = *reinterpret_cast<const bf16x8_t*>(q + TN * XP);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ni], ah[mi], acc[ni][mi], 0, 0, 0);
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], al[mi], acc[ni][mi], 0, 0, 0);
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], ah[mi], acc[ni][mi], 0, 0, 0);
                }
        } else {
        const float* Ab = As[cur];
        const float* Wb = Ws[cur];
#pragma unroll
        for (int kb = 0; kb < GK / 16; ++kb) {
            float4 af[MI], wf[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + (wm * 16 * MI + i * 16 + fr) * GP + kb * 16 + 4 * fq);
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const float4*>(Wb + (wn * 64 + i * 16 + fr) * GP + kb * 16 + 4 * fq);
            // step e of a 16-block multiplies k = 16 kb + 4 kk + e for the four lane groups kk at once
#define RS_F32_STEP(E)                                                                                          \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                   \
                _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                              \
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni].E, af[mi].E, acc[ni][mi], 0, 0, 0);
            RS_F32_STEP(x) RS_F32_STEP(y) RS_F32_STEP(z) RS_F32_STEP(w)
#undef RS_F32_STEP
        }
        }
        if (more) RS_F32_STASH(cur ^ 1);              // nobody reads that buffer: its readers passed the barrier that ended the previous stage
        __syncthreads();
        cur ^= 1;
    }
#undef RS_F32_GLOAD
#undef RS_F32_ALOAD
#undef RS_F32_STASH
    // epilogue, the order of k_gemm_bf16.hip: + bias, activation, * alpha, + residual, row mask
    const bool has_bias = p.flags & RS_GEMM_BIAS, relu = p.flags & RS_GEMM_RELU, silu = p.flags & RS_GEMM_SILU;
    const bool res = p.flags & RS_GEMM_RESIDUAL, rowmask = p.flags & RS_GEMM_ROWMASK;
    const bool swl = p.flags & RS_GEMM_SWOOSHL, swr = p.flags & RS_GEMM_SWOOSHR, gelu = p.flags & RS_GEMM_GELU;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * 16 * MI + mi * 16 + fr;
        if (m >= p.M) continue;
        bool keep = true;
        if (rowmask) {
            const int step = m / p.mask_rows_per_step;
            const int b = step / p.mask_steps;
            keep = step - b * p.mask_steps < p.mask_lens[b];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + 4 * fq;
            if (n >= p.N) continue;
            float4 v = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
            if (p.col_scale) {                            // folded BatchNorm (+ residual) (+ PReLU): the arithmetic of k_avsr.hip's avsr_bn_act_kernel
                const float4 a = *reinterpret_cast<const float4*>(p.col_scale + n), bb = *reinterpret_cast<const float4*>(p.col_shift + n);
                v.x = fmaf(v.x, a.x, bb.x); v.y = fmaf(v.y, a.y, bb.y); v.z = fmaf(v.z, a.z, bb.z); v.w = fmaf(v.w, a.w, bb.w);
                if (res) {
                    const float4 r = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.ldc + n);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (p.prelu) {
                    const float4 sl = *reinterpret_cast<const float4*>(p.prelu + n);
                    v.x = v.x >= 0.f ? v.x : sl.x * v.x; v.y = v.y >= 0.f ? v.y : sl.y * v.y;
                    v.z = v.z >= 0.f ? v.z : sl.z * v.z; v.w = v.w >= 0.f ? v.w : sl.w * v.w;
                }
                *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldc + n) = v;
                continue;
            }
            if (has_bias) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (silu) { v.x = silu_exact(v.x); v.y = silu_exact(v.y); v.z = silu_exact(v.z); v.w = silu_exact(v.w); }
            if (gelu) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
            if (swl) { v.x = swoosh_l_exact(v.x); v.y = swoosh_l_exact(v.y); v.z = swoosh_l_exact(v.z); v.w = swoosh_l_exact(v.w); }
            if (swr) { v.x = swoosh_r_exact(v.x); v.y = swoosh_r_exact(v.y); v.z = swoosh_r_exact(v.z); v.w = swoosh_r_exact(v.w); }
            v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
            if (res) {
                const float4 r = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.ldc + n);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (!keep) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldc + n) = v;
        }
    }
}

// out[M][N] = epilogue(A[M][K] . W[N][K]^T) for a FEW rows (M <= 128: the hypothesis rows of an autoregressive decoder step, k_avsr.hip).
// gemm_f32_kernel gives such a product N / 128 workgroups that each walk all of K alone — 6 workgroups and 24 dependent global-load
// round trips for a 768 x 768 layer, ~125 us per launch in profiles/r06_06_f32_avsr_*.  Here a workgroup owns 16 output columns, its
// eight waves split K into eight contiguous runs of 16-blocks (weights and rows straight from global memory / L2 as 16-byte pieces,
// v_mfma_f32_16x16x4_f32, the weight fragment first: a lane holds four consecutive columns of one row), and the partial sums are
// added in LDS in wave order 0 + 1 + .. + 7.  N / 16 workgroups, K / 128 blocks per wave.  The summation order differs from
// gemm_f32_kernel's (K is cut in four): callers that promise bit-identical results across batch sizes must not mix the two.
// K % 16 == 0, N % 4 == 0.  grid (ceil(N / 16)), block 512
constexpr int SK_WAVES = 8;      // waves of a skinny workgroup = the contiguous runs K is cut into
template <int MT>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_f32_skinny_kernel(GemmF32 p) {
    __shared__ __attribute__((aligned(16))) float part[SK_WAVES - 1][MT][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nblk = p.K / 16, per = (nblk + SK_WAVES - 1) / SK_WAVES;
    const int s_lo = wave * per, s_hi = s_lo + per < nblk ? s_lo + per : nblk;
    int nrow = n0 + li;
    nrow = nrow < p.N ? nrow : p.N - 1;
    const float* wp = p.W + (size_t)nrow * p.ldw + 4 * kq;
    const float* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = 16 * t + li;
        m = m < p.M ? m : p.M - 1;
        ap[t] = p.A + (size_t)m * p.lda + 4 * kq;
    }
    f32x4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // three 16-blocks per trip, every load of a trip issued before its first product: the loop is a chain of L2 round trips
    // (29 us per launch with one block per trip: profiles/r06_12_*), so fewer and fatter trips and twice the waves
    constexpr int U = 3;
    for (int S = s_lo; S < s_hi; S += U) {
        float4 wf[U], af[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int Su = S + u < s_hi ? S + u : s_hi - 1;           // (wave-uniform) a block past the run: read again, multiplied by zeros
            wf[u] = *reinterpret_cast<const float4*>(wp + 16 * Su);
            if (S + u >= s_hi) wf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < MT; ++t) af[u][t] = *reinterpret_cast<const float4*>(ap[t] + 16 * Su);
        }
#define RS_SK_STEP(u, E)                                                                           \
        _Pragma("unroll") for (int t = 0; t < MT; ++t)                                            \
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u].E, af[u][t].E, acc[t], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (S + u < s_hi) { RS_SK_STEP(u, x) RS_SK_STEP(u, y) RS_SK_STEP(u, z) RS_SK_STEP(u, w) }
        }
#undef RS_SK_STEP
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) *reinterpret_cast<f32x4_t*>(part[wave - 1][t][lane]) = acc[t];
    }
    __syncthreads();
    if (wave > 0) return;
    const bool has_bias = p.flags & RS_GEMM_BIAS, relu = p.flags & RS_GEMM_RELU, silu = p.flags & RS_GEMM_SILU, gelu = p.flags & RS_GEMM_GELU;
    const bool res = p.flags & RS_GEMM_RESIDUAL;
    const int n = n0 + 4 * kq;
    if (n >= p.N) return;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_bias) bb = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + li;
        if (m >= p.M) continue;
        f32x4_t a = acc[t];
#pragma unroll
        for (int w = 0; w < SK_WAVES - 1; ++w) {
            const f32x4_t q = *reinterpret_cast<const f32x4_t*>(part[w][t][lane]);
            a[0] += q[0]; a[1] += q[1]; a[2] += q[2]; a[3] += q[3];
        }
        float4 v = make_float4(a[0] + bb.x, a[1] + bb.y, a[2] + bb.z, a[3] + bb.w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (silu) { v.x = silu_exact(v.x); v.y = silu_exact(v.y); v.z = silu_exact(v.z); v.w = silu_exact(v.w); }
        if (gelu) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
        if (res) {
            const float4 r = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.ldc + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldc + n) = v;
    }
}

// Rel-pos attention in float32 (oracle/model.py: attention_core).  One wave per query row; a lane holds elements
// lane, lane + 64, .. of the head dimension.  Two passes over the visible keys (maximum, then exp / sum / PV), the
// scores are recomputed in the second pass.
//   s[i][j] = ((q_i + u) . k_j + (q_i + v) . p[j - i + T - 1]) * scale;  masked keys weigh 0, padded queries -> 0
template <int NV>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ pos,
                                                            const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                            const int32_t* __restrict__ lens, float* __restrict__ out, int T,
                                                            int d, int hd, int att_left, int att_right, int n_global, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
    if (i >= T) return;
    int len = lens[b];
    len = len < T ? len : T;
    float* orow = out + ((size_t)b * T + i) * d + h * hd;
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ok[v] = lane + 64 * v < hd;
    if (i >= len) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (ok[v]) orow[lane + 64 * v] = 0.0f;
        return;
    }
    const size_t ld = 3 * (size_t)d;
    const float* base = qkv + (size_t)b * T * ld + h * hd;
    float qu[NV], qv[NV], acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int e = lane + 64 * v;
        const float q = ok[v] ? base[(size_t)i * ld + e] : 0.0f;
        qu[v] = ok[v] ? q + bias_u[h * hd + e] : 0.0f;
        qv[v] = ok[v] ? q + bias_v[h * hd + e] : 0.0f;
        acc[v] = 0.0f;
    }
    const bool window = att_left >= 0 || att_right >= 0;
    const int left = att_left >= 0 ? att_left : T, right = att_right >= 0 ? att_right : T;
    auto visible = [&](int j) -> bool {
        if (!window) return true;
        return ((i - j) <= left && (j - i) <= right) || i < n_global || j < n_global;
    };
    auto score = [&](int j) -> float {
        const float* kr = base + (size_t)j * ld + d;
        const float* pr = pos + (size_t)(j - i + T - 1) * d + h * hd;
        float ac = 0.0f, bd = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (ok[v]) {
                ac = fmaf(qu[v], kr[lane + 64 * v], ac);
                bd = fmaf(qv[v], pr[lane + 64 * v], bd);
            }
        }
        return (wave_sum(ac) + wave_sum(bd)) * scale;
    };
    float mx = -INFINITY;
    for (int j = 0; j < len; ++j)
        if (visible(j)) mx = fmaxf(mx, score(j));
    float den = 0.0f;
    for (int j = 0; j < len; ++j) {
        if (!visible(j)) continue;
        const float e = expf(score(j) - mx);
        den += e;
        const float* vr = base + (size_t)j * ld + 2 * d;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (ok[v]) acc[v] = fmaf(e, vr[lane + 64 * v], acc[v]);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (ok[v]) orow[lane + 64 * v] = den > 0.0f ? acc[v] / den : 0.0f;
}

// The same attention with the LANES OVER THE KEYS (round 6): the kernel above spends its time in the two 6-step wave reductions
// it makes per key (profiles/r06_05_f32_*_before.txt: 34 % of the NeMo float32 mode, 70 % of the ESPnet one).  Here a lane owns keys
// lane, lane + 64, .. and computes the whole head_dim dot products of ITS keys (ascending e, one fma chain per term; the wave's query
// vectors q + u, q + v sit in LDS and are read as broadcasts), so a query costs two wave reductions in total (maximum, denominator);
// the P.V sum then walks the keys with the probability read from its lane (v_readlane) and the value row read coalesced.  Scores live
// in registers: T <= 64 NCH (NCH = 4 / 8); longer inputs (long-form windows) take the kernel above.  Same mask semantics.
template <int NCH>
__global__ __launch_bounds__(256) void attention_f32_keys_kernel(const float* __restrict__ qkv, const float* __restrict__ pos,
                                                                 const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                                 const int32_t* __restrict__ lens, float* __restrict__ out, int T,
                                                                 int d, int hd, int att_left, int att_right, int n_global, float scale) {
    __shared__ __attribute__((aligned(16))) float qs[4][2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
    if (i >= T) return;                                    // (no workgroup barrier below: a wave only reads the LDS rows it wrote)
    int len = lens[b];
    len = len < T ? len : T;
    float* orow = out + ((size_t)b * T + i) * d + h * hd;
    if (i >= len) {
        for (int e = lane; e < hd; e += 64) orow[e] = 0.0f;
        return;
    }
    const size_t ld = 3 * (size_t)d;
    const float* base = qkv + (size_t)b * T * ld + h * hd;
    for (int e = lane; e < hd; e += 64) {
        const float q = base[(size_t)i * ld + e];
        qs[wave][0][e] = q + bias_u[h * hd + e];
        qs[wave][1][e] = q + bias_v[h * hd + e];
    }
    __builtin_amdgcn_wave_barrier();
    const float4* qu4 = reinterpret_cast<const float4*>(qs[wave][0]);
    const float4* qv4 = reinterpret_cast<const float4*>(qs[wave][1]);
    const bool window = att_left >= 0 || att_right >= 0;
    const int left = att_left >= 0 ? att_left : T, right = att_right >= 0 ? att_right : T;
    float sc[NCH];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = lane + 64 * c;
        bool vis = j < len;
        if (vis && window) vis = ((i - j) <= left && (j - i) <= right) || i < n_global || j < n_global;
        sc[c] = -INFINITY;
        if (vis) {
            const float4* kr = reinterpret_cast<const float4*>(base + (size_t)j * ld + d);
            const float4* pr = reinterpret_cast<const float4*>(pos + (size_t)(j - i + T - 1) * d + h * hd);
            float ac = 0.0f, bd = 0.0f;
            for (int e4 = 0; e4 < hd / 4; ++e4) {
                const float4 kv = kr[e4], pv = pr[e4], a = qu4[e4], g = qv4[e4];
                ac = fmaf(a.x, kv.x, ac); ac = fmaf(a.y, kv.y, ac); ac = fmaf(a.z, kv.z, ac); ac = fmaf(a.w, kv.w, ac);
                bd = fmaf(g.x, pv.x, bd); bd = fmaf(g.y, pv.y, bd); bd = fmaf(g.z, pv.z, bd); bd = fmaf(g.w, pv.w, bd);
            }
            sc[c] = (ac + bd) * scale;
            mx = fmaxf(mx, sc[c]);
        }
    }
    mx = wave_max(mx);
    float den = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        sc[c] = sc[c] > -INFINITY ? expf(sc[c] - mx) : 0.0f;
        den += sc[c];
    }
    den = wave_sum(den);
    // P.V: this lane's output elements e = lane, lane + 64, ..; keys in ascending order
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* vbase = base + 2 * d;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int jn = len - 64 * c < 64 ? len - 64 * c : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const float p = __shfl(sc[c], jj, 64);
            if (p != 0.0f) {                               // (wave-uniform: masked keys contribute nothing)
                const float* vr = vbase + (size_t)(64 * c + jj) * ld;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (lane + 64 * v < hd) acc[v] = fmaf(p, vr[lane + 64 * v], acc[v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
        if (lane + 64 * v < hd) orow[lane + 64 * v] = den > 0.0f ? acc[v] / den : 0.0f;
}

// Full-context rel-pos attention in float32 on the matrix cores (round 6).  A wave owns 16 queries of one (utterance, head) and walks
// the keys 16 at a time with an online softmax; every product is v_mfma_f32_16x16x4_f32 with operands read straight from global
// memory as 16-byte pieces (lane (li, kq) takes elements 16 S + 4 kq .. + 3 of row li: step (S, E) of a chain multiplies element
// 16 S + 4 kq + E of the four lane groups at once — the summation order of gemm_f32_kernel):
//   AC   D[key][query]  = K rows x (q + u) rows                                   1 accumulator tile
//   BD   G[c][query]    = P rows (r0 + c, c = 0 .. 31) x (q + v) rows              2 tiles; r0 = j0 - i0 - 15 + T - 1, so the score of
//        (query i0 + a, key j0 + k) needs G[k - a + 15][a]: the rel-shift is a read of a query-major skew tile in wave-private LDS
//   P.V  O[e][query]   += V^T x probabilities, the four keys of an MFMA step chosen as j0 + 4 kq + m so that a lane's own probability
//        register m IS its operand (no shuffle); element e = 64 g + 4 (4 kq + r) + n of tile (g, n), so a lane ends up with 16
//        consecutive output elements per 64-wide group
// Masking: keys at or past the utterance's length weigh 0, padded queries give zeros.  Limited-context variants (att_left / att_right /
// global tokens) take attention_f32_keys_kernel.  HD = 64 or 128.  grid (ceil(T / 64), H, B), block 256 (four independent waves).
template <int HD>
__global__ __launch_bounds__(256) void attention_f32_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ pos,
                                                                 const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                                 const int32_t* __restrict__ lens, float* __restrict__ out, int T, int d, float scale) {
    constexpr int NS = HD / 16, NG = HD / 64;
    __shared__ float skew[4][16][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.x * 64 + wave * 16, h = blockIdx.y, b = blockIdx.z;
    if (i0 >= T) return;
    int len = lens[b];
    len = len < T ? len : T;
    const size_t ld = 3 * (size_t)d;
    const float* base = qkv + (size_t)b * T * ld + h * HD;
    const int qi = i0 + li;
    float* orow = out + ((size_t)b * T + (qi < T ? qi : T - 1)) * d + h * HD;
    if (i0 >= len) {                                        // a tile of padding queries
        if (qi < T)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(orow + 64 * g + 16 * kq + 4 * r) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int qrow = qi < T ? qi : T - 1;
    float4 qu[NS], qv[NS];
#pragma unroll
    for (int S = 0; S < NS; ++S) {
        const float4 q4 = *reinterpret_cast<const float4*>(base + (size_t)qrow * ld + 16 * S + 4 * kq);
        const float4 u4 = *reinterpret_cast<const float4*>(bias_u + h * HD + 16 * S + 4 * kq);
        const float4 v4 = *reinterpret_cast<const float4*>(bias_v + h * HD + 16 * S + 4 * kq);
        qu[S] = make_float4(q4.x + u4.x, q4.y + u4.y, q4.z + u4.z, q4.w + u4.w);
        qv[S] = make_float4(q4.x + v4.x, q4.y + v4.y, q4.z + v4.z, q4.w + v4.w);
    }
    f32x4_t O[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int n = 0; n < 4; ++n) O[g][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const int ntile = (len + 15) / 16;
    float (*sk)[33] = skew[wave];
    for (int jt = 0; jt < ntile; ++jt) {
        const int j0 = jt * 16;
        const int krow = j0 + li < len ? j0 + li : len - 1;
        const int r0 = j0 - i0 - 15 + T - 1;
        int pr0 = r0 + li, pr1 = r0 + 16 + li;
        pr0 = pr0 < 0 ? 0 : (pr0 > 2 * T - 2 ? 2 * T - 2 : pr0);
        pr1 = pr1 < 0 ? 0 : (pr1 > 2 * T - 2 ? 2 * T - 2 : pr1);
        const float* kp = base + (size_t)krow * ld + d + 4 * kq;
        const float* p0p = pos + (size_t)pr0 * d + h * HD + 4 * kq;
        const float* p1p = pos + (size_t)pr1 * d + h * HD + 4 * kq;
        f32x4_t ac = {0.f, 0.f, 0.f, 0.f}, g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            const float4 kf = *reinterpret_cast<const float4*>(kp + 16 * S);
            const float4 pa = *reinterpret_cast<const float4*>(p0p + 16 * S);
            const float4 pb = *reinterpret_cast<const float4*>(p1p + 16 * S);
#define RS_ATT_STEP(E)                                                                      \
            ac = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.E, qu[S].E, ac, 0, 0, 0);          \
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa.E, qv[S].E, g0, 0, 0, 0);          \
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pb.E, qv[S].E, g1, 0, 0, 0);
            RS_ATT_STEP(x) RS_ATT_STEP(y) RS_ATT_STEP(z) RS_ATT_STEP(w)
#undef RS_ATT_STEP
        }
        // rel-shift through the query-major skew tile: G[c][query li] with c = 16 t + 4 kq + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sk[li][4 * kq + r] = g0[r];
            sk[li][16 + 4 * kq + r] = g1[r];
        }
        __builtin_amdgcn_wave_barrier();
        float sc[4];
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bd = sk[li][4 * kq + r - li + 15];
            sc[r] = j0 + 4 * kq + r < len ? (ac[r] + bd) * scale : -INFINITY;
            mt = fmaxf(mt, sc[r]);
        }
        __builtin_amdgcn_wave_barrier();
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);               // finite from the first tile on (key 0 is always visible)
        const float alpha = expf(m_run - m_new);
        float pr[4], lt = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pr[r] = expf(sc[r] - m_new);
            lt += pr[r];
        }
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        l_run = l_run * alpha + lt;
        m_run = m_new;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) O[g][n][r] *= alpha;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int vrow = j0 + 4 * kq + m < len ? j0 + 4 * kq + m : len - 1;
            const float* vp = base + (size_t)vrow * ld + 2 * d + 4 * li;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 v4 = *reinterpret_cast<const float4*>(vp + 64 * g);
                O[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.x, pr[m], O[g][0], 0, 0, 0);
                O[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.y, pr[m], O[g][1], 0, 0, 0);
                O[g][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.z, pr[m], O[g][2], 0, 0, 0);
                O[g][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.w, pr[m], O[g][3], 0, 0, 0);
            }
        }
    }
    if (qi < T) {
        const float inv = qi < len ? 1.0f / l_run : 0.0f;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(orow + 64 * g + 16 * kq + 4 * r) =
                    make_float4(O[g][0][r] * inv, O[g][1][r] * inv, O[g][2][r] * inv, O[g][3][r] * inv);
    }
}

// conv-module middle in float32: x [B*T][2d] (values | gates, NeMo's own order) -> GLU -> frame mask -> depthwise k
// (BatchNorm folded) -> SiLU -> out [B*T][d].  One thread per output element, channel fastest.
__global__ __launch_bounds__(256) void glu_dwconv_silu_f32_kernel(const float* __restrict__ x, const float* __restrict__ w /* [k][d] */,
                                                                  const float* __restrict__ bias, const int32_t* __restrict__ lens,
                                                                  int T, int d, int k, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, b = blockIdx.z;
    if (c >= d) return;
    int len = lens[b];
    len = len < T ? len : T;
    const int half = (k - 1) >> 1;
    float acc = 0.0f;                                 // F.conv1d: products first, bias last
    for (int j = 0; j < k; ++j) {
        const int tj = t + j - half;
        if (tj < 0 || tj >= len) continue;
        const float* px = x + ((size_t)b * T + tj) * 2 * d;
        const float u = px[c] * sigmoid_exact(px[d + c]);
        acc = fmaf(w[(size_t)j * d + c], u, acc);
    }
    acc += bias[c];
    out[((size_t)b * T + t) * d + c] = silu_exact(acc);
}

struct EncPlanF32 {
    int T[5], F[5];
    size_t off_lens, off_sa, off_sb, off_col, off_x, off_hn, off_big, off_ctx, off_posp, off_ctc, total;
    int chunk;       // Conv2dSubsampling: utterances per pass of conv0 / patch gather / dense-conv GEMM
};

EncPlanF32 plan_f32(const rs_ctx* ctx, int B, int t_max) {
    const rs_dims& d = ctx->d;
    EncPlanF32 p{};
    p.T[0] = t_max; p.F[0] = d.n_mels;
    for (int s = 1; s <= d.sub_stages; ++s) { p.T[s] = rs_conv_len(p.T[s - 1], d.sub_kind); p.F[s] = rs_conv_len(p.F[s - 1], d.sub_kind); }
    const size_t C = d.sub_channels, dm = d.d_model;
    const size_t Tp = p.T[d.sub_stages] > 0 ? p.T[d.sub_stages] : 1, M = (size_t)B * Tp;
    size_t widest = (size_t)d.ff_dim;
    if (3 * dm > widest) widest = 3 * dm;
    size_t o = 0;
    p.off_lens = o; o += rs_align((size_t)4 * B * 4);
    p.chunk = B;
    p.off_col = 0;
    if (d.sub_kind == 1) {
        // the layout of rs_api.hip's plan_encoder with float32 elements: conv0 output and 3x3 patches of ONE chunk of
        // utterances (patch matrix near 1 GiB), the dense conv's output of the whole batch
        const size_t T2 = p.T[2] > 0 ? p.T[2] : 1;
        const size_t per_utt_col = T2 * p.F[2] * 9 * C * 4;
        size_t chunk = ((size_t)1 << 30) / per_utt_col;
        if (chunk < 1) chunk = 1;
        if (chunk > (size_t)B) chunk = (size_t)B;
        while (chunk > 1 && chunk * T2 > 65535) --chunk;
        p.chunk = (int)chunk;
        p.off_sa = o; o += rs_align(chunk * (size_t)(p.T[1] > 0 ? p.T[1] : 1) * p.F[1] * C * 4);
        p.off_col = o; o += rs_align(chunk * per_utt_col);
        p.off_sb = o; o += rs_align((size_t)B * T2 * p.F[2] * C * 4);
    } else {
        const size_t sub_elems = (size_t)B * p.T[2] * p.F[2] * C;
        p.off_sa = o; o += rs_align(sub_elems * 4);
        p.off_sb = o; o += rs_align(sub_elems * 4);
    }
    p.off_x = o; o += rs_align(M * dm * 4);
    p.off_hn = o; o += rs_align(M * dm * 4);
    p.off_big = o; o += rs_align(M * widest * 4);
    p.off_ctx = o; o += rs_align(M * dm * 4);
    p.off_posp = o; o += rs_align((2 * Tp) * dm * 4);
    p.off_ctc = o;
    if (d.ctc_vocab > 0) o += rs_align(M * (size_t)rs_ctc_pad(d.ctc_vocab) * 4);
    p.total = o + 256;
    return p;
}

}  // namespace

int rs_launch_gemm_f32(rs_ctx* ctx, const float* A, int lda, const float* W, int ldw, float* out, int ldc, int M, int N, int K,
                       int flags, const float* bias, float alpha, const float* residual, const int32_t* mask_lens,
                       int mask_rows_per_step, int mask_steps, hipStream_t s) {
    if (M <= 0 || N <= 0) return RS_OK;
    if (K <= 0 || K % GK || N % 4 || (lda % 4) || (ldw % 4) || (ldc % 4))
        return rs_fail(ctx, RS_EINVAL, "gemm_f32: K %% %d, N %% 4 and 16-byte row pitches required (M %d N %d K %d)", GK, M, N, K);
    if (flags & ~(RS_GEMM_BIAS | RS_GEMM_RELU | RS_GEMM_SILU | RS_GEMM_RESIDUAL | RS_GEMM_OUT_F32 | RS_GEMM_ROWMASK | RS_GEMM_SWOOSHL | RS_GEMM_SWOOSHR | RS_GEMM_GELU))
        return rs_fail(ctx, RS_EINVAL, "gemm_f32: unsupported flags %d", flags);
    if ((flags & RS_GEMM_BIAS) && !bias) return rs_fail(ctx, RS_EINVAL, "gemm_f32: bias flag without a bias");
    if ((flags & RS_GEMM_RESIDUAL) && !residual) return rs_fail(ctx, RS_EINVAL, "gemm_f32: residual flag without a residual");
    if ((flags & RS_GEMM_ROWMASK) && (!mask_lens || mask_rows_per_step <= 0 || mask_steps <= 0))
        return rs_fail(ctx, RS_EINVAL, "gemm_f32: row mask without lengths");
    GemmF32 p{A, W, out, bias, residual, mask_lens, lda, ldw, ldc, M, N, K, flags, alpha, mask_rows_per_step, mask_steps, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr};
    const dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT), block(256);
    rs_prof_begin(ctx, RS_PROF_GEMM, s, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    if (ctx->gemm_f32_x3) hipLaunchKernelGGL((gemm_f32_kernel<false, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, p);
    rs_prof_end(ctx, RS_PROF_GEMM, s);
    RS_CHECK_LAUNCH(ctx, "gemm_f32");
    return RS_OK;
}

// A 3 x 3 convolution (padding 1, stride 1 or 2) of a channels-last float32 map as ONE product: out[(n, oh, ow)][co] =
// sum over (kh, kw, c) of in[n][s oh + kh - 1][s ow + kw - 1][c] * W[co][(kh, kw, c)], the patches gathered by the GEMM's loader (no patch
// matrix in HBM: the AV-HuBERT trunk's were 4.5 GB per convolution of the first stage), followed by the folded inference BatchNorm,
// an optional residual [rows][Cout] and an optional per-channel PReLU in the epilogue.  in_rows_as_matrix: the 1 x 1 form (K = C, no
// gather) with the same epilogue, for the down-sampling branch.  Same summation order as patches + rs_launch_gemm_f32.
int rs_launch_conv3x3_f32(rs_ctx* ctx, const float* in, int n_img, int H, int Wd, int C, int OH, int OW, int stride, const float* W, int Cout,
                          const float* bn_scale, const float* bn_shift, const float* residual, const float* prelu, float* out, int one_by_one, hipStream_t s) {
    const long long rows = (long long)n_img * OH * OW;
    if (rows <= 0) return RS_OK;
    const int K = one_by_one ? C : 9 * C;
    if (C % GK || Cout % 4 || rows > 0x7fffffffLL || !bn_scale || !bn_shift)
        return rs_fail(ctx, RS_EINVAL, "conv3x3_f32: C %% %d, Cout %% 4, < 2^31 output rows and a folded BatchNorm required (C %d Cout %d rows %lld)", GK, C, Cout, rows);
    GemmF32 p{in, W, out, nullptr, residual, nullptr, C, K, Cout, (int)rows, Cout, K, residual ? RS_GEMM_RESIDUAL : 0, 1.0f, 0, 0,
              H, Wd, C, OH, OW, stride, bn_scale, bn_shift, prelu};
    const bool n64 = Cout <= 64;
    const dim3 grid((Cout + (n64 ? 64 : GT) - 1) / (n64 ? 64 : GT), (unsigned)((rows + GT - 1) / GT)), block(256);
    rs_prof_begin(ctx, RS_PROF_GEMM, s, 2.0 * rows * (double)Cout * K, 4.0 * ((double)n_img * H * Wd * C + (double)Cout * K + 2.0 * rows * Cout));
    if (ctx->gemm_f32_x3) {
        if (one_by_one) {
            if (n64) hipLaunchKernelGGL((gemm_f32_kernel<false, true, true>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((gemm_f32_kernel<false, false, true>), grid, block, 0, s, p);
        } else if (n64) hipLaunchKernelGGL((gemm_f32_kernel<true, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, false, true>), grid, block, 0, s, p);
    } else
    if (one_by_one) {
        if (n64) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, p);
    } else if (n64) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, p);
    rs_prof_end(ctx, RS_PROF_GEMM, s);
    RS_CHECK_LAUNCH(ctx, "conv3x3_f32");
    return RS_OK;
}

int rs_launch_attention_f32(rs_ctx* ctx, const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                            const int32_t* lens, int B, int T, float* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return RS_OK;
    const rs_dims& dm = ctx->d;
    const int hd = dm.d_model / dm.n_heads;
    if (hd > 256) return rs_fail(ctx, RS_EINVAL, "attention_f32: head_dim %d > 256", hd);
    const dim3 grid((T + 3) / 4, dm.n_heads, B), block(256);
    const float scale = 1.0f / sqrtf((float)hd);
    rs_prof_begin(ctx, RS_PROF_ATTN, s, (double)B * dm.n_heads * 3.0 * 2.0 * T * (double)T * hd, (double)B * T * dm.d_model * 4.0 * 4.0);
#define RS_ATT_KEYS(NCH)                                                                                                 \
    hipLaunchKernelGGL((attention_f32_keys_kernel<NCH>), grid, block, 0, s, qkv, pos, bias_u, bias_v, lens, out, T, dm.d_model, hd, \
                       dm.att_left, dm.att_right, dm.n_global, scale)
#define RS_ATT_CASE(NV)                                                                                                  \
    hipLaunchKernelGGL((attention_f32_kernel<NV>), grid, block, 0, s, qkv, pos, bias_u, bias_v, lens, out, T, dm.d_model, hd, \
                       dm.att_left, dm.att_right, dm.n_global, scale)
    static const bool one_wave_per_key_sum = getenv("RS_ATTN_F32_OLD") != nullptr;      // A/B and test hook: the first form
    static const bool no_mfma = getenv("RS_ATTN_F32_KEYS") != nullptr;                 // A/B and test hook: the second form
    if (!one_wave_per_key_sum && !no_mfma && dm.att_left < 0 && dm.att_right < 0 && (hd == 64 || hd == 128)) {
        const dim3 grid16((T + 63) / 64, dm.n_heads, B);
        if (hd == 128) hipLaunchKernelGGL((attention_f32_mfma_kernel<128>), grid16, block, 0, s, qkv, pos, bias_u, bias_v, lens, out, T, dm.d_model, scale);
        else hipLaunchKernelGGL((attention_f32_mfma_kernel<64>), grid16, block, 0, s, qkv, pos, bias_u, bias_v, lens, out, T, dm.d_model, scale);
    } else
    if (!one_wave_per_key_sum && T <= 512 && hd % 4 == 0) {
        if (T <= 256) RS_ATT_KEYS(4);
        else RS_ATT_KEYS(8);
    } else
    if (hd <= 64) RS_ATT_CASE(1);
    else if (hd <= 128) RS_ATT_CASE(2);
    else RS_ATT_CASE(4);
#undef RS_ATT_CASE
#undef RS_ATT_KEYS
    rs_prof_end(ctx, RS_PROF_ATTN, s);
    RS_CHECK_LAUNCH(ctx, "attention_f32");
    return RS_OK;
}

int rs_launch_glu_dwconv_f32(rs_ctx* ctx, const float* x, const float* w, const float* b, const int32_t* lens, int B, int T,
                             int d, int k, float* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return RS_OK;
    if (k < 1 || !(k & 1)) return rs_fail(ctx, RS_EINVAL, "glu_dwconv_f32: kernel size %d unsupported", k);
    const dim3 grid((d + 255) / 256, T, B), block(256);
    rs_prof_begin(ctx, RS_PROF_ELEMENTWISE, s, (double)B * T * d * (2.0 * k + 12.0), (double)B * T * d * 12.0);
    hipLaunchKernelGGL(glu_dwconv_silu_f32_kernel, grid, block, 0, s, x, w, b, lens, T, d, k, out);
    rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s);
    RS_CHECK_LAUNCH(ctx, "glu_dwconv_silu_f32");
    return RS_OK;
}

int rs_launch_gemm_f32_skinny(rs_ctx* ctx, const float* A, int lda, const float* W, int ldw, float* out, int ldc, int M, int N, int K, int flags,
                              const float* bias, const float* residual, hipStream_t s) {
    if (M <= 0 || N <= 0) return RS_OK;
    if (M > 128 || K <= 0 || K % 16 || N % 4 || (lda % 4) || (ldw % 4) || (ldc % 4))
        return rs_fail(ctx, RS_EINVAL, "gemm_f32 (skinny): M <= 128, K %% 16, N %% 4 and 16-byte row pitches required (M %d N %d K %d)", M, N, K);
    if (flags & ~(RS_GEMM_BIAS | RS_GEMM_RELU | RS_GEMM_SILU | RS_GEMM_GELU | RS_GEMM_RESIDUAL | RS_GEMM_OUT_F32))
        return rs_fail(ctx, RS_EINVAL, "gemm_f32 (skinny): unsupported flags %d", flags);
    GemmF32 p{A, W, out, bias, residual, nullptr, lda, ldw, ldc, M, N, K, flags, 1.0f, 0, 0};
    const dim3 grid((N + 15) / 16), block(64 * SK_WAVES);
    rs_prof_begin(ctx, RS_PROF_GEMM, s, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    switch ((M + 15) / 16) {                      // row tiles of 16: only as many as there are rows
        case 1: hipLaunchKernelGGL((gemm_f32_skinny_kernel<1>), grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL((gemm_f32_skinny_kernel<2>), grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL((gemm_f32_skinny_kernel<3>), grid, block, 0, s, p); break;
        case 4: hipLaunchKernelGGL((gemm_f32_skinny_kernel<4>), grid, block, 0, s, p); break;
        case 5: hipLaunchKernelGGL((gemm_f32_skinny_kernel<5>), grid, block, 0, s, p); break;
        case 6: hipLaunchKernelGGL((gemm_f32_skinny_kernel<6>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_f32_skinny_kernel<8>), grid, block, 0, s, p); break;
    }
    rs_prof_end(ctx, RS_PROF_GEMM, s);
    RS_CHECK_LAUNCH(ctx, "gemm_f32_skinny");
    return RS_OK;
}

size_t rs_encoder_f32_workspace_bytes(const rs_ctx* ctx, int B, int t_max) { return plan_f32(ctx, B, t_max).total; }

// The float32 encoder: the call sequence of rs_encoder_forward (rs_api.hip) with float32 operands everywhere and no
// fusion that moves a rounding (there is none to move: nothing is rounded below float32).
int rs_encoder_forward_f32(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max, float* enc_out,
                           float* joint_enc, int32_t* enc_lens, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const rs_f32_weights& w = ctx->f32;
    const EncPlanF32 pl = plan_f32(ctx, B, t_max);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "encoder (float32 mode): workspace %zu < %zu", workspace_bytes, pl.total);
    char* ws = reinterpret_cast<char*>(workspace);
    int32_t* lens_stage = reinterpret_cast<int32_t*>(ws + pl.off_lens);
    float* sa = reinterpret_cast<float*>(ws + pl.off_sa);
    float* sb = reinterpret_cast<float*>(ws + pl.off_sb);
    float* x = reinterpret_cast<float*>(ws + pl.off_x);
    float* hn = reinterpret_cast<float*>(ws + pl.off_hn);
    float* big = reinterpret_cast<float*>(ws + pl.off_big);
    float* ctxb = reinterpret_cast<float*>(ws + pl.off_ctx);
    float* posp = reinterpret_cast<float*>(ws + pl.off_posp);
    const int C = d.sub_channels, dm = d.d_model, ff = d.ff_dim, S = d.sub_stages;
    const int Tp = pl.T[S], M = B * Tp;
    int rc;
#define RS_TRY(call) do { rc = (call); if (rc != RS_OK) return rc; } while (0)
    auto gemm = [&](const float* A, int lda, const float* W, int K, float* out, int ldc, int Mr, int N, int flags, const float* bias,
                    float alpha, const float* res) -> int {
        return rs_launch_gemm_f32(ctx, A, lda, W, K, out, ldc, Mr, N, K, flags, bias, alpha, res, nullptr, 0, 0, s);
    };
    // ---- subsampling
    RS_TRY(rs_launch_enc_lens(ctx, n_frames, B, lens_stage, s));
    if (Tp <= 0) return rs_fail(ctx, RS_EINVAL, "encoder (float32 mode): %d feature frames are too few for the subsampling", t_max);
    if (d.sub_kind == 1) {
        // ESPnet Conv2dSubsampling in float32: conv0 -> 3x3 patches -> the dense conv as one exact-f32 GEMM per chunk of utterances
        float* col = reinterpret_cast<float*>(ws + pl.off_col);
        const int T1 = pl.T[1], F1 = pl.F[1], T2 = pl.T[2], F2 = pl.F[2];
        for (int b0 = 0; b0 < B; b0 += pl.chunk) {
            const int bc = B - b0 < pl.chunk ? B - b0 : pl.chunk;
            RS_TRY(rs_launch_sub2d_conv0_f32(ctx, feats, lens_stage, b0, bc, t_max, T1, F1, sa, s));
            RS_TRY(rs_launch_im2col3x3s2_f32(ctx, sa, bc, T1, F1, T2, F2, col, s));
            RS_TRY(rs_launch_gemm_f32(ctx, col, 9 * C, w.sub_conv1_w, 9 * C, sb + (size_t)b0 * T2 * F2 * C, C, bc * T2 * F2, C, 9 * C,
                                      RS_GEMM_BIAS | RS_GEMM_RELU | RS_GEMM_ROWMASK, ctx->sub_conv1_b, 1.0f, nullptr,
                                      lens_stage + B + b0, F2, T2, s));
        }
    } else
    RS_TRY(rs_launch_sub_conv0_dw1_f32(ctx, feats, lens_stage, B, t_max, pl.T[2], pl.F[2], sa, s));
    for (int st = 2; st <= S && d.sub_kind == 0; ++st) {
        if (st > 2)
            RS_TRY(rs_launch_sub_dw_f32(ctx, sb, ctx->sub_dw_w[st - 2], ctx->sub_dw_b[st - 2], lens_stage + (st - 1) * B, B,
                                        pl.T[st - 1], pl.F[st - 1], pl.T[st], pl.F[st], sa, s));
        RS_TRY(rs_launch_gemm_f32(ctx, sa, C, w.sub_pw_w[st - 2], C, sb, C, B * pl.T[st] * pl.F[st], C, C,
                                  RS_GEMM_BIAS | RS_GEMM_RELU | RS_GEMM_ROWMASK, ctx->sub_pw_b[st - 2], 1.0f, nullptr,
                                  lens_stage + (st - 1) * B, pl.F[st], pl.T[st], s));
    }
    {
        const int K = C * pl.F[S];
        RS_TRY(gemm(sb, K, w.sub_out_w, K, x, dm, M, dm, RS_GEMM_BIAS, ctx->sub_out_b, d.xscaling ? sqrtf((float)dm) : 1.0f, nullptr));
    }
    const int32_t* lens = lens_stage + (S - 1) * B;
    RS_HIP(ctx, hipMemcpyAsync(enc_lens, lens, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    if (ctx->tap_sub) RS_HIP(ctx, hipMemcpyAsync(ctx->tap_sub, x, (size_t)M * dm * 4, hipMemcpyDeviceToDevice, s));
    // ---- position rows for this T'
    const int tcap = (int)((w.pos_table_bytes / ((size_t)dm * 4) + 1) / 2);
    if (Tp > tcap) return rs_fail(ctx, RS_EINVAL, "encoder (float32 mode): T'=%d exceeds the registered pos.table.f32 capacity %d", Tp, tcap);
    const float* pos_slice = w.pos_table + (size_t)(tcap - Tp) * dm;
    const int npos = 2 * Tp - 1;
    const int RES = RS_GEMM_BIAS | RS_GEMM_RESIDUAL;
    for (int i = 0; i < d.n_layers; ++i) {
        const rs_layer_w& L = ctx->layers[i];
        const rs_layer_w32& L32 = w.layers[i];
        // 1/2 FFN
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_ff1_g, L.ln_ff1_b, M, dm, d.ln_eps, nullptr, hn, s));
        RS_TRY(gemm(hn, dm, L32.ff1_w1, dm, big, ff, M, ff, RS_GEMM_BIAS | RS_GEMM_SILU, L.ff1_b1, 1.0f, nullptr));
        RS_TRY(gemm(big, ff, L32.ff1_w2, ff, x, dm, M, dm, RES, L.ff1_b2, 0.5f, x));
        // rel-pos MHSA
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_att_g, L.ln_att_b, M, dm, d.ln_eps, nullptr, hn, s));
        RS_TRY(gemm(hn, dm, L32.qkv_w, dm, big, 3 * dm, M, 3 * dm, RS_GEMM_BIAS, L.qkv_b, 1.0f, nullptr));
        RS_TRY(gemm(pos_slice, dm, L32.pos_w, dm, posp, dm, npos, dm, 0, nullptr, 1.0f, nullptr));
        RS_TRY(rs_launch_attention_f32(ctx, big, posp, L.bias_u, L.bias_v, lens, B, Tp, ctxb, s));
        RS_TRY(gemm(ctxb, dm, L32.out_w, dm, x, dm, M, dm, RES, L.out_b, 1.0f, x));
        // conv module (pw1 in NeMo's own row order: values | gates)
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_conv_g, L.ln_conv_b, M, dm, d.ln_eps, nullptr, hn, s));
        RS_TRY(gemm(hn, dm, L32.pw1_w, dm, big, 2 * dm, M, 2 * dm, RS_GEMM_BIAS, L32.pw1_b, 1.0f, nullptr));
        RS_TRY(rs_launch_glu_dwconv_f32(ctx, big, L.dw_w, L.dw_b, lens, B, Tp, dm, d.conv_kernel, ctxb, s));
        RS_TRY(gemm(ctxb, dm, L32.pw2_w, dm, x, dm, M, dm, RES, L.pw2_b, 1.0f, x));
        // 1/2 FFN
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_ff2_g, L.ln_ff2_b, M, dm, d.ln_eps, nullptr, hn, s));
        RS_TRY(gemm(hn, dm, L32.ff2_w1, dm, big, ff, M, ff, RS_GEMM_BIAS | RS_GEMM_SILU, L.ff2_b1, 1.0f, nullptr));
        RS_TRY(gemm(big, ff, L32.ff2_w2, ff, x, dm, M, dm, RES, L.ff2_b2, 0.5f, x));
        // output norm, in place (a wave holds its whole row in registers before it stores)
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_out_g, L.ln_out_b, M, dm, d.ln_eps, nullptr, x, s));
        for (size_t k = 0; k < ctx->tap_ids.size(); ++k)
            if (ctx->tap_ids[k] == i)
                RS_HIP(ctx, hipMemcpyAsync(ctx->tap_layers + k * (size_t)M * dm, x, (size_t)M * dm * 4, hipMemcpyDeviceToDevice, s));
    }
    // ESPnet: the encoder's after_norm on top of the last block's norm_final
    if (d.final_norm) RS_TRY(rs_launch_layernorm(ctx, x, ctx->final_norm_g, ctx->final_norm_b, M, dm, d.ln_eps, nullptr, x, s));
    if (enc_out) RS_HIP(ctx, hipMemcpyAsync(enc_out, x, (size_t)M * dm * 4, hipMemcpyDeviceToDevice, s));
    RS_TRY(gemm(x, dm, w.jenc_w, dm, joint_enc, d.joint_hidden, M, d.joint_hidden, RS_GEMM_BIAS, ctx->jenc_b, 1.0f, nullptr));
    if (d.ctc_vocab > 0 && (ctx->ctc_probs || ctx->ctc_blank)) {
        const int Vp = rs_ctc_pad(d.ctc_vocab);
        float* z = ctx->ctc_probs ? ctx->ctc_probs : reinterpret_cast<float*>(ws + pl.off_ctc);
        RS_TRY(gemm(x, dm, w.ctc_w, dm, z, Vp, M, Vp, RS_GEMM_BIAS, ctx->ctc_b, 1.0f, nullptr));
        RS_TRY(rs_launch_ctc_softmax(ctx, z, M, d.ctc_vocab, Vp, d.blank_id, ctx->ctc_blank, s));
    }
#undef RS_TRY
    return RS_OK;
}

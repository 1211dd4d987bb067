// k_zipformer.hip — the Zipformer2 encoder of `reazonspeech.k2.asr` (BASELINE.json configs[3]).
//
// The reference hands three ONNX files to sherpa-onnx (pkg/k2-asr/src/huggingface.py:73-83) and calls decode_stream once per
// utterance (transcribe.py:36-39).  [UPSTREAM] the encoder file holds icefall's encoder_embed (Conv2dSubsampling), the
// Zipformer2 stacks and joiner.encoder_proj (export-onnx.py OnnxEncoder); restated module by module in oracle/zipformer.py,
// which this file is compared with.  What runs where:
//
//   dense contractions            every Linear / 1x1 conv / the third 3x3 conv (as patches) on gemm_smf16 (k_gemm_bf16.hip, MFMA
//                                 16x16x32 bf16) with bias / SwooshL / SwooshR / GLU / f32-residual epilogues
//   attention weights             k2_attn_weights_kernel: scores = q.k (one MFMA per 16 x 16 tile: head_dim 32 IS the MFMA's K)
//                                 + p.pos[j - i] (4-wide position head on the VALU from an LDS-staged table), two-pass softmax,
//                                 weights stored once as bf16 [B][H][T][T] and shared by the three consumers of a layer
//   weights x values              k2_vt_kernel (V^T once per branch; the non-linear attention's tanh gate applied there) +
//                                 k2_pv_kernel: MFMA over 32-key chunks, V^T staged in LDS, 128 queries per block, the
//                                 non-linear attention's output gate in the epilogue
//   encoder_embed                 conv0 on the VALU, conv1 on MFMA straight from global memory (k2_conv1_mfma_kernel), conv2 as
//                                 patches + GEMM, ConvNeXt depthwise 7 x 7 with a ring of seven output frames
//   everything else               HBM-bound element-wise kernels (casts, BiasNorm, bypass, down/up-sampling, depthwise convs)
//
// Batch semantics: the reference runs one utterance per call, so every kernel masks by the utterance's own length (keys past it
// weigh 0, convolutions see zeros, SimpleDownsample repeats the utterance's own last frame): a row's result does not depend on
// its batch mates.
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "rs_common.h"

struct rs_k2_layer {
    const uint16_t *attw_in_w, *sa_in_w[2], *sa_out_w[2], *ff_in_w[3], *ff_out_w[3], *na_in_w, *na_out_w, *cm_in_w[2], *cm_out_w[2];
    const float *attw_in_b, *pos_proj, *sa_in_b[2], *sa_out_b[2], *ff_in_b[3], *ff_out_b[3], *na_in_b, *na_out_b, *cm_in_b[2], *cm_dw_w[2],
        *cm_dw_b[2], *cm_out_b[2], *norm_bias, *norm_scale, *bypass, *bypass_mid;
};

// float32 parity mode: the dense weights once more, unrounded ("<name>.f32"; the conv modules' in_proj in icefall's own row
// order: values, then gates).  Biases, BiasNorm / bypass / down-sampling constants, depthwise taps and the projected position
// rows are float32 in the throughput mode already and are shared.
struct rs_k2_layer32 {
    const float *attw_in_w, *sa_in_w[2], *sa_out_w[2], *ff_in_w[3], *ff_out_w[3], *na_in_w, *na_out_w, *cm_in_w[2], *cm_in_b[2], *cm_out_w[2];
};

struct rs_k2 {
    rs_k2_dims d{};
    std::vector<std::vector<rs_k2_layer>> stacks;
    std::vector<std::vector<rs_k2_layer32>> stacks32;
    const float *conv2_w32 = nullptr, *cnx_pw1_w32 = nullptr, *cnx_pw2_w32 = nullptr, *emb_out_w32 = nullptr, *jenc_w32 = nullptr;
    const float *ds_w[8] = {}, *comb_scale[8] = {}, *out_ds_w = nullptr;
    const float *conv0_w = nullptr, *conv0_b = nullptr, *conv1_w = nullptr, *conv1_b = nullptr, *conv2_b = nullptr, *cnx_dw_w = nullptr,
                *cnx_dw_b = nullptr, *cnx_pw1_b = nullptr, *cnx_pw2_b = nullptr, *emb_out_b = nullptr, *emb_norm_bias = nullptr,
                *emb_norm_scale = nullptr;
    const uint16_t *conv2_w = nullptr, *cnx_pw1_w = nullptr, *cnx_pw2_w = nullptr, *emb_out_w = nullptr;
    int pos_cap = 0;                 // rows of every "attw.pos_proj" table = 2 * pos_cap - 1
    int embed_freq = 0, out_dim = 0;
    float* tap_embed = nullptr;      // parity taps (rs_k2_encoder_set_taps)
    float* tap_stacks = nullptr;
};

namespace {

constexpr int K2_QD = 32, K2_PD = 4, K2_VD = 12;

__host__ __device__ inline int pad64(int n) { return (n + 63) / 64 * 64; }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- encoder_embed ---------------------------------------------------------------------------------------------------------
// conv0: Conv2d(1, C1, 3, padding (0, 1)) + SwooshR.  feats f32 [B][T][F] -> a0 bf16 [B][T - 2][F][C1] (channels last)
// grid (T1, B), block 256
__global__ __launch_bounds__(256) void k2_conv0_kernel(const float* __restrict__ feats, int T, int F, int C1, const float* __restrict__ w /* [9][C1] */,
                                                       const float* __restrict__ bias, uint16_t* __restrict__ out) {
    __shared__ float rows[3][136];
    const int t1 = blockIdx.x, b = blockIdx.y, T1 = T - 2;
    for (int i = threadIdx.x; i < 3 * (F + 2); i += 256) {
        const int r = i / (F + 2), f = i - r * (F + 2) - 1;
        rows[r][f + 1] = (f >= 0 && f < F) ? feats[((size_t)b * T + t1 + r) * F + f] : 0.0f;
    }
    __syncthreads();
    uint16_t* orow = out + ((size_t)b * T1 + t1) * F * C1;
    for (int i = threadIdx.x; i < F * C1; i += 256) {
        const int f = i / C1, c = i - f * C1;
        float acc = bias[c];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = fmaf(w[(kh * 3 + kw) * C1 + c], rows[kh][f + kw], acc);
        orow[i] = f32_to_bf16(swoosh_r_f(acc));
    }
}

// conv1: Conv2d(C1, C2, 3, stride 2) + SwooshR.  a0 bf16 [B][T1][F][C1] -> a1 bf16 [B][T2][F2][C2]; T2 = (T1 - 3) / 2 + 1.
// grid (T2, B), block 256; the three input rows (3 x F x C1) and the weights (9 x C1 x C2) sit in LDS as f32
__global__ __launch_bounds__(256) void k2_conv1_kernel(const uint16_t* __restrict__ a0, int T1, int F, int C1, int T2, int F2, int C2,
                                                       const float* __restrict__ w /* [3][3][C1][C2] */, const float* __restrict__ bias,
                                                       uint16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* rows = reinterpret_cast<float*>(smem);                 // [3][F * C1]
    float* ws = rows + 3 * F * C1;                                 // [9 * C1][C2]
    const int t2 = blockIdx.x, b = blockIdx.y;
    for (int i = threadIdx.x; i < 3 * F * C1; i += 256) {
        const int r = i / (F * C1), q = i - r * (F * C1);
        rows[i] = bf16_to_f32(a0[((size_t)b * T1 + 2 * t2 + r) * F * C1 + q]);
    }
    for (int i = threadIdx.x; i < 9 * C1 * C2; i += 256) ws[i] = round_bf16(w[i]);     // (bf16 weights like every GEMM: the MFMA form's operand)
    __syncthreads();
    uint16_t* orow = out + ((size_t)b * T2 + t2) * F2 * C2;
    for (int i = threadIdx.x; i < F2 * C2; i += 256) {
        const int f2 = i / C2, c2 = i - f2 * C2;
        float acc = bias[c2];
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const float* xr = rows + kh * F * C1 + (2 * f2 + kw) * C1;
                const float* wr = ws + ((kh * 3 + kw) * C1) * C2 + c2;
                for (int c1 = 0; c1 < C1; ++c1) acc = fmaf(wr[c1 * C2], xr[c1], acc);
            }
        orow[i] = f32_to_bf16(swoosh_r_f(acc));
    }
}

// conv1 on the matrix cores (second form; the first keeps rows and weights in LDS as f32 and pays two LDS reads per FMA: 1.9 ms
// per batch of 256).  Per output frame t2 the convolution is a [C2 = 32] x [K = 72 -> 96] x [F2 = 39 -> 48 pixels] product; with
// channels-last input the K index (kh, kw, c1) makes every 8-element piece of a pixel's patch — (kh, kw, c1 = 0 .. 7) — 16
// contiguous bytes of the input row 2 t2 + kh at column 2 f2 + kw: the second MFMA operand is read straight from global memory,
// no patch matrix, no LDS.  The weights (first operand: rows = output channels) are rounded to bf16 once per wave and stay in
// registers; a wave walks CONV1_TT consecutive frames.  D[c2 = 4 kq + e][pixel = lane & 15]: a lane stores four consecutive
// channels of one pixel.  Requires C1 == 8, C2 == 32, F2 <= 48 (the launcher checks; anything else runs the first form).
constexpr int CONV1_TT = 4;
// grid (ceil(T2 / (4 * CONV1_TT)), B), block 256
__global__ __launch_bounds__(256) void k2_conv1_mfma_kernel(const uint16_t* __restrict__ a0, int T1, int F, int T2, int F2,
                                                            const float* __restrict__ w /* [3][3][8][32] */, const float* __restrict__ bias,
                                                            uint16_t* __restrict__ out) {
    constexpr int C1 = 8, C2 = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int li = lane & 15, kq = lane >> 4;
    // weights: fragment (nt, ks) = W[c2 = 16 nt + li][k = 32 ks + 8 kq + e], k = (kh * 3 + kw) * 8 + c1; zeros for k >= 72
    bf16x8_t wf[2][3];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            u16x8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * ks + 8 * kq + e;
                v[e] = k < 72 ? f32_to_bf16(w[(size_t)k * C2 + 16 * nt + li]) : (unsigned short)0;
            }
            wf[nt][ks] = __builtin_bit_cast(bf16x8_t, v);
        }
    float bs[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[nt][e] = bias[16 * nt + 4 * kq + e];
    const int t_first = (blockIdx.x * 4 + wave) * CONV1_TT;
    for (int tt = 0; tt < CONV1_TT; ++tt) {
        const int t2 = t_first + tt;
        if (t2 >= T2) return;
        // patches: fragment (mt, ks) of pixel f2 = 16 mt + li: piece g = 4 ks + kq -> (kh = g / 3, kw = g % 3)
        bf16x8_t pf[3][3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int f2 = 16 * mt + li, g = 4 * ks + kq;
                u16x8_t v = (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
                if (f2 < F2 && g < 9) {
                    const int kh = g / 3, kw = g - 3 * kh;
                    v = *reinterpret_cast<const u16x8_t*>(a0 + (((size_t)b * T1 + 2 * t2 + kh) * F + 2 * f2 + kw) * C1);
                }
                pf[mt][ks] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int f2 = 16 * mt + li;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], pf[mt][ks], acc, 0, 0, 0);
                if (f2 < F2)
                    *reinterpret_cast<u16x4_t*>(out + (((size_t)b * T2 + t2) * F2 + f2) * C2 + 16 * nt + 4 * kq) =
                        pack_bf16x4(swoosh_r_f(acc[0] + bs[nt][0]), swoosh_r_f(acc[1] + bs[nt][1]), swoosh_r_f(acc[2] + bs[nt][2]),
                                    swoosh_r_f(acc[3] + bs[nt][3]));
            }
        }
    }
}

// patches of conv2 = Conv2d(C2, C3, 3, stride (1, 2)): row (b, t3, f3) of the patch matrix holds the 3 x 3 x C2 inputs
// a1[b][t3 + kh][2 f3 + kw][:] in (kh, kw, c) order, zero-padded to Kp columns.  One wave per row, 16-byte pieces.
__global__ __launch_bounds__(256) void k2_im2col_kernel(const uint16_t* __restrict__ a1, int T2, int F2, int C2, int T3, int F3, int Kp,
                                                        long long rows, uint16_t* __restrict__ col) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int f3 = (int)(row % F3);
    const long long bt = row / F3;
    const int t3 = (int)(bt % T3), b = (int)(bt / T3);
    const int c8 = C2 / 8, n16 = Kp / 8;
    uint4* dst = reinterpret_cast<uint4*>(col + row * Kp);
    for (int q = lane; q < n16; q += 64) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (q < 9 * c8) {
            const int tap = q / c8, c = q - tap * c8;
            const int kh = tap / 3, kw = tap - 3 * kh;
            v = reinterpret_cast<const uint4*>(a1 + (((size_t)b * T2 + t3 + kh) * F2 + 2 * f3 + kw) * C2)[c];
        }
        dst[q] = v;
    }
}

// conv2 = Conv2d(32, 128, 3, stride (1, 2)) + SwooshR as ONE kernel (round 6; the pattern of k2_cnx_pw_fused_kernel below): until now a
// patch matrix [rows][320] went through HBM (k2_im2col_kernel, 1.8 GB at the benchmark batch) into a GEMM launch: 0.78 + 1.07 ms.
// Here a persistent workgroup gathers the 3 x 3 x 32 patches of 128 output positions straight into LDS with global_load_lds (36
// 16-byte pieces per row, two tile buffers of 72 KB, the next tile's gather in flight under this tile's products), the weights
// (128 x 288 bf16) live in registers, cut over the eight waves by output channel (9 fragments each), and a wave multiplies its 16
// channels against all 128 rows.  K = 288 is nine 32-deep steps; the GEMM's tenth step (its K is padded to 320) multiplies zeros:
// same sums.  Epilogue of the GEMM launch: + bias, SwooshR, float32 out.  $RS_K2_CONV2_FUSED=0 / rs_set_option("k2_conv2_fused", 0)
// runs the two launches; a test compares the bits.  grid = CUs, block 512, 144 KB of dynamic LDS.
constexpr int C2_K = 288, C2_ROWB = C2_K * 2, C2_TILE = 128 * C2_ROWB, C2_LDS = 2 * C2_TILE;
__global__ __launch_bounds__(512) void k2_conv2_fused_kernel(const uint16_t* __restrict__ a1, int T2, int F2, int T3, int F3, const uint16_t* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias, float* __restrict__ out, long long rows) {
    extern __shared__ __attribute__((aligned(16))) char c2_smem[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)c2_smem;
    const long long n_tiles = (rows + 127) / 128;
    bf16x8_t wf[9];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(W + (size_t)(16 * wave + li) * ldw + 32 * ks + 8 * kq);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * wave + 4 * kq);
    // 128 rows x 36 pieces = 72 DMA instructions of 64 consecutive pieces, nine per wave.  Piece pc of row r (physical position in
    // LDS) holds logical piece (pc & ~3) | ((pc & 3) ^ ((r >> 2) & 3)): rows are 576 bytes apart, i.e. rows r and r + 4 start in the
    // same bank group, and the XOR spreads the sixteen rows of a fragment read over all sixteen 16-byte slots of a bank row.
    auto issue_tile = [&](long long tile, int buf) {
#pragma unroll
        for (int jj = 0; jj < 9; ++jj) {
            const int q = wave + 8 * jj;
            const int g = 64 * q + lane, r = g / 36, pc = g - 36 * r;
            const int piece = (pc & ~3) | ((pc & 3) ^ ((r >> 2) & 3));
            long long row = tile * 128 + r;
            row = row < rows ? row : rows - 1;
            const int f3 = (int)(row % F3);
            const long long bt = row / F3;
            const int t3 = (int)(bt % T3);
            const long long b = bt / T3;
            const int tap = piece >> 2, c = piece & 3, kh = tap / 3, kw = tap - 3 * kh;
            const long long src = (((b * T2 + t3 + kh) * F2 + 2 * f3 + kw) * 32 + 8 * c) * 2;      // bytes into a1 (may exceed 4 GiB: 64-bit address)
            // per-lane 64-bit address: global_load_lds with a vector address (no scalar base)
            const char* ptr = reinterpret_cast<const char*>(a1) + src;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(ptr), "s"(lds0 + buf * C2_TILE + q * 1024) : "memory");
        }
    };
    long long tile = blockIdx.x;
    if (tile < n_tiles) issue_tile(tile, 0);
    int buf = 0;
    bool first = true;
    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        // this tile's patches have landed; the previous tile's eight output stores (younger in the in-order queue) may still be in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tile + gridDim.x < n_tiles) issue_tile(tile + gridDim.x, buf ^ 1);
        const char* Ps = c2_smem + buf * C2_TILE;
#pragma unroll 2
        for (int mt = 0; mt < 8; ++mt) {
            const int row = 16 * mt + li;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const int c = 4 * ks + kq;
                const int pc = (c & ~3) | ((c & 3) ^ ((row >> 2) & 3));
                const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(Ps + row * C2_ROWB + (pc << 4));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], pf, acc, 0, 0, 0);
            }
            const long long gr = tile * 128 + row;
            const float4 v = make_float4(swoosh_r_f(acc[0] + b4.x) * 1.0f, swoosh_r_f(acc[1] + b4.y) * 1.0f, swoosh_r_f(acc[2] + b4.z) * 1.0f,
                                         swoosh_r_f(acc[3] + b4.w) * 1.0f);
            if (gr < rows) *reinterpret_cast<float4*>(out + gr * 128 + 16 * wave + 4 * kq) = v;
        }
    }
}

// ConvNeXt's two pointwise convolutions as ONE kernel (round 6): out = bf16(res + W2 . swooshL(W1 . a + b1) + b2) for rows of C = 128
// channels, hidden width 3 C = 384.  As two GEMM launches the [rows][384] hidden tensor (2.2 GB at the benchmark batch) was written and
// read back: 1.85 + 1.13 ms per batch.  Here it never leaves the CU.  Both weight matrices are only 96 KB each, but 160 KB of LDS cannot
// hold both next to the operand tiles — and a row tile's stages are shorter than one L2 round trip, so streaming them per tile stalls.
// So the WEIGHTS live in registers, cut over the eight waves of a persistent workgroup: wave w owns hidden channels 48 w .. + 47 of W1
// (12 fragments) and output channels 16 w .. + 15 of W2 (12 fragments), 96 VGPRs for the whole launch; LDS holds the row tile (two
// buffers of 128 x 128 bf16, filled by global_load_lds one tile ahead) and the hidden tile (128 x 384 bf16), through which the waves
// exchange what each computed for ALL 128 rows.  Per tile: stage 1 (wave: 3 x 8 output tiles, K = 128) -> + b1, SwooshL, bf16 -> hidden
// tile -> barrier -> stage 2 (wave: 1 x 8 output tiles, K = 384) -> + b2 + residual -> bf16 through the dead row-tile buffer ->
// whole-row stores.  The K order of both products and every rounding point are those of the two GEMM launches: bit-identical
// ($RS_K2_CNX_FUSED=0 runs the launches; tests compare the two).  grid = CUs, block 512, 160 KB of dynamic LDS.
__device__ __forceinline__ void cx_glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
constexpr int CX_C = 128, CX_H = 384, CX_ROWS = 128;
constexpr int CX_A_BYTES = CX_ROWS * CX_C * 2, CX_H_BYTES = CX_ROWS * CX_H * 2, CX_LDS = 2 * CX_A_BYTES + CX_H_BYTES;
__global__ __launch_bounds__(512) void k2_cnx_pw_fused_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W1, const float* __restrict__ b1,
                                                              const uint16_t* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ res,
                                                              uint16_t* __restrict__ out, long long rows) {
    extern __shared__ __attribute__((aligned(16))) char cx_smem[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)cx_smem;
    char* Hs = cx_smem + 2 * CX_A_BYTES;
    const long long n_tiles = (rows + CX_ROWS - 1) / CX_ROWS;
    // the wave's weight fragments, for the whole launch: lane (channel li of a 16-tile, k group kq) holds 8 consecutive k
    bf16x8_t w1[3][4], w2[12];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            w1[t][ks] = *reinterpret_cast<const bf16x8_t*>(W1 + (size_t)(48 * wave + 16 * t + li) * CX_C + 32 * ks + 8 * kq);
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) w2[ks] = *reinterpret_cast<const bf16x8_t*>(W2 + (size_t)(16 * wave + li) * CX_H + 32 * ks + 8 * kq);
    float4 bias1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bias1[t] = *reinterpret_cast<const float4*>(b1 + 48 * wave + 16 * t + 4 * kq);
    const float4 bias2 = *reinterpret_cast<const float4*>(b2 + 16 * wave + 4 * kq);
    // row tile -> LDS buffer: 32 DMA instructions of 4 rows (1 KB), four per wave; 16-byte pieces XOR-swizzled by the row
    auto issue_a = [&](long long tile, int buf) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int q = wave + 8 * jj;
            const int r = 4 * q + (lane >> 4), pc = lane & 15;
            long long gr = tile * CX_ROWS + r;
            gr = gr < rows ? gr : rows - 1;
            // (the matrix can exceed 4 GiB: the tile's base goes into the scalar address, the lane offset stays below 32 KiB)
            const long long gr0 = tile * CX_ROWS < rows ? tile * CX_ROWS : rows - 1;
            const unsigned voff = (unsigned)((gr - gr0) * (CX_C * 2) + ((pc ^ (r & 15)) << 4));
            cx_glds16(voff, reinterpret_cast<const char*>(A) + gr0 * (CX_C * 2), lds0 + buf * CX_A_BYTES + q * 1024);
        }
    };
    long long tile = blockIdx.x;
    if (tile < n_tiles) issue_a(tile, 0);
    int buf = 0;
    bool first = true;
    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        // this tile's rows have landed: the only younger entries of the (in-order) queue are the previous tile's four output stores
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();                                // ... for every wave; the hidden tile and the other buffer are free
        asm volatile("" ::: "memory");
        if (tile + gridDim.x < n_tiles) issue_a(tile + gridDim.x, buf ^ 1);
        // the residual rows of stage 2, asked for now: a 1.5 GB stream that no cache holds — loaded where they are used they cost a DRAM
        // round trip per 16 rows (the first version: 1.50 ms per batch)
        float4 rres[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            long long gr = tile * CX_ROWS + 16 * mt + li;
            gr = gr < rows ? gr : rows - 1;
            rres[mt] = *reinterpret_cast<const float4*>(res + gr * CX_C + 16 * wave + 4 * kq);
        }
        const char* As = cx_smem + buf * CX_A_BYTES;
        // ---- stage 1: hidden channels 48 wave .. + 47 of all 128 rows
#pragma unroll 2
        for (int mt = 0; mt < 8; ++mt) {
            const int row = 16 * mt + li;
            f32x4_t acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(As + row * (CX_C * 2) + (((4 * ks + kq) ^ (row & 15)) << 4));
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[t][ks], af, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const u16x4_t hv = pack_bf16x4(swoosh_l_f(acc[t][0] + bias1[t].x), swoosh_l_f(acc[t][1] + bias1[t].y),
                                               swoosh_l_f(acc[t][2] + bias1[t].z), swoosh_l_f(acc[t][3] + bias1[t].w));
                const int c = 6 * wave + 2 * t + (kq >> 1);          // 16-byte piece of the hidden row that holds channels 48 w + 16 t + 4 kq ..
                const int cs = (c & ~15) | ((c & 15) ^ (row & 15));
                *reinterpret_cast<u16x4_t*>(Hs + row * (CX_H * 2) + (cs << 4) + 8 * (kq & 1)) = hv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the hidden tile is complete; nobody reads the row tile any more
        asm volatile("" ::: "memory");
        // ---- stage 2: output channels 16 wave .. + 15 of all 128 rows
        char* Os = cx_smem + buf * CX_A_BYTES;                       // the dead row tile: the output tile's staging
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int row = 16 * mt + li;
            const float4 r4 = rres[mt];
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                const int c = 4 * ks + kq;
                const int cs = (c & ~15) | ((c & 15) ^ (row & 15));
                const bf16x8_t hf = *reinterpret_cast<const bf16x8_t*>(Hs + row * (CX_H * 2) + (cs << 4));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[ks], hf, acc, 0, 0, 0);
            }
            // the residual GEMM's epilogue: + bias, (* alpha = 1,) + residual, round to bf16
            const u16x4_t ov = pack_bf16x4((acc[0] + bias2.x) * 1.0f + r4.x, (acc[1] + bias2.y) * 1.0f + r4.y, (acc[2] + bias2.z) * 1.0f + r4.z,
                                           (acc[3] + bias2.w) * 1.0f + r4.w);
            const int c = 2 * wave + (kq >> 1);
            *reinterpret_cast<u16x4_t*>(Os + row * (CX_C * 2) + ((c ^ (row & 15)) << 4) + 8 * (kq & 1)) = ov;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the output tile is complete
        asm volatile("" ::: "memory");
#pragma unroll
        for (int pss = 0; pss < 4; ++pss) {                          // wave: rows 16 wave .. + 15 as whole 256-byte rows
            const int row = 16 * wave + 4 * pss + (lane >> 4), pc = lane & 15;
            const uint4 v = *reinterpret_cast<const uint4*>(Os + row * (CX_C * 2) + ((pc ^ (row & 15)) << 4));
            const long long gr = tile * CX_ROWS + row;
            if (gr < rows) *reinterpret_cast<uint4*>(out + gr * CX_C + 8 * pc) = v;
        }
    }
}

// ConvNeXt depthwise 7 x 7 (padding 3) over (time, frequency), channels last.  a2 f32 [B][T3][F3][C] -> bf16 same shape.
// Frames at or past the utterance's own length are zeros (the reference's single-utterance call ends there).
// A workgroup makes CNX_TT consecutive frames of one utterance; a thread owns one channel and FH consecutive frequencies and
// walks time: every input row is loaded once (FH + 6 values for 49 FH multiply-adds; the next row's loads fly under this row's
// arithmetic) and scattered into the seven output frames it contributes to, which live in a ring of seven accumulator sets
// (slot = output frame mod 7, resolved at compile time by unrolling rows in groups of seven); the 49 taps of the channel sit in
// registers.  History: all 49 inputs fetched per output, 11 ms per batch of 256 (profiles/r05b_k2_kernel_stats.txt); the ring
// with one frequency per thread (7 loads per 49 multiply-adds), 2.1 ms; this form 1.1 ms less (profiles/r05x_k2_forms_ab.txt).
// Order of the sums: rows ascending, taps left to right, bias last.
constexpr int CNX_TT = 29;          // frames per workgroup: CNX_TT + 6 input rows = 5 groups of 7
// grid (ceil(T3 / CNX_TT), B), block 256
template <int FH>
__global__ __launch_bounds__(256, 2) void k2_cnx_dw_kernel(const float* __restrict__ a2, const int32_t* __restrict__ lens3, int T3, int F3, int C,
                                                         const float* __restrict__ w /* [49][C] */, const float* __restrict__ bias,
                                                         uint16_t* __restrict__ out) {
    const int t0 = blockIdx.x * CNX_TT, b = blockIdx.y;
    int len = lens3[b];
    len = len < T3 ? len : T3;
    const float* xin = a2 + (size_t)b * T3 * F3 * C;
    uint16_t* xo = out + (size_t)b * T3 * F3 * C;
    const int nfh = (F3 + FH - 1) / FH;
    for (int item = threadIdx.x; item < nfh * C; item += 256) {
        const int fh = item / C, c = item - fh * C, f_lo = fh * FH;
        float wt[49];
#pragma unroll
        for (int k = 0; k < 49; ++k) wt[k] = w[k * C + c];
        const float bb = bias[c];
        float acc[7][FH];
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int fo = 0; fo < FH; ++fo) acc[k][fo] = 0.0f;
        // branch-free loads: clamped addresses (the row base is wave-uniform, the FH + 6 offsets are per-thread constants) and the
        // out-of-range inputs zeroed by a bit mask
        int offq[FH + 6];
        unsigned okm[FH + 6];
#pragma unroll
        for (int q = 0; q < FH + 6; ++q) {
            const int ff = f_lo + q - 3;
            okm[q] = (ff >= 0 && ff < F3) ? 0xffffffffu : 0u;
            offq[q] = (ff < 0 ? 0 : ff >= F3 ? F3 - 1 : ff) * C + c;
        }
        auto load_row = [&](int r, float (&x)[FH + 6]) {
            const unsigned rowmask = (r >= 0 && r < len) ? 0xffffffffu : 0u;
            const int rc = max(0, min(r, T3 - 1));
            const float* row = xin + (size_t)rc * F3 * C;
#pragma unroll
            for (int q = 0; q < FH + 6; ++q) x[q] = __uint_as_float(__float_as_uint(row[offq[q]]) & okm[q] & rowmask);
        };
        float x[FH + 6];
        load_row(t0 - 3, x);
        for (int g = 0; g < (CNX_TT + 6) / 7; ++g) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int r = t0 - 3 + 7 * g + j;
                float xn[FH + 6];
                load_row(r + 1, xn);                     // the next row's loads fly under this row's multiply-adds
#pragma unroll
                for (int kh = 0; kh < 7; ++kh) {
                    const int slot = (j + 3 - kh + 7) % 7;
#pragma unroll
                    for (int fo = 0; fo < FH; ++fo)
#pragma unroll
                        for (int kw = 0; kw < 7; ++kw) acc[slot][fo] = fmaf(wt[kh * 7 + kw], x[fo + kw], acc[slot][fo]);
                }
                const int done = r - 3, dslot = (j + 4) % 7;
                if (done >= t0 && done < t0 + CNX_TT && done < T3) {
#pragma unroll
                    for (int fo = 0; fo < FH; ++fo)
                        if (f_lo + fo < F3) xo[((size_t)done * F3 + f_lo + fo) * C + c] = f32_to_bf16(acc[dslot][fo] + bb);
                }
#pragma unroll
                for (int fo = 0; fo < FH; ++fo) acc[dslot][fo] = 0.0f;
#pragma unroll
                for (int q = 0; q < FH + 6; ++q) x[q] = xn[q];
                __builtin_amdgcn_sched_barrier(0);       // keeps the scheduler from hoisting all seven rows' loads (256 VGPRs + spills)
            }
        }
    }
}

// ---- element-wise pieces of the stacks ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k2_cast_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<u16x4_t*>(out)[i] = pack_bf16x4(v.x, v.y, v.z, v.w);
}

// BiasNorm (+ optional bypass): y = x * rsqrt(mean((x - bias)^2)) * scale; with x0: y = x0 + (y - x0) * bypass[c].
// One wave per row; writes f32 (in place allowed) and optionally the bf16 copy the next GEMM reads.
__global__ __launch_bounds__(256) void k2_biasnorm_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ scale_p,
                                                          const float* __restrict__ x0, const float* __restrict__ bypass, int M, int d,
                                                          float* __restrict__ out, uint16_t* __restrict__ out_bf16) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * d;
    float ss = 0.0f;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c), bb = *reinterpret_cast<const float4*>(bias + c);
        const float a = v.x - bb.x, b2 = v.y - bb.y, c2 = v.z - bb.z, e = v.w - bb.w;
        ss += a * a + b2 * b2 + c2 * c2 + e * e;
    }
    ss = wave_sum(ss);
    const float sc = rsqrtf(ss / (float)d) * scale_p[0];
    for (int c = lane * 4; c < d; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(xr + c);
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        if (x0) {
            const float4 o = *reinterpret_cast<const float4*>(x0 + (size_t)row * d + c), s = *reinterpret_cast<const float4*>(bypass + c);
            v.x = o.x + (v.x - o.x) * s.x; v.y = o.y + (v.y - o.y) * s.y; v.z = o.z + (v.z - o.z) * s.z; v.w = o.w + (v.w - o.w) * s.w;
        }
        *reinterpret_cast<float4*>(out + (size_t)row * d + c) = v;
        if (out_bf16) *reinterpret_cast<u16x4_t*>(out_bf16 + (size_t)row * d + c) = pack_bf16x4(v.x, v.y, v.z, v.w);
    }
}

// bypass in the middle of a layer: x = x0 + (x - x0) * scale[c] (in place) and its bf16 copy
__global__ __launch_bounds__(256) void k2_bypass_kernel(float* __restrict__ x, const float* __restrict__ x0, const float* __restrict__ scale, int d,
                                                        size_t n4, uint16_t* __restrict__ out_bf16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % (size_t)d);
    float4 v = reinterpret_cast<float4*>(x)[i];
    const float4 o = reinterpret_cast<const float4*>(x0)[i], s = *reinterpret_cast<const float4*>(scale + c);
    v.x = o.x + (v.x - o.x) * s.x; v.y = o.y + (v.y - o.y) * s.y; v.z = o.z + (v.z - o.z) * s.z; v.w = o.w + (v.w - o.w) * s.w;
    reinterpret_cast<float4*>(x)[i] = v;
    if (out_bf16) reinterpret_cast<u16x4_t*>(out_bf16)[i] = pack_bf16x4(v.x, v.y, v.z, v.w);
}

// Entry of a stack: channel conversion (cut or zero-pad to d) of the previous stack's output -> src [B][T][d] (the operand of
// the stack's out_combiner), and SimpleDownsample by ds -> xs [B][Ts][d]: weighted sum (softmax(bias), done on the host) of
// each group of ds frames, frames past the utterance's end replaced by ITS last frame; rows past ceil(len / ds) are zeros.
// grid (Ts, B), block 256
__global__ __launch_bounds__(256) void k2_stack_in_kernel(const float* __restrict__ prev, int d_prev, const int32_t* __restrict__ lens, int T, int d,
                                                          int ds, int Ts, const float* __restrict__ wds, float* __restrict__ src,
                                                          float* __restrict__ xs) {
    const int ts = blockIdx.x, b = blockIdx.y;
    int len = lens[b];
    len = len < T ? len : T;
    const int dmin = d < d_prev ? d : d_prev;
    for (int c = threadIdx.x; c < d; c += 256) {
        float acc = 0.0f;
        for (int k = 0; k < ds; ++k) {
            const int t = ts * ds + k;
            if (t < T && src) src[((size_t)b * T + t) * d + c] = c < dmin ? prev[((size_t)b * T + t) * d_prev + c] : 0.0f;
            int tt = t < len ? t : len - 1;
            const float v = (c < dmin && tt >= 0) ? prev[((size_t)b * T + tt) * d_prev + c] : 0.0f;
            acc = fmaf(v, ds > 1 ? wds[k] : 1.0f, acc);
        }
        if (xs) xs[((size_t)b * Ts + ts) * d + c] = ts * ds < len ? acc : 0.0f;
    }
}

// Exit of a down-sampled stack: SimpleUpsample (repeat each frame ds times, cut to T) + out_combiner:
// out[b][t][c] = src + (xs[b][t / ds][c] - src) * scale[c].  grid (T, B)
__global__ __launch_bounds__(256) void k2_stack_out_kernel(const float* __restrict__ src, const float* __restrict__ xs, int T, int Ts, int d, int ds,
                                                           const float* __restrict__ scale, float* __restrict__ out) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < d; c += 256) {
        const float o = src[((size_t)b * T + t) * d + c];
        const float v = xs[((size_t)b * Ts + t / ds) * d + c];
        out[((size_t)b * T + t) * d + c] = o + (v - o) * scale[c];
    }
}

// Output of the encoder: the last stack's channels extended by the extra channels of earlier, wider stacks (`pieces`: up to 8
// (pointer, first channel, channel count, row pitch)), then SimpleDownsample by 2 with the utterance's own last frame repeated.
// -> enc f32 [B][To][D] (optional) and its bf16 copy (the operand of joiner.encoder_proj); rows past (len + 1) / 2 are zeros.
struct K2Pieces { const float* p[8]; int c0[8], n[8], ld[8]; int count; };
__global__ __launch_bounds__(256) void k2_output_kernel(K2Pieces pc, const int32_t* __restrict__ lens, int T, int To, int D, const float* __restrict__ wds,
                                                        float* __restrict__ enc, uint16_t* __restrict__ enc_bf16, int32_t* __restrict__ out_lens) {
    const int to = blockIdx.x, b = blockIdx.y;
    int len = lens[b];
    len = len < T ? len : T;
    if (to == 0 && threadIdx.x == 0) out_lens[b] = (len + 1) / 2;
    for (int c = threadIdx.x; c < D; c += 256) {
        int pi = 0, base = 0;
        while (pi + 1 < pc.count && c >= base + pc.n[pi]) { base += pc.n[pi]; ++pi; }
        const float* src = pc.p[pi];
        const int col = pc.c0[pi] + (c - base), ld = pc.ld[pi];
        float acc = 0.0f;
        for (int k = 0; k < 2; ++k) {
            int t = 2 * to + k;
            t = t < len ? t : len - 1;
            acc = fmaf(t >= 0 ? src[((size_t)b * T + t) * ld + col] : 0.0f, wds[k], acc);
        }
        if (2 * to >= len) acc = 0.0f;
        if (enc) enc[((size_t)b * To + to) * D + c] = acc;
        if (enc_bf16) enc_bf16[((size_t)b * To + to) * D + c] = f32_to_bf16(acc);
    }
}

// ---- attention weights ----------------------------------------------------------------------------------------------------------
// qkp bf16 [B*T][ld]: columns [0, H*32) queries, [H*32, 2*H*32) keys, [2*H*32, 2*H*32 + H*4) position queries (head-major).
// pos f32 [2*cap-1][H*4]: linear_pos of the compact relative-position encoding, row n <-> relative position n - (cap - 1).
// W bf16 [B][H][T][Tp]: softmax over the utterance's own keys of  q_i.k_j + p_i.pos[j - i];  keys / queries past the length: 0.
// Block = (64 queries, head, utterance), 4 waves of 16 queries.  A 16 x 16 score tile is ONE v_mfma_f32_16x16x32_bf16
// (head_dim 32 = the MFMA's K extent): first operand the key rows, second the query rows, so a lane holds the scores of query
// (lane & 15) against keys 4 * (lane >> 4) .. + 3.  Two passes over the keys (maximum and sum, then the normalised weights):
// the scores are recomputed instead of kept, so any T fits.  The position rows the block can touch (rel in [-(i0 + 63), len - 1 - i0])
// sit in LDS as float4.
// one score: the q.k product of the MFMA plus the 4-wide position term as ONE explicit fma chain (both attention-weight kernels call
// this, so the compiler cannot contract the two differently: their weights are bit-identical)
__device__ __forceinline__ float k2_score(float qk, const float4& pq, const float4& pr) {
    float t = pq.x * pr.x;
    t = fmaf(pq.y, pr.y, t);
    t = fmaf(pq.z, pr.z, t);
    t = fmaf(pq.w, pr.w, t);
    return qk + t;
}

__global__ __launch_bounds__(256) void k2_attn_weights_kernel(const uint16_t* __restrict__ qkp, int ld, const float* __restrict__ pos, int cap, int H,
                                                              const int32_t* __restrict__ lens, int T, int Tp, uint16_t* __restrict__ W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* ps = reinterpret_cast<float4*>(smem);
    const int h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int len = lens[b];
    len = len < T ? len : T;
    uint16_t* wbase = W + (((size_t)b * H + h) * T) * Tp;
    if (i0 >= len) {                                     // a tile of padding queries: zeros
        for (int idx = threadIdx.x; idx < 64 * (Tp / 4); idx += 256) {
            const int r = idx / (Tp / 4), q = idx - r * (Tp / 4);
            if (i0 + r < T) reinterpret_cast<u16x4_t*>(wbase + (size_t)(i0 + r) * Tp)[q] = (u16x4_t){0, 0, 0, 0};
        }
        return;
    }
    // position rows for rel = -(i0 + 63) .. len - 1 - i0  ->  ps[rel + i0 + 63]
    const int nrel = len + 63;
    for (int r = threadIdx.x; r < nrel; r += 256) {
        int n = r - (i0 + 63) + cap - 1;                  // (rows only padding queries could ask for are clamped: never used)
        n = n < 0 ? 0 : (n > 2 * cap - 2 ? 2 * cap - 2 : n);
        ps[r] = *reinterpret_cast<const float4*>(pos + (size_t)n * (H * K2_PD) + h * K2_PD);
    }
    __syncthreads();
    const int qi = i0 + wave * 16 + (lane & 15);         // this lane's query
    const int kq = lane >> 4;                             // its key quad within a 16-key tile
    const bool q_ok = qi < len;
    const int qrow = qi < T ? qi : T - 1;
    const uint16_t* qp = qkp + ((size_t)b * T + qrow) * ld;
    const bf16x8_t qfrag = *reinterpret_cast<const bf16x8_t*>(qp + h * K2_QD + 8 * kq);
    float4 pq;
    {
        const u16x4_t pv = *reinterpret_cast<const u16x4_t*>(qp + 2 * H * K2_QD + h * K2_PD);
        pq = make_float4(bf16_to_f32(pv[0]), bf16_to_f32(pv[1]), bf16_to_f32(pv[2]), bf16_to_f32(pv[3]));
    }
    const int ntile = (len + 15) / 16;
    auto scores = [&](int jt, float (&s)[4]) {
        int krow = jt * 16 + (lane & 15);
        krow = krow < len ? krow : len - 1;
        const bf16x8_t kfrag = *reinterpret_cast<const bf16x8_t*>(qkp + ((size_t)b * T + krow) * ld + H * K2_QD + h * K2_QD + 8 * kq);
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qfrag, acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jt * 16 + 4 * kq + e;
            int r = j - qi + i0 + 63;
            r = r < 0 ? 0 : (r >= nrel ? nrel - 1 : r);
            const float4 pr = ps[r];
            const float v = k2_score(acc[e], pq, pr);
            s[e] = j < len ? v : -INFINITY;
        }
    };
    // pass 1: row maximum, then the sum of exp (two sweeps keep the arithmetic of a plain softmax: max first)
    float mx = -INFINITY;
    for (int jt = 0; jt < ntile; ++jt) {
        float s[4];
        scores(jt, s);
        mx = fmaxf(mx, fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
    for (int jt = 0; jt < ntile; ++jt) {
        float s[4];
        scores(jt, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += __expf(s[e] - mx);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = q_ok ? 1.0f / sum : 0.0f;
    // pass 2: the weights
    const int ntile_all = Tp / 16;
    for (int jt = 0; jt < ntile_all; ++jt) {
        u16x4_t o = {0, 0, 0, 0};
        if (jt < ntile) {
            float s[4];
            scores(jt, s);
            o = pack_bf16x4(__expf(s[0] - mx) * inv, __expf(s[1] - mx) * inv, __expf(s[2] - mx) * inv, __expf(s[3] - mx) * inv);
        }
        if (qi < T) *reinterpret_cast<u16x4_t*>(wbase + (size_t)qi * Tp + jt * 16 + 4 * kq) = o;
    }
}

// The same weights in ONE sweep over the keys (round 6): the 16 x 16 score tiles of a wave's 16 queries stay
// in registers — NT tiles of four scores per lane, T <= 16 NT — so the q.k MFMA, the position-table reads and the 16 position
// multiply-adds of a tile run once instead of three times.  Same score arithmetic, same order of the maximum, of the sum (tiles
// ascending, then the lane's four scores, then the two shuffles) and of the normalisation as the three-sweep kernel above:
// BIT-IDENTICAL weights (both kernels take a score from k2_score; scripts/k2_attw_bits.py compares the two forms' encoder output;
// $RS_K2_ATTW_SWEEPS=3 selects the old one).  Instantiated for NT = 10 / 20 / 40 (T <= 160 / 320 / 640: every stack of a 12.8 s utterance); longer inputs take the
// three-sweep kernel, which has no length limit.
template <int NT>
__global__ __launch_bounds__(256) void k2_attn_weights1_kernel(const uint16_t* __restrict__ qkp, int ld, const float* __restrict__ pos, int cap, int H,
                                                               const int32_t* __restrict__ lens, int T, int Tp, uint16_t* __restrict__ W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* ps = reinterpret_cast<float4*>(smem);
    const int h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int len = lens[b];
    len = len < T ? len : T;
    uint16_t* wbase = W + (((size_t)b * H + h) * T) * Tp;
    if (i0 >= len) {
        for (int idx = threadIdx.x; idx < 64 * (Tp / 4); idx += 256) {
            const int r = idx / (Tp / 4), q = idx - r * (Tp / 4);
            if (i0 + r < T) reinterpret_cast<u16x4_t*>(wbase + (size_t)(i0 + r) * Tp)[q] = (u16x4_t){0, 0, 0, 0};
        }
        return;
    }
    const int nrel = len + 63;
    for (int r = threadIdx.x; r < nrel; r += 256) {
        int n = r - (i0 + 63) + cap - 1;
        n = n < 0 ? 0 : (n > 2 * cap - 2 ? 2 * cap - 2 : n);
        ps[r] = *reinterpret_cast<const float4*>(pos + (size_t)n * (H * K2_PD) + h * K2_PD);
    }
    __syncthreads();
    const int qi = i0 + wave * 16 + (lane & 15);
    const int kq = lane >> 4;
    const bool q_ok = qi < len;
    const int qrow = qi < T ? qi : T - 1;
    const uint16_t* qp = qkp + ((size_t)b * T + qrow) * ld;
    const bf16x8_t qfrag = *reinterpret_cast<const bf16x8_t*>(qp + h * K2_QD + 8 * kq);
    float4 pq;
    {
        const u16x4_t pv = *reinterpret_cast<const u16x4_t*>(qp + 2 * H * K2_QD + h * K2_PD);
        pq = make_float4(bf16_to_f32(pv[0]), bf16_to_f32(pv[1]), bf16_to_f32(pv[2]), bf16_to_f32(pv[3]));
    }
    const int ntile = (len + 15) / 16;
    float s[NT][4];
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        if (jt < ntile) {                                   // (wave-uniform: the MFMA below runs with every lane)
            int krow = jt * 16 + (lane & 15);
            krow = krow < len ? krow : len - 1;
            const bf16x8_t kfrag = *reinterpret_cast<const bf16x8_t*>(qkp + ((size_t)b * T + krow) * ld + H * K2_QD + h * K2_QD + 8 * kq);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qfrag, acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = jt * 16 + 4 * kq + e;
                int r = j - qi + i0 + 63;
                r = r < 0 ? 0 : (r >= nrel ? nrel - 1 : r);
                const float4 pr = ps[r];
                const float v = k2_score(acc[e], pq, pr);
                s[jt][e] = j < len ? v : -INFINITY;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(s[jt][0], s[jt][1]), fmaxf(s[jt][2], s[jt][3])));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[jt][e] = -INFINITY;
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
        if (jt < ntile) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[jt][e] = __expf(s[jt][e] - mx);
                sum += s[jt][e];
            }
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = q_ok ? 1.0f / sum : 0.0f;
    const int ntile_all = Tp / 16;
    if (qi < T) {
        uint16_t* wr = wbase + (size_t)qi * Tp + 4 * kq;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            if (jt < ntile_all) {                          // (no `break`: the loops must unroll completely for s[][] to live in registers)
                u16x4_t o = {0, 0, 0, 0};
                if (jt < ntile) o = pack_bf16x4(s[jt][0] * inv, s[jt][1] * inv, s[jt][2] * inv, s[jt][3] * inv);
                *reinterpret_cast<u16x4_t*>(wr + jt * 16) = o;
            }
        }
    }
}

// ---- weights x values ---------------------------------------------------------------------------------------------------------------
// out[b][i][c] = sum_j W[b][h][i][j] * V[b][j][c]  for the channels of one head (self-attention: 12 of a 16-wide tile) or, MODE 1,
// for 64-channel slices of the non-linear attention with head 0's weights:  V = u[:, hid + c] * tanh(u[:, c]) (rounded to bf16
// like the oracle), result multiplied by u[:, 2 hid + c].  Block = (64 queries, channel tile, utterance x head), 4 waves of 16
// queries; V^T of the block's channels is staged in LDS in chunks of KB keys ([channels][KB + 8] bf16), a step is one MFMA per
// 16-channel tile over 32 keys: first operand V^T rows (channels), second the weight rows (queries), so a lane holds query
// (lane & 15), channels 4 * (lane >> 4) .. + 3 of the tile.
// ---- weights x values: V^T is built ONCE per branch by k2_vt_kernel, the product kernel stages it with 16-byte loads and covers
// 128 queries per block.  (The first form transposed — and, for the non-linear attention, re-evaluated tanh — inside every
// 64-query block: 128 two-byte loads per thread and 256 keys against 32 MFMAs; same values and accumulation order, 1.3 ms per
// batch slower: profiles/r05x_k2_forms_ab.txt.)
//   vT[b][c][key], pitch Tp, C rows per utterance (a multiple of 64): MODE 0  c = 16 h + channel (12 real, 4 zero),
//   MODE 1  c = channel of x * tanh(s); zeros past the utterance's length and past the real channels.
template <int MODE>
__global__ __launch_bounds__(256) void k2_vt_kernel(const uint16_t* __restrict__ u, int ldu, int hid, const int32_t* __restrict__ lens, int T, int Tp,
                                                    int C, uint16_t* __restrict__ vT) {
    constexpr int TP = 72;                               // pitch of the tile in LDS: 144 bytes, rows stay 16-byte aligned
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * TP];
    const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    int len = lens[b];
    len = len < T ? len : T;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int c = idx & 63, key = idx >> 6;
        unsigned short val = 0;
        if (k0 + key < len) {
            const uint16_t* ur = u + ((size_t)b * T + k0 + key) * ldu;
            if constexpr (MODE == 1) {
                if (c0 + c < hid) {
                    const float xv = bf16_to_f32(ur[hid + c0 + c]), sv = bf16_to_f32(ur[c0 + c]);
                    const float th = 1.0f - 2.0f / (__expf(2.0f * sv) + 1.0f);
                    val = f32_to_bf16(xv * th);
                }
            } else {
                const int h = (c0 + c) >> 4, cc = (c0 + c) & 15;
                if (cc < K2_VD && h * K2_VD + cc < hid) val = ur[h * K2_VD + cc];        // (hid = H * 12 here)
            }
        }
        tile[c * TP + key] = val;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 8; idx += 256) {
        const int c = idx >> 3, piece = idx & 7;
        if (k0 + piece * 8 < Tp)
            *reinterpret_cast<u16x8_t*>(vT + ((size_t)b * C + c0 + c) * Tp + k0 + piece * 8) = *reinterpret_cast<const u16x8_t*>(tile + c * TP + piece * 8);
    }
}

template <int CT, int MODE>
__global__ __launch_bounds__(256) void k2_pv_kernel(const uint16_t* __restrict__ W, int H, int Tp, const uint16_t* __restrict__ vT, int C,
                                                     const uint16_t* __restrict__ u, int ldu, int hid, const int32_t* __restrict__ lens, int T,
                                                     uint16_t* __restrict__ out, int ldo) {
    constexpr int KB = 256, PITCH = KB + 8, QT = 2;
    __shared__ __attribute__((aligned(16))) uint16_t vt[CT * 16 * PITCH];
    const int i0 = blockIdx.x * (64 * QT), ctile = blockIdx.y;
    const int b = MODE == 1 ? blockIdx.z : blockIdx.z / H, h = MODE == 1 ? 0 : blockIdx.z % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int len = lens[b];
    len = len < T ? len : T;
    const int nch = MODE == 1 ? hid : K2_VD;
    const int cbase = MODE == 1 ? ctile * CT * 16 : 0;
    const uint16_t* vbase = vT + ((size_t)b * C + (MODE == 1 ? cbase : h * 16)) * Tp;
    const int kq = lane >> 4;
    int qi[QT];
    const uint16_t* wrow[QT];
    f32x4_t acc[QT][CT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        qi[q] = i0 + (wave * QT + q) * 16 + (lane & 15);
        wrow[q] = W + (((size_t)b * H + h) * T + (qi[q] < T ? qi[q] : T - 1)) * Tp;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[q][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    for (int k0 = 0; k0 < len; k0 += KB) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < CT * 16 * (KB / 8); idx += 256) {
            const int c = idx / (KB / 8), piece = idx % (KB / 8);
            u16x8_t v = (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            if (k0 + piece * 8 < Tp) v = *reinterpret_cast<const u16x8_t*>(vbase + (size_t)c * Tp + k0 + piece * 8);
            *reinterpret_cast<u16x8_t*>(vt + c * PITCH + piece * 8) = v;
        }
        __syncthreads();
        const int kend = len - k0 < KB ? len - k0 : KB;
        for (int kk = 0; kk < kend; kk += 32) {
            bf16x8_t vf[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) vf[c] = *reinterpret_cast<const bf16x8_t*>(vt + (c * 16 + (lane & 15)) * PITCH + kk + 8 * kq);
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wrow[q] + k0 + kk + 8 * kq);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[c], wf, acc[q][c], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        if (qi[q] >= T) continue;
        const bool q_ok = qi[q] < len;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int ch = cbase + c * 16 + 4 * kq;          // first of this lane's four channels
            if (ch >= nch) continue;
            float v[4] = {acc[q][c][0], acc[q][c][1], acc[q][c][2], acc[q][c][3]};
            if constexpr (MODE == 1) {
                const u16x4_t y = *reinterpret_cast<const u16x4_t*>(u + ((size_t)b * T + qi[q]) * ldu + 2 * hid + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= bf16_to_f32(y[e]);
            }
            if (!q_ok) v[0] = v[1] = v[2] = v[3] = 0.0f;
            const int oc = MODE == 1 ? ch : h * K2_VD + ch;
            *reinterpret_cast<u16x4_t*>(out + ((size_t)b * T + qi[q]) * ldo + oc) = pack_bf16x4(v[0], v[1], v[2], v[3]);
        }
    }
}


// ---- float32 parity mode (rs_set_option "precision_f32") ----------------------------------------------------------------------------
// The reference's default files are the float32 ONNX graphs (pkg/k2-asr/src/huggingface.py:16,40-45: precision="fp32") and
// onnxruntime computes them in float32.  These kernels are the same encoder with float32 weights, activations and arithmetic end
// to end (dense products on rs_launch_gemm_f32's exact v_mfma_f32_16x16x4_f32 chain, IEEE exp / log1p / tanh / divide), written
// to be read against oracle/zipformer.py statement by statement: one thread or one wave per output, sums in ascending index order,
// nothing tiled, nothing rounded below float32.  Speed is not the object.  Batch semantics as above: every kernel masks by the
// utterance's own length.

// conv0 in float32: feats [B][T][F] -> a0 f32 [B][T - 2][F][C1].  grid (T1, B), block 256
__global__ __launch_bounds__(256) void k2f_conv0_kernel(const float* __restrict__ feats, int T, int F, int C1, const float* __restrict__ w /* [9][C1] */,
                                                        const float* __restrict__ bias, float* __restrict__ out) {
    const int t1 = blockIdx.x, b = blockIdx.y, T1 = T - 2;
    float* orow = out + ((size_t)b * T1 + t1) * F * C1;
    for (int i = threadIdx.x; i < F * C1; i += 256) {
        const int f = i / C1, c = i - f * C1;
        float acc = 0.0f;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const int ff = f + kw - 1;
                const float x = (ff >= 0 && ff < F) ? feats[((size_t)b * T + t1 + kh) * F + ff] : 0.0f;
                acc = fmaf(w[(kh * 3 + kw) * C1 + c], x, acc);
            }
        orow[i] = swoosh_r_exact(acc + bias[c]);
    }
}

// conv1 in float32: a0 [B][T1][F][C1] -> a1 [B][T2][F2][C2], stride 2.  One thread per output, taps (kh, kw, c1) ascending.
__global__ __launch_bounds__(256) void k2f_conv1_kernel(const float* __restrict__ a0, int T1, int F, int C1, int T2, int F2, int C2,
                                                        const float* __restrict__ w /* [3][3][C1][C2] */, const float* __restrict__ bias,
                                                        float* __restrict__ out) {
    const int t2 = blockIdx.x, b = blockIdx.y;
    float* orow = out + ((size_t)b * T2 + t2) * F2 * C2;
    for (int i = threadIdx.x; i < F2 * C2; i += 256) {
        const int f2 = i / C2, c2 = i - f2 * C2;
        float acc = 0.0f;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const float* xr = a0 + (((size_t)b * T1 + 2 * t2 + kh) * F + 2 * f2 + kw) * C1;
                const float* wr = w + ((size_t)(kh * 3 + kw) * C1) * C2 + c2;
                for (int c1 = 0; c1 < C1; ++c1) acc = fmaf(wr[(size_t)c1 * C2], xr[c1], acc);
            }
        orow[i] = swoosh_r_exact(acc + bias[c2]);
    }
}

// patches of conv2 in float32: row (b, t3, f3) = a1[b][t3 + kh][2 f3 + kw][:] in (kh, kw, c) order, zero-padded to Kp = 9 C2 rounded up to 32
__global__ __launch_bounds__(256) void k2f_im2col_kernel(const float* __restrict__ a1, int T2, int F2, int C2, int T3, int F3, int Kp, long long rows,
                                                         float* __restrict__ col) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int f3 = (int)(row % F3);
    const long long bt = row / F3;
    const int t3 = (int)(bt % T3), b = (int)(bt / T3);
    const int c4 = C2 / 4;
    float4* dst = reinterpret_cast<float4*>(col + row * Kp);
    for (int q = lane; q < Kp / 4; q += 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < 9 * c4) {
            const int tap = q / c4, c = q - tap * c4;
            const int kh = tap / 3, kw = tap - 3 * kh;
            v = reinterpret_cast<const float4*>(a1 + (((size_t)b * T2 + t3 + kh) * F2 + 2 * f3 + kw) * C2)[c];
        }
        dst[q] = v;
    }
}

// ConvNeXt depthwise 7 x 7 in float32, one thread per output; rows past the utterance's length and columns outside the map are zeros
__global__ __launch_bounds__(256) void k2f_cnx_dw_kernel(const float* __restrict__ a2, const int32_t* __restrict__ lens3, int T3, int F3, int C,
                                                         const float* __restrict__ w /* [49][C] */, const float* __restrict__ bias,
                                                         float* __restrict__ out) {
    const int t = blockIdx.x, b = blockIdx.y;
    int len = lens3[b];
    len = len < T3 ? len : T3;
    for (int i = threadIdx.x; i < F3 * C; i += 256) {
        const int f = i / C, c = i - f * C;
        float acc = 0.0f;
        for (int kh = 0; kh < 7; ++kh) {
            const int r = t + kh - 3;
            if (r < 0 || r >= len) continue;
            for (int kw = 0; kw < 7; ++kw) {
                const int ff = f + kw - 3;
                if (ff < 0 || ff >= F3) continue;
                acc = fmaf(w[(kh * 7 + kw) * C + c], a2[(((size_t)b * T3 + r) * F3 + ff) * C + c], acc);
            }
        }
        out[(((size_t)b * T3 + t) * F3 + f) * C + c] = acc + bias[c];
    }
}

// attention weights in float32: one wave per (query, head, utterance); a lane owns keys lane, lane + 64, ..
//   s[i][j] = q_i . k_j + p_i . pos[j - i];  W[b][h][i][:] = softmax over the utterance's own keys; padded queries / keys: 0
// qkp f32 [B*T][ld] as in the bf16 kernel; W f32 [B][H][T][Tp].  grid (ceil(T / 4), H, B), block 256
__global__ __launch_bounds__(256) void k2f_attn_weights_kernel(const float* __restrict__ qkp, int ld, const float* __restrict__ pos, int cap, int H,
                                                               const int32_t* __restrict__ lens, int T, int Tp, float* __restrict__ W) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z;
    if (i >= T) return;
    int len = lens[b];
    len = len < T ? len : T;
    float* wrow = W + (((size_t)b * H + h) * T + i) * Tp;
    if (i >= len) {
        for (int j = lane; j < Tp; j += 64) wrow[j] = 0.0f;
        return;
    }
    const float* qr = qkp + ((size_t)b * T + i) * ld;
    float q[K2_QD], pq[K2_PD];
#pragma unroll
    for (int e = 0; e < K2_QD; ++e) q[e] = qr[h * K2_QD + e];
#pragma unroll
    for (int e = 0; e < K2_PD; ++e) pq[e] = qr[2 * H * K2_QD + h * K2_PD + e];
    float mx = -INFINITY;
    for (int j = lane; j < len; j += 64) {
        const float* kr = qkp + ((size_t)b * T + j) * ld + H * K2_QD + h * K2_QD;
        const float* pr = pos + (size_t)(j - i + cap - 1) * (H * K2_PD) + h * K2_PD;
        float qk = 0.0f, pp = 0.0f;
#pragma unroll
        for (int e = 0; e < K2_QD; ++e) qk = fmaf(q[e], kr[e], qk);
#pragma unroll
        for (int e = 0; e < K2_PD; ++e) pp = fmaf(pq[e], pr[e], pp);
        const float sc = qk + pp;
        wrow[j] = sc;
        mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j < len; j += 64) {          // (a lane re-reads only what it wrote itself)
        const float e = expf(wrow[j] - mx);
        wrow[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < Tp; j += 64) wrow[j] = j < len ? wrow[j] / sum : 0.0f;
}

// self-attention values in float32: out[b][i][h * 12 + c] = sum_j W[b][h][i][j] v[b][j][h * 12 + c].  One thread per output column
// of one query, keys ascending.  v f32 [B*T][ldv], out f32 [B*T][ldo] (columns >= H * 12 are left alone: zeroed by the caller).
// grid (T, B), block = H * 12 rounded up to 64
__global__ void k2f_pv_kernel(const float* __restrict__ W, int H, int Tp, const float* __restrict__ v, int ldv, const int32_t* __restrict__ lens,
                              int T, float* __restrict__ out, int ldo) {
    const int i = blockIdx.x, b = blockIdx.y, col = threadIdx.x;
    if (col >= H * K2_VD) return;
    int len = lens[b];
    len = len < T ? len : T;
    float acc = 0.0f;
    if (i < len) {
        const float* wrow = W + (((size_t)b * H + col / K2_VD) * T + i) * Tp;
        const float* vc = v + (size_t)b * T * ldv + col;
        for (int j = 0; j < len; ++j) acc = fmaf(wrow[j], vc[(size_t)j * ldv], acc);
    }
    out[((size_t)b * T + i) * ldo + col] = acc;
}

// non-linear attention in float32, first half: g[b][j][c] = u[.., hid + c] * tanh(u[.., c])   (u = in_proj output [B*T][3 hid])
__global__ __launch_bounds__(256) void k2f_na_gate_kernel(const float* __restrict__ u, int hid, size_t rows, float* __restrict__ g) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * hid) return;
    const size_t r = idx / hid;
    const int c = (int)(idx - r * hid);
    const float* ur = u + r * 3 * hid;
    g[idx] = ur[hid + c] * tanhf(ur[c]);
}

// second half: out[b][i][c] = (sum_j W[b][0][i][j] g[b][j][c]) * u[b][i][2 hid + c]; four queries per thread (they share the
// loads of g), keys ascending.  grid (ceil(hid / 256), ceil(T / 4), B), block 256
__global__ __launch_bounds__(256) void k2f_na_pv_kernel(const float* __restrict__ W, int H, int Tp, const float* __restrict__ g, const float* __restrict__ u,
                                                        int hid, const int32_t* __restrict__ lens, int T, float* __restrict__ out, int ldo) {
    const int c = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * 4, b = blockIdx.z;
    if (c >= hid) return;
    int len = lens[b];
    len = len < T ? len : T;
    const float* wr[4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) wr[q] = W + (((size_t)b * H + 0) * T + (i0 + q < T ? i0 + q : T - 1)) * Tp;
    const float* gc = g + (size_t)b * T * hid + c;
    for (int j = 0; j < len; ++j) {
        const float gv = gc[(size_t)j * hid];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fmaf(wr[q][j], gv, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + q;
        if (i >= T) continue;
        out[((size_t)b * T + i) * ldo + c] = i < len ? acc[q] * u[((size_t)b * T + i) * 3 * hid + 2 * hid + c] : 0.0f;
    }
}

// conv-module middle in float32: x [B*T][2d] (values | gates, icefall's own order) -> x * sigmoid(gate) -> frame mask -> depthwise k
// -> + bias -> SwooshR -> out [B*T][d].  One thread per output, taps ascending (absent taps skipped)
__global__ __launch_bounds__(256) void k2f_glu_dwconv_swoosh_kernel(const float* __restrict__ x, const float* __restrict__ w /* [k][d] */,
                                                                    const float* __restrict__ bias, const int32_t* __restrict__ lens, int T, int d,
                                                                    int k, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, b = blockIdx.z;
    if (c >= d) return;
    int len = lens[b];
    len = len < T ? len : T;
    const int half = (k - 1) >> 1;
    float acc = 0.0f;
    for (int j = 0; j < k; ++j) {
        const int tj = t + j - half;
        if (tj < 0 || tj >= len) continue;
        const float* px = x + ((size_t)b * T + tj) * 2 * d;
        acc = fmaf(w[(size_t)j * d + c], px[c] * (1.0f / (1.0f + expf(-px[d + c]))), acc);
    }
    out[((size_t)b * T + t) * d + c] = swoosh_r_exact(acc + bias[c]);
}

// per-stack lengths: l[s][b] = ceil(len3[b] / ds[s]); len3[b] = max((n_frames[b] - 7) / 2, 0).  rows: 0 = len3, 1 + s = stack s
__global__ void k2_lens_kernel(const int32_t* __restrict__ n_frames, int B, int n_stacks, int ds0, int ds1, int ds2, int ds3, int ds4, int ds5, int ds6,
                               int ds7, int32_t* __restrict__ out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int ds[8] = {ds0, ds1, ds2, ds3, ds4, ds5, ds6, ds7};
    int l3 = (n_frames[b] - 7) / 2;
    l3 = l3 > 0 ? l3 : 0;
    out[b] = l3;
    for (int s = 0; s < n_stacks; ++s) out[(size_t)(1 + s) * B + b] = (l3 + ds[s] - 1) / ds[s];
}

struct K2Plan {
    int T, T1, T2, T3, To, F, F2, F3, Kp;
    int Ts[8], Tp[8];
    size_t off_lens, off_a0, off_a1, off_col, off_a2, off_dwo, off_h, off_stackout[8], off_x, off_x0, off_src, off_xb, off_qkp, off_w, off_big, off_av, off_vt,
        off_encb, total;
};

K2Plan k2_plan(const rs_ctx* ctx, int B, int t_max) {
    const rs_k2& k = *ctx->k2;
    const rs_k2_dims& d = k.d;
    K2Plan p{};
    p.T = t_max; p.F = d.n_mels;
    p.T1 = t_max - 2; p.T2 = p.T1 >= 3 ? (p.T1 - 3) / 2 + 1 : 0; p.T3 = p.T2 - 2;
    if (p.T3 < 0) p.T3 = 0;
    p.F2 = (p.F - 3) / 2 + 1; p.F3 = (p.F2 - 3) / 2 + 1;
    p.To = (p.T3 + 1) / 2;
    p.Kp = pad64(9 * d.embed_c2);
    const size_t T3 = p.T3 > 0 ? p.T3 : 1, T1 = p.T1 > 0 ? p.T1 : 1, T2 = p.T2 > 0 ? p.T2 : 1;
    const size_t rows3 = (size_t)B * T3 * p.F3;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += rs_align(bytes); return at; };
    p.off_lens = take((size_t)(2 + d.n_stacks) * B * 4);
    p.off_a0 = take((size_t)B * T1 * p.F * d.embed_c1 * 2);
    p.off_a1 = take((size_t)B * T2 * p.F2 * d.embed_c2 * 2);
    p.off_col = take(rows3 * p.Kp * 2);
    p.off_a2 = take(rows3 * d.embed_c3 * 4);
    p.off_dwo = take(rows3 * d.embed_c3 * 2);
    p.off_h = take(rows3 * 3 * d.embed_c3 * 2);
    int dmax = 0, nin_max = 0, big_max = 0, av_max = 0;
    size_t w_max = 0;
    for (int s = 0; s < d.n_stacks; ++s) {
        const int dd = d.encoder_dim[s], H = d.num_heads[s], ds = d.downsampling[s];
        p.Ts[s] = ceil_div((int)T3, ds);
        p.Tp[s] = (p.Ts[s] + 31) / 32 * 32;
        p.off_stackout[s] = take((size_t)B * T3 * dd * 4);
        dmax = dd > dmax ? dd : dmax;
        const int nin = (2 * K2_QD + K2_PD) * H;
        const size_t rows = (size_t)B * p.Ts[s];
        if ((size_t)nin * rows > (size_t)nin_max) nin_max = 0;       // (sized below in bytes)
        const size_t wb = (size_t)B * H * p.Ts[s] * p.Tp[s] * 2;
        w_max = wb > w_max ? wb : w_max;
        (void)big_max; (void)av_max;
    }
    size_t x_b = 0, qkp_b = 0, big_b = 0, av_b = 0, vt_b = 0;
    for (int s = 0; s < d.n_stacks; ++s) {
        const size_t rows = (size_t)B * p.Ts[s], dd = d.encoder_dim[s], H = d.num_heads[s];
        const size_t full = (size_t)B * T3 * dd * 4;
        x_b = full > x_b ? full : x_b;
        const size_t q = rows * (2 * K2_QD + K2_PD) * H * 2;
        qkp_b = q > qkp_b ? q : qkp_b;
        size_t widest = (size_t)d.ff_dim[s] * 5 / 4;
        const size_t na = 3 * (3 * dd / 4);
        widest = na > widest ? na : widest;
        widest = 2 * dd > widest ? 2 * dd : widest;
        big_b = rows * widest * 2 > big_b ? rows * widest * 2 : big_b;
        size_t avw = pad64(3 * (int)dd / 4);
        avw = (size_t)pad64((int)H * K2_VD) > avw ? (size_t)pad64((int)H * K2_VD) : avw;
        avw = dd > avw ? dd : avw;
        av_b = rows * avw * 2 > av_b ? rows * avw * 2 : av_b;
        size_t vc = pad64(3 * (int)dd / 4);
        vc = (size_t)pad64((int)H * 16) > vc ? (size_t)pad64((int)H * 16) : vc;
        const size_t vb = (size_t)B * vc * p.Tp[s] * 2;
        vt_b = vb > vt_b ? vb : vt_b;
    }
    p.off_x = take(x_b); p.off_x0 = take(x_b); p.off_src = take(x_b);
    p.off_xb = take(x_b / 2);
    p.off_qkp = take(qkp_b);
    p.off_w = take(w_max);
    p.off_big = take(big_b);
    p.off_av = take(av_b);
    p.off_vt = take(vt_b);
    p.off_encb = take((size_t)B * (p.To > 0 ? p.To : 1) * k.out_dim * 2);
    p.total = o + 256;
    return p;
}


__host__ __device__ inline int pad32(int n) { return (n + 31) / 32 * 32; }

// workspace of the float32 parity mode: the bf16 plan's tensors as float32, patches un-padded (K = 9 C2), no V^T copies
struct K2PlanF32 {
    int T, T1, T2, T3, To, F, F2, F3;
    int Ts[8], Tp[8];
    size_t off_lens, off_a0, off_a1, off_col, off_a2, off_dwo, off_h, off_stackout[8], off_x, off_x0, off_src, off_qkp, off_w, off_big, off_av, off_gate,
        off_enc, total;
};

K2PlanF32 k2_plan_f32(const rs_ctx* ctx, int B, int t_max) {
    const rs_k2& k = *ctx->k2;
    const rs_k2_dims& d = k.d;
    K2PlanF32 p{};
    p.T = t_max; p.F = d.n_mels;
    p.T1 = t_max - 2; p.T2 = p.T1 >= 3 ? (p.T1 - 3) / 2 + 1 : 0; p.T3 = p.T2 - 2;
    if (p.T3 < 0) p.T3 = 0;
    p.F2 = (p.F - 3) / 2 + 1; p.F3 = (p.F2 - 3) / 2 + 1;
    p.To = (p.T3 + 1) / 2;
    const size_t T3 = p.T3 > 0 ? p.T3 : 1, T1 = p.T1 > 0 ? p.T1 : 1, T2 = p.T2 > 0 ? p.T2 : 1;
    const size_t rows3 = (size_t)B * T3 * p.F3;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += rs_align(bytes); return at; };
    p.off_lens = take((size_t)(2 + d.n_stacks) * B * 4);
    p.off_a0 = take((size_t)B * T1 * p.F * d.embed_c1 * 4);
    p.off_a1 = take((size_t)B * T2 * p.F2 * d.embed_c2 * 4);
    p.off_col = take(rows3 * pad32(9 * d.embed_c2) * 4);
    p.off_a2 = take(rows3 * d.embed_c3 * 4);
    p.off_dwo = take(rows3 * d.embed_c3 * 4);
    p.off_h = take(rows3 * 3 * d.embed_c3 * 4);
    size_t x_b = 0, qkp_b = 0, w_b = 0, big_b = 0, av_b = 0, gate_b = 0;
    for (int s = 0; s < d.n_stacks; ++s) {
        const size_t dd = d.encoder_dim[s], H = d.num_heads[s], hid = 3 * dd / 4;
        p.Ts[s] = ceil_div((int)T3, d.downsampling[s]);
        p.Tp[s] = (p.Ts[s] + 31) / 32 * 32;
        const size_t rows = (size_t)B * p.Ts[s];
        p.off_stackout[s] = take((size_t)B * T3 * dd * 4);
        x_b = std::max(x_b, (size_t)B * T3 * dd * 4);
        qkp_b = std::max(qkp_b, rows * (2 * K2_QD + K2_PD) * H * 4);
        w_b = std::max(w_b, (size_t)B * H * p.Ts[s] * p.Tp[s] * 4);
        const size_t widest = std::max(std::max((size_t)d.ff_dim[s] * 5 / 4, 3 * hid), 2 * dd);
        big_b = std::max(big_b, rows * widest * 4);
        const size_t avw = std::max(std::max((size_t)pad32((int)hid), (size_t)pad32((int)H * K2_VD)), dd);
        av_b = std::max(av_b, rows * avw * 4);
        gate_b = std::max(gate_b, rows * hid * 4);
    }
    p.off_x = take(x_b); p.off_x0 = take(x_b); p.off_src = take(x_b);
    p.off_qkp = take(qkp_b);
    p.off_w = take(w_b);
    p.off_big = take(big_b);
    p.off_av = take(av_b);
    p.off_gate = take(gate_b);
    p.off_enc = take((size_t)B * (p.To > 0 ? p.To : 1) * k.out_dim * 4);
    p.total = o + 256;
    return p;
}

template <typename T>
int k2_get(rs_ctx* ctx, const std::string& name, size_t elems, const T*& out) {
    auto it = ctx->tensors.find(name);
    if (it == ctx->tensors.end()) return rs_fail(ctx, RS_EMISSING, "weight tensor '%s' was not registered", name.c_str());
    if (it->second.second != elems * sizeof(T))
        return rs_fail(ctx, RS_EINVAL, "tensor '%s': expected %zu bytes, got %zu", name.c_str(), elems * sizeof(T), it->second.second);
    if ((uintptr_t)it->second.first & 15) return rs_fail(ctx, RS_EINVAL, "tensor '%s' is not 16-byte aligned", name.c_str());
    out = reinterpret_cast<const T*>(it->second.first);
    return RS_OK;
}

void k2_free(rs_k2* k) { delete k; }

}  // namespace

extern "C" int rs_k2_create(rs_ctx** out, int device, const rs_k2_dims* dims) {
    if (!out || !dims) return RS_EINVAL;
    *out = nullptr;
    rs_ctx* ctx = new (std::nothrow) rs_ctx();
    if (!ctx) return RS_EINVAL;
    *out = ctx;
    ctx->device = device;
    rs_k2* k = new (std::nothrow) rs_k2();
    if (!k) return rs_fail(ctx, RS_EINVAL, "out of memory");
    ctx->k2 = k;
    ctx->k2_free = k2_free;
    k->d = *dims;
    const rs_k2_dims& d = k->d;
    if (d.n_stacks < 1 || d.n_stacks > 8) return rs_fail(ctx, RS_EINVAL, "zipformer: 1..8 stacks");
    if (d.query_head_dim != K2_QD || d.pos_head_dim != K2_PD || d.value_head_dim != K2_VD)
        return rs_fail(ctx, RS_EINVAL, "zipformer: the attention kernels are built for head dims 32 / 4 / 12 (got %d / %d / %d)", d.query_head_dim,
                       d.pos_head_dim, d.value_head_dim);
    if (d.n_mels < 16 || d.n_mels > 128 || d.n_mels % 4 || d.frame_length > 512 || d.frame_shift < 1 || d.frame_shift > 256)
        return rs_fail(ctx, RS_EINVAL, "zipformer: feature geometry (n_mels %d, frame %d / %d)", d.n_mels, d.frame_length, d.frame_shift);
    if (d.embed_c1 < 1 || d.embed_c2 % 8 || d.embed_c3 % 64 || d.embed_c1 * d.n_mels > 4096) return rs_fail(ctx, RS_EINVAL, "zipformer: encoder_embed channels");
    if (d.downsampling[0] != 1) return rs_fail(ctx, RS_EINVAL, "zipformer: the first stack runs at the full rate");
    k->out_dim = 0;
    for (int s = 0; s < d.n_stacks; ++s) {
        const int ds = d.downsampling[s], kk = d.cnn_kernel[s];
        if (d.encoder_dim[s] % 64 || d.ff_dim[s] % 256 || d.num_heads[s] < 1 || d.num_heads[s] > 16 || d.num_layers[s] < 1)
            return rs_fail(ctx, RS_EINVAL, "zipformer: stack %d: encoder_dim %% 64, feedforward_dim %% 256, 1..16 heads", s);
        if (ds != 1 && ds != 2 && ds != 4 && ds != 8) return rs_fail(ctx, RS_EINVAL, "zipformer: stack %d: down-sampling %d", s, ds);
        if (kk != 7 && kk != 15 && kk != 31) return rs_fail(ctx, RS_EINVAL, "zipformer: stack %d: cnn_module_kernel %d (7, 15, 31 are built)", s, kk);
        k->out_dim = d.encoder_dim[s] > k->out_dim ? d.encoder_dim[s] : k->out_dim;
    }
    if (d.decoder_dim % 128 || d.joiner_dim % 128 || d.context_size != 2 || d.blank_id != 0 || d.vocab_size < 2)
        return rs_fail(ctx, RS_EINVAL, "zipformer: decoder_dim / joiner_dim %% 128, context_size 2, blank 0");
    k->embed_freq = (((d.n_mels - 1) / 2) - 1) / 2;
    // the shared stages (front-end, greedy search) read the dimensions they need from rs_dims
    rs_dims& g = ctx->d;
    g.n_mels = d.n_mels; g.n_fft = 512; g.win_length = d.frame_length; g.hop_length = d.frame_shift; g.preemph = d.preemph;
    g.log_guard = 1.1920928955078125e-07f; g.norm_eps = 0.0f;
    g.frontend_kind = 2;
    g.d_model = k->out_dim; g.n_logits = d.vocab_size; g.blank_id = d.blank_id; g.pred_hidden = d.decoder_dim; g.pred_layers = 1;
    g.joint_hidden = d.joiner_dim; g.max_symbols = 1; g.joint_act = 1;
    if (hipSetDevice(device) != hipSuccess) return rs_fail(ctx, RS_EHIP, "hipSetDevice(%d) failed", device);
    return RS_OK;
}

extern "C" int rs_k2_encoder_set_taps(rs_ctx* ctx, float* embed_out, float* stack_out) {
    if (!ctx || !ctx->k2) return RS_EINVAL;
    ctx->k2->tap_embed = embed_out;
    ctx->k2->tap_stacks = stack_out;
    return RS_OK;
}

int rs_k2_finalize_impl(rs_ctx* ctx) {
    rs_k2& k = *ctx->k2;
    const rs_k2_dims& d = k.d;
    int rc;
#define K2_GET(name, elems, field) do { rc = k2_get(ctx, name, (size_t)(elems), field); if (rc != RS_OK) return rc; } while (0)
    K2_GET("fe.window", d.frame_length, ctx->fe_window);
    K2_GET("fe.twiddle", 512, ctx->fe_twiddle);
    K2_GET("fe.fb_idx", d.n_mels * 2, ctx->fe_fb_idx);
    K2_GET("fe.fb_w", d.n_mels * 32, ctx->fe_fb_w);
    const int c1 = d.embed_c1, c2 = d.embed_c2, c3 = d.embed_c3, d0 = d.encoder_dim[0];
    K2_GET("emb.conv0.w", 9 * c1, k.conv0_w); K2_GET("emb.conv0.b", c1, k.conv0_b);
    K2_GET("emb.conv1.w", 9 * c1 * c2, k.conv1_w); K2_GET("emb.conv1.b", c2, k.conv1_b);
    K2_GET("emb.conv2.w", (size_t)c3 * pad64(9 * c2), k.conv2_w); K2_GET("emb.conv2.b", c3, k.conv2_b);
    K2_GET("emb.cnx.dw.w", 49 * c3, k.cnx_dw_w); K2_GET("emb.cnx.dw.b", c3, k.cnx_dw_b);
    K2_GET("emb.cnx.pw1.w", 3 * c3 * c3, k.cnx_pw1_w); K2_GET("emb.cnx.pw1.b", 3 * c3, k.cnx_pw1_b);
    K2_GET("emb.cnx.pw2.w", 3 * c3 * c3, k.cnx_pw2_w); K2_GET("emb.cnx.pw2.b", c3, k.cnx_pw2_b);
    K2_GET("emb.out.w", (size_t)d0 * k.embed_freq * c3, k.emb_out_w); K2_GET("emb.out.b", d0, k.emb_out_b);
    K2_GET("emb.norm.bias", d0, k.emb_norm_bias); K2_GET("emb.norm.scale", 4, k.emb_norm_scale);
    // the position tables all have 2 * cap - 1 rows
    {
        auto it = ctx->tensors.find("S0.L0.attw.pos_proj");
        if (it == ctx->tensors.end()) return rs_fail(ctx, RS_EMISSING, "weight tensor 'S0.L0.attw.pos_proj' was not registered");
        const size_t row = (size_t)d.num_heads[0] * K2_PD * 4;
        if (it->second.second % row || !((it->second.second / row) & 1)) return rs_fail(ctx, RS_EINVAL, "attw.pos_proj must be f32 [2*cap-1][H*4]");
        k.pos_cap = (int)((it->second.second / row + 1) / 2);
    }
    k.stacks.assign(d.n_stacks, {});
    for (int s = 0; s < d.n_stacks; ++s) {
        const size_t dd = d.encoder_dim[s], H = d.num_heads[s], hid = 3 * dd / 4, kk = d.cnn_kernel[s];
        const size_t ff[3] = {(size_t)d.ff_dim[s] * 3 / 4, (size_t)d.ff_dim[s], (size_t)d.ff_dim[s] * 5 / 4};
        k.stacks[s].assign(d.num_layers[s], rs_k2_layer{});
        for (int j = 0; j < d.num_layers[s]; ++j) {
            rs_k2_layer& L = k.stacks[s][j];
            const std::string p = "S" + std::to_string(s) + ".L" + std::to_string(j) + ".";
            const size_t nin = (2 * K2_QD + K2_PD) * H;
            K2_GET(p + "attw.in.w", nin * dd, L.attw_in_w); K2_GET(p + "attw.in.b", nin, L.attw_in_b);
            K2_GET(p + "attw.pos_proj", (size_t)(2 * k.pos_cap - 1) * H * K2_PD, L.pos_proj);
            for (int a = 0; a < 2; ++a) {
                const std::string q = p + (a ? "sa2." : "sa1.");
                K2_GET(q + "in.w", H * K2_VD * dd, L.sa_in_w[a]); K2_GET(q + "in.b", H * K2_VD, L.sa_in_b[a]);
                K2_GET(q + "out.w", dd * pad64((int)(H * K2_VD)), L.sa_out_w[a]); K2_GET(q + "out.b", dd, L.sa_out_b[a]);
                const std::string c = p + (a ? "cm2." : "cm1.");
                K2_GET(c + "in.w", 2 * dd * dd, L.cm_in_w[a]); K2_GET(c + "in.b", 2 * dd, L.cm_in_b[a]);
                K2_GET(c + "dw.w", kk * dd, L.cm_dw_w[a]); K2_GET(c + "dw.b", dd, L.cm_dw_b[a]);
                K2_GET(c + "out.w", dd * dd, L.cm_out_w[a]); K2_GET(c + "out.b", dd, L.cm_out_b[a]);
            }
            for (int f = 0; f < 3; ++f) {
                const std::string q = p + "ff" + std::to_string(f + 1) + ".";
                K2_GET(q + "in.w", ff[f] * dd, L.ff_in_w[f]); K2_GET(q + "in.b", ff[f], L.ff_in_b[f]);
                K2_GET(q + "out.w", dd * ff[f], L.ff_out_w[f]); K2_GET(q + "out.b", dd, L.ff_out_b[f]);
            }
            K2_GET(p + "na.in.w", 3 * hid * dd, L.na_in_w); K2_GET(p + "na.in.b", 3 * hid, L.na_in_b);
            K2_GET(p + "na.out.w", dd * pad64((int)hid), L.na_out_w); K2_GET(p + "na.out.b", dd, L.na_out_b);
            K2_GET(p + "norm.bias", dd, L.norm_bias); K2_GET(p + "norm.scale", 4, L.norm_scale);
            K2_GET(p + "bypass.scale", dd, L.bypass); K2_GET(p + "bypass_mid.scale", dd, L.bypass_mid);
        }
        if (d.downsampling[s] > 1) {
            K2_GET("S" + std::to_string(s) + ".ds.w", 8, k.ds_w[s]);
            K2_GET("S" + std::to_string(s) + ".comb.scale", dd, k.comb_scale[s]);
        }
    }
    K2_GET("out.ds.w", 8, k.out_ds_w);
    const size_t J = d.joiner_dim, D = d.decoder_dim, V = d.vocab_size;
    K2_GET("joint.enc.w", J * k.out_dim, ctx->jenc_w); K2_GET("joint.enc.b", J, ctx->jenc_b);
    K2_GET("dec.embed", V * D, ctx->embed);
    K2_GET("dec.conv.w", D * 4 * 2, ctx->k2_conv_w);
    K2_GET("joint.pred.w", J * D, ctx->jpred_w); K2_GET("joint.pred.b", J, ctx->jpred_b);
    K2_GET("joint.out.w", ((V + 15) / 16 * 16) * J, ctx->jout_w); K2_GET("joint.out.b", V, ctx->jout_b);
    // optional: the screened joint's operands (all four or none), as for the other families
    ctx->jout_w16 = nullptr; ctx->jout_wrm = ctx->jout_bpad = ctx->jout_wmax = nullptr;
    if (ctx->tensors.count("joint.out.w16") || ctx->tensors.count("joint.out.wrm") || ctx->tensors.count("joint.out.bpad") || ctx->tensors.count("joint.out.wmax")) {
        const size_t Vpad = (V + 15) / 16 * 16;
        K2_GET("joint.out.w16", Vpad * J, ctx->jout_w16);
        K2_GET("joint.out.wrm", V * J, ctx->jout_wrm);
        K2_GET("joint.out.bpad", Vpad, ctx->jout_bpad);
        K2_GET("joint.out.wmax", 4, ctx->jout_wmax);
    }
    // optional: the float32 parity mode's dense weights ("<name>.f32": all or none)
    ctx->has_f32 = false;
    k.stacks32.clear();
    if (ctx->tensors.count("emb.out.w.f32")) {
        K2_GET("emb.conv2.w.f32", (size_t)c3 * pad32(9 * c2), k.conv2_w32);
        K2_GET("emb.cnx.pw1.w.f32", 3 * c3 * c3, k.cnx_pw1_w32); K2_GET("emb.cnx.pw2.w.f32", 3 * c3 * c3, k.cnx_pw2_w32);
        K2_GET("emb.out.w.f32", (size_t)d0 * k.embed_freq * c3, k.emb_out_w32);
        K2_GET("joint.enc.w.f32", J * k.out_dim, k.jenc_w32);
        k.stacks32.assign(d.n_stacks, {});
        for (int s = 0; s < d.n_stacks; ++s) {
            const size_t dd = d.encoder_dim[s], H = d.num_heads[s], hid = 3 * dd / 4;
            const size_t ff[3] = {(size_t)d.ff_dim[s] * 3 / 4, (size_t)d.ff_dim[s], (size_t)d.ff_dim[s] * 5 / 4};
            k.stacks32[s].assign(d.num_layers[s], rs_k2_layer32{});
            for (int j = 0; j < d.num_layers[s]; ++j) {
                rs_k2_layer32& L = k.stacks32[s][j];
                const std::string p = "S" + std::to_string(s) + ".L" + std::to_string(j) + ".";
                K2_GET(p + "attw.in.w.f32", (2 * K2_QD + K2_PD) * H * dd, L.attw_in_w);
                for (int a = 0; a < 2; ++a) {
                    const std::string q = p + (a ? "sa2." : "sa1."), c = p + (a ? "cm2." : "cm1.");
                    K2_GET(q + "in.w.f32", H * K2_VD * dd, L.sa_in_w[a]);
                    K2_GET(q + "out.w.f32", dd * pad32((int)(H * K2_VD)), L.sa_out_w[a]);
                    K2_GET(c + "in.w.f32", 2 * dd * dd, L.cm_in_w[a]); K2_GET(c + "in.b.f32", 2 * dd, L.cm_in_b[a]);
                    K2_GET(c + "out.w.f32", dd * dd, L.cm_out_w[a]);
                }
                for (int f = 0; f < 3; ++f) {
                    const std::string q = p + "ff" + std::to_string(f + 1) + ".";
                    K2_GET(q + "in.w.f32", ff[f] * dd, L.ff_in_w[f]); K2_GET(q + "out.w.f32", dd * ff[f], L.ff_out_w[f]);
                }
                K2_GET(p + "na.in.w.f32", 3 * hid * dd, L.na_in_w);
                K2_GET(p + "na.out.w.f32", dd * pad32((int)hid), L.na_out_w);
            }
        }
        ctx->has_f32 = true;
    }
    if (ctx->precision_f32 && !ctx->has_f32) ctx->precision_f32 = 0;
#undef K2_GET
    ctx->decode_narrow = true;
    ctx->finalized = true;
    return RS_OK;
}

int rs_k2_unk_id(const rs_ctx* ctx) { return ctx->k2 ? ctx->k2->d.unk_id : -1; }

int rs_k2_enc_frames_impl(const rs_ctx* ctx, int n_feat) {
    (void)ctx;
    const int t3 = (n_feat - 7) / 2;
    return t3 > 0 ? (t3 + 1) / 2 : 0;
}

size_t rs_k2_workspace_bytes_impl(const rs_ctx* ctx, int B, int t_max) {
    const size_t a = k2_plan(ctx, B, t_max > 9 ? t_max : 9).total;
    if (!ctx->has_f32) return a;                      // the float32 parity mode keeps float32 activations: about twice the scratch
    const size_t b = k2_plan_f32(ctx, B, t_max > 9 ? t_max : 9).total;
    return a > b ? a : b;
}

static int rs_k2_encoder_forward_f32(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max, float* enc_out, float* joint_enc,
                                     int32_t* enc_lens, void* workspace, size_t workspace_bytes, hipStream_t s);

int rs_k2_encoder_forward_impl(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max, float* enc_out, float* joint_enc,
                               int32_t* enc_lens, void* workspace, size_t workspace_bytes, hipStream_t s) {
    rs_k2& k = *ctx->k2;
    const rs_k2_dims& d = k.d;
    if (t_max < 9) return rs_fail(ctx, RS_EINVAL, "zipformer: %d feature frames are too few for encoder_embed (9 are needed)", t_max);
    if (ctx->precision_f32) return rs_k2_encoder_forward_f32(ctx, feats, n_frames, B, t_max, enc_out, joint_enc, enc_lens, workspace, workspace_bytes, s);
    const K2Plan pl = k2_plan(ctx, B, t_max);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "zipformer: workspace %zu < %zu", workspace_bytes, pl.total);
    char* ws = reinterpret_cast<char*>(workspace);
    int32_t* lens_all = reinterpret_cast<int32_t*>(ws + pl.off_lens);
    const int32_t* lens3 = lens_all;
    uint16_t* a0 = reinterpret_cast<uint16_t*>(ws + pl.off_a0);
    uint16_t* a1 = reinterpret_cast<uint16_t*>(ws + pl.off_a1);
    uint16_t* col = reinterpret_cast<uint16_t*>(ws + pl.off_col);
    float* a2 = reinterpret_cast<float*>(ws + pl.off_a2);
    uint16_t* dwo = reinterpret_cast<uint16_t*>(ws + pl.off_dwo);
    uint16_t* hbuf = reinterpret_cast<uint16_t*>(ws + pl.off_h);
    float* x = reinterpret_cast<float*>(ws + pl.off_x);
    float* x0 = reinterpret_cast<float*>(ws + pl.off_x0);
    float* src = reinterpret_cast<float*>(ws + pl.off_src);
    uint16_t* xb = reinterpret_cast<uint16_t*>(ws + pl.off_xb);
    uint16_t* qkp = reinterpret_cast<uint16_t*>(ws + pl.off_qkp);
    uint16_t* W = reinterpret_cast<uint16_t*>(ws + pl.off_w);
    uint16_t* big = reinterpret_cast<uint16_t*>(ws + pl.off_big);
    uint16_t* av = reinterpret_cast<uint16_t*>(ws + pl.off_av);
    uint16_t* vT = reinterpret_cast<uint16_t*>(ws + pl.off_vt);
    uint16_t* encb = reinterpret_cast<uint16_t*>(ws + pl.off_encb);
    const int c1 = d.embed_c1, c2 = d.embed_c2, c3 = d.embed_c3, T3 = pl.T3, F3 = pl.F3;
    int rc;
#define RS_TRY(call) do { rc = (call); if (rc != RS_OK) return rc; } while (0)
    // `copy`: the residual GEMM also stores its result rounded to bf16 (row pitch N) — the A operand of the branch that follows
    auto gemm = [&](const uint16_t* A, int lda, const uint16_t* Wt, int K, void* out, int ldc, long long M, int N, int flags, const float* bias,
                    const float* res, uint16_t* copy = nullptr) -> int {
        rs_gemm_args g{};
        g.A = A; g.lda = lda; g.W = Wt; g.ldw = K; g.out = out; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K;
        g.flags = flags; g.bias = bias; g.alpha = 1.0f; g.residual = res;
        g.out_bf16 = copy; g.ld_bf16 = N;
        return rs_launch_gemm(ctx, g, s);
    };
    const int RES = RS_GEMM_BIAS | RS_GEMM_RESIDUAL | RS_GEMM_OUT_F32;
    auto cast = [&](const float* in, uint16_t* out, size_t n) {
        hipLaunchKernelGGL(k2_cast_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, in, out, n / 4);
    };
    hipLaunchKernelGGL(k2_lens_kernel, dim3((B + 255) / 256), dim3(256), 0, s, n_frames, B, d.n_stacks, d.downsampling[0], d.downsampling[1],
                       d.downsampling[2], d.downsampling[3], d.downsampling[4], d.downsampling[5], d.downsampling[6], d.downsampling[7], lens_all);
    // ---- encoder_embed ------------------------------------------------------------------------------------------------
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, 0.0, 0.0);
    hipLaunchKernelGGL(k2_conv0_kernel, dim3(pl.T1, B), dim3(256), 0, s, feats, t_max, pl.F, c1, k.conv0_w, k.conv0_b, a0);
    static const bool conv1_first_form = getenv("RS_K2_CONV1_OLD") != nullptr;   // A/B and test hook (the form other channel counts run)
    if (c1 == 8 && c2 == 32 && pl.F2 <= 48 && !conv1_first_form) {
        hipLaunchKernelGGL(k2_conv1_mfma_kernel, dim3((pl.T2 + 4 * CONV1_TT - 1) / (4 * CONV1_TT), B), dim3(256), 0, s, a0, pl.T1, pl.F, pl.T2, pl.F2, k.conv1_w,
                           k.conv1_b, a1);
    } else {
        const size_t lds = (size_t)(3 * pl.F * c1 + 9 * c1 * c2) * 4;
        if (lds > 64 * 1024) RS_TRY(rs_ensure_dynamic_lds(ctx, (const void*)k2_conv1_kernel, (int)lds));
        hipLaunchKernelGGL(k2_conv1_kernel, dim3(pl.T2, B), dim3(256), lds, s, a0, pl.T1, pl.F, c1, pl.T2, pl.F2, c2, k.conv1_w, k.conv1_b, a1);
    }
    const long long rows3 = (long long)B * T3 * F3;
    static const bool conv2_fused_env = [] { const char* e = getenv("RS_K2_CONV2_FUSED"); return e ? atoi(e) != 0 : true; }();
    const bool conv2_fused = (ctx->k2_conv2_fused < 0 ? conv2_fused_env : ctx->k2_conv2_fused != 0) && c2 == 32 && c3 == 128;
    if (ctx->n_cus <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
        ctx->n_cus = n;
    }
    if (conv2_fused) {
        // patches gathered into LDS, weights in registers: no patch matrix in HBM
        if (int rc2 = rs_ensure_dynamic_lds(ctx, (const void*)k2_conv2_fused_kernel, C2_LDS); rc2 != RS_OK) { rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s); return rc2; }
        const long long n_tiles = (rows3 + 127) / 128;
        hipLaunchKernelGGL(k2_conv2_fused_kernel, dim3((unsigned)(n_tiles < ctx->n_cus ? n_tiles : ctx->n_cus)), dim3(512), C2_LDS, s, a1, pl.T2, pl.F2, T3, F3, k.conv2_w,
                           pl.Kp, k.conv2_b, a2, rows3);
        rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
        RS_CHECK_LAUNCH(ctx, "zipformer encoder_embed convs");
    } else {
    hipLaunchKernelGGL(k2_im2col_kernel, dim3((unsigned)((rows3 + 3) / 4)), dim3(256), 0, s, a1, pl.T2, pl.F2, c2, T3, F3, pl.Kp, rows3, col);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "zipformer encoder_embed convs");
    RS_TRY(gemm(col, pl.Kp, k.conv2_w, pl.Kp, a2, c3, rows3, c3, RS_GEMM_BIAS | RS_GEMM_SWOOSHR | RS_GEMM_OUT_F32, k.conv2_b, nullptr));
    }
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, 0.0, 0.0);
    hipLaunchKernelGGL(k2_cnx_dw_kernel<10>, dim3((T3 + CNX_TT - 1) / CNX_TT, B), dim3(256), 0, s, a2, lens3, T3, F3, c3, k.cnx_dw_w, k.cnx_dw_b, dwo);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    static const bool cnx_fused_env = [] { const char* e = getenv("RS_K2_CNX_FUSED"); return e ? atoi(e) != 0 : true; }();
    const bool cnx_fused = ctx->k2_cnx_fused < 0 ? cnx_fused_env : ctx->k2_cnx_fused != 0;
    if (cnx_fused && c3 == CX_C) {
        // both pointwise convolutions in one launch, the hidden tensor stays on the CU; result (bf16) in place over dwo
        RS_TRY(rs_ensure_dynamic_lds(ctx, (const void*)k2_cnx_pw_fused_kernel, CX_LDS));
        if (ctx->n_cus <= 0) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
            ctx->n_cus = n;
        }
        const long long n_tiles = (rows3 + CX_ROWS - 1) / CX_ROWS;
        hipLaunchKernelGGL(k2_cnx_pw_fused_kernel, dim3((unsigned)(n_tiles < ctx->n_cus ? n_tiles : ctx->n_cus)), dim3(512), CX_LDS, s, dwo, k.cnx_pw1_w, k.cnx_pw1_b,
                           k.cnx_pw2_w, k.cnx_pw2_b, a2, dwo, rows3);
        RS_CHECK_LAUNCH(ctx, "zipformer ConvNeXt pointwise pair");
    } else {
    RS_TRY(gemm(dwo, c3, k.cnx_pw1_w, c3, hbuf, 3 * c3, rows3, 3 * c3, RS_GEMM_BIAS | RS_GEMM_SWOOSHL, k.cnx_pw1_b, nullptr));
    // (its bf16 copy lands in dwo, [B*T3][F3 * c3] in (f, c) order: the operand of `out`)
    RS_TRY(gemm(hbuf, 3 * c3, k.cnx_pw2_w, 3 * c3, a2, c3, rows3, c3, RES, k.cnx_pw2_b, a2, dwo));
    }
    const long long M3 = (long long)B * T3;
    const int d0 = d.encoder_dim[0];
    float* emb = x0;                       // encoder_embed's output; stack 0 reads it as `prev`
    RS_TRY(gemm(dwo, F3 * c3, k.emb_out_w, F3 * c3, emb, d0, M3, d0, RS_GEMM_BIAS | RS_GEMM_OUT_F32, k.emb_out_b, nullptr));
    hipLaunchKernelGGL(k2_biasnorm_kernel, dim3((unsigned)((M3 + 3) / 4)), dim3(256), 0, s, emb, k.emb_norm_bias, k.emb_norm_scale, (const float*)nullptr,
                       (const float*)nullptr, (int)M3, d0, emb, (uint16_t*)nullptr);
    if (k.tap_embed) RS_HIP(ctx, hipMemcpyAsync(k.tap_embed, emb, (size_t)M3 * d0 * 4, hipMemcpyDeviceToDevice, s));
    RS_CHECK_LAUNCH(ctx, "zipformer encoder_embed");
    // ---- the stacks --------------------------------------------------------------------------------------------------------
    const float* prev = emb;
    int d_prev = d0;
    size_t tap_off = 0;
    for (int st = 0; st < d.n_stacks; ++st) {
        const int dd = d.encoder_dim[st], H = d.num_heads[st], ds = d.downsampling[st], Ts = pl.Ts[st], Tp = pl.Tp[st], kk = d.cnn_kernel[st];
        const int hid = 3 * dd / 4, hidp = pad64(hid), vw = H * K2_VD, vwp = pad64(vw), nin = (2 * K2_QD + K2_PD) * H;
        const int32_t* lens = lens_all + (size_t)(1 + st) * B;
        const long long M = (long long)B * Ts;
        float* stack_out = reinterpret_cast<float*>(ws + pl.off_stackout[st]);
        // Two working buffers per stack: `cur` holds a layer's input and stays intact for the whole layer (it is the x0 of both
        // bypass modules), the layer's first residual GEMM writes into `nxt` (out != residual) and everything after it works in
        // place there; the buffers swap roles per layer — no copy of the layer input (it was a 1.1 ms memcpy per batch).  The
        // last layer of a full-rate stack writes its result straight into the stack's output buffer.
        float* cur = x;
        float* nxt = x0;                       // (stack 0: `prev` IS x0 — encoder_embed's output — and is dead once stack_in has read it)
        hipLaunchKernelGGL(k2_stack_in_kernel, dim3(Ts, B), dim3(256), 0, s, prev, d_prev, lens3, T3, dd, ds, Ts, k.ds_w[st], ds == 1 ? (float*)nullptr : src, cur);
        for (int j = 0; j < d.num_layers[st]; ++j) {
            const rs_k2_layer& L = k.stacks[st][j];
            const size_t n = (size_t)M * dd;
            const bool last = j == d.num_layers[st] - 1;
            float* xs = nxt;
            if (j == 0) cast(cur, xb, n);           // later layers: the previous layer's BiasNorm wrote the bf16 copy
            // attention weights, shared by the three attention modules of the layer
            RS_TRY(gemm(xb, dd, L.attw_in_w, dd, qkp, nin, M, nin, RS_GEMM_BIAS, L.attw_in_b, nullptr));
            {
                const size_t lds = (size_t)(Ts + 64) * 16;
                if (lds > 160 * 1024 - 1024) return rs_fail(ctx, RS_EINVAL, "zipformer: %d frames in stack %d exceed the attention kernel's position table in LDS", Ts, st);
                if (lds > 64 * 1024) RS_TRY(rs_ensure_dynamic_lds(ctx, (const void*)k2_attn_weights_kernel, (int)lds));
                if (Ts > k.pos_cap) return rs_fail(ctx, RS_EINVAL, "zipformer: %d frames exceed the registered position tables (%d)", Ts, k.pos_cap);
                rs_prof_begin(ctx, RS_PROF_ATTN, s, (double)B * H * Ts * (double)Ts * (2.0 * K2_QD + 2.0 * K2_PD + 8.0) * 2.0, (double)B * H * Ts * (double)Tp * 2.0);
                static const bool three_sweeps = getenv("RS_K2_ATTW_SWEEPS") != nullptr && atoi(getenv("RS_K2_ATTW_SWEEPS")) == 3;   // A/B and test hook
                const dim3 grid((Ts + 63) / 64, H, B);
#define RS_K2_ATTW1(NT)                                                                                                                     \
                do {                                                                                                                        \
                    if (lds > 64 * 1024) RS_TRY(rs_ensure_dynamic_lds(ctx, (const void*)k2_attn_weights1_kernel<NT>, (int)lds));            \
                    hipLaunchKernelGGL((k2_attn_weights1_kernel<NT>), grid, dim3(256), lds, s, qkp, nin, L.pos_proj, k.pos_cap, H, lens, Ts, Tp, W); \
                } while (0)
                if (three_sweeps || Tp > 640) hipLaunchKernelGGL(k2_attn_weights_kernel, grid, dim3(256), lds, s, qkp, nin, L.pos_proj, k.pos_cap, H, lens, Ts, Tp, W);
                else if (Tp <= 160) RS_K2_ATTW1(10);
                else if (Tp <= 320) RS_K2_ATTW1(20);
                else RS_K2_ATTW1(40);
#undef RS_K2_ATTW1
                rs_prof_end(ctx, RS_PROF_ATTN, s);
            }
            auto ffn = [&](int f, int width, const float* res, bool emit) -> int {
                if (int r = gemm(xb, dd, L.ff_in_w[f], dd, big, width, M, width, RS_GEMM_BIAS | RS_GEMM_SWOOSHL, L.ff_in_b[f], nullptr); r != RS_OK) return r;
                return gemm(big, width, L.ff_out_w[f], width, xs, dd, M, dd, RES, L.ff_out_b[f], res, emit ? xb : nullptr);
            };
            auto self_attn = [&](int a) -> int {       // (the out projection always feeds another branch: it writes the bf16 copy)
                if (int r = gemm(xb, dd, L.sa_in_w[a], dd, big, vw, M, vw, RS_GEMM_BIAS, L.sa_in_b[a], nullptr); r != RS_OK) return r;
                // `av` is shared by the three kinds of branch (row pitches vwp / hidp / dd): the columns that pad the out projection's
                // K extent to a multiple of 64 are zeroed before every use
                if (vwp != vw && hipMemsetAsync(av, 0, (size_t)M * vwp * 2, s) != hipSuccess) return rs_fail(ctx, RS_EHIP, "memset failed");
                rs_prof_begin(ctx, RS_PROF_ATTN, s, (double)B * H * Ts * (double)Ts * 2.0 * 16.0, (double)B * H * Ts * (double)Tp * 2.0);
                const int C = pad64(H * 16);
                hipLaunchKernelGGL((k2_vt_kernel<0>), dim3((Tp + 63) / 64, C / 64, B), dim3(256), 0, s, big, vw, vw, lens, Ts, Tp, C, vT);
                hipLaunchKernelGGL((k2_pv_kernel<1, 0>), dim3((Ts + 127) / 128, 1, B * H), dim3(256), 0, s, W, H, Tp, vT, C, big, vw, 0, lens, Ts, av, vwp);
                rs_prof_end(ctx, RS_PROF_ATTN, s);
                return gemm(av, vwp, L.sa_out_w[a], vwp, xs, dd, M, dd, RES, L.sa_out_b[a], xs, xb);
            };
            auto conv_module = [&](int a) -> int {
                if (int r = gemm(xb, dd, L.cm_in_w[a], dd, big, dd, M, 2 * dd, RS_GEMM_BIAS | RS_GEMM_GLU, L.cm_in_b[a], nullptr); r != RS_OK) return r;
                if (int r = rs_launch_dwconv_act(ctx, big, L.cm_dw_w[a], L.cm_dw_b[a], lens, B, Ts, dd, kk, 1, av, s); r != RS_OK) return r;
                return gemm(av, dd, L.cm_out_w[a], dd, xs, dd, M, dd, RES, L.cm_out_b[a], xs, xb);
            };
            RS_TRY(ffn(0, d.ff_dim[st] * 3 / 4, cur, true));    // xs = cur + ff1(cur): the layer leaves `cur` and moves into `nxt`
            // non-linear attention: head 0's weights over tanh-gated values, output gate, out projection
            RS_TRY(gemm(xb, dd, L.na_in_w, dd, big, 3 * hid, M, 3 * hid, RS_GEMM_BIAS, L.na_in_b, nullptr));
            if (hidp != hid) RS_HIP(ctx, hipMemsetAsync(av, 0, (size_t)M * hidp * 2, s));
            rs_prof_begin(ctx, RS_PROF_ATTN, s, (double)B * Ts * (double)Ts * 2.0 * hid, (double)B * Ts * (double)Tp * 2.0);
            hipLaunchKernelGGL((k2_vt_kernel<1>), dim3((Tp + 63) / 64, hidp / 64, B), dim3(256), 0, s, big, 3 * hid, hid, lens, Ts, Tp, hidp, vT);
            hipLaunchKernelGGL((k2_pv_kernel<4, 1>), dim3((Ts + 127) / 128, (hid + 63) / 64, B), dim3(256), 0, s, W, H, Tp, vT, hidp, big, 3 * hid, hid, lens, Ts, av, hidp);
            rs_prof_end(ctx, RS_PROF_ATTN, s);
            RS_TRY(gemm(av, hidp, L.na_out_w, hidp, xs, dd, M, dd, RES, L.na_out_b, xs, xb));
            RS_TRY(self_attn(0));
            RS_TRY(conv_module(0));
            RS_TRY(ffn(1, d.ff_dim[st], xs, false));            // (bypass_mid rewrites x and its bf16 copy)
            hipLaunchKernelGGL(k2_bypass_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, xs, cur, L.bypass_mid, dd, n / 4, xb);
            RS_TRY(self_attn(1));
            RS_TRY(conv_module(1));
            RS_TRY(ffn(2, d.ff_dim[st] * 5 / 4, xs, false));
            float* dst = (last && ds == 1) ? stack_out : xs;
            hipLaunchKernelGGL(k2_biasnorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, xs, L.norm_bias, L.norm_scale, cur, L.bypass, (int)M, dd, dst,
                               last ? (uint16_t*)nullptr : xb);
            RS_CHECK_LAUNCH(ctx, "zipformer layer");
            nxt = cur;
            cur = xs;
        }
        float* xs = cur;                       // the stack's result at its own rate (down-sampled stacks)
        if (ds > 1) hipLaunchKernelGGL(k2_stack_out_kernel, dim3(T3, B), dim3(256), 0, s, src, xs, T3, Ts, dd, ds, k.comb_scale[st], stack_out);
        if (k.tap_stacks) {
            RS_HIP(ctx, hipMemcpyAsync(k.tap_stacks + tap_off, stack_out, (size_t)M3 * dd * 4, hipMemcpyDeviceToDevice, s));
            tap_off += (size_t)M3 * dd;
        }
        prev = stack_out;
        d_prev = dd;
    }
    // ---- output: widest channels of every stack, down-sampling by 2, joiner.encoder_proj -----------------------------------------
    K2Pieces pc{};
    {
        int n = 0, cur_dim = d.encoder_dim[d.n_stacks - 1];
        pc.p[n] = reinterpret_cast<const float*>(ws + pl.off_stackout[d.n_stacks - 1]); pc.c0[n] = 0; pc.n[n] = cur_dim; pc.ld[n] = cur_dim; ++n;
        for (int st = d.n_stacks - 2; st >= 0; --st) {
            const int dd = d.encoder_dim[st];
            if (dd > cur_dim) {
                pc.p[n] = reinterpret_cast<const float*>(ws + pl.off_stackout[st]); pc.c0[n] = cur_dim; pc.n[n] = dd - cur_dim; pc.ld[n] = dd; ++n;
                cur_dim = dd;
            }
        }
        pc.count = n;
    }
    const int To = pl.To;
    hipLaunchKernelGGL(k2_output_kernel, dim3(To, B), dim3(256), 0, s, pc, lens3, T3, To, k.out_dim, k.out_ds_w, enc_out, encb, enc_lens);
    RS_CHECK_LAUNCH(ctx, "zipformer output");
    RS_TRY(gemm(encb, k.out_dim, ctx->jenc_w, k.out_dim, joint_enc, d.joiner_dim, (long long)B * To, d.joiner_dim, RS_GEMM_BIAS | RS_GEMM_OUT_F32, ctx->jenc_b,
                nullptr));
#undef RS_TRY
    return RS_OK;
}

// The float32 encoder: the call sequence of rs_k2_encoder_forward_impl with float32 operands everywhere (oracle/zipformer.py with
// recipe "fp32" is its line-by-line counterpart).
static int rs_k2_encoder_forward_f32(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max, float* enc_out, float* joint_enc,
                                     int32_t* enc_lens, void* workspace, size_t workspace_bytes, hipStream_t s) {
    rs_k2& k = *ctx->k2;
    const rs_k2_dims& d = k.d;
    const K2PlanF32 pl = k2_plan_f32(ctx, B, t_max);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "zipformer (float32 mode): workspace %zu < %zu", workspace_bytes, pl.total);
    char* ws = reinterpret_cast<char*>(workspace);
    auto fp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    int32_t* lens_all = reinterpret_cast<int32_t*>(ws + pl.off_lens);
    const int32_t* lens3 = lens_all;
    float *a0 = fp(pl.off_a0), *a1 = fp(pl.off_a1), *col = fp(pl.off_col), *a2 = fp(pl.off_a2), *dwo = fp(pl.off_dwo), *hbuf = fp(pl.off_h);
    float *x = fp(pl.off_x), *x0 = fp(pl.off_x0), *src = fp(pl.off_src), *qkp = fp(pl.off_qkp), *W = fp(pl.off_w), *big = fp(pl.off_big);
    float *av = fp(pl.off_av), *gate = fp(pl.off_gate), *encf = enc_out ? enc_out : fp(pl.off_enc);
    const int c1 = d.embed_c1, c2 = d.embed_c2, c3 = d.embed_c3, T3 = pl.T3, F3 = pl.F3;
    int rc;
#define RS_TRY(call) do { rc = (call); if (rc != RS_OK) return rc; } while (0)
    auto gemm = [&](const float* A, int lda, const float* Wt, int K, float* out, int ldc, long long M, int N, int flags, const float* bias,
                    const float* res) -> int {
        return rs_launch_gemm_f32(ctx, A, lda, Wt, K, out, ldc, (int)M, N, K, flags, bias, 1.0f, res, nullptr, 0, 0, s);
    };
    const int RES = RS_GEMM_BIAS | RS_GEMM_RESIDUAL;
    hipLaunchKernelGGL(k2_lens_kernel, dim3((B + 255) / 256), dim3(256), 0, s, n_frames, B, d.n_stacks, d.downsampling[0], d.downsampling[1],
                       d.downsampling[2], d.downsampling[3], d.downsampling[4], d.downsampling[5], d.downsampling[6], d.downsampling[7], lens_all);
    // ---- encoder_embed
    const long long rows3 = (long long)B * T3 * F3, M3 = (long long)B * T3;
    if (rows3 > 0x7fffffffLL / 4) return rs_fail(ctx, RS_EINVAL, "zipformer (float32 mode): %lld patch rows exceed the float32 GEMM's row count", rows3);
    hipLaunchKernelGGL(k2f_conv0_kernel, dim3(pl.T1, B), dim3(256), 0, s, feats, t_max, pl.F, c1, k.conv0_w, k.conv0_b, a0);
    hipLaunchKernelGGL(k2f_conv1_kernel, dim3(pl.T2, B), dim3(256), 0, s, a0, pl.T1, pl.F, c1, pl.T2, pl.F2, c2, k.conv1_w, k.conv1_b, a1);
    const int Kc = pad32(9 * c2);
    hipLaunchKernelGGL(k2f_im2col_kernel, dim3((unsigned)((rows3 + 3) / 4)), dim3(256), 0, s, a1, pl.T2, pl.F2, c2, T3, F3, Kc, rows3, col);
    RS_CHECK_LAUNCH(ctx, "zipformer (float32 mode) encoder_embed convs");
    RS_TRY(gemm(col, Kc, k.conv2_w32, Kc, a2, c3, rows3, c3, RS_GEMM_BIAS | RS_GEMM_SWOOSHR, k.conv2_b, nullptr));
    hipLaunchKernelGGL(k2f_cnx_dw_kernel, dim3(T3, B), dim3(256), 0, s, a2, lens3, T3, F3, c3, k.cnx_dw_w, k.cnx_dw_b, dwo);
    RS_TRY(gemm(dwo, c3, k.cnx_pw1_w32, c3, hbuf, 3 * c3, rows3, 3 * c3, RS_GEMM_BIAS | RS_GEMM_SWOOSHL, k.cnx_pw1_b, nullptr));
    RS_TRY(gemm(hbuf, 3 * c3, k.cnx_pw2_w32, 3 * c3, a2, c3, rows3, c3, RES, k.cnx_pw2_b, a2));
    const int d0 = d.encoder_dim[0];
    float* emb = x0;
    RS_TRY(gemm(a2, F3 * c3, k.emb_out_w32, F3 * c3, emb, d0, M3, d0, RS_GEMM_BIAS, k.emb_out_b, nullptr));
    hipLaunchKernelGGL(k2_biasnorm_kernel, dim3((unsigned)((M3 + 3) / 4)), dim3(256), 0, s, emb, k.emb_norm_bias, k.emb_norm_scale, (const float*)nullptr,
                       (const float*)nullptr, (int)M3, d0, emb, (uint16_t*)nullptr);
    if (k.tap_embed) RS_HIP(ctx, hipMemcpyAsync(k.tap_embed, emb, (size_t)M3 * d0 * 4, hipMemcpyDeviceToDevice, s));
    RS_CHECK_LAUNCH(ctx, "zipformer (float32 mode) encoder_embed");
    // ---- the stacks
    const float* prev = emb;
    int d_prev = d0;
    size_t tap_off = 0;
    for (int st = 0; st < d.n_stacks; ++st) {
        const int dd = d.encoder_dim[st], H = d.num_heads[st], ds = d.downsampling[st], Ts = pl.Ts[st], Tp = pl.Tp[st], kk = d.cnn_kernel[st];
        const int hid = 3 * dd / 4, hidp = pad32(hid), vw = H * K2_VD, vwp = pad32(vw), nin = (2 * K2_QD + K2_PD) * H;
        const int32_t* lens = lens_all + (size_t)(1 + st) * B;
        const long long M = (long long)B * Ts;
        float* stack_out = fp(pl.off_stackout[st]);
        float* cur = x;
        float* nxt = x0;
        if (Ts > k.pos_cap) return rs_fail(ctx, RS_EINVAL, "zipformer: %d frames exceed the registered position tables (%d)", Ts, k.pos_cap);
        hipLaunchKernelGGL(k2_stack_in_kernel, dim3(Ts, B), dim3(256), 0, s, prev, d_prev, lens3, T3, dd, ds, Ts, k.ds_w[st], ds == 1 ? (float*)nullptr : src, cur);
        for (int j = 0; j < d.num_layers[st]; ++j) {
            const rs_k2_layer& L = k.stacks[st][j];
            const rs_k2_layer32& L32 = k.stacks32[st][j];
            const size_t n = (size_t)M * dd;
            const bool last = j == d.num_layers[st] - 1;
            float* xs = nxt;
            // attention weights from the layer's input
            RS_TRY(gemm(cur, dd, L32.attw_in_w, dd, qkp, nin, M, nin, RS_GEMM_BIAS, L.attw_in_b, nullptr));
            hipLaunchKernelGGL(k2f_attn_weights_kernel, dim3((Ts + 3) / 4, H, B), dim3(256), 0, s, qkp, nin, L.pos_proj, k.pos_cap, H, lens, Ts, Tp, W);
            auto ffn = [&](int f, int width, const float* in, const float* res) -> int {
                if (int r = gemm(in, dd, L32.ff_in_w[f], dd, big, width, M, width, RS_GEMM_BIAS | RS_GEMM_SWOOSHL, L.ff_in_b[f], nullptr); r != RS_OK) return r;
                return gemm(big, width, L32.ff_out_w[f], width, xs, dd, M, dd, RES, L.ff_out_b[f], res);
            };
            auto self_attn = [&](int a) -> int {
                if (int r = gemm(xs, dd, L32.sa_in_w[a], dd, big, vw, M, vw, RS_GEMM_BIAS, L.sa_in_b[a], nullptr); r != RS_OK) return r;
                if (vwp != vw && hipMemsetAsync(av, 0, (size_t)M * vwp * 4, s) != hipSuccess) return rs_fail(ctx, RS_EHIP, "memset failed");
                hipLaunchKernelGGL(k2f_pv_kernel, dim3(Ts, B), dim3((vw + 63) / 64 * 64), 0, s, W, H, Tp, big, vw, lens, Ts, av, vwp);
                return gemm(av, vwp, L32.sa_out_w[a], vwp, xs, dd, M, dd, RES, L.sa_out_b[a], xs);
            };
            auto conv_module = [&](int a) -> int {
                if (int r = gemm(xs, dd, L32.cm_in_w[a], dd, big, 2 * dd, M, 2 * dd, RS_GEMM_BIAS, L32.cm_in_b[a], nullptr); r != RS_OK) return r;
                hipLaunchKernelGGL(k2f_glu_dwconv_swoosh_kernel, dim3((dd + 255) / 256, Ts, B), dim3(256), 0, s, big, L.cm_dw_w[a], L.cm_dw_b[a], lens, Ts, dd, kk, av);
                return gemm(av, dd, L32.cm_out_w[a], dd, xs, dd, M, dd, RES, L.cm_out_b[a], xs);
            };
            RS_TRY(ffn(0, d.ff_dim[st] * 3 / 4, cur, cur));         // xs = cur + ff1(cur)
            // non-linear attention
            RS_TRY(gemm(xs, dd, L32.na_in_w, dd, big, 3 * hid, M, 3 * hid, RS_GEMM_BIAS, L.na_in_b, nullptr));
            if (hidp != hid) RS_HIP(ctx, hipMemsetAsync(av, 0, (size_t)M * hidp * 4, s));
            hipLaunchKernelGGL(k2f_na_gate_kernel, dim3((unsigned)(((size_t)M * hid + 255) / 256)), dim3(256), 0, s, big, hid, (size_t)M, gate);
            hipLaunchKernelGGL(k2f_na_pv_kernel, dim3((hid + 255) / 256, (Ts + 3) / 4, B), dim3(256), 0, s, W, H, Tp, gate, big, hid, lens, Ts, av, hidp);
            RS_TRY(gemm(av, hidp, L32.na_out_w, hidp, xs, dd, M, dd, RES, L.na_out_b, xs));
            RS_TRY(self_attn(0));
            RS_TRY(conv_module(0));
            RS_TRY(ffn(1, d.ff_dim[st], xs, xs));
            hipLaunchKernelGGL(k2_bypass_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, xs, cur, L.bypass_mid, dd, n / 4, (uint16_t*)nullptr);
            RS_TRY(self_attn(1));
            RS_TRY(conv_module(1));
            RS_TRY(ffn(2, d.ff_dim[st] * 5 / 4, xs, xs));
            float* dst = (last && ds == 1) ? stack_out : xs;
            hipLaunchKernelGGL(k2_biasnorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, xs, L.norm_bias, L.norm_scale, cur, L.bypass, (int)M, dd, dst,
                               (uint16_t*)nullptr);
            RS_CHECK_LAUNCH(ctx, "zipformer (float32 mode) layer");
            nxt = cur;
            cur = xs;
        }
        if (ds > 1) hipLaunchKernelGGL(k2_stack_out_kernel, dim3(T3, B), dim3(256), 0, s, src, cur, T3, Ts, dd, ds, k.comb_scale[st], stack_out);
        if (k.tap_stacks) {
            RS_HIP(ctx, hipMemcpyAsync(k.tap_stacks + tap_off, stack_out, (size_t)M3 * dd * 4, hipMemcpyDeviceToDevice, s));
            tap_off += (size_t)M3 * dd;
        }
        prev = stack_out;
        d_prev = dd;
    }
    K2Pieces pc{};
    {
        int n = 0, cur_dim = d.encoder_dim[d.n_stacks - 1];
        pc.p[n] = fp(pl.off_stackout[d.n_stacks - 1]); pc.c0[n] = 0; pc.n[n] = cur_dim; pc.ld[n] = cur_dim; ++n;
        for (int st = d.n_stacks - 2; st >= 0; --st) {
            const int dd = d.encoder_dim[st];
            if (dd > cur_dim) {
                pc.p[n] = fp(pl.off_stackout[st]); pc.c0[n] = cur_dim; pc.n[n] = dd - cur_dim; pc.ld[n] = dd; ++n;
                cur_dim = dd;
            }
        }
        pc.count = n;
    }
    hipLaunchKernelGGL(k2_output_kernel, dim3(pl.To, B), dim3(256), 0, s, pc, lens3, T3, pl.To, k.out_dim, k.out_ds_w, encf, (uint16_t*)nullptr, enc_lens);
    RS_CHECK_LAUNCH(ctx, "zipformer (float32 mode) output");
    RS_TRY(gemm(encf, k.out_dim, k.jenc_w32, k.out_dim, joint_enc, d.joiner_dim, (long long)B * pl.To, d.joiner_dim, RS_GEMM_BIAS, ctx->jenc_b, nullptr));
#undef RS_TRY
    return RS_OK;
}

// k_layernorm.hip — LayerNorm over the model dimension and the fused
// conv-module middle (row L6: GLU -> frame mask -> depthwise conv + folded BatchNorm -> SiLU).
// Both are HBM-bound: one pass over the data, 16-byte accesses, wave-shuffle reductions.
#include "rs_common.h"

namespace {

// the normalisation of one element, written out so that the GEMM epilogue that applies a deferred output norm
// (k_gemm_bf16.hip, OUT_RESLN) produces the same bits from the same (mean, rstd)
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float g, float b) { return fmaf((v - mean) * rstd, g, b); }

// One wave per row; d = 256 * NV (NV float4 per lane).  Two-pass statistics in registers:
// mean, then sum((x-mean)^2) — the same formulation torch.layer_norm uses in float32.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, int M, float eps,
                                                        uint16_t* __restrict__ out_bf16, float* __restrict__ out_f32) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[i * 64 + lane];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, dd = v[i].w - mean;
        q += (a * a + bb * bb) + (c * c + dd * dd);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
    const float4* gr = reinterpret_cast<const float4*>(g);
    const float4* br = reinterpret_cast<const float4*>(b);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 gg = gr[i * 64 + lane], bb = br[i * 64 + lane];
        float4 y;
        y.x = ln_apply(v[i].x, mean, rstd, gg.x, bb.x);
        y.y = ln_apply(v[i].y, mean, rstd, gg.y, bb.y);
        y.z = ln_apply(v[i].z, mean, rstd, gg.z, bb.z);
        y.w = ln_apply(v[i].w, mean, rstd, gg.w, bb.w);
        if (out_f32) reinterpret_cast<float4*>(out_f32 + (size_t)row * D)[i * 64 + lane] = y;
        if (out_bf16) {
            u16x4_t o;
            o[0] = f32_to_bf16(y.x); o[1] = f32_to_bf16(y.y); o[2] = f32_to_bf16(y.z); o[3] = f32_to_bf16(y.w);
            reinterpret_cast<u16x4_t*>(out_bf16 + (size_t)row * D)[i * 64 + lane] = o;
        }
    }
}

// Two LayerNorms back to back on one read of the row: y = LN(x; g1, b1) is stored as f32 (the residual
// stream after a layer's output norm), z = LN(y; g2, b2) as bf16 (the next layer's first FFN input).
// Same arithmetic as two layernorm_kernel launches (y is normalised from registers instead of being
// re-read), 362 MB instead of 507 MB of traffic per layer boundary at B=256.
template <int NV>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, const float* __restrict__ g1,
                                                         const float* __restrict__ b1, const float* __restrict__ g2,
                                                         const float* __restrict__ b2, int M, float eps,
                                                         float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16,
                                                         float* __restrict__ stats) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[i * 64 + lane];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    float mean = wave_sum(s) * (1.0f / D);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, dd = v[i].w - mean;
        q += (a * a + bb * bb) + (c * c + dd * dd);
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
    // deferred output norm: y is not stored; the consumer of the residual stream (the next layer's first residual GEMM)
    // normalises x on the fly from these two numbers
    if (stats && lane == 0) *reinterpret_cast<float2*>(stats + (size_t)row * 2) = make_float2(mean, rstd);
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 gg = reinterpret_cast<const float4*>(g1)[i * 64 + lane], bb = reinterpret_cast<const float4*>(b1)[i * 64 + lane];
        float4 y;
        y.x = ln_apply(v[i].x, mean, rstd, gg.x, bb.x);
        y.y = ln_apply(v[i].y, mean, rstd, gg.y, bb.y);
        y.z = ln_apply(v[i].z, mean, rstd, gg.z, bb.z);
        y.w = ln_apply(v[i].w, mean, rstd, gg.w, bb.w);
        if (out_f32) reinterpret_cast<float4*>(out_f32 + (size_t)row * D)[i * 64 + lane] = y;
        v[i] = y;
        s += (y.x + y.y) + (y.z + y.w);
    }
    mean = wave_sum(s) * (1.0f / D);
    q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, dd = v[i].w - mean;
        q += (a * a + bb * bb) + (c * c + dd * dd);
    }
    rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 gg = reinterpret_cast<const float4*>(g2)[i * 64 + lane], bb = reinterpret_cast<const float4*>(b2)[i * 64 + lane];
        u16x4_t o;
        o[0] = f32_to_bf16(ln_apply(v[i].x, mean, rstd, gg.x, bb.x));
        o[1] = f32_to_bf16(ln_apply(v[i].y, mean, rstd, gg.y, bb.y));
        o[2] = f32_to_bf16(ln_apply(v[i].z, mean, rstd, gg.z, bb.z));
        o[3] = f32_to_bf16(ln_apply(v[i].w, mean, rstd, gg.w, bb.w));
        reinterpret_cast<u16x4_t*>(out_bf16 + (size_t)row * D)[i * 64 + lane] = o;
    }
}

// GLU + mask + depthwise conv (BatchNorm folded) + SiLU.
//   x  bf16 [B*T][2d]  (a | gate),  w f32 [k][d] tap-major, bias f32 [d]  ->  out bf16 [B*T][d]
// Input layouts (rs_asr.h RS_GLU_*): 0 = halves (values in columns [0, d), gates in [d, 2d)); 1 = blocks of 32
// (columns 64j .. 64j+31 are the values of channels 32j .. 32j+31, the next 32 their gates: what the pw1 GEMM
// produces from the loader's interleaved weight rows); 2 = GLU already applied by the GEMM epilogue, x is [B*T][d].
// Workgroup = 256 threads = 32 channel groups (8 channels, one 16-B load) x 8 time lanes; it
// produces a tile of TT frames x 256 channels.  The GLU'd, masked input tile (TT + k - 1 frames)
// is staged once in LDS as f32, so every pw1 output element is read from HBM exactly once.
constexpr int CT = 256;  // channels per workgroup
constexpr int TT = 32;   // output frames per workgroup
constexpr int KMAX = 31;

__global__ __launch_bounds__(256) void glu_dwconv_silu_kernel(const uint16_t* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const int32_t* __restrict__ lens, int T, int d, int k,
                                                              int layout, uint16_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);  // [(TT + k - 1)][CT]
    const int b = blockIdx.z, c0 = blockIdx.y * CT, t0 = blockIdx.x * TT;
    const int cg = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int len = lens[b];
    const int half = (k - 1) >> 1;
    const int rows = TT + k - 1;
    const int c = c0 + cg * 8;
    for (int r = tl; r < rows; r += 8) {
        const int t = t0 + r - half;
        float u[8];
        if (t >= 0 && t < T && t < len) {
            const int ld = layout == 2 ? d : 2 * d;
            const int ca = layout == 1 ? 64 * (c >> 5) + (c & 31) : c, goff = layout == 1 ? 32 : d;
            const uint16_t* px = x + ((size_t)b * T + t) * ld + ca;
            const u16x8_t a = *reinterpret_cast<const u16x8_t*>(px);
            if (layout == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = bf16_to_f32(a[e]);
            } else {
                const u16x8_t gt = *reinterpret_cast<const u16x8_t*>(px + goff);
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = bf16_to_f32(a[e]) * sigmoid_f(bf16_to_f32(gt[e]));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = 0.0f;
        }
        float4* dst = reinterpret_cast<float4*>(tile + r * CT + cg * 8);
        dst[0] = make_float4(u[0], u[1], u[2], u[3]);
        dst[1] = make_float4(u[4], u[5], u[6], u[7]);
    }
    __syncthreads();
    float bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bb[e] = bias[c + e];
    for (int tt = tl; tt < TT; tt += 8) {
        const int t = t0 + tt;
        if (t >= T) break;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bb[e];
        for (int j = 0; j < k; ++j) {
            const float4* src = reinterpret_cast<const float4*>(tile + (tt + j) * CT + cg * 8);
            const float4 x0 = src[0], x1 = src[1];
            const float4* wj = reinterpret_cast<const float4*>(w + (size_t)j * d + c);
            const float4 w0 = wj[0], w1 = wj[1];
            acc[0] = fmaf(x0.x, w0.x, acc[0]); acc[1] = fmaf(x0.y, w0.y, acc[1]);
            acc[2] = fmaf(x0.z, w0.z, acc[2]); acc[3] = fmaf(x0.w, w0.w, acc[3]);
            acc[4] = fmaf(x1.x, w1.x, acc[4]); acc[5] = fmaf(x1.y, w1.y, acc[5]);
            acc[6] = fmaf(x1.z, w1.z, acc[6]); acc[7] = fmaf(x1.w, w1.w, acc[7]);
        }
        u16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(silu_f(acc[e]));
        *reinterpret_cast<u16x8_t*>(out + ((size_t)b * T + t) * d + c) = o;
    }
}

// Fast path of the same operator for a compile-time kernel size K: a thread owns 8 channels x R
// CONSECUTIVE frames.  The K taps of its channels live in registers (loaded once), the R + K - 1
// input frames it needs are read from LDS once each (a sliding window: 2 x 16 B per frame instead
// of 2 x 16 B per frame AND tap), and every HBM load of the tile is in flight before the first GLU.
// Workgroup tile = 8 * R frames x 256 channels (R = 6: 48 frames, three tiles cover T' = 138 + 6).
template <int K, int R, int LAYOUT>
__global__ __launch_bounds__(256) void glu_dwconv_silu_fast_kernel(const uint16_t* __restrict__ x,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   const int32_t* __restrict__ lens, int T, int d,
                                                                   uint16_t* __restrict__ out) {
    constexpr int TTF = 8 * R, ROWS = TTF + K - 1, HALF = (K - 1) / 2;
    constexpr int NLD = (ROWS + 7) / 8;          // rows staged per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);  // [ROWS][CT]
    const int b = blockIdx.z, c0 = blockIdx.y * CT, t0 = blockIdx.x * TTF;
    const int cg = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int len = lens[b];
    const int c = c0 + cg * 8;
    // ---- stage: all loads first (rows outside [0, min(T, len)) are zero: "same" padding + frame mask)
    uint4 av[NLD], gv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int r = tl + 8 * i;
        int t = t0 + r - HALF;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);     // clamped address, masked below
        const int ld = LAYOUT == 2 ? d : 2 * d;
        const int ca = LAYOUT == 1 ? 64 * (c >> 5) + (c & 31) : c;
        const uint16_t* px = x + ((size_t)b * T + t) * ld + ca;
        av[i] = *reinterpret_cast<const uint4*>(px);
        if constexpr (LAYOUT != 2) gv[i] = *reinterpret_cast<const uint4*>(px + (LAYOUT == 1 ? 32 : d));
    }
    float wt[K][8];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        *reinterpret_cast<float4*>(&wt[j][0]) = *reinterpret_cast<const float4*>(w + (size_t)j * d + c);
        *reinterpret_cast<float4*>(&wt[j][4]) = *reinterpret_cast<const float4*>(w + (size_t)j * d + c + 4);
    }
    float bb[8];
    *reinterpret_cast<float4*>(&bb[0]) = *reinterpret_cast<const float4*>(bias + c);
    *reinterpret_cast<float4*>(&bb[4]) = *reinterpret_cast<const float4*>(bias + c + 4);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int r = tl + 8 * i;
        const int t = t0 + r - HALF;
        const bool ok = t >= 0 && t < T && t < len;
        const u16x8_t a = __builtin_bit_cast(u16x8_t, av[i]);
        float u[8];
        if constexpr (LAYOUT == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = ok ? bf16_to_f32(a[e]) : 0.0f;
        } else {
            const u16x8_t gt = __builtin_bit_cast(u16x8_t, gv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = ok ? bf16_to_f32(a[e]) * sigmoid_f(bf16_to_f32(gt[e])) : 0.0f;
        }
        if (r < ROWS) {
            float4* dst = reinterpret_cast<float4*>(tile + r * CT + cg * 8);
            dst[0] = make_float4(u[0], u[1], u[2], u[3]);
            dst[1] = make_float4(u[4], u[5], u[6], u[7]);
        }
    }
    __syncthreads();
    // ---- sliding window over this thread's R consecutive frames
    float acc[R][8];
#pragma unroll
    for (int o = 0; o < R; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = bb[e];
#pragma unroll
    for (int row = 0; row < R + K - 1; ++row) {
        const float4* src = reinterpret_cast<const float4*>(tile + (tl * R + row) * CT + cg * 8);
        const float4 x0 = src[0], x1 = src[1];
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int o = 0; o < R; ++o) {
            const int j = row - o;               // tap index, compile-time after unrolling
            if (j >= 0 && j < K) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(xv[e], wt[j][e], acc[o][e]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
        const int t = t0 + tl * R + o;
        if (t < T) {
            u16x8_t ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = f32_to_bf16(silu_f(acc[o][e]));
            *reinterpret_cast<u16x8_t*>(out + ((size_t)b * T + t) * d + c) = ov;
        }
    }
}

// The already-gated layout ([M][d] bf16: the GLU was applied by the pw1 GEMM epilogue — the product default) needs no
// arithmetic while staging, so its rows go HBM -> LDS as they are, by global_load_lds_dwordx4 (a wave instruction copies
// two 512-byte rows of the 256-channel tile; no VGPR round trip, no float conversion, half the LDS bytes of the f32
// tile above and conflict-free 16-byte reads).  Zero padding / the frame mask are applied when a row is read.
template <int K, int R>
__global__ __launch_bounds__(256) void dwconv_silu_dma_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const int32_t* __restrict__ lens,
                                                              int T, int d, uint16_t* __restrict__ out) {
    constexpr int TTF = 8 * R, ROWS = TTF + K - 1, HALF = (K - 1) / 2;
    static_assert(ROWS % 2 == 0, "a DMA instruction copies two rows");
    __shared__ __attribute__((aligned(16))) uint16_t tile[ROWS * CT];
    const int b = blockIdx.z, c0 = blockIdx.y * CT, t0 = blockIdx.x * TTF;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int len = lens[b];
    const int c = c0 + cg * 8;
    for (int inst = wave; inst < ROWS / 2; inst += 4) {           // wave-uniform trip count
        const int r = 2 * inst + (lane >> 5);
        int t = t0 + r - HALF;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);                     // clamped address, masked at the read
        const uint16_t* src = x + ((size_t)b * T + t) * d + c0 + (lane & 31) * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + inst * 2 * CT), 16, 0, 0);
    }
    float wt[K][8];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        *reinterpret_cast<float4*>(&wt[j][0]) = *reinterpret_cast<const float4*>(w + (size_t)j * d + c);
        *reinterpret_cast<float4*>(&wt[j][4]) = *reinterpret_cast<const float4*>(w + (size_t)j * d + c + 4);
    }
    float acc[R][8];
    {
        float bb[8];
        *reinterpret_cast<float4*>(&bb[0]) = *reinterpret_cast<const float4*>(bias + c);
        *reinterpret_cast<float4*>(&bb[4]) = *reinterpret_cast<const float4*>(bias + c + 4);
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = bb[e];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's rows landed; the barrier covers the others'
    __syncthreads();
#pragma unroll
    for (int row = 0; row < R + K - 1; ++row) {
        const int t = t0 + tl * R + row - HALF;
        const bool ok = t >= 0 && t < T && t < len;
        const u16x8_t a = *reinterpret_cast<const u16x8_t*>(tile + (tl * R + row) * CT + cg * 8);
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = ok ? bf16_to_f32(a[e]) : 0.0f;
#pragma unroll
        for (int o = 0; o < R; ++o) {
            const int j = row - o;               // tap index, compile-time after unrolling
            if (j >= 0 && j < K) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(xv[e], wt[j][e], acc[o][e]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
        const int t = t0 + tl * R + o;
        if (t < T) {
            u16x8_t ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = f32_to_bf16(silu_f(acc[o][e]));
            *reinterpret_cast<u16x8_t*>(out + ((size_t)b * T + t) * d + c) = ov;
        }
    }
}

// Depthwise Conv1d over time for a compile-time kernel size K on the already-gated layout ([M][d] bf16), any d % 64 == 0:
// the ESPnet conformer's k = 31 conv module and the Zipformer family's k = 31 / 15 / 7 modules (bias, no BatchNorm, SwooshR).
// A workgroup makes 128 frames x 64 channels: thread = (8 channels, 4 CONSECUTIVE frames); the input tile (128 + K - 1 frames,
// zero outside [0, min(T, len))) and the K taps are staged in LDS once; a thread reads each of its 4 + K - 1 input rows and each
// tap once (a sliding window: 48 LDS bytes per 32 FMAs where the generic kernel reads 48 per 8).  Rows are 160 bytes apart so
// that the 16 lanes a ds_read_b128 serves together (2 time lanes x 8 channel groups) cover all banks.
// ACT: 0 = SiLU (conformer conv module, BatchNorm folded into w / bias), 1 = SwooshR (icefall ConvolutionModule).
template <int K, int ACT>
__global__ __launch_bounds__(256) void dwconv_act_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const int32_t* __restrict__ lens,
                                                         int T, int d, uint16_t* __restrict__ out) {
    constexpr int R = 4, TTF = 32 * R, ROWS = TTF + K - 1, HALF = (K - 1) / 2, PITCH = 80;   // bf16 elements per LDS row
    __shared__ __attribute__((aligned(16))) uint16_t xs[ROWS * PITCH];
    __shared__ __attribute__((aligned(16))) float ws[K * 64];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * TTF;
    const int cg = threadIdx.x & 7, tl = threadIdx.x >> 3;
    int len = lens[b];
    len = len < T ? len : T;
    for (int idx = threadIdx.x; idx < ROWS * 8; idx += 256) {
        const int r = idx >> 3, g = idx & 7;
        const int t = t0 + r - HALF;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (t >= 0 && t < len) v = *reinterpret_cast<const uint4*>(x + ((size_t)b * T + t) * d + c0 + g * 8);
        *reinterpret_cast<uint4*>(xs + r * PITCH + g * 8) = v;
    }
    for (int idx = threadIdx.x; idx < K * 16; idx += 256) {
        const int j = idx >> 4, q = idx & 15;
        *reinterpret_cast<float4*>(ws + j * 64 + q * 4) = *reinterpret_cast<const float4*>(w + (size_t)j * d + c0 + q * 4);
    }
    const int c = c0 + cg * 8;
    float acc[R][8];
    {
        float bb[8];
        *reinterpret_cast<float4*>(&bb[0]) = *reinterpret_cast<const float4*>(bias + c);
        *reinterpret_cast<float4*>(&bb[4]) = *reinterpret_cast<const float4*>(bias + c + 4);
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = bb[e];
    }
    __syncthreads();
    // wt[o] holds tap (row - o) of this thread's channels, zero when that tap does not exist: every FMA is unconditional and
    // the loop stays rolled (unrolled by R, so the rotation of the tap registers is a renaming; fully unrolled the compiler
    // hoisted every load and spilled at K = 31)
    float wt[R][8];
#pragma unroll
    for (int o = 0; o < R; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) wt[o][e] = 0.0f;
#pragma unroll 4
    for (int row = 0; row < R + K - 1; ++row) {
        const u16x8_t a = *reinterpret_cast<const u16x8_t*>(xs + (tl * R + row) * PITCH + cg * 8);
#pragma unroll
        for (int o = R - 1; o > 0; --o)
#pragma unroll
            for (int e = 0; e < 8; ++e) wt[o][e] = wt[o - 1][e];
        const int jr = row < K ? row : 0;
        const float4 w0 = *reinterpret_cast<const float4*>(ws + jr * 64 + cg * 8), w1 = *reinterpret_cast<const float4*>(ws + jr * 64 + cg * 8 + 4);
        const float keep = row < K ? 1.0f : 0.0f;
        wt[0][0] = w0.x * keep; wt[0][1] = w0.y * keep; wt[0][2] = w0.z * keep; wt[0][3] = w0.w * keep;
        wt[0][4] = w1.x * keep; wt[0][5] = w1.y * keep; wt[0][6] = w1.z * keep; wt[0][7] = w1.w * keep;
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(bf16_to_f32(a[e]), wt[o][e], acc[o][e]);
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
        const int t = t0 + tl * R + o;
        if (t < T) {
            u16x8_t ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = f32_to_bf16(ACT == 1 ? swoosh_r_f(acc[o][e]) : silu_f(acc[o][e]));
            *reinterpret_cast<u16x8_t*>(out + ((size_t)b * T + t) * d + c) = ov;
        }
    }
}

int g_glu_generic = 0;

}  // namespace

// A/B hook (scripts/elementwise_bench.py): 1 = the generic any-kernel-size path, 2 = the f32-tile fast kernel also for the
// gated layout (instead of the DMA-staged one)
extern "C" void rs_debug_set_glu_generic(int v) { g_glu_generic = v; }

int rs_launch_layernorm(rs_ctx* ctx, const float* x, const float* g, const float* b, int M, int d, float eps,
                        uint16_t* out_bf16, float* out_f32, hipStream_t s) {
    if (M <= 0) return RS_OK;
    if (d % 256 || d > 2048) return rs_fail(ctx, RS_EINVAL, "layernorm: d=%d must be a multiple of 256, <= 2048", d);
    const dim3 grid((M + 3) / 4), block(256);
    const double bytes = (double)M * d * (4.0 + (out_bf16 ? 2.0 : 0.0) + (out_f32 ? 4.0 : 0.0));
    rs_prof_begin(ctx, RS_PROF_ELEMENTWISE, s, 8.0 * M * d, bytes);
    switch (d / 256) {
#define LN_CASE(NV) case NV: hipLaunchKernelGGL(layernorm_kernel<NV>, grid, block, 0, s, x, g, b, M, eps, out_bf16, out_f32); break;
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
#undef LN_CASE
    }
    rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s);
    RS_CHECK_LAUNCH(ctx, "layernorm");
    return RS_OK;
}

int rs_launch_layernorm2(rs_ctx* ctx, const float* x, const float* g1, const float* b1, const float* g2, const float* b2,
                         int M, int d, float eps, float* out_f32, uint16_t* out_bf16, float* stats, hipStream_t s) {
    if (M <= 0) return RS_OK;
    if (d % 256 || d > 2048) return rs_fail(ctx, RS_EINVAL, "layernorm: d=%d must be a multiple of 256, <= 2048", d);
    const dim3 grid((M + 3) / 4), block(256);
    if (!out_f32 && !stats) return rs_fail(ctx, RS_EINVAL, "layernorm2: neither the first norm's rows nor its statistics requested");
    rs_prof_begin(ctx, RS_PROF_ELEMENTWISE, s, 16.0 * M * d, (double)M * d * (out_f32 ? 10.0 : 6.0));
    switch (d / 256) {
#define LN_CASE(NV) case NV: hipLaunchKernelGGL(layernorm2_kernel<NV>, grid, block, 0, s, x, g1, b1, g2, b2, M, eps, out_f32, out_bf16, stats); break;
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
#undef LN_CASE
    }
    rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s);
    RS_CHECK_LAUNCH(ctx, "layernorm2");
    return RS_OK;
}

int rs_launch_dwconv_act(rs_ctx* ctx, const uint16_t* x, const float* w, const float* b, const int32_t* lens, int B, int T, int d, int k,
                         int act, uint16_t* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return RS_OK;
    if (d % 64) return rs_fail(ctx, RS_EINVAL, "dwconv: d=%d must be a multiple of 64", d);
    if (act < 0 || act > 1) return rs_fail(ctx, RS_EINVAL, "dwconv: unknown activation %d", act);
    const dim3 grid((T + 127) / 128, d / 64, B), block(256);
    rs_prof_begin(ctx, RS_PROF_ELEMENTWISE, s, (double)B * T * d * (2.0 * k + 12.0), (double)B * T * d * 4.0);
#define RS_DW(KK)                                                                                              \
    case KK:                                                                                                   \
        if (act) hipLaunchKernelGGL((dwconv_act_kernel<KK, 1>), grid, block, 0, s, x, w, b, lens, T, d, out);  \
        else hipLaunchKernelGGL((dwconv_act_kernel<KK, 0>), grid, block, 0, s, x, w, b, lens, T, d, out);      \
        break;
    switch (k) {
        RS_DW(7) RS_DW(15) RS_DW(31)
        default:
            rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s);
            return rs_fail(ctx, RS_EINVAL, "dwconv: kernel size %d is not built (7, 15, 31)", k);
    }
#undef RS_DW
    rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s);
    RS_CHECK_LAUNCH(ctx, "dwconv_act");
    return RS_OK;
}

int rs_launch_glu_dwconv(rs_ctx* ctx, const uint16_t* x, int layout, const float* w, const float* b, const int32_t* lens,
                         int B, int T, int d, int k, uint16_t* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return RS_OK;
    if (d % CT) return rs_fail(ctx, RS_EINVAL, "glu_dwconv: d=%d must be a multiple of %d", d, CT);
    if (k < 1 || k > KMAX || !(k & 1)) return rs_fail(ctx, RS_EINVAL, "glu_dwconv: kernel size %d unsupported", k);
    if (layout < 0 || layout > 2) return rs_fail(ctx, RS_EINVAL, "glu_dwconv: unknown input layout %d", layout);
    // the gated layout with a kernel size the sliding-window kernel is built for (the ESPnet conformer's k = 31)
    if (layout == 2 && (k == 7 || k == 15 || k == 31) && g_glu_generic != 1) return rs_launch_dwconv_act(ctx, x, w, b, lens, B, T, d, k, 0, out, s);
    const double bytes = (double)B * T * d * ((layout == 2 ? 2.0 : 4.0) + 2.0);
    rs_prof_begin(ctx, RS_PROF_ELEMENTWISE, s, (double)B * T * d * (2.0 * k + 12.0), bytes);
    if (k == 9 && g_glu_generic != 1) {
        // register-window fast path (the FastConformer kernel size); 48-frame tiles
        constexpr int R = 6, TTF = 8 * R;
        const dim3 grid((T + TTF - 1) / TTF, d / CT, B), block(256);
        const size_t lds = (size_t)(TTF + 9 - 1) * CT * sizeof(float);
#define RS_GLU_FAST(LY)                                                                                                   \
        do {                                                                                                              \
            if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)glu_dwconv_silu_fast_kernel<9, R, LY>, (int)lds); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s); return rc; } \
            hipLaunchKernelGGL((glu_dwconv_silu_fast_kernel<9, R, LY>), grid, block, lds, s, x, w, b, lens, T, d, out);   \
        } while (0)
        if (layout == 0) RS_GLU_FAST(0);
        else if (layout == 1) RS_GLU_FAST(1);
        else if (g_glu_generic == 2) RS_GLU_FAST(2);          // A/B: the f32-tile kernel on the gated layout
        else hipLaunchKernelGGL((dwconv_silu_dma_kernel<9, R>), grid, block, 0, s, x, w, b, lens, T, d, out);
#undef RS_GLU_FAST
    } else {
        const dim3 grid((T + TT - 1) / TT, d / CT, B), block(256);
        const size_t lds = (size_t)(TT + k - 1) * CT * sizeof(float);
        hipLaunchKernelGGL(glu_dwconv_silu_kernel, grid, block, lds, s, x, w, b, lens, T, d, k, layout, out);
    }
    rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s);
    RS_CHECK_LAUNCH(ctx, "glu_dwconv_silu");
    return RS_OK;
}

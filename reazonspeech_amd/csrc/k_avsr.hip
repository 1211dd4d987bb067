// k_avsr.hip — the AV-HuBERT encoder-decoder of `reazonspeech.avsr` (BASELINE.json configs[4]: "audio-visual
// path: video-frame + log-mel fusion encoder, batch = 16 clips").
//
// The reference is in-tree torch code (pkg/avsr/src/avhubert/): modeling_resnet.py:140-178 (Conv3d front-end + ResNet-18 trunk),
// modeling_avhubert.py:40-65 (audio / video projections), :162-213 (fusion, LayerNorm, post_extract_proj, transformers'
// HubertEncoder), decoder.py:297-369, :467-617 (Transformer decoder), generate() through transformers' GenerationMixin
// (modeling_avhubert.py:372-391 re-feeds the prefix and re-runs the encoder every step).  Restated in oracle/avsr.py, which is
// pinned to the reference itself (tests/golden/avsr_ref_*.npz).  What runs where:
//
//   dense contractions     every Linear, the 3 x 3 / 1 x 1 convolutions of the ResNet trunk (channels-last patches gathered by the loader) on
//                          rs_launch_gemm_f32's exact v_mfma_f32_16x16x4_f32 chain: the reference computes float32 and so does
//                          this path, end to end (there is no reduced-precision mode of this family yet)
//   Conv3d(1, 64, 5x7x7)   avsr_conv3d_kernel: the five frames' seven input rows of an output row staged in LDS, the 245 taps (padded to
//                          280) on the exact-f32 MFMA, a wave per 16 channels; BatchNorm + PReLU in its epilogue
//   positional conv        avsr_posconv_kernel: grouped Conv1d (kernel 128, 16 groups) with the group's frames in LDS, GELU and the
//                          residual add fused
//   attention              avsr_attn_kernel: one wave per (query, head), two passes over the visible keys; serves the encoder
//                          (padding mask), the decoder's self-attention over its KV cache and the cross-attention
//   decoder                rs_avsr_decoder_begin projects the encoder states once per layer (cross-attention K / V),
//                          rs_avsr_decoder_step advances every hypothesis row by one token with a per-layer KV cache that is
//                          re-gathered by beam index — the reference recomputes everything every step; results are the same
//
// Batch semantics: the reference runs the whole padded batch through the front-ends (padded frames are real inputs to the
// Conv3d) and masks only attention keys and the encoder input rows; so does this file — a clip's result inside a batch equals the
// reference's result for that batch.
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "rs_common.h"

struct rs_avsr_block {
    const float *conv1_w, *conv2_w, *bn1_a, *bn1_b, *bn2_a, *bn2_b, *relu1, *relu2, *ds_w, *ds_a, *ds_b;
};
struct rs_avsr_attn {
    const float *qkv_w, *qkv_b, *o_w, *o_b;          // self-attention: q | k | v rows concatenated
    const float *q_w, *q_b, *kv_w, *kv_b;            // cross-attention: q, then k | v
};
struct rs_avsr_layer {
    rs_avsr_attn sa, ca;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b, *ff1_w, *ff1_b, *ff2_w, *ff2_b;
};
struct rs_avsr {
    rs_avsr_dims d{};
    const float *audio_w = nullptr, *audio_b = nullptr, *conv3d_w = nullptr, *bn0_a = nullptr, *bn0_b = nullptr, *prelu0 = nullptr;
    rs_avsr_block blocks[4][2] = {};
    const float *vproj_w = nullptr, *vproj_b = nullptr, *fuse_g = nullptr, *fuse_b = nullptr, *fproj_w = nullptr, *fproj_b = nullptr;
    const float *pos_w = nullptr, *pos_b = nullptr, *encln_g = nullptr, *encln_b = nullptr;
    std::vector<rs_avsr_layer> enc, dec;
    const float *embed = nullptr, *dec_pos = nullptr, *decln_g = nullptr, *decln_b = nullptr, *lm_w = nullptr;
    // parity taps (rs_avsr_encoder_set_taps)
    float *tap_video = nullptr, *tap_fused = nullptr, *tap_encln = nullptr, *tap_layers = nullptr;
    std::vector<int> tap_ids;
};

namespace {

__host__ __device__ inline int pad32(int n) { return (n + 31) / 32 * 32; }
__host__ __device__ inline int pad4(int n) { return (n + 3) / 4 * 4; }
constexpr int TRUNK_C[5] = {64, 64, 128, 256, 512};

// ---- video front-end ----------------------------------------------------------------------------------------------------------------
// Conv3d(1, 64, (5, 7, 7), stride (1, 2, 2), padding (2, 3, 3)) + BatchNorm3d (inference form: x * alpha + beta) + PReLU.
// pixels f32 [B][T][H][W] -> out f32 [B*T][H/2][W/2][64] (channels last).  Frames outside [0, T) and pixels outside the image are zeros
// (the convolution's own padding); padded frames of a clip are ordinary inputs, as in the reference.
//
// On the exact-f32 matrix cores (round 6; the VALU form — a thread per channel and every fourth pixel — ran at 23 TF/s, 10.4 ms per
// 16 x 10 s): grid (H/2, T, B), a workgroup computes one output row; its 35 input rows (5 frames x 7 rows) sit in LDS with three
// zero columns of padding on the left and five on the right.  The contraction index is laid out as k' = 8 (7 kt + kh) + kw with the
// eighth tap of every row a ZERO weight, so that one v_mfma_f32_16x16x4_f32 step s covers taps kw = 4 (s & 1) + 0..3 of input row
// s / 2 and a lane's patch operand is rows[s / 2][2 ow + 4 (s & 1) + kq]: the address is the lane's base plus a compile-time offset.
// Wave w owns channels 16 w .. + 15 (the weight operand: 70 registers, loaded once) and walks the row's pixel tiles of 16.  The taps
// of an output element are added in the order (kt, kh, kw) — the order of the VALU form; the zero tap adds an exact zero.
__global__ __launch_bounds__(256) void avsr_conv3d_kernel(const float* __restrict__ pix, int T, int H, int W, const float* __restrict__ w /* [245][64] */,
                                                          const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ slope,
                                                          float* __restrict__ out) {
    constexpr int WMAX = 96, RP = WMAX + 8;               // row pitch: x = 2 ow + kw <= 2 * 47 + 7
    __shared__ float rows[35][RP];
    const int oh = blockIdx.x, t = blockIdx.y, b = blockIdx.z, OW = W / 2, OH = H / 2;
    for (int i = threadIdx.x; i < 35 * RP; i += 256) {
        const int r = i / RP, x = i - r * RP, kt = r / 7, kh = r - 7 * kt;
        const int tt = t + kt - 2, ih = 2 * oh + kh - 3, iw = x - 3;
        rows[r][x] = (tt >= 0 && tt < T && ih >= 0 && ih < H && iw >= 0 && iw < W) ? pix[(((size_t)b * T + tt) * H + ih) * W + iw] : 0.0f;
    }
    const int lane = threadIdx.x & 63, ct = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    float wreg[70];
#pragma unroll
    for (int s = 0; s < 70; ++s) {
        const int kw = 4 * (s & 1) + kq;
        const float v = w[((s >> 1) * 7 + (kw < 7 ? kw : 6)) * 64 + 16 * ct + li];
        wreg[s] = kw < 7 ? v : 0.0f;
    }
    __syncthreads();
    const int tiles = (OW + 15) / 16;                      // <= 3
    const float4 a4 = *reinterpret_cast<const float4*>(alpha + 16 * ct + 4 * kq), b4 = *reinterpret_cast<const float4*>(beta + 16 * ct + 4 * kq);
    const float4 s4 = *reinterpret_cast<const float4*>(slope + 16 * ct + 4 * kq);
    float* orow = out + (((size_t)b * T + t) * OH + oh) * OW * 64 + 16 * ct + 4 * kq;
    for (int pt = 0; pt < tiles; ++pt) {
        const float* pr = &rows[0][2 * (16 * pt + li) + kq];
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 70; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], pr[(s >> 1) * RP + 4 * (s & 1)], acc, 0, 0, 0);
        const int ow = 16 * pt + li;
        if (ow < OW) {
            float4 v = make_float4(fmaf(acc[0], a4.x, b4.x), fmaf(acc[1], a4.y, b4.y), fmaf(acc[2], a4.z, b4.z), fmaf(acc[3], a4.w, b4.w));
            v.x = v.x >= 0.0f ? v.x : v.x * s4.x; v.y = v.y >= 0.0f ? v.y : v.y * s4.y;
            v.z = v.z >= 0.0f ? v.z : v.z * s4.z; v.w = v.w >= 0.0f ? v.w : v.w * s4.w;
            *reinterpret_cast<float4*>(orow + (size_t)ow * 64) = v;
        }
    }
}

// MaxPool over (3, 3) windows, stride 2, padding 1 (-inf): [N][H][W][C] -> [N][(H + 1) / 2][(W + 1) / 2][C]
__global__ __launch_bounds__(256) void avsr_maxpool_kernel(const float* __restrict__ in, int H, int W, int C, int OH, int OW, size_t total, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    size_t r = idx / C;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const size_t n = r / OH;
    float m = -INFINITY;
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = 2 * oh + kh - 1;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = 2 * ow + kw - 1;
            if (iw < 0 || iw >= W) continue;
            m = fmaxf(m, in[((n * H + ih) * W + iw) * C + c]);
        }
    }
    out[idx] = m;
}

// the 1 x 1 stride-2 down-sampling convolution's input: in[n][2 oh][2 ow][:] -> rows of C
__global__ __launch_bounds__(256) void avsr_stride2_kernel(const float* __restrict__ in, int H, int W, int C, int OH, int OW, size_t pieces, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= pieces) return;
    const int c4 = C / 4;
    const size_t row = idx / c4;
    const int cc = (int)(idx - row * c4), ow = (int)(row % OW);
    const size_t r = row / OW;
    const int oh = (int)(r % OH);
    const size_t n = r / OH;
    reinterpret_cast<float4*>(out)[idx] = reinterpret_cast<const float4*>(in + ((n * H + 2 * oh) * W + 2 * ow) * C)[cc];
}

// AdaptiveAvgPool2d(1): [N][HW][C] -> [N][C]
__global__ __launch_bounds__(256) void avsr_avgpool_kernel(const float* __restrict__ in, int HW, int C, size_t total, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t n = idx / C;
    const int c = (int)(idx - n * C);
    float s = 0.0f;
    for (int p = 0; p < HW; ++p) s += in[(n * HW + p) * C + c];
    out[idx] = s / (float)HW;
}

// rows [M][K] -> [M][Kp] zero-padded (the audio features' K = 104 -> 128)
__global__ __launch_bounds__(256) void avsr_padcols_kernel(const float* __restrict__ in, int K, int Kp, size_t total, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t m = idx / Kp;
    const int k = (int)(idx - m * Kp);
    out[idx] = k < K ? in[m * K + k] : 0.0f;
}

// LayerNorm over the last axis, one wave per row: mean, then the biased variance of the centred values (two passes, like torch),
// (x - mean) / sqrt(var + eps) * g + b.  `zero_rows` (optional, [M], nonzero = write zeros instead: HubertEncoder zeroes padded
// frames before the positional convolution) is applied to the OUTPUT.
__global__ __launch_bounds__(256) void avsr_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, int M, int d,
                                                             float eps, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.0f;
    for (int c = lane; c < d; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)d;
    float v = 0.0f;
    for (int c = lane; c < d; c += 64) { const float e = xr[c] - mean; v = fmaf(e, e, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
    for (int c = lane; c < d; c += 64) out[(size_t)row * d + c] = (xr[c] - mean) * rstd * g[c] + b[c];
}

// rows whose mask entry is nonzero become zeros (HubertEncoder.forward: hidden_states[~attention_mask] = 0)
__global__ __launch_bounds__(256) void avsr_mask_rows_kernel(float* __restrict__ x, const float* __restrict__ mask, int d, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    if (mask[idx / d] != 0.0f) x[idx] = 0.0f;
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// HubertPositionalConvEmbedding + residual: y[b][t][g cg + oc] = x + gelu(bias + sum_{k, ic} w[g][k][ic][oc] x[b][t + k - K / 2][g cg + ic])
// (an even kernel drops its last output frame: the window of frame t is [t - K/2, t + K/2 - 1]).  grid (ceil(T / 8), G, B); the
// group's frames [t0 - K/2, t0 + 8 + K/2) sit in LDS; a thread owns (frame, output channel) pairs, taps then input channels ascending.
__global__ __launch_bounds__(256) void avsr_posconv_kernel(const float* __restrict__ x, int T, int d, int cg, int K, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);                    // [8 + K][cg]
    const int t0 = blockIdx.x * 8, g = blockIdx.y, b = blockIdx.z, half = K / 2;
    for (int i = threadIdx.x; i < (8 + K) * cg; i += 256) {
        const int r = i / cg, c = i - r * cg, t = t0 + r - half;
        xs[i] = (t >= 0 && t < T) ? x[((size_t)b * T + t) * d + g * cg + c] : 0.0f;
    }
    __syncthreads();
    const float* wg = w + (size_t)g * K * cg * cg;
    for (int o = threadIdx.x; o < 8 * cg; o += 256) {
        const int tl = o / cg, oc = o - tl * cg, t = t0 + tl;
        if (t >= T) continue;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float* xr = xs + (tl + k) * cg;
            const float* wr = wg + (size_t)k * cg * cg + oc;
            for (int ic = 0; ic < cg; ++ic) acc = fmaf(wr[(size_t)ic * cg], xr[ic], acc);
        }
        const size_t at = ((size_t)b * T + t) * d + g * cg + oc;
        out[at] = x[at] + gelu_exact(acc + bias[g * cg + oc]);
    }
}

// The same positional convolution on the matrix cores (round 6: the form above ran 18 ms per batch of 16 x 250 frames, one scalar FMA
// chain of 6144 terms per output: profiles/r06_05_f32_avsr_kernel_stats_before.txt).  Per tap k the group's product is a
// [cg out] x [cg in] x [frames] matrix product: v_mfma_f32_16x16x4_f32 with the weights w[g][k][ic][oc] as the first operand (16
// output channels x 4 input channels, read from L2: 4 runs of 64 bytes per wave instruction) and the frames' inputs as the second
// (16 frames x 4 input channels from the LDS tile, row pitch cg + 1: conflict-free), so a lane's accumulator registers are four
// consecutive output channels of one frame.  A wave owns 16 frames x cg output channels of one group (cg / 16 accumulator tiles);
// a workgroup 64 frames.  Order of a sum: taps ascending, input channels ascending in 4-deep exact-f32 chains, bias last.
// cg % 16 == 0, cg <= 64; grid (ceil(T / 64), G, B)
template <int NT>
__global__ __launch_bounds__(256) void avsr_posconv_mfma_kernel(const float* __restrict__ x, int T, int d, int K, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ out) {
    constexpr int CG = NT * 16, PITCH = CG + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);                    // [64 + K][PITCH]
    const int t0 = blockIdx.x * 64, g = blockIdx.y, b = blockIdx.z, half = K / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    for (int i = threadIdx.x; i < (64 + K) * CG; i += 256) {
        const int r = i / CG, c = i - r * CG, t = t0 + r - half;
        xs[r * PITCH + c] = (t >= 0 && t < T) ? x[((size_t)b * T + t) * d + g * CG + c] : 0.0f;
    }
    __syncthreads();
    f32x4_t acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const float* wg = w + (size_t)g * K * CG * CG;
    const float* xrow = xs + (wave * 16 + li) * PITCH;             // this lane's frame, tap 0
    for (int k = 0; k < K; ++k) {
        const float* wk = wg + (size_t)k * CG * CG;
        const float* xr = xrow + k * PITCH;
#pragma unroll
        for (int ic0 = 0; ic0 < CG; ic0 += 4) {
            const float xv = xr[ic0 + kq];
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[(size_t)(ic0 + kq) * CG + n * 16 + li], xv, acc[n], 0, 0, 0);
        }
    }
    const int t = t0 + wave * 16 + li;
    if (t < T) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int oc = g * CG + n * 16 + 4 * kq;
            const size_t at = ((size_t)b * T + t) * d + oc;
            const float4 xin = *reinterpret_cast<const float4*>(x + at), bb = *reinterpret_cast<const float4*>(bias + oc);
            float4 o;
            o.x = xin.x + gelu_exact(acc[n][0] + bb.x); o.y = xin.y + gelu_exact(acc[n][1] + bb.y);
            o.z = xin.z + gelu_exact(acc[n][2] + bb.z); o.w = xin.w + gelu_exact(acc[n][3] + bb.w);
            *reinterpret_cast<float4*>(out + at) = o;
        }
    }
}

// Scaled dot-product attention, one wave per (query, head, query batch row); a lane owns head-dim elements lane, lane + 64, ..
//   s_j = (q . k_j) * scaling over the visible keys j < n_keys, j <= causal limit, kmask[kb][j] == 0;  out = softmax(s) v
// Two passes (maximum, then exp / sum / PV; scores recomputed).  Query rows qb = blockIdx.z read keys of batch kb = qb / rows_per_kb.
// q [Bq][Tq][ldq], k / v [Bk][Tk_pitch][ldk] (+ head offset), out [Bq][Tq][ldo].
template <int NV>
__global__ __launch_bounds__(256) void avsr_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldk,
                                                        size_t k_batch_stride, const float* __restrict__ kmask, int mask_pitch, int rows_per_kb, int Tq,
                                                        int n_keys, int hd, float scaling, int causal, float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, qb = blockIdx.z;
    if (i >= Tq) return;
    const int kb = qb / rows_per_kb;
    const float* qr = q + ((size_t)qb * Tq + i) * ldq + h * hd;
    const float* kbase = k + (size_t)kb * k_batch_stride + h * hd;
    const float* vbase = v + (size_t)kb * k_batch_stride + h * hd;
    const float* mk = kmask ? kmask + (size_t)kb * mask_pitch : nullptr;
    const int last = causal ? (i < n_keys - 1 ? i : n_keys - 1) : n_keys - 1;
    float qv[NV], acc[NV];
    bool ok[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        ok[e] = lane + 64 * e < hd;
        qv[e] = ok[e] ? qr[lane + 64 * e] : 0.0f;
        acc[e] = 0.0f;
    }
    auto score = [&](int j) -> float {
        const float* kr = kbase + (size_t)j * ldk;
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < NV; ++e)
            if (ok[e]) s = fmaf(qv[e], kr[lane + 64 * e], s);
        return wave_sum(s) * scaling;
    };
    float mx = -INFINITY;
    for (int j = 0; j <= last; ++j)
        if (!mk || mk[j] == 0.0f) mx = fmaxf(mx, score(j));
    float den = 0.0f;
    for (int j = 0; j <= last; ++j) {
        if (mk && mk[j] != 0.0f) continue;
        const float p = expf(score(j) - mx);
        den += p;
        const float* vr = vbase + (size_t)j * ldk;
#pragma unroll
        for (int e = 0; e < NV; ++e)
            if (ok[e]) acc[e] = fmaf(p, vr[lane + 64 * e], acc[e]);
    }
    float* orow = out + ((size_t)qb * Tq + i) * ldo + h * hd;
#pragma unroll
    for (int e = 0; e < NV; ++e)
        if (ok[e]) orow[lane + 64 * e] = den > 0.0f ? acc[e] / den : 0.0f;
}

// The same attention with the lanes over the KEYS (see attention_f32_keys_kernel in k_f32.hip: the per-key wave reductions of the form
// above were 30 % of this family's kernel time, profiles/r06_05_f32_avsr_kernel_stats_before.txt): a lane computes the whole dot
// products of keys lane, lane + 64, .. against the wave's query (in LDS, read as broadcasts), two wave reductions per query, then the
// P.V sum walks the keys with coalesced value rows.  n_keys <= 64 NCH; more keys take the kernel above.
template <int NCH>
__global__ __launch_bounds__(256) void avsr_attn_keys_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldk,
                                                             size_t k_batch_stride, const float* __restrict__ kmask, int mask_pitch, int rows_per_kb, int Tq,
                                                             int n_keys, int hd, float scaling, float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) float qs[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = blockIdx.x * 4 + wave, h = blockIdx.y, qb = blockIdx.z;
    if (i >= Tq) return;
    const int kb = qb / rows_per_kb;
    const float* qr = q + ((size_t)qb * Tq + i) * ldq + h * hd;
    const float* kbase = k + (size_t)kb * k_batch_stride + h * hd;
    const float* vbase = v + (size_t)kb * k_batch_stride + h * hd;
    const float* mk = kmask ? kmask + (size_t)kb * mask_pitch : nullptr;
    for (int e = lane; e < hd; e += 64) qs[wave][e] = qr[e];
    __builtin_amdgcn_wave_barrier();
    const float4* q4 = reinterpret_cast<const float4*>(qs[wave]);
    float sc[NCH];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = lane + 64 * c;
        sc[c] = -INFINITY;
        if (j < n_keys && (!mk || mk[j] == 0.0f)) {
            const float4* kr = reinterpret_cast<const float4*>(kbase + (size_t)j * ldk);
            float a = 0.0f;
            for (int e4 = 0; e4 < hd / 4; ++e4) {
                const float4 kv = kr[e4], qq = q4[e4];
                a = fmaf(qq.x, kv.x, a); a = fmaf(qq.y, kv.y, a); a = fmaf(qq.z, kv.z, a); a = fmaf(qq.w, kv.w, a);
            }
            sc[c] = a * scaling;
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
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int jn = n_keys - 64 * c < 64 ? n_keys - 64 * c : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const float p = __shfl(sc[c], jj, 64);
            if (p != 0.0f) {
                const float* vr = vbase + (size_t)(64 * c + jj) * ldk;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (lane + 64 * e < hd) acc[e] = fmaf(p, vr[lane + 64 * e], acc[e]);
            }
        }
    }
    float* orow = out + ((size_t)qb * Tq + i) * ldo + h * hd;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (lane + 64 * e < hd) orow[lane + 64 * e] = den > 0.0f ? acc[e] / den : 0.0f;
}

// Encoder self-attention on the exact-f32 matrix cores (the form of attention_f32_mfma_kernel in k_f32.hip without the position
// term): a wave owns 16 queries, walks the keys 16 at a time — S^T = K . Q^T with the key rows as the MFMA's first operand straight
// from global memory, online softmax per tile (scores scaled AFTER the dot product, like the reference; a masked key weighs 0), and
// O^T += V^T . P^T with the four keys of a step chosen so that a lane's own probability register is its operand.  Padded QUERIES
// are computed like any other (the reference does not mask them either).  HD % 64 == 0.
template <int HD>
__global__ __launch_bounds__(256) void avsr_attn_mfma_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldk,
                                                             size_t k_batch_stride, const float* __restrict__ kmask, int mask_pitch, int rows_per_kb, int Tq,
                                                             int n_keys, float scaling, float* __restrict__ out, int ldo) {
    constexpr int NS = HD / 16, NG = HD / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.x * 64 + wave * 16, h = blockIdx.y, qb = blockIdx.z;
    if (i0 >= Tq) return;
    const int kb = qb / rows_per_kb;
    const float* kbase = k + (size_t)kb * k_batch_stride + h * HD;
    const float* vbase = v + (size_t)kb * k_batch_stride + h * HD;
    const float* mk = kmask ? kmask + (size_t)kb * mask_pitch : nullptr;
    const int qi = i0 + li, qrow = qi < Tq ? qi : Tq - 1;
    const float* qr = q + ((size_t)qb * Tq + qrow) * ldq + h * HD + 4 * kq;
    float4 qf[NS];
#pragma unroll
    for (int S = 0; S < NS; ++S) qf[S] = *reinterpret_cast<const float4*>(qr + 16 * S);
    f32x4_t O[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int n = 0; n < 4; ++n) O[g][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const int ntile = (n_keys + 15) / 16;
    for (int jt = 0; jt < ntile; ++jt) {
        const int j0 = jt * 16;
        const int krow = j0 + li < n_keys ? j0 + li : n_keys - 1;
        const float* kp = kbase + (size_t)krow * ldk + 4 * kq;
        f32x4_t ac = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            const float4 kf = *reinterpret_cast<const float4*>(kp + 16 * S);
            ac = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[S].x, ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[S].y, ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[S].z, ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[S].w, ac, 0, 0, 0);
        }
        float sc[4];
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * kq + r;
            const bool ok = j < n_keys && (!mk || mk[j < n_keys ? j : n_keys - 1] == 0.0f);
            sc[r] = ok ? ac[r] * scaling : -INFINITY;
            mt = fmaxf(mt, sc[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        if (m_new == -INFINITY) continue;                  // every key so far is masked — the same for all 16 queries of the wave (they
                                                           // share the clip's key mask): a wave-uniform skip, no MFMA under a partial EXEC
        const float alpha = expf(m_run - m_new);           // exp(-inf) = 0 on the first visible tile
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
            const int vrow = j0 + 4 * kq + m < n_keys ? j0 + 4 * kq + m : n_keys - 1;
            const float* vp = vbase + (size_t)vrow * ldk + 4 * li;
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
    if (qi < Tq) {
        const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
        float* orow = out + ((size_t)qb * Tq + qi) * ldo + h * HD;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(orow + 64 * g + 16 * kq + 4 * r) =
                    make_float4(O[g][0][r] * inv, O[g][1][r] * inv, O[g][2][r] * inv, O[g][3][r] * inv);
    }
}

// One decoding step (Tq == 1): the kernel above gives a (row, head) to ONE wave — 320 waves on a 1024-SIMD chip walking 250 keys
// each, 157 us per launch and 12 launches per token (profiles/r06_09_f32_avsr_kernel_stats.txt: more than half of a step).  Here a
// workgroup owns the (row, head) and its four waves split the keys (chunk c * 4 + wave of 64): a lane computes the whole dot
// product of its key, a wave its own max / sum / P.V partial, and the four partials combine in LDS with the usual
// exp(m_w - M) weights.  n_keys <= 256 NC.
template <int NC>
__global__ __launch_bounds__(256) void avsr_attn_step_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldk,
                                                             size_t k_batch_stride, const float* __restrict__ kmask, int mask_pitch, int rows_per_kb, int n_keys,
                                                             int hd, float scaling, float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) float qs[256];
    __shared__ float part_m[4], part_l[4];
    __shared__ float part_acc[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.x, qb = blockIdx.y;
    const int kb = qb / rows_per_kb;
    const float* qr = q + (size_t)qb * ldq + h * hd;
    const float* kbase = k + (size_t)kb * k_batch_stride + h * hd;
    const float* vbase = v + (size_t)kb * k_batch_stride + h * hd;
    const float* mk = kmask ? kmask + (size_t)kb * mask_pitch : nullptr;
    if ((int)threadIdx.x < hd) qs[threadIdx.x] = qr[threadIdx.x];
    __syncthreads();
    const float4* q4 = reinterpret_cast<const float4*>(qs);
    float sc[NC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = (c * 4 + wave) * 64 + lane;
        sc[c] = -INFINITY;
        if (j < n_keys && (!mk || mk[j] == 0.0f)) {
            const float4* kr = reinterpret_cast<const float4*>(kbase + (size_t)j * ldk);
            float a0 = 0.0f, a1 = 0.0f;
            int e4 = 0;
            for (; e4 + 1 < hd / 4; e4 += 2) {                        // two chains: the loads of a key row are independent
                const float4 k0 = kr[e4], q0 = q4[e4], k1 = kr[e4 + 1], q1 = q4[e4 + 1];
                a0 = fmaf(q0.x, k0.x, a0); a0 = fmaf(q0.y, k0.y, a0); a0 = fmaf(q0.z, k0.z, a0); a0 = fmaf(q0.w, k0.w, a0);
                a1 = fmaf(q1.x, k1.x, a1); a1 = fmaf(q1.y, k1.y, a1); a1 = fmaf(q1.z, k1.z, a1); a1 = fmaf(q1.w, k1.w, a1);
            }
            if (e4 < hd / 4) {
                const float4 k0 = kr[e4], q0 = q4[e4];
                a0 = fmaf(q0.x, k0.x, a0); a0 = fmaf(q0.y, k0.y, a0); a0 = fmaf(q0.z, k0.z, a0); a0 = fmaf(q0.w, k0.w, a0);
            }
            sc[c] = (a0 + a1) * scaling;
            mx = fmaxf(mx, sc[c]);
        }
    }
    mx = wave_max(mx);
    float den = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        sc[c] = sc[c] > -INFINITY ? expf(sc[c] - mx) : 0.0f;
        den += sc[c];
    }
    den = wave_sum(den);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j0 = (c * 4 + wave) * 64;
        const int jn = n_keys - j0 < 64 ? n_keys - j0 : 64;
        // eight value rows per trip, their loads issued before the first product (a masked key weighs 0: its row is multiplied, not
        // skipped — one branch-free chain of L2 round trips per eight keys instead of one per key)
        for (int jj = 0; jj < jn; jj += 8) {
            float vv[8][4], pj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ju = jj + u < jn ? jj + u : jn - 1;
                const float* vr = vbase + (size_t)(j0 + ju) * ldk;
                pj[u] = jj + u < jn ? __shfl(sc[c], ju, 64) : 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[u][e] = lane + 64 * e < hd ? vr[lane + 64 * e] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(pj[u], vv[u][e], acc[e]);
        }
    }
    if (lane == 0) { part_m[wave] = mx; part_l[wave] = den; }
#pragma unroll
    for (int e = 0; e < 4; ++e) part_acc[wave][lane + 64 * e] = acc[e];
    __syncthreads();
    if ((int)threadIdx.x < hd) {
        const float M = fmaxf(fmaxf(part_m[0], part_m[1]), fmaxf(part_m[2], part_m[3]));
        float L = 0.0f, o = 0.0f;
        if (M > -INFINITY) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float f = part_m[w] > -INFINITY ? expf(part_m[w] - M) : 0.0f;
                L = fmaf(part_l[w], f, L);
                o = fmaf(part_acc[w][threadIdx.x], f, o);
            }
        }
        out[(size_t)qb * ldo + h * hd + threadIdx.x] = L > 0.0f ? o / L : 0.0f;
    }
}

// decoder input of one step: x[r] = embed[token[r]] + pos[step]
__global__ __launch_bounds__(256) void avsr_embed_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ embed, const float* __restrict__ pos, int step, int d,
                                                         int V, size_t total, float* __restrict__ x) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t r = idx / d;
    const int c = (int)(idx - r * d);
    int tok = tokens[r];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    x[idx] = embed[(size_t)tok * d + c] + pos[(size_t)step * d + c];
}

// self-attention cache of one step: the row's prefix is re-gathered from its source row (beam search: rows change parents), then this
// step's keys / values are appended.  old / neu: [layers][2][R][Lmax][d]; qkv of the CURRENT layer only is appended by the caller's
// per-layer launch (append_kernel); this kernel moves the prefixes of ALL layers at once.  grid (step, R, layers * 2)
__global__ __launch_bounds__(256) void avsr_cache_gather_kernel(const float* __restrict__ old, float* __restrict__ neu, const int32_t* __restrict__ src_rows, int R,
                                                                int Lmax, int d) {
    const int pos = blockIdx.x, r = blockIdx.y, lk = blockIdx.z;
    const int src = src_rows[r];
    const float4* from = reinterpret_cast<const float4*>(old + (((size_t)lk * R + src) * Lmax + pos) * d);
    float4* to = reinterpret_cast<float4*>(neu + (((size_t)lk * R + r) * Lmax + pos) * d);
    for (int c = threadIdx.x; c < d / 4; c += 256) to[c] = from[c];
}

// append this step's k | v (columns [d, 3d) of the layer's qkv rows) to the layer's cache at position `step`
__global__ __launch_bounds__(256) void avsr_cache_append_kernel(const float* __restrict__ qkv, float* __restrict__ kcache, float* __restrict__ vcache, int Lmax, int d,
                                                                int step, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t r = idx / d;
    const int c = (int)(idx - r * d);
    kcache[(r * Lmax + step) * d + c] = qkv[r * 3 * d + d + c];
    vcache[(r * Lmax + step) * d + c] = qkv[r * 3 * d + 2 * d + c];
}

struct AvsrPlan {
    int H1, H2;                      // after the Conv3d (H / 2) and after the max-pool
    size_t off_a0, off_a1, off_col, off_x, off_y, off_z, off_pool, off_apad, off_fused, off_h, off_t, off_qkv, off_big, total;
};

AvsrPlan avsr_plan(const rs_avsr& k, int B, int T) {
    const rs_avsr_dims& d = k.d;
    AvsrPlan p{};
    p.H1 = d.image_size / 2;
    p.H2 = (p.H1 + 1) / 2;
    const size_t N = (size_t)B * T, dm = d.encoder_embed_dim;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += rs_align(bytes); return at; };
    p.off_a0 = take(N * p.H1 * p.H1 * 64 * 4);
    const size_t map = N * p.H2 * p.H2 * 64 * 4;           // the largest trunk map (layer 1); later layers halve the pixels and double the channels
    p.off_x = take(map); p.off_y = take(map); p.off_z = take(map);
    p.off_col = take(N * p.H2 * p.H2 * 64 * 4);           // the down-sampling branch's output (at most a stage-1 map)
    p.off_a1 = take(map / 2);                              // strided rows of a down-sampling convolution
    p.off_pool = take(N * 512 * 4);
    p.off_apad = take(N * pad32(d.audio_feat_dim) * 4);
    p.off_fused = take(N * 2 * dm * 4);
    p.off_h = take(N * dm * 4);
    p.off_t = take(N * dm * 4);
    p.off_qkv = take(N * 3 * dm * 4);
    p.off_big = take(N * (size_t)std::max(d.encoder_ffn_dim, 2 * d.encoder_embed_dim) * 4);
    p.total = o + 256;
    return p;
}

template <typename T>
int avsr_get(rs_ctx* ctx, const std::string& name, size_t elems, const T*& out) {
    auto it = ctx->tensors.find(name);
    if (it == ctx->tensors.end()) return rs_fail(ctx, RS_EMISSING, "weight tensor '%s' was not registered", name.c_str());
    if (it->second.second != elems * sizeof(T))
        return rs_fail(ctx, RS_EINVAL, "tensor '%s': expected %zu bytes, got %zu", name.c_str(), elems * sizeof(T), it->second.second);
    if ((uintptr_t)it->second.first & 15) return rs_fail(ctx, RS_EINVAL, "tensor '%s' is not 16-byte aligned", name.c_str());
    out = reinterpret_cast<const T*>(it->second.first);
    return RS_OK;
}

void avsr_free(rs_avsr* k) { delete k; }

int launch_attn(rs_ctx* ctx, const float* q, int ldq, const float* k, const float* v, int ldk, size_t kstride, const float* kmask, int mask_pitch, int rows_per_kb,
                int Bq, int Tq, int n_keys, int H, int hd, int causal, float* out, int ldo, hipStream_t s) {
    if (Bq <= 0 || Tq <= 0 || n_keys <= 0) return RS_OK;
    if (hd > 256) return rs_fail(ctx, RS_EINVAL, "avsr attention: head_dim %d > 256", hd);
    const dim3 grid((Tq + 3) / 4, H, Bq), block(256);
    const float scaling = 1.0f / sqrtf((float)hd);
    if (Tq > 1 && !causal && (hd == 64 || hd == 128) && ldq % 4 == 0 && ldk % 4 == 0 && ldo % 4 == 0) {       // the encoder: exact-f32 MFMA
        const dim3 grid16((Tq + 63) / 64, H, Bq);
        if (hd == 64)
            hipLaunchKernelGGL((avsr_attn_mfma_kernel<64>), grid16, block, 0, s, q, ldq, k, v, ldk, kstride, kmask, mask_pitch, rows_per_kb, Tq, n_keys, scaling, out, ldo);
        else
            hipLaunchKernelGGL((avsr_attn_mfma_kernel<128>), grid16, block, 0, s, q, ldq, k, v, ldk, kstride, kmask, mask_pitch, rows_per_kb, Tq, n_keys, scaling, out, ldo);
        RS_CHECK_LAUNCH(ctx, "avsr attention (mfma)");
        return RS_OK;
    }
    if (Tq == 1 && n_keys <= 512 && hd % 4 == 0) {         // a decoding step: the keys over the four waves of a workgroup
        if (n_keys <= 256)
            hipLaunchKernelGGL((avsr_attn_step_kernel<1>), dim3(H, Bq), block, 0, s, q, ldq, k, v, ldk, kstride, kmask, mask_pitch, rows_per_kb, n_keys, hd, scaling, out, ldo);
        else
            hipLaunchKernelGGL((avsr_attn_step_kernel<2>), dim3(H, Bq), block, 0, s, q, ldq, k, v, ldk, kstride, kmask, mask_pitch, rows_per_kb, n_keys, hd, scaling, out, ldo);
        RS_CHECK_LAUNCH(ctx, "avsr attention (step)");
        return RS_OK;
    }
    if (!causal && n_keys <= 512 && hd % 4 == 0) {         // lanes over the keys (scores in registers)
        if (n_keys <= 256)
            hipLaunchKernelGGL((avsr_attn_keys_kernel<4>), grid, block, 0, s, q, ldq, k, v, ldk, kstride, kmask, mask_pitch, rows_per_kb, Tq, n_keys, hd, scaling, out, ldo);
        else
            hipLaunchKernelGGL((avsr_attn_keys_kernel<8>), grid, block, 0, s, q, ldq, k, v, ldk, kstride, kmask, mask_pitch, rows_per_kb, Tq, n_keys, hd, scaling, out, ldo);
        RS_CHECK_LAUNCH(ctx, "avsr attention");
        return RS_OK;
    }
#define RS_AV_ATT(NV) hipLaunchKernelGGL((avsr_attn_kernel<NV>), grid, block, 0, s, q, ldq, k, v, ldk, kstride, kmask, mask_pitch, rows_per_kb, Tq, n_keys, hd, scaling, causal, out, ldo)
    if (hd <= 64) RS_AV_ATT(1);
    else if (hd <= 128) RS_AV_ATT(2);
    else RS_AV_ATT(4);
#undef RS_AV_ATT
    RS_CHECK_LAUNCH(ctx, "avsr attention");
    return RS_OK;
}

}  // namespace

extern "C" int rs_avsr_create(rs_ctx** out, int device, const rs_avsr_dims* dims) {
    if (!out || !dims) return RS_EINVAL;
    *out = nullptr;
    rs_ctx* ctx = new (std::nothrow) rs_ctx();
    if (!ctx) return RS_EINVAL;
    *out = ctx;
    ctx->device = device;
    rs_avsr* k = new (std::nothrow) rs_avsr();
    if (!k) return rs_fail(ctx, RS_EINVAL, "out of memory");
    ctx->avsr = k;
    ctx->avsr_free = avsr_free;
    k->d = *dims;
    const rs_avsr_dims& d = k->d;
    if (d.encoder_layers < 1 || d.decoder_layers < 1 || d.encoder_embed_dim % 32 || d.decoder_embed_dim != d.encoder_embed_dim || d.encoder_ffn_dim % 32 ||
        d.decoder_ffn_dim % 32 || d.encoder_heads < 1 || d.decoder_heads < 1 || d.encoder_embed_dim % d.encoder_heads || d.decoder_embed_dim % d.decoder_heads)
        return rs_fail(ctx, RS_EINVAL, "avsr: widths must be multiples of 32 and divisible by the head counts; encoder and decoder share one width");
    if (d.conv_pos < 2 || d.conv_pos % 2 || d.conv_pos_groups < 1 || d.encoder_embed_dim % d.conv_pos_groups)
        return rs_fail(ctx, RS_EINVAL, "avsr: positional convolution (kernel %d, groups %d)", d.conv_pos, d.conv_pos_groups);
    if (d.image_size < 16 || d.image_size > 96 || d.image_size % 8) return rs_fail(ctx, RS_EINVAL, "avsr: image_size %d (16 .. 96, a multiple of 8)", d.image_size);
    if (d.audio_feat_dim < 1 || d.vocab_size < 2 || d.max_positions < 2 || !d.fuse_concat)
        return rs_fail(ctx, RS_EINVAL, "avsr: audio_feat_dim / vocab_size / max_positions; modality_fuse 'concat' is built");
    if (hipSetDevice(device) != hipSuccess) return rs_fail(ctx, RS_EHIP, "hipSetDevice(%d) failed", device);
    return RS_OK;
}

int rs_avsr_finalize_impl(rs_ctx* ctx) {
    rs_avsr& k = *ctx->avsr;
    const rs_avsr_dims& d = k.d;
    int rc;
    const size_t dm = d.encoder_embed_dim, ffn = d.encoder_ffn_dim, dffn = d.decoder_ffn_dim;
#define AV_GET(name, elems, field) do { rc = avsr_get(ctx, name, (size_t)(elems), field); if (rc != RS_OK) return rc; } while (0)
    AV_GET("fe.audio.w", dm * pad32(d.audio_feat_dim), k.audio_w); AV_GET("fe.audio.b", dm, k.audio_b);
    AV_GET("v.conv3d.w", 245 * 64, k.conv3d_w); AV_GET("v.bn0.alpha", 64, k.bn0_a); AV_GET("v.bn0.beta", 64, k.bn0_b); AV_GET("v.prelu0", 64, k.prelu0);
    for (int L = 1; L <= 4; ++L)
        for (int b = 0; b < 2; ++b) {
            rs_avsr_block& B = k.blocks[L - 1][b];
            const std::string p = "v.l" + std::to_string(L) + "." + std::to_string(b) + ".";
            const size_t cin = b == 0 ? TRUNK_C[L - 1] : TRUNK_C[L], c = TRUNK_C[L];
            AV_GET(p + "conv1.w", c * 9 * cin, B.conv1_w); AV_GET(p + "conv2.w", c * 9 * c, B.conv2_w);
            AV_GET(p + "bn1.alpha", c, B.bn1_a); AV_GET(p + "bn1.beta", c, B.bn1_b); AV_GET(p + "bn2.alpha", c, B.bn2_a); AV_GET(p + "bn2.beta", c, B.bn2_b);
            AV_GET(p + "relu1", c, B.relu1); AV_GET(p + "relu2", c, B.relu2);
            B.ds_w = B.ds_a = B.ds_b = nullptr;
            if (b == 0 && L > 1) { AV_GET(p + "ds.w", c * cin, B.ds_w); AV_GET(p + "ds.bn.alpha", c, B.ds_a); AV_GET(p + "ds.bn.beta", c, B.ds_b); }
        }
    AV_GET("v.proj.w", dm * 512, k.vproj_w); AV_GET("v.proj.b", dm, k.vproj_b);
    AV_GET("fuse.ln.g", 2 * dm, k.fuse_g); AV_GET("fuse.ln.b", 2 * dm, k.fuse_b);
    AV_GET("fuse.proj.w", dm * 2 * dm, k.fproj_w); AV_GET("fuse.proj.b", dm, k.fproj_b);
    const size_t cg = dm / d.conv_pos_groups;
    AV_GET("enc.pos.w", (size_t)d.conv_pos_groups * d.conv_pos * cg * cg, k.pos_w); AV_GET("enc.pos.b", dm, k.pos_b);
    AV_GET("enc.ln.g", dm, k.encln_g); AV_GET("enc.ln.b", dm, k.encln_b);
    k.enc.assign(d.encoder_layers, rs_avsr_layer{});
    for (int i = 0; i < d.encoder_layers; ++i) {
        rs_avsr_layer& L = k.enc[i];
        const std::string p = "E" + std::to_string(i) + ".";
        AV_GET(p + "qkv.w", 3 * dm * dm, L.sa.qkv_w); AV_GET(p + "qkv.b", 3 * dm, L.sa.qkv_b); AV_GET(p + "o.w", dm * dm, L.sa.o_w); AV_GET(p + "o.b", dm, L.sa.o_b);
        AV_GET(p + "ln1.g", dm, L.ln1_g); AV_GET(p + "ln1.b", dm, L.ln1_b); AV_GET(p + "ln2.g", dm, L.ln2_g); AV_GET(p + "ln2.b", dm, L.ln2_b);
        AV_GET(p + "ff1.w", ffn * dm, L.ff1_w); AV_GET(p + "ff1.b", ffn, L.ff1_b); AV_GET(p + "ff2.w", dm * ffn, L.ff2_w); AV_GET(p + "ff2.b", dm, L.ff2_b);
    }
    AV_GET("dec.embed", (size_t)d.vocab_size * dm, k.embed); AV_GET("dec.pos", (size_t)d.max_positions * dm, k.dec_pos);
    AV_GET("dec.ln.g", dm, k.decln_g); AV_GET("dec.ln.b", dm, k.decln_b);
    AV_GET("dec.lm.w", (size_t)pad4(d.vocab_size) * dm, k.lm_w);
    k.dec.assign(d.decoder_layers, rs_avsr_layer{});
    for (int i = 0; i < d.decoder_layers; ++i) {
        rs_avsr_layer& L = k.dec[i];
        const std::string p = "D" + std::to_string(i) + ".";
        AV_GET(p + "sa.qkv.w", 3 * dm * dm, L.sa.qkv_w); AV_GET(p + "sa.qkv.b", 3 * dm, L.sa.qkv_b); AV_GET(p + "sa.o.w", dm * dm, L.sa.o_w); AV_GET(p + "sa.o.b", dm, L.sa.o_b);
        AV_GET(p + "ca.q.w", dm * dm, L.ca.q_w); AV_GET(p + "ca.q.b", dm, L.ca.q_b); AV_GET(p + "ca.kv.w", 2 * dm * dm, L.ca.kv_w); AV_GET(p + "ca.kv.b", 2 * dm, L.ca.kv_b);
        AV_GET(p + "ca.o.w", dm * dm, L.ca.o_w); AV_GET(p + "ca.o.b", dm, L.ca.o_b);
        AV_GET(p + "ln1.g", dm, L.ln1_g); AV_GET(p + "ln1.b", dm, L.ln1_b); AV_GET(p + "ln2.g", dm, L.ln2_g); AV_GET(p + "ln2.b", dm, L.ln2_b);
        AV_GET(p + "ln3.g", dm, L.ln3_g); AV_GET(p + "ln3.b", dm, L.ln3_b);
        AV_GET(p + "ff1.w", dffn * dm, L.ff1_w); AV_GET(p + "ff1.b", dffn, L.ff1_b); AV_GET(p + "ff2.w", dm * dffn, L.ff2_w); AV_GET(p + "ff2.b", dm, L.ff2_b);
    }
#undef AV_GET
    ctx->finalized = true;
    return RS_OK;
}

extern "C" int rs_avsr_encoder_set_taps(rs_ctx* ctx, float* video, float* fused_ln, float* enc_ln, float* layer_out, const int32_t* layer_ids, int n_layer_ids) {
    if (!ctx || !ctx->avsr) return RS_EINVAL;
    rs_avsr& k = *ctx->avsr;
    if (n_layer_ids < 0 || (n_layer_ids > 0 && (!layer_out || !layer_ids))) return rs_fail(ctx, RS_EINVAL, "avsr taps: null pointer");
    for (int i = 0; i < n_layer_ids; ++i)
        if (layer_ids[i] < 0 || layer_ids[i] >= k.d.encoder_layers) return rs_fail(ctx, RS_EINVAL, "avsr taps: layer %d out of range", layer_ids[i]);
    k.tap_video = video; k.tap_fused = fused_ln; k.tap_encln = enc_ln;
    k.tap_layers = n_layer_ids > 0 ? layer_out : nullptr;
    k.tap_ids.assign(layer_ids, layer_ids + n_layer_ids);
    return RS_OK;
}

extern "C" size_t rs_avsr_workspace_bytes(const rs_ctx* ctx, int B, int T) {
    if (!ctx || !ctx->avsr || B <= 0 || T <= 0) return 0;
    return avsr_plan(*ctx->avsr, B, T).total;
}

extern "C" int rs_avsr_encoder_forward(rs_ctx* ctx, const float* input_values, const float* pixel_values, const float* padding_mask, int B, int T, float* enc_out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx || !ctx->avsr) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_avsr_encoder_forward");
    if (B <= 0 || T <= 0) return B < 0 || T < 0 ? rs_fail(ctx, RS_EINVAL, "avsr encoder: negative size") : RS_OK;
    if ((!input_values && !pixel_values) || !padding_mask || !enc_out || !workspace)
        return rs_fail(ctx, RS_EINVAL, "avsr encoder: null pointer (one of input_values / pixel_values may be NULL: that modality's FEATURES are zeros, modeling_avhubert.py:172-177)");
    rs_avsr& k = *ctx->avsr;
    const rs_avsr_dims& d = k.d;
    hipStream_t s = (hipStream_t)stream;
    const AvsrPlan pl = avsr_plan(k, B, T);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "avsr encoder: workspace %zu < %zu", workspace_bytes, pl.total);
    if (T > 65535 || B > 65535) return rs_fail(ctx, RS_EINVAL, "avsr encoder: more than 65535 frames / clips per call");
    char* ws = reinterpret_cast<char*>(workspace);
    auto fp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    float *a0 = fp(pl.off_a0), *x = fp(pl.off_x), *y = fp(pl.off_y), *z = fp(pl.off_z), *col = fp(pl.off_col), *a1 = fp(pl.off_a1), *pool = fp(pl.off_pool);
    float *apad = fp(pl.off_apad), *fused = fp(pl.off_fused), *h = fp(pl.off_h), *t = fp(pl.off_t), *qkv = fp(pl.off_qkv), *big = fp(pl.off_big);
    const int dm = d.encoder_embed_dim, ffn = d.encoder_ffn_dim, H = d.image_size;
    const size_t N = (size_t)B * T;
    const int M = (int)N;
    int rc;
#define RS_TRY(call) do { rc = (call); if (rc != RS_OK) return rc; } while (0)
    auto gemm = [&](const float* A, int lda, const float* W, int K, float* out, int ldc, long long rows, int Nc, int flags, const float* bias, const float* res) -> int {
        if (rows > 0x7fffffffLL) return rs_fail(ctx, RS_EINVAL, "avsr: %lld GEMM rows", rows);
        return rs_launch_gemm_f32(ctx, A, lda, W, K, out, ldc, (int)rows, Nc, K, flags, bias, 1.0f, res, nullptr, 0, 0, s);
    };
    auto blocks1d = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    if (pixel_values) {
        // ---- video: Conv3d + BN + PReLU, max-pool, ResNet-18 trunk, average pool, projection
        rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, 2.0 * N * pl.H1 * pl.H1 * 64 * 245.0, 0.0);
        hipLaunchKernelGGL(avsr_conv3d_kernel, dim3(pl.H1, T, B), dim3(256), 0, s, pixel_values, T, H, H, k.conv3d_w, k.bn0_a, k.bn0_b, k.prelu0, a0);
        {
            const size_t total = N * pl.H2 * pl.H2 * 64;
            hipLaunchKernelGGL(avsr_maxpool_kernel, blocks1d(total), dim3(256), 0, s, a0, pl.H1, pl.H1, 64, pl.H2, pl.H2, total, x);
        }
        rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
        RS_CHECK_LAUNCH(ctx, "avsr video front-end");
        int hw = pl.H2;
        float *cur = x, *o1 = y, *o2 = z;
        for (int L = 1; L <= 4; ++L)
            for (int b = 0; b < 2; ++b) {
                const rs_avsr_block& Bk = k.blocks[L - 1][b];
                const int cin = b == 0 ? TRUNK_C[L - 1] : TRUNK_C[L], c = TRUNK_C[L];
                const int stride = (b == 0 && L > 1) ? 2 : 1;
                const int ohw = stride == 2 ? (hw - 1) / 2 + 1 : hw;            // 3 x 3, padding 1
                const size_t rows_out = N * ohw * ohw;
                // conv1 -> bn1 -> relu1 and conv2 -> bn2 -> (+ residual) -> relu2, each ONE launch: the 3 x 3 patches are gathered by the
                // GEMM's loader, BatchNorm / residual / PReLU run in its epilogue (rs_launch_conv3x3_f32; until round 6 the patch
                // matrices went through HBM — 4.5 GB per convolution of the first stage — followed by a BatchNorm pass)
                RS_TRY(rs_launch_conv3x3_f32(ctx, cur, (int)N, hw, hw, cin, ohw, ohw, stride, Bk.conv1_w, c, Bk.bn1_a, Bk.bn1_b, nullptr, Bk.relu1, o1, 0, s));
                const float* res = cur;
                if (Bk.ds_w) {                                                   // 1 x 1 convolution with the block's stride + BatchNorm on the block input
                    const float* src = cur;
                    if (stride == 2) {
                        hipLaunchKernelGGL(avsr_stride2_kernel, blocks1d(rows_out * cin / 4), dim3(256), 0, s, cur, hw, hw, cin, ohw, ohw, rows_out * cin / 4, a1);
                        src = a1;
                    }
                    RS_TRY(rs_launch_conv3x3_f32(ctx, src, (int)N, ohw, ohw, cin, ohw, ohw, 1, Bk.ds_w, c, Bk.ds_a, Bk.ds_b, nullptr, nullptr, col, 1, s));
                    res = col;
                }
                RS_TRY(rs_launch_conv3x3_f32(ctx, o1, (int)N, ohw, ohw, c, ohw, ohw, 1, Bk.conv2_w, c, Bk.bn2_a, Bk.bn2_b, res, Bk.relu2, o2, 0, s));
                RS_CHECK_LAUNCH(ctx, "avsr ResNet block");
                float* nxt = o2;                      // rotate: the block's output becomes the input, the old input and o1 are scratch
                o2 = cur; cur = nxt;
                hw = ohw;
            }
        hipLaunchKernelGGL(avsr_avgpool_kernel, blocks1d(N * 512), dim3(256), 0, s, cur, hw * hw, 512, N * 512, pool);
    }
    // audio and video projections straight into the two halves of the fused rows [M][2 d]; a missing modality contributes ZERO
    // features (after where its projection would be: modeling_avhubert.py:172-177 torch.zeros_like of the other extractor's output)
    if (!input_values || !pixel_values) RS_HIP(ctx, hipMemsetAsync(fused, 0, N * 2 * dm * 4, s));
    const int Ka = pad32(d.audio_feat_dim);
    if (input_values) {
        hipLaunchKernelGGL(avsr_padcols_kernel, blocks1d(N * Ka), dim3(256), 0, s, input_values, d.audio_feat_dim, Ka, N * Ka, apad);
        RS_TRY(gemm(apad, Ka, k.audio_w, Ka, fused, 2 * dm, M, dm, RS_GEMM_BIAS, k.audio_b, nullptr));
    }
    if (pixel_values) RS_TRY(gemm(pool, 512, k.vproj_w, 512, fused + dm, 2 * dm, M, dm, RS_GEMM_BIAS, k.vproj_b, nullptr));
    if (k.tap_video) RS_HIP(ctx, hipMemcpy2DAsync(k.tap_video, (size_t)dm * 4, fused + dm, (size_t)2 * dm * 4, (size_t)dm * 4, N, hipMemcpyDeviceToDevice, s));
    // fusion LayerNorm, post_extract_proj, padded frames zeroed, positional convolution, encoder LayerNorm
    hipLaunchKernelGGL(avsr_layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, fused, k.fuse_g, k.fuse_b, M, 2 * dm, 1e-5f, fused);
    if (k.tap_fused) RS_HIP(ctx, hipMemcpyAsync(k.tap_fused, fused, N * 2 * dm * 4, hipMemcpyDeviceToDevice, s));
    RS_TRY(gemm(fused, 2 * dm, k.fproj_w, 2 * dm, h, dm, M, dm, RS_GEMM_BIAS, k.fproj_b, nullptr));
    hipLaunchKernelGGL(avsr_mask_rows_kernel, blocks1d(N * dm), dim3(256), 0, s, h, padding_mask, dm, N * dm);
    {
        const int cg = dm / d.conv_pos_groups;
        if (cg % 16 == 0 && cg <= 64 && !getenv("RS_AVSR_POSCONV_OLD")) {
            const size_t lds = (size_t)(64 + d.conv_pos) * (cg + 1) * 4;
            const dim3 grid((T + 63) / 64, d.conv_pos_groups, B);
#define RS_AV_POS(NT)                                                                                                                         \
            do {                                                                                                                              \
                if (lds > 64 * 1024) RS_TRY(rs_ensure_dynamic_lds(ctx, (const void*)avsr_posconv_mfma_kernel<NT>, (int)lds));                  \
                hipLaunchKernelGGL((avsr_posconv_mfma_kernel<NT>), grid, dim3(256), lds, s, h, T, dm, d.conv_pos, k.pos_w, k.pos_b, t);          \
            } while (0)
            if (cg == 16) RS_AV_POS(1); else if (cg == 32) RS_AV_POS(2); else if (cg == 48) RS_AV_POS(3); else RS_AV_POS(4);
#undef RS_AV_POS
        } else {
            const size_t lds = (size_t)(8 + d.conv_pos) * cg * 4;
            if (lds > 64 * 1024) RS_TRY(rs_ensure_dynamic_lds(ctx, (const void*)avsr_posconv_kernel, (int)lds));
            hipLaunchKernelGGL(avsr_posconv_kernel, dim3((T + 7) / 8, d.conv_pos_groups, B), dim3(256), lds, s, h, T, dm, cg, d.conv_pos, k.pos_w, k.pos_b, t);
        }
    }
    hipLaunchKernelGGL(avsr_layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, t, k.encln_g, k.encln_b, M, dm, d.layer_norm_eps, h);
    if (k.tap_encln) RS_HIP(ctx, hipMemcpyAsync(k.tap_encln, h, N * dm * 4, hipMemcpyDeviceToDevice, s));
    RS_CHECK_LAUNCH(ctx, "avsr fusion");
    // ---- HuBERT encoder layers (post-LayerNorm): x = LN(x + attn(x)); x = LN(x + ffn(x))
    const int hd = dm / d.encoder_heads;
    for (int i = 0; i < d.encoder_layers; ++i) {
        const rs_avsr_layer& L = k.enc[i];
        RS_TRY(gemm(h, dm, L.sa.qkv_w, dm, qkv, 3 * dm, M, 3 * dm, RS_GEMM_BIAS, L.sa.qkv_b, nullptr));
        RS_TRY(launch_attn(ctx, qkv, 3 * dm, qkv + dm, qkv + 2 * dm, 3 * dm, (size_t)T * 3 * dm, padding_mask, T, 1, B, T, T, d.encoder_heads, hd, 0, t, dm, s));
        RS_TRY(gemm(t, dm, L.sa.o_w, dm, big, dm, M, dm, RS_GEMM_BIAS | RS_GEMM_RESIDUAL, L.sa.o_b, h));
        hipLaunchKernelGGL(avsr_layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, big, L.ln1_g, L.ln1_b, M, dm, d.layer_norm_eps, h);
        RS_TRY(gemm(h, dm, L.ff1_w, dm, big, ffn, M, ffn, RS_GEMM_BIAS | RS_GEMM_GELU, L.ff1_b, nullptr));
        RS_TRY(gemm(big, ffn, L.ff2_w, ffn, t, dm, M, dm, RS_GEMM_BIAS | RS_GEMM_RESIDUAL, L.ff2_b, h));
        hipLaunchKernelGGL(avsr_layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, t, L.ln2_g, L.ln2_b, M, dm, d.layer_norm_eps, h);
        for (size_t q = 0; q < k.tap_ids.size(); ++q)
            if (k.tap_ids[q] == i) RS_HIP(ctx, hipMemcpyAsync(k.tap_layers + q * N * dm, h, N * dm * 4, hipMemcpyDeviceToDevice, s));
        RS_CHECK_LAUNCH(ctx, "avsr encoder layer");
    }
    RS_HIP(ctx, hipMemcpyAsync(enc_out, h, N * dm * 4, hipMemcpyDeviceToDevice, s));
#undef RS_TRY
    return RS_OK;
}

// ---- decoder ----------------------------------------------------------------------------------------------------------------------------
namespace {
struct AvsrDecPlan {
    size_t off_cross, off_cache[2], off_x, off_t, off_u, off_qkv, off_big, off_src, total;
};
AvsrDecPlan avsr_dec_plan(const rs_avsr& k, int B, int T, int beams, int max_len) {
    const rs_avsr_dims& d = k.d;
    const size_t dm = d.decoder_embed_dim, R = (size_t)B * beams, Ld = d.decoder_layers;
    AvsrDecPlan p{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += rs_align(bytes); return at; };
    p.off_cross = take(Ld * (size_t)B * T * 2 * dm * 4);
    p.off_cache[0] = take(Ld * 2 * R * max_len * dm * 4);
    p.off_cache[1] = take(Ld * 2 * R * max_len * dm * 4);
    p.off_x = take(R * dm * 4); p.off_t = take(R * dm * 4); p.off_u = take(R * dm * 4);
    p.off_qkv = take(R * 3 * dm * 4);
    p.off_big = take(R * (size_t)std::max(d.decoder_ffn_dim, d.decoder_embed_dim) * 4);
    p.off_src = take(R * 4);
    p.total = o + 256;
    return p;
}
}  // namespace

extern "C" size_t rs_avsr_decoder_state_bytes(const rs_ctx* ctx, int B, int T, int beams, int max_len) {
    if (!ctx || !ctx->avsr || B <= 0 || T <= 0 || beams <= 0 || max_len <= 0) return 0;
    return avsr_dec_plan(*ctx->avsr, B, T, beams, max_len).total;
}

extern "C" int rs_avsr_decoder_begin(rs_ctx* ctx, const float* enc, int B, int T, int beams, int max_len, void* state, size_t state_bytes, void* stream) {
    if (!ctx || !ctx->avsr) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_avsr_decoder_begin");
    if (B <= 0 || T <= 0 || beams <= 0 || max_len <= 0 || !enc || !state) return rs_fail(ctx, RS_EINVAL, "avsr decoder: bad argument");
    rs_avsr& k = *ctx->avsr;
    const rs_avsr_dims& d = k.d;
    if (max_len > d.max_positions) return rs_fail(ctx, RS_EINVAL, "avsr decoder: %d positions exceed max_target_positions %d", max_len, d.max_positions);
    const AvsrDecPlan pl = avsr_dec_plan(k, B, T, beams, max_len);
    if (state_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "avsr decoder: state %zu < %zu", state_bytes, pl.total);
    hipStream_t s = (hipStream_t)stream;
    const int dm = d.decoder_embed_dim;
    float* cross = reinterpret_cast<float*>(reinterpret_cast<char*>(state) + pl.off_cross);
    for (int i = 0; i < d.decoder_layers; ++i) {          // cross-attention keys | values of every layer, once per utterance batch
        const rs_avsr_layer& L = k.dec[i];
        const int rc = rs_launch_gemm_f32(ctx, enc, dm, L.ca.kv_w, dm, cross + (size_t)i * B * T * 2 * dm, 2 * dm, B * T, 2 * dm, dm, RS_GEMM_BIAS, L.ca.kv_b, 1.0f, nullptr,
                                          nullptr, 0, 0, s);
        if (rc != RS_OK) return rc;
    }
    return RS_OK;
}

extern "C" int rs_avsr_decoder_step(rs_ctx* ctx, const int32_t* tokens, const int32_t* src_rows, int step, const float* padding_mask, int B, int T, int beams,
                                    int max_len, float* logits, void* state, size_t state_bytes, void* stream) {
    if (!ctx || !ctx->avsr) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_avsr_decoder_step");
    if (B <= 0 || T <= 0 || beams <= 0 || max_len <= 0 || step < 0 || step >= max_len || !tokens || !padding_mask || !logits || !state)
        return rs_fail(ctx, RS_EINVAL, "avsr decoder step: bad argument (step %d of %d)", step, max_len);
    rs_avsr& k = *ctx->avsr;
    const rs_avsr_dims& d = k.d;
    const AvsrDecPlan pl = avsr_dec_plan(k, B, T, beams, max_len);
    if (state_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "avsr decoder: state %zu < %zu", state_bytes, pl.total);
    hipStream_t s = (hipStream_t)stream;
    char* st = reinterpret_cast<char*>(state);
    auto fp = [&](size_t off) { return reinterpret_cast<float*>(st + off); };
    const int dm = d.decoder_embed_dim, R = B * beams, ffn = d.decoder_ffn_dim, Ld = d.decoder_layers, hd = dm / d.decoder_heads, Vp = pad4(d.vocab_size);
    // Self-attention caches: without re-parenting (greedy search: src_rows == NULL at every step) everything lives in buffer 0.  With
    // re-parenting (beam search: src_rows given at every step >= 1) step s writes buffer s & 1: the prefixes [0, s) are gathered from
    // the other buffer by source row first, then this step's keys / values are appended.
    const bool reorder = src_rows != nullptr;
    float* cache_cur = fp(pl.off_cache[reorder ? (step & 1) : 0]);
    if (reorder && step > 0)
        hipLaunchKernelGGL(avsr_cache_gather_kernel, dim3(step, R, Ld * 2), dim3(256), 0, s, fp(pl.off_cache[(step & 1) ^ 1]), cache_cur, src_rows, R, max_len, dm);
    float *x = fp(pl.off_x), *t = fp(pl.off_t), *u = fp(pl.off_u), *qkv = fp(pl.off_qkv), *big = fp(pl.off_big), *cross = fp(pl.off_cross);
    int rc;
#define RS_TRY(call) do { rc = (call); if (rc != RS_OK) return rc; } while (0)
    // up to 128 hypothesis rows: the few-rows form (N / 16 workgroups, K cut over the waves); more rows: the tiled kernel
    auto gemm = [&](const float* A, int lda, const float* W, int K, float* out, int ldc, int Nc, int flags, const float* bias, const float* res) -> int {
        if (R <= 128) return rs_launch_gemm_f32_skinny(ctx, A, lda, W, K, out, ldc, R, Nc, K, flags, bias, res, s);
        return rs_launch_gemm_f32(ctx, A, lda, W, K, out, ldc, R, Nc, K, flags, bias, 1.0f, res, nullptr, 0, 0, s);
    };
    auto ln = [&](const float* in, const float* g, const float* b, float* out) {
        hipLaunchKernelGGL(avsr_layernorm_kernel, dim3((R + 3) / 4), dim3(256), 0, s, in, g, b, R, dm, d.layer_norm_eps, out);
    };
    hipLaunchKernelGGL(avsr_embed_kernel, dim3((unsigned)(((size_t)R * dm + 255) / 256)), dim3(256), 0, s, tokens, k.embed, k.dec_pos, step, dm, d.vocab_size, (size_t)R * dm, x);
    for (int i = 0; i < Ld; ++i) {
        const rs_avsr_layer& L = k.dec[i];
        float* kc = cache_cur + ((size_t)(2 * i) * R) * max_len * dm;
        float* vc = cache_cur + ((size_t)(2 * i + 1) * R) * max_len * dm;
        RS_TRY(gemm(x, dm, L.sa.qkv_w, dm, qkv, 3 * dm, 3 * dm, RS_GEMM_BIAS, L.sa.qkv_b, nullptr));
        hipLaunchKernelGGL(avsr_cache_append_kernel, dim3((unsigned)(((size_t)R * dm + 255) / 256)), dim3(256), 0, s, qkv, kc, vc, max_len, dm, step, (size_t)R * dm);
        RS_TRY(launch_attn(ctx, qkv, 3 * dm, kc, vc, dm, (size_t)max_len * dm, nullptr, 0, 1, R, 1, step + 1, d.decoder_heads, hd, 0, t, dm, s));
        RS_TRY(gemm(t, dm, L.sa.o_w, dm, u, dm, dm, RS_GEMM_BIAS | RS_GEMM_RESIDUAL, L.sa.o_b, x));
        ln(u, L.ln1_g, L.ln1_b, x);
        RS_TRY(gemm(x, dm, L.ca.q_w, dm, qkv, dm, dm, RS_GEMM_BIAS, L.ca.q_b, nullptr));
        const float* ck = cross + (size_t)i * B * T * 2 * dm;
        RS_TRY(launch_attn(ctx, qkv, dm, ck, ck + dm, 2 * dm, (size_t)T * 2 * dm, padding_mask, T, beams, R, 1, T, d.decoder_heads, hd, 0, t, dm, s));
        RS_TRY(gemm(t, dm, L.ca.o_w, dm, u, dm, dm, RS_GEMM_BIAS | RS_GEMM_RESIDUAL, L.ca.o_b, x));
        ln(u, L.ln2_g, L.ln2_b, x);
        RS_TRY(gemm(x, dm, L.ff1_w, dm, big, ffn, ffn, RS_GEMM_BIAS | RS_GEMM_GELU, L.ff1_b, nullptr));
        RS_TRY(gemm(big, ffn, L.ff2_w, ffn, u, dm, dm, RS_GEMM_BIAS | RS_GEMM_RESIDUAL, L.ff2_b, x));
        ln(u, L.ln3_g, L.ln3_b, x);
    }
    ln(x, k.decln_g, k.decln_b, t);
    RS_TRY(gemm(t, dm, k.lm_w, dm, logits, Vp, Vp, 0, nullptr, nullptr));
    RS_CHECK_LAUNCH(ctx, "avsr decoder step");
#undef RS_TRY
    return RS_OK;
}

// k_frontend.hip — log-mel front-end.
//
// [UPSTREAM] AudioToMelSpectrogramPreprocessor / FilterbankFeatures.forward:
//   pre-emphasis 0.97 -> STFT(n_fft 512, hann 400 centred in the frame, hop 160, center=True with
//   zero edge padding) -> |.|^2 -> Slaney mel (80 banded filters) -> log(x + 2^-24)
//   -> per-utterance, per-feature (mean, unbiased std) normalisation over the valid frames
//   -> zero the padding frames.
// pad_audio (pkg/nemo-asr/src/audio.py:70-83) is folded into the sample fetch: the kernel reads
// raw[i - pad_left] and everything outside [0, len) is 0.0, which is what np.pad would have stored.
//
// HBM-bound by construction: the audio is read once (each sample is touched by ~3 overlapping
// frames, served by L2), the 257-bin spectrum lives only in LDS, and 80 floats per frame go out.
// The 512-point FFT is a radix-2 in-LDS transform, one frame per wave; this is integer/float
// streaming work, deliberately NOT reshaped into a DFT GEMM.
#include "rs_common.h"

namespace {

constexpr int NFFT = 512;
constexpr int NBIN = 257;
constexpr int FB_MAXW = 32;          // widest Slaney filter has 18 taps at 80 mels / 512 fft
constexpr int WAVES = 4;
constexpr int FRAMES_PER_WAVE = 4;   // frames per block = 16

struct FrontParams {
    const float* audio; const int32_t* lens; float* raw; int32_t* n_frames;
    const float* window;   // [win_length]
    const float* twiddle;  // [256][2] cos, -sin of 2*pi*j/512
    const int32_t* fb_idx; // [n_mels][2] first bin, tap count
    const float* fb_w;     // [n_mels][FB_MAXW]
    int audio_stride, pad_left, pad_right, t_max, n_mels, win_length, hop;
    float preemph, log_guard;
};

__device__ __forceinline__ float fetch_sample(const float* __restrict__ a, int i, int pad_left, int len) {
    const int j = i - pad_left;
    return (j >= 0 && j < len) ? a[j] : 0.0f;
}

__device__ __forceinline__ int bitrev9(int v) { return (int)(__brev((unsigned)v) >> 23); }

// One wave transforms TWO frames per 512-point complex FFT (frame t in the real part, t+1 in the
// imaginary part; the two real spectra are separated afterwards: X_a[k] = (Z[k] + conj Z[N-k]) / 2,
// X_b[k] = (Z[k] - conj Z[N-k]) / 2i).  Everything a wave touches in LDS is its own, so the stages are
// separated by wave-level fences, not workgroup barriers; the twiddles are staged once per workgroup.
__global__ __launch_bounds__(64 * WAVES) void logmel_kernel(FrontParams p) {
    __shared__ float2 buf[WAVES][NFFT];
    __shared__ float pw[WAVES][2][NBIN + 3];
    __shared__ float2 tw[NFFT / 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int len = p.lens[b];
    const int Lp = len + p.pad_left + p.pad_right;       // padded length (reference: after pad_audio)
    const int n_valid = Lp / p.hop;                       // floor((Lp + 2*(n_fft/2) - n_fft) / hop)
    if (blockIdx.x == 0 && threadIdx.x == 0) p.n_frames[b] = n_valid;
    const float* a = p.audio + (size_t)b * p.audio_stride;
    if ((int)blockIdx.x * WAVES * FRAMES_PER_WAVE >= min(n_valid, p.t_max)) return;  // block-uniform
    tw[threadIdx.x] = reinterpret_cast<const float2*>(p.twiddle)[threadIdx.x];      // 256 threads, 256 twiddles
    __syncthreads();
    const int frame_base = (blockIdx.x * WAVES + wave) * FRAMES_PER_WAVE;
    const int centre_off = (NFFT - p.win_length) / 2;     // 56: window centred in the 512 frame
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // windowed, pre-emphasised sample n of frame t (0 outside the frame / the padded signal)
    auto sample = [&](int t, int n) -> float {
        if (t >= n_valid || t >= p.t_max || n >= p.win_length) return 0.0f;
        const int i = t * p.hop - NFFT / 2 + centre_off + n;  // index in the padded signal
        if (i < 0 || i >= Lp) return 0.0f;
        const float x0 = fetch_sample(a, i, p.pad_left, len);
        const float x1 = (i >= 1) ? fetch_sample(a, i - 1, p.pad_left, len) : 0.0f;
        const float y = (i >= 1) ? x0 - p.preemph * x1 : x0;
        return y * p.window[n];
    };

    for (int fi = 0; fi < FRAMES_PER_WAVE; fi += 2) {
        const int t = frame_base + fi;
        if (t >= n_valid || t >= p.t_max) break;          // wave-uniform
        // ---- two frames -> bit-reversed LDS order.  A circular shift of the FFT input only changes the
        // phase, so the 400 samples go to slots 0..399 directly.
        float2* z = buf[wave];
        // (tried this round: walking the LDS slots in order and GATHERING the sample of the bit-reversed index from global
        // memory instead — the scattered store below puts a 32-lane store group on two banks — made the kernel slower,
        // 663 -> 963 us: 64 scattered 4-byte loads per instruction cost more than the 16-way conflict.  profiles/r03c_kernel_stats.txt)
#pragma unroll
        for (int q = 0; q < NFFT / 64; ++q) {
            const int n = q * 64 + lane;
            z[bitrev9(n)] = make_float2(sample(t, n), sample(t + 1, n));
        }
        wave_sync();
        // ---- 9 radix-2 DIT stages, 256 butterflies each (4 per lane)
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int half = 1 << s;
            float2 u[4], v[4], w[4];
            int i0[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tb = q * 64 + lane;
                const int pos = tb & (half - 1);
                i0[q] = ((tb >> s) << (s + 1)) + pos;
                w[q] = tw[pos << (8 - s)];
                u[q] = z[i0[q]];
                v[q] = z[i0[q] + half];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float tr = v[q].x * w[q].x - v[q].y * w[q].y;
                const float ti = v[q].x * w[q].y + v[q].y * w[q].x;
                z[i0[q]] = make_float2(u[q].x + tr, u[q].y + ti);
                z[i0[q] + half] = make_float2(u[q].x - tr, u[q].y - ti);
            }
            wave_sync();
        }
        // ---- split the two spectra, power of bins 0..256
        for (int k = lane; k < NBIN; k += 64) {
            const float2 zk = z[k], zn = z[(NFFT - k) & (NFFT - 1)];
            const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);
            const float br = 0.5f * (zk.y + zn.y), bi = 0.5f * (zn.x - zk.x);
            pw[wave][0][k] = ar * ar + ai * ai;
            pw[wave][1][k] = br * br + bi * bi;
        }
        wave_sync();
        // ---- banded mel filterbank + log: lane -> (frame of the pair, mel bin)
        for (int idx = lane; idx < 2 * p.n_mels; idx += 64) {
            const int f = idx >= p.n_mels, m = idx - f * p.n_mels;
            if (t + f < n_valid && t + f < p.t_max) {
                const int k0 = p.fb_idx[2 * m], cnt = p.fb_idx[2 * m + 1];
                const float* w = p.fb_w + m * FB_MAXW;
                float acc = 0.0f;
                for (int j = 0; j < cnt; ++j) acc = fmaf(w[j], pw[wave][f][k0 + j], acc);
                p.raw[((size_t)b * p.t_max + t + f) * p.n_mels + m] = logf(acc + p.log_guard);
            }
        }
        wave_sync();
    }
}

// per utterance: mean / unbiased std over the valid frames, normalise, zero the padding.
// One workgroup of 1000 (+24 idle) threads per utterance; the [t][n_mels] block is walked as float4
// columns: thread i owns mel quad i % Q and rows i / Q + k * (1000 / Q), so its partial sums stay in
// registers and every pass is ~22 independent 16-byte loads per thread instead of 550 dependent
// 4-byte ones (the previous 256-thread version: 480 us, latency-bound).  Statistics are two-pass
// (mean, then sum of squared deviations) like torch's; passes 2 and 3 re-read the block from L2.
__global__ __launch_bounds__(1024) void feat_normalize_kernel(const float* __restrict__ raw,
                                                              const int32_t* __restrict__ n_frames, int t_max,
                                                              int n_mels, float eps, float* __restrict__ out) {
    __shared__ float4 red[1024];
    __shared__ float4 mean_s[32], rstd_s[32];
    const int b = blockIdx.x;
    const int n = min(n_frames[b], t_max);
    const int Q = n_mels >> 2;                       // float4 per row (n_mels % 4 == 0, Q <= 32)
    const int RG = 1000 / Q;                         // row groups walked in parallel
    const int tid = threadIdx.x;
    const int q = tid % Q, rg = tid / Q;
    const bool act = rg < RG;
    const float4* r4 = reinterpret_cast<const float4*>(raw + (size_t)b * t_max * n_mels);
    auto reduce_rows = [&](float4 v) -> float4 {     // sum over row groups; result valid in threads rg == 0
        red[tid] = act ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < Q) {
            for (int g = 0; g < RG; ++g) {
                const float4 x = red[g * Q + tid];
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
        }
        return acc;
    };
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act)
#pragma unroll 8
        for (int t = rg; t < n; t += RG) {
            const float4 x = r4[(size_t)t * Q + q];
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
    s = reduce_rows(s);
    if (tid < Q) {
        const float inv = 1.0f / (float)n;
        mean_s[tid] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    }
    __syncthreads();
    const float4 mean = mean_s[q];
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act)
#pragma unroll 8
        for (int t = rg; t < n; t += RG) {
            const float4 x = r4[(size_t)t * Q + q];
            const float a = x.x - mean.x, bb = x.y - mean.y, c = x.z - mean.z, d = x.w - mean.w;
            m2.x += a * a; m2.y += bb * bb; m2.z += c * c; m2.w += d * d;
        }
    m2 = reduce_rows(m2);
    if (tid < Q) {
        // n == 1: the unbiased variance is 0/0 (NaN in the reference stack too); emit zeros instead of
        // poisoning the encoder and the argmax
        const float dn = (float)(n - 1);
        rstd_s[tid] = n > 1 ? make_float4(1.0f / (sqrtf(m2.x / dn) + eps), 1.0f / (sqrtf(m2.y / dn) + eps),
                                          1.0f / (sqrtf(m2.z / dn) + eps), 1.0f / (sqrtf(m2.w / dn) + eps))
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float4 rstd = rstd_s[q];
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)b * t_max * n_mels);
    if (act)
#pragma unroll 8
        for (int t = rg; t < t_max; t += RG) {
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < n) {
                const float4 x = r4[(size_t)t * Q + q];
                y = make_float4((x.x - mean.x) * rstd.x, (x.y - mean.y) * rstd.y, (x.z - mean.z) * rstd.z,
                                (x.w - mean.w) * rstd.w);
            }
            o4[(size_t)t * Q + q] = y;
        }
}

}  // namespace

int rs_launch_frontend(rs_ctx* ctx, const float* audio, const int32_t* lens, int B, int audio_stride,
                       int pad_left, int pad_right, int t_max, float* feats, int32_t* n_frames, float* raw,
                       hipStream_t s) {
    const rs_dims& d = ctx->d;
    if (B <= 0 || t_max <= 0) return RS_OK;
    if (d.n_fft != NFFT) return rs_fail(ctx, RS_EINVAL, "frontend: only n_fft=512 is built");
    if (d.n_mels > 128 || d.n_mels % 4 || d.win_length > NFFT)
        return rs_fail(ctx, RS_EINVAL, "frontend: n_mels must be a multiple of 4 up to 128, win<=512");
    FrontParams p;
    p.audio = audio; p.lens = lens; p.raw = raw; p.n_frames = n_frames;
    p.window = ctx->fe_window; p.twiddle = ctx->fe_twiddle; p.fb_idx = ctx->fe_fb_idx; p.fb_w = ctx->fe_fb_w;
    p.audio_stride = audio_stride; p.pad_left = pad_left; p.pad_right = pad_right; p.t_max = t_max;
    p.n_mels = d.n_mels; p.win_length = d.win_length; p.hop = d.hop_length;
    p.preemph = d.preemph; p.log_guard = d.log_guard;
    const int fpb = WAVES * FRAMES_PER_WAVE;
    const dim3 grid((t_max + fpb - 1) / fpb, B), block(64 * WAVES);
    const double bytes = (double)B * ((double)t_max * d.hop_length * 4.0 + (double)t_max * d.n_mels * 4.0 * 3.0);
    rs_prof_begin(ctx, RS_PROF_FRONTEND, s, (double)B * t_max * (5.0 * 512 * 9 + 3 * 257 + 2 * 600), bytes);
    hipLaunchKernelGGL(logmel_kernel, grid, block, 0, s, p);
    hipLaunchKernelGGL(feat_normalize_kernel, dim3(B), dim3(1024), 0, s, raw, n_frames, t_max, d.n_mels, d.norm_eps,
                       feats);
    rs_prof_end(ctx, RS_PROF_FRONTEND, s);
    RS_CHECK_LAUNCH(ctx, "frontend");
    return RS_OK;
}

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
// The 512-point FFT is an 8 x 8 x 8 register transform with two LDS transposes, two frames per wave; this is
// integer/float streaming work, deliberately NOT reshaped into a DFT GEMM.
#include "rs_common.h"

namespace {

constexpr int NFFT = 512;
constexpr int NBIN = 257;
constexpr int FB_MAXW = 32;          // widest Slaney filter has 18 taps at 80 mels / 512 fft
constexpr int WAVES = 4;
constexpr int FRAMES_PER_WAVE = 4;   // frames per block = 16
constexpr int MAX_HOP = 256;         // the workgroup's span of the signal is staged in LDS: (frames per block - 1) * hop + n_fft samples
constexpr int SPAN_MAX = (WAVES * FRAMES_PER_WAVE - 1) * MAX_HOP + NFFT;

struct FrontParams {
    const float* audio; const int32_t* lens; float* raw; int32_t* n_frames;
    const float* window;   // [win_length]
    const float* twiddle;  // [256][2] cos, -sin of 2*pi*j/512
    const int32_t* fb_idx; // [n_mels][2] first bin, tap count
    const float* fb_w;     // [n_mels][FB_MAXW]
    int audio_stride, pad_left, pad_right, t_max, n_mels, win_length, hop;
    float preemph, log_guard;
    int kind;              // 0: NeMo (zero edge padding, floor(L / hop) frames, log(x + guard)); 1: ESPnet DefaultFrontend
                           // (reflect edge padding, 1 + floor(L / hop) frames, log(max(x, guard)); normalised by feat_mvn_kernel);
                           // 2: kaldi-native-fbank as sherpa-onnx configures it (reazonspeech.k2.asr): snip_edges = false — frame f
                           // covers samples [hop f + hop / 2 - win / 2, .. + win), edges REFLECTED (-1 -> 0, L -> L - 1) —, per frame:
                           // subtract the mean, x[i] -= preemph x[i - 1] (x[0] -= preemph x[0]), window, log(max(x, guard)); the
                           // features are written as they are (no normalisation)
};

__device__ __forceinline__ float fetch_sample(const float* __restrict__ a, int i, int pad_left, int len) {
    const int j = i - pad_left;
    return (j >= 0 && j < len) ? a[j] : 0.0f;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }

// forward 8-point DFT of eight complex values in registers (three radix-2 decimation-in-frequency stages), natural order out
__device__ __forceinline__ void dft8(float2 (&a)[8]) {
    constexpr float S = 0.70710678118654752f;
    float2 b[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b[j] = make_float2(a[j].x + a[j + 4].x, a[j].y + a[j + 4].y);
        b[j + 4] = make_float2(a[j].x - a[j + 4].x, a[j].y - a[j + 4].y);
    }
    b[5] = make_float2(S * (b[5].x + b[5].y), S * (b[5].y - b[5].x));          // * W8^1 = (1 - i) / sqrt 2
    b[6] = make_float2(b[6].y, -b[6].x);                                         // * W8^2 = -i
    b[7] = make_float2(S * (b[7].y - b[7].x), -S * (b[7].x + b[7].y));          // * W8^3 = (-1 - i) / sqrt 2
    float2 c[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
        c[h + 0] = make_float2(b[h].x + b[h + 2].x, b[h].y + b[h + 2].y);
        c[h + 2] = make_float2(b[h].x - b[h + 2].x, b[h].y - b[h + 2].y);
        c[h + 1] = make_float2(b[h + 1].x + b[h + 3].x, b[h + 1].y + b[h + 3].y);
        const float2 d = make_float2(b[h + 1].x - b[h + 3].x, b[h + 1].y - b[h + 3].y);
        c[h + 3] = make_float2(d.y, -d.x);                                       // * -i
    }
    a[0] = make_float2(c[0].x + c[1].x, c[0].y + c[1].y); a[4] = make_float2(c[0].x - c[1].x, c[0].y - c[1].y);
    a[2] = make_float2(c[2].x + c[3].x, c[2].y + c[3].y); a[6] = make_float2(c[2].x - c[3].x, c[2].y - c[3].y);
    a[1] = make_float2(c[4].x + c[5].x, c[4].y + c[5].y); a[5] = make_float2(c[4].x - c[5].x, c[4].y - c[5].y);
    a[3] = make_float2(c[6].x + c[7].x, c[6].y + c[7].y); a[7] = make_float2(c[6].x - c[7].x, c[6].y - c[7].y);
}

// One wave transforms TWO frames per 512-point complex FFT (frame t in the real part, t+1 in the
// imaginary part; the two real spectra are separated afterwards: X_a[k] = (Z[k] + conj Z[N-k]) / 2,
// X_b[k] = (Z[k] - conj Z[N-k]) / 2i).  The FFT is 512 = 8 x 8 x 8 (Cooley-Tukey): every lane holds eight complex values
// and runs three 8-point DFTs in registers, with two transposes through the wave's own LDS buffer between them —
// three LDS round trips per transform where a radix-2 in-LDS FFT makes nine (that version spent its time in LDS: 51 %
// bank conflicts, profiles/r02z_pmc_per_kernel.txt).  Layouts (in complex slots; a ds_*_b64 serves 32 lanes per cycle,
// conflict-free when their slots differ mod 32):
//   n = l + 64 k:  lane l, element k  -- DFT over k -> y[l][k1], * W512^(l k1) -> slot k1 * 68 + l
//   lane (k1 = j & 7, l2 = j >> 3) reads y[l2 + 8 l1][k1] -- DFT over l1 -> z[k1][l2][m1], * W64^(l2 m1)
//                                                                         -> slot m1 * 64 + ((l2 * 8 + k1 + 8 m1) & 63)
//   lane (k1, m1 = j >> 3) reads z[k1][.][m1] -- DFT over l2 -> Z[k1 + 8 (m1 + 8 m2)] -> slot j + 64 m2 (natural order)
// Everything a wave touches in LDS is its own, so the steps are separated by wave-level fences, not workgroup
// barriers; the twiddles are staged once per workgroup.
__global__ __launch_bounds__(64 * WAVES) void logmel_kernel(FrontParams p) {
    __shared__ float2 buf[WAVES][8 * 68];
    __shared__ float pw[WAVES][2][NBIN + 3];
    __shared__ float2 tw[NFFT / 2];
    __shared__ float ys[SPAN_MAX];
    __shared__ float win_s[NFFT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int len = p.lens[b];
    const int Lp = len + p.pad_left + p.pad_right;       // padded length (reference: after pad_audio)
    const int n_valid = p.kind == 2 ? (Lp + p.hop / 2) / p.hop
                                    : Lp / p.hop + (p.kind == 1 ? 1 : 0);   // NeMo: floor((Lp + 2*(n_fft/2) - n_fft) / hop); ESPnet Stft keeps the last frame
    if (blockIdx.x == 0 && threadIdx.x == 0) p.n_frames[b] = n_valid;
    const float* a = p.audio + (size_t)b * p.audio_stride;
    if ((int)blockIdx.x * WAVES * FRAMES_PER_WAVE >= min(n_valid, p.t_max)) return;  // block-uniform
    tw[threadIdx.x] = reinterpret_cast<const float2*>(p.twiddle)[threadIdx.x];      // 256 threads, 256 twiddles
    // The padded, pre-emphasised signal under this workgroup's 16 frames goes through LDS once (coalesced loads; the
    // samples of a frame were fetched lane by lane before: 48 small loads per lane and transform) together with the
    // window, zero beyond its length.  Index i runs in the PADDED signal (reference: after pad_audio); everything
    // outside [0, Lp) is zero, and y[0] = x[0] has no predecessor.
    const int centre_off = (NFFT - p.win_length) / 2;     // 56: window centred in the 512 frame
    const int f0 = blockIdx.x * WAVES * FRAMES_PER_WAVE;
    const int i_base = p.kind == 2 ? f0 * p.hop + p.hop / 2 - p.win_length / 2 : f0 * p.hop - NFFT / 2 + centre_off;
    const int span = (WAVES * FRAMES_PER_WAVE - 1) * p.hop + NFFT;
    for (int idx = threadIdx.x; idx < span; idx += blockDim.x) {
        int i = i_base + idx;
        float y = 0.0f;
        if (p.kind == 1 && Lp > 1) {                      // torch.stft(pad_mode="reflect"): x[-k] = x[k], x[L-1+k] = x[L-1-k]
            i = i < 0 ? -i : i;
            i = i >= Lp ? 2 * (Lp - 1) - i : i;
            i = i < 0 ? 0 : i;                            // (an utterance shorter than n_fft / 2: torch refuses it, clamp)
        }
        if (p.kind == 2 && Lp > 0) {                      // kaldi ExtractWindow: reflect without repeating the end sample, as often as needed
            for (int bounce = 0; bounce < 8 && (i < 0 || i >= Lp); ++bounce) i = i < 0 ? -i - 1 : 2 * Lp - 1 - i;
            i = i < 0 ? 0 : (i >= Lp ? Lp - 1 : i);
        }
        if (i >= 0 && i < Lp) {
            const float x0 = fetch_sample(a, i, p.pad_left, len);
            if (p.kind == 2) y = x0;                      // DC removal and pre-emphasis are per FRAME (below)
            else y = i >= 1 ? x0 - p.preemph * fetch_sample(a, i - 1, p.pad_left, len) : x0;
        }
        ys[idx] = y;
    }
    for (int idx = threadIdx.x; idx < NFFT; idx += blockDim.x) win_s[idx] = idx < p.win_length ? p.window[idx] : 0.0f;
    __syncthreads();
    const int frame_base = (blockIdx.x * WAVES + wave) * FRAMES_PER_WAVE;
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // W512^j = (cos, -sin)(2 pi j / 512) for j in [0, 512): the table holds the first half, the second is its negation
    auto w512 = [&](int j) -> float2 {
        const float2 w = tw[j & 255];
        return (j & 256) ? make_float2(-w.x, -w.y) : w;
    };
    // windowed, pre-emphasised sample n of frame t (0 for a frame past the utterance)
    auto sample = [&](int t, int n) -> float {
        if (t >= n_valid || t >= p.t_max) return 0.0f;
        return ys[(t - f0) * p.hop + n] * win_s[n];
    };
    // kaldi: (x[n] - mean) - preemph (x[n - 1] - mean), the first sample against itself; mean over the frame's win_length samples
    auto frame_mean = [&](int t) -> float {
        if (t >= n_valid || t >= p.t_max) return 0.0f;
        float sm = 0.0f;
        for (int n = lane; n < p.win_length; n += 64) sm += ys[(t - f0) * p.hop + n];
        return wave_sum(sm) / (float)p.win_length;
    };
    auto sample_kaldi = [&](int t, int n, float mean) -> float {
        if (t >= n_valid || t >= p.t_max || n >= p.win_length) return 0.0f;
        const float* fr = ys + (t - f0) * p.hop;
        const float cur = fr[n] - mean, prev = fr[n > 0 ? n - 1 : 0] - mean;
        return (cur - p.preemph * prev) * win_s[n];
    };
    const int k1 = lane & 7, hi = lane >> 3;              // (k1, l2) in the second step, (k1, m1) in the third

    for (int fi = 0; fi < FRAMES_PER_WAVE; fi += 2) {
        const int t = frame_base + fi;
        if (t >= n_valid || t >= p.t_max) break;          // wave-uniform
        float2* z = buf[wave];
        // ---- step 1: two frames, samples n = lane + 64 k.  A circular shift of the FFT input only changes the phase,
        // so the 400 samples sit at n = 0 .. 399 directly.
        float2 v[8];
        if (p.kind == 2) {
            const float m0 = frame_mean(t), m1 = frame_mean(t + 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = make_float2(sample_kaldi(t, lane + 64 * k, m0), sample_kaldi(t + 1, lane + 64 * k, m1));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = make_float2(sample(t, lane + 64 * k), sample(t + 1, lane + 64 * k));
        }
        dft8(v);
#pragma unroll
        for (int q = 0; q < 8; ++q) z[q * 68 + lane] = q ? cmul(v[q], w512(lane * q)) : v[q];
        wave_sync();
        // ---- step 2: DFT over l1 for (k1, l2)
#pragma unroll
        for (int l1 = 0; l1 < 8; ++l1) v[l1] = z[k1 * 68 + hi + 8 * l1];
        wave_sync();                                       // every lane has its eight values before the buffer is rewritten
        dft8(v);
#pragma unroll
        for (int m1 = 0; m1 < 8; ++m1)
            z[m1 * 64 + ((hi * 8 + k1 + 8 * m1) & 63)] = m1 ? cmul(v[m1], w512(8 * hi * m1)) : v[m1];
        wave_sync();
        // ---- step 3: DFT over l2 for (k1, m1): Z[k1 + 8 (m1 + 8 m2)] = natural slot lane + 64 m2
#pragma unroll
        for (int l2 = 0; l2 < 8; ++l2) v[l2] = z[hi * 64 + ((l2 * 8 + k1 + 8 * hi) & 63)];
        wave_sync();
        dft8(v);
#pragma unroll
        for (int m2 = 0; m2 < 8; ++m2) z[lane + 64 * m2] = v[m2];
        wave_sync();
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
                p.raw[((size_t)b * p.t_max + t + f) * p.n_mels + m] = p.kind >= 1 ? logf(fmaxf(acc, p.log_guard)) : logf(acc + p.log_guard);
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

// ESPnet GlobalMVN: (x - mean[m]) * istd[m] on the valid frames, zeros beyond.  grid (ceil(t_max * n_mels / 256), B)
__global__ __launch_bounds__(256) void feat_mvn_kernel(const float* __restrict__ raw, const int32_t* __restrict__ n_frames, int t_max,
                                                       int n_mels, const float* __restrict__ mean, const float* __restrict__ istd,
                                                       float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t_max * n_mels) return;
    const int t = i / n_mels, m = i - t * n_mels;
    const size_t at = (size_t)b * t_max * n_mels + i;
    out[at] = t < n_frames[b] ? (raw[at] - mean[m]) * istd[m] : 0.0f;
}

}  // namespace

int rs_launch_frontend(rs_ctx* ctx, const float* audio, const int32_t* lens, int B, int audio_stride,
                       int pad_left, int pad_right, int t_max, float* feats, int32_t* n_frames, float* raw,
                       hipStream_t s) {
    const rs_dims& d = ctx->d;
    if (B <= 0 || t_max <= 0) return RS_OK;
    if (d.n_fft != NFFT) return rs_fail(ctx, RS_EINVAL, "frontend: only n_fft=512 is built");
    if (d.n_mels > 128 || d.n_mels % 4 || d.win_length > NFFT || d.hop_length < 1 || d.hop_length > MAX_HOP)
        return rs_fail(ctx, RS_EINVAL, "frontend: n_mels must be a multiple of 4 up to 128, win <= 512, hop <= %d", MAX_HOP);
    FrontParams p;
    p.audio = audio; p.lens = lens; p.raw = raw; p.n_frames = n_frames;
    p.window = ctx->fe_window; p.twiddle = ctx->fe_twiddle; p.fb_idx = ctx->fe_fb_idx; p.fb_w = ctx->fe_fb_w;
    p.audio_stride = audio_stride; p.pad_left = pad_left; p.pad_right = pad_right; p.t_max = t_max;
    p.n_mels = d.n_mels; p.win_length = d.win_length; p.hop = d.hop_length;
    p.preemph = d.preemph; p.log_guard = d.log_guard;
    p.kind = d.frontend_kind;
    const int fpb = WAVES * FRAMES_PER_WAVE;
    const dim3 grid((t_max + fpb - 1) / fpb, B), block(64 * WAVES);
    if (d.frontend_kind == 2) {                   // kaldi fbank: the log-mel energies ARE the features; frames past an utterance are zeros
        p.raw = feats;
        RS_HIP(ctx, hipMemsetAsync(feats, 0, (size_t)B * t_max * d.n_mels * 4, s));
    }
    const double bytes = (double)B * ((double)t_max * d.hop_length * 4.0 + (double)t_max * d.n_mels * 4.0 * 3.0);
    rs_prof_begin(ctx, RS_PROF_FRONTEND, s, (double)B * t_max * (5.0 * 512 * 9 + 3 * 257 + 2 * 600), bytes);
    hipLaunchKernelGGL(logmel_kernel, grid, block, 0, s, p);
    if (d.frontend_kind == 2) {
    } else if (d.frontend_kind == 1)
        hipLaunchKernelGGL(feat_mvn_kernel, dim3((t_max * d.n_mels + 255) / 256, B), dim3(256), 0, s, raw, n_frames, t_max, d.n_mels,
                           ctx->fe_mvn_mean, ctx->fe_mvn_istd, feats);
    else
        hipLaunchKernelGGL(feat_normalize_kernel, dim3(B), dim3(1024), 0, s, raw, n_frames, t_max, d.n_mels, d.norm_eps,
                           feats);
    rs_prof_end(ctx, RS_PROF_FRONTEND, s);
    RS_CHECK_LAUNCH(ctx, "frontend");
    return RS_OK;
}

/*
 * rnnt_greedy.c — CPU restatement of the RNN-T prediction network, joint network and greedy
 * decode loop.  TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/, smoke() and the
 * cpu_baseline leg of bench.py use it.
 *
 * Restates [UPSTREAM, NeMo >= 2.6.1, not vendored in /root/reference]:
 *   RNNTDecoder.predict        Embedding -> LSTM x L (gate order i,f,g,o)       (HF modeling_parakeet.py:831-876)
 *   RNNTJoint.joint_after_projection   relu(enc + pred) -> Linear -> argmax      (HF modeling_parakeet.py:879-894)
 *   GreedyBatchedRNNTInfer     blank -> next frame; non-blank -> emit, update state,
 *                              at most max_symbols emissions per frame           (HF generation_parakeet.py:141-163)
 * Call site in the reference: model.transcribe(..) pkg/nemo-asr/src/transcribe.py:48-53.
 *
 * Float32 with a FIXED accumulation order — the same one the HIP kernels use
 * (reazonspeech_amd/csrc/k_rnnt.hip) so that token ids can be compared bit for bit:
 *   dot(a, w, K) = ((p0 + p1) + p2) + .. ,  p_s = chain over slice s of K/S contiguous k
 *   (S = 16 for the LSTM gate products, 8 for the joint and prediction projections),
 *   chain order inside a slice: for u in 16-blocks: for e in 0..3: for kk in 0..3: k = base+16u+4kk+e,
 *   each step acc = fmaf(a[k], w[k], acc) starting from 0.
 * exp/sigmoid/tanh are the same polynomial (only + - * / and fmaf).
 * Build: gcc -O2 -mfma -ffp-contract=off -shared -fPIC (oracle/build.py).
 */
#include <stdlib.h>
#include <string.h>

#include "rnnt_math.h"

float rs_oracle_expf(float x) { return rs_expf(x); }
float rs_oracle_sigmoidf(float x) { return rs_sigmoidf(x); }
float rs_oracle_tanhf(float x) { return rs_tanhf(x); }

/* a may be the concatenation of two vectors: a0[0..K0) then a1[0..K-K0) */
static inline float dot_ordered2(const float* a0, int K0, const float* a1, const float* w, int K, int splitk) {
    float part[SPLITK_MAX];
    const int ks = K / splitk;
    for (int s = 0; s < splitk; ++s) {
        float acc = 0.0f;
        const int base = s * ks;
        for (int u = 0; u < ks; u += 16)
            for (int e = 0; e < 4; ++e)
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = base + u + 4 * kk + e;
                    const float av = k < K0 ? a0[k] : a1[k - K0];
                    acc = fmaf(av, w[k], acc);
                }
        part[s] = acc;
    }
    float sum = part[0];
    for (int s = 1; s < splitk; ++s) sum = sum + part[s];
    return sum;
}

float rs_oracle_dot(const float* a, const float* w, int K) { return dot_ordered2(a, K, a, w, K, SPLITK_TILE); }

/* one LSTM layer step for one row: x[H], h[H], c[H] -> h_out[H], c_out[H]; W [4H][2H] = [W_ih | W_hh],
 * bias [4H] = b_ih + b_hh (summed in float32 on the host, same as the device weight prep) */
void rs_oracle_lstm_step(const float* x, const float* h, const float* c, const float* W, const float* bias, int H,
                         float* h_out, float* c_out) {
    const int K = 2 * H;
    for (int u = 0; u < H; ++u) {
        float z[4];
        for (int g = 0; g < 4; ++g) z[g] = dot_ordered2(x, H, h, W + (size_t)(g * H + u) * K, K, SPLITK_LSTM) + bias[g * H + u];
        const float ig = rs_sigmoidf(z[0]), fg = rs_sigmoidf(z[1]), gg = rs_tanhf(z[2]), og = rs_sigmoidf(z[3]);
        const float cn = fmaf(fg, c[u], ig * gg);
        c_out[u] = cn;
        h_out[u] = og * rs_tanhf(cn);
    }
}

/* joint activation: 0 = ReLU ([UPSTREAM] NeMo RNNTJoint), 1 = tanh ([UPSTREAM] ESPnet2 JointNetwork, the model family of
 * reazonspeech.espnet.asr: pkg/espnet-asr/src/transcribe.py:26-32) with the shared polynomial of rnnt_math.h, which the HIP
 * tile kernel evaluates operation for operation (k_rnnt.hip: DecodeState.joint_act). */
static int g_joint_act = 0;
void rs_oracle_set_joint_act(int act) { g_joint_act = act; }
int rs_oracle_get_joint_act(void) { return g_joint_act; }

/* logits[V] = Wo . act(f + g) + bo ; returns argmax (lowest index on ties) */
int rs_oracle_joint_argmax(const float* f, const float* g, const float* Wo, const float* bo, int J, int V,
                           float* logits_out /* may be NULL */) {
    float* a = (float*)malloc(sizeof(float) * J);
    for (int k = 0; k < J; ++k) a[k] = g_joint_act ? rs_tanhf(f[k] + g[k]) : fmaxf(f[k] + g[k], 0.0f);
    int best = 0;
    float bestv = -INFINITY;
    for (int v = 0; v < V; ++v) {
        const float val = rs_oracle_dot(a, Wo + (size_t)v * J, J) + bo[v];
        if (logits_out) logits_out[v] = val;
        if (val > bestv) { bestv = val; best = v; }
    }
    free(a);
    return best;
}

/* Greedy decode of B utterances (each independently, exactly as the reference runs them one at
 * a time: transcribe.py:48-50 batch_size=1).  Returns 0, or -5 if some utterance overflowed u_max. */
int rs_oracle_rnnt_greedy(const float* f, const int32_t* enc_lens, int B, int Tp, int J, int H, int L, int V,
                          int blank, int max_symbols, const float* embed, const float* const* lstm_w,
                          const float* const* lstm_b, const float* Wp, const float* bp, const float* Wo,
                          const float* bo, int u_max, int32_t* ids, int32_t* frames, int32_t* n_ids) {
    int overflow = 0;
    float* h = (float*)malloc(sizeof(float) * L * H * 2);
    float* c = (float*)malloc(sizeof(float) * L * H * 2);
    float* g = (float*)malloc(sizeof(float) * J);
    for (int b = 0; b < B; ++b) {
        float *hc = h, *hn = h + L * H, *cc = c, *cn = c + L * H;
        memset(h, 0, sizeof(float) * L * H * 2);
        memset(c, 0, sizeof(float) * L * H * 2);
        int token = blank, t = 0, sym = 0, n = 0;
        int need_pred = 1;
        while (1) {
            if (need_pred) {
                const float* x = embed + (size_t)token * H;
                for (int l = 0; l < L; ++l) {
                    rs_oracle_lstm_step(x, hc + l * H, cc + l * H, lstm_w[l], lstm_b[l], H, hn + l * H, cn + l * H);
                    x = hn + l * H;
                }
                for (int j = 0; j < J; ++j) g[j] = rs_oracle_dot(hn + (L - 1) * H, Wp + (size_t)j * H, H) + bp[j];
                float* tmp = hc; hc = hn; hn = tmp;
                tmp = cc; cc = cn; cn = tmp;
                need_pred = 0;
            }
            if (t >= enc_lens[b]) break;
            const int k = rs_oracle_joint_argmax(f + ((size_t)b * Tp + t) * J, g, Wo, bo, J, V, NULL);
            if (k == blank) {
                t += 1; sym = 0;
            } else {
                if (n < u_max) { ids[(size_t)b * u_max + n] = k; frames[(size_t)b * u_max + n] = t; n += 1; }
                else overflow = 1;
                token = k;
                need_pred = 1;
                sym += 1;
                if (sym >= max_symbols) { t += 1; sym = 0; }
            }
        }
        n_ids[b] = n;
    }
    free(h); free(c); free(g);
    return overflow ? -5 : 0;
}

/*
 * k2_greedy.c — CPU restatement of the decode side of `reazonspeech.k2.asr`: icefall's stateless decoder, the joiner and
 * sherpa-onnx's offline greedy search.  TEST INFRASTRUCTURE (see oracle/__init__.py): the bit-exact checker of the HIP decode
 * loop for the Zipformer family (reazonspeech_amd/csrc/k_rnnt.hip: k2_decoder_kernel + the exact-f32 tile kernels).
 *
 * Restates [UPSTREAM, not vendored in /root/reference; PARITY UNPINNED — see oracle/zipformer.py]:
 *   icefall zipformer/decoder.py Decoder.forward(need_pad=False)   embedding of the last two tokens (-1 -> zeros), Conv1d(D, D, 2,
 *                                                                  groups = D / 4, bias = False), ReLU
 *   icefall export-onnx.py OnnxDecoder / OnnxJoiner                decoder_proj; output_linear(tanh(encoder_proj + decoder_proj))
 *   sherpa-onnx OfflineTransducerGreedySearchDecoder::Decode       context [-1, blank]; one argmax per frame; blank (0) and <unk>
 *                                                                  are not emitted; timestamps = frame indices
 * Call site in the reference: model.decode_stream(stream), pkg/k2-asr/src/transcribe.py:39.
 *
 * Float32 in the fixed order of rnnt_greedy.c (rs_oracle_dot: 8 K slices, 16-blocks, e then kk); the grouped convolution sums
 * the token before last first, then the last token, input channels ascending.
 */
#include <stdlib.h>
#include <string.h>

#include "rnnt_math.h"

float rs_oracle_dot(const float* a, const float* w, int K);
int rs_oracle_joint_argmax(const float* f, const float* g, const float* Wo, const float* bo, int J, int V, float* logits_out);
void rs_oracle_set_joint_act(int act);
int rs_oracle_get_joint_act(void);

/* h[D] = relu(conv(embed[t0], embed[t1])) ; conv_w [D][4][2] */
void rs_oracle_k2_decoder(const float* embed, const float* conv_w, int D, int t0, int t1, float* h) {
    for (int c = 0; c < D; ++c) {
        const int g4 = c & ~3;
        float acc = 0.0f;
        for (int i = 0; i < 4; ++i) acc = fmaf(conv_w[(c * 4 + i) * 2 + 0], t0 >= 0 ? embed[(size_t)t0 * D + g4 + i] : 0.0f, acc);
        for (int i = 0; i < 4; ++i) acc = fmaf(conv_w[(c * 4 + i) * 2 + 1], t1 >= 0 ? embed[(size_t)t1 * D + g4 + i] : 0.0f, acc);
        h[c] = fmaxf(acc, 0.0f);
    }
}

int rs_oracle_k2_greedy(const float* f, const int32_t* enc_lens, int B, int Tp, int J, int D, int V, int blank, int unk,
                        const float* embed, const float* conv_w, const float* Wp, const float* bp, const float* Wo, const float* bo,
                        int u_max, int32_t* ids, int32_t* frames, int32_t* n_ids) {
    int overflow = 0;
    float* h = (float*)malloc(sizeof(float) * D);
    float* g = (float*)malloc(sizeof(float) * J);
    const int act_before = rs_oracle_get_joint_act();    /* the joint activation is a setting of the shared checker library */
    rs_oracle_set_joint_act(1);
    for (int b = 0; b < B; ++b) {
        int t0 = -1, t1 = blank, n = 0;
        int need = 1;
        for (int t = 0; t < enc_lens[b]; ++t) {
            if (need) {
                rs_oracle_k2_decoder(embed, conv_w, D, t0, t1, h);
                for (int j = 0; j < J; ++j) g[j] = rs_oracle_dot(h, Wp + (size_t)j * D, D) + bp[j];
                need = 0;
            }
            const int k = rs_oracle_joint_argmax(f + ((size_t)b * Tp + t) * J, g, Wo, bo, J, V, NULL);
            if (k != blank && k != unk) {
                if (n < u_max) { ids[(size_t)b * u_max + n] = k; frames[(size_t)b * u_max + n] = t; n += 1; }
                else overflow = 1;
                t0 = t1; t1 = k;
                need = 1;
            }
        }
        n_ids[b] = n;
    }
    free(h); free(g);
    rs_oracle_set_joint_act(act_before);
    return overflow ? -5 : 0;
}

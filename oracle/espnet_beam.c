/*
 * espnet_beam.c — CPU restatement of ESPnet's "default" transducer beam search (Graves 2012, as ESPnet2 implements it),
 * the decode that reazonspeech.espnet.asr actually runs: the reference builds Speech2Text with its defaults
 * (pkg/espnet-asr/src/transcribe.py:27-31: only lm_weight=0 is overridden), i.e. beam_size 20, search_type "default",
 * score_norm True, nbest 1, no LM.  TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/ use it, as the checker of
 * the HIP search (reazonspeech_amd/csrc/k_rnnt_beam.hip).
 *
 * PARITY UNPINNED against upstream: espnet2/asr/transducer/beam_search_transducer.py (BeamSearchTransducer.
 * default_beam_search, sort_nbest) is a third-party dependency that is absent here (pkg/espnet-asr/pyproject.toml:
 * espnet == 202308).  oracle/espnet.py::default_beam_search_torch restates the published algorithm statement for
 * statement in torch; this file follows it decision for decision (tests/test_oracle_espnet_beam.py compares the two)
 * and adds a FIXED float32 evaluation order, shared with the HIP kernels, so that the device search can be checked bit
 * for bit (tokens and scores):
 *
 *   prediction net / logits   exactly rs_oracle_lstm_step / rs_oracle_dot / rs_oracle_joint_argmax (rnnt_greedy.c; the
 *                             joint activation is the one set by rs_oracle_set_joint_act: tanh for ESPnet)
 *   log-softmax               rs_oracle_lse (rnnt_alsd.c): logp(v) = z[v] - lse
 *   per frame                 hyps = the hypotheses kept at the previous frame, in ascending score order (stable);
 *                             loop: pop the FIRST maximum of hyps by score; run the prediction net on its last label
 *                             from the state stored with it (the state BEFORE that label); append to `kept` the blank
 *                             extension (score + logp(blank), same labels, same stored state); append to hyps the
 *                             beam_k = min(beam, V - 1) best non-blank labels by (z desc, v asc), in that order, each with
 *                             score + logp(v) and the state AFTER the popped hypothesis' last label;
 *                             stop when at least `beam` entries of `kept` score strictly above the maximum of hyps: those
 *                             entries (all of them, there may be more than `beam`), sorted ascending by score with ties
 *                             in `kept` order, are the next frame's hyps
 *   scores                    accumulate in float32 (upstream: Python floats of float32 log-probabilities)
 *   final                     the first maximum of score / len(yseq) (score_norm; yseq counts the leading blank) or of
 *                             score over the last frame's survivors in their (ascending score) order
 *
 * A frame that needs more than `max_pops` pops, or an utterance with more than `out_cap` labels, is reported through the
 * return code (-5) exactly as the device does (RS_EOVERFLOW); nothing is truncated silently.
 */
#include <stdlib.h>
#include <string.h>

#include "rnnt_math.h"

void rs_oracle_lstm_step(const float* x, const float* h, const float* c, const float* W, const float* bias, int H,
                         float* h_out, float* c_out);
int rs_oracle_joint_argmax(const float* f, const float* g, const float* Wo, const float* bo, int J, int V,
                           float* logits_out);
float rs_oracle_dot(const float* a, const float* w, int K);
float rs_oracle_lse(const float* z, int V);

typedef struct {
    float score;
    int node;      /* label-trie node of the whole sequence when tok < 0, of the sequence without its last label otherwise */
    int tok;       /* last label not yet in the trie, or -1 */
    int len;       /* len(yseq): labels + the leading blank */
    int state;     /* index of the prediction-net state stored BEFORE the last label */
    int alive;
} bhyp_t;

typedef struct { int parent, tok, t; } node_t;   /* t: the frame at which the label was appended (its hypothesis was opened AND popped there) */

typedef struct { float* v; int n, cap, width; } pool_t;     /* states: h [L][H] then c [L][H] */

static int pool_push(pool_t* p, const float* h, const float* c, int LH) {
    if (p->n == p->cap) {
        p->cap = p->cap ? 2 * p->cap : 64;
        p->v = (float*)realloc(p->v, sizeof(float) * (size_t)p->cap * p->width);
    }
    memcpy(p->v + (size_t)p->n * p->width, h, sizeof(float) * LH);
    memcpy(p->v + (size_t)p->n * p->width + LH, c, sizeof(float) * LH);
    return p->n++;
}

/* f [B][Tp][J] (joint.enc output).  Outputs the best hypothesis per utterance: ids [B][out_cap] (without the leading
 * blank), frames [B][out_cap] (may be NULL: the frame each label was appended at — upstream ESPnet keeps none, [UPSTREAM] NeMo's
 * default_beam_search keeps them as Hypothesis.timestep), n_ids [B], scores [B], and pops [B] = prediction-net evaluations spent
 * (the work measure of the search).  Returns 0, or -5 on overflow (max_pops per frame / out_cap). */
int rs_oracle_espnet_beam(const float* f, const int32_t* enc_lens, int B, int Tp, int J, int H, int L, int V, int blank,
                          const float* embed, const float* const* lstm_w, const float* const* lstm_b, const float* Wp,
                          const float* bp, const float* Wo, const float* bo, int beam, int score_norm, int max_pops,
                          int out_cap, int32_t* ids, int32_t* frames, int32_t* n_ids, float* scores, int32_t* pops) {
    int overflow = 0;
    if (beam > V) beam = V;
    const int beam_k = beam < V - 1 ? beam : V - 1;
    const int LH = L * H;
    float* z = (float*)malloc(sizeof(float) * V);
    float* g = (float*)malloc(sizeof(float) * J);
    float* hn = (float*)malloc(sizeof(float) * LH);
    float* cn = (float*)malloc(sizeof(float) * LH);
    for (int b = 0; b < B; ++b) {
        const int T = enc_lens[b];
        int n_nodes = 1, cap_nodes = 1024;
        node_t* nodes = (node_t*)malloc(sizeof(node_t) * cap_nodes);
        nodes[0].parent = -1; nodes[0].tok = blank; nodes[0].t = -1;
        pool_t pool[2] = {{NULL, 0, 0, 2 * LH}, {NULL, 0, 0, 2 * LH}};
        int cur = 0;
        memset(hn, 0, sizeof(float) * LH); memset(cn, 0, sizeof(float) * LH);
        pool_push(&pool[0], hn, cn, LH);
        const int cap_h = max_pops * (beam_k + 1) + 1;
        bhyp_t* hyps = (bhyp_t*)malloc(sizeof(bhyp_t) * cap_h);
        bhyp_t* kept = (bhyp_t*)malloc(sizeof(bhyp_t) * (max_pops + 1));
        int n_kept = 1, n_pops_total = 0, failed = 0;
        kept[0] = (bhyp_t){0.0f, 0, -1, 1, 0, 1};
        for (int t = 0; t < T && !failed; ++t) {
            /* hyps <- kept (already ordered), states stay in pool[cur] */
            int n_h = n_kept;
            memcpy(hyps, kept, sizeof(bhyp_t) * n_kept);
            n_kept = 0;
            int n_pop = 0;
            for (;;) {
                if (n_pop == max_pops) { failed = 1; break; }
                int mi = -1;
                for (int i = 0; i < n_h; ++i) if (hyps[i].alive && (mi < 0 || hyps[i].score > hyps[mi].score)) mi = i;
                bhyp_t mh = hyps[mi];
                hyps[mi].alive = 0;
                ++n_pop; ++n_pops_total;
                if (mh.tok >= 0) {                                     /* its sequence enters the trie now */
                    if (n_nodes == cap_nodes) { cap_nodes *= 2; nodes = (node_t*)realloc(nodes, sizeof(node_t) * cap_nodes); }
                    nodes[n_nodes].parent = mh.node; nodes[n_nodes].tok = mh.tok; nodes[n_nodes].t = t;
                    mh.node = n_nodes++; mh.tok = -1;
                }
                const float* sv = pool[cur].v + (size_t)mh.state * 2 * LH;
                const float* x = embed + (size_t)nodes[mh.node].tok * H;
                for (int l = 0; l < L; ++l) {
                    rs_oracle_lstm_step(x, sv + l * H, sv + LH + l * H, lstm_w[l], lstm_b[l], H, hn + l * H, cn + l * H);
                    x = hn + l * H;
                }
                for (int j = 0; j < J; ++j) g[j] = rs_oracle_dot(hn + (L - 1) * H, Wp + (size_t)j * H, H) + bp[j];
                const int after = pool_push(&pool[cur], hn, cn, LH);
                rs_oracle_joint_argmax(f + ((size_t)b * Tp + t) * J, g, Wo, bo, J, V, z);
                const float lse = rs_oracle_lse(z, V);
                kept[n_kept] = mh;
                kept[n_kept].score = mh.score + (z[blank] - lse);
                kept[n_kept].alive = 1;
                ++n_kept;
                float pz = INFINITY; int pv = -1;
                for (int j = 0; j < beam_k; ++j) {
                    float bz = -INFINITY; int bv = -1;
                    for (int v = 0; v < V; ++v) {
                        if (v == blank) continue;
                        if (!(z[v] < pz || (z[v] == pz && v > pv))) continue;
                        if (bv < 0 || z[v] > bz) { bz = z[v]; bv = v; }
                    }
                    if (bv < 0) break;
                    hyps[n_h++] = (bhyp_t){mh.score + (bz - lse), mh.node, bv, mh.len + 1, after, 1};
                    pz = bz; pv = bv;
                }
                float hmax = -INFINITY;
                for (int i = 0; i < n_h; ++i) if (hyps[i].alive && hyps[i].score > hmax) hmax = hyps[i].score;
                int n_good = 0;
                for (int i = 0; i < n_kept; ++i) if (kept[i].score > hmax) ++n_good;
                if (n_good >= beam) {
                    /* survivors ascending by score, ties in kept order; their states move to the other pool */
                    bhyp_t* out = hyps;                             /* hyps is dead from here: reuse it as the sort buffer */
                    pool[cur ^ 1].n = 0;
                    int n_out = 0;
                    for (int i = 0; i < n_kept; ++i) {
                        if (!(kept[i].score > hmax)) continue;
                        int rank = 0;
                        for (int o = 0; o < n_kept; ++o) {
                            if (!(kept[o].score > hmax)) continue;
                            if (kept[o].score < kept[i].score || (kept[o].score == kept[i].score && o < i)) ++rank;
                        }
                        out[rank] = kept[i];
                        ++n_out;
                    }
                    for (int i = 0; i < n_out; ++i) {
                        const float* s0 = pool[cur].v + (size_t)out[i].state * 2 * LH;
                        out[i].state = pool_push(&pool[cur ^ 1], s0, s0 + LH, LH);
                    }
                    memcpy(kept, out, sizeof(bhyp_t) * n_out);
                    n_kept = n_out;
                    cur ^= 1;
                    break;
                }
            }
        }
        pops[b] = n_pops_total;
        if (failed) { overflow = 1; n_ids[b] = 0; scores[b] = 0.0f; }
        else {
            int best = 0;
            float bn = 0.0f;
            for (int i = 0; i < n_kept; ++i) {
                const float norm = score_norm ? kept[i].score / (float)kept[i].len : kept[i].score;
                if (i == 0 || norm > bn) { best = i; bn = norm; }
            }
            int n = kept[best].len - 1;
            scores[b] = kept[best].score;
            if (n > out_cap) { overflow = 1; n_ids[b] = 0; }
            else {
                int node = kept[best].node;                          /* survivors are always in the trie */
                for (int q = n - 1; q >= 0; --q) {
                    ids[(size_t)b * out_cap + q] = nodes[node].tok;
                    if (frames) frames[(size_t)b * out_cap + q] = nodes[node].t;
                    node = nodes[node].parent;
                }
                n_ids[b] = n;
            }
        }
        free(nodes); free(pool[0].v); free(pool[1].v); free(hyps); free(kept);
    }
    free(z); free(g); free(hn); free(cn);
    return overflow ? -5 : 0;
}

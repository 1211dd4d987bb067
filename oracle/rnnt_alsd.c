/*
 * rnnt_alsd.c — CPU restatement of alignment-length synchronous beam search (ALSD) over the RNN-T prediction and
 * joint networks.  TEST INFRASTRUCTURE (see oracle/__init__.py): only tests/ use it, as the checker of the HIP
 * ALSD path (reazonspeech_amd/csrc/k_rnnt_alsd.hip).
 *
 * PARITY UNPINNED against upstream: the search is [UPSTREAM] NeMo BeamRNNTInfer.align_length_sync_decoding, the
 * strategy the reference's post-processing is written for (pkg/nemo-asr/src/decode.py:29,38-41,48), and NeMo is
 * not available here.  This file follows oracle/alsd.py (the readable restatement of the published algorithm, Saon
 * et al., ICASSP 2020, with the upstream structure as far as the reference shows it) decision for decision; the
 * two are compared in tests/test_oracle_alsd.py.  What this file adds is a FIXED float32 evaluation order, shared
 * with the HIP kernels, so that the device search can be checked bit for bit (tokens, alignment steps, scores):
 *
 *   logits      exactly rs_oracle_joint_argmax's (rnnt_greedy.c)
 *   log-softmax m = max_v z[v];  S = sum of rs_expf(z[v] - m): 64 partial sums, partial l adding v = l, l+64, ..
 *               in increasing v, combined by the tree p[l] += p[l + off], off = 32,16,..,1;  lse = m + rs_logf(S);
 *               logp(v) = z[v] - lse
 *   expansion   per live hypothesis, in beam order: [blank, the `beam` best non-blank tokens by (z desc, v asc)],
 *               candidate score = hypothesis score + logp (float32)
 *   selection   the `beam` best candidates by (score desc, expansion order asc)
 *   recombine   in selection order: a hypothesis whose label sequence equals an earlier kept one adds its score
 *               into that one (rs_logaddexpf); "upstream" mode keeps the duplicate in the beam, "merge" drops it
 *   final       a blank expansion taken at the last frame is a finished hypothesis; its score is read AFTER this
 *               step's recombination (oracle/alsd.py appends the same object to `final` and to the beam); the
 *               winner is the first maximum of score / (labels + 1) (score_norm) or of score
 */
#include <stdlib.h>
#include <string.h>

#include "rnnt_math.h"

void rs_oracle_lstm_step(const float* x, const float* h, const float* c, const float* W, const float* bias, int H,
                         float* h_out, float* c_out);
int rs_oracle_joint_argmax(const float* f, const float* g, const float* Wo, const float* bo, int J, int V,
                           float* logits_out);
float rs_oracle_dot(const float* a, const float* w, int K);

float rs_oracle_logf(float x) { return rs_logf(x); }
float rs_oracle_logaddexpf(float a, float b) { return rs_logaddexpf(a, b); }

/* log-sum-exp of z[0..V) in the order documented above */
float rs_oracle_lse(const float* z, int V) {
    float m = -INFINITY;
    for (int v = 0; v < V; ++v) if (z[v] > m) m = z[v];
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float s = 0.0f;
        for (int v = l; v < V; v += 64) s = s + rs_expf(z[v] - m);
        p[l] = s;
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int l = 0; l < off; ++l) p[l] = p[l] + p[l + off];
    return m + rs_logf(p[0]);
}

typedef struct {
    int n;          /* labels emitted */
    float score;
    int* y;         /* [cap] */
    int* steps;     /* [cap] alignment index i = t + u of each label */
    float* h;       /* [L][H] prediction-net state after consuming the last label (blank at the start) */
    float* c;
    float* g;       /* [J] joint.pred(prediction-net output) */
} hyp_t;

typedef struct { float score; int parent; int tok; /* -1 = blank */ } cand_t;

static hyp_t* hyps_new(int n, int cap, int L, int H, int J) {
    hyp_t* a = (hyp_t*)calloc(n, sizeof(hyp_t));
    for (int i = 0; i < n; ++i) {
        a[i].y = (int*)malloc(sizeof(int) * (cap + 1));
        a[i].steps = (int*)malloc(sizeof(int) * (cap + 1));
        a[i].h = (float*)malloc(sizeof(float) * L * H);
        a[i].c = (float*)malloc(sizeof(float) * L * H);
        a[i].g = (float*)malloc(sizeof(float) * J);
    }
    return a;
}
static void hyps_free(hyp_t* a, int n) {
    for (int i = 0; i < n; ++i) { free(a[i].y); free(a[i].steps); free(a[i].h); free(a[i].c); free(a[i].g); }
    free(a);
}

/* label sequence of candidate c: its parent's labels, plus c.tok unless blank */
static int cand_len(const hyp_t* cur, const cand_t* c) { return cur[c->parent].n + (c->tok >= 0); }
static int cand_label(const hyp_t* cur, const cand_t* c, int q) { return q < cur[c->parent].n ? cur[c->parent].y[q] : c->tok; }
static int cand_equal(const hyp_t* cur, const cand_t* a, const cand_t* b) {
    const int n = cand_len(cur, a);
    if (n != cand_len(cur, b)) return 0;
    for (int q = 0; q < n; ++q) if (cand_label(cur, a, q) != cand_label(cur, b, q)) return 0;
    return 1;
}

/* ALSD over B utterances, each independently.  f [B][Tp][J] (joint.enc output), u_max[b] = label budget beyond
 * the frames of utterance b (the search runs enc_lens[b] + u_max[b] alignment steps).  Outputs the best
 * hypothesis per utterance: ids/steps [B][out_cap], n_ids [B], scores [B].  Returns 0, or -5 when a result did
 * not fit out_cap (it is truncated). */
int rs_oracle_rnnt_alsd(const float* f, const int32_t* enc_lens, int B, int Tp, int J, int H, int L, int V, int blank,
                        const float* embed, const float* const* lstm_w, const float* const* lstm_b, const float* Wp,
                        const float* bp, const float* Wo, const float* bo, int beam, const int32_t* u_max,
                        int score_norm, int merge, int out_cap, int32_t* ids, int32_t* steps, int32_t* n_ids,
                        float* scores) {
    int overflow = 0;
    if (beam > V - 1) beam = V - 1;
    float* z = (float*)malloc(sizeof(float) * V);
    float* hn = (float*)malloc(sizeof(float) * L * H);
    float* cn = (float*)malloc(sizeof(float) * L * H);
    cand_t* cands = (cand_t*)malloc(sizeof(cand_t) * beam * (beam + 1));
    cand_t* sel = (cand_t*)malloc(sizeof(cand_t) * beam);
    int* sel_src = (int*)malloc(sizeof(int) * beam);
    int* dup = (int*)malloc(sizeof(int) * beam);
    for (int b = 0; b < B; ++b) {
        const int T = enc_lens[b], n_steps = T + u_max[b];
        const int cap = n_steps > 0 ? n_steps : 1;
        hyp_t* cur = hyps_new(beam, cap, L, H, J);
        hyp_t* nxt = hyps_new(beam, cap, L, H, J);
        hyp_t fin = {-1, 0.0f, (int*)malloc(sizeof(int) * (cap + 1)), (int*)malloc(sizeof(int) * (cap + 1)), NULL, NULL, NULL};
        float fin_norm = 0.0f;
        int n_cur = 1;
        {   /* start: no labels, the prediction net has consumed the blank (start-of-sequence) token from zero state */
            hyp_t* h0 = &cur[0];
            h0->n = 0; h0->score = 0.0f;
            memset(hn, 0, sizeof(float) * L * H);
            memset(cn, 0, sizeof(float) * L * H);
            const float* x = embed + (size_t)blank * H;
            for (int l = 0; l < L; ++l) {
                rs_oracle_lstm_step(x, hn + l * H, cn + l * H, lstm_w[l], lstm_b[l], H, h0->h + l * H, h0->c + l * H);
                x = h0->h + l * H;
            }
            for (int j = 0; j < J; ++j) h0->g[j] = rs_oracle_dot(h0->h + (L - 1) * H, Wp + (size_t)j * H, H) + bp[j];
        }
        for (int i = 0; i < n_steps; ++i) {
            int n_c = 0, any_live = 0;
            for (int s = 0; s < n_cur; ++s) {
                const int t = i - cur[s].n;
                if (t > T - 1) continue;
                any_live = 1;
                rs_oracle_joint_argmax(f + ((size_t)b * Tp + t) * J, cur[s].g, Wo, bo, J, V, z);
                const float lse = rs_oracle_lse(z, V);
                cands[n_c].score = cur[s].score + (z[blank] - lse);
                cands[n_c].parent = s; cands[n_c].tok = -1;
                ++n_c;
                float pz = INFINITY; int pv = -1;          /* previous pick in the (z desc, v asc) order */
                for (int j = 0; j < beam; ++j) {
                    float bz = -INFINITY; int bv = -1;
                    for (int v = 0; v < V; ++v) {
                        if (v == blank) continue;
                        if (!(z[v] < pz || (z[v] == pz && v > pv))) continue;
                        if (bv < 0 || z[v] > bz) { bz = z[v]; bv = v; }
                    }
                    if (bv < 0) break;
                    cands[n_c].score = cur[s].score + (bz - lse);
                    cands[n_c].parent = s; cands[n_c].tok = bv;
                    ++n_c;
                    pz = bz; pv = bv;
                }
            }
            if (!any_live) break;
            /* the `beam` best candidates, ties in expansion order */
            int n_sel = 0;
            for (int c = 0; c < n_c; ++c) {
                int rank = 0;
                for (int o = 0; o < n_c; ++o)
                    if (cands[o].score > cands[c].score || (cands[o].score == cands[c].score && o < c)) ++rank;
                if (rank < beam) { sel[rank] = cands[c]; sel_src[rank] = c; if (rank + 1 > n_sel) n_sel = rank + 1; }
            }
            /* recombination */
            for (int j = 0; j < n_sel; ++j) {
                dup[j] = 0;
                for (int k = 0; k < j; ++k)
                    if (!dup[k] && cand_equal(cur, &sel[k], &sel[j])) {
                        sel[k].score = rs_logaddexpf(sel[k].score, sel[j].score);
                        dup[j] = 1;
                        break;
                    }
            }
            /* finished hypotheses: blank taken at the last frame (score as it stands after recombination) */
            for (int c = 0; c < n_c; ++c) {
                if (cands[c].tok >= 0 || i - cur[cands[c].parent].n != T - 1) continue;
                float sc = cands[c].score;
                for (int j = 0; j < n_sel; ++j) if (sel_src[j] == c) sc = sel[j].score;
                const hyp_t* p = &cur[cands[c].parent];
                const float norm = score_norm ? sc / (float)(p->n + 1) : sc;
                if (fin.n < 0 || norm > fin_norm) {
                    fin.n = p->n; fin.score = sc; fin_norm = norm;
                    memcpy(fin.y, p->y, sizeof(int) * p->n);
                    memcpy(fin.steps, p->steps, sizeof(int) * p->n);
                }
            }
            /* the new beam */
            int n_nxt = 0;
            for (int j = 0; j < n_sel; ++j) {
                if (merge && dup[j]) continue;
                const hyp_t* p = &cur[sel[j].parent];
                hyp_t* q = &nxt[n_nxt++];
                q->n = p->n; q->score = sel[j].score;
                memcpy(q->y, p->y, sizeof(int) * p->n);
                memcpy(q->steps, p->steps, sizeof(int) * p->n);
                if (sel[j].tok < 0) {
                    memcpy(q->h, p->h, sizeof(float) * L * H);
                    memcpy(q->c, p->c, sizeof(float) * L * H);
                    memcpy(q->g, p->g, sizeof(float) * J);
                } else {
                    q->y[q->n] = sel[j].tok; q->steps[q->n] = i; q->n += 1;
                    const float* x = embed + (size_t)sel[j].tok * H;
                    for (int l = 0; l < L; ++l) {
                        rs_oracle_lstm_step(x, p->h + l * H, p->c + l * H, lstm_w[l], lstm_b[l], H, q->h + l * H, q->c + l * H);
                        x = q->h + l * H;
                    }
                    for (int jj = 0; jj < J; ++jj) q->g[jj] = rs_oracle_dot(q->h + (L - 1) * H, Wp + (size_t)jj * H, H) + bp[jj];
                }
            }
            hyp_t* tmp = cur; cur = nxt; nxt = tmp;
            n_cur = n_nxt;
        }
        const hyp_t* best = NULL;
        float best_score = 0.0f;
        if (fin.n >= 0) { best = &fin; best_score = fin.score; }
        else {                                                  /* nothing finished: the best of the beam */
            float bn = 0.0f;
            for (int s = 0; s < n_cur; ++s) {
                const float norm = score_norm ? cur[s].score / (float)(cur[s].n + 1) : cur[s].score;
                if (!best || norm > bn) { best = &cur[s]; bn = norm; best_score = cur[s].score; }
            }
        }
        int n = best->n;
        if (n > out_cap) { n = out_cap; overflow = 1; }
        for (int q = 0; q < n; ++q) { ids[(size_t)b * out_cap + q] = best->y[q]; steps[(size_t)b * out_cap + q] = best->steps[q]; }
        n_ids[b] = n;
        scores[b] = best_score;
        hyps_free(cur, beam); hyps_free(nxt, beam);
        free(fin.y); free(fin.steps);
    }
    free(z); free(hn); free(cn); free(cands); free(sel); free(sel_src); free(dup);
    return overflow ? -5 : 0;
}

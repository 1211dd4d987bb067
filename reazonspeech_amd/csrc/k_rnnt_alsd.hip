// k_rnnt_alsd.hip — alignment-length synchronous beam search (ALSD) over the RNN-T prediction and joint networks
// (SURVEY.md §8f "next" row 2: [UPSTREAM] BeamRNNTInfer.align_length_sync_decoding, the strategy the reference's
// post-processing is written for — pkg/nemo-asr/src/decode.py:29,38-41,48).
//
// Hypotheses of all utterances advance in lockstep over the alignment index i = t + u.  Hypothesis rows are
// r = utterance * beam + slot; the prediction-network and joint kernels of the greedy path (k_rnnt.hip) run over
// those rows unchanged (exact f32, fixed accumulation order), so every number here can be compared bit for bit
// with oracle/rnnt_alsd.c, which documents the evaluation order of the search itself (log-softmax reduction tree,
// expansion / selection / recombination order, where a finished hypothesis reads its score).
//
// One alignment step = 6 launches:
//   joint logits of the live rows (rnnt_tile_kernel<2>)         -> z [rows][Vpad]
//   alsd_select_kernel (one workgroup per utterance, one wave per beam slot)
//        log-softmax + top-`beam` tokens per live hypothesis, the `beam` best expansions, recombination,
//        finished-hypothesis bookkeeping; writes the new beam's labels / alignment steps / scores (ping-pong
//        buffers), each new row's parent row, and the work lists of the next step
//   alsd_reorder_kernel   new row <- parent row's prediction-network state (h, c, g), ping-pong
//   LSTM x L + joint.pred over the rows that took a token (rnnt_lstm4 / rnnt_pred16 or the wide variants)
// Compiled with -ffp-contract=off.
#include "k_rnnt_common.h"

int rs_rnnt_launch_lstm_pred(rs_ctx* ctx, const void* st_ptr, int rows, hipStream_t s);
int rs_rnnt_launch_joint_logits(rs_ctx* ctx, const void* st_ptr, const float* joint_enc, int rows, int tp_max, int rows_per_utt,
                                int step, hipStream_t s);

namespace {

constexpr int MAX_BEAM = 8;
constexpr int MAX_CAND = MAX_BEAM * (MAX_BEAM + 1);

// natural log, mirrored operation for operation in oracle/rnnt_math.h
__device__ __forceinline__ float rs_logf(float x) {
    unsigned u = __float_as_uint(x);
    int e = (int)(u >> 23) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float fe = (float)e;
    const float r = m - 1.0f;
    const float z = r * r;
    float p = 7.0376836292e-2f;
    p = fmaf(p, r, -1.1514610310e-1f);
    p = fmaf(p, r, 1.1676998740e-1f);
    p = fmaf(p, r, -1.2420140846e-1f);
    p = fmaf(p, r, 1.4249322787e-1f);
    p = fmaf(p, r, -1.6668057665e-1f);
    p = fmaf(p, r, 2.0000714765e-1f);
    p = fmaf(p, r, -2.4999993993e-1f);
    p = fmaf(p, r, 3.3333331174e-1f);
    float y = (p * r) * z;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    return fmaf(fe, 0.693359375f, r + y);
}
__device__ __forceinline__ float rs_logaddexpf(float a, float b) {
    const float hi = a >= b ? a : b, lo = a >= b ? b : a;
    return hi + rs_logf(1.0f + rs_expf(lo - hi));
}

struct AlsdState {
    // per hypothesis row; [2] = ping-pong: step i reads set (i & 1) and writes set ((i + 1) & 1)
    int32_t* len[2];     // [rows] labels emitted
    float* score[2];     // [rows]
    int32_t* y[2];       // [rows][cap] labels
    int32_t* al[2];      // [rows][cap] alignment index of each label
    int32_t* n_hyp[2];   // [B] hypotheses in the beam
    int32_t* parent;     // [rows] row whose prediction-network state the row written this step continues
    // finished hypotheses: the best one so far per utterance
    int32_t* fin_n;      // [B], -1 = none yet
    float* fin_norm;     // [B]
    float* fin_score;    // [B]
    int32_t* fin_y;      // [B][cap]
    int32_t* fin_al;     // [B][cap]
    int32_t* done;       // [B]
    int cap;
};

// label budget of an utterance with T frames (oracle/alsd.py: a float is a multiple of T, an int is absolute)
__host__ __device__ inline int alsd_budget(int T, double ratio, int abs_len) { return abs_len >= 0 ? abs_len : (int)(ratio * (double)T); }

__global__ void alsd_init_kernel(DecodeState st, AlsdState as, const int32_t* __restrict__ enc_lens, int B, int W, int blank,
                                 double ratio, int abs_len) {
    // single workgroup: the start hypothesis of every utterance and the first work lists, in row order
    __shared__ int n_alive_s;
    if (threadIdx.x == 0) n_alive_s = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int row = b * W;
        as.len[0][row] = 0; as.score[0][row] = 0.0f; as.n_hyp[0][b] = 1;
        as.fin_n[b] = -1; as.fin_norm[b] = 0.0f; as.fin_score[b] = 0.0f; as.done[b] = 0;
        st.token[row] = blank; st.tcur[row] = 0; st.act[b] = row;
        const int T = enc_lens[b];
        if (T > 0 && T + alsd_budget(T, ratio, abs_len) > 0) st.alive[atomicAdd(&n_alive_s, 1)] = row;
    }
    __syncthreads();
    // counters: [0] rows that took a token, [1] overflow flag, [2],[3] live rows of list 0 / 1, [4] finished utterances
    if (threadIdx.x == 0) { st.counters[0] = B; st.counters[1] = 0; st.counters[2] = n_alive_s; st.counters[3] = 0; st.counters[4] = 0; }
}

// label q of a candidate = its parent's label q, or the candidate's own token at the end
__device__ __forceinline__ int cand_len(int parent_len, int tok) { return parent_len + (tok >= 0 ? 1 : 0); }

// NVR > 0: the logits row of a hypothesis is read ONCE into NVR registers per lane (V <= 64 NVR), every load in flight before the
// first comparison; the scan for the maximum / the W best and the exp-sum then walk the registers in the same ascending order as
// the two memory passes of the NVR = 0 form — identical arithmetic, one memory round trip instead of two chains of 47.
template <int NVR>
__global__ __launch_bounds__(64 * MAX_BEAM) void alsd_select_kernel(
    DecodeState st, AlsdState as, const float* __restrict__ zbuf, int zstride, const int32_t* __restrict__ enc_lens, int B, int W,
    int V, int blank, int i, double ratio, int abs_len, int score_norm, int merge, int out_cap, int32_t* __restrict__ ids,
    int32_t* __restrict__ steps, int32_t* __restrict__ n_ids, float* __restrict__ scores) {
    const int b = blockIdx.x;
    if (as.done[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = i & 1, pn = p ^ 1;
    const int T = enc_lens[b];
    const int n_steps = T + alsd_budget(T, ratio, abs_len);
    const int n_cur = as.n_hyp[p][b];
    const int cap = as.cap;
    const int NC = W * (W + 1);

    __shared__ float c_score[MAX_CAND];
    __shared__ int c_tok[MAX_CAND];
    __shared__ int c_ok[MAX_CAND];
    __shared__ int s_sel[MAX_BEAM];        // selected candidate of beam position j
    __shared__ float s_score[MAX_BEAM];    // its score after recombination
    __shared__ int s_keep[MAX_BEAM];       // beam position written to new slot k
    __shared__ int s_nkeep, s_live;

    for (int c = tid; c < MAX_CAND; c += blockDim.x) c_ok[c] = 0;
    if (tid == 0) { s_live = 0; s_nkeep = 0; }
    __syncthreads();

    // ---- phase 1: one wave per hypothesis: log-softmax, blank and top-W token expansions -----------------------------
    {
        const int row = b * W + wave;
        bool live = false;
        if (i < n_steps && wave < n_cur) live = (i - as.len[p][row]) <= T - 1;
        if (live) {
            const float* zr = zbuf + (size_t)row * zstride;
            // ONE scan finds the row maximum AND each lane's own W best non-blank logits (sorted by (logit desc, index asc): a lane
            // meets its columns in ascending order, so a later equal logit never displaces an earlier one); the W best of the
            // row are then popped from the 64 sorted lists in W register rounds.  (The row used to be scanned once for the
            // maximum and once more per beam slot: 2 + W passes over 3001 logits, W of them only to find what this scan
            // already held in registers.)  The selection is the same total order as before: identical candidates.
            float m = -INFINITY;
            float tz[MAX_BEAM];
            int tv[MAX_BEAM];
#pragma unroll
            for (int k = 0; k < MAX_BEAM; ++k) { tz[k] = -INFINITY; tv[k] = -1; }
            auto scan = [&](int v, float zv) {
                if (zv > m) m = zv;
                if (v != blank) {
                    float cz = zv;
                    int cv = v;
                    bool ins = false;                                       // once placed, everything behind shifts down one slot
#pragma unroll
                    for (int k = 0; k < MAX_BEAM; ++k) {
                        if (k < W && (ins || tv[k] < 0 || cz > tz[k])) {   // an empty slot takes anything (also a -inf logit)
                            const float sz = tz[k]; const int sv = tv[k];
                            tz[k] = cz; tv[k] = cv;
                            cz = sz; cv = sv;
                            ins = true;
                        }
                    }
                }
            };
            float zreg[NVR > 0 ? NVR : 1];
            if constexpr (NVR > 0) {
#pragma unroll
                for (int q = 0; q < NVR; ++q) {
                    const int v = lane + 64 * q;
                    zreg[q] = zr[v < V ? v : V - 1];
                }
#pragma unroll
                for (int q = 0; q < NVR; ++q) {
                    const int v = lane + 64 * q;
                    if (v < V) scan(v, zreg[q]);
                }
            } else {
                for (int v = lane; v < V; v += 64) scan(v, zr[v]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off, 64); if (o > m) m = o; }
            float sum = 0.0f;
            if constexpr (NVR > 0) {
#pragma unroll
                for (int q = 0; q < NVR; ++q)
                    if (lane + 64 * q < V) sum = sum + rs_expf(zreg[q] - m);
            } else {
                for (int v = lane; v < V; v += 64) sum = sum + rs_expf(zr[v] - m);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum = sum + __shfl_xor(sum, off, 64);
            sum = __shfl(sum, 0, 64);
            const float lse = m + rs_logf(sum);
            const float hs = as.score[p][row];
            const int base = wave * (W + 1);
            if (lane == 0) { c_score[base] = hs + (zr[blank] - lse); c_tok[base] = -1; c_ok[base] = 1; s_live = 1; }
            for (int j = 0; j < W; ++j) {
                float bz = tz[0];
                int bv = tv[0];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float oz = __shfl_xor(bz, off, 64);
                    const int ov = __shfl_xor(bv, off, 64);
                    if (ov >= 0 && (bv < 0 || oz > bz || (oz == bz && ov < bv))) { bz = oz; bv = ov; }
                }
                if (bv < 0) break;
                if (lane == 0) { c_score[base + 1 + j] = hs + (bz - lse); c_tok[base + 1 + j] = bv; c_ok[base + 1 + j] = 1; }
                if (tv[0] == bv) {                                           // the owner pops its head
#pragma unroll
                    for (int k = 0; k + 1 < MAX_BEAM; ++k) { tz[k] = tz[k + 1]; tv[k] = tv[k + 1]; }
                    tz[MAX_BEAM - 1] = -INFINITY; tv[MAX_BEAM - 1] = -1;
                }
            }
        }
    }
    __syncthreads();

    if (!s_live) {
        // ---- the search of this utterance is over (alignment budget spent, or every hypothesis is past the last frame)
        if (wave == 0) {
            int src_n = as.fin_n[b];
            float src_score = as.fin_score[b];
            const int32_t *src_y = as.fin_y + (size_t)b * cap, *src_al = as.fin_al + (size_t)b * cap;
            if (src_n < 0) {   // nothing finished: the best of the beam
                float bn = 0.0f;
                for (int s = 0; s < n_cur; ++s) {
                    const int row = b * W + s;
                    const int n = as.len[p][row];
                    const float sc = as.score[p][row];
                    const float norm = score_norm ? sc / (float)(n + 1) : sc;
                    if (src_n < 0 || norm > bn) {
                        src_n = n; src_score = sc; bn = norm;
                        src_y = as.y[p] + (size_t)row * cap; src_al = as.al[p] + (size_t)row * cap;
                    }
                }
            }
            int n = src_n < 0 ? 0 : src_n;
            if (n > out_cap) { n = out_cap; if (lane == 0) st.counters[1] = 1; }
            for (int q = lane; q < n; q += 64) { ids[(size_t)b * out_cap + q] = src_y[q]; steps[(size_t)b * out_cap + q] = src_al[q]; }
            if (lane == 0) { n_ids[b] = n; scores[b] = src_score; as.done[b] = 1; atomicAdd(&st.counters[4], 1); }
        }
        return;
    }

    // ---- phase 2 (wave 0, every lane computes the same values): selection, recombination, finished hypotheses ---------
    if (wave == 0) {
        for (int c = lane; c < NC; c += 64) {
            if (!c_ok[c]) continue;
            const float sc = c_score[c];
            int rank = 0;
            for (int o = 0; o < NC; ++o)
                if (c_ok[o] && (c_score[o] > sc || (c_score[o] == sc && o < c))) ++rank;
            if (rank < W) s_sel[rank] = c;
        }
    }
    __syncthreads();
    if (wave == 0) {
        int n_valid = 0;
        for (int c = lane; c < NC; c += 64) n_valid += c_ok[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_xor(n_valid, off, 64);
        const int n_sel = n_valid < W ? n_valid : W;
        int sel[MAX_BEAM], stok[MAX_BEAM], splen[MAX_BEAM];
        float ssc[MAX_BEAM];
        const int32_t* sy[MAX_BEAM];
        unsigned dup = 0;
#pragma unroll
        for (int j = 0; j < MAX_BEAM; ++j) {
            sel[j] = 0; stok[j] = -1; splen[j] = 0; ssc[j] = 0.0f; sy[j] = as.y[p];
            if (j < n_sel) {
                sel[j] = s_sel[j];
                ssc[j] = c_score[sel[j]];
                stok[j] = c_tok[sel[j]];
                const int prow = b * W + sel[j] / (W + 1);
                splen[j] = as.len[p][prow];
                sy[j] = as.y[p] + (size_t)prow * cap;
            }
        }
        // recombination: equal label sequences add into the first occurrence
#pragma unroll
        for (int j = 1; j < MAX_BEAM; ++j) {
            if (j >= n_sel) continue;
            bool merged = false;
#pragma unroll
            for (int k = 0; k < j; ++k) {
                if (merged || ((dup >> k) & 1u)) continue;
                const int n = cand_len(splen[j], stok[j]);
                if (n != cand_len(splen[k], stok[k])) continue;
                bool differ = false;
                for (int q = lane; q < n; q += 64) {
                    const int a = q < splen[j] ? sy[j][q] : stok[j];
                    const int c = q < splen[k] ? sy[k][q] : stok[k];
                    differ = differ || (a != c);
                }
                if (__ballot(differ) != 0ull) continue;
                ssc[k] = rs_logaddexpf(ssc[k], ssc[j]);
                dup |= 1u << j;
                merged = true;
            }
        }
        // finished hypotheses: a blank taken at the last frame; the score is read after recombination
        int fn = as.fin_n[b];
        float fnorm = as.fin_norm[b], fscore = as.fin_score[b];
        for (int s = 0; s < n_cur; ++s) {
            const int c = s * (W + 1);
            if (!c_ok[c]) continue;
            const int row = b * W + s;
            const int n = as.len[p][row];
            if (i - n != T - 1) continue;
            float sc = c_score[c];
#pragma unroll
            for (int j = 0; j < MAX_BEAM; ++j) if (j < n_sel && sel[j] == c) sc = ssc[j];
            const float norm = score_norm ? sc / (float)(n + 1) : sc;
            if (fn < 0 || norm > fnorm) {
                fn = n; fnorm = norm; fscore = sc;
                for (int q = lane; q < n; q += 64) {
                    as.fin_y[(size_t)b * cap + q] = as.y[p][(size_t)row * cap + q];
                    as.fin_al[(size_t)b * cap + q] = as.al[p][(size_t)row * cap + q];
                }
            }
        }
        if (lane == 0) { as.fin_n[b] = fn; as.fin_norm[b] = fnorm; as.fin_score[b] = fscore; }
        // the new beam, in selection order ("merge" drops the duplicates)
        int nk = 0;
#pragma unroll
        for (int j = 0; j < MAX_BEAM; ++j) {
            if (j >= n_sel || (merge && ((dup >> j) & 1u))) continue;
            if (lane == 0) { s_keep[nk] = sel[j]; s_score[nk] = ssc[j]; }
            ++nk;
        }
        if (lane == 0) { s_nkeep = nk; as.n_hyp[pn][b] = nk; }
    }
    __syncthreads();

    // ---- phase 3: one wave per new beam slot: labels and alignment steps of the parent (+ the token taken) -------------
    if (wave < s_nkeep) {
        const int c = s_keep[wave];
        const int tok = c_tok[c];
        const int prow = b * W + c / (W + 1), row = b * W + wave;
        const int n = as.len[p][prow];
        const int32_t *py = as.y[p] + (size_t)prow * cap, *pal = as.al[p] + (size_t)prow * cap;
        int32_t *ny = as.y[pn] + (size_t)row * cap, *nal = as.al[pn] + (size_t)row * cap;
        for (int q = lane; q < n; q += 64) { ny[q] = py[q]; nal[q] = pal[q]; }
        if (lane == 0) {
            int nn = n;
            if (tok >= 0) {
                if (n < cap) { ny[n] = tok; nal[n] = i; nn = n + 1; }
                else st.counters[1] = 1;
                st.token[row] = tok;
                st.act[atomicAdd(&st.counters[0], 1)] = row;
            }
            as.len[pn][row] = nn;
            as.score[pn][row] = s_score[wave];
            as.parent[row] = prow;
            const int tn = (i + 1) - nn;
            st.tcur[row] = tn;
            if (i + 1 < n_steps && tn <= T - 1) {
                const int pos = atomicAdd(&st.counters[2 + ((i + 1) & 1)], 1);
                st.alive[(size_t)((i + 1) & 1) * (B * W) + pos] = row;
            }
        }
    }
}

// new row <- parent row's prediction-network state; src/dst are the two state sets
__global__ __launch_bounds__(256) void alsd_reorder_kernel(AlsdState as, int pn, int W, int rows, int L, int H, int J,
                                                           const float* __restrict__ h_src, const float* __restrict__ c_src,
                                                           const float* __restrict__ g_src, float* __restrict__ h_dst,
                                                           float* __restrict__ c_dst, float* __restrict__ g_dst) {
    const int row = blockIdx.x, b = row / W, slot = row % W;
    if (as.done[b] || slot >= as.n_hyp[pn][b]) return;
    const int prow = as.parent[row];
    for (int l = 0; l < L; ++l) {
        const size_t so = ((size_t)l * rows + prow) * H, dof = ((size_t)l * rows + row) * H;
        for (int k = 4 * threadIdx.x; k < H; k += 4 * blockDim.x) {
            *reinterpret_cast<float4*>(h_dst + dof + k) = *reinterpret_cast<const float4*>(h_src + so + k);
            *reinterpret_cast<float4*>(c_dst + dof + k) = *reinterpret_cast<const float4*>(c_src + so + k);
        }
    }
    for (int k = 4 * threadIdx.x; k < J; k += 4 * blockDim.x)
        *reinterpret_cast<float4*>(g_dst + (size_t)row * J + k) = *reinterpret_cast<const float4*>(g_src + (size_t)prow * J + k);
}

struct AlsdPlan {
    size_t state, g, rows4, rows_cap4, b4, b_cap4, z, total;
};

AlsdPlan alsd_plan(const rs_ctx* ctx, int B, int W, int cap) {
    const rs_dims& d = ctx->d;
    const size_t rows = (size_t)B * W;
    AlsdPlan p;
    p.state = rs_align((size_t)d.pred_layers * rows * d.pred_hidden * 4);
    p.g = rs_align(rows * d.joint_hidden * 4);
    p.rows4 = rs_align(rows * 4);
    p.rows_cap4 = rs_align(rows * (size_t)cap * 4);
    p.b4 = rs_align((size_t)B * 4);
    p.b_cap4 = rs_align((size_t)B * cap * 4);
    p.z = rs_align(rows * (size_t)((d.n_logits + 63) / 64 * 64) * 4);
    //        h,c x2 sets + h_tmp,c_tmp   g x2   tcur sym token act parent len x2 score x2   alive x2   y,al x2
    p.total = 6 * p.state + 2 * p.g + 9 * p.rows4 + 2 * p.rows4 + 4 * p.rows_cap4 +
              //  n_hyp x2, fin_n, fin_norm, fin_score, done    fin_y, fin_al    counters   z
              6 * p.b4 + 2 * p.b_cap4 + rs_align(64) + p.z + 1024;
    return p;
}

}  // namespace

size_t rs_rnnt_alsd_workspace_bytes_impl(const rs_ctx* ctx, int B, int beam, int cap) {
    if (B <= 0 || beam <= 0 || cap <= 0) return 0;
    return alsd_plan(ctx, B, beam, cap).total;
}

int rs_rnnt_alsd_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam, double ratio,
                      int abs_len, int score_norm, int merge, int out_cap, int32_t* ids, int32_t* steps, int32_t* n_ids,
                      float* scores, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int L = d.pred_layers, H = d.pred_hidden, J = d.joint_hidden, V = d.n_logits;
    if (B <= 0) return RS_OK;
    if (H % 128 || J % 128) return rs_fail(ctx, RS_EINVAL, "alsd: pred_hidden/joint_hidden must be multiples of 128");
    if (L < 1 || L > 4) return rs_fail(ctx, RS_EINVAL, "alsd: 1..4 LSTM layers supported");
    const int W = beam < V - 1 ? beam : V - 1;
    if (W < 1 || W > MAX_BEAM) return rs_fail(ctx, RS_EINVAL, "alsd: beam size must be 1..%d", MAX_BEAM);
    const int max_steps = tp_max + alsd_budget(tp_max, ratio, abs_len);      // >= every utterance's alignment length
    const int cap = max_steps > 0 ? max_steps : 1;
    const AlsdPlan pl = alsd_plan(ctx, B, W, cap);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "alsd: workspace %zu < %zu", workspace_bytes, pl.total);
    const int rows = B * W;
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* q = w; w += bytes; return q; };
    DecodeState st[2];
    AlsdState as;
    float* hset[2]; float* cset[2]; float* gset[2];
    hset[0] = (float*)take(pl.state); cset[0] = (float*)take(pl.state);      // adjacent: one memset
    hset[1] = (float*)take(pl.state); cset[1] = (float*)take(pl.state);
    float* h_tmp = (float*)take(pl.state); float* c_tmp = (float*)take(pl.state);
    gset[0] = (float*)take(pl.g); gset[1] = (float*)take(pl.g);
    int32_t* tcur = (int32_t*)take(pl.rows4); int32_t* sym = (int32_t*)take(pl.rows4);
    int32_t* token = (int32_t*)take(pl.rows4); int32_t* act = (int32_t*)take(pl.rows4);
    as.parent = (int32_t*)take(pl.rows4);
    as.len[0] = (int32_t*)take(pl.rows4); as.len[1] = (int32_t*)take(pl.rows4);
    as.score[0] = (float*)take(pl.rows4); as.score[1] = (float*)take(pl.rows4);
    int32_t* alive = (int32_t*)take(2 * pl.rows4);
    as.y[0] = (int32_t*)take(pl.rows_cap4); as.y[1] = (int32_t*)take(pl.rows_cap4);
    as.al[0] = (int32_t*)take(pl.rows_cap4); as.al[1] = (int32_t*)take(pl.rows_cap4);
    as.n_hyp[0] = (int32_t*)take(pl.b4); as.n_hyp[1] = (int32_t*)take(pl.b4);
    as.fin_n = (int32_t*)take(pl.b4); as.fin_norm = (float*)take(pl.b4); as.fin_score = (float*)take(pl.b4);
    as.done = (int32_t*)take(pl.b4);
    as.fin_y = (int32_t*)take(pl.b_cap4); as.fin_al = (int32_t*)take(pl.b_cap4);
    int32_t* counters = (int32_t*)take(rs_align(64));
    float* zbuf = (float*)take(pl.z);
    as.cap = cap;
    for (int k = 0; k < 2; ++k) {
        DecodeState& q = st[k];
        q.h = hset[k]; q.c = cset[k]; q.h_tmp = h_tmp; q.c_tmp = c_tmp; q.g = gset[k];
        q.tcur = tcur; q.sym = sym; q.token = token; q.act = act; q.alive = alive; q.counters = counters;
        q.pmax = nullptr; q.pidx = nullptr; q.a16 = nullptr; q.anorm = nullptr; q.zapprox = zbuf;
        q.g_off = nullptr; q.a_pre = nullptr;
        q.joint_act = d.joint_act;
    }
    const int zstride = (V + 63) / 64 * 64;

    rs_prof_begin(ctx, RS_PROF_DECODE, s, 0.0, 0.0);
    RS_HIP(ctx, hipMemsetAsync(hset[0], 0, 2 * pl.state, s));
    hipLaunchKernelGGL(alsd_init_kernel, dim3(1), dim3(256), 0, s, st[0], as, enc_lens, B, W, d.blank_id, ratio, abs_len);
    // start of sequence: blank token from zero state, slot 0 of every utterance (list built by the init kernel)
    if (int rc = rs_rnnt_launch_lstm_pred(ctx, &st[0], rows, s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
    RS_CHECK_LAUNCH(ctx, "alsd init");

    const int CHUNK = 16;
    int32_t hc[8] = {0};
    bool finished = false;
    int i = 0;
    while (!finished && i <= max_steps) {
        for (int c = 0; c < CHUNK && i <= max_steps; ++c, ++i) {
            const int p = i & 1, pn = p ^ 1;
            if (int rc = rs_rnnt_launch_joint_logits(ctx, &st[p], joint_enc, rows, tp_max, W, i, s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
            if (V <= 64 * 48) hipLaunchKernelGGL((alsd_select_kernel<48>), dim3(B), dim3(64 * W), 0, s, st[p], as, zbuf, zstride, enc_lens, B, W, V,
                               d.blank_id, i, ratio, abs_len, score_norm, merge, out_cap, ids, steps, n_ids, scores);
            else hipLaunchKernelGGL((alsd_select_kernel<0>), dim3(B), dim3(64 * W), 0, s, st[p], as, zbuf, zstride, enc_lens, B, W, V,
                               d.blank_id, i, ratio, abs_len, score_norm, merge, out_cap, ids, steps, n_ids, scores);
            hipLaunchKernelGGL(alsd_reorder_kernel, dim3(rows), dim3(256), 0, s, as, pn, W, rows, L, H, J, hset[p], cset[p],
                               gset[p], hset[pn], cset[pn], gset[pn]);
            if (int rc = rs_rnnt_launch_lstm_pred(ctx, &st[pn], rows, s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
        }
        RS_CHECK_LAUNCH(ctx, "alsd step");
        RS_HIP(ctx, hipMemcpyAsync(hc, counters, sizeof hc, hipMemcpyDeviceToHost, s));
        RS_HIP(ctx, hipStreamSynchronize(s));
        finished = hc[4] >= B;
    }
    rs_prof_end(ctx, RS_PROF_DECODE, s);
    if (!finished) return rs_fail(ctx, RS_ESTATE, "alsd: %d of %d utterances unfinished after %d alignment steps", B - hc[4], B, max_steps + 1);
    if (hc[1]) return rs_fail(ctx, RS_EOVERFLOW, "alsd: a hypothesis has more than out_cap=%d labels", out_cap);
    return RS_OK;
}

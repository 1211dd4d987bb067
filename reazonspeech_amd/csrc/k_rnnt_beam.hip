// k_rnnt_beam.hip — the "default" transducer beam search (Graves 2012) as ESPnet2 implements it: [UPSTREAM]
// BeamSearchTransducer.default_beam_search + sort_nbest.  It is the decode reazonspeech.espnet.asr runs: the reference builds
// Speech2Text with its defaults — beam_size 20, search_type "default", score_norm, nbest 1, no LM
// (pkg/espnet-asr/src/transcribe.py:27-31).
//
// The search is frame-synchronous per utterance but the number of prediction-network evaluations ("pops") a frame takes is
// data dependent, so utterances are NOT kept in frame lockstep: every utterance is its own state machine and one device
// iteration performs ONE pop for every utterance that is still searching, whatever frame it is at:
//
//   beam_pop_kernel      (one wave per utterance) first maximum of the open list `hyps`; the popped hypothesis' sequence enters
//                        the label trie; its stored prediction-net state (the state BEFORE its last label) and that label are
//                        placed in row b of the decode state; the utterance joins this iteration's work lists
//   LSTM x L + joint.pred, joint logits of frame t_b   the exact-f32 kernels of the greedy path (k_rnnt.hip) over those rows
//   beam_expand_kernel   (one workgroup per utterance) log-softmax, the blank extension -> `kept`, the beam_k best labels ->
//                        `hyps` (with the state AFTER the popped hypothesis' last label, parked in a state-pool slot), the
//                        end-of-frame test (>= beam entries of kept above max(hyps)), and at the end of a frame: survivors sorted
//                        ascending, their states compacted into the other pool, t += 1; at the last frame the winner by
//                        score / len(yseq) is read back through the trie
//
// Evaluation order (float32 sums, log-sum-exp tree, tie rules) is documented in oracle/espnet_beam.c and the results are
// bit-identical to it: labels, scores and the pop count.  Compiled with -ffp-contract=off.
#include "k_rnnt_common.h"

int rs_rnnt_launch_lstm_pred(rs_ctx* ctx, const void* st_ptr, int rows, hipStream_t s);
int rs_rnnt_launch_joint_logits(rs_ctx* ctx, const void* st_ptr, const float* joint_enc, int rows, int tp_max, int rows_per_utt,
                                int step, hipStream_t s);

namespace {

// natural log, mirrored operation for operation in oracle/rnnt_math.h (the same routine as k_rnnt_alsd.hip)
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

struct BeamState {
    // per utterance [B]
    int32_t* t;          // frame being searched
    int32_t* done;
    int32_t* nh;         // entries of hyps (dead ones included)
    int32_t* nk;         // entries of kept
    int32_t* npop;       // pops of the current frame
    int32_t* ninit;      // hypotheses the current frame started with (state slots [0, ninit) of the current pool)
    int32_t* pool;       // current state pool (0 / 1)
    int32_t* nnode;      // trie nodes in use
    int32_t* pops;       // pops over the whole utterance (the work measure, returned)
    // the hypothesis popped in this iteration [B]
    float* cur_score;
    int32_t* cur_node;
    int32_t* cur_state;
    int32_t* cur_len;
    // open list [B][max_h]
    float* h_score;
    int32_t* h_node;     // trie node of the sequence when h_tok < 0, of the sequence without its last label otherwise
    int32_t* h_tok;
    int32_t* h_state;    // pool slot of the state BEFORE the last label
    int32_t* h_len;      // len(yseq): labels + the leading blank
    int32_t* h_alive;
    // blank extensions of this frame [B][max_pops]
    float* k_score;
    int32_t* k_node;
    int32_t* k_state;
    int32_t* k_len;
    int2* nodes;         // [B][max_nodes] (parent, label)
    float* states;       // [2][B][slots][2 * L * H]  (h then c)
    int32_t* flags;      // [0] utterances done, [1] overflow
    int max_h, max_pops, max_nodes, slots;
};

__device__ __forceinline__ void beam_fail(const BeamState& bs, int b) {   // one thread
    bs.done[b] = 1;
    bs.flags[1] = 1;
    atomicAdd(&bs.flags[0], 1);
}

// grid ceil(B / 256)
__global__ __launch_bounds__(256) void beam_init_kernel(BeamState bs, DecodeState st, const int32_t* __restrict__ enc_lens, int B,
                                                        int blank, int32_t* __restrict__ n_ids, float* __restrict__ scores,
                                                        int32_t* __restrict__ pops) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    bs.t[b] = 0; bs.nk[b] = 0; bs.npop[b] = 0; bs.ninit[b] = 1; bs.pool[b] = 0; bs.pops[b] = 0;
    bs.nodes[(size_t)b * bs.max_nodes] = make_int2(-1, blank);
    bs.nnode[b] = 1;
    const size_t h0 = (size_t)b * bs.max_h;
    bs.h_score[h0] = 0.0f; bs.h_node[h0] = 0; bs.h_tok[h0] = -1; bs.h_state[h0] = 0; bs.h_len[h0] = 1; bs.h_alive[h0] = 1;
    bs.nh[b] = 1;
    st.token[b] = blank; st.tcur[b] = 0;
    n_ids[b] = 0; scores[b] = 0.0f; pops[b] = 0;
    const int fin = enc_lens[b] <= 0;            // nothing to search: the empty hypothesis, score 0
    bs.done[b] = fin;
    if (fin) atomicAdd(&bs.flags[0], 1);
}

// grid B, block 64
__global__ __launch_bounds__(64) void beam_pop_kernel(BeamState bs, DecodeState st, int B, int L, int H) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (bs.done[b]) return;
    const int n = bs.nh[b];
    const size_t hb = (size_t)b * bs.max_h;
    float best = -INFINITY;
    int bi = -1;
    for (int i = lane; i < n; i += 64)
        if (bs.h_alive[hb + i]) {
            const float s = bs.h_score[hb + i];
            if (bi < 0 || s > best) { best = s; bi = i; }          // a lane meets its entries in ascending order
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (oi >= 0 && (bi < 0 || os > best || (os == best && oi < bi))) { best = os; bi = oi; }
    }
    int node = bs.h_node[hb + bi];
    const int tok = bs.h_tok[hb + bi];
    const int state = bs.h_state[hb + bi];
    int ok = 1;
    if (lane == 0) {
        if (bs.npop[b] >= bs.max_pops || (tok >= 0 && bs.nnode[b] >= bs.max_nodes)) { beam_fail(bs, b); ok = 0; }
    }
    ok = __shfl(ok, 0, 64);
    if (!ok) return;
    int last = tok;
    if (lane == 0) {
        bs.h_alive[hb + bi] = 0;
        if (tok >= 0) {                                            // the sequence enters the trie now
            const int nn = bs.nnode[b];
            bs.nodes[(size_t)b * bs.max_nodes + nn] = make_int2(node, tok);
            node = nn;
            bs.nnode[b] = nn + 1;
        } else {
            last = bs.nodes[(size_t)b * bs.max_nodes + node].y;
        }
        bs.cur_score[b] = best; bs.cur_node[b] = node; bs.cur_state[b] = state; bs.cur_len[b] = bs.h_len[hb + bi];
        st.token[b] = last;
        st.tcur[b] = bs.t[b];
        st.act[atomicAdd(&st.counters[0], 1)] = b;                 // LSTM / joint.pred work list
        st.alive[atomicAdd(&st.counters[2], 1)] = b;               // joint-logits work list (list 0: the joint runs with step 0)
    }
    const int LH = L * H;
    const float* src = bs.states + (((size_t)bs.pool[b] * B + b) * bs.slots + state) * (2 * (size_t)LH);
    for (int i = lane; i < LH; i += 64) {
        const int l = i / H, u = i - l * H;
        st.h[((size_t)l * B + b) * H + u] = src[i];
        st.c[((size_t)l * B + b) * H + u] = src[LH + i];
    }
}

// grid B, block 256, dynamic LDS: z row [V] + kept scores [max_pops] + survivor slots [max_pops]
__global__ __launch_bounds__(256) void beam_expand_kernel(BeamState bs, DecodeState st, const float* __restrict__ zbuf, int zstride,
                                                          const int32_t* __restrict__ enc_lens, int B, int L, int H, int V, int blank,
                                                          int beam, int beam_k, int score_norm, int out_cap,
                                                          int32_t* __restrict__ ids, int32_t* __restrict__ n_ids,
                                                          float* __restrict__ scores, int32_t* __restrict__ pops) {
    extern __shared__ __attribute__((aligned(16))) char beam_smem[];
    float* zs = reinterpret_cast<float*>(beam_smem);
    float* ks = zs + zstride;                                        // kept scores
    int* kslot = reinterpret_cast<int*>(ks + bs.max_pops);           // rank -> source pool slot
    __shared__ float w_f[2][4];
    __shared__ int w_i[2][4];
    __shared__ float s_lse, s_hmax;
    __shared__ int s_good, s_best;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (b == 0 && tid == 0) st.counters[2] = 0;                      // the next pop rebuilds list 0 (counters[0] was zeroed by the joint)
    if (bs.done[b]) return;
    const int LH = L * H;
    const int pool = bs.pool[b];
    float* pool_cur = bs.states + (((size_t)pool * B + b) * bs.slots) * (2 * (size_t)LH);
    float* pool_nxt = bs.states + (((size_t)(pool ^ 1) * B + b) * bs.slots) * (2 * (size_t)LH);
    const int npop = bs.npop[b], ninit = bs.ninit[b], nk = bs.nk[b], nh0 = bs.nh[b];
    const int after = ninit + npop;                                  // < slots: ninit <= max_pops, npop < max_pops
    // park the state after the popped hypothesis' last label; stage the logits
    for (int i = tid; i < LH; i += 256) {
        const int l = i / H, u = i - l * H;
        pool_cur[(size_t)after * 2 * LH + i] = st.h[((size_t)l * B + b) * H + u];
        pool_cur[(size_t)after * 2 * LH + LH + i] = st.c[((size_t)l * B + b) * H + u];
    }
    const float* zr = zbuf + (size_t)b * zstride;
    for (int v = tid; v < V; v += 256) zs[v] = zr[v];
    __syncthreads();
    if (wave == 0) {                                                 // log-sum-exp in the documented order
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) { const float zv = zs[v]; if (zv > m) m = zv; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off, 64); if (o > m) m = o; }
        float sum = 0.0f;
        for (int v = lane; v < V; v += 64) sum = sum + rs_expf(zs[v] - m);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum = sum + __shfl_xor(sum, off, 64);
        if (lane == 0) s_lse = m + rs_logf(sum);
    }
    __syncthreads();
    const float lse = s_lse;
    const float hs = bs.cur_score[b];
    const int cnode = bs.cur_node[b], clen = bs.cur_len[b];
    const size_t hb = (size_t)b * bs.max_h, kb = (size_t)b * bs.max_pops;
    if (tid == 0) {
        bs.k_score[kb + nk] = hs + (zs[blank] - lse);
        bs.k_node[kb + nk] = cnode; bs.k_state[kb + nk] = bs.cur_state[b]; bs.k_len[kb + nk] = clen;
    }
    // the beam_k best labels by (logit desc, index asc), one per round: every round takes the best entry that comes strictly
    // after the previous pick in that order
    float pz = INFINITY;
    int pv = -1, n_child = 0;
    for (int j = 0; j < beam_k; ++j) {
        float bz = -INFINITY;
        int bv = -1;
        for (int v = tid; v < V; v += 256) {
            if (v == blank) continue;
            const float zv = zs[v];
            if (!(zv < pz || (zv == pz && v > pv))) continue;
            if (bv < 0 || zv > bz) { bz = zv; bv = v; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float oz = __shfl_xor(bz, off, 64);
            const int ov = __shfl_xor(bv, off, 64);
            if (ov >= 0 && (bv < 0 || oz > bz || (oz == bz && ov < bv))) { bz = oz; bv = ov; }
        }
        if (lane == 0) { w_f[j & 1][wave] = bz; w_i[j & 1][wave] = bv; }
        __syncthreads();
        bz = w_f[j & 1][0]; bv = w_i[j & 1][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float oz = w_f[j & 1][w];
            const int ov = w_i[j & 1][w];
            if (ov >= 0 && (bv < 0 || oz > bz || (oz == bz && ov < bv))) { bz = oz; bv = ov; }
        }
        if (bv < 0) break;
        if (tid == 0) {
            const size_t o = hb + nh0 + j;
            bs.h_score[o] = hs + (bz - lse);
            bs.h_node[o] = cnode; bs.h_tok[o] = bv; bs.h_state[o] = after; bs.h_len[o] = clen + 1; bs.h_alive[o] = 1;
        }
        pz = bz; pv = bv;
        ++n_child;
    }
    const int nh = nh0 + n_child, nkk = nk + 1;
    __syncthreads();                                                 // thread 0's entries are visible to the block below
    // end-of-frame test: at least `beam` kept entries strictly above the maximum of the open list
    float hm = -INFINITY;
    for (int i = tid; i < nh; i += 256)
        if (bs.h_alive[hb + i]) { const float s = bs.h_score[hb + i]; if (s > hm) hm = s; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(hm, off, 64); if (o > hm) hm = o; }
    if (lane == 0) w_f[0][wave] = hm;
    for (int i = tid; i < nkk; i += 256) ks[i] = bs.k_score[kb + i];
    __syncthreads();
    hm = fmaxf(fmaxf(w_f[0][0], w_f[0][1]), fmaxf(w_f[0][2], w_f[0][3]));
    if (tid == 0) { s_good = 0; s_hmax = hm; }
    __syncthreads();
    int good = 0;
    for (int i = tid; i < nkk; i += 256) good += ks[i] > hm;
    if (good) atomicAdd(&s_good, good);
    __syncthreads();
    const int n_good = s_good;
    if (n_good < beam) {                                             // the frame goes on
        if (tid == 0) { bs.nh[b] = nh; bs.nk[b] = nkk; bs.npop[b] = npop + 1; bs.pops[b] += 1; }
        return;
    }
    // ---- end of frame: survivors ascending by score (ties in kept order) become the next frame's open list ----
    for (int i = tid; i < nkk; i += 256) {
        const float si = ks[i];
        if (!(si > hm)) continue;
        int rank = 0;
        for (int o = 0; o < nkk; ++o) {
            const float so = ks[o];
            if (so > hm && (so < si || (so == si && o < i))) ++rank;
        }
        bs.h_score[hb + rank] = si;
        bs.h_node[hb + rank] = bs.k_node[kb + i]; bs.h_tok[hb + rank] = -1; bs.h_state[hb + rank] = rank;
        bs.h_len[hb + rank] = bs.k_len[kb + i]; bs.h_alive[hb + rank] = 1;
        kslot[rank] = bs.k_state[kb + i];
    }
    __syncthreads();
    const int t_next = bs.t[b] + 1;
    const bool last = t_next >= enc_lens[b];
    if (!last) {
        for (int r = 0; r < n_good; ++r) {
            const float* src = pool_cur + (size_t)kslot[r] * 2 * LH;
            float* dst = pool_nxt + (size_t)r * 2 * LH;
            for (int i = tid; i < 2 * LH; i += 256) dst[i] = src[i];
        }
        if (tid == 0) {
            bs.nh[b] = n_good; bs.nk[b] = 0; bs.npop[b] = 0; bs.ninit[b] = n_good; bs.pool[b] = pool ^ 1; bs.t[b] = t_next;
            bs.pops[b] += 1;
        }
        return;
    }
    // ---- last frame: the first maximum of score / len(yseq) (or of score) over the survivors in their order ----
    if (wave == 0) {
        float bn = -INFINITY;
        int br = -1;
        for (int r = lane; r < n_good; r += 64) {
            const float sc = bs.h_score[hb + r];
            const float norm = score_norm ? sc / (float)bs.h_len[hb + r] : sc;
            if (br < 0 || norm > bn) { bn = norm; br = r; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float on = __shfl_xor(bn, off, 64);
            const int orr = __shfl_xor(br, off, 64);
            if (orr >= 0 && (br < 0 || on > bn || (on == bn && orr < br))) { bn = on; br = orr; }
        }
        if (lane == 0) s_best = br;
    }
    __syncthreads();
    if (tid == 0) {
        const int br = s_best;
        const int n = bs.h_len[hb + br] - 1;
        scores[b] = bs.h_score[hb + br];
        pops[b] = bs.pops[b] + 1;
        bs.done[b] = 1;
        if (n > out_cap) { n_ids[b] = 0; bs.flags[1] = 1; }
        else {
            int node = bs.h_node[hb + br];
            for (int q = n - 1; q >= 0; --q) {
                const int2 nd = bs.nodes[(size_t)b * bs.max_nodes + node];
                ids[(size_t)b * out_cap + q] = nd.y;
                node = nd.x;
            }
            n_ids[b] = n;
        }
        atomicAdd(&bs.flags[0], 1);
    }
}

struct BeamPlan {
    size_t b4, h4, k4, nodes, states, state1, g, rows4, z, total;
    int max_h, max_nodes, slots, zstride;
};

BeamPlan beam_plan(const rs_ctx* ctx, int B, int beam_k, int tp_max, int max_pops) {
    const rs_dims& d = ctx->d;
    BeamPlan p;
    p.max_h = max_pops * (beam_k + 1) + 1;
    p.max_nodes = (tp_max > 0 ? tp_max : 1) * max_pops + 1;        // a pop adds at most one node
    p.slots = 2 * max_pops;
    p.zstride = (d.n_logits + 63) / 64 * 64;
    p.b4 = rs_align((size_t)B * 4);
    p.h4 = rs_align((size_t)B * p.max_h * 4);
    p.k4 = rs_align((size_t)B * max_pops * 4);
    p.nodes = rs_align((size_t)B * p.max_nodes * 8);
    p.state1 = rs_align((size_t)d.pred_layers * B * d.pred_hidden * 4);
    p.states = rs_align((size_t)2 * B * p.slots * 2 * d.pred_layers * d.pred_hidden * 4);
    p.g = rs_align((size_t)B * d.joint_hidden * 4);
    p.rows4 = rs_align((size_t)B * 4);
    p.z = rs_align((size_t)B * p.zstride * 4);
    p.total = 13 * p.b4 + 6 * p.h4 + 4 * p.k4 + p.nodes + p.states + 4 * p.state1 + p.g + 6 * p.rows4 + 2 * rs_align(64) + p.z + 1024;
    return p;
}

int clamp_pops(int beam, int max_pops) { return max_pops > 0 ? max_pops : 16 * beam; }

}  // namespace

size_t rs_rnnt_beam_workspace_bytes_impl(const rs_ctx* ctx, int B, int beam, int tp_max, int max_pops) {
    const int V = ctx->d.n_logits;
    const int bm = beam < V ? beam : V, beam_k = bm < V - 1 ? bm : V - 1;
    return beam_plan(ctx, B, beam_k, tp_max, clamp_pops(bm, max_pops)).total;
}

int rs_rnnt_beam_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam, int score_norm,
                      int max_pops, int out_cap, int32_t* ids, int32_t* n_ids, float* scores, int32_t* pops, void* workspace,
                      size_t workspace_bytes, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int L = d.pred_layers, H = d.pred_hidden, J = d.joint_hidden, V = d.n_logits;
    if (H % 128 || J % 128) return rs_fail(ctx, RS_EINVAL, "beam search: pred_hidden/joint_hidden must be multiples of 128");
    if (L < 1 || L > 4) return rs_fail(ctx, RS_EINVAL, "beam search: 1..4 LSTM layers supported");
    if (V < 2) return rs_fail(ctx, RS_EINVAL, "beam search: vocabulary of %d", V);
    const int bm = beam < V ? beam : V, beam_k = bm < V - 1 ? bm : V - 1;
    const int mp = clamp_pops(bm, max_pops);
    if (mp < bm) return rs_fail(ctx, RS_EINVAL, "beam search: max_pops %d < beam %d (a frame needs at least `beam` pops)", mp, bm);
    const BeamPlan pl = beam_plan(ctx, B, beam_k, tp_max, mp);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "beam search: workspace %zu < %zu", workspace_bytes, pl.total);
    const size_t lds = (size_t)pl.zstride * 4 + (size_t)mp * 8;
    if (lds > 60 * 1024) return rs_fail(ctx, RS_EINVAL, "beam search: vocabulary %d / max_pops %d exceed the expand kernel's LDS", V, mp);
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* q = w; w += bytes; return q; };
    BeamState bs;
    DecodeState st;
    // every int32 / float bookkeeping array first (one memset), then the big buffers
    char* zero_from = w;
    bs.t = (int32_t*)take(pl.b4); bs.done = (int32_t*)take(pl.b4); bs.nh = (int32_t*)take(pl.b4); bs.nk = (int32_t*)take(pl.b4);
    bs.npop = (int32_t*)take(pl.b4); bs.ninit = (int32_t*)take(pl.b4); bs.pool = (int32_t*)take(pl.b4);
    bs.nnode = (int32_t*)take(pl.b4); bs.pops = (int32_t*)take(pl.b4);
    bs.cur_score = (float*)take(pl.b4); bs.cur_node = (int32_t*)take(pl.b4); bs.cur_state = (int32_t*)take(pl.b4);
    bs.cur_len = (int32_t*)take(pl.b4);
    bs.flags = (int32_t*)take(rs_align(64));
    int32_t* counters = (int32_t*)take(rs_align(64));
    const size_t zero_bytes = (size_t)(w - zero_from);
    bs.h_score = (float*)take(pl.h4); bs.h_node = (int32_t*)take(pl.h4); bs.h_tok = (int32_t*)take(pl.h4);
    bs.h_state = (int32_t*)take(pl.h4); bs.h_len = (int32_t*)take(pl.h4); bs.h_alive = (int32_t*)take(pl.h4);
    bs.k_score = (float*)take(pl.k4); bs.k_node = (int32_t*)take(pl.k4); bs.k_state = (int32_t*)take(pl.k4);
    bs.k_len = (int32_t*)take(pl.k4);
    bs.nodes = (int2*)take(pl.nodes);
    bs.states = (float*)take(pl.states);
    bs.max_h = pl.max_h; bs.max_pops = mp; bs.max_nodes = pl.max_nodes; bs.slots = pl.slots;
    st.h = (float*)take(pl.state1); st.c = (float*)take(pl.state1);
    st.h_tmp = (float*)take(pl.state1); st.c_tmp = (float*)take(pl.state1);
    st.g = (float*)take(pl.g);
    st.tcur = (int32_t*)take(pl.rows4); st.sym = (int32_t*)take(pl.rows4); st.token = (int32_t*)take(pl.rows4);
    st.act = (int32_t*)take(pl.rows4);
    st.alive = (int32_t*)take(2 * pl.rows4);
    st.counters = counters;
    st.pmax = nullptr; st.pidx = nullptr; st.a16 = nullptr; st.anorm = nullptr;
    st.zapprox = (float*)take(pl.z);
    st.joint_act = d.joint_act;

    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)beam_expand_kernel, (int)lds); rc != RS_OK) return rc;
    rs_prof_begin(ctx, RS_PROF_DECODE, s, 0.0, 0.0);
    RS_HIP(ctx, hipMemsetAsync(zero_from, 0, zero_bytes, s));
    // slot 0 of pool 0 of every utterance: the zero state the search starts from
    RS_HIP(ctx, hipMemset2DAsync(bs.states, (size_t)pl.slots * 2 * L * H * 4, 0, (size_t)2 * L * H * 4, B, s));
    hipLaunchKernelGGL(beam_init_kernel, dim3((B + 255) / 256), dim3(256), 0, s, bs, st, enc_lens, B, d.blank_id, n_ids, scores, pops);
    RS_CHECK_LAUNCH(ctx, "beam init");

    const int CHUNK = 32;
    const long long max_iters = (long long)(tp_max > 0 ? tp_max : 1) * mp + 1;
    int32_t hf[2] = {0, 0};
    long long it = 0;
    bool finished = false;
    while (!finished && it < max_iters) {
        for (int c = 0; c < CHUNK; ++c, ++it) {
            hipLaunchKernelGGL(beam_pop_kernel, dim3(B), dim3(64), 0, s, bs, st, B, L, H);
            if (int rc = rs_rnnt_launch_lstm_pred(ctx, &st, B, s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
            if (int rc = rs_rnnt_launch_joint_logits(ctx, &st, joint_enc, B, tp_max, 1, 0, s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
            hipLaunchKernelGGL(beam_expand_kernel, dim3(B), dim3(256), lds, s, bs, st, st.zapprox, pl.zstride, enc_lens, B, L, H, V,
                               d.blank_id, bm, beam_k, score_norm, out_cap, ids, n_ids, scores, pops);
        }
        RS_CHECK_LAUNCH(ctx, "beam step");
        RS_HIP(ctx, hipMemcpyAsync(hf, bs.flags, sizeof hf, hipMemcpyDeviceToHost, s));
        RS_HIP(ctx, hipStreamSynchronize(s));
        finished = hf[0] >= B;
    }
    rs_prof_end(ctx, RS_PROF_DECODE, s);
    if (!finished) return rs_fail(ctx, RS_ESTATE, "beam search: %d of %d utterances unfinished after %lld iterations", B - hf[0], B, it);
    if (hf[1]) return rs_fail(ctx, RS_EOVERFLOW, "beam search: a frame needed more than max_pops=%d prediction-network evaluations, or a result has more than out_cap=%d labels", mp, out_cap);
    return RS_OK;
}

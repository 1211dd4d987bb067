// k_rnnt_beam.hip — the "default" transducer beam search (Graves 2012) as ESPnet2 implements it: [UPSTREAM]
// BeamSearchTransducer.default_beam_search + sort_nbest.  It is the decode reazonspeech.espnet.asr runs: the reference builds
// Speech2Text with its defaults — beam_size 20, search_type "default", score_norm, nbest 1, no LM
// (pkg/espnet-asr/src/transcribe.py:27-31).
//
// The search is frame-synchronous per utterance but the number of pops a frame takes is data dependent, so utterances are NOT
// kept in frame lockstep: every utterance is its own state machine and one device iteration performs ONE pop for every
// utterance that is still searching, whatever frame it is at.  One iteration = 3 + L launches:
//
//   LSTM x L + joint.pred   over the utterances whose popped sequence has not been evaluated yet (k_rnnt.hip, exact f32)
//   joint logits            of every searching utterance at its frame t_b (rnnt_tile_kernel<2>)
//   beam_expand_kernel      (one workgroup per utterance) log-softmax, the blank extension -> `kept`, the beam_k best labels ->
//                           the open list, the end-of-frame test (>= beam entries of kept above the maximum of the open list) and,
//                           at the end of a frame: survivors sorted ascending, their slots compacted into the other pool, t += 1;
//                           at the last frame the winner by score / len(yseq) is read back through the label trie.  Then the NEXT
//                           pop of the utterance (the first maximum of the open list, which the end-of-frame test just found):
//                           a sequence seen for the first time enters the trie and the LSTM work list with the state it starts
//                           from; one evaluated before only hands its cached joint.pred output to the joint
//
// Evaluation order (float32 sums, log-sum-exp tree, tie rules) is documented in oracle/espnet_beam.c and the results are
// bit-identical to it: labels, scores and the pop count.  Compiled with -ffp-contract=off.
#include <cstdio>
#include <cstdlib>

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
    int32_t* nfree;      // free slots (entries of freelist)
    int32_t* nnode;      // trie nodes in use
    int32_t* pops;       // pops over the whole utterance (the work measure, returned)
    // the hypothesis popped for this iteration [B]
    float* cur_score;
    int32_t* cur_node;
    int32_t* cur_slot;   // its own slot when cur_new == 0
    int32_t* cur_len;
    int32_t* cur_new;    // 1: its prediction-network output is being computed in this iteration (row b of the decode state)
    // open list [B][max_h]
    float* h_score;
    int32_t* h_node;     // trie node of the sequence when h_tok < 0, of the sequence without its last label otherwise
    int32_t* h_tok;      // last label of a sequence that has not been evaluated yet, or -1
    int32_t* h_slot;     // h_tok >= 0: slot of the sequence without its last label (the state to start from); else its own slot
    int32_t* h_len;      // len(yseq): labels + the leading blank
    int32_t* h_alive;
    // blank extensions of this frame [B][max_pops]
    float* k_score;
    int32_t* k_node;
    int32_t* k_slot;     // own slot
    int32_t* k_len;
    int2* nodes;         // [B][max_nodes] (parent, label)
    float* slots;        // [B][n_slots][slot_floats]: per evaluated sequence  h [L][H], c [L][H] after its last label, g [J]
    int32_t* freelist;   // [B][n_slots] free slot ids; a frame's end returns every slot no survivor owns
    int32_t* flags;      // [0] utterances done, [1] overflow
    unsigned long long* trace;   // $RS_BEAM_TRACE: cycles per phase of the expand kernel, summed over workgroups (else null)
    int max_h, max_pops, max_nodes, n_slots, slot_floats;
};

// [UPSTREAM] decoder.score(hyp, cache) caches the prediction-network output by label sequence; so does the slot pool: a
// sequence is evaluated ONCE (when a label extension is popped for the first time) and the ~beam survivors that open every
// frame are popped again without touching the LSTM — only the joint depends on the frame.  The cached values are the ones
// the oracle recomputes (same kernels, same inputs), so this changes no bit of the result.

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
    bs.t[b] = 0; bs.nk[b] = 0; bs.npop[b] = 0; bs.pops[b] = 0;
    bs.nnode[b] = 0;
    for (int i = 1; i < bs.n_slots; ++i) bs.freelist[(size_t)b * bs.n_slots + i - 1] = i;
    bs.nfree[b] = bs.n_slots - 1;
    // the start: [blank] is a sequence that has not been evaluated; it starts from slot 0 (never handed out), the zero state
    const size_t h0 = (size_t)b * bs.max_h;
    bs.h_score[h0] = 0.0f; bs.h_node[h0] = -1; bs.h_tok[h0] = blank; bs.h_slot[h0] = 0; bs.h_len[h0] = 1; bs.h_alive[h0] = 1;
    bs.nh[b] = 1;
    st.token[b] = blank; st.tcur[b] = 0;
    n_ids[b] = 0; scores[b] = 0.0f; pops[b] = 0;
    const int fin = enc_lens[b] <= 0;            // nothing to search: the empty hypothesis, score 0
    bs.done[b] = fin;
    if (fin) atomicAdd(&bs.flags[0], 1);
}

// ---- wave-wide first maximum by (value desc, index asc) on the DPP path: row_shr 1/2/4/8 fold each row of 16 lanes into
// its lane 15, row_bcast:15 / row_bcast:31 fold the rows into lane 63 (full-rate VALU moves, no LDS crossbar: a ds_bpermute
// butterfly is 12 dependent ~100-cycle hops, and the search pays one such reduction per label of every expansion).  The order
// is total, so the fold order does not matter.  Entries with index < 0 are "nothing".  Every lane returns the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_fold(float& z, int& v) {
    const float oz = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(z), CTRL, ROW_MASK, 0xf, false));
    const int ov = __builtin_amdgcn_update_dpp(-1, v, CTRL, ROW_MASK, 0xf, false);
    if (ov >= 0 && (v < 0 || oz > z || (oz == z && ov < v))) { z = oz; v = ov; }
}
__device__ __forceinline__ void wave_argmax(float& z, int& v) {
    dpp_fold<0x111, 0xf>(z, v);   // row_shr:1
    dpp_fold<0x112, 0xf>(z, v);   // row_shr:2
    dpp_fold<0x114, 0xf>(z, v);   // row_shr:4
    dpp_fold<0x118, 0xf>(z, v);   // row_shr:8
    dpp_fold<0x142, 0xa>(z, v);   // row_bcast:15 into rows 1 and 3
    dpp_fold<0x143, 0xc>(z, v);   // row_bcast:31 into rows 2 and 3
    z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 63));
    v = __builtin_amdgcn_readlane(v, 63);
}

// ---- block-wide (256 threads) first maximum of the alive entries of the open list: (score desc, index asc) ----
__device__ __forceinline__ void beam_argmax(const BeamState& bs, size_t hb, int n, float* w_f, int* w_i, float& best, int& bi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    best = -INFINITY; bi = -1;
    for (int i = tid; i < n; i += 256)
        if (bs.h_alive[hb + i]) {
            const float s = bs.h_score[hb + i];
            if (bi < 0 || s > best) { best = s; bi = i; }          // a thread meets its entries in ascending order
        }
    wave_argmax(best, bi);
    __syncthreads();                                               // w_f / w_i may still be read from an earlier use
    if (lane == 0) { w_f[wave] = best; w_i[wave] = bi; }
    __syncthreads();
    best = w_f[0]; bi = w_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float os = w_f[w];
        const int oi = w_i[w];
        if (oi >= 0 && (bi < 0 || os > best || (os == best && oi < bi))) { best = os; bi = oi; }
    }
}

// ---- pop entry `bi` of the open list for the next iteration (block-wide; every thread passes the same bi) ----
// list = the joint work list the next iteration reads
__device__ __forceinline__ void beam_pop(const BeamState& bs, const DecodeState& st, int b, int B, int L, int H, int J, int bi,
                                         float score, int list) {
    const int tid = threadIdx.x;
    const size_t hb = (size_t)b * bs.max_h;
    const bool none = bi < 0;                                      // cannot happen (a pop always opens beam_k >= 1 extensions)
    if (none) bi = 0;
    const int tok = bs.h_tok[hb + bi], slot = bs.h_slot[hb + bi];
    int node = bs.h_node[hb + bi];
    const bool bad = none || bs.npop[b] >= bs.max_pops || (tok >= 0 && bs.nnode[b] >= bs.max_nodes);
    __syncthreads();                                               // every thread has read the entry and the counters
    if (bad) { if (tid == 0) beam_fail(bs, b); return; }
    const int LH = L * H;
    const float* src = bs.slots + ((size_t)b * bs.n_slots + slot) * (size_t)bs.slot_floats;
    if (tid == 0) {
        bs.h_alive[hb + bi] = 0;
        if (tok >= 0) {                                            // the sequence enters the trie, and the LSTM work list
            const int nn = bs.nnode[b];
            bs.nodes[(size_t)b * bs.max_nodes + nn] = make_int2(node, tok);
            node = nn;
            bs.nnode[b] = nn + 1;
            st.token[b] = tok;
            st.act[atomicAdd(&st.counters[0], 1)] = b;
        }
        bs.cur_score[b] = score; bs.cur_node[b] = node; bs.cur_slot[b] = slot; bs.cur_len[b] = bs.h_len[hb + bi];
        bs.cur_new[b] = tok >= 0;
        st.tcur[b] = bs.t[b];
        st.alive[(size_t)list * B + atomicAdd(&st.counters[2 + list], 1)] = b;
    }
    if (tok >= 0) {                                                // start state of the evaluation
        for (int i = tid; i < LH; i += 256) {
            const int l = i / H, u = i - l * H;
            st.h[((size_t)l * B + b) * H + u] = src[i];
            st.c[((size_t)l * B + b) * H + u] = src[LH + i];
        }
    } else {                                                       // evaluated before: only the joint needs it
        for (int i = tid; i < J; i += 256) st.g[(size_t)b * J + i] = src[2 * LH + i];
    }
}

// the first pop of every utterance; grid B, block 256
__global__ __launch_bounds__(256) void beam_first_pop_kernel(BeamState bs, DecodeState st, int B, int L, int H, int J) {
    const int b = blockIdx.x;
    if (bs.done[b]) return;
    beam_pop(bs, st, b, B, L, H, J, 0, 0.0f, 0);
}

#define BEAM_MARK(phase)                                                                       \
    if (bs.trace && tid == 0) {                                                                \
        const unsigned long long now = wall_clock64();                                         \
        atomicAdd(&bs.trace[2 * (phase)], now - t_mark);                                       \
        atomicAdd(&bs.trace[2 * (phase) + 1], 1ull);                                           \
        t_mark = now;                                                                          \
    }

// grid B, block 256, dynamic LDS: z row [zstride] + exp terms [zstride] + kept scores [max_pops] + slot marks [n_slots]
// EPT = logits per thread (V <= 256 * EPT): a thread keeps its logits v = tid + 256 e in registers for the label rounds
template <int EPT>
__global__ __launch_bounds__(256) void beam_expand_kernel(BeamState bs, DecodeState st, const float* __restrict__ zbuf, int zstride,
                                                          const int32_t* __restrict__ enc_lens, int B, int L, int H, int J, int V,
                                                          int blank, int beam, int beam_k, int score_norm, int out_cap, int iter,
                                                          int32_t* __restrict__ ids, int32_t* __restrict__ n_ids,
                                                          float* __restrict__ scores, int32_t* __restrict__ pops) {
    extern __shared__ __attribute__((aligned(16))) char beam_smem[];
    float* zs = reinterpret_cast<float*>(beam_smem);
    float* es = zs + zstride;                                        // exp(z - max)
    float* ks = es + zstride;                                        // kept scores
    int* used = reinterpret_cast<int*>(ks + bs.max_pops);            // slot -> owned by a survivor
    __shared__ float w_f[2][4];
    __shared__ int w_i[2][4];
    __shared__ float s_lse;
    __shared__ int s_good, s_best, s_nfree;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (bs.done[b]) return;
    unsigned long long t_mark = bs.trace ? wall_clock64() : 0ull;
    const int LH = L * H, SS = bs.slot_floats;
    float* pool = bs.slots + ((size_t)b * bs.n_slots) * (size_t)SS;
    int32_t* fl = bs.freelist + (size_t)b * bs.n_slots;
    const int npop = bs.npop[b], nfree = bs.nfree[b], nk = bs.nk[b], nh0 = bs.nh[b];
    const int is_new = bs.cur_new[b];
    // the popped hypothesis' own slot: a fresh one when it was evaluated in this iteration (never short: n_slots - 1 =
    // 2 * max_pops >= survivors of the last frame + evaluations of this one)
    const int own = is_new ? fl[nfree - 1] : bs.cur_slot[b];
    if (is_new) {
        float* dst = pool + (size_t)own * SS;
        for (int i = tid; i < LH; i += 256) {
            const int l = i / H, u = i - l * H;
            dst[i] = st.h[((size_t)l * B + b) * H + u];
            dst[LH + i] = st.c[((size_t)l * B + b) * H + u];
        }
        for (int i = tid; i < J; i += 256) dst[2 * LH + i] = st.g[(size_t)b * J + i];
    }
    const float* zr = zbuf + (size_t)b * zstride;
    float zreg[EPT];
    unsigned gone = 0;                                               // bit e: logit e of this thread is not a candidate (any more)
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int v = tid + 256 * e;
        const bool in = v < V;
        zreg[e] = in ? zr[v] : -INFINITY;
        if (in) zs[v] = zreg[e];
        m = fmaxf(m, zreg[e]);
        gone |= (unsigned)(!in || v == blank) << e;
    }
    m = wave_max(m);
    if (lane == 0) w_f[0][wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(w_f[0][0], w_f[0][1]), fmaxf(w_f[0][2], w_f[0][3]));
    BEAM_MARK(0)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {                                  // the terms of the sum; added below in the documented order
        const int v = tid + 256 * e;
        if (v < V) es[v] = rs_expf(zreg[e] - m);
    }
    __syncthreads();
    if (wave == 0) {
        float sum = 0.0f;
        for (int v = lane; v < V; v += 64) sum = sum + es[v];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum = sum + __shfl_xor(sum, off, 64);
        if (lane == 0) s_lse = m + rs_logf(sum);
    }
    __syncthreads();
    BEAM_MARK(1)
    const float lse = s_lse;
    const float zblank = zs[blank];
    const float hs = bs.cur_score[b];
    const int cnode = bs.cur_node[b], clen = bs.cur_len[b];
    const size_t hb = (size_t)b * bs.max_h, kb = (size_t)b * bs.max_pops;
    // ---- the beam_k best labels by (logit desc, index asc) ----
    // Fast path: a threshold that at least beam_k candidates reach, the (few) logits at or above it gathered in LDS, each ranked
    // by counting.  theta = the smallest, over the four waves, of a wave's q-th largest per-thread maximum with q =
    // ceil(beam_k / 4): every wave then holds q threads whose maximum is >= theta, so at least beam_k logits qualify, and the
    // beam_k best overall are among them.  (Picking them one at a time costs a block-wide reduction per label.)
    float my_z = 0.0f;
    int n_child = 0, my_v = -1;
    bool ranked = false;
    if (beam_k <= 128) {
        float tmax = -INFINITY;                                      // this thread's best candidate
        int targ = -1;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const bool take = !((gone >> e) & 1u) & ((targ < 0) | (zreg[e] > tmax));
            tmax = take ? zreg[e] : tmax;
            targ = take ? tid + 256 * e : targ;
        }
        const int q = (beam_k + 3) / 4;
        float cz = tmax, theta_w = INFINITY;
        int cv = targ;
        for (int r = 0; r < q; ++r) {                                // the wave's q-th largest thread maximum (wave-local rounds: no barrier)
            float bz = cz;
            int bv = cv;
            wave_argmax(bz, bv);
            theta_w = bv < 0 ? -INFINITY : bz;
            if (bv < 0) break;
            if (cv == bv) cv = -1;                                   // its owner steps aside
        }
        if (lane == 0) w_f[1][wave] = theta_w;
        if (tid == 0) s_good = 0;                                    // candidate counter
        __syncthreads();
        const float theta = fminf(fminf(w_f[1][0], w_f[1][1]), fminf(w_f[1][2], w_f[1][3]));
        // gather (logit, label) of every candidate >= theta into es[] / zs[] (both free from here: zs[blank] was read above)
        int* cand_v = reinterpret_cast<int*>(zs);
        float* cand_z = es;
        const int cap = V < 256 ? V : 256;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (!((gone >> e) & 1u) && zreg[e] >= theta) {
                const int slot = atomicAdd(&s_good, 1);
                if (slot < cap) { cand_z[slot] = zreg[e]; cand_v[slot] = tid + 256 * e; }
            }
        __syncthreads();
        const int n_c = s_good;
        if (n_c <= cap) {                                            // (else: a plateau of equal logits; the rounds below handle it)
            ranked = true;
            n_child = n_c < beam_k ? n_c : beam_k;
            if (tid < n_c) {
                const float zi = cand_z[tid];
                const int vi = cand_v[tid];
                int rank = 0;
                for (int o = 0; o < n_c; ++o) {
                    const float zo = cand_z[o];
                    const int vo = cand_v[o];
                    rank += (zo > zi) | ((zo == zi) & (vo < vi));
                }
                if (rank < n_child) {                                // pick `rank`: the open list takes it in that position
                    const size_t o = hb + nh0 + rank;
                    bs.h_score[o] = hs + (zi - lse);
                    bs.h_node[o] = cnode; bs.h_tok[o] = vi; bs.h_slot[o] = own; bs.h_len[o] = clen + 1; bs.h_alive[o] = 1;
                }
            }
        }
    }
    // General path, one label per round: the first maximum of what has not been taken yet
    for (int j = 0; !ranked && j < beam_k; ++j) {
        float bz = -INFINITY;
        int bv = -1;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {                              // ascending v: a later equal logit does not displace
            const bool take = !((gone >> e) & 1u) & ((bv < 0) | (zreg[e] > bz));
            bz = take ? zreg[e] : bz;
            bv = take ? tid + 256 * e : bv;
        }
        wave_argmax(bz, bv);
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
        if ((bv & 255) == tid) gone |= 1u << (bv >> 8);              // its owner retires it
        if (tid == j) { my_z = bz; my_v = bv; }                       // thread j keeps pick j (a global store here would put its
        ++n_child;                                                   // round trip into every round's barrier)
    }
    if (tid == 255) {                                                // the blank extension
        bs.k_score[kb + nk] = hs + (zblank - lse);
        bs.k_node[kb + nk] = cnode; bs.k_slot[kb + nk] = own; bs.k_len[kb + nk] = clen;
    }
    if (!ranked && tid < n_child) {
        const size_t o = hb + nh0 + tid;
        bs.h_score[o] = hs + (my_z - lse);
        bs.h_node[o] = cnode; bs.h_tok[o] = my_v; bs.h_slot[o] = own; bs.h_len[o] = clen + 1; bs.h_alive[o] = 1;
    }
    const int nh = nh0 + n_child, nkk = nk + 1;
    __syncthreads();                                                 // the new entries are visible to the block below
    BEAM_MARK(2)
    // end-of-frame test: at least `beam` kept entries strictly above the maximum of the open list — whose first maximum is
    // also the next hypothesis to pop if the frame goes on
    float hm;
    int bi;
    beam_argmax(bs, hb, nh, w_f[0], w_i[0], hm, bi);
    for (int i = tid; i < nkk; i += 256) ks[i] = bs.k_score[kb + i];
    if (tid == 0) s_good = 0;
    __syncthreads();
    int good = 0;
    for (int i = tid; i < nkk; i += 256) good += ks[i] > hm;
    if (good) atomicAdd(&s_good, good);
    __syncthreads();
    const int n_good = s_good;
    const int list = (iter + 1) & 1;
    BEAM_MARK(3)
    if (n_good < beam) {                                             // the frame goes on
        if (tid == 0) { bs.nh[b] = nh; bs.nk[b] = nkk; bs.npop[b] = npop + 1; bs.nfree[b] = nfree - is_new; bs.pops[b] += 1; }
        __syncthreads();
        beam_pop(bs, st, b, B, L, H, J, bi, hm, list);
        BEAM_MARK(4)
        return;
    }
    // ---- end of frame: survivors ascending by score (ties in kept order) become the next frame's open list; every slot that
    // no survivor owns goes back to the free list (in any order: slot ids never reach a result) ----
    for (int sl = tid; sl < bs.n_slots; sl += 256) used[sl] = sl == 0;
    if (tid == 0) s_nfree = 0;
    __syncthreads();
    for (int i = tid; i < nkk; i += 256) {
        const float si = ks[i];
        if (!(si > hm)) continue;
        int rank = 0;
        for (int o = 0; o < nkk; ++o) {
            const float so = ks[o];
            if (so > hm && (so < si || (so == si && o < i))) ++rank;
        }
        const int slot = bs.k_slot[kb + i];
        bs.h_score[hb + rank] = si;
        bs.h_node[hb + rank] = bs.k_node[kb + i]; bs.h_tok[hb + rank] = -1; bs.h_slot[hb + rank] = slot;
        bs.h_len[hb + rank] = bs.k_len[kb + i]; bs.h_alive[hb + rank] = 1;
        used[slot] = 1;
    }
    __syncthreads();
    BEAM_MARK(5)
    const int t_next = bs.t[b] + 1;
    const bool last = t_next >= enc_lens[b];
    if (!last) {
        for (int sl = tid; sl < bs.n_slots; sl += 256)
            if (!used[sl]) fl[atomicAdd(&s_nfree, 1)] = sl;
        __syncthreads();
        BEAM_MARK(6)
        if (tid == 0) {
            bs.nh[b] = n_good; bs.nk[b] = 0; bs.npop[b] = 0; bs.nfree[b] = s_nfree; bs.t[b] = t_next;
            bs.pops[b] += 1;
        }
        __syncthreads();                                             // the new list and the counters are in place
        beam_argmax(bs, hb, n_good, w_f[1], w_i[1], hm, bi);
        beam_pop(bs, st, b, B, L, H, J, bi, hm, list);
        BEAM_MARK(7)
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
        wave_argmax(bn, br);
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
    BEAM_MARK(8)
}

struct BeamPlan {
    size_t b4, h4, k4, nodes, slots, freelist, state1, g, rows4, z, total;
    int max_h, max_nodes, n_slots, slot_floats, zstride;
};

BeamPlan beam_plan(const rs_ctx* ctx, int B, int beam_k, int tp_max, int max_pops) {
    const rs_dims& d = ctx->d;
    BeamPlan p;
    p.max_h = max_pops * (beam_k + 1) + 1;
    p.max_nodes = (tp_max > 0 ? tp_max : 1) * max_pops + 1;        // a pop adds at most one node
    p.n_slots = 2 * max_pops + 1;                                    // zero state + survivors (<= max_pops) + evaluations of a frame (<= max_pops)
    p.slot_floats = 2 * d.pred_layers * d.pred_hidden + d.joint_hidden;
    p.zstride = (d.n_logits + 63) / 64 * 64;
    p.b4 = rs_align((size_t)B * 4);
    p.h4 = rs_align((size_t)B * p.max_h * 4);
    p.k4 = rs_align((size_t)B * max_pops * 4);
    p.nodes = rs_align((size_t)B * p.max_nodes * 8);
    p.state1 = rs_align((size_t)d.pred_layers * B * d.pred_hidden * 4);
    p.slots = rs_align((size_t)B * p.n_slots * p.slot_floats * 4);
    p.freelist = rs_align((size_t)B * p.n_slots * 4);
    p.g = rs_align((size_t)B * d.joint_hidden * 4);
    p.rows4 = rs_align((size_t)B * 4);
    p.z = rs_align((size_t)B * p.zstride * 4);
    p.total = 13 * p.b4 + 6 * p.h4 + 4 * p.k4 + p.nodes + p.slots + p.freelist + 4 * p.state1 + p.g + 6 * p.rows4 + 2 * rs_align(64) + rs_align(256) + p.z + 1024;
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
    if (bm > 128) return rs_fail(ctx, RS_EINVAL, "beam search: beam size must be 1..128");
    if (mp < bm) return rs_fail(ctx, RS_EINVAL, "beam search: max_pops %d < beam %d (a frame needs at least `beam` pops)", mp, bm);
    const BeamPlan pl = beam_plan(ctx, B, beam_k, tp_max, mp);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "beam search: workspace %zu < %zu", workspace_bytes, pl.total);
    const size_t lds = (size_t)pl.zstride * 8 + (size_t)mp * 4 + (size_t)pl.n_slots * 4;
    if (lds > 60 * 1024) return rs_fail(ctx, RS_EINVAL, "beam search: vocabulary %d / max_pops %d exceed the expand kernel's LDS", V, mp);
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* q = w; w += bytes; return q; };
    BeamState bs;
    DecodeState st;
    // every int32 / float bookkeeping array first (one memset), then the big buffers
    char* zero_from = w;
    bs.t = (int32_t*)take(pl.b4); bs.done = (int32_t*)take(pl.b4); bs.nh = (int32_t*)take(pl.b4); bs.nk = (int32_t*)take(pl.b4);
    bs.npop = (int32_t*)take(pl.b4); bs.nfree = (int32_t*)take(pl.b4);
    bs.nnode = (int32_t*)take(pl.b4); bs.pops = (int32_t*)take(pl.b4);
    bs.cur_score = (float*)take(pl.b4); bs.cur_node = (int32_t*)take(pl.b4); bs.cur_slot = (int32_t*)take(pl.b4);
    bs.cur_len = (int32_t*)take(pl.b4); bs.cur_new = (int32_t*)take(pl.b4);
    bs.flags = (int32_t*)take(rs_align(64));
    int32_t* counters = (int32_t*)take(rs_align(64));
    unsigned long long* trace = (unsigned long long*)take(rs_align(256));
    bs.trace = getenv("RS_BEAM_TRACE") ? trace : nullptr;
    const size_t zero_bytes = (size_t)(w - zero_from);
    bs.h_score = (float*)take(pl.h4); bs.h_node = (int32_t*)take(pl.h4); bs.h_tok = (int32_t*)take(pl.h4);
    bs.h_slot = (int32_t*)take(pl.h4); bs.h_len = (int32_t*)take(pl.h4); bs.h_alive = (int32_t*)take(pl.h4);
    bs.k_score = (float*)take(pl.k4); bs.k_node = (int32_t*)take(pl.k4); bs.k_slot = (int32_t*)take(pl.k4);
    bs.k_len = (int32_t*)take(pl.k4);
    bs.nodes = (int2*)take(pl.nodes);
    bs.slots = (float*)take(pl.slots);
    bs.freelist = (int32_t*)take(pl.freelist);
    bs.max_h = pl.max_h; bs.max_pops = mp; bs.max_nodes = pl.max_nodes; bs.n_slots = pl.n_slots; bs.slot_floats = pl.slot_floats;
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

    auto expand = V <= 256 * 4 ? beam_expand_kernel<4> : V <= 256 * 12 ? beam_expand_kernel<12> : V <= 256 * 20 ? beam_expand_kernel<20> : beam_expand_kernel<32>;
    if (V > 256 * 32) return rs_fail(ctx, RS_EINVAL, "beam search: vocabulary %d > 8192", V);
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)expand, (int)lds); rc != RS_OK) return rc;
    rs_prof_begin(ctx, RS_PROF_DECODE, s, 0.0, 0.0);
    RS_HIP(ctx, hipMemsetAsync(zero_from, 0, zero_bytes, s));
    // slot 0 of every utterance: the zero state the search starts from
    RS_HIP(ctx, hipMemset2DAsync(bs.slots, (size_t)pl.n_slots * pl.slot_floats * 4, 0, (size_t)pl.slot_floats * 4, B, s));
    hipLaunchKernelGGL(beam_init_kernel, dim3((B + 255) / 256), dim3(256), 0, s, bs, st, enc_lens, B, d.blank_id, n_ids, scores, pops);
    hipLaunchKernelGGL(beam_first_pop_kernel, dim3(B), dim3(256), 0, s, bs, st, B, L, H, J);
    RS_CHECK_LAUNCH(ctx, "beam init");

    const int CHUNK = 32;
    const long long max_iters = (long long)(tp_max > 0 ? tp_max : 1) * mp + 1;
    int32_t hf[2] = {0, 0};
    long long it = 0;
    bool finished = false;
    while (!finished && it < max_iters) {
        for (int c = 0; c < CHUNK; ++c, ++it) {
            if (int rc = rs_rnnt_launch_lstm_pred(ctx, &st, B, s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
            if (int rc = rs_rnnt_launch_joint_logits(ctx, &st, joint_enc, B, tp_max, 1, (int)(it & 1), s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
            hipLaunchKernelGGL(expand, dim3(B), dim3(256), lds, s, bs, st, st.zapprox, pl.zstride, enc_lens, B, L, H, J, V,
                               d.blank_id, bm, beam_k, score_norm, out_cap, (int)(it & 1), ids, n_ids, scores, pops);
        }
        RS_CHECK_LAUNCH(ctx, "beam step");
        RS_HIP(ctx, hipMemcpyAsync(hf, bs.flags, sizeof hf, hipMemcpyDeviceToHost, s));
        RS_HIP(ctx, hipStreamSynchronize(s));
        finished = hf[0] >= B;
    }
    rs_prof_end(ctx, RS_PROF_DECODE, s);
    if (bs.trace) {                                                  // diagnostic: where the expand kernel's workgroups spend their time
        unsigned long long tr[18];
        RS_HIP(ctx, hipMemcpy(tr, trace, sizeof tr, hipMemcpyDeviceToHost));
        static const char* names[9] = {"park+stage+max", "exp+lse", "label rounds", "argmax+count", "pop (frame goes on)", "rank survivors",
                                       "free slots", "argmax+pop (new frame)", "read back"};
        for (int i = 0; i < 9; ++i)
            fprintf(stderr, "[beam trace] %-24s %10llu passes  %8.2f us each (100 MHz wall clock)\n", names[i], tr[2 * i + 1],
                    tr[2 * i + 1] ? (double)tr[2 * i] / (double)tr[2 * i + 1] / 100.0 : 0.0);
    }
    if (!finished) return rs_fail(ctx, RS_ESTATE, "beam search: %d of %d utterances unfinished after %lld iterations", B - hf[0], B, it);
    if (hf[1]) return rs_fail(ctx, RS_EOVERFLOW, "beam search: a frame needed more than max_pops=%d prediction-network evaluations, or a result has more than out_cap=%d labels", mp, out_cap);
    return RS_OK;
}

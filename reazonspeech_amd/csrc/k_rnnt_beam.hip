// k_rnnt_beam.hip — the "default" transducer beam search (Graves 2012) as ESPnet2 implements it: [UPSTREAM]
// BeamSearchTransducer.default_beam_search + sort_nbest.  It is the decode reazonspeech.espnet.asr runs: the reference builds
// Speech2Text with its defaults — beam_size 20, search_type "default", score_norm, nbest 1, no LM
// (pkg/espnet-asr/src/transcribe.py:27-31).
//
// Per frame the search pops the best open hypothesis, keeps its blank extension, opens its `beam` best label extensions, and
// repeats until `beam` kept hypotheses beat everything still open.  What a pop needs is the log-softmax of ONE joint row
// (hypothesis, frame); which hypotheses get popped is only known as the search goes.  Two facts shape the device version:
//
//   * the hypotheses that open a frame are the survivors of the previous one (>= beam of them) and nearly all of them are
//     popped again — so their joint rows at the new frame are computed TOGETHER when the frame starts (up to R of them, best
//     first), as one tall launch instead of ~beam launches of one row per utterance; a row that turns out not to be needed
//     costs a few MFLOP and changes nothing;
//   * the prediction network only depends on the label sequence ([UPSTREAM] decoder.score caches by sequence): a sequence is
//     evaluated once, when one of its label extensions is popped for the first time, and the result (LSTM state + joint.pred
//     vector) stays in a slot that its extensions and its own later pops refer to.
//
// Utterances are independent state machines (the pop count of a frame is data dependent, so they are NOT kept in frame
// lockstep).  One device iteration = 5 + L launches:
//
//   LSTM x L + joint.pred   over the utterances waiting for a new sequence's evaluation (k_rnnt.hip, exact f32)
//   beam_act_kernel         act(f + g) of this iteration's rows: the batch rows of utterances that just entered a frame, the single
//                           row of those waiting for one evaluation
//   joint logits            rnnt_tile_kernel<4> over those rows
//   beam_record_kernel      one wave per row: log-softmax, log p(blank), the beam_k best labels with their log-probabilities
//   beam_step_kernel        one workgroup per utterance, open-list scores in LDS: pops for as long as the best open hypothesis has
//                           its record at this frame; ends the frame when the test says so (survivors sorted, slots of
//                           everything else freed, t += 1, next batch scheduled, or the winner read back through the label trie
//                           after the last frame); at the first pop that has no record it asks for the expansions of the best
//                           KS record-less hypotheses (the first is needed, the others are guesses that usually come true)
//
// Evaluation order (float32 sums, log-sum-exp tree, tie rules) is documented in oracle/espnet_beam.c and the results are
// bit-identical to it: labels, scores and the pop count.  Compiled with -ffp-contract=off.
#include <cstdio>
#include <cstdlib>

#include "k_rnnt_common.h"

int rs_rnnt_launch_lstm_pred(rs_ctx* ctx, const void* st_ptr, int rows, hipStream_t s);
int rs_rnnt_launch_joint_logits_indirect(rs_ctx* ctx, const void* st_ptr, const float* joint_enc, int rows, int rows_bound, int tp_max,
                                         int rows_per_utt, int step, hipStream_t s);

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
    int32_t* nh;         // entries of the open list (dead ones included)
    int32_t* nk;         // entries of kept
    int32_t* npop;       // pops of the current frame
    int32_t* nfree;      // free slots (entries of freelist)
    int32_t* ninit;      // hypotheses the current frame started with = entries [0, ninit) of the open list
    int32_t* nnode;      // trie nodes in use
    int32_t* pops;       // pops over the whole utterance (the work measure, returned)
    int32_t* nb;         // records [0, nb): the frame's batch
    int32_t* nrec;       // records [R, nrec): evaluations asked for one iteration at a time
    int32_t* npark;      // prediction-network rows of the last iteration whose results still sit in the decode state
    int32_t* park_slot;  // [B][KS] the slot each of them goes to
    // open list [B][max_h]
    float* h_score;
    int32_t* h_node;     // trie node of the sequence when h_tok < 0, of the sequence without its last label otherwise
    int32_t* h_tok;      // last label of a sequence that is not in the trie yet, or -1
    int32_t* h_slot;     // h_tok >= 0: slot of the sequence without its last label (the state to start from); else its own slot
    int32_t* h_len;      // len(yseq): labels + the leading blank
    int32_t* h_alive;
    int32_t* h_rec;      // record that holds its expansion at the current frame, or -1
    // blank extensions of this frame [B][max_pops]
    float* k_score;
    int32_t* k_node;
    int32_t* k_slot;     // own slot
    int32_t* k_len;
    int2* nodes;         // [B][max_nodes] (parent, label)
    int32_t* node_frame; // [B][max_nodes] the frame the label was appended at
    float* slots;        // [B][n_slots][slot_floats]: per evaluated sequence  h [L][H], c [L][H] after its last label, g [J]
    int32_t* freelist;   // [B][n_slots] free slot ids; a frame's end returns every slot no survivor owns
    float* rec;          // [B][RP][rec_floats]: log p(blank), label count, log p(label j) x beam_k, label j x beam_k
    int32_t* rec_slot;   // [B][RP] own slot of the hypothesis a record belongs to
    int32_t* row_rec;    // [B][R + KS] record a joint row of this iteration fills
    long long* g_off;    // [B][R + KS] where a joint row's joint.pred vector is, in floats from DecodeState.g
    long long slots_off; // slots - DecodeState.g
    int32_t* flags;      // [0] utterances done, [1] overflow
    unsigned long long* trace;   // $RS_BEAM_TRACE: cycles per phase of the step kernel, summed over workgroups (else null)
    int max_h, max_pops, max_nodes, n_slots, slot_floats, R, KS, RP, rec_floats, beam_k;
};

__device__ __forceinline__ void beam_fail(const BeamState& bs, int b) {   // one thread
    bs.done[b] = 1;
    bs.flags[1] = 1;
    atomicAdd(&bs.flags[0], 1);
}

// LDS-only barrier: waits for this wave's LDS traffic, not for its global stores (a __syncthreads() also drains vmcnt, which
// would put the round trip of every fire-and-forget store of the pop loop into the loop)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// grid ceil(B / 256)
__global__ __launch_bounds__(256) void beam_init_kernel(BeamState bs, const int32_t* __restrict__ enc_lens, int B, int blank,
                                                        int32_t* __restrict__ n_ids, float* __restrict__ scores,
                                                        int32_t* __restrict__ pops) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    bs.t[b] = 0; bs.nk[b] = 0; bs.npop[b] = 0; bs.pops[b] = 0; bs.ninit[b] = 0; bs.nb[b] = 0; bs.nrec[b] = bs.R; bs.npark[b] = 0;
    bs.nnode[b] = 0;
    for (int i = 1; i < bs.n_slots; ++i) bs.freelist[(size_t)b * bs.n_slots + i - 1] = i;
    bs.nfree[b] = bs.n_slots - 1;
    // the start: [blank] is a sequence that has not been evaluated; it starts from slot 0 (never handed out), the zero state
    const size_t h0 = (size_t)b * bs.max_h;
    bs.h_score[h0] = 0.0f; bs.h_node[h0] = -1; bs.h_tok[h0] = blank; bs.h_slot[h0] = 0; bs.h_len[h0] = 1; bs.h_alive[h0] = 1;
    bs.h_rec[h0] = -1;
    bs.nh[b] = 1;
    n_ids[b] = 0; scores[b] = 0.0f; pops[b] = 0;
    const int fin = enc_lens[b] <= 0;            // nothing to search: the empty hypothesis, score 0
    bs.done[b] = fin;
    if (fin) atomicAdd(&bs.flags[0], 1);
}

// ---- wave-wide first maximum by (value desc, index asc) on the DPP path: row_shr 1/2/4/8 fold each row of 16 lanes into
// its lane 15, row_bcast:15 / row_bcast:31 fold the rows into lane 63 (full-rate VALU moves, no LDS crossbar: a ds_bpermute
// butterfly is 12 dependent ~100-cycle hops).  The order is total, so the fold order does not matter.  Entries with index < 0
// are "nothing".  Every lane returns the result.
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

// ---- a_pre[row] = act(f[utt][t] + g[row]) for the rows of this iteration's joint list (the joint's column tiles then read it
// instead of each recomputing it); one thread per 4 elements, workgroups stride over the list ----
__global__ __launch_bounds__(256) void beam_act_kernel(DecodeState st, const float* __restrict__ f, float* __restrict__ a_pre, int rows,
                                                       int Tp, int J, int rows_per_utt, int list) {
    const int n = st.counters[2 + list];
    const int q4 = J / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)n * q4; i += (long long)gridDim.x * 256) {
        const int idx = (int)(i / q4), q = (int)(i - (long long)idx * q4);
        const int row = st.alive[(size_t)list * rows + idx];
        int t = st.tcur[row];
        t = t < Tp ? t : Tp - 1;
        const float4 a = reinterpret_cast<const float4*>(f + ((size_t)(row / rows_per_utt) * Tp + t) * J)[q];
        const float4 g = reinterpret_cast<const float4*>(st.g + st.g_off[row])[q];
        float4 r;
        if (st.joint_act) { r.x = rs_tanhf(a.x + g.x); r.y = rs_tanhf(a.y + g.y); r.z = rs_tanhf(a.z + g.z); r.w = rs_tanhf(a.w + g.w); }
        else { r.x = fmaxf(a.x + g.x, 0.0f); r.y = fmaxf(a.y + g.y, 0.0f); r.z = fmaxf(a.z + g.z, 0.0f); r.w = fmaxf(a.w + g.w, 0.0f); }
        reinterpret_cast<float4*>(a_pre + (size_t)row * J)[q] = r;
    }
}

// ---- records: one wave per row of this iteration's joint list ----------------------------------------------------------
// rec[row] = { log p(blank), n (int bits), log p(label_j) for j < beam_k, label_j (int bits) for j < beam_k }, labels by
// (logit desc, index asc).  log-sum-exp in the documented order: lane l adds exp(z[v] - max) for v = l, l + 64, ..., then the
// tree p[l] += p[l + off].  Selection: theta = the beam_k-th largest per-lane maximum — at least beam_k logits reach it — the few
// logits >= theta are gathered in LDS and ranked by counting (a plateau of more than 128 equal logits falls back to one
// wave-wide maximum per label).
// EPT > 0: V <= 64 * EPT and a lane keeps its logits v = lane + 64 e in registers (a wave-wide load of 64 consecutive floats
// is one coalesced request; EPT of them are in flight); EPT = 0: any V, the row is staged in LDS and re-read from there.
// grid: any (workgroups stride over the list), block 256, dynamic LDS 4 x (zstride + 256) floats
template <int EPT>
__global__ __launch_bounds__(256) void beam_record_kernel(BeamState bs, DecodeState st, const float* __restrict__ zbuf, int zstride,
                                                          int rows, int rows_per_utt, int V, int blank, int list) {
    extern __shared__ __attribute__((aligned(16))) char rec_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* zs = reinterpret_cast<float*>(rec_smem) + (size_t)wave * (zstride + 256);
    float* cz = zs + zstride;
    int* cv = reinterpret_cast<int*>(cz + 128);
    const int n = st.counters[2 + list];
    const int beam_k = bs.beam_k;
    const int ne = EPT > 0 ? EPT : (V + 63) / 64;
    for (int idx = blockIdx.x * 4 + wave; idx < n; idx += gridDim.x * 4) {
        const int row = st.alive[(size_t)list * rows + idx];
        const float* zr = zbuf + (size_t)row * zstride;
        float zreg[EPT > 0 ? EPT : 1];
        if (EPT > 0) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) { const int v = lane + 64 * e; zreg[e] = v < V ? zr[v] : -INFINITY; }
        } else {
            const float4* zr4 = reinterpret_cast<const float4*>(zr);          // rows are zstride (multiple of 64) floats apart
            float4* zs4 = reinterpret_cast<float4*>(zs);
            const int n4 = zstride / 4;
            for (int q0 = lane; q0 < n4; q0 += 64 * 8) {
                float4 buf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int q = q0 + 64 * u; buf[u] = q < n4 ? zr4[q] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int q = q0 + 64 * u; if (q < n4) zs4[q] = buf[u]; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#define ZAT(e) (EPT > 0 ? zreg[EPT > 0 ? (e) : 0] : (lane + 64 * (e) < V ? zs[lane + 64 * (e)] : -INFINITY))
        float m = -INFINITY, tmax = -INFINITY, zb = -INFINITY;
        int targ = -1;
#pragma unroll
        for (int e = 0; e < ne; ++e) {
            const int v = lane + 64 * e;
            const float zv = ZAT(e);
            m = fmaxf(m, zv);
            zb = v == blank ? zv : zb;
            const bool take = (v < V) & (v != blank) & ((targ < 0) | (zv > tmax));
            tmax = take ? zv : tmax;
            targ = take ? v : targ;
        }
        m = wave_max(m);
        zb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zb), blank & 63));
        float sum = 0.0f;
#pragma unroll
        for (int e = 0; e < ne; ++e)
            if (lane + 64 * e < V) sum = sum + rs_expf(ZAT(e) - m);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum = sum + __shfl_xor(sum, off, 64);
        sum = __shfl(sum, 0, 64);
        const float lse = m + rs_logf(sum);
        // theta = the beam_k-th largest lane maximum: every lane ranks its own against the other 63 (readlane broadcasts)
        float theta;
        {
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const float oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tmax), j));
                const int ov = __builtin_amdgcn_readlane(targ, j);
                rank += (ov >= 0) & ((targ < 0) | (oz > tmax) | ((oz == tmax) & (ov < targ)));
            }
            theta = wave_max((rank == beam_k - 1 && targ >= 0) ? tmax : -INFINITY);   // no such lane: fewer than beam_k candidates
        }
        // gather the logits >= theta: ballot compaction (positions by lane order within a pass; any order would do)
        int n_c = 0;
#pragma unroll
        for (int e = 0; e < ne; ++e) {
            const int v = lane + 64 * e;
            const float zv = ZAT(e);
            const bool in = (v < V) & (v != blank) & (zv >= theta);
            const unsigned long long mask = __ballot(in);
            const int pos = n_c + __popcll(mask & ((1ull << lane) - 1ull));
            if (in && pos < 128) { cz[pos] = zv; cv[pos] = v; }
            n_c += __popcll(mask);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float* out = bs.rec + ((size_t)(row / rows_per_utt) * bs.RP + bs.row_rec[row]) * bs.rec_floats;
        int n_lab = 0;
        if (n_c <= 64) {                                             // the usual case: one candidate per lane, ranked in registers
            n_lab = n_c < beam_k ? n_c : beam_k;
            const float zi = lane < n_c ? cz[lane] : -INFINITY;
            const int vi = lane < n_c ? cv[lane] : -1;
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const float zo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zi), j));
                const int vo = __builtin_amdgcn_readlane(vi, j);
                rank += (vo >= 0) & ((zo > zi) | ((zo == zi) & (vo < vi)));
            }
            if (vi >= 0 && rank < n_lab) { out[2 + rank] = zi - lse; out[2 + beam_k + rank] = __int_as_float(vi); }
        } else if (n_c <= 128) {
            n_lab = n_c < beam_k ? n_c : beam_k;
            for (int c = lane; c < n_c; c += 64) {
                const float zi = cz[c];
                const int vi = cv[c];
                int rank = 0;
                for (int o = 0; o < n_c; ++o) {
                    const float zo = cz[o];
                    const int vo = cv[o];
                    rank += (zo > zi) | ((zo == zi) & (vo < vi));
                }
                if (rank < n_lab) { out[2 + rank] = zi - lse; out[2 + beam_k + rank] = __int_as_float(vi); }
            }
        } else {                                                     // a plateau: one wave-wide maximum per label
            if (EPT > 0) {
#pragma unroll
                for (int e = 0; e < ne; ++e) if (lane + 64 * e < V) zs[lane + 64 * e] = ZAT(e);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            float pz = INFINITY;
            int pv = -1;
            for (int j = 0; j < beam_k; ++j) {
                float bz = -INFINITY;
                int bv = -1;
                for (int v = lane; v < V; v += 64) {
                    const float zv = zs[v];
                    const bool after = (v != blank) & ((zv < pz) | ((zv == pz) & (v > pv)));
                    const bool take = after & ((bv < 0) | (zv > bz));
                    bz = take ? zv : bz;
                    bv = take ? v : bv;
                }
                wave_argmax(bz, bv);
                if (bv < 0) break;
                if (lane == 0) { out[2 + j] = bz - lse; out[2 + beam_k + j] = __int_as_float(bv); }
                pz = bz; pv = bv;
                ++n_lab;
            }
        }
#undef ZAT
        if (lane == 0) { out[0] = zb - lse; out[1] = __int_as_float(n_lab); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // zs / cz are rewritten by this wave's next row
    }
}

// ---- block-wide (256 threads) first maximum of the live entries of the LDS score list (dead = NaN) + the number of kept
// scores strictly above it (only counted when there are at least `beam` of them: the end-of-frame test cannot pass before).
// NEED_NO_REC: only entries without a record (lrec == -1) compete.  w_f / w_i / cnt are double-buffered by the parity of the
// call, so one barrier separates a call from the next. ----
struct ArgmaxScratch { float w_f[2][4]; int w_i[2][4]; int cnt[2]; };
template <bool NEED_NO_REC>
__device__ __forceinline__ void list_argmax_count(const float* lsc, const short* lrec, int n, const float* ks, int nk, int beam,
                                                  ArgmaxScratch* sc, int par, float& best, int& bi, int& n_good) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    best = -INFINITY; bi = -1;
    for (int i = tid; i < n; i += 256) {
        const float s = lsc[i];
        bool take = (s == s) & ((bi < 0) | (s > best));            // a thread meets its entries in ascending order
        if (NEED_NO_REC) take = take & (lrec[i] == -1);
        best = take ? s : best;
        bi = take ? i : bi;
    }
    wave_argmax(best, bi);
    if (lane == 0) { sc->w_f[par][wave] = best; sc->w_i[par][wave] = bi; }
    lds_barrier();
    best = sc->w_f[par][0]; bi = sc->w_i[par][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float os = sc->w_f[par][w];
        const int oi = sc->w_i[par][w];
        if (oi >= 0 && (bi < 0 || os > best || (os == best && oi < bi))) { best = os; bi = oi; }
    }
    if (tid == 0) sc->cnt[par ^ 1] = 0;                           // the next call's counter (its last readers are past the barrier above)
    n_good = 0;
    if (nk < beam) return;
    int good = 0;
    for (int i = tid; i < nk; i += 256) good += ks[i] > best;
    if (good) atomicAdd(&sc->cnt[par], good);
    lds_barrier();
    n_good = sc->cnt[par];
}

// ---- ask for the expansions the search will want next: the best KS open hypotheses that have no record at this frame, best
// first.  The first one is the hypothesis the pop loop stopped at; the others are a guess (their turn comes unless the frame ends
// first or new extensions overtake them — then their records simply wait or are dropped with the frame: asking changes no
// result).  A hypothesis that is not in the label trie yet gets its prediction-network evaluation (a slot for the result is taken
// now; the trie node is made when it is popped); one that is has its joint.pred vector cached.
// Block-wide; the caller has made this launch's global stores visible.  list = the joint work list the next iteration reads ----
__device__ __forceinline__ void schedule_evals(const BeamState& bs, const DecodeState& st, int b, int B, int L, int H, int J, const float* lsc,
                                               short* lrec, int nh, int t, int npop, int nfree, int nrec, ArgmaxScratch* sc, int par,
                                               int list) {
    __shared__ int c_bi[8], c_tok[8], c_slot[8];
    const int tid = threadIdx.x;
    const size_t hb = (size_t)b * bs.max_h;
    const int LH = L * H, KS = bs.KS, rpu = bs.R + KS, rows = B * rpu;
    int32_t* fl = bs.freelist + (size_t)b * bs.n_slots;
    // the candidates, best first (LDS only); a chosen one is marked so that the next round skips it
    int n_cand = 0;
    for (int k = 0; k < KS; ++k, par ^= 1) {
        float sc_k;
        int bi, dummy;
        list_argmax_count<true>(lsc, lrec, nh, nullptr, 0, 1, sc, par, sc_k, bi, dummy);
        if (bi < 0) break;
        if (tid == 0) { c_bi[k] = bi; lrec[bi] = -2; }
        ++n_cand;
        lds_barrier();
    }
    if (tid < n_cand) { c_tok[tid] = bs.h_tok[hb + c_bi[tid]]; c_slot[tid] = bs.h_slot[hb + c_bi[tid]]; }
    __syncthreads();
    // which of them are asked for: the first always; a guess only while records are left, and — if it needs a slot — while the
    // frame's remaining pops keep theirs (free slots >= pops the frame may still make: an invariant every branch here keeps,
    // which is also why the needed one always finds a slot)
    int n_acc = 0, n_lstm = 0;
    for (int k = 0; k < n_cand; ++k) {
        const bool is_new = c_tok[k] >= 0;
        if (k > 0 && (nrec >= bs.RP - 1 || (is_new && nfree - 1 < bs.max_pops - npop))) break;
        const int rp = k == 0 ? bs.RP - 1 : nrec++;                 // the needed one is consumed by the next launch: one record serves
        const int own = is_new ? fl[--nfree] : c_slot[k];
        if (tid == 0) {
            const int bi = c_bi[k], row = b * rpu + bs.R + k, lrow = b * KS + n_lstm;
            const size_t src_off = ((size_t)b * bs.n_slots + c_slot[k]) * (size_t)bs.slot_floats;
            lrec[bi] = (short)rp;
            bs.h_rec[hb + bi] = rp;
            bs.rec_slot[(size_t)b * bs.RP + rp] = own;
            bs.row_rec[row] = rp;
            st.tcur[row] = t;
            st.alive[(size_t)list * rows + atomicAdd(&st.counters[2 + list], 1)] = row;
            if (is_new) {
                st.token[lrow] = c_tok[k];
                st.act[atomicAdd(&st.counters[0], 1)] = lrow;
                bs.g_off[row] = (long long)lrow * J;               // joint.pred writes that row of the decode state
                bs.park_slot[(size_t)b * KS + n_lstm] = own;
                c_bi[k] = n_lstm;                                  // (reused below: the prediction-network row of candidate k)
            } else {
                bs.g_off[row] = bs.slots_off + (long long)src_off + 2 * LH;   // evaluated before: the cached vector
            }
        }
        n_lstm += is_new;
        ++n_acc;
    }
    if (tid == 0) {
        for (int k = n_acc; k < n_cand; ++k) lrec[c_bi[k]] = -1;     // not asked for after all
        bs.nrec[b] = nrec; bs.nfree[b] = nfree; bs.npark[b] = n_lstm;
    }
    lds_barrier();
    // start states of the new evaluations
    const int n_rows = B * KS;
    for (int k = 0; k < n_acc; ++k) {
        if (c_tok[k] < 0) continue;
        const float* src = bs.slots + ((size_t)b * bs.n_slots + c_slot[k]) * (size_t)bs.slot_floats;
        const int lrow = b * KS + c_bi[k];
        for (int i = tid; i < LH; i += 256) {
            const int l = i / H, u = i - l * H;
            st.h[((size_t)l * n_rows + lrow) * H + u] = src[i];
            st.c[((size_t)l * n_rows + lrow) * H + u] = src[LH + i];
        }
    }
}

// the first iteration's request: the start hypothesis; grid B, block 256, dynamic LDS like beam_step_kernel
__global__ __launch_bounds__(256) void beam_first_kernel(BeamState bs, DecodeState st, int B, int L, int H, int J) {
    extern __shared__ __attribute__((aligned(16))) char beam_smem[];
    float* lsc = reinterpret_cast<float*>(beam_smem);
    short* lrec = reinterpret_cast<short*>(lsc + bs.max_h);
    __shared__ ArgmaxScratch sc;
    const int b = blockIdx.x;
    if (bs.done[b]) return;
    if (threadIdx.x == 0) { lsc[0] = 0.0f; lrec[0] = -1; sc.cnt[0] = 0; sc.cnt[1] = 0; }
    __syncthreads();
    schedule_evals(bs, st, b, B, L, H, J, lsc, lrec, 1, 0, 0, bs.nfree[b], bs.nrec[b], &sc, 0, 0);
}

#define BEAM_MARK(phase)                                                                       \
    if (bs.trace && tid == 0) {                                                                \
        const unsigned long long now = wall_clock64();                                         \
        atomicAdd(&bs.trace[2 * (phase)], now - t_mark);                                       \
        atomicAdd(&bs.trace[2 * (phase) + 1], 1ull);                                           \
        t_mark = now;                                                                          \
    }

// grid B, block 256.  Dynamic LDS: open-list scores [max_h] f32 + records [max_h] i16, kept score / node / slot / len
// [max_pops] each, frame-start hypotheses' node / len [max_pops] each, slot marks [n_slots], records [RP * rec_floats], their
// slots [RP]
__global__ __launch_bounds__(256) void beam_step_kernel(BeamState bs, DecodeState st, const int32_t* __restrict__ enc_lens, int B,
                                                        int L, int H, int J, int beam, int score_norm, int out_cap, int iter,
                                                        int32_t* __restrict__ ids, int32_t* __restrict__ frames,
                                                        int32_t* __restrict__ n_ids, float* __restrict__ scores,
                                                        int32_t* __restrict__ pops) {
    extern __shared__ __attribute__((aligned(16))) char beam_smem[];
    const int MP = bs.max_pops, R = bs.R, KS = bs.KS, RP = bs.RP, RF = bs.rec_floats, K = bs.beam_k;
    float* lsc = reinterpret_cast<float*>(beam_smem);
    short* lrec = reinterpret_cast<short*>(lsc + bs.max_h);
    float* ks = reinterpret_cast<float*>(lrec + (bs.max_h + 1) / 2 * 2);
    int* kn = reinterpret_cast<int*>(ks + MP);
    int* ksl = kn + MP;
    int* kl = ksl + MP;
    int* sn = kl + MP;
    int* sl = sn + MP;
    int* used = sl + MP;
    float* recs = reinterpret_cast<float*>(used + bs.n_slots);
    int* rslot = reinterpret_cast<int*>(recs + (size_t)RP * RF);
    __shared__ ArgmaxScratch sc;
    __shared__ int s_nfree;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (bs.done[b]) return;
    unsigned long long t_mark = bs.trace ? wall_clock64() : 0ull;
    const int LH = L * H, SS = bs.slot_floats;
    const size_t hb = (size_t)b * bs.max_h, kb = (size_t)b * MP;
    float* pool = bs.slots + ((size_t)b * bs.n_slots) * (size_t)SS;
    int32_t* fl = bs.freelist + (size_t)b * bs.n_slots;
    const int t = bs.t[b], ninit = bs.ninit[b], nbatch = bs.nb[b], nrec = bs.nrec[b], npark = bs.npark[b];
    int nh = bs.nh[b], nk = bs.nk[b], npop = bs.npop[b], nnode = bs.nnode[b], pops_add = 0;
    const int nfree = bs.nfree[b];
    const int list = (iter + 1) & 1;

    // ---- this launch's view of the utterance, in LDS ----
    for (int i = tid; i < nh; i += 256) {
        lsc[i] = bs.h_alive[hb + i] ? bs.h_score[hb + i] : __int_as_float(0x7fc00000);
        lrec[i] = (short)bs.h_rec[hb + i];
    }
    for (int i = tid; i < nk; i += 256) { ks[i] = bs.k_score[kb + i]; kn[i] = bs.k_node[kb + i]; ksl[i] = bs.k_slot[kb + i]; kl[i] = bs.k_len[kb + i]; }
    for (int i = tid; i < ninit; i += 256) { sn[i] = bs.h_node[hb + i]; sl[i] = bs.h_len[hb + i]; }
    {   // the records of this frame: the batch [0, nbatch) and what was asked for since [R, nrec)
        const float* src = bs.rec + ((size_t)b * RP) * RF;
        for (int i = tid; i < nbatch * RF; i += 256) recs[i] = src[i];
        for (int i = R * RF + tid; i < nrec * RF; i += 256) recs[i] = src[i];
        for (int i = (RP - 1) * RF + tid; i < RP * RF; i += 256) recs[i] = src[i];
        for (int i = tid; i < RP; i += 256) rslot[i] = bs.rec_slot[(size_t)b * RP + i];
    }
    // the prediction-network results of the last iteration go to the slots taken for them
    {
        const int n_rows = B * KS, per = 2 * LH + J;
        for (int i = tid; i < npark * per; i += 256) {
            const int k = i / per, e = i - k * per;
            const int lrow = b * KS + k;
            float v;
            if (e < 2 * LH) {
                const int which = e >= LH, q = e - which * LH, l = q / H, u = q - l * H;
                v = (which ? st.c : st.h)[((size_t)l * n_rows + lrow) * H + u];
            } else {
                v = st.g[(size_t)lrow * J + (e - 2 * LH)];
            }
            pool[(size_t)bs.park_slot[(size_t)b * KS + k] * SS + e] = v;
        }
    }
    if (tid == 0) { sc.cnt[0] = 0; sc.cnt[1] = 0; }
    __syncthreads();
    BEAM_MARK(0)

    bool have = false;
    float e_score = 0.0f, hm;
    int e_node = 0, e_len = 0, e_own = 0, e_row = 0, bi, n_good, par = 0;
    for (;; par ^= 1) {
        if (have) {                                                  // ---- apply one pop: blank extension + label extensions ----
            const float* rc = recs + (size_t)e_row * RF;
            const int n_lab = __float_as_int(rc[1]);
            if (tid == 255) {
                const float sc_b = e_score + rc[0];
                ks[nk] = sc_b; kn[nk] = e_node; ksl[nk] = e_own; kl[nk] = e_len;
                bs.k_score[kb + nk] = sc_b; bs.k_node[kb + nk] = e_node; bs.k_slot[kb + nk] = e_own; bs.k_len[kb + nk] = e_len;
            }
            if (tid < n_lab) {
                const float sc_c = e_score + rc[2 + tid];
                const size_t o = hb + nh + tid;
                lsc[nh + tid] = sc_c; lrec[nh + tid] = -1;
                bs.h_score[o] = sc_c; bs.h_node[o] = e_node; bs.h_tok[o] = __float_as_int(rc[2 + K + tid]); bs.h_slot[o] = e_own;
                bs.h_len[o] = e_len + 1; bs.h_alive[o] = 1; bs.h_rec[o] = -1;
            }
            nh += n_lab; nk += 1; npop += 1; pops_add += 1;
            lds_barrier();
        }
        // end-of-frame test: at least `beam` kept entries strictly above the maximum of the open list — whose first maximum is
        // also the next hypothesis to pop if the frame goes on
        list_argmax_count<false>(lsc, lrec, nh, ks, nk, beam, &sc, par, hm, bi, n_good);
        if (n_good >= beam) break;
        if (bi < 0 || npop >= MP) { if (tid == 0) beam_fail(bs, b); return; }
        const int rp = lrec[bi];
        if (rp < 0) break;                                           // it needs an evaluation
        // ---- its record at this frame is here: pop it now ----
        if (bi < ninit) { e_node = sn[bi]; e_len = sl[bi]; }
        else {                                                       // a label extension (opened in an earlier launch): it enters the trie
            const int tok = bs.h_tok[hb + bi], parent = bs.h_node[hb + bi];
            e_len = bs.h_len[hb + bi];
            if (nnode >= bs.max_nodes) { if (tid == 0) beam_fail(bs, b); return; }
            if (tid == 0) { bs.nodes[(size_t)b * bs.max_nodes + nnode] = make_int2(parent, tok); bs.node_frame[(size_t)b * bs.max_nodes + nnode] = t; }
            e_node = nnode++;
        }
        if (tid == 0) { lsc[bi] = __int_as_float(0x7fc00000); bs.h_alive[hb + bi] = 0; }
        e_score = hm; e_own = rslot[rp]; e_row = rp;
        have = true;
    }
    if (n_good < beam) {
        // ---- hand the hypotheses the search wants next to the next iteration ----
        if (tid == 0) { bs.nh[b] = nh; bs.nk[b] = nk; bs.npop[b] = npop; bs.nnode[b] = nnode; bs.pops[b] += pops_add; }
        __syncthreads();                                             // entries appended in this launch are read back from global
        BEAM_MARK(1)
        schedule_evals(bs, st, b, B, L, H, J, lsc, lrec, nh, t, npop, nfree, nrec, &sc, par ^ 1, list);
        BEAM_MARK(2)
        return;
    }
    BEAM_MARK(3)
    // ---- end of frame: survivors ascending by score (ties in kept order) become the next frame's open list; every slot that
    // no survivor owns goes back to the free list (in any order: slot ids never reach a result).  The survivors are popped
    // best first (ties: lowest position): the first R in that order get the rows of the next frame's batch. ----
    for (int s = tid; s < bs.n_slots; s += 256) used[s] = s == 0;
    if (tid == 0) s_nfree = 0;
    lds_barrier();
    const int t_next = t + 1;
    const bool last = t_next >= enc_lens[b];
    const int rpu = R + KS, rows = B * rpu;
    for (int i = tid; i < nk; i += 256) {
        const float si = ks[i];
        if (!(si > hm)) continue;
        int rank = 0, desc = 0;
        for (int o = 0; o < nk; ++o) {
            const float so = ks[o];
            if (!(so > hm)) continue;
            rank += (so < si) | ((so == si) & (o < i));
            desc += (so > si) | ((so == si) & (o < i));
        }
        const int slot = ksl[i];
        if (last) {                                                  // by position: what the read-back below needs
            lsc[rank] = score_norm ? si / (float)kl[i] : si;
            sn[rank] = kn[i]; sl[rank] = kl[i]; used[rank] = __float_as_int(si);
        } else {
            used[slot] = 1;
            bs.h_score[hb + rank] = si;
            bs.h_node[hb + rank] = kn[i]; bs.h_tok[hb + rank] = -1; bs.h_slot[hb + rank] = slot;
            bs.h_len[hb + rank] = kl[i]; bs.h_alive[hb + rank] = 1;
            bs.h_rec[hb + rank] = desc < R ? desc : -1;
            if (desc < R) {
                const int row = b * rpu + desc;
                bs.rec_slot[(size_t)b * RP + desc] = slot;
                bs.row_rec[row] = desc;
                bs.g_off[row] = bs.slots_off + (long long)(((size_t)b * bs.n_slots + slot) * (size_t)SS) + 2 * LH;
                st.tcur[row] = t_next;
                st.alive[(size_t)list * rows + atomicAdd(&st.counters[2 + list], 1)] = row;
            }
        }
    }
    lds_barrier();
    BEAM_MARK(4)
    if (!last) {
        for (int s = tid; s < bs.n_slots; s += 256)
            if (!used[s]) fl[atomicAdd(&s_nfree, 1)] = s;
        lds_barrier();
        if (tid == 0) {
            bs.nh[b] = n_good; bs.nk[b] = 0; bs.npop[b] = 0; bs.nfree[b] = s_nfree; bs.ninit[b] = n_good; bs.t[b] = t_next;
            bs.nnode[b] = nnode; bs.pops[b] += pops_add;
            bs.nb[b] = n_good < R ? n_good : R; bs.nrec[b] = R; bs.npark[b] = 0;
        }
        BEAM_MARK(5)
        return;
    }
    // ---- last frame: the first maximum of score / len(yseq) (or of score) over the survivors in their order ----
    int dummy;
    list_argmax_count<false>(lsc, lrec, n_good, ks, 0, 1, &sc, par ^ 1, hm, bi, dummy);
    if (tid == 0) {
        const int n = sl[bi] - 1;
        scores[b] = __int_as_float(used[bi]);
        pops[b] = bs.pops[b] + pops_add;
        bs.done[b] = 1;
        if (n > out_cap) { n_ids[b] = 0; bs.flags[1] = 1; }
        else {
            int node = sn[bi];
            for (int q = n - 1; q >= 0; --q) {
                const int2 nd = bs.nodes[(size_t)b * bs.max_nodes + node];
                ids[(size_t)b * out_cap + q] = nd.y;
                if (frames) frames[(size_t)b * out_cap + q] = bs.node_frame[(size_t)b * bs.max_nodes + node];
                node = nd.x;
            }
            n_ids[b] = n;
        }
        atomicAdd(&bs.flags[0], 1);
    }
    BEAM_MARK(6)
}

struct BeamPlan {
    size_t b4, bk4, h4, k4, nodes, nodes4, slots, freelist, rec, rp4, goff, state1, g, rows4, z, apre, total, step_lds, rec_lds;
    int max_h, max_nodes, n_slots, slot_floats, zstride, R, KS, RP, rec_floats, rows;
};

BeamPlan beam_plan(const rs_ctx* ctx, int B, int beam, int beam_k, int tp_max, int max_pops) {
    const rs_dims& d = ctx->d;
    BeamPlan p;
    p.max_h = max_pops * (beam_k + 1) + 1;
    p.max_nodes = (tp_max > 0 ? tp_max : 1) * max_pops + 1;        // a pop adds at most one node
    p.n_slots = 2 * max_pops + 1;                                    // zero state + survivors (<= max_pops) + evaluations of a frame (<= max_pops)
    p.slot_floats = 2 * d.pred_layers * d.pred_hidden + d.joint_hidden;
    p.zstride = (d.n_logits + 63) / 64 * 64;
    int R = (beam + 12 + 3) / 4 * 4;                                 // a frame opens with >= beam survivors, usually a few more
    if (R > 64) R = 64;
    if (R > max_pops) R = max_pops;
    p.R = R;
    int KS = 3;                                                      // evaluations asked for per iteration (the first is needed, the rest are guesses)
    if (const char* e = getenv("RS_BEAM_SPEC")) KS = atoi(e);
    p.KS = KS < 1 ? 1 : KS > 8 ? 8 : KS;
    p.RP = R + 32 + 1;                                               // batch + guesses of a frame + the needed one
    p.rec_floats = 2 + 2 * beam_k;
    p.rows = B * (R + p.KS);
    p.b4 = rs_align((size_t)B * 4);
    p.bk4 = rs_align((size_t)B * p.KS * 4);
    p.h4 = rs_align((size_t)B * p.max_h * 4);
    p.k4 = rs_align((size_t)B * max_pops * 4);
    p.nodes = rs_align((size_t)B * p.max_nodes * 8);
    p.nodes4 = rs_align((size_t)B * p.max_nodes * 4);
    p.state1 = rs_align((size_t)d.pred_layers * B * p.KS * d.pred_hidden * 4);
    p.slots = rs_align((size_t)B * p.n_slots * p.slot_floats * 4);
    p.freelist = rs_align((size_t)B * p.n_slots * 4);
    p.rec = rs_align((size_t)B * p.RP * p.rec_floats * 4);
    p.rp4 = rs_align((size_t)B * p.RP * 4);
    p.goff = rs_align((size_t)p.rows * 8);
    p.g = rs_align((size_t)B * p.KS * d.joint_hidden * 4);
    p.rows4 = rs_align((size_t)p.rows * 4);
    p.z = rs_align((size_t)p.rows * p.zstride * 4);
    p.apre = rs_align((size_t)p.rows * d.joint_hidden * 4);
    p.total = 12 * p.b4 + 3 * p.bk4 + 7 * p.h4 + 4 * p.k4 + p.nodes + p.nodes4 + p.slots + p.freelist + p.rec + p.rp4 + p.goff + 4 * p.state1 + p.g +
              4 * p.rows4 + 2 * rs_align(64) + rs_align(256) + p.z + p.apre + 1024;
    p.step_lds = (size_t)p.max_h * 4 + (size_t)(p.max_h + 1) / 2 * 4 + (size_t)max_pops * 4 * 6 + (size_t)p.n_slots * 4 +
                 (size_t)p.RP * p.rec_floats * 4 + (size_t)p.RP * 4;
    p.rec_lds = (size_t)4 * (p.zstride + 256) * 4;
    return p;
}

int clamp_pops(int beam, int max_pops) { return max_pops > 0 ? max_pops : 16 * beam; }

}  // namespace

size_t rs_rnnt_beam_workspace_bytes_impl(const rs_ctx* ctx, int B, int beam, int tp_max, int max_pops) {
    const int V = ctx->d.n_logits;
    const int bm = beam < V ? beam : V, beam_k = bm < V - 1 ? bm : V - 1;
    return beam_plan(ctx, B, bm, beam_k, tp_max, clamp_pops(bm, max_pops)).total;
}

int rs_rnnt_beam_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam, int score_norm,
                      int max_pops, int out_cap, int32_t* ids, int32_t* frames, int32_t* n_ids, float* scores, int32_t* pops,
                      void* workspace, size_t workspace_bytes, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int L = d.pred_layers, H = d.pred_hidden, J = d.joint_hidden, V = d.n_logits;
    if (H % 128 || J % 128) return rs_fail(ctx, RS_EINVAL, "beam search: pred_hidden/joint_hidden must be multiples of 128");
    if (L < 1 || L > 4) return rs_fail(ctx, RS_EINVAL, "beam search: 1..4 LSTM layers supported");
    if (V < 2) return rs_fail(ctx, RS_EINVAL, "beam search: vocabulary of %d", V);
    const int bm = beam < V ? beam : V, beam_k = bm < V - 1 ? bm : V - 1;
    const int mp = clamp_pops(bm, max_pops);
    if (bm > 128) return rs_fail(ctx, RS_EINVAL, "beam search: beam size must be 1..128");
    if (mp < bm) return rs_fail(ctx, RS_EINVAL, "beam search: max_pops %d < beam %d (a frame needs at least `beam` pops)", mp, bm);
    const BeamPlan pl = beam_plan(ctx, B, bm, beam_k, tp_max, mp);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "beam search: workspace %zu < %zu", workspace_bytes, pl.total);
    if (pl.step_lds > 150 * 1024) return rs_fail(ctx, RS_EINVAL, "beam search: beam %d x max_pops %d needs %zu bytes of LDS (> 150 KB): lower max_pops", bm, mp, pl.step_lds);
    if (pl.rec_lds > 150 * 1024) return rs_fail(ctx, RS_EINVAL, "beam search: vocabulary %d exceeds the record kernel's LDS", V);
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* q = w; w += bytes; return q; };
    BeamState bs;
    DecodeState st;
    // every int32 / float bookkeeping array first (one memset), then the big buffers
    char* zero_from = w;
    bs.t = (int32_t*)take(pl.b4); bs.done = (int32_t*)take(pl.b4); bs.nh = (int32_t*)take(pl.b4); bs.nk = (int32_t*)take(pl.b4);
    bs.npop = (int32_t*)take(pl.b4); bs.nfree = (int32_t*)take(pl.b4); bs.ninit = (int32_t*)take(pl.b4);
    bs.nnode = (int32_t*)take(pl.b4); bs.pops = (int32_t*)take(pl.b4); bs.nb = (int32_t*)take(pl.b4);
    bs.nrec = (int32_t*)take(pl.b4); bs.npark = (int32_t*)take(pl.b4);
    bs.park_slot = (int32_t*)take(pl.bk4);
    bs.flags = (int32_t*)take(rs_align(64));
    int32_t* counters = (int32_t*)take(rs_align(64));
    unsigned long long* trace = (unsigned long long*)take(rs_align(256));
    bs.trace = getenv("RS_BEAM_TRACE") ? trace : nullptr;
    const size_t zero_bytes = (size_t)(w - zero_from);
    bs.h_score = (float*)take(pl.h4); bs.h_node = (int32_t*)take(pl.h4); bs.h_tok = (int32_t*)take(pl.h4);
    bs.h_slot = (int32_t*)take(pl.h4); bs.h_len = (int32_t*)take(pl.h4); bs.h_alive = (int32_t*)take(pl.h4);
    bs.h_rec = (int32_t*)take(pl.h4);
    bs.k_score = (float*)take(pl.k4); bs.k_node = (int32_t*)take(pl.k4); bs.k_slot = (int32_t*)take(pl.k4);
    bs.k_len = (int32_t*)take(pl.k4);
    bs.nodes = (int2*)take(pl.nodes);
    bs.node_frame = (int32_t*)take(pl.nodes4);
    bs.freelist = (int32_t*)take(pl.freelist);
    bs.rec = (float*)take(pl.rec);
    bs.rec_slot = (int32_t*)take(pl.rp4);
    bs.row_rec = (int32_t*)take(pl.rows4);
    bs.g_off = (long long*)take(pl.goff);
    st.g = (float*)take(pl.g);                                      // joint.pred rows of the LSTM launch; the slots follow in the same
    bs.slots = (float*)take(pl.slots);                               // allocation, so one base + offset addresses both
    bs.slots_off = (long long)(bs.slots - st.g);
    bs.max_h = pl.max_h; bs.max_pops = mp; bs.max_nodes = pl.max_nodes; bs.n_slots = pl.n_slots; bs.slot_floats = pl.slot_floats;
    bs.R = pl.R; bs.KS = pl.KS; bs.RP = pl.RP; bs.rec_floats = pl.rec_floats; bs.beam_k = beam_k;
    st.h = (float*)take(pl.state1); st.c = (float*)take(pl.state1);
    st.h_tmp = (float*)take(pl.state1); st.c_tmp = (float*)take(pl.state1);
    st.tcur = (int32_t*)take(pl.rows4);                              // per joint row
    st.sym = nullptr;
    st.token = (int32_t*)take(pl.bk4); st.act = (int32_t*)take(pl.bk4);   // per prediction-network row
    st.alive = (int32_t*)take(2 * pl.rows4);
    st.counters = counters;
    st.pmax = nullptr; st.pidx = nullptr; st.a16 = nullptr; st.anorm = nullptr;
    st.zapprox = (float*)take(pl.z);
    st.g_off = bs.g_off;
    float* a_pre = (float*)take(pl.apre);
    st.a_pre = a_pre;
    st.joint_act = d.joint_act;

    const bool rec_lds = getenv("RS_BEAM_RECORD_LDS") != nullptr;   // test hook: the any-vocabulary variant on a small one
    auto record = rec_lds ? beam_record_kernel<0> : V <= 64 * 16 ? beam_record_kernel<16> : V <= 64 * 48 ? beam_record_kernel<48> : beam_record_kernel<0>;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)beam_step_kernel, (int)pl.step_lds); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)beam_first_kernel, (int)pl.step_lds); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)record, (int)pl.rec_lds); rc != RS_OK) return rc;
    rs_prof_begin(ctx, RS_PROF_DECODE, s, 0.0, 0.0);
    RS_HIP(ctx, hipMemsetAsync(zero_from, 0, zero_bytes, s));
    // slot 0 of every utterance: the zero state the search starts from
    RS_HIP(ctx, hipMemset2DAsync(bs.slots, (size_t)pl.n_slots * pl.slot_floats * 4, 0, (size_t)pl.slot_floats * 4, B, s));
    hipLaunchKernelGGL(beam_init_kernel, dim3((B + 255) / 256), dim3(256), 0, s, bs, enc_lens, B, d.blank_id, n_ids, scores, pops);
    hipLaunchKernelGGL(beam_first_kernel, dim3(B), dim3(256), pl.step_lds, s, bs, st, B, L, H, J);
    RS_CHECK_LAUNCH(ctx, "beam init");

    const int CHUNK = 32;
    const long long max_iters = (long long)(tp_max > 0 ? tp_max : 1) * (mp + 1) + 1;
    // the joint walks its list with a fixed number of row tiles (the list holds between B and B * R rows); the act and record
    // kernels stride over it the same way
    const int joint_rts = pl.rows / 32 < 64 ? (pl.rows + 31) / 32 : 64;
    const int act_blocks = 512;
    const int rec_blocks = pl.rows / 4 < 1024 ? (pl.rows + 3) / 4 : 1024;
    const int rpu = pl.R + pl.KS;
    int32_t hf[2] = {0, 0};
    long long it = 0;
    bool finished = false;
    while (!finished && it < max_iters) {
        for (int c = 0; c < CHUNK; ++c, ++it) {
            const int step = (int)(it & 1);
            if (int rc = rs_rnnt_launch_lstm_pred(ctx, &st, B * pl.KS, s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_DECODE, s); return rc; }
            hipLaunchKernelGGL(beam_act_kernel, dim3(act_blocks), dim3(256), 0, s, st, joint_enc, a_pre, pl.rows, tp_max, J, rpu, step);
            if (int rc = rs_rnnt_launch_joint_logits_indirect(ctx, &st, joint_enc, pl.rows, joint_rts * 32, tp_max, rpu, step, s); rc != RS_OK) {
                rs_prof_end(ctx, RS_PROF_DECODE, s);
                return rc;
            }
            hipLaunchKernelGGL(record, dim3(rec_blocks), dim3(256), pl.rec_lds, s, bs, st, st.zapprox, pl.zstride, pl.rows, rpu, V, d.blank_id,
                               step);
            hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(256), pl.step_lds, s, bs, st, enc_lens, B, L, H, J, bm, score_norm, out_cap,
                               step, ids, frames, n_ids, scores, pops);
        }
        RS_CHECK_LAUNCH(ctx, "beam step");
        RS_HIP(ctx, hipMemcpyAsync(hf, bs.flags, sizeof hf, hipMemcpyDeviceToHost, s));
        RS_HIP(ctx, hipStreamSynchronize(s));
        finished = hf[0] >= B;
    }
    rs_prof_end(ctx, RS_PROF_DECODE, s);
    if (bs.trace) {                                                  // diagnostic: where the step kernel's workgroups spend their time
        unsigned long long tr[14];
        RS_HIP(ctx, hipMemcpy(tr, trace, sizeof tr, hipMemcpyDeviceToHost));
        static const char* names[7] = {"load state", "pops until an evaluation", "ask for evaluations", "pops until frame end", "rank survivors",
                                       "free slots", "read back"};
        for (int i = 0; i < 7; ++i)
            fprintf(stderr, "[beam trace] %-26s %10llu passes  %8.2f us each (100 MHz wall clock)\n", names[i], tr[2 * i + 1],
                    tr[2 * i + 1] ? (double)tr[2 * i] / (double)tr[2 * i + 1] / 100.0 : 0.0);
        fprintf(stderr, "[beam trace] %lld iterations\n", it);
    }
    if (!finished) return rs_fail(ctx, RS_ESTATE, "beam search: %d of %d utterances unfinished after %lld iterations", B - hf[0], B, it);
    if (hf[1]) return rs_fail(ctx, RS_EOVERFLOW, "beam search: a frame needed more than max_pops=%d prediction-network evaluations, or a result has more than out_cap=%d labels", mp, out_cap);
    return RS_OK;
}

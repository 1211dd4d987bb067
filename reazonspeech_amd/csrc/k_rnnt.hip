// k_rnnt.hip — RNN-T prediction network, joint network and batched greedy decode loop
// ([UPSTREAM] RNNTDecoder.predict, RNNTJoint.joint_after_projection,
// GreedyBatchedRNNTInfer with max_symbols).
//
// Everything here is EXACT float32 with a fixed accumulation order, so that greedy token ids can
// be compared bit-for-bit with oracle/rnnt_greedy.c:
//   * dot products run on v_mfma_f32_16x16x4_f32 (exact f32 fma chain, guide §3); every output is
//     S partial chains over contiguous K slices (S = 16 for the LSTM gates, 8 for the joint and the
//     prediction projection), slice s accumulating from 0 in the order
//         for u in 16-blocks: for e in 0..3: for kk in 0..3:  k = base_s + 16u + 4kk + e
//     (what a lane's float4 loads feed the MFMA), combined left to right ((p0 + p1) + p2) + ..,
//     then + bias.
//   * exp / sigmoid / tanh are the polynomial below (only +,-,*,/ and fmaf), not libm.
// This translation unit is compiled with -ffp-contract=off.
//
// One decode step = 5 launches:  joint+argmax partials -> finalize (state machine, emits tokens,
// builds the work lists) -> LSTM layer 0 -> LSTM layer 1 -> prediction projection (+ state commit).
// Work lists keep the step cost proportional to the rows that still need it: the joint runs over
// the rows still decoding (`alive`), the LSTM over the rows that just emitted a non-blank (`act`).
// Tiles are [32 rows] x [64 columns] so that operands fetched from L2 are reused 2-4x in registers.
#include <cstdio>
#include <cstdlib>

#include "k_rnnt_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
__global__ void rnnt_init_kernel(DecodeState st, const int32_t* __restrict__ enc_lens, int B, int blank,
                                 int32_t* __restrict__ n_ids) {
    // single workgroup: builds the initial work lists in row order
    __shared__ int n_alive_s;
    if (threadIdx.x == 0) n_alive_s = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        st.tcur[b] = 0; st.sym[b] = 0; st.token[b] = blank; st.act[b] = b;
        if (st.token2) st.token2[b] = -1;                 // sherpa-onnx: the context starts as [-1, blank]
        n_ids[b] = 0;
        if (enc_lens[b] > 0) st.alive[atomicAdd(&n_alive_s, 1)] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) { st.counters[0] = B; st.counters[1] = 0; st.counters[2] = n_alive_s; st.counters[3] = 0; }
}

// ---- LSTM layer: gates = [x ; h_prev] . [W_ih | W_hh]^T + b, cell update ---------------------------
// grid (H/16 unit tiles, ceil(B/32) row tiles); block 1024 = 16 waves = the 16 K slices of one
// [32 rows] x [16 units x 4 gates] output tile (rows come from the compacted `act` list).  Short
// per-wave chains matter more than anything else here: a decode step is a dependency chain of
// five small launches, so each kernel's latency is what the batch pays.
__global__ __launch_bounds__(1024) void rnnt_lstm_kernel(DecodeState st, int layer, int B, int H,
                                                        const float* __restrict__ embed,
                                                        const float* __restrict__ W /* [4H][2H] */,
                                                        const float* __restrict__ bias /* [4H] = b_ih + b_hh */) {
    extern __shared__ __attribute__((aligned(16))) char lstm_smem[];
    // [SPLITK_LSTM][4][32][16] floats = 128 KiB; 4 rows x 16 lanes = 64 consecutive floats per store: conflict free
    float (*part)[4][32][16] = reinterpret_cast<float (*)[4][32][16]>(lstm_smem);
    int* rows_s = reinterpret_cast<int*>(lstm_smem + SPLITK_LSTM * 4 * 32 * 16 * 4);
    const int n_act = st.counters[0];
    const int rt = blockIdx.y, ut = blockIdx.x;
    if (rt * 32 >= n_act) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 32) {
        const int i = rt * 32 + tid;
        rows_s[tid] = st.act[i < n_act ? i : n_act - 1];
    }
    __syncthreads();
    const int li = lane & 15, kk = lane >> 4;
    const int lr = lane >> 2, lc = lane & 3;             // load mapping: row, 16-byte chunk
    const int perm = 4 * (4 * li + kk);                   // bpermute source lane (bytes)
    const int K = 2 * H, kslice = K / SPLITK_LSTM, nkb = K / 16;
    const float* xsrc[2];
    const float* hsrc[2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
        const int row = rows_s[ri * 16 + lr];
        xsrc[ri] = layer == 0 ? embed + (size_t)st.token[row] * H : st.h_tmp + ((size_t)(layer - 1) * B + row) * H;
        hsrc[ri] = st.h + ((size_t)layer * B + row) * H;
    }
    // weights are stored fragment-major (weights.py: to_fragment_major): [n/16][k/16][lane][4]
    const float* wfrag[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) wfrag[gt] = W + ((size_t)((gt * H) / 16 + ut) * nkb) * 256 + lane * 4;
    f32x4_t acc[2][4];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) acc[ri][gt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // operands stream straight from L2 (row-strided 16-byte pieces): a 2-deep register prefetch keeps
    // two k-blocks of loads in flight under the 32 MFMAs of the current one (one wave per SIMD here,
    // so there is no other wave to hide the latency)
    struct Frag { float4 a[2], w[4]; };
    auto load = [&](int k0) {
        Frag fr;
        const int k = k0 + 4 * lc;
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
            fr.a[ri] = (k < H) ? *reinterpret_cast<const float4*>(xsrc[ri] + k)
                               : *reinterpret_cast<const float4*>(hsrc[ri] + (k - H));
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) fr.w[gt] = *reinterpret_cast<const float4*>(wfrag[gt] + (size_t)(k0 >> 4) * 256);
        return fr;
    };
    auto compute = [&](const Frag& fin) {
        Frag fr = fin;
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) fr.a[ri] = to_mfma_a_layout(fin.a[ri], perm);
        // e-major issue order: 8 independent accumulators between two MFMAs on the same one
        // (the f32 16x16x4 MFMA has a 40-cycle dependent latency but a 32-cycle issue interval)
#define RS_MFMA_E(c)                                                                                            \
        _Pragma("unroll") for (int ri = 0; ri < 2; ++ri) _Pragma("unroll") for (int gt = 0; gt < 4; ++gt)      \
            acc[ri][gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.a[ri].c, fr.w[gt].c, acc[ri][gt], 0, 0, 0);
        RS_MFMA_E(x) RS_MFMA_E(y) RS_MFMA_E(z) RS_MFMA_E(w)
#undef RS_MFMA_E
    };
    const int kbeg = wave * kslice, nblk = kslice / 16;
    {
        Frag f0 = load(kbeg);
        Frag f1 = nblk > 1 ? load(kbeg + 16) : f0;
        int u = 0;
        for (; u + 2 < nblk; ++u) {
            const Frag f2 = load(kbeg + 16 * (u + 2));
            compute(f0);
            f0 = f1;
            f1 = f2;
        }
        if (nblk >= 2) { compute(f0); f0 = f1; }
        compute(f0);
    }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][gt][ri * 16 + 4 * kk + r][li] = acc[ri][gt][r];
    __syncthreads();
    // epilogue: thread = (row i, unit j)
    if (tid >= 512) return;
    const int i = tid >> 4, j = tid & 15;
    if (rt * 32 + i >= n_act) return;
    const int brow = rows_s[i];
    const int unit = ut * 16 + j;
    float z[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
        float s = part[0][gt][i][j];
#pragma unroll
        for (int sl = 1; sl < SPLITK_LSTM; ++sl) s = s + part[sl][gt][i][j];
        z[gt] = s + bias[gt * H + unit];
    }
    const float ig = rs_sigmoidf(z[0]), fg = rs_sigmoidf(z[1]), gg = rs_tanhf(z[2]), og = rs_sigmoidf(z[3]);
    const size_t o = ((size_t)layer * B + brow) * H + unit;
    const float cn = fmaf(fg, st.c[o], ig * gg);
    st.c_tmp[o] = cn;
    st.h_tmp[o] = og * rs_tanhf(cn);
}

// (Round 5 tried a throughput variant for the beam searches' hundreds of rows: one wave per [32 x 64] tile walking the 16 K slices
// itself and folding them left to right in registers — bit-identical, no LDS reduction, weights fetched once per 128 rows.  It
// LOST to this kernel: ALSD 141.7 vs 135.4 ms per step, ESPnet beam-20 379 vs 311 ms per batch (profiles/r05g_lstm_tp_ab.txt):
// with one wave per SIMD the L2 round trip of every operand fragment is exposed, which the 16 short chains per tile here hide.
// Removed again; its skeleton is kept in profiles/r05g_lstm_tp_kernel.hip.txt.)
// ---- [32 rows] x [64 cols] tile of  out = W . a + bias  with the 8 K slices on the 8 waves --------------
// MODE 0: prediction projection g = W_p . h_top + b_p over the `act` rows (+ LSTM state commit)
// MODE 1: joint logits  W_o . relu(f[b][t_b] + g[b]) + b_o over the `alive` rows, per-tile argmax
// MODE 3: MODE 2 with row r's prediction vector at st.g + st.g_off[r] (the default beam search, k_rnnt_beam.hip: the vectors
//         stay where they are cached)
// MODE 5: MODE 1 with look-ahead (greedy search on small batches): an alive utterance brings rows_per_utt rows, its frames
//         t .. t + rows_per_utt - 1 under the prediction vector it has now; pmax / pidx rows are utterance * rows_per_utt + j
// MODE 4: MODE 3 with the activated rows act(f + g) precomputed in st.a_pre [row][K] (with a tanh joint every one of the
//         ceil(V / 64) column tiles recomputed the same 32 x K tanh values, about as many VALU cycles as the tile's MFMA cycles)
// MODE 2: the same logits written out in full (beam search, k_rnnt_alsd.hip): rows are hypotheses, `rows_per_utt`
//         consecutive rows share an utterance's encoder frames; logits go to st.zapprox with row stride 64 * n_ctiles
template <int MODE>
__global__ __launch_bounds__(512) void rnnt_tile_kernel(DecodeState st, const float* __restrict__ f, int B, int Tp,
                                                        int L, int H, int K, int N, const float* __restrict__ W,
                                                        const float* __restrict__ bias, int n_ctiles, int step,
                                                        int rows_per_utt) {
    extern __shared__ __attribute__((aligned(16))) char tile_smem[];
    float (*part)[32][65] = reinterpret_cast<float (*)[32][65]>(tile_smem);   // [SPLITK_TILE][32][65]
    int* rows_s = reinterpret_cast<int*>(tile_smem + SPLITK_TILE * 32 * 65 * 4);
    int rt = blockIdx.y;
    const int ct = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool WALK = MODE == 3 || MODE == 4;       // strides over the row tiles
    constexpr bool ARGMAX = MODE == 1 || MODE == 5;     // per-tile argmax instead of logits
    const int32_t* list = MODE == 0 ? st.act : st.alive + (size_t)(step & 1) * B;
    // MODE 5: every alive utterance contributes `rows_per_utt` rows, its next frames t, t + 1, ... under the SAME prediction vector
    const int la = MODE == 5 ? rows_per_utt : 1;
    const int n_rows = MODE == 0 ? st.counters[0] : st.counters[2 + (step & 1)] * la;
    if (MODE >= 1 && rt == 0 && ct == 0 && tid == 0) {   // lists that this step's finalize will build
        st.counters[0] = 0;
        st.counters[2 + ((step + 1) & 1)] = 0;
    }
    // MODE >= 1 is launched with gridDim.x rounded up to a multiple of 8 (joint_grid_x): workgroups go to the 8 XCDs round-robin
    // by linear id = ct + gridDim.x * rt, so with a multiple of 8 a column tile lands on XCD ct % 8 in EVERY row tile and each
    // XCD's L2 keeps its eighth of W_o (0.8 of 6.6 MB at V = 2600) across row tiles and across steps; with gridDim.x = 41 every
    // XCD pulled all of W_o through the fabric once per launch.
    if (MODE >= 1 && ct >= n_ctiles) return;
    // MODE 3 / 4 walk the row tiles with stride gridDim.y (its list can be anything from a handful of rows to rows_per_utt per
    // utterance, and the launch cannot know): every other mode has one row tile per workgroup and leaves after the first pass
    for (;; rt += gridDim.y) {
    if (rt * 32 >= n_rows) return;
    if (tid < 32) {
        int i = rt * 32 + tid;
        i = i < n_rows ? i : n_rows - 1;
        rows_s[tid] = MODE == 5 ? list[i / la] * la + i % la : list[i];
    }
    __syncthreads();
    const int li = lane & 15, kk = lane >> 4;
    const int lr = lane >> 2, lc = lane & 3;             // load mapping: row, 16-byte chunk
    const int perm = 4 * (4 * li + kk);                   // bpermute source lane (bytes)
    const float* asrc[2];
    const float* gsrc[2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
        const int row = rows_s[ri * 16 + lr];
        if (MODE == 0) {
            asrc[ri] = st.h_tmp + ((size_t)(L - 1) * B + row) * H;
            gsrc[ri] = nullptr;
        } else {
            if (MODE == 4) {
                asrc[ri] = st.a_pre + (size_t)row * K;
                gsrc[ri] = nullptr;
            } else if (MODE == 5) {
                const int utt = row / la;
                int t = st.tcur[utt] + row % la;                 // (frames past the utterance's end are computed and ignored)
                t = t < Tp ? t : Tp - 1;
                asrc[ri] = f + ((size_t)utt * Tp + t) * K;
                gsrc[ri] = st.g + (size_t)utt * K;
            } else {
                int t = st.tcur[row];
                t = t < Tp ? t : Tp - 1;
                const int utt = MODE >= 2 ? row / rows_per_utt : row;
                asrc[ri] = f + ((size_t)utt * Tp + t) * K;
                gsrc[ri] = MODE == 3 ? st.g + st.g_off[row] : st.g + (size_t)row * K;
            }
        }
    }
    // fragment-major weights ([ceil(N/16)][K/16][lane][4], rows past N are zero)
    const int nkb = K / 16, ntile = (N + 15) / 16;
    const float* wfrag[4];
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
        int tn = ct * 4 + cj;
        tn = tn < ntile ? tn : ntile - 1;
        wfrag[cj] = W + ((size_t)tn * nkb) * 256 + lane * 4;
    }
    f32x4_t acc[2][4];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) acc[ri][cj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    constexpr bool HAS_G = (MODE >= 1 && MODE <= 3) || MODE == 5;          // MODE 4 reads rows that are already act(f + g)
    struct Frag { float4 a[2], g[2], w[4]; };
    auto load = [&](int k0) {
        Frag fr;
        const int k = k0 + 4 * lc;
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            fr.a[ri] = *reinterpret_cast<const float4*>(asrc[ri] + k);
            if (HAS_G) fr.g[ri] = *reinterpret_cast<const float4*>(gsrc[ri] + k);
        }
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) fr.w[cj] = *reinterpret_cast<const float4*>(wfrag[cj] + (size_t)(k0 >> 4) * 256);
        return fr;
    };
    auto compute = [&](const Frag& fr) {
        float4 a[2];
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            a[ri] = fr.a[ri];
            if (HAS_G) {       // the joint activation of (f + g) is elementwise: apply it before the lane permutation
                if (st.joint_act) {            // ESPnet JointNetwork: tanh (the shared polynomial: bit-exact with the C oracle)
                    a[ri].x = rs_tanhf(a[ri].x + fr.g[ri].x); a[ri].y = rs_tanhf(a[ri].y + fr.g[ri].y);
                    a[ri].z = rs_tanhf(a[ri].z + fr.g[ri].z); a[ri].w = rs_tanhf(a[ri].w + fr.g[ri].w);
                } else {
                    a[ri].x = fmaxf(a[ri].x + fr.g[ri].x, 0.0f); a[ri].y = fmaxf(a[ri].y + fr.g[ri].y, 0.0f);
                    a[ri].z = fmaxf(a[ri].z + fr.g[ri].z, 0.0f); a[ri].w = fmaxf(a[ri].w + fr.g[ri].w, 0.0f);
                }
            }
            a[ri] = to_mfma_a_layout(a[ri], perm);
        }
#define RS_MFMA_E(c)                                                                                            \
        _Pragma("unroll") for (int ri = 0; ri < 2; ++ri) _Pragma("unroll") for (int cj = 0; cj < 4; ++cj)      \
            acc[ri][cj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ri].c, fr.w[cj].c, acc[ri][cj], 0, 0, 0);
        RS_MFMA_E(x) RS_MFMA_E(y) RS_MFMA_E(z) RS_MFMA_E(w)
#undef RS_MFMA_E
    };
    const int kslice = K / SPLITK_TILE, kbeg = wave * kslice, nblk = kslice / 16;
    {
        Frag f0 = load(kbeg);
        Frag f1 = nblk > 1 ? load(kbeg + 16) : f0;
        int u = 0;
        for (; u + 2 < nblk; ++u) {
            const Frag f2 = load(kbeg + 16 * (u + 2));
            compute(f0);
            f0 = f1;
            f1 = f2;
        }
        if (nblk >= 2) { compute(f0); f0 = f1; }
        compute(f0);
    }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][ri * 16 + 4 * kk + r][cj * 16 + li] = acc[ri][cj][r];
    __syncthreads();
    // thread = (row i, 8 consecutive columns); waves 4..7 are done
    if (tid >= 256) { if (WALK) { __syncthreads(); continue; } return; }
    const int i = tid >> 3, c0 = (tid & 7) * 8;
    const bool row_ok = rt * 32 + i < n_rows;
    const int brow = rows_s[i];
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int v = ct * 64 + c0 + e;
        float s = part[0][i][c0 + e];
#pragma unroll
        for (int sl = 1; sl < SPLITK_TILE; ++sl) s = s + part[sl][i][c0 + e];
        if (v < N) {
            s = s + bias[v];
            if (MODE == 0) { if (row_ok) st.g[(size_t)brow * N + v] = s; }
            else if (!ARGMAX) { if (row_ok) st.zapprox[(size_t)brow * (64 * n_ctiles) + v] = s; }
            else if (s > best) { best = s; best_idx = v; }
        }
    }
    if (ARGMAX) {
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {   // the 8 lanes of a row are adjacent
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(best_idx, off, 64);
            if (ov > best || (ov == best && oi < best_idx)) { best = ov; best_idx = oi; }
        }
        if ((tid & 7) == 0 && row_ok) {
            st.pmax[(size_t)brow * n_ctiles + ct] = best;
            st.pidx[(size_t)brow * n_ctiles + ct] = best_idx;
        }
    } else if (MODE == 0 && row_ok) {
        // commit this row's new LSTM state (each column tile copies its share of the H units)
        const int stride = gridDim.x * 64;
        for (int l = 0; l < L; ++l)
            for (int u = ct * 64 + c0; u < H; u += stride)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (u + e >= H) break;
                    const size_t o = ((size_t)l * B + brow) * H + u + e;
                    st.h[o] = st.h_tmp[o];
                    st.c[o] = st.c_tmp[o];
                }
    }
    if (!WALK) return;
    __syncthreads();                                     // `part` / `rows_s` are rewritten by the next row tile
    }
}

// ---- finalize: full argmax + greedy state machine; one wave per alive row ---------------------------
__global__ __launch_bounds__(256) void rnnt_finalize_kernel(DecodeState st, const int32_t* __restrict__ enc_lens,
                                                            int B, int n_ctiles, int blank, int max_symbols,
                                                            int u_max, int step, int32_t* __restrict__ ids,
                                                            int32_t* __restrict__ frames, int32_t* __restrict__ n_ids) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_alive = st.counters[2 + (step & 1)];
    if (slot >= n_alive) return;
    const int b = st.alive[(size_t)(step & 1) * B + slot];
    float val = -INFINITY;
    int idx = 0x7fffffff;
    for (int ctile = lane; ctile < n_ctiles; ctile += 64) {
        const float ov = st.pmax[(size_t)b * n_ctiles + ctile];
        const int oi = st.pidx[(size_t)b * n_ctiles + ctile];
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(val, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
    if (lane != 0) return;
    int t = st.tcur[b], sy = st.sym[b];
    bool emitted = false;
    if (idx == blank || idx == st.unk || idx == 0x7fffffff) {   // sentinel: every logit was NaN (cannot outrank -inf) -> treat as blank
        t += 1; sy = 0;
    } else {
        const int n = n_ids[b];
        if (n < u_max) { ids[(size_t)b * u_max + n] = idx; frames[(size_t)b * u_max + n] = t; n_ids[b] = n + 1; }
        else st.counters[1] = 1;
        if (st.token2) st.token2[b] = st.token[b];
        st.token[b] = idx;
        emitted = true;
        sy += 1;
        if (sy >= max_symbols) { t += 1; sy = 0; }
    }
    st.tcur[b] = t; st.sym[b] = sy;
    if (t < enc_lens[b]) {
        const int pos = atomicAdd(&st.counters[2 + ((step + 1) & 1)], 1);
        st.alive[(size_t)((step + 1) & 1) * B + pos] = b;
        if (emitted) { const int pa = atomicAdd(&st.counters[0], 1); st.act[pa] = b; }
    }
}



// ---- finalize with look-ahead (rnnt_tile_kernel<5>): frames t, t + 1, ... of an utterance were scored under the prediction
// vector it has now.  Blank at a frame means the vector is unchanged at the next one, so its row is exactly what the frame-by-
// frame loop would have computed: the rows are consumed in order up to and including the first label (whose successors are
// discarded: the vector changes).  Same ids and frames as one frame per step, in up to `la` times fewer steps. ----
__global__ __launch_bounds__(256) void rnnt_finalize_la_kernel(DecodeState st, const int32_t* __restrict__ enc_lens, int B, int n_ctiles,
                                                               int blank, int max_symbols, int u_max, int step, int la,
                                                               int32_t* __restrict__ ids, int32_t* __restrict__ frames,
                                                               int32_t* __restrict__ n_ids) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_alive = st.counters[2 + (step & 1)];
    if (slot >= n_alive) return;
    const int b = st.alive[(size_t)(step & 1) * B + slot];
    const int T = enc_lens[b];
    int t = st.tcur[b], sy = st.sym[b];
    bool emitted = false;
    for (int j = 0; j < la && t < T; ++j) {
        float val = -INFINITY;
        int idx = 0x7fffffff;
        const size_t row = (size_t)b * la + j;
        for (int ctile = lane; ctile < n_ctiles; ctile += 64) {
            const float ov = st.pmax[row * n_ctiles + ctile];
            const int oi = st.pidx[row * n_ctiles + ctile];
            if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(val, off, 64);
            const int oi = __shfl_xor(idx, off, 64);
            if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        if (idx == blank || idx == st.unk || idx == 0x7fffffff) { t += 1; sy = 0; continue; }
        if (lane == 0) {
            const int n = n_ids[b];
            if (n < u_max) { ids[(size_t)b * u_max + n] = idx; frames[(size_t)b * u_max + n] = t; n_ids[b] = n + 1; }
            else st.counters[1] = 1;
            if (st.token2) st.token2[b] = st.token[b];
            st.token[b] = idx;
        }
        emitted = true;
        sy += 1;
        if (sy >= max_symbols) { t += 1; sy = 0; }
        break;
    }
    if (lane != 0) return;
    st.tcur[b] = t; st.sym[b] = sy;
    if (t < T) {
        const int pos = atomicAdd(&st.counters[2 + ((step + 1) & 1)], 1);
        st.alive[(size_t)((step + 1) & 1) * B + pos] = b;
        if (emitted) { const int pa = atomicAdd(&st.counters[0], 1); st.act[pa] = b; }
    }
}

// ---- narrow-tile variants (the default) -----------------------------------------------------------------------
// A decode step's LSTM / projection work is tiny (~0.2 GFLOP per layer at 32 emitting rows); what a batch pays is
// the LATENCY of each launch.  The wide tiles above put a whole [32 x 64] tile's 2560 f32 MFMAs on ONE CU (16 us per
// launch in profiles/r02i_*): 40 workgroups busy, 216 CUs idle.  These variants keep the K slices on the waves of a
// workgroup (same accumulation order) but give every workgroup a single 16-column tile, so four times as many CUs
// share the step and each workgroup's chain is a quarter as long.
//   LSTM: the 16 columns of a workgroup are 4 units x 4 gates — weights are stored fragment-major in that permuted
//   row order ("pred.lstm{l}.w4": row ug*16 + gate*4 + u  <-  row gate*H + 4*ug + u), so the cell update still finds
//   its four gates in one workgroup.
__global__ __launch_bounds__(1024) void rnnt_lstm4_kernel(DecodeState st, int layer, int B, int H,
                                                         const float* __restrict__ embed,
                                                         const float* __restrict__ W4 /* fragment-major, permuted rows */,
                                                         const float* __restrict__ bias /* [4H] = b_ih + b_hh */) {
    __shared__ float part[SPLITK_LSTM][32][17];
    __shared__ int rows_s[32];
    const int n_act = st.counters[0];
    const int rt = blockIdx.y, ug = blockIdx.x;
    if (rt * 32 >= n_act) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 32) {
        const int i = rt * 32 + tid;
        rows_s[tid] = st.act[i < n_act ? i : n_act - 1];
    }
    __syncthreads();
    const int li = lane & 15, kk = lane >> 4;
    const int lr = lane >> 2, lc = lane & 3;
    const int perm = 4 * (4 * li + kk);
    const int K = 2 * H, kslice = K / SPLITK_LSTM, nkb = K / 16;
    const float* xsrc[2];
    const float* hsrc[2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
        const int row = rows_s[ri * 16 + lr];
        xsrc[ri] = layer == 0 ? embed + (size_t)st.token[row] * H : st.h_tmp + ((size_t)(layer - 1) * B + row) * H;
        hsrc[ri] = st.h + ((size_t)layer * B + row) * H;
    }
    const float* wfrag = W4 + ((size_t)ug * nkb) * 256 + lane * 4;
    f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
    const int kbeg = wave * kslice, nblk = kslice / 16;
    // every operand of the slice is requested up front (nblk <= 8: at most 24 float4 in flight per lane)
    float4 av[2][8], wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nblk) {
            const int k = kbeg + 16 * u + 4 * lc;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
                av[ri][u] = (k < H) ? *reinterpret_cast<const float4*>(xsrc[ri] + k) : *reinterpret_cast<const float4*>(hsrc[ri] + (k - H));
            wv[u] = *reinterpret_cast<const float4*>(wfrag + (size_t)((kbeg >> 4) + u) * 256);
        }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nblk) {
            const float4 a0 = to_mfma_a_layout(av[0][u], perm), a1 = to_mfma_a_layout(av[1][u], perm);
#define RS_MFMA_E(c)                                                                         \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.c, wv[u].c, acc[0], 0, 0, 0);   \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.c, wv[u].c, acc[1], 0, 0, 0);
            RS_MFMA_E(x) RS_MFMA_E(y) RS_MFMA_E(z) RS_MFMA_E(w)
#undef RS_MFMA_E
        }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][ri * 16 + 4 * kk + r][li] = acc[ri][r];
    __syncthreads();
    if (tid >= 128) return;                               // thread = (row i, unit u4)
    const int i = tid >> 2, u4 = tid & 3;
    if (rt * 32 + i >= n_act) return;
    const int brow = rows_s[i];
    const int unit = ug * 4 + u4;
    float z[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
        float sm = part[0][i][gt * 4 + u4];
#pragma unroll
        for (int sl = 1; sl < SPLITK_LSTM; ++sl) sm = sm + part[sl][i][gt * 4 + u4];
        z[gt] = sm + bias[gt * H + unit];
    }
    const float ig = rs_sigmoidf(z[0]), fg = rs_sigmoidf(z[1]), gg = rs_tanhf(z[2]), og = rs_sigmoidf(z[3]);
    const size_t o = ((size_t)layer * B + brow) * H + unit;
    const float cn = fmaf(fg, st.c[o], ig * gg);
    st.c_tmp[o] = cn;
    st.h_tmp[o] = og * rs_tanhf(cn);
}

// prediction projection g = W_p . h_top + b_p over the `act` rows, one 16-column tile per workgroup, the 8 K slices
// on its 8 waves; the workgroups of column tile 0 also commit the new LSTM state of their rows
__global__ __launch_bounds__(512) void rnnt_pred16_kernel(DecodeState st, int B, int L, int H, int N /* = J */,
                                                         const float* __restrict__ W, const float* __restrict__ bias) {
    __shared__ float part[SPLITK_TILE][32][17];
    __shared__ int rows_s[32];
    const int rt = blockIdx.y, ct = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_rows = st.counters[0];
    if (rt * 32 >= n_rows) return;
    if (tid < 32) {
        const int i = rt * 32 + tid;
        rows_s[tid] = st.act[i < n_rows ? i : n_rows - 1];
    }
    __syncthreads();
    const int li = lane & 15, kk = lane >> 4;
    const int lr = lane >> 2, lc = lane & 3;
    const int perm = 4 * (4 * li + kk);
    const int K = H, nkb = K / 16;
    const float* asrc[2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) asrc[ri] = st.h_tmp + ((size_t)(L - 1) * B + rows_s[ri * 16 + lr]) * H;
    const float* wfrag = W + ((size_t)ct * nkb) * 256 + lane * 4;
    f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
    const int kslice = K / SPLITK_TILE, kbeg = wave * kslice, nblk = kslice / 16;
    float4 av[2][8], wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nblk) {
            const int k = kbeg + 16 * u + 4 * lc;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) av[ri][u] = *reinterpret_cast<const float4*>(asrc[ri] + k);
            wv[u] = *reinterpret_cast<const float4*>(wfrag + (size_t)((kbeg >> 4) + u) * 256);
        }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nblk) {
            const float4 a0 = to_mfma_a_layout(av[0][u], perm), a1 = to_mfma_a_layout(av[1][u], perm);
#define RS_MFMA_E(c)                                                                         \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.c, wv[u].c, acc[0], 0, 0, 0);   \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.c, wv[u].c, acc[1], 0, 0, 0);
            RS_MFMA_E(x) RS_MFMA_E(y) RS_MFMA_E(z) RS_MFMA_E(w)
#undef RS_MFMA_E
        }
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][ri * 16 + 4 * kk + r][li] = acc[ri][r];
    __syncthreads();
    const int i = tid >> 4, c = tid & 15;                 // thread = (row i, column c)
    const bool row_ok = rt * 32 + i < n_rows;
    const int brow = rows_s[i];
    const int v = ct * 16 + c;
    if (row_ok && v < N) {
        float sm = part[0][i][c];
#pragma unroll
        for (int sl = 1; sl < SPLITK_TILE; ++sl) sm = sm + part[sl][i][c];
        st.g[(size_t)brow * N + v] = sm + bias[v];
    }
    if (row_ok) {
        // commit this row's new LSTM state: the column tiles share the H units of every layer
        const int nct = gridDim.x;
        for (int l = 0; l < L; ++l)
            for (int u = ct * 16 + c; u < H; u += nct * 16) {
                const size_t o = ((size_t)l * B + brow) * H + u;
                st.h[o] = st.h_tmp[o];
                st.c[o] = st.c_tmp[o];
            }
    }
}

// a = relu(f[b][t_b] + g[b]) of every alive slot, once per step: bf16 copy for the screening GEMM and ||a||_2
__global__ __launch_bounds__(256) void rnnt_prep_kernel(DecodeState st, const float* __restrict__ f, int B, int Tp, int J,
                                                        int step) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the lists this step's verify kernel will build
        st.counters[0] = 0;
        st.counters[2 + ((step + 1) & 1)] = 0;
    }
    if (slot >= st.counters[2 + (step & 1)]) return;
    const int b = st.alive[(size_t)(step & 1) * B + slot];
    int t = st.tcur[b];
    t = t < Tp ? t : Tp - 1;
    const float* fr = f + ((size_t)b * Tp + t) * J;
    const float* gr = st.g + (size_t)b * J;
    float ss = 0.0f;
    for (int k = 4 * lane; k < J; k += 256) {
        const float4 fv = *reinterpret_cast<const float4*>(fr + k), gv = *reinterpret_cast<const float4*>(gr + k);
        float a0, a1, a2, a3;
        if (st.joint_act) {       // tanh joint (ESPnet, Zipformer): the exact polynomial, as rnnt_tile_kernel applies it
            a0 = rs_tanhf(fv.x + gv.x); a1 = rs_tanhf(fv.y + gv.y); a2 = rs_tanhf(fv.z + gv.z); a3 = rs_tanhf(fv.w + gv.w);
        } else {
            a0 = fmaxf(fv.x + gv.x, 0.0f); a1 = fmaxf(fv.y + gv.y, 0.0f); a2 = fmaxf(fv.z + gv.z, 0.0f); a3 = fmaxf(fv.w + gv.w, 0.0f);
        }
        ss = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, fmaf(a3, a3, ss))));
        *reinterpret_cast<u16x4_t*>(st.a16 + (size_t)slot * J + k) = pack_bf16x4(a0, a1, a2, a3);
    }
    ss = wave_sum(ss);
    if (lane == 0) st.anorm[slot] = sqrtf(ss) * 1.0001f;        // any summation order is fine: the bound has slack
}

// screening GEMM of the joint (see "screened joint" below):
//   workgroup (column group of 64, 32 alive slots): the slots' bf16 operand rows -> LDS, then
//   z~[slot][v] = a . W16[v] + b_pad[v] on v_mfma_f32_16x16x32_bf16, one 16-column tile per wave.
__global__ __launch_bounds__(256) void rnnt_screen_kernel(DecodeState st, const float* __restrict__ f, int B, int Tp, int J,
                                                          int Vpad, const uint16_t* __restrict__ W16,
                                                          const float* __restrict__ bpad, int step) {
    extern __shared__ __attribute__((aligned(16))) char scr_smem[];
    const int ldrow = J * 2 + 16;                         // bytes per row: +16 keeps the 16-row fragment reads conflict free
    const int ct = blockIdx.x, rt = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_alive = st.counters[2 + (step & 1)];
    if (rt * 32 >= n_alive) return;
    // the wave's weight fragments do not depend on the rows: request them first (J / 32 <= 20 k-steps of 16 bytes)
    const int col0 = (ct * 4 + wave) * 16;
    const bool has_cols = col0 < Vpad;
    const int li = lane & 15, kc = lane >> 4;
    const int nks = J / 32;
    const uint16_t* wrow = W16 + (size_t)((has_cols ? col0 : 0) + li) * J + kc * 8;
    bf16x8_t wf[20];
#pragma unroll
    for (int q = 0; q < 20; ++q)
        if (q < nks) wf[q] = *reinterpret_cast<const bf16x8_t*>(wrow + q * 32);
    const float bv = bpad[(has_cols ? col0 : 0) + li];
    // the bf16 operand rows of this row tile (built once per step by rnnt_prep_kernel): 16-byte pieces into LDS
    {
        const int per_row = J / 8;                        // 16-byte pieces per row
        for (int idx = tid; idx < 32 * per_row; idx += 256) {
            const int r = idx / per_row, c = idx - r * per_row;
            const int i = rt * 32 + r;
            const uint4 v = *reinterpret_cast<const uint4*>(st.a16 + (size_t)(i < n_alive ? i : n_alive - 1) * J + c * 8);
            *reinterpret_cast<uint4*>(scr_smem + r * ldrow + c * 16) = v;
        }
    }
    __syncthreads();
    if (!has_cols) return;
    f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int q = 0; q < 20; ++q)
        if (q < nks) {
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(scr_smem + (ri * 16 + li) * ldrow + (q * 32 + kc * 8) * 2);
                acc[ri] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[q], acc[ri], 0, 0, 0);
            }
        }
    // D: lane = (column li, rows 4*kc + r)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int slot = rt * 32 + ri * 16 + 4 * kc + r;
            if (slot < n_alive) st.zapprox[(size_t)slot * Vpad + col0 + li] = acc[ri][r] + bv;
        }
}

// ---- screened joint ---------------------------------------------------------------------------------------
// The joint's output layer (640 -> 3001 per row and step, 0.98 GFLOP at 256 rows) is two thirds of the decode
// loop's work when it runs in exact float32.  Greedy decoding only needs its ARGMAX, so the exact evaluation is
// restricted to the columns that can possibly win:
//   1. rnnt_screen_kernel a = relu(f[b][t_b] + g[b]) per alive row (bf16 tile in LDS, ||a||_2), then
//                         z~ = bf16(a) . bf16(W_o)^T + b_o   (bf16 MFMA, f32 accumulate: 1/16 of the f32 MFMA cost)
//   2. rnnt_verify_kernel per row: m = max z~; every column with z~_v >= m - 2 eps is a candidate, where
//        |z~_v - z_v| <= eps = 2^-7 * 1.25 * ||a|| * max_v ||w_v||
//      (bf16 rounding of both operands is <= 2^-8 relative each, the products are exact in f32, Cauchy-Schwarz bounds
//      sum |a_k w_vk|, the factor 1.25 covers the f32 accumulation error of both sums).  The true argmax of the exact
//      logits is therefore always a candidate; each candidate's logit is then recomputed in EXACT float32 with the
//      accumulation order documented at the top of this file, and the argmax (lowest index on ties) of those exact
//      values is taken.  The result is bit-identical to evaluating all 3001 columns exactly (oracle/rnnt_greedy.c);
//      the number of candidates only changes the cost (typically 1-3; every column in the worst case).
//   The same kernel then runs the greedy state machine for the row (emit / advance / work lists).
// (NCH == 0: any V — the approximate logits are scanned from memory twice and the candidates are evaluated in batches of up to
//  VER_CAP; the Zipformer family's 10 720 symbols.)  The error bound does not depend on the activation: a = act(f + g) is
//  computed exactly (ReLU, or the shared tanh polynomial) and only then rounded for the screening product.
constexpr int VER_CAP = 1024;

// exact float32 logits of the candidate columns cand_s[0 .. n_cand) of one row (a_s = its exact act(f + g) in LDS), one wave:
// running argmax (lowest index on ties) into best / best_idx.  8 candidates per pass x 8 K slices.
__device__ __forceinline__ void verify_evaluate(const int* cand_s, int n_cand, const float* a_s, const float* __restrict__ Wrm,
                                                const float* __restrict__ bo, int J, int lane, float& best, int& best_idx) {
    const int cgrp = lane >> 3, sl = lane & 7;
    const int kslice = J / SPLITK_TILE, nblk = kslice / 16;       // nblk <= 8 (launcher)
    for (int c0 = 0; c0 < n_cand; c0 += 8) {                      // wave-uniform trip count
        const bool valid = c0 + cgrp < n_cand;
        const int cand = valid ? cand_s[c0 + cgrp] : 0;
        const float* w = Wrm + (size_t)cand * J + sl * kslice;
        const float* as = a_s + sl * kslice;
        float4 wv[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nblk) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wv[u][kk] = *reinterpret_cast<const float4*>(w + 16 * u + 4 * kk);
            }
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nblk) {
                // a 16-block: float4 loads give (e = 0..3) of each kk; the chain runs e-major, kk-minor
                float4 av[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) av[kk] = *reinterpret_cast<const float4*>(as + 16 * u + 4 * kk);
#define RS_CHAIN_E(cc) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) acc = fmaf(av[kk].cc, wv[u][kk].cc, acc);
                RS_CHAIN_E(x) RS_CHAIN_E(y) RS_CHAIN_E(z) RS_CHAIN_E(w)
#undef RS_CHAIN_E
            }
        // partial chains combined left to right by the group's first lane, then + bias
        float sum = __shfl(acc, lane & ~7, 64);
#pragma unroll
        for (int q = 1; q < SPLITK_TILE; ++q) sum = sum + __shfl(acc, (lane & ~7) + q, 64);
        float val = valid ? sum + bo[cand] : -INFINITY;
        int idx = valid ? cand : 0x7fffffff;
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {                  // argmax over the 8 groups (exact values, lowest index on ties)
            const float ov = __shfl_xor(val, off, 64);
            const int oi = __shfl_xor(idx, off, 64);
            if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        if (val > best || (val == best && idx < best_idx)) { best = val; best_idx = idx; }
    }
}

// the row's exact a = act(f[b][t] + g[b]) into LDS: `nthr` threads, thread `tid`
__device__ __forceinline__ void verify_act_row(const DecodeState& st, const float* fr, const float* gr, float* a_s, int J, int tid, int nthr) {
    for (int k = 4 * tid; k < J; k += 4 * nthr) {
        const float4 fv = *reinterpret_cast<const float4*>(fr + k), gv = *reinterpret_cast<const float4*>(gr + k);
        if (st.joint_act)
            *reinterpret_cast<float4*>(a_s + k) = make_float4(rs_tanhf(fv.x + gv.x), rs_tanhf(fv.y + gv.y), rs_tanhf(fv.z + gv.z), rs_tanhf(fv.w + gv.w));
        else
            *reinterpret_cast<float4*>(a_s + k) = make_float4(fmaxf(fv.x + gv.x, 0.0f), fmaxf(fv.y + gv.y, 0.0f),
                                                              fmaxf(fv.z + gv.z, 0.0f), fmaxf(fv.w + gv.w, 0.0f));
    }
}

// greedy state machine of one row (identical to rnnt_finalize_kernel); one thread
__device__ __forceinline__ void verify_commit(const DecodeState& st, int b, int t, int idx, const int32_t* __restrict__ enc_lens, int B,
                                              int blank, int max_symbols, int u_max, int step, int32_t* __restrict__ ids,
                                              int32_t* __restrict__ frames, int32_t* __restrict__ n_ids) {
    int sy = st.sym[b];
    bool emitted = false;
    if (idx == blank || idx == st.unk || idx == 0x7fffffff) {
        t += 1; sy = 0;
    } else {
        const int n = n_ids[b];
        if (n < u_max) { ids[(size_t)b * u_max + n] = idx; frames[(size_t)b * u_max + n] = t; n_ids[b] = n + 1; }
        else st.counters[1] = 1;
        if (st.token2) st.token2[b] = st.token[b];
        st.token[b] = idx;
        emitted = true;
        sy += 1;
        if (sy >= max_symbols) { t += 1; sy = 0; }
    }
    st.tcur[b] = t; st.sym[b] = sy;
    if (t < enc_lens[b]) {
        const int pos = atomicAdd(&st.counters[2 + ((step + 1) & 1)], 1);
        st.alive[(size_t)((step + 1) & 1) * B + pos] = b;
        if (emitted) { const int pa = atomicAdd(&st.counters[0], 1); st.act[pa] = b; }
    }
}

template <int NCH>   // one WAVE per row.  NCH > 0: NCH * 64 >= V and the row's approximate logits live in registers
__global__ __launch_bounds__(256) void rnnt_verify_kernel(DecodeState st, const float* __restrict__ f,
                                                          const int32_t* __restrict__ enc_lens, int B, int Tp, int J, int V,
                                                          int Vpad, const float* __restrict__ Wrm /* [V][J] */,
                                                          const float* __restrict__ bo, const float* __restrict__ wmax, int blank,
                                                          int max_symbols, int u_max, int step, int32_t* __restrict__ ids,
                                                          int32_t* __restrict__ frames, int32_t* __restrict__ n_ids) {
    extern __shared__ __attribute__((aligned(16))) char ver_smem[];          // [4 waves][J] float: the exact a = act(f + g)
    constexpr int CAND_ROWS = NCH > 0 ? NCH * 64 : VER_CAP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    const int n_alive = st.counters[2 + (step & 1)];
    if (slot >= n_alive) return;
    const int b = st.alive[(size_t)(step & 1) * B + slot];
    const float* z = st.zapprox + (size_t)slot * Vpad;
    const int nchunk = (V + 63) / 64;
    float zr[NCH > 0 ? NCH : 1];
    float m = -INFINITY;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) zr[c] = (c * 64 + lane < V) ? z[c * 64 + lane] : -INFINITY;
#pragma unroll
        for (int c = 0; c < NCH; ++c) m = fmaxf(m, zr[c]);
    } else {
        for (int c = 0; c < nchunk; ++c) m = fmaxf(m, (c * 64 + lane < V) ? z[c * 64 + lane] : -INFINITY);
    }
    m = wave_max(m);
    const int t = st.tcur[b];
    const int tc = t < Tp ? t : Tp - 1;
    float* a_s = reinterpret_cast<float*>(ver_smem) + wave * J;
    verify_act_row(st, f + ((size_t)b * Tp + tc) * J, st.g + (size_t)b * J, a_s, J, lane, 64);
    // 2 eps = 2 * 2^-7 * 1.25 * ||a|| * max_v ||w_v||  (wmax[0] holds the largest row norm of W_o, rounded up)
    const float thr = m - 0.01953125f * wmax[0] * st.anorm[slot];
    // candidate columns, ascending, compacted into LDS (the unrolled part stays tiny: the exact evaluation exists once in
    // the instruction stream)
    int* cand_s = reinterpret_cast<int*>(ver_smem + 4 * J * 4) + wave * CAND_ROWS;
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    int n_cand = 0;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool is = zr[c] >= thr;                         // -inf padding never passes (thr is finite)
            const unsigned long long mask = __ballot(is);
            if (is) cand_s[n_cand + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = c * 64 + lane;
            n_cand += __builtin_popcountll(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        verify_evaluate(cand_s, n_cand, a_s, Wrm, bo, J, lane, best, best_idx);
    } else {
        for (int c = 0; c < nchunk; ++c) {
            const bool is = c * 64 + lane < V && z[c * 64 + lane] >= thr;
            const unsigned long long mask = __ballot(is);
            if (is) cand_s[n_cand + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = c * 64 + lane;
            n_cand += __builtin_popcountll(mask);
            if (n_cand + 64 > VER_CAP || c == nchunk - 1) {       // wave-uniform: the batch is full (or the scan is over)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                verify_evaluate(cand_s, n_cand, a_s, Wrm, bo, J, lane, best, best_idx);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                n_cand = 0;
            }
        }
    }
    if (lane != 0) return;
    verify_commit(st, b, t, best_idx, enc_lens, B, blank, max_symbols, u_max, step, ids, frames, n_ids);
}

// One WORKGROUP per row: wave w owns the columns of chunks [w * NCHW, (w + 1) * NCHW) — V <= 4 * NCHW * 64 — with its part of the
// approximate logits in registers (one pass over memory, every load independent: the one-wave scan of the Zipformer family's
// 10 720 columns was 168 dependent load round trips, 86 us per step), compacts and evaluates its own candidates, and the four
// (value, index) pairs meet in LDS.  The argmax with the lowest index on ties does not depend on how the candidates are grouped.
template <int NCHW>
__global__ __launch_bounds__(256) void rnnt_verify_wide_kernel(DecodeState st, const float* __restrict__ f,
                                                               const int32_t* __restrict__ enc_lens, int B, int Tp, int J, int V,
                                                               int Vpad, const float* __restrict__ Wrm /* [V][J] */,
                                                               const float* __restrict__ bo, const float* __restrict__ wmax, int blank,
                                                               int max_symbols, int u_max, int step, int32_t* __restrict__ ids,
                                                               int32_t* __restrict__ frames, int32_t* __restrict__ n_ids) {
    extern __shared__ __attribute__((aligned(16))) char ver_smem[];          // [J] float a | [4][NCHW * 64] candidates | [4] max, [4] best, [4] index
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x;
    const int n_alive = st.counters[2 + (step & 1)];
    if (slot >= n_alive) return;                                   // workgroup-uniform
    const int b = st.alive[(size_t)(step & 1) * B + slot];
    const float* z = st.zapprox + (size_t)slot * Vpad;
    float zr[NCHW];
#pragma unroll
    for (int c = 0; c < NCHW; ++c) {
        const int col = (wave * NCHW + c) * 64 + lane;
        zr[c] = col < V ? z[col] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCHW; ++c) m = fmaxf(m, zr[c]);
    m = wave_max(m);
    float* a_s = reinterpret_cast<float*>(ver_smem);
    int* cand_s = reinterpret_cast<int*>(ver_smem + (size_t)J * 4) + wave * (NCHW * 64);
    float* red = reinterpret_cast<float*>(ver_smem + (size_t)J * 4 + (size_t)4 * NCHW * 64 * 4);
    const int t = st.tcur[b];
    const int tc = t < Tp ? t : Tp - 1;
    verify_act_row(st, f + ((size_t)b * Tp + tc) * J, st.g + (size_t)b * J, a_s, J, threadIdx.x, 256);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float thr = m - 0.01953125f * wmax[0] * st.anorm[slot];
    int n_cand = 0;
#pragma unroll
    for (int c = 0; c < NCHW; ++c) {
        const bool is = zr[c] >= thr;
        const unsigned long long mask = __ballot(is);
        if (is) cand_s[n_cand + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (wave * NCHW + c) * 64 + lane;
        n_cand += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    verify_evaluate(cand_s, n_cand, a_s, Wrm, bo, J, lane, best, best_idx);
    if (lane == 0) { red[4 + wave] = best; reinterpret_cast<int*>(red)[8 + wave] = best_idx; }
    __syncthreads();
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float v = red[4 + w];
        const int i = reinterpret_cast<int*>(red)[8 + w];
        if (v > best || (v == best && i < best_idx)) { best = v; best_idx = i; }
    }
    verify_commit(st, b, t, best_idx, enc_lens, B, blank, max_symbols, u_max, step, ids, frames, n_ids);
}

}  // namespace


namespace {

// ---- stateless decoder of the Zipformer family ([UPSTREAM] icefall decoder.py Decoder.forward(need_pad=False)): embedding of
// the last two tokens (-1 embeds to zero), grouped Conv1d over them (groups = D / 4: output channel c sees input channels
// 4 (c / 4) .. + 3 of both tokens, no bias), ReLU -> h_tmp[0][row][D], the operand of the joiner's decoder_proj
// (rnnt_pred16_kernel).  Exact float32 in a fixed order (token before last first, then channel), mirrored by oracle/k2_greedy.c.
// One workgroup per `act` row.
__global__ __launch_bounds__(256) void k2_decoder_kernel(DecodeState st, int B, int D, const float* __restrict__ embed,
                                                         const float* __restrict__ conv_w /* [D][4][2] */) {
    const int n_rows = st.counters[0];
    if ((int)blockIdx.x >= n_rows) return;
    const int row = st.act[blockIdx.x];
    const int t0 = st.token2[row], t1 = st.token[row];
    for (int c = threadIdx.x; c < D; c += 256) {
        const int g4 = c & ~3;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e = t0 >= 0 ? embed[(size_t)t0 * D + g4 + i] : 0.0f;
            acc = fmaf(conv_w[(c * 4 + i) * 2 + 0], e, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e = t1 >= 0 ? embed[(size_t)t1 * D + g4 + i] : 0.0f;
            acc = fmaf(conv_w[(c * 4 + i) * 2 + 1], e, acc);
        }
        st.h_tmp[(size_t)row * D + c] = fmaxf(acc, 0.0f);
    }
}

constexpr int LSTM_LDS = SPLITK_LSTM * 4 * 32 * 16 * 4 + 32 * 4;
constexpr int TILE_LDS = SPLITK_TILE * 32 * 65 * 4 + 32 * 4;

// narrow-tile LSTM / projection kernels: need the permuted weights and keep a whole K slice in registers
bool narrow_kernels_usable(const rs_ctx* ctx) {
    const int L = ctx->d.pred_layers, H = ctx->d.pred_hidden;
    bool narrow = ctx->decode_narrow;
    for (int l = 0; l < L; ++l) narrow = narrow && ctx->lstm_w4[l] != nullptr;
    return narrow && 2 * H / SPLITK_LSTM / 16 <= 8 && H / SPLITK_TILE / 16 <= 8;
}

// prediction network over the `act` rows: LSTM layers, joint.pred projection, state commit
void launch_lstm_pred(rs_ctx* ctx, const DecodeState& st, int B, int rows_bound, bool narrow, hipStream_t s) {
    const int L = ctx->d.pred_layers, H = ctx->d.pred_hidden, J = ctx->d.joint_hidden;
    const int rts = (rows_bound + 31) / 32 > 0 ? (rows_bound + 31) / 32 : 1;
    if (ctx->k2_conv_w) {                         // Zipformer family: the prediction network is the stateless decoder
        hipLaunchKernelGGL(k2_decoder_kernel, dim3(rows_bound > 0 ? rows_bound : 1), dim3(256), 0, s, st, B, H, ctx->embed, ctx->k2_conv_w);
        hipLaunchKernelGGL(rnnt_pred16_kernel, dim3((J + 15) / 16, rts), dim3(512), 0, s, st, B, 1, H, J, ctx->jpred_w, ctx->jpred_b);
        return;
    }
    if (narrow) {
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL(rnnt_lstm4_kernel, dim3(H / 4, rts), dim3(1024), 0, s, st, l, B, H, ctx->embed, ctx->lstm_w4[l],
                               ctx->lstm_b[l]);
        hipLaunchKernelGGL(rnnt_pred16_kernel, dim3((J + 15) / 16, rts), dim3(512), 0, s, st, B, L, H, J, ctx->jpred_w,
                           ctx->jpred_b);
        return;
    }
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL(rnnt_lstm_kernel, dim3(H / 16, rts), dim3(1024), LSTM_LDS, s, st, l, B, H, ctx->embed,
                           ctx->lstm_w[l], ctx->lstm_b[l]);
    hipLaunchKernelGGL(rnnt_tile_kernel<0>, dim3((J + 63) / 64, rts), dim3(512), TILE_LDS, s, st, (const float*)nullptr, B,
                       0, L, H, H, J, ctx->jpred_w, ctx->jpred_b, 0, 0, 1);
}

// column-tile extent of a joint launch: a multiple of 8 when there is more than one row tile (see rnnt_tile_kernel)
int joint_grid_x(int nct, int rts) {
    static const bool off = getenv("RS_DECODE_NO_XCD") != nullptr;     // A/B hook
    return rts > 1 && !off ? (nct + 7) / 8 * 8 : nct;
}

int ensure_decode_lds(rs_ctx* ctx) {
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)rnnt_lstm_kernel, LSTM_LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)rnnt_tile_kernel<0>, TILE_LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)rnnt_tile_kernel<1>, TILE_LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)rnnt_tile_kernel<2>, TILE_LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)rnnt_tile_kernel<3>, TILE_LDS); rc != RS_OK) return rc;
    if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)rnnt_tile_kernel<4>, TILE_LDS); rc != RS_OK) return rc;
    return rs_ensure_dynamic_lds(ctx, (const void*)rnnt_tile_kernel<5>, TILE_LDS);
}

}  // namespace

// ---- launch helpers for the beam search (k_rnnt_alsd.hip); `st_ptr` points at a DecodeState over hypothesis rows ----
int rs_rnnt_launch_lstm_pred(rs_ctx* ctx, const void* st_ptr, int rows, hipStream_t s) {
    if (int rc = ensure_decode_lds(ctx); rc != RS_OK) return rc;
    launch_lstm_pred(ctx, *reinterpret_cast<const DecodeState*>(st_ptr), rows, rows, narrow_kernels_usable(ctx), s);
    return RS_OK;
}

// exact f32 joint logits of the rows on alive list (step & 1) -> st.zapprox [rows][64 * ceil(V / 64)]
int rs_rnnt_launch_joint_logits(rs_ctx* ctx, const void* st_ptr, const float* joint_enc, int rows, int tp_max, int rows_per_utt,
                                int step, hipStream_t s) {
    if (int rc = ensure_decode_lds(ctx); rc != RS_OK) return rc;
    const rs_dims& d = ctx->d;
    const int nct = (d.n_logits + 63) / 64;
    hipLaunchKernelGGL(rnnt_tile_kernel<2>, dim3(joint_grid_x(nct, (rows + 31) / 32), (rows + 31) / 32), dim3(512), TILE_LDS, s,
                       *reinterpret_cast<const DecodeState*>(st_ptr), joint_enc, rows, tp_max, d.pred_layers, d.pred_hidden,
                       d.joint_hidden, d.n_logits, ctx->jout_w, ctx->jout_b, nct, step, rows_per_utt);
    return RS_OK;
}

// the same with indirect prediction vectors (st.g_off); `rows` = extent of the row space (the alive lists have that stride)
int rs_rnnt_launch_joint_logits_indirect(rs_ctx* ctx, const void* st_ptr, const float* joint_enc, int rows, int rows_bound, int tp_max,
                                         int rows_per_utt, int step, hipStream_t s) {
    if (int rc = ensure_decode_lds(ctx); rc != RS_OK) return rc;
    const rs_dims& d = ctx->d;
    const int nct = (d.n_logits + 63) / 64;
    const int rts = (rows_bound + 31) / 32 > 0 ? (rows_bound + 31) / 32 : 1;
    const DecodeState& st = *reinterpret_cast<const DecodeState*>(st_ptr);
    if (st.a_pre)
        hipLaunchKernelGGL(rnnt_tile_kernel<4>, dim3(joint_grid_x(nct, rts), rts), dim3(512), TILE_LDS, s, st, joint_enc, rows, tp_max,
                           d.pred_layers, d.pred_hidden, d.joint_hidden, d.n_logits, ctx->jout_w, ctx->jout_b, nct, step, rows_per_utt);
    else
        hipLaunchKernelGGL(rnnt_tile_kernel<3>, dim3(joint_grid_x(nct, rts), rts), dim3(512), TILE_LDS, s, st, joint_enc, rows, tp_max,
                           d.pred_layers, d.pred_hidden, d.joint_hidden, d.n_logits, ctx->jout_w, ctx->jout_b, nct, step, rows_per_utt);
    return RS_OK;
}

// --------------------------------------------------------------------------------------------------
// frames an utterance may look ahead per greedy step (rnnt_tile_kernel<5>): the pmax / pidx scratch is sized for it
constexpr int LOOKAHEAD_MAX = 8;

size_t rs_rnnt_workspace_bytes(const rs_ctx* ctx, int B) {
    const rs_dims& d = ctx->d;
    const int L = d.pred_layers, H = d.pred_hidden, J = d.joint_hidden;
    const int nct = (d.n_logits + 63) / 64;
    size_t n = 0;
    n += 4 * rs_align((size_t)L * B * H * 4);
    n += rs_align((size_t)B * J * 4);
    n += 7 * rs_align((size_t)B * 4);
    n += rs_align(64);
    n += 2 * rs_align((size_t)B * LOOKAHEAD_MAX * nct * 4);
    const size_t vpad = (size_t)(d.n_logits + 15) / 16 * 16;
    n += rs_align((size_t)B * J * 2) + rs_align((size_t)B * 4) + rs_align((size_t)B * vpad * 4);   // screened joint
    return n + 1024;
}

int rs_rnnt_greedy_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int u_max,
                        int32_t* ids, int32_t* frames, int32_t* n_ids, void* workspace, size_t workspace_bytes,
                        hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int L = d.pred_layers, H = d.pred_hidden, J = d.joint_hidden, V = d.n_logits;
    if (B <= 0) return RS_OK;
    if (H % 128 || J % 128) return rs_fail(ctx, RS_EINVAL, "rnnt: pred_hidden/joint_hidden must be multiples of 128");
    if (L < 1 || L > 4) return rs_fail(ctx, RS_EINVAL, "rnnt: 1..4 LSTM layers supported");
    if (workspace_bytes < rs_rnnt_workspace_bytes(ctx, B)) return rs_fail(ctx, RS_EWORKSPACE, "rnnt: workspace too small");
    const int nct = (V + 63) / 64;
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* p = w; w += rs_align(bytes); return p; };
    DecodeState st;
    st.g_off = nullptr; st.a_pre = nullptr;
    st.joint_act = d.joint_act;
    const size_t state_bytes = (size_t)L * B * H * 4;
    st.h = (float*)take(state_bytes); st.c = (float*)take(state_bytes);
    st.h_tmp = (float*)take(state_bytes); st.c_tmp = (float*)take(state_bytes);
    st.g = (float*)take((size_t)B * J * 4);
    st.tcur = (int32_t*)take((size_t)B * 4); st.sym = (int32_t*)take((size_t)B * 4);
    st.token = (int32_t*)take((size_t)B * 4); st.act = (int32_t*)take((size_t)B * 4);
    if (ctx->k2_conv_w) { st.token2 = (int32_t*)take((size_t)B * 4); st.unk = ctx->k2 ? rs_k2_unk_id(ctx) : -1; }
    st.alive = (int32_t*)take((size_t)2 * B * 4);
    st.counters = (int32_t*)take(64);
    st.pmax = (float*)take((size_t)B * LOOKAHEAD_MAX * nct * 4); st.pidx = (int32_t*)take((size_t)B * LOOKAHEAD_MAX * nct * 4);
    const int Vpad = (V + 15) / 16 * 16;
    st.a16 = (uint16_t*)take((size_t)B * J * 2); st.anorm = (float*)take((size_t)B * 4);
    st.zapprox = (float*)take((size_t)B * Vpad * 4);
    // the screened joint keeps a row's logits (<= 48 x 64) and a K slice (<= 8 blocks of 16) in registers, J / 32 <= 20 weight fragments
    const bool screen = ctx->decode_screen && ctx->jout_w16 && ctx->jout_wrm && ctx->jout_bpad && ctx->jout_wmax &&
                        J / SPLITK_TILE / 16 <= 8 && J / 32 <= 20;

    if (int rc = ensure_decode_lds(ctx); rc != RS_OK) return rc;
    const bool narrow = narrow_kernels_usable(ctx);
    auto lstm_and_pred = [&](int rows_bound) { launch_lstm_pred(ctx, st, B, rows_bound, narrow, s); };

    rs_prof_begin(ctx, RS_PROF_DECODE, s, 0.0, 0.0);
    RS_HIP(ctx, hipMemsetAsync(st.h, 0, 2 * rs_align(state_bytes), s));   // h and c are adjacent
    RS_HIP(ctx, hipMemsetAsync(st.g, 0, (size_t)B * J * 4, s));
    hipLaunchKernelGGL(rnnt_init_kernel, dim3(1), dim3(256), 0, s, st, enc_lens, B, d.blank_id, n_ids);
    lstm_and_pred(B);  // SOS: blank token, zero state, all rows
    RS_CHECK_LAUNCH(ctx, "rnnt init");

    const int max_steps = tp_max + (u_max < tp_max * d.max_symbols ? u_max : tp_max * d.max_symbols) + 1;
    // (the whole loop as ONE persistent launch with agent-scope grid barriers between the phases of a step was built and
    // measured in round 2 — bit-identical, slower: five barriers per step cost more than five launch boundaries,
    // profiles/r02p_persistent_decode_ab.txt — and removed this round; k_rnnt_persist.hip is in the history at 6afb282.
    // A chunk of 16 steps captured as a hipGraph and replayed (the kernels use only the parity of the step index, so every
    // replay has the same arguments) was tried for the launch-bound small batches: bit-identical, and slower for one 10 s
    // utterance — decode 3.7 - 3.8 ms against 2.9 - 3.4 ms of plain launches, -1 ... -2 ms only on 20 - 30 s utterances
    // and 40 ms for the first capture of a geometry — profiles/r03x_decode_graph_b1_ab.txt: not kept.)
    const int CHUNK = 16;
    const bool lookahead = getenv("RS_DECODE_NO_LOOKAHEAD") == nullptr;   // A/B and test hook
    const bool verify_wide = getenv("RS_VERIFY_WIDE") && atoi(getenv("RS_VERIFY_WIDE")) != 0;         // A/B and test hook: a workgroup per row for V <= 3072 too
    int32_t host_counters[4] = {0, 0, 0, 0};
    int steps = 0, alive_bound = B;
    bool finished = false;
    while (!finished && steps < max_steps) {
        for (int c = 0; c < CHUNK; ++c, ++steps) {
            // rows still decoding never increase: the count read back after the previous chunk bounds this one
            const int rows = alive_bound < B ? alive_bound : B;
            if (screen) {
                const int rts = (rows + 31) / 32 > 0 ? (rows + 31) / 32 : 1;
                hipLaunchKernelGGL(rnnt_prep_kernel, dim3((rows + 3) / 4 > 0 ? (rows + 3) / 4 : 1), dim3(256), 0, s, st, joint_enc, B,
                                   tp_max, J, steps);
                hipLaunchKernelGGL(rnnt_screen_kernel, dim3((Vpad + 63) / 64, rts), dim3(256), 32 * (J * 2 + 16), s, st, joint_enc, B,
                                   tp_max, J, Vpad, ctx->jout_w16, ctx->jout_bpad, steps);
                const dim3 vg((rows + 3) / 4 > 0 ? (rows + 3) / 4 : 1), vgw(rows > 0 ? rows : 1);
#define RS_VERIFY_ARGS st, joint_enc, enc_lens, B, tp_max, J, V, Vpad, ctx->jout_wrm, ctx->jout_b, ctx->jout_wmax, d.blank_id, d.max_symbols, u_max, steps, ids, frames, n_ids
                if (V > 64 * 48 && V <= 4 * 64 * 48)
                    hipLaunchKernelGGL(rnnt_verify_wide_kernel<48>, vgw, dim3(256), J * 4 + 4 * 48 * 64 * 4 + 64, s, RS_VERIFY_ARGS);
                else if (V > 64 * 48)
                    hipLaunchKernelGGL(rnnt_verify_kernel<0>, vg, dim3(256), 4 * J * 4 + 4 * VER_CAP * 4, s, RS_VERIFY_ARGS);
                else if (verify_wide)
                    hipLaunchKernelGGL(rnnt_verify_wide_kernel<12>, vgw, dim3(256), J * 4 + 4 * 12 * 64 * 4 + 64, s, RS_VERIFY_ARGS);
                else if (V <= 64 * 8)
                    hipLaunchKernelGGL(rnnt_verify_kernel<8>, vg, dim3(256), 4 * J * 4 + 4 * 8 * 64 * 4, s, RS_VERIFY_ARGS);
                else
                    hipLaunchKernelGGL(rnnt_verify_kernel<48>, vg, dim3(256), 4 * J * 4 + 4 * 48 * 64 * 4, s, RS_VERIFY_ARGS);
#undef RS_VERIFY_ARGS
            } else {
                // both kernels walk the compacted alive list: only the row tiles / slots the bound covers are launched
                // (next to the encoder every workgroup, even one that exits at once, has to wait for a free CU)
                // few rows leave the chip idle: each utterance then scores its next `la` frames in the same launch (look-ahead:
                // blank keeps the prediction vector, so those rows are what the next steps would compute) and a step consumes
                // them up to the first label — up to la times fewer steps, same output.  la x rows stays within 256 joint rows.
                int la = 1;
                if (lookahead) while (la < LOOKAHEAD_MAX && 2 * la * (rows > 0 ? rows : 1) <= 256) la *= 2;
                if (la > 1) {
                    const int rts = (rows * la + 31) / 32 > 0 ? (rows * la + 31) / 32 : 1;
                    hipLaunchKernelGGL(rnnt_tile_kernel<5>, dim3(joint_grid_x(nct, rts), rts), dim3(512), TILE_LDS, s, st, joint_enc, B, tp_max, L, H,
                                       J, V, ctx->jout_w, ctx->jout_b, nct, steps, la);
                    hipLaunchKernelGGL(rnnt_finalize_la_kernel, dim3((rows + 3) / 4 > 0 ? (rows + 3) / 4 : 1), dim3(256), 0, s, st, enc_lens, B, nct,
                                       d.blank_id, d.max_symbols, u_max, steps, la, ids, frames, n_ids);
                } else {
                    const int rts = (rows + 31) / 32 > 0 ? (rows + 31) / 32 : 1;
                    hipLaunchKernelGGL(rnnt_tile_kernel<1>, dim3(joint_grid_x(nct, rts), rts), dim3(512), TILE_LDS, s, st, joint_enc, B, tp_max, L, H,
                                       J, V, ctx->jout_w, ctx->jout_b, nct, steps, 1);
                    hipLaunchKernelGGL(rnnt_finalize_kernel, dim3((rows + 3) / 4 > 0 ? (rows + 3) / 4 : 1), dim3(256), 0, s, st, enc_lens, B, nct,
                                       d.blank_id, d.max_symbols, u_max, steps, ids, frames, n_ids);
                }
            }
            lstm_and_pred(rows);
        }
        RS_CHECK_LAUNCH(ctx, "rnnt step");
        RS_HIP(ctx, hipMemcpyAsync(host_counters, st.counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        RS_HIP(ctx, hipStreamSynchronize(s));
        finished = host_counters[2 + (steps & 1)] == 0;
        alive_bound = host_counters[2 + (steps & 1)];
    }
    rs_prof_end(ctx, RS_PROF_DECODE, s);
    if (getenv("RS_DECODE_TRACE")) fprintf(stderr, "[decode trace] B=%d T'=%d: %d steps (screen %d, narrow %d, lookahead %d)\n", B, tp_max, steps, (int)screen, (int)narrow, (int)lookahead);
    if (host_counters[1]) return rs_fail(ctx, RS_EOVERFLOW, "rnnt: an utterance emitted more than u_max=%d tokens", u_max);
    if (!finished) return rs_fail(ctx, RS_ESTATE, "rnnt: decode did not finish in %d steps", max_steps);
    return RS_OK;
}

// k_rnnt_persist.hip — the whole greedy RNN-T decode of a batch as ONE persistent launch
// (same arithmetic, bit for bit, as the per-phase kernels of k_rnnt.hip).
//
// Why: a decode step is a dependency chain of five or six small kernels, ~390 steps per batch.  Inside the
// two-stage pipeline every one of those launches has to WAIT FOR FREE CUs — the encoder GEMM of the next batch owns
// every CU (160 KiB of LDS and all registers per workgroup) and only releases them at tile boundaries — so the chain
// stretched to the length of the encoder step and became the critical path, while its workgroups in turn delayed the
// GEMM's tile rounds (profiles/r02n_kernel_stats.txt: 88 ms of decode kernel time per step, GEMM launches +11 %).
// Here G workgroups of 512 threads (default 64) take their CUs ONCE per batch and keep them until the batch is decoded:
//   * phases of a step — [screen: bf16 logits of every alive row] -> [verify: exact f32 re-evaluation of the columns
//     that can still win, argmax, greedy state machine] -> [LSTM layer 0] -> .. -> [prediction projection] — are
//     separated by a grid barrier over the G workgroups (one monotonic counter; every wave drains its stores, one
//     lane releases at agent scope, arrives, polls relaxed with s_sleep, acquires at agent scope: guide §6 G16 in its
//     counter form).  ~2-3 us per barrier at 32 workgroups instead of a launch boundary PLUS the wait for CUs;
//   * the loop ends on the device when no row is alive: no host round trip every 16 steps;
//   * every spin is bounded: a workgroup that waits too long sets an error word and leaves; the others follow.
// Work inside a phase is cut into jobs that the workgroups take round-robin; the job bodies are the per-phase
// kernels' bodies (narrow tiles: one 16-column tile per job, K slices on the waves), so the accumulation orders and
// therefore every emitted id are unchanged — the same bit-exact tests cover both paths.
#include <stdlib.h>

#include "k_rnnt_common.h"

namespace {

struct PersistArgs {
    DecodeState st;
    const float* f;              // joint encoder projection [B][Tp][J]
    const int32_t* enc_lens;
    int B, Tp, J, H, L, V, Vpad, blank, max_symbols, u_max, max_steps;
    const float* embed;
    const float* lstm_w4[4];     // fragment-major, rows permuted to (unit group, gate, unit)
    const float* lstm_b[4];
    const float* Wp;             // joint.pred, fragment-major
    const float* bp;
    const uint16_t* W16;         // joint output layer, bf16 [Vpad][J]
    const float* bpad;           // bias padded with -3e38
    const float* Wrm;            // joint output layer, f32 [V][J]
    const float* bo;
    const float* wmax;
    int32_t* ids; int32_t* frames; int32_t* n_ids;
    unsigned* sync;              // [0] barrier arrivals, [1] error word (1 = barrier timeout), [2] steps executed
};

constexpr int PWAVES = 8;                        // waves per workgroup (512 threads: 256 VGPRs per lane, no spills)
constexpr unsigned SPIN_LIMIT = 2000000u;        // ~1 s of polling before a workgroup gives up
constexpr int CAND_CAP = 256;                    // candidate columns a wave keeps per row before it falls back to all columns

struct Smem {
    // one buffer, reused phase by phase
    static constexpr int PART_BYTES = 16 * 32 * 17 * 4;   // K-slice partial sums [16][32][17]
};

// ---- grid barrier over gridDim.x workgroups; returns false once the error word is set ----------------------------
__device__ __forceinline__ bool grid_barrier(unsigned* sync, unsigned& epoch, int* sh_flag) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // every wave: its stores are out
    __syncthreads();
    epoch += 1;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // restated where the compiler cannot drop it (G16 pitfall 12)
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * gridDim.x;
        unsigned spins = 0;
        int bad = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT || __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *sh_flag = bad;
    }
    __syncthreads();
    return *sh_flag == 0;
}

// ---- LSTM layer over the `act` rows: job = (unit group of 4 units x 4 gates, 32-row tile) ---------------------------
template <int NBL>   // NBL >= 16-blocks per K slice
__device__ __forceinline__ void lstm_job(const PersistArgs& a, int layer, int ug, int rt, int n_act, char* smem) {
    const DecodeState& st = a.st;
    float (*part)[32][17] = reinterpret_cast<float (*)[32][17]>(smem);
    int* rows_s = reinterpret_cast<int*>(smem + Smem::PART_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, H = a.H;
    if (tid < 32) {
        const int i = rt * 32 + tid;
        rows_s[tid] = st.act[i < n_act ? i : n_act - 1];
    }
    __syncthreads();
    const bool two = n_act - rt * 32 > 16;                // rows 16..31 of the tile are in use (workgroup-uniform)
    const int li = lane & 15, kk = lane >> 4;
    const int lr = lane >> 2, lc = lane & 3;
    const int perm = 4 * (4 * li + kk);
    const int K = 2 * H, kslice = K / SPLITK_LSTM, nkb = K / 16;
    const float* xsrc[2];
    const float* hsrc[2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
        const int row = rows_s[ri * 16 + lr];
        xsrc[ri] = layer == 0 ? a.embed + (size_t)st.token[row] * H : st.h_tmp + ((size_t)(layer - 1) * B + row) * H;
        hsrc[ri] = st.h + ((size_t)layer * B + row) * H;
    }
    const float* wfrag = a.lstm_w4[layer] + ((size_t)ug * nkb) * 256 + lane * 4;
    const int nblk = kslice / 16;
    // 16 K slices on 8 waves: wave w accumulates slices w and w + 8, each from zero in the documented order;
    // the operands of both are requested before the first MFMA
    float4 av[2][2][NBL], wv[2][NBL];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        const int kbeg = (wave + PWAVES * hs) * kslice;
#pragma unroll
        for (int u = 0; u < NBL; ++u)
            if (u < nblk) {
                const int k = kbeg + 16 * u + 4 * lc;
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
                    if (ri == 0 || two)
                        av[hs][ri][u] = (k < H) ? *reinterpret_cast<const float4*>(xsrc[ri] + k) : *reinterpret_cast<const float4*>(hsrc[ri] + (k - H));
                wv[hs][u] = *reinterpret_cast<const float4*>(wfrag + (size_t)((kbeg >> 4) + u) * 256);
            }
    }
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int u = 0; u < NBL; ++u)
            if (u < nblk) {
                const float4 a0 = to_mfma_a_layout(av[hs][0][u], perm);
                float4 a1 = a0;
                if (two) a1 = to_mfma_a_layout(av[hs][1][u], perm);
#define RS_MFMA_E(c)                                                                                       \
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.c, wv[hs][u].c, acc[0], 0, 0, 0);       \
                if (two) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.c, wv[hs][u].c, acc[1], 0, 0, 0);
                RS_MFMA_E(x) RS_MFMA_E(y) RS_MFMA_E(z) RS_MFMA_E(w)
#undef RS_MFMA_E
            }
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave + PWAVES * hs][ri * 16 + 4 * kk + r][li] = acc[ri][r];
    }
    __syncthreads();
    if (tid < 128) {                                      // thread = (row i, unit u4); 512 threads per workgroup
        const int i = tid >> 2, u4 = tid & 3;
        if (rt * 32 + i < n_act) {
            const int brow = rows_s[i];
            const int unit = ug * 4 + u4;
            float z[4];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) {
                float sm = part[0][i][gt * 4 + u4];
#pragma unroll
                for (int sl = 1; sl < SPLITK_LSTM; ++sl) sm = sm + part[sl][i][gt * 4 + u4];
                z[gt] = sm + a.lstm_b[layer][gt * H + unit];
            }
            const float ig = rs_sigmoidf(z[0]), fg = rs_sigmoidf(z[1]), gg = rs_tanhf(z[2]), og = rs_sigmoidf(z[3]);
            const size_t o = ((size_t)layer * B + brow) * H + unit;
            const float cn = fmaf(fg, st.c[o], ig * gg);
            st.c_tmp[o] = cn;
            st.h_tmp[o] = og * rs_tanhf(cn);
        }
    }
    __syncthreads();                                      // the partial sums are free for the next job
}

// ---- prediction projection + state commit: job = (16-column tile, 32-row tile), the 8 K slices on the 8 waves -----------
template <int NBL>
__device__ __forceinline__ void pred_jobs(const PersistArgs& a, int job0, int njobs, int nct, int n_act, char* smem) {
    const DecodeState& st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w8 = wave;
    float (*part)[32][17] = reinterpret_cast<float (*)[32][17]>(smem);
    int* rows_s = reinterpret_cast<int*>(smem + Smem::PART_BYTES);
    const int job = job0;
    const bool live = job < njobs;
    const int rt = live ? job / nct : 0, ct = live ? job % nct : 0;
    const int B = a.B, H = a.H, L = a.L, N = a.J;
    const int t8 = tid & 511;
    if (t8 < 32) {
        const int i = rt * 32 + t8;
        rows_s[t8] = st.act[i < n_act ? i : n_act - 1];
    }
    __syncthreads();
    if (live) {
        const bool two = n_act - rt * 32 > 16;
        const int li = lane & 15, kk = lane >> 4;
        const int lr = lane >> 2, lc = lane & 3;
        const int perm = 4 * (4 * li + kk);
        const int K = H, nkb = K / 16;
        const float* asrc[2];
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) asrc[ri] = st.h_tmp + ((size_t)(L - 1) * B + rows_s[ri * 16 + lr]) * H;
        const float* wfrag = a.Wp + ((size_t)ct * nkb) * 256 + lane * 4;
        f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        const int kslice = K / SPLITK_TILE, kbeg = w8 * kslice, nblk = kslice / 16;
        float4 av[2][NBL], wv[NBL];
#pragma unroll
        for (int u = 0; u < NBL; ++u)
            if (u < nblk) {
                const int k = kbeg + 16 * u + 4 * lc;
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
                    if (ri == 0 || two) av[ri][u] = *reinterpret_cast<const float4*>(asrc[ri] + k);
                wv[u] = *reinterpret_cast<const float4*>(wfrag + (size_t)((kbeg >> 4) + u) * 256);
            }
#pragma unroll
        for (int u = 0; u < NBL; ++u)
            if (u < nblk) {
                const float4 a0 = to_mfma_a_layout(av[0][u], perm);
                float4 a1 = a0;
                if (two) a1 = to_mfma_a_layout(av[1][u], perm);
#define RS_MFMA_E(c)                                                                                   \
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.c, wv[u].c, acc[0], 0, 0, 0);       \
                if (two) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.c, wv[u].c, acc[1], 0, 0, 0);
                RS_MFMA_E(x) RS_MFMA_E(y) RS_MFMA_E(z) RS_MFMA_E(w)
#undef RS_MFMA_E
            }
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[w8][ri * 16 + 4 * kk + r][li] = acc[ri][r];
    }
    __syncthreads();
    if (live) {
        const int i = t8 >> 4, c = t8 & 15;               // thread = (row i, column c)
        const bool row_ok = rt * 32 + i < n_act;
        const int brow = rows_s[i];
        const int v = ct * 16 + c;
        if (row_ok && v < N) {
            float sm = part[0][i][c];
#pragma unroll
            for (int sl = 1; sl < SPLITK_TILE; ++sl) sm = sm + part[sl][i][c];
            st.g[(size_t)brow * N + v] = sm + a.bp[v];
        }
        if (row_ok) {                                     // commit this row's new LSTM state: the column tiles share the H units
            for (int l = 0; l < L; ++l)
                for (int u = ct * 16 + c; u < H; u += nct * 16) {
                    const size_t o = ((size_t)l * B + brow) * H + u;
                    st.h[o] = st.h_tmp[o];
                    st.c[o] = st.c_tmp[o];
                }
        }
    }
    __syncthreads();
}

// ---- screening GEMM: job = (32-row tile of alive slots, group of column tiles) --------------------------------------------
__device__ __forceinline__ void screen_job(const PersistArgs& a, int step, int rt, int ct0, int ct1, bool write_norm, int n_alive,
                                           char* smem) {
    const DecodeState& st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int J = a.J, B = a.B, Tp = a.Tp, Vpad = a.Vpad;
    const int ldrow = J * 2 + 16;                         // +16: the 16-row fragment reads stay conflict free
    // operand tile a = relu(f[b][t_b] + g[b]) -> bf16: thread = (row tid >> 4, 16th tid & 15 of the columns)
    {
        const int r = tid >> 4, part = tid & 15;
        const int i = rt * 32 + r;
        const int b = st.alive[(size_t)(step & 1) * B + (i < n_alive ? i : n_alive - 1)];
        int t = st.tcur[b];
        t = t < Tp ? t : Tp - 1;
        const float* fr = a.f + ((size_t)b * Tp + t) * J;
        const float* gr = st.g + (size_t)b * J;
        float ss = 0.0f;
        for (int k = 4 * part; k < J; k += 64) {
            const float4 fv = *reinterpret_cast<const float4*>(fr + k), gv = *reinterpret_cast<const float4*>(gr + k);
            const float a0 = fmaxf(fv.x + gv.x, 0.0f), a1 = fmaxf(fv.y + gv.y, 0.0f), a2 = fmaxf(fv.z + gv.z, 0.0f),
                        a3 = fmaxf(fv.w + gv.w, 0.0f);
            ss = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, fmaf(a3, a3, ss))));
            *reinterpret_cast<u16x4_t*>(smem + r * ldrow + k * 2) = pack_bf16x4(a0, a1, a2, a3);
        }
        if (write_norm) {                                 // workgroup-uniform
            ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
            ss += __shfl_xor(ss, 8, 64);
            if (part == 0 && i < n_alive) st.anorm[i] = sqrtf(ss) * 1.0001f;   // any summation order: the bound has slack
        }
    }
    __syncthreads();
    const int li = lane & 15, kc = lane >> 4;
    const int nks = J / 32;
    for (int ct = ct0 + wave; ct < ct1; ct += PWAVES) {   // one 16-column tile per wave and pass
        const int col0 = ct * 16;
        const uint16_t* wrow = a.W16 + (size_t)(col0 + li) * J + kc * 8;
        bf16x8_t wf[20];
#pragma unroll
        for (int q = 0; q < 20; ++q)
            if (q < nks) wf[q] = *reinterpret_cast<const bf16x8_t*>(wrow + q * 32);
        const float bv = a.bpad[col0 + li];
        f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < 20; ++q)
            if (q < nks) {
#pragma unroll
                for (int ri = 0; ri < 2; ++ri) {
                    const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(smem + (ri * 16 + li) * ldrow + (q * 32 + kc * 8) * 2);
                    acc[ri] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[q], acc[ri], 0, 0, 0);
                }
            }
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int slot = rt * 32 + ri * 16 + 4 * kc + r;
                if (slot < n_alive) st.zapprox[(size_t)slot * Vpad + col0 + li] = acc[ri][r] + bv;
            }
    }
    __syncthreads();
}

// ---- verify + greedy state machine: one wave per alive slot --------------------------------------------------------------
// (not inlined: its 48 logit registers + weight registers get their own allocation instead of spilling the caller)
template <int NCH>
__device__ __attribute__((noinline)) void verify_row(const PersistArgs& a, int step, int slot, float* a_s, int* cand_s) {
    const DecodeState& st = a.st;
    const int lane = threadIdx.x & 63;
    const int B = a.B, Tp = a.Tp, J = a.J, V = a.V, Vpad = a.Vpad;
    const int b = st.alive[(size_t)(step & 1) * B + slot];
    const float* z = st.zapprox + (size_t)slot * Vpad;
    float zr[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) zr[c] = (c * 64 + lane < V) ? z[c * 64 + lane] : -INFINITY;
    int t = st.tcur[b];
    const int tc = t < Tp ? t : Tp - 1;
    const float* fr = a.f + ((size_t)b * Tp + tc) * J;
    const float* gr = st.g + (size_t)b * J;
    for (int k = 4 * lane; k < J; k += 256) {
        const float4 fv = *reinterpret_cast<const float4*>(fr + k), gv = *reinterpret_cast<const float4*>(gr + k);
        *reinterpret_cast<float4*>(a_s + k) = make_float4(fmaxf(fv.x + gv.x, 0.0f), fmaxf(fv.y + gv.y, 0.0f),
                                                          fmaxf(fv.z + gv.z, 0.0f), fmaxf(fv.w + gv.w, 0.0f));
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; ++c) m = fmaxf(m, zr[c]);
    m = wave_max(m);
    // 2 eps = 2 * 2^-7 * 1.25 * ||a|| * max_v ||w_v||   (k_rnnt.hip, "screened joint")
    const float thr = m - 0.01953125f * a.wmax[0] * st.anorm[slot];
    int n_cand = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool is = zr[c] >= thr;
        const unsigned long long mask = __ballot(is);
        const int pos = n_cand + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (is && pos < CAND_CAP) cand_s[pos] = c * 64 + lane;
        n_cand += __builtin_popcountll(mask);
    }
    const bool all_cols = n_cand > CAND_CAP;              // (never seen in practice) too many to list: evaluate every column
    if (all_cols) n_cand = V;
    const int cgrp = lane >> 3, sl = lane & 7;            // 8 candidates per pass x 8 K slices
    const int kslice = J / SPLITK_TILE, nblk = kslice / 16;
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    for (int c0 = 0; c0 < n_cand; c0 += 8) {
        const bool valid = c0 + cgrp < n_cand;
        const int cand = valid ? (all_cols ? c0 + cgrp : cand_s[c0 + cgrp]) : 0;
        const float* w = a.Wrm + (size_t)cand * J + sl * kslice;
        const float* as = a_s + sl * kslice;
        float acc = 0.0f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                     // the slice in two halves of <= 4 blocks: 64 weight registers at a time
            float4 wv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * h + u < nblk) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) wv[u][kk] = *reinterpret_cast<const float4*>(w + 16 * (4 * h + u) + 4 * kk);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * h + u < nblk) {
                    float4 av[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) av[kk] = *reinterpret_cast<const float4*>(as + 16 * (4 * h + u) + 4 * kk);
#define RS_CHAIN_E(cc) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) acc = fmaf(av[kk].cc, wv[u][kk].cc, acc);
                    RS_CHAIN_E(x) RS_CHAIN_E(y) RS_CHAIN_E(z) RS_CHAIN_E(w)
#undef RS_CHAIN_E
                }
        }
        float sum = __shfl(acc, lane & ~7, 64);
#pragma unroll
        for (int q = 1; q < SPLITK_TILE; ++q) sum = sum + __shfl(acc, (lane & ~7) + q, 64);
        float val = valid ? sum + a.bo[cand] : -INFINITY;
        int idx = valid ? cand : 0x7fffffff;
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const float ov = __shfl_xor(val, off, 64);
            const int oi = __shfl_xor(idx, off, 64);
            if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        if (val > best || (val == best && idx < best_idx)) { best = val; best_idx = idx; }
    }
    if (lane != 0) return;
    const int idx = best_idx;
    int sy = st.sym[b];
    bool emitted = false;
    if (idx == a.blank || idx == 0x7fffffff) {
        t += 1; sy = 0;
    } else {
        const int n = a.n_ids[b];
        if (n < a.u_max) { a.ids[(size_t)b * a.u_max + n] = idx; a.frames[(size_t)b * a.u_max + n] = t; a.n_ids[b] = n + 1; }
        else st.counters[1] = 1;
        st.token[b] = idx;
        emitted = true;
        sy += 1;
        if (sy >= a.max_symbols) { t += 1; sy = 0; }
    }
    st.tcur[b] = t; st.sym[b] = sy;
    if (t < a.enc_lens[b]) {
        const int pos = atomicAdd(&st.counters[2 + ((step + 1) & 1)], 1);
        st.alive[(size_t)((step + 1) & 1) * B + pos] = b;
        if (emitted) { const int pa = atomicAdd(&st.counters[0], 1); st.act[pa] = b; }
    }
}

template <int NCH, int NBL>
__global__ __launch_bounds__(512) void rnnt_persist_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sh_flag;
    const DecodeState& st = a.st;
    const int G = gridDim.x, wg = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6;
    unsigned epoch = 0;
    const int nug = a.H / 4, npc = (a.J + 15) / 16, ntile = a.Vpad / 16;

    auto lstm_and_pred = [&](int n_act) -> bool {
        const int rts = (n_act + 31) / 32;
        for (int l = 0; l < a.L; ++l) {
            for (int job = wg; job < rts * nug; job += G) lstm_job<NBL>(a, l, job % nug, job / nug, n_act, smem);
            if (!grid_barrier(a.sync, epoch, &sh_flag)) return false;
        }
        const int njobs = rts * npc;
        for (int job0 = wg; job0 < njobs; job0 += G) pred_jobs<NBL>(a, job0, njobs, npc, n_act, smem);
        return grid_barrier(a.sync, epoch, &sh_flag);
    };

    // SOS: blank token, zero state, every row (rnnt_init_kernel left act = all rows)
    int steps = 0;
    if (!lstm_and_pred(st.counters[0])) return;
    for (; steps < a.max_steps; ++steps) {
        const int n_alive = st.counters[2 + (steps & 1)];
        if (n_alive == 0) break;
        // ---- screen (also: reset the lists the verify phase builds)
        if (wg == 0 && tid == 0) { st.counters[0] = 0; st.counters[2 + ((steps + 1) & 1)] = 0; }
        {
            const int rts = (n_alive + 31) / 32;
            int cq = G / rts;                              // column groups per row tile: every workgroup gets a job
            cq = cq < 1 ? 1 : (cq > ntile ? ntile : cq);
            const int per = (ntile + cq - 1) / cq;
            for (int job = wg; job < rts * cq; job += G) {
                const int rt = job / cq, q = job % cq;
                const int ct0 = q * per, ct1 = (q + 1) * per < ntile ? (q + 1) * per : ntile;
                screen_job(a, steps, rt, ct0, ct1, q == 0, n_alive, smem);
            }
        }
        if (!grid_barrier(a.sync, epoch, &sh_flag)) return;
        // ---- verify + state machine
        {
            float* a_s = reinterpret_cast<float*>(smem) + wave * a.J;
            int* cand_s = reinterpret_cast<int*>(smem + PWAVES * a.J * 4) + wave * CAND_CAP;
            for (int slot = wg * PWAVES + wave; slot < n_alive; slot += G * PWAVES) verify_row<NCH>(a, steps, slot, a_s, cand_s);
        }
        if (!grid_barrier(a.sync, epoch, &sh_flag)) return;
        const int n_act = st.counters[0];
        if (n_act > 0 && !lstm_and_pred(n_act)) return;
    }
    if (wg == 0 && tid == 0) a.sync[2] = (unsigned)steps;
}

}  // namespace

size_t rs_rnnt_persist_lds_bytes(int J) {
    size_t lds = Smem::PART_BYTES + 2 * 32 * 4;                     // LSTM / projection partial sums + row lists
    const size_t scr = (size_t)32 * (J * 2 + 16);                   // screening operand tile
    const size_t ver = (size_t)PWAVES * J * 4 + (size_t)PWAVES * CAND_CAP * 4;
    if (scr > lds) lds = scr;
    if (ver > lds) lds = ver;
    return (lds + 255) / 256 * 256;
}

// Launch the persistent decode of one batch (state already initialised by rnnt_init_kernel + memsets).
int rs_rnnt_persist_launch(rs_ctx* ctx, const void* st_ptr, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max,
                           int u_max, int max_steps, int32_t* ids, int32_t* frames, int32_t* n_ids, unsigned* sync, int n_wgs,
                           hipStream_t s) {
    const rs_dims& d = ctx->d;
    PersistArgs a{};
    a.st = *reinterpret_cast<const DecodeState*>(st_ptr);
    a.f = joint_enc; a.enc_lens = enc_lens;
    a.B = B; a.Tp = tp_max; a.J = d.joint_hidden; a.H = d.pred_hidden; a.L = d.pred_layers; a.V = d.n_logits;
    a.Vpad = (d.n_logits + 15) / 16 * 16; a.blank = d.blank_id; a.max_symbols = d.max_symbols; a.u_max = u_max; a.max_steps = max_steps;
    a.embed = ctx->embed;
    for (int l = 0; l < d.pred_layers; ++l) { a.lstm_w4[l] = ctx->lstm_w4[l]; a.lstm_b[l] = ctx->lstm_b[l]; }
    a.Wp = ctx->jpred_w; a.bp = ctx->jpred_b;
    a.W16 = ctx->jout_w16; a.bpad = ctx->jout_bpad; a.Wrm = ctx->jout_wrm; a.bo = ctx->jout_b; a.wmax = ctx->jout_wmax;
    a.ids = ids; a.frames = frames; a.n_ids = n_ids; a.sync = sync;
    const int lds = (int)rs_rnnt_persist_lds_bytes(a.J);
    const int nb_l = 2 * a.H / SPLITK_LSTM / 16, nb_t = a.H / SPLITK_TILE / 16;
    const bool small = nb_l <= 5 && nb_t <= 5;             // blocks of 16 per K slice (5 at the 619M geometry)
#define RS_PERSIST(NCH, NBL)                                                                                          \
    do {                                                                                                              \
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)rnnt_persist_kernel<NCH, NBL>, lds); rc != RS_OK) return rc; \
        hipLaunchKernelGGL((rnnt_persist_kernel<NCH, NBL>), dim3(n_wgs), dim3(512), lds, s, a);                      \
    } while (0)
    if (a.V <= 64 * 8) { if (small) RS_PERSIST(8, 5); else RS_PERSIST(8, 8); }
    else { if (small) RS_PERSIST(48, 5); else RS_PERSIST(48, 8); }
#undef RS_PERSIST
    return RS_OK;
}

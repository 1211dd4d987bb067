// k_attention.hip — relative-position multi-head self-attention core for gfx950
// ([UPSTREAM] RelPositionMultiHeadAttention.forward).
//
//   s[i][j] = ( (q_i + u) . k_j  +  (q_i + v) . p[j - i + T - 1] ) / sqrt(128)
//   ctx_i   = sum_j softmax_j(s[i][.]) v_j          (masked keys weigh 0; padded queries -> 0)
//
// The rel-shift of Transformer-XL is folded into indexing (closed form): no
// [T][2T-1] matrix is materialised.  Flash-style: one wave owns 32 query rows, loops over 32-key
// blocks with an online softmax; every product runs on v_mfma_f32_32x32x16_bf16.
//
// Orientation trick: the wave computes S^T = K . (Q+u)^T, so in the MFMA C layout each lane holds
// one QUERY column (lane&31) and 16 of the 32 keys of the block (the other 16 sit in lane^32):
// softmax reductions are 16 in-register ops + one cross-half shuffle, and the probabilities are
// already laid out as the B operand of  O^T += V^T . P^T  (no LDS round trip for P).  The keys a
// lane holds are not contiguous ({0-3,8-11,..} / {4-7,12-15,..}); V^T is staged in LDS with its
// key axis permuted the same way, so the contraction pairs up.
//
// The position term needs, for the (query block, key block) pair, the 63 relative positions
// n0 .. n0+62 with n0 = j0 - i0 - 31 + T - 1:  BD^T[64][32] = P[n0..n0+63] . (Q+v)^T.  Moving to the
// next key block shifts n0 by 32, so only the upper 32 rows are new: 8 MFMAs per block, the lower
// half is the previous block's upper half.  The per-row skew  bd[key][query] = BD^T[key - query + 31][query]
// goes through a per-wave LDS scratch (conflict-free both ways: the query is the fastest index);
// the same scratch first stages the 32 position rows (coalesced 256-byte reads from L2).
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "rs_common.h"

namespace {

constexpr int VROW = 80;            // bytes per V^T row in LDS (64 + 16 pad)
constexpr int SKEW_LD = 68;            // floats per query row of the skew tile (64 used)
// Geometry per head dimension HD (128: FastConformer-XL, the shape the kernel was tuned on; 64: the ESPnet Conformer's
// 512 / 8).  A K / P row is HD bf16 = NCH 16-byte chunks.
template <int HD>
struct AttnGeom {
    static_assert(HD == 128 || HD == 64, "head_dim 128 or 64");
    static constexpr int NCH = HD / 8;             // 16-byte chunks per row
    static constexpr int KS = HD / 16;             // 16-deep k-steps of the S^T / BD^T products
    static constexpr int DB = HD / 32;             // 32-row blocks of O^T
    static constexpr int KROW = 2 * HD + 16;       // bytes per K / P row in LDS (+16 pad: conflict-free b128 reads)
    static constexpr int K_BYTES = 32 * KROW;
    static constexpr int VT_BYTES = HD * VROW;
    // per-wave scratch: 32 staged P rows, later the [32 queries][68] f32 skew tile
    static constexpr int SCR_BYTES = 32 * KROW > 32 * SKEW_LD * 4 ? 32 * KROW : 32 * SKEW_LD * 4;
    static constexpr int RPP = 64 / NCH;           // rows one 64-lane pass of 16-byte pieces covers
    static constexpr int VH = NCH / 8;             // wave-wide V staging items per key block
    // key blocks staged per workgroup barrier: HD 128: 5 (160 keys: all of T' = 138); HD 64 (half the bytes per key): 6, so that
    // the ESPnet model's T' = 358 (12 key blocks) is two chunks, both staged by the register-transposing fast path
    static constexpr int KB_CHUNK = HD == 128 ? 5 : 6;
};
constexpr float NEG = -1.0e30f;

struct AttnParams {
    const uint16_t* qkv; const uint16_t* pos; const float* bias_u; const float* bias_v;
    const int32_t* lens; uint16_t* out;
    int T, d_model, att_left, att_right, n_global;
    float scale;
    long long* trace;   // debug: per-workgroup phase timestamps of wave 1 (scripts/attn_trace.py); nullptr in production
    // persistent launch (n_items > 0): a 1-D grid of resident workgroups, each walking items blockIdx.x, + gridDim.x, ...;
    // item = (query group, head, utterance), query group fastest.  0: the classic (query groups, heads, batch) grid
    int n_items, n_groups, n_heads;
};

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// position of key `key` (0..31) inside a V^T row: the order in which the S^T accumulator registers
// of a lane enumerate the keys, so P^T can feed the PV MFMA straight from registers
__device__ __forceinline__ int vt_pos(int key) {
    const int kh = (key >> 2) & 1, kr = (key & 3) + 4 * (key >> 3);
    return (kr >> 3) * 16 + kh * 8 + (kr & 7);
}

// One workgroup = (batch b, head h, up to 6 query blocks of 32); one wave = one query block.
// (Tried in round 3: delaying every second workgroup of the first dispatch round by 2k .. 14k cycles so that the Q / K / V
// prologue bursts of the two halves of the chip stop coinciding — no effect, 154 - 156 us either way:
// profiles/r03i_attn_skew_ab.txt.  Also tried: the loads issued in consumption order (Q, bias, first position rows, K by
// global_load_lds into a swizzled 256-byte-pitch image, V) with the Q fragments and the first position block built BEFORE
// the K / V barrier instead of after it — bit-identical, 152.4 / 153.1 vs 151.1 / 152.4 us on one box
// (profiles/r03w_attn_prologue_ab.txt): the microsecond moved under the loads comes back as a longer load phase.)
// WINDOW: limited-context / global-token masks compiled in (full attention otherwise: no per-score branches)
// KBC: key blocks staged per barrier (0: the geometry's default).  The small-chunk instantiations of head_dim 64 (4 blocks with
// <= 4 waves: 74 KB) fit TWO workgroups per CU: the staging of one overlaps the products of the other (launcher).  The same for
// head_dim 128 (2 blocks, <= 3 waves) was measured and is slower — its later chunks restage on the one-item path, there are no
// registers for the fast one: 72.7 -> 73.4 ms per batch, profiles/r05u_attn_2wg_ab.txt — and is not built.
template <int HD, bool TRACE, bool WINDOW, int KBC = 0, bool PERSIST = false>
__global__ __launch_bounds__(384) void relpos_attention_kernel(AttnParams p) {
    using G = AttnGeom<HD>;
    constexpr int NCH = G::NCH, KS = G::KS, DB = G::DB, KROW = G::KROW, K_BYTES = G::K_BYTES, VT_BYTES = G::VT_BYTES;
    constexpr int SCR_BYTES = G::SCR_BYTES, RPP = G::RPP, VH = G::VH, KB_CHUNK = KBC > 0 ? KBC : G::KB_CHUNK;
    long long ts[40];
    int nts = 0;
    // TRACE: drain the memory counters, make the newest MFMA result architecturally visible, then stamp
    auto stamp = [&](float dep) {
        if constexpr (TRACE) {
            float tmp;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_mov_b32 %0, %1" : "=v"(tmp) : "v"(dep) : "memory");
            if (nts < 40) ts[nts++] = __builtin_readcyclecounter();
            asm volatile("" :: "v"(tmp));
        }
    };
    stamp(0.0f);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                              // [KB_CHUNK][32 keys][272 B]
    char* Vts = smem + KB_CHUNK * K_BYTES;        // [KB_CHUNK][128 d][80 B]
    const int nw = blockDim.x >> 6;
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    char* scr = smem + KB_CHUNK * (K_BYTES + VT_BYTES) + wave * SCR_BYTES;
    float* bias_s = reinterpret_cast<float*>(smem + KB_CHUNK * (K_BYTES + VT_BYTES) + nw * SCR_BYTES);   // [u(128) | v(128)]
    float* scr_f = reinterpret_cast<float*>(scr);

    const int T = p.T, d = p.d_model, ld = 3 * d;
    for (int item = blockIdx.x;; item += gridDim.x) {      // (one pass on the classic grid)
    // the lane id is laundered per item: hoisted out of this loop, the lane-dependent staging addresses of an item stay live
    // across the whole key loop and spill (20 VGPRs in the first persistent build)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    int b = blockIdx.z, h = blockIdx.y, qg = blockIdx.x;
    if constexpr (PERSIST) { qg = item % p.n_groups; h = (item / p.n_groups) % p.n_heads; b = item / (p.n_groups * p.n_heads); }
    const int len = p.lens[b];
    const int i0 = (qg * nw + wave) * 32;
    const int il = lane & 31, hh = lane >> 5;
    const int qi = i0 + il;
    const bool q_valid = qi < len && qi < T;
    const uint16_t* base = p.qkv + (size_t)b * T * ld;

    const int n_kblocks = ((len < T ? len : T) + 31) >> 5;  // key blocks at/after len are fully masked
    // ---- K / V staging of `nb` key blocks starting at block jc, first chunk only (nothing else is live
    //      in registers yet, so every global load of the chunk is in flight before the first LDS store).
    //      K: thread -> (slot, key, 16-byte chunk), rows copied as they are.
    //      V: thread -> (slot, 4 consecutive keys, 8 d columns): the 4 x 8 block is transposed in
    //      registers (v_perm_b32) and stored as 8-byte runs of V^T.  The element-wise scatter this
    //      replaces (64 two-byte stores per thread, 16 lanes per bank) was 7.6 of a workgroup's 24 us
    //      (profiles/r01o_attention_phase_timeline.txt).
    auto stage_kv = [&](int jc, int nb) {
        // nb * 512 K items <= 8 * blockDim for every launch shape.  Named scalars, unconditional (clamped)
        // loads: an indexed array or a predicated definition is demoted to scratch memory by the compiler.
        const int items = nb * 32 * NCH;
        auto k_src = [&](int it) -> const uint4* {
            int idx = tid + it * (int)blockDim.x;
            idx = idx < items ? idx : items - 1;
            const int slot = idx / (32 * NCH), key = (idx / NCH) & 31, ch = idx % NCH;
            int krow = (jc + slot) * 32 + key;
            krow = krow < T ? krow : T - 1;
            return reinterpret_cast<const uint4*>(base + (size_t)krow * ld + d + h * HD + ch * 8);
        };
        const uint4 kr0 = *k_src(0), kr1 = *k_src(1), kr2 = *k_src(2), kr3 = *k_src(3);
        const uint4 kr4 = *k_src(4), kr5 = *k_src(5), kr6 = *k_src(6), kr7 = *k_src(7);
        // V work items are wave-wide: (slot, half of the 16 chunks); lane = (key group g, chunk c_lo);
        // a 16-lane store group is 8 key groups x 2 chunks (2-way bank conflict at worst)
        constexpr int MAXV = 2;               // VH * nb wave items over nw waves
        const int g = lane & 7, c_lo = (lane >> 3) & 7;
        uint4 vreg[MAXV][4];
#pragma unroll
        for (int it = 0; it < MAXV; ++it) {
            int wi = wave + it * nw;
            wi = wi < VH * nb ? wi : VH * nb - 1;
            const int slot = wi / VH, ch = (wi % VH) * 8 + c_lo;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                int krow = (jc + slot) * 32 + 4 * g + a;
                krow = krow < T ? krow : T - 1;
                vreg[it][a] = *reinterpret_cast<const uint4*>(base + (size_t)krow * ld + 2 * d + h * HD + ch * 8);
            }
        }
        auto k_put = [&](int it, const uint4& v) {
            const int idx = tid + it * (int)blockDim.x;
            if (idx < items) {
                const int slot = idx / (32 * NCH), key = (idx / NCH) & 31, ch = idx % NCH;
                *reinterpret_cast<uint4*>(Ks + slot * K_BYTES + key * KROW + ch * 16) = v;
            }
        };
        k_put(0, kr0); k_put(1, kr1); k_put(2, kr2); k_put(3, kr3);
        k_put(4, kr4); k_put(5, kr5); k_put(6, kr6); k_put(7, kr7);
#pragma unroll
        for (int it = 0; it < MAXV; ++it) {
            const int wi = wave + it * nw;
            if (wi < VH * nb) {
                const int slot = wi / VH, ch = (wi % VH) * 8 + c_lo;
                const int pos0 = vt_pos(4 * g);           // keys 4g .. 4g+3 sit at positions pos0 .. pos0+3
                char* dst = Vts + slot * VT_BYTES + (ch * 8) * VROW + ((((pos0 >> 3) ^ (ch & 3)) << 4) + (pos0 & 7) * 2);
                const unsigned k0[4] = {vreg[it][0].x, vreg[it][0].y, vreg[it][0].z, vreg[it][0].w};
                const unsigned k1[4] = {vreg[it][1].x, vreg[it][1].y, vreg[it][1].z, vreg[it][1].w};
                const unsigned k2[4] = {vreg[it][2].x, vreg[it][2].y, vreg[it][2].z, vreg[it][2].w};
                const unsigned k3[4] = {vreg[it][3].x, vreg[it][3].y, vreg[it][3].z, vreg[it][3].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // v_perm_b32 D, S0, S1: bytes 0-3 = S1, 4-7 = S0; pick the low (even d) or high halves
                    const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    const unsigned lo = __builtin_amdgcn_perm(k1[e >> 1], k0[e >> 1], sel);
                    const unsigned hi = __builtin_amdgcn_perm(k3[e >> 1], k2[e >> 1], sel);
                    *reinterpret_cast<uint2*>(dst + e * VROW) = make_uint2(lo, hi);
                }
            }
        }
    };

    // ---- prologue.  Everything the workgroup needs from HBM is requested up front: the raw Q rows
    // first, then the first K/V chunk (staged while nothing else is live in registers).  The two bias
    // vectors of this head go through LDS: read per lane from global memory they were 32 x 16 bytes
    // per lane, identical across the wave, and held the prologue for ~3 us.
    const int qrow = qi < T ? qi : T - 1;
    const uint16_t* qp = base + (size_t)qrow * ld + h * HD;
    u16x8_t raw[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) raw[ks] = *reinterpret_cast<const u16x8_t*>(qp + 8 * (2 * ks + hh));
    if (wave == 0 && 4 * lane < 2 * HD)                  // bias_s = [u(HD) | v(HD)]
        *reinterpret_cast<float4*>(bias_s + 4 * lane) =
            *reinterpret_cast<const float4*>((4 * lane < HD ? p.bias_u : p.bias_v - HD) + h * HD + 4 * lane);
    stage_kv(0, n_kblocks < KB_CHUNK ? n_kblocks : KB_CHUNK);
    stamp(0.0f);                                   // [1] own K/V stores issued and landed
    __syncthreads();
    stamp(0.0f);                                   // [2] workgroup barrier passed

    // ---- Q fragments (+u, +v), rounded to bf16: B operand, lane = (query il, k-chunk hh)
    bf16x8_t qu[KS], qv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = 8 * (2 * ks + hh);
        float bu[8], bvv[8];
        *reinterpret_cast<float4*>(&bu[0]) = *reinterpret_cast<const float4*>(bias_s + c);
        *reinterpret_cast<float4*>(&bu[4]) = *reinterpret_cast<const float4*>(bias_s + c + 4);
        *reinterpret_cast<float4*>(&bvv[0]) = *reinterpret_cast<const float4*>(bias_s + HD + c);
        *reinterpret_cast<float4*>(&bvv[4]) = *reinterpret_cast<const float4*>(bias_s + HD + c + 4);
        u16x8_t a, bq;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = bf16_to_f32(raw[ks][e]);
            a[e] = f32_to_bf16(f + bu[e]);
            bq[e] = f32_to_bf16(f + bvv[e]);
        }
        qu[ks] = __builtin_bit_cast(bf16x8_t, a);
        qv[ks] = __builtin_bit_cast(bf16x8_t, bq);
    }

    f32x16_t o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.0f;
    float m_run = NEG, l_run = 0.0f;
    stamp(0.0f);                                   // [3] Q fragments built

    const uint16_t* pos_h = p.pos + h * HD;
    const int n_pos = 2 * T - 1;

    // ---- BD^T block: 32 relative-position rows starting at nrow0, through the wave's scratch
    //      (coalesced 256-byte row reads from L2 -> LDS rows of 272 B -> conflict-free fragment reads)
    auto bd_block = [&](int nrow0) -> f32x16_t {
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // two passes of 16 rows (HD 128: 4 rows-of-4, 16 staging VGPRs, not 32)
            // (named scalars: a uint4 array here is demoted to scratch memory, with dead stores in the loop)
            auto p_src = [&](int q) -> const uint4* {
                int n = nrow0 + 16 * half + RPP * q + lane / NCH;
                n = n < 0 ? 0 : (n >= n_pos ? n_pos - 1 : n);
                return reinterpret_cast<const uint4*>(pos_h + (size_t)n * d + (lane % NCH) * 8);
            };
            char* dst = scr + (16 * half + lane / NCH) * KROW + (lane % NCH) * 16;
            if constexpr (RPP == 4) {
                const uint4 p0 = *p_src(0), p1 = *p_src(1), p2 = *p_src(2), p3 = *p_src(3);
                *reinterpret_cast<uint4*>(dst) = p0;
                *reinterpret_cast<uint4*>(dst + 4 * KROW) = p1;
                *reinterpret_cast<uint4*>(dst + 8 * KROW) = p2;
                *reinterpret_cast<uint4*>(dst + 12 * KROW) = p3;
            } else {
                const uint4 p0 = *p_src(0), p1 = *p_src(1);
                *reinterpret_cast<uint4*>(dst) = p0;
                *reinterpret_cast<uint4*>(dst + 8 * KROW) = p1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(scr + il * KROW + (2 * ks + hh) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qv[ks], acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // fragment reads done before the scratch is reused
        return acc;
    };

    // ---- limited context (WINDOW): only the key blocks a query block can see are visited, so the cost is
    // O(T * (left + right)) instead of O(T^2) — what makes hour-long single utterances (the reference hands a whole
    // file to the model: pkg/nemo-asr/src/transcribe.py:44-53) affordable.  A workgroup skips the staging of key
    // chunks none of its query blocks needs; a wave skips the blocks outside its own window.  Blocks holding
    // global keys (j < n_global) are always visited, and query blocks holding global queries visit everything.
    int wg_lo = 0, wg_hi = n_kblocks - 1, w_lo = 0, w_hi = n_kblocks - 1, gb_hi = -1;
    if constexpr (WINDOW) {
        const int wq_lo = qg * nw * 32;
        int wq_hi = wq_lo + nw * 32 - 1;
        wq_hi = wq_hi < T - 1 ? wq_hi : T - 1;
        gb_hi = p.n_global > 0 ? (p.n_global - 1) >> 5 : -1;
        if (!(p.n_global > 0 && wq_lo < p.n_global)) {
            if (p.att_left >= 0) { const int lo = wq_lo - p.att_left; wg_lo = lo > 0 ? lo >> 5 : 0; }
            if (p.att_right >= 0) { const int hi = (wq_hi + p.att_right) >> 5; wg_hi = hi < wg_hi ? hi : wg_hi; }
        }
        if (!(p.n_global > 0 && i0 < p.n_global)) {
            if (p.att_left >= 0) { const int lo = i0 - p.att_left; w_lo = lo > 0 ? lo >> 5 : 0; }
            if (p.att_right >= 0) { const int hi = (i0 + 31 + p.att_right) >> 5; w_hi = hi < w_hi ? hi : w_hi; }
        }
    }
    auto chunk_needed = [&](int jc, int nb) { return !WINDOW || (jc <= wg_hi && jc + nb - 1 >= wg_lo) || jc <= gb_hi; };
    auto block_needed = [&](int blk) { return !WINDOW || (blk >= w_lo && blk <= w_hi) || blk <= gb_hi; };

    // rows n0 .. n0+31 of the first key block; every later block reuses the previous block's upper half
    // (bd_for = the key block whose lower half bd_lo currently holds)
    f32x16_t bd_lo = bd_block(0 - i0 - 31 + T - 1);
    int bd_for = 0;
    stamp(bd_lo[0]);                               // [4] first position block

    // Keys are staged KB_CHUNK blocks at a time (all of them for T' <= 160): between two workgroup
    // barriers every wave walks its key blocks on its own, so the waves drift apart and hide each
    // other's L2 / LDS latencies instead of marching in lockstep.
    for (int jc = 0; jc < n_kblocks; jc += KB_CHUNK) {
        const int nb = n_kblocks - jc < KB_CHUNK ? n_kblocks - jc : KB_CHUNK;
        if (!chunk_needed(jc, nb)) continue;      // workgroup-uniform
        if (jc > 0 && HD == 64) {
            // HD 64 has the registers (166 + the staging's 64 of 256) to run the prologue's staging path again: every load of
            // the chunk in flight at once, V transposed in registers (the one-item-at-a-time path below cost the ESPnet shape,
            // T' = 358 = three chunks of five, most of its 490 us per launch: profiles/r04zz_beam_kernel_stats.txt)
            __syncthreads();
            stage_kv(jc, nb);
            __syncthreads();
        } else if (jc > 0) {
            __syncthreads();                  // previous chunk fully consumed
            // later chunks (T' > 160 only) restage with everything live: one item at a time, 8 staging VGPRs
#pragma unroll 1
            for (int idx = tid; idx < nb * 32 * NCH; idx += blockDim.x) {
                const int slot = idx / (32 * NCH), key = (idx / NCH) & 31, ch = idx % NCH;
                int krow = (jc + slot) * 32 + key;
                krow = krow < T ? krow : T - 1;
                const uint16_t* kp = base + (size_t)krow * ld + d + h * HD + ch * 8;
                const uint4 kv = *reinterpret_cast<const uint4*>(kp);
                const u16x8_t vv = *reinterpret_cast<const u16x8_t*>(kp + d);
                *reinterpret_cast<uint4*>(Ks + slot * K_BYTES + key * KROW + ch * 16) = kv;
                const int posk = vt_pos(key);
                const int col = (((posk >> 3) ^ (ch & 3)) << 4) + (posk & 7) * 2;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<unsigned short*>(Vts + slot * VT_BYTES + (ch * 8 + e) * VROW + col) = vv[e];
            }
            __syncthreads();
        }

        for (int sl = 0; sl < nb; ++sl) {
            if (!block_needed(jc + sl)) continue; // wave-uniform
            const int j0 = (jc + sl) * 32;
            if (WINDOW && bd_for != jc + sl) bd_lo = bd_block(j0 - i0 - 31 + T - 1);   // blocks were skipped: rebuild the lower half
            const char* ks_t = Ks + sl * K_BYTES;
            const char* vts_t = Vts + sl * VT_BYTES;

            // ---- S^T = K . (Q+u)^T   (8 MFMAs)
            f32x16_t s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks_t + il * KROW + (2 * ks + hh) * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qu[ks], s, 0, 0, 0);
            }
            stamp(s[0]);                           // [5+5k] S^T
            // ---- upper half of BD^T for this key block: relative positions n0+32 .. n0+63   (8 MFMAs)
            const int n0 = j0 - i0 - 31 + T - 1;
            const f32x16_t bd_hi = bd_block(n0 + 32);
            stamp(bd_hi[0]);                       // [6+5k] position rows fetched, BD^T

            // ---- skew through the per-wave scratch, query-major: scr_f[query][n_local], row stride 68 floats.
            //      A lane's register quad (r & 3 = 0..3) is 4 consecutive n_local: 8 x 16-byte stores
            //      (slots 17*il mod 16: conflict-free) instead of 32 x 4-byte; the skewed reads
            //      scr_f[il][jl - il + 31] land in bank (3*il + const) mod 32: conflict-free too.
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(scr_f + il * SKEW_LD + 8 * g + 4 * hh) =
                    make_float4(bd_lo[4 * g], bd_lo[4 * g + 1], bd_lo[4 * g + 2], bd_lo[4 * g + 3]);
                *reinterpret_cast<float4*>(scr_f + il * SKEW_LD + 32 + 8 * g + 4 * hh) =
                    make_float4(bd_hi[4 * g], bd_hi[4 * g + 1], bd_hi[4 * g + 2], bd_hi[4 * g + 3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // all 16 skewed reads are issued together and pinned: left to itself the compiler sinks each
            // read under its own `ok` branch (16 x exec-mask branch + a full LDS round trip each)
            float bdw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bdw[r] = scr_f[il * SKEW_LD + (rowmap(r, hh) - il + 31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bdw[r]));
            float pr[16];
            float mblk = NEG;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = rowmap(r, hh);
                const int j = j0 + jl;
                const float bdv = bdw[r];
                bool ok = q_valid && j < len;
                if constexpr (WINDOW) {
                    bool win = (p.att_left < 0 || qi - j <= p.att_left) && (p.att_right < 0 || j - qi <= p.att_right);
                    if (p.n_global > 0) win = win || qi < p.n_global || j < p.n_global;
                    ok = ok && win;
                }
                const float sc = ok ? (s[r] + bdv) * p.scale : NEG;
                pr[r] = sc;
                mblk = fmaxf(mblk, sc);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();  // scratch reads done before the next block overwrites it
            stamp(pr[0]);                          // [7+5k] skew + masked scores
            bd_lo = bd_hi;
            bd_for = jc + sl + 1;
            mblk = fmaxf(mblk, __shfl_xor(mblk, 32, 64));
            const float m_new = fmaxf(m_run, mblk);
            const float alpha = __expf(m_run - m_new);
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = pr[r] > 0.5f * NEG ? __expf(pr[r] - m_new) : 0.0f;
                pr[r] = e;
                psum += e;
            }
            psum += __shfl_xor(psum, 32, 64);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
            stamp(o[0][0]);                        // [8+5k] softmax update
            // ---- P^T fragments (bf16) and  O^T += V^T . P^T   (8 MFMAs)
            bf16x8_t pf[2];
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const u16x4_t lo = pack_bf16x4(pr[8 * sidx], pr[8 * sidx + 1], pr[8 * sidx + 2], pr[8 * sidx + 3]);
                const u16x4_t hi = pack_bf16x4(pr[8 * sidx + 4], pr[8 * sidx + 5], pr[8 * sidx + 6], pr[8 * sidx + 7]);
                u16x8_t t;
#pragma unroll
                for (int e = 0; e < 4; ++e) { t[e] = lo[e]; t[4 + e] = hi[e]; }
                pf[sidx] = __builtin_bit_cast(bf16x8_t, t);
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int drow = db * 32 + il;
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    const int chunk = (2 * sidx + hh) ^ ((drow >> 3) & 3);
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vts_t + drow * VROW + chunk * 16);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sidx], o[db], 0, 0, 0);
                }
            }
        }
    }

    stamp(o[0][0]);                                // after the last block's PV ([9+5k] inside the loop is implied by the next S stamp)
    // ---- write ctx.  lane = (query il, half hh); reg r of block db -> d = db*32 + rowmap(r, hh): a lane owns
    //      8-byte pieces of its query's row.  They are gathered through the wave's scratch (rows of 272 B)
    //      so that the global stores are whole 256-byte rows, 16 bytes per lane.
    {
        const float inv = (q_valid && l_run > 0.0f) ? 1.0f / l_run : 0.0f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<u16x4_t*>(scr + il * KROW + (db * 32 + 8 * g + 4 * hh) * 2) =
                    pack_bf16x4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 32 / RPP; ++q) {
            const int row = RPP * q + lane / NCH;
            const uint4 v = *reinterpret_cast<const uint4*>(scr + row * KROW + (lane % NCH) * 16);
            if (i0 + row < T)
                *reinterpret_cast<uint4*>(p.out + ((size_t)b * T + i0 + row) * d + h * HD + (lane % NCH) * 8) = v;
        }
    }
    if (!PERSIST || item + (int)gridDim.x >= p.n_items) break;
    __syncthreads();                                       // every wave is done with the K / V image before the next item's staging
    }
    if constexpr (TRACE) {
        stamp(0.0f);
        if (wave == 1 && (tid0 & 63) == 0) {
            long long* tr = p.trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 40;
            for (int i = 0; i < 40; ++i) tr[i] = i < nts ? ts[i] - ts[0] : -1;
        }
    }
}

// =====================================================================================================================
// Streaming form (round 6): full attention (no window), any T.
//
// What held the kernel above at 168 us per launch (head_dim 128, T' = 138, B = 256) is not arithmetic — its MFMAs are 8 % of a
// workgroup's 21 us — but ONE workgroup per CU (147 KB of LDS: the whole K and V^T of a head) whose load phase (4.8 us at the
// HBM rate, every CU at once) and product phase (9 us of dependent steps, 1.25 waves per SIMD) never overlap.  Here:
//   * a wave owns 16 queries (v_mfma_f32_16x16x32_bf16: S^T, BD^T and O^T tiles of 16 queries; half the registers per wave),
//     a workgroup NW such waves; K and V rows stream through a ring of NSLOT 32-key slots, filled by global_load_lds DMAs
//     (no staging registers, no register transposition) two key blocks ahead: 71 KB (head_dim 128) / 46 KB (64) per
//     workgroup, so two / three workgroups share a CU and one's loads run under the other's products;
//   * K rows sit in LDS as they are in memory, 16-byte pieces XOR-swizzled by the row (conflict-free ds_read_b128 of the A
//     fragments); V rows too, 32-byte units swizzled by the row, and the V^T fragments of O^T += V^T . P^T come out of
//     ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group turns a [4 keys][16 d] block into column-major
//     registers).  The keys a lane contributes to the contraction are the keys its own S^T registers hold (4 kq + r of
//     each 16-key tile), so the probabilities feed the PV product from registers;
//   * the relative-position rows go global -> VGPR as MFMA A fragments (the table of a head is 70 KB and L2-resident),
//     issued one key block ahead; of the three 16-row tiles a block needs, one is the previous block's last;
//   * every global access of the loop is issued from inline asm, so ONE in-order queue is waited on with counted vmcnt:
//     per iteration a wave issues  pos(k + 1) [2 KS loads], then its share of DMA(k + 2).
// Same arithmetic as above (scores in f32, probabilities rounded to bf16 for the PV product, online softmax per 32-key
// block); the summation order inside a product differs (16x16x32 instead of 32x32x16 fragments), so the two forms agree to
// rounding, not bit for bit.
//
// MEASURED (profiles/r06_10_attn_forms_ab.txt, same box, interleaved): correct on the first run (same error against a float32
// reference as the staged kernel) and SLOWER — 178.8 vs 141.4 us (head_dim 128, B = 256, T' = 138), 417.6 vs 345.2 us
// (head_dim 64, T' = 358), 344 vs 237 us (B = 8, T' <= 1500) — so it is NOT the default ($RS_ATTN_STREAM=1 selects it; a GPU
// test keeps it correct).  Where its time goes, by leaving phases out ($RS_ATTN_STREAM = 1 + 16 x mask, `dbg` below;
// profiles/r06_10_attn_stream_phases.txt; head_dim 128, T' = 138, 217 us with the debug branches compiled in): prologue + ctx
// store of 4096 workgroups alone 60 us; + the loop's scalar softmax arithmetic 42 (a wave64 VALU instruction occupies its
// SIMD for four cycles: at 2.5 waves per SIMD the loop is VALU-issue bound, not latency bound — the per-iteration address
// arithmetic of the DMAs, the position loads and the swizzled fragment reads is as many instructions as the softmax itself);
// + the per-block workgroup barrier 20; + the position loads 29; + DMAs, S^T, BD^T, skew, PV 65.  Halving the queries per wave
// halved the registers but doubled every per-block fixed cost per query; what the staged kernel amortises over 32 queries
// (addresses, barrier, waits) this one pays per 16.  The lesson for a next attempt is in that list: the per-block fixed
// instruction count per query has to go DOWN, not the latency.
template <int HD>
struct SGeom {
    static constexpr int KS = HD / 32;          // 32-deep k-steps of a 16x16x32 product over head_dim
    static constexpr int DT = HD / 16;          // 16-row tiles of O^T
    static constexpr int ROWB = HD * 2;         // bytes per K / V row
    static constexpr int SPR = HD / 8;          // 16-byte pieces per row
    static constexpr int RPI = 64 / SPR;        // rows one DMA instruction (64 lanes x 16 bytes) covers
    static constexpr int HALF = 32 * ROWB;      // the K (or V) rows of a 32-key block
    static constexpr int SLOT = 2 * HALF;
    static constexpr int NI = SLOT / 1024;      // DMA instructions per key block
};
constexpr int S_PITCH = 272;                    // bytes per query row of a wave's scratch (68 floats: skew tile; HD bf16: ctx rows)
constexpr int S_SCR = 16 * S_PITCH;

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

template <int HD> __device__ __forceinline__ int swk(int r) { return HD == 128 ? (r & 15) : ((r >> 1) & 7); }
template <int HD> __device__ __forceinline__ int swv(int r) { return HD == 128 ? (r & 7) : ((r >> 1) & 3); }

// M0 (the DMA's LDS base) is written and consumed inside one asm statement; nothing else in this kernel uses M0
__device__ __forceinline__ void a_glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int OFF>
__device__ __forceinline__ void a_gload16(u32x4_t& dst, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void a_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
// BASE loads are in flight behind the one waited for, plus this wave's DMAs of one key block (ND or ND - 1 of them) when
// `dma` says they were issued; the arguments are wave-uniform: scalar branches to literal waits
template <int BASE, int ND>
__device__ __forceinline__ void a_wait_behind(bool dma, int nd) {
    if (!dma) a_wait_vm<BASE>();
    else if (nd == ND) a_wait_vm<BASE + ND>();
    else a_wait_vm<BASE + ND - 1>();
}
__device__ __forceinline__ float x16_max(float x) {      // max over the lanes l ^ 16 (v_permlane16_swap: no LDS round trip)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float x32_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float x16_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float x32_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// grid: 8 x ceil(items / 8) workgroups, items = (query groups of NW tiles) x heads x batch; the workgroups of one XCD
// (id mod 8) walk a contiguous run of items, so the query groups of a (batch, head) — which stream the same K / V — share an L2
// DBG: phases can be left out by a mask (timing decomposition only: the results are then wrong)
template <int HD, int NW, int NSLOT, int WPE, bool DBG>
__global__ __launch_bounds__(64 * NW, WPE) void relpos_attention_stream_kernel(AttnParams p, int n_groups, int n_heads, int n_items, int dbg_mask) {
    const int dbg = DBG ? dbg_mask : 0;
    using G = SGeom<HD>;
    constexpr int KS = G::KS, DT = G::DT, ROWB = G::ROWB, SPR = G::SPR, RPI = G::RPI, HALF = G::HALF, SLOT = G::SLOT, NI = G::NI;
    static_assert(NSLOT == 3, "the counted waits below are written for DMAs two key blocks ahead");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int item;
    {
        const int per_xcd = (n_items + 7) >> 3;
        item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= per_xcd || item >= n_items) return;
    }
    const int grp = item % n_groups, h = (item / n_groups) % n_heads, b = item / (n_groups * n_heads);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    char* scr = smem + NSLOT * SLOT + wave * S_SCR;
    float* scr_f = reinterpret_cast<float*>(scr);
    const int T = p.T, d = p.d_model, ld = 3 * d;
    const int len = p.lens[b] < T ? p.lens[b] : T;
    const int nb = (dbg & 128) ? 0 : (len + 31) >> 5;                      // key blocks at / after len are fully masked: never visited
    const int i0 = (grp * NW + wave) * 16;
    const int li = lane & 15, kq = lane >> 4;
    const int qi = i0 + li;
    const bool q_valid = qi < len;
    const char* base = reinterpret_cast<const char*>(p.qkv + (size_t)b * T * ld);
    const unsigned ldb = (unsigned)ld * 2;
    constexpr int NDMAX = (NI + NW - 1) / NW;
    static_assert(NI >= NW, "every wave issues NDMAX or NDMAX - 1 DMA instructions per key block");
    const int nd = (NI - wave + NW - 1) / NW;              // DMA instructions this wave issues per key block

    auto issue_block = [&](int blk, int sl) {
#pragma unroll
        for (int it = 0; it < (NI + NW - 1) / NW; ++it) {
            const int ins = wave + NW * it;              // wave-uniform
            if (ins < NI) {
                const bool isv = ins >= NI / 2;
                const int m = isv ? ins - NI / 2 : ins;
                const int r = RPI * m + lane / SPR, sp = lane % SPR;
                const int piece = isv ? ((((sp >> 1) ^ swv<HD>(r)) << 1) | (sp & 1)) : (sp ^ swk<HD>(r));
                int krow = blk * 32 + r;
                krow = krow < T ? krow : T - 1;
                const unsigned voff = (unsigned)krow * ldb + (unsigned)(((isv ? 2 * d : d) + h * HD) * 2 + piece * 16);
                a_glds16(voff, base, lds0 + sl * SLOT + (isv ? HALF : 0) + m * 1024);
            }
        }
    };
    const char* pos_h = reinterpret_cast<const char*>(p.pos + h * HD);
    const int n_pos = 2 * T - 1;
    auto issue_pos = [&](int nrow0, u32x4_t (&dst)[KS]) {       // A fragments of 16 relative-position rows
        int n = nrow0 + li;
        n = n < 0 ? 0 : (n >= n_pos ? n_pos - 1 : n);
        const unsigned off = (unsigned)n * (unsigned)(d * 2) + kq * 16;
        a_gload16<0>(dst[0], off, pos_h);
        a_gload16<64>(dst[1], off, pos_h);
        if constexpr (KS == 4) { a_gload16<128>(dst[2], off, pos_h); a_gload16<192>(dst[3], off, pos_h); }
    };
    auto pos_fence = [&](u32x4_t (&r)[KS]) {
        if constexpr (KS == 4) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
        else asm volatile("" : "+v"(r[0]), "+v"(r[1]) :: "memory");
    };

    // ---- prologue: Q rows and biases (plain loads: the compiler's own waits, conservative beside the asm queue), DMA(0),
    //      the first position tile
    const int qrow = qi < T ? qi : T - 1;
    const uint16_t* qp = p.qkv + ((size_t)b * T + qrow) * ld + h * HD + 8 * kq;
    u16x8_t raw[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) raw[ks] = *reinterpret_cast<const u16x8_t*>(qp + 32 * ks);
    bf16x8_t qu[KS], qv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = h * HD + 32 * ks + 8 * kq;
        float bu[8], bvv[8];
        *reinterpret_cast<float4*>(&bu[0]) = *reinterpret_cast<const float4*>(p.bias_u + c);
        *reinterpret_cast<float4*>(&bu[4]) = *reinterpret_cast<const float4*>(p.bias_u + c + 4);
        *reinterpret_cast<float4*>(&bvv[0]) = *reinterpret_cast<const float4*>(p.bias_v + c);
        *reinterpret_cast<float4*>(&bvv[4]) = *reinterpret_cast<const float4*>(p.bias_v + c + 4);
        u16x8_t a, bq;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = bf16_to_f32(raw[ks][e]);
            a[e] = f32_to_bf16(f + bu[e]);
            bq[e] = f32_to_bf16(f + bvv[e]);
        }
        qu[ks] = __builtin_bit_cast(bf16x8_t, a);
        qv[ks] = __builtin_bit_cast(bf16x8_t, bq);
    }
    u32x4_t pf0[KS], pf1[KS];
    int n0 = 0 - i0 - 15 + T - 1;                         // relative position of (key 0, query i0 + 15)
    if (nb > 0) issue_block(0, 0);
    issue_pos(n0, pf0);
    a_wait_vm<0>();
    pos_fence(pf0);
    f32x4_t carry = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        carry = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf0[ks]), qv[ks], carry, 0, 0, 0);
    issue_pos(n0 + 16, pf0);
    issue_pos(n0 + 32, pf1);
    if (nb > 1 && !(dbg & 8)) issue_block(1, 1);

    f32x4_t o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = NEG, l_run = 0.0f;

    // per-lane constants of the V^T reads: lane i of a 16-lane group hands ds_read_b64_tr_b16 the address of
    // V[key 4 kq + i / 4][d0 + 4 (i mod 4) ..+3] and receives V[4 kq + 0..3][d0 + i]
    const int vrow = 4 * kq + (li >> 2);
    const unsigned v_lane = (unsigned)(vrow * ROWB + 8 * (li & 3));
    const int v_x = swv<HD>(vrow);                        // = swv(vrow + 16)
    const unsigned k_lane = (unsigned)(li * ROWB);
    const int k_x = swk<HD>(li);                          // = swk(li + 16)

    int sl = 0;
    for (int k = 0; k < nb; ++k) {
        const bool more1 = k + 1 < nb, more2 = k + 2 < nb;
        // in flight, oldest first:  DMA(k) | pos(k) | DMA(k + 1)
        if (dbg & 32) a_wait_vm<0>(); else
        a_wait_behind<2 * KS, NDMAX>(more1 && !(dbg & 8), nd);
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 64)) __builtin_amdgcn_s_barrier();                     // block k is in LDS; every wave is done with block k - 1
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        const unsigned kt = lds0 + sl * SLOT, vt = kt + HALF;

        // ---- S^T = K . (Q + u)^T : two 16-key tiles
        f32x4_t s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
        if (!(dbg & 16))
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned a = kt + k_lane + (unsigned)(((4 * ks + kq) ^ k_x) * 16);
            const bf16x8_t k0 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8_t*>(a);
            const bf16x8_t k1 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8_t*>(a + 16 * ROWB);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qu[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qu[ks], s1, 0, 0, 0);
        }
        // ---- BD^T: the two new 16-row tiles of this block
        a_wait_behind<0, NDMAX>(more1 && !(dbg & 8), nd);
        pos_fence(pf0);
        pos_fence(pf1);
        f32x4_t bd1 = {0.0f, 0.0f, 0.0f, 0.0f}, bd2 = bd1;
        if (!(dbg & 2))
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bd1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf0[ks]), qv[ks], bd1, 0, 0, 0);
            bd2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf1[ks]), qv[ks], bd2, 0, 0, 0);
        }
        // ---- next block's position rows (always issued: a constant count, rows clamped), then the DMAs two blocks ahead
        n0 += 32;
        asm volatile("" : "+v"(bd1), "+v"(bd2));          // the fragments have been consumed before their registers are reloaded
        if (!(dbg & 32)) {
        issue_pos(n0 + 16, pf0);
        issue_pos(n0 + 32, pf1);
        }
        if (more2 && !(dbg & 8)) issue_block(k + 2, sl == 0 ? 2 : sl - 1);

        // ---- skew through the wave's scratch, query-major: scr_f[query][n_local], n_local = key_local - query + 15
        float bdw[8];
        if (dbg & 4) {
#pragma unroll
            for (int r = 0; r < 8; ++r) bdw[r] = r < 4 ? bd1[r & 3] : bd2[r & 3];
            carry = bd2;
        } else {
        *reinterpret_cast<f32x4_t*>(scr_f + li * 68 + 4 * kq) = carry;
        *reinterpret_cast<f32x4_t*>(scr_f + li * 68 + 16 + 4 * kq) = bd1;
        *reinterpret_cast<f32x4_t*>(scr_f + li * 68 + 32 + 4 * kq) = bd2;
        carry = bd2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) bdw[r] = scr_f[li * 68 + (16 * (r >> 2) + 4 * kq + (r & 3) - li + 15)];
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(bdw[r]));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        }
        const int j0 = 32 * k;
        float pr[8];
        float mblk = NEG;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int j = j0 + 16 * (r >> 2) + 4 * kq + (r & 3);
            const float sv = r < 4 ? s0[r & 3] : s1[r & 3];
            const float sc = (q_valid && j < len) ? (sv + bdw[r]) * p.scale : NEG;
            pr[r] = sc;
            mblk = fmaxf(mblk, sc);
        }
        mblk = x32_max(x16_max(mblk));
        const float m_new = fmaxf(m_run, mblk);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float e = pr[r] > 0.5f * NEG ? __expf(pr[r] - m_new) : 0.0f;
            pr[r] = e;
            psum += e;
        }
        l_run = l_run * alpha + psum;                     // per lane: the four lanes of a query are added up at the end
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[i] *= alpha;
        // ---- O^T += V^T . P^T
        bf16x8_t pfrag;
        {
            const u16x4_t lo = pack_bf16x4(pr[0], pr[1], pr[2], pr[3]);
            const u16x4_t hi = pack_bf16x4(pr[4], pr[5], pr[6], pr[7]);
            u16x8_t t;
#pragma unroll
            for (int e = 0; e < 4; ++e) { t[e] = lo[e]; t[4 + e] = hi[e]; }
            pfrag = __builtin_bit_cast(bf16x8_t, t);
        }
        if (!(dbg & 1))
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const unsigned a = vt + v_lane + (unsigned)((dt ^ v_x) * 32);
            const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s16x4_t*>(a));
            const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s16x4_t*>(a + 16 * ROWB));
            u16x8_t t;
#pragma unroll
            for (int e = 0; e < 4; ++e) { t[e] = (unsigned short)v0[e]; t[4 + e] = (unsigned short)v1[e]; }
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, t), pfrag, o[dt], 0, 0, 0);
        }
        sl = sl == 2 ? 0 : sl + 1;
    }
    a_wait_vm<0>();                                       // the last iteration's (unused) position rows

    // ---- ctx rows: O^T tiles -> the wave's scratch as bf16 rows -> whole-row global stores
    {
        const float l = x32_sum(x16_sum(l_run));
        const float inv = (q_valid && l > 0.0f) ? 1.0f / l : 0.0f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            *reinterpret_cast<u16x4_t*>(scr + li * S_PITCH + (16 * dt + 4 * kq) * 2) =
                pack_bf16x4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16 / RPI; ++q) {
            const int row = RPI * q + lane / SPR;
            const uint4 v = *reinterpret_cast<const uint4*>(scr + row * S_PITCH + (lane % SPR) * 16);
            if (i0 + row < T)
                *reinterpret_cast<uint4*>(p.out + ((size_t)b * T + i0 + row) * d + h * HD + (lane % SPR) * 8) = v;
        }
    }
}

long long* g_attn_trace = nullptr;
std::atomic<int> g_attn_persist{-1};         // -1: $RS_ATTN_PERSIST (default 0: measured slower) on first use
int attn_persist() {
    int v = g_attn_persist.load();
    if (v < 0) { const char* e = getenv("RS_ATTN_PERSIST"); v = e ? atoi(e) : 0; g_attn_persist = v; }
    return v;
}
std::atomic<int> g_attn_stream{-1};          // -1: $RS_ATTN_STREAM (default 0: measured slower, see the kernel's header) on first use
int attn_stream() {
    int v = g_attn_stream.load();
    if (v < 0) { const char* e = getenv("RS_ATTN_STREAM"); v = e ? atoi(e) : 0; g_attn_stream = v; }
    return v;
}

}  // namespace

// debug hook (scripts/attn_trace.py): buffer of 40 int64 per workgroup, or nullptr to switch tracing off
extern "C" void rs_debug_set_attn_trace(long long* buf) { g_attn_trace = buf; }
// A/B hook: 1 = the streaming form for full attention (default), 0 = the staged kernel
extern "C" void rs_debug_set_attn_stream(int v) { g_attn_stream = v; }
// A/B hook: 1 = full attention runs on resident workgroups that walk the (query group, head, utterance) items (n >= 2: on
// exactly n of them), 0 = one workgroup per item (default)
extern "C" void rs_debug_set_attn_persist(int v) { g_attn_persist = v; }

namespace {

template <int HD>
int launch_attention_hd(rs_ctx* ctx, AttnParams& p, int B, int T, hipStream_t s) {
    using G = AttnGeom<HD>;
    const rs_dims& dm = ctx->d;
    p.scale = 1.0f / sqrtf((float)HD);
    const int qblocks = (T + 31) / 32;
    // at most 6 query blocks per workgroup (HD 128: 5 * (8704 + 10240) + 6 * 8704 = 147 KB of the 160 KB LDS)
    const int nw = qblocks < 6 ? qblocks : 6;
    const dim3 grid((qblocks + nw - 1) / nw, dm.n_heads, B), block(64 * nw);
    constexpr int KB_CHUNK = G::KB_CHUNK;
    const size_t lds = (size_t)KB_CHUNK * (G::K_BYTES + G::VT_BYTES) + (size_t)nw * G::SCR_BYTES + 2 * HD * sizeof(float);
    {
        constexpr int MAX_LDS = KB_CHUNK * (G::K_BYTES + G::VT_BYTES) + 6 * G::SCR_BYTES + 2 * HD * (int)sizeof(float);
        int rc = rs_ensure_dynamic_lds(ctx, (const void*)relpos_attention_kernel<HD, false, false>, MAX_LDS);
        if (rc == RS_OK) rc = rs_ensure_dynamic_lds(ctx, (const void*)relpos_attention_kernel<HD, false, true>, MAX_LDS);
        if constexpr (HD == 128)
            if (rc == RS_OK) rc = rs_ensure_dynamic_lds(ctx, (const void*)relpos_attention_kernel<HD, true, false>, MAX_LDS);
        if (rc != RS_OK) return rc;
    }
    // algorithmic: ac + bd + pv = 3 * 2*T*T*HD per (b,h)
    const double flops = (double)B * dm.n_heads * 3.0 * 2.0 * T * (double)T * HD;
    const double bytes = (double)B * T * dm.d_model * 2.0 * 4.0;
    rs_prof_begin(ctx, RS_PROF_ATTN, s, flops, bytes);
    p.trace = HD == 128 ? g_attn_trace : nullptr;
    const bool window = p.att_left >= 0 || p.att_right >= 0;
    // Persistent launch of the staged kernel ($RS_ATTN_PERSIST=1; off): as many workgroups as the chip holds at once, each
    // walking its items in one launch.  The idea: a launch takes 160.7 us against 8 rounds x 16.2 us of traced workgroup time
    // (profiles/r01p_attention_phase_timeline.txt) — if the 4 us per round were dispatch cost, a resident workgroup would pay it
    // once.  Measured (profiles/r06_10_attn_forms_ab.txt, bit-identical): SLOWER, 145.5 -> 150.5 us (head_dim 128, T' = 138),
    // 343.5 -> 386.4 us (head_dim 64, T' = 358): the hardware dispatcher refills a CU as soon as a workgroup's LDS is free,
    // an in-kernel item loop waits at its barrier for the slowest wave first.
    int resident = 0;
    auto persist_grid = [&](const dim3& g, size_t lds_bytes, int threads) {
        if (attn_persist() <= 0 || window || p.trace) return false;
        if (ctx->n_cus <= 0) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
            ctx->n_cus = n;
        }
        int per_cu = (int)((160 * 1024) / lds_bytes);
        const int by_threads = 2048 / threads;
        per_cu = per_cu < by_threads ? per_cu : by_threads;
        if (per_cu < 1) per_cu = 1;
        resident = attn_persist() >= 2 ? attn_persist() : per_cu * ctx->n_cus;      // (>= 2: that many resident workgroups — tests)
        p.n_groups = (int)g.x; p.n_heads = (int)g.y; p.n_items = (int)(g.x * g.y * g.z);
        return p.n_items > resident;                    // fewer items than slots: the classic grid is the same thing
    };
    {
        // the streaming form: full attention, any T ($RS_ATTN_STREAM=0: the staged kernel above, for the A/B)
        if (attn_stream() > 0 && !window && !p.trace) {
            using S = SGeom<HD>;
            constexpr int NW = 5, NSLOT = 3, WPE = HD == 128 ? 3 : 4;
            const int qtiles = (T + 15) / 16, n_groups = (qtiles + NW - 1) / NW;
            const int n_items = n_groups * dm.n_heads * B;
            const size_t lds_s = (size_t)NSLOT * S::SLOT + (size_t)NW * S_SCR;
            const int dbg = attn_stream() >> 4;
            auto kern = dbg ? relpos_attention_stream_kernel<HD, NW, NSLOT, WPE, true> : relpos_attention_stream_kernel<HD, NW, NSLOT, WPE, false>;
            if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)kern, (int)lds_s); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_ATTN, s); return rc; }
            hipLaunchKernelGGL(kern, dim3(8 * ((n_items + 7) / 8)), dim3(64 * NW), lds_s, s, p, n_groups, dm.n_heads, n_items, dbg);
            rs_prof_end(ctx, RS_PROF_ATTN, s);
            RS_CHECK_LAUNCH(ctx, "relpos_attention (streaming)");
            return RS_OK;
        }
    }
    if constexpr (HD == 64) {
        // head_dim 64: small key chunks and fewer waves per workgroup so that TWO workgroups share a CU and the K / V staging of
        // one overlaps the products of the other (default 4 key blocks x 4 waves: 74 KB; $RS_ATTN64="kbc,nw" for the A/B,
        // "0" = the one-workgroup geometry).  Same order of the online softmax: bit-identical.
        int kbc = 4, nwmax = 4;
        if (const char* e = getenv("RS_ATTN64")) { kbc = atoi(e); const char* c = strchr(e, ','); nwmax = c ? atoi(c + 1) : 4; }
        if (kbc > 0 && !window && !p.trace) {
            const int nw2 = qblocks < nwmax ? qblocks : (nwmax < 1 ? 1 : nwmax > 6 ? 6 : nwmax);
            const dim3 grid2((qblocks + nw2 - 1) / nw2, dm.n_heads, B), block2(64 * nw2);
            int rc = RS_EINVAL;
            auto go = [&](auto kern, auto kern_p, int KBC) {
                const size_t lds2 = (size_t)KBC * (G::K_BYTES + G::VT_BYTES) + (size_t)nw2 * G::SCR_BYTES + 2 * HD * sizeof(float);
                const bool pers = persist_grid(grid2, lds2, 64 * nw2);
                rc = rs_ensure_dynamic_lds(ctx, pers ? (const void*)kern_p : (const void*)kern, (int)lds2);
                if (rc != RS_OK) return;
                if (pers) hipLaunchKernelGGL(kern_p, dim3(resident), block2, lds2, s, p);
                else hipLaunchKernelGGL(kern, grid2, block2, lds2, s, p);
            };
            // stage_kv's register budget: nb <= 2 nw key blocks per chunk
            if (kbc == 2 && nw2 >= 1) go(relpos_attention_kernel<HD, false, false, 2>, relpos_attention_kernel<HD, false, false, 2, true>, 2);
            else if (kbc == 3 && nw2 >= 2) go(relpos_attention_kernel<HD, false, false, 3>, relpos_attention_kernel<HD, false, false, 3, true>, 3);
            else if (kbc == 4 && nw2 >= 2) go(relpos_attention_kernel<HD, false, false, 4>, relpos_attention_kernel<HD, false, false, 4, true>, 4);
            else if (nw2 == 1) go(relpos_attention_kernel<HD, false, false, 2>, relpos_attention_kernel<HD, false, false, 2, true>, 2);
            rs_prof_end(ctx, RS_PROF_ATTN, s);
            if (rc != RS_OK) return rc == RS_EINVAL ? rs_fail(ctx, RS_EINVAL, "attention: $RS_ATTN64 geometry %d,%d is not built", kbc, nwmax) : rc;
            RS_CHECK_LAUNCH(ctx, "relpos_attention (head_dim 64)");
            return RS_OK;
        }
    }
    if (window) hipLaunchKernelGGL((relpos_attention_kernel<HD, false, true>), grid, block, lds, s, p);
    else if (p.trace) {
        if constexpr (HD == 128) hipLaunchKernelGGL((relpos_attention_kernel<HD, true, false>), grid, block, lds, s, p);
    } else if (persist_grid(grid, lds, 64 * nw)) {
        if (int rc = rs_ensure_dynamic_lds(ctx, (const void*)relpos_attention_kernel<HD, false, false, 0, true>, (int)lds); rc != RS_OK) { rs_prof_end(ctx, RS_PROF_ATTN, s); return rc; }
        hipLaunchKernelGGL((relpos_attention_kernel<HD, false, false, 0, true>), dim3(resident), block, lds, s, p);
    } else hipLaunchKernelGGL((relpos_attention_kernel<HD, false, false>), grid, block, lds, s, p);
    rs_prof_end(ctx, RS_PROF_ATTN, s);
    RS_CHECK_LAUNCH(ctx, "relpos_attention");
    return RS_OK;
}

}  // namespace

int rs_launch_attention(rs_ctx* ctx, const uint16_t* qkv, const uint16_t* pos, const float* bias_u,
                        const float* bias_v, const int32_t* lens, int B, int T, uint16_t* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return RS_OK;
    const rs_dims& dm = ctx->d;
    const int hd = dm.n_heads > 0 ? dm.d_model / dm.n_heads : 0;
    AttnParams p;
    p.n_items = p.n_groups = p.n_heads = 0;
    p.qkv = qkv; p.pos = pos; p.bias_u = bias_u; p.bias_v = bias_v; p.lens = lens; p.out = out;
    p.T = T; p.d_model = dm.d_model; p.att_left = dm.att_left; p.att_right = dm.att_right; p.n_global = dm.n_global;
    if (hd == 128) return launch_attention_hd<128>(ctx, p, B, T, s);
    if (hd == 64) return launch_attention_hd<64>(ctx, p, B, T, s);
    return rs_fail(ctx, RS_EINVAL, "attention: head_dim %d (128 or 64 are built)", hd);
}

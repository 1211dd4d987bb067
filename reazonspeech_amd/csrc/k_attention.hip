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
template <int HD, bool TRACE, bool WINDOW, int KBC = 0>
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* scr = smem + KB_CHUNK * (K_BYTES + VT_BYTES) + wave * SCR_BYTES;
    float* bias_s = reinterpret_cast<float*>(smem + KB_CHUNK * (K_BYTES + VT_BYTES) + nw * SCR_BYTES);   // [u(128) | v(128)]
    float* scr_f = reinterpret_cast<float*>(scr);

    const int b = blockIdx.z, h = blockIdx.y;
    const int T = p.T, d = p.d_model, ld = 3 * d;
    const int len = p.lens[b];
    const int i0 = (blockIdx.x * nw + wave) * 32;
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
        const int wq_lo = blockIdx.x * nw * 32;
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
    if constexpr (TRACE) {
        stamp(0.0f);
        if (wave == 1 && lane == 0) {
            long long* tr = p.trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 40;
            for (int i = 0; i < 40; ++i) tr[i] = i < nts ? ts[i] - ts[0] : -1;
        }
    }
}

long long* g_attn_trace = nullptr;

}  // namespace

// debug hook (scripts/attn_trace.py): buffer of 40 int64 per workgroup, or nullptr to switch tracing off
extern "C" void rs_debug_set_attn_trace(long long* buf) { g_attn_trace = buf; }

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
            auto go = [&](auto kern, int KBC) {
                const size_t lds2 = (size_t)KBC * (G::K_BYTES + G::VT_BYTES) + (size_t)nw2 * G::SCR_BYTES + 2 * HD * sizeof(float);
                rc = rs_ensure_dynamic_lds(ctx, (const void*)kern, (int)lds2);
                if (rc == RS_OK) hipLaunchKernelGGL(kern, grid2, block2, lds2, s, p);
            };
            // stage_kv's register budget: nb <= 2 nw key blocks per chunk
            if (kbc == 2 && nw2 >= 1) go(relpos_attention_kernel<HD, false, false, 2>, 2);
            else if (kbc == 3 && nw2 >= 2) go(relpos_attention_kernel<HD, false, false, 3>, 3);
            else if (kbc == 4 && nw2 >= 2) go(relpos_attention_kernel<HD, false, false, 4>, 4);
            else if (nw2 == 1) go(relpos_attention_kernel<HD, false, false, 2>, 2);
            rs_prof_end(ctx, RS_PROF_ATTN, s);
            if (rc != RS_OK) return rc == RS_EINVAL ? rs_fail(ctx, RS_EINVAL, "attention: $RS_ATTN64 geometry %d,%d is not built", kbc, nwmax) : rc;
            RS_CHECK_LAUNCH(ctx, "relpos_attention (head_dim 64)");
            return RS_OK;
        }
    }
    if (window) hipLaunchKernelGGL((relpos_attention_kernel<HD, false, true>), grid, block, lds, s, p);
    else if (p.trace) {
        if constexpr (HD == 128) hipLaunchKernelGGL((relpos_attention_kernel<HD, true, false>), grid, block, lds, s, p);
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
    p.qkv = qkv; p.pos = pos; p.bias_u = bias_u; p.bias_v = bias_v; p.lens = lens; p.out = out;
    p.T = T; p.d_model = dm.d_model; p.att_left = dm.att_left; p.att_right = dm.att_right; p.n_global = dm.n_global;
    if (hd == 128) return launch_attention_hd<128>(ctx, p, B, T, s);
    if (hd == 64) return launch_attention_hd<64>(ctx, p, B, T, s);
    return rs_fail(ctx, RS_EINVAL, "attention: head_dim %d (128 or 64 are built)", hd);
}

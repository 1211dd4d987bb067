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
#include "rs_common.h"

namespace {

constexpr int HD = 128;             // head dim (fixed)
constexpr int KROW = 272;           // bytes per K / P row in LDS (256 + 16 pad: conflict-free b128 reads)
constexpr int VROW = 80;            // bytes per V^T row in LDS (64 + 16 pad)
constexpr int K_BYTES = 32 * KROW;  // 8704
constexpr int VT_BYTES = HD * VROW; // 10240
constexpr int SCR_BYTES = 32 * KROW;   // per-wave scratch: 32 staged P rows, later the [64][32] f32 skew tile
constexpr int KB_CHUNK = 5;            // key blocks staged per workgroup barrier (160 keys: all of T' = 138)
constexpr float NEG = -1.0e30f;

struct AttnParams {
    const uint16_t* qkv; const uint16_t* pos; const float* bias_u; const float* bias_v;
    const int32_t* lens; uint16_t* out;
    int T, d_model, att_left, att_right, n_global;
    float scale;
};

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// position of key `key` (0..31) inside a V^T row: the order in which the S^T accumulator registers
// of a lane enumerate the keys, so P^T can feed the PV MFMA straight from registers
__device__ __forceinline__ int vt_pos(int key) {
    const int kh = (key >> 2) & 1, kr = (key & 3) + 4 * (key >> 3);
    return (kr >> 3) * 16 + kh * 8 + (kr & 7);
}

// One workgroup = (batch b, head h, up to 6 query blocks of 32); one wave = one query block.
__global__ __launch_bounds__(384) void relpos_attention_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                              // [KB_CHUNK][32 keys][272 B]
    char* Vts = smem + KB_CHUNK * K_BYTES;        // [KB_CHUNK][128 d][80 B]
    const int nw = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* scr = smem + KB_CHUNK * (K_BYTES + VT_BYTES) + wave * SCR_BYTES;
    float* scr_f = reinterpret_cast<float*>(scr);

    const int b = blockIdx.z, h = blockIdx.y;
    const int T = p.T, d = p.d_model, ld = 3 * d;
    const int len = p.lens[b];
    const int i0 = (blockIdx.x * nw + wave) * 32;
    const int il = lane & 31, hh = lane >> 5;
    const int qi = i0 + il;
    const bool q_valid = qi < len && qi < T;
    const uint16_t* base = p.qkv + (size_t)b * T * ld;

    // ---- Q fragments (+u, +v), rounded to bf16: B operand, lane = (query il, k-chunk hh)
    bf16x8_t qu[8], qv[8];
    {
        const int qrow = qi < T ? qi : T - 1;
        const uint16_t* qp = base + (size_t)qrow * ld + h * HD;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int c = 8 * (2 * ks + hh);
            const u16x8_t raw = *reinterpret_cast<const u16x8_t*>(qp + c);
            u16x8_t a, bq;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf16_to_f32(raw[e]);
                a[e] = f32_to_bf16(f + p.bias_u[h * HD + c + e]);
                bq[e] = f32_to_bf16(f + p.bias_v[h * HD + c + e]);
            }
            qu[ks] = __builtin_bit_cast(bf16x8_t, a);
            qv[ks] = __builtin_bit_cast(bf16x8_t, bq);
        }
    }

    f32x16_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.0f;
    float m_run = NEG, l_run = 0.0f;

    const int n_kblocks = ((len < T ? len : T) + 31) >> 5;  // key blocks at/after len are fully masked
    const uint16_t* pos_h = p.pos + h * HD;
    const int n_pos = 2 * T - 1;

    // ---- K / V staging of one key block into slot `slot`: thread -> (key, 16-byte chunk)
    auto stage_kv = [&](int j0, int slot) {
        char* ks = Ks + slot * K_BYTES;
        char* vts = Vts + slot * VT_BYTES;
        for (int idx = tid; idx < 32 * 16; idx += blockDim.x) {
            const int key = idx >> 4, ch = idx & 15;
            int krow = j0 + key;
            krow = krow < T ? krow : T - 1;
            const uint16_t* kp = base + (size_t)krow * ld + d + h * HD + ch * 8;
            const u16x8_t kv = *reinterpret_cast<const u16x8_t*>(kp);
            const u16x8_t vv = *reinterpret_cast<const u16x8_t*>(kp + d);
            *reinterpret_cast<u16x8_t*>(ks + key * KROW + ch * 16) = kv;
            const int posk = vt_pos(key);
            // 16-byte chunk of the key axis XOR-ed with (d>>3)&3 = ch&3: the 16 threads that share a
            // key hit 4 bank groups instead of one
            const int col = (((posk >> 3) ^ (ch & 3)) << 4) + (posk & 7) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<unsigned short*>(vts + (ch * 8 + e) * VROW + col) = vv[e];
        }
    };

    // ---- BD^T block: 32 relative-position rows starting at nrow0, through the wave's scratch
    //      (coalesced 256-byte row reads from L2 -> LDS rows of 272 B -> conflict-free fragment reads)
    auto bd_block = [&](int nrow0) -> f32x16_t {
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // two passes of 4 rows-of-4: 16 staging VGPRs, not 32
            uint4 pr4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int n = nrow0 + 4 * (4 * half + q) + (lane >> 4);
                n = n < 0 ? 0 : (n >= n_pos ? n_pos - 1 : n);
                pr4[q] = *reinterpret_cast<const uint4*>(pos_h + (size_t)n * d + (lane & 15) * 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4*>(scr + (4 * (4 * half + q) + (lane >> 4)) * KROW + (lane & 15) * 16) = pr4[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(scr + il * KROW + (2 * ks + hh) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qv[ks], acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // fragment reads done before the scratch is reused
        return acc;
    };

    // rows n0 .. n0+31 of the first key block; every later block reuses the previous block's upper half
    f32x16_t bd_lo = bd_block(0 - i0 - 31 + T - 1);

    // Keys are staged KB_CHUNK blocks at a time (all of them for T' <= 160): between two workgroup
    // barriers every wave walks its key blocks on its own, so the waves drift apart and hide each
    // other's L2 / LDS latencies instead of marching in lockstep.
    for (int jc = 0; jc < n_kblocks; jc += KB_CHUNK) {
        const int nb = n_kblocks - jc < KB_CHUNK ? n_kblocks - jc : KB_CHUNK;
        if (jc > 0) __syncthreads();          // previous chunk fully consumed
        for (int sl = 0; sl < nb; ++sl) stage_kv((jc + sl) * 32, sl);
        __syncthreads();

        for (int sl = 0; sl < nb; ++sl) {
            const int j0 = (jc + sl) * 32;
            const char* ks_t = Ks + sl * K_BYTES;
            const char* vts_t = Vts + sl * VT_BYTES;

            // ---- S^T = K . (Q+u)^T   (8 MFMAs)
            f32x16_t s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks_t + il * KROW + (2 * ks + hh) * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qu[ks], s, 0, 0, 0);
            }
            // ---- upper half of BD^T for this key block: relative positions n0+32 .. n0+63   (8 MFMAs)
            const int n0 = j0 - i0 - 31 + T - 1;
            const f32x16_t bd_hi = bd_block(n0 + 32);

            // ---- skew through the per-wave scratch: scr_f[n_local][query]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                scr_f[rowmap(r, hh) * 32 + il] = bd_lo[r];
                scr_f[(32 + rowmap(r, hh)) * 32 + il] = bd_hi[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            float pr[16];
            float mblk = NEG;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = rowmap(r, hh);
                const int j = j0 + jl;
                const float bdv = scr_f[(jl - il + 31) * 32 + il];
                bool ok = q_valid && j < len;
                if (p.att_left >= 0 || p.att_right >= 0) {
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
            bd_lo = bd_hi;
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
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
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
            for (int db = 0; db < 4; ++db) {
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

    // ---- write ctx: lane = (query il, half hh); reg r of block db -> d = db*32 + rowmap(r, hh)
    if (qi < T) {
        const float inv = (q_valid && l_run > 0.0f) ? 1.0f / l_run : 0.0f;
        uint16_t* op = p.out + ((size_t)b * T + qi) * d + h * HD;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<u16x4_t*>(op + db * 32 + 8 * g + 4 * hh) =
                    pack_bf16x4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
    }
}

}  // namespace

int rs_launch_attention(rs_ctx* ctx, const uint16_t* qkv, const uint16_t* pos, const float* bias_u,
                        const float* bias_v, const int32_t* lens, int B, int T, uint16_t* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return RS_OK;
    const rs_dims& dm = ctx->d;
    if (dm.d_model / dm.n_heads != HD) return rs_fail(ctx, RS_EINVAL, "attention: head_dim must be %d", HD);
    AttnParams p;
    p.qkv = qkv; p.pos = pos; p.bias_u = bias_u; p.bias_v = bias_v; p.lens = lens; p.out = out;
    p.T = T; p.d_model = dm.d_model; p.att_left = dm.att_left; p.att_right = dm.att_right; p.n_global = dm.n_global;
    p.scale = 1.0f / sqrtf((float)HD);
    const int qblocks = (T + 31) / 32;
    // at most 6 query blocks per workgroup: 5 * (8704 + 10240) + 6 * 8704 = 147 KB of the 160 KB LDS
    const int nw = qblocks < 6 ? qblocks : 6;
    const dim3 grid((qblocks + nw - 1) / nw, dm.n_heads, B), block(64 * nw);
    const size_t lds = (size_t)KB_CHUNK * (K_BYTES + VT_BYTES) + (size_t)nw * SCR_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)relpos_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                KB_CHUNK * (K_BYTES + VT_BYTES) + 6 * SCR_BYTES) != hipSuccess)
            return rs_fail(ctx, RS_EHIP, "attention: cannot reserve LDS");
        attr_set = true;
    }
    // algorithmic: ac + bd + pv = 3 * 2*T*T*128 per (b,h)
    const double flops = (double)B * dm.n_heads * 3.0 * 2.0 * T * (double)T * HD;
    const double bytes = (double)B * T * dm.d_model * 2.0 * 4.0;
    rs_prof_begin(ctx, RS_PROF_ATTN, s, flops, bytes);
    hipLaunchKernelGGL(relpos_attention_kernel, grid, block, lds, s, p);
    rs_prof_end(ctx, RS_PROF_ATTN, s);
    RS_CHECK_LAUNCH(ctx, "relpos_attention");
    return RS_OK;
}

// k_rnnt.hip — RNN-T prediction network, joint network and batched greedy decode loop
// ([UPSTREAM] RNNTDecoder.predict, RNNTJoint.joint_after_projection,
// GreedyBatchedRNNTInfer with max_symbols).
//
// Everything here is EXACT float32 with a fixed accumulation order, so that greedy token ids can
// be compared bit-for-bit with oracle/rnnt_greedy.c:
//   * dot products run on v_mfma_f32_16x16x4_f32 (exact f32 fma chain, guide §3); every output is
//     S = 4 partial chains over contiguous K slices, slice s accumulating from 0 in the order
//         for u in 16-blocks: for e in 0..3: for kk in 0..3:  k = base_s + 16u + 4kk + e
//     (what a lane's float4 loads feed the MFMA), combined as ((p0 + p1) + p2) + p3, then + bias.
//   * exp / sigmoid / tanh are the polynomial below (only +,-,*,/ and fmaf), not libm.
// This translation unit is compiled with -ffp-contract=off.
//
// One decode step = 5 launches:  joint+argmax partials -> finalize (state machine, emits tokens,
// builds the list of rows that emitted) -> LSTM layer 0 -> LSTM layer 1 -> prediction projection
// (+ state commit).  Only rows that emitted a non-blank run the LSTM (compacted row list).
#include "rs_common.h"

namespace {

constexpr int SPLITK = 4;

// ---- exact-order math (mirrored verbatim in oracle/rnnt_greedy.c) --------------------------------
__device__ __forceinline__ float rs_expf(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = fmaf(p, r2, r) + 1.0f;
    const int ni = (int)n;
    return y * __uint_as_float((unsigned)(ni + 127) << 23);
}
__device__ __forceinline__ float rs_sigmoidf(float x) { return 1.0f / (1.0f + rs_expf(-x)); }
__device__ __forceinline__ float rs_tanhf(float x) { return 1.0f - 2.0f / (rs_expf(2.0f * x) + 1.0f); }

struct DecodeState {
    // per-row state (B rows)
    float* h;        // [L][B][H] committed hidden
    float* c;        // [L][B][H] committed cell
    float* h_tmp;    // [L][B][H] this step's new hidden (active rows only)
    float* c_tmp;    // [L][B][H]
    float* g;        // [B][J]   prediction-net output after joint.pred
    int32_t* tcur;   // [B] encoder frame pointer
    int32_t* sym;    // [B] symbols emitted at the current frame
    int32_t* token;  // [B] last emitted token (LSTM input)
    int32_t* done;   // [B]
    int32_t* act;    // [B] compacted list of rows that emitted this step
    int32_t* counters;  // [0]=n_act [1]=n_unfinished [2]=overflow
    float* pmax;     // [B][n_ctiles] partial max
    int32_t* pidx;   // [B][n_ctiles] partial argmax
};

// ------------------------------------------------------------------------------------------------
__global__ void rnnt_init_kernel(DecodeState st, const int32_t* __restrict__ enc_lens, int B, int L, int H, int J,
                                 int blank, int32_t* __restrict__ n_ids) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) { st.counters[0] = B; st.counters[1] = 0; st.counters[2] = 0; }
    if (b >= B) return;
    st.tcur[b] = 0; st.sym[b] = 0; st.token[b] = blank; st.act[b] = b;
    st.done[b] = enc_lens[b] <= 0 ? 1 : 0;
    n_ids[b] = 0;
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < H; ++k) { st.h[((size_t)l * B + b) * H + k] = 0.0f; st.c[((size_t)l * B + b) * H + k] = 0.0f; }
    for (int k = 0; k < J; ++k) st.g[(size_t)b * J + k] = 0.0f;
}

// ---- LSTM layer: gates = [x ; h_prev] . [W_ih | W_hh]^T + b, cell update ---------------------------
// grid (H/16 unit tiles, B/16 row tiles); block 256 = 4 waves = the 4 K slices of one
// [16 rows] x [16 units x 4 gates] output tile.
__global__ __launch_bounds__(256) void rnnt_lstm_kernel(DecodeState st, int layer, int B, int H,
                                                        const float* __restrict__ embed,
                                                        const float* __restrict__ W /* [4H][2H] */,
                                                        const float* __restrict__ bias /* [4H] = b_ih + b_hh */) {
    __shared__ float part[SPLITK][4][16][17];
    __shared__ int rows_s[16];
    const int n_act = st.counters[0];
    const int rt = blockIdx.y, ut = blockIdx.x;
    if (rt * 16 >= n_act) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) {
        const int i = rt * 16 + tid;
        rows_s[tid] = st.act[i < n_act ? i : n_act - 1];
    }
    __syncthreads();
    const int li = lane & 15, kk = lane >> 4;
    const int row = rows_s[li];
    const int K = 2 * H, kslice = K / SPLITK;
    const float* xsrc = layer == 0 ? embed + (size_t)st.token[row] * H
                                   : st.h_tmp + ((size_t)(layer - 1) * B + row) * H;
    const float* hsrc = st.h + ((size_t)layer * B + row) * H;
    const float* wrow[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) wrow[gt] = W + (size_t)(gt * H + ut * 16 + li) * K;
    f32x4_t acc[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) acc[gt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int kbeg = wave * kslice;
    for (int k0 = kbeg; k0 < kbeg + kslice; k0 += 16) {
        const int k = k0 + 4 * kk;
        const float4 a = (k < H) ? *reinterpret_cast<const float4*>(xsrc + k)
                                 : *reinterpret_cast<const float4*>(hsrc + (k - H));
        float4 w[4];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) w[gt] = *reinterpret_cast<const float4*>(wrow[gt] + k);
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) {
            acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[gt].x, acc[gt], 0, 0, 0);
            acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[gt].y, acc[gt], 0, 0, 0);
            acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[gt].z, acc[gt], 0, 0, 0);
            acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[gt].w, acc[gt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][gt][4 * kk + r][li] = acc[gt][r];
    __syncthreads();
    // epilogue: thread = (row i, unit j)
    const int i = tid >> 4, j = tid & 15;
    if (rt * 16 + i >= n_act) return;
    const int brow = rows_s[i];
    const int unit = ut * 16 + j;
    float z[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
        const float s = ((part[0][gt][i][j] + part[1][gt][i][j]) + part[2][gt][i][j]) + part[3][gt][i][j];
        z[gt] = s + bias[gt * H + unit];
    }
    const float ig = rs_sigmoidf(z[0]), fg = rs_sigmoidf(z[1]), gg = rs_tanhf(z[2]), og = rs_sigmoidf(z[3]);
    const size_t o = ((size_t)layer * B + brow) * H + unit;
    const float cn = fmaf(fg, st.c[o], ig * gg);
    st.c_tmp[o] = cn;
    st.h_tmp[o] = og * rs_tanhf(cn);
}

// ---- prediction projection g = W_p . h_top + b_p for the active rows, plus state commit -----------
// grid (J/16, B/16)
__global__ __launch_bounds__(256) void rnnt_pred_kernel(DecodeState st, int B, int L, int H, int J,
                                                        const float* __restrict__ Wp /* [J][H] */,
                                                        const float* __restrict__ bp) {
    __shared__ float part[SPLITK][16][17];
    __shared__ int rows_s[16];
    const int n_act = st.counters[0];
    const int rt = blockIdx.y, jt = blockIdx.x;
    if (rt * 16 >= n_act) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) {
        const int i = rt * 16 + tid;
        rows_s[tid] = st.act[i < n_act ? i : n_act - 1];
    }
    __syncthreads();
    const int li = lane & 15, kk = lane >> 4;
    const int row = rows_s[li];
    const float* hsrc = st.h_tmp + ((size_t)(L - 1) * B + row) * H;
    const float* wr = Wp + (size_t)(jt * 16 + li) * H;
    const int kslice = H / SPLITK, kbeg = wave * kslice;
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kbeg + kslice; k0 += 16) {
        const int k = k0 + 4 * kk;
        const float4 a = *reinterpret_cast<const float4*>(hsrc + k);
        const float4 w = *reinterpret_cast<const float4*>(wr + k);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][4 * kk + r][li] = acc[r];
    __syncthreads();
    const int i = tid >> 4, j = tid & 15;
    if (rt * 16 + i >= n_act) return;
    const int brow = rows_s[i];
    const float s = ((part[0][i][j] + part[1][i][j]) + part[2][i][j]) + part[3][i][j];
    st.g[(size_t)brow * J + jt * 16 + j] = s + bp[jt * 16 + j];
    // commit this row's new LSTM state (each column tile copies its share of the H units)
    const int ntile = gridDim.x;
    for (int l = 0; l < L; ++l)
        for (int u = jt * 16 + j; u < H; u += ntile * 16) {
            const size_t o = ((size_t)l * B + brow) * H + u;
            st.h[o] = st.h_tmp[o];
            st.c[o] = st.c_tmp[o];
        }
}

// ---- joint: logits = W_o . relu(f[b][t_b] + g[b]) + b_o, per-tile argmax ----------------------------
// grid (ceil(V/16), B/16)
__global__ __launch_bounds__(256) void rnnt_joint_kernel(DecodeState st, const float* __restrict__ f /* [B][Tp][J] */,
                                                         int B, int Tp, int J, int V,
                                                         const float* __restrict__ Wo /* [V][J] */,
                                                         const float* __restrict__ bo, int n_ctiles) {
    __shared__ float part[SPLITK][16][17];
    __shared__ int alive_s;
    const int rt = blockIdx.y, ct = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (rt == 0 && ct == 0 && tid == 0) { st.counters[0] = 0; st.counters[1] = 0; }  // consumed by finalize
    if (tid == 0) alive_s = 0;
    __syncthreads();
    if (tid < 16) {
        const int b = rt * 16 + tid;
        if (b < B && !st.done[b]) alive_s = 1;
    }
    __syncthreads();
    if (!alive_s) return;
    const int li = lane & 15, kk = lane >> 4;
    int brow = rt * 16 + li;
    brow = brow < B ? brow : B - 1;
    int t = st.tcur[brow];
    t = t < Tp ? t : Tp - 1;
    const float* fsrc = f + ((size_t)brow * Tp + t) * J;
    const float* gsrc = st.g + (size_t)brow * J;
    int vrow = ct * 16 + li;
    vrow = vrow < V ? vrow : V - 1;
    const float* wr = Wo + (size_t)vrow * J;
    const int kslice = J / SPLITK, kbeg = wave * kslice;
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kbeg + kslice; k0 += 16) {
        const int k = k0 + 4 * kk;
        const float4 fv = *reinterpret_cast<const float4*>(fsrc + k);
        const float4 gv = *reinterpret_cast<const float4*>(gsrc + k);
        const float4 w = *reinterpret_cast<const float4*>(wr + k);
        const float a0 = fmaxf(fv.x + gv.x, 0.0f), a1 = fmaxf(fv.y + gv.y, 0.0f);
        const float a2 = fmaxf(fv.z + gv.z, 0.0f), a3 = fmaxf(fv.w + gv.w, 0.0f);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][4 * kk + r][li] = acc[r];
    __syncthreads();
    // thread = (row i, column j): combine slices, add bias, argmax over the 16 columns of the tile
    const int i = tid >> 4, j = tid & 15;
    const int v = ct * 16 + j;
    float val = ((part[0][i][j] + part[1][i][j]) + part[2][i][j]) + part[3][i][j];
    val = v < V ? val + bo[v] : -INFINITY;
    int idx = v;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {  // 16 consecutive lanes hold one row
        const float ov = __shfl_xor(val, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > val || (ov == val && oi < idx)) { val = ov; idx = oi; }
    }
    const int b = rt * 16 + i;
    if (j == 0 && b < B) {
        st.pmax[(size_t)b * n_ctiles + ct] = val;
        st.pidx[(size_t)b * n_ctiles + ct] = idx;
    }
}

// ---- finalize: full argmax + greedy state machine; one wave per row ------------------------------
__global__ __launch_bounds__(256) void rnnt_finalize_kernel(DecodeState st, const int32_t* __restrict__ enc_lens,
                                                            int B, int n_ctiles, int blank, int max_symbols,
                                                            int u_max, int32_t* __restrict__ ids,
                                                            int32_t* __restrict__ frames, int32_t* __restrict__ n_ids) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B || st.done[b]) return;
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
    if (idx == blank) {
        t += 1; sy = 0;
    } else {
        const int n = n_ids[b];
        if (n < u_max) { ids[(size_t)b * u_max + n] = idx; frames[(size_t)b * u_max + n] = t; n_ids[b] = n + 1; }
        else st.counters[2] = 1;
        st.token[b] = idx;
        emitted = true;
        sy += 1;
        if (sy >= max_symbols) { t += 1; sy = 0; }
    }
    st.tcur[b] = t; st.sym[b] = sy;
    if (t >= enc_lens[b]) {
        st.done[b] = 1;
    } else {
        atomicAdd(&st.counters[1], 1);
        if (emitted) { const int pos = atomicAdd(&st.counters[0], 1); st.act[pos] = b; }
    }
}

}  // namespace

// --------------------------------------------------------------------------------------------------
size_t rs_rnnt_workspace_bytes(const rs_ctx* ctx, int B) {
    const rs_dims& d = ctx->d;
    const int L = d.pred_layers, H = d.pred_hidden, J = d.joint_hidden;
    const int nct = (d.n_logits + 15) / 16;
    size_t n = 0;
    n += 4 * rs_align((size_t)L * B * H * 4);
    n += rs_align((size_t)B * J * 4);
    n += 5 * rs_align((size_t)B * 4);
    n += rs_align(64);
    n += 2 * rs_align((size_t)B * nct * 4);
    return n + 1024;
}

int rs_rnnt_greedy_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int u_max,
                        int32_t* ids, int32_t* frames, int32_t* n_ids, void* workspace, size_t workspace_bytes,
                        hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int L = d.pred_layers, H = d.pred_hidden, J = d.joint_hidden, V = d.n_logits;
    if (B <= 0) return RS_OK;
    if (H % 64 || J % 64) return rs_fail(ctx, RS_EINVAL, "rnnt: pred_hidden/joint_hidden must be multiples of 64");
    if (L < 1 || L > 4) return rs_fail(ctx, RS_EINVAL, "rnnt: 1..4 LSTM layers supported");
    if (workspace_bytes < rs_rnnt_workspace_bytes(ctx, B)) return rs_fail(ctx, RS_EWORKSPACE, "rnnt: workspace too small");
    const int nct = (V + 15) / 16;
    char* w = reinterpret_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* p = w; w += rs_align(bytes); return p; };
    DecodeState st;
    st.h = (float*)take((size_t)L * B * H * 4); st.c = (float*)take((size_t)L * B * H * 4);
    st.h_tmp = (float*)take((size_t)L * B * H * 4); st.c_tmp = (float*)take((size_t)L * B * H * 4);
    st.g = (float*)take((size_t)B * J * 4);
    st.tcur = (int32_t*)take((size_t)B * 4); st.sym = (int32_t*)take((size_t)B * 4);
    st.token = (int32_t*)take((size_t)B * 4); st.done = (int32_t*)take((size_t)B * 4);
    st.act = (int32_t*)take((size_t)B * 4);
    st.counters = (int32_t*)take(64);
    st.pmax = (float*)take((size_t)B * nct * 4); st.pidx = (int32_t*)take((size_t)B * nct * 4);

    const int rtiles = (B + 15) / 16;
    auto lstm_and_pred = [&]() -> int {
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL(rnnt_lstm_kernel, dim3(H / 16, rtiles), dim3(256), 0, s, st, l, B, H, ctx->embed,
                               ctx->lstm_w[l], ctx->lstm_b[l]);
        hipLaunchKernelGGL(rnnt_pred_kernel, dim3(J / 16, rtiles), dim3(256), 0, s, st, B, L, H, J, ctx->jpred_w,
                           ctx->jpred_b);
        return RS_OK;
    };

    rs_prof_begin(ctx, RS_PROF_DECODE, s, 0.0, 0.0);
    hipLaunchKernelGGL(rnnt_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, st, enc_lens, B, L, H, J, d.blank_id, n_ids);
    lstm_and_pred();  // SOS: blank token, zero state
    RS_CHECK_LAUNCH(ctx, "rnnt init");

    const int max_steps = tp_max + (u_max < tp_max * d.max_symbols ? u_max : tp_max * d.max_symbols) + 1;
    const int CHUNK = 16;
    int32_t host_counters[4] = {0, 0, 0, 0};
    int steps = 0;
    bool finished = false;
    while (!finished && steps < max_steps) {
        for (int c = 0; c < CHUNK; ++c, ++steps) {
            hipLaunchKernelGGL(rnnt_joint_kernel, dim3(nct, rtiles), dim3(256), 0, s, st, joint_enc, B, tp_max, J, V,
                               ctx->jout_w, ctx->jout_b, nct);
            hipLaunchKernelGGL(rnnt_finalize_kernel, dim3((B + 3) / 4), dim3(256), 0, s, st, enc_lens, B, nct,
                               d.blank_id, d.max_symbols, u_max, ids, frames, n_ids);
            lstm_and_pred();
        }
        RS_CHECK_LAUNCH(ctx, "rnnt step");
        RS_HIP(ctx, hipMemcpyAsync(host_counters, st.counters, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        RS_HIP(ctx, hipStreamSynchronize(s));
        finished = host_counters[1] == 0;
    }
    rs_prof_end(ctx, RS_PROF_DECODE, s);
    if (host_counters[2]) return rs_fail(ctx, RS_EOVERFLOW, "rnnt: an utterance emitted more than u_max=%d tokens", u_max);
    if (!finished) return rs_fail(ctx, RS_ESTATE, "rnnt: decode did not finish in %d steps", max_steps);
    return RS_OK;
}

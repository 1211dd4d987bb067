// k_rnnt_common.h — pieces shared by the decode translation units (k_rnnt.hip: greedy, k_rnnt_alsd.hip: beam search).
// Both are compiled with -ffp-contract=off.
#pragma once
#include "rs_common.h"

namespace {

constexpr int SPLITK_LSTM = 16;  // K slices of the LSTM gate products
constexpr int SPLITK_TILE = 8;   // K slices of the joint / prediction projections

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

// Activation rows are gathered by index, so a lane-per-row load (what the MFMA A operand wants:
// lane = row + 16*kk) would be 64 separate 16-byte requests per instruction and the texture
// addresser, not the MFMA pipe, would set the pace.  Instead lane l loads (row l>>2, 16-byte chunk
// l&3) — four adjacent lanes cover one contiguous 64-byte run — and one ds_bpermute per dword moves
// the data to the MFMA layout: lane m = (li, kk) takes it from lane 4*li + kk.
__device__ __forceinline__ float4 to_mfma_a_layout(float4 v, int src_lane_bytes) {
    float4 r;
    r.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_bytes, __float_as_int(v.x)));
    r.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_bytes, __float_as_int(v.y)));
    r.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_bytes, __float_as_int(v.z)));
    r.w = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane_bytes, __float_as_int(v.w)));
    return r;
}

struct DecodeState {
    // per-row state (B rows)
    float* h;        // [L][B][H] committed hidden
    float* c;        // [L][B][H] committed cell
    float* h_tmp;    // [L][B][H] this step's new hidden (rows that emitted only)
    float* c_tmp;    // [L][B][H]
    float* g;        // [B][J]   prediction-net output after joint.pred
    int32_t* tcur;   // [B] encoder frame pointer
    int32_t* sym;    // [B] symbols emitted at the current frame
    int32_t* token;  // [B] last emitted token (LSTM input)
    int32_t* act;    // [B] rows that emitted a non-blank this step (LSTM work list)
    int32_t* alive;  // [2][B] rows still decoding; list (s&1) is read by step s, (s+1)&1 is built by it
    int32_t* counters;  // [0]=n_act [1]=overflow flag [2],[3]=n_alive of list 0 / 1
    float* pmax;     // [B][n_ctiles] partial max
    int32_t* pidx;   // [B][n_ctiles] partial argmax
    // screened joint (see rnnt_prep_kernel / rnnt_verify_kernel)
    uint16_t* a16;   // [B][J] bf16 relu(f + g) of alive slot i
    float* anorm;    // [B]    ||relu(f + g)||_2 of alive slot i (rounded up)
    float* zapprox;  // [B][Vpad] approximate logits of alive slot i (bf16 MFMA GEMM, f32 accumulate, + bias)
    const long long* g_off;   // rnnt_tile_kernel<3> only: [rows] offset (floats, from `g`) of row r's joint.pred vector
    const float* a_pre;       // rnnt_tile_kernel<4> only: [rows][J] act(f + g) of row r
    int joint_act;   // 0: relu(f + g) (NeMo RNNTJoint); 1: tanh(f + g) (ESPnet JointNetwork) — exact-tile kernels only
    // Zipformer family (stateless decoder over the last two tokens; greedy search only): the token before `token`, and the id
    // greedy search treats like blank besides the blank itself (sherpa-onnx: "<unk>" is not emitted); nullptr / -1 elsewhere
    int32_t* token2 = nullptr;   // [B]
    int unk = -1;
};

}  // namespace

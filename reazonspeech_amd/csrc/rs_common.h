// rs_common.h — internal declarations shared by the HIP translation units of librs_asr.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rs_asr.h"

// ----------------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short v) {
    return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even (same as torch .to(bfloat16)); lowers to v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ u16x4_t pack_bf16x4(float a, float b, float c, float d) {
    const f32x4_t v = {a, b, c, d};
    return __builtin_bit_cast(u16x4_t, __builtin_convertvector(v, bf16x4_t));
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// fast forms for the bf16 encoder path (v_exp_f32 + v_rcp_f32, ~1 ulp each; outputs are rounded to
// bf16 right after).  The exact-f32 decode path has its own polynomial versions in k_rnnt.hip.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// icefall's Swoosh activations (scaling.py; the Zipformer family): log(1 + exp(x - o)) - 0.08 x - c, softplus in its stable form
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x))); }
__device__ __forceinline__ float swoosh_l_f(float x) { return softplus_f(x - 4.0f) - 0.08f * x - 0.035f; }
__device__ __forceinline__ float swoosh_r_f(float x) { return softplus_f(x - 1.0f) - 0.08f * x - 0.313261687f; }
// the same in IEEE arithmetic (float32 parity mode): torch's F.softplus (log1p(exp(x)), identity above the threshold 20)
__device__ __forceinline__ float softplus_exact(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float swoosh_l_exact(float x) { return softplus_exact(x - 4.0f) - 0.08f * x - 0.035f; }
__device__ __forceinline__ float swoosh_r_exact(float x) { return softplus_exact(x - 1.0f) - 0.08f * x - 0.313261687f; }

// ----------------------------------------------------------------------------------------
// host side: context
// ----------------------------------------------------------------------------------------
struct rs_prof_launch { int32_t M, N, K, flags; double flops; float ms; };
struct rs_prof_slot {
    std::vector<hipEvent_t> ev;  // pairs (start, stop)
    std::vector<rs_prof_launch> detail;   // one record per bracketed launch since the last reset (shape tag + its time once read)
    size_t read = 0;             // records of `detail` whose events were already folded into ms_acc
    size_t used = 0;
    double flops = 0, bytes = 0;
    double ms_acc = 0;
    int64_t launches = 0;
};

struct rs_layer_w {
    const float *ln_ff1_g, *ln_ff1_b, *ln_att_g, *ln_att_b, *ln_conv_g, *ln_conv_b, *ln_ff2_g, *ln_ff2_b,
        *ln_out_g, *ln_out_b;
    const uint16_t *ff1_w1, *ff1_w2, *ff2_w1, *ff2_w2, *qkv_w, *out_w, *pos_w, *pw1_w, *pw2_w;
    const float *ff1_b1, *ff1_b2, *ff2_b1, *ff2_b2, *qkv_b, *out_b, *bias_u, *bias_v, *pw1_b, *pw2_b;
    const float *dw_w, *dw_b;
    const uint16_t* pos_proj;   // optional "L{i}.att.pos_proj": pos.table @ pos_w^T, bf16 [2*Tcap-1][d]; nullptr = project per call
};

// float32 parity mode (k_f32.hip): the dense weights once more, unrounded.  Norm weights, biases, depthwise taps and
// the attention position biases are float32 in the throughput mode already and are shared.
struct rs_layer_w32 {
    const float *ff1_w1, *ff1_w2, *ff2_w1, *ff2_w2, *qkv_w, *out_w, *pos_w, *pw1_w, *pw2_w;
    const float* pw1_b;         // pointwise_conv1 bias in NeMo's own row order (values | gates); the bf16 copy is interleaved
};
struct rs_f32_weights {
    const float* sub_pw_w[4] = {};
    const float* sub_conv1_w = nullptr;  // ESPnet family: "sub.conv1.w.f32" [C][9C]
    const float* ctc_w = nullptr;        // ESPnet family: "ctc.w.f32" [ctc rows padded to 4][d]
    const float* sub_out_w = nullptr;
    std::vector<rs_layer_w32> layers;
    const float* jenc_w = nullptr;
    const float* pos_table = nullptr;   // "pos.table.f32": f32 [2*Tcap-1][d]
    size_t pos_table_bytes = 0;
};

struct rs_k2;                       // the Zipformer family's resolved weights and dimensions (k_zipformer.hip)
struct rs_avsr;                     // the AV-HuBERT family's (k_avsr.hip)

struct rs_ctx {
    int device = 0;
    rs_k2* k2 = nullptr;            // non-null: a context made by rs_k2_create (reazonspeech.k2.asr); the stage entry points dispatch on it
    void (*k2_free)(rs_k2*) = nullptr;
    rs_avsr* avsr = nullptr;        // non-null: a context made by rs_avsr_create (reazonspeech.avsr); only the rs_avsr_* stage entry points apply
    void (*avsr_free)(rs_avsr*) = nullptr;
    rs_dims d{};
    int head_dim = 0, sub_freq = 0;
    bool finalized = false;
    std::string err;
    std::unordered_map<std::string, std::pair<const void*, size_t>> tensors;
    // resolved weights
    const float *fe_window = nullptr, *fe_fb_w = nullptr, *fe_twiddle = nullptr;
    const int32_t* fe_fb_idx = nullptr;
    const float *sub_conv0_w = nullptr, *sub_conv0_b = nullptr;
    const float *sub_dw_w[4] = {}, *sub_dw_b[4] = {};
    const uint16_t* sub_pw_w[4] = {};
    const float* sub_pw_b[4] = {};
    const uint16_t* sub_out_w = nullptr;
    const float* sub_out_b = nullptr;
    std::vector<rs_layer_w> layers;
    // ESPnet family (rs_dims.frontend_kind / sub_kind / final_norm / ctc_vocab)
    const float *fe_mvn_mean = nullptr, *fe_mvn_istd = nullptr;
    const uint16_t* sub_conv1_w = nullptr;
    const float* sub_conv1_b = nullptr;
    const float *final_norm_g = nullptr, *final_norm_b = nullptr;
    const uint16_t* ctc_w = nullptr;
    const float* ctc_b = nullptr;
    float *ctc_probs = nullptr, *ctc_blank = nullptr;     // rs_encoder_set_ctc_out
    const uint16_t* jenc_w = nullptr;
    const float* jenc_b = nullptr;
    const float* lstm_w4[8] = {};   // optional "pred.lstm{l}.w4": fragment-major with rows permuted to (unit group, gate, unit)
    const float* k2_conv_w = nullptr;   // stateless decoder (Zipformer family): grouped Conv1d over the last two tokens, f32 [D][4][2]
    const float *embed = nullptr, *lstm_w[8] = {}, *lstm_b[8] = {}, *jpred_w = nullptr, *jpred_b = nullptr,
                *jout_w = nullptr, *jout_b = nullptr;
    // screened joint (optional tensors joint.out.w16 / .wrm / .bpad / .wmax; k_rnnt.hip)
    const uint16_t* jout_w16 = nullptr;
    const float *jout_wrm = nullptr, *jout_bpad = nullptr, *jout_wmax = nullptr;
    bool decode_screen = true;
    bool decode_narrow = true;      // narrow-tile LSTM / projection kernels (k_rnnt.hip)
    bool gemm_f32_x3 = false;       // rs_launch_gemm_f32 / rs_launch_conv3x3_f32 multiply with three bf16 terms per float32 product (k_f32.hip X3; avsr only)
    int k2_conv2_fused = -1;        // Zipformer conv2 with its patches gathered into LDS: -1 = $RS_K2_CONV2_FUSED (default 1); rs_set_option("k2_conv2_fused")
    int k2_cnx_fused = -1;          // Zipformer ConvNeXt pointwise pair as one kernel: -1 = $RS_K2_CNX_FUSED (default 1); rs_set_option("k2_cnx_fused")
    // position table cache: the caller registers "pos_table.<T>" tensors (bf16 [2T-1][d])
    // options (rs_set_option)
    int n_cus = 0;                  // compute units of the device (queried on first use)
    int defer_out_norm = 1;         // 1 = a layer's output norm is applied by the next layer's first residual GEMM (f32 rows not stored); bit-identical to 0
    int fuse_glu = 1;               // conv module: 1 = GLU in the pw1 GEMM epilogue (every batch size: one rounding point, batch-invariant);
                                    // 0 = plain pw1 product, GLU in the depthwise kernel ($RS_FUSE_GLU; A/B and layout tests)
    bool has_f32 = false;           // the "*.f32" tensors of the float32 parity mode are registered (all or none)
    int precision_f32 = 0;          // rs_set_option("precision_f32"): 1 = rs_encoder_forward runs k_f32.hip's float32 encoder
    rs_f32_weights f32;
    bool env_read = false;          // the $RS_* defaults were applied (once, by the first rs_finalize; rs_set_option wins afterwards)
    // parity taps (rs_encoder_set_taps): copies of the residual stream taken inside rs_encoder_forward
    float* tap_sub = nullptr;
    float* tap_layers = nullptr;
    std::vector<int> tap_ids;
    // k_gemm_bf16.hip, register-resident weight form ($RS_GEMM_BREG): fragment-major copies of registered GEMM weights (made on first
    // use, owned by the context) and a scratch for operands that are not registered tensors (re-shuffled on every launch)
    std::unordered_map<const void*, void*> wfm;
    void* wfm_scratch = nullptr;
    size_t wfm_scratch_bytes = 0;
    // profiling
    int prof_mask = 0;
    rs_prof_slot prof[8];
    int32_t prof_tag[4] = {0, 0, 0, 0};   // (M, N, K, flags) of the launch about to be bracketed (set by the GEMM launcher)
};

// Opt-in for more than 64 KiB of dynamic LDS, once per (device, kernel): the attribute is per device and the
// first launch of a kernel can come from any thread (the decode worker and the encoder thread both start
// kernels), so the bookkeeping is a mutex-guarded set keyed by the context's device — not a function-local flag.
int rs_ensure_dynamic_lds(rs_ctx* ctx, const void* func, int bytes);

int rs_fail(rs_ctx* ctx, int code, const char* fmt, ...);

#define RS_HIP(ctx, call)                                                              \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess)                                                          \
            return rs_fail((ctx), RS_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, \
                           hipGetErrorString(_e));                                     \
    } while (0)

#define RS_CHECK_LAUNCH(ctx, what)                                                            \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return rs_fail((ctx), RS_EHIP, "launch %s failed: %s", (what), hipGetErrorString(_e)); \
    } while (0)

// RAII-ish profiling bracket around kernel launches of one class
int rs_prof_class_index(int klass);
void rs_prof_begin(rs_ctx* ctx, int klass, hipStream_t s, double flops, double bytes);
void rs_prof_end(rs_ctx* ctx, int klass, hipStream_t s);

static inline size_t rs_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ----------------------------------------------------------------------------------------
// kernel launchers (one per .hip file); all return RS_OK / RS_E*
// ----------------------------------------------------------------------------------------
struct rs_gemm_args {
    const uint16_t* A; int lda;
    const uint16_t* W; int ldw;
    void* out; int ldc;
    int M, N, K, flags;
    const float* bias; float alpha;
    const float* residual;
    const int32_t* mask_lens; int mask_rows_per_step; int mask_steps;
    // residual = LayerNorm(residual operand; res_ln_g, res_ln_b) with per-row (mean, rstd) in res_ln_stats [M][2]
    // (nullptr: the residual operand is used as it is)
    const float* res_ln_stats; const float* res_ln_g; const float* res_ln_b;
    // residual GEMMs only: also store the result rounded to bf16 at out_bf16[m * ld_bf16 + n] (the A operand of the next GEMM:
    // the Zipformer family has no norm between a residual add and the next Linear, so the copy would otherwise be a pass of its own)
    uint16_t* out_bf16; int ld_bf16;
    // conv_C > 0: A is not a matrix but the channels-last input [Bc][conv_T1][conv_F1][conv_C] of a 3 x 3, stride-2, un-padded
    // convolution; row m = (b * conv_T2 + t2) * conv_F2 + f2 of the product is the patch of output pixel (t2, f2), K = 9 * conv_C
    // ordered (kernel row, kernel column, channel) — the matrix im2col3x3s2 would write, read in place: a kernel row of a patch
    // is 3 * conv_C contiguous elements of the input (lda is ignored; plain bf16 output with the row mask only)
    int conv_C, conv_T1, conv_F1, conv_T2, conv_F2;
};
int rs_launch_gemm(rs_ctx* ctx, const rs_gemm_args& a, hipStream_t s);
int rs_launch_layernorm(rs_ctx* ctx, const float* x, const float* g, const float* b, int M, int d, float eps,
                        uint16_t* out_bf16, float* out_f32, hipStream_t s);
// out_f32 == nullptr: the first norm's rows are not stored; `stats` [M][2] receives its (mean, rstd) per row instead
int rs_launch_layernorm2(rs_ctx* ctx, const float* x, const float* g1, const float* b1, const float* g2, const float* b2,
                         int M, int d, float eps, float* out_f32, uint16_t* out_bf16, float* stats, hipStream_t s);
int rs_launch_attention(rs_ctx* ctx, const uint16_t* qkv, const uint16_t* pos, const float* bias_u,
                        const float* bias_v, const int32_t* lens, int B, int T, uint16_t* out, hipStream_t s);
int rs_launch_glu_dwconv(rs_ctx* ctx, const uint16_t* x, int layout, const float* w, const float* b, const int32_t* lens,
                         int B, int T, int d, int k, uint16_t* out, hipStream_t s);
int rs_launch_frontend(rs_ctx* ctx, const float* audio, const int32_t* lens, int B, int audio_stride,
                       int pad_left, int pad_right, int t_max, float* feats, int32_t* n_frames, float* raw,
                       hipStream_t s);
int rs_launch_sub_conv0_dw1(rs_ctx* ctx, const float* feats, const int32_t* lens_stage, int B, int t_max, int T2,
                            int F2, uint16_t* out, hipStream_t s);
int rs_launch_sub_dw(rs_ctx* ctx, const uint16_t* in, const float* w, const float* b, const int32_t* lens_out,
                     int stage, int B, int t_in, int f_in, int t_out, int f_out, uint16_t* out, hipStream_t s);
// float32 parity mode (k_f32.hip, k_subsample.hip)
int rs_launch_gemm_f32(rs_ctx* ctx, const float* A, int lda, const float* W, int ldw, float* out, int ldc, int M, int N, int K,
                       int flags, const float* bias, float alpha, const float* residual, const int32_t* mask_lens,
                       int mask_rows_per_step, int mask_steps, hipStream_t s);
// the same product for M <= 128 rows (a decoder step's hypothesis rows): N / 16 workgroups, K cut over the four waves (k_f32.hip)
int rs_launch_conv3x3_f32(rs_ctx* ctx, const float* in, int n_img, int H, int Wd, int C, int OH, int OW, int stride, const float* W, int Cout,
                          const float* bn_scale, const float* bn_shift, const float* residual, const float* prelu, float* out, int one_by_one, hipStream_t s);
int rs_launch_gemm_f32_skinny(rs_ctx* ctx, const float* A, int lda, const float* W, int ldw, float* out, int ldc, int M, int N, int K, int flags,
                              const float* bias, const float* residual, hipStream_t s);
int rs_launch_attention_f32(rs_ctx* ctx, const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                            const int32_t* lens, int B, int T, float* out, hipStream_t s);
int rs_launch_glu_dwconv_f32(rs_ctx* ctx, const float* x, const float* w, const float* b, const int32_t* lens, int B, int T,
                             int d, int k, float* out, hipStream_t s);
size_t rs_encoder_f32_workspace_bytes(const rs_ctx* ctx, int B, int t_max);
int rs_encoder_forward_f32(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max, float* enc_out,
                           float* joint_enc, int32_t* enc_lens, void* workspace, size_t workspace_bytes, hipStream_t s);
int rs_launch_enc_lens(rs_ctx* ctx, const int32_t* n_frames, int B, int32_t* lens_out, hipStream_t s);
// ESPnet family (k_espnet.hip)
int rs_launch_sub2d_conv0(rs_ctx* ctx, const float* feats, const int32_t* lens1, int b0, int Bc, int t_max, int T1, int F1,
                          uint16_t* out, hipStream_t s);
int rs_launch_im2col3x3s2(rs_ctx* ctx, const uint16_t* in, int Bc, int T1, int F1, int T2, int F2, uint16_t* out, hipStream_t s);
int rs_launch_ctc_softmax(rs_ctx* ctx, float* logits, int M, int V, int ld, int blank, float* blank_out, hipStream_t s);
int rs_launch_sub2d_conv0_f32(rs_ctx* ctx, const float* feats, const int32_t* lens1, int b0, int Bc, int t_max, int T1, int F1,
                              float* out, hipStream_t s);
int rs_launch_im2col3x3s2_f32(rs_ctx* ctx, const float* in, int Bc, int T1, int F1, int T2, int F2, float* out, hipStream_t s);
// rows of the registered CTC head ("ctc.w" / "ctc.b") and row pitch of the posteriors: the vocabulary padded to a multiple of 4
static inline int rs_ctc_pad(int v) { return (v + 3) / 4 * 4; }
// Zipformer family (k_zipformer.hip): the entry points of rs_api.hip dispatch to these when ctx->k2 is set
int rs_avsr_finalize_impl(rs_ctx* ctx);
int rs_k2_finalize_impl(rs_ctx* ctx);
int rs_k2_unk_id(const rs_ctx* ctx);
size_t rs_k2_workspace_bytes_impl(const rs_ctx* ctx, int B, int t_max);
int rs_k2_enc_frames_impl(const rs_ctx* ctx, int n_feat);
int rs_k2_encoder_forward_impl(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max, float* enc_out, float* joint_enc,
                               int32_t* enc_lens, void* workspace, size_t workspace_bytes, hipStream_t s);
// depthwise Conv1d over time with a compile-time kernel size (7 / 15 / 31), frame mask, bias, activation (0 SiLU, 1 SwooshR):
// x bf16 [B*T][d] -> out bf16 [B*T][d]; d % 64 == 0 (k_layernorm.hip)
int rs_launch_dwconv_act(rs_ctx* ctx, const uint16_t* x, const float* w, const float* b, const int32_t* lens, int B, int T, int d, int k,
                         int act, uint16_t* out, hipStream_t s);
// output length of one subsampling conv (k 3, s 2): padding 1 (NeMo dw_striding) or none (ESPnet Conv2dSubsampling)
static inline int rs_conv_len(int n, int sub_kind) { return sub_kind ? (n >= 3 ? (n - 3) / 2 + 1 : 0) : (n > 0 ? (n + 2 - 3) / 2 + 1 : 0); }

// rs_api.hip — the C ABI of librs_asr.so (include/rs_asr.h): context, weight registry, workspace
// carving and the stage orchestrators that enqueue the kernels of k_*.hip on the caller's stream.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <set>
#include <utility>

#include "rs_common.h"

size_t rs_rnnt_workspace_bytes(const rs_ctx* ctx, int B);
size_t rs_rnnt_alsd_workspace_bytes_impl(const rs_ctx* ctx, int B, int beam, int cap);
size_t rs_rnnt_beam_workspace_bytes_impl(const rs_ctx* ctx, int B, int beam, int tp_max, int max_pops);
int rs_rnnt_beam_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam, int score_norm,
                      int max_pops, int out_cap, int32_t* ids, int32_t* frames, int32_t* n_ids, float* scores, int32_t* pops,
                      void* workspace, size_t workspace_bytes, hipStream_t s);
int rs_rnnt_alsd_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam, double ratio,
                      int abs_len, int score_norm, int merge, int out_cap, int32_t* ids, int32_t* steps, int32_t* n_ids,
                      float* scores, void* workspace, size_t workspace_bytes, hipStream_t s);
int rs_rnnt_greedy_impl(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int u_max,
                        int32_t* ids, int32_t* frames, int32_t* n_ids, void* workspace, size_t workspace_bytes,
                        hipStream_t s);

int rs_fail(rs_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int rs_ensure_dynamic_lds(rs_ctx* ctx, const void* func, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(ctx->device, func);
    if (done.count(key)) return RS_OK;
    // the attribute applies to the calling thread's CURRENT device: set it on the context's device whatever is current
    // (a caller may drive two GPUs from one thread), and leave the caller's current device as it was
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return rs_fail(ctx, RS_EHIP, "hipGetDevice failed");
    if (cur != ctx->device && hipSetDevice(ctx->device) != hipSuccess)
        return rs_fail(ctx, RS_EHIP, "hipSetDevice(%d) failed", ctx->device);
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (cur != ctx->device) hipSetDevice(cur);
    if (e != hipSuccess) return rs_fail(ctx, RS_EHIP, "cannot reserve %d bytes of dynamic LDS on device %d", bytes, ctx->device);
    done.insert(key);
    return RS_OK;
}

// ---- profiling ---------------------------------------------------------------------------------
int rs_prof_class_index(int klass) {
    int i = 0;
    while (klass > 1) { klass >>= 1; ++i; }
    return i;
}
void rs_prof_begin(rs_ctx* ctx, int klass, hipStream_t s, double flops, double bytes) {
    if (!(ctx->prof_mask & klass)) return;
    rs_prof_slot& p = ctx->prof[rs_prof_class_index(klass)];
    if (p.used + 2 > p.ev.size()) {
        const size_t old = p.ev.size();
        p.ev.resize(old + 512);
        for (size_t i = old; i < p.ev.size(); ++i) hipEventCreate(&p.ev[i]);
    }
    hipEventRecord(p.ev[p.used], s);
    p.flops += flops;
    p.bytes += bytes;
    p.launches += 1;
    if (p.detail.size() < (1u << 20)) p.detail.push_back(rs_prof_launch{ctx->prof_tag[0], ctx->prof_tag[1], ctx->prof_tag[2], ctx->prof_tag[3], flops, -1.0f});
    ctx->prof_tag[0] = ctx->prof_tag[1] = ctx->prof_tag[2] = ctx->prof_tag[3] = 0;
}
void rs_prof_end(rs_ctx* ctx, int klass, hipStream_t s) {
    if (!(ctx->prof_mask & klass)) return;
    rs_prof_slot& p = ctx->prof[rs_prof_class_index(klass)];
    hipEventRecord(p.ev[p.used + 1], s);
    p.used += 2;
}

extern "C" {

int rs_abi_version(void) { return RS_ABI_VERSION; }

int rs_create(rs_ctx** out, int device, const rs_dims* dims) {
    if (!out || !dims) return RS_EINVAL;
    *out = nullptr;
    rs_ctx* ctx = new (std::nothrow) rs_ctx();
    if (!ctx) return RS_EINVAL;
    ctx->device = device;
    ctx->d = *dims;
    const rs_dims& d = ctx->d;
    int rc = RS_OK;
    if (d.n_heads <= 0 || d.d_model % d.n_heads || (d.d_model / d.n_heads != 128 && d.d_model / d.n_heads != 64))
        rc = rs_fail(ctx, RS_EINVAL, "head_dim must be 128 or 64 (d_model %d, heads %d)", d.d_model, d.n_heads);
    else if (d.d_model % 256 || d.ff_dim % 64) rc = rs_fail(ctx, RS_EINVAL, "d_model %% 256, ff_dim %% 64 required");
    else if (d.sub_stages < 2 || d.sub_stages > 4) rc = rs_fail(ctx, RS_EINVAL, "2..4 subsampling stages supported");
    else if (d.n_layers < 1) rc = rs_fail(ctx, RS_EINVAL, "n_layers");
    else if (d.pred_layers < 1 || d.pred_layers > 4) rc = rs_fail(ctx, RS_EINVAL, "pred_layers");
    else if ((unsigned)d.frontend_kind > 1u /* 2 = kaldi fbank: rs_k2_create only */ || (unsigned)d.sub_kind > 1u || (unsigned)d.final_norm > 1u || (unsigned)d.joint_act > 1u)
        rc = rs_fail(ctx, RS_EINVAL, "model family switches must be 0 or 1");
    else if (d.sub_kind == 1 && (d.sub_stages != 2 || d.sub_channels % 64)) rc = rs_fail(ctx, RS_EINVAL, "Conv2dSubsampling: x4 (two stages), channels %% 64");
    else if (d.frontend_kind == 1 && d.preemph != 0.0f) rc = rs_fail(ctx, RS_EINVAL, "the ESPnet front-end has no pre-emphasis");
    else if (d.ctc_vocab < 0) rc = rs_fail(ctx, RS_EINVAL, "ctc_vocab < 0");
    if (rc != RS_OK) { *out = ctx; return rc; }  // caller can read rs_last_error, then rs_destroy
    ctx->head_dim = d.d_model / d.n_heads;
    int f = d.n_mels;
    for (int s = 0; s < d.sub_stages; ++s) f = rs_conv_len(f, d.sub_kind);
    ctx->sub_freq = f;
    if (hipSetDevice(device) != hipSuccess) { *out = ctx; return rs_fail(ctx, RS_EHIP, "hipSetDevice(%d) failed", device); }
    *out = ctx;
    return RS_OK;
}

void rs_destroy(rs_ctx* ctx) {
    if (!ctx) return;
    for (auto& p : ctx->prof)
        for (auto e : p.ev) hipEventDestroy(e);
    for (auto& kv : ctx->wfm) (void)hipFree(kv.second);
    if (ctx->wfm_scratch) (void)hipFree(ctx->wfm_scratch);
    if (ctx->k2 && ctx->k2_free) ctx->k2_free(ctx->k2);
    if (ctx->avsr && ctx->avsr_free) ctx->avsr_free(ctx->avsr);
    delete ctx;
}

const char* rs_last_error(const rs_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rs_set_tensor(rs_ctx* ctx, const char* name, const void* dev_ptr, size_t nbytes) {
    if (!ctx || !name || !dev_ptr) return RS_EINVAL;
    ctx->tensors[name] = {dev_ptr, nbytes};
    ctx->finalized = false;
    return RS_OK;
}

}  // extern "C"

namespace {

struct Resolver {
    rs_ctx* ctx;
    int rc = RS_OK;
    template <typename T>
    void get(const std::string& name, size_t elems, const T*& out) {
        if (rc != RS_OK) return;
        auto it = ctx->tensors.find(name);
        if (it == ctx->tensors.end()) { rc = rs_fail(ctx, RS_EMISSING, "weight tensor '%s' was not registered", name.c_str()); return; }
        if (it->second.second != elems * sizeof(T)) {
            rc = rs_fail(ctx, RS_EINVAL, "tensor '%s': expected %zu bytes, got %zu", name.c_str(), elems * sizeof(T),
                         it->second.second);
            return;
        }
        if ((uintptr_t)it->second.first & 15) { rc = rs_fail(ctx, RS_EINVAL, "tensor '%s' is not 16-byte aligned", name.c_str()); return; }
        out = reinterpret_cast<const T*>(it->second.first);
    }
};

}  // namespace

extern "C" {

int rs_finalize(rs_ctx* ctx) {
    if (!ctx) return RS_EINVAL;
    if (ctx->avsr) return rs_avsr_finalize_impl(ctx);
    if (ctx->k2) return rs_k2_finalize_impl(ctx);
    const rs_dims& d = ctx->d;
    Resolver r{ctx};
    const size_t C = d.sub_channels, dm = d.d_model, ff = d.ff_dim, H = d.pred_hidden, J = d.joint_hidden, V = d.n_logits;
    r.get("fe.window", (size_t)d.win_length, ctx->fe_window);
    r.get("fe.twiddle", (size_t)512, ctx->fe_twiddle);
    r.get("fe.fb_idx", (size_t)d.n_mels * 2, ctx->fe_fb_idx);
    r.get("fe.fb_w", (size_t)d.n_mels * 32, ctx->fe_fb_w);
    r.get("sub.conv0.w", 9 * C, ctx->sub_conv0_w);
    r.get("sub.conv0.b", C, ctx->sub_conv0_b);
    if (d.frontend_kind == 1) { r.get("fe.mvn_mean", (size_t)d.n_mels, ctx->fe_mvn_mean); r.get("fe.mvn_istd", (size_t)d.n_mels, ctx->fe_mvn_istd); }
    if (d.sub_kind == 1) { r.get("sub.conv1.w", C * 9 * C, ctx->sub_conv1_w); r.get("sub.conv1.b", C, ctx->sub_conv1_b); }
    if (d.final_norm) { r.get("final_norm.g", dm, ctx->final_norm_g); r.get("final_norm.b", dm, ctx->final_norm_b); }
    if (d.ctc_vocab > 0) { r.get("ctc.w", (size_t)rs_ctc_pad(d.ctc_vocab) * dm, ctx->ctc_w); r.get("ctc.b", (size_t)rs_ctc_pad(d.ctc_vocab), ctx->ctc_b); }
    for (int s = 1; s < d.sub_stages && d.sub_kind == 0; ++s) {
        const std::string p = "sub.dw" + std::to_string(s), q = "sub.pw" + std::to_string(s);
        r.get(p + ".w", 9 * C, ctx->sub_dw_w[s - 1]);
        r.get(p + ".b", C, ctx->sub_dw_b[s - 1]);
        r.get(q + ".w", C * C, ctx->sub_pw_w[s - 1]);
        r.get(q + ".b", C, ctx->sub_pw_b[s - 1]);
    }
    r.get("sub.out.w", dm * C * ctx->sub_freq, ctx->sub_out_w);
    r.get("sub.out.b", dm, ctx->sub_out_b);
    ctx->layers.assign(d.n_layers, rs_layer_w{});
    for (int i = 0; i < d.n_layers; ++i) {
        rs_layer_w& L = ctx->layers[i];
        const std::string p = "L" + std::to_string(i) + ".";
        r.get(p + "ln_ff1.g", dm, L.ln_ff1_g); r.get(p + "ln_ff1.b", dm, L.ln_ff1_b);
        r.get(p + "ff1.w1", ff * dm, L.ff1_w1); r.get(p + "ff1.b1", ff, L.ff1_b1);
        r.get(p + "ff1.w2", dm * ff, L.ff1_w2); r.get(p + "ff1.b2", dm, L.ff1_b2);
        r.get(p + "ln_att.g", dm, L.ln_att_g); r.get(p + "ln_att.b", dm, L.ln_att_b);
        r.get(p + "att.qkv.w", 3 * dm * dm, L.qkv_w); r.get(p + "att.qkv.b", 3 * dm, L.qkv_b);
        r.get(p + "att.out.w", dm * dm, L.out_w); r.get(p + "att.out.b", dm, L.out_b);
        r.get(p + "att.pos.w", dm * dm, L.pos_w);
        r.get(p + "att.bias_u", dm, L.bias_u); r.get(p + "att.bias_v", dm, L.bias_v);
        r.get(p + "ln_conv.g", dm, L.ln_conv_g); r.get(p + "ln_conv.b", dm, L.ln_conv_b);
        r.get(p + "conv.pw1.w", 2 * dm * dm, L.pw1_w); r.get(p + "conv.pw1.b", 2 * dm, L.pw1_b);
        r.get(p + "conv.dw.w", (size_t)d.conv_kernel * dm, L.dw_w); r.get(p + "conv.dw.b", dm, L.dw_b);
        r.get(p + "conv.pw2.w", dm * dm, L.pw2_w); r.get(p + "conv.pw2.b", dm, L.pw2_b);
        r.get(p + "ln_ff2.g", dm, L.ln_ff2_g); r.get(p + "ln_ff2.b", dm, L.ln_ff2_b);
        r.get(p + "ff2.w1", ff * dm, L.ff2_w1); r.get(p + "ff2.b1", ff, L.ff2_b1);
        r.get(p + "ff2.w2", dm * ff, L.ff2_w2); r.get(p + "ff2.b2", dm, L.ff2_b2);
        r.get(p + "ln_out.g", dm, L.ln_out_g); r.get(p + "ln_out.b", dm, L.ln_out_b);
    }
    r.get("joint.enc.w", J * dm, ctx->jenc_w); r.get("joint.enc.b", J, ctx->jenc_b);
    r.get("pred.embed", V * H, ctx->embed);
    for (int l = 0; l < d.pred_layers; ++l) {
        const std::string p = "pred.lstm" + std::to_string(l);
        r.get(p + ".w", 4 * H * 2 * H, ctx->lstm_w[l]);
        r.get(p + ".b", 4 * H, ctx->lstm_b[l]);
    }
    r.get("joint.pred.w", J * H, ctx->jpred_w); r.get("joint.pred.b", J, ctx->jpred_b);
    r.get("joint.out.w", ((V + 15) / 16 * 16) * J, ctx->jout_w); r.get("joint.out.b", V, ctx->jout_b);   // fragment-major, rows padded to 16
    if (r.rc != RS_OK) return r.rc;
    // optional: the screened joint's operands (bf16 [Vpad][J] row-major, f32 [V][J] row-major, bias padded with -3e38,
    // the largest row norm).  All four or none; without them the decode loop evaluates every column in exact f32.
    {
        const size_t Vpad = (V + 15) / 16 * 16;
        const bool any = ctx->tensors.count("joint.out.w16") || ctx->tensors.count("joint.out.wrm") ||
                         ctx->tensors.count("joint.out.bpad") || ctx->tensors.count("joint.out.wmax");
        ctx->jout_w16 = nullptr; ctx->jout_wrm = ctx->jout_bpad = ctx->jout_wmax = nullptr;
        if (any) {
            r.get("joint.out.w16", Vpad * J, ctx->jout_w16);
            r.get("joint.out.wrm", V * J, ctx->jout_wrm);
            r.get("joint.out.bpad", Vpad, ctx->jout_bpad);
            r.get("joint.out.wmax", (size_t)4, ctx->jout_wmax);
            if (r.rc != RS_OK) return r.rc;
        }
        for (int l = 0; l < d.pred_layers; ++l) {
            const std::string nm = "pred.lstm" + std::to_string(l) + ".w4";
            ctx->lstm_w4[l] = nullptr;
            if (ctx->tensors.count(nm)) { r.get(nm, 4 * H * 2 * H, ctx->lstm_w4[l]); if (r.rc != RS_OK) return r.rc; }
        }
        // $RS_* A/B knobs are defaults: applied ONCE per context (rs_finalize runs again after every rs_set_tensor, e.g.
        // when the position tables grow; a value chosen with rs_set_option must survive that)
        if (!ctx->env_read) {
            ctx->env_read = true;
            if (const char* e = getenv("RS_DECODE_SCREEN")) ctx->decode_screen = atoi(e) != 0;     // 0 = exact evaluation of every column
            if (const char* nw = getenv("RS_DECODE_NARROW")) ctx->decode_narrow = atoi(nw) != 0;   // 0 = the wide-tile kernels of round 1
            if (const char* fg = getenv("RS_FUSE_GLU")) ctx->fuse_glu = atoi(fg) != 0;             // 0 = GLU in the conv kernel
            if (const char* dn = getenv("RS_DEFER_OUT_NORM")) ctx->defer_out_norm = atoi(dn) != 0; // 0 = every output norm stores its f32 rows
        }
    }
    // optional: the float32 parity mode's dense weights ("<name>.f32", unrounded, the layouts of the bf16 tensors except
    // conv.pw1, which keeps NeMo's own row order) and its position table.  All or none.
    ctx->has_f32 = false;
    if (ctx->tensors.count("sub.out.w.f32")) {
        rs_f32_weights& w = ctx->f32;
        for (int s = 1; s < d.sub_stages && d.sub_kind == 0; ++s) r.get("sub.pw" + std::to_string(s) + ".w.f32", C * C, w.sub_pw_w[s - 1]);
        if (d.sub_kind == 1) r.get("sub.conv1.w.f32", C * 9 * C, w.sub_conv1_w);
        if (d.ctc_vocab > 0) r.get("ctc.w.f32", (size_t)rs_ctc_pad(d.ctc_vocab) * dm, w.ctc_w);
        r.get("sub.out.w.f32", dm * C * ctx->sub_freq, w.sub_out_w);
        w.layers.assign(d.n_layers, rs_layer_w32{});
        for (int i = 0; i < d.n_layers; ++i) {
            rs_layer_w32& L = w.layers[i];
            const std::string p = "L" + std::to_string(i) + ".";
            r.get(p + "ff1.w1.f32", ff * dm, L.ff1_w1); r.get(p + "ff1.w2.f32", dm * ff, L.ff1_w2);
            r.get(p + "ff2.w1.f32", ff * dm, L.ff2_w1); r.get(p + "ff2.w2.f32", dm * ff, L.ff2_w2);
            r.get(p + "att.qkv.w.f32", 3 * dm * dm, L.qkv_w); r.get(p + "att.out.w.f32", dm * dm, L.out_w);
            r.get(p + "att.pos.w.f32", dm * dm, L.pos_w);
            r.get(p + "conv.pw1.w.f32", 2 * dm * dm, L.pw1_w); r.get(p + "conv.pw1.b.f32", 2 * dm, L.pw1_b);
            r.get(p + "conv.pw2.w.f32", dm * dm, L.pw2_w);
        }
        r.get("joint.enc.w.f32", J * dm, w.jenc_w);
        if (r.rc != RS_OK) return r.rc;
        auto pt = ctx->tensors.find("pos.table.f32");
        if (pt == ctx->tensors.end()) return rs_fail(ctx, RS_EMISSING, "weight tensor 'pos.table.f32' was not registered");
        const size_t rowf = dm * sizeof(float);
        if (pt->second.second % rowf || !((pt->second.second / rowf) & 1) || ((uintptr_t)pt->second.first & 15))
            return rs_fail(ctx, RS_EINVAL, "pos.table.f32 must be 16-byte aligned f32 [2*Tcap-1][d_model]");
        w.pos_table = reinterpret_cast<const float*>(pt->second.first);
        w.pos_table_bytes = pt->second.second;
        ctx->has_f32 = true;
    }
    if (ctx->precision_f32 && !ctx->has_f32) ctx->precision_f32 = 0;     // (the tensors were re-registered without the f32 set)
    auto it = ctx->tensors.find("pos.table");
    if (it == ctx->tensors.end()) return rs_fail(ctx, RS_EMISSING, "weight tensor 'pos.table' was not registered");
    const size_t rowb = dm * sizeof(uint16_t);
    if (it->second.second % rowb || !((it->second.second / rowb) & 1))
        return rs_fail(ctx, RS_EINVAL, "pos.table must be bf16 [2*Tcap-1][d_model]");
    // optional derived weights: the position table already projected by each layer's linear_pos
    // (it depends on nothing but the weights, so a caller can compute it once with rs_gemm_bf16)
    for (int i = 0; i < d.n_layers; ++i) {
        auto pp = ctx->tensors.find("L" + std::to_string(i) + ".att.pos_proj");
        if (pp == ctx->tensors.end()) { ctx->layers[i].pos_proj = nullptr; continue; }
        if (pp->second.second != it->second.second || ((uintptr_t)pp->second.first & 15))
            return rs_fail(ctx, RS_EINVAL, "L%d.att.pos_proj must be 16-byte aligned and as large as pos.table", i);
        ctx->layers[i].pos_proj = reinterpret_cast<const uint16_t*>(pp->second.first);
    }
    ctx->finalized = true;
    return RS_OK;
}

int rs_stream_create(void** out, int device, const uint32_t* cu_mask, int n_words, int priority) {
    if (!out) return RS_EINVAL;
    *out = nullptr;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return RS_EHIP;
    if (cur != device && hipSetDevice(device) != hipSuccess) return RS_EHIP;
    hipStream_t s = nullptr;
    hipError_t e;
    if (cu_mask && n_words > 0) e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask);
    else e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
    if (cur != device) hipSetDevice(cur);          // the caller's current device is not a side effect of this call
    if (e != hipSuccess) return RS_EHIP;
    *out = (void*)s;
    return RS_OK;
}

int rs_stream_destroy(void* stream) {
    if (!stream) return RS_OK;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? RS_OK : RS_EHIP;
}

int rs_set_option(rs_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return RS_EINVAL;
    if (!strcmp(key, "decode_screen")) { ctx->decode_screen = value != 0; return RS_OK; }
    if (!strcmp(key, "decode_narrow")) { ctx->decode_narrow = value != 0; return RS_OK; }
    if (!strcmp(key, "precision_f32")) {
        if (value && !ctx->has_f32)
            return rs_fail(ctx, RS_EMISSING, "precision_f32: the float32 weights (\"*.f32\" tensors) are not registered / rs_finalize has not run");
        ctx->precision_f32 = value != 0;
        return RS_OK;
    }
    if (!strcmp(key, "gemm_f32_x3")) {             // float32 products as three bf16 matrix-core terms (k_f32.hip X3): the AV-HuBERT family, and the
        ctx->gemm_f32_x3 = value != 0;             // "precision_f32" mode of the others ("fp32x3": not an IEEE chain; held to the same goldens)
        return RS_OK;
    }
    if (!strcmp(key, "k2_conv2_fused")) {          // conv2 with its patches gathered into LDS (1, default) or as patch matrix + GEMM launch (0): same bits
        if (!ctx->k2) return rs_fail(ctx, RS_EINVAL, "option 'k2_conv2_fused' applies to a Zipformer context only");
        ctx->k2_conv2_fused = value != 0;
        return RS_OK;
    }
    if (!strcmp(key, "k2_cnx_fused")) {            // the ConvNeXt pointwise pair as one kernel (1, default) or as two GEMM launches (0): same bits
        if (!ctx->k2) return rs_fail(ctx, RS_EINVAL, "option 'k2_cnx_fused' applies to a Zipformer context only");
        ctx->k2_cnx_fused = value != 0;
        return RS_OK;
    }
    if (ctx->k2) return rs_fail(ctx, RS_EINVAL, "option '%s' does not apply to a Zipformer context", key);
    if (!strcmp(key, "fuse_glu")) {
        if (value < 0 || value > 1) return rs_fail(ctx, RS_EINVAL, "fuse_glu must be 0 or 1");
        ctx->fuse_glu = value;
        return RS_OK;
    }
    if (!strcmp(key, "defer_out_norm")) { ctx->defer_out_norm = value != 0; return RS_OK; }
    return rs_fail(ctx, RS_EINVAL, "unknown option '%s'", key);
}

int rs_encoder_set_taps(rs_ctx* ctx, float* sub_out, float* layer_out, const int32_t* layer_ids, int n_layer_ids) {
    if (!ctx) return RS_EINVAL;
    if (ctx->k2) return rs_fail(ctx, RS_EINVAL, "taps: a Zipformer context takes rs_k2_encoder_set_taps");
    if (n_layer_ids < 0 || (n_layer_ids > 0 && (!layer_out || !layer_ids))) return rs_fail(ctx, RS_EINVAL, "taps: null pointer");
    for (int i = 0; i < n_layer_ids; ++i)
        if (layer_ids[i] < 0 || layer_ids[i] >= ctx->d.n_layers) return rs_fail(ctx, RS_EINVAL, "taps: layer %d out of range", layer_ids[i]);
    ctx->tap_sub = sub_out;
    ctx->tap_layers = n_layer_ids > 0 ? layer_out : nullptr;
    ctx->tap_ids.assign(layer_ids, layer_ids + n_layer_ids);
    return RS_OK;
}

int rs_mel_frames(const rs_ctx* ctx, int n_samples) {
    if (ctx->d.frontend_kind == 2) return (n_samples + ctx->d.hop_length / 2) / ctx->d.hop_length;     // kaldi, snip_edges = false
    return n_samples / ctx->d.hop_length + (ctx->d.frontend_kind == 1 ? 1 : 0);
}

int rs_enc_frames(const rs_ctx* ctx, int n) {
    if (ctx->k2) return rs_k2_enc_frames_impl(ctx, n);
    for (int s = 0; s < ctx->d.sub_stages; ++s) n = rs_conv_len(n, ctx->d.sub_kind);
    return n;
}

int rs_encoder_set_ctc_out(rs_ctx* ctx, float* probs, float* blank_prob) {
    if (!ctx) return RS_EINVAL;
    if (ctx->k2 && (probs || blank_prob)) return rs_fail(ctx, RS_EINVAL, "a Zipformer context has no CTC head");
    if ((probs || blank_prob) && ctx->d.ctc_vocab <= 0) return rs_fail(ctx, RS_EINVAL, "this model has no CTC head (rs_dims.ctc_vocab)");
    ctx->ctc_probs = probs;
    ctx->ctc_blank = blank_prob;
    return RS_OK;
}

}  // extern "C"

namespace {

struct EncPlan {
    int T[5], F[5];  // per stage time / freq extents (index 0 = mel)
    size_t off_lens, off_sa, off_sb, off_x, off_hn, off_big, off_ctx, off_posp, off_stats, off_col, off_ctc, total;
    int chunk;       // Conv2dSubsampling: utterances per pass of conv0 / patch gather / dense-conv GEMM
    size_t col_bytes = 0;   // > 0: the plan holds the gathered patch matrix ($RS_SUB_IM2COL)
};

EncPlan plan_encoder(const rs_ctx* ctx, int B, int t_max) {
    const rs_dims& d = ctx->d;
    EncPlan p{};
    p.T[0] = t_max; p.F[0] = d.n_mels;
    for (int s = 1; s <= d.sub_stages; ++s) { p.T[s] = rs_conv_len(p.T[s - 1], d.sub_kind); p.F[s] = rs_conv_len(p.F[s - 1], d.sub_kind); }
    const size_t C = d.sub_channels, dm = d.d_model;
    const size_t Tp = p.T[d.sub_stages] > 0 ? p.T[d.sub_stages] : 1, M = (size_t)B * Tp;
    size_t widest = (size_t)d.ff_dim;
    if (3 * dm > widest) widest = 3 * dm;
    size_t o = 0;
    p.off_lens = o; o += rs_align((size_t)4 * B * 4);
    p.chunk = B;
    p.off_col = 0;
    if (d.sub_kind == 1) {
        // sa = conv0 output of ONE chunk of utterances [chunk][T1][F1][C]; sb = the dense conv's output of the WHOLE batch
        // [B][T2][F2][C].  The GEMM reads the 3 x 3 patches in place (rs_gemm_args.conv_C): the chunk keeps sa within the
        // kernel's 32-bit offsets (2 GiB here).  With $RS_SUB_IM2COL (A/B and test hook: the gathered patch matrix
        // col [chunk * T2 * F2][9C], the first form — 16 GB written and read again per 256 x 10 s, 4.2 ms) the chunk keeps col
        // near 1 GiB instead.
        const bool gathered = getenv("RS_SUB_IM2COL") != nullptr;
        const size_t per_utt_col = (size_t)(p.T[2] > 0 ? p.T[2] : 1) * p.F[2] * 9 * C * 2;
        const size_t per_utt_sa = (size_t)(p.T[1] > 0 ? p.T[1] : 1) * p.F[1] * C * 2;
        size_t chunk = gathered ? ((size_t)1 << 30) / per_utt_col : ((size_t)1 << 31) / per_utt_sa;
        if (chunk < 1) chunk = 1;
        if (chunk > (size_t)B) chunk = (size_t)B;
        while (chunk > 1 && chunk * (size_t)(p.T[1] > 0 ? p.T[1] : 1) > 65535) --chunk;   // grid limit of the conv0 / gather kernels
        p.chunk = (int)chunk;
        p.off_sa = o; o += rs_align(chunk * per_utt_sa);
        p.off_col = o; o += gathered ? rs_align(chunk * per_utt_col) : 0;
        p.col_bytes = gathered ? chunk * per_utt_col : 0;
        p.off_sb = o; o += rs_align((size_t)B * (p.T[2] > 0 ? p.T[2] : 1) * p.F[2] * C * 2);
    } else {
        const size_t sub_elems = (size_t)B * p.T[2] * p.F[2] * C;  // stage-2 extent is the largest stored one
        p.off_sa = o; o += rs_align(sub_elems * 2);
        p.off_sb = o; o += rs_align(sub_elems * 2);
    }
    p.off_x = o; o += rs_align(M * dm * 4);
    p.off_hn = o; o += rs_align(M * dm * 2);
    p.off_big = o; o += rs_align(M * widest * 2);
    p.off_ctx = o; o += rs_align(M * dm * 2);
    p.off_posp = o; o += rs_align((2 * Tp) * dm * 2);
    p.off_stats = o; o += rs_align(M * 2 * 4);          // (mean, rstd) per row of a deferred output norm
    p.off_ctc = o;
    if (d.ctc_vocab > 0) o += rs_align(M * (size_t)rs_ctc_pad(d.ctc_vocab) * 4);   // CTC logits when only the blank column is wanted
    p.total = o + 256;
    return p;
}

}  // namespace

extern "C" {

size_t rs_workspace_bytes(const rs_ctx* ctx, int B, int max_samples) {
    if (!ctx || B <= 0 || max_samples < 0) return 0;
    const int t_max = rs_mel_frames(ctx, max_samples);
    const size_t fe = rs_align((size_t)B * (t_max > 0 ? t_max : 1) * ctx->d.n_mels * 4) + 256;
    if (ctx->k2) {
        const size_t enc2 = rs_k2_workspace_bytes_impl(ctx, B, t_max), dec2 = rs_rnnt_workspace_bytes(ctx, B);
        const size_t m2 = fe > enc2 ? fe : enc2;
        return m2 > dec2 ? m2 : dec2;
    }
    size_t enc = plan_encoder(ctx, B, t_max > 0 ? t_max : 1).total;
    if (ctx->has_f32) {                      // the float32 parity mode keeps float32 activations: about twice the scratch
        const size_t enc32 = rs_encoder_f32_workspace_bytes(ctx, B, t_max > 0 ? t_max : 1);
        if (enc32 > enc) enc = enc32;
    }
    const size_t dec = rs_rnnt_workspace_bytes(ctx, B);
    size_t m = fe > enc ? fe : enc;
    return m > dec ? m : dec;
}

int rs_host_stage_rows(float* dst, size_t dst_pitch, int width, const float* const* rows, const int32_t* lens,
                       int n_rows, int total_rows, int32_t* dst_lens) {
    if (!dst || !dst_lens || n_rows < 0 || total_rows < n_rows || width < 0 || (size_t)width > dst_pitch) return RS_EINVAL;
    if (n_rows > 0 && (!rows || !lens)) return RS_EINVAL;
    for (int b = 0; b < n_rows; ++b) {
        const int n = lens[b];
        if (n < 0 || n > width || (n > 0 && !rows[b])) return RS_EINVAL;
        float* row = dst + (size_t)b * dst_pitch;
        if (n > 0) memcpy(row, rows[b], (size_t)n * sizeof(float));
        if (n < width) memset(row + n, 0, (size_t)(width - n) * sizeof(float));
        dst_lens[b] = n;
    }
    for (int b = n_rows; b < total_rows; ++b) dst_lens[b] = 0;
    return RS_OK;
}

int rs_frontend_logmel(rs_ctx* ctx, const float* audio, const int32_t* lens, int B, int audio_stride,
                       int pad_left, int pad_right, int t_max, float* feats, int32_t* n_frames, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_frontend_logmel");
    if (B < 0 || t_max < 0 || pad_left < 0 || pad_right < 0) return rs_fail(ctx, RS_EINVAL, "frontend: negative size");
    if (B == 0 || t_max == 0) return RS_OK;
    if (!audio || !lens || !feats || !n_frames || !workspace) return rs_fail(ctx, RS_EINVAL, "frontend: null pointer");
    const size_t need = (size_t)B * t_max * ctx->d.n_mels * 4;
    if (workspace_bytes < need) return rs_fail(ctx, RS_EWORKSPACE, "frontend: workspace %zu < %zu", workspace_bytes, need);
    return rs_launch_frontend(ctx, audio, lens, B, audio_stride, pad_left, pad_right, t_max, feats, n_frames,
                              reinterpret_cast<float*>(workspace), (hipStream_t)stream);
}

int rs_encoder_forward(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max, float* enc_out,
                       float* joint_enc, int32_t* enc_lens, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_encoder_forward");
    if (B <= 0 || t_max <= 0) return B < 0 || t_max < 0 ? rs_fail(ctx, RS_EINVAL, "encoder: negative size") : RS_OK;
    if (!feats || !n_frames || !joint_enc || !enc_lens || !workspace) return rs_fail(ctx, RS_EINVAL, "encoder: null pointer");
    const rs_dims& d = ctx->d;
    hipStream_t s = (hipStream_t)stream;
    if (ctx->k2) return rs_k2_encoder_forward_impl(ctx, feats, n_frames, B, t_max, enc_out, joint_enc, enc_lens, workspace, workspace_bytes, s);
    if (ctx->precision_f32)
        return rs_encoder_forward_f32(ctx, feats, n_frames, B, t_max, enc_out, joint_enc, enc_lens, workspace, workspace_bytes, s);
    const EncPlan pl = plan_encoder(ctx, B, t_max);
    if (workspace_bytes < pl.total) return rs_fail(ctx, RS_EWORKSPACE, "encoder: workspace %zu < %zu", workspace_bytes, pl.total);
    char* ws = reinterpret_cast<char*>(workspace);
    int32_t* lens_stage = reinterpret_cast<int32_t*>(ws + pl.off_lens);
    uint16_t* sa = reinterpret_cast<uint16_t*>(ws + pl.off_sa);
    uint16_t* sb = reinterpret_cast<uint16_t*>(ws + pl.off_sb);
    float* x = reinterpret_cast<float*>(ws + pl.off_x);
    uint16_t* hn = reinterpret_cast<uint16_t*>(ws + pl.off_hn);
    uint16_t* big = reinterpret_cast<uint16_t*>(ws + pl.off_big);
    uint16_t* ctxb = reinterpret_cast<uint16_t*>(ws + pl.off_ctx);
    uint16_t* posp = reinterpret_cast<uint16_t*>(ws + pl.off_posp);
    float* ln_stats = reinterpret_cast<float*>(ws + pl.off_stats);
    const int C = d.sub_channels, dm = d.d_model, ff = d.ff_dim, S = d.sub_stages;
    const int Tp = pl.T[S], M = B * Tp;
    int rc;
#define RS_TRY(call) do { rc = (call); if (rc != RS_OK) return rc; } while (0)

    // ---- subsampling -------------------------------------------------------------------------
    RS_TRY(rs_launch_enc_lens(ctx, n_frames, B, lens_stage, s));
    if (Tp <= 0) return rs_fail(ctx, RS_EINVAL, "encoder: %d feature frames are too few for the subsampling", t_max);
    if (d.sub_kind == 1) {
        // ESPnet Conv2dSubsampling: conv0 (VALU) -> 3x3 patches -> dense conv as one GEMM per chunk of utterances (bias, ReLU,
        // rows past an utterance's T2 zeroed), into sb [B][T2][F2][C]
        uint16_t* col = reinterpret_cast<uint16_t*>(ws + pl.off_col);
        const int T1 = pl.T[1], F1 = pl.F[1], T2 = pl.T[2], F2 = pl.F[2];
        for (int b0 = 0; b0 < B; b0 += pl.chunk) {
            const int bc = B - b0 < pl.chunk ? B - b0 : pl.chunk;
            RS_TRY(rs_launch_sub2d_conv0(ctx, feats, lens_stage, b0, bc, t_max, T1, F1, sa, s));
            const bool patches_in_place = pl.col_bytes == 0;                              // (plan_encoder: $RS_SUB_IM2COL)
            if (!patches_in_place) RS_TRY(rs_launch_im2col3x3s2(ctx, sa, bc, T1, F1, T2, F2, col, s));
            rs_gemm_args g{};
            if (patches_in_place) { g.A = sa; g.conv_C = C; g.conv_T1 = T1; g.conv_F1 = F1; g.conv_T2 = T2; g.conv_F2 = F2; }
            else g.A = col;
            g.lda = 9 * C; g.W = ctx->sub_conv1_w; g.ldw = 9 * C; g.out = sb + (size_t)b0 * T2 * F2 * C; g.ldc = C;
            g.M = bc * T2 * F2; g.N = C; g.K = 9 * C;
            g.flags = RS_GEMM_BIAS | RS_GEMM_RELU | RS_GEMM_ROWMASK; g.bias = ctx->sub_conv1_b; g.alpha = 1.0f;
            g.mask_lens = lens_stage + B + b0; g.mask_rows_per_step = F2; g.mask_steps = T2;
            RS_TRY(rs_launch_gemm(ctx, g, s));
        }
    } else
    RS_TRY(rs_launch_sub_conv0_dw1(ctx, feats, lens_stage, B, t_max, pl.T[2], pl.F[2], sa, s));
    for (int st = 2; st <= S && d.sub_kind == 0; ++st) {
        if (st > 2)
            RS_TRY(rs_launch_sub_dw(ctx, sb, ctx->sub_dw_w[st - 2], ctx->sub_dw_b[st - 2], lens_stage + (st - 1) * B, st, B,
                                    pl.T[st - 1], pl.F[st - 1], pl.T[st], pl.F[st], sa, s));
        rs_gemm_args g{};
        g.A = sa; g.lda = C; g.W = ctx->sub_pw_w[st - 2]; g.ldw = C; g.out = sb; g.ldc = C;
        g.M = B * pl.T[st] * pl.F[st]; g.N = C; g.K = C;
        g.flags = RS_GEMM_BIAS | RS_GEMM_RELU | RS_GEMM_ROWMASK; g.bias = ctx->sub_pw_b[st - 2]; g.alpha = 1.0f;
        g.mask_lens = lens_stage + (st - 1) * B; g.mask_rows_per_step = pl.F[st]; g.mask_steps = pl.T[st];
        RS_TRY(rs_launch_gemm(ctx, g, s));
    }
    {
        rs_gemm_args g{};
        const int K = C * pl.F[S];
        g.A = sb; g.lda = K; g.W = ctx->sub_out_w; g.ldw = K; g.out = x; g.ldc = dm; g.M = M; g.N = dm; g.K = K;
        g.flags = RS_GEMM_BIAS | RS_GEMM_OUT_F32; g.bias = ctx->sub_out_b;
        g.alpha = d.xscaling ? sqrtf((float)dm) : 1.0f;
        RS_TRY(rs_launch_gemm(ctx, g, s));
    }
    const int32_t* lens = lens_stage + (S - 1) * B;
    RS_HIP(ctx, hipMemcpyAsync(enc_lens, lens, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    if (ctx->tap_sub) RS_HIP(ctx, hipMemcpyAsync(ctx->tap_sub, x, (size_t)M * dm * 4, hipMemcpyDeviceToDevice, s));

    // ---- position table slice for this T ---------------------------------------------------------
    const auto& pt = ctx->tensors.at("pos.table");
    const int tcap = (int)((pt.second / ((size_t)dm * 2) + 1) / 2);
    if (Tp > tcap) return rs_fail(ctx, RS_EINVAL, "encoder: T'=%d exceeds the registered pos.table capacity %d", Tp, tcap);
    const uint16_t* pos_slice = reinterpret_cast<const uint16_t*>(pt.first) + (size_t)(tcap - Tp) * dm;
    const int npos = 2 * Tp - 1;

    // res_ln: the residual operand still has to go through the PREVIOUS layer's output norm (deferred, see below)
    const rs_layer_w* res_ln = nullptr;
    auto gemm = [&](const uint16_t* A, int lda, const uint16_t* W, int K, void* out, int ldc, int Mr, int N, int flags,
                    const float* bias, float alpha, const float* res) -> int {
        rs_gemm_args g{};
        g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.out = out; g.ldc = ldc; g.M = Mr; g.N = N; g.K = K;
        g.flags = flags; g.bias = bias; g.alpha = alpha; g.residual = res;
        if (res && res_ln) { g.res_ln_stats = ln_stats; g.res_ln_g = res_ln->ln_out_g; g.res_ln_b = res_ln->ln_out_b; res_ln = nullptr; }
        return rs_launch_gemm(ctx, g, s);
    };
    const int RES = RS_GEMM_BIAS | RS_GEMM_RESIDUAL | RS_GEMM_OUT_F32;
    // fuse_glu (default 1): the GLU is applied to the float32 pw1 accumulators in the GEMM epilogue for EVERY batch
    // size — one rounding point, so an utterance's arithmetic does not depend on the batch it rides in
    const bool glu_fused = ctx->fuse_glu != 0 && (dm % 32) == 0;

    for (int i = 0; i < d.n_layers; ++i) {
        const rs_layer_w& L = ctx->layers[i];
        const bool last = i == d.n_layers - 1;
        // 1/2 FFN  (layers > 0: hn was produced by the previous layer's fused output-norm kernel)
        if (i == 0) RS_TRY(rs_launch_layernorm(ctx, x, L.ln_ff1_g, L.ln_ff1_b, M, dm, d.ln_eps, hn, nullptr, s));
        RS_TRY(gemm(hn, dm, L.ff1_w1, dm, big, ff, M, ff, RS_GEMM_BIAS | RS_GEMM_SILU, L.ff1_b1, 1.0f, nullptr));
        RS_TRY(gemm(big, ff, L.ff1_w2, ff, x, dm, M, dm, RES, L.ff1_b2, 0.5f, x));
        // rel-pos MHSA
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_att_g, L.ln_att_b, M, dm, d.ln_eps, hn, nullptr, s));
        RS_TRY(gemm(hn, dm, L.qkv_w, dm, big, 3 * dm, M, 3 * dm, RS_GEMM_BIAS, L.qkv_b, 1.0f, nullptr));
        const uint16_t* pproj = posp;
        if (L.pos_proj) pproj = L.pos_proj + (size_t)(tcap - Tp) * dm;      // rows of the pre-projected table
        else RS_TRY(gemm(pos_slice, dm, L.pos_w, dm, posp, dm, npos, dm, 0, nullptr, 1.0f, nullptr));
        RS_TRY(rs_launch_attention(ctx, big, pproj, L.bias_u, L.bias_v, lens, B, Tp, ctxb, s));
        RS_TRY(gemm(ctxb, dm, L.out_w, dm, x, dm, M, dm, RES, L.out_b, 1.0f, x));
        // conv module
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_conv_g, L.ln_conv_b, M, dm, d.ln_eps, hn, nullptr, s));
        // pw1's weight rows are interleaved (values / gates in blocks of 32): the GLU is applied in the GEMM
        // epilogue ([M][d] out, half the bytes written and read back); with fuse_glu = 0 the plain product is
        // stored and the conv kernel pairs the columns up itself
        if (glu_fused) {
            RS_TRY(gemm(hn, dm, L.pw1_w, dm, big, dm, M, 2 * dm, RS_GEMM_BIAS | RS_GEMM_GLU, L.pw1_b, 1.0f, nullptr));
            RS_TRY(rs_launch_glu_dwconv(ctx, big, RS_GLU_APPLIED, L.dw_w, L.dw_b, lens, B, Tp, dm, d.conv_kernel, ctxb, s));
        } else {
            RS_TRY(gemm(hn, dm, L.pw1_w, dm, big, 2 * dm, M, 2 * dm, RS_GEMM_BIAS, L.pw1_b, 1.0f, nullptr));
            RS_TRY(rs_launch_glu_dwconv(ctx, big, RS_GLU_BLOCK32, L.dw_w, L.dw_b, lens, B, Tp, dm, d.conv_kernel, ctxb, s));
        }
        RS_TRY(gemm(ctxb, dm, L.pw2_w, dm, x, dm, M, dm, RES, L.pw2_b, 1.0f, x));
        // 1/2 FFN
        RS_TRY(rs_launch_layernorm(ctx, x, L.ln_ff2_g, L.ln_ff2_b, M, dm, d.ln_eps, hn, nullptr, s));
        RS_TRY(gemm(hn, dm, L.ff2_w1, dm, big, ff, M, ff, RS_GEMM_BIAS | RS_GEMM_SILU, L.ff2_b1, 1.0f, nullptr));
        RS_TRY(gemm(big, ff, L.ff2_w2, ff, x, dm, M, dm, RES, L.ff2_b2, 0.5f, x));
        // output norm (in place on the residual stream; the last layer also emits the bf16 copy
        // that feeds the joint's encoder projection)
        if (last && d.final_norm) {
            // ESPnet: the block's norm_final, then the encoder's after_norm; hn = bf16 of the latter feeds the joint / CTC heads
            RS_TRY(rs_launch_layernorm(ctx, x, L.ln_out_g, L.ln_out_b, M, dm, d.ln_eps, nullptr, x, s));
        } else if (last) {
            // the f32 rows of the final norm are read by the caller's enc_out copy / a parity tap only
            bool want_f32 = enc_out != nullptr;
            for (size_t k = 0; k < ctx->tap_ids.size(); ++k) want_f32 = want_f32 || ctx->tap_ids[k] == i;
            RS_TRY(rs_launch_layernorm(ctx, x, L.ln_out_g, L.ln_out_b, M, dm, d.ln_eps, hn, want_f32 ? x : nullptr, s));
        } else {
            // output norm + the next layer's first norm on one read of the row.  The normalised f32 rows themselves have
            // ONE reader, the residual operand of the next layer's first FFN: unless a parity tap wants this layer's
            // output, they are not written (145 MB per boundary at B = 256) — the kernel leaves (mean, rstd) per row and
            // that GEMM's epilogue normalises x on the fly, with the same arithmetic (bit-identical: tests/test_gpu_pipeline.py)
            const rs_layer_w& Ln = ctx->layers[i + 1];
            bool tapped = ctx->defer_out_norm == 0;
            for (size_t k = 0; k < ctx->tap_ids.size(); ++k) tapped = tapped || ctx->tap_ids[k] == i;
            RS_TRY(rs_launch_layernorm2(ctx, x, L.ln_out_g, L.ln_out_b, Ln.ln_ff1_g, Ln.ln_ff1_b, M, dm, d.ln_eps,
                                        tapped ? x : nullptr, hn, tapped ? nullptr : ln_stats, s));
            if (!tapped) res_ln = &L;
        }
        for (size_t k = 0; k < ctx->tap_ids.size(); ++k)     // parity taps: x now holds this layer's output
            if (ctx->tap_ids[k] == i)
                RS_HIP(ctx, hipMemcpyAsync(ctx->tap_layers + k * (size_t)M * dm, x, (size_t)M * dm * 4, hipMemcpyDeviceToDevice, s));
    }
    if (d.final_norm) RS_TRY(rs_launch_layernorm(ctx, x, ctx->final_norm_g, ctx->final_norm_b, M, dm, d.ln_eps, hn, enc_out ? x : nullptr, s));
    if (enc_out) RS_HIP(ctx, hipMemcpyAsync(enc_out, x, (size_t)M * dm * 4, hipMemcpyDeviceToDevice, s));
    RS_TRY(gemm(hn, dm, ctx->jenc_w, dm, joint_enc, d.joint_hidden, M, d.joint_hidden, RS_GEMM_BIAS | RS_GEMM_OUT_F32,
                ctx->jenc_b, 1.0f, nullptr));
    if (d.ctc_vocab > 0 && (ctx->ctc_probs || ctx->ctc_blank)) {
        // CTC posteriors of every frame: logits by the same GEMM family, softmax in place
        float* z = ctx->ctc_probs ? ctx->ctc_probs : reinterpret_cast<float*>(ws + pl.off_ctc);
        const int Vp = rs_ctc_pad(d.ctc_vocab);
        RS_TRY(gemm(hn, dm, ctx->ctc_w, dm, z, Vp, M, Vp, RS_GEMM_BIAS | RS_GEMM_OUT_F32, ctx->ctc_b, 1.0f, nullptr));
        RS_TRY(rs_launch_ctc_softmax(ctx, z, M, d.ctc_vocab, Vp, d.blank_id, ctx->ctc_blank, s));
    }
#undef RS_TRY
    return RS_OK;
}

int rs_rnnt_greedy(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int u_max,
                   int32_t* ids, int32_t* frames, int32_t* n_ids, void* workspace, size_t workspace_bytes,
                   void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_rnnt_greedy");
    if (B < 0 || tp_max < 0 || u_max < 0) return rs_fail(ctx, RS_EINVAL, "rnnt: negative size");
    if (B == 0) return RS_OK;
    if (!joint_enc || !enc_lens || !ids || !frames || !n_ids || !workspace) return rs_fail(ctx, RS_EINVAL, "rnnt: null pointer");
    if (tp_max == 0) { RS_HIP(ctx, hipMemsetAsync(n_ids, 0, (size_t)B * 4, (hipStream_t)stream)); return RS_OK; }
    return rs_rnnt_greedy_impl(ctx, joint_enc, enc_lens, B, tp_max, u_max, ids, frames, n_ids, workspace,
                               workspace_bytes, (hipStream_t)stream);
}

static int alsd_steps(int tp_max, double ratio, int abs_len) {
    const int budget = abs_len >= 0 ? abs_len : (int)(ratio * (double)tp_max);
    return tp_max + budget > 0 ? tp_max + budget : 1;
}

size_t rs_rnnt_alsd_workspace_bytes(const rs_ctx* ctx, int B, int beam, int tp_max, double max_target_ratio, int max_target_abs) {
    if (!ctx || B <= 0 || beam <= 0 || tp_max < 0 || (max_target_abs < 0 && !(max_target_ratio >= 0.0))) return 0;
    const int W = beam < ctx->d.n_logits - 1 ? beam : ctx->d.n_logits - 1;
    return rs_rnnt_alsd_workspace_bytes_impl(ctx, B, W, alsd_steps(tp_max, max_target_ratio, max_target_abs));
}

int rs_rnnt_alsd(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam,
                 double max_target_ratio, int max_target_abs, int flags, int out_cap, int32_t* ids, int32_t* steps,
                 int32_t* n_ids, float* scores, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_rnnt_alsd");
    if (ctx->k2) return rs_fail(ctx, RS_EINVAL, "alsd: a Zipformer context has a stateless decoder; only rs_rnnt_greedy applies (sherpa-onnx greedy_search)");
    if (B < 0 || tp_max < 0 || out_cap < 0) return rs_fail(ctx, RS_EINVAL, "alsd: negative size");
    if (max_target_abs < 0 && !(max_target_ratio >= 0.0 && max_target_ratio <= 64.0))
        return rs_fail(ctx, RS_EINVAL, "alsd: max_target_ratio must be in [0, 64]");
    if (B == 0) return RS_OK;
    if (!joint_enc || !enc_lens || !ids || !steps || !n_ids || !scores || !workspace) return rs_fail(ctx, RS_EINVAL, "alsd: null pointer");
    if (tp_max == 0) {
        RS_HIP(ctx, hipMemsetAsync(n_ids, 0, (size_t)B * 4, (hipStream_t)stream));
        RS_HIP(ctx, hipMemsetAsync(scores, 0, (size_t)B * 4, (hipStream_t)stream));
        return RS_OK;
    }
    return rs_rnnt_alsd_impl(ctx, joint_enc, enc_lens, B, tp_max, beam, max_target_ratio, max_target_abs,
                             (flags & RS_ALSD_SCORE_NORM) != 0, (flags & RS_ALSD_MERGE) != 0, out_cap, ids, steps, n_ids, scores,
                             workspace, workspace_bytes, (hipStream_t)stream);
}

size_t rs_rnnt_beam_workspace_bytes(const rs_ctx* ctx, int B, int beam, int tp_max, int max_pops) {
    if (!ctx || B <= 0 || beam <= 0 || tp_max < 0 || max_pops < 0 || ctx->d.n_logits < 2) return 0;
    return rs_rnnt_beam_workspace_bytes_impl(ctx, B, beam, tp_max, max_pops);
}

int rs_rnnt_beam(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam, int flags, int max_pops,
                 int out_cap, int32_t* ids, int32_t* frames, int32_t* n_ids, float* scores, int32_t* pops, void* workspace,
                 size_t workspace_bytes, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!ctx->finalized) return rs_fail(ctx, RS_ESTATE, "rs_finalize must precede rs_rnnt_beam");
    if (ctx->k2) return rs_fail(ctx, RS_EINVAL, "beam search: a Zipformer context has a stateless decoder; only rs_rnnt_greedy applies (sherpa-onnx greedy_search)");
    if (B < 0 || tp_max < 0 || out_cap < 0 || max_pops < 0) return rs_fail(ctx, RS_EINVAL, "beam search: negative size");
    if (beam < 1) return rs_fail(ctx, RS_EINVAL, "beam search: beam size must be >= 1");
    if (B == 0) return RS_OK;
    if (!joint_enc || !enc_lens || !ids || !n_ids || !scores || !pops || !workspace) return rs_fail(ctx, RS_EINVAL, "beam search: null pointer");
    if (tp_max == 0) {
        RS_HIP(ctx, hipMemsetAsync(n_ids, 0, (size_t)B * 4, (hipStream_t)stream));
        RS_HIP(ctx, hipMemsetAsync(scores, 0, (size_t)B * 4, (hipStream_t)stream));
        RS_HIP(ctx, hipMemsetAsync(pops, 0, (size_t)B * 4, (hipStream_t)stream));
        return RS_OK;
    }
    return rs_rnnt_beam_impl(ctx, joint_enc, enc_lens, B, tp_max, beam, (flags & RS_BEAM_SCORE_NORM) != 0, max_pops, out_cap, ids,
                             frames, n_ids, scores, pops, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- profiling -------------------------------------------------------------------------------------
int rs_profile_enable(rs_ctx* ctx, int class_mask) {
    if (!ctx) return RS_EINVAL;
    ctx->prof_mask = class_mask;
    return RS_OK;
}
int rs_profile_reset(rs_ctx* ctx) {
    if (!ctx) return RS_EINVAL;
    for (auto& p : ctx->prof) { p.used = 0; p.flops = p.bytes = p.ms_acc = 0; p.launches = 0; p.detail.clear(); p.read = 0; }
    return RS_OK;
}
int rs_profile_read(rs_ctx* ctx, int klass, double* ms, int64_t* launches, double* flops, double* bytes) {
    if (!ctx || klass <= 0) return RS_EINVAL;
    rs_prof_slot& p = ctx->prof[rs_prof_class_index(klass)];
    double total = p.ms_acc;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        RS_HIP(ctx, hipEventSynchronize(p.ev[i + 1]));
        float t = 0;
        RS_HIP(ctx, hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]));
        total += t;
        if (p.read < p.detail.size()) p.detail[p.read++].ms = t;      // launches are bracketed and read in the same order
    }
    p.ms_acc = total;
    p.used = 0;
    if (ms) *ms = total;
    if (launches) *launches = p.launches;
    if (flops) *flops = p.flops;
    if (bytes) *bytes = p.bytes;
    return RS_OK;
}

int rs_profile_read_launches(rs_ctx* ctx, int klass, int32_t* shapes, double* flops, float* ms, int cap, int* n_out) {
    if (!ctx || klass <= 0 || cap < 0 || !n_out) return RS_EINVAL;
    if (int rc = rs_profile_read(ctx, klass, nullptr, nullptr, nullptr, nullptr); rc != RS_OK) return rc;   // folds pending events in
    const rs_prof_slot& p = ctx->prof[rs_prof_class_index(klass)];
    const int n = (int)p.read < cap ? (int)p.read : cap;
    for (int i = 0; i < n; ++i) {
        const rs_prof_launch& l = p.detail[i];
        if (shapes) { shapes[4 * i] = l.M; shapes[4 * i + 1] = l.N; shapes[4 * i + 2] = l.K; shapes[4 * i + 3] = l.flags; }
        if (flops) flops[i] = l.flops;
        if (ms) ms[i] = l.ms;
    }
    *n_out = (int)p.read;
    return RS_OK;
}

// ---- single-operator entry points --------------------------------------------------------------------
int rs_gemm_bf16(rs_ctx* ctx, const uint16_t* A, int lda, const uint16_t* W, int ldw, void* out, int ldc, int M, int N,
                 int K, int flags, const float* bias, float alpha, const float* residual, const int32_t* mask_lens,
                 int mask_rows_per_step, int mask_steps, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (M == 0 || N == 0) return RS_OK;
    if (!A || !W || !out) return rs_fail(ctx, RS_EINVAL, "gemm: null pointer");
    rs_gemm_args g{A, lda, W, ldw, out, ldc, M, N, K, flags, bias, alpha, residual, mask_lens, mask_rows_per_step, mask_steps};
    return rs_launch_gemm(ctx, g, (hipStream_t)stream);
}

int rs_gemm_f32(rs_ctx* ctx, const float* A, int lda, const float* W, int ldw, float* out, int ldc, int M, int N, int K, int flags,
                const float* bias, float alpha, const float* residual, const int32_t* mask_lens, int mask_rows_per_step,
                int mask_steps, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (M == 0 || N == 0) return RS_OK;
    if (!A || !W || !out) return rs_fail(ctx, RS_EINVAL, "gemm_f32: null pointer");
    return rs_launch_gemm_f32(ctx, A, lda, W, ldw, out, ldc, M, N, K, flags, bias, alpha, residual, mask_lens, mask_rows_per_step,
                              mask_steps, (hipStream_t)stream);
}

int rs_relpos_attention_f32(rs_ctx* ctx, const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                            const int32_t* lens, int B, int T, float* ctx_out, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!qkv || !pos || !bias_u || !bias_v || !lens || !ctx_out) return rs_fail(ctx, RS_EINVAL, "attention_f32: null pointer");
    return rs_launch_attention_f32(ctx, qkv, pos, bias_u, bias_v, lens, B, T, ctx_out, (hipStream_t)stream);
}

int rs_glu_dwconv_silu_f32(rs_ctx* ctx, const float* x, const float* dw_w, const float* dw_b, const int32_t* lens, int B, int T,
                           int d, int k, float* out, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!x || !dw_w || !dw_b || !lens || !out) return rs_fail(ctx, RS_EINVAL, "glu_dwconv_f32: null pointer");
    return rs_launch_glu_dwconv_f32(ctx, x, dw_w, dw_b, lens, B, T, d, k, out, (hipStream_t)stream);
}

int rs_layernorm(rs_ctx* ctx, const float* x, const float* gamma, const float* beta, int M, int d, float eps,
                 uint16_t* out_bf16, float* out_f32, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!x || !gamma || !beta) return rs_fail(ctx, RS_EINVAL, "layernorm: null pointer");
    return rs_launch_layernorm(ctx, x, gamma, beta, M, d, eps, out_bf16, out_f32, (hipStream_t)stream);
}

int rs_relpos_attention(rs_ctx* ctx, const uint16_t* qkv, const uint16_t* pos, const float* bias_u, const float* bias_v,
                        const int32_t* lens, int B, int T, uint16_t* ctx_out, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!qkv || !pos || !bias_u || !bias_v || !lens || !ctx_out) return rs_fail(ctx, RS_EINVAL, "attention: null pointer");
    return rs_launch_attention(ctx, qkv, pos, bias_u, bias_v, lens, B, T, ctx_out, (hipStream_t)stream);
}

int rs_glu_dwconv_silu(rs_ctx* ctx, const uint16_t* x, const float* dw_w, const float* dw_b, const int32_t* lens, int B,
                       int T, int d, int k, uint16_t* out, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!x || !dw_w || !dw_b || !lens || !out) return rs_fail(ctx, RS_EINVAL, "glu_dwconv: null pointer");
    return rs_launch_glu_dwconv(ctx, x, RS_GLU_HALVES, dw_w, dw_b, lens, B, T, d, k, out, (hipStream_t)stream);
}

int rs_glu_dwconv_silu_layout(rs_ctx* ctx, const uint16_t* x, int layout, const float* dw_w, const float* dw_b,
                              const int32_t* lens, int B, int T, int d, int k, uint16_t* out, void* stream) {
    if (!ctx) return RS_EINVAL;
    if (!x || !dw_w || !dw_b || !lens || !out) return rs_fail(ctx, RS_EINVAL, "glu_dwconv: null pointer");
    return rs_launch_glu_dwconv(ctx, x, layout, dw_w, dw_b, lens, B, T, d, k, out, (hipStream_t)stream);
}


}  // extern "C"

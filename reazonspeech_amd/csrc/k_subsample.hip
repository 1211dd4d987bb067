// k_subsample.hip — depthwise-striding x8 subsampling, the non-GEMM parts
// ([UPSTREAM] ConvSubsampling(dw_striding) with length masking after
// every conv).  Activations are channels-last ([B][T][F][C], C fastest) so that the pointwise
// convs are plain row-major GEMMs (k_gemm_bf16.hip) and the flatten before the output Linear is
// a no-op (the Linear's columns are permuted to (f, c) order at weight-prep time).
//
//   sub_conv0_dw1 : mel f32[B][T][80] -> conv0(1->C, k3 s2 p1) + ReLU + mask -> depthwise(k3 s2 p1)
//                   + bias + mask -> bf16 [B][T2][F2][C].  The conv0 activation ([B][C][T/2][40],
//                   2.9 GB in bf16 at B=256) is never written: each thread (= one channel) keeps a
//                   rolling 3x3 window of conv0 outputs in registers and the mel rows in LDS.
//   sub_dw        : bf16 [B][Tin][Fin][C] -> depthwise(k3 s2 p1) + bias + mask -> bf16 [B][Tout][Fout][C]
//   enc_lens      : per-utterance valid lengths after each of the three strided convs.
#include "rs_common.h"

namespace {

__global__ void enc_lens_kernel(const int32_t* __restrict__ n_frames, int B, int stages, int sub_kind, int32_t* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = n_frames[b];
    for (int s = 0; s < stages; ++s) {
        n = n > 0 ? (sub_kind ? (n >= 3 ? (n - 3) / 2 + 1 : 0) : (n + 2 - 3) / 2 + 1) : 0;      // rs_conv_len
        out[s * B + b] = n;
    }
}

__device__ __forceinline__ void put(uint16_t* p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ void put(float* p, float v) { *p = v; }

// grid (T2, B), block C threads (thread = channel); OT = uint16_t (bf16, the throughput mode) or float (parity mode)
template <typename OT>
__global__ __launch_bounds__(256) void sub_conv0_dw1_kernel(
    const float* __restrict__ feats, const int32_t* __restrict__ lens_stage /* [stages][B] */, int B, int t_max,
    int n_mels, int T2, int F1, int F2, int C, const float* __restrict__ w0 /* [9][C] */,
    const float* __restrict__ b0, const float* __restrict__ wd /* [9][C] */, const float* __restrict__ bd,
    OT* __restrict__ out) {
    const int b = blockIdx.y, t2 = blockIdx.x, c = threadIdx.x;
    const int L1 = lens_stage[0 * B + b], L2 = lens_stage[1 * B + b];
    OT* orow = out + (((size_t)b * T2 + t2) * F2) * C + c;
    if (t2 >= L2) {  // masked output row
        for (int f2 = 0; f2 < F2; ++f2) put(orow + (size_t)f2 * C, 0.0f);
        return;
    }
    float k0[9], kd[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { k0[j] = w0[j * C + c]; kd[j] = wd[j * C + c]; }
    const float bias0 = b0[c], biasd = bd[c];
    const float* fb = feats + (size_t)b * t_max * n_mels;

    // The mel operand of every conv0 tap is the same for all 256 channels of the workgroup: its
    // address depends on (block, loop counter) only, so the row loads below are SCALAR loads
    // (s_load_dwordx4, 4 new mel bins of each of the 7 rows per output column) and the taps are v_fma
    // with an SGPR operand — no LDS staging and no LDS read per FMA (the LDS-broadcast version was
    // LDS-issue-bound at a quarter of the f32 VALU rate).  Zero padding costs nothing per tap: the
    // left mel bin -1 is the initial value of the carried column, out-of-range mel rows are read from
    // a clamped row with their taps' weights zeroed (kz), invalid conv0 rows are skipped (uniform).
    const float* rowp[7];
    bool rowok[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int tm = 4 * t2 - 3 + r;
        rowok[r] = tm >= 0 && tm < t_max;
        rowp[r] = fb + (size_t)(tm < 0 ? 0 : (tm >= t_max ? t_max - 1 : tm)) * n_mels;
    }
    float kz[3][9];
    bool a_ok[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int t1 = 2 * t2 - 1 + a;
        a_ok[a] = t1 >= 0 && t1 < L1;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) kz[a][i * 3 + j] = rowok[2 * a + i] ? k0[i * 3 + j] : 0.0f;
    }
    float carry[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) carry[r] = 0.0f;     // mel bin -1
    float left[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) left[a] = 0.0f;      // f1 = -1 is padding
    for (int f2 = 0; f2 < F2; ++f2) {
        float m[7][5];                               // mel bins 4*f2-1 .. 4*f2+3 of the 7 rows
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const float4 q = *reinterpret_cast<const float4*>(rowp[r] + 4 * f2);
            m[r][0] = carry[r]; m[r][1] = q.x; m[r][2] = q.y; m[r][3] = q.z; m[r][4] = q.w;
            carry[r] = q.w;
        }
        float mid[3], right[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float am = 0.0f, ar = 0.0f;
            if (a_ok[a]) {
                am = bias0; ar = bias0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        am = fmaf(kz[a][i * 3 + j], m[2 * a + i][j], am);          // f1 = 2*f2   : bins 4*f2-1 .. 4*f2+1
                        ar = fmaf(kz[a][i * 3 + j], m[2 * a + i][2 + j], ar);      // f1 = 2*f2+1 : bins 4*f2+1 .. 4*f2+3
                    }
                am = fmaxf(am, 0.0f); ar = fmaxf(ar, 0.0f);
            }
            mid[a] = am; right[a] = ar;
        }
        float acc = biasd;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc = fmaf(kd[a * 3 + 0], left[a], acc);
            acc = fmaf(kd[a * 3 + 1], mid[a], acc);
            acc = fmaf(kd[a * 3 + 2], right[a], acc);
        }
        put(orow + (size_t)f2 * C, acc);
#pragma unroll
        for (int a = 0; a < 3; ++a) left[a] = right[a];
    }
}

// grid (ceil(Fout / fpb), Tout, B); block 256 = (C/8 channel groups) x fpb
__global__ __launch_bounds__(256) void sub_dw_kernel(const uint16_t* __restrict__ in, const float* __restrict__ w /* [9][C] */,
                                                     const float* __restrict__ bias, const int32_t* __restrict__ lens_out,
                                                     int Tin, int Fin, int Tout, int Fout, int C,
                                                     uint16_t* __restrict__ out) {
    const int cgs = C >> 3;
    const int cg = threadIdx.x % cgs, fl = threadIdx.x / cgs, fpb = blockDim.x / cgs;
    const int fo = blockIdx.x * fpb + fl, to = blockIdx.y, b = blockIdx.z;
    if (fo >= Fout) return;
    const int c = cg * 8;
    uint16_t* op = out + (((size_t)b * Tout + to) * Fout + fo) * C + c;
    u16x8_t o;
    if (to >= lens_out[b]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0;
        *reinterpret_cast<u16x8_t*>(op) = o;
        return;
    }
    float acc[8];
    {
        const float4 b0 = reinterpret_cast<const float4*>(bias + c)[0], b1 = reinterpret_cast<const float4*>(bias + c)[1];
        acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ti = 2 * to - 1 + i;
        if (ti < 0 || ti >= Tin) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int fi = 2 * fo - 1 + j;
            if (fi < 0 || fi >= Fin) continue;
            const u16x8_t x = *reinterpret_cast<const u16x8_t*>(in + (((size_t)b * Tin + ti) * Fin + fi) * C + c);
            const float4 w0 = reinterpret_cast<const float4*>(w + (i * 3 + j) * C + c)[0];
            const float4 w1 = reinterpret_cast<const float4*>(w + (i * 3 + j) * C + c)[1];
            acc[0] = fmaf(w0.x, bf16_to_f32(x[0]), acc[0]); acc[1] = fmaf(w0.y, bf16_to_f32(x[1]), acc[1]);
            acc[2] = fmaf(w0.z, bf16_to_f32(x[2]), acc[2]); acc[3] = fmaf(w0.w, bf16_to_f32(x[3]), acc[3]);
            acc[4] = fmaf(w1.x, bf16_to_f32(x[4]), acc[4]); acc[5] = fmaf(w1.y, bf16_to_f32(x[5]), acc[5]);
            acc[6] = fmaf(w1.z, bf16_to_f32(x[6]), acc[6]); acc[7] = fmaf(w1.w, bf16_to_f32(x[7]), acc[7]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(acc[e]);
    *reinterpret_cast<u16x8_t*>(op) = o;
}

// the same depthwise conv on float32 activations (parity mode): one thread per output element, channel fastest
__global__ __launch_bounds__(256) void sub_dw_f32_kernel(const float* __restrict__ in, const float* __restrict__ w /* [9][C] */,
                                                         const float* __restrict__ bias, const int32_t* __restrict__ lens_out,
                                                         int Tin, int Fin, int Tout, int Fout, int C, float* __restrict__ out) {
    const int c = threadIdx.x, fo = blockIdx.x, to = blockIdx.y, b = blockIdx.z;
    float* op = out + (((size_t)b * Tout + to) * Fout + fo) * C + c;
    if (to >= lens_out[b]) { *op = 0.0f; return; }
    float acc = bias[c];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ti = 2 * to - 1 + i;
        if (ti < 0 || ti >= Tin) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int fi = 2 * fo - 1 + j;
            if (fi < 0 || fi >= Fin) continue;
            acc = fmaf(w[(i * 3 + j) * C + c], in[(((size_t)b * Tin + ti) * Fin + fi) * C + c], acc);
        }
    }
    *op = acc;
}

}  // namespace

int rs_launch_enc_lens(rs_ctx* ctx, const int32_t* n_frames, int B, int32_t* lens_out, hipStream_t s) {
    hipLaunchKernelGGL(enc_lens_kernel, dim3((B + 127) / 128), dim3(128), 0, s, n_frames, B, ctx->d.sub_stages, ctx->d.sub_kind, lens_out);
    RS_CHECK_LAUNCH(ctx, "enc_lens");
    return RS_OK;
}

int rs_launch_sub_conv0_dw1(rs_ctx* ctx, const float* feats, const int32_t* lens_stage, int B, int t_max, int T2,
                            int F2, uint16_t* out, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int C = d.sub_channels;
    if (C > 256 || (C % 64)) return rs_fail(ctx, RS_EINVAL, "subsampling: channels %d unsupported (<=256, %%64)", C);
    const int F1 = (d.n_mels + 2 - 3) / 2 + 1;
    if (d.n_mels % 4 || 4 * F2 > d.n_mels || 2 * F2 < F1)
        return rs_fail(ctx, RS_EINVAL, "subsampling: n_mels=%d must be a multiple of 4 with 4*F2 <= n_mels", d.n_mels);
    const size_t lds = 0;
    const double flops = (double)B * T2 * F2 * C * 2.0 * (6 * 9 + 9);
    const double bytes = (double)B * t_max * d.n_mels * 4.0 + (double)B * T2 * F2 * C * 2.0;
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, flops, bytes);
    hipLaunchKernelGGL(sub_conv0_dw1_kernel<uint16_t>, dim3(T2, B), dim3(C), lds, s, feats, lens_stage, B, t_max, d.n_mels, T2,
                       F1, F2, C, ctx->sub_conv0_w, ctx->sub_conv0_b, ctx->sub_dw_w[0], ctx->sub_dw_b[0], out);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "sub_conv0_dw1");
    return RS_OK;
}

int rs_launch_sub_dw(rs_ctx* ctx, const uint16_t* in, const float* w, const float* b, const int32_t* lens_out,
                     int stage, int B, int t_in, int f_in, int t_out, int f_out, uint16_t* out, hipStream_t s) {
    (void)stage;
    const int C = ctx->d.sub_channels;
    const int cgs = C / 8;
    const int fpb = 256 / cgs;
    const dim3 grid((f_out + fpb - 1) / fpb, t_out, B), block(cgs * fpb);
    const double bytes = (double)B * C * 2.0 * ((double)t_in * f_in + (double)t_out * f_out);
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, (double)B * t_out * f_out * C * 18.0, bytes);
    hipLaunchKernelGGL(sub_dw_kernel, grid, block, 0, s, in, w, b, lens_out, t_in, f_in, t_out, f_out, C, out);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "sub_dw");
    return RS_OK;
}

// ---- float32 parity mode (k_f32.hip): the same two operators storing / reading float32 activations ----------------
int rs_launch_sub_conv0_dw1_f32(rs_ctx* ctx, const float* feats, const int32_t* lens_stage, int B, int t_max, int T2, int F2,
                                float* out, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int C = d.sub_channels;
    if (C > 256 || (C % 64)) return rs_fail(ctx, RS_EINVAL, "subsampling: channels %d unsupported (<=256, %%64)", C);
    const int F1 = (d.n_mels + 2 - 3) / 2 + 1;
    if (d.n_mels % 4 || 4 * F2 > d.n_mels || 2 * F2 < F1)
        return rs_fail(ctx, RS_EINVAL, "subsampling: n_mels=%d must be a multiple of 4 with 4*F2 <= n_mels", d.n_mels);
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, (double)B * T2 * F2 * C * 2.0 * (6 * 9 + 9),
                  (double)B * t_max * d.n_mels * 4.0 + (double)B * T2 * F2 * C * 4.0);
    hipLaunchKernelGGL(sub_conv0_dw1_kernel<float>, dim3(T2, B), dim3(C), 0, s, feats, lens_stage, B, t_max, d.n_mels, T2, F1, F2, C,
                       ctx->sub_conv0_w, ctx->sub_conv0_b, ctx->sub_dw_w[0], ctx->sub_dw_b[0], out);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "sub_conv0_dw1_f32");
    return RS_OK;
}

int rs_launch_sub_dw_f32(rs_ctx* ctx, const float* in, const float* w, const float* b, const int32_t* lens_out, int B, int t_in,
                         int f_in, int t_out, int f_out, float* out, hipStream_t s) {
    const int C = ctx->d.sub_channels;
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, (double)B * t_out * f_out * C * 18.0,
                  (double)B * C * 4.0 * ((double)t_in * f_in + (double)t_out * f_out));
    hipLaunchKernelGGL(sub_dw_f32_kernel, dim3(f_out, t_out, B), dim3(C), 0, s, in, w, b, lens_out, t_in, f_in, t_out, f_out, C, out);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "sub_dw_f32");
    return RS_OK;
}

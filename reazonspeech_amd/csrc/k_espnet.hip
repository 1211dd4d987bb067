// k_espnet.hip — what the ESPnet2 Conformer-Transducer of reazonspeech.espnet.asr needs beyond the FastConformer kernels
// (reference: pkg/espnet-asr/src/transcribe.py:26-32, ctc.py:12-27).  The conformer blocks themselves
// run on the kernels of the NeMo path (same arithmetic: k_gemm_bf16 / k_layernorm / k_attention with head_dim 64 / the
// generic depthwise kernel for k = 31); new here:
//
//   sub2d_conv0_kernel   [UPSTREAM] Conv2dSubsampling.conv[0..1]: Conv2d(1, C, 3, stride 2, NO padding) + ReLU on the
//                        log-mel map -> bf16 channels-last [Bc][T1][F1][C]  (f32 VALU, 9 taps per output)
//   im2col3x3s2_kernel   the patches of Conv2d(C, C, 3, 2) gathered row-major [Bc*T2*F2][9*C], K ordered (kernel row, kernel
//                        column, channel): the dense conv then is ONE k_gemm_bf16 launch (M x C x 9C, bias + ReLU + row
//                        mask in its epilogue).  Pure 16-byte copies, HBM-bound, one workgroup per output position.  The encoder runs the two kernels and
//                        the GEMM over chunks of utterances so that the patch matrix stays around 1 GB.
//   ctc_softmax_kernel   [UPSTREAM] CTC.softmax: row softmax of the ctc_lo logits in place (probabilities, not logs: what the
//                        reference hands to its blank finder and to ctc_segmentation), plus the blank column on its own
#include "rs_common.h"

namespace {

__device__ __forceinline__ void store_act(uint16_t* p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ void store_act(float* p, float v) { *p = v; }

// grid (T1, Bc), block 256: a thread owns channels c = tid, tid + 256, ...; the three mel rows of this output row sit in LDS
// (OutT = uint16_t: bf16 for the throughput mode; float: the float32 parity mode)
template <typename OutT>
__global__ __launch_bounds__(256) void sub2d_conv0_kernel(const float* __restrict__ feats, const int32_t* __restrict__ lens1 /* [B] valid T1 rows */,
                                                          int b0, int t_max, int n_mels, int T1, int F1, int C,
                                                          const float* __restrict__ w0 /* [9][C] */, const float* __restrict__ bias,
                                                          OutT* __restrict__ out) {
    __shared__ float rows[3][132];
    const int t1 = blockIdx.x, bl = blockIdx.y, b = b0 + bl;
    OutT* orow = out + (((size_t)bl * T1 + t1) * F1) * C;
    if (t1 >= lens1[b]) {                           // rows past the utterance: zeros (never read by a valid output row)
        for (int i = threadIdx.x; i < F1 * C; i += 256) orow[i] = 0;
        return;
    }
    for (int i = threadIdx.x; i < 3 * n_mels; i += 256) {
        const int r = i / n_mels, m = i - r * n_mels;
        const int t = 2 * t1 + r;
        rows[r][m] = t < t_max ? feats[((size_t)b * t_max + t) * n_mels + m] : 0.0f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float k[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = w0[j * C + c];
        const float bb = bias[c];
        for (int f1 = 0; f1 < F1; ++f1) {
            float acc = bb;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc = fmaf(k[i * 3 + j], rows[i][2 * f1 + j], acc);
            store_act(orow + (size_t)f1 * C + c, fmaxf(acc, 0.0f));
        }
    }
}

// one workgroup per output position (t2, f2): the 9 taps x C channels of its patch are 9 contiguous runs of C bf16 in the input
// and ONE contiguous run of 9 C in the output; a thread moves 16 bytes at a time (9 C / 8 = 576 moves at C = 512).
// (One 64-thread workgroup per tap — 1 KB each, 766k workgroups per chunk — was dispatch-bound: 417 us per chunk for 1.2 GB.)
// grid (F2, Bc * T2), block 256
// (element-size generic: `c8` = 16-byte pieces per C channels — C / 8 for bf16, C / 4 for the float32 parity mode)
__global__ __launch_bounds__(256) void im2col3x3s2_kernel(const uint4* __restrict__ in /* [Bc][T1][F1][c8] */, int T1, int F1,
                                                          int T2, int F2, int c8, uint4* __restrict__ out /* [Bc*T2*F2][9*c8] */) {
    const int f2 = blockIdx.x, bt = blockIdx.y;
    const int bl = bt / T2, t2 = bt - bl * T2;
    uint4* dst = out + ((size_t)bt * F2 + f2) * 9 * c8;
    const uint4* base = in + (((size_t)bl * T1 + 2 * t2) * F1 + 2 * f2) * c8;
    for (int q = threadIdx.x; q < 9 * c8; q += 256) {
        const int tap = q / c8, c = q - tap * c8;
        const int i = tap / 3, j = tap - 3 * i;
        dst[q] = base[((size_t)i * F1 + j) * c8 + c];
    }
}

// row softmax in place, one wave per row; blank_out[row] = p[row][blank] (may be null)
__global__ __launch_bounds__(256) void ctc_softmax_kernel(float* __restrict__ z, int M, int V, int ld, int blank, float* __restrict__ blank_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float* p = z + (size_t)row * ld;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, p[v]);
    mx = wave_max(mx);
    float s = 0.0f;
    for (int v = lane; v < V; v += 64) s += expf(p[v] - mx);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int v = lane; v < V; v += 64) {
        const float q = expf(p[v] - mx) * inv;
        p[v] = q;
        if (v == blank && blank_out) blank_out[row] = q;
    }
    for (int v = V + lane; v < ld; v += 64) p[v] = 0.0f;       // the columns that pad the vocabulary to a multiple of 4
}

}  // namespace

int rs_launch_sub2d_conv0(rs_ctx* ctx, const float* feats, const int32_t* lens1, int b0, int Bc, int t_max, int T1, int F1,
                          uint16_t* out, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int C = d.sub_channels;
    if (d.n_mels > 128 || 2 * (F1 - 1) + 2 >= d.n_mels + 1) return rs_fail(ctx, RS_EINVAL, "conv2d subsampling: n_mels %d / F1 %d", d.n_mels, F1);
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, (double)Bc * T1 * F1 * C * 18.0, (double)Bc * (t_max * d.n_mels * 4.0 + (double)T1 * F1 * C * 2.0));
    hipLaunchKernelGGL(sub2d_conv0_kernel<uint16_t>, dim3(T1, Bc), dim3(256), 0, s, feats, lens1, b0, t_max, d.n_mels, T1, F1, C,
                       ctx->sub_conv0_w, ctx->sub_conv0_b, out);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "sub2d_conv0");
    return RS_OK;
}

int rs_launch_sub2d_conv0_f32(rs_ctx* ctx, const float* feats, const int32_t* lens1, int b0, int Bc, int t_max, int T1, int F1,
                              float* out, hipStream_t s) {
    const rs_dims& d = ctx->d;
    const int C = d.sub_channels;
    if (d.n_mels > 128 || 2 * (F1 - 1) + 2 >= d.n_mels + 1) return rs_fail(ctx, RS_EINVAL, "conv2d subsampling: n_mels %d / F1 %d", d.n_mels, F1);
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, (double)Bc * T1 * F1 * C * 18.0, (double)Bc * (t_max * d.n_mels * 4.0 + (double)T1 * F1 * C * 4.0));
    hipLaunchKernelGGL(sub2d_conv0_kernel<float>, dim3(T1, Bc), dim3(256), 0, s, feats, lens1, b0, t_max, d.n_mels, T1, F1, C,
                       ctx->sub_conv0_w, ctx->sub_conv0_b, out);
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "sub2d_conv0_f32");
    return RS_OK;
}

int rs_launch_im2col3x3s2(rs_ctx* ctx, const uint16_t* in, int Bc, int T1, int F1, int T2, int F2, uint16_t* out, hipStream_t s) {
    const int C = ctx->d.sub_channels;
    if (C % 8) return rs_fail(ctx, RS_EINVAL, "im2col: channels %d must be a multiple of 8", C);
    if ((long long)Bc * T2 > 65535LL) return rs_fail(ctx, RS_EINVAL, "im2col: chunk of %d x %d rows exceeds the grid", Bc, T2);
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, 0.0, (double)Bc * T2 * F2 * 9.0 * C * 2.0 * 2.0);
    hipLaunchKernelGGL(im2col3x3s2_kernel, dim3(F2, Bc * T2), dim3(256), 0, s, reinterpret_cast<const uint4*>(in), T1, F1, T2, F2, C / 8,
                       reinterpret_cast<uint4*>(out));
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "im2col3x3s2");
    return RS_OK;
}

int rs_launch_im2col3x3s2_f32(rs_ctx* ctx, const float* in, int Bc, int T1, int F1, int T2, int F2, float* out, hipStream_t s) {
    const int C = ctx->d.sub_channels;
    if (C % 4) return rs_fail(ctx, RS_EINVAL, "im2col: channels %d must be a multiple of 4", C);
    if ((long long)Bc * T2 > 65535LL) return rs_fail(ctx, RS_EINVAL, "im2col: chunk of %d x %d rows exceeds the grid", Bc, T2);
    rs_prof_begin(ctx, RS_PROF_SUBSAMPLE, s, 0.0, (double)Bc * T2 * F2 * 9.0 * C * 4.0 * 2.0);
    hipLaunchKernelGGL(im2col3x3s2_kernel, dim3(F2, Bc * T2), dim3(256), 0, s, reinterpret_cast<const uint4*>(in), T1, F1, T2, F2, C / 4,
                       reinterpret_cast<uint4*>(out));
    rs_prof_end(ctx, RS_PROF_SUBSAMPLE, s);
    RS_CHECK_LAUNCH(ctx, "im2col3x3s2_f32");
    return RS_OK;
}

int rs_launch_ctc_softmax(rs_ctx* ctx, float* logits, int M, int V, int ld, int blank, float* blank_out, hipStream_t s) {
    if (M <= 0) return RS_OK;
    rs_prof_begin(ctx, RS_PROF_ELEMENTWISE, s, 4.0 * M * V, 8.0 * M * V);
    hipLaunchKernelGGL(ctc_softmax_kernel, dim3((M + 3) / 4), dim3(256), 0, s, logits, M, V, ld, blank, blank_out);
    rs_prof_end(ctx, RS_PROF_ELEMENTWISE, s);
    RS_CHECK_LAUNCH(ctx, "ctc_softmax");
    return RS_OK;
}

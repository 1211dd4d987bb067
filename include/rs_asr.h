/*
 * rs_asr.h — C ABI of librs_asr.so, the MI355X (gfx950) FastConformer-RNNT inference path.
 *
 * The reference (reazon-research/ReazonSpeech) has NO FFI / plugin boundary for this path:
 * its only interface is the Python pair load_model()/transcribe()
 * (pkg/nemo-asr/src/transcribe.py:9-28, :30-60) which hands everything to NeMo at two call
 * sites (EncDecRNNTBPEModel.from_pretrained :26-28, model.transcribe :48-53).  This header is
 * the boundary introduced *underneath* that pair: each entry point below names
 * the reference-side step it replaces.  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types.
 *   - The CALLER owns every device buffer (weights, activations, workspace); the library never
 *     allocates or frees device memory.  All pointers are device pointers unless named host_*.
 *   - Every function returns RS_OK (0) or a negative RS_E* code; rs_last_error() returns a
 *     message for the last failure on that context (thread-compatible, not re-entrant: one
 *     context per stream).  No exit/abort, no C++ exception crosses the ABI.
 *   - Work is enqueued asynchronously on the given hipStream_t (passed as void*; NULL = the
 *     default stream).  Nothing synchronises unless stated.
 *   - bf16 tensors are raw uint16 bit patterns (round-to-nearest-even of the f32 value).
 */
#ifndef RS_ASR_H
#define RS_ASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_ABI_VERSION 6

enum {
    RS_OK = 0,
    RS_EINVAL = -1,      /* bad argument / shape */
    RS_EMISSING = -2,    /* a required weight tensor was not registered */
    RS_EWORKSPACE = -3,  /* workspace too small */
    RS_EHIP = -4,        /* a HIP runtime call failed */
    RS_EOVERFLOW = -5,   /* decode output buffer (u_max) too small for some utterance */
    RS_ESTATE = -6       /* call order violated (e.g. forward before rs_finalize) */
};

typedef struct rs_ctx rs_ctx;

/* Model dimensions: what NeMo reads from model_config.yaml inside the .nemo checkpoint
 * (reference: pkg/nemo-asr/src/transcribe.py:26-28 [UPSTREAM]). */
typedef struct rs_dims {
    int32_t n_mels;        /* 80 */
    int32_t n_fft;         /* 512 (only 512 is built) */
    int32_t win_length;    /* 400 */
    int32_t hop_length;    /* 160 */
    float preemph;         /* 0.97, 0 disables */
    float log_guard;       /* 2^-24 */
    float norm_eps;        /* 1e-5, added to the per-feature std */
    int32_t d_model;       /* 1024 */
    int32_t n_heads;       /* 8 (head_dim must be 128) */
    int32_t ff_dim;        /* 4096 */
    int32_t n_layers;      /* 24 */
    int32_t conv_kernel;   /* 9 */
    int32_t sub_channels;  /* 256 */
    int32_t sub_stages;    /* 3 (x8) */
    int32_t xscaling;      /* 1: multiply the subsampling output by sqrt(d_model) */
    float ln_eps;          /* 1e-5 */
    int32_t att_left;      /* -1 = unlimited */
    int32_t att_right;     /* -1 = unlimited */
    int32_t n_global;      /* global tokens of the local-attention variant (0 = none) */
    int32_t n_logits;      /* vocab + 1 (3001) */
    int32_t blank_id;      /* 3000 */
    int32_t pred_hidden;   /* 640 */
    int32_t pred_layers;   /* 2 */
    int32_t joint_hidden;  /* 640 */
    int32_t max_symbols;   /* 10 */
    /* ---- model family switches (ABI 3).  All zero = NeMo FastConformer-RNNT, the path everything above describes.  The
     * ESPnet2 Conformer-Transducer of reazonspeech.espnet.asr (pkg/espnet-asr/src/transcribe.py:26-32) sets all five. ---- */
    int32_t frontend_kind; /* 0: NeMo AudioToMelSpectrogramPreprocessor.  1: ESPnet DefaultFrontend + GlobalMVN — reflect edge
                              padding, 1 + L / hop frames, log(max(x, log_guard)), (x - "fe.mvn_mean") * "fe.mvn_istd"; preemph
                              must be 0 */
    int32_t sub_kind;      /* 0: dw_striding x 2^sub_stages.  1: ESPnet Conv2dSubsampling x4 — Conv2d(1, C, 3, 2) ReLU
                              Conv2d(C, C, 3, 2) ReLU without padding ("sub.conv0.*" f32 tap-major, "sub.conv1.w" bf16 [C][9C] with K
                              ordered (kernel row, kernel column, channel)), Linear(C * F2, d_model); sub_stages must be 2 */
    int32_t final_norm;    /* 1: a LayerNorm after the last block ("final_norm.g" / ".b"; ESPnet encoder.after_norm) */
    int32_t joint_act;     /* joint activation: 0 = ReLU (NeMo), 1 = tanh (ESPnet JointNetwork); tanh decodes with the exact
                              (un-screened) joint kernels */
    int32_t ctc_vocab;     /* > 0: a CTC head Linear(d_model, ctc_vocab) is registered with its rows padded to Vp = the next
                              multiple of 4 ("ctc.w" bf16 [Vp][d_model], "ctc.b" f32 [Vp]; the padding rows are never read back:
                              the softmax runs over ctc_vocab columns and writes 0 to the rest); see rs_encoder_set_ctc_out */
} rs_dims;

/* Dimensions of the Zipformer2 transducer of reazonspeech.k2.asr: what sherpa-onnx reads out of the three ONNX files the
 * reference hands it (pkg/k2-asr/src/huggingface.py:73-83; [UPSTREAM] icefall zipformer recipe: --num-encoder-layers,
 * --feedforward-dim, --num-heads, --encoder-dim, --cnn-module-kernel, --downsampling-factor, --query-head-dim, ...). */
typedef struct rs_k2_dims {
    int32_t n_mels;            /* 80 (feature_dim, huggingface.py:80) */
    int32_t frame_length;      /* 400 samples */
    int32_t frame_shift;       /* 160 samples */
    float preemph;             /* 0.97 (kaldi-native-fbank default) */
    int32_t embed_c1, embed_c2, embed_c3;   /* encoder_embed conv channels: 8, 32, 128 */
    int32_t n_stacks;          /* 6 */
    int32_t encoder_dim[8];    /* 192,256,512,768,512,256 */
    int32_t num_layers[8];     /* 2,2,4,5,4,2 */
    int32_t ff_dim[8];         /* 512,768,1536,2048,1536,768 */
    int32_t num_heads[8];      /* 4,4,4,8,4,4 */
    int32_t cnn_kernel[8];     /* 31,31,15,15,15,31 (7, 15 and 31 are built) */
    int32_t downsampling[8];   /* 1,2,4,8,4,2 */
    int32_t query_head_dim;    /* 32 */
    int32_t value_head_dim;    /* 12 */
    int32_t pos_head_dim;      /* 4 */
    int32_t pos_dim;           /* 48 (host side only: the position tables arrive projected) */
    int32_t vocab_size;        /* lines of tokens.txt */
    int32_t decoder_dim;       /* 512 */
    int32_t joiner_dim;        /* 512 */
    int32_t context_size;      /* 2 */
    int32_t blank_id;          /* 0 */
    int32_t unk_id;            /* id of "<unk>", -1 = none: sherpa-onnx's greedy search does not emit it */
} rs_k2_dims;

/* ---- context ------------------------------------------------------------------------- */

/* Replaces: model object construction inside EncDecRNNTBPEModel.from_pretrained
 * (pkg/nemo-asr/src/transcribe.py:26-28).  `device` is the HIP device ordinal. */
int rs_create(rs_ctx** out, int device, const rs_dims* dims);
/* The same for reazonspeech.k2.asr — replaces: sherpa_onnx.OfflineRecognizer.from_transducer(encoder, decoder, joiner, ...)
 * (pkg/k2-asr/src/huggingface.py:73-83).  The context answers the SAME stage entry points: rs_set_tensor / rs_finalize,
 * rs_workspace_bytes, rs_mel_frames ((n + shift / 2) / shift kaldi-style frames), rs_enc_frames (((T - 7) / 2 + 1) / 2),
 * rs_frontend_logmel (kaldi-native-fbank features: snip_edges false with reflected edges, per-frame DC removal and
 * pre-emphasis, povey window, log(max(mel energy, FLT_EPSILON)), no normalisation), rs_encoder_forward (encoder_embed +
 * Zipformer2 stacks + joiner.encoder_proj, csrc/k_zipformer.hip) and rs_rnnt_greedy (stateless decoder + tanh joiner, one
 * symbol per frame, `<unk>` treated as blank: sherpa-onnx OfflineTransducerGreedySearchDecoder).  Tensor names: DESIGN.md
 * "reazonspeech.k2.asr". */
int rs_k2_create(rs_ctx** out, int device, const rs_k2_dims* dims);
/* Parity taps of a Zipformer context (tests only): embed_out f32 [B*T3][encoder_dim[0]] (encoder_embed's output) and stack_out,
 * the stacks' outputs f32 [B*T3][encoder_dim[s]] one after the other; NULL disables. */
int rs_k2_encoder_set_taps(rs_ctx* ctx, float* embed_out, float* stack_out);

/* ---- reazonspeech.avsr: the AV-HuBERT encoder-decoder (csrc/k_avsr.hip) ---------------------------------------------------------
 * The reference is in-tree torch code: AVHubertForConditionalGeneration (pkg/avsr/src/avhubert/modeling_avhubert.py:216-391) over
 * AVHubertModel (:119-213: audio Linear, Conv3d + ResNet-18 video front-end modeling_resnet.py:140-178, concatenation, LayerNorm,
 * post_extract_proj, transformers' HubertEncoder) and AVHubertDecoder (decoder.py:467-617).  Everything is float32, as in the
 * reference.  Tensor names and layouts: reazonspeech_amd/runtime/avsr_weights.py: prepare_weights_avsr. */
typedef struct rs_avsr_dims {
    int32_t encoder_layers;        /* 12  (configuration_avhubert.py:9-13) */
    int32_t encoder_embed_dim;     /* 768 */
    int32_t encoder_ffn_dim;       /* 3072 */
    int32_t encoder_heads;         /* 12 */
    int32_t conv_pos;              /* 128: kernel of the positional convolution (even) */
    int32_t conv_pos_groups;       /* 16 */
    int32_t audio_feat_dim;        /* 104 = 4 stacked 26-dim log filterbank frames (feature_extraction_avhubert.py:120-137) */
    int32_t fuse_concat;           /* 1: modality_fuse "concat" (the only form built) */
    int32_t image_size;            /* 88 (feature_extraction_avhubert.py:24) */
    int32_t decoder_layers;        /* 6 */
    int32_t decoder_embed_dim;     /* 768 (== encoder_embed_dim) */
    int32_t decoder_ffn_dim;       /* 3072 */
    int32_t decoder_heads;         /* 4 */
    int32_t max_positions;         /* 2048 rows of the sinusoidal table */
    int32_t vocab_size;
    float layer_norm_eps;          /* 1e-5 */
} rs_avsr_dims;
/* Replaces: AVHubertForConditionalGeneration.__init__ / from_pretrained (modeling_avhubert.py:216-254).  The context takes
 * rs_set_tensor / rs_finalize / rs_destroy / rs_last_error and the rs_avsr_* stage functions below (the transducer stages do not
 * apply to it). */
int rs_avsr_create(rs_ctx** out, int device, const rs_avsr_dims* dims);
size_t rs_avsr_workspace_bytes(const rs_ctx* ctx, int B, int T);
/* Replaces: AVHubertModel.forward (modeling_avhubert.py:162-213).
 *   input_values f32[B][T][audio_feat_dim], pixel_values f32[B][T][image_size][image_size] (the reference's [B][T][1][H][W]),
 *   padding_mask f32[B][T] (nonzero = padding frame; the reference's padding_mask, :192-195) -> enc_out f32[B][T][encoder_embed_dim]
 *   = last_hidden_state.  Padded frames go through the front-ends like any other (as in the reference); their encoder inputs are
 *   zeroed and they are masked as attention keys. */
int rs_avsr_encoder_forward(rs_ctx* ctx, const float* input_values, const float* pixel_values, const float* padding_mask, int B,
                            int T, float* enc_out, void* workspace, size_t workspace_bytes, void* stream);
/* Parity taps (tests only; NULL disables): video = feature_extractor_video output f32[B*T][d], fused_ln = avhubert.layer_norm output
 * f32[B*T][2d], enc_ln = encoder.layer_norm output f32[B*T][d], layer_out = the listed encoder layers' outputs one after the other. */
int rs_avsr_encoder_set_taps(rs_ctx* ctx, float* video, float* fused_ln, float* enc_ln, float* layer_out, const int32_t* layer_ids,
                             int n_layer_ids);
/* Replaces: the decoder half of AVHubertForConditionalGeneration.forward inside generate() (modeling_avhubert.py:283-298,
 * decoder.py:488-617), one token per call with a KV cache (the reference re-feeds the prefix and re-runs the encoder every step,
 * :372-391; same results).  rows = B * beams hypothesis rows, row = clip * beams + beam (transformers' flattening).
 *   rs_avsr_decoder_begin   cross-attention keys / values of every layer from enc f32[B][T][d]; call once per batch
 *   rs_avsr_decoder_step    tokens i32[rows] = the token at position `step` of every row (step 0: the bos prompt); src_rows i32[rows]
 *                           or NULL = the row whose prefix each row continues (beam search re-parenting; give it at EVERY step >= 1
 *                           or at none); padding_mask as above; -> logits f32[rows][vocab_size rounded up to 4] of position `step`
 * `state` is caller-owned device memory of rs_avsr_decoder_state_bytes(B, T, beams, max_len) bytes, untouched between calls. */
size_t rs_avsr_decoder_state_bytes(const rs_ctx* ctx, int B, int T, int beams, int max_len);
int rs_avsr_decoder_begin(rs_ctx* ctx, const float* enc, int B, int T, int beams, int max_len, void* state, size_t state_bytes,
                          void* stream);
int rs_avsr_decoder_step(rs_ctx* ctx, const int32_t* tokens, const int32_t* src_rows, int step, const float* padding_mask, int B,
                         int T, int beams, int max_len, float* logits, void* state, size_t state_bytes, void* stream);
void rs_destroy(rs_ctx* ctx);
const char* rs_last_error(const rs_ctx* ctx);
int rs_abi_version(void);

/* Register one prepared weight tensor by name (device pointer, caller-owned, must outlive the
 * context).  Names and layouts: DESIGN.md §"Weights in HBM".  Replaces: load_state_dict inside
 * from_pretrained (transcribe.py:26-28).
 * Optional derived tensors "L{i}.att.pos_proj" (bf16, same shape as "pos.table"): the position
 * table already multiplied by layer i's linear_pos weight (one rs_gemm_bf16 call per layer at load
 * time); when registered, rs_encoder_forward skips that projection in every call.
 * "L{i}.conv.pw1.w" / ".b" (the conv module's first pointwise convolution, 2*d_model output rows) are registered
 * with their rows interleaved in blocks of 32: rows 64j .. 64j+31 are the value rows 32j .. 32j+31 of NeMo's
 * pointwise_conv1, rows 64j+32 .. 64j+63 the matching gate rows d_model + 32j ..: a GLU pair then sits in one
 * MFMA lane of the GEMM and is applied in its epilogue (RS_GEMM_GLU). */
int rs_set_tensor(rs_ctx* ctx, const char* name, const void* dev_ptr, size_t nbytes);
/* Check that every tensor the dims require is present; must precede any forward call. */
int rs_finalize(rs_ctx* ctx);

/* Bytes of scratch the three stage functions need for a batch of B utterances whose padded
 * sample count is at most max_samples (pad included). */
size_t rs_workspace_bytes(const rs_ctx* ctx, int B, int max_samples);

/* Shape helpers (host arithmetic, no device work). */
int rs_mel_frames(const rs_ctx* ctx, int n_samples);   /* floor(L / hop) */
int rs_enc_frames(const rs_ctx* ctx, int n_mel_frames); /* three k3 s2 p1 convs */

/* ---- host staging (A4: `torch.from_numpy` + list wrap, transcribe.py:46-50, for a whole batch) -------------------
 * Pure host function, no device work: gathers `n_rows` utterances (host float32, 16 kHz mono, un-padded) into the
 * caller's pinned staging matrix dst[total_rows][dst_pitch], zero-filling every row from its length up to `width`
 * (the batch's padded extent) and writing the lengths (0 for rows n_rows .. total_rows - 1: a short last batch).
 * One call per batch instead of one interpreter-level copy per utterance: a Python binding releases its interpreter lock
 * for the call, so the staging of batch i+2 does not contend with host post-processing of batch i. */
int rs_host_stage_rows(float* dst, size_t dst_pitch, int width, const float* const* rows, const int32_t* lens,
                       int n_rows, int total_rows, int32_t* dst_lens);

/* ---- stage 1: log-mel front-end -------------------------------------------------------
 * Replaces: pad_audio (pkg/nemo-asr/src/audio.py:70-83, folded into the load: the kernel
 * reads `audio[b][i - pad_left]` and treats everything outside [0, lens[b]) as 0) and NeMo's
 * AudioToMelSpectrogramPreprocessor reached through model.transcribe (transcribe.py:48-53):
 * pre-emphasis, STFT(512, hann 400, hop 160, centre, zero pad), power, Slaney mel 80,
 * log(x + 2^-24), per-feature normalisation over the valid frames, zeroed padding.
 *
 *   audio   f32[B][audio_stride]   raw (un-padded) samples
 *   lens    i32[B]                  valid samples per utterance (un-padded)
 *   feats   f32[B][t_max][n_mels]   t_max = rs_mel_frames(max(lens) + pad_left + pad_right)
 *   n_frames i32[B]                 valid frames per utterance (written)
 */
int rs_frontend_logmel(rs_ctx* ctx, const float* audio, const int32_t* lens, int B,
                       int audio_stride, int pad_left, int pad_right, int t_max,
                       float* feats, int32_t* n_frames, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- stage 2: FastConformer encoder + joint encoder projection --------------------------
 * Replaces: ConformerEncoder.forward and RNNTJoint.enc inside model.transcribe
 * (transcribe.py:48-53): dw-striding x8 subsampling, 24 x [1/2 FFN, rel-pos MHSA, conv
 * module, 1/2 FFN, LayerNorm], Linear d_model -> joint_hidden.
 *
 *   feats, n_frames      as produced by rs_frontend_logmel
 *   enc_out  f32[B][tp_max][d_model]      (may be NULL if not wanted) tp_max = rs_enc_frames(t_max)
 *   joint_enc f32[B][tp_max][joint_hidden]
 *   enc_lens i32[B]
 */
int rs_encoder_forward(rs_ctx* ctx, const float* feats, const int32_t* n_frames, int B, int t_max,
                       float* enc_out, float* joint_enc, int32_t* enc_lens,
                       void* workspace, size_t workspace_bytes, void* stream);

/* A HIP stream restricted to a set of compute units (bit i of cu_mask = CU i may run this stream's workgroups;
 * hipExtStreamCreateWithCUMask), or, with cu_mask == NULL, a plain non-blocking stream of the given priority.
 * The two-stage pipeline can confine the latency-bound decode loop to a slice of the chip so that its many small
 * launches stop delaying the encoder GEMMs' tile rounds on the other CUs.  No reference counterpart. */
int rs_stream_create(void** stream_out, int device, const uint32_t* cu_mask, int n_words, int priority);
int rs_stream_destroy(void* stream);

/* Scheduling options of a context (no reference counterpart).
 *   "fuse_glu"           1 (default): the conv module's GLU is applied to the float32 accumulators of the pw1 GEMM in its
 *                        epilogue, for EVERY batch size (one rounding point: an utterance's arithmetic does not depend on
 *                        the batch it rides in); 0 = the plain pw1 product is stored and the depthwise kernel applies the
 *                        GLU (moves one bf16 rounding; A/B and layout tests; $RS_FUSE_GLU).
 *   "defer_out_norm"     1 (default): the float32 rows of a layer's output LayerNorm are not stored — their one reader, the
 *                        residual operand of the next layer's first FFN, normalises the un-normed rows in its GEMM epilogue
 *                        from per-row (mean, rstd); layers with a parity tap, and the last layer, store them.  0 = every
 *                        output norm stores its rows.  Bit-identical results either way ($RS_DEFER_OUT_NORM).
 *   "decode_screen"      1 (default when the tensors joint.out.w16 / .wrm / .bpad / .wmax are registered): the joint's
 *                        output layer runs as a bf16 screening GEMM followed by an exact float32 evaluation of every
 *                        column that can still be the argmax (bit-identical result); 0 = every column in exact float32.
 *   "decode_narrow"      1 (default): LSTM / prediction-projection kernels with one 16-column tile per workgroup (4x the
 *                        workgroups, a quarter of the per-launch latency on an idle chip); 0 = the wide-tile kernels, which
 *                        need fewer free CUs per launch and do better next to the encoder GEMMs of the two-stage pipeline.
 *   "precision_f32"      0 (default): the throughput mode — bf16 GEMM operands and stored activations, float32 accumulation,
 *                        float32 residual stream.  1: the PARITY mode — rs_encoder_forward runs with float32 weights,
 *                        activations and arithmetic end to end (exact-f32 matrix-core GEMMs, IEEE exp / divide), which is what
 *                        the reference computes (pkg/nemo-asr/src/transcribe.py:26-28, :48-53: float32, no autocast).  Needs the
 *                        "<name>.f32" tensors: float32 copies of every bf16 GEMM weight (same layout; "L{i}.conv.pw1.w.f32" /
 *                        ".b.f32" in NeMo's own row order, values then gates) and "pos.table.f32"; rs_workspace_bytes
 *                        accounts for the mode once they are registered.  ~20x slower; front-end and decode are float32 in
 *                        both modes.
 *   "gemm_f32_x3"        (default 0) 1: the float32 products of rs_gemm_f32-class launches (the "precision_f32" encoder, the AV-HuBERT
 *                        encoder) are formed from three bf16 matrix-core terms — hi / lo split of both operands, hi.hi + hi.lo + lo.hi,
 *                        float32 accumulation: 16 mantissa bits per operand — instead of the exact v_mfma_f32_16x16x4_f32 chain:
 *                        ~2x faster.  NOT an IEEE float32 chain: a mode of its own (Python: precision="fp32x3", products="x3"),
 *                        held to the same goldens as the exact mode (tests/test_gpu_*fp32*.py, tests/test_gpu_avsr.py).
 *   "k2_cnx_fused", "k2_conv2_fused"   (Zipformer contexts; default 1) the encoder_embed's ConvNeXt pointwise pair as one kernel /
 *                        its 32 -> 128 convolution with the patches gathered into LDS; 0 = the GEMM launches they replace.  Both
 *                        forms give the same bits (tests/test_gpu_k2.py): the switch exists for that comparison. */
int rs_set_option(rs_ctx* ctx, const char* key, int value);

/* Parity taps (tests only; no reference counterpart — NeMo exposes intermediate activations through
 * forward hooks): when set, the next rs_encoder_forward calls also copy the f32 residual stream
 * [B*tp_max][d_model] after the subsampling block (sub_out) and after each listed
 * conformer layer (layer_out[k] for layer_ids[k]; rows L1-L7).  NULL / 0 disables.  layer_ids is a host array. */
int rs_encoder_set_taps(rs_ctx* ctx, float* sub_out, float* layer_out, const int32_t* layer_ids,
                        int n_layer_ids);

/* CTC posteriors (ESPnet family; replaces model.asr_model.ctc.softmax(model.asr_model.encode(...)), pkg/espnet-asr/src/ctc.py:12-27):
 * when set, the next rs_encoder_forward calls also write softmax(ctc_lo(encoder output)) — probabilities, not logarithms, as the
 * reference's blank finder (ctc.py:29-58) and its ctc_segmentation call (ctc.py:60-75) consume them — to probs f32
 * [B*tp_max][Vp] (Vp = ctc_vocab rounded up to a multiple of 4: the row pitch; columns >= ctc_vocab are 0) and / or only the
 * blank column to blank_prob f32 [B*tp_max].  Either may be NULL; both NULL disables. */
int rs_encoder_set_ctc_out(rs_ctx* ctx, float* probs, float* blank_prob);

/* ---- stage 3: RNN-T greedy decode -------------------------------------------------------
 * Replaces: decoding.rnnt_decoder_predictions_tensor inside model.transcribe
 * (transcribe.py:48-53) — prediction network (Embedding + LSTM), joint (ReLU + Linear),
 * argmax, max_symbols loop — in its batched-greedy form (the north-star's decode strategy;
 * the reference post-processing expects ALSD-shaped output, see Hypothesis.from_greedy).
 *
 *   joint_enc f32[B][tp_max][joint_hidden], enc_lens i32[B]
 *   ids     i32[B][u_max]   emitted token ids
 *   frames  i32[B][u_max]   encoder frame index each token was emitted at
 *   n_ids   i32[B]
 * Synchronises the stream internally (the trip count is data dependent).  Returns
 * RS_EOVERFLOW if an utterance would emit more than u_max tokens.
 */
int rs_rnnt_greedy(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max,
                   int u_max, int32_t* ids, int32_t* frames, int32_t* n_ids,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- stage 3b: RNN-T alignment-length synchronous beam search (ALSD) ----------------------
 * Replaces: the same call as rs_rnnt_greedy (transcribe.py:48-53) when the checkpoint's decoding
 * strategy is "alsd" — what the reference's post-processing was written for (decode.py:29,38-41:
 * "NeMo prepends a blank token to y_sequence with ALSD"; decode.py:48 converts alignment steps to
 * frames).  [UPSTREAM] BeamRNNTInfer.align_length_sync_decoding; the evaluation order of every float
 * is documented in oracle/rnnt_alsd.c and the results are bit-identical to it.
 *
 *   beam              hypotheses kept per utterance (1..8)
 *   max_target_ratio / max_target_abs
 *                     label budget per utterance: max_target_abs when >= 0, else
 *                     (int)(max_target_ratio * enc_lens[b]); the search runs enc_lens[b] + budget steps
 *   flags             RS_ALSD_SCORE_NORM: rank finished hypotheses by score / (labels + 1)
 *                     RS_ALSD_MERGE: drop recombined duplicates from the beam (default keeps them)
 *   ids    i32[B][out_cap]  labels of the best hypothesis (no leading blank)
 *   steps  i32[B][out_cap]  alignment index i = frame + labels-before of each label
 *   n_ids  i32[B],  scores f32[B] (log-probability of the best hypothesis)
 * The workspace is separate from rs_workspace_bytes (it grows with beam and the alignment length):
 * rs_rnnt_alsd_workspace_bytes(ctx, B, beam, tp_max, max_target_ratio, max_target_abs).
 * Synchronises the stream internally.  RS_EOVERFLOW if a result has more than out_cap labels. */
enum { RS_ALSD_SCORE_NORM = 1, RS_ALSD_MERGE = 2 };
size_t rs_rnnt_alsd_workspace_bytes(const rs_ctx* ctx, int B, int beam, int tp_max, double max_target_ratio,
                                    int max_target_abs);
int rs_rnnt_alsd(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam,
                 double max_target_ratio, int max_target_abs, int flags, int out_cap, int32_t* ids,
                 int32_t* steps, int32_t* n_ids, float* scores, void* workspace, size_t workspace_bytes,
                 void* stream);

/* ---- the "default" transducer beam search ---------------------------------------------------------
 * replaces: espnet2 BeamSearchTransducer.default_beam_search + sort_nbest behind Speech2Text.__call__, which the reference
 * builds with ESPnet's defaults (beam_size 20, search_type "default", score_norm, nbest 1, lm_weight 0:
 * pkg/espnet-asr/src/transcribe.py:27-31) and calls once per window (transcribe.py:68).  Graves' search: per frame, pop the
 * best open hypothesis, keep its blank extension, open its `beam` best label extensions, until `beam` kept hypotheses beat
 * everything still open.  The evaluation order is documented in oracle/espnet_beam.c and the results are bit-identical to it.
 *
 *   beam       beam_size (>= 1; clamped to the vocabulary)
 *   flags      RS_BEAM_SCORE_NORM: the winner is the best score / len(yseq) (yseq counts the leading blank), else the best score
 *   max_pops   prediction-network evaluations allowed per frame (0 = 16 * beam).  Upstream has no bound; a trained model needs
 *              about `beam` to 2 * beam.  The workspace grows with it.
 *   ids    i32[B][out_cap]  labels of the best hypothesis (no leading blank),  n_ids i32[B],  scores f32[B] (log-probability)
 *   frames i32[B][out_cap]  the encoder frame each label was appended at, or NULL ([UPSTREAM] NeMo keeps them as
 *                           Hypothesis.timestep; ESPnet's hypotheses carry none — the espnet package times its segments by CTC)
 *   pops   i32[B]           prediction-network evaluations spent on utterance b
 * Workspace: rs_rnnt_beam_workspace_bytes(ctx, B, beam, tp_max, max_pops), separate from rs_workspace_bytes.
 * Synchronises the stream internally.  RS_EOVERFLOW if a frame needed more than max_pops pops or a result has more than out_cap
 * labels (the affected rows return n_ids = 0). */
enum { RS_BEAM_SCORE_NORM = 1 };
size_t rs_rnnt_beam_workspace_bytes(const rs_ctx* ctx, int B, int beam, int tp_max, int max_pops);
int rs_rnnt_beam(rs_ctx* ctx, const float* joint_enc, const int32_t* enc_lens, int B, int tp_max, int beam, int flags,
                 int max_pops, int out_cap, int32_t* ids, int32_t* frames, int32_t* n_ids, float* scores, int32_t* pops,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- profiling hooks for bench.py (roofline.achieved) ------------------------------------
 * When enabled, the launcher brackets every launch of the selected kernel class with HIP
 * events on the launch stream.  rs_profile_read synchronises those events and returns the
 * accumulated milliseconds, launch count and algorithmic FLOPs / bytes since the last reset. */
enum { RS_PROF_NONE = 0, RS_PROF_GEMM = 1, RS_PROF_ATTN = 2, RS_PROF_FRONTEND = 4,
       RS_PROF_DECODE = 8, RS_PROF_ELEMENTWISE = 16, RS_PROF_SUBSAMPLE = 32 };
int rs_profile_enable(rs_ctx* ctx, int class_mask);
int rs_profile_read(rs_ctx* ctx, int klass, double* ms, int64_t* launches, double* flops,
                    double* bytes);
int rs_profile_reset(rs_ctx* ctx);
/* Per-launch detail of a class since the last reset, in launch order: shapes[4 * i ..] = (M, N, K, flags) for GEMM launches
 * (zeros for other classes), flops[i], ms[i].  Up to `cap` records are copied (arrays may be NULL); *n_out = how many exist.
 * bench.py groups them into `roofline.per_shape`. */
int rs_profile_read_launches(rs_ctx* ctx, int klass, int32_t* shapes, double* flops, float* ms, int cap, int* n_out);

/* ---- single-operator entry points (parity tests call these one by one) -------------------*/

/* C[M][N] = epilogue(A[M][K] . W[N][K]^T); A, W bf16 row-major, K % 64 == 0; bf16 output needs N % 8 == 0, f32 N % 4 == 0.
 * flags: see RS_GEMM_* ; bias f32[N]; residual f32[M][ldc] (may alias out when out is f32); RS_GEMM_ROWMASK combines
 * with bf16 output only.  One kernel family serves every shape and an output row's bits do not depend on M or on the
 * tile height the launcher picks (batch invariance of the encoder). */
enum { RS_GEMM_BIAS = 1, RS_GEMM_RELU = 2, RS_GEMM_SILU = 4, RS_GEMM_RESIDUAL = 8,
       RS_GEMM_OUT_F32 = 16, RS_GEMM_ROWMASK = 32,
       /* out bf16[M][N/2] = (a + bias_a) * sigmoid(g + bias_g): columns 64j .. 64j+31 of the product are values,
        * 64j+32 .. 64j+63 their gates (weight rows interleaved in blocks of 32, see rs_set_tensor); bias only,
        * N % 64 == 0 */
       RS_GEMM_GLU = 64,
       /* icefall's SwooshL / SwooshR activations (the Zipformer family; plain bf16 or f32 output only) */
       RS_GEMM_SWOOSHL = 128, RS_GEMM_SWOOSHR = 256,
       /* exact GELU, 0.5 x (1 + erf(x / sqrt 2)) (the AV-HuBERT family; rs_gemm_f32 only) */
       RS_GEMM_GELU = 512 };
int rs_gemm_bf16(rs_ctx* ctx, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                 void* out, int ldc, int M, int N, int K, int flags, const float* bias, float alpha,
                 const float* residual, const int32_t* mask_lens, int mask_rows_per_step,
                 int mask_steps, void* stream);

/* The float32 parity mode's operators (rs_set_option "precision_f32"), one by one: the same contracts as rs_gemm_bf16 (flags
 * BIAS / RELU / SILU / RESIDUAL / ROWMASK; K % 32 == 0, N % 4 == 0), rs_relpos_attention (any head_dim <= 256) and
 * rs_glu_dwconv_silu (x f32[B*T][2*d], values | gates) on float32 tensors. */
int rs_gemm_f32(rs_ctx* ctx, const float* A, int lda, const float* W, int ldw, float* out, int ldc, int M, int N, int K,
                int flags, const float* bias, float alpha, const float* residual, const int32_t* mask_lens,
                int mask_rows_per_step, int mask_steps, void* stream);
int rs_relpos_attention_f32(rs_ctx* ctx, const float* qkv, const float* pos, const float* bias_u, const float* bias_v,
                            const int32_t* lens, int B, int T, float* ctx_out, void* stream);
int rs_glu_dwconv_silu_f32(rs_ctx* ctx, const float* x, const float* dw_w, const float* dw_b, const int32_t* lens, int B,
                           int T, int d, int k, float* out, void* stream);

/* y = LayerNorm(x) over the last dim (d); x f32[M][d]; out_bf16 and/or out_f32 may be NULL. */
int rs_layernorm(rs_ctx* ctx, const float* x, const float* gamma, const float* beta, int M, int d,
                 float eps, uint16_t* out_bf16, float* out_f32, void* stream);

/* Relative-position multi-head attention core.
 *   qkv bf16[B*T][3*d_model] (q | k | v), pos bf16[2T-1][d_model] (linear_pos of the table),
 *   bias_u/bias_v f32[n_heads][128], lens i32[B]; ctx_out bf16[B*T][d_model]. */
int rs_relpos_attention(rs_ctx* ctx, const uint16_t* qkv, const uint16_t* pos, const float* bias_u,
                        const float* bias_v, const int32_t* lens, int B, int T, uint16_t* ctx_out,
                        void* stream);

/* Conv-module middle: GLU -> zero padded frames -> depthwise k (BatchNorm folded) -> SiLU.
 *   x bf16[B*T][2*d], dw_w f32[k][d] (tap-major), dw_b f32[d]; out bf16[B*T][d]. */
int rs_glu_dwconv_silu(rs_ctx* ctx, const uint16_t* x, const float* dw_w, const float* dw_b,
                       const int32_t* lens, int B, int T, int d, int k, uint16_t* out, void* stream);
/* The same operator for the other layouts of its input: RS_GLU_HALVES = the layout above (values | gates);
 * RS_GLU_BLOCK32 = x bf16[B*T][2*d] with values / gates interleaved in blocks of 32 columns (the plain product of
 * the interleaved pw1 weight); RS_GLU_APPLIED = x bf16[B*T][d], GLU already applied by RS_GEMM_GLU. */
enum { RS_GLU_HALVES = 0, RS_GLU_BLOCK32 = 1, RS_GLU_APPLIED = 2 };
int rs_glu_dwconv_silu_layout(rs_ctx* ctx, const uint16_t* x, int layout, const float* dw_w, const float* dw_b,
                              const int32_t* lens, int B, int T, int d, int k, uint16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RS_ASR_H */

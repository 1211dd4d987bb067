"""Model dimensions of the FastConformer-RNNT path.

The reference never states them: they live in `model_config.yaml` inside the `.nemo`
checkpoint that `EncDecRNNTBPEModel.from_pretrained` downloads
(pkg/nemo-asr/src/transcribe.py:26-28).  The defaults below are the FastConformer-XL
shape that reproduces the published 619M parameters (README.rst:34-35)
and the 0.08 s frame period (pkg/nemo-asr/src/decode.py:5).  Everything [UPSTREAM] is a
field, not a constant, so a real checkpoint's YAML can override it (`from_nemo_yaml`).
"""
from dataclasses import dataclass, asdict, replace
import math


@dataclass(frozen=True)
class ModelConfig:
    # --- front-end (AudioToMelSpectrogramPreprocessor) ---
    sample_rate: int = 16000
    n_fft: int = 512
    win_length: int = 400
    hop_length: int = 160
    n_mels: int = 80
    preemph: float = 0.97
    log_guard: float = 2.0 ** -24
    norm_eps: float = 1e-5
    # --- encoder (ConformerEncoder, dw_striding subsampling) ---
    d_model: int = 1024
    n_heads: int = 8
    ff_dim: int = 4096
    n_layers: int = 24
    conv_kernel: int = 9
    sub_channels: int = 256
    sub_factor: int = 8
    xscaling: bool = True
    use_bias: bool = True   # encoder Linear / Conv1d biases ([UPSTREAM] ConformerEncoder(use_bias=...)); False -> zeros
    ln_eps: float = 1e-5
    bn_eps: float = 1e-5
    att_left: int = -1      # -1 = unlimited (full attention); SURVEY.md §8a row L5
    att_right: int = -1
    n_global: int = 0
    # --- RNN-T decoder / joint ---
    vocab_size: int = 3000  # blank id == vocab_size, logits are vocab_size + 1 wide
    pred_hidden: int = 640
    pred_layers: int = 2
    joint_hidden: int = 640
    max_symbols: int = 10
    # --- decoding strategy (checkpoint cfg `decoding.*`; [UPSTREAM] RNNTDecodingConfig / BeamRNNTInferConfig) ---
    decoding: str = "greedy_batch"       # "greedy" / "greedy_batch": batched greedy; "alsd": alignment-length synchronous beam search;
                                         # "beam": the default (Graves) beam search as ESPnet2 implements it (k_rnnt_beam.hip)
    beam_size: int = 4                   # decoding.beam.beam_size (alsd: 1..8 on the device; beam: 1..64)
    beam_max_pops: int = 0               # "beam": prediction-network evaluations allowed per frame (0 = 16 * beam_size)
    alsd_max_target_len: float = 2.0     # decoding.beam.alsd_max_target_len: float = multiple of T', int = absolute label budget
    beam_score_norm: bool = True         # decoding.beam.score_norm: rank finished hypotheses by score / len(y_sequence)
    # --- model family: "nemo" = FastConformer-RNNT (everything above as NeMo defines it); "espnet" = the ESPnet2
    #     Conformer-Transducer of reazonspeech.espnet.asr (pkg/espnet-asr/src/transcribe.py:12-32).  [UPSTREAM] ESPnet differences,
    #     each a switch of the same kernels: DefaultFrontend (periodic Hann over win_length, reflect edge padding, 1 + L // hop
    #     frames, log(max(x, 1e-10)), global mean / variance normalisation), Conv2dSubsampling x4 (two dense 3x3 stride-2 convs
    #     without padding, d_model channels), LayerNorm eps 1e-12, a LayerNorm after the last block, tanh joint without a
    #     decoder-side bias, blank = token 0 (vocab_size counts every token), one symbol per frame in greedy search, a CTC head.
    family: str = "nemo"

    # ---- derived ----
    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    @property
    def espnet(self) -> bool:
        return self.family == "espnet"

    @property
    def blank_id(self) -> int:
        return 0 if self.espnet else self.vocab_size

    @property
    def n_logits(self) -> int:
        return self.vocab_size if self.espnet else self.vocab_size + 1

    @property
    def n_sub_stages(self) -> int:
        return int(round(math.log2(self.sub_factor)))

    @property
    def sub_freq(self) -> int:
        """mel bins left after the strided convs (NeMo: 80 -> 40 -> 20 -> 10; ESPnet: 80 -> 39 -> 19)."""
        f = self.n_mels
        for _ in range(self.n_sub_stages):
            f = self.conv_out_len(f)
        return f

    def mel_frames(self, n_samples: int) -> int:
        """valid log-mel frames for `n_samples` samples: floor(L / hop)
        ([UPSTREAM] FilterbankFeatures.get_seq_len with center=True); ESPnet's Stft keeps all 1 + floor(L / hop) frames."""
        if self.espnet:
            return 1 + n_samples // self.hop_length
        return (n_samples + (self.n_fft // 2) * 2 - self.n_fft) // self.hop_length

    def stft_frames(self, n_samples: int) -> int:
        """frames torch.stft(center=True) produces: 1 + floor(L / hop)."""
        return 1 + n_samples // self.hop_length

    def conv_out_len(self, n: int) -> int:
        """output length of one subsampling conv: k=3, s=2 with p=1 (NeMo dw_striding) or p=0 (ESPnet Conv2dSubsampling)."""
        if self.espnet:
            return (n - 3) // 2 + 1 if n >= 3 else 0
        return (n + 2 - 3) // 2 + 1

    def enc_frames(self, n_mel_frames: int) -> int:
        n = n_mel_frames
        for _ in range(self.n_sub_stages):
            n = self.conv_out_len(n)
        return n

    def n_params(self) -> int:
        if self.espnet:
            return self._n_params_espnet()
        d, f, c = self.d_model, self.ff_dim, self.sub_channels
        sub = (c * 9 + c) + (self.n_sub_stages - 1) * ((c * 9 + c) + (c * c + c)) \
            + (c * self.sub_freq * d + d)
        layer = 2 * (d * f + f + f * d + d) + 4 * (d * d + d) + d * d + 2 * d \
            + (2 * d * d + 2 * d) + (d * self.conv_kernel + d) + 2 * d + (d * d + d) + 5 * 2 * d
        h = self.pred_hidden
        pred = self.n_logits * h + self.pred_layers * (4 * h * h * 2 + 8 * h)
        joint = (h * self.joint_hidden + self.joint_hidden) + (d * self.joint_hidden + self.joint_hidden) \
            + (self.joint_hidden * self.n_logits + self.n_logits)
        return sub + self.n_layers * layer + pred + joint

    def _n_params_espnet(self) -> int:
        """[UPSTREAM] ESPnet2 ESPnetASRModel with a ConformerEncoder (Conv2dSubsampling), CTC, TransducerDecoder, JointNetwork"""
        d, f, k, V, H, J = self.d_model, self.ff_dim, self.conv_kernel, self.vocab_size, self.pred_hidden, self.joint_hidden
        embed = (d * 9 + d) + (d * d * 9 + d) + (d * self.sub_freq * d + d)
        layer = 2 * (d * f + f + f * d + d) + 4 * (d * d + d) + d * d + 2 * d \
            + (2 * d * d + 2 * d) + (d * k + d) + 2 * d + (d * d + d) + 5 * 2 * d
        dec = V * H + self.pred_layers * (4 * H * H * 2 + 8 * H)
        joint = (d * J + J) + H * J + (J * V + V)
        return embed + self.n_layers * layer + 2 * d + (d * V + V) + dec + joint

    def to_dict(self):
        return asdict(self)

    def with_(self, **kw):
        return replace(self, **kw)

    @property
    def has_scores(self):
        """the decoding strategy returns a log-probability per hypothesis (the beam searches)"""
        return self.decoding in ("alsd", "beam")

    def label_cap(self, tp_max: int) -> int:
        """labels a hypothesis of a tp_max-frame utterance can hold in the output buffers"""
        if self.decoding == "alsd":       # a hypothesis has at most one label per alignment step
            budget = int(self.alsd_max_target_len * tp_max) if isinstance(self.alsd_max_target_len, float) else int(self.alsd_max_target_len)
            return max(1, tp_max + budget)
        if self.decoding == "beam":       # no per-frame limit upstream; a result longer than this is reported (RS_EOVERFLOW)
            return 2 * tp_max + 16
        return tp_max * self.max_symbols

    def validate(self):
        assert self.head_dim in (128, 64), "the attention kernel is built for head_dim 128 and 64"
        assert self.d_model % 64 == 0 and self.ff_dim % 64 == 0
        assert self.sub_channels % 64 == 0
        assert self.pred_hidden % 128 == 0 and self.joint_hidden % 128 == 0   # K slices of k_rnnt.hip
        assert self.pred_hidden == self.joint_hidden or True
        assert self.family in ("nemo", "espnet"), f"model family {self.family!r}"
        if self.espnet:
            assert self.sub_factor == 4 and self.sub_channels == self.d_model, "ESPnet Conv2dSubsampling: x4, d_model channels"
            assert self.decoding in ("greedy", "greedy_batch", "beam"), "the ESPnet path decodes greedily or with the default beam search"
        assert self.conv_kernel % 2 == 1 and self.conv_kernel <= 31
        assert self.n_fft == 512 and self.win_length <= 512
        assert self.decoding in ("greedy", "greedy_batch", "alsd", "beam"), f"decoding strategy {self.decoding!r}"
        assert 1 <= self.beam_size <= (64 if self.decoding == "beam" else 8), "beam_size 1..8 (alsd) / 1..64 (beam)"
        assert self.beam_max_pops == 0 or self.beam_max_pops >= self.beam_size
        assert self.alsd_max_target_len >= 0
        return self


# the configuration the metric is quoted on (BASELINE.json)
FASTCONFORMER_619M = ModelConfig()

# a shape small enough for CPU oracle runs and committed golden fixtures
TINY = ModelConfig(d_model=256, n_heads=2, ff_dim=512, n_layers=2, sub_channels=64,
                   vocab_size=63, pred_hidden=128, joint_hidden=128)


# the 619M geometry (d = 1024, 8 heads, C = 256, FFN 4096, V + 1 = 3001, 640-wide LSTM / joint) with two
# layers: every kernel instantiation and tile path of the benchmark configuration at a size the CPU oracle
# and the HF golden generator finish in seconds (tests/golden/parakeet_wide.npz)
WIDE2 = ModelConfig(n_layers=2)

# reazonspeech.espnet.asr: "Conformer-Transducer ... 120M parameters" (README.rst:37-40).  The architecture itself lives in
# the config.yaml of an unreachable Hugging Face repository (pkg/espnet-asr/src/transcribe.py:27-31): [UPSTREAM] this is the
# ESPnet2 conformer recipe shape (512 / 8 heads / 2048 / kernel 31, DefaultFrontend 512 / 128) with the depth and
# vocabulary that give 120M parameters; every number is a field a real config overrides.
ESPNET_CONFORMER_120M = ModelConfig(
    family="espnet", win_length=512, hop_length=128, preemph=0.0, log_guard=1e-10, norm_eps=1e-20,
    d_model=512, n_heads=8, ff_dim=2048, n_layers=17, conv_kernel=31, sub_channels=512, sub_factor=4, xscaling=True,
    ln_eps=1e-12, vocab_size=2600, pred_hidden=512, pred_layers=1, joint_hidden=640, max_symbols=1)

# its toy shape for CPU oracle runs
ESPNET_TINY = ModelConfig(
    family="espnet", win_length=512, hop_length=128, preemph=0.0, log_guard=1e-10, norm_eps=1e-20,
    d_model=256, n_heads=4, ff_dim=512, n_layers=2, conv_kernel=15, sub_channels=256, sub_factor=4, xscaling=True,
    ln_eps=1e-12, vocab_size=96, pred_hidden=128, pred_layers=1, joint_hidden=128, max_symbols=1)


class UnsupportedCheckpoint(ValueError):
    """the checkpoint asks for an architecture variant the HIP kernels do not implement"""


def resolve_interpolations(cfg: dict) -> dict:
    """Resolve OmegaConf-style `${a.b.c}` references (NeMo configs use them freely:
    `feat_in: ${model.preprocessor.features}`, `pred_hidden: ${model.model_defaults.pred_hidden}`) against the
    document itself, with or without the leading `model.`; `???` (OmegaConf's "missing") becomes None."""
    import re
    pat = re.compile(r"^\$\{([^}]+)\}$")

    def lookup(path):
        for keys in (path.split("."), path.split(".")[1:] if path.startswith("model.") else None):
            if keys is None:
                continue
            node = cfg
            try:
                for k in keys:
                    node = node[int(k)] if isinstance(node, list) else node[k]
                return node
            except (KeyError, IndexError, TypeError, ValueError):
                continue
        raise UnsupportedCheckpoint(f"model_config.yaml: cannot resolve ${{{path}}}")

    def walk(node, depth=0):
        if depth > 32:
            raise UnsupportedCheckpoint("model_config.yaml: interpolation cycle")
        if isinstance(node, dict):
            return {k: walk(v, depth) for k, v in node.items()}
        if isinstance(node, list):
            return [walk(v, depth) for v in node]
        if isinstance(node, str):
            if node == "???":
                return None
            m = pat.match(node.strip())
            if m:
                return walk(lookup(m.group(1)), depth + 1)
        return node

    return walk(cfg)


# Every key of the four model sections that this loader either interprets ("used") or knows to have no effect on
# inference ("inert": training / logging / memory knobs).  [UPSTREAM] NeMo >= 2.6:
# examples/asr/conf/fastconformer/fast-conformer_transducer_bpe.yaml, ConformerEncoder / RNNTDecoder / RNNTJoint /
# AudioToMelSpectrogramPreprocessor constructor arguments.  In strict mode a key outside these tables raises: a setting
# nobody mapped is a setting that may change what the checkpoint computes.
KNOWN_KEYS = {
    "preprocessor": {
        "used": {"sample_rate", "n_fft", "window_size", "window_stride", "features", "preemph", "log_zero_guard_value",
                 "normalize", "window", "frame_splicing", "log", "log_zero_guard_type", "mag_power", "n_window_size",
                 "n_window_stride", "exact_pad", "highfreq", "lowfreq", "mel_norm"},
        "inert": {"_target_", "dither", "pad_to", "nb_augmentation_prob", "pad_value", "use_torchaudio", "rng", "nb_max_freq", "stft_exact_pad",
                  "stft_conv", "use_grads"},
    },
    "encoder": {
        "used": {"feat_in", "n_layers", "d_model", "use_bias", "subsampling", "subsampling_factor", "subsampling_conv_channels",
                 "causal_downsampling", "reduction", "reduction_position", "reduction_factor", "ff_expansion_factor",
                 "self_attention_model", "n_heads", "att_context_size", "att_context_style", "att_context_probs", "xscaling",
                 "untie_biases", "conv_kernel_size", "conv_norm_type", "conv_context_size", "global_tokens",
                 "global_tokens_spacing", "global_attn_separate", "feat_out", "subsampling_conv_chunking_factor",
                 "use_pytorch_sdpa", "use_pytorch_sdpa_backends", "sync_max_audio_length"},
        "inert": {"_target_", "pos_emb_max_len", "dropout", "dropout_pre_encoder", "dropout_emb", "dropout_att",
                  "stochastic_depth_drop_prob", "stochastic_depth_mode", "stochastic_depth_start_layer"},
    },
    "decoder": {
        "used": {"vocab_size", "prednet", "blank_as_pad", "normalization_mode"},
        "inert": {"_target_", "random_state_sampling"},
    },
    "decoder.prednet": {
        "used": {"pred_hidden", "pred_rnn_layers", "rnn_type", "rnn_hidden_size"},
        "inert": {"t_max", "dropout", "forget_gate_bias", "weights_init_scale", "hidden_hidden_bias_scale"},
    },
    "joint": {
        "used": {"num_classes", "vocabulary", "jointnet", "num_extra_outputs"},
        "inert": {"_target_", "log_softmax", "preserve_memory", "fuse_loss_wer", "fused_batch_size", "masking_prob"},
    },
    "joint.jointnet": {
        "used": {"joint_hidden", "activation", "encoder_hidden", "pred_hidden"},
        "inert": {"dropout"},
    },
}


def _check_known(section: str, node: dict):
    known = KNOWN_KEYS[section]["used"] | KNOWN_KEYS[section]["inert"]
    unknown = sorted(k for k in node if k not in known)
    if unknown:
        raise UnsupportedCheckpoint(f"model_config.yaml: unknown setting(s) {section}.{{{', '.join(unknown)}}} — not mapped by "
                                    f"this loader, so their effect on the computation is unknown (strict mode; "
                                    f"from_nemo_yaml(..., strict=False) ignores them)")


def from_nemo_yaml(cfg: dict, strict: bool = True) -> ModelConfig:
    """Map a NeMo `model_config.yaml` (already parsed to a dict) onto ModelConfig.
    [UPSTREAM] key names follow NeMo >= 2.6 `EncDecRNNTBPEModel` configs
    (examples/asr/conf/fastconformer/fast-conformer_transducer_bpe.yaml).

    With `strict` (the default) every setting that would make the kernels compute something else than the
    checkpoint's architecture raises `UnsupportedCheckpoint` instead of loading silently: other
    subsampling types, LayerNorm in the conv module, Longformer global attention with separate projections,
    a non-per-feature front-end normalisation, frame splicing, a non-ReLU joint, and so on.  `decoding.strategy`
    greedy / greedy_batch / alsd select the device search; anything else warns and decodes greedily."""
    import warnings
    if "model" in cfg and isinstance(cfg["model"], dict) and "encoder" in cfg["model"]:
        cfg = {**cfg["model"], "model": cfg["model"]}        # training-style file: the model subtree is the config
    cfg = resolve_interpolations(cfg)
    pre = cfg.get("preprocessor", {}) or {}
    enc = cfg.get("encoder", {}) or {}
    dec = cfg.get("decoder", {}) or {}
    joint = cfg.get("joint", {}) or {}
    sr = int(pre.get("sample_rate", cfg.get("sample_rate", 16000)) or 16000)
    prednet = dec.get("prednet", {}) or {}
    jn = joint.get("jointnet", {}) or {}
    ctx = enc.get("att_context_size", [-1, -1]) or [-1, -1]
    if ctx and isinstance(ctx[0], (list, tuple)):
        ctx = ctx[0]
    att_model = str(enc.get("self_attention_model", "rel_pos"))
    local = att_model == "rel_pos_local_attn"
    decoding = cfg.get("decoding", {}) or {}
    greedy = decoding.get("greedy", {}) or {}
    vocab = dec.get("vocab_size", None)
    if vocab is None:
        vocab = joint.get("num_classes", None)
    if vocab is None and isinstance(joint.get("vocabulary"), (list, tuple)):
        vocab = len(joint["vocabulary"])
    if vocab is None:
        vocab = 3000
    if strict:
        def need(cond, what):
            if not cond:
                raise UnsupportedCheckpoint(f"model_config.yaml: {what} is not implemented by the gfx950 kernels")
        for section, node in (("preprocessor", pre), ("encoder", enc), ("decoder", dec), ("decoder.prednet", prednet),
                              ("joint", joint), ("joint.jointnet", jn)):
            _check_known(section, node)
        need(str(enc.get("subsampling", "dw_striding")) == "dw_striding", f"encoder.subsampling={enc.get('subsampling')!r}")
        need(bool(enc.get("untie_biases", True)), "encoder.untie_biases=false (position biases shared by all layers)")
        need(int(enc.get("feat_out", -1) or -1) in (-1, int(enc.get("d_model", 1024))), "encoder.feat_out (an output projection)")
        need(not enc.get("global_tokens_spacing") or int(enc.get("global_tokens_spacing")) == 1, "encoder.global_tokens_spacing != 1")
        need(dec.get("normalization_mode") in (None, "null"), f"decoder.normalization_mode={dec.get('normalization_mode')!r}")
        need(int(joint.get("num_extra_outputs", 0) or 0) == 0, "joint.num_extra_outputs (TDT durations)")
        need(pre.get("highfreq") in (None, "null") or float(pre.get("highfreq")) == sr / 2.0, f"preprocessor.highfreq={pre.get('highfreq')!r}")
        need(float(pre.get("lowfreq", 0) or 0) == 0.0, f"preprocessor.lowfreq={pre.get('lowfreq')!r}")
        need(str(pre.get("mel_norm", "slaney")) == "slaney", f"preprocessor.mel_norm={pre.get('mel_norm')!r}")
        need(not pre.get("exact_pad", False), "preprocessor.exact_pad")
        need(prednet.get("rnn_hidden_size") in (None, "null") or int(prednet.get("rnn_hidden_size")) == int(prednet.get("pred_hidden", 640)),
             "decoder.prednet.rnn_hidden_size != pred_hidden (projected LSTM)")
        need(jn.get("encoder_hidden") in (None, "null") or int(jn.get("encoder_hidden")) == int(enc.get("d_model", 1024)),
             "joint.jointnet.encoder_hidden != encoder.d_model")
        need(jn.get("pred_hidden") in (None, "null") or int(jn.get("pred_hidden")) == int(prednet.get("pred_hidden", 640)),
             "joint.jointnet.pred_hidden != decoder.prednet.pred_hidden")
        need(not enc.get("causal_downsampling", False), "encoder.causal_downsampling")
        need(att_model in ("rel_pos", "rel_pos_local_attn"), f"encoder.self_attention_model={att_model!r}")
        need(str(enc.get("conv_norm_type", "batch_norm")) == "batch_norm", f"encoder.conv_norm_type={enc.get('conv_norm_type')!r}")
        need(enc.get("conv_context_size") in (None, "null") or list(enc.get("conv_context_size")) == [
            (int(enc.get("conv_kernel_size", 9)) - 1) // 2] * 2, "a causal / asymmetric encoder.conv_context_size")
        need(str(enc.get("att_context_style", "regular")) == "regular", f"encoder.att_context_style={enc.get('att_context_style')!r}")
        need(int(enc.get("reduction_factor", 1) or 1) == 1 and enc.get("reduction") in (None, "null"), "encoder.reduction")
        need(not (local and enc.get("global_attn_separate", False)), "encoder.global_attn_separate (separate global q/k/v)")
        need(int(enc.get("feat_in", pre.get("features", 80)) or 80) == int(pre.get("features", 80)), "encoder.feat_in != preprocessor.features")
        need(str(pre.get("normalize", "per_feature")) == "per_feature", f"preprocessor.normalize={pre.get('normalize')!r}")
        need(str(pre.get("window", "hann")) == "hann", f"preprocessor.window={pre.get('window')!r}")
        need(int(pre.get("frame_splicing", 1) or 1) == 1, "preprocessor.frame_splicing")
        need(bool(pre.get("log", True)), "preprocessor.log=false")
        need(str(pre.get("log_zero_guard_type", "add")) == "add", "preprocessor.log_zero_guard_type")
        need(float(pre.get("mag_power", 2.0) or 2.0) == 2.0, "preprocessor.mag_power != 2")
        need(str(jn.get("activation", "relu")).lower() == "relu", f"joint.jointnet.activation={jn.get('activation')!r}")
        need(dec.get("blank_as_pad", True) in (True, None), "decoder.blank_as_pad=false")
        need(str(prednet.get("rnn_type", "lstm") or "lstm").lower() == "lstm" if "rnn_type" in prednet else True, "a non-LSTM prediction network")
    strategy = str(decoding.get("strategy", "greedy_batch"))
    beam = decoding.get("beam", {}) or {}
    if strategy not in ("greedy", "greedy_batch", "alsd", "beam"):
        warnings.warn(f"checkpoint decoding.strategy={strategy!r}: only greedy, alsd and beam (the default beam search) are implemented; decoding greedily "
                      "(batched greedy RNN-T), transcripts can differ from the reference's beam search", RuntimeWarning,
                      stacklevel=2)
        strategy = "greedy_batch"
    beam_size = int(beam.get("beam_size", 4) or 4)
    if strategy == "alsd" and not 1 <= beam_size <= 8:
        raise UnsupportedCheckpoint(f"decoding.beam.beam_size={beam_size}: the device beam search keeps 1..8 hypotheses")
    if strategy == "beam" and not 1 <= beam_size <= 64:
        # [UPSTREAM] BeamRNNTInfer.default_beam_search is the same Graves search ESPnet implements (k_rnnt_beam.hip)
        raise UnsupportedCheckpoint(f"decoding.beam.beam_size={beam_size}: the device default beam search keeps 1..64 hypotheses")
    max_target = beam.get("alsd_max_target_len", 2.0)
    max_target = 2.0 if max_target is None else (int(max_target) if isinstance(max_target, int) else float(max_target))
    guard = pre.get("log_zero_guard_value", 2.0 ** -24)
    if isinstance(guard, str):            # NeMo also accepts the names of torch.finfo fields
        import torch
        guard = {"tiny": torch.finfo(torch.float32).tiny, "eps": torch.finfo(torch.float32).eps}.get(guard, 2.0 ** -24)
    return ModelConfig(
        sample_rate=sr,
        n_fft=int(pre.get("n_fft", 512) or 512),
        # (n_window_size / n_window_stride give the same two numbers in samples)
        win_length=int(pre["n_window_size"]) if pre.get("n_window_size") else int(round(float(pre.get("window_size") or 0.025) * sr)),
        hop_length=int(pre["n_window_stride"]) if pre.get("n_window_stride") else int(round(float(pre.get("window_stride") or 0.01) * sr)),
        n_mels=int(pre.get("features", 80)),
        preemph=float(pre.get("preemph", 0.97) or 0.0),
        log_guard=float(guard),
        d_model=int(enc.get("d_model", 1024)),
        n_heads=int(enc.get("n_heads", 8)),
        ff_dim=int(enc.get("d_model", 1024)) * int(enc.get("ff_expansion_factor", 4)),
        n_layers=int(enc.get("n_layers", 24)),
        conv_kernel=int(enc.get("conv_kernel_size", 9)),
        sub_channels=int(enc.get("subsampling_conv_channels", 256)),
        sub_factor=int(enc.get("subsampling_factor", 8)),
        xscaling=bool(enc.get("xscaling", True)),
        use_bias=bool(enc.get("use_bias", True)),
        att_left=int(ctx[0]) if local else -1,
        att_right=int(ctx[1]) if local else -1,
        n_global=int(enc.get("global_tokens", 0) or 0) if local else 0,
        vocab_size=int(vocab),
        pred_hidden=int(prednet.get("pred_hidden", 640)),
        pred_layers=int(prednet.get("pred_rnn_layers", 2)),
        joint_hidden=int(jn.get("joint_hidden", 640)),
        max_symbols=int(greedy.get("max_symbols", greedy.get("max_symbols_per_step", 10)) or 10),
        decoding=strategy,
        beam_size=min(max(beam_size, 1), 64 if strategy == "beam" else 8),
        alsd_max_target_len=max_target,
        beam_score_norm=bool(beam.get("score_norm", True)),
    ).validate()

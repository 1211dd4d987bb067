"""`AVHubertModel` / `AVHubertForConditionalGeneration` of `reazonspeech.avsr` (pkg/avsr/src/avhubert/modeling_avhubert.py:119-213,
:216-391) on one MI355X.  Same call forms as the reference's classes for inference:

    model = AVHubertForConditionalGeneration.from_pretrained(path)            # or (config, state_dict) directly
    out = model.generate(**inputs, num_beams=5, max_new_tokens=256)           # README.rst
    enc = model.avhubert(input_values=..., pixel_values=..., padding_mask=...).last_hidden_state

Everything between the input tensors and the logits runs in librs_asr.so (csrc/k_avsr.hip, float32 like the reference); the search
over the logits is host logic (generation.py).  Training-time arguments (labels, dropout, layerdrop) have no counterpart."""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from ..runtime.avsr_config import AvsrConfig, AVSR_BASE
from ..runtime.avsr_model import AvsrDevice
from . import generation

AVHubertConfig = AvsrConfig


@dataclass
class AVHubertOutput:
    """modeling_avhubert.py:33-37"""
    last_hidden_state: Optional[torch.Tensor] = None
    hidden_states: Optional[torch.Tensor] = None
    attentions: Optional[torch.Tensor] = None


@dataclass
class Seq2SeqLMOutput:
    logits: Optional[torch.Tensor] = None
    encoder_last_hidden_state: Optional[torch.Tensor] = None


@dataclass
class BeamOutput:
    sequences: torch.Tensor
    sequences_scores: Optional[torch.Tensor] = None


class AVHubertModel:
    """the encoder (modeling_avhubert.py:119-213)"""

    def __init__(self, config: AvsrConfig, state_dict=None, device="cuda", _dev=None, products=None):
        self.config = config
        self.dev = _dev if _dev is not None else AvsrDevice(config, state_dict, device, products=products)
        self.device = self.dev.device

    def forward(self, input_values=None, pixel_values=None, padding_mask=None, **kwargs):
        if input_values is None and pixel_values is None:
            raise ValueError("Either `input_values` or `pixel_values` must be passed")            # modeling_avhubert.py:181
        # :172-177 a missing modality contributes zero FEATURES in its half of the fused vector (rs_avsr_encoder_forward takes NULL for it)
        B, T = (input_values if input_values is not None else pixel_values).shape[:2]
        if padding_mask is None:
            padding_mask = np.zeros((B, T), np.float32)
        return AVHubertOutput(last_hidden_state=self.dev.encode(input_values, pixel_values, padding_mask))

    __call__ = forward


class AVHubertForConditionalGeneration:
    def __init__(self, config: AvsrConfig, state_dict, device="cuda", products=None):
        """products: None ($REAZONSPEECH_AVSR_PRODUCTS, default "exact") | "exact" | "x3" — runtime/avsr_model.py set_products"""
        if config.vocab_size is None:
            raise ValueError("the configuration does not define `vocab_size`")                    # modeling_avhubert.py:232-238
        self.config = config
        self.dev = AvsrDevice(config, state_dict, device, products=products)
        self.device = self.dev.device
        self.avhubert = AVHubertModel(config, _dev=self.dev)

    @classmethod
    def from_pretrained(cls, path, device="cuda", products=None):
        """a directory with config.json + model.safetensors / pytorch_model.bin under the reference's parameter names"""
        from ..runtime.avsr_weights import read_avsr
        cfg, sd = read_avsr(path)
        return cls(cfg, sd, device=device, products=products)

    def get_encoder(self):
        return self.avhubert

    def eval(self):
        return self

    def forward(self, input_values=None, pixel_values=None, padding_mask=None, decoder_input_ids=None, **kwargs):
        """teacher-forced logits (modeling_avhubert.py:256-314): decoder_input_ids int [B][L] -> logits float32 [B][L][V]"""
        enc = self.avhubert(input_values=input_values, pixel_values=pixel_values, padding_mask=padding_mask).last_hidden_state
        ids = np.asarray(decoder_input_ids.cpu() if torch.is_tensor(decoder_input_ids) else decoder_input_ids)
        B, L = ids.shape
        mask = padding_mask if padding_mask is not None else np.zeros(enc.shape[:2], np.float32)
        dec = self.dev.decoding(enc, mask, 1, L)
        logits = torch.empty((B, L, self.config.vocab_size), dtype=torch.float32, device=self.device)
        for t in range(L):
            logits[:, t] = dec.step(ids[:, t], t)
        return Seq2SeqLMOutput(logits=logits, encoder_last_hidden_state=enc)

    __call__ = forward

    def generate(self, input_values=None, pixel_values=None, padding_mask=None, num_beams=1, max_new_tokens=20, do_sample=False,
                 length_penalty=1.0, return_dict_in_generate=False, **kwargs):
        """transformers' generate() for the two modes the reference's README uses: greedy (num_beams 1) and beam search.
        -> LongTensor [B][<= 1 + max_new_tokens] starting with bos (CPU), or BeamOutput with `sequences_scores`"""
        if do_sample:
            raise NotImplementedError("sampling is not built (the reference's documented call is deterministic beam search)")
        # transformers' generate() takes dozens of options; the ones that change the search and are not restated here must not be
        # dropped silently
        if "max_length" in kwargs and kwargs["max_length"] is not None:
            max_new_tokens = int(kwargs.pop("max_length")) - 1              # the prompt is the one bos token
        neutral = {"use_cache": None, "output_scores": None, "early_stopping": False, "num_return_sequences": 1, "num_beam_groups": 1,
                   "repetition_penalty": 1.0, "no_repeat_ngram_size": 0, "temperature": 1.0, "top_k": None, "top_p": None,
                   "attention_mask": None, "max_length": None}
        for k, v in kwargs.items():
            if k not in neutral:
                raise TypeError(f"generate(): option `{k}` is not built (greedy and beam search with num_beams, max_new_tokens / max_length, length_penalty are)")
            if neutral[k] is not None and v is not None and v != neutral[k]:
                raise NotImplementedError(f"generate(): `{k}={v!r}` changes the search and is not built (only {neutral[k]!r})")
        enc = self.avhubert(input_values=input_values, pixel_values=pixel_values, padding_mask=padding_mask).last_hidden_state
        mask = padding_mask if padding_mask is not None else np.zeros(enc.shape[:2], np.float32)
        if num_beams <= 1:
            seq, scores = generation.greedy_search(self.dev, enc, mask, int(max_new_tokens)), None
        else:
            seq, scores = generation.beam_search(self.dev, enc, mask, int(num_beams), int(max_new_tokens), float(length_penalty))
        seq = torch.from_numpy(np.ascontiguousarray(seq))
        if return_dict_in_generate:
            return BeamOutput(seq, None if scores is None else torch.from_numpy(np.ascontiguousarray(scores)))
        return seq


def synthetic_model(config: AvsrConfig = AVSR_BASE, seed: int = 0, device="cuda", products=None):
    """seeded synthetic weights under the reference's parameter names (benchmarks / tests: no checkpoint is reachable offline)"""
    from ..runtime.avsr_weights import synthetic_state_dict_avsr
    return AVHubertForConditionalGeneration(config, synthetic_state_dict_avsr(config, seed), device=device, products=products)

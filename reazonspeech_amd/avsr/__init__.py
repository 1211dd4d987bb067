"""Drop-in surface of `reazonspeech.avsr` (pkg/avsr/src/__init__.py:9-22): AVHubertConfig, AVHubertModel,
AVHubertForConditionalGeneration, AVHubertFeatureExtractor, AVHubertProcessor — the model classes run on MI355X through
librs_asr.so (csrc/k_avsr.hip); the transformers Auto* registrations of the reference have no counterpart (this package does not
subclass PreTrainedModel)."""
from .modeling import AVHubertConfig, AVHubertModel, AVHubertForConditionalGeneration, synthetic_model
from .feature_extraction import AVHubertFeatureExtractor, AVHubertProcessor

__all__ = ["AVHubertConfig", "AVHubertModel", "AVHubertForConditionalGeneration", "AVHubertFeatureExtractor", "AVHubertProcessor", "synthetic_model"]

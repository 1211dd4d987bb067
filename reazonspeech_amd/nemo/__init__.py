"""Namespace mirror of `reazonspeech.nemo` (reference: pkg/nemo-asr/pyproject.toml:16-17)."""

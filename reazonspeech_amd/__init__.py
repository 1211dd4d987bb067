"""reazonspeech_amd — MI355X-native (gfx950) drop-in for the `reazonspeech.nemo.asr`
FastConformer-RNNT inference path of reazon-research/ReazonSpeech.

Only the hot path named by BASELINE.json:north_star lives here:
the Python boundary (`reazonspeech_amd.nemo.asr`), the host runtime that owns
device buffers through PyTorch-ROCm (`reazonspeech_amd.runtime`) and the
hand-written HIP kernels behind the C-ABI of `include/rs_asr.h`
(`reazonspeech_amd/csrc`, built into `reazonspeech_amd/lib/librs_asr.so`).
"""

__version__ = "0.1.0"

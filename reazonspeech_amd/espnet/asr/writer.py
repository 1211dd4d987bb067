"""Subtitle / transcript writers of `reazonspeech.espnet.asr` (pkg/espnet-asr/src/writer.py — byte-identical to the nemo
package's writer.py in the reference: `diff pkg/espnet-asr/src/writer.py pkg/nemo-asr/src/writer.py` is empty)."""
from ...nemo.asr.writer import *           # noqa: F401,F403
from ...nemo.asr.writer import get_writer  # noqa: F401

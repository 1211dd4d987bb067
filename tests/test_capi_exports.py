"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/rs_asr.h declares (no compute: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from reazonspeech_amd import build as rs_build
from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import TINY, FASTCONFORMER_619M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rs_asr.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    lib_path = rs_build.build()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in rs_asr.h but not exported"
    assert sorted(capi.EXPORTS) == syms, "capi.EXPORTS must list exactly the header's functions"


def test_header_cites_reference_and_is_plain_c():
    src = open(HEADER).read()
    assert "pkg/nemo-asr/src/transcribe.py" in src
    assert "torch" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S)      # no torch types in signatures
    assert 'extern "C"' in src


def test_dims_struct_matches_header_field_order():
    src = open(HEADER).read()
    body = src[src.index("typedef struct rs_dims {"):src.index("} rs_dims;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int32_t|float)\s+([a-z_0-9]+)\s*;", body)
    assert fields == [f[0] for f in capi.RsDims._fields_]
    d = capi.RsDims.from_config(FASTCONFORMER_619M)
    assert (d.d_model, d.n_layers, d.n_logits, d.blank_id, d.sub_stages) == (1024, 24, 3001, 3000, 3)


def test_context_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.RsError):
        capi.Context(TINY, 0)
    from reazonspeech_amd.nemo.asr import load_model
    with pytest.raises(RuntimeError):
        load_model()            # no silent CPU fallback


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "reazonspeech_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "librs_oracle" not in text, f


def test_host_stage_rows_gathers_a_ragged_batch():
    """rs_host_stage_rows (pure host code of the C ABI): rows copied, tails zeroed up to the batch width only, lengths
    written, rows of a short last batch get length 0, bad arguments are rejected"""
    import numpy as np
    import torch
    dst = torch.full((5, 100), 7.0)
    lens = torch.full((5,), -1, dtype=torch.int32)
    waves = [np.arange(10, dtype=np.float32), np.zeros(0, np.float32), np.arange(64, dtype=np.float64)[::2]]
    capi.host_stage_rows(dst, 64, waves, lens)
    assert lens.tolist() == [10, 0, 32, 0, 0]
    assert dst[0, :10].tolist() == list(range(10)) and (dst[0, 10:64] == 0).all() and (dst[0, 64:] == 7).all()
    assert (dst[1, :64] == 0).all() and dst[2, :32].tolist() == list(range(0, 64, 2)) and (dst[2, 32:64] == 0).all()
    assert (dst[3:] == 7).all()                       # rows past the batch keep their bytes; only their length is reset
    with pytest.raises(capi.RsError):
        capi.host_stage_rows(dst, 8, waves, lens)     # an utterance longer than the width


def test_host_stage_rows_into_a_narrowed_contiguous_view():
    """What the host pipeline does for a batch shorter than its buffer set (runtime/model.py: _BufView): the pinned
    staging matrix is re-viewed as a contiguous [rows][l_max] block of the same memory and the batch is gathered with THAT
    pitch, so the host-to-device copy of the batch is one contiguous range (a column slice of the full-pitch matrix was
    copied through a synchronous temporary)."""
    import numpy as np
    import torch
    base = torch.full((4, 256), 9.0)
    view = base.view(-1)[:4 * 64].view(4, 64)              # what _BufView's `cut` builds
    assert view.is_contiguous() and view.data_ptr() == base.data_ptr()
    lens = torch.zeros((4,), dtype=torch.int32)
    waves = [np.full(40, 1.0, np.float32), np.full(64, 2.0, np.float32), np.full(3, 3.0, np.float32)]
    capi.host_stage_rows(view, 64, waves, lens)
    flat = base.view(-1)
    assert lens.tolist() == [40, 64, 3, 0]
    assert (flat[0:40] == 1).all() and (flat[40:64] == 0).all()
    assert (flat[64:128] == 2).all()
    assert (flat[128:131] == 3).all() and (flat[131:192] == 0).all()
    assert (flat[256:] == 9).all()                          # nothing beyond the view's rows was touched

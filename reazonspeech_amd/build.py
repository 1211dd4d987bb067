"""Build librs_asr.so (gfx950) from reazonspeech_amd/csrc with hipcc.

    python -m reazonspeech_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  Objects are cached under reazonspeech_amd/csrc/.obj and
rebuilt when a source or header is newer.  The decode kernels (k_rnnt.hip) are compiled with
-ffp-contract=off: their float32 results are compared bit for bit with the C oracle.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librs_asr.so")
ARCH = "gfx950"

SOURCES = ["rs_api.hip", "k_gemm_bf16.hip", "k_layernorm.hip", "k_attention.hip", "k_frontend.hip",
           "k_subsample.hip", "k_rnnt.hip", "k_rnnt_alsd.hip", "k_rnnt_beam.hip", "k_f32.hip", "k_espnet.hip", "k_zipformer.hip", "k_avsr.hip"]
EXTRA = {"k_rnnt.hip": ["-ffp-contract=off"],
         "k_rnnt_alsd.hip": ["-ffp-contract=off"],
         "k_rnnt_beam.hip": ["-ffp-contract=off"]}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "rs_asr.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    cmd = [_hipcc()] + COMMON + EXTRA.get(src, []) + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = _headers()
    todo = []
    for src in SOURCES:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(obj, [os.path.join(CSRC, src)] + hdrs):
            todo.append(src)
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda s: _compile(s, verbose), todo))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if todo or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))

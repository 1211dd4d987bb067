"""Build the C part of the oracle (oracle/_ref is not applicable: the reference's arithmetic
lives in NeMo, which is neither vendored nor buildable here — see oracle/__init__.py).

    python oracle/build.py        -> oracle/librs_oracle.so
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(HERE, n) for n in ("rnnt_greedy.c", "rnnt_alsd.c", "espnet_beam.c", "k2_greedy.c")]
DEPENDS = SOURCES + [os.path.join(HERE, "rnnt_math.h")]
OUT = os.path.join(HERE, "librs_oracle.so")


def stale():
    return not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(p) for p in DEPENDS)


def build(force=False):
    if not force and not stale():
        return OUT
    cmd = ["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
           "-o", OUT] + SOURCES + ["-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

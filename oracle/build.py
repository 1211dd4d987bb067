"""Build the C part of the oracle (oracle/_ref is not applicable: the reference's arithmetic
lives in NeMo, which is neither vendored nor buildable here — see oracle/__init__.py).

    python oracle/build.py        -> oracle/librs_oracle.so
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rnnt_greedy.c")
OUT = os.path.join(HERE, "librs_oracle.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
           "-o", OUT, SRC, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

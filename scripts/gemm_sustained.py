"""Sustained (seconds-long) GEMM throughput per kernel variant, with power / clock samples from rocm-smi.

    python scripts/gemm_sustained.py [variants...]
Short microbenchmarks (scripts/gemm_bench.py: ~10 ms bursts) and the real pipeline disagree on how much a better
main loop is worth; this runs one shape back to back for ~2 s per variant to see whether the chip is power/clock
limited in the sustained regime.
"""
import ctypes, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M

variants = [int(v) for v in sys.argv[1:]] or [2, 20]       # variant -1 = torch.matmul (hipBLASLt), the library yardstick
M, N, K = 35328, 4096, 1024
dev = torch.device("cuda", 0)
ctx = capi.Context(FASTCONFORMER_619M, 0)
ctx.lib.rs_debug_set_gemm_variant.argtypes = [ctypes.c_int]
g = torch.Generator().manual_seed(0)
# several independent operand sets so that consecutive launches do not reuse hot cache lines
sets = []
for i in range(4):
    A = torch.randn((M, K), generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn((N, K), generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    sets.append((A, W, out))
bias = torch.randn((N,), generator=g).to(dev)
samples = []
stop = False


def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "-P", "-c", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), o.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.time(), f"rocm-smi failed: {e}"))
        time.sleep(0.25)


th = threading.Thread(target=poll, daemon=True)
th.start()
secs = float(os.environ.get('SECS', '2.0'))
def launch(v, A, W, out):
    if v < 0:
        torch.matmul(A, W.t(), out=out)
    else:
        ctx.gemm(A, W, out, flags=capi.GEMM_BIAS | capi.GEMM_SILU, bias=bias)


for v in variants * int(os.environ.get('REPS', '2')):
    if v >= 0:
        ctx.lib.rs_debug_set_gemm_variant(v)
    for A, W, out in sets:
        launch(v, A, W, out)
    torch.cuda.synchronize()
    n, t0 = 0, time.time()
    marks = []
    while time.time() - t0 < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            A, W, out = sets[i % 4]
            launch(v, A, W, out)
        e1.record()
        torch.cuda.synchronize()
        marks.append(e0.elapsed_time(e1) * 10.0)      # us per launch
        n += 100
    tf = [2.0 * M * N * K / us / 1e6 for us in marks]
    tail = tf[len(tf) // 2:]
    print(f"v{v}: {n} launches; TF/s first window {tf[0]:.0f}, second-half mean {sum(tail) / len(tail):.0f} (min {min(tail):.0f}, max {max(tail):.0f})", flush=True)
    mine = [s for (t, s) in samples if t >= t0]
    if mine:
        print("   rocm-smi:", mine[len(mine) // 2].split("|")[-1].strip(), "(fclk, lvl, mclk, lvl, sclk, lvl, socclk, lvl, package W)", flush=True)
stop = True

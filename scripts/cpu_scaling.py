#!/usr/bin/env python3
"""Why does the CPU port's all-thread DECODE scale so much worse than its encode (bench.py cpu_baseline: efficiency 0.05 against
0.20)?  Encode and decode rates of the C2 batch on 1 ... all logical CPUs (pinned, buffers first touched by their threads), the
decoder with its output elided (CST_ORACLE_DECODE_SINK=1: every thread writes its streams into one 16-KiB buffer), and the box's
store bandwidth with the same threads.  Host only: no GPU needed.  usage: cpu_scaling.py [n_streams]"""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import oracle as O

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N, P, LO = 4096, 12, -50
cdf = O.GaussianModel(LO, 50, 3.2, 9.6, P, 32).cdf_table()
lut = O.lookup_from_cdf(cdf, P)
sym = O.synth_symbols(0xC0FFEE, 0, n_streams, N, LO, cdf, P)
cpus = os.cpu_count()
try:
    cpus = len(os.sched_getaffinity(0))
except AttributeError:
    pass
print(f"logical CPUs visible: {cpus}; cgroup cpu.max: {Path('/sys/fs/cgroup/cpu.max').read_text().strip() if Path('/sys/fs/cgroup/cpu.max').exists() else 'n/a'}", flush=True)
os.environ["CST_ORACLE_PIN"] = "1"
buf = np.zeros(1 << 30, np.uint8)
lib = O.load(True)
for th in (1, 16, 64, 128, cpus):
    lib.cst_oracle_fill_threads(buf.ctypes.data, buf.size, th)
    t0 = time.perf_counter(); lib.cst_oracle_fill_threads(buf.ctypes.data, buf.size, th); dt = time.perf_counter() - t0
    print(f"memset of 1 GiB on {th:3d} threads: {buf.size / dt / 1e9:7.1f} GB/s", flush=True)
del buf
base = {}
for th in [1, 2, 4, 8, 16, 32, 64, 128, 256, cpus]:
    if th > cpus:
        continue
    n_use = n_streams if th > 1 else 2048
    s = sym[:n_use]
    row = []
    for sink in ("0", "1"):
        os.environ["CST_ORACLE_DECODE_SINK"] = sink
        enc = O.ans_encode_batch(s, LO, cdf, P, n_threads=th, native=True)
        dec = O.ans_decode_batch(enc[0], enc[1], N, LO, cdf, P, lookup=lut, n_threads=th, native=True)
        te = td = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); O.ans_encode_batch(s, LO, cdf, P, n_threads=th, native=True, out=enc); t1 = time.perf_counter()
            O.ans_decode_batch(enc[0], enc[1], N, LO, cdf, P, lookup=lut, n_threads=th, native=True, out=dec); t2 = time.perf_counter()
            te, td = min(te, t1 - t0), min(td, t2 - t1)
        row += [s.size / te / 1e6, s.size / td / 1e6]
    os.environ["CST_ORACLE_DECODE_SINK"] = "0"
    if th == 1:
        base = {"e": row[0], "d": row[1], "ds": row[3]}
    print(f"{th:3d} threads: encode {row[0]:9.0f} Msym/s (eff {row[0] / th / base['e']:.2f})  decode {row[1]:9.0f} (eff {row[1] / th / base['d']:.2f})  "
          f"decode without its output {row[3]:9.0f} (eff {row[3] / th / base['ds']:.2f})", flush=True)

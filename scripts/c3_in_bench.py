import sys, numpy as np, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import bench
from constriction_amd import batched as B
n, k = 65536, 4096
def c3(tag, check=False):
    mu_d, sigma_d = bench.c3_parameters(bench.SEED, 0, n, k, "cuda")
    m3 = B.Model.quantized_gaussian_per_stream(-127, 127, mu_d, sigma_d, 12)
    sym3 = bench.synth_symbols_per_stream(bench.SEED, 0, k, -127, m3.cdfs_device(), 12)
    cdfs = bench.cpu_tables(-127, 127, mu_d.cpu().numpy(), sigma_d.cpu().numpy(), 12) if check else None
    e = bench.run_config(B, "C3", "ans", (32, 64, 12), m3, sym3, 5, check, cdfs, lo=-127)[0]
    print(tag, e["encode_ms"], e["decode_ms"], e.get("bit_exact"), flush=True)
c3("alone")
c3("alone again")
c3("with check", True)
c3("after check")
big = B.Model.quantized_gaussian(-350, 349, 3.2, 96.0, 16)
cdf_big = big.cdf()
sym_big = bench.synth_symbols_device(bench.SEED, 0, n, 4100, -350, torch.from_numpy(cdf_big.astype(np.int64)).cuda(), 16)
e = bench.run_config(B, "big", "ans", (32, 64, 16), big, sym_big, 5, False, None, lo=-350)[0]
print("big", e["encode_ms"], e["decode_ms"])
c3("after big (still allocated)")
del sym_big, big
c3("after big freed")

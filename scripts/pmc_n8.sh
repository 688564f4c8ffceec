#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_n8.sh <tag> -- the kernels that read / write INT8 symbol matrices themselves (round 5:
# ans_encode_pc_n8_kernel, ans_decode_n8_kernel, ans_decode_small_n8_kernel and their int16 forms) next to the int32 kernels of the same batch:
# SQ counters (two rocprofv3 --pmc passes), HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each; counters never share a run with a
# trace) and one --kernel-trace --stats pass.  65 536 x 4096 whole, with two jump points per stream, and 131 072 x 4096.
# Output: gpurun_out/<tag>_n8_counters.md (copy to profiles/).
set -u
tag=${1:-r05}
# PMC_P=24 scripts/pmc_n8.sh <tag>: the same at 12 < P <= 24 (wide producer / consumer encoders, bucket-entry decoders and their small-footprint
# forms; whole streams only) -> gpurun_out/<tag>_hp_counters.md
PMC_P=${PMC_P:-12}
name_out=n8; [ "$PMC_P" != 12 ] && name_out=hp
export TMPDIR=/tmp
R=$PWD
cat > /tmp/n8_run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import bench
from constriction_amd import batched as B
k, P = 4096, $PMC_P
m = B.Model.quantized_gaussian(-50, 50, 3.2, 9.6, P)
cdf = torch.from_numpy(m.cdf().astype(np.int64)).cuda()
for n in (65536, 131072):
    sym32 = bench.synth_symbols_device(0xC0FFEE, 0, n, k, -50, cdf, P)
    sym8, sym16 = sym32.to(torch.int8), sym32.to(torch.int16)
    for sym in (sym32, sym8, sym16):
        enc = B.ans_encode(sym, m, (32, 64, P))
        dec = torch.empty_like(sym)
        for _ in range(4):
            B.ans_encode(sym, m, (32, 64, P), out=enc)
            B.ans_decode(enc, m, k, out=dec)
        assert torch.equal(dec, sym)
        if n == 65536 and P == 12:
            pair = B.ans_encode_checkpointed(sym, m, k // 2, (32, 64, P))
            for _ in range(4):
                B.ans_encode_checkpointed(sym, m, k // 2, (32, 64, P), out=pair)
                B.ans_decode_checkpointed(pair[0], pair[1], m, k, out=dec)
            assert torch.equal(dec, sym)
            del pair
        del enc, dec
    del sym32, sym8, sym16
torch.cuda.synchronize()
PY
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
            "b SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "f FETCH_SIZE" "w WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  d=gpurun_out/${tag}_n8_$name; mkdir -p $d
  timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $R/$d -o pmc -- python /tmp/n8_run.py > /dev/null 2> $d/err.log
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
done
d=gpurun_out/${tag}_n8_stats; mkdir -p $d
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$d -o st -- python /tmp/n8_run.py > /dev/null 2> $d/err.log
find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
python - <<PY > gpurun_out/${tag}_${name_out}_counters.md
import csv, glob, collections, statistics
want = ("ans_encode_pc_kernel", "ans_encode_pc_n8_kernel", "ans_encode_pc_n16_kernel", "ans_decode_kernel<32, 64, 0, true, 1, true, 8, true>",
        "ans_decode_n8_kernel", "ans_decode_small_kernel", "ans_decode_small_n8_kernel", "ans_encode_small_kernel", "ans_decode_b16_kernel",
        "ans_decode_b16_narrow_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${tag}_n8_[abfw]/*counter_collection.csv"):
    per_dispatch = collections.defaultdict(float)
    grid = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(w in k for w in want): continue
        per_dispatch[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        grid[(k, r["Dispatch_Id"])] = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
    for (k, d, c), v in per_dispatch.items():
        agg[(k, grid[(k, d)])][c].append(v)
# wave-symbols of a launch: grid threads / 64 coder lanes ... x 4096 symbols (the encoders' grids hold as many helper as coder waves)
def streams(k, g):
    return g // 2 if "ans_encode_pc" in k else g
print("# ${tag}: the kernels that read / write int8 symbol matrices themselves, next to the int32 kernels of the same batch" + ("" if $PMC_P == 12 else " -- at P = $PMC_P (ans_decode_b16_narrow_kernel<BYTES, SMALL>: <1 | 2, false> one wave per SIMD, <1 | 2 | 4, true> the small-footprint form; ans_encode_pc_*_kernel<JUMP, true> the wide step)") + "\n")
print("scripts/pmc_n8.sh: rocprofv3 --pmc (two SQ passes, FETCH_SIZE and WRITE_SIZE in a pass each), medians over the launches of a kernel at one grid")
print("size; per symbol and wave = counter / (streams / 64 x 4096) wave-symbols (cycle counters x 4: they count in units of four cycles; the")
print("producer / consumer encoders' counters are sums over coder AND helper waves, normalised by the coder waves).  FETCH x 2 as the guide")
print("prescribes for wide streaming reads on gfx950; traffic in bytes per symbol (algorithmic: 4, 2 or 1 B per symbol + 4 B per word, 0.677 B).")
print("Template arguments: ans_decode_n8_kernel<1> / <2> = int8 / int16; ans_encode_pc_n8_kernel<false> / <true>, ..._n16_kernel<...> = without / with jump points.\n")
cols = ["SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES"]
cyc = {"SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"}
print("| kernel | streams | " + " | ".join(c.replace("SQ_", "") for c in cols) + " | FETCH x 2 B/sym | WRITE B/sym |")
print("|---|---|" + "---|" * (len(cols) + 2))
for (k, g) in sorted(agg, key=lambda t: (t[1], t[0])):
    n_str = streams(k, g)
    if n_str < 65536: continue
    norm = n_str / 64 * 4096
    cells = []
    for c in cols:
        if c not in agg[(k, g)]: cells.append("n/a"); continue
        v = statistics.median(agg[(k, g)][c])
        cells.append(f"{v:.0f}" if c == "SQ_WAVES" else f"{v * (4 if c in cyc else 1) / norm:.1f}")
    f = agg[(k, g)].get("FETCH_SIZE"); w = agg[(k, g)].get("WRITE_SIZE")
    cells.append(f"{statistics.median(f) * 1024 * 2 / (n_str * 4096):.3f}" if f else "n/a")
    cells.append(f"{statistics.median(w) * 1024 / (n_str * 4096):.3f}" if w else "n/a")
    name = k.replace("void cst::", "").replace("cst::", "").split("(")[0]
    print(f"| \`{name[:64]}\` | {n_str} | " + " | ".join(cells) + " |")
print("\n\`rocprofv3 --kernel-trace --stats\` of the same script (all launches of a name, whatever the grid):\n")
print("| kernel | calls | average us | min us | max us |\n|---|---|---|---|---|")
for fpath in glob.glob("gpurun_out/${tag}_n8_stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(fpath)):
        if any(w in r["Name"] for w in want):
            print(f"| \`{r['Name'].replace('void cst::', '').replace('cst::', '').split('(')[0][:64]}\` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} |")
PY
cat gpurun_out/${tag}_${name_out}_counters.md
rm -rf gpurun_out/${tag}_n8_a gpurun_out/${tag}_n8_b gpurun_out/${tag}_n8_f gpurun_out/${tag}_n8_w gpurun_out/${tag}_n8_stats

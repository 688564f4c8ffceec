#!/usr/bin/env python3
"""Re-checks tests/golden/reference_vectors.json against the reference checkout.

The fixture holds VALUES transcribed from the reference's tests (no reference code).  When
/root/reference is mounted (build container only; it does not exist on the GPU box) this script
verifies that every expected array literal of every vector occurs in the cited source file near
the cited lines.  Run:  python tests/golden/check_transcription.py
"""
import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent


def literal_variants(values, hexes=None):
    out = [", ".join(str(v) for v in values)]
    if hexes:
        out.append(", ".join(hexes))
    return out


def normalise(text):
    text = re.sub(r"\s+", " ", text)
    text = re.sub(r"0x([0-9A-Fa-f]{4})_([0-9A-Fa-f]{4})", r"0x\1\2", text)  # 0x1E34_22B0 -> 0x1E3422B0
    text = re.sub(r"(0x[0-9A-Fa-f]+)u(16|32)", r"\1", text)
    return text


def main():
    if not REF.exists():
        print("reference checkout not mounted; nothing to check")
        return 0
    data = json.loads((HERE / "reference_vectors.json").read_text())
    bad = 0
    for vec in data["vectors"]:
        path, _, lines = vec["source"].partition(":")
        lo, _, hi = lines.partition("-")
        lo, hi = int(lo), int(hi or lo)
        src = (REF / path).read_text().splitlines()
        window = normalise(" ".join(src[max(0, lo - 3): hi + 3]))
        checks = []
        if "expect_compressed" in vec:
            checks.append(literal_variants(vec["expect_compressed"], vec.get("expect_compressed_hex")))
        if "init" in vec:
            checks.append(literal_variants(vec["init"]["compressed"], vec["init"].get("compressed_hex")))
        for st in vec["steps"]:
            if "expect" in st and len(st["expect"]) > 1:
                checks.append(literal_variants(st["expect"]))
        for variants in checks:
            norm_window = window.lower().replace("_", "")
            tokens = set(re.findall(r"-?\d+", norm_window))
            if not any(v.lower().replace("_", "") in norm_window for v in variants) and not all(
                    t.strip() in tokens for t in variants[0].split(",")):
                print(f"MISMATCH {vec['id']}: none of {variants} found in {vec['source']}")
                bad += 1
    print(f"checked {len(data['vectors'])} vectors, {bad} mismatches")
    bad += check_perfect()
    return 1 if bad else 0


def check_perfect():
    """perfect_categorical.json: every number of every histogram / probability list must occur, in order, in the cited lines"""
    data = json.loads((HERE / "perfect_categorical.json").read_text())
    bad = 0
    for vec in data["vectors"]:
        path, _, lines = vec["source"].partition(":")
        lo, _, hi = lines.partition("-")
        src = (REF / path).read_text().splitlines()
        window = " ".join(src[int(lo) - 1: int(hi)])
        numbers = re.findall(r"\d+\.\d+e-?\d+|\d+\.\d+|\d+", window.replace("u32", ""))
        if "hist" in vec:
            want = [str(v) for v in vec["hist"]]
        else:
            want = [float(v) for v in vec["probs"]]
            numbers = [float(t) for t in numbers]
        it = iter(numbers)
        if not all(any(w == t for t in it) for w in want):
            print(f"MISMATCH {vec['id']}: values not found in order in {vec['source']}")
            bad += 1
    print(f"checked {len(data['vectors'])} perfect-quantisation vectors, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(main())

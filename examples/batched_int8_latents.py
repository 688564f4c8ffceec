#!/usr/bin/env python3
"""int8 latents through the batched ANS coder: what a learned-compression codec hands over (one stream per latent channel / tile,
symbols in -50 .. 50, a shared 12-bit quantized Gaussian) -- encoded, packed, decoded, and decoded again through two jump points
per stream (the reference's `AnsCoder.pos()` / `seek()` side information, src/stream/stack.rs:1107-1139).  The int8 matrix is read
and written by the coder loops themselves; the compressed words are those of the reference's CPU coder on the same symbols.

    python examples/batched_int8_latents.py [n_streams] [n_per_stream]
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from constriction_amd import batched as B                                  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 1024                    # (whole 128-symbol lines: the loops read int8 themselves)
lo, hi, mean, std, P = -50, 50, 3.2, 9.6, 12

rng = np.random.default_rng(1)
latents = np.clip(np.rint(mean + std * rng.standard_normal((n_streams, n_per))), lo, hi).astype(np.int8)
model = B.Model.quantized_gaussian(lo, hi, mean, std, P)
d_latents = torch.from_numpy(latents).cuda()

encoded = B.ans_encode(d_latents, model, (32, 64, P))                     # slabs: one row of words per stream
print("encoder:", B.last_kernel())
packed, offsets = B.compact(encoded)                                       # into_compressed for every stream, back to back
total = int(offsets[-1])
print(f"{n_streams} streams x {n_per} int8 symbols -> {total} words ({32 * total / latents.size:.3f} bits per symbol)")

decoded, status = B.ans_decode((packed, encoded.n_words), model, n_per, offsets=offsets, config=(32, 64, P), dtype=torch.int8)
print("decoder:", B.last_kernel())
assert int(status.abs().sum()) == 0 and torch.equal(decoded, d_latents)

# the same words through jump points: every half of a stream decodes on a lane of its own
enc2, jump = B.ans_encode_checkpointed(d_latents, model, n_per // 2, (32, 64, P))
assert torch.equal(enc2.n_words, encoded.n_words)
dec2, st2 = B.ans_decode_checkpointed(enc2, jump, model, n_per, dtype=torch.int8)
print("with two jump points per stream:", B.last_kernel())
assert int(st2.abs().sum()) == 0 and torch.equal(dec2, d_latents)

# the words do not depend on the symbol type: the int32 call on the widened latents gives the same streams
enc32 = B.ans_encode(d_latents.to(torch.int32), model, (32, 64, P))
used = torch.arange(enc32.words.shape[1], device="cuda")[None, :] < enc32.n_words[:, None]
assert torch.equal(enc32.n_words, encoded.n_words) and bool(((enc32.words == encoded.words) | ~used).all())
print("ok: the int8 call's words are the int32 call's")

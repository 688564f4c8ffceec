"""Helpers that turn an entry of tests/golden/reference_vectors.json into model objects.

`backend` is any module/object exposing GaussianModel / TableModel / categorical_fast_cdf with the
oracle's signatures (oracle.oracle itself, or the product's host-side equivalents in the GPU tests).
"""
import numpy as np


def _params(values, dtype):
    a = np.asarray(values, dtype=np.float32 if dtype == "f32" else np.float64)
    return a.astype(np.float64)  # f32 parameters are widened before use (pybindings/mod.rs:211-216)


def models_for(step, P, backend, prob_bits=32):
    """Returns (models, n): `models` is one model (iid) or a per-symbol list."""
    m = step["model"]
    kind = m["kind"]
    n = step.get("n")
    if kind == "gaussian":
        if "means" in m:
            means, stds = _params(m["means"], m.get("dtype", "f64")), _params(m["stds"], m.get("dtype", "f64"))
            return [backend.GaussianModel(m["lo"], m["hi"], mu, sd, P, prob_bits) for mu, sd in zip(means, stds)], len(means)
        return backend.GaussianModel(m["lo"], m["hi"], m["mean"], m["std"], P, prob_bits), n
    if kind == "categorical_fast":
        probs = np.asarray(m["probs"], dtype=np.float32 if m["dtype"] == "f32" else np.float64)
        return backend.TableModel(backend.categorical_fast_cdf(probs, P), 0, P), n
    if kind == "categorical_fast_rows":
        rows = np.asarray(m["probs"], dtype=np.float32 if m["dtype"] == "f32" else np.float64)
        return [backend.TableModel(backend.categorical_fast_cdf(r, P), 0, P) for r in rows], len(rows)
    if kind == "table":
        return backend.TableModel(np.asarray(m["cdf"], dtype=np.uint32), m["lo"], P), n
    raise ValueError(kind)

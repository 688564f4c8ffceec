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
    if kind == "categorical_lazy":          # Categorical(probs, lazy=True): nothing tabulated (lazy_contiguous.rs:228-331)
        probs = np.asarray(m["probs"], dtype=np.float32 if m["dtype"] == "f32" else np.float64)
        return (backend.LazyCategoricalModel(probs, P) if hasattr(backend, "LazyCategoricalModel")
                else backend.TableModel(backend.categorical_fast_cdf(probs, P), 0, P)), n
    if kind == "categorical_lazy_rows":
        rows = np.asarray(m["probs"], dtype=np.float32 if m["dtype"] == "f32" else np.float64)
        mk = (lambda r: backend.LazyCategoricalModel(r, P)) if hasattr(backend, "LazyCategoricalModel") else \
             (lambda r: backend.TableModel(backend.categorical_fast_cdf(r, P), 0, P))
        return [mk(r) for r in rows], len(rows)
    if kind == "table":
        return backend.TableModel(np.asarray(m["cdf"], dtype=np.uint32), m["lo"], P), n
    if kind in ("scipy_norm", "scipy_norm_family"):
        import scipy.stats
        if kind == "scipy_norm":
            return backend.TableModel(leaky_table(scipy.stats.norm(m["loc"], m["scale"]).cdf, m["lo"], m["hi"], P), m["lo"], P), n
        return [backend.TableModel(leaky_table(lambda x, a=a, b=b: scipy.stats.norm.cdf(x, a, b), m["lo"], m["hi"], P), m["lo"], P)
                for a, b in zip(m["locs"], m["scales"])], len(m["locs"])
    raise ValueError(kind)


def leaky_table(cdf, lo, hi, P):
    """LeakyQuantizer over an arbitrary CDF (src/stream/model/quantize.rs:525-568), tabulated:
    L[0] = 0, L[i] = trunc_sat(fw * cdf(lo + i - 0.5)) + i, L[n] = 2^P with fw = f64((2^P - 1) - (hi - lo))."""
    n = hi - lo + 1
    fw = float(((1 << P) - 1) - (hi - lo))
    out = np.zeros(n + 1, dtype=np.uint32)
    for i in range(1, n):
        x = fw * float(cdf(lo + i - 0.5))
        out[i] = (min(max(int(x), 0), 0xFFFFFFFF) if x == x else 0) + i
    out[n] = 1 << P
    return out

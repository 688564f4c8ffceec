"""Drop-in mirror of `constriction.stream` for the hot path: `stack.AnsCoder`, `queue.RangeEncoder/RangeDecoder`, `chain.ChainCoder`
and the `model` families that path uses (QuantizedGaussian, Categorical(perfect=False)).  Same call signatures,
return dtypes and error types as the reference's Python API (src/pybindings/stream/), computed on the MI355X."""
from . import chain, model, queue, stack  # noqa: F401

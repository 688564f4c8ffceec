"""Kernel names as cst_last_kernel_name reports them, for tests that pin the dispatch."""


def with_jump(name: str, enc) -> str:
    """the checkpointing form of encoder kernel `name` if batch `enc` carries jump points (the default encode calls take them where
    they pay: batched.ans_encode(..., jump_points="auto")), else `name`"""
    if getattr(enc, "jump", None) is None:
        return name
    return name[:-1] + ", ckpt>" if name.endswith(">") else name + "<ckpt>"

"""MI355X-native FlashAttention (forward + backward) behind the reference's ``flash_attention(q,k,v,causal)``
entry point (santiweide/flash-attention-impls, code/triton_fa2/FA2-triton.py:240)."""
from .flash_attn import (FlashAttnArgumentError, FlashAttnFn, check_args, flash_attention, flash_attn,
                         load_library)

__all__ = ["flash_attn", "flash_attention", "FlashAttnArgumentError", "FlashAttnFn", "check_args", "load_library"]


def _header_version() -> str:
    """"0.1.NN" from FA_VERSION (1NN) of include/fa_mi355.h -- the one place the version is written."""
    import os
    import re
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "fa_mi355.h")) as f:
            v = int(re.search(r"#define\s+FA_VERSION\s+(\d+)", f.read()).group(1))
        return f"{v // 10000}.{v // 100 % 100}.{v % 100}"
    except (OSError, AttributeError):
        return "0.0.0"


__version__ = _header_version()

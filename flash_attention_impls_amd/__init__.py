"""MI355X-native FlashAttention (forward + backward) behind the reference's ``flash_attention(q,k,v,causal)``
entry point (santiweide/flash-attention-impls, code/triton_fa2/FA2-triton.py:240)."""
from .flash_attn import (FlashAttnArgumentError, FlashAttnFn, check_args, flash_attention, flash_attn,
                         load_library)

__all__ = ["flash_attn", "flash_attention", "FlashAttnArgumentError", "FlashAttnFn", "check_args", "load_library"]
__version__ = "0.1.0"

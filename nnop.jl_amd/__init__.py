"""nnop.jl_amd -- MI355X-native Flash Attention behind NNop.jl's operator API.

Holds only what the hot path needs: ``csrc/`` (hand-written gfx950 HIP kernels + the C ABI of
``include/nnop_hip.h``), the host-side mirror of the reference interface (``attention.py``; ``rope.py`` for the
Llama rotary embedding applied to q, k right before attention -- SURVEY.md section 8(f) rank 2), the
(batch, kv-head) sharding used for multi-GPU runs (``shard.py``) and the Julia package-extension
shim (``julia/``, source only: no Julia in this image).

The directory name contains a dot, so import it through ``__graft_entry__.load_package()``
(registers it as module ``nnop_jl_amd``).
"""
from .attention import (NNopError, flash_attention, _flash_attention, grad_flash_attention,
                        shared_memory, bwd_workspace_bytes, fa_fwd_into, fa_bwd_into)
from .rope import LlamaRotaryEmbedding, llama_rope, _llama_rope, grad_llama_rope, llama_rope_into
from .softmax import online_softmax, grad_online_softmax, online_softmax_into
from .norms import rms_norm, _rms_norm, grad_rms_norm, layer_norm, _layer_norm, grad_layer_norm
from . import _lib, shard, workmodel

__all__ = ["NNopError", "flash_attention", "_flash_attention", "grad_flash_attention",
           "shared_memory", "bwd_workspace_bytes", "fa_fwd_into", "fa_bwd_into", "LlamaRotaryEmbedding", "llama_rope", "_llama_rope", "grad_llama_rope", "llama_rope_into",
           "online_softmax", "grad_online_softmax", "online_softmax_into",
           "rms_norm", "_rms_norm", "grad_rms_norm", "layer_norm", "_layer_norm", "grad_layer_norm",
           "shard", "workmodel", "_lib"]
__version__ = "0.1.0"

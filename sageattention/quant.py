"""``sageattention.quant`` of the reference (sageattention/quant.py) -> the gfx950 implementation."""
from sageattention_amd.quant import (per_block_int8, per_warp_int8, per_thread_int8, sub_mean, per_channel_fp8,  # noqa: F401
                                     k_mean, fp8_token_order)

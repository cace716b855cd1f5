"""``sageattention.core`` of the reference (sageattention/core.py) -> the gfx950 implementation."""
from sageattention_amd.core import *  # noqa: F401,F403
from sageattention_amd.core import dispatch_pv  # noqa: F401
from sageattention_amd.ring import ring_sageattn  # noqa: F401
from sageattention_amd.ulysses import ulysses_sageattn  # noqa: F401

"""uavppo -- MI355X-native PPO hot path (rollout, GAE, clipped-PPO update) behind the
reference's config / environment / model surface.  All arithmetic is in libuavppo.so (HIP,
gfx950); importing this package does not load the library, the first op call does."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]

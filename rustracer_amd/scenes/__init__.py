"""Deterministic synthetic scenes standing in for the reference's LFS-only .pbrt scenes
(S1 cornell, S2 blob-1M, S3 mis-plates, S4 room-env)."""
from .cornell import cornell_box  # noqa: F401
from .soup import random_soup  # noqa: F401
from .blob import blob_scene  # noqa: F401
from .mis import mis_plates  # noqa: F401
from .room import room_env  # noqa: F401
from .forest import forest  # noqa: F401

"""mapalign — MI355X-native point-cloud map alignment (drop-in for the reference's
``src/map_alignment`` path).  Compute lives in ``lib/libmapalign.so``
(hand-written HIP for gfx950); this package is the thin ctypes host side.
"""
from ._lib import MapAlignError, MapAlignLibraryError  # noqa: F401
from .api import GoldMap, align, align_many, release_cached_engines  # noqa: F401
from .engine import AlignResult, MapAligner  # noqa: F401
from .alignment import Alignment  # noqa: F401
from .boxes import BBoxComparison  # noqa: F401

__all__ = ["align", "align_many", "GoldMap", "release_cached_engines", "Alignment", "BBoxComparison", "MapAligner", "AlignResult",
           "MapAlignError", "MapAlignLibraryError"]

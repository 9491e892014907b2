"""instant-distance_amd — MI355X (gfx950) HNSW build+search engine behind
instant-distance's Builder / Hnsw / HnswMap / Search API.

The directory name is fixed by the repo contract; import it as
`instant_distance_amd` (the alias module at the repo root).
"""
from ._capi import (INVALID, M, M2, METRIC_L2, METRIC_L2SQ, MAX_EF, TIES_DROP, TIES_STRICT, IdistError, LIB_PATH)
from .api import (BatchResult, Builder, Heuristic, Hnsw, HnswMap, Item, MapItem, PointId, Search)

__all__ = ["Builder", "Heuristic", "Hnsw", "HnswMap", "Search", "Item", "MapItem", "PointId", "BatchResult",
           "IdistError", "INVALID", "M", "M2", "METRIC_L2", "METRIC_L2SQ", "MAX_EF", "TIES_DROP", "TIES_STRICT", "LIB_PATH"]

"""instant-distance_amd — MI355X (gfx950) HNSW build+search engine behind
instant-distance's Builder / Hnsw / HnswMap / Search API.

The directory name is fixed by the repo contract; import it as
`instant_distance_amd` (the alias module at the repo root).
"""
# One `Search` per host thread = one HIP stream per thread, and the HIP runtime multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4, read ONCE when the runtime starts): at 16 threads that is ~6k instead of ~14k
# scalar calls/s.  That is a process-wide setting of the HOST's: neither libidist.so nor this package touches the environment —
# an application that searches from many threads exports GPU_MAX_HW_QUEUES=16 before its first HIP call (bench.py does so before
# it imports torch).
from ._capi import (INVALID, M, M2, METRIC_L2, METRIC_L2SQ, MAX_EF, TIES_DROP, TIES_STRICT, IdistError, LIB_PATH)
from .api import (BatchResult, Builder, Heuristic, Hnsw, HnswMap, Item, MapItem, PointId, Search)

__all__ = ["Builder", "Heuristic", "Hnsw", "HnswMap", "Search", "Item", "MapItem", "PointId", "BatchResult",
           "IdistError", "INVALID", "M", "M2", "METRIC_L2", "METRIC_L2SQ", "MAX_EF", "TIES_DROP", "TIES_STRICT", "LIB_PATH"]

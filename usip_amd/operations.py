"""Live helpers of the reference's models/operations.py (its numba kernels are commented out
there): neighbour gathers and the two shared-memory sizing constants."""
from . import functional as Fh

# models/operations.py:16-18.  The reference's CUDA kernels need batch <= 24 threads and
# node_num <= 512 shared-memory slots; the gfx950 kernels have no such limit, the names stay
# because models/keypoint_detector.py:99-102 reads them.
CUDA_SHARED_MEM_DIM_X = 24
CUDA_SHARED_MEM_DIM_Y = 512


def knn_gather_by_indexing(som_node, som_node_knn_I):
    """som_node BxCxN, som_node_knn_I BxNxK -> BxCxNxK (models/operations.py:271-287)."""
    return Fh.gather_neighbours(som_node, som_node_knn_I)


def knn_gather_wrapper(som_node, som_node_knn_I):
    """models/operations.py:243-268 (C must be 3 there)."""
    assert som_node.size(1) == 3
    return Fh.gather_neighbours(som_node, som_node_knn_I)

"""text2pos-cvpr2022_amd: MI355X-native coarse cell-retrieval forward path of Text2Pos (CVPR 2022).

Layout: csrc/ (hand-written gfx950 HIP kernels + the C ABI of include/t2p.h), _lib.py / ops.py (ctypes binding),
and the host-side mirror of the reference's model interface (cell_retrieval.py, object_encoder.py, pointnet2.py,
modules.py, data.py, superglue_matcher.py for the fine stage), retrieval.py (top-k), distributed.py (cell sharding over RCCL), synthetic.py (bench inputs).

The directory name contains '-', so import it through the `text2pos_amd` alias module at the repository root.
"""
from .cell_retrieval import CellRetrievalNetwork  # noqa: F401
from .losses import HardestRankingLoss, PairwiseRankingLoss  # noqa: F401
from .modules import LanguageEncoder, get_mlp  # noqa: F401
from .object_encoder import ObjectEncoder  # noqa: F401
from .pointnet2 import PointNet2  # noqa: F401
from .retrieval import retrieve_topk  # noqa: F401
from .superglue_matcher import SuperGlueMatch, get_pos_in_cell  # noqa: F401

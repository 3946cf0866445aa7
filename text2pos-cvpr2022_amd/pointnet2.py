"""Parameter containers mirroring models/pointcloud/pointnet2.py (PointNet2 with three set-abstraction layers, a
global-abstraction layer and two linear heads) so that reference state_dicts load unchanged.  The arithmetic of
`forward` lives in libt2p_hip.so (csrc/sample_group.hip, csrc/ws_gemm.hip); PointNet2 is only ever run as part
of CellRetrievalNetwork.encode_objects, batched over all objects of all cells.
"""
import torch.nn as nn

from .modules import get_mlp


class PointConv(nn.Module):
    """Holds `local_nn` under the key torch_geometric.nn.PointConv uses (`point_conv.local_nn.*`)."""

    def __init__(self, local_nn):
        super().__init__()
        self.local_nn = local_nn


class SetAbstractionLayer(nn.Module):
    def __init__(self, ratio, radius, mlp):
        super().__init__()
        self.ratio = ratio
        self.radius = radius
        self.point_conv = PointConv(local_nn=mlp)


class GlobalAbstractionLayer(nn.Module):
    def __init__(self, mlp):
        super().__init__()
        self.mlp = mlp


class PointNet2(nn.Module):
    def __init__(self, num_classes, num_colors, args):
        super().__init__()
        assert args.pointnet_layers == 3 and args.pointnet_variation == 0  # models/pointcloud/pointnet2.py:55
        self.sa1 = SetAbstractionLayer(0.5, 0.2, get_mlp([3 + 3, 32, 64]))
        self.sa2 = SetAbstractionLayer(0.5, 0.3, get_mlp([64 + 3, 128, 128]))
        self.sa3 = SetAbstractionLayer(0.5, 0.4, get_mlp([128 + 3, 256, 256]))
        self.ga = GlobalAbstractionLayer(get_mlp([256 + 3, 512, 1024]))
        self.lin1 = nn.Linear(1024, 512)
        self.lin2 = nn.Linear(512, 256)
        # heads kept for state_dict compatibility; their outputs are unused on the retrieval path
        self.class_classifier = nn.Linear(256, num_classes)
        self.color_classifier = nn.Linear(256, num_colors)
        self.dim0, self.dim1, self.dim2 = 1024, 512, 256

    @property
    def radii(self):
        return (self.sa1.radius, self.sa2.radius, self.sa3.radius)

    def forward(self, data):
        raise NotImplementedError(
            "PointNet2 runs fused inside CellRetrievalNetwork.encode_objects (HIP); it has no stand-alone forward")

    @property
    def device(self):
        return next(self.lin1.parameters()).device

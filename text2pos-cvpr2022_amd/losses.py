"""Retrieval training loss of the reference on the HIP path: `PairwiseRankingLoss`
(training/losses.py:126-164; `--margin 0.35`, training/args.py:46; constructed at training/coarse.py:279-282).
Same call `criterion(anchor, positive)`; the reference's hard-coded `.cuda()` is gone (tensors stay on their device).
The hinge terms, their sum and the gradient with respect to the score matrix come from t2p_pairwise_ranking
(csrc/small_kernels.hip); the score matrix and its two gradient products run on the library's own fp32-MFMA GEMM (ops.matmul).
`HardestRankingLoss` (training/losses.py:167-201, --ranking_loss hardest) shares the wrapper on t2p_hardest_ranking."""
import torch
import torch.nn as nn

from . import ops


class _PairwiseRankingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s, margin, hardest=False):
        n_im = torch.norm(im.detach(), dim=1, keepdim=True)
        n_s = torch.norm(s.detach(), dim=1, keepdim=True)
        im_n, s_n = (im.detach() / n_im).contiguous(), (s.detach() / n_s).contiguous()
        scores = ops.matmul(im_n, s_n.t().contiguous())
        terms, d_scores = (ops.hardest_ranking if hardest else ops.pairwise_ranking)(scores, margin)
        ctx.save_for_backward(im_n, s_n, n_im, n_s, d_scores)
        return terms.sum() / im.shape[0]

    @staticmethod
    def backward(ctx, g):
        im_n, s_n, n_im, n_s, d_scores = ctx.saved_tensors
        d_imn, d_sn = ops.matmul(d_scores, s_n), ops.gemm_tn(d_scores, im_n)
        # x / |x|: d x = (d x_n - x_n <x_n, d x_n>) / |x|
        d_im = (d_imn - im_n * (im_n * d_imn).sum(1, keepdim=True)) / n_im
        d_s = (d_sn - s_n * (s_n * d_sn).sum(1, keepdim=True)) / n_s
        return g * d_im, g * d_s, None, None


class PairwiseRankingLoss(nn.Module):
    def __init__(self, margin: float = 1.0):
        super().__init__()
        self.margin = margin

    def forward(self, im, s):
        if im.shape != s.shape or im.dim() != 2:
            raise RuntimeError("PairwiseRankingLoss: anchor and positive must both be [B, D]")
        return _PairwiseRankingFn.apply(im, s, float(self.margin))


class HardestRankingLoss(nn.Module):
    """training/losses.py:167-201 (--ranking_loss hardest): only the hardest negative of every anchor / positive counts."""

    def __init__(self, margin: float = 1.0):
        super().__init__()
        self.margin = margin

    def forward(self, images, captions):
        if images.shape != captions.shape or images.dim() != 2:
            raise RuntimeError("HardestRankingLoss: images and captions must both be [B, D]")
        return _PairwiseRankingFn.apply(images, captions, float(self.margin), True)

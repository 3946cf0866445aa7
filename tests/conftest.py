import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def vocab():
    import text2pos_amd  # noqa: F401
    from text2pos_amd import synthetic as S
    return dict(classes=S.LABELS + ["pad"], colors=S.COLOR_NAMES, words=S.known_words())


@pytest.fixture(scope="session")
def oracle_model(vocab):
    """CPU oracle with the deterministic golden weights (seed 11, as tests/golden/make_golden.py)."""
    import weights as W
    from oracle import model as OM
    m = OM.OracleCellRetrieval(vocab["classes"], vocab["colors"], vocab["words"], OM.default_args()).eval()
    W.fill_state_dict(m, 11)
    return m


@pytest.fixture(scope="session")
def hip_model(vocab, oracle_model):
    """The product module on cuda:0 carrying the same weights (state_dict interchange)."""
    import torch
    import text2pos_amd as t2p
    from text2pos_amd import synthetic as S
    m = t2p.CellRetrievalNetwork(vocab["classes"], vocab["colors"], vocab["words"], S.default_args())
    m.load_state_dict(oracle_model.state_dict(), strict=True)
    return m.to("cuda:0").eval()


@pytest.fixture(scope="session")
def trained_checkpoint(tmp_path_factory):
    """A TRAINED coarse checkpoint in the reference's file format, made on this box by the repo's own training path
    (train_checkpoint.py: 320 Adam steps of training/coarse.py:31-62 on synthetic (description, cell) pairs, then
    `torch.save(model, path)`).  Returns (path, info: epoch losses, hit@k on held-out pairs before / after)."""
    import train_checkpoint as TC
    path = str(tmp_path_factory.mktemp("ckpt") / "coarse_trained.pth")
    model, info = TC.trained_model(path)
    del model
    return path, info


def fine_args(num_layers):
    from oracle import model as OM
    return OM.default_args(embed_dim=128, num_layers=num_layers, sinkhorn_iters=50)


def make_fine_pair(vocab, num_layers, seed, device=None):
    """(product SuperGlueMatch with deterministic weights, CPU oracle carrying the same weights).  The product module
    has the reference's parameter names and shapes, so tests/golden/weights.py fills it exactly as it filled the
    reference model when the fixture was generated; the oracle copies from its state_dict."""
    import weights as W
    import text2pos_amd as t2p
    from oracle import fine as OF
    args = fine_args(num_layers)
    prod = t2p.SuperGlueMatch(vocab["classes"], vocab["colors"], vocab["words"], args).eval()
    W.fill_state_dict(prod, seed)
    sd = prod.state_dict()
    orc = OF.OracleSuperGlueMatch(vocab["classes"], vocab["colors"], vocab["words"], args).eval()
    own = orc.state_dict()
    orc.load_state_dict({k: sd[k] for k in own if not k.startswith("superglue.")}, strict=False)
    orc.superglue.load_reference_state(sd)
    if device is not None:
        prod = prod.to(device)
    return prod, orc


@pytest.fixture(scope="session")
def fine_pair_cpu(vocab):
    return make_fine_pair(vocab, 2, 14)


@pytest.fixture(scope="session")
def fine_pair_gpu(vocab):
    return make_fine_pair(vocab, 2, 14, "cuda:0")

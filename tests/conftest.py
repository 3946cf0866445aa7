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

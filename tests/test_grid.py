import numpy as np
import torch

from gshell_amd import grid


def test_bcc_counts_match_survey():
    v, t = grid.bcc_grid(26)
    assert v.shape == (37259, 3) and t.shape == (202800, 4)   # SURVEY.md 8d "res64"
    assert int(t.min()) >= 0 and int(t.max()) == v.shape[0] - 1


def test_tets_positively_oriented_and_unique():
    for v, t in (grid.bcc_grid(5), grid.kuhn_grid(4)):
        p = v[t.reshape(-1)].reshape(-1, 4, 3).double()
        vol = (torch.linalg.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 0])).sum(-1)
        assert bool((vol > 0).all())
        key = torch.sort(t, dim=1)[0]
        assert torch.unique(key, dim=0).shape[0] == t.shape[0]


def test_kuhn_fills_the_cube():
    v, t = grid.kuhn_grid(3)
    p = v[t.reshape(-1)].reshape(-1, 4, 3).double()
    vol = (torch.linalg.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 0])).sum(-1) / 6
    assert abs(float(vol.sum()) - 1.0) < 1e-9


def test_npz_layout_matches_reference_loader(tmp_path):
    v, t = grid.bcc_grid(3)
    path = tmp_path / "8_tets.npz"
    grid.save_npz(str(path), v, t)
    d = np.load(path)
    assert d["vertices"].dtype == np.float32 and d["indices"].dtype == np.int64

"""Checkpoint / resume and the validation loop at the edge of the path."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_checkpoint_resume_and_validate(tmp_path):
    from gshell_amd import workload
    from gshell_amd.render import obj
    from gshell_amd.train import validate
    tr = workload.build(res=24, n_samples=2, batch=2, train_res=(64, 64), fit_steps=60)
    target = workload.make_targets(tr, [0, 7], (64, 64))
    for _ in range(2):
        tr.step(target)
    ck = str(tmp_path / "ck.pt")
    tr.save_checkpoint(ck)
    saved = [p.detach().clone() for p in tr.all_params()]
    m_saved = tr.opt_mesh.state_dict()['state'][0]['exp_avg'].clone()
    tr.step(target)
    assert any(not torch.equal(a, b) for a, b in zip(saved, tr.all_params()))
    tr2 = workload.build(res=24, n_samples=2, batch=2, train_res=(64, 64), fit_steps=0, seed=5)
    tr2.load_checkpoint(ck)
    assert tr2.it == 2
    for a, b in zip(saved, tr2.all_params()):
        assert torch.equal(a, b)
    assert torch.equal(m_saved, tr2.opt_mesh.state_dict()['state'][0]['exp_avg'])
    assert abs(tr2.scheds[0].get_last_lr()[0] - tr.opt_mat.param_groups[0]['initial_lr'] * 10 ** (-2 * 0.0002)) < 1e-9
    # the resumed state renders the same picture as the state that was saved (up to MC/atomic noise): validate both ways
    singles = [{k: (v[i:i + 1] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in target.items()} for i in range(2)]
    psnr = validate(tr2, singles, out_dir=str(tmp_path / "val"))
    assert 5.0 < psnr < 80.0
    lines = open(tmp_path / "val" / "metrics.txt").read().splitlines()
    assert lines[0] == "ID, MSE, PSNR" and len(lines) == 4 and lines[-1].startswith("AVERAGES:")
    l2, _ = tr2.step(target)
    assert torch.isfinite(l2)
    # exported mesh reloads with the same topology
    m = tr2.geometry.getMesh(tr2.mat)['imesh']
    path = obj.write_obj(str(tmp_path / "mesh"), m, save_material=True)
    back = obj.load_obj(path)
    assert torch.equal(back.t_pos_idx, m.t_pos_idx) and torch.allclose(back.v_pos, m.v_pos.detach(), atol=1e-6)
    assert os.path.exists(tmp_path / "mesh" / "mesh.mtl")

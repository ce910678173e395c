"""Host logic of GShellTetsGeometry.tick that needs no GPU: the loss-term groups of view-sharded training and the constant weight
vectors over regularizer.frame_sums."""
import types

import torch


def test_loss_term_groups_are_summed_on_demand():
    from gshell_amd.geometry.gshell_tets_geometry import _LossTerms
    calls = []

    def total(ts):
        calls.append(len(ts))
        return sum(ts, torch.zeros(()))
    a, b, c = torch.tensor(1.0), torch.tensor(2.0), torch.tensor(4.0)
    t = _LossTerms({'per_view': [a, b], 'global': [c], 'presharded': []}, total)
    assert calls == []                                     # nothing is added until a group is asked for
    assert float(t['per_view']) == 3.0 and float(t['global']) == 4.0 and float(t['presharded']) == 0.0
    assert t.get('missing') is None and float(t.get('global')) == 4.0
    t.add('global', torch.tensor(8.0))                     # G-FlexiCubes appends its regulariser
    assert float(t['global']) == 12.0
    t['presharded'] = torch.tensor(0.5)
    assert float(t['presharded']) == 0.5


def test_frame_sum_weight_vectors():
    from gshell_amd.geometry.gshell_tets_geometry import GShellTetsGeometry
    FL = types.SimpleNamespace(lambda_diffuse=0.15, lambda_kd=0.1, lambda_ks=0.05, lambda_nrm=0.025)
    g = types.SimpleNamespace(FLAGS=FL, __dict__={})
    obj = types.SimpleNamespace(FLAGS=FL)
    w_img, w_reg = GShellTetsGeometry._frame_sum_weights(obj, torch.device("cpu"), 1000.0, True, True, True)
    assert w_img.shape == (10,) and w_reg.shape == (10,)
    fs = torch.arange(1.0, 11.0)
    n = 1000.0
    want_img = fs[0] / n + 0.5 * (fs[1] + fs[2]) / n + fs[9] / (3 * n)
    want_reg = fs[3] / n * 0.15 + fs[6] / n * 0.1 + fs[7] / (3 * n) * 0.05 + fs[8] / (3 * n) * 0.025
    assert torch.allclose(torch.dot(fs, w_img), want_img) and torch.allclose(torch.dot(fs, w_reg), want_reg)
    w_img9, w_reg9 = GShellTetsGeometry._frame_sum_weights(obj, torch.device("cpu"), 1000.0, False, False, False)
    assert w_img9.shape == (9,) and float(w_img9[1]) == 0.0 and float(w_reg9[3]) == 0.0
    assert GShellTetsGeometry._frame_sum_weights(obj, torch.device("cpu"), 1000.0, True, True, True)[0] is w_img      # cached

"""gs_shade_assemble (one kernel for shade()'s buffer dictionary + the background composite) against the op-by-op torch
formulation it replaces (reference render/render.py:74, :105-112, :160-186, :352-359, :417-433): same frame, same gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("denoise", [True, False])
def test_fused_assembly_equals_the_torch_formulation(denoise):
    from gshell_amd import workload
    from gshell_amd.render import render
    torch.manual_seed(0)       # the texture MLP is initialised from the global generator: same scene whatever ran before this test
    tr = workload.build(res=16, n_samples=2, batch=2, train_res=(48, 56), fit_steps=40, denoiser='bilateral' if denoise else 'none')
    target = workload.make_targets(tr, [0, 5], (48, 56))
    g = torch.Generator(device="cuda").manual_seed(0)
    results = []
    for fused in (True, False):
        tr.FLAGS.fused_assemble = fused
        render.rnd_seed = 11
        tr.FLAGS.noise_stream.set_iteration(3)
        for p in tr.all_params():
            p.grad = None
        d = tr.geometry.render(tr.glctx, target, tr.lgt, tr.mat, denoiser=tr.denoiser, shadow_scale=0.7)
        st, keys, sizes = d['buffers'].stacked
        if not results:
            w = torch.rand(st.shape, device="cuda", generator=g)
        (st * w).sum().backward()
        results.append((st.detach().clone(), keys, sizes, [None if p.grad is None else p.grad.clone() for p in tr.all_params()]))
    (a, ka, sa, ga), (b, kb, sb, gb) = results
    assert ka == kb and sa == sb and a.shape[-1] == 45
    # same arithmetic up to fma contraction / division order: a few ulp of O(1) radiance values (measured max |diff| 1.1e-6)
    assert torch.allclose(a, b, rtol=1e-5, atol=5e-6)
    assert float((a[..., 3] > 0).float().mean()) > 0.02          # something is covered
    for x, y in zip(ga, gb):
        assert (x is None) == (y is None)
        if x is not None:
            assert float((x - y).norm()) <= 1e-4 * float(y.norm()) + 1e-12


def test_tick_with_the_colour_term_in_the_frame_pass_equals_the_separate_image_loss():
    """FLAGS.fused_image_loss: tick folds loss_fn's colour term into regularizer.frame_sums (one consumer, one gradient tensor of
    the frame) when the loss object names its (loss, tonemapper); same losses (1e-6) and parameter gradients (1e-4: float-atomic order) as calling
    loss_fn on the masked colours."""
    from gshell_amd import workload
    from gshell_amd.render import render
    torch.manual_seed(0)
    tr = workload.build(res=16, n_samples=2, batch=2, train_res=(48, 56), fit_steps=40)
    target = workload.make_targets(tr, [1, 4], (48, 56))
    res = []
    for fused in (True, False):
        tr.FLAGS.fused_image_loss = fused
        render.rnd_seed = 5
        tr.FLAGS.noise_stream.set_iteration(7)
        for p in tr.all_params():
            p.grad = None
        tr.lgt.update_pdf()
        img_loss, _, reg_loss = tr.geometry.tick(tr.glctx, target, tr.lgt, tr.mat, tr.loss_fn, 1200, denoiser=tr.denoiser)
        (img_loss + reg_loss).backward()
        res.append((float(img_loss), float(reg_loss), [None if p.grad is None else p.grad.clone() for p in tr.all_params()]))
    (ia, ra, ga), (ib, rb, gb) = res
    assert abs(ia - ib) <= 1e-6 * abs(ib) + 1e-9 and abs(ra - rb) <= 1e-6 * abs(rb) + 1e-9
    for x, y in zip(ga, gb):
        assert (x is None) == (y is None)
        if x is not None:
            assert float((x - y).norm()) <= 1e-4 * float(y.norm()) + 1e-12      # float-atomic order differs between the two runs

"""Fused RealNVP log-density kernels (forward + backward) against the CPU oracle's restatement of
real_nvp.py `log_prob`, evaluated in float64 with torch autograd. GPU only."""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _flow(dim, seed):
    from das_amd.pose_heads import RealNVP, RealNVP2D
    torch.manual_seed(seed)
    flow = (RealNVP if dim == 3 else RealNVP2D)()
    # default nn.Linear init gives a nearly-identity flow; widen it so that every term matters
    for i, p in enumerate(flow.parameters()):
        p.data = cases.randn(500 + 7 * i + seed, *p.shape) * (0.15 if p.dim() == 2 else 0.2)
    return flow


def _oracle(flow, x64):
    from oracle.loss import realnvp_log_prob
    sd = {'f.' + k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point and k != 'mask')
          for k, v in flow.state_dict().items()}
    sd['f.mask'] = flow.mask.double()
    x = x64.clone().requires_grad_(True)
    lp = realnvp_log_prob(sd, 'f', x)
    return lp, x, sd


@pytest.mark.parametrize('dim,N', [(3, 1), (3, 300), (3, 1000), (2, 517)])
def test_realnvp_log_prob_forward_backward(dim, N):
    from das_amd.train_ops import realnvp_log_prob
    flow = _flow(dim, N)
    x = cases.randn(900 + N, N, dim) * 1.5
    lp_ref, xr, sd = _oracle(flow, x.double())
    gout = cases.randn(901 + N, N)
    lp_ref.backward(gout.double())

    flow = flow.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    lp = realnvp_log_prob(flow, xd)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), lp_ref.detach().numpy(), rtol=2e-5, atol=2e-4)
    lp.backward(gout.to(DEV))
    scale = float(xr.grad.abs().max())
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) < 2e-4 * max(scale, 1.0)
    for k, p in flow.named_parameters():
        ref = sd['f.' + k].grad
        err = float((p.grad.cpu().double() - ref).abs().max())
        assert err < 3e-4 * max(float(ref.abs().max()), 1.0), (k, err, float(ref.abs().max()))


def test_realnvp_two_flows_in_one_launch():
    """Two flows with different weights and row counts (second job starts on a padded 256-row boundary):
    values and all gradients equal the one-flow-per-launch results."""
    from das_amd.train_ops import realnvp_log_prob, realnvp_log_prob_multi
    fa, fb = _flow(3, 11).to(DEV), _flow(3, 12).to(DEV)
    xa = (cases.randn(80, 300, 3) * 1.5).to(DEV).requires_grad_(True)
    xb = (cases.randn(81, 777, 3) * 1.5).to(DEV).requires_grad_(True)
    ga, gb = cases.randn(82, 300).to(DEV), cases.randn(83, 777).to(DEV)
    la, lb = realnvp_log_prob_multi([(fa, xa), (fb, xb)])
    (la * ga).sum().backward(retain_graph=True)
    (lb * gb).sum().backward()
    got = [la.detach().clone(), lb.detach().clone(), xa.grad.clone(), xb.grad.clone()] + \
        [p.grad.clone() for f in (fa, fb) for p in f.parameters()]
    for t in (xa, xb):
        t.grad = None
    for f in (fa, fb):
        f.zero_grad()
    ra, rb = realnvp_log_prob(fa, xa), realnvp_log_prob(fb, xb)
    ra.backward(ga)
    rb.backward(gb)
    ref = [ra.detach(), rb.detach(), xa.grad, xb.grad] + [p.grad for f in (fa, fb) for p in f.parameters()]
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-4)


def test_realnvp_matches_the_torch_path_in_the_loss():
    """Same flow, same input: the fused op equals the layer-by-layer torch evaluation the loss used before."""
    from das_amd.losses import realnvp_log_prob_torch
    from das_amd.train_ops import realnvp_log_prob
    flow = _flow(3, 5).to(DEV)
    x = (cases.randn(77, 2048, 3) * 2).to(DEV)
    a = realnvp_log_prob(flow, x)
    b = realnvp_log_prob_torch(flow, x)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-3)

"""The fused readout-sum + MLP head of Net_1 (npi_gnn_amd/head.py, csrc/head.hip; reference src/classes.py:74-80) against
the same arithmetic written with torch ops in float64 on the CPU, with the SAME dropout mask: output and every gradient."""
import pytest
import torch
import torch.nn.functional as F

from npi_gnn_amd import head as NH

pytestmark = pytest.mark.gpu


def reference(rs, lins, mask, scale, y):
    rs = [r.double().requires_grad_(True) for r in rs]
    ps = [(l.weight.detach().double().requires_grad_(True), l.bias.detach().double().requires_grad_(True)) for l in lins]
    x = rs[0]
    for r in rs[1:]:
        x = x + r
    x = F.relu(F.linear(x, *ps[0]))
    if mask is not None:
        x = x * mask.double() * scale
    x = F.relu(F.linear(x, *ps[1]))
    out = F.log_softmax(F.linear(x, *ps[2]), -1)
    F.nll_loss(out, y).backward()
    return out.detach(), [r.grad for r in rs], [g for p in ps for g in (p[0].grad, p[1].grad)]


@pytest.mark.parametrize("B,dims,n_r,train", [(200, (256, 128, 64, 2), 3, True), (200, (256, 128, 64, 2), 3, False),
                                              (1, (256, 128, 64, 2), 3, True), (37, (64, 256, 128, 5), 2, True),
                                              (13, (1024, 4, 8, 32), 1, True)])
def test_fused_head_matches_the_torch_ops_in_float64(dev, B, dims, n_r, train):
    g = torch.Generator().manual_seed(B + dims[0])
    D0, D1, D2, D3 = dims
    lins = [torch.nn.Linear(D0, D1), torch.nn.Linear(D1, D2), torch.nn.Linear(D2, D3)]
    rs = [torch.randn(B, D0, generator=g) for _ in range(n_r)]
    y = torch.randint(0, D3, (B,), generator=g)
    mask = (torch.rand(B, D1, generator=g) < 0.5).float() if train else None
    out_ref, dr_ref, dp_ref = reference(rs, lins, mask, 2.0, y)
    dl = [l.to(dev) for l in (torch.nn.Linear(D0, D1), torch.nn.Linear(D1, D2), torch.nn.Linear(D2, D3))]
    for a, b in zip(dl, lins):
        a.load_state_dict(b.state_dict())
    rd = [r.to(dev).requires_grad_(True) for r in rs]
    out = NH.mlp_head(rd, *dl, p=0.5, training=train, mask=mask.to(dev) if train else None)
    F.nll_loss(out, y.to(dev)).backward()
    torch.testing.assert_close(out.detach().cpu().double(), out_ref, atol=2e-5, rtol=1e-5)
    for r, ref in zip(rd, dr_ref):
        torch.testing.assert_close(r.grad.cpu().double(), ref, atol=1e-6, rtol=1e-4)
    got = [t for l in dl for t in (l.weight.grad, l.bias.grad)]
    for t, ref in zip(got, dp_ref):
        torch.testing.assert_close(t.cpu().double(), ref, atol=2e-6, rtol=1e-4)


def test_fused_head_draws_a_fresh_bernoulli_mask_and_skips_it_in_evaluation(dev):
    torch.manual_seed(0)
    lins = [torch.nn.Linear(256, 128).to(dev), torch.nn.Linear(128, 64).to(dev), torch.nn.Linear(64, 2).to(dev)]
    rs = [torch.randn(200, 256, device=dev) for _ in range(3)]
    a = NH.mlp_head(rs, *lins, p=0.5, training=True)
    b = NH.mlp_head(rs, *lins, p=0.5, training=True)
    assert not torch.equal(a, b)                                  # two masks
    e1 = NH.mlp_head(rs, *lins, p=0.5, training=False)
    e2 = NH.mlp_head(rs, *lins, p=0.5, training=False)
    assert torch.equal(e1, e2)
    x = rs[0] + rs[1] + rs[2]
    ref = F.log_softmax(lins[2](F.relu(lins[1](F.relu(lins[0](x))))), -1)
    torch.testing.assert_close(e1, ref, atol=2e-5, rtol=1e-5)
    with pytest.raises(ValueError):
        NH.mlp_head(rs + rs, *lins)


@pytest.mark.parametrize("B,train", [(200, True), (200, False), (7, True)])
def test_the_one_output_head_sigmoid_and_bce_match_torch_in_float64(dev, B, train):
    """The reference's one-output variant (src/train_with_twoDataset_modelOnlyOneOutput.py:45-98): lin3 is 64 -> 1, the head
    ends in torch.sigmoid and is trained with binary cross entropy -- the fused head with activation="sigmoid" against the torch
    ops in float64 with the same dropout mask: output and every gradient; and the model class that uses it."""
    g = torch.Generator().manual_seed(B)
    lins = [torch.nn.Linear(256, 128), torch.nn.Linear(128, 64), torch.nn.Linear(64, 1)]
    rs = [torch.randn(B, 256, generator=g) for _ in range(3)]
    y = torch.randint(0, 2, (B,), generator=g)
    mask = (torch.rand(B, 128, generator=g) < 0.5).float() if train else None
    r6 = [r.double().requires_grad_(True) for r in rs]
    ps = [(l.weight.detach().double().requires_grad_(True), l.bias.detach().double().requires_grad_(True)) for l in lins]
    x = r6[0] + r6[1] + r6[2]
    x = F.relu(F.linear(x, *ps[0]))
    if mask is not None:
        x = x * mask.double() * 2.0
    ref = torch.sigmoid(F.linear(F.relu(F.linear(x, *ps[1])), *ps[2]))
    F.binary_cross_entropy(ref, y.double().view(-1, 1)).backward()
    dl = [torch.nn.Linear(256, 128).to(dev), torch.nn.Linear(128, 64).to(dev), torch.nn.Linear(64, 1).to(dev)]
    for a, b in zip(dl, lins):
        a.load_state_dict(b.state_dict())
    rd = [r.to(dev).requires_grad_(True) for r in rs]
    out = NH.mlp_head(rd, *dl, p=0.5, training=train, mask=mask.to(dev) if train else None, activation="sigmoid")
    assert out.shape == (B, 1)
    F.binary_cross_entropy(out, y.to(dev).float().view(-1, 1)).backward()
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), atol=2e-6, rtol=1e-5)
    for r, want in zip(rd, r6):
        torch.testing.assert_close(r.grad.cpu().double(), want.grad, atol=1e-6, rtol=1e-4)
    for t, want in zip([t for l in dl for t in (l.weight.grad, l.bias.grad)], [g_ for p in ps for g_ in (p[0].grad, p[1].grad)]):
        torch.testing.assert_close(t.cpu().double(), want, atol=2e-6, rtol=1e-4)
    with pytest.raises(ValueError):
        NH.mlp_head(rd, *dl, activation="tanh")
    from npi_gnn_amd import net1
    m = net1.Net_1_onlyOneOutput(178, 2)
    assert tuple(m.lin3.weight.shape) == (1, 64) and m.head_activation == "sigmoid"
    assert sorted(m.state_dict()) == sorted(net1.Net_1(178, 2).state_dict())      # the reference's parameter names

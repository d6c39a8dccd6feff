"""GPU parity of the fused optimizer step against torch.optim.AdamW(amsgrad=True, eps=1e-5) / SGD run on
the same device with the same gradients, and of a short training run driven by the reference's scheduler."""
import copy

import pytest
import torch

from tests.parity_util import rel_err

pytestmark = pytest.mark.gpu


def _model():
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from gnnepcsaft_amd.train.models import create_model
    cfg = default_config(2)
    cfg.update(hidden_dim=32, propagation_depth=2, learning_rate=1e-3, weight_decay=1e-2, warmup_steps=2)
    batch = synthetic_batch(32, 2)
    torch.manual_seed(0)
    return cfg, batch, create_model(cfg, calc_deg(batch)).to("cuda:0")


@pytest.mark.parametrize("kind", ["adam", "sgd"])
def test_fused_optimizer_matches_torch(gpu_device, kind):
    from gnnepcsaft_amd.optim import configure_fused_optimizers
    cfg, batch, m1 = _model()
    _, _, m2 = _model()
    m2.load_state_dict(m1.state_dict())
    m1.config["optimizer"] = kind
    m2.config["optimizer"] = kind
    ref = m1.configure_optimizers()
    fus = configure_fused_optimizers(m2)
    o1, s1, o2, s2 = ref["optimizer"], ref["lr_scheduler"]["scheduler"], fus["optimizer"], fus["lr_scheduler"]["scheduler"]
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for step in range(7):
        grads = [torch.randn(p.shape, device="cuda:0", generator=g) * 0.1 for p in m1.parameters()]
        o1.zero_grad()
        o2.zero_grad()
        for p, q, gr in zip(m1.parameters(), m2.parameters(), grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)  # views of the flat gradient buffer
        o1.step()
        o2.step()
        if step % 2 == 1:  # the reference steps the scheduler every 10 epochs; here every other step
            s1.step()
            s2.step()
        assert abs(o1.param_groups[0]["lr"] - o2.param_groups[0]["lr"]) < 1e-12
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert rel_err(q, p) <= 2e-6, n


def test_training_loop_with_fused_adamw_reduces_loss(gpu_device):
    from gnnepcsaft_amd import functional as Fn
    from gnnepcsaft_amd.optim import configure_fused_optimizers
    cfg, batch, model = _model()
    oc = configure_fused_optimizers(model)
    opt = oc["optimizer"]
    b = batch.to("cuda:0")
    losses = []
    try:
        Fn.set_grad_in_place(True)
        for step in range(8):
            opt.zero_grad()
            loss = model.training_step(b, step)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        Fn.set_grad_in_place(False)
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses

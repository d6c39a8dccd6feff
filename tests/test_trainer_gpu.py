"""Trainer shim (reference train.py:85-115): fit() on synthetic molecules through the HIP path with the
fused optimizer and the reference's scheduler cadence; (1) its loss trajectory tracks the CPU oracle trained with
torch.optim.AdamW(amsgrad) + CosineAnnealingWarmRestarts on the same batches; (2) a run interrupted in the middle of an
epoch and resumed from its checkpoint (weights, optimizer moments and step, scheduler, epoch position, shuffle state)
continues like the uninterrupted run; (3) the checkpoint is read back with weights_only=True."""
import copy
import os

import pytest
import torch

from oracle import pyg_restatement as O

pytestmark = pytest.mark.gpu


def _setup(conv="GINE", **kw):
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    cfg = default_config(2)
    cfg.update(conv=conv, hidden_dim=32, propagation_depth=2, warmup_steps=2, learning_rate=2e-3, **kw)
    dataset = synthetic_batch(96, 3 if conv == "GINE" else 2).to_data_list()
    return cfg, dataset, calc_deg(dataset)


@pytest.mark.parametrize("conv,kw,tol", [("GINE", {}, 2e-4), ("PNA", dict(pre_layers=1, post_layers=1), 4e-3)])
def test_fit_tracks_the_oracle_training_trajectory(gpu_device, conv, kw, tol):
    """N optimizer steps of Trainer.fit vs N steps of the oracle model + torch.optim.AdamW(amsgrad=True, eps=1e-5) + the
    same scheduler (models.py:47-75), same initial weights, same batches in the same order: loss per step.  Tolerance:
    the first step is the forward parity (1e-5); later steps compare two fp32 training runs whose parameters drift
    apart at the rate of their gradient differences (GINE: no discrete events; PNA pre1/post1: few; the weight gradients
    are summed with fp32 atomics, so the HIP run itself varies from run to run: 1.6e-3 ... 2.1e-3 seen for PNA)."""
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
    from gnnepcsaft_amd.train.models import create_model
    from gnnepcsaft_amd.train.trainer import DataLoader, Trainer
    cfg, dataset, deg = _setup(conv, **kw)
    torch.manual_seed(0)
    model = create_model(copy.deepcopy(cfg), deg)
    oracle = O.GNNePCSAFT(dict(cfg, deg=deg)).train()
    oracle.load_state_dict(model.model.state_dict(), strict=True)
    steps = 24  # 8 epochs of 3 batches
    tr = Trainer(max_steps=steps, log_every_n_steps=1, enable_checkpointing=False)
    tr.fit(model, DataLoader(dataset, batch_size=32, shuffle=True, seed=1))
    got = [r["train_huber"] for r in tr.logged]
    opt = torch.optim.AdamW(oracle.parameters(), lr=cfg["learning_rate"], weight_decay=cfg["weight_decay"], amsgrad=True,
                            eps=1e-5)
    sched = CosineAnnealingWarmRestarts(opt, cfg["warmup_steps"], T_mult=2, eta_min=1e-6)
    want, epoch = [], 0
    loader = DataLoader(dataset, batch_size=32, shuffle=True, seed=1)
    while len(want) < steps:
        for b in loader:
            opt.zero_grad()
            loss = O.ape_huber_loss(oracle(b.x, b.edge_index, b.edge_attr, b.batch), b.para)
            loss.backward()
            opt.step()
            want.append(float(loss))
            if len(want) == steps:
                break
        epoch += 1
        if epoch % 10 == 0:
            sched.step()
    assert len(got) == steps
    assert abs(got[0] - want[0]) <= 1e-5 * want[0]
    worst = max(abs(g - w) / w for g, w in zip(got, want))
    print(conv, "worst relative loss difference over", steps, "steps:", worst, "first/last loss", want[0], want[-1])
    assert worst <= tol, (worst, got, want)
    assert want[-1] < want[0]  # it trains


def test_resume_mid_epoch_continues_like_the_uninterrupted_run(gpu_device, tmp_path):
    """Run A: 14 steps straight (12 batches would be 4 epochs: the scheduler cadence is set to every 2 epochs here so
    that it fires).  Run B: 8 steps (stops after batch 2 of epoch 2), checkpoint, NEW model + trainer + loader,
    fit(ckpt_path=...) to 14.  State that must come back bit for bit: weights, Adam moments and step count, learning
    rate and scheduler counters, epoch, batch position, shuffle order.  The continued trajectories then agree to the
    run-to-run reproducibility of the step itself (weight-gradient kernels accumulate with fp32 atomics, so two
    identical runs differ in the last bits)."""
    from gnnepcsaft_amd.train.models import create_model
    from gnnepcsaft_amd.train.trainer import DataLoader, Trainer, read_checkpoint
    cfg, dataset, deg = _setup("GINE", dropout=0.1)  # GINE: no std / min / max decisions for the last bits to flip

    def new_model():
        torch.manual_seed(0)
        m = create_model(copy.deepcopy(cfg), deg)
        orig = m.configure_optimizers
        return m, orig

    class EveryTwoEpochs:  # same optimizer / scheduler, cadence 2 instead of 10 so that it fires inside the test
        pass

    import gnnepcsaft_amd.train.trainer as T
    real = T.configure_fused_optimizers

    def cadence2(model, grads=None):
        oc = real(model, grads)
        oc["lr_scheduler"]["frequency"] = 2
        return oc

    T.configure_fused_optimizers = cadence2
    try:
        mA, _ = new_model()
        trA = Trainer(max_steps=14, log_every_n_steps=1, enable_checkpointing=False)
        trA.fit(mA, DataLoader(dataset, batch_size=32, shuffle=True, seed=3))
        mB, _ = new_model()
        trB = Trainer(max_steps=8, log_every_n_steps=1, default_root_dir=str(tmp_path), enable_checkpointing=True)
        trB.fit(mB, DataLoader(dataset, batch_size=32, shuffle=True, seed=3))
        path = os.path.join(str(tmp_path), "last.ckpt")
        ckpt = read_checkpoint(path)  # weights_only=True inside
        assert ckpt["global_step"] == 8 and ckpt["epoch"] == 2 and ckpt["loops"]["batch_in_epoch"] == 2
        assert ckpt["optimizer_states"][0]["step"] == 8 and ckpt["dropout"]["calls"] == 16
        assert torch.equal(ckpt["optimizer_states"][0]["exp_avg"], trB._opt.exp_avg.cpu())
        mC, _ = new_model()
        with torch.no_grad():
            for p in mC.parameters():
                p.add_(1.0)  # the resumed model starts from garbage: everything must come from the file
        trC = Trainer(max_steps=14, log_every_n_steps=1, enable_checkpointing=False)
        loaderC = DataLoader(dataset, batch_size=32, shuffle=True, seed=999)  # wrong seed: the state comes from the file
        trC.fit(mC, loaderC, ckpt_path=path)
    finally:
        T.configure_fused_optimizers = real
    assert trC.global_step == 14 and trC.current_epoch == trA.current_epoch
    assert trC._opt._step == trA._opt._step == 14
    assert trC._sched.state_dict()["last_epoch"] == trA._sched.state_dict()["last_epoch"] >= 1
    assert [r["lr"] for r in trC.logged] == [r["lr"] for r in trA.logged[8:]]
    assert [r["step"] for r in trC.logged] == list(range(9, 15))
    a = [r["train_huber"] for r in trA.logged[8:]]
    c = [r["train_huber"] for r in trC.logged]
    # (two runs of the SAME code differ in the last bits -- fp32 atomics in the weight gradients -- and six optimizer steps
    # amplify that: 2e-5 of the largest parameter has been seen; a broken resume is off by percents)
    assert max(abs(x - y) / x for x, y in zip(a, c)) <= 5e-5, (a, c)
    pa = torch.cat([p.detach().reshape(-1) for p in mA.parameters()])
    pc = torch.cat([p.detach().reshape(-1) for p in mC.parameters()])
    assert float((pa - pc).abs().max()) <= 1e-4 * float(pa.abs().max())
    assert mC.model.dropout.calls == mA.model.dropout.calls == 28

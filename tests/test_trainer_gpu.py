"""Trainer shim: a short fit() on synthetic molecules through the HIP path with the fused optimizer, the
reference's scheduler cadence, logging, and a checkpoint round trip read back with weights_only=True."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fit_logs_checkpoints_and_resumes(gpu_device, tmp_path):
    from gnnepcsaft_amd.data import calc_deg, default_config, synthetic_batch
    from gnnepcsaft_amd.train.models import create_model
    from gnnepcsaft_amd.train.trainer import DataLoader, Trainer
    cfg = default_config(2)
    cfg.update(hidden_dim=32, propagation_depth=2, warmup_steps=2)
    big = synthetic_batch(96, 2)
    dataset = big.to_data_list()
    deg = calc_deg(dataset)
    loader = DataLoader(dataset, batch_size=32, shuffle=True, seed=1)
    assert len(loader) == 3
    torch.manual_seed(0)
    model = create_model(cfg, deg)
    tr = Trainer(max_steps=9, log_every_n_steps=3, default_root_dir=str(tmp_path), enable_checkpointing=True)
    tr.fit(model, loader)
    assert tr.global_step == 9 and tr.current_epoch == 3 and len(tr.logged) == 3
    assert all(k in tr.logged[0] for k in ("train_huber", "train_mape", "lr", "step"))
    assert tr.logged[-1]["train_huber"] < tr.logged[0]["train_huber"] * 1.5  # finite and not diverging
    path = os.path.join(str(tmp_path), "last.ckpt")
    assert os.path.exists(path)
    model2 = create_model(dict(cfg), deg)
    ckpt = Trainer.load_state_dict(model2, path)
    assert ckpt["global_step"] == 9
    for (n, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), n
    tr2 = Trainer(max_steps=12, log_every_n_steps=1, enable_checkpointing=False)
    tr2.fit(model2, loader, ckpt_path=path)
    assert tr2.global_step == 12 and len(tr2.logged) == 3

"""The data-parallel step (demf_amd/engine.py) on CPU with gloo, world_size 2:
one flat-bucket all-reduce per step, replicas stay identical, and the result equals a
single-process step on the mean gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from demf_amd import engine


class Toy(nn.Module):
    """Stands in for DeMFHotPath on CPU: same forward_train / param_groups contract."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.backbone = nn.Linear(5, 7, bias=False)  # a bias before BN has a ~0 gradient; Adam would amplify noise
        self.decoder = nn.Linear(7, 3)     # 'decoder' params train at lr * 0.05
        self.bn = nn.BatchNorm1d(7)

    def forward_train(self, points, img_features, img_metas, gt_bboxes_3d, gt_labels_3d):
        y = self.decoder(torch.relu(self.bn(self.backbone(points))))
        return dict(a=(y - gt_bboxes_3d).pow(2).sum(), b=y.abs().sum() * 0.1)

    def param_groups(self, lr=0.008, weight_decay=0.01):
        dec = [p for n, p in self.named_parameters() if "decoder" in n]
        rest = [p for n, p in self.named_parameters() if "decoder" not in n]
        return [dict(params=rest, lr=lr, weight_decay=weight_decay),
                dict(params=dec, lr=lr * 0.05, weight_decay=weight_decay)]


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return dict(points=torch.randn(6, 5, generator=g), img_features=None, img_metas=None,
                gt_bboxes_3d=torch.randn(6, 3, generator=g), gt_labels_3d=None)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    engine.init_distributed()
    assert dist.get_backend() == "gloo" and dist.get_world_size() == world
    model = Toy()
    if rank == 1:                      # replicas start different: Trainer must broadcast rank 0
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    tr = engine.Trainer(model, max_grad_norm=10.0)
    assert all(p.grad.data_ptr() >= tr.flat.flat.data_ptr() for p in tr.flat.params)
    for _ in range(3):
        tr.step(_batch(rank))
    torch.save({k: v.clone() for k, v in model.state_dict().items() if "running" not in k and "num_b" not in k},
               os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_step_matches_mean_gradient(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for k in a:
        assert torch.equal(a[k], b[k]), f"replicas diverged on {k}"
    # single-process reference: average the two ranks' gradients by hand
    ref = Toy()
    opt = torch.optim.AdamW(ref.param_groups(), lr=0.008, weight_decay=0.01)
    params = [p for g in ref.param_groups() for p in g["params"]]
    for _ in range(3):
        grads = []
        for r in range(world):
            ref.zero_grad()
            sum(ref.forward_train(**_batch(r)).values()).backward()
            grads.append([p.grad.clone() for p in params])
        for i, p in enumerate(params):
            p.grad = (grads[0][i] + grads[1][i]) / world
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
    for k, v in ref.state_dict().items():
        if k in a:
            torch.testing.assert_close(a[k], v, rtol=1e-5, atol=1e-6, msg=k)


def test_flat_grads_views_and_clip():
    m = Toy()
    fg = engine.FlatGrads(m.parameters())
    sum(m.forward_train(**_batch(0)).values()).backward()
    assert fg.flat.abs().sum() > 0
    total = torch.sqrt(sum(p.grad.pow(2).sum() for p in m.parameters()))
    n = fg.clip_(0.5)
    torch.testing.assert_close(n, total)
    torch.testing.assert_close(torch.linalg.vector_norm(fg.flat), torch.tensor(0.5), rtol=1e-4, atol=1e-6)
    fg.zero_()
    assert all((p.grad == 0).all() for p in m.parameters())

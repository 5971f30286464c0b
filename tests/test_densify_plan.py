"""Densify-and-prune selection logic on the CPU (no GPU needed): `igs_amd.densify.plan` -- the gather plan the HIP remap kernel
executes -- against a literal restatement of the reference's mask / cat sequence (igs/models/gaussian_model.py:586-663)."""
import torch

from igs_amd import densify as dn
from igs_amd.scenes import cfg1_scene


def _reference_sequence(t, stats, cfg, gen):
    def cat_all(new):
        for k in t:
            t[k] = torch.cat((t[k], new[k]), dim=0)
    def prune(mask):
        for k in t:
            t[k] = t[k][~mask]
    P = t["xyz"].shape[0]
    grads = stats["accum"].view(P, 1) / stats["denom"].view(P, 1)
    grads[grads.isnan()] = 0.0
    max_num_add = cfg.max_num - P
    sel = torch.norm(grads, dim=-1) >= cfg.grad_threshold
    if cfg.control_max and sel.sum() > max_num_add:
        tv, ti = torch.topk(grads, max_num_add, dim=0)
        grads = torch.zeros_like(grads)
        grads.scatter_(0, ti, tv)
    sel = (torch.norm(grads, dim=-1) >= cfg.grad_threshold) & (torch.exp(t["scaling"]).max(dim=1).values <= cfg.percent_dense * cfg.extent)
    cat_all({k: t[k][sel] for k in t})
    n_init = t["xyz"].shape[0]
    padded = torch.zeros((n_init,))
    padded[:grads.shape[0]] = grads.squeeze()
    sel = (padded >= cfg.grad_threshold) & (torch.exp(t["scaling"]).max(dim=1).values > cfg.percent_dense * cfg.extent)
    N = 2
    stds = torch.exp(t["scaling"])[sel].repeat(N, 1)
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3)), std=stds, generator=gen)
    rots = dn.build_rotation(t["rotation"][sel]).repeat(N, 1, 1)
    cat_all(dict(xyz=torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + t["xyz"][sel].repeat(N, 1),
                 scaling=torch.log(torch.exp(t["scaling"])[sel].repeat(N, 1) / (0.8 * N)), rotation=t["rotation"][sel].repeat(N, 1),
                 opacity=t["opacity"][sel].repeat(N, 1), shs=t["shs"][sel].repeat(N, 1, 1)))
    prune(torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=bool))))
    prune((torch.sigmoid(t["opacity"]) < cfg.min_opacity).squeeze())


def _apply_plan(t, pl):
    src = pl["src"].long()
    out = {k: v[src].clone() for k, v in t.items()}
    child = pl["ovr"].long() >= 0
    out["xyz"][child] = pl["ovr_xyz"][pl["ovr"].long()[child]]
    out["scaling"][child] = pl["ovr_scale"][pl["ovr"].long()[child]]
    return out


def test_plan_equals_reference_sequence_on_cpu():
    raw, _, _ = cfg1_scene(P=3000, size=32)
    for max_num in (100000, 3200):
        g = torch.Generator().manual_seed(1)

        class S:            # DensifyState without the GPU kernel
            grad_accum = torch.rand(3000, generator=g) * 6e-4
            denom = torch.randint(0, 3, (3000,), generator=g).float()
        cfg = dn.DensifyConfig(grad_threshold=0.00015, min_opacity=0.005, max_num=max_num, percent_dense=0.01, extent=5.0)
        t = {k: v.clone() for k, v in raw.items()}
        _reference_sequence(t, dict(accum=S.grad_accum.clone(), denom=S.denom.clone()), cfg, torch.Generator().manual_seed(4))
        pl = dn.plan(raw["xyz"], raw["rotation"], raw["opacity"], raw["scaling"], S, cfg, torch.Generator().manual_seed(4))
        got = _apply_plan(raw, pl)
        assert pl["n_clone"] > 0 and pl["n_split"] > 0
        for k in t:
            assert got[k].shape == t[k].shape, (k, got[k].shape, t[k].shape)
            assert torch.equal(got[k], t[k]), k
        # every row flagged fresh is a clone or a split child (some of them may have been pruned again for low opacity)
        assert int(pl["fresh"].sum()) <= pl["n_clone"] + 2 * pl["n_split"]
        assert int((pl["ovr"] >= 0).sum()) <= 2 * pl["n_split"] and bool((pl["fresh"][pl["ovr"] >= 0] == 1).all())
        if max_num == 3200:
            assert pl["n_clone"] + pl["n_split"] <= 200


def test_no_candidates_means_only_pruning():
    raw, _, _ = cfg1_scene(P=500, size=32)

    class S:
        grad_accum = torch.zeros(500)
        denom = torch.ones(500)
    cfg = dn.DensifyConfig(grad_threshold=1.0, min_opacity=0.2, max_num=100000, extent=5.0)
    pl = dn.plan(raw["xyz"], raw["rotation"], raw["opacity"], raw["scaling"], S, cfg)
    keep = torch.sigmoid(raw["opacity"].view(-1)) >= 0.2
    assert pl["n_clone"] == 0 and pl["n_split"] == 0
    assert torch.equal(pl["src"].long(), torch.nonzero(keep).squeeze(1)) and int(pl["fresh"].sum()) == 0

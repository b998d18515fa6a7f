"""Checkpoints in the reference's file format (reference edm2/utils.py:15-64): `g12_ckpt_ref.pt` was WRITTEN
BY THE REFERENCE's BetterModule.save_to_state_dict (tests/golden/make_golden.py g12); `g12_ckpt.npz` holds an input and
the outputs of the model the reference rebuilt from that file.  CPU: the file's layout and the oracle on it; GPU:
UNet.from_pretrained on the HIP path, and a save -> load round trip through this package's own writer."""
import os
import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(G, "g12_ckpt_ref.pt")


def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _fixture():
    z = np.load(os.path.join(G, "g12_ckpt.npz"), allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(z[k])) if z[k].dtype.kind in "fi" else z[k] for k in z.files}


def test_reference_written_checkpoint_layout_and_oracle():
    from oracle import oniris_oracle as O
    ck = torch.load(CKPT, weights_only=True)
    assert set(ck) == {"state_dict", "kwargs"}
    z = _fixture()
    assert sorted(ck["state_dict"]) == list(z["keys"])
    assert sum(v.numel() for k, v in ck["state_dict"].items() if "freqs" not in k and "phases" not in k and "rope" not in k) \
        == int(z["n_params"])
    cfg = {k: v for k, v in ck["kwargs"].items() if k not in ("label_balance", "concat_balance") and v is not None}
    p = ck["state_dict"]
    y, cache = O.unet_forward(p, cfg, z["x"], z["c_noise"], z["labels"], update_cache=True, training=False)
    y1, _ = O.unet_forward(p, cfg, z["x1"], z["c_noise"][:, :1], z["labels"][:, :1], cache=cache, training=False)
    y2, _ = O.unet_forward(p, cfg, z["x"], z["c_noise"], z["labels"], just_2d=True, training=True)
    e = (rel(y, z["y_eval"]), rel(y1, z["y_next"]), rel(y2, z["y_2d"]))
    print("oracle on the reference-written checkpoint: eval / cached next frame / 2-D training", e)
    assert max(e) < 2e-5


def test_module_tree_matches_the_checkpoint_without_a_gpu():
    """Constructor + load_state_dict(strict) from the reference's file: key names, shapes and kwargs (no kernel runs)."""
    from edm2.networks_edm2 import UNet
    net = UNet.from_pretrained(CKPT)
    ck = torch.load(CKPT, weights_only=True)
    assert net.kwargs == ck["kwargs"]
    sd = net.state_dict()
    assert list(sd) == list(ck["state_dict"])                       # same keys in the same order
    assert all(torch.equal(sd[k], ck["state_dict"][k]) for k in sd)


@pytest.mark.gpu
def test_from_pretrained_on_the_hip_path_and_save_load_round_trip(tmp_path):
    from edm2.networks_edm2 import UNet
    z = _fixture()
    net = UNet.from_pretrained(CKPT).to("cuda").eval()
    x, cn, lab = z["x"].cuda(), z["c_noise"].cuda(), z["labels"].cuda()
    with torch.no_grad():
        y, cache = net(x, cn, lab, update_cache=True)
        y1, _ = net(z["x1"].cuda(), cn[:, :1], lab[:, :1], cache=cache)
    # written back in the reference's format by this package, re-read by both loaders
    out = str(tmp_path / "again.pt")
    net.save_to_state_dict(out)
    mine, ref = torch.load(out, weights_only=True), torch.load(CKPT, weights_only=True)
    assert set(mine) == {"state_dict", "kwargs"} and mine["kwargs"] == ref["kwargs"]
    assert list(mine["state_dict"]) == list(ref["state_dict"])
    assert all(torch.equal(mine["state_dict"][k].cpu(), ref["state_dict"][k]) for k in ref["state_dict"])   # eval: weights untouched
    net2 = UNet.from_pretrained(out).to("cuda").eval()
    with torch.no_grad():
        y_b, _ = net2(x, cn, lab)
    assert torch.equal(y_b, y)                                      # bit-identical after the round trip
    net.train()
    with torch.no_grad():
        y2, _ = net(x, cn, lab, just_2d=True)
    e = (rel(y, z["y_eval"]), rel(y1, z["y_next"]), rel(y2, z["y_2d"]))
    print("HIP path on the reference-written checkpoint: eval / cached next frame / 2-D training", e)
    assert max(e) < 2e-2                                            # bf16 kernels vs the reference's fp32 (G8's bound)

"""CPU checks of the drop-in boundary: liboniris_hip.so loads, exports every symbol include/oniris.h declares, the
ctypes struct mirrors match the C structs, the host-side mask builder is bit-exact against the golden tables, and
the edm2 namespace exposes the reference's import surface.  No kernel is launched."""
import os
import re
import numpy as np
import pytest
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    from autoregressive_diffusion_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "oniris.h")).read()
    declared = set(re.findall(r"\b(oniris_[a-z0-9_]+)\s*\(", hdr)) - {"oniris_stream_t"}
    assert declared == set(_lib.EXPORTED), (declared ^ set(_lib.EXPORTED))
    for name in declared:
        assert hasattr(_lib.lib, name)
    assert _lib.lib.oniris_abi_version() == 14


def test_mask_tables_against_golden():
    from autoregressive_diffusion_amd import ops
    z = np.load(os.path.join(ROOT, "tests", "golden", "g1_masks.npz"))
    for (T, P) in [(64, 64), (32, 16), (64, 16), (8, 64), (4, 256), (3, 128)]:
        num, idx, blk = ops.train_mask_table(T, P)
        assert num.dtype == np.int32 and idx.dtype == np.int32
        assert np.array_equal(num, z[f"train_{T}_{P}_num"][0, 0]) and np.array_equal(idx, z[f"train_{T}_{P}_idx"][0, 0])
        assert blk == int(z[f"train_{T}_{P}_blk"])
        qn, qi = ops.mask_transpose(num, idx)
        for c in range(idx.shape[1]):          # transposed table lists exactly the rows that list column c
            rows = [r for r in range(idx.shape[0]) if c in idx[r, :num[r]]]
            assert list(qi[c, :qn[c]]) == rows
    assert ops.train_mask_table(3, 64) is None
    for (t, P) in [(4, 64), (8, 16), (6, 256)]:
        num, idx, _ = ops.infer_mask_table(t, P)
        assert np.array_equal(num, z[f"infer_{t}_{P}_num"][0, 0]) and np.array_equal(idx, z[f"infer_{t}_{P}_idx"][0, 0])
    assert ops.infer_mask_table(5, 64) is None and ops.infer_mask_table(1, 64) is None


def test_edm2_surface_and_state_dict_keys():
    import paramgen
    from edm2.networks_edm2 import UNet, Precond, Block  # noqa: F401
    from edm2.attention import VideoAttention, FrameAttention  # noqa: F401
    from edm2.conv import MPConv, MPCausal3DGatedConv, Gating, NormalizedWeight  # noqa: F401
    from edm2.attention.attention_masking import make_train_mask, make_infer_mask, TrainingMask  # noqa: F401
    from edm2.loss import EDM2Loss, learning_rate_schedule  # noqa: F401
    from edm2.sampler import edm_sampler_with_mse  # noqa: F401
    cfg = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=32, channel_mult=[1, 2, 4, 8],
               num_blocks=2, video_attn_resolutions=[8], frame_attn_resolutions=[16])
    unet = UNet(**cfg)
    assert unet.n_params == 46248671          # gym_train.py:37-47 net; SURVEY: 46.2 M
    net = Precond(unet, sigma_data=1.0)
    want = paramgen.unet_param_shapes(cfg)    # verified against the reference's own state_dict (strict load) in make_golden
    sd = net.state_dict()
    assert {("unet." + k) for k in want} | {"noise_weight.fourier_approximator.coefficients"} == set(sd)
    for k, s in want.items():
        assert tuple(sd["unet." + k].shape) == tuple(s), k
    assert unet.kwargs["model_channels"] == 32 and set(unet.kwargs) >= {"img_resolution", "channel_mult", "concat_balance"} and "block_kwargs" not in unet.kwargs
    bm = make_train_mask(2, 4, 8, 64)
    assert bm.kv_num_blocks.dtype == torch.int32 and tuple(bm.kv_indices.shape) == (2, 4, 8, 8)
    import copy
    copy.deepcopy(net)                        # phema.py:95 deep-copies the net


def test_product_path_has_no_cpu_fallback():
    from edm2.conv import MPConv
    import pytest
    m = MPConv(16, 16, [3, 3])
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 16, 8, 8))           # CPU tensors: loud failure, never a silent PyTorch path


def test_attn_schedule_is_a_balanced_partition():
    """oniris_attn_schedule (host): every (pair, block) item exactly once, pairs stay inside their workgroup group,
    and at the C2 shape (8 pairs x 64 query blocks, 256 workgroups) every workgroup gets the same load."""
    import ctypes
    from autoregressive_diffusion_amd import ops, _lib
    num, idx, blk = ops.train_mask_table(64, 64)
    w = np.repeat(num * (blk // 128) + 1, blk // 128).astype(np.int32)
    for pairs, n_wg in ((8, 256), (16, 256), (4, 256), (12, 256), (6, 64), (1, 8)):
        need = _lib.lib.oniris_attn_schedule(pairs, len(w), w.ctypes.data_as(ctypes.c_void_p), n_wg, None, 0)
        assert need >= 1
        tab = np.full((n_wg, need), -7, np.int32)
        assert _lib.lib.oniris_attn_schedule(pairs, len(w), w.ctypes.data_as(ctypes.c_void_p), n_wg,
                                             tab.ctypes.data_as(ctypes.c_void_p), need) == need
        items = sorted(int(e) for e in tab.ravel() if e >= 0)
        assert items == sorted((p << 16) | b for p in range(pairs) for b in range(len(w)))
        assert all((row[np.argmax(row < 0):] < 0).all() for row in tab if (row < 0).any()), "items first, then -1"
        ng = next(g for g in (8, 4, 2, 1) if pairs % g == 0 and n_wg % g == 0)
        for wg, row in enumerate(tab):
            assert all((int(e) >> 16) % ng == wg % ng for e in row if e >= 0)
        loads = np.array([sum(int(w[e & 0xffff]) for e in row if e >= 0) for row in tab])
        if pairs in (8, 16):
            assert loads.min() == loads.max()
    # too few slots -> error code, message through oniris_last_error
    tab = np.zeros((256, 1), np.int32)
    assert _lib.lib.oniris_attn_schedule(8, len(w), w.ctypes.data_as(ctypes.c_void_p), 256,
                                         tab.ctypes.data_as(ctypes.c_void_p), 1) < 0


def test_attention_modules_head_dimensions():
    """64-channel heads (every shipped configuration, networks_edm2.py:28,39) and -- through the padded path -- every multiple of
    8 below 64 (the reference's unit tests build 4 heads of 16 channels, consistency_test.py:39,61); wider heads up to 256 channels go
    to the generic fp32 attention kernel; anything else (a width that does not divide the channels, is not a multiple of 8, or
    exceeds 256) must be refused instead of computing garbage."""
    import pytest
    import autoregressive_diffusion_amd  # noqa: F401
    from edm2.attention import VideoAttention, FrameAttention
    for cls in (VideoAttention, FrameAttention):
        cls(channels=64, num_heads=4)          # 16-channel heads
        cls(channels=128, num_heads=2)
        cls(channels=64, num_heads=0)
        cls(channels=96, num_heads=2)          # 48-channel heads
        cls(channels=48, num_heads=2)          # 24
        cls(channels=256, num_heads=2)         # 128-channel heads: the generic fp32 attention kernel (round 6)
        for channels, heads in ((64, 3), (24, 2), (1024, 2)):     # not a divisor, 12, 512
            with pytest.raises(NotImplementedError):
                cls(channels=channels, num_heads=heads)


def test_attention_flop_count_is_the_unmasked_pair_count():
    """bench.py's attention roofline prices the kernel on ALGORITHMIC FLOPs = unmasked token pairs of `table AND mask_mod`
    (T(T+1) over-counted by 0.8 % at P = 64 and 12 % at P = 16)."""
    from autoregressive_diffusion_amd import ops
    from oracle import oniris_oracle as O
    for T, P, want in [(64, 64, 4128), (32, 16, 944), (64, 16, 3936), (8, 64, 68), (4, 256, 20), (3, 128, 12), (16, 32, 248)]:
        assert ops.train_frame_pairs(T, P) == want == int(O.train_allowed_tokens(T, P).sum()) // (P * P)
    assert ops._attn_flops("video", 2, 64, 4, 8192, 64) == 4.0 * 64 * 4 * 2 * 4128 * 64 * 64


def test_optimizer_and_prelude_have_no_cpu_arithmetic_in_the_product():
    """the plain-torch AdamW for CPU tensors and the torch formulation of the conditioning prelude
    used to ship inside the package; they live under tests/ now (cpu_reference_optimizer.py, torch_prelude.py) and the
    product refuses without them.  Checked in a fresh interpreter (this process may have them installed)."""
    import subprocess, sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW\n"
        "from autoregressive_diffusion_amd import ops\n"
        "net = torch.nn.Linear(3, 2); flat = FlatParams(net); opt = FlatAdamW(flat)\n"
        "net(torch.randn(4, 3)).sum().backward()\n"
        "try:\n    opt.step(); print('STEPPED')\nexcept RuntimeError as e:\n    print('REFUSED', 'no CPU fallback' in str(e))\n"
        "try:\n    ops._prelude_ref('batched_gates'); print('HAS_REF')\nexcept RuntimeError as e:\n    print('NOREF')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED True" in out.stdout and "NOREF" in out.stdout, out.stdout + out.stderr


def test_edm2_shim_leaves_the_references_other_modules_importable(tmp_path):
    """gym_train.py:18-23 imports edm2.plotting / edm2.vae / edm2.gym_dataloader / edm2.phema next to the accelerated modules.
    With this repository FIRST on the path its `edm2` package must not hide the rest of the reference's `edm2` directory
    (a namespace package): checked with a stand-in directory, in a fresh interpreter."""
    import subprocess, sys
    ref = tmp_path / "refcheckout" / "edm2"
    (ref / "vae").mkdir(parents=True)
    (ref / "plotting.py").write_text("from edm2.sampler import edm_sampler_with_mse\nMARK = 'reference plotting'\n")
    (ref / "vae" / "__init__.py").write_text("VAE = 'reference vae'\n")
    (ref / "conv.py").write_text("raise RuntimeError('the reference conv.py must not be imported')\n")
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import edm2, edm2.plotting\n"
            "from edm2.vae import VAE\n"
            "from edm2.conv import MPConv\n"
            "from edm2.networks_edm2 import UNet\n"
            "print(edm2.plotting.MARK, '|', VAE, '|', MPConv.__module__, '|', edm2.plotting.edm_sampler_with_mse.__module__)\n"
            % (ROOT, str(tmp_path / "refcheckout")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "reference plotting | reference vae | autoregressive_diffusion_amd.edm2.conv | autoregressive_diffusion_amd.edm2.sampler" in out.stdout, out.stdout + out.stderr


def test_edm2_utils_public_names():
    """The reference's public `edm2.utils` names (utils.py:13-235) exist with the same meaning: the reference's VAE and
    training scripts import bmult / GaussianLoss from here (vae/vae.py:13, cs_vae_train.py:19).  Compared with the formulas."""
    import autoregressive_diffusion_amd  # noqa: F401
    from edm2 import utils as U
    for name in ("BetterModule", "normalize", "resample", "mp_silu", "mp_sum", "mp_cat", "MPFourier", "bmult", "nan_hook",
                 "nan_inspector", "GaussianLoss", "compare_caches"):
        assert hasattr(U, name), name
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 4, 6, 6, generator=g)
    assert torch.allclose(U.bmult(x, torch.tensor(2.0)), 2 * x)
    t1, t2 = torch.rand(3, generator=g), torch.rand(3, 4, generator=g)
    assert torch.allclose(U.bmult(x, t1), x * t1[:, None, None, None]) and torch.allclose(U.bmult(x, t2), x * t2[:, :, None, None])
    y = torch.randn(3, 4, 6, 6, generator=g)
    assert torch.allclose(U.mp_sum(x, y, 0.3), (0.7 * x + 0.3 * y) / np.sqrt(0.7 ** 2 + 0.3 ** 2), atol=1e-6)
    assert torch.allclose(U.mp_sum(x, y, t1), (x + (y - x) * t1[:, None, None, None]) / torch.sqrt((1 - t1) ** 2 + t1 ** 2)[:, None, None, None], atol=1e-6)
    b = torch.randn(3, 2, 6, 6, generator=g)
    c = U.mp_cat(x, b, dim=1, t=0.5)
    assert c.shape == (3, 6, 6, 6) and torch.allclose(c[:, :4], x * np.sqrt(6 / 0.5) / np.sqrt(4) * 0.5, atol=1e-6)
    assert torch.allclose(U.resample(x, mode="down"), x.reshape(3, 4, 3, 2, 3, 2).mean(dim=(3, 5)), atol=1e-6)
    up = U.resample(x, mode="up")
    assert up.shape == (3, 4, 12, 12) and torch.equal(up[:, :, ::2, ::2], x) and torch.equal(up[:, :, 1::2, 1::2], x)
    assert U.resample(x, mode="keep") is x
    m, lv = torch.randn(5, 7, generator=g), torch.randn(5, 7, generator=g)
    tgt = torch.randn(5, 7, generator=g)
    assert torch.allclose(U.GaussianLoss(m, lv, tgt), ((lv + (m - tgt) ** 2 * torch.exp(-lv)) * 0.5 + 0.918).mean())
    c1 = {"a": {"x": torch.ones(2), "n": 3}, "b": [torch.zeros(1), 0.5]}
    c2 = {"a": {"x": torch.ones(2), "n": 3}, "b": [torch.zeros(1), 0.5]}
    assert U.compare_caches(c1, c2, verbose=False)
    c2["a"]["x"] = torch.ones(2) * 1.1
    assert not U.compare_caches(c1, c2, verbose=False)
    lin = torch.nn.Sequential(torch.nn.Linear(2, 2))
    with U.nan_inspector(lin):
        lin(torch.zeros(1, 2))
        try:
            lin(torch.full((1, 2), float("nan")))
            raise AssertionError("nan_inspector did not fire")
        except Exception as e:
            assert "NaN detected" in str(e)
    assert len(lin[0]._forward_hooks) == 0


def test_better_module_s3_checkpoints(tmp_path, monkeypatch):
    """save_to_state_dict / from_pretrained accept s3://bucket/key like the reference (utils.py:15-58, through boto3;
    generation_code.py:34 loads the UNet that way).  Checked against a stand-in boto3 whose "bucket" is a directory: upload,
    cached download, reuse of the cached file, and the error without boto3."""
    import sys, types, shutil
    import pytest
    import autoregressive_diffusion_amd  # noqa: F401
    from edm2.utils import BetterModule
    store, calls = tmp_path / "bucket", []
    store.mkdir()

    class Client:
        def upload_file(self, local, bucket, key):
            calls.append(("up", bucket, key)); dst = store / bucket / key; dst.parent.mkdir(parents=True, exist_ok=True); shutil.copy(local, dst)

        def download_file(self, bucket, key, local):
            calls.append(("down", bucket, key)); shutil.copy(store / bucket / key, local)
    monkeypatch.setitem(sys.modules, "boto3", types.SimpleNamespace(client=lambda name: Client()))
    monkeypatch.setattr(BetterModule, "S3_CACHE_DIR", str(tmp_path / "cache") + "/")

    class Tiny(BetterModule):
        def __init__(self, width=3):
            super().__init__()
            self.kwargs = dict(width=width)
            self.lin = torch.nn.Linear(width, width)
    m = Tiny(5)
    m.save_to_state_dict("s3://models/saved/tiny.pt")
    assert calls == [("up", "models", "saved/tiny.pt")] and (store / "models" / "saved" / "tiny.pt").exists()
    back = Tiny.from_pretrained("s3://models/saved/tiny.pt")
    again = Tiny.from_pretrained("s3://models/saved/tiny.pt")                     # second load: the cached file
    assert calls[1:] == [("down", "models", "saved/tiny.pt")]
    assert back.kwargs == dict(width=5) and torch.equal(back.lin.weight, m.lin.weight) and torch.equal(again.lin.bias, m.lin.bias)
    with pytest.raises(ValueError):
        Tiny.from_pretrained("s3://nokey")
    monkeypatch.setitem(sys.modules, "boto3", None)                               # `import boto3` now fails
    with pytest.raises(ImportError, match="boto3"):
        m.save_to_state_dict("s3://models/other.pt")


def test_checkpoint_kwargs_with_numpy_values_load_without_the_full_pickle_machinery(tmp_path, monkeypatch):
    """the reference's VAE is a BetterModule whose `kwargs` are whatever its constructor was given -- numpy
    statistics included -- and gym_train.py:33 / cs_train.py:32 load such files first thing (reference: weights_only=False).
    The restricted loader must take them (numpy scalars / arrays are data), refuse anything else with a message that names the
    opt-in, and the opt-in must load like the reference."""
    import pickle
    import numpy as np
    import pytest
    import autoregressive_diffusion_amd  # noqa: F401
    from edm2.utils import BetterModule

    class Stat(BetterModule):
        def __init__(self, width=3, mean=0.0, std=1.0, table=None):
            super().__init__()
            self.kwargs = dict(width=width, mean=mean, std=std, table=table)
            self.lin = torch.nn.Linear(int(width), int(width))
    m = Stat(np.int64(4), np.float64(0.25), np.float32(1.5), np.arange(3, dtype=np.float32))
    path = str(tmp_path / "stat.pt")
    m.save_to_state_dict(path)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, weights_only=True)                       # what round 5 did: a hard failure before training
    back = Stat.from_pretrained(path)
    assert back.kwargs["mean"] == 0.25 and isinstance(back.kwargs["mean"], np.float64) and back.kwargs["width"] == 4
    assert np.array_equal(back.kwargs["table"], np.arange(3, dtype=np.float32)) and torch.equal(back.lin.weight, m.lin.weight)

    # an arbitrary class in kwargs is not data
    torch.save({"state_dict": m.state_dict(), "kwargs": dict(width=4, mean=_NotData())}, str(tmp_path / "odd.pt"))
    with pytest.raises(pickle.UnpicklingError, match="ONIRIS_TRUST_CHECKPOINT=1"):
        Stat.from_pretrained(str(tmp_path / "odd.pt"))
    monkeypatch.setenv("ONIRIS_TRUST_CHECKPOINT", "1")             # the reference's own load (utils.py:59)
    assert isinstance(Stat.from_pretrained(str(tmp_path / "odd.pt")).kwargs["mean"], _NotData)


class _NotData:
    pass


def test_normalized_weight_forward_matches_the_reference_formula():
    """NormalizedWeight.forward(gain) (conv.py:14-21) as a tensor-returning helper: forced normalisation of the stored parameter in
    training mode, normalize(w) * gain / sqrt(fan_in) -- against the oracle (itself pinned by fixture G2)."""
    import autoregressive_diffusion_amd  # noqa: F401
    from edm2.conv import NormalizedWeight
    from oracle import oniris_oracle as O
    torch.manual_seed(3)
    m = NormalizedWeight(6, 4, (2, 3, 3))
    w0 = m.weight.detach().clone()
    e, wn = O.weight_effective(w0.clone(), 0.8, training=True)
    assert torch.allclose(m(gain=0.8), e, atol=1e-6) and torch.allclose(m.weight.detach(), wn, atol=1e-6)
    m.eval()
    w1 = m.weight.detach().clone()
    assert torch.allclose(m(), O.weight_effective(w1, 1.0, training=False)[0], atol=1e-6) and torch.equal(m.weight.detach(), w1)


def test_precond_fp32_switch_enters_the_fp32_path():
    """`force_fp32=True` / `use_fp16=False` select fp32 arithmetic in the reference (networks_edm2.py:285,294).  Round 5 accepted
    the switch and ignored it (with a warning); now it routes the whole net through the fp32 kernels (autoregressive_diffusion_amd/
    fp32.py, csrc/fp32.hip).  Without a GPU that is visible as: no warning, and the fp32 contraction refusing host tensors (the
    bf16 path refuses them too -- neither has a CPU form).  The numerics are tests/test_fp32_gpu.py."""
    import warnings
    from edm2.networks_edm2 import UNet, Precond
    from autoregressive_diffusion_amd import fp32
    net = Precond(UNet(img_resolution=16, img_channels=4, label_dim=0, model_channels=16, channel_mult=[1, 2], num_blocks=1,
                       video_attn_resolutions=[], frame_attn_resolutions=[]), use_fp16=True, sigma_data=1.0).eval()
    x, sigma = torch.randn(1, 2, 4, 16, 16), torch.ones(1, 2)
    assert not fp32.active()
    for kw, mod in ((dict(force_fp32=True), net), (dict(), Precond(net.unet, use_fp16=False, sigma_data=1.0).eval())):
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            with pytest.raises(RuntimeError, match="fp32 path runs on HIP kernels"):
                mod(x, sigma, **kw)
        assert not fp32.active()                                   # (the switch is scoped to the call, also when it raises)


def test_learning_rate_schedule_values():
    """edm2/loss.py:63-69 of the reference: ref_lr / sqrt(max(step / ref_step, 1)) * min(step / rampup, 1)."""
    from edm2.loss import learning_rate_schedule
    import math
    assert learning_rate_schedule(0) == 0.0
    assert learning_rate_schedule(500) == 1e-2 * 0.5
    assert learning_rate_schedule(1000) == 1e-2
    assert learning_rate_schedule(7e4) == 1e-2
    assert abs(learning_rate_schedule(28e4) - 0.5e-2) < 1e-18
    assert learning_rate_schedule(10, 3e-3, 0, 0) == 3e-3
    assert learning_rate_schedule(16, 1e-2, 4, 4) == 1e-2 / math.sqrt(4.0)


def test_gradslot_scaled_alias_semantics():
    """ops.GradSlot (round 5): a consumer parks (g, scale) -- the residual gradient of an mp_sum epilogue is ta * g, a scaled copy
    nobody writes; take_scaled() hands both to a taker whose kernel applies the factor, take() materialises, a second put adds."""
    from autoregressive_diffusion_amd import ops
    ops.GradSlot.live = []
    g, h = torch.arange(6.0).reshape(2, 3), torch.ones(2, 3)
    s = ops.GradSlot()
    s.put(g, 0.5)
    t, sc = s.take_scaled()
    assert t is g and sc == 0.5 and s.g is None and s.scale == 1.0
    s.put(g, 0.5)
    assert torch.equal(s.take(), g * 0.5) and s.g is None
    s.put(g, 0.5); s.put(h)
    t, sc = s.take_scaled()
    assert sc == 1.0 and torch.equal(t, g * 0.5 + h)
    s.put(h)
    assert s.take() is h
    s.put(g, 2.0)
    with pytest.raises(RuntimeError, match="parked"):
        ops.GradSlot.check_all_taken()                        # a parked gradient nobody took = a partial backward: loud
    assert ops.GradSlot.live == []


def test_flatparams_only_subset_and_deepcopy_strips_runtime_state():
    """parallel.FlatParams(only=...) re-homes just the given parameters (torch DDP around the UNet: the kernel-owned weights);
    copy.deepcopy of this package's modules leaves the run-time state (weight bank handles, cached plans, inner DDP engine) behind."""
    import copy
    from autoregressive_diffusion_amd.parallel import FlatParams
    from edm2.networks_edm2 import UNet
    from edm2.conv import NormalizedWeight
    unet = UNet(img_resolution=16, img_channels=4, label_dim=4, model_channels=8, channel_mult=[1, 2], num_blocks=1)
    owned = [m.weight for m in unet.modules() if isinstance(m, NormalizedWeight)]
    before = {n: p.detach().clone() for n, p in unet.named_parameters()}
    flat = FlatParams(unet, only=owned)
    assert len(flat.params) == len(owned) and {id(p) for p in flat.params} == {id(p) for p in owned}
    base = flat.flat.data_ptr()
    for n, p in unet.named_parameters():
        assert torch.equal(p.detach(), before[n])
        inside = base <= p.data_ptr() < base + 4 * flat.numel
        assert inside == (id(p) in {id(q) for q in owned}), n
    assert flat.check()
    # run-time attachments disappear in a copy; the marker of a torch-DDP-wrapped original survives as "lost"
    unet.__dict__["_oniris_bank"] = object()
    unet.__dict__["_oniris_inner_ddp"] = object()
    next(m for m in unet.modules() if isinstance(m, NormalizedWeight)).pw = object()
    cp = copy.deepcopy(unet)
    assert "_oniris_bank" not in cp.__dict__ and cp.__dict__["_oniris_inner_ddp"] == "lost"
    assert all(m.pw is None for m in cp.modules() if isinstance(m, NormalizedWeight))
    assert cp._ddp_inner() is None                                   # (no process group: evaluation / single-process use is fine)
    for (n, p), (m, q) in zip(unet.named_parameters(), cp.named_parameters()):
        assert n == m and torch.equal(p, q) and p.data_ptr() != q.data_ptr()


def test_dkv_item_keys_follow_the_launch_size():
    """ops._dkv_item_keys: 128-key items only when a workgroup's average load is well above the heaviest item (C2: B = 8, not B = 2)."""
    from autoregressive_diffusion_amd import ops
    ops._cu_count["cpu"] = 256
    try:
        assert ops._dkv_item_keys("video", 64, 64, 2 * 4, "cpu") == 64
        assert ops._dkv_item_keys("video", 64, 64, 8 * 4, "cpu") == 128
        assert ops._dkv_item_keys("video", 32, 16, 2 * 8, "cpu") == 64           # Counter-Strike T = 32: 4 table blocks per pair
    finally:
        del ops._cu_count["cpu"]

"""`edm2` compatibility namespace: re-exports autoregressive_diffusion_amd.edm2 (the MI355X-native implementation
of the reference's edm2 package surface used by gym_train.py, cs_train.py, generation_code.py and edm2/sampler.py)."""
import importlib
import sys

_impl = "autoregressive_diffusion_amd.edm2"
for _name in ("utils", "conv", "attention", "attention.attention_modules", "attention.attention_masking",
              "attention.RoPe", "loss_weight", "loss", "sampler", "networks_edm2"):
    _m = importlib.import_module(f"{_impl}.{_name}")
    sys.modules[f"edm2.{_name}"] = _m
    if "." not in _name:
        globals()[_name] = _m

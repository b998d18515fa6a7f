"""`edm2` compatibility namespace: re-exports autoregressive_diffusion_amd.edm2 (the MI355X-native implementation
of the reference's edm2 package surface used by gym_train.py, cs_train.py, generation_code.py and edm2/sampler.py)."""
import importlib
import os
import sys

_impl = "autoregressive_diffusion_amd.edm2"
for _name in ("utils", "conv", "attention", "attention.attention_modules", "attention.attention_masking",
              "attention.RoPe", "loss_weight", "loss", "sampler", "networks_edm2"):
    _m = importlib.import_module(f"{_impl}.{_name}")
    sys.modules[f"edm2.{_name}"] = _m
    if "." not in _name:
        globals()[_name] = _m

# The reference's scripts also import modules of ITS `edm2` directory that are outside the accelerated path and are not
# re-implemented here (gym_train.py:18-23: edm2.plotting, edm2.vae, edm2.gym_dataloader, edm2.phema; cs_train.py:18-24:
# edm2.cs_dataloading, edm2.vae.stability).  The reference's `edm2` is a namespace package (no __init__.py); this regular
# package would hide it, so every other `edm2` directory on sys.path is appended to this package's search path: the modules
# registered above keep winning (sys.modules is consulted first), everything else resolves to the reference's own file.
_here = os.path.abspath(os.path.dirname(__file__))
for _p in list(sys.path):
    _d = os.path.join(_p or os.getcwd(), "edm2")
    if os.path.isdir(_d) and os.path.abspath(_d) != _here and _d not in __path__:
        __path__.append(_d)

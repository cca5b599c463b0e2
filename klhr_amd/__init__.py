"""klhr_amd — MI355X-native many-chain KL Hit-and-Run sampler (drop-in for the hot path of roualdes/klhr).

Python host classes keep the reference's module and class names; all compute goes through the C ABI of
klhr_amd/lib/libklhr_hip.so (include/klhr.h).  Importing the package never needs a GPU; using it does."""
from ._lib import KlhrError  # noqa: F401

__all__ = ["BSModel", "KLHRSINH", "KLHR", "SUBKLHRSINH", "Slice", "MH", "MCMCBase", "OnlineMoments", "OnlinePCA", "WindowedAdaptation", "Smoother", "KlhrError"]


def __getattr__(name):
    import importlib
    table = {"BSModel": "bsmodel", "KLHRSINH": "klhr_sinh", "KLHR": "klhr", "SUBKLHRSINH": "sub_klhr_sinh", "Slice": "slice", "MH": "mh", "MCMCBase": "mcmc", "OnlineMoments": "onlinemoments",
             "OnlinePCA": "onlinepca", "WindowedAdaptation": "windowedadaptation", "Smoother": "smoother"}
    if name in table:
        return getattr(importlib.import_module(f".{table[name]}", __name__), name)
    raise AttributeError(name)

"""TEST INFRASTRUCTURE ONLY -- harness that imports the *unmodified* reference
(/root/reference, ZiyuanMa/MAPF_RL) inside this container so that golden vectors can be
captured from it and the C restatement (oracle/mapf_oracle.c) can be pinned against it.

Nothing here is shipped or measured, and nothing here can run on the GPU box
(/root/reference does not exist there).  Only tests/golden/make_goldens.py and CPU-side
tests marked `needs_reference` call into this module.

Shims (all harness-side, the reference stays untouched on disk):
  1. MPLBACKEND=Agg, no bytecode writes (reference dir is read-only).
  2. numpy.int = int             (removed in numpy >= 1.24; environment.py:12,26,102 ...).
  3. environment.np is rebound to a stand-in whose where() returns arrays with the
     pre-numpy-2.2 truth value for empty arrays (environment.py:343 `if target_agent_id:`).
  4. a stub `ray` module so worker.py imports and its classes are plain Python classes.
"""
import os
import sys
import types

REFERENCE_DIR = os.environ.get("MAPF_REFERENCE_DIR", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "environment.py"))


_loaded = {}


def _install_shims():
    import numpy as np

    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    if not hasattr(np, "int"):
        np.int = int  # noqa: NPY001 - shim for the 2020-era reference
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    if "ray" not in sys.modules:
        ray = types.ModuleType("ray")

        def remote(*args, **kwargs):
            if len(args) == 1 and not kwargs and (isinstance(args[0], type) or callable(args[0])):
                return args[0]
            return lambda cls: cls

        ray.remote = remote
        ray.put = lambda x: x
        ray.get = lambda x: x
        ray.init = lambda *a, **k: None
        sys.modules["ray"] = ray


class _OldTruthArray:
    pass


def _make_np_standin():
    import numpy as np

    class OldTruth(np.ndarray):
        """ndarray whose bool() follows numpy < 2.2: empty -> False."""

        def __bool__(self):
            if self.size == 0:
                return False
            return bool(np.asarray(self).item()) if self.size == 1 else np.ndarray.__bool__(self)

    ns = types.SimpleNamespace()
    for name in dir(np):
        try:
            setattr(ns, name, getattr(np, name))
        except Exception:
            pass

    def where(*args, **kwargs):
        out = np.where(*args, **kwargs)
        if isinstance(out, tuple):
            return tuple(o.view(OldTruth) for o in out)
        return out

    ns.where = where
    return ns


def load_reference():
    """Returns a namespace with the reference modules (environment, search, model, buffer, worker, config)."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_DIR)
    _install_shims()
    # the repo root has its own config.py / environment.py entry points: make sure the
    # *reference* ones are imported here, under private names, without polluting sys.modules.
    import importlib.util

    saved = {k: sys.modules.get(k) for k in ("config", "environment", "search", "model", "buffer", "worker")}
    for k in saved:
        sys.modules.pop(k, None)
    mods = {}
    try:
        for name in ("config", "environment", "search", "model", "buffer", "worker"):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_DIR, name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod  # reference modules import each other by bare name
            spec.loader.exec_module(mod)
            mods[name] = mod
        mods["environment"].np = _make_np_standin()
    finally:
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    _loaded.update(mods)
    return types.SimpleNamespace(**mods)


def load_fixture(path):
    """Restricted unpickler for the reference's test*.pkl scenario files (numpy arrays only)."""
    import pickle

    allowed = {
        ("numpy.core.multiarray", "_reconstruct"),
        ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"),
        ("numpy", "dtype"),
    }

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in allowed:
                return super().find_class(module, name)
            raise pickle.UnpicklingError("forbidden global %s.%s" % (module, name))

    with open(path, "rb") as f:
        return U(f).load()

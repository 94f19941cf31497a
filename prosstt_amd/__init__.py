"""
prosstt_amd -- the PROSSTT simulation hot path on AMD MI355X (gfx950).

Drop-in for the reference package's modules:

    from prosstt_amd import tree, simulation as sim, sim_utils as sut, count_model as cm

or, for unmodified scripts that say ``from prosstt import ...``:

    import prosstt_amd; prosstt_amd.install_as_prosstt()

Host code is Python; every O(time x genes) / O(cells x genes) step runs in
hand-written HIP kernels behind the C ABI of include/prosstt_amd.h.  No CPU fallback.
"""
import sys

__version__ = "0.1.0"


def install_as_prosstt():
    """Alias this package as ``prosstt`` in ``sys.modules`` (tree, simulation, sim_utils,
    count_model, tree_utils), so ``from prosstt import simulation as sim`` resolves here."""
    import importlib
    pkg = sys.modules[__name__]
    sys.modules["prosstt"] = pkg
    for name in ("tree", "simulation", "sim_utils", "count_model", "tree_utils"):
        sys.modules["prosstt." + name] = importlib.import_module(__name__ + "." + name)
    return pkg

"""pde_amd - the efficient_kan KAN layer and its Black-Scholes PINN train step,
MI355X-native (hand-written gfx950 HIP kernels behind a C ABI).

    from pde_amd import KAN, KANLinear          # same interface as efficient_kan
    pde_amd.install_as_efficient_kan()          # `from efficient_kan import KAN` resolves here
"""
from .kan import KAN, KANLinear  # noqa: F401

__all__ = ["KAN", "KANLinear", "install_as_efficient_kan"]


def install_as_efficient_kan() -> None:
    """Register this package under the name the reference scripts import
    (``from efficient_kan import KAN``, run1.py:2)."""
    import sys
    import types

    mod = types.ModuleType("efficient_kan")
    mod.KAN, mod.KANLinear = KAN, KANLinear
    mod.__all__ = ["KAN", "KANLinear"]
    sys.modules["efficient_kan"] = mod
    sys.modules["efficient_kan.kan"] = mod

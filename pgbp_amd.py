"""
Import shim: the product package lives in the directory `phylogaussianbeliefprop.jl_amd/`
(a name Python's import statement cannot spell because of the dot).  `import pgbp_amd`
loads that directory as the package `pgbp_amd`.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkgdir = os.path.join(_here, "phylogaussianbeliefprop.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "pgbp_amd", os.path.join(_pkgdir, "__init__.py"), submodule_search_locations=[_pkgdir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pgbp_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Helper of the opt-in import shims under dropin/ (INTEGRATION.md §2).

A shim for a module whose parity with the reference's third-party arithmetic is unpinned (OpenCV: `warp_learn.planes_utils`,
the Rodrigues epilogue of `utils.pnp_utils`) must, by default, BE the reference's own module.  `become_reference_module`
loads the reference checkout's file of the same dotted name - found further down the merged package's `__path__` - as a
regular module (own spec, `__file__`, `__package__`: relative imports and `__file__`-relative paths inside it keep working)
and installs it in `sys.modules` under that name, which is what the running `import` statement then returns."""
import importlib.util
import os
import sys


def become_reference_module(name: str, here_file: str):
    pkg_name, leaf = name.rsplit(".", 1)
    pkg = sys.modules[pkg_name]
    here = os.path.dirname(os.path.abspath(here_file))
    for d in pkg.__path__:
        f = os.path.join(d, leaf + ".py")
        if os.path.abspath(d) != here and os.path.exists(f):
            spec = importlib.util.spec_from_file_location(name, f)
            mod = importlib.util.module_from_spec(spec)
            mod.FUSG_DROPIN = False
            sys.modules[name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(name, None)
                raise
            setattr(pkg, leaf, mod)
            return mod
    raise ImportError(f"{name}: no reference checkout behind dropin/ on sys.path (the default of this shim is the reference's "
                      f"own {leaf}.py; set the shim's FUSG_DROPIN_* switch to use the MI355X version)")

"""Import alias: `import text2pos_amd` == the package directory `text2pos-cvpr2022_amd/` (not a valid identifier).

`text2pos_amd.<sub>` resolves to the SAME module object as `text2pos-cvpr2022_amd.<sub>` (a meta-path finder maps the
names): a dotted import through the alias must not execute a second copy of a submodule, whose classes would then fail
`isinstance` checks against the first copy's."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = "text2pos-cvpr2022_amd"
_ALIAS = __name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + "."):
            return None
        real = importlib.util.find_spec(_REAL + fullname[len(_ALIAS):])
        if real is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, origin=real.origin)

    def create_module(self, spec):
        module = importlib.import_module(_REAL + spec.name[len(_ALIAS):])
        self._real_spec = getattr(self, "_real_spec", {})
        self._real_spec[spec.name] = module.__spec__
        return module

    def exec_module(self, module):   # already executed under its real name; the import machinery re-labelled its spec
        for alias, spec in list(getattr(self, "_real_spec", {}).items()):
            if sys.modules.get(spec.name) is module:
                module.__spec__ = spec
                del self._real_spec[alias]

    def get_code(self, fullname):    # `python -m text2pos_amd.<sub>` (runpy executes the code object as __main__)
        real = _REAL + fullname[len(_ALIAS):]
        return importlib.util.find_spec(real).loader.get_code(real)


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg

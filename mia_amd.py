"""Import shim: the package directory is named `mapping-iterative-assembler_amd`
(with a hyphen, as the project layout prescribes), which Python cannot import by
name.  `import mia_amd` loads that directory as the package `mia_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mapping-iterative-assembler_amd")
_spec = importlib.util.spec_from_file_location("mia_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mia_amd"] = _mod
_spec.loader.exec_module(_mod)

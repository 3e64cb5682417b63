"""Import shim: the package directory is `java-sdr_amd/` (hyphenated by the repo's naming rule), which
Python cannot import by name.  `import java_sdr_amd` loads that directory as the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "java-sdr_amd")
_spec = importlib.util.spec_from_file_location("java_sdr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["java_sdr_amd"] = _mod
_spec.loader.exec_module(_mod)

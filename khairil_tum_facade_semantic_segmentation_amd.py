"""Import alias: the package directory carries the reference repo's name, which has a hyphen
(`khairil_tum-facade_semantic_segmentation_amd/`) and so cannot be named in an `import`
statement.  `import khairil_tum_facade_semantic_segmentation_amd as pn2` loads that directory
as a regular package under this (underscore) name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "khairil_tum-facade_semantic_segmentation_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

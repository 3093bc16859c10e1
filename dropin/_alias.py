"""shared by the alias packages: names that exist in passion_amd resolve to THOSE module objects (so `mmformer.patch_size = 8`
set through `models.mmformer` is seen by the product), every other submodule (utils.lr_scheduler, utils.parser, data.data_utils ...:
logging / schedules / host-side loaders outside the hot path, SURVEY 2 OUT-OF-SCOPE) is looked up in the reference's own
package of the same name further down sys.path."""
import importlib
import os
import sys


def alias(pkg_name, pkg_path, target_pkg, names):
    pkg = sys.modules[pkg_name]
    for n in names:
        mod = importlib.import_module(f"{target_pkg}.{n}")
        sys.modules[f"{pkg_name}.{n}"] = mod
        setattr(pkg, n, mod)
    # the reference's directory of the same name (code/<pkg_name>/) supplies what the product does not replace
    here = os.path.abspath(pkg_path[0])
    for entry in sys.path:
        cand = os.path.join(entry or ".", pkg_name)
        if os.path.isdir(cand) and os.path.abspath(cand) != here and cand not in pkg_path:
            pkg_path.append(cand)

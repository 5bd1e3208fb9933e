"""oracle/ref_extract.py -- TEST INFRASTRUCTURE ONLY.

Executes selected function / method bodies of reference files whose module-level imports
(torchvision, detectron2, fvcore) are not installed here: the chosen `def`s are cut out of the
file with `ast` and exec'd in a namespace that only has torch.  Nothing of the reference is
copied into the repo; this only runs in the build container to produce golden vectors."""
from __future__ import annotations

import ast
import math
import textwrap

import torch
import torch.nn.functional as F


def extract(path, names, extra_ns=None):
    """Return {name: function} for module-level functions and class methods called `names`."""
    src = open(path).read()
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            found[node.name] = textwrap.dedent(ast.get_source_segment(src, node))
    missing = set(names) - set(found)
    if missing:
        raise KeyError(f"{path}: {sorted(missing)} not found")
    ns = {"torch": torch, "F": F, "math": math, "nn": torch.nn}
    ns.update(extra_ns or {})
    for name in names:
        exec(compile(found[name], f"{path}:{name}", "exec"), ns)
    return {name: ns[name] for name in names}

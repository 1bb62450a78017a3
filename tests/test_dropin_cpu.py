"""CPU check of the module-level drop-in: every name the reference's train.py / make_submission.py
import resolves through dropin/ (skipped where the reference checkout is absent, e.g. on the GPU box)."""
import ast
import importlib
import os
import sys

import pytest

REF = '/root/reference'
STDLIB_OR_THIRD_PARTY = {'glob', 'os', 'numpy', 'pandas', 'tqdm', '__future__'}


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("script", ["train.py", "make_submission.py"])
def test_reference_script_imports_resolve(script, repo_root, monkeypatch):
    monkeypatch.syspath_prepend(repo_root)
    monkeypatch.syspath_prepend(os.path.join(repo_root, 'dropin'))
    for m in ('tensorflow', 'keras', 'keras.backend', 'keras.callbacks', 'keras.models', 'IPython', 'input_data',
              'model', 'utils', 'classes', 'callbacks', 'audio'):
        sys.modules.pop(m, None)
    tree = ast.parse(open(os.path.join(REF, script)).read())
    checked = 0
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module not in STDLIB_OR_THIRD_PARTY:
            mod = importlib.import_module(node.module)
            for a in node.names:
                assert hasattr(mod, a.name), (script, node.module, a.name)
                checked += 1
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name not in STDLIB_OR_THIRD_PARTY:
                    importlib.import_module(a.name)
                    checked += 1
    assert checked >= 8
    assert 'dropin' in sys.modules['input_data'].__file__

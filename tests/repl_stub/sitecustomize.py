"""TEST INFRASTRUCTURE: loaded automatically by the child interpreter of tests/test_reference_repl.py (this directory is on its PYTHONPATH)."""
import os
import sys

for p in os.environ.get("VCLA_REPL_PATHS", "").split(os.pathsep):
    if p and p not in sys.path:
        sys.path.insert(0, p)
if os.environ.get("VCLA_REPL_PATHS"):
    from oracle_backed import install
    install()

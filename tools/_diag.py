"""Shared by the diagnostic tools: select a variant / diagnostic build of the library (tools/diaglib/, made by
tools/diag_dense.sh and tools/build_variant.sh) BEFORE mistra_amd.chem loads it.  The product library
(mistra_amd/lib/libmistra_chem.so) reads none of the diagnostic environment switches."""
import os
import sys

REPO = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def lib_path(name):
    return os.path.join(REPO, 'mistra_amd', 'lib', name) if name == 'libmistra_chem.so' else os.path.join(REPO, 'tools', 'diaglib', name)


def diag_env(name, **extra):
    """Environment for a child process that is to run on build `name`."""
    path = lib_path(name)
    if not os.path.exists(path):
        sys.exit('%s is missing: build it first (tools/diag_dense.sh env|stamps|pN, tools/build_variant.sh NAME ...)' % path)
    return dict(os.environ, MISTRA_CHEM_LIB=path, MISTRA_MECH_DIR=os.path.join(REPO, 'mistra_amd', 'mech'), **extra)


def use_diag_lib(name='libdiag_env.so', **extra):
    os.environ.update(diag_env(name, **extra))

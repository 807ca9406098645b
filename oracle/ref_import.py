"""Import the REAL reference (jina-ai/annlite under /root/reference) for pinning the oracle.

TEST INFRASTRUCTURE ONLY -- never imported by ``annlite_amd`` (the product), by the
``-m gpu`` tests, by ``__graft_entry__.smoke()`` or by ``bench.py``.  It only works in the
build container, where ``/root/reference`` exists and ``oracle/build_ref.sh`` has compiled the
reference's two native extensions into ``oracle/_ref/`` (git-ignored; ``$ANNLITE_REF_BUILD`` overrides).  On the GPU box ``available()`` is False
and everything that depends on it is skipped; parity there rests on the committed golden
fixtures (``tests/golden/*.npz``) produced by ``tests/golden/make_golden.py`` through this module.

How it works: ``/root/reference`` is put on ``sys.path`` (bytecode writing
disabled, the tree is read-only), the packages the reference imports but this image lacks
(``docarray``, ``loguru``, ``rocksdict``) are replaced by ``MagicMock`` -- none of them is on the
PQ/ADC path -- and ``annlite.pq_bind`` / ``annlite.hnsw_bind`` are pre-seeded from that build directory.
No reference source is copied anywhere.
"""
import importlib.machinery
import importlib.util
import os
import sys
import sysconfig
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get('ANNLITE_REFERENCE', '/root/reference')
# where oracle/build_ref.sh puts the reference's compiled extensions: the repository's own, git-ignored oracle/_ref/ (a
# predictable world-writable /tmp path would let another local user plant a module this process then executes)
_REF_DIR = os.environ.get('ANNLITE_REF_BUILD') or os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')
_EXT = sysconfig.get_config_var('EXT_SUFFIX')

_loaded = None


def _trusted(path: str) -> bool:
    """the module file and its directory belong to this user and are not writable by anyone else"""
    try:
        for p in (path, os.path.dirname(path)):
            st = os.stat(p)
            if st.st_uid != os.getuid() or (st.st_mode & 0o022):
                return False
        return True
    except OSError:
        return False


def available() -> bool:
    path = os.path.join(_REF_DIR, 'pq_bind' + _EXT)
    return os.path.isdir(os.path.join(REF_ROOT, 'annlite')) and os.path.isfile(path) and _trusted(path)


def _load_ext(fullname: str, filename: str):
    path = os.path.join(_REF_DIR, filename + _EXT)
    if not os.path.isfile(path) or not _trusted(path):
        return None
    loader = importlib.machinery.ExtensionFileLoader(fullname, path)
    spec = importlib.util.spec_from_file_location(fullname, path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def load():
    """Return a namespace with the reference's hot-path symbols."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(
            'reference not available: need %s and the build of oracle/build_ref.sh in %s' % (REF_ROOT, _REF_DIR)
        )
    sys.dont_write_bytecode = True
    for name in ('docarray', 'docarray.math', 'docarray.math.ndarray', 'loguru', 'rocksdict'):
        sys.modules.setdefault(name, MagicMock())

    # a bare package object so that ``annlite/__init__.py`` (which pulls in AnnLite ->
    # docarray/rocksdict storage) is not executed; sub-modules import normally from disk.
    pkg = types.ModuleType('annlite')
    pkg.__path__ = [os.path.join(REF_ROOT, 'annlite')]
    sys.modules['annlite'] = pkg
    pq_bind = _load_ext('annlite.pq_bind', 'pq_bind')
    sys.modules['annlite.pq_bind'] = pq_bind
    pkg.pq_bind = pq_bind
    hnsw_bind = _load_ext('annlite.hnsw_bind', 'hnsw_bind')
    if hnsw_bind is not None:
        sys.modules['annlite.hnsw_bind'] = hnsw_bind
        pkg.hnsw_bind = hnsw_bind

    ns = types.SimpleNamespace()
    ns.pq_bind = pq_bind
    ns.hnsw_bind = hnsw_bind
    ns.math = importlib.import_module('annlite.math')
    ns.enums = importlib.import_module('annlite.enums')
    pq = importlib.import_module('annlite.core.codec.pq')
    ns.PQCodec = pq.PQCodec
    ns.DistanceTable = pq.DistanceTable
    ns.VQCodec = importlib.import_module('annlite.core.codec.vq').VQCodec
    ns.PQIndex = importlib.import_module('annlite.core.index.pq_index').PQIndex
    ns.FlatIndex = importlib.import_module('annlite.core.index.flat_index').FlatIndex
    ns.HnswIndex = None
    if hnsw_bind is not None:
        ns.HnswIndex = importlib.import_module('annlite.core.index.hnsw.index').HnswIndex
    ns.Metric = ns.enums.Metric
    _loaded = ns
    return ns

"""CPU: the C-ABI library loads and exports every symbol include/annlite_hip.h declares (no compute)."""
import os
import re
import subprocess

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'annlite_hip.h')).read()
    return sorted(set(re.findall(r'ANNLITE_API\s+(?:const\s+char\s+\*|int64_t\s+|int\s+)(annlite_\w+)\s*\(', src)))


def test_header_symbols_are_exported_and_bound():
    from annlite_amd import _capi

    declared = _declared()
    assert len(declared) >= 20
    assert sorted(_capi.SYMBOLS) == declared, set(declared) ^ set(_capi.SYMBOLS)
    lib = _capi.lib()  # raises if the .so is missing or lacks a symbol
    for name in declared:
        assert hasattr(lib, name)
    out = subprocess.check_output(['nm', '-D', '--defined-only', _capi.LIB_PATH]).decode()
    exported = set(re.findall(r'\bT (annlite_\w+)', out))
    assert exported == set(declared), exported ^ set(declared)
    assert lib.annlite_hip_abi_version() == 2


def test_scan_plan_arithmetic_no_gpu():
    from annlite_amd import _capi

    p = _capi.scan_plan(10_000_000, 16, 256, 1, 1024, 10)
    assert (p.fast, p.qi) == (1, 4) and p.qt in (8, 16, 32) and p.waves in (8, 12, 16) and (p.n_slices in (1, 2, 4) or p.n_slices % 8 == 0)
    assert p.lut_floats == 1024 * 16 * 256
    assert p.workspace_bytes >= 1024 * p.n_slices * 10 * 8
    p = _capi.scan_plan(1000, 64, 256, 1, 3, 10)
    assert (p.fast, p.qi, p.qt) == (1, 4, 8) and p.lut_floats == 16 * 64 * 256  # (M = 64, k <= 16: byte tables, 8 queries per workgroup)
    p = _capi.scan_plan(1000, 8, 256, 1, 5, 10)  # M = 8, uint8 codes, k <= 16 -> byte tables, 32 queries per workgroup (k > 16: u16 tables, 16)
    assert (p.fast, p.qi, p.qt) == (1, 4, 32) and _capi.scan_plan(1000, 8, 256, 1, 5, 40).qt == 16
    p = _capi.scan_plan(1000, 8, 512, 2, 5, 10)  # uint16 codes, M = 8, Ks <= 512, k <= 16 -> byte-table kernel, 32 queries per workgroup
    assert (p.fast, p.qi, p.qt) == (1, 4, 32)
    p = _capi.scan_plan(1000, 8, 512, 2, 5, 50)  # ... k > 16 -> u16-table kernel, 16 queries per workgroup
    assert (p.fast, p.qi, p.qt) == (1, 4, 16)
    p = _capi.scan_plan(1000, 8, 768, 2, 5, 10)  # 512 < Ks <= 1024: byte tables of one entry group, 16 queries per workgroup
    assert (p.fast, p.qi, p.qt) == (1, 4, 16)
    p = _capi.scan_plan(1000, 8, 768, 2, 5, 40)  # ... k > 16: u16 tables, 8 queries per workgroup
    assert (p.fast, p.qi, p.qt) == (1, 4, 8)
    p = _capi.scan_plan(1000, 32, 256, 1, 5, 10)  # M = 32, k <= 16 -> byte tables of one entry group, 16 queries per workgroup
    assert (p.fast, p.qi, p.qt, p.waves) == (1, 4, 16, 16) and p.lut_floats == 16 * 32 * 256
    p = _capi.scan_plan(1000, 32, 256, 1, 5, 17)  # ... k > 16: u16 tables, 8 queries per workgroup, 12 waves
    assert (p.fast, p.qi, p.qt, p.waves) == (1, 4, 8, 12)
    p = _capi.scan_plan(1000, 16, 768, 2, 5, 10)  # uint16 codes that do not -> generic kernel
    assert p.fast == 0 and p.qt == 1
    import pytest

    with pytest.raises(AssertionError):
        _capi.scan_plan(1000, 16, 256, 1, 4, 65)  # k > 64
    with pytest.raises(AssertionError):
        _capi.scan_plan(1000, 16, 256, 3, 4, 10)  # bad code width


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no file under annlite_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'annlite_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'pq_oracle' not in txt and 'oracle/' not in txt.replace('(oracle/', ''), os.path.join(dirpath, f)

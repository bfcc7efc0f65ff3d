"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/genome_amd.h
declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re
import random

import pytest

from genome_amd import _lib as L
from genome_amd import dna
from oracle import pyref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    return set(re.findall(r"\b(gk_[a-z0-9_]+)\s*\(", hdr)) - {"gk_status"}


def test_library_exports_every_declared_symbol():
    """The PRODUCT library exports exactly what include/genome_amd.h declares — and none of the test hooks; the TEST build adds
    what include/genome_amd_test.h declares (tests/conftest.py loads that one)."""
    declared = _declared("genome_amd.h")
    hooks = _declared("genome_amd_test.h") - declared          # (its comments name product entry points too)
    assert len(declared) >= 35 and hooks
    product = C.CDLL(os.path.join(ROOT, "genome_amd", "libgenome_amd.so"))
    missing = [s for s in sorted(declared) if not hasattr(product, s)]
    assert not missing, missing
    assert not [s for s in sorted(hooks) if hasattr(product, s)], "test hooks in the product library"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    test_build = C.CDLL(L.TEST_LIB_PATH)
    assert not [s for s in sorted(declared | hooks) if not hasattr(test_build, s)]
    assert hooks == set(L.TEST_SIGNATURES), hooks ^ set(L.TEST_SIGNATURES)
    assert os.path.samefile(L.LIB_PATH, L.TEST_LIB_PATH) or "variants" in L.LIB_PATH      # what this test session has loaded


def test_integration_md_accounts_for_every_declared_symbol():
    """INTEGRATION.md shows the reference-side binding of the path's entry points (its JNI layer, §2-§4) and lists the rest with
    the reason they are not bound (§7): together they must cover the header."""
    declared = _declared("genome_amd.h")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(s for s in declared if s not in text)
    assert not missing, missing


def test_no_cpu_fallback_without_device(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    h = L.vp()
    rc = L.lib().gk_ctx_create(0, C.byref(h))
    assert rc == L.GK_E_NODEVICE
    assert b"no HIP device" in L.lib().gk_last_error(None)
    from genome_amd.dnamap import Context
    with pytest.raises(L.GkError):
        Context(0)


def test_null_handles_are_errors_not_crashes():
    lib = L.lib()
    n = C.c_uint64()
    assert lib.gk_map_size(None, C.byref(n)) == L.GK_E_INVALID
    assert lib.gk_map_filter_lt(None, 3) == L.GK_E_INVALID
    assert lib.gk_ctx_sync(None) == L.GK_E_INVALID
    lib.gk_map_destroy(None)
    lib.gk_ctx_destroy(None)
    lib.gk_graph_destroy(None)


@pytest.mark.parametrize("k", [5, 11, 21, 31, 34, 47, 55, 63, 64])
def test_owner_is_strand_symmetric(k):
    """The owner function must send x and rc(x) — hence both hash-rule candidates, incl. the tie
    case (FreqFilter.scala:31-32) — to the same partition."""
    rnd = random.Random(k)
    lib = L.lib()
    seen = set()
    for _ in range(300):
        s = "".join(rnd.choice("AGCT") for _ in range(k))
        lo, hi = dna.pack(s)
        rlo, rhi = dna.pack(R.rev_comp(s))
        for P in (1, 2, 3, 8, 14, 64):
            a, b = lib.gk_owner_of(k, lo, hi, P), lib.gk_owner_of(k, rlo, rhi, P)
            assert a == b and 0 <= a < P
        seen.add(lib.gk_owner_of(k, lo, hi, 8))
    assert len(seen) == 8 or k < 8
    assert lib.gk_owner_of(32, 1, 0, 4) == -1 and lib.gk_owner_of(31, 1, 0, 0) == -1


def test_host_dna_helpers_match_pyref():
    rnd = random.Random(3)
    for k in (1, 7, 31, 32, 33, 64):
        s = "".join(rnd.choice("AGCT") for _ in range(k))
        assert dna.pack(s) == R.pack(s)
        assert dna.unpack(*dna.pack(s), k) == s
        assert dna.rev_complement(s) == R.rev_comp(s)
    reads = ["", "A", "AGCTT", "".join(rnd.choice("AGCT") for _ in range(255))]
    assert dna.reads_to_bin(reads) == R.reads_to_bin(reads)

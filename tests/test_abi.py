"""The C-ABI library loads on a machine without a GPU and exports every symbol include/polr_hip.h
declares; nothing is computed here."""
import os
import re

import pytest

import common
from polr_amd import capi


def header_functions():
    text = open(os.path.join(common.ROOT, "include", "polr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(polr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libpolr_hip.so does not export %s" % n
    assert sorted(capi.EXPORTS) == names, "python binding and header disagree"
    assert lib.polr_abi_version() == 1


def test_no_device_fails_loudly():
    """without a gfx950 device the context cannot be created -- there is no CPU fallback"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.PolrError) as e:
        capi.Context(0)
    assert e.value.code == capi.E_NO_DEVICE


def test_product_never_imports_oracle():
    """the product tree must not reference the oracle in any form"""
    bad = []
    pkg = os.path.join(common.ROOT, "duckdb-polr_amd")
    for root, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", "Makefile")):
                t = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"polr_oracle|oracle/|import oracle|from oracle", t):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_library_reads_no_tuning_from_the_environment():
    """pool tuning travels through polr_ctx_set_pool_tuning (include/polr_hip.h); the device library has no getenv"""
    pkg = os.path.join(common.ROOT, "duckdb-polr_amd", "csrc")
    bad = []
    for f in os.listdir(pkg):
        if f.endswith((".hip", ".h")):
            # (the one environment variable the library knows switches a debugging trace on: POLR_DEBUG_HIP_ERRORS)
            bad += [(f, v) for v in re.findall(r'getenv\("([^"]*)"\)', open(os.path.join(pkg, f)).read())
                    if not v.startswith("POLR_DEBUG_")]
    assert not bad, bad
    lib = capi.load()
    assert lib.polr_ctx_set_pool_tuning(None, None) == capi.E_INVALID

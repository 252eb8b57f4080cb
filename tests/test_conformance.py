"""Boundary conformance (build container only: needs /root/reference): the host mirror's operator classes compiled in ONE
translation unit with the reference's own headers, member-function shapes and enumerations static_assert'ed equal
(tests/conformance/check_signatures.cpp; reference interfaces: physical_operator.hpp:130-160,
physical_multiplexer.hpp:27-48, physical_adaptive_union.hpp:21-33, physical_hash_join.hpp:61-73)."""
import os
import subprocess

import pytest

import common

REF = os.environ.get("POLR_REFERENCE", "/root/reference")
SRC = os.path.join(common.ROOT, "tests", "conformance", "check_signatures.cpp")


def _compile(extra=()):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + os.path.join(REF, "src", "include"),
           "-I" + os.path.join(common.ROOT, "duckdb-polr_amd", "host"), "-I" + os.path.join(common.ROOT, "include"),
           *extra, SRC]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "include")), reason="the reference is not mounted here")
def test_operator_signatures_match_the_reference_headers():
    r = _compile()
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "include")), reason="the reference is not mounted here")
def test_the_checker_rejects_a_mismatch():
    r = _compile(["-DCONF_NEGATIVE"])
    assert r.returncode != 0 and "does not have the shape of" in r.stderr


ADAPTER_SRC = os.path.join(common.ROOT, "tests", "conformance", "adapter_main.cpp")
ADAPTER_BIN = os.path.join(common.ROOT, "oracle", "_ref", "adapter_test")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "include")), reason="the reference is not mounted here")
def test_duckdb_adapter_compiles_against_the_reference_headers():
    """duckdb-polr_amd/host/duckdb_adapter/polr_duckdb_adapter.hpp -- INTEGRATION.md section 2 over the REAL duckdb:: types
    (JoinHashTable, RowLayout, RowDataCollection, BufferManager, DataChunk, UnifiedVectorFormat) -- and its driver are one
    translation unit with the reference's own headers"""
    tp = os.path.join(REF, "third_party")
    inc = ["-I" + os.path.join(REF, "src", "include")] + ["-I" + os.path.join(tp, d) for d in (
        "fsst", "fmt/include", "hyperloglog", "fastpforlib", "fast_float", "re2", "miniz", "utf8proc/include", "miniparquet",
        "concurrentqueue", "pcg", "tdigest", "mbedtls/include", "jaro_winkler", "libpg_query/include", "httplib")]
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-w", "-DDUCKDB", "-DDUCKDB_MAIN_LIBRARY", "-DNDEBUG", *inc,
           "-I" + os.path.join(common.ROOT, "include"), "-I" + os.path.join(common.ROOT, "duckdb-polr_amd", "host"), ADAPTER_SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(ADAPTER_BIN), reason="oracle/_ref/adapter_test is built in the build container "
                                                            "(make -f oracle/ref_build.mk adapter)")
def test_duckdb_adapter_against_the_reference_hash_table():
    """the adapter, linked against the reference compiled from its sources: duckdb::JoinHashTable built and finalized by the
    reference's own code, uploaded through PolrUploadBuildSide, every probe DataChunk answered by JoinHashTable::Probe +
    ScanStructure::Next AND by the device through the adapter -- equal result rows on every chunk, for `=` and for
    IS NOT DISTINCT FROM (NULL keys and NULL payloads on both sides, repeated keys)"""
    r = subprocess.run([ADAPTER_BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "device == reference on every chunk" in r.stdout and "IS NOT DISTINCT FROM" in r.stdout

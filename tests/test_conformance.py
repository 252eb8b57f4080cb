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

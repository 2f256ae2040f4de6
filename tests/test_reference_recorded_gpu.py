"""The GPU path against outputs of the REFERENCE BINARY itself -- not against the restatement.

SURVEY.md Appendix A records three complete .lrz files the survey's build of lrzip-next 0.14.0 wrote for one seeded 64 MiB
input (tests/gen_syn.py regenerates it; tests/golden/known_answers.json holds size + sha256 of each).  The oracle is checked
against them on the CPU (tests/test_oracle_golden.py); here liblrzgpu.so compresses the same input with the same
parameters through the C ABI and must produce the same bytes: rzip scan, token format, lz4 gate, block sizing (both
branches of src/stream.c:1311-1323), LZMA, container, MD5 trailer."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))["reference_recorded"]
sha = lambda b: hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def syn64():
    import gen_syn
    d = gen_syn.syn64()
    assert len(d) == KA["input"]["size"] and sha(d) == KA["input"]["sha256"]
    return d


def test_k1_pinned_parameters(B, syn64):
    """`-L7 -p4 -m80`: five slots of 71 729 152 B, one LZMA block of 33 554 463 literal bytes + one of stream 0."""
    rec = KA["k1"]
    img, _ = B.compress_buffer(syn64, level=7, threads=4, processors=8, ramsize=80 * 100 * 1048576, host_threads=8)
    assert (len(img), sha(img)) == (rec["size"], rec["sha256"])
    plan, _ = B.plan(len(syn64), level=7, threads=4, processors=8, ramsize=80 * 100 * 1048576)
    assert plan.stream_bufsize == rec["stream_bufsize"]


def test_k2_rzip_only(B, syn64):
    """`-n -p1 -m80`: the raw streams of rzip level 7 (513 matches, 515 literal runs) in the container, no back end."""
    rec = KA["k2"]
    img, _ = B.compress_buffer(syn64, level=7, no_compress=True, threads=1, processors=1, ramsize=80 * 100 * 1048576, host_threads=4)
    assert (len(img), sha(img)) == (rec["size"], rec["sha256"])
    assert img[-16:] == hashlib.md5(syn64).digest()


def test_k3_reference_defaults(B, syn64):
    """The reference's defaults on the survey container (8 cores => 9 slots, 62 GiB): 10 MiB blocks, seven LZMA blocks."""
    rec = KA["k3"]
    img, _ = B.compress_buffer(syn64, level=7, threads=8, processors=rec["processors"], ramsize=rec["ramsize"], host_threads=8)
    assert (len(img), sha(img)) == (rec["size"], rec["sha256"])
    plan, _ = B.plan(len(syn64), level=7, threads=8, processors=rec["processors"], ramsize=rec["ramsize"])
    assert plan.stream_bufsize == rec["stream_bufsize"]

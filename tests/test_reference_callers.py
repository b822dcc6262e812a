"""Source-level drop-in: code of the reference's own callers, compiled FROM WHERE IT LIES under /root/reference against the C++ face
(colibri-core_amd/host/include) and linked with the C++ face + the HIP library. Build container only: /root/reference does not exist on the GPU box
(the tests skip there). No reference source is copied into the repository: translation units are generated in a temporary directory and #include or
quote the reference files in place.

  * src/extractngrams.cpp — the sliding-window tool (Pattern(istream), Pattern::ngrams, PatternPointer::tostring, ClassDecoder): compiled, linked,
    and RUN here (it needs no device) on the reference's hamlet fixture; its output is checked against the n-grams of the decoded text.
  * src/test.cpp:1206-1232 — the hot-path block of the reference's test driver (PatternModelOptions, PatternModel<uint32_t>(&corpus), train(file,
    options), size/types/tokens/totalwordtypesingroup): the lines are taken from the file at test time and compiled inside a main() that provides what the
    surrounding test driver provides (corpus, file names, the test() helper). Compile + link only: training needs a GPU.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"
HOST = os.path.join(ROOT, "colibri-core_amd", "host")
LIB = os.path.join(ROOT, "colibri-core_amd", "lib")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference sources exist only in the build container")


def build(tmp_path, source, name, extra=()):
    subprocess.check_call(["make", "-s", "-C", HOST])  # libcolibri_amd_host.a (libcolibri_hip.so is built by __graft_entry__.build())
    if not os.path.exists(os.path.join(LIB, "libcolibri_hip.so")):
        pytest.skip("libcolibri_hip.so not built")
    (tmp_path / "config.h").write_text('#define VERSION "0"\n')  # what the autotools build would generate: a version string for the usage text
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-I" + str(tmp_path), "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(ROOT, "include"), *extra, source,
           os.path.join(LIB, "libcolibri_amd_host.a"), "-L" + LIB, "-lcolibri_hip", "-Wl,-rpath," + LIB, "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
           "-lpthread", "-o", exe]  # (the sharded driver in the host library talks to RCCL and the HIP runtime directly)
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
    return exe


def test_reference_extractngrams_compiles_and_runs_against_the_cpp_face(tmp_path):
    exe = build(tmp_path, os.path.join(REF, "src", "extractngrams.cpp"), "extractngrams")
    cls, dat = os.path.join(ROOT, "tests", "golden", "hamlet.colibri.cls"), os.path.join(ROOT, "tests", "golden", "hamlet.v2.colibri.dat")
    out = subprocess.run([exe, "-n", "3", "-c", cls, dat], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-2000:]
    got = [l for l in out.stdout.split("\n") if l]
    # the same file compiled against the reference's own library (the objects oracle/Makefile builds from the reference's sources): byte-identical output
    objs = [os.path.join(ROOT, "oracle", "_ref", f + ".o") for f in ("SpookyV2", "common", "algorithms", "classdecoder", "classencoder", "pattern", "patternmodel")]
    if not all(os.path.exists(o) for o in objs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    ref_exe = str(tmp_path / "extractngrams_ref")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(REF, "include"), "-idirafter", "/opt/conda/include", os.path.join(REF, "src", "extractngrams.cpp"), *objs,
                           "/lib/x86_64-linux-gnu/libbz2.so.1.0", "-o", ref_exe])
    ref = subprocess.run([ref_exe, "-n", "3", "-c", cls, dat], capture_output=True, text=True, timeout=60)
    assert ref.returncode == 0
    assert out.stdout == ref.stdout and len(got) > 100
    # and what that output is: the 3-token windows of every line, decoded (the reference's tool does not skip the two header bytes of a v2 file: they
    # decode as one unknown token at the head of the first line)
    words = {}
    for line in open(cls, encoding="utf-8"):
        k, w = line.rstrip("\n").split("\t", 1)
        words[int(k)] = w
    want, sent, val, shift = [], [], 0, 0
    for b in open(dat, "rb").read():
        val |= (b & 0x7F) << shift
        shift += 7
        if b < 128:
            if val == 0:
                want += [" ".join(sent[i:i + 3]) for i in range(len(sent) - 2)]
                sent = []
            else:
                sent.append(words.get(val, "{?}"))
            val = shift = 0
    assert got[1:50] == want[1:50]


def test_reference_test_driver_hot_path_block_compiles_against_the_cpp_face(tmp_path):
    lines = open(os.path.join(REF, "src", "test.cpp"), encoding="utf-8").read().split("\n")
    block = "\n".join(lines[1205:1232])  # 1206..1232: from "--- unindexed model without skipgrams ---" to the second model's token check
    assert "PatternModelOptions options;" in block and "unindexedmodelNSR.train(infilename, options);" in block and "totalwordtypesingroup(0, 1)" in block
    tu = tmp_path / "hotpath_block.cpp"
    tu.write_text('#include <iostream>\n#include <string>\n#include "patternmodel.h"\n#include "classdecoder.h"\nusing namespace std;\n'
                  'template <class A, class B> void test(const A& a, const B& b) { cerr << (a == (A)b ? " ok" : " FAILED") << endl; }\n'
                  'int main(int argc, char** argv) {\n    if (argc < 2) return 0;\n    IndexedCorpus corpus(argv[1]);\n    std::string infilename = argv[1];\n    {\n'
                  + block + '\n    }\n    return 0;\n}\n')
    build(tmp_path, str(tu), "hotpath_block")

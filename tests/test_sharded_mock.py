"""CPU (no GPU): the product's own multi-GPU driver — host/src/sharded.cpp, unchanged — over a CPU stand-in for the device layer (host/mock/mock_device.cpp ->
lib/libcolibri_sharded_mock.so): rank threads, rendezvous, the agreement before every exchange, the routing of sizes and buffers through the all-to-alls and
all-reduces, "None found" termination, and what happens when one rank fails. Round 3 tested a numpy restatement of the protocol instead of the driver.
Each case runs in its own process (the library reads COLIBRI_SHARDED_LIB / COLIBRI_FAULT once)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "colibri-core_amd", "lib", "libcolibri_sharded_mock.so")

SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import conftest, oracle
from colibri_amd import capi
world, name, maxlength, thr = int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
payload = conftest.small_corpora()[name]
want = oracle.train(payload, thr, maxlength)
with capi.ShardedTrainer(world, devices=[0] * world) as tr:
    tr.upload_split(payload)
    for rep in range(2):  # the trainer keeps its shards between runs
        st = tr.train(mintokens=thr, maxlength=maxlength)
        got = tr.export_dict()
        assert got == want.counts, ("model differs", len(got), len(want.counts))
        assert (st.totaltokens, st.totaltypes, st.maxn, st.npatterns) == (want.tokens, want.types, want.maxn, len(want.counts)), (st.totaltokens, st.totaltypes, st.maxn, st.npatterns)
        for n in range(1, min(maxlength, 20) + 1):
            assert (st.found[n], st.kept[n]) == (want.stats[n][0], want.stats[n][2]), n
        assert tr.info.protocol == 0 and tr.info.rccl == 0
print("OK")
"""


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "colibri-core_amd", "host"), "mock"])


def run(args, fault=None, timeout=120):
    build()
    env = dict(os.environ, COLIBRI_SHARDED_LIB=MOCK, COLIBRI_NO_RCCL="1")
    if fault:
        env["COLIBRI_FAULT"] = fault
    return subprocess.run([sys.executable, "-c", SCRIPT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")] + [str(a) for a in args], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("name,maxlength,thr", [("zipf20k", 5, 2), ("rand_noempty", 8, 3), ("repeat", 4, 1), ("short_sentences", 5, 2), ("empty", 5, 2), ("one_token", 3, 2),
                                                ("only_delims", 5, 2), ("one_long_sentence", 5, 2)])
def test_the_cxx_driver_builds_the_oracles_model_on_the_mock(world, name, maxlength, thr):
    p = run([world, name, maxlength, thr])
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("fault", ["2:colibri_kshard_begin", "1:colibri_kshard_uni_count", "0:colibri_kshard_uni_apply", "3:colibri_kshard_emit", "2:colibri_kshard_recv_buffers",
                                   "1:colibri_kshard_count", "0:colibri_kshard_feedback_buffers", "3:colibri_kshard_apply", "1:colibri_kshard_local_stats"])
def test_a_failing_rank_takes_all_ranks_out_of_the_run_together(fault):
    """one rank of four reports a failure at one step: nobody may stay behind in a barrier (the subprocess would time out); every rank leaves at the same agreement,
    the trainer turns to the candidate exchange — which the mock does not have — and the run ends with an error that names it, on a trainer that can be destroyed"""
    p = run([4, "zipf20k", 5, 2], fault=fault, timeout=60)
    assert p.returncode != 0
    assert "failed on rank(s) " + fault.split(":")[0] in p.stderr, p.stderr[-1500:]
    assert "candidate exchange" in p.stderr and "not mocked" in p.stderr, p.stderr[-1500:]

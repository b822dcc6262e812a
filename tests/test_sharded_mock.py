"""CPU (no GPU): the product's own multi-GPU driver — host/src/sharded.cpp, unchanged — over a CPU stand-in for the device layer (tests/standin/mock_device.cpp ->
lib/libcolibri_sharded_mock.so): rank threads, rendezvous, the agreement before every exchange, the routing of sizes and buffers through the all-to-alls and
all-reduces, "None found" termination, and what happens when one rank fails. Round 3 tested a numpy restatement of the protocol instead of the driver.
Both protocols: key-sharded counting (plain models) and the candidate exchange — what exhaustive-skipgram models (BASELINE configs[3]), indexed models and
indexed skipgram models (configs[4]) take at N > 1, pass by pass, level by level — against the oracle's model of the whole corpus.
Each case runs in its own process (the library reads COLIBRI_SHARDED_LIB / COLIBRI_FAULT once)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "standin", "lib", "libcolibri_sharded_mock.so")

SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import conftest, oracle
from colibri_amd import capi
world, name, maxlength, thr = int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
indexed = len(sys.argv) > 7 and sys.argv[7] == "indexed"
payload = conftest.small_corpora()[name]
want = oracle.train(payload, thr, maxlength, indexed=indexed)
with capi.ShardedTrainer(world, devices=[0] * world) as tr:
    tr.upload_split(payload)
    for rep in range(2):  # the trainer keeps its shards between runs
        st = tr.train(mintokens=thr, maxlength=maxlength, indexed=int(indexed))
        got = tr.export_dict()
        assert got == want.counts, ("model differs", len(got), len(want.counts))
        assert (st.totaltokens, st.totaltypes, st.maxn, st.npatterns) == (want.tokens, want.types, want.maxn, len(want.counts)), (st.totaltokens, st.totaltypes, st.maxn, st.npatterns)
        for n in range(1, min(maxlength, 20) + 1):
            assert (st.found[n], st.kept[n]) == (want.stats[n][0], want.stats[n][2]), n
        assert tr.info.protocol == 0 and tr.info.rccl == 0
print("OK")
"""


KINDS = {"u": {}, "us": dict(doskipgrams_exhaustive=True), "usT1": dict(doskipgrams_exhaustive=True, minskiptypes=1), "usy3": dict(doskipgrams_exhaustive=True, mintokens_skipgrams=3),
         "i": dict(indexed=True), "is": dict(indexed=True, doskipgrams=True), "isT1": dict(indexed=True, doskipgrams=True, minskiptypes=1), "isT3": dict(indexed=True, doskipgrams=True, minskiptypes=3),
         "uW3": dict(mintokens_unigrams=3), "iW4": dict(indexed=True, mintokens_unigrams=4)}  # (the word threshold, reference include/patternmodel.h:1019-1022: longer windows need every word at it)

SCRIPT_CANDIDATES = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import conftest, oracle
from colibri_amd import capi
world, name, maxlength, thr, kind = int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), eval(sys.argv[7])
payload = conftest.small_corpora()[name]
want = oracle.train(payload, thr, maxlength, **kind)
with capi.ShardedTrainer(world, devices=[0] * world) as tr:
    tr.upload_split(payload)
    if not any("skipgram" in k for k in kind):
        tr.set_protocol(1)  # a plain or indexed model without skipgrams would be counted key-sharded
    for rep in range(2):
        st = tr.train(mintokens=thr, maxlength=maxlength, **{k: int(v) for k, v in kind.items()})
        got = tr.export_dict()
        assert tr.info.protocol == 1 and tr.info.rccl == 0
        assert got == want.counts, ("model differs", len(got), len(want.counts), sorted(set(got.items()) ^ set(want.counts.items()))[:6])
        assert (st.totaltokens, st.totaltypes, st.maxn, st.npatterns) == (want.tokens, want.types, want.maxn, len(want.counts)), (st.totaltokens, st.totaltypes, st.maxn, st.npatterns)
        for n in range(1, min(maxlength, 20) + 1):
            assert (st.found[n], st.kept[n]) == (want.stats[n][0], want.stats[n][2]), (n, st.found[n], st.kept[n], want.stats[n])
print("OK")
"""


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "standin"), "mock"])


def run(args, fault=None, timeout=120, script=SCRIPT, env_extra=None):
    build()
    env = dict(os.environ, COLIBRI_SHARDED_LIB=MOCK, COLIBRI_NO_RCCL="1")
    if fault:
        env["COLIBRI_FAULT"] = fault
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", script, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")] + [str(a) for a in args], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("name,maxlength,thr", [("zipf20k", 5, 2), ("rand_noempty", 8, 3), ("repeat", 4, 1), ("short_sentences", 5, 2), ("empty", 5, 2), ("one_token", 3, 2),
                                                ("only_delims", 5, 2), ("one_long_sentence", 5, 2)])
def test_the_cxx_driver_builds_the_oracles_model_on_the_mock(world, name, maxlength, thr):
    p = run([world, name, maxlength, thr])
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("name,maxlength,thr", [("zipf20k", 5, 2), ("rand_noempty", 8, 3), ("repeat", 4, 1), ("short_sentences", 5, 2), ("one_token", 3, 2)])
def test_indexed_models_are_counted_key_sharded(world, name, maxlength, thr):
    """an indexed model takes the key-sharded protocol too (its feedback numbers the survivors of the last order as well: the references are keyed by those numbers);
    counts and figures vs the oracle here, the reference lists through the CLI below (hamlet.i at 2 and 4 ranks)"""
    p = run([world, name, maxlength, thr, "indexed"])
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("kind", ["u", "us", "usT1", "usy3", "i", "is", "isT1", "isT3"])
@pytest.mark.parametrize("name,maxlength,thr", [("zipf20k", 5, 2), ("rand3", 6, 2), ("repeat", 9, 3), ("one_long_sentence", 5, 2)])
def test_the_cxx_driver_builds_every_model_kind_by_candidate_exchange(world, kind, name, maxlength, thr):
    """train_candidates (host/src/sharded.cpp): order 1 by all-reduce of the class-indexed arrays, every n-gram pass, every level of every gap mask of the
    exhaustive-skipgram orders (reference include/patternmodel.h:1163-1171) and of IndexedPatternModel::trainskipgrams (:2969-3010: distinct-filler counts travel
    with the last level), termination at "None found" — the union of what the ranks export is the oracle's model, with its per-order found / kept figures"""
    p = run([world, name, maxlength, thr, repr(KINDS[kind])], script=SCRIPT_CANDIDATES)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("world,name,maxlength,thr,kind", [(3, "zipf20k", 5, 2, "us"), (3, "rand_noempty", 5, 2, "is"), (2, "zipf20k", 4, 1, "us"), (2, "rand2", 4, 1, "isT1"), (4, "empty", 5, 2, "us"),
                                                           (4, "one_token", 5, 2, "is"), (2, "only_delims", 5, 2, "i"), (8, "short_sentences", 5, 2, "us"), (2, "cls_2p21", 5, 2, "us"),
                                                           (5, "zipf200k_phrases", 4, 2, "u"), (2, "zipf20k", 5, 2, "uW3"), (4, "zipf20k", 4, 2, "iW4"), (3, "rand_noempty", 5, 2, "uW3")])
def test_candidate_exchange_edges(world, name, maxlength, thr, kind):
    """a world that is not a power of two (never key-sharded), threshold 1 (every window and every masked form kept), ranks whose shard is empty, wide class ids"""
    p = run([world, name, maxlength, thr, repr(KINDS[kind])], script=SCRIPT_CANDIDATES, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


def test_order_1_by_key_exchange_when_a_rank_is_not_canonical():
    """a rank whose class encoding is not canonical takes every rank to the keyed unigram pass (sharded.cpp unigrams_dense -> false)"""
    p = run([2, "zipf20k", 4, 2, repr(KINDS["us"])], script=SCRIPT_CANDIDATES, env_extra={"COLIBRI_MOCK_NO_DENSE_UNIGRAMS": "1"})
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


KSHARD_STEPS = ["2:colibri_kshard_begin", "1:colibri_kshard_uni_count", "0:colibri_kshard_uni_apply", "3:colibri_kshard_emit", "2:colibri_kshard_recv_buffers", "1:colibri_kshard_count",
                "0:colibri_kshard_feedback_buffers", "3:colibri_kshard_apply", "1:colibri_kshard_local_stats"]

SCRIPT_FALLBACK = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import conftest, oracle
from colibri_amd import capi
world, name, maxlength, thr = int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
payload = conftest.small_corpora()[name]
want = oracle.train(payload, thr, maxlength)
with capi.ShardedTrainer(world, devices=[0] * world) as tr:
    tr.upload_split(payload)
    st = tr.train(mintokens=thr, maxlength=maxlength)
    assert tr.info.protocol == 1, "the run did not fall back"
    assert tr.export_dict() == want.counts
    assert (st.totaltokens, st.totaltypes, st.maxn, st.npatterns) == (want.tokens, want.types, want.maxn, len(want.counts))
print("OK")
"""


@pytest.mark.parametrize("fault", KSHARD_STEPS)
def test_a_failing_rank_takes_all_ranks_out_of_the_run_together(fault):
    """one rank of four reports a failure at one step: nobody may stay behind in a barrier (the subprocess would time out); every rank leaves at the same agreement,
    the trainer turns to the candidate exchange; with that switched off in the mock the run ends with an error that names it, on a trainer that can be destroyed"""
    p = run([4, "zipf20k", 5, 2], fault=fault, timeout=60, env_extra={"COLIBRI_MOCK_NO_CANDIDATES": "1"})
    assert p.returncode != 0
    assert "failed on rank(s) " + fault.split(":")[0] in p.stderr, p.stderr[-1500:]
    assert "candidate exchange" in p.stderr and "switched off" in p.stderr, p.stderr[-1500:]


@pytest.mark.parametrize("fault", KSHARD_STEPS)
def test_a_key_sharded_run_that_gives_up_is_repeated_by_candidate_exchange(fault):
    """the same failures with the whole driver behind them: all ranks leave the key-sharded run together, repeat it with the candidate exchange (what a record
    region or final bin that overflows on one rank leads to on the device) and build the oracle's model"""
    p = run([4, "zipf20k", 5, 2], fault=fault, timeout=60, script=SCRIPT_FALLBACK)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
    assert "failed on rank(s) " + fault.split(":")[0] in p.stderr and "repeating the run with the candidate exchange" in p.stderr, p.stderr[-1500:]


@pytest.mark.parametrize("kind", ["us", "is"])
@pytest.mark.parametrize("fault", ["1:colibri_shard_begin", "2:colibri_shard_uni_count", "0:colibri_shard_count", "3:colibri_shard_send_view", "1:colibri_shard_merge", "2:colibri_shard_reply",
                                   "3:colibri_shard_apply"])
def test_a_failing_rank_of_a_candidate_exchange_ends_the_run_on_every_rank(fault, kind):
    """the candidate exchange has nothing to fall back to: a step that fails on one rank ends the run on all four — at the agreement that follows it, or (the apply step,
    which no agreement follows) through the aborted rendezvous — with an error that names the step; nobody hangs, the trainer can be destroyed"""
    p = run([4, "zipf20k", 5, 2, repr(KINDS[kind])], fault=fault, timeout=60, script=SCRIPT_CANDIDATES)
    assert p.returncode != 0
    if "apply" in fault:
        assert fault.split(":")[1] + ": injected fault" in p.stderr, p.stderr[-1500:]
    else:  # (the message of whichever rank reported first: it names the failing rank, not necessarily the reason)
        assert "failed on rank(s) " + fault.split(":")[0] in p.stderr, p.stderr[-1500:]


# ---- the trainer's RCCL back end over the stand-in's in-process RCCL: what `bench.py --gpus N` and `colibri-patternmodeller --gpus N` run on N devices --------------------------
SCRIPT_RCCL = r"""
import sys, threading
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import conftest, oracle
from colibri_amd import capi
from colibri_amd.dist import shard_payload
world, name, maxlength, thr, kind, mode = int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), eval(sys.argv[7]), sys.argv[8]
payload = conftest.small_corpora()[name]
want = oracle.train(payload, thr, maxlength, **kind)
kw = dict(mintokens=thr, maxlength=maxlength, **{k: int(v) for k, v in kind.items()})
if mode == "threads":
    with capi.ShardedTrainer(world) as tr:
        tr.upload_split(payload)
        for rep in range(2):
            st = tr.train(**kw)
            assert tr.info.rccl == 1, "not on the RCCL back end"
            assert tr.export_dict() == want.counts
            assert (st.totaltokens, st.totaltypes, st.maxn, st.npatterns) == (want.tokens, want.types, want.maxn, len(want.counts))
        print("protocol", tr.info.protocol, "a2a bytes", tr.info.alltoall_bytes)
else:
    uid = capi.sharded_unique_id()
    shards = shard_payload(payload, world)
    got, errs, stats, shares = [None] * world, [], [None] * world, [None] * world
    def rank_main(r):
        try:
            with capi.ShardedTrainer(world, nlocal=1, first_rank=r, devices=[r], unique_id=uid) as tr:
                tr.upload(0, shards[r][0], shards[r][1])
                for rep in range(2):
                    stats[r] = tr.train(**kw)
                    assert tr.info.rccl == 1
                    got[r] = tr.export_dict()
                    if kind.get("indexed"):
                        shares[r] = tr.export_local(0)
        except Exception as e:
            errs.append((r, repr(e)))
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    union = {}
    for g in got:
        assert not (set(g) & set(union)), "a pattern exported by two ranks"
        union.update(g)
    assert union == want.counts, (len(union), len(want.counts))
    for st in stats:
        assert (st.totaltokens, st.totaltypes, st.maxn) == (want.tokens, want.types, want.maxn)
        for n in range(1, min(maxlength, 20) + 1):
            assert (st.found[n], st.kept[n]) == (want.stats[n][0], want.stats[n][2]), n
    assert sum(st.npatterns for st in stats) == len(want.counts)
    if kind.get("indexed"):  # every rank's forward index by global number (sentences numbered from the first_sentence its caller gave): merged in rank order = the oracle's lists
        from colibri_amd.dist import merge_exports
        counts, refs = merge_exports(shares)
        assert counts == want.counts and refs == want.refs
print("OK")
"""


def run_rccl(args, fault=None, timeout=180):
    build()
    env = dict(os.environ, COLIBRI_SHARDED_LIB=MOCK)
    env.pop("COLIBRI_NO_RCCL", None)
    if fault:
        env["COLIBRI_FAULT"] = fault
    return subprocess.run([sys.executable, "-c", SCRIPT_RCCL, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")] + [str(a) for a in args], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("mode", ["threads", "procs"])
@pytest.mark.parametrize("world,name,maxlength,thr,kind", [(2, "zipf20k", 5, 2, "u"), (4, "zipf20k", 5, 2, "u"), (8, "zipf20k", 5, 2, "u"), (4, "rand_noempty", 8, 3, "u"), (8, "short_sentences", 5, 2, "u"),
                                                           (4, "empty", 5, 2, "u"), (3, "zipf20k", 5, 2, "u"), (2, "zipf20k", 5, 2, "us"), (4, "zipf20k", 5, 2, "us"), (8, "rand3", 5, 2, "us"),
                                                           (4, "zipf20k", 5, 2, "i"), (4, "zipf20k", 5, 2, "is"), (3, "rand2", 6, 2, "isT1"), (8, "one_token", 5, 2, "is"), (4, "zipf20k", 5, 2, "uW3"), (2, "zipf20k", 4, 2, "iW4")])
def test_the_rccl_back_end_of_the_cxx_driver(mode, world, name, maxlength, thr, kind):
    """ranks on distinct devices: every exchange of host/src/sharded.cpp goes through ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, ncclAllReduce (SUM, MIN) and — one
    trainer per rank, communicators from a unique id, as bench.py --gpus N builds them — ncclAllGather for the ranks' host values. The stand-in's RCCL refuses a send without
    a receive of the same size on the peer it names, and collectives whose ranks disagree: sizes, offsets and orders of the driver's calls are checked, not only their results.
    `threads`: one trainer holding all ranks (ncclCommInitAll); `procs`: one trainer per rank, each on its own Python thread. Both protocols (plain models: key-sharded)."""
    p = run_rccl([world, name, maxlength, thr, repr(KINDS[kind]), mode])
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("mode", ["threads", "procs"])
@pytest.mark.parametrize("fault,kind", [("1:colibri_kshard_count", "u"), ("2:colibri_shard_merge", "us"), ("3:colibri_shard_apply", "is"), ("0:colibri_shard_count", "usT1")])
def test_a_failing_rank_on_the_rccl_back_end(mode, fault, kind):
    """agreed failures leave the communicators intact (the plain model's run is repeated by candidate exchange and succeeds); a failure no agreement follows aborts them
    (ncclCommAbort wakes the peers parked inside a collective): every rank ends, nobody hangs"""
    p = run_rccl([4, "zipf20k", 5, 2, repr(KINDS[kind]), mode], fault=fault, timeout=150)
    if kind == "u":
        assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
        assert "repeating the run with the candidate exchange" in p.stderr
    else:
        assert p.returncode != 0
        assert ("injected fault" in p.stderr + p.stdout) or ("failed on rank(s) " + fault.split(":")[0] in p.stderr + p.stdout), (p.stdout[-800:], p.stderr[-1500:])


# ---- the CLI over the stand-in: PatternModel::train -> device_train_sharded -> rank threads -> the merge of the ranks' exports and forward indexes -> the reference's text ----------
MOCK_CLI = os.path.join(ROOT, "tests", "standin", "bin", "colibri-patternmodeller-mock")


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("case", ["hamlet.u", "hamlet.us", "hamlet.i", "hamlet.is", "zipf20k.us", "zipf20k.is"])
def test_the_cli_trained_over_n_ranks_prints_what_the_reference_prints(case, world):
    """`colibri-patternmodeller --gpus N` (host/src/patternmodeller.cpp, host/include/patternmodel.h train(), host/src/sharded.cpp device_train_sharded) on the CPU stand-in:
    print (with every pattern's reference list, merged from the ranks' forward indexes in rank order), report and histogram equal the text the REAL reference printed
    for its own model of the same corpus (tests/golden/views/, reference include/patternmodel.h:2294-2601, :2907-2959, :3390-3450)"""
    import test_views
    build()
    corpus, flags, cls = test_views.CASES[case]
    for view, vf in test_views.VIEW_FLAGS.items():
        p = subprocess.run([MOCK_CLI, "-f", os.path.join(test_views.GOLD, f"{corpus}.colibri.dat"), "-c", os.path.join(test_views.GOLD, cls), vf, "--gpus", str(world)] + flags,
                           capture_output=True, timeout=300, env=dict(os.environ, COLIBRI_NO_RCCL="1"))
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"sentence-sharded over " + str(world).encode() in p.stderr
        assert (b"Counted key-sharded" in p.stderr) == (case in ("hamlet.u", "hamlet.i") and world in (2, 4)), p.stderr.decode()[-800:]
        test_views.check(case, view, p.stdout)


SCRIPT_CONTRACT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import conftest, oracle
from colibri_amd import capi
payload = conftest.small_corpora()["zipf20k"]
with capi.ShardedTrainer(2, devices=[0, 0]) as tr:  # nothing uploaded yet
    try:
        tr.train(mintokens=2, maxlength=3)
        raise SystemExit("a run without a corpus succeeded")
    except capi.ColibriError as e:
        assert "no corpus uploaded" in str(e), str(e)
    tr.upload_split(payload)  # ... and the same trainer is usable afterwards
    tr.train(mintokens=2, maxlength=3)
    assert tr.export_dict() == oracle.train(payload, 2, 3).counts
with capi.ShardedTrainer(4) as tr:  # RCCL back end; rank 3 fails where no agreement follows: the communicators are aborted
    tr.upload_split(payload)
    try:
        tr.train(mintokens=2, maxlength=4, doskipgrams_exhaustive=1)
        raise SystemExit("the injected fault did not fail the run")
    except capi.ColibriError as e:
        assert "injected fault" in str(e), str(e)
    try:
        tr.train(mintokens=2, maxlength=4)
        raise SystemExit("a trainer whose communicators were aborted ran again")
    except capi.ColibriError as e:
        assert "create a new trainer" in str(e), str(e)
for bad in (0, 65):
    try:
        capi.ShardedTrainer(bad)
        raise SystemExit(f"a trainer of {bad} ranks was created")
    except capi.ColibriError:
        pass
print("OK")
"""


def test_the_trainers_contract_between_runs():
    """include/colibri_sharded.h: a run without a corpus is an error the trainer survives; a trainer whose communicators a failure aborted refuses further runs and says
    what to do; rank counts outside 1..64 are refused"""
    build()
    env = dict(os.environ, COLIBRI_SHARDED_LIB=MOCK, COLIBRI_FAULT="3:colibri_shard_apply")
    env.pop("COLIBRI_NO_RCCL", None)
    p = subprocess.run([sys.executable, "-c", SCRIPT_CONTRACT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


SCRIPT_INDEX = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import conftest, oracle
from colibri_amd import capi
from colibri_amd.dist import merge_exports
world, name, maxlength, thr, kind, protocol = int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), eval(sys.argv[7]), int(sys.argv[8])
payload = conftest.small_corpora()[name]
want = oracle.train(payload, thr, maxlength, **kind)
with capi.ShardedTrainer(world, devices=[0] * world) as tr:
    tr.upload_split(payload)
    tr.set_protocol(protocol)
    st = tr.train(mintokens=thr, maxlength=maxlength, **{k: int(v) for k, v in kind.items()})
    counts, refs = merge_exports([tr.export_local(r) for r in range(world)])
    assert counts == want.counts
    assert refs == want.refs, "reference lists differ"
    any_skip = any("skipgram" in k for k in kind)
    assert tr.info.protocol == (1 if (protocol == 1 or any_skip or world not in (1, 2, 4, 8)) else 0), tr.info.protocol
print("OK")
"""


@pytest.mark.parametrize("protocol", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 3, 4])
@pytest.mark.parametrize("name,maxlength,thr,kind", [("zipf20k", 5, 2, "i"), ("rand3", 5, 2, "is"), ("zipf20k", 4, 2, "isT1"), ("rand_noempty", 6, 2, "i"), ("short_sentences", 5, 1, "i"),
                                                     ("zipf20k", 4, 2, "iW4")])
def test_the_c_face_exports_an_indexed_models_reference_lists(protocol, world, name, maxlength, thr, kind):
    """include/colibri_sharded.h colibri_sharded_export_gids / _index_sizes / _export_index: every rank's patterns by global number and its forward index by global number;
    the runs of a number concatenated in rank order are the pattern's reference list — equal to the oracle's for every pattern, on both protocols (an indexed model without
    skipgrams is counted key-sharded unless the candidate exchange is asked for; a world that is not a power of two always takes the latter)"""
    p = run([world, name, maxlength, thr, repr(KINDS[kind]), protocol], script=SCRIPT_INDEX)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
